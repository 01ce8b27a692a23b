// mesh_program_check.cpp -- TEST ONLY.  Executes the step program mesh::build_plan schedules for a network
// (gym_anm_amd/csrc/anm_mesh.hpp: StepType) on the host, with the semantics the kernel gives it -- within a
// step every lane reads before any lane writes; steps are separated by fences -- on a random block matrix with
// the network's sparsity pattern, and compares the solution with dense Gaussian elimination.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../gym_anm_amd/csrc/anm_mesh.hpp"

using namespace anm;

extern "C" int mesh_program_check(const anm_network_desc* n, uint64_t seed, double* max_err, int32_t* n_steps,
                                  int32_t* n_levels, int32_t* group) {
  mesh::Plan P;
  std::string err;
  if (!mesh::build_plan(*n, P, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return -1; }
  const mesh::Dims& d = P.d;
  const int G = d.G, NB = d.NB, N = 2 * (NB - 1);
  *n_steps = d.n_steps; *n_levels = d.n_levels; *group = G;
  auto I = [&](int f, int l) { return P.hi[size_t(f) * G + l]; };
  uint64_t s = seed * 6364136223846793005ull + 1442695040888963407ull;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return double(s >> 11) / 9007199254740992.0 - 0.5; };
  std::vector<double> S(d.lds_per_env, 0.0), A(size_t(N) * N, 0.0), rhs(N);
  for (int k = 0; k < d.lds_per_env; ++k) S[k] = NAN;          // whatever is read must have been written
  for (int k = 0; k < 6; ++k) S[d.l_zero + k] = 0.0;
  auto set_blk = [&](int b, int i, int j, double scale, double shift) {
    for (int u = 0; u < 2; ++u)
      for (int v = 0; v < 2; ++v) {
        const double x = scale * rnd() + ((u == v) ? shift : 0.0);
        S[d.l_blk + 4 * b + 2 * u + v] = x;
        A[size_t(2 * (i - 1) + u) * N + 2 * (j - 1) + v] = x;
      }
  };
  for (int l = 0; l + 1 < NB; ++l) set_blk(I(mesh::IF_DIAG, l), l + 1, l + 1, 1.0, 8.0);
  for (int br = 0; br < d.NBR; ++br) {
    auto IB = [&](int fld) { return I(fld + (br / G) * mesh::IF_BR_FIELDS, br % G); };   // branch br: slot br / G of lane br % G
    const int f = IB(mesh::IF_BR_F), t = IB(mesh::IF_BR_T);
    if (IB(mesh::IF_BR_BLK_FT) >= 0) { set_blk(IB(mesh::IF_BR_BLK_FT), f, t, 1.0, 0.0); set_blk(IB(mesh::IF_BR_BLK_TF), t, f, 1.0, 0.0); }
  }
  for (int k = 0; k < d.n_fill; ++k)
    for (int u = 0; u < 4; ++u) S[d.l_blk + 4 * P.hi[d.off_fill + k] + u] = 0.0;
  for (int i = 1; i < NB; ++i) {
    for (int u = 0; u < 4; ++u) S[d.l_r + 4 * i + u] = 0.0;
    for (int u = 0; u < 2; ++u) { rhs[2 * (i - 1) + u] = rnd(); S[d.l_r + 4 * i + 2 * u] = rhs[2 * (i - 1) + u]; }
  }

  // ---- the program, as k_mesh interprets it
  struct W { int o; double v; int lane; };
  for (int st = 0; st < d.n_steps; ++st) {
    const int ty = P.hi[d.off_stype + st] & 0xff, run = P.hi[d.off_stype + st] >> 8;
    std::vector<W> writes;
    std::vector<int> read_by(d.lds_per_env, -1);     // lane that read a place in this step (-2: several)
    int lane = 0;
    auto rd = [&](int o) {
      if (o < 0 || o >= d.lds_per_env) return double(NAN);
      read_by[o] = (read_by[o] == -1 || read_by[o] == lane) ? lane : -2;
      return S[o];
    };
    auto ld4 = [&](int o, double* m) { for (int u = 0; u < 4; ++u) m[u] = rd(o + u); };
    auto mul = [&](const double* a, const double* b, double* c) {
      c[0] = a[0] * b[0] + a[1] * b[2]; c[1] = a[0] * b[1] + a[1] * b[3];
      c[2] = a[2] * b[0] + a[3] * b[2]; c[3] = a[2] * b[1] + a[3] * b[3];
    };
    auto inv = [&](const double* D, double* Di) {
      const double det = D[0] * D[3] - D[1] * D[2];
      Di[0] = D[3] / det; Di[1] = -D[1] / det; Di[2] = -D[2] / det; Di[3] = D[0] / det;
    };
    for (int l = 0; l < G; ++l) {
      lane = l;
      unsigned w[4];
      for (int u = 0; u < 4; ++u) w[u] = unsigned(P.hi[d.off_desc + (size_t(st) * G + l) * 4 + u]);
      const int kind = w[0] & 0xffff, o1 = w[0] >> 16, o2 = w[1] & 0xffff, o3 = w[1] >> 16, o4 = w[2] & 0xffff,
                o5 = w[2] >> 16, o6 = w[3] & 0xffff, o7 = w[3] >> 16;
      if (kind == 0) continue;
      // an operation may only ride in a step of its own type
      const int want = (kind == mesh::OP_PROD) ? mesh::ST_PROD : (kind == mesh::OP_SUM) ? mesh::ST_SUM
                       : (kind == mesh::OP_TAIL2) ? mesh::ST_TAIL
                       : (kind >= mesh::OP_COL && kind <= (mesh::OP_COL | mesh::OP_COL_INV1 | mesh::OP_COL_INV2)) ? mesh::ST_COL : -1;
      if (want != ty) return -5;
      if (ty == mesh::ST_PROD) {
        double D[4], Di[4], Aik[4], L[4], X[4], M[4], Z[4];
        ld4(o1, D); inv(D, Di); ld4(o2, Aik); mul(Aik, Di, L); ld4(o3, X); mul(L, X, M); ld4(o6, Z);
        for (int u = 0; u < 4; ++u) writes.push_back(W{o4 + u, Z[u] - M[u], l});
        // the first contribution goes to the destination it read; the others, negated, to a product slot
        const bool parked = o4 >= d.l_m && o4 + 4 <= d.l_m + 4 * d.n_m;
        if (parked ? o6 != d.l_zero : o6 != o4) return -6;
        if (o5) for (int u = 0; u < 4; ++u) writes.push_back(W{o5 + u, Di[u], l});
      } else if (ty == mesh::ST_TAIL) {
        double Da[4], Dai[4], Aab[4], Aba[4], Db[4], Dbi[4], L[4], M[4];
        ld4(o1, Da); inv(Da, Dai); ld4(o2, Aab); ld4(o3, Aba); ld4(o4, Db);
        mul(Aba, Dai, L); mul(L, Aab, M);
        for (int u = 0; u < 4; ++u) Db[u] -= M[u];
        const double ra0 = rd(o5), ra1 = rd(o5 + 2);
        const double rb0 = rd(o6) - (L[0] * ra0 + L[1] * ra1), rb1 = rd(o6 + 2) - (L[2] * ra0 + L[3] * ra1);
        inv(Db, Dbi);
        const double xb0 = Dbi[0] * rb0 + Dbi[1] * rb1, xb1 = Dbi[2] * rb0 + Dbi[3] * rb1;
        const double a0 = ra0 - (Aab[0] * xb0 + Aab[1] * xb1), a1 = ra1 - (Aab[2] * xb0 + Aab[3] * xb1);
        writes.push_back(W{o6, rb0, l}); writes.push_back(W{o6 + 2, rb1, l});
        writes.push_back(W{o5, a0, l}); writes.push_back(W{o5 + 2, a1, l});
        for (int u = 0; u < 4; ++u) {
          writes.push_back(W{o6 - d.l_r + d.l_dinv + u, Dbi[u], l});
          writes.push_back(W{o5 - d.l_r + d.l_dinv + u, Dai[u], l});
        }
      } else if (ty == mesh::ST_SUM) {
        if (run < 1 || run > 4 || o6 > run) return -2;   // o6: how many products this lane really has
        double Z[4];
        ld4(o1, Z);
        const int slots[4] = {o2, o3, o4, o5};
        for (int c = 0; c < (run > 2 ? 4 : run); ++c) {
          if (c >= o6 && slots[c] != d.l_zero) return -7;
          for (int u = 0; u < 4; ++u) Z[u] += rd(slots[c] + u);
        }
        for (int u = 0; u < 4; ++u) writes.push_back(W{o1 + u, Z[u], l});
      } else {
        double a0 = rd(o1), a1 = rd(o1 + 2);
        const int trip[2][3] = {{o2, o3, o4}, {o5, o6, o7}};
        for (int c = 0; c < 2; ++c) {
          double Ab[4], Dk[4];
          ld4(trip[c][0], Ab);
          const double r0 = rd(trip[c][1]), r1 = rd(trip[c][1] + 2);
          ld4(trip[c][2], Dk);
          if (kind & (c == 0 ? mesh::OP_COL_INV1 : mesh::OP_COL_INV2)) { double D[4] = {Dk[0], Dk[1], Dk[2], Dk[3]}; inv(D, Dk); }
          const double x0 = Dk[0] * r0 + Dk[1] * r1, x1 = Dk[2] * r0 + Dk[3] * r1;
          a0 -= Ab[0] * x0 + Ab[1] * x1;
          a1 -= Ab[2] * x0 + Ab[3] * x1;
        }
        writes.push_back(W{o1, a0, l}); writes.push_back(W{o1 + 2, a1, l});
      }
    }
    // no two lanes of a step may write the same place, and nothing a step reads may be written in it by another lane
    // (a violated order also shows up as a wrong solution: the memory starts as NaN)
    std::vector<char> seen(d.lds_per_env, 0);
    for (const W& q : writes) {
      if (q.o < 0 || q.o >= d.lds_per_env || seen[q.o]) return -3;
      seen[q.o] = 1;
      if (q.o >= d.l_zero && q.o < d.l_zero + 6) return -4;
      if (read_by[q.o] != -1 && read_by[q.o] != q.lane) return -9;
    }
    for (const W& q : writes) S[q.o] = q.v;
  }
  // what every bus lane forms when the program has ended: x_i = D_i^-1 r_i (IF_XINV: D_i was never inverted)
  std::vector<double> xs(2 * NB, NAN);
  for (int i = 1; i < NB; ++i) {
    double Di[4];
    if (I(mesh::IF_XINV, i - 1)) {
      const double* D = &S[d.l_blk + 4 * I(mesh::IF_DIAG, i - 1)];
      const double det = D[0] * D[3] - D[1] * D[2];
      Di[0] = D[3] / det; Di[1] = -D[1] / det; Di[2] = -D[2] / det; Di[3] = D[0] / det;
    } else {
      for (int u = 0; u < 4; ++u) Di[u] = S[d.l_dinv + 4 * i + u];
    }
    const double r0 = S[d.l_r + 4 * i], r1 = S[d.l_r + 4 * i + 2];
    xs[2 * i] = Di[0] * r0 + Di[1] * r1;
    xs[2 * i + 1] = Di[2] * r0 + Di[3] * r1;
  }

  // ---- dense reference: Gaussian elimination with partial pivoting
  std::vector<double> M = A, b = rhs;
  for (int c = 0; c < N; ++c) {
    int pv = c;
    for (int r = c + 1; r < N; ++r) if (std::fabs(M[size_t(r) * N + c]) > std::fabs(M[size_t(pv) * N + c])) pv = r;
    if (pv != c) { for (int k = 0; k < N; ++k) std::swap(M[size_t(c) * N + k], M[size_t(pv) * N + k]); std::swap(b[c], b[pv]); }
    for (int r = c + 1; r < N; ++r) {
      const double f = M[size_t(r) * N + c] / M[size_t(c) * N + c];
      for (int k = c; k < N; ++k) M[size_t(r) * N + k] -= f * M[size_t(c) * N + k];
      b[r] -= f * b[c];
    }
  }
  for (int c = N - 1; c >= 0; --c) {
    for (int k = c + 1; k < N; ++k) b[c] -= M[size_t(c) * N + k] * b[k];
    b[c] /= M[size_t(c) * N + c];
  }
  double e = 0.0;
  for (int i = 1; i < NB; ++i)
    for (int u = 0; u < 2; ++u) {
      const double x = xs[2 * i + u];
      const double dd = std::fabs(x - b[2 * (i - 1) + u]);
      e = std::fmax(e, (dd == dd) ? dd : INFINITY);
    }
  *max_err = e;
  return 0;
}

// how a plan is launched (mesh::Launch): out[0..7] = group size, steps per trip, doubles of LDS per environment, wavefronts per
// workgroup, LDS bytes per workgroup, wavefronts per compute unit, wavefronts per SIMD the kernel variant is budgeted for,
// tables staged in LDS (1) or read from global memory (0)
extern "C" int mesh_plan_stats(const anm_network_desc* n, int64_t* out) {
  mesh::Plan P;
  std::string err;
  if (!mesh::build_plan(*n, P, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return -1; }
  const mesh::Launch L = mesh::launch_of(P.d);
  out[0] = P.d.G; out[1] = P.d.n_steps; out[2] = P.d.lds_per_env; out[3] = L.waves;
  out[4] = int64_t(L.lds); out[5] = int64_t(L.waves_per_cu); out[6] = L.simd_waves; out[7] = L.tables_in_lds ? 1 : 0;
  return 0;
}
