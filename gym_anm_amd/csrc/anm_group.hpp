// anm_group.hpp -- lane-group Newton-Raphson for a compile-time TREE topology (gfx950 only).
//
// Why: a diverging power flow runs to the reference's iteration cap (100, solve_load_flow.py:176) while
// every converging solve of the batch is done after <= 9 iterations.  In the thread-per-environment
// kernels such a solve keeps ONE lane of its wavefront busy for ~90 more trips of ~500-665 fp64
// instructions each: the launch time of the headline configuration (65 536 environments = one wave
// per SIMD) is exactly that serial chain.  Here a solve that is still running after `handoff_after`
// iterations leaves its lane and continues spread over a group of G lanes of the same wavefront,
//     lane l of the group  <->  bus l + 1  (and the branch joining it to its parent bus),
// so that one trip costs the wavefront ~1/3 of the instructions: every lane forms the three W products
// of its own tree edge, the tree is eliminated by height (all leaves at once, then their parents, ...)
// and back-substituted by depth.  All hand-overs between lanes are register-to-register
// (`ds_bpermute_b32`: the LDS crossbar without touching LDS memory); nothing is stored anywhere inside the
// loop.  Everything structural (parent, children, height, depth of each bus, position of its admittances
// in the constant buffer) is a constexpr table of the Topo descriptor (codegen.py: tree_tables), so the
// level loops are unrolled and a level only pulls as many child slots as a bus of that height can have.
//
// Same reference semantics and the same formulation as the thread-per-environment kernels (W_ik =
// V_i conj(Y_ik V_k), magnitude columns scaled by |V|, rotated loop, (cos, sin) state: see
// anm_device.hpp); results differ from them by summation order only (~1e-16).  Reference:
//   _newton_raphson_sparse / _f / _dfdx   gym_anm/simulator/solve_load_flow.py:84-226
#pragma once

#include "anm_device.hpp"

#if defined(__HIPCC__)
namespace anm {
namespace group {

// value of `x` in lane `src4 / 4` of the wavefront (every lane must be executing: an inactive source lane
// reads as 0)
__device__ __forceinline__ double lane_get(double x, int src4) {
  const int lo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_get(float x, int src4) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(x)));
}

// LDS slot of one handed-over solve: 5 per-bus arrays + the iteration count / flags
template <class T>
struct Slot {
  static constexpr int VM = 0, CS = T::NB, SN = 2 * T::NB, P = 3 * T::NB, Q = 4 * T::NB, IT = 5 * T::NB,
                       FLAGS = IT + 1, SIZE = FLAGS + 1;
};

template <class T>
struct Shape {
  static constexpr int G = T::GRP;        // lanes per environment
  static constexpr int NG = 64 / G;       // environments per wavefront
};

// Continue the Newton-Raphson solves of the lanes with st.active (they have done st.it iterations and
// hold their iterate in w.vm / st.cs / st.sn) until each converges, fails (NaN) or reaches max_iter --
// NG solves at a time, one per lane group.  On return the owner lanes hold the final iterate, st.it and
// a st.diff that reproduces the reference's flags (NaN: failed; 0: converged; +inf: cap reached).
// `lds`: >= NG * Slot<T>::SIZE doubles, private to this wavefront; every lane of the wave must call.
template <class T, class JT>
__device__ __forceinline__ void continue_in_groups(cptr_t C, EnvWork<T>& w, PFState<T>& st, bool mine, double tol, int max_iter,
                                   double* lds) {
  typedef Layout<T> L;
  typedef Slot<T> S;
  constexpr int G = Shape<T>::G, NG = Shape<T>::NG, NB = T::NB;
  const int lane = threadIdx.x & 63;
  const int l = lane & (G - 1);        // bus l + 1
  const int gb = lane - l;             // first lane of this group
  const int grp = lane / G;
  const int b = (l + 1 < NB) ? l + 1 : 0;
  const bool lane_bus = (l + 1 < NB);

  // ---- per-lane view of the tree (constant tables indexed by the lane: loaded once)
  const int parent = T::T_PARENT[b];                 // 0: slack, -1: padding lane
  const int height = lane_bus ? T::T_HEIGHT[b] : -1;
  const int depth = lane_bus ? T::T_DEPTH[b] : -1;
  const int nch = lane_bus ? T::T_NCH[b] : 0;
  const int psrc4 = 4 * ((lane_bus && parent > 0) ? gb + parent - 1 : lane);
  int csrc4[T::T_MAXCH];
  static_for<0, T::T_MAXCH>([&](auto Cc) {
    const int c = lane_bus ? T::T_CH[b * T::T_MAXCH + Cc] : -1;
    csrc4[Cc] = 4 * (c > 0 ? gb + c - 1 : lane);
  });
  const bool root = lane_bus && parent == 0;         // attached to the slack bus (V_0 = 1)
  double ybb_r = 0, ybb_i = 0, ybp_r = 0, ybp_i = 0, ypb_r = 0, ypb_i = 0;
  if (lane_bus) {
    ybb_r = C[L::Y_RE + T::T_ZBB[b]]; ybb_i = C[L::Y_IM + T::T_ZBB[b]];
    ybp_r = C[L::Y_RE + T::T_ZBP[b]]; ybp_i = C[L::Y_IM + T::T_ZBP[b]];
    ypb_r = C[L::Y_RE + T::T_ZPB[b]]; ypb_i = C[L::Y_IM + T::T_ZPB[b]];
  }
  const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << gb;

  // ---- which lanes hand over, in lane order; NG at a time
  const unsigned long long todo = __ballot(mine);
  const int n_todo = __popcll(todo);
  const int rank = __popcll(todo & ((1ull << lane) - 1ull));
  for (int base = 0; base < n_todo; base += NG) {
    const bool send = mine && rank >= base && rank < base + NG;
    if (send) {
      double* s = lds + (rank - base) * S::SIZE;
      static_for<1, NB>([&](auto I) {
        constexpr int i = I;
        s[S::VM + i] = w.vm[i]; s[S::CS + i] = st.cs[i]; s[S::SN + i] = st.sn[i];
        s[S::P + i] = w.bus_p[i]; s[S::Q + i] = w.bus_q[i];
      });
      s[S::IT] = double(st.it);
    }
    ANM_WAVE_SYNC();
    const bool gvalid = base + grp < n_todo;           // this group holds a solve
    const bool isbus = lane_bus && gvalid;
    double vm = 1.0, cs = 1.0, sn = 0.0, bus_p = 0.0, bus_q = 0.0;
    int it = 0;
    {
      const double* s = lds + grp * S::SIZE;
      if (isbus) { vm = s[S::VM + b]; cs = s[S::CS + b]; sn = s[S::SN + b]; bus_p = s[S::P + b]; bus_q = s[S::Q + b]; }
      if (gvalid) it = int(s[S::IT]);
    }
    ANM_WAVE_SYNC();

    // ---- the reference's loop, rotated like pf_iterate (anm_device.hpp): evaluate; account for the update of
    // the previous trip; leave when no group runs; update
    bool f_gt = false, f_nan = false;   // ||F||inf > tol / F has a NaN, for the iterate of this group
    bool inc = false, take = gvalid;
    for (;;) {
      const double vr = vm * cs, vi = vm * sn;
      double vpr = lane_get(vr, psrc4), vpi = lane_get(vi, psrc4);
      vpr = root ? 1.0 : vpr;
      vpi = root ? 0.0 : vpi;
      // W_bb = conj(Y_bb) vm^2;  P = V_b conj(V_p):  W_bp = conj(Y_bp) P,  W_pb = conj(Y_pb) conj(P)
      const double m2 = vm * vm;
      const double wbb_r = ybb_r * m2, wbb_i = -(ybb_i * m2);
      const double pr = fma(vr, vpr, vi * vpi), pim = fma(vi, vpr, -(vr * vpi));
      const double wbp_r = fma(ybp_r, pr, ybp_i * pim), wbp_i = fma(ybp_r, pim, -(ybp_i * pr));
      const double wpb_r = fma(ypb_r, pr, -(ypb_i * pim)), wpb_i = -fma(ypb_r, pim, ypb_i * pr);
      // S_b = W_bb + W_bp + sum over the children c of W_pb(c)
      double sr = wbb_r + wbp_r, si = wbb_i + wbp_i;
      static_for<0, T::T_MAXCH>([&](auto Cc) {
        const double cr = lane_get(wpb_r, csrc4[Cc]), ci = lane_get(wpb_i, csrc4[Cc]);
        const bool has = Cc < nch;
        sr += has ? cr : 0.0;
        si += has ? ci : 0.0;
      });
      const double fr = sr - bus_p, fi = si - bus_q;
      // group-wide stop test: ||F||inf > tol and "F has a NaN" are all the reference's loop and flags need
      const bool gt = isbus && (fmax(fabs(fr), fabs(fi)) > tol);
      const bool nn = isbus && ((fr != fr) || (fi != fi));
      const bool any_gt = (__ballot(gt) & gmask) != 0ull;
      const bool any_nn = (__ballot(nn) & gmask) != 0ull;
      it = inc ? it + 1 : it;
      f_gt = take ? any_gt : f_gt;
      f_nan = take ? any_nn : f_nan;
      const bool run = gvalid && f_gt && !f_nan && (it < max_iter);   // NaN > tol is false, like the reference
      if (!__any(run)) break;

      // ---- Jacobian blocks (anm_device.hpp: newton_update): own diagonal and the two couplings with the parent
      Blk<JT> Dg = Blk<JT>{JT(-(si - wbb_i)), JT(sr + wbb_r), JT(sr - wbb_r), JT(si + wbb_i)};
      const Blk<JT> Jbp = Blk<JT>{JT(wbp_i), JT(wbp_r), JT(-wbp_r), JT(wbp_i)};
      const Blk<JT> Jpb = Blk<JT>{JT(wpb_i), JT(wpb_r), JT(-wpb_r), JT(wpb_i)};
      JT r0 = JT(fr), r1 = JT(fi);
      // ---- elimination by height: a bus folds the Schur complements and reduced right-hand sides its
      // children published (registers Sc / Lr of the child lanes), inverts its pivot and publishes its own
      Blk<JT> Sc = Blk<JT>{JT(0), JT(0), JT(0), JT(0)};
      JT Lr0 = JT(0), Lr1 = JT(0);
      static_for<0, T::T_MAXH + 1>([&](auto H) {
        constexpr int h = H;
        constexpr int NC = T::T_NCH_H[h];
        JT ga[NC > 0 ? NC : 1], gbb[NC > 0 ? NC : 1], gc[NC > 0 ? NC : 1], gd[NC > 0 ? NC : 1], g0[NC > 0 ? NC : 1],
            g1[NC > 0 ? NC : 1];
        static_for<0, NC>([&](auto Cc) {
          ga[Cc] = lane_get(Sc.a, csrc4[Cc]); gbb[Cc] = lane_get(Sc.b, csrc4[Cc]);
          gc[Cc] = lane_get(Sc.c, csrc4[Cc]); gd[Cc] = lane_get(Sc.d, csrc4[Cc]);
          g0[Cc] = lane_get(Lr0, csrc4[Cc]); g1[Cc] = lane_get(Lr1, csrc4[Cc]);
        });
        if (height == h) {
          static_for<0, NC>([&](auto Cc) {
            if (Cc < nch) {
              Dg.a -= ga[Cc]; Dg.b -= gbb[Cc]; Dg.c -= gc[Cc]; Dg.d -= gd[Cc];
              r0 -= g0[Cc]; r1 -= g1[Cc];
            }
          });
          Dg = blk_inv(Dg);
          const Blk<JT> Lk = blk_mul(Jpb, Dg);
          Sc = blk_mul(Lk, Jbp);
          Lr0 = fm(Lk.a, r0, Lk.b * r1);
          Lr1 = fm(Lk.c, r0, Lk.d * r1);
        }
      });
      // ---- back substitution by depth (Dg now holds the inverted pivots)
      JT d0 = JT(0), d1 = JT(0);
      static_for<0, T::T_MAXD + 1>([&](auto Dd) {
        constexpr int dd = Dd;
        JT p0 = JT(0), p1 = JT(0);
        if constexpr (dd > 0) {
          p0 = lane_get(d0, psrc4);
          p1 = lane_get(d1, psrc4);
        }
        if (depth == dd) {
          JT a0 = r0, a1 = r1;
          if constexpr (dd > 0) {
            a0 = fm(-Jbp.b, p1, fm(-Jbp.a, p0, a0));
            a1 = fm(-Jbp.d, p1, fm(-Jbp.c, p0, a1));
          }
          d0 = fm(Dg.a, a0, Dg.b * a1);
          d1 = fm(Dg.c, a0, Dg.d * a1);
        }
      });
      // ---- update (group-uniform `run`); d1 is the relative magnitude step.  Rotation of (cos, sin) by the
      // angle step, path chosen per wavefront exactly as update_angles does
      const bool upd = run && isbus;
      const double dth = upd ? double(d0) : 0.0;
      const bool small_step = fabs(dth) <= 0.78;  // false for NaN
      double sd_, cd_;
      if (!__any(!small_step)) {
        sincos_kernel(dth, 0, sd_, cd_);
      } else if (fabs(dth) < 4.0e15) {
        sincos_medium(dth, sd_, cd_);
      } else {
        const SinCos r = sincos_huge(dth);
        sd_ = r.s;
        cd_ = r.c;
      }
      if (upd) {
        vm = fma(-double(d1), fabs(vm), vm);
        const double c0 = cs, s0 = sn;
        cs = fma(c0, cd_, s0 * sd_);
        sn = fma(s0, cd_, -(c0 * sd_));
      }
      inc = take = run;
    }

    // ---- results back to the owner lanes
    if (isbus) {
      double* s = lds + grp * S::SIZE;
      s[S::VM + b] = vm; s[S::CS + b] = cs; s[S::SN + b] = sn;
      if (l == 0) {
        s[S::IT] = double(it);
        s[S::FLAGS] = f_nan ? NAN : (f_gt ? INFINITY : 0.0);
      }
    }
    ANM_WAVE_SYNC();
    if (send) {
      const double* s = lds + (rank - base) * S::SIZE;
      static_for<1, NB>([&](auto I) {
        constexpr int i = I;
        w.vm[i] = s[S::VM + i]; st.cs[i] = s[S::CS + i]; st.sn[i] = s[S::SN + i];
        w.vr[i] = w.vm[i] * st.cs[i];   // the iterate's V, as eval_mismatch leaves it (pf_end reads it)
        w.vi[i] = w.vm[i] * st.sn[i];
      });
      st.it = int(s[S::IT]);
      st.diff = s[S::FLAGS];
      st.active = false;
    }
    ANM_WAVE_SYNC();
  }
}

}  // namespace group
}  // namespace anm
#endif  // __HIPCC__
