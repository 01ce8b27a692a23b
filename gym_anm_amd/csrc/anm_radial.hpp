// anm_radial.hpp -- lane-group kernel for RADIAL (tree) networks of any size up to 64 buses.
//
// The thread-per-environment kernels (anm_device.hpp) keep a whole environment in one thread's
// registers; that is the fastest mapping while the working set fits (ANM6: 0 bytes of scratch)
// but a 30-bus feeder needs ~600 doubles per environment and spills (512 registers + 4.7 KB of
// scratch per lane).  Here one environment is spread over a group of G = 8/16/32/64 lanes of ONE
// wavefront; lane l plays three roles at once:
//     bus l+1 (and the branch that joins it to its parent bus),   device l.
// The tree structure replaces the general sparse LU: every Newton iteration is
//   V, E  -> LDS;  each lane sums its row of Y V from its parent and its children (LDS gathers);
//   mismatch + group-wide inf-norm (shuffle butterfly);
//   2x2 Jacobian blocks: own diagonal + the two couplings with the parent bus;
//   block elimination leaves -> roots, one tree level per step (children publish their Schur
//   complement and reduced right-hand side in LDS, parents gather them in bus order);
//   back substitution roots -> leaves.
// Same reference semantics as anm_device.hpp (see the citations there); results differ from the
// thread-per-environment kernels only by summation order (~1e-16).  Nothing here is compile-time
// specialised: the per-lane tables built by build_plan() drive one generic kernel.
#pragma once

#include <queue>
#include <string>
#include <vector>

#include "anm_device.hpp"
#include "anm_env_ops.hpp"
#include "anm_pack.hpp"

namespace anm {
namespace radial {

enum IField : int { IF_PARENT = 0, IF_DEPTH, IF_CH_BEG, IF_CH_END, IF_BD_BEG, IF_BD_END, IF_BR_INDEX, IF_BR_FROM,
                    IF_DEV_TYPE, IF_DEV_SLOT, IF_DEV_SET, IF_COUNT };
enum DField : int { DF_YBB_RE = 0, DF_YBB_IM, DF_YBP_RE, DF_YBP_IM, DF_YPB_RE, DF_YPB_IM, DF_VMIN, DF_VMAX, DF_BRC,
                    DF_COUNT = DF_BRC + 9 };
enum SField : int { SF_BASE = 0, SF_DT, SF_LAMB, SF_C1, SF_C2, SF_RTERM, SF_PERIOD, SF_Y00_RE, SF_Y00_IM,
                    SF_SLACK_VMIN, SF_SLACK_VMAX, SF_COUNT = 16 };
constexpr int DEV_NONE = -99;
constexpr int KMAX = 8;

struct Dims {
  int G, NB, ND, NBR, NLOAD, NGEN, NDES, NSET, SDIM, FS, max_depth, slack_dev;
  int off_lists, n_lists;              // ints: [IF_COUNT][G] then child / bus-device lists (n_lists <= 256)
  int off_lane, off_dev, off_obs_lo, off_obs_hi, n_double;  // doubles
  // offsets inside one row of the full-state dump (same order as FullState<T>)
  int f_bus_p, f_bus_q, f_bus_vm, f_bus_va, f_bus_im, f_bus_ia, f_dev_p, f_dev_q, f_des_soc, f_gen_pmax, f_br_p,
      f_br_q, f_br_s, f_br_im, f_br_ia;
};

struct Plan {
  Dims d;
  std::vector<int> hi;
  std::vector<double> hd;
  const int* di = nullptr;     // device copies
  const double* dd = nullptr;
};

inline bool is_radial(const anm_network_desc& n) {
  if (n.n_branch != n.n_bus - 1 || n.n_bus < 2 || n.n_bus - 1 > 64 || n.n_dev > 64) return false;
  std::vector<std::vector<int>> adj(n.n_bus);
  for (int b = 0; b < n.n_branch; ++b) {
    adj[n.br_from[b]].push_back(n.br_to[b]);
    adj[n.br_to[b]].push_back(n.br_from[b]);
  }
  std::vector<int> seen(n.n_bus, 0);
  std::queue<int> q;
  q.push(0);
  seen[0] = 1;
  int cnt = 1;
  while (!q.empty()) {
    int u = q.front();
    q.pop();
    for (int v : adj[u])
      if (!seen[v]) { seen[v] = 1; ++cnt; q.push(v); }
  }
  return cnt == n.n_bus;
}

// Per-lane tables for one network, built from the description alone (same formulas as pack_constants).
inline bool build_plan(const anm_network_desc& n, Plan& P, std::string& err) {
  if (!is_radial(n)) { err = "network is not a radial tree with <= 64 buses/devices"; return false; }
  Dims& d = P.d;
  d.NB = n.n_bus; d.ND = n.n_dev; d.NBR = n.n_branch;
  d.NLOAD = d.NGEN = d.NDES = 0;
  std::vector<int> slot(n.n_dev, -1), sset(n.n_dev, -1);
  d.slack_dev = -1;
  for (int k = 0; k < n.n_dev; ++k) {
    const int t = n.dev_type[k];
    if (t == DEV_LOAD) slot[k] = d.NLOAD++;
    else if (t == DEV_CLASSICAL || t == DEV_RENEWABLE) slot[k] = d.NGEN++;
    else if (t == DEV_STORAGE) slot[k] = d.NDES++;
    else d.slack_dev = k;
  }
  d.NSET = 0;
  for (int k = 0; k < n.n_dev; ++k)
    if (n.dev_type[k] == DEV_CLASSICAL || n.dev_type[k] == DEV_RENEWABLE || n.dev_type[k] == DEV_STORAGE) sset[k] = d.NSET++;
  d.SDIM = 2 * d.ND + d.NDES + d.NGEN;
  int need = std::max(std::max(d.NB - 1, d.ND), d.NBR);
  d.G = 8;
  while (d.G < need) d.G *= 2;
  const int G = d.G;
  // full-state layout
  d.f_bus_p = 0; d.f_bus_q = d.f_bus_p + d.NB; d.f_bus_vm = d.f_bus_q + d.NB; d.f_bus_va = d.f_bus_vm + d.NB;
  d.f_bus_im = d.f_bus_va + d.NB; d.f_bus_ia = d.f_bus_im + d.NB; d.f_dev_p = d.f_bus_ia + d.NB;
  d.f_dev_q = d.f_dev_p + d.ND; d.f_des_soc = d.f_dev_q + d.ND; d.f_gen_pmax = d.f_des_soc + d.NDES;
  d.f_br_p = d.f_gen_pmax + d.NGEN; d.f_br_q = d.f_br_p + d.NBR; d.f_br_s = d.f_br_q + d.NBR;
  d.f_br_im = d.f_br_s + d.NBR; d.f_br_ia = d.f_br_im + d.NBR; d.FS = d.f_br_ia + d.NBR;

  // tree rooted at the slack bus: BFS gives parent bus and the branch to it
  std::vector<std::vector<std::pair<int, int>>> adj(d.NB);  // (neighbour, branch)
  for (int b = 0; b < d.NBR; ++b) {
    adj[n.br_from[b]].push_back({n.br_to[b], b});
    adj[n.br_to[b]].push_back({n.br_from[b], b});
  }
  std::vector<int> parent(d.NB, -2), pbranch(d.NB, -1), depth(d.NB, -1);
  std::queue<int> q;
  q.push(0);
  parent[0] = -1;
  while (!q.empty()) {
    int u = q.front();
    q.pop();
    for (auto& e : adj[u])
      if (parent[e.first] == -2) {
        parent[e.first] = u;
        pbranch[e.first] = e.second;
        depth[e.first] = (u == 0) ? 0 : depth[u] + 1;
        q.push(e.first);
      }
  }
  d.max_depth = 0;
  for (int b = 1; b < d.NB; ++b) d.max_depth = std::max(d.max_depth, depth[b]);

  // admittances exactly like build_ybus / pack_constants
  std::vector<cplx> Y(size_t(d.NB) * d.NB, cplx(0, 0));
  std::vector<cplx> aff(d.NBR), aft(d.NBR), att(d.NBR), atf(d.NBR);
  for (int b = 0; b < d.NBR; ++b) {
    const int f = n.br_from[b], t = n.br_to[b];
    const cplx ys(n.br_series[2 * b], n.br_series[2 * b + 1]), sh(n.br_shunt[2 * b], n.br_shunt[2 * b + 1]);
    const cplx tap(n.br_tap[2 * b], n.br_tap[2 * b + 1]);
    Y[f * d.NB + t] = -ys / std::conj(tap);
    Y[t * d.NB + f] = -ys / tap;
    Y[f * d.NB + f] += (ys + sh) / (std::abs(tap) * std::abs(tap));
    Y[t * d.NB + t] += ys + sh;
    aff[b] = (ys + sh) / (std::abs(tap) * std::abs(tap));
    aft[b] = -ys / std::conj(tap);
    att[b] = ys + sh;
    atf[b] = -ys / tap;
  }

  // ---- int tables
  P.hi.assign(size_t(IF_COUNT) * G, 0);
  auto I = [&](int f, int l) -> int& { return P.hi[size_t(f) * G + l]; };
  std::vector<int> lists;
  for (int l = 0; l < G; ++l) {
    I(IF_PARENT, l) = -1; I(IF_DEPTH, l) = -1; I(IF_BR_INDEX, l) = -1; I(IF_DEV_TYPE, l) = DEV_NONE;
    I(IF_DEV_SLOT, l) = -1; I(IF_DEV_SET, l) = -1;
  }
  for (int l = 0; l + 1 < d.NB; ++l) {
    const int b = l + 1;
    I(IF_PARENT, l) = parent[b] == 0 ? -1 : parent[b] - 1;
    I(IF_DEPTH, l) = depth[b];
    I(IF_BR_INDEX, l) = pbranch[b];
    I(IF_BR_FROM, l) = n.br_from[pbranch[b]] == b ? 1 : 0;
    I(IF_CH_BEG, l) = int(lists.size());
    for (int c = 1; c < d.NB; ++c)
      if (parent[c] == b) lists.push_back(c - 1);  // ascending bus order
    I(IF_CH_END, l) = int(lists.size());
  }
  for (int l = 0; l + 1 < d.NB; ++l) {
    const int b = l + 1;
    I(IF_BD_BEG, l) = int(lists.size());
    for (int k = 0; k < d.ND; ++k)
      if (n.dev_bus[k] == b) lists.push_back(k);  // ascending device order (simulator.py:547-549)
    I(IF_BD_END, l) = int(lists.size());
  }
  for (int k = 0; k < d.ND; ++k) {
    I(IF_DEV_TYPE, k) = n.dev_type[k];
    I(IF_DEV_SLOT, k) = slot[k];
    I(IF_DEV_SET, k) = sset[k];
  }
  d.off_lists = int(P.hi.size());
  d.n_lists = int(lists.size());
  if (d.n_lists > 256) { err = "radial plan: index lists exceed the LDS table"; return false; }
  P.hi.insert(P.hi.end(), lists.begin(), lists.end());
  P.hi.push_back(0);

  // ---- double tables
  d.off_lane = SF_COUNT;
  d.off_dev = d.off_lane + DF_COUNT * G;
  d.off_obs_lo = d.off_dev + d.ND * SD_SIZE;
  d.off_obs_hi = d.off_obs_lo + d.SDIM + KMAX;
  d.n_double = d.off_obs_hi + d.SDIM + KMAX;
  const double inf = std::numeric_limits<double>::infinity();
  P.hd.assign(d.n_double, 0.0);
  P.hd[SF_BASE] = n.base_mva; P.hd[SF_DT] = n.delta_t; P.hd[SF_LAMB] = n.lamb;
  P.hd[SF_C1] = inf; P.hd[SF_C2] = inf; P.hd[SF_RTERM] = -inf;
  P.hd[SF_Y00_RE] = Y[0].real(); P.hd[SF_Y00_IM] = Y[0].imag();
  P.hd[SF_SLACK_VMIN] = n.bus_vmin[0]; P.hd[SF_SLACK_VMAX] = n.bus_vmax[0];
  for (int k = 0; k < d.SDIM + KMAX; ++k) { P.hd[d.off_obs_lo + k] = -inf; P.hd[d.off_obs_hi + k] = inf; }
  auto D = [&](int f, int l) -> double& { return P.hd[d.off_lane + size_t(f) * G + l]; };
  for (int l = 0; l + 1 < d.NB; ++l) {
    const int b = l + 1, p = parent[b], br = pbranch[b];
    D(DF_YBB_RE, l) = Y[b * d.NB + b].real(); D(DF_YBB_IM, l) = Y[b * d.NB + b].imag();
    D(DF_YBP_RE, l) = Y[b * d.NB + p].real(); D(DF_YBP_IM, l) = Y[b * d.NB + p].imag();
    D(DF_YPB_RE, l) = Y[p * d.NB + b].real(); D(DF_YPB_IM, l) = Y[p * d.NB + b].imag();
    D(DF_VMIN, l) = n.bus_vmin[b]; D(DF_VMAX, l) = n.bus_vmax[b];
    const cplx c4[4] = {aff[br], aft[br], att[br], atf[br]};
    for (int j = 0; j < 4; ++j) { D(DF_BRC + 2 * j, l) = c4[j].real(); D(DF_BRC + 2 * j + 1, l) = c4[j].imag(); }
    D(DF_BRC + 8, l) = n.br_rate[br];
  }
  // device constants: the per-device packing of anm_pack.hpp, so that the projection tables are
  // computed by exactly the same code as for the thread-per-environment kernels
  for (int k = 0; k < d.ND; ++k) {
    double* sd = &P.hd[d.off_dev + k * SD_SIZE];
    pack_device(n, k, sd);
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__)

using View = anm::View;   // (anm_env_ops.hpp)

struct IO {
  int mode;  // 0 transition, 1 reset, 2 step
  TransitionIO t;
  EnvIO e;
  View v;
};

// A workgroup is exactly one wavefront and a wavefront's LDS operations complete in program
// order, so publishing values for the other lanes needs no s_barrier and no counter drain: only
// the compiler must be kept from moving LDS accesses across the hand-over point.
#define ANM_GROUP_SYNC()                                         \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)

enum { A_VR = 0, A_VI, A_UPR, A_UPI, A_S0, A_S1, A_N = 10 };  // LDS of a wavefront: 10 x 64 doubles (see k_radial)

#ifndef ANM_RADIAL_WAVES
#define ANM_RADIAL_WAVES 1
#endif
// TT: the compiled topology (Topo) when the model IS that network and it is a tree -- the Newton loop is then
// the compile-time specialised lane-group loop of anm_group.hpp (unrolled levels, register hand-overs) instead
// of the table-driven one below; void: any radial network (generic mode).
// PG: the parameter class is that of the lane group's own environment (any assignment of classes to environments;
// the constants are then read by vector loads) instead of one class per wavefront (scalar loads).
template <class JT, class TT, bool PG = false>
__global__ __launch_bounds__(64, ANM_RADIAL_WAVES) void k_radial(Dims d, const int* __restrict__ ri, const double* __restrict__ rd0, IO io,
                                              SolverOpts so, int64_t n_env, ClassSel cls) {
  constexpr bool USE_T = !std::is_void<TT>::value;
  // parameter class of this wavefront's environments (uniform over aligned blocks of 64 environments), or, PG, of
  // this lane group's environment
  const int64_t first_env = int64_t(blockIdx.x) * (64 / d.G);
  const int64_t my_env = first_env + int(threadIdx.x) / d.G;
  const double* __restrict__ rd =
      PG ? rd0 + int64_t(cls.env_class[my_env < n_env ? my_env : 0]) * cls.stride
         : rd0 + int64_t(__builtin_amdgcn_readfirstlane(cls.env_class[(first_env < n_env ? first_env : 0) * cls.per_env])) * cls.stride;
  // Hand-overs between the lanes of an environment go through LDS in 16-BYTE units -- (re, im) pairs, and the six values
  // a parent folds as three of them -- so that every access is a ds_read_b128 / ds_write_b128 (four LDS cycles per
  // wavefront instruction; two 8-byte accesses 64 slots apart, what separate arrays give, take eight): the wavefronts of
  // a compute unit share ONE LDS pipe, and the Newton loop keeps it busier than any VALU (config 4: 82 -> 76 us).
  //   pV[lane]: V;  pU[lane]: W_pb, later the Newton step;  pS[3 lane + 0 .. 2]: (Sc.a, Sc.b), (Sc.c, Sc.d), (Lr0, Lr1)
  //   -- the layout group::newton_groups<.., LDSX> uses on the same memory.  What happens once per transition (device
  //   sums before the loop, the currents after it) keeps plain arrays sh[A_*][lane] on that memory: the phases do not overlap.
  __shared__ __align__(16) double sh[A_N][64];
  double2* const pV = reinterpret_cast<double2*>(&sh[0][0]);
  double2* const pU = pV + 64;
  double2* const pS = pV + 128;
  __shared__ int sh_lists[256];  // children / bus-device index lists (read every Newton iteration)
  // list-form observation gathered in this kernel (anm_model_set_obs; step mode): one row of the electrical state per
  // environment, FS + KMAX doubles (the layout of `full`, the aux variables behind it) -- dynamic, 0 bytes without a list
  extern __shared__ double sh_obs_rows[];
  const int t = threadIdx.x;
  const int G = d.G;
  const int l = t & (G - 1);
  const int gb = t - l;
  const int per_wave = 64 / G;
  // slot of this launch -> environment: itself, or (a view is bound: the launch serves a SUB-batch of a larger batch whose
  // rows are padded to common widths, anm_model_bind_view) the index the view names; the row strides of the batch arrays are
  // the network's own widths unless the view gives others
  const int64_t slot_e = int64_t(blockIdx.x) * per_wave + (t / G);
  const bool env_ok = slot_e < n_env;
  const int64_t ee = io.v.index ? int64_t(io.v.index[env_ok ? slot_e : 0]) : (env_ok ? slot_e : 0);  // inactive groups compute on another env and never store
  const int64_t e = ee;
  const int W_LOAD = io.v.w_load > 0 ? io.v.w_load : d.NLOAD, W_GEN = io.v.w_gen > 0 ? io.v.w_gen : d.NGEN;
  const int W_SET = io.v.w_set > 0 ? io.v.w_set : d.NSET, W_DES = io.v.w_des > 0 ? io.v.w_des : d.NDES;
  const int W_ACT = io.v.w_action > 0 ? io.v.w_action : 2 * (d.NGEN + d.NDES);
  const int W_ST = io.v.w_state > 0 ? io.v.w_state : d.SDIM + io.e.K;
  const int W_EXO = io.v.w_exo > 0 ? io.v.w_exo : d.NLOAD + d.NGEN, W_AUX = io.v.w_aux > 0 ? io.v.w_aux : io.e.K;
  const int W_FS = io.v.w_full > 0 ? io.v.w_full : d.FS;
  auto RI = [&](int f) { return ri[f * G + l]; };
  auto RD = [&](int f) { return rd[d.off_lane + f * G + l]; };
  cptr_t C = (cptr_t)rd;
  const double base = rd[SF_BASE], dt = rd[SF_DT];

  const bool isbus = l < d.NB - 1;
  const int parent = RI(IF_PARENT), depth = RI(IF_DEPTH);
  const int ch_beg = RI(IF_CH_BEG), ch_end = RI(IF_CH_END);
  const int typ = RI(IF_DEV_TYPE), slot = RI(IF_DEV_SLOT), sset = RI(IF_DEV_SET);
  {
    const int* glists = ri + d.off_lists;
    for (int k = t; k < d.n_lists; k += 64) sh_lists[k] = glists[k];
    ANM_GROUP_SYNC();
  }
  const int* lists = sh_lists;
  const int mode = io.mode;
  const int K = io.e.K;
  const int S = d.SDIM + K;
  // The constants of this lane's device (limits, the static part of the projection: SD_SIZE doubles), asked for NOW: their
  // address depends on the lane alone, and the ~35 loads the device maps would otherwise issue inside their type branches --
  // after the inputs have arrived -- are a second trip to memory on every wavefront's critical path.  (ANM_RADIAL_PREFETCH=0:
  // tuning switch, the loads where they are used.)
#ifndef ANM_RADIAL_VPOLY
#define ANM_RADIAL_VPOLY 1   // the update's polynomials as in the thread family's lane-group loop (group::newton_groups: VPOLY)
#endif
#ifndef ANM_RADIAL_PREFETCH
#define ANM_RADIAL_PREFETCH 1
#endif
  double sdr[ANM_RADIAL_PREFETCH ? SD_SIZE : 1];
  if constexpr (ANM_RADIAL_PREFETCH != 0) {
    cptr_t row = C + d.off_dev + (l < d.ND ? l : d.ND - 1) * SD_SIZE;
    static_for<0, SD_SIZE>([&](auto Kk) { sdr[Kk] = row[Kk]; });
  }

  ANM_PHASE(0);
  // ---------------- inputs per device lane -------------------------------------------------
  bool skip = false;        // env in the absorbing terminal state (step mode, no autoreset)
  bool resetting = false;
  bool sampled = false;     // initial state drawn by the in-kernel RNG (autoreset, or reset without init_state)
  int aux = 0;
  double in_p = 0.0, in_q = 0.0, in_pot = 0.0, soc = 0.0, s0_q = 0.0;
  double soc_req = 0.0;     // reset: requested SoC (MWh slot)
  if (mode == 0) {
    if (typ == DEV_LOAD) in_p = io.t.p_load[ee * W_LOAD + slot];
    else if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) {
      in_pot = io.t.p_pot[ee * W_GEN + slot];
      in_p = io.t.p_set[ee * W_SET + sset];
      in_q = io.t.q_set[ee * W_SET + sset];
    } else if (typ == DEV_STORAGE) {
      in_p = io.t.p_set[ee * W_SET + sset];
      in_q = io.t.q_set[ee * W_SET + sset];
      soc = io.t.soc[ee * W_DES + slot];
    }
  } else {
    const bool was_term = (mode == 2) && io.e.terminated[ee] != 0;
    const bool series = io.e.exo == nullptr;
    resetting = (mode == 1) || (was_term && io.e.autoreset && series);
    skip = (mode == 2) && was_term && !resetting;
    if (mode == 1 && io.e.mask && !io.e.mask[ee]) skip = true;
    const double* s0 = nullptr;
    double s0_p = 0.0, s0_pm = 0.0;
    sampled = resetting && !(mode == 1 && io.e.init_state);
    if (mode == 1 && io.e.init_state) {
      s0 = io.e.init_state + ee * W_ST;
      if (typ != DEV_NONE) { s0_p = s0[l]; s0_q = s0[d.ND + l]; }
      if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) s0_pm = s0[2 * d.ND + d.NDES + slot];
      if (typ == DEV_STORAGE) soc_req = s0[2 * d.ND + slot];
    } else if (resetting) {  // autoreset: ANM6Easy.init_state with the counter-based RNG
      const uint32_t epoch = uint32_t(io.e.reset_count[ee]);
      uint32_t r[4];
      Philox::generate(io.e.rng_seed, io.e.env_offset + uint64_t(ee), epoch, 0u, r);
      aux = int((uint64_t(r[0]) * uint64_t(io.e.period)) >> 32);
      cptr_t sd = C + d.off_dev + l * SD_SIZE;
      if (typ == DEV_LOAD) s0_p = io.e.series[slot * io.e.period + aux];
      else if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) {
        const int u = slot;
        uint32_t qd[4];
        Philox::generate(io.e.rng_seed, io.e.env_offset + uint64_t(ee), epoch, 1u + u / 2, qd);
        const double uu = Philox::u01(qd[2 * (u % 2)], qd[2 * (u % 2) + 1]);
        s0_p = s0_pm = io.e.series[(d.NLOAD + slot) * io.e.period + aux];
        s0_q = sd[SD_QMIN] + (sd[SD_QMAX] - sd[SD_QMIN]) * uu;
      } else if (typ == DEV_STORAGE) {
        const int u = d.NGEN + slot;
        uint32_t qd[4];
        Philox::generate(io.e.rng_seed, io.e.env_offset + uint64_t(ee), epoch, 1u + u / 2, qd);
        const double uu = Philox::u01(qd[2 * (u % 2)], qd[2 * (u % 2) + 1]);
        soc_req = sd[SD_SOC_MIN] + (sd[SD_SOC_MAX] - sd[SD_SOC_MIN]) * uu;
      }
    }
    if (resetting) {
      in_p = s0_p;
      in_q = s0_q;
      in_pot = s0_pm;
      if (typ == DEV_STORAGE) {
        cptr_t sd = C + d.off_dev + l * SD_SIZE;
        soc = (s0_p <= 0.0) ? sd[SD_SOC_MIN] : sd[SD_SOC_MAX];  // simulator.py:273-278
      }
    } else if (!skip) {
      const double* a = io.e.action + ee * W_ACT;
      if (series) {
        const double av = io.e.state[ee * W_ST + d.SDIM];
        aux = int(fmod(av + 1.0, double(io.e.period)));
        if (typ == DEV_LOAD) in_p = io.e.series[slot * io.e.period + aux];
        else if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) in_pot = io.e.series[(d.NLOAD + slot) * io.e.period + aux];
      } else {
        if (typ == DEV_LOAD) in_p = io.e.exo[ee * W_EXO + slot];
        else if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) in_pot = io.e.exo[ee * W_EXO + d.NLOAD + slot];
      }
      if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) { in_p = a[slot]; in_q = a[d.NGEN + slot]; }
      else if (typ == DEV_STORAGE) {
        in_p = a[2 * d.NGEN + slot];
        in_q = a[2 * d.NGEN + d.NDES + slot];
        soc = io.e.soc[ee * W_DES + slot];
      }
    }
  }

  ANM_PHASE(1);
  // ---------------- device maps (lane = device) ---------------------------------------------
  double dev_p = 0.0, dev_q = 0.0, p_pot = 0.0;
  auto device_maps = [&](const auto& sd) {
    const Recip rbase = make_recip(base);   // (one reciprocal for the three divisions of a lane: see div_by)
    if (typ == DEV_LOAD) {
      const double p = fmin(fmax(div_by(in_p, rbase), sd[0]), sd[1]);
      dev_p = p;
      dev_q = p * sd[2];
    } else if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE || typ == DEV_STORAGE) {
      // ONE projection for generators and storage units (the lanes of a wavefront hold both: two calls are two
      // instruction streams, each with the other kind's lanes masked).  A generator's third and fourth slanted line
      // are switched off in its table (sense 0, anm_pack.hpp: pack_device), which makes the four-line projection
      // return exactly what the two-line one does.
      const bool des = typ == DEV_STORAGE;
      const double eff = sd[SD_EFF];   // (generators: 1)
      double pl = sd[SD_PMIN], pu = sd[SD_PMAX];
      if (des) {
        pl = fmax(pl, (soc - sd[SD_SOC_MAX]) / (dt * eff));
        pu = fmin(pu, eff * (soc - sd[SD_SOC_MIN]) / dt);
      } else {
        p_pot = fmin(fmax(div_by(in_pot, rbase), sd[SD_PMIN]), sd[SD_PMAX]);
        pu = fmin(pu, p_pot);
      }
      project_pq<4>(sd, div_by(in_p, rbase), div_by(in_q, rbase), pl, pu, dev_p, dev_q);
      if (des) {
        const double ns = (dev_p <= 0.0) ? (soc - dt * eff * dev_p) : (soc - dt * dev_p / eff);
        soc = fmin(fmax(ns, sd[SD_SOC_MIN]), sd[SD_SOC_MAX]);
      }
    }
  };
  if constexpr (ANM_RADIAL_PREFETCH != 0) device_maps(sdr);
  else device_maps(C + d.off_dev + l * SD_SIZE);
  sh[A_S0][t] = dev_p;
  sh[A_S1][t] = dev_q;
  ANM_GROUP_SYNC();
  double bus_p = 0.0, bus_q = 0.0;
  if (isbus)
    for (int k = RI(IF_BD_BEG); k < RI(IF_BD_END); ++k) {
      const int dd = lists[k];
      bus_p += sh[A_S0][gb + dd];
      bus_q += sh[A_S1][gb + dd];
    }
  ANM_GROUP_SYNC();

  ANM_PHASE(2);
  // ---------------- Newton-Raphson (lane = bus) ----------------------------------------------
  const double ybb_r = RD(DF_YBB_RE), ybb_i = RD(DF_YBB_IM), ybp_r = RD(DF_YBP_RE), ybp_i = RD(DF_YBP_IM);
  const double ypb_r = RD(DF_YPB_RE), ypb_i = RD(DF_YPB_IM);
  double vm = 1.0, cs = 1.0, sn = 0.0;
  if (mode == 0 && io.t.nr_start && isbus) {   // anm_model_bind_nr_start: the reference solver's v_guess instead of the flat start
    const double* x0 = io.t.nr_start + ee * (2 * (d.NB - 1));
    vm = x0[d.NB - 1 + l];
    const SinCos r0 = sincos_huge(x0[l]);
    sn = r0.s; cs = r0.c;
  }
  double vr = 1.0, vi = 0.0, ir = 0.0, ii = 0.0, vpr = 1.0, vpi = 0.0;
  int it = 0;
  bool g_bad = false, g_nan = false;   // the group's ||F||inf > tol / F has a NaN, as of its last evaluation
  bool active = true;
  const unsigned long long busm = __builtin_amdgcn_uicmp(isbus ? 1u : 0u, 0u, group::ICMP_NE);
  const unsigned long long gmask = (G >= 64) ? ~0ull : (((1ull << G) - 1ull) << ((t & 63) - l));
  const unsigned glo = unsigned(gmask), ghi = unsigned(gmask >> 32);
  const int pl = gb + (parent >= 0 ? parent : 0);
  // Same formulation as the thread-per-environment kernels (PFState in anm_device.hpp): everything
  // comes from W_ik = V_i conj(Y_ik V_k).  A lane owns W_bb, W_bp (its row, parent column) and W_pb
  // (parent row, its column); S_b = W_bb + W_bp + sum over children c of W_pb(c), gathered through LDS.
  double wpb_r = 0.0, wpb_i = 0.0;
  bool f_nan = false, f_bad = false;   // verdict on the final iterate: F has a NaN / ||F||inf > tol (or NaN)
  if constexpr (USE_T) {
    // lane l plays bus l + 1 here; the specialised loop may place the buses differently inside the group
    // (TT::T_LANE_BUS): move the inputs there, the final iterate back
    group::LaneView<TT> V;
    V.init();
    const int to4 = 4 * (gb + (V.lane_bus ? V.b - 1 : l));
    V.ybb_r = group::bperm(ybb_r, to4); V.ybb_i = group::bperm(ybb_i, to4);
    V.ybp_r = group::bperm(ybp_r, to4); V.ybp_i = group::bperm(ybp_i, to4);
    V.ypb_r = group::bperm(ypb_r, to4); V.ypb_i = group::bperm(ypb_i, to4);
    double gp = group::bperm(bus_p, to4), gq = group::bperm(bus_q, to4);
    if (!V.lane_bus) { V.ybb_r = V.ybb_i = V.ybp_r = V.ybp_i = V.ypb_r = V.ypb_i = 0.0; gp = gq = 0.0; }
    // flat start (solve_load_flow.py:36-39), or the bound initial guess moved to this loop's lane layout
    double gvm = group::bperm(vm, to4), gcs = group::bperm(cs, to4), gsn = group::bperm(sn, to4);
    if (!V.lane_bus) { gvm = 1.0; gcs = 1.0; gsn = 0.0; }
    int git = 0;
    unsigned tb, tn;
    // (trees without a DPP plan hand over through the LDS arrays the table-driven loop would use)
    static_assert(A_N >= 10, "group::newton_groups<.., LDSX> takes 640 doubles");
    // (per-environment parameters cost this kernel nine registers: two child slots per fetch round instead of three keep
    // it at three wavefronts per SIMD)
    group::newton_groups<TT, JT, 10, TT::T_DPP == 0, PG ? 2 : ANM_LDSX_FETCH, true, ANM_RADIAL_VPOLY != 0 && !PG,
                         (ANM_GROUP_MERGED_REGIONS != 0) && !(PG && TT::T_DPP != 0)>(V, env_ok && !skip, gvm, gcs, gsn, gp, gq, git, tb, tn,
                                                                            so.tol, so.max_iter, &sh[0][0]);
    int back_lane;   // the lane that plays bus l + 1
    if constexpr (TT::T_LP_NW > 0) back_lane = int(V.pk[TT::T_LP_NW - 1]);
    else back_lane = isbus ? TT::T_POS[l + 1] : l;
    const int back4 = 4 * (gb + back_lane);
    vm = group::bperm(gvm, back4); cs = group::bperm(gcs, back4); sn = group::bperm(gsn, back4);
    // iteration count and verdict are uniform over a group: every lane takes those of the lane playing bus 1
    // (all lanes execute the hand-over: an inactive source lane would read as 0)
    const int one4 = 4 * (gb + TT::T_POS[1]);
    it = __builtin_amdgcn_ds_bpermute(one4, git);
    f_nan = __builtin_amdgcn_ds_bpermute(one4, int(tn)) != 0;
    f_bad = __builtin_amdgcn_ds_bpermute(one4, int(tb)) != 0;
  } else {
  for (;;) {
    // ---- V; the parent's V comes through LDS
    vr = vm * cs;
    vi = vm * sn;
    pV[t] = double2{vr, vi};
    ANM_GROUP_SYNC();
    vpr = 1.0; vpi = 0.0;
    if (parent >= 0) { const double2 vp = pV[pl]; vpr = vp.x; vpi = vp.y; }
    // W_bb = conj(Y_bb) vm^2;  P = V_b conj(V_p): W_bp = conj(Y_bp) P, W_pb = conj(Y_pb) conj(P)
    const double m2 = vm * vm;
    const double wbb_r = ybb_r * m2, wbb_i = -(ybb_i * m2);
    const double pr = fma(vr, vpr, vi * vpi), pim = fma(vi, vpr, -(vr * vpi));
    const double wbp_r = fma(ybp_r, pr, ybp_i * pim), wbp_i = fma(ybp_r, pim, -(ybp_i * pr));
    wpb_r = fma(ypb_r, pr, -(ypb_i * pim));
    wpb_i = -fma(ypb_r, pim, ypb_i * pr);
    pU[t] = double2{wpb_r, wpb_i};
    ANM_GROUP_SYNC();
    double sr = wbb_r + wbp_r, si = wbb_i + wbp_i;
    for (int k = ch_beg; k < ch_end; ++k) {
      const double2 cw = pU[gb + lists[k]];
      sr += cw.x;
      si += cw.y;
    }
    // ---- mismatch; "||F||inf > tol" and "F has a NaN" over the group, on lane masks (as anm_group.hpp and anm_mesh.hpp:
    // two compares and a few scalar instructions instead of ten ds_bpermute butterflies): all the reference's loop
    // and flags need; a group that stopped keeps its verdict
    const double fr = sr - bus_p, fi = si - bus_q;
    const unsigned long long badm = __builtin_amdgcn_fcmp(isbus ? fmax(fabs(fr), fabs(fi)) : 0.0, so.tol, group::FCMP_UGT);
    const unsigned long long nanm = __builtin_amdgcn_fcmp(fr, fi, group::FCMP_UNO) & busm;
    const bool nb = ((unsigned(badm) & glo) | (unsigned(badm >> 32) & ghi)) != 0u;
    const bool nn = ((unsigned(nanm) & glo) | (unsigned(nanm >> 32) & ghi)) != 0u;
    if (it == 0 || active) { g_bad = nb; g_nan = nn; }
    active = g_bad && !g_nan && (it < so.max_iter);   // NaN > tol is false, like the reference
    ANM_GROUP_SYNC();
    if (!__any(active && env_ok && !skip)) break;
    // ---- Jacobian blocks (magnitude columns scaled by |V|): own diagonal, coupling with the parent
    Blk<JT> Dg = Blk<JT>{JT(-(si - wbb_i)), JT(sr + wbb_r), JT(sr - wbb_r), JT(si + wbb_i)};
    const Blk<JT> Jbp = Blk<JT>{JT(wbp_i), JT(wbp_r), JT(-wbp_r), JT(wbp_i)};
    const Blk<JT> Jpb = Blk<JT>{JT(wpb_i), JT(wpb_r), JT(-wpb_r), JT(wpb_i)};
    JT r0 = JT(fr), r1 = JT(fi);
    // ---- elimination, deepest level first
    for (int lev = d.max_depth; lev >= 0; --lev) {
      if (isbus && depth == lev) {
        for (int k = ch_beg; k < ch_end; ++k) {
          const int c = 3 * (gb + lists[k]);
          const double2 q0 = pS[c], q1 = pS[c + 1], q2 = pS[c + 2];
          Dg.a -= JT(q0.x); Dg.b -= JT(q0.y); Dg.c -= JT(q1.x); Dg.d -= JT(q1.y);
          r0 -= JT(q2.x); r1 -= JT(q2.y);
        }
        Dg = blk_inv(Dg);
        if (parent >= 0) {
          const Blk<JT> Lk = blk_mul(Jpb, Dg);
          const Blk<JT> Sc = blk_mul(Lk, Jbp);
          pS[3 * t] = double2{double(Sc.a), double(Sc.b)};
          pS[3 * t + 1] = double2{double(Sc.c), double(Sc.d)};
          pS[3 * t + 2] = double2{double(fm(Lk.a, r0, Lk.b * r1)), double(fm(Lk.c, r0, Lk.d * r1))};
        }
      }
      ANM_GROUP_SYNC();
    }
    // ---- back substitution, roots first (dx published in pU)
    JT d0 = JT(0), d1 = JT(0);
    for (int lev = 0; lev <= d.max_depth; ++lev) {
      if (isbus && depth == lev) {
        JT a0 = r0, a1 = r1;
        if (parent >= 0) {
          const double2 dp = pU[pl];
          const JT p0 = JT(dp.x), p1 = JT(dp.y);
          a0 = fm(-Jbp.b, p1, fm(-Jbp.a, p0, a0));
          a1 = fm(-Jbp.d, p1, fm(-Jbp.c, p0, a1));
        }
        d0 = fm(Dg.a, a0, Dg.b * a1);
        d1 = fm(Dg.c, a0, Dg.d * a1);
        pU[t] = double2{double(d0), double(d1)};
      }
      ANM_GROUP_SYNC();
    }
    // ---- update (group-uniform `active`); d1 is the relative magnitude step
    if (active && isbus) {
      // rotation of (cos, sin) by the angle step, see update_angles (anm_device.hpp)
      const double dth = double(d0);
      vm = fma(-double(d1), fabs(vm), vm);
      double sd_, cd_;
      if (fabs(dth) <= 0.78) {
        sincos_kernel(dth, 0, sd_, cd_);
      } else if (fabs(dth) < SINCOS_MEDIUM_MAX) {
        sincos_medium(dth, sd_, cd_);
      } else {
        const SinCos r = sincos_huge(dth);
        sd_ = r.s; cd_ = r.c;
      }
      const double c0 = cs, s0v = sn;
      cs = fma(c0, cd_, s0v * sd_);
      sn = fma(s0v, cd_, -(c0 * sd_));
    }
    it = active ? it + 1 : it;
  }
  f_nan = g_nan;
  f_bad = g_nan || g_bad;
  }
  // V of the final iterate; the parent's through LDS
  vr = vm * cs;
  vi = vm * sn;
  sh[A_VR][t] = vr; sh[A_VI][t] = vi;
  ANM_GROUP_SYNC();
  vpr = 1.0; vpi = 0.0;
  if (parent >= 0) { vpr = sh[A_VR][pl]; vpi = sh[A_VI][pl]; }
  ANM_GROUP_SYNC();
  // bus currents I = Y V of the final iterate (solve_load_flow.py:52-60), once
  {
    sh[A_UPR][t] = fma(ypb_r, vr, -(ypb_i * vi));
    sh[A_UPI][t] = fma(ypb_r, vi, ypb_i * vr);
    ANM_GROUP_SYNC();
    ir = fma(ybb_r, vr, -(ybb_i * vi)) + fma(ybp_r, vpr, -(ybp_i * vpi));
    ii = fma(ybb_r, vi, ybb_i * vr) + fma(ybp_r, vpi, ybp_i * vpr);
    for (int k = ch_beg; k < ch_end; ++k) {
      const int c = gb + lists[k];
      ir += sh[A_UPR][c];
      ii += sh[A_UPI][c];
    }
    ANM_GROUP_SYNC();
  }
  const bool converged = !f_nan && !f_bad;
  // ||F||inf of the final iterate (anm_model_bind_nr_diff; the reference's `diff`, solve_load_flow.py:84-120, 218-224):
  // the stop test above works on lane masks and never forms the norm, so it is recomputed here, once, from the final V
  // and I = Y V: F_b = V_b conj(I_b) - (p_b + j q_b); NaN if an entry is NaN (numpy's norm propagates it)
  double* const nr_diff = (mode == 0) ? io.t.nr_diff : ((mode == 1) ? io.e.nr_diff : nullptr);
  double fdiff = 0.0;
  if (nr_diff) {
    const double fa = fabs(fma(vr, ir, vi * ii) - bus_p), fb = fabs(fma(vi, ir, -(vr * ii)) - bus_q);
    double mx = isbus ? fmax(fa, fb) : 0.0;                            // (fmax drops a NaN: counted separately)
    double nn = (isbus && (fa != fa || fb != fb)) ? 1.0 : 0.0;
    for (int m = 1; m < G; m <<= 1) {
      mx = fmax(mx, __shfl_xor(mx, m, G));
      nn += __shfl_xor(nn, m, G);
    }
    fdiff = (nn > 0.0) ? NAN : mx;
  }

  ANM_PHASE(3);
  // ---------------- slack injection, branch flows, reward --------------------------------------
  // I_0 = Y_00 + sum over the buses attached to the slack of Y_0b V_b  (fixed butterfly order)
  double s0r = (isbus && parent < 0) ? (ypb_r * vr - ypb_i * vi) : 0.0;
  double s0i = (isbus && parent < 0) ? (ypb_r * vi + ypb_i * vr) : 0.0;
  for (int m = 1; m < G; m <<= 1) {
    s0r += __shfl_xor(s0r, m, G);
    s0i += __shfl_xor(s0i, m, G);
  }
  const double i0r = rd[SF_Y00_RE] + s0r, i0i = rd[SF_Y00_IM] + s0i;
  const double slack_p = (i0r != i0r) ? INFINITY : i0r;
  const double slack_q = (i0i != i0i) ? INFINITY : -i0i;
  if (typ == DEV_SLACK) { dev_p = slack_p; dev_q = slack_q; }

  double br_pf = 0, br_qf = 0, br_s = 0, br_ifr = 0, br_ifi = 0, pen = 0.0;
  if (isbus) {
    const bool from = RI(IF_BR_FROM) != 0;
    const double vfr = from ? vr : vpr, vfi = from ? vi : vpi, vtr = from ? vpr : vr, vti = from ? vpi : vi;
    const double c0 = RD(DF_BRC + 0), c1 = RD(DF_BRC + 1), c2 = RD(DF_BRC + 2), c3 = RD(DF_BRC + 3);
    const double c4 = RD(DF_BRC + 4), c5 = RD(DF_BRC + 5), c6 = RD(DF_BRC + 6), c7 = RD(DF_BRC + 7);
    br_ifr = c0 * vfr - c1 * vfi + c2 * vtr - c3 * vti;
    br_ifi = c0 * vfi + c1 * vfr + c2 * vti + c3 * vtr;
    const double itr = c4 * vtr - c5 * vti + c6 * vfr - c7 * vfi;
    const double iti = c4 * vti + c5 * vtr + c6 * vfi + c7 * vfr;
    br_pf = vfr * br_ifr + vfi * br_ifi;
    br_qf = vfi * br_ifr - vfr * br_ifi;
    const double pt = vtr * itr + vti * iti, qt = vti * itr - vtr * iti;
    const double sf2 = br_pf * br_pf + br_qf * br_qf, st2 = pt * pt + qt * qt;   // one root: see flows_and_reward
    const double sgn = (br_pf > 0.0) ? 1.0 : ((br_pf < 0.0) ? -1.0 : ((br_pf == 0.0) ? 0.0 : NAN));
    const double smax = (sf2 != sf2 || st2 != st2) ? NAN : sqrt(fmax(sf2, st2));
    br_s = sgn * smax;
    const double over = fabs(br_s) - RD(DF_BRC + 8);
    pen += (over != over) ? NAN : fmax(0.0, over);
    const double vmag = fabs(vm);
    const double hi = vmag - RD(DF_VMAX), lo = RD(DF_VMIN) - vmag;
    pen += (vmag != vmag) ? NAN : (fmax(0.0, hi) + fmax(0.0, lo));
  }
  if (l == 0) pen += fmax(0.0, 1.0 - rd[SF_SLACK_VMAX]) + fmax(0.0, rd[SF_SLACK_VMIN] - 1.0);
  double el = 0.0;
  if (typ != DEV_NONE && typ != DEV_STORAGE) el += dev_p;
  if (typ == DEV_RENEWABLE) {
    const double curt = p_pot - dev_p;
    el += (curt != curt) ? NAN : fmax(0.0, curt);
  }
  for (int m = 1; m < G; m <<= 1) {
    pen += __shfl_xor(pen, m, G);
    el += __shfl_xor(el, m, G);
  }
  const double e_loss = el * dt, penalty = pen * (dt * rd[SF_LAMB]);
  const double reward = -(e_loss + penalty);

  if (!env_ok) return;

  ANM_PHASE(4);
  // ---------------- outputs ------------------------------------------------------------------
  double* full = (mode == 0) ? io.t.full : io.e.full;
  auto write_full = [&]() {
    if (!full) return;
    double* f = full + e * W_FS;
    if (isbus) {
      const int b = l + 1;
      f[d.f_bus_p + b] = bus_p; f[d.f_bus_q + b] = bus_q;
      f[d.f_bus_vm + b] = dump_abs(vr, vi); f[d.f_bus_va + b] = dump_arg(vi, vr);
      f[d.f_bus_im + b] = dump_abs(ir, ii); f[d.f_bus_ia + b] = dump_arg(ii, ir);
      const int bi = RI(IF_BR_INDEX);
      f[d.f_br_p + bi] = br_pf; f[d.f_br_q + bi] = br_qf; f[d.f_br_s + bi] = br_s;
      f[d.f_br_im + bi] = dump_signed_abs(br_ifr, dump_abs(br_ifr, br_ifi));
      f[d.f_br_ia + bi] = dump_arg(br_ifi, br_ifr);
    }
    if (l == 0) {
      f[d.f_bus_p] = slack_p; f[d.f_bus_q] = slack_q; f[d.f_bus_vm] = 1.0; f[d.f_bus_va] = 0.0;
      f[d.f_bus_im] = dump_abs(i0r, i0i); f[d.f_bus_ia] = dump_arg(i0i, i0r);
    }
    if (typ != DEV_NONE) { f[d.f_dev_p + l] = dev_p; f[d.f_dev_q + l] = dev_q; }
    if (typ == DEV_STORAGE) f[d.f_des_soc + slot] = soc;
    if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) f[d.f_gen_pmax + slot] = p_pot;
  };

  // (one call site for the dump: its |z| / arg z code exists once, not once per mode)
  bool dump = false;
  do {
  if (mode == 0) {
    if (typ == DEV_STORAGE) io.t.soc[e * W_DES + slot] = soc;
    if (l == 0) {
      io.t.reward[e] = reward; io.t.e_loss[e] = e_loss; io.t.penalty[e] = penalty;
      io.t.converged[e] = converged ? 1 : 0;
      if (io.t.nr_iters) io.t.nr_iters[e] = it;
      if (nr_diff) nr_diff[e] = fdiff;
    }
    dump = true;
    break;
  }

  double* state = io.e.state + e * W_ST;
  // the observation: clip(state, Box) next to the state row, or (a list is set: anm_env.py:497-521, 562-592) n_obs entries
  // gathered from this environment's electrical state
  const bool list = mode == 2 && io.e.n_obs > 0;
  const int ON = list ? io.e.n_obs : W_ST;                                  // entries of an observation row
  const int OW = list ? (io.v.w_obs > 0 ? io.v.w_obs : io.e.n_obs) : W_ST;   // its stride (a view pads the rows)
  double* obs = io.e.obs + e * OW;
  cptr_t lo = C + d.off_obs_lo, hi = C + d.off_obs_hi;
  auto put = [&](int k, double v) {
    state[k] = v;
    if (!list) obs[k] = fmin(fmax(v, lo[k]), hi[k]);
  };
  // (called by all lanes of an environment together -- every condition around it is uniform over the lane group -- so the
  // row is written and read back in program order of one control path: LDS operations of a wavefront complete in order)
  auto list_obs = [&](bool zero) {
    if (!list) return;
    if (zero) {   // terminal / absorbing: the observation is 0 (anm_env.py:365-367, 442-446)
      for (int k = l; k < ON; k += G) obs[k] = 0.0;
      return;
    }
    double* row = sh_obs_rows + (t / G) * (d.FS + KMAX);
    const unsigned need = io.e.obs_need;
    auto want = [&](unsigned c) { return ((need >> c) & 1u) != 0u; };
    if (isbus) {
      const int b = l + 1, bi = RI(IF_BR_INDEX);
      if (want(FC_BUS_P)) row[d.f_bus_p + b] = bus_p;
      if (want(FC_BUS_Q)) row[d.f_bus_q + b] = bus_q;
      if (want(FC_BUS_VM)) row[d.f_bus_vm + b] = dump_abs(vr, vi);
      if (want(FC_BUS_VA)) row[d.f_bus_va + b] = dump_arg(vi, vr);
      if (want(FC_BUS_IM)) row[d.f_bus_im + b] = dump_abs(ir, ii);
      if (want(FC_BUS_IA)) row[d.f_bus_ia + b] = dump_arg(ii, ir);
      if (want(FC_BR_P)) row[d.f_br_p + bi] = br_pf;
      if (want(FC_BR_Q)) row[d.f_br_q + bi] = br_qf;
      if (want(FC_BR_S)) row[d.f_br_s + bi] = br_s;
      if (want(FC_BR_IM)) row[d.f_br_im + bi] = dump_signed_abs(br_ifr, dump_abs(br_ifr, br_ifi));
      if (want(FC_BR_IA)) row[d.f_br_ia + bi] = dump_arg(br_ifi, br_ifr);
    }
    if (l == 0) {
      row[d.f_bus_p] = slack_p; row[d.f_bus_q] = slack_q; row[d.f_bus_vm] = 1.0; row[d.f_bus_va] = 0.0;
      if (want(FC_BUS_IM)) row[d.f_bus_im] = dump_abs(i0r, i0i);
      if (want(FC_BUS_IA)) row[d.f_bus_ia] = dump_arg(i0i, i0r);
    }
    if (typ != DEV_NONE) { row[d.f_dev_p + l] = dev_p; row[d.f_dev_q + l] = dev_q; }
    if (typ == DEV_STORAGE) row[d.f_des_soc + slot] = soc;
    if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) row[d.f_gen_pmax + slot] = p_pot;
    // the aux variables sit behind the electrical state
    if (resetting || io.e.exo == nullptr) { if (l == 0) row[d.FS] = double(aux); }
    else for (int k = l; k < K; k += G) row[d.FS + k] = io.e.aux_next[e * W_AUX + k];
    ANM_GROUP_SYNC();
    for (int k = l; k < ON; k += G) {
      const double v = row[io.e.obs_index[k]] * io.e.obs_scale[k];
      obs[k] = fmin(fmax(v, io.e.obs_lo[k]), io.e.obs_hi[k]);
    }
  };
  if (skip) {
    if (mode == 2) {  // absorbing terminal state
      if (list) list_obs(true);
      else for (int k = l; k < S; k += G) obs[k] = 0.0;
      if (l == 0) { io.e.reward[e] = 0.0; if (io.e.nr_iters) io.e.nr_iters[e] = 0; }
    }
    break;
  }
  if (l == 0 && io.e.nr_iters) io.e.nr_iters[e] = it;
  if (resetting) {
    if (typ == DEV_STORAGE) {
      soc = soc_req / base;  // simulator.py:284-288
      io.e.soc[e * W_DES + slot] = soc;
    }
    if (mode == 2 && !converged) {
      // a redraw whose first power flow does not converge looks like the absorbing terminal state
      // until the next call draws again
      for (int k = l; k < S; k += G) { state[k] = 0.0; if (!list) obs[k] = 0.0; }
    } else {
      if (typ != DEV_NONE) { put(l, dev_p * base); put(d.ND + l, dev_q * base); }
      if (typ == DEV_STORAGE) put(2 * d.ND + slot, soc * base);
      if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) put(2 * d.ND + d.NDES + slot, p_pot * base);
    }
    if (mode == 1) {
      if (sampled) {
        if (l == 0) { put(d.SDIM, double(aux)); io.e.reset_count[e] += 1; }
      } else {
        const double* s0 = io.e.init_state + e * W_ST;
        for (int k = l; k < K; k += G) put(d.SDIM + k, s0[d.SDIM + k]);
      }
      if (l == 0) { io.e.converged[e] = converged ? 1 : 0; io.e.terminated[e] = 0; if (io.e.timestep) io.e.timestep[e] = 0; }
      if (l == 0 && nr_diff) nr_diff[e] = fdiff;
    } else {
      if (l == 0) {
        if (converged) put(d.SDIM, double(aux));
        io.e.reset_count[e] += 1;
        io.e.terminated[e] = converged ? 0 : 1;
        if (io.e.timestep) io.e.timestep[e] = 0;
        io.e.reward[e] = 0.0; io.e.e_loss[e] = 0.0; io.e.penalty[e] = 0.0;
      }
      list_obs(!converged);
    }
    dump = true;
    break;
  }
  // regular step
  if (typ == DEV_STORAGE) io.e.soc[e * W_DES + slot] = soc;
  const bool term = !converged;
  if (!term) {
    if (typ != DEV_NONE) { put(l, dev_p * base); put(d.ND + l, dev_q * base); }
    if (typ == DEV_STORAGE) put(2 * d.ND + slot, soc * base);
    if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) put(2 * d.ND + d.NDES + slot, p_pot * base);
    if (io.e.exo == nullptr) {
      if (l == 0) put(d.SDIM, double(aux));
    } else {
      for (int k = l; k < K; k += G) put(d.SDIM + k, io.e.aux_next[e * W_AUX + k]);
    }
  } else {
    for (int k = l; k < S; k += G) { state[k] = 0.0; if (!list) obs[k] = 0.0; }
  }
  list_obs(term);
  if (l == 0) {
    const double c1 = rd[SF_C1], c2 = rd[SF_C2];
    io.e.terminated[e] = term ? 1 : 0;
    if (!term) {
      const double sg2 = (e_loss > 0.0) ? 1.0 : ((e_loss < 0.0) ? -1.0 : 0.0);
      const double elc = sg2 * fmin(fabs(e_loss), c1);
      const double pn = fmin(fmax(penalty, 0.0), c2);
      io.e.e_loss[e] = elc; io.e.penalty[e] = pn; io.e.reward[e] = -(elc + pn);
    } else {
      io.e.reward[e] = rd[SF_RTERM]; io.e.e_loss[e] = c1; io.e.penalty[e] = c2;
    }
    if (io.e.timestep) io.e.timestep[e] += 1;
  }
  dump = true;
  } while (false);
  if (dump) write_full();
  ANM_PHASE(5);
}
#endif  // __HIPCC__

}  // namespace radial
}  // namespace anm
