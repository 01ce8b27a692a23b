// anm_env_ops.hpp -- the three per-environment operations behind the C ABI, written once for
// device (gfx950 kernels in anm_capi.hip) and host (tests/hostsim test double).
//
//   op_transition  Simulator.transition                       simulator.py:464-537
//   op_reset       Simulator.reset + ANMEnv.reset tail        simulator.py:225-293, anm_env.py:292-311
//   op_step        ANMEnv.step (+ fused next_vars / autoreset) anm_env.py:333-453, anm6_easy.py:25-65
#pragma once

#include "anm_device.hpp"
#include "anm_group.hpp"

// In-wave continuation of the diverging solves: the lane-group loop WITHOUT its short sincos path (group::newton_groups:
// REDUCE_ALWAYS), on the idea that a launch lasts as long as its slowest wavefront, whose trips reduce anyway -- measured and
// left off: headline kernel 72.6 -> 76.3 us, same box (profiles/r06_r_s_worst_wavefront.txt): the slowest wavefront, too, takes
// the short path on most of its trips.
#ifndef ANM_INWAVE_REDUCE_ALWAYS
#define ANM_INWAVE_REDUCE_ALWAYS 0
#endif

namespace anm {

// Parameter classes (anm_model_set_classes): networks with the topology of the model and other numbers.  The
// constants of class k start k * stride doubles behind those of class 0; env_class[e] (device, int32) names
// the class of environment e and is constant over every aligned block of 64 environments, so the constants
// of a wavefront stay ONE wave-uniform buffer read through scalar loads -- parameter classes cost nothing.
struct ClassSel {
  const int32_t* env_class;  // never null: without classes it points at one zero and `per_env` is 0
  int stride;
  int per_env;               // 1: index env_class by the environment, 0: always entry 0
  int per_group;             // lane-group families: 1 = the class may change from one environment to the next (every
                             // lane group reads its own constants: vector loads), 0 = one class per aligned block of 64
};
#if defined(__HIPCC__)
// (branch-free on purpose: a conditional here keeps the selector alive in scalar registers through the whole
// kernel, and the step kernel has none to spare -- measured: 80 bytes of scratch)
__device__ __forceinline__ cptr_t class_constants(cptr_t C, const ClassSel& cs, int64_t first_env) {
  const int k = __builtin_amdgcn_readfirstlane(cs.env_class[first_env * cs.per_env]);
  return C + int64_t(k) * cs.stride;
}
#endif

// A launch may serve a SUB-batch of a larger batch (anm_model_bind_view): slot s of the launch is environment index[s] of
// the batch arrays, whose rows are padded to common widths -- how one batch holds environments over networks of DIFFERENT
// topologies, one launch per topology, each in its own kernel family
struct View {
  const int32_t* index;   // null: slot = environment
  int w_load, w_gen, w_set, w_des, w_action, w_state, w_exo, w_aux, w_full;   // row strides (0: the network's own widths)
  int w_obs;              // ... of the rows of a list-form observation (0: n_obs)
};

struct SolverOpts {
  double tol;
  int max_iter;
  int handoff;  // >= 0: Newton iterations in thread mode before a running solve moves to a lane group; < 0: never
};

template <class T>
struct Dims {
  static constexpr int NEXO = T::NLOAD + T::NGEN;
  static constexpr int ADIM = 2 * (T::NGEN + T::NDES);
  static constexpr int SBASE = T::SDIM;  // 2 ND + NDES + NGEN
};

// state vector (anm_env.py:139-147): [dev_p MW, dev_q MVAr, des_soc MWh, gen_p_max MW, aux(K)]
// `out`: where the state / observation entries go -- anything with put_st(k, v) / put_ob(k, v) (register
// arrays: StepOut; rows in memory: RowPtrs; an LDS row: op_step_general).  Setters, not references: a
// reference that may point at LDS or at a local is a generic pointer, and a local whose address becomes a
// generic pointer lives in scratch.
template <class T, class Out>
ANM_HD void write_state_obs(cptr_t C, const EnvWork<T>& w, Out& out) {
  typedef Layout<T> L;
  const double base = C[L::SCALARS + SC_BASE];
  auto put = [&](int k, double v) {
    out.put_st(k, v);
    // np.clip(obs, low, high)
    out.put_ob(k, vmin(vmax(v, C[L::OBS_LO + k]), C[L::OBS_HI + k]));
  };
  static_for<0, T::ND>([&](auto Di) {
    constexpr int d = Di;
    put(d, w.dev_p[d] * base);
    put(T::ND + d, w.dev_q[d] * base);
  });
  static_for<0, T::NDES>([&](auto E) { put(2 * T::ND + E, w.soc[E] * base); });
  static_for<0, T::NGEN>([&](auto G) { put(2 * T::ND + T::NDES + G, w.p_pot[G] * base); });
}

// state / obs rows in memory
struct RowPtrs {
  double* state;
  double* obs;
  ANM_HD void put_st(int k, double v) { state[k] = v; }
  ANM_HD void put_ob(int k, double v) { obs[k] = v; }
};

struct TransitionIO {
  const double* p_load;
  const double* p_pot;
  const double* p_set;
  const double* q_set;
  double* soc;
  double* full;
  double* reward;
  double* e_loss;
  double* penalty;
  uint8_t* converged;
  int32_t* nr_iters;
  double* nr_diff;   // [E] or null (anm_model_bind_nr_diff): ||F||inf of the final iterate
  const double* nr_start;  // [E, 2 (NB - 1)] or null (anm_model_bind_nr_start): initial guess (angles, magnitudes)
};

// (slot: position in the launch; with a view -- anm_model_bind_view -- the environment is index[slot] and the rows of the
// batch arrays have the view's strides)
template <class T, class JT, bool START = false>
ANM_HD void op_transition(cptr_t C, const TransitionIO& io, SolverOpts so, int64_t slot, const View& v = View{}, bool valid = true,
                          double* lds = nullptr) {
  // valid / lds (GPU): `slot` is clamped for the lanes beyond the batch (they compute along and store nothing), and a tree
  // topology hands the solves still running after so.handoff iterations over to lane groups inside the wavefront, like the
  // step kernels (group::continue_in_groups, collective: hence no early return) -- 65 536 ANM6 transitions 242 -> ~100 us:
  // the launch waits for its diverging solves, and a lane group's trip is a quarter of a thread's
  const int64_t e = v.index ? int64_t(v.index[slot]) : slot;
  const int WL = v.w_load > 0 ? v.w_load : T::NLOAD, WG = v.w_gen > 0 ? v.w_gen : T::NGEN, WS = v.w_set > 0 ? v.w_set : T::NSET;
  const int WD = v.w_des > 0 ? v.w_des : T::NDES, WF = v.w_full > 0 ? v.w_full : FullState<T>::SIZE;
  EnvWork<T> w;
  double P_load[T::NLOAD > 0 ? T::NLOAD : 1], P_pot[T::NGEN > 0 ? T::NGEN : 1];
  double P_set[T::NSET > 0 ? T::NSET : 1], Q_set[T::NSET > 0 ? T::NSET : 1];
  static_for<0, T::NLOAD>([&](auto I) { P_load[I] = io.p_load[e * WL + I]; });
  static_for<0, T::NGEN>([&](auto I) { P_pot[I] = io.p_pot[e * WG + I]; });
  static_for<0, T::NSET>([&](auto I) {
    P_set[I] = io.p_set[e * WS + I];
    Q_set[I] = io.q_set[e * WS + I];
  });
  static_for<0, T::NDES>([&](auto I) { w.soc[I] = io.soc[e * WD + I]; });
  if (START && io.nr_start) {
    // the reference's solver from a given initial guess (v_guess of _newton_raphson_sparse, solve_load_flow.py:176)
    PFState<T> st;
    transition_begin<T, JT>(C, C, w, st, P_load, P_pot, P_set, Q_set, so.tol, so.max_iter, -1);
    const double* x0 = io.nr_start + e * (2 * (T::NB - 1));
    static_for<1, T::NB>([&](auto I) {
      constexpr int i = I;
      w.vm[i] = x0[T::NB - 1 + (i - 1)];
      const SinCos r = sincos_huge(x0[i - 1]);   // (the library's sincos: any angle)
      st.sn[i] = r.s;
      st.cs[i] = r.c;
    });
    pf_iterate<T, JT>(C, w, st, so.tol, so.max_iter, so.max_iter);
    transition_end<T>(C, w, st, so.tol);
  } else {
    PFState<T> st;
    int cap = so.max_iter;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (T::TREE != 0)
      if (lds && so.handoff >= 0 && so.handoff < so.max_iter) cap = so.handoff;
#endif
    transition_begin<T, JT>(C, C, w, st, P_load, P_pot, P_set, Q_set, so.tol, so.max_iter, cap);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (T::TREE != 0)
      if (cap < so.max_iter && ANM_WAVE_ANY(st.active && valid))
        group::continue_in_groups<T, JT, true, ANM_INWAVE_REDUCE_ALWAYS != 0>(C, w, st, st.active && valid, so.tol, so.max_iter, lds);
#endif
    transition_end<T>(C, w, st, so.tol);
  }
  if (!valid) return;
  static_for<0, T::NDES>([&](auto I) { io.soc[e * WD + I] = w.soc[I]; });
  io.reward[e] = w.reward;
  io.e_loss[e] = w.e_loss;
  io.penalty[e] = w.penalty;
  io.converged[e] = w.converged ? 1 : 0;
  if (io.nr_iters) io.nr_iters[e] = w.n_iter;
  if (io.nr_diff) io.nr_diff[e] = w.diff;
  if (io.full) write_full_state<T>(w, io.full + e * WF);
}

struct EnvIO {
  int K;                    // number of aux variables
  const double* action;     // [E, ADIM]
  const double* exo;        // [E, NEXO] or null (series mode)
  const double* aux_next;   // [E, K] or null (series mode)
  const double* init_state; // [E, SBASE + K] (reset)
  const uint8_t* mask;      // [E] or null (reset)
  const double* series;     // [NEXO, period] (series mode)
  int period;
  double* soc;              // [E, NDES]
  double* state;            // [E, SBASE + K]
  uint8_t* terminated;      // [E]
  int32_t* timestep;        // [E] or null
  double* obs;              // [E, SBASE + K]
  double* reward;
  double* e_loss;
  double* penalty;
  uint8_t* converged;       // [E] (reset)
  int32_t* nr_iters;        // [E] or null
  double* full;             // [E, FS] or null
  int autoreset;
  uint64_t rng_seed;
  uint64_t env_offset;      // global index of environment 0 of this batch (RNG key)
  int32_t* reset_count;     // [E] (autoreset)
  int32_t* aux_index;       // [E] series mode: compact copy of the time index (optional)
  uint8_t* state_same;      // [E] or null (anm_model_bind_state_same): 1 = the state row equals the obs row and was not written
  // list-form observation produced inside the step kernel (anm_model_set_obs): obs[e, k] =
  // clip(src(e, obs_index[k]) * obs_scale[k], obs_lo[k], obs_hi[k]); n_obs == 0: obs = clip(state)
  int n_obs;
  unsigned obs_magic;       // ceil(2^32 / n_obs): idx / n_obs by multiply-high
  unsigned state_magic;     // ceil(2^32 / (SDIM + K))
  unsigned obs_need;        // FullClass bits the list reads
  int row_stride;           // doubles per LDS row of the electrical state (odd)
  int state_row_off;        // op_step_general: where the state rows start in its LDS buffer (doubles)
  int aux_off;              // where the aux variables sit in such a row
  short cls_off[FC_COUNT];  // where each class sits in such a row (compact: only the classes the list reads)
  const int32_t* obs_index; // indices into such a row
  const double* obs_scale;
  const double* obs_lo;
  const double* obs_hi;
  double* ws;               // two-phase step: workspace (counters + straggler records) or null
  int64_t ws_cap;           // number of straggler records the workspace can hold
  int iter_cap;             // two-phase step: Newton iterations done by the first launch
  int mid_cap;              // ... by the first straggler launch (0: it runs every record to the end, one level)
  int32_t* ws_list2;        // records the first straggler launch did not finish (count: counter 2 of the header)
  double* nr_diff;          // [E] or null (anm_model_bind_nr_diff; reset only): ||F||inf of the final iterate
  int aux_stride;           // row stride of aux_next (0: K; a batch view pads the rows)
};

// Split an init_state row (anm_env.py / simulator.py:248-268) into transition inputs.
// (S0: anything indexable -- a row in memory or a register array with constant indices)
template <class T, class S0>
ANM_HD void inputs_from_init_state(cptr_t C, const S0& s0, EnvWork<T>& w,
                                   double (&P_load)[T::NLOAD > 0 ? T::NLOAD : 1],
                                   double (&P_pot)[T::NGEN > 0 ? T::NGEN : 1],
                                   double (&P_set)[T::NSET > 0 ? T::NSET : 1],
                                   double (&Q_set)[T::NSET > 0 ? T::NSET : 1]) {
  typedef Layout<T> L;
  static_for<0, T::ND>([&](auto Di) {
    constexpr int d = Di;
    constexpr int typ = T::DEV_TYPE[d];
    if constexpr (typ == DEV_LOAD) {
      P_load[T::DEV_SLOT[d]] = s0[d];
    } else if constexpr (typ != DEV_SLACK) {
      P_set[T::DEV_SET[d]] = s0[d];
      Q_set[T::DEV_SET[d]] = s0[T::ND + d];
      if constexpr (typ == DEV_STORAGE) {
        // SoC pre-set so that the requested injection is feasible (simulator.py:273-278)
        cptr_t sd = C + L::SETDEV + SD_SIZE * T::DEV_SET[d];
        w.soc[T::DEV_SLOT[d]] = (s0[d] <= 0.0) ? sd[SD_SOC_MIN] : sd[SD_SOC_MAX];
      } else {
        P_pot[T::DEV_SLOT[d]] = s0[2 * T::ND + T::NDES + T::DEV_SLOT[d]];
      }
    }
  });
}

// Tail of Simulator.reset / ANMEnv.reset: SoC overwritten with the requested one, state & obs built.
template <class T, int KCAP = Layout<T>::KMAX, class S0 = const double*, class Out = RowPtrs>
ANM_HD void finish_reset(cptr_t C, EnvWork<T>& w, const S0& s0, int K, double* soc, Out& out) {
  typedef Layout<T> L;
  const double base = C[L::SCALARS + SC_BASE];
  static_for<0, T::NDES>([&](auto E) {
    w.soc[E] = s0[2 * T::ND + E] / base;  // simulator.py:284-288
    soc[E] = w.soc[E];
  });
  write_state_obs<T>(C, w, out);
  // constant indices + predicate (not a runtime-indexed loop): callers keep these rows in registers
  static_for<0, KCAP>([&](auto Kc) {
    constexpr int k = Kc;
    if (k < K) {
      const double a = s0[T::SDIM + k];
      out.put_st(T::SDIM + k, a);
      out.put_ob(T::SDIM + k, fmin(fmax(a, C[L::OBS_LO + T::SDIM + k]), C[L::OBS_HI + T::SDIM + k]));
    }
  });
}

// ANM6Easy.init_state (anm6_easy.py:25-52) generalised to any series-mode task, drawn on the device:
// time index uniform in [0, period), loads and generator potentials from the series at that index,
// generator Q uniform in its p.u. range (stored in the MVAr slot, sic), storage SoC uniform in its
// p.u. range (stored in the MWh slot, sic).  Counter-based: (seed, global env index, epoch).
// SER: where the exogenous series are read from -- io.series (global memory; the default) or a copy in LDS
template <class T, class SER = const double*>
ANM_HD int sample_series_init_state(cptr_t C, const EnvIO& io, int64_t e, uint32_t epoch, double (&s0)[T::SDIM + 1], SER ser = nullptr) {
  if constexpr (std::is_same<SER, const double*>::value) { if (!ser) ser = io.series; }
  typedef Layout<T> L;
  uint32_t r[4];
  Philox::generate(io.rng_seed, io.env_offset + uint64_t(e), epoch, 0u, r);
  const int aux = int((uint64_t(r[0]) * uint64_t(io.period)) >> 32);
  static_for<0, T::SDIM>([&](auto I) { s0[I] = 0.0; });
  s0[T::SDIM] = double(aux);
  static_for<0, T::ND>([&](auto Di) {
    constexpr int d = Di;
    constexpr int typ = T::DEV_TYPE[d];
    if constexpr (typ == DEV_LOAD) {
      s0[d] = ser[T::DEV_SLOT[d] * io.period + aux];
    } else if constexpr (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) {
      constexpr int g = T::DEV_SLOT[d];
      constexpr int u = g;  // uniform index
      cptr_t sd = C + L::SETDEV + SD_SIZE * T::DEV_SET[d];
      uint32_t q[4];
      Philox::generate(io.rng_seed, io.env_offset + uint64_t(e), epoch, 1u + u / 2, q);
      const double uu = Philox::u01(q[2 * (u % 2)], q[2 * (u % 2) + 1]);
      const double pm = ser[(T::NLOAD + g) * io.period + aux];
      s0[d] = pm;
      s0[2 * T::ND + T::NDES + g] = pm;
      s0[T::ND + d] = sd[SD_QMIN] + (sd[SD_QMAX] - sd[SD_QMIN]) * uu;
    } else if constexpr (typ == DEV_STORAGE) {
      constexpr int u = T::NGEN + T::DEV_SLOT[d];
      cptr_t sd = C + L::SETDEV + SD_SIZE * T::DEV_SET[d];
      uint32_t q[4];
      Philox::generate(io.rng_seed, io.env_offset + uint64_t(e), epoch, 1u + u / 2, q);
      const double uu = Philox::u01(q[2 * (u % 2)], q[2 * (u % 2) + 1]);
      s0[2 * T::ND + T::DEV_SLOT[d]] = sd[SD_SOC_MIN] + (sd[SD_SOC_MAX] - sd[SD_SOC_MIN]) * uu;
    }
  });
  return aux;
}

template <class T, class JT, class S0>
ANM_HD void reset_from(cptr_t C, const EnvIO& io, SolverOpts so, int64_t e, const S0& s0, const View& v = View{}, bool act = true,
                       double* lds = nullptr) {   // act / lds: see op_transition (lanes with !act compute along and store nothing)
  const int S = v.w_state > 0 ? v.w_state : T::SDIM + io.K;        // row stride of state / obs
  const int WD = v.w_des > 0 ? v.w_des : T::NDES, WF = v.w_full > 0 ? v.w_full : FullState<T>::SIZE;
  EnvWork<T> w;
  double P_load[T::NLOAD > 0 ? T::NLOAD : 1], P_pot[T::NGEN > 0 ? T::NGEN : 1];
  double P_set[T::NSET > 0 ? T::NSET : 1], Q_set[T::NSET > 0 ? T::NSET : 1];
  inputs_from_init_state<T>(C, s0, w, P_load, P_pot, P_set, Q_set);
  {
    PFState<T> st;
    int cap = so.max_iter;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (T::TREE != 0)
      if (lds && so.handoff >= 0 && so.handoff < so.max_iter) cap = so.handoff;
#endif
    // (a lane that stores nothing -- masked out of a partial reset, or beyond the batch -- evaluates F once and iterates
    // no further: a redraw round of ANMEnv.reset must not wait for 63 solves nobody asked for)
    transition_begin<T, JT>(C, C, w, st, P_load, P_pot, P_set, Q_set, so.tol, so.max_iter, act ? cap : 0);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (T::TREE != 0)
      if (cap < so.max_iter && ANM_WAVE_ANY(st.active && act))
        group::continue_in_groups<T, JT, true, ANM_INWAVE_REDUCE_ALWAYS != 0>(C, w, st, st.active && act, so.tol, so.max_iter, lds);
#endif
    transition_end<T>(C, w, st, so.tol);
  }
  if (!act) return;
  RowPtrs rows{io.state + e * S, io.obs + e * S};
  finish_reset<T, Layout<T>::KMAX, S0>(C, w, s0, io.K, io.soc + e * WD, rows);
  io.converged[e] = w.converged ? 1 : 0;
  io.terminated[e] = 0;
  if (io.timestep) io.timestep[e] = 0;
  if (io.nr_iters) io.nr_iters[e] = w.n_iter;
  if (io.nr_diff) io.nr_diff[e] = w.diff;
  if (io.aux_index && io.K == 1) io.aux_index[e] = int32_t(s0[T::SDIM]);
  if (io.full) write_full_state<T>(w, io.full + e * WF);
}

template <class T, class JT>
ANM_HD void op_reset(cptr_t C, const EnvIO& io, SolverOpts so, int64_t slot, const View& v = View{}, bool valid = true, double* lds = nullptr) {
  const int64_t e = v.index ? int64_t(v.index[slot]) : slot;
  const bool act = valid && !(io.mask && !io.mask[e]);
  if (!ANM_WAVE_ANY(act)) return;   // (wavefront-uniform: the hand-over to lane groups inside reset_from is collective)
  if (io.init_state) {  // the row given by the caller
    reset_from<T, JT>(C, io, so, e, io.init_state + e * (v.w_state > 0 ? v.w_state : T::SDIM + io.K), v, act, lds);
  } else {              // device sampler (series mode, K = 1): same draws as the autoreset path, kept in registers
    double s0_drawn[T::SDIM + 1];
    sample_series_init_state<T>(C, io, e, uint32_t(io.reset_count[e]), s0_drawn);
    if (act) io.reset_count[e] += 1;
    reset_from<T, JT>(C, io, so, e, s0_drawn, v, act, lds);
  }
}

// ---------------------------------------------------------------------------------------------
// ANMEnv.step for ONE environment, as a pure function of its inputs (StepIn -> StepOut); the two
// I/O layers below (plain per-environment loads/stores, and wave-cooperative coalesced rows on the
// GPU) share it.
// ---------------------------------------------------------------------------------------------
template <class T>
struct StepIn {
  bool was_term;
  double action[Dims<T>::ADIM > 0 ? Dims<T>::ADIM : 1];
  double exo[Dims<T>::NEXO > 0 ? Dims<T>::NEXO : 1];   // generic mode only
  double soc[T::NDES > 0 ? T::NDES : 1];
  double aux_prev;                                       // series mode: time index before the step
  int aux_next = -1;                                     // ... or, >= 0, the next one, already formed (integer callers)
  int reset_count;
};

// what a step decides besides the state / observation rows
template <class T>
struct StepFlags {
  bool write_state, write_obs;     // absorbing terminal state: only the (zero) observation is written
  double reward, e_loss, penalty;
  bool write_costs;                // e_loss / penalty are left untouched in the absorbing state
  int terminated;                  // -1: leave as is
  int timestep_op;                 // 0 leave, 1 set to 0, 2 increment
  int n_iter;
  double soc[T::NDES > 0 ? T::NDES : 1];
  bool write_soc;
  bool inc_reset;
  int aux;
};

template <class T, int KCAP = Layout<T>::KMAX>  // KCAP: compile-time cap on the number of aux variables
struct StepOut : StepFlags<T> {      // ... with the rows in registers
  static constexpr int CAP = KCAP;
  double state[T::SDIM + KCAP], obs[T::SDIM + KCAP];
  ANM_HD void put_st(int k, double v) { state[k] = v; }
  ANM_HD void put_ob(int k, double v) { obs[k] = v; }
};

// what step_end needs to know about how the step started
template <class T>
struct StepCtx {
  bool absorbing;   // env was terminated and is not being re-initialised: nothing to solve
  bool resetting;   // autoreset: the transition is the first one of a freshly drawn initial state
  int aux;          // new time index (series mode)
  double soc_req[T::NDES > 0 ? T::NDES : 1];  // resetting: SoC requested by the drawn state (MWh slot)
};

// first half of a step: inputs -> device maps, bus sums, up to `iter_cap` Newton iterations
template <class T, class JT, class CD, class SER = const double*>
ANM_HD void step_begin(cptr_t C, CD Cd, const EnvIO& io, SolverOpts so, int64_t e, const StepIn<T>& in, StepCtx<T>& ctx,
                       EnvWork<T>& w, PFState<T>& st, int iter_cap, SER ser = nullptr) {
  if constexpr (std::is_same<SER, const double*>::value) { if (!ser) ser = io.series; }
  typedef Layout<T> L;
  const bool series = io.exo == nullptr;
  ctx.resetting = in.was_term && io.autoreset && series;
  ctx.absorbing = in.was_term && !ctx.resetting;
  ctx.aux = 0;
  pf_idle<T>(st);
  ANM_PHASE(0);
  if (ctx.absorbing) return;

  double P_load[T::NLOAD > 0 ? T::NLOAD : 1], P_pot[T::NGEN > 0 ? T::NGEN : 1];
  double P_set[T::NSET > 0 ? T::NSET : 1], Q_set[T::NSET > 0 ? T::NSET : 1];
  int aux = 0;
  if (ctx.resetting) {
    double s0[T::SDIM + 1];  // sampled initial state (K == 1 in series mode)
    aux = sample_series_init_state<T>(C, io, e, uint32_t(in.reset_count), s0, ser);
    static_for<0, T::NDES>([&](auto I) { ctx.soc_req[I] = s0[2 * T::ND + I]; });
    inputs_from_init_state<T>(C, s0, w, P_load, P_pot, P_set, Q_set);
  } else {
    // 1. exogenous variables (next_vars, anm6_easy.py:54-65 in series mode)
    if (series) {
      // (a caller that keeps the time index as an integer has formed the next one itself: in.aux_next >= 0)
      aux = in.aux_next >= 0 ? in.aux_next : int(fmod(in.aux_prev + 1.0, double(io.period)));
      static_for<0, T::NLOAD>([&](auto I) { P_load[I] = ser[I * io.period + aux]; });
      static_for<0, T::NGEN>([&](auto I) { P_pot[I] = ser[(T::NLOAD + I) * io.period + aux]; });
    } else {
      static_for<0, T::NLOAD>([&](auto I) { P_load[I] = in.exo[I]; });
      static_for<0, T::NGEN>([&](auto I) { P_pot[I] = in.exo[T::NLOAD + I]; });
    }
    // 2. action layout [P_gen.., Q_gen.., P_des.., Q_des..] by ascending device id (anm_env.py:393-410)
    static_for<0, T::ND>([&](auto Di) {
      constexpr int d = Di;
      constexpr int typ = T::DEV_TYPE[d];
      if constexpr (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) {
        P_set[T::DEV_SET[d]] = in.action[T::DEV_SLOT[d]];
        Q_set[T::DEV_SET[d]] = in.action[T::NGEN + T::DEV_SLOT[d]];
      } else if constexpr (typ == DEV_STORAGE) {
        P_set[T::DEV_SET[d]] = in.action[2 * T::NGEN + T::DEV_SLOT[d]];
        Q_set[T::DEV_SET[d]] = in.action[2 * T::NGEN + T::NDES + T::DEV_SLOT[d]];
      }
    });
    static_for<0, T::NDES>([&](auto I) { w.soc[I] = in.soc[I]; });
  }
  ctx.aux = aux;
  // 3. one simulator transition, shared by the step and the autoreset path
  ANM_PHASE(1);
  transition_begin<T, JT>(C, Cd, w, st, P_load, P_pot, P_set, Q_set, so.tol, so.max_iter, iter_cap);
  (void)L::TOTAL;
}

// second half: solution -> flows, reward, clipping, terminal handling, state and observation rows
// DO_TRANSITION_END = false: the caller has already run transition_end (it reads the electrical state in
// between, see op_step_general)
template <class T, int KCAP, bool DO_TRANSITION_END = true, class Out = StepOut<T, KCAP>>
ANM_HD void step_end(cptr_t C, const EnvIO& io, SolverOpts so, int64_t e, const StepCtx<T>& ctx, EnvWork<T>& w,
                     const PFState<T>& st, Out& out) {
  typedef Layout<T> L;
  const bool series = io.exo == nullptr;
  out.write_state = out.write_obs = out.write_costs = out.write_soc = out.inc_reset = false;
  out.terminated = -1;
  out.timestep_op = 0;
  out.n_iter = 0;
  out.aux = ctx.aux;
  out.reward = out.e_loss = out.penalty = 0.0;
  if (ctx.absorbing) {  // absorbing terminal state (anm_env.py:365-367)
    static_for<0, T::SDIM + KCAP>([&](auto Kc) { out.put_ob(Kc, 0.0); });
    out.write_obs = true;
    return;
  }
  if (DO_TRANSITION_END) transition_end<T>(C, w, st, so.tol);
  out.n_iter = w.n_iter;
  out.write_soc = true;
  out.write_state = out.write_obs = true;

  if (ctx.resetting) {
    // tail of Simulator.reset / ANMEnv.reset for the drawn state: only its SoC and time index matter
    double s0[T::SDIM + 1];
    static_for<0, T::NDES>([&](auto I) { s0[2 * T::ND + I] = ctx.soc_req[I]; });
    s0[T::SDIM] = double(ctx.aux);
    finish_reset<T, 1>(C, w, s0, 1, out.soc, out);
    out.inc_reset = true;
    out.terminated = w.converged ? 0 : 1;  // not converged: try another draw at the next call
    if (!w.converged) {  // ... and look exactly like the absorbing terminal state until then
      static_for<0, T::SDIM + 1>([&](auto Kc) {
        out.put_st(Kc, 0.0);
        out.put_ob(Kc, 0.0);
      });
    }
    out.timestep_op = 1;
    out.write_costs = true;  // reward = e_loss = penalty = 0
    return;
  }

  static_for<0, T::NDES>([&](auto I) { out.soc[I] = w.soc[I]; });
  const bool term = !w.converged;
  out.terminated = term ? 1 : 0;
  out.write_costs = true;
  const double c1 = C[L::SCALARS + SC_C1], c2 = C[L::SCALARS + SC_C2];
  if (!term) {
    // reward clipping (anm_env.py:423-427)
    const double sg = (w.e_loss > 0.0) ? 1.0 : ((w.e_loss < 0.0) ? -1.0 : 0.0);
    const double el = sg * vmin(fabs(w.e_loss), c1);
    const double pn = vmin(vmax(w.penalty, 0.0), c2);
    out.e_loss = el;
    out.penalty = pn;
    out.reward = -(el + pn);
    write_state_obs<T>(C, w, out);
    if (series) {
      out.put_st(T::SDIM, double(ctx.aux));
      out.put_ob(T::SDIM, vmin(vmax(double(ctx.aux), C[L::OBS_LO + T::SDIM]), C[L::OBS_HI + T::SDIM]));
    } else {
      if (io.K > 0) {
        static_for<0, KCAP>([&](auto Kc) {  // unconditional stores, see finish_reset
          constexpr int k = Kc;
          const double v = io.aux_next[e * (io.aux_stride > 0 ? io.aux_stride : io.K) + (k < io.K ? k : 0)];
          out.put_st(T::SDIM + k, v);
          out.put_ob(T::SDIM + k, fmin(fmax(v, C[L::OBS_LO + T::SDIM + k]), C[L::OBS_HI + T::SDIM + k]));
        });
      }
    }
  } else {
    out.reward = C[L::SCALARS + SC_RTERM];  // -c2 / (1 - gamma)
    out.e_loss = c1;
    out.penalty = c2;
    static_for<0, T::SDIM + KCAP>([&](auto Kc) {
      out.put_st(Kc, 0.0);
      out.put_ob(Kc, 0.0);
    });
  }
  out.timestep_op = 2;
}

// one-shot composition
template <class T, class JT, int KCAP>
ANM_HD void step_compute(cptr_t C, const EnvIO& io, SolverOpts so, int64_t e, const StepIn<T>& in,
                         StepOut<T, KCAP>& out, EnvWork<T>& w) {
  StepCtx<T> ctx;
  PFState<T> st;
  step_begin<T, JT>(C, C, io, so, e, in, ctx, w, st, so.max_iter);
  step_end<T, KCAP>(C, io, so, e, ctx, w, st, out);
}

// everything of StepOut except the state / obs rows
template <class T, int KCAP>
ANM_HD void store_step_scalars(const EnvIO& io, int64_t e, const StepFlags<T>& o, bool have_ts = false,
                               int32_t ts_prev = 0) {  // have_ts: the caller already read timestep[e]
  if (o.write_soc) static_for<0, T::NDES>([&](auto I) { io.soc[e * T::NDES + I] = o.soc[I]; });
  if (o.terminated >= 0) io.terminated[e] = uint8_t(o.terminated);
  io.reward[e] = o.reward;
  if (o.write_costs) {
    io.e_loss[e] = o.e_loss;
    io.penalty[e] = o.penalty;
  }
  if (io.nr_iters) io.nr_iters[e] = o.n_iter;
  if (io.timestep) {
    if (o.timestep_op == 1) io.timestep[e] = 0;
    else if (o.timestep_op == 2) io.timestep[e] = (have_ts ? ts_prev : io.timestep[e]) + 1;
  }
  if (o.inc_reset) io.reset_count[e] += 1;
}

// I/O layer 1: plain per-environment loads and stores (host test double, generic mode, `full` dumps)
template <class T, class JT>
ANM_HD void op_step(cptr_t C, const EnvIO& io, SolverOpts so, int64_t e) {
  typedef Dims<T> D;
  const int S = T::SDIM + io.K;
  StepIn<T> in;
  StepOut<T> out;
  EnvWork<T> w;
  in.was_term = io.terminated[e] != 0;
  static_for<0, D::ADIM>([&](auto I) { in.action[I] = io.action[e * D::ADIM + I]; });
  if (io.exo) static_for<0, D::NEXO>([&](auto I) { in.exo[I] = io.exo[e * D::NEXO + I]; });
  static_for<0, T::NDES>([&](auto I) { in.soc[I] = io.soc[e * T::NDES + I]; });
  in.aux_prev = io.exo ? 0.0 : io.state[e * S + T::SDIM];
  in.reset_count = (io.autoreset && io.reset_count) ? io.reset_count[e] : 0;
  step_compute<T, JT, Layout<T>::KMAX>(C, io, so, e, in, out, w);
  double* state = io.state + e * S;
  double* obs = io.obs + e * S;
  static_for<0, T::SDIM + Layout<T>::KMAX>([&](auto Kc) {
    constexpr int k = Kc;
    if (k < S) {
      if (out.write_state) state[k] = out.state[k];
      if (out.write_obs) obs[k] = out.obs[k];
    }
  });
  store_step_scalars<T, Layout<T>::KMAX>(io, e, out);
  if (io.full && out.write_state) write_full_state<T>(w, io.full + e * FullState<T>::SIZE);
  ANM_PHASE(6);
}

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------------------------
// Straggler records (two-phase step).  A diverging Newton solve runs to the iteration cap while the
// 63 other environments of its wave finished long ago.  The first launch therefore stops after
// `iter_cap` iterations; every environment that is still iterating then writes its iterate to a
// record (slot from an atomic counter) and a second launch continues those, densely packed, with the
// same code, so the results are bit-identical to the one-launch path.
// ---------------------------------------------------------------------------------------------
template <class T>
struct Rec {
  static constexpr int E = 0, IT = 1, RESET = 2, AUX = 3, SOCREQ = 4, VM = SOCREQ + T::NDES,
                       CS = VM + T::NB, SN = CS + T::NB, BUSP = SN + T::NB, BUSQ = BUSP + T::NB, DEVP = BUSQ + T::NB,
                       DEVQ = DEVP + T::ND, PPOT = DEVQ + T::ND, SOC = PPOT + T::NGEN, IN_SIZE = SOC + T::NDES,
                       OUT_SIZE = 2 * (T::SDIM + 1) + T::NDES + 8,  // ResRec<T>::SIZE
                       SIZE = IN_SIZE > OUT_SIZE ? IN_SIZE : OUT_SIZE;
  static constexpr int HEADER = 8;  // doubles reserved at the start of the workspace (two int32 counters)
};

// layout of a record after the straggler launch (results; slot 0 still holds the environment index)
template <class T>
struct ResRec {
  static constexpr int STATE = 1, OBS = STATE + T::SDIM + 1, SOC = OBS + T::SDIM + 1, REWARD = SOC + T::NDES,
                       ELOSS = REWARD + 1, PENALTY = ELOSS + 1, NITER = PENALTY + 1, TERM = NITER + 1, TSOP = TERM + 1,
                       FLAGS = TSOP + 1, SIZE = FLAGS + 1;
};

template <class T>
__device__ void save_record(double* r, int64_t e, const StepCtx<T>& ctx, const EnvWork<T>& w, const PFState<T>& st) {
  typedef Rec<T> R;
  r[R::E] = double(e);
  r[R::IT] = double(st.it);
  r[R::RESET] = ctx.resetting ? 1.0 : 0.0;
  r[R::AUX] = double(ctx.aux);
  static_for<0, T::NDES>([&](auto I) { r[R::SOCREQ + I] = ctx.soc_req[I]; r[R::SOC + I] = w.soc[I]; });
  static_for<0, T::NB>([&](auto I) {
    r[R::VM + I] = w.vm[I]; r[R::CS + I] = st.cs[I]; r[R::SN + I] = st.sn[I];
    r[R::BUSP + I] = w.bus_p[I]; r[R::BUSQ + I] = w.bus_q[I];
  });
  static_for<0, T::ND>([&](auto I) { r[R::DEVP + I] = w.dev_p[I]; r[R::DEVQ + I] = w.dev_q[I]; });
  static_for<0, T::NGEN>([&](auto I) { r[R::PPOT + I] = w.p_pot[I]; });
}

template <class T>
__device__ int64_t load_record(const double* r, StepCtx<T>& ctx, EnvWork<T>& w, PFState<T>& st) {
  typedef Rec<T> R;
  st.it = int(r[R::IT]);
  ctx.absorbing = false;
  ctx.resetting = r[R::RESET] != 0.0;
  ctx.aux = int(r[R::AUX]);
  static_for<0, T::NDES>([&](auto I) { ctx.soc_req[I] = r[R::SOCREQ + I]; w.soc[I] = r[R::SOC + I]; });
  static_for<0, T::NB>([&](auto I) {
    w.vm[I] = r[R::VM + I]; st.cs[I] = r[R::CS + I]; st.sn[I] = r[R::SN + I];
    w.bus_p[I] = r[R::BUSP + I]; w.bus_q[I] = r[R::BUSQ + I];
  });
  static_for<0, T::ND>([&](auto I) { w.dev_p[I] = r[R::DEVP + I]; w.dev_q[I] = r[R::DEVQ + I]; });
  static_for<0, T::NGEN>([&](auto I) { w.p_pot[I] = r[R::PPOT + I]; });
  return int64_t(r[R::E]);
}

template <class T, bool FULL>
__device__ __forceinline__ void epilogue_stores(const EnvIO& io, int64_t e0, int64_t e, int lane, bool store, int32_t ts_prev,
                                                StepOut<T, 1>& out, const EnvWork<T>& w, double* lds) {
  constexpr int S = T::SDIM + 1;
  constexpr int SP = S + 1;
  // obs = clip(state) almost always IS the state: the duplicate row is then not written, only a flag
  bool state_dup = false;
  if (io.state_same) {
    bool same = out.write_state && out.write_obs;
    static_for<0, S>([&](auto Kc) { same = same && (out.state[Kc] == out.obs[Kc]); });
    state_dup = same;
  }
  if (store) {
    store_step_scalars<T, 1>(io, e, out, true, ts_prev);
    if (io.state_same && out.write_state) io.state_same[e] = state_dup ? 1 : 0;
    if (out.write_state) io.aux_index[e] = int32_t(out.state[T::SDIM]);
    if constexpr (FULL) {
      if (io.full && out.write_state) write_full_state<T>(w, io.full + e * FullState<T>::SIZE);
    }
  }
  // ---- coalesced stores of the state and obs rows
  auto store_rows = [&](double* gbase, const double* row, bool wr) {
    const unsigned long long mask = __ballot(wr && store);
    if (mask == 0ull) return;   // no row of this wavefront is written (the state rows that equal their obs rows: the usual case)
    static_for<0, S>([&](auto K) { lds[lane * SP + K] = row[K]; });
    ANM_WAVE_SYNC();
    // element j*64 + lane of the block is (row l, column k); all LDS reads first, then the stores
    double v[S];
    if (mask == ~0ull) {        // every row is written (the obs rows of a full wavefront): no per-element predicate
      static_for<0, S>([&](auto J) {
        const int idx = J * 64 + lane;
        const int l = idx / S, k = idx - l * S;
        v[J] = lds[l * SP + k];
      });
      static_for<0, S>([&](auto J) { gbase[J * 64 + lane] = v[J]; });
    } else {
      bool on[S];
      static_for<0, S>([&](auto J) {
        const int idx = J * 64 + lane;
        const int l = idx / S, k = idx - l * S;
        v[J] = lds[l * SP + k];
        on[J] = ((mask >> l) & 1ull) != 0;
      });
      static_for<0, S>([&](auto J) {
        if (on[J]) gbase[J * 64 + lane] = v[J];
      });
    }
    ANM_WAVE_SYNC();
  };
  store_rows(io.state + e0 * S, out.state, out.write_state && !state_dup);
  store_rows(io.obs + e0 * S, out.obs, out.write_obs);
  ANM_PHASE(6);
}

// I/O layer 1b (GPU): the step of the environments a batch VIEW names (anm_model_bind_view: a sub-batch of a larger batch
// whose rows are padded to common widths -- environments over different topologies in one batch, each topology stepped by
// its own kernel family).  Slot s of the launch is environment index[s]; its rows are scattered over the batch, so the
// loads and stores are per lane (the coalesced row blocks of layers 2 and 3 need 64 consecutive environments); the solve
// is the same as theirs, the in-wave hand-over of diverging solves to lane groups included.  K <= 1, "state" observation.
template <class T, class JT>
__device__ void op_step_view(cptr_t C, const EnvIO& io, SolverOpts so, int64_t n, const View& v, double* lds) {
  typedef Dims<T> D;
  constexpr int S1 = T::SDIM + 1;
  const int lane = threadIdx.x;
  const int64_t slot = int64_t(blockIdx.x) * 64 + lane;
  const bool valid = slot < n;
  const int64_t sc = valid ? slot : n - 1;
  const int64_t e = v.index ? int64_t(v.index[sc]) : sc;
  const int WA = v.w_action > 0 ? v.w_action : D::ADIM, WS = v.w_state > 0 ? v.w_state : T::SDIM + io.K;
  const int WD = v.w_des > 0 ? v.w_des : T::NDES, WX = v.w_exo > 0 ? v.w_exo : D::NEXO, WF = v.w_full > 0 ? v.w_full : FullState<T>::SIZE;
  const bool series = io.exo == nullptr;
  StepIn<T> in;
  StepOut<T, 1> out;
  StepCtx<T> ctx;
  PFState<T> st;
  EnvWork<T> w;
  in.was_term = io.terminated[e] != 0;
  static_for<0, D::ADIM>([&](auto I) { in.action[I] = io.action[e * WA + I]; });
  if (!series) static_for<0, D::NEXO>([&](auto I) { in.exo[I] = io.exo[e * WX + I]; });
  static_for<0, T::NDES>([&](auto I) { in.soc[I] = io.soc[e * WD + I]; });
  in.aux_prev = series ? io.state[e * WS + T::SDIM] : 0.0;
  in.reset_count = (io.autoreset && io.reset_count) ? io.reset_count[e] : 0;
  const int32_t ts_prev = io.timestep ? io.timestep[e] : 0;

  step_begin<T, JT>(C, C, io, so, e, in, ctx, w, st, -1);
  constexpr bool CAN_GROUP = T::TREE != 0;
  const int handoff = (CAN_GROUP && so.handoff >= 0 && so.handoff < so.max_iter) ? so.handoff : -1;
  pf_iterate<T, JT>(C, w, st, so.tol, so.max_iter, handoff >= 0 ? handoff : so.max_iter);
  if constexpr (CAN_GROUP) {
    if (handoff >= 0 && ANM_WAVE_ANY(st.active && valid))
      group::continue_in_groups<T, JT, true, ANM_INWAVE_REDUCE_ALWAYS != 0>(C, w, st, st.active && valid, so.tol, so.max_iter, lds);
  }
  step_end<T, 1>(C, io, so, e, ctx, w, st, out);
  if (!valid) return;
  // stores (store_step_scalars with the view's strides)
  if (out.write_soc) static_for<0, T::NDES>([&](auto I) { io.soc[e * WD + I] = out.soc[I]; });
  if (out.terminated >= 0) io.terminated[e] = uint8_t(out.terminated);
  io.reward[e] = out.reward;
  if (out.write_costs) {
    io.e_loss[e] = out.e_loss;
    io.penalty[e] = out.penalty;
  }
  if (io.nr_iters) io.nr_iters[e] = out.n_iter;
  if (io.timestep) {
    if (out.timestep_op == 1) io.timestep[e] = 0;
    else if (out.timestep_op == 2) io.timestep[e] = ts_prev + 1;
  }
  if (out.inc_reset) io.reset_count[e] += 1;
  const int SK = T::SDIM + io.K;   // (K <= 1)
  double* state = io.state + e * WS;
  double* obs = io.obs + e * WS;
  static_for<0, S1>([&](auto Kc) {
    constexpr int k = Kc;
    if (k < SK) {
      if (out.write_state) state[k] = out.state[k];
      if (out.write_obs) obs[k] = out.obs[k];
    }
  });
  if (io.full && out.write_state) write_full_state<T>(w, io.full + e * WF);
}

// I/O layer 2 (GPU, series mode, K = 1): one wavefront = 64 consecutive environments whose action /
// state / obs rows are contiguous in memory.  Rows travel through LDS so that every global access is
// a fully coalesced 512-byte wave transaction (the plain layer issues 16-byte stores with a 144-byte
// lane stride: every store instruction touches 64 cache lines and partial lines are written twice).
// The time index is kept in a compact int32 array instead of being re-read from the state rows.
// FULL: also write the `full` electrical-state dump (a separate instantiation: the dump costs the
// default kernel its second wave per SIMD)
// (Measured and left out, profiles/r05_c_*: the epilogue re-reading its dozen output pointers from the kernarg segment instead of
// keeping them in scalar registers through the whole kernel -- 80 fewer SGPR spills to vector lanes, nothing gained in the
// throughput regime and 1.8 us LOST on the headline, where the epilogue of the chain-holding wavefront then starts with a
// scalar load it has to wait for.)
// (Measured and left out, profiles/r05_c_*: the series table copied to LDS by every wavefront, so that the look-up by the time
// index is no second, dependent trip to memory -- 524 288 environments 176.3 -> 180.7 us: the trip hides behind the first
// scalar loads of the constants, the copy does not.)
// (Measured and left out, profiles/r05_h_*: one throw-away scalar load per 64-byte line of the constant buffer while the wavefront
// waits for its action rows, so that the dozen dependent batches of constant loads along the prologue hit the scalar cache --
// headline kernel 87.03 -> 86.79 us over four same-box pairs, 524 288 environments unchanged: not worth 40 instructions.)
template <class T, class JT, bool FULL>
__device__ void op_step_rows(cptr_t C, const EnvIO& io, SolverOpts so, int64_t n, double* lds) {
  typedef Dims<T> D;
  constexpr int S = T::SDIM + 1;
  constexpr int SP = S + 1;  // padded row stride (odd number of doubles: conflict-free column access)
  const int lane = threadIdx.x;
  const int64_t e0 = int64_t(blockIdx.x) * 64;
  const int64_t e = e0 + lane;
  const bool valid = e < n;
  const int64_t ec = valid ? e : n - 1;
  const int rows = int((n - e0) < 64 ? (n - e0) : 64);
  ANM_PHASE(7);  // kernel entry
  StepIn<T> in;
  StepOut<T, 1> out;
  StepCtx<T> ctx;
  PFState<T> st;
  EnvWork<T> w;
  // the per-environment scalars first: their loads are then in flight together with the action rows (behind the fences
  // of the transposition below they would wait for a second trip to memory)
  in.was_term = io.terminated[ec] != 0;
  static_for<0, T::NDES>([&](auto I) { in.soc[I] = io.soc[ec * T::NDES + I]; });
  const int32_t aux_prev_i = io.aux_index[ec];
  in.aux_prev = double(aux_prev_i);
  in.reset_count = io.autoreset ? io.reset_count[ec] : 0;
  const int32_t ts_prev = io.timestep ? io.timestep[ec] : 0;  // read now: the epilogue only stores

  // ---- coalesced loads: 64 x ADIM doubles of actions
  {
    const double* g = io.action + e0 * D::ADIM;
    constexpr int AP = D::ADIM + 1;
    // all loads are issued before the first use (one exposed memory latency, not ADIM of them):
    // out-of-range lanes re-read the last element instead of branching around the load
    const int last = rows * D::ADIM - 1;
    double tmp[D::ADIM > 0 ? D::ADIM : 1];
    static_for<0, D::ADIM>([&](auto J) {
      const int idx = J * 64 + lane;
      tmp[J] = g[idx < last ? idx : last];
    });
    static_for<0, D::ADIM>([&](auto J) {
      const int idx = J * 64 + lane;
      if (idx <= last) lds[(idx / D::ADIM) * AP + (idx % D::ADIM)] = tmp[J];
    });
    ANM_WAVE_SYNC();
    const int lr = valid ? lane : rows - 1;
    static_for<0, D::ADIM>([&](auto I) { in.action[I] = lds[lr * AP + I]; });
    ANM_WAVE_SYNC();
  }
  {
    // the next time index from the integer one: (t + 1) mod period for t in [0, period) without a division; anything else
    // (an initial state handed over with a time index beyond the period) goes through the reference's own expression
    const int t1 = aux_prev_i + 1;
    int an = t1 - (t1 >= io.period ? io.period : 0);
    if (ANM_WAVE_ANY(an < 0 || an >= io.period)) an = int(fmod(in.aux_prev + 1.0, double(io.period)));
    in.aux_next = an;
  }

  const bool two_phase = io.ws != nullptr && io.iter_cap < so.max_iter;
  step_begin<T, JT>(C, C, io, so, ec, in, ctx, w, st, -1);  // device maps, bus sums, flat start; iterated below
  bool pending = false;
  // in-wave straggler hand-over (anm_group.hpp): tree topologies.  In the two-launch mode the straggler launch
  // does the continuing (on lane groups as well, op_step_stragglers); only solves that found no record slot
  // continue here.
  constexpr bool CAN_GROUP = T::TREE != 0 && group::Shape<T>::NG * group::Slot<T>::SIZE <= 64 * (T::SDIM + 2);
  const int handoff = (CAN_GROUP && !two_phase && so.handoff >= 0 && so.handoff < so.max_iter) ? so.handoff : -1;
  const bool overflow_to_groups = CAN_GROUP && two_phase && so.handoff >= 0 && so.handoff < so.max_iter;
  int cap = two_phase ? io.iter_cap : (handoff >= 0 ? handoff : so.max_iter);
  // One copy of the Newton loop serves both passes (a second inlined copy costs registers and code):
  // pass 0 iterates up to `cap`; in two-launch mode the environments still iterating are then handed
  // to the straggler launch, and pass 1 only runs for those that found no record slot (thread mode only).
#pragma nounroll
  for (int pass = 0; pass < 2; ++pass) {
    pf_iterate<T, JT>(C, w, st, so.tol, so.max_iter, cap);
    if (!two_phase || pass == 1) break;
    int* cnt = reinterpret_cast<int*>(io.ws);  // cnt[0]: records of this step (zeroed by the scatter launch)
    {
      // one reservation per wavefront (the lanes' ranks order its slots), not one atomic per straggler
      const unsigned long long am = __ballot(valid && st.active);
      if (am != 0ull) {
        int base = 0;
        if (lane == 0) base = atomicAdd(cnt, __popcll(am));
        base = __builtin_amdgcn_readfirstlane(base);
        const int slot = base + __popcll(am & ((1ull << lane) - 1ull));
        if (valid && st.active && slot < io.ws_cap) {
          save_record<T>(io.ws + Rec<T>::HEADER + int64_t(slot) * Rec<T>::SIZE, e, ctx, w, st);
          pending = true;
        }
      }
    }
    if (pending) {  // handed over: this lane's own solve ends here (its outputs are not stored)
      st.diff = 0.0;
      st.active = false;
    }
    cap = so.max_iter;
    if (overflow_to_groups || !ANM_WAVE_ANY(st.active)) break;  // else: record space exhausted (or a padding lane)
  }
  if constexpr (CAN_GROUP) {
    if ((handoff >= 0 || overflow_to_groups) && ANM_WAVE_ANY(st.active && valid))
      group::continue_in_groups<T, JT, true, ANM_INWAVE_REDUCE_ALWAYS != 0>(C, w, st, st.active && valid, so.tol, so.max_iter, lds);
  }
  step_end<T, 1>(C, io, so, ec, ctx, w, st, out);
  const bool store = valid && !pending;
  epilogue_stores<T, FULL>(io, e0, e, lane, store, ts_prev, out, w, lds);
}

// ---------------------------------------------------------------------------------------------
// I/O layer 3 (GPU): the general step -- everything the fast path above leaves out:
//   * tasks whose next_vars() runs on the host (exo / aux_next given, K = 0..KMAX auxiliary variables),
//   * list-form observations (anm_env.py:497-521, 562-592): gathered, scaled and clipped inside this kernel
//     from the electrical state of the step, only the quantity classes the list names being computed,
//   * the optional `full` dump of the electrical state (Simulator.state).
// Same wave-per-64-environments shape and the same coalesced row traffic as op_step_rows; the electrical
// state of the 64 environments is staged in LDS (one row per lane, odd stride), read back transposed for the
// observation gather and for the dump.  Networks whose rows do not fit (FULL_OK false) keep the rows in
// registers: no fused list, plain per-lane dump.
// ---------------------------------------------------------------------------------------------
template <class T>
struct GenLds {
  static constexpr int KM = Layout<T>::KMAX;
  // identity layout of the rows (full dump): FullState + aux; fused paths need it to fit in LDS
  static constexpr int FSP = (FullState<T>::SIZE + KM) | 1;
  static constexpr bool FULL_OK = 64 * FSP * 8 <= 60 * 1024;
  static constexpr int SP = (T::SDIM + KM) | 1;
  static constexpr int B_MIN = 64 * (Dims<T>::ADIM + 1) > 64 * SP ? 64 * (Dims<T>::ADIM + 1) : 64 * SP;
  static constexpr int G_MIN = T::TREE != 0 ? group::Shape<T>::NG * group::Slot<T>::SIZE : 0;
  static constexpr int B_DOUBLES = B_MIN > G_MIN ? B_MIN : G_MIN;  // action / state / obs rows, hand-over slots
};

template <class T, class JT>
__device__ void op_step_general(cptr_t C, const EnvIO& io, SolverOpts so, int64_t n, double* lds) {
  typedef Dims<T> D;
  typedef GenLds<T> G;
  typedef Layout<T> L;
  constexpr int KM = G::KM;
  const int K = io.K;
  const int S = T::SDIM + K;
  const int lane = threadIdx.x;
  const int64_t e0 = int64_t(blockIdx.x) * 64;
  const int64_t e = e0 + lane;
  const bool valid = e < n;
  const int64_t ec = valid ? e : n - 1;
  const int rows = int((n - e0) < 64 ? (n - e0) : 64);
  // One LDS buffer (dynamic, sized by the launch), used in turn for: the action transposes, the hand-over
  // slots of the lane groups, the electrical-state rows [64][row_stride], the state / obs rows [64][SP]
  double* ldsA = lds;
  double* ldsB = lds;
  const int RS = io.row_stride;
  const bool series = io.exo == nullptr;
  StepIn<T> in;
  StepCtx<T> ctx;
  PFState<T> st;
  EnvWork<T> w;
  // ---- coalesced loads of the action rows (as op_step_rows)
  {
    const double* g = io.action + e0 * D::ADIM;
    constexpr int AP = D::ADIM + 1;
    const int last = rows * D::ADIM - 1;
    double tmp[D::ADIM > 0 ? D::ADIM : 1];
    static_for<0, D::ADIM>([&](auto J) {
      const int idx = J * 64 + lane;
      tmp[J] = g[idx < last ? idx : last];
    });
    static_for<0, D::ADIM>([&](auto J) {
      const int idx = J * 64 + lane;
      if (idx <= last) ldsB[(idx / D::ADIM) * AP + (idx % D::ADIM)] = tmp[J];
    });
    ANM_WAVE_SYNC();
    const int lr = valid ? lane : rows - 1;
    static_for<0, D::ADIM>([&](auto I) { in.action[I] = ldsB[lr * AP + I]; });
    ANM_WAVE_SYNC();
  }
  in.was_term = io.terminated[ec] != 0;
  static_for<0, T::NDES>([&](auto I) { in.soc[I] = io.soc[ec * T::NDES + I]; });
  if (!series) static_for<0, D::NEXO>([&](auto I) { in.exo[I] = io.exo[ec * D::NEXO + I]; });
  in.aux_prev = series ? (io.aux_index ? double(io.aux_index[ec]) : io.state[ec * S + T::SDIM]) : 0.0;
  in.reset_count = (io.autoreset && io.reset_count) ? io.reset_count[ec] : 0;
  const int32_t ts_prev = io.timestep ? io.timestep[ec] : 0;

  step_begin<T, JT>(C, C, io, so, ec, in, ctx, w, st, -1);
  constexpr bool CAN_GROUP = T::TREE != 0;
  const int handoff = (CAN_GROUP && so.handoff >= 0 && so.handoff < so.max_iter) ? so.handoff : -1;
  pf_iterate<T, JT>(C, w, st, so.tol, so.max_iter, handoff >= 0 ? handoff : so.max_iter);
  if constexpr (CAN_GROUP) {
    if (handoff >= 0 && ANM_WAVE_ANY(st.active && valid))
      group::continue_in_groups<T, JT, true, ANM_INWAVE_REDUCE_ALWAYS != 0>(C, w, st, st.active && valid, so.tol, so.max_iter, ldsB);
  }
  if (!ctx.absorbing) transition_end<T>(C, w, st, so.tol);

  // ---- electrical state of this environment -> its LDS row (before the state / observation rows are built:
  // the voltages, currents and flows die here)
  const bool list = io.n_obs > 0;
  bool dumped = false;
  if constexpr (G::FULL_OK) {
    if (list || io.full) {
      const unsigned need = io.obs_need;
      double* row = ldsA + lane * RS;
      if (!ctx.absorbing) {
        emit_full_state<T>(w, need, [&](unsigned c, int i, double v) { row[io.cls_off[c] + i] = v; });
        if (ctx.resetting && ((need >> FC_DES_SOC) & 1u)) {
          // Simulator.reset overwrites the SoC with the requested one (simulator.py:284-288)
          const double base = C[L::SCALARS + SC_BASE];
          static_for<0, T::NDES>([&](auto I) { row[io.cls_off[FC_DES_SOC] + I] = ctx.soc_req[I] / base; });
        }
      }
      dumped = true;
    }
  }
  // the state row of this environment is built directly in LDS (behind the electrical-state rows)
  const int SPr = S | 1;
  double* ldsS = lds + io.state_row_off;
  struct RowOut : StepFlags<T> {
    double* row;
    int width;   // entries beyond the row (aux slots above K) are dropped: the next lane's row starts there
    __device__ __forceinline__ void put_st(int k, double v) { if (k < width) row[k] = v; }
    __device__ __forceinline__ void put_ob(int, double) {}   // formed from the state row when it is stored
  } out;
  out.row = ldsS + lane * SPr;
  out.width = S;
  step_end<T, KM, false, RowOut>(C, io, so, ec, ctx, w, st, out);
  const bool store = valid;
  if (store) {
    store_step_scalars<T, KM>(io, e, out, true, ts_prev);
    if (io.aux_index && out.write_state) io.aux_index[e] = int32_t(out.row[T::SDIM]);
  }
  const bool zero_obs = ctx.absorbing || out.terminated == 1;   // anm_env.py:365-367, 442-446
  const unsigned long long zero_rows = __ballot(zero_obs);
  const unsigned long long state_rows = __ballot(out.write_state && store);
  const unsigned long long obs_rows = __ballot(out.write_obs && store);

  if constexpr (G::FULL_OK) {
    if (dumped) {
      double* row = ldsA + lane * RS;
      for (int k = 0; k < K; ++k) row[io.aux_off + k] = out.row[T::SDIM + k];   // aux variables: behind the electrical state
      ANM_WAVE_SYNC();
      if (list) {  // obs[e0 + r, k] for flat index idx = r * n_obs + k: coalesced stores, LDS gathers
        const int total = rows * io.n_obs;
        double* gobs = io.obs + e0 * io.n_obs;
        for (int idx = lane; idx < total; idx += 64) {
          const int r = int(__umulhi(unsigned(idx), io.obs_magic));
          const int k = idx - r * io.n_obs;
          const double v = ldsA[r * RS + io.obs_index[k]] * io.obs_scale[k];
          const double c = fmin(fmax(v, io.obs_lo[k]), io.obs_hi[k]);
          if ((obs_rows >> r) & 1ull) gobs[idx] = ((zero_rows >> r) & 1ull) ? 0.0 : c;
        }
      }
      if (io.full) {
        constexpr int FS = FullState<T>::SIZE;
        const int total = rows * FS;
        double* gfull = io.full + e0 * FS;
        for (int idx = lane; idx < total; idx += 64) {
          const int r = idx / FS, k = idx - r * FS;
          if ((state_rows >> r) & 1ull) gfull[idx] = ldsA[r * RS + k];   // identity layout when dumping
        }
      }
    }
  } else {
    if (io.full && out.write_state && store) write_full_state<T>(w, io.full + e * FullState<T>::SIZE);
  }
  // ---- state rows through LDS, coalesced; the "state" observation is clip(state, Box) of the same rows
  // (zeros for terminal rows, anm_env.py:365-367, 442-446), formed while they are stored
  ANM_WAVE_SYNC();
  {
    const int total = rows * S;
    double* gstate = io.state + e0 * S;
    double* gobs = io.obs + e0 * S;
    for (int idx = lane; idx < total; idx += 64) {
      const int r = int(__umulhi(unsigned(idx), io.state_magic));
      const int k = idx - r * S;
      const double v = ldsS[r * SPr + k];
      if ((state_rows >> r) & 1ull) gstate[idx] = v;
      if (!list && ((obs_rows >> r) & 1ull)) {
        const double c = fmin(fmax(v, C[L::OBS_LO + k]), C[L::OBS_HI + k]);
        gobs[idx] = ((zero_rows >> r) & 1ull) ? 0.0 : c;
      }
    }
  }
}

// second launch of the two-phase step: continue the handed-over solves.
// GROUPS (tree topologies, unless the hand-over to lane groups is switched off): a wavefront takes as many
// records as it has lane groups (8 for ANM6) and runs them through group::continue_in_groups -- the trip of a
// lane group is half as long as a thread's, and these solves are the ones that run ~94 more of them; the
// records of a million environments still fill the chip 8 per wavefront.  Otherwise: 64 records per
// wavefront, one per thread.
// Two levels (GROUPS, io.mid_cap > 0).  A wavefront runs until its LAST group is done, and of the solves handed
// over, most are slow convergers (a few more iterations) while the diverging ones run to the reference's cap: one
// of those among its eight records keeps a wavefront for ~94 trips.  So the first straggler launch (level 1) stops
// at mid_cap iterations, saves the iterates of the records still running back into their records and lists them;
// a second launch (level 2) takes the listed records, again 8 per wavefront -- now all long runners -- to the end.
// 1 M environments: ~20 000 records at hand-over, ~5 000 of them diverging: 2 500 wavefronts x 94 trips become
// 2 500 x 6 + 625 x 88.  Restarting from a saved iterate is bit-transparent (the first trip evaluates F of it).
template <class T, class JT, bool GROUPS>
__device__ void op_step_stragglers(cptr_t C, const EnvIO& io, SolverOpts so, double* lds, int level) {
  constexpr int S = T::SDIM + 1;
  int* cnt = reinterpret_cast<int*>(io.ws);
  int n_rec = cnt[0];
  if (n_rec > io.ws_cap) n_rec = int(io.ws_cap);
  // The count is handed to the scatter launch in cnt[1] so that the scatter launch can zero cnt[0] for the
  // next step: every step leaves the counters as it found them (no host-side parity, no memset), which is
  // what makes one captured step replayable from a HIP graph any number of times.
  if (level == 1 && blockIdx.x == 0 && threadIdx.x == 0) cnt[1] = n_rec;
  const bool two_level = GROUPS && io.mid_cap > 0 && io.mid_cap < so.max_iter;
  const int n_work = level == 2 ? cnt[2] : n_rec;
  const int cap = (level == 1 && two_level) ? io.mid_cap : so.max_iter;
  constexpr int PER = GROUPS ? group::Shape<T>::NG : 64;
  // (one pass: the grid covers the workspace; a loop over passes costs the kernel its second wavefront per SIMD)
  const int64_t base = int64_t(blockIdx.x) * PER;
  if (base >= n_work) return;
  do {
    StepCtx<T> ctx;
    PFState<T> st;
    EnvWork<T> w;
    StepOut<T, 1> out;
    int64_t j, e = 0;
    bool mine;
    if constexpr (GROUPS) {
      const int lane = threadIdx.x & 63;
      j = base + lane;
      mine = lane < PER && j < n_work;
    } else {
      // Dense packing: 64 consecutive records per wavefront.  Measured alternatives (MI355X): dealing the
      // records one per wavefront makes every thread-mode iteration 2-3x slower once several sparse wavefronts
      // share a CU (issue stalls, SQ_WAIT_INST_ANY 60 %), although each wave then only runs the sin/cos tier its
      // own solve needs.  One record per thread: the Newton loop keeps every value in registers.
      j = base + threadIdx.x;
      mine = j < n_work;
    }
    if (level == 2 && mine) j = io.ws_list2[j];
    double* r = io.ws + Rec<T>::HEADER + (mine ? j : 0) * Rec<T>::SIZE;
    if (mine) e = load_record<T>(r, ctx, w, st);
    if constexpr (GROUPS) {
      // (the first trip of newton_groups evaluates F of the saved iterate before anything else)
      st.active = mine;
      group::continue_in_groups<T, JT>(C, w, st, mine, so.tol, cap, lds);
      if (!mine) continue;
      if (cap < so.max_iter && st.diff == INFINITY) {  // stopped by mid_cap, still running: level 2 takes it
        typedef Rec<T> R;
        r[R::IT] = double(st.it);
        static_for<0, T::NB>([&](auto I) { r[R::VM + I] = w.vm[I]; r[R::CS + I] = st.cs[I]; r[R::SN + I] = st.sn[I]; });
        io.ws_list2[atomicAdd(cnt + 2, 1)] = int32_t(j);
        continue;
      }
      w.vr[0] = 1.0;   // the slack bus: what eval_mismatch leaves there (continue_in_groups restores the others)
      w.vi[0] = 0.0;
    } else {
      if (!mine) continue;
      st.fresh = true;  // F and diff of the saved iterate are recomputed (same code, same inputs, same bits)
      pf_iterate<T, JT>(C, w, st, so.tol, so.max_iter, so.max_iter);
    }
    step_end<T, 1>(C, io, so, e, ctx, w, st, out);
    // The results go back into the record (r[0] keeps the environment index); the scatter launch
    // writes them to the batch arrays.  Keeping the dozen output pointers out of this kernel keeps its
    // Newton loop free of scalar-register spills.
    typedef ResRec<T> Q;
    static_for<0, S>([&](auto K) { r[Q::STATE + K] = out.state[K]; r[Q::OBS + K] = out.obs[K]; });
    static_for<0, T::NDES>([&](auto I) { r[Q::SOC + I] = out.soc[I]; });
    r[Q::REWARD] = out.reward; r[Q::ELOSS] = out.e_loss; r[Q::PENALTY] = out.penalty;
    r[Q::NITER] = double(out.n_iter); r[Q::TERM] = double(out.terminated); r[Q::TSOP] = double(out.timestep_op);
    r[Q::FLAGS] = double((out.write_state ? 1 : 0) | (out.write_obs ? 2 : 0) | (out.write_costs ? 4 : 0) |
                         (out.write_soc ? 8 : 0) | (out.inc_reset ? 16 : 0));
  } while (false);
}

// third launch: scatter the straggler results to the batch arrays.  SCATTER_LANES lanes per record: a record is one
// contiguous run of doubles and so are the state / observation rows it goes to -- consecutive lanes move consecutive
// doubles (a thread per record read and wrote with a stride of a whole record: 16 us for 20 000 records at 1 M
// environments, most of it uncoalesced traffic)
constexpr int SCATTER_LANES = 16;
template <class T>
__device__ void op_step_scatter(const EnvIO& io) {
  constexpr int S = T::SDIM + 1;
  typedef ResRec<T> Q;
  int* cnt = reinterpret_cast<int*>(io.ws);
  const int n_rec = cnt[1];  // written by the straggler launch (already clamped to the workspace)
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // nobody reads these in this launch
    cnt[0] = 0;
    cnt[2] = 0;
  }
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t j = t / SCATTER_LANES;
  const int l = int(t % SCATTER_LANES);
  if (j >= n_rec) return;
  const double* r = io.ws + Rec<T>::HEADER + j * Rec<T>::SIZE;
  const int64_t e = int64_t(r[0]);
  const int flags = int(r[Q::FLAGS]);
  if (l == 0) {
    StepOut<T, 1> out;
    out.write_state = flags & 1; out.write_obs = flags & 2; out.write_costs = flags & 4;
    out.write_soc = flags & 8; out.inc_reset = flags & 16;
    out.reward = r[Q::REWARD]; out.e_loss = r[Q::ELOSS]; out.penalty = r[Q::PENALTY];
    out.n_iter = int(r[Q::NITER]); out.terminated = int(r[Q::TERM]); out.timestep_op = int(r[Q::TSOP]);
    static_for<0, T::NDES>([&](auto I) { out.soc[I] = r[Q::SOC + I]; });
    store_step_scalars<T, 1>(io, e, out);
    if (flags & 1) {
      if (io.state_same) io.state_same[e] = 0;
      io.aux_index[e] = int32_t(r[Q::STATE + T::SDIM]);
    }
  }
  if (flags & 1)
    for (int k = l; k < S; k += SCATTER_LANES) io.state[e * S + k] = r[Q::STATE + k];
  if (flags & 2)
    for (int k = l; k < S; k += SCATTER_LANES) io.obs[e * S + k] = r[Q::OBS + k];
}
#endif

}  // namespace anm
