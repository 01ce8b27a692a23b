// anm_env_ops.hpp -- the three per-environment operations behind the C ABI, written once for
// device (gfx950 kernels in anm_capi.hip) and host (tests/hostsim test double).
//
//   op_transition  Simulator.transition                       simulator.py:464-537
//   op_reset       Simulator.reset + ANMEnv.reset tail        simulator.py:225-293, anm_env.py:292-311
//   op_step        ANMEnv.step (+ fused next_vars / autoreset) anm_env.py:333-453, anm6_easy.py:25-65
#pragma once

#include "anm_device.hpp"

namespace anm {

struct SolverOpts {
  double tol;
  int max_iter;
};

template <class T>
struct Dims {
  static constexpr int NEXO = T::NLOAD + T::NGEN;
  static constexpr int ADIM = 2 * (T::NGEN + T::NDES);
  static constexpr int SBASE = T::SDIM;  // 2 ND + NDES + NGEN
};

// state vector (anm_env.py:139-147): [dev_p MW, dev_q MVAr, des_soc MWh, gen_p_max MW, aux(K)]
template <class T>
ANM_HD void write_state_obs(cptr_t C, const EnvWork<T>& w, double* state, double* obs) {
  typedef Layout<T> L;
  const double base = C[L::SCALARS + SC_BASE];
  auto put = [&](int k, double v) {
    state[k] = v;
    // np.clip(obs, low, high)
    obs[k] = fmin(fmax(v, C[L::OBS_LO + k]), C[L::OBS_HI + k]);
  };
  static_for<0, T::ND>([&](auto Di) {
    constexpr int d = Di;
    put(d, w.dev_p[d] * base);
    put(T::ND + d, w.dev_q[d] * base);
  });
  static_for<0, T::NDES>([&](auto E) { put(2 * T::ND + E, w.soc[E] * base); });
  static_for<0, T::NGEN>([&](auto G) { put(2 * T::ND + T::NDES + G, w.p_pot[G] * base); });
}

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
};

template <class T, class JT>
ANM_HD void op_transition(cptr_t C, const TransitionIO& io, SolverOpts so, int64_t e) {
  EnvWork<T> w;
  double P_load[T::NLOAD > 0 ? T::NLOAD : 1], P_pot[T::NGEN > 0 ? T::NGEN : 1];
  double P_set[T::NSET > 0 ? T::NSET : 1], Q_set[T::NSET > 0 ? T::NSET : 1];
  static_for<0, T::NLOAD>([&](auto I) { P_load[I] = io.p_load[e * T::NLOAD + I]; });
  static_for<0, T::NGEN>([&](auto I) { P_pot[I] = io.p_pot[e * T::NGEN + I]; });
  static_for<0, T::NSET>([&](auto I) {
    P_set[I] = io.p_set[e * T::NSET + I];
    Q_set[I] = io.q_set[e * T::NSET + I];
  });
  static_for<0, T::NDES>([&](auto I) { w.soc[I] = io.soc[e * T::NDES + I]; });
  transition<T, JT>(C, w, P_load, P_pot, P_set, Q_set, so.tol, so.max_iter);
  static_for<0, T::NDES>([&](auto I) { io.soc[e * T::NDES + I] = w.soc[I]; });
  io.reward[e] = w.reward;
  io.e_loss[e] = w.e_loss;
  io.penalty[e] = w.penalty;
  io.converged[e] = w.converged ? 1 : 0;
  if (io.nr_iters) io.nr_iters[e] = w.n_iter;
  if (io.full) write_full_state<T>(w, io.full + e * FullState<T>::SIZE);
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
};

// Split an init_state row (anm_env.py / simulator.py:248-268) into transition inputs.
template <class T>
ANM_HD void inputs_from_init_state(cptr_t C, const double* s0, EnvWork<T>& w,
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
template <class T>
ANM_HD void finish_reset(cptr_t C, EnvWork<T>& w, const double* s0, int K, double* soc, double* state,
                         double* obs) {
  typedef Layout<T> L;
  const double base = C[L::SCALARS + SC_BASE];
  static_for<0, T::NDES>([&](auto E) {
    w.soc[E] = s0[2 * T::ND + E] / base;  // simulator.py:284-288
    soc[E] = w.soc[E];
  });
  write_state_obs<T>(C, w, state, obs);
  for (int k = 0; k < K; ++k) {
    const double a = s0[T::SDIM + k];
    state[T::SDIM + k] = a;
    obs[T::SDIM + k] = fmin(fmax(a, C[L::OBS_LO + T::SDIM + k]), C[L::OBS_HI + T::SDIM + k]);
  }
}

template <class T, class JT>
ANM_HD void op_reset(cptr_t C, const EnvIO& io, SolverOpts so, int64_t e) {
  if (io.mask && !io.mask[e]) return;
  const int S = T::SDIM + io.K;
  EnvWork<T> w;
  double P_load[T::NLOAD > 0 ? T::NLOAD : 1], P_pot[T::NGEN > 0 ? T::NGEN : 1];
  double P_set[T::NSET > 0 ? T::NSET : 1], Q_set[T::NSET > 0 ? T::NSET : 1];
  const double* s0 = io.init_state + e * S;
  inputs_from_init_state<T>(C, s0, w, P_load, P_pot, P_set, Q_set);
  transition<T, JT>(C, w, P_load, P_pot, P_set, Q_set, so.tol, so.max_iter);
  finish_reset<T>(C, w, s0, io.K, io.soc + e * T::NDES, io.state + e * S, io.obs + e * S);
  io.converged[e] = w.converged ? 1 : 0;
  io.terminated[e] = 0;
  if (io.timestep) io.timestep[e] = 0;
  if (io.nr_iters) io.nr_iters[e] = w.n_iter;
  if (io.full) write_full_state<T>(w, io.full + e * FullState<T>::SIZE);
}

template <class T, class JT>
ANM_HD void op_step(cptr_t C, const EnvIO& io, SolverOpts so, int64_t e) {
  typedef Layout<T> L;
  typedef Dims<T> D;
  const int S = T::SDIM + io.K;
  double* state = io.state + e * S;
  double* obs = io.obs + e * S;
  const bool was_term = io.terminated[e] != 0;
  const bool series = io.exo == nullptr;
  const bool resetting = was_term && io.autoreset && series;
  ANM_PHASE(0);

  if (was_term && !resetting) {  // absorbing terminal state (anm_env.py:365-367)
    for (int k = 0; k < S; ++k) obs[k] = 0.0;
    io.reward[e] = 0.0;
    if (io.nr_iters) io.nr_iters[e] = 0;
    return;
  }

  EnvWork<T> w;
  double P_load[T::NLOAD > 0 ? T::NLOAD : 1], P_pot[T::NGEN > 0 ? T::NGEN : 1];
  double P_set[T::NSET > 0 ? T::NSET : 1], Q_set[T::NSET > 0 ? T::NSET : 1];
  double s0[T::SDIM + 1];  // sampled initial state (autoreset only; K == 1 in series mode)
  int aux = 0;

  if (resetting) {
    // ANM6Easy.init_state (anm6_easy.py:25-52) with a counter-based RNG
    const uint32_t epoch = uint32_t(io.reset_count[e]);
    uint32_t r[4];
    Philox::generate(io.rng_seed, io.env_offset + uint64_t(e), epoch, 0u, r);
    aux = int((uint64_t(r[0]) * uint64_t(io.period)) >> 32);
    static_for<0, T::SDIM>([&](auto I) { s0[I] = 0.0; });
    s0[T::SDIM] = double(aux);
    static_for<0, T::ND>([&](auto Di) {
      constexpr int d = Di;
      constexpr int typ = T::DEV_TYPE[d];
      if constexpr (typ == DEV_LOAD) {
        s0[d] = io.series[T::DEV_SLOT[d] * io.period + aux];
      } else if constexpr (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) {
        constexpr int g = T::DEV_SLOT[d];
        constexpr int u = g;  // uniform index
        cptr_t sd = C + L::SETDEV + SD_SIZE * T::DEV_SET[d];
        uint32_t q[4];
        Philox::generate(io.rng_seed, io.env_offset + uint64_t(e), epoch, 1u + u / 2, q);
        const double uu = Philox::u01(q[2 * (u % 2)], q[2 * (u % 2) + 1]);
        const double pm = io.series[(T::NLOAD + g) * io.period + aux];
        s0[d] = pm;
        s0[2 * T::ND + T::NDES + g] = pm;
        s0[T::ND + d] = sd[SD_QMIN] + (sd[SD_QMAX] - sd[SD_QMIN]) * uu;  // p.u. range, MVAr slot (sic)
      } else if constexpr (typ == DEV_STORAGE) {
        constexpr int u = T::NGEN + T::DEV_SLOT[d];
        cptr_t sd = C + L::SETDEV + SD_SIZE * T::DEV_SET[d];
        uint32_t q[4];
        Philox::generate(io.rng_seed, io.env_offset + uint64_t(e), epoch, 1u + u / 2, q);
        const double uu = Philox::u01(q[2 * (u % 2)], q[2 * (u % 2) + 1]);
        s0[2 * T::ND + T::DEV_SLOT[d]] = sd[SD_SOC_MIN] + (sd[SD_SOC_MAX] - sd[SD_SOC_MIN]) * uu;  // sic
      }
    });
    inputs_from_init_state<T>(C, s0, w, P_load, P_pot, P_set, Q_set);
  } else {
    // 1. exogenous variables (next_vars, anm6_easy.py:54-65 in series mode)
    if (series) {
      const double a = state[T::SDIM];
      aux = int(fmod(a + 1.0, double(io.period)));
      static_for<0, T::NLOAD>([&](auto I) { P_load[I] = io.series[I * io.period + aux]; });
      static_for<0, T::NGEN>([&](auto I) { P_pot[I] = io.series[(T::NLOAD + I) * io.period + aux]; });
    } else {
      static_for<0, T::NLOAD>([&](auto I) { P_load[I] = io.exo[e * D::NEXO + I]; });
      static_for<0, T::NGEN>([&](auto I) { P_pot[I] = io.exo[e * D::NEXO + T::NLOAD + I]; });
    }
    // 2. action layout [P_gen.., Q_gen.., P_des.., Q_des..] by ascending device id (anm_env.py:393-410)
    const double* a = io.action + e * D::ADIM;
    static_for<0, T::ND>([&](auto Di) {
      constexpr int d = Di;
      constexpr int typ = T::DEV_TYPE[d];
      if constexpr (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) {
        P_set[T::DEV_SET[d]] = a[T::DEV_SLOT[d]];
        Q_set[T::DEV_SET[d]] = a[T::NGEN + T::DEV_SLOT[d]];
      } else if constexpr (typ == DEV_STORAGE) {
        P_set[T::DEV_SET[d]] = a[2 * T::NGEN + T::DEV_SLOT[d]];
        Q_set[T::DEV_SET[d]] = a[2 * T::NGEN + T::NDES + T::DEV_SLOT[d]];
      }
    });
    static_for<0, T::NDES>([&](auto I) { w.soc[I] = io.soc[e * T::NDES + I]; });
  }

  // 3. one simulator transition, shared by the step and the autoreset path
  ANM_PHASE(1);
  transition<T, JT>(C, w, P_load, P_pot, P_set, Q_set, so.tol, so.max_iter);
  if (io.nr_iters) io.nr_iters[e] = w.n_iter;

  if (resetting) {
    finish_reset<T>(C, w, s0, 1, io.soc + e * T::NDES, state, obs);
    io.reset_count[e] += 1;
    io.terminated[e] = w.converged ? 0 : 1;  // not converged: try another draw at the next call
    if (io.timestep) io.timestep[e] = 0;
    io.reward[e] = 0.0;
    io.e_loss[e] = 0.0;
    io.penalty[e] = 0.0;
    if (io.full) write_full_state<T>(w, io.full + e * FullState<T>::SIZE);
    return;
  }

  static_for<0, T::NDES>([&](auto I) { io.soc[e * T::NDES + I] = w.soc[I]; });
  const bool term = !w.converged;
  io.terminated[e] = term ? 1 : 0;
  const double c1 = C[L::SCALARS + SC_C1], c2 = C[L::SCALARS + SC_C2];
  if (!term) {
    // reward clipping (anm_env.py:423-427)
    const double sg = (w.e_loss > 0.0) ? 1.0 : ((w.e_loss < 0.0) ? -1.0 : 0.0);
    const double el = sg * fmin(fabs(w.e_loss), c1);
    const double pn = fmin(fmax(w.penalty, 0.0), c2);
    io.e_loss[e] = el;
    io.penalty[e] = pn;
    io.reward[e] = -(el + pn);
    write_state_obs<T>(C, w, state, obs);
    if (series) {
      state[T::SDIM] = double(aux);
      obs[T::SDIM] = fmin(fmax(double(aux), C[L::OBS_LO + T::SDIM]), C[L::OBS_HI + T::SDIM]);
    } else {
      for (int k = 0; k < io.K; ++k) {
        const double v = io.aux_next[e * io.K + k];
        state[T::SDIM + k] = v;
        obs[T::SDIM + k] = fmin(fmax(v, C[L::OBS_LO + T::SDIM + k]), C[L::OBS_HI + T::SDIM + k]);
      }
    }
  } else {
    io.reward[e] = C[L::SCALARS + SC_RTERM];  // -c2 / (1 - gamma)
    io.e_loss[e] = c1;
    io.penalty[e] = c2;
    for (int k = 0; k < S; ++k) {
      state[k] = 0.0;
      obs[k] = 0.0;
    }
  }
  if (io.timestep) io.timestep[e] += 1;
  if (io.full) write_full_state<T>(w, io.full + e * FullState<T>::SIZE);
  ANM_PHASE(6);
}

}  // namespace anm
