// hostsim.cpp -- TEST DOUBLE, not a product path.
//
// Compiles the very same per-environment templates the gfx950 kernels instantiate
// (gym_anm_amd/csrc/anm_env_ops.hpp) with g++ and exposes them behind the same C ABI
// (include/anm_mi355x.h) on HOST pointers, one environment after the other.  The CPU-only test
// tier (-m "not gpu") uses it to check the kernel logic, the constant packing and the Python host
// layer against the oracle without a GPU.  The product package never loads this library: its
// loader (gym_anm_amd/_lib.py) only accepts the hipcc-built libanm_<topology>.so and raises when
// that is missing.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include ANM_TOPO_HEADER
#include "../../gym_anm_amd/csrc/anm_env_ops.hpp"
#include "../../gym_anm_amd/csrc/anm_pack.hpp"
#include "../../gym_anm_amd/csrc/anm_mpc.hpp"

#include <pthread.h>

#include <thread>

using namespace anm;

static thread_local std::string g_err;
static int fail(const std::string& s) { g_err = s; return -1; }

struct anm_model {
  std::vector<double> c;
  std::vector<cplx> ybus;
  std::vector<double> series;
  int period = 0, K = 0;
  bool env_set = false;
  double* nr_diff = nullptr;
  const double* nr_start = nullptr;
};

struct anm_mpc {
  int N = 0;
  std::vector<double> tab;
  double theta_bound = 0.0;
  double base_mva = 1.0;
  int nl = 0, ng = 0, ns = 0, nd = 0;
  bool angle_rows = true;
};

// the lanes of one group as host threads
struct HostGroupShared {
  int n;
  std::vector<double> slot;
  pthread_barrier_t bar;
  explicit HostGroupShared(int n_) : n(n_), slot(n_, 0.0) { pthread_barrier_init(&bar, nullptr, unsigned(n_)); }
  ~HostGroupShared() { pthread_barrier_destroy(&bar); }
};
struct HostGroup {
  HostGroupShared* g;
  int st;
  int stage() const { return st; }
  template <class F>
  double exch(double v, F&& f) const {
    g->slot[st] = v;
    pthread_barrier_wait(&g->bar);
    const double r = f(g->slot);
    pthread_barrier_wait(&g->bar);
    return r;
  }
  double up(double v) const { return exch(v, [&](const std::vector<double>& s) { return st > 0 ? s[st - 1] : 0.0; }); }
  double down(double v) const { return exch(v, [&](const std::vector<double>& s) { return st + 1 < g->n ? s[st + 1] : 0.0; }); }
  double sum(double v) const { return exch(v, [&](const std::vector<double>& s) { double a = 0; for (double q : s) a += q; return a; }); }
  double max(double v) const { return exch(v, [&](const std::vector<double>& s) { double a = s[0]; for (double q : s) a = std::fmax(a, q); return a; }); }
  double min(double v) const { return exch(v, [&](const std::vector<double>& s) { double a = s[0]; for (double q : s) a = std::fmin(a, q); return a; }); }
  double scan(double v) const { return exch(v, [&](const std::vector<double>& s) { double a = 0; for (int k = 0; k <= st; ++k) a += s[k]; return a; }); }
  bool all_done(bool d) const { return exch(d ? 1.0 : 0.0, [&](const std::vector<double>& s) { double a = 1; for (double q : s) a = std::fmin(a, q); return a; }) > 0.5; }
};

static SolverOpts solver(const anm_solver_opts* o, int& prec) {
  SolverOpts s{1e-5, 100};
  prec = ANM_SOLVE_F64;
  if (o) { s.tol = o->tol; s.max_iter = o->max_iter; prec = o->precision; }
  return s;
}

extern "C" {
const char* anm_last_error(void) { return g_err.c_str(); }
const char* anm_topology_name(void) { return Topo::NAME; }
const char* anm_topology_signature(void) { return Topo::SIGNATURE; }
int anm_device_count(void) { return 0; }

int anm_model_create(const anm_network_desc* d, anm_model** out) {
  anm_model* m = new anm_model();
  std::string err;
  if (!pack_constants<Topo>(*d, m->c, m->ybus, err)) { delete m; g_err = err; return -3; }
  *out = m;
  return 0;
}
void anm_model_destroy(anm_model* m) { delete m; }
int anm_model_dims(const anm_model*, anm_dims* o) {
  o->n_bus = Topo::NB; o->n_dev = Topo::ND; o->n_branch = Topo::NBR; o->n_load = Topo::NLOAD;
  o->n_gen = Topo::NGEN; o->n_des = Topo::NDES; o->action_dim = Dims<Topo>::ADIM;
  o->state_base_dim = Topo::SDIM; o->full_dim = FullState<Topo>::SIZE; o->const_doubles = Layout<Topo>::TOTAL;
  return 0;
}
int anm_model_full_layout(const anm_model*, anm_full_layout* o) {
  typedef FullState<Topo> F;
  o->bus_p = F::BUS_P; o->bus_q = F::BUS_Q; o->bus_v_magn = F::BUS_VM; o->bus_v_ang = F::BUS_VA;
  o->bus_i_magn = F::BUS_IM; o->bus_i_ang = F::BUS_IA; o->dev_p = F::DEV_P; o->dev_q = F::DEV_Q;
  o->des_soc = F::DES_SOC; o->gen_p_max = F::GEN_PMAX; o->branch_p = F::BR_P; o->branch_q = F::BR_Q;
  o->branch_s = F::BR_S; o->branch_i_magn = F::BR_IM; o->branch_i_ang = F::BR_IA; o->size = F::SIZE;
  return 0;
}
int anm_model_set_env(anm_model* m, const anm_env_config* cfg) {
  std::string err;
  if (!pack_env<Topo>(*cfg, m->c, err)) { g_err = err; return -3; }
  m->K = cfg->K;
  m->series.clear();
  m->period = 0;
  if (cfg->series && cfg->period > 0) {
    if (cfg->K != 1) return fail("series mode needs exactly K = 1 auxiliary variable (the time index)");
    m->series.assign(cfg->series, cfg->series + size_t(Dims<Topo>::NEXO) * cfg->period);
    m->period = cfg->period;
  }
  m->env_set = true;
  return 0;
}
int anm_step_ws_record_doubles(void) { return 0; }
int anm_model_set_impl(anm_model*, int32_t impl) {
  if (impl != ANM_IMPL_THREAD) return fail("hostsim: only the thread-per-environment templates have a host build");
  return 0;
}
int anm_model_get_impl(const anm_model*) { return ANM_IMPL_THREAD; }
int anm_model_lanes_per_env(const anm_model*) { return 1; }
int anm_model_get_ybus(const anm_model* m, double* y) {
  for (size_t k = 0; k < m->ybus.size(); ++k) { y[2 * k] = m->ybus[k].real(); y[2 * k + 1] = m->ybus[k].imag(); }
  return 0;
}

int anm_transition_f64(anm_model* m, int64_t n, const double* p_load, const double* p_pot, const double* p_set,
                       const double* q_set, double* soc, double* full, double* reward, double* e_loss,
                       double* penalty, uint8_t* converged, int32_t* nr_iters, const anm_solver_opts* opts, void*) {
  TransitionIO io{p_load, p_pot, p_set, q_set, soc, full, reward, e_loss, penalty, converged, nr_iters, m->nr_diff, m->nr_start};
  int prec;
  SolverOpts so = solver(opts, prec);
  for (int64_t e = 0; e < n; ++e) {
    if (prec == ANM_SOLVE_F32) op_transition<Topo, float, true>(m->c.data(), io, so, e);
    else op_transition<Topo, double, true>(m->c.data(), io, so, e);
  }
  return 0;
}

int anm_reset_f64(anm_model* m, int64_t n, const double* init_state, const uint8_t* mask, uint64_t rng_seed,
                  uint64_t env_offset, int32_t* reset_count, double* soc, double* state,
                  double* obs, uint8_t* converged, uint8_t* terminated, int32_t* timestep, int32_t* nr_iters,
                  double* full, int32_t* aux_index, const anm_solver_opts* opts, void*) {
  if (!m->env_set) return fail("anm_reset_f64: call anm_model_set_env first");
  EnvIO io{};
  if (!init_state && (m->period <= 0 || m->K != 1 || !reset_count))
    return fail("anm_reset_f64: drawing initial states on the device needs a series-mode model and reset_count");
  io.K = m->K; io.init_state = init_state; io.mask = mask; io.soc = soc; io.state = state; io.obs = obs;
  io.series = m->series.data(); io.period = m->period; io.rng_seed = rng_seed; io.env_offset = env_offset;
  io.reset_count = reset_count;
  io.converged = converged; io.terminated = terminated; io.timestep = timestep; io.nr_iters = nr_iters; io.full = full; io.aux_index = aux_index;
  io.nr_diff = m->nr_diff;
  int prec;
  SolverOpts so = solver(opts, prec);
  for (int64_t e = 0; e < n; ++e) {
    if (prec == ANM_SOLVE_F32) op_reset<Topo, float>(m->c.data(), io, so, e);
    else op_reset<Topo, double>(m->c.data(), io, so, e);
  }
  return 0;
}

int anm_sample_init_state_f64(anm_model* m, int64_t n, uint64_t rng_seed, uint64_t env_offset, const int32_t* reset_count,
                              double* init_state, uint32_t* raw, void*) {
  if (!m->env_set || m->period <= 0 || m->K != 1) return fail("anm_sample_init_state_f64: the model needs a series-mode task");
  EnvIO io{};
  io.K = 1; io.series = m->series.data(); io.period = m->period; io.rng_seed = rng_seed; io.env_offset = env_offset;
  const int n_blocks = 1 + (Topo::NGEN + Topo::NDES + 1) / 2;
  for (int64_t e = 0; e < n; ++e) {
    const uint32_t epoch = reset_count ? uint32_t(reset_count[e]) : 0u;
    double s0[Topo::SDIM + 1];
    sample_series_init_state<Topo>(m->c.data(), io, e, epoch, s0);
    for (int k = 0; k <= Topo::SDIM; ++k) init_state[e * (Topo::SDIM + 1) + k] = s0[k];
    if (raw)
      for (int b = 0; b < n_blocks; ++b) {
        uint32_t q[4];
        Philox::generate(rng_seed, env_offset + uint64_t(e), epoch, uint32_t(b), q);
        for (int k = 0; k < 4; ++k) raw[(e * n_blocks + b) * 4 + k] = q[k];
      }
  }
  return 0;
}

int anm_step_f64(anm_model* m, int64_t n, const double* action, const double* exo, const double* aux_next,
                 double* soc, double* state, uint8_t* terminated, int32_t* timestep, double* obs, double* reward,
                 double* e_loss, double* penalty, int32_t* nr_iters, double* full, int32_t autoreset,
                 uint64_t rng_seed, uint64_t env_offset, int32_t* reset_count, int32_t* aux_index, const anm_step_ws*,
                 const anm_solver_opts* opts, void*) {
  if (!m->env_set) return fail("anm_step_f64: call anm_model_set_env first");
  const bool series = exo == nullptr;
  if (series && m->period <= 0) return fail("anm_step_f64: no exo given and the model has no series (set_env)");
  if (!series && m->K > 0 && !aux_next) return fail("anm_step_f64: exo given without aux_next");
  if (autoreset && (!series || !reset_count)) return fail("anm_step_f64: autoreset needs series mode and reset_count");
  EnvIO io{};
  io.K = m->K; io.action = action; io.exo = exo; io.aux_next = aux_next; io.series = m->series.data();
  io.period = m->period; io.soc = soc; io.state = state; io.terminated = terminated; io.timestep = timestep;
  io.obs = obs; io.reward = reward; io.e_loss = e_loss; io.penalty = penalty; io.nr_iters = nr_iters; io.full = full;
  io.autoreset = autoreset; io.rng_seed = rng_seed;
  io.env_offset = env_offset; io.reset_count = reset_count;
  int prec;
  SolverOpts so = solver(opts, prec);
  for (int64_t e = 0; e < n; ++e) {
    if (prec == ANM_SOLVE_F32) op_step<Topo, float>(m->c.data(), io, so, e);
    else op_step<Topo, double>(m->c.data(), io, so, e);
  }
  return 0;
}

int anm_time_step_launches(anm_model*, int64_t, const double*, double*, double*, uint8_t*, int32_t*, double*,
                           double*, double*, double*, int32_t, uint64_t, uint64_t, int32_t*, int32_t*, anm_step_ws*,
                           const anm_solver_opts*, void*,
                           int32_t, float*) {
  return fail("hostsim: no device timing");
}

int anm_model_set_classes(anm_model*, int32_t n_classes, const anm_network_desc* const*) {
  return n_classes > 1 ? fail("the host test double has no parameter classes") : 0;
}
int anm_model_set_class_obs_bounds(anm_model*, int32_t cls, const double*, const double*) {
  return cls > 0 ? fail("the host test double has no parameter classes") : 0;
}
int anm_model_bind_env_classes(anm_model*, const int32_t* env_class, int64_t) {
  return env_class ? fail("the host test double has no parameter classes") : 0;
}
// ---- MPC DC-OPF: the solver of gym_anm_amd/csrc/anm_mpc.hpp with one HOST THREAD per lane (= stage); what the
// wavefront shuffles do on the GPU goes through an exchange array between two barriers
typedef mpc::NoTheta<Topo> TopoNoTheta;
int anm_mpc_create(const anm_network_desc* desc, double gamma, double safety_margin, int32_t planning_steps, anm_mpc** out) {
  if constexpr (!mpc::Sz<TopoNoTheta>::FITS) {
    return fail("anm_mpc_create: this network has too many rows per stage for the register-resident MPC kernel");
  } else {
    std::string err;
    if (!check_topology<Topo>(*desc, err)) { g_err = err; return -3; }
    anm_mpc* m = new anm_mpc();
    if (!mpc::build_tables<Topo>(*desc, gamma, safety_margin, planning_steps, m->tab, err, &m->theta_bound)) { delete m; g_err = err; return -3; }
    m->N = planning_steps;
    m->base_mva = desc->base_mva;
    m->nl = Topo::NLOAD; m->ng = Topo::NGEN; m->ns = Topo::NDES; m->nd = Topo::ND;
    m->angle_rows = m->theta_bound >= 0.98 * 3.14159265358979323846;   // (as anm_capi.hip)
    if (m->angle_rows && !mpc::Sz<Topo>::FITS) { delete m; return fail("anm_mpc_create: too many rows per stage with the angle rows"); }
    *out = m;
    return 0;
  }
}
void anm_mpc_destroy(anm_mpc* m) { delete m; }
int anm_mpc_dims_of(const anm_mpc* m, anm_mpc_dims* o) {
  typedef mpc::Sz<Topo> S;
  o->planning_steps = m->N; o->n_load = S::NL; o->n_gen = S::NG; o->n_des = S::NS; o->n_branch = S::NBR;
  o->n_ctrl = S::NC; o->n_stage_vars = S::NV; o->table_doubles = S::T_TOTAL;
  o->n_stage_rows = m->angle_rows ? S::NR : mpc::Sz<TopoNoTheta>::NR;
  o->angle_rows = m->angle_rows ? 1 : 0;
  o->angle_bound = m->theta_bound;
  return 0;
}
int anm_mpc_get_tables(const anm_mpc* m, double* out) {
  std::memcpy(out, m->tab.data(), m->tab.size() * sizeof(double));
  return 0;
}
extern "C++" template <class TT>
void host_mpc_solve(anm_mpc* m, const mpc::IO& io, const mpc::Opts& o, int64_t num_envs) {
  if constexpr (mpc::Sz<TT>::FITS) {
    const int N = m->N;
    HostGroupShared sh(N);
    std::vector<std::thread> th;
    for (int st = 0; st < N; ++st)
      th.emplace_back([&, st]() {
        HostGroup x{&sh, st};
        for (int64_t e = 0; e < num_envs; ++e) mpc::solve<TT>(m->tab.data(), io, o, e, true, N, x);
      });
    for (auto& t : th) t.join();
  }
}
static int host_mpc_run(anm_mpc* m, int64_t num_envs, mpc::IO io, const anm_mpc_opts* opts) {
  mpc::Opts o{1e-11, 40};
  if (opts) {
    if (opts->tol > 0.0) o.tol = opts->tol;
    if (opts->max_iter > 0) o.max_iter = opts->max_iter;
    io.trace = opts->trace;
  }
  bool full = m->angle_rows;
  if (opts && opts->angle_rows == 1) full = true;
  if (opts && opts->angle_rows == 2) full = false;
  if (full && !mpc::Sz<Topo>::FITS) return fail("anm_mpc_solve_f64: too many rows per stage with the angle rows");
  if (full) host_mpc_solve<Topo>(m, io, o, num_envs);
  else host_mpc_solve<TopoNoTheta>(m, io, o, num_envs);
  return 0;
}
int anm_mpc_solve_f64(anm_mpc* m, int64_t num_envs, const double* p_load_forecast, const double* p_gen_forecast,
                      const double* soc, double* u0, double* objective, int32_t* iters, double* info, double* solution,
                      const anm_mpc_opts* opts, void*) {
  mpc::IO io{p_load_forecast, p_gen_forecast, soc, u0, objective, iters, info, solution, nullptr, mpc::Act{}, Topo::NLOAD, Topo::NGEN};
  return host_mpc_run(m, num_envs, io, opts);
}
int anm_mpc_act_f64(anm_mpc* m, int64_t num_envs, int32_t forecast, const double* state, const double* state_alt,
                    const uint8_t* state_same, int32_t state_dim, const int32_t* aux_index, const double* series,
                    int32_t period, const double* soc, const double* act_low, const double* act_high, double* action,
                    double* u0, double* objective, int32_t* iters, double* info, const anm_mpc_opts* opts, void*) {
  if (forecast != ANM_MPC_FORECAST_CONSTANT && forecast != ANM_MPC_FORECAST_PERFECT) return fail("anm_mpc_act_f64: unknown forecast");
  if (forecast == ANM_MPC_FORECAST_PERFECT && (!series || period <= 0)) return fail("anm_mpc_act_f64: a perfect forecast needs the task's periodic tables");
  mpc::Act a{forecast, state, state_alt, state_same, state_dim, aux_index, series, period, m->base_mva, action, act_low, act_high, {0}, 0};
  mpc::IO io{nullptr, nullptr, soc, u0, objective, iters, info, nullptr, nullptr, a, Topo::NLOAD, Topo::NGEN};
  return host_mpc_run(m, num_envs, io, opts);
}
int anm_model_bind_view(anm_model*, const anm_batch_view* v) { return v ? fail("the host test double has no lane-group kernels") : 0; }
int anm_model_bind_nr_diff(anm_model* m, double* p) { m->nr_diff = p; return 0; }
int anm_model_bind_nr_start(anm_model* m, const double* p) { m->nr_start = p; return 0; }
int anm_model_bind_state_same(anm_model*, uint8_t* p) { return p ? fail("the host test double writes every state row") : 0; }
int anm_model_obs_fusable(const anm_model*) { return 0; }  // the test double has no fused gather
int anm_model_set_obs(anm_model*, int32_t n_obs, const int32_t*, const double*, const double*, const double*) {
  return n_obs > 0 ? fail("the host test double gathers with anm_gather_obs_f64") : 0;
}

// TEST HOOK (not part of the C ABI): |z| and arg z as the electrical-state dump forms them (csrc/anm_device.hpp: dump_abs,
// dump_arg; on the host the division inside is the exact one, on the GPU a reciprocal + Newton sequence)
void anm_hostsim_dump_abs_arg(int64_t n, const double* x, const double* y, double* mag, double* ang) {
  for (int64_t i = 0; i < n; ++i) { mag[i] = anm::dump_abs(x[i], y[i]); ang[i] = anm::dump_arg(y[i], x[i]); }
}

int anm_gather_obs_f64(int64_t n, int32_t full_dim, const double* full, int32_t state_dim, int32_t K, const double* state,
                       const uint8_t* terminated, int32_t n_obs, const int32_t* index, const double* scale,
                       const double* low, const double* high, double* obs, void*) {
  for (int64_t e = 0; e < n; ++e)
    for (int k = 0; k < n_obs; ++k) {
      const int idx = index[k];
      const double src = (idx < full_dim) ? full[e * full_dim + idx] : state[e * state_dim + (state_dim - K) + (idx - full_dim)];
      const double v = std::fmin(std::fmax(src * scale[k], low[k]), high[k]);
      obs[e * n_obs + k] = (terminated && terminated[e]) ? 0.0 : v;
    }
  return 0;
}
}
