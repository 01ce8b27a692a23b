// anm_capi.hip -- gfx950 kernels + the C ABI of include/anm_mi355x.h for ONE network topology
// (the descriptor header is injected with -DANM_TOPO_HEADER=...; see gym_anm_amd/codegen.py).
//
// Launch shapes: 64-thread workgroups = one wavefront each (the general family: up to 512 lanes per environment above 65
// buses), so that a diverging Newton-Raphson solve only ever holds back the other environments of its own wavefront and
// workgroups spread round-robin over the 8 XCDs; no inter-workgroup traffic.  Thread family: one thread per environment,
// LDS for the coalesced row transposes and the in-wave hand-over slots; lane-group families: one environment per group of
// lanes, hand-overs and Jacobian blocks in LDS.  HBM traffic per environment step is the action row in and the obs /
// reward rows out (~250 B for ANM6Easy); the network constants are wave-uniform scalar loads that stay in the scalar cache
// (thread family) or per-lane tables (lane groups).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include ANM_TOPO_HEADER
#include "anm_env_ops.hpp"
#include "anm_pack.hpp"
#include "anm_radial.hpp"
#include "anm_mesh.hpp"
#include "anm_mpc.hpp"

using namespace anm;

#include "anm_mpc_capi_types.inc"

namespace {

constexpr int BLOCK = 64;

thread_local std::string g_err;

int fail(const char* what) {
  g_err = what;
  return -1;
}
int fail_hip(hipError_t e, const char* where) {
  g_err = std::string(where) + ": " + hipGetErrorString(e);
  return -2;
}

// (in-wave hand-over point; end of round 6, with the lane-group trip at half a thread-mode trip: 5 / 6 / 7 -> 73.7 / 74.3 / 74.8 us
// per step at 65 536 environments, four rounds, profiles/r06_z_handoff.txt.  6 stays: it is also where the two-launch step of the
// larger batches hands over, and the two step modes are bit-identical under the automatic policy only at the same point --
// a batch's results must not depend on its size.)
#ifndef ANM_HANDOFF_DEFAULT
#define ANM_HANDOFF_DEFAULT 6
#endif
#ifndef ANM_MID_CAP_DEFAULT
#define ANM_MID_CAP_DEFAULT 12  // where the first straggler launch leaves the solves still running to the second
#endif
#ifndef ANM_ROWS_WAVES
#define ANM_ROWS_WAVES 1  // min. waves per SIMD the step kernel is compiled for (register budget 512 / waves)
#endif

// VIEW: the launch serves the environments a batch view names (anm_model_bind_view); without one the view is a compile-time
// identity and costs the kernels nothing
// START: the launch honours anm_model_bind_nr_start (an instantiation of its own, always with VIEW: see op_transition)
template <class JT, bool VIEW = false, bool START = false>
__global__ __launch_bounds__(BLOCK) void k_transition(cptr_t C0, TransitionIO io, SolverOpts so, int64_t n, ClassSel cs, View v) {
#ifndef ANM_DEV_LANE_GROUPS_ONLY   // (tuning builds of the lane-group kernels alone: minutes less to compile)
  __shared__ double lds[Topo::TREE != 0 ? group::Shape<Topo>::NG * group::Slot<Topo>::SIZE : 1];
  const int64_t e0 = int64_t(blockIdx.x) * BLOCK + threadIdx.x;
  const bool valid = e0 < n;
  const int64_t e = valid ? e0 : n - 1;
  const cptr_t C = class_constants(C0, cs, int64_t(blockIdx.x) * BLOCK);
  if constexpr (VIEW) op_transition<Topo, JT, START>(C, io, so, e, v, valid, lds);
  else op_transition<Topo, JT>(C, io, so, e, View{}, valid, lds);
#endif
}

template <class JT, bool VIEW = false>
__global__ __launch_bounds__(BLOCK) void k_reset(cptr_t C0, EnvIO io, SolverOpts so, int64_t n, ClassSel cs, View v) {
#ifndef ANM_DEV_LANE_GROUPS_ONLY   // (tuning builds of the lane-group kernels alone: minutes less to compile)
  __shared__ double lds[Topo::TREE != 0 ? group::Shape<Topo>::NG * group::Slot<Topo>::SIZE : 1];
  const int64_t e0 = int64_t(blockIdx.x) * BLOCK + threadIdx.x;
  const bool valid = e0 < n;
  const int64_t e = valid ? e0 : n - 1;
  const cptr_t C = class_constants(C0, cs, int64_t(blockIdx.x) * BLOCK);
  if constexpr (VIEW) op_reset<Topo, JT>(C, io, so, e, v, valid, lds);
  else op_reset<Topo, JT>(C, io, so, e, View{}, valid, lds);
#endif
}

// the step of the environments a batch view names (thread-per-environment family): see op_step_view.
// SW: wavefronts per SIMD the registers are budgeted for -- 1: 288 registers; 2: 256 + 80 bytes of scratch, the faster
// kernel once the batch gives every SIMD more than two wavefronts (two ANM6 models, 524 288 environments: 385 -> 315 us;
// 131 072: 219 -> 223; 16 384, where the step waits for one diverging solve: 124 -> 139.  profiles/r05_k_view_step.txt)
template <class JT, int SW>
__global__ __launch_bounds__(BLOCK, SW) void k_step_view(cptr_t C, EnvIO io, SolverOpts so, int64_t n, View v) {
#ifndef ANM_DEV_LANE_GROUPS_ONLY   // (tuning builds of the lane-group kernels alone: minutes less to compile)
  __shared__ double lds[Topo::TREE != 0 ? group::Shape<Topo>::NG * group::Slot<Topo>::SIZE : 1];
  op_step_view<Topo, JT>(C, io, so, n, v, lds);
#endif
}

// fast path of the step (series mode, K = 1, "state" observation): see op_step_rows
template <class JT, bool FULL>
__global__ __launch_bounds__(BLOCK, ANM_ROWS_WAVES) void k_step_rows(cptr_t C0, EnvIO io, SolverOpts so, int64_t n, ClassSel cs) {
#ifndef ANM_DEV_LANE_GROUPS_ONLY   // (tuning builds of the lane-group kernels alone: minutes less to compile)
  __shared__ double lds[64 * (Topo::SDIM + 2)];
  const cptr_t C = class_constants(C0, cs, int64_t(blockIdx.x) * BLOCK);
  op_step_rows<Topo, JT, FULL>(C, io, so, n, lds);
#endif
}

// general step (host next_vars, K != 1, list-form observations, `full` dump): see op_step_general
template <class JT>
__global__ __launch_bounds__(BLOCK) void k_step_general(cptr_t C0, EnvIO io, SolverOpts so, int64_t n, ClassSel cs) {
#ifndef ANM_DEV_LANE_GROUPS_ONLY   // (tuning builds of the lane-group kernels alone: minutes less to compile)
  extern __shared__ double lds_dyn[];
  const cptr_t C = class_constants(C0, cs, int64_t(blockIdx.x) * BLOCK);
  op_step_general<Topo, JT>(C, io, so, n, lds_dyn);
#endif
}

template <class JT, bool GROUPS>
__global__ __launch_bounds__(BLOCK) void k_step_stragglers(cptr_t C, EnvIO io, SolverOpts so, int level) {
#ifndef ANM_DEV_LANE_GROUPS_ONLY   // (tuning builds of the lane-group kernels alone: minutes less to compile)
  __shared__ double lds[GROUPS ? group::Shape<Topo>::NG * group::Slot<Topo>::SIZE : 1];
  op_step_stragglers<Topo, JT, GROUPS>(C, io, so, lds, level);
#endif
}

__global__ void k_step_scatter(EnvIO io) { op_step_scatter<Topo>(io); }

// obs[e, k] = clip(src(e, index[k]) * scale[k], low[k], high[k]); src is the `full` row for
// index < full_dim and the aux tail of the state row beyond it; terminated environments observe 0
// (anm_env.py:365-367, 442-446)
__global__ void k_gather_obs(int64_t n, int full_dim, const double* __restrict__ full, int state_dim, int K,
                             const double* __restrict__ state, const uint8_t* __restrict__ terminated, int n_obs,
                             const int32_t* __restrict__ index, const double* __restrict__ scale,
                             const double* __restrict__ low, const double* __restrict__ high,
                             double* __restrict__ obs) {
  const int64_t total = n * n_obs;
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += int64_t(gridDim.x) * blockDim.x) {
    const int64_t e = t / n_obs;
    const int k = int(t - e * n_obs);
    const int idx = index[k];
    const double src = (idx < full_dim) ? full[e * full_dim + idx] : state[e * state_dim + (state_dim - K) + (idx - full_dim)];
    const double v = fmin(fmax(src * scale[k], low[k]), high[k]);
    obs[t] = (terminated && terminated[e]) ? 0.0 : v;
  }
}

// ANM6Easy.init_state (anm6_easy.py:25-52) for any series-mode task, the draws alone: one thread per environment, table-driven
// (tab[d] = type, slot, q_min, q_max, soc_min, soc_max of device d), the very expressions of the samplers inside the reset /
// step kernels (sample_series_init_state, anm_radial.hpp, anm_mesh.hpp) -- tests/test_gpu_sampler.py holds them to it bit for
// bit.  raw (nullable): the Philox words behind the row, blocks 0 .. n_blocks - 1 of key (seed, env_offset + e, epoch).
__global__ void k_sample_init_state(int64_t n, int nd, int nload, int ngen, int ndes, const double* __restrict__ tab,
                                    const double* __restrict__ series, int period, uint64_t seed, uint64_t env_offset,
                                    const int32_t* __restrict__ reset_count, double* __restrict__ out, uint32_t* __restrict__ raw,
                                    int n_blocks) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int W = 2 * nd + ndes + ngen + 1;
  const uint32_t epoch = reset_count ? uint32_t(reset_count[e]) : 0u;
  uint32_t r[4];
  Philox::generate(seed, env_offset + uint64_t(e), epoch, 0u, r);
  const int aux = int((uint64_t(r[0]) * uint64_t(period)) >> 32);
  double* s0 = out + e * W;
  for (int k = 0; k < W; ++k) s0[k] = 0.0;
  s0[W - 1] = double(aux);
  for (int d = 0; d < nd; ++d) {
    const double* t = tab + 6 * d;
    const int typ = int(t[0]), slot = int(t[1]);
    if (typ == DEV_LOAD) {
      s0[d] = series[slot * period + aux];
    } else if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE || typ == DEV_STORAGE) {
      const bool des = typ == DEV_STORAGE;
      const int u = des ? ngen + slot : slot;
      uint32_t q[4];
      Philox::generate(seed, env_offset + uint64_t(e), epoch, 1u + u / 2, q);
      const double uu = Philox::u01(q[2 * (u % 2)], q[2 * (u % 2) + 1]);
      if (des) {
        s0[2 * nd + slot] = t[4] + (t[5] - t[4]) * uu;
      } else {
        const double pm = series[(nload + slot) * period + aux];
        s0[d] = pm;
        s0[2 * nd + ndes + slot] = pm;
        s0[nd + d] = t[2] + (t[3] - t[2]) * uu;
      }
    }
  }
  if (raw)
    for (int b = 0; b < n_blocks; ++b) {
      uint32_t q[4];
      Philox::generate(seed, env_offset + uint64_t(e), epoch, uint32_t(b), q);
      for (int k = 0; k < 4; ++k) raw[(e * n_blocks + b) * 4 + k] = q[k];
    }
}

}  // namespace

struct anm_model {
  int impl = ANM_IMPL_THREAD;   // which kernel family serves this model
  int impl_unbound = -1;        // the family to go back to when a per-environment class binding is lifted (-1: none pending)
  radial::View view{};          // anm_model_bind_view: the launches serve a sub-batch of a larger, padded batch
  bool has_view = false;
  int view_waves = 0;   // k_step_view variant: 0 = by batch size; 1 | 2 (ANM_VIEW_WAVES, read at anm_model_create: tuning, tests)
  bool tpe_ok = false;          // the network has the topology this library was compiled for
  bool radial_ok = false;       // the network is a tree that fits one wavefront
  radial::Plan plan;            // per-lane tables of the lane-group kernel
  bool mesh_ok = false;         // the general lane-group kernel can take the network
  mesh::Plan mplan;
  mesh::Launch mlaunch{2, 1, true, 0, 0};
  int* d_mi = nullptr;
  double* d_md = nullptr;
  std::vector<std::vector<double>> x_md;      // general lane-group tables of each extra class
  int* d_ri = nullptr;
  double* d_rd = nullptr;
  double* d_const = nullptr;    // device constant buffer (Layout<Topo>)
  double* d_series = nullptr;   // device exogenous series [NEXO][period]
  int period = 0;
  int K = 0;
  bool env_set = false;
  // list-form observation (anm_model_set_obs)
  int n_obs = 0;
  unsigned obs_need = 0;
  int obs_row_stride = 0, obs_aux_off = 0;   // compact row layout (only the classes the list reads)
  short obs_cls_off[FC_COUNT] = {0};
  int32_t* d_obs_index = nullptr;            // [2][n_obs]: compact-layout indices, identity-layout indices
  double* d_obs_tab = nullptr;               // [3][n_obs]: scale, low, high
  std::vector<double> h_const;
  // parameter classes 1.. (class 0 is the network the model was created from): same topology, other numbers
  std::vector<std::vector<double>> x_const;   // thread-per-environment constants of each extra class
  std::vector<std::vector<double>> x_hd;      // lane-group tables of each extra class
  const int32_t* d_env_class = nullptr;       // caller's device array [num_envs] (anm_model_bind_env_classes)
  bool class_per_env = false;                 // the classes do not come in aligned blocks of 64 environments
  uint8_t* d_state_same = nullptr;            // caller's device array [num_envs] (anm_model_bind_state_same)
  double* d_nr_diff = nullptr;                // caller's device array [num_envs] (anm_model_bind_nr_diff)
  const double* d_nr_start = nullptr;         // caller's device array [num_envs, 2 (n_bus - 1)] (anm_model_bind_nr_start)
  int32_t* d_zero = nullptr;                  // one zero: the class of every environment when no classes are bound
  double* d_samp = nullptr;                   // [n_dev][6] sampler table (anm_sample_init_state_f64)
  int s_nd = 0, s_nload = 0, s_ngen = 0, s_ndes = 0;
  std::vector<cplx> ybus;
};


namespace {

int upload_const(anm_model* m) {
  hipError_t e = hipSuccess;
  if (m->tpe_ok) {
    const size_t n = m->h_const.size();
    e = hipMemcpy(m->d_const, m->h_const.data(), n * sizeof(double), hipMemcpyHostToDevice);
    for (size_t k = 0; k < m->x_const.size() && e == hipSuccess; ++k)
      e = hipMemcpy(m->d_const + (k + 1) * n, m->x_const[k].data(), n * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpy(constants)");
  }
  if (m->radial_ok) {
    const size_t n = m->plan.hd.size();
    e = hipMemcpy(m->d_rd, m->plan.hd.data(), n * sizeof(double), hipMemcpyHostToDevice);
    for (size_t k = 0; k < m->x_hd.size() && e == hipSuccess; ++k)
      e = hipMemcpy(m->d_rd + (k + 1) * n, m->x_hd[k].data(), n * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpy(radial tables)");
  }
  if (m->mesh_ok) {
    const size_t n = m->mplan.hd.size();
    e = hipMemcpy(m->d_md, m->mplan.hd.data(), n * sizeof(double), hipMemcpyHostToDevice);
    for (size_t k = 0; k < m->x_md.size() && e == hipSuccess; ++k)
      e = hipMemcpy(m->d_md + (k + 1) * n, m->x_md[k].data(), n * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpy(mesh tables)");
  }
  return 0;
}

ClassSel class_sel(const anm_model* m, bool radial) {
  if (!m->d_env_class) return ClassSel{m->d_zero, 0, 0, 0};
  return ClassSel{m->d_env_class, radial ? int(m->plan.hd.size()) : int(m->h_const.size()), 1, m->class_per_env ? 1 : 0};
}

template <bool WG, int SW = 2, bool TL = !WG>
void launch_mesh_as(anm_model* m, int precision, unsigned grid, unsigned threads, size_t lds, hipStream_t s, const radial::IO& io,
                    SolverOpts so, int64_t n, const ClassSel& cs) {
  const mesh::Dims& d = m->mplan.d;
  if (cs.per_group) {
    if (precision == ANM_SOLVE_F32)
      hipLaunchKernelGGL((mesh::k_mesh<float, true, WG, SW, TL>), dim3(grid), dim3(threads), lds, s, d, m->d_mi, m->d_md, io, so, n, cs);
    else
      hipLaunchKernelGGL((mesh::k_mesh<double, true, WG, SW, TL>), dim3(grid), dim3(threads), lds, s, d, m->d_mi, m->d_md, io, so, n, cs);
  } else if (precision == ANM_SOLVE_F32)
    hipLaunchKernelGGL((mesh::k_mesh<float, false, WG, SW, TL>), dim3(grid), dim3(threads), lds, s, d, m->d_mi, m->d_md, io, so, n, cs);
  else
    hipLaunchKernelGGL((mesh::k_mesh<double, false, WG, SW, TL>), dim3(grid), dim3(threads), lds, s, d, m->d_mi, m->d_md, io, so, n, cs);
}

// the variants of k_mesh a launch may pick (mesh::Launch), as function pointers: for hipFuncSetAttribute
template <bool WG, int SW, bool TL>
void mesh_variants(std::vector<const void*>& out) {
  out.push_back((const void*)mesh::k_mesh<float, false, WG, SW, TL>);
  out.push_back((const void*)mesh::k_mesh<double, false, WG, SW, TL>);
  out.push_back((const void*)mesh::k_mesh<float, true, WG, SW, TL>);
  out.push_back((const void*)mesh::k_mesh<double, true, WG, SW, TL>);
}

int launch_mesh(anm_model* m, int precision, int64_t n, hipStream_t s, const radial::IO& io_in, SolverOpts so) {
  radial::IO io = io_in;
  io.v = m->view;
  const mesh::Dims& d = m->mplan.d;
  const mesh::Launch& L = m->mlaunch;   // (decided once, at anm_model_create)
  const int per_block = mesh::envs_per_block(d, L);
  const unsigned grid = unsigned((n + per_block - 1) / per_block), threads = unsigned(64 * L.waves);
  const ClassSel cs = m->d_env_class ? ClassSel{m->d_env_class, int(m->mplan.hd.size()), 1, m->class_per_env ? 1 : 0} : ClassSel{m->d_zero, 0, 0, 0};
  if (mesh::is_workgroup(d)) launch_mesh_as<true>(m, precision, grid, threads, L.lds, s, io, so, n, cs);
  else if (!L.tables_in_lds) launch_mesh_as<false, 2, false>(m, precision, grid, threads, L.lds, s, io, so, n, cs);
  else if (L.simd_waves == 3) launch_mesh_as<false, 3>(m, precision, grid, threads, L.lds, s, io, so, n, cs);
  else launch_mesh_as<false>(m, precision, grid, threads, L.lds, s, io, so, n, cs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "launch k_mesh");
  return 0;
}

// the lane-group families gather a list-form observation from one LDS row of the electrical state per environment
// (FS + KMAX doubles): the radial kernel in dynamic LDS next to its static 6 KB, the general kernel where its Jacobian
// blocks were
size_t radial_obs_lds_bytes(const anm_model* m) {
  return size_t(64 / m->plan.d.G) * size_t(m->plan.d.FS + radial::KMAX) * sizeof(double);
}

int launch_radial(anm_model* m, int precision, int64_t n, hipStream_t s, const radial::IO& io_in, SolverOpts so) {
  radial::IO io = io_in;
  io.v = m->view;
  const int per_wave = 64 / m->plan.d.G;
  const unsigned grid = unsigned((n + per_wave - 1) / per_wave);
  // the model is the compiled tree: the specialised Newton loop; else (generic mode) the table-driven one
  bool spec = false;
  if constexpr (Topo::TREE != 0) spec = m->tpe_ok && m->plan.d.G == Topo::GRP && !getenv("ANM_RADIAL_GENERIC");
  const ClassSel cs = class_sel(m, true);
  const bool f32 = precision == ANM_SOLVE_F32;
  const size_t obs_lds = (io.mode == 2 && io.e.n_obs > 0) ? radial_obs_lds_bytes(m) : 0;   // rows of a list-form observation
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(64), obs_lds, s, m->plan.d, m->d_ri, m->d_rd, io, so, n, cs); };
  if (spec) {
    if constexpr (Topo::TREE != 0) {
      if (cs.per_group) { if (f32) go(radial::k_radial<float, Topo, true>); else go(radial::k_radial<double, Topo, true>); }
      else { if (f32) go(radial::k_radial<float, Topo>); else go(radial::k_radial<double, Topo>); }
    }
  } else if (cs.per_group) {
    if (f32) go(radial::k_radial<float, void, true>); else go(radial::k_radial<double, void, true>);
  } else {
    if (f32) go(radial::k_radial<float, void>); else go(radial::k_radial<double, void>);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "launch k_radial");
  return 0;
}

SolverOpts solver(const anm_solver_opts* o, int& precision) {
  SolverOpts s{1e-5, 100, ANM_HANDOFF_AUTO};
  precision = ANM_SOLVE_F64;
  if (o) {
    s.tol = o->tol;
    s.max_iter = o->max_iter;
    precision = o->precision;
    s.handoff = o->handoff_after;
  }
  // default hand-over point: every converging solve seen so far needs <= 9 iterations at tol 1e-6, so after
  // ANM_HANDOFF_DEFAULT trips the lanes still iterating are (almost only) diverging solves
  if (s.handoff == ANM_HANDOFF_AUTO) s.handoff = Topo::TREE ? ANM_HANDOFF_DEFAULT : ANM_HANDOFF_NEVER;
  return s;
}

inline unsigned grid_for(int64_t n) { return unsigned((n + BLOCK - 1) / BLOCK); }

}  // namespace

extern "C" {

const char* anm_last_error(void) { return g_err.c_str(); }
const char* anm_topology_name(void) { return Topo::NAME; }
const char* anm_topology_signature(void) { return Topo::SIGNATURE; }

int anm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int anm_model_create(const anm_network_desc* desc, anm_model** out) {
  if (!desc || !out) return fail("anm_model_create: null argument");
  anm_model* m = new (std::nothrow) anm_model();
  if (!m) return fail("out of host memory");
  std::string err, err_topo;
  m->tpe_ok = pack_constants<Topo>(*desc, m->h_const, m->ybus, err_topo);
  if (m->tpe_ok) {
    // k_step_general's dynamic LDS: the electrical-state rows (at most 60 KB, GenLds::FULL_OK) plus the state rows can
    // exceed the default 64 KB per workgroup (an 8-bus tree with 9 devices: 68 KB).  The attribute is set here, once,
    // not in a launch path that may be under stream capture
    typedef GenLds<Topo> GL;
    constexpr size_t worst = (size_t(GL::FULL_OK ? (64 * GL::FSP > GL::B_DOUBLES ? 64 * GL::FSP : GL::B_DOUBLES) : GL::B_DOUBLES) +
                              size_t(64) * size_t(GL::SP)) * sizeof(double);
    if (worst > 64 * 1024) {
      hipError_t a1 = hipFuncSetAttribute((const void*)k_step_general<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipError_t a2 = hipFuncSetAttribute((const void*)k_step_general<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (a1 != hipSuccess || a2 != hipSuccess) {
        delete m;
        return fail_hip(a1 != hipSuccess ? a1 : a2, "k_step_general: LDS size attribute");
      }
    }
    hipError_t e = hipMalloc(&m->d_const, m->h_const.size() * sizeof(double));
    if (e != hipSuccess) {
      delete m;
      return fail_hip(e, "hipMalloc(constants)");
    }
  }
  if (radial::is_radial(*desc) && radial::build_plan(*desc, m->plan, err)) {
    hipError_t e1 = hipMalloc(&m->d_ri, m->plan.hi.size() * sizeof(int));
    hipError_t e2 = hipMalloc(&m->d_rd, m->plan.hd.size() * sizeof(double));
    if (e1 == hipSuccess && e2 == hipSuccess &&
        hipMemcpy(m->d_ri, m->plan.hi.data(), m->plan.hi.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess) {
      m->radial_ok = true;
      // default: the lane-group kernel once the thread-per-environment working set no longer fits
      // in registers (measured: a 30-bus feeder spills 4.7 KB/lane); ANM_IMPL=thread|radial overrides
      m->impl = (!m->tpe_ok || desc->n_bus > 12) ? ANM_IMPL_RADIAL : ANM_IMPL_THREAD;
      const char* ev = getenv("ANM_IMPL");
      if (ev && std::string(ev) == "thread" && m->tpe_ok) m->impl = ANM_IMPL_THREAD;
      if (ev && std::string(ev) == "radial") m->impl = ANM_IMPL_RADIAL;
    }
  }
  {  // the general lane-group kernel: any topology that fits a wavefront
    std::string err_mesh;
    if (mesh::build_plan(*desc, m->mplan, err_mesh)) {
      hipError_t e1 = hipMalloc(&m->d_mi, m->mplan.hi.size() * sizeof(int));
      hipError_t e2 = hipMalloc(&m->d_md, m->mplan.hd.size() * sizeof(double));
      if (e1 == hipSuccess && e2 == hipSuccess &&
          hipMemcpy(m->d_mi, m->mplan.hi.data(), m->mplan.hi.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess) {
        m->mesh_ok = true;
        m->mlaunch = mesh::launch_of(m->mplan.d);
        if (m->mlaunch.lds > 64 * 1024) {
          // above the default per-workgroup limit: ask for the compute unit's whole LDS (once, here: an
          // attribute call has no place in a launch path that may be under stream capture)
          std::vector<const void*> fns;
          mesh_variants<false, 2, true>(fns); mesh_variants<false, 3, true>(fns); mesh_variants<false, 2, false>(fns); mesh_variants<true, 2, false>(fns);
          for (const void* fn : fns)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) m->mesh_ok = false;
        }
      }
      if (m->mesh_ok) {
        // default for what neither of the other families serves well: not a tree, and either not the compiled
        // topology or too large for one thread's registers
        if (!m->radial_ok && (!m->tpe_ok || desc->n_bus > 12)) m->impl = ANM_IMPL_MESH;
        const char* ev = getenv("ANM_IMPL");
        if (ev && std::string(ev) == "mesh") m->impl = ANM_IMPL_MESH;
      }
    }
  }
  if (const char* ev = getenv("ANM_VIEW_WAVES")) {
    const int v = atoi(ev);
    if (v == 1 || v == 2) m->view_waves = v;
  }
  if (!m->tpe_ok && !m->radial_ok && !m->mesh_ok) {
    // neither the compiled topology nor a network the generic lane-group kernels can take
    if (m->d_ri) hipFree(m->d_ri);
    if (m->d_rd) hipFree(m->d_rd);
    if (m->d_mi) hipFree(m->d_mi);
    if (m->d_md) hipFree(m->d_md);
    delete m;
    g_err = err_topo;
    return -3;
  }
  if (!m->tpe_ok) {  // generic model: dense Y_bus for diagnostics
    const int NB = desc->n_bus;
    m->ybus.assign(size_t(NB) * NB, cplx(0, 0));
    for (int b = 0; b < desc->n_branch; ++b) {
      const int f = desc->br_from[b], t = desc->br_to[b];
      const cplx ys(desc->br_series[2 * b], desc->br_series[2 * b + 1]), sh(desc->br_shunt[2 * b], desc->br_shunt[2 * b + 1]);
      const cplx tap(desc->br_tap[2 * b], desc->br_tap[2 * b + 1]);
      m->ybus[f * NB + t] = -ys / std::conj(tap);
      m->ybus[t * NB + f] = -ys / tap;
      m->ybus[f * NB + f] += (ys + sh) / (std::abs(tap) * std::abs(tap));
      m->ybus[t * NB + t] += ys + sh;
    }
  }
  int rc = upload_const(m);
  if (rc == 0 && (hipMalloc(&m->d_zero, sizeof(int32_t)) != hipSuccess || hipMemset(m->d_zero, 0, sizeof(int32_t)) != hipSuccess))
    rc = fail("hipMalloc(class selector)");
  if (rc == 0) {   // table of the stand-alone sampler (class 0)
    std::vector<double> tab(size_t(6) * desc->n_dev, 0.0);
    m->s_nd = desc->n_dev;
    for (int k = 0; k < desc->n_dev; ++k) {
      const int t = desc->dev_type[k];
      int slot = -1;
      if (t == DEV_LOAD) slot = m->s_nload++;
      else if (t == DEV_CLASSICAL || t == DEV_RENEWABLE) slot = m->s_ngen++;
      else if (t == DEV_STORAGE) slot = m->s_ndes++;
      double* o = &tab[6 * size_t(k)];
      o[0] = t; o[1] = slot; o[2] = desc->dev_qmin[k]; o[3] = desc->dev_qmax[k];
      o[4] = t == DEV_STORAGE ? desc->dev_soc_min[k] : 0.0; o[5] = t == DEV_STORAGE ? desc->dev_soc_max[k] : 0.0;
    }
    if (hipMalloc(&m->d_samp, tab.size() * sizeof(double) + 8) != hipSuccess ||
        hipMemcpy(m->d_samp, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
      rc = fail("hipMalloc(sampler table)");
  }
  if (rc) {
    anm_model_destroy(m);
    return rc;
  }
  *out = m;
  return 0;
}

void anm_model_destroy(anm_model* m) {
  if (!m) return;
  if (m->d_const) hipFree(m->d_const);
  if (m->d_series) hipFree(m->d_series);
  if (m->d_ri) hipFree(m->d_ri);
  if (m->d_rd) hipFree(m->d_rd);
  if (m->d_mi) hipFree(m->d_mi);
  if (m->d_md) hipFree(m->d_md);
  if (m->d_zero) hipFree(m->d_zero);
  if (m->d_samp) hipFree(m->d_samp);
  if (m->d_obs_index) hipFree(m->d_obs_index);
  if (m->d_obs_tab) hipFree(m->d_obs_tab);
  delete m;
}

int anm_model_dims(const anm_model* m, anm_dims* out) {
  if (!m || !out) return fail("anm_model_dims: null argument");
  if (!m->tpe_ok) {
    auto fill = [&](const auto& d) {
      out->n_bus = d.NB; out->n_dev = d.ND; out->n_branch = d.NBR; out->n_load = d.NLOAD; out->n_gen = d.NGEN;
      out->n_des = d.NDES; out->action_dim = 2 * (d.NGEN + d.NDES); out->state_base_dim = d.SDIM;
      out->full_dim = d.FS; out->const_doubles = d.n_double;
    };
    if (m->radial_ok) fill(m->plan.d);
    else fill(m->mplan.d);
    return 0;
  }
  out->n_bus = Topo::NB;
  out->n_dev = Topo::ND;
  out->n_branch = Topo::NBR;
  out->n_load = Topo::NLOAD;
  out->n_gen = Topo::NGEN;
  out->n_des = Topo::NDES;
  out->action_dim = Dims<Topo>::ADIM;
  out->state_base_dim = Topo::SDIM;
  out->full_dim = FullState<Topo>::SIZE;
  out->const_doubles = Layout<Topo>::TOTAL;
  return 0;
}

int anm_model_full_layout(const anm_model* m, anm_full_layout* o) {
  if (!m || !o) return fail("anm_model_full_layout: null argument");
  if (!m->tpe_ok && !m->radial_ok) {
    const mesh::Dims& d = m->mplan.d;
    o->bus_p = d.f_bus_p; o->bus_q = d.f_bus_q; o->bus_v_magn = d.f_bus_vm; o->bus_v_ang = d.f_bus_va;
    o->bus_i_magn = d.f_bus_im; o->bus_i_ang = d.f_bus_ia; o->dev_p = d.f_dev_p; o->dev_q = d.f_dev_q;
    o->des_soc = d.f_des_soc; o->gen_p_max = d.f_gen_pmax; o->branch_p = d.f_br_p; o->branch_q = d.f_br_q;
    o->branch_s = d.f_br_s; o->branch_i_magn = d.f_br_im; o->branch_i_ang = d.f_br_ia; o->size = d.FS;
    return 0;
  }
  if (!m->tpe_ok) {
    const radial::Dims& d = m->plan.d;
    o->bus_p = d.f_bus_p; o->bus_q = d.f_bus_q; o->bus_v_magn = d.f_bus_vm; o->bus_v_ang = d.f_bus_va;
    o->bus_i_magn = d.f_bus_im; o->bus_i_ang = d.f_bus_ia; o->dev_p = d.f_dev_p; o->dev_q = d.f_dev_q;
    o->des_soc = d.f_des_soc; o->gen_p_max = d.f_gen_pmax; o->branch_p = d.f_br_p; o->branch_q = d.f_br_q;
    o->branch_s = d.f_br_s; o->branch_i_magn = d.f_br_im; o->branch_i_ang = d.f_br_ia; o->size = d.FS;
    return 0;
  }
  typedef FullState<Topo> F;
  o->bus_p = F::BUS_P; o->bus_q = F::BUS_Q; o->bus_v_magn = F::BUS_VM; o->bus_v_ang = F::BUS_VA;
  o->bus_i_magn = F::BUS_IM; o->bus_i_ang = F::BUS_IA; o->dev_p = F::DEV_P; o->dev_q = F::DEV_Q;
  o->des_soc = F::DES_SOC; o->gen_p_max = F::GEN_PMAX; o->branch_p = F::BR_P; o->branch_q = F::BR_Q;
  o->branch_s = F::BR_S; o->branch_i_magn = F::BR_IM; o->branch_i_ang = F::BR_IA; o->size = F::SIZE;
  return 0;
}

int anm_model_set_env(anm_model* m, const anm_env_config* cfg) {
  if (!m || !cfg) return fail("anm_model_set_env: null argument");
  std::string err;
  if (cfg->K < 0 || cfg->K > radial::KMAX) return fail("K (number of aux variables) must be in [0, 8]");
  if (m->tpe_ok && !pack_env<Topo>(*cfg, m->h_const, err)) {
    g_err = err;
    return -3;
  }
  if (m->tpe_ok)
    for (auto& xc : m->x_const) pack_env<Topo>(*cfg, xc, err);
  m->K = cfg->K;
  if (m->radial_ok) {
    radial::Plan& P = m->plan;
    P.hd[radial::SF_C1] = cfg->clip_e_loss;
    P.hd[radial::SF_C2] = cfg->clip_penalty;
    P.hd[radial::SF_RTERM] = -cfg->clip_penalty / (1 - cfg->gamma);
    P.hd[radial::SF_PERIOD] = cfg->period;
    for (int k = 0; k < P.d.SDIM + cfg->K; ++k) {
      if (cfg->obs_low) P.hd[P.d.off_obs_lo + k] = cfg->obs_low[k];
      if (cfg->obs_high) P.hd[P.d.off_obs_hi + k] = cfg->obs_high[k];
    }
    for (auto& xh : m->x_hd) {
      for (int f : {int(radial::SF_C1), int(radial::SF_C2), int(radial::SF_RTERM), int(radial::SF_PERIOD)}) xh[f] = P.hd[f];
      for (int k = 0; k < P.d.SDIM + cfg->K; ++k) {
        xh[P.d.off_obs_lo + k] = P.hd[P.d.off_obs_lo + k];
        xh[P.d.off_obs_hi + k] = P.hd[P.d.off_obs_hi + k];
      }
    }
  }
  if (m->mesh_ok) {
    mesh::Plan& P = m->mplan;
    auto apply = [&](std::vector<double>& hd) {
      hd[radial::SF_C1] = cfg->clip_e_loss;
      hd[radial::SF_C2] = cfg->clip_penalty;
      hd[radial::SF_RTERM] = -cfg->clip_penalty / (1 - cfg->gamma);
      hd[radial::SF_PERIOD] = cfg->period;
      for (int k = 0; k < P.d.SDIM + cfg->K; ++k) {
        if (cfg->obs_low) hd[P.d.off_obs_lo + k] = cfg->obs_low[k];
        if (cfg->obs_high) hd[P.d.off_obs_hi + k] = cfg->obs_high[k];
      }
    };
    apply(P.hd);
    for (auto& x : m->x_md) apply(x);
  }
  if (m->d_series) {
    hipFree(m->d_series);
    m->d_series = nullptr;
  }
  m->period = 0;
  if (cfg->series && cfg->period > 0) {
    if (cfg->K != 1) return fail("series mode needs exactly K = 1 auxiliary variable (the time index)");
    const size_t nexo = m->tpe_ok ? size_t(Dims<Topo>::NEXO)
                                  : (m->radial_ok ? size_t(m->plan.d.NLOAD + m->plan.d.NGEN) : size_t(m->mplan.d.NLOAD + m->mplan.d.NGEN));
    const size_t bytes = sizeof(double) * nexo * size_t(cfg->period);
    hipError_t e = hipMalloc(&m->d_series, bytes ? bytes : 8);
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(series)");
    e = hipMemcpy(m->d_series, cfg->series, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpy(series)");
    m->period = cfg->period;
  }
  m->env_set = true;
  return upload_const(m);
}

int anm_model_set_classes(anm_model* m, int32_t n_classes, const anm_network_desc* const* descs) {
  if (!m) return fail("anm_model_set_classes: null model");
  if (n_classes > 1 && m->has_view)
    return fail("anm_model_set_classes: not while a batch view is bound (anm_model_bind_view)");
  if (n_classes < 1 || n_classes > 65536) return fail("anm_model_set_classes: n_classes must be in [1, 65536]");
  if (n_classes > 1 && !descs) return fail("anm_model_set_classes: null descriptions");
  std::vector<std::vector<double>> xc, xh, xm;
  for (int k = 1; k < n_classes; ++k) {
    if (!descs[k]) return fail("anm_model_set_classes: null description");
    std::string err;
    if (m->tpe_ok) {
      std::vector<double> cbuf;
      std::vector<cplx> y;
      if (!pack_constants<Topo>(*descs[k], cbuf, y, err)) {
        g_err = "class " + std::to_string(k) + ": " + err;
        return -3;
      }
      xc.push_back(std::move(cbuf));
    }
    if (m->radial_ok) {
      radial::Plan P;
      if (!radial::build_plan(*descs[k], P, err) || P.hi != m->plan.hi || P.hd.size() != m->plan.hd.size()) {
        g_err = "class " + std::to_string(k) + ": not the topology of the model" + (err.empty() ? "" : " (" + err + ")");
        return -3;
      }
      xh.push_back(std::move(P.hd));
    }
    if (m->mesh_ok) {
      mesh::Plan P;
      if (!mesh::build_plan(*descs[k], P, err) || P.hi != m->mplan.hi || P.hd.size() != m->mplan.hd.size()) {
        g_err = "class " + std::to_string(k) + ": not the topology of the model" + (err.empty() ? "" : " (" + err + ")");
        return -3;
      }
      xm.push_back(std::move(P.hd));
    }
  }
  // (re)allocate the device buffers for n_classes consecutive copies
  if (m->tpe_ok) {
    double* p = nullptr;
    hipError_t e = hipMalloc(&p, size_t(n_classes) * m->h_const.size() * sizeof(double));
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(class constants)");
    hipFree(m->d_const);
    m->d_const = p;
  }
  if (m->radial_ok) {
    double* p = nullptr;
    hipError_t e = hipMalloc(&p, size_t(n_classes) * m->plan.hd.size() * sizeof(double));
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(class tables)");
    hipFree(m->d_rd);
    m->d_rd = p;
  }
  if (m->mesh_ok) {
    double* p = nullptr;
    hipError_t e = hipMalloc(&p, size_t(n_classes) * m->mplan.hd.size() * sizeof(double));
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(class tables)");
    hipFree(m->d_md);
    m->d_md = p;
  }
  m->x_const = std::move(xc);
  m->x_hd = std::move(xh);
  m->x_md = std::move(xm);
  m->d_env_class = nullptr;
  m->class_per_env = false;
  if (m->impl_unbound >= 0) { m->impl = m->impl_unbound; m->impl_unbound = -1; }
  m->env_set = false;   // the task constants (anm_model_set_env) must be set again: they live in every class
  return upload_const(m);
}

int anm_model_set_class_obs_bounds(anm_model* m, int32_t cls, const double* low, const double* high) {
  if (!m || !low || !high) return fail("anm_model_set_class_obs_bounds: null argument");
  if (!m->env_set) return fail("anm_model_set_class_obs_bounds: call anm_model_set_env first (it writes every class)");
  const int n_cls = 1 + int(std::max(std::max(m->x_const.size(), m->x_hd.size()), m->x_md.size()));
  if (cls < 0 || cls >= n_cls) return fail("anm_model_set_class_obs_bounds: no such class");
  if (m->tpe_ok) {
    typedef Layout<Topo> L;
    std::vector<double>& c = cls == 0 ? m->h_const : m->x_const[cls - 1];
    for (int k = 0; k < Topo::SDIM + m->K; ++k) { c[L::OBS_LO + k] = low[k]; c[L::OBS_HI + k] = high[k]; }
  }
  if (m->radial_ok) {
    std::vector<double>& h = cls == 0 ? m->plan.hd : m->x_hd[cls - 1];
    for (int k = 0; k < m->plan.d.SDIM + m->K; ++k) { h[m->plan.d.off_obs_lo + k] = low[k]; h[m->plan.d.off_obs_hi + k] = high[k]; }
  }
  if (m->mesh_ok) {
    std::vector<double>& h = cls == 0 ? m->mplan.hd : m->x_md[cls - 1];
    for (int k = 0; k < m->mplan.d.SDIM + m->K; ++k) { h[m->mplan.d.off_obs_lo + k] = low[k]; h[m->mplan.d.off_obs_hi + k] = high[k]; }
  }
  return upload_const(m);
}

int anm_model_bind_env_classes(anm_model* m, const int32_t* env_class, int64_t num_envs) {
  if (!m) return fail("anm_model_bind_env_classes: null model");
  auto restore_impl = [&]() {   // a binding that forced a lane-group family is gone: back to the family it displaced
    if (m->impl_unbound >= 0) m->impl = m->impl_unbound;
    m->impl_unbound = -1;
  };
  // Lifting a binding (or rebinding in aligned blocks) takes the model back to the family the binding displaced.  A
  // list-form observation set meanwhile has the tables of the lane-group family it was set in (identity layout, no row
  // stride): the thread-per-environment kernels would gather from overlapping LDS rows.  Refused, like the forward
  // direction below and anm_model_set_impl.
  auto restore_refused = [&]() {
    return m->n_obs > 0 && m->impl_unbound >= 0 && m->impl_unbound != m->impl;
  };
  const char* const restore_msg =
      "anm_model_bind_env_classes: lifting this binding moves the model back to the kernel family it displaced, and the "
      "list-form observation that is set (anm_model_set_obs) has the tables of the current family: clear it first "
      "(n_obs = 0) and set it again afterwards";
  if (!env_class) {
    if (restore_refused()) return fail(restore_msg);
    m->d_env_class = nullptr;
    m->class_per_env = false;
    restore_impl();
    return 0;
  }
  if (m->has_view)   // (the same rule as anm_model_bind_view, from the other side: k_mesh looks a block's
    // class up by launch slot and an environment's by its index in the batch)
    return fail("anm_model_bind_env_classes: not while a batch view is bound (anm_model_bind_view)");
  if (num_envs <= 0) return fail("anm_model_bind_env_classes: num_envs must be positive");
  const int n_classes = 1 + int(m->tpe_ok ? m->x_const.size() : (m->radial_ok ? m->x_hd.size() : m->x_md.size()));
  std::vector<int32_t> h(static_cast<size_t>(num_envs));
  hipError_t e = hipMemcpy(h.data(), env_class, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail_hip(e, "anm_model_bind_env_classes");
  bool blocks = true;
  for (int64_t i = 0; i < num_envs; ++i) {
    if (h[i] < 0 || h[i] >= n_classes) return fail("anm_model_bind_env_classes: class index out of range");
    blocks = blocks && h[i] == h[i - i % 64];
  }
  // One class per aligned block of 64 environments: the constants of a wavefront stay wave-uniform (scalar loads) in
  // every kernel family.  Any other assignment -- a different network in every environment -- is served by the
  // lane-group families (one environment per lane group, its constants read by vector loads); the
  // thread-per-environment kernels cannot (64 environments share a wavefront's scalar constants).
  if (!blocks && !(m->radial_ok || m->mesh_ok))
    return fail("anm_model_bind_env_classes: a class must cover whole aligned blocks of 64 environments (this network has no lane-group kernel)");
  // (the same guard as anm_model_set_impl: the lane-group kernels do not gather a list-form observation -- moving a
  // model with one to them would silently turn its observation into clip(state))
  if (!blocks && m->impl == ANM_IMPL_THREAD && m->n_obs > 0)
    return fail("anm_model_bind_env_classes: classes that change inside blocks of 64 environments move the model to a lane-group "
                "kernel, and the list-form observation that is set (anm_model_set_obs) has the tables of the thread-per-environment "
                "kernel: clear it first (n_obs = 0) and set it again after binding, or bind the classes in aligned blocks of 64");
  if (blocks && restore_refused()) return fail(restore_msg);
  m->class_per_env = !blocks;
  m->d_env_class = env_class;
  if (blocks) restore_impl();
  else if (m->impl == ANM_IMPL_THREAD) {
    m->impl_unbound = ANM_IMPL_THREAD;
    m->impl = m->radial_ok ? ANM_IMPL_RADIAL : ANM_IMPL_MESH;
  }
  return 0;
}

int anm_model_bind_state_same(anm_model* m, uint8_t* state_same) {
  if (!m) return fail("anm_model_bind_state_same: null model");
  if (state_same && m->has_view) return fail("anm_model_bind_state_same: not while a batch view is bound");
  m->d_state_same = state_same;
  return 0;
}

int anm_model_bind_nr_diff(anm_model* m, double* nr_diff) {
  if (!m) return fail("anm_model_bind_nr_diff: null model");
  m->d_nr_diff = nr_diff;
  return 0;
}

int anm_model_bind_nr_start(anm_model* m, const double* x0) {
  if (!m) return fail("anm_model_bind_nr_start: null model");
  m->d_nr_start = x0;
  return 0;
}

int anm_model_bind_view(anm_model* m, const anm_batch_view* v) {
  if (!m) return fail("anm_model_bind_view: null model");
  if (!v) {
    m->view = radial::View{};
    m->has_view = false;
    return 0;
  }
  if (m->d_env_class) return fail("anm_model_bind_view: not together with parameter classes (anm_model_bind_env_classes)");
  if (m->n_obs > 0 && m->impl == ANM_IMPL_THREAD)
    return fail("anm_model_bind_view: the thread-per-environment family gathers no list-form observation through a view "
                "(anm_model_set_obs): clear it, or move the model to a lane-group family first (anm_model_set_impl)");
  if (m->n_obs > 0 && v->w_obs != 0 && v->w_obs < m->n_obs) return fail("anm_model_bind_view: w_obs is narrower than the observation list");
  if (m->d_state_same) return fail("anm_model_bind_view: not together with anm_model_bind_state_same (the flags are indexed by launch slot)");
  anm_dims d;
  anm_model_dims(m, &d);
  const int K = m->K;
  const int n_set = d.n_gen + d.n_des;
  struct { int given, own; const char* what; } w[] = {
      {v->w_load, d.n_load, "w_load"}, {v->w_gen, d.n_gen, "w_gen"}, {v->w_set, n_set, "w_set"}, {v->w_des, d.n_des, "w_des"},
      {v->w_action, 2 * n_set, "w_action"}, {v->w_state, d.state_base_dim + K, "w_state"}, {v->w_exo, d.n_load + d.n_gen, "w_exo"},
      {v->w_aux, K, "w_aux"}, {v->w_full, d.full_dim, "w_full"}};
  for (auto& x : w)
    if (x.given != 0 && x.given < x.own) {
      g_err = std::string("anm_model_bind_view: ") + x.what + " is narrower than this network's own rows";
      return -1;
    }
  m->view = radial::View{v->env_index, v->w_load, v->w_gen, v->w_set, v->w_des, v->w_action, v->w_state, v->w_exo, v->w_aux, v->w_full, v->w_obs};
  m->has_view = true;
  return 0;   // (every kernel family serves a view: the model stays in the family it is in)
}

int anm_step_ws_record_doubles(void) { return Rec<Topo>::SIZE; }

static unsigned magic_div(int d) { return d > 0 ? unsigned((0x100000000ull + uint64_t(d) - 1) / uint64_t(d)) : 0u; }

int anm_model_obs_fusable(const anm_model* m) {
  if (!m) return 0;
  if (m->has_view && m->impl == ANM_IMPL_THREAD) return 0;   // (its step through a view moves rows per lane: no LDS rows to gather from)
  if (m->impl == ANM_IMPL_THREAD) return (m->tpe_ok && GenLds<Topo>::FULL_OK) ? 1 : 0;
  if (m->impl == ANM_IMPL_RADIAL) return (m->radial_ok && radial_obs_lds_bytes(m) <= 48 * 1024) ? 1 : 0;
  if (m->impl == ANM_IMPL_MESH) return (m->mesh_ok && m->mplan.d.l_bw - m->mplan.d.l_blk >= m->mplan.d.FS + radial::KMAX) ? 1 : 0;
  return 0;
}

int anm_model_set_obs(anm_model* m, int32_t n_obs, const int32_t* index, const double* scale, const double* low,
                      const double* high) {
  if (!m) return fail("anm_model_set_obs: null model");
  if (m->d_obs_index) { hipFree(m->d_obs_index); m->d_obs_index = nullptr; }
  if (m->d_obs_tab) { hipFree(m->d_obs_tab); m->d_obs_tab = nullptr; }
  m->n_obs = 0;
  if (n_obs <= 0) return 0;   // back to the "state" observation
  if (!anm_model_obs_fusable(m))
    return fail("anm_model_set_obs: this model cannot gather inside the step kernel (use anm_gather_obs_f64)");
  if (!index || !scale || !low || !high) return fail("anm_model_set_obs: null argument");
  if (n_obs > 4096) return fail("anm_model_set_obs: more than 4096 observation entries");
  if (m->has_view && m->view.w_obs != 0 && m->view.w_obs < n_obs) return fail("anm_model_set_obs: the bound view's w_obs is narrower than the list");
  if (m->impl != ANM_IMPL_THREAD) {
    // lane-group families: the identity layout of `full` (runtime offsets of the network), the classes the list reads
    const bool rad = m->impl == ANM_IMPL_RADIAL;
    const int FS = rad ? m->plan.d.FS : m->mplan.d.FS;
    int base[FC_COUNT + 1];
    if (rad) {
      const radial::Dims& d = m->plan.d;
      const int b[FC_COUNT] = {d.f_bus_p, d.f_bus_q, d.f_bus_vm, d.f_bus_va, d.f_bus_im, d.f_bus_ia, d.f_dev_p, d.f_dev_q, d.f_des_soc,
                               d.f_gen_pmax, d.f_br_p, d.f_br_q, d.f_br_s, d.f_br_im, d.f_br_ia};
      for (unsigned c = 0; c < FC_COUNT; ++c) base[c] = b[c];
    } else {
      const mesh::Dims& d = m->mplan.d;
      const int b[FC_COUNT] = {d.f_bus_p, d.f_bus_q, d.f_bus_vm, d.f_bus_va, d.f_bus_im, d.f_bus_ia, d.f_dev_p, d.f_dev_q, d.f_des_soc,
                               d.f_gen_pmax, d.f_br_p, d.f_br_q, d.f_br_s, d.f_br_im, d.f_br_ia};
      for (unsigned c = 0; c < FC_COUNT; ++c) base[c] = b[c];
    }
    base[FC_COUNT] = FS;
    unsigned need = 0;
    std::vector<int32_t> idx(2 * size_t(n_obs));
    for (int k = 0; k < n_obs; ++k) {
      const int i = index[k];
      // (the aux slots behind the electrical state: only the K of the task are written, anm_model_set_env)
      if (i < 0 || i >= FS + (m->env_set ? m->K : radial::KMAX)) return fail("anm_model_set_obs: index out of range");
      if (i < FS) {
        unsigned c = 0;
        while (c + 1 < FC_COUNT && base[c + 1] <= i) ++c;
        need |= 1u << c;
      }
      idx[k] = idx[n_obs + k] = i;   // (both halves: the identity layout)
    }
    std::vector<double> tab(3 * size_t(n_obs));
    for (int k = 0; k < n_obs; ++k) { tab[k] = scale[k]; tab[n_obs + k] = low[k]; tab[2 * n_obs + k] = high[k]; }
    hipError_t e = hipMalloc(&m->d_obs_index, idx.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&m->d_obs_tab, tab.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(m->d_obs_index, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_obs_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail_hip(e, "anm_model_set_obs");
    m->obs_need = need;
    m->n_obs = n_obs;
    return 0;
  }
  typedef FullState<Topo> F;
  const int FS = F::SIZE;
  int cls_size[FC_COUNT];
  for (unsigned c = 0; c < FC_COUNT; ++c)
    cls_size[c] = (c + 1 < FC_COUNT ? full_class_base<Topo>(c + 1) : FS) - full_class_base<Topo>(c);
  unsigned need = 0;
  for (int k = 0; k < n_obs; ++k) {
    const int i = index[k];
    if (i < 0 || i >= FS + (m->env_set ? m->K : Layout<Topo>::KMAX)) return fail("anm_model_set_obs: index out of range");
    if (i < FS) {
      unsigned c = 0;
      while (c + 1 < FC_COUNT && full_class_base<Topo>(c + 1) <= i) ++c;
      need |= 1u << c;
    }
  }
  int off = 0;
  for (unsigned c = 0; c < FC_COUNT; ++c) {
    m->obs_cls_off[c] = short(off);
    if ((need >> c) & 1u) off += cls_size[c];
  }
  m->obs_need = need;
  m->obs_aux_off = off;
  m->obs_row_stride = (off + Layout<Topo>::KMAX) | 1;
  std::vector<int32_t> idx(2 * size_t(n_obs));
  for (int k = 0; k < n_obs; ++k) {
    const int i = index[k];
    idx[n_obs + k] = i;  // identity layout: FullState offsets, aux behind them
    if (i >= FS) {
      idx[k] = m->obs_aux_off + (i - FS);
    } else {
      unsigned c = 0;
      while (c + 1 < FC_COUNT && full_class_base<Topo>(c + 1) <= i) ++c;
      idx[k] = m->obs_cls_off[c] + (i - full_class_base<Topo>(c));
    }
  }
  std::vector<double> tab(3 * size_t(n_obs));
  for (int k = 0; k < n_obs; ++k) { tab[k] = scale[k]; tab[n_obs + k] = low[k]; tab[2 * n_obs + k] = high[k]; }
  hipError_t e = hipMalloc(&m->d_obs_index, idx.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&m->d_obs_tab, tab.size() * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(m->d_obs_index, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_obs_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail_hip(e, "anm_model_set_obs");
  m->n_obs = n_obs;
  return 0;
}

int anm_model_set_impl(anm_model* m, int32_t impl) {
  if (!m) return fail("anm_model_set_impl: null model");
  if (impl == ANM_IMPL_RADIAL && !m->radial_ok)
    return fail("the lane-group kernel needs a radial (tree) network with at most 64 buses and devices");
  if (impl == ANM_IMPL_MESH && !m->mesh_ok)
    return fail("the general lane-group kernel needs a network of at most 65 buses, 128 branches, 64 devices");
  if (impl != ANM_IMPL_THREAD && impl != ANM_IMPL_RADIAL && impl != ANM_IMPL_MESH)
    return fail("anm_model_set_impl: unknown implementation");
  if (impl == ANM_IMPL_THREAD && !m->tpe_ok)
    return fail("this library was compiled for another topology: only the generic lane-group kernel is available");
  if (impl == ANM_IMPL_THREAD && m->class_per_env)
    return fail("anm_model_set_impl: the bound parameter classes change inside blocks of 64 environments: only a "
                "lane-group kernel can serve them");
  if (impl != m->impl && m->n_obs > 0)
    return fail("anm_model_set_impl: a list-form observation is set (anm_model_set_obs), whose tables belong to the current "
                "kernel family; clear it before switching and set it again afterwards");
  m->impl = impl;
  m->impl_unbound = -1;   // an explicit choice is not undone by a later unbind
  return 0;
}

int anm_model_get_impl(const anm_model* m) { return m ? m->impl : -1; }
int anm_model_lanes_per_env(const anm_model* m) {
  if (!m) return -1;
  return m->impl == ANM_IMPL_MESH ? m->mplan.d.G : (m->impl == ANM_IMPL_RADIAL ? m->plan.d.G : 1);
}

int anm_model_get_ybus(const anm_model* m, double* y) {
  if (!m || !y) return fail("anm_model_get_ybus: null argument");
  for (size_t k = 0; k < m->ybus.size(); ++k) {
    y[2 * k] = m->ybus[k].real();
    y[2 * k + 1] = m->ybus[k].imag();
  }
  return 0;
}

int anm_transition_f64(anm_model* m, int64_t n, const double* p_load, const double* p_pot, const double* p_set,
                       const double* q_set, double* soc, double* full, double* reward, double* e_loss,
                       double* penalty, uint8_t* converged, int32_t* nr_iters, const anm_solver_opts* opts,
                       void* stream) {
  if (!m) return fail("anm_transition_f64: null model");
  if (n <= 0) return 0;
  if (!reward || !e_loss || !penalty || !converged) return fail("anm_transition_f64: null output");
  TransitionIO io{p_load, p_pot, p_set, q_set, soc, full, reward, e_loss, penalty, converged, nr_iters, m->d_nr_diff, m->d_nr_start};
  int prec;
  SolverOpts so = solver(opts, prec);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (m->impl == ANM_IMPL_RADIAL || m->impl == ANM_IMPL_MESH) {
    radial::IO rio{};
    rio.mode = 0;
    rio.t = io;
    return m->impl == ANM_IMPL_MESH ? launch_mesh(m, prec, n, s, rio, so) : launch_radial(m, prec, n, s, rio, so);
  }
  cptr_t C = (cptr_t)m->d_const;
  const bool viewed = m->has_view;
  if (io.nr_start) {
    if (prec == ANM_SOLVE_F32)
      hipLaunchKernelGGL((k_transition<float, true, true>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false), m->view);
    else
      hipLaunchKernelGGL((k_transition<double, true, true>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false), m->view);
  } else if (viewed) {

    if (prec == ANM_SOLVE_F32)
      hipLaunchKernelGGL((k_transition<float, true>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false), m->view);
    else
      hipLaunchKernelGGL((k_transition<double, true>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false), m->view);
  } else if (prec == ANM_SOLVE_F32)
    hipLaunchKernelGGL((k_transition<float, false>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false), m->view);
  else
    hipLaunchKernelGGL((k_transition<double, false>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false), m->view);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "launch k_transition");
  return 0;
}

int anm_reset_f64(anm_model* m, int64_t n, const double* init_state, const uint8_t* mask, uint64_t rng_seed,
                  uint64_t env_offset, int32_t* reset_count, double* soc, double* state,
                  double* obs, uint8_t* converged, uint8_t* terminated, int32_t* timestep, int32_t* nr_iters,
                  double* full, int32_t* aux_index, const anm_solver_opts* opts, void* stream) {
  if (!m) return fail("anm_reset_f64: null model");
  if (!m->env_set) return fail("anm_reset_f64: call anm_model_set_env first");
  if (n <= 0) return 0;
  if (!state || !obs || !converged || !terminated) return fail("anm_reset_f64: null argument");
  if (!init_state && (m->period <= 0 || m->K != 1 || !reset_count))
    return fail("anm_reset_f64: drawing initial states on the device needs a series-mode model and reset_count");
  EnvIO io{};
  io.K = m->K;
  io.init_state = init_state;
  io.series = m->d_series;
  io.period = m->period;
  io.rng_seed = rng_seed;
  io.env_offset = env_offset;
  io.reset_count = reset_count;
  io.mask = mask;
  io.soc = soc;
  io.state = state;
  io.obs = obs;
  io.converged = converged;
  io.terminated = terminated;
  io.timestep = timestep;
  io.nr_iters = nr_iters;
  io.full = full;
  io.aux_index = aux_index;
  io.nr_diff = m->d_nr_diff;
  int prec;
  SolverOpts so = solver(opts, prec);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (m->impl == ANM_IMPL_RADIAL || m->impl == ANM_IMPL_MESH) {
    radial::IO rio{};
    rio.mode = 1;
    rio.e = io;
    return m->impl == ANM_IMPL_MESH ? launch_mesh(m, prec, n, s, rio, so) : launch_radial(m, prec, n, s, rio, so);
  }
  cptr_t C = (cptr_t)m->d_const;
  const bool viewed = m->has_view;
  if (viewed) {
    if (prec == ANM_SOLVE_F32)
      hipLaunchKernelGGL((k_reset<float, true>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false), m->view);
    else
      hipLaunchKernelGGL((k_reset<double, true>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false), m->view);
  } else if (prec == ANM_SOLVE_F32)
    hipLaunchKernelGGL((k_reset<float, false>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false), m->view);
  else
    hipLaunchKernelGGL((k_reset<double, false>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false), m->view);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "launch k_reset");
  return 0;
}

int anm_sample_init_state_f64(anm_model* m, int64_t n, uint64_t rng_seed, uint64_t env_offset, const int32_t* reset_count,
                              double* init_state, uint32_t* raw, void* stream) {
  if (!m || !init_state) return fail("anm_sample_init_state_f64: null argument");
  if (!m->env_set || m->period <= 0 || m->K != 1 || !m->d_series)
    return fail("anm_sample_init_state_f64: the model needs a series-mode task (anm_model_set_env with series, K = 1)");
  if (m->d_env_class) return fail("anm_sample_init_state_f64: not while parameter classes are bound (the table is class 0's)");
  if (n <= 0) return 0;
  const int n_blocks = 1 + (m->s_ngen + m->s_ndes + 1) / 2;
  hipLaunchKernelGGL(k_sample_init_state, dim3(unsigned((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n,
                     m->s_nd, m->s_nload, m->s_ngen, m->s_ndes, m->d_samp, m->d_series, m->period, rng_seed, env_offset,
                     reset_count, init_state, raw, n_blocks);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "launch k_sample_init_state");
  return 0;
}

static int make_step_io(anm_model* m, const double* action, const double* exo, const double* aux_next, double* soc,
                        double* state, uint8_t* terminated, int32_t* timestep, double* obs, double* reward,
                        double* e_loss, double* penalty, int32_t* nr_iters, double* full, int32_t autoreset,
                        uint64_t rng_seed, uint64_t env_offset, int32_t* reset_count, int32_t* aux_index, const anm_step_ws* ws,
                        EnvIO& io) {
  if (!m) return fail("anm_step_f64: null model");
  if (!m->env_set) return fail("anm_step_f64: call anm_model_set_env first");
  // (a network without set-point devices has an empty action vector: examples/simple_env.py of the reference)
  const int action_dim = 2 * (m->tpe_ok ? Topo::NSET : (m->radial_ok ? m->plan.d.NSET : m->mplan.d.NSET));
  if ((!action && action_dim > 0) || !state || !terminated || !obs || !reward || !e_loss || !penalty)
    return fail("anm_step_f64: null argument");
  if ((m->tpe_ok ? Topo::NDES : (m->radial_ok ? m->plan.d.NDES : m->mplan.d.NDES)) > 0 && !soc) return fail("anm_step_f64: null soc");
  const bool series = exo == nullptr;
  if (series && m->period <= 0) return fail("anm_step_f64: no exo given and the model has no series (set_env)");
  if (!series && m->K > 0 && !aux_next) return fail("anm_step_f64: exo given without aux_next");
  if (autoreset && (!series || !reset_count)) return fail("anm_step_f64: autoreset needs series mode and reset_count");
  io = EnvIO{};
  io.K = m->K;
  io.action = action;
  io.exo = exo;
  io.aux_next = aux_next;
  io.series = m->d_series;
  io.period = m->period;
  io.soc = soc;
  io.state = state;
  io.terminated = terminated;
  io.timestep = timestep;
  io.obs = obs;
  io.reward = reward;
  io.e_loss = e_loss;
  io.penalty = penalty;
  io.nr_iters = nr_iters;
  io.full = full;
  io.autoreset = autoreset;
  io.rng_seed = rng_seed;
  io.env_offset = env_offset;
  io.reset_count = reset_count;
  io.aux_index = aux_index;
  io.state_same = m->d_state_same;
  io.n_obs = 0;
  io.state_magic = magic_div((m->tpe_ok ? Topo::SDIM : (m->radial_ok ? m->plan.d.SDIM : m->mplan.d.SDIM)) + m->K);
  if (m->tpe_ok && m->impl == ANM_IMPL_THREAD && (m->n_obs > 0 || full)) {
    // rows of the electrical state in LDS: identity layout when the dump is asked for, else only the classes
    // the observation list reads
    const bool ident = full != nullptr;
    io.n_obs = m->n_obs;
    io.obs_magic = magic_div(m->n_obs);
    io.obs_need = ident ? ~0u : m->obs_need;
    io.row_stride = ident ? GenLds<Topo>::FSP : m->obs_row_stride;
    io.aux_off = ident ? FullState<Topo>::SIZE : m->obs_aux_off;
    for (unsigned cc = 0; cc < FC_COUNT; ++cc)
      io.cls_off[cc] = ident ? short(full_class_base<Topo>(cc)) : m->obs_cls_off[cc];
    if (m->n_obs > 0) {
      io.obs_index = m->d_obs_index + (ident ? m->n_obs : 0);
      io.obs_scale = m->d_obs_tab;
      io.obs_lo = m->d_obs_tab + m->n_obs;
      io.obs_hi = m->d_obs_tab + 2 * m->n_obs;
    }
  }
  if (m->impl != ANM_IMPL_THREAD && m->n_obs > 0) {   // lane-group families: identity layout (see anm_model_set_obs)
    io.n_obs = m->n_obs;
    io.obs_need = m->obs_need;
    io.obs_index = m->d_obs_index;
    io.obs_scale = m->d_obs_tab;
    io.obs_lo = m->d_obs_tab + m->n_obs;
    io.obs_hi = m->d_obs_tab + 2 * m->n_obs;
  }
  io.ws = nullptr;
  if (ws && ws->buf && m->tpe_ok && !m->d_env_class) {  // (the straggler launch packs records of all blocks together)
    // records, then the int32 list of the records the first straggler launch leaves to the second
    const int64_t cap = 2 * (ws->n_doubles - Rec<Topo>::HEADER) / (2 * Rec<Topo>::SIZE + 1);
    if (cap < 1 || ws->iter_cap < 1) return fail("anm_step_f64: step workspace too small or iter_cap < 1");
    io.ws = ws->buf;
    io.ws_cap = cap;
    io.iter_cap = ws->iter_cap;
    io.ws_list2 = reinterpret_cast<int32_t*>(ws->buf + Rec<Topo>::HEADER + cap * Rec<Topo>::SIZE);
    io.mid_cap = ws->mid_cap == 0 ? ANM_MID_CAP_DEFAULT : (ws->mid_cap < 0 ? 0 : ws->mid_cap);
    if (io.mid_cap > 0 && io.mid_cap <= io.iter_cap) io.mid_cap = 0;
  }
  return 0;
}

static int launch_step(anm_model* m, const EnvIO& io_in, int64_t n, const anm_solver_opts* opts, hipStream_t s) {
  EnvIO io = io_in;
  int prec;
  SolverOpts so = solver(opts, prec);
  if (m->impl == ANM_IMPL_RADIAL || m->impl == ANM_IMPL_MESH) {
    if (io.state_same) {
      hipError_t em = hipMemsetAsync(io.state_same, 0, size_t(n), s);
      if (em != hipSuccess) return fail_hip(em, "hipMemsetAsync(state_same)");
    }
    radial::IO rio{};
    rio.mode = 2;
    rio.e = io;
    return m->impl == ANM_IMPL_MESH ? launch_mesh(m, prec, n, s, rio, so) : launch_radial(m, prec, n, s, rio, so);
  }
  cptr_t C = (cptr_t)m->d_const;
  if (m->has_view) {
    // a batch view: per-lane rows (the coalesced-row kernels need 64 consecutive environments)
    if (io.K > 1) return fail("anm_step_f64: a batch view in the thread-per-environment family takes K <= 1 auxiliary variables (use the general lane-group family)");
    if (io.n_obs > 0) return fail("anm_step_f64: a batch view and a list-form observation do not go together");
    io.ws = nullptr;
    io.state_same = nullptr;
    io.aux_stride = m->view.w_aux;
    const bool dense = m->view_waves ? m->view_waves == 2 : n > int64_t(2) * 64 * 4 * 256;   // more than two wavefronts per SIMD of the chip
    if (prec == ANM_SOLVE_F32) {
      if (dense) hipLaunchKernelGGL((k_step_view<float, 2>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, m->view);
      else hipLaunchKernelGGL((k_step_view<float, 1>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, m->view);
    } else {
      if (dense) hipLaunchKernelGGL((k_step_view<double, 2>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, m->view);
      else hipLaunchKernelGGL((k_step_view<double, 1>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, m->view);
    }
    hipError_t ev = hipGetLastError();
    if (ev != hipSuccess) return fail_hip(ev, "launch k_step_view");
    return 0;
  }
  if (io.aux_index && io.exo == nullptr && io.K == 1 && !io.full && io.n_obs == 0) {
    // fast path: series mode, "state" observation, nothing but the batch tensors
    if (prec == ANM_SOLVE_F32)
      hipLaunchKernelGGL((k_step_rows<float, false>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false));
    else
      hipLaunchKernelGGL((k_step_rows<double, false>), dim3(grid_for(n)), dim3(BLOCK), 0, s, C, io, so, n, class_sel(m, false));
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return fail_hip(e2, "launch k_step_rows");
    if (io.ws && io.iter_cap < so.max_iter) {
      // second launch: the handed-over solves, grid-stride over the records (count lives on the device)
      // (a tree topology continues them on lane groups, 8 records per wavefront, unless that was switched off)
      constexpr bool CAN_GROUP = Topo::TREE != 0;
      const bool groups = CAN_GROUP && so.handoff >= 0;
      const int per_block = groups ? group::Shape<Topo>::NG : BLOCK;
      const unsigned g2 = unsigned((io.ws_cap + per_block - 1) / per_block);  // covers every record
      const bool two_level = groups && io.mid_cap > 0 && io.mid_cap < so.max_iter;
      for (int level = 1; level <= (two_level ? 2 : 1); ++level) {
        const unsigned g = g2;  // (level 2: most wavefronts find nothing and leave)
        if constexpr (CAN_GROUP) {
          if (groups) {
            if (prec == ANM_SOLVE_F32)
              hipLaunchKernelGGL((k_step_stragglers<float, true>), dim3(g), dim3(BLOCK), 0, s, C, io, so, level);
            else
              hipLaunchKernelGGL((k_step_stragglers<double, true>), dim3(g), dim3(BLOCK), 0, s, C, io, so, level);
          }
        }
        if (!groups) {
          if (prec == ANM_SOLVE_F32)
            hipLaunchKernelGGL((k_step_stragglers<float, false>), dim3(g), dim3(BLOCK), 0, s, C, io, so, level);
          else
            hipLaunchKernelGGL((k_step_stragglers<double, false>), dim3(g), dim3(BLOCK), 0, s, C, io, so, level);
        }
        e2 = hipGetLastError();
        if (e2 != hipSuccess) return fail_hip(e2, "launch k_step_stragglers");
      }
      const unsigned g3 = unsigned((int64_t(io.ws_cap) * SCATTER_LANES + 255) / 256);
      hipLaunchKernelGGL(k_step_scatter, dim3(g3), dim3(256), 0, s, io);
      e2 = hipGetLastError();
      if (e2 != hipSuccess) return fail_hip(e2, "launch k_step_scatter");
    }
    return 0;
  }
  io.ws = nullptr;
  if (io.state_same) {  // this kernel writes every state row: no row is "the same as obs and left out"
    hipError_t em = hipMemsetAsync(io.state_same, 0, size_t(n), s);
    if (em != hipSuccess) return fail_hip(em, "hipMemsetAsync(state_same)");
  }
  if (io.full && !GenLds<Topo>::FULL_OK) {  // rows too wide for LDS: plain per-lane dump, no fused list
    io.n_obs = 0;
  }
  // dynamic LDS: the widest of the buffers op_step_general overlays
  const int S = Topo::SDIM + io.K;
  size_t doubles = size_t(GenLds<Topo>::G_MIN);
  doubles = std::max(doubles, size_t(64) * size_t(Dims<Topo>::ADIM + 1));
  if (GenLds<Topo>::FULL_OK && (io.n_obs > 0 || io.full)) doubles = std::max(doubles, size_t(64) * size_t(io.row_stride));
  io.state_row_off = int(doubles);               // the state rows follow the buffer the other uses overlay
  const size_t lds_bytes = (doubles + size_t(64) * size_t(S | 1)) * sizeof(double);
  if (prec == ANM_SOLVE_F32)
    hipLaunchKernelGGL(k_step_general<float>, dim3(grid_for(n)), dim3(BLOCK), lds_bytes, s, C, io, so, n, class_sel(m, false));
  else
    hipLaunchKernelGGL(k_step_general<double>, dim3(grid_for(n)), dim3(BLOCK), lds_bytes, s, C, io, so, n, class_sel(m, false));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "launch k_step_general");
  return 0;
}

int anm_step_f64(anm_model* m, int64_t n, const double* action, const double* exo, const double* aux_next,
                 double* soc, double* state, uint8_t* terminated, int32_t* timestep, double* obs, double* reward,
                 double* e_loss, double* penalty, int32_t* nr_iters, double* full, int32_t autoreset,
                 uint64_t rng_seed, uint64_t env_offset, int32_t* reset_count, int32_t* aux_index, const anm_step_ws* ws,
                 const anm_solver_opts* opts, void* stream) {
  EnvIO io;
  int rc = make_step_io(m, action, exo, aux_next, soc, state, terminated, timestep, obs, reward, e_loss, penalty,
                        nr_iters, full, autoreset, rng_seed, env_offset, reset_count, aux_index, ws, io);
  if (rc) return rc;
  if (n <= 0) return 0;
  return launch_step(m, io, n, opts, static_cast<hipStream_t>(stream));
}

int anm_time_step_launches(anm_model* m, int64_t n, const double* action, double* soc, double* state,
                           uint8_t* terminated, int32_t* timestep, double* obs, double* reward, double* e_loss,
                           double* penalty, int32_t autoreset, uint64_t rng_seed, uint64_t env_offset, int32_t* reset_count,
                           int32_t* aux_index, anm_step_ws* ws, const anm_solver_opts* opts, void* stream, int32_t n_launch,
                           float* ms_per_launch) {
  EnvIO io;
  int rc = make_step_io(m, action, nullptr, nullptr, soc, state, terminated, timestep, obs, reward, e_loss, penalty,
                        nullptr, nullptr, autoreset, rng_seed, env_offset, reset_count, aux_index, ws, io);
  if (rc) return rc;
  if (n <= 0 || n_launch <= 0 || !ms_per_launch) return fail("anm_time_step_launches: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t t0, t1;
  hipError_t e;
  if ((e = hipEventCreate(&t0)) != hipSuccess) return fail_hip(e, "hipEventCreate");
  if ((e = hipEventCreate(&t1)) != hipSuccess) return fail_hip(e, "hipEventCreate");
  hipEventRecord(t0, s);
  for (int k = 0; k < n_launch && rc == 0; ++k) {
    rc = launch_step(m, io, n, opts, s);
  }
  hipEventRecord(t1, s);
  e = hipEventSynchronize(t1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, t0, t1);
  hipEventDestroy(t0);
  hipEventDestroy(t1);
  if (rc) return rc;
  if (e != hipSuccess) return fail_hip(e, "event timing");
  *ms_per_launch = ms / float(n_launch);
  return 0;
}

#include "anm_mpc_capi.inc"

int anm_gather_obs_f64(int64_t n, int32_t full_dim, const double* full, int32_t state_dim, int32_t K,
                       const double* state, const uint8_t* terminated, int32_t n_obs, const int32_t* index,
                       const double* scale, const double* low, const double* high, double* obs, void* stream) {
  if (!full || !index || !scale || !low || !high || !obs) return fail("anm_gather_obs_f64: null argument");
  if (K > 0 && !state) return fail("anm_gather_obs_f64: aux variables need the state array");
  if (n <= 0 || n_obs <= 0) return 0;
  const int64_t total = n * n_obs;
  unsigned grid = unsigned(std::min<int64_t>((total + 255) / 256, 2048));
  hipLaunchKernelGGL(k_gather_obs, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), n, full_dim, full,
                     state_dim, K, state, terminated, n_obs, index, scale, low, high, obs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "launch k_gather_obs");
  return 0;
}

#ifdef ANM_PHASE_TIMING
// tuning builds only: copy the per-wave phase timestamps of the last launches to the host
__attribute__((visibility("default"))) int anm_debug_phase_times(unsigned long long* out, int n_waves) {
  std::vector<unsigned long long> tmp(size_t(ANM_N_PHASES) * ANM_MAX_WAVES);
  if (hipMemcpyFromSymbol(tmp.data(), HIP_SYMBOL(g_anm_phase), tmp.size() * 8) != hipSuccess) return -1;
  for (int k = 0; k < ANM_N_PHASES; ++k)
    for (int w = 0; w < n_waves && w < ANM_MAX_WAVES; ++w) out[k * n_waves + w] = tmp[size_t(k) * ANM_MAX_WAVES + w];
  return 0;
}
#endif

}  // extern "C"
