/* Driving the C ABI (include/anm_mi355x.h) from plain C: no Python, no torch.
 *
 * Four environments of the 2-bus network of the reference's own tests
 * (tests/simulator/test_simulator_transitions.py:18-32: slack generator at bus 0, one load at bus 1
 * over a single line r = 0.01, x = 0.1 p.u., baseMVA = 1) take one Simulator.transition each with
 * loads of 0.5, 1, 2 and 5 MW.  The last one has no power-flow solution: like the reference, that is
 * not an error, the environment simply reports converged = 0.
 *
 *   hipcc -x c examples/transition_from_c.c -I include gym_anm_amd/_build/libanm_n2b1d2_*.so -o /tmp/ex
 *   (or gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ... -L/opt/rocm/lib -lamdhip64)
 *
 * The per-unit arrays below are what gym_anm_amd/_lib.py::network_desc derives from the network
 * dictionary (gym_anm_amd/networks.py::two_bus_network).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "anm_mi355x.h"

#define CHECK_HIP(x)                                                         \
  do {                                                                       \
    hipError_t e_ = (x);                                                     \
    if (e_ != hipSuccess) {                                                  \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
      return 2;                                                              \
    }                                                                        \
  } while (0)
#define CHECK_ANM(x)                                                         \
  do {                                                                       \
    if ((x) != 0) {                                                          \
      fprintf(stderr, "%s: %s\n", #x, anm_last_error());                     \
      return 3;                                                              \
    }                                                                        \
  } while (0)

enum { E = 4 };

int main(void) {
  /* ---- the network, per unit ---- */
  const double bus_vmin[2] = {1.0, 0.9}, bus_vmax[2] = {1.0, 1.1};
  const int32_t br_from[1] = {0}, br_to[1] = {1};
  const double br_series[2] = {0.99009900990099, -9.900990099009901}; /* 1 / (0.01 + 0.1j) */
  const double br_shunt[2] = {0.0, 0.0}, br_tap[2] = {1.0, 0.0}, br_rate[1] = {32.0};
  const int32_t dev_type[2] = {0, -1}, dev_bus[2] = {0, 1}; /* slack generator, load */
  const double dev_pmin[2] = {-200.0, -10.0}, dev_pmax[2] = {200.0, 0.0};
  const double dev_qmin[2] = {-200.0, -2.0}, dev_qmax[2] = {200.0, 0.0};
  const double dev_qp[2] = {NAN, 0.2};
  const double dev_tau[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dev_rho[8] = {200.0, -200.0, 0, 0, 0, 0, 0, 0};
  const double nan2[2] = {NAN, NAN};
  anm_network_desc d;
  d.n_bus = 2; d.n_dev = 2; d.n_branch = 1;
  d.base_mva = 1.0; d.delta_t = 0.25; d.lamb = 100.0;
  d.bus_vmin = bus_vmin; d.bus_vmax = bus_vmax;
  d.br_from = br_from; d.br_to = br_to; d.br_series = br_series; d.br_shunt = br_shunt; d.br_tap = br_tap;
  d.br_rate = br_rate;
  d.dev_type = dev_type; d.dev_bus = dev_bus; d.dev_pmin = dev_pmin; d.dev_pmax = dev_pmax;
  d.dev_qmin = dev_qmin; d.dev_qmax = dev_qmax; d.dev_qp = dev_qp; d.dev_tau = dev_tau; d.dev_rho = dev_rho;
  d.dev_soc_min = nan2; d.dev_soc_max = nan2; d.dev_eff = nan2;

  if (anm_device_count() < 1) {
    fprintf(stderr, "no GPU visible (library built for topology %s)\n", anm_topology_name());
    return 1;
  }
  anm_model* m = NULL;
  CHECK_ANM(anm_model_create(&d, &m));
  anm_dims dims;
  CHECK_ANM(anm_model_dims(m, &dims));
  anm_full_layout lay;
  CHECK_ANM(anm_model_full_layout(m, &lay));

  /* ---- batch buffers (device memory owned by the caller) ---- */
  const double h_load[E] = {-0.5, -1.0, -2.0, -5.0}; /* MW, one load per environment */
  double *p_load, *full, *reward, *e_loss, *penalty;
  uint8_t* conv;
  int32_t* iters;
  CHECK_HIP(hipMalloc((void**)&p_load, sizeof(h_load)));
  CHECK_HIP(hipMalloc((void**)&full, sizeof(double) * E * dims.full_dim));
  CHECK_HIP(hipMalloc((void**)&reward, sizeof(double) * E));
  CHECK_HIP(hipMalloc((void**)&e_loss, sizeof(double) * E));
  CHECK_HIP(hipMalloc((void**)&penalty, sizeof(double) * E));
  CHECK_HIP(hipMalloc((void**)&conv, E));
  CHECK_HIP(hipMalloc((void**)&iters, sizeof(int32_t) * E));
  CHECK_HIP(hipMemcpy(p_load, h_load, sizeof(h_load), hipMemcpyHostToDevice));

  /* no generators besides the slack, no storage: p_pot / p_set / q_set / soc have width 0 */
  anm_solver_opts opts;
  opts.tol = 1e-8; opts.max_iter = 100; opts.precision = ANM_SOLVE_F64;
  CHECK_ANM(anm_transition_f64(m, E, p_load, NULL, NULL, NULL, NULL, full, reward, e_loss, penalty, conv, iters,
                               &opts, NULL));
  CHECK_HIP(hipDeviceSynchronize());

  double* h_full = (double*)malloc(sizeof(double) * E * dims.full_dim);
  double h_reward[E];
  uint8_t h_conv[E];
  int32_t h_iters[E];
  CHECK_HIP(hipMemcpy(h_full, full, sizeof(double) * E * dims.full_dim, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_reward, reward, sizeof(h_reward), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_conv, conv, sizeof(h_conv), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_iters, iters, sizeof(h_iters), hipMemcpyDeviceToHost));
  for (int e = 0; e < E; ++e) {
    const double* f = h_full + (size_t)e * dims.full_dim;
    printf("env %d load %.1f MW converged %d iterations %d |V1| %.12f theta1 %.12f slack_p %.12f reward %.12e\n", e,
           h_load[e], (int)h_conv[e], (int)h_iters[e], f[lay.bus_v_magn + 1], f[lay.bus_v_ang + 1], f[lay.dev_p + 0],
           h_reward[e]);
  }
  free(h_full);
  hipFree(p_load); hipFree(full); hipFree(reward); hipFree(e_loss); hipFree(penalty); hipFree(conv); hipFree(iters);
  anm_model_destroy(m);
  return 0;
}
