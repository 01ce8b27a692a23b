// anm_mpc.hpp -- batched N-stage DC optimal power flow of the MPC policies, one LANE per STAGE.
//
// Replaces, for a whole batch of environments and in ONE launch per act(), what the reference does per
// environment through cvxpy:  MPCAgent._create_optimization_problem / _single_step_optimization_problem /
// _solve  (gym_anm/agents/mpc.py:163-319, 372-388).  The program (a linear program) is restated in the reduced,
// stage-structured form of gym_anm_amd/agents/dcopf.py -- the DC balance (mpc.py:232-245) and the load rows
// (:247-251) are eliminated on the host (build_tables below), so that per stage i there are
//
//   inputs   xi_g in [0,1] (P_g = P_min + xi_g max(0, min(P_max, forecast) - P_min), :253-258, 267-271),
//            p_c >= 0, d >= 0 (charge / discharge, P_des = d - p_c, :273-291), t_e >= 0 (epigraph of
//            max(0, |flow_e| - beta rate_e), :308-311)
//   state    sig_i = sig_{i-1} + dt eff p_c - dt/eff d   (state of charge after the stage, :282-287)
//
// and every inequality row touches either inputs of its own stage only (boxes, |theta| <= pi :293-299, the
// three epigraph rows per branch) or the state only (SoC window :288-291).  It is solved by a primal-dual
// interior-point method (Mehrotra predictor-corrector).  A Newton step is then a linear-quadratic control
// problem: each lane factors its own stage (a dense (n_gen + 2 n_des)^2 Cholesky, the epigraph variables
// eliminated in closed form), and the stages are coupled only through the n_des states of charge -- a sweep over
// the lanes of a group that is the BLOCK CHOLESKY factorisation of the step's matrix in stage order (the classical
// Riccati recursion: Lam = T + B' P B = L L', W = L^-1 B' P, P' = P - W' W; backward stable, unlike the
// information form P (I + M P)^-1 with M = B T^-1 B', which was tried first and is not).
//
// Lane layout: a group of G = 2^k >= N lanes is one environment, lane = stage; 64 / G environments per
// wavefront.  N = 1: one thread per environment.  What a lane owns (its rows' slacks and multipliers, its factor)
// lives in registers, two work arrays of its rows in LDS; the network tables are wave-uniform scalar loads; the
// lanes of a group exchange by DPP moves and v_permlane swaps (WaveGroup below).  The angle rows are left out where
// the device limits keep every angle away from pi (NoTheta below).
//
// Plain C++17 with the ANM_HD decoration: tests/hostsim compiles the very same solver with g++ (one host
// thread per lane, the exchanges replaced by a barrier-protected exchange array) as a TEST DOUBLE.
#pragma once

#include <cmath>
#include <complex>
#include <cstdint>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/anm_mi355x.h"
#include "anm_device.hpp"

namespace anm {
namespace mpc {

constexpr int pos(int n) { return n > 0 ? n : 1; }

// every loop over rows / inputs / table columns has compile-time bounds and must be unrolled: the arrays it
// indexes are registers
#if defined(__HIPCC__)
#define ANM_UFOR _Pragma("unroll") for
#else
#define ANM_UFOR for
#endif

// The angle rows |theta_b| <= pi (mpc.py:291-292) are redundant for most networks: over the whole box of the device
// limits no angle comes near pi (ANM6: 0.2 rad).  NoTheta<Topo> is the topology WITHOUT those rows in the program: the
// solver then checks the angles of its solution instead (a solution of the relaxed program that satisfies the dropped
// rows is a solution of the full one; one that does not is reported, info[1] = 2).  anm_mpc_create picks the variant
// from a bound over the device limits (build_tables: theta_bound).
template <class T>
struct NoTheta : T {
  static constexpr bool MPC_NO_THETA = true;
};
template <class T, class = void>
struct has_theta_rows : std::true_type {};
template <class T>
struct has_theta_rows<T, std::void_t<decltype(T::MPC_NO_THETA)>> : std::false_type {};

// A SIZE CLASS instead of a topology (csrc/anm_mpc_only.hip with a class header, codegen.mpc_class_header): the solver
// depends on the network through its sizes only -- the topology lives in the tables build_tables computes on the host --
// so a kernel compiled for (n_load, n_gen, n_des, n_bus, n_branch) upper bounds serves any SMALLER network padded into
// it: loads with zero coefficients and a zero forecast, generators with the empty range [0, 0] (the solver's "the
// forecast closes the interval" case), branches with zero coefficients and a huge rating; the number of storage units
// must match (a padded unit would have no interior).  The actual counts ride in IO::nl / IO::ng.
template <class T, class = void>
struct is_padded : std::false_type {};
template <class T>
struct is_padded<T, std::void_t<decltype(T::MPC_PADDED)>> : std::true_type {};

template <class T>
struct Sz {
  static constexpr int NG = T::NGEN, NS = T::NDES, NL = T::NLOAD, NB1 = T::NB - 1, NBR = T::NBR;
  static constexpr int NC = NG + NS, NA = NG + 2 * NS, NV = NA + NBR;
  static constexpr int NTH = has_theta_rows<T>::value ? NB1 : 0;   // angle rows carried through the solve (per side)
  // rows of one stage, in this order
  static constexpr int R_XI_UP = 0, R_XI_LO = NG, R_PD_UP = 2 * NG, R_PD_LO = R_PD_UP + NS, R_PC = R_PD_LO + NS,
                       R_D = R_PC + NS, R_SOC_UP = R_D + NS, R_SOC_LO = R_SOC_UP + NS, R_TH_UP = R_SOC_LO + NS,
                       R_TH_LO = R_TH_UP + NTH, R_FL1 = R_TH_LO + NTH, R_FL2 = R_FL1 + NBR, R_FL3 = R_FL2 + NBR,
                       NR = R_FL3 + NBR;
  // what the register-resident kernel is compiled for (slacks + multipliers + work arrays of NR rows per lane)
  // Up to 72 rows, 2 storage units and 8 inputs per stage the lane's working set fits its 512 registers (no scratch);
  // beyond, up to what the LDS row arrays of a wavefront allow (2 NR x 64 doubles <= 160 KB), the same code is compiled
  // with the row arrays spilling to scratch (a 30-bus feeder with 5 storage units: 123 rows, 7 KB per lane): slower
  // per program, still one launch for all of them
  static constexpr bool IN_REGISTERS = NR <= 72 && NS <= 2 && NA <= 8;
  static constexpr bool FITS = NR <= 156 && NS <= 6 && NA <= 16;
  // the work arrays of NR rows: ds*dz always lives in LDS, 1/s above this many rows (in registers up to it: the ANM6
  // program without its angle rows, 25 rows, then runs at 470 / 498 registers, no scratch, and 6 % faster; with both
  // arrays in registers the compiler spills 476 B)
  static constexpr int ROWS_IN_REGISTERS = 28;
  static constexpr bool ROWS_IN_LDS = true;
  static constexpr size_t LDS_BYTES = ROWS_IN_LDS ? size_t(2) * NR * 64 * sizeof(double) : 0;
  // table of constants (doubles), wave-uniform
  static constexpr int T_THC = 0, T_THL = T_THC + NB1 * NC, T_PHC = T_THL + NB1 * NL, T_PHL = T_PHC + NBR * NC,
                       T_COST = T_PHL + NBR * NL, T_SGL = T_COST + pos(NC), T_LIM = T_SGL + pos(NL),
                       T_GPMIN = T_LIM + pos(NBR), T_GPMAX = T_GPMIN + pos(NG), T_SPMIN = T_GPMAX + pos(NG),
                       T_SPMAX = T_SPMIN + pos(NS), T_SOCMIN = T_SPMAX + pos(NS), T_SOCMAX = T_SOCMIN + pos(NS),
                       T_BC = T_SOCMAX + pos(NS), T_BD = T_BC + pos(NS), T_LAMB = T_BD + pos(NS), T_WGT = T_LAMB + 1,
                       T_TOTAL = T_WGT + 64;
};

struct Opts {
  double tol;        // stop: complementarity mu <= tol (1 + |objective|) (the iterate satisfies every row throughout; the
                     // dual residual is reported: a solve is to be trusted when it is small against the costs)
  int max_iter;
};

// MPCAgent.act() inside the kernel (mpc.py:321-346, 348-372, 383-388): the forecasts are gathered by the lanes (constant:
// from the state row, mpc_constant.py:24-35; perfect: from the task's periodic tables, mpc_perfect.py:24-40) and the
// first stage's set-points leave as the clipped MW action row -- no forecast tensors, no permute / scale / cat / clip launches.
struct Act {
  int mode;                   // 0: off (forecast arrays given); 1: constant forecast; 2: perfect forecast
  const double* state;        // [E][state_dim] state rows, MW: dev_p of the loads, gen_p_max; the time index in the last column
  const double* state_alt;    // rows to read where state_same[e] != 0 (the observation rows: anm_model_bind_state_same), or null
  const uint8_t* state_same;  // [E] or null
  int state_dim;
  const int32_t* aux_index;   // [E] time index (perfect forecast), or null: the last column of the state row
  const double* series;       // [NL + NG][period], MW (perfect forecast)
  int period;
  double base;                // MVA
  double* action;             // [E][2 NG + 2 NS] out: [P_gen.., 0.., P_des.., 0..] MW, clipped to [act_lo, act_hi]
  const double* act_lo;       // [2 NG + 2 NS] dev
  const double* act_hi;
  short load_col[32];         // size classes: column of each load's dev_p in a state row (a topology knows them at compile time)
  int gen_col0;               // ... and the first column of gen_p_max
};

// column of the k-th load's dev_p in a state row = its device index (anm_env.py:139-147: dev_p of every device first)
template <class T>
constexpr int load_device(int k) {
  if constexpr (is_padded<T>::value) {
    return 0;
  } else {
    for (int d = 0; d < T::ND; ++d)
      if (T::DEV_TYPE[d] == DEV_LOAD && T::DEV_SLOT[d] == k) return d;
    return 0;
  }
}
// nl: the network's own number of loads (the rows of `series` before the generators')
template <class T>
ANM_HD double act_forecast(const Act& a, int64_t env, int stage, bool gen, int k, int col_load, int nl) {
  const double* row = ((a.state_same && a.state_same[env]) ? a.state_alt : a.state) + env * a.state_dim;
  double v;
  if (a.mode == 1) {
    if constexpr (is_padded<T>::value) v = row[gen ? a.gen_col0 + k : int(a.load_col[k])];
    else v = row[gen ? 2 * T::ND + T::NDES + k : col_load];
  } else {
    const int t0 = (a.aux_index ? int(a.aux_index[env]) : int(row[a.state_dim - 1])) + 1;
    v = a.series[(gen ? nl + k : k) * a.period + (t0 + stage) % a.period];
  }
  return v / a.base;
}

// per-environment I/O of one solve (device pointers; row-major)
struct IO {
  const double* p_load;  // [E][N][NL]  forecasts, p.u.
  const double* p_gen;   // [E][N][NG]
  const double* soc0;    // [E][NS]
  double* u0;            // [E][NC]    first-stage [P_gen.., P_des..], p.u.
  double* objective;     // [E]
  int32_t* iters;        // [E]
  double* info;          // [E][3]     final mu; STATUS: 0 ok, 1 a stage has no interior start, 2 an angle row left out of
                         //            the solve is violated by the solution; largest dual residual   (may be null)
  double* solution;      // [E][N][NV] P_g, p_c, d, t per stage         (may be null)
  double* trace;         // [E][max_iter + 1][12] per iteration: mu, 0, dual residual, objective, then (of the
                         //   step taken from there) primal and dual step length, centring, mu of the predictor, and the row
                         //   that limits the primal step: stage, row, 0, 0   (may be null)
  Act act;               // mode 0 unless the solve is an act() (then p_load / p_gen are not read)
  int nl, ng;            // size classes: the network's own numbers of loads / generators (rows of the arrays above: nl, ng, ng + NS wide)
};

// ------------------------------------------------------------------------------------------------------------
// host: DC reduction (gym_anm_amd/agents/dcopf.py restated; tests compare the two)
// ------------------------------------------------------------------------------------------------------------
inline bool invert(std::vector<double>& a, int n) {  // Gauss-Jordan with partial pivoting, in place
  std::vector<double> inv(size_t(n) * n, 0.0);
  for (int i = 0; i < n; ++i) inv[size_t(i) * n + i] = 1.0;
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(a[size_t(r) * n + c]) > std::fabs(a[size_t(p) * n + c])) p = r;
    if (!(std::fabs(a[size_t(p) * n + c]) > 1e-300)) return false;
    if (p != c)
      for (int k = 0; k < n; ++k) {
        std::swap(a[size_t(p) * n + k], a[size_t(c) * n + k]);
        std::swap(inv[size_t(p) * n + k], inv[size_t(c) * n + k]);
      }
    const double piv = 1.0 / a[size_t(c) * n + c];
    for (int k = 0; k < n; ++k) {
      a[size_t(c) * n + k] *= piv;
      inv[size_t(c) * n + k] *= piv;
    }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = a[size_t(r) * n + c];
      if (f == 0.0) continue;
      for (int k = 0; k < n; ++k) {
        a[size_t(r) * n + k] -= f * a[size_t(c) * n + k];
        inv[size_t(r) * n + k] -= f * inv[size_t(c) * n + k];
      }
    }
  }
  a.swap(inv);
  return true;
}

template <class T>
inline bool build_tables(const anm_network_desc& d, double gamma, double safety_margin, int N, std::vector<double>& tab,
                         std::string& err, double* theta_bound = nullptr) {
  typedef Sz<T> S;
  // the network's own sizes: those of T for a topology, at most those of T for a size class (the rest of the tables
  // stays zero: padded loads / generators / branches change nothing, see is_padded)
  const int nb = d.n_bus, nd = d.n_dev, nbr = d.n_branch;
  int n_load = 0, n_gen = 0, n_des = 0;
  for (int k = 0; k < nd; ++k) {
    const int ty = d.dev_type[k];
    n_load += ty == DEV_LOAD;
    n_gen += ty == DEV_CLASSICAL || ty == DEV_RENEWABLE;
    n_des += ty == DEV_STORAGE;
  }
  if (nb - 1 > S::NB1 || nbr > S::NBR || n_load > S::NL || n_gen > S::NG || n_des != S::NS) {
    err = "the network does not fit the sizes this MPC kernel was compiled for (its number of storage units must match)";
    return false;
  }
  const int nb1 = nb - 1;
  if (N < 1 || N > 64) { err = "planning_steps must be in [1, 64] (one lane per stage)"; return false; }
  // B = Im(Y_bus) (mpc.py:113), Y as in simulator.py:189-197
  std::vector<double> B(size_t(nb) * nb, 0.0);
  for (int b = 0; b < nbr; ++b) {
    const int f = d.br_from[b], t = d.br_to[b];
    const std::complex<double> ys(d.br_series[2 * b], d.br_series[2 * b + 1]), sh(d.br_shunt[2 * b], d.br_shunt[2 * b + 1]),
        tap(d.br_tap[2 * b], d.br_tap[2 * b + 1]);
    B[size_t(f) * nb + t] = (-ys / std::conj(tap)).imag();
    B[size_t(t) * nb + f] = (-ys / tap).imag();
    B[size_t(f) * nb + f] += ((ys + sh) / (std::abs(tap) * std::abs(tap))).imag();
    B[size_t(t) * nb + t] += (ys + sh).imag();
  }
  int slack_dev = -1;
  for (int k = 0; k < nd; ++k)
    if (d.dev_type[k] == DEV_SLACK) slack_dev = k;
  if (slack_dev < 0 || slack_dev >= nb) {
    err = "the reference's DC-OPF pins the angle at the slack DEVICE's position (mpc.py:302), which is not a bus here";
    return false;
  }
  const int pin = slack_dev, slack_bus = d.dev_bus[slack_dev];
  std::vector<double> Lap(size_t(nb) * nb, 0.0);
  for (int b = 0; b < nbr; ++b) {  // mpc.py:232-245
    const int f = d.br_from[b], t = d.br_to[b];
    Lap[size_t(f) * nb + f] += B[size_t(f) * nb + t];
    Lap[size_t(f) * nb + t] -= B[size_t(f) * nb + t];
    Lap[size_t(t) * nb + t] += B[size_t(t) * nb + f];
    Lap[size_t(t) * nb + f] -= B[size_t(t) * nb + f];
  }
  std::vector<int> keep;
  for (int b = 0; b < nb; ++b)
    if (b != pin) keep.push_back(b);
  std::vector<double> M(size_t(nb) * nb, 0.0);
  for (int r = 0; r < nb; ++r)
    for (int c = 0; c < nb - 1; ++c) M[size_t(r) * nb + c] = Lap[size_t(r) * nb + keep[c]];
  M[size_t(slack_bus) * nb + (nb - 1)] = -1.0;
  if (!invert(M, nb)) { err = "the DC balance of this network is singular"; return false; }
  // Th[b][k]: angle of bus b per unit injection of device k; sg[k]: slack injection per unit injection of device k
  std::vector<double> Th(size_t(nb) * nd, 0.0), sg(nd, 0.0);
  for (int k = 0; k < nd; ++k) {
    const int bus = d.dev_bus[k];
    for (int c = 0; c < nb - 1; ++c) Th[size_t(keep[c]) * nd + k] = M[size_t(c) * nb + bus];
    sg[k] = M[size_t(nb - 1) * nb + bus];
  }
  int loads[pos(S::NL)], ctrl[pos(S::NC)];
  {
    int il = 0, ig = 0, is = 0;
    for (int k = 0; k < nd; ++k) {
      const int ty = d.dev_type[k];
      if (ty == DEV_LOAD) loads[il++] = k;
      else if (ty == DEV_CLASSICAL || ty == DEV_RENEWABLE) ctrl[ig++] = k;
    }
    for (int k = 0; k < nd; ++k)
      if (d.dev_type[k] == DEV_STORAGE) ctrl[S::NG + is++] = k;
    for (int g = ig; g < S::NG; ++g) ctrl[g] = -1;   // (padding)
    for (int l = il; l < S::NL; ++l) loads[l] = -1;
  }
  tab.assign(S::T_TOTAL, 0.0);
  for (int r = 0; r < nb1; ++r) {
    const int b = keep[r];
    for (int c = 0; c < S::NC; ++c)
      if (ctrl[c] >= 0) tab[S::T_THC + r * S::NC + c] = Th[size_t(b) * nd + ctrl[c]];
    for (int l = 0; l < n_load; ++l) tab[S::T_THL + r * S::NL + l] = Th[size_t(b) * nd + loads[l]];
  }
  for (int e = 0; e < S::NBR; ++e) {
    if (e >= nbr) { tab[S::T_LIM + e] = 1e6; continue; }   // a padded branch: no flow, a rating nothing reaches
    const int f = d.br_from[e], t = d.br_to[e];
    const double bft = B[size_t(f) * nb + t];
    for (int c = 0; c < S::NC; ++c)
      if (ctrl[c] >= 0) tab[S::T_PHC + e * S::NC + c] = bft * (Th[size_t(f) * nd + ctrl[c]] - Th[size_t(t) * nd + ctrl[c]]);
    for (int l = 0; l < n_load; ++l) tab[S::T_PHL + e * S::NL + l] = bft * (Th[size_t(f) * nd + loads[l]] - Th[size_t(t) * nd + loads[l]]);
    tab[S::T_LIM + e] = safety_margin * d.br_rate[e];
  }
  for (int c = 0; c < S::NC; ++c) {
    if (ctrl[c] < 0) continue;
    double cost = sg[ctrl[c]];
    if (c < S::NG && d.dev_type[ctrl[c]] == DEV_CLASSICAL) cost += 1.0;  // mpc.py:304-306
    tab[S::T_COST + c] = cost;
  }
  for (int l = 0; l < n_load; ++l) tab[S::T_SGL + l] = sg[loads[l]];
  for (int g = 0; g < n_gen; ++g) {   // (a padded generator keeps the range [0, 0])
    tab[S::T_GPMIN + g] = d.dev_pmin[ctrl[g]];
    tab[S::T_GPMAX + g] = d.dev_pmax[ctrl[g]];
  }
  for (int j = 0; j < S::NS; ++j) {
    const int k = ctrl[S::NG + j];
    tab[S::T_SPMIN + j] = d.dev_pmin[k];
    tab[S::T_SPMAX + j] = d.dev_pmax[k];
    tab[S::T_SOCMIN + j] = d.dev_soc_min[k];
    tab[S::T_SOCMAX + j] = d.dev_soc_max[k];
    tab[S::T_BC + j] = d.delta_t * d.dev_eff[k];
    tab[S::T_BD + j] = d.delta_t / d.dev_eff[k];
  }
  if (theta_bound) {
    // the largest |angle| any bus can reach while every device stays inside its limits (loads: [P_min, 0])
    double worst = 0.0;
    for (int r = 0; r < nb1; ++r) {
      double a = 0.0;
      for (int c = 0; c < S::NC; ++c)
        if (ctrl[c] >= 0) a += std::fabs(tab[S::T_THC + r * S::NC + c]) * std::fmax(std::fabs(d.dev_pmin[ctrl[c]]), std::fabs(d.dev_pmax[ctrl[c]]));
      for (int l = 0; l < n_load; ++l) a += std::fabs(tab[S::T_THL + r * S::NL + l]) * std::fabs(d.dev_pmin[loads[l]]);
      worst = std::fmax(worst, a);
    }
    *theta_bound = worst;
  }
  tab[S::T_LAMB] = std::fmax(d.lamb, 1e-30);   // (lamb = 0: the overloads cost nothing; the epigraph rows keep an interior)
  double w = 1.0;
  for (int i = 0; i < 64; ++i) {
    // gamma^i (mpc.py:212); a weight of exactly 0 (gamma = 0, or an underflow) would leave the later stages without
    // any cost, and the epigraph multipliers without an interior: such stages get a weight that changes nothing in
    // the first 25 digits of the value
    tab[S::T_WGT + i] = std::fmax(w, 1e-30);
    w *= gamma;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------------------
// the solver, as one lane sees it.  `X` gives the lane its place in the group and the exchanges with the other
// lanes:  stage(), up(v) / down(v) (value of the previous / next stage, 0 outside), sum / min / max over the
// group, scan (inclusive prefix sum over the stages).
// ------------------------------------------------------------------------------------------------------------
template <class T>
struct Lane {
  typedef Sz<T> S;
  static constexpr int NG = S::NG, NS = S::NS, NL = S::NL, NB1 = S::NB1, NBR = S::NBR, NC = S::NC, NA = S::NA, NR = S::NR;

  // variables of the stage and of its rows: slack s, multiplier z, 1/s (of the iteration), ds*dz of the predictor.
  // The iterate is kept strictly inside every row (the start is, the steps stay): s is then the row's true slack and
  // moves with the inputs; nothing else has to be remembered per row.
  double xi[pos(NG)], pc[pos(NS)], d[pos(NS)], t[pos(NBR)];
  double s[NR], z[NR];
  // 1/s and the predictor's ds*dz: on the GPU in LDS (row-major [row][lane] of the wavefront: conflict-free, and 4 NR
  // registers less per lane -- with them the kernel needs more than the 512 a lane can have), registers on the host
  // (1/s only for programs of more than Sz::ROWS_IN_REGISTERS rows per stage)
#if defined(__HIP_DEVICE_COMPILE__)
  static constexpr bool IN_LDS = S::ROWS_IN_LDS, IS_IN_LDS = S::NR > S::ROWS_IN_REGISTERS;
  // (volatile: every use is a ds_read -- otherwise the compiler keeps what it has read in registers, which is what
  // moving the arrays to LDS is meant to avoid)
  volatile __attribute__((address_space(3))) double* lds;   // this lane's column of the wavefront's [2 NR][64] block
#else
  static constexpr bool IN_LDS = false, IS_IN_LDS = false;
#endif
  double is_[IS_IN_LDS ? 1 : NR], cross_[IN_LDS ? 1 : NR];
  ANM_HD double is(int r) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (IS_IN_LDS) return lds[r * 64];
#endif
    return is_[IS_IN_LDS ? 0 : r];
  }
  ANM_HD double cross(int r) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (IN_LDS) return lds[(NR + r) * 64];
#endif
    return cross_[IN_LDS ? 0 : r];
  }
  ANM_HD void set_is(int r, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (IS_IN_LDS) { lds[r * 64] = v; return; }
#endif
    is_[IS_IN_LDS ? 0 : r] = v;
  }
  ANM_HD void set_cross(int r, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (IN_LDS) { lds[(NR + r) * 64] = v; return; }
#endif
    cross_[IN_LDS ? 0 : r] = v;
  }
  // constants of the stage
  double wd[pos(NG)], th0[pos(NB1)], f0[pos(NBR)], wgt, ct, objc;
  // per iteration
  double htt[pos(NBR)], hut[pos(NBR)];
  // factor of the stage: Lc = lower Cholesky factor (row-major triangle, diagonal stored inverted) of
  //   [R_xx R_xs; R_sx R_ss + Bs' P Bs]   over (xi.., p_c.., d..): the xi columns are local (factor_stage, every lane
  //   at once), the (p_c, d) block needs the P of the stage (factor_coupled, last stage first);
  // Ts = R_ss - L_sx L_sx' (what the xi columns leave of the (p_c, d) block), Wm = L_ss^-1 Bs' P
  double Lc[pos(NA * (NA + 1) / 2)];
  double Ts[pos(NS * (2 * NS + 1))], Wm[pos(2 * NS * NS)], Po[pos(NS * NS)];

  static ANM_HD constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // i >= j
  ANM_HD double w(int r) const { return z[r] * is(r); }

  // u = [P_g.., P_des..]
  ANM_HD void phys(cptr_t C, double (&u)[pos(NC)]) const {
    ANM_UFOR (int g = 0; g < NG; ++g) u[g] = fma(wd[g], xi[g], C[S::T_GPMIN + g]);
    ANM_UFOR (int j = 0; j < NS; ++j) u[NG + j] = d[j] - pc[j];
  }

  // g.v - h of every row (sig: the states of charge after the stage)
  ANM_HD void row_values(cptr_t C, const double (&sig)[pos(NS)], double (&val)[NR]) const {
    double u[pos(NC)];
    phys(C, u);
    ANM_UFOR (int g = 0; g < NG; ++g) {
      val[S::R_XI_UP + g] = xi[g] - 1.0;
      val[S::R_XI_LO + g] = -xi[g];
    }
    ANM_UFOR (int j = 0; j < NS; ++j) {
      val[S::R_PD_UP + j] = u[NG + j] - C[S::T_SPMAX + j];
      val[S::R_PD_LO + j] = C[S::T_SPMIN + j] - u[NG + j];
      val[S::R_PC + j] = -pc[j];
      val[S::R_D + j] = -d[j];
      val[S::R_SOC_UP + j] = sig[j] - C[S::T_SOCMAX + j];
      val[S::R_SOC_LO + j] = C[S::T_SOCMIN + j] - sig[j];
    }
    const double PI = 3.14159265358979323846;
    ANM_UFOR (int b = 0; b < S::NTH; ++b) {
      double th = th0[b];
      ANM_UFOR (int c = 0; c < NC; ++c) th = fma(C[S::T_THC + b * NC + c], u[c], th);
      val[S::R_TH_UP + b] = th - PI;
      val[S::R_TH_LO + b] = -th - PI;
    }
    ANM_UFOR (int e = 0; e < NBR; ++e) {
      double fl = f0[e];
      ANM_UFOR (int c = 0; c < NC; ++c) fl = fma(C[S::T_PHC + e * NC + c], u[c], fl);
      const double lim = C[S::T_LIM + e];
      val[S::R_FL1 + e] = fl - t[e] - lim;
      val[S::R_FL2 + e] = -fl - t[e] - lim;
      val[S::R_FL3 + e] = -t[e];
    }
  }

  // stage part of the objective (the reference's expression, mpc.py:304-313, with its constants)
  ANM_HD double objective(cptr_t C) const {
    double u[pos(NC)];
    phys(C, u);
    double o = objc;
    ANM_UFOR (int c = 0; c < NC; ++c) o = fma(wgt * C[S::T_COST + c], u[c], o);
    ANM_UFOR (int e = 0; e < NBR; ++e) o = fma(ct, t[e], o);
    return o;
  }

  // gradient of the Lagrangian for multipliers zh(r): inputs (xi, p_c, d) and state; with `eliminate` the epigraph
  // variables' share is folded into the flows (right-hand sides of a Newton step); their own gradient in gt
  template <class ZH>
  ANM_HD void gradient(cptr_t C, ZH&& zh, bool eliminate, double (&g_in)[pos(NA)], double (&g_st)[pos(NS)],
                       double (&gt)[pos(NBR)]) const {
    double gu[pos(NC)];
    ANM_UFOR (int c = 0; c < NC; ++c) gu[c] = wgt * C[S::T_COST + c];
    ANM_UFOR (int b = 0; b < S::NTH; ++b) {
      const double dz = zh(S::R_TH_UP + b) - zh(S::R_TH_LO + b);
      ANM_UFOR (int c = 0; c < NC; ++c) gu[c] = fma(dz, C[S::T_THC + b * NC + c], gu[c]);
    }
    ANM_UFOR (int e = 0; e < NBR; ++e) {
      const double z1 = zh(S::R_FL1 + e), z2 = zh(S::R_FL2 + e);
      gt[e] = ct - z1 - z2 - zh(S::R_FL3 + e);
      double ga = z1 - z2;
      if (eliminate) ga = fma(-hut[e], gt[e], ga);
      ANM_UFOR (int c = 0; c < NC; ++c) gu[c] = fma(ga, C[S::T_PHC + e * NC + c], gu[c]);
    }
    ANM_UFOR (int j = 0; j < NS; ++j) gu[NG + j] += zh(S::R_PD_UP + j) - zh(S::R_PD_LO + j);
    ANM_UFOR (int g = 0; g < NG; ++g) g_in[g] = fma(wd[g], gu[g], zh(S::R_XI_UP + g) - zh(S::R_XI_LO + g));
    ANM_UFOR (int j = 0; j < NS; ++j) {
      g_in[NG + j] = -gu[NG + j] - zh(S::R_PC + j);
      g_in[NG + NS + j] = gu[NG + j] - zh(S::R_D + j);
      g_st[j] = zh(S::R_SOC_UP + j) - zh(S::R_SOC_LO + j);
    }
  }

  // R = J' Hu J + (weights of the direct rows) + rho I: the xi columns of its Cholesky factor and what they leave
  // of the (p_c, d) block.  rho = a few units of roundoff of the largest diagonal entry: it keeps R numerically
  // positive definite when the weights of a stage span more than 1/eps (directions only inactive rows see, next to
  // rows about to become active); anything larger freezes those directions and leaves a dual residual rho |dv|
  ANM_HD void factor_stage(cptr_t C) {
    double Hu[pos(NC * (NC + 1) / 2)];
    ANM_UFOR (int k = 0; k < NC * (NC + 1) / 2; ++k) Hu[k] = 0.0;
    ANM_UFOR (int b = 0; b < S::NTH; ++b) {
      const double wb = w(S::R_TH_UP + b) + w(S::R_TH_LO + b);
      ANM_UFOR (int i = 0; i < NC; ++i) {
        const double wi = wb * C[S::T_THC + b * NC + i];
        ANM_UFOR (int j = 0; j <= i; ++j) Hu[tri(i, j)] = fma(wi, C[S::T_THC + b * NC + j], Hu[tri(i, j)]);
      }
    }
    ANM_UFOR (int e = 0; e < NBR; ++e) {
      const double w1 = w(S::R_FL1 + e), w2 = w(S::R_FL2 + e), w3 = w(S::R_FL3 + e);
      const double ih = recip(w1 + w2 + w3);
      htt[e] = ih;                          // (the reciprocal is what every later use needs)
      hut[e] = (w2 - w1) * ih;
      const double hat = (4.0 * w1 * w2 + w3 * (w1 + w2)) * ih;  // weight of (a.du)^2 with t_e eliminated
      ANM_UFOR (int i = 0; i < NC; ++i) {
        const double wi = hat * C[S::T_PHC + e * NC + i];
        ANM_UFOR (int j = 0; j <= i; ++j) Hu[tri(i, j)] = fma(wi, C[S::T_PHC + e * NC + j], Hu[tri(i, j)]);
      }
    }
    ANM_UFOR (int j = 0; j < NS; ++j) Hu[tri(NG + j, NG + j)] += w(S::R_PD_UP + j) + w(S::R_PD_LO + j);
    // R over (xi.., pc.., d..):  u_g = wd_g xi_g,  u_des_j = d_j - pc_j
    double R[pos(NA * (NA + 1) / 2)];
    auto ju = [&](int a, int& c, double& f) {  // input a moves u_c by f
      if (a < NG) { c = a; f = wd[a < NG ? a : 0]; }
      else if (a < NG + NS) { c = a; f = -1.0; }
      else { c = a - NS; f = 1.0; }
    };
    ANM_UFOR (int a = 0; a < NA; ++a)
      ANM_UFOR (int b = 0; b <= a; ++b) {
        int ca, cb;
        double fa, fb;
        ju(a, ca, fa);
        ju(b, cb, fb);
        R[tri(a, b)] = fa * fb * (ca >= cb ? Hu[tri(ca, cb)] : Hu[tri(cb, ca)]);
      }
    ANM_UFOR (int g = 0; g < NG; ++g) R[tri(g, g)] += w(S::R_XI_UP + g) + w(S::R_XI_LO + g);
    ANM_UFOR (int j = 0; j < NS; ++j) {
      R[tri(NG + j, NG + j)] += w(S::R_PC + j);
      R[tri(NG + NS + j, NG + NS + j)] += w(S::R_D + j);
    }
    double dmax = 0.0;
    ANM_UFOR (int a = 0; a < NA; ++a) dmax = fmax(dmax, R[tri(a, a)]);
    const double rho = fma(4e-16, dmax, 1e-14);
    ANM_UFOR (int a = 0; a < NA; ++a) R[tri(a, a)] += rho;
    ANM_UFOR (int jc = 0; jc < NG; ++jc)
      ANM_UFOR (int ir = jc; ir < NA; ++ir) {
        double acc = R[tri(ir, jc)];
        ANM_UFOR (int k = 0; k < jc; ++k) acc = fma(-Lc[tri(ir, k)], Lc[tri(jc, k)], acc);
        if (ir == jc) Lc[tri(jc, jc)] = 1.0 / sqrt(acc);
        else Lc[tri(ir, jc)] = acc * Lc[tri(jc, jc)];
      }
    ANM_UFOR (int a = 0; a < 2 * NS; ++a)
      ANM_UFOR (int b = 0; b <= a; ++b) {
        double acc = R[tri(NG + a, NG + b)];
        ANM_UFOR (int k = 0; k < NG; ++k) acc = fma(-Lc[tri(NG + a, k)], Lc[tri(NG + b, k)], acc);
        Ts[tri(a, b)] = acc;
      }
  }

  // coefficient of input s (0..NS-1: p_c, NS..2NS-1: d) in the state equation of its storage unit
  static ANM_HD double bcoef(cptr_t C, int a) { return a < NS ? C[S::T_BC + (a < NS ? a : 0)] : -C[S::T_BD + (a < NS ? 0 : a - NS)]; }

  // The coupled part of the factor, for the value function  V(x) = 1/2 x' P x  of the state AFTER this stage:
  //   Lam = Ts + Bs' P Bs = L_ss L_ss',   Wm = L_ss^-1 Bs' P,   Po = P - Wm' Wm
  // (one block step of the Cholesky factorisation of the whole Newton matrix, states eliminated, last stage first:
  // every quantity is a sum of positive terms or what a Cholesky step subtracts -- nothing is inverted on its own,
  // so the weights of the rows may span the whole double range, like in a dense factorisation)
  ANM_HD void factor_coupled(cptr_t C, const double (&P)[pos(NS * NS)]) {
    double cm[pos(2 * NS * NS)];  // Bs' P
    ANM_UFOR (int a = 0; a < 2 * NS; ++a)
      ANM_UFOR (int q = 0; q < NS; ++q) cm[a * NS + q] = bcoef(C, a) * P[(a % pos(NS)) * NS + q];
    ANM_UFOR (int a = 0; a < 2 * NS; ++a)
      ANM_UFOR (int b = 0; b <= a; ++b) {
        double acc = fma(cm[a * NS + (b % pos(NS))], bcoef(C, b), Ts[tri(a, b)]);
        ANM_UFOR (int k = 0; k < b; ++k) acc = fma(-Lc[tri(NG + a, NG + k)], Lc[tri(NG + b, NG + k)], acc);
        if (a == b) Lc[tri(NG + a, NG + a)] = 1.0 / sqrt(acc);
        else Lc[tri(NG + a, NG + b)] = acc * Lc[tri(NG + b, NG + b)];
      }
    ANM_UFOR (int q = 0; q < NS; ++q)
      ANM_UFOR (int a = 0; a < 2 * NS; ++a) {
        double acc = cm[a * NS + q];
        ANM_UFOR (int k = 0; k < a; ++k) acc = fma(-Lc[tri(NG + a, NG + k)], Wm[k * NS + q], acc);
        Wm[a * NS + q] = acc * Lc[tri(NG + a, NG + a)];
      }
    ANM_UFOR (int q = 0; q < NS; ++q)
      ANM_UFOR (int r = 0; r < NS; ++r) {
        double acc = P[q * NS + r];
        ANM_UFOR (int a = 0; a < 2 * NS; ++a) acc = fma(-Wm[a * NS + q], Wm[a * NS + r], acc);
        Po[q * NS + r] = acc;
      }
    ANM_UFOR (int q = 0; q < NS; ++q) Po[q * NS + q] = fmax(Po[q * NS + q], 0.0);
  }
};

// "this value may have changed": the compiler must not keep what it computed from it before (see Step::forget)
ANM_HD void forget(double& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#else
  (void)v;
#endif
}

// a Newton step of one stage, and what it does to the rows
template <class T>
struct Step {
  typedef Sz<T> S;
  double dxi[pos(S::NG)], dpc[pos(S::NS)], dd[pos(S::NS)], dt[pos(S::NBR)], xs[pos(S::NS)], gt[pos(S::NBR)];

  // The rows are visited twice with the same step (once for the step length, once, after the group's reduction,
  // for the update), and recomputing g.dv, dz and the target multiplier of a row costs a handful of instructions.
  // Left alone, the compiler keeps all of them from the first visit to the second instead: 6 registers per row,
  // more than a lane has.  Passing the step through an empty asm between the visits makes the second one recompute.
  ANM_HD void forget_all() {
    ANM_UFOR (int g = 0; g < S::NG; ++g) forget(dxi[g]);
    ANM_UFOR (int j = 0; j < S::NS; ++j) { forget(dpc[j]); forget(dd[j]); forget(xs[j]); }
    ANM_UFOR (int e = 0; e < S::NBR; ++e) { forget(dt[e]); forget(gt[e]); }
  }

  // f(r, g_r . dv) for every row r of the stage, in row order
  template <class F>
  ANM_HD void rows(cptr_t C, const Lane<T>& ln, F&& f) const {
    constexpr int NG = S::NG, NS = S::NS, NB1 = S::NB1, NBR = S::NBR, NC = S::NC;
    double du[pos(NC)];
    ANM_UFOR (int g = 0; g < NG; ++g) du[g] = ln.wd[g] * dxi[g];
    ANM_UFOR (int j = 0; j < NS; ++j) du[NG + j] = dd[j] - dpc[j];
    ANM_UFOR (int g = 0; g < NG; ++g) { f(S::R_XI_UP + g, dxi[g]); f(S::R_XI_LO + g, -dxi[g]); }
    ANM_UFOR (int j = 0; j < NS; ++j) {
      f(S::R_PD_UP + j, du[NG + j]);
      f(S::R_PD_LO + j, -du[NG + j]);
      f(S::R_PC + j, -dpc[j]);
      f(S::R_D + j, -dd[j]);
      f(S::R_SOC_UP + j, xs[j]);
      f(S::R_SOC_LO + j, -xs[j]);
    }
    ANM_UFOR (int b = 0; b < S::NTH; ++b) {
      double a = 0.0;
      ANM_UFOR (int c = 0; c < NC; ++c) a = fma(C[S::T_THC + b * NC + c], du[c], a);
      f(S::R_TH_UP + b, a);
      f(S::R_TH_LO + b, -a);
    }
    ANM_UFOR (int e = 0; e < NBR; ++e) {
      double al = 0.0;
      ANM_UFOR (int c = 0; c < NC; ++c) al = fma(C[S::T_PHC + e * NC + c], du[c], al);
      const double rt = -gt[e], ih = ln.htt[e];
      const double w1 = ln.w(S::R_FL1 + e), w2 = ln.w(S::R_FL2 + e), w3 = ln.w(S::R_FL3 + e);
      // al - dt and -al - dt without the cancellation when one weight dominates
      f(S::R_FL1 + e, (al * (2.0 * w2 + w3) - rt) * ih);
      f(S::R_FL2 + e, (-al * (2.0 * w1 + w3) - rt) * ih);
      f(S::R_FL3 + e, -dt[e]);
    }
  }
};

// One lane's run of the whole solve.
template <class T, class X>
ANM_HD void solve(cptr_t C, const IO& io, const Opts& opt, int64_t env, bool valid, int N, X& x, double* lane_lds = nullptr) {
  typedef Sz<T> S;
  typedef Lane<T> L;
  constexpr int NG = S::NG, NS = S::NS, NL = S::NL, NB1 = S::NB1, NBR = S::NBR, NC = S::NC, NA = S::NA, NR = S::NR;
  const int i = x.stage();
  const bool on = valid && i < N;  // a lane without a stage runs along with neutral contributions
  L ln;
#if defined(__HIP_DEVICE_COMPILE__)
  ln.lds = (volatile __attribute__((address_space(3))) double*)lane_lds;
#else
  (void)lane_lds;
#endif
  ANM_UFOR (int k = 0; k < pos(NS * NS); ++k) ln.Po[k] = 0.0;
  ANM_UFOR (int k = 0; k < pos(2 * NS * NS); ++k) ln.Wm[k] = 0.0;
  ANM_UFOR (int k = 0; k < pos(NA * (NA + 1) / 2); ++k) ln.Lc[k] = 0.0;
  // ---- constants of the stage ----
  ln.wgt = C[S::T_WGT + (i < 64 ? i : 63)];
  ln.ct = ln.wgt * C[S::T_LAMB];
  double soc0[pos(NS)];
  {
    double pl[pos(NL)];
    constexpr bool PAD = is_padded<T>::value;
    const int nl = PAD ? io.nl : NL, ng = PAD ? io.ng : NG;   // (a topology's kernel: compile-time constants)
    ANM_UFOR (int l = 0; l < NL; ++l)
      pl[l] = (!on || l >= nl) ? 0.0 : (io.act.mode ? act_forecast<T>(io.act, env, i, false, l, load_device<T>(l), nl)
                                                    : io.p_load[(env * N + i) * nl + l]);
    ANM_UFOR (int g = 0; g < NG; ++g) {
      const double fc = (!on || g >= ng) ? 0.0 : (io.act.mode ? act_forecast<T>(io.act, env, i, true, g, 0, nl) : io.p_gen[(env * N + i) * ng + g]);
      ln.wd[g] = fmax(fmin(C[S::T_GPMAX + g], fc) - C[S::T_GPMIN + g], 0.0);
    }
    ANM_UFOR (int j = 0; j < NS; ++j) {  // (a state of charge outside its window -- not a state the simulator produces -- is moved onto it)
      const double raw = valid ? io.soc0[env * NS + j] : 0.5 * (C[S::T_SOCMIN + j] + C[S::T_SOCMAX + j]);
      soc0[j] = fmin(fmax(raw, C[S::T_SOCMIN + j]), C[S::T_SOCMAX + j]);
    }
    ANM_UFOR (int b = 0; b < NB1; ++b) {
      double a = 0.0;
      ANM_UFOR (int l = 0; l < NL; ++l) a = fma(C[S::T_THL + b * NL + l], pl[l], a);
      ln.th0[b] = a;
    }
    ANM_UFOR (int e = 0; e < NBR; ++e) {
      double a = 0.0;
      ANM_UFOR (int l = 0; l < NL; ++l) a = fma(C[S::T_PHL + e * NL + l], pl[l], a);
      ln.f0[e] = a;
    }
    double a = 0.0;
    ANM_UFOR (int l = 0; l < NL; ++l) a = fma(C[S::T_SGL + l], pl[l], a);
    ln.objc = ln.wgt * a;
  }
  // ---- start: strictly inside every row -- mid-interval generators; a little charge and the discharge that undoes
  // it, plus what moves the state of charge a fifth of the way towards the middle of its window over the horizon
  // (a state of charge AT a bound is the usual case: an empty or a full unit); epigraph variables above the
  // overloads of that point.  The slacks are the rows' own: from here on they move with the inputs. ----
  bool inside = true;
  {
    double sig[pos(NS)];
    // The angle rows |theta| <= pi are the only ones such a point may violate (networks whose per-unit injections
    // are tens of p.u.): the whole start is then scaled towards "generators at P_min, storage idle", halving until
    // every stage of the environment is inside (20 times at most: then only the loads are left)
    double f = 1.0;
    for (int pass = 0; pass < 20; ++pass) {
      ANM_UFOR (int g = 0; g < NG; ++g) ln.xi[g] = 0.5 * f;
      ANM_UFOR (int j = 0; j < NS; ++j) {
        const double bc = C[S::T_BC + j], bd = C[S::T_BD + j], pmax = C[S::T_SPMAX + j], pmin = C[S::T_SPMIN + j];
        const double base = 0.05 * f * (pmax - pmin);
        const double mid = 0.5 * (C[S::T_SOCMIN + j] + C[S::T_SOCMAX + j]);
        double dsig = 0.2 * (mid - soc0[j]) / double(N);
        dsig = f * fmin(fmax(dsig, -0.25 * pmax * bd), -0.25 * pmin * bc);   // (a quarter of the power limits at most)
        ln.pc[j] = base + fmax(dsig, 0.0) / bc;
        ln.d[j] = base * (bc / bd) + fmax(-dsig, 0.0) / bd;
        sig[j] = soc0[j] + double(i + 1) * dsig;
      }
      double u[pos(NC)];
      ln.phys(C, u);
      ANM_UFOR (int e = 0; e < NBR; ++e) {
        double fl = ln.f0[e];
        ANM_UFOR (int c = 0; c < NC; ++c) fl = fma(C[S::T_PHC + e * NC + c], u[c], fl);
        ln.t[e] = fmax(fabs(fl) - C[S::T_LIM + e], 0.0) + 0.1;
      }
      double val[NR];
      ln.row_values(C, sig, val);
      inside = true;
      ANM_UFOR (int r = 0; r < NR; ++r) {
        inside = inside && (-val[r] > 0.0);
        ln.s[r] = -val[r] > 0.0 ? -val[r] : 1.0;
      }
      if (x.min(on ? (inside ? 1.0 : 0.0) : 1.0) > 0.5) break;
      f *= 0.5;
    }
    ANM_UFOR (int r = 0; r < NR; ++r) {
      ln.z[r] = r >= S::R_FL1 ? ln.ct * (1.0 / 3.0) : 1.0;  // dual feasible for the epigraph variables
      ln.set_cross(r, 0.0);
    }
  }
  // (no interior start: a storage unit that cannot both charge and discharge, an angle limit the loads alone
  // violate ... -- reported as not converged)
  const bool startable = x.min(on ? (inside ? 1.0 : 0.0) : 1.0) > 0.5;
  const double m_rows = double(NR) * double(N);
  int it = 0;
  bool done = !valid;
  double mu = 0.0, rdmax = 0.0, obj = 0.0;
  for (;; ++it) {
#if defined(__HIP_DEVICE_COMPILE__)
    // the network tables are re-read (scalar loads, scalar cache) in every iteration: kept across iterations they
    // would occupy some 150 registers of every lane
    asm volatile("" ::: "memory");
#endif
    // ---- where the iterate stands ----
    double a_mu = 0.0, a_rd = 0.0;
    ANM_UFOR (int r = 0; r < NR; ++r) a_mu = fma(ln.s[r], ln.z[r], a_mu);
    {  // dual residual: gradient of the Lagrangian; the window rows of the stages from here on act on this stage's
       // inputs through B
      double g_in[pos(NA)], g_st[pos(NS)], gt[pos(NBR)];
      ln.gradient(C, [&](int r) { return ln.z[r]; }, false, g_in, g_st, gt);
      ANM_UFOR (int j = 0; j < NS; ++j) {
        const double mine = on ? g_st[j] : 0.0;
        const double lam = x.sum(mine) - x.scan(mine) + mine;
        g_in[NG + j] = fma(C[S::T_BC + j], lam, g_in[NG + j]);
        g_in[NG + NS + j] = fma(-C[S::T_BD + j], lam, g_in[NG + NS + j]);
      }
      ANM_UFOR (int a = 0; a < NA; ++a) a_rd = fmax(a_rd, fabs(g_in[a]));
      ANM_UFOR (int e = 0; e < NBR; ++e) a_rd = fmax(a_rd, fabs(gt[e]));
    }
    const double n_mu = x.sum(on ? a_mu : 0.0) / m_rows, n_rd = x.max(on ? a_rd : 0.0);
    const double n_obj = x.sum(on ? ln.objective(C) : 0.0);
    if (!done) { mu = n_mu; rdmax = n_rd; obj = n_obj; }
    if (!done && (mu <= opt.tol * (1.0 + fabs(obj)) || it >= opt.max_iter || !(mu == mu) || !startable)) {
      done = true;
      bool relaxed_ok = true;
      if constexpr (S::NTH == 0 && NB1 > 0) {   // the angle rows were left out: the solution must satisfy them
        double u[pos(NC)];
        ln.phys(C, u);
        bool viol = false;
        ANM_UFOR (int b = 0; b < NB1; ++b) {
          double th = ln.th0[b];
          ANM_UFOR (int c = 0; c < NC; ++c) th = fma(C[S::T_THC + b * NC + c], u[c], th);
          viol = viol || !(fabs(th) <= 3.14159265358979323846);
        }
        relaxed_ok = x.max(on && viol ? 1.0 : 0.0) < 0.5;
      }
      if (valid && i == 0) {
        io.objective[env] = obj;
        io.iters[env] = (startable && relaxed_ok) ? it : opt.max_iter;
        if (io.info) { io.info[env * 3] = mu; io.info[env * 3 + 1] = !startable ? 1.0 : (relaxed_ok ? 0.0 : 2.0); io.info[env * 3 + 2] = rdmax; }
        double u[pos(NC)];
        ln.phys(C, u);
        const int ng = is_padded<T>::value ? io.ng : NG;   // the rows of u0 / action are as wide as the NETWORK's own counts
        ANM_UFOR (int g = 0; g < NG; ++g)
          if (g < ng) io.u0[env * (ng + NS) + g] = u[g];
        ANM_UFOR (int j = 0; j < NS; ++j) io.u0[env * (ng + NS) + ng + j] = u[NG + j];
        if (io.act.mode && io.act.action) {   // [P_gen.., Q_gen.. = 0, P_des.., Q_des.. = 0] MW, clipped (mpc.py:341-344, 383-388)
          double* a = io.act.action + env * (2 * ng + 2 * NS);
          auto clip = [&](int k, double v) { a[k] = fmin(fmax(v, io.act.act_lo[k]), io.act.act_hi[k]); };
          ANM_UFOR (int g = 0; g < NG; ++g)
            if (g < ng) { clip(g, u[g] * io.act.base); clip(ng + g, 0.0); }
          ANM_UFOR (int j = 0; j < NS; ++j) { clip(2 * ng + j, u[NG + j] * io.act.base); clip(2 * ng + NS + j, 0.0); }
        }
      }
      if (on && io.solution) {
        double u[pos(NC)];
        ln.phys(C, u);
        double* o = io.solution + (env * N + i) * S::NV;
        ANM_UFOR (int g = 0; g < NG; ++g) o[g] = u[g];
        ANM_UFOR (int j = 0; j < NS; ++j) { o[NG + j] = ln.pc[j]; o[NG + NS + j] = ln.d[j]; }
        ANM_UFOR (int e = 0; e < NBR; ++e) o[NA + e] = ln.t[e];
      }
    }
    double* trg = (io.trace && on && !done && it <= opt.max_iter) ? io.trace + (env * (opt.max_iter + 1) + it) * 12 : nullptr;
    double* tr = i == 0 ? trg : nullptr;
    if (tr) { tr[0] = n_mu; tr[1] = 0.0; tr[2] = n_rd; tr[3] = n_obj; }
    if (x.all_done(done)) break;
    // ---- factor (once per iteration) ----
    ANM_UFOR (int r = 0; r < NR; ++r) ln.set_is(r, recip(ln.s[r]));
    ln.factor_stage(C);
    {  // value functions, last stage first:  P_i = Q_i + P'_{i+1}  (Q: the weights of the window rows).  Systolic:
       // in every step EVERY lane refactors with what its successor handed down last; stage k has its final input
       // from step N - 1 - k on (the lanes of a wavefront execute the step anyway)
      double Pn[pos(NS * NS)];
      ANM_UFOR (int k = 0; k < NS * NS; ++k) Pn[k] = 0.0;
      for (int k = N - 1; k >= 0; --k) {
        double P[pos(NS * NS)];
        ANM_UFOR (int a = 0; a < NS; ++a)
          ANM_UFOR (int b = 0; b < NS; ++b) P[a * NS + b] = Pn[a * NS + b] + (a == b ? ln.w(S::R_SOC_UP + a) + ln.w(S::R_SOC_LO + a) : 0.0);
        ln.factor_coupled(C, P);
        ANM_UFOR (int q = 0; q < NS * NS; ++q) Pn[q] = x.down(i < N ? ln.Po[q] : 0.0);  // (the last stage receives 0)
      }
    }
    // ---- Newton step for multipliers zh(r):  H dv = -(c + G' zh) ----
    Step<T> st;
    auto newton = [&](auto&& zh) {
      double g_in[pos(NA)], g_st[pos(NS)];
      ln.gradient(C, zh, true, g_in, g_st, st.gt);
      // Residual form.  The multipliers of the window rows of this and the later stages act on the inputs of this
      // stage through B: lam_i = sum_{k >= i} g_st_k is the costate the multipliers THEMSELVES imply.  With it
      // folded into the input gradient the sweeps only have to find the CORRECTION of the costate.
      ANM_UFOR (int j = 0; j < NS; ++j) {
        const double mine = on ? g_st[j] : 0.0;
        const double lam = x.sum(mine) - x.scan(mine) + mine;
        g_in[NG + j] = fma(C[S::T_BC + j], lam, g_in[NG + j]);
        g_in[NG + NS + j] = fma(-C[S::T_BD + j], lam, g_in[NG + NS + j]);
      }
      // forward substitution: the xi part at once, the (p_c, d) part stage by stage (last first), each stage
      // handing the linear term of its value function down:  y_s = L_ss^-1 (g_s - L_sx y_x + Bs' p),  p' = p - Wm' y_s
      double y[pos(NA)];
      ANM_UFOR (int a = 0; a < NA; ++a) {
        double acc = g_in[a];
        ANM_UFOR (int k = 0; k < (a < NG ? a : NG); ++k) acc = fma(-ln.Lc[L::tri(a, k)], y[k], acc);
        y[a] = a < NG ? acc * ln.Lc[L::tri(a, a)] : acc;
      }
      double pn[pos(NS)], ys[pos(2 * NS)];
      ANM_UFOR (int j = 0; j < NS; ++j) pn[j] = 0.0;
      ANM_UFOR (int a = 0; a < 2 * NS; ++a) ys[a] = 0.0;
      for (int k = N - 1; k >= 0; --k) {  // (systolic, like the factor sweep)
        ANM_UFOR (int a = 0; a < 2 * NS; ++a) {
          double acc = fma(L::bcoef(C, a), pn[a % pos(NS)], y[NG + a]);
          ANM_UFOR (int kk = 0; kk < a; ++kk) acc = fma(-ln.Lc[L::tri(NG + a, NG + kk)], ys[kk], acc);
          ys[a] = acc * ln.Lc[L::tri(NG + a, NG + a)];
        }
        ANM_UFOR (int q = 0; q < NS; ++q) {
          double acc = pn[q];
          ANM_UFOR (int a = 0; a < 2 * NS; ++a) acc = fma(-ln.Wm[a * NS + q], ys[a], acc);
          pn[q] = x.down(i < N ? acc : 0.0);
        }
      }
      ANM_UFOR (int a = 0; a < 2 * NS; ++a) y[NG + a] = ys[a];
      // back substitution, first stage first:  a_s = -L_ss^-T (y_s + Wm x_{i-1}),  x_i = x_{i-1} + Bs a_s  -- the
      // state step IS what the inputs add up to; then the xi part
      double da[pos(NA)], xp[pos(NS)];
      ANM_UFOR (int a = 0; a < NA; ++a) da[a] = 0.0;
      ANM_UFOR (int j = 0; j < NS; ++j) { st.xs[j] = 0.0; xp[j] = 0.0; }
      for (int k = 0; k < N; ++k) {
        ANM_UFOR (int a = 2 * NS - 1; a >= 0; --a) {
          double acc = y[NG + a];
          ANM_UFOR (int q = 0; q < NS; ++q) acc = fma(ln.Wm[a * NS + q], xp[q], acc);
          ANM_UFOR (int kk = a + 1; kk < 2 * NS; ++kk) acc = fma(ln.Lc[L::tri(NG + kk, NG + a)], da[NG + kk], acc);
          da[NG + a] = -acc * ln.Lc[L::tri(NG + a, NG + a)];
        }
        ANM_UFOR (int j = 0; j < NS; ++j) {
          st.xs[j] = xp[j] + fma(C[S::T_BC + j], da[NG + j], -C[S::T_BD + j] * da[NG + NS + j]);
          xp[j] = x.up(i < N ? st.xs[j] : 0.0);
        }
      }
      // (xp is now the state step of the previous stage; the last pass used the final one)
      ANM_UFOR (int a = NG - 1; a >= 0; --a) {
        double acc = y[a];
        ANM_UFOR (int kk = a + 1; kk < NA; ++kk) acc = fma(ln.Lc[L::tri(kk, a)], da[kk], acc);
        da[a] = -acc * ln.Lc[L::tri(a, a)];
      }
      ANM_UFOR (int g = 0; g < NG; ++g) st.dxi[g] = da[g];
      ANM_UFOR (int j = 0; j < NS; ++j) { st.dpc[j] = da[NG + j]; st.dd[j] = da[NG + NS + j]; }
      {  // the epigraph variables follow their flows
        double du[pos(NC)];
        ANM_UFOR (int g = 0; g < NG; ++g) du[g] = ln.wd[g] * st.dxi[g];
        ANM_UFOR (int j = 0; j < NS; ++j) du[NG + j] = st.dd[j] - st.dpc[j];
        ANM_UFOR (int e = 0; e < NBR; ++e) {
          double al = 0.0;
          ANM_UFOR (int c = 0; c < NC; ++c) al = fma(C[S::T_PHC + e * NC + c], du[c], al);
          st.dt[e] = fma(-st.gt[e], ln.htt[e], -ln.hut[e] * al);
        }
      }
    };
    // ---- predictor:  s dz + z ds = -s z  ->  zh = 0:  ds = -g.dv,  dz = -z + w g.dv ----
    newton([](int) { return 0.0; });
    double ms = 0.0, mz = 0.0, sdz = 0.0, zds = 0.0, dsdz = 0.0;  // largest -ds/s, -dz/z; sums for the predictor's mu
    st.rows(C, ln, [&](int r, double gv) {
      const double ds = -gv, dz = fma(ln.w(r), gv, -ln.z[r]);
      ms = fmax(ms, -ds * ln.is(r));
      mz = fmax(mz, -dz * recip(ln.z[r]));
      sdz = fma(ln.s[r], dz, sdz);
      zds = fma(ln.z[r], ds, zds);
      dsdz = fma(ds, dz, dsdz);
      ln.set_cross(r, ds * dz);
    });
    ms = x.max(on ? ms : 0.0);
    mz = x.max(on ? mz : 0.0);
    double ap = ms > 1.0 ? 1.0 / ms : 1.0, ad = mz > 1.0 ? 1.0 / mz : 1.0;
    const double mu_aff = x.sum(on ? a_mu + ap * zds + ad * sdz + ap * ad * dsdz : 0.0) / m_rows;
    const double ratio = mu_aff / n_mu;
    // centring: Mehrotra's cube, not below what keeps mu from undershooting the target in one step
    const double sg = fmin(1.0, fmax(ratio * ratio * ratio, 0.1 * opt.tol * (1.0 + fabs(n_obj)) / n_mu));
    const double smu = sg * n_mu;
    // ---- corrector:  s dz + z ds = sigma mu - s z - ds dz  ->  zh = (sigma mu - ds dz) / s ----
    auto zh = [&](int r) { return (smu - ln.cross(r)) * ln.is(r); };
    newton(zh);
    ms = 0.0;
    mz = 0.0;
    int rb = -1;
    st.rows(C, ln, [&](int r, double gv) {
      const double ds = -gv, dz = fma(ln.w(r), gv, zh(r) - ln.z[r]);
      const double q = -ds * ln.is(r);
      rb = q > ms ? r : rb;
      ms = fmax(ms, q);
      mz = fmax(mz, -dz * recip(ln.z[r]));
    });
    const double my_ms = on ? ms : 0.0;
    ms = x.max(my_ms);
    mz = x.max(on ? mz : 0.0);
    ap = ms > 0.995 ? 0.995 / ms : 1.0;   // min(1, 0.995 x the step to the nearest row)
    ad = mz > 0.995 ? 0.995 / mz : 1.0;
    if (tr) { tr[4] = ap; tr[5] = ad; tr[6] = sg; tr[7] = mu_aff; }
    if (trg && my_ms == ms && rb >= 0) { trg[8] = double(i); trg[9] = double(rb); trg[10] = 0.0; trg[11] = 0.0; }
    st.forget_all();
    double smu2 = smu;
    forget(smu2);
    auto zh2 = [&](int r) { return (smu2 - ln.cross(r)) * ln.is(r); };
    if (!done) {
      ANM_UFOR (int g = 0; g < NG; ++g) ln.xi[g] = fma(ap, st.dxi[g], ln.xi[g]);
      ANM_UFOR (int j = 0; j < NS; ++j) { ln.pc[j] = fma(ap, st.dpc[j], ln.pc[j]); ln.d[j] = fma(ap, st.dd[j], ln.d[j]); }
      ANM_UFOR (int e = 0; e < NBR; ++e) ln.t[e] = fma(ap, st.dt[e], ln.t[e]);
      st.rows(C, ln, [&](int r, double gv) {
        const double dz = fma(ln.w(r), gv, zh2(r) - ln.z[r]);
        ln.s[r] = fma(-ap, gv, ln.s[r]);
        ln.z[r] = fma(ad, dz, ln.z[r]);
      });
    }
  }
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------------------
// gfx950: the lanes of a group exchange through wavefront shuffles.  SINGLE: one stage, one thread per
// environment, no exchange at all.
// ------------------------------------------------------------------------------------------------------------
// one DPP move of a double inside its row of 16 lanes; lanes a shift brings in from outside the row read 0
template <int CTRL>
__device__ __forceinline__ double row_move(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// (masked form: rows outside `ROWS` read 0)
template <int CTRL, int ROWS>
__device__ __forceinline__ double row_move_masked(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWS, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWS, 0xF, false);
  return __hiloint2double(hi, lo);
}
// the value of the lane 16 / 32 lanes away (lane ^ 16, lane ^ 32): gfx950's v_permlane16_swap / v_permlane32_swap of a
// register with a copy of itself leave the odd rows' (upper half's) values in one result and the even rows' (lower
// half's) in the other
__device__ __forceinline__ double lane_xor16(double x, bool odd_row) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
  return __hiloint2double(odd_row ? hi[0] : hi[1], odd_row ? lo[0] : lo[1]);
}
__device__ __forceinline__ double lane_xor32(double x, bool upper) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
  return __hiloint2double(upper ? hi[0] : hi[1], upper ? lo[0] : lo[1]);
}

// MODE 0: one stage, one thread per environment, no exchange at all.
// MODE 1: groups of 2, 4, 8 or 16 lanes (N <= 16): a group lies inside a DPP row.
// MODE 2: groups of 32 or 64 lanes.
// Every exchange is a handful of v_mov_b32_dpp / v_permlane*_swap (no LDS crossbar, no latency to wait for):
// neighbours by row_shr:1 / row_shl:1 (across rows: wave_shr:1 / wave_shl:1); all-reduces by the quad permutations,
// row_half_mirror and row_mirror, then lane ^ 16 and lane ^ 32 (each step pairs a lane with one of the other half
// of its 2-, 4-, ... 64-lane block: both add the same two numbers, so every lane of the group ends with the same
// bits); prefix sums by row_shr:1, 2, 4, 8, then row_bcast15 / row_bcast31.  A step beyond the group is carried
// out and dropped: no loop whose trip count depends on G -- with such loops between them the row arrays of the lane
// no longer fit its registers.
template <int MODE>
struct WaveGroup {
  static constexpr bool SINGLE = MODE == 0, ROW = MODE == 1;
  int G, st, lane;
  __device__ int stage() const { return st; }
  __device__ double up(double v) const {
    if (SINGLE) return 0.0;
    const double r = ROW ? row_move<0x111>(v) : row_move<0x138>(v);
    return st == 0 ? 0.0 : r;
  }
  __device__ double down(double v) const {
    if (SINGLE) return 0.0;
    const double r = ROW ? row_move<0x101>(v) : row_move<0x130>(v);
    return st == G - 1 ? 0.0 : r;
  }
  template <class Op>
  __device__ double all(double v, Op op) const {
    if constexpr (!SINGLE) {
      v = op(v, row_move<0xB1>(v));                                      // quad_perm [1, 0, 3, 2]
      { const double t = op(v, row_move<0x4E>(v)); v = G > 2 ? t : v; }   // quad_perm [2, 3, 0, 1]
      { const double t = op(v, row_move<0x141>(v)); v = G > 4 ? t : v; }  // row_half_mirror
      { const double t = op(v, row_move<0x140>(v)); v = G > 8 ? t : v; }  // row_mirror
      if constexpr (!ROW) {
        v = op(v, lane_xor16(v, (lane & 16) != 0));
        { const double t = op(v, lane_xor32(v, (lane & 32) != 0)); v = G > 32 ? t : v; }
      }
    }
    return v;
  }
  __device__ double sum(double v) const { return all(v, [](double a, double b) { return a + b; }); }
  __device__ double max(double v) const { return all(v, [](double a, double b) { return fmax(a, b); }); }
  __device__ double min(double v) const { return all(v, [](double a, double b) { return fmin(a, b); }); }
  __device__ double scan(double v) const {
    if constexpr (!SINGLE) {
      const int r = st & 15;   // place in the row
      { const double t = row_move<0x111>(v); v += r >= 1 ? t : 0.0; }
      { const double t = row_move<0x112>(v); v += r >= 2 ? t : 0.0; }
      { const double t = row_move<0x114>(v); v += r >= 4 ? t : 0.0; }
      { const double t = row_move<0x118>(v); v += r >= 8 ? t : 0.0; }
      if constexpr (!ROW) {
        v += row_move_masked<0x142, 0xA>(v);                                        // rows 1, 3 += the total of rows 0, 2
        { const double t = row_move_masked<0x143, 0xC>(v); v += G > 32 ? t : 0.0; }  // rows 2, 3 += the total of rows 0 + 1
      }
    }
    return v;
  }
  __device__ bool all_done(bool d) const { return __all(d); }
};

template <class T, int MODE>
__global__ __launch_bounds__(64) void k_mpc(cptr_t C, IO io, Opts opt, int64_t n_envs, int N, int G) {
  if constexpr (Sz<T>::FITS) {
    const int lane = threadIdx.x;
    const int64_t env = int64_t(blockIdx.x) * (64 / G) + lane / G;
    WaveGroup<MODE> x{G, lane % G, lane};
    extern __shared__ double mpc_lds[];   // [2 NR][64]: 1/s and the predictor's ds*dz of every row of every lane
    solve<T>(C, io, opt, env, env < n_envs, N, x, mpc_lds + lane);
  }
}
#endif

}  // namespace mpc
}  // namespace anm
