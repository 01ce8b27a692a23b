// anm_mesh.hpp -- lane-group kernel for ANY network topology (loops, parallel feeders, transformers) of up
// to 65 buses, 128 branches, 64 devices: the general-sparsity sibling of anm_radial.hpp.
//
// The thread-per-environment kernels keep a whole environment in one thread's registers; that stops
// working around a dozen buses (a meshed 30-bus network: 512 registers + scratch per lane, 2.8 ms per
// 16 384 transitions).  The radial kernel spreads an environment over a lane group but relies on the
// network being a tree.  Here the same lane group works on a general sparse Jacobian:
//     lane l  <->  bus l + 1,  device l,  branches l and G + l      (the group size follows the buses)
// Per Newton iteration
//   bus roles publish V; BRANCH roles form the products W_ft = V_f conj(Y_ft V_t), W_tf of their branch and
//   with them the two off-diagonal 2x2 Jacobian blocks of the branch; bus roles sum their row of W (mismatch,
//   diagonal block); the group-wide stop test is two lane masks;
//   the block matrix lives in LDS ([blocks][4] doubles per environment, the right-hand side as one more block
//   column) and is factorised by a PROGRAM OF STEPS that build_plan schedules once per network on the host
//   (order: rounds of pairwise non-adjacent pivots of near-minimal degree; see StepType / OpKind below): in a
//   step every lane carries out at most one small operation named by its descriptor, operations go to
//   whatever lanes are free, a fence separates the steps; per level one step of products
//   M = A_ik D_k^-1 A_kj (which subtract the first contribution to a destination themselves) and, where a
//   destination has several, one of sums; then the back substitution by columns, one step per link of its
//   longest dependency chain (the last two buses as one dense step).  tests/hostsim/mesh_program_check.cpp
//   executes the schedule on the host.
// A workgroup is 1, 2 or 4 wavefronts that share one LDS copy of the tables and otherwise never meet; a lane
// group lies within one wavefront, whose LDS operations complete in program order, so the hand-overs between
// the steps need compiler fences only.  Nothing is compiled per topology: integer tables drive one generic
// kernel (the reference solves whatever network it is given: solve_load_flow.py:123-164, 220 -- SciPy's
// sparse LU on the same matrix).  Same reference semantics and the same formulation (W products, magnitude
// columns scaled by |V|, (cos, sin) state, rotated stop test) as anm_device.hpp; see the citations there.
#pragma once

#include <algorithm>
#include <array>
#include <map>
#include <cstdlib>
#include <set>
#include <string>
#include <vector>

#include "anm_radial.hpp"

namespace anm {
namespace mesh {

using radial::DEV_NONE;
using radial::KMAX;
using radial::SF_BASE; using radial::SF_DT; using radial::SF_LAMB; using radial::SF_C1; using radial::SF_C2;
using radial::SF_RTERM; using radial::SF_PERIOD; using radial::SF_Y00_RE; using radial::SF_Y00_IM;
using radial::SF_SLACK_VMIN; using radial::SF_SLACK_VMAX; using radial::SF_COUNT;

enum IField : int { IF_LEVEL = 0, IF_DIAG, IF_INC_BEG, IF_INC_END, IF_BD_BEG, IF_BD_END, IF_INC0, IF_DEG,
                    IF_DEV_TYPE, IF_DEV_SLOT, IF_DEV_SET, IF_XINV,   // IF_XINV: the bus's pivot block was never inverted (see ST_COL)
                    IF_BR_F, IF_BR_T, IF_BR_BLK_FT, IF_BR_BLK_TF, IF_BR_POS_F, IF_BR_POS_T,   // branch l ...
                    IF_BR_FIELDS = IF_BR_POS_T - IF_BR_F + 1,
                    IF_COUNT = IF_BR_F + 2 * IF_BR_FIELDS };                                  // ... and branch G + l
// (a lane plays up to BR_SLOTS branches: meshed networks have more branches than buses, and the group size
// follows the buses -- two environments of a 30-bus, 41-branch network share a wavefront)
constexpr int BR_SLOTS = 2;
// The elimination as a PROGRAM of steps (build_plan schedules it once per network; k_mesh interprets it): in every
// step each lane carries out at most one small operation named by its descriptor -- 8 offsets (in doubles) into
// the environment's LDS -- and a wavefront-wide fence separates the steps.  Operations are handed to lanes
// wherever one is free (no lane "owns" a row), so a level of the elimination costs a fixed, short sequence of
// instructions whatever the sparsity pattern, instead of a per-lane loop over task lists.
// The right-hand side rides along as one more block column: r_i is kept as the block (r0, 0; r1, 0).
//   step ST_PROD  OP_PROD (i, k, j)    s1 = D_k, s2 = A_ik, s3 = A_kj (or r_k), s6 = Z, s4 = where Z - A_ik D_k^-1 A_kj
//                                      goes; s5 != 0: also leave D_k^-1 at s5.  Z = s4 = A_ij (r_i) for the FIRST
//                                      contribution to a destination: the product operation subtracts it itself (a
//                                      destination is no operand of its own level).  Z = the zero block for the
//                                      others: the NEGATED product is parked for the sums
//   step ST_SUM   OP_SUM  (i, j)       s1 = A_ij (or r_i) += the parked products at s2, s3, s4, s5 in this order (the step
//                                      says how many of them any of its lanes needs; an unused one is the zero block).
//                                      A level whose destinations all have one contribution has no such step
//   step ST_TAIL  OP_TAIL2 (a, b)      the last two buses of the order when each is alone in its level: ONE lane
//                                      eliminates a into b and prepares both -- one step instead of the three
//                                      (products, sums, the column of b) the last two levels would take.
//                                      s1 = D_a, s2 = A_ab, s3 = A_ba, s4 = D_b, s5 = r_a, s6 = r_b; it leaves
//                                      r_b, D_b^-1, r_a - A_ab x_b, D_a^-1 (the inverses where the r's index says)
//   step ST_COL   OP_COL  (i; k, k')   back substitution by columns: s1 = r_i -= A_ik x_k + A_ik' x_k' with
//                                      (s2, s3, s4) = (A_ik, r_k, D_k^-1) and (s5, s6, s7) the same for k' (or three
//                                      zero blocks), x_k = D_k^-1 r_k formed by the operation itself: r_k is final
//                                      once every later bus has left it, nobody stores x.  | OP_COL_INV1 / _INV2: the
//                                      bus k / k' had nothing eliminated into it, s4 / s7 is D_k itself.  When the
//                                      program ends every bus lane forms its own x_i = D_i^-1 r_i (IF_XINV: from D_i)
// s0 = the operation (0: the lane idles in this step).  Nothing a step reads is written in the same step by ANOTHER
// lane, and D_k, A_ik, A_kj stay as they are (the factor L_ik = A_ik D_k^-1 is never stored: every product recomputes
// it, lanes are plentiful), so the operations of a level may be dealt to the lanes in any number of rounds.
// A ZONE of trailing levels is eliminated Gauss-Jordan style (build_plan_zone): there the product step of a pivot also
// takes it out of the earlier rows of the zone, so those rows owe the column sweep the terms of the tail only.
// (Rounds 2-4: the back substitution went row-wise -- x_k stored, up to four steps per level; products and sums were
// two steps on every level; "fused levels", every contribution subtracted by its product operation two at a time, were
// tried in round 3 and lost what they won to the longer steps: profiles/r03_n_mesh_fused_levels.txt.)
enum OpKind : int { OP_NONE = 0, OP_PROD, OP_SUM, OP_TAIL2, OP_COL = 8, OP_COL_INV1 = 1, OP_COL_INV2 = 2 };   // OP_COL | INV bits: 8 ... 11
enum StepType : int { ST_PROD = 0, ST_SUM = 1, ST_TAIL = 2, ST_COL = 3 };
enum DField : int { DF_YII_RE = 0, DF_YII_IM, DF_VMIN, DF_VMAX, DF_YFT_RE, DF_YFT_IM, DF_YTF_RE, DF_YTF_IM, DF_BRC,
                    DF_BR_FIELDS = DF_BRC + 9 - DF_YFT_RE, DF_COUNT = DF_YFT_RE + 2 * DF_BR_FIELDS };

struct Dims {
  int G, NB, ND, NBR, NLOAD, NGEN, NDES, NSET, SDIM, FS, n_levels, zone_level, slack_dev, NBLK, n_fill, br_slots;
  int off_lists, n_lists, off_fill;                        // ints: [IF_COUNT][G], lists, fill ids (staged in LDS) ...
  int n_steps, off_stype, off_desc, n_stage, max_deg;      // ... step types (| run length << 8) [n_steps], descriptors [n_steps][G][4]
  int off_lane, off_dev, off_obs_lo, off_obs_hi, n_double;  // doubles
  int l_v, l_x, l_blk, l_r, l_dinv, l_zero, l_bw, l_m, n_m, l_dev, l_red, lds_per_env;   // LDS layout of one environment (doubles)
  int f_bus_p, f_bus_q, f_bus_vm, f_bus_va, f_bus_im, f_bus_ia, f_dev_p, f_dev_q, f_des_soc, f_gen_pmax, f_br_p,
      f_br_q, f_br_s, f_br_im, f_br_ia;
};

struct Plan {
  Dims d;
  std::vector<int> hi;
  std::vector<double> hd;
};

// Groups above a wavefront (networks of more than 65 buses): the environment IS the workgroup (2 ... 8 wavefronts,
// k_mesh<.., WG = true>): workgroup barriers instead of wavefront fences, the tables are read from global memory
// (the program of a 300-bus network is larger than the LDS), reductions and the stop test go through RED_DOUBLES
// doubles of LDS
constexpr int MAX_GROUP = 512;   // (a 1024-lane workgroup would leave 128 registers per lane: the kernel needs 186)
constexpr int RED_DOUBLES = 64;
inline bool is_workgroup(const Dims& d) { return d.G > 64; }

// LDS of a workgroup of `waves` wavefronts: the environments' own areas + (tables: the staged tables, one copy)
inline size_t lds_bytes(const Dims& d, int waves, bool tables) {
  if (is_workgroup(d)) return size_t(d.lds_per_env) * sizeof(double);
  return size_t(waves) * (64 / d.G) * d.lds_per_env * sizeof(double) + (tables ? size_t(d.n_stage) * sizeof(int) : 0);
}
inline size_t waves_on_cu(const Dims& d, int w, int cap, bool tables) {
  const size_t b = lds_bytes(d, w, tables);
  return b <= 160 * 1024 ? std::min<size_t>(size_t(cap), (160 * 1024 / b) * w) : 0;
}
inline size_t most_waves_on_cu(const Dims& d, int cap, bool tables) {
  size_t best = 0;
  for (int w = 1; w <= 4; w *= 2) best = std::max(best, waves_on_cu(d, w, cap, tables));
  return best;
}

// How a plan is launched: which variant of k_mesh, in workgroups of how many wavefronts.
//
// simd_waves -- wavefronts per SIMD the kernel is compiled for.  k_mesh holds ~190 registers per lane (two wavefronts per
//   SIMD); budgeted for three (168 registers, ~110 bytes per lane more in scratch) it is the faster kernel where the LDS of
//   the environments lets a compute unit hold twelve wavefronts AND the batch is throughput-bound -- a meshed 20-bus
//   network, 16 384 transitions: 166 -> 148 us.  Where LDS stops at eight to ten wavefronts the spills cost 1-2 % and buy
//   nothing (meshed 30 buses: 549 -> 558 us at eight, 544 at nine, 613 at ten in two workgroups of five); and a batch that
//   waits for one diverging solve's hundred trips pays for them on every trip (ANM6 forced through this family, 8
//   environments per wavefront, 65 536 transitions: 361 -> 373 us) -- so: groups of 16 lanes and more only (what reaches
//   this family by default).  profiles/r05_i_mesh_occupancy.txt
// tables_in_lds -- the lists and the step program staged in LDS, one copy per workgroup (a 30-bus network: 5.5 KB), or read
//   from global memory where they are.  Staged they are the faster tables (meshed 30 buses 547 us against 566, 20 buses 150
//   against 153) -- unless the copy costs the compute unit wavefronts: a meshed 64-bus network has 29 KB of tables next to
//   20 KB per environment, four wavefronts per compute unit with the copy and eight without: 8 192 transitions
//   8.04 -> 5.45 ms.  profiles/r05_j_mesh_tables.txt
// waves -- wavefronts per workgroup (they share the staged tables and otherwise never meet): what puts most wavefronts on
//   a compute unit.  On a tie the larger workgroup for groups of 32 lanes and more (fewer copies of the tables to stage:
//   the batch of the meshed 30-bus network -4 %), the smaller one below: a workgroup gives its wavefront slots back when
//   its LAST wavefront ends, and with 8 environments per wavefront a fair share of the wavefronts carry a diverging solve
//   to the iteration cap (ANM6 through this family, 65 536 transitions: 358 us with one wavefront per workgroup, 445 us
//   with four; profiles/r04_m_mesh_step_program.txt)
// Deterministic in the plan and the tuning switches ANM_MESH_SIMD_WAVES=2|3, ANM_MESH_TABLES=lds|global, ANM_MESH_WAVES=1|2|4.
struct Launch {
  int simd_waves, waves;
  bool tables_in_lds;
  size_t lds, waves_per_cu;
};
inline Launch launch_of(const Dims& d) {
  Launch L{2, 1, false, 0, 0};
  if (is_workgroup(d)) {
    L.waves = d.G / 64;
    L.lds = lds_bytes(d, L.waves, false);
    L.waves_per_cu = std::min<size_t>(8, (160 * 1024 / L.lds) * L.waves);
    return L;
  }
  L.tables_in_lds = true;
  if (d.G >= 16 && most_waves_on_cu(d, 12, true) >= 12) L.simd_waves = 3;
  if (const char* ev = getenv("ANM_MESH_SIMD_WAVES")) {
    const int v = atoi(ev);
    if (v == 2 || v == 3) L.simd_waves = v;
  }
  // (the variant that leaves the tables where they are exists for two wavefronts per SIMD: where three fit, LDS is no issue)
  if (L.simd_waves == 2 && most_waves_on_cu(d, 8, false) > most_waves_on_cu(d, 8, true)) L.tables_in_lds = false;
  if (const char* ev = getenv("ANM_MESH_TABLES")) {
    if (std::string(ev) == "lds" && lds_bytes(d, 1, true) <= 160 * 1024) L.tables_in_lds = true;
    if (std::string(ev) == "global") { L.tables_in_lds = false; L.simd_waves = 2; }
  }
  const int cap = 4 * L.simd_waves;
  size_t best_waves = 0;
  for (int w = 1; w <= 4; w *= 2) {
    const size_t per_cu = waves_on_cu(d, w, cap, L.tables_in_lds);
    if (per_cu > best_waves || (per_cu == best_waves && d.G >= 32)) { best_waves = per_cu; L.waves = w; }
  }
  if (const char* ev = getenv("ANM_MESH_WAVES")) {
    const int w = atoi(ev);
    if ((w == 1 || w == 2 || w == 4) && lds_bytes(d, w, L.tables_in_lds) <= 160 * 1024) L.waves = w;
  }
  L.lds = lds_bytes(d, L.waves, L.tables_in_lds);
  L.waves_per_cu = waves_on_cu(d, L.waves, cap, L.tables_in_lds);
  return L;
}
inline int envs_per_block(const Dims& d, const Launch& L) { return is_workgroup(d) ? 1 : L.waves * (64 / d.G); }

inline bool fits(const anm_network_desc& n) {
  return n.n_bus >= 2 && n.n_bus - 1 <= MAX_GROUP && n.n_dev <= MAX_GROUP && n.n_branch <= MAX_GROUP * BR_SLOTS;
}

// Symbolic analysis + per-lane tables for one network; the levels from `zone_level` on are eliminated Gauss-Jordan
// style (see "the ZONE" below; build_plan picks the level).
inline bool build_plan_zone(const anm_network_desc& n, Plan& P, std::string& err, int zone_level, int c_max = 1 << 30) {
  if (!fits(n)) { err = "the general lane-group kernel takes networks of at most 513 buses, 1024 branches, 512 devices"; return false; }
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
  const int need = std::max(std::max(d.NB - 1, d.ND), (d.NBR + BR_SLOTS - 1) / BR_SLOTS);
  d.G = 8;
  while (d.G < need) d.G *= 2;
  d.br_slots = (d.NBR + d.G - 1) / d.G;
  const int G = d.G, NB = d.NB;
  d.f_bus_p = 0; d.f_bus_q = d.f_bus_p + NB; d.f_bus_vm = d.f_bus_q + NB; d.f_bus_va = d.f_bus_vm + NB;
  d.f_bus_im = d.f_bus_va + NB; d.f_bus_ia = d.f_bus_im + NB; d.f_dev_p = d.f_bus_ia + NB;
  d.f_dev_q = d.f_dev_p + d.ND; d.f_des_soc = d.f_dev_q + d.ND; d.f_gen_pmax = d.f_des_soc + d.NDES;
  d.f_br_p = d.f_gen_pmax + d.NGEN; d.f_br_q = d.f_br_p + d.NBR; d.f_br_s = d.f_br_q + d.NBR;
  d.f_br_im = d.f_br_s + d.NBR; d.f_br_ia = d.f_br_im + d.NBR; d.FS = d.f_br_ia + d.NBR;

  // ---- Y (dense on the host; parallel branches between the same pair of buses are not supported, as in
  // the reference whose Y[f, t] assignment keeps only the last one: refuse instead of silently differing)
  std::vector<cplx> Y(size_t(NB) * NB, cplx(0, 0));
  std::set<std::pair<int, int>> pairs;
  for (int b = 0; b < d.NBR; ++b) {
    const int f = n.br_from[b], t = n.br_to[b];
    if (f == t || !pairs.insert({std::min(f, t), std::max(f, t)}).second) {
      err = "parallel branches / self loops are not supported";
      return false;
    }
    const cplx ys(n.br_series[2 * b], n.br_series[2 * b + 1]), sh(n.br_shunt[2 * b], n.br_shunt[2 * b + 1]);
    const cplx tap(n.br_tap[2 * b], n.br_tap[2 * b + 1]);
    Y[f * NB + t] = -ys / std::conj(tap);
    Y[t * NB + f] = -ys / tap;
    Y[f * NB + f] += (ys + sh) / (std::abs(tap) * std::abs(tap));
    Y[t * NB + t] += ys + sh;
  }

  // ---- symbolic block LU over the non-slack buses: minimum degree (ties: larger bus first, leaves of a
  // feeder before its trunk), fill, elimination levels
  std::vector<std::set<int>> adj(NB);
  for (int b = 0; b < d.NBR; ++b) {
    const int f = n.br_from[b], t = n.br_to[b];
    if (f != 0 && t != 0) { adj[f].insert(t); adj[t].insert(f); }
  }
  std::vector<std::vector<int>> blk(NB, std::vector<int>(NB, -1));
  int nblk = 0;
  for (int i = 1; i < NB; ++i) {
    blk[i][i] = nblk++;
    for (int j : adj[i]) blk[i][j] = nblk++;
  }
  std::vector<int> fill_ids;
  std::vector<std::set<int>> work = adj;
  std::vector<char> done(NB, 0);
  std::vector<int> order, level(NB, 0);
  std::vector<std::vector<int>> upper(NB);
  // Order: rounds of pairwise non-adjacent pivots of (near-)minimal degree -- the parallel tree contraction
  // ("rake" the leaves, "compress" every other bus of a chain) generalised by the degree rule.  Plain minimum
  // degree peels a feeder from its ends, one level per bus of its longest chain; with independent sets the
  // number of LEVELS, which is what the level-synchronous kernel pays for, grows like log(n) for feeders.
  for (int n_done = 0; n_done < NB - 1;) {
    int dmin = 1 << 30;
    std::vector<int> deg(NB, 0);
    for (int u = 1; u < NB; ++u) {
      if (done[u]) continue;
      for (int v : work[u]) deg[u] += !done[v];
      dmin = std::min(dmin, deg[u]);
    }
    const int dcut = std::max(dmin, 2);   // chains (degree 2) are compressed together with the leaves
    std::vector<int> cand;
    for (int u = NB - 1; u >= 1; --u)
      if (!done[u] && deg[u] <= dcut) cand.push_back(u);
    std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return deg[a] < deg[b]; });
    std::vector<char> blocked(NB, 0);
    std::vector<int> round;
    for (int u : cand) {
      if (blocked[u]) continue;
      round.push_back(u);
      blocked[u] = 1;
      for (int v : work[u]) blocked[v] = 1;
    }
    for (int k : round) {
      done[k] = 1;
      ++n_done;
      order.push_back(k);
      std::vector<int> nb;
      for (int v : work[k]) if (!done[v]) nb.push_back(v);
      upper[k] = nb;
      for (int i : nb)
        for (int j : nb) {
          if (blk[i][j] < 0) { blk[i][j] = nblk++; fill_ids.push_back(blk[i][j]); }
          if (i != j) work[i].insert(j);
        }
      for (int i : nb) level[i] = std::max(level[i], level[k] + 1);
    }
  }
  if (c_max < (1 << 30)) {
    // at most c_max contributions per destination and level (1: no sums steps at all): a pivot moves to the first level
    // at or after the one its dependencies give it where its contributions still fit (later is always allowed: what it
    // reads is final, and the buses that wait for it are placed after it)
    std::vector<int> lv2(NB, 0);
    std::vector<std::map<std::pair<int, int>, int>> cnt;
    for (int k : order) {
      int lv = 0;
      for (int q = 1; q < NB; ++q) {   // eliminated neighbours of k: those with k in their upper
        if (q == k) continue;
        for (int j : upper[q]) if (j == k) lv = std::max(lv, lv2[q] + 1);
      }
      for (;; ++lv) {
        if (int(cnt.size()) <= lv) cnt.resize(lv + 1);
        bool ok = true;
        for (int i : upper[k]) {
          for (int j : upper[k]) if (cnt[lv].count({i, j}) && cnt[lv][{i, j}] >= c_max) ok = false;
          if (cnt[lv].count({i, NB}) && cnt[lv][{i, NB}] >= c_max) ok = false;
        }
        if (ok) break;
      }
      lv2[k] = lv;
      for (int i : upper[k]) { for (int j : upper[k]) cnt[lv][{i, j}]++; cnt[lv][{i, NB}]++; }
    }
    level = lv2;
  }
  d.n_levels = 0;
  for (int i = 1; i < NB; ++i) d.n_levels = std::max(d.n_levels, level[i] + 1);
  std::vector<std::vector<int>> piv(d.n_levels);
  for (int k : order) piv[level[k]].push_back(k);
  // the last two buses of the order, each alone in its level (the second-to-last is then coupled to the last only)
  int tail_a = -1, tail_b = -1;
  if (!getenv("ANM_MESH_NO_TAIL") && d.n_levels >= 2 && piv[d.n_levels - 1].size() == 1 && piv[d.n_levels - 2].size() == 1 &&
      upper[piv[d.n_levels - 2][0]].size() == 1 && upper[piv[d.n_levels - 2][0]][0] == piv[d.n_levels - 1][0]) {
    tail_a = piv[d.n_levels - 2][0];
    tail_b = piv[d.n_levels - 1][0];
  }
  // The ZONE: the levels from zone_level on (the tail apart) are eliminated Gauss-Jordan style -- the product step of a
  // pivot k of the zone also takes k out of the rows of the EARLIER buses i of the zone that are coupled to it
  // (A_ij -= A_ik D_k^-1 A_kj for the later buses j of k, r_i -= A_ik D_k^-1 r_k; a block (i, j) that does not exist yet
  // is more fill).  Lanes are plentiful, steps are not: the buses of the zone then owe the back substitution only the
  // terms of the tail and become ready TOGETHER, where the column sweep walks one step per link of their dependency
  // chain (the dense end of the order -- one pivot per level, each coupled to all later buses: what the fill leaves of a
  // meshed core -- cost one step per bus).  rows_up[k]: those earlier rows; zstruct[i]: the later buses row i of the
  // zone is still coupled to.
  const int last_lv = d.n_levels - 1 - (tail_a >= 0 ? 2 : 0);    // last level with a product step
  d.zone_level = std::min(std::max(zone_level, 0), last_lv + 1);
  std::vector<std::vector<int>> rows_up(NB);
  std::vector<std::set<int>> zstruct(NB);
  std::vector<char> in_zone(NB, 0);
  {
    std::vector<int> zone_done;
    for (int lv = d.zone_level; lv <= last_lv; ++lv)
      for (int k : piv[lv]) { in_zone[k] = 1; zstruct[k] = std::set<int>(upper[k].begin(), upper[k].end()); }
    for (int lv = d.zone_level; lv <= last_lv; ++lv) {
      for (int k : piv[lv])
        for (int i : zone_done)
          if (zstruct[i].count(k)) rows_up[k].push_back(i);
      for (int k : piv[lv])            // (pivots of a level are not coupled to each other: the order does not matter)
        for (int i : rows_up[k]) {
          for (int j : upper[k]) {
            if (blk[i][j] < 0) { blk[i][j] = nblk++; fill_ids.push_back(blk[i][j]); }
            zstruct[i].insert(j);
          }
          zstruct[i].erase(k);
        }
      for (int k : piv[lv]) zone_done.push_back(k);
    }
  }
  d.NBLK = nblk;
  d.n_fill = int(fill_ids.size());

  // ---- LDS layout of one environment (doubles).  The branch products W (written and summed before the step
  // program starts) and the parked products of the elimination are never alive together and share their place
  d.l_v = 0;
  d.l_x = d.l_v;                         // vr[NB], vi[NB]
  d.l_blk = d.l_v + 2 * NB;
  d.l_r = d.l_blk + 4 * d.NBLK;          // r[NB] as blocks (r0, 0; r1, 0)
  d.l_dinv = d.l_r + 4 * NB;             // inverted pivots [NB][4]
  d.l_zero = d.l_dinv + 4 * NB;          // 6 zeros nobody writes: what an unused operand slot of a descriptor reads
  d.l_bw = d.l_zero + 6;                 // W entries [2 NBR][2] / I = Y V terms [4][NBR] ...
  d.l_m = d.l_bw;                        // ... / products [n_m][4]

  // ---- int tables
  P.hi.assign(size_t(IF_COUNT) * G, 0);
  auto I = [&](int f, int l) -> int& { return P.hi[size_t(f) * G + l]; };
  for (int l = 0; l < G; ++l) {
    I(IF_LEVEL, l) = -1; I(IF_DIAG, l) = -1; I(IF_DEV_TYPE, l) = DEV_NONE; I(IF_DEV_SLOT, l) = -1; I(IF_DEV_SET, l) = -1;
    for (int sl = 0; sl < BR_SLOTS; ++sl) {
      I(IF_BR_F + sl * IF_BR_FIELDS, l) = -1; I(IF_BR_T + sl * IF_BR_FIELDS, l) = -1;
      I(IF_BR_BLK_FT + sl * IF_BR_FIELDS, l) = -1; I(IF_BR_BLK_TF + sl * IF_BR_FIELDS, l) = -1;
    }
  }
  auto IB = [&](int f, int br) -> int& { return P.hi[size_t(f + (br / G) * IF_BR_FIELDS) * G + br % G]; };   // field of a branch
  std::vector<int> lists;
  // W entries: entry e holds (re, im) of one end of one branch at LDS doubles l_bw + 2e; the entries a bus sums
  // are consecutive, in branch order (the order the reference sums a row of Y V in); the slack ends come last
  int n_ent = 0;
  d.max_deg = 0;
  for (int l = 0; l + 1 < NB; ++l) {
    const int b = l + 1;
    I(IF_LEVEL, l) = level[b];
    I(IF_DIAG, l) = blk[b][b];
    I(IF_INC_BEG, l) = int(lists.size());
    I(IF_INC0, l) = n_ent;
    for (int br = 0; br < d.NBR; ++br) {
      if (n.br_from[br] == b) { lists.push_back(br << 1); IB(IF_BR_POS_F, br) = n_ent++; }
      else if (n.br_to[br] == b) { lists.push_back((br << 1) | 1); IB(IF_BR_POS_T, br) = n_ent++; }
    }
    I(IF_INC_END, l) = int(lists.size());
    I(IF_DEG, l) = n_ent - I(IF_INC0, l);
    d.max_deg = std::max(d.max_deg, I(IF_DEG, l));
    I(IF_BD_BEG, l) = int(lists.size());
    for (int k = 0; k < d.ND; ++k)
      if (n.dev_bus[k] == b) lists.push_back(k);
    I(IF_BD_END, l) = int(lists.size());
  }
  for (int br = 0; br < d.NBR; ++br) {
    const int f = n.br_from[br], t = n.br_to[br];
    if (f == 0) IB(IF_BR_POS_F, br) = n_ent++;
    if (t == 0) IB(IF_BR_POS_T, br) = n_ent++;
    IB(IF_BR_F, br) = f; IB(IF_BR_T, br) = t;
    IB(IF_BR_BLK_FT, br) = (f != 0 && t != 0) ? blk[f][t] : -1;
    IB(IF_BR_BLK_TF, br) = (f != 0 && t != 0) ? blk[t][f] : -1;
  }
  for (int k = 0; k < d.ND; ++k) {
    I(IF_DEV_TYPE, k) = n.dev_type[k];
    I(IF_DEV_SLOT, k) = slot[k];
    I(IF_DEV_SET, k) = sset[k];
  }
  d.off_lists = int(P.hi.size());
  d.n_lists = int(lists.size());
  P.hi.insert(P.hi.end(), lists.begin(), lists.end());
  d.off_fill = int(P.hi.size());
  P.hi.insert(P.hi.end(), fill_ids.begin(), fill_ids.end());
  P.hi.push_back(0);

  // ---- the step program
  typedef std::array<int, 8> Desc;
  std::vector<int> stype;
  std::vector<std::vector<Desc>> steps;          // [step][lane]; a product slot m is written -(m + 1) until the layout is known
  const Desc none = {0, 0, 0, 0, 0, 0, 0, 0};
  auto new_step = [&](int ty) { stype.push_back(ty); steps.push_back(std::vector<Desc>()); return int(steps.size()) - 1; };
  const int oB = d.l_blk, oR = d.l_r, oDI = d.l_dinv, oZ = d.l_zero;
  int n_m = 0;
  std::vector<char> dinv_stored(NB, 0);
  for (int lv = 0; lv < d.n_levels; ++lv) {
    if (tail_a >= 0 && lv >= d.n_levels - 2) continue;   // folded into the ST_TAIL step
    // destinations (block (i, j); j == NB: r_i) and, in pivot order, who contributes to them
    struct Contribution { int i, k, j; };
    std::vector<std::pair<int, int>> dsts;
    std::vector<std::vector<Contribution>> contrib;
    std::vector<std::vector<int>> dst_of(NB, std::vector<int>(NB + 1, -1));
    for (int k : piv[lv]) {
      std::vector<int> rows = upper[k];
      rows.insert(rows.end(), rows_up[k].begin(), rows_up[k].end());
      for (int i : rows)
        for (size_t q = 0; q <= upper[k].size(); ++q) {
          const int j = q < upper[k].size() ? upper[k][q] : NB;
          if (blk[i][k] < 0 || (j < NB && blk[i][j] < 0)) { err = "internal: a block of the zone is missing"; return false; }
          if (dst_of[i][j] < 0) { dst_of[i][j] = int(dsts.size()); dsts.push_back({i, j}); contrib.push_back({}); }
          contrib[dst_of[i][j]].push_back(Contribution{i, k, j});
        }
    }
    if (dsts.empty()) continue;
    // The FIRST contribution of a destination is subtracted by the product operation itself (the destination is no
    // operand of its level: neither of its indices is a pivot of the level); the others are parked, NEGATED, and added
    // by summation rounds of up to four per destination (an unused operand is the zero block), in contribution order:
    // ((Z - M0) + (-M1)) + (-M2) ... -- a level whose destinations all have one contribution (one pivot: the dense end
    // of the order) is ONE step.  The parked products of a level are all alive at once, the next level reuses their places.
    size_t n_rounds = 0;
    for (const auto& c : contrib) n_rounds = std::max(n_rounds, (c.size() - 1 + 3) / 4);
    std::vector<Desc> prods;
    std::vector<std::vector<Desc>> sums(n_rounds);
    std::vector<char> dinv_done(NB, 0);
    int m_lv = 0;
    auto product = [&](const Contribution& x, int where, int z) {
      const int src = x.j == NB ? oR + 4 * x.k : oB + 4 * blk[x.k][x.j];
      prods.push_back(Desc{OP_PROD, oB + 4 * blk[x.k][x.k], oB + 4 * blk[x.i][x.k], src, where,
                           dinv_done[x.k] ? 0 : oDI + 4 * x.k, z, 0});
      dinv_done[x.k] = 1;
      dinv_stored[x.k] = 1;
    };
    for (size_t q = 0; q < dsts.size(); ++q) {
      const auto& c = contrib[q];
      const int dst = dsts[q].second == NB ? oR + 4 * dsts[q].first : oB + 4 * blk[dsts[q].first][dsts[q].second];
      product(c[0], dst, dst);
      for (size_t r = 0; 1 + 4 * r < c.size(); ++r) {
        Desc sum = {OP_SUM, dst, oZ, oZ, oZ, oZ, int(std::min<size_t>(4, c.size() - 1 - 4 * r)), 0};
        for (size_t u = 1 + 4 * r; u < std::min(c.size(), 1 + 4 * r + 4); ++u) {
          const int m = m_lv++;
          sum[2 + (u - 1 - 4 * r)] = -(m + 1);
          product(c[u], -(m + 1), oZ);
        }
        sums[r].push_back(sum);
      }
    }
    n_m = std::max(n_m, m_lv);
    for (size_t q = 0; q < prods.size(); q += G) {
      const int st = new_step(ST_PROD);
      for (size_t u = q; u < std::min(prods.size(), q + G); ++u) steps[st].push_back(prods[u]);
    }
    for (size_t r = 0; r < n_rounds; ++r)
      for (size_t q = 0; q < sums[r].size(); q += G) {
        int run = 1;
        for (size_t u = q; u < std::min(sums[r].size(), q + G); ++u) run = std::max(run, sums[r][u][6]);
        const int st = new_step(ST_SUM | (run << 8));
        for (size_t u = q; u < std::min(sums[r].size(), q + G); ++u) steps[st].push_back(sums[r][u]);
      }
  }
  if (tail_a >= 0) {
    const int st = new_step(ST_TAIL);
    steps[st].push_back(Desc{OP_TAIL2, oB + 4 * blk[tail_a][tail_a], oB + 4 * blk[tail_a][tail_b], oB + 4 * blk[tail_b][tail_a],
                             oB + 4 * blk[tail_b][tail_b], oR + 4 * tail_a, oR + 4 * tail_b, 0});
  }
  // ---- back substitution, by COLUMNS: x_k = D_k^-1 r_k is known ("ready") once every later bus j of upper[k] has
  // taken A_kj x_j out of r_k.  A step gives every row i with ready buses among its pending ones one operation
  // r_i -= A_ik x_k + A_ik' x_k' (two of them, in the order of upper[i]); the operation forms x_k from (r_k, D_k^-1)
  // itself -- nobody stores x, and a bus is ready the step after its last term went.  The steps follow the
  // dependencies, not the levels: a dense end of the order costs one step per bus (the row-wise form took up to four),
  // and whatever is ready goes together.  What is left when the program ends is x_i = D_i^-1 r_i for every bus: each
  // bus lane forms its own (k_mesh, the update).  D^-1 was stored when the bus was eliminated (or by the tail
  // step); a bus nothing was eliminated into still has D itself: the operation inverts it (OP_COL_INV1 / _INV2 bits).
  std::vector<std::vector<int>> pend(NB);
  std::vector<char> ready(NB, 0), self_inv(NB, 0);
  if (tail_a >= 0) dinv_stored[tail_a] = dinv_stored[tail_b] = 1;
  for (int k = 1; k < NB; ++k) {
    if (in_zone[k]) pend[k].assign(zstruct[k].begin(), zstruct[k].end());   // (the products took the rest of the zone out)
    else pend[k] = upper[k];
    self_inv[k] = !dinv_stored[k];
  }
  if (tail_a >= 0) pend[tail_a].clear();             // (the tail step takes A_ab x_b out of r_a)
  for (int k = 1; k < NB; ++k) ready[k] = pend[k].empty();
  for (;;) {
    std::vector<Desc> ops;
    std::vector<int> newly;
    for (int i = 1; i < NB; ++i) {
      if (pend[i].empty()) continue;
      Desc op = {OP_COL, oR + 4 * i, oZ, oZ, oZ, oZ, oZ, oZ};
      int n_take = 0;
      for (size_t u = 0; u < pend[i].size() && n_take < 2;) {
        const int k = pend[i][u];
        if (!ready[k]) { ++u; continue; }
        op[2 + 3 * n_take] = oB + 4 * blk[i][k];
        op[3 + 3 * n_take] = oR + 4 * k;
        op[4 + 3 * n_take] = self_inv[k] ? oB + 4 * blk[k][k] : oDI + 4 * k;
        if (self_inv[k]) op[0] |= (n_take == 0 ? OP_COL_INV1 : OP_COL_INV2);
        ++n_take;
        pend[i].erase(pend[i].begin() + u);
      }
      if (n_take == 0) continue;
      ops.push_back(op);
      if (pend[i].empty()) newly.push_back(i);
    }
    if (ops.empty()) break;
    for (size_t q = 0; q < ops.size(); q += G) {
      const int st = new_step(ST_COL);
      for (size_t u = q; u < std::min(ops.size(), q + G); ++u) steps[st].push_back(ops[u]);
    }
    for (int i : newly) ready[i] = 1;
  }
  for (int k = 1; k < NB; ++k)
    if (!pend[k].empty()) { err = "internal: back substitution left terms behind"; return false; }
  for (int l = 0; l + 1 < NB; ++l) I(IF_XINV, l) = self_inv[l + 1];
  d.n_m = n_m;
  d.l_dev = d.l_m + std::max(4 * n_m, 4 * d.NBR);
  // 16-byte aligned (blocks and vector pairs move as ds_read_b128 / ds_write_b128), and = 2 (mod 4) doubles: the
  // environments of a wavefront execute the same descriptors, so their areas must not start on the same banks --
  // with this size equal offsets of neighbouring environments are 4 banks (one b128 access) apart
  d.l_red = d.l_dev + 2 * d.ND;
  d.lds_per_env = d.l_red + (G > 64 ? RED_DOUBLES : 0) + 1;
  while (d.lds_per_env % 4 != 2) ++d.lds_per_env;
  if (d.lds_per_env >= 65536) { err = "network too large for the general lane-group kernel (LDS)"; return false; }
  for (auto& st : steps)
    for (Desc& q : st)
      for (int& f : q)
        if (f < 0) f = d.l_m + 4 * (-f - 1);
  d.n_steps = int(steps.size());
  d.off_stype = int(P.hi.size());
  P.hi.insert(P.hi.end(), stype.begin(), stype.end());
  while (P.hi.size() % 4) P.hi.push_back(0);     // descriptors are read as int4
  d.off_desc = int(P.hi.size());
  for (int st = 0; st < d.n_steps; ++st) {
    if (int(steps[st].size()) > G) { err = "internal: step over-subscribed"; return false; }
    for (int l = 0; l < G; ++l) {
      const Desc& q = l < int(steps[st].size()) ? steps[st][l] : none;
      for (int w = 0; w < 4; ++w) P.hi.push_back(int(unsigned(q[2 * w]) | (unsigned(q[2 * w + 1]) << 16)));
    }
  }

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
  for (int l = 0; l + 1 < NB; ++l) {
    const int b = l + 1;
    D(DF_YII_RE, l) = Y[b * NB + b].real(); D(DF_YII_IM, l) = Y[b * NB + b].imag();
    D(DF_VMIN, l) = n.bus_vmin[b]; D(DF_VMAX, l) = n.bus_vmax[b];
  }
  for (int br = 0; br < d.NBR; ++br) {
    const int f = n.br_from[br], t = n.br_to[br];
    const cplx ys(n.br_series[2 * br], n.br_series[2 * br + 1]), sh(n.br_shunt[2 * br], n.br_shunt[2 * br + 1]);
    const cplx tap(n.br_tap[2 * br], n.br_tap[2 * br + 1]);
    auto DB = [&](int fld) -> double& { return D(fld + (br / G) * DF_BR_FIELDS, br % G); };
    DB(DF_YFT_RE) = Y[f * NB + t].real(); DB(DF_YFT_IM) = Y[f * NB + t].imag();
    DB(DF_YTF_RE) = Y[t * NB + f].real(); DB(DF_YTF_IM) = Y[t * NB + f].imag();
    const cplx c4[4] = {(ys + sh) / (std::abs(tap) * std::abs(tap)), -ys / std::conj(tap), ys + sh, -ys / tap};
    for (int j = 0; j < 4; ++j) { DB(DF_BRC + 2 * j) = c4[j].real(); DB(DF_BRC + 2 * j + 1) = c4[j].imag(); }
    DB(DF_BRC + 8) = n.br_rate[br];
  }
  for (int k = 0; k < d.ND; ++k) pack_device(n, k, &P.hd[d.off_dev + k * SD_SIZE]);

  d.n_stage = int(P.hi.size()) - d.off_lists;   // what a workgroup stages in LDS: lists, fill ids, the program
  if (lds_bytes(d, 1, false) > 160 * 1024) {
    err = "network too large for the general lane-group kernel (LDS)";
    return false;
  }
  return true;
}

// wavefronts a compute unit holds of a plan (LDS, registers)
inline size_t waves_per_cu(const Dims& d) { return launch_of(d).waves_per_cu; }

// The plan of a network: the program with the fewest steps per Newton trip among
//   * the zones "from level z on" (none, the last level, the last two ...; a search down the levels stops when five in a
//     row bring nothing), and
//   * for each, levels as the dependencies give them (a destination may collect several contributions per level: one
//     sums step per four of them) or levels that admit ONE contribution per destination (no sums steps at all: what
//     trees like -- a 30-bus feeder 10 -> 7 steps),
// as long as the extra fill costs no wavefront on a compute unit.  Deterministic in the topology (the parameter classes of
// a model share their integer tables) and in the tuning switches ANM_MESH_WAVES / ANM_MESH_NO_ZONE / ANM_MESH_NO_TAIL, which
// waves_per_cu and build_plan_zone read from the environment.
inline bool build_plan(const anm_network_desc& n, Plan& P, std::string& err) {
  if (!build_plan_zone(n, P, err, 1 << 30)) return false;
  if (getenv("ANM_MESH_NO_ZONE")) return true;
  const size_t base_waves = waves_per_cu(P.d);
  const Plan base = P;   // (the c_max = 1 << 30 search starts from the plan just built: no second build of it)
  for (int c_max : {1 << 30, 1}) {
    Plan Q0;
    std::string e0;
    if (c_max == (1 << 30)) Q0 = base;
    else if (!build_plan_zone(n, Q0, e0, 1 << 30, c_max)) continue;
    if (Q0.d.n_steps < P.d.n_steps && waves_per_cu(Q0.d) >= base_waves) P = Q0;
    int misses = 0;
    for (int z = Q0.d.zone_level - 1; z >= 0 && misses < 5; --z) {   // (zone_level of "no zone" = one past the last product level)
      Plan Q;
      std::string e2;
      if (!build_plan_zone(n, Q, e2, z, c_max)) break;             // (more fill than the LDS takes)
      if (Q.d.n_steps < P.d.n_steps && waves_per_cu(Q.d) >= base_waves) { P = Q; misses = 0; }
      else ++misses;
    }
  }
  return true;
}

#if defined(__HIPCC__)

// one environment's lanes meet: a wavefront fence, or (WG) the workgroup's barrier
#define ANM_MESH_SYNC()                        \
  do {                                         \
    if constexpr (WG) __syncthreads();         \
    else ANM_WAVE_SYNC();                      \
  } while (0)

// PG: see k_radial; WG: the environment is the workgroup; SW: wavefronts per SIMD the registers are budgeted for; TL: the tables
// are staged in LDS (Launch)
template <class JT, bool PG = false, bool WG = false, int SW = 2, bool TL = !WG>
#ifndef ANM_MESH_MINWAVES2
// The waves-per-SIMD hint of the SW = 2 variants: none (1).  Round 6 traced most of the 522 -> 546 us of the meshed 30-bus batch
// between rounds 4 and 5 to `__launch_bounds__(256, 2)` (round 4 compiled without a hint): the same 192 registers either way,
// but hipcc's schedule of the trip is 7-18 us slower with it (profiles/r06_a_mesh_bisect.txt, r06_b_mesh_regression.txt).
// CAUTION, measured the hard way: this kernel sits on a code-generation fault of hipcc 7.2.  With the outputs section written
// as in round 5 (dump and list observation called from each branch) and the kernel arguments of this round, the variant with
// its tables in LDS returned wrong branch_p entries and rewards -- V, I and every other quantity bit-equal to the other
// variants -- on the star / complete / five-feeder test networks without the hint, and faulted on the complete graph with it;
// with ONE call site each (below: lo_mode / dump) every variant passes the whole GPU tier under either setting.  Any edit here
// is followed by tests/test_gpu_headline.py (structured topologies, launch variants bit-identical) before anything else.
#define ANM_MESH_MINWAVES2 1
#endif
__global__ __launch_bounds__(WG ? 512 : 256, SW == 2 ? ANM_MESH_MINWAVES2 : SW) void k_mesh(Dims d, const int* __restrict__ ri, const double* __restrict__ rd0,
                                                              radial::IO io, SolverOpts so, int64_t n_env, ClassSel cls) {
  // a workgroup = 1, 2 or 4 wavefronts that share one LDS copy of the tables and otherwise never meet: a lane
  // group lies within one wavefront, whose LDS operations complete in program order (fences only).
  // WG: the workgroup (2 ... 8 wavefronts) is ONE environment; every place where lanes of an environment hand
  // something over is a workgroup barrier, and the tables stay in global memory
  extern __shared__ __align__(16) double sh_dyn[];
  const int t = threadIdx.x;
  const int G = d.G;
  const int l = t & (G - 1);
  const int per_wave = WG ? 0 : 64 / G;
  const int per_block = WG ? 1 : int(blockDim.x) / G;
  const int grp = WG ? 0 : t / G;
  const int wave = t >> 6, n_waves = int(blockDim.x) >> 6;
  // slot of this launch -> environment: itself, or (a view is bound: the launch serves a SUB-batch of a larger batch whose
  // rows are padded to common widths, anm_model_bind_view) the index the view names; the row strides of the batch arrays
  // are the network's own widths unless the view gives others
  const int64_t slot_e = int64_t(blockIdx.x) * per_block + grp;
  const bool env_ok = slot_e < n_env;
  const int64_t ee = io.v.index ? int64_t(io.v.index[env_ok ? slot_e : 0]) : (env_ok ? slot_e : 0);
  const int64_t e = ee;   // (lanes of a slot beyond the batch compute on another environment and never store)
  const int W_LOAD = io.v.w_load > 0 ? io.v.w_load : d.NLOAD, W_GEN = io.v.w_gen > 0 ? io.v.w_gen : d.NGEN;
  const int W_SET = io.v.w_set > 0 ? io.v.w_set : d.NSET, W_DES = io.v.w_des > 0 ? io.v.w_des : d.NDES;
  const int W_ACT = io.v.w_action > 0 ? io.v.w_action : 2 * (d.NGEN + d.NDES);
  const int W_ST = io.v.w_state > 0 ? io.v.w_state : d.SDIM + io.e.K;
  const int W_EXO = io.v.w_exo > 0 ? io.v.w_exo : d.NLOAD + d.NGEN, W_AUX = io.v.w_aux > 0 ? io.v.w_aux : io.e.K;
  const int W_FS = io.v.w_full > 0 ? io.v.w_full : d.FS;
  const int64_t first_env = int64_t(blockIdx.x) * per_block + (WG ? 0 : (t >> 6) * per_wave);
  const double* __restrict__ rd =
      PG ? rd0 + int64_t(cls.env_class[ee]) * cls.stride
         : rd0 + int64_t(__builtin_amdgcn_readfirstlane(cls.env_class[(first_env < n_env ? first_env : 0) * cls.per_env])) * cls.stride;
  double* S = sh_dyn + grp * d.lds_per_env;                 // this environment's LDS
  double* RED = S + d.l_red;                                // WG: flags and partial sums of the wavefronts
  const int* tab;                                           // lists, fill ids, the program
  if constexpr (WG || !TL) {
    tab = ri + d.off_lists;
  } else {
    int* stage = reinterpret_cast<int*>(sh_dyn + per_block * d.lds_per_env);   // (shared by the workgroup)
    for (int k = t; k < d.n_stage; k += int(blockDim.x)) stage[k] = ri[d.off_lists + k];
    __syncthreads();
    tab = stage;
  }
  // sums over the lanes of the environment, the same value in every lane: butterflies; WG: per wavefront, then the
  // wavefronts' partial sums in wavefront order
  auto group_sum2 = [&](double& a, double& b) {
    if constexpr (WG) {
      for (int m = 1; m < 64; m <<= 1) {
        a += __shfl_xor(a, m, 64);
        b += __shfl_xor(b, m, 64);
      }
      __syncthreads();
      if ((t & 63) == 0) { RED[32 + 2 * wave] = a; RED[33 + 2 * wave] = b; }
      __syncthreads();
      a = 0.0; b = 0.0;
      for (int w = 0; w < n_waves; ++w) { a += RED[32 + 2 * w]; b += RED[33 + 2 * w]; }
    } else {
      for (int m = 1; m < G; m <<= 1) {
        a += __shfl_xor(a, m, G);
        b += __shfl_xor(b, m, G);
      }
    }
  };
  const int* lists = tab;
  const int* fills = tab + (d.off_fill - d.off_lists);
  auto RI = [&](int f) { return ri[f * G + l]; };
  auto RD = [&](int f) { return rd[d.off_lane + f * G + l]; };
  cptr_t C = (cptr_t)rd;
  const double base = rd[SF_BASE], dt = rd[SF_DT];

  const bool isbus = l < d.NB - 1;
  bool isbr[BR_SLOTS];
  int br_f[BR_SLOTS], br_t[BR_SLOTS], blk_ft[BR_SLOTS], blk_tf[BR_SLOTS], pos_f[BR_SLOTS], pos_t[BR_SLOTS];
  const int bus = l + 1;
  const int level = RI(IF_LEVEL), diag = RI(IF_DIAG);
  const bool xinv = RI(IF_XINV) != 0;
  const int inc_beg = RI(IF_INC_BEG), inc_end = RI(IF_INC_END);
  const int inc0 = RI(IF_INC0), deg = RI(IF_DEG);
#pragma unroll
  for (int sl = 0; sl < BR_SLOTS; ++sl) {
    isbr[sl] = sl < d.br_slots && sl * G + l < d.NBR;
    br_f[sl] = br_t[sl] = 0; blk_ft[sl] = blk_tf[sl] = -1; pos_f[sl] = pos_t[sl] = 0;
    if (sl < d.br_slots) {
      br_f[sl] = RI(IF_BR_F + sl * IF_BR_FIELDS); br_t[sl] = RI(IF_BR_T + sl * IF_BR_FIELDS);
      blk_ft[sl] = RI(IF_BR_BLK_FT + sl * IF_BR_FIELDS); blk_tf[sl] = RI(IF_BR_BLK_TF + sl * IF_BR_FIELDS);
      pos_f[sl] = RI(IF_BR_POS_F + sl * IF_BR_FIELDS); pos_t[sl] = RI(IF_BR_POS_T + sl * IF_BR_FIELDS);
    }
  }
  const int typ = RI(IF_DEV_TYPE), slot = RI(IF_DEV_SLOT), sset = RI(IF_DEV_SET);
  const int mode = io.mode;
  const int K = io.e.K;
  const int SD_ = d.SDIM + K;

  // ---------------- inputs per device lane (as anm_radial.hpp) ----------------------------------
  bool skip = false, resetting = false, sampled = false;
  int aux = 0;
  double in_p = 0.0, in_q = 0.0, in_pot = 0.0, soc = 0.0, s0_q = 0.0, soc_req = 0.0;
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
    double s0_p = 0.0, s0_pm = 0.0;
    sampled = resetting && !(mode == 1 && io.e.init_state);
    if (mode == 1 && io.e.init_state) {
      const double* s0 = io.e.init_state + ee * W_ST;
      if (typ != DEV_NONE) { s0_p = s0[l]; s0_q = s0[d.ND + l]; }
      if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) s0_pm = s0[2 * d.ND + d.NDES + slot];
      if (typ == DEV_STORAGE) soc_req = s0[2 * d.ND + slot];
    } else if (resetting) {
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
      in_p = s0_p; in_q = s0_q; in_pot = s0_pm;
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

  // ---------------- device maps (lane = device), bus injection sums (lane = bus) ------------------
  double dev_p = 0.0, dev_q = 0.0, p_pot = 0.0;
  {
    cptr_t sd = C + d.off_dev + l * SD_SIZE;
    if (typ == DEV_LOAD) {
      const double p = fmin(fmax(in_p / base, sd[0]), sd[1]);
      dev_p = p;
      dev_q = p * sd[2];
    } else if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE || typ == DEV_STORAGE) {
      // one projection for both kinds (anm_radial.hpp: a generator's lines 3 and 4 are switched off in its table)
      const bool des = typ == DEV_STORAGE;
      const double eff = sd[SD_EFF];
      double pl = sd[SD_PMIN], pu = sd[SD_PMAX];
      if (des) {
        pl = fmax(pl, (soc - sd[SD_SOC_MAX]) / (dt * eff));
        pu = fmin(pu, eff * (soc - sd[SD_SOC_MIN]) / dt);
      } else {
        p_pot = fmin(fmax(in_pot / base, sd[SD_PMIN]), sd[SD_PMAX]);
        pu = fmin(pu, p_pot);
      }
      project_pq<4>(sd, in_p / base, in_q / base, pl, pu, dev_p, dev_q);
      if (des) {
        const double ns = (dev_p <= 0.0) ? (soc - dt * eff * dev_p) : (soc - dt * dev_p / eff);
        soc = fmin(fmax(ns, sd[SD_SOC_MIN]), sd[SD_SOC_MAX]);
      }
    }
  }
  double* Ldev = S + d.l_dev;
  if (l < d.ND) { Ldev[l] = dev_p; Ldev[d.ND + l] = dev_q; }
  ANM_MESH_SYNC();
  double bus_p = 0.0, bus_q = 0.0;
  if (isbus)
    for (int k = RI(IF_BD_BEG); k < RI(IF_BD_END); ++k) {
      const int dd = lists[k];
      bus_p += Ldev[dd];
      bus_q += Ldev[d.ND + dd];
    }

  // ---------------- Newton-Raphson ---------------------------------------------------------------
  double* LV = S + d.l_v;       // vr[NB], vi[NB]
  double* LBW = S + d.l_bw;     // wft_r, wft_i, wtf_r, wtf_i [NBR] each
  double* LBLK = S + d.l_blk;   // [NBLK][4]
  double* LR = S + d.l_r;       // r[NB] as blocks (r0, 0; r1, 0)
  const int NB = d.NB, NBR = d.NBR;
  const double yii_r = RD(DF_YII_RE), yii_i = RD(DF_YII_IM);
  double yft_r[BR_SLOTS], yft_i[BR_SLOTS], ytf_r[BR_SLOTS], ytf_i[BR_SLOTS];
#pragma unroll
  for (int sl = 0; sl < BR_SLOTS; ++sl) {
    yft_r[sl] = yft_i[sl] = ytf_r[sl] = ytf_i[sl] = 0.0;
    if (sl < d.br_slots) {
      yft_r[sl] = RD(DF_YFT_RE + sl * DF_BR_FIELDS); yft_i[sl] = RD(DF_YFT_IM + sl * DF_BR_FIELDS);
      ytf_r[sl] = RD(DF_YTF_RE + sl * DF_BR_FIELDS); ytf_i[sl] = RD(DF_YTF_IM + sl * DF_BR_FIELDS);
    }
  }
  double vm = 1.0, cs = 1.0, sn = 0.0, vr = 1.0, vi = 0.0;
  if (mode == 0 && io.t.nr_start && isbus) {   // anm_model_bind_nr_start: the reference solver's v_guess instead of the flat start
    const double* x0 = io.t.nr_start + ee * (2 * (d.NB - 1));
    vm = x0[d.NB - 1 + l];
    const SinCos r0 = sincos_huge(x0[l]);
    sn = r0.s; cs = r0.c;
  }
  int it = 0;
  bool g_bad = false, g_nan = false;   // the group's ||F||inf > tol / F has a NaN, as of its last evaluation
  bool active = true;
  const unsigned long long busm = __builtin_amdgcn_uicmp(isbus ? 1u : 0u, 0u, group::ICMP_NE);
  const unsigned long long gmask = (G >= 64) ? ~0ull : (((1ull << G) - 1ull) << ((t & 63) - l));
  const unsigned glo = unsigned(gmask), ghi = unsigned(gmask >> 32);
  if (l < 6) S[d.l_zero + l] = 0.0;
  for (int k = l; k < 4 * NB; k += G) LR[k] = 0.0;                // second column of the right-hand-side blocks
  auto put_blk = [&](int b, double a, double bb, double c, double dd) {
    double* p = LBLK + 4 * b;
    p[0] = a; p[1] = bb; p[2] = c; p[3] = dd;
  };
  // blocks and (x0, x1) pairs sit at even offsets of a 16-byte aligned area: 128-bit LDS accesses (half the bank
  // conflicts of 64-bit ones on these 32-byte-strided gathers)
  auto ld2 = [&](int o) { return *reinterpret_cast<const double2*>(S + o); };
  auto st2 = [&](int o, double a, double b) { *reinterpret_cast<double2*>(S + o) = double2{a, b}; };
  auto ld4 = [&](int o) {
    const double2 lo = ld2(o), hi = ld2(o + 2);
    return Blk<JT>{JT(lo.x), JT(lo.y), JT(hi.x), JT(hi.y)};
  };
  auto st4 = [&](int o, const Blk<JT>& m) {
    st2(o, double(m.a), double(m.b));
    st2(o + 2, double(m.c), double(m.d));
  };
  // a right-hand side r_i = the first column of its block (r0, 0; r1, 0)
  auto ldr = [&](int o) { return double2{S[o], S[o + 2]}; };
  auto str = [&](int o, double a, double b) { S[o] = a; S[o + 2] = b; };
  // the step program (see OpKind): descriptor and type of the next step are fetched while this one runs
  const int4* prog = reinterpret_cast<const int4*>(tab + (d.off_desc - d.off_lists));
  const int* stype = tab + (d.off_stype - d.off_lists);
  int4 nxt = prog[l];
  int nxt_ty = stype[0];
  int un[8];
  auto unpack = [](const int4& q, int (&o)[8]) {
    o[0] = q.x & 0xffff; o[1] = int(unsigned(q.x) >> 16); o[2] = q.y & 0xffff; o[3] = int(unsigned(q.y) >> 16);
    o[4] = q.z & 0xffff; o[5] = int(unsigned(q.z) >> 16); o[6] = q.w & 0xffff; o[7] = int(unsigned(q.w) >> 16);
  };
  unpack(nxt, un);
#if defined(ANM_PHASE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
  // tuning builds: cycles per part of the Newton trip, summed over the trips of a wavefront
  // (scripts/mesh_trip_times.py): 0 publish V + branch products, 1 bus sums + stop test + diagonal, 2 product steps,
  // 3 sum steps, 4 the tail step, 5 back substitution, 6 update, 7 number of trips
  unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ph_t = __builtin_readcyclecounter();
#define ANM_MESH_ACC(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); ph_acc[k] += n_ - ph_t; ph_t = n_; } while (0)
#else
#define ANM_MESH_ACC(k) do { } while (0)
#endif
  for (;;) {
    vr = vm * cs;
    vi = vm * sn;
    if (isbus) { LV[bus] = vr; LV[NB + bus] = vi; }
    if (l == 0) { LV[0] = 1.0; LV[NB] = 0.0; }   // slack bus: V_0 = 1 (every trip: the area also serves as x)
    ANM_MESH_SYNC();
    // ---- branch lanes: W_ft = V_f conj(Y_ft V_t), W_tf = V_t conj(Y_tf V_f) and the two off-diagonal blocks
#pragma unroll
    for (int sl = 0; sl < BR_SLOTS; ++sl)
      if (isbr[sl]) {
        const double vfr = LV[br_f[sl]], vfi = LV[NB + br_f[sl]], vtr = LV[br_t[sl]], vti = LV[NB + br_t[sl]];
        const double pr = fma(vfr, vtr, vfi * vti), pim = fma(vfi, vtr, -(vfr * vti));   // P = V_f conj(V_t)
        const double wft_r = fma(yft_r[sl], pr, yft_i[sl] * pim), wft_i = fma(yft_r[sl], pim, -(yft_i[sl] * pr));
        const double wtf_r = fma(ytf_r[sl], pr, -(ytf_i[sl] * pim)), wtf_i = -fma(ytf_r[sl], pim, ytf_i[sl] * pr);
        LBW[2 * pos_f[sl]] = wft_r; LBW[2 * pos_f[sl] + 1] = wft_i; LBW[2 * pos_t[sl]] = wtf_r; LBW[2 * pos_t[sl] + 1] = wtf_i;
        if (blk_ft[sl] >= 0) {
          put_blk(blk_ft[sl], wft_i, wft_r, -wft_r, wft_i);
          put_blk(blk_tf[sl], wtf_i, wtf_r, -wtf_r, wtf_i);
        }
      }
    for (int k = l; k < d.n_fill; k += G) put_blk(fills[k], 0.0, 0.0, 0.0, 0.0);
    ANM_MESH_SYNC();
    ANM_MESH_ACC(0);
    // ---- bus lanes: S_i = W_ii + sum over the incident branches, mismatch, diagonal block
    const double m2 = vm * vm;
    const double wii_r = yii_r * m2, wii_i = -(yii_i * m2);
    double sr = wii_r, si = wii_i;
    {
      // the first four terms are fetched together (most buses have no more); what lies behind a bus's own
      // entries is somebody else's, read and dropped
      double er[4], ei[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { er[k] = LBW[2 * (inc0 + k)]; ei[k] = LBW[2 * (inc0 + k) + 1]; }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < deg) { sr += er[k]; si += ei[k]; }
      for (int k = 4; k < d.max_deg; ++k)
        if (k < deg) {
          sr += LBW[2 * (inc0 + k)];
          si += LBW[2 * (inc0 + k) + 1];
        }
    }
    const double fr = sr - bus_p, fi = si - bus_q;
    // group-wide stop test on lane masks (as anm_group.hpp): "||F||inf > tol" and "F has a NaN" are all the
    // reference's loop and flags need; a group that stopped keeps its verdict
    const unsigned long long badm = __builtin_amdgcn_fcmp(isbus ? fmax(fabs(fr), fabs(fi)) : 0.0, so.tol, group::FCMP_UGT);
    const unsigned long long nanm = __builtin_amdgcn_fcmp(fr, fi, group::FCMP_UNO) & busm;
    bool nb = ((unsigned(badm) & glo) | (unsigned(badm >> 32) & ghi)) != 0u;
    bool nn = ((unsigned(nanm) & glo) | (unsigned(nanm >> 32) & ghi)) != 0u;
    if constexpr (WG) {   // the verdicts of the other wavefronts (the flags are next written a barrier later at least)
      if ((t & 63) == 0) { RED[2 * wave] = nb ? 1.0 : 0.0; RED[2 * wave + 1] = nn ? 1.0 : 0.0; }
      __syncthreads();
      double fb = 0.0, fn = 0.0;
      for (int w = 0; w < n_waves; ++w) { fb += RED[2 * w]; fn += RED[2 * w + 1]; }
      nb = fb > 0.0;
      nn = fn > 0.0;
    }
    if (it == 0 || active) { g_bad = nb; g_nan = nn; }
    active = g_bad && !g_nan && (it < so.max_iter);       // NaN > tol is false, like the reference
    if (!__any(active && env_ok && !skip)) break;
    if (isbus) {
      put_blk(diag, -(si - wii_i), sr + wii_r, sr - wii_r, si + wii_i);
      LR[4 * bus] = double(JT(fr)); LR[4 * bus + 2] = double(JT(fi));
    }
    ANM_MESH_SYNC();
    ANM_MESH_ACC(1);
    // ---- elimination and back substitution: the step program
    for (int sidx = 0; sidx < d.n_steps; ++sidx) {
      // the operands of this step were unpacked while the previous step's stores drained (below); the descriptor of
      // the next one is fetched now
      const int ty = __builtin_amdgcn_readfirstlane(nxt_ty);
      const int sn = (sidx + 1 == d.n_steps) ? 0 : sidx + 1;
      const int kind = un[0], o1 = un[1], o2 = un[2], o3 = un[3], o4 = un[4], o5 = un[5], o6 = un[6], o7 = un[7];
      nxt = prog[sn * G + l];
      nxt_ty = stype[sn];
      if ((ty & 0xff) == ST_PROD) {
        if (kind != OP_NONE) {
          const Blk<JT> Di = blk_inv(ld4(o1));
          const Blk<JT> Lik = blk_mul(ld4(o2), Di);
          Blk<JT> Z = ld4(o6);                        // the destination itself (first contribution) or the zero block
          const Blk<JT> M = blk_mul(Lik, ld4(o3));
          Z.a -= M.a; Z.b -= M.b; Z.c -= M.c; Z.d -= M.d;
          st4(o4, Z);
          if (o5 != 0) st4(o5, Di);
        }
      } else if ((ty & 0xff) == ST_SUM) {
        if (kind != OP_NONE) {
          Blk<JT> Z = ld4(o1);
          const Blk<JT> M0 = ld4(o2);
          Z.a += M0.a; Z.b += M0.b; Z.c += M0.c; Z.d += M0.d;
          if ((ty >> 8) > 1) {
            const Blk<JT> M1 = ld4(o3);
            Z.a += M1.a; Z.b += M1.b; Z.c += M1.c; Z.d += M1.d;
          }
          if ((ty >> 8) > 2) {
            const Blk<JT> M2 = ld4(o4), M3 = ld4(o5);
            Z.a += M2.a; Z.b += M2.b; Z.c += M2.c; Z.d += M2.d;
            Z.a += M3.a; Z.b += M3.b; Z.c += M3.c; Z.d += M3.d;
          }
          st4(o1, Z);
        }
      } else if ((ty & 0xff) == ST_TAIL) {
        if (kind != OP_NONE) {
          const Blk<JT> Dai = blk_inv(ld4(o1)), Aab = ld4(o2);
          const double2 ra = ldr(o5), rb = ldr(o6);
          const Blk<JT> L = blk_mul(ld4(o3), Dai);
          Blk<JT> Db = ld4(o4);
          blk_submul(Db, L, Aab);
          const JT ra0 = JT(ra.x), ra1 = JT(ra.y);
          const JT rb0 = fm(-L.b, ra1, fm(-L.a, ra0, JT(rb.x))), rb1 = fm(-L.d, ra1, fm(-L.c, ra0, JT(rb.y)));
          const Blk<JT> Dbi = blk_inv(Db);
          const JT xb0 = fm(Dbi.a, rb0, Dbi.b * rb1), xb1 = fm(Dbi.c, rb0, Dbi.d * rb1);
          const JT a0 = fm(-Aab.b, xb1, fm(-Aab.a, xb0, ra0)), a1 = fm(-Aab.d, xb1, fm(-Aab.c, xb0, ra1));
          str(o6, double(rb0), double(rb1));
          st4(o6 - d.l_r + d.l_dinv, Dbi);
          str(o5, double(a0), double(a1));
          st4(o5 - d.l_r + d.l_dinv, Dai);
        }
      } else {
        if (kind != OP_NONE) {
          const double2 ri = ldr(o1), rk0 = ldr(o3), rk1 = ldr(o6);
          const Blk<JT> A0 = ld4(o2), A1 = ld4(o5);
          Blk<JT> D0 = ld4(o4), D1 = ld4(o7);
          if (kind & OP_COL_INV1) D0 = blk_inv(D0);
          if (kind & OP_COL_INV2) D1 = blk_inv(D1);
          const JT x00 = fm(D0.a, JT(rk0.x), D0.b * JT(rk0.y)), x01 = fm(D0.c, JT(rk0.x), D0.d * JT(rk0.y));
          const JT x10 = fm(D1.a, JT(rk1.x), D1.b * JT(rk1.y)), x11 = fm(D1.c, JT(rk1.x), D1.d * JT(rk1.y));
          JT a0 = JT(ri.x), a1 = JT(ri.y);
          a0 = fm(-A0.b, x01, fm(-A0.a, x00, a0));
          a1 = fm(-A0.d, x01, fm(-A0.c, x00, a1));
          a0 = fm(-A1.b, x11, fm(-A1.a, x10, a0));
          a1 = fm(-A1.d, x11, fm(-A1.c, x10, a1));
          str(o1, double(a0), double(a1));
        }
      }
      unpack(nxt, un);   // (before the fence: in the shadow of this step's stores)
      ANM_MESH_SYNC();
#if defined(ANM_PHASE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
      if ((ty & 0xff) == ST_PROD) ANM_MESH_ACC(2);
      else if ((ty & 0xff) == ST_SUM) ANM_MESH_ACC(3);
      else if ((ty & 0xff) == ST_TAIL) ANM_MESH_ACC(4);
      else ANM_MESH_ACC(5);
#endif
    }
    // ---- update (group-uniform `active`); d1 is the relative magnitude step
    // x_i = D_i^-1 r_i: what the program leaves is every bus's final r and (unless nothing was eliminated into it) D^-1
    double dth = 0.0, d1 = 0.0;
    if (active && isbus) {
      const double2 rr = ldr(d.l_r + 4 * bus);
      Blk<JT> Di = ld4(xinv ? d.l_blk + 4 * diag : d.l_dinv + 4 * bus);
      if (xinv) Di = blk_inv(Di);
      dth = double(fm(Di.a, JT(rr.x), Di.b * JT(rr.y)));
      d1 = double(fm(Di.c, JT(rr.x), Di.d * JT(rr.y)));
    }
    if (active && isbus) {
      vm = fma(-d1, fabs(vm), vm);
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
#if defined(ANM_PHASE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    ANM_MESH_ACC(6);
    ph_acc[7] += 1;
#endif
  }
#if defined(ANM_PHASE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
  {
    const int slot_w = WG ? int(blockIdx.x) : int(blockIdx.x) * n_waves + wave;   // WG: the workgroup's first wavefront speaks for it
    if ((t & 63) == 0 && (!WG || t == 0) && slot_w < ANM_MAX_WAVES)
      for (int k = 0; k < 8; ++k) g_anm_phase[k][slot_w] = ph_acc[k];
  }
#endif
  const bool f_nan = g_nan;
  const bool converged = !g_nan && !g_bad;

  // ---------------- currents, slack injection, branch flows, reward -------------------------------
  // (LV holds the final V: the loop left through its break right after publishing it)
  ANM_MESH_SYNC();
  // branch lanes: the two terms of I = Y V their branch contributes, Y_ft V_t (to bus f) and Y_tf V_f (to bus t)
  double vfr[BR_SLOTS], vfi[BR_SLOTS], vtr[BR_SLOTS], vti[BR_SLOTS];
#pragma unroll
  for (int sl = 0; sl < BR_SLOTS; ++sl) {
    vfr[sl] = 1.0; vfi[sl] = 0.0; vtr[sl] = 1.0; vti[sl] = 0.0;
    if (isbr[sl]) {
      const int b = sl * G + l;
      vfr[sl] = LV[br_f[sl]]; vfi[sl] = LV[NB + br_f[sl]]; vtr[sl] = LV[br_t[sl]]; vti[sl] = LV[NB + br_t[sl]];
      LBW[b] = fma(yft_r[sl], vtr[sl], -(yft_i[sl] * vti[sl])); LBW[NBR + b] = fma(yft_r[sl], vti[sl], yft_i[sl] * vtr[sl]);
      LBW[2 * NBR + b] = fma(ytf_r[sl], vfr[sl], -(ytf_i[sl] * vfi[sl])); LBW[3 * NBR + b] = fma(ytf_r[sl], vfi[sl], ytf_i[sl] * vfr[sl]);
    }
  }
  ANM_MESH_SYNC();
  double ir = fma(yii_r, vr, -(yii_i * vi)), ii = fma(yii_r, vi, yii_i * vr);
  for (int k = inc_beg; k < inc_end; ++k) {
    const int code = lists[k];
    const int br = code >> 1, side = code & 1;
    ir += LBW[(2 * side) * NBR + br];
    ii += LBW[(2 * side + 1) * NBR + br];
  }
  // ||F||inf of the final iterate (anm_model_bind_nr_diff; see anm_radial.hpp): once, from the final V and I = Y V
  double* const nr_diff = (mode == 0) ? io.t.nr_diff : ((mode == 1) ? io.e.nr_diff : nullptr);
  double fdiff = 0.0;
  if (nr_diff) {
    const double fa = fabs(fma(vr, ir, vi * ii) - bus_p), fb = fabs(fma(vi, ir, -(vr * ii)) - bus_q);
    double mx = isbus ? fmax(fa, fb) : 0.0;                            // (fmax drops a NaN: counted separately)
    double nn = (isbus && (fa != fa || fb != fb)) ? 1.0 : 0.0;
    if constexpr (WG) {
      for (int m = 1; m < 64; m <<= 1) {
        mx = fmax(mx, __shfl_xor(mx, m, 64));
        nn += __shfl_xor(nn, m, 64);
      }
      __syncthreads();
      if ((t & 63) == 0) { RED[32 + 2 * wave] = mx; RED[33 + 2 * wave] = nn; }
      __syncthreads();
      mx = 0.0; nn = 0.0;
      for (int w = 0; w < n_waves; ++w) { mx = fmax(mx, RED[32 + 2 * w]); nn += RED[33 + 2 * w]; }
      __syncthreads();   // (the slots serve group_sum2 next)
    } else {
      for (int m = 1; m < G; m <<= 1) {
        mx = fmax(mx, __shfl_xor(mx, m, G));
        nn += __shfl_xor(nn, m, G);
      }
    }
    fdiff = (nn > 0.0) ? NAN : mx;
  }
  // I_0 = Y_00 + sum over the branches at the slack bus (fixed butterfly order)
  double s0r = 0.0, s0i = 0.0;
#pragma unroll
  for (int sl = 0; sl < BR_SLOTS; ++sl) {
    const int b = sl * G + l;
    if (isbr[sl] && br_f[sl] == 0) { s0r += LBW[b]; s0i += LBW[NBR + b]; }
    if (isbr[sl] && br_t[sl] == 0) { s0r += LBW[2 * NBR + b]; s0i += LBW[3 * NBR + b]; }
  }
  group_sum2(s0r, s0i);
  const double i0r = rd[SF_Y00_RE] + s0r, i0i = rd[SF_Y00_IM] + s0i;
  const double slack_p = (i0r != i0r) ? INFINITY : i0r;
  const double slack_q = (i0i != i0i) ? INFINITY : -i0i;
  if (typ == DEV_SLACK) { dev_p = slack_p; dev_q = slack_q; }

  double br_pf[BR_SLOTS], br_qf[BR_SLOTS], br_s[BR_SLOTS], br_ifr[BR_SLOTS], br_ifi[BR_SLOTS], pen = 0.0;
#pragma unroll
  for (int sl = 0; sl < BR_SLOTS; ++sl) {
    br_pf[sl] = br_qf[sl] = br_s[sl] = br_ifr[sl] = br_ifi[sl] = 0.0;
    if (isbr[sl]) {
      const int o = sl * DF_BR_FIELDS;
      const double c0 = RD(DF_BRC + 0 + o), c1 = RD(DF_BRC + 1 + o), c2 = RD(DF_BRC + 2 + o), c3 = RD(DF_BRC + 3 + o);
      const double c4 = RD(DF_BRC + 4 + o), c5 = RD(DF_BRC + 5 + o), c6 = RD(DF_BRC + 6 + o), c7 = RD(DF_BRC + 7 + o);
      br_ifr[sl] = c0 * vfr[sl] - c1 * vfi[sl] + c2 * vtr[sl] - c3 * vti[sl];
      br_ifi[sl] = c0 * vfi[sl] + c1 * vfr[sl] + c2 * vti[sl] + c3 * vtr[sl];
      const double itr = c4 * vtr[sl] - c5 * vti[sl] + c6 * vfr[sl] - c7 * vfi[sl];
      const double iti = c4 * vti[sl] + c5 * vtr[sl] + c6 * vfi[sl] + c7 * vfr[sl];
      br_pf[sl] = vfr[sl] * br_ifr[sl] + vfi[sl] * br_ifi[sl];
      br_qf[sl] = vfi[sl] * br_ifr[sl] - vfr[sl] * br_ifi[sl];
      const double pt = vtr[sl] * itr + vti[sl] * iti, qt = vti[sl] * itr - vtr[sl] * iti;
      const double sf2 = br_pf[sl] * br_pf[sl] + br_qf[sl] * br_qf[sl], st2 = pt * pt + qt * qt;
      const double sgn = (br_pf[sl] > 0.0) ? 1.0 : ((br_pf[sl] < 0.0) ? -1.0 : ((br_pf[sl] == 0.0) ? 0.0 : NAN));
      const double smax = (sf2 != sf2 || st2 != st2) ? NAN : sqrt(fmax(sf2, st2));
      br_s[sl] = sgn * smax;
      const double over = fabs(br_s[sl]) - RD(DF_BRC + 8 + o);
      pen += (over != over) ? NAN : fmax(0.0, over);
    }
  }
  if (isbus) {
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
  group_sum2(pen, el);
  const double e_loss = el * dt, penalty = pen * (dt * rd[SF_LAMB]);
  const double reward = -(e_loss + penalty);

  if (!env_ok) return;

  // ---------------- outputs (as anm_radial.hpp; branch quantities come from the branch lanes) ------
  double* full = (mode == 0) ? io.t.full : io.e.full;
  auto write_full = [&]() {
    if (!full) return;
    double* f = full + e * W_FS;
    if (isbus) {
      f[d.f_bus_p + bus] = bus_p; f[d.f_bus_q + bus] = bus_q;
      f[d.f_bus_vm + bus] = dump_abs(vr, vi); f[d.f_bus_va + bus] = dump_arg(vi, vr);
      f[d.f_bus_im + bus] = dump_abs(ir, ii); f[d.f_bus_ia + bus] = dump_arg(ii, ir);
    }
#pragma unroll
    for (int sl = 0; sl < BR_SLOTS; ++sl)
      if (isbr[sl]) {
        const int b = sl * G + l;
        f[d.f_br_p + b] = br_pf[sl]; f[d.f_br_q + b] = br_qf[sl]; f[d.f_br_s + b] = br_s[sl];
        f[d.f_br_im + b] = dump_signed_abs(br_ifr[sl], dump_abs(br_ifr[sl], br_ifi[sl]));
        f[d.f_br_ia + b] = dump_arg(br_ifi[sl], br_ifr[sl]);
      }
    if (l == 0) {
      f[d.f_bus_p] = slack_p; f[d.f_bus_q] = slack_q; f[d.f_bus_vm] = 1.0; f[d.f_bus_va] = 0.0;
      f[d.f_bus_im] = dump_abs(i0r, i0i); f[d.f_bus_ia] = dump_arg(i0i, i0r);
    }
    if (typ != DEV_NONE) { f[d.f_dev_p + l] = dev_p; f[d.f_dev_q + l] = dev_q; }
    if (typ == DEV_STORAGE) f[d.f_des_soc + slot] = soc;
    if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) f[d.f_gen_pmax + slot] = p_pot;
  };

  // (one call site each for the list-form observation and for the dump: their |z| / arg z code exists once, not once per
  // mode and branch -- the branches only decide: lo_mode 0 none, 1 zeros, 2 gather; dump)
  int lo_mode = 0;
  bool dump = false;
  double* state = io.e.state + e * W_ST;
  // the observation: clip(state, Box) next to the state row, or (a list is set: anm_env.py:497-521, 562-592; see
  // anm_radial.hpp) n_obs entries gathered from this environment's electrical state
  const bool list = mode == 2 && io.e.n_obs > 0;
  const int ON = list ? io.e.n_obs : W_ST;                                  // entries of an observation row
  const int OW = list ? (io.v.w_obs > 0 ? io.v.w_obs : io.e.n_obs) : W_ST;   // its stride (a view pads the rows)
  double* obs = io.e.obs + e * OW;
  cptr_t lo = C + d.off_obs_lo, hi = C + d.off_obs_hi;
  auto put = [&](int k, double v) {
    state[k] = v;
    if (!list) obs[k] = fmin(fmax(v, lo[k]), hi[k]);
  };
  auto list_obs = [&](bool zero) {
    if (!list) return;
    if (zero) {   // terminal / absorbing: the observation is 0 (anm_env.py:365-367, 442-446)
      for (int k = l; k < ON; k += G) obs[k] = 0.0;
      return;
    }
    // the row lives where the Jacobian blocks were (l_blk ... l_bw: free once the solve is over; anm_model_set_obs has
    // checked that FS + KMAX doubles fit there)
    double* row = S + d.l_blk;
    const unsigned need = io.e.obs_need;
    auto want = [&](unsigned c) { return ((need >> c) & 1u) != 0u; };
    ANM_MESH_SYNC();   // (every lane is done with the areas the row overwrites)
    if (isbus) {
      row[d.f_bus_p + bus] = bus_p; row[d.f_bus_q + bus] = bus_q;
      if (want(FC_BUS_VM)) row[d.f_bus_vm + bus] = dump_abs(vr, vi);
      if (want(FC_BUS_VA)) row[d.f_bus_va + bus] = dump_arg(vi, vr);
      if (want(FC_BUS_IM)) row[d.f_bus_im + bus] = dump_abs(ir, ii);
      if (want(FC_BUS_IA)) row[d.f_bus_ia + bus] = dump_arg(ii, ir);
    }
#pragma unroll
    for (int sl = 0; sl < BR_SLOTS; ++sl)
      if (isbr[sl]) {
        const int b = sl * G + l;
        row[d.f_br_p + b] = br_pf[sl]; row[d.f_br_q + b] = br_qf[sl]; row[d.f_br_s + b] = br_s[sl];
        if (want(FC_BR_IM)) row[d.f_br_im + b] = dump_signed_abs(br_ifr[sl], dump_abs(br_ifr[sl], br_ifi[sl]));
        if (want(FC_BR_IA)) row[d.f_br_ia + b] = dump_arg(br_ifi[sl], br_ifr[sl]);
      }
    if (l == 0) {
      row[d.f_bus_p] = slack_p; row[d.f_bus_q] = slack_q; row[d.f_bus_vm] = 1.0; row[d.f_bus_va] = 0.0;
      if (want(FC_BUS_IM)) row[d.f_bus_im] = dump_abs(i0r, i0i);
      if (want(FC_BUS_IA)) row[d.f_bus_ia] = dump_arg(i0i, i0r);
    }
    if (typ != DEV_NONE) { row[d.f_dev_p + l] = dev_p; row[d.f_dev_q + l] = dev_q; }
    if (typ == DEV_STORAGE) row[d.f_des_soc + slot] = soc;
    if (typ == DEV_CLASSICAL || typ == DEV_RENEWABLE) row[d.f_gen_pmax + slot] = p_pot;
    if (resetting || io.e.exo == nullptr) { if (l == 0) row[d.FS] = double(aux); }
    else for (int k = l; k < K; k += G) row[d.FS + k] = io.e.aux_next[e * W_AUX + k];
    ANM_MESH_SYNC();
    for (int k = l; k < ON; k += G) {
      const double v = row[io.e.obs_index[k]] * io.e.obs_scale[k];
      obs[k] = fmin(fmax(v, io.e.obs_lo[k]), io.e.obs_hi[k]);
    }
  };
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
  if (skip) {
    if (mode == 2) {  // absorbing terminal state
      if (list) lo_mode = 1;
      else for (int k = l; k < SD_; k += G) obs[k] = 0.0;
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
      for (int k = l; k < SD_; k += G) { state[k] = 0.0; if (!list) obs[k] = 0.0; }
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
      lo_mode = converged ? 2 : 1;
    }
    dump = true;
    break;
  }
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
    for (int k = l; k < SD_; k += G) { state[k] = 0.0; if (!list) obs[k] = 0.0; }
  }
  lo_mode = term ? 1 : 2;
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
  // (lo_mode and dump are uniform over the lanes of an environment: list_obs synchronises them)
  if (list && lo_mode != 0) list_obs(lo_mode == 1);
  if (dump) write_full();
}
#endif  // __HIPCC__

}  // namespace mesh
}  // namespace anm
