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
// and back-substituted by depth.  Hand-overs between lanes are register-to-register where the tree allows
// it (DPP row shifts, else `ds_bpermute_b32`: nothing is stored anywhere inside the loop) or go through LDS
// slots (the radial kernel's larger trees, see LDSX below).  Everything structural (parent, children, height, depth of each bus, position of its admittances
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
#ifndef ANM_LDSX_FETCH
#define ANM_LDSX_FETCH 3   // child slots a parent lane fetches from LDS before folding them (LDS hand-overs)
#endif
namespace anm {
namespace group {

// llvm CmpInst predicate codes taken by __builtin_amdgcn_{fcmp,uicmp,sicmp}: they return the compare as a
// 64-bit lane mask in scalar registers (no bool -> ballot round trip)
enum : int { FCMP_UNO = 8, FCMP_UGT = 10, ICMP_EQ = 32, ICMP_NE = 33, ICMP_SLT = 40 };

// ---------------------------------------------------------------------------------------------
// Hand-overs between the lanes of a group.  Two register-to-register implementations behind one interface,
// chosen per topology by codegen.dpp_plan (a third one, through LDS slots, lives in newton_groups<.., LDSX>):
//   * DPP row shifts (v_mov_b32_dpp row_shl/row_shr + bank mask): no latency to wait for, when the tree has
//     a lane layout in which every hand-over class is at most two whole-row shifts;
//   * ds_bpermute_b32 (the LDS crossbar, no LDS memory): any tree, ~100 cycles of latency per hand-over.
// Both rely on the PADDING lanes of a group (there is always at least one) holding neutral values in every
// group of the wavefront -- V = 1 + 0j, W = 0, Schur complement 0, step 0 -- so that "the parent of a bus
// attached to the slack" and "a child this bus does not have" are ordinary sources: no selects.
// Every lane of the wavefront must be executing at a hand-over.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double bperm(double x, int src4) {
#ifdef ANM_GROUP_FAKE_XFER  // timing experiment only (wrong results): what the trips cost without the hand-overs
  return x + double(src4) * 1e-300;
#endif
  const int lo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bperm(float x, int src4) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(x)));
}
// One DPP move of x.  FULL: whole-row move with bound_ctrl (lanes shifted in from outside the row read 0).
// FIRST && !FULL: a bank-masked move with nothing to merge into -- the lanes it does not write are left
// UNDEFINED (no register to initialise); the caller only consumes it on the lanes that have a source.
template <int CTRL, int BANK, int FULL, bool FIRST>
__device__ __forceinline__ int dpp_move32(int old, int x) {
  if constexpr (FIRST) {
    (void)old;
    return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, FULL ? 0xF : BANK, FULL != 0);
  } else {
    return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xF, BANK, false);
  }
}
template <int CTRL, int BANK, int FULL, bool FIRST>
__device__ __forceinline__ double dpp_move(double old, double x) {
  const int lo = dpp_move32<CTRL, BANK, FULL, FIRST>(__double2loint(old), __double2loint(x));
  const int hi = dpp_move32<CTRL, BANK, FULL, FIRST>(__double2hiint(old), __double2hiint(x));
  return __hiloint2double(hi, lo);
}
template <int CTRL, int BANK, int FULL, bool FIRST>
__device__ __forceinline__ float dpp_move(float old, float x) {
  return __int_as_float(dpp_move32<CTRL, BANK, FULL, FIRST>(__float_as_int(old), __float_as_int(x)));
}

template <class T>
struct Lanes {
  int psrc4;               // bpermute address of the parent's lane (roots, padding lanes: a padding lane)
  int csrc4[T::T_MAXCH];   // ... of the c-th child's lane (no such child: a padding lane)

  // x as held by the lane of this lane's parent bus (a bus attached to the slack, a padding lane: what a
  // padding lane holds, `neutral`)
  template <class R>
  __device__ __forceinline__ R from_parent(R x, R neutral) const {
    if constexpr (T::T_DPP != 0) {
      R v = neutral;
      if constexpr (T::T_PAR_N > 0) {
        static_assert(T::T_PAR_FULL[0] != 0, "codegen.dpp_plan: the first parent move covers the whole row");
        v = dpp_move<T::T_PAR_CTRL[0], T::T_PAR_BANK[0], 1, true>(v, x);
      }
      if constexpr (T::T_PAR_N > 1) v = dpp_move<T::T_PAR_CTRL[1], T::T_PAR_BANK[1], 0, false>(v, x);
      return v;
    } else {
      return bperm(x, psrc4);
    }
  }
  // does from_child<CH> deliver the neutral 0 on the lanes without such a child?  (else: undefined there)
  template <int CH>
  static constexpr bool child_neutral() {
    return T::T_DPP == 0 || T::T_CH_N[CH] == 0 || T::T_CH_FULL[2 * CH] != 0;
  }
  // x as held by the lane of this lane's CH-th child
  template <int CH, class R>
  __device__ __forceinline__ R from_child(R x) const {
    if constexpr (T::T_DPP != 0) {
      R v = R(0);
      if constexpr (T::T_CH_N[CH] > 0)
        v = dpp_move<T::T_CH_CTRL[2 * CH], T::T_CH_BANK[2 * CH], T::T_CH_FULL[2 * CH], true>(v, x);
      if constexpr (T::T_CH_N[CH] > 1)
        v = dpp_move<T::T_CH_CTRL[2 * CH + 1], T::T_CH_BANK[2 * CH + 1], 0, false>(v, x);
      return v;
    } else {
      return bperm(x, csrc4[CH]);
    }
  }
};

template <class T>
constexpr int first_padding_lane() {
  for (int l = 0; l < T::GRP; ++l)
    if (T::T_LANE_BUS[l] == 0) return l;
  return -1;
}

// Which buses fold their c-th child at elimination level h (the child has height h - 1) -- of the buses of height h:
// 2 = every one, 1 = some, 0 = none; and: does a bus of a GREATER height fold a c-th child there (an "early" fold: the child
// is folded at the level right after its own, whatever the height of its parent).  Bus 0 (the slack) folds nothing.
template <class T>
constexpr int fold_at_own_level(int h, int c) {
  int n = 0, k = 0;
  for (int b = 1; b < T::NB; ++b)
    if (T::T_HEIGHT[b] == h) {
      ++n;
      const int ch = T::T_CH[b * T::T_MAXCH + c];
      if (ch > 0 && T::T_HEIGHT[ch] == h - 1) ++k;
    }
  return k == 0 ? 0 : (k == n ? 2 : 1);
}
template <class T>
constexpr bool fold_early(int h, int c) {
  for (int b = 1; b < T::NB; ++b)
    if (T::T_HEIGHT[b] > h) {
      const int ch = T::T_CH[b * T::T_MAXCH + c];
      if (ch > 0 && T::T_HEIGHT[ch] == h - 1) return true;
    }
  return false;
}
// do the moves of all child classes write the same lanes of a row?  (one loop-carried register then receives them all in turn)
template <class T>
constexpr bool child_moves_write_the_same_lanes() {
  if (T::T_DPP == 0) return false;
  unsigned first = 0u;
  for (int c = 0; c < T::T_MAXCH; ++c) {
    const int ctrl = T::T_CH_CTRL[2 * c], bank = T::T_CH_BANK[2 * c], k = ctrl & 0xF;
    const bool shl = (ctrl & 0x1F0) == 0x100;
    unsigned m = 0u;
    for (int l = 0; l < 16; ++l) {
      const int sl = shl ? l + k : l - k;
      if (((bank >> (l / 4)) & 1) != 0 && sl >= 0 && sl <= 15) m |= 1u << l;
    }
    if (c == 0) first = m;
    else if (m != first) return false;
  }
  return true;
}
// of the buses of height h: 2 = every one hangs off the slack (depth 0), 1 = some, 0 = none
template <class T>
constexpr int roots_at_height(int h) {
  int n = 0, k = 0;
  for (int b = 1; b < T::NB; ++b)
    if (T::T_HEIGHT[b] == h) {
      ++n;
      if (T::T_DEPTH[b] == 0) ++k;
    }
  return k == 0 ? 0 : (k == n ? 2 : 1);
}
// Child sums without a predicate: the move of child class c (ONE bank-masked row shift, written into a register that is
// zero on the lanes the move never writes and keeps them so) delivers to every bus lane it writes either the value of that
// bus's c-th child or that of a padding lane -- whose W product is zero: its admittances are.  Checked lane by lane over a
// 16-lane row; a shift that would cross into the row's other group must read a padding lane there too.
template <class T>
constexpr bool child_moves_land_on_parents_or_zero() {
  if (T::T_DPP == 0 || T::GRP > 16) return false;
  for (int c = 0; c < T::T_MAXCH; ++c) {
    if (T::T_CH_N[c] != 1 || T::T_CH_FULL[2 * c] != 0) return false;
    const int ctrl = T::T_CH_CTRL[2 * c], bank = T::T_CH_BANK[2 * c];
    const int k = ctrl & 0xF, shl = (ctrl & 0x1F0) == 0x100, shr = (ctrl & 0x1F0) == 0x110;
    if (!shl && !shr) return false;
    for (int l = 0; l < 16; ++l) {
      if (((bank >> (l / 4)) & 1) == 0) continue;
      const int sl = shl ? l + k : l - k;
      if (sl < 0 || sl > 15) continue;                       // no source: not written
      const int R = T::T_LANE_BUS[l % T::GRP], S = T::T_LANE_BUS[sl % T::GRP];
      if (R == 0) continue;                                  // a padding lane receives: nobody reads its sums
      if (S == 0) {
        // a padding lane's product arrives: 0 x (the voltage of whatever lane its own "parent" moves read) -- a zero unless
        // that voltage is Inf / NaN, so that lane must belong to the receiver's environment (whose solve is lost then
        // anyway), never to the neighbouring group of the row
        int src = -1;
        for (int m = 0; m < T::T_PAR_N; ++m) {
          const int pc = T::T_PAR_CTRL[m], pk = pc & 0xF;
          const bool pshl = (pc & 0x1F0) == 0x100, pshr = (pc & 0x1F0) == 0x110;
          if (!pshl && !pshr) return false;
          if (T::T_PAR_FULL[m] == 0 && ((T::T_PAR_BANK[m] >> (sl / 4)) & 1) == 0) continue;
          const int q = pshl ? sl + pk : sl - pk;
          if (q < 0 || q > 15) { if (T::T_PAR_FULL[m] != 0) src = -1; continue; }
          src = q;
        }
        if (src >= 0 && src / T::GRP != l / T::GRP) return false;
        continue;
      }
      if (l / T::GRP != sl / T::GRP) return false;           // another environment's bus
      if (T::T_CH[R * T::T_MAXCH + c] != S) return false;
    }
  }
  // ... and every bus's c-th child IS delivered by that move
  for (int b = 1; b < T::NB; ++b)
    for (int c = 0; c < T::T_MAXCH; ++c) {
      const int ch = T::T_CH[b * T::T_MAXCH + c];
      if (ch <= 0) continue;
      const int ctrl = T::T_CH_CTRL[2 * c], bank = T::T_CH_BANK[2 * c], k = ctrl & 0xF;
      const int l = T::T_POS[b], sl = ((ctrl & 0x1F0) == 0x100) ? l + k : l - k;
      if (((bank >> (l / 4)) & 1) == 0 || sl < 0 || sl >= T::GRP || T::T_LANE_BUS[sl] != ch) return false;
    }
  return true;
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

// 2x2 inverse with ONE Newton step on v_rcp_f64 (measured on MI355X: rcp 24.4 bits, one step 48.7 bits = 2.2e-15
// relative, two steps 52.2 bits).  An inexact inverse perturbs a Newton STEP by that much and nothing else -- the
// fixed point is set by F, evaluated in full precision -- so the lane-group path, which only ever serves
// diverging solves and the rare 7- or 8-iteration ones, trades the second step for a shorter dependent chain;
// the thread-per-environment path keeps two steps (its iterates are pinned bit for bit by round-1 tests).
__device__ __forceinline__ Blk<double> blk_inv_fast(const Blk<double>& m) {
  const double det = fm(m.a, m.d, -(m.b * m.c));
  double r = __builtin_amdgcn_rcp(det);
  r = r * fma(-det, r, 2.0);
  return Blk<double>{m.d * r, -m.b * r, -m.c * r, m.a * r};
}
__device__ __forceinline__ Blk<float> blk_inv_fast(const Blk<float>& m) { return blk_inv(m); }

// What a lane knows about its place in the tree (constant tables indexed by the lane: loaded once).
template <class T>
struct LaneView {
  int l, gb, b;            // position in the group, first lane of the group, bus played (0: padding lane)
  bool lane_bus;
  int height, depth, nch;
  Lanes<T> X;
  double ybb_r, ybb_i, ybp_r, ybp_i, ypb_r, ypb_i;   // Y_bb, Y_b,parent, Y_parent,b (padding lanes: 0 -> W = 0)
  unsigned long long gmask; // lanes of my group
  unsigned pk[T::T_LP_NW > 0 ? T::T_LP_NW : 1];   // this lane's row of codegen.lane_pack (group-relative lanes)
  bool dpp_child;           // hybrid plan: the heavy child of this lane's bus sits in the next lane

  __device__ __forceinline__ void init() {
    constexpr int G = T::GRP;
    constexpr int PAD = first_padding_lane<T>();
    static_assert(PAD >= 0, "a lane group needs a padding lane (codegen.tree_tables)");
    const int lane = threadIdx.x & 63;
    l = lane & (G - 1);
    gb = lane - l;
    gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << gb;
    ybb_r = ybb_i = ybp_r = ybp_i = ypb_r = ypb_i = 0.0;
    dpp_child = false;
    if constexpr (T::T_LP_NW > 0) {
      // ONE row of packed words per lane (a single vector load) instead of the chain lane -> bus -> parent -> lane of the
      // parent through three tables: three dependent loads in front of the first Newton trip of every wavefront
      static_for<0, T::T_LP_NW>([&](auto K) { pk[K] = T::T_LP[l * T::T_LP_NW + K]; });
      const unsigned w0 = pk[0];
      b = int(w0 & 0x7Fu);
      lane_bus = b > 0;
      height = int((w0 >> 7) & 0x7Fu) - 1;
      depth = int((w0 >> 14) & 0x7Fu) - 1;
      nch = int((w0 >> 21) & 0xFu);
      X.psrc4 = 4 * (gb + int((w0 >> 25) & 0x3Fu));
      dpp_child = (w0 >> 31) != 0u;
      static_for<0, T::T_MAXCH>([&](auto Cc) { X.csrc4[Cc] = 4 * (gb + int((pk[1 + Cc / 4] >> (8 * (Cc % 4))) & 0xFFu)); });
      return;
    }
    b = T::T_LANE_BUS[l];
    lane_bus = b > 0;
    const int parent = T::T_PARENT[b];               // 0: slack (b is a root of the elimination forest)
    height = lane_bus ? T::T_HEIGHT[b] : -1;
    depth = lane_bus ? T::T_DEPTH[b] : -1;
    nch = lane_bus ? T::T_NCH[b] : 0;
    X.psrc4 = 4 * (gb + ((lane_bus && parent > 0) ? T::T_POS[parent] : PAD));
    static_for<0, T::T_MAXCH>([&](auto Cc) {
      const int c = lane_bus ? T::T_CH[b * T::T_MAXCH + Cc] : -1;
      X.csrc4[Cc] = 4 * (gb + (c > 0 ? T::T_POS[c] : PAD));
    });
  }
  __device__ __forceinline__ void load_y(cptr_t C) {   // admittances of the compiled network
    typedef Layout<T> L;
    if (lane_bus) {
      ybb_r = C[L::Y_RE + T::T_ZBB[b]]; ybb_i = C[L::Y_IM + T::T_ZBB[b]];
      ybp_r = C[L::Y_RE + T::T_ZBP[b]]; ybp_i = C[L::Y_IM + T::T_ZBP[b]];
      ypb_r = C[L::Y_RE + T::T_ZPB[b]]; ypb_i = C[L::Y_IM + T::T_ZPB[b]];
    }
  }
};

// hybrid hand-overs: position of level h's first light-child slot in the packed list (sum of T_NLH below h)
template <class T>
constexpr int hyb_slot_base(int h) {
  int n = 0;
  for (int k = 0; k < h && k <= T::T_MAXH; ++k) n += T::T_NLH[k];
  return n;
}

// The Newton-Raphson loop of every lane group of the wavefront, from the iterate (vm, cs, sn) each bus lane
// holds, `it` iterations already done.  gvalid (uniform per group): the group holds a solve.  Every lane of
// the wavefront must call.  On return: the final iterate, `it`, and the group's verdict in tb / tn
// (tb != 0: ||F||inf > tol or NaN; tn != 0: F has a NaN).
// LDSX (trees without a DPP plan; xl: 640 doubles private to the wavefront, 16-byte aligned): the hand-overs go
// through LDS -- a lane PUBLISHES what its parent (or its children) will need in its own slot, the consumer reads
// the slot of the lane it needs -- instead of ds_bpermute: a pair of doubles is one 16-byte write / read instead of
// four bpermutes, and the 5 child slots x 6 values of a bushy level cost 15 reads instead of 60 bpermutes.
#ifndef ANM_GROUP_MERGED_REGIONS
#define ANM_GROUP_MERGED_REGIONS 1
#endif
#ifndef ANM_HYB_LIGHT_ALL
#define ANM_HYB_LIGHT_ALL 1
#endif
#ifndef ANM_LDSX_BS_ALL
#define ANM_LDSX_BS_ALL 1
#endif
// Padding lanes never publish; their slots hold the neutral values (zeroed here, V = 1).
// MERGED = false: the loop as it stood before the instruction-count cuts of round 6 (one region per fold, predicated child
// sums, no copies): 21 registers fewer -- what the radial kernel with per-group classes needs on a tree with a DPP plan to keep
// three wavefronts per SIMD.
template <class T, class JT, int EARLY_EXIT_TRIPS = 0, bool LDSX = false, int FETCH = ANM_LDSX_FETCH, bool WFREE = true, bool VPOLY = !LDSX,
          bool MERGED = (ANM_GROUP_MERGED_REGIONS != 0), bool REDUCE_ALWAYS = false>
__device__ __forceinline__ void newton_groups(const LaneView<T>& V, bool gvalid, double& vm, double& cs, double& sn,
                                              double bus_p, double bus_q, int& it, unsigned& tb, unsigned& tn,
                                              double tol, int max_iter, double* xl = nullptr) {
  const Lanes<T>& X = V.X;
  static_assert(!LDSX || T::T_DPP == 0, "LDS hand-overs are the alternative to ds_bpermute, not to DPP moves");
  [[maybe_unused]] const int wl = threadIdx.x & 63;
  [[maybe_unused]] const int pl = X.psrc4 >> 2;                       // lane of the parent (or a padding lane)
  // LDS variant, back substitution without regions (ANM_LDSX_BS_ALL): every lane recomputes its step at every level -- from
  // its own level on its parent's step is final, so it recomputes the same value -- and a bus that hangs off the slack reads
  // a step of zero: the W slot of a padding lane (its admittances are zero), 64 slots behind the V slots.
  constexpr bool BS_ALL = LDSX && ANM_LDSX_BS_ALL != 0;
  [[maybe_unused]] const int plb = (BS_ALL && V.depth == 0) ? 64 + pl : pl;
  [[maybe_unused]] int cl[T::T_MAXCH > 0 ? T::T_MAXCH : 1];           // lanes of the children (or a padding lane)
  static_for<0, T::T_MAXCH>([&](auto Cc) { cl[Cc] = X.csrc4[Cc] >> 2; });
  // LDS hand-overs: what a lane publishes sits TOGETHER in its slot -- (re, im) pairs and the six values a parent folds
  // as 16-byte units -- so that every access is a ds_read_b128 / ds_write_b128: four LDS cycles per wavefront
  // instruction where two 8-byte accesses 64 slots apart (ds_read2st64_b64, what separate arrays give) take eight.
  // The 12 wavefronts of a compute unit share ONE LDS pipe, and this loop keeps it busier than any VALU.
  //   xV[lane] = V (re, im), later the Newton step;  xW[lane] = W_pb;  xS[3 lane + 0 .. 2] = (Sc.a, Sc.b), (Sc.c, Sc.d), (Lr0, Lr1)
  [[maybe_unused]] double2* const xV = reinterpret_cast<double2*>(xl);
  [[maybe_unused]] double2* const xW = xV + 64;
  [[maybe_unused]] double2* const xS = xV + 128;
  if constexpr (LDSX) {
    xW[wl] = double2{0.0, 0.0};
    xS[3 * wl] = double2{0.0, 0.0}; xS[3 * wl + 1] = double2{0.0, 0.0}; xS[3 * wl + 2] = double2{0.0, 0.0};
    ANM_WAVE_SYNC();
  }
  // Hybrid hand-overs of the elimination (codegen.hybrid_plan; LDS variant only): the buses lie along their heavy edges, so
  // the child a bus's own elimination waits for -- the one of height h - 1 -- sits in the next lane and its six values
  // arrive by DPP (row_shl:1, no LDS round trip on the longest chain); every other child is folded through its LDS slot at
  // the level right after its own, by a parent that has nothing else to do then.  lq[h][j]: slot index (3 x lane) of the
  // j-th light child this lane's bus folds at level h (no such child: a padding lane's slot, zeros).
  constexpr bool HYB = LDSX && T::T_HYB != 0;
  // (the lanes of the children are per-lane constants of the whole loop: packed four to a register -- a byte each, one
  // v_bfe_u32 to take one out -- or they cost the kernel its third wavefront per SIMD: 170 registers instead of <= 168)
  constexpr int NLQ = HYB ? hyb_slot_base<T>(T::T_MAXH + 1) : 0;
  [[maybe_unused]] unsigned lqp[HYB ? (NLQ + 3) / 4 + 1 : 1] = {0u};
  [[maybe_unused]] unsigned clp[HYB ? (T::T_MAXCH + 3) / 4 : 1] = {0u};
  [[maybe_unused]] const bool dpp_child = V.dpp_child;   // this lane's bus has its heavy child in the next lane
  if constexpr (HYB) {
    static_assert(T::T_LP_NW > 0, "the hybrid plan comes with the packed lane table (codegen.lane_pack)");
    // (the table holds group-relative lanes, a byte each: + the first lane of the group in every byte)
    const unsigned gb1 = unsigned(V.gb) * 0x01010101u;
    static_for<0, (NLQ + 3) / 4>([&](auto K) { lqp[K] = V.pk[T::T_LP_LQ + K] + gb1; });
    static_for<0, (T::T_MAXCH + 3) / 4>([&](auto K) { clp[K] = V.pk[1 + K] + gb1; });
  }
  [[maybe_unused]] auto child_lane = [&](auto Cc) -> int {   // lane of the Cc-th child (W sums)
    if constexpr (HYB) return int(__builtin_amdgcn_ubfe(clp[Cc / 4], 8u * (Cc % 4), 8u));
    else return cl[Cc];
  };
  const int height = V.height, depth = V.depth, nch = V.nch;
  const double ybb_r = V.ybb_r, ybb_i = V.ybb_i, ybp_r = V.ybp_r, ybp_i = V.ybp_i, ypb_r = V.ypb_r, ypb_i = V.ypb_i;
  const unsigned long long gmask = V.gmask;
  const bool isbus = V.lane_bus && gvalid;
  {
    // ---- the reference's loop, rotated like pf_iterate (anm_device.hpp): evaluate; account for the update of
    // the previous trip; leave when no group runs; update.  `running` is uniform over a group; a group that
    // stopped (or holds no solve) rides along: its lanes keep executing, nothing of it changes any more.
    // Predicates live as 64-bit lane masks in scalar registers (v_cmp writes them there; the loop control is
    // then SALU work): `runm` = lanes of the groups that are still iterating.
    const unsigned long long busm = __builtin_amdgcn_uicmp(isbus ? 1u : 0u, 0u, ICMP_NE);
    unsigned long long runm = __builtin_amdgcn_uicmp(gvalid ? 1u : 0u, 0u, ICMP_NE);
    const unsigned glo = unsigned(gmask & busm), ghi = unsigned((gmask & busm) >> 32);   // bus lanes of my group
    tb = 0u; tn = 0u;
    [[maybe_unused]] const unsigned fbus1 = isbus ? 1u : 0u, fbus3 = isbus ? 3u : 0u;
    it -= gvalid ? 1 : 0;        // the first trip only evaluates: its `it += running` is undone here
    // What a lane publishes for its parent (Schur complement, reduced right-hand side) and its Newton step:
    // a bus lane rewrites them at its own level of every trip before anybody reads them; a padding lane never
    // does, so these zeros are the neutral values the hand-overs rely on.
    Blk<JT> Sc = Blk<JT>{JT(0), JT(0), JT(0), JT(0)};
    JT Lr0 = JT(0), Lr1 = JT(0), d0 = JT(0), d1 = JT(0);
    constexpr bool ROOT_STEP_IN_PIVOT = !LDSX && !(LDSX && T::T_HYB != 0) && T::T_LP_NW > 0 && MERGED;
    constexpr bool WSUM_FREE = WFREE && !LDSX && MERGED && T::T_MAXCH > 0 && child_moves_land_on_parents_or_zero<T>();
    constexpr bool WSUM_ONE = WSUM_FREE && child_moves_write_the_same_lanes<T>();
    [[maybe_unused]] double wr[T::T_MAXCH > 0 ? T::T_MAXCH : 1] = {0.0}, wi[T::T_MAXCH > 0 ? T::T_MAXCH : 1] = {0.0};
    [[maybe_unused]] int trip = 0;
    for (;;) {
      const double vr = vm * cs, vi = vm * sn;
      double vpr, vpi;
      if constexpr (LDSX) {
        xV[wl] = double2{vr, vi};
        ANM_WAVE_SYNC();
        const double2 vp = xV[pl];
        vpr = vp.x; vpi = vp.y;
      } else {
        vpr = X.from_parent(vr, 1.0); vpi = X.from_parent(vi, 0.0);
      }
      // W_bb = conj(Y_bb) vm^2;  P = V_b conj(V_p):  W_bp = conj(Y_bp) P,  W_pb = conj(Y_pb) conj(P)
      const double m2 = vm * vm;
      const double wbb_r = ybb_r * m2, wbb_i = -(ybb_i * m2);
      const double pr = fma(vr, vpr, vi * vpi), pim = fma(vi, vpr, -(vr * vpi));
      const double wbp_r = fma(ybp_r, pr, ybp_i * pim), wbp_i = fma(ybp_r, pim, -(ybp_i * pr));
      // (TRIM: the imaginary part is carried with its sign flipped -- the flip rides on the operands of its consumers instead of
      // being an instruction in front of the moves that hand it to the parent)
      constexpr bool TRIM = MERGED;   // (the LDS variant too: what it publishes is the unflipped value)
      const double wpb_r = fma(ypb_r, pr, -(ypb_i * pim)), nwpb_i = fma(ypb_r, pim, ypb_i * pr);
      [[maybe_unused]] const double wpb_i = -nwpb_i;
      // S_b = W_bb + W_bp + sum over the children c of W_pb(c)
      double sr = wbb_r + wbp_r, si = wbb_i + wbp_i;
      if constexpr (LDSX) {
        xW[wl] = TRIM ? double2{wpb_r, nwpb_i} : double2{wpb_r, wpb_i};
        ANM_WAVE_SYNC();
      }
      if constexpr (WSUM_FREE) {
        // (no predicate: see child_moves_land_on_parents_or_zero; wr / wi are loop-carried so that the lanes no move writes
        // keep their zeros)
        static_for<0, T::T_MAXCH>([&](auto Cc) {
          constexpr int K = WSUM_ONE ? 0 : int(Cc);   // (all classes write the same lanes: one register receives them in turn)
          wr[K] = dpp_move<T::T_CH_CTRL[2 * Cc], T::T_CH_BANK[2 * Cc], 0, false>(wr[K], wpb_r);
          wi[K] = dpp_move<T::T_CH_CTRL[2 * Cc], T::T_CH_BANK[2 * Cc], 0, false>(wi[K], nwpb_i);
          sr += wr[K];
          si -= wi[K];
        });
      } else
      static_for<0, T::T_MAXCH>([&](auto Cc) {
        double cr, ci;
        if constexpr (LDSX) {
          const double2 cw = xW[child_lane(Cc)];
          cr = cw.x; ci = cw.y;
        } else {
          cr = X.template from_child<Cc>(wpb_r); ci = X.template from_child<Cc>(TRIM ? nwpb_i : wpb_i);
        }
        if (Lanes<T>::template child_neutral<Cc>() || Cc < nch) {
          sr += cr;
          if constexpr (TRIM) si -= ci; else si += ci;
        }
      });
      const double fr = sr - bus_p, fi = si - bus_q;
      // group-wide stop test: "||F||inf > tol" and "F has a NaN" are all the reference's loop and flags need.
      // A group that has stopped keeps evaluating the same frozen iterate, so what the last trip computed
      // is also each group's final verdict: nothing but `runm` and `it` is carried over.
#ifndef ANM_GROUP_FLAG_OR
#define ANM_GROUP_FLAG_OR 0   // (measured, same box: headline kernel 74.7 us without, 79.2 us with -- the butterfly sits on the exit chain)
#endif
      constexpr bool FLAG_OR = ANM_GROUP_FLAG_OR != 0 && !LDSX && MERGED && T::T_DPP != 0 && (T::GRP == 8 || T::GRP == 16);
      if constexpr (FLAG_OR) {
        // the two verdicts as bits of one word per bus lane (1: above the tolerance or NaN, 2: NaN; a padding lane: 0), OR-ed
        // over the group by a butterfly of DPP-operand v_or_b32 (quad swaps, half-row mirror, row mirror for 16 lanes): every
        // lane then holds its group's verdict, and "keeps iterating" is the one comparison  verdict == 1
        const bool bad = !(fmax(fabs(fr), fabs(fi)) <= tol), nan = !(fr == fr) || !(fi == fi);
        unsigned f = bad ? fbus1 : 0u;
        f = nan ? fbus3 : f;
        f |= unsigned(__builtin_amdgcn_update_dpp(0, int(f), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        f |= unsigned(__builtin_amdgcn_update_dpp(0, int(f), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        f |= unsigned(__builtin_amdgcn_update_dpp(0, int(f), 0x141, 0xF, 0xF, true));   // row_half_mirror
        if constexpr (T::GRP == 16) f |= unsigned(__builtin_amdgcn_update_dpp(0, int(f), 0x140, 0xF, 0xF, true));   // row_mirror
        tb = f & 1u; tn = f & 2u;
        it += __builtin_amdgcn_inverse_ballot_w64(runm) ? 1 : 0;   // the update applied in the previous trip
        runm &= __builtin_amdgcn_uicmp(f, 1u, ICMP_EQ) & __builtin_amdgcn_sicmp(it, max_iter, ICMP_SLT);
      } else {
      const unsigned long long badm = __builtin_amdgcn_fcmp(fmax(fabs(fr), fabs(fi)), tol, FCMP_UGT);
      const unsigned long long nanm = __builtin_amdgcn_fcmp(fr, fi, FCMP_UNO);
      tb = (unsigned(badm) & glo) | (unsigned(badm >> 32) & ghi);
      tn = (unsigned(nanm) & glo) | (unsigned(nanm >> 32) & ghi);
      it += __builtin_amdgcn_inverse_ballot_w64(runm) ? 1 : 0;   // the update applied in the previous trip
      runm &= __builtin_amdgcn_uicmp(tb, 0u, ICMP_NE) & ~__builtin_amdgcn_uicmp(tn, 0u, ICMP_NE) &
              __builtin_amdgcn_sicmp(it, max_iter, ICMP_SLT);   // NaN > tol is false, like the reference
      }
      // The loop is left at the END of the trip: the scalar chain compare -> masks -> branch then overlaps the
      // elimination instead of stalling the wavefront in front of it every trip (what counts for the ~94 trips
      // of a diverging solve); the price is one idle elimination when the last group stops (its update is
      // masked: `runm` is 0).  A caller whose whole batch passes through here (anm_radial.hpp: four or five
      // trips per solve, so an idle elimination is one trip in five) asks for the branch to be taken on the
      // spot during the first EARLY_EXIT_TRIPS trips.
      const bool all_done = runm == 0ull;
      if constexpr (EARLY_EXIT_TRIPS > 0) {
        if (all_done && trip < EARLY_EXIT_TRIPS) break;
        ++trip;
      }

      // ---- Jacobian blocks (anm_device.hpp: newton_update): own diagonal and the two couplings with the parent
      // (TRIM: wbb_i - si for -(si - wbb_i): one subtraction, not a subtraction and a sign flip of a value that is then
      // updated in place; the two differ in the sign of a zero)
      Blk<JT> Dg = Blk<JT>{TRIM ? JT(wbb_i - si) : JT(-(si - wbb_i)), JT(sr + wbb_r), JT(sr - wbb_r), JT(si + wbb_i)};
      const Blk<JT> Jbp = Blk<JT>{JT(wbp_i), JT(wbp_r), JT(-wbp_r), JT(wbp_i)};
      const Blk<JT> Jpb = TRIM ? Blk<JT>{JT(-nwpb_i), JT(wpb_r), JT(-wpb_r), JT(-nwpb_i)}
                               : Blk<JT>{JT(wpb_i), JT(wpb_r), JT(-wpb_r), JT(wpb_i)};
      JT r0 = JT(fr), r1 = JT(fi);
      // ---- elimination by height: a bus folds the Schur complements and reduced right-hand sides its
      // children published (registers Sc / Lr of the child lanes; 0 on padding lanes), inverts its pivot
      // and publishes its own
      if constexpr (HYB) {
        static_for<0, T::T_MAXH + 1>([&](auto H) {
          constexpr int h = H;
          if constexpr (h > 0) {
            constexpr int NL = T::T_NLH[h];
            // light children of height h - 1 (published at the end of the previous level), folded by every bus that has some --
            // whatever its own height: all NL slots fetched together, then folded
            if constexpr (NL > 0) {
              // (ANM_HYB_LIGHT_ALL: every lane reads and folds -- a lane without such a child reads a padding lane's slot, zeros,
              // also after its own pivot: no region around the reads)
              if (ANM_HYB_LIGHT_ALL != 0 || height >= h) {
#ifndef ANM_HYB_FETCH
#define ANM_HYB_FETCH 1   // light-child slots fetched together before they are folded (2: 170 registers, the third wavefront per SIMD lost)
#endif
                static_for<0, (NL + ANM_HYB_FETCH - 1) / ANM_HYB_FETCH>([&](auto R) {
                  constexpr int Q0 = R * ANM_HYB_FETCH;
                  constexpr int NF = (NL - Q0) < ANM_HYB_FETCH ? (NL - Q0) : ANM_HYB_FETCH;
                  double2 qv[NF][3];
                  static_for<0, NF>([&](auto Q) {
                    constexpr int q = hyb_slot_base<T>(h) + Q0 + Q;
                    const int cq = 3 * int(__builtin_amdgcn_ubfe(lqp[q / 4], 8u * (q % 4), 8u));
                    qv[Q][0] = xS[cq]; qv[Q][1] = xS[cq + 1]; qv[Q][2] = xS[cq + 2];
                  });
                  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): ONE wait for the round's reads (else one per 16-byte unit)
                  static_for<0, NF>([&](auto Q) {
                    Dg.a -= JT(qv[Q][0].x); Dg.b -= JT(qv[Q][0].y); Dg.c -= JT(qv[Q][1].x); Dg.d -= JT(qv[Q][1].y);
                    r0 -= JT(qv[Q][2].x); r1 -= JT(qv[Q][2].y);
                  });
                });
              }
            }
            // the heavy child's publication: registers of the next lane (every lane executes the moves; only the buses of
            // this height, whose next lane IS their child of height h - 1, use what arrives)
            constexpr int SHL1 = 0x101;   // row_shl:1 -- lane i reads lane i + 1 of its 16-lane row
            const JT ha = dpp_move<SHL1, 0xF, 1, true>(JT(0), Sc.a), hb = dpp_move<SHL1, 0xF, 1, true>(JT(0), Sc.b);
            const JT hc = dpp_move<SHL1, 0xF, 1, true>(JT(0), Sc.c), hd = dpp_move<SHL1, 0xF, 1, true>(JT(0), Sc.d);
            const JT h0 = dpp_move<SHL1, 0xF, 1, true>(JT(0), Lr0), h1 = dpp_move<SHL1, 0xF, 1, true>(JT(0), Lr1);
            if (height == h && (T::T_ALL_HEAVY != 0 || dpp_child)) {
              Dg.a -= ha; Dg.b -= hb; Dg.c -= hc; Dg.d -= hd;
              r0 -= h0; r1 -= h1;
            }
          }
          if (height == h) {
            Dg = blk_inv_fast(Dg);
            if constexpr (h < T::T_MAXH) {
              const Blk<JT> Lk = blk_mul(Jpb, Dg);
              Sc = blk_mul(Lk, Jbp);
              Lr0 = fm(Lk.a, r0, Lk.b * r1);
              Lr1 = fm(Lk.c, r0, Lk.d * r1);
              xS[3 * wl] = double2{double(Sc.a), double(Sc.b)};
              xS[3 * wl + 1] = double2{double(Sc.c), double(Sc.d)};
              xS[3 * wl + 2] = double2{double(Lr0), double(Lr1)};
            }
          }
          if constexpr (h < T::T_MAXH) ANM_WAVE_SYNC();
        });
      } else if constexpr (!LDSX && T::T_LP_NW > 0 && MERGED) {
        // Register hand-overs, every child folded at the level right after its OWN (see the next variant), with the
        // predicated code of a level gathered: all moves of the level first (every lane executes them), then ONE region
        // for the buses of this height -- their folds (no predicate of its own for a child class every bus of this height
        // folds now: the 6-bus feeder's only kind) and the pivot -- and one per child class a HIGHER bus folds early.  Each
        // region the compiler opens is a save / restore pair of the exec mask (+ a skip branch for a long one): issue
        // slots like any other for a wavefront that iterates alone on its SIMD.  Same operations on every bus, in the
        // same order: bit-identical to the one-region-per-fold form (ANM_GROUP_MERGED_REGIONS=0).
        // Level 0 has nothing to fold, so its pivot needs no region either when it writes to a copy (Di) and no leaf hangs
        // off the slack: every lane computes it, a bus of a greater height (and a padding lane) on an unfinished block --
        // what that publishes is rewritten at the bus's own level before its parent reads it, a padding lane's is read by nobody.
        constexpr bool L0_FREE = roots_at_height<T>(0) == 0;
        Blk<JT> Di = Dg;   // (a padding lane's copy stays as it is: nobody reads it)
        static_for<0, T::T_MAXH + 1>([&](auto H) {
          constexpr int h = H;
          [[maybe_unused]] JT g[T::T_MAXCH][6];
          [[maybe_unused]] auto mine_now = [&](auto Cc) {   // my Cc-th child has height h - 1
            return ((V.pk[T::T_LP_HH + Cc / 4] >> (8 * (Cc % 4))) & 0xFFu) == unsigned(h);
          };
          [[maybe_unused]] auto fold = [&](auto Cc) {
            Dg.a -= g[Cc][0]; Dg.b -= g[Cc][1]; Dg.c -= g[Cc][2]; Dg.d -= g[Cc][3];
            r0 -= g[Cc][4]; r1 -= g[Cc][5];
          };
          if constexpr (h > 0) {
            static_for<0, T::T_MAXCH>([&](auto Cc) {
              if constexpr (T::T_CLS_H[h * T::T_MAXCH + Cc] != 0) {
                g[Cc][0] = X.template from_child<Cc>(Sc.a); g[Cc][1] = X.template from_child<Cc>(Sc.b);
                g[Cc][2] = X.template from_child<Cc>(Sc.c); g[Cc][3] = X.template from_child<Cc>(Sc.d);
                g[Cc][4] = X.template from_child<Cc>(Lr0); g[Cc][5] = X.template from_child<Cc>(Lr1);
              }
            });
          }
          const bool at_level = (h == 0 && L0_FREE) ? true : (height == h);   // (a constant at a free level 0: no region)
          if (at_level) {
            if constexpr (h > 0) {
              static_for<0, T::T_MAXCH>([&](auto Cc) {
                if constexpr (fold_at_own_level<T>(h, Cc) == 2) fold(Cc);
                else if constexpr (fold_at_own_level<T>(h, Cc) == 1) { if (mine_now(Cc)) fold(Cc); }
              });
            }
            Di = blk_inv_fast(Dg);
            // (the Newton step of a bus that hangs off the slack -- the first level of the back substitution -- right here,
            // inside the region of its pivot: the same expression, no region or select of its own)
            if constexpr (roots_at_height<T>(h) == 2) {
              d0 = fm(Di.a, r0, Di.b * r1);
              d1 = fm(Di.c, r0, Di.d * r1);
            } else if constexpr (roots_at_height<T>(h) == 1) {
              if (depth == 0) {
                d0 = fm(Di.a, r0, Di.b * r1);
                d1 = fm(Di.c, r0, Di.d * r1);
              }
            }
            if constexpr (h < T::T_MAXH) {
              const Blk<JT> Lk = blk_mul(Jpb, Di);
              Sc = blk_mul(Lk, Jbp);
              Lr0 = fm(Lk.a, r0, Lk.b * r1);
              Lr1 = fm(Lk.c, r0, Lk.d * r1);
            }
          }
          if constexpr (h > 0) {
            static_for<0, T::T_MAXCH>([&](auto Cc) {
              if constexpr (fold_early<T>(h, Cc)) { if (height > h && mine_now(Cc)) fold(Cc); }
            });
          }
        });
        Dg = Di;   // (the back substitution reads the inverted pivots from Dg)
      } else if constexpr (!LDSX && T::T_LP_NW > 0) {
        // Register hand-overs (DPP moves / ds_bpermute), every child folded at the level right after its OWN (whatever the
        // height of its parent: a parent of a higher level has nothing else to do then) -- a level moves only the child
        // classes some bus really folds there (T_CLS_H): the 6-bus feeder moves 2 + 1 classes instead of 2 + 2, twelve to
        // twenty-four DPP moves of a 375-instruction trip.  (height + 1 of each child of this lane's bus: a byte each in the packed row)
        static_for<0, T::T_MAXH + 1>([&](auto H) {
          constexpr int h = H;
          if constexpr (h > 0) {
            static_for<0, T::T_MAXCH>([&](auto Cc) {
              if constexpr (T::T_CLS_H[h * T::T_MAXCH + Cc] != 0) {
                const JT ga = X.template from_child<Cc>(Sc.a), gbb = X.template from_child<Cc>(Sc.b);
                const JT gc = X.template from_child<Cc>(Sc.c), gd = X.template from_child<Cc>(Sc.d);
                const JT g0 = X.template from_child<Cc>(Lr0), g1 = X.template from_child<Cc>(Lr1);
                if (((V.pk[T::T_LP_HH + Cc / 4] >> (8 * (Cc % 4))) & 0xFFu) == unsigned(h)) {   // my Cc-th child has height h - 1
                  Dg.a -= ga; Dg.b -= gbb; Dg.c -= gc; Dg.d -= gd;
                  r0 -= g0; r1 -= g1;
                }
              }
            });
          }
          if (height == h) {
            Dg = blk_inv_fast(Dg);
            if constexpr (h < T::T_MAXH) {
              const Blk<JT> Lk = blk_mul(Jpb, Dg);
              Sc = blk_mul(Lk, Jbp);
              Lr0 = fm(Lk.a, r0, Lk.b * r1);
              Lr1 = fm(Lk.c, r0, Lk.d * r1);
            }
          }
        });
      } else
      static_for<0, T::T_MAXH + 1>([&](auto H) {
        constexpr int h = H;
        constexpr int NC = T::T_NCH_H[h];
        // register hand-overs: every lane executes the moves, so they are collected first (NC child slots x 6
        // values); LDS hand-overs: only the lanes of this height read, and fold as they read -- no staging arrays
        // (they cost the LDS variant 9 registers too many for a third wavefront per SIMD)
        JT ga[(!LDSX && NC > 0) ? NC : 1], gbb[(!LDSX && NC > 0) ? NC : 1], gc[(!LDSX && NC > 0) ? NC : 1],
            gd[(!LDSX && NC > 0) ? NC : 1], g0[(!LDSX && NC > 0) ? NC : 1], g1[(!LDSX && NC > 0) ? NC : 1];
        if constexpr (!LDSX) {
          static_for<0, NC>([&](auto Cc) {
            ga[Cc] = X.template from_child<Cc>(Sc.a); gbb[Cc] = X.template from_child<Cc>(Sc.b);
            gc[Cc] = X.template from_child<Cc>(Sc.c); gd[Cc] = X.template from_child<Cc>(Sc.d);
            g0[Cc] = X.template from_child<Cc>(Lr0); g1[Cc] = X.template from_child<Cc>(Lr1);
          });
        }
        if (height == h) {
          static_for<0, NC>([&](auto Cc) {
            if constexpr (LDSX) {
              // FETCH (3; 2 where the caller is short of registers) child slots are fetched together, then folded (a padding lane's slots hold zeros):
              // staging all NC slots costs up to 60 registers (and the third wavefront per SIMD), one slot at a time one
              // LDS latency per child
              constexpr int F = FETCH;
              if constexpr (Cc % F == 0) {
                constexpr int NF = (NC - Cc) < F ? (NC - Cc) : F;   // slots of this round
                double2 qv[NF][3];
                static_for<0, NF>([&](auto Q) {
                  const int cq = 3 * cl[Cc + Q];
                  qv[Q][0] = xS[cq]; qv[Q][1] = xS[cq + 1]; qv[Q][2] = xS[cq + 2];
                });
                static_for<0, NF>([&](auto Q) {
                  Dg.a -= JT(qv[Q][0].x); Dg.b -= JT(qv[Q][0].y); Dg.c -= JT(qv[Q][1].x); Dg.d -= JT(qv[Q][1].y);
                  r0 -= JT(qv[Q][2].x); r1 -= JT(qv[Q][2].y);
                });
              }
            } else if (Lanes<T>::template child_neutral<Cc>() || Cc < nch) {
              Dg.a -= ga[Cc]; Dg.b -= gbb[Cc]; Dg.c -= gc[Cc]; Dg.d -= gd[Cc];
              r0 -= g0[Cc]; r1 -= g1[Cc];
            }
          });
          Dg = blk_inv_fast(Dg);
          if constexpr (h < T::T_MAXH) {  // a bus of maximal height hangs off the slack: nobody folds it
            const Blk<JT> Lk = blk_mul(Jpb, Dg);
            Sc = blk_mul(Lk, Jbp);
            Lr0 = fm(Lk.a, r0, Lk.b * r1);
            Lr1 = fm(Lk.c, r0, Lk.d * r1);
            if constexpr (LDSX) {
              xS[3 * wl] = double2{double(Sc.a), double(Sc.b)};
              xS[3 * wl + 1] = double2{double(Sc.c), double(Sc.d)};
              xS[3 * wl + 2] = double2{double(Lr0), double(Lr1)};
            }
          }
        }
        if constexpr (LDSX && h < T::T_MAXH) ANM_WAVE_SYNC();
      });
      // ---- back substitution by depth (Dg now holds the inverted pivots)
      static_for<0, T::T_MAXD + 1>([&](auto Dd) {
        constexpr int dd = Dd;
        JT p0 = JT(0), p1 = JT(0);
        if constexpr (dd > 0) {
          if constexpr (LDSX) {
            const double2 dp = xV[BS_ALL ? plb : pl];
            p0 = JT(dp.x); p1 = JT(dp.y);
          } else {
            p0 = X.from_parent(d0, JT(0));
            p1 = X.from_parent(d1, JT(0));
          }
        }
        if constexpr (!(ROOT_STEP_IN_PIVOT && dd == 0))
        if (BS_ALL || depth == dd) {
          JT a0 = r0, a1 = r1;
          if constexpr (dd > 0) {
            a0 = fm(-Jbp.b, p1, fm(-Jbp.a, p0, a0));
            a1 = fm(-Jbp.d, p1, fm(-Jbp.c, p0, a1));
          }
          d0 = fm(Dg.a, a0, Dg.b * a1);
          d1 = fm(Dg.c, a0, Dg.d * a1);
          if constexpr (LDSX && dd < T::T_MAXD) xV[wl] = double2{double(d0), double(d1)};
        }
        if constexpr (LDSX && dd < T::T_MAXD) ANM_WAVE_SYNC();
      });
      // ---- update of the running groups; d1 is the relative magnitude step.  Rotation of (cos, sin) by the
      // angle step, path chosen per wavefront exactly as update_angles does (the reduction of a small step
      // returns the same bits as the short path)
      // (a padding lane's step is zero -- it never writes d0 / d1 -- and a rotation by zero returns its V = 1 bit for bit:
      // where the pads ride along, `& busm` is an instruction saved)
      const unsigned long long updm = (TRIM && !LDSX) ? runm : (runm & busm);
      const bool upd = __builtin_amdgcn_inverse_ballot_w64(updm);
      const double dth = double(d0);
      const unsigned long long bigm = __builtin_amdgcn_fcmp(fabs(dth), 0.78, FCMP_UGT) & updm;  // NaN counts
#ifndef ANM_GROUP_POLY_UNMASKED
#define ANM_GROUP_POLY_UNMASKED 1
#endif
      if constexpr (ANM_GROUP_POLY_UNMASKED != 0 && VPOLY) {
        // (VPOLY: the register hand-over variant = the continuation of the thread family's diverging solves, where a trip's
        // instruction COUNT is its duration, and the LDS variant where the caller's register budget has room for the
        // coefficients in vector registers (the radial kernel without per-group classes: 163 of 168): the short path evaluates the polynomials on every lane -- a lane that does not update
        // computes on whatever its dth holds and drops the result -- so that the wave-uniform branch is the only control
        // flow in front of them, and the rotation stands once behind both paths)
        double sd_, cd_;
        // (REDUCE_ALWAYS: no short path -- the range reduction of a small step returns the step itself and quadrant 0, the
        // same bits; for a launch that lasts as long as its slowest wavefront, whose trips reduce anyway, the test and the
        // branch in front of the short path are four instructions of every trip)
        if (!REDUCE_ALWAYS && bigm == 0ull) {
          sincos_kernel<true>(dth, 0, sd_, cd_);
        } else {
          sincos_medium<true>(dth, sd_, cd_);
          // (beyond the two-stage reduction: a handful of lanes per million solves -- the library routine then runs on every
          // lane of the wavefront and the lanes it was called for take its result: no divergent region in this tail)
          const bool ish = !(fabs(dth) < SINCOS_MEDIUM_MAX);
          if ((__builtin_amdgcn_uicmp(ish ? 1u : 0u, 0u, ICMP_NE) & updm) != 0ull) {
            const SinCos r = sincos_huge(dth);
            sd_ = ish ? r.s : sd_;
            cd_ = ish ? r.c : cd_;
          }
        }
        if (upd) {
          vm = fma(-double(d1), fabs(vm), vm);
          // (three-address v_fma_f64 in place of the compiler's v_fmac into a temporary + v_mov_b64 back into the loop's register)
          const double t1 = sn * sd_, t2 = cs * sd_;
          cs = fma3<false>(cs, cd_, t1);
          sn = fma3<true>(sn, cd_, t2);
        }
      } else
      if (bigm == 0ull) {
        if (upd) {
          double sd_, cd_;
          // (one-instruction Horner steps with the coefficients in vector registers -- see horner() -- where there is room: the
          // LDS-hand-over variant, which serves the radial kernel's larger trees, sits at its 168-register budget for three
          // wavefronts per SIMD and keeps the compiler's own form)
          sincos_kernel<!LDSX>(dth, 0, sd_, cd_);
          vm = fma(-double(d1), fabs(vm), vm);
          const double c0 = cs, s0 = sn;
          cs = fma(c0, cd_, s0 * sd_);
          sn = fma(s0, cd_, -(c0 * sd_));
        }
      } else if (upd) {
        double sd_, cd_;
        if (fabs(dth) < SINCOS_MEDIUM_MAX) {
          sincos_medium<!LDSX>(dth, sd_, cd_);
        } else {
          const SinCos r = sincos_huge(dth);
          sd_ = r.s;
          cd_ = r.c;
        }
        vm = fma(-double(d1), fabs(vm), vm);
        const double c0 = cs, s0 = sn;
        cs = fma(c0, cd_, s0 * sd_);
        sn = fma(s0, cd_, -(c0 * sd_));
      }
      if (all_done) break;
    }

  }
}

// Continue the Newton-Raphson solves of the lanes with st.active (they have done st.it iterations and
// hold their iterate in w.vm / st.cs / st.sn) until each converges, fails (NaN) or reaches max_iter --
// NG solves at a time, one per lane group.  On return the owner lanes hold the final iterate, st.it and
// a st.diff that reproduces the reference's flags (NaN: failed; 0: converged; +inf: cap reached).
// `lds`: >= NG * Slot<T>::SIZE doubles, private to this wavefront; every lane of the wave must call.
// WFREE = false: the predicate-free child sums (four to eight loop-carried registers) left out -- the straggler launch, whose
// budget is two wavefronts per SIMD.
template <class T, class JT, bool WFREE = true, bool REDUCE_ALWAYS = false>
__device__ __forceinline__ void continue_in_groups(cptr_t C, EnvWork<T>& w, PFState<T>& st, bool mine, double tol, int max_iter,
                                   double* lds) {
  typedef Slot<T> S;
  constexpr int G = Shape<T>::G, NG = Shape<T>::NG, NB = T::NB;
  const int lane = threadIdx.x & 63;
  const int grp = lane / G;
  LaneView<T> V;
  V.init();
  V.load_y(C);
  const int b = V.b;

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
    const bool isbus = V.lane_bus && gvalid;
    double vm = 1.0, cs = 1.0, sn = 0.0, bus_p = 0.0, bus_q = 0.0;
    int it = 0;
    {
      const double* s = lds + grp * S::SIZE;
      if (isbus) { vm = s[S::VM + b]; cs = s[S::CS + b]; sn = s[S::SN + b]; bus_p = s[S::P + b]; bus_q = s[S::Q + b]; }
      if (gvalid) it = int(s[S::IT]);
    }
    ANM_WAVE_SYNC();
    unsigned tb, tn;
    newton_groups<T, JT, 0, false, ANM_LDSX_FETCH, WFREE, true, (ANM_GROUP_MERGED_REGIONS != 0), REDUCE_ALWAYS>(V, gvalid, vm, cs, sn, bus_p, bus_q, it, tb, tn, tol, max_iter);

    // ---- results back to the owner lanes
    if (isbus) {
      double* s = lds + grp * S::SIZE;
      s[S::VM + b] = vm; s[S::CS + b] = cs; s[S::SN + b] = sn;
      if (b == 1) {
        s[S::IT] = double(it);
        s[S::FLAGS] = (tn != 0u) ? NAN : ((tb != 0u) ? INFINITY : 0.0);
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
