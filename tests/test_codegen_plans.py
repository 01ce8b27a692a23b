"""CPU tier: the lane plans codegen.py hands to the lane-group Newton loop (csrc/anm_group.hpp) are checked as DATA, for random
trees of every group size: the hybrid plan of the tree kernel (heavy child in the next lane of the same DPP row, every other
child folded through LDS at the level right after its own), the packed per-lane rows the kernels read, and the child classes a
level of the register hand-over path moves.  What the kernels do with them is the GPU tier's business (golden vectors,
iterates); a wrong table would show there as wrong physics -- here it shows as a broken invariant, with the tree that breaks it."""
import re

import numpy as np
import pytest

from gym_anm_amd import codegen


def random_tree(n_bus, rng, shape):
    """branches of a tree over buses 0 .. n_bus - 1, bus 0 the slack with one feeder (like the synthetic feeders) or several"""
    br = [(0, 1)]
    for i in range(2, n_bus):
        if shape == "chain":
            p = i - 1
        elif shape == "star":
            p = 1
        elif shape == "feeders" and rng.uniform() < 0.2:
            p = 0
        else:
            p = int(rng.integers(1, i))
        br.append((p, i))
    return br


def tables(n_bus, branches):
    pat = set()
    for f, t in branches:
        pat.update({(f, t), (t, f), (f, f), (t, t)})
    for i in range(n_bus):
        pat.add((i, i))
    return codegen.tree_tables(n_bus, branches, sorted(pat))


CASES = [(n, shape, seed) for n in (3, 6, 9, 13, 17, 30, 33, 48, 64) for shape in ("random", "chain", "star", "feeders") for seed in (0, 1)]


@pytest.mark.parametrize("n_bus,shape,seed", CASES)
def test_hybrid_plan_invariants(n_bus, shape, seed):
    rng = np.random.default_rng(1000 * n_bus + seed)
    branches = random_tree(n_bus, rng, shape)
    tt = tables(n_bus, branches)
    assert tt is not None
    hy = codegen.hybrid_plan(n_bus, tt)
    assert hy is not None
    G, maxch, maxh = tt["GRP"], tt["MAXCH"], tt["MAXH"]
    height, parent = tt["HEIGHT"], tt["PARENT"]
    ch = [[c for c in tt["CH"][b * maxch:(b + 1) * maxch] if c > 0] for b in range(n_bus)]
    pos, lane_bus, heavy = hy["POS"], hy["LANE_BUS"], hy["HEAVY"]
    # every bus in exactly one lane, at least one padding lane, POS and LANE_BUS inverse to each other
    assert sorted(b for b in lane_bus if b) == list(range(1, n_bus)) and lane_bus.count(0) >= 1 and len(lane_bus) == G
    assert all(lane_bus[pos[b]] == b for b in range(1, n_bus))
    for b in range(1, n_bus):
        h = heavy[b]
        if h:
            # the DPP child: a child of height exactly h - 1, in the NEXT lane, inside the same 16-lane row
            assert parent[h] == b and height[h] == height[b] - 1
            assert pos[h] == pos[b] + 1 and pos[h] // 16 == pos[b] // 16
        elif height[b] >= 1:
            assert hy["ALL_HEAVY"] == 0    # (a path cut at a row boundary)
    assert hy["ALL_HEAVY"] == int(all(heavy[b] for b in range(1, n_bus) if height[b] >= 1))
    # every child is folded exactly once: the heavy one by DPP at its parent's level, the others at the level after their own
    folded = {b: [] for b in range(1, n_bus)}
    maxl = hy["MAXL"]
    for h in range(maxh + 1):
        for b in range(n_bus):
            row = hy["LCH"][(h * n_bus + b) * maxl:(h * n_bus + b + 1) * maxl]
            kids = [c for c in row if c > 0]
            assert len(kids) <= hy["NLH"][h]
            for c in kids:
                assert parent[c] == b and height[c] == h - 1 and c != heavy[b] and h <= height[b]
                folded[b].append(c)
    for b in range(1, n_bus):
        assert sorted(folded[b] + ([heavy[b]] if heavy[b] else [])) == sorted(ch[b]), (b, folded[b], heavy[b], ch[b])
    assert hy["NLH"][0] == 0 and max(hy["NLH"]) <= maxl


@pytest.mark.parametrize("n_bus,shape,seed", CASES)
def test_lane_pack_decodes_to_the_tree(n_bus, shape, seed):
    rng = np.random.default_rng(1000 * n_bus + seed)
    branches = random_tree(n_bus, rng, shape)
    tt = tables(n_bus, branches)
    hy = codegen.hybrid_plan(n_bus, tt)
    lp = codegen.lane_pack(n_bus, tt, hy["LANE_BUS"], hy["POS"], hy)
    maxch = tt["MAXCH"]
    if maxch > 15:
        assert lp is None
        return
    nw, o_lq, o_hh, words = lp
    G, pos, lane_bus = tt["GRP"], hy["POS"], hy["LANE_BUS"]
    pad = lane_bus.index(0)
    slots = [(h, j) for h in range(tt["MAXH"] + 1) for j in range(hy["NLH"][h])]
    assert len(words) == G * nw and o_lq == 1 + (maxch + 3) // 4 and o_hh == o_lq + (len(slots) + 3) // 4
    for l in range(G):
        w = words[l * nw:(l + 1) * nw]
        b = w[0] & 0x7F
        assert b == lane_bus[l] and w[0] < 2 ** 32
        if not b:
            assert ((w[0] >> 7) & 0x7F) == 0 and ((w[0] >> 14) & 0x7F) == 0 and ((w[0] >> 21) & 0xF) == 0
            continue
        assert ((w[0] >> 7) & 0x7F) - 1 == tt["HEIGHT"][b] and ((w[0] >> 14) & 0x7F) - 1 == tt["DEPTH"][b]
        assert ((w[0] >> 21) & 0xF) == tt["NCH"][b]
        par = tt["PARENT"][b]
        assert ((w[0] >> 25) & 0x3F) == (pos[par] if par > 0 else pad)
        assert (w[0] >> 31) == (1 if hy["HEAVY"][b] else 0)
        for c in range(maxch):
            k = tt["CH"][b * maxch + c]
            assert ((w[1 + c // 4] >> (8 * (c % 4))) & 0xFF) == (pos[k] if k > 0 else pad)
            assert ((w[o_hh + c // 4] >> (8 * (c % 4))) & 0xFF) == (tt["HEIGHT"][k] + 1 if k > 0 else 0)
        for q, (h, j) in enumerate(slots):
            k = hy["LCH"][(h * n_bus + b) * hy["MAXL"] + j]
            assert ((w[o_lq + q // 4] >> (8 * (q % 4))) & 0xFF) == (pos[k] if k > 0 else pad)
        assert w[nw - 1] == (pos[l + 1] if l + 1 < n_bus else l)


def test_stock_headers_carry_the_plans():
    """ANM6: the all-DPP plan, two child classes at level 1 and ONE at level 2 (bus 1 folds its leaf child 3 a level early);
    the 30-bus feeder of BASELINE config 4: the hybrid plan, 2 + 1 + 1 + 2 light slots instead of 3 + 4 + 3 + 5 child slots"""
    t = codegen.stock_topologies()
    h6, h30 = codegen.emit_header(t["anm6"]), codegen.emit_header(t["case30"])
    get = lambda h, name: [int(x, 0) for x in re.search(r"%s\[\d+\] = \{([^}]*)\}" % name, h).group(1).replace("u", "").split(",")]
    assert "T_DPP = 1" in h6 and "T_HYB = 0" in h6 and get(h6, "T_CLS_H") == [0, 0, 1, 1, 1, 0]
    assert "T_DPP = 0" in h30 and "T_HYB = 1" in h30 and "T_ALL_HEAVY = 1" in h30
    assert get(h30, "T_NLH") == [0, 2, 1, 1, 2] and get(h30, "T_NCH_H") == [0, 3, 4, 3, 5]


def _ints(hdr, name):
    return [int(x) for x in re.search(r"%s\[\d+\] = \{([^}]*)\}" % name, hdr).group(1).split(",")]


def child_moves_land_on_parents_or_zero(hdr):
    """csrc/anm_group.hpp's compile-time check of the same name, restated on a descriptor's text: the (single, bank-masked)
    DPP move of every child class delivers to each bus lane it writes that bus's child of the class, or the product of a
    padding lane whose own parent moves read a lane of the SAME group (or nothing); and every child is delivered."""
    grp, maxch = int(re.search(r"GRP = (\d+)", hdr).group(1)), int(re.search(r"T_MAXCH = (\d+)", hdr).group(1))
    if "T_DPP = 1" not in hdr or grp > 16:
        return False
    lane_bus, pos, ch = _ints(hdr, "T_LANE_BUS"), _ints(hdr, "T_POS"), _ints(hdr, "T_CH")
    ch_n, ch_ctrl, ch_bank, ch_full = (_ints(hdr, "T_CH_" + k) for k in ("N", "CTRL", "BANK", "FULL"))
    par_n = int(re.search(r"T_PAR_N = (\d+)", hdr).group(1))
    par_ctrl, par_bank, par_full = (_ints(hdr, "T_PAR_" + k) for k in ("CTRL", "BANK", "FULL"))
    n_bus = len(pos)

    def src_of(ctrl, lane):
        k, kind = ctrl & 0xF, ctrl & 0x1F0
        assert kind in (0x100, 0x110)
        s = lane + k if kind == 0x100 else lane - k
        return s if 0 <= s <= 15 else None

    for c in range(maxch):
        if ch_n[c] != 1 or ch_full[2 * c] != 0:
            return False
        for lane in range(16):
            if not (ch_bank[2 * c] >> (lane // 4)) & 1:
                continue
            sl = src_of(ch_ctrl[2 * c], lane)
            if sl is None:
                continue
            R, S = lane_bus[lane % grp], lane_bus[sl % grp]
            if R == 0:
                continue
            if S == 0:
                src = None
                for m in range(par_n):
                    if not par_full[m] and not (par_bank[m] >> (sl // 4)) & 1:
                        continue
                    q = src_of(par_ctrl[m], sl)
                    if q is None:
                        if par_full[m]:
                            src = None
                        continue
                    src = q
                if src is not None and src // grp != lane // grp:
                    return False
                continue
            if lane // grp != sl // grp or ch[R * maxch + c] != S:
                return False
    for b in range(1, n_bus):
        for c in range(maxch):
            k = ch[b * maxch + c]
            if k <= 0:
                continue
            sl = src_of(ch_ctrl[2 * c], pos[b])
            if not (ch_bank[2 * c] >> (pos[b] // 4)) & 1 or sl is None or sl >= grp or lane_bus[sl] != k:
                return False
    return True


def test_the_stock_feeder_gets_a_layout_with_predicate_free_child_sums():
    """codegen.dpp_plan's tie-break (round 6): among the equal-cost DPP layouts of the 6-bus feeder one whose child moves land on
    parents or on same-environment zeros -- the lane-group loop then sums the children's W products without a predicate
    (what the headline's trip count rests on)."""
    from gym_anm_amd import networks
    from gym_anm_amd.model import NetworkModel

    assert child_moves_land_on_parents_or_zero(codegen.emit_header(codegen.stock_topologies()["anm6"]))
    for seed in (0, 2):   # (the 5-bus test trees of the GPU tier have whole-row first moves: zeros arrive by bound_ctrl, another path)
        topo = NetworkModel(networks.synthetic_radial_network(5, seed), 0.25, 100).topology()
        hdr = codegen.emit_header(topo)
        assert "T_DPP = 1" in hdr and not child_moves_land_on_parents_or_zero(hdr) and 1 in _ints(hdr, "T_CH_FULL")
    # and the check does say no: the feeder's layout of the first half of the round (a padding lane there read the first bus of
    # the NEXT group as its parent, so its zero product could have been 0 x that environment's Inf)
    hdr = codegen.emit_header(codegen.stock_topologies()["anm6"])
    old = hdr.replace(_line(hdr, "T_LANE_BUS"), "  static constexpr int T_LANE_BUS[8] = {5, 3, 0, 0, 4, 2, 1, 0};")
    old = old.replace(_line(old, "T_POS"), "  static constexpr int T_POS[6] = {0, 6, 5, 1, 4, 0};")
    old = old.replace(_line(old, "T_PAR_CTRL"), "  static constexpr int T_PAR_CTRL[2] = {257, 261};")
    old = old.replace(_line(old, "T_CH_CTRL"), "  static constexpr int T_CH_CTRL[4] = {273, 0, 277, 0};")
    assert not child_moves_land_on_parents_or_zero(old)


def _line(hdr, name):
    return next(l for l in hdr.split("\n") if (" " + name + "[") in l)
