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
