"""Topology compiler: network structure -> constexpr descriptor -> one gfx950 library per topology.

The per-environment kernels (csrc/anm_device.hpp) are templates over a ``Topo`` descriptor so that
every loop over buses / branches / devices / admittance non-zeros / Jacobian blocks is unrolled
with constant indices and the whole per-environment working set stays in VGPRs.  This module

1. does the *symbolic* part of the sparse block-LU used inside Newton-Raphson once per topology
   (minimum-degree elimination order over the non-slack buses, fill-in, the flat op lists the
   kernel replays every iteration),
2. writes the descriptor header, and
3. drives ``hipcc --offload-arch=gfx950`` to build ``libanm_<name>.so`` in-tree
   (``gym_anm_amd/_build/``), ahead of time for the stock topologies (``build_stock``) or on
   first use for a new one.

Nothing here has a counterpart in the reference (which rebuilds a SciPy sparse Jacobian and calls
SuperLU on every Newton iteration, ``solve_load_flow.py:123-164,220``); numerically the kernel
solves the same linear system.
"""

from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

from . import errors as E

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
BUILD_DIR = os.path.join(PKG_DIR, "_build")

LOAD, SLACK, CLASSICAL, RENEWABLE, STORAGE = -1, 0, 1, 2, 3


def topology_signature(topo) -> str:
    n_bus, branches, devices = topo
    return "N%d|B%s|D%s" % (
        n_bus,
        ",".join("%d-%d" % ft for ft in branches),
        ",".join("%d@%d" % tb for tb in devices),
    )


def topology_name(topo) -> str:
    h = hashlib.sha1(topology_signature(topo).encode()).hexdigest()[:10]
    return "n%db%dd%d_%s" % (topo[0], len(topo[1]), len(topo[2]), h)


def symbolic_block_lu(n_bus, branches):
    """Symbolic factorisation of the bus-block Jacobian over buses 1..n_bus-1.

    Returns a dict of flat lists (see emit_header).  Bus indices are the real ones (1-based among
    unknowns because bus 0 is the slack); block ids index the compact list of structurally
    non-zero 2x2 blocks after fill-in.
    """
    unknown = list(range(1, n_bus))
    adj = {u: set() for u in unknown}
    for f, t in branches:
        if f != 0 and t != 0 and f != t:
            adj[f].add(t)
            adj[t].add(f)
    # Y-pattern blocks first (row-major), fill appended later
    blocks = {}
    for i in unknown:
        for k in sorted(adj[i] | {i}):
            blocks[(i, k)] = len(blocks)
    fill = []
    work = {u: set(v) for u, v in adj.items()}
    remaining = set(unknown)
    piv, l_i, l_blk, lseg = [], [], [], [0]
    u_dst, u_src, useg = [], [], [0]
    upper = {}
    while remaining:
        # minimum degree, ties -> larger bus index first (leaves of a feeder before its trunk)
        k = min(remaining, key=lambda u: (len(work[u] & remaining), -u))
        remaining.discard(k)
        nb = sorted(work[k] & remaining)
        piv.append(k)
        upper[k] = nb
        for i in nb:
            for j in nb:
                if (i, j) not in blocks:
                    blocks[(i, j)] = len(blocks)
                    fill.append(blocks[(i, j)])
                if i != j:
                    work[i].add(j)
        for i in nb:
            l_i.append(i)
            l_blk.append(blocks[(i, k)])
            for j in nb:
                u_dst.append(blocks[(i, j)])
                u_src.append(blocks[(k, j)])
            useg.append(len(u_dst))
        lseg.append(len(l_i))
    b_j, b_blk, bseg = [], [], [0]
    for k in piv:
        for j in upper[k]:
            b_j.append(j)
            b_blk.append(blocks[(k, j)])
        bseg.append(len(b_j))
    return dict(blocks=blocks, fill=fill, piv=piv, l_i=l_i, l_blk=l_blk, lseg=lseg, u_dst=u_dst, u_src=u_src,
                useg=useg, b_j=b_j, b_blk=b_blk, bseg=bseg)  # fmt: skip


def tree_tables(n_bus, branches, ypat):
    """Rooted-tree view of a radial network for the lane-group Newton continuation (csrc/anm_group.hpp):
    lane l of a group plays bus l + 1.  Returns None when the network is not a tree (or too large for one
    wavefront).  parent 0 = the slack bus; height 0 = leaf (elimination goes by height: every bus of height h
    folds the Schur complements of its children, which are all lower); depth 0 = attached to the slack
    (back substitution goes by depth)."""
    if len(branches) != n_bus - 1 or n_bus < 2 or n_bus - 1 > 64:
        return None
    adj = {i: [] for i in range(n_bus)}
    for f, t in branches:
        adj[f].append(t)
        adj[t].append(f)
    parent, depth, order = {0: -1}, {0: -1}, [0]
    for u in order:
        for v in sorted(adj[u]):
            if v not in parent:
                parent[v] = u
                depth[v] = depth[u] + 1
                order.append(v)
    if len(order) != n_bus:
        return None
    children = {b: sorted(c for c in range(1, n_bus) if parent[c] == b) for b in range(n_bus)}
    height = {}
    for b in reversed(order):
        height[b] = 0 if (b == 0 or not children[b]) else 1 + max(height[c] for c in children[b])
    if n_bus > 1:
        height[0] = 1 + max(height[c] for c in children[0])
    maxch = max([len(children[b]) for b in range(1, n_bus)] + [1])
    maxh = max(height[b] for b in range(1, n_bus))
    maxd = max(depth[b] for b in range(1, n_bus))
    ch = []
    for b in range(n_bus):
        cs = children[b] if b != 0 else []
        ch += cs + [-1] * (maxch - len(cs))
    nch_h = [max([len(children[b]) for b in range(1, n_bus) if height[b] == h] + [0]) for h in range(maxh + 1)]
    pos = {ij: z for z, ij in enumerate(ypat)}
    grp = 8
    while grp < n_bus:  # n_bus - 1 bus lanes + at least one padding lane (the neutral source of the hand-overs)
        grp *= 2
    if grp > 64:
        return None
    return dict(
        GRP=grp, MAXCH=maxch, MAXH=maxh, MAXD=maxd,
        PARENT=[-1] + [parent[b] for b in range(1, n_bus)],
        HEIGHT=[height[b] for b in range(n_bus)], DEPTH=[-1] + [depth[b] for b in range(1, n_bus)],
        NCH=[0] + [len(children[b]) for b in range(1, n_bus)], CH=ch, NCH_H=nch_h,
        ZBB=[-1] + [pos[(b, b)] for b in range(1, n_bus)],
        ZBP=[-1] + [pos[(b, parent[b])] for b in range(1, n_bus)],
        ZPB=[-1] + [pos[(parent[b], b)] for b in range(1, n_bus)],
    )  # fmt: skip


def dpp_plan(n_bus, tt):
    """Lane layout + DPP steps for the hand-overs of the lane-group Newton trip, when the tree allows it.

    A hand-over ("every lane gets the value its parent holds", "... its c-th child holds") is a general
    permutation, served by ds_bpermute_b32 (one LDS-pipe instruction per dword plus the s_waitcnt that
    follows).  A DPP move (v_mov_b32_dpp row_shl/row_shr) is one plain VALU instruction -- but it only shifts
    whole 16-lane rows by a constant, for whole banks of 4 lanes.  This search looks for positions of the
    buses inside the group such that every hand-over class is at most two such moves: all the
    (destination <- source) pairs of a move share one offset; every other lane a move writes either does not
    care (padding lane), is overwritten by the next move, or reads a NEUTRAL value.  Padding lanes hold
    V = 1 + 0j, W = 0, Schur complement 0 in every group of the wavefront, which is exactly "the parent of a
    bus attached to the slack" and "no such child".  The first move of a class is preferably a full-row move
    with bound_ctrl (lanes shifted in from outside the row read 0, no `old` operand to initialise); a second
    move is restricted to banks.  Groups of 8 lanes only (two groups per DPP row).  Returns None when no
    layout works (the hand-overs then use ds_bpermute_b32)."""
    import itertools

    parent, maxch = tt["PARENT"], tt["MAXCH"]
    ch = [tt["CH"][b * maxch:(b + 1) * maxch] for b in range(n_bus)]
    buses = list(range(1, n_bus))
    roots = [b for b in buses if parent[b] == 0]
    if maxch > 2 or len(roots) > 2 or n_bus - 1 > 15:
        return None
    # the smallest group that has a plan: more solves per wavefront matters more than a cheaper move or two (a
    # wavefront with more stragglers than groups runs its Newton chain twice)
    for G in (8, 16):
        if n_bus > G:
            continue
        r = _dpp_plan_for_group(G, n_bus, parent, ch, maxch, buses, roots, tt)
        if r is not None:
            return r[1]
    return None


def _dpp_plan_for_group(G, n_bus, parent, ch, maxch, buses, roots, tt):
    import itertools

    ngr = 16 // G

    def simulate(steps, lane_bus, pairs, accept_zero):
        """steps: [(offset, bank set within the group or None = full row)].  True when every lane of the row
        ends with what it needs.  accept_zero: a lane without a source may read 0 / keep `old`; otherwise
        (parent class: V of the slack = 1) it must keep `old` or read a padding lane."""
        out = ["OLD"] * 16
        for off, banks in steps:
            for i in range(16):
                g, l = divmod(i, G)
                if banks is not None and (l // 4) not in banks:
                    continue
                src = i + off
                if 0 <= src < 16:
                    out[i] = divmod(src, G)
                elif banks is None:
                    out[i] = "ZERO"  # bound_ctrl
        for i in range(16):
            g, l = divmod(i, G)
            v = out[i]
            if lane_bus[l] == 0:
                continue  # padding lane: don't care
            if l in pairs:
                if v != (g, pairs[l]):
                    return False
            elif v == "OLD":
                continue
            elif v == "ZERO":
                if not accept_zero:
                    return False
            elif lane_bus[v[1]] != 0:
                return False  # reads some bus lane's value
        return True

    def solve_class(lane_bus, pairs, accept_zero):
        by_off = {}
        for d, s_ in pairs.items():
            by_off.setdefault(s_ - d, set()).add(d // 4)
        offs = sorted(by_off)
        if not offs:
            return []
        if len(offs) > 2:
            return None
        cands = []
        if len(offs) == 1:
            cands = [[(offs[0], None)], [(offs[0], by_off[offs[0]])]]
        else:
            a, b = offs
            cands = [[(a, None), (b, by_off[b])], [(b, None), (a, by_off[a])], [(a, by_off[a]), (b, by_off[b])]]
        best = None
        for st in cands:
            if simulate(st, lane_bus, pairs, accept_zero):
                # a bank-masked first move leaves the other lanes undefined: its consumers then run under an
                # exec mask ("has such a child"), a few scalar instructions per use
                cost = len(st) + (0 if st[0][1] is None else 0.25)
                if best is None or cost < best[0]:
                    best = (cost, st)
        return None if best is None else best

    def final_sources(steps):
        out = ["OLD"] * 16
        for off, banks in steps:
            for i in range(16):
                if banks is not None and ((i % G) // 4) not in banks:
                    continue
                src = i + off
                if 0 <= src < 16:
                    out[i] = src
                elif banks is None:
                    out[i] = "ZERO"
        return out

    def child_moves_safe(lane_bus, plan):
        par = final_sources(plan["PAR"])
        for c in range(maxch):
            st = plan["CH%d" % c]
            if len(st) != 1 or st[0][1] is None:
                return False
            off, banks = st[0]
            for i in range(16):
                if ((i % G) // 4) not in banks or not 0 <= i + off < 16:
                    continue
                src = i + off
                R, S = lane_bus[i % G], lane_bus[src % G]
                if R == 0:
                    continue
                if S == 0:  # a padding lane's product: zero x the voltage its own "parent" moves read
                    if isinstance(par[src], int) and par[src] // G != i // G:
                        return False
                    continue
                if src // G != i // G or ch[R][c] != S:
                    return False
        return True

    # Layouts worth trying: the c-th child of every bus sits at a fixed offset o_c from its parent, so that
    # "value of my c-th child" is ONE row shift; the layout is then fixed by the offsets and the root positions.
    # Weights: how many dwords per Newton trip go through each class (V and the Newton step from the parent;
    # W and, per elimination level, Schur complement + reduced right-hand side from each child).
    w_par = 4 + 4 * tt["MAXD"]
    w_ch = 4 + 12 * sum(1 for x in tt["NCH_H"] if x > 0)
    offs = [o for o in range(-(G - 1), G) if o != 0]
    best = None
    for o in itertools.product(offs, repeat=maxch):
        if len(set(o)) < maxch:
            continue
        for rpos in itertools.permutations(range(G), len(roots)):
            pos, ok, todo = {}, True, []
            for rb, rp in zip(roots, rpos):
                pos[rb] = rp
                todo.append(rb)
            while todo and ok:
                b = todo.pop()
                for c in range(maxch):
                    k = ch[b][c]
                    if k > 0:
                        pos[k] = pos[b] + o[c]
                        ok = ok and 0 <= pos[k] < G
                        todo.append(k)
            if not ok or len(set(pos.values())) != len(buses):
                continue
            lane_bus = [0] * G
            for b, l in pos.items():
                lane_bus[l] = b
            r = solve_class(lane_bus, {pos[b]: pos[parent[b]] for b in buses if parent[b] > 0}, False)
            if r is None:
                continue
            plan, cost = {"PAR": r[1] if r else []}, (r[0] if r else 0) * w_par
            for c in range(maxch):
                r = solve_class(lane_bus, {pos[b]: pos[ch[b][c]] for b in buses if ch[b][c] > 0}, True)
                if r is None:
                    plan = None
                    break
                plan["CH%d" % c] = r[1] if r else []
                cost += (r[0] if r else 0) * w_ch
            if plan is None:
                continue
            # equal move counts: a layout whose child moves deliver, to every bus lane they write, the value of that
            # bus's child or the zero of a padding lane of the same environment lets the child sums stand without a
            # predicate (csrc/anm_group.hpp: child_moves_land_on_parents_or_zero -- the compile-time check of the same)
            key = (cost, 0 if child_moves_safe(lane_bus, plan) else 1)
            if best is None or key < best[0]:
                best = (key, pos, lane_bus, plan)
    if best is None:
        return None
    (cost, _), pos, lane_bus, plan = best

    def encode(st):
        out = []
        for off, banks in st:
            ctrl = (0x100 + off) if off > 0 else (0x110 - off)  # row_shl:n reads lane i + n, row_shr:n lane i - n
            mask = 0xF if banks is None else sum(1 << (b + g * (G // 4)) for b in banks for g in range(ngr))
            out.append((ctrl, mask, 1 if banks is None else 0))
        return out

    return cost, dict(GRP=G, POS=[0] + [pos[b] for b in buses], LANE_BUS=lane_bus,
                      plan={k: encode(v) for k, v in plan.items()})


def hybrid_plan(n_bus, tt):
    """Lane layout + per-level tables of the HYBRID hand-overs of the lane-group Newton trip (csrc/anm_group.hpp), for trees
    that have no all-DPP layout (dpp_plan: a random feeder with up to five children per bus).

    The elimination goes by height: a bus of height h folds the Schur complements its children published, inverts its pivot,
    publishes its own.  Two observations cut what that costs through LDS:

    * every bus of height h >= 1 has a child of height exactly h - 1 -- the one its own elimination waits for.  The buses are
      laid out along these HEAVY edges (vertical paths top -> heavy child -> ..., consecutive lanes, never across a 16-lane
      DPP row): the heavy child's six values then arrive by `row_shl:1` moves, no LDS round trip on the longest chain;
    * a LIGHT child of height k < h - 1 has published long before level h: its parent folds it at level k + 1 (while it has
      nothing else to do), so a level reads as many slots as some bus has light children of height EXACTLY h - 1 -- for the
      30-bus feeder 2 + 1 + 1 + 2 instead of 3 + 4 + 3 + 5 -- and the level of a bus with many children no longer pays
      for all of them at once.

    Returns None when the tree is a chain-free corner case the layout cannot place (no padding lane left)."""
    parent, height, maxch = tt["PARENT"], tt["HEIGHT"], tt["MAXCH"]
    G, maxh = tt["GRP"], tt["MAXH"]
    ch = [[c for c in tt["CH"][b * maxch:(b + 1) * maxch] if c > 0] for b in range(n_bus)]
    ch[0] = []
    size = [1] * n_bus
    for b in sorted(range(1, n_bus), key=lambda b: height[b]):
        for c in ch[b]:
            size[b] += size[c]
    # heavy child: height h - 1, largest subtree among those (ties: lowest bus)
    heavy = [0] * n_bus
    for b in range(1, n_bus):
        cands = [c for c in ch[b] if height[c] == height[b] - 1]
        if cands:
            heavy[b] = max(cands, key=lambda c: (size[c], -c))
    is_heavy = [False] * n_bus
    for b in range(1, n_bus):
        if heavy[b]:
            is_heavy[heavy[b]] = True
    paths = []
    for b in range(1, n_bus):
        if not is_heavy[b]:
            p = [b]
            while heavy[p[-1]]:
                p.append(heavy[p[-1]])
            paths.append(p)
    # pack the paths into the 16-lane DPP rows of the group (first fit, longest first); a path that fits nowhere is cut (the
    # child at the cut becomes a light child)
    n_rows = max(1, G // 16)
    row_len = min(G, 16)
    rows = [[] for _ in range(n_rows)]
    free = [row_len] * n_rows
    todo = sorted(paths, key=lambda p: (-len(p), p[0]))
    while todo:
        p = todo.pop(0)
        fit = [r for r in range(n_rows) if free[r] >= len(p)]
        if fit:
            r = fit[0]
            rows[r].append(p)
            free[r] -= len(p)
            continue
        r = max(range(n_rows), key=lambda r: free[r])
        if free[r] == 0:
            return None
        k = free[r]
        head, tail = p[:k], p[k:]
        heavy[head[-1]] = 0
        is_heavy[tail[0]] = False
        rows[r].append(head)
        free[r] = 0
        todo.append(tail)
        todo.sort(key=lambda p: (-len(p), p[0]))
    if sum(free) < 1:
        return None   # the hand-overs through LDS need one padding lane (the neutral slot)
    pos, lane_bus = [0] * n_bus, [0] * G
    for r in range(n_rows):
        lane = r * row_len
        for p in rows[r]:
            for b in p:
                pos[b] = lane
                lane_bus[lane] = b
                lane += 1
    # light children by level: LCH[h][b] = the children of b folded at level h (height h - 1, not its DPP child)
    lch = [[[] for _ in range(n_bus)] for _ in range(maxh + 1)]
    for b in range(1, n_bus):
        for c in ch[b]:
            if c != heavy[b]:
                lch[height[c] + 1][b].append(c)
    nlh = [max([len(x) for x in lch[h]] + [0]) for h in range(maxh + 1)]
    maxl = max(nlh + [1])
    flat = []
    for h in range(maxh + 1):
        for b in range(n_bus):
            flat += lch[h][b] + [-1] * (maxl - len(lch[h][b]))
    return dict(POS=pos, LANE_BUS=lane_bus, HEAVY=heavy, NLH=nlh, MAXL=maxl, LCH=flat,
                ALL_HEAVY=int(all(heavy[b] for b in range(1, n_bus) if height[b] >= 1)))


def lane_pack(n_bus, tt, lane_bus, pos, hy):
    """What a lane of a group needs to know about its place in the tree, as ONE row of 32-bit words per lane (csrc/anm_group.hpp:
    LaneView::init reads its row with one vector load): the tables indexed by bus -- lane -> bus -> parent -> lane of the
    parent -- cost every wavefront a chain of three dependent loads before its first Newton trip.  Lanes are group-relative.
      word 0: bus (7 bits) | height + 1 (7) | depth + 1 (7) | number of children (4) | lane of the parent or a padding lane (6)
              | bit 31: the heavy child sits in the next lane (hybrid plan)
      words 1 ..: lanes of the children, a byte each (no such child: a padding lane)
      then (hybrid plan): lanes of the light children folded at level 1, 2, ..., a byte each, in level order
      then: height + 1 of every child, a byte each (0: no such child)
      last word: the lane that plays bus l + 1 (lane l does the input / output of bus l + 1 in the tree kernel)
    Returns (words per lane, first word of the light-child list, first word of the child heights, flat list) or None (a bus
    with more than 15 children)."""
    G, maxch = tt["GRP"], tt["MAXCH"]
    if maxch > 15:
        return None
    pad = lane_bus.index(0)
    nw_ch = (maxch + 3) // 4
    slots = []   # (level, j) of the hybrid plan's light-child slots
    if hy is not None:
        for h in range(tt["MAXH"] + 1):
            slots += [(h, j) for j in range(hy["NLH"][h])]
    nw_lq = (len(slots) + 3) // 4
    nw_hh = (maxch + 3) // 4     # heights of the children, a byte each (height + 1; 0: no such child)
    nw = 1 + nw_ch + nw_lq + nw_hh + 1   # last word: the lane that plays bus l + 1 (where lane l, which does bus l + 1's I/O, finds it)
    out = []
    for l in range(G):
        b = lane_bus[l]
        words = [0] * nw
        par = tt["PARENT"][b] if b else 0
        pl = pos[par] if (b and par > 0) else pad
        ch = [c for c in tt["CH"][b * maxch:(b + 1) * maxch]] if b else [-1] * maxch
        words[0] = (b & 0x7F) | (((tt["HEIGHT"][b] + 1) if b else 0) << 7) | (((tt["DEPTH"][b] + 1) if b else 0) << 14) \
            | ((tt["NCH"][b] if b else 0) << 21) | (pl << 25)
        if hy is not None and b and hy["HEAVY"][b]:
            words[0] |= 1 << 31
        for c in range(maxch):
            lane = pos[ch[c]] if ch[c] > 0 else pad
            words[1 + c // 4] |= lane << (8 * (c % 4))
        for q, (h, j) in enumerate(slots):
            c = hy["LCH"][(h * n_bus + b) * hy["MAXL"] + j] if b else -1
            lane = pos[c] if c > 0 else pad
            words[1 + nw_ch + q // 4] |= lane << (8 * (q % 4))
        for c in range(maxch):
            if ch[c] > 0:
                words[1 + nw_ch + nw_lq + c // 4] |= (tt["HEIGHT"][ch[c]] + 1) << (8 * (c % 4))
        words[nw - 1] = pos[l + 1] if l + 1 < n_bus else l
        out += words
    return nw, 1 + nw_ch, 1 + nw_ch + nw_lq, out


def _arr(name, values, typ="int"):
    vals = list(values)
    body = ", ".join(str(int(v)) for v in vals) if vals else "0"
    return "  static constexpr %s %s[%d] = {%s};" % (typ, name, max(len(vals), 1), body)


def emit_header(topo, name=None) -> str:
    n_bus, branches, devices = topo
    name = name or topology_name(topo)
    nd = len(devices)
    loads = [k for k, (t, _) in enumerate(devices) if t == LOAD]
    gens = [k for k, (t, _) in enumerate(devices) if t in (CLASSICAL, RENEWABLE)]
    des = [k for k, (t, _) in enumerate(devices) if t == STORAGE]
    setp = sorted(gens + des)
    slot, sset = [-1] * nd, [-1] * nd
    for lst in (loads, gens, des):
        for s, k in enumerate(lst):
            slot[k] = s
    for s, k in enumerate(setp):
        sset[k] = s
    # Y pattern (row-major): diagonal of every bus that has a branch + both directions of each branch
    pat = set()
    for f, t in branches:
        pat.update({(f, t), (t, f), (f, f), (t, t)})
    for i in range(n_bus):
        pat.add((i, i))
    ypat = sorted(pat)
    sym = symbolic_block_lu(n_bus, branches)
    yblk = [sym["blocks"].get((i, j), -1) if (i != 0 and j != 0) else -1 for (i, j) in ypat]
    diag = [-1] + [sym["blocks"][(u, u)] for u in range(1, n_bus)]
    lines = [
        "// Generated by gym_anm_amd/codegen.py -- topology descriptor '%s'." % name,
        "// signature: %s" % topology_signature(topo),
        "#pragma once",
        "// internal linkage: several per-topology libraries can live in one process without their",
        "// template instantiations (same mangled names) being merged by the dynamic linker",
        "namespace {",
        "struct Topo {",
        '  static constexpr const char* NAME = "%s";' % name,
        '  static constexpr const char* SIGNATURE = "%s";' % topology_signature(topo),
        "  static constexpr int NB = %d, NU = %d, NBR = %d, ND = %d;" % (n_bus, n_bus - 1, len(branches), nd),
        "  static constexpr int NLOAD = %d, NGEN = %d, NDES = %d, NSET = %d;" % (len(loads), len(gens), len(des), len(setp)),
        "  static constexpr int SDIM = %d;  // 2 ND + NDES + NGEN" % (2 * nd + len(des) + len(gens)),
        _arr("BR_F", [f for f, _ in branches]),
        _arr("BR_T", [t for _, t in branches]),
        _arr("DEV_TYPE", [t for t, _ in devices]),
        _arr("DEV_BUS", [b for _, b in devices]),
        _arr("DEV_SLOT", slot),
        _arr("DEV_SET", sset),
        "  static constexpr int NNZ = %d;" % len(ypat),
        _arr("Y_I", [i for i, _ in ypat]),
        _arr("Y_J", [j for _, j in ypat]),
        _arr("YBLK", yblk),
        "  static constexpr int NBLK = %d, NFILL = %d;" % (len(sym["blocks"]), len(sym["fill"])),
        _arr("FILL_BLK", sym["fill"]),
        _arr("DIAG", diag),
        _arr("PIV", sym["piv"]),
        "  static constexpr int NL = %d, NUOP = %d, NBS = %d;" % (len(sym["l_i"]), len(sym["u_dst"]), len(sym["b_j"])),
        _arr("L_I", sym["l_i"]),
        _arr("L_BLK", sym["l_blk"]),
        _arr("LSEG", sym["lseg"]),
        _arr("U_DST", sym["u_dst"]),
        _arr("U_SRC", sym["u_src"]),
        _arr("USEG", sym["useg"]),
        _arr("B_J", sym["b_j"]),
        _arr("B_BLK", sym["b_blk"]),
        _arr("BSEG", sym["bseg"]),
    ]
    tt = tree_tables(n_bus, branches, ypat)
    if tt is None:
        lines += ["  static constexpr int TREE = 0, GRP = 64;"]
    else:
        dp = None if os.environ.get("ANM_NO_DPP") else dpp_plan(n_bus, tt)  # ANM_NO_DPP: tuning switch
        lines += [
            "  // rooted-tree view (lane-group Newton continuation, csrc/anm_group.hpp): lane l <-> bus l + 1",
            "  static constexpr int TREE = 1, GRP = %d, T_MAXCH = %d, T_MAXH = %d, T_MAXD = %d;"
            % (dp["GRP"] if dp else tt["GRP"], tt["MAXCH"], tt["MAXH"], tt["MAXD"]),
            _arr("T_PARENT", tt["PARENT"]), _arr("T_HEIGHT", tt["HEIGHT"]), _arr("T_DEPTH", tt["DEPTH"]),
            _arr("T_NCH", tt["NCH"]), _arr("T_CH", tt["CH"]), _arr("T_NCH_H", tt["NCH_H"]),
            _arr("T_ZBB", tt["ZBB"]), _arr("T_ZBP", tt["ZBP"]), _arr("T_ZPB", tt["ZPB"]),
        ]  # fmt: skip
        hy = None if (dp is not None or os.environ.get("ANM_NO_HYBRID")) else hybrid_plan(n_bus, tt)  # ANM_NO_HYBRID: tuning switch
        if hy is not None and lane_pack(n_bus, tt, hy["LANE_BUS"], hy["POS"], hy) is None:
            hy = None   # (a bus with more than 15 children: no packed rows, the plain LDS hand-overs)
        if hy is not None:
            lines += [
                "  // hybrid hand-overs (codegen.hybrid_plan): buses laid out along their heavy edges (DPP row_shl:1 from the child of",
                "  // height h - 1), light children through LDS slots, folded at the level right after their own",
                "  static constexpr int T_HYB = 1, T_MAXL = %d, T_ALL_HEAVY = %d;" % (hy["MAXL"], hy["ALL_HEAVY"]),
                _arr("T_HEAVY", hy["HEAVY"]), _arr("T_NLH", hy["NLH"]), _arr("T_LCH", hy["LCH"]),
            ]
        else:
            lines += ["  static constexpr int T_HYB = 0, T_MAXL = 1, T_ALL_HEAVY = 0;", _arr("T_HEAVY", [0]), _arr("T_NLH", [0] * (tt["MAXH"] + 1)),
                      _arr("T_LCH", [0])]
        if dp is None:  # hand-overs by ds_bpermute_b32 / LDS slots, bus b in lane b - 1 (or the hybrid layout)
            lane_bus = [b if b < n_bus else 0 for b in range(1, tt["GRP"] + 1)]
            pos = [0] + list(range(n_bus - 1))
            if hy is not None:
                lane_bus, pos = hy["LANE_BUS"], hy["POS"]
            lines += ["  static constexpr int T_DPP = 0;", _arr("T_LANE_BUS", lane_bus),
                      _arr("T_POS", pos)]
            lines += ["  static constexpr int T_PAR_N = 0;", _arr("T_PAR_CTRL", [0, 0]), _arr("T_PAR_BANK", [0, 0]),
                      _arr("T_PAR_FULL", [0, 0]),
                      _arr("T_CH_N", [0] * tt["MAXCH"]), _arr("T_CH_CTRL", [0] * (2 * tt["MAXCH"])),
                      _arr("T_CH_BANK", [0] * (2 * tt["MAXCH"])), _arr("T_CH_FULL", [0] * (2 * tt["MAXCH"]))]
        else:
            pl = dp["plan"]
            pad2 = lambda st: [x for x in st] + [(0, 0, 0)] * (2 - len(st))
            chn, chc, chb, chf = [], [], [], []
            for c in range(tt["MAXCH"]):
                st = pl["CH%d" % c]
                chn.append(len(st))
                chc += [x[0] for x in pad2(st)]
                chb += [x[1] for x in pad2(st)]
                chf += [x[2] for x in pad2(st)]
            lines += [
                "  // hand-overs as DPP row shifts (codegen.dpp_plan): lane of each bus, bus of each lane (0 = padding)",
                "  static constexpr int T_DPP = 1;", _arr("T_LANE_BUS", dp["LANE_BUS"]), _arr("T_POS", dp["POS"]),
                "  static constexpr int T_PAR_N = %d;" % len(pl["PAR"]),
                _arr("T_PAR_CTRL", [x[0] for x in pad2(pl["PAR"])]), _arr("T_PAR_BANK", [x[1] for x in pad2(pl["PAR"])]),
                _arr("T_PAR_FULL", [x[2] for x in pad2(pl["PAR"])]),
                _arr("T_CH_N", chn), _arr("T_CH_CTRL", chc), _arr("T_CH_BANK", chb), _arr("T_CH_FULL", chf),
            ]  # fmt: skip
    if tt is not None:
        if dp is not None:
            lane_bus_, pos_ = dp["LANE_BUS"], dp["POS"]
        elif hy is not None:
            lane_bus_, pos_ = hy["LANE_BUS"], hy["POS"]
        else:
            lane_bus_, pos_ = [b if b < n_bus else 0 for b in range(1, tt["GRP"] + 1)], [0] + list(range(n_bus - 1))
        lp = lane_pack(n_bus, tt, lane_bus_, pos_, hy if dp is None else None)
        if lp is None:
            lines += ["  static constexpr int T_LP_NW = 0, T_LP_LQ = 0, T_LP_HH = 0;", "  alignas(16) static constexpr unsigned T_LP[1] = {0};"]
        else:
            lines += ["  // one row of words per lane: its place in the tree (codegen.lane_pack)",
                      "  static constexpr int T_LP_NW = %d, T_LP_LQ = %d, T_LP_HH = %d;" % (lp[0], lp[1], lp[2]),
                      "  alignas(16) static constexpr unsigned T_LP[%d] = {%s};" % (len(lp[3]), ", ".join("0x%xu" % w for w in lp[3]))]
        # register hand-overs (DPP / ds_bpermute): the child classes a level folds -- class c at level h when some bus has a
        # c-th child of height h - 1 (a child is folded at the level right after its own, whatever the height of its parent)
        maxch_ = tt["MAXCH"]
        cls_h = [0] * ((tt["MAXH"] + 1) * maxch_)
        for b in range(1, n_bus):
            for c in range(maxch_):
                k = tt["CH"][b * maxch_ + c]
                if k > 0:
                    cls_h[(tt["HEIGHT"][k] + 1) * maxch_ + c] = 1
        lines += [_arr("T_CLS_H", cls_h)]
    lines += [
        "};",
        "}  // namespace",
        "",
    ]
    return "\n".join(lines)


# --------------------------------------------------------------------------------------------
# building
# --------------------------------------------------------------------------------------------
def _tag() -> str:
    """Kernel-tuning experiments: ANM_BUILD_TAG=<tag> ANM_EXTRA_HIPCC_FLAGS="..." build and load
    side-by-side variants (libanm_<topology>.<tag>.so) without touching the default library."""
    t = os.environ.get("ANM_BUILD_TAG", "")
    return "." + t if t else ""


def lib_path(name: str, mpc_only=False) -> str:
    return os.path.join(BUILD_DIR, "lib%s_%s%s.so" % ("mpc" if mpc_only else "anm", name, _tag()))


def header_path(name: str) -> str:
    return os.path.join(BUILD_DIR, "topo_%s.h" % name)


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


# -ffp-contract=on: a*b+c is fused only inside one source expression (front-end rule), never across
# statements by the back end, so every kernel that instantiates the same template computes the same bits
# (the two-launch step is bit-identical to the one-launch step; measured cost: none).
# -fno-slp-vectorize: with SLP on, hipcc 7.2 packs the fp32 Jacobian arithmetic of the 30-bus kernel
# (512 registers + scratch) into v_pk_* pairs and produces wrong numbers (caught by
# tests/test_gpu_parity.py::test_transition_golden_f32_solve[case30]); scalar f32/f64 VALU code is correct.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ftemplate-depth=4096",
               "-fno-gpu-rdc", "-Wno-unused-value", "-fno-slp-vectorize", "-ffp-contract=on"]  # fmt: skip


# what csrc/anm_mpc_only.hip is made of (its own #include closure): the MPC-only libraries do not go stale when a
# step kernel changes
MPC_ONLY_SOURCES = ("anm_device.hpp", "anm_mpc.hpp", "anm_mpc_capi.inc", "anm_mpc_capi_types.inc", "anm_mpc_only.hip", "anm_pack.hpp")


def _build_stamp(header_text, flags, mpc_only=False):
    """Content hash of everything a library is built from (descriptor, kernel sources, C-ABI header,
    compiler flags).  Freshness is decided by this stamp, not by mtimes: the built libraries travel
    between machines with the tree and file times do not survive that reliably."""
    h = hashlib.sha256()
    h.update(header_text.encode())
    h.update("\0".join(flags).encode())
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))
             if f.endswith((".hpp", ".hip", ".h", ".inc")) and (not mpc_only or f in MPC_ONLY_SOURCES)]
    files.append(os.path.join(os.path.dirname(PKG_DIR), "include", "anm_mi355x.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def library_is_fresh(lib: str) -> bool:
    """True when ``lib`` was built from the kernel sources, C-ABI header and flags of this tree (its
    ``.stamp`` equals the content hash recomputed from its own descriptor header).  Used before a
    dlopen that does not go through build_library (generic mode picks any built library)."""
    base = os.path.basename(lib)
    if not (base.startswith("libanm_") and base.endswith(".so")):
        return False
    name = base[len("libanm_"):-len(".so")].split(".")[0]
    try:
        text = open(header_path(name)).read()
        stamp = open(lib + ".stamp").read().strip()
    except OSError:
        return False
    extra = os.environ.get("ANM_EXTRA_HIPCC_FLAGS", "").split()
    return stamp == _build_stamp(text, HIPCC_FLAGS + extra)


def mpc_library_is_fresh(topo) -> bool:
    """``libmpc_<topology>.so`` exists and was built from this tree's MPC sources, C-ABI header and flags"""
    lib = lib_path(topology_name(topo), mpc_only=True)
    try:
        stamp = open(lib + ".stamp").read().strip()
    except OSError:
        return False
    extra = os.environ.get("ANM_EXTRA_HIPCC_FLAGS", "").split()
    return os.path.exists(lib) and stamp == _build_stamp(emit_header(topo, topology_name(topo)), HIPCC_FLAGS + extra, True)


def build_library(topo, name=None, force=False, verbose=False, extra_flags=(), mpc_only=False):
    """Write the descriptor and compile ``libanm_<name>.so`` for gfx950 (no GPU needed to build).

    ``mpc_only``: ``libmpc_<name>.so`` instead -- the MPC kernel and the ``anm_mpc_*`` entry points alone
    (csrc/anm_mpc_only.hip), for networks whose step runs on the table-driven kernels of another library.

    Safe to call from several processes at once (one rank per GPU): the compile is serialised by a
    lock file and the library is put in place by an atomic rename."""
    import fcntl

    name = name or topology_name(topo)
    os.makedirs(BUILD_DIR, exist_ok=True)
    hdr, lib = header_path(name), lib_path(name, mpc_only)
    text = emit_header(topo, name)
    extra_flags = list(extra_flags) + os.environ.get("ANM_EXTRA_HIPCC_FLAGS", "").split()
    stamp = _build_stamp(text, HIPCC_FLAGS + extra_flags, mpc_only)
    stamp_path = lib + ".stamp"

    def fresh():
        try:
            return os.path.exists(lib) and open(stamp_path).read().strip() == stamp
        except OSError:
            return False

    if fresh() and not force:
        return lib
    hipcc = hipcc_path()
    if hipcc is None:
        raise E.HipExtensionError(
            "no prebuilt gfx950 library for topology '%s' (%s) and hipcc was not found to build it"
            % (name, topology_signature(topo))
        )
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if fresh() and not force:  # another process built it while this one waited
                return lib
            if not os.path.exists(hdr) or open(hdr).read() != text:
                tmp_hdr = "%s.%d.tmp" % (hdr, os.getpid())
                with open(tmp_hdr, "w") as f:
                    f.write(text)
                os.replace(tmp_hdr, hdr)
            tmp = "%s.%d.tmp" % (lib, os.getpid())
            cmd = [hipcc] + HIPCC_FLAGS + extra_flags + [
                '-DANM_TOPO_HEADER="%s"' % hdr,
                "-I", os.path.join(os.path.dirname(PKG_DIR), "include"),
                os.path.join(CSRC, "anm_mpc_only.hip" if mpc_only else "anm_capi.hip"),
                "-o", tmp,
            ]  # fmt: skip
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.unlink(tmp)
                raise E.HipExtensionError(
                    "hipcc failed for topology '%s':\n%s\n%s" % (name, res.stdout[-4000:], res.stderr[-8000:])
                )
            os.replace(tmp, lib)
            with open(stamp_path + ".tmp", "w") as f:
                f.write(stamp + "\n")
            os.replace(stamp_path + ".tmp", stamp_path)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


def stock_topologies():
    """Topologies built ahead of time by ``__graft_entry__.build()``: the ANM6 case, the 30-bus
    synthetic feeder of BASELINE.json config 4 and the small nets of the reference's own tests."""
    from . import networks
    from .model import NetworkModel

    nets = {
        "anm6": networks.anm6_network(),
        "case30": networks.synthetic_radial_network(30, 0),
        "2bus": networks.two_bus_network(),
        "3bus": networks.three_bus_loop_network(),
    }
    return {k: NetworkModel(v, 0.25, 100).topology() for k, v in nets.items()}


# MPC size classes: (n_load, n_gen, n_des, n_bus, n_branch) upper bounds of kernels compiled ONCE (libmpc_class_<name>.so)
# that serve any smaller network padded into them (csrc/anm_mpc.hpp: is_padded) -- a network that steps on the
# table-driven kernels (no library of its own) then gets its MPC agent without a compile.  The storage count must match.
# "s*": at most 72 rows per stage, the working set of a lane in registers; "l*": up to 156 rows, row arrays in scratch.
MPC_CLASSES = {
    "s0": (8, 2, 0, 18, 22), "s1": (8, 2, 1, 18, 20), "s2": (8, 2, 2, 18, 18),
    "l0": (24, 4, 0, 42, 44), "l1": (24, 4, 1, 42, 44), "l2": (24, 4, 2, 42, 44),
}


def mpc_class_for(model_or_counts):
    """smallest size class that takes a network of (n_load, n_gen, n_des, n_bus, n_branch), or None"""
    nl, ng, ns, nb, nbr = model_or_counts
    for name in ("s%d" % ns, "l%d" % ns):
        c = MPC_CLASSES.get(name)
        if c and nl <= c[0] and ng <= c[1] and ns == c[2] and nb <= c[3] and nbr <= c[4]:
            return name
    return None


def mpc_class_header(name) -> str:
    nl, ng, ns, nb, nbr = MPC_CLASSES[name]
    return "\n".join([
        "// Generated by gym_anm_amd/codegen.py -- MPC size class '%s' (no topology: csrc/anm_mpc.hpp, is_padded)." % name,
        "#pragma once",
        "namespace {",
        "struct Topo {",
        '  static constexpr const char* NAME = "mpcclass_%s";' % name,
        '  static constexpr const char* SIGNATURE = "MPCCLASS|L%d|G%d|S%d|B%d|R%d";' % (nl, ng, ns, nb, nbr),
        "  static constexpr int NB = %d, NBR = %d, ND = %d;" % (nb, nbr, nl + ng + ns + 1),
        "  static constexpr int NLOAD = %d, NGEN = %d, NDES = %d;" % (nl, ng, ns),
        "  static constexpr bool MPC_PADDED = true;",
        "};",
        "}  // namespace",
        "",
    ])


def build_mpc_class(name, force=False, verbose=False):
    """Compile ``libmpc_class_<name>.so`` (csrc/anm_mpc_only.hip over the class header) for gfx950."""
    import fcntl

    os.makedirs(BUILD_DIR, exist_ok=True)
    hdr = os.path.join(BUILD_DIR, "mpcclass_%s.h" % name)
    lib = os.path.join(BUILD_DIR, "libmpc_class_%s%s.so" % (name, _tag()))
    text = mpc_class_header(name)
    extra = os.environ.get("ANM_EXTRA_HIPCC_FLAGS", "").split()
    stamp = _build_stamp(text, HIPCC_FLAGS + extra, True)
    stamp_path = lib + ".stamp"

    def fresh():
        try:
            return os.path.exists(lib) and open(stamp_path).read().strip() == stamp
        except OSError:
            return False

    if fresh() and not force:
        return lib
    hipcc = hipcc_path()
    if hipcc is None:
        raise E.HipExtensionError("no prebuilt MPC size-class library '%s' and hipcc was not found to build it" % name)
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if fresh() and not force:
                return lib
            with open(hdr + ".tmp", "w") as f:
                f.write(text)
            os.replace(hdr + ".tmp", hdr)
            tmp = "%s.%d.tmp" % (lib, os.getpid())
            cmd = [hipcc] + HIPCC_FLAGS + extra + ['-DANM_TOPO_HEADER="%s"' % hdr, "-I", os.path.join(os.path.dirname(PKG_DIR), "include"),
                   os.path.join(CSRC, "anm_mpc_only.hip"), "-o", tmp]  # fmt: skip
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.unlink(tmp)
                raise E.HipExtensionError("hipcc failed for MPC size class '%s':\n%s\n%s" % (name, res.stdout[-4000:], res.stderr[-8000:]))
            os.replace(tmp, lib)
            with open(stamp_path + ".tmp", "w") as f:
                f.write(stamp + "\n")
            os.replace(stamp_path + ".tmp", stamp_path)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


def stock_mpc_only_topologies():
    """MPC-only libraries built ahead of time: a meshed 20-bus network (the GPU tests' example of a network that steps
    in generic mode and has its MPC kernel compiled for its own topology)."""
    from . import networks
    from .model import NetworkModel

    return {"mpc:mesh20": NetworkModel(networks.synthetic_meshed_network(20, 3, 6), 0.25, 100).topology()}


def build_stock(force=False, verbose=False):
    """Build the stock libraries (in parallel: each is one independent hipcc invocation)."""
    from concurrent.futures import ThreadPoolExecutor

    topos, mpc_topos = stock_topologies(), stock_mpc_only_topologies()
    with ThreadPoolExecutor(max_workers=min(8, len(topos) + len(mpc_topos) + len(MPC_CLASSES))) as pool:
        futs = {nm: pool.submit(build_library, topo, None, force, verbose) for nm, topo in topos.items()}
        futs.update({nm: pool.submit(build_library, topo, None, force, verbose, (), True) for nm, topo in mpc_topos.items()})
        futs.update({"mpcclass:" + c: pool.submit(build_mpc_class, c, force, verbose) for c in MPC_CLASSES})
        return {nm: f.result() for nm, f in futs.items()}
