"""Host logic of the sharded path (no GPU): partition, ownership, halo plans.  The halo exchange is emulated in numpy
through the exported plan (pack -> swap segments -> combine in ascending rank order) and must reproduce the global
right-hand side on every copy."""
import numpy as np
import pytest

CASES = [("quadrant", 4, 4, 2), ("quadrant", 5, 2, 3), ("hypercube", 4, 2, 2), ("annulus", 6, 1, 4), ("quadrant", 5, 1, 8)]


def keyset(keys):
    return [tuple(int(v) for v in k) for k in keys]


def exchange_add(plans, tails):
    """tails[r]: tail segment of rank r (modified in place)"""
    sends = [t[p["pack_idx"]] for p, t in zip(plans, tails)]
    recvs = []
    for r, p in enumerate(plans):
        buf = np.zeros(len(p["pack_idx"]))
        for j, q in enumerate(p["peers"]):
            pq = plans[q]
            k = list(pq["peers"]).index(r)
            seg = sends[q][pq["peer_offset"][k]:pq["peer_offset"][k + 1]]
            assert len(seg) == p["peer_offset"][j + 1] - p["peer_offset"][j]  # symmetric plans
            buf[p["peer_offset"][j]:p["peer_offset"][j + 1]] = seg
        recvs.append(buf)
    for p, t, rv in zip(plans, tails, recvs):
        new = []
        for i, ti in enumerate(p["sh_tail"]):
            acc = 0.0
            for e in range(p["sh_ptr"][i], p["sh_ptr"][i + 1]):
                s = p["sh_src"][e]
                acc += t[ti] if s < 0 else rv[s]
            new.append(acc)
        t[p["sh_tail"]] = new


@pytest.mark.parametrize("geo,L,p,n_ranks", CASES)
def test_partition_ownership_and_halo(mgamd, geo, L, p, n_ranks):
    trias = mgamd.create_geometric_coarsening_sequence(mgamd.Triangulation(geo, L))
    part = mgamd.Partition(trias, n_ranks)
    assert 1 <= part.root_level < len(trias)
    lvl = len(trias) - 1
    owner = part.owner(lvl)
    assert set(owner.tolist()) == set(range(n_ranks))  # every rank has cells
    assert np.all(np.diff(owner.astype(int)) >= 0)  # contiguous Morton chunks
    # load balance of the weighted cut (hanging cells weigh 2): no rank more than 2x the mean
    _, _, _, _, mask = trias[lvl].cells()
    w = np.where(mask >> 3, 2.0, 1.0)
    loads = np.array([w[owner == r].sum() for r in range(n_ranks)])
    assert loads.max() < 2.0 * loads.mean()
    full = mgamd.DoFs(trias[lvl], p)
    loc = [mgamd.DoFs(trias[lvl], p, 0, part, lvl, r) for r in range(n_ranks)]
    fi = full.info
    assert sum(d.info.n_interior + d.info.n_tail_owned for d in loc) == fi.n_interior + fi.n_tail  # every free DoF owned once
    assert sum(d.info.n_dirichlet_owned for d in loc) == fi.n_dirichlet
    assert sum(d.info.n_hanging_owned for d in loc) == fi.n_hanging
    # cells of a level above the root level sit on the rank of their parent
    if lvl - 1 >= part.root_level:
        oc = part.owner(lvl - 1)
        lev_c, ic, jc, kc, _ = trias[lvl - 1].cells()
        lev_f, i_f, j_f, k_f, _ = trias[lvl].cells()
        cmap = {(int(a), int(b), int(c), int(d)): int(o) for a, b, c, d, o in zip(lev_c, ic, jc, kc, oc)}
        for a, b, c, d, o in zip(lev_f, i_f, j_f, k_f, owner):
            key = (int(a), int(b), int(c), int(d))
            par = (int(a) - 1, int(b) >> 1, int(c) >> 1, int(d) >> 1)
            assert cmap.get(key, cmap.get(par)) == int(o)
    # halo exchange of the right-hand side, emulated on the host
    plans = [d.halo_plan() for d in loc]
    gref = dict(zip(keyset(full.keys()), full.rhs_constant()))
    tails, rhs = [], []
    for d in loc:
        b = d.rhs_constant()  # local contributions only
        rhs.append(b)
        tails.append(b[d.info.n_interior:d.info.n_interior + d.info.n_tail])  # view
    exchange_add(plans, tails)
    for d, b in zip(loc, rhs):
        ref = np.array([gref[k] for k in keyset(d.keys())])
        assert np.abs(b - ref).max() < 1e-15
    # owner bookkeeping: exactly one sharer owns every shared DoF
    for r, (d, pl) in enumerate(zip(loc, plans)):
        owned_tail = d.info.n_tail_owned
        for ti, osrc in zip(pl["sh_tail"], pl["sh_owner_src"]):
            assert (osrc < 0) == (ti < owned_tail)


def test_min_root_cells_keeps_small_levels_replicated(mgamd):
    """levels with fewer cells than `min_root_cells` stay replicated (no halo exchange on latency-bound levels)"""
    trias = mgamd.create_geometric_coarsening_sequence(mgamd.Triangulation("quadrant", 5))
    default = mgamd.Partition(trias, 4)
    assert default.root_level < len(trias) - 1
    n_fine, n_prev = trias[-1].n_cells, trias[-2].n_cells
    part = mgamd.Partition(trias, 4, 2.0, min_root_cells=n_prev + 1)
    assert part.root_level == len(trias) - 1
    assert mgamd.Partition(trias, 4, 2.0, min_root_cells=n_fine + 1).root_level == len(trias) - 1  # never beyond the finest
    o = part.owner(len(trias) - 1)
    assert set(np.unique(o)) == {0, 1, 2, 3}


def test_partition_statistics_match_a_python_restatement(mgamd):
    """MGTools::print_multigrid_statistics (ref:include/mg_tools.h:9-38,140-247,267-512) for the Morton-chunk partition:
    workload efficiency / longest path, vertical efficiency (children on the parent's rank) and horizontal efficiency
    (ghost cells = foreign vertex neighbours), restated here cell by cell in Python."""
    fine = mgamd.Triangulation("quadrant", 5)
    trias = mgamd.create_geometric_coarsening_sequence(fine)
    n_ranks = 4
    part = mgamd.Partition(trias, n_ranks, 2.0, 0)
    st = part.statistics()
    root = part.root_level
    cells, owners = [], []
    for l, t in enumerate(trias):
        lev, i, j, k, _ = t.cells()
        cells.append(list(zip(lev.tolist(), i.tolist(), j.tolist(), k.tolist())))
        owners.append(part.owner(l).tolist() if l >= root else None)
    # workload
    per_level = []
    for l in range(len(trias)):
        n = [len(cells[l])] * n_ranks if owners[l] is None else [owners[l].count(r) for r in range(n_ranks)]
        per_level.append(n)
    path = sum(max(n) for n in per_level)
    assert st["workload_path_max"] == path
    assert st["workload_eff"] == pytest.approx(sum(sum(n) for n in per_level) / n_ranks / path, rel=1e-12)
    # vertical
    loc = rem = 0
    for l in range(len(trias) - 1):
        fine_idx = {c: t for t, c in enumerate(cells[l + 1])}
        for ci, (lv, i, j, k) in enumerate(cells[l]):
            if (lv, i, j, k) in fine_idx:
                continue
            for q in range(8):
                f = fine_idx.get((lv + 1, 2 * i + (q & 1), 2 * j + ((q >> 1) & 1), 2 * k + (q >> 2)))
                if f is None:
                    continue
                if owners[l] is None and owners[l + 1] is None:
                    loc += n_ranks
                elif owners[l] is None or owners[l][ci] == owners[l + 1][f]:
                    loc += 1
                else:
                    rem += 1
    assert st["vertical_eff"] == pytest.approx(loc / (loc + rem), rel=1e-12) and st["vertical_eff"] == 1.0  # children inherit the rank
    # horizontal: ghosts by geometric adjacency (closed cells intersect)
    h_loc = sum(sum(n) for n in per_level)
    h_rem = 0
    for l in range(root, len(trias)):
        L = max(c[0] for c in cells[l])
        boxes = [((i << (L - lv)), (j << (L - lv)), (k << (L - lv)), 1 << (L - lv)) for lv, i, j, k in cells[l]]
        import itertools
        grid = {}
        for t, (x, y, z, s) in enumerate(boxes):
            for gx, gy, gz in itertools.product(range(x, x + s), range(y, y + s), range(z, z + s)):
                grid[(gx, gy, gz)] = t
        for t, (x, y, z, s) in enumerate(boxes):
            ranks = set()
            for gx, gy, gz in itertools.product(range(x - 1, x + s + 1), range(y - 1, y + s + 1), range(z - 1, z + s + 1)):
                nb = grid.get((gx, gy, gz))
                if nb is not None and nb != t:
                    ranks.add(owners[l][nb])
            ranks.discard(owners[l][t])
            h_rem += len(ranks)
    assert st["horizontal_eff"] == pytest.approx((h_loc + 0.5 * h_rem) / (h_loc + h_rem), rel=1e-12)
    assert 0.5 < st["horizontal_eff"] < 1.0 and st["mem_total"] > 0
    one = mgamd.Partition(trias, 1, 2.0, 0).statistics()
    assert one["workload_eff"] == 1.0 and one["vertical_eff"] == 1.0 and one["horizontal_eff"] == 1.0


@pytest.mark.parametrize("geo,L,p,n_ranks", [("quadrant", 5, 2, 3), ("quadrant", 5, 4, 2), ("annulus", 6, 1, 4), ("hypercube", 5, 2, 2)])
def test_halo_slots_come_first(mgamd, geo, L, p, n_ranks):
    """sharded level: in every slot group the slots that touch DoFs shared with other ranks are at the front
    (group_halo_slots), so that the halo exchange of an operator application can run underneath the remaining slots"""
    trias = mgamd.create_geometric_coarsening_sequence(mgamd.Triangulation(geo, L))
    part = mgamd.Partition(trias, n_ranks, 2.0, 0)
    lvl = len(trias) - 1
    owner = part.owner(lvl)
    for rank in range(n_ranks):
        d = mgamd.DoFs(trias[lvl], p, 0, part, lvl, rank)
        plan = d.halo_plan()
        info = d.info
        shared = np.zeros(d.n_dofs, bool)
        shared[info.n_interior + plan["sh_tail"]] = True
        shared[info.n_interior + plan["pack_idx"]] = True  # everything the pack kernel reads while the interior slots run
        cg, cs = d.cell_slots()
        cd = d.cell_dofs()
        touches = {}  # (group, slot) -> touches a shared tail DoF
        for ci in np.nonzero(owner == rank)[0]:
            idx = cd[ci]
            idx = idx[idx != mgamd.INVALID_DOF]
            key = (int(cg[ci]), int(cs[ci]))
            touches[key] = touches.get(key, False) or bool(shared[idx].any())
        n_front = 0
        for (g, sl), t in touches.items():
            nh = info.group_halo_slots[g]
            if t:
                assert sl < nh, (g, sl, nh)  # every slot touching a shared tail DoF is in the halo part
            n_front += sl < nh
        assert 0 < n_front < len(touches)  # and there is an interior part to overlap with


@pytest.mark.parametrize("geo,L,n_ranks,group", [("quadrant", 6, 8, 4), ("quadrant", 6, 8, 2), ("annulus", 7, 4, 2), ("hypercube", 5, 8, 4)])
def test_two_tier_partition_is_nested_and_balanced(mgamd, geo, L, n_ranks, group):
    """Partition tiers (csrc/partition.hpp; the counterpart of the reference's agglomeration of coarse levels onto fewer processes,
    ref:multigrid_throughput.cc:379-418,1464-1501): the levels [sub_root_level, root_level) are cut into n_ranks / group parts, the
    cuts are nested (every cell of a rank on the root level and above descends from a cell of the rank's part), parts and ranks are
    contiguous Morton ranges, and the weights are balanced on both tiers."""
    trias = mgamd.create_geometric_coarsening_sequence(mgamd.Triangulation(geo, L))
    nl = len(trias)
    # root level = the finest but one, parts from two levels below it
    part = mgamd.Partition(trias, n_ranks, 2.0, min_root_cells=trias[nl - 2].n_cells, group=group, min_sub_root_cells=trias[nl - 4].n_cells)
    n_parts = n_ranks // group
    # (a level also needs 32 cells per piece, like the root level)
    sub_expected = next(l for l in range(1, nl - 1) if trias[l].n_cells >= max(32 * n_parts, trias[nl - 4].n_cells))
    assert part.group == group and part.root_level == nl - 2 and part.sub_root_level == sub_expected < nl - 2
    assert [part.n_parts(l) for l in range(nl)] == [1] * sub_expected + [n_parts] * (nl - 2 - sub_expected) + [n_ranks] * 2
    plain = mgamd.Partition(trias, n_ranks, 2.0, min_root_cells=trias[nl - 2].n_cells)
    assert plain.group == 1 and plain.sub_root_level == plain.root_level == nl - 2
    cells = []
    for t in trias:
        lev, i, j, k, mask = t.cells()
        cells.append((lev.astype(np.int64), i.astype(np.int64), j.astype(np.int64), k.astype(np.int64), mask))

    def ancestor_owner(lc, lf):
        """owner (on level mesh lc) of the ancestor of every cell of level mesh lf"""
        lev, i, j, k, _ = cells[lc]
        table = {(a, b, c, d): o for a, b, c, d, o in zip(lev.tolist(), i.tolist(), j.tolist(), k.tolist(), part.owner(lc).tolist())}
        out = []
        for a, b, c, d in zip(*[x.tolist() for x in cells[lf][:4]]):
            while (a, b, c, d) not in table:
                a, b, c, d = a - 1, b >> 1, c >> 1, d >> 1
            out.append(table[(a, b, c, d)])
        return np.array(out)

    for l in range(part.sub_root_level, nl):
        o = part.owner(l)
        assert set(np.unique(o)) == set(range(part.n_parts(l)))
        assert np.all(np.diff(o.astype(np.int64)) >= 0)  # contiguous Morton ranges, ascending
    # nesting: part of a root-level rank = rank // group; the subset levels inherit the parts; finer levels inherit the ranks
    sub, root = part.sub_root_level, part.root_level
    for l in range(sub + 1, nl):
        assert np.array_equal(ancestor_owner(sub, l), part.owner(l) // (group if l >= root else 1))
    assert np.array_equal(ancestor_owner(root, nl - 1), part.owner(nl - 1))
    # balance of the finest-level weights (hanging-node cells x 2) over the ranks and over the parts
    w = np.where((cells[-1][4] >> 3) != 0, 2.0, 1.0)
    of = part.owner(nl - 1)
    load_r = np.array([w[of == r].sum() for r in range(n_ranks)])
    load_p = np.array([w[of // group == q].sum() for q in range(n_parts)])
    assert load_p.max() / load_p.mean() < 1.15 and load_r.max() / load_r.mean() < 1.25
    # local tables of a subset level: the ranks of one group get the same piece, the halo peers are parts
    d = [mgamd.DoFs(trias[sub], 2, 0, part, sub, r) for r in range(n_ranks)]
    for r in range(n_ranks):
        assert d[r].n_dofs == d[(r // group) * group].n_dofs
        assert all(0 <= q < n_parts and q != r // group for q in d[r].halo_plan()["peers"])
    st = part.statistics()
    assert st["vertical_eff"] == 1.0 and 0 < st["workload_eff"] <= 1.0
