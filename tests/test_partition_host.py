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
