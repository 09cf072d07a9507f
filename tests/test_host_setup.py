"""CPU tests of the host logic behind the C ABI: meshes, DoF numbering, constraints, tables.
The product's matrix-free formulation (parent-resolved indices + in-cell hanging-node interpolation)
is emulated in numpy from the exported tables and compared with the oracle's assembled C^T K C."""
import ctypes as C
import re
import os

import numpy as np
import pytest

from conftest import oracle_level, ROOT

CASES = [("quadrant", 3, 1, 0), ("quadrant", 3, 2, 0), ("quadrant", 3, 4, 0), ("quadrant", 4, 1, 0), ("quadrant", 3, 4, 1),
         ("quadrant", 3, 3, 0), ("hypercube", 3, 1, 0), ("hypercube", 2, 4, 0), ("annulus", 5, 2, 0), ("circle", 4, 1, 0)]


def test_c_abi_exports_every_declared_symbol(mgamd):
    hdr = open(os.path.join(ROOT, "include", "mgamd.h")).read()
    names = sorted(set(re.findall(r"\b(mgamd_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) > 40
    # the boundary header carries no development entry points; those live in mgamd_dev.h (and are exported too)
    assert not [n for n in names if "profile" in n or "debug" in n]
    dev = sorted(set(re.findall(r"\b(mgamd_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "mgamd_dev.h")).read())))
    assert len(dev) >= 5 and not [n for n in dev if not hasattr(mgamd._lib, n)]
    missing = [n for n in names if not hasattr(mgamd._lib, n)]
    assert not missing, missing


def test_error_reporting(mgamd):
    with pytest.raises(mgamd.MgamdError, match="not implemented"):
        mgamd.Triangulation("torus", 2)
    t = mgamd.Triangulation("hypercube", 1)
    with pytest.raises(mgamd.MgamdError, match="degree"):
        mgamd.DoFs(t, 0)


def test_device_calls_fail_loudly_without_gpu(mgamd):
    try:
        import torch

        have_gpu = torch.cuda.device_count() > 0
    except Exception:
        have_gpu = False
    if have_gpu:
        pytest.skip("GPU present")
    with pytest.raises(mgamd.NoDeviceError):
        mgamd.Context(0)


def test_mesh_generators_match_oracle(mgamd, oracle):
    for geo, L in (("quadrant", 4), ("quadrant", 5), ("annulus", 5), ("annulus", 6), ("circle", 5), ("hypercube", 3), ("quadrant", 0)):
        t = mgamd.Triangulation(geo, L)
        m = oracle.create_mesh(geo, L)
        lev, i, j, k, mask = t.cells()
        assert set(zip(lev.tolist(), i.tolist(), j.tolist(), k.tolist())) == m
        assert [oracle.morton_key(c) for c in zip(lev.tolist(), i.tolist(), j.tolist(), k.tolist())] == sorted(oracle.morton_key(c) for c in m)
        assert t.n_cells_hn == sum(oracle.is_cell_constrained(m, c) for c in m)
    t = mgamd.Triangulation("quadrant_flexible", 2, 2)
    assert t.n_cells == len(oracle.create_mesh("quadrant_flexible", 2, 2))


def test_geometric_coarsening_sequence(mgamd, oracle):
    for geo, L in (("quadrant", 5), ("annulus", 6), ("hypercube", 3)):
        seq = mgamd.create_geometric_coarsening_sequence(mgamd.Triangulation(geo, L))
        ref = oracle.coarsening_sequence(oracle.create_mesh(geo, L))
        assert [t.n_cells for t in seq] == [len(m) for m in ref]
        for t, m in zip(seq, ref):
            lev, i, j, k, _ = t.cells()
            assert set(zip(lev.tolist(), i.tolist(), j.tolist(), k.tolist())) == m


def test_survey_appendix_b_counts(mgamd):
    # quadrant p=4: L=4 50,553 ; L=5 321,243 DoFs, cells 701 / 4,712, n_hn 260 / 1,083
    t4, t5 = mgamd.Triangulation("quadrant", 4), mgamd.Triangulation("quadrant", 5)
    assert (t4.n_cells, t4.n_cells_hn, t5.n_cells, t5.n_cells_hn) == (701, 260, 4712, 1083)
    assert mgamd.DoFs(t4, 4).n_dofs == 50553
    assert mgamd.DoFs(t5, 4).n_dofs == 321243
    assert mgamd.DoFs(t5, 1).n_dofs == 5703
    assert mgamd.DoFs(t5, 2).n_dofs == 42467
    a6 = mgamd.Triangulation("annulus", 6)
    assert (a6.n_cells, a6.n_cells_hn, mgamd.DoFs(a6, 1).n_dofs, mgamd.DoFs(a6, 2).n_dofs) == (6840, 5360, 9763, 71509)


def _interp(oracle, fe, mask, v, transpose):
    p = fe.p
    n = p + 1
    if not (mask >> 3):
        return v
    v = v.reshape(n, n, n).copy()
    cp = [mask & 1, (mask >> 1) & 1, (mask >> 2) & 1]
    I = [oracle.lagrange_eval(fe.nodes, 0.5 * (fe.nodes + c))[0] for c in (0, 1)]
    for d in ([0, 1, 2] if not transpose else [2, 1, 0]):
        e, f = (d + 1) % 3, (d + 2) % 3
        Ic = I[cp[d]].T if transpose else I[cp[d]]
        for ae in range(n):
            for af in range(n):
                on_e, on_f = ae == cp[e] * p, af == cp[f] * p
                if not (((mask >> (3 + e)) & 1 and on_e) or ((mask >> (3 + f)) & 1 and on_f) or ((mask >> (6 + d)) & 1 and on_e and on_f)):
                    continue
                idx = [None] * 3
                idx[d], idx[e], idx[f] = slice(None), ae, af
                sl = (idx[2], idx[1], idx[0])
                v[sl] = Ic @ v[sl]
    return v.ravel()


@pytest.mark.parametrize("geo,L,p,max_brick", CASES)
def test_tables_reproduce_assembled_operator(mgamd, oracle, geo, L, p, max_brick):
    t = mgamd.Triangulation(geo, L)
    d = mgamd.DoFs(t, p, max_brick)
    lv = oracle_level(oracle, d, geo, L, p)
    info = d.info
    assert d.n_dofs == lv.n
    first_c = info.n_interior + info.n_tail
    assert not lv.constrained[:first_c].any() and lv.constrained[first_c:].all()
    assert info.n_hanging == lv.hanging.sum() and info.n_dirichlet == (lv.dirichlet & ~lv.hanging).sum()
    if max_brick == 1:
        assert d.groups() == [(1, t.n_cells)]
    # right-hand side (ref:include/operator.h:362-413)
    assert np.abs(d.rhs_constant() - lv.rhs_constant).max() < 1e-15
    # operator
    cd = d.cell_dofs()
    lev, _, _, _, mask = t.cells()
    fe = lv.fe
    Kc = np.kron(np.kron(fe.M, fe.M), fe.K) + np.kron(np.kron(fe.M, fe.K), fe.M) + np.kron(np.kron(fe.K, fe.M), fe.M)
    x = np.random.default_rng(1).standard_normal(d.n_dofs)
    y = np.zeros(d.n_dofs)
    for ci in range(t.n_cells):
        idx = cd[ci]
        val = idx != mgamd.INVALID_DOF
        g = np.where(val, x[np.where(val, idx, 0)], 0.0)
        r = (2.0 / (1 << int(lev[ci]))) * (Kc @ _interp(oracle, fe, int(mask[ci]), g, False))
        r = _interp(oracle, fe, int(mask[ci]), r, True)
        np.add.at(y, idx[val], r[val])
    y[first_c:] = x[first_c:]
    ref = lv.A @ x
    assert np.abs(y - ref).max() < 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize("geo,L,p,mg_type", [("quadrant", 3, 2, "HMG-global"), ("quadrant", 3, 4, "PMG"), ("annulus", 5, 1, "HMG-global")])
def test_transfer_tables_reproduce_oracle_prolongation(mgamd, oracle, geo, L, p, mg_type):
    fine_t = mgamd.Triangulation(geo, L)
    if mg_type == "PMG":
        tf, tc, pf, pc = fine_t, fine_t, p, p // 2
        mf = mc = oracle.create_mesh(geo, L)
    else:
        tf, tc, pf, pc = fine_t, fine_t.coarsen(), p, p
        mf = oracle.create_mesh(geo, L)
        mc = oracle.coarsen_global(mf)
    df, dc = mgamd.DoFs(tf, pf), mgamd.DoFs(tc, pc)
    lf, lc = oracle.Level(mf, pf, df.keys()), oracle.Level(mc, pc, dc.keys())
    P = oracle.build_transfer(lf, lc)
    xc = np.random.default_rng(2).standard_normal(dc.n_dofs)
    ref = P @ xc
    fec = lc.fe
    out = np.zeros(df.n_dofs)
    for kind, nf, ci, cm, fi in mgamd.transfer_tables(df, dc):
        if kind == 0:
            E = np.eye(pc + 1)
        elif kind == 1:
            E = np.vstack([oracle.lagrange_eval(fec.nodes, 0.5 * fec.nodes)[0], oracle.lagrange_eval(fec.nodes, 0.5 + 0.5 * fec.nodes)[0][1:]])
        else:
            E = oracle.lagrange_eval(fec.nodes, oracle.gll_nodes(pf))[0]
        E3 = np.kron(np.kron(E, E), E)
        for n in range(ci.shape[0]):
            val = ci[n] != mgamd.INVALID_DOF
            g = np.where(val, xc[np.where(val, ci[n], 0)], 0.0)
            v = E3 @ _interp(oracle, fec, int(cm[n]), g, False)
            ok = fi[n] != mgamd.INVALID_DOF
            assert not np.any(out[fi[n][ok]] != 0.0)  # every fine DoF owned by exactly one patch
            out[fi[n][ok]] += v[ok]
    assert np.abs(out - ref).max() < 1e-13 * max(np.abs(ref).max(), 1)


@pytest.mark.parametrize("geo,L", [("quadrant", 4), ("annulus", 5), ("circle", 4), ("hypercube", 3)])
@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_slot_decomposition_covers_every_cell_once(mgamd, geo, L, p, monkeypatch):
    """bricks (incl. constrained 2^3 families), and single cells partition the leaves; switching the constrained families
    off only moves cells between the groups and never changes the DoF count"""
    t = mgamd.Triangulation(geo, L)
    d = mgamd.DoFs(t, p, 0)
    assert sum(B ** 3 * n for B, n in d.groups()) == t.n_cells
    monkeypatch.setenv("MGAMD_NO_HANGING_BRICKS", "1")
    d0 = mgamd.DoFs(t, p, 0)
    assert sum(B ** 3 * n for B, n in d0.groups()) == t.n_cells
    assert d0.n_dofs == d.n_dofs and d0.info.n_hanging == d.info.n_hanging and d0.info.n_dirichlet == d.info.n_dirichlet
    if p >= 2 and t.n_cells_hn > 0:
        assert dict(d.groups())[1] <= dict(d0.groups())[1]  # families absorb single cells
        assert d.info.n_interior >= d0.info.n_interior


@pytest.mark.parametrize("geo,L,p", [("quadrant", 3, 1), ("quadrant", 3, 2), ("quadrant", 3, 4), ("annulus", 5, 2), ("hypercube", 2, 3), ("circle", 4, 3)])
def test_gaussian_right_hand_side_and_distribute(mgamd, oracle, geo, L, p):
    """SimulationType "Gaussian" (ref:multigrid_throughput.cc:60-125,2294-2298): quadrature load vector of f minus the
    Dirichlet lifting of g (ref:include/operator.h:362-447), and constraints.distribute, against the textbook oracle
    (assembled K, explicit hanging-node matrix)."""
    t = mgamd.Triangulation(geo, L)
    d = mgamd.DoFs(t, p, 0)
    lv = oracle_level(oracle, d, geo, L, p)
    ref = lv.rhs_function(oracle.gaussian_rhs, oracle.gaussian_solution)
    got = d.rhs_function(1)
    assert np.abs(got - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300)
    assert np.array_equal(d.rhs_function(0), d.rhs_constant())
    x = np.random.default_rng(2).standard_normal(d.n_dofs)
    first_c = d.info.n_interior + d.info.n_tail
    x[first_c:] = 0.0
    xd, xr = d.distribute(x, 1), lv.distribute(x, oracle.gaussian_solution)
    assert np.abs(xd - xr).max() <= 1e-13 * np.abs(xr).max()
    assert np.array_equal(xd[:first_c], x[:first_c])
    # homogeneous data: only the hanging nodes move
    x0 = d.distribute(x, 0)
    assert np.abs(x0 - lv.distribute(x, lambda a, b, c: 0 * a)).max() <= 1e-13 * np.abs(x).max()


@pytest.mark.parametrize("geo,L,p", [("quadrant", 3, 1), ("quadrant", 4, 2), ("annulus", 5, 1), ("quadrant", 3, 4)])
def test_local_smoothing_level_tables(mgamd, geo, L, p):
    """HMG-local host tables against the textbook oracle: level meshes (all cells of a refinement level), DoF counts,
    refinement-edge sets (numbered [I | T | E | D]), copy_to_mg / copy_from_mg index pairs -- through the DoF keys"""
    import ls_oracle

    fine = mgamd.Triangulation(geo, L)
    act = mgamd.DoFs(fine, p, 0)
    s = ls_oracle.LocalSmoothing(geo, L, p)
    act_keys = [tuple(int(v) for v in k) for k in act.keys()]
    for l in range(fine.n_levels):
        d = mgamd.DoFs(fine.level_mesh(l), p, 0, local_smoothing_level=True)
        Lv = s.levels[l]
        assert d.n_dofs == Lv.n and d.info.n_edge == Lv.edge.sum() and d.info.n_hanging == 0
        keys = [tuple(int(v) for v in k) for k in d.keys()]
        fe = d.info.n_interior + d.info.n_tail
        assert {keys[i] for i in range(fe, fe + d.info.n_edge)} == {Lv.keys[i] for i in np.nonzero(Lv.edge)[0]}
        assert {keys[i] for i in range(fe + d.info.n_edge, d.n_dofs)} == {Lv.keys[i] for i in np.nonzero(Lv.dirichlet)[0]}
        g, li = mgamd.ls_copy_indices(act, d, l)
        og, ol = s.copy[l]
        assert {(act_keys[a], keys[b]) for a, b in zip(g, li)} == {(s.G.keys[a], Lv.keys[b]) for a, b in zip(og, ol)}
    with pytest.raises(mgamd.MgamdError, match="level"):
        fine.level_mesh(fine.n_levels)


def test_triangulation_from_caller_leaves(mgamd, oracle):
    """mgamd_tria_create_from_leaves: a caller-built octree (the leaves of the independent oracle mesh generator, shuffled)
    gives the same triangulation, DoF tables and right-hand side as the named geometry; invalid input is refused with
    MGAMD_ERR_INVALID: gaps, overlaps, out-of-range indices, and meshes that are not 2:1 balanced across faces, edges and
    corners."""
    for geo, L, p in (("quadrant", 4, 2), ("annulus", 4, 1), ("hypercube", 2, 4)):
        leaves = np.array(sorted(oracle.create_mesh(geo, L)), dtype=np.int64)  # (level, i, j, k)
        rng = np.random.default_rng(5)
        leaves = leaves[rng.permutation(len(leaves))]
        t = mgamd.Triangulation.from_leaves(leaves[:, 0], leaves[:, 1], leaves[:, 2], leaves[:, 3])
        ref = mgamd.Triangulation(geo, L)
        assert t.n_cells == ref.n_cells and t.n_levels == ref.n_levels and t.n_cells_hn == ref.n_cells_hn
        assert all(np.array_equal(a, b) for a, b in zip(t.cells(), ref.cells()))
        d, dref = mgamd.DoFs(t, p, 0), mgamd.DoFs(ref, p, 0)
        assert d.n_dofs == dref.n_dofs and np.array_equal(d.keys(), dref.keys()) and np.array_equal(d.cell_dofs(), dref.cell_dofs())
        assert np.array_equal(d.rhs_constant(), dref.rhs_constant())
        assert t.coarsen().n_cells == ref.coarsen().n_cells
    one = lambda cells: mgamd.Triangulation.from_leaves(*np.array(cells, dtype=np.int64).T)
    kids = lambda l, i, j, k: [(l + 1, 2 * i + a, 2 * j + b, 2 * k + c) for c in (0, 1) for b in (0, 1) for a in (0, 1)]
    assert one([(0, 0, 0, 0)]).n_cells == 1
    base = kids(0, 0, 0, 0)
    with pytest.raises(mgamd.MgamdError, match="cover"):
        one(base[:-1])  # gap
    with pytest.raises(mgamd.MgamdError, match="overlap"):
        one(base + [(2, 0, 0, 0)])  # a cell and its descendant
    with pytest.raises(mgamd.MgamdError, match="range"):
        one([(1, 2, 0, 0)] + base[1:])
    # refine the corner child (1,0,0,0) twice towards the cube's centre: its grandchild at the centre touches the level-1
    # cell (1,1,1,1) only in a CORNER -- balanced across faces and edges, not across corners
    lvl2 = kids(1, 0, 0, 0)
    unbalanced = [c for c in base if c != (1, 0, 0, 0)] + [c for c in lvl2 if c != (2, 1, 1, 1)] + kids(2, 1, 1, 1)
    with pytest.raises(mgamd.MgamdError, match="2:1"):
        one(unbalanced)
    # corner ONLY: every level-1 cell but the diagonally opposite one (1,1,1,1) refined once, then (2,1,1,1) once more: its child
    # at the cube's centre meets (1,1,1,1) in one vertex and is two levels finer
    corner = [(1, 1, 1, 1)] + kids(2, 1, 1, 1)
    for c in base:
        if c != (1, 1, 1, 1):
            corner += [k for k in kids(*c) if k != (2, 1, 1, 1)]
    with pytest.raises(mgamd.MgamdError, match="2:1"):
        one(corner)
    # the same refinement with the neighbours refined once is fine
    ok = [c for c in lvl2 if c != (2, 1, 1, 1)] + kids(2, 1, 1, 1)
    for c in base:
        if c != (1, 0, 0, 0):
            ok += kids(*c)
    assert one(ok).n_cells == len(ok)
