"""Caller-built meshes (mgamd_tria_create_from_leaves) beyond the benchmark's four geometries: RANDOMLY refined,
2:1-balanced octrees of the cube -- hanging faces, lone hanging edges, refinement islands, cells of four levels side by side --
through the whole path (DoF tables with bricks, level operators, transfers incl. the fused ones, Chebyshev, V-cycle, CG) against
the independent numpy oracle on the same leaves, matched through the geometric DoF keys.  The named geometries are regular enough
that a table error can hide (the ownership plan of the fused transfers broke on the annulus at NRefGlobal 8 only)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def random_mesh(oracle, seed, n_global, rounds, fraction):
    rng = np.random.default_rng(seed)
    leaves = oracle.refine_global({(0, 0, 0, 0)}, n_global)
    for _ in range(rounds):
        cells = sorted(leaves)
        # a random blob (cells near a random point) plus a sprinkle of single cells
        c = rng.random(3)
        centre = lambda cell: (np.array(cell[1:]) + 0.5) / (1 << cell[0])
        near = [cell for cell in cells if np.linalg.norm(centre(cell) - c) < 0.3 and rng.random() < 0.7]
        single = [cells[t] for t in rng.choice(len(cells), max(1, int(fraction * len(cells))), replace=False)]
        leaves = oracle.refine(leaves, list(set(near) | set(single)))
    return leaves


# (seed, global refinements, random rounds, sprinkle fraction, degree)
CASES = [(1, 3, 2, 0.02, 1), (2, 3, 1, 0.02, 2), (3, 3, 1, 0.01, 4), (4, 4, 1, 0.004, 1), (5, 2, 2, 0.05, 4), (6, 2, 3, 0.02, 2), (7, 3, 1, 0.0, 4)]


@pytest.mark.parametrize("seed,n_global,rounds,fraction,p", CASES)
def test_random_octree_through_the_whole_path(mgamd, oracle, ctx, seed, n_global, rounds, fraction, p):
    leaves = random_mesh(oracle, seed, n_global, rounds, fraction)
    arr = np.array(sorted(leaves), dtype=np.int64)
    assert len(set(arr[:, 0].tolist())) >= 2  # cells of several levels
    tria = mgamd.Triangulation.from_leaves(arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3])
    assert tria.n_cells == len(leaves) and tria.n_cells_hn > 0
    h = mgamd.Hierarchy(ctx, tria, None, p, "HMG-global", coarse_solver="amg")  # (bricks where they fit: max_brick default)
    # the oracle's coarsening sequence on the same leaves: the same meshes level by level
    meshes = oracle.coarsening_sequence(leaves)
    assert [t.n_cells for t in h.trias] == [len(m) for m in meshes]
    levels = [oracle.Level(m, p, d.keys()) for m, d in zip(meshes, h.dofs)]
    P = [None] + [oracle.build_transfer(levels[l], levels[l - 1]) for l in range(1, len(levels))]
    omg = oracle.Multigrid(levels, P, 3, coarse="direct")
    Lf = levels[-1]
    rng = np.random.default_rng(100 + seed)
    # operator, right-hand side, inverse diagonal on every level
    for lv, op, sm in zip(levels, h.operators, h.smoothers):
        u = rng.standard_normal(lv.n)
        vu, vA = op.initialize_dof_vector().from_host(u), op.initialize_dof_vector()
        op.vmult(vA, vu)
        assert rel_err(vA.to_host(), lv.A @ u) < 1e-13
    b = h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    assert np.abs(b.to_host() - Lf.rhs_constant).max() < 1e-14
    # transfers
    for l in range(1, len(levels)):
        xc, xf0 = rng.standard_normal(levels[l - 1].n), rng.standard_normal(levels[l].n)
        vc, vf = h.operators[l - 1].initialize_dof_vector().from_host(xc), h.operators[l].initialize_dof_vector().from_host(xf0)
        h.transfers[l].prolongate_and_add(vf, vc)
        assert rel_err(vf.to_host(), xf0 + P[l] @ xc) < 1e-13
        rf, dc0 = rng.standard_normal(levels[l].n), rng.standard_normal(levels[l - 1].n)
        vr, vd = h.operators[l].initialize_dof_vector().from_host(rf), h.operators[l - 1].initialize_dof_vector().from_host(dc0)
        h.transfers[l].restrict_and_add(vd, vr)
        assert rel_err(vd.to_host(), dc0 + P[l].T @ rf) < 1e-13
    # V-cycle and solve
    r = rng.standard_normal(Lf.n)
    r[Lf.constrained] = 0.0
    vr, vz = mgamd.Vector(ctx, Lf.n).from_host(r), mgamd.Vector(ctx, Lf.n)
    h.mg.vmult(vz, vr)
    assert rel_err(vz.to_host(), omg.vcycle(r)) < 1e-11
    xref, itref, hist = oracle.pcg(Lf.A, Lf.rhs_constant, omg.vcycle, 1e-4)
    x = h.fine_operator.initialize_dof_vector()
    it, res = mgamd.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    assert it == itref
    assert rel_err(x.to_host(), xref) < 1e-10


@pytest.mark.parametrize("seed,n_global,rounds,fraction,p,n_ranks", [(2, 3, 1, 0.02, 2, 3), (5, 2, 2, 0.05, 4, 2), (1, 3, 2, 0.02, 1, 4)])
def test_random_octree_sharded(mgamd, oracle, seed, n_global, rounds, fraction, p, n_ranks, monkeypatch):
    """the same caller-built meshes through the SHARDED path (partition of the caller's octree, halo plans, rank-local transfers,
    all-reduce onto the replicated levels) on simulated ranks, against the numpy oracle through the DoF keys"""
    from test_gpu_distributed_sim import key_vector, keyset, run_ranks

    monkeypatch.setenv("MGAMD_CHEB_KEY_INIT", "1")  # numbering-independent Chebyshev start vector (the sharded path has no global index)
    leaves = random_mesh(oracle, seed, n_global, rounds, fraction)
    arr = np.array(sorted(leaves), dtype=np.int64)
    meshes = oracle.coarsening_sequence(leaves)
    levels = [oracle.Level(m, p) for m in meshes]
    P = [None] + [oracle.build_transfer(levels[l], levels[l - 1]) for l in range(1, len(levels))]
    omg = oracle.Multigrid(levels, P, 3, coarse="direct", start_vectors=[oracle.key_hash_start_vector(lv) for lv in levels])
    Lf = levels[-1]
    kf = {tuple(int(v) for v in k): i for i, k in enumerate(Lf.keys)}
    r = key_vector(Lf.keys, 2)
    r[Lf.constrained] = 0.0
    zref = omg.vcycle(r)
    xref, itref, hist = oracle.pcg(Lf.A, Lf.rhs_constant, omg.vcycle, 1e-4)
    sim = mgamd.SimGroup(n_ranks)

    def rank_main(rk):
        ctx = mgamd.Context(0)
        tria = mgamd.Triangulation.from_leaves(arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3])
        h = mgamd.DistributedHierarchy(ctx, sim.comm(rk), tria, None, p, coarse_solver="amg", min_root_dofs=0, min_subset_dofs=0)
        idx = np.array([kf[k] for k in keyset(h.dofs[-1].keys())])
        op = h.fine_operator
        vr, vz = op.initialize_dof_vector().from_host(r[idx]), op.initialize_dof_vector()
        h.mg.vmult(vz, vr)
        b, x = op.initialize_dof_vector(), op.initialize_dof_vector()
        op.rhs(b)
        it, res = mgamd.solve_cg(op, h.mg, x, b, 1e-4)
        return dict(idx=idx, z=vz.to_host(), x=x.to_host(), it=it, n_dofs=h.n_dofs, layout=h.layout())

    for o in run_ranks(n_ranks, rank_main):
        assert o["n_dofs"] == Lf.n and o["layout"][-1] == n_ranks
        assert rel_err(o["z"], zref[o["idx"]]) < 1e-11
        assert o["it"] == itref
        assert rel_err(o["x"], xref[o["idx"]]) < 1e-10


@pytest.mark.parametrize("seed,n_global,rounds,fraction,p,mg_type", [(3, 3, 1, 0.01, 4, "PMG"), (5, 2, 2, 0.05, 4, "HPMG"), (6, 2, 3, 0.02, 2, "PMG"),
                                                                    (7, 3, 1, 0.0, 4, "HPMG")])
def test_random_octree_p_multigrid(mgamd, oracle, ctx, seed, n_global, rounds, fraction, p, mg_type):
    """polynomial (PMG) and hybrid (HPMG) coarsening on caller-built random octrees: the p-transfers between levels that share
    the mesh, and for HPMG the h-levels below them, against the numpy oracle (coarse level small enough for the exact solve)"""
    leaves = random_mesh(oracle, seed, n_global, rounds, fraction)
    arr = np.array(sorted(leaves), dtype=np.int64)
    tria = mgamd.Triangulation.from_leaves(arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3])
    h = mgamd.Hierarchy(ctx, tria, None, p, mg_type, coarse_solver="amg")
    assert h.mg.coarse_solver_used() == "direct"
    pseq = [p]
    while pseq[-1] > 1:
        pseq.append(max(pseq[-1] // 2, 1))
    pseq = pseq[::-1]
    if mg_type == "PMG":
        meshes, degs = [leaves] * len(pseq), pseq
    else:
        hm = oracle.coarsening_sequence(leaves)
        meshes, degs = hm + [leaves] * (len(pseq) - 1), [pseq[0]] * len(hm) + pseq[1:]
    assert [d.degree for d in h.dofs] == degs and [t.n_cells for t in h.trias] == [len(m) for m in meshes]
    levels = [oracle.Level(m, q, d.keys()) for m, q, d in zip(meshes, degs, h.dofs)]
    P = [None] + [oracle.build_transfer(levels[l], levels[l - 1]) for l in range(1, len(levels))]
    omg = oracle.Multigrid(levels, P, 3, coarse="direct")
    Lf = levels[-1]
    rng = np.random.default_rng(200 + seed)
    for l in range(1, len(levels)):
        xc, xf0 = rng.standard_normal(levels[l - 1].n), rng.standard_normal(levels[l].n)
        vc, vf = h.operators[l - 1].initialize_dof_vector().from_host(xc), h.operators[l].initialize_dof_vector().from_host(xf0)
        h.transfers[l].prolongate_and_add(vf, vc)
        assert rel_err(vf.to_host(), xf0 + P[l] @ xc) < 1e-13
        rf, dc0 = rng.standard_normal(levels[l].n), rng.standard_normal(levels[l - 1].n)
        vr, vd = h.operators[l].initialize_dof_vector().from_host(rf), h.operators[l - 1].initialize_dof_vector().from_host(dc0)
        h.transfers[l].restrict_and_add(vd, vr)
        assert rel_err(vd.to_host(), dc0 + P[l].T @ rf) < 1e-13
    r = rng.standard_normal(Lf.n)
    r[Lf.constrained] = 0.0
    vr, vz = mgamd.Vector(ctx, Lf.n).from_host(r), mgamd.Vector(ctx, Lf.n)
    h.mg.vmult(vz, vr)
    assert rel_err(vz.to_host(), omg.vcycle(r)) < 1e-11
    xref, itref, hist = oracle.pcg(Lf.A, Lf.rhs_constant, omg.vcycle, 1e-4)
    b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    it, res = mgamd.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    assert it == itref
    assert rel_err(x.to_host(), xref) < 1e-10


@pytest.mark.parametrize("seed,n_global,rounds,fraction,p", [(2, 3, 1, 0.02, 2), (6, 2, 3, 0.02, 2), (1, 3, 2, 0.02, 1), (5, 2, 2, 0.05, 4)])
def test_random_octree_local_smoothing(mgamd, oracle, ctx, seed, n_global, rounds, fraction, p):
    """HMG-local (solve_with_local_smoothing, ref:multigrid_throughput.cc:1670-1873) on caller-built random octrees: refinement
    levels with several disconnected refined regions and refinement edges of every shape -- level meshes, edge matrices,
    copy indices, V-cycle (symmetric) and solve against oracle/ls_oracle.py on the same leaves"""
    import ls_oracle

    leaves = random_mesh(oracle, seed, n_global, rounds, fraction)
    arr = np.array(sorted(leaves), dtype=np.int64)
    tria = mgamd.Triangulation.from_leaves(arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3])
    h = mgamd.Hierarchy(ctx, tria, None, p, "HMG-local", coarse_solver="amg")
    ref = ls_oracle.LocalSmoothing(leaves, None, p, numbering_keys_global=h.active_dofs.keys(), numbering_keys_levels=[d.keys() for d in h.dofs])
    n = ref.G.n
    assert h.n_dofs == n and h.mg.coarse_solver_used() == "direct" and any(d.info.n_edge for d in h.dofs)
    rng = np.random.default_rng(300 + seed)
    r, u = rng.standard_normal(n), rng.standard_normal(n)
    r[ref.G.constrained] = 0.0
    u[ref.G.constrained] = 0.0
    vr, vz, vu, vw = (mgamd.Vector(ctx, n) for _ in range(4))
    vr.from_host(r), vu.from_host(u)
    h.mg.vmult(vz, vr)
    h.mg.vmult(vw, vu)
    assert rel_err(vz.to_host(), ref.vcycle(r)) < 1e-11
    assert abs(u @ vz.to_host() - r @ vw.to_host()) < 1e-10 * abs(u @ vz.to_host())  # a symmetric preconditioner
    xref, itref, hist = ref.solve(1e-4)
    b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    it, res = mgamd.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    assert it == itref and rel_err(x.to_host(), xref) < 1e-10
