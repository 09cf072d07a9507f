"""Local smoothing (`HMG-local`, ref:multigrid_throughput.cc:1670-1873, ref:include/operator.h:49-120,152-226): operators
on the refinement levels with refinement-edge DoFs, the edge matrix, MGTransferMatrixFree between refinement levels,
copy_to_mg/copy_from_mg and the resulting preconditioner -- every piece against the independent textbook oracle
(oracle/ls_oracle.py: assembled level matrices, explicit edge index sets / edge matrices / transfer matrices), matched
through the geometric DoF keys."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

CASES = [("quadrant", 3, 1), ("quadrant", 4, 1), ("quadrant", 4, 2), ("quadrant", 3, 4), ("annulus", 5, 1), ("annulus", 5, 2), ("quadrant", 5, 1),
         ("hypercube", 3, 2)]


@pytest.fixture(scope="module")
def ls_hierarchies(mgamd, ctx):
    import ls_oracle

    cache = {}

    def get(geo, L, p):
        if (geo, L, p) not in cache:
            h = mgamd.Hierarchy(ctx, geo, L, p, "HMG-local", coarse_solver="amg", max_brick=0)
            ref = ls_oracle.LocalSmoothing(geo, L, p, numbering_keys_global=h.active_dofs.keys(), numbering_keys_levels=[d.keys() for d in h.dofs])
            cache[(geo, L, p)] = (h, ref)
        return cache[(geo, L, p)]

    return get


@pytest.mark.parametrize("geo,L,p", CASES)
def test_level_operators_edge_matrices_and_transfers(mgamd, ctx, ls_hierarchies, geo, L, p):
    h, ref = ls_hierarchies(geo, L, p)
    rng = np.random.default_rng(41)
    assert len(h.operators) == len(ref.levels)
    for l, (op, Lv) in enumerate(zip(h.operators, ref.levels)):
        info = h.dofs[l].info
        first_edge = info.n_interior + info.n_tail
        assert h.dofs[l].n_dofs == Lv.n and info.n_edge == Lv.edge.sum() and info.n_hanging == 0
        assert Lv.edge[first_edge:first_edge + info.n_edge].all() and Lv.edge.sum() == info.n_edge  # [I | T | E | D]
        x = rng.standard_normal(Lv.n)
        src, dst = op.initialize_dof_vector().from_host(x), op.initialize_dof_vector()
        op.vmult(dst, src)  # edge DoFs: zero input, identity rows (ref:include/operator.h:152-183)
        assert rel_err(dst.to_host(), Lv.A @ x) < 1e-13
        assert np.array_equal(src.to_host(), x)
        diag = op.initialize_dof_vector()
        op.compute_inverse_diagonal(diag)
        assert rel_err(diag.to_host(), Lv.inv_diag) < 1e-13
        op.vmult_interface_up(dst, src)  # ref:include/operator.h:203-226
        t = Lv.A_edge_in @ x
        assert np.abs(dst.to_host() - t).max() <= 1e-13 * max(np.abs(t).max(), 1.0)
        op.vmult_interface_down(dst, src)  # ref:include/operator.h:191-201: the matrix of Multigrid's residual step
        assert rel_err(dst.to_host(), Lv.A_down @ x) < 1e-13
        if l > 0:
            xc, xf0 = rng.standard_normal(ref.levels[l - 1].n), rng.standard_normal(Lv.n)
            vc, vf = h.operators[l - 1].initialize_dof_vector().from_host(xc), op.initialize_dof_vector().from_host(xf0)
            h.transfers[l].prolongate_and_add(vf, vc)
            assert rel_err(vf.to_host(), xf0 + ref.P[l] @ xc) < 1e-13
            rf, dc0 = rng.standard_normal(Lv.n), rng.standard_normal(ref.levels[l - 1].n)
            vr, vd = op.initialize_dof_vector().from_host(rf), h.operators[l - 1].initialize_dof_vector().from_host(dc0)
            h.transfers[l].restrict_and_add(vd, vr)
            assert rel_err(vd.to_host(), dc0 + ref.P[l].T @ rf) < 1e-13
        assert h.smoothers[l].eigenvalue_estimates()[1] == pytest.approx(ref.sm[l].max_ev, rel=1e-9)


@pytest.mark.parametrize("geo,L,p", CASES)
def test_local_smoothing_vcycle_and_solve(mgamd, oracle, ctx, ls_hierarchies, geo, L, p):
    h, ref = ls_hierarchies(geo, L, p)
    n = ref.G.n
    assert h.n_dofs == n and h.mg.coarse_solver_used() == "direct"
    r = np.random.default_rng(42).standard_normal(n)
    r[ref.G.constrained] = 0.0
    vr, vz = mgamd.Vector(ctx, n).from_host(r), mgamd.Vector(ctx, n)
    h.mg.vmult(vz, vr)
    zref = ref.vcycle(r)
    assert rel_err(vz.to_host(), zref) < 1e-11
    u = np.random.default_rng(43).standard_normal(n)
    u[ref.G.constrained] = 0.0
    vu, vw = mgamd.Vector(ctx, n).from_host(u), mgamd.Vector(ctx, n)
    h.mg.vmult(vw, vu)
    assert abs(u @ vz.to_host() - r @ vw.to_host()) < 1e-10 * abs(u @ vz.to_host())  # a symmetric preconditioner
    xref, itref, hist = ref.solve(1e-4)
    b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    it, res = mgamd.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    assert it == itref and rel_err(x.to_host(), xref) < 1e-10
    # edge prolongation is a stage of the cycle (time_edge_pro, ref:multigrid_throughput.cc:1189-1190,1391)
    h.mg.stage_timing(True)
    h.mg.vmult(vz, vr)
    ms = h.mg.stage_times()
    h.mg.stage_timing(False)
    has_edges = any(d.info.n_edge for d in h.dofs)
    assert ms[5].sum() > 0 and ms[7].sum() > 0 and ms[8].sum() > 0
    if not has_edges:
        assert ms[5].sum() < 0.2 * ms.sum()  # empty stage: only the event pairs


def test_float_levels_local_smoothing(mgamd, ctx):
    """MGNumberType float (the reference's default) under the FP64 outer CG"""
    import ls_oracle

    ref = ls_oracle.LocalSmoothing("quadrant", 4, 2)
    h = mgamd.Hierarchy(ctx, "quadrant", 4, 2, "HMG-local", coarse_solver="amg", number_type=mgamd.F32, max_brick=0)
    b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    it, res = mgamd.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    assert abs(it - ref.solve(1e-4)[1]) <= 1


@pytest.mark.parametrize("geo,L,p", [("quadrant", 3, 4), ("annulus", 5, 2), ("quadrant", 4, 2)])
def test_hpmg_local(mgamd, ctx, geo, L, p):
    """`HPMG-local` (ref:multigrid_throughput.cc:1685-1695,1846-1860): p-multigrid on the active mesh over ONE local-smoothing
    V-cycle at the lowest degree, against the composed oracle"""
    import ls_oracle

    h = mgamd.Hierarchy(ctx, geo, L, p, "HPMG-local", coarse_solver="amg", max_brick=0)
    assert h.mg.coarse_solver_used() == "gmg_vcycle" and h.degrees[-1] == p and h.degrees[0] == 1
    ref = ls_oracle.PolynomialOverLocalSmoothing(geo, L, p, numbering_keys_p=[d.keys() for d in h.dofs],
                                                 numbering_keys_levels=[d.keys() for d in h.ls["dofs"]])
    n = ref.G.n
    r = np.random.default_rng(44).standard_normal(n)
    r[ref.G.constrained] = 0.0
    vr, vz = mgamd.Vector(ctx, n).from_host(r), mgamd.Vector(ctx, n)
    h.mg.vmult(vz, vr)
    assert rel_err(vz.to_host(), ref.vcycle(r)) < 1e-11
    xref, itref, hist = ref.solve(1e-4)
    b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    it, res = mgamd.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    assert it == itref and rel_err(x.to_host(), xref) < 1e-10
