"""Sharded path on ONE GPU: n ranks run as host threads over the in-process simulator communicator (device-to-device
copies instead of RCCL send/recv); everything else -- partition, local numbering, halo plans, pack/combine kernels,
distributed transfer, all-reduce onto the replicated levels, global dots -- is the production code.  Results are
compared with the single-rank solve (CG iterates within the FP64 tolerance, iteration counts equal)."""
import os
import threading

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def run_ranks(n_ranks, fn):
    out, err = [None] * n_ranks, [None] * n_ranks

    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:  # noqa
            err[r] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    for e in err:
        if e is not None:
            raise e
    return out


def keyset(keys):
    return [tuple(int(v) for v in k) for k in keys]


@pytest.fixture(autouse=True)
def key_based_chebyshev_start_vector():
    # the sharded path has no global DoF index: its Chebyshev start vector hashes the geometric DoF key; use the same
    # start vector for the single-rank reference (and the replicated levels) so that the comparison is arithmetic-exact
    os.environ["MGAMD_CHEB_KEY_INIT"] = "1"
    yield
    del os.environ["MGAMD_CHEB_KEY_INIT"]


@pytest.mark.parametrize("geo,L,p,n_ranks", [("quadrant", 5, 1, 2), ("quadrant", 4, 4, 2), ("quadrant", 5, 2, 3), ("annulus", 6, 1, 4),
                                             ("hypercube", 4, 2, 2), ("quadrant", 5, 4, 4)])
def test_sharded_solve_matches_single_rank(mgamd, geo, L, p, n_ranks):
    # single-rank reference on the same GPU, with the numbering-independent Chebyshev start vector of the sharded path
    ctx0 = mgamd.Context(0)
    h0 = mgamd.Hierarchy(ctx0, geo, L, p, "HMG-global", coarse_solver="amg", max_brick=0)
    b0, x0 = h0.fine_operator.initialize_dof_vector(), h0.fine_operator.initialize_dof_vector()
    h0.fine_operator.rhs(b0)
    it0, res0 = mgamd.solve_cg(h0.fine_operator, h0.mg, x0, b0, 1e-4)
    ref = dict(zip(keyset(h0.dofs[-1].keys()), x0.to_host()))
    rng = np.random.default_rng(31)
    u0 = rng.standard_normal(h0.n_dofs)
    vu, vAu = h0.fine_operator.initialize_dof_vector().from_host(u0), h0.fine_operator.initialize_dof_vector()
    h0.fine_operator.vmult(vAu, vu)
    uref, Auref = dict(zip(keyset(h0.dofs[-1].keys()), u0)), dict(zip(keyset(h0.dofs[-1].keys()), vAu.to_host()))

    group = mgamd.SimGroup(n_ranks)

    def rank_main(r):
        ctx = mgamd.Context(0)
        comm = group.comm(r)
        h = mgamd.DistributedHierarchy(ctx, comm, geo, L, p, coarse_solver="amg", max_brick=0, min_root_dofs=0)
        keys = keyset(h.dofs[-1].keys())
        # operator application on consistent copies of a global vector
        u = h.fine_operator.initialize_dof_vector().from_host(np.array([uref[k] for k in keys]))
        Au = h.fine_operator.initialize_dof_vector()
        h.fine_operator.vmult(Au, u)
        err_A = max(abs(a - Auref[k]) for a, k in zip(Au.to_host(), keys)) / max(abs(v) for v in Auref.values())
        b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
        h.fine_operator.rhs(b)
        it, res = mgamd.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
        info = h.dofs[-1].info
        return dict(it=it, res=res, keys=keys, x=x.to_host(), n_dofs=h.n_dofs, err_A=err_A, root=h.partition.root_level, peers=info.n_peers,
                    send=info.n_halo_send, n_local=h.n_local)

    out = run_ranks(n_ranks, rank_main)
    assert all(o["n_dofs"] == h0.n_dofs for o in out)  # every DoF owned exactly once
    assert all(o["peers"] >= 1 and o["send"] > 0 for o in out)
    assert sum(o["n_local"] for o in out) > h0.n_dofs  # shared copies exist
    for o in out:
        assert o["err_A"] < 1e-13
        assert o["it"] == it0
        assert o["res"] == pytest.approx(res0, rel=1e-7)
        xr = np.array([ref[k] for k in o["keys"]])
        assert rel_err(o["x"], xr) < 1e-10
    # copies of shared DoFs agree across ranks
    seen = {}
    for o in out:
        for k, v in zip(o["keys"], o["x"]):
            if k in seen:
                assert abs(seen[k] - v) <= 1e-12 * max(1.0, abs(v))
            seen[k] = v
    assert len(seen) == h0.n_dofs
