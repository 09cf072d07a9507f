"""Helper of tests/test_gpu_kernel_variants.py (run as a script in a child process so that library-level switches read from
the environment at load time take effect): builds a hierarchy, applies the operator, one Chebyshev step() and one V-cycle to
fixed inputs and writes the three result vectors.
  python tests/_vcycle_dump.py <geometry> <n_ref> <degree> <out.npz>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dealii_multigrid_amd as m  # noqa: E402

geo, L, p, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ctx = m.Context(0)
h = m.Hierarchy(ctx, geo, L, p, "HMG-global", coarse_solver="amg")
op = h.fine_operator
n = op.m()
rng = np.random.default_rng(11)
x, b = rng.standard_normal(n), rng.standard_normal(n)
keys = np.asarray(h.dofs[-1].keys(), dtype=np.int64)
if len(sys.argv) > 5 and sys.argv[5] == "keys":
    # inputs as functions of the geometric DoF key: the same global vectors whatever the slot decomposition numbers the DoFs like
    hk = (keys[:, 0] * 73856093) ^ (keys[:, 1] * 19349663) ^ (keys[:, 2] * 83492791) ^ (keys[:, 3] * 2654435761) ^ (keys[:, 4] * 97)
    x = np.sin(hk.astype(np.float64) * 1e-3) + 0.25 * np.cos(hk.astype(np.float64) * 7e-5)
    b = np.cos(hk.astype(np.float64) * 3e-3) - 0.5 * np.sin(hk.astype(np.float64) * 9e-5)
    first_c = h.dofs[-1].info.n_interior + h.dofs[-1].info.n_tail
    b[first_c:] = 0.0
vx, vb, vy, vz = (op.initialize_dof_vector() for _ in range(4))
vx.from_host(x), vb.from_host(b)
op.vmult(vy, vx)
ax = vy.to_host()
ch = m.PreconditionChebyshev(op, 3, 20.0, 2)
vy.from_host(x)
ch.step(vy, vb)
st = vy.to_host()
h.mg.vmult(vz, vb)
np.savez(out, ax=ax, step=st, vcycle=vz.to_host(), groups=np.array(h.dofs[-1].groups()), keys=keys)
