"""Development check: CG iteration counts of geo:L:p[:mg_type] cases under the kernel-path switches (each combination in a child
process, since the switches are read when the library is loaded).
  python3 tools/solve_check.py annulus:6:4 annulus:7:4"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os
sys.path.insert(0, %r)
import dealii_multigrid_amd as m
geo, L, p, mg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ctx = m.Context(0)
h = m.Hierarchy(ctx, geo, L, p, mg, coarse_solver="direct" if mg == "HMG-global" else "amg")
b, z = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
h.fine_operator.rhs(b)
it, res = m.solve_cg(h.fine_operator, h.mg, z, b, 1e-4, maxiter=50)
fused = [t.n_fused_bricks() for t in h.transfers if t is not None]
print(f"  iterations {it:3d} residual {res:.3e} n_dofs {h.n_dofs} fused bricks per transfer {fused}", flush=True)
""" % ROOT

SWITCHES = [{}, {"MGAMD_NO_FUSED_TRANSFER": "1"}, {"MGAMD_NO_CELL_WAVES": "1"}, {"MGAMD_NO_PERSISTENT": "1"}]
for case in sys.argv[1:]:
    parts = case.split(":")
    geo, L, p = parts[0], parts[1], parts[2]
    mg = parts[3] if len(parts) > 3 else "HMG-global"
    for sw in SWITCHES:
        print(f"{geo} L={L} p={p} {mg} {sw}", flush=True)
        env = dict(os.environ)
        env.update(sw)
        subprocess.run([sys.executable, "-c", CHILD, geo, L, p, mg], env=env, timeout=600)
