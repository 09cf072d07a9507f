"""Development aid: which DoFs of the inverse diagonal differ from the oracle (by geometric key)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np
import dealii_multigrid_amd as m
import mgoracle as o
geo, L, p = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ctx = m.Context(0)
t = m.Triangulation(geo, L); d = m.DoFs(t, p, 0); op = m.Operator(ctx, d)
keys = d.keys()
lv = o.Level(o.create_mesh(geo, L), p, numbering_keys=keys)
diag = op.initialize_dof_vector(); op.compute_inverse_diagonal(diag)
g = diag.to_host(); r = lv.inv_diag
bad = np.where(np.abs(g - r) > 1e-12 * np.abs(r))[0]
print("groups", d.groups(), "n_dofs", d.n_dofs, "bad", len(bad), "I/T", d.info.n_interior, d.info.n_tail)
for i in bad[:40]:
    print(i, keys[i].tolist(), "gpu 1/d", 1 / g[i], "oracle 1/d", 1 / r[i], "ratio", r[i] / g[i])
