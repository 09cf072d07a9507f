"""Per-level comparison of the sharded path (simulator) with the single-rank hierarchy (development tool)."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dealii_multigrid_amd as m

geo, L, p, n_ranks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ks = lambda keys: [tuple(int(v) for v in k) for k in keys]
ctx0 = m.Context(0)
os.environ["MGAMD_CHEB_KEY_INIT"] = "1"
h0 = m.Hierarchy(ctx0, geo, L, p, "HMG-global", coarse_solver="amg")
nl = len(h0.dofs)
rng = np.random.default_rng(0)
K0 = [ks(d.keys()) for d in h0.dofs]
ref = []
for l in range(nl):
    n = h0.dofs[l].n_dofs
    u = rng.standard_normal(n)
    first_c = h0.dofs[l].info.n_interior + h0.dofs[l].info.n_tail
    u[first_c:] = 0.0   # sharded solver vectors are zero on constrained DoFs
    vu, vA, vd = (h0.operators[l].initialize_dof_vector() for _ in range(3))
    vu.from_host(u); h0.operators[l].vmult(vA, vu); h0.operators[l].compute_inverse_diagonal(vd)
    r = dict(u=dict(zip(K0[l], u)), Au=dict(zip(K0[l], vA.to_host())), dinv=dict(zip(K0[l], vd.to_host())), eig=h0.smoothers[l].eigenvalue_estimates()[1])
    if l > 0:
        uc = np.array([ref[l-1]["u"][k] for k in K0[l-1]])
        vf = h0.operators[l].initialize_dof_vector(); vc = h0.operators[l-1].initialize_dof_vector().from_host(uc)
        h0.transfers[l].prolongate_and_add(vf, vc)
        r["P"] = dict(zip(K0[l], vf.to_host()))
        vr = h0.operators[l-1].initialize_dof_vector()
        h0.transfers[l].restrict_and_add(vr, vu)
        r["R"] = dict(zip(K0[l-1], vr.to_host()))
    ref.append(r)
group = m.SimGroup(n_ranks)
res = [None] * n_ranks
def main(rk):
    ctx = m.Context(0); comm = group.comm(rk)
    h = m.DistributedHierarchy(ctx, comm, geo, L, p, coarse_solver="amg", min_root_dofs=int(os.environ.get("MGAMD_MIN_ROOT_DOFS", "0")))
    out = []
    K = [ks(d.keys()) for d in h.dofs]
    for l in range(nl):
        op = h.operators[l]
        u = np.array([ref[l]["u"][k] for k in K[l]])
        vu, vA, vd = (op.initialize_dof_vector() for _ in range(3))
        vu.from_host(u); op.vmult(vA, vu); op.compute_inverse_diagonal(vd)
        def err(vals, table, keys, free_only=False):
            a = np.array(vals); b = np.array([table[k] for k in keys])
            return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
        e = dict(level=l, n=h.dofs[l].n_dofs, A=err(vA.to_host(), ref[l]["Au"], K[l]), dinv=err(vd.to_host(), ref[l]["dinv"], K[l]),
                 eig=(h.smoothers[l].eigenvalue_estimates()[1], ref[l]["eig"]), dist=l >= h.partition.root_level)
        if l > 0:
            uc = np.array([ref[l-1]["u"][k] for k in K[l-1]])
            vf = op.initialize_dof_vector(); vc = h.operators[l-1].initialize_dof_vector().from_host(uc)
            h.transfers[l].prolongate_and_add(vf, vc)
            e["P"] = err(vf.to_host(), ref[l]["P"], K[l])
            vr = h.operators[l-1].initialize_dof_vector()
            h.transfers[l].restrict_and_add(vr, vu)
            e["R"] = err(vr.to_host(), ref[l]["R"], K[l-1])
        out.append(e)
    res[rk] = out
th = [threading.Thread(target=main, args=(r,)) for r in range(n_ranks)]
[t.start() for t in th]; [t.join() for t in th]
for rk in range(n_ranks):
    for e in res[rk] or []:
        print(rk, e)

# ---- second stage: rhs, one V-cycle, Chebyshev smoother on the finest level
n0 = h0.n_dofs
b0 = h0.fine_operator.initialize_dof_vector(); h0.fine_operator.rhs(b0)
rr = rng.standard_normal(n0); rr[h0.dofs[-1].info.n_interior + h0.dofs[-1].info.n_tail:] = 0.0
vr0 = m.Vector(ctx0, n0).from_host(rr); vz0 = m.Vector(ctx0, n0)
h0.mg.vmult(vz0, vr0)
vs0 = m.Vector(ctx0, n0); h0.smoothers[-1].vmult(vs0, vr0)
vt0 = m.Vector(ctx0, n0).from_host(rr); h0.smoothers[-1].step(vt0, vr0)
print("STAGE2 single: dot(b,b)=%.17g dot(z,r)=%.17g dot(z,z)=%.17g" % (b0.dot(b0), vz0.dot(vr0), vz0.dot(vz0)))
x0s = h0.fine_operator.initialize_dof_vector()
print("STAGE2 single solve:", m.solve_cg(h0.fine_operator, h0.mg, x0s, b0, 1e-4))
R2 = dict(b=dict(zip(K0[-1], b0.to_host())), r=dict(zip(K0[-1], rr)), z=dict(zip(K0[-1], vz0.to_host())), s=dict(zip(K0[-1], vs0.to_host())),
          t=dict(zip(K0[-1], vt0.to_host())))
group2 = m.SimGroup(n_ranks)
res2 = [None] * n_ranks
def main2(rk):
    ctx = m.Context(0); comm = group2.comm(rk)
    h = m.DistributedHierarchy(ctx, comm, geo, L, p, coarse_solver="amg", min_root_dofs=int(os.environ.get("MGAMD_MIN_ROOT_DOFS", "0")))
    K = ks(h.dofs[-1].keys())
    def err(vals, table):
        a = np.array(vals); b = np.array([table[k] for k in K])
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)), int(np.abs(a - b).argmax())
    b = h.fine_operator.initialize_dof_vector(); h.fine_operator.rhs(b)
    r = m.Vector(ctx, h.n_local).from_host(np.array([R2["r"][k] for k in K])); z = m.Vector(ctx, h.n_local)
    h.mg.vmult(z, r)
    s = m.Vector(ctx, h.n_local); h.smoothers[-1].vmult(s, r)
    t = m.Vector(ctx, h.n_local).from_host(np.array([R2["r"][k] for k in K])); h.smoothers[-1].step(t, r)
    info = h.dofs[-1].info
    dots = (h.fine_operator.dot(b, b), h.fine_operator.dot(z, r), h.fine_operator.dot(z, z))
    xs = h.fine_operator.initialize_dof_vector()
    sol = m.solve_cg(h.fine_operator, h.mg, xs, b, 1e-4)
    ez = err(z.to_host(), R2["z"])
    res2[rk] = dict(rhs=err(b.to_host(), R2["b"]), cheb_vmult=err(s.to_host(), R2["s"]), cheb_step=err(t.to_host(), R2["t"]), vcycle=ez,
                    dots="%.17g %.17g %.17g" % dots, solve=sol, worst_key=K[ez[1]], I=info.n_interior, Town=info.n_tail_owned, T=info.n_tail)
th = [threading.Thread(target=main2, args=(r,)) for r in range(n_ranks)]
[t.start() for t in th]; [t.join() for t in th]
for rk in range(n_ranks):
    print("STAGE2", rk, res2[rk])
