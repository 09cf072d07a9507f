"""ORACLE (test infrastructure, never imported by the product): local-smoothing multigrid -- the reference's `HMG-local`
(ref:multigrid_throughput.cc:1670-1873, edge data of ref:include/operator.h:49-120,152-226) -- restated from the published
deal.II algorithm in textbook form: assembled level matrices on the refinement levels of the octree, explicit
refinement-edge index sets, explicit edge matrix, explicit transfer matrices and copy-index maps.  PARITY UNPINNED
against deal.II for the same reason as mgoracle.py (the reference holds no recorded outputs and deal.II cannot be built
here); it is independent of the product's matrix-free formulation.

deal.II pieces restated:
  * level l = ALL cells of refinement level l, active or not (DoFHandler::distribute_mg_dofs); no hanging nodes on a level
  * MGConstrainedDoFs: zero Dirichlet boundary + refinement edge (MGTools::extract_inner_interface_dofs: DoFs on faces
    between a level cell and a coarser cell)
  * level operator: edge DoFs are zeroed in the source and become identity rows (operator.h:152-183)
  * edge matrices (Janssen & Kanschat 2011; deal.II Multigrid::level_v_step): the level operator only couples the
    interior of the refined region; the coupling across the refinement edge enters twice,
      up:   after the coarse-grid correction  defect_l -= A_l^{no edge constraints} (x_l restricted to the edge DoFs)
            (Multigrid::set_edge_in_matrix, ref:multigrid_throughput.cc:1105,1130, + operator.h:203-226
            vmult_interface_up), and
      down: the matrix of the RESIDUAL step is not Operator::vmult: the reference builds Multigrid's mg::Matrix from
            MatrixFreeOperators::MGInterfaceOperator<LevelMatrixType> (ref:multigrid_throughput.cc:857-862), whose vmult
            forwards to Operator::vmult_interface_down (ref:include/operator.h:191-201) -- the plain cell loop with identity
            on the Dirichlet rows only, the refinement-edge DoFs being ordinary rows AND columns (`A_down` below).  Hence the
            residual that is restricted carries the edge rows, t = d - A_down x.  The zero-start pre-smoother (level operator
            with identity edge rows, zero defect on the edge DoFs) leaves x_E = 0, so t_E = d_E - A_{E,I} x_I.
    (Rounds 1-2 recorded this as an ambiguity because Operator::vmult has identity edge rows; lines 857-862 settle it.  The
    variant WITHOUT the edge rows is a non-symmetric preconditioner under which CG stalls in this oracle.)
  * MGTransferMatrixFree: P_l per refined cell of level l-1, weights 1/multiplicity, boundary DoFs zero
  * copy_to_mg / copy_from_mg (MGLevelGlobalTransfer, skip_interface_dofs): a DoF of the active mesh lives on the level of
    its active cell unless it sits on that level's refinement edge
"""
import itertools

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import mgoracle as o


def level_meshes(leaves):
    L = max(c[0] for c in leaves)
    S = [set() for _ in range(L + 1)]
    for (l, i, j, k) in leaves:
        for ll in range(l + 1):
            s = l - ll
            S[ll].add((ll, i >> s, j >> s, k >> s))
    return S


class LSLevel:
    def __init__(self, cells, p, numbering_keys=None):
        lv = o.Level(cells, p, numbering_keys)
        assert not lv.hanging.any()
        self.base, self.n, self.p, self.keys, self.key_to_dof = lv, lv.n, p, lv.keys, lv.key_to_dof
        self.cells, self.cell_dofs, self.dirichlet = lv.cells, lv.cell_dofs, lv.dirichlet
        n1 = p + 1
        edge = np.zeros(lv.n, dtype=bool)
        cellset = set(cells)
        for ci, (l, i, j, k) in enumerate(lv.cells):
            N = 1 << l
            loc = lv.cell_dofs[ci].reshape(n1, n1, n1)  # [c, b, a], a (x) fastest
            for d, side in itertools.product(range(3), (0, 1)):
                nb = [i, j, k]
                nb[d] += 1 if side else -1
                if nb[d] < 0 or nb[d] >= N or (l, nb[0], nb[1], nb[2]) in cellset:
                    continue  # domain boundary, or a neighbour on this level
                edge[np.take(loc, side * p, axis=2 - d).ravel()] = True
        edge &= ~lv.dirichlet
        self.edge = edge
        self.constrained = lv.dirichlet | edge
        K = lv.Kraw
        free = sp.diags((~self.constrained).astype(float))
        self.A = (free @ K @ free + sp.diags(self.constrained.astype(float))).tocsr()
        self.A_edge_in = (sp.diags((~lv.dirichlet).astype(float)) @ K @ sp.diags(edge.astype(float))).tocsr()
        # Operator::vmult_interface_down (ref:include/operator.h:191-201): only the Dirichlet DoFs are constrained
        nd = sp.diags((~lv.dirichlet).astype(float))
        self.A_down = (nd @ K @ nd + sp.diags(lv.dirichlet.astype(float))).tocsr()
        d = self.A.diagonal()
        self.inv_diag = np.where(np.abs(d) > 1e-10, 1.0 / d, 1.0)


class LocalSmoothing:
    def __init__(self, geometry, n_ref_global, degree, smoother_degree=3, numbering_keys_global=None, numbering_keys_levels=None):
        # (geometry may be a set of leaves: a caller-built, 2:1-balanced octree)
        leaves = set(geometry) if isinstance(geometry, (set, frozenset)) else o.create_mesh(geometry, n_ref_global)
        self.G = o.Level(leaves, degree, numbering_keys_global)  # the outer (active mesh) problem
        S = level_meshes(leaves)
        nk = numbering_keys_levels or [None] * len(S)
        self.levels = [LSLevel(S[l], degree, nk[l]) for l in range(len(S))]
        self.P = [None] + [o.build_transfer(self.levels[l].base, self.levels[l - 1].base, allow_uncovered=True) for l in range(1, len(S))]
        self.sm = [o.Chebyshev(L.A, L.inv_diag, smoother_degree, 20.0, 20) for L in self.levels]
        self.A0 = spla.splu(sp.csc_matrix(self.levels[0].A))
        # copy indices
        self.copy = []
        p, n1 = degree, degree + 1
        for l, Lv in enumerate(self.levels):
            g, lidx, seen = [], [], set()
            for cell in o.sorted_cells([c for c in leaves if c[0] == l]):
                for c in range(n1):
                    for b in range(n1):
                        for a in range(n1):
                            key = o.node_key(cell, a, b, c, p)
                            ld = Lv.key_to_dof[key]
                            if Lv.edge[ld] or Lv.dirichlet[ld] or ld in seen:
                                continue
                            gd = self.G.key_to_dof[key]
                            assert not self.G.constrained[gd]
                            seen.add(ld)
                            g.append(gd)
                            lidx.append(ld)
            self.copy.append((np.array(g, dtype=np.int64), np.array(lidx, dtype=np.int64)))
        cnt = np.zeros(self.G.n, dtype=int)
        for g, _ in self.copy:
            cnt[g] += 1
        assert (cnt[~self.G.constrained] == 1).all() and (cnt[self.G.constrained] == 0).all()  # every free DoF on exactly one level

    def vcycle(self, r):
        nl = len(self.levels)
        defect = [np.zeros(L.n) for L in self.levels]
        for l, (g, li) in enumerate(self.copy):
            defect[l][li] = r[g]
        sol = [None] * nl

        def step(l):
            if l == 0:
                sol[0] = self.A0.solve(defect[0])
                return
            Lv = self.levels[l]
            x = self.sm[l].vmult(defect[l])
            t = defect[l] - Lv.A_down @ x  # mg::Matrix over MGInterfaceOperator: vmult_interface_down
            defect[l - 1] += self.P[l].T @ t
            step(l - 1)
            x = x + self.P[l] @ sol[l - 1]
            if Lv.edge.any():
                defect[l] -= Lv.A_edge_in @ x
            sol[l] = self.sm[l].step(x, defect[l])

        step(nl - 1)
        z = np.zeros(self.G.n)
        for l, (g, li) in enumerate(self.copy):
            z[g] = sol[l][li]
        return z

    def solve(self, reltol=1e-4):
        return o.pcg(self.G.A, self.G.rhs_constant, self.vcycle, reltol)


class PolynomialOverLocalSmoothing:
    """`HPMG-local` (ref:multigrid_throughput.cc:1685-1695,1846-1860): p-multigrid (bisection sequence) on the active mesh whose
    coarse problem -- the lowest degree -- is handed to ONE local-smoothing V-cycle."""

    def __init__(self, geometry, n_ref_global, degree, smoother_degree=3, numbering_keys_p=None, numbering_keys_levels=None):
        leaves = o.create_mesh(geometry, n_ref_global)
        pseq = [degree]
        while pseq[-1] > 1:
            pseq.append(max(pseq[-1] // 2, 1))
        pseq = pseq[::-1]
        nk = numbering_keys_p or [None] * len(pseq)
        self.ls = LocalSmoothing(geometry, n_ref_global, pseq[0], smoother_degree, numbering_keys_global=nk[0],
                                 numbering_keys_levels=numbering_keys_levels)
        self.levels = [self.ls.G] + [o.Level(leaves, p, nk[i + 1]) for i, p in enumerate(pseq[1:])]
        self.P = [None] + [o.build_transfer(self.levels[l], self.levels[l - 1]) for l in range(1, len(self.levels))]
        self.mg = o.Multigrid(self.levels, self.P, smoother_degree, coarse=self.ls.vcycle)
        self.G = self.levels[-1]

    def vcycle(self, r):
        return self.mg.vcycle(r)

    def solve(self, reltol=1e-4):
        return o.pcg(self.G.A, self.G.rhs_constant, self.vcycle, reltol)
