"""
ORACLE (test infrastructure, NOT product code) -- numpy/scipy restatement of the
hot path of peterrum/dealii-multigrid's `multigrid_throughput.cc`:
matrix-free global-coarsening multigrid V-cycle for the 3D Laplace operator on
octree meshes of the cube [-1,1]^3.

PARITY UNPINNED: the reference's arithmetic lives in deal.II (v9.4 / master-2022),
which is not vendored in /root/reference, not installed here and not buildable
offline; the reference ships no tests, golden vectors or logs.
This oracle is therefore pinned only against mathematical known-answer tests
(tests/test_oracle_known_answers.py) and restates deal.II's published algorithms
at the reference's call sites:

  mesh generators ........ ref:include/grid_generator.h:3-140
  level operator ......... ref:include/operator.h:152-183 (vmult, identity on constrained rows)
  cell integral .......... ref:include/operator.h:461-472 (gradients in -> same gradients out)
  inverse diagonal ....... ref:include/operator.h:228-242 (1e-10 guard -> 1.0)
  right-hand side ........ ref:include/operator.h:362-447
  smoother parameters .... ref:multigrid_throughput.cc:312-315, 867-883
  coarse solver .......... ref:multigrid_throughput.cc:888-944
  V-cycle wiring ......... ref:multigrid_throughput.cc:1093-1133
  solve protocol ......... ref:multigrid_throughput.cc:1140-1147, 1238-1254
  level hierarchy ........ ref:multigrid_throughput.cc:1506-1604, 2219-2224

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It is deliberately a *different algorithm* from the product path:
it assembles sparse matrices  A = C^T K C + I_constrained  with an explicit
constraint matrix C (textbook formulation), whereas the product evaluates the
operator matrix-free with in-cell hanging-node interpolation.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from numpy.polynomial import legendre as npleg

LMAX = 15  # integer coordinate resolution: cell of level l has size 2^(LMAX-l) (matches csrc/octree.hpp)


# ----------------------------------------------------------------------------
# 1D finite element tables on the reference interval [0,1]
# ----------------------------------------------------------------------------
def gll_nodes(p: int) -> np.ndarray:
    """Gauss-Lobatto nodes of FE_Q(p) on [0,1] (deal.II: support points of FE_Q)."""
    if p == 1:
        return np.array([0.0, 1.0])
    # interior nodes: roots of P'_p
    c = np.zeros(p + 1)
    c[p] = 1.0
    dr = npleg.legroots(npleg.legder(c))
    x = np.concatenate(([-1.0], np.sort(dr), [1.0]))
    return 0.5 * (x + 1.0)


def gauss(n: int):
    x, w = npleg.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def lagrange_eval(nodes: np.ndarray, x: np.ndarray):
    """values V[i,a] = phi_a(x_i) and derivatives D[i,a] = phi_a'(x_i)."""
    x = np.atleast_1d(np.asarray(x, dtype=float))
    n = len(nodes)
    V = np.ones((len(x), n))
    D = np.zeros((len(x), n))
    for a in range(n):
        den = 1.0
        for b in range(n):
            if b != a:
                den *= nodes[a] - nodes[b]
        for i, xi in enumerate(x):
            val = 1.0
            for b in range(n):
                if b != a:
                    val *= xi - nodes[b]
            V[i, a] = val / den
            s = 0.0
            for c in range(n):
                if c == a:
                    continue
                t = 1.0
                for b in range(n):
                    if b != a and b != c:
                        t *= xi - nodes[b]
                s += t
            D[i, a] = s / den
    return V, D


class FE1D:
    def __init__(self, p: int):
        self.p = p
        self.nodes = gll_nodes(p)
        self.xq, self.wq = gauss(p + 1)  # QGauss(p+1), ref:multigrid_throughput.cc:1562
        self.S, self.G = lagrange_eval(self.nodes, self.xq)  # [q,a]
        # quadrature-evaluated 1D mass / stiffness (exact for these integrands)
        self.M = self.S.T @ (self.wq[:, None] * self.S)
        self.K = self.G.T @ (self.wq[:, None] * self.G)
        self.m = self.S.T @ self.wq  # int phi_a


# ----------------------------------------------------------------------------
# Octree meshes (ref:include/grid_generator.h)
# ----------------------------------------------------------------------------
def _center(cell):
    l, i, j, k = cell
    h = 2.0 / (1 << l)
    return np.array([-1.0 + (i + 0.5) * h, -1.0 + (j + 0.5) * h, -1.0 + (k + 0.5) * h])


def _children(cell):
    l, i, j, k = cell
    return [(l + 1, 2 * i + a, 2 * j + b, 2 * k + c) for c in (0, 1) for b in (0, 1) for a in (0, 1)]


def _find_leaf(leaves: set, l, i, j, k):
    """leaf covering region (l,i,j,k) at level <= l, or None (outside / finer)."""
    n = 1 << l
    if not (0 <= i < n and 0 <= j < n and 0 <= k < n):
        return None
    for ll in range(l, -1, -1):
        s = l - ll
        c = (ll, i >> s, j >> s, k >> s)
        if c in leaves:
            return c
    return None


def balance(leaves: set) -> set:
    """full (face+edge+corner) 2:1 balance by refinement, as p4est does for deal.II."""
    leaves = set(leaves)
    work = list(leaves)
    while work:
        cell = work.pop()
        if cell not in leaves:
            continue
        l, i, j, k = cell
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dx == dy == dz == 0:
                        continue
                    nb = _find_leaf(leaves, l, i + dx, j + dy, k + dz)
                    if nb is not None and nb[0] < l - 1:
                        leaves.remove(nb)
                        ch = _children(nb)
                        leaves.update(ch)
                        work.extend(ch)
                        work.append(cell)
    return leaves


def refine(leaves: set, flagged) -> set:
    leaves = set(leaves)
    for c in flagged:
        leaves.remove(c)
        leaves.update(_children(c))
    return balance(leaves)


def refine_global(leaves: set, n=1) -> set:
    for _ in range(n):
        leaves = refine(leaves, list(leaves))
    return leaves


def coarsen_global(leaves: set) -> set:
    """one step of create_geometric_coarsening_sequence (ref:multigrid_throughput.cc:2219-2224):
    coarsen every complete family once, then re-balance."""
    fam = {}
    for (l, i, j, k) in leaves:
        if l > 0:
            fam.setdefault((l - 1, i >> 1, j >> 1, k >> 1), []).append((l, i, j, k))
    out = set(leaves)
    for par, ch in fam.items():
        if len(ch) == 8:
            out.difference_update(ch)
            out.add(par)
    return balance(out)


def create_mesh(geometry: str, n_ref_global: int, n_ref_local: int = 0) -> set:
    root = {(0, 0, 0, 0)}
    if geometry == "hypercube":  # ref:multigrid_throughput.cc:2056-2060
        return refine_global(root, n_ref_global)
    if geometry == "quadrant":  # ref:include/grid_generator.h:34-65
        if n_ref_global == 0:
            return root
        m = refine_global(root, 1)
        for _ in range(1, n_ref_global):
            m = refine(m, [c for c in m if np.all(_center(c) <= 0.0)])
        return m
    if geometry == "quadrant_flexible":  # ref:include/grid_generator.h:69-92
        m = refine_global(root, n_ref_global)
        for _ in range(n_ref_local):
            m = refine(m, [c for c in m if np.all(_center(c) <= 0.0)])
        return m
    if geometry == "annulus":  # ref:include/grid_generator.h:96-140
        if n_ref_global == 0:
            return root
        m = refine_global(root, max(n_ref_global - 3, 0))
        if n_ref_global >= 1:
            m = refine(m, [c for c in m if np.linalg.norm(_center(c)) < 0.55])
        if n_ref_global >= 2:
            m = refine(m, [c for c in m if 0.3 <= np.linalg.norm(_center(c)) <= 0.43])
        if n_ref_global >= 3:
            m = refine(m, [c for c in m if 0.335 <= np.linalg.norm(_center(c)) <= 0.39])
        return m
    if geometry == "circle":  # ref:include/grid_generator.h:3-30
        m = refine_global(root, min(n_ref_global, 3))
        for _ in range(3, n_ref_global):
            fl = []
            for c in m:
                l, i, j, k = c
                h = 2.0 / (1 << l)
                hit = False
                for v in range(8):
                    pt = np.array([-1 + (i + (v & 1)) * h, -1 + (j + ((v >> 1) & 1)) * h, -1 + (k + (v >> 2)) * h])
                    if np.linalg.norm(pt) < 1.0 / (4.0 * np.pi):
                        hit = True
                if hit:
                    fl.append(c)
            m = refine(m, fl)
        return m
    raise ValueError("unknown geometry " + geometry)


def morton_key(cell):
    l, i, j, k = cell
    s = LMAX - l
    x, y, z = i << s, j << s, k << s
    code = 0
    for b in range(LMAX):
        code |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
    return (code, l)


def sorted_cells(leaves):
    return sorted(leaves, key=morton_key)


def coarsening_sequence(fine: set):
    """levels[0] = coarsest (1 cell) ... levels[-1] = fine."""
    seq = [set(fine)]
    while len(seq[-1]) > 1:
        seq.append(coarsen_global(seq[-1]))
    return seq[::-1]


def is_cell_constrained(leaves: set, cell) -> bool:
    """dealii::parallel::Helper::is_constrained (ref:multigrid_throughput.cc:231-267): a coarser
    face- or edge-neighbour exists."""
    l, i, j, k = cell
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if abs(dx) + abs(dy) + abs(dz) not in (1, 2):
                    continue
                nb = _find_leaf(leaves, l, i + dx, j + dy, k + dz)
                if nb is not None and nb[0] < l:
                    return True
    return False


# ----------------------------------------------------------------------------
# DoFs (continuous Q_p on the octree, hanging nodes kept as own DoFs like deal.II)
# ----------------------------------------------------------------------------
def node_key(cell, a, b, c, p):
    """geometric identity of the DoF at local node (a,b,c) of `cell`:
    (px,py,pz, dirmask, level); vertices carry level 0."""
    l, i, j, k = cell
    S = 1 << (LMAX - l)
    pos = ((i * p + a) * S, (j * p + b) * S, (k * p + c) * S)
    dm = (1 if a % p else 0) | (2 if b % p else 0) | (4 if c % p else 0)
    return (pos[0], pos[1], pos[2], dm, l if dm else 0)


class Level:
    """One multigrid level: mesh + Q_p space + constraints + assembled operator."""

    def __init__(self, leaves: set, p: int, numbering_keys=None):
        self.p = p
        self.fe = FE1D(p)
        self.leaves = set(leaves)
        self.cells = sorted_cells(leaves)
        n1 = p + 1
        # own-node keys per cell, lexicographic (c,b,a) with a fastest
        self.key_to_dof = {}
        self.cell_dofs = np.zeros((len(self.cells), n1 ** 3), dtype=np.int64)
        keys = []
        for ci, cell in enumerate(self.cells):
            t = 0
            for c in range(n1):
                for b in range(n1):
                    for a in range(n1):
                        key = node_key(cell, a, b, c, p)
                        d = self.key_to_dof.get(key)
                        if d is None:
                            d = len(keys)
                            self.key_to_dof[key] = d
                            keys.append(key)
                        self.cell_dofs[ci, t] = d
                        t += 1
        self.keys = keys
        self.n = len(keys)
        if numbering_keys is not None:
            # adopt an external numbering (array of keys, index = external dof number)
            ext = {tuple(int(v) for v in k): idx for idx, k in enumerate(numbering_keys)}
            assert len(ext) == self.n, (len(ext), self.n)
            perm = np.array([ext[k] for k in keys], dtype=np.int64)
            self.cell_dofs = perm[self.cell_dofs]
            newkeys = [None] * self.n
            for old, new in enumerate(perm):
                newkeys[new] = keys[old]
            self.keys = newkeys
            self.key_to_dof = {k: i for i, k in enumerate(newkeys)}
        self._constraints()
        self._assemble()

    # -- constraints: zero Dirichlet on the whole boundary + hanging nodes
    #    (ref:multigrid_throughput.cc:1585-1593, 2309-2312)
    def _constraints(self):
        p, n1 = self.p, self.p + 1
        top = p << LMAX
        self.dirichlet = np.zeros(self.n, dtype=bool)
        for d, key in enumerate(self.keys):
            if any(key[t] == 0 or key[t] == top for t in range(3)):
                self.dirichlet[d] = True
        rows, cols, vals = [], [], []
        self.hanging = np.zeros(self.n, dtype=bool)
        cell_index = {c: i for i, c in enumerate(self.cells)}
        nodes = self.fe.nodes
        for ci, cell in enumerate(self.cells):
            l, i, j, k = cell
            S = 1 << (LMAX - l)
            nbs = set()
            for dz in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        if dx == dy == dz == 0:
                            continue
                        nb = _find_leaf(self.leaves, l, i + dx, j + dy, k + dz)
                        if nb is not None and nb[0] < l:
                            assert nb[0] == l - 1
                            nbs.add(nb)
            for nb in nbs:
                nl, ni, nj, nk = nb
                NS = 1 << (LMAX - nl)
                lo = (ni * p * NS, nj * p * NS, nk * p * NS)
                hi = tuple(v + p * NS for v in lo)
                nbd = self.cell_dofs[cell_index[nb]]
                nbset = set(nbd.tolist())
                t = -1
                for c in range(n1):
                    for b in range(n1):
                        for a in range(n1):
                            t += 1
                            d = self.cell_dofs[ci, t]
                            if self.hanging[d] or d in nbset:
                                continue
                            pos = ((i * p + a) * S, (j * p + b) * S, (k * p + c) * S)
                            if not all(lo[q] <= pos[q] <= hi[q] for q in range(3)):
                                continue
                            # reference coordinates of the node inside the coarse neighbour
                            xi = [
                                ((i + nodes[a]) * S - ni * NS) / NS,
                                ((j + nodes[b]) * S - nj * NS) / NS,
                                ((k + nodes[c]) * S - nk * NS) / NS,
                            ]
                            V = [lagrange_eval(nodes, [x])[0][0] for x in xi]
                            self.hanging[d] = True
                            tt = -1
                            for cc in range(n1):
                                for bb in range(n1):
                                    for aa in range(n1):
                                        tt += 1
                                        w = V[0][aa] * V[1][bb] * V[2][cc]
                                        if abs(w) > 1e-14:
                                            rows.append(d)
                                            cols.append(nbd[tt])
                                            vals.append(w)
        hang_cols = set(cols)
        assert not any(self.hanging[c] for c in hang_cols), "chained hanging-node constraint"
        self.constrained = self.dirichlet | self.hanging
        free_or_dir = ~self.hanging
        Ch = sp.coo_matrix((vals, (rows, cols)), shape=(self.n, self.n)).tocsr() + sp.diags(free_or_dir.astype(float))
        self.Ch = Ch  # hanging-node constraints only (Dirichlet DoFs kept): nodal values = Ch @ dof values
        # Dirichlet (homogeneous) columns vanish
        self.C = (Ch @ sp.diags((~self.dirichlet).astype(float))).tocsr()

    def _assemble(self):
        fe, p = self.fe, self.p
        Kc = np.kron(np.kron(fe.M, fe.M), fe.K) + np.kron(np.kron(fe.M, fe.K), fe.M) + np.kron(np.kron(fe.K, fe.M), fe.M)
        mc = np.kron(np.kron(fe.m, fe.m), fe.m)
        nloc = (p + 1) ** 3
        rows = np.repeat(self.cell_dofs, nloc, axis=1).ravel()
        cols = np.tile(self.cell_dofs, (1, nloc)).ravel()
        hs = np.array([2.0 / (1 << c[0]) for c in self.cells])
        vals = (hs[:, None] * Kc.ravel()[None, :]).ravel()
        K = sp.coo_matrix((vals, (rows, cols)), shape=(self.n, self.n)).tocsr()
        self.Kraw = K
        A = (self.C.T @ K @ self.C).tocsr()
        self.A = (A + sp.diags(self.constrained.astype(float))).tocsr()
        b = np.zeros(self.n)
        np.add.at(b, self.cell_dofs.ravel(), ((hs ** 3)[:, None] * mc[None, :]).ravel())
        self.rhs_constant = self.C.T @ b  # f == 1, g == 0 (ref:multigrid_throughput.cc:2286-2291)
        self.rhs_constant[self.constrained] = 0.0
        d = self.A.diagonal().copy()
        self.inv_diag = np.where(np.abs(d) > 1e-10, 1.0 / d, 1.0)  # ref:include/operator.h:240-241

    def vmult(self, x):
        return self.A @ x

    # -- general data (SimulationType "Gaussian", ref:multigrid_throughput.cc:60-125,2294-2298)
    def node_positions(self):
        """support point of every node/DoF (own nodes), shape (n, 3)"""
        pos = np.zeros((self.n, 3))
        n1, nodes = self.p + 1, self.fe.nodes
        for ci, (l, i, j, k) in enumerate(self.cells):
            h = 2.0 / (1 << l)
            t = 0
            for c in range(n1):
                for b in range(n1):
                    for a in range(n1):
                        pos[self.cell_dofs[ci, t]] = (-1 + h * (i + nodes[a]), -1 + h * (j + nodes[b]), -1 + h * (k + nodes[c]))
                        t += 1
        return pos

    def rhs_function(self, f, g):
        """Operator::rhs (ref:include/operator.h:362-447): QGauss(p+1) load vector of f, minus the stiffness matrix without
        Dirichlet constraints applied to the boundary interpolant of g (hanging nodes interpolated); constrained rows 0."""
        fe, p, n1 = self.fe, self.p, self.p + 1
        S = fe.S.reshape(n1, n1)  # [q, a]
        F = np.zeros(self.n)
        for ci, (l, i, j, k) in enumerate(self.cells):
            h = 2.0 / (1 << l)
            X = -1 + h * (i + fe.xq)[None, None, :] + 0 * fe.xq[:, None, None] + 0 * fe.xq[None, :, None]
            Y = -1 + h * (j + fe.xq)[None, :, None] + 0 * X
            Z = -1 + h * (k + fe.xq)[:, None, None] + 0 * X
            W = h ** 3 * fe.wq[:, None, None] * fe.wq[None, :, None] * fe.wq[None, None, :]
            fq = f(X, Y, Z) * W  # [qz, qy, qx]
            loc = np.einsum("zyx,zc,yb,xa->cba", fq, S, S, S)
            np.add.at(F, self.cell_dofs[ci], loc.ravel())
        xg = np.zeros(self.n)
        bd = self.dirichlet & ~self.hanging
        pos = self.node_positions()
        xg[bd] = g(pos[bd, 0], pos[bd, 1], pos[bd, 2])
        b = self.Ch.T @ (F - self.Kraw @ (self.Ch @ xg))
        b[self.constrained] = 0.0
        return b

    def distribute(self, x, g):
        """AffineConstraints::distribute: Dirichlet values, then hanging nodes from their parents"""
        x = x.copy()
        bd = self.dirichlet & ~self.hanging
        pos = self.node_positions()
        x[bd] = g(pos[bd, 0], pos[bd, 1], pos[bd, 2])
        x[self.hanging] = 0.0
        return self.Ch @ x


def gaussian_solution(x, y, z, width=0.1, centre=(-0.5, -0.5, -0.5)):
    r2 = (x - centre[0]) ** 2 + (y - centre[1]) ** 2 + (z - centre[2]) ** 2
    return np.exp(-r2 / width ** 2) / (np.sqrt(2 * np.pi) * width) ** 3


def gaussian_rhs(x, y, z, width=0.1, centre=(-0.5, -0.5, -0.5)):
    r2 = (x - centre[0]) ** 2 + (y - centre[1]) ** 2 + (z - centre[2]) ** 2
    return (2 * 3 - 4 * r2 / width ** 2) / width ** 2 * np.exp(-r2 / width ** 2) / (np.sqrt(2 * np.pi) * width) ** 3


# ----------------------------------------------------------------------------
# Two-level transfer (deal.II MGTwoLevelTransfer)
# ----------------------------------------------------------------------------
def build_transfer(fine: Level, coarse: Level, allow_uncovered=False) -> sp.csr_matrix:
    """P (n_f x n_c): x_f += P x_c is prolongate_and_add, d_c += P^T r_f is restrict_and_add.
    allow_uncovered: coarse cells without a counterpart on the fine level are skipped (local smoothing: the active cells
    of the coarser level)."""
    pf, pc = fine.p, coarse.p
    nf1, nc1 = pf + 1, pc + 1
    fcell_index = {c: i for i, c in enumerate(fine.cells)}
    rows, cols, vals = [], [], []
    touch = np.zeros(fine.n)
    for ci, cc in enumerate(coarse.cells):
        cd = coarse.cell_dofs[ci]
        l, i, j, k = cc
        if cc in fine.leaves:
            fcells = [cc]
        else:
            fcells = _children(cc)
            if allow_uncovered and not any(f in fine.leaves for f in fcells):
                continue
            assert all(f in fine.leaves for f in fcells)
        pts = {}  # fine dof -> reference point in coarse cell
        for fc in fcells:
            fl, fi, fj, fk = fc
            fd = fine.cell_dofs[fcell_index[fc]]
            sc = 1 << (fl - l)
            t = -1
            for c in range(nf1):
                for b in range(nf1):
                    for a in range(nf1):
                        t += 1
                        pts[fd[t]] = (
                            (fi + fine.fe.nodes[a]) / sc - i,
                            (fj + fine.fe.nodes[b]) / sc - j,
                            (fk + fine.fe.nodes[c]) / sc - k,
                        )
        for d, xi in pts.items():
            touch[d] += 1
            V = [lagrange_eval(coarse.fe.nodes, [x])[0][0] for x in xi]
            t = -1
            for c in range(nc1):
                for b in range(nc1):
                    for a in range(nc1):
                        t += 1
                        w = V[0][a] * V[1][b] * V[2][c]
                        if abs(w) > 1e-14:
                            rows.append(d)
                            cols.append(cd[t])
                            vals.append(w)
    Praw = sp.coo_matrix((vals, (rows, cols)), shape=(fine.n, coarse.n)).tocsr()
    w = np.where(fine.constrained, 0.0, 1.0 / np.maximum(touch, 1))
    return (sp.diags(w) @ Praw @ coarse.C).tocsr()


# ----------------------------------------------------------------------------
# Chebyshev smoother (deal.II PreconditionChebyshev)
# ----------------------------------------------------------------------------
def lanczos_from_cg(alphas, betas):
    n = len(alphas)
    T = np.zeros((n, n))
    for j in range(n):
        T[j, j] = 1.0 / alphas[j] + (betas[j - 1] / alphas[j - 1] if j > 0 else 0.0)
        if j + 1 < n:
            T[j, j + 1] = T[j + 1, j] = np.sqrt(betas[j]) / alphas[j]
    return T


class Chebyshev:
    def __init__(self, A, inv_diag, degree=3, smoothing_range=20.0, eig_cg_n_iterations=20, number=np.float64, start=None):
        """start: start vector of the eigenvalue estimate; default deal.II's (i mod 11) - mean"""
        self.A, self.dinv, self.k = A, inv_diag, degree
        n = A.shape[0]
        if start is None:
            v = (np.arange(n) % 11).astype(number)
            v = v - v.mean()
        else:
            v = np.array(start, dtype=number)
        # PCG with D^-1, x0 = 0, at most eig_cg_n_iterations iterations, tol 1e-10 (IterationNumberControl)
        x = np.zeros(n)
        r = v.copy()
        alphas, betas = [], []
        res0 = np.linalg.norm(r)
        if res0 > 0:
            z = self.dinv * r
            d = z.copy()
            rz = r @ z
            for it in range(eig_cg_n_iterations):
                Ad = A @ d
                alpha = rz / (d @ Ad)
                x += alpha * d
                r -= alpha * Ad
                alphas.append(alpha)
                res = np.linalg.norm(r)
                if res <= 1e-10:
                    break
                z = self.dinv * r
                rz_new = r @ z
                beta = rz_new / rz
                betas.append(beta)
                rz = rz_new
                d = z + beta * d
        if alphas:
            T = lanczos_from_cg(alphas, betas[: len(alphas) - 1] + [0.0])
            ev = np.linalg.eigvalsh(T)
            self.min_ev, self.max_ev_raw = ev[0], ev[-1]
        else:
            self.min_ev = self.max_ev_raw = 1.0
        self.max_ev = 1.2 * self.max_ev_raw
        alpha = self.max_ev / smoothing_range if smoothing_range > 1.0 else min(0.9 * self.max_ev, self.min_ev)
        self.delta = 0.5 * (self.max_ev - alpha)
        self.theta = 0.5 * (self.max_ev + alpha)

    def _iterate(self, x, xold, b):
        if self.k < 2 or abs(self.delta) < 1e-40:
            return x
        rhok, sigma = self.delta / self.theta, self.theta / self.delta
        for _ in range(self.k - 1):
            rhokp = 1.0 / (2.0 * sigma - rhok)
            f1, f2 = rhokp * rhok, 2.0 * rhokp / self.delta
            rhok = rhokp
            xn = x + f1 * (x - xold) + f2 * self.dinv * (b - self.A @ x)
            xold, x = x, xn
        return x

    def vmult(self, b):
        """zero initial guess (pre-smoothing): k-1 operator applications."""
        x1 = (1.0 / self.theta) * self.dinv * b
        return self._iterate(x1, np.zeros_like(b), b)

    def step(self, x0, b):
        """general initial guess (post-smoothing): k operator applications."""
        x1 = x0 + (1.0 / self.theta) * self.dinv * (b - self.A @ x0)
        return self._iterate(x1, x0, b)


# ----------------------------------------------------------------------------
# PCG (deal.II SolverCG + ReductionControl)
# ----------------------------------------------------------------------------
def pcg(A, b, precond, reltol=1e-4, abstol=1e-20, maxiter=10000, x0=None):
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x if x0 is not None else b.copy()
    res0 = np.linalg.norm(r)
    hist = [res0]
    if res0 <= abstol:
        return x, 0, hist
    z = precond(r)
    d = z.copy()
    rz = r @ z
    it = 0
    while it < maxiter:
        it += 1
        Ad = A @ d
        alpha = rz / (d @ Ad)
        x += alpha * d
        r -= alpha * Ad
        res = np.linalg.norm(r)
        hist.append(res)
        if res < reltol * res0 or res <= abstol:
            break
        z = precond(r)
        rz_new = r @ z
        beta = rz_new / rz
        rz = rz_new
        d = z + beta * d
    return x, it, hist


def key_hash_start_vector(level):
    """Numbering-independent start vector of the eigenvalue estimate, as the product's sharded runs define it (there is no
    global DoF index across ranks): a 64-bit mix of the packed geometric DoF key, mod 11, on unconstrained DoFs, minus
    its mean over them; zero on constrained DoFs."""
    M = (1 << 64) - 1

    def mix(x):
        x ^= x >> 33
        x = (x * 0xff51afd7ed558ccd) & M
        x ^= x >> 33
        x = (x * 0xc4ceb9fe1a85ec53) & M
        x ^= x >> 33
        return x

    v = np.zeros(level.n)
    for i, (px, py, pz, dm, lev) in enumerate(level.keys):
        if not level.constrained[i]:
            v[i] = mix(((px << 43) | (py << 25) | (pz << 7) | (dm << 4) | (lev if dm else 0)) & M) % 11
    free = ~level.constrained
    if free.any():
        v[free] -= v[free].mean()
    return v


# ----------------------------------------------------------------------------
# Multigrid V-cycle (deal.II Multigrid::level_v_step / PreconditionMG)
# ----------------------------------------------------------------------------
class Multigrid:
    def __init__(self, levels, transfers, smoother_degree=3, smoothing_range=20.0, eig_cg_n_iterations=20,
                 coarse="direct", coarse_reltol=1e-4, start_vectors=None):
        """start_vectors: per level the start vector of the smoother's eigenvalue estimate (default: deal.II's index-based
        one); key_hash_start_vector(level) gives the numbering-independent vector the product's sharded runs use"""
        self.levels, self.P = levels, transfers  # P[l]: level l-1 -> l  (P[0] unused)
        sv = start_vectors or [None] * len(levels)
        self.sm = [Chebyshev(L.A, L.inv_diag, smoother_degree, smoothing_range, eig_cg_n_iterations, start=sv[l]) for l, L in enumerate(levels)]
        self.coarse, self.coarse_reltol = coarse, coarse_reltol
        if coarse == "direct":
            self.A0 = spla.splu(sp.csc_matrix(levels[0].A))

    def coarse_solve(self, d):
        L0 = self.levels[0]
        if callable(self.coarse):  # e.g. a local-smoothing V-cycle on level 0's space (HPMG-local)
            return self.coarse(d)
        if self.coarse == "direct":
            return self.A0.solve(d)
        if self.coarse == "cg":  # ref:multigrid_throughput.cc:911-921
            return pcg(L0.A, d, lambda r: r, self.coarse_reltol, 1e-20, 10000)[0]
        if self.coarse == "cg_with_chebyshev":  # ref:multigrid_throughput.cc:922-944
            return pcg(L0.A, d, self.sm[0].vmult, self.coarse_reltol, 1e-20, 10000)[0]
        raise ValueError(self.coarse)

    def vcycle(self, r):
        nl = len(self.levels)
        defect = [np.zeros(L.n) for L in self.levels]
        sol = [None] * nl
        defect[-1] = r.copy()
        for l in range(nl - 1, 0, -1):
            sol[l] = self.sm[l].vmult(defect[l])
            t = defect[l] - self.levels[l].A @ sol[l]
            defect[l - 1] += self.P[l].T @ t
        sol[0] = self.coarse_solve(defect[0])
        for l in range(1, nl):
            sol[l] = sol[l] + self.P[l] @ sol[l - 1]
            sol[l] = self.sm[l].step(sol[l], defect[l])
        return sol[-1]


def build_hierarchy(geometry, n_ref_global, degree, mg_type="HMG-global", n_ref_local=0, numbering_keys=None):
    """levels (coarse -> fine) and transfers for `HMG-global` / `PMG`
    (ref:multigrid_throughput.cc:1506-1604).  numbering_keys: optional list (per level) of external
    DoF numberings as key arrays."""
    fine = create_mesh(geometry, n_ref_global, n_ref_local)
    if mg_type == "HMG-global":
        meshes = coarsening_sequence(fine)
        degs = [degree] * len(meshes)
    elif mg_type == "PMG":
        seq = [degree]
        while seq[-1] > 1:  # bisect: 4 -> 2 -> 1
            seq.append(max(seq[-1] // 2, 1))
        degs = seq[::-1]
        meshes = [fine] * len(degs)
    elif mg_type == "HPMG":  # ref:multigrid_throughput.cc:1518-1519,1551-1553,1569-1571
        seq = [degree]
        while seq[-1] > 1:
            seq.append(max(seq[-1] // 2, 1))
        pseq = seq[::-1]
        hmeshes = coarsening_sequence(fine)
        meshes = hmeshes + [fine] * (len(pseq) - 1)
        degs = [pseq[0]] * len(hmeshes) + pseq[1:]
    else:
        raise ValueError(mg_type)
    levels = []
    for li, (m, p) in enumerate(zip(meshes, degs)):
        nk = numbering_keys[li] if numbering_keys is not None else None
        levels.append(Level(m, p, nk))
    P = [None] + [build_transfer(levels[l], levels[l - 1]) for l in range(1, len(levels))]
    return levels, P


def solve(geometry, n_ref_global, degree, mg_type="HMG-global", smoother_degree=3, reltol=1e-4,
          coarse="direct", numbering_keys=None):
    levels, P = build_hierarchy(geometry, n_ref_global, degree, mg_type, numbering_keys=numbering_keys)
    mg = Multigrid(levels, P, smoother_degree, coarse=coarse)
    L = levels[-1]
    x, it, hist = pcg(L.A, L.rhs_constant, mg.vcycle, reltol)
    return dict(levels=levels, P=P, mg=mg, x=x, n_iterations=it, history=hist)
