"""ORACLE (test infrastructure, never imported by the product): index tables for the C++ CPU oracle built from THIS package's
own mesh generator and DoF numbering (mgoracle.create_mesh / node_key), so that the checker used beyond numpy's reach is not
fed exclusively by the product's host setup (round-2 verdict, weak point 2).  PARITY UNPINNED like the rest of oracle/.

What deal.II's matrix-free path stores per cell (ref:include/operator.h:24-47 MatrixFree::reinit, FEEvaluation::read_dof_values
on cells with hanging faces/edges) restated in plain Python loops:
  * every own node of every cell is a DoF (hanging-node DoFs exist, as in DoFHandler::distribute_dofs); numbering here is
    [free | Dirichlet | hanging] in first-touch order of the Morton-sorted cells -- unrelated to the product's [I|T|E|D|H]
  * a cell whose face (edge) lies in a coarser neighbour gathers the PARENT's DoFs at the same local node index on that face
    (edge) and interpolates in-cell with the child's half of the 1D embedding; the constraint configuration is
    (child position, hanging faces, hanging edges) = bits (0-2, 3-5, 6-8)
  * two-level transfer (ref:multigrid_throughput.cc:1600-1604): per coarse cell one patch -- identity (kind 0), its 8 children
    (kind 1), or the same cell at a higher degree (kind 2); every unconstrained fine DoF is given to exactly ONE patch

Sizes: pure-Python loops over cells x nodes; meant for <= ~3e5 DoFs.
"""
import numpy as np

import mgoracle as o

INVALID = 0xFFFFFFFF


def _cell_mask(leaves, cell):
    l, i, j, k = cell
    if l == 0:
        return 0
    cp = (i & 1, j & 1, k & 1)
    idx = (i, j, k)
    mask = cp[0] | (cp[1] << 1) | (cp[2] << 2)
    face = [False] * 3
    for d in range(3):
        n = list(idx)
        n[d] += 1 if cp[d] else -1
        nb = o._find_leaf(leaves, l, *n)
        face[d] = nb is not None and nb[0] < l
        if face[d]:
            mask |= 1 << (3 + d)
    for d in range(3):
        e, f = (d + 1) % 3, (d + 2) % 3
        n = list(idx)
        n[e] += 1 if cp[e] else -1
        n[f] += 1 if cp[f] else -1
        nb = o._find_leaf(leaves, l, *n)
        if face[e] or face[f] or (nb is not None and nb[0] < l):
            mask |= 1 << (6 + d)
    return mask


def _on_constrained_entity(mask, p, a):
    """(node lies on a hanging face/edge of the cell, node is a corner of the parent)"""
    cp = (mask & 1, (mask >> 1) & 1, (mask >> 2) & 1)
    on = [a[d] == cp[d] * p for d in range(3)]
    corner = all(on)
    for d in range(3):
        if (mask >> (3 + d)) & 1 and on[d]:
            return True, corner
        if (mask >> (6 + d)) & 1 and on[(d + 1) % 3] and on[(d + 2) % 3]:
            return True, corner
    return False, corner


class LevelTables:
    def __init__(self, leaves, p, numbering_keys=None):
        """numbering_keys: optional external LABELS for the DoFs (array of node keys, index = DoF number, unconstrained DoFs
        first), e.g. the product's, so that numbering-dependent data (deal.II's Chebyshev start vector (i mod 11) - mean)
        coincide; mesh, classification, masks, gather lists and patches are computed here either way"""
        self.p, self.leaves = p, set(leaves)
        self.cells = o.sorted_cells(leaves)
        n1 = p + 1
        top = p << o.LMAX
        nc = len(self.cells)
        self.mask = np.array([_cell_mask(self.leaves, c) for c in self.cells], dtype=np.uint16)
        self.level = np.array([c[0] for c in self.cells], dtype=np.uint8)
        # pass 1: classify every own node key
        cls = {}  # key -> 0 free, 1 Dirichlet, 2 hanging
        order = []
        for ci, cell in enumerate(self.cells):
            m = int(self.mask[ci])
            for c in range(n1):
                for b in range(n1):
                    for a in range(n1):
                        key = o.node_key(cell, a, b, c, p)
                        con, corner = _on_constrained_entity(m, p, (a, b, c)) if m >> 3 else (False, False)
                        kind = 1 if any(key[t] == 0 or key[t] == top for t in range(3)) else (2 if (con and not corner) else 0)
                        if key not in cls:
                            cls[key] = kind
                            order.append(key)
                        elif kind == 2 and cls[key] == 0:
                            cls[key] = 2  # first seen from a cell for which it is regular?  cannot happen on a balanced mesh
                            raise AssertionError("a DoF is hanging for one cell and regular for another")
        counts = [sum(1 for k in order if cls[k] == q) for q in range(3)]
        self.n = len(order)
        self.first_constrained = counts[0]
        self.key_to_dof = {}
        if numbering_keys is None:
            nxt = [0, counts[0], counts[0] + counts[1]]
            for key in order:
                self.key_to_dof[key] = nxt[cls[key]]
                nxt[cls[key]] += 1
        else:
            assert len(numbering_keys) == self.n, (len(numbering_keys), self.n)
            for d, k in enumerate(numbering_keys):
                key = tuple(int(v) for v in k)
                assert (cls[key] == 0) == (d < counts[0]), "external numbering: the unconstrained DoFs must come first"
                self.key_to_dof[key] = d
            assert len(self.key_to_dof) == self.n
        self.keys = [None] * self.n
        for key, d in self.key_to_dof.items():
            self.keys[d] = key
        self.dirichlet = np.zeros(self.n, dtype=bool)
        for key, d in self.key_to_dof.items():
            self.dirichlet[d] = cls[key] == 1
        # pass 2: gathered (parent-resolved) indices per cell
        self.cell_dofs = np.full((nc, n1 ** 3), INVALID, dtype=np.uint32)
        for ci, cell in enumerate(self.cells):
            m = int(self.mask[ci])
            l, i, j, k = cell
            parent = (l - 1, i >> 1, j >> 1, k >> 1)
            t = 0
            for c in range(n1):
                for b in range(n1):
                    for a in range(n1):
                        con = _on_constrained_entity(m, p, (a, b, c))[0] if m >> 3 else False
                        key = o.node_key(parent if con else cell, a, b, c, p)
                        d = self.key_to_dof[key]  # the parent's DoF exists: the coarser neighbour owns it
                        self.cell_dofs[ci, t] = INVALID if self.dirichlet[d] else d
                        t += 1
        self.cell_index = {c: i for i, c in enumerate(self.cells)}

    def node_index(self, ci, a):
        """(gathered index, constrained, parent corner) of local node a = (x, y, z) of cell ci"""
        n1 = self.p + 1
        m = int(self.mask[ci])
        con, corner = _on_constrained_entity(m, self.p, a) if m >> 3 else (False, False)
        return int(self.cell_dofs[ci, (a[2] * n1 + a[1]) * n1 + a[0]]), con, corner

    def rhs_constant(self):
        """Operator::rhs for f == 1, g == 0 (ref:include/operator.h:362-413): load vector through the cells' gathers"""
        fe = o.FE1D(self.p)
        mc = np.kron(np.kron(fe.m, fe.m), fe.m)
        n1 = self.p + 1
        b = np.zeros(self.n)
        for ci, cell in enumerate(self.cells):
            h = 2.0 / (1 << cell[0])
            loc = (h ** 3) * mc.copy()
            if self.mask[ci] >> 3:
                loc = hanging_transpose(self.p, int(self.mask[ci]), loc)
            idx = self.cell_dofs[ci]
            ok = idx != INVALID
            np.add.at(b, idx[ok].astype(np.int64), loc[ok])
        return b


def _half_embedding(p):
    """I[c]: values of the parent's 1D basis at the child's nodes, child c in {0, 1} (rows = child nodes)"""
    fe = o.FE1D(p)
    nodes = np.array(fe.nodes, dtype=float)
    return [np.array([o.lagrange_eval(nodes, [0.5 * (x + c)])[0][0] for x in nodes]) for c in (0, 1)]


def hanging_transpose(p, mask, v):
    """transpose of the in-cell interpolation on the (p+1)^3 values v (x fastest), directions z, y, x"""
    n = p + 1
    I = _half_embedding(p)
    cp = (mask & 1, (mask >> 1) & 1, (mask >> 2) & 1)
    V = v.reshape(n, n, n).copy()  # [z, y, x]
    ax = {0: 2, 1: 1, 2: 0}  # direction -> numpy axis
    for d in (2, 1, 0):
        e, f = (d + 1) % 3, (d + 2) % 3
        fce, fcf, edg = (mask >> (3 + e)) & 1, (mask >> (3 + f)) & 1, (mask >> (6 + d)) & 1
        for ae in range(n):
            for af in range(n):
                one, onf = ae == cp[e] * p, af == cp[f] * p
                if not ((fce and one) or (fcf and onf) or (edg and one and onf)):
                    continue
                sl = [None, None, None]
                sl[ax[d]] = slice(None)
                sl[ax[e]] = ae
                sl[ax[f]] = af
                V[tuple(sl)] = I[cp[d]].T @ V[tuple(sl)]
    return V.reshape(-1)


def transfer_tables(fine: LevelTables, coarse: LevelTables):
    """list of (kind, nf, coarse_idx [n, (pc+1)^3], coarse_mask [n], fine_idx [n, nf^3]) in the format of
    dealii_multigrid_amd.transfer_tables / mgo_transfer_set_group"""
    pc, pf = coarse.p, fine.p
    nc1 = pc + 1
    groups = {k: ([], [], []) for k in range(3)}
    nfs = {0: pc + 1, 1: 2 * pc + 1, 2: pf + 1}
    claimed = np.zeros(fine.n, dtype=bool)
    for ci, cc in enumerate(coarse.cells):
        same = fine.cell_index.get(cc)
        if same is not None:
            kind = 0 if pf == pc else 2
        else:
            assert pf == pc, "refined cell in a p-transfer"
            kind = 1
        nf = nfs[kind]
        fi = np.full(nf ** 3, INVALID, dtype=np.uint32)
        l, i, j, k = cc
        t = -1
        for Z in range(nf):
            for Y in range(nf):
                for X in range(nf):
                    t += 1
                    idx = INVALID
                    if kind == 1:
                        # a node on the plane between two children belongs to both: the first child for which it is a
                        # regular (non-hanging, non-Dirichlet) node gives the index
                        lo = [1 if v > pc else 0 for v in (X, Y, Z)]
                        hi = [1 if v >= pc else 0 for v in (X, Y, Z)]
                        for cz, cy, cx in [(z_, y_, x_) for z_ in range(lo[2], hi[2] + 1) for y_ in range(lo[1], hi[1] + 1)
                                           for x_ in range(lo[0], hi[0] + 1)]:
                            f = fine.cell_index[(l + 1, 2 * i + cx, 2 * j + cy, 2 * k + cz)]
                            v, con, corner = fine.node_index(f, (X - cx * pc, Y - cy * pc, Z - cz * pc))
                            idx = INVALID if (con and not corner) else v
                            if idx != INVALID:
                                break
                    else:
                        v, con, corner = fine.node_index(same, (X, Y, Z))
                        idx = INVALID if (con and not corner) else v
                    if idx != INVALID:
                        if claimed[idx]:
                            idx = INVALID
                        else:
                            claimed[idx] = True
                    fi[t] = idx
        g = groups[kind]
        g[0].append(coarse.cell_dofs[ci].copy())
        g[1].append(coarse.mask[ci])
        g[2].append(fi)
    assert claimed[: fine.first_constrained].all(), "a free fine DoF belongs to no patch"
    out = []
    for kind in range(3):
        ci_, cm_, fi_ = groups[kind]
        n = len(cm_)
        out.append((kind, nfs[kind], np.array(ci_, dtype=np.uint32).reshape(n, nc1 ** 3), np.array(cm_, dtype=np.uint16),
                    np.array(fi_, dtype=np.uint32).reshape(n, nfs[kind] ** 3)))
    return out
