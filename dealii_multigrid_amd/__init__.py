"""MI355X-native matrix-free multigrid V-cycle -- Python host mirror of the reference's plugin surface.

Thin ctypes layer over the C ABI (`include/mgamd.h`, `lib/libmgamd.so`).  Class and method names
follow deal.II as the reference uses it:

  Triangulation            <-> parallel::distributed::Triangulation + GridGenerator  (ref:multigrid_throughput.cc:2041-2062)
  DoFs                     <-> DoFHandler + AffineConstraints + MatrixFree tables      (ref:multigrid_throughput.cc:1578-1595)
  Operator                 <-> Operator<3,1,Number>                                    (ref:include/operator.h:11-557)
  PreconditionChebyshev    <-> PreconditionChebyshev<Operator,Vector,DiagonalMatrix>   (ref:multigrid_throughput.cc:849-883)
  MGTwoLevelTransfer       <-> MGTwoLevelTransfer<3,Vector>                            (ref:multigrid_throughput.cc:1600-1604)
  PreconditionMG           <-> Multigrid + PreconditionMG + MGTransferGlobalCoarsening (ref:multigrid_throughput.cc:1093-1133)
  solve_cg                 <-> SolverCG + ReductionControl                             (ref:multigrid_throughput.cc:1140-1147)
  solve_with_global_coarsening  (ref:multigrid_throughput.cc:1443-1666)

There is no CPU fallback: every device call raises if the HIP library or a gfx950 GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

# MGAMD_LIBRARY: another build of the same library (development: the -DMGAMD_KERNEL_DEBUG build of `make debug`)
_LIB_PATH = os.environ.get("MGAMD_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmgamd.so")
F64, F32 = 8, 4
INVALID_DOF = 0xFFFFFFFF


class MgamdError(RuntimeError):
    pass


class NoDeviceError(MgamdError):
    pass


def _load():
    if not os.path.exists(_LIB_PATH):
        raise MgamdError(
            f"{_LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()); there is no fallback path"
        )
    lib = C.CDLL(_LIB_PATH)
    lib.mgamd_last_error.restype = C.c_char_p
    lib.mgamd_version.restype = C.c_char_p
    return lib


_lib = _load()


def _chk(status):
    if status != 0:
        msg = _lib.mgamd_last_error().decode()
        if status == 2:
            raise NoDeviceError(msg)
        raise MgamdError(msg)


class DofsInfo(C.Structure):
    _fields_ = [
        ("degree", C.c_uint32),
        ("n_cells", C.c_uint64),
        ("n_dofs", C.c_uint32),
        ("n_interior", C.c_uint32),
        ("n_tail", C.c_uint32),
        ("n_dirichlet", C.c_uint32),
        ("n_hanging", C.c_uint32),
        ("n_groups", C.c_uint32),
        ("group_B", C.c_uint32 * 8),
        ("group_slots", C.c_uint64 * 8),
        ("n_tail_owned", C.c_uint32),
        ("n_dirichlet_owned", C.c_uint32),
        ("n_hanging_owned", C.c_uint32),
        ("n_peers", C.c_uint32),
        ("n_halo_send", C.c_uint32),
        ("n_edge", C.c_uint32),
        ("group_halo_slots", C.c_uint64 * 8),
    ]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Triangulation:
    def __init__(self, geometry="hypercube", n_ref_global=0, n_ref_local=0, _handle=None):
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            _chk(_lib.mgamd_tria_create(geometry.encode(), C.c_uint(n_ref_global), C.c_uint(n_ref_local), C.byref(self._h)))
        nc, nl, nh = C.c_uint64(), C.c_uint32(), C.c_uint64()
        _chk(_lib.mgamd_tria_info(self._h, C.byref(nc), C.byref(nl), C.byref(nh)))
        self.n_cells, self.n_levels, self.n_cells_hn = nc.value, nl.value, nh.value

    @classmethod
    def from_leaves(cls, level, i, j, k) -> "Triangulation":
        """a caller-built octree over one root cell (mgamd_tria_create_from_leaves): leaves (level, i, j, k) in any order; must
        tile the cube and be 2:1 balanced across faces, edges and corners, else MgamdError (status MGAMD_ERR_INVALID)"""
        level, i, j, k = (np.ascontiguousarray(level, np.uint8), np.ascontiguousarray(i, np.uint32), np.ascontiguousarray(j, np.uint32),
                          np.ascontiguousarray(k, np.uint32))
        if not (len(level) == len(i) == len(j) == len(k)):
            raise ValueError("from_leaves: arrays of different lengths")
        h = C.c_void_p()
        _chk(_lib.mgamd_tria_create_from_leaves(C.c_uint64(len(level)), _ptr(level), _ptr(i), _ptr(j), _ptr(k), C.byref(h)))
        return cls(_handle=h)

    def level_mesh(self, level: int) -> "Triangulation":
        """local smoothing: all cells of refinement level `level`, active or not (distribute_mg_dofs levels)"""
        h = C.c_void_p()
        _chk(_lib.mgamd_tria_level_mesh(self._h, level, C.byref(h)))
        return Triangulation(_handle=h)

    def coarsen(self) -> "Triangulation":
        h = C.c_void_p()
        _chk(_lib.mgamd_tria_coarsen(self._h, C.byref(h)))
        return Triangulation(_handle=h)

    def cells(self):
        n = self.n_cells
        lev = np.zeros(n, np.uint8)
        i, j, k = (np.zeros(n, np.uint32) for _ in range(3))
        mask = np.zeros(n, np.uint16)
        _chk(_lib.mgamd_tria_get_cells(self._h, _ptr(lev), _ptr(i), _ptr(j), _ptr(k), _ptr(mask)))
        return lev, i, j, k, mask

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_tria_destroy(self._h)
            self._h = None


def create_geometric_coarsening_sequence(fine: Triangulation):
    """MGTransferGlobalCoarseningTools::create_geometric_coarsening_sequence: coarsest first."""
    seq = [fine]
    while seq[-1].n_cells > 1:
        seq.append(seq[-1].coarsen())
    return seq[::-1]


def create_polynomial_coarsening_sequence(degree: int):
    """...::create_polynomial_coarsening_sequence(degree, bisect): e.g. 4 -> [1, 2, 4]."""
    seq = [degree]
    while seq[-1] > 1:
        seq.append(max(seq[-1] // 2, 1))
    return seq[::-1]


class Partition:
    """Domain decomposition of a level hierarchy over n_ranks GPUs."""

    def __init__(self, trias, n_ranks: int, hanging_weight: float = 2.0, min_root_cells: int = 0, group: int = 1,
                 min_sub_root_cells: int = 0):
        """group > 1: two tiers -- levels below the root level with >= min_sub_root_cells cells are cut into n_ranks / group
        parts, each held by `group` consecutive ranks (mgamd_partition_create_tiered)"""
        self.trias, self.n_ranks = list(trias), n_ranks
        arr = (C.c_void_p * len(self.trias))(*[t._h for t in self.trias])
        self._h = C.c_void_p()
        _chk(_lib.mgamd_partition_create_tiered(arr, len(self.trias), n_ranks, C.c_double(hanging_weight), C.c_uint64(min_root_cells),
                                                group, C.c_uint64(min_sub_root_cells), C.byref(self._h)))
        rl, sl, g = C.c_uint(), C.c_uint(), C.c_uint()
        _chk(_lib.mgamd_partition_info(self._h, C.byref(rl), None))
        _chk(_lib.mgamd_partition_tiers(self._h, C.byref(sl), C.byref(g)))
        self.root_level, self.sub_root_level, self.group = rl.value, sl.value, g.value

    def n_parts(self, level: int) -> int:
        """pieces the level is cut into: n_ranks, n_ranks / group on the subset levels, 1 on the replicated ones"""
        return self.n_ranks if level >= self.root_level else (self.n_ranks // self.group if level >= self.sub_root_level else 1)

    def statistics(self):
        """MGTools::print_multigrid_statistics (ref:include/mg_tools.h:267-512) for this partition"""
        st = (C.c_double * 5)()
        _chk(_lib.mgamd_partition_statistics(self._h, st))
        return dict(zip(("workload_eff", "workload_path_max", "vertical_eff", "horizontal_eff", "mem_total"), st))

    def owner(self, level: int):
        o = np.zeros(self.trias[level].n_cells, np.uint16)
        _chk(_lib.mgamd_partition_get_owner(self._h, level, _ptr(o)))
        return o

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_partition_destroy(self._h)
            self._h = None


class DoFs:
    def __init__(self, tria: Triangulation, degree: int, max_brick: int = 0, partition: "Partition" = None, level: int = 0, rank: int = 0,
                 local_smoothing_level: bool = False):
        """local_smoothing_level: `tria` is Triangulation.level_mesh(l); refinement-edge DoFs are numbered [I | T | E | D | H]"""
        self.tria = tria
        self._h = C.c_void_p()
        if local_smoothing_level:
            _chk(_lib.mgamd_dofs_create_level(tria._h, degree, max_brick, C.byref(self._h)))
        elif partition is None:
            _chk(_lib.mgamd_dofs_create(tria._h, degree, max_brick, C.byref(self._h)))
        else:
            _chk(_lib.mgamd_dofs_create_local(partition._h, level, rank, degree, max_brick, C.byref(self._h)))
        self.info = DofsInfo()
        _chk(_lib.mgamd_dofs_info(self._h, C.byref(self.info)))
        self.degree = degree
        self.n_dofs = self.info.n_dofs

    def keys(self):
        k = np.zeros((self.n_dofs, 5), np.int32)
        _chk(_lib.mgamd_dofs_get_keys(self._h, _ptr(k)))
        return k

    def cell_dofs(self):
        out = np.zeros((self.info.n_cells, (self.degree + 1) ** 3), np.uint32)
        _chk(_lib.mgamd_dofs_get_cell_dofs(self._h, _ptr(out)))
        return out

    def rhs_function(self, kind: int):
        """Operator::rhs for SimulationType kind (0 Constant, 1 Gaussian), host vector"""
        b = np.zeros(self.n_dofs)
        _chk(_lib.mgamd_dofs_rhs(self._h, kind, _ptr(b)))
        return b

    def distribute(self, x, kind: int):
        """AffineConstraints::distribute on a host vector (Dirichlet values of `kind`, hanging-node interpolation)"""
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        _chk(_lib.mgamd_dofs_distribute(self._h, kind, _ptr(x)))
        return x

    def rhs_constant(self):
        b = np.zeros(self.n_dofs)
        _chk(_lib.mgamd_dofs_rhs_constant(self._h, _ptr(b)))
        return b

    def groups(self):
        return [(self.info.group_B[g], self.info.group_slots[g]) for g in range(self.info.n_groups)]

    def matrix(self):
        """the assembled level matrix (Operator::get_trilinos_system_matrix) as (row_ptr, col, val), CSR with sorted columns"""
        nnz = C.c_uint64()
        _chk(_lib.mgamd_dofs_matrix(self._h, C.byref(nnz), None, None, None))
        ptr, col, val = np.zeros(self.n_dofs + 1, np.uint32), np.zeros(nnz.value, np.uint32), np.zeros(nnz.value)
        _chk(_lib.mgamd_dofs_matrix(self._h, C.byref(nnz), _ptr(ptr), _ptr(col), _ptr(val)))
        return ptr, col, val

    def amg_setup_info(self):
        """[(rows, nnz)] of the smoothed-aggregation hierarchy of the AMG coarse solver on this level, finest first"""
        n, rows, nnz = C.c_uint32(), (C.c_uint32 * 32)(), (C.c_uint64 * 32)()
        _chk(_lib.mgamd_dofs_amg_setup_info(self._h, C.byref(n), rows, nnz, 32))
        return [(rows[l], nnz[l]) for l in range(n.value)]

    def cell_slots(self):
        grp, slot = np.zeros(self.info.n_cells, np.uint8), np.zeros(self.info.n_cells, np.uint32)
        _chk(_lib.mgamd_dofs_get_cell_slots(self._h, _ptr(grp), _ptr(slot)))
        return grp, slot

    def halo_plan(self):
        """host copy of the halo plan of a distributed level (see mgamd_dofs_halo_get)."""
        sz = (C.c_uint32 * 4)()
        _chk(_lib.mgamd_dofs_halo_sizes(self._h, sz))
        npeer, nsend, nsh, nc = (int(v) for v in sz)
        plan = dict(peers=np.zeros(npeer, np.int32), peer_offset=np.zeros(npeer + 1, np.uint32), pack_idx=np.zeros(nsend, np.uint32),
                    sh_tail=np.zeros(nsh, np.uint32), sh_ptr=np.zeros(nsh + 1, np.uint32), sh_src=np.zeros(nc, np.int32),
                    sh_owner_src=np.zeros(nsh, np.int32))
        if npeer:
            _chk(_lib.mgamd_dofs_halo_get(self._h, _ptr(plan["peers"]), _ptr(plan["peer_offset"]), _ptr(plan["pack_idx"]), _ptr(plan["sh_tail"]),
                                          _ptr(plan["sh_ptr"]), _ptr(plan["sh_src"]), _ptr(plan["sh_owner_src"])))
        return plan

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_dofs_destroy(self._h)
            self._h = None


def ls_copy_indices(active: DoFs, level_dofs: DoFs, level: int):
    """copy_to_mg / copy_from_mg index pairs (active-mesh index, level index) of refinement level `level`"""
    n = C.c_uint64()
    _chk(_lib.mgamd_ls_copy_indices(active._h, level_dofs._h, level, C.byref(n), None, None))
    g, l = np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint32)
    if n.value:
        _chk(_lib.mgamd_ls_copy_indices(active._h, level_dofs._h, level, C.byref(n), _ptr(g), _ptr(l)))
    return g, l


def transfer_tables(fine: DoFs, coarse: DoFs):
    """raw two-level tables for the CPU oracle: list of (kind, nf, coarse_idx, coarse_mask, fine_idx)."""
    npatch = (C.c_uint64 * 3)()
    nf = (C.c_uint32 * 3)()
    _chk(_lib.mgamd_transfer_tables_info(fine._h, coarse._h, npatch, nf))
    out = []
    nc3 = (coarse.degree + 1) ** 3
    for kind in range(3):
        n = npatch[kind]
        ci = np.zeros((n, nc3), np.uint32)
        cm = np.zeros(n, np.uint16)
        fi = np.zeros((n, nf[kind] ** 3), np.uint32)
        if n:
            _chk(_lib.mgamd_transfer_tables_get(fine._h, coarse._h, kind, _ptr(ci), _ptr(cm), _ptr(fi)))
        out.append((kind, int(nf[kind]), ci, cm, fi))
    return out


class Context:
    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _chk(_lib.mgamd_ctx_create(device, C.byref(self._h)))

    def synchronize(self):
        _chk(_lib.mgamd_ctx_synchronize(self._h))

    def stream(self) -> int:
        s = C.c_void_p()
        _chk(_lib.mgamd_ctx_stream(self._h, C.byref(s)))
        return s.value

    def kernel_profile(self, enable: bool, brick_size: int = 0):
        _chk(_lib.mgamd_ctx_kernel_profile_brick(self._h, brick_size))
        _chk(_lib.mgamd_ctx_kernel_profile(self._h, 1 if enable else 0))

    def kernel_profile_read(self):
        ms, n, b = C.c_double(), C.c_uint64(), C.c_double()
        _chk(_lib.mgamd_ctx_kernel_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value

    def kernel_profile_bytes_moved(self):
        b = C.c_double()
        _chk(_lib.mgamd_ctx_kernel_profile_bytes_moved(self._h, C.byref(b)))
        return b.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_ctx_destroy(self._h)
            self._h = None


class SimGroup:
    """In-process simulation of n ranks on one GPU (every rank is a host thread); for tests."""

    def __init__(self, n_ranks: int):
        self.n_ranks = n_ranks
        self._h = C.c_void_p()
        _chk(_lib.mgamd_sim_group_create(n_ranks, C.byref(self._h)))

    def comm(self, rank: int) -> "Communicator":
        h = C.c_void_p()
        _chk(_lib.mgamd_comm_sim_create(self._h, rank, C.byref(h)))
        return Communicator(h, self.n_ranks, rank, self)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_sim_group_destroy(self._h)
            self._h = None


class Communicator:
    def __init__(self, handle, n_ranks, rank, keepalive=None):
        self._h, self.n_ranks, self.rank, self._keep = handle, n_ranks, rank, keepalive

    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _chk(_lib.mgamd_comm_rccl_unique_id(buf))
        return buf.raw

    @staticmethod
    def rccl(ctx: "Context", n_ranks: int, rank: int, unique_id: bytes) -> "Communicator":
        h = C.c_void_p()
        _chk(_lib.mgamd_comm_rccl_create(ctx._h, n_ranks, rank, C.c_char_p(unique_id), C.byref(h)))
        return Communicator(h, n_ranks, rank)

    def subset(self, group: int) -> "Communicator":
        """communicator of a level that is cut into n_ranks / group parts (Partition tiers): rank = part"""
        if group == 1:
            return self
        h = C.c_void_p()
        _chk(_lib.mgamd_comm_subset(self._h, group, C.byref(h)))
        return Communicator(h, self.n_ranks // group, self.rank // group, self)

    def allreduce_sum(self, ctx: "Context", value: float) -> float:
        r = C.c_double()
        _chk(_lib.mgamd_comm_allreduce_sum(self._h, ctx._h, C.c_double(value), C.byref(r)))
        return r.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_comm_destroy(self._h)
            self._h = None


class Vector:
    """LinearAlgebra::distributed::Vector<Number>, device resident."""

    def __init__(self, ctx: Context, n: int, number_type: int = F64, _handle=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            _chk(_lib.mgamd_vec_create(ctx._h, C.c_uint64(n), number_type, C.byref(self._h)))
        sz = C.c_uint64()
        _chk(_lib.mgamd_vec_size(self._h, C.byref(sz)))
        self.n = sz.value

    def from_host(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n
        _chk(_lib.mgamd_vec_from_host(self._h, _ptr(a)))
        return self

    def to_host(self):
        a = np.zeros(self.n)
        _chk(_lib.mgamd_vec_to_host(self._h, _ptr(a)))
        return a

    def set(self, value: float):
        _chk(_lib.mgamd_vec_set(self._h, C.c_double(value)))

    def copy_from(self, other: "Vector"):
        _chk(_lib.mgamd_vec_copy(self._h, other._h))

    def sadd(self, s: float, a: float, x: "Vector"):
        _chk(_lib.mgamd_vec_sadd(self._h, C.c_double(s), C.c_double(a), x._h))

    def dot(self, other: "Vector") -> float:
        r = C.c_double()
        _chk(_lib.mgamd_vec_dot(self._h, other._h, C.byref(r)))
        return r.value

    def l2_norm(self) -> float:
        r = C.c_double()
        _chk(_lib.mgamd_vec_norm2(self._h, C.byref(r)))
        return r.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_vec_destroy(self._h)
            self._h = None


class Operator:
    """Operator<3,1,Number> (ref:include/operator.h:11-557)."""

    def __init__(self, ctx: Context, dofs: DoFs, number_type: int = F64, comm: "Communicator" = None):
        self.ctx, self.dofs, self.number_type, self.comm = ctx, dofs, number_type, comm
        self._h = C.c_void_p()
        if comm is None:
            _chk(_lib.mgamd_level_op_create(ctx._h, dofs._h, number_type, C.byref(self._h)))
        else:
            _chk(_lib.mgamd_level_op_create_distributed(ctx._h, dofs._h, number_type, comm._h, C.byref(self._h)))

    def dot(self, x: "Vector", y: "Vector") -> float:
        r = C.c_double()
        _chk(_lib.mgamd_level_op_dot(self._h, x._h, y._h, C.byref(r)))
        return r.value

    def n_owned(self) -> int:
        n = C.c_uint64()
        _chk(_lib.mgamd_level_op_n_owned(self._h, C.byref(n)))
        return n.value

    def m(self) -> int:
        n = C.c_uint64()
        _chk(_lib.mgamd_level_op_m(self._h, C.byref(n)))
        return n.value

    def initialize_dof_vector(self) -> Vector:
        h = C.c_void_p()
        _chk(_lib.mgamd_level_op_init_vector(self._h, C.byref(h)))
        return Vector(self.ctx, 0, _handle=h)

    def vmult(self, dst: Vector, src: Vector):
        _chk(_lib.mgamd_level_op_vmult(self._h, dst._h, src._h))

    def vmult_interface_up(self, dst: Vector, src: Vector):
        """local-smoothing level: the edge matrix, A with the refinement-edge DoFs unconstrained applied to src|edge"""
        _chk(_lib.mgamd_level_op_vmult_interface_up(self._h, dst._h, src._h))

    def exchange_add_tail(self, v: Vector):
        """distributed level: the shared entries of v become the sum over the sharing ranks (compress(add))"""
        _chk(_lib.mgamd_level_op_exchange_add_tail(self._h, v._h))

    def vmult_interface_down(self, dst: Vector, src: Vector):
        """Operator::vmult_interface_down: the plain cell loop (refinement-edge DoFs as ordinary DoFs), identity on the
        constrained rows -- the matrix of Multigrid's residual step (MGInterfaceOperator::vmult)"""
        _chk(_lib.mgamd_level_op_vmult_interface_down(self._h, dst._h, src._h))

    def compute_inverse_diagonal(self, diagonal: Vector):
        _chk(_lib.mgamd_level_op_inverse_diagonal(self._h, diagonal._h))

    def rhs(self, b: Vector, kind: int = 0):
        """Operator::rhs; kind = SimulationType (0 "Constant": f = 1, g = 0; 1 "Gaussian")"""
        _chk(_lib.mgamd_level_op_rhs_kind(self._h, kind, b._h))

    def distribute(self, x: Vector, kind: int = 0):
        """constraints.distribute(solution): Dirichlet values of `kind`, hanging nodes interpolated"""
        _chk(_lib.mgamd_level_op_distribute(self._h, kind, x._h))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_level_op_destroy(self._h)
            self._h = None


class PreconditionChebyshev:
    def __init__(self, op: Operator, degree=3, smoothing_range=20.0, eig_cg_n_iterations=20):
        self.op = op
        self._h = C.c_void_p()
        _chk(_lib.mgamd_cheb_create(op._h, degree, C.c_double(smoothing_range), eig_cg_n_iterations, C.byref(self._h)))

    def vmult(self, dst: Vector, src: Vector):
        _chk(_lib.mgamd_cheb_vmult(self._h, dst._h, src._h))

    def step(self, dst: Vector, src: Vector):
        _chk(_lib.mgamd_cheb_step(self._h, dst._h, src._h))

    def eigenvalue_estimates(self):
        lo, hi = C.c_double(), C.c_double()
        _chk(_lib.mgamd_cheb_get_eigen_estimates(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_cheb_destroy(self._h)
            self._h = None


class MGTwoLevelTransfer:
    def __init__(self, fine: Operator, coarse: Operator):
        self.fine, self.coarse = fine, coarse
        self._h = C.c_void_p()
        _chk(_lib.mgamd_transfer2_create(fine._h, coarse._h, C.byref(self._h)))

    def prolongate_and_add(self, dst_fine: Vector, src_coarse: Vector):
        _chk(_lib.mgamd_transfer2_prolongate_and_add(self._h, dst_fine._h, src_coarse._h))

    def restrict_and_add(self, dst_coarse: Vector, src_fine: Vector):
        _chk(_lib.mgamd_transfer2_restrict_and_add(self._h, dst_coarse._h, src_fine._h))

    def n_fused_bricks(self):
        """fine bricks whose share of this transfer runs inside the operator passes of the V-cycle (0: none)"""
        n = C.c_uint64()
        _chk(_lib.mgamd_transfer2_n_fused_bricks(self._h, C.byref(n)))
        return n.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_transfer2_destroy(self._h)
            self._h = None


STAGE_NAMES = ["pre_smoother_step", "residual_step", "restriction", "coarse_solve", "prolongation", "edge_prolongation",
               "post_smoother_step", "transfer_to_mg", "transfer_to_global"]
_STAGE_CB = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_uint, C.c_void_p)


class PreconditionMG:
    """Multigrid<Vector> + PreconditionMG over MGTransferGlobalCoarsening."""

    def __init__(self, ctx: Context, levels, transfers, smoothers, coarse_solver="direct", nested: "PreconditionMG" = None,
                 n_cycles: int = 1, local_smoothing: "DoFs" = None):
        """nested: geometric stand-in for the AMG coarse solvers on a large coarse level (an h-multigrid whose finest level
        is levels[0]), applied n_cycles times per coarse solve; see mgamd.h.
        local_smoothing: the DoFs of the ACTIVE mesh; `levels` are then operators on the refinement levels (HMG-local)"""
        self.ctx, self.levels, self.transfers, self.smoothers, self.nested = ctx, levels, transfers, smoothers, nested
        n = len(levels)
        L = (C.c_void_p * n)(*[l._h for l in levels])
        T = (C.c_void_p * n)(*[(t._h if t is not None else None) for t in transfers])
        S = (C.c_void_p * n)(*[(s._h if s is not None else None) for s in smoothers])
        self._h = C.c_void_p()
        if local_smoothing is not None:
            _chk(_lib.mgamd_mg_create_local_smoothing(ctx._h, n, L, T, S, local_smoothing._h, coarse_solver.encode(), C.byref(self._h)))
        else:
            _chk(_lib.mgamd_mg_create_nested(ctx._h, n, L, T, S, coarse_solver.encode(), nested._h if nested is not None else None, n_cycles,
                                             C.byref(self._h)))
        self._cb = None

    def set_collapse(self, enable: bool) -> int:
        """switch the tabulated coarse levels off/on; returns the collapse level (0: none)"""
        l = C.c_uint()
        _chk(_lib.mgamd_mg_set_collapse(self._h, 1 if enable else 0, C.byref(l)))
        return l.value

    def coarse_solver_used(self) -> str:
        buf = C.create_string_buffer(32)
        _chk(_lib.mgamd_mg_coarse_solver_used(self._h, buf))
        return buf.value.decode()

    def vmult(self, z: Vector, r: Vector):
        _chk(_lib.mgamd_mg_vcycle(self._h, z._h, r._h))

    def connect_stages(self, fn):
        """fn(stage:int, start:bool, level:int) or None; see STAGE_NAMES."""
        if fn is None:
            self._cb = None
            _chk(_lib.mgamd_mg_set_stage_callback(self._h, None, None))
            return
        self._cb = _STAGE_CB(lambda s, st, lv, u: fn(s, bool(st), lv))
        _chk(_lib.mgamd_mg_set_stage_callback(self._h, self._cb, None))

    def stage_timing(self, enable: bool):
        """HIP-event stage timing of the unchanged cycle (no host synchronisation inside the cycle)"""
        _chk(_lib.mgamd_mg_stage_timing(self._h, 1 if enable else 0))

    def stage_times(self):
        """milliseconds accumulated per [stage, level] since the last call (see STAGE_NAMES)"""
        n = len(self.levels)
        ms = np.zeros((9, n))
        cnt = C.c_uint64()
        _chk(_lib.mgamd_mg_stage_times(self._h, _ptr(ms), n, C.byref(cnt)))
        return ms

    def time_vcycles(self, z: Vector, r: Vector, n: int, use_graph: bool = True) -> float:
        ms = C.c_double()
        _chk(_lib.mgamd_mg_time_vcycles(self._h, z._h, r._h, n, 1 if use_graph else 0, C.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mgamd_mg_destroy(self._h)
            self._h = None


def solve_cg(A: Operator, preconditioner, x: Vector, b: Vector, reltol=1e-4, abstol=1e-20, maxiter=10000):
    """SolverCG<Vector>(ReductionControl(maxiter, abstol, reltol)).solve(A, x, b, preconditioner) from x = 0."""
    it, res = C.c_uint(), C.c_double()
    ph = preconditioner._h if preconditioner is not None else None
    _chk(_lib.mgamd_solve_cg(A._h, ph, x._h, b._h, C.c_double(reltol), C.c_double(abstol), maxiter, C.byref(it), C.byref(res)))
    return it.value, res.value


AMG_COARSE_SOLVERS = ("amg", "cg_with_amg", "amg_petsc")


class CoarseHierarchy:
    """Geometric stand-in for the reference's AMG coarse solvers (mgamd.h: "gmg_vcycle"): the h-multigrid on the coarse
    level of a PMG hierarchy (the lowest-degree space on the finest mesh); its finest level shares that level's objects."""

    def __init__(self, ctx, tria, dofs0, op0, smoother0, smoother_degree, smoothing_range, eig_cg_n_iterations, number_type, max_brick):
        self.trias = create_geometric_coarsening_sequence(tria)
        p0 = dofs0.degree
        self.dofs = [DoFs(t, p0, max_brick) for t in self.trias[:-1]] + [dofs0]
        self.operators = [Operator(ctx, d, number_type) for d in self.dofs[:-1]] + [op0]
        self.transfers = [None] + [MGTwoLevelTransfer(self.operators[l], self.operators[l - 1]) for l in range(1, len(self.dofs))]
        self.smoothers = [PreconditionChebyshev(op, smoother_degree, smoothing_range, eig_cg_n_iterations) for op in self.operators[:-1]]
        self.smoothers.append(smoother0)
        self.mg = PreconditionMG(ctx, self.operators, self.transfers, self.smoothers, "amg")


class Hierarchy:
    """What solve_with_global_coarsening builds (ref:multigrid_throughput.cc:1443-1666)."""

    def __init__(self, ctx: Context, geometry="quadrant", n_ref_global=3, degree=1, mg_type="HMG-global", n_ref_local=0,
                 smoother_degree=3, smoothing_range=20.0, eig_cg_n_iterations=20, coarse_solver="amg", number_type=F64,
                 max_brick=-1, coarse_n_cycles=1):
        self.ctx = ctx
        # `geometry` may also be a caller-built Triangulation (Triangulation.from_leaves)
        fine = geometry if isinstance(geometry, Triangulation) else Triangulation(geometry, n_ref_global, n_ref_local)
        if mg_type == "HMG-global":
            self.trias = create_geometric_coarsening_sequence(fine)
            self.degrees = [degree] * len(self.trias)
        elif mg_type == "PMG":
            self.degrees = create_polynomial_coarsening_sequence(degree)
            self.trias = [fine] * len(self.degrees)
        elif mg_type == "HPMG":
            # h-levels at the lowest degree of the bisection sequence, then the p-levels on the finest mesh
            # (ref:multigrid_throughput.cc:1518-1519, 1551-1553, 1569-1571)
            pseq = create_polynomial_coarsening_sequence(degree)
            hseq = create_geometric_coarsening_sequence(fine)
            self.trias = hseq + [fine] * (len(pseq) - 1)
            self.degrees = [pseq[0]] * len(hseq) + pseq[1:]
        elif mg_type == "HMG-local":
            self._build_local_smoothing(ctx, fine, degree, smoother_degree, smoothing_range, eig_cg_n_iterations, coarse_solver, number_type,
                                        max_brick)
            return
        elif mg_type == "HPMG-local":
            # ref:multigrid_throughput.cc:1685-1695,1846-1860: p-multigrid on the active mesh whose coarse problem (lowest degree)
            # is handed to ONE local-smoothing V-cycle (MGCoarseGridApplyPreconditioner of the intermediate PreconditionMG)
            pseq = create_polynomial_coarsening_sequence(degree)
            self._build_local_smoothing(ctx, fine, pseq[0], smoother_degree, smoothing_range, eig_cg_n_iterations, coarse_solver, number_type,
                                        max_brick)
            if len(pseq) == 1:
                return
            self.ls = dict(trias=self.trias, dofs=self.dofs, operators=self.operators, transfers=self.transfers, smoothers=self.smoothers,
                           mg=self.mg)
            self.trias = [fine] * len(pseq)
            self.degrees = pseq
            self.dofs = [self.active_dofs] + [DoFs(fine, p, max_brick) for p in pseq[1:]]
            self.operators = [Operator(ctx, d, number_type) for d in self.dofs]
            self.transfers = [None] + [MGTwoLevelTransfer(self.operators[l], self.operators[l - 1]) for l in range(1, len(self.dofs))]
            self.smoothers = [PreconditionChebyshev(op, smoother_degree, smoothing_range, eig_cg_n_iterations) for op in self.operators]
            self.mg = PreconditionMG(ctx, self.operators, self.transfers, self.smoothers, "gmg_vcycle", self.ls["mg"], 1)
            self.fine_operator = self.operators[-1] if number_type == F64 else Operator(ctx, self.dofs[-1], F64)
            self.n_dofs = self.dofs[-1].n_dofs
            return
        else:
            raise MgamdError(f"Type '{mg_type}' not implemented")
        # max_brick=-1: bricks on large levels, single-cell slots on the latency-bound small ones (level_tables.hpp)
        self.dofs = [DoFs(t, p, max_brick) for t, p in zip(self.trias, self.degrees)]
        self.operators = [Operator(ctx, d, number_type) for d in self.dofs]
        self.transfers = [None] + [MGTwoLevelTransfer(self.operators[l], self.operators[l - 1]) for l in range(1, len(self.dofs))]
        self.smoothers = [PreconditionChebyshev(op, smoother_degree, smoothing_range, eig_cg_n_iterations) for op in self.operators]
        self.coarse = None
        if coarse_solver == "gmg_vcycle" and self.dofs[0].n_dofs > 4096:
            # the geometric stand-in for the AMG coarse solvers on a large coarse level (PMG), now by explicit request only:
            # V-cycles of the h-multigrid on level 0 ("amg", "cg_with_amg" run the library's smoothed-aggregation AMG)
            self.coarse = CoarseHierarchy(ctx, self.trias[0], self.dofs[0], self.operators[0], self.smoothers[0], smoother_degree,
                                          smoothing_range, eig_cg_n_iterations, number_type, max_brick)
        self.mg = PreconditionMG(ctx, self.operators, self.transfers, self.smoothers, coarse_solver,
                                 self.coarse.mg if self.coarse else None, coarse_n_cycles)
        self.fine_operator = self.operators[-1] if number_type == F64 else Operator(ctx, self.dofs[-1], F64)
        self.n_dofs = self.dofs[-1].n_dofs


def _hierarchy_build_local_smoothing(self, ctx, fine, degree, smoother_degree, smoothing_range, eig_cg_n_iterations, coarse_solver,
                                     number_type, max_brick):
    """solve_with_local_smoothing (ref:multigrid_throughput.cc:1670-1873): operators on the refinement levels 0..L of the
    octree, MGTransferMatrixFree between them, edge matrices, the outer operator on the active mesh"""
    self.active_dofs = DoFs(fine, degree, max_brick)
    self.trias = [fine.level_mesh(l) for l in range(fine.n_levels)]
    self.degrees = [degree] * len(self.trias)
    self.dofs = [DoFs(t, degree, max_brick, local_smoothing_level=True) for t in self.trias]
    self.operators = [Operator(ctx, d, number_type) for d in self.dofs]
    self.transfers = [None] + [MGTwoLevelTransfer(self.operators[l], self.operators[l - 1]) for l in range(1, len(self.dofs))]
    self.smoothers = [PreconditionChebyshev(op, smoother_degree, smoothing_range, eig_cg_n_iterations) for op in self.operators]
    self.coarse = None
    self.mg = PreconditionMG(ctx, self.operators, self.transfers, self.smoothers, coarse_solver, local_smoothing=self.active_dofs)
    self.fine_operator = Operator(ctx, self.active_dofs, F64)
    self.n_dofs = self.active_dofs.n_dofs


Hierarchy._build_local_smoothing = _hierarchy_build_local_smoothing


class DistributedHierarchy:
    """One rank's share of the hierarchy (HMG-global, PMG or HPMG): levels on meshes below the partition's root level are
    replicated, the others hold this rank's cells and exchange the partial sums of shared DoFs through `comm` (RCCL over
    xGMI in production).  The p-levels of PMG/HPMG live on the finest mesh: they inherit its partition, p-transfers are
    rank-local, and the p = 1 coarse problem of PMG is solved by the distributed CG (+ Chebyshev) or, for the AMG choices,
    by V-cycles of the (equally sharded) h-multigrid below it."""

    def __init__(self, ctx: Context, comm: Communicator, geometry="quadrant", n_ref_global=3, degree=1, smoother_degree=3,
                 smoothing_range=20.0, eig_cg_n_iterations=20, coarse_solver="amg", number_type=F64, hanging_weight=2.0, max_brick=-1,
                 min_root_dofs=4_000_000, mg_type="HMG-global", coarse_n_cycles=1, subset_group=None, min_subset_dofs=1_000_000):
        self.ctx, self.comm = ctx, comm
        fine = geometry if isinstance(geometry, Triangulation) else Triangulation(geometry, n_ref_global)
        self.mesh_sequence = create_geometric_coarsening_sequence(fine)
        nm = len(self.mesh_sequence)
        # (mesh index, degree) of every multigrid level, coarse -> fine (ref:multigrid_throughput.cc:1506-1571)
        if mg_type == "HMG-global":
            plan = [(l, degree) for l in range(nm)]
        elif mg_type == "PMG":
            plan = [(nm - 1, p) for p in create_polynomial_coarsening_sequence(degree)]
        elif mg_type == "HPMG":
            pseq = create_polynomial_coarsening_sequence(degree)
            plan = [(l, pseq[0]) for l in range(nm)] + [(nm - 1, p) for p in pseq[1:]]
        else:
            raise MgamdError(f"Type '{mg_type}' not implemented")
        # levels below ~4 M DoFs stay replicated: their single-GPU time (latency-bound: 0.34 ms for 2.3 M DoFs at p=4,
        # 0.33 ms for 2.2 M at p=1) is below what a distributed level pays for its 8 halo exchanges per cycle on top of its
        # own (also latency-bound) kernels
        # (round 3) levels between min_subset_dofs and min_root_dofs are cut into n_ranks / subset_group parts, each held by a
        # group of ranks (Partition tiers): default groups of 4 from 8 ranks on, of 2 for 4-7 ranks
        p_low = min(p for _, p in plan)
        if subset_group is None:
            subset_group = 4 if comm.n_ranks % 4 == 0 and comm.n_ranks >= 8 else (2 if comm.n_ranks % 2 == 0 and comm.n_ranks >= 4 else 1)
        self.partition = Partition(self.mesh_sequence, comm.n_ranks, hanging_weight, min_root_dofs // p_low ** 3, subset_group,
                                   min_subset_dofs // p_low ** 3)
        sharded = comm.n_ranks > 1
        sub_comm = comm.subset(self.partition.group) if self.partition.group > 1 else comm
        self.level_comm = lambda mi: comm if mi >= self.partition.root_level else sub_comm

        def build(levels, shared_level0=None):
            """levels: list of (mesh index, degree); shared_level0: (dofs, operator, smoother) reused as the LAST level"""
            dofs = [DoFs(self.mesh_sequence[mi], p, max_brick, self.partition, mi, comm.rank) for mi, p in levels]
            dist = [sharded and mi >= self.partition.sub_root_level for mi, _ in levels]
            if shared_level0 is not None:
                dofs[-1] = shared_level0[0]
            ops = [Operator(ctx, d, number_type, self.level_comm(levels[l][0]) if dist[l] else None) for l, d in enumerate(dofs)]
            if shared_level0 is not None:
                ops[-1] = shared_level0[1]
            tr = [None] + [MGTwoLevelTransfer(ops[l], ops[l - 1]) for l in range(1, len(ops))]
            sm = [PreconditionChebyshev(op, smoother_degree, smoothing_range, eig_cg_n_iterations) for op in ops]
            if shared_level0 is not None:
                sm[-1] = shared_level0[2]
            return dofs, dist, ops, tr, sm

        self.trias = [self.mesh_sequence[mi] for mi, _ in plan]
        self.degrees = [p for _, p in plan]
        self.dofs, self.distributed, self.operators, self.transfers, self.smoothers = build(plan)
        self.coarse = None
        self.plan = plan
        n0 = self.global_level_dofs(ctx)[0]
        if (coarse_solver == "gmg_vcycle" or (coarse_solver in AMG_COARSE_SOLVERS and self.distributed[0])) and n0 > 4096:
            # geometric stand-in for the AMG coarse solvers: the h-multigrid on level 0's space (mgamd.h, "gmg_vcycle"); the
            # algebraic multigrid is built from ONE rank's assembled matrix, so a sharded coarse level takes the stand-in
            mi0, p0 = plan[0]
            cd, cdist, cops, ctr, csm = build([(l, p0) for l in range(mi0 + 1)], (self.dofs[0], self.operators[0], self.smoothers[0]))
            self.coarse = PreconditionMG(ctx, cops, ctr, csm, "amg")
            self.coarse.parts = (cd, cdist)
        self.mg = PreconditionMG(ctx, self.operators, self.transfers, self.smoothers, coarse_solver, self.coarse, coarse_n_cycles)
        self.fine_operator = self.operators[-1]
        self.n_local = self.dofs[-1].n_dofs
        self.n_dofs = self.global_level_dofs(ctx)[-1]

    def global_level_dofs(self, ctx):
        """DoFs per multigrid level: owned DoFs summed over the pieces of the level (replicated levels are complete on every rank)"""
        return [int(round(self.level_comm(self.plan[l][0]).allreduce_sum(ctx, float(op.n_owned())))) if self.distributed[l]
                else self.dofs[l].n_dofs for l, op in enumerate(self.operators)]

    def layout(self):
        """how many pieces every multigrid level is cut into (n_ranks, n_ranks / group on the subset tier, 1 = replicated)"""
        return [self.partition.n_parts(mi) if self.comm.n_ranks > 1 else 1 for mi, _ in self.plan]
