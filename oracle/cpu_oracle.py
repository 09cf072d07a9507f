"""ORACLE (test infrastructure): ctypes wrapper of oracle/_build/libmgoracle_cpu.so, the C++/OpenMP CPU
restatement of the hot path (oracle/cpu/mg_cpu_oracle.cpp).  PARITY UNPINNED, see mgoracle.py.
It consumes the index tables exported by the product's host setup (validated against the independent
numpy oracle in tests/test_host_setup.py) and redoes all arithmetic on the CPU in the reference's
quadrature-based formulation.  Only tests/, smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmgoracle_cpu.so")


def _cpu_share():
    """CPUs this process may really use: affinity mask and cgroup quota (the GPU box gives 16 of 128)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def _host_cpu_flags():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return " ".join(sorted(line.split(":", 1)[1].split()))
    except Exception:
        pass
    return "unknown"


def _load():
    if "OMP_NUM_THREADS" not in os.environ:
        os.environ["OMP_NUM_THREADS"] = str(min(_cpu_share(), 16))  # gpurun box: 16 CPUs per GPU
    # the library is built with -march=native: a copy that travelled from another host (the build container -> the GPU box) is
    # rebuilt here when the CPU's feature flags differ from the ones it was built on (3 s), so that the cpu_baseline is timed on
    # code generated for THIS host
    stamp = os.path.join(_HERE, "_build", "cpu_flags.txt")
    flags = _host_cpu_flags()
    stale = not os.path.exists(_SO) or not os.path.exists(stamp) or open(stamp).read() != flags
    if stale:
        try:
            subprocess.check_call(["make", "-B", "-C", _HERE], stdout=subprocess.DEVNULL)
            with open(stamp, "w") as f:
                f.write(flags)
        except Exception:
            if not os.path.exists(_SO):
                raise
    lib = C.CDLL(_SO)
    lib.mgo_level_create.restype = C.c_void_p
    lib.mgo_transfer_create.restype = C.c_void_p
    lib.mgo_mg_create.restype = C.c_void_p
    lib.mgo_mg_max_eigenvalue.restype = C.c_double
    lib.mgo_mg_time_vcycles.restype = C.c_double
    # libgomp may already be initialised (torch/numpy) with the machine's core count: set the team size explicitly
    lib.mgo_set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
    return lib


_lib = _load()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def built_for_this_host():
    """True if the loaded library was compiled on a host with this host's CPU feature flags (see _load)"""
    stamp = os.path.join(_HERE, "_build", "cpu_flags.txt")
    return os.path.exists(stamp) and open(stamp).read() == _host_cpu_flags()


def num_threads():
    return _lib.mgo_num_threads()


def set_num_threads(n):
    _lib.mgo_set_num_threads(int(n))


class CpuLevel:
    def __init__(self, dofs=None, tables=None):
        """dofs: dealii_multigrid_amd.DoFs (only its exported tables are used); or tables: own_tables.LevelTables (the
        oracle's own mesh, numbering and gather lists: nothing of the product involved)."""
        if tables is not None:
            self.p, self.n = tables.p, tables.n
            cd = np.ascontiguousarray(tables.cell_dofs, dtype=np.uint32)
            lev, mask = np.ascontiguousarray(tables.level, np.uint8), np.ascontiguousarray(tables.mask, np.uint16)
            first_c = tables.first_constrained
        else:
            self.p = dofs.degree
            self.n = dofs.n_dofs
            cd = np.ascontiguousarray(dofs.cell_dofs())
            lev, _, _, _, mask = dofs.tria.cells()
            first_c = dofs.info.n_interior + dofs.info.n_tail
        self._h = C.c_void_p(
            _lib.mgo_level_create(self.p, C.c_uint64(cd.shape[0]), C.c_uint32(self.n), C.c_uint32(first_c), _p(cd), _p(lev), _p(mask))
        )

    def vmult(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.n)
        _lib.mgo_level_vmult(self._h, _p(y), _p(x))
        return y

    def inverse_diagonal(self):
        d = np.zeros(self.n)
        _lib.mgo_level_inverse_diagonal(self._h, _p(d))
        return d

    def n_colors(self):
        return _lib.mgo_level_n_colors(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.mgo_level_destroy(self._h)
            self._h = None


class CpuTransfer:
    def __init__(self, fine: CpuLevel, coarse: CpuLevel, tables):
        """tables: dealii_multigrid_amd.transfer_tables(fine_dofs, coarse_dofs)."""
        self.fine, self.coarse = fine, coarse
        self._h = C.c_void_p(_lib.mgo_transfer_create(fine._h, coarse._h))
        for kind, nf, ci, cm, fi in tables:
            if ci.shape[0]:
                _lib.mgo_transfer_set_group(
                    self._h, kind, C.c_uint64(ci.shape[0]), nf, _p(np.ascontiguousarray(ci)), _p(np.ascontiguousarray(cm)), _p(np.ascontiguousarray(fi))
                )

    def prolongate_and_add(self, dst, src):
        dst = np.ascontiguousarray(dst, dtype=np.float64).copy()
        src = np.ascontiguousarray(src, dtype=np.float64)
        _lib.mgo_transfer_prolongate_and_add(self._h, _p(dst), _p(src))
        return dst

    def restrict_and_add(self, dst, src):
        dst = np.ascontiguousarray(dst, dtype=np.float64).copy()
        src = np.ascontiguousarray(src, dtype=np.float64)
        _lib.mgo_transfer_restrict_and_add(self._h, _p(dst), _p(src))
        return dst

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.mgo_transfer_destroy(self._h)
            self._h = None


class CpuMultigrid:
    def __init__(self, levels, transfers, smoother_degree=3, smoothing_range=20.0, eig_cg_n_iterations=20, coarse="direct"):
        self.levels, self.transfers = levels, transfers
        n = len(levels)
        L = (C.c_void_p * n)(*[l._h for l in levels])
        T = (C.c_void_p * n)(*[(t._h if t is not None else None) for t in transfers])
        self._h = C.c_void_p(_lib.mgo_mg_create(n, L, T, smoother_degree, C.c_double(smoothing_range), eig_cg_n_iterations, coarse.encode()))

    def max_eigenvalue(self, level):
        return _lib.mgo_mg_max_eigenvalue(self._h, level)

    def vcycle(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        _lib.mgo_mg_vcycle(self._h, _p(z), _p(r))
        return z

    def time_vcycles(self, r, n):
        r = np.ascontiguousarray(r, dtype=np.float64)
        return _lib.mgo_mg_time_vcycles(self._h, _p(r), n)

    def solve_cg(self, b, reltol=1e-4, abstol=1e-20, maxiter=10000):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        res = C.c_double()
        it = _lib.mgo_solve_cg(self.levels[-1]._h, self._h, _p(x), _p(b), C.c_double(reltol), C.c_double(abstol), maxiter, C.byref(res))
        return x, it, res.value

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.mgo_mg_destroy(self._h)
            self._h = None


def build_from_own_tables(geometry, n_ref, degree, mg_type="HMG-global", smoother_degree=3, coarse="direct", numbering_keys=None):
    """the whole hierarchy from the oracle's own tables (own_tables.py): meshes from mgoracle.create_mesh / coarsening_sequence,
    the oracle's numbering (or the DoF labels numbering_keys[level], see own_tables.LevelTables).  Returns (tables per level,
    levels, transfers, multigrid)."""
    import mgoracle as o
    import own_tables as ot

    fine = o.create_mesh(geometry, n_ref)
    if mg_type == "HMG-global":
        spaces = [(m, degree) for m in o.coarsening_sequence(fine)]
    elif mg_type == "PMG":
        seq = [degree]
        while seq[-1] > 1:
            seq.append(max(seq[-1] // 2, 1))
        spaces = [(fine, p) for p in seq[::-1]]
    else:
        raise ValueError(mg_type)
    tabs = [ot.LevelTables(m, p, None if numbering_keys is None else numbering_keys[l]) for l, (m, p) in enumerate(spaces)]
    levels = [CpuLevel(tables=t) for t in tabs]
    transfers = [None] + [CpuTransfer(levels[l], levels[l - 1], ot.transfer_tables(tabs[l], tabs[l - 1])) for l in range(1, len(levels))]
    return tabs, levels, transfers, CpuMultigrid(levels, transfers, smoother_degree, coarse=coarse)


def build_from_dofs(dofs_list, transfer_tables_fn, smoother_degree=3, coarse="direct"):
    """dofs_list: product DoFs objects coarse -> fine; transfer_tables_fn(fine, coarse) -> tables."""
    levels = [CpuLevel(d) for d in dofs_list]
    transfers = [None] + [CpuTransfer(levels[l], levels[l - 1], transfer_tables_fn(dofs_list[l], dofs_list[l - 1])) for l in range(1, len(levels))]
    return levels, transfers, CpuMultigrid(levels, transfers, smoother_degree, coarse=coarse)
