// Device runtime: HIP stream + device vectors + the objects behind the C ABI.
// Class and method names follow deal.II's plugin surface as the reference uses it
// (SURVEY.md section 8b): LevelOperator <-> Operator<3,1,Number> (ref:include/operator.h),
// Chebyshev <-> PreconditionChebyshev, Transfer2 <-> MGTwoLevelTransfer,
// Multigrid <-> Multigrid + PreconditionMG + MGTransferGlobalCoarsening, solve_cg <-> SolverCG.
#pragma once
#include "api_common.hpp"
#include "comm.hpp"

#include <hip/hip_runtime.h>

#include <functional>
#include <map>
#include <sstream>

namespace mgamd
{
#define HIP_CHECK(expr)                                                                                       \
  do                                                                                                          \
    {                                                                                                         \
      hipError_t e_ = (expr);                                                                                 \
      if (e_ != hipSuccess)                                                                                   \
        {                                                                                                     \
          std::ostringstream os_;                                                                             \
          os_ << "HIP error " << hipGetErrorString(e_) << " at " << __FILE__ << ":" << __LINE__ << ": " #expr; \
          throw std::runtime_error(os_.str());                                                                \
        }                                                                                                     \
    }                                                                                                         \
  while (0)

  struct Ctx
  {
    int         device = 0;
    hipStream_t stream = nullptr;
    int         n_cu   = 256;     // compute units (grid of the persistent-workgroup kernels)
    double     *d_partial = nullptr; // 1024 block partials
    double     *d_result  = nullptr; // 8 scalars
    double     *h_result  = nullptr; // pinned
    double     *d_cg      = nullptr; // 8 scalars of the device-resident CG (kernels.hpp)
    // dominant-kernel profiling (HIP events around the largest lattice_apply launches)
    bool                                           profile = false;
    int                                            prof_brick = 0; // 0: the dominant group of each level; B: groups of B^3 bricks only
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t                                         prof_used  = 0;
    double                                         prof_bytes = 0.0;       // SURVEY 8(d) figure: the algorithm's words
    double                                         prof_bytes_moved = 0.0; // what the kernels are written to move (closed-form D^-1)
    double                                         prof_ms_accum = 0.0; // already harvested
    uint64_t                                       prof_n_accum  = 0;

    // second queue of the pipelined operator pass (small-slot kernels and tail stages overlap with the brick chunks) and a
    // ring of timing-free events for the cross-queue ordering
    hipStream_t             side = nullptr;
    std::vector<hipEvent_t> sync_events;
    size_t                  sync_next = 0;

    explicit Ctx(int dev);
    ~Ctx();
    void
    sync()
    {
      HIP_CHECK(hipStreamSynchronize(stream));
    }
    // `to` waits for everything enqueued on `from` so far
    void
    order_after(hipStream_t to, hipStream_t from)
    {
      hipEvent_t e = sync_events[sync_next];
      sync_next    = (sync_next + 1) % sync_events.size();
      HIP_CHECK(hipEventRecord(e, from));
      HIP_CHECK(hipStreamWaitEvent(to, e, 0));
    }
    void
    harvest_profile();
  };

  template <typename T>
  struct DBuf
  {
    T     *p = nullptr;
    size_t n = 0;
    DBuf() = default;
    DBuf(const DBuf &) = delete;
    DBuf &
    operator=(const DBuf &) = delete;
    ~DBuf()
    {
      if (p)
        (void)hipFree(p);
    }
    void
    alloc(size_t count)
    {
      if (p)
        (void)hipFree(p);
      p = nullptr;
      n = count;
      if (count)
        HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
    }
    void
    upload(const std::vector<T> &h)
    {
      alloc(h.size());
      if (n)
        HIP_CHECK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    }
    void
    zero(hipStream_t s)
    {
      if (n)
        HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
  };
} // namespace mgamd

// ---- opaque handles -------------------------------------------------------------------------
struct mgamd_ctx
{
  std::unique_ptr<mgamd::Ctx> ctx;
};

struct mgamd_vec
{
  mgamd::Ctx *ctx  = nullptr;
  size_t      n    = 0;
  int         type = MGAMD_F64;
  void       *data = nullptr;
  ~mgamd_vec();
  template <typename T>
  T *
  as()
  {
    check<T>();
    return static_cast<T *>(data);
  }
  template <typename T>
  const T *
  as() const
  {
    check<T>();
    return static_cast<const T *>(data);
  }
  template <typename T>
  void
  check() const
  {
    if ((int)sizeof(T) != type)
      throw std::invalid_argument("vector number type does not match the operator's");
  }
};

namespace mgamd
{
  mgamd_vec *
  vec_create(Ctx *ctx, size_t n, int type);

  // launch helpers on type-erased vectors
  void
  vec_set(mgamd_vec &v, double value);
  void
  vec_copy(mgamd_vec &dst, const mgamd_vec &src);
  void
  vec_sadd(mgamd_vec &y, double s, double a, const mgamd_vec &x);
  double
  vec_dot(const mgamd_vec &x, const mgamd_vec &y);

  struct LevelOperatorBase
  {
    Ctx                         *ctx  = nullptr;
    int                          type = MGAMD_F64;
    std::shared_ptr<LevelTables> tables;
    std::shared_ptr<Tria>        tria;
    std::shared_ptr<Comm>        comm; // sharded runs: set when this level is distributed
    std::shared_ptr<HaloPlan>    halo_plan;
    virtual ~LevelOperatorBase() = default;
    // inner product counting every DoF once across ranks (copies of DoFs owned by other ranks are skipped)
    virtual double
    dot(const mgamd_vec &x, const mgamd_vec &y) = 0;
    // sum over the sharing ranks of the tail entries of a vector (in place); no-op on one rank
    virtual void
    exchange_add_tail(mgamd_vec &v) = 0;
    uint64_t
    n_dofs_owned() const
    {
      return (uint64_t)tables->n_interior + tables->n_tail_owned + tables->n_dirichlet_owned + tables->n_hanging_owned;
    }
    uint32_t
    n_dofs() const
    {
      return tables->n_dofs;
    }
    virtual void
    vmult(mgamd_vec &dst, const mgamd_vec &src) = 0;
    virtual void
    compute_inverse_diagonal(mgamd_vec &d) = 0;
    virtual void
    vmult_interface_up(mgamd_vec &dst, const mgamd_vec &src) = 0; // local-smoothing levels: the edge matrix
    virtual void
    vmult_interface_down(mgamd_vec &dst, const mgamd_vec &src) = 0; // the level matrix of Multigrid's residual step
    virtual size_t
    read_debug_stamps(unsigned long long *out, size_t max_count) = 0;
    void
    rhs(mgamd_vec &b, int kind = 0);
    void
    distribute(mgamd_vec &x, int kind);
  };

  struct ChebyshevBase
  {
    virtual ~ChebyshevBase() = default;
    LevelOperatorBase *op = nullptr;
    unsigned           degree = 3;
    double             min_eig = 0, max_eig = 0, theta = 1, delta = 0;
    virtual void
    vmult(mgamd_vec &dst, const mgamd_vec &src) = 0;
    virtual void
    step(mgamd_vec &dst, const mgamd_vec &src) = 0;
  };

  struct Transfer2Base
  {
    virtual ~Transfer2Base() = default;
    LevelOperatorBase *fine = nullptr, *coarse = nullptr;
    virtual void
    prolongate_and_add(mgamd_vec &dst_fine, const mgamd_vec &src_coarse) = 0;
    virtual void
    restrict_and_add(mgamd_vec &dst_coarse, const mgamd_vec &src_fine) = 0;
    // bricks of the fine level whose part of this transfer can run inside the level operator's passes (0: none)
    virtual uint64_t
    n_fused_bricks_total() const = 0;
  };

  struct MultigridBase
  {
    virtual ~MultigridBase() = default;
    Ctx                 *ctx = nullptr;
    mgamd_stage_callback cb  = nullptr;
    void                *cb_user = nullptr;
    virtual void
    vcycle(mgamd_vec &z, const mgamd_vec &r) = 0; // PreconditionMG::vmult
    virtual double
    time_vcycles(mgamd_vec &z, const mgamd_vec &r, unsigned n, bool use_graph) = 0;
    virtual unsigned
    n_levels() const = 0;
    // turn this hierarchy (levels built with mgamd_dofs_create_level) into a local-smoothing preconditioner of the problem on
    // the active mesh `active`
    virtual void
    setup_local_smoothing(const LevelTables &active) = 0;
    // switch the tabulated (collapsed) coarse levels on/off at run time; returns the collapse level (0: none)
    virtual unsigned
    set_collapse(bool on) = 0;
    // the coarse solver actually in use ("direct", "cg", "cg_with_chebyshev", "gmg_vcycle")
    std::string coarse_used;
    int         number_type = MGAMD_F64;
    virtual LevelOperatorBase *
    finest_operator() const = 0;
    // tables of the vectors vcycle() acts on: the finest level's, or the active mesh's for local smoothing
    virtual const LevelTables *
    outer_tables() const = 0;
    // z = V-cycle(r) on raw device pointers of the LEVEL number type (nested use)
    virtual void
    vcycle_level_raw(void *z, const void *r) = 0;

    // Stage times without host synchronisation: a HIP event pair is recorded on the stream around every stage of the
    // UNCHANGED cycle (same code path as an un-instrumented cycle, collapsed coarse levels included) and resolved when read.
    struct StageRecord
    {
      int        stage;
      unsigned   level;
      hipEvent_t e0, e1;
    };
    bool                     stage_timing = false;
    std::vector<StageRecord> stage_records;
    size_t                   stage_used = 0;
    void
    set_stage_timing(bool on)
    {
      ctx->sync();
      stage_timing = on;
      stage_used   = 0;
    }
    // ms[stage * n_levels + level] += elapsed; returns the number of records consumed
    size_t
    read_stage_times(double *ms)
    {
      ctx->sync();
      for (size_t i = 0; i < stage_used; ++i)
        {
          float t = 0;
          HIP_CHECK(hipEventElapsedTime(&t, stage_records[i].e0, stage_records[i].e1));
          ms[(size_t)stage_records[i].stage * n_levels() + stage_records[i].level] += t;
        }
      const size_t n = stage_used;
      stage_used     = 0;
      return n;
    }
    void
    release_stage_records()
    {
      for (auto &r : stage_records)
        {
          (void)hipEventDestroy(r.e0);
          (void)hipEventDestroy(r.e1);
        }
      stage_records.clear();
    }
  };

  LevelOperatorBase *
  make_level_operator(Ctx *ctx, const mgamd_dofs *dofs, int type, std::shared_ptr<Comm> comm = nullptr);
  ChebyshevBase *
  make_chebyshev(LevelOperatorBase *op, unsigned degree, double smoothing_range, unsigned eig_cg_n_iterations);
  Transfer2Base *
  make_transfer2(LevelOperatorBase *fine, LevelOperatorBase *coarse);
  // nested != nullptr: the coarse problem is handed to `n_cycles` V-cycles of another multigrid whose finest level is
  // levels[0] (the geometric stand-in for the reference's Trilinos/PETSc AMG coarse solvers on large coarse levels)
  MultigridBase *
  make_multigrid(Ctx *ctx, unsigned n_levels, LevelOperatorBase *const *levels, Transfer2Base *const *transfers,
                 ChebyshevBase *const *smoothers, const std::string &coarse_solver, MultigridBase *nested = nullptr,
                 unsigned n_cycles = 1);
  void
  solve_cg(LevelOperatorBase &A, MultigridBase *M, mgamd_vec &x, const mgamd_vec &b, double reltol, double abstol, unsigned maxiter,
           unsigned &n_iterations, double &residual);
} // namespace mgamd

struct mgamd_level_op
{
  std::unique_ptr<mgamd::LevelOperatorBase> op;
};
struct mgamd_cheb
{
  std::unique_ptr<mgamd::ChebyshevBase> c;
};
struct mgamd_transfer2
{
  std::unique_ptr<mgamd::Transfer2Base> t;
};
struct mgamd_mg
{
  std::unique_ptr<mgamd::MultigridBase> mg;
};
