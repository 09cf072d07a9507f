// mgamd.hpp -- header-only C++ layer over the C ABI (include/mgamd.h) that re-exposes the MI355X path with
// the class and method names deal.II gives it in the reference, so that code written like
// ref:multigrid_throughput.cc:817-1666 (mg_solve / solve_with_global_coarsening) compiles against it:
//
//   mgamd::Triangulation          <-> parallel::distributed::Triangulation<3> + GridGenerator::create_*
//   mgamd::DoFHandler             <-> DoFHandler<3> + AffineConstraints + MatrixFree::reinit
//   mgamd::Vector                 <-> LinearAlgebra::distributed::Vector<Number>
//   mgamd::Operator               <-> Operator<3,1,Number>                 (ref:include/operator.h:11-557)
//   mgamd::PreconditionChebyshev  <-> PreconditionChebyshev<Operator,Vector,DiagonalMatrix<Vector>>
//   mgamd::MGTwoLevelTransfer     <-> MGTwoLevelTransfer<3,Vector>
//   mgamd::PreconditionMG         <-> Multigrid<Vector> + PreconditionMG + MGTransferGlobalCoarsening
//   mgamd::SolverCG / ReductionControl
//
// Status codes become exceptions (std::runtime_error), as AssertThrow does in the reference
// (ref:multigrid_throughput.cc:2444-2468 catches them in main).
#pragma once
#include "../../include/mgamd.h"

#include <array>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <thread>

#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mgamd
{
  inline void
  check(int status)
  {
    if (status != MGAMD_OK)
      throw std::runtime_error(mgamd_last_error());
  }

  class Triangulation
  {
  public:
    Triangulation(const std::string &geometry_type, unsigned n_ref_global, unsigned n_ref_local = 0)
    {
      mgamd_tria *t = nullptr;
      check(mgamd_tria_create(geometry_type.c_str(), n_ref_global, n_ref_local, &t));
      h.reset(t, mgamd_tria_destroy);
      query();
    }
    // a caller-built octree over one root cell (the active cells of a parallel::distributed::Triangulation as (level, i, j, k));
    // checked for exact tiling and full 2:1 balance (mgamd_tria_create_from_leaves)
    static std::shared_ptr<Triangulation>
    from_leaves(const std::vector<uint8_t> &level, const std::vector<uint32_t> &i, const std::vector<uint32_t> &j, const std::vector<uint32_t> &k)
    {
      if (level.size() != i.size() || level.size() != j.size() || level.size() != k.size())
        throw std::runtime_error("from_leaves: arrays of different lengths");
      mgamd_tria *t = nullptr;
      check(mgamd_tria_create_from_leaves(level.size(), level.data(), i.data(), j.data(), k.data(), &t));
      auto r = std::shared_ptr<Triangulation>(new Triangulation());
      r->h.reset(t, mgamd_tria_destroy);
      r->query();
      return r;
    }
    // one step of MGTransferGlobalCoarseningTools::create_geometric_coarsening_sequence
    std::shared_ptr<Triangulation>
    coarsen() const
    {
      mgamd_tria *t = nullptr;
      check(mgamd_tria_coarsen(h.get(), &t));
      auto r = std::shared_ptr<Triangulation>(new Triangulation());
      r->h.reset(t, mgamd_tria_destroy);
      r->query();
      return r;
    }
    // local smoothing: all cells of refinement level `level`, active or not (the level of DoFHandler::distribute_mg_dofs)
    std::shared_ptr<Triangulation>
    level_mesh(unsigned level) const
    {
      mgamd_tria *t = nullptr;
      check(mgamd_tria_level_mesh(h.get(), level, &t));
      auto r = std::shared_ptr<Triangulation>(new Triangulation());
      r->h.reset(t, mgamd_tria_destroy);
      r->query();
      return r;
    }
    uint64_t
    n_global_active_cells() const
    {
      return n_cells;
    }
    unsigned
    n_global_levels() const
    {
      return n_levels;
    }
    uint64_t
    n_cells_with_hanging_nodes() const
    {
      return n_cells_hn;
    }
    mgamd_tria *
    get() const
    {
      return h.get();
    }

  private:
    Triangulation() = default;
    void
    query()
    {
      check(mgamd_tria_info(h.get(), &n_cells, &n_levels, &n_cells_hn));
    }
    std::shared_ptr<mgamd_tria> h;
    uint64_t                    n_cells = 0, n_cells_hn = 0;
    uint32_t                    n_levels = 0;
  };

  // MGTransferGlobalCoarseningTools::create_geometric_coarsening_sequence (ref:multigrid_throughput.cc:2219-2224)
  inline std::vector<std::shared_ptr<const Triangulation>>
  create_geometric_coarsening_sequence(const std::shared_ptr<const Triangulation> &fine)
  {
    std::vector<std::shared_ptr<const Triangulation>> seq{fine};
    while (seq.back()->n_global_active_cells() > 1)
      seq.push_back(seq.back()->coarsen());
    return {seq.rbegin(), seq.rend()};
  }

  // MGTools::print_multigrid_statistics(triangulations) (ref:include/mg_tools.h:267-512, ref:multigrid_throughput.cc:1657-1665):
  // {workload_eff, workload_path_max, vertical_eff, horizontal_eff, mem_total} of the n_ranks-way partition of the level
  // meshes (coarse -> fine)
  inline std::vector<std::pair<std::string, double>>
  print_multigrid_statistics(const std::vector<std::shared_ptr<const Triangulation>> &triangulations, unsigned n_ranks = 1)
  {
    std::vector<const mgamd_tria *> t;
    for (const auto &x : triangulations)
      t.push_back(x->get());
    mgamd_partition *p = nullptr;
    check(mgamd_partition_create_ex(t.data(), (unsigned)t.size(), n_ranks, 2.0, 0, &p));
    double     st[5];
    const int  rc = mgamd_partition_statistics(p, st);
    mgamd_partition_destroy(p);
    check(rc);
    static const char *names[5] = {"workload_eff", "workload_path_max", "vertical_eff", "horizontal_eff", "mem_total"};
    std::vector<std::pair<std::string, double>> out;
    for (int i = 0; i < 5; ++i)
      out.emplace_back(names[i], st[i]);
    return out;
  }

  // ...::create_polynomial_coarsening_sequence(degree, bisect) (ref:multigrid_throughput.cc:1506-1510)
  inline std::vector<unsigned>
  create_polynomial_coarsening_sequence(unsigned degree)
  {
    std::vector<unsigned> seq{degree};
    while (seq.back() > 1)
      seq.push_back(std::max(seq.back() / 2, 1u));
    return {seq.rbegin(), seq.rend()};
  }

  // Domain decomposition of a level hierarchy, one rank per GPU (the reference partitions with p4est + RepartitioningPolicyTools,
  // ref:multigrid_throughput.cc:2066-2175): Morton cut of a root level into n_ranks weighted chunks; finer levels inherit it,
  // coarser levels are replicated.
  class Partition
  {
  public:
    // group > 1: two tiers -- levels below the root level with >= min_sub_root_cells cells are cut into n_ranks / group parts, each
    // held by `group` consecutive ranks (what the reference's agglomeration onto fewer processes is for,
    // ref:multigrid_throughput.cc:379-418,1464-1501); their operators take Communicator::subset(group)
    Partition(const std::vector<std::shared_ptr<const Triangulation>> &triangulations, unsigned n_ranks, double hanging_weight = 2.0,
              uint64_t min_root_cells = 0, unsigned group = 1, uint64_t min_sub_root_cells = 0)
      : triangulations(triangulations)
    {
      std::vector<const mgamd_tria *> t;
      for (const auto &x : triangulations)
        t.push_back(x->get());
      mgamd_partition *p = nullptr;
      check(mgamd_partition_create_tiered(t.data(), (unsigned)t.size(), n_ranks, hanging_weight, min_root_cells, group, min_sub_root_cells, &p));
      h.reset(p, mgamd_partition_destroy);
      check(mgamd_partition_info(h.get(), &root, &nr));
      check(mgamd_partition_tiers(h.get(), &sub_root, &grp));
    }
    unsigned
    root_level() const
    {
      return root;
    }
    unsigned
    sub_root_level() const // == root_level() without a subset tier
    {
      return sub_root;
    }
    unsigned
    group() const
    {
      return grp;
    }
    unsigned
    n_parts(unsigned level) const // pieces of the level: n_ranks, n_ranks / group on the subset tier, 1 = replicated
    {
      return level >= root ? nr : (level >= sub_root ? nr / grp : 1);
    }
    mgamd_partition *
    get() const
    {
      return h.get();
    }
    std::vector<std::shared_ptr<const Triangulation>> triangulations;

  private:
    std::shared_ptr<mgamd_partition> h;
    unsigned                         root = 0, sub_root = 0, grp = 1, nr = 1;
  };

  class DoFHandler
  {
  public:
    // mg_level = true: `tria` is Triangulation::level_mesh(l): the level DoFs of distribute_mg_dofs with the refinement-edge
    // set of MGConstrainedDoFs
    DoFHandler(const std::shared_ptr<const Triangulation> &tria, unsigned fe_degree, int max_brick = -1, bool mg_level = false)
      : tria(tria)
    {
      mgamd_dofs *d = nullptr;
      if (mg_level)
        check(mgamd_dofs_create_level(tria->get(), (int)fe_degree, max_brick, &d));
      else
        check(mgamd_dofs_create(tria->get(), (int)fe_degree, max_brick, &d));
      h.reset(d, mgamd_dofs_destroy);
      check(mgamd_dofs_info(h.get(), &info));
    }
    // one rank's share of level `level` of a partitioned hierarchy (mgamd_dofs_create_local): its cells + halo plan on a
    // distributed level, the whole level on a replicated one
    DoFHandler(const Partition &partition, unsigned level, unsigned rank, unsigned fe_degree, int max_brick = -1)
      : tria(partition.triangulations.at(level))
    {
      mgamd_dofs *d = nullptr;
      check(mgamd_dofs_create_local(partition.get(), level, rank, (int)fe_degree, max_brick, &d));
      h.reset(d, mgamd_dofs_destroy);
      check(mgamd_dofs_info(h.get(), &info));
    }
    uint64_t
    n_dofs() const
    {
      return info.n_dofs;
    }
    const Triangulation &
    get_triangulation() const
    {
      return *tria;
    }
    mgamd_dofs *
    get() const
    {
      return h.get();
    }
    mgamd_dofs_info_t info;

  private:
    std::shared_ptr<const Triangulation> tria;
    std::shared_ptr<mgamd_dofs>          h;
  };

  class Context
  {
  public:
    explicit Context(int device = 0)
    {
      mgamd_ctx *c = nullptr;
      check(mgamd_ctx_create(device, &c));
      h.reset(c, mgamd_ctx_destroy);
    }
    void
    synchronize() const
    {
      check(mgamd_ctx_synchronize(h.get()));
    }
    mgamd_ctx *
    get() const
    {
      return h.get();
    }

  private:
    std::shared_ptr<mgamd_ctx> h;
  };

  // The communicator of a sharded run: RCCL over xGMI, one rank per GPU (what MPI_COMM_WORLD is to the reference,
  // ref:multigrid_throughput.cc:2403-2442).  Every rank calls rccl() with the same 128-byte id; exchange_id_through_file is the
  // host-side channel for launchers that only provide RANK / WORLD_SIZE (torchrun --no-python, a shell loop): rank 0 writes the
  // id, the others wait for the file.
  class Communicator
  {
  public:
    Communicator() = default;
    static std::array<char, 128>
    exchange_id_through_file(const std::string &path, unsigned rank, double timeout_s = 120.0)
    {
      std::array<char, 128> id{};
      if (rank == 0)
        {
          check(mgamd_comm_rccl_unique_id(id.data()));
          const std::string tmp = path + ".tmp";
          {
            std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
            f.write(id.data(), 128);
          }
          if (std::rename(tmp.c_str(), path.c_str()) != 0)
            throw std::runtime_error("cannot publish the RCCL id at " + path);
          return id;
        }
      const auto t0 = std::chrono::steady_clock::now();
      for (;;)
        {
          std::ifstream f(path, std::ios::binary);
          if (f && f.read(id.data(), 128) && f.gcount() == 128)
            return id;
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
            throw std::runtime_error("timed out waiting for the RCCL id at " + path);
          std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    static Communicator
    rccl(const Context &ctx, unsigned n_ranks, unsigned rank, const std::array<char, 128> &id)
    {
      Communicator c;
      mgamd_comm  *m = nullptr;
      check(mgamd_comm_rccl_create(ctx.get(), n_ranks, rank, id.data(), &m));
      c.h.reset(m, mgamd_comm_destroy);
      c.n = n_ranks;
      c.r = rank;
      return c;
    }
    // the communicator of the levels that are cut into n_ranks / group parts (Partition tiers): rank = part
    Communicator
    subset(unsigned group) const
    {
      if (group <= 1)
        return *this;
      Communicator c;
      mgamd_comm  *m = nullptr;
      check(mgamd_comm_subset(h.get(), group, &m));
      const std::shared_ptr<mgamd_comm> base = h; // (the C object keeps the base alive as well)
      c.h.reset(m, [base](mgamd_comm *x) { mgamd_comm_destroy(x); });
      c.n = n / group;
      c.r = r / group;
      return c;
    }
    double
    allreduce_sum(const Context &ctx, double v) const // Utilities::MPI::sum
    {
      double s = 0;
      check(mgamd_comm_allreduce_sum(h.get(), ctx.get(), v, &s));
      return s;
    }
    unsigned
    n_ranks() const
    {
      return n;
    }
    unsigned
    rank() const
    {
      return r;
    }
    mgamd_comm *
    get() const
    {
      return h.get();
    }

  private:
    std::shared_ptr<mgamd_comm> h;
    unsigned                    n = 1, r = 0;
  };

  class Vector
  {
  public:
    Vector() = default;
    explicit Vector(mgamd_vec *v)
    {
      h.reset(v, mgamd_vec_destroy);
    }
    Vector(const Context &ctx, uint64_t n, int number_type = MGAMD_F64)
    {
      mgamd_vec *v = nullptr;
      check(mgamd_vec_create(ctx.get(), n, number_type, &v));
      h.reset(v, mgamd_vec_destroy);
    }
    uint64_t
    size() const
    {
      uint64_t n = 0;
      check(mgamd_vec_size(h.get(), &n));
      return n;
    }
    Vector &
    operator=(double value)
    {
      check(mgamd_vec_set(h.get(), value));
      return *this;
    }
    void
    sadd(double s, double a, const Vector &x)
    {
      check(mgamd_vec_sadd(h.get(), s, a, x.get()));
    }
    void
    add(double a, const Vector &x)
    {
      check(mgamd_vec_axpy(h.get(), a, x.get()));
    }
    double
    operator*(const Vector &other) const
    {
      double r = 0;
      check(mgamd_vec_dot(h.get(), other.get(), &r));
      return r;
    }
    double
    l2_norm() const
    {
      double r = 0;
      check(mgamd_vec_norm2(h.get(), &r));
      return r;
    }
    void
    copy_from_host(const std::vector<double> &v)
    {
      check(mgamd_vec_from_host(h.get(), v.data()));
    }
    std::vector<double>
    copy_to_host() const
    {
      std::vector<double> v(size());
      check(mgamd_vec_to_host(h.get(), v.data()));
      return v;
    }
    mgamd_vec *
    get() const
    {
      return h.get();
    }

  private:
    std::shared_ptr<mgamd_vec> h;
  };

  // Operator<dim=3, n_components=1, Number> (ref:include/operator.h:11)
  class Operator
  {
  public:
    using VectorType = Vector;
    // Operator::reinit(mapping, dof_handler, quad, constraints) (ref:include/operator.h:24-47): mapping is
    // MappingQ1 on cubes, the quadrature QGauss(degree+1), constraints zero Dirichlet + hanging nodes.
    void
    reinit(const Context &ctx, const DoFHandler &dof_handler, int number_type = MGAMD_F64)
    {
      mgamd_level_op *o = nullptr;
      check(mgamd_level_op_create(ctx.get(), dof_handler.get(), number_type, &o));
      h.reset(o, mgamd_level_op_destroy);
    }
    // a level of a sharded hierarchy (DoFHandler(partition, level, rank, ...)): sums of shared DoFs are completed through `comm`
    // (nullptr: replicated level)
    void
    reinit(const Context &ctx, const DoFHandler &dof_handler, int number_type, const Communicator *comm)
    {
      mgamd_level_op *o = nullptr;
      check(mgamd_level_op_create_distributed(ctx.get(), dof_handler.get(), number_type, comm ? comm->get() : nullptr, &o));
      h.reset(o, mgamd_level_op_destroy);
    }
    // DoFs this rank owns (sums to DoFHandler::n_dofs() over the ranks)
    uint64_t
    n_owned() const
    {
      uint64_t n = 0;
      check(mgamd_level_op_n_owned(h.get(), &n));
      return n;
    }
    uint64_t
    m() const
    {
      uint64_t n = 0;
      check(mgamd_level_op_m(h.get(), &n));
      return n;
    }
    void
    initialize_dof_vector(Vector &vec) const
    {
      mgamd_vec *v = nullptr;
      check(mgamd_level_op_init_vector(h.get(), &v));
      vec = Vector(v);
    }
    void
    vmult(Vector &dst, const Vector &src) const
    {
      check(mgamd_level_op_vmult(h.get(), dst.get(), src.get()));
    }
    void
    Tvmult(Vector &dst, const Vector &src) const
    {
      vmult(dst, src); // ref:include/operator.h:185-189
    }
    // ref:include/operator.h:191-201 and :203-226 (what MatrixFreeOperators::MGInterfaceOperator::vmult / Tvmult forward to)
    void
    vmult_interface_down(Vector &dst, const Vector &src) const
    {
      check(mgamd_level_op_vmult_interface_down(h.get(), dst.get(), src.get()));
    }
    void
    vmult_interface_up(Vector &dst, const Vector &src) const
    {
      check(mgamd_level_op_vmult_interface_up(h.get(), dst.get(), src.get()));
    }
    void
    compute_inverse_diagonal(Vector &diagonal) const
    {
      if (!diagonal.get())
        initialize_dof_vector(diagonal);
      check(mgamd_level_op_inverse_diagonal(h.get(), diagonal.get()));
    }
    // simulation_kind: 0 "Constant" (f = 1, g = 0), 1 "Gaussian" (ref:multigrid_throughput.cc:2286-2298)
    void
    rhs(Vector &system_rhs, int simulation_kind = 0) const
    {
      check(mgamd_level_op_rhs_kind(h.get(), simulation_kind, system_rhs.get()));
    }
    // constraints.distribute(solution)
    void
    distribute(Vector &solution, int simulation_kind = 0) const
    {
      check(mgamd_level_op_distribute(h.get(), simulation_kind, solution.get()));
    }
    mgamd_level_op *
    get() const
    {
      return h.get();
    }

  private:
    std::shared_ptr<mgamd_level_op> h;
  };

  class PreconditionChebyshev
  {
  public:
    struct AdditionalData
    {
      double   smoothing_range     = 20.;
      unsigned degree              = 5;
      unsigned eig_cg_n_iterations = 20;
    };
    void
    initialize(const Operator &matrix, const AdditionalData &data)
    {
      mgamd_cheb *c = nullptr;
      check(mgamd_cheb_create(matrix.get(), data.degree, data.smoothing_range, data.eig_cg_n_iterations, &c));
      h.reset(c, mgamd_cheb_destroy);
    }
    void
    vmult(Vector &dst, const Vector &src) const
    {
      check(mgamd_cheb_vmult(h.get(), dst.get(), src.get()));
    }
    void
    step(Vector &dst, const Vector &src) const
    {
      check(mgamd_cheb_step(h.get(), dst.get(), src.get()));
    }
    mgamd_cheb *
    get() const
    {
      return h.get();
    }

  private:
    std::shared_ptr<mgamd_cheb> h;
  };

  class MGTwoLevelTransfer
  {
  public:
    // reinit(dof_fine, dof_coarse, constraint_fine, constraint_coarse): the level operators carry all four
    void
    reinit(const Operator &fine, const Operator &coarse)
    {
      mgamd_transfer2 *t = nullptr;
      check(mgamd_transfer2_create(fine.get(), coarse.get(), &t));
      h.reset(t, mgamd_transfer2_destroy);
    }
    void
    prolongate_and_add(Vector &dst, const Vector &src) const
    {
      check(mgamd_transfer2_prolongate_and_add(h.get(), dst.get(), src.get()));
    }
    void
    restrict_and_add(Vector &dst, const Vector &src) const
    {
      check(mgamd_transfer2_restrict_and_add(h.get(), dst.get(), src.get()));
    }
    mgamd_transfer2 *
    get() const
    {
      return h.get();
    }

  private:
    std::shared_ptr<mgamd_transfer2> h;
  };

  // Multigrid<VectorType> + PreconditionMG<dim,VectorType,MGTransferGlobalCoarsening> in one object
  class PreconditionMG
  {
  public:
    using StageSlot = std::function<void(bool /*start*/, unsigned /*level*/)>;
    // active_mesh_dofs: local smoothing (solve_with_local_smoothing, ref:multigrid_throughput.cc:1670-1873): mg_matrices are the
    // operators on the refinement levels (DoFHandler with mg_level = true), vmult acts on vectors of the active mesh.
    // coarse_mg: the geometric stand-in for the AMG coarse solvers on a large coarse level (an h-multigrid whose finest level
    // is mg_matrices[0]), applied n_cycles times per coarse solve
    PreconditionMG(const Context &ctx, const std::vector<Operator> &mg_matrices, const std::vector<MGTwoLevelTransfer> &transfers,
                   const std::vector<PreconditionChebyshev> &smoothers, const std::string &coarse_grid_solver_type,
                   const PreconditionMG *coarse_mg = nullptr, unsigned n_cycles = 1, const DoFHandler *active_mesh_dofs = nullptr)
    {
      const unsigned                 n = mg_matrices.size();
      std::vector<mgamd_level_op *>  L(n);
      std::vector<mgamd_transfer2 *> T(n, nullptr);
      std::vector<mgamd_cheb *>      S(n);
      for (unsigned l = 0; l < n; ++l)
        {
          L[l] = mg_matrices[l].get();
          S[l] = smoothers[l].get();
          if (l > 0)
            T[l] = transfers[l].get();
        }
      mgamd_mg *m = nullptr;
      if (active_mesh_dofs)
        check(mgamd_mg_create_local_smoothing(ctx.get(), n, L.data(), T.data(), S.data(), active_mesh_dofs->get(),
                                              coarse_grid_solver_type.c_str(), &m));
      else if (coarse_mg)
        {
          check(mgamd_mg_create_nested(ctx.get(), n, L.data(), T.data(), S.data(), coarse_grid_solver_type.c_str(), coarse_mg->get(),
                                       n_cycles, &m));
          nested = coarse_mg->h; // keep the nested hierarchy's handle alive
        }
      else // (n_cycles = CoarseSolverNCycles of the algebraic coarse solvers "amg" / "cg_with_amg")
        check(mgamd_mg_create_nested(ctx.get(), n, L.data(), T.data(), S.data(), coarse_grid_solver_type.c_str(), nullptr, n_cycles, &m));
      h.reset(m, mgamd_mg_destroy);
      slots.resize(9);
      n_levels = n;
    }
    // the coarse solver that actually runs: "direct" | "cg" | "cg_with_chebyshev" | "amg" | "cg_with_amg" | "gmg_vcycle" (mgamd.h)
    std::string
    coarse_solver_used() const
    {
      char name[32];
      check(mgamd_mg_coarse_solver_used(h.get(), name));
      return name;
    }
    // stage times of the UNCHANGED cycle from HIP events (no host synchronisation inside the cycle): enable, run solves,
    // then read_stage_times() adds seconds to t[stage][level] (stages as in the connect_* slots, 7/8 = transfer to mg/global)
    void
    enable_stage_timing(bool on) const
    {
      check(mgamd_mg_stage_timing(h.get(), on ? 1 : 0));
    }
    void
    read_stage_times(std::vector<std::vector<double>> &t) const
    {
      std::vector<double> ms(9 * (size_t)n_levels, 0.0);
      uint64_t            n = 0;
      check(mgamd_mg_stage_times(h.get(), ms.data(), n_levels, &n));
      t.assign(9, std::vector<double>(n_levels, 0.0));
      for (unsigned st = 0; st < 9; ++st)
        for (unsigned l = 0; l < n_levels; ++l)
          t[st][l] = ms[st * (size_t)n_levels + l] * 1e-3;
    }
    void
    vmult(Vector &dst, const Vector &src) const
    {
      check(mgamd_mg_vcycle(h.get(), dst.get(), src.get()));
    }
    // Multigrid::connect_* / PreconditionMG::connect_transfer_to_* (ref:multigrid_throughput.cc:1183-1192,1233-1234)
    void
    connect_pre_smoother_step(StageSlot s)
    {
      connect(0, std::move(s));
    }
    void
    connect_residual_step(StageSlot s)
    {
      connect(1, std::move(s));
    }
    void
    connect_restriction(StageSlot s)
    {
      connect(2, std::move(s));
    }
    void
    connect_coarse_solve(StageSlot s)
    {
      connect(3, std::move(s));
    }
    void
    connect_prolongation(StageSlot s)
    {
      connect(4, std::move(s));
    }
    void
    connect_edge_prolongation(StageSlot s)
    {
      connect(5, std::move(s));
    }
    void
    connect_post_smoother_step(StageSlot s)
    {
      connect(6, std::move(s));
    }
    void
    connect_transfer_to_mg(std::function<void(bool)> s)
    {
      connect(7, [s](bool f, unsigned) { s(f); });
    }
    void
    connect_transfer_to_global(std::function<void(bool)> s)
    {
      connect(8, [s](bool f, unsigned) { s(f); });
    }
    void
    disconnect_all()
    {
      for (auto &s : slots)
        s = nullptr;
      check(mgamd_mg_set_stage_callback(h.get(), nullptr, nullptr));
    }
    mgamd_mg *
    get() const
    {
      return h.get();
    }

  private:
    void
    connect(int stage, StageSlot s)
    {
      slots[stage] = std::move(s);
      check(mgamd_mg_set_stage_callback(h.get(), &PreconditionMG::trampoline, this));
    }
    static void
    trampoline(int stage, int start, unsigned level, void *user)
    {
      auto *self = static_cast<PreconditionMG *>(user);
      if (stage >= 0 && stage < (int)self->slots.size() && self->slots[stage])
        self->slots[stage](start != 0, level);
    }
    std::shared_ptr<mgamd_mg> h, nested;
    std::vector<StageSlot>    slots;
    unsigned                  n_levels = 0;
  };

  class ReductionControl
  {
  public:
    ReductionControl(unsigned maxiter = 10000, double abstol = 1e-20, double reltol = 1e-4)
      : maxiter(maxiter)
      , abstol(abstol)
      , reltol(reltol)
    {}
    unsigned
    last_step() const
    {
      return steps;
    }
    double
    last_value() const
    {
      return value;
    }
    unsigned maxiter;
    double   abstol, reltol;
    unsigned steps = 0;
    double   value = 0;
  };

  class SolverCG
  {
  public:
    explicit SolverCG(ReductionControl &control)
      : control(control)
    {}
    // solve(A, x, b, preconditioner) starting from x = 0; non-convergence is reported through last_step()
    // == maxiter like the reference, which swallows SolverControl::NoConvergence (ref:multigrid_throughput.cc:1146,1250)
    void
    solve(const Operator &A, Vector &x, const Vector &b, const PreconditionMG &preconditioner)
    {
      check(mgamd_solve_cg(A.get(), preconditioner.get(), x.get(), b.get(), control.reltol, control.abstol, control.maxiter,
                           &control.steps, &control.value));
    }

  private:
    ReductionControl &control;
  };
} // namespace mgamd
