// multigrid_throughput -- JSON-in / table-out harness of the MI355X-native path, mirroring the reference's
// driver (ref:multigrid_throughput.cc:1970-2470): same 17 JSON keys (RunParameters::parse, :1989-2014), same
// run protocol (1 warm-up solve, then n_repetitions = 5 timed solves, best time wins, :1140-1147,1238-1268),
// same table columns in the same order (:2328-2335, 1488, 1278-1283, 1381-1401).
//
//   ./multigrid_throughput input_0000.json [input_0001.json ...]
//
// Implemented `Type`s: HMG-global, PMG, HPMG (global coarsening), HMG-local and HPMG-local (local smoothing).  AMG/AMGPETSc
// raise "not implemented" exactly like the reference's AssertThrow(false, ExcNotImplemented()) for unknown strings.
#include "../csrc/mgamd.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>

using namespace mgamd;

// ---- ScopedTimer (ref:include/scoped_timer.h:1-20) ---------------------------------------------------------
class ScopedTimer
{
public:
  explicit ScopedTimer(double &result)
    : result(result)
    , start(std::chrono::system_clock::now())
  {}
  ~ScopedTimer()
  {
    result += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now() - start).count() / 1e9;
  }

private:
  double                                            &result;
  std::chrono::time_point<std::chrono::system_clock> start;
};

// ---- minimal JSON object reader (flat object; values: string | number | true | false) ----------------------
static std::map<std::string, std::string>
parse_json_object(const std::string &file_name)
{
  std::ifstream f(file_name);
  if (!f)
    throw std::runtime_error("cannot open " + file_name);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string                  s = ss.str();
  std::map<std::string, std::string> out;
  size_t                             i = 0;
  auto skip = [&]() {
    while (i < s.size() && (isspace((unsigned char)s[i]) || s[i] == ',' || s[i] == ':'))
      ++i;
  };
  auto read_string = [&]() {
    std::string r;
    ++i;
    while (i < s.size() && s[i] != '"')
      {
        if (s[i] == '\\' && i + 1 < s.size())
          ++i;
        r += s[i++];
      }
    ++i;
    return r;
  };
  skip();
  if (i >= s.size() || s[i] != '{')
    throw std::runtime_error(file_name + ": not a JSON object");
  ++i;
  while (true)
    {
      skip();
      if (i >= s.size() || s[i] == '}')
        break;
      if (s[i] != '"')
        throw std::runtime_error(file_name + ": expected a key");
      const std::string key = read_string();
      skip();
      std::string value;
      if (s[i] == '"')
        value = read_string();
      else
        while (i < s.size() && s[i] != ',' && s[i] != '}' && !isspace((unsigned char)s[i]))
          value += s[i++];
      out[key] = value;
    }
  return out;
}

// ---- parameters (ref:multigrid_throughput.cc:297-334, 1970-2015) -------------------------------------------
struct MultigridParameters
{
  struct
  {
    std::string type    = "amg";
    unsigned    maxiter = 10000;
    double      abstol  = 1e-20;
    double      reltol  = 1e-4;
    unsigned    n_cycles = 1;
  } coarse_solver;
  struct
  {
    double   smoothing_range     = 20;
    unsigned degree              = 5;
    unsigned eig_cg_n_iterations = 20;
  } smoother;
  struct
  {
    unsigned maxiter = 10000;
    double   abstol  = 1e-20;
    double   reltol  = 1e-4;
  } cg_normal;
  unsigned n_repetitions = 5;
};

struct RunParameters
{
  std::string  type            = "PMG";
  std::string  geometry_type   = "quadrant_flexible";
  unsigned     n_ref_global    = 6;
  unsigned     n_ref_local     = 0;
  unsigned     fe_degree_fine  = 4;
  bool         paraview        = false;
  bool         verbose         = true;
  unsigned     p               = 0;
  std::string  policy_name     = "";
  std::string  mg_number_type  = "float";
  std::string  simulation_type = "Constant";
  int          min_level       = -1;
  int          min_n_cells     = -1;
  MultigridParameters mg_data;

  void
  parse(const std::string &file_name)
  {
    const auto kv  = parse_json_object(file_name);
    auto       get = [&](const char *k, auto &dst) {
      auto it = kv.find(k);
      if (it == kv.end())
        return;
      using D = std::decay_t<decltype(dst)>;
      if constexpr (std::is_same<D, std::string>::value)
        dst = it->second;
      else if constexpr (std::is_same<D, bool>::value)
        dst = it->second == "true" || it->second == "1";
      else if constexpr (std::is_floating_point<D>::value)
        dst = std::stod(it->second);
      else
        dst = (D)std::stol(it->second);
    };
    get("Type", type);
    get("GeometryType", geometry_type);
    get("NRefGlobal", n_ref_global);
    get("NRefLocal", n_ref_local);
    get("Degree", fe_degree_fine);
    get("Paraview", paraview);
    get("Verbosity", verbose);
    get("Partitioner", p);
    get("PartitionerName", policy_name);
    get("MinLevel", min_level);
    get("MinNCells", min_n_cells);
    get("CoarseGridSolverType", mg_data.coarse_solver.type);
    get("SmootherDegree", mg_data.smoother.degree);
    get("CoarseSolverNCycles", mg_data.coarse_solver.n_cycles);
    get("RelativeTolerance", mg_data.cg_normal.reltol);
    get("MGNumberType", mg_number_type);
    get("SimulationType", simulation_type); // unknown keys are ignored (skip_undefined = true)
  }
};

// ---- ConvergenceTable stand-in -------------------------------------------------------------------------------
class ConvergenceTable
{
public:
  template <typename V>
  void
  add_value(const std::string &key, const V &value, bool scientific = false)
  {
    std::ostringstream os;
    if constexpr (std::is_arithmetic<V>::value)
      {
        if (scientific)
          os << std::scientific << std::setprecision(4) << (double)value;
        else
          os << value;
      }
    else
      os << value;
    if (std::find(order.begin(), order.end(), key) == order.end())
      order.push_back(key);
    columns[key].push_back(os.str());
  }
  void
  write_text(std::ostream &out) const
  {
    std::vector<size_t> w;
    size_t              rows = 0;
    for (const auto &k : order)
      {
        size_t m = k.size();
        for (const auto &v : columns.at(k))
          m = std::max(m, v.size());
        w.push_back(m);
        rows = std::max(rows, columns.at(k).size());
      }
    for (size_t c = 0; c < order.size(); ++c)
      out << std::left << std::setw(w[c] + 1) << order[c];
    out << "\n";
    for (size_t r = 0; r < rows; ++r)
      {
        for (size_t c = 0; c < order.size(); ++c)
          {
            const auto &col = columns.at(order[c]);
            out << std::left << std::setw(w[c] + 1) << (r < col.size() ? col[r] : "");
          }
        out << "\n";
      }
    out << std::flush;
  }

private:
  std::vector<std::string>                        order;
  std::map<std::string, std::vector<std::string>> columns;
};

static std::string
resolve_policy_name(const RunParameters &params)
{
  // ref:multigrid_throughput.cc:2074-2105
  if (!params.policy_name.empty())
    return params.policy_name;
  static const char *names[] = {"DefaultPolicy", "MinimalGranularityPolicy-40", "CellWeightPolicy-1.0", "CellWeightPolicy-1.5",
                                "CellWeightPolicy-2.0", "CellWeightPolicy-2.5", "FirstChildPolicy", "BalancedGranularityPartitionPolicy"};
  if (params.p > 7)
    throw std::runtime_error("Partitioner: not implemented");
  return names[params.p];
}

// comm != nullptr: the run is SHARDED over comm->n_ranks() GPUs, one rank per GPU (the reference's MPI ranks,
// ref:multigrid_throughput.cc:2403-2442): spatial domain decomposition of the level hierarchy (mgamd.hpp Partition), halo
// exchange and coarse-level all-reduce over RCCL inside the library.  Implemented for Type "HMG-global" (BASELINE configs[3]).
static void
run(const Context &ctx, const RunParameters &params, ConvergenceTable &table, const Communicator *comm = nullptr)
{
  const std::string policy = resolve_policy_name(params);
  {
    // the partitioning policy only matters across ranks; validate the name like ref:multigrid_throughput.cc:2127-2174
    auto pre = [&](const char *x) { return policy.rfind(x, 0) == 0; };
    if (!(policy == "DefaultPolicy" || policy == "BalancedGranularityPartitionPolicy" || pre("MinimalGranularityPolicy") ||
          pre("CellWeightPolicy") || pre("FirstChildPolicy")))
      throw std::runtime_error("PartitionerName '" + policy + "': not implemented");
  }
  int simulation_kind; // ref:multigrid_throughput.cc:2286-2300
  if (params.simulation_type == "Constant")
    simulation_kind = 0;
  else if (params.simulation_type == "Gaussian")
    simulation_kind = 1;
  else
    throw std::runtime_error("SimulationType '" + params.simulation_type + "': not implemented");
  int level_number_type;
  if (params.mg_number_type == "double")
    level_number_type = MGAMD_F64;
  else if (params.mg_number_type == "float")
    level_number_type = MGAMD_F32;
  else
    throw std::runtime_error("MGNumberType '" + params.mg_number_type + "': not implemented");

  auto tria = std::make_shared<const Triangulation>(params.geometry_type, params.n_ref_global, params.n_ref_local);

  // ---- level hierarchy (ref:multigrid_throughput.cc:2219-2260, 1506-1596)
  std::vector<std::shared_ptr<const Triangulation>> triangulations;
  std::vector<unsigned>                             degrees;
  if (params.type == "HMG-global")
    {
      triangulations = create_geometric_coarsening_sequence(tria);
      if (triangulations.size() > 1)
        {
          auto ptr = std::find_if(triangulations.begin(), triangulations.end() - 1, [&](const auto &t) {
            if (params.min_level != -1)
              return params.min_level <= (int)t->n_global_levels();
            if (params.min_n_cells != -1)
              return (int)t->n_global_active_cells() >= params.min_n_cells;
            return true;
          });
          triangulations.erase(triangulations.begin(), ptr);
        }
      degrees.assign(triangulations.size(), params.fe_degree_fine);
    }
  else if (params.type == "PMG")
    {
      degrees = create_polynomial_coarsening_sequence(params.fe_degree_fine);
      triangulations.assign(degrees.size(), tria);
    }
  else if (params.type == "HPMG")
    {
      // ref:multigrid_throughput.cc:1518-1519,1551-1553,1569-1571: h-levels at the lowest degree, then p-levels
      const auto pseq = create_polynomial_coarsening_sequence(params.fe_degree_fine);
      triangulations  = create_geometric_coarsening_sequence(tria);
      degrees.assign(triangulations.size(), pseq.front());
      for (size_t i = 1; i < pseq.size(); ++i)
        {
          triangulations.push_back(tria);
          degrees.push_back(pseq[i]);
        }
    }
  else if (params.type == "HPMG-local")
    {
      // ref:multigrid_throughput.cc:1685-1695,1846-1860: p-multigrid on the active mesh; its coarse problem (lowest degree) is
      // handed to one local-smoothing V-cycle, built below
      degrees = create_polynomial_coarsening_sequence(params.fe_degree_fine);
      triangulations.assign(degrees.size(), tria);
    }
  else if (params.type == "HMG-local")
    {
      // solve_with_local_smoothing (ref:multigrid_throughput.cc:1670-1873): the levels are the refinement levels of the mesh
      for (unsigned l = 0; l < tria->n_global_levels(); ++l)
        triangulations.push_back(tria->level_mesh(l));
      degrees.assign(triangulations.size(), params.fe_degree_fine);
    }
  else
    throw std::runtime_error("Type '" + params.type + "': not implemented");
  const bool local_smoothing = params.type == "HMG-local";
  if (comm && !(params.type == "HMG-global" || params.type == "PMG" || params.type == "HPMG"))
    throw std::runtime_error("sharded harness: Type '" + params.type + "' is not implemented (HMG-global, PMG and HPMG are)");
  // the geometric meshes the partition is built on, and the mesh of every multigrid level: the p-levels of PMG / HPMG live on the
  // finest mesh and inherit its partition (ref:multigrid_throughput.cc:1506-1571)
  std::vector<std::shared_ptr<const Triangulation>> mesh_sequence = triangulations;
  std::vector<unsigned>                             mesh_index(triangulations.size());
  for (unsigned l = 0; l < mesh_index.size(); ++l)
    mesh_index[l] = l;
  if (comm && (params.type == "PMG" || params.type == "HPMG"))
    {
      mesh_sequence = create_geometric_coarsening_sequence(tria);
      for (unsigned l = 0; l < mesh_index.size(); ++l)
        mesh_index[l] = params.type == "PMG" ? (unsigned)mesh_sequence.size() - 1 : std::min<unsigned>(l, (unsigned)mesh_sequence.size() - 1);
    }
  // levels of >= ~4 M DoFs are cut into one chunk per rank, those of >= ~1 M DoFs into n_ranks / group parts that a group of ranks
  // holds together (groups of 4 from 8 ranks on, of 2 from 4 on), the others are replicated (the counterpart
  // of the reference's min_level / min_n_cells_per_process agglomeration, ref:multigrid_throughput.cc:379-418,1464-1501)
  std::unique_ptr<Partition> partition;
  Communicator               sub_comm;
  if (comm)
    {
      const uint64_t p_low = *std::min_element(degrees.begin(), degrees.end());
      const uint64_t p3    = p_low * p_low * p_low;
      const unsigned nr    = comm->n_ranks();
      const unsigned group = (nr >= 8 && nr % 4 == 0) ? 4 : ((nr >= 4 && nr % 2 == 0) ? 2 : 1);
      partition            = std::make_unique<Partition>(mesh_sequence, nr, 2.0, (uint64_t)4000000 / p3, group, (uint64_t)1000000 / p3);
      sub_comm             = comm->subset(partition->group());
    }
  // (by mesh index: what the partition knows)
  auto mesh_distributed = [&](unsigned mi) { return comm && comm->n_ranks() > 1 && mi >= partition->sub_root_level(); };
  auto mesh_comm        = [&](unsigned mi) -> const Communicator        *{ return mi >= partition->root_level() ? comm : &sub_comm; };
  auto distributed      = [&](unsigned l) { return mesh_distributed(mesh_index[l]); };
  auto level_comm       = [&](unsigned l) { return mesh_comm(mesh_index[l]); };

  const bool hp_local        = params.type == "HPMG-local";
  PreconditionChebyshev::AdditionalData sd;
  sd.smoothing_range     = params.mg_data.smoother.smoothing_range;
  sd.degree              = params.mg_data.smoother.degree;
  sd.eig_cg_n_iterations = params.mg_data.smoother.eig_cg_n_iterations;
  // HPMG-local: the local-smoothing hierarchy of the lowest degree (the coarse solver of the p-levels)
  std::vector<DoFHandler>            ls_dof_handlers;
  std::vector<Operator>              ls_operators;
  std::vector<MGTwoLevelTransfer>    ls_transfers;
  std::vector<PreconditionChebyshev> ls_smoothers;
  std::unique_ptr<DoFHandler>        ls_active;
  std::unique_ptr<PreconditionMG>    ls_mg;
  if (hp_local)
    {
      const unsigned nls = tria->n_global_levels();
      ls_operators.resize(nls);
      ls_transfers.resize(nls);
      ls_smoothers.resize(nls);
      for (unsigned l = 0; l < nls; ++l)
        ls_dof_handlers.emplace_back(tria->level_mesh(l), degrees.front(), -1, true);
      ls_active = std::make_unique<DoFHandler>(tria, degrees.front());
      for (unsigned l = 0; l < nls; ++l)
        ls_operators[l].reinit(ctx, ls_dof_handlers[l], level_number_type);
      for (unsigned l = 1; l < nls; ++l)
        ls_transfers[l].reinit(ls_operators[l], ls_operators[l - 1]);
      for (unsigned l = 0; l < nls; ++l)
        ls_smoothers[l].initialize(ls_operators[l], sd);
      ls_mg = std::make_unique<PreconditionMG>(ctx, ls_operators, ls_transfers, ls_smoothers, params.mg_data.coarse_solver.type, nullptr, 1,
                                               ls_active.get());
    }

  const unsigned                  n_levels = degrees.size();
  std::vector<DoFHandler>         dof_handlers;
  std::vector<Operator>           operators(n_levels);
  std::vector<MGTwoLevelTransfer> transfers(n_levels);
  std::vector<PreconditionChebyshev> smoothers(n_levels);
  for (unsigned l = 0; l < n_levels; ++l)
    if (hp_local && l == 0)
      dof_handlers.push_back(*ls_active); // the SAME DoFs as the local-smoothing cycle acts on
    else if (comm)
      dof_handlers.emplace_back(*partition, mesh_index[l], comm->rank(), degrees[l]);
    else
      dof_handlers.emplace_back(triangulations[l], degrees[l], -1, local_smoothing);
  std::unique_ptr<DoFHandler> active_dof_handler;
  if (local_smoothing)
    active_dof_handler = std::make_unique<DoFHandler>(tria, params.fe_degree_fine);
  const DoFHandler &fine_dof_handler = local_smoothing ? *active_dof_handler : dof_handlers.back();
  for (unsigned l = 0; l < n_levels; ++l)
    if (comm)
      operators[l].reinit(ctx, dof_handlers[l], level_number_type, distributed(l) ? level_comm(l) : nullptr);
    else
      operators[l].reinit(ctx, dof_handlers[l], level_number_type);
  for (unsigned l = 1; l < n_levels; ++l)
    transfers[l].reinit(operators[l], operators[l - 1]);
  for (unsigned l = 0; l < n_levels; ++l)
    smoothers[l].initialize(operators[l], sd);

  // coarse solver (library policy, include/mgamd.h): the Trilinos/PETSc AMG options are an exact solve on the one-cell coarse
  // level of global coarsening and, on a large coarse level (PMG, MinLevel), the library's own smoothed-aggregation AMG.
  // CoarseGridSolverType "gmg_vcycle" (this project's extension) selects the geometric stand-in of rounds 1-2 instead: V-cycles
  // of the h-multigrid on that level.  The table says what ran in its `coarse_solver` column.
  const std::string coarse = hp_local ? std::string("gmg_vcycle") : params.mg_data.coarse_solver.type;
  // (a SHARDED coarse level takes the geometric stand-in for the AMG choices too: the AMG is built from one rank's matrix)
  const bool amg_name = coarse == "amg" || coarse == "cg_with_amg" || coarse == "amg_petsc";
  const bool amg_like = !hp_local && (coarse == "gmg_vcycle" || (comm && distributed(0) && amg_name));
  std::vector<DoFHandler>            c_dof_handlers;
  std::vector<Operator>              c_operators;
  std::vector<MGTwoLevelTransfer>    c_transfers;
  std::vector<PreconditionChebyshev> c_smoothers;
  std::unique_ptr<PreconditionMG>    coarse_mg;
  if (amg_like && dof_handlers[0].n_dofs() > 4096)
    {
      const auto c_trias = create_geometric_coarsening_sequence(triangulations[0]);
      const unsigned nc  = c_trias.size();
      c_operators.resize(nc);
      c_transfers.resize(nc);
      c_smoothers.resize(nc);
      if (comm && nc != mesh_index[0] + 1)
        throw std::runtime_error("sharded harness: the coarse level's mesh is not the end of the partition's mesh sequence");
      for (unsigned l = 0; l + 1 < nc; ++l)
        if (comm)
          c_dof_handlers.emplace_back(*partition, l, comm->rank(), degrees[0]); // (mesh index = level of the coarse hierarchy)
        else
          c_dof_handlers.emplace_back(c_trias[l], degrees[0]);
      for (unsigned l = 0; l + 1 < nc; ++l)
        if (comm)
          c_operators[l].reinit(ctx, c_dof_handlers[l], level_number_type, mesh_distributed(l) ? mesh_comm(l) : nullptr);
        else
          c_operators[l].reinit(ctx, c_dof_handlers[l], level_number_type);
      c_operators[nc - 1] = operators[0];
      for (unsigned l = 1; l < nc; ++l)
        c_transfers[l].reinit(c_operators[l], c_operators[l - 1]);
      for (unsigned l = 0; l + 1 < nc; ++l)
        c_smoothers[l].initialize(c_operators[l], sd);
      c_smoothers[nc - 1] = smoothers[0];
      coarse_mg           = std::make_unique<PreconditionMG>(ctx, c_operators, c_transfers, c_smoothers, "amg");
      std::cout << "note: CoarseGridSolverType '" << coarse << "' on the " << dof_handlers[0].n_dofs() << "-DoF coarse level: "
                << params.mg_data.coarse_solver.n_cycles << " V-cycle(s) of the geometric multigrid on that level" << std::endl;
    }
  PreconditionMG preconditioner(ctx, operators, transfers, smoothers, coarse, hp_local ? ls_mg.get() : coarse_mg.get(),
                                hp_local ? 1u : params.mg_data.coarse_solver.n_cycles, active_dof_handler.get());

  // fine (outer, double) operator, right-hand side (ref:multigrid_throughput.cc:2262-2324)
  Operator op;
  if (level_number_type == MGAMD_F64 && !local_smoothing)
    op = operators.back();
  else if (comm)
    op.reinit(ctx, fine_dof_handler, MGAMD_F64, distributed(n_levels - 1) ? level_comm(n_levels - 1) : nullptr);
  else
    op.reinit(ctx, fine_dof_handler, MGAMD_F64);
  // DoFHandler::n_dofs() of the GLOBAL problem
  const uint64_t n_dofs_global = (comm && distributed(n_levels - 1)) ? (uint64_t)std::llround(level_comm(n_levels - 1)->allreduce_sum(ctx, (double)op.n_owned())) :
                                                                       fine_dof_handler.n_dofs();
  Vector solution, rhs;
  op.initialize_dof_vector(solution);
  op.initialize_dof_vector(rhs);
  op.rhs(rhs, simulation_kind);

  table.add_value("dim", 3);
  table.add_value("n_cells", tria->n_global_active_cells());
  table.add_value("n_cells_hn", tria->n_cells_with_hanging_nodes());
  table.add_value("n_cells_n", tria->n_global_active_cells() - tria->n_cells_with_hanging_nodes());
  table.add_value("degree", params.fe_degree_fine);
  table.add_value("n_ref_global", params.n_ref_global);
  table.add_value("n_ref_local", params.n_ref_local);
  table.add_value("n_dofs", n_dofs_global);
  // (the reference: the ranks that own cells of the coarsest level, ref:multigrid_throughput.cc:1464-1488; here every rank holds it)
  table.add_value("sub_comm_size", comm ? comm->n_ranks() : 1);

  if (params.verbose)
    {
      ConvergenceTable t;
      for (unsigned l = 0; l < n_levels; ++l)
        {
          t.add_value("cells", triangulations[l]->n_global_active_cells());
          t.add_value("dofs", dof_handlers[l].n_dofs());
        }
      t.write_text(std::cout);
    }

  // ---- solve protocol (ref:multigrid_throughput.cc:1137-1268)
  ReductionControl solver_control(params.mg_data.cg_normal.maxiter, params.mg_data.cg_normal.abstol, params.mg_data.cg_normal.reltol);
  ctx.synchronize();
  solution = 0.0;
  SolverCG(solver_control).solve(op, solution, rhs, preconditioner); // warm-up

  // Stage times: the reference connects host timers to Multigrid's signals (ref:multigrid_throughput.cc:1150-1234).  Here the
  // stages are bracketed by HIP events on the stream of the UNCHANGED cycle (no host synchronisation, collapsed coarse levels
  // included) and read after each solve, so `time`, `throughput` and the stage columns describe the same run.
  const unsigned n_repetitions = params.mg_data.n_repetitions;
  std::vector<std::vector<std::vector<double>>> all_stage_times(n_repetitions); // [repetition][stage 0..8][level], seconds
  std::vector<double>                           times(n_repetitions);
  preconditioner.enable_stage_timing(true);
  for (unsigned counter = 0; counter < n_repetitions; ++counter)
    {
      ctx.synchronize(); // MPI_Barrier
      double time = 0.0;
      {
        solution = 0.0;
        ScopedTimer timer(time);
        SolverCG(solver_control).solve(op, solution, rhs, preconditioner);
      }
      times[counter] = time;
      preconditioner.read_stage_times(all_stage_times[counter]);
    }
  preconditioner.enable_stage_timing(false);
  const unsigned min_index = std::min_element(times.begin(), times.end()) - times.begin();
  const double   time      = times[min_index];
  const auto    &st        = all_stage_times[min_index];
  double         time_cg   = time;
  for (const auto &stage : st)
    for (const double v : stage)
      time_cg -= v;
  const unsigned its = std::max(1u, solver_control.last_step());
  table.add_value("n_levels", n_levels);
  table.add_value("n_iterations", solver_control.last_step());
  table.add_value("time", time);
  table.add_value("time_cg", time_cg / its);
  table.add_value("throughput", (double)n_dofs_global * solver_control.last_step() / time, true);
  static const char *stage_cols[7] = {"time_pre", "time_residuum", "time_res", "time_cs", "time_pro", "time_edge_pro", "time_post"};
  double             t_v           = 0.0;
  for (unsigned j = 0; j < 7; ++j)
    {
      double s = 0;
      for (unsigned l = 0; l < n_levels; ++l)
        s += st[j][l] / its;
      t_v += s;
      table.add_value(stage_cols[j], s, true);
    }
  double t_to_mg = 0, t_to_global = 0;
  for (unsigned l = 0; l < n_levels; ++l)
    {
      t_to_mg += st[7][l] / its;
      t_to_global += st[8][l] / its;
    }
  table.add_value("time_to_mg", t_to_mg, true);
  table.add_value("time_to_global", t_to_global, true);
  t_v += t_to_mg + t_to_global;
  if (params.verbose)
    {
      // partition statistics of the level meshes (ref:multigrid_throughput.cc:1657-1665, ref:include/mg_tools.h:267-512); the
      // p-levels of PMG/HPMG repeat the finest mesh: every distinct mesh once, coarse -> fine
      std::vector<std::shared_ptr<const Triangulation>> meshes;
      for (const auto &t : triangulations)
        if (meshes.empty() || meshes.back().get() != t.get())
          meshes.push_back(t);
      // local smoothing: the reference calls the single-triangulation overload (ref:multigrid_throughput.cc:1862-1872,
      // ref:include/mg_tools.h:39-61,85,195), which counts ALL cells of each refinement level of the one mesh: `triangulations`
      // holds exactly those level meshes here.  PMG (one mesh, several degrees): its coarsening sequence.
      if (!local_smoothing && meshes.size() == 1)
        meshes = create_geometric_coarsening_sequence(tria);
      if (hp_local)
        {
          meshes.clear();
          for (unsigned l = 0; l < tria->n_global_levels(); ++l)
            meshes.push_back(tria->level_mesh(l));
        }
      for (const auto &stat : print_multigrid_statistics(meshes, comm ? comm->n_ranks() : 1))
        table.add_value(stat.first, stat.second, true);
    }
  // this project's additions (after the reference's columns): DoF/s per V-cycle = n_dofs / (sum of the nine stage columns)
  //  and the coarse solver that actually ran
  table.add_value("dofs_per_s_per_vcycle", (double)n_dofs_global / t_v, true);
  table.add_value("coarse_solver", preconditioner.coarse_solver_used());

  if (params.verbose)
    {
      std::cout << "per-level stage times per CG iteration [s] (pre, residuum, res, cs, pro, edge_pro, post):" << std::endl;
      for (unsigned l = 0; l < n_levels; ++l)
        {
          std::cout << "  level " << l << ":";
          for (unsigned j = 0; j < 7; ++j)
            std::cout << " " << std::scientific << std::setprecision(3) << st[j][l] / its;
          std::cout << std::endl;
        }
    }
}

int
main(int argc, char **argv)
{
  try
    {
      if (argc == 1)
        {
          printf("ERROR: No .json parameter files has been provided!\n");
          return 1;
        }
      // One rank per GPU when launched with WORLD_SIZE / RANK / LOCAL_RANK in the environment (torchrun --no-python, mpirun -x,
      // a shell loop): the RCCL id travels through a file that rank 0 writes (MGAMD_RCCL_ID_FILE, default derived from
      // MASTER_PORT and the launcher's run id).  MGAMD_HARNESS_SHARDED=1 runs the sharded code path on one rank too.
      auto env_uint = [](const char *name, unsigned dflt) {
        const char *v = std::getenv(name);
        return v ? (unsigned)std::strtoul(v, nullptr, 10) : dflt;
      };
      const unsigned world = env_uint("WORLD_SIZE", 1), rank = env_uint("RANK", 0), local_rank = env_uint("LOCAL_RANK", rank);
      const bool     sharded = world > 1 || std::getenv("MGAMD_HARNESS_SHARDED") != nullptr;
      Context          ctx((int)(sharded ? local_rank : 0));
      ConvergenceTable table;
      Communicator     comm;
      std::string      id_file;
      if (sharded)
        {
          if (const char *f = std::getenv("MGAMD_RCCL_ID_FILE"))
            id_file = f;
          else
            {
              const char *port = std::getenv("MASTER_PORT"), *run = std::getenv("TORCHELASTIC_RUN_ID");
              id_file = std::string("/tmp/mgamd_rccl_id_") + (port ? port : "0") + "_" + (run ? run : "default");
            }
          if (rank == 0)
            std::remove(id_file.c_str()); // a stale id of an earlier job under the same name
          comm = Communicator::rccl(ctx, world, rank, Communicator::exchange_id_through_file(id_file, rank));
        }
      for (int i = 1; i < argc; i++)
        {
          if (rank == 0)
            std::cout << std::string(argv[i]) << std::endl;
          RunParameters params;
          params.parse(std::string(argv[i]));
          run(ctx, params, table, sharded ? &comm : nullptr);
          if (rank == 0)
            table.write_text(std::cout);
        }
      if (rank == 0)
        table.write_text(std::cout);
      if (sharded && rank == 0)
        std::remove(id_file.c_str());
    }
  catch (std::exception &exc)
    {
      std::cerr << std::endl
                << std::endl
                << "----------------------------------------------------" << std::endl;
      std::cerr << "Exception on processing: " << std::endl << exc.what() << std::endl << "Aborting!" << std::endl
                << "----------------------------------------------------" << std::endl;
      return 1;
    }
  catch (...)
    {
      std::cerr << std::endl
                << std::endl
                << "----------------------------------------------------" << std::endl;
      std::cerr << "Unknown exception!" << std::endl << "Aborting!" << std::endl
                << "----------------------------------------------------" << std::endl;
      return 1;
    }
  return 0;
}
