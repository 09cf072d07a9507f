/* mgamd.h -- C ABI of the MI355X-native matrix-free multigrid V-cycle.
 *
 * Drop-in boundary for the hot path of peterrum/dealii-multigrid's multigrid_throughput.cc.
 * The reference has no FFI: its path sits behind deal.II's duck-typed C++ template concepts
 * (SURVEY.md section 8b).  Every entry point below names the reference interface it replaces; the
 * header-only C++ layer `dealii_multigrid_amd/csrc/mgamd.hpp` re-exposes them as classes with
 * deal.II's method names (Operator::vmult, PreconditionChebyshev::vmult/step,
 * MGTwoLevelTransfer::prolongate_and_add, PreconditionMG::vmult, SolverCG::solve ...).
 *
 * Conventions: every function returns 0 on success and a non-zero status on failure, never
 * throws across the ABI; `mgamd_last_error()` returns the message of the calling thread's last
 * failure.  Every *_create has a *_destroy.  Vectors are device resident and owned by the library;
 * host transfer is explicit.  Calls on one context are ordered on that context's HIP stream and
 * are asynchronous unless they return data to the host.  One host thread per context.
 * Device functions fail with status MGAMD_ERR_NO_DEVICE when no gfx950 device is usable: there is
 * no CPU fallback in this library.
 */
#ifndef MGAMD_H
#define MGAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGAMD_OK 0
#define MGAMD_ERR 1
#define MGAMD_ERR_NO_DEVICE 2
#define MGAMD_ERR_INVALID 3

#define MGAMD_INVALID_DOF 0xFFFFFFFFu

typedef struct mgamd_ctx       mgamd_ctx;       /* device + stream (+ communicator)                    */
typedef struct mgamd_tria      mgamd_tria;      /* octree mesh  <-> parallel::distributed::Triangulation */
typedef struct mgamd_dofs      mgamd_dofs;      /* DoFHandler + AffineConstraints + MatrixFree tables    */
typedef struct mgamd_vec       mgamd_vec;       /* LinearAlgebra::distributed::Vector<Number>            */
typedef struct mgamd_level_op  mgamd_level_op;  /* Operator<3,1,Number>  (ref:include/operator.h:11)    */
typedef struct mgamd_cheb      mgamd_cheb;      /* PreconditionChebyshev<Operator,Vector,DiagonalMatrix> */
typedef struct mgamd_transfer2 mgamd_transfer2; /* MGTwoLevelTransfer<3,Vector>                          */
typedef struct mgamd_mg        mgamd_mg;        /* Multigrid + PreconditionMG + MGTransferGlobalCoarsening */
typedef struct mgamd_partition mgamd_partition; /* domain decomposition of the level hierarchy (one rank per GPU) */
typedef struct mgamd_comm      mgamd_comm;      /* communicator: RCCL over xGMI, or the in-process simulator      */
typedef struct mgamd_sim_group mgamd_sim_group; /* the ranks of one in-process simulation                         */

const char *mgamd_last_error(void);
const char *mgamd_version(void);

/* ---------------------------------------------------------------------------------------------
 * Host-side setup (no GPU needed)
 * ------------------------------------------------------------------------------------------- */

/* GridGenerator::{hyper_cube+refine_global, create_quadrant, create_quadrant_flexible,
 * create_annulus, create_circle}: ref:multigrid_throughput.cc:2048-2062, ref:include/grid_generator.h.
 * geometry in {"hypercube","quadrant","quadrant_flexible","annulus","circle"}. */
int mgamd_tria_create(const char *geometry, unsigned n_ref_global, unsigned n_ref_local, mgamd_tria **out);
/* A caller-built mesh: the active cells of a parallel::distributed::Triangulation<3> over ONE root cell [-1,1]^3 (what the
 * reference hands to the path: ref:multigrid_throughput.cc:2048-2062 builds it, :1540-1604 consumes it), given as octree leaves
 * (level, i, j, k), 0 <= i, j, k < 2^level, in any order.  The leaves must tile the cube exactly and be 2:1 balanced across
 * faces, edges and corners (p4est's full balance, deal.II's default); otherwise MGAMD_ERR_INVALID with the offending leaf in
 * mgamd_last_error().  Zero Dirichlet data on the whole boundary, as for the named geometries. */
int mgamd_tria_create_from_leaves(uint64_t n_leaves, const uint8_t *level, const uint32_t *i, const uint32_t *j, const uint32_t *k,
                                  mgamd_tria **out);
/* one level of MGTransferGlobalCoarseningTools::create_geometric_coarsening_sequence
 * (ref:multigrid_throughput.cc:2219-2224): coarsen every family once, re-balance. */
int mgamd_tria_coarsen(const mgamd_tria *fine, mgamd_tria **out);
int mgamd_tria_destroy(mgamd_tria *t);
/* n_global_active_cells(), n_global_levels(), cells with a coarser face/edge neighbour
 * (table columns n_cells / n_cells_hn: ref:multigrid_throughput.cc:2177-2190, 2329-2331). */
int mgamd_tria_info(const mgamd_tria *t, uint64_t *n_cells, uint32_t *n_levels, uint64_t *n_cells_hn);
/* leaves in Morton (p4est) order; arrays of n_cells entries, any may be NULL */
int mgamd_tria_get_cells(const mgamd_tria *t, uint8_t *level, uint32_t *i, uint32_t *j, uint32_t *k, uint16_t *mask);

typedef struct
{
  uint32_t degree;
  uint64_t n_cells;
  uint32_t n_dofs;      /* DoFHandler::n_dofs()                                              */
  uint32_t n_interior;  /* [0, n_interior): slot-interior DoFs, contiguous per slot          */
  uint32_t n_tail;      /* then the shared unconstrained DoFs                                */
  uint32_t n_dirichlet; /* then Dirichlet DoFs                                               */
  uint32_t n_hanging;   /* then hanging-node DoFs                                            */
  uint32_t n_groups;    /* slot groups (brick sizes)                                         */
  uint32_t group_B[8];  /* cells per direction of each group's bricks                        */
  uint64_t group_slots[8];
  /* sharded levels (mgamd_dofs_create_local): the tail is [owned | copies]; *_owned count each DoF once globally */
  uint32_t n_tail_owned, n_dirichlet_owned, n_hanging_owned;
  uint32_t n_peers;       /* ranks this rank exchanges partial sums with on this level               */
  uint32_t n_halo_send;   /* tail entries sent (= received) per exchange                              */
  /* local-smoothing levels (mgamd_dofs_create_level): refinement-edge DoFs, numbered between the tail and the Dirichlet
   * DoFs: [ I | T | E | D | H ] */
  uint32_t n_edge;
  /* sharded levels: the first group_halo_slots[g] slots of group g touch DoFs shared with other ranks (the halo exchange of
   * an operator application runs underneath the remaining slots) */
  uint64_t group_halo_slots[8];
} mgamd_dofs_info_t;

/* DoFHandler::distribute_dofs(FE_Q(degree)) + zero Dirichlet on boundary id 0 + hanging-node
 * constraints + MatrixFree::reinit tables (ref:multigrid_throughput.cc:1578-1595,
 * ref:include/operator.h:24-47).  max_brick = 0 selects the largest brick that fits LDS;
 * max_brick = 1 disables bricks (every cell a generic slot); max_brick = -1 ("auto", what the
 * harness and the hierarchies use) is 0 on large levels and 1 on latency-bound small ones. */
int mgamd_dofs_create(const mgamd_tria *t, int degree, int max_brick, mgamd_dofs **out);
int mgamd_dofs_destroy(mgamd_dofs *d);
int mgamd_dofs_info(const mgamd_dofs *d, mgamd_dofs_info_t *info);
/* slot group and slot index of every cell (0xFE: cell of another rank) */
int mgamd_dofs_get_cell_slots(const mgamd_dofs *d, uint8_t *group, uint32_t *slot);
/* geometric identity of each DoF: keys[5*i..] = {px,py,pz,dirmask,level} (tests / oracle matching) */
int mgamd_dofs_get_keys(const mgamd_dofs *d, int32_t *keys);
/* per cell (p+1)^3 gathered DoF indices, x fastest; hanging entities resolved to the parent's
 * DoFs; MGAMD_INVALID_DOF marks Dirichlet DoFs */
int mgamd_dofs_get_cell_dofs(const mgamd_dofs *d, uint32_t *out);
/* Operator::rhs for f == 1, g == 0 (ref:include/operator.h:362-447, ref:multigrid_throughput.cc:2286-2291) */
int mgamd_dofs_rhs_constant(const mgamd_dofs *d, double *out);
/* Operator::rhs for SimulationType `kind` (0 "Constant": f = 1, g = 0; 1 "Gaussian": source and boundary values of
 * ref:multigrid_throughput.cc:60-125,2294-2298): QGauss(p+1) load vector minus the Dirichlet lifting, constrained rows 0
 * (ref:include/operator.h:362-447); and AffineConstraints::distribute on a host vector of n_dofs values */
int mgamd_dofs_rhs(const mgamd_dofs *d, int kind, double *out);
int mgamd_dofs_distribute(const mgamd_dofs *d, int kind, double *x);

/* Operator::get_trilinos_system_matrix / get_petsc_system_matrix (ref:include/operator.h:244-358: MatrixFreeTools::compute_matrix of
 * the cell kernel with the constraints): the assembled level matrix C^T K C + identity on the constrained rows, CSR with sorted
 * columns, in this library's DoF numbering.  Call with NULL arrays for *nnz; row_ptr has n_dofs + 1 entries.  It is what the AMG
 * coarse solver is built on. */
int mgamd_dofs_matrix(const mgamd_dofs *d, uint64_t *nnz, uint32_t *row_ptr, uint32_t *col, double *val);
/* sizes of the smoothed-aggregation hierarchy the "amg" coarse solver builds on that matrix (mgamd_mg_create): rows and non-zeros
 * per AMG level, finest first (arrays of max_levels entries, may be NULL) */
int mgamd_dofs_amg_setup_info(const mgamd_dofs *d, uint32_t *n_levels, uint32_t *rows, uint64_t *nnz, uint32_t max_levels);

/* Local smoothing (`HMG-local`, ref:multigrid_throughput.cc:1670-1873): level `level` of the refinement hierarchy = ALL
 * cells of that refinement level, active or not (DoFHandler::distribute_mg_dofs); its DoFs + zero Dirichlet boundary +
 * refinement-edge set (MGConstrainedDoFs / MGTools::extract_inner_interface_dofs, ref:include/operator.h:49-70,539-556);
 * and the copy_to_mg / copy_from_mg index pairs of that level (MGLevelGlobalTransfer, interface DoFs skipped): call with
 * NULL arrays for the count. */
int mgamd_tria_level_mesh(const mgamd_tria *fine, unsigned level, mgamd_tria **out);
int mgamd_dofs_create_level(const mgamd_tria *level_mesh, int degree, int max_brick, mgamd_dofs **out);
int mgamd_ls_copy_indices(const mgamd_dofs *active_mesh_dofs, const mgamd_dofs *level_dofs, unsigned level, uint64_t *count,
                          uint32_t *global_idx, uint32_t *level_idx);

/* Domain decomposition (the reference partitions with p4est + RepartitioningPolicyTools,
 * ref:multigrid_throughput.cc:2066-2175).  trias: the level meshes, coarsest first.  A root level is chosen; its leaves are
 * cut in Morton order into n_ranks chunks of equal weight (hanging-node cells weigh `hanging_weight`, cf. CellWeightPolicy);
 * finer levels inherit the owner of their root ancestor, coarser levels are replicated on every rank. */
int mgamd_partition_create(const mgamd_tria *const *trias, unsigned n_levels, unsigned n_ranks, double hanging_weight,
                           mgamd_partition **out);
/* same, keeping every level with fewer than min_root_cells cells replicated (a distributed level pays one halo exchange
 * per operator application whatever its size) */
int mgamd_partition_create_ex(const mgamd_tria *const *trias, unsigned n_levels, unsigned n_ranks, double hanging_weight,
                              uint64_t min_root_cells, mgamd_partition **out);
/* Two tiers (csrc/partition.hpp): levels below the root level with at least min_sub_root_cells cells are cut into
 * n_ranks / group PARTS, each held (and worked on) by `group` consecutive ranks -- the counterpart of the reference's
 * agglomeration of coarse levels onto fewer processes (ref:multigrid_throughput.cc:379-418,1464-1501), without idle ranks.
 * group: a power of two that divides n_ranks; 1 = mgamd_partition_create_ex.  mgamd_partition_tiers reports the first such level
 * (== root level if there is none) and the group size; mgamd_dofs_create_local / mgamd_partition_get_owner accept those levels
 * (owners are part numbers there); their operators take mgamd_comm_subset(comm, group). */
int mgamd_partition_create_tiered(const mgamd_tria *const *trias, unsigned n_levels, unsigned n_ranks, double hanging_weight,
                                  uint64_t min_root_cells, unsigned group, uint64_t min_sub_root_cells, mgamd_partition **out);
int mgamd_partition_tiers(const mgamd_partition *p, unsigned *sub_root_level, unsigned *group);
int mgamd_partition_destroy(mgamd_partition *p);
/* MGTools::print_multigrid_statistics for this partition (ref:include/mg_tools.h:267-512; verbose-mode table columns
 * workload_eff, workload_path_max, vertical_eff, horizontal_eff, mem_total: ref:multigrid_throughput.cc:1657-1665), in that
 * order in stats[5]; definitions in csrc/partition.hpp */
int mgamd_partition_statistics(const mgamd_partition *p, double stats[5]);
int mgamd_partition_info(const mgamd_partition *p, unsigned *root_level, unsigned *n_ranks);
/* owner rank of every cell of a distributed level (level >= root_level) */
int mgamd_partition_get_owner(const mgamd_partition *p, unsigned level, uint16_t *owner);
/* level tables of one rank: distributed levels hold the rank's cells only (plus its halo plan), replicated levels are
 * identical to mgamd_dofs_create */
int mgamd_dofs_create_local(const mgamd_partition *p, unsigned level, unsigned rank, int degree, int max_brick, mgamd_dofs **out);

/* halo plan of a distributed level (tests / host-side emulation): sizes = {n_peers, n_send, n_shared, n_contrib};
 * peers[n_peers], peer_offset[n_peers+1], pack_idx[n_send] (tail index = global index - n_interior), and for every shared
 * tail DoF its tail index sh_tail[n_shared], CSR sh_ptr[n_shared+1] into sh_src[n_contrib] (-1 = own partial, else position
 * in the concatenated receive buffer, ascending rank order) and sh_owner_src[n_shared] (-1 if this rank owns the DoF) */
int mgamd_dofs_halo_sizes(const mgamd_dofs *d, uint32_t sizes[4]);
int mgamd_dofs_halo_get(const mgamd_dofs *d, int32_t *peers, uint32_t *peer_offset, uint32_t *pack_idx, uint32_t *sh_tail,
                        uint32_t *sh_ptr, int32_t *sh_src, int32_t *sh_owner_src);

/* raw two-level transfer tables (for the CPU oracle): kind 0 identity, 1 h-embedding, 2 p-embedding */
int mgamd_transfer_tables_info(const mgamd_dofs *fine, const mgamd_dofs *coarse, uint64_t n_patches[3], uint32_t nf[3]);
int mgamd_transfer_tables_get(const mgamd_dofs *fine, const mgamd_dofs *coarse, int kind, uint32_t *coarse_idx,
                              uint16_t *coarse_mask, uint32_t *fine_idx);

/* ---------------------------------------------------------------------------------------------
 * Device runtime
 * ------------------------------------------------------------------------------------------- */
int mgamd_ctx_create(int device, mgamd_ctx **out);
int mgamd_ctx_destroy(mgamd_ctx *ctx);
int mgamd_ctx_synchronize(mgamd_ctx *ctx);
/* the HIP stream all work of this context is submitted to (hipStream_t as void*) */
int mgamd_ctx_stream(mgamd_ctx *ctx, void **stream);

/* Communicators.  RCCL: every rank calls mgamd_comm_rccl_create with the same 128-byte id obtained once from
 * mgamd_comm_rccl_unique_id (distribute it with any host-side channel, e.g. torch.distributed / MPI).  Simulator: all
 * ranks live in one process as host threads sharing one GPU; used to test the sharded path on a single device. */
int mgamd_comm_rccl_unique_id(char id[128]);
int mgamd_comm_rccl_create(mgamd_ctx *ctx, unsigned n_ranks, unsigned rank, const char id[128], mgamd_comm **out);
int mgamd_sim_group_create(unsigned n_ranks, mgamd_sim_group **out);
int mgamd_sim_group_destroy(mgamd_sim_group *g);
int mgamd_comm_sim_create(mgamd_sim_group *g, unsigned rank, mgamd_comm **out);
/* the communicator of a level that is cut into n_ranks / group parts (mgamd_partition_create_tiered): rank = part, halo peers are
 * parts, sums over the parts; keeps `base` alive */
int mgamd_comm_subset(mgamd_comm *base, unsigned group, mgamd_comm **out);
int mgamd_comm_destroy(mgamd_comm *c);
/* sum of a host scalar over all ranks (blocking) */
int mgamd_comm_allreduce_sum(mgamd_comm *c, mgamd_ctx *ctx, double value, double *result);

/* number_type: 8 = double, 4 = float (MGNumberType, ref:multigrid_throughput.cc:2430-2433) */
#define MGAMD_F64 8
#define MGAMD_F32 4

/* Vectors (LinearAlgebra::distributed::Vector: ref:include/operator.h:17) */
int mgamd_vec_create(mgamd_ctx *ctx, uint64_t n, int number_type, mgamd_vec **out);
int mgamd_vec_destroy(mgamd_vec *v);
int mgamd_vec_size(const mgamd_vec *v, uint64_t *n);
int mgamd_vec_from_host(mgamd_vec *v, const double *src); /* casts when the vector is float */
int mgamd_vec_to_host(const mgamd_vec *v, double *dst);
int mgamd_vec_set(mgamd_vec *v, double value);                       /* v = value               */
int mgamd_vec_copy(mgamd_vec *dst, const mgamd_vec *src);            /* dst = src (casts)       */
int mgamd_vec_axpy(mgamd_vec *y, double a, const mgamd_vec *x);      /* y += a x                */
int mgamd_vec_sadd(mgamd_vec *y, double s, double a, const mgamd_vec *x); /* y = s y + a x      */
int mgamd_vec_dot(const mgamd_vec *x, const mgamd_vec *y, double *result);
int mgamd_vec_norm2(const mgamd_vec *x, double *result);

/* Operator::reinit (ref:include/operator.h:24-47) */
int mgamd_level_op_create(mgamd_ctx *ctx, const mgamd_dofs *dofs, int number_type, mgamd_level_op **out);
/* same, for a level built with mgamd_dofs_create_local: vmult / inverse diagonal / rhs / smoother / transfers complete
 * the sums of shared DoFs across ranks through `comm` (may be NULL for replicated levels) */
int mgamd_level_op_create_distributed(mgamd_ctx *ctx, const mgamd_dofs *dofs, int number_type, mgamd_comm *comm, mgamd_level_op **out);
int mgamd_level_op_destroy(mgamd_level_op *op);
/* inner product over the GLOBAL vector (every DoF counted once across ranks) */
int mgamd_level_op_dot(mgamd_level_op *op, const mgamd_vec *x, const mgamd_vec *y, double *result);
/* the export half of MatrixFree::cell_loop's ghost exchange on a distributed level (LinearAlgebra::distributed::Vector::
 * compress(VectorOperation::add), ref:include/operator.h:166-167): the entries of shared DoFs of v become the sum over the
 * sharing ranks, identical on all of them; stream-ordered; no-op on a level that is not distributed.  bench.py times it. */
int mgamd_level_op_exchange_add_tail(mgamd_level_op *op, mgamd_vec *v);
/* number of DoFs this rank owns (sums to DoFHandler::n_dofs() over the ranks) */
int mgamd_level_op_n_owned(const mgamd_level_op *op, uint64_t *n);
int mgamd_level_op_m(const mgamd_level_op *op, uint64_t *n); /* Operator::m (ref:include/operator.h:123) */
/* Operator::initialize_dof_vector (ref:include/operator.h:140) */
int mgamd_level_op_init_vector(const mgamd_level_op *op, mgamd_vec **out);
/* Operator::vmult: dst = A src, identity on constrained rows (ref:include/operator.h:152-183) */
int mgamd_level_op_vmult(mgamd_level_op *op, mgamd_vec *dst, const mgamd_vec *src);
/* Operator::compute_inverse_diagonal (ref:include/operator.h:228-242) */
int mgamd_level_op_inverse_diagonal(mgamd_level_op *op, mgamd_vec *diagonal);
/* Operator::rhs with f == 1, g == 0 (ref:include/operator.h:362-447) */
int mgamd_level_op_rhs(mgamd_level_op *op, mgamd_vec *rhs);
/* the same for SimulationType `kind` (see mgamd_dofs_rhs), and constraints.distribute(solution) after the solve */
int mgamd_level_op_rhs_kind(mgamd_level_op *op, int kind, mgamd_vec *rhs);
int mgamd_level_op_distribute(mgamd_level_op *op, int kind, mgamd_vec *x);

/* PreconditionChebyshev (ref:multigrid_throughput.cc:849-852, 867-883): the inverse diagonal is
 * computed internally (DiagonalMatrix preconditioner); eigenvalues are estimated at creation with
 * eig_cg_n_iterations CG steps, max *= 1.2, min = max / smoothing_range. */
int mgamd_cheb_create(mgamd_level_op *op, unsigned degree, double smoothing_range, unsigned eig_cg_n_iterations,
                      mgamd_cheb **out);
int mgamd_cheb_destroy(mgamd_cheb *c);
int mgamd_cheb_vmult(mgamd_cheb *c, mgamd_vec *dst, const mgamd_vec *src); /* zero start: MGSmoother::apply  */
int mgamd_cheb_step(mgamd_cheb *c, mgamd_vec *dst, const mgamd_vec *src);  /* general start: MGSmoother::smooth */
int mgamd_cheb_get_eigen_estimates(const mgamd_cheb *c, double *min_eigenvalue, double *max_eigenvalue);

/* MGTwoLevelTransfer::reinit(dof_fine, dof_coarse, constraint_fine, constraint_coarse)
 * (ref:multigrid_throughput.cc:1600-1604) */
int mgamd_transfer2_create(mgamd_level_op *fine, mgamd_level_op *coarse, mgamd_transfer2 **out);
int mgamd_transfer2_destroy(mgamd_transfer2 *t);
int mgamd_transfer2_prolongate_and_add(mgamd_transfer2 *t, mgamd_vec *dst_fine, const mgamd_vec *src_coarse);
int mgamd_transfer2_restrict_and_add(mgamd_transfer2 *t, mgamd_vec *dst_coarse, const mgamd_vec *src_fine);
/* Inside mgamd_mg_vcycle the part of a transfer that belongs to the fine level's 17-point lattice bricks runs INSIDE the level
 * operator's passes (restriction in the residual pass, prolongation in the first post-smoothing pass: the steps
 * Multigrid::level_v_step runs back to back, ref:multigrid_throughput.cc:1093-1099); *n = number of such bricks (0: none;
 * MGAMD_NO_FUSED_TRANSFER=1 in the environment disables it).  The two entry points above always do the whole transfer. */
int mgamd_transfer2_n_fused_bricks(const mgamd_transfer2 *t, uint64_t *n);

/* Multigrid + PreconditionMG over MGTransferGlobalCoarsening (ref:multigrid_throughput.cc:1093-1133,
 * 1618-1621).  levels[0] is the coarsest; transfers[l] connects levels l-1 and l (transfers[0] unused,
 * may be NULL); smoothers[0] is only used by coarse solver "cg_with_chebyshev".
 * coarse_solver (CoarseGridSolverType, ref:multigrid_throughput.cc:909-1077):
 *   "direct"             dense inverse (coarse levels up to 4096 DoFs)
 *   "cg", "cg_with_chebyshev"   SolverCG to reltol 1e-4, unpreconditioned / with the level-0 Chebyshev smoother (:911-944)
 *   "amg", "cg_with_amg", "amg_petsc"   the reference applies Trilinos ML / BoomerAMG to Operator::get_trilinos_system_matrix
 *                        (:945-1077).  Here: on a coarse level of <= 4096 DoFs (global coarsening ends on one cell) an exact solve
 *                        ("direct"); on a larger one (PMG: the p = 1 space on the finest mesh) the library's OWN
 *                        smoothed-aggregation AMG on the assembled level matrix (mgamd_dofs_matrix; Chebyshev(2) smoothing, dense
 *                        coarsest solve), applied CoarseSolverNCycles times as the coarse solver ("amg") or as the preconditioner
 *                        of the coarse CG ("cg_with_amg").  Same role and inputs as ML, not the same aggregates: its iteration
 *                        counts cannot be parity-checked against ML.
 * mgamd_mg_create_nested additionally takes n_cycles (CoarseSolverNCycles) and, optionally, `coarse_mg`: a geometric multigrid
 * whose finest level IS levels[0]; if given, it takes the AMG's place ("gmg_vcycle": x = V(b), x += V(b - A x) ...; the only choice
 * on a sharded coarse level).  mgamd_mg_coarse_solver_used returns what runs: "direct" | "cg" | "cg_with_chebyshev" | "amg" |
 * "cg_with_amg" | "gmg_vcycle". */
int mgamd_mg_create(mgamd_ctx *ctx, unsigned n_levels, mgamd_level_op *const *levels, mgamd_transfer2 *const *transfers,
                    mgamd_cheb *const *smoothers, const char *coarse_solver, mgamd_mg **out);
int mgamd_mg_create_nested(mgamd_ctx *ctx, unsigned n_levels, mgamd_level_op *const *levels, mgamd_transfer2 *const *transfers,
                           mgamd_cheb *const *smoothers, const char *coarse_solver, mgamd_mg *coarse_mg, unsigned n_cycles,
                           mgamd_mg **out);
int mgamd_mg_coarse_solver_used(const mgamd_mg *mg, char name[32]);
/* Local smoothing (`HMG-local`: solve_with_local_smoothing, ref:multigrid_throughput.cc:1670-1873).  levels[l] = Operator on
 * refinement level l (mgamd_dofs_create_level: edge-constrained level operator, ref:include/operator.h:49-120,152-183),
 * transfers = MGTransferMatrixFree between the refinement levels, Multigrid with the interface (edge) matrices
 * (ref:multigrid_throughput.cc:1081-1130), PreconditionMG::vmult on vectors of the ACTIVE mesh `active_mesh_dofs`
 * (copy_to_mg / copy_from_mg).  Stage 5 of the stage callback / timing is the edge prolongation. */
int mgamd_mg_create_local_smoothing(mgamd_ctx *ctx, unsigned n_levels, mgamd_level_op *const *levels, mgamd_transfer2 *const *transfers,
                                    mgamd_cheb *const *smoothers, const mgamd_dofs *active_mesh_dofs, const char *coarse_solver,
                                    mgamd_mg **out);
/* Operator::vmult_interface_up (ref:include/operator.h:203-226): dst = A with the refinement-edge DoFs unconstrained, applied
 * to src restricted to those DoFs (MatrixFreeOperators::MGInterfaceOperator::Tvmult) */
int mgamd_level_op_vmult_interface_up(mgamd_level_op *op, mgamd_vec *dst, const mgamd_vec *src);
/* Operator::vmult_interface_down (ref:include/operator.h:191-201): the plain cell loop, identity on the constrained (Dirichlet,
 * hanging) rows only; on a local-smoothing level the refinement-edge DoFs take part as ordinary DoFs.  This is what
 * MatrixFreeOperators::MGInterfaceOperator::vmult forwards to, i.e. the matrix of Multigrid's residual step in the reference
 * (ref:multigrid_throughput.cc:857-862).  On a level without refinement edges it equals mgamd_level_op_vmult. */
int mgamd_level_op_vmult_interface_down(mgamd_level_op *op, mgamd_vec *dst, const mgamd_vec *src);
/* Levels up to 2048 DoFs (MGAMD_COLLAPSE_MAX_DOFS) with a direct coarse solver are applied as ONE tabulated dense matrix
 * (the zero-start V-cycle below a level is a linear map of its defect; result-equivalent).  This switches the
 * tabulated path off/on at run time (bench.py reports the cycle time both ways); *collapse_level = the level it replaces
 * (0: none). */
int mgamd_mg_set_collapse(mgamd_mg *mg, int enable, unsigned *collapse_level);
int mgamd_mg_destroy(mgamd_mg *mg);
/* PreconditionMG::vmult: z = V-cycle(r)  (ref:multigrid_throughput.cc:1132-1133) -- the metric's unit of work.
 * z and r are vectors of the finest level's outer number type (double). */
int mgamd_mg_vcycle(mgamd_mg *mg, mgamd_vec *z, const mgamd_vec *r);
/* Multigrid::connect_{pre_smoother_step,residual_step,restriction,coarse_solve,prolongation,
 * edge_prolongation,post_smoother_step} + PreconditionMG::connect_transfer_to_{mg,global}
 * (ref:multigrid_throughput.cc:1183-1192, 1233-1234).  stage 0..6 as in the reference's timer index,
 * 7 = transfer_to_mg, 8 = transfer_to_global.  With a callback installed the V-cycle runs eagerly
 * and synchronises the stream around every stage so host clocks are meaningful. */
typedef void (*mgamd_stage_callback)(int stage, int start, unsigned level, void *user);
int mgamd_mg_set_stage_callback(mgamd_mg *mg, mgamd_stage_callback cb, void *user);
/* Stage times WITHOUT host synchronisation: while enabled, a HIP event pair is recorded on the stream around every stage
 * of the unchanged cycle (collapsed coarse levels included: they appear as the coarse solve, stage 3, of the collapse
 * level).  mgamd_mg_stage_times synchronises once, ADDS the elapsed milliseconds of all stages recorded since the last
 * read to ms[stage * n_levels + level] (9 x n_levels entries) and returns the number of stage records consumed.  This is
 * how the harness fills the reference's time_pre ... time_to_global columns (ref:multigrid_throughput.cc:1381-1401). */
int mgamd_mg_stage_timing(mgamd_mg *mg, int enable);
int mgamd_mg_stage_times(mgamd_mg *mg, double *ms, unsigned n_levels, uint64_t *n_records);
/* run `n` V-cycles back to back and return the average time per cycle in milliseconds, measured
 * with HIP events on the context's stream (bench.py / harness). use_graph != 0 replays a captured
 * hipGraph of one cycle. */
int mgamd_mg_time_vcycles(mgamd_mg *mg, mgamd_vec *z, const mgamd_vec *r, unsigned n, int use_graph, double *ms_per_cycle);

/* SolverCG + ReductionControl with PreconditionMG (ref:multigrid_throughput.cc:1140-1147, 1238-1254,
 * 1625-1635): solves A x = b from x = 0; returns last_step() and the final residual norm. */
int mgamd_solve_cg(mgamd_level_op *A, mgamd_mg *preconditioner, mgamd_vec *x, const mgamd_vec *b, double reltol, double abstol,
                   unsigned maxiter, unsigned *n_iterations, double *residual_norm);

#ifdef __cplusplus
}
#endif
#endif /* MGAMD_H */
