// C ABI, device entry points (include/mgamd.h, section "Device runtime").
#include "runtime.hpp"
#include <cstdio>

using namespace mgamd;

#define REQUIRE(cond)                                    \
  if (!(cond))                                           \
  throw std::invalid_argument("null or invalid argument: " #cond)

extern "C" {

int
mgamd_ctx_create(int device, mgamd_ctx **out)
{
  MGAMD_TRY
  REQUIRE(out);
  auto *c = new mgamd_ctx;
  try
    {
      c->ctx = std::make_unique<Ctx>(device);
    }
  catch (...)
    {
      delete c;
      throw;
    }
  *out = c;
  MGAMD_CATCH
}

int
mgamd_ctx_destroy(mgamd_ctx *ctx)
{
  delete ctx;
  return MGAMD_OK;
}

int
mgamd_ctx_synchronize(mgamd_ctx *ctx)
{
  MGAMD_TRY
  REQUIRE(ctx);
  ctx->ctx->sync();
  MGAMD_CATCH
}

int
mgamd_ctx_stream(mgamd_ctx *ctx, void **stream)
{
  MGAMD_TRY
  REQUIRE(ctx && stream);
  *stream = (void *)ctx->ctx->stream;
  MGAMD_CATCH
}

int
mgamd_vec_create(mgamd_ctx *ctx, uint64_t n, int number_type, mgamd_vec **out)
{
  MGAMD_TRY
  REQUIRE(ctx && out);
  *out = vec_create(ctx->ctx.get(), n, number_type);
  MGAMD_CATCH
}

int
mgamd_vec_destroy(mgamd_vec *v)
{
  delete v;
  return MGAMD_OK;
}

int
mgamd_vec_size(const mgamd_vec *v, uint64_t *n)
{
  MGAMD_TRY
  REQUIRE(v && n);
  *n = v->n;
  MGAMD_CATCH
}

int
mgamd_vec_from_host(mgamd_vec *v, const double *src)
{
  MGAMD_TRY
  REQUIRE(v && src);
  if (v->type == MGAMD_F64)
    HIP_CHECK(hipMemcpyAsync(v->data, src, v->n * 8, hipMemcpyHostToDevice, v->ctx->stream));
  else
    {
      std::vector<float> f(src, src + v->n);
      HIP_CHECK(hipMemcpyAsync(v->data, f.data(), v->n * 4, hipMemcpyHostToDevice, v->ctx->stream));
      v->ctx->sync();
    }
  v->ctx->sync();
  MGAMD_CATCH
}

int
mgamd_vec_to_host(const mgamd_vec *v, double *dst)
{
  MGAMD_TRY
  REQUIRE(v && dst);
  if (v->type == MGAMD_F64)
    {
      HIP_CHECK(hipMemcpyAsync(dst, v->data, v->n * 8, hipMemcpyDeviceToHost, v->ctx->stream));
      v->ctx->sync();
    }
  else
    {
      std::vector<float> f(v->n);
      HIP_CHECK(hipMemcpyAsync(f.data(), v->data, v->n * 4, hipMemcpyDeviceToHost, v->ctx->stream));
      v->ctx->sync();
      for (size_t i = 0; i < v->n; ++i)
        dst[i] = f[i];
    }
  MGAMD_CATCH
}

int
mgamd_vec_set(mgamd_vec *v, double value)
{
  MGAMD_TRY
  REQUIRE(v);
  vec_set(*v, value);
  MGAMD_CATCH
}

int
mgamd_vec_copy(mgamd_vec *dst, const mgamd_vec *src)
{
  MGAMD_TRY
  REQUIRE(dst && src);
  vec_copy(*dst, *src);
  MGAMD_CATCH
}

int
mgamd_vec_axpy(mgamd_vec *y, double a, const mgamd_vec *x)
{
  MGAMD_TRY
  REQUIRE(y && x);
  vec_sadd(*y, 1.0, a, *x);
  MGAMD_CATCH
}

int
mgamd_vec_sadd(mgamd_vec *y, double s, double a, const mgamd_vec *x)
{
  MGAMD_TRY
  REQUIRE(y && x);
  vec_sadd(*y, s, a, *x);
  MGAMD_CATCH
}

int
mgamd_vec_dot(const mgamd_vec *x, const mgamd_vec *y, double *result)
{
  MGAMD_TRY
  REQUIRE(x && y && result);
  *result = vec_dot(*x, *y);
  MGAMD_CATCH
}

int
mgamd_vec_norm2(const mgamd_vec *x, double *result)
{
  MGAMD_TRY
  REQUIRE(x && result);
  *result = std::sqrt(vec_dot(*x, *x));
  MGAMD_CATCH
}

int
mgamd_level_op_create(mgamd_ctx *ctx, const mgamd_dofs *dofs, int number_type, mgamd_level_op **out)
{
  MGAMD_TRY
  REQUIRE(ctx && dofs && out);
  auto *h = new mgamd_level_op;
  try
    {
      h->op.reset(make_level_operator(ctx->ctx.get(), dofs, number_type));
    }
  catch (...)
    {
      delete h;
      throw;
    }
  *out = h;
  MGAMD_CATCH
}

int
mgamd_level_op_create_distributed(mgamd_ctx *ctx, const mgamd_dofs *dofs, int number_type, mgamd_comm *comm, mgamd_level_op **out)
{
  MGAMD_TRY
  REQUIRE(ctx && dofs && out);
  if (dofs->halo && !comm)
    throw std::invalid_argument("a distributed level needs a communicator");
  auto *h = new mgamd_level_op;
  try
    {
      h->op.reset(make_level_operator(ctx->ctx.get(), dofs, number_type, comm ? comm->comm : nullptr));
    }
  catch (...)
    {
      delete h;
      throw;
    }
  *out = h;
  MGAMD_CATCH
}

int
mgamd_level_op_dot(mgamd_level_op *op, const mgamd_vec *x, const mgamd_vec *y, double *result)
{
  MGAMD_TRY
  REQUIRE(op && x && y && result);
  *result = op->op->dot(*x, *y);
  MGAMD_CATCH
}

int
mgamd_level_op_n_owned(const mgamd_level_op *op, uint64_t *n)
{
  MGAMD_TRY
  REQUIRE(op && n);
  *n = op->op->comm ? op->op->n_dofs_owned() : op->op->n_dofs();
  MGAMD_CATCH
}

int
mgamd_comm_rccl_unique_id(char id[128])
{
  MGAMD_TRY
  REQUIRE(id);
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId u;
  RcclComm::check(ncclGetUniqueId(&u), "ncclGetUniqueId");
  std::memcpy(id, &u, 128);
  MGAMD_CATCH
}

int
mgamd_comm_rccl_create(mgamd_ctx *ctx, unsigned n_ranks, unsigned rank, const char id[128], mgamd_comm **out)
{
  MGAMD_TRY
  REQUIRE(ctx && id && out && rank < n_ranks);
  HIP_CHECK(hipSetDevice(ctx->ctx->device));
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  auto *c = new mgamd_comm;
  try
    {
      c->comm = std::make_shared<RcclComm>((int)n_ranks, (int)rank, u);
    }
  catch (...)
    {
      delete c;
      throw;
    }
  *out = c;
  MGAMD_CATCH
}

int
mgamd_sim_group_create(unsigned n_ranks, mgamd_sim_group **out)
{
  MGAMD_TRY
  REQUIRE(out && n_ranks > 0);
  auto *g  = new mgamd_sim_group;
  g->group = std::make_shared<SimGroup>((int)n_ranks);
  *out     = g;
  MGAMD_CATCH
}

int
mgamd_sim_group_destroy(mgamd_sim_group *g)
{
  delete g;
  return MGAMD_OK;
}

int
mgamd_comm_sim_create(mgamd_sim_group *g, unsigned rank, mgamd_comm **out)
{
  MGAMD_TRY
  REQUIRE(g && out && (int)rank < g->group->n);
  auto *c = new mgamd_comm;
  c->comm = std::make_shared<SimComm>(g->group, (int)rank);
  *out    = c;
  MGAMD_CATCH
}

int
mgamd_comm_subset(mgamd_comm *base, unsigned group, mgamd_comm **out)
{
  MGAMD_TRY
  if (!base || !base->comm || !out || group == 0)
    throw std::invalid_argument("bad argument");
  auto *c = new mgamd_comm;
  try
    {
      c->comm = std::make_shared<SubsetComm>(base->comm, (int)group);
    }
  catch (...)
    {
      delete c;
      throw;
    }
  *out = c;
  MGAMD_CATCH
}

int
mgamd_comm_destroy(mgamd_comm *c)
{
  delete c;
  return MGAMD_OK;
}

int
mgamd_comm_allreduce_sum(mgamd_comm *c, mgamd_ctx *ctx, double value, double *result)
{
  MGAMD_TRY
  REQUIRE(c && ctx && result);
  *result = c->comm->allreduce_sum_host(value, ctx->ctx->stream);
  MGAMD_CATCH
}

int
mgamd_level_op_destroy(mgamd_level_op *op)
{
  delete op;
  return MGAMD_OK;
}

int
mgamd_level_op_m(const mgamd_level_op *op, uint64_t *n)
{
  MGAMD_TRY
  REQUIRE(op && n);
  *n = op->op->n_dofs();
  MGAMD_CATCH
}

int
mgamd_level_op_init_vector(const mgamd_level_op *op, mgamd_vec **out)
{
  MGAMD_TRY
  REQUIRE(op && out);
  *out = vec_create(op->op->ctx, op->op->n_dofs(), op->op->type);
  MGAMD_CATCH
}

int
mgamd_level_op_vmult(mgamd_level_op *op, mgamd_vec *dst, const mgamd_vec *src)
{
  MGAMD_TRY
  REQUIRE(op && dst && src);
  op->op->vmult(*dst, *src);
  MGAMD_CATCH
}

int
mgamd_level_op_inverse_diagonal(mgamd_level_op *op, mgamd_vec *diagonal)
{
  MGAMD_TRY
  REQUIRE(op && diagonal);
  if (diagonal->type != op->op->type)
    throw std::invalid_argument("vector number type does not match the operator's");
  op->op->compute_inverse_diagonal(*diagonal);
  MGAMD_CATCH
}

int
mgamd_level_op_rhs(mgamd_level_op *op, mgamd_vec *rhs)
{
  MGAMD_TRY
  REQUIRE(op && rhs);
  op->op->rhs(*rhs);
  MGAMD_CATCH
}

int
mgamd_level_op_rhs_kind(mgamd_level_op *op, int kind, mgamd_vec *rhs)
{
  MGAMD_TRY
  REQUIRE(op && rhs);
  if (kind != 0 && kind != 1)
    throw std::invalid_argument("SimulationType kind must be 0 (Constant) or 1 (Gaussian)");
  op->op->rhs(*rhs, kind);
  MGAMD_CATCH
}

int
mgamd_level_op_distribute(mgamd_level_op *op, int kind, mgamd_vec *x)
{
  MGAMD_TRY
  REQUIRE(op && x);
  if (kind != 0 && kind != 1)
    throw std::invalid_argument("SimulationType kind must be 0 (Constant) or 1 (Gaussian)");
  op->op->distribute(*x, kind);
  MGAMD_CATCH
}

int
mgamd_level_op_debug_stamps(mgamd_level_op *op, unsigned long long *out, uint64_t max_count, uint64_t *count)
{
  MGAMD_TRY
  REQUIRE(op && out && count);
  *count = op->op->read_debug_stamps(out, max_count);
  MGAMD_CATCH
}

int
mgamd_cheb_create(mgamd_level_op *op, unsigned degree, double smoothing_range, unsigned eig_cg_n_iterations, mgamd_cheb **out)
{
  MGAMD_TRY
  REQUIRE(op && out);
  auto *h = new mgamd_cheb;
  try
    {
      h->c.reset(make_chebyshev(op->op.get(), degree, smoothing_range, eig_cg_n_iterations));
    }
  catch (...)
    {
      delete h;
      throw;
    }
  *out = h;
  MGAMD_CATCH
}

int
mgamd_cheb_destroy(mgamd_cheb *c)
{
  delete c;
  return MGAMD_OK;
}

int
mgamd_cheb_vmult(mgamd_cheb *c, mgamd_vec *dst, const mgamd_vec *src)
{
  MGAMD_TRY
  REQUIRE(c && dst && src);
  c->c->vmult(*dst, *src);
  MGAMD_CATCH
}

int
mgamd_cheb_step(mgamd_cheb *c, mgamd_vec *dst, const mgamd_vec *src)
{
  MGAMD_TRY
  REQUIRE(c && dst && src);
  c->c->step(*dst, *src);
  MGAMD_CATCH
}

int
mgamd_cheb_get_eigen_estimates(const mgamd_cheb *c, double *min_eigenvalue, double *max_eigenvalue)
{
  MGAMD_TRY
  REQUIRE(c);
  if (min_eigenvalue)
    *min_eigenvalue = c->c->min_eig;
  if (max_eigenvalue)
    *max_eigenvalue = c->c->max_eig;
  MGAMD_CATCH
}

int
mgamd_transfer2_create(mgamd_level_op *fine, mgamd_level_op *coarse, mgamd_transfer2 **out)
{
  MGAMD_TRY
  REQUIRE(fine && coarse && out);
  auto *h = new mgamd_transfer2;
  try
    {
      h->t.reset(make_transfer2(fine->op.get(), coarse->op.get()));
    }
  catch (...)
    {
      delete h;
      throw;
    }
  *out = h;
  MGAMD_CATCH
}

int
mgamd_transfer2_destroy(mgamd_transfer2 *t)
{
  delete t;
  return MGAMD_OK;
}

int
mgamd_transfer2_prolongate_and_add(mgamd_transfer2 *t, mgamd_vec *dst_fine, const mgamd_vec *src_coarse)
{
  MGAMD_TRY
  REQUIRE(t && dst_fine && src_coarse);
  t->t->prolongate_and_add(*dst_fine, *src_coarse);
  MGAMD_CATCH
}

int
mgamd_transfer2_restrict_and_add(mgamd_transfer2 *t, mgamd_vec *dst_coarse, const mgamd_vec *src_fine)
{
  MGAMD_TRY
  REQUIRE(t && dst_coarse && src_fine);
  t->t->restrict_and_add(*dst_coarse, *src_fine);
  MGAMD_CATCH
}

int
mgamd_transfer2_n_fused_bricks(const mgamd_transfer2 *t, uint64_t *n)
{
  MGAMD_TRY
  REQUIRE(t && n);
  *n = t->t->n_fused_bricks_total();
  MGAMD_CATCH
}

int
mgamd_mg_create(mgamd_ctx *ctx, unsigned n_levels, mgamd_level_op *const *levels, mgamd_transfer2 *const *transfers,
                mgamd_cheb *const *smoothers, const char *coarse_solver, mgamd_mg **out)
{
  MGAMD_TRY
  REQUIRE(ctx && levels && out && n_levels > 0 && (n_levels == 1 || (transfers && smoothers)));
  std::vector<LevelOperatorBase *> L(n_levels, nullptr);
  std::vector<Transfer2Base *>     Tr(n_levels, nullptr);
  std::vector<ChebyshevBase *>     Sm(n_levels, nullptr);
  for (unsigned l = 0; l < n_levels; ++l)
    {
      REQUIRE(levels[l]);
      L[l] = levels[l]->op.get();
      if (transfers && transfers[l])
        Tr[l] = transfers[l]->t.get();
      if (smoothers && smoothers[l])
        Sm[l] = smoothers[l]->c.get();
    }
  auto *h = new mgamd_mg;
  try
    {
      h->mg.reset(make_multigrid(ctx->ctx.get(), n_levels, L.data(), Tr.data(), Sm.data(), coarse_solver ? coarse_solver : "direct"));
    }
  catch (...)
    {
      delete h;
      throw;
    }
  *out = h;
  MGAMD_CATCH
}

int
mgamd_mg_create_nested(mgamd_ctx *ctx, unsigned n_levels, mgamd_level_op *const *levels, mgamd_transfer2 *const *transfers,
                       mgamd_cheb *const *smoothers, const char *coarse_solver, mgamd_mg *coarse_mg, unsigned n_cycles, mgamd_mg **out)
{
  MGAMD_TRY
  REQUIRE(ctx && levels && out && n_levels > 0 && (n_levels == 1 || (transfers && smoothers)));
  std::vector<LevelOperatorBase *> L(n_levels, nullptr);
  std::vector<Transfer2Base *>     Tr(n_levels, nullptr);
  std::vector<ChebyshevBase *>     Sm(n_levels, nullptr);
  for (unsigned l = 0; l < n_levels; ++l)
    {
      REQUIRE(levels[l]);
      L[l] = levels[l]->op.get();
      if (transfers && transfers[l])
        Tr[l] = transfers[l]->t.get();
      if (smoothers && smoothers[l])
        Sm[l] = smoothers[l]->c.get();
    }
  auto *h = new mgamd_mg;
  try
    {
      h->mg.reset(make_multigrid(ctx->ctx.get(), n_levels, L.data(), Tr.data(), Sm.data(), coarse_solver ? coarse_solver : "amg",
                                 coarse_mg ? coarse_mg->mg.get() : nullptr, n_cycles));
    }
  catch (...)
    {
      delete h;
      throw;
    }
  *out = h;
  MGAMD_CATCH
}

int
mgamd_mg_create_local_smoothing(mgamd_ctx *ctx, unsigned n_levels, mgamd_level_op *const *levels, mgamd_transfer2 *const *transfers,
                                mgamd_cheb *const *smoothers, const mgamd_dofs *active_mesh_dofs, const char *coarse_solver, mgamd_mg **out)
{
  MGAMD_TRY
  REQUIRE(ctx && levels && out && active_mesh_dofs && n_levels > 0 && (n_levels == 1 || (transfers && smoothers)));
  std::vector<LevelOperatorBase *> L(n_levels, nullptr);
  std::vector<Transfer2Base *>     Tr(n_levels, nullptr);
  std::vector<ChebyshevBase *>     Sm(n_levels, nullptr);
  for (unsigned l = 0; l < n_levels; ++l)
    {
      REQUIRE(levels[l]);
      L[l] = levels[l]->op.get();
      if (transfers && transfers[l])
        Tr[l] = transfers[l]->t.get();
      if (smoothers && smoothers[l])
        Sm[l] = smoothers[l]->c.get();
    }
  auto *h = new mgamd_mg;
  try
    {
      h->mg.reset(make_multigrid(ctx->ctx.get(), n_levels, L.data(), Tr.data(), Sm.data(), coarse_solver ? coarse_solver : "amg"));
      h->mg->setup_local_smoothing(*active_mesh_dofs->tables);
    }
  catch (...)
    {
      delete h;
      throw;
    }
  *out = h;
  MGAMD_CATCH
}

int
mgamd_level_op_exchange_add_tail(mgamd_level_op *op, mgamd_vec *v)
{
  MGAMD_TRY
  REQUIRE(op && v);
  op->op->exchange_add_tail(*v);
  MGAMD_CATCH
}

int
mgamd_level_op_vmult_interface_up(mgamd_level_op *op, mgamd_vec *dst, const mgamd_vec *src)
{
  MGAMD_TRY
  REQUIRE(op && dst && src);
  op->op->vmult_interface_up(*dst, *src);
  MGAMD_CATCH
}

int
mgamd_level_op_vmult_interface_down(mgamd_level_op *op, mgamd_vec *dst, const mgamd_vec *src)
{
  MGAMD_TRY
  REQUIRE(op && dst && src);
  op->op->vmult_interface_down(*dst, *src);
  MGAMD_CATCH
}

int
mgamd_mg_set_collapse(mgamd_mg *mg, int enable, unsigned *collapse_level)
{
  MGAMD_TRY
  REQUIRE(mg);
  const unsigned l = mg->mg->set_collapse(enable != 0);
  if (collapse_level)
    *collapse_level = l;
  MGAMD_CATCH
}

int
mgamd_mg_coarse_solver_used(const mgamd_mg *mg, char name[32])
{
  MGAMD_TRY
  REQUIRE(mg && name);
  std::snprintf(name, 32, "%s", mg->mg->coarse_used.c_str());
  MGAMD_CATCH
}

int
mgamd_mg_destroy(mgamd_mg *mg)
{
  delete mg;
  return MGAMD_OK;
}

int
mgamd_mg_vcycle(mgamd_mg *mg, mgamd_vec *z, const mgamd_vec *r)
{
  MGAMD_TRY
  REQUIRE(mg && z && r);
  mg->mg->vcycle(*z, *r);
  MGAMD_CATCH
}

int
mgamd_mg_set_stage_callback(mgamd_mg *mg, mgamd_stage_callback cb, void *user)
{
  MGAMD_TRY
  REQUIRE(mg);
  mg->mg->cb      = cb;
  mg->mg->cb_user = user;
  MGAMD_CATCH
}

int
mgamd_mg_stage_timing(mgamd_mg *mg, int enable)
{
  MGAMD_TRY
  if (!mg)
    throw std::invalid_argument("null argument");
  mg->mg->set_stage_timing(enable != 0);
  MGAMD_CATCH
}

int
mgamd_mg_stage_times(mgamd_mg *mg, double *ms, unsigned n_levels, uint64_t *n_records)
{
  MGAMD_TRY
  if (!mg || !ms)
    throw std::invalid_argument("null argument");
  if (n_levels != mg->mg->n_levels())
    throw std::invalid_argument("stage_times: ms must hold 9 x n_levels entries");
  const size_t n = mg->mg->read_stage_times(ms);
  if (n_records)
    *n_records = n;
  MGAMD_CATCH
}

int
mgamd_mg_time_vcycles(mgamd_mg *mg, mgamd_vec *z, const mgamd_vec *r, unsigned n, int use_graph, double *ms_per_cycle)
{
  MGAMD_TRY
  REQUIRE(mg && z && r && ms_per_cycle);
  *ms_per_cycle = mg->mg->time_vcycles(*z, *r, n, use_graph != 0);
  MGAMD_CATCH
}

int
mgamd_solve_cg(mgamd_level_op *A, mgamd_mg *preconditioner, mgamd_vec *x, const mgamd_vec *b, double reltol, double abstol,
               unsigned maxiter, unsigned *n_iterations, double *residual_norm)
{
  MGAMD_TRY
  REQUIRE(A && x && b);
  unsigned it  = 0;
  double   res = 0;
  solve_cg(*A->op, preconditioner ? preconditioner->mg.get() : nullptr, *x, *b, reltol, abstol, maxiter, it, res);
  if (n_iterations)
    *n_iterations = it;
  if (residual_norm)
    *residual_norm = res;
  MGAMD_CATCH
}

int
mgamd_ctx_kernel_profile(mgamd_ctx *ctx, int enable)
{
  MGAMD_TRY
  REQUIRE(ctx);
  Ctx &c = *ctx->ctx;
  c.harvest_profile();
  c.profile       = enable != 0;
  c.prof_ms_accum = 0;
  c.prof_n_accum  = 0;
  c.prof_bytes    = 0;
  c.prof_bytes_moved = 0;
  MGAMD_CATCH
}

int
mgamd_ctx_kernel_profile_brick(mgamd_ctx *ctx, int brick_size)
{
  MGAMD_TRY
  REQUIRE(ctx);
  ctx->ctx->prof_brick = brick_size;
  MGAMD_CATCH
}

int
mgamd_ctx_kernel_profile_read(mgamd_ctx *ctx, double *total_ms, uint64_t *n_launches, double *algorithmic_bytes)
{
  MGAMD_TRY
  REQUIRE(ctx);
  Ctx &c = *ctx->ctx;
  c.harvest_profile();
  if (total_ms)
    *total_ms = c.prof_ms_accum;
  if (n_launches)
    *n_launches = c.prof_n_accum;
  if (algorithmic_bytes)
    *algorithmic_bytes = c.prof_bytes;
  MGAMD_CATCH
}

int
mgamd_ctx_kernel_profile_bytes_moved(mgamd_ctx *ctx, double *bytes_moved)
{
  MGAMD_TRY
  REQUIRE(ctx);
  REQUIRE(bytes_moved);
  *bytes_moved = ctx->ctx->prof_bytes_moved;
  MGAMD_CATCH
}

} // extern "C"
