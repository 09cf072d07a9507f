// Device runtime implementation (see runtime.hpp).  Compiled with hipcc --offload-arch=gfx950.
#include "runtime.hpp"
#include "level_operator.hpp"
#include "amg.hpp"

#include <chrono>
#include <cmath>
#include <mutex>
#include <set>
#include <unordered_map>
#include <cstring>
#include <type_traits>

namespace mgamd
{
  // ------------------------------------------------------------------------------------------
  // Context
  // ------------------------------------------------------------------------------------------
  Ctx::Ctx(int dev)
    : device(dev)
  {
    int        count = 0;
    hipError_t e     = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
      throw NoDeviceError("no HIP device available: this library has no CPU fallback");
    if (dev < 0 || dev >= count)
      throw std::invalid_argument("device index out of range");
    HIP_CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
      throw NoDeviceError(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    sync_events.resize(64);
    for (hipEvent_t &e : sync_events)
      HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_CHECK(hipMalloc((void **)&d_partial, 1024 * sizeof(double)));
    HIP_CHECK(hipMalloc((void **)&d_result, 8 * sizeof(double)));
    HIP_CHECK(hipMalloc((void **)&d_cg, 8 * sizeof(double)));
    HIP_CHECK(hipHostMalloc((void **)&h_result, 8 * sizeof(double)));
  }

  Ctx::~Ctx()
  {
    (void)hipStreamSynchronize(stream);
    (void)hipStreamSynchronize(side);
    for (hipEvent_t e : sync_events)
      (void)hipEventDestroy(e);
    (void)hipStreamDestroy(side);
    for (auto &p : prof_events)
      {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
      }
    (void)hipFree(d_partial);
    (void)hipFree(d_result);
    (void)hipFree(d_cg);
    (void)hipHostFree(h_result);
    (void)hipStreamDestroy(stream);
  }

  void
  Ctx::harvest_profile()
  {
    sync();
    for (size_t i = 0; i < prof_used; ++i)
      {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, prof_events[i].first, prof_events[i].second));
        prof_ms_accum += ms;
      }
    prof_n_accum += prof_used;
    prof_used = 0;
  }

} // namespace mgamd

mgamd_vec::~mgamd_vec()
{
  if (data)
    (void)hipFree(data);
}

namespace mgamd
{
  mgamd_vec *
  vec_create(Ctx *ctx, size_t n, int type)
  {
    if (type != MGAMD_F64 && type != MGAMD_F32)
      throw std::invalid_argument("number_type must be MGAMD_F64 or MGAMD_F32");
    auto v  = std::make_unique<mgamd_vec>();
    v->ctx  = ctx;
    v->n    = n;
    v->type = type;
    if (n)
      {
        HIP_CHECK(hipMalloc(&v->data, n * (size_t)type));
        HIP_CHECK(hipMemsetAsync(v->data, 0, n * (size_t)type, ctx->stream));
      }
    return v.release();
  }

  void
  vec_set(mgamd_vec &v, double value)
  {
    if (!v.n)
      return;
    if (v.type == MGAMD_F64)
      hipLaunchKernelGGL(vec_set_kernel<double>, grid_for(v.n), 256, 0, v.ctx->stream, v.as<double>(), value, v.n);
    else
      hipLaunchKernelGGL(vec_set_kernel<float>, grid_for(v.n), 256, 0, v.ctx->stream, v.as<float>(), (float)value, v.n);
  }

  void
  vec_copy(mgamd_vec &d, const mgamd_vec &s)
  {
    if (d.n != s.n)
      throw std::invalid_argument("vector size mismatch");
    if (!d.n)
      return;
    const int    g  = grid_for(d.n);
    hipStream_t  st = d.ctx->stream;
    if (d.type == MGAMD_F64 && s.type == MGAMD_F64)
      HIP_CHECK(hipMemcpyAsync(d.data, s.data, d.n * 8, hipMemcpyDeviceToDevice, st));
    else if (d.type == MGAMD_F32 && s.type == MGAMD_F32)
      HIP_CHECK(hipMemcpyAsync(d.data, s.data, d.n * 4, hipMemcpyDeviceToDevice, st));
    else if (d.type == MGAMD_F32)
      hipLaunchKernelGGL((vec_copy_kernel<float, double>), g, 256, 0, st, d.as<float>(), s.as<double>(), d.n);
    else
      hipLaunchKernelGGL((vec_copy_kernel<double, float>), g, 256, 0, st, d.as<double>(), s.as<float>(), d.n);
  }

  void
  vec_sadd(mgamd_vec &y, double s, double a, const mgamd_vec &x)
  {
    if (y.n != x.n || y.type != x.type)
      throw std::invalid_argument("vector mismatch");
    if (!y.n)
      return;
    if (y.type == MGAMD_F64)
      hipLaunchKernelGGL(vec_sadd_kernel<double>, grid_for(y.n), 256, 0, y.ctx->stream, y.as<double>(), s, a, x.as<double>(), y.n);
    else
      hipLaunchKernelGGL(vec_sadd_kernel<float>, grid_for(y.n), 256, 0, y.ctx->stream, y.as<float>(), (float)s, (float)a, x.as<float>(),
                         y.n);
  }


  // S[slot] = x . y over the first n entries, summed over the ranks of `comm` (stream-ordered, no host synchronisation)
  template <typename T>
  static void
  dot_to_device(Ctx *ctx, Comm *comm, double *S, const T *x, const T *y, size_t n, int slot)
  {
    int g = std::min(grid_for(std::max<size_t>(n, 1)), 1024);
    hipLaunchKernelGGL(cg_dot_kernel<T>, g, 256, 0, ctx->stream, x, y, n, ctx->d_partial);
    hipLaunchKernelGGL(vec_dot_final_kernel<>, 1, 256, 0, ctx->stream, ctx->d_partial, g, S + slot);
    if (comm)
      comm->allreduce_sum(S + slot, 1, MGAMD_F64, ctx->stream);
  }
  static double
  read_scalar(Ctx *ctx, const double *S, int slot)
  {
    HIP_CHECK(hipMemcpyAsync(ctx->h_result, S + slot, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    return ctx->h_result[0];
  }

  // Device-resident preconditioned CG from x = 0 (SolverCG + ReductionControl: ref:multigrid_throughput.cc:1143-1144,
  // 888-895).  vmult(Ap, p) and precond(z, r) enqueue work on the stream; r holds b on entry.  n_dot: entries counted by
  // the inner products (the owned prefix on a sharded level, all otherwise).  One host read-back per iteration (the
  // residual norm); alpha, beta and the three inner products never leave the device: they live in S (8 doubles, one array per
  // solver instance -- the coarse CG of a V-cycle runs INSIDE the outer CG's preconditioner).
  template <typename T, typename VMULT, typename PRECOND>
  static void
  device_pcg(Ctx *ctx, Comm *comm, double *S, size_t n, size_t n_dot, T *x, T *r, T *z, T *p, T *Ap, VMULT vmult, PRECOND precond, double reltol,
             double abstol, unsigned maxiter, unsigned &n_iterations, double &residual)
  {
    const int g  = grid_for(n);
    const int gd = std::min(g, 1024);
    HIP_CHECK(hipMemsetAsync(x, 0, n * sizeof(T), ctx->stream));
    dot_to_device(ctx, comm, S, r, r, n_dot, 3);
    double       res  = std::sqrt(read_scalar(ctx, S, 3));
    const double res0 = res;
    n_iterations      = 0;
    residual          = res;
    if (res <= abstol)
      return;
    precond(z, r);
    HIP_CHECK(hipMemcpyAsync(p, z, n * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    int cur = 0; // S[cur] = r.z
    dot_to_device(ctx, comm, S, r, z, n_dot, cur);
    for (unsigned it = 1; it <= maxiter; ++it)
      {
        vmult(Ap, p);
        dot_to_device(ctx, comm, S, p, Ap, n_dot, 2);
        hipLaunchKernelGGL(cg_update_xr_kernel<T>, gd, 256, 0, ctx->stream, x, r, p, Ap, n, n_dot, S, cur, ctx->d_partial);
        hipLaunchKernelGGL(vec_dot_final_kernel<>, 1, 256, 0, ctx->stream, ctx->d_partial, gd, S + 3);
        if (comm)
          comm->allreduce_sum(S + 3, 1, MGAMD_F64, ctx->stream);
        res          = std::sqrt(read_scalar(ctx, S, 3));
        n_iterations = it;
        residual     = res;
        if (res < reltol * res0 || res <= abstol)
          break;
        precond(z, r);
        dot_to_device(ctx, comm, S, r, z, n_dot, 1 - cur);
        hipLaunchKernelGGL(cg_update_p_kernel<T>, g, 256, 0, ctx->stream, p, z, n, S, 1 - cur, cur);
        cur = 1 - cur;
      }
    HIP_CHECK(hipGetLastError());
  }

  double
  vec_dot(const mgamd_vec &x, const mgamd_vec &y)
  {
    if (x.n != y.n || x.type != y.type)
      throw std::invalid_argument("vector mismatch");
    return x.type == MGAMD_F64 ? dot_raw(x.ctx, x.as<double>(), y.as<double>(), x.n) : dot_raw(x.ctx, x.as<float>(), y.as<float>(), x.n);
  }

  void
  LevelOperatorBase::distribute(mgamd_vec &x, int kind)
  {
    if (x.n != n_dofs())
      throw std::invalid_argument("distribute: vector size mismatch");
    std::vector<double> h(x.n);
    ctx->sync();
    if (x.type == MGAMD_F64)
      HIP_CHECK(hipMemcpy(h.data(), x.data, h.size() * 8, hipMemcpyDeviceToHost));
    else
      {
        std::vector<float> f(x.n);
        HIP_CHECK(hipMemcpy(f.data(), x.data, f.size() * 4, hipMemcpyDeviceToHost));
        std::copy(f.begin(), f.end(), h.begin());
      }
    tables->distribute(kind, h);
    if (x.type == MGAMD_F64)
      HIP_CHECK(hipMemcpy(x.data, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    else
      {
        std::vector<float> f(h.begin(), h.end());
        HIP_CHECK(hipMemcpy(x.data, f.data(), f.size() * 4, hipMemcpyHostToDevice));
      }
  }

  void
  LevelOperatorBase::rhs(mgamd_vec &b, int kind)
  {
    if (b.n != n_dofs())
      throw std::invalid_argument("rhs: vector size mismatch");
    std::vector<double> h;
    tables->compute_rhs_function(kind, h);
    if (b.type == MGAMD_F64)
      HIP_CHECK(hipMemcpyAsync(b.data, h.data(), h.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    else
      {
        std::vector<float> f(h.begin(), h.end());
        HIP_CHECK(hipMemcpyAsync(b.data, f.data(), f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ctx->sync();
      }
    ctx->sync();
    exchange_add_tail(b); // contributions of the other ranks' cells to shared DoFs
    ctx->sync();
  }

  LevelOperatorBase *
  make_level_operator(Ctx *ctx, const mgamd_dofs *dofs, int type, std::shared_ptr<Comm> comm)
  {
    if (type == MGAMD_F64)
      return new LevelOperator<double>(ctx, dofs, comm);
    if (type == MGAMD_F32)
      return new LevelOperator<float>(ctx, dofs, comm);
    throw std::invalid_argument("number_type must be MGAMD_F64 or MGAMD_F32");
  }

  // ------------------------------------------------------------------------------------------
  // Chebyshev smoother (deal.II PreconditionChebyshev; parameters ref:multigrid_throughput.cc:312-315,867-883)
  // ------------------------------------------------------------------------------------------
  static double
  lanczos_max_eigenvalue(const std::vector<double> &alpha, const std::vector<double> &beta, double *min_ev)
  {
    // symmetric tridiagonal T from the CG coefficients; eigenvalues by implicit QL (tqli, no vectors)
    const int           n = (int)alpha.size();
    std::vector<double> d(n), e(n, 0.0);
    for (int j = 0; j < n; ++j)
      {
        d[j] = 1.0 / alpha[j] + (j > 0 ? beta[j - 1] / alpha[j - 1] : 0.0);
        if (j + 1 < n)
          e[j] = std::sqrt(beta[j]) / alpha[j];
      }
    for (int l = 0; l < n; ++l)
      {
        int iter = 0, m;
        do
          {
            for (m = l; m < n - 1; ++m)
              {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 1e-300 + 2.3e-16 * dd)
                  break;
              }
            if (m != l)
              {
                if (iter++ == 200)
                  throw std::runtime_error("tridiagonal eigenvalue iteration did not converge");
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g        = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int    i;
                for (i = m - 1; i >= l; --i)
                  {
                    double f = s * e[i], b = c * e[i];
                    e[i + 1] = (r = std::hypot(f, g));
                    if (r == 0.0)
                      {
                        d[i + 1] -= p;
                        e[m] = 0.0;
                        break;
                      }
                    s        = f / r;
                    c        = g / r;
                    g        = d[i + 1] - p;
                    r        = (d[i] - g) * s + 2.0 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g        = c * r - b;
                  }
                if (r == 0.0 && i >= l)
                  continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0;
              }
          }
        while (m != l);
      }
    double mx = d[0], mn = d[0];
    for (double v : d)
      {
        mx = std::max(mx, v);
        mn = std::min(mn, v);
      }
    if (min_ev)
      *min_ev = mn;
    return mx;
  }

  template <typename T>
  struct Chebyshev : ChebyshevBase
  {
    LevelOperator<T> *lop;
    DBuf<T>           dinv, tmp;
    bool              fuse_first = true; // MGAMD_NO_FUSED_START=1: materialise x_1 (development A/B)

    Chebyshev(LevelOperator<T> *o, unsigned deg, double smoothing_range, unsigned eig_cg_n_iterations)
      : lop(o)
    {
      fuse_first = getenv("MGAMD_NO_FUSED_START") == nullptr;
      op     = o;
      degree = deg;
      Ctx         *ctx = o->ctx;
      const size_t n   = o->n_dofs();
      dinv.alloc(n);
      tmp.alloc(n);
      {
        // DiagonalMatrix preconditioner (ref:multigrid_throughput.cc:872-875)
        mgamd_vec dv;
        dv.ctx  = ctx;
        dv.n    = n;
        dv.type = (int)sizeof(T);
        dv.data = dinv.p;
        try
          {
            o->compute_inverse_diagonal(dv);
          }
        catch (...)
          {
            dv.data = nullptr;
            throw;
          }
        dv.data = nullptr;
      }
      o->build_dinv_codes(dinv.p);
      estimate_eigenvalues(smoothing_range, eig_cg_n_iterations);
    }

    void
    estimate_eigenvalues(double smoothing_range, unsigned n_it)
    {
      Ctx         *ctx = lop->ctx;
      const size_t n   = lop->n_dofs();
      // initial guess (deal.II set_initial_guess): v_i = (i mod 11) - mean
      std::vector<T> v(n);
      const char    *key_init = getenv("MGAMD_CHEB_KEY_INIT"); // tests: numbering-independent start vector on one rank too
      if (!lop->comm && !(key_init && atoi(key_init)))
        {
          double sum = 0;
          for (size_t i = 0; i < n; ++i)
            sum += (double)(i % 11);
          const T mean = (T)(sum / (double)n);
          for (size_t i = 0; i < n; ++i)
            v[i] = (T)(i % 11) - mean;
        }
      else
        {
          // sharded: there is no global index; use a numbering-independent value per DoF (a hash of its geometric
          // key, identical on every sharing rank), zero on constrained DoFs so that every entry is counted once
          std::vector<DofKey> keys;
          lop->tables->export_dof_keys(keys);
          const size_t nf = (size_t)lop->tables->n_interior + lop->tables->n_tail, nown = (size_t)lop->tables->n_interior + lop->tables->n_tail_owned;
          double       sum = 0;
          for (size_t i = 0; i < n; ++i)
            {
              const uint64_t hk = FlatMap::mix(pack_key((uint32_t)keys[i].px, (uint32_t)keys[i].py, (uint32_t)keys[i].pz, keys[i].dirmask, keys[i].level));
              v[i]              = i < nf ? (T)(hk % 11) : T(0);
              if (i < nown)
                sum += (double)v[i];
            }
          const double gsum = lop->comm ? lop->comm->allreduce_sum_host(sum, ctx->stream) : sum;
          const double gcnt = lop->comm ? lop->comm->allreduce_sum_host((double)nown, ctx->stream) : (double)nown;
          const T      mean = (T)(gsum / gcnt);
          for (size_t i = 0; i < nf; ++i)
            v[i] -= mean;
        }
      DBuf<T> r, z, d, Ad;
      r.upload(v);
      z.alloc(n);
      d.alloc(n);
      Ad.alloc(n);
      std::vector<double> alphas, betas;
      const int           g    = grid_for(n);
      auto                dot_raw = [&](Ctx *, const T *a, const T *b, size_t) { return lop->dot_raw_global(a, b); };
      double              res0 = std::sqrt(dot_raw(ctx, r.p, r.p, n));
      if (res0 > 0 && n_it > 0)
        {
          // z = D^-1 r ; d = z
          hipLaunchKernelGGL(vec_scaled_product_kernel<T>, g, 256, 0, ctx->stream, z.p, T(1), dinv.p, r.p, n);
          HIP_CHECK(hipMemcpyAsync(d.p, z.p, n * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
          double rz = dot_raw(ctx, r.p, z.p, n);
          for (unsigned it = 0; it < n_it; ++it)
            {
              lop->vmult_raw(Ad.p, d.p);
              const double dAd = dot_raw(ctx, d.p, Ad.p, n);
              if (!(dAd > 0))
                break;
              const double alpha = rz / dAd;
              hipLaunchKernelGGL(vec_sadd_kernel<T>, g, 256, 0, ctx->stream, r.p, T(1), T(-alpha), Ad.p, n);
              alphas.push_back(alpha);
              const double res = std::sqrt(dot_raw(ctx, r.p, r.p, n));
              if (res <= 1e-10)
                break;
              hipLaunchKernelGGL(vec_scaled_product_kernel<T>, g, 256, 0, ctx->stream, z.p, T(1), dinv.p, r.p, n);
              const double rz_new = dot_raw(ctx, r.p, z.p, n);
              const double beta   = rz_new / rz;
              betas.push_back(beta);
              rz = rz_new;
              hipLaunchKernelGGL(vec_sadd_kernel<T>, g, 256, 0, ctx->stream, d.p, T(beta), T(1), z.p, n);
            }
        }
      if (!alphas.empty())
        {
          betas.resize(alphas.size(), 0.0);
          max_eig = lanczos_max_eigenvalue(alphas, betas, &min_eig);
        }
      else
        min_eig = max_eig = 1.0;
      max_eig *= 1.2; // safety factor, the CG is not converged
      const double a = smoothing_range > 1.0 ? max_eig / smoothing_range : std::min(0.9 * max_eig, min_eig);
      delta          = 0.5 * (max_eig - a);
      theta          = 0.5 * (max_eig + a);
      ctx->sync();
    }

    // zero initial guess; result guaranteed in S; Tb is scratch of the same size
    void
    vmult_raw(T *S, T *Tb, const T *b)
    {
      Ctx         *ctx = lop->ctx;
      const size_t n   = lop->n_dofs();
      T           *cur = (degree % 2 == 1) ? S : Tb;
      T           *oth = (cur == S) ? Tb : S;
      // x_1 = D^-1 b / theta is only materialised when it is the result; the first operator pass computes it on the
      // fly from b and D^-1 (which it reads anyway) and the second one uses it as x_old the same way
      const bool passes = degree >= 2 && std::fabs(delta) >= 1e-40;
      if (!passes || !fuse_first)
        hipLaunchKernelGGL(vec_scaled_product_kernel<T>, grid_for(n), 256, 0, ctx->stream, cur, T(1.0 / theta), dinv.p, b, n);
      if (passes)
        {
          double rhok = delta / theta, sigma = theta / delta;
          for (unsigned j = 0; j + 1 < degree; ++j)
            {
              const double rhokp = 1.0 / (2.0 * sigma - rhok);
              const double f1 = rhokp * rhok, f2 = 2.0 * rhokp / delta;
              rhok = rhokp;
              if (fuse_first && j < 2)
                lop->cheb_raw(oth, j == 0 ? nullptr : cur, nullptr, b, dinv.p, f1, f2, (int)j + 1, 1.0 / theta);
              else
                lop->cheb_raw(oth, cur, j == 0 ? nullptr : oth, b, dinv.p, f1, f2);
              std::swap(cur, oth);
            }
        }
      if (cur != S)
        hipLaunchKernelGGL((vec_copy_kernel<T, T>), grid_for(n), 256, 0, ctx->stream, S, cur, n);
    }

    // general initial guess x0 in X; O is scratch; returns the buffer holding the result.  prolongate != nullptr: the initial
    // guess is X + P x_c with the prolongation fused into the first operator pass (X becomes X + P x_c on the way)
    T *
    step_raw(T *X, T *O, const T *b, const FusedTransferHost<T> *prolongate = nullptr, double *out_wide = nullptr)
    {
      // out_wide (float levels): the result is wanted there as doubles; nullptr is returned.  With a plain Chebyshev pass at the end
      // (degree >= 2) that pass stores the doubles itself, otherwise a cast follows
      const bool wide_in_pass = out_wide && degree >= 2 && std::fabs(delta) >= 1e-40 && sizeof(T) == 4;
      T *cur = X, *oth = O;
      if (prolongate)
        lop->cheb_prolongate_raw(oth, cur, b, dinv.p, 1.0 / theta, *prolongate);
      else
        lop->cheb_raw(oth, cur, nullptr, b, dinv.p, 0.0, 1.0 / theta);
      std::swap(cur, oth);
      if (degree >= 2 && std::fabs(delta) >= 1e-40)
        {
          double rhok = delta / theta, sigma = theta / delta;
          for (unsigned j = 0; j + 1 < degree; ++j)
            {
              const double rhokp = 1.0 / (2.0 * sigma - rhok);
              const double f1 = rhokp * rhok, f2 = 2.0 * rhokp / delta;
              rhok = rhokp;
              const bool last = j + 2 == degree;
              lop->cheb_raw(oth, cur, oth, b, dinv.p, f1, f2, 0, 0.0, (last && wide_in_pass) ? out_wide : nullptr);
              std::swap(cur, oth);
            }
        }
      if (out_wide && !wide_in_pass)
        hipLaunchKernelGGL((vec_copy_kernel<double, T>), grid_for(lop->n_dofs()), 256, 0, lop->ctx->stream, out_wide, cur, lop->n_dofs());
      return out_wide ? nullptr : cur;
    }

    void
    vmult(mgamd_vec &dst, const mgamd_vec &src) override
    {
      if (dst.n != lop->n_dofs() || src.n != lop->n_dofs() || dst.data == src.data)
        throw std::invalid_argument("PreconditionChebyshev::vmult: bad vectors");
      vmult_raw(dst.as<T>(), tmp.p, src.as<T>());
    }
    void
    step(mgamd_vec &dst, const mgamd_vec &src) override
    {
      if (dst.n != lop->n_dofs() || src.n != lop->n_dofs() || dst.data == src.data)
        throw std::invalid_argument("PreconditionChebyshev::step: bad vectors");
      T *res = step_raw(dst.as<T>(), tmp.p, src.as<T>());
      if (res != dst.as<T>())
        hipLaunchKernelGGL((vec_copy_kernel<T, T>), grid_for(dst.n), 256, 0, lop->ctx->stream, dst.as<T>(), res, dst.n);
    }
  };

  ChebyshevBase *
  make_chebyshev(LevelOperatorBase *op, unsigned degree, double smoothing_range, unsigned eig_cg_n_iterations)
  {
    if (degree < 1)
      throw std::invalid_argument("SmootherDegree must be >= 1");
    if (op->type == MGAMD_F64)
      return new Chebyshev<double>(static_cast<LevelOperator<double> *>(op), degree, smoothing_range, eig_cg_n_iterations);
    return new Chebyshev<float>(static_cast<LevelOperator<float> *>(op), degree, smoothing_range, eig_cg_n_iterations);
  }

  // ------------------------------------------------------------------------------------------
  // Two-level transfer
  // ------------------------------------------------------------------------------------------
  template <typename T>
  struct Transfer2 : Transfer2Base
  {
    struct GroupD
    {
      int                 kind = 0, nf = 2;
      size_t              n_patches = 0;
      DBuf<uint32_t>      coarse_idx, fine_idx, fine_idx_restrict; // restrict list: copies of other ranks' DoFs removed
      DBuf<uint16_t>      coarse_mask;
      std::vector<double> E;
      // p = 1 h-patches: tables of the register kernels (patch_p1_*_kernel)
      DBuf<uint32_t> p1_uniq_ptr, p1_uniq_idx, p1_fine_t, p1_fine_t_restrict;
      DBuf<uint16_t> p1_loc;
      uint32_t       p1_max_uniq = 0;
      void
      build_p1(const std::vector<uint32_t> &cidx, const std::vector<uint32_t> &fidx, const std::vector<uint32_t> *fidx_restrict)
      {
        const size_t          np = n_patches, nwg = (np + PATCH_P1_BLOCK - 1) / PATCH_P1_BLOCK;
        std::vector<uint32_t> ptr(nwg + 1, 0), idx, tmp;
        std::vector<uint16_t> l(np * 8, 0xFFFFu);
        for (size_t w = 0; w < nwg; ++w)
          {
            const size_t a = w * PATCH_P1_BLOCK, b = std::min(np, a + PATCH_P1_BLOCK);
            tmp.clear();
            for (size_t i = a * 8; i < b * 8; ++i)
              if (cidx[i] != INVALID_DOF)
                tmp.push_back(cidx[i]);
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            for (size_t i = a * 8; i < b * 8; ++i)
              if (cidx[i] != INVALID_DOF)
                l[i] = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), cidx[i]) - tmp.begin());
            idx.insert(idx.end(), tmp.begin(), tmp.end());
            ptr[w + 1]  = (uint32_t)idx.size();
            p1_max_uniq = std::max<uint32_t>(p1_max_uniq, (uint32_t)tmp.size());
          }
        if (idx.empty())
          idx.push_back(0);
        auto transpose = [&](const std::vector<uint32_t> &f) {
          std::vector<uint32_t> t(f.size());
          for (size_t q = 0; q < np; ++q)
            for (int k = 0; k < 27; ++k)
              t[(size_t)k * np + q] = f[q * 27 + k];
          return t;
        };
        p1_uniq_ptr.upload(ptr);
        p1_uniq_idx.upload(idx);
        p1_loc.upload(l);
        p1_fine_t.upload(transpose(fidx));
        if (fidx_restrict)
          p1_fine_t_restrict.upload(transpose(*fidx_restrict));
      }
    };
    struct BrickD
    {
      int            B = 2, fine_group = 0;
      size_t         n_bricks = 0;
      size_t         n_unfused = 0; // fused group: the bricks [n_unfused, n_bricks) are transferred inside the operator passes
      bool           fused = false;
      DBuf<uint32_t> slot, coarse_idx, own_shell, own_shell_restrict;
    };
    // Transfers fused into the fine level's operator passes (kernels.hpp MODE_RESIDUAL_RESTRICT / MODE_CHEB_PROLONGATE): tables by
    // SLOT of the fused group.  MGAMD_NO_FUSED_TRANSFER=1 disables them (development A/B; same results up to rounding).
    FusedTransferHost<T> fused;
    DBuf<uint32_t>       fused_flags;
    DBuf<uint32_t>       fused_coarse_idx;
    DBuf<uint8_t>        fused_tail_flags;
    size_t               n_fused_bricks = 0;
    bool
    fused_ready() const
    {
      return fused.group >= 0 && n_fused_bricks > 0;
    }
    uint64_t
    n_fused_bricks_total() const override
    {
      return fused_ready() ? n_fused_bricks : 0;
    }
    GroupD                               grp[3];
    std::vector<std::unique_ptr<BrickD>> bricks;
    LevelOperator<T>                    *fop = nullptr, *cop = nullptr;
    int                                  pc = 1, pf = 1;
    Ctx                                 *ctx = nullptr;
    FE1D                                 fec;

    Transfer2(LevelOperator<T> *f, LevelOperator<T> *c)
      : fec(c->tables->p)
    {
      fine   = f;
      coarse = c;
      fop    = f;
      cop    = c;
      ctx    = f->ctx;
      const char *nb         = getenv("MGAMD_NO_BRICK_TRANSFER");
      const bool  use_bricks = !(nb && atoi(nb));
      // the fine group whose bricks carry the fused transfers: the plain 17-point lattices of an h-transfer, if the level's
      // operator runs them with persistent workgroups; not on local-smoothing levels (edge rows, partial coverage)
      int fuse_group = -1;
      if (use_bricks && !getenv("MGAMD_NO_FUSED_TRANSFER") && f->tables->tria != c->tables->tria && f->tables->p == c->tables->p &&
          !f->tables->ls_level && !c->tables->ls_level && fused_transfer_supported<T>(f->tables->p))
        for (size_t gi = 0; gi < f->tables->groups.size(); ++gi)
          {
            const SlotGroup &g = f->tables->groups[gi];
            if (g.N == 17 && !g.constrained_group && g.n_slots() > 0)
              fuse_group = (int)gi;
          }
      TransferTables tt(*f->tables, *c->tables, use_bricks, fuse_group);
      // sharded fine level: a residual entry that is a copy of another rank's DoF is restricted by its owner only
      const uint32_t copy_lo = f->tables->n_interior + f->tables->n_tail_owned, copy_hi = f->tables->n_interior + f->tables->n_tail;
      auto           owned_only = [&](std::vector<uint32_t> v) {
        for (uint32_t &i : v)
          if (i != INVALID_DOF && i >= copy_lo && i < copy_hi)
            i = INVALID_DOF;
        return v;
      };
      for (const BrickTransferGroup &bg : tt.bricks)
        {
          auto d        = std::make_unique<BrickD>();
          d->B          = bg.B;
          d->fine_group = bg.fine_group;
          d->n_bricks   = bg.n_bricks();
          d->n_unfused  = bg.fused ? bg.n_unfused : bg.n_bricks();
          d->fused      = bg.fused;
          if (bg.fused && bg.n_bricks() > bg.n_unfused)
            {
              const SlotGroup &fg  = f->tables->groups[bg.fine_group];
              const size_t     ns  = fg.n_slots(), nsh = (size_t)fg.n_shell, nc3 = (size_t)bg.Nc * bg.Nc * bg.Nc;
              if (2 * ((nsh + 255) / 256) > 15)
                throw std::runtime_error("fused transfer: shell too large for the flag word");
              std::vector<uint32_t> fl(ns * 256, 0);
              std::vector<uint32_t> ci(ns * nc3, INVALID_DOF);
              for (size_t q = bg.n_unfused; q < bg.n_bricks(); ++q)
                {
                  const size_t sl = bg.slot[q];
                  for (size_t t = 0; t < 256; ++t)
                    fl[sl * 256 + t] = 0x8000u;
                  for (size_t t = 0; t < nsh; ++t)
                    fl[sl * 256 + t % 256] |= (uint32_t)(bg.shell_flags[q * nsh + t] & 3u) << (2 * (t / 256));
                  std::copy(bg.coarse_idx.begin() + q * nc3, bg.coarse_idx.begin() + (q + 1) * nc3, ci.begin() + sl * nc3);
                  for (size_t t = 0; t < nc3; ++t)
                    if (bg.coarse_idx[q * nc3 + t] == INVALID_DOF)
                      fl[sl * 256 + t % 256] |= 1u << (16 + t / 256);
                }
              fused_flags.upload(fl);
              fused_coarse_idx.upload(ci);
              fused_tail_flags.upload(tt.tail_owned_by_fused);
              fused.group      = bg.fine_group;
              fused.flags      = fused_flags.p;
              fused.coarse_idx = fused_coarse_idx.p;
              fused.nc3        = (uint32_t)nc3;
              fused.E          = fec.embedding(1, f->tables->p);
              fused.tail_flags = fused_tail_flags.p;
              n_fused_bricks   = bg.n_bricks() - bg.n_unfused;
            }
          d->slot.upload(bg.slot);
          d->coarse_idx.upload(bg.coarse_idx);
          d->own_shell.upload(bg.own_shell);
          if (copy_hi > copy_lo)
            d->own_shell_restrict.upload(owned_only(bg.own_shell));
          bricks.push_back(std::move(d));
        }
      pc = tt.pc;
      pf = tt.pf;
      for (int k = 0; k < 3; ++k)
        {
          grp[k].kind      = k;
          grp[k].nf        = tt.groups[k].nf;
          grp[k].n_patches = tt.groups[k].n_patches();
          if (grp[k].n_patches)
            {
              grp[k].coarse_idx.upload(tt.groups[k].coarse_idx);
              grp[k].coarse_mask.upload(tt.groups[k].coarse_mask);
              grp[k].fine_idx.upload(tt.groups[k].fine_idx);
              if (copy_hi > copy_lo)
                grp[k].fine_idx_restrict.upload(owned_only(tt.groups[k].fine_idx));
              if (k == 1 && tt.pc == 1 && grp[k].nf == 3 && !getenv("MGAMD_NO_P1_PATCH_KERNELS"))
                {
                  if (copy_hi > copy_lo)
                    {
                      const std::vector<uint32_t> fr = owned_only(tt.groups[k].fine_idx);
                      grp[k].build_p1(tt.groups[k].coarse_idx, tt.groups[k].fine_idx, &fr);
                    }
                  else
                    grp[k].build_p1(tt.groups[k].coarse_idx, tt.groups[k].fine_idx, nullptr);
                }
            }
          grp[k].E = fec.embedding(k, pf);
        }
    }

    template <int PC, int NF, bool IDENTITY>
    void
    launch(const GroupD &g, const T *src, T *dst, bool prolongate)
    {
      using G = TransferGeo<PC, NF>;
      TransferArgs<T, PC, NF> a;
      a.coarse_idx  = g.coarse_idx.p;
      a.coarse_mask = g.coarse_mask.p;
      a.fine_idx    = (!prolongate && g.fine_idx_restrict.p) ? g.fine_idx_restrict.p : g.fine_idx.p;
      a.n_patches   = (uint32_t)g.n_patches;
      const int n   = PC + 1;
      for (int i = 0; i < n * n; ++i)
        {
          a.m.M[i] = a.m.K[i] = 0.0;
          a.m.I0[i]           = fec.I[0][i];
          a.m.I1[i]           = fec.I[1][i];
        }
      if ((int)g.E.size() != NF * n)
        throw std::runtime_error("transfer: embedding size mismatch");
      for (int i = 0; i < NF * n; ++i)
        a.E[i] = g.E[i];
      a.src            = src;
      a.dst            = dst;
      const size_t lds = 2 * (size_t)G::SPW * G::NF3 * sizeof(T);
      const int    grid = (int)((g.n_patches + G::SPW - 1) / G::SPW);
      if (prolongate)
        hipLaunchKernelGGL((prolongate_kernel<T, PC, NF, IDENTITY>), grid, G::BLOCK, lds, ctx->stream, a);
      else
        hipLaunchKernelGGL((restrict_kernel<T, PC, NF, IDENTITY>), grid, G::BLOCK, lds, ctx->stream, a);
      HIP_CHECK(hipGetLastError());
    }

    void
    launch_p1(const GroupD &g, const T *src, T *dst, bool prolongate)
    {
      PatchP1Args<T> a;
      a.uniq_ptr    = g.p1_uniq_ptr.p;
      a.uniq_idx    = g.p1_uniq_idx.p;
      a.loc         = g.p1_loc.p;
      a.coarse_mask = g.coarse_mask.p;
      a.fine_idx_t  = (!prolongate && g.p1_fine_t_restrict.p) ? g.p1_fine_t_restrict.p : g.p1_fine_t.p;
      a.n_patches   = (uint32_t)g.n_patches;
      a.max_uniq    = g.p1_max_uniq;
      for (int i = 0; i < 4; ++i)
        {
          a.m.M[i] = a.m.K[i] = 0.0;
          a.m.I0[i]           = fec.I[0][i];
          a.m.I1[i]           = fec.I[1][i];
        }
      a.src = src;
      a.dst = dst;
      const int    grid = (int)((g.n_patches + PATCH_P1_BLOCK - 1) / PATCH_P1_BLOCK);
      const size_t lds  = (size_t)std::max<uint32_t>(g.p1_max_uniq, 1) * sizeof(T);
      if (prolongate)
        hipLaunchKernelGGL(patch_p1_prolongate_kernel<T>, grid, PATCH_P1_BLOCK, lds, ctx->stream, a);
      else
        hipLaunchKernelGGL(patch_p1_restrict_kernel<T>, grid, PATCH_P1_BLOCK, lds, ctx->stream, a);
      HIP_CHECK(hipGetLastError());
    }

    // the first n_launch bricks of the group (all of them, or the un-fused ones of a fused group)
    template <int P, int B>
    void
    launch_brick(const BrickD &b, const T *src, T *dst, bool prolongate, size_t n_launch)
    {
      if (n_launch == 0)
        return;
      using G = BrickTransferGeo<P, B>;
      BrickTransferArgs<T, P> a;
      const GroupDev<T>      &fg = *fop->groups[b.fine_group];
      a.slot          = b.slot.p;
      a.interior_base = fg.interior_base.p;
      a.shell_pos     = fg.shell_pos.p;
      a.coarse_idx    = b.coarse_idx.p;
      a.own_shell     = (!prolongate && b.own_shell_restrict.p) ? b.own_shell_restrict.p : b.own_shell.p;
      a.n_bricks      = (uint32_t)n_launch;
      const std::vector<double> E = fec.embedding(1, P);
      for (int i = 0; i < (2 * P + 1) * (P + 1); ++i)
        a.E[i] = E[i];
      a.src            = src;
      a.dst            = dst;
      const size_t lds = (size_t)G::LDS * sizeof(T);
      if (prolongate)
        {
          auto kern = brick_prolongate_kernel<T, P, B>;
          ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), lds);
          hipLaunchKernelGGL(kern, (int)n_launch, G::BLOCK, lds, ctx->stream, a);
        }
      else if (G::NF == 17 && getenv("MGAMD_NO_PERSISTENT") == nullptr)
        {
          // persistent workgroups, three per CU (<= 168 VGPRs), a multiple of 8 (kernels.hpp)
          if constexpr (G::NF == 17)
            {
              auto kern = brick_restrict_persistent_kernel<T, P, B>;
              ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), lds);
              const int resident = std::max(8, 3 * ctx->n_cu / 8 * 8);
              hipLaunchKernelGGL(kern, std::min((int)n_launch, resident), G::BLOCK, lds, ctx->stream, a);
            }
        }
      else
        {
          auto kern = brick_restrict_kernel<T, P, B>;
          ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), lds);
          hipLaunchKernelGGL(kern, (int)n_launch, G::BLOCK, lds, ctx->stream, a);
        }
      HIP_CHECK(hipGetLastError());
    }

    template <int P>
    void
    dispatch_brick(const BrickD &b, const T *src, T *dst, bool prolongate, size_t n_launch)
    {
      switch (b.B)
        {
          case 2:
            if constexpr (P * 2 + 1 <= 17)
              return launch_brick<P, 2>(b, src, dst, prolongate, n_launch);
            break;
          case 4:
            if constexpr (P * 4 + 1 <= 17)
              return launch_brick<P, 4>(b, src, dst, prolongate, n_launch);
            break;
          case 8:
            if constexpr (P * 8 + 1 <= 17)
              return launch_brick<P, 8>(b, src, dst, prolongate, n_launch);
            break;
          case 16:
            if constexpr (P * 16 + 1 <= 17)
              return launch_brick<P, 16>(b, src, dst, prolongate, n_launch);
            break;
        }
      throw std::runtime_error("brick transfer: unsupported brick size");
    }

    // skip_fused: leave out the bricks whose transfer happens inside the operator passes
    void
    run(const T *src, T *dst, bool prolongate, bool skip_fused = false)
    {
      for (auto &b : bricks)
        {
          const size_t nb = skip_fused ? b->n_unfused : b->n_bricks;
          switch (pc)
          {
            case 1:
              dispatch_brick<1>(*b, src, dst, prolongate, nb);
              break;
            case 2:
              dispatch_brick<2>(*b, src, dst, prolongate, nb);
              break;
            case 3:
              dispatch_brick<3>(*b, src, dst, prolongate, nb);
              break;
            case 4:
              dispatch_brick<4>(*b, src, dst, prolongate, nb);
              break;
            default:
              throw std::runtime_error("brick transfer: degree not instantiated");
          }
        }
      for (int k = 0; k < 3; ++k)
        {
          const GroupD &g = grp[k];
          if (!g.n_patches)
            continue;
          const int key = pc * 100 + g.nf;
          if (k == 0)
            switch (pc)
              {
                case 1:
                  launch<1, 2, true>(g, src, dst, prolongate);
                  break;
                case 2:
                  launch<2, 3, true>(g, src, dst, prolongate);
                  break;
                case 3:
                  launch<3, 4, true>(g, src, dst, prolongate);
                  break;
                case 4:
                  launch<4, 5, true>(g, src, dst, prolongate);
                  break;
                default:
                  throw std::runtime_error("transfer: degree not instantiated");
              }
          else
            switch (key)
              {
                case 103:
                  if (g.p1_uniq_ptr.p)
                    launch_p1(g, src, dst, prolongate);
                  else
                    launch<1, 3, false>(g, src, dst, prolongate);
                  break;
                case 104:
                  launch<1, 4, false>(g, src, dst, prolongate);
                  break;
                case 205:
                  launch<2, 5, false>(g, src, dst, prolongate);
                  break;
                case 307:
                  launch<3, 7, false>(g, src, dst, prolongate);
                  break;
                case 409:
                  launch<4, 9, false>(g, src, dst, prolongate);
                  break;
                default:
                  throw std::runtime_error("transfer: (coarse degree, fine patch) combination not instantiated");
              }
        }
    }

    // Fused passes (fused_ready()): the residual step and the restriction of Multigrid::level_v_step in ONE operator pass
    // (ref:multigrid_throughput.cc:1093-1099 wiring; deal.II level_v_step: residual, restrict_and_add).  t receives the residual
    // rows that the un-fused patches still restrict (everything outside the fused bricks); the coarse defect receives the fused
    // bricks' part directly.  Call restrict_unfused_raw afterwards.
    void
    residual_restrict_raw(T *dst_coarse, T *t, const T *b, const T *x)
    {
      FusedTransferHost<T> f = fused;
      f.coarse               = dst_coarse;
      Epilogue<T> e{t, x, nullptr, b, nullptr, T(0), T(0), T(0)};
      fop->template apply<MODE_RESIDUAL_RESTRICT>(x, e, false, 0, 0, &f);
    }
    void
    restrict_unfused_raw(T *dst_coarse, const T *src_fine)
    {
      run(src_fine, dst_coarse, false, true);
      finish_restriction(dst_coarse);
    }
    // the un-fused part of the prolongation (in place); the fused bricks add theirs while the first post-smoothing pass
    // gathers x (Chebyshev::step_raw with this transfer)
    void
    prolongate_unfused_raw(T *dst_fine, const T *src_coarse)
    {
      run(src_coarse, dst_fine, true, true);
      if (fop->halo)
        fop->import_from_owner_raw(dst_fine + fop->tables->n_interior);
    }
    void
    prolongate_raw(T *dst_fine, const T *src_coarse)
    {
      run(src_coarse, dst_fine, true);
      // sharded runs: a rank that references a shared DoF only through hanging-node resolution has no patch writing its
      // copy, and different ranks evaluate the embedding through different patches: all copies take the owner's value
      if (fop->halo)
        fop->import_from_owner_raw(dst_fine + fop->tables->n_interior);
    }
    void
    restrict_raw(T *dst_coarse, const T *src_fine)
    {
      run(src_fine, dst_coarse, false);
      finish_restriction(dst_coarse);
    }
    void
    finish_restriction(T *dst_coarse)
    {
      // sharded runs: complete the coarse defect across ranks.  Onto a level that is cut into fewer pieces than the fine one (a
      // part held by a group of ranks, Partition tiers): every member of the group has restricted the cells of its own, so the
      // members' vectors are summed first; then the shared DoFs between the parts
      if (auto *sc = dynamic_cast<SubsetComm *>(cop->comm.get()))
        {
          auto     *sf = dynamic_cast<SubsetComm *>(fop->comm.get());
          const int gf = sf ? sf->group : (fop->comm ? 1 : sc->group);
          if (gf != sc->group)
            {
              if (gf != 1)
                throw std::runtime_error("restriction between two rank-subset tiers of different group sizes is not supported");
              sc->replica_sum(dst_coarse, cop->n_dofs(), (int)sizeof(T), ctx->stream);
            }
        }
      if (cop->halo)
        cop->exchange_add_raw(dst_coarse + cop->tables->n_interior);
      else if (fop->comm)
        fop->comm->allreduce_sum(dst_coarse, cop->n_dofs(), (int)sizeof(T), ctx->stream); // onto the replicated level
    }
    void
    prolongate_and_add(mgamd_vec &dst, const mgamd_vec &src) override
    {
      if (dst.n != fine->n_dofs() || src.n != coarse->n_dofs())
        throw std::invalid_argument("prolongate_and_add: vector size mismatch");
      prolongate_raw(dst.as<T>(), src.as<T>());
    }
    void
    restrict_and_add(mgamd_vec &dst, const mgamd_vec &src) override
    {
      if (dst.n != coarse->n_dofs() || src.n != fine->n_dofs())
        throw std::invalid_argument("restrict_and_add: vector size mismatch");
      restrict_raw(dst.as<T>(), src.as<T>());
    }
  };

  Transfer2Base *
  make_transfer2(LevelOperatorBase *fine, LevelOperatorBase *coarse)
  {
    if (fine->type != coarse->type)
      throw std::invalid_argument("transfer: level number types differ");
    if (fine->type == MGAMD_F64)
      return new Transfer2<double>(static_cast<LevelOperator<double> *>(fine), static_cast<LevelOperator<double> *>(coarse));
    return new Transfer2<float>(static_cast<LevelOperator<float> *>(fine), static_cast<LevelOperator<float> *>(coarse));
  }

  // ------------------------------------------------------------------------------------------
  // Algebraic multigrid coarse solver on the device (host setup: amg.hpp).  The reference's "amg" / "cg_with_amg" coarse solvers
  // (ref:multigrid_throughput.cc:945-1016) apply Trilinos ML to Operator::get_trilinos_system_matrix; this is an own
  // smoothed-aggregation V-cycle on the same matrix: CSR products fused with the Chebyshev update (kernels.hpp K7).
  // ------------------------------------------------------------------------------------------
  template <typename T>
  struct AmgDevice
  {
    struct Mat
    {
      DBuf<uint32_t> ptr, col;
      DBuf<T>        val;
      uint32_t       n_rows = 0, n_cols = 0;
      int            lanes  = 8;
      void
      upload(const CSR &A)
      {
        n_rows = A.n_rows;
        n_cols = A.n_cols;
        ptr.upload(A.ptr);
        col.upload(A.col.empty() ? std::vector<uint32_t>(1, 0) : A.col);
        std::vector<T> v(std::max<size_t>(A.val.size(), 1), T(0));
        std::copy(A.val.begin(), A.val.end(), v.begin());
        val.upload(v);
        const double avg = A.n_rows ? (double)A.nnz() / A.n_rows : 0.0;
        lanes            = avg <= 6 ? 4 : (avg <= 24 ? 8 : (avg <= 64 ? 16 : 32));
      }
    };
    struct Lvl
    {
      Mat      A, P, R;
      DBuf<T>  dinv, x, b, r, t;
      double   theta = 1, delta = 0;
      uint32_t n = 0;
    };
    Ctx                                        *ctx = nullptr;
    std::vector<std::unique_ptr<Lvl>>           lv;
    DBuf<double>                                coarse_inv;
    std::vector<std::pair<uint32_t, uint64_t>>  sizes; // rows, non-zeros per level
    unsigned                                    degree = 2; // Chebyshev smoother degree (MGAMD_AMG_SMOOTHER_DEGREE: development)

    AmgDevice(Ctx *c, const LevelTables &tables)
      : ctx(c)
    {
      if (const char *e = getenv("MGAMD_AMG_SMOOTHER_DEGREE"))
        degree = std::max(1, atoi(e));
      AmgHierarchyHost H = build_smoothed_aggregation(assemble_level_matrix(tables));
      if (H.levels.back().A.n_rows > 4096)
        throw std::runtime_error("AMG: coarsening stalled at " + std::to_string(H.levels.back().A.n_rows) + " rows");
      for (size_t l = 0; l < H.levels.size(); ++l)
        {
          auto L = std::make_unique<Lvl>();
          L->n   = H.levels[l].A.n_rows;
          L->A.upload(H.levels[l].A);
          if (l + 1 < H.levels.size())
            {
              L->P.upload(H.levels[l].P);
              L->R.upload(H.levels[l].R);
            }
          std::vector<T> d(H.levels[l].dinv.begin(), H.levels[l].dinv.end());
          L->dinv.upload(d);
          L->r.alloc(L->n);
          L->t.alloc(L->n);
          if (l > 0)
            {
              L->x.alloc(L->n);
              L->b.alloc(L->n);
            }
          // Chebyshev on [lambda_max / 20, lambda_max] (ML's "smoother: Chebyshev alpha" = 20)
          const double mx = H.levels[l].lambda_max, mn = mx / 20.0;
          L->theta = 0.5 * (mx + mn);
          L->delta = 0.5 * (mx - mn);
          sizes.push_back({L->n, H.levels[l].A.nnz()});
          lv.push_back(std::move(L));
        }
      coarse_inv.upload(H.coarse_inv);
    }

    template <int MODE>
    void
    spmv(const Mat &A, const T *x, T *y, const T *b = nullptr, const T *xold = nullptr, const T *dinv = nullptr, double f1 = 0, double f2 = 0)
    {
      if (!A.n_rows)
        return;
      auto launch = [&](auto lanes_tag) {
        constexpr int LANES = decltype(lanes_tag)::value;
        const int     grid  = (int)std::min<size_t>(((size_t)A.n_rows * LANES + 255) / 256, 4096);
        hipLaunchKernelGGL((csr_spmv_kernel<T, MODE, LANES>), grid, 256, 0, ctx->stream, A.n_rows, A.ptr.p, A.col.p, A.val.p, x, y, b, xold, dinv,
                           T(f1), T(f2));
      };
      switch (A.lanes)
        {
          case 4:
            launch(std::integral_constant<int, 4>());
            break;
          case 8:
            launch(std::integral_constant<int, 8>());
            break;
          case 16:
            launch(std::integral_constant<int, 16>());
            break;
          default:
            launch(std::integral_constant<int, 32>());
        }
      HIP_CHECK(hipGetLastError());
    }
    // Chebyshev of degree `degree` in D^-1 A, zero start (deal.II / ML recurrences): result in x (t: scratch)
    void
    smooth_zero(Lvl &L, T *x, T *t, const T *b)
    {
      T *cur = (degree % 2 == 1) ? x : t, *oth = (cur == x) ? t : x;
      hipLaunchKernelGGL(vec_scaled_product_kernel<T>, grid_for(L.n), 256, 0, ctx->stream, cur, T(1.0 / L.theta), L.dinv.p, b, (size_t)L.n);
      double rhok = L.delta / L.theta;
      const double sigma = L.theta / L.delta;
      for (unsigned j = 0; j + 1 < degree; ++j)
        {
          const double rhokp = 1.0 / (2.0 * sigma - rhok);
          spmv<SPMV_CHEB>(L.A, cur, oth, b, j == 0 ? nullptr : oth, L.dinv.p, rhokp * rhok, 2.0 * rhokp / L.delta); // x_old in place
          rhok = rhokp;
          std::swap(cur, oth);
        }
      // (degree - 1 swaps from the buffer chosen above: cur == x)
    }
    // general start x0 in x: result in x again
    void
    smooth_step(Lvl &L, T *x, T *t, const T *b)
    {
      T *cur = x, *oth = t;
      spmv<SPMV_CHEB>(L.A, cur, oth, b, nullptr, L.dinv.p, 0.0, 1.0 / L.theta);
      std::swap(cur, oth);
      double rhok = L.delta / L.theta;
      const double sigma = L.theta / L.delta;
      for (unsigned j = 0; j + 1 < degree; ++j)
        {
          const double rhokp = 1.0 / (2.0 * sigma - rhok);
          spmv<SPMV_CHEB>(L.A, cur, oth, b, oth, L.dinv.p, rhokp * rhok, 2.0 * rhokp / L.delta);
          rhok = rhokp;
          std::swap(cur, oth);
        }
      if (cur != x)
        HIP_CHECK(hipMemcpyAsync(x, cur, (size_t)L.n * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    }
    void
    cycle(size_t l, T *x, const T *b)
    {
      Lvl &L = *lv[l];
      if (l + 1 == lv.size())
        {
          hipLaunchKernelGGL(dense_matvec_kernel<T>, (int)std::min<uint32_t>(L.n, 1024), 256, 0, ctx->stream, coarse_inv.p, b, x, (int)L.n);
          return;
        }
      Lvl &C = *lv[l + 1];
      smooth_zero(L, x, L.t.p, b);
      spmv<SPMV_RESID>(L.A, x, L.r.p, b);
      spmv<SPMV_PLAIN>(L.R, L.r.p, C.b.p);
      cycle(l + 1, C.x.p, C.b.p);
      spmv<SPMV_ADD>(L.P, C.x.p, x);
      smooth_step(L, x, L.t.p, b);
    }
    // z = V(r): one V-cycle from a zero initial guess; z and r have the level's n_dofs entries
    void
    vcycle(T *z, const T *r)
    {
      cycle(0, z, r);
      HIP_CHECK(hipGetLastError());
    }
  };

  // ------------------------------------------------------------------------------------------
  // Multigrid V-cycle  (deal.II Multigrid::level_v_step + PreconditionMG::vmult)
  // ------------------------------------------------------------------------------------------
  template <typename T>
  struct MultigridT : MultigridBase
  {
    unsigned                         nl = 0;
    std::vector<LevelOperator<T> *>  ops;
    std::vector<Transfer2<T> *>      tr;
    std::vector<Chebyshev<T> *>      sm;
    double *wide_out          = nullptr;                                  // see vcycle_raw: the outer result vector while a cycle runs
    bool    wide_copy_from_mg = getenv("MGAMD_NO_WIDE_COPY_FROM_MG") == nullptr; // development switch
    std::vector<std::unique_ptr<DBuf<T>>> defect, S, Tb, res; // defect: only the finest level owns memory,
    DBuf<T>                               defect_slab;        // the coarser defects share one slab (ONE memset per cycle)
    std::vector<T *>                      dptr;               // defect vector of every level
    std::vector<T *>                 sol; // where the level solution currently lives
    // per-cycle views of the finest level: when the outer vectors have the level number type, r IS the
    // finest defect and z is one of the two smoother buffers (no copy_to_mg / copy_from_mg traffic)
    std::vector<const T *> dview;
    std::vector<T *>       sview, tview;
    std::string                      coarse_type;
    DBuf<double>                     coarse_inv; // dense inverse for "direct"
    // The V-cycle restricted to levels 0..collapse_level is a fixed linear map of that level's defect (zero start): on
    // levels this small every kernel is pure launch latency, so the map is tabulated once (n unit defects through the
    // regular level code) and applied as ONE dense matvec.  Not used while stage callbacks are installed (the
    // reference's per-level timers need the real stages).
    unsigned     collapse_level = 0;
    DBuf<double> collapse_M;
    bool         collapse_enabled = true;
    bool         fuse_transfers   = true; // level transfers inside the operator passes where the transfer offers them
    unsigned
    set_collapse(bool on) override
    {
      if (collapse_enabled != on)
        drop_graph(); // a captured cycle replays the work enqueued under the old setting
      collapse_enabled = on;
      return collapse_level;
    }
    void
    drop_graph()
    {
      if (graph_exec)
        (void)hipGraphExecDestroy(graph_exec);
      graph_exec = nullptr;
      graph_z = graph_r = nullptr;
    }
    DBuf<T>                          cg_r, cg_z, cg_p, cg_Ap;
    DBuf<double>                     cg_S; // device scalars of the coarse CG (device_pcg)
    hipGraphExec_t                   graph_exec = nullptr;
    const void                      *graph_z = nullptr, *graph_r = nullptr;

    // Local smoothing (`HMG-local`): the levels are the refinement levels of the octree; the outer vectors live on the
    // ACTIVE mesh and are copied level by level (copy_to_mg / copy_from_mg index pairs, transfer_tables.hpp)
    struct LsCopy
    {
      DBuf<uint32_t> gidx, lidx;
      uint32_t       n = 0;
    };
    std::vector<std::unique_ptr<LsCopy>> ls_copy;
    size_t                               ls_n_global = 0;
    const LevelTables                   *ls_active   = nullptr;
    const LevelTables *
    outer_tables() const override
    {
      return ls_active ? ls_active : ops[nl - 1]->tables.get();
    }
    void
    setup_local_smoothing(const LevelTables &active) override
    {
      if (collapse_level || nested)
        throw std::invalid_argument("local smoothing: not combinable with a nested coarse solver");
      ls_copy.clear();
      std::vector<uint32_t> g, l;
      size_t                total = 0;
      for (unsigned lv = 0; lv < nl; ++lv)
        {
          const LevelTables &T_l = *ops[lv]->tables;
          if (!T_l.ls_level || T_l.tria->cells.empty())
            throw std::invalid_argument("local smoothing: every level must be built with mgamd_dofs_create_level");
          ls_copy_indices(active, T_l, (int)T_l.tria->cells[0].level, g, l);
          auto c = std::make_unique<LsCopy>();
          c->n   = (uint32_t)g.size();
          if (c->n)
            {
              c->gidx.upload(g);
              c->lidx.upload(l);
            }
          total += g.size();
          ls_copy.push_back(std::move(c));
        }
      if (total != (size_t)active.n_interior + active.n_tail)
        throw std::runtime_error("local smoothing: the levels do not cover every unconstrained DoF of the active mesh exactly once");
      ls_n_global = active.n_dofs;
      ls_active   = &active;
      if (!defect[nl - 1]->p)
        defect[nl - 1]->alloc(ops[nl - 1]->n_dofs());
    }
    size_t
    n_outer() const
    {
      return ls_copy.empty() ? (size_t)ops[nl - 1]->n_dofs() : ls_n_global;
    }

    uint64_t       coarse_iterations = 0; // inner CG iterations of the coarse solver, accumulated
    std::unique_ptr<AmgDevice<T>> amg;  // coarse solvers "amg", "cg_with_amg"
    DBuf<T>                       amg_r, amg_z;
    MultigridBase *nested   = nullptr; // coarse solver "gmg_vcycle"
    unsigned       n_cycles = 1;

    LevelOperatorBase *
    finest_operator() const override
    {
      return ops[nl - 1];
    }
    void
    vcycle_level_raw(void *z, const void *r) override
    {
      vcycle_raw<T>(static_cast<T *>(z), static_cast<const T *>(r));
    }

    MultigridT(Ctx *c, unsigned n_levels, LevelOperatorBase *const *levels, Transfer2Base *const *transfers,
               ChebyshevBase *const *smoothers, const std::string &coarse, MultigridBase *nested_mg, unsigned nested_cycles)
    {
      ctx         = c;
      nl          = n_levels;
      number_type = (int)sizeof(T);
      nested      = nested_mg;
      n_cycles    = std::max(1u, nested_cycles);
      if (nl < 1)
        throw std::invalid_argument("multigrid: need at least one level");
      for (unsigned l = 0; l < nl; ++l)
        {
          if (!levels[l] || levels[l]->type != (int)sizeof(T))
            throw std::invalid_argument("multigrid: level operators must share one number type");
          ops.push_back(static_cast<LevelOperator<T> *>(levels[l]));
          tr.push_back(l > 0 ? static_cast<Transfer2<T> *>(transfers[l]) : nullptr);
          sm.push_back(smoothers[l] ? static_cast<Chebyshev<T> *>(smoothers[l]) : nullptr);
          if (l > 0 && (!transfers[l] || !smoothers[l]))
            throw std::invalid_argument("multigrid: missing transfer or smoother");
          if (l > 0 && (transfers[l]->fine != levels[l] || transfers[l]->coarse != levels[l - 1]))
            throw std::invalid_argument("multigrid: transfer does not connect the given levels");
          const size_t n = ops[l]->n_dofs();
          defect.emplace_back(new DBuf<T>);
          S.emplace_back(new DBuf<T>);
          Tb.emplace_back(new DBuf<T>);
          res.emplace_back(new DBuf<T>);
          if (l + 1 == nl)
            defect[l]->alloc(n);
          S[l]->alloc(n);
          Tb[l]->alloc(n);
          res[l]->alloc(n);
        }
      {
        // coarser defects: 256-byte aligned pieces of one allocation
        std::vector<size_t> off(nl, 0);
        size_t              total = 0;
        for (unsigned l = 0; l + 1 < nl; ++l)
          {
            off[l] = total;
            total += (ops[l]->n_dofs() + 31) / 32 * 32;
          }
        defect_slab.alloc(std::max<size_t>(total, 1));
        dptr.assign(nl, nullptr);
        for (unsigned l = 0; l + 1 < nl; ++l)
          dptr[l] = defect_slab.p + off[l];
        dptr[nl - 1] = defect[nl - 1]->p;
      }
      sol.assign(nl, nullptr);
      dview.assign(nl, nullptr);
      sview.assign(nl, nullptr);
      tview.assign(nl, nullptr);
      for (unsigned l = 0; l < nl; ++l)
        {
          dview[l] = dptr[l];
          sview[l] = S[l]->p;
          tview[l] = Tb[l]->p;
        }
      // ONE policy for the reference's Trilinos/PETSc choices ("amg", "cg_with_amg", "amg_petsc"):
      //   coarse level of <= 4096 DoFs (global coarsening ends on one cell): any AMG degenerates to an exact solve -> "direct";
      //   larger coarse level (PMG, HPMG with MinLevel): the library's own smoothed-aggregation AMG on the assembled level matrix
      //     (amg.hpp, AmgDevice) -> "amg" / "cg_with_amg" (amg_petsc: BoomerAMG's role, the same SA hierarchy -> "amg");
      //   a `nested` geometric multigrid on level 0's space, if the caller supplies one, takes the coarse solver's place as
      //     "gmg_vcycle" (the round-1/2 stand-in, still the only choice on a SHARDED coarse level).
      // Never a silent substitution: coarse_used names what runs (harness table column / bench JSON).
      coarse_type = coarse;
      const bool amg_like = coarse == "amg" || coarse == "cg_with_amg" || coarse == "amg_petsc";
      if (nested)
        {
          if (!amg_like && coarse != "gmg_vcycle")
            throw std::invalid_argument("multigrid: a nested multigrid is the stand-in for the AMG coarse solvers only");
          if (nested->number_type != (int)sizeof(T) || nested->outer_tables() != ops[0]->tables.get())
            throw std::invalid_argument("multigrid: the nested multigrid must act on the DoFs of this hierarchy's level 0 (same number type)");
          coarse_type = "gmg_vcycle";
        }
      else if (amg_like)
        {
          if (ops[0]->n_dofs() <= 4096)
            coarse_type = "direct";
          else
            {
              if (ops[0]->comm)
                throw std::runtime_error("CoarseGridSolverType '" + coarse + "' on a sharded coarse level: the algebraic multigrid is built "
                                         "from one rank's assembled matrix; use the nested geometric multigrid (gmg_vcycle), cg or "
                                         "cg_with_chebyshev");
              coarse_type = coarse == "cg_with_amg" ? "cg_with_amg" : "amg";
              amg         = std::make_unique<AmgDevice<T>>(ctx, *ops[0]->tables);
            }
        }
      coarse_used = coarse_type;
      if (coarse_type == "gmg_vcycle" || coarse_type == "amg")
        {
          if (n_cycles > 1)
            cg_r.alloc(ops[0]->n_dofs()), cg_z.alloc(ops[0]->n_dofs());
        }
      else if (coarse_type == "direct")
        setup_direct();
      else if (coarse_type == "cg" || coarse_type == "cg_with_chebyshev" || coarse_type == "cg_with_amg")
        {
          if (coarse_type == "cg_with_amg" && n_cycles > 1)
            amg_r.alloc(ops[0]->n_dofs()), amg_z.alloc(ops[0]->n_dofs());
          const size_t n = ops[0]->n_dofs();
          cg_r.alloc(n);
          cg_z.alloc(n);
          cg_p.alloc(n);
          cg_Ap.alloc(n);
          cg_S.alloc(8);
          if (coarse_type == "cg_with_chebyshev" && !sm[0])
            throw std::invalid_argument("multigrid: cg_with_chebyshev needs a smoother on level 0");
        }
      else
        throw std::invalid_argument("CoarseGridSolverType '" + coarse + "' not implemented");
      setup_collapse();
    }

    void
    setup_collapse()
    {
      size_t max_n = 2048;
      if (const char *e = getenv("MGAMD_COLLAPSE_MAX_DOFS")) // 0 disables
        max_n = (size_t)atol(e);
      for (unsigned l = 0; l < nl; ++l)
        if (ops[l]->tables->ls_level)
          return; // local smoothing: every level also receives its own part of the outer residual
      unsigned lc = 0;
      for (unsigned l = 1; l < nl; ++l)
        if (ops[l]->n_dofs() <= max_n && !ops[l]->comm)
          lc = l;
        else
          break;
      // an iterative coarse solver (cg, cg_with_chebyshev) is not a linear map of its right-hand side
      if (lc == 0 || ops[0]->comm || coarse_type != "direct")
        return;
      const size_t        n = ops[lc]->n_dofs();
      std::vector<T>      col(n);
      std::vector<double> M(n * n);
      const T             one = T(1);
      for (size_t j = 0; j < n; ++j)
        {
          // unit defect on level lc (coarser defects zero), the regular cycle below it
          if (lc + 1 < nl)
            defect_slab.zero(ctx->stream);
          else
            {
              HIP_CHECK(hipMemsetAsync(dptr[lc], 0, n * sizeof(T), ctx->stream));
              if (nl > 1)
                defect_slab.zero(ctx->stream);
            }
          HIP_CHECK(hipMemcpyAsync(dptr[lc] + j, &one, sizeof(T), hipMemcpyHostToDevice, ctx->stream));
          level_v_step(lc);
          HIP_CHECK(hipMemcpyAsync(col.data(), sol[lc], n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
          ctx->sync();
          for (size_t r = 0; r < n; ++r)
            M[r * n + j] = (double)col[r];
        }
      collapse_M.upload(M);
      collapse_level = lc;
    }

    ~MultigridT() override
    {
      release_stage_records();
      if (graph_exec)
        (void)hipGraphExecDestroy(graph_exec);
    }

    void
    setup_direct()
    {
      const size_t n = ops[0]->n_dofs();
      if (n > 4096)
        throw std::runtime_error("coarse level too large for the direct solver (" + std::to_string(n) +
                                 " DoFs): use CoarseGridSolverType cg or cg_with_chebyshev");
      // dense A_0 column by column through the level operator
      std::vector<double> A(n * n, 0.0);
      DBuf<T>             e, col;
      e.alloc(n);
      col.alloc(n);
      std::vector<T> h(n);
      for (size_t j = 0; j < n; ++j)
        {
          e.zero(ctx->stream);
          const T one = T(1);
          HIP_CHECK(hipMemcpyAsync(e.p + j, &one, sizeof(T), hipMemcpyHostToDevice, ctx->stream));
          ops[0]->vmult_raw(col.p, e.p);
          HIP_CHECK(hipMemcpyAsync(h.data(), col.p, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
          ctx->sync();
          for (size_t i = 0; i < n; ++i)
            A[i * n + j] = (double)h[i];
        }
      // symmetrise and invert by Gauss-Jordan with partial pivoting
      for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j)
          A[i * n + j] = A[j * n + i] = 0.5 * (A[i * n + j] + A[j * n + i]);
      std::vector<double> inv(n * n, 0.0);
      for (size_t i = 0; i < n; ++i)
        inv[i * n + i] = 1.0;
      for (size_t c = 0; c < n; ++c)
        {
          size_t piv = c;
          for (size_t r = c + 1; r < n; ++r)
            if (std::fabs(A[r * n + c]) > std::fabs(A[piv * n + c]))
              piv = r;
          if (std::fabs(A[piv * n + c]) < 1e-300)
            throw std::runtime_error("coarse matrix is singular");
          if (piv != c)
            for (size_t k = 0; k < n; ++k)
              {
                std::swap(A[c * n + k], A[piv * n + k]);
                std::swap(inv[c * n + k], inv[piv * n + k]);
              }
          const double d = 1.0 / A[c * n + c];
          for (size_t k = 0; k < n; ++k)
            {
              A[c * n + k] *= d;
              inv[c * n + k] *= d;
            }
          for (size_t r = 0; r < n; ++r)
            if (r != c)
              {
                const double f = A[r * n + c];
                if (f != 0.0)
                  for (size_t k = 0; k < n; ++k)
                    {
                      A[r * n + k] -= f * A[c * n + k];
                      inv[r * n + k] -= f * inv[c * n + k];
                    }
              }
        }
      coarse_inv.upload(inv);
    }

    void
    stage(int s, bool start, unsigned level)
    {
      if (cb)
        {
          ctx->sync();
          cb(s, start ? 1 : 0, level, cb_user);
        }
      if (stage_timing)
        {
          if (start)
            {
              if (stage_used == stage_records.size())
                {
                  StageRecord r{s, level, nullptr, nullptr};
                  HIP_CHECK(hipEventCreate(&r.e0));
                  HIP_CHECK(hipEventCreate(&r.e1));
                  stage_records.push_back(r);
                }
              stage_records[stage_used].stage = s;
              stage_records[stage_used].level = level;
              HIP_CHECK(hipEventRecord(stage_records[stage_used].e0, ctx->stream));
            }
          else
            HIP_CHECK(hipEventRecord(stage_records[stage_used++].e1, ctx->stream));
        }
    }
    unsigned
    n_levels() const override
    {
      return nl;
    }

    // z = n_cycles V-cycles of the algebraic multigrid applied to r (x = V(r); x += V(r - A x) ...: ML's `cycle applications`,
    // CoarseSolverNCycles); rr, zz: scratch of n entries (only used for n_cycles > 1)
    void
    amg_apply(T *z, const T *r, T *rr, T *zz)
    {
      const size_t n = ops[0]->n_dofs();
      amg->vcycle(z, r);
      for (unsigned c = 1; c < n_cycles; ++c)
        {
          ops[0]->residual_raw(rr, r, z);
          amg->vcycle(zz, rr);
          hipLaunchKernelGGL(vec_sadd_kernel<T>, grid_for(n), 256, 0, ctx->stream, z, T(1), T(1), zz, n);
        }
    }

    void
    coarse_cg(T *x, const T *b, int precond_kind) // 0 identity, 1 Chebyshev smoother of level 0, 2 algebraic multigrid
    {
      // SolverCG + ReductionControl(maxiter 10000, abstol 1e-20, reltol 1e-4): ref:multigrid_throughput.cc:888-895;
      // device-resident iteration, inner products over the GLOBAL vector (owned prefix + one scalar all-reduce on a
      // sharded level)
      LevelOperator<T> *op  = ops[0];
      const size_t      n   = op->n_dofs();
      const size_t      nd  = op->comm ? (size_t)op->tables->n_interior + op->tables->n_tail_owned : n;
      HIP_CHECK(hipMemcpyAsync(cg_r.p, b, n * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
      unsigned its = 0;
      double   res = 0;
      device_pcg<T>(
        ctx, op->comm.get(), cg_S.p, n, nd, x, cg_r.p, cg_z.p, cg_p.p, cg_Ap.p, [&](T *Ap, const T *p) { op->vmult_raw(Ap, p); },
        [&](T *z, const T *r) {
          if (precond_kind == 1)
            sm[0]->vmult_raw(z, sm[0]->tmp.p, r);
          else if (precond_kind == 2)
            amg_apply(z, r, amg_r.p, amg_z.p);
          else
            HIP_CHECK(hipMemcpyAsync(z, r, n * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
        },
        1e-4, 1e-20, 10000, its, res);
      coarse_iterations += its;
    }

    // the cycle on level vectors; defect[nl-1] must be set, coarser defects zero
    void
    level_v_step(unsigned l)
    {
      if (l == 0)
        {
          stage(3, true, 0);
          const size_t n = ops[0]->n_dofs();
          if (coarse_type == "direct")
            hipLaunchKernelGGL(dense_matvec_kernel<T>, (int)std::min<size_t>(n, 1024), 256, 0, ctx->stream, coarse_inv.p, dptr[0],
                               S[0]->p, (int)n);
          else if (coarse_type == "gmg_vcycle")
            {
              // x = V(b); then n_cycles - 1 corrections x += V(b - A x)   (AMG applied n_cycles times, ref CoarseSolverNCycles)
              nested->vcycle_level_raw(S[0]->p, dptr[0]);
              for (unsigned c = 1; c < n_cycles; ++c)
                {
                  ops[0]->residual_raw(cg_r.p, dptr[0], S[0]->p);
                  nested->vcycle_level_raw(cg_z.p, cg_r.p);
                  hipLaunchKernelGGL(vec_sadd_kernel<T>, grid_for(n), 256, 0, ctx->stream, S[0]->p, T(1), T(1), cg_z.p, n);
                }
            }
          else if (coarse_type == "amg")
            amg_apply(S[0]->p, dptr[0], cg_r.p, cg_z.p);
          else
            coarse_cg(S[0]->p, dptr[0], coarse_type == "cg_with_chebyshev" ? 1 : (coarse_type == "cg_with_amg" ? 2 : 0));
          sol[0] = S[0]->p;
          stage(3, false, 0);
          return;
        }
      if (l == collapse_level && !cb && collapse_enabled)
        {
          // the tabulated cycle below this level: timed as the coarse solve of level l
          const size_t n = ops[l]->n_dofs();
          if (stage_timing)
            stage(3, true, l);
          hipLaunchKernelGGL(dense_matvec_kernel<T>, (int)std::min<size_t>(n, 1024), 256, 0, ctx->stream, collapse_M.p, dview[l], sview[l],
                             (int)n);
          if (stage_timing)
            stage(3, false, l);
          sol[l] = sview[l];
          return;
        }
      // level transfers inside the operator passes (Transfer2::fused_ready; not while stage callbacks want the reference's
      // separate stages): stage 1 is then residual + fused restriction, stage 2 the restriction of the remaining patches, stage 4
      // the un-fused prolongation, and the fused prolongation is part of the first post-smoothing pass (stage 6)
      const bool fuse = fuse_transfers && !cb && tr[l]->fused_ready();
      stage(0, true, l);
      sm[l]->vmult_raw(sview[l], tview[l], dview[l]); // pre-smoothing, zero start
      stage(0, false, l);
      stage(1, true, l);
      if (fuse)
        tr[l]->residual_restrict_raw(dptr[l - 1], res[l]->p, dview[l], sview[l]);
      else
        ops[l]->residual_raw(res[l]->p, dview[l], sview[l]); // t = d - A x
      stage(1, false, l);
      stage(2, true, l);
      if (fuse)
        tr[l]->restrict_unfused_raw(dptr[l - 1], res[l]->p);
      else
        tr[l]->restrict_raw(dptr[l - 1], res[l]->p);
      stage(2, false, l);
      level_v_step(l - 1);
      stage(4, true, l);
      if (fuse)
        tr[l]->prolongate_unfused_raw(sview[l], sol[l - 1]);
      else
        tr[l]->prolongate_raw(sview[l], sol[l - 1]);
      stage(4, false, l);
      stage(5, true, l); // edge_prolongation: no-op for global coarsening (ref:multigrid_throughput.cc:1126-1130)
      if (ops[l]->tables->n_edge)
        {
          // local smoothing, Multigrid::set_edge_in_matrix (ref:multigrid_throughput.cc:1105,1130): the corrected solution on
          // the refinement edge couples into the interior: defect_l -= A_l^{edge unconstrained} (x_l restricted to the edge)
          const size_t n = ops[l]->n_dofs();
          ops[l]->interface_up_raw(res[l]->p, sview[l], tview[l]);
          hipLaunchKernelGGL(vec_sadd_kernel<T>, grid_for(n), 256, 0, ctx->stream, dptr[l], T(1), T(-1), res[l]->p, n);
        }
      stage(5, false, l);
      stage(6, true, l);
      // (finest level of float levels under double outer vectors: the last pass writes the doubles of copy_from_mg itself)
      double *wide = (l == nl - 1) ? wide_out : nullptr;
      if (fuse)
        {
          FusedTransferHost<T> f = tr[l]->fused;
          f.coarse               = sol[l - 1];
          f.scratch              = res[l]->p; // the residual vector is free again
          sol[l]                 = sm[l]->step_raw(sview[l], tview[l], dview[l], &f, wide);
        }
      else
        sol[l] = sm[l]->step_raw(sview[l], tview[l], dview[l], nullptr, wide); // post-smoothing
      stage(6, false, l);
    }

    template <typename TO>
    void
    vcycle_ls_raw(TO *z, const TO *r)
    {
      const unsigned L = nl - 1;
      stage(7, true, L); // copy_to_mg: every level receives the residual entries of ITS active cells
      if (nl > 1)
        defect_slab.zero(ctx->stream);
      HIP_CHECK(hipMemsetAsync(dptr[L], 0, (size_t)ops[L]->n_dofs() * sizeof(T), ctx->stream));
      dview[L] = dptr[L];
      sview[L] = S[L]->p;
      tview[L] = Tb[L]->p;
      for (unsigned l = 0; l < nl; ++l)
        if (ls_copy[l]->n)
          hipLaunchKernelGGL((indexed_copy_kernel<T, TO>), grid_for(ls_copy[l]->n), 256, 0, ctx->stream, dptr[l], ls_copy[l]->lidx.p, r,
                             ls_copy[l]->gidx.p, ls_copy[l]->n);
      stage(7, false, L);
      level_v_step(L);
      stage(8, true, L); // copy_from_mg
      HIP_CHECK(hipMemsetAsync(z, 0, ls_n_global * sizeof(TO), ctx->stream));
      for (unsigned l = 0; l < nl; ++l)
        if (ls_copy[l]->n)
          hipLaunchKernelGGL((indexed_copy_kernel<TO, T>), grid_for(ls_copy[l]->n), 256, 0, ctx->stream, z, ls_copy[l]->gidx.p,
                             (const T *)sol[l], ls_copy[l]->lidx.p, ls_copy[l]->n);
      stage(8, false, L);
      HIP_CHECK(hipGetLastError());
    }

    template <typename TO>
    void
    vcycle_raw(TO *z, const TO *r)
    {
      if (!ls_copy.empty())
        {
          vcycle_ls_raw<TO>(z, r);
          return;
        }
      const size_t   n    = ops[nl - 1]->n_dofs();
      const unsigned L    = nl - 1;
      const bool     same = sizeof(TO) == sizeof(T) && nl > 1;
      // copy_to_mg: defect_L = cast(r), coarser defects zero
      stage(7, true, L);
      if (same)
        {
          // r is the finest defect; z takes the role of the smoother buffer the post-smoother ends in
          // (Chebyshev::step_raw returns its second buffer for odd degree, the first for even degree)
          dview[L] = reinterpret_cast<const T *>(r);
          if (sm[L]->degree % 2 == 1)
            {
              sview[L] = S[L]->p;
              tview[L] = reinterpret_cast<T *>(z);
            }
          else
            {
              sview[L] = reinterpret_cast<T *>(z);
              tview[L] = S[L]->p;
            }
        }
      else
        {
          dview[L] = defect[L]->p;
          sview[L] = S[L]->p;
          tview[L] = Tb[L]->p;
          if (sizeof(TO) == sizeof(T))
            hipLaunchKernelGGL((vec_copy_kernel<T, TO>), grid_for(n), 256, 0, ctx->stream, defect[L]->p, r, n);
          else
            hipLaunchKernelGGL((vec_copy_kernel<T, TO>), grid_for(n), 256, 0, ctx->stream, defect[L]->p, r, n);
        }
      if (nl > 1)
        defect_slab.zero(ctx->stream);
      stage(7, false, L);
      // copy_from_mg inside the last post-smoothing pass (Epilogue::out_wide): float levels, double outer vectors, a smoother of
      // degree >= 2 (its last pass is a plain Chebyshev pass), no stage callbacks (they want the reference's separate stages)
      wide_out = nullptr;
      if constexpr (sizeof(T) == 4 && sizeof(TO) == 8)
        if (nl > 1 && !cb && wide_copy_from_mg)
          wide_out = reinterpret_cast<double *>(z);
      level_v_step(L);
      stage(8, true, L);
      // (sol[L] == nullptr: the last smoothing pass has written z; a collapsed finest level leaves its result in a level vector)
      if (sol[L] != nullptr && (const void *)sol[L] != (const void *)z)
        hipLaunchKernelGGL((vec_copy_kernel<TO, T>), grid_for(n), 256, 0, ctx->stream, z, sol[L], n);
      wide_out = nullptr;
      stage(8, false, L);
    }

    void
    vcycle(mgamd_vec &z, const mgamd_vec &r) override
    {
      const size_t n = n_outer();
      if (z.n != n || r.n != n || z.type != r.type || z.data == r.data)
        throw std::invalid_argument("PreconditionMG::vmult: bad vectors");
      if (z.type == MGAMD_F64)
        vcycle_raw<double>(z.as<double>(), r.as<double>());
      else
        vcycle_raw<float>(z.as<float>(), r.as<float>());
    }

    double
    time_vcycles(mgamd_vec &z, const mgamd_vec &r, unsigned n, bool use_graph) override
    {
      if (n == 0)
        return 0.0;
      if (cb || stage_timing)
        throw std::invalid_argument("time_vcycles: remove the stage callback / stage timing first");
      const bool graphable = coarse_type == "direct" && !ops[nl - 1]->comm; // (nested cycles are not captured)
      vcycle(z, r); // warm-up: sets kernel attributes, touches memory
      ctx->sync();
      if (use_graph && graphable && (!graph_exec || graph_z != z.data || graph_r != r.data))
        {
          if (graph_exec)
            {
              (void)hipGraphExecDestroy(graph_exec);
              graph_exec = nullptr;
            }
          hipGraph_t graph;
          HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
          const bool saved = ctx->profile;
          ctx->profile     = false;
          try
            {
              vcycle(z, r);
            }
          catch (...)
            {
              ctx->profile = saved;
              (void)hipStreamEndCapture(ctx->stream, &graph);
              throw;
            }
          ctx->profile = saved;
          HIP_CHECK(hipStreamEndCapture(ctx->stream, &graph));
          HIP_CHECK(hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0));
          HIP_CHECK(hipGraphDestroy(graph));
          graph_z = z.data;
          graph_r = r.data;
          HIP_CHECK(hipGraphLaunch(graph_exec, ctx->stream));
          ctx->sync();
        }
      hipEvent_t e0, e1;
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      HIP_CHECK(hipEventRecord(e0, ctx->stream));
      for (unsigned i = 0; i < n; ++i)
        {
          if (use_graph && graphable)
            HIP_CHECK(hipGraphLaunch(graph_exec, ctx->stream));
          else
            vcycle(z, r);
        }
      HIP_CHECK(hipEventRecord(e1, ctx->stream));
      HIP_CHECK(hipEventSynchronize(e1));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      return (double)ms / n;
    }
  };

  MultigridBase *
  make_multigrid(Ctx *ctx, unsigned n_levels, LevelOperatorBase *const *levels, Transfer2Base *const *transfers,
                 ChebyshevBase *const *smoothers, const std::string &coarse_solver, MultigridBase *nested, unsigned n_cycles)
  {
    if (!n_levels || !levels || !levels[0])
      throw std::invalid_argument("multigrid: no levels");
    if (levels[0]->type == MGAMD_F64)
      return new MultigridT<double>(ctx, n_levels, levels, transfers, smoothers, coarse_solver, nested, n_cycles);
    return new MultigridT<float>(ctx, n_levels, levels, transfers, smoothers, coarse_solver, nested, n_cycles);
  }

  // ------------------------------------------------------------------------------------------
  // Outer solver: deal.II SolverCG + ReductionControl (ref:multigrid_throughput.cc:1140-1147,1625-1635)
  // ------------------------------------------------------------------------------------------
  template <typename T>
  static void
  solve_cg_T(LevelOperatorBase &A, MultigridBase *M, mgamd_vec &x, const mgamd_vec &b, double reltol, double abstol, unsigned maxiter,
             unsigned &n_iterations, double &residual)
  {
    Ctx         *ctx = A.ctx;
    const size_t n   = A.n_dofs();
    auto        *op  = static_cast<LevelOperator<T> *>(&A);
    const size_t nd  = op->comm ? (size_t)op->tables->n_interior + op->tables->n_tail_owned : n;
    std::unique_ptr<mgamd_vec> g(vec_create(ctx, n, A.type)), h(vec_create(ctx, n, A.type)), d(vec_create(ctx, n, A.type)),
      Ad(vec_create(ctx, n, A.type));
    vec_copy(*g, b); // residual r = b - A*0; dst = 0 (ref:multigrid_throughput.cc:1142,1245) is set by device_pcg
    // the V-cycle works on whole vectors (it may alias them as level buffers): hand it the mgamd_vec objects
    device_pcg<T>(
      ctx, op->comm.get(), ctx->d_cg, n, nd, x.as<T>(), g->as<T>(), h->as<T>(), d->as<T>(), Ad->as<T>(), [&](T *Ap, const T *p) { op->vmult_raw(Ap, p); },
      [&](T *z, const T *r) {
        (void)z;
        (void)r;
        if (M)
          M->vcycle(*h, *g);
        else
          vec_copy(*h, *g);
      },
      reltol, abstol, maxiter, n_iterations, residual);
    ctx->sync();
  }

  void
  solve_cg(LevelOperatorBase &A, MultigridBase *M, mgamd_vec &x, const mgamd_vec &b, double reltol, double abstol, unsigned maxiter,
           unsigned &n_iterations, double &residual)
  {
    const size_t n = A.n_dofs();
    if (x.n != n || b.n != n || x.type != A.type || b.type != A.type)
      throw std::invalid_argument("SolverCG::solve: bad vectors");
    if (A.type == MGAMD_F64)
      solve_cg_T<double>(A, M, x, b, reltol, abstol, maxiter, n_iterations, residual);
    else
      solve_cg_T<float>(A, M, x, b, reltol, abstol, maxiter, n_iterations, residual);
  }
} // namespace mgamd
