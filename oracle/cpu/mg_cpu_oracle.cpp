// ORACLE (test infrastructure, NOT product code): CPU restatement in C++/OpenMP of the hot path of
// peterrum/dealii-multigrid's multigrid_throughput.cc.  PARITY UNPINNED: deal.II, where the reference's
// arithmetic lives, is not available here (see oracle/mgoracle.py and DESIGN.md); this file restates
// the algorithms deal.II runs at the reference's call sites, in the reference's own formulation:
//
//   cell integral ....... ref:include/operator.h:461-472  gather_evaluate(gradients) -> for q:
//                         submit_gradient(get_gradient(q)) -> integrate_scatter(gradients), i.e.
//                         sum-factorised evaluation at QGauss(p+1) points (12 one-dimensional sweeps),
//                         NOT the product's brick/tensor-matrix formulation
//   vmult ............... ref:include/operator.h:152-183 (zero dst, cell loop, dst[c] = src[c] on constrained rows)
//   inverse diagonal .... ref:include/operator.h:228-242 (unit vectors per cell, 1e-10 guard)
//   smoother ............ ref:multigrid_throughput.cc:867-883 + deal.II PreconditionChebyshev
//   transfer ............ ref:multigrid_throughput.cc:1600-1604 + deal.II MGTwoLevelTransfer
//   V-cycle ............. ref:multigrid_throughput.cc:1093-1133 + deal.II Multigrid::level_v_step
//   outer CG ............ ref:multigrid_throughput.cc:1140-1147 + deal.II SolverCG/ReductionControl
//
// It consumes plain index tables (per-cell gathered DoF indices with hanging entities resolved to the
// parent's DoFs, per-cell constraint masks, per-patch transfer index lists) so that it can be timed
// as the host-CPU baseline at sizes the numpy oracle cannot reach.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include <omp.h>

namespace
{
  constexpr uint32_t INVALID = 0xFFFFFFFFu;
  constexpr int      MAXN    = 8; // p <= 7
  using vec                  = std::vector<double>;

  // ---------------------------------------------------------------- 1D tables
  long double
  legendre(int n, long double x, long double *dp)
  {
    long double p0 = 1, p1 = x;
    if (n == 0)
      {
        *dp = 0;
        return 1;
      }
    for (int k = 2; k <= n; ++k)
      {
        long double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
        p0             = p1;
        p1             = pk;
      }
    *dp = n * (x * p1 - p0) / (x * x - 1);
    return p1;
  }

  struct FE
  {
    int    p, n;
    double nodes[MAXN], xq[MAXN], wq[MAXN];
    double S[MAXN * MAXN];  // S[q][a] shape values at Gauss points
    double D[MAXN * MAXN];  // collocation derivative: dU/dx(q) = sum_r D[q][r] U(r) for U given at Gauss points
    double I[2][MAXN * MAXN];

    static void
    lagrange(const double *nd, int n, double x, double *val, double *der)
    {
      for (int a = 0; a < n; ++a)
        {
          long double den = 1, v = 1, s = 0;
          for (int b = 0; b < n; ++b)
            if (b != a)
              {
                den *= (long double)nd[a] - nd[b];
                v *= (long double)x - nd[b];
              }
          for (int c = 0; c < n; ++c)
            {
              if (c == a)
                continue;
              long double t = 1;
              for (int b = 0; b < n; ++b)
                if (b != a && b != c)
                  t *= (long double)x - nd[b];
              s += t;
            }
          val[a] = (double)(v / den);
          if (der)
            der[a] = (double)(s / den);
        }
    }

    explicit FE(int degree)
      : p(degree)
      , n(degree + 1)
    {
      const long double pi = 3.14159265358979323846264338327950288L;
      nodes[0]             = 0;
      nodes[p]             = 1;
      for (int i = 1; i < p; ++i)
        { // roots of P'_p: bisection-safe Newton from Chebyshev guesses
          long double x = -std::cos(pi * i / p);
          for (int it = 0; it < 200; ++it)
            {
              long double d1, pp = legendre(p, x, &d1);
              long double d2 = (2 * x * d1 - (long double)p * (p + 1) * pp) / (1 - x * x);
              long double dx = d1 / d2;
              x -= dx;
              if (std::fabs((double)dx) < 1e-19)
                break;
            }
          nodes[i] = (double)((x + 1) / 2);
        }
      for (int i = 0; i < n; ++i)
        {
          long double x = -std::cos(pi * (i + 0.75L) / (n + 0.5L)), d1;
          for (int it = 0; it < 200; ++it)
            {
              long double pn = legendre(n, x, &d1);
              long double dx = pn / d1;
              x -= dx;
              if (std::fabs((double)dx) < 1e-19)
                break;
            }
          legendre(n, x, &d1);
          xq[i] = (double)((x + 1) / 2);
          wq[i] = (double)(1 / ((1 - x * x) * d1 * d1));
        }
      double G[MAXN * MAXN];
      for (int q = 0; q < n; ++q)
        lagrange(nodes, n, xq[q], &S[q * n], &G[q * n]);
      // D = G S^{-1}: derivative of the Lagrange basis ON THE GAUSS POINTS evaluated at the Gauss points
      for (int q = 0; q < n; ++q)
        {
          double dummy[MAXN];
          lagrange(xq, n, xq[q], dummy, &D[q * n]);
        }
      (void)G;
      for (int c = 0; c < 2; ++c)
        for (int a = 0; a < n; ++a)
          lagrange(nodes, n, 0.5 * (nodes[a] + c), &I[c][a * n], nullptr);
    }
  };

  // one 1D sweep of a (n x n) matrix along direction `dir` of an n^3 array (x fastest)
  template <bool TRANSPOSE, bool ADD>
  inline void
  sweep(const double *M, int n, int dir, const double *in, double *out)
  {
    const int stride = dir == 0 ? 1 : (dir == 1 ? n : n * n);
    for (int o2 = 0; o2 < n; ++o2)
      for (int o1 = 0; o1 < n; ++o1)
        {
          int base;
          if (dir == 0)
            base = (o2 * n + o1) * n;
          else if (dir == 1)
            base = o2 * n * n + o1;
          else
            base = o2 * n + o1;
          double tmp[MAXN];
          for (int a = 0; a < n; ++a)
            {
              double s = 0;
              for (int b = 0; b < n; ++b)
                s += (TRANSPOSE ? M[b * n + a] : M[a * n + b]) * in[base + b * stride];
              tmp[a] = s;
            }
          for (int a = 0; a < n; ++a)
            if (ADD)
              out[base + a * stride] += tmp[a];
            else
              out[base + a * stride] = tmp[a];
        }
  }

  inline void
  hanging(const FE &fe, uint16_t mask, double *v, bool transpose)
  {
    if (!(mask >> 3))
      return;
    const int p = fe.p, n = fe.n;
    const int cp[3]     = {mask & 1, (mask >> 1) & 1, (mask >> 2) & 1};
    const int stride[3] = {1, n, n * n};
    for (int dd = 0; dd < 3; ++dd)
      {
        const int     d = transpose ? 2 - dd : dd, e = (d + 1) % 3, f = (d + 2) % 3;
        const bool    fce = (mask >> (3 + e)) & 1, fcf = (mask >> (3 + f)) & 1, edg = (mask >> (6 + d)) & 1;
        const double *Ic  = fe.I[cp[d]];
        for (int ae = 0; ae < n; ++ae)
          for (int af = 0; af < n; ++af)
            {
              const bool one = ae == cp[e] * p, onf = af == cp[f] * p;
              if (!((fce && one) || (fcf && onf) || (edg && one && onf)))
                continue;
              double *line = v + ae * stride[e] + af * stride[f];
              double  tmp[MAXN];
              for (int a = 0; a < n; ++a)
                {
                  double s = 0;
                  for (int b = 0; b < n; ++b)
                    s += (transpose ? Ic[b * n + a] : Ic[a * n + b]) * line[b * stride[d]];
                  tmp[a] = s;
                }
              for (int a = 0; a < n; ++a)
                line[a * stride[d]] = tmp[a];
            }
      }
  }

  // ---------------------------------------------------------------- level operator
  struct Level
  {
    FE                    fe;
    int                   n3;
    uint64_t              n_cells;
    uint32_t              n_dofs, first_constrained;
    std::vector<uint32_t> cell_dofs;
    std::vector<double>   cell_h;
    std::vector<uint16_t> cell_mask;
    std::vector<std::vector<uint32_t>> colors; // cells of one colour share no DoF
    vec                   inv_diag;

    Level(int p, uint64_t nc, uint32_t nd, uint32_t fc, const uint32_t *cd, const uint8_t *lev, const uint16_t *mask)
      : fe(p)
      , n3((p + 1) * (p + 1) * (p + 1))
      , n_cells(nc)
      , n_dofs(nd)
      , first_constrained(fc)
      , cell_dofs(cd, cd + nc * n3)
      , cell_mask(mask, mask + nc)
    {
      cell_h.resize(nc);
      for (uint64_t c = 0; c < nc; ++c)
        cell_h[c] = 2.0 / (double)(1u << lev[c]);
      // greedy colouring with a per-DoF bit set of used colours
      std::vector<uint64_t> used(nd, 0);
      for (uint64_t c = 0; c < nc; ++c)
        {
          uint64_t m = 0;
          for (int t = 0; t < n3; ++t)
            {
              const uint32_t d = cell_dofs[c * n3 + t];
              if (d != INVALID)
                m |= used[d];
            }
          int col = 0;
          while (col < 63 && ((m >> col) & 1))
            ++col;
          if (colors.size() <= (size_t)col)
            colors.resize(col + 1);
          colors[col].push_back((uint32_t)c);
          for (int t = 0; t < n3; ++t)
            {
              const uint32_t d = cell_dofs[c * n3 + t];
              if (d != INVALID)
                used[d] |= 1ull << col;
            }
        }
    }

    // local operator on gathered values u (in place result in r): the reference's do_cell_integral_global
    void
    cell_apply(uint64_t c, double *u, double *r) const
    {
      const int n = fe.n;
      double    U[MAXN * MAXN * MAXN], t1[MAXN * MAXN * MAXN], g[MAXN * MAXN * MAXN];
      hanging(fe, cell_mask[c], u, false);
      // evaluate: values at quadrature points (collocation basis), then the three gradient components
      sweep<false, false>(fe.S, n, 0, u, t1);
      sweep<false, false>(fe.S, n, 1, t1, U);
      sweep<false, false>(fe.S, n, 2, U, t1); // t1 = values at Gauss points
      const double h = cell_h[c];
      // per quadrature point: grad_ref -> J^{-T} grad_ref (1/h) ; submit: * J^{-1} * JxW = (1/h)*(h^3 w) => h * w_q
      std::fill(U, U + n3, 0.0);
      for (int d = 0; d < 3; ++d)
        {
          sweep<false, false>(fe.D, n, d, t1, g); // reference gradient component d at all q
          for (int qz = 0; qz < n; ++qz)
            for (int qy = 0; qy < n; ++qy)
              for (int qx = 0; qx < n; ++qx)
                g[(qz * n + qy) * n + qx] *= h * fe.wq[qx] * fe.wq[qy] * fe.wq[qz];
          sweep<true, true>(fe.D, n, d, g, U); // integrate: test with collocation gradients
        }
      // back to the nodal basis
      sweep<true, false>(fe.S, n, 2, U, t1);
      sweep<true, false>(fe.S, n, 1, t1, U);
      sweep<true, false>(fe.S, n, 0, U, r);
      hanging(fe, cell_mask[c], r, true);
    }

    void
    vmult(double *dst, const double *src) const
    {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < (int64_t)n_dofs; ++i)
        dst[i] = 0.0;
      for (const auto &col : colors)
        {
#pragma omp parallel for schedule(static)
          for (int64_t k = 0; k < (int64_t)col.size(); ++k)
            {
              const uint64_t  c   = col[k];
              const uint32_t *idx = &cell_dofs[c * n3];
              double          u[MAXN * MAXN * MAXN], r[MAXN * MAXN * MAXN];
              for (int t = 0; t < n3; ++t)
                u[t] = idx[t] != INVALID ? src[idx[t]] : 0.0;
              cell_apply(c, u, r);
              for (int t = 0; t < n3; ++t)
                if (idx[t] != INVALID)
                  dst[idx[t]] += r[t];
            }
        }
#pragma omp parallel for schedule(static)
      for (int64_t i = first_constrained; i < (int64_t)n_dofs; ++i)
        dst[i] = src[i]; // ref:include/operator.h:170-172
    }

    void
    compute_inverse_diagonal()
    {
      inv_diag.assign(n_dofs, 0.0);
      for (const auto &col : colors)
        {
#pragma omp parallel for schedule(static)
          for (int64_t k = 0; k < (int64_t)col.size(); ++k)
            {
              const uint64_t  c   = col[k];
              const uint32_t *idx = &cell_dofs[c * n3];
              double          u[MAXN * MAXN * MAXN], r[MAXN * MAXN * MAXN];
              for (int j = 0; j < n3; ++j)
                {
                  if (idx[j] == INVALID)
                    continue;
                  std::fill(u, u + n3, 0.0);
                  u[j] = 1.0;
                  cell_apply(c, u, r);
                  inv_diag[idx[j]] += r[j];
                }
            }
        }
      for (auto &d : inv_diag)
        d = std::fabs(d) > 1e-10 ? 1.0 / d : 1.0; // ref:include/operator.h:240-241
    }
  };

  // ---------------------------------------------------------------- vector helpers
  double
  dot(const vec &a, const vec &b)
  {
    double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int64_t i = 0; i < (int64_t)a.size(); ++i)
      s += a[i] * b[i];
    return s;
  }

  // ---------------------------------------------------------------- Chebyshev
  double
  tridiag_max_eig(const vec &alpha, const vec &beta, double *mn)
  {
    // Sturm-sequence bisection on the Lanczos matrix
    const int n = alpha.size();
    vec       d(n), e(n, 0.0);
    for (int j = 0; j < n; ++j)
      {
        d[j] = 1.0 / alpha[j] + (j > 0 ? beta[j - 1] / alpha[j - 1] : 0.0);
        if (j + 1 < n)
          e[j] = std::sqrt(beta[j]) / alpha[j];
      }
    double lo = d[0], hi = d[0];
    for (int j = 0; j < n; ++j)
      {
        const double r = (j > 0 ? std::fabs(e[j - 1]) : 0) + (j + 1 < n ? std::fabs(e[j]) : 0);
        lo             = std::min(lo, d[j] - r);
        hi             = std::max(hi, d[j] + r);
      }
    auto count_below = [&](double x) {
      int    cnt = 0;
      double q   = 1;
      for (int j = 0; j < n; ++j)
        {
          q = d[j] - x - (j > 0 ? e[j - 1] * e[j - 1] / q : 0.0);
          if (q == 0)
            q = 1e-300;
          if (q < 0)
            ++cnt;
        }
      return cnt;
    };
    auto kth = [&](int k) { // k-th smallest eigenvalue (0-based)
      double a = lo, b = hi;
      for (int it = 0; it < 200; ++it)
        {
          const double m = 0.5 * (a + b);
          if (count_below(m) > k)
            b = m;
          else
            a = m;
        }
      return 0.5 * (a + b);
    };
    if (mn)
      *mn = kth(0);
    return kth(n - 1);
  }

  struct Cheb
  {
    const Level *L;
    int          k;
    double       theta = 1, delta = 0, max_eig = 1, min_eig = 1;
    Cheb(const Level *lv, int degree, double smoothing_range, int n_it)
      : L(lv)
      , k(degree)
    {
      const size_t n = L->n_dofs;
      vec          r(n), z(n), d(n), Ad(n);
      double       sum = 0;
      for (size_t i = 0; i < n; ++i)
        sum += (double)(i % 11);
      const double mean = sum / n;
      for (size_t i = 0; i < n; ++i)
        r[i] = (double)(i % 11) - mean;
      vec alphas, betas;
      if (std::sqrt(dot(r, r)) > 0 && n_it > 0)
        {
          for (size_t i = 0; i < n; ++i)
            d[i] = z[i] = L->inv_diag[i] * r[i];
          double rz = dot(r, z);
          for (int it = 0; it < n_it; ++it)
            {
              L->vmult(Ad.data(), d.data());
              const double dAd = dot(d, Ad);
              if (!(dAd > 0))
                break;
              const double alpha = rz / dAd;
              for (size_t i = 0; i < n; ++i)
                r[i] -= alpha * Ad[i];
              alphas.push_back(alpha);
              if (std::sqrt(dot(r, r)) <= 1e-10)
                break;
              for (size_t i = 0; i < n; ++i)
                z[i] = L->inv_diag[i] * r[i];
              const double rzn = dot(r, z), beta = rzn / rz;
              betas.push_back(beta);
              rz = rzn;
              for (size_t i = 0; i < n; ++i)
                d[i] = z[i] + beta * d[i];
            }
        }
      if (!alphas.empty())
        {
          betas.resize(alphas.size(), 0.0);
          max_eig = tridiag_max_eig(alphas, betas, &min_eig);
        }
      max_eig *= 1.2;
      const double a = smoothing_range > 1 ? max_eig / smoothing_range : std::min(0.9 * max_eig, min_eig);
      delta          = 0.5 * (max_eig - a);
      theta          = 0.5 * (max_eig + a);
    }

    void
    iterate(vec &x, vec &xold, const vec &b, vec &t) const
    {
      if (k < 2 || std::fabs(delta) < 1e-40)
        return;
      const size_t n    = x.size();
      double       rhok = delta / theta, sigma = theta / delta;
      for (int j = 0; j + 1 < k; ++j)
        {
          const double rhokp = 1.0 / (2.0 * sigma - rhok), f1 = rhokp * rhok, f2 = 2.0 * rhokp / delta;
          rhok = rhokp;
          L->vmult(t.data(), x.data());
#pragma omp parallel for schedule(static)
          for (int64_t i = 0; i < (int64_t)n; ++i)
            {
              const double xn = x[i] + f1 * (x[i] - xold[i]) + f2 * L->inv_diag[i] * (b[i] - t[i]);
              xold[i]         = x[i];
              x[i]            = xn;
            }
        }
    }
    void
    vmult(vec &x, const vec &b) const
    {
      const size_t n = b.size();
      vec          xold(n, 0.0), t(n);
      for (size_t i = 0; i < n; ++i)
        x[i] = L->inv_diag[i] * b[i] / theta;
      iterate(x, xold, b, t);
    }
    void
    step(vec &x, const vec &b) const
    {
      const size_t n = b.size();
      vec          xold(x), t(n);
      L->vmult(t.data(), x.data());
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < (int64_t)n; ++i)
        x[i] = xold[i] + L->inv_diag[i] * (b[i] - t[i]) / theta;
      iterate(x, xold, b, t);
    }
  };

  // ---------------------------------------------------------------- transfer
  struct TGroup
  {
    int                   kind, nf;
    uint64_t              n;
    std::vector<uint32_t> coarse_idx, fine_idx;
    std::vector<uint16_t> coarse_mask;
    vec                   E; // nf x (pc+1)
  };
  struct Transfer
  {
    const Level *fine, *coarse;
    TGroup       g[3];

    void
    embed(const TGroup &G, const double *c, double *f, bool transpose) const
    {
      // f (nf^3) = E (x) E (x) E c ((pc+1)^3), or c += transpose applied to f
      const int nc = coarse->fe.n, nf = G.nf;
      double    t1[9 * 9 * 9], t2[9 * 9 * 9];
      if (!transpose)
        {
          for (int z = 0; z < nc; ++z)
            for (int y = 0; y < nc; ++y)
              for (int X = 0; X < nf; ++X)
                {
                  double s = 0;
                  for (int x = 0; x < nc; ++x)
                    s += G.E[X * nc + x] * c[(z * nc + y) * nc + x];
                  t1[(z * nc + y) * nf + X] = s;
                }
          for (int z = 0; z < nc; ++z)
            for (int Y = 0; Y < nf; ++Y)
              for (int X = 0; X < nf; ++X)
                {
                  double s = 0;
                  for (int y = 0; y < nc; ++y)
                    s += G.E[Y * nc + y] * t1[(z * nc + y) * nf + X];
                  t2[(z * nf + Y) * nf + X] = s;
                }
          for (int Z = 0; Z < nf; ++Z)
            for (int Y = 0; Y < nf; ++Y)
              for (int X = 0; X < nf; ++X)
                {
                  double s = 0;
                  for (int z = 0; z < nc; ++z)
                    s += G.E[Z * nc + z] * t2[(z * nf + Y) * nf + X];
                  f[(Z * nf + Y) * nf + X] = s;
                }
        }
      else
        {
          double *cc = const_cast<double *>(c);
          for (int z = 0; z < nc; ++z)
            for (int Y = 0; Y < nf; ++Y)
              for (int X = 0; X < nf; ++X)
                {
                  double s = 0;
                  for (int Z = 0; Z < nf; ++Z)
                    s += G.E[Z * nc + z] * f[(Z * nf + Y) * nf + X];
                  t2[(z * nf + Y) * nf + X] = s;
                }
          for (int z = 0; z < nc; ++z)
            for (int y = 0; y < nc; ++y)
              for (int X = 0; X < nf; ++X)
                {
                  double s = 0;
                  for (int Y = 0; Y < nf; ++Y)
                    s += G.E[Y * nc + y] * t2[(z * nf + Y) * nf + X];
                  t1[(z * nc + y) * nf + X] = s;
                }
          for (int z = 0; z < nc; ++z)
            for (int y = 0; y < nc; ++y)
              for (int x = 0; x < nc; ++x)
                {
                  double s = 0;
                  for (int X = 0; X < nf; ++X)
                    s += G.E[X * nc + x] * t1[(z * nc + y) * nf + X];
                  cc[(z * nc + y) * nc + x] = s;
                }
        }
    }

    void
    prolongate_and_add(vec &dst, const vec &src) const
    {
      const int nc3 = coarse->n3;
      for (int k = 0; k < 3; ++k)
        {
          const TGroup &G   = g[k];
          const int     nf3 = G.nf * G.nf * G.nf;
#pragma omp parallel for schedule(static)
          for (int64_t pch = 0; pch < (int64_t)G.n; ++pch)
            {
              double c[MAXN * MAXN * MAXN], f[9 * 9 * 9];
              for (int t = 0; t < nc3; ++t)
                {
                  const uint32_t i = G.coarse_idx[pch * nc3 + t];
                  c[t]             = i != INVALID ? src[i] : 0.0;
                }
              hanging(coarse->fe, G.coarse_mask[pch], c, false);
              embed(G, c, f, false);
              for (int t = 0; t < nf3; ++t)
                {
                  const uint32_t i = G.fine_idx[pch * nf3 + t];
                  if (i != INVALID)
                    dst[i] += f[t]; // every fine DoF is owned by exactly one patch
                }
            }
        }
    }
    void
    restrict_and_add(vec &dst, const vec &src) const
    {
      const int nc3 = coarse->n3;
      for (int k = 0; k < 3; ++k)
        {
          const TGroup &G   = g[k];
          const int     nf3 = G.nf * G.nf * G.nf;
#pragma omp parallel for schedule(static)
          for (int64_t pch = 0; pch < (int64_t)G.n; ++pch)
            {
              double c[MAXN * MAXN * MAXN], f[9 * 9 * 9];
              for (int t = 0; t < nf3; ++t)
                {
                  const uint32_t i = G.fine_idx[pch * nf3 + t];
                  f[t]             = i != INVALID ? src[i] : 0.0;
                }
              embed(G, c, f, true);
              hanging(coarse->fe, G.coarse_mask[pch], c, true);
              for (int t = 0; t < nc3; ++t)
                {
                  const uint32_t i = G.coarse_idx[pch * nc3 + t];
                  if (i != INVALID)
                    {
#pragma omp atomic
                      dst[i] += c[t];
                    }
                }
            }
        }
    }
  };

  // ---------------------------------------------------------------- multigrid + CG
  int
  pcg(const Level &A, const std::function<void(vec &, const vec &)> *prec, vec &x, const vec &b, double reltol, double abstol,
      int maxiter, double *final_res);

  struct MG
  {
    std::vector<const Level *>         levels;
    std::vector<const Transfer *>      tr;
    std::vector<std::unique_ptr<Cheb>> sm;
    std::string                        coarse;
    vec                                A0inv; // dense inverse for "direct"
    std::vector<vec>                   defect, sol, t;

    void
    setup_direct()
    {
      const Level &L = *levels[0];
      const size_t n = L.n_dofs;
      vec          A(n * n), e(n), col(n);
      for (size_t j = 0; j < n; ++j)
        {
          std::fill(e.begin(), e.end(), 0.0);
          e[j] = 1;
          L.vmult(col.data(), e.data());
          for (size_t i = 0; i < n; ++i)
            A[i * n + j] = col[i];
        }
      A0inv.assign(n * n, 0.0);
      for (size_t i = 0; i < n; ++i)
        A0inv[i * n + i] = 1;
      for (size_t c = 0; c < n; ++c)
        {
          size_t piv = c;
          for (size_t r = c + 1; r < n; ++r)
            if (std::fabs(A[r * n + c]) > std::fabs(A[piv * n + c]))
              piv = r;
          for (size_t k = 0; k < n; ++k)
            {
              std::swap(A[c * n + k], A[piv * n + k]);
              std::swap(A0inv[c * n + k], A0inv[piv * n + k]);
            }
          const double d = 1.0 / A[c * n + c];
          for (size_t k = 0; k < n; ++k)
            {
              A[c * n + k] *= d;
              A0inv[c * n + k] *= d;
            }
          for (size_t r = 0; r < n; ++r)
            if (r != c && A[r * n + c] != 0.0)
              {
                const double f = A[r * n + c];
                for (size_t k = 0; k < n; ++k)
                  {
                    A[r * n + k] -= f * A[c * n + k];
                    A0inv[r * n + k] -= f * A0inv[c * n + k];
                  }
              }
        }
    }

    void
    coarse_solve(vec &x, const vec &d)
    {
      const Level &L = *levels[0];
      const size_t n = L.n_dofs;
      if (coarse == "direct")
        {
          for (size_t i = 0; i < n; ++i)
            {
              double s = 0;
              for (size_t j = 0; j < n; ++j)
                s += A0inv[i * n + j] * d[j];
              x[i] = s;
            }
          return;
        }
      std::fill(x.begin(), x.end(), 0.0);
      if (coarse == "cg")
        pcg(L, nullptr, x, d, 1e-4, 1e-20, 10000, nullptr);
      else
        {
          std::function<void(vec &, const vec &)> pr = [&](vec &z, const vec &r) { sm[0]->vmult(z, r); };
          pcg(L, &pr, x, d, 1e-4, 1e-20, 10000, nullptr);
        }
    }

    void
    vcycle(vec &z, const vec &r)
    {
      const int nl = levels.size();
      for (int l = 0; l < nl; ++l)
        std::fill(defect[l].begin(), defect[l].end(), 0.0);
      defect[nl - 1] = r;
      for (int l = nl - 1; l > 0; --l)
        {
          sm[l]->vmult(sol[l], defect[l]);
          levels[l]->vmult(t[l].data(), sol[l].data());
          const size_t n = t[l].size();
#pragma omp parallel for schedule(static)
          for (int64_t i = 0; i < (int64_t)n; ++i)
            t[l][i] = defect[l][i] - t[l][i];
          tr[l]->restrict_and_add(defect[l - 1], t[l]);
        }
      coarse_solve(sol[0], defect[0]);
      for (int l = 1; l < nl; ++l)
        {
          tr[l]->prolongate_and_add(sol[l], sol[l - 1]);
          sm[l]->step(sol[l], defect[l]);
        }
      z = sol[nl - 1];
    }
  };

  int
  pcg(const Level &A, const std::function<void(vec &, const vec &)> *prec, vec &x, const vec &b, double reltol, double abstol,
      int maxiter, double *final_res)
  {
    const size_t n = A.n_dofs;
    vec          g(b), h(n), d(n), Ad(n);
    std::fill(x.begin(), x.end(), 0.0);
    double       res  = std::sqrt(dot(g, g));
    const double res0 = res;
    int          it   = 0;
    if (final_res)
      *final_res = res;
    if (res <= abstol)
      return 0;
    if (prec)
      (*prec)(h, g);
    else
      h = g;
    d         = h;
    double gh = dot(g, h);
    while (it < maxiter)
      {
        ++it;
        A.vmult(Ad.data(), d.data());
        const double alpha = gh / dot(d, Ad);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; ++i)
          {
            x[i] += alpha * d[i];
            g[i] -= alpha * Ad[i];
          }
        res = std::sqrt(dot(g, g));
        if (final_res)
          *final_res = res;
        if (res < reltol * res0 || res <= abstol)
          break;
        if (prec)
          (*prec)(h, g);
        else
          h = g;
        const double ghn = dot(g, h), beta = ghn / gh;
        gh = ghn;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; ++i)
          d[i] = h[i] + beta * d[i];
      }
    return it;
  }
} // namespace

extern "C" {

void *
mgo_level_create(int p, uint64_t n_cells, uint32_t n_dofs, uint32_t first_constrained, const uint32_t *cell_dofs, const uint8_t *cell_level,
                 const uint16_t *cell_mask)
{
  Level *L = new Level(p, n_cells, n_dofs, first_constrained, cell_dofs, cell_level, cell_mask);
  L->compute_inverse_diagonal();
  return L;
}
void
mgo_level_destroy(void *l)
{
  delete static_cast<Level *>(l);
}
int
mgo_level_n_colors(void *l)
{
  return (int)static_cast<Level *>(l)->colors.size();
}
void
mgo_level_vmult(void *l, double *dst, const double *src)
{
  static_cast<Level *>(l)->vmult(dst, src);
}
void
mgo_level_inverse_diagonal(void *l, double *d)
{
  const Level *L = static_cast<Level *>(l);
  std::memcpy(d, L->inv_diag.data(), L->n_dofs * sizeof(double));
}

void *
mgo_transfer_create(void *fine, void *coarse)
{
  Transfer *T = new Transfer;
  T->fine     = static_cast<Level *>(fine);
  T->coarse   = static_cast<Level *>(coarse);
  for (int k = 0; k < 3; ++k)
    {
      T->g[k].kind = k;
      T->g[k].n    = 0;
      T->g[k].nf   = 1;
    }
  return T;
}
void
mgo_transfer_set_group(void *t, int kind, uint64_t n, int nf, const uint32_t *coarse_idx, const uint16_t *coarse_mask,
                       const uint32_t *fine_idx)
{
  Transfer *T  = static_cast<Transfer *>(t);
  TGroup   &G  = T->g[kind];
  const int nc = T->coarse->fe.n, pc = T->coarse->fe.p;
  G.kind       = kind;
  G.n          = n;
  G.nf         = nf;
  G.coarse_idx.assign(coarse_idx, coarse_idx + n * nc * nc * nc);
  G.coarse_mask.assign(coarse_mask, coarse_mask + n);
  G.fine_idx.assign(fine_idx, fine_idx + n * (uint64_t)nf * nf * nf);
  G.E.assign((size_t)nf * nc, 0.0);
  const FE &fc = T->coarse->fe;
  if (kind == 0)
    for (int a = 0; a < nc; ++a)
      G.E[a * nc + a] = 1.0;
  else if (kind == 1)
    for (int a = 0; a <= pc; ++a)
      {
        FE::lagrange(fc.nodes, nc, 0.5 * fc.nodes[a], &G.E[a * nc], nullptr);
        FE::lagrange(fc.nodes, nc, 0.5 + 0.5 * fc.nodes[a], &G.E[(pc + a) * nc], nullptr);
      }
  else
    for (int a = 0; a < nf; ++a)
      FE::lagrange(fc.nodes, nc, T->fine->fe.nodes[a], &G.E[a * nc], nullptr);
}
void
mgo_transfer_destroy(void *t)
{
  delete static_cast<Transfer *>(t);
}
void
mgo_transfer_prolongate_and_add(void *t, double *dst, const double *src)
{
  const Transfer *T = static_cast<Transfer *>(t);
  vec             d(dst, dst + T->fine->n_dofs), s(src, src + T->coarse->n_dofs);
  T->prolongate_and_add(d, s);
  std::memcpy(dst, d.data(), d.size() * sizeof(double));
}
void
mgo_transfer_restrict_and_add(void *t, double *dst, const double *src)
{
  const Transfer *T = static_cast<Transfer *>(t);
  vec             d(dst, dst + T->coarse->n_dofs), s(src, src + T->fine->n_dofs);
  T->restrict_and_add(d, s);
  std::memcpy(dst, d.data(), d.size() * sizeof(double));
}

void *
mgo_mg_create(int n_levels, void **levels, void **transfers, int smoother_degree, double smoothing_range, int eig_cg_n_iterations,
              const char *coarse)
{
  MG *M = new MG;
  for (int l = 0; l < n_levels; ++l)
    {
      M->levels.push_back(static_cast<Level *>(levels[l]));
      M->tr.push_back(l > 0 ? static_cast<Transfer *>(transfers[l]) : nullptr);
      M->sm.emplace_back(new Cheb(M->levels[l], smoother_degree, smoothing_range, eig_cg_n_iterations));
      const size_t n = M->levels[l]->n_dofs;
      M->defect.emplace_back(n);
      M->sol.emplace_back(n);
      M->t.emplace_back(n);
    }
  M->coarse = coarse;
  if (M->coarse == "amg" || M->coarse == "cg_with_amg" || M->coarse == "amg_petsc")
    M->coarse = "direct";
  if (M->coarse == "direct")
    M->setup_direct();
  return M;
}
void
mgo_mg_destroy(void *m)
{
  delete static_cast<MG *>(m);
}
double
mgo_mg_max_eigenvalue(void *m, int level)
{
  return static_cast<MG *>(m)->sm[level]->max_eig;
}
void
mgo_mg_vcycle(void *m, double *z, const double *r)
{
  MG          *M = static_cast<MG *>(m);
  const size_t n = M->levels.back()->n_dofs;
  vec          zz(n), rr(r, r + n);
  M->vcycle(zz, rr);
  std::memcpy(z, zz.data(), n * sizeof(double));
}
// times `n` V-cycles, returns seconds per cycle
double
mgo_mg_time_vcycles(void *m, const double *r, int n)
{
  MG          *M  = static_cast<MG *>(m);
  const size_t nd = M->levels.back()->n_dofs;
  vec          zz(nd), rr(r, r + nd);
  M->vcycle(zz, rr); // warm-up
  const double t0 = omp_get_wtime();
  for (int i = 0; i < n; ++i)
    M->vcycle(zz, rr);
  return (omp_get_wtime() - t0) / n;
}
int
mgo_solve_cg(void *level, void *m, double *x, const double *b, double reltol, double abstol, int maxiter, double *residual)
{
  const Level                            *L = static_cast<Level *>(level);
  MG                                     *M = static_cast<MG *>(m);
  vec                                     xx(L->n_dofs), bb(b, b + L->n_dofs);
  std::function<void(vec &, const vec &)> pr = [&](vec &z, const vec &r) { M->vcycle(z, r); };
  const int                               it = pcg(*L, M ? &pr : nullptr, xx, bb, reltol, abstol, maxiter, residual);
  std::memcpy(x, xx.data(), xx.size() * sizeof(double));
  return it;
}
int
mgo_num_threads(void)
{
  return omp_get_max_threads();
}
void
mgo_set_num_threads(int n)
{
  omp_set_num_threads(n);
}
}
