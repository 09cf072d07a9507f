// 1D finite-element tables on the reference interval [0,1] for FE_Q(p) with
// Gauss-Lobatto support points and QGauss(p+1) quadrature -- the 1D building blocks of
// everything deal.II's FEEvaluation does for the reference operator
// (ref:include/operator.h:22 `FEEvaluation<dim,-1,0,n_components,number>`;
//  ref:multigrid_throughput.cc:1560-1562 `FE_Q<dim>{degree}`, `QGauss<dim>(fe.degree+1)`).
//
// Every cell of every mesh this benchmark can produce is an axis-aligned cube
// (ref:include/grid_generator.h:11,42,75,104 hyper_cube + MappingQ1), so the cell
// stiffness matrix is  h * (K (x) M (x) M + M (x) K (x) M + M (x) M (x) K)  with the 1D
// matrices below; no per-quadrature-point geometry exists on this path.
#pragma once
#include <cmath>
#include <vector>

namespace mgamd
{
  constexpr int MAX_DEGREE = 7;

  struct FE1D
  {
    int                 p = 0;
    std::vector<double> nodes;  // p+1 Gauss-Lobatto points
    std::vector<double> xq, wq; // p+1 Gauss points / weights
    std::vector<double> S, G;   // [q*(p+1)+a] shape values / derivatives at Gauss points
    std::vector<double> M, K;   // [(p+1)^2] mass / stiffness (quadrature-evaluated, exact)
    std::vector<double> m;      // [p+1] int phi_a
    std::vector<double> I[2];   // [(p+1)^2] hanging-node interpolation: I[c][a*(p+1)+b] = phi_b((x_a+c)/2)

    static long double
    legendre(int n, long double x, long double *dp = nullptr)
    {
      long double p0 = 1.0L, p1 = x;
      if (n == 0)
        {
          if (dp)
            *dp = 0;
          return 1.0L;
        }
      for (int k = 2; k <= n; ++k)
        {
          long double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
          p0             = p1;
          p1             = pk;
        }
      if (dp)
        *dp = n * (x * p1 - p0) / (x * x - 1.0L);
      return p1;
    }

    // values of the Lagrange basis on `nd` at point x
    static void
    lagrange(const std::vector<double> &nd, double x, double *val, double *der = nullptr)
    {
      const int n = nd.size();
      for (int a = 0; a < n; ++a)
        {
          long double den = 1.0L, v = 1.0L, s = 0.0L;
          for (int b = 0; b < n; ++b)
            if (b != a)
              {
                den *= (long double)nd[a] - nd[b];
                v *= (long double)x - nd[b];
              }
          if (der)
            for (int c = 0; c < n; ++c)
              {
                if (c == a)
                  continue;
                long double t = 1.0L;
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

    explicit FE1D(int degree)
      : p(degree)
    {
      const int         n  = p + 1;
      const long double pi = 3.14159265358979323846264338327950288L;
      nodes.assign(n, 0.0);
      nodes[0] = 0.0;
      nodes[p] = 1.0;
      // interior GLL nodes: roots of P'_p on (-1,1)
      for (int i = 1; i < p; ++i)
        {
          long double x = -std::cos(pi * i / p);
          for (int it = 0; it < 100; ++it)
            {
              // f = P'_p(x), f' = P''_p = (2x P'_p - p(p+1) P_p)/(1-x^2)
              long double dp, pp = legendre(p, x, &dp);
              long double ddp = (2 * x * dp - (long double)p * (p + 1) * pp) / (1 - x * x);
              long double dx  = dp / ddp;
              x -= dx;
              if (std::fabs((double)dx) < 1e-19)
                break;
            }
          nodes[i] = (double)(0.5L * (x + 1.0L));
        }
      // symmetrise
      for (int i = 0; i <= p / 2; ++i)
        {
          double a     = 0.5 * (nodes[i] + (1.0 - nodes[p - i]));
          nodes[i]     = a;
          nodes[p - i] = 1.0 - a;
        }
      if (p % 2 == 0)
        nodes[p / 2] = 0.5;
      // Gauss points
      xq.assign(n, 0.0);
      wq.assign(n, 0.0);
      for (int i = 0; i < n; ++i)
        {
          long double x = -std::cos(pi * (i + 0.75L) / (n + 0.5L));
          long double dp;
          for (int it = 0; it < 100; ++it)
            {
              long double pn = legendre(n, x, &dp);
              long double dx = pn / dp;
              x -= dx;
              if (std::fabs((double)dx) < 1e-19)
                break;
            }
          legendre(n, x, &dp);
          xq[i] = (double)(0.5L * (x + 1.0L));
          wq[i] = (double)(1.0L / ((1 - x * x) * dp * dp)); // = 0.5 * 2/((1-x^2)P'^2)
        }
      S.assign(n * n, 0.0);
      G.assign(n * n, 0.0);
      for (int q = 0; q < n; ++q)
        lagrange(nodes, xq[q], &S[q * n], &G[q * n]);
      M.assign(n * n, 0.0);
      K.assign(n * n, 0.0);
      m.assign(n, 0.0);
      for (int a = 0; a < n; ++a)
        {
          for (int b = 0; b < n; ++b)
            {
              long double sm = 0, sk = 0;
              for (int q = 0; q < n; ++q)
                {
                  sm += (long double)wq[q] * S[q * n + a] * S[q * n + b];
                  sk += (long double)wq[q] * G[q * n + a] * G[q * n + b];
                }
              M[a * n + b] = (double)sm;
              K[a * n + b] = (double)sk;
            }
          long double s = 0;
          for (int q = 0; q < n; ++q)
            s += (long double)wq[q] * S[q * n + a];
          m[a] = (double)s;
        }
      for (int c = 0; c < 2; ++c)
        {
          I[c].assign(n * n, 0.0);
          for (int a = 0; a < n; ++a)
            {
              lagrange(nodes, 0.5 * (nodes[a] + c), &I[c][a * n]);
              // exact identity row where the child node coincides with the parent's end node
              if ((c == 0 && a == 0) || (c == 1 && a == p))
                for (int b = 0; b < n; ++b)
                  I[c][a * n + b] = (b == a) ? 1.0 : 0.0;
            }
        }
    }

    // embedding of this (coarse) space into a fine 1D space: rows = fine nodes
    //  kind 0: identity (same cell, same degree)          -> (p+1) x (p+1)
    //  kind 1: h-refinement, two children of degree p     -> (2p+1) x (p+1)
    //  kind 2: p-refinement, same cell, fine degree pf    -> (pf+1) x (p+1)
    std::vector<double>
    embedding(int kind, int pf) const
    {
      const int           n = p + 1;
      std::vector<double> P;
      if (kind == 0)
        {
          P.assign(n * n, 0.0);
          for (int a = 0; a < n; ++a)
            P[a * n + a] = 1.0;
        }
      else if (kind == 1)
        {
          P.assign((2 * p + 1) * n, 0.0);
          for (int a = 0; a <= p; ++a)
            for (int b = 0; b < n; ++b)
              {
                P[a * n + b]       = I[0][a * n + b];
                P[(p + a) * n + b] = I[1][a * n + b];
              }
        }
      else
        {
          FE1D fine(pf);
          P.assign((pf + 1) * n, 0.0);
          for (int a = 0; a <= pf; ++a)
            {
              lagrange(nodes, fine.nodes[a], &P[a * n]);
              if (a == 0 || a == pf)
                for (int b = 0; b < n; ++b)
                  P[a * n + b] = (b == (a == 0 ? 0 : p)) ? 1.0 : 0.0;
            }
        }
      return P;
    }
  };
} // namespace mgamd
