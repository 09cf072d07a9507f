// Algebraic multigrid coarse solver, host side: the assembled level matrix and a smoothed-aggregation hierarchy.
//
// The reference's default CoarseGridSolverType is "amg" (ref:scripts/default.json:11): one (CoarseSolverNCycles) V-cycle(s) of
// Trilinos ML built on Operator::get_trilinos_system_matrix (ref:multigrid_throughput.cc:945-1016, ref:include/operator.h:244-287:
// MatrixFreeTools::compute_matrix of the cell kernel with the constraints).  Trilinos is not available here and ML's exact
// aggregates cannot be reproduced, so this is an OWN smoothed-aggregation AMG of the same family (ML's default for elliptic
// problems): same role, same inputs (the assembled matrix), not the same numbers -- its iteration counts cannot be
// parity-checked against ML.
//
//   matrix      A = C^T K C + I on the constrained rows: per cell  h K_ref  (K_ref = K(x)M(x)M + M(x)K(x)M + M(x)M(x)K, the
//               Gauss(p+1) cell kernel in closed form on cubes), with the in-cell hanging-node interpolation applied from both
//               sides, scattered through the cell's gathered DoF indices
//   strength    |a_ij| >= theta sqrt(a_ii a_jj), theta = 1e-4 (deal.II PreconditionAMG::AdditionalData::aggregation_threshold)
//   aggregates  Vanek's greedy passes: roots whose strong neighbourhood is free; leftovers join the strongest neighbouring
//               aggregate; the rest form aggregates of their own.  Decoupled rows (constrained DoFs) stay out.
//   prolongator P = (I - omega D^-1 A) P_tent, P_tent piecewise constant and column-normalised, omega = 4 / (3 lambda_max(D^-1 A))
//   coarse      A_c = P^T A P (Galerkin), recursively until <= max_coarse rows, then a dense inverse
//   smoother    Chebyshev of degree 2 in D^-1 A on [lambda_max / 20, lambda_max] (ML's Chebyshev defaults: alpha = 20), zero
//               start before, general start after the coarse correction: a symmetric V-cycle, usable inside CG
#pragma once
#include "level_tables.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace mgamd
{
  struct CSR
  {
    uint32_t              n_rows = 0, n_cols = 0;
    std::vector<uint32_t> ptr, col;
    std::vector<double>   val;
    size_t
    nnz() const
    {
      return col.size();
    }
  };

  // Operator::get_trilinos_system_matrix (ref:include/operator.h:244-287) without Trilinos
  inline CSR
  assemble_level_matrix(const LevelTables &L)
  {
    if (L.n_edge)
      throw std::invalid_argument("assemble_level_matrix: local-smoothing levels are not supported");
    const FE1D &fe = L.fe;
    const int   p = L.p, n = p + 1, n3 = n * n * n;
    // reference element matrix, x fastest
    std::vector<double> Kref((size_t)n3 * n3);
    for (int c = 0; c < n; ++c)
      for (int b = 0; b < n; ++b)
        for (int a = 0; a < n; ++a)
          for (int c2 = 0; c2 < n; ++c2)
            for (int b2 = 0; b2 < n; ++b2)
              for (int a2 = 0; a2 < n; ++a2)
                Kref[(size_t)((c * n + b) * n + a) * n3 + (c2 * n + b2) * n + a2] =
                  fe.K[a * n + a2] * fe.M[b * n + b2] * fe.M[c * n + c2] + fe.M[a * n + a2] * fe.K[b * n + b2] * fe.M[c * n + c2] +
                  fe.M[a * n + a2] * fe.M[b * n + b2] * fe.K[c * n + c2];
    // element matrices of hanging configurations, cached by mask: I^T K_ref I
    std::map<uint16_t, std::vector<double>> cache;
    auto                                    element = [&](uint16_t mask) -> const std::vector<double> & {
      if (!(mask >> MASK_FACE_SHIFT))
        return Kref;
      auto it = cache.find(mask);
      if (it != cache.end())
        return it->second;
      std::vector<double> Ke((size_t)n3 * n3), v(n3), w(n3);
      for (int j = 0; j < n3; ++j)
        {
          std::fill(v.begin(), v.end(), 0.0);
          v[j] = 1.0;
          interpolate_hanging(fe, mask, v.data(), false);
          for (int i = 0; i < n3; ++i)
            {
              double s = 0;
              for (int k = 0; k < n3; ++k)
                s += Kref[(size_t)i * n3 + k] * v[k];
              w[i] = s;
            }
          interpolate_hanging(fe, mask, w.data(), true);
          for (int i = 0; i < n3; ++i)
            Ke[(size_t)i * n3 + j] = w[i];
        }
      return cache.emplace(mask, std::move(Ke)).first->second;
    };
    const size_t          nc = L.tria->cells.size();
    std::vector<uint32_t> idx((size_t)nc * n3, INVALID_DOF);
    for (size_t ci = 0; ci < nc; ++ci)
      if (L.cell_is_local(ci))
        for (int c = 0; c < n; ++c)
          for (int b = 0; b < n; ++b)
            for (int a = 0; a < n; ++a)
              {
                const int l[3]                              = {a, b, c};
                idx[ci * n3 + (size_t)(c * n + b) * n + a] = L.cell_node_index(ci, l);
              }
    CSR A;
    A.n_rows = A.n_cols = L.n_dofs;
    // pattern: per row the columns of all touching cells, then sort + unique
    std::vector<uint32_t> cnt(L.n_dofs + 1, 0);
    for (size_t ci = 0; ci < nc; ++ci)
      for (int i = 0; i < n3; ++i)
        if (idx[ci * n3 + i] != INVALID_DOF)
          for (int j = 0; j < n3; ++j)
            if (idx[ci * n3 + j] != INVALID_DOF)
              ++cnt[idx[ci * n3 + i] + 1];
    for (uint32_t i = L.first_constrained(); i < L.n_dofs; ++i)
      ++cnt[i + 1];
    std::vector<size_t> start(L.n_dofs + 1, 0);
    for (uint32_t i = 0; i < L.n_dofs; ++i)
      start[i + 1] = start[i] + cnt[i + 1];
    std::vector<uint32_t> cols(start[L.n_dofs]);
    {
      std::vector<size_t> fill(start.begin(), start.end() - 1);
      for (size_t ci = 0; ci < nc; ++ci)
        for (int i = 0; i < n3; ++i)
          if (idx[ci * n3 + i] != INVALID_DOF)
            for (int j = 0; j < n3; ++j)
              if (idx[ci * n3 + j] != INVALID_DOF)
                cols[fill[idx[ci * n3 + i]]++] = idx[ci * n3 + j];
      for (uint32_t i = L.first_constrained(); i < L.n_dofs; ++i)
        cols[fill[i]++] = i;
    }
    A.ptr.assign(L.n_dofs + 1, 0);
    for (uint32_t i = 0; i < L.n_dofs; ++i)
      {
        auto b = cols.begin() + start[i], e = cols.begin() + start[i + 1];
        std::sort(b, e);
        e            = std::unique(b, e);
        A.ptr[i + 1] = A.ptr[i] + (uint32_t)(e - b);
      }
    A.col.resize(A.ptr[L.n_dofs]);
    for (uint32_t i = 0; i < L.n_dofs; ++i)
      std::copy(cols.begin() + start[i], cols.begin() + start[i] + (A.ptr[i + 1] - A.ptr[i]), A.col.begin() + A.ptr[i]);
    cols.clear();
    cols.shrink_to_fit();
    A.val.assign(A.col.size(), 0.0);
    auto at = [&](uint32_t i, uint32_t j) -> double & {
      auto b = A.col.begin() + A.ptr[i], e = A.col.begin() + A.ptr[i + 1];
      return A.val[std::lower_bound(b, e, j) - A.col.begin()];
    };
    for (size_t ci = 0; ci < nc; ++ci)
      {
        if (!L.cell_is_local(ci))
          continue;
        const std::vector<double> &Ke = element(L.tria->masks[ci]);
        const double               h  = 2.0 / (double)(1u << L.tria->cells[ci].level);
        for (int i = 0; i < n3; ++i)
          if (idx[ci * n3 + i] != INVALID_DOF)
            for (int j = 0; j < n3; ++j)
              if (idx[ci * n3 + j] != INVALID_DOF)
                at(idx[ci * n3 + i], idx[ci * n3 + j]) += h * Ke[(size_t)i * n3 + j];
      }
    for (uint32_t i = L.first_constrained(); i < L.n_dofs; ++i)
      at(i, i) = 1.0;
    return A;
  }

  // ---------------------------------------------------------------- sparse kernels of the setup
  inline CSR
  csr_transpose(const CSR &A)
  {
    CSR T;
    T.n_rows = A.n_cols;
    T.n_cols = A.n_rows;
    T.ptr.assign(T.n_rows + 1, 0);
    for (uint32_t c : A.col)
      ++T.ptr[c + 1];
    for (uint32_t i = 0; i < T.n_rows; ++i)
      T.ptr[i + 1] += T.ptr[i];
    T.col.resize(A.nnz());
    T.val.resize(A.nnz());
    std::vector<uint32_t> fill(T.ptr.begin(), T.ptr.end() - 1);
    for (uint32_t i = 0; i < A.n_rows; ++i)
      for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
        {
          const uint32_t q = fill[A.col[k]]++;
          T.col[q]         = i;
          T.val[q]         = A.val[k];
        }
    return T;
  }
  inline CSR
  csr_multiply(const CSR &A, const CSR &B) // rows sorted by column
  {
    CSR C;
    C.n_rows = A.n_rows;
    C.n_cols = B.n_cols;
    C.ptr.assign(A.n_rows + 1, 0);
    std::vector<int32_t>  marker(B.n_cols, -1);
    std::vector<double>   acc(B.n_cols, 0.0);
    std::vector<uint32_t> row;
    for (uint32_t i = 0; i < A.n_rows; ++i)
      {
        row.clear();
        for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
          {
            const uint32_t j = A.col[k];
            const double   a = A.val[k];
            for (uint32_t q = B.ptr[j]; q < B.ptr[j + 1]; ++q)
              {
                const uint32_t c = B.col[q];
                if (marker[c] != (int32_t)i)
                  {
                    marker[c] = (int32_t)i;
                    acc[c]    = 0.0;
                    row.push_back(c);
                  }
                acc[c] += a * B.val[q];
              }
          }
        std::sort(row.begin(), row.end());
        for (uint32_t c : row)
          {
            C.col.push_back(c);
            C.val.push_back(acc[c]);
          }
        C.ptr[i + 1] = (uint32_t)C.col.size();
      }
    return C;
  }
  inline void
  csr_vmult(const CSR &A, const std::vector<double> &x, std::vector<double> &y)
  {
    y.assign(A.n_rows, 0.0);
    for (uint32_t i = 0; i < A.n_rows; ++i)
      {
        double s = 0;
        for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
          s += A.val[k] * x[A.col[k]];
        y[i] = s;
      }
  }

  // ---------------------------------------------------------------- smoothed aggregation
  struct AmgLevelHost
  {
    CSR                 A, P, R; // P: this level <- next coarser, R = P^T (empty on the coarsest level)
    std::vector<double> dinv;    // 1 / a_ii
    double              lambda_max = 1.0; // of D^-1 A (power iteration x 1.1)
    uint32_t            n_aggregates = 0;
  };
  struct AmgHierarchyHost
  {
    std::vector<AmgLevelHost> levels;     // finest first
    std::vector<double>       coarse_inv; // dense inverse of the coarsest A, row-major
  };
  struct AmgParameters
  {
    double   strength_threshold = 1e-4;
    uint32_t max_coarse         = 1000;
    unsigned max_levels         = 12;
    unsigned power_iterations   = 20;
  };

  inline double
  estimate_lambda_max(const CSR &A, const std::vector<double> &dinv, unsigned its)
  {
    const uint32_t      n = A.n_rows;
    std::vector<double> v(n), w;
    for (uint32_t i = 0; i < n; ++i)
      v[i] = 1.0 + 0.25 * (double)((i * 2654435761u >> 16) % 7); // deterministic, no constant vector (the near null space)
    double lambda = 1.0;
    for (unsigned it = 0; it < its; ++it)
      {
        csr_vmult(A, v, w);
        double nw = 0, nv = 0;
        for (uint32_t i = 0; i < n; ++i)
          {
            w[i] *= dinv[i];
            nw += w[i] * w[i];
            nv += v[i] * v[i];
          }
        lambda = std::sqrt(nw / nv);
        const double s = 1.0 / std::sqrt(nw);
        for (uint32_t i = 0; i < n; ++i)
          v[i] = w[i] * s;
      }
    return 1.1 * lambda;
  }

  // aggregate index per row (-1: decoupled row), number of aggregates
  inline uint32_t
  aggregate(const CSR &A, double theta, std::vector<int32_t> &agg)
  {
    const uint32_t      n = A.n_rows;
    std::vector<double> diag(n, 1.0);
    for (uint32_t i = 0; i < n; ++i)
      for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
        if (A.col[k] == i)
          diag[i] = std::fabs(A.val[k]);
    auto strong = [&](uint32_t i, uint32_t k) {
      const uint32_t j = A.col[k];
      return j != i && std::fabs(A.val[k]) > 0.0 && std::fabs(A.val[k]) >= theta * std::sqrt(diag[i] * diag[j]);
    };
    agg.assign(n, -1);
    std::vector<uint8_t> coupled(n, 0);
    for (uint32_t i = 0; i < n; ++i)
      for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1] && !coupled[i]; ++k)
        coupled[i] = strong(i, k);
    uint32_t na = 0;
    // pass 1: roots with a completely free strong neighbourhood
    for (uint32_t i = 0; i < n; ++i)
      {
        if (!coupled[i] || agg[i] >= 0)
          continue;
        bool free_nb = true;
        for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1] && free_nb; ++k)
          if (strong(i, k) && agg[A.col[k]] >= 0)
            free_nb = false;
        if (!free_nb)
          continue;
        agg[i] = (int32_t)na;
        for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
          if (strong(i, k))
            agg[A.col[k]] = (int32_t)na;
        ++na;
      }
    // pass 2: leftovers join the aggregate (of pass 1) they are most strongly connected to
    {
      const std::vector<int32_t> agg1 = agg;
      for (uint32_t i = 0; i < n; ++i)
        {
          if (!coupled[i] || agg1[i] >= 0)
            continue;
          double  best = 0;
          int32_t to   = -1;
          for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
            if (strong(i, k) && agg1[A.col[k]] >= 0 && std::fabs(A.val[k]) > best)
              {
                best = std::fabs(A.val[k]);
                to   = agg1[A.col[k]];
              }
          agg[i] = to;
        }
    }
    // pass 3: what is still free forms aggregates of its own
    for (uint32_t i = 0; i < n; ++i)
      {
        if (!coupled[i] || agg[i] >= 0)
          continue;
        agg[i] = (int32_t)na;
        for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
          if (strong(i, k) && agg[A.col[k]] < 0)
            agg[A.col[k]] = (int32_t)na;
        ++na;
      }
    return na;
  }

  inline void
  dense_inverse(const CSR &A, std::vector<double> &inv)
  {
    const size_t        n = A.n_rows;
    std::vector<double> M(n * n, 0.0);
    for (uint32_t i = 0; i < n; ++i)
      for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
        M[i * n + A.col[k]] = A.val[k];
    inv.assign(n * n, 0.0);
    for (size_t i = 0; i < n; ++i)
      inv[i * n + i] = 1.0;
    for (size_t c = 0; c < n; ++c)
      {
        size_t piv = c;
        for (size_t r = c + 1; r < n; ++r)
          if (std::fabs(M[r * n + c]) > std::fabs(M[piv * n + c]))
            piv = r;
        if (std::fabs(M[piv * n + c]) < 1e-300)
          throw std::runtime_error("AMG: the coarsest matrix is singular");
        if (piv != c)
          for (size_t k = 0; k < n; ++k)
            {
              std::swap(M[c * n + k], M[piv * n + k]);
              std::swap(inv[c * n + k], inv[piv * n + k]);
            }
        const double d = 1.0 / M[c * n + c];
        for (size_t k = 0; k < n; ++k)
          {
            M[c * n + k] *= d;
            inv[c * n + k] *= d;
          }
        for (size_t r = 0; r < n; ++r)
          if (r != c)
            {
              const double f = M[r * n + c];
              if (f != 0.0)
                for (size_t k = 0; k < n; ++k)
                  {
                    M[r * n + k] -= f * M[c * n + k];
                    inv[r * n + k] -= f * inv[c * n + k];
                  }
            }
      }
  }

  inline AmgHierarchyHost
  build_smoothed_aggregation(CSR A0, const AmgParameters &prm = AmgParameters())
  {
    AmgHierarchyHost H;
    H.levels.emplace_back();
    H.levels.back().A = std::move(A0);
    for (;;)
      {
        AmgLevelHost  &lv = H.levels.back();
        const CSR     &A  = lv.A;
        const uint32_t n  = A.n_rows;
        lv.dinv.assign(n, 1.0);
        for (uint32_t i = 0; i < n; ++i)
          for (uint32_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
            if (A.col[k] == i && A.val[k] != 0.0)
              lv.dinv[i] = 1.0 / A.val[k];
        lv.lambda_max = estimate_lambda_max(A, lv.dinv, prm.power_iterations);
        if (n <= prm.max_coarse || H.levels.size() >= prm.max_levels)
          break;
        std::vector<int32_t> agg;
        const uint32_t       na = aggregate(A, prm.strength_threshold, agg);
        if (na == 0 || na >= n)
          break; // no coarsening possible
        lv.n_aggregates = na;
        // tentative prolongator: piecewise constant, columns of unit length
        std::vector<uint32_t> size(na, 0);
        for (uint32_t i = 0; i < n; ++i)
          if (agg[i] >= 0)
            ++size[agg[i]];
        CSR Pt;
        Pt.n_rows = n;
        Pt.n_cols = na;
        Pt.ptr.assign(n + 1, 0);
        for (uint32_t i = 0; i < n; ++i)
          {
            if (agg[i] >= 0)
              {
                Pt.col.push_back((uint32_t)agg[i]);
                Pt.val.push_back(1.0 / std::sqrt((double)size[agg[i]]));
              }
            Pt.ptr[i + 1] = (uint32_t)Pt.col.size();
          }
        // P = P_t - omega D^-1 A P_t
        const double omega = 4.0 / (3.0 * lv.lambda_max);
        CSR          AP    = csr_multiply(A, Pt);
        for (uint32_t i = 0; i < n; ++i)
          for (uint32_t k = AP.ptr[i]; k < AP.ptr[i + 1]; ++k)
            AP.val[k] *= -omega * lv.dinv[i];
        // add P_t (both have sorted rows)
        CSR P;
        P.n_rows = n;
        P.n_cols = na;
        P.ptr.assign(n + 1, 0);
        for (uint32_t i = 0; i < n; ++i)
          {
            uint32_t a = AP.ptr[i], ae = AP.ptr[i + 1], b = Pt.ptr[i], be = Pt.ptr[i + 1];
            while (a < ae || b < be)
              {
                if (b >= be || (a < ae && AP.col[a] < Pt.col[b]))
                  {
                    P.col.push_back(AP.col[a]);
                    P.val.push_back(AP.val[a]);
                    ++a;
                  }
                else if (a >= ae || Pt.col[b] < AP.col[a])
                  {
                    P.col.push_back(Pt.col[b]);
                    P.val.push_back(Pt.val[b]);
                    ++b;
                  }
                else
                  {
                    P.col.push_back(AP.col[a]);
                    P.val.push_back(AP.val[a] + Pt.val[b]);
                    ++a;
                    ++b;
                  }
              }
            P.ptr[i + 1] = (uint32_t)P.col.size();
          }
        lv.P   = std::move(P);
        lv.R   = csr_transpose(lv.P);
        CSR Ac = csr_multiply(lv.R, csr_multiply(A, lv.P));
        // symmetrise (rounding) -- the Galerkin product of a symmetric matrix
        H.levels.emplace_back();
        H.levels.back().A = std::move(Ac);
      }
    dense_inverse(H.levels.back().A, H.coarse_inv);
    return H;
  }
} // namespace mgamd
