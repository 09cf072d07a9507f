#pragma once
#include "kernels_common.hpp"

// ------------------------------------------------------------------------------------------------
// K7  CSR kernels of the algebraic coarse solver (amg.hpp; runtime.hip AmgDevice): one row per group of LANES adjacent lanes
// (the level matrices have 27-70 entries per row), fused with the vector update they feed:
//   SPMV_PLAIN   y = A x                      SPMV_ADD      y += A x  (prolongation)
//   SPMV_RESID   y = b - A x                  SPMV_CHEB     y = x + f1 (x - xold) + f2 dinv (b - A x)   (xold may be null)
// ------------------------------------------------------------------------------------------------
namespace mgamd
{
  enum SpmvMode
  {
    SPMV_PLAIN = 0,
    SPMV_ADD   = 1,
    SPMV_RESID = 2,
    SPMV_CHEB  = 3
  };
  template <typename T, int MODE, int LANES>
  __global__ void
  __launch_bounds__(256) csr_spmv_kernel(uint32_t n_rows, const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ col,
                                         const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y, const T *__restrict__ b,
                                         const T *__restrict__ xold, const T *__restrict__ dinv, T f1, T f2)
  {
    const uint32_t rows_per_block = 256 / LANES;
    const uint32_t sub = threadIdx.x % LANES, lrow = threadIdx.x / LANES;
    for (uint32_t row0 = blockIdx.x * rows_per_block; row0 < n_rows; row0 += gridDim.x * rows_per_block)
      {
        const uint32_t row = row0 + lrow;
        T              s   = T(0);
        if (row < n_rows)
          {
            const uint32_t e = ptr[row + 1];
            for (uint32_t k = ptr[row] + sub; k < e; k += LANES)
              s += val[k] * x[col[k]];
          }
#pragma unroll
        for (int off = LANES / 2; off > 0; off >>= 1)
          s += __shfl_down(s, off, LANES);
        if (row < n_rows && sub == 0)
          {
            if (MODE == SPMV_PLAIN)
              y[row] = s;
            else if (MODE == SPMV_ADD)
              y[row] += s;
            else if (MODE == SPMV_RESID)
              y[row] = b[row] - s;
            else
              {
                const T xv = x[row], xo = xold ? xold[row] : T(0);
                y[row]     = xv + f1 * (xv - xo) + f2 * dinv[row] * (b[row] - s);
              }
          }
      }
  }
} // namespace mgamd

