// K5 / K6: halo pack / combine, indexed copies of local smoothing, vector kernels, the device-resident PCG updates and the dense
// coarse matvec.  See kernels.hpp for the overview.
#pragma once
#include "kernels_common.hpp"

namespace mgamd
{
  // ------------------------------------------------------------------------------------------
  // Halo exchange of shared tail DoFs (sharded runs): pack the partial sums per peer, and after the exchange
  // combine own + received contributions in ascending rank order (bitwise identical on every sharing rank).
  // ------------------------------------------------------------------------------------------
  template <typename T>
  __global__ void
  __launch_bounds__(256) halo_pack_kernel(T *__restrict__ send, const T *__restrict__ tail, const uint32_t *__restrict__ pack_idx, uint32_t n)
  {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      send[i] = tail[pack_idx[i]];
  }

  template <typename T>
  __global__ void
  __launch_bounds__(256) halo_combine_kernel(T *__restrict__ tail, const T *__restrict__ recv, const uint32_t *__restrict__ sh_tail,
                                             const uint32_t *__restrict__ sh_ptr, const int32_t *__restrict__ sh_src, uint32_t n_shared)
  {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_shared; i += stride)
      {
        const uint32_t t   = sh_tail[i];
        const T        own = tail[t];
        T              acc = T(0);
        for (uint32_t e = sh_ptr[i]; e < sh_ptr[i + 1]; ++e)
          {
            const int32_t src = sh_src[e];
            acc += src < 0 ? own : recv[src];
          }
        tail[t] = acc;
      }
  }

  // copies of shared DoFs take the owner's value (after prolongation: ranks that reference a coarse face only through
  // hanging-node resolution have no patch that writes their copy)
  template <typename T>
  __global__ void
  __launch_bounds__(256) halo_import_kernel(T *__restrict__ tail, const T *__restrict__ recv, const uint32_t *__restrict__ sh_tail,
                                            const int32_t *__restrict__ sh_owner_src, uint32_t n_shared)
  {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_shared; i += stride)
      if (sh_owner_src[i] >= 0)
        tail[sh_tail[i]] = recv[sh_owner_src[i]];
  }

  // ------------------------------------------------------------------------------------------
  // Local smoothing: copy_to_mg / copy_from_mg between the active-mesh vector and a level vector (index pairs of
  // LevelTables / ls_copy_indices), with the cast between the outer and the level number type
  // ------------------------------------------------------------------------------------------
  template <typename TD, typename TS>
  __global__ void
  __launch_bounds__(256) indexed_copy_kernel(TD *__restrict__ dst, const uint32_t *__restrict__ dst_idx, const TS *__restrict__ src,
                                             const uint32_t *__restrict__ src_idx, uint32_t n)
  {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      dst[dst_idx[i]] = (TD)src[src_idx[i]];
  }

  // ------------------------------------------------------------------------------------------
  // Vector kernels
  // ------------------------------------------------------------------------------------------
  template <typename T>
  __global__ void
  __launch_bounds__(256) vec_set_kernel(T *__restrict__ v, T value, size_t n)
  {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      v[i] = value;
  }

  template <typename TD, typename TS>
  __global__ void
  __launch_bounds__(256) vec_copy_kernel(TD *__restrict__ d, const TS *__restrict__ s, size_t n)
  {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      d[i] = (TD)s[i];
  }

  // y = s*y + a*x
  template <typename T>
  __global__ void
  __launch_bounds__(256) vec_sadd_kernel(T *__restrict__ y, T s, T a, const T *__restrict__ x, size_t n)
  {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      y[i] = s * y[i] + a * x[i];
  }

  // y = a * d .* b      (Chebyshev zero-start first iterate: x1 = (1/theta) D^-1 b)
  template <typename T>
  __global__ void
  __launch_bounds__(256) vec_scaled_product_kernel(T *__restrict__ y, T a, const T *__restrict__ d, const T *__restrict__ b, size_t n)
  {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      y[i] = a * d[i] * b[i];
  }

  __device__ __forceinline__ double
  wave_reduce_sum(double v)
  {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      v += __shfl_down(v, off, 64);
    return v;
  }

  // stage 1: per-block partial sums of x.y (double accumulation); stage 2 (grid 1): final sum
  template <typename T>
  __global__ void
  __launch_bounds__(256) vec_dot_kernel(const T *__restrict__ x, const T *__restrict__ y, size_t n, double *__restrict__ partial)
  {
    __shared__ double wsum[4];
    double            s      = 0.0;
    const size_t      stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      s += (double)x[i] * (double)y[i];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0)
      wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
      partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }

  template <typename = void> // (a template only so that the header can be included by several translation units)
  __global__ void
  __launch_bounds__(256) vec_dot_final_kernel(const double *__restrict__ partial, int n, double *__restrict__ result)
  {
    __shared__ double wsum[4];
    double            s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256)
      s += partial[i];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0)
      wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
      *result = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }

  // ------------------------------------------------------------------------------------------
  // Device-resident PCG (SolverCG, ref:multigrid_throughput.cc:1143-1144,1625-1635): every scalar of the iteration lives
  // in a small device array S; the vector updates read alpha/beta from it and are fused with the reductions they feed.
  // The host reads ONE number per iteration (the residual norm, for ReductionControl).
  //   S[0], S[1]: r.z of the current / next iteration (ping-pong)   S[2]: p.Ap   S[3]: r.r
  // ------------------------------------------------------------------------------------------
  template <typename T>
  __global__ void
  __launch_bounds__(256) cg_dot_kernel(const T *__restrict__ x, const T *__restrict__ y, size_t n, double *__restrict__ partial)
  {
    __shared__ double wsum[4];
    double            s      = 0.0;
    const size_t      stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      s += (double)x[i] * (double)y[i];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0)
      wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
      partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  // x += alpha p, r -= alpha Ap with alpha = S[rz] / S[2], fused with the partial sums of r.r over the first n_dot entries
  // (n_dot <= n: the owned prefix on a sharded level)
  template <typename T>
  __global__ void
  __launch_bounds__(256) cg_update_xr_kernel(T *__restrict__ x, T *__restrict__ r, const T *__restrict__ p, const T *__restrict__ Ap, size_t n,
                                             size_t n_dot, const double *__restrict__ S, int rz, double *__restrict__ partial)
  {
    __shared__ double wsum[4];
    const T           alpha  = (T)(S[rz] / S[2]);
    double            s      = 0.0;
    const size_t      stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      {
        x[i] += alpha * p[i];
        const T rn = r[i] - alpha * Ap[i];
        r[i]       = rn;
        if (i < n_dot)
          s += (double)rn * (double)rn;
      }
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0)
      wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
      partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  // p = z + beta p with beta = S[rz_new] / S[rz_old]
  template <typename T>
  __global__ void
  __launch_bounds__(256) cg_update_p_kernel(T *__restrict__ p, const T *__restrict__ z, size_t n, const double *__restrict__ S, int rz_new, int rz_old)
  {
    const T      beta   = (T)(S[rz_new] / S[rz_old]);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      p[i] = z[i] + beta * p[i];
  }

  // y = Minv x for the coarse-grid direct solve (Minv: n x n row-major, double)
  template <typename T>
  __global__ void
  __launch_bounds__(256) dense_matvec_kernel(const double *__restrict__ Minv, const T *__restrict__ x, T *__restrict__ y, int n)
  {
    __shared__ double wsum[4];
    for (int row = blockIdx.x; row < n; row += gridDim.x)
      {
        double s = 0.0;
        for (int j = threadIdx.x; j < n; j += 256)
          s += Minv[(size_t)row * n + j] * (double)x[j];
        s = wave_reduce_sum(s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0)
          wsum[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0)
          y[row] = (T)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
      }
  }
} // namespace mgamd
