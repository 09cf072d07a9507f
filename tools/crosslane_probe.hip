// Development probe for north_star's "wavefront shuffles for the 1D sweeps" (round-2 verdict, item 3): what does it cost to hand the
// results of one sweep direction to the next direction's product (a) through LDS, as the kernels do, or (b) across lanes?
//
// Setting of the 17^3 p = 4 lattice kernels (kernels.hpp, lattice_sweeps): in the y sweep thread (x, z) holds the 17 values of its
// y line for two arrays (c = M_y a, g = K_y a + M_y b) and the x sweep needs, for every point (x, y, z), the values of the
// <= 9 points x' of the one or two cells that contain x.  The thread index is l = 17 z + x, so those points sit in ADJACENT lanes.
//   (a) LDS:        17 + 17 ds_write_b64, barrier, 17 + 17 ds_read_b64 per thread, then the even-odd 5x5 products per cell
//                   (13 multiply-adds per product) from registers
//   (b) cross-lane: no LDS; every thread forms its 17 outputs from the values of the lanes l - 4 .. l + 4:
//                   2 arrays x 17 values x 8 neighbours, each a 64-bit __shfl (= 2 ds_bpermute_b32), and a dense 9-term stencil
//                   with lane-dependent weights (the node's position in its cell decides which of the 9 are non-zero): no
//                   even-odd saving, and lines that straddle a wavefront (17 does not divide 64) would still need LDS
//   (c) DPP:        the same stencil with v_mov_b32 row_shr/row_shl (16-lane rows) instead of ds_bpermute: an upper bound on what
//                   DPP could do -- it is NOT a correct sweep for 17-point lines (a row holds 16 lanes), it only prices the
//                   instruction mix
// Prints ns per thread-line hand-over (+ product) for the three, 2 workgroups of 256 threads per CU like the lattice kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/crosslane_probe.hip -o tools/bin/crosslane_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int N = 17, NW = 9;

__device__ __forceinline__ double
dpp_shift(double v, int k) // value of lane - k within a row of 16 (k in -4..4), both halves of the double
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  switch (k)
    {
#define C(K, CTRL)                                                         \
  case K:                                                                  \
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);        \
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);        \
    break;
      C(1, 0x111) C(2, 0x112) C(3, 0x113) C(4, 0x114)   // row_shr:1..4
      C(-1, 0x101) C(-2, 0x102) C(-3, 0x103) C(-4, 0x104) // row_shl:1..4
#undef C
      default:
        break;
    }
  return __hiloint2double(hi, lo);
}

template <int VARIANT>
__global__ void
__launch_bounds__(256, 2) probe(int reps, const double *__restrict__ w_in, double *__restrict__ out)
{
  extern __shared__ __align__(16) unsigned char smem[];
  double   *A = reinterpret_cast<double *>(smem), *Bm = A + N * N * N;
  const int tid = threadIdx.x, lane = tid & 63;
  const int x = tid % N, z = tid / N; // (lines 256..288 of the real kernel are left out: same work per line)
  double    c[N], g[N];
  for (int i = 0; i < N; ++i)
    {
      c[i] = 1.0 + 0.001 * (tid + i);
      g[i] = 0.5 - 0.002 * (tid - i);
    }
  double w[2 * NW]; // lane-dependent 9-point weights of the two matrices (K row, M row of node x)
  for (int k = 0; k < 2 * NW; ++k)
    w[k] = w_in[(x % 4) * 2 * NW + k];
  double acc = 0;
  for (int r = 0; r < reps; ++r)
    {
      if constexpr (VARIANT == 0)
        {
          // y sweep's stores (line (x, z): stride N), barrier, x sweep's loads (line (y = x, z): stride 1) + even-odd-sized products
          const int wb = z * N * N + x, rb = (z * N + x) * N;
          for (int i = 0; i < N; ++i)
            {
              A[wb + i * N]  = c[i];
              Bm[wb + i * N] = g[i];
            }
          __syncthreads();
          double a[N], b[N];
          for (int i = 0; i < N; ++i)
            {
              a[i] = A[rb + i];
              b[i] = Bm[rb + i];
            }
          __syncthreads();
          // 4 cells x (5x5 K a + 5x5 M b) with the even-odd count of multiply-adds (13 per product)
          for (int cidx = 0; cidx < 4; ++cidx)
            for (int m = 0; m < 13; ++m)
              {
                c[cidx * 4 + (m % 5)] = fma(w[m % NW], a[cidx * 4 + ((m + 1) % 5)], c[cidx * 4 + (m % 5)]);
                g[cidx * 4 + (m % 5)] = fma(w[NW + m % NW], b[cidx * 4 + ((m + 2) % 5)], g[cidx * 4 + (m % 5)]);
              }
        }
      else
        {
          // every output from the 9 lanes around this one, both arrays
          for (int i = 0; i < N; ++i)
            {
              double s = 0;
#pragma unroll
              for (int k = -4; k <= 4; ++k)
                {
                  double vc, vg;
                  if constexpr (VARIANT == 1)
                    {
                      vc = k == 0 ? c[i] : __shfl(c[i], lane + k, 64);
                      vg = k == 0 ? g[i] : __shfl(g[i], lane + k, 64);
                    }
                  else
                    {
                      vc = dpp_shift(c[i], k);
                      vg = dpp_shift(g[i], k);
                    }
                  s = fma(w[k + 4], vc, s);
                  s = fma(w[NW + k + 4], vg, s);
                }
              c[i] = 0.999 * s;
              g[i] = 0.5 * g[i] + 0.001 * s;
            }
        }
    }
  for (int i = 0; i < N; ++i)
    acc += c[i] + g[i];
  out[blockIdx.x * 256 + tid] = acc;
}

int
main()
{
  const int           nwg = 512, reps = 400;
  std::vector<double> w(4 * 2 * NW);
  for (size_t i = 0; i < w.size(); ++i)
    w[i] = 0.01 * (double)(i % 7) - 0.02;
  double *dw, *dout;
  hipMalloc(&dw, w.size() * 8);
  hipMalloc(&dout, (size_t)nwg * 256 * 8);
  hipMemcpy(dw, w.data(), w.size() * 8, hipMemcpyHostToDevice);
  const size_t lds = 2 * (size_t)N * N * N * 8;
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char *names[3] = {"(a) LDS hand-over + even-odd products", "(b) __shfl (ds_bpermute) 9-point stencil", "(c) DPP row shifts 9-point stencil (instruction mix only)"};
  printf("x-direction hand-over + product of one 17^3 p = 4 lattice sweep (256 lines per workgroup, 2 workgroups per CU, %d workgroups x %d)\n", nwg, reps);
  for (int v = 0; v < 3; ++v)
    {
      float best = 1e30f;
      for (int trial = 0; trial < 3; ++trial)
        {
          hipEventRecord(e0);
          if (v == 0)
            hipLaunchKernelGGL(probe<0>, nwg, 256, lds, 0, reps, dw, dout);
          else if (v == 1)
            hipLaunchKernelGGL(probe<1>, nwg, 256, lds, 0, reps, dw, dout); // (same LDS reservation: same occupancy)
          else
            hipLaunchKernelGGL(probe<2>, nwg, 256, lds, 0, reps, dw, dout);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
      if (hipGetLastError() != hipSuccess)
        {
          printf("launch failed\n");
          return 1;
        }
      printf("  %-60s %8.3f ms = %7.1f ns per sweep of a workgroup slot\n", names[v], best, best * 1e6 / reps);
    }
  return 0;
}
