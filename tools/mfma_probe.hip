// The experiment SURVEY.md section 7 / north_star ask for ("MFMA only on the small dense (p+1)^d per-cell contractions where
// it actually pays, evidenced by rocprof"): the three 1D sweeps of the p = 4 level operator on one 17^3 lattice held in LDS,
//   (a) VALU: the production code (kernels.hpp lattice_sweeps: one thread per lattice line, even-odd 5x5 products from
//       SGPR-resident matrices, segment tasks for the 33 left-over lines),
//   (b) MFMA: the same products batched per cell as (5x5) . (5 x 16 lines) on v_mfma_f64_16x16x4_f64 (rows and k padded to
//       16 and 8: 19.5 % of each instruction's 2048 FLOP are useful; no even-odd splitting is possible on a fixed tile),
// with the same launch shape as the production kernel (256 threads, two lattices = two workgroups per CU, 512 workgroups).
// Prints the time per lattice application of both and the maximal difference of their results.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idealii_multigrid_amd/csrc -Iinclude tools/mfma_probe.hip -o tools/bin/mfma_probe
#include "kernels.hpp"
#include "fe1d.hpp"

#include <cstdio>
#include <vector>

using namespace mgamd;
typedef double double4_t __attribute__((ext_vector_type(4)));

#define CHECK(e)                                                                            \
  do                                                                                        \
    {                                                                                       \
      hipError_t s_ = (e);                                                                  \
      if (s_ != hipSuccess)                                                                 \
        {                                                                                   \
          std::fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(s_), __LINE__); \
          return 1;                                                                         \
        }                                                                                   \
    }                                                                                       \
  while (0)

constexpr int P = 4, B = 4, N = 17, N3 = N * N * N, LINES = N * N;

__device__ __forceinline__ void
init_lattice(double *bufA, int tid, uint32_t seed)
{
  for (int i = tid; i < N3; i += 256)
    {
      uint32_t h = (uint32_t)i * 2654435761u + seed * 40503u;
      h ^= h >> 15;
      h *= 2246822519u;
      h ^= h >> 13;
      bufA[i] = (double)(h & 0xFFFF) / 65536.0 - 0.5;
    }
  __syncthreads();
}

__global__ void
__launch_bounds__(256, 2) valu_kernel(const Mats<P> m, int reps, double *out, double *lattice0)
{
  extern __shared__ __align__(16) unsigned char smem[];
  double *bufA = reinterpret_cast<double *>(smem), *bufB = bufA + N3;
  const int    tid = threadIdx.x;
  const double h   = 1.0; // the repeated operator would blow up; renormalise through h = 1 and a scale per repetition
  init_lattice(bufA, tid, blockIdx.x);
  for (int r = 0; r < reps; ++r)
    {
      lattice_sweeps<double, P, B, 256>(bufA, bufB, m, tid, 1, &h);
      for (int i = tid; i < N3; i += 256)
        bufA[i] *= 0.25;
      __syncthreads();
    }
  double s = 0;
  for (int i = tid; i < N3; i += 256)
    s += bufA[i];
  atomicAdd(&out[blockIdx.x], s);
  if (blockIdx.x == 0 && lattice0)
    for (int i = tid; i < N3; i += 256)
      lattice0[i] = bufA[i];
}

// one sweep direction with MFMA.  KIND 0: (A, B) <- (M a, K a);  1: (A, B) <- (M a, K a + M b);  2: A <- scale (K a + M b)
template <int KIND>
__device__ __forceinline__ void
mfma_sweep(double *__restrict__ bufA, double *__restrict__ bufB, const double aM[2], const double aK[2], int tid, int dir, double scale)
{
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4; // B operand: column j, k = g;   D: column j, rows g + 4 reg
  constexpr int NGROUPS = (LINES + 15) / 16;
  for (int q = wave; q < NGROUPS; q += 4)
    {
      const int  L     = 16 * q + j;
      const bool valid = L < LINES;
      const int  u = (valid ? L : 0) % N, v = (valid ? L : 0) / N;
      const int  base   = dir == 2 ? v * N + u : (dir == 1 ? v * N * N + u : (v * N + u) * N);
      const int  stride = dir == 2 ? N * N : (dir == 1 ? N : 1);
      // B operands of the 4 cells x 2 k-steps: node 4 c + 4 s + g (s = 1: only g = 0, the cell's last node)
      double xa[4][2], xb[4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        {
          xa[c][0] = valid ? bufA[base + (4 * c + g) * stride] : 0.0;
          xa[c][1] = (valid && g == 0) ? bufA[base + (4 * c + 4) * stride] : 0.0;
          if (KIND != 0)
            {
              xb[c][0] = valid ? bufB[base + (4 * c + g) * stride] : 0.0;
              xb[c][1] = (valid && g == 0) ? bufB[base + (4 * c + 4) * stride] : 0.0;
            }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every lane of the wave has its inputs before the in-place stores
      double4_t d1[4], d2[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        {
          double4_t z = {0, 0, 0, 0};
          if (KIND != 2)
            { // d1 = M a
              d1[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[0], xa[c][0], z, 0, 0, 0);
              d1[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[1], xa[c][1], d1[c], 0, 0, 0);
            }
          d2[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aK[0], xa[c][0], z, 0, 0, 0);
          d2[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aK[1], xa[c][1], d2[c], 0, 0, 0);
          if (KIND != 0)
            { // d2 += M b
              d2[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[0], xb[c][0], d2[c], 0, 0, 0);
              d2[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[1], xb[c][1], d2[c], 0, 0, 0);
            }
        }
      // rows 0..3 of a cell sit in register 0 of the lane groups g = 0..3, row 4 in register 1 of group 0; the node shared by
      // two cells gets row 4 of the left and row 0 of the right cell, both in the lanes of group 0
      if (valid)
        {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            {
              double v1 = KIND != 2 ? d1[c][0] : 0.0, v2 = d2[c][0];
              if (g == 0 && c > 0)
                {
                  if (KIND != 2)
                    v1 += d1[c - 1][1];
                  v2 += d2[c - 1][1];
                }
              if (KIND == 2)
                bufA[base + (4 * c + g) * stride] = scale * v2;
              else
                {
                  bufA[base + (4 * c + g) * stride] = v1;
                  bufB[base + (4 * c + g) * stride] = v2;
                }
            }
          if (g == 0)
            {
              if (KIND == 2)
                bufA[base + 16 * stride] = scale * d2[3][1];
              else
                {
                  bufA[base + 16 * stride] = d1[3][1];
                  bufB[base + 16 * stride] = d2[3][1];
                }
            }
        }
    }
  __syncthreads();
}

__global__ void
__launch_bounds__(256, 2) mfma_kernel(const Mats<P> m, int reps, double *out, double *lattice0)
{
  extern __shared__ __align__(16) unsigned char smem[];
  double *bufA = reinterpret_cast<double *>(smem), *bufB = bufA + N3;
  const int tid = threadIdx.x, lane = tid & 63;
  // A operand of k-step s: lane holds A[i = lane & 15][k = lane >> 4] = Mc[i][4 s + k] (zero padding outside 5 x 5)
  const int i = lane & 15, k = lane >> 4;
  double    aM[2], aK[2];
  for (int s = 0; s < 2; ++s)
    {
      const bool in = i <= P && 4 * s + k <= P;
      aM[s]         = in ? m.M[i * (P + 1) + 4 * s + k] : 0.0;
      aK[s]         = in ? m.K[i * (P + 1) + 4 * s + k] : 0.0;
    }
  init_lattice(bufA, tid, blockIdx.x);
  for (int r = 0; r < reps; ++r)
    {
      mfma_sweep<0>(bufA, bufB, aM, aK, tid, 2, 1.0); // z
      mfma_sweep<1>(bufA, bufB, aM, aK, tid, 1, 1.0); // y
      mfma_sweep<2>(bufA, bufB, aM, aK, tid, 0, 1.0); // x, h = 1
      for (int t = tid; t < N3; t += 256)
        bufA[t] *= 0.25;
      __syncthreads();
    }
  double s = 0;
  for (int t = tid; t < N3; t += 256)
    s += bufA[t];
  atomicAdd(&out[blockIdx.x], s);
  if (blockIdx.x == 0 && lattice0)
    for (int t = tid; t < N3; t += 256)
      lattice0[t] = bufA[t];
}

int
main(int argc, char **argv)
{
  const int reps = argc > 1 ? atoi(argv[1]) : 200, grid = argc > 2 ? atoi(argv[2]) : 512;
  FE1D      fe(P);
  Mats<P>   m;
  const int n = P + 1;
  for (int i = 0; i < n * n; ++i)
    {
      m.M[i]  = fe.M[i];
      m.K[i]  = fe.K[i];
      m.I0[i] = fe.I[0][i];
      m.I1[i] = fe.I[1][i];
    }
  constexpr int NH = Mats<P>::NH, NO = Mats<P>::NO;
  auto          eo = [&](const double *A, double *Ae, double *Ao) {
    for (int i = 0; i < NH; ++i)
      for (int j = 0; j < NH; ++j)
        Ae[i * NH + j] = (j < NO) ? 0.5 * (A[i * n + j] + A[i * n + P - j]) : A[i * n + j];
    for (int i = 0; i < NO; ++i)
      for (int j = 0; j < NO; ++j)
        Ao[i * NO + j] = 0.5 * (A[i * n + j] - A[i * n + P - j]);
  };
  eo(m.M, m.Me, m.Mo);
  eo(m.K, m.Ke, m.Ko);
  double *out, *lat;
  CHECK(hipMalloc((void **)&out, grid * sizeof(double)));
  CHECK(hipMalloc((void **)&lat, N3 * sizeof(double)));
  const size_t lds = 2 * (size_t)N3 * sizeof(double);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(valu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  std::vector<double> ref(N3), got(N3);
  hipEvent_t          e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double ms[2] = {0, 0};
  for (int variant = 0; variant < 2; ++variant)
    {
      // correctness: one application of workgroup 0
      CHECK(hipMemset(out, 0, grid * sizeof(double)));
      if (variant == 0)
        hipLaunchKernelGGL(valu_kernel, 1, 256, lds, 0, m, 1, out, lat);
      else
        hipLaunchKernelGGL(mfma_kernel, 1, 256, lds, 0, m, 1, out, lat);
      CHECK(hipMemcpy(variant == 0 ? ref.data() : got.data(), lat, N3 * sizeof(double), hipMemcpyDeviceToHost));
      for (int round = 0; round < 3; ++round) // timed (the last of three launches counts)
        {
          CHECK(hipEventRecord(e0, 0));
          if (variant == 0)
            hipLaunchKernelGGL(valu_kernel, grid, 256, lds, 0, m, reps, out, nullptr);
          else
            hipLaunchKernelGGL(mfma_kernel, grid, 256, lds, 0, m, reps, out, nullptr);
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          float t;
          CHECK(hipEventElapsedTime(&t, e0, e1));
          ms[variant] = t;
        }
    }
  double err = 0, mx = 0;
  for (int i = 0; i < N3; ++i)
    {
      err = std::max(err, std::fabs(ref[i] - got[i]));
      mx  = std::max(mx, std::fabs(ref[i]));
    }
  // useful FLOP of one application in the even-odd VALU form: 661 f64 operations per line triple x 289 lines
  const double apps = (double)grid * reps;
  std::printf("17^3 lattice, p = 4, %d workgroups x %d applications each (2 workgroups per CU)\n", grid, reps);
  std::printf("  VALU even-odd sweeps : %8.3f ms  = %7.1f ns per lattice application per workgroup slot\n", ms[0], ms[0] * 1e6 / reps);
  std::printf("  MFMA f64 16x16x4     : %8.3f ms  = %7.1f ns per lattice application per workgroup slot  (%.2fx the VALU time)\n", ms[1],
              ms[1] * 1e6 / reps, ms[1] / ms[0]);
  std::printf("  throughput: VALU %.1f M lattices/s, MFMA %.1f M lattices/s;  max |difference| of the two results %.3e (max |value| %.3e)\n",
              apps / ms[0] * 1e-3, apps / ms[1] * 1e-3, err, mx);
  return err <= 1e-12 * mx ? 0 : 2;
}
