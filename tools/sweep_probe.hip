// Development probe: the three sweeps of one 17^3 p = 4 lattice resident in LDS (two workgroups per CU, 512 workgroups x
// reps applications), as (0) whole lines in registers with 256 threads (lattice_sweeps), (1) the same lines streamed cell
// by cell, (2) streamed with the next cell's inputs prefetched, (3) 512 threads: half-line + quarter-line tasks
// (lattice_sweeps_wide).  Prints the time per lattice application and the difference of every variant to (0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idealii_multigrid_amd/csrc -Iinclude tools/sweep_probe.hip -o tools/bin/sweep_probe
#include "kernels.hpp"
#include "fe1d.hpp"

#include <cstdio>
#include <vector>

using namespace mgamd;

constexpr int P = 4, B = 4, N = 17, N3 = N * N * N;

template <int BLOCK>
__device__ __forceinline__ void
init_lattice(double *bufA, int tid, uint32_t seed)
{
  for (int i = tid; i < N3; i += BLOCK)
    {
      uint32_t h = (uint32_t)i * 2654435761u + seed * 40503u;
      h ^= h >> 15;
      h *= 2246822519u;
      h ^= h >> 13;
      bufA[i] = (double)(h & 0xFFFF) / 65536.0 - 0.5;
    }
  __syncthreads();
}

template <int VARIANT>
__global__ void
__launch_bounds__((VARIANT == 3 ? 512 : 256), (VARIANT == 3 ? 4 : 2)) sweep_kernel(const Mats<P> m, int reps, double *out, double *lattice0)
{
  constexpr int BLOCK = VARIANT == 3 ? 512 : 256;
  extern __shared__ __align__(16) unsigned char smem[];
  double *bufA = reinterpret_cast<double *>(smem), *bufB = bufA + N3;
  const int    tid = threadIdx.x;
  const double h   = 1.0;
  init_lattice<BLOCK>(bufA, tid, blockIdx.x);
  for (int r = 0; r < reps; ++r)
    {
      if constexpr (VARIANT == 0)
        lattice_sweeps<double, P, B, 256>(bufA, bufB, m, tid, 1, &h);
      else if constexpr (VARIANT == 1)
        lattice_sweeps<double, P, B, 256, NoHook, true, false>(bufA, bufB, m, tid, 1, &h);
      else if constexpr (VARIANT == 2)
        lattice_sweeps<double, P, B, 256, NoHook, true, true>(bufA, bufB, m, tid, 1, &h);
      else
        lattice_sweeps_wide<double, P>(bufA, bufB, m, tid, h);
      for (int i = tid; i < N3; i += BLOCK)
        bufA[i] *= 0.25;
      __syncthreads();
    }
  double s = 0;
  for (int i = tid; i < N3; i += BLOCK)
    s += bufA[i];
  atomicAdd(&out[blockIdx.x], s);
  if (blockIdx.x == 0 && lattice0)
    for (int i = tid; i < N3; i += BLOCK)
      lattice0[i] = bufA[i];
}

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

template <int V>
int
run(const Mats<P> &m, int reps, int grid, double *out, double *lat, std::vector<double> &res, float &ms)
{
  const size_t lds  = 2 * (size_t)N3 * sizeof(double);
  auto         kern = sweep_kernel<V>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  constexpr int BLOCK = V == 3 ? 512 : 256;
  CHECK(hipMemset(out, 0, grid * sizeof(double)));
  hipLaunchKernelGGL(kern, 1, BLOCK, lds, 0, m, 1, out, lat);
  res.resize(N3);
  CHECK(hipMemcpy(res.data(), lat, N3 * sizeof(double), hipMemcpyDeviceToHost));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int round = 0; round < 3; ++round)
    {
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(kern, grid, BLOCK, lds, 0, m, reps, out, nullptr);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
  return 0;
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
  std::vector<double> r[4];
  float               ms[4];
  if (run<0>(m, reps, grid, out, lat, r[0], ms[0]) || run<1>(m, reps, grid, out, lat, r[1], ms[1]) ||
      run<2>(m, reps, grid, out, lat, r[2], ms[2]) || run<3>(m, reps, grid, out, lat, r[3], ms[3]))
    return 1;
  const char *name[4] = {"whole lines, 256 threads", "streamed, 256 threads", "streamed + prefetch, 256", "half/quarter lines, 512"};
  int         rc      = 0;
  std::printf("17^3 lattice, p = 4, %d workgroups x %d applications each\n", grid, reps);
  for (int v = 0; v < 4; ++v)
    {
      double err = 0, mx = 0;
      for (int i = 0; i < N3; ++i)
        {
          err = std::max(err, std::fabs(r[0][i] - r[v][i]));
          mx  = std::max(mx, std::fabs(r[0][i]));
        }
      std::printf("  %-28s %8.3f ms = %7.1f ns per application per workgroup slot   max |diff to (0)| %.3e (max |value| %.3e)\n", name[v], ms[v],
                  ms[v] * 1e6 / reps, err, mx);
      if (!(err <= 1e-12 * mx))
        rc = 2;
    }
  return rc;
}
