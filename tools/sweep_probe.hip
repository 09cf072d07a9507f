// Development probe: the three sweeps of one 17^3 p = 4 lattice resident in LDS (two workgroups per CU, 512 workgroups x
// reps applications), as (0) whole lines in registers with 256 threads (lattice_sweeps), (1) the same lines streamed cell
// by cell, (2) streamed with the next cell's inputs prefetched, (3) 512 threads: half-line + quarter-line tasks
// (lattice_sweeps_wide).  Prints the time per lattice application and the difference of every variant to (0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idealii_multigrid_amd/csrc -Iinclude tools/sweep_probe.hip -o tools/bin/sweep_probe
#include "kernels.hpp"
#include "fe1d.hpp"

#include <cstdio>
#include <vector>

// The 512-thread sweeps measured here lived in kernels.hpp while a whole-kernel variant used them (round 2: not faster, see
// DESIGN.md); they are kept with their only user.
namespace mgamd
{
  // ---- WIDE sweeps: 512 threads on one 17-point lattice (4 waves per SIMD with two workgroups per CU instead of 2).
  // A sweep is 512 half-line tasks (lines 0..255, nodes 8 s .. 8 s + 7 (+ 16), s = tid & 1: the two halves of a line in
  // adjacent lanes) followed by 132 quarter-line tasks for the 33 left-over lines (as in lattice_sweeps).  A task of SEGN
  // nodes reads, UP FRONT, the P nodes of the cell to its left (whose last row acts on its first node) and the node to its
  // right (owned, and overwritten early, by the next task of the line); all tasks of a line sit in one wavefront and run in
  // lock step, so every lane has these before any lane stores (seg_fence).  Its own nodes are streamed cell by cell.
  // KIND as in line_stream.  A, Bb point at the task's first node.
  template <typename T, int P, int KIND, int SEGN>
  __device__ __forceinline__ void
  seg_task(const Mats<P> &m, T *__restrict__ A, T *__restrict__ Bb, const int stride, const bool has_left, const bool is_last,
           const T scale)
  {
    static_assert(SEGN % P == 0, "segment = whole cells");
    constexpr int n = P + 1, CPS = SEGN / P;
    T             la[n], lb[n];
#pragma unroll
    for (int j = 0; j < P; ++j)
      {
        la[j] = has_left ? A[(j - P) * stride] : T(0);
        lb[j] = (KIND != 0 && has_left) ? Bb[(j - P) * stride] : T(0);
      }
    la[P]      = A[0];
    lb[P]      = KIND != 0 ? Bb[0] : T(0);
    const T ra = A[SEGN * stride], rb = KIND != 0 ? Bb[SEGN * stride] : T(0);
    seg_fence();
    T c1 = T(0), c2 = T(0);
#pragma unroll
    for (int j = 0; j <= P; ++j)
      {
        const T Mj = T(m.M[P * (P + 1) + j]), Kj = T(m.K[P * (P + 1) + j]);
        if (KIND != 2)
          c1 += Mj * la[j];
        c2 += Kj * la[j];
        if (KIND != 0)
          c2 += Mj * lb[j];
      }
    if (!has_left)
      c1 = c2 = T(0);
    T a[n], b[n];
    a[0] = la[P];
    b[0] = lb[P];
#pragma unroll
    for (int c = 0; c < CPS; ++c)
      {
#pragma unroll
        for (int j = 1; j < n; ++j)
          {
            const bool right = c == CPS - 1 && j == P;
            a[j]             = right ? ra : A[(c * P + j) * stride];
            if (KIND != 0)
              b[j] = right ? rb : Bb[(c * P + j) * stride];
          }
        T o1[n], o2[n];
#pragma unroll
        for (int i = 0; i < n; ++i)
          o1[i] = o2[i] = T(0);
        if constexpr (P < 4)
          {
#pragma unroll
            for (int i = 0; i < n; ++i)
#pragma unroll
              for (int j = 0; j < n; ++j)
                {
                  if (KIND != 2)
                    o1[i] += T(m.M[i * n + j]) * a[j];
                  o2[i] += T(m.K[i * n + j]) * a[j];
                  if (KIND != 0)
                    o2[i] += T(m.M[i * n + j]) * b[j];
                }
          }
        else
          {
            EvenOdd<T, P> xa, xb, y;
            xa.split(a);
            if (KIND != 0)
              xb.split(b);
            if (KIND != 2)
              {
                y.template apply<false>(m.Me, m.Mo, xa);
                y.add_to(o1);
              }
            y.template apply<false>(m.Ke, m.Ko, xa);
            if (KIND != 0)
              y.template apply<true>(m.Me, m.Mo, xb);
            y.add_to(o2);
          }
        o1[0] += c1;
        o2[0] += c2;
#pragma unroll
        for (int j = 0; j < P; ++j)
          {
            if (KIND != 2)
              {
                A[(c * P + j) * stride]  = o1[j];
                Bb[(c * P + j) * stride] = o2[j];
              }
            else
              A[(c * P + j) * stride] = scale * o2[j];
          }
        c1   = o1[P];
        c2   = o2[P];
        a[0] = a[P];
        b[0] = b[P];
      }
    if (is_last)
      {
        if (KIND != 2)
          {
            A[SEGN * stride]  = c1;
            Bb[SEGN * stride] = c2;
          }
        else
          A[SEGN * stride] = scale * c2;
      }
  }

  // the three sweeps of ONE 17-point lattice with 512 threads; ends with a barrier
  template <typename T, int P, typename Hook = NoHook>
  __device__ __forceinline__ void
  lattice_sweeps_wide(T *__restrict__ bufA, T *__restrict__ bufB, const Mats<P> &m, const int tid, const T h, const Hook &before_x = Hook())
  {
    static_assert(16 % P == 0 && 8 % P == 0 && 4 % P == 0, "17-point lattices: P in {1, 2, 4}");
    constexpr int N = 17;
    // half-line task: line tid / 2, half tid % 2;  quarter-line task (tid < 132): line 256 + tid / 4, quarter tid % 4
    const int  hl = tid >> 1, hs = tid & 1, hu = hl % N, hv = hl / N;
    const int  ql = 256 + (tid >> 2), qs = tid & 3, qu = ql % N, qv = ql / N;
    const bool qt = tid < 4 * (N * N - 256);
    // z sweep: line (x = u, y = v), stride N^2
    {
      const int base = hv * N + hu + 8 * hs * N * N;
      seg_task<T, P, 0, 8>(m, bufA + base, bufB + base, N * N, hs > 0, hs == 1, T(1));
      if (qt)
        {
          const int b2 = qv * N + qu + 4 * qs * N * N;
          seg_task<T, P, 0, 4>(m, bufA + b2, bufB + b2, N * N, qs > 0, qs == 3, T(1));
        }
    }
    __syncthreads();
    // y sweep: line (x = u, z = v), stride N
    {
      const int base = hv * N * N + hu + 8 * hs * N;
      seg_task<T, P, 1, 8>(m, bufA + base, bufB + base, N, hs > 0, hs == 1, T(1));
      if (qt)
        {
          const int b2 = qv * N * N + qu + 4 * qs * N;
          seg_task<T, P, 1, 4>(m, bufA + b2, bufB + b2, N, qs > 0, qs == 3, T(1));
        }
    }
    __syncthreads();
    before_x();
    // x sweep: line (y = u, z = v), stride 1
    {
      const int base = (hv * N + hu) * N + 8 * hs;
      seg_task<T, P, 2, 8>(m, bufA + base, bufB + base, 1, hs > 0, hs == 1, h);
      if (qt)
        {
          const int b2 = (qv * N + qu) * N + 4 * qs;
          seg_task<T, P, 2, 4>(m, bufA + b2, bufB + b2, 1, qs > 0, qs == 3, h);
        }
    }
    __syncthreads();
  }

} // namespace mgamd

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
