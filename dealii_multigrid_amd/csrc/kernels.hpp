// HIP kernels for gfx950 (CDNA4, wave64).  Hand-written for MI355X: no hipify, no dual paths.
//
// K1  lattice_apply_kernel   the level operator (ref:include/operator.h:152-183,461-472) on one SLOT
//                            (brick of B^3 cells or single cell) per work-item group: node lattice
//                            of N = p*B+1 points per direction staged in LDS, three 1D sweeps with
//                            one thread per lattice line and the line in registers, epilogue fused
//                            (vmult / residual / Chebyshev update) for slot-interior DoFs, partial
//                            sums of shell DoFs atomically added to the small 'tail' accumulator.
// K1c cell_cluster_apply_kernel  the same operator on single cells at p = 1: one cell per thread, nodes deduplicated per
//                            256-cell cluster in LDS (one load and one global atomic per distinct node).
// K2  tail_kernel            same epilogue for the tail + constrained DoFs (identity rows).
// K3  lattice_diag_kernel    diagonal of C^T K C (ref:include/operator.h:228-242).
// K4  prolongate/restrict    MGTwoLevelTransfer embeddings (ref:multigrid_throughput.cc:1600-1604).
// K5  vector kernels         set/copy/axpy/sadd/scaled pointwise product/dot.
// K6  dense_matvec_kernel    coarse-grid direct solve (precomputed inverse).
//
// Because every cell is a cube (ref:include/grid_generator.h, MappingQ1) the brick operator is
//   A_brick = h (K (x) M (x) M + M (x) K (x) M + M (x) M (x) K)
// with 1D matrices assembled over the B cells of a lattice line; each 1D product is evaluated
// cell by cell with the dense (p+1)^2 reference matrices held in SGPRs (kernel arguments).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgamd
{
  constexpr uint32_t DEV_INVALID = 0xFFFFFFFFu;

  // Once-touched streams (slot-interior x / x_old / b / out, the tail epilogue's vectors, the fine vectors of the brick
  // transfers) are loaded and stored NON-TEMPORALLY, so that L2 and the Infinity Cache keep what IS touched again within a
  // pass: the shell values several bricks gather and the tail accumulator lines that take several atomic adds and are then
  // read by tail_kernel.  Measured: octant p=4 V-cycle 10.50 -> 10.26 ms, uniform p=1 8.25 -> 7.94 ms (same box, A-B-A).
  // -DMGAMD_NO_NT_STREAMS: plain loads and stores.
#ifndef MGAMD_NO_NT_STREAMS
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#define NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define NT_LOAD(p) (*(p))
#define NT_STORE(v, p) (*(p) = (v))
#endif

  // XCD-aware work mapping: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so
  // workgroup b works on item start(b % 8) + b / 8: every XCD gets one contiguous (Morton) range of the n items and
  // slots that share faces meet in the same L2 (+1.5 % at p=4, +2 % at p=1 once the kernels were spill-free).
  __device__ __forceinline__ uint32_t
  xcd_contiguous(uint32_t b, uint32_t n)
  {
    const uint32_t k = b & 7u, q = n >> 3, r = n & 7u;
    return k * q + (k < r ? k : r) + (b >> 3);
  }

  template <int P>
  struct Mats
  {
    double M[(P + 1) * (P + 1)];
    double K[(P + 1) * (P + 1)];
    double I0[(P + 1) * (P + 1)];
    double I1[(P + 1) * (P + 1)];
    // even-odd decomposition of the centrosymmetric M and K (GLL nodes are symmetric): A x = Ae xe + Ao xo with
    // xe_j = x_j + x_{P-j}, xo_j = x_j - x_{P-j};  Ae_ij = (A_ij + A_i,P-j)/2 (middle column: A_i,mid), Ao_ij = (A_ij - A_i,P-j)/2
    static constexpr int NH = (P + 2) / 2, NO = (P + 1) / 2;
    double               Me[NH * NH], Mo[NO * NO], Ke[NH * NH], Ko[NO * NO];
  };

  // Bricks that may carry whole-face / whole-edge hanging-node constraints (see lattice_apply_body): families (B = 2) share
  // the kernel of the 2^3 bricks; larger constrained bricks (p = 1 only, LevelTables::build) are a slot group of their own
  // with their own kernel instantiation (CONSTR), so that the plain bricks keep the lean kernel.
  constexpr bool
  brick_may_be_constrained(int B, bool constr)
  {
    return B == 2 || (constr && B > 2);
  }

  // THREADS: the threads that share one set of slots: a 256-thread workgroup, or ONE WAVE (64) for the wave-scoped single-cell
  // path (lattice_apply_body<..., WAVE = true>), where every wave of a workgroup works on its own cells without workgroup barriers
  template <int P, int B, int THREADS = 256>
  struct Geo
  {
    static constexpr int N       = P * B + 1;
    static constexpr int N3      = N * N * N;
    static constexpr int NI      = N - 2;
    static constexpr int N_INT   = NI > 0 ? NI * NI * NI : 0;
    static constexpr int N_SHELL = N3 - N_INT;
    static constexpr int LINES   = N * N;
    static constexpr int SPW     = LINES >= THREADS ? 1 : THREADS / LINES;
    static constexpr int BLOCK   = ((SPW * LINES + 63) / 64) * 64;
    // the operator kernel never uses more than 4 waves: two workgroups of 4 waves fit one CU with up to
    // 256 VGPRs each, whereas two 5-wave workgroups need 4 waves on one SIMD (<= 128 VGPRs).  Lines beyond
    // ABLOCK are handled in a second round by the first threads.
    static constexpr int ABLOCK  = BLOCK > 256 ? 256 : BLOCK;
    static constexpr int ROUNDS  = (SPW * LINES + ABLOCK - 1) / ABLOCK;
  };

  struct SlotGroupDev
  {
    const uint32_t *interior_base;
    const uint32_t *shell_idx;
    const uint16_t *mask;
    const double   *h;
    const uint16_t *shell_pos;
    uint32_t        n_slots;
    const uint32_t *fmask; // bricks: masks of the constrained ones (level_tables.hpp), nullptr if the group has none
  };

  enum ApplyMode
  {
    MODE_VMULT    = 0, // out = A x
    MODE_RESIDUAL = 1, // out = b - A x
    MODE_CHEB     = 2, // out = x + f1 (x - xold) + f2 dinv (b - A x)      (xold == nullptr: xold = 0; Epilogue::from_b)
    MODE_INVDIAG  = 3, // out = |d| > 1e-10 ? 1/d : 1                       (d delivered as 'A x')
    // zero-start Chebyshev without materialising x_1 = c0 dinv b:
    MODE_CHEB_FIRST  = 4, // MODE_CHEB with x := c0 dinv b computed on the fly (no x, no xold is read)
    MODE_CHEB_SECOND = 5, // MODE_CHEB with xold := c0 dinv b computed on the fly
    // level transfers FUSED into the operator pass that holds the same 17-point lattice in LDS (persistent brick kernel only;
    // every other kernel of such a pass runs the base mode):
    MODE_RESIDUAL_RESTRICT = 6, // MODE_RESIDUAL; bricks flagged as fused restrict their part of b - A x into the coarse defect
                                // (interior rows complete, shell rows as partial sums) instead of storing it
    MODE_CHEB_PROLONGATE = 7    // MODE_CHEB (x_old = 0: first pass of a smoothing step) on x + P x_c: fused bricks add the
                                // coarse correction on their lattice while gathering x and store x + P x_c once
  };
  constexpr int
  base_mode(int mode)
  {
    return mode == MODE_RESIDUAL_RESTRICT ? MODE_RESIDUAL : (mode == MODE_CHEB_PROLONGATE ? MODE_CHEB : mode);
  }

  template <typename T>
  struct Epilogue
  {
    T       *out;
    const T *x; // operator input (src)
    const T *xold;
    const T *b;
    const T *dinv;
    T        f1, f2;
    T        c0; // MODE_CHEB_FIRST / MODE_CHEB_SECOND
    // tail_kernel only: D^-1 of the tail / constrained DoFs as a one-byte code into a table of the 255 most frequent values
    // of this level (255: read dinv[]); entry i belongs to DoF n_interior + i.  Bit-identical values, 7 bytes less per DoF.
    const uint8_t *dinv_code  = nullptr;
    const T       *dinv_table = nullptr;
    // tail_kernel<MODE_CHEB_PROLONGATE> only: tail DoF n_interior + i with xs_flag[i] != 0 is owned by a fused brick, which has
    // left x_i + (P x_c)_i in xs (indexed like x); the kernel uses it as x and stores it to x_inout (= x), so that the next pass
    // finds x + P x_c everywhere
    const uint8_t *xs_flag = nullptr;
    const T       *xs      = nullptr;
    T             *x_inout = nullptr;
  };
  constexpr bool
  is_cheb(int mode)
  {
    return mode == MODE_CHEB || mode == MODE_CHEB_FIRST || mode == MODE_CHEB_SECOND;
  }

  template <typename T, int MODE>
  __device__ __forceinline__ void
  apply_epilogue(const Epilogue<T> &e, uint32_t gi, T Ax)
  {
    if (MODE == MODE_VMULT)
      e.out[gi] = Ax;
    else if (MODE == MODE_RESIDUAL)
      e.out[gi] = e.b[gi] - Ax;
    else if (MODE == MODE_CHEB)
      {
        const T xv = e.x[gi];
        const T xo = e.xold ? e.xold[gi] : T(0);
        e.out[gi]  = xv + e.f1 * (xv - xo) + e.f2 * e.dinv[gi] * (e.b[gi] - Ax);
      }
    else
      e.out[gi] = (fabs((double)Ax) > 1.0e-10) ? T(1) / Ax : T(1);
  }

  __device__ __forceinline__ void
  atomic_add(double *p, double v)
  {
    unsafeAtomicAdd(p, v);
  }
  __device__ __forceinline__ void
  atomic_add(float *p, float v)
  {
    unsafeAtomicAdd(p, v);
  }

  // out[0..N) = (1D matrix assembled from B copies of the (P+1)^2 cell matrix Mc) * in
  template <typename T, int P, int B>
  __device__ __forceinline__ void
  line_mult(const double *__restrict__ Mc, const T (&in)[P * B + 1], T (&out)[P * B + 1])
  {
#pragma unroll
    for (int i = 0; i < P * B + 1; ++i)
      out[i] = T(0);
#pragma unroll
    for (int c = 0; c < B; ++c)
#pragma unroll
      for (int a = 0; a <= P; ++a)
#pragma unroll
        for (int b = 0; b <= P; ++b)
          out[c * P + a] += T(Mc[a * (P + 1) + b]) * in[c * P + b];
  }

  // ---- the same products through the even-odd decomposition: 13 multiply-adds instead of 25 per 5x5 block, and the
  // even/odd splits of a line are shared by all products that use it (deal.II's sum factorisation does the same).
  template <typename T, int P>
  struct EvenOdd
  {
    static constexpr int NH = (P + 2) / 2, NO = (P + 1) / 2;
    T                    e[NH], o[NO];
    __device__ __forceinline__ void
    split(const T *x) // x[0..P]
    {
#pragma unroll
      for (int j = 0; j < NO; ++j)
        {
          e[j] = x[j] + x[P - j];
          o[j] = x[j] - x[P - j];
        }
      if (NH > NO)
        e[NH - 1] = x[NH - 1];
    }
    // this = Ae * xe, Ao * xo (accumulating if ACC)
    template <bool ACC>
    __device__ __forceinline__ void
    apply(const double *__restrict__ Ae, const double *__restrict__ Ao, const EvenOdd &x)
    {
#pragma unroll
      for (int i = 0; i < NH; ++i)
        {
          T acc = ACC ? e[i] : T(0);
#pragma unroll
          for (int j = 0; j < NH; ++j)
            acc += T(Ae[i * NH + j]) * x.e[j];
          e[i] = acc;
        }
#pragma unroll
      for (int i = 0; i < NO; ++i)
        {
          T acc = ACC ? o[i] : T(0);
#pragma unroll
          for (int j = 0; j < NO; ++j)
            acc += T(Ao[i * NO + j]) * x.o[j];
          o[i] = acc;
        }
    }
    // y[0..P] += recombination
    __device__ __forceinline__ void
    add_to(T *y) const
    {
#pragma unroll
      for (int i = 0; i < NO; ++i)
        {
          y[i] += e[i] + o[i];
          y[P - i] += e[i] - o[i];
        }
      if (NH > NO)
        y[NH - 1] += e[NH - 1];
    }
  };

  // outM = M a, outK = K a
  template <typename T, int P, int B>
  __device__ __forceinline__ void
  line_MK(const Mats<P> &m, const T (&a)[P * B + 1], T (&outM)[P * B + 1], T (&outK)[P * B + 1])
  {
    if constexpr (P < 4) // no saving below 5x5 blocks
      {
        line_mult<T, P, B>(m.M, a, outM);
        line_mult<T, P, B>(m.K, a, outK);
        return;
      }
#pragma unroll
    for (int i = 0; i < P * B + 1; ++i)
      outM[i] = outK[i] = T(0);
#pragma unroll
    for (int c = 0; c < B; ++c)
      {
        EvenOdd<T, P> xa, y;
        xa.split(&a[c * P]);
        y.template apply<false>(m.Me, m.Mo, xa);
        y.add_to(&outM[c * P]);
        y.template apply<false>(m.Ke, m.Ko, xa);
        y.add_to(&outK[c * P]);
      }
  }
  // outM = M a, outS = K a + M b
  template <typename T, int P, int B>
  __device__ __forceinline__ void
  line_M_KM(const Mats<P> &m, const T (&a)[P * B + 1], const T (&b)[P * B + 1], T (&outM)[P * B + 1], T (&outS)[P * B + 1])
  {
    if constexpr (P < 4)
      {
        T t[P * B + 1];
        line_mult<T, P, B>(m.M, a, outM);
        line_mult<T, P, B>(m.K, a, outS);
        line_mult<T, P, B>(m.M, b, t);
#pragma unroll
        for (int i = 0; i < P * B + 1; ++i)
          outS[i] += t[i];
        return;
      }
#pragma unroll
    for (int i = 0; i < P * B + 1; ++i)
      outM[i] = outS[i] = T(0);
#pragma unroll
    for (int c = 0; c < B; ++c)
      {
        EvenOdd<T, P> xa, xb, y;
        xa.split(&a[c * P]);
        xb.split(&b[c * P]);
        y.template apply<false>(m.Me, m.Mo, xa);
        y.add_to(&outM[c * P]);
        y.template apply<false>(m.Ke, m.Ko, xa);
        y.template apply<true>(m.Me, m.Mo, xb);
        y.add_to(&outS[c * P]);
      }
  }
  // outS = K a + M b
  template <typename T, int P, int B>
  __device__ __forceinline__ void
  line_KM(const Mats<P> &m, const T (&a)[P * B + 1], const T (&b)[P * B + 1], T (&outS)[P * B + 1])
  {
    if constexpr (P < 4)
      {
        T t[P * B + 1];
        line_mult<T, P, B>(m.K, a, outS);
        line_mult<T, P, B>(m.M, b, t);
#pragma unroll
        for (int i = 0; i < P * B + 1; ++i)
          outS[i] += t[i];
        return;
      }
#pragma unroll
    for (int i = 0; i < P * B + 1; ++i)
      outS[i] = T(0);
#pragma unroll
    for (int c = 0; c < B; ++c)
      {
        EvenOdd<T, P> xa, xb, y;
        xa.split(&a[c * P]);
        xb.split(&b[c * P]);
        y.template apply<false>(m.Ke, m.Ko, xa);
        y.template apply<true>(m.Me, m.Mo, xb);
        y.add_to(&outS[c * P]);
      }
  }

  // ---- one lattice line, in place, CELL BY CELL (streamed): the inputs of a cell are read, its products formed, its first
  // P nodes stored (the first one with the carry of the previous cell), its last node carried on.  Same arithmetic as
  // line_MK / line_M_KM / line_KM on whole lines, but ~35 instead of ~70 doubles in registers per thread: what lets the
  // persistent kernel keep its epilogue operands and the next slot's tables in flight across the sweeps.
  // KIND 0: A <- M a, Bb <- K a;  1: A <- M a, Bb <- K a + M b;  2: A <- scale (K a + M b)     (a from A, b from Bb)
  // PREFETCH: the inputs of the next cell are requested before this cell's products (LDS latency under the arithmetic; 8
  // more doubles in registers: measured -4 % on the 2-4-word passes, but the 5-word Chebyshev pass then spills)
  template <typename T, int P, int B, int KIND, bool PREFETCH>
  __device__ __forceinline__ void
  line_stream(const Mats<P> &m, T *__restrict__ A, T *__restrict__ Bb, const int stride, const T scale)
  {
    constexpr int n = P + 1;
    T             a[n], b[n], an[n], bn[n], c1 = T(0), c2 = T(0);
    a[0] = A[0];
    b[0] = KIND == 0 ? T(0) : Bb[0];
    if (PREFETCH)
      {
#pragma unroll
        for (int j = 1; j < n; ++j)
          {
            a[j] = A[j * stride];
            b[j] = KIND == 0 ? T(0) : Bb[j * stride];
          }
      }
#pragma unroll
    for (int c = 0; c < B; ++c)
      {
        if (!PREFETCH)
          {
#pragma unroll
            for (int j = 1; j < n; ++j)
              {
                a[j] = A[(c * P + j) * stride];
                if (KIND != 0)
                  b[j] = Bb[(c * P + j) * stride];
              }
          }
        else if (c + 1 < B)
          {
#pragma unroll
            for (int j = 1; j < n; ++j)
              {
                an[j] = A[((c + 1) * P + j) * stride];
                bn[j] = KIND == 0 ? T(0) : Bb[((c + 1) * P + j) * stride];
              }
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
            if (KIND == 0)
              {
                A[(c * P + j) * stride]  = o1[j];
                Bb[(c * P + j) * stride] = o2[j];
              }
            else if (KIND == 1)
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
        if (PREFETCH)
          {
#pragma unroll
            for (int j = 1; j < n; ++j)
              {
                a[j] = an[j];
                b[j] = bn[j];
              }
          }
      }
    if (KIND != 2)
      {
        A[P * B * stride]  = c1;
        Bb[P * B * stride] = c2;
      }
    else
      A[P * B * stride] = scale * c2;
  }

  // ---- SEGMENT tasks for the 17-point lattices.  17^2 = 289 lines do not fit one round of 256 threads, and a second round
  // of whole lines runs with 33 of 256 lanes (measured: the sweeps are 5.9 of the 13-16 us a workgroup lives).  The 33
  // left-over lines are cut into 4 segments of 5 nodes (4 s .. 4 s + 4: one cell at p = 4, two at p = 2, four at p = 1):
  // 132 tasks of a quarter line each.  A task owns the nodes 4 s .. 4 s + 3 (and node 16 for s = 3): it also adds the
  // contribution of the cell to its left to node 4 s, for which it reads that cell's other P nodes.  The four tasks of a
  // line sit in adjacent lanes of ONE wavefront and update the line in place: every lane has read its inputs before any
  // lane writes (lock step, LDS operations of a wave complete in order; seg_fence() keeps the compiler from sinking a
  // load below the stores).
  // barrier among the threads that share a set of slots: the workgroup, or -- WAVE -- one wavefront, whose LDS operations
  // are issued and completed in order (no s_barrier: only the compiler must not move LDS accesses across it)
  template <bool WAVE>
  __device__ __forceinline__ void
  slot_sync()
  {
    if constexpr (WAVE)
      {
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    else
      __syncthreads();
  }

  __device__ __forceinline__ void
  seg_fence()
  {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // KIND 0: o1 = M a, o2 = K a;  1: o1 = M a, o2 = K a + M b;  2: o2 = K a + M b.   a, b: [P left nodes | 5 own nodes]
  template <typename T, int P, int KIND>
  __device__ __forceinline__ void
  seg_products(const Mats<P> &m, const T (&a)[P + 5], const T (&b)[P + 5], const bool has_left, T (&o1)[5], T (&o2)[5])
  {
    static_assert(4 % P == 0, "segments of 5 nodes need P in {1, 2, 4}");
    constexpr int CPS = 4 / P; // cells per segment
    T             oa[5], ob[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
      {
        oa[i] = a[P + i];
        ob[i] = b[P + i];
      }
    if constexpr (KIND == 0)
      line_MK<T, P, CPS>(m, oa, o1, o2);
    else if constexpr (KIND == 1)
      line_M_KM<T, P, CPS>(m, oa, ob, o1, o2);
    else
      line_KM<T, P, CPS>(m, oa, ob, o2);
    // left cell [4 s - P, 4 s]: its last row acts on node 4 s
    T l1 = T(0), l2 = T(0);
#pragma unroll
    for (int j = 0; j <= P; ++j)
      {
        const T Mj = T(m.M[P * (P + 1) + j]), Kj = T(m.K[P * (P + 1) + j]);
        if constexpr (KIND == 0)
          {
            l1 += Mj * a[j];
            l2 += Kj * a[j];
          }
        else if constexpr (KIND == 1)
          {
            l1 += Mj * a[j];
            l2 += Kj * a[j] + Mj * b[j];
          }
        else
          l2 += Kj * a[j] + Mj * b[j];
      }
    if (has_left)
      {
        if constexpr (KIND != 2)
          o1[0] += l1;
        o2[0] += l2;
      }
  }

  // The three sweeps.  Line l = tid + r*BLOCK (r < ROUNDS) of the workgroup is (slot sl, u, v) in every sweep.
  // bufA holds the input and receives the result; bufB is scratch.  Ends with a barrier.
  struct NoHook
  {
    __device__ __forceinline__ void
    operator()() const
    {}
  };
  // before_x: called between the y and the x sweep (the x sweep holds one line less in registers than the y sweep: the
  // persistent kernel requests its epilogue operands there)
  template <typename T, int P, int B, int BLOCK, typename Hook = NoHook, bool STREAMED = false, bool PREFETCH = false, bool WAVE = false>
  __device__ __forceinline__ void
  lattice_sweeps(T *__restrict__ bufA, T *__restrict__ bufB, const Mats<P> &m, int tid, int nslots, const double *__restrict__ hslot,
                 const Hook &before_x = Hook(), const bool h_is_mine = false) // h_is_mine: hslot[0] is the h of THIS thread's line
  {
    using G              = Geo<P, B, WAVE ? 64 : 256>;
    constexpr int N      = G::N;
    constexpr int N3     = G::N3;
    constexpr int TOT    = G::SPW * G::LINES;
    // 17-point lattices: one round of whole lines + segment tasks for the rest
    constexpr bool SEGMENTS = N == 17 && G::SPW == 1 && TOT > BLOCK && 4 * (TOT - BLOCK) <= BLOCK && (4 % P == 0);
    constexpr int  ROUNDS   = SEGMENTS ? 1 : (TOT + BLOCK - 1) / BLOCK;
    constexpr int  NSEG     = SEGMENTS ? 4 * (TOT - BLOCK) : 0;
    constexpr bool STREAM   = STREAMED; // whole lines cell by cell (line_stream)
    // segment task of this thread: line BLOCK + tid / 4, segment tid % 4
    const int  sg_l = BLOCK + (tid >> 2), sg_s = tid & 3, sg_u = sg_l % N, sg_v = sg_l / N;
    const bool sg   = SEGMENTS && tid < NSEG;
    T          r0[N], r1[N], r2[N];
    // z sweep: line = (x=u, y=v)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      {
        const int l = tid + r * BLOCK, sl = l / G::LINES, ln = l % G::LINES, u = ln % N, v = ln / N;
        if (l < TOT && sl < nslots)
          {
            const int base = sl * N3 + v * N + u;
            if constexpr (STREAM)
              line_stream<T, P, B, 0, PREFETCH>(m, bufA + base, bufB + base, N * N, T(1));
            else
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  r0[i] = bufA[base + i * N * N];
                line_MK<T, P, B>(m, r0, r1, r2);
#pragma unroll
                for (int i = 0; i < N; ++i)
                  {
                    bufA[base + i * N * N] = r1[i];
                    bufB[base + i * N * N] = r2[i];
                  }
              }
          }
      }
    if constexpr (SEGMENTS)
      if (sg)
        {
          const int base = sg_v * N + sg_u + 4 * sg_s * N * N;
          T         a[P + 5], o1[5], o2[5];
#pragma unroll
          for (int i = 0; i < P + 5; ++i)
            a[i] = (sg_s > 0 || i >= P) ? bufA[base + (i - P) * N * N] : T(0);
          seg_fence();
          seg_products<T, P, 0>(m, a, a, sg_s > 0, o1, o2);
#pragma unroll
          for (int i = 0; i < 5; ++i)
            if (i < 4 || sg_s == 3)
              {
                bufA[base + i * N * N] = o1[i];
                bufB[base + i * N * N] = o2[i];
              }
        }
    slot_sync<WAVE>();
    // y sweep: line = (x=u, z=v):  c = My a ; g = Ky a + My b
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      {
        const int l = tid + r * BLOCK, sl = l / G::LINES, ln = l % G::LINES, u = ln % N, v = ln / N;
        if (l < TOT && sl < nslots)
          {
            const int base = sl * N3 + v * N * N + u;
            if constexpr (STREAM)
              line_stream<T, P, B, 1, PREFETCH>(m, bufA + base, bufB + base, N, T(1));
            else
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  r0[i] = bufA[base + i * N];
                T rb[N];
#pragma unroll
                for (int i = 0; i < N; ++i)
                  rb[i] = bufB[base + i * N];
                line_M_KM<T, P, B>(m, r0, rb, r1, r2);
#pragma unroll
                for (int i = 0; i < N; ++i)
                  {
                    bufA[base + i * N] = r1[i];
                    bufB[base + i * N] = r2[i];
                  }
              }
          }
      }
    if constexpr (SEGMENTS)
      if (sg)
        {
          const int base = sg_v * N * N + sg_u + 4 * sg_s * N;
          T         a[P + 5], b[P + 5], o1[5], o2[5];
#pragma unroll
          for (int i = 0; i < P + 5; ++i)
            {
              a[i] = (sg_s > 0 || i >= P) ? bufA[base + (i - P) * N] : T(0);
              b[i] = (sg_s > 0 || i >= P) ? bufB[base + (i - P) * N] : T(0);
            }
          seg_fence();
          seg_products<T, P, 1>(m, a, b, sg_s > 0, o1, o2);
#pragma unroll
          for (int i = 0; i < 5; ++i)
            if (i < 4 || sg_s == 3)
              {
                bufA[base + i * N] = o1[i];
                bufB[base + i * N] = o2[i];
              }
        }
    slot_sync<WAVE>();
    before_x();
    // x sweep: line = (y=u, z=v): out = h (Kx c + Mx g)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      {
        const int l = tid + r * BLOCK, sl = l / G::LINES, ln = l % G::LINES, u = ln % N, v = ln / N;
        if (l < TOT && sl < nslots)
          {
            const T   h    = T(h_is_mine ? hslot[0] : hslot[sl]);
            const int base = sl * N3 + (v * N + u) * N;
            if constexpr (STREAM)
              line_stream<T, P, B, 2, PREFETCH>(m, bufA + base, bufB + base, 1, h);
            else
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  r0[i] = bufA[base + i];
#pragma unroll
                for (int i = 0; i < N; ++i)
                  r1[i] = bufB[base + i];
                line_KM<T, P, B>(m, r0, r1, r2);
#pragma unroll
                for (int i = 0; i < N; ++i)
                  bufA[base + i] = h * r2[i];
              }
          }
      }
    if constexpr (SEGMENTS)
      if (sg)
        {
          const T   h    = T(hslot[0]);
          const int base = (sg_v * N + sg_u) * N + 4 * sg_s;
          T         a[P + 5], b[P + 5], o1[5], o2[5];
#pragma unroll
          for (int i = 0; i < P + 5; ++i)
            {
              a[i] = (sg_s > 0 || i >= P) ? bufA[base + (i - P)] : T(0);
              b[i] = (sg_s > 0 || i >= P) ? bufB[base + (i - P)] : T(0);
            }
          seg_fence();
          seg_products<T, P, 2>(m, a, b, sg_s > 0, o1, o2);
#pragma unroll
          for (int i = 0; i < 5; ++i)
            if (i < 4 || sg_s == 3)
              bufA[base + i] = h * o2[i];
        }
    slot_sync<WAVE>();
  }

  // In-cell hanging-node interpolation (transpose = false, before the sweeps) or its transpose
  // (after), for single-cell slots (N = P+1).  One thread per line; only lines on hanging
  // faces/edges do work.  Ends with a barrier.
  template <typename T, int P, bool WAVE = false>
  __device__ __forceinline__ void
  hanging_passes(T *__restrict__ buf, const Mats<P> &m, int sl, int u, int v, bool act, uint32_t mask, bool transpose)
  {
    constexpr int N  = P + 1;
    constexpr int N3 = N * N * N;
    const int     cx = mask & 1, cy = (mask >> 1) & 1, cz = (mask >> 2) & 1;
    const bool    fx = (mask >> 3) & 1, fy = (mask >> 4) & 1, fz = (mask >> 5) & 1;
    const bool    ex = (mask >> 6) & 1, ey = (mask >> 7) & 1, ez = (mask >> 8) & 1;
#pragma unroll
    for (int dd = 0; dd < 3; ++dd)
      {
        const int d = transpose ? 2 - dd : dd;
        bool      on;
        int       base, stride, c;
        if (d == 0)
          { // x lines, (u,v) = (y,z)
            const bool ou = u == cy * P, ov = v == cz * P;
            on            = (fy && ou) || (fz && ov) || (ex && ou && ov);
            base          = sl * N3 + (v * N + u) * N;
            stride        = 1;
            c             = cx;
          }
        else if (d == 1)
          { // y lines, (u,v) = (x,z)
            const bool ou = u == cx * P, ov = v == cz * P;
            on            = (fx && ou) || (fz && ov) || (ey && ou && ov);
            base          = sl * N3 + v * N * N + u;
            stride        = N;
            c             = cy;
          }
        else
          { // z lines, (u,v) = (x,y)
            const bool ou = u == cx * P, ov = v == cy * P;
            on            = (fx && ou) || (fy && ov) || (ez && ou && ov);
            base          = sl * N3 + v * N + u;
            stride        = N * N;
            c             = cz;
          }
        if (act && on && (mask >> 3))
          {
            T in[N], out[N];
#pragma unroll
            for (int i = 0; i < N; ++i)
              in[i] = buf[base + i * stride];
#pragma unroll
            for (int a = 0; a < N; ++a)
              {
                T s = T(0);
#pragma unroll
                for (int b = 0; b < N; ++b)
                  {
                    const int    k = transpose ? b * N + a : a * N + b;
                    const double w = c ? m.I1[k] : m.I0[k];
                    s += T(w) * in[b];
                  }
                out[a] = s;
              }
#pragma unroll
            for (int i = 0; i < N; ++i)
              buf[base + i * stride] = out[i];
          }
        slot_sync<WAVE>();
      }
  }

  // Constrained bricks (level_tables.hpp): bricks next to coarser cells whose hanging entities are whole faces / whole edges
  // of the brick (B = 2: a family, the 8 children of one cell).  The parents' face/edge DoFs sit ON the entity: along a
  // lattice line the parent DoF k P + c of parent cell k is at lattice coordinate 2 k P + c (c < P; c = P: the next parent
  // cell's first position, B P at the end); this embeds them in place along every lattice line that lies in a hanging face
  // (or is a hanging edge), direction by direction (x, y, z), with E = [I0; I1] per parent cell; transpose = the reverse.
  // One thread per line (sl, u, v) as in the sweeps, in rounds of BLOCK lines.  Ends with a barrier.
  template <typename T, int P, int B, int BLOCK>
  __device__ __forceinline__ void
  brick_constraint_passes(T *__restrict__ buf, const Mats<P> &m, int tid, int nslots, const uint32_t *__restrict__ fmask, bool transpose,
                          const uint32_t *fm_mine = nullptr) // fm_mine: the mask of this thread's line(s), already in a register
  {
    using G               = Geo<P, B>;
    constexpr int N       = G::N;
    constexpr int N3      = G::N3;
    constexpr int n       = P + 1;
    constexpr int BC      = B / 2;      // parent cells per direction
    constexpr int NCL     = P * BC + 1; // parent DoFs per line
    constexpr int TOT     = G::SPW * G::LINES;
    constexpr int ROUNDS  = (TOT + BLOCK - 1) / BLOCK;
    // the masks of this thread's lines, loaded ONCE (a global load per direction and round would sit on the critical path)
    uint32_t fmr[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      {
        const int l = tid + r * BLOCK, sl = l / G::LINES;
        if (fm_mine != nullptr) // every line of this thread lies in ONE slot (one line per thread, or one slot per workgroup)
          fmr[r] = (l < TOT && sl < nslots) ? *fm_mine : 0u;
        else
          fmr[r] = (l < TOT && sl < nslots && fmask != nullptr) ? fmask[sl] : 0u;
      }
#pragma unroll
    for (int dd = 0; dd < 3; ++dd)
      {
        const int d = transpose ? 2 - dd : dd;
        // (u, v) are the coordinates in directions (e, f): d = 0: (y, z); d = 1: (x, z); d = 2: (x, y)
        const int e = d == 0 ? 1 : 0, f = d == 2 ? 1 : 2;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
          {
            const int  l = tid + r * BLOCK, sl = l / G::LINES, ln = l % G::LINES, u = ln % N, v = ln / N;
            const uint32_t fm = fmr[r];
            const bool     xu = u == 0 || u == N - 1, xv = v == 0 || v == N - 1;
            const int  su = u == N - 1, sv = v == N - 1;
            // edge along d at sides (s1 of (d+1)%3, s2 of (d+2)%3)
            const int  s1 = d == 1 ? sv : su, s2 = d == 1 ? su : sv;
            const bool on = fm != 0 &&
                            (((u == 0 && ((fm >> (2 * e)) & 1)) || (u == N - 1 && ((fm >> (2 * e + 1)) & 1)) ||
                              (v == 0 && ((fm >> (2 * f)) & 1)) || (v == N - 1 && ((fm >> (2 * f + 1)) & 1))) ||
                             (xu && xv && ((fm >> (6 + 4 * d + s1 + 2 * s2)) & 1)));
            const int base   = d == 0 ? sl * N3 + (v * N + u) * N : (d == 1 ? sl * N3 + v * N * N + u : sl * N3 + v * N + u);
            const int stride = d == 0 ? 1 : (d == 1 ? N : N * N);
            if (on)
              {
                T line[N];
#pragma unroll
                for (int i = 0; i < N; ++i)
                  line[i] = buf[base + i * stride];
                if (!transpose)
                  {
                    T par[NCL];
#pragma unroll
                    for (int k = 0; k < BC; ++k)
#pragma unroll
                      for (int b = 0; b < P; ++b)
                        par[k * P + b] = line[2 * k * P + b];
                    par[NCL - 1] = line[N - 1];
#pragma unroll
                    for (int k = 0; k < BC; ++k)
#pragma unroll
                      for (int a = (k == 0 ? 0 : 1); a <= 2 * P; ++a)
                        {
                          T acc = T(0);
#pragma unroll
                          for (int b = 0; b < n; ++b)
                            acc += T(a <= P ? m.I0[a * n + b] : m.I1[(a - P) * n + b]) * par[k * P + b];
                          buf[base + (2 * k * P + a) * stride] = acc;
                        }
                  }
                else
                  {
                    T par[NCL];
#pragma unroll
                    for (int i = 0; i < NCL; ++i)
                      par[i] = T(0);
#pragma unroll
                    for (int k = 0; k < BC; ++k)
#pragma unroll
                      for (int a = (k == 0 ? 0 : 1); a <= 2 * P; ++a) // a fine node shared by two parent cells counts once
#pragma unroll
                        for (int b = 0; b < n; ++b)
                          par[k * P + b] += T(a <= P ? m.I0[a * n + b] : m.I1[(a - P) * n + b]) * line[2 * k * P + a];
#pragma unroll
                    for (int k = 0; k < BC; ++k)
#pragma unroll
                      for (int a = 0; a < 2 * P; ++a)
                        buf[base + (2 * k * P + a) * stride] = a < P ? par[k * P + a] : T(0);
                    buf[base + (N - 1) * stride] = par[NCL - 1];
                  }
              }
          }
        __syncthreads();
      }
  }

  // ---- 1D embedding of the h-transfer along one lattice line (brick transfers and the transfers fused into the operator)
  // fine line (P*BC*2+1) from coarse line (P*BC+1), cell by cell
  template <typename T, int P, int BC>
  __device__ __forceinline__ void
  line_embed(const double *__restrict__ E, const T (&in)[P * BC + 1], T (&out)[2 * P * BC + 1])
  {
#pragma unroll
    for (int c = 0; c < BC; ++c)
#pragma unroll
      for (int a = (c == 0 ? 0 : 1); a <= 2 * P; ++a)
        {
          T s = T(0);
#pragma unroll
          for (int b = 0; b <= P; ++b)
            s += T(E[a * (P + 1) + b]) * in[c * P + b];
          out[c * 2 * P + a] = s;
        }
  }
  // transpose: coarse line += E^T fine line; fine nodes shared by two coarse cells are counted once
  template <typename T, int P, int BC>
  __device__ __forceinline__ void
  line_embed_T(const double *__restrict__ E, const T (&in)[2 * P * BC + 1], T (&out)[P * BC + 1])
  {
#pragma unroll
    for (int i = 0; i < P * BC + 1; ++i)
      out[i] = T(0);
#pragma unroll
    for (int c = 0; c < BC; ++c)
#pragma unroll
      for (int a = (c == 0 ? 0 : 1); a <= 2 * P; ++a)
#pragma unroll
        for (int b = 0; b <= P; ++b)
          out[c * P + b] += T(E[a * (P + 1) + b]) * in[c * 2 * P + a];
  }

  // The same with HALF of the matrix: the GLL nodes are symmetric, so E[2P - a][P - b] = E[a][b], and the rows of fine nodes that
  // coincide with a coarse node (a = 0; a = P for even P) are unit vectors.  Eh = rows 0..P of E; only the entries of the
  // non-trivial rows are ever read (15 doubles at p = 4 instead of 45: the fused kernels keep them in scalar registers next to
  // the operator's matrices -- with the full matrix the compiler spilled scalars into vector lanes and vectors to scratch,
  // whose reloads wait for EVERY outstanding memory operation of the wave).
  template <int P>
  __device__ __forceinline__ constexpr bool
  embed_row_is_unit(int a) // a in 0..P
  {
    return a == 0 || (P % 2 == 0 && a == P);
  }
  template <typename T, int P, int BC>
  __device__ __forceinline__ void
  line_embed_sym(const double *__restrict__ Eh, const T (&in)[P * BC + 1], T (&out)[2 * P * BC + 1])
  {
#pragma unroll
    for (int c = 0; c < BC; ++c)
#pragma unroll
      for (int a = (c == 0 ? 0 : 1); a <= 2 * P; ++a)
        {
          const int  ar  = a <= P ? a : 2 * P - a; // row of Eh
          const bool mir = a > P;                  // mirrored columns
          T          s;
          if (embed_row_is_unit<P>(ar))
            s = in[c * P + (ar == 0 ? (mir ? P : 0) : P / 2)];
          else
            {
              s = T(0);
#pragma unroll
              for (int b = 0; b <= P; ++b)
                s += T(Eh[ar * (P + 1) + b]) * in[c * P + (mir ? P - b : b)];
            }
          out[c * 2 * P + a] = s;
        }
  }
  template <typename T, int P, int BC>
  __device__ __forceinline__ void
  line_embed_sym_T(const double *__restrict__ Eh, const T (&in)[2 * P * BC + 1], T (&out)[P * BC + 1])
  {
#pragma unroll
    for (int i = 0; i < P * BC + 1; ++i)
      out[i] = T(0);
#pragma unroll
    for (int c = 0; c < BC; ++c)
#pragma unroll
      for (int a = (c == 0 ? 0 : 1); a <= 2 * P; ++a) // a fine node shared by two coarse cells counts once
        {
          const int  ar  = a <= P ? a : 2 * P - a;
          const bool mir = a > P;
          const T    v   = in[c * 2 * P + a];
          if (embed_row_is_unit<P>(ar))
            out[c * P + (ar == 0 ? (mir ? P : 0) : P / 2)] += v;
          else
            {
#pragma unroll
              for (int b = 0; b <= P; ++b)
                out[c * P + (mir ? P - b : b)] += T(Eh[ar * (P + 1) + b]) * v;
            }
        }
  }

  // Tables of the transfers fused into the operator (Transfer2 in runtime.hip builds them, indexed by SLOT of the fused group):
  //   flags[slot * 256 + tid]  bit 15: the slot is fused; bits 2 it, 2 it + 1: BrickTransferGroup::SHELL_OWN / SHELL_OTHER of
  //                            shell entry tid + 256 it of that slot; bit 16 + it: coarse lattice node tid + 256 it is a
  //                            Dirichlet DoF (its coarse_idx entry is DEV_INVALID)
  //   coarse_idx[slot * NC^3 + c]  coarse DoF of coarse lattice node c (x fastest), DEV_INVALID = Dirichlet or slot not fused
  template <typename T, int P>
  struct FusedTransferDev
  {
    const uint32_t *flags      = nullptr;
    const uint32_t *coarse_idx = nullptr;
    double          Eh[(P + 1) * (P + 1)]; // rows 0..P of the 1D h-embedding (line_embed_sym)
    T              *coarse  = nullptr; // RESTRICT: the coarse defect (+=);  PROLONGATE: the coarse solution (read only)
    T              *x_inout = nullptr; // PROLONGATE: == src; x + P x_c of the fused bricks' interior DoFs is stored in place
    T              *scratch = nullptr; // PROLONGATE: x + P x_c of the shell DoFs a fused brick owns (indexed like x)
  };

  template <typename T, int P>
  struct ApplyArgs
  {
    SlotGroupDev g;
    Mats<P>      m;
    const T     *src;
    T           *tail_acc; // [n_tail] accumulators of shell partial sums
    uint32_t     n_interior;
    unsigned long long *stamps; // debug only (MGAMD_STAMPS): 8 wall-clock stamps per workgroup, nullptr normally
    uint32_t     ablate; // debug only (MGAMD_ABLATE): 1 no sweeps, 2 no shell atomics, 4 no interior epilogue, 8 no interior gather, 16 no shell gather
    Epilogue<T>  epi;
    // shell entries are gathered if their index is below gather_limit and receive partial sums if it is below
    // scatter_limit.  Dirichlet entries are DEV_INVALID (above every limit).  Local-smoothing levels number their
    // refinement-edge DoFs right after the tail: the level operator keeps them out (both limits = first edge index), the
    // residual scatters to their rows, the edge matrix gathers and scatters them (runtime.hip, EdgeMode).
    uint32_t gather_limit, scatter_limit;
    FusedTransferDev<T, P> fused; // MODE_RESIDUAL_RESTRICT / MODE_CHEB_PROLONGATE only
  };

  // Interior-slot bookkeeping shared by the gather and the epilogue of lattice_apply_kernel: thread `tid`
  // handles interior entries idx = tid + it*BLOCK, it < ITER, of the workgroup's slots.
  template <int P, int B, int THREADS = 256>
  struct InteriorMap
  {
    using G                   = Geo<P, B, THREADS>;
    static constexpr int NI_  = G::NI > 0 ? G::NI : 1;
    static constexpr int NIN_ = G::N_INT > 0 ? G::N_INT : 1;
    static constexpr int ITER = (G::SPW * NIN_ + G::ABLOCK - 1) / G::ABLOCK;
    __device__ static __forceinline__ void
    decode(int idx, int nslots, bool &ok, int &sl, int &i, int &lds)
    {
      ok = idx < nslots * NIN_;
      sl = (G::SPW == 1 || !ok) ? 0 : idx / NIN_;
      i  = G::SPW == 1 ? (ok ? idx : 0) : (ok ? idx % NIN_ : 0);
      const int x = i % NI_, y = (i / NI_) % NI_, z = i / (NI_ * NI_);
      lds = sl * G::N3 + ((z + 1) * G::N + (y + 1)) * G::N + x + 1;
    }
  };

  // K1: the level operator on the slots of one group.  All global loads of a phase are issued before the
  // first dependent use (fully unrolled, branch-free clamped addresses): a rolled loop keeps ONE load per
  // thread in flight and makes every phase latency-bound (measured 0.9 TB/s -> 3+ TB/s for the gather).
  // K1: the level operator on the slots of one group.  All global loads of a phase are issued before the
  // first dependent use (fully unrolled, branch-free clamped addresses): a rolled loop keeps ONE load per
  // thread in flight and makes every phase latency-bound (measured 0.9 TB/s -> 3+ TB/s for the gather).
  // The epilogue operands (x_old, b, D^-1) are requested BEFORE the sweeps so that their latency hides behind
  // the arithmetic; the gathered x is kept in registers for the Chebyshev update instead of being re-read.
#ifdef MGAMD_KERNEL_DEBUG
#define MGAMD_STAMP(k)                         \
  if (args.stamps && tid == 0)                 \
    args.stamps[(size_t)block * 8 + (k)] = wall_clock64();
#define MGAMD_ABLATED(bit) (args.ablate & (bit))
#else
#define MGAMD_STAMP(k)
#define MGAMD_ABLATED(bit) false
#endif

  // waves per SIMD: 2 for the 17^3 lattices (<= 256 VGPRs); 6 for single-cell slots (<= 80 VGPRs, measured 5 % faster at
  // p = 4 than unconstrained with 110 VGPRs)
  // the work of workgroup `block` of `nblocks` on the slots of args.g (kernels below)
  // WAVE (single-cell slots only): the slots are shared by ONE WAVEFRONT instead of a workgroup -- `block` / `nblocks` then count
  // wavefronts, smem_raw is the wavefront's own region, and every barrier below is a compiler-level fence (slot_sync): the nine
  // dependent phases of a hanging cell (three interpolation passes, three sweeps, three transposed passes) cost an LDS round
  // trip each instead of a workgroup barrier with the slowest of four waves (measured with tools/stamps.py on the
  // workgroup-scoped kernel: 5-7 of the 9 us a workgroup lives are spent between those barriers).
  template <typename T, int P, int B, int MODE, bool CONSTR = false, bool WAVE = false>
  __device__ __forceinline__ void
  lattice_apply_body(const ApplyArgs<T, P> &args, const uint32_t block, const uint32_t nblocks, unsigned char *smem_raw)
  {
    static_assert(!WAVE || (B == 1 && !CONSTR), "wave-scoped slots: single cells");
    using G  = Geo<P, B, WAVE ? 64 : 256>;
    using IM = InteriorMap<P, B, WAVE ? 64 : 256>;
    T *bufA = reinterpret_cast<T *>(smem_raw);
    T *bufB = bufA + G::SPW * G::N3;

    constexpr int BLOCK = G::ABLOCK;
    constexpr int ITER  = IM::ITER;
    constexpr int ITERS = (G::SPW * G::N_SHELL + BLOCK - 1) / BLOCK;

    const int tid    = WAVE ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;
    const int slot0  = (int)(WAVE ? block : xcd_contiguous(block, nblocks)) * G::SPW; // (WAVE: the caller has mapped the wavefront)
    const int nslots = min((int)G::SPW, (int)args.g.n_slots - slot0);
    MGAMD_STAMP(0)
    // per-slot scalars of this thread's line (hanging-node mask, constraint mask, cell size): requested with the gather, not
    // between the barriers that follow it (each was a full memory round trip on the critical path of the small-slot kernels)
    const int  sl = tid / G::LINES, ln = tid % G::LINES;
    const int  u = ln % G::N, v = ln / G::N;
    const bool act = tid < G::SPW * G::LINES && sl < nslots;
    uint32_t   mask = 0;
    if (B == 1 && act)
      mask = args.g.mask[slot0 + sl];
    uint32_t fm_early = 0;
    if constexpr (brick_may_be_constrained(B, CONSTR))
      if (args.g.fmask != nullptr && tid < nslots)
        fm_early = args.g.fmask[slot0 + tid];
    const double h_mine = G::ROUNDS == 1 ? args.g.h[slot0 + (act ? sl : 0)] : 0.0;
    uint32_t     fm_line = 0; // constraint mask of this thread's line (one line per thread when ROUNDS == 1)
    if constexpr (brick_may_be_constrained(B, CONSTR) && G::ROUNDS == 1)
      if (args.g.fmask != nullptr && act)
        fm_line = args.g.fmask[slot0 + sl];

    // D^-1 of slot-interior DoFs is not read from memory: they only see this slot's cells, so their diagonal is the
    // closed tensor form  d = h (k_x m_y m_z + m_x k_y m_z + m_x m_y k_z)  of the assembled 1D diagonals (what
    // lattice_diag_kernel stores), which depends on the node TYPE per direction only (t = lattice coordinate mod P:
    // 0 = node shared by two cells, a = a-th interior node of a cell): a P^3 table of s = d/h and 1/s in LDS, and 1/h
    // per slot.  One vector word less per interior DoF and Chebyshev pass.
    // Used at P = 1 (one node type: the look-up is a broadcast, -11 % on the 17^3 kernel); at P = 4 the 64-entry look-up
    // per entry pushes the 17^3 kernel over its 256 VGPRs (measured 1113 -> 1829 us), so D^-1 is read from memory there.
    constexpr bool CLOSED_DINV = P == 1 || (P == 4 && B == 4);
    T *dtab = bufB + G::SPW * G::N3; // [P^3] s, [P^3] 1/s, [SPW] 1/h
    if (CLOSED_DINV && is_cheb(MODE) && G::N_INT > 0)
      {
        constexpr int P3 = P * P * P;
        for (int t = tid; t < P3; t += BLOCK)
          {
            const int tt[3] = {t % P, (t / P) % P, t / (P * P)};
            T         m[3], k[3];
#pragma unroll
            for (int d = 0; d < 3; ++d)
              {
                T dm = T(0), dk = T(0);
                if (tt[d] == 0)
                  { // last node of one cell + first node of the next
                    dm = T(args.m.M[P * (P + 1) + P]) + T(args.m.M[0]);
                    dk = T(args.m.K[P * (P + 1) + P]) + T(args.m.K[0]);
                  }
#pragma unroll
                for (int q = 1; q < P; ++q)
                  if (q == tt[d])
                    {
                      dm = T(args.m.M[q * (P + 1) + q]);
                      dk = T(args.m.K[q * (P + 1) + q]);
                    }
                m[d] = dm;
                k[d] = dk;
              }
            const T sv   = k[0] * m[1] * m[2] + m[0] * k[1] * m[2] + m[0] * m[1] * k[2];
            dtab[t]      = sv;
            dtab[P3 + t] = T(1) / sv;
          }
        for (int t = tid; t < nslots; t += BLOCK)
          dtab[2 * P3 + t] = T(1) / T(args.g.h[slot0 + t]);
        slot_sync<WAVE>();
      }

    // ---- gather: addresses ----------------------------------------------------------------------------
    // interior entry `it` of this thread: global index (always a valid address) and LDS position (-1: no entry) with
    // the node type for the D^-1 table in bits 16+.  With one slot per workgroup the global index is base + entry
    // number and is not held in registers (the 17^3 kernels sit at the 256-VGPR limit).
    constexpr bool REMAT = G::SPW == 1;
    struct Ent
    {
      uint32_t g;
      int      l, t;
    };
    const uint32_t base0 = G::N_INT > 0 ? args.g.interior_base[slot0] : 0u;
    uint32_t       gbase[REMAT ? 1 : ITER];
    int            glds[ITER];
    if (G::N_INT > 0)
      {
#pragma unroll
        for (int it = 0; it < ITER; ++it)
          {
            bool ok;
            int  s2, i, lds;
            IM::decode(tid + it * BLOCK, nslots, ok, s2, i, lds);
            if (!REMAT)
              gbase[REMAT ? 0 : it] = args.g.interior_base[slot0 + s2] + (uint32_t)i;
            const int x = i % IM::NI_ + 1, y = (i / IM::NI_) % IM::NI_ + 1, z = i / (IM::NI_ * IM::NI_) + 1;
            const int t = (x % P) + P * ((y % P) + P * (z % P));
            glds[it]    = ok ? (lds | (t << 16) | (s2 << 24)) : -1;
          }
      }
    auto ent = [&](int it) -> Ent {
      const int      l = glds[it];
      const uint32_t g = REMAT ? base0 + (uint32_t)(l >= 0 ? tid + it * BLOCK : 0) : gbase[REMAT ? 0 : it];
      return Ent{g, l >= 0 ? (l & 0xFFFF) : -1, l >> 16};
    };
    uint32_t sgi[ITERS];
    int      spos[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
      {
        const int  idx = tid + it * BLOCK;
        const bool ok  = idx < nslots * G::N_SHELL;
        const int  s2 = ok ? idx / G::N_SHELL : 0, s = idx % G::N_SHELL;
        sgi[it]  = NT_LOAD(&args.g.shell_idx[(size_t)(slot0 + s2) * G::N_SHELL + (ok ? s : 0)]);
        spos[it] = s2 * G::N3 + (int)args.g.shell_pos[s];
        if (!ok)
          spos[it] = -1;
      }
    // ---- gather: values ---------------------------------------------------------------------------------
    constexpr bool x_from_b = MODE == MODE_CHEB_FIRST; // x = c0 dinv b, never stored
    T              xg[ITER]; // gathered interior values, kept for the Chebyshev epilogue
    if (G::N_INT > 0 && !MGAMD_ABLATED(8) && !x_from_b)
      {
#pragma unroll
        for (int it = 0; it < ITER; ++it)
          xg[it] = NT_LOAD(&args.src[ent(it).g]);
      }
    T sval[ITERS];
    if (!MGAMD_ABLATED(16))
      {
        if (x_from_b)
          {
            T sb[ITERS];
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
              {
                const uint32_t gi = sgi[it] < args.gather_limit ? sgi[it] : 0;
                sval[it]          = args.epi.dinv[gi];
                sb[it]            = args.epi.b[gi];
              }
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
              sval[it] = args.epi.c0 * sval[it] * sb[it];
          }
        else
          {
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
              sval[it] = args.src[sgi[it] < args.gather_limit ? sgi[it] : 0];
          }
      }
    // D^-1 of this thread's interior entry `it` (see above); looked up where needed, never held in registers
    auto interior_dinv = [&](int it) -> T {
      constexpr int P3 = P * P * P;
      const int     gl = glds[it] >= 0 ? glds[it] : 0, t = (gl >> 16) & 0xFF, s2 = (gl >> 24) & 0x7F;
      const T       rh = dtab[2 * P3 + s2];
      // |d| > 1e-10 ? 1/d : 1 with d = h s  (ref:include/operator.h:228-242)
      return fabs((double)dtab[t]) > 1.0e-10 * fabs((double)rh) ? rh * dtab[P3 + t] : T(1);
    };
    // ---- epilogue operands, requested now, consumed after the sweeps ----------------------------------------
    T xo[ITER], bv[ITER], dvm[CLOSED_DINV ? 1 : ITER]; // dvm: D^-1 from memory
    if (G::N_INT > 0 && !MGAMD_ABLATED(4))
      {
#pragma unroll
        for (int it = 0; it < ITER; ++it)
          {
            const uint32_t g = ent(it).g;
            xo[it] = bv[it] = T(0);
            if (MODE == MODE_RESIDUAL)
              bv[it] = NT_LOAD(&args.epi.b[g]);
            if (is_cheb(MODE))
              {
                if (MODE == MODE_CHEB && args.epi.xold)
                  xo[it] = NT_LOAD(&args.epi.xold[g]);
                bv[it] = NT_LOAD(&args.epi.b[g]);
                if (!CLOSED_DINV)
                  dvm[it] = NT_LOAD(&args.epi.dinv[g]);
              }
          }
        if (MODE == MODE_CHEB_FIRST)
          {
#pragma unroll
            for (int it = 0; it < ITER; ++it)
              xg[it] = args.epi.c0 * (CLOSED_DINV ? interior_dinv(it) : dvm[it]) * bv[it];
          }
      }
    // ---- gather: into LDS -------------------------------------------------------------------------------
    if (!MGAMD_ABLATED(16))
      {
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
          if (spos[it] >= 0)
            bufA[spos[it]] = sgi[it] < args.gather_limit ? sval[it] : T(0);
      }
    if (G::N_INT > 0 && !MGAMD_ABLATED(8))
      {
#pragma unroll
        for (int it = 0; it < ITER; ++it)
          {
            const int l = ent(it).l;
            if (l >= 0)
              bufA[l] = xg[it];
          }
      }
    slot_sync<WAVE>();
    MGAMD_STAMP(1)

    // ---- hanging-node interpolation (single-cell slots only) ---------------------------------------------
    bool any_hanging = false;
    if (B == 1)
      {
        any_hanging = (WAVE ? __any((int)(mask >> 3)) : __syncthreads_or((int)(mask >> 3))) != 0 && !MGAMD_ABLATED(32);
        if (any_hanging)
          hanging_passes<T, P, WAVE>(bufA, args.m, sl, u, v, act, mask, false);
      }

    // Which bricks may be constrained (must match LevelTables::build): families (B = 2) at every degree; larger bricks at
    // P = 1 only, in a slot group and kernel instantiation of their own (CONSTR).  Measured on MI355X: at p = 1 the 16^3-cell rim bricks of the octant replace the 2.5x slower single-cell
    // cluster path (V-cycle 1.95 -> 1.76 ms); at p = 4 the 4^3 rim bricks cost what the 2^3 families cost (0.5 vs 0.8 ns per
    // cell, eaten by the passes), while the extra SGPR pressure of the pass code slowed EVERY 17^3 workgroup by 4-6 %.
    if constexpr (brick_may_be_constrained(B, CONSTR))
      {
        // constrained bricks: whole-face / whole-edge hanging nodes (uniform branch: one mask per slot)
        any_hanging = __syncthreads_or((int)(fm_early != 0)) != 0;
        if (any_hanging)
          brick_constraint_passes<T, P, B, BLOCK>(bufA, args.m, tid, nslots, args.g.fmask + slot0, false, G::ROUNDS == 1 ? &fm_line : nullptr);
      }

    if (!MGAMD_ABLATED(1))
      {
        if constexpr (G::ROUNDS == 1)
          lattice_sweeps<T, P, B, BLOCK, NoHook, false, false, WAVE>(bufA, bufB, args.m, tid, nslots, &h_mine, NoHook(), true);
        else
          lattice_sweeps<T, P, B, BLOCK, NoHook, false, false, WAVE>(bufA, bufB, args.m, tid, nslots, args.g.h + slot0);
      }

    if (B == 1 && any_hanging)
      hanging_passes<T, P, WAVE>(bufA, args.m, sl, u, v, act, mask, true);
    if constexpr (brick_may_be_constrained(B, CONSTR))
      if (any_hanging)
        brick_constraint_passes<T, P, B, BLOCK>(bufA, args.m, tid, nslots, args.g.fmask + slot0, true, G::ROUNDS == 1 ? &fm_line : nullptr);
    MGAMD_STAMP(2)

    // ---- interior DoFs are complete: fused epilogue, contiguous stores -----------------------------------
    if (G::N_INT > 0 && !MGAMD_ABLATED(4))
      {
#pragma unroll
        for (int it = 0; it < ITER; ++it)
          if (const Ent e = ent(it); e.l >= 0)
            {
              const T ax = bufA[e.l];
              T       r;
              if (MODE == MODE_VMULT)
                r = ax;
              else if (MODE == MODE_RESIDUAL)
                r = bv[it] - ax;
              else
                {
                  const T dv  = CLOSED_DINV ? interior_dinv(it) : dvm[CLOSED_DINV ? 0 : it];
                  const T xov = MODE == MODE_CHEB_SECOND ? args.epi.c0 * dv * bv[it] : xo[it];
                  r           = xg[it] + args.epi.f1 * (xg[it] - xov) + args.epi.f2 * dv * (bv[it] - ax);
                }
              NT_STORE(r, &args.epi.out[e.g]);
            }
      }
    MGAMD_STAMP(3)
    // ---- shell DoFs: partial sums into the tail accumulator ------------------------------------------------
    if (!MGAMD_ABLATED(2))
      {
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
          if (spos[it] >= 0 && sgi[it] < args.scatter_limit)
            atomic_add(&args.tail_acc[sgi[it] - args.n_interior], bufA[spos[it]]);
      }
#ifdef MGAMD_KERNEL_DEBUG
    if (args.stamps)
      {
        __builtin_amdgcn_s_waitcnt(0); // drain this wave's memory operations before the final stamp
        slot_sync<WAVE>();
        MGAMD_STAMP(4)
      }
#endif
  }

  // K1p: the same operator application with PERSISTENT workgroups (one-slot-per-workgroup lattices: N^2 >= 256 lines).
  // Measured on MI355X (tools/stamps.py, octant p=4 L=8, 5-word Chebyshev pass): a workgroup of lattice_apply_body lives
  // 16.2 us per brick, 7.8 us of them in the gather (two DEPENDENT global round trips: slot tables -> values, at the
  // loaded-memory latency), 4.4 us in the sweeps, 3.7 us in epilogue + atomics; with 2 workgroups per CU (LDS) nothing hides
  // the gather, and removing 17 % of the kernel's HBM bytes (closed-form D^-1) changed nothing: the pass is bound by that
  // latency chain, not by bytes.  Here workgroup w walks the slots v = w, w + stride, ... (stride = number of resident
  // workgroups, a multiple of 8: xcd_contiguous keeps every workgroup inside the Morton range of its XCD) and runs a
  // software pipeline over them:
  //     top      values of slot v (requested one iteration earlier) -> LDS;  slot tables of v' = v + stride requested
  //     sweeps   (epilogue operands of v in flight, as before)
  //     after    the VALUES of v' are requested (the registers of the sweeps are free again), then epilogue + atomics of v
  // so both round trips of the next gather overlap with work of the current slot.  D^-1 of interior DoFs always in closed
  // form (see lattice_apply_body): that is what frees the registers for the second set of gathered values.
  //
  // FUSED LEVEL TRANSFERS (MODE_RESIDUAL_RESTRICT, MODE_CHEB_PROLONGATE; ref:multigrid_throughput.cc:1600-1604 between
  // Multigrid's residual / restriction and prolongation / post-smoothing steps).  The brick's 17-point lattice is the fine
  // patch of the (B/2)^3 coarse cells under it, so the three embedding sweeps run on the lattice the operator holds in LDS:
  //   RESTRICT    after the operator sweeps the lattice becomes  r = [interior: b - A x | shell: (owned ? b : 0) - this brick's
  //               partial sum of A x], three transposed embedding sweeps reduce it in place to the 9^3 coarse lattice, which is
  //               added to the coarse defect.  Restriction is linear: the partial sums of a shell row restricted by the bricks
  //               that produced them add up to the row's restricted residual.  Neither t nor the tail accumulator is written.
  //   PROLONGATE  before the operator sweeps the 9^3 coarse values are embedded on the scratch lattice and added to the gathered
  //               x (shell entries that an un-fused patch owns arrive corrected already); x + P x_c is stored once (interior in
  //               place, owned shell entries to a scratch vector that tail_kernel folds in) and never re-read by this pass.
  // Slots that are not flagged as fused (no brick patch, halo slots of a sharded level) take the base mode's path.
  template <typename T, int P, int B, int MODE_, bool CONSTR = false>
  __device__ __forceinline__ void
  lattice_apply_persistent_body(const ApplyArgs<T, P> &args, const uint32_t w, const uint32_t stride, unsigned char *smem_raw)
  {
    using G  = Geo<P, B>;
    using IM = InteriorMap<P, B>;
    static_assert(G::SPW == 1 && G::N_INT > 0, "persistent workgroups: one slot per workgroup");
    constexpr int  MODE   = base_mode(MODE_);
    constexpr bool FUSE_R = MODE_ == MODE_RESIDUAL_RESTRICT, FUSE_P = MODE_ == MODE_CHEB_PROLONGATE, FUSE = FUSE_R || FUSE_P;
    static_assert(!FUSE || (!CONSTR && B >= 2 && G::ABLOCK == 256), "fused transfers: plain bricks, 256 threads");
    constexpr int BC = B >= 2 ? B / 2 : 1, NC = P * BC + 1, NC3 = NC * NC * NC; // coarse lattice under the brick
    T *bufA = reinterpret_cast<T *>(smem_raw);
    T *bufB = bufA + G::N3;
    T *dtab = bufB + G::N3; // [P^3] s = d/h, [P^3] 1/s

    constexpr int BLOCK = G::ABLOCK;
    constexpr int ITER  = (G::N_INT + BLOCK - 1) / BLOCK;
    constexpr int ITERS = (G::N_SHELL + BLOCK - 1) / BLOCK;
    constexpr int ITC   = FUSE ? (NC3 + BLOCK - 1) / BLOCK : 1;
    constexpr int P3    = P * P * P;
    static_assert(!FUSE || 2 * ITERS <= 15, "two flag bits per shell entry in a 16-bit word");

    const int      tid = threadIdx.x;
    const uint32_t n   = args.g.n_slots;
    if (w >= n)
      return;

    if (is_cheb(MODE))
      {
        for (int t = tid; t < P3; t += BLOCK)
          {
            const int tt[3] = {t % P, (t / P) % P, t / (P * P)};
            T         m[3], k[3];
#pragma unroll
            for (int d = 0; d < 3; ++d)
              {
                T dm = T(0), dk = T(0);
                if (tt[d] == 0)
                  { // last node of one cell + first node of the next
                    dm = T(args.m.M[P * (P + 1) + P]) + T(args.m.M[0]);
                    dk = T(args.m.K[P * (P + 1) + P]) + T(args.m.K[0]);
                  }
#pragma unroll
                for (int q = 1; q < P; ++q)
                  if (q == tt[d])
                    {
                      dm = T(args.m.M[q * (P + 1) + q]);
                      dk = T(args.m.K[q * (P + 1) + q]);
                    }
                m[d] = dm;
                k[d] = dk;
              }
            const T sv   = k[0] * m[1] * m[2] + m[0] * k[1] * m[2] + m[0] * m[1] * k[2];
            dtab[t]      = sv;
            dtab[P3 + t] = T(1) / sv;
          }
      }
    // interior entry `it` of this thread is entry i = tid + it BLOCK of the slot (lattice order, NI^3 entries).  Its lattice
    // coordinates are NOT held in registers (14 VGPRs that the pipeline needs): every phase walks them from (x0, y0, z0) of
    // entry `tid` by the constant step BLOCK = DZ NI^2 + DY NI + DX with two carries (InteriorWalk).
    constexpr int NI = IM::NI_;
    struct InteriorWalk
    {
      int x, y, z; // 1-based lattice coordinates of the current entry
      __device__ __forceinline__ int
      pos() const
      {
        return (z * G::N + y) * G::N + x;
      }
      __device__ __forceinline__ int
      type() const
      {
        return (x % P) + P * ((y % P) + P * (z % P));
      }
      __device__ __forceinline__ void
      next()
      {
        constexpr int DZ = BLOCK / (NI * NI), DY = (BLOCK % (NI * NI)) / NI, DX = BLOCK % NI;
        x += DX;
        if (x > NI)
          {
            x -= NI;
            ++y;
          }
        y += DY;
        if (y > NI)
          {
            y -= NI;
            ++z;
          }
        z += DZ;
      }
    };
    const InteriorWalk walk0{tid % NI + 1, (tid / NI) % NI + 1, tid / (NI * NI) + 1};
    // entry `it` exists: always below the last round
    auto has_entry = [&](int it) -> bool { return (it + 1) * BLOCK <= G::N_INT || tid + it * BLOCK < G::N_INT; };
    int spos[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
      {
        const int idx = tid + it * BLOCK;
        spos[it]      = idx < G::N_SHELL ? (int)args.g.shell_pos[idx] : -1;
      }
    __syncthreads();

    constexpr bool x_from_b = MODE == MODE_CHEB_FIRST; // x = c0 dinv b, never stored
    // the epilogue operands (x_old, b) of a slot are requested BEFORE its sweeps (their latency hides behind the arithmetic);
    // they are live across the sweeps, which is why the sweeps are streamed cell by cell (line_stream: ~35 instead of ~70
    // doubles per thread).  Requested after the sweeps, or between the y and the x sweep, they fit whole-line sweeps but expose
    // their latency (measured: no gain over the one-workgroup-per-brick kernel).
    // slot tables of virtual block v
    auto load_tables = [&](uint32_t v, uint32_t &slot, uint32_t &base, uint32_t(&sg)[ITERS], double &h, uint32_t &fm, uint32_t &fl,
                           uint32_t(&cg)[ITC]) {
      slot = xcd_contiguous(v, n);
      base = args.g.interior_base[slot];
      h    = args.g.h[slot];
      fm   = 0;
      fl   = 0;
      if constexpr (FUSE)
        {
          fl = args.fused.flags[(size_t)slot * BLOCK + tid];
          const uint32_t *__restrict__ c0 = args.fused.coarse_idx + (size_t)slot * NC3;
#pragma unroll
          for (int it = 0; it < ITC; ++it)
            cg[it] = NT_LOAD(c0 + min(tid + it * BLOCK, NC3 - 1));
        }
      if constexpr (brick_may_be_constrained(B, CONSTR)) // the constraint mask travels with the tables (it was three exposed
        if (args.g.fmask != nullptr)                     // memory round trips per constrained brick when loaded where used)
          fm = args.g.fmask[slot];
      // (uniform base + round offset in scalar registers, ONE lane offset for every round: per-round lane offsets cost a VGPR
      // each, were spilled in the 5-word Chebyshev mode and reloaded here behind `s_waitcnt vmcnt(0)` - four exposed memory
      // round trips per slot)
      const uint32_t *__restrict__ p0 = args.g.shell_idx + (size_t)slot * G::N_SHELL;
#pragma unroll
      for (int it = 0; it < ITERS; ++it)
        {
          if ((it + 1) * BLOCK <= G::N_SHELL)
            sg[it] = NT_LOAD(p0 + it * BLOCK + tid);
          else
            sg[it] = NT_LOAD(p0 + min(tid + it * BLOCK, G::N_SHELL - 1));
        }
    };
    // operator input on the lattice of a slot (x_from_b: b and, on the shell, D^-1)
    // (FUSE_R: sbv = b on the shell entries this brick owns; FUSE_P: cv = the coarse values under the brick)
    auto load_values = [&](uint32_t base, const uint32_t(&sg)[ITERS], const uint32_t fl, const uint32_t(&cg)[ITC], T(&xv)[ITER], T(&sv)[ITERS],
                           T(&sbv)[ITERS], T(&cv)[ITC]) {
      const T *__restrict__ in = x_from_b ? args.epi.b : args.src;
#pragma unroll
      for (int it = 0; it < ITER; ++it)
        xv[it] = NT_LOAD(&in[base + (uint32_t)(has_entry(it) ? tid + it * BLOCK : 0)]);
#pragma unroll
      for (int it = 0; it < ITERS; ++it)
        {
          const uint32_t gi = sg[it] < args.gather_limit ? sg[it] : 0;
          sv[it]            = in[gi];
          if (x_from_b)
            sbv[it] = args.epi.dinv[gi];
          if (FUSE_R)
            sbv[it] = args.epi.b[((fl >> (2 * it)) & 1u) ? gi : 0];
        }
      // (the coarse values last: measured 3.80 vs 3.91 ms per octant p=4 post-smoothing against loading them first, which lets
      // the embedding start while x is still in flight but delays x behind 3 more loads)
      if constexpr (FUSE_P)
        {
#pragma unroll
          for (int it = 0; it < ITC; ++it)
            cv[it] = args.fused.coarse[cg[it] != DEV_INVALID ? cg[it] : 0];
        }
    };

    uint32_t slot, base, sgi[ITERS], fmcur, flcur, cgi[ITC];
    double   hcur;
    T        xg[ITER], sval[ITERS], sb[ITERS], cval[ITC];
    load_tables(w, slot, base, sgi, hcur, fmcur, flcur, cgi);
    load_values(base, sgi, flcur, cgi, xg, sval, sb, cval);

    for (uint32_t v = w;;)
      {
        const uint32_t block    = v;
        (void)block;
        const uint32_t vn       = v + stride;
        const bool     has_next = vn < n;
        MGAMD_STAMP(0)
        const T rh = T(1) / T(hcur);
        // D^-1 of this thread's interior entry `it`: |d| > 1e-10 ? 1/d : 1 with d = h s  (ref:include/operator.h:228-242)
        auto interior_dinv = [&](int t) -> T { // t: node type (InteriorWalk::type)
          return fabs((double)dtab[t]) > 1.0e-10 * fabs((double)rh) ? rh * dtab[P3 + t] : T(1);
        };
        // one flag word per thread, bit 15 equal in all of them: a workgroup-uniform branch
        const bool fused_slot = FUSE && ((__builtin_amdgcn_readfirstlane((int)flcur) >> 15) & 1);
        T          bv[ITER], xo[ITER];
        // ---- slot tables of the next slot, epilogue operands of this one: requested now ------------------------
        uint32_t slotn = slot, basen = base, sgn[ITERS], fmn = fmcur, fln = flcur, cgn[ITC];
        double   hn = hcur;
        if (has_next)
          load_tables(vn, slotn, basen, sgn, hn, fmn, fln, cgn);
        {
#pragma unroll
          for (int it = 0; it < ITER; ++it)
            {
              const uint32_t g = base + (uint32_t)(has_entry(it) ? tid + it * BLOCK : 0);
              xo[it]           = T(0);
              bv[it]           = x_from_b ? xg[it] : T(0); // (x_from_b: x itself is recomputed in the epilogue: one value less across the sweeps)
              if (MODE == MODE_CHEB && !FUSE_P && args.epi.xold) // (the fused prolongation pass has x_old = 0 by construction)
                xo[it] = NT_LOAD(&args.epi.xold[g]);
              if ((MODE == MODE_RESIDUAL || is_cheb(MODE)) && !x_from_b)
                bv[it] = NT_LOAD(&args.epi.b[g]);
            }
        }
        // ---- fused prolongation: x + P x_c on the lattice -----------------------------------------------------------
        if constexpr (FUSE_P)
          if (fused_slot)
            {
              // the coarse values, COMPACT (NC^3, x fastest) in the first lattice, which is free until this slot's values go
              // there: linear addresses (positions computed from the thread index were hoisted out of the loop and spilled).
              // z sweep from there into the scratch lattice, then y and x in place (a thread reads its line into registers
              // before it writes it back; lines of one sweep are disjoint)
#pragma unroll
              for (int it = 0; it < ITC; ++it)
                if (tid + it * BLOCK < NC3)
                  bufA[tid + it * BLOCK] = ((flcur >> (16 + it)) & 1u) ? T(0) : cval[it]; // (the flag, not the index: 3 registers)
              __syncthreads();
              T cin[NC], cout[G::N];
              for (int l = tid; l < NC * NC; l += BLOCK)
                {
                  const int b0 = (l / NC) * G::N + l % NC;
#pragma unroll
                  for (int i = 0; i < NC; ++i)
                    cin[i] = bufA[l + i * NC * NC];
                  line_embed_sym<T, P, BC>(args.fused.Eh, cin, cout);
#pragma unroll
                  for (int i = 0; i < G::N; ++i)
                    bufB[b0 + i * G::N * G::N] = cout[i];
                }
              __syncthreads();
              for (int l = tid; l < NC * G::N; l += BLOCK)
                {
                  const int b0 = (l / NC) * G::N * G::N + l % NC;
#pragma unroll
                  for (int i = 0; i < NC; ++i)
                    cin[i] = bufB[b0 + i * G::N];
                  line_embed_sym<T, P, BC>(args.fused.Eh, cin, cout);
#pragma unroll
                  for (int i = 0; i < G::N; ++i)
                    bufB[b0 + i * G::N] = cout[i];
                }
              __syncthreads();
              for (int l = tid; l < G::N * G::N; l += BLOCK)
                {
                  const int b0 = l * G::N;
#pragma unroll
                  for (int i = 0; i < NC; ++i)
                    cin[i] = bufB[b0 + i];
                  line_embed_sym<T, P, BC>(args.fused.Eh, cin, cout);
#pragma unroll
                  for (int i = 0; i < G::N; ++i)
                    bufB[b0 + i] = cout[i];
                }
              __syncthreads();
              // x <- x + P x_c: shell entries owned by an un-fused patch are corrected already; this brick's own ones go to the
              // scratch vector (tail_kernel folds them into x), the interior is stored in place
#pragma unroll
              for (int it = 0; it < ITERS; ++it)
                if (spos[it] >= 0 && sgi[it] < args.gather_limit)
                  {
                    const uint32_t f2 = (flcur >> (2 * it)) & 3u;
                    if (!(f2 & 2u))
                      sval[it] += bufB[spos[it]];
                    if (f2 & 1u)
                      args.fused.scratch[sgi[it]] = sval[it];
                  }
              InteriorWalk wk = walk0;
#pragma unroll
              for (int it = 0; it < ITER; ++it, wk.next())
                if (has_entry(it))
                  {
                    xg[it] += bufB[wk.pos()];
                    NT_STORE(xg[it], &args.fused.x_inout[base + (uint32_t)(tid + it * BLOCK)]);
                  }
            }
        // ---- values of this slot -> LDS ---------------------------------------------------------------------
        if (x_from_b)
          {
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
              sval[it] = args.epi.c0 * sb[it] * sval[it];
          }
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
          if (spos[it] >= 0)
            bufA[spos[it]] = sgi[it] < args.gather_limit ? sval[it] : T(0);
        {
          InteriorWalk wk = walk0;
#pragma unroll
          for (int it = 0; it < ITER; ++it, wk.next())
            if (has_entry(it))
              bufA[wk.pos()] = x_from_b ? args.epi.c0 * interior_dinv(wk.type()) * bv[it] : xg[it];
        }
        __syncthreads();
        MGAMD_STAMP(1)

        bool any_hanging = false;
        if constexpr (brick_may_be_constrained(B, CONSTR))
          {
            // constrained bricks: whole-face / whole-edge hanging nodes (uniform branch: one mask per slot)
            any_hanging = fmcur != 0;
            if (any_hanging)
              brick_constraint_passes<T, P, B, BLOCK>(bufA, args.m, tid, 1, args.g.fmask + slot, false, &fmcur);
          }
        // cell prefetch in the streamed sweeps (-4 % on the 2-4-word passes); the 5-word mode has no registers left for it
        lattice_sweeps<T, P, B, BLOCK, NoHook, true, MODE != MODE_CHEB>(bufA, bufB, args.m, tid, 1, &hcur);
        if constexpr (brick_may_be_constrained(B, CONSTR))
          if (any_hanging)
            brick_constraint_passes<T, P, B, BLOCK>(bufA, args.m, tid, 1, args.g.fmask + slot, true, &fmcur);
        MGAMD_STAMP(2)

        // ---- values of the next slot: requested now, consumed at the top of the next iteration ------------------
        T xgn[ITER], svaln[ITERS], sbn[ITERS], cvaln[ITC];
        if (has_next)
          load_values(basen, sgn, fln, cgn, xgn, svaln, sbn, cvaln);

        if (FUSE_R && fused_slot)
          {
            // ---- fused restriction: the lattice becomes this brick's part of b - A x, is reduced to the coarse lattice in
            // place (x^T, y^T, z^T) and added to the coarse defect
            if constexpr (FUSE_R)
              {
                InteriorWalk wk = walk0;
#pragma unroll
                for (int it = 0; it < ITER; ++it, wk.next())
                  if (has_entry(it))
                    bufA[wk.pos()] = bv[it] - bufA[wk.pos()];
#pragma unroll
                for (int it = 0; it < ITERS; ++it)
                  if (spos[it] >= 0)
                    bufA[spos[it]] = sgi[it] < args.scatter_limit ? (((flcur >> (2 * it)) & 1u) ? sb[it] : T(0)) - bufA[spos[it]] : T(0);
                __syncthreads();
                T rin[G::N], rout[NC];
                for (int l = tid; l < G::N * G::N; l += BLOCK)
                  {
                    const int b0 = l * G::N;
#pragma unroll
                    for (int i = 0; i < G::N; ++i)
                      rin[i] = bufA[b0 + i];
                    line_embed_sym_T<T, P, BC>(args.fused.Eh, rin, rout);
#pragma unroll
                    for (int i = 0; i < NC; ++i)
                      bufA[b0 + i] = rout[i];
                  }
                __syncthreads();
                for (int l = tid; l < NC * G::N; l += BLOCK)
                  {
                    const int b0 = (l / NC) * G::N * G::N + l % NC;
#pragma unroll
                    for (int i = 0; i < G::N; ++i)
                      rin[i] = bufA[b0 + i * G::N];
                    line_embed_sym_T<T, P, BC>(args.fused.Eh, rin, rout);
#pragma unroll
                    for (int i = 0; i < NC; ++i)
                      bufA[b0 + i * G::N] = rout[i];
                  }
                __syncthreads();
                // (the last sweep leaves the coarse lattice COMPACT in the scratch lattice: linear addresses for the scatter)
                for (int l = tid; l < NC * NC; l += BLOCK)
                  {
                    const int b0 = (l / NC) * G::N + l % NC;
#pragma unroll
                    for (int i = 0; i < G::N; ++i)
                      rin[i] = bufA[b0 + i * G::N * G::N];
                    line_embed_sym_T<T, P, BC>(args.fused.Eh, rin, rout);
#pragma unroll
                    for (int i = 0; i < NC; ++i)
                      bufB[l + i * NC * NC] = rout[i];
                  }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < ITC; ++it)
                  if (tid + it * BLOCK < NC3 && cgi[it] != DEV_INVALID)
                    atomic_add(&args.fused.coarse[cgi[it]], bufB[tid + it * BLOCK]);
              }
          }
        else
          {
            // ---- interior DoFs are complete: fused epilogue, contiguous stores -----------------------------------
            InteriorWalk wk = walk0;
#pragma unroll
            for (int it = 0; it < ITER; ++it, wk.next())
              if (has_entry(it))
                {
                  const uint32_t g  = base + (uint32_t)(tid + it * BLOCK);
                  const T        ax = bufA[wk.pos()];
                  T              r;
                  if (MODE == MODE_VMULT)
                    r = ax;
                  else if (MODE == MODE_RESIDUAL)
                    r = bv[it] - ax;
                  else
                    {
                      const T dv  = interior_dinv(wk.type());
                      const T xov = MODE == MODE_CHEB_SECOND ? args.epi.c0 * dv * bv[it] : xo[it];
                      const T xv  = x_from_b ? args.epi.c0 * dv * bv[it] : xg[it];
                      if (FUSE_P)
                        r = xv + args.epi.f2 * dv * (bv[it] - ax);
                      else
                        r = xv + args.epi.f1 * (xv - xov) + args.epi.f2 * dv * (bv[it] - ax);
                    }
                  NT_STORE(r, &args.epi.out[g]);
                }
            MGAMD_STAMP(3)
            // ---- shell DoFs: partial sums into the tail accumulator ------------------------------------------------
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
              if (spos[it] >= 0 && sgi[it] < args.scatter_limit)
                atomic_add(&args.tail_acc[sgi[it] - args.n_interior], bufA[spos[it]]);
          }
        MGAMD_STAMP(4)
        if (!has_next)
          break;
        v    = vn;
        slot  = slotn;
        base  = basen;
        hcur  = hn;
        fmcur = fmn;
        flcur = fln;
#pragma unroll
        for (int it = 0; it < ITC; ++it)
          {
            cgi[it]  = cgn[it];
            cval[it] = cvaln[it];
          }
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
          {
            sgi[it]  = sgn[it];
            sval[it] = svaln[it];
            sb[it]   = sbn[it];
          }
#pragma unroll
        for (int it = 0; it < ITER; ++it)
          xg[it] = xgn[it];
        __syncthreads(); // every thread has read its results of this slot from bufA
      }
  }
#undef MGAMD_STAMP
#undef MGAMD_ABLATED

  template <typename T, int P, int B, int MODE, bool CONSTR = false>
  __global__ void
  __launch_bounds__((Geo<P, B>::ABLOCK), (Geo<P, B>::ROUNDS > 1 ? 2 : (B == 1 ? 6 : (B == 2 ? 4 : 1)))) lattice_apply_kernel(const ApplyArgs<T, P> args)
  {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    lattice_apply_body<T, P, B, MODE, CONSTR>(args, blockIdx.x, gridDim.x, smem_raw);
  }

  // single cells, WAVE-SCOPED: a 256-thread workgroup = four wavefronts with their own cells (64 / (p+1)^2 cells each) and their
  // own LDS regions; no workgroup barrier anywhere (lattice_apply_body, WAVE)
  constexpr int CELL_WAVES = 4;
  template <typename T, int P>
  constexpr size_t
  cell_wave_lds()
  {
    using G = Geo<P, 1, 64>;
    return (((2 * (size_t)G::SPW * G::N3 + 2 * P * P * P + G::SPW) * sizeof(T) + 15) / 16) * 16;
  }
  template <typename T, int P, int MODE>
  __device__ __forceinline__ void
  cell_waves_body(const ApplyArgs<T, P> &args, const uint32_t block, const uint32_t nblocks, unsigned char *smem_raw)
  {
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t n_w  = (args.g.n_slots + Geo<P, 1, 64>::SPW - 1) / Geo<P, 1, 64>::SPW; // wavefronts with work
    // workgroups in XCD-contiguous (Morton) ranges like every other kernel, the four wavefronts of one on neighbouring cells
    const uint32_t vb = xcd_contiguous(block, nblocks) * CELL_WAVES + wave;
    if (vb < n_w)
      lattice_apply_body<T, P, 1, MODE, false, true>(args, vb, n_w, smem_raw + wave * cell_wave_lds<T, P>());
  }
  template <typename T, int P, int MODE>
  __global__ void
  __launch_bounds__(64 * CELL_WAVES, 6) cell_waves_kernel(const ApplyArgs<T, P> args)
  {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    cell_waves_body<T, P, MODE>(args, blockIdx.x, gridDim.x, smem_raw);
  }

  // persistent workgroups (lattice_apply_persistent_body); the grid is the number of RESIDENT workgroups (runtime.hip)
  template <typename T, int P, int B, int MODE, bool CONSTR = false>
  __global__ void
  __launch_bounds__((Geo<P, B>::ABLOCK), 2) lattice_apply_persistent_kernel(const ApplyArgs<T, P> args)
  {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    lattice_apply_persistent_body<T, P, B, MODE, CONSTR>(args, blockIdx.x, gridDim.x, smem_raw);
  }

  // The plain and the constrained bricks of one size in ONE launch (same lattice, same LDS, same block size): the
  // constrained group alone is a fraction of a round of workgroups on most levels.  Two inlined copies of the body: the plain
  // bricks keep the lean instruction stream (the embedding passes are only in the second copy).
  template <typename T, int P>
  struct BrickPairArgs
  {
    ApplyArgs<T, P> a;             // a.g = the plain group
    SlotGroupDev    g_constrained; // the constrained bricks of the same size
    uint32_t        n_wg_plain;
  };
  template <typename T, int P, int B, int MODE>
  __global__ void
  __launch_bounds__((Geo<P, B>::ABLOCK), (Geo<P, B>::ROUNDS > 1 ? 2 : 1)) lattice_apply_pair_kernel(const BrickPairArgs<T, P> args)
  {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    if (blockIdx.x < args.n_wg_plain)
      lattice_apply_body<T, P, B, MODE, false>(args.a, blockIdx.x, args.n_wg_plain, smem_raw);
    else
      {
        ApplyArgs<T, P> a = args.a;
        a.g               = args.g_constrained;
        a.stamps          = nullptr;
        lattice_apply_body<T, P, B, MODE, true>(a, blockIdx.x - args.n_wg_plain, gridDim.x - args.n_wg_plain, smem_raw);
      }
  }

  // (A 512-thread variant with half-line sweep tasks - 4 waves per SIMD, <= 128 VGPRs - was measured on MI355X: the sweeps alone
  // are 19 % faster (tools/sweep_probe.hip), the kernel is not: vmult 950 -> 964 us per pass, and the Chebyshev modes spill.)
  // the pair launch with persistent workgroups.  n_wg_plain > 0 (all slots resident at once): the first n_wg_plain workgroups
  // take one plain brick each, the others one constrained brick each; n_wg_plain == 0: see below
  template <typename T, int P, int B, int MODE>
  __global__ void
  __launch_bounds__((Geo<P, B>::ABLOCK), 2) lattice_apply_persistent_pair_kernel(const BrickPairArgs<T, P> args)
  {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    if (args.n_wg_plain == 0)
      {
        // more slots than resident workgroups: EVERY workgroup walks its share of the constrained bricks, then its share
        // of the plain ones (a static split of the workgroups between the two kinds is only balanced for one cost ratio:
        // measured 1.59 / 1.52 / 1.50 / 1.54 ms per octant p=1 V-cycle for assumed ratios 1.3 / 1.7 / 2.2 / 3.0)
        ApplyArgs<T, P> a = args.a;
        a.g               = args.g_constrained;
        a.stamps          = nullptr;
        lattice_apply_persistent_body<T, P, B, base_mode(MODE), true>(a, blockIdx.x, gridDim.x, smem_raw);
        __syncthreads(); // the lattice of the last constrained brick has been read by every thread
        lattice_apply_persistent_body<T, P, B, MODE, false>(args.a, blockIdx.x, gridDim.x, smem_raw);
      }
    else if (blockIdx.x < args.n_wg_plain)
      lattice_apply_persistent_body<T, P, B, MODE, false>(args.a, blockIdx.x, args.n_wg_plain, smem_raw);
    else
      {
        ApplyArgs<T, P> a = args.a;
        a.g               = args.g_constrained;
        a.stamps          = nullptr;
        lattice_apply_persistent_body<T, P, B, base_mode(MODE), true>(a, blockIdx.x - args.n_wg_plain, gridDim.x - args.n_wg_plain, smem_raw);
      }
  }

  // The 2^3 bricks and the single cells of a level in ONE launch (both have 256-thread workgroups and 20-35 KB of LDS):
  // on levels where each of them is a fraction of one round of workgroups, a launch costs a workgroup lifetime whatever
  // it does.  (Merging the 17^3 bricks in as well was measured slower: every workgroup then reserves their 78 KB.)
  template <typename T, int P>
  struct SmallSlotsArgs
  {
    ApplyArgs<T, P> a;       // a.g = the 2^3-brick group
    SlotGroupDev    g_cells; // the single-cell group
    uint32_t        n_wg_bricks;
  };
  template <typename T, int P, int MODE>
  __global__ void
  __launch_bounds__(256, 4) lattice_apply_small_kernel(const SmallSlotsArgs<T, P> args)
  {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    if (blockIdx.x < args.n_wg_bricks)
      lattice_apply_body<T, P, 2, MODE>(args.a, blockIdx.x, args.n_wg_bricks, smem_raw);
    else
      {
        ApplyArgs<T, P> a = args.a;
        a.g               = args.g_cells;
        a.stamps          = nullptr;
        cell_waves_body<T, P, MODE>(a, blockIdx.x - args.n_wg_bricks, gridDim.x - args.n_wg_bricks, smem_raw);
      }
  }

  // K1c: the level operator on SINGLE CELLS at p = 1, one cell per thread, 256 consecutive (Morton) cells per
  // workgroup (a CLUSTER).  The generic kernel spends one scattered load and one global atomic per (cell, node):
  // 8 of each per cell at p = 1, which bounds it at ~35 G atomics/s (measured: removing the atomics halves its time).
  // Here every distinct node of the cluster is loaded once into LDS, the cell operator runs in registers (2x2x2 lattice,
  // hanging-node interpolation included), results are pre-reduced with LDS atomics and every distinct node costs ONE
  // global atomic.  All nodes of such cells are tail DoFs, so the kernel is the same for every epilogue mode.
  struct CellClusterDev
  {
    const uint32_t *uniq_ptr; // [n_clusters + 1] into uniq_idx
    const uint32_t *uniq_idx; // global DoF index of every cluster-local node (ascending within a cluster)
    const uint16_t *loc;      // [n_slots * 8] cluster-local id of lattice node x + 2y + 4z; 0xFFFF = constrained (zero, no scatter)
    const uint16_t *mask;
    const double   *h;
    uint32_t        n_slots;
    uint32_t        max_uniq; // LDS: 2 * max_uniq values
  };
  constexpr int CLUSTER_CELLS = 256;
  constexpr int CLUSTER_ITERS = 8; // 256 cells x 8 nodes / 256 threads: the worst case, nothing shared

  template <typename T>
  struct ClusterArgs
  {
    CellClusterDev c;
    Mats<1>        m;
    const T       *src;
    T             *tail_acc;
    uint32_t       n_interior;
    // from_b = 1: the input is c0 dinv b (Epilogue::from_b), never stored
    const T *b, *dinv;
    T        c0;
    int      from_b;
    uint32_t cluster_offset; // first cluster of this launch (the halo / interior split of sharded levels)
  };

  template <typename T, bool TRANSPOSE>
  __device__ __forceinline__ void
  hanging_in_registers_p1(T (&x)[8], const uint32_t mask, const Mats<1> &m)
  {
    const int  cx = mask & 1, cy = (mask >> 1) & 1, cz = (mask >> 2) & 1;
    const bool fx = (mask >> 3) & 1, fy = (mask >> 4) & 1, fz = (mask >> 5) & 1;
    const bool ex = (mask >> 6) & 1, ey = (mask >> 7) & 1, ez = (mask >> 8) & 1;
#pragma unroll
    for (int dd = 0; dd < 3; ++dd)
      {
        const int d = TRANSPOSE ? 2 - dd : dd;
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            {
              bool on;
              int  c;
              // same line classification as hanging_passes with P = 1
              if (d == 0)
                {
                  const bool ou = u == cy, ov = v == cz;
                  on            = (fy && ou) || (fz && ov) || (ex && ou && ov);
                  c             = cx;
                }
              else if (d == 1)
                {
                  const bool ou = u == cx, ov = v == cz;
                  on            = (fx && ou) || (fz && ov) || (ey && ou && ov);
                  c             = cy;
                }
              else
                {
                  const bool ou = u == cx, ov = v == cy;
                  on            = (fx && ou) || (fy && ov) || (ez && ou && ov);
                  c             = cz;
                }
              const int i0 = d == 0 ? (v * 2 + u) * 2 : (d == 1 ? v * 4 + u : v * 2 + u);
              const int i1 = i0 + (d == 0 ? 1 : (d == 1 ? 2 : 4));
              if (on)
                {
                  const double *w  = c ? m.I1 : m.I0;
                  const T       a0 = x[i0], a1 = x[i1];
                  if (TRANSPOSE)
                    {
                      x[i0] = T(w[0]) * a0 + T(w[2]) * a1;
                      x[i1] = T(w[1]) * a0 + T(w[3]) * a1;
                    }
                  else
                    {
                      x[i0] = T(w[0]) * a0 + T(w[1]) * a1;
                      x[i1] = T(w[2]) * a0 + T(w[3]) * a1;
                    }
                }
            }
      }
  }

  template <typename T>
  __device__ __forceinline__ void
  cell_cluster_body(const ClusterArgs<T> &a, const uint32_t block, const uint32_t nblocks, unsigned char *smem_raw)
  {
    T *U   = reinterpret_cast<T *>(smem_raw);
    T *Acc = U + a.c.max_uniq;

    const int      tid  = threadIdx.x;
    const uint32_t cl   = a.cluster_offset + xcd_contiguous(block, nblocks);
    const uint32_t slot = cl * CLUSTER_CELLS + tid;
    const bool     act  = slot < a.c.n_slots;
    const uint32_t p0   = a.c.uniq_ptr[cl];
    const int      nu   = (int)(a.c.uniq_ptr[cl + 1] - p0);

    // the cell's own table entries: requested first, consumed after the barrier
    const uint4    lw   = reinterpret_cast<const uint4 *>(a.c.loc)[act ? slot : 0];
    const uint32_t mask = act ? a.c.mask[slot] : 0u;
    const T        h    = act ? T(a.c.h[slot]) : T(0);

    // distinct nodes of the cluster -> LDS (all loads in flight before the first use)
    uint32_t gi[CLUSTER_ITERS];
    T        gv[CLUSTER_ITERS];
#pragma unroll
    for (int k = 0; k < CLUSTER_ITERS; ++k)
      {
        const int j = tid + k * CLUSTER_CELLS;
        gi[k]       = nu > 0 ? a.c.uniq_idx[p0 + (j < nu ? j : nu - 1)] : a.n_interior;
      }
    if (a.from_b == 1)
      {
        T gb[CLUSTER_ITERS];
#pragma unroll
        for (int k = 0; k < CLUSTER_ITERS; ++k)
          {
            gv[k] = a.dinv[gi[k]];
            gb[k] = a.b[gi[k]];
          }
#pragma unroll
        for (int k = 0; k < CLUSTER_ITERS; ++k)
          gv[k] = a.c0 * gv[k] * gb[k];
      }
    else
      {
#pragma unroll
        for (int k = 0; k < CLUSTER_ITERS; ++k)
          gv[k] = a.src[gi[k]];
      }
#pragma unroll
    for (int k = 0; k < CLUSTER_ITERS; ++k)
      {
        const int j = tid + k * CLUSTER_CELLS;
        if (j < nu)
          {
            U[j]   = gv[k];
            Acc[j] = T(0);
          }
      }
    __syncthreads();

    uint32_t      l[8];
    const uint32_t lw4[4] = {lw.x, lw.y, lw.z, lw.w};
    T             x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      {
        l[i] = (lw4[i / 2] >> (16 * (i % 2))) & 0xFFFFu;
        x[i] = (act && l[i] != 0xFFFFu) ? U[l[i]] : T(0);
      }
    if (mask >> 3)
      hanging_in_registers_p1<T, false>(x, mask, a.m);
    // three sweeps of the 2x2x2 lattice, as in lattice_sweeps
    const T M0 = T(a.m.M[0]), M1 = T(a.m.M[1]), M2 = T(a.m.M[2]), M3 = T(a.m.M[3]);
    const T K0 = T(a.m.K[0]), K1 = T(a.m.K[1]), K2 = T(a.m.K[2]), K3 = T(a.m.K[3]);
    T       A[8], Bv[8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      { // z lines: nodes q, q + 4
        const T r0 = x[q], r1 = x[q + 4];
        A[q]       = M0 * r0 + M1 * r1;
        A[q + 4]   = M2 * r0 + M3 * r1;
        Bv[q]      = K0 * r0 + K1 * r1;
        Bv[q + 4]  = K2 * r0 + K3 * r1;
      }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      { // y lines: nodes i0, i0 + 2 with i0 = x + 4z
        const int i0 = (q & 1) + 4 * (q >> 1);
        const T   a0 = A[i0], a1 = A[i0 + 2], b0 = Bv[i0], b1 = Bv[i0 + 2];
        A[i0]        = M0 * a0 + M1 * a1;
        A[i0 + 2]    = M2 * a0 + M3 * a1;
        Bv[i0]       = K0 * a0 + K1 * a1 + M0 * b0 + M1 * b1;
        Bv[i0 + 2]   = K2 * a0 + K3 * a1 + M2 * b0 + M3 * b1;
      }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      { // x lines: nodes 2q, 2q + 1
        const T a0 = A[2 * q], a1 = A[2 * q + 1], b0 = Bv[2 * q], b1 = Bv[2 * q + 1];
        x[2 * q]     = h * (K0 * a0 + K1 * a1 + M0 * b0 + M1 * b1);
        x[2 * q + 1] = h * (K2 * a0 + K3 * a1 + M2 * b0 + M3 * b1);
      }
    if (mask >> 3)
      hanging_in_registers_p1<T, true>(x, mask, a.m);
    if (act)
      {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (l[i] != 0xFFFFu)
            atomic_add(&Acc[l[i]], x[i]);
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CLUSTER_ITERS; ++k)
      {
        const int j = tid + k * CLUSTER_CELLS;
        if (j < nu)
          atomic_add(&a.tail_acc[gi[k] - a.n_interior], Acc[j]);
      }
  }

  template <typename T>
  __global__ void
  __launch_bounds__(CLUSTER_CELLS) cell_cluster_apply_kernel(const ClusterArgs<T> a)
  {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    cell_cluster_body<T>(a, blockIdx.x, gridDim.x, smem_raw);
  }

  // p = 1: the 8^3 bricks and the cell clusters of a level in one launch (same reason as lattice_apply_small_kernel)
  template <typename T>
  struct P1SmallArgs
  {
    ApplyArgs<T, 1> a; // a.g = the 8^3-brick group
    ClusterArgs<T>  c;
    uint32_t        n_wg_bricks;
  };
  template <typename T, int MODE>
  __global__ void
  __launch_bounds__(256) lattice_cluster_kernel(const P1SmallArgs<T> args)
  {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    if (blockIdx.x < args.n_wg_bricks)
      lattice_apply_body<T, 1, 8, MODE>(args.a, blockIdx.x, args.n_wg_bricks, smem_raw);
    else
      cell_cluster_body<T>(args.c, blockIdx.x - args.n_wg_bricks, gridDim.x - args.n_wg_bricks, smem_raw);
  }

  // Diagonal of C^T K C.  Slots without hanging nodes: closed tensor form; single cells with hanging
  // faces/edges: one unit vector per local node through interpolation, sweeps and transpose.
  template <typename T, int P, int B, bool CONSTR = false>
  __global__ void
  __launch_bounds__((Geo<P, B>::BLOCK)) lattice_diag_kernel(const ApplyArgs<T, P> args)
  {
    using G = Geo<P, B>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *bufA = reinterpret_cast<T *>(smem_raw);
    T *bufB = bufA + G::SPW * G::N3;
    T *bufD = bufB + G::SPW * G::N3;

    const int tid    = threadIdx.x;
    const int slot0  = blockIdx.x * G::SPW;
    const int nslots = min((int)G::SPW, (int)args.g.n_slots - slot0);
    const int sl     = tid / G::LINES;
    const int ln     = tid % G::LINES;
    const int u = ln % G::N, v = ln / G::N;
    const bool act = tid < G::SPW * G::LINES && sl < nslots;

    uint32_t mask = 0;
    T        h    = T(0);
    if (act)
      {
        h = T(args.g.h[slot0 + sl]);
        if (B == 1)
          mask = args.g.mask[slot0 + sl];
      }
    // closed form: thread (u,v) = (x,y) fills its z column
    if (act)
      {
        T dM[G::N], dK[G::N];
#pragma unroll
        for (int i = 0; i < G::N; ++i)
          dM[i] = dK[i] = T(0);
#pragma unroll
        for (int c = 0; c < B; ++c)
#pragma unroll
          for (int a = 0; a <= P; ++a)
            {
              dM[c * P + a] += T(args.m.M[a * (P + 1) + a]);
              dK[c * P + a] += T(args.m.K[a * (P + 1) + a]);
            }
        T mx = T(0), kx = T(0), my = T(0), ky = T(0);
#pragma unroll
        for (int i = 0; i < G::N; ++i)
          {
            if (i == u)
              {
                mx = dM[i];
                kx = dK[i];
              }
            if (i == v)
              {
                my = dM[i];
                ky = dK[i];
              }
          }
#pragma unroll
        for (int i = 0; i < G::N; ++i)
          bufD[sl * G::N3 + (i * G::N + v) * G::N + u] = h * (kx * my * dM[i] + mx * ky * dM[i] + mx * my * dK[i]);
      }
    __syncthreads();
    if (B == 1)
      {
        const bool any_hanging = __syncthreads_or((int)(mask >> 3)) != 0;
        if (any_hanging)
          for (int j = 0; j < G::N3; ++j)
            {
              // e_j on every slot of this workgroup
              for (int idx = tid; idx < G::SPW * G::N3; idx += G::BLOCK)
                bufA[idx] = (idx % G::N3) == j ? T(1) : T(0);
              __syncthreads();
              hanging_passes<T, P>(bufA, args.m, sl, u, v, act, mask, false);
              lattice_sweeps<T, P, B, G::BLOCK>(bufA, bufB, args.m, tid, nslots, args.g.h + slot0);
              hanging_passes<T, P>(bufA, args.m, sl, u, v, act, mask, true);
              if (act && ln == 0 && (mask >> 3))
                bufD[sl * G::N3 + j] = bufA[sl * G::N3 + j];
              __syncthreads();
            }
      }
    if constexpr (brick_may_be_constrained(B, CONSTR))
      {
        // constrained bricks: the parent DoFs on hanging faces/edges need (C^T A C)_jj: one unit vector per shell
        // position through embedding, sweeps and transpose (the other shell entries reproduce the closed form)
        uint32_t fm = 0;
        if (args.g.fmask != nullptr && tid < nslots)
          fm = args.g.fmask[slot0 + tid];
        const bool any_family = __syncthreads_or((int)(fm != 0)) != 0;
        if (any_family)
          for (int s = 0; s < G::N_SHELL; ++s)
            {
              const int j = args.g.shell_pos[s];
              for (int idx = tid; idx < G::SPW * G::N3; idx += G::BLOCK)
                bufA[idx] = (idx % G::N3) == j ? T(1) : T(0);
              __syncthreads();
              brick_constraint_passes<T, P, B, G::BLOCK>(bufA, args.m, tid, nslots, args.g.fmask + slot0, false);
              lattice_sweeps<T, P, B, G::BLOCK>(bufA, bufB, args.m, tid, nslots, args.g.h + slot0);
              brick_constraint_passes<T, P, B, G::BLOCK>(bufA, args.m, tid, nslots, args.g.fmask + slot0, true);
              if (act && ln == 0 && args.g.fmask[slot0 + sl])
                bufD[sl * G::N3 + j] = bufA[sl * G::N3 + j];
              __syncthreads();
            }
      }
    if (G::N_INT > 0)
      for (int idx = tid; idx < nslots * G::N_INT; idx += G::BLOCK)
        {
          const int sl2 = idx / (G::N_INT > 0 ? G::N_INT : 1), i = idx % (G::N_INT > 0 ? G::N_INT : 1);
          const int x = i % (G::NI > 0 ? G::NI : 1), y = (i / (G::NI > 0 ? G::NI : 1)) % (G::NI > 0 ? G::NI : 1),
                    z = i / (G::NI > 0 ? G::NI * G::NI : 1);
          const T d = bufD[sl2 * G::N3 + ((z + 1) * G::N + (y + 1)) * G::N + x + 1];
          apply_epilogue<T, MODE_INVDIAG>(args.epi, args.g.interior_base[slot0 + sl2] + i, d);
        }
    for (int idx = tid; idx < nslots * G::N_SHELL; idx += G::BLOCK)
      {
        const int      sl2 = idx / G::N_SHELL, s = idx % G::N_SHELL;
        const uint32_t gi  = args.g.shell_idx[(size_t)(slot0 + sl2) * G::N_SHELL + s];
        if (gi < args.scatter_limit)
          atomic_add(&args.tail_acc[gi - args.n_interior], bufD[sl2 * G::N3 + args.g.shell_pos[s]]);
      }
  }

  // Epilogue for the tail (accumulated shell sums) and the constrained DoFs (identity rows:
  // ref:include/operator.h:170-172); re-zeroes the accumulator for the next application.
  template <typename T, int MODE_>
  __global__ void
  __launch_bounds__(256) tail_kernel(T *__restrict__ tail_acc, uint32_t n_interior, uint32_t n_tail, uint32_t n_rest, Epilogue<T> epi)
  {
    constexpr int  MODE   = base_mode(MODE_); // (MODE_RESIDUAL_RESTRICT: plain residual rows for the un-fused restriction)
    constexpr bool FUSE_P = MODE_ == MODE_CHEB_PROLONGATE;
    constexpr int  U      = 4;
    const uint32_t total  = n_tail + n_rest;
    const uint32_t stride = gridDim.x * blockDim.x;
    __shared__ T   dtable[256];
    const bool     coded = is_cheb(MODE) && epi.dinv_code != nullptr;
    if (coded)
      {
        dtable[threadIdx.x] = epi.dinv_table[threadIdx.x]; // 256 threads
        __syncthreads();
      }
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += U * stride)
      {
        T ax[U], xv[U], xo[U], bv[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          {
            const uint32_t i  = i0 + u * stride;
            const uint32_t gi = n_interior + i;
            ax[u] = xv[u] = xo[u] = bv[u] = dv[u] = T(0);
            if (i < total)
              {
                if (MODE != MODE_INVDIAG && MODE != MODE_CHEB_FIRST && (is_cheb(MODE) || i >= n_tail))
                  {
                    if (FUSE_P && i < n_tail && NT_LOAD(&epi.xs_flag[i]))
                      { // x + P x_c as the owning fused brick left it: folded into x here, after every brick has gathered x
                        xv[u] = NT_LOAD(&epi.xs[gi]);
                        epi.x_inout[gi] = xv[u];
                      }
                    else
                      xv[u] = NT_LOAD(&epi.x[gi]);
                  }
                ax[u] = i < n_tail ? tail_acc[i] : xv[u];
                if (MODE == MODE_RESIDUAL || is_cheb(MODE))
                  bv[u] = NT_LOAD(&epi.b[gi]);
                if (is_cheb(MODE))
                  {
                    if (MODE == MODE_CHEB && epi.xold)
                      xo[u] = NT_LOAD(&epi.xold[gi]);
                    if (coded)
                      {
                        const uint32_t c = NT_LOAD(&epi.dinv_code[i]);
                        dv[u]            = c != 255u ? dtable[c] : epi.dinv[gi];
                      }
                    else
                      dv[u] = epi.dinv[gi];
                  }
              }
          }
#pragma unroll
        for (int u = 0; u < U; ++u)
          {
            const uint32_t i  = i0 + u * stride;
            const uint32_t gi = n_interior + i;
            if (i < total)
              {
                if (i < n_tail)
                  NT_STORE(T(0), &tail_acc[i]);
                if (MODE == MODE_CHEB_FIRST || MODE == MODE_CHEB_SECOND)
                  {
                    const T x1 = epi.c0 * dv[u] * bv[u];
                    if (MODE == MODE_CHEB_FIRST)
                      {
                        xv[u] = x1;
                        if (i >= n_tail)
                          ax[u] = x1; // identity row
                      }
                    else
                      xo[u] = x1;
                  }
                if (MODE == MODE_VMULT)
                  NT_STORE(ax[u], &epi.out[gi]);
                else if (MODE == MODE_RESIDUAL)
                  NT_STORE(bv[u] - ax[u], &epi.out[gi]);
                else if (is_cheb(MODE))
                  NT_STORE(xv[u] + epi.f1 * (xv[u] - xo[u]) + epi.f2 * dv[u] * (bv[u] - ax[u]), &epi.out[gi]);
                else
                  epi.out[gi] = (i < n_tail && fabs((double)ax[u]) > 1.0e-10) ? T(1) / ax[u] : T(1);
              }
          }
      }
  }

  // ------------------------------------------------------------------------------------------
  // Two-level transfer.  One patch (coarse cell) per thread group; fine patch lattice NF^3.
  // ------------------------------------------------------------------------------------------
  template <int PC, int NF>
  struct TransferGeo
  {
    static constexpr int NC    = PC + 1;
    static constexpr int NC3   = NC * NC * NC;
    static constexpr int NF3   = NF * NF * NF;
    static constexpr int LINES = NF * NF;
    static constexpr int SPW   = LINES >= 256 ? 1 : 256 / LINES;
    static constexpr int BLOCK = ((SPW * LINES + 63) / 64) * 64;
  };

  template <typename T, int PC, int NF>
  struct TransferArgs
  {
    const uint32_t *coarse_idx;  // [n_patches][NC3]
    const uint16_t *coarse_mask; // [n_patches]
    const uint32_t *fine_idx;    // [n_patches][NF3]
    uint32_t        n_patches;
    Mats<PC>        m;              // only I0/I1 are used (coarse hanging nodes)
    double          E[NF * (PC + 1)]; // 1D embedding, rows = fine nodes
    const T        *src;
    T              *dst;
  };

  // dst[owned fine DoFs] += buf  (every fine DoF has exactly one owning patch: plain read-modify-write);
  // all index loads, then all value loads, then the stores
  template <typename T, int PC, int NF>
  __device__ __forceinline__ void
  fine_rmw(const TransferArgs<T, PC, NF> &args, const T *__restrict__ buf, int p0, int np, int tid)
  {
    using G           = TransferGeo<PC, NF>;
    constexpr int ITF = (G::SPW * G::NF3 + G::BLOCK - 1) / G::BLOCK;
    uint32_t      gi[ITF];
    T             val[ITF];
#pragma unroll
    for (int it = 0; it < ITF; ++it)
      {
        const int idx = tid + it * G::BLOCK;
        gi[it]        = idx < np * G::NF3 ? args.fine_idx[(size_t)p0 * G::NF3 + idx] : DEV_INVALID;
      }
#pragma unroll
    for (int it = 0; it < ITF; ++it)
      val[it] = args.dst[gi[it] != DEV_INVALID ? gi[it] : 0];
#pragma unroll
    for (int it = 0; it < ITF; ++it)
      if (gi[it] != DEV_INVALID)
        args.dst[gi[it]] = val[it] + buf[tid + it * G::BLOCK];
  }

  // x_f[owned fine DoFs] += E (x) E (x) E  (C_cell x_c)
  template <typename T, int PC, int NF, bool IDENTITY>
  __global__ void
  __launch_bounds__((TransferGeo<PC, NF>::BLOCK)) prolongate_kernel(const TransferArgs<T, PC, NF> args)
  {
    using G = TransferGeo<PC, NF>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *bufA = reinterpret_cast<T *>(smem_raw); // SPW * NF3
    T *bufB = bufA + G::SPW * G::NF3;

    const int tid    = threadIdx.x;
    const int p0     = blockIdx.x * G::SPW;
    const int np     = min((int)G::SPW, (int)args.n_patches - p0);
    const int sl     = tid / G::LINES;
    const int ln     = tid % G::LINES;
    const int u = ln % NF, v = ln / NF;
    const bool act = tid < G::SPW * G::LINES && sl < np;

    // gather coarse values into bufB (stride NF3 per patch); loads batched: indices, then values, then LDS
    {
      constexpr int ITC = (G::SPW * G::NC3 + G::BLOCK - 1) / G::BLOCK;
      uint32_t      gi[ITC];
      T             val[ITC];
#pragma unroll
      for (int it = 0; it < ITC; ++it)
        {
          const int idx = tid + it * G::BLOCK;
          gi[it]        = idx < np * G::NC3 ? args.coarse_idx[(size_t)p0 * G::NC3 + idx] : DEV_INVALID;
        }
#pragma unroll
      for (int it = 0; it < ITC; ++it)
        val[it] = args.src[gi[it] != DEV_INVALID ? gi[it] : 0];
#pragma unroll
      for (int it = 0; it < ITC; ++it)
        {
          const int idx = tid + it * G::BLOCK;
          if (idx < np * G::NC3)
            bufB[(idx / G::NC3) * G::NF3 + idx % G::NC3] = gi[it] != DEV_INVALID ? val[it] : T(0);
        }
    }
    __syncthreads();
    uint32_t mask = 0;
    if (act)
      mask = args.coarse_mask[p0 + sl];
    const bool any_hanging = __syncthreads_or((int)(mask >> 3)) != 0;
    if (any_hanging)
      {
        // hanging_passes expects patch stride NC3: run it on a view with that stride
        // (bufB patches are NF3 apart, so handle the offset by hand)
        T        *view = bufB + sl * (G::NF3 - G::NC3);
        const bool la  = act && u < G::NC && v < G::NC;
        hanging_passes<T, PC>(view, args.m, sl, u, v, la, la ? mask : 0u, false);
      }
    if (IDENTITY)
      {
        fine_rmw<T, PC, NF>(args, bufB, p0, np, tid);
        return;
      }
    // x: (NC,NC,NC) -> (NC,NC,NF); thread (u,v) = (y,z) < NC
    if (act && u < G::NC && v < G::NC)
      {
        T in[G::NC];
#pragma unroll
        for (int b = 0; b < G::NC; ++b)
          in[b] = bufB[sl * G::NF3 + (v * G::NC + u) * G::NC + b];
#pragma unroll
        for (int a = 0; a < NF; ++a)
          {
            T s = T(0);
#pragma unroll
            for (int b = 0; b < G::NC; ++b)
              s += T(args.E[a * G::NC + b]) * in[b];
            bufA[sl * G::NF3 + (v * G::NC + u) * NF + a] = s;
          }
      }
    __syncthreads();
    // y: (NC z, NC y, NF x) -> (NC z, NF y, NF x); thread (u,v) = (x < NF, z < NC)
    if (act && v < G::NC)
      {
        T in[G::NC];
#pragma unroll
        for (int b = 0; b < G::NC; ++b)
          in[b] = bufA[sl * G::NF3 + (v * G::NC + b) * NF + u];
#pragma unroll
        for (int a = 0; a < NF; ++a)
          {
            T s = T(0);
#pragma unroll
            for (int b = 0; b < G::NC; ++b)
              s += T(args.E[a * G::NC + b]) * in[b];
            bufB[sl * G::NF3 + (v * NF + a) * NF + u] = s;
          }
      }
    __syncthreads();
    // z: (NC z, NF, NF) -> (NF, NF, NF); thread (u,v) = (x,y) < NF; result straight to global
    if (act)
      {
        T in[G::NC];
#pragma unroll
        for (int b = 0; b < G::NC; ++b)
          in[b] = bufB[sl * G::NF3 + (b * NF + v) * NF + u];
#pragma unroll
        for (int a = 0; a < NF; ++a)
          {
            T s = T(0);
#pragma unroll
            for (int b = 0; b < G::NC; ++b)
              s += T(args.E[a * G::NC + b]) * in[b];
            bufA[sl * G::NF3 + (a * NF + v) * NF + u] = s;
          }
      }
    __syncthreads();
    fine_rmw<T, PC, NF>(args, bufA, p0, np, tid);
  }

  // d_c += C_cell^T (E (x) E (x) E)^T r_f[owned fine DoFs]
  template <typename T, int PC, int NF, bool IDENTITY>
  __global__ void
  __launch_bounds__((TransferGeo<PC, NF>::BLOCK)) restrict_kernel(const TransferArgs<T, PC, NF> args)
  {
    using G = TransferGeo<PC, NF>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *bufA = reinterpret_cast<T *>(smem_raw);
    T *bufB = bufA + G::SPW * G::NF3;

    const int tid    = threadIdx.x;
    const int p0     = blockIdx.x * G::SPW;
    const int np     = min((int)G::SPW, (int)args.n_patches - p0);
    const int sl     = tid / G::LINES;
    const int ln     = tid % G::LINES;
    const int u = ln % NF, v = ln / NF;
    const bool act = tid < G::SPW * G::LINES && sl < np;

    {
      constexpr int ITF = (G::SPW * G::NF3 + G::BLOCK - 1) / G::BLOCK;
      uint32_t      gi[ITF];
      T             val[ITF];
#pragma unroll
      for (int it = 0; it < ITF; ++it)
        {
          const int idx = tid + it * G::BLOCK;
          gi[it]        = idx < np * G::NF3 ? args.fine_idx[(size_t)p0 * G::NF3 + idx] : DEV_INVALID;
        }
#pragma unroll
      for (int it = 0; it < ITF; ++it)
        val[it] = args.src[gi[it] != DEV_INVALID ? gi[it] : 0];
#pragma unroll
      for (int it = 0; it < ITF; ++it)
        {
          const int idx = tid + it * G::BLOCK;
          if (idx < G::SPW * G::NF3)
            bufA[idx] = gi[it] != DEV_INVALID ? val[it] : T(0);
        }
    }
    __syncthreads();
    if (!IDENTITY)
      {
        // z^T: (NF,NF,NF) -> (NC z, NF, NF); thread (x,y) < NF
        if (act)
          {
            T in[NF];
#pragma unroll
            for (int a = 0; a < NF; ++a)
              in[a] = bufA[sl * G::NF3 + (a * NF + v) * NF + u];
#pragma unroll
            for (int b = 0; b < G::NC; ++b)
              {
                T s = T(0);
#pragma unroll
                for (int a = 0; a < NF; ++a)
                  s += T(args.E[a * G::NC + b]) * in[a];
                bufB[sl * G::NF3 + (b * NF + v) * NF + u] = s;
              }
          }
        __syncthreads();
        // y^T: (NC z, NF y, NF x) -> (NC z, NC y, NF x); thread (x < NF, z < NC)
        if (act && v < G::NC)
          {
            T in[NF];
#pragma unroll
            for (int a = 0; a < NF; ++a)
              in[a] = bufB[sl * G::NF3 + (v * NF + a) * NF + u];
#pragma unroll
            for (int b = 0; b < G::NC; ++b)
              {
                T s = T(0);
#pragma unroll
                for (int a = 0; a < NF; ++a)
                  s += T(args.E[a * G::NC + b]) * in[a];
                bufA[sl * G::NF3 + (v * G::NC + b) * NF + u] = s;
              }
          }
        __syncthreads();
        // x^T: (NC, NC, NF x) -> (NC,NC,NC) compact with stride NC3 inside the patch's bufB region
        if (act && u < G::NC && v < G::NC)
          {
            T in[NF];
#pragma unroll
            for (int a = 0; a < NF; ++a)
              in[a] = bufA[sl * G::NF3 + (v * G::NC + u) * NF + a];
#pragma unroll
            for (int b = 0; b < G::NC; ++b)
              {
                T s = T(0);
#pragma unroll
                for (int a = 0; a < NF; ++a)
                  s += T(args.E[a * G::NC + b]) * in[a];
                bufB[sl * G::NF3 + (v * G::NC + u) * G::NC + b] = s;
              }
          }
        __syncthreads();
      }
    T *res = IDENTITY ? bufA : bufB;
    uint32_t mask = 0;
    if (act)
      mask = args.coarse_mask[p0 + sl];
    const bool any_hanging = __syncthreads_or((int)(mask >> 3)) != 0;
    if (any_hanging)
      {
        T         *view = res + sl * (G::NF3 - G::NC3);
        const bool la   = act && u < G::NC && v < G::NC;
        hanging_passes<T, PC>(view, args.m, sl, u, v, la, la ? mask : 0u, true);
      }
    {
      constexpr int ITC = (G::SPW * G::NC3 + G::BLOCK - 1) / G::BLOCK;
      uint32_t      gi[ITC];
#pragma unroll
      for (int it = 0; it < ITC; ++it)
        {
          const int idx = tid + it * G::BLOCK;
          gi[it]        = idx < np * G::NC3 ? args.coarse_idx[(size_t)p0 * G::NC3 + idx] : DEV_INVALID;
        }
#pragma unroll
      for (int it = 0; it < ITC; ++it)
        if (gi[it] != DEV_INVALID)
          {
            const int idx = tid + it * G::BLOCK;
            atomic_add(&args.dst[gi[it]], res[(idx / G::NC3) * G::NF3 + idx % G::NC3]);
          }
    }
  }

  // ------------------------------------------------------------------------------------------
  // p = 1 h-patches (coarse cell -> its 8 children, 3^3 fine nodes) in registers: one patch per thread, 256 consecutive
  // patches per workgroup.  The coarse nodes of the workgroup's patches are deduplicated through LDS like the cell
  // clusters of K1c: one load (prolongation) or one global atomic (restriction) per distinct coarse node instead of
  // 8 per patch.  Index tables are stored transposed ([node][patch]) so that the threads of a wave read them coalesced.
  // ------------------------------------------------------------------------------------------
  template <typename T>
  struct PatchP1Args
  {
    const uint32_t *uniq_ptr;    // [n_workgroups + 1] into uniq_idx
    const uint32_t *uniq_idx;    // distinct coarse DoFs of a workgroup's patches
    const uint16_t *loc;         // [n_patches * 8] workgroup-local id of coarse node x + 2y + 4z; 0xFFFF = Dirichlet
    const uint16_t *coarse_mask; // [n_patches] hanging-node configuration of the coarse cell
    const uint32_t *fine_idx_t;  // [27][n_patches] owned fine DoF of fine node X + 3Y + 9Z, INVALID otherwise
    uint32_t        n_patches, max_uniq;
    Mats<1>         m; // only I0/I1 are used
    const T        *src;
    T              *dst;
  };
  constexpr int PATCH_P1_BLOCK = 256;
  constexpr int PATCH_P1_ITERS = 8;

  template <typename T>
  __global__ void
  __launch_bounds__(PATCH_P1_BLOCK) patch_p1_prolongate_kernel(const PatchP1Args<T> a)
  {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T             *U    = reinterpret_cast<T *>(smem_raw);
    const int      tid  = threadIdx.x;
    const uint32_t wg   = blockIdx.x;
    const uint32_t pch  = wg * PATCH_P1_BLOCK + tid;
    const bool     act  = pch < a.n_patches;
    const uint32_t p0   = a.uniq_ptr[wg];
    const int      nu   = (int)(a.uniq_ptr[wg + 1] - p0);
    const uint4    lw   = reinterpret_cast<const uint4 *>(a.loc)[act ? pch : 0];
    const uint32_t mask = act ? a.coarse_mask[pch] : 0u;
    uint32_t       fi[27];
#pragma unroll
    for (int t = 0; t < 27; ++t)
      fi[t] = act ? a.fine_idx_t[(size_t)t * a.n_patches + pch] : DEV_INVALID;
    T gv[PATCH_P1_ITERS];
#pragma unroll
    for (int k = 0; k < PATCH_P1_ITERS; ++k)
      {
        const int j = tid + k * PATCH_P1_BLOCK;
        gv[k]       = nu > 0 ? a.src[a.uniq_idx[p0 + (j < nu ? j : nu - 1)]] : T(0);
      }
#pragma unroll
    for (int k = 0; k < PATCH_P1_ITERS; ++k)
      {
        const int j = tid + k * PATCH_P1_BLOCK;
        if (j < nu)
          U[j] = gv[k];
      }
    __syncthreads();
    const uint32_t lw4[4] = {lw.x, lw.y, lw.z, lw.w};
    T              c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      {
        const uint32_t l = (lw4[i / 2] >> (16 * (i % 2))) & 0xFFFFu;
        c[i]             = (act && l != 0xFFFFu) ? U[l] : T(0);
      }
    if (mask >> 3)
      hanging_in_registers_p1<T, false>(c, mask, a.m);
    // embedding 2 -> 3 nodes per direction: (c0, (c0 + c1)/2, c1)
    T gx[12], gy[18];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      { // x: rows (y, z)
        gx[3 * q + 0] = c[2 * q];
        gx[3 * q + 1] = T(0.5) * (c[2 * q] + c[2 * q + 1]);
        gx[3 * q + 2] = c[2 * q + 1];
      }
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
      for (int X = 0; X < 3; ++X)
        { // y: gx index X + 3 (y + 2 z)
          const T y0 = gx[X + 3 * (0 + 2 * z)], y1 = gx[X + 3 * (1 + 2 * z)];
          gy[X + 3 * (0 + 3 * z)] = y0;
          gy[X + 3 * (1 + 3 * z)] = T(0.5) * (y0 + y1);
          gy[X + 3 * (2 + 3 * z)] = y1;
        }
    T oldv[27];
#pragma unroll
    for (int t = 0; t < 27; ++t)
      oldv[t] = a.dst[fi[t] != DEV_INVALID ? fi[t] : 0];
#pragma unroll
    for (int XY = 0; XY < 9; ++XY)
      { // z
        const T z0 = gy[XY], z1 = gy[XY + 9];
        const T f[3] = {z0, T(0.5) * (z0 + z1), z1};
#pragma unroll
        for (int Z = 0; Z < 3; ++Z)
          if (fi[XY + 9 * Z] != DEV_INVALID)
            a.dst[fi[XY + 9 * Z]] = oldv[XY + 9 * Z] + f[Z];
      }
  }

  template <typename T>
  __global__ void
  __launch_bounds__(PATCH_P1_BLOCK) patch_p1_restrict_kernel(const PatchP1Args<T> a)
  {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T             *Acc  = reinterpret_cast<T *>(smem_raw);
    const int      tid  = threadIdx.x;
    const uint32_t wg   = blockIdx.x;
    const uint32_t pch  = wg * PATCH_P1_BLOCK + tid;
    const bool     act  = pch < a.n_patches;
    const uint32_t p0   = a.uniq_ptr[wg];
    const int      nu   = (int)(a.uniq_ptr[wg + 1] - p0);
    const uint4    lw   = reinterpret_cast<const uint4 *>(a.loc)[act ? pch : 0];
    const uint32_t mask = act ? a.coarse_mask[pch] : 0u;
    uint32_t       fi[27];
#pragma unroll
    for (int t = 0; t < 27; ++t)
      fi[t] = act ? a.fine_idx_t[(size_t)t * a.n_patches + pch] : DEV_INVALID;
    T r[27];
#pragma unroll
    for (int t = 0; t < 27; ++t)
      r[t] = a.src[fi[t] != DEV_INVALID ? fi[t] : 0];
#pragma unroll
    for (int t = 0; t < 27; ++t)
      if (fi[t] == DEV_INVALID)
        r[t] = T(0);
    for (int j = tid; j < nu; j += PATCH_P1_BLOCK)
      Acc[j] = T(0);
    // transpose of the embedding, z then y then x: (f0 + f1/2, f1/2 + f2)
    T gy[18], gx[12], c[8];
#pragma unroll
    for (int XY = 0; XY < 9; ++XY)
      {
        gy[XY]     = r[XY] + T(0.5) * r[XY + 9];
        gy[XY + 9] = T(0.5) * r[XY + 9] + r[XY + 18];
      }
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
      for (int X = 0; X < 3; ++X)
        {
          const T y0 = gy[X + 3 * (0 + 3 * z)], y1 = gy[X + 3 * (1 + 3 * z)], y2 = gy[X + 3 * (2 + 3 * z)];
          gx[X + 3 * (0 + 2 * z)] = y0 + T(0.5) * y1;
          gx[X + 3 * (1 + 2 * z)] = T(0.5) * y1 + y2;
        }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      {
        c[2 * q]     = gx[3 * q] + T(0.5) * gx[3 * q + 1];
        c[2 * q + 1] = T(0.5) * gx[3 * q + 1] + gx[3 * q + 2];
      }
    if (mask >> 3)
      hanging_in_registers_p1<T, true>(c, mask, a.m);
    __syncthreads();
    const uint32_t lw4[4] = {lw.x, lw.y, lw.z, lw.w};
    if (act)
      {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          {
            const uint32_t l = (lw4[i / 2] >> (16 * (i % 2))) & 0xFFFFu;
            if (l != 0xFFFFu)
              atomic_add(&Acc[l], c[i]);
          }
      }
    __syncthreads();
    for (int j = tid; j < nu; j += PATCH_P1_BLOCK)
      atomic_add(&a.dst[a.uniq_idx[p0 + j]], Acc[j]);
  }

  // ------------------------------------------------------------------------------------------
  // Brick-level h-transfer: one fine brick (lattice NF = P*B+1) <-> the (B/2)^3 coarse cells under it
  // (lattice NC = P*B/2+1).  Interior fine DoFs are contiguous (coalesced read-modify-write); the shell
  // uses the brick's ownership list.  Restriction adds to the coarse vector with plain read-modify-write
  // for coarse lattice nodes strictly inside the patch (no other patch touches them) and atomics only on
  // its surface.
  // ------------------------------------------------------------------------------------------
  template <int P, int B>
  struct BrickTransferGeo
  {
    static constexpr int NF    = P * B + 1;
    static constexpr int BC    = B / 2;
    static constexpr int NC    = P * BC + 1;
    static constexpr int NF3   = NF * NF * NF;
    static constexpr int NC3   = NC * NC * NC;
    static constexpr int BLOCK = 256;
    // ONE lattice of NF^3 values: the coarse data sit at coordinates < NC and every sweep works IN PLACE (a thread reads
    // its whole line into registers before it writes it back, lines of one sweep are disjoint).  39 KB at NF = 17, four
    // workgroups per CU; with separate buffers per stage (77 KB, two per CU) the kernels were latency-bound at 1.9 TB/s.
    static constexpr int LDS = NF3;
  };

  template <typename T, int P>
  struct BrickTransferArgs
  {
    const uint32_t *slot;          // [n_bricks] fine slot index
    const uint32_t *interior_base; // fine group's table
    const uint16_t *shell_pos;     // fine group's table
    const uint32_t *coarse_idx;    // [n_bricks][NC3]
    const uint32_t *own_shell;     // [n_bricks][N_SHELL]
    uint32_t        n_bricks;
    double          E[(2 * P + 1) * (P + 1)];
    const T        *src;
    T              *dst;
  };

  template <typename T, int P, int B>
  __global__ void
  __launch_bounds__(256, 4) brick_prolongate_kernel(const BrickTransferArgs<T, P> args)
  {
    using G  = BrickTransferGeo<P, B>;
    using LG = Geo<P, B>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *buf = reinterpret_cast<T *>(smem_raw); // NF^3 lattice, index (z NF + y) NF + x
    constexpr int NC = G::NC, NF = G::NF, BLOCK = G::BLOCK;
    const int      tid   = threadIdx.x;
    const uint32_t brick = xcd_contiguous(blockIdx.x, gridDim.x);
    const uint32_t slot  = args.slot[brick];

    // operands of the final read-modify-write of dst: requested now, consumed after the sweeps (the barriers in between
    // wait on LDS traffic only, so these loads stay in flight under the embedding arithmetic)
    constexpr int  NI_P  = LG::NI > 0 ? LG::NI : 1;
    constexpr int  NIN_P = LG::N_INT > 0 ? LG::N_INT : 1;
    constexpr int  ITI_P = (NIN_P + BLOCK - 1) / BLOCK;
    constexpr int  ITS_P = (LG::N_SHELL + BLOCK - 1) / BLOCK;
    const uint32_t ibase = LG::N_INT > 0 ? args.interior_base[slot] : 0u;
    T              ival[ITI_P];
    uint32_t       sgi_p[ITS_P];
    T              sval_p[ITS_P];
    if (LG::N_INT > 0)
      {
#pragma unroll
        for (int it = 0; it < ITI_P; ++it)
          {
            const int i = tid + it * BLOCK;
            ival[it]    = NT_LOAD(&args.dst[ibase + (i < NIN_P ? i : 0)]);
          }
      }
#pragma unroll
    for (int it = 0; it < ITS_P; ++it)
      {
        const int s = tid + it * BLOCK;
        sgi_p[it]   = s < LG::N_SHELL ? NT_LOAD(&args.own_shell[(size_t)brick * LG::N_SHELL + s]) : DEV_INVALID;
      }
#pragma unroll
    for (int it = 0; it < ITS_P; ++it)
      sval_p[it] = args.dst[sgi_p[it] != DEV_INVALID ? sgi_p[it] : 0];

    {
      constexpr int ITC = (G::NC3 + BLOCK - 1) / BLOCK;
      uint32_t      gi[ITC];
      T             val[ITC];
#pragma unroll
      for (int it = 0; it < ITC; ++it)
        {
          const int idx = tid + it * BLOCK;
          gi[it]        = idx < G::NC3 ? NT_LOAD(&args.coarse_idx[(size_t)brick * G::NC3 + idx]) : DEV_INVALID;
        }
#pragma unroll
      for (int it = 0; it < ITC; ++it)
        val[it] = args.src[gi[it] != DEV_INVALID ? gi[it] : 0];
#pragma unroll
      for (int it = 0; it < ITC; ++it)
        {
          const int idx = tid + it * BLOCK;
          if (idx < G::NC3)
            {
              const int x = idx % NC, y = (idx / NC) % NC, z = idx / (NC * NC);
              buf[(z * NF + y) * NF + x] = gi[it] != DEV_INVALID ? val[it] : T(0);
            }
        }
    }
    __syncthreads();
    T in[NC], out[NF];
    // z: lines (x, y), x, y < NC
    for (int l = tid; l < NC * NC; l += BLOCK)
      {
        const int base = (l / NC) * NF + l % NC;
#pragma unroll
        for (int i = 0; i < NC; ++i)
          in[i] = buf[base + i * NF * NF];
        line_embed<T, P, G::BC>(args.E, in, out);
#pragma unroll
        for (int i = 0; i < NF; ++i)
          buf[base + i * NF * NF] = out[i];
      }
    __syncthreads();
    // y: lines (x, Z), x < NC
    for (int l = tid; l < NC * NF; l += BLOCK)
      {
        const int base = (l / NC) * NF * NF + l % NC;
#pragma unroll
        for (int i = 0; i < NC; ++i)
          in[i] = buf[base + i * NF];
        line_embed<T, P, G::BC>(args.E, in, out);
#pragma unroll
        for (int i = 0; i < NF; ++i)
          buf[base + i * NF] = out[i];
      }
    __syncthreads();
    // x: lines (Y, Z)
    for (int l = tid; l < NF * NF; l += BLOCK)
      {
        const int base = l * NF;
#pragma unroll
        for (int i = 0; i < NC; ++i)
          in[i] = buf[base + i];
        line_embed<T, P, G::BC>(args.E, in, out);
#pragma unroll
        for (int i = 0; i < NF; ++i)
          buf[base + i] = out[i];
      }
    __syncthreads();
    // dst += : interior contiguous, then the owned shell (old values already in registers)
    if (LG::N_INT > 0)
      {
#pragma unroll
        for (int it = 0; it < ITI_P; ++it)
          {
            const int i = tid + it * BLOCK;
            if (i < NIN_P)
              {
                const int x = i % NI_P, y = (i / NI_P) % NI_P, z = i / (NI_P * NI_P);
                NT_STORE(ival[it] + buf[((z + 1) * NF + (y + 1)) * NF + x + 1], &args.dst[ibase + i]);
              }
          }
      }
#pragma unroll
    for (int it = 0; it < ITS_P; ++it)
      if (sgi_p[it] != DEV_INVALID)
        args.dst[sgi_p[it]] = sval_p[it] + buf[args.shell_pos[tid + it * BLOCK]];
  }

  template <typename T, int P, int B>
  __global__ void
  __launch_bounds__(256) brick_restrict_kernel(const BrickTransferArgs<T, P> args)
  {
    using G  = BrickTransferGeo<P, B>;
    using LG = Geo<P, B>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *buf = reinterpret_cast<T *>(smem_raw); // NF^3 lattice, reduced in place to the coarse lattice at coordinates < NC
    constexpr int NC = G::NC, NF = G::NF, BLOCK = G::BLOCK;
    const int      tid   = threadIdx.x;
    const uint32_t brick = xcd_contiguous(blockIdx.x, gridDim.x);
    const uint32_t slot  = args.slot[brick];

    // gather the owned fine residuals (not owned / constrained: 0)
    if (LG::N_INT > 0)
      {
        constexpr int  NI_  = LG::NI > 0 ? LG::NI : 1;
        constexpr int  NIN_ = LG::N_INT > 0 ? LG::N_INT : 1;
        constexpr int  ITI  = (NIN_ + BLOCK - 1) / BLOCK;
        const uint32_t base = args.interior_base[slot];
        T              val[ITI];
#pragma unroll
        for (int it = 0; it < ITI; ++it)
          {
            const int i = tid + it * BLOCK;
            val[it]     = NT_LOAD(&args.src[base + (i < NIN_ ? i : 0)]);
          }
#pragma unroll
        for (int it = 0; it < ITI; ++it)
          {
            const int i = tid + it * BLOCK;
            if (i < NIN_)
              {
                const int x = i % NI_, y = (i / NI_) % NI_, z = i / (NI_ * NI_);
                buf[((z + 1) * NF + (y + 1)) * NF + x + 1] = val[it];
              }
          }
      }
    {
      constexpr int ITS = (LG::N_SHELL + BLOCK - 1) / BLOCK;
      uint32_t      gi[ITS];
      T             val[ITS];
#pragma unroll
      for (int it = 0; it < ITS; ++it)
        {
          const int s = tid + it * BLOCK;
          gi[it]      = s < LG::N_SHELL ? NT_LOAD(&args.own_shell[(size_t)brick * LG::N_SHELL + s]) : DEV_INVALID;
        }
#pragma unroll
      for (int it = 0; it < ITS; ++it)
        val[it] = args.src[gi[it] != DEV_INVALID ? gi[it] : 0];
#pragma unroll
      for (int it = 0; it < ITS; ++it)
        if (tid + it * BLOCK < LG::N_SHELL)
          buf[args.shell_pos[tid + it * BLOCK]] = gi[it] != DEV_INVALID ? val[it] : T(0);
    }
    // the coarse indices and the old values of the patch-interior coarse nodes (plain read-modify-write at the end):
    // requested before the sweeps
    constexpr int ITC_R = (G::NC3 + BLOCK - 1) / BLOCK;
    uint32_t      cgi[ITC_R];
    T             cold[ITC_R];
    bool          cinner[ITC_R];
#pragma unroll
    for (int it = 0; it < ITC_R; ++it)
      {
        const int idx = tid + it * BLOCK;
        cgi[it]       = idx < G::NC3 ? NT_LOAD(&args.coarse_idx[(size_t)brick * G::NC3 + idx]) : DEV_INVALID;
        const int x = idx % NC, y = (idx / NC) % NC, z = idx / (NC * NC);
        cinner[it] = x > 0 && y > 0 && z > 0 && x < NC - 1 && y < NC - 1 && z < NC - 1;
      }
#pragma unroll
    for (int it = 0; it < ITC_R; ++it)
      cold[it] = args.dst[(cgi[it] != DEV_INVALID && cinner[it]) ? cgi[it] : 0];
    __syncthreads();
    T in[NF], out[NC];
    // x^T: lines (Y, Z)
    for (int l = tid; l < NF * NF; l += BLOCK)
      {
        const int base = l * NF;
#pragma unroll
        for (int i = 0; i < NF; ++i)
          in[i] = buf[base + i];
        line_embed_T<T, P, G::BC>(args.E, in, out);
#pragma unroll
        for (int i = 0; i < NC; ++i)
          buf[base + i] = out[i];
      }
    __syncthreads();
    // y^T: lines (x, Z), x < NC
    for (int l = tid; l < NC * NF; l += BLOCK)
      {
        const int base = (l / NC) * NF * NF + l % NC;
#pragma unroll
        for (int i = 0; i < NF; ++i)
          in[i] = buf[base + i * NF];
        line_embed_T<T, P, G::BC>(args.E, in, out);
#pragma unroll
        for (int i = 0; i < NC; ++i)
          buf[base + i * NF] = out[i];
      }
    __syncthreads();
    // z^T: lines (x, y), x, y < NC
    for (int l = tid; l < NC * NC; l += BLOCK)
      {
        const int base = (l / NC) * NF + l % NC;
#pragma unroll
        for (int i = 0; i < NF; ++i)
          in[i] = buf[base + i * NF * NF];
        line_embed_T<T, P, G::BC>(args.E, in, out);
#pragma unroll
        for (int i = 0; i < NC; ++i)
          buf[base + i * NF * NF] = out[i];
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITC_R; ++it)
      if (cgi[it] != DEV_INVALID)
        {
          const int idx = tid + it * BLOCK;
          const int x = idx % NC, y = (idx / NC) % NC, z = idx / (NC * NC);
          const T   v = buf[(z * NF + y) * NF + x];
          if (cinner[it])
            args.dst[cgi[it]] = cold[it] + v; // only this patch touches coarse nodes strictly inside it
          else
            atomic_add(&args.dst[cgi[it]], v);
        }
  }

  // brick_restrict_kernel with PERSISTENT workgroups (17-point fine lattices): workgroup w restricts the bricks w, w + stride,
  // ... with the same software pipeline as lattice_apply_persistent_body: the tables of the next brick (slot -> interior base,
  // ownership list, coarse indices) are requested before the sweeps of the current one, its values (fine residuals, old
  // coarse values) after them.  The one-brick-per-workgroup kernel spends its life in three dependent round trips
  // (slot -> base -> values): measured 486 us for 1.6 GB at octant p=4 L=8.
  template <typename T, int P, int B>
  __global__ void
  __launch_bounds__(256, 3) brick_restrict_persistent_kernel(const BrickTransferArgs<T, P> args)
  {
    using G  = BrickTransferGeo<P, B>;
    using LG = Geo<P, B>;
    static_assert(LG::N_INT > 0, "bricks with interior nodes");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *buf = reinterpret_cast<T *>(smem_raw); // NF^3 lattice, reduced in place to the coarse lattice at coordinates < NC
    constexpr int NC = G::NC, NF = G::NF, BLOCK = G::BLOCK;
    constexpr int NI = LG::NI, NIN = LG::N_INT;
    constexpr int ITI = (NIN + BLOCK - 1) / BLOCK, ITS = (LG::N_SHELL + BLOCK - 1) / BLOCK, ITC = (G::NC3 + BLOCK - 1) / BLOCK;
    const int      tid = threadIdx.x;
    const uint32_t n = args.n_bricks, w = blockIdx.x, stride = gridDim.x;
    if (w >= n)
      return;
    // loop-invariant positions
    int  spos[ITS], cpos[ITC];
    bool cinner[ITC];
#pragma unroll
    for (int it = 0; it < ITS; ++it)
      spos[it] = tid + it * BLOCK < LG::N_SHELL ? (int)args.shell_pos[tid + it * BLOCK] : -1;
#pragma unroll
    for (int it = 0; it < ITC; ++it)
      {
        const int idx = tid + it * BLOCK;
        const int x = idx % NC, y = (idx / NC) % NC, z = idx / (NC * NC);
        cinner[it] = x > 0 && y > 0 && z > 0 && x < NC - 1 && y < NC - 1 && z < NC - 1;
        cpos[it]   = idx < G::NC3 ? (z * NF + y) * NF + x : -1;
      }
    auto load_tables = [&](uint32_t v, uint32_t &base, uint32_t(&gi)[ITS], uint32_t(&cgi)[ITC]) {
      const uint32_t brick = xcd_contiguous(v, n);
      base                 = args.interior_base[args.slot[brick]];
#pragma unroll
      for (int it = 0; it < ITS; ++it)
        gi[it] = spos[it] >= 0 ? NT_LOAD(&args.own_shell[(size_t)brick * LG::N_SHELL + tid + it * BLOCK]) : DEV_INVALID;
#pragma unroll
      for (int it = 0; it < ITC; ++it)
        cgi[it] = cpos[it] >= 0 ? NT_LOAD(&args.coarse_idx[(size_t)brick * G::NC3 + tid + it * BLOCK]) : DEV_INVALID;
    };
    auto load_values = [&](uint32_t base, const uint32_t(&gi)[ITS], const uint32_t(&cgi)[ITC], T(&val)[ITI], T(&sval)[ITS], T(&cold)[ITC]) {
#pragma unroll
      for (int it = 0; it < ITI; ++it)
        val[it] = NT_LOAD(&args.src[base + (uint32_t)(tid + it * BLOCK < NIN ? tid + it * BLOCK : 0)]);
#pragma unroll
      for (int it = 0; it < ITS; ++it)
        sval[it] = args.src[gi[it] != DEV_INVALID ? gi[it] : 0];
#pragma unroll
      for (int it = 0; it < ITC; ++it)
        cold[it] = args.dst[(cgi[it] != DEV_INVALID && cinner[it]) ? cgi[it] : 0];
    };
    uint32_t base, gi[ITS], cgi[ITC];
    T        val[ITI], sval[ITS], cold[ITC];
    load_tables(w, base, gi, cgi);
    load_values(base, gi, cgi, val, sval, cold);
    for (uint32_t v = w;;)
      {
        const uint32_t vn       = v + stride;
        const bool     has_next = vn < n;
        // fine residuals of this brick -> LDS (interior entry i = tid + it BLOCK walks the lattice as in InteriorWalk)
        {
          int x = tid % NI + 1, y = (tid / NI) % NI + 1, z = tid / (NI * NI) + 1;
#pragma unroll
          for (int it = 0; it < ITI; ++it)
            {
              if ((it + 1) * BLOCK <= NIN || tid + it * BLOCK < NIN)
                buf[(z * NF + y) * NF + x] = val[it];
              constexpr int DZ = BLOCK / (NI * NI), DY = (BLOCK % (NI * NI)) / NI, DX = BLOCK % NI;
              x += DX;
              if (x > NI)
                {
                  x -= NI;
                  ++y;
                }
              y += DY;
              if (y > NI)
                {
                  y -= NI;
                  ++z;
                }
              z += DZ;
            }
        }
#pragma unroll
        for (int it = 0; it < ITS; ++it)
          if (spos[it] >= 0)
            buf[spos[it]] = gi[it] != DEV_INVALID ? sval[it] : T(0);
        uint32_t basen = base, gin[ITS], cgn[ITC];
        if (has_next)
          load_tables(vn, basen, gin, cgn);
        __syncthreads();
        T in[NF], out[NC];
        // x^T: lines (Y, Z)
        for (int l = tid; l < NF * NF; l += BLOCK)
          {
            const int b0 = l * NF;
#pragma unroll
            for (int i = 0; i < NF; ++i)
              in[i] = buf[b0 + i];
            line_embed_T<T, P, G::BC>(args.E, in, out);
#pragma unroll
            for (int i = 0; i < NC; ++i)
              buf[b0 + i] = out[i];
          }
        __syncthreads();
        // y^T: lines (x, Z), x < NC
        for (int l = tid; l < NC * NF; l += BLOCK)
          {
            const int b0 = (l / NC) * NF * NF + l % NC;
#pragma unroll
            for (int i = 0; i < NF; ++i)
              in[i] = buf[b0 + i * NF];
            line_embed_T<T, P, G::BC>(args.E, in, out);
#pragma unroll
            for (int i = 0; i < NC; ++i)
              buf[b0 + i * NF] = out[i];
          }
        __syncthreads();
        // z^T: lines (x, y), x, y < NC
        for (int l = tid; l < NC * NC; l += BLOCK)
          {
            const int b0 = (l / NC) * NF + l % NC;
#pragma unroll
            for (int i = 0; i < NF; ++i)
              in[i] = buf[b0 + i * NF * NF];
            line_embed_T<T, P, G::BC>(args.E, in, out);
#pragma unroll
            for (int i = 0; i < NC; ++i)
              buf[b0 + i * NF * NF] = out[i];
          }
        __syncthreads();
        T valn[ITI], svaln[ITS], coldn[ITC];
        if (has_next)
          load_values(basen, gin, cgn, valn, svaln, coldn);
#pragma unroll
        for (int it = 0; it < ITC; ++it)
          if (cgi[it] != DEV_INVALID)
            {
              const T r = buf[cpos[it]];
              if (cinner[it])
                args.dst[cgi[it]] = cold[it] + r; // only this patch touches coarse nodes strictly inside it
              else
                atomic_add(&args.dst[cgi[it]], r);
            }
        if (!has_next)
          break;
        v    = vn;
        base = basen;
#pragma unroll
        for (int it = 0; it < ITS; ++it)
          {
            gi[it]   = gin[it];
            sval[it] = svaln[it];
          }
#pragma unroll
        for (int it = 0; it < ITC; ++it)
          {
            cgi[it]  = cgn[it];
            cold[it] = coldn[it];
          }
#pragma unroll
        for (int it = 0; it < ITI; ++it)
          val[it] = valn[it];
        __syncthreads(); // every thread has read its coarse results from buf
      }
  }

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
