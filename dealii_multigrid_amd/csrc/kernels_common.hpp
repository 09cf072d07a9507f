// Shared pieces of the HIP kernels (overview: kernels.hpp): non-temporal access macros, XCD-contiguous work mapping, the 1D
// reference matrices (Mats) and lattice geometry (Geo), argument structs, 1D line products (dense and even-odd), the three lattice
// sweeps (whole lines, streamed, segment tasks), hanging-node and brick-constraint passes, the 1D h-embedding.
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

  // S: the number type of the kernel that takes them as arguments (round 3: float kernels converted the double constants at every
  // use, which cost them the registers of the double kernels)
  template <int P, typename S = double>
  struct Mats
  {
    S M[(P + 1) * (P + 1)];
    S K[(P + 1) * (P + 1)];
    S I0[(P + 1) * (P + 1)];
    S I1[(P + 1) * (P + 1)];
    // even-odd decomposition of the centrosymmetric M and K (GLL nodes are symmetric): A x = Ae xe + Ao xo with
    // xe_j = x_j + x_{P-j}, xo_j = x_j - x_{P-j};  Ae_ij = (A_ij + A_i,P-j)/2 (middle column: A_i,mid), Ao_ij = (A_ij - A_i,P-j)/2
    static constexpr int NH = (P + 2) / 2, NO = (P + 1) / 2;
    S                    Me[NH * NH], Mo[NO * NO], Ke[NH * NH], Ko[NO * NO];
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
    // MODE_CHEB only, float levels under double outer vectors (MGNumberType float): the LAST post-smoothing pass of the finest
    // level stores its result as doubles here instead of `out` -- copy_from_mg (ref:multigrid_throughput.cc:1201-1203) without a
    // pass of its own over the 137 M-DoF vectors
    double *out_wide = nullptr;
  };
  // the store of a Chebyshev result (interior and tail epilogues)
  template <typename T>
  __device__ __forceinline__ void
  store_result(const Epilogue<T> &e, const uint32_t g, const T r)
  {
    if (sizeof(T) == 4 && e.out_wide != nullptr) // (uniform; never taken by the double kernels)
      __builtin_nontemporal_store((double)r, &e.out_wide[g]);
    else
      __builtin_nontemporal_store(r, &e.out[g]);
  }
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
  line_mult(const T *__restrict__ Mc, const T (&in)[P * B + 1], T (&out)[P * B + 1])
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
    apply(const T *__restrict__ Ae, const T *__restrict__ Ao, const EvenOdd &x)
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
  line_MK(const Mats<P, T> &m, const T (&a)[P * B + 1], T (&outM)[P * B + 1], T (&outK)[P * B + 1])
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
  line_M_KM(const Mats<P, T> &m, const T (&a)[P * B + 1], const T (&b)[P * B + 1], T (&outM)[P * B + 1], T (&outS)[P * B + 1])
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
  line_KM(const Mats<P, T> &m, const T (&a)[P * B + 1], const T (&b)[P * B + 1], T (&outS)[P * B + 1])
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
  line_stream(const Mats<P, T> &m, T *__restrict__ A, T *__restrict__ Bb, const int stride, const T scale)
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
  seg_products(const Mats<P, T> &m, const T (&a)[P + 5], const T (&b)[P + 5], const bool has_left, T (&o1)[5], T (&o2)[5])
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
  lattice_sweeps(T *__restrict__ bufA, T *__restrict__ bufB, const Mats<P, T> &m, int tid, int nslots, const double *__restrict__ hslot,
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
  hanging_passes(T *__restrict__ buf, const Mats<P, T> &m, int sl, int u, int v, bool act, uint32_t mask, bool transpose)
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
  brick_constraint_passes(T *__restrict__ buf, const Mats<P, T> &m, int tid, int nslots, const uint32_t *__restrict__ fmask, bool transpose,
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

  // The same passes for ONE constrained brick per workgroup (N^2 >= 256 lines: the 17-point lattices), node by node over the
  // brick's hanging faces.  The line form above gives every hanging line to one thread -- 17 of 256 threads busy per hanging face
  // and direction, each with a serial chain of 17 loads, 18 (P + 1) multiply-adds and 17 stores whose weights stream through the
  // scalar registers -- and was the reason why constrained 4^3 bricks at p = 4 cost twice a plain brick (rounds 2 and 3).  Here,
  // per direction d:
  //   faces   each of the (up to four) hanging faces that contain d: its N (N - 2) nodes on lines STRICTLY inside the face,
  //           thread t -> (line t / N, node t % N): one round of 255 threads at N = 17
  //   edges   the four lattice edges along d: 4 N nodes; an edge line is hanging if one of its two faces is, or the edge itself
  // every node needs P + 1 (forward) or up to 2 (2 P + 1) (transposed) multiply-adds with weights E = [I0; I1] from LDS; the new
  // values wait in registers for a barrier because the passes work in place.  fm is uniform in the workgroup.
  template <int P, int B>
  constexpr int
  face_table_words() // LDS words (of the number type) behind the D^-1 table of the persistent brick kernel: E = [I0; I1]
  {
    return (B > 2 && (P * B + 1) * (P * B + 1) >= 256) ? (2 * P + 1) * (P + 1) : 0;
  }
  template <typename T, int P, int B, int BLOCK>
  __device__ __forceinline__ void
  brick_face_passes(T *__restrict__ buf, const T *__restrict__ E, const uint32_t fm, const int tid_in, const bool transpose)
  {
    constexpr int N = P * B + 1, n = P + 1, BC = B / 2;
    static_assert(N * (N - 2) <= BLOCK && 4 * N <= BLOCK, "one round per face and one for the edges");
    // an opaque copy of the thread index: everything below is invariant across the slots of a persistent workgroup, and the compiler
    // would hoist the address arithmetic of all thirty node rounds out of the slot loop and spill it (measured: 30-140 spills)
    int tid;
    asm volatile("v_mov_b32_e32 %0, %1" : "=v"(tid) : "v"(tid_in));
    // new value of the node at coordinate i of the line (base, stride)
    auto node_value = [&](const int base, const int stride, const int i) -> T {
      if (!transpose)
        {
          // fine node a of parent cell k from the cell's P + 1 parent values (b < P: at 2 k P + b; b = P: at 2 (k + 1) P)
          const int k = min(i / (2 * P), BC - 1), a = i - 2 * k * P;
          T         s = T(0);
#pragma unroll
          for (int b = 0; b < n; ++b)
            s += E[a * n + b] * buf[base + (b < P ? 2 * k * P + b : 2 * (k + 1) * P) * stride];
          return s;
        }
      // parent value b of parent cell k collects its column of E over the cell's fine nodes (a node shared by two cells counts
      // once, with the lower cell); the other positions of the line become zero
      const bool last = i == N - 1;
      const int  k = last ? BC - 1 : i / (2 * P), r = i - 2 * k * P;
      if (!last && r >= P)
        return T(0);
      const int b = last ? P : r;
      T         s = T(0);
#pragma unroll
      for (int a = 0; a <= 2 * P; ++a)
        if (a > 0 || k == 0)
          s += E[a * n + b] * buf[base + (2 * k * P + a) * stride];
      if (b == 0 && k > 0)
        {
#pragma unroll
          for (int a = 0; a <= 2 * P; ++a)
            if (a > 0 || k == 1)
              s += E[a * n + P] * buf[base + (2 * (k - 1) * P + a) * stride];
        }
      return s;
    };
#pragma unroll
    for (int dd = 0; dd < 3; ++dd)
      {
        const int d = transpose ? 2 - dd : dd;
        const int e = d == 0 ? 1 : 0, f = d == 2 ? 1 : 2; // the other two directions, (u, v) as in brick_constraint_passes
        const int sd = d == 0 ? 1 : (d == 1 ? N : N * N), se = e == 0 ? 1 : (e == 1 ? N : N * N), sf = f == 0 ? 1 : (f == 1 ? N : N * N);
        const uint32_t he0 = (fm >> (2 * e)) & 1u, he1 = (fm >> (2 * e + 1)) & 1u, hf0 = (fm >> (2 * f)) & 1u, hf1 = (fm >> (2 * f + 1)) & 1u;
        // edge along d at sides (s1 of (d+1)%3, s2 of (d+2)%3); (u, v) = coordinates in (e, f): d = 1 has (e, f) = (x, z) = ((d+2)%3, (d+1)%3)
        auto edge_bit = [&](int su, int sv) -> uint32_t {
          const int s1 = d == 1 ? sv : su, s2 = d == 1 ? su : sv;
          return (fm >> (6 + 4 * d + s1 + 2 * s2)) & 1u;
        };
        const bool any = he0 | he1 | hf0 | hf1 | edge_bit(0, 0) | edge_bit(1, 0) | edge_bit(0, 1) | edge_bit(1, 1);
        if (!any)
          continue; // (uniform)
        // (one face or the edges at a time: read, barrier, write, barrier -- a single new value per thread is live, the kernel has
        // no registers to spare next to its pipeline state; only hanging entities cost anything: the branches are uniform)
        const int line = tid / N + 1, i = tid % N;
#pragma unroll
        for (int q = 0; q < 5; ++q)
          {
            bool on = false;
            int  base = 0;
            if (q < 4)
              {
                // faces u = 0, u = N - 1 (lines at v = 1 .. N - 2), v = 0, v = N - 1 (lines at u = 1 .. N - 2)
                const uint32_t h = q == 0 ? he0 : (q == 1 ? he1 : (q == 2 ? hf0 : hf1));
                if (!h)
                  continue; // (uniform)
                const int u = q == 0 ? 0 : (q == 1 ? N - 1 : line), v = q == 2 ? 0 : (q == 3 ? N - 1 : line);
                on   = tid < N * (N - 2);
                base = u * se + v * sf;
              }
            else
              {
                // the four edges along d
                const int c = tid / N, su = c & 1, sv = (c >> 1) & 1;
                on   = tid < 4 * N && (((su ? he1 : he0) | (sv ? hf1 : hf0) | edge_bit(su, sv)) != 0u);
                base = su * (N - 1) * se + sv * (N - 1) * sf;
              }
            T nv = T(0);
            if (on)
              nv = node_value(base, sd, i);
            __syncthreads(); // every thread has read the old values of its line
            if (on)
              buf[base + i * sd] = nv;
            __syncthreads();
          }
      }
  }

  // ---- 1D embedding of the h-transfer along one lattice line (brick transfers and the transfers fused into the operator)
  // fine line (P*BC*2+1) from coarse line (P*BC+1), cell by cell
  template <typename T, int P, int BC>
  __device__ __forceinline__ void
  line_embed(const T *__restrict__ E, const T (&in)[P * BC + 1], T (&out)[2 * P * BC + 1])
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
  line_embed_T(const T *__restrict__ E, const T (&in)[2 * P * BC + 1], T (&out)[P * BC + 1])
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
  line_embed_sym(const T *__restrict__ Eh, const T (&in)[P * BC + 1], T (&out)[2 * P * BC + 1])
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
  line_embed_sym_T(const T *__restrict__ Eh, const T (&in)[2 * P * BC + 1], T (&out)[P * BC + 1])
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
    T               Eh[(P + 1) * (P + 1)]; // rows 0..P of the 1D h-embedding (line_embed_sym)
    T              *coarse  = nullptr; // RESTRICT: the coarse defect (+=);  PROLONGATE: the coarse solution (read only)
    T              *x_inout = nullptr; // PROLONGATE: == src; x + P x_c of the fused bricks' interior DoFs is stored in place
    T              *scratch = nullptr; // PROLONGATE: x + P x_c of the shell DoFs a fused brick owns (indexed like x)
  };

  template <typename T, int P>
  struct ApplyArgs
  {
    SlotGroupDev g;
    Mats<P, T>   m;
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

} // namespace mgamd
