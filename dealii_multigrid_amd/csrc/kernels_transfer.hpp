// K4: two-level transfer kernels (MGTwoLevelTransfer, ref:multigrid_throughput.cc:1600-1604): per-cell patches, p = 1 patches in
// registers, brick transfers (one-brick-per-workgroup and persistent).  See kernels.hpp for the overview.
#pragma once
#include "kernels_common.hpp"

namespace mgamd
{
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
    Mats<PC, T>     m;              // only I0/I1 are used (coarse hanging nodes)
    T               E[NF * (PC + 1)]; // 1D embedding, rows = fine nodes
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
    Mats<1, T>      m; // only I0/I1 are used
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
    T               E[(2 * P + 1) * (P + 1)];
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

} // namespace mgamd
