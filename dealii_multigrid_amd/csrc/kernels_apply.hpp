// K1 / K1p / K1c / K2 / K3: the level operator on slot lattices (one workgroup per slot group, persistent workgroups with fused
// level transfers for the 17-point lattices, wave-scoped single cells, p = 1 cell clusters), the tail epilogue and the diagonal.
// See kernels.hpp for the overview.
#pragma once
#include "kernels_common.hpp"

namespace mgamd
{
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
    MGAMD_STAMP(5)

    if (!MGAMD_ABLATED(1))
      {
        if constexpr (G::ROUNDS == 1)
          lattice_sweeps<T, P, B, BLOCK, NoHook, false, false, WAVE>(bufA, bufB, args.m, tid, nslots, &h_mine, NoHook(), true);
        else
          lattice_sweeps<T, P, B, BLOCK, NoHook, false, false, WAVE>(bufA, bufB, args.m, tid, nslots, args.g.h + slot0);
      }
    MGAMD_STAMP(6)

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
              if (MODE == MODE_CHEB)
                store_result(args.epi, e.g, r);
              else
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
    T *Etab = dtab + 2 * P * P * P + 1; // constrained bricks: the embedding E = [I0; I1] for brick_face_passes
    // hanging-node passes node by node over the hanging faces (brick_face_passes) instead of line by line
    // (P >= 2: at p = 1 the line form is cheap -- 3 x 2 weights -- and faster: octant p = 1 L = 9 1.37 ms against 1.58 ms with the node form)
    constexpr bool FACE_PASSES = CONSTR && P >= 2 && B > 2 && face_table_words<P, B>() > 0 && G::N * (G::N - 2) <= G::ABLOCK;

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

    if constexpr (FACE_PASSES)
      for (int t = tid; t < (2 * P + 1) * (P + 1); t += BLOCK)
        {
          const int a = t / (P + 1), b = t % (P + 1);
          Etab[t]     = a <= P ? args.m.I0[a * (P + 1) + b] : args.m.I1[(a - P) * (P + 1) + b];
        }
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
              {
                if constexpr (FACE_PASSES)
                  brick_face_passes<T, P, B, BLOCK>(bufA, Etab, (uint32_t)__builtin_amdgcn_readfirstlane((int)fmcur), tid, false);
                else
                  brick_constraint_passes<T, P, B, BLOCK>(bufA, args.m, tid, 1, args.g.fmask + slot, false, &fmcur);
              }
          }
        // cell prefetch in the streamed sweeps (-4 % on the 2-4-word passes); the 5-word mode has no registers left for it
        lattice_sweeps<T, P, B, BLOCK, NoHook, true, (MODE != MODE_CHEB || sizeof(T) == 4)>(bufA, bufB, args.m, tid, 1, &hcur);
        if constexpr (brick_may_be_constrained(B, CONSTR))
          if (any_hanging)
            {
              if constexpr (FACE_PASSES)
                brick_face_passes<T, P, B, BLOCK>(bufA, Etab, (uint32_t)__builtin_amdgcn_readfirstlane((int)fmcur), tid, true);
              else
                brick_constraint_passes<T, P, B, BLOCK>(bufA, args.m, tid, 1, args.g.fmask + slot, true, &fmcur);
            }
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
                  if (MODE == MODE_CHEB && !FUSE_P)
                    store_result(args.epi, g, r);
                  else
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
  // workgroups per CU: two f64 lattice pairs fill the LDS; three f32 pairs would fit, and at p = 4 the float kernels can be held
  // to 168 VGPRs, but measured no gain (octant p = 4 float 7.07 vs 6.73 ms per V-cycle, boxes 4 % apart): two everywhere
  template <typename T, int P>
  constexpr int
  persistent_wgs_per_cu()
  {
    return 2;
  }
  template <typename T, int P, int B, int MODE, bool CONSTR = false>
  __global__ void
  __launch_bounds__((Geo<P, B>::ABLOCK), (persistent_wgs_per_cu<T, P>())) lattice_apply_persistent_kernel(const ApplyArgs<T, P> args)
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
    Mats<1, T>     m;
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
  hanging_in_registers_p1(T (&x)[8], const uint32_t mask, const Mats<1, T> &m)
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
                  const T *w       = c ? m.I1 : m.I0;
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
                  {
                    const T r = xv[u] + epi.f1 * (xv[u] - xo[u]) + epi.f2 * dv[u] * (bv[u] - ax[u]);
                    if (MODE_ == MODE_CHEB)
                      store_result(epi, gi, r);
                    else
                      NT_STORE(r, &epi.out[gi]);
                  }
                else
                  epi.out[gi] = (i < n_tail && fabs((double)ax[u]) > 1.0e-10) ? T(1) / ax[u] : T(1);
              }
          }
      }
  }

} // namespace mgamd
