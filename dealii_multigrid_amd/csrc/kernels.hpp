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
//
// K7  csr_spmv_kernel        CSR products of the algebraic coarse solver, fused with the Chebyshev update.
//
// The kernels live in kernels_common.hpp (1D products, sweeps, constraint passes, argument structs), kernels_apply.hpp (K1-K3),
// kernels_transfer.hpp (K4), kernels_vector.hpp (K5, K6) and kernels_amg.hpp (K7); this header includes them all.
#pragma once
#include "kernels_common.hpp"
#include "kernels_apply.hpp"
#include "kernels_transfer.hpp"
#include "kernels_vector.hpp"
#include "kernels_amg.hpp"
