/* mgamd_dev.h -- development and measurement entry points of libmgamd.so that are NOT part of the drop-in boundary
 * (include/mgamd.h): in-kernel phase stamps of debug builds and the HIP-event profile of the dominant kernel that bench.py turns
 * into `roofline.achieved`.  Nothing here replaces an interface of the reference. */
#ifndef MGAMD_DEV_H
#define MGAMD_DEV_H

#include "mgamd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* development aid: with MGAMD_STAMPS=<mode> in the environment the largest slot group's kernel of that mode
 * records 8 wall-clock stamps (10 ns ticks) per workgroup at its phase boundaries; returns them */
int mgamd_level_op_debug_stamps(mgamd_level_op *op, unsigned long long *out, uint64_t max_count, uint64_t *count);

/* per-kernel device time of the dominant kernel (cell operator) accumulated since the last reset,
 * measured with HIP events when profiling is enabled (bench.py roofline.achieved) */
int mgamd_ctx_kernel_profile(mgamd_ctx *ctx, int enable);
/* which launches are measured: brick_size = 0 (default) the slot group with the most work on every level;
 * brick_size = B only groups of B^3-cell bricks, i.e. the launches of ONE kernel symbol (what rocprofv3 averages) */
int mgamd_ctx_kernel_profile_brick(mgamd_ctx *ctx, int brick_size);
int mgamd_ctx_kernel_profile_read(mgamd_ctx *ctx, double *total_ms, uint64_t *n_launches, double *algorithmic_bytes);
/* the bytes the measured launches are written to move themselves (the slot-interior D^-1 is evaluated in closed form by the
 * p = 1 and the persistent 17-point lattice kernels, one word less than SURVEY 8(d)'s figure in `algorithmic_bytes`) */
int mgamd_ctx_kernel_profile_bytes_moved(mgamd_ctx *ctx, double *bytes_moved);

#ifdef __cplusplus
}
#endif
#endif /* MGAMD_DEV_H */
