// Explicit instantiations of LevelOperator<T>::apply_P<P, MODE> for ONE number type and ONE degree
// (-DMGAMD_INST_T=double|float -DMGAMD_INST_P=1..4): the kernels of kernels.hpp are compiled here, in eight translation units
// built in parallel (Makefile), instead of in one.
#include "level_operator_apply.hpp"

#if !defined(MGAMD_INST_T) || !defined(MGAMD_INST_P)
#error "compile with -DMGAMD_INST_T=<double|float> -DMGAMD_INST_P=<degree>"
#endif

namespace mgamd
{
#define MGAMD_INST(MODE)                                                                                                                  \
  template void LevelOperator<MGAMD_INST_T>::apply_P<MGAMD_INST_P, MODE>(const MGAMD_INST_T *, const Epilogue<MGAMD_INST_T> &, bool, double, \
                                                                         int, const FusedTransferHost<MGAMD_INST_T> *);
  MGAMD_INST(MODE_VMULT)
  MGAMD_INST(MODE_RESIDUAL)
  MGAMD_INST(MODE_CHEB)
  MGAMD_INST(MODE_CHEB_FIRST)
  MGAMD_INST(MODE_CHEB_SECOND)
  // the passes with fused level transfers: degrees with a persistent 17-point lattice kernel
#if MGAMD_INST_P == 1 || MGAMD_INST_P == 2 || MGAMD_INST_P == 4
  MGAMD_INST(MODE_RESIDUAL_RESTRICT)
  MGAMD_INST(MODE_CHEB_PROLONGATE)
#endif
#undef MGAMD_INST
} // namespace mgamd
