// stage_tu.hip — one translation unit per (arithmetic mode, dimension):
//   hipcc -DLSM_STRICT={0,1} -DLSM_TU_NDIM={1,2,3} [-ffp-contract=off for STRICT]
// so the 66 fused-kernel instantiations compile in parallel.
#ifndef LSM_STRICT
#error "define LSM_STRICT"
#endif
#ifndef LSM_TU_NDIM
#error "define LSM_TU_NDIM"
#endif
#if LSM_STRICT
#define LSM_NS strict_math
#else
#define LSM_NS fast_math
#endif
#include "stage_kernel.h"

#define LSM_CAT_(a, b, c) a##b##c
#define LSM_CAT(a, b, c) LSM_CAT_(a, b, c)

namespace lsm {
#if LSM_STRICT
int LSM_CAT(launch_stage_strict_, LSM_TU_NDIM, d)(const Combo& c, const StageArgs& a, hipStream_t s) {
#else
int LSM_CAT(launch_stage_fast_, LSM_TU_NDIM, d)(const Combo& c, const StageArgs& a, hipStream_t s) {
#endif
    return LSM_NS::launch_ndim<LSM_TU_NDIM>(c, a, s);
}

#if !LSM_STRICT && LSM_TU_NDIM == 3
// tile footprint of the stage kernel per dimension (the narrow-band tile flags must match it)
void stage_tile_shape(int ndim, int* tx, int* ty) {
    if (ndim == 1) { *tx = LSM_NS::TileCfg<1>::TX; *ty = 1; }
    else if (ndim == 2) { *tx = LSM_NS::TileCfg<2>::TX; *ty = 1; }
    else { *tx = LSM_NS::TileCfg<3>::TX; *ty = LSM_NS::TileCfg<3>::TY; }
}
#endif
}  // namespace lsm
