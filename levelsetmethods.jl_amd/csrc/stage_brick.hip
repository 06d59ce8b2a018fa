// stage_brick.hip — the brick kernels of the narrow band (stage_brick.h), FAST arithmetic, their own translation unit
#define LSM_STRICT 0
#define LSM_NS fast_math
#include "stage_brick.h"

namespace lsm {
int launch_stage_brick(const Combo& c, const StageArgs& a, hipStream_t s) {
#define LSM_X(ADV, NM, CURV, EIK) \
    if (c.adv == ADV && c.nm == NM && c.curv == CURV && c.eik == EIK) return LSM_NS::launch_bricks<ADV, NM, CURV, EIK>(a, s);
    LSM_FOR_EACH_COMBO(LSM_X)
#undef LSM_X
    return -1;
}
}  // namespace lsm
