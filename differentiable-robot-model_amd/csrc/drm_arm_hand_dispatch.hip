// drm_arm_hand_dispatch.hip — the entry points of the "arm that carries a hand" kernels (drm_arm_hand.hip), which are compiled
// once per sub-chain length L into four objects: pick the object by the L of the walk's shape.
#include "drm_common.hpp"

namespace drm {

#define DRM_AH_DECLARE(L)                                                                                                        \
    bool arm_hand_compiled_l##L(const drm_walk *w);                                                                              \
    bool crba_arm_hand_applies_l##L(const drm_walk *w);                                                                          \
    int64_t launch_rnea_arm_hand_l##L(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, \
                                      float *tau, hipStream_t s);                                                                \
    int64_t launch_forward_dynamics_arm_hand_l##L(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, \
                                                  int flags, float *qdd, hipStream_t s);                                         \
    int64_t launch_crba_arm_hand_l##L(const drm_walk *w, const float *q, int64_t B, float *H, hipStream_t s);                    \
    int64_t launch_rnea_backward_arm_hand_l##L(const drm_walk *w, const float *q, const float *qd, const float *qdd,             \
                                               const float *gtau, int64_t B, int flags, uint64_t param_mask, float *gq,         \
                                               float *gqd, float *gqdd, float *partials, int &partial_rows, hipStream_t s);
DRM_AH_DECLARE(1) DRM_AH_DECLARE(2) DRM_AH_DECLARE(3) DRM_AH_DECLARE(4)
#undef DRM_AH_DECLARE

#define DRM_AH_DISPATCH(call, none)                                                                                              \
    if (!(w->shape & DRM_WALK_ARM_HAND)) return none;                                                                            \
    switch (DRM_WALK_AH_L(w->shape)) {                                                                                           \
    case 1: return call(1);                                                                                                      \
    case 2: return call(2);                                                                                                      \
    case 3: return call(3);                                                                                                      \
    case 4: return call(4);                                                                                                      \
    default: return none;                                                                                                        \
    }

bool arm_hand_compiled(const drm_walk *w) {
#define C(L) arm_hand_compiled_l##L(w)
    DRM_AH_DISPATCH(C, false)
#undef C
}
bool crba_arm_hand_applies(const drm_walk *w) {
#define C(L) crba_arm_hand_applies_l##L(w)
    DRM_AH_DISPATCH(C, false)
#undef C
}
int64_t launch_rnea_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau,
                             hipStream_t s) {
#define C(L) launch_rnea_arm_hand_l##L(w, q, qd, qdd, B, flags, tau, s)
    DRM_AH_DISPATCH(C, 0)
#undef C
}
int64_t launch_forward_dynamics_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int flags,
                                         float *qdd, hipStream_t s) {
#define C(L) launch_forward_dynamics_arm_hand_l##L(w, q, qd, f, B, flags, qdd, s)
    DRM_AH_DISPATCH(C, 0)
#undef C
}
int64_t launch_crba_arm_hand(const drm_walk *w, const float *q, int64_t B, float *H, hipStream_t s) {
#define C(L) launch_crba_arm_hand_l##L(w, q, B, H, s)
    DRM_AH_DISPATCH(C, 0)
#undef C
}
int64_t launch_rnea_backward_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *qdd, const float *gtau,
                                      int64_t B, int flags, uint64_t param_mask, float *gq, float *gqd, float *gqdd, float *partials,
                                      int &partial_rows, hipStream_t s) {
#define C(L) launch_rnea_backward_arm_hand_l##L(w, q, qd, qdd, gtau, B, flags, param_mask, gq, gqd, gqdd, partials, partial_rows, s)
    DRM_AH_DISPATCH(C, 0)
#undef C
}

} // namespace drm
