// drm_host.hip — host-side pieces of the C ABI shared by all kernels: error string, walk checks, launch geometry.
#include <string.h>

#include "drm_common.hpp"

namespace drm {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *a, long b, long c) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

int check_walk(const drm_walk *w) {
    if (!w) return fail(DRM_ERR_INVALID, "walk is NULL");
    if (!w->ops_f || !w->ops_i) return fail(DRM_ERR_INVALID, "walk tables are NULL");
    const int c = w->capacity;
    if (c != 4 && c != 8 && c != 12 && c != 16 && c != 24 && c != 32)
        return fail(DRM_ERR_UNSUPPORTED, "walk capacity %s%ld is not one of 4/8/12/16/24/32", "", c);
    if (w->n_ops < 0 || w->n_ops > c)
        return fail(DRM_ERR_INVALID, "walk has %s%ld ops but capacity %ld", "", w->n_ops, c);
    if (w->n_dofs < 1 || w->n_dofs > DRM_MAX_DOFS)
        return fail(DRM_ERR_UNSUPPORTED, "n_dofs %s%ld outside [1, %ld]", "", w->n_dofs, DRM_MAX_DOFS);
    if (w->n_slots < 0 || w->n_slots > DRM_MAX_SLOTS)
        return fail(DRM_ERR_UNSUPPORTED, "walk needs %s%ld save slots, kernels have %ld", "", w->n_slots, DRM_MAX_SLOTS);
    return DRM_OK;
}

int launched() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return DRM_OK;
}

int make_geometry(int64_t B, int lds_floats_per_wave, Geometry &g) {
    const size_t per_wave = (size_t)round4(lds_floats_per_wave) * sizeof(float);
    if (per_wave > (size_t)MAX_LDS_BYTES)
        return fail(DRM_ERR_UNSUPPORTED, "tile needs %s%ld bytes of LDS per wave (max %ld)", "", (long)per_wave,
                    (long)MAX_LDS_BYTES);
    int wpb = MAX_WAVES_PER_BLOCK;
    while (wpb > 1 && per_wave * wpb > (size_t)64 * 1024) wpb >>= 1;
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    const int64_t blocks = (tiles + wpb - 1) / wpb;
    if (blocks > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    g.grid = dim3((unsigned)blocks);
    g.block = dim3(WAVE * wpb);
    g.lds_per_wave = round4(lds_floats_per_wave);
    g.lds_bytes = per_wave * wpb;
    return DRM_OK;
}

const char *last_error() { return g_err; }

} // namespace drm

extern "C" {
int drm_abi_version(void) { return DRM_ABI_VERSION; }
const char *drm_last_error(void) { return drm::last_error(); }
}
