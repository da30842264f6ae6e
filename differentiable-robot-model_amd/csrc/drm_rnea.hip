// drm_rnea.hip — K3: recursive Newton-Euler inverse dynamics over the whole tree.
//
// Replaces DifferentiableRobotModel.compute_inverse_dynamics (robot_model.py:305-375: update_kinematic_state
// 139-195 + update_joint_acc rigid_body.py:159-165 + iterative_newton_euler robot_model.py:250-303) and
// compute_non_linear_effects (robot_model.py:377-400, qdd = NULL).
//
// Per sample: in q, qd, qdd [n] (12 n bytes), out tau[n] (4 n bytes).   n = 7: 112 B, ~2.6 kflop.
// LDS per wave: [ q ][ qd ][ qdd ][ tau ] each 64 (n|1), then [ motion slots : n_slots*12*64 ][ force slots : n_slots*6*64 ]
#include "drm_common.hpp"
#include "drm_sample.hpp"

namespace drm {

template <int CAP>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    rnea_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int n, int n_slots, int flags,
                const float *__restrict__ q, const float *__restrict__ qd, const float *__restrict__ qdd, int64_t B,
                float *__restrict__ tau, uint32_t magic_q, int lds_per_wave, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx cx;
    if (!wave_begin(B, lds_per_wave, smem, cx)) return;
    const unsigned lane = cx.lane;
    const int Sq = pad_odd(n);
    const int region = round4(WAVE * Sq);
    float *lq = cx.lds, *lqd = lq + region, *lqdd = lqd + region, *ltau = lqdd + region;
    float *lms = ltau + region;                 // motion slots [slot][12][64]
    float *lfs = lms + n_slots * (12 * WAVE);   // force slots  [slot][6][64]
    const bool fast = cx.full && (n & 1);

    tile_load<0>(q + cx.b0 * n, cx.rows, n, magic_q, lq, lane, fast && (align & AL_Q));
    tile_load<0>(qd + cx.b0 * n, cx.rows, n, magic_q, lqd, lane, fast && (align & AL_QD));
    if (qdd) tile_load<0>(qdd + cx.b0 * n, cx.rows, n, magic_q, lqdd, lane, fast && (align & AL_QDD));
    for (int s = 0; s < n_slots * 6; ++s) lfs[s * WAVE + lane] = 0.0f;
    wave_lds_sync();

    const unsigned row = lane * Sq; // lanes past a partial tile's last row compute garbage, never stored
    const bool has_qdd = qdd != nullptr;
    auto qf = [&](int d, float &a, float &v, float &acc) {
        a = lq[row + d];
        v = lqd[row + d];
        acc = has_qdd ? lqdd[row + d] : 0.0f;
    };
    auto tau_out = [&](int d, float v) { ltau[row + d] = v; };
    auto motion_save = [&](int s, const Motion &M) {
        float *b = lms + s * (12 * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            b[i * WAVE] = M.w[i]; b[(3 + i) * WAVE] = M.v[i]; b[(6 + i) * WAVE] = M.al[i]; b[(9 + i) * WAVE] = M.a[i];
        }
    };
    auto motion_load = [&](int s, Motion &M) {
        const float *b = lms + s * (12 * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            M.w[i] = b[i * WAVE]; M.v[i] = b[(3 + i) * WAVE]; M.al[i] = b[(6 + i) * WAVE]; M.a[i] = b[(9 + i) * WAVE];
        }
    };
    auto force_add = [&](int s, const Force &F) {
        float *b = lfs + s * (6 * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) { b[i * WAVE] += F.l[i]; b[(3 + i) * WAVE] += F.a[i]; }
    };
    auto force_take = [&](int s, Force &F) {
        float *b = lfs + s * (6 * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            F.l[i] += b[i * WAVE]; F.a[i] += b[(3 + i) * WAVE];
            b[i * WAVE] = 0.0f; b[(3 + i) * WAVE] = 0.0f;
        }
    };
    rnea_walk<CAP>(ops_f, ops_i, flags, qf, tau_out, motion_save, motion_load, force_add, force_take);
    wave_lds_sync();
    tile_store<0>(tau + cx.b0 * n, cx.rows, n, magic_q, ltau, lane, fast && (align & AL_TAU));
}

} // namespace drm

using namespace drm;

extern "C" int drm_rnea(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags,
                        float *tau, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !qd || !tau) return fail(DRM_ERR_INVALID, "q / qd / tau must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs;
    Geometry g;
    rc = make_geometry(B, 4 * round4(WAVE * pad_odd(n)) + w->n_slots * 18 * WAVE, g);
    if (rc) return rc;
    const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(tau, AL_TAU);
    hipStream_t s = (hipStream_t)stream;
    DRM_DISPATCH_CAP(w->capacity, {
        rc = ensure_lds(rnea_kernel<C>, g.lds_bytes);
        if (rc) return rc;
        hipLaunchKernelGGL(rnea_kernel<C>, g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, n, (int)w->n_slots,
                           (int)flags, q, qd, qdd, B, tau, div_magic(n), g.lds_per_wave, align);
    })
    return launched();
}
