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

    tile_load<0>(q + cx.b0 * n, cx.rows, n, magic_q, lq, lane, fast && (align & AL_Q), cx.full && (align & AL_Q));
    tile_load<0>(qd + cx.b0 * n, cx.rows, n, magic_q, lqd, lane, fast && (align & AL_QD), cx.full && (align & AL_QD));
    if (qdd) tile_load<0>(qdd + cx.b0 * n, cx.rows, n, magic_q, lqdd, lane, fast && (align & AL_QDD), cx.full && (align & AL_QDD));
    for (int s = 0; s < n_slots * 6; ++s) lfs[s * WAVE + lane] = 0.0f;
    wave_lds_sync();

    // lanes past a partial tile read zeros (not stale LDS): their angles must not be able to push the wave onto
    // the rare large-angle sincos path, which would change the rounding of the live lanes from run to run
    const unsigned row = lane * Sq;
    const bool live = (int)lane < cx.rows;
    const bool has_qdd = qdd != nullptr;
    auto qf = [&](int d, float &a, float &v, float &acc) {
        a = live ? lq[row + d] : 0.0f;
        v = lqd[row + d];
        acc = has_qdd ? lqdd[row + d] : 0.0f;
    };
    auto tau_out = [&](int d, float v) { ltau[row + d] = v; };
    auto motion_save = [&](int s, const Motion &M) {
        float *b = lms + s * (12 * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            b[i * WAVE] = M.wa[i][0]; b[(3 + i) * WAVE] = M.va[i][0]; b[(6 + i) * WAVE] = M.wa[i][1];
            b[(9 + i) * WAVE] = M.va[i][1];
        }
    };
    auto motion_load = [&](int s, Motion &M) {
        const float *b = lms + s * (12 * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            M.wa[i] = f2_make(b[i * WAVE], b[(6 + i) * WAVE]);
            M.va[i] = f2_make(b[(3 + i) * WAVE], b[(9 + i) * WAVE]);
        }
    };
    auto force_add = [&](int s, const Force &F) {
        float *b = lfs + s * (6 * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) { b[i * WAVE] += F.la[i][0]; b[(3 + i) * WAVE] += F.la[i][1]; }
    };
    auto force_take = [&](int s, Force &F) {
        float *b = lfs + s * (6 * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            F.la[i] += f2_make(b[i * WAVE], b[(3 + i) * WAVE]);
            b[i * WAVE] = 0.0f; b[(3 + i) * WAVE] = 0.0f;
        }
    };
    rnea_walk<CAP>(ops_f, ops_i, flags, qf, tau_out, motion_save, motion_load, force_add, force_take);
    wave_lds_sync();
    tile_store<0>(tau + cx.b0 * n, cx.rows, n, magic_q, ltau, lane, fast && (align & AL_TAU), cx.full && (align & AL_TAU));
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
#ifndef DRM_NO_ARM_KERNEL
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7 && B >= WAVE && B / WAVE < 0x7fffffffLL &&
        align == (AL_Q | AL_QD | AL_TAU | (qdd ? AL_QDD : 0u)) && (((uintptr_t)w->ops_f) & 15u) == 0) {
        // 7-DoF arms: full tiles through the packed-FP32 chain kernel, ragged tail through the generic one
        const int n_tiles = (int)(B / WAVE);
        launch_rnea_arm(w->ops_f, q, qd, qdd, n_tiles, (int)flags, tau, s);
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done == B) return launched();
        rc = launched();
        if (rc) return rc;
        drm_walk generic = *w;
        generic.shape &= ~DRM_WALK_ARM_CHAIN;
        return drm_rnea(&generic, q + done * n, qd + done * n, qdd ? qdd + done * n : nullptr, B - done, flags,
                        tau + done * n, stream);
    }
#endif
    DRM_DISPATCH_CAP(w->capacity, {
        rc = ensure_lds(rnea_kernel<C>, g.lds_bytes);
        if (rc) return rc;
        hipLaunchKernelGGL(rnea_kernel<C>, g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, n, (int)w->n_slots,
                           (int)flags, q, qd, qdd, B, tau, div_magic(n), g.lds_per_wave, align);
    })
    return launched();
}

extern "C" int drm_fk_rnea(const drm_walk *tree, const drm_walk *chain, int32_t target_op, const float *q, const float *qd,
                           const float *qdd, int64_t B, int32_t flags, float *tau, float *pos, float *quat, void *stream) {
    int rc = check_walk(tree);
    if (rc) return rc;
    rc = check_walk(chain);
    if (rc) return rc;
    if (!q || !qd || !tau || !pos || !quat) return fail(DRM_ERR_INVALID, "q / qd / tau / pos / quat must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (tree->n_dofs != chain->n_dofs) return fail(DRM_ERR_INVALID, "the two walks belong to different robots");
    if (B == 0) return DRM_OK;
    const int n = tree->n_dofs;
    hipStream_t s = (hipStream_t)stream;
#ifndef DRM_NO_ARM_KERNEL
    const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(tau, AL_TAU) | al16(pos, AL_POS) |
                           al16(quat, AL_QUAT);
    if ((tree->shape & DRM_WALK_ARM_CHAIN) && tree->capacity == 8 && n == 7 && target_op == tree->n_ops - 1 &&
        chain->target_perm == 2 && B >= WAVE && B / WAVE < 0x7fffffffLL && (((uintptr_t)tree->ops_f) & 15u) == 0 &&
        align == (AL_Q | AL_QD | AL_TAU | AL_POS | AL_QUAT | (qdd ? AL_QDD : 0u))) {
        // a serial 7-DoF arm whose last link is the target: full tiles through the fused kernel
        const int n_tiles = (int)(B / WAVE);
        launch_fk_rnea_arm(tree->ops_f, q, qd, qdd, n_tiles, (int)flags, tau, pos, quat, s);
        rc = launched();
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (rc || done == B) return rc;
        q += done * n; qd += done * n; qdd = qdd ? qdd + done * n : nullptr;
        tau += done * n; pos += done * 3; quat += done * 4; B -= done;
    }
#endif
    // every other robot (and a ragged tail): the two walks one after the other on the same stream
    rc = drm_fk(chain, q, B, 1, pos, quat, stream);
    if (rc) return rc;
    return drm_rnea(tree, q, qd, qdd, B, flags, tau, stream);
}
