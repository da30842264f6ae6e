// drm_fk_jacobian.hip — K2: FK + geometric Jacobian of one chain (the metric kernel).
//
// Replaces DifferentiableRobotModel.compute_endeffector_jacobian (robot_model.py:626-667) and the
// compute_forward_kinematics call inside it (robot_model.py:641 -> 223-248, 139-195).
//
// Per sample: in  q[n]                                   ( 4 n bytes)
//             out pos[3] quat[4] lin_jac[3,n] ang_jac[3,n] (4 (7 + 6 n) bytes)      n = 7: 224 B
// LDS per wave: [ q tile : 64 (n|1) ][ staging : 64 max(3n|1, 3) ]
#include "drm_common.hpp"
#include "drm_sample.hpp"

namespace drm {

// NDOF > 0 fixes the row width at compile time (tile copies fully unrolled, immediate LDS offsets).
template <int CAP, int NDOF>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    fk_jacobian_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int n_rt, uint64_t dof_mask,
                       const float *__restrict__ q, int64_t B, float *__restrict__ pos, float *__restrict__ quat,
                       float *__restrict__ lin, float *__restrict__ ang, uint32_t magic_q, uint32_t magic_j,
                       int lds_per_wave, uint32_t align, int target_perm) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx cx;
    if (!wave_begin(B, lds_per_wave, smem, cx)) return;
    const unsigned lane = cx.lane;
    const int n = NDOF ? NDOF : n_rt;
    const int Sq = pad_odd(n), Sj = pad_odd(3 * n);
    float *lq = cx.lds;
    float *stage = cx.lds + round4(WAVE * Sq);

    // the wave-uniform DoF column of every op in one wide scalar load; the float table's cache
    // lines are touched now so their misses overlap with the q tile load
    int dof[CAP];
    load_field<CAP>(ops_i, DRM_OPI_DOF, dof);
    WarmRegs<CAP> warm;
    warm_walk_issue<CAP, 1>(ops_f, warm);
    tile_load<NDOF>(q + cx.b0 * n, cx.rows, n, magic_q, lq, lane, cx.full && (n & 1) && (align & AL_Q), cx.full && (align & AL_Q));
    warm_walk_wait(warm);
    wave_lds_sync();

    // lanes beyond the last valid row of a partial tile read stale LDS and compute garbage that is
    // never stored (the arithmetic is branch-free, so garbage is harmless)
    const bool live = (int)lane < cx.rows;
    const float *qrow = lq + lane * Sq;
    auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; }; // zeros, not stale LDS, past a partial tile

    Pose ee;
    float z[CAP][3], pj[CAP][3];
    fk_chain<CAP>(ops_f, dof, qf, ee, z, pj);

    // ---- pos [B,3]: 3 floats per lane -> LDS -> coalesced store ------------------------
    if (pos) {
        stage[lane * 3 + 0] = ee.p[0];
        stage[lane * 3 + 1] = ee.p[1];
        stage[lane * 3 + 2] = ee.p[2];
        wave_lds_sync();
        tile_store<3>(pos + cx.b0 * 3, cx.rows, 3, 0u, stage, lane, cx.full && (align & AL_POS));
        wave_lds_sync();
    }
    // ---- quat [B,4]: one 16-byte store per lane is already coalesced -------------------
    if (quat && live) {
        // the target is the last REAL op; padding ops are identities, so `ee` still carries the
        // target's canonical frame: undo its column permutation before the quaternion
        float Ru[9], qt[4];
#pragma unroll
        for (int i = 0; i < 9; ++i) Ru[i] = ee.R[i];
        unpermute(target_perm, Ru);
        quat_xyzw(Ru, qt);
        float *dst = quat + (cx.b0 + lane) * 4;
        if (align & AL_QUAT) {
            store16_wt(dst, make_float4(qt[0], qt[1], qt[2], qt[3]));
        } else {
            dst[0] = qt[0]; dst[1] = qt[1]; dst[2] = qt[2]; dst[3] = qt[3];
        }
    }
    // ---- Jacobians [B,3,n]: column d of op k at row offset r*n + d ----------------------
    float *jrow = stage + lane * Sj;
    const uint64_t all = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    const bool fast_j = cx.full; // 3n is odd iff n is odd; checked per tensor below
    const bool odd_j = (3 * n) & 1;

    // linear part: z_k x (p_e - p_k)   (robot_model.py:661)
    if ((dof_mask & all) != all) {
        for (int d = 0; d < n; ++d)
            if (!((dof_mask >> d) & 1ull)) { jrow[d] = 0.0f; jrow[n + d] = 0.0f; jrow[2 * n + d] = 0.0f; }
    }
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const int d = dof[k];
        if (d >= 0) {
            const float dp[3] = {ee.p[0] - pj[k][0], ee.p[1] - pj[k][1], ee.p[2] - pj[k][2]};
            float c[3];
            cross3(z[k], dp, c);
            jrow[d] = c[0];
            jrow[n + d] = c[1];
            jrow[2 * n + d] = c[2];
        }
    }
    wave_lds_sync();
    tile_store<3 * NDOF>(lin + cx.b0 * 3 * n, cx.rows, 3 * n, magic_j, stage, lane, fast_j && odd_j && (align & AL_LIN),
                         fast_j && (align & AL_LIN));
    wave_lds_sync();

    // angular part: z_k   (robot_model.py:662); the off-chain zeros are still in place
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const int d = dof[k];
        if (d >= 0) {
            jrow[d] = z[k][0];
            jrow[n + d] = z[k][1];
            jrow[2 * n + d] = z[k][2];
        }
    }
    wave_lds_sync();
    tile_store<3 * NDOF>(ang + cx.b0 * 3 * n, cx.rows, 3 * n, magic_j, stage, lane, fast_j && odd_j && (align & AL_ANG),
                         fast_j && (align & AL_ANG));
}

} // namespace drm

using namespace drm;

extern "C" int drm_fk_jacobian(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin_jac,
                               float *ang_jac, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !lin_jac || !ang_jac) return fail(DRM_ERR_INVALID, "q / lin_jac / ang_jac must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (w->target_perm < 0 || w->target_perm > 5) return fail(DRM_ERR_INVALID, "target_perm must be in 0..5");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs;
    const int Sq = pad_odd(n), Sj = pad_odd(3 * n);
    Geometry g;
    rc = make_geometry(B, round4(WAVE * Sq) + round4(WAVE * (Sj > 3 ? Sj : 3)), g);
    if (rc) return rc;
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT) | al16(lin_jac, AL_LIN) |
                           al16(ang_jac, AL_ANG);
    const uint32_t mq = div_magic(n), mj = div_magic(3 * n);
    hipStream_t s = (hipStream_t)stream;
#define DRM_LAUNCH_FKJ(C, N)                                                                                         \
    {                                                                                                                \
        rc = ensure_lds(fk_jacobian_kernel<C, N>, g.lds_bytes);                                                      \
        if (rc) return rc;                                                                                           \
        hipLaunchKernelGGL((fk_jacobian_kernel<C, N>), g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, n,       \
                           w->dof_mask, q, B, pos, quat, lin_jac, ang_jac, mq, mj, g.lds_per_wave, align,            \
                           (int)w->target_perm);                                                             \
    }
    const uint32_t all_al = AL_Q | AL_POS | AL_QUAT | AL_LIN | AL_ANG;
#ifdef DRM_NO_ARM_KERNEL
    if (false) {
#else
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7 && w->target_perm == 2 && align == all_al &&
        (((uintptr_t)w->ops_f) & 15u) == 0 && B >= WAVE && B / WAVE < 0x7fffffffLL) {
#endif
        // 7-DoF arms (Franka Panda, KUKA iiwa): full tiles through the packed-FP32 chain kernel, ragged tail (if
        // any) through the generic one
        const int n_tiles = (int)(B / WAVE);
        launch_fk_jacobian_arm(w->ops_f, q, n_tiles, pos, quat, lin_jac, ang_jac, s);
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done < B) {
            rc = launched();
            if (rc) return rc;
            drm_walk generic = *w;
            generic.shape &= ~DRM_WALK_ARM_CHAIN;
            return drm_fk_jacobian(&generic, q + done * n, B - done, pos + done * 3, quat + done * 4,
                                   lin_jac + done * 3 * n, ang_jac + done * 3 * n, stream);
        }
    } else if (w->capacity == 8 && n == 7) {
        DRM_LAUNCH_FKJ(8, 7) // 7-DoF robots, ragged / unaligned / partial-output calls: static tile shapes
    } else {
        DRM_DISPATCH_CAP(w->capacity, DRM_LAUNCH_FKJ(C, 0))
    }
#undef DRM_LAUNCH_FKJ
    return launched();
}
