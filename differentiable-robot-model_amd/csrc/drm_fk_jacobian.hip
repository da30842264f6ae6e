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
#include "drm_tree_dev.hpp"

namespace drm {

// Loop-structured FK + Jacobian of one chain of any robot (drm_tree.hpp fk_jacobian_tree_walk): one wavefront per tile.
// The walk leaves the world axis z_k of every moving joint in the ang_jac tile and its origin p_k in the lin_jac tile
// (same positions); once the end position is known a second loop over the chain turns p_k into z_k x (p_e - p_k)
// (robot_model.py:661; a prismatic joint's column is (z_k, 0)).  Columns of DoFs off the chain stay zero.
// LDS: [ table ][ q : 64 (n|1) ][ pos : 64 x 3 ][ lin : 64 (3n|1) ][ ang : 64 (3n|1) ]
__global__ void __launch_bounds__(WAVE)
    fk_jacobian_tree_kernel(TreeArgs a, uint64_t dof_mask, const float *__restrict__ q, int64_t B, float *__restrict__ pos,
                            float *__restrict__ quat, float *__restrict__ lin, float *__restrict__ ang, uint32_t magic_q,
                            uint32_t magic_j, uint32_t align, int target_perm) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n), Sj = pad_odd(3 * n);
    float *lq = smem + table_lds_floats(a.n_ops);
    float *lp = lq + round4(WAVE * Sq);
    float *ll = lp + round4(WAVE * 3);
    float *la = ll + round4(WAVE * Sj);

    const TableLds tab = stage_tree_table(a, smem);
    tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, tc.full && (n & 1) && (align & AL_Q), tc.full && (align & AL_Q));
    wave_lds_sync();

    // lanes beyond the last valid row of a partial tile compute garbage that is never stored; they read zero angles
    const bool live = (int)lane < tc.rows;
    const float *qrow = lq + lane * Sq;
    float *lrow = ll + lane * Sj, *arow = la + lane * Sj;
    const uint64_t all = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    if ((dof_mask & all) != all) {
        for (int d = 0; d < n; ++d)
            if (!((dof_mask >> d) & 1ull)) {
                lrow[d] = 0.0f; lrow[n + d] = 0.0f; lrow[2 * n + d] = 0.0f;
                arow[d] = 0.0f; arow[n + d] = 0.0f; arow[2 * n + d] = 0.0f;
            }
    }
    const TableLds &ctl = tab;
    PoseP ee;
    fk_jacobian_tree_walk(a.n_ops, ctl, [&](int k) { return tab.row(k); }, [&](int d) -> float { return live ? qrow[d] : 0.0f; }, ee,
                          [&](int d, const float *z, const float *p, bool) {
                              arow[d] = z[0]; arow[n + d] = z[1]; arow[2 * n + d] = z[2];
                              lrow[d] = p[0]; lrow[n + d] = p[1]; lrow[2 * n + d] = p[2];
                          });
    Pose E;
    pose_from_pairs(ee, E);
#pragma unroll 1
    for (int k = 0; k < a.n_ops; ++k) {
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        if (ct.padding || ct.dof < 0) continue;
        const int d = ct.dof;
        const float z[3] = {arow[d], arow[n + d], arow[2 * n + d]};
        if (ct.prismatic) {
            lrow[d] = z[0]; lrow[n + d] = z[1]; lrow[2 * n + d] = z[2];
            arow[d] = 0.0f; arow[n + d] = 0.0f; arow[2 * n + d] = 0.0f;
        } else {
            const float dp[3] = {E.p[0] - lrow[d], E.p[1] - lrow[n + d], E.p[2] - lrow[2 * n + d]};
            float c[3];
            cross3(z, dp, c);
            lrow[d] = c[0]; lrow[n + d] = c[1]; lrow[2 * n + d] = c[2];
        }
    }
    if (pos) { lp[lane * 3 + 0] = E.p[0]; lp[lane * 3 + 1] = E.p[1]; lp[lane * 3 + 2] = E.p[2]; }
    wave_lds_sync();
    const bool fast_j = tc.full, odd_j = (3 * n) & 1;
    tile_store<0>(ang + tc.b0 * 3 * n, tc.rows, 3 * n, magic_j, la, lane, fast_j && odd_j && (align & AL_ANG), fast_j && (align & AL_ANG));
    tile_store<0>(lin + tc.b0 * 3 * n, tc.rows, 3 * n, magic_j, ll, lane, fast_j && odd_j && (align & AL_LIN), fast_j && (align & AL_LIN));
    if (pos) tile_store<3>(pos + tc.b0 * 3, tc.rows, 3, 0u, lp, lane, tc.full && (align & AL_POS));
    // ---- quat [B,4]: one 16-byte store per lane is already coalesced -------------------
    if (quat && live) {
        // the target is the last REAL op: undo its column permutation before the quaternion
        float qt[4];
        unpermute(target_perm, E.R);
        quat_xyzw(E.R, qt);
        float *dst = quat + (tc.b0 + lane) * 4;
        if (align & AL_QUAT) {
            store16_wt(dst, make_float4(qt[0], qt[1], qt[2], qt[3]));
        } else {
            dst[0] = qt[0]; dst[1] = qt[1]; dst[2] = qt[2]; dst[3] = qt[3];
        }
    }
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
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT) | al16(lin_jac, AL_LIN) |
                           al16(ang_jac, AL_ANG);
    hipStream_t s = (hipStream_t)stream;
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
    } else {
        // any other serial chain of up to 16 ops (an arm with its gripper up to a fingertip, a finger of a hand): full tiles
        // through the straight-line chain kernel, a ragged tail through the loop-structured one below
        const int64_t done = launch_chain_fk_jacobian(w, q, B, pos, quat, lin_jac, ang_jac, s);
        if (done > 0) {
            rc = launched();
            if (rc || done == B) return rc;
            drm_walk generic = *w;
            generic.shape &= ~DRM_WALK_SERIAL_CHAIN;
            return drm_fk_jacobian(&generic, q + done * n, B - done, pos + done * 3, quat + done * 4, lin_jac + done * 3 * n,
                                   ang_jac + done * 3 * n, stream);
        }
        if ((((uintptr_t)w->ops_f) & 15u) != 0) return fail(DRM_ERR_INVALID, "ops_f must be 16-byte aligned");
        TreeArgs a = tree_args(w);
        a.n_segments = 1; a.prefix_end = 0;
        const size_t lds = sizeof(float) * (size_t)(table_lds_floats(a.n_ops) + round4(WAVE * pad_odd(n)) + round4(WAVE * 3) +
                                                    2 * round4(WAVE * pad_odd(3 * n)));
        rc = ensure_lds_tree(fk_jacobian_tree_kernel, lds);
        if (rc) return rc;
        const int64_t tiles = (B + WAVE - 1) / WAVE;
        if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
        hipLaunchKernelGGL(fk_jacobian_tree_kernel, dim3((unsigned)tiles), dim3(WAVE), lds, s, a, w->dof_mask, q, B, pos, quat, lin_jac,
                           ang_jac, div_magic(n), div_magic(3 * n), align, (int)w->target_perm);
    }
    return launched();
}
