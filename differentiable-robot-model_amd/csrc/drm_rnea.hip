// drm_rnea.hip — K3: recursive Newton-Euler inverse dynamics over the whole tree.
//
// Replaces DifferentiableRobotModel.compute_inverse_dynamics (robot_model.py:305-375: update_kinematic_state
// 139-195 + update_joint_acc rigid_body.py:159-165 + iterative_newton_euler robot_model.py:250-303) and
// compute_non_linear_effects (robot_model.py:377-400, qdd = NULL).
//
// Per sample: in q, qd, qdd [n] (12 n bytes), out tau[n] (4 n bytes).   n = 7: 112 B, ~2.6 kflop.
// 7-DoF arm chains run rnea_arm_kernel (drm_arm_kernels.hip); every other robot the loop-structured kernel below.
#include "drm_common.hpp"
#include "drm_sample.hpp"
#include "drm_tree_dev.hpp"

namespace drm {

// Loop-structured RNEA of any robot (drm_tree.hpp rnea_tree_walk, block layout in drm_tree_dev.hpp): one tile of 64
// samples per block, one wavefront per segment of the walk.
// LDS: [ table ][ tau ]([ q ][ qd ][ qdd ] in the SHORT form) shared, then per wavefront [ body forces : ops * 6 * 64 ]
//      [ motion slots : n_slots * 12 * 64 ][ force slots : n_slots * 6 * 64 ]   (per-wavefront areas: TreeArgs.wave_off)
// SHORT > 0: no segment has more than SHORT ops (the fingers of a hand): drm_tree.hpp rnea_tree_walk_short keeps the per-op
// records in registers, the per-wavefront LDS area holds the save slots only.
constexpr int RNEA_SHORT_OPS = 6;
constexpr int RNEA_FORCE_FLOATS = 6; // what this kernel parks per op between the sweeps: the body force
template <int SHORT>
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    rnea_tree_kernel(TreeArgs a, int flags, const float *__restrict__ q, const float *__restrict__ qd,
                     const float *__restrict__ qdd, int64_t B, float *__restrict__ tau, uint32_t magic_q, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n), region = round4(WAVE * Sq);
    constexpr bool tiles = SHORT > 0; // the short form stages the inputs of a tile in LDS once for its wavefronts (see below)
    float *ltau = smem + table_lds_floats(a.n_ops);
    float *lq = ltau + region, *lqd = lq + region, *lqdd = lqd + region;
    const int first = a.seg_begin[wave], last = a.seg_begin[wave + 1];
    float *park = smem + a.wave_off[wave];
    float *lms = park + (SHORT ? 0 : (last - first) * (RNEA_FORCE_FLOATS * WAVE)); // motion slots [slot][12][64]
    float *lfs = lms + a.n_slots * (12 * WAVE);                     // force slots  [slot][6][64]

    // LDS per sample is what bounds the wavefronts per CU of a big single-segment robot (an arm carrying a hand: one
    // wavefront per CU with the inputs staged and 9 floats parked per op), so: the joint state is read straight from each
    // lane's rows of q / qd / qdd (two or three reads per op, cache hits after the first), only the body force is parked
    // (6 floats per op: cos / sin are recomputed on the way back), and the one LDS tile left is tau's, for a coalesced store.
    // The SHORT form (fingers of a hand, a wavefront each; nothing parked) stages the three input tiles once per block, as
    // before: they are shared, and the form is bounded by registers (96 VGPR = five waves per SIMD), not by LDS.
    const TableLds tab = stage_tree_table(a, smem);
    const bool fast = tc.full && (n & 1);
    if constexpr (tiles) {
        if (wave == 0) tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, fast && (align & AL_Q), tc.full && (align & AL_Q));
        if (wave == (a.n_segments > 1 ? 1 : 0))
            tile_load<0>(qd + tc.b0 * n, tc.rows, n, magic_q, lqd, lane, fast && (align & AL_QD), tc.full && (align & AL_QD));
        if (qdd && wave == (a.n_segments > 2 ? 2 : 0))
            tile_load<0>(qdd + tc.b0 * n, tc.rows, n, magic_q, lqdd, lane, fast && (align & AL_QDD), tc.full && (align & AL_QDD));
    }
    for (int s = 0; s < a.n_slots * 6; ++s) lfs[s * WAVE + lane] = 0.0f;
    __syncthreads();

    // lanes past a partial tile read zeros (not stale LDS): their angles must not be able to push the wave onto
    // the rare large-angle sincos path, which would change the rounding of the live lanes from run to run
    const unsigned row = lane * Sq;
    const bool live = (int)lane < tc.rows;
    const bool has_qdd = qdd != nullptr;
    const TableLds &ctl = tab;
    auto rowf = [&](int k) { return tab.row(k); };
    const int64_t grow = (tc.b0 + (live ? (int64_t)lane : 0)) * n; // (lanes past a partial tile read row 0 of it, then zeros)
    auto qf = [&](int d, float &x, float &v, float &acc) {
        if constexpr (tiles) {
            x = live ? lq[row + d] : 0.0f;
            v = lqd[row + d];
            acc = has_qdd ? lqdd[row + d] : 0.0f;
        } else {
            x = live ? q[grow + d] : 0.0f;
            v = live ? qd[grow + d] : 0.0f;
            acc = (has_qdd && live) ? qdd[grow + d] : 0.0f;
        }
    };
    auto tau_out = [&](int d, float v) { ltau[row + d] = v; };
    auto msave = [&](int s, const Motion &M) { lds_put_motion(lms, s, lane, M); };
    auto mload = [&](int s, Motion &M) { lds_get_motion(lms, s, lane, M); };
    auto fadd = [&](int s, const Force &F) { lds_add_force(lfs, s, lane, F); };
    auto ftake = [&](int s, Force &F) { lds_take_force(lfs, s, lane, F); };
    if constexpr (SHORT > 0) {
        rnea_tree_walk_short<SHORT>(a.prefix_end, first, last, ctl, rowf, flags, qf, tau_out, msave, mload, fadd, ftake);
    } else {
        rnea_tree_walk(
            a.prefix_end, first, last, ctl, rowf, flags, qf, tau_out,
            [&](int k, const Force &F, float, float, float) {
                float *b = park + (k - first) * (RNEA_FORCE_FLOATS * WAVE) + lane;
#pragma unroll
                for (int i = 0; i < 3; ++i) { b[i * WAVE] = F.la[i][0]; b[(3 + i) * WAVE] = F.la[i][1]; }
            },
            [&](int k, Force &F, float &c, float &s, float &x) {
                const float *b = park + (k - first) * (RNEA_FORCE_FLOATS * WAVE) + lane;
#pragma unroll
                for (int i = 0; i < 3; ++i) F.la[i] = f2_make(b[i * WAVE], b[(3 + i) * WAVE]);
                int w0, w1;
                ctl_words(ctl, k, w0, w1);
                const OpCtl ct = decode_ctl(w0, w1);
                x = 0.0f; c = 1.0f; s = 0.0f;
                if (ct.dof >= 0) {
                    float v, acc;
                    qf(ct.dof, x, v, acc);
                    if (!ct.prismatic) sincos_one(x, s, c);
                }
            },
            msave, mload, fadd, ftake);
    }
    __syncthreads();
    if (wave == 0)
        tile_store<0>(tau + tc.b0 * n, tc.rows, n, magic_q, ltau, lane, fast && (align & AL_TAU), tc.full && (align & AL_TAU));
}

// LDS bytes of a launch (fills a.wave_off)
static size_t rnea_tree_lds(TreeArgs &a, bool records_in_registers) {
    // table + the tau tile (+ the three input tiles when several wavefronts share them)
    const size_t shared = (size_t)table_lds_floats(a.n_ops) + (records_in_registers ? 4 : 1) * (size_t)round4(WAVE * pad_odd(a.n));
    return sizeof(float) * layout_waves(a, shared, records_in_registers ? 0 : RNEA_FORCE_FLOATS * WAVE, a.n_slots * 18 * WAVE,
                                        [](int) { return 0; });
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
    const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(tau, AL_TAU);
    hipStream_t s = (hipStream_t)stream;
#ifndef DRM_NO_ARM_KERNEL
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7 && B >= WAVE && B / WAVE < 0x7fffffffLL &&
        align == (AL_Q | AL_QD | AL_TAU | (qdd ? AL_QDD : 0u)) && (((uintptr_t)w->ops_f) & 15u) == 0) {
        // 7-DoF arms: full tiles through the packed-FP32 chain kernel, ragged tail through the generic one
        const int n_tiles = (int)(B / WAVE);
        launch_rnea_arm(w->ops_f, arm_links(w), q, qd, qdd, n_tiles, (int)flags, tau, s);
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
    if (!segments_ok(w)) return fail(DRM_ERR_INVALID, "walk segments are inconsistent");
    if ((((uintptr_t)w->ops_f) & 15u) != 0) return fail(DRM_ERR_INVALID, "ops_f must be 16-byte aligned");
    TreeArgs a = tree_args(w);
    const bool shorts = a.max_seg_ops <= RNEA_SHORT_OPS;
    size_t lds = rnea_tree_lds(a, shorts);
    if (lds > (size_t)MAX_LDS_BYTES && a.n_segments > 1) { // the segments do not fit side by side: one wavefront walks them all
        a = tree_args(w, true);
        lds = rnea_tree_lds(a, false);
    }
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    if (a.max_seg_ops <= RNEA_SHORT_OPS) {
        rc = ensure_lds_tree(rnea_tree_kernel<RNEA_SHORT_OPS>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(rnea_tree_kernel<RNEA_SHORT_OPS>, dim3((unsigned)tiles), dim3(WAVE * a.n_segments), lds, s, a, (int)flags, q, qd,
                           qdd, B, tau, div_magic(n), align);
    } else {
        rc = ensure_lds_tree(rnea_tree_kernel<0>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(rnea_tree_kernel<0>, dim3((unsigned)tiles), dim3(WAVE * a.n_segments), lds, s, a, (int)flags, q, qd, qdd, B,
                           tau, div_magic(n), align);
    }
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
    // a serial 7-DoF arm whose last link is the target.  Two shapes: the tree walk IS the chain (target_op its last op), or
    // the tree walk holds the moving joints only (the host folded the fixed tail into the last of them and built BOTH tables
    // from the folded link table, so the chain's first n rows carry the tree's constants): one table either way.
    const bool same = target_op == tree->n_ops - 1 && tree->n_ops == chain->n_ops;
    const bool folded = tree->n_ops == n && chain->n_ops > n && (chain->shape & DRM_WALK_ARM_CHAIN) && chain->capacity == 8;
    const float *table = same ? tree->ops_f : chain->ops_f;
    if ((tree->shape & DRM_WALK_ARM_CHAIN) && tree->capacity == 8 && n == 7 && (same || folded) &&
        chain->target_perm == 2 && B >= WAVE && B / WAVE < 0x7fffffffLL && (((uintptr_t)table) & 15u) == 0 &&
        align == (AL_Q | AL_QD | AL_TAU | AL_POS | AL_QUAT | (qdd ? AL_QDD : 0u))) {
        const int n_tiles = (int)(B / WAVE);
        launch_fk_rnea_arm(table, same ? arm_links(tree) : n, q, qd, qdd, n_tiles, (int)flags, tau, pos, quat, s);
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
