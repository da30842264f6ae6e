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

// Robots whose segments are all SHORT (the fingers of a hand, at most RNEA_SHORT_OPS ops each): one tile of 64 samples per
// block, one wavefront per segment (block layout in drm_tree_dev.hpp); drm_tree.hpp rnea_tree_walk_short keeps the per-op
// records in registers, so the per-wavefront LDS area holds the save slots only and the form is bounded by registers (96 VGPR =
// five waves per SIMD), not by LDS.
// LDS: [ table ][ tau ][ q ][ qd ][ qdd ] shared (staged once per block with coalesced 16-byte accesses), then per wavefront
//      [ motion slots : n_slots * 12 * 64 ][ force slots : n_slots * 6 * 64 ]   (per-wavefront areas: TreeArgs.wave_off)
constexpr int RNEA_SHORT_OPS = 6;
constexpr int RNEA_FORCE_FLOATS = 6; // what the long form keeps per op between the sweeps: the body force
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    rnea_tree_kernel(TreeArgs a, int flags, const float *__restrict__ q, const float *__restrict__ qd,
                     const float *__restrict__ qdd, int64_t B, float *__restrict__ tau, uint32_t magic_q, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n), region = round4(WAVE * Sq);
    float *ltau = smem + table_lds_floats(a.n_ops);
    float *lq = ltau + region, *lqd = lq + region, *lqdd = lqd + region;
    const int first = a.seg_begin[wave], last = a.seg_begin[wave + 1];
    float *lms = smem + a.wave_off[wave];          // motion slots [slot][12][64]
    float *lfs = lms + a.n_slots * (12 * WAVE);    // force slots  [slot][6][64]

    const TableLds tab = stage_tree_table(a, smem);
    const bool fast = tc.full && (n & 1);
    if (wave == 0) tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, fast && (align & AL_Q), tc.full && (align & AL_Q));
    if (wave == (a.n_segments > 1 ? 1 : 0))
        tile_load<0>(qd + tc.b0 * n, tc.rows, n, magic_q, lqd, lane, fast && (align & AL_QD), tc.full && (align & AL_QD));
    if (qdd && wave == (a.n_segments > 2 ? 2 : 0))
        tile_load<0>(qdd + tc.b0 * n, tc.rows, n, magic_q, lqdd, lane, fast && (align & AL_QDD), tc.full && (align & AL_QDD));
    for (int s = 0; s < a.n_slots * 6; ++s) lfs[s * WAVE + lane] = 0.0f;
    __syncthreads();

    // lanes past a partial tile read zeros (not stale LDS): their angles must not be able to push the wave onto
    // the rare large-angle sincos path, which would change the rounding of the live lanes from run to run
    const unsigned row = lane * Sq;
    const bool live = (int)lane < tc.rows;
    const bool has_qdd = qdd != nullptr;
    const TableLds &ctl = tab;
    auto rowf = [&](int k) { return tab.row(k); };
    auto qf = [&](int d, float &x, float &v, float &acc) {
        x = live ? lq[row + d] : 0.0f;
        v = lqd[row + d];
        acc = has_qdd ? lqdd[row + d] : 0.0f;
    };
    auto tau_out = [&](int d, float v) { ltau[row + d] = v; };
    auto msave = [&](int s, const Motion &M) { lds_put_motion(lms, s, lane, M); };
    auto mload = [&](int s, Motion &M) { lds_get_motion(lms, s, lane, M); };
    auto fadd = [&](int s, const Force &F) { lds_add_force(lfs, s, lane, F); };
    auto ftake = [&](int s, Force &F) { lds_take_force(lfs, s, lane, F); };
    rnea_tree_walk_short<RNEA_SHORT_OPS>(a.prefix_end, first, last, ctl, rowf, flags, qf, tau_out, msave, mload, fadd, ftake);
    __syncthreads();
    if (wave == 0)
        tile_store<0>(tau + tc.b0 * n, tc.rows, n, magic_q, ltau, lane, fast && (align & AL_TAU), tc.full && (align & AL_TAU));
}

// LDS bytes of that launch (0: does not apply — a segment is too long, or the hand does not fit)
static size_t rnea_short_plan(const drm_walk *w, TreeArgs &a) {
    a = tree_args(w, false);
    if (a.max_seg_ops > RNEA_SHORT_OPS) return 0;
    const size_t shared = (size_t)table_lds_floats(a.n_ops) + 4 * (size_t)round4(WAVE * pad_odd(a.n));
    const size_t lds = sizeof(float) * layout_waves(a, shared, 0, a.n_slots * 18 * WAVE, [](int) { return 0; });
    return lds <= (size_t)MAX_LDS_BYTES ? lds : 0;
}

// Every other robot (an arm carrying a gripper or a hand, a mobile manipulator: a segment of more than RNEA_SHORT_OPS ops):
// the loop form of the walk (drm_tree.hpp rnea_tree_walk), one wavefront per segment, on the pattern of the articulated-body
// kernel (drm_forward_dynamics.hip).  The body forces the backward sweep needs again — 6 floats per link and sample, which in
// LDS left an arm with a hand three wavefronts per CU — live in HBM scratch, [op][6][64] per block (coalesced; fetched one op
// ahead of their use); the grid is PERSISTENT, so the scratch is sized by what the chip holds at once and stays in L2; q / qd /
// qdd are staged per tile with coalesced 16-byte loads (read straight from global memory, one 4-byte access per lane and DoF,
// they thrash a CU's L1 at this occupancy) and tau leaves through the tile qdd came in by.
// LDS: [ table ][ q ][ qd ][ qdd -> tau ] shared, then per wavefront [ motion slots : n_slots * 12 * 64 ][ force slots : n_slots * 6 * 64 ]
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    rnea_records_kernel(TreeArgs a, int flags, const float *__restrict__ q, const float *__restrict__ qd,
                        const float *__restrict__ qdd, int64_t B, int n_tiles, float *__restrict__ tau, float *__restrict__ scratch,
                        uint32_t magic_q, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n), region = round4(WAVE * Sq);
    float *lq = smem + table_lds_floats(a.n_ops), *lqd = lq + region, *lt = lqd + region; // lt: qdd in, tau out
    const int first = a.seg_begin[wave], last = a.seg_begin[wave + 1];
    float *lms = smem + a.wave_off[wave];          // motion slots [slot][12][64]
    float *lfs = lms + a.n_slots * (12 * WAVE);    // force slots  [slot][6][64]
    float *recs = scratch + ((int64_t)blockIdx.x * a.n_ops + first) * (RNEA_FORCE_FLOATS * WAVE) + lane;
    const TableLds tab = stage_tree_table(a, smem);
    for (int s = 0; s < a.n_slots * 6; ++s) lfs[s * WAVE + lane] = 0.0f; // (every take leaves its slot at zero again)
    const unsigned row = lane * Sq;
    const bool has_qdd = qdd != nullptr;
    const TableLds &ctl = tab;
    auto rowf = [&](int k) { return tab.row(k); };
    auto rec = [&](int k) -> float * { return recs + (k - first) * (RNEA_FORCE_FLOATS * WAVE); };

#pragma unroll 1
    for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
        __syncthreads(); // the table is staged / the previous tile's torques have left the tile
        const int64_t b0 = (int64_t)tile * WAVE;
        const int rows = B - b0 < WAVE ? (int)(B - b0) : WAVE;
        const bool full = rows == WAVE, fast = full && (n & 1);
        if (wave == 0) tile_load<0>(q + b0 * n, rows, n, magic_q, lq, lane, fast && (align & AL_Q), full && (align & AL_Q));
        if (wave == (a.n_segments > 1 ? 1 : 0))
            tile_load<0>(qd + b0 * n, rows, n, magic_q, lqd, lane, fast && (align & AL_QD), full && (align & AL_QD));
        if (has_qdd && wave == (a.n_segments > 2 ? 2 : 0))
            tile_load<0>(qdd + b0 * n, rows, n, magic_q, lt, lane, fast && (align & AL_QDD), full && (align & AL_QDD));
        __syncthreads();
        // lanes past a partial tile walk a robot at rest (zeros, not stale LDS: see rnea_tree_kernel); nothing of theirs is stored
        const bool live = (int)lane < rows;
        auto qf = [&](int d, float &x, float &v, float &acc) {
            x = live ? lq[row + d] : 0.0f;
            v = live ? lqd[row + d] : 0.0f;
            acc = (has_qdd && live) ? lt[row + d] : 0.0f;
        };
        rnea_tree_walk(
            a.prefix_end, first, last, ctl, rowf, flags, qf, [&](int d, float v) { lt[row + d] = v; },
            [&](int k, const Force &F, float, float, float) {
                float *b = rec(k);
#pragma unroll
                for (int i = 0; i < 3; ++i) { b[i * WAVE] = F.la[i][0]; b[(3 + i) * WAVE] = F.la[i][1]; }
            },
            [&](int k, Force &F) {
                const float *b = rec(k);
#pragma unroll
                for (int i = 0; i < 3; ++i) F.la[i] = f2_make(b[i * WAVE], b[(3 + i) * WAVE]);
            },
            [&](int k, float &c, float &s, float &x) {
                int w0, w1;
                ctl_words(ctl, k, w0, w1);
                const OpCtl ct = decode_ctl(w0, w1);
                x = 0.0f; c = 1.0f; s = 0.0f;
                if (ct.dof >= 0) {
                    x = live ? lq[row + ct.dof] : 0.0f;
                    if (!ct.prismatic) sincos_one(x, s, c);
                }
            },
            [&](int s, const Motion &M) { lds_put_motion(lms, s, lane, M); }, [&](int s, Motion &M) { lds_get_motion(lms, s, lane, M); },
            [&](int s, const Force &F) { lds_add_force(lfs, s, lane, F); }, [&](int s, Force &F) { lds_take_force(lfs, s, lane, F); });
        __syncthreads();
        if (wave == 0) tile_store<0>(tau + b0 * n, rows, n, magic_q, lt, lane, fast && (align & AL_TAU), full && (align & AL_TAU));
    }
}

// geometry of that launch: LDS bytes per block and the number of blocks the device holds at once (the persistent grid)
struct RneaRecordsPlan {
    TreeArgs a;
    size_t lds;
    int resident;
};
static int rnea_records_plan(const drm_walk *w, RneaRecordsPlan &p) {
    // one wavefront per segment; all the segments through one wavefront when their save slots do not fit side by side
    for (int single = segments_worth_fanning_out(w) ? 0 : 1; single < 2; ++single) {
        p.a = tree_args(w, single != 0);
        const size_t shared = (size_t)table_lds_floats(p.a.n_ops) + 3 * (size_t)round4(WAVE * pad_odd(p.a.n));
        p.lds = sizeof(float) * layout_waves(p.a, shared, 0, p.a.n_slots * 18 * WAVE, [](int) { return 0; });
        if (p.lds <= (size_t)MAX_LDS_BYTES || w->n_segments <= 1) break;
    }
    int rc = ensure_lds_tree(rnea_records_kernel, p.lds);
    if (rc) return rc;
    return resident_blocks(rnea_records_kernel, WAVE * p.a.n_segments, p.lds, p.resident);
}

} // namespace drm

using namespace drm;

static int64_t drm_rnea_scratch_floats_impl(const drm_walk *w, int64_t B, bool aligned) {
    if (check_walk(w) || B <= 0 || !segments_ok(w)) return 0;
    // (7-DoF arms / arms with a hand: full aligned tiles run straight-line kernels without scratch; sized for the ragged tail and
    // for a misaligned call, drm_common.hpp fast_path_scratch_tiles)
    // (a walk with its own straight-line kernel, drm_walk.special: no alignment condition at all — only the ragged tail is sized)
    if (w->special[DRM_SPECIAL_RNEA]) aligned = true;
    const bool fast = ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && w->n_dofs == 7) || arm_hand_compiled(w) ||
                      w->special[DRM_SPECIAL_RNEA] != nullptr;
    TreeArgs a;
    if (rnea_short_plan(w, a)) return 0;
    RneaRecordsPlan p;
    if (rnea_records_plan(w, p)) return 0;
    int64_t tiles = fast ? (aligned ? (B % WAVE ? 1 : 0) : fast_path_scratch_tiles(B)) : (B + WAVE - 1) / WAVE;
    return (tiles < p.resident ? tiles : (int64_t)p.resident) * p.a.n_ops * RNEA_FORCE_FLOATS * WAVE;
}
extern "C" int64_t drm_rnea_scratch_floats(const drm_walk *w, int64_t B) { return drm_rnea_scratch_floats_impl(w, B, false); }
// ... for a caller that GUARANTEES 16-byte aligned q / qd / qdd (f) / outputs (both Python bindings do: they clone a misaligned
// slice): the full tiles of a 7-DoF arm / an arm with a hand then run straight-line kernels that need no scratch — only a ragged
// tail's one tile is sized
extern "C" int64_t drm_rnea_scratch_floats_aligned(const drm_walk *w, int64_t B) { return drm_rnea_scratch_floats_impl(w, B, true); }

extern "C" int drm_rnea(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags,
                        float *tau, float *scratch, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !qd || !tau) return fail(DRM_ERR_INVALID, "q / qd / tau must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs;
    const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(tau, AL_TAU);
    hipStream_t s = (hipStream_t)stream;
    if (w->special[DRM_SPECIAL_RNEA] && B >= WAVE && B / WAVE < 0x7fffffffLL && (((uintptr_t)w->ops_f) & 15u) == 0) {
        // a per-robot straight-line kernel built for exactly this walk (csrc/drm_static.hpp, specialize.py): full tiles, any
        // pointer alignment, no scratch; the ragged tail through the kernels below
        int n_tiles = (int)(B / WAVE), fl = (int)flags;
        uint32_t magic = div_magic(n), al = align;
        void *args[] = {(void *)&w->ops_f, (void *)&q, (void *)&qd, (void *)&qdd, (void *)&n_tiles, (void *)&fl, (void *)&tau, (void *)&magic, (void *)&al};
        int grid = 0; // persistent wavefronts: what the device holds at once (each reads its next tile's rows while it walks one)
        rc = resident_blocks_module((hipFunction_t)w->special[DRM_SPECIAL_RNEA], WAVE, grid);
        if (rc) return rc;
        if (grid > n_tiles || w->n_ops < STATIC_LONE_OPS) grid = n_tiles; // (small robots: one tile per block, drm_common.hpp STATIC_LONE_OPS)
        hipError_t e = hipModuleLaunchKernel((hipFunction_t)w->special[DRM_SPECIAL_RNEA], (unsigned)grid, 1, 1, WAVE, 1, 1, 0, s, args, nullptr);
        if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_rnea_static): %s", hipGetErrorString(e));
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done == B) return DRM_OK;
        drm_walk rest = *w;
        rest.special[DRM_SPECIAL_RNEA] = nullptr;
        return drm_rnea(&rest, q + done * n, qd + done * n, qdd ? qdd + done * n : nullptr, B - done, flags, tau + done * n, scratch, stream);
    }
#ifndef DRM_NO_ARM_KERNEL
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7 && B >= WAVE && B / WAVE < 0x7fffffffLL &&
        align == (AL_Q | AL_QD | AL_TAU | (qdd ? AL_QDD : 0u)) && (((uintptr_t)w->ops_f) & 15u) == 0) {
        // 7-DoF arms: full tiles through the packed-FP32 chain kernel, ragged tail through the generic one
        if (w->special[DRM_SPECIAL_RNEA_ARM] && B / (2 * WAVE) >= DRM_ARM_STATIC_MIN_PAIRS) {
            // this arm's own kernel, its constants folded into the instruction stream (csrc/drm_arm_stream.hpp, specialize.py): the
            // 128-row pairs of tiles; an odd tile and the ragged tail through the kernels below
            int n_pairs = (int)(B / (2 * WAVE)), fl = (int)flags;
            void *args[] = {(void *)&q, (void *)&qd, (void *)&qdd, (void *)&n_pairs, (void *)&fl, (void *)&tau};
            hipError_t e = hipModuleLaunchKernel((hipFunction_t)w->special[DRM_SPECIAL_RNEA_ARM], (unsigned)arm_stream_grid(n_pairs), 1, 1, WAVE, 1, 1, 0, s, args, nullptr);
            if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_rnea_arm_static): %s", hipGetErrorString(e));
            const int64_t done = (int64_t)n_pairs * 2 * WAVE;
            if (done == B) return DRM_OK;
            drm_walk rest = *w;
            rest.special[DRM_SPECIAL_RNEA_ARM] = nullptr;
            return drm_rnea(&rest, q + done * n, qd + done * n, qdd ? qdd + done * n : nullptr, B - done, flags, tau + done * n, scratch, stream);
        }
        const int n_tiles = (int)(B / WAVE);
        launch_rnea_arm(w->ops_f, arm_links(w), q, qd, qdd, n_tiles, (int)flags, tau, s);
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done == B) return launched();
        rc = launched();
        if (rc) return rc;
        drm_walk generic = *w;
        generic.shape &= ~DRM_WALK_ARM_CHAIN;
        return drm_rnea(&generic, q + done * n, qd + done * n, qdd ? qdd + done * n : nullptr, B - done, flags,
                        tau + done * n, scratch, stream);
    }
#endif
    {   // a hand (fingers off the root): full 128-row tiles through the two-samples-per-lane finger kernel
        const int64_t done = launch_rnea_fingers(w, q, qd, qdd, B, (int)flags, tau, s);
        if (done > 0) {
            rc = launched();
            if (rc || done == B) return rc;
            drm_walk generic = *w;
            generic.shape &= ~DRM_WALK_FINGERS;
            return drm_rnea(&generic, q + done * n, qd + done * n, qdd ? qdd + done * n : nullptr, B - done, flags, tau + done * n,
                            scratch, stream);
        }
    }
    {   // an arm that carries a hand (Panda with gripper, Jaco, iiwa7 + Allegro): full tiles through the straight-line kernel
        const int64_t done = launch_rnea_arm_hand(w, q, qd, qdd, B, (int)flags, tau, s);
        if (done > 0) {
            rc = launched();
            if (rc || done == B) return rc;
            drm_walk generic = *w;
            generic.shape &= ~DRM_WALK_ARM_HAND;
            return drm_rnea(&generic, q + done * n, qd + done * n, qdd ? qdd + done * n : nullptr, B - done, flags, tau + done * n,
                            scratch, stream);
        }
    }
    // A walk whose full tiles would have run a straight-line kernel got here because its pointers are not 16-byte aligned (or as
    // the ragged tail of such a call): its scratch holds fast_path_scratch_tiles(B) tiles of the loop kernel — the persistent grid
    // below is held to that many blocks.
    const bool fast_walk = (((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7)) || arm_hand_compiled(w);
    if (!segments_ok(w)) return fail(DRM_ERR_INVALID, "walk segments are inconsistent");
    if ((((uintptr_t)w->ops_f) & 15u) != 0) return fail(DRM_ERR_INVALID, "ops_f must be 16-byte aligned");
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    TreeArgs fingers;
    if (const size_t lds = rnea_short_plan(w, fingers)) {
        rc = ensure_lds_tree(rnea_tree_kernel, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(rnea_tree_kernel, dim3((unsigned)tiles), dim3(WAVE * fingers.n_segments), lds, s, fingers, (int)flags, q, qd, qdd, B,
                           tau, div_magic(n), align);
        return launched();
    }
    RneaRecordsPlan p;
    rc = rnea_records_plan(w, p);
    if (rc) return rc;
    if (!scratch)
        return fail(DRM_ERR_INVALID, "this robot's inverse dynamics keeps its per-link records in scratch: pass drm_rnea_scratch_floats() floats");
    int64_t grid = tiles < p.resident ? tiles : (int64_t)p.resident;
    if (fast_walk && grid > MISALIGNED_TILES) grid = MISALIGNED_TILES;
    hipLaunchKernelGGL(rnea_records_kernel, dim3((unsigned)grid), dim3(WAVE * p.a.n_segments), p.lds, s, p.a, (int)flags, q, qd, qdd, B,
                       (int)tiles, tau, scratch, div_magic(n), align);
    return launched();
}

extern "C" int drm_fk_rnea(const drm_walk *tree, const drm_walk *chain, int32_t target_op, const float *q, const float *qd,
                           const float *qdd, int64_t B, int32_t flags, float *tau, float *pos, float *quat, float *scratch,
                           void *stream) {
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
    // the tree walk holds the moving joints only (the host folded the fixed tail into the last of them).  Either way the
    // dynamics rows come from the TREE walk's table and only the fixed tail the FK chain still walks from the CHAIN walk's.
    const bool same = target_op == tree->n_ops - 1 && tree->n_ops == chain->n_ops;
    const bool folded = tree->n_ops == n && chain->n_ops > n && (chain->shape & DRM_WALK_ARM_CHAIN) && chain->capacity == 8;
    if ((tree->shape & DRM_WALK_ARM_CHAIN) && tree->capacity == 8 && n == 7 && (same || folded) &&
        chain->target_perm == 2 && B >= WAVE && B / WAVE < 0x7fffffffLL && (((uintptr_t)tree->ops_f) & 15u) == 0 &&
        (((uintptr_t)chain->ops_f) & 15u) == 0 &&
        align == (AL_Q | AL_QD | AL_TAU | AL_POS | AL_QUAT | (qdd ? AL_QDD : 0u))) {
        if (tree->special[DRM_SPECIAL_FK_RNEA_ARM] && tree->special[DRM_SPECIAL_FK_RNEA_ARM] == chain->special[DRM_SPECIAL_FK_RNEA_ARM] &&
            B / (2 * WAVE) >= DRM_ARM_STATIC_MIN_PAIRS) {
            // the constant-folded fused kernel built for exactly this (dynamics walk, target chain) pair: the same handle on both
            int n_pairs = (int)(B / (2 * WAVE)), fl = (int)flags;
            void *args[] = {(void *)&q, (void *)&qd, (void *)&qdd, (void *)&n_pairs, (void *)&fl, (void *)&tau, (void *)&pos, (void *)&quat};
            hipError_t e = hipModuleLaunchKernel((hipFunction_t)tree->special[DRM_SPECIAL_FK_RNEA_ARM], (unsigned)arm_stream_grid(n_pairs), 1, 1, WAVE, 1, 1, 0, s, args, nullptr);
            if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_fk_rnea_arm_static): %s", hipGetErrorString(e));
            const int64_t done = (int64_t)n_pairs * 2 * WAVE;
            if (done == B) return DRM_OK;
            drm_walk t2 = *tree, c2 = *chain;
            t2.special[DRM_SPECIAL_FK_RNEA_ARM] = nullptr;
            t2.special[DRM_SPECIAL_RNEA_ARM] = nullptr; // (a tail of less than a pair of tiles never reaches it anyway)
            c2.special[DRM_SPECIAL_FK_RNEA_ARM] = nullptr;
            return drm_fk_rnea(&t2, &c2, target_op, q + done * n, qd + done * n, qdd ? qdd + done * n : nullptr, B - done, flags, tau + done * n,
                               pos + done * 3, quat + done * 4, scratch, stream);
        }
        const int n_tiles = (int)(B / WAVE);
        launch_fk_rnea_arm(tree->ops_f, chain->ops_f, same ? arm_links(tree) : n, q, qd, qdd, n_tiles, (int)flags, tau, pos, quat, s);
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
    return drm_rnea(tree, q, qd, qdd, B, flags, tau, scratch, stream);
}

// ABI 11: drm_fk_rnea with a one-sided gather of its outputs (include/drm_hip.h drm_put).  The in-kernel form: the arm's own fused
// kernel with the destinations in its argument block (csrc/drm_arm_stream.hpp PUT) covers the 128-row tile pairs of a launch of at
// least DRM_ARM_STATIC_MIN_PAIRS pairs; everything else is computed by drm_fk_rnea and copied on the stream.
extern "C" int drm_fk_rnea_put(const drm_walk *tree, const drm_walk *chain, int32_t target_op, const float *q, const float *qd,
                               const float *qdd, int64_t B, int32_t flags, float *tau, float *pos, float *quat, float *scratch,
                               const drm_put *put, void *stream) {
    if (!put || put->n_peers == 0) return drm_fk_rnea(tree, chain, target_op, q, qd, qdd, B, flags, tau, pos, quat, scratch, stream);
    if (put->n_peers < 0 || put->n_peers > DRM_MAX_PEERS || put->row_offset < 0)
        return fail(DRM_ERR_INVALID, "drm_put: n_peers must be 0 .. DRM_MAX_PEERS and row_offset >= 0");
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
    int64_t done = 0;
#ifndef DRM_NO_ARM_KERNEL
    {
        uintptr_t dst = 0;
        for (int p = 0; p < put->n_peers; ++p) dst |= (uintptr_t)put->tau[p] | (uintptr_t)put->pos[p] | (uintptr_t)put->quat[p];
        const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(tau, AL_TAU) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
        const bool same = target_op == tree->n_ops - 1 && tree->n_ops == chain->n_ops;
        const bool folded = tree->n_ops == n && chain->n_ops > n && (chain->shape & DRM_WALK_ARM_CHAIN) && chain->capacity == 8;
        const void *own = tree->special[DRM_SPECIAL_FK_RNEA_ARM_PUT];
        if (own && own == chain->special[DRM_SPECIAL_FK_RNEA_ARM_PUT] && (tree->shape & DRM_WALK_ARM_CHAIN) && tree->capacity == 8 && n == 7 &&
            (same || folded) && chain->target_perm == 2 && B / (2 * WAVE) >= DRM_ARM_STATIC_MIN_PAIRS && B / WAVE < 0x7fffffffLL &&
            align == (AL_Q | AL_QD | AL_TAU | AL_POS | AL_QUAT | (qdd ? AL_QDD : 0u)) && (dst & 15u) == 0 && (put->row_offset & 3) == 0) {
            int n_pairs = (int)(B / (2 * WAVE)), fl = (int)flags;
            drm_put dsts = *put;
            void *args[] = {(void *)&q, (void *)&qd, (void *)&qdd, (void *)&n_pairs, (void *)&fl, (void *)&tau, (void *)&pos, (void *)&quat, (void *)&dsts};
            hipError_t e = hipModuleLaunchKernel((hipFunction_t)own, (unsigned)arm_stream_grid(n_pairs), 1, 1, WAVE, 1, 1, 0, s, args, nullptr);
            if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_fk_rnea_arm_put_static): %s", hipGetErrorString(e));
            done = (int64_t)n_pairs * 2 * WAVE;
            if (done == B) return DRM_OK;
        }
    }
#endif
    // what is left (everything, for walks without the in-kernel form): compute, then one copy per destination and array
    const float *rq = q + done * n, *rqd = qd + done * n, *rqdd = qdd ? qdd + done * n : nullptr;
    float *rtau = tau + done * n, *rpos = pos + done * 3, *rquat = quat + done * 4;
    const int64_t rest = B - done, at = put->row_offset + done;
    drm_walk t2 = *tree, c2 = *chain;
    rc = drm_fk_rnea(&t2, &c2, target_op, rq, rqd, rqdd, rest, flags, rtau, rpos, rquat, scratch, stream);
    if (rc) return rc;
    for (int p = 0; p < put->n_peers; ++p) {
        hipError_t e = hipSuccess;
        if (put->tau[p]) e = hipMemcpyAsync(put->tau[p] + at * n, rtau, sizeof(float) * (size_t)rest * n, hipMemcpyDefault, s);
        if (e == hipSuccess && put->pos[p]) e = hipMemcpyAsync(put->pos[p] + at * 3, rpos, sizeof(float) * (size_t)rest * 3, hipMemcpyDefault, s);
        if (e == hipSuccess && put->quat[p]) e = hipMemcpyAsync(put->quat[p] + at * 4, rquat, sizeof(float) * (size_t)rest * 4, hipMemcpyDefault, s);
        if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipMemcpyAsync to a gather destination: %s", hipGetErrorString(e));
    }
    return DRM_OK;
}
