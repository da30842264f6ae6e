// drm_fk.hip — K1/K4: FK of T target links over a (possibly branching) walk.
//
// Replaces DifferentiableRobotModel.compute_forward_kinematics (robot_model.py:223-248) and, with
// T = all links, compute_forward_kinematics_all_links (robot_model.py:197-221; recursion rigid_body.py:85-127).
//
// Per sample: in q[n] (4 n bytes), out pos[T,3] quat[T,4] (28 T bytes).
#include "drm_common.hpp"
#include "drm_sample.hpp"
#include "drm_tree_dev.hpp"

namespace drm {

// Loop-structured FK of T targets over a (possibly branching) walk (drm_tree.hpp fk_tree_walk): one wavefront per tile.
// LDS: [ table ][ q : 64 (n|1) ][ pos : 64 (3T|1) ][ quat : 64 (4T+1) ][ slots : n_slots * 12 * 64 ]
__global__ void __launch_bounds__(WAVE)
    fk_tree_kernel(TreeArgs a, const float *__restrict__ q, int64_t B, int T, float *__restrict__ pos, float *__restrict__ quat,
                   uint32_t magic_q, uint32_t magic_p, uint32_t magic_r, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n), Sp = pad_odd(3 * T), Sr = pad_odd(4 * T);
    float *lq = smem + table_lds_floats(a.n_ops);
    float *lp = lq + round4(WAVE * Sq);
    float *lr = lp + round4(WAVE * Sp);
    float *ls = lr + round4(WAVE * Sr); // save slots: [slot][12][64]

    const TableLds tab = stage_tree_table(a, smem);
    tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, tc.full && (n & 1) && (align & AL_Q), tc.full && (align & AL_Q));
    wave_lds_sync();

    // lanes past a partial tile read zeros (not stale LDS): their angles must not be able to push the wave onto
    // the rare large-angle sincos path, which would change the rounding of the live lanes from run to run
    const bool live = (int)lane < tc.rows;
    const float *qrow = lq + lane * Sq;
    float *prow = lp + lane * Sp;
    float *rrow = lr + lane * Sr;
    auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; };
    auto ssave = [&](int s, const PoseP &P) { lds_put_pose(ls, s, lane, P); };
    auto sload = [&](int s, PoseP &P) { lds_get_pose(ls, s, lane, P); };
    auto emit = [&](int t, const float *p, const float *qt) {
        prow[t * 3 + 0] = p[0]; prow[t * 3 + 1] = p[1]; prow[t * 3 + 2] = p[2];
        rrow[t * 4 + 0] = qt[0]; rrow[t * 4 + 1] = qt[1]; rrow[t * 4 + 2] = qt[2]; rrow[t * 4 + 3] = qt[3];
    };
    fk_tree_walk(a.n_ops, tab, [&](int k) { return tab.row(k); }, qf, ssave, sload, emit);
    wave_lds_sync();
    tile_store<0>(pos + tc.b0 * 3 * T, tc.rows, 3 * T, magic_p, lp, lane, tc.full && ((3 * T) & 1) && (align & AL_POS),
                  tc.full && (align & AL_POS));
    tile_store<0>(quat + tc.b0 * 4 * T, tc.rows, 4 * T, magic_r, lr, lane, false, tc.full && (align & AL_QUAT));
}

// The same walk for MANY targets whose slots are numbered in walk order (DRM_WALK_TARGETS_ORDERED: all links of a robot,
// compute_forward_kinematics_all_links, robot_model.py:197-221).  Staging all T poses of a tile (64 x 28 T bytes: 50 KB at 28
// links) leaves two wavefronts per CU; here the outputs leave a GROUP of eight consecutive slots at a time — per sample a run
// of 96 B of positions and one of 128 B of quaternions — through two small tiles: 15 KB, and the writes of one group overlap
// the walk to the next.  16-byte sc1 stores (nt beyond the Infinity Cache); positions fall back to 4-byte stores when 12 T is
// not a multiple of 16.
// LDS: [ table ][ q : 64 (n|1) ][ pos group : 64 x 25 ][ quat group : 64 x 33 ][ slots : n_slots * 12 * 64 ]
#ifndef DRM_FK_LINKS_FAN_TILES
#define DRM_FK_LINKS_FAN_TILES 4096 /* drm_fk_links: tiles up to which a walk with a shared part is fanned out over wavefronts */
#endif
#ifndef DRM_FK_GROUP
#define DRM_FK_GROUP 8
#endif
constexpr int FK_GROUP = DRM_FK_GROUP, FK_GP = 3 * FK_GROUP + 1, FK_GR = 4 * FK_GROUP + 1;
template <bool NT>
__global__ void __launch_bounds__(WAVE)
    fk_tree_groups_kernel(TreeArgs a, const float *__restrict__ q, int64_t B, int T, float *__restrict__ pos, float *__restrict__ quat,
                          uint32_t magic_q, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n);
    float *lq = smem + table_lds_floats(a.n_ops);
    float *lp = lq + round4(WAVE * Sq);
    float *lr = lp + round4(WAVE * FK_GP);
    float *ls = lr + round4(WAVE * FK_GR); // save slots: [slot][12][64]

    const TableLds tab = stage_tree_table(a, smem);
    tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, tc.full && (n & 1) && (align & AL_Q), tc.full && (align & AL_Q));
    wave_lds_sync();
    const bool live = (int)lane < tc.rows;
    const float *qrow = lq + lane * Sq;
    float *prow = lp + lane * FK_GP, *rrow = lr + lane * FK_GR;
    const bool pos16 = (T & 3) == 0 && (align & AL_POS);
    const bool quat16 = (align & AL_QUAT) != 0;
    float *gp = pos + tc.b0 * 3 * T, *gr = quat + tc.b0 * 4 * T;
    // the group's two tiles -> global: consecutive lanes write consecutive 16 bytes of a sample's run, then the next sample's
    auto flush = [&](int g, int cnt) {
        wave_lds_sync();
        if (cnt == FK_GROUP && pos16) {
            constexpr int PARTS = 3 * FK_GROUP / 4;
            for (int i = (int)lane; i < tc.rows * PARTS; i += WAVE) {
                const int s = i / PARTS, part = i - s * PARTS;
                const float *src = lp + s * FK_GP + part * 4;
                store16_wt<NT>(gp + (int64_t)s * 3 * T + g * (3 * FK_GROUP) + part * 4, make_float4(src[0], src[1], src[2], src[3]));
            }
        } else {
            const int w = 3 * cnt;
            for (int i = (int)lane; i < tc.rows * w; i += WAVE) {
                const int s = i / w, c = i - s * w;
                gp[(int64_t)s * 3 * T + g * (3 * FK_GROUP) + c] = lp[s * FK_GP + c];
            }
        }
        if (quat16) {
            for (int i = (int)lane; i < tc.rows * cnt; i += WAVE) {
                const int s = i / cnt, part = i - s * cnt;
                const float *src = lr + s * FK_GR + part * 4;
                store16_wt<NT>(gr + (int64_t)s * 4 * T + g * (4 * FK_GROUP) + part * 4, make_float4(src[0], src[1], src[2], src[3]));
            }
        } else {
            const int w = 4 * cnt;
            for (int i = (int)lane; i < tc.rows * w; i += WAVE) {
                const int s = i / w, c = i - s * w;
                gr[(int64_t)s * 4 * T + g * (4 * FK_GROUP) + c] = lr[s * FK_GR + c];
            }
        }
        wave_lds_sync(); // the tiles are free for the next group
    };
    fk_tree_walk(
        a.n_ops, tab, [&](int k) { return tab.row(k); }, [&](int d) -> float { return live ? qrow[d] : 0.0f; },
        [&](int s, const PoseP &P) { lds_put_pose(ls, s, lane, P); }, [&](int s, PoseP &P) { lds_get_pose(ls, s, lane, P); },
        [&](int t, const float *p, const float *qt) {
            const int g = t / FK_GROUP, c = t - g * FK_GROUP;
            prow[c * 3 + 0] = p[0]; prow[c * 3 + 1] = p[1]; prow[c * 3 + 2] = p[2];
            rrow[c * 4 + 0] = qt[0]; rrow[c * 4 + 1] = qt[1]; rrow[c * 4 + 2] = qt[2]; rrow[c * 4 + 3] = qt[3];
            if (c == FK_GROUP - 1 || t == T - 1) flush(g, c + 1); // (t is wave-uniform: the slots come in walk order)
        });
}

// Many targets FANNED OUT (DRM_WALK_FK_FAN: the host found a hub — the root of a hand, a palm — behind which the walk splits
// into sub-trees, flatten.fk_fan_partition): a block of K <= 4 wavefronts per tile, each walking the ops [0, prefix_end) every
// sub-tree hangs off and then its own run [seg_begin[j], seg_begin[j+1]), writing its targets' columns of the [64, 3T] / [64, 4T]
// tiles in LDS (28 T bytes per sample: 50 KB at 28 links — two blocks of four wavefronts per CU); the block stores the tiles as
// linear 16-byte runs.  Allegro hand, all 20 links, 2^20 samples: 176 us against 285 us grouped.  (Letting every wavefront
// flush groups of its own slots instead — four writers of short runs per sample row — was slower than the single-wavefront
// grouped form: 397-513 us.)  The wavefronts share the save slots: they all write the same poses in the shared part, and the
// host only fans out when no two runs write the same slot.
// LDS: [ table ][ q : 64 (n|1) ][ slots : n_slots * 12 * 64 ][ pos : 64 (3T|1) ][ quat : 64 (4T|1) ]
template <bool NT>
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    fk_tree_fan_kernel(TreeArgs a, const float *__restrict__ q, int64_t B, int T, float *__restrict__ pos, float *__restrict__ quat,
                       uint32_t magic_q, uint32_t magic_p, uint32_t magic_r, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n), Sp = pad_odd(3 * T), Sr = pad_odd(4 * T);
    float *lq = smem + table_lds_floats(a.n_ops);
    float *ls = lq + round4(WAVE * Sq);
    float *lp = ls + a.n_slots * 12 * WAVE;
    float *lr = lp + round4(WAVE * Sp);
    const TableLds tab = stage_tree_table(a, smem);
    if (wave == 0) tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, tc.full && (n & 1) && (align & AL_Q), tc.full && (align & AL_Q));
    __syncthreads();
    const bool live = (int)lane < tc.rows;
    const float *qrow = lq + lane * Sq;
    float *prow = lp + lane * Sp, *rrow = lr + lane * Sr;
    const int P = a.prefix_end, first = a.seg_begin[wave], last = a.seg_begin[wave + 1];
    fk_tree_walk_ranges(
        P, first, last, tab, [&](int k) { return tab.row(k); }, [&](int d) -> float { return live ? qrow[d] : 0.0f; },
        [&](int s, const PoseP &Q) { lds_put_pose(ls, s, lane, Q); }, [&](int s, PoseP &Q) { lds_get_pose(ls, s, lane, Q); },
        [&](int k, int t, const float *p, const float *qt) {
            if (k < P && wave != 0) return; // the shared part's targets are wavefront 0's
            prow[t * 3 + 0] = p[0]; prow[t * 3 + 1] = p[1]; prow[t * 3 + 2] = p[2];
            rrow[t * 4 + 0] = qt[0]; rrow[t * 4 + 1] = qt[1]; rrow[t * 4 + 2] = qt[2]; rrow[t * 4 + 3] = qt[3];
        });
    __syncthreads();
    block_tile_store<NT>(pos + tc.b0 * 3 * T, tc.rows, 3 * T, magic_p, lp, tc.full && (align & AL_POS));
    block_tile_store<NT>(quat + tc.b0 * 4 * T, tc.rows, 4 * T, magic_r, lr, tc.full && (align & AL_QUAT));
}

// LINK-MAJOR outputs (drm_fk_links: pos [T, B, 3], quat [T, B, 4] — every link's poses a contiguous array, which is what
// compute_forward_kinematics_all_links hands out per link, robot_model.py:197-221): a target's 64 poses of a tile are ONE
// contiguous run — 768 B of positions (through a 64 x 3 LDS stage, 48 16-byte stores) and 1 KB of quaternions (a 16-byte store
// per lane, straight from registers) — written the moment the walk reaches the target.  Nothing of a tile's outputs waits in
// LDS: table + q tile + save slots, ~8 KB instead of the 15-50 KB of the sample-major forms above.  A walk that splits behind a
// hub (DRM_WALK_FK_FAN) gets a wavefront per run of sub-trees; they share nothing but the table and the q tile.
// LDS: [ table ][ q : 64 (n|1) ][ slots : n_slots * 12 * 64 ][ position stage : 64 x 3 per wavefront ]
template <bool NT>
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    fk_tree_links_kernel(TreeArgs a, const float *__restrict__ q, int64_t B, float *__restrict__ pos, float *__restrict__ quat,
                         uint32_t magic_q, uint32_t align, int fan) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n);
    float *lq = smem + table_lds_floats(a.n_ops);
    float *ls = lq + round4(WAVE * Sq);
    float *lp = ls + a.n_slots * 12 * WAVE + wave * (WAVE * 3);
    const TableLds tab = stage_tree_table(a, smem);
    if (wave == 0) tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, tc.full && (n & 1) && (align & AL_Q), tc.full && (align & AL_Q));
    __syncthreads();
    const bool live = (int)lane < tc.rows;
    const float *qrow = lq + lane * Sq;
    // a link's [B, 3] array starts 12 B t bytes in: 16-byte aligned for every t when B is a multiple of 4
    const bool pos16 = tc.full && (align & AL_POS) && (B & 3) == 0;
    const bool quat16 = (align & AL_QUAT) != 0;
    const int P = fan ? a.prefix_end : 0, first = fan ? a.seg_begin[wave] : 0, last = fan ? a.seg_begin[wave + 1] : a.n_ops;
    fk_tree_walk_ranges(
        P, first, last, tab, [&](int k) { return tab.row(k); }, [&](int d) -> float { return live ? qrow[d] : 0.0f; },
        [&](int s, const PoseP &Q) { lds_put_pose(ls, s, lane, Q); }, [&](int s, PoseP &Q) { lds_get_pose(ls, s, lane, Q); },
        [&](int k, int t, const float *p, const float *qt) {
            if (k < P && wave != 0) return; // the shared part's targets are wavefront 0's
            float *gr = quat + ((int64_t)t * B + tc.b0) * 4, *gp = pos + ((int64_t)t * B + tc.b0) * 3;
            if (live) {
                if (quat16) store16_wt<NT>(gr + lane * 4, make_float4(qt[0], qt[1], qt[2], qt[3]));
                else { gr[lane * 4 + 0] = qt[0]; gr[lane * 4 + 1] = qt[1]; gr[lane * 4 + 2] = qt[2]; gr[lane * 4 + 3] = qt[3]; }
            }
            lp[lane * 3 + 0] = p[0]; lp[lane * 3 + 1] = p[1]; lp[lane * 3 + 2] = p[2];
            wave_lds_sync();
            if (pos16) {
                if (lane < 48u) store16_wt<NT>(gp + lane * 4, reinterpret_cast<const float4 *>(lp)[lane]);
            } else {
                for (int i = (int)lane; i < tc.rows * 3; i += WAVE) gp[i] = lp[i];
            }
            wave_lds_sync(); // the stage is free for the next target
        });
}

// ---------------------------------------------------------------------------------------------------
// Fan-out FK: T <= 4 targets whose root->target chains share (almost) nothing — the fingertips of a hand that hang
// off a common palm (Allegro, TriFinger).  The merged walk above makes ONE lane compute all T chains of a sample
// one after the other; here a block of T wavefronts owns a tile of 64 samples and wavefront t walks ONLY the chain of
// target t (constants stay wave-uniform), so the serial work per wave is one chain and T waves per SIMD hide each
// other's latencies.  The q tile is loaded once per block into shared LDS, the T x (pos, quat) results of a sample
// are assembled in LDS and leave as one contiguous row-major tile.
// LDS: [ q ][ pos ][ quat ] shared, then per wavefront its chain's table.
// ---------------------------------------------------------------------------------------------------
struct FanoutTables {
    const float *ops_f[4];
    const int32_t *ops_i[4];
    int32_t n_ops[4];
    int32_t cap;
};

__global__ void __launch_bounds__(WAVE * 4)
    fk_fanout_kernel(FanoutTables tab, int T, int n, int max_ops, const float *__restrict__ q, int64_t B, float *__restrict__ pos,
                     float *__restrict__ quat, uint32_t magic_q, uint32_t magic_p, uint32_t magic_r, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const TileCtx tc = tile_begin(B);
    const int Sq = pad_odd(n), Sp = pad_odd(3 * T), Sr = pad_odd(4 * T);
    float *lq = smem;
    float *lp = lq + round4(WAVE * Sq);
    float *lr = lp + round4(WAVE * Sp);
    float *lt = lr + round4(WAVE * Sr) + wave * table_lds_floats(max_ops); // this wavefront's chain table

    if (wave == 0) tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, tc.full && (n & 1) && (align & AL_Q), tc.full && (align & AL_Q));
    // wave-uniform choice of this wave's chain tables (select chain: T <= 4), staged by the wavefront itself
    const float *ops_f = wave == 0 ? tab.ops_f[0] : wave == 1 ? tab.ops_f[1] : wave == 2 ? tab.ops_f[2] : tab.ops_f[3];
    const int32_t *ops_i = wave == 0 ? tab.ops_i[0] : wave == 1 ? tab.ops_i[1] : wave == 2 ? tab.ops_i[2] : tab.ops_i[3];
    const int n_ops = wave == 0 ? tab.n_ops[0] : wave == 1 ? tab.n_ops[1] : wave == 2 ? tab.n_ops[2] : tab.n_ops[3];
    int *lw = reinterpret_cast<int *>(lt + round4(max_ops * DRM_OPF_STRIDE));
    for (unsigned i = lane; i < (unsigned)n_ops * (DRM_OPF_STRIDE / 4); i += WAVE)
        reinterpret_cast<float4 *>(lt)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    for (unsigned i = lane; i < (unsigned)n_ops; i += WAVE) {
        lw[i] = ops_i[DRM_OPI_W0 * tab.cap + i];
        lw[n_ops + i] = ops_i[DRM_OPI_W1 * tab.cap + i];
    }
    __syncthreads();

    const bool live = (int)lane < tc.rows;
    const float *qrow = lq + lane * Sq;
    float *prow = lp + lane * Sp + wave * 3;
    float *rrow = lr + lane * Sr + wave * 4;
    TableLds chain;
    chain.f = lt; chain.w = lw; chain.n_ops = n_ops;
    auto emit1 = [&](int, const float *p, const float *qt) { // the chain's single target = column `wave` of the row
        prow[0] = p[0]; prow[1] = p[1]; prow[2] = p[2];
        rrow[0] = qt[0]; rrow[1] = qt[1]; rrow[2] = qt[2]; rrow[3] = qt[3];
    };
    fk_tree_walk(
        n_ops, chain, [&](int k) { return lt + k * DRM_OPF_STRIDE; }, [&](int d) -> float { return live ? qrow[d] : 0.0f; },
        [&](int, const PoseP &) {}, [&](int, PoseP &) {}, emit1);
    __syncthreads();
    // the assembled [64, 3T] and [64, 4T] tiles leave with coalesced stores, one tensor per wave pair
    if (wave == 0)
        tile_store<0>(pos + tc.b0 * 3 * T, tc.rows, 3 * T, magic_p, lp, lane, tc.full && ((3 * T) & 1) && (align & AL_POS),
                      tc.full && (align & AL_POS));
    if (wave == T - 1)
        tile_store<0>(quat + tc.b0 * 4 * T, tc.rows, 4 * T, magic_r, lr, lane, false, tc.full && (align & AL_QUAT));
}

} // namespace drm

using namespace drm;

extern "C" int drm_fk_links(const drm_walk *w, const float *q, int64_t B, int32_t n_targets, float *pos, float *quat, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !pos || !quat) return fail(DRM_ERR_INVALID, "q / pos / quat must not be NULL");
    if (B < 0 || n_targets < 1) return fail(DRM_ERR_INVALID, "negative batch or no targets");
    if (n_targets > w->n_ops) return fail(DRM_ERR_INVALID, "more targets than ops in the walk");
    if (B == 0) return DRM_OK;
    if ((((uintptr_t)w->ops_f) & 15u) != 0) return fail(DRM_ERR_INVALID, "ops_f must be 16-byte aligned");
    const int n = w->n_dofs, T = n_targets;
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    // A wavefront per run of sub-trees behind the hub: every wavefront repeats the ops in front of the hub, and with ~8 KB of LDS
    // per tile the kernel is bound by instruction issue once the device is full (iiwa7 + Allegro at 2^20: 248 us fanned out,
    // eight of a wavefront's thirteen ops repeated) — so a walk with a shared part is fanned out only while the tiles do not
    // fill the device anyway; a hand (hub = the root, nothing repeated) always is.
    const bool fan = (w->shape & DRM_WALK_FK_FAN) && w->n_segments >= 2 && w->n_segments <= DRM_MAX_SEGMENTS && segments_ok(w) &&
                     (w->prefix_end == 0 || tiles <= DRM_FK_LINKS_FAN_TILES);
    TreeArgs a = tree_args(w, false);
    if (!fan) { a.n_segments = 1; a.prefix_end = 0; }
    const int K = fan ? a.n_segments : 1;
    const size_t lds = sizeof(float) * (size_t)(table_lds_floats(a.n_ops) + round4(WAVE * pad_odd(n)) + w->n_slots * 12 * WAVE + K * WAVE * 3);
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
    hipStream_t s = (hipStream_t)stream;
    if (stream_past_llc(B * 28 * T)) {
        rc = ensure_lds_tree(fk_tree_links_kernel<true>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(fk_tree_links_kernel<true>, dim3((unsigned)tiles), dim3(WAVE * K), lds, s, a, q, B, pos, quat, div_magic(n), align, (int)fan);
    } else {
        rc = ensure_lds_tree(fk_tree_links_kernel<false>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(fk_tree_links_kernel<false>, dim3((unsigned)tiles), dim3(WAVE * K), lds, s, a, q, B, pos, quat, div_magic(n), align, (int)fan);
    }
    return launched();
}

extern "C" int drm_fk_fanout(const drm_walk *chains, int32_t n_chains, const float *q, int64_t B, float *pos, float *quat,
                             void *stream) {
    if (!chains || n_chains < 2 || n_chains > 4) return fail(DRM_ERR_INVALID, "fan-out FK takes 2 to 4 chains");
    if (!q || !pos || !quat) return fail(DRM_ERR_INVALID, "q / pos / quat must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    FanoutTables tab;
    int max_ops = 1;
    for (int t = 0; t < 4; ++t) {
        const drm_walk *w = chains + (t < n_chains ? t : 0);
        int rc = check_walk(w);
        if (rc) return rc;
        if (w->capacity != chains[0].capacity || w->n_dofs != chains[0].n_dofs || w->n_slots != 0)
            return fail(DRM_ERR_INVALID, "fan-out chains must share capacity and n_dofs and have no branch points");
        if ((((uintptr_t)w->ops_f) & 15u) != 0) return fail(DRM_ERR_INVALID, "ops_f must be 16-byte aligned");
        tab.ops_f[t] = w->ops_f;
        tab.ops_i[t] = w->ops_i;
        tab.n_ops[t] = w->n_ops;
        if (w->n_ops > max_ops) max_ops = w->n_ops;
    }
    tab.cap = chains[0].capacity;
    if (B == 0) return DRM_OK;
    const int n = chains[0].n_dofs, T = n_chains;
    {   // chains of capacity 8 (the fingers of a hand): full tiles through the straight-line fan-out kernel
        const int64_t done = launch_fk_fan_chains(chains, T, q, B, pos, quat, (hipStream_t)stream);
        if (done > 0) {
            int rc = launched();
            if (rc || done == B) return rc;
            q += done * n; pos += done * 3 * T; quat += done * 4 * T; B -= done;
        }
    }
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    const size_t lds = sizeof(float) * (size_t)(round4(WAVE * pad_odd(n)) + round4(WAVE * pad_odd(3 * T)) + round4(WAVE * pad_odd(4 * T)) +
                                                T * table_lds_floats(max_ops));
    int rc = ensure_lds_tree(fk_fanout_kernel, lds);
    if (rc) return rc;
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fk_fanout_kernel, dim3((unsigned)tiles), dim3(WAVE * T), lds, s, tab, T, n, max_ops, q, B, pos, quat,
                       div_magic(n), div_magic(3 * T), div_magic(4 * T), align);
    return launched();
}

extern "C" int drm_fk(const drm_walk *w, const float *q, int64_t B, int32_t n_targets, float *pos, float *quat,
                      void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !pos || !quat) return fail(DRM_ERR_INVALID, "q / pos / quat must not be NULL");
    if (B < 0 || n_targets < 1) return fail(DRM_ERR_INVALID, "negative batch or no targets");
    if (n_targets > w->n_ops) return fail(DRM_ERR_INVALID, "more targets than ops in the walk");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs, T = n_targets;
    if (T == 1 && w->n_ops >= 1) {
        // 7-DoF arms, one target at the end of the chain: full tiles through the packed-FP32 chain kernel
        int64_t done = launch_fk_arm(w, q, B, pos, quat, (hipStream_t)stream);
        // ... any other serial chain of up to 16 ops through the straight-line chain kernel (drm_chain_kernels.hip)
        if (done == 0) done = launch_chain_fk(w, q, B, pos, quat, (hipStream_t)stream);
        if (done > 0) {
            rc = launched();
            if (rc || done == B) return rc;
            drm_walk generic = *w;
            generic.shape &= ~(DRM_WALK_ARM_CHAIN | DRM_WALK_SERIAL_CHAIN);
            return drm_fk(&generic, q + done * n, B - done, 1, pos + done * 3, quat + done * 4, stream);
        }
    }
    if ((((uintptr_t)w->ops_f) & 15u) != 0) return fail(DRM_ERR_INVALID, "ops_f must be 16-byte aligned");
    TreeArgs a = tree_args(w);
    a.n_segments = 1; a.prefix_end = 0;
#ifndef DRM_NO_FK_GROUPS
    if ((w->shape & DRM_WALK_TARGETS_ORDERED) && T > 8) {
        if ((w->shape & DRM_WALK_FK_FAN) && w->n_segments >= 2 && w->n_segments <= DRM_MAX_SEGMENTS && segments_ok(w)) {
            // fanned out over the sub-trees behind the hub, the whole tile staged — when two such blocks fit a CU
            const size_t lds_fan = sizeof(float) * (size_t)(table_lds_floats(w->n_ops) + round4(WAVE * pad_odd(n)) + w->n_slots * 12 * WAVE +
                                                            round4(WAVE * pad_odd(3 * T)) + round4(WAVE * pad_odd(4 * T)));
            if (lds_fan <= (size_t)80 * 1024) {
                const TreeArgs af = tree_args(w, false);
                const int64_t tiles = (B + WAVE - 1) / WAVE;
                if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
                const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
                hipStream_t s = (hipStream_t)stream;
                if (stream_past_llc(B * 28 * T)) {
                    rc = ensure_lds_tree(fk_tree_fan_kernel<true>, lds_fan);
                    if (rc) return rc;
                    hipLaunchKernelGGL(fk_tree_fan_kernel<true>, dim3((unsigned)tiles), dim3(WAVE * af.n_segments), lds_fan, s, af, q, B, T, pos,
                                       quat, div_magic(n), div_magic(3 * T), div_magic(4 * T), align);
                } else {
                    rc = ensure_lds_tree(fk_tree_fan_kernel<false>, lds_fan);
                    if (rc) return rc;
                    hipLaunchKernelGGL(fk_tree_fan_kernel<false>, dim3((unsigned)tiles), dim3(WAVE * af.n_segments), lds_fan, s, af, q, B, T, pos,
                                       quat, div_magic(n), div_magic(3 * T), div_magic(4 * T), align);
                }
                return launched();
            }
        }
        // many targets in walk order (all links of a robot): outputs leave a group of eight slots at a time
        const size_t lds = sizeof(float) * (size_t)(table_lds_floats(a.n_ops) + round4(WAVE * pad_odd(n)) + round4(WAVE * FK_GP) +
                                                    round4(WAVE * FK_GR) + w->n_slots * 12 * WAVE);
        const int64_t tiles = (B + WAVE - 1) / WAVE;
        if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
        const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
        hipStream_t s = (hipStream_t)stream;
        if (stream_past_llc(B * 28 * T)) {
            rc = ensure_lds_tree(fk_tree_groups_kernel<true>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(fk_tree_groups_kernel<true>, dim3((unsigned)tiles), dim3(WAVE), lds, s, a, q, B, T, pos, quat, div_magic(n), align);
        } else {
            rc = ensure_lds_tree(fk_tree_groups_kernel<false>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(fk_tree_groups_kernel<false>, dim3((unsigned)tiles), dim3(WAVE), lds, s, a, q, B, T, pos, quat, div_magic(n), align);
        }
        return launched();
    }
#endif
    const size_t lds = sizeof(float) * (size_t)(table_lds_floats(a.n_ops) + round4(WAVE * pad_odd(n)) + round4(WAVE * pad_odd(3 * T)) +
                                                round4(WAVE * pad_odd(4 * T)) + w->n_slots * 12 * WAVE);
    rc = ensure_lds_tree(fk_tree_kernel, lds);
    if (rc) return rc;
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fk_tree_kernel, dim3((unsigned)tiles), dim3(WAVE), lds, s, a, q, B, T, pos, quat, div_magic(n), div_magic(3 * T),
                       div_magic(4 * T), align);
    return launched();
}

extern "C" int drm_fk_fanout_links(const drm_walk *chains, int32_t n_chains, const float *q, int64_t B, float *pos, float *quat,
                                   void *stream) {
    if (!chains || n_chains < 2 || n_chains > 4) return fail(DRM_ERR_INVALID, "fan-out FK takes 2 to 4 chains");
    if (!q || !pos || !quat) return fail(DRM_ERR_INVALID, "q / pos / quat must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    for (int t = 0; t < n_chains; ++t) {
        int rc = check_walk(chains + t);
        if (rc) return rc;
        if (chains[t].capacity != chains[0].capacity || chains[t].n_dofs != chains[0].n_dofs || chains[t].n_slots != 0)
            return fail(DRM_ERR_INVALID, "fan-out chains must share capacity and n_dofs and have no branch points");
    }
    if (B == 0) return DRM_OK;
    const int n = chains[0].n_dofs;
    // full tiles: a wavefront per chain, each writing its link's arrays; what is left (a ragged tail; everything when the fan-out
    // kernel does not take the call): one single-target launch per chain into the same link-major arrays
    int64_t done = 0;
    {   // the hand's own kernel: every chain's constants folded into the instruction stream (csrc/drm_arm_static.hpp, specialize.py) —
        // the host stored the SAME handle on every chain walk of the call it was built for
        const void *own = chains[0].special[DRM_SPECIAL_FK_FAN_LINKS];
        for (int t = 1; t < n_chains; ++t)
            if (chains[t].special[DRM_SPECIAL_FK_FAN_LINKS] != own) own = nullptr;
        if (own && B >= WAVE && !(B & 3) && B / WAVE < 0x7fffffffLL && ((((uintptr_t)q | (uintptr_t)pos | (uintptr_t)quat) & 15u) == 0)) {
            const int64_t n_tiles = B / WAVE;
            int64_t rows = B;
            void *args[] = {(void *)&q, (void *)&pos, (void *)&quat, (void *)&rows};
            hipError_t e = hipModuleLaunchKernel((hipFunction_t)own, (unsigned)n_tiles, 1, 1, WAVE * (unsigned)n_chains, 1, 1, 0, (hipStream_t)stream, args, nullptr);
            if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_fk_fan_links_static): %s", hipGetErrorString(e));
            done = n_tiles * WAVE;
        }
    }
    if (done == 0) done = launch_fk_fan_chains(chains, n_chains, q, B, pos, quat, (hipStream_t)stream, true);
    if (done > 0) {
        int rc = launched();
        if (rc) return rc;
    }
    if (done < B)
        for (int t = 0; t < n_chains; ++t) {
            int rc = drm_fk(chains + t, q + done * n, B - done, 1, pos + ((int64_t)t * B + done) * 3, quat + ((int64_t)t * B + done) * 4, stream);
            if (rc) return rc;
        }
    return DRM_OK;
}

