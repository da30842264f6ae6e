// drm_rnea_backward.hip — K7: reverse-mode derivative of the RNEA inverse dynamics with respect to the per-link
// constants (R_fixed, trans, mass, mass * com, I_o, damping) and to q, qd, qdd.
//
// Replaces what torch autograd does for the reference when a loss on compute_inverse_dynamics' torques is
// back-propagated (robot_model.py:305-375 with learnable link parameters robot_model.py:669-713;
// examples/learn_dynamics_iiwa.py:49-96): there one backward node per tiny torch op of ~11 k ops per call, here four
// sweeps per sample over the walk (drm_sample.hpp rnea_backward_walk) that recompute the forward pass, plus the
// same deterministic batch reduction of the constant gradients as drm_fk_backward.
//
// Per sample: in q, qd, qdd, grad_tau [n]; out grad_q, grad_qd, grad_qdd [n] (optional).      n = 7: 112 + 84 B
// Per launch: out grad_ops_f[cap, 32] (every constant of the ops selected by param_mask, zeros elsewhere).
// Between the two sweeps every link keeps (cos, sin) of its joint angle and every LEAF link its motion and force adjoint
// (2 + 18 floats, drm_sample.hpp rnea_backward_walk) in LDS, or, when a big walk does not fit, in a slice of the caller's
// scratch buffer in HBM (PARK_HBM).  The input tiles double as the gradient tiles (a DoF's q, qd, qdd are last read by
// the op that then writes its gradients).
// LDS per wave: [ q -> grad_q, qd -> grad_qd, qdd -> grad_qdd, grad_tau : 4 x 64 (n|1) ][ constant-gradient sums : cap*32 ]
//               [ slots : n_slots*36*64 ][ records : (n_ops*2 + n_leaves*18)*64 unless PARK_HBM ]
#include "drm_common.hpp"
#include "drm_sample.hpp"
#include "drm_tree_dev.hpp"

namespace drm {

constexpr int REC_FLOATS = 26, SLOT_FLOATS = 36; // (REC_FLOATS: the conservative per-op figure of the scratch query)
constexpr int TRIG_FLOATS = 2, LEAF_FLOATS = 18;
constexpr int BWD_SHORT_OPS = 6; // segments of up to this many ops in a row take rnea_backward_walk_short in the fanned-out kernel
__host__ __device__ static inline int record_floats(int n_ops, int n_leaves) { return (n_ops * TRIG_FLOATS + n_leaves * LEAF_FLOATS) * WAVE; }

// One kernel for every walk: the sweeps loop over the n_ops links (drm_sample.hpp rnea_backward_walk), so neither the
// code nor the register file grows with the robot; `cap` only fixes the row pitch of grad_ops_f / the partial sums.
template <bool PARK_HBM>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    rnea_backward_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int cap, int n_ops, int n_leaves,
                         int n, int n_slots, int flags, const float *__restrict__ q, const float *__restrict__ qd,
                         const float *__restrict__ qdd, const float *__restrict__ gtau, int64_t B, float *__restrict__ gq,
                         float *__restrict__ gqd, float *__restrict__ gqdd, uint64_t param_mask,
                         float *__restrict__ partials, float *__restrict__ park_hbm, uint32_t magic_q, int lds_per_wave,
                         uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NV = cap * DRM_OPF_STRIDE;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const unsigned lane = threadIdx.x & 63u;
    const int64_t wave_id = (int64_t)blockIdx.x * wpb + wave_in_block;
    const int64_t n_waves = (int64_t)gridDim.x * wpb;
    const int64_t n_tiles = (B + WAVE - 1) / WAVE;

    const int Sq = pad_odd(n), region = round4(WAVE * Sq);
    float *lq = smem + wave_in_block * lds_per_wave;
    float *lqd = lq + region, *lqdd = lqd + region, *lgt = lqdd + region;
    float *lgq = lq, *lgqd = lqd, *lgqdd = lqdd;                   // gradients over the inputs (see the header comment)
    float *lacc = lgt + region;                                    // this wave's running sums of the constant gradients
    float *lsl = lacc + NV;                                        // slots   [slot][36][64]
    float *lrec = lsl + n_slots * (SLOT_FLOATS * WAVE);            // records: [op][2][64] trig, then [leaf][18][64]
    float *rec = (PARK_HBM ? park_hbm + wave_id * (int64_t)record_floats(n_ops, n_leaves) : lrec) + lane;
    float *rec_leaf = rec + n_ops * (TRIG_FLOATS * WAVE);
    float *slot = lsl + lane;
    const int32_t *ctl = ops_i + DRM_OPI_CTRL * cap;

    for (int i = (int)lane; i < NV; i += WAVE) lacc[i] = 0.0f;

    for (int64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
        const int64_t b0 = tile * WAVE;
        const int64_t left = B - b0;
        const int rows = left < WAVE ? (int)left : WAVE;
        const bool full = rows == WAVE, fast = full && (n & 1);
        const bool live = (int)lane < rows;
        wave_lds_sync(); // the previous tile's LDS reads are done before this tile overwrites
        tile_load<0>(q + b0 * n, rows, n, magic_q, lq, lane, fast && (align & AL_Q), full && (align & AL_Q));
        tile_load<0>(qd + b0 * n, rows, n, magic_q, lqd, lane, fast && (align & AL_QD), full && (align & AL_QD));
        if (qdd) tile_load<0>(qdd + b0 * n, rows, n, magic_q, lqdd, lane, fast && (align & AL_QDD), full && (align & AL_QDD));
        tile_load<0>(gtau + b0 * n, rows, n, magic_q, lgt, lane, fast && (align & AL_TAU), full && (align & AL_TAU));
        for (int s = 0; s < n_slots * SLOT_FLOATS; ++s) slot[s * WAVE] = 0.0f;
        wave_lds_sync();

        const unsigned row = lane * Sq;
        const bool has_qdd = qdd != nullptr;
        auto qf = [&](int d, float &a, float &v, float &c) {
            a = live ? lq[row + d] : 0.0f; // zeros, not stale LDS, past a partial tile
            v = lqd[row + d];
            c = has_qdd ? lqdd[row + d] : 0.0f;
        };
        auto gt = [&](int d) -> float { return live ? lgt[row + d] : 0.0f; };
        // record of op k: offset 24 = (cos, sin) of every op; offsets 0 (motion, 12) and 18 (tbar, 6) of LEAF ops only, at
        // the leaf's ordinal (bits 26..31 of the control word)
        auto where = [&](int k, int off) -> float * {
            if (off == 24) return rec + k * (TRIG_FLOATS * WAVE);
            const int leaf = (int)(((uint32_t)ctl[k]) >> 26);
            return rec_leaf + (leaf * LEAF_FLOATS + (off == 0 ? 0 : 12)) * WAVE;
        };
        auto park = [&](int k, int off, const float *v, int cnt) {
            float *r = where(k, off);
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) r[i * WAVE] = v[i];
        };
        auto unpark = [&](int k, int off, float *v, int cnt) {
            const float *r = where(k, off);
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) v[i] = r[i * WAVE];
        };
        auto slot_put = [&](int s, int off, const float *v, int cnt) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) slot[(s * SLOT_FLOATS + off + i) * WAVE] = v[i];
        };
        auto slot_get = [&](int s, int off, float *v, int cnt) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) v[i] = slot[(s * SLOT_FLOATS + off + i) * WAVE];
        };
        auto slot_add = [&](int s, int off, const float *v, int cnt) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) slot[(s * SLOT_FLOATS + off + i) * WAVE] += v[i];
        };
        auto slot_take = [&](int s, int off, float *v, int cnt) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) {
                    v[i] = slot[(s * SLOT_FLOATS + off + i) * WAVE];
                    slot[(s * SLOT_FLOATS + off + i) * WAVE] = 0.0f;
                }
        };
        auto gout = [&](int d, float a, float v, float c) { lgq[row + d] = a; lgqd[row + d] = v; lgqdd[row + d] = c; };
        auto param_out = [&](int k, const float *g) { // wave-uniform call: only for the ops param_mask selects
            wave_sums_lane63<DRM_OPF_DAMP + 1>(lane, [&](int j) { return live ? g[j] : 0.0f; }, // lanes past a partial tile hold garbage
                                               [&](int j, float total) { lacc[k * DRM_OPF_STRIDE + j] += total; }); // tiles in this wavefront's fixed order
        };
        rnea_backward_walk(ops_f, ctl, 0, 0, n_ops, flags, param_mask, gq != nullptr, qf, gt, park, unpark, slot_put, slot_get,
                           slot_add, slot_take, gout, param_out);
        if (gq) {
            wave_lds_sync();
            tile_store<0>(gq + b0 * n, rows, n, magic_q, lgq, lane, fast && (align & AL_POS), full && (align & AL_POS));
            tile_store<0>(gqd + b0 * n, rows, n, magic_q, lgqd, lane, fast && (align & AL_QUAT), full && (align & AL_QUAT));
            tile_store<0>(gqdd + b0 * n, rows, n, magic_q, lgqdd, lane, fast && (align & AL_LIN), full && (align & AL_LIN));
        }
    }
    wave_lds_sync();
    float *prow = partials + wave_id * NV;
    for (int i = (int)lane; i < NV; i += WAVE) prow[i] = lacc[i];
}


// The same walk FANNED OUT over the walk's segments (independent sub-trees hanging off the static prefix — the fingers of a
// hand): a block owns one 64-sample tile at a time, wavefront s walks segment s (rnea_backward_walk replays the prefix for
// it).  What one wavefront parks is a finger's worth (2 floats per op + 18 per leaf), so a hand fits LDS with eight
// wavefronts per CU where the single-wavefront form, parking the whole tree, ran one or two.
// Shared LDS: [ q -> grad_q, qd -> grad_qd, qdd -> grad_qdd, grad_tau ][ slots : n_slots*36*64 ]; the slots of prefix branch
// points are written by every wavefront with the same values (their replay of the prefix) and only read afterwards; slots
// inside a segment belong to that segment.  Per wavefront: [ constant-gradient sums : cap*32 ][ records ].
// Not for launches that want the constant gradients of PREFIX ops (the host checks): nothing flows into the prefix here.
struct FanArgs {
    int32_t n_seg, p_end;
    int32_t seg_begin[DRM_MAX_SEGMENTS + 1], leaf_begin[DRM_MAX_SEGMENTS + 1], wave_off[DRM_MAX_SEGMENTS];
};
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    rnea_backward_fan_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, FanArgs fa, int cap, int n,
                             int n_slots, int flags, const float *__restrict__ q, const float *__restrict__ qd,
                             const float *__restrict__ qdd, const float *__restrict__ gtau, int64_t B, float *__restrict__ gq,
                             float *__restrict__ gqd, float *__restrict__ gqdd, uint64_t param_mask,
                             float *__restrict__ partials, uint32_t magic_q, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NV = cap * DRM_OPF_STRIDE;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int64_t n_tiles = (B + WAVE - 1) / WAVE;
    const int Sq = pad_odd(n), region = round4(WAVE * Sq);
    float *lq = smem, *lqd = lq + region, *lqdd = lqd + region, *lgt = lqdd + region;
    float *lsl = lgt + region;                                      // slots [slot][36][64], shared
    float *lacc = smem + fa.wave_off[wave];                         // this wavefront's sums of the constant gradients
    float *rec = lacc + NV + lane;                                  // its records: [op - a][2][64], then [leaf][18][64]
    const int a = fa.seg_begin[wave], b = fa.seg_begin[wave + 1], leaf0 = fa.leaf_begin[wave];
    float *rec_leaf = rec + (b - a) * (TRIG_FLOATS * WAVE);
    float *slot = lsl + lane;
    const int32_t *ctl = ops_i + DRM_OPI_CTRL * cap;
    for (int i = (int)lane; i < NV; i += WAVE) lacc[i] = 0.0f;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * WAVE;
        const int64_t left = B - b0;
        const int rows = left < WAVE ? (int)left : WAVE;
        const bool full = rows == WAVE, fast = full && (n & 1);
        const bool live = (int)lane < rows;
        __syncthreads(); // the previous tile's gradients have left, its slots are dead
        if (wave == 0) tile_load<0>(q + b0 * n, rows, n, magic_q, lq, lane, fast && (align & AL_Q), full && (align & AL_Q));
        if (wave == 1 % fa.n_seg) tile_load<0>(qd + b0 * n, rows, n, magic_q, lqd, lane, fast && (align & AL_QD), full && (align & AL_QD));
        if (qdd && wave == 2 % fa.n_seg)
            tile_load<0>(qdd + b0 * n, rows, n, magic_q, lqdd, lane, fast && (align & AL_QDD), full && (align & AL_QDD));
        if (wave == 3 % fa.n_seg) tile_load<0>(gtau + b0 * n, rows, n, magic_q, lgt, lane, fast && (align & AL_TAU), full && (align & AL_TAU));
        for (int s = wave; s < n_slots * SLOT_FLOATS; s += fa.n_seg) slot[s * WAVE] = 0.0f;
        __syncthreads();

        const unsigned row = lane * Sq;
        const bool has_qdd = qdd != nullptr;
        auto qf = [&](int d, float &x, float &v, float &c) {
            x = live ? lq[row + d] : 0.0f; // zeros, not stale LDS, past a partial tile
            v = lqd[row + d];
            c = has_qdd ? lqdd[row + d] : 0.0f;
        };
        auto gt = [&](int d) -> float { return live ? lgt[row + d] : 0.0f; };
        auto where = [&](int k, int off) -> float * {
            if (off == 24) return rec + (k - a) * (TRIG_FLOATS * WAVE);
            const int leaf = (int)(((uint32_t)ctl[k]) >> 26) - leaf0;
            return rec_leaf + (leaf * LEAF_FLOATS + (off == 0 ? 0 : 12)) * WAVE;
        };
        auto park = [&](int k, int off, const float *v, int cnt) {
            float *r = where(k, off);
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) r[i * WAVE] = v[i];
        };
        auto unpark = [&](int k, int off, float *v, int cnt) {
            const float *r = where(k, off);
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) v[i] = r[i * WAVE];
        };
        auto slot_put = [&](int s, int off, const float *v, int cnt) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) slot[(s * SLOT_FLOATS + off + i) * WAVE] = v[i];
        };
        auto slot_get = [&](int s, int off, float *v, int cnt) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) v[i] = slot[(s * SLOT_FLOATS + off + i) * WAVE];
        };
        auto slot_add = [&](int s, int off, const float *v, int cnt) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) slot[(s * SLOT_FLOATS + off + i) * WAVE] += v[i];
        };
        auto slot_take = [&](int s, int off, float *v, int cnt) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i < cnt) {
                    v[i] = slot[(s * SLOT_FLOATS + off + i) * WAVE];
                    slot[(s * SLOT_FLOATS + off + i) * WAVE] = 0.0f;
                }
        };
        auto gout = [&](int d, float x, float v, float c) { lq[row + d] = x; lqd[row + d] = v; lqdd[row + d] = c; };
        auto param_out = [&](int k, const float *g) { // wave-uniform call: only for the ops param_mask selects
            wave_sums_lane63<DRM_OPF_DAMP + 1>(lane, [&](int j) { return live ? g[j] : 0.0f; }, // lanes past a partial tile hold garbage
                                               [&](int j, float total) { lacc[k * DRM_OPF_STRIDE + j] += total; }); // tiles in this wavefront's fixed order
        };
        // a short serial segment (a finger): the unrolled walk that parks nothing; anything else: the loops
        if (!rnea_backward_walk_short<BWD_SHORT_OPS>(ops_f, ctl, fa.p_end, a, b, flags, param_mask, gq != nullptr, qf, gt, slot_put,
                                                     slot_get, gout, param_out))
            rnea_backward_walk(ops_f, ctl, fa.p_end, a, b, flags, param_mask, gq != nullptr, qf, gt, park, unpark, slot_put, slot_get,
                               slot_add, slot_take, gout, param_out);
        __syncthreads();
        if (gq) {
            if (wave == 0) tile_store<0>(gq + b0 * n, rows, n, magic_q, lq, lane, fast && (align & AL_POS), full && (align & AL_POS));
            if (wave == 1 % fa.n_seg) tile_store<0>(gqd + b0 * n, rows, n, magic_q, lqd, lane, fast && (align & AL_QUAT), full && (align & AL_QUAT));
            if (wave == 2 % fa.n_seg) tile_store<0>(gqdd + b0 * n, rows, n, magic_q, lqdd, lane, fast && (align & AL_LIN), full && (align & AL_LIN));
        }
    }
    wave_lds_sync();
    float *prow = partials + ((int64_t)blockIdx.x * fa.n_seg + wave) * NV;
    for (int i = (int)lane; i < NV; i += WAVE) prow[i] = lacc[i];
}

// grad_ops_f[e] = sum over the partial rows of column e, in a fixed order (drm_common.hpp column_sum)
__global__ void __launch_bounds__(WAVE *REDUCE_WAVES)
    rnea_backward_reduce_kernel(const float *__restrict__ partials, int n_rows, int cap, float *__restrict__ grad_ops_f) {
    __shared__ float lds[REDUCE_WAVES][WAVE];
    const int NV = cap * DRM_OPF_STRIDE, e = (int)blockIdx.x * WAVE + (int)(threadIdx.x & 63u);
    const float total = column_sum(partials, n_rows, NV, e, e < NV, lds);
    if (threadIdx.x < WAVE && e < NV) grad_ops_f[e] = total;
}

// Serial-chain ("arm") specialisation, full tiles only: drm_sample.hpp rnea_backward_chain (per-link forces and their
// adjoints in registers, motions recovered on the way back instead of stored), constants staged once per wave in LDS,
// gradient tiles staged over the dead input tiles.  Same persistent-wave structure and batch reduction as the
// generic kernel.  LINKS: the links the sweeps visit (NJ when the walk ends at its last moving joint — the host folded the
// fixed tail into it, flatten.fold_link_table, which it only does when nothing is learnable — else CAP).
template <int CAP, int NJ, int LINKS>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    rnea_backward_arm_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                             const float *__restrict__ qdd, const float *__restrict__ gtau, int n_tiles, int flags,
                             uint64_t param_mask, float *__restrict__ gq, float *__restrict__ gqd,
                             float *__restrict__ gqdd, float *__restrict__ partials) {
    static_assert(NJ & 1, "odd row widths only (linear LDS image)");
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, Q_FLOATS = round4(WAVE * NJ);
    constexpr int PER_WAVE = C_FLOATS + 3 * Q_FLOATS; // constants + the three gradient tiles
    constexpr int NV = CAP * DRM_OPF_STRIDE, NACC = NV / WAVE;
    __shared__ __attribute__((aligned(16))) float smem[MAX_WAVES_PER_BLOCK * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int wave_id = (int)blockIdx.x * MAX_WAVES_PER_BLOCK + wave, n_waves = (int)gridDim.x * MAX_WAVES_PER_BLOCK;
    float *lc = smem + wave * PER_WAVE;
    float *lq = lc + C_FLOATS, *lqd = lq + Q_FLOATS, *lqdd = lqd + Q_FLOATS; // staging of grad_q / grad_qd / grad_qdd

    float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    float acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = 0.0f;
    // ONE learnable link (the common case: system identification of a payload or of one link): its 26 constant gradients
    // are summed per LANE over this wave's tiles and reduced across the wave once at the end — 26 adds per tile instead of
    // 26 DPP reductions (230 VALU).  Several learnable links keep the per-tile reduction (a register set each would not fit).
    const bool single = __builtin_popcountll(param_mask) == 1;
    const int single_k = single ? __builtin_ctzll(param_mask) : 0;
    float psum[DRM_OPF_DAMP + 1];
#pragma unroll
    for (int j = 0; j < DRM_OPF_DAMP + 1; ++j) psum[j] = 0.0f;

    for (int tile = wave_id; tile < n_tiles; tile += n_waves) {
        const int64_t b0 = (int64_t)tile * WAVE;
        // every lane reads its own rows of q / qd / qdd / grad_tau straight into registers (28 contiguous bytes per lane and
        // array: the cache lines are shared by neighbouring lanes; no LDS round trip, as in the forward arm kernels)
        float qv[NJ], qdv[NJ], qddv[NJ], gtv[NJ];
        {
            const int64_t r0 = (b0 + lane) * NJ;
#pragma unroll
            for (int d = 0; d < NJ; ++d) qv[d] = q[r0 + d];
#pragma unroll
            for (int d = 0; d < NJ; ++d) qdv[d] = qd[r0 + d];
#pragma unroll
            for (int d = 0; d < NJ; ++d) qddv[d] = qdd ? qdd[r0 + d] : 0.0f;
#pragma unroll
            for (int d = 0; d < NJ; ++d) gtv[d] = gtau[r0 + d];
        }
        wave_lds_sync(); // the previous tile's staged gradients have left the LDS tiles
        float add[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a) add[a] = 0.0f;
        rnea_backward_chain<LINKS, NJ>(
            [&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, flags & DRM_RNEA_GRAVITY,
            flags & DRM_RNEA_DAMPING, param_mask, gq != nullptr, qv, qdv, qddv, gtv,
            [&](int d, float a, float v, float c) {
                lq[lane * NJ + d] = a; lqd[lane * NJ + d] = v; lqdd[lane * NJ + d] = c;
            },
            [&](int k, const float *g) {
                if (single) {
#pragma unroll
                    for (int j = 0; j < DRM_OPF_DAMP + 1; ++j) psum[j] += g[j];
                    return;
                }
#pragma unroll
                for (int j = 0; j < DRM_OPF_DAMP + 1; ++j) {
                    const float total = wave_sum_lane63(g[j]);
                    const int idx = k * DRM_OPF_STRIDE + j;
                    const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, total), 63));
                    if (lane == (unsigned)(idx % WAVE)) add[idx / WAVE] = s;
                }
            });
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] += add[a];
        if (gq) {
            wave_lds_sync();
            tile_store<NJ>(gq + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
            tile_store<NJ>(gqd + b0 * NJ, WAVE, NJ, 0u, lqd, lane, true);
            tile_store<NJ>(gqdd + b0 * NJ, WAVE, NJ, 0u, lqdd, lane, true);
        }
    }
    if (single) {
#pragma unroll
        for (int j = 0; j < DRM_OPF_DAMP + 1; ++j) {
            const float total = wave_sum_lane63(psum[j]);
            const int idx = single_k * DRM_OPF_STRIDE + j;
            const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, total), 63));
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                if (idx / WAVE == a && lane == (unsigned)(idx % WAVE)) acc[a] = s;
        }
    }
    float *prow = partials + (int64_t)wave_id * NV;
#pragma unroll
    for (int a = 0; a < NACC; ++a) prow[a * WAVE + (int)lane] = acc[a];
}

// A HAND (DRM_WALK_FINGERS: K serial chains of L revolute ops off the root): every finger is a short arm, so it takes the arm
// form of the adjoint walk (rnea_backward_chain<L, L>: nothing stored per link).  A block of K wavefronts owns a 64-sample tile
// at a time (persistent blocks), wavefront w walks finger w: its L table rows in LDS, its L columns of q / qd / qdd / grad_tau
// read per lane and its L columns of the three gradients written per lane (one 16-byte access per sample and array at L = 4);
// the block's row of constant-gradient sums lives in global memory, each wavefront adding to the slice of its own ops.
// The loop form (rnea_backward_fan_kernel + rnea_backward_walk_short) takes 436 us per 2^20 samples of the Allegro hand.
#ifndef DRM_BWD_FINGERS_WAVES
#define DRM_BWD_FINGERS_WAVES 3 /* 162-166 VGPRs: three wavefronts per SIMD (TriFinger: 141 -> 103 us against two) */
#endif
template <int L>
__global__ void __launch_bounds__(WAVE * 4) __attribute__((amdgpu_waves_per_eu(DRM_BWD_FINGERS_WAVES, DRM_BWD_FINGERS_WAVES)))
    rnea_backward_fingers_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                                 const float *__restrict__ qdd, const float *__restrict__ gtau, int n, int cap, int n_tiles, int flags,
                                 uint64_t param_mask, float *__restrict__ gq, float *__restrict__ gqd, float *__restrict__ gqdd,
                                 float *__restrict__ partials, int vec) {
    constexpr int C_FLOATS = L * DRM_OPF_STRIDE, G_FLOATS = 3 * L * WAVE;
    __shared__ __attribute__((aligned(16))) float smem[4 * (C_FLOATS + G_FLOATS)];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * (C_FLOATS + G_FLOATS);
    float *lg = lc + C_FLOATS + lane; // this lane's gradients until the walk is through: [array][joint][64] (each lane its own words)
    if (lane < (unsigned)(L * (DRM_OPF_STRIDE / 4)))
        reinterpret_cast<float4 *>(lc)[lane] = reinterpret_cast<const float4 *>(ops_f + (size_t)wave * C_FLOATS)[lane];
    float *prow = partials + (int64_t)blockIdx.x * cap * DRM_OPF_STRIDE + wave * C_FLOATS; // this finger's slice of the block's row
    for (int i = (int)lane; i < C_FLOATS; i += WAVE) prow[i] = 0.0f;
    if (wave == 0) { // the rows of the table's padding ops belong to nobody: zero
        const int K = (int)(blockDim.x >> 6);
        float *pad = partials + (int64_t)blockIdx.x * cap * DRM_OPF_STRIDE;
        for (int i = K * C_FLOATS + (int)lane; i < cap * DRM_OPF_STRIDE; i += WAVE) pad[i] = 0.0f;
    }
    __builtin_amdgcn_s_waitcnt(0);
    wave_lds_sync();
    const uint64_t mask = (param_mask >> (wave * L)) & ((1ull << L) - 1ull);
#pragma unroll 1
    for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
        const int64_t r0 = ((int64_t)tile * WAVE + lane) * n + wave * L;
        float qv[L], qdv[L], qddv[L], gtv[L];
        auto load = [&](const float *src, float (&dst)[L]) {
            if (L == 4 && vec) {
                const float4 a = *reinterpret_cast<const float4 *>(src + r0);
                dst[0] = a.x; dst[1] = a.y; dst[2 % L] = a.z; dst[3 % L] = a.w;
            } else {
#pragma unroll
                for (int d = 0; d < L; ++d) dst[d] = src[r0 + d];
            }
        };
        load(q, qv); load(qd, qdv); load(gtau, gtv);
        if (qdd) load(qdd, qddv);
        else {
#pragma unroll
            for (int d = 0; d < L; ++d) qddv[d] = 0.0f;
        }
        rnea_backward_chain<L, L>(
            [&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, mask,
            gq != nullptr, qv, qdv, qddv, gtv,
            [&](int d, float a, float v, float c) { lg[d * WAVE] = a; lg[(L + d) * WAVE] = v; lg[(2 * L + d) * WAVE] = c; },
            [&](int k, const float *g) { // wave-uniform call: only for the ops the mask selects
                wave_sums_lane63<DRM_OPF_DAMP + 1>(lane, [&](int j) { return g[j]; },
                                                   [&](int j, float total) { prow[k * DRM_OPF_STRIDE + j] += total; }); // tiles in this block's fixed order
            });
        if (gq) {
            auto store = [&](float *dst, int arr) {
                float v[L];
#pragma unroll
                for (int d = 0; d < L; ++d) v[d] = lg[(arr * L + d) * WAVE];
                if (L == 4 && vec) *reinterpret_cast<float4 *>(dst + r0) = make_float4(v[0], v[1], v[2 % L], v[3 % L]);
                else {
#pragma unroll
                    for (int d = 0; d < L; ++d) dst[r0 + d] = v[d];
                }
            };
            store(gq, 0); store(gqd, 1); store(gqdd, 2);
        }
    }
}

static size_t rnea_backward_lds_floats(int n, int n_slots, int cap, int n_ops, int n_leaves, bool park_hbm) {
    return (size_t)4 * round4(WAVE * pad_odd(n)) + (size_t)cap * DRM_OPF_STRIDE + (size_t)n_slots * SLOT_FLOATS * WAVE +
           (park_hbm ? 0 : (size_t)record_floats(n_ops, n_leaves));
}

} // namespace drm

using namespace drm;

extern "C" int64_t drm_rnea_backward_scratch_floats(int64_t B, int32_t capacity, int32_t n_dofs, int32_t n_slots) {
    if (B < 0 || capacity < 1 || capacity > DRM_MAX_OPS || n_dofs < 1 || n_dofs > DRM_MAX_DOFS) return 0;
    const int64_t waves = backward_waves(B, MAX_WAVES_PER_BLOCK);
    // rows of partial sums: one per wavefront (+ a ragged tail's); the fanned-out launch has one per tile AND segment
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    int64_t rows = tiles * DRM_MAX_SEGMENTS < BWD_MAX_WAVES ? tiles * DRM_MAX_SEGMENTS : BWD_MAX_WAVES;
    if (rows < waves) rows = waves;
    int64_t floats = (rows + MAX_WAVES_PER_BLOCK) * capacity * DRM_OPF_STRIDE;
    // records parked in HBM when they do not fit LDS; sized for the worst case (every op a leaf): the query does not see the walk
    if (rnea_backward_lds_floats(n_dofs, n_slots, capacity, capacity, capacity, false) * sizeof(float) > (size_t)MAX_LDS_BYTES)
        floats += waves * (int64_t)capacity * REC_FLOATS * WAVE;
    return floats;
}

extern "C" int drm_rnea_backward(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B,
                                 int32_t flags, const float *grad_tau, uint64_t param_mask, float *grad_q, float *grad_qd,
                                 float *grad_qdd, float *grad_ops_f, float *scratch, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !qd || !grad_tau) return fail(DRM_ERR_INVALID, "q / qd / grad_tau must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if ((param_mask != 0) != (grad_ops_f != nullptr))
        return fail(DRM_ERR_INVALID, "grad_ops_f must be given exactly when param_mask selects ops");
    const bool want_q = grad_q != nullptr;
    if (want_q != (grad_qd != nullptr) || want_q != (grad_qdd != nullptr))
        return fail(DRM_ERR_INVALID, "grad_q, grad_qd and grad_qdd are produced together: give all three or none");
    if (!want_q && !grad_ops_f) return fail(DRM_ERR_INVALID, "nothing to compute");
    if (!scratch) return fail(DRM_ERR_INVALID, "scratch must not be NULL (drm_rnea_backward_scratch_floats)");
    if (w->capacity < 64 && (param_mask >> w->capacity)) return fail(DRM_ERR_INVALID, "param_mask selects ops beyond the walk's capacity");
    const int n = w->n_dofs, cap = w->capacity, n_leaves = DRM_WALK_LEAVES(w->shape);
    if (w->n_ops > 0 && (n_leaves < 1 || n_leaves > w->n_ops || n_leaves > 64))
        return fail(DRM_ERR_INVALID, "walk without its leaf count (drm_walk.shape bits 16..23; host built for an older ABI?)");
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        if (grad_ops_f) {
            hipError_t e = hipMemsetAsync(grad_ops_f, 0, sizeof(float) * cap * DRM_OPF_STRIDE, s);
            if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
        return DRM_OK;
    }
    float *partials = scratch;
    if (w->special[DRM_SPECIAL_RNEA_BACKWARD] && B >= WAVE && B / WAVE < 0x7fffffffLL && (((uintptr_t)w->ops_f) & 15u) == 0) {
        // a per-robot straight-line kernel built for exactly this walk (csrc/drm_static.hpp, specialize.py): full tiles on as many
        // wavefronts as the device holds at once, any pointer alignment; the ragged tail through the loop kernel, its row of
        // partial sums appended
        hipFunction_t fn = (hipFunction_t)w->special[DRM_SPECIAL_RNEA_BACKWARD];
        int n_tiles = (int)(B / WAVE), fl = (int)flags, grid = 0;
        rc = resident_blocks_module(fn, WAVE, grid);
        if (rc) return rc;
        if (grid > BWD_MAX_WAVES) grid = BWD_MAX_WAVES;
        if (grid > n_tiles) grid = n_tiles;
        void *args[] = {(void *)&w->ops_f, (void *)&q, (void *)&qd, (void *)&qdd, (void *)&grad_tau, (void *)&n_tiles, (void *)&fl,
                        (void *)&param_mask, (void *)&grad_q, (void *)&grad_qd, (void *)&grad_qdd, (void *)&partials};
        hipError_t e = hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, WAVE, 1, 1, 0, s, args, nullptr);
        if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_rnea_backward_static): %s", hipGetErrorString(e));
        int rows = grid;
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done < B) {
            const size_t need = rnea_backward_lds_floats(n, w->n_slots, cap, w->n_ops, n_leaves, false) * sizeof(float);
            const bool park_tail = need > (size_t)MAX_LDS_BYTES; // (one wavefront's records behind the rows of partial sums)
            Geometry gt;
            rc = make_geometry(B - done, (int)rnea_backward_lds_floats(n, w->n_slots, cap, w->n_ops, n_leaves, park_tail), gt);
            if (rc) return rc;
            gt.grid = dim3(1);
            gt.block = dim3(WAVE);
            gt.lds_bytes = (size_t)gt.lds_per_wave * sizeof(float);
            const uint32_t al = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(grad_tau, AL_TAU);
            float *park = partials + (int64_t)(rows + 1) * cap * DRM_OPF_STRIDE;
#define DRM_LAUNCH_TAIL(HBM)                                                                                                     \
    {                                                                                                                            \
        rc = ensure_lds(rnea_backward_kernel<HBM>, gt.lds_bytes);                                                                \
        if (rc) return rc;                                                                                                       \
        hipLaunchKernelGGL((rnea_backward_kernel<HBM>), gt.grid, gt.block, gt.lds_bytes, s, w->ops_f, w->ops_i, cap, (int)w->n_ops, \
                           n_leaves, n, (int)w->n_slots, (int)flags, q + done * n, qd + done * n, qdd ? qdd + done * n : nullptr,   \
                           grad_tau + done * n, B - done, grad_q ? grad_q + done * n : nullptr,                                   \
                           grad_qd ? grad_qd + done * n : nullptr, grad_qdd ? grad_qdd + done * n : nullptr, param_mask,           \
                           partials + (int64_t)rows * cap * DRM_OPF_STRIDE, park, div_magic(n), gt.lds_per_wave,                   \
                           al & ~(AL_POS | AL_QUAT | AL_LIN));                                                                    \
    }
            if (park_tail) DRM_LAUNCH_TAIL(true) else DRM_LAUNCH_TAIL(false)
#undef DRM_LAUNCH_TAIL
            rc = launched();
            if (rc) return rc;
            rows += 1;
        }
        if (grad_ops_f) {
            hipLaunchKernelGGL(rnea_backward_reduce_kernel, dim3((unsigned)(cap * DRM_OPF_STRIDE / WAVE)), dim3(WAVE * REDUCE_WAVES), 0, s,
                               partials, rows, cap, grad_ops_f);
            rc = launched();
        }
        return rc;
    }
#ifndef DRM_NO_ARM_KERNEL
    {
        const uintptr_t ptrs = (uintptr_t)q | (uintptr_t)qd | (uintptr_t)qdd | (uintptr_t)grad_tau | (uintptr_t)grad_q |
                               (uintptr_t)grad_qd | (uintptr_t)grad_qdd | (uintptr_t)w->ops_f;
        if ((w->shape & DRM_WALK_ARM_CHAIN) && cap == 8 && n == 7 && (ptrs & 15u) == 0 && B >= WAVE &&
            B / WAVE < 0x7fffffffLL) {
            // 7-DoF arms: full tiles through the chain kernel, the ragged tail (if any) through the generic one with
            // its own rows of partial sums appended after the chain kernel's
            int n_tiles = (int)(B / WAVE);
            const int64_t done = (int64_t)n_tiles * WAVE;
            if (w->special[DRM_SPECIAL_RNEA_BACKWARD_ARM2] && w->special[DRM_SPECIAL_RNEA_BACKWARD_ARM] && param_mask == 0 && want_q &&
                n_tiles / 2 >= DRM_ARM_STATIC_MIN_PAIRS) {
                // ABI 11: two samples per lane for the pairs of tiles of a large launch; an odd tile and the tail follow below
                int n_pairs = n_tiles / 2, fl = (int)flags;
                void *args[] = {(void *)&q, (void *)&qd, (void *)&qdd, (void *)&grad_tau, (void *)&n_pairs, (void *)&fl,
                                (void *)&grad_q, (void *)&grad_qd, (void *)&grad_qdd};
                hipError_t e = hipModuleLaunchKernel((hipFunction_t)w->special[DRM_SPECIAL_RNEA_BACKWARD_ARM2], (unsigned)n_pairs, 1, 1, WAVE, 1, 1, 0, s, args, nullptr);
                if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_rnea_backward_arm2_static): %s", hipGetErrorString(e));
                const int64_t done2 = (int64_t)n_pairs * 2 * WAVE;
                if (done2 == B) return DRM_OK;
                drm_walk rest = *w;
                rest.special[DRM_SPECIAL_RNEA_BACKWARD_ARM2] = nullptr;
                return drm_rnea_backward(&rest, q + done2 * n, qd + done2 * n, qdd ? qdd + done2 * n : nullptr, B - done2, flags, grad_tau + done2 * n,
                                         param_mask, grad_q + done2 * n, grad_qd + done2 * n, grad_qdd + done2 * n, grad_ops_f, scratch, stream);
            }
            if (w->special[DRM_SPECIAL_RNEA_BACKWARD_ARM] && param_mask == 0 && want_q) {
                // input gradients of a constant model through this arm's own kernel, its constants folded into the instruction
                // stream (csrc/drm_arm_static.hpp, specialize.py): nothing is summed over the batch, so no partial rows
                int fl = (int)flags;
                void *args[] = {(void *)&q, (void *)&qd, (void *)&qdd, (void *)&grad_tau, (void *)&n_tiles, (void *)&fl,
                                (void *)&grad_q, (void *)&grad_qd, (void *)&grad_qdd};
                hipError_t e = hipModuleLaunchKernel((hipFunction_t)w->special[DRM_SPECIAL_RNEA_BACKWARD_ARM], (unsigned)n_tiles, 1, 1, WAVE, 1, 1, 0, s, args, nullptr);
                if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_rnea_backward_arm_static): %s", hipGetErrorString(e));
                if (done == B) return DRM_OK;
                drm_walk rest = *w;
                rest.special[DRM_SPECIAL_RNEA_BACKWARD_ARM] = nullptr;
                return drm_rnea_backward(&rest, q + done * n, qd + done * n, qdd ? qdd + done * n : nullptr, B - done, flags, grad_tau + done * n,
                                         param_mask, grad_q + done * n, grad_qd + done * n, grad_qdd + done * n, grad_ops_f, scratch, stream);
            }
            const int waves_a = backward_waves(done, MAX_WAVES_PER_BLOCK);
            if (w->special[DRM_SPECIAL_RNEA_BACKWARD_ARM_PARAM] && param_mask != 0 && param_mask == (uint64_t)w->reserved0) {
                // ABI 11: this arm's own reverse-mode kernel for exactly this set of learnable blocks (csrc/drm_arm_static.hpp
                // rnea_backward_arm_param_static_body, specialize.py): the constant blocks of the table folded into the instruction
                // stream, the learnable ones read from ops_f; same rows of partial sums, same reduction below
                int fl = (int)flags;
                void *args[] = {(void *)&w->ops_f, (void *)&q, (void *)&qd, (void *)&qdd, (void *)&grad_tau, (void *)&n_tiles, (void *)&fl,
                                (void *)&grad_q, (void *)&grad_qd, (void *)&grad_qdd, (void *)&partials};
                hipError_t e = hipModuleLaunchKernel((hipFunction_t)w->special[DRM_SPECIAL_RNEA_BACKWARD_ARM_PARAM],
                                                     (unsigned)(waves_a / MAX_WAVES_PER_BLOCK), 1, 1, WAVE * MAX_WAVES_PER_BLOCK, 1, 1, 0, s, args, nullptr);
                if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_rnea_backward_arm_param_static): %s", hipGetErrorString(e));
            } else if (arm_links(w) == 7)
                hipLaunchKernelGGL((rnea_backward_arm_kernel<8, 7, 7>), dim3((unsigned)(waves_a / MAX_WAVES_PER_BLOCK)),
                                   dim3(WAVE * MAX_WAVES_PER_BLOCK), 0, s, w->ops_f, q, qd, qdd, grad_tau, n_tiles, (int)flags,
                                   param_mask, grad_q, grad_qd, grad_qdd, partials);
            else
                hipLaunchKernelGGL((rnea_backward_arm_kernel<8, 7, 8>), dim3((unsigned)(waves_a / MAX_WAVES_PER_BLOCK)),
                                   dim3(WAVE * MAX_WAVES_PER_BLOCK), 0, s, w->ops_f, q, qd, qdd, grad_tau, n_tiles, (int)flags,
                                   param_mask, grad_q, grad_qd, grad_qdd, partials);
            rc = launched();
            if (rc) return rc;
            int rows = waves_a;
            if (done < B) {
                // tail: < 64 rows, one wave, through the generic kernel (LDS parking: an arm always fits)
                Geometry gt;
                rc = make_geometry(B - done, (int)rnea_backward_lds_floats(n, w->n_slots, cap, w->n_ops, n_leaves, false), gt);
                if (rc) return rc;
                gt.grid = dim3(1);
                gt.block = dim3(WAVE);
                gt.lds_bytes = (size_t)gt.lds_per_wave * sizeof(float);
                rc = ensure_lds(rnea_backward_kernel<false>, gt.lds_bytes);
                if (rc) return rc;
                const uint32_t al = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(grad_tau, AL_TAU);
                hipLaunchKernelGGL((rnea_backward_kernel<false>), gt.grid, gt.block, gt.lds_bytes, s, w->ops_f, w->ops_i,
                                   cap, (int)w->n_ops, n_leaves, n, (int)w->n_slots, (int)flags, q + done * n, qd + done * n,
                                   qdd ? qdd + done * n : nullptr, grad_tau + done * n, B - done,
                                   grad_q ? grad_q + done * n : nullptr, grad_qd ? grad_qd + done * n : nullptr,
                                   grad_qdd ? grad_qdd + done * n : nullptr, param_mask,
                                   partials + (int64_t)waves_a * cap * DRM_OPF_STRIDE, (float *)nullptr, div_magic(n),
                                   gt.lds_per_wave, al & ~(AL_POS | AL_QUAT | AL_LIN));
                rc = launched();
                if (rc) return rc;
                rows += 1;
            }
            if (grad_ops_f) {
                hipLaunchKernelGGL(rnea_backward_reduce_kernel, dim3((unsigned)(cap * DRM_OPF_STRIDE / WAVE)), dim3(WAVE * REDUCE_WAVES), 0, s, partials, rows, cap,
                                   grad_ops_f);
                rc = launched();
            }
            return rc;
        }
    }
#endif
    {   // a hand (DRM_WALK_FINGERS) / an arm that carries a hand (Panda with gripper, Jaco, iiwa7 + Allegro): full tiles through
        // the straight-line kernels, the ragged tail (< 64 rows, one wavefront) through the loop kernel with its row of partial
        // sums appended
        int rows = 0;
        int64_t done = 0;
#ifndef DRM_NO_FINGERS_KERNEL
        if ((w->shape & DRM_WALK_FINGERS) && B >= WAVE && B / WAVE < 0x7fffffffLL && (((uintptr_t)w->ops_f) & 15u) == 0) {
            const int K = DRM_WALK_AH_K(w->shape), L = DRM_WALK_AH_L(w->shape);
            if (K * L == w->n_ops && n == w->n_ops && K >= 2 && K <= 4 && L >= 2 && L <= 4 && cap >= w->n_ops) {
                const int n_tiles = (int)(B / WAVE);
                int resident = 0;
                rc = L == 2 ? resident_blocks(rnea_backward_fingers_kernel<2>, WAVE * K, 0, resident)
                            : L == 3 ? resident_blocks(rnea_backward_fingers_kernel<3>, WAVE * K, 0, resident)
                                     : resident_blocks(rnea_backward_fingers_kernel<4>, WAVE * K, 0, resident);
                if (rc) return rc;
                if (resident > BWD_MAX_WAVES) resident = BWD_MAX_WAVES;
                const int grid = n_tiles < resident ? n_tiles : resident;
                const int vec = (n % 4 == 0) && ((((uintptr_t)q | (uintptr_t)qd | (uintptr_t)qdd | (uintptr_t)grad_tau | (uintptr_t)grad_q |
                                                   (uintptr_t)grad_qd | (uintptr_t)grad_qdd) & 15u) == 0);
#define X(l)                                                                                                                    \
    if (L == l)                                                                                                                  \
        hipLaunchKernelGGL((rnea_backward_fingers_kernel<l>), dim3((unsigned)grid), dim3(WAVE * K), 0, s, w->ops_f, q, qd, qdd,    \
                           grad_tau, n, cap, n_tiles, (int)flags, param_mask, grad_q, grad_qd, grad_qdd, partials, vec);
                X(2) X(3) X(4)
#undef X
                rows = grid;
                done = (int64_t)n_tiles * WAVE;
            }
        }
#endif
        if (done == 0)
            done = launch_rnea_backward_arm_hand(w, q, qd, qdd, grad_tau, B, (int)flags, param_mask, grad_q, grad_qd, grad_qdd,
                                                 partials, rows, s);
        if (done > 0) {
            rc = launched();
            if (rc) return rc;
            if (done < B) {
                const size_t need = rnea_backward_lds_floats(n, w->n_slots, cap, w->n_ops, n_leaves, false) * sizeof(float);
                if (need > (size_t)MAX_LDS_BYTES) return fail(DRM_ERR_UNSUPPORTED, "the tail of this walk does not fit LDS");
                Geometry gt;
                rc = make_geometry(B - done, (int)(need / sizeof(float)), gt);
                if (rc) return rc;
                gt.grid = dim3(1);
                gt.block = dim3(WAVE);
                gt.lds_bytes = (size_t)gt.lds_per_wave * sizeof(float);
                rc = ensure_lds(rnea_backward_kernel<false>, gt.lds_bytes);
                if (rc) return rc;
                const uint32_t al = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(grad_tau, AL_TAU);
                hipLaunchKernelGGL((rnea_backward_kernel<false>), gt.grid, gt.block, gt.lds_bytes, s, w->ops_f, w->ops_i, cap,
                                   (int)w->n_ops, n_leaves, n, (int)w->n_slots, (int)flags, q + done * n, qd + done * n,
                                   qdd ? qdd + done * n : nullptr, grad_tau + done * n, B - done,
                                   grad_q ? grad_q + done * n : nullptr, grad_qd ? grad_qd + done * n : nullptr,
                                   grad_qdd ? grad_qdd + done * n : nullptr, param_mask,
                                   partials + (int64_t)rows * cap * DRM_OPF_STRIDE, (float *)nullptr, div_magic(n), gt.lds_per_wave,
                                   al & ~(AL_POS | AL_QUAT | AL_LIN));
                rc = launched();
                if (rc) return rc;
                rows += 1;
            }
            if (grad_ops_f) {
                hipLaunchKernelGGL(rnea_backward_reduce_kernel, dim3((unsigned)(cap * DRM_OPF_STRIDE / WAVE)), dim3(WAVE * REDUCE_WAVES), 0, s,
                                   partials, rows, cap, grad_ops_f);
                rc = launched();
            }
            return rc;
        }
    }
    // fanned out over the segments when the walk has several, none of the prefix ops is learnable and a block's LDS fits twice
    // per CU or better than the single-wavefront form would
    if (w->n_segments > 1 && segments_ok(w) && w->prefix_end < 64 && !(param_mask & ((1ull << w->prefix_end) - 1ull))) {
        FanArgs fa;
        fa.n_seg = w->n_segments; fa.p_end = w->prefix_end;
        const size_t shared = (size_t)4 * round4(WAVE * pad_odd(n)) + (size_t)w->n_slots * SLOT_FLOATS * WAVE;
        size_t off = shared;
        bool ok = true;
        for (int sgm = 0; sgm <= DRM_MAX_SEGMENTS; ++sgm) {
            fa.seg_begin[sgm] = sgm <= w->n_segments ? w->seg_begin[sgm] : w->n_ops;
            fa.leaf_begin[sgm] = sgm <= w->n_segments ? w->seg_leaf_begin[sgm] : n_leaves;
        }
        for (int sgm = 0; sgm < DRM_MAX_SEGMENTS; ++sgm) {
            fa.wave_off[sgm] = (int32_t)off;
            if (sgm < w->n_segments) {
                const int ops_s = fa.seg_begin[sgm + 1] - fa.seg_begin[sgm], leaves_s = fa.leaf_begin[sgm + 1] - fa.leaf_begin[sgm];
                if (ops_s < 0 || leaves_s < 0 || leaves_s > ops_s) ok = false;
                off += (size_t)round4(cap * DRM_OPF_STRIDE + record_floats(ops_s, leaves_s));
            }
        }
        const size_t lds_bytes = off * sizeof(float);
        if (ok && lds_bytes <= (size_t)MAX_LDS_BYTES) {
            const int64_t tiles = (B + WAVE - 1) / WAVE;
            int64_t blocks = BWD_MAX_WAVES / w->n_segments;
            if (blocks > tiles) blocks = tiles;
            rc = ensure_lds(rnea_backward_fan_kernel, lds_bytes);
            if (rc) return rc;
            const uint32_t al = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(grad_tau, AL_TAU) |
                                al16(grad_q, AL_POS) | al16(grad_qd, AL_QUAT) | al16(grad_qdd, AL_LIN);
            hipLaunchKernelGGL(rnea_backward_fan_kernel, dim3((unsigned)blocks), dim3((unsigned)(WAVE * w->n_segments)), lds_bytes, s,
                               w->ops_f, w->ops_i, fa, cap, n, (int)w->n_slots, (int)flags, q, qd, qdd, grad_tau, B, grad_q, grad_qd,
                               grad_qdd, param_mask, partials, div_magic(n), al);
            rc = launched();
            if (rc) return rc;
            if (grad_ops_f) {
                hipLaunchKernelGGL(rnea_backward_reduce_kernel, dim3((unsigned)(cap * DRM_OPF_STRIDE / WAVE)), dim3(WAVE * REDUCE_WAVES), 0, s,
                                   partials, (int)(blocks * w->n_segments), cap, grad_ops_f);
                rc = launched();
            }
            return rc;
        }
    }
    const bool park_hbm = rnea_backward_lds_floats(n, w->n_slots, cap, w->n_ops, n_leaves, false) * sizeof(float) > (size_t)MAX_LDS_BYTES;
    Geometry g;
    rc = make_geometry(B, (int)rnea_backward_lds_floats(n, w->n_slots, cap, w->n_ops, n_leaves, park_hbm), g);
    if (rc) return rc;
    const int wpb = (int)(g.block.x / WAVE);
    const int waves = backward_waves(B, wpb);
    g.grid = dim3((unsigned)(waves / wpb));
    float *park;
    {
        const int64_t tiles = (B + WAVE - 1) / WAVE, w0 = backward_waves(B, MAX_WAVES_PER_BLOCK);
        int64_t rows = tiles * DRM_MAX_SEGMENTS < BWD_MAX_WAVES ? tiles * DRM_MAX_SEGMENTS : BWD_MAX_WAVES;
        if (rows < w0) rows = w0;
        park = scratch + (rows + MAX_WAVES_PER_BLOCK) * cap * DRM_OPF_STRIDE; // behind the rows of partial sums (scratch query)
    }
    const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(grad_tau, AL_TAU) |
                           al16(grad_q, AL_POS) | al16(grad_qd, AL_QUAT) | al16(grad_qdd, AL_LIN);
#define DRM_LAUNCH_RB(HBM)                                                                                             \
    {                                                                                                                  \
        rc = ensure_lds(rnea_backward_kernel<HBM>, g.lds_bytes);                                                       \
        if (rc) return rc;                                                                                             \
        hipLaunchKernelGGL((rnea_backward_kernel<HBM>), g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, cap,      \
                           (int)w->n_ops, n_leaves, n, (int)w->n_slots, (int)flags, q, qd, qdd, grad_tau, B, grad_q, grad_qd,    \
                           grad_qdd, param_mask, partials, park, div_magic(n), g.lds_per_wave, align);                 \
    }
    if (park_hbm) DRM_LAUNCH_RB(true) else DRM_LAUNCH_RB(false)
#undef DRM_LAUNCH_RB
    rc = launched();
    if (rc) return rc;
    if (grad_ops_f) {
        hipLaunchKernelGGL(rnea_backward_reduce_kernel, dim3((unsigned)(cap * DRM_OPF_STRIDE / WAVE)), dim3(WAVE * REDUCE_WAVES), 0, s, partials, waves, cap,
                           grad_ops_f);
        rc = launched();
    }
    return rc;
}
