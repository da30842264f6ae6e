// drm_fk.hip — K1/K4: FK of T target links over a (possibly branching) walk.
//
// Replaces DifferentiableRobotModel.compute_forward_kinematics (robot_model.py:223-248) and, with
// T = all links, compute_forward_kinematics_all_links (robot_model.py:197-221; recursion rigid_body.py:85-127).
//
// Per sample: in q[n] (4 n bytes), out pos[T,3] quat[T,4] (28 T bytes).
// LDS per wave: [ constant rows : CAP * 32 ][ q : 64 (n|1) ][ pos : 64 (3T|1) ][ quat : 64 (4T+1) ][ slots : n_slots * 12 * 64 ]
#include "drm_common.hpp"
#include "drm_sample.hpp"

namespace drm {

template <int CAP>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    fk_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int n, const float *__restrict__ q,
              int64_t B, int T, float *__restrict__ pos, float *__restrict__ quat, uint32_t magic_q, uint32_t magic_p,
              uint32_t magic_r, int lds_per_wave, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx cx;
    if (!wave_begin(B, lds_per_wave, smem, cx)) return;
    const unsigned lane = cx.lane;
    const int Sq = pad_odd(n), Sp = pad_odd(3 * T), Sr = pad_odd(4 * T);
    float *lc = cx.lds;
    float *lq = lc + CAP * DRM_OPF_STRIDE;
    float *lp = lq + round4(WAVE * Sq);
    float *lr = lp + round4(WAVE * Sp);
    float *ls = lr + round4(WAVE * Sr); // save slots: [slot][12][64]

    stage_table<CAP>(ops_f, lc, lane, align & AL_TABLE);
    tile_load<0>(q + cx.b0 * n, cx.rows, n, magic_q, lq, lane, cx.full && (n & 1) && (align & AL_Q), cx.full && (align & AL_Q));
    wave_lds_sync();

    // lanes past a partial tile read zeros (not stale LDS): their angles must not be able to push the wave onto
    // the rare large-angle sincos path, which would change the rounding of the live lanes from run to run
    const bool live = (int)lane < cx.rows;
    const float *qrow = lq + lane * Sq;
    auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; };
    float *prow = lp + lane * Sp;
    float *rrow = lr + lane * Sr;
    auto slot_save = [&](int s, const PoseP &P) {
        float *b = ls + s * (12 * WAVE) + lane;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            b[(4 * c + 0) * WAVE] = P.A[c][0]; b[(4 * c + 1) * WAVE] = P.A[c][1];
            b[(4 * c + 2) * WAVE] = P.B[c][0]; b[(4 * c + 3) * WAVE] = P.B[c][1];
        }
    };
    auto slot_load = [&](int s, PoseP &P) {
        const float *b = ls + s * (12 * WAVE) + lane;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            P.A[c] = f2_make(b[(4 * c + 0) * WAVE], b[(4 * c + 1) * WAVE]);
            P.B[c] = f2_make(b[(4 * c + 2) * WAVE], b[(4 * c + 3) * WAVE]);
        }
    };
    auto emit = [&](int t, const float *p, const float *qt) {
        prow[t * 3 + 0] = p[0]; prow[t * 3 + 1] = p[1]; prow[t * 3 + 2] = p[2];
        rrow[t * 4 + 0] = qt[0]; rrow[t * 4 + 1] = qt[1]; rrow[t * 4 + 2] = qt[2]; rrow[t * 4 + 3] = qt[3];
    };
    fk_walk<CAP>(lc, ops_i, qf, slot_save, slot_load, emit);
    wave_lds_sync();
    tile_store<0>(pos + cx.b0 * 3 * T, cx.rows, 3 * T, magic_p, lp, lane, cx.full && ((3 * T) & 1) && (align & AL_POS),
                  cx.full && (align & AL_POS));
    tile_store<0>(quat + cx.b0 * 4 * T, cx.rows, 4 * T, magic_r, lr, lane, false, cx.full && (align & AL_QUAT));
}

// ---------------------------------------------------------------------------------------------------
// Fan-out FK: T <= 4 targets whose root->target chains share (almost) nothing — the fingertips of a hand that hang
// off a common palm (Allegro, TriFinger).  The merged walk above makes ONE lane compute all T chains of a sample
// one after the other; here a block of T wavefronts owns a tile of 64 samples and wavefront t walks ONLY the chain of
// target t (constants stay wave-uniform), so the serial work per wave is one chain and T waves per SIMD hide each
// other's latencies.  The q tile is loaded once per block into shared LDS, the T x (pos, quat) results of a sample
// are assembled in LDS and leave as one contiguous row-major tile.
// ---------------------------------------------------------------------------------------------------
struct FanoutTables {
    const float *ops_f[4];
    const int32_t *ops_i[4];
};

template <int CAP>
__global__ void __launch_bounds__(WAVE * 4)
    fk_fanout_kernel(FanoutTables tab, int T, int n, const float *__restrict__ q, int64_t B, float *__restrict__ pos,
                     float *__restrict__ quat, uint32_t magic_q, uint32_t magic_p, uint32_t magic_r, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int64_t b0 = (int64_t)blockIdx.x * WAVE;
    const int64_t left = B - b0;
    const int rows = left < WAVE ? (int)left : WAVE;
    const bool full = rows == WAVE;
    const int Sq = pad_odd(n), Sp = pad_odd(3 * T), Sr = pad_odd(4 * T);
    float *lq = smem;
    float *lp = lq + round4(WAVE * Sq);
    float *lr = lp + round4(WAVE * Sp);

    if (wave == 0) tile_load<0>(q + b0 * n, rows, n, magic_q, lq, lane, full && (n & 1) && (align & AL_Q), full && (align & AL_Q));
    __syncthreads();

    // wave-uniform choice of this wave's chain tables (select chain: T <= 4)
    const float *ops_f = wave == 0 ? tab.ops_f[0] : wave == 1 ? tab.ops_f[1] : wave == 2 ? tab.ops_f[2] : tab.ops_f[3];
    const int32_t *ops_i = wave == 0 ? tab.ops_i[0] : wave == 1 ? tab.ops_i[1] : wave == 2 ? tab.ops_i[2] : tab.ops_i[3];
    const bool live = (int)lane < rows;
    const float *qrow = lq + lane * Sq;
    float *prow = lp + lane * Sp + wave * 3;
    float *rrow = lr + lane * Sr + wave * 4;
    auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; };
    auto no_save = [&](int, const PoseP &) {};
    auto no_load = [&](int, PoseP &) {};
    auto emit = [&](int, const float *p, const float *qt) { // the chain's single target = column `wave` of the row
        prow[0] = p[0]; prow[1] = p[1]; prow[2] = p[2];
        rrow[0] = qt[0]; rrow[1] = qt[1]; rrow[2] = qt[2]; rrow[3] = qt[3];
    };
    fk_walk<CAP>(ops_f, ops_i, qf, no_save, no_load, emit);
    __syncthreads();
    // the assembled [64, 3T] and [64, 4T] tiles leave with coalesced stores, one tensor per wave pair
    if (wave == 0)
        tile_store<0>(pos + b0 * 3 * T, rows, 3 * T, magic_p, lp, lane, full && ((3 * T) & 1) && (align & AL_POS),
                      full && (align & AL_POS));
    if (wave == T - 1)
        tile_store<0>(quat + b0 * 4 * T, rows, 4 * T, magic_r, lr, lane, false, full && (align & AL_QUAT));
}

} // namespace drm

using namespace drm;

extern "C" int drm_fk_fanout(const drm_walk *chains, int32_t n_chains, const float *q, int64_t B, float *pos, float *quat,
                             void *stream) {
    if (!chains || n_chains < 2 || n_chains > 4) return fail(DRM_ERR_INVALID, "fan-out FK takes 2 to 4 chains");
    if (!q || !pos || !quat) return fail(DRM_ERR_INVALID, "q / pos / quat must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    FanoutTables tab;
    for (int t = 0; t < 4; ++t) {
        const drm_walk *w = chains + (t < n_chains ? t : 0);
        int rc = check_walk(w);
        if (rc) return rc;
        if (w->capacity != chains[0].capacity || w->n_dofs != chains[0].n_dofs || w->n_slots != 0)
            return fail(DRM_ERR_INVALID, "fan-out chains must share capacity and n_dofs and have no branch points");
        tab.ops_f[t] = w->ops_f;
        tab.ops_i[t] = w->ops_i;
    }
    if (B == 0) return DRM_OK;
    const int n = chains[0].n_dofs, T = n_chains;
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    const size_t lds = sizeof(float) * (size_t)(round4(WAVE * pad_odd(n)) + round4(WAVE * pad_odd(3 * T)) + round4(WAVE * pad_odd(4 * T)));
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
    hipStream_t s = (hipStream_t)stream;
    DRM_DISPATCH_CAP(chains[0].capacity, {
        hipLaunchKernelGGL(fk_fanout_kernel<C>, dim3((unsigned)tiles), dim3(WAVE * T), lds, s, tab, T, n, q, B, pos, quat,
                           div_magic(n), div_magic(3 * T), div_magic(4 * T), align);
    })
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
        const int64_t done = launch_fk_arm(w, q, B, pos, quat, (hipStream_t)stream);
        if (done > 0) {
            rc = launched();
            if (rc || done == B) return rc;
            drm_walk generic = *w;
            generic.shape &= ~DRM_WALK_ARM_CHAIN;
            return drm_fk(&generic, q + done * n, B - done, 1, pos + done * 3, quat + done * 4, stream);
        }
    }
    Geometry g;
    rc = make_geometry(B, w->capacity * DRM_OPF_STRIDE + round4(WAVE * pad_odd(n)) + round4(WAVE * pad_odd(3 * T)) +
                              round4(WAVE * pad_odd(4 * T)) + w->n_slots * 12 * WAVE, g);
    if (rc) return rc;
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT) | al16(w->ops_f, AL_TABLE);
    hipStream_t s = (hipStream_t)stream;
    DRM_DISPATCH_CAP(w->capacity, {
        rc = ensure_lds(fk_kernel<C>, g.lds_bytes);
        if (rc) return rc;
        hipLaunchKernelGGL(fk_kernel<C>, g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, n, q, B, T, pos, quat,
                           div_magic(n), div_magic(3 * T), div_magic(4 * T), g.lds_per_wave, align);
    })
    return launched();
}
