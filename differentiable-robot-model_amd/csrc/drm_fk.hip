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

} // namespace drm

using namespace drm;

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
