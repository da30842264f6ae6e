// drm_forward_dynamics.hip — K8: joint accelerations from joint torques, qdd = H(q)^-1 (f - nle(q, qd)).
//
// Replaces DifferentiableRobotModel.compute_forward_dynamics (robot_model.py:487-624, Featherstone's
// articulated-body algorithm written as three Python loops over the links with 6x6 bmm's per link).  The
// articulated-body recursion is an O(n) elimination of the linear system H qdd = f - nle; this kernel forms the same
// system with the two walks it already has — crba_walk for H, rnea_walk with qdd = 0 for the bias torques nle — and
// solves it per sample by Cholesky in LDS (n <= ~20: ~n^3/3 FMAs, less than one of the walks).  Same result up to
// fp32 rounding amplified by cond(H), like the reference's own recursion (tolerances in tests/).
//
// Per sample: in q, qd, f [n] (12 n bytes), out qdd [n] (4 n bytes).          n = 7: 112 B
// LDS per wave: [ q qd f : 3 x 64 (n|1) ][ lower triangle of H : 64 (n(n+1)/2 | 1) ]
//               [ slots : n_slots * max(10 + 6 depth, 18) * 64, the two walks use them one after the other ]
#include "drm_common.hpp"
#include "drm_sample.hpp"

namespace drm {

template <int CAP>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    forward_dynamics_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int n, int n_slots,
                            int sdepth, int flags, int zero_fill, const float *__restrict__ q, const float *__restrict__ qd,
                            const float *__restrict__ f, int64_t B, float *__restrict__ qdd, uint32_t magic_q,
                            int lds_per_wave, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx cx;
    if (!wave_begin(B, lds_per_wave, smem, cx)) return;
    const unsigned lane = cx.lane;
    const int nn = n * (n + 1) / 2; // packed lower triangle
    const int Sq = pad_odd(n), Sh = pad_odd(nn), region = round4(WAVE * Sq);
    float *lq = cx.lds, *lqd = lq + region, *lf = lqd + region;
    float *lh = lf + region;
    float *lis = lh + round4(WAVE * Sh);             // crba inertia slots [slot][10][64]
    float *lss = lis + n_slots * (10 * WAVE);        // crba axis slots    [slot][op < sdepth][6][64]
    float *lms = lis;                                // rnea motion slots  [slot][12][64]   (same memory, later)
    float *lfs = lms + n_slots * (12 * WAVE);        // rnea force slots   [slot][6][64]
    const bool fast = cx.full && (n & 1);

    tile_load<0>(q + cx.b0 * n, cx.rows, n, magic_q, lq, lane, fast && (align & AL_Q), cx.full && (align & AL_Q));
    tile_load<0>(qd + cx.b0 * n, cx.rows, n, magic_q, lqd, lane, fast && (align & AL_QD), cx.full && (align & AL_QD));
    tile_load<0>(f + cx.b0 * n, cx.rows, n, magic_q, lf, lane, fast && (align & AL_QDD), cx.full && (align & AL_QDD));
    for (int s = 0; s < n_slots * 10; ++s) lis[s * WAVE + lane] = 0.0f;
    float *hrow = lh + lane * Sh;
    if (zero_fill)
        for (int i = 0; i < nn; ++i) hrow[i] = 0.0f; // pairs of joints on different branches
    wave_lds_sync();

    // lanes past a partial tile read zeros (see drm_fk.hip); their H is then a valid inertia matrix as well
    const bool live = (int)lane < cx.rows;
    const unsigned row = lane * Sq;
    auto qf1 = [&](int d) -> float { return live ? lq[row + d] : 0.0f; };
    {
        auto islot_add = [&](int s, const Inertia &a) {
            float *b = lis + s * (10 * WAVE) + lane;
            b[0] += a.m;
#pragma unroll
            for (int i = 0; i < 3; ++i) b[(1 + i) * WAVE] += a.h[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) b[(4 + i) * WAVE] += a.I[i];
        };
        auto islot_take = [&](int s, Inertia &a) {
            float *b = lis + s * (10 * WAVE) + lane;
            a.m += b[0]; b[0] = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i) { a.h[i] += b[(1 + i) * WAVE]; b[(1 + i) * WAVE] = 0.0f; }
#pragma unroll
            for (int i = 0; i < 6; ++i) { a.I[i] += b[(4 + i) * WAVE]; b[(4 + i) * WAVE] = 0.0f; }
        };
        auto sslot_save = [&](int s, int j, const Axis &a) {
            float *b = lss + (s * sdepth + j) * (6 * WAVE) + lane;
#pragma unroll
            for (int i = 0; i < 3; ++i) { b[i * WAVE] = a.ang[i]; b[(3 + i) * WAVE] = a.lin[i]; }
        };
        auto sslot_load = [&](int s, int j, Axis &a) {
            const float *b = lss + (s * sdepth + j) * (6 * WAVE) + lane;
#pragma unroll
            for (int i = 0; i < 3; ++i) { a.ang[i] = b[i * WAVE]; a.lin[i] = b[(3 + i) * WAVE]; }
        };
        auto hout = [&](int di, int dj, float v) {
            if (di >= dj) hrow[tri_index(di, dj)] = v;
        };
        crba_walk<CAP>(ops_f, ops_i, qf1, islot_add, islot_take, sslot_save, sslot_load, hout);
    }
    wave_lds_sync(); // the composite-inertia walk is done with the slot memory
    for (int s = 0; s < n_slots * 6; ++s) lfs[s * WAVE + lane] = 0.0f;
    wave_lds_sync();
    {
        // bias torques: RNEA with zero joint accelerations (robot_model.py:377-400); rhs = f - nle, over f
        auto qf3 = [&](int d, float &a, float &v, float &acc) {
            a = live ? lq[row + d] : 0.0f;
            v = lqd[row + d];
            acc = 0.0f;
        };
        auto tau_out = [&](int d, float v) { lf[row + d] -= v; };
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
        rnea_walk<CAP>(ops_f, ops_i, flags, qf3, tau_out, motion_save, motion_load, force_add, force_take);
    }
    cholesky_solve(n, hrow, lf + row);
    wave_lds_sync();
    tile_store<0>(qdd + cx.b0 * n, cx.rows, n, magic_q, lf, lane, fast && (align & AL_TAU), cx.full && (align & AL_TAU));
}

// Serial-chain ("arm") specialisation, full tiles only: the chain forms of the two walks (drm_sample.hpp crba_chain,
// rnea_chain) with H's lower triangle and the right-hand side in REGISTERS and a fully unrolled Cholesky; constants
// staged once per wave in LDS, RNEA's body forces parked over the dead input tiles, qdd staged over them at the end.
template <int CAP, int NJ>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    forward_dynamics_arm_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                                const float *__restrict__ f, int n_tiles, int flags, float *__restrict__ qdd) {
    static_assert(NJ & 1, "odd row widths only (linear LDS image)");
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, Q_FLOATS = round4(WAVE * NJ), F_FLOATS = CAP * 6 * WAVE;
    static_assert(3 * Q_FLOATS <= F_FLOATS, "the input tiles fit under the parking area");
    constexpr int PER_WAVE = C_FLOATS + F_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[MAX_WAVES_PER_BLOCK * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = (int)blockIdx.x * MAX_WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * PER_WAVE;
    float *lq = lc + C_FLOATS, *lqd = lq + Q_FLOATS, *lf = lqd + Q_FLOATS;
    float *park = lq + lane; // body forces between RNEA's sweeps: [link][6][64], over the (by then dead) input tiles
    const int64_t b0 = (int64_t)tile * WAVE;

    float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];
    tile_load<NJ>(q + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
    tile_load<NJ>(qd + b0 * NJ, WAVE, NJ, 0u, lqd, lane, true);
    tile_load<NJ>(f + b0 * NJ, WAVE, NJ, 0u, lf, lane, true);
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    wave_lds_sync();
    float qv[NJ], qdv[NJ], rhs[NJ], zero[NJ], nle[NJ];
#pragma unroll
    for (int d = 0; d < NJ; ++d) {
        qv[d] = lq[lane * NJ + d];
        qdv[d] = lqd[lane * NJ + d];
        rhs[d] = lf[lane * NJ + d];
        zero[d] = 0.0f;
    }
    wave_lds_sync(); // all rows are in registers: the tiles may be overwritten
    auto row = [&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; };
    float Ht[NJ * (NJ + 1) / 2];
    crba_chain<CAP, NJ>(row, qv, [&](int i, int j, float v) {
        if (i >= j) Ht[tri_index(i, j)] = v;
    });
    rnea_chain<CAP, NJ>(row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, qv, qdv, zero, nle,
                        [&](int k, const Force &F) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                park[(k * 6 + i) * WAVE] = F.la[i][0];
                                park[(k * 6 + 3 + i) * WAVE] = F.la[i][1];
                            }
                        },
                        [&](int k, Force &F) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) F.la[i] = f2_make(park[(k * 6 + i) * WAVE], park[(k * 6 + 3 + i) * WAVE]);
                        });
#pragma unroll
    for (int d = 0; d < NJ; ++d) rhs[d] -= nle[d];
    cholesky_solve_unrolled<NJ>(Ht, rhs);
    wave_lds_sync(); // every lane is done with the parking area before qdd is staged over it
#pragma unroll
    for (int d = 0; d < NJ; ++d) lq[lane * NJ + d] = rhs[d];
    wave_lds_sync();
    tile_store<NJ>(qdd + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
}

} // namespace drm

using namespace drm;

extern "C" int drm_forward_dynamics(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B,
                                    int32_t flags, float *qdd, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !qd || !f || !qdd) return fail(DRM_ERR_INVALID, "q / qd / f / qdd must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs, nn = n * (n + 1) / 2;
#ifndef DRM_NO_ARM_KERNEL
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7 && B >= WAVE && B / WAVE < 0x7fffffffLL &&
        (((uintptr_t)q | (uintptr_t)qd | (uintptr_t)f | (uintptr_t)qdd | (uintptr_t)w->ops_f) & 15u) == 0) {
        // 7-DoF arms: full tiles through the register-resident chain kernel, ragged tail through the generic one
        const int n_tiles = (int)(B / WAVE);
        hipLaunchKernelGGL((forward_dynamics_arm_kernel<8, 7>),
                           dim3((unsigned)((n_tiles + MAX_WAVES_PER_BLOCK - 1) / MAX_WAVES_PER_BLOCK)),
                           dim3(WAVE * MAX_WAVES_PER_BLOCK), 0, (hipStream_t)stream, w->ops_f, q, qd, f, n_tiles, (int)flags,
                           qdd);
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done == B) return launched();
        rc = launched();
        if (rc) return rc;
        drm_walk generic = *w;
        generic.shape &= ~DRM_WALK_ARM_CHAIN;
        return drm_forward_dynamics(&generic, q + done * n, qd + done * n, f + done * n, B - done, flags, qdd + done * n,
                                    stream);
    }
#endif
    const int sdepth = DRM_WALK_BRANCH_DEPTH(w->shape);
    if (w->n_slots > 0 && sdepth == 0) return fail(DRM_ERR_INVALID, "walk has save slots but no branch depth in shape");
    Geometry g;
    rc = make_geometry(B, 3 * round4(WAVE * pad_odd(n)) + round4(WAVE * pad_odd(nn)) +
                              w->n_slots * ((10 + 6 * sdepth) > 18 ? (10 + 6 * sdepth) : 18) * WAVE, g);
    if (rc) return fail(DRM_ERR_UNSUPPORTED, "forward dynamics keeps the %s%ld x %ld inertia matrix (lower triangle) of 64 samples in LDS; "
                                             "this robot does not fit", "", (long)n, (long)n);
    const int zero_fill = (w->shape & DRM_WALK_ARM_CHAIN) ? 0 : 1;
    const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(f, AL_QDD) | al16(qdd, AL_TAU);
    hipStream_t s = (hipStream_t)stream;
    DRM_DISPATCH_CAP(w->capacity, {
        rc = ensure_lds(forward_dynamics_kernel<C>, g.lds_bytes);
        if (rc) return rc;
        hipLaunchKernelGGL(forward_dynamics_kernel<C>, g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, n,
                           (int)w->n_slots, sdepth, (int)flags, zero_fill, q, qd, f, B, qdd, div_magic(n), g.lds_per_wave, align);
    })
    return launched();
}
