// drm_arm_static.hpp — the bodies of a serial arm's OWN dynamics kernels (per-robot translation units written by specialize.py:
// the walk table a constexpr array, so the robot's constants fold into the instruction stream — products with exact zeros and
// ones disappear, no table is staged in or read from LDS).  Inverse dynamics and the fused FK + RNEA launch take the streaming
// two-samples-per-lane walk of csrc/drm_arm_stream.hpp; the three kernels here are the one-sample-per-lane chain walks of
// csrc/drm_sample.hpp behind the same kind of wrapper the library's arm kernels give them (csrc/drm_crba.hip crba_arm_kernel,
// csrc/drm_forward_dynamics.hip forward_dynamics_arm_kernel, csrc/drm_rnea_backward.hip rnea_backward_arm_kernel), one
// wavefront per block and 64-row tile, every lane its own rows straight into registers:
//   crba_arm_static_body            H [B, n, n] from q                            (reference robot_model.py:402-450)
//   forward_dynamics_arm_static_body qdd from q, qd, f: bias torques, H, L^T D L solve in registers   (robot_model.py:487-624)
//   rnea_backward_arm_static_body   dL/dq, dL/dqd, dL/dqdd from dL/dtau — INPUT gradients of a constant model
//                                   (what autograd computes through robot_model.py:305-375)
#pragma once
#include "drm_arm_stream.hpp"

namespace drm {

template <int NJ>
__device__ __forceinline__ void lane_row(const float *__restrict__ src, int64_t row, float (&v)[NJ]) {
#pragma unroll
    for (int d = 0; d < NJ; ++d) v[d] = src[row * NJ + d];
}

template <int NJ, int LINKS, class ROW>
__device__ __forceinline__ void crba_arm_static_body(ROW row, const float *__restrict__ q, int n_tiles, float *__restrict__ H) {
    static_assert((NJ * NJ) & 1, "odd row width (linear LDS image)");
    constexpr int NN = NJ * NJ;
    __shared__ __attribute__((aligned(16))) float lh[round4(WAVE * NN)];
    const int tile = (int)blockIdx.x;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x;
    const int64_t b0 = (int64_t)tile * WAVE;
    float qv[NJ];
    lane_row<NJ>(q, b0 + lane, qv);
    float *hrow = lh + lane * NN;
    crba_chain<LINKS, NJ>(row, qv, [&](int i, int j, float v) { hrow[i * NJ + j] = v; });
    wave_lds_sync();
    tile_store<NN>(H + b0 * NN, WAVE, NN, 0u, lh, lane, true);
}

template <int NJ, int LINKS, class ROW>
__device__ __forceinline__ void forward_dynamics_arm_static_body(ROW row, const float *__restrict__ q, const float *__restrict__ qd,
                                                                 const float *__restrict__ f, int n_tiles, int flags,
                                                                 float *__restrict__ qdd) {
    static_assert(NJ & 1, "odd row width (linear LDS image)");
    __shared__ __attribute__((aligned(16))) float lq[round4(WAVE * NJ)];
    const int tile = (int)blockIdx.x;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x;
    const int64_t b0 = (int64_t)tile * WAVE;
    float qv[NJ], qdv[NJ], rhs[NJ], zero[NJ], nle[NJ];
    lane_row<NJ>(q, b0 + lane, qv);
    lane_row<NJ>(qd, b0 + lane, qdv);
    lane_row<NJ>(f, b0 + lane, rhs);
#pragma unroll
    for (int d = 0; d < NJ; ++d) zero[d] = 0.0f;
    float cs[NJ], sn[NJ];
    chain_trig<NJ>(qv, cs, sn);
    // bias torques (RNEA with qdd = 0; every body force in registers: KEEP = LINKS), then H on the same cos / sin, then the solve
    rnea_chain_trig<LINKS, NJ, LINKS, false>(row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, cs, sn, qdv, zero, nle,
                                            [](int, const Force &) {}, [](int, Force &) {});
    float Ht[NJ * (NJ + 1) / 2];
    crba_chain_trig<LINKS, NJ>(row, cs, sn, [&](int i, int j, float v) {
        if (i >= j) Ht[tri_index(i, j)] = v;
    });
#pragma unroll
    for (int d = 0; d < NJ; ++d) rhs[d] -= nle[d];
    ltdl_solve_unrolled<NJ>(Ht, rhs);
#pragma unroll
    for (int d = 0; d < NJ; ++d) lq[lane * NJ + d] = rhs[d];
    wave_lds_sync();
    tile_store<NJ>(qdd + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
}

// forward dynamics, TWO SAMPLES PER LANE (round 6): one wavefront per 128-row pair of tiles, rows l and l + 64 in the two halves of
// every register pair — bias torques (rnea_chain2_trig, qdd = 0), H (crba_chain2_trig), L^T D L solve, all on pairs
template <int NJ, int LINKS, class ROW>
__device__ __forceinline__ void forward_dynamics_arm2_static_body(ROW row, const float *__restrict__ q, const float *__restrict__ qd,
                                                                  const float *__restrict__ f, int n_pairs, int flags,
                                                                  float *__restrict__ qdd) {
    static_assert(NJ & 1, "odd row width (linear LDS image)");
    __shared__ __attribute__((aligned(16))) float lq[round4(STREAM_TILE * NJ)];
    const int pair = (int)blockIdx.x;
    if (pair >= n_pairs) return;
    const unsigned lane = threadIdx.x;
    const int64_t b0 = (int64_t)pair * STREAM_TILE;
    f2 qv[NJ], qdv[NJ], rhs[NJ], zero[NJ], nle[NJ];
    {
        float a[NJ], b[NJ];
        lane_row<NJ>(q, b0 + lane, a); lane_row<NJ>(q, b0 + WAVE + lane, b);
#pragma unroll
        for (int d = 0; d < NJ; ++d) qv[d] = f2_make(a[d], b[d]);
        lane_row<NJ>(qd, b0 + lane, a); lane_row<NJ>(qd, b0 + WAVE + lane, b);
#pragma unroll
        for (int d = 0; d < NJ; ++d) qdv[d] = f2_make(a[d], b[d]);
        lane_row<NJ>(f, b0 + lane, a); lane_row<NJ>(f, b0 + WAVE + lane, b);
#pragma unroll
        for (int d = 0; d < NJ; ++d) { rhs[d] = f2_make(a[d], b[d]); zero[d] = f2_bcast(0.0f); }
    }
    f2 cs[NJ], sn[NJ];
    chain_trig2<NJ>(qv, cs, sn);
    rnea_chain2_trig<LINKS, NJ, LINKS - 1, false>(row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, cs, sn, qdv, zero, nle,
                                                 [](int, const Force2 &) {}, [](int, Force2 &) {});
    f2 Ht[NJ * (NJ + 1) / 2];
    crba_chain2_trig<LINKS, NJ>(row, cs, sn, [&](int i, int j, f2 v) { Ht[tri_index(i, j)] = v; });
#pragma unroll
    for (int d = 0; d < NJ; ++d) rhs[d] -= nle[d];
    ltdl_solve_unrolled<NJ, f2>(Ht, rhs);
#pragma unroll
    for (int d = 0; d < NJ; ++d) {
        lq[lane * NJ + d] = rhs[d][0];
        lq[(WAVE + lane) * NJ + d] = rhs[d][1];
    }
    wave_lds_sync();
    tile_store<2 * NJ>(qdd + b0 * NJ, WAVE, 2 * NJ, 0u, lq, lane, true);
}

template <int NJ, int LINKS, class ROW>
__device__ __forceinline__ void rnea_backward_arm_static_body(ROW row, const float *__restrict__ q, const float *__restrict__ qd,
                                                              const float *__restrict__ qdd, const float *__restrict__ gtau,
                                                              int n_tiles, int flags, float *__restrict__ gq, float *__restrict__ gqd,
                                                              float *__restrict__ gqdd) {
    static_assert(NJ & 1, "odd row width (linear LDS image)");
    constexpr int Q_FLOATS = round4(WAVE * NJ);
    __shared__ __attribute__((aligned(16))) float lg[3 * Q_FLOATS];
    const int tile = (int)blockIdx.x;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x;
    const int64_t b0 = (int64_t)tile * WAVE;
    float *lq = lg, *lqd = lq + Q_FLOATS, *lqdd = lqd + Q_FLOATS;
    float qv[NJ], qdv[NJ], qddv[NJ], gtv[NJ];
    lane_row<NJ>(q, b0 + lane, qv);
    lane_row<NJ>(qd, b0 + lane, qdv);
    if (qdd) lane_row<NJ>(qdd, b0 + lane, qddv);
    else {
#pragma unroll
        for (int d = 0; d < NJ; ++d) qddv[d] = 0.0f;
    }
    lane_row<NJ>(gtau, b0 + lane, gtv);
    rnea_backward_chain<LINKS, NJ>(row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, (uint64_t)0, true, qv, qdv, qddv, gtv,
                                   [&](int d, float a, float v, float c) {
                                       lq[lane * NJ + d] = a; lqd[lane * NJ + d] = v; lqdd[lane * NJ + d] = c;
                                   },
                                   [](int, const float *) {});
    wave_lds_sync();
    tile_store<NJ>(gq + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
    tile_store<NJ>(gqd + b0 * NJ, WAVE, NJ, 0u, lqd, lane, true);
    tile_store<NJ>(gqdd + b0 * NJ, WAVE, NJ, 0u, lqdd, lane, true);
}


// input gradients of inverse dynamics, TWO SAMPLES PER LANE (round 6): one wavefront per 128-row pair of tiles
template <int NJ, int LINKS, class ROW>
__device__ __forceinline__ void rnea_backward_arm2_static_body(ROW row, const float *__restrict__ q, const float *__restrict__ qd,
                                                               const float *__restrict__ qdd, const float *__restrict__ gtau,
                                                               int n_pairs, int flags, float *__restrict__ gq, float *__restrict__ gqd,
                                                               float *__restrict__ gqdd) {
    static_assert(NJ & 1, "odd row width (linear LDS image)");
    constexpr int Q_FLOATS = round4(STREAM_TILE * NJ);
    __shared__ __attribute__((aligned(16))) float lg[3 * Q_FLOATS];
    const int pair = (int)blockIdx.x;
    if (pair >= n_pairs) return;
    const unsigned lane = threadIdx.x;
    const int64_t b0 = (int64_t)pair * STREAM_TILE;
    float *lq = lg, *lqd = lq + Q_FLOATS, *lqdd = lqd + Q_FLOATS;
    f2 qv[NJ], qdv[NJ], qddv[NJ], gtv[NJ];
    auto rows2 = [&](const float *src, f2 (&v)[NJ]) {
        float a[NJ], b[NJ];
        lane_row<NJ>(src, b0 + lane, a);
        lane_row<NJ>(src, b0 + WAVE + lane, b);
#pragma unroll
        for (int d = 0; d < NJ; ++d) v[d] = f2_make(a[d], b[d]);
    };
    rows2(q, qv);
    rows2(qd, qdv);
    if (qdd) rows2(qdd, qddv);
    else {
#pragma unroll
        for (int d = 0; d < NJ; ++d) qddv[d] = f2_bcast(0.0f);
    }
    rows2(gtau, gtv);
    rnea_backward_chain2<LINKS, NJ>(row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, qv, qdv, qddv, gtv,
                                    [&](int d, f2 a, f2 v, f2 c) {
                                        lq[lane * NJ + d] = a[0]; lq[(WAVE + lane) * NJ + d] = a[1];
                                        lqd[lane * NJ + d] = v[0]; lqd[(WAVE + lane) * NJ + d] = v[1];
                                        lqdd[lane * NJ + d] = c[0]; lqdd[(WAVE + lane) * NJ + d] = c[1];
                                    });
    wave_lds_sync();
    tile_store<2 * NJ>(gq + b0 * NJ, WAVE, 2 * NJ, 0u, lq, lane, true);
    tile_store<2 * NJ>(gqd + b0 * NJ, WAVE, 2 * NJ, 0u, lqd, lane, true);
    tile_store<2 * NJ>(gqdd + b0 * NJ, WAVE, 2 * NJ, 0u, lqdd, lane, true);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Reverse-mode inverse dynamics of a serial arm WITH learnable link parameters (round 6; the learn-dynamics workload, reference
// robot_model.py:305-375 under autograd with rigid_body_params.py parametrisations, examples/learn_dynamics_iiwa.py): the gradient
// of a loss on the torques with respect to the learnable entries of the walk table (+ the input gradients, when asked for).
// A model's table changes every optimiser step — but only in the blocks a learnable parameter feeds: KIN = F / t of a link whose
// `trans` / `rot_angles` are learnable, DYN = mass / mcom / I_o / damping of a link whose `mass` / `com` / `inertia_mat` /
// `joint_damping` are.  Every other block is a constant of the robot and is folded into the instruction stream here (MIXED::ft(k) /
// MIXED::operator()(k) return the constexpr table or this launch's LDS copy, per op and block, decided at compile time); the adjoint
// code of constant blocks is dead and gone.  The sums over the batch of the live gradient entries are kept PER LANE in registers
// across the tiles of a (persistent) wavefront and reduced across the wavefront once at the end (the library kernel does that for
// ONE learnable link and pays 26 DPP reductions per tile and link otherwise); the per-wavefront rows of partial sums have the
// library's layout ([waves, CAP * 32], reduced by rnea_backward_reduce_kernel in a fixed order).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int LINKS, uint32_t MASK_KIN, uint32_t MASK_DYN>
struct ParamSlots {      // where op k's live entries sit in the per-lane sums: KIN entries [0, 12) first, then DYN entries [12, 26)
    static constexpr int KIN = DRM_OPF_FT_FLOATS, DYN = DRM_OPF_DAMP + 1 - DRM_OPF_FT_FLOATS;
    static constexpr int base(int k) {
        int b = 0;
        for (int i = 0; i < k; ++i) b += (((MASK_KIN >> i) & 1u) ? KIN : 0) + (((MASK_DYN >> i) & 1u) ? DYN : 0);
        return b;
    }
    static constexpr int COUNT = base(LINKS);
};

template <int NJ, int LINKS, uint32_t MASK_KIN, uint32_t MASK_DYN, class MIXED>
__device__ __forceinline__ void rnea_backward_arm_param_static_body(const float *__restrict__ ops_f, const float *__restrict__ q,
                                                                    const float *__restrict__ qd, const float *__restrict__ qdd,
                                                                    const float *__restrict__ gtau, int n_tiles, int flags,
                                                                    float *__restrict__ gq, float *__restrict__ gqd,
                                                                    float *__restrict__ gqdd, float *__restrict__ partials) {
    static_assert(NJ & 1, "odd row width (linear LDS image)");
    constexpr int CAP = 8, C_FLOATS = CAP * DRM_OPF_STRIDE, Q_FLOATS = round4(WAVE * NJ), PER_WAVE = C_FLOATS + 3 * Q_FLOATS;
    constexpr int NV = CAP * DRM_OPF_STRIDE;
    using Slots = ParamSlots<LINKS, MASK_KIN, MASK_DYN>;
    constexpr int NSUM = Slots::COUNT;
    static_assert(NSUM > 0 && NSUM <= 8 * 26, "at least one learnable block");
    __shared__ __attribute__((aligned(16))) float smem[MAX_WAVES_PER_BLOCK * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int wave_id = (int)blockIdx.x * MAX_WAVES_PER_BLOCK + wave, n_waves = (int)gridDim.x * MAX_WAVES_PER_BLOCK;
    float *lc = smem + wave * PER_WAVE;
    float *lq = lc + C_FLOATS, *lqd = lq + Q_FLOATS, *lqdd = lqd + Q_FLOATS;
    {
        float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];      // this launch's table: only the live blocks are read back
        pin(cv);
        reinterpret_cast<float4 *>(lc)[lane] = cv;
    }
    wave_lds_sync();
    float psum[NSUM];
#pragma unroll
    for (int j = 0; j < NSUM; ++j) psum[j] = 0.0f;
    const MIXED row{lc};
    constexpr uint64_t MASK = (uint64_t)(MASK_KIN | MASK_DYN);

    for (int tile = wave_id; tile < n_tiles; tile += n_waves) {
        const int64_t b0 = (int64_t)tile * WAVE;
        float qv[NJ], qdv[NJ], qddv[NJ], gtv[NJ];
        lane_row<NJ>(q, b0 + lane, qv);
        lane_row<NJ>(qd, b0 + lane, qdv);
        if (qdd) lane_row<NJ>(qdd, b0 + lane, qddv);
        else {
#pragma unroll
            for (int d = 0; d < NJ; ++d) qddv[d] = 0.0f;
        }
        lane_row<NJ>(gtau, b0 + lane, gtv);
        wave_lds_sync(); // the previous tile's staged gradients have left the LDS tiles
        rnea_backward_chain<LINKS, NJ>(
            row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, MASK, gq != nullptr, qv, qdv, qddv, gtv,
            [&](int d, float a, float v, float c) {
                lq[lane * NJ + d] = a; lqd[lane * NJ + d] = v; lqdd[lane * NJ + d] = c;
            },
            [&](int k, const float *g) {      // (k is a constant of the unrolled walk: every index below folds)
                int at = Slots::base(k);
                if ((MASK_KIN >> k) & 1u) {
#pragma unroll
                    for (int j = 0; j < Slots::KIN; ++j) psum[at + j] += g[j];
                    at += Slots::KIN;
                }
                if ((MASK_DYN >> k) & 1u) {
#pragma unroll
                    for (int j = 0; j < Slots::DYN; ++j) psum[at + j] += g[DRM_OPF_FT_FLOATS + j];
                }
            });
        if (gq) {
            wave_lds_sync();
            tile_store<NJ>(gq + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
            tile_store<NJ>(gqd + b0 * NJ, WAVE, NJ, 0u, lqd, lane, true);
            tile_store<NJ>(gqdd + b0 * NJ, WAVE, NJ, 0u, lqdd, lane, true);
        }
    }
    // this wavefront's row of partial sums: zeros, the live entries summed over its 64 lanes — assembled in LDS (over the table,
    // which is dead now), written as one coalesced row
    wave_lds_sync();
    reinterpret_cast<float4 *>(lc)[lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < LINKS; ++k) {
        int at = Slots::base(k);
        if ((MASK_KIN >> k) & 1u) {
            wave_sums_lane63<Slots::KIN>(lane, [&](int j) { return psum[at + j]; }, [&](int j, float total) { lc[k * DRM_OPF_STRIDE + j] = total; });
            at += Slots::KIN;
        }
        if ((MASK_DYN >> k) & 1u) {
            wave_sums_lane63<Slots::DYN>(lane, [&](int j) { return psum[at + j]; },
                                         [&](int j, float total) { lc[k * DRM_OPF_STRIDE + DRM_OPF_FT_FLOATS + j] = total; });
        }
    }
    wave_lds_sync();
    reinterpret_cast<float4 *>(partials + (int64_t)wave_id * NV)[lane] = reinterpret_cast<const float4 *>(lc)[lane];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Forward kinematics of 2 .. 4 disjoint serial chains (the fingertips of a hand, BASELINE configuration 4) with link-major outputs
// (drm_fk_fanout_links: pos [T, B, 3], quat [T, B, 4]), one wavefront per chain and 64-row tile — the layout of
// csrc/drm_chain_kernels.hip fk_fan_chain_kernel<.., LINKS = true> — with every chain's constants folded into the instruction stream.
// The chain is walked in SCALAR form (joint_rot_z / compose of csrc/drm_sample.hpp): the library's pair-packed form multiplies PAIRS
// of constants, of which a zero half cannot be dropped (the metric kernel folded a tenth of its instructions that way), while a
// hand's chain tables are almost pure axis permutations (Allegro: 50 of a finger's 60 F / t entries are 0 or +-1) and a scalar
// product with a constant 0 or 1 is no instruction at all.
//   C (generated per chain): USED ops, DOF[k] = the DoF column op k reads or -1 (fixed), MOVING = how many move, Q4 = the moving ops
//   read four consecutive columns starting at a multiple of four of rows of n % 4 == 0 floats (ONE 16-byte load), PERM = the target
//   frame's un-permutation code, row(k) -> op k's constant row
// ---------------------------------------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void fk_fan_links_static_wave(const float *__restrict__ q, float *__restrict__ pos, float *__restrict__ quat,
                                                         int64_t B, int chain, float *st) {
    constexpr int M = C::MOVING, N = C::NDOFS;
    const unsigned lane = threadIdx.x & 63u;
    const int64_t b0 = (int64_t)blockIdx.x * WAVE;
    float qv[M];
    if constexpr (C::Q4) {
        const float4 v = *reinterpret_cast<const float4 *>(q + (b0 + lane) * N + C::FIRST);
        qv[0] = v.x; qv[1] = v.y; qv[2] = v.z; qv[3] = v.w;
    } else {
        int j = 0;
#pragma unroll
        for (int k = 0; k < C::USED; ++k)
            if (C::dof(k) >= 0) qv[j++] = q[(b0 + lane) * N + C::dof(k)];
    }
    float cs[M], sn[M];
    chain_trig<M>(qv, cs, sn);
    Pose ee;
    int j = 0;
#pragma unroll
    for (int k = 0; k < C::USED; ++k) {
        const OpFT o = load_ft(C::row(k));
        float J[9];
        if (C::dof(k) >= 0) {
            joint_rot_z(o.F, cs[j], sn[j], J);
            ++j;
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) J[i] = o.F[i];
        }
        if (k == 0) { // the parent is the identity root: R = J, p = t exactly
#pragma unroll
            for (int i = 0; i < 9; ++i) ee.R[i] = J[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) ee.p[i] = o.t[i];
        } else {
            compose(ee, J, o.t, ee);
        }
    }
    st[lane * 3 + 0] = ee.p[0]; st[lane * 3 + 1] = ee.p[1]; st[lane * 3 + 2] = ee.p[2];
    wave_lds_sync();
    if (lane < 48u) store16_wt(reinterpret_cast<float4 *>(pos + ((int64_t)chain * B + b0) * 3) + lane, reinterpret_cast<const float4 *>(st)[lane]);
    float qt[4];
    unpermute(C::PERM, ee.R);
    quat_xyzw(ee.R, qt);
    store16_wt(reinterpret_cast<float4 *>(quat + ((int64_t)chain * B + b0) * 4) + lane, make_float4(qt[0], qt[1], qt[2], qt[3]));
}

} // namespace drm
