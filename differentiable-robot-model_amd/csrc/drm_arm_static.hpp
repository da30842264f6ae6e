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

} // namespace drm
