// drm_kernels.hip — hand-written HIP kernels (gfx950 / MI355X) + the C ABI of include/drm_hip.h.
//
// Execution model:
//   * one wavefront (64 lanes) owns a tile of 64 consecutive samples, one lane per sample;
//     waves never talk to each other (no __syncthreads), so the block size is only a
//     packing choice;
//   * the API hands over row-major [B, n] / [B, 3, n] tensors (one ROW per sample), the
//     lanes want one COLUMN element per lane.  Every tensor therefore crosses HBM exactly
//     once with fully coalesced 16-byte accesses and is transposed through a wave-private
//     LDS tile whose row stride is forced odd (S | 1), which makes both the row-per-lane
//     accesses and the linear copy bank-conflict free;
//   * per-link constants are wave-uniform: they are read with scalar loads straight from
//     the walk tables (ops_f / ops_i) and live in SGPRs;
//   * the walk is unrolled against a compile-time capacity (8/16/32/64 ops) so per-op state
//     (joint axes, origins, body forces) stays in VGPRs with static indices.
//
// Arithmetic: csrc/drm_sample.hpp.  Reference behaviour replaced: robot_model.py:139-195,
// 223-248, 250-375, 626-667 (+ rigid_body.py, spatial_vector_algebra.py, utils.py).
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/drm_hip.h"
#include "drm_sample.hpp"

namespace drm {

constexpr int WAVE = 64;
constexpr int MAX_WAVES_PER_BLOCK = 4;
constexpr int MAX_LDS_BYTES = 160 * 1024;

// ordering point between LDS writes of some lanes and LDS reads of other lanes of the SAME
// wave: the LDS executes a wave's instructions in order, so only the compiler must be kept
// from reordering.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A [rows, S] row-major float tile (rows <= 64) in HBM <-> wave-private LDS with row stride
// Sp = S | 1.  Word w of the tile lives at LDS word  w + (S even ? w / S : 0).
struct TileShape {
    int S;          // floats per row (per sample)
    int Sp;         // LDS row stride
    uint32_t magic; // ceil(2^32 / S), for w / S when S is even
};

__host__ __device__ inline int pad_odd(int S) { return S | 1; }
__host__ __device__ inline int round4(int x) { return (x + 3) & ~3; }

__device__ __forceinline__ int lds_word(int w, const TileShape &ts) {
    return (ts.Sp == ts.S) ? w : w + (int)__umulhi((uint32_t)w, ts.magic);
}

// HBM -> LDS.  `vec` = tile is full (64 rows), S odd (linear image) and g is 16-byte aligned.
__device__ __forceinline__ void tile_load(const float *__restrict__ g, int rows, const TileShape ts, float *lds,
                                          int lane, bool vec) {
    if (vec) {
        const int nvec = 16 * ts.S; // float4s in a full tile
        const float4 *g4 = reinterpret_cast<const float4 *>(g);
        float4 *l4 = reinterpret_cast<float4 *>(lds);
#pragma unroll 2
        for (int i = lane; i < nvec; i += WAVE) l4[i] = g4[i];
    } else {
        const int total = rows * ts.S;
#pragma unroll 4
        for (int w = lane; w < total; w += WAVE) lds[lds_word(w, ts)] = g[w];
    }
}

// LDS -> HBM, same conventions.
__device__ __forceinline__ void tile_store(float *__restrict__ g, int rows, const TileShape ts, const float *lds,
                                           int lane, bool vec) {
    if (vec) {
        const int nvec = 16 * ts.S;
        float4 *g4 = reinterpret_cast<float4 *>(g);
        const float4 *l4 = reinterpret_cast<const float4 *>(lds);
#pragma unroll 2
        for (int i = lane; i < nvec; i += WAVE) g4[i] = l4[i];
    } else {
        const int total = rows * ts.S;
#pragma unroll 4
        for (int w = lane; w < total; w += WAVE) g[w] = lds[lds_word(w, ts)];
    }
}

struct WaveCtx {
    int lane;
    int rows;      // valid samples of this tile
    int64_t b0;    // first sample of this tile
    float *lds;    // wave-private LDS
    bool full;     // rows == 64
};

__device__ __forceinline__ bool wave_begin(int64_t B, int lds_floats_per_wave, float *smem, WaveCtx &cx) {
    const int wave = threadIdx.x >> 6;
    cx.lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    cx.b0 = tile * WAVE;
    if (cx.b0 >= B) return false;
    const int64_t left = B - cx.b0;
    cx.rows = left < WAVE ? (int)left : WAVE;
    cx.full = cx.rows == WAVE;
    cx.lds = smem + (size_t)wave * lds_floats_per_wave;
    return true;
}

// alignment bits handed over by the launcher (bit set = pointer is 16-byte aligned)
enum : uint32_t { AL_Q = 1, AL_QD = 2, AL_QDD = 4, AL_POS = 8, AL_QUAT = 16, AL_LIN = 32, AL_ANG = 64, AL_TAU = 128 };

// ---------------------------------------------------------------------------------------
// K2: FK + geometric Jacobian of one chain (the metric kernel).
// LDS per wave: [ q tile : 64 * (n|1) ][ staging : 64 * (3n|1) ]
// ---------------------------------------------------------------------------------------
template <int CAP>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    fk_jacobian_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int n_ops, int n,
                       uint64_t dof_mask, const float *__restrict__ q, int64_t B, float *__restrict__ pos,
                       float *__restrict__ quat, float *__restrict__ lin, float *__restrict__ ang, TileShape ts_q,
                       TileShape ts_j, int lds_per_wave, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx cx;
    if (!wave_begin(B, lds_per_wave, smem, cx)) return;
    const int lane = cx.lane;
    float *lq = cx.lds;
    float *stage = cx.lds + round4(WAVE * ts_q.Sp);

    tile_load(q + cx.b0 * n, cx.rows, ts_q, lq, lane, cx.full && ts_q.Sp == ts_q.S && (align & AL_Q));
    wave_lds_sync();

    const bool live = lane < cx.rows;
    const float *qrow = lq + lane * ts_q.Sp;
    auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; };

    Pose ee;
    float z[CAP][3], pj[CAP][3];
    fk_chain<CAP>(ops_f, ops_i, n_ops, qf, ee, z, pj);

    // ---- pos [B,3]: 3 floats per lane -> LDS -> coalesced store ------------------------
    const TileShape ts3 = {3, 3, 0};
    if (pos) {
        stage[lane * 3 + 0] = ee.p[0];
        stage[lane * 3 + 1] = ee.p[1];
        stage[lane * 3 + 2] = ee.p[2];
        wave_lds_sync();
        tile_store(pos + cx.b0 * 3, cx.rows, ts3, stage, lane, cx.full && (align & AL_POS));
        wave_lds_sync();
    }
    // ---- quat [B,4]: one 16-byte store per lane is already coalesced -------------------
    if (quat && live) {
        float qt[4];
        quat_xyzw(ee.R, qt);
        float *dst = quat + (cx.b0 + lane) * 4;
        if (align & AL_QUAT) {
            *reinterpret_cast<float4 *>(dst) = make_float4(qt[0], qt[1], qt[2], qt[3]);
        } else {
            dst[0] = qt[0]; dst[1] = qt[1]; dst[2] = qt[2]; dst[3] = qt[3];
        }
    }
    // ---- Jacobians [B,3,n]: column d of op k at row offset r*n + d ----------------------
    float *jrow = stage + lane * ts_j.Sp;
    const uint64_t all = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    const bool has_off_chain = (dof_mask & all) != all;
    const bool vec_j = cx.full && ts_j.Sp == ts_j.S;

    // linear part: z_k x (p_e - p_k)   (robot_model.py:661)
    if (has_off_chain) {
        for (int d = 0; d < n; ++d)
            if (!((dof_mask >> d) & 1ull)) { jrow[d] = 0.0f; jrow[n + d] = 0.0f; jrow[2 * n + d] = 0.0f; }
    }
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        if (k < n_ops) {
            const int d = ops_i[k * DRM_OPI_STRIDE + DRM_OPI_DOF];
            if (d >= 0) {
                const float dp[3] = {ee.p[0] - pj[k][0], ee.p[1] - pj[k][1], ee.p[2] - pj[k][2]};
                float c[3];
                cross3(z[k], dp, c);
                jrow[d] = c[0];
                jrow[n + d] = c[1];
                jrow[2 * n + d] = c[2];
            }
        }
    }
    wave_lds_sync();
    tile_store(lin + cx.b0 * 3 * n, cx.rows, ts_j, stage, lane, vec_j && (align & AL_LIN));
    wave_lds_sync();

    // angular part: z_k   (robot_model.py:662); off-chain zeros are still in place
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        if (k < n_ops) {
            const int d = ops_i[k * DRM_OPI_STRIDE + DRM_OPI_DOF];
            if (d >= 0) {
                jrow[d] = z[k][0];
                jrow[n + d] = z[k][1];
                jrow[2 * n + d] = z[k][2];
            }
        }
    }
    wave_lds_sync();
    tile_store(ang + cx.b0 * 3 * n, cx.rows, ts_j, stage, lane, vec_j && (align & AL_ANG));
}

// ---------------------------------------------------------------------------------------
// K1/K4: FK of T target links over a (possibly branching) walk.
// LDS per wave: [ q : 64*(n|1) ][ pos : 64*(3T|1) ][ quat : 64*(4T+1) ]
// ---------------------------------------------------------------------------------------
template <int CAP>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    fk_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int n_ops, int n,
              const float *__restrict__ q, int64_t B, int T, float *__restrict__ pos, float *__restrict__ quat,
              TileShape ts_q, TileShape ts_p, TileShape ts_r, int lds_per_wave, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx cx;
    if (!wave_begin(B, lds_per_wave, smem, cx)) return;
    const int lane = cx.lane;
    float *lq = cx.lds;
    float *lp = lq + round4(WAVE * ts_q.Sp);
    float *lr = lp + round4(WAVE * ts_p.Sp);

    tile_load(q + cx.b0 * n, cx.rows, ts_q, lq, lane, cx.full && ts_q.Sp == ts_q.S && (align & AL_Q));
    wave_lds_sync();

    const bool live = lane < cx.rows;
    const float *qrow = lq + lane * ts_q.Sp;
    auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; };
    float *prow = lp + lane * ts_p.Sp;
    float *rrow = lr + lane * ts_r.Sp;
    auto emit = [&](int t, const Pose &P) {
        float qt[4];
        quat_xyzw(P.R, qt);
        prow[t * 3 + 0] = P.p[0]; prow[t * 3 + 1] = P.p[1]; prow[t * 3 + 2] = P.p[2];
        rrow[t * 4 + 0] = qt[0]; rrow[t * 4 + 1] = qt[1]; rrow[t * 4 + 2] = qt[2]; rrow[t * 4 + 3] = qt[3];
    };
    fk_walk<CAP>(ops_f, ops_i, n_ops, qf, emit);
    wave_lds_sync();
    tile_store(pos + cx.b0 * 3 * T, cx.rows, ts_p, lp, lane, cx.full && ts_p.Sp == ts_p.S && (align & AL_POS));
    tile_store(quat + cx.b0 * 4 * T, cx.rows, ts_r, lr, lane, false);
}

// ---------------------------------------------------------------------------------------
// K3: RNEA over the whole tree.
// LDS per wave: [ q ][ qd ][ qdd ][ tau ], each 64*(n|1)
// ---------------------------------------------------------------------------------------
template <int CAP>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    rnea_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int n_ops, int n, int flags,
                const float *__restrict__ q, const float *__restrict__ qd, const float *__restrict__ qdd, int64_t B,
                float *__restrict__ tau, TileShape ts_q, int lds_per_wave, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx cx;
    if (!wave_begin(B, lds_per_wave, smem, cx)) return;
    const int lane = cx.lane;
    const int region = round4(WAVE * ts_q.Sp);
    float *lq = cx.lds, *lqd = lq + region, *lqdd = lqd + region, *ltau = lqdd + region;
    const bool lin_img = cx.full && ts_q.Sp == ts_q.S;

    tile_load(q + cx.b0 * n, cx.rows, ts_q, lq, lane, lin_img && (align & AL_Q));
    tile_load(qd + cx.b0 * n, cx.rows, ts_q, lqd, lane, lin_img && (align & AL_QD));
    if (qdd) tile_load(qdd + cx.b0 * n, cx.rows, ts_q, lqdd, lane, lin_img && (align & AL_QDD));
    wave_lds_sync();

    const bool live = lane < cx.rows;
    const int row = lane * ts_q.Sp;
    auto qf = [&](int d, float &a, float &v, float &acc) {
        a = live ? lq[row + d] : 0.0f;
        v = live ? lqd[row + d] : 0.0f;
        acc = (live && qdd) ? lqdd[row + d] : 0.0f;
    };
    auto tau_out = [&](int d, float v) { ltau[row + d] = v; };
    rnea_walk<CAP>(ops_f, ops_i, n_ops, flags, qf, tau_out);
    wave_lds_sync();
    tile_store(tau + cx.b0 * n, cx.rows, ts_q, ltau, lane, lin_img && (align & AL_TAU));
}

// ---------------------------------------------------------------------------------------
// host side: argument checks, launch geometry, dispatch on the walk capacity
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *a = "", long b = 0, long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

static TileShape make_shape(int S) {
    TileShape ts;
    ts.S = S;
    ts.Sp = pad_odd(S);
    ts.magic = (S >= 2) ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)S - 1) / (uint64_t)S) : 0u;
    return ts;
}

static inline uint32_t al16(const void *p, uint32_t bit) { return (p && (((uintptr_t)p) & 15u) == 0) ? bit : 0u; }

static int check_walk(const drm_walk *w) {
    if (!w) return fail(DRM_ERR_INVALID, "walk is NULL");
    if (!w->ops_f || !w->ops_i) return fail(DRM_ERR_INVALID, "walk tables are NULL");
    if (w->n_ops < 0 || w->n_ops > w->capacity)
        return fail(DRM_ERR_INVALID, "walk has %s%ld ops but capacity %ld", "", w->n_ops, w->capacity);
    if (w->capacity != 8 && w->capacity != 16 && w->capacity != 32 && w->capacity != 64)
        return fail(DRM_ERR_UNSUPPORTED, "walk capacity %s%ld is not one of 8/16/32/64", "", w->capacity);
    if (w->n_dofs < 1 || w->n_dofs > DRM_MAX_DOFS)
        return fail(DRM_ERR_UNSUPPORTED, "n_dofs %s%ld outside [1, %ld]", "", w->n_dofs, DRM_MAX_DOFS);
    if (w->n_slots < 0 || w->n_slots > DRM_MAX_SLOTS)
        return fail(DRM_ERR_UNSUPPORTED, "walk needs %s%ld save slots, kernels have %ld", "", w->n_slots, DRM_MAX_SLOTS);
    return DRM_OK;
}

struct Geometry {
    dim3 grid, block;
    size_t lds_bytes;
    int lds_per_wave;
};

// waves per block: as many as fit the LDS budget (<= 4); one wave per 64 samples.
static int make_geometry(int64_t B, int lds_floats_per_wave, Geometry &g) {
    const size_t per_wave = (size_t)round4(lds_floats_per_wave) * sizeof(float);
    if (per_wave > (size_t)MAX_LDS_BYTES)
        return fail(DRM_ERR_UNSUPPORTED, "tile needs %s%ld bytes of LDS per wave (max %ld)", "", (long)per_wave,
                    (long)MAX_LDS_BYTES);
    int wpb = MAX_WAVES_PER_BLOCK;
    while (wpb > 1 && per_wave * wpb > (size_t)64 * 1024) wpb >>= 1;
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    const int64_t blocks = (tiles + wpb - 1) / wpb;
    if (blocks > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    g.grid = dim3((unsigned)blocks);
    g.block = dim3(WAVE * wpb);
    g.lds_per_wave = round4(lds_floats_per_wave);
    g.lds_bytes = per_wave * wpb;
    return DRM_OK;
}

template <class K>
static int ensure_lds(K kernel, size_t bytes) {
    if (bytes > (size_t)64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    return DRM_OK;
}

static int launched() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return DRM_OK;
}

#define DRM_DISPATCH_CAP(cap, CALL)          \
    switch (cap) {                           \
    case 8: { constexpr int C = 8; CALL; } break;   \
    case 16: { constexpr int C = 16; CALL; } break; \
    case 32: { constexpr int C = 32; CALL; } break; \
    default: { constexpr int C = 64; CALL; } break; \
    }

} // namespace drm

using namespace drm;

extern "C" {

int drm_abi_version(void) { return DRM_ABI_VERSION; }

const char *drm_last_error(void) { return g_err; }

int drm_fk_jacobian(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin_jac,
                    float *ang_jac, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !lin_jac || !ang_jac) return fail(DRM_ERR_INVALID, "q / lin_jac / ang_jac must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs;
    const TileShape ts_q = make_shape(n), ts_j = make_shape(3 * n);
    Geometry g;
    rc = make_geometry(B, round4(WAVE * ts_q.Sp) + round4(WAVE * (ts_j.Sp > 3 ? ts_j.Sp : 3)), g);
    if (rc) return rc;
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT) | al16(lin_jac, AL_LIN) |
                           al16(ang_jac, AL_ANG);
    hipStream_t s = (hipStream_t)stream;
    DRM_DISPATCH_CAP(w->capacity, {
        rc = ensure_lds(fk_jacobian_kernel<C>, g.lds_bytes);
        if (rc) return rc;
        hipLaunchKernelGGL(fk_jacobian_kernel<C>, g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, w->n_ops, n,
                           w->dof_mask, q, B, pos, quat, lin_jac, ang_jac, ts_q, ts_j, g.lds_per_wave, align);
    })
    return launched();
}

int drm_fk(const drm_walk *w, const float *q, int64_t B, int32_t n_targets, float *pos, float *quat, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !pos || !quat) return fail(DRM_ERR_INVALID, "q / pos / quat must not be NULL");
    if (B < 0 || n_targets < 1) return fail(DRM_ERR_INVALID, "negative batch or no targets");
    if (n_targets > w->n_ops) return fail(DRM_ERR_INVALID, "more targets than ops in the walk");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs, T = n_targets;
    const TileShape ts_q = make_shape(n), ts_p = make_shape(3 * T);
    TileShape ts_r = make_shape(4 * T);
    Geometry g;
    rc = make_geometry(B, round4(WAVE * ts_q.Sp) + round4(WAVE * ts_p.Sp) + round4(WAVE * ts_r.Sp), g);
    if (rc) return rc;
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
    hipStream_t s = (hipStream_t)stream;
    DRM_DISPATCH_CAP(w->capacity, {
        rc = ensure_lds(fk_kernel<C>, g.lds_bytes);
        if (rc) return rc;
        hipLaunchKernelGGL(fk_kernel<C>, g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, w->n_ops, n, q, B, T, pos,
                           quat, ts_q, ts_p, ts_r, g.lds_per_wave, align);
    })
    return launched();
}

int drm_rnea(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags,
             float *tau, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !qd || !tau) return fail(DRM_ERR_INVALID, "q / qd / tau must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs;
    const TileShape ts_q = make_shape(n);
    Geometry g;
    rc = make_geometry(B, 4 * round4(WAVE * ts_q.Sp), g);
    if (rc) return rc;
    const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(qdd, AL_QDD) | al16(tau, AL_TAU);
    hipStream_t s = (hipStream_t)stream;
    DRM_DISPATCH_CAP(w->capacity, {
        rc = ensure_lds(rnea_kernel<C>, g.lds_bytes);
        if (rc) return rc;
        hipLaunchKernelGGL(rnea_kernel<C>, g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, w->n_ops, n, (int)flags,
                           q, qd, qdd, B, tau, ts_q, g.lds_per_wave, align);
    })
    return launched();
}

} // extern "C"
