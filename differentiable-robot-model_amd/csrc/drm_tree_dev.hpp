// drm_tree_dev.hpp — device-side plumbing shared by the loop-structured ("tree") kernels: K1 drm_fk, K2 drm_fk_jacobian,
// K3 drm_rnea, K6 drm_crba, K8 drm_forward_dynamics for every robot that is not a 7-DoF arm chain (drm_tree.hpp holds
// their per-sample arithmetic).
//
// Block layout.  A block works on tiles of 64 consecutive samples and has one wavefront per SEGMENT of the walk (sub-trees
// hanging off the fixed root are independent dynamics problems, include/drm_hip.h drm_walk.seg_begin; the kinematics
// kernels have one segment): lane = sample, wavefront = sub-tree — the per-link fan-out over wavefronts that K1 already did
// for the fingertips of a hand, now for the dynamics too (Allegro: four 5-link fingers per tile instead of one 20-link
// walk; four times the waves and a quarter of the per-wave LDS).
// Two shapes of launch:
//   * one block per tile (kinematics; dynamics of robots whose segments are all short — fingers — with both sweeps unrolled
//     and the per-op records in registers);
//   * a PERSISTENT grid (dynamics of robots with a long segment: rnea_records_kernel, crba_rows_kernel,
//     forward_dynamics_aba_kernel): as many blocks as the device holds at once (resident_blocks below), each looping over
//     tiles and owning a slice of caller-provided HBM scratch for what a sample needs between two sweeps.
// Shared by the block, in LDS:
//   * the op table of the walk (n_ops x 32 floats) and its two control-word columns, staged once per block: every
//     per-link constant is a broadcast LDS read, control words come back through v_readfirstlane; no scalar-memory
//     round trip inside the loops, no compile-time capacity;
//   * the input tiles (row-per-sample tensors <-> lane-per-sample access through an odd row stride, drm_common.hpp)
//     and the output tile, loaded / stored once per tile with coalesced 16-byte accesses.
// Private to a wavefront, in LDS: the save slots of its branch points (and whatever a kernel adds: a finger's block of H,
// the forces of the mass-matrix walk).
#pragma once

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

#include "drm_common.hpp"
#include "drm_tree.hpp"

namespace drm {

struct TreeArgs { // by value in the kernel arguments
    const float *ops_f;
    const int32_t *ops_i;
    int32_t cap, n_ops, n, n_slots, n_segments, prefix_end, max_seg_ops;
    int32_t seg_begin[DRM_MAX_SEGMENTS + 1];
    int32_t seg_dof_lo[DRM_MAX_SEGMENTS];
    int32_t seg_dof_cnt[DRM_MAX_SEGMENTS];
    int32_t wave_off[DRM_MAX_SEGMENTS]; // LDS offset (floats) of every wavefront's private area, set by the launcher
};

// single = true: ignore the segments, one wavefront walks everything (what a launch falls back to when the private areas
// of all the segments do not fit a CU's LDS side by side)
// One wavefront per segment pays when the segments are of similar length (the fingers of a hand).  A long segment next to short
// ones (a mobile manipulator: two wheels of one op each beside a torso + head + arm + gripper tree of twelve) leaves the short
// segments' wavefronts idle while they hold registers and LDS: such walks run all their segments through ONE wavefront per tile
// (twice the tiles in flight per CU for the same number of wavefronts).  Walks whose segments are all short keep the fan-out.
// Used by the inverse- and forward-dynamics loop kernels (Fetch at 2^20: 262 -> 231 us, 715 -> 605 us); the mass-matrix and the
// reverse-mode loop kernels are faster fanned out even then (one triangle / one set of records per segment: 662 vs 989 us,
// 971 vs 1 106 us).
static inline bool segments_worth_fanning_out(const drm_walk *w) {
    if (w->n_segments < 2 || w->n_segments > DRM_MAX_SEGMENTS) return false;
    int lo = 1 << 30, hi = 0;
    for (int s = 0; s < w->n_segments; ++s) {
        const int len = w->seg_begin[s + 1] - w->seg_begin[s];
        lo = len < lo ? len : lo;
        hi = len > hi ? len : hi;
    }
#ifdef DRM_ALWAYS_FAN_OUT
    return true;
#else
    return hi <= 6 || 2 * lo >= hi;
#endif
}

static inline TreeArgs tree_args(const drm_walk *w, bool single = false) {
    TreeArgs a;
    a.ops_f = w->ops_f; a.ops_i = w->ops_i;
    a.cap = w->capacity; a.n_ops = w->n_ops; a.n = w->n_dofs; a.n_slots = w->n_slots;
    a.n_segments = (!single && w->n_segments >= 1 && w->n_segments <= DRM_MAX_SEGMENTS) ? w->n_segments : 1;
    a.prefix_end = a.n_segments > 1 ? w->prefix_end : 0;
    a.max_seg_ops = 0;
    for (int s = 0; s < DRM_MAX_SEGMENTS; ++s) {
        const bool on = s < a.n_segments;
        a.seg_begin[s] = on ? (a.n_segments > 1 ? w->seg_begin[s] : 0) : w->n_ops;
        a.seg_dof_lo[s] = on ? (a.n_segments > 1 ? w->seg_dof_lo[s] : 0) : 0;
        a.seg_dof_cnt[s] = on ? (a.n_segments > 1 ? w->seg_dof_cnt[s] : w->n_dofs) : 0;
    }
    a.seg_begin[a.n_segments] = w->n_ops;
    for (int s = a.n_segments + 1; s <= DRM_MAX_SEGMENTS; ++s) a.seg_begin[s] = w->n_ops;
    for (int s = 0; s < a.n_segments; ++s) {
        const int len = a.seg_begin[s + 1] - a.seg_begin[s];
        if (len > a.max_seg_ops) a.max_seg_ops = len;
    }
    for (int s = 0; s < DRM_MAX_SEGMENTS; ++s) a.wave_off[s] = 0;
    return a;
}

// Lay the wavefronts' private LDS areas out one after the other behind `shared` floats: wavefront s gets
// ops_s * per_op + per_wave + extra(s) floats.  Returns the total in floats.
template <class EXTRA>
static inline size_t layout_waves(TreeArgs &a, size_t shared, int per_op, int per_wave, EXTRA extra) {
    size_t off = shared;
    for (int s = 0; s < a.n_segments; ++s) {
        a.wave_off[s] = (int32_t)off;
        off += (size_t)round4((a.seg_begin[s + 1] - a.seg_begin[s]) * per_op + per_wave + extra(s));
    }
    return off;
}

// validity of the segment description handed over in a drm_walk (host side)
static inline bool segments_ok(const drm_walk *w) {
    if (w->n_segments < 1 || w->n_segments > DRM_MAX_SEGMENTS) return false;
    if (w->n_segments == 1) return true;
    if (w->prefix_end < 0 || w->seg_begin[0] != w->prefix_end || w->seg_begin[w->n_segments] != w->n_ops) return false;
    for (int s = 0; s < w->n_segments; ++s) {
        if (w->seg_begin[s + 1] < w->seg_begin[s]) return false;
        if (w->seg_dof_lo[s] < 0 || w->seg_dof_cnt[s] < 0 || w->seg_dof_lo[s] + w->seg_dof_cnt[s] > w->n_dofs) return false;
    }
    return true;
}

// floats of LDS the staged table takes (op rows + two control-word columns)
__host__ __device__ static inline int table_lds_floats(int n_ops) { return round4(n_ops * DRM_OPF_STRIDE) + round4(2 * (n_ops > 0 ? n_ops : 1)); }

struct TableLds {
    const float *f;   // [n_ops][32]
    const int *w;     // [2][n_ops]
    int n_ops;
    __device__ __forceinline__ const float *row(int k) const { return f + k * DRM_OPF_STRIDE; }
    // the CTL interface of drm_tree.hpp: request the two control words of op k, then make a requested word wave-uniform
    __device__ __forceinline__ void raw(int k, int &r0, int &r1) const { r0 = w[k]; r1 = w[n_ops + k]; }
    __device__ __forceinline__ int uniform(int r) const { return __builtin_amdgcn_readfirstlane(r); }
};

// all threads of the block copy the table into LDS (16 bytes per thread and round); the caller synchronises
__device__ __forceinline__ TableLds stage_tree_table(const TreeArgs &a, float *lds) {
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    float *lf = lds;
    int *lw = reinterpret_cast<int *>(lds + round4(a.n_ops * DRM_OPF_STRIDE));
    const unsigned n4 = (unsigned)a.n_ops * (DRM_OPF_STRIDE / 4);
    const float4 *g4 = reinterpret_cast<const float4 *>(a.ops_f); // ops_f is 16-byte aligned (checked by the launcher)
    for (unsigned i = tid; i < n4; i += nt) reinterpret_cast<float4 *>(lf)[i] = g4[i];
    for (unsigned i = tid; i < (unsigned)a.n_ops; i += nt) {
        lw[i] = a.ops_i[DRM_OPI_W0 * a.cap + i];
        lw[a.n_ops + i] = a.ops_i[DRM_OPI_W1 * a.cap + i];
    }
    TableLds t;
    t.f = lf; t.w = lw; t.n_ops = a.n_ops;
    return t;
}

// ---- save slots in LDS, [slot][component][64] ------------------------------------------------------------------
__device__ __forceinline__ void lds_put_pose(float *slots, int s, unsigned lane, const PoseP &P) {
    float *b = slots + s * (12 * WAVE) + lane;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        b[(4 * c + 0) * WAVE] = P.A[c][0]; b[(4 * c + 1) * WAVE] = P.A[c][1];
        b[(4 * c + 2) * WAVE] = P.B[c][0]; b[(4 * c + 3) * WAVE] = P.B[c][1];
    }
}
__device__ __forceinline__ void lds_get_pose(const float *slots, int s, unsigned lane, PoseP &P) {
    const float *b = slots + s * (12 * WAVE) + lane;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        P.A[c] = f2_make(b[(4 * c + 0) * WAVE], b[(4 * c + 1) * WAVE]);
        P.B[c] = f2_make(b[(4 * c + 2) * WAVE], b[(4 * c + 3) * WAVE]);
    }
}
__device__ __forceinline__ void lds_put_motion(float *slots, int s, unsigned lane, const Motion &M) {
    float *b = slots + s * (12 * WAVE) + lane;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        b[i * WAVE] = M.wa[i][0]; b[(3 + i) * WAVE] = M.va[i][0]; b[(6 + i) * WAVE] = M.wa[i][1]; b[(9 + i) * WAVE] = M.va[i][1];
    }
}
__device__ __forceinline__ void lds_get_motion(const float *slots, int s, unsigned lane, Motion &M) {
    const float *b = slots + s * (12 * WAVE) + lane;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M.wa[i] = f2_make(b[i * WAVE], b[(6 + i) * WAVE]);
        M.va[i] = f2_make(b[(3 + i) * WAVE], b[(9 + i) * WAVE]);
    }
}
__device__ __forceinline__ void lds_add_force(float *slots, int s, unsigned lane, const Force &F) {
    float *b = slots + s * (6 * WAVE) + lane;
#pragma unroll
    for (int i = 0; i < 3; ++i) { b[i * WAVE] += F.la[i][0]; b[(3 + i) * WAVE] += F.la[i][1]; }
}
__device__ __forceinline__ void lds_take_force(float *slots, int s, unsigned lane, Force &F) {
    float *b = slots + s * (6 * WAVE) + lane;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        F.la[i] += f2_make(b[i * WAVE], b[(3 + i) * WAVE]);
        b[i * WAVE] = 0.0f; b[(3 + i) * WAVE] = 0.0f;
    }
}
__device__ __forceinline__ void lds_add_inertia(float *slots, int s, unsigned lane, const Inertia &a) {
    float *b = slots + s * (10 * WAVE) + lane;
    b[0] += a.m;
#pragma unroll
    for (int i = 0; i < 3; ++i) b[(1 + i) * WAVE] += a.h[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) b[(4 + i) * WAVE] += a.I[i];
}
__device__ __forceinline__ void lds_take_inertia(float *slots, int s, unsigned lane, Inertia &a) {
    float *b = slots + s * (10 * WAVE) + lane;
    a.m += b[0]; b[0] = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { a.h[i] += b[(1 + i) * WAVE]; b[(1 + i) * WAVE] = 0.0f; }
#pragma unroll
    for (int i = 0; i < 6; ++i) { a.I[i] += b[(4 + i) * WAVE]; b[(4 + i) * WAVE] = 0.0f; }
}

// one block = one tile: rows of the tile and whether it is full
struct TileCtx {
    int64_t b0;
    int rows;
    bool full;
};
__device__ __forceinline__ TileCtx tile_begin(int64_t B) {
    TileCtx t;
    t.b0 = (int64_t)blockIdx.x * WAVE;
    const int64_t left = B - t.b0;
    t.rows = left < WAVE ? (int)left : WAVE;
    t.full = t.rows == WAVE;
    return t;
}

template <class K>
static int ensure_lds_tree(K kernel, size_t bytes) {
    if (bytes > (size_t)MAX_LDS_BYTES)
        return fail(DRM_ERR_UNSUPPORTED, "the per-link records of this robot need %s%ld bytes of LDS per 64-sample tile (max %ld)",
                    "", (long)bytes, (long)MAX_LDS_BYTES);
    return ensure_lds(kernel, bytes);
}


// How many blocks of `kernel` (threads per block, dynamic LDS bytes) the current device holds at once: the grid of the
// PERSISTENT kernels, whose blocks loop over tiles and own a slice of caller-provided scratch each.  The runtime's own
// occupancy rule (LDS, registers) per CU times the number of CUs; cached per kernel, geometry and device.
// DRM_MAX_RESIDENT_BLOCKS (environment, read per query): clamps the grid of every persistent kernel, so that tests reach the
// several-tiles-per-wavefront paths (LDS staging of the next tile, partial sums carried across tiles) with small batches.
static inline int clamp_resident(int blocks) {
    const char *e = getenv("DRM_MAX_RESIDENT_BLOCKS");
    if (!e || !*e) return blocks;
    const long v = strtol(e, nullptr, 10);
    return v >= 1 && v < blocks ? (int)v : blocks;
}
template <class K>
static int resident_blocks(K kernel, int threads, size_t lds, int &out) {
    static std::mutex mu;
    static std::map<std::tuple<const void *, int, int, size_t>, int> cache; // (kernels of one signature share this instantiation)
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipGetDevice failed");
    const auto key = std::make_tuple(reinterpret_cast<const void *>(kernel), dev, threads, lds);
    auto it = cache.find(key);
    if (it == cache.end()) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1)
            return fail(DRM_ERR_LAUNCH, "occupancy query failed");
        it = cache.emplace(key, per_cu * cus).first;
    }
    out = clamp_resident(it->second);
    return DRM_OK;
}
// the same for a kernel of a loaded code object (drm_walk.special[]: per-robot kernels, one wavefront per block, static LDS)
static int resident_blocks_module(hipFunction_t fn, int threads, int &out) {
    static std::mutex mu;
    static std::map<std::tuple<const void *, int, int>, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipGetDevice failed");
    const auto key = std::make_tuple(reinterpret_cast<const void *>(fn), dev, threads);
    auto it = cache.find(key);
    if (it == cache.end()) {
        int per_cu = 0, cus = 0;
        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1)
            return fail(DRM_ERR_LAUNCH, "occupancy query failed");
        it = cache.emplace(key, per_cu * cus).first;
    }
    out = clamp_resident(it->second);
    return DRM_OK;
}

} // namespace drm
