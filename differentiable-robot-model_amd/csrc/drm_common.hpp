// drm_common.hpp — device-side tile I/O and host-side launch helpers shared by the kernels.
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
//   * everything derived from the wave index is made provably wave-uniform
//     (readfirstlane) so tile bases live in SGPRs and per-lane offsets are 32-bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/drm_hip.h"

namespace drm {

constexpr int WAVE = 64;
constexpr int MAX_WAVES_PER_BLOCK = 4;
constexpr int MAX_LDS_BYTES = 160 * 1024;

// Ordering point between LDS writes of some lanes and LDS reads of other lanes of the SAME
// wave: the LDS executes a wave's instructions in order, so only the compiler must be kept
// from reordering.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- development only (-DDRM_TIMELINE variant builds, tools/timeline.py): per-wave time stamps (the 100 MHz real-time counter,
// the same clock on every XCD) at named points of a kernel, one record of 8 slots per wavefront, in a buffer owned by the
// translation unit; drm_tl_read_<unit>() copies it out.  DRM_STAMP(slot) / DRM_STAMP_DRAINED(slot) compile to nothing in the
// shipped library.
#ifdef DRM_TIMELINE
constexpr int TL_SLOTS = 8, TL_WAVES = 1 << 16;
__device__ static unsigned long long tl_buf[TL_SLOTS * TL_WAVES];
__device__ __forceinline__ void tl_stamp(int slot, bool drained) {
    __builtin_amdgcn_sched_barrier(0);
    if (drained) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t = wall_clock64();
    const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0 && w < (unsigned)TL_WAVES) tl_buf[w * TL_SLOTS + slot] = t;
    if (slot == 0 && (threadIdx.x & 63u) == 0 && w < (unsigned)TL_WAVES) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tl_buf[w * TL_SLOTS + 7] = ((unsigned long long)xcc << 32) | hw;
    }
    __builtin_amdgcn_sched_barrier(0);
}
#define DRM_STAMP(slot) ::drm::tl_stamp(slot, false)
#define DRM_STAMP_DRAINED(slot) ::drm::tl_stamp(slot, true)
#define DRM_TL_READER(unit)                                                                                                      \
    extern "C" __attribute__((visibility("default"))) int drm_tl_read_##unit(void *host, size_t bytes, int clear) {             \
        if (hipMemcpyFromSymbol(host, HIP_SYMBOL(::drm::tl_buf), bytes) != hipSuccess) return -1;                                \
        if (clear) {                                                                                                             \
            void *d = nullptr;                                                                                                   \
            if (hipGetSymbolAddress(&d, HIP_SYMBOL(::drm::tl_buf)) != hipSuccess || hipMemset(d, 0, sizeof(unsigned long long) * ::drm::TL_SLOTS * ::drm::TL_WAVES) != hipSuccess) return -2; \
        }                                                                                                                        \
        return 0;                                                                                                                \
    }
#else
#define DRM_STAMP(slot) ((void)0)
#define DRM_STAMP_DRAINED(slot) ((void)0)
#define DRM_TL_READER(unit)
#endif

// keep a just-loaded value materialised HERE: stops the compiler from sinking the load into the
// predicated block that consumes it (which would serialise load -> wait -> store per iteration)
__device__ __forceinline__ void pin(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// 16-byte store of an OUTPUT tensor that writes THROUGH the XCD's L2 (`sc1`).  A plain store leaves its line dirty in the
// L2 and the end-of-kernel release then writes all of them back at once, serialised behind the last wave: 12.8 MB of
// Jacobians cost ~1.8 us there.  Written through, the bytes leave while the other waves (and this wave's later links)
// still compute and the release finds nothing to do: the metric launch (Panda, 65 536 samples) went 4.83 -> 3.92 us, a
// launch that only moves its bytes 3.67 -> 2.90 us (tools/ubench/metric_lab.hip, profiles/r02_metric_lab.txt); `nt`
// alone does not do it.  Outputs are never re-read by the kernel that wrote them, so dropping the line from the L2
// costs nothing.
// NT adds `nt`: the line is not kept in the 256 MiB Infinity Cache either.  That is right for outputs that are larger
// than the cache (they would only evict each other: 2^22 samples 159 -> 130 us = 7.2 TB/s) and wrong for outputs that fit
// (2^20 samples: 33 -> 35 us, and the consumer of a 200 MB result finds it on die), so the launchers choose by size
// (stream_past_llc below).
// (Inline asm: there is no builtin for a 16-byte store with cache-policy bits on a flat pointer.  The store reads its
// data registers for one more state, hence the s_nop inside the statement — cdna_hip_programming.md §5.7.)
typedef float drm_v4f __attribute__((ext_vector_type(4)));
template <bool NT = false>
__device__ __forceinline__ void store16_wt(void *p, float4 v) {
    const drm_v4f x = {v.x, v.y, v.z, v.w};
    if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
// outputs of a launch that exceed the Infinity Cache are streamed past it
static inline bool stream_past_llc(int64_t output_bytes) { return output_bytes > (int64_t)256 * 1024 * 1024; }

// Touch the first `LINES` 64-byte lines of every op of the walk table with scalar loads issued
// back to back, so the (cold) misses overlap in ONE latency round instead of one per op.
// Split in two so the kernel can put the q tile's global loads between issue and wait.
template <int N>
struct WarmRegs {
    float x[N];
};
template <int CAP, int LINES>
__device__ __forceinline__ void warm_walk_issue(const float *__restrict__ ops_f, WarmRegs<CAP * LINES> &w) {
#pragma unroll
    for (int i = 0; i < CAP * LINES; ++i) w.x[i] = ops_f[(i / LINES) * DRM_OPF_STRIDE + (i % LINES) * 16];
}
template <int N>
__device__ __forceinline__ void warm_walk_wait(const WarmRegs<N> &w) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" ::"s"(w.x[i]));
}

__host__ __device__ constexpr int pad_odd(int S) { return S | 1; }
__host__ __device__ constexpr int round4(int x) { return (x + 3) & ~3; }
// ceil(2^32 / S): floor(w / S) == umulhi(w, magic) for w * S < 2^32
inline uint32_t div_magic(int S) {
    return (S >= 2) ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)S - 1) / (uint64_t)S) : 0u;
}

// A [rows, S] row-major float tile (rows <= 64) in HBM <-> wave-private LDS with row stride
// Sp = S | 1: word w of the tile lives at LDS word  w + (S even ? w / S : 0).
__device__ __forceinline__ unsigned lds_word(unsigned w, bool padded, uint32_t magic) {
    return padded ? w + __umulhi(w, magic) : w;
}

// ---- HBM -> LDS --------------------------------------------------------------------
// fast: the tile is full (64 rows), S is odd (the LDS image is linear) and g is 16-byte
// aligned -> float4 per lane.  S_CT > 0 fixes S at compile time (all loads are issued
// before the first LDS write).
// vec: the tile is full and g is 16-byte aligned but S is even (padded LDS image): 16-byte global accesses, the
// four floats go to their padded LDS words one by one.
template <int S_CT>
__device__ __forceinline__ void tile_load(const float *__restrict__ g, int rows, int S_rt, uint32_t magic, float *lds,
                                          unsigned lane, bool fast, bool vec = false) {
    const int S = S_CT ? S_CT : S_rt;
    if (!fast && vec && !(S & 1)) {
        const float4 *g4 = reinterpret_cast<const float4 *>(g);
        const unsigned nvec = 16u * (unsigned)S;
#pragma unroll 2
        for (unsigned i = lane; i < nvec; i += 64u) {
            const float4 v = g4[i];
            const unsigned w = 4u * i;
            lds[lds_word(w, true, magic)] = v.x;
            lds[lds_word(w + 1u, true, magic)] = v.y;
            lds[lds_word(w + 2u, true, magic)] = v.z;
            lds[lds_word(w + 3u, true, magic)] = v.w;
        }
        return;
    }
    if (fast) {
        const char *gb = reinterpret_cast<const char *>(g); // uniform base + 32-bit byte offset per lane
        float4 *l4 = reinterpret_cast<float4 *>(lds);
        const unsigned nvec = 16u * (unsigned)S;
        if constexpr (S_CT > 0) {
            // branch-free: every lane loads in every iteration (index clamped into the tile), so all
            // loads are in flight before the first LDS write; only the LDS write is predicated
            constexpr unsigned IT = (16u * S_CT + 63u) / 64u;
            float4 v[IT];
#pragma unroll
            for (unsigned it = 0; it < IT; ++it) {
                const unsigned i = lane + 64u * it;
                v[it] = *reinterpret_cast<const float4 *>(gb + (i < nvec ? i : nvec - 1u) * 16u);
            }
#pragma unroll
            for (unsigned it = 0; it < IT; ++it) pin(v[it]);
#pragma unroll
            for (unsigned it = 0; it < IT; ++it) {
                const unsigned i = lane + 64u * it;
                if ((it + 1u) * 64u <= nvec || i < nvec) l4[i] = v[it];
            }
        } else {
#pragma unroll 2
            for (unsigned i = lane; i < nvec; i += 64u) l4[i] = *reinterpret_cast<const float4 *>(gb + i * 16u);
        }
    } else {
        const unsigned total = (unsigned)(rows * S);
        const bool padded = !(S & 1);
#pragma unroll 4
        for (unsigned w = lane; w < total; w += 64u) lds[lds_word(w, padded, magic)] = g[w];
    }
}

// ---- LDS -> HBM, same conventions ---------------------------------------------------
template <int S_CT, bool NT = false>
__device__ __forceinline__ void tile_store(float *__restrict__ g, int rows, int S_rt, uint32_t magic, const float *lds,
                                           unsigned lane, bool fast, bool vec = false) {
    const int S = S_CT ? S_CT : S_rt;
    if (!fast && vec && !(S & 1)) {
        float4 *g4 = reinterpret_cast<float4 *>(g);
        const unsigned nvec = 16u * (unsigned)S;
        for (unsigned i = lane; i < nvec; i += 64u) { // (a loop with an asm statement is not unrolled by hipcc)
            const unsigned w = 4u * i;
            store16_wt<NT>(g4 + i, make_float4(lds[lds_word(w, true, magic)], lds[lds_word(w + 1u, true, magic)],
                                           lds[lds_word(w + 2u, true, magic)], lds[lds_word(w + 3u, true, magic)]));
        }
        return;
    }
    if (fast) {
        char *gb = reinterpret_cast<char *>(g);
        const float4 *l4 = reinterpret_cast<const float4 *>(lds);
        const unsigned nvec = 16u * (unsigned)S;
        if constexpr (S_CT > 0) {
            constexpr unsigned IT = (16u * S_CT + 63u) / 64u;
            float4 v[IT];
#pragma unroll
            for (unsigned it = 0; it < IT; ++it) {
                const unsigned i = lane + 64u * it;
                v[it] = l4[i < nvec ? i : nvec - 1u];
            }
#pragma unroll
            for (unsigned it = 0; it < IT; ++it) {
                const unsigned i = lane + 64u * it;
                if ((it + 1u) * 64u <= nvec || i < nvec) store16_wt<NT>(gb + i * 16u, v[it]);
            }
        } else {
            for (unsigned i = lane; i < nvec; i += 64u) store16_wt<NT>(gb + i * 16u, l4[i]);
        }
    } else {
        const unsigned total = (unsigned)(rows * S);
        const bool padded = !(S & 1);
#pragma unroll 4
        for (unsigned w = lane; w < total; w += 64u) g[w] = lds[lds_word(w, padded, magic)];
    }
}

// LDS -> HBM of a whole [rows, S] tile by ALL threads of a block (the tile of a fanned-out launch belongs to the block, not
// to one wavefront): 16-byte accesses when the tile is full and g is 16-byte aligned (`vec`), 4-byte ones otherwise.
template <bool NT = false>
__device__ __forceinline__ void block_tile_store(float *__restrict__ g, int rows, int S, uint32_t magic, const float *lds,
                                                 bool vec) {
    const bool padded = !(S & 1);
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    if (vec) {
        const unsigned nvec = 16u * (unsigned)S; // rows == 64
        for (unsigned i = tid; i < nvec; i += nt) {
            const unsigned w = 4u * i;
            store16_wt<NT>(g + w, make_float4(lds[lds_word(w, padded, magic)], lds[lds_word(w + 1u, padded, magic)],
                                              lds[lds_word(w + 2u, padded, magic)], lds[lds_word(w + 3u, padded, magic)]));
        }
    } else {
        const unsigned total = (unsigned)(rows * S);
        for (unsigned w = tid; w < total; w += nt) g[w] = lds[lds_word(w, padded, magic)];
    }
}

// Sum over the 64 lanes of a wave, result valid in lane 63: DPP row shifts, then row broadcasts (fixed order).
// Lanes that are shifted in from outside a row or masked off receive `old` = 0, i.e. they add nothing.
#define DRM_DPP_ADD(v, ctrl, row_mask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
__device__ __forceinline__ float wave_sum_lane63(float v) {
    DRM_DPP_ADD(v, 0x111, 0xf); // row_shr:1
    DRM_DPP_ADD(v, 0x112, 0xf); // row_shr:2
    DRM_DPP_ADD(v, 0x114, 0xf); // row_shr:4
    DRM_DPP_ADD(v, 0x118, 0xf); // row_shr:8   -> lane 15 of every row holds the row's sum
    DRM_DPP_ADD(v, 0x142, 0xa); // row_bcast:15 into rows 1 and 3
    DRM_DPP_ADD(v, 0x143, 0xc); // row_bcast:31 into rows 2 and 3
    return v;
}
#undef DRM_DPP_ADD
// N cross-lane sums, G at a time: the DPP steps of a group interleave — a branch or an LDS update after every single sum serialised
// them: 1.4 us of a 5 us wavefront in drm_fk_mse at 16 384 rows, profiles/r06_timeline_links.txt — while only G totals are live at once
// (a whole row's 26 cost a streaming kernel its occupancy).  value(j): this lane's addend of sum j; sink(j, total): runs on lane 63.
template <int N, int G = 4, typename V, typename S>
__device__ __forceinline__ void wave_sums_lane63(unsigned lane, V &&value, S &&sink) {
#pragma unroll
    for (int j0 = 0; j0 < N; j0 += G) {
        float t[G];
#pragma unroll
        for (int u = 0; u < G; ++u)
            if (j0 + u < N) t[u] = wave_sum_lane63(value(j0 + u));
        if (lane == 63u) {
#pragma unroll
            for (int u = 0; u < G; ++u)
                if (j0 + u < N) sink(j0 + u, t[u]);
        }
    }
}

// The walk's constant rows (CAP x 32 floats, 0.5 - 4 KB) -> wave-private LDS, once per wave: every later read of
// a link constant is a broadcast LDS read (in-order returns, so the compiler can wait per link) instead of a scalar
// load whose latency a lone wave cannot hide.  16 bytes per lane and round when the table is 16-byte aligned.
template <int CAP>
__device__ __forceinline__ void stage_table(const float *__restrict__ ops_f, float *lds, unsigned lane, bool aligned) {
    constexpr unsigned N4 = CAP * DRM_OPF_STRIDE / 4;
    if (aligned) {
        const float4 *g4 = reinterpret_cast<const float4 *>(ops_f);
        float4 *l4 = reinterpret_cast<float4 *>(lds);
        constexpr unsigned IT = (N4 + WAVE - 1) / WAVE;
        float4 v[IT];
#pragma unroll
        for (unsigned it = 0; it < IT; ++it) {
            const unsigned i = lane + WAVE * it;
            v[it] = g4[i < N4 ? i : N4 - 1u];
        }
#pragma unroll
        for (unsigned it = 0; it < IT; ++it) pin(v[it]);
#pragma unroll
        for (unsigned it = 0; it < IT; ++it) {
            const unsigned i = lane + WAVE * it;
            if ((it + 1u) * WAVE <= N4 || i < N4) l4[i] = v[it];
        }
    } else {
#pragma unroll 4
        for (unsigned i = lane; i < CAP * DRM_OPF_STRIDE; i += WAVE) lds[i] = ops_f[i];
    }
}

// Per-robot static kernels (drm_static.hpp) of walks with at least this many ops run ONE wavefront per SIMD whatever is done (their
// register files are full): they get the forms that hide latency inside a wavefront — persistent grid, the next tile's rows staged
// in LDS, constants one op ahead.  Smaller robots keep two or three wavefronts per SIMD with the plain one-tile-per-block form, and
// the latency forms cost them that (Panda with gripper, 9 ops: inverse dynamics 58 -> 73 us, forward dynamics 125 -> 172 us at 2^20
// rows when they were applied to every robot; Jaco, 12 ops: 121 -> 108 / 247 -> 240; Fetch, 14 ops: 124 -> 104 / 330 -> 235).
constexpr int STATIC_LONE_OPS = 12;
#ifndef DRM_BWD_MAX_WAVES
#define DRM_BWD_MAX_WAVES 2048 /* 256 CUs x 4 SIMDs x 2 */
#endif
constexpr int BWD_MAX_WAVES = DRM_BWD_MAX_WAVES; // waves of a backward launch = rows of its partial sums

// Sum of the partial-sum rows a backward launch left behind (one row of NV floats per wave), in a FIXED order so that
// parameter gradients are bit-stable from run to run: a block of REDUCE_WAVES wavefronts owns 64 consecutive
// columns (lane = column: every row is read as one 256-byte segment); wavefront w adds rows w, w+16, w+32, ... into
// REDUCE_UNROLL independent running sums (that many rows in flight per wavefront: the loads are latency-bound),
// merges them pairwise, and wavefront 0 adds the 16 wave sums in order.  Returns the column total on wavefront 0.
constexpr int REDUCE_WAVES = 16, REDUCE_UNROLL = 16;
__device__ __forceinline__ float column_sum(const float *__restrict__ partials, int n_rows, int NV, int column, bool live,
                                            float (*lds)[WAVE]) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const float *p = partials + (live ? column : 0);
    float s[REDUCE_UNROLL];
#pragma unroll
    for (int u = 0; u < REDUCE_UNROLL; ++u) s[u] = 0.0f;
    int r = wave;
    for (; r + (REDUCE_UNROLL - 1) * REDUCE_WAVES < n_rows; r += REDUCE_UNROLL * REDUCE_WAVES) {
        float v[REDUCE_UNROLL];
#pragma unroll
        for (int u = 0; u < REDUCE_UNROLL; ++u) v[u] = p[(int64_t)(r + u * REDUCE_WAVES) * NV];
#pragma unroll
        for (int u = 0; u < REDUCE_UNROLL; ++u) s[u] += v[u];
    }
#pragma unroll
    for (int u = 0; u < REDUCE_UNROLL; ++u) // the last, partial round: same slots, rows past the end add nothing
        s[u] += (r + u * REDUCE_WAVES < n_rows) ? p[(int64_t)(r + u * REDUCE_WAVES) * NV] : 0.0f;
#pragma unroll
    for (int w = REDUCE_UNROLL / 2; w >= 1; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) s[u] += s[u + w];
    lds[wave][lane] = live ? s[0] : 0.0f;
    __syncthreads();
    float total = 0.0f;
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < REDUCE_WAVES; ++w) total += lds[w][lane];
    }
    return total;
}

struct WaveCtx {
    unsigned lane;
    int rows;    // valid samples of this tile (wave-uniform)
    int64_t b0;  // first sample of this tile  (wave-uniform)
    float *lds;  // wave-private LDS           (wave-uniform)
    bool full;   // rows == 64
};

__device__ __forceinline__ bool wave_begin(int64_t B, int lds_floats_per_wave, float *smem, WaveCtx &cx) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    cx.lane = threadIdx.x & 63u;
    const int64_t tile = (int64_t)blockIdx.x * (int)(blockDim.x >> 6) + wave;
    cx.b0 = tile * WAVE;
    if (cx.b0 >= B) return false;
    const int64_t left = B - cx.b0;
    cx.rows = left < WAVE ? (int)left : WAVE;
    cx.full = cx.rows == WAVE;
    cx.lds = smem + wave * lds_floats_per_wave;
    return true;
}

// alignment bits handed over by the launcher (bit set = pointer is 16-byte aligned)
enum : uint32_t { AL_Q = 1, AL_QD = 2, AL_QDD = 4, AL_POS = 8, AL_QUAT = 16, AL_LIN = 32, AL_ANG = 64, AL_TAU = 128, AL_TABLE = 256 };

// ---- host side ----------------------------------------------------------------------
int fail(int code, const char *fmt, const char *a = "", long b = 0, long c = 0);
int check_walk(const drm_walk *w);
int launched();

static inline uint32_t al16(const void *p, uint32_t bit) { return (p && (((uintptr_t)p) & 15u) == 0) ? bit : 0u; }

struct Geometry {
    dim3 grid, block;
    size_t lds_bytes;
    int lds_per_wave; // floats
};

// waves per block: as many as fit a 64 KiB LDS budget (<= 4); one wave per 64 samples.
int make_geometry(int64_t B, int lds_floats_per_wave, Geometry &g);

// Scratch of the walks whose FULL, ALIGNED tiles run straight-line kernels that need none (7-DoF arms, arms that carry a hand): the
// size queries (drm_rnea / drm_crba / drm_forward_dynamics _scratch_floats) cannot see the caller's pointers, so they always size
// for min(tiles, MISALIGNED_TILES) tiles of the loop kernels — enough for a ragged tail (one tile) and for a call whose pointers
// are not 16-byte aligned, which runs the persistent loop kernel on at most that many blocks instead of being refused (ABI 9).
constexpr int MISALIGNED_TILES = 64;
static inline int64_t fast_path_scratch_tiles(int64_t B) {
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    return tiles < MISALIGNED_TILES ? tiles : MISALIGNED_TILES;
}

// waves of a backward launch (each strides over tiles and writes one row of partial sums), a multiple of wpb
static inline int backward_waves(int64_t B, int wpb) {
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    int64_t waves = tiles < BWD_MAX_WAVES ? tiles : BWD_MAX_WAVES;
    waves = (waves + wpb - 1) / wpb * wpb;
    return (int)(waves < 1 ? wpb : waves);
}

// drm_arm_kernels.hip: the serial-chain kernels of K1 / K2 / K3 (full 64-row tiles, 16-byte aligned pointers)
int64_t launch_fk_arm(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, hipStream_t s);
void launch_fk_jacobian_arm(const float *ops_f, const float *q, int n_tiles, float *pos, float *quat, float *lin_jac,
                            float *ang_jac, hipStream_t s);
// links the dynamics sweeps of an arm-chain walk visit: n_dofs when nothing follows the moving joints (fixed tail folded into
// the last moving link by the host, flatten.fold_link_table), else the whole capacity
inline int arm_links(const drm_walk *w) { return w->n_ops == w->n_dofs ? w->n_dofs : w->capacity; }
int64_t launch_rnea_fingers(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau,
                            hipStream_t s);
// the grid of the streaming two-samples-per-lane arm kernels (csrc/drm_arm_stream.hpp): two wavefronts per SIMD on every CU
int arm_stream_grid(int n_pairs);
// launches of at least this many 128-row pairs of tiles run a walk's constant-folded arm kernel (drm_walk.special[DRM_SPECIAL_*_ARM])
#ifndef DRM_ARM_STATIC_MIN_PAIRS
#define DRM_ARM_STATIC_MIN_PAIRS 1024 /* 131 072 rows: one two-sample wavefront per SIMD */
#endif
void launch_rnea_arm(const float *ops_f, int links, const float *q, const float *qd, const float *qdd, int n_tiles, int flags,
                     float *tau, hipStream_t s);
void launch_fk_rnea_arm(const float *ops_f, const float *ops_tail, int links, const float *q, const float *qd, const float *qdd, int n_tiles, int flags,
                        float *tau, float *pos, float *quat, hipStream_t s);

// drm_chain_kernels.hip: straight-line kernels for any serial chain of capacity 8 / 12 / 16 (DRM_WALK_SERIAL_CHAIN); they
// return the rows they covered (full 64-row tiles), 0 when the call does not qualify
int64_t launch_chain_fk_jacobian(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin, float *ang,
                                 hipStream_t s);
int64_t launch_chain_fk(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, hipStream_t s);
int64_t launch_fk_fan_chains(const drm_walk *chains, int T, const float *q, int64_t B, float *pos, float *quat, hipStream_t s, bool links = false);

// drm_arm_hand.hip: inverse dynamics of DRM_WALK_ARM_HAND walks whose (P, K, L) is compiled in; rows covered, 0 = not taken
int64_t launch_rnea_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau,
                             hipStream_t s);
bool arm_hand_compiled(const drm_walk *w);
bool crba_arm_hand_applies(const drm_walk *w);
int64_t launch_rnea_backward_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *qdd, const float *gtau,
                                      int64_t B, int flags, uint64_t param_mask, float *gq, float *gqd, float *gqdd, float *partials,
                                      int &partial_rows, hipStream_t s);
int64_t launch_crba_arm_hand(const drm_walk *w, const float *q, int64_t B, float *H, hipStream_t s);
int64_t launch_forward_dynamics_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int flags,
                                         float *qdd, hipStream_t s);

template <class K>
static int ensure_lds(K kernel, size_t bytes) {
    if (bytes > (size_t)64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    return DRM_OK;
}

} // namespace drm
