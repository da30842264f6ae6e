// drm_arm_stream.hpp — the streaming (persistent, software-pipelined) two-samples-per-lane walk of a serial arm: FK(target) + RNEA,
// or RNEA alone.  Shared by the library's kernel (constant table in LDS: csrc/drm_arm_dynamics.hip arm2_stream_kernel) and by the
// per-robot kernels that specialize.py generates with the robot's constants folded into the instruction stream (the table a
// `constexpr` array in the generated source: multiplications by exact zeros and ones disappear, the other constants become
// literals / SGPR operands, and no table is read from LDS at all).
#pragma once
#include "drm_common.hpp"
#include "drm_sample.hpp"

namespace drm {

constexpr int STREAM_TILE = 2 * WAVE; // rows per wavefront and tile

// one array's rows of a 128-row tile (STREAM_TILE * NJ floats = 3.5 KB for NJ = 7) from global memory into LDS, linear image
template <int NJ>
__device__ __forceinline__ void tile_rows_to_lds(const float *__restrict__ src, float *dst, unsigned lane) {
    constexpr int BYTES = STREAM_TILE * NJ * 4, FULL = BYTES / 1024, REST = BYTES % 1024; // 16 B per lane: 1 KB per instruction
    static_assert(REST % 16 == 0, "whole 16-byte pieces");
    const uint32_t voff = lane * 16u;
    const uint32_t lds = (uint32_t)(uintptr_t)dst; // (the low half of a generic LDS pointer is the LDS address)
    unsigned keep;
#pragma unroll
    for (int i = 0; i < FULL; ++i)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src + i * 256), "s"(lds + i * 1024u) : "memory");
    if constexpr (REST > 0) {
        if (lane < (unsigned)(REST / 16))
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(src + FULL * 256), "s"(lds + FULL * 1024u) : "memory");
    }
}
template <int NJ>
__device__ __forceinline__ void rows_from_lds(const float *st, unsigned lane, f2 (&v)[NJ]) { // rows l and l + 64 of the tile
#pragma unroll
    for (int d = 0; d < NJ; ++d) v[d] = f2_make(st[lane * NJ + d], st[(WAVE + lane) * NJ + d]);
}
// ROWS: rows() is called once per tile and returns row(k) -> const float * (op k's constant row).
// smem: [ tau / pos staging : round4(STREAM_TILE * NJ) ][ next rows : 3 * STREAM_TILE * NJ ] floats of LDS, 16-byte aligned.
//   table(): run once, after the first tile's loads have been issued.
// PUT (ABI 11, drm_fk_rnea_put): every output tile is ALSO stored to the destination sets of `put` at row put->row_offset + b0 — the
// gathered arrays of the other GPUs of the node (peer memory over xGMI), as the tile leaves the wavefront: the one-sided gather of
// a sharded batch rides on the walk instead of following it as a collective.
template <int CAP, int NJ, int LINKS, bool FK, bool PREF, bool PUT = false, class TABLE, class ROWS>
__device__ __forceinline__ void arm2_stream_body(TABLE table, ROWS rows, float *smem, const float *__restrict__ q, const float *__restrict__ qd,
                                                 const float *__restrict__ qdd, int n_tiles, int flags, float *__restrict__ tau,
                                                 float *__restrict__ pos, float *__restrict__ quat, const drm_put *put = nullptr) {
    constexpr int T_FLOATS = round4(STREAM_TILE * NJ), ROWS_F = STREAM_TILE * NJ;
    static_assert(STREAM_TILE * 3 <= T_FLOATS && ROWS_F % 4 == 0, "the position tile fits into the staging area; 16-byte aligned arrays");
    const unsigned lane = threadIdx.x;
    float *lt = smem, *st = lt + T_FLOATS;
    int tile = (int)blockIdx.x;
    if (tile >= n_tiles) return;
    auto stage = [&](int t) {
        const int64_t base = (int64_t)t * ROWS_F;
        tile_rows_to_lds<NJ>(q + base, st, lane);
        tile_rows_to_lds<NJ>(qd + base, st + ROWS_F, lane);
        if (qdd) tile_rows_to_lds<NJ>(qdd + base, st + 2 * ROWS_F, lane);
    };
    f2 qv[NJ], qdv[NJ], qddv[NJ], tv[NJ];
    auto unstage = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the staged rows have landed (LDS-DMA is invisible to the compiler's counters)
        wave_lds_sync();
        rows_from_lds<NJ>(st, lane, qv);
        rows_from_lds<NJ>(st + ROWS_F, lane, qdv);
        rows_from_lds<NJ>(st + 2 * ROWS_F, lane, qddv); // (no branch: without qdd nothing was staged there and the reads are discarded)
#pragma unroll
        for (int d = 0; d < NJ; ++d) qddv[d] = qdd ? qddv[d] : f2_bcast(0.0f);

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // ... and are in registers before the staging area is written again
    };
    stage(tile);
    table(); // (the library's kernel copies its constant table into LDS here, behind the first tile's loads)
    unstage();
#pragma unroll 1
    for (;;) {
        auto row = rows();
        const int next = tile + (int)gridDim.x;
        const bool more = next < n_tiles;
        if (more) stage(next);
        const int64_t b0 = (int64_t)tile * STREAM_TILE;
        f2 cs[NJ], sn[NJ];
        chain_trig2<NJ>(qv, cs, sn);
        if constexpr (FK) { // forward kinematics of the last link (robot_model.py:223-248)
            Pose2 ee;
            fk_chain2_trig<CAP, NJ>(row, cs, sn, ee);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                lt[lane * 3 + c] = ee.p[c][0];
                lt[(WAVE + lane) * 3 + c] = ee.p[c][1];
            }
            wave_lds_sync();
            tile_store<6>(pos + b0 * 3, WAVE, 6, 0u, lt, lane, true); // 128 rows of 3 floats
            if constexpr (PUT) {
                for (int p = 0; p < put->n_peers; ++p)
                    if (put->pos[p]) tile_store<6>(put->pos[p] + (put->row_offset + b0) * 3, WAVE, 6, 0u, lt, lane, true);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float R[9], qt[4];
#pragma unroll
                for (int i = 0; i < 9; ++i) R[i] = ee.R[i][h];
                quat_xyzw(R, qt);
                const float4 qv4 = make_float4(qt[0], qt[1], qt[2], qt[3]);
                store16_wt(quat + (b0 + h * WAVE + lane) * 4, qv4);
                if constexpr (PUT) {
                    for (int p = 0; p < put->n_peers; ++p)
                        if (put->quat[p]) store16_wt(put->quat[p] + (put->row_offset + b0 + h * WAVE + lane) * 4, qv4);
                }
            }
        }
        // inverse dynamics (robot_model.py:305-375); nothing parked: KEEP2 = LINKS - 1
        rnea_chain2_trig<LINKS, NJ, LINKS - 1, PREF>(row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, cs, sn, qdv, qddv, tv,
                                                    [&](int, const Force2 &) {}, [&](int, Force2 &) {});
        // tau is staged FIRST: it pins the sweeps above the wait below.  (With the tile's last use of tv behind a branch or behind the
        // wait, the compiler sinks the whole dynamics arithmetic after it — the table reads cannot follow, so 200 constants sit in
        // registers across the wait, 160 of them spilled, and the wait for the next tile's rows runs BEFORE the sweeps.)
        wave_lds_sync(); // (the position tile has been read out of lt long ago: LDS runs a wave's instructions in order)
#pragma unroll
        for (int d = 0; d < NJ; ++d) {
            lt[lane * NJ + d] = tv[d][0];
            lt[(WAVE + lane) * NJ + d] = tv[d][1];
        }
        unstage(); // the next tile's rows into the (now dead) input registers — after the last tile: stale rows nobody uses
        wave_lds_sync();
        tile_store<2 * NJ>(tau + b0 * NJ, WAVE, 2 * NJ, 0u, lt, lane, true);
        if constexpr (PUT) {
            for (int p = 0; p < put->n_peers; ++p)
                if (put->tau[p]) tile_store<2 * NJ>(put->tau[p] + (put->row_offset + b0) * NJ, WAVE, 2 * NJ, 0u, lt, lane, true);
        }
        if (!more) break;
        tile = next;
    }
}

} // namespace drm
