// drm_arm_kernels.hip — the serial-chain ("arm") kernels of the two hottest calls, K2 FK+Jacobian (the metric kernel)
// and K3 RNEA, in a translation unit of their own: it is the only one compiled with kernel-argument preload
// (-mllvm -amdgpu-kernarg-preload-count, see the Makefile) — these kernels order their arguments for it, use < 170
// registers and spill nothing, whereas the big table-driven kernels gain nothing from it and one of them faulted with it.
// The C ABI entry points (drm_fk_jacobian.hip, drm_fk.hip, drm_rnea.hip) call the launch_* functions below.
#include "drm_common.hpp"
#include "drm_sample.hpp"

namespace drm {

// ---------------------------------------------------------------------------------------------------
// Serial-chain ("arm") specialisation, full tiles only: DRM_WALK_ARM_CHAIN walks (Franka Panda, KUKA iiwa: NJ
// moving joints in DoF order, then fixed links), every tile has 64 rows, every pointer is 16-byte aligned, pos /
// quat wanted, NJ odd.  This is the shape of the metric workload.  Differences from the generic kernel:
//   * the arithmetic runs on packed FP32 pairs (drm_sample.hpp "Packed-FP32 form"): 27 v_pk_* per link instead
//     of 48 scalar VALU ops, two joints per sincos evaluation;
//   * the constant rows of the walk (1 KB) are staged ONCE per wave into LDS by one 16-byte load per lane and
//     the FT blocks are read back as broadcast ds_read_b128s into VGPR pairs (in-order LDS returns let the compiler wait per link, and
//     packed ops take VGPR pairs without constant-bus limits); no scalar loads, no int table;
//   * the first 14 argument dwords are preloaded into SGPRs by the command processor (kernarg preload,
//     HIPFLAGS in the Makefile), so the constant and q-tile loads are issued in the wave's first cycles;
//   * no runtime shape flags: one basic block from the loads to the stores; pos / lin_jac / ang_jac are staged in
//     separate LDS regions, so there is one LDS turnaround before the 16-byte stores go out back to back.
//   * outputs are written THROUGH the L2 (store16_wt, drm_common.hpp) in the order they become final, so the 12.8 MB
//     of a 65 536-sample launch drain while the chain is still being walked instead of after the last wave.
// With one wave per SIMD (batch 65 536) a launch lasts launch floor + load + issue time + what is left of the store
// drain after the last store (tools/ubench/metric_lab.hip), so instruction count and the time of the FIRST store are
// what this kernel minimises.  The launcher sends a ragged tail (B % 64 rows) through the generic kernel.
// WPB = waves per block, a packing choice (waves are independent): single-wave blocks let a CU hold four waves per
// SIMD (a 4-wave block's 54 KB of LDS stops at three) — 3.90 -> 3.85 us at 65 536 samples, 33.3 -> 31.0 us at 2^20;
// the NT form (outputs larger than the Infinity Cache) dispatches fewer, larger blocks.
// ---------------------------------------------------------------------------------------------------
// PRE ("latency form", launches of at most two waves per SIMD — the metric's 65 536 rows are ONE wave per SIMD): the FT
// blocks of all CAP ops are read from LDS into registers in one burst right after the table is staged (24 broadcast
// ds_read_b128 back to back, 96 VGPRs) instead of op by op inside the walk.  A lone wave has nobody to hide its LDS round
// trips behind: op by op, the walk waited ~8 times for a read it had issued a few instructions earlier, and the fixed tail's
// rows could not be hoisted above the ang_jac staging stores (same LDS array: may alias).  Per-wave timeline
// (tools/timeline.py, profiles/r04_timeline.txt): inputs landed -> first store 1.00 -> 0.8 us.  Launches that put more waves
// on a SIMD keep the op-by-op form (106 VGPRs: occupancy is what hides latency there).
template <int CAP, int NJ, bool JAC, int WPB, bool NT, bool PRE = false>
__global__ void __launch_bounds__(WAVE *WPB)
    fk_jacobian_arm_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, int n_tiles,
                           float *__restrict__ pos, float *__restrict__ quat, float *__restrict__ lin,
                           float *__restrict__ ang) {
    // argument order: everything needed to issue the first loads sits in the preloaded dwords.
    // JAC = false is the FK-only form (drm_fk of one target): same chain, no Jacobian columns.
    static_assert(NJ & 1, "odd row widths only (linear LDS image)");
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int SJ = 3 * NJ;
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, P_FLOATS = WAVE * 3, J_FLOATS = round4(WAVE * SJ);
    constexpr int PER_WAVE = C_FLOATS + P_FLOATS + (JAC ? 2 * J_FLOATS : 0);
    __shared__ __attribute__((aligned(16))) float smem[WPB * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = (int)blockIdx.x * WPB + wave;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * PER_WAVE;
    float *lp = lc + C_FLOATS, *ll = lp + P_FLOATS, *la = ll + J_FLOATS;
    const int64_t b0 = (int64_t)tile * WAVE;
    DRM_STAMP(0);

    // the walk's constant rows (1 KB) -> LDS: one 16-byte load per lane, in flight together with this lane's own row
    // of q, which is read straight into registers (28 contiguous bytes per lane, 1 792 per wave: the lines are shared
    // by neighbouring lanes, and skipping the LDS round trip of a q tile is worth 0.09 us of a 3.9 us launch)
    float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];
    float qv[NJ];
    {
        const float *qrow = q + (b0 + lane) * NJ;
#pragma unroll
        for (int d = 0; d < NJ; ++d) qv[d] = qrow[d];
    }
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    wave_lds_sync();
    DRM_STAMP_DRAINED(1); // loads have landed

    float tabr[PRE ? CAP : 1][DRM_OPF_FT_FLOATS];
    if constexpr (PRE) {
#pragma unroll
        for (int k = 0; k < CAP; ++k)
#pragma unroll
            for (int i = 0; i < DRM_OPF_FT_FLOATS; ++i) tabr[k][i] = lc[k * DRM_OPF_STRIDE + i];
        __builtin_amdgcn_sched_barrier(0); // all 24 reads are issued here, ahead of sincos and the walk
    }
    PoseP ee;
    f2 Bk[NJ][3];
    // The outputs leave in the order they become available, so that the store drain (12.8 MB per launch, the
    // longest single item of a one-wave-per-SIMD launch) starts as early as possible: ang_jac needs only the joint
    // axes and goes out while the fixed tail of the chain is still being composed; lin_jac and pos need the end
    // position; the quaternion takes the most arithmetic and goes last.
    fk_chain_pairs<CAP, NJ>([&](int k) -> const float * { if constexpr (PRE) return tabr[k]; else return lc + k * DRM_OPF_STRIDE; }, qv, ee, Bk, [&]() {
        if constexpr (JAC) {
            float *arow = la + lane * SJ;
#pragma unroll
            for (int k = 0; k < NJ; ++k) { // robot_model.py:662
                arow[k] = Bk[k][0][0]; arow[NJ + k] = Bk[k][1][0]; arow[2 * NJ + k] = Bk[k][2][0];
            }
            wave_lds_sync();
            DRM_STAMP(2); // the moving joints are walked: first stores (ang_jac) go out
            tile_store<SJ, NT>(ang + b0 * SJ, WAVE, SJ, 0u, la, lane, true);
        }
    });
    DRM_STAMP(3); // chain done

    const float pe[3] = {ee.B[0][1], ee.B[1][1], ee.B[2][1]};
    if constexpr (JAC) {
        float *lrow = ll + lane * SJ;
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const float z[3] = {Bk[k][0][0], Bk[k][1][0], Bk[k][2][0]};
            const float dp[3] = {pe[0] - Bk[k][0][1], pe[1] - Bk[k][1][1], pe[2] - Bk[k][2][1]};
            float c[3];
            cross3(z, dp, c); // robot_model.py:661
            // keep the columns scalar: packing two joints' cross products costs more register shuffles than it saves
            asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]));
            lrow[k] = c[0]; lrow[NJ + k] = c[1]; lrow[2 * NJ + k] = c[2];
        }
        lp[lane * 3 + 0] = pe[0];
        lp[lane * 3 + 1] = pe[1];
        lp[lane * 3 + 2] = pe[2];
        wave_lds_sync();
        tile_store<SJ, NT>(lin + b0 * SJ, WAVE, SJ, 0u, ll, lane, true);
        tile_store<3, NT>(pos + b0 * 3, WAVE, 3, 0u, lp, lane, true);
    } else {
        lp[lane * 3 + 0] = pe[0];
        lp[lane * 3 + 1] = pe[1];
        lp[lane * 3 + 2] = pe[2];
        wave_lds_sync();
        tile_store<3, NT>(pos + b0 * 3, WAVE, 3, 0u, lp, lane, true);
    }
    // quat [B,4]: one 16-byte store per lane is already coalesced.  The target of an arm-shaped walk that ends in
    // a fixed link (or a z joint) stores its frame un-permuted (DRM_OPI_PERM code 2, checked by the launcher).
    {
        Pose E;
        float qt[4];
        pose_from_pairs(ee, E);
        quat_xyzw(E.R, qt);
        store16_wt<NT>(quat + (b0 + lane) * 4, make_float4(qt[0], qt[1], qt[2], qt[3]));
    }
    DRM_STAMP(4);         // last store issued
    DRM_STAMP_DRAINED(5); // ... and acknowledged
}

// FK of the single target of an arm-shaped walk through the packed chain kernel (called by drm_fk); returns the
// number of rows it covered (full tiles), 0 if the call does not qualify.
int64_t launch_fk_arm(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, hipStream_t s) {
#ifdef DRM_NO_ARM_KERNEL
    return 0;
#else
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
    if (!((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && w->n_dofs == 7 && w->target_perm == 2 &&
          align == (AL_Q | AL_POS | AL_QUAT) && (((uintptr_t)w->ops_f) & 15u) == 0 && B >= WAVE &&
          B / WAVE < 0x7fffffffLL))
        return 0;
    const int n_tiles = (int)(B / WAVE);
    hipLaunchKernelGGL((fk_jacobian_arm_kernel<8, 7, false, 1, false>), dim3((unsigned)n_tiles), dim3(WAVE), 0, s, w->ops_f, q,
                       n_tiles, pos, quat, (float *)nullptr, (float *)nullptr);
    return (int64_t)n_tiles * WAVE;
#endif
}


#ifndef DRM_STREAM_CHUNK_TILES
#define DRM_STREAM_CHUNK_TILES (1 << 15) /* 2^21 rows = 470 MB of outputs per launch */
#endif
constexpr int STREAM_CHUNK_TILES = DRM_STREAM_CHUNK_TILES;
#ifndef DRM_PRE_MAX_TILES
#define DRM_PRE_MAX_TILES 2048 /* 256 CUs x 4 SIMDs x 2 waves */
#endif
constexpr int PRE_MAX_TILES = DRM_PRE_MAX_TILES;
void launch_fk_jacobian_arm(const float *ops_f, const float *q, int n_tiles, float *pos, float *quat, float *lin_jac,
                            float *ang_jac, hipStream_t s) {
    if (stream_past_llc((int64_t)n_tiles * WAVE * 4 * (7 + 6 * 7))) {
        // outputs streamed past the Infinity Cache.  Launches of more than STREAM_CHUNK_TILES tiles go as back-to-back launches of that
        // many over slices of the same arrays (round 6): one launch over 2^24 rows (3.8 GB) streamed at 0.67 of the HBM peak, the same
        // rows as eight launches of 2^21 at 0.78 (profiles/r06_probe_chunks.txt: the wavefronts of a launch then work inside a 0.5 GB
        // window instead of all over the allocation); a kernel boundary costs ~2 us of a ~75 us chunk
        constexpr int WPB = MAX_WAVES_PER_BLOCK;
        for (int t0 = 0; t0 < n_tiles; t0 += STREAM_CHUNK_TILES) {
            const int nt = n_tiles - t0 < STREAM_CHUNK_TILES + STREAM_CHUNK_TILES / 2 ? n_tiles - t0 : STREAM_CHUNK_TILES;   // (no short last chunk)
            const int64_t r0 = (int64_t)t0 * WAVE;
            hipLaunchKernelGGL((fk_jacobian_arm_kernel<8, 7, true, WPB, true>), dim3((unsigned)((nt + WPB - 1) / WPB)), dim3(WAVE * WPB), 0, s,
                               ops_f, q + r0 * 7, nt, pos + r0 * 3, quat + r0 * 4, lin_jac + r0 * 21, ang_jac + r0 * 21);
            if (nt != STREAM_CHUNK_TILES) break;
        }
    } else if (n_tiles <= PRE_MAX_TILES) { // at most two waves per SIMD: the register-resident table
        hipLaunchKernelGGL((fk_jacobian_arm_kernel<8, 7, true, 1, false, true>), dim3((unsigned)n_tiles), dim3(WAVE), 0, s, ops_f, q,
                           n_tiles, pos, quat, lin_jac, ang_jac);
    } else {
        hipLaunchKernelGGL((fk_jacobian_arm_kernel<8, 7, true, 1, false>), dim3((unsigned)n_tiles), dim3(WAVE), 0, s, ops_f, q,
                           n_tiles, pos, quat, lin_jac, ang_jac);
    }
}


} // namespace drm
DRM_TL_READER(arm)
