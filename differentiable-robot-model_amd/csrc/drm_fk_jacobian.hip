// drm_fk_jacobian.hip — K2: FK + geometric Jacobian of one chain (the metric kernel).
//
// Replaces DifferentiableRobotModel.compute_endeffector_jacobian (robot_model.py:626-667) and the
// compute_forward_kinematics call inside it (robot_model.py:641 -> 223-248, 139-195).
//
// Per sample: in  q[n]                                   ( 4 n bytes)
//             out pos[3] quat[4] lin_jac[3,n] ang_jac[3,n] (4 (7 + 6 n) bytes)      n = 7: 224 B
// LDS per wave: [ q tile : 64 (n|1) ][ staging : 64 max(3n|1, 3) ]
#include "drm_common.hpp"
#include "drm_sample.hpp"

namespace drm {

// NDOF > 0 fixes the row width at compile time (tile copies fully unrolled, immediate LDS offsets).
template <int CAP, int NDOF>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    fk_jacobian_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int n_rt, uint64_t dof_mask,
                       const float *__restrict__ q, int64_t B, float *__restrict__ pos, float *__restrict__ quat,
                       float *__restrict__ lin, float *__restrict__ ang, uint32_t magic_q, uint32_t magic_j,
                       int lds_per_wave, uint32_t align, int target_perm) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx cx;
    if (!wave_begin(B, lds_per_wave, smem, cx)) return;
    const unsigned lane = cx.lane;
    const int n = NDOF ? NDOF : n_rt;
    const int Sq = pad_odd(n), Sj = pad_odd(3 * n);
    float *lq = cx.lds;
    float *stage = cx.lds + round4(WAVE * Sq);

    // the wave-uniform DoF column of every op in one wide scalar load; the float table's cache
    // lines are touched now so their misses overlap with the q tile load
    int dof[CAP];
    load_field<CAP>(ops_i, DRM_OPI_DOF, dof);
    WarmRegs<CAP> warm;
    warm_walk_issue<CAP, 1>(ops_f, warm);
    tile_load<NDOF>(q + cx.b0 * n, cx.rows, n, magic_q, lq, lane, cx.full && (n & 1) && (align & AL_Q), cx.full && (align & AL_Q));
    warm_walk_wait(warm);
    wave_lds_sync();

    // lanes beyond the last valid row of a partial tile read stale LDS and compute garbage that is
    // never stored (the arithmetic is branch-free, so garbage is harmless)
    const bool live = (int)lane < cx.rows;
    const float *qrow = lq + lane * Sq;
    auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; }; // zeros, not stale LDS, past a partial tile

    Pose ee;
    float z[CAP][3], pj[CAP][3];
    fk_chain<CAP>(ops_f, dof, qf, ee, z, pj);

    // ---- pos [B,3]: 3 floats per lane -> LDS -> coalesced store ------------------------
    if (pos) {
        stage[lane * 3 + 0] = ee.p[0];
        stage[lane * 3 + 1] = ee.p[1];
        stage[lane * 3 + 2] = ee.p[2];
        wave_lds_sync();
        tile_store<3>(pos + cx.b0 * 3, cx.rows, 3, 0u, stage, lane, cx.full && (align & AL_POS));
        wave_lds_sync();
    }
    // ---- quat [B,4]: one 16-byte store per lane is already coalesced -------------------
    if (quat && live) {
        // the target is the last REAL op; padding ops are identities, so `ee` still carries the
        // target's canonical frame: undo its column permutation before the quaternion
        float Ru[9], qt[4];
#pragma unroll
        for (int i = 0; i < 9; ++i) Ru[i] = ee.R[i];
        unpermute(target_perm, Ru);
        quat_xyzw(Ru, qt);
        float *dst = quat + (cx.b0 + lane) * 4;
        if (align & AL_QUAT) {
            *reinterpret_cast<float4 *>(dst) = make_float4(qt[0], qt[1], qt[2], qt[3]);
        } else {
            dst[0] = qt[0]; dst[1] = qt[1]; dst[2] = qt[2]; dst[3] = qt[3];
        }
    }
    // ---- Jacobians [B,3,n]: column d of op k at row offset r*n + d ----------------------
    float *jrow = stage + lane * Sj;
    const uint64_t all = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    const bool fast_j = cx.full; // 3n is odd iff n is odd; checked per tensor below
    const bool odd_j = (3 * n) & 1;

    // linear part: z_k x (p_e - p_k)   (robot_model.py:661)
    if ((dof_mask & all) != all) {
        for (int d = 0; d < n; ++d)
            if (!((dof_mask >> d) & 1ull)) { jrow[d] = 0.0f; jrow[n + d] = 0.0f; jrow[2 * n + d] = 0.0f; }
    }
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const int d = dof[k];
        if (d >= 0) {
            const float dp[3] = {ee.p[0] - pj[k][0], ee.p[1] - pj[k][1], ee.p[2] - pj[k][2]};
            float c[3];
            cross3(z[k], dp, c);
            jrow[d] = c[0];
            jrow[n + d] = c[1];
            jrow[2 * n + d] = c[2];
        }
    }
    wave_lds_sync();
    tile_store<3 * NDOF>(lin + cx.b0 * 3 * n, cx.rows, 3 * n, magic_j, stage, lane, fast_j && odd_j && (align & AL_LIN),
                         fast_j && (align & AL_LIN));
    wave_lds_sync();

    // angular part: z_k   (robot_model.py:662); the off-chain zeros are still in place
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const int d = dof[k];
        if (d >= 0) {
            jrow[d] = z[k][0];
            jrow[n + d] = z[k][1];
            jrow[2 * n + d] = z[k][2];
        }
    }
    wave_lds_sync();
    tile_store<3 * NDOF>(ang + cx.b0 * 3 * n, cx.rows, 3 * n, magic_j, stage, lane, fast_j && odd_j && (align & AL_ANG),
                         fast_j && (align & AL_ANG));
}

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
// With one wave per SIMD (batch 65 536) a launch lasts launch floor + issue time + store drain (nothing
// overlaps, tools/ubench/io_floor.hip), so instruction count is what this kernel minimises.
// The launcher sends a ragged tail (B % 64 rows) through the generic kernel.
// ---------------------------------------------------------------------------------------------------
// waves per block of the arm kernels: a packing choice (waves are independent); 1, 2 and 4 measure the same
#define DRM_ARM_WPB MAX_WAVES_PER_BLOCK
template <int CAP, int NJ, bool JAC>
__global__ void __launch_bounds__(WAVE *DRM_ARM_WPB)
    fk_jacobian_arm_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, int n_tiles,
                           float *__restrict__ pos, float *__restrict__ quat, float *__restrict__ lin,
                           float *__restrict__ ang) {
    // argument order: everything needed to issue the first loads sits in the preloaded dwords.
    // JAC = false is the FK-only form (drm_fk of one target): same chain, no Jacobian columns.
    static_assert(NJ & 1, "odd row widths only (linear LDS image)");
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int SQ = NJ, SJ = 3 * NJ;
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, Q_FLOATS = round4(WAVE * SQ), P_FLOATS = WAVE * 3,
                  J_FLOATS = round4(WAVE * SJ);
    // pos is staged over the q tile (q lives in registers by then): 51.7 KB per block, three blocks per CU
    static_assert(P_FLOATS <= Q_FLOATS, "pos staging overlays the q tile");
    constexpr int PER_WAVE = C_FLOATS + Q_FLOATS + (JAC ? 2 * J_FLOATS : 0);
    __shared__ __attribute__((aligned(16))) float smem[DRM_ARM_WPB * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = (int)blockIdx.x * DRM_ARM_WPB + wave;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * PER_WAVE;
    float *lq = lc + C_FLOATS, *lp = lq, *ll = lq + Q_FLOATS, *la = ll + J_FLOATS;
    const int64_t b0 = (int64_t)tile * WAVE;

    // the walk's constant rows (1 KB) -> LDS: one 16-byte load per lane, in flight together with the q tile
    float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];
    tile_load<NJ>(q + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    wave_lds_sync();

    float qv[NJ];
#pragma unroll
    for (int d = 0; d < NJ; ++d) qv[d] = lq[lane * SQ + d];
    PoseP ee;
    f2 Bk[NJ][3];
    // The outputs leave in the order they become available, so that the store drain (12.8 MB per launch, the
    // longest single item of a one-wave-per-SIMD launch) starts as early as possible: ang_jac needs only the joint
    // axes and goes out while the fixed tail of the chain is still being composed; lin_jac and pos need the end
    // position; the quaternion takes the most arithmetic and goes last.
    fk_chain_pairs<CAP, NJ>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, qv, ee, Bk, [&]() {
        if constexpr (JAC) {
            float *arow = la + lane * SJ;
#pragma unroll
            for (int k = 0; k < NJ; ++k) { // robot_model.py:662
                arow[k] = Bk[k][0][0]; arow[NJ + k] = Bk[k][1][0]; arow[2 * NJ + k] = Bk[k][2][0];
            }
            wave_lds_sync();
            tile_store<SJ>(ang + b0 * SJ, WAVE, SJ, 0u, la, lane, true);
        }
    });

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
        tile_store<SJ>(lin + b0 * SJ, WAVE, SJ, 0u, ll, lane, true);
        tile_store<3>(pos + b0 * 3, WAVE, 3, 0u, lp, lane, true);
    } else {
        wave_lds_sync(); // every lane has read its q row before pos is staged over the q tile
        lp[lane * 3 + 0] = pe[0];
        lp[lane * 3 + 1] = pe[1];
        lp[lane * 3 + 2] = pe[2];
        wave_lds_sync();
        tile_store<3>(pos + b0 * 3, WAVE, 3, 0u, lp, lane, true);
    }
    // quat [B,4]: one 16-byte store per lane is already coalesced.  The target of an arm-shaped walk that ends in
    // a fixed link (or a z joint) stores its frame un-permuted (DRM_OPI_PERM code 2, checked by the launcher).
    {
        Pose E;
        float qt[4];
        pose_from_pairs(ee, E);
        quat_xyzw(E.R, qt);
        *reinterpret_cast<float4 *>(quat + (b0 + lane) * 4) = make_float4(qt[0], qt[1], qt[2], qt[3]);
    }
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
    hipLaunchKernelGGL((fk_jacobian_arm_kernel<8, 7, false>),
                       dim3((unsigned)((n_tiles + DRM_ARM_WPB - 1) / DRM_ARM_WPB)),
                       dim3(WAVE * DRM_ARM_WPB), 0, s, w->ops_f, q, n_tiles, pos, quat, (float *)nullptr,
                       (float *)nullptr);
    return (int64_t)n_tiles * WAVE;
#endif
}

} // namespace drm

using namespace drm;

extern "C" int drm_fk_jacobian(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin_jac,
                               float *ang_jac, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !lin_jac || !ang_jac) return fail(DRM_ERR_INVALID, "q / lin_jac / ang_jac must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (w->target_perm < 0 || w->target_perm > 5) return fail(DRM_ERR_INVALID, "target_perm must be in 0..5");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs;
    const int Sq = pad_odd(n), Sj = pad_odd(3 * n);
    Geometry g;
    rc = make_geometry(B, round4(WAVE * Sq) + round4(WAVE * (Sj > 3 ? Sj : 3)), g);
    if (rc) return rc;
    const uint32_t align = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT) | al16(lin_jac, AL_LIN) |
                           al16(ang_jac, AL_ANG);
    const uint32_t mq = div_magic(n), mj = div_magic(3 * n);
    hipStream_t s = (hipStream_t)stream;
#define DRM_LAUNCH_FKJ(C, N)                                                                                         \
    {                                                                                                                \
        rc = ensure_lds(fk_jacobian_kernel<C, N>, g.lds_bytes);                                                      \
        if (rc) return rc;                                                                                           \
        hipLaunchKernelGGL((fk_jacobian_kernel<C, N>), g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, n,       \
                           w->dof_mask, q, B, pos, quat, lin_jac, ang_jac, mq, mj, g.lds_per_wave, align,            \
                           (int)w->target_perm);                                                             \
    }
    const uint32_t all_al = AL_Q | AL_POS | AL_QUAT | AL_LIN | AL_ANG;
#ifdef DRM_NO_ARM_KERNEL
    if (false) {
#else
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7 && w->target_perm == 2 && align == all_al &&
        (((uintptr_t)w->ops_f) & 15u) == 0 && B >= WAVE && B / WAVE < 0x7fffffffLL) {
#endif
        // 7-DoF arms (Franka Panda, KUKA iiwa): full tiles through the packed-FP32 chain kernel, ragged tail (if
        // any) through the generic one
        const int n_tiles = (int)(B / WAVE);
        hipLaunchKernelGGL((fk_jacobian_arm_kernel<8, 7, true>),
                           dim3((unsigned)((n_tiles + DRM_ARM_WPB - 1) / DRM_ARM_WPB)),
                           dim3(WAVE * DRM_ARM_WPB), 0, s, w->ops_f, q, n_tiles, pos, quat, lin_jac, ang_jac);
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done < B) {
            rc = launched();
            if (rc) return rc;
            drm_walk generic = *w;
            generic.shape &= ~DRM_WALK_ARM_CHAIN;
            return drm_fk_jacobian(&generic, q + done * n, B - done, pos + done * 3, quat + done * 4,
                                   lin_jac + done * 3 * n, ang_jac + done * 3 * n, stream);
        }
    } else if (w->capacity == 8 && n == 7) {
        DRM_LAUNCH_FKJ(8, 7) // 7-DoF robots, ragged / unaligned / partial-output calls: static tile shapes
    } else {
        DRM_DISPATCH_CAP(w->capacity, DRM_LAUNCH_FKJ(C, 0))
    }
#undef DRM_LAUNCH_FKJ
    return launched();
}
