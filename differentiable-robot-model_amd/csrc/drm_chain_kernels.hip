// drm_chain_kernels.hip — straight-line, register-resident kernels for ANY serial chain of up to 16 ops (round 3; walk capacities 4 / 8 / 12 / 16).
//
// The 7-DoF arm kernels (drm_arm_kernels.hip) are compiled for one shape: ops 0..6 moving and driving DoF columns 0..6.
// Every other chain — the arm of a Panda WITH its gripper up to a fingertip (11 ops, 9 DoF columns), a Jaco finger (13),
// an iiwa7 + Allegro fingertip (14), a finger of a hand (6) — used to take the loop-structured kernels (drm_tree.hpp):
// control words decoded per op, q staged through LDS, two to three times the instructions per link.  The kernels here
// keep the arm kernels' design (constant rows staged once per wave in LDS and read back as broadcast ds_read_b128s,
// packed-FP32 pose pairs in registers, per-lane q loads, write-through / non-temporal 16-byte stores) and take the
// shape of the chain from the walk's control words instead of from template parameters:
//   * W0[k] (include/drm_hip.h) of the CAP ops arrives with ONE wide scalar load; the DoF column of op k is a
//     wave-uniform SGPR value: q is read per lane at a uniform offset, Jacobian columns land at a uniform LDS offset;
//   * a fixed op is a joint at angle 0 (cos = 1, sin = 0 make J == F exactly): the walk is one basic block for every chain;
//   * identity padding ops (F = I, t = 0) compose exactly and are simply walked.
// A prismatic joint (W0 bit 26) keeps J = F and slides its origin along +z by q; its Jacobian column is (z, 0).  A ragged
// tail of B % 64 rows goes through the loop-structured kernel, as for the arm kernels.
//
//   chain_fk_kernel<CAP, JAC, NT>   K1 / K2 of one chain: pos, quat (+ lin_jac, ang_jac [B, 3, n]; columns of DoFs off the
//                                   chain are zero, robot_model.py:645-648)             one wavefront per 64-sample tile
//   fk_fan_chain_kernel<CAP>        K1 of T <= 4 disjoint chains (the fingertips of a hand, BASELINE configuration 4):
//                                   wavefront t of a block walks chain t, the [64, 3T] / [64, 4T] tiles are assembled in
//                                   LDS and leave as linear 16-byte stores
#include "drm_common.hpp"
#include "drm_sample.hpp"

namespace drm {

// DoF column of every op of a serial chain (-1: fixed joint or padding), from the walk's W0 words (wave-uniform)
//   pris: bit k set <=> op k is a prismatic joint (slides along +z of its frame by q instead of turning about it)
// The DoF column (+1; 0 = the op does not move) of the first 16 ops and the prismatic bits, from drm_walk.chain_dof1 /
// chain_prismatic: launch arguments, so the loads of a lane's joint angles do not wait for a read of the control words.
struct ChainDofs {
    uint32_t w[4];
    uint32_t pris;
};
static inline ChainDofs chain_dofs_of(const drm_walk *w) {
    ChainDofs c;
    for (int i = 0; i < 4; ++i)
        c.w[i] = (uint32_t)w->chain_dof1[4 * i] | ((uint32_t)w->chain_dof1[4 * i + 1] << 8) | ((uint32_t)w->chain_dof1[4 * i + 2] << 16) |
                 ((uint32_t)w->chain_dof1[4 * i + 3] << 24);
    c.pris = w->chain_prismatic;
    return c;
}

template <int CAP>
__device__ __forceinline__ void chain_dofs(const ChainDofs &c, int (&dof)[CAP], unsigned &pris) {
#pragma unroll
    for (int k = 0; k < CAP; ++k) dof[k] = (int)((c.w[k >> 2] >> (8 * (k & 3))) & 0xffu) - 1;
    pris = c.pris;
}

// cos / sin of the joint angle of every op, two ops per packed evaluation.  Ops that do not move carry the angle 0 and are
// evaluated like the others (cos 0 = 1, sin 0 = 0 exactly): skipping them pair by pair behind wave-uniform branches was
// tried and costs more than it saves — the merges after every branch triple the register count of the 16-op kernel
// USED: ops USED .. CAP-1 are known to be padding (the launcher picks the instantiation from n_ops)
// and are neither evaluated nor walked
// TRIG < USED: the ops TRIG .. USED-1 are known NOT to move (the fixed tail of a chain: a fingertip frame) — no evaluation for them
template <int CAP, int USED, int TRIG = USED>
__device__ __forceinline__ void chain_trig_all(const float (&q)[CAP], float (&cs)[CAP], float (&sn)[CAP]) {
    static_assert(USED <= CAP && TRIG <= USED, "ops USED .. CAP-1 are padding");
#pragma unroll
    for (int k = TRIG; k < USED; ++k) { cs[k] = 1.0f; sn[k] = 0.0f; }
    bool big = false;
#pragma unroll
    for (int k = 0; k < TRIG; ++k) big = big || !(fabsf(q[k]) <= SINCOS_PAIR_MAX_ARG);
    if (DRM_WAVE_ANY(big)) { // rare; wave-uniform
#pragma unroll
        for (int k = 0; k < TRIG; ++k) sincos_f(q[k], sn[k], cs[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < TRIG; k += 2) {
        f2 s2, c2;
        sincos_pair(f2_make(q[k], q[k + 1 < TRIG ? k + 1 : k]), s2, c2);
        sn[k] = s2[0]; cs[k] = c2[0];
        if (k + 1 < TRIG) { sn[k + 1] = s2[1]; cs[k + 1] = c2[1]; }
    }
}

// this lane's joint angles, one dword load per op at a wave-uniform column (ops that do not move: column 0, value dropped).
// Addressing: uniform base (tile + column, an SGPR pair) + one 32-bit per-lane byte offset shared by all the loads
template <int CAP, int USED>
__device__ __forceinline__ void chain_load_q(const float *__restrict__ qtile, unsigned row_off, const int (&dof)[CAP], float (&qv)[CAP]) {
    const char *qb = reinterpret_cast<const char *>(qtile);
#pragma unroll
    for (int k = 0; k < USED; ++k) {
        const float v = *reinterpret_cast<const float *>(qb + (dof[k] < 0 ? 0 : dof[k]) * 4 + row_off);
        qv[k] = dof[k] < 0 ? 0.0f : v;
    }
}

template <int CAP>
__device__ __forceinline__ void chain_stage_table(const float *__restrict__ ops_f, float *lc, unsigned lane) {
    constexpr unsigned N4 = CAP * DRM_OPF_STRIDE / 4, IT = (N4 + WAVE - 1) / WAVE;
    float4 v[IT];
#pragma unroll
    for (unsigned it = 0; it < IT; ++it) {
        const unsigned i = lane + WAVE * it;
        v[it] = reinterpret_cast<const float4 *>(ops_f)[i < N4 ? i : N4 - 1];
    }
#pragma unroll
    for (unsigned it = 0; it < IT; ++it) {
        const unsigned i = lane + WAVE * it;
        if ((it + 1) * WAVE <= N4 || i < N4) reinterpret_cast<float4 *>(lc)[i] = v[it];
    }
}

// the chain itself: pose of the last op; frame(k, B) is handed the (z_k, p_k) pairs of every op's frame as soon as they exist
// A prismatic op (wave-uniform bit of `pris`) keeps J = F and slides: t += F e_z q (drm_tree.hpp joint_transform).
template <int CAP, int USED, bool FENCE, class FRAME, int TRIG = USED>
__device__ __forceinline__ void chain_walk(const float *lc, const float (&q)[CAP], const float (&cs)[CAP], const float (&sn)[CAP],
                                           unsigned pris, PoseP &ee, FRAME frame) {
#pragma unroll
    for (int k = 0; k < USED; ++k) {
        // a scheduling barrier per op: without it the compiler hoists the constant reads of all later ops to the top of the walk
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
        OpPairs o = load_pairs(lc + k * DRM_OPF_STRIDE);
        f2 J01[3];
        if (k >= TRIG) { // the fixed tail: J = F
#pragma unroll
            for (int i = 0; i < 3; ++i) J01[i] = o.f01[i];
        } else if ((pris >> k) & 1u) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { J01[i] = o.f01[i]; o.f2t[i][1] += o.f2t[i][0] * q[k]; }
        } else {
            joint_pairs(o, cs[k], sn[k], J01);
        }
        if (k == 0) compose_pairs_root(J01, o, ee);
        else compose_pairs(ee, J01, o, ee);
        frame(k, ee.B);
    }
}
// (a prismatic joint's sincos is never used; its angle is zeroed before the evaluation so that it cannot trip the large-argument path)

// LDS (dynamic): [ table : CAP x 32 ][ pos : 64 x 3 ][ Jacobian staging : 64 x (3n | 1), ang_jac first, then lin_jac ]
template <int CAP, int USED, bool JAC, bool NT>
__global__ void __launch_bounds__(WAVE)
    chain_fk_kernel(const float *__restrict__ ops_f, ChainDofs cd, const float *__restrict__ q, int n,
                    int target_perm, uint64_t dof_mask, float *__restrict__ pos, float *__restrict__ quat, float *__restrict__ lin,
                    float *__restrict__ ang, uint32_t magic_j) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, P_FLOATS = WAVE * 3;
    const unsigned lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * WAVE;
    float *lc = smem, *lp = smem + C_FLOATS, *lj = lp + P_FLOATS;

    int dof[CAP];
    unsigned pris;
    chain_dofs<CAP>(cd, dof, pris);
    chain_stage_table<CAP>(ops_f, lc, lane);
    float qv[CAP];
    chain_load_q<CAP, USED>(q + b0 * n, lane * (unsigned)n * 4u, dof, qv);
    wave_lds_sync();
    float cs[CAP], sn[CAP], qa[CAP];
#pragma unroll
    for (int k = 0; k < USED; ++k) qa[k] = ((pris >> k) & 1u) ? 0.0f : qv[k];
    chain_trig_all<CAP, USED>(qa, cs, sn);

    // Jacobian (robot_model.py:651-665): column `dof` of ang_jac is z_k, of lin_jac z_k x (p_e - p_k) = z_k x p_e - z_k x p_k.
    // z_k goes into the staging tile the moment frame k exists (it IS the ang_jac column); what stays in registers per op is
    // z_k x p_k, three floats instead of the six of (z_k, p_k) — the difference between two and three waves per SIMD at 10 ops.
    const int S = 3 * n, Sj = pad_odd(S);
    float *row = lj + lane * Sj;
    float zxp[JAC ? USED : 1][3];
    if constexpr (JAC) {
        const uint64_t all = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
        if ((dof_mask & all) != all) { // columns of DoFs off the chain are zero in both Jacobians (wave-uniform loop)
            for (int d = 0; d < n; ++d)
                if (!((dof_mask >> d) & 1ull)) { row[d] = 0.0f; row[n + d] = 0.0f; row[2 * n + d] = 0.0f; }
        }
    }
    PoseP ee;
    // (a prismatic joint's column is (lin, ang) = (z_k, 0): z_k waits in the registers of z_k x p_k)
    chain_walk<CAP, USED, (CAP > 8 || JAC)>(lc, qv, cs, sn, pris, ee, [&](int k, const f2 (&B)[3]) {
        if constexpr (JAC) {
            const float z[3] = {B[0][0], B[1][0], B[2][0]}, p[3] = {B[0][1], B[1][1], B[2][1]};
            if (dof[k] >= 0) {
                float *c = row + dof[k];
                if ((pris >> k) & 1u) {
                    zxp[k][0] = z[0]; zxp[k][1] = z[1]; zxp[k][2] = z[2];
                    c[0] = 0.0f; c[n] = 0.0f; c[2 * n] = 0.0f;
                } else {
                    cross3(z, p, zxp[k]);
                    c[0] = z[0]; c[n] = z[1]; c[2 * n] = z[2];
                }
            }
        }
    });
    const float pe[3] = {ee.B[0][1], ee.B[1][1], ee.B[2][1]};

    if constexpr (JAC) {
        wave_lds_sync();
        tile_store<0, NT>(ang + b0 * S, WAVE, S, magic_j, lj, lane, (S & 1) != 0, true);
        wave_lds_sync(); // the tile has left; each lane now turns the z_k of ITS row into the lin_jac columns, in place
#pragma unroll
        for (int k = 0; k < USED; ++k)
            if (dof[k] >= 0) {
                float *c = row + dof[k];
                if ((pris >> k) & 1u) {
                    c[0] = zxp[k][0]; c[n] = zxp[k][1]; c[2 * n] = zxp[k][2];
                } else {
                    const float z[3] = {c[0], c[n], c[2 * n]};
                    float cr[3];
                    cross3(z, pe, cr);
                    c[0] = cr[0] - zxp[k][0]; c[n] = cr[1] - zxp[k][1]; c[2 * n] = cr[2] - zxp[k][2];
                }
            }
    }
    lp[lane * 3 + 0] = pe[0];
    lp[lane * 3 + 1] = pe[1];
    lp[lane * 3 + 2] = pe[2];
    wave_lds_sync();
    if constexpr (JAC) tile_store<0, NT>(lin + b0 * 3 * n, WAVE, 3 * n, magic_j, lj, lane, ((3 * n) & 1) != 0, true);
    tile_store<3, NT>(pos + b0 * 3, WAVE, 3, 0u, lp, lane, true);
    {
        Pose E;
        float qt[4];
        pose_from_pairs(ee, E);
        unpermute(target_perm, E.R); // the target is the last REAL op: undo its column permutation before the quaternion
        quat_xyzw(E.R, qt);
        store16_wt<NT>(quat + (b0 + lane) * 4, make_float4(qt[0], qt[1], qt[2], qt[3]));
    }
}

// One 32-byte record per chain, so that a wavefront picks its chain up with ONE scalar load at a wave-uniform offset (the launch's
// other arguments sit in the preloaded kernel-argument registers ahead of the records): table pointer, DoF columns, prismatic
// bits and target permutation arrive together, and the loads of the table and of the joint angles go out right behind them.
struct FanChain {
    const float *ops_f;
    ChainDofs dofs; // 20 bytes
    int32_t perm;
};
static_assert(sizeof(FanChain) == 32, "one s_load_dwordx8 per wavefront");
struct FanChains {
    FanChain c[4];
};

// LDS (static): four chain tables, then the block's [64, 3T] position and [64, 4T] quaternion tiles (linear images)
// LINKS: link-major outputs, pos [T, B, 3] / quat [T, B, 4] (drm_fk_fanout_links) — a wavefront writes ITS chain's 64 poses as
// contiguous runs (the quaternion a 16-byte store per lane straight from registers, the position through its own 64 x 3 stage):
// no tile shared by the block, no block barrier, no second pass over LDS.
template <int CAP, int USED, int TRIG = USED, bool LINKS = false>
__global__ void __launch_bounds__(WAVE * 4)
    fk_fan_chain_kernel(const float *__restrict__ q, float *__restrict__ pos, float *__restrict__ quat, int T, int n, int64_t B,
                        FanChains tab) {
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE;
    __shared__ __attribute__((aligned(16))) float smem[4 * C_FLOATS + WAVE * 12 + WAVE * 16];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int64_t b0 = (int64_t)blockIdx.x * WAVE;
    float *lc = smem + wave * C_FLOATS, *lp = smem + 4 * C_FLOATS, *lr = lp + WAVE * 12;
        const FanChain mine = tab.c[wave]; // (wave-uniform index into the kernel arguments: one scalar load)
    const float *ops_f = mine.ops_f;
    const ChainDofs &cd = mine.dofs;
    const int perm = mine.perm;

    DRM_STAMP(0);
    int dof[CAP];
    unsigned pris;
    chain_dofs<CAP>(cd, dof, pris);
    chain_stage_table<CAP>(ops_f, lc, lane);
    float qv[CAP];
#ifndef DRM_NO_FAN_Q4
    // a finger whose TRIG == 4 moving ops drive four consecutive DoF columns starting at a multiple of four (rows of n % 4 == 0
    // floats; the Allegro hand's): ONE 16-byte load per lane instead of four dword loads that walk the same cache lines — with 16
    // waves per CU behind one L1 the four passes over the tile's 32 lines were the longest phase of the launch ("inputs landed"
    // 1.24 us into a wave's life at 65 536 rows against 0.6 us for a lone wave, tools/timeline.py): 4.74 -> 4.02 us
    bool q4 = (TRIG == 4) && !(n & 3) && dof[0] >= 0 && !(dof[0] & 3);
#pragma unroll
    for (int k = 1; k < (TRIG == 4 ? 4 : 1); ++k) q4 = q4 && dof[k] == dof[0] + k;
    if (q4) {
        const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(q + b0 * n) + dof[0] * 4 + lane * (unsigned)n * 4u);
        qv[0] = v.x; qv[1] = v.y; qv[2] = v.z; qv[3] = v.w;
#pragma unroll
        for (int k = 4; k < CAP; ++k) qv[k] = 0.0f;
    } else
#endif
    chain_load_q<CAP, USED>(q + b0 * n, lane * (unsigned)n * 4u, dof, qv);
    wave_lds_sync();
    DRM_STAMP_DRAINED(1);
    float cs[CAP], sn[CAP], qa[CAP];
#pragma unroll
    for (int k = 0; k < USED; ++k) qa[k] = ((pris >> k) & 1u) ? 0.0f : qv[k];
    chain_trig_all<CAP, USED, TRIG>(qa, cs, sn);
    PoseP ee;
    auto none = [](int, const f2 (&)[3]) {};
    chain_walk<CAP, USED, false, decltype(none), TRIG>(lc, qv, cs, sn, pris, ee, none);
    DRM_STAMP(3);
    if constexpr (LINKS) {
        // (the launcher takes this form only for 16-byte aligned outputs and B a multiple of 4: every link's arrays are aligned)
        float *st = lp + wave * (WAVE * 3); // 4 x 192 floats of the 28 x 64 the tiles of the other form take
        st[lane * 3 + 0] = ee.B[0][1]; st[lane * 3 + 1] = ee.B[1][1]; st[lane * 3 + 2] = ee.B[2][1];
        wave_lds_sync();
        if (lane < 48u) store16_wt(reinterpret_cast<float4 *>(pos + ((int64_t)wave * B + b0) * 3) + lane, reinterpret_cast<const float4 *>(st)[lane]);
        Pose E;
        float qt[4];
        pose_from_pairs(ee, E);
        unpermute(perm, E.R);
        quat_xyzw(E.R, qt);
        store16_wt(reinterpret_cast<float4 *>(quat + ((int64_t)wave * B + b0) * 4) + lane, make_float4(qt[0], qt[1], qt[2], qt[3]));
        DRM_STAMP(4);
        DRM_STAMP_DRAINED(5);
        return;
    }
    {
        float *p = lp + (lane * T + wave) * 3;
        p[0] = ee.B[0][1]; p[1] = ee.B[1][1]; p[2] = ee.B[2][1];
        Pose E;
        float qt[4];
        pose_from_pairs(ee, E);
        unpermute(perm, E.R);
        quat_xyzw(E.R, qt);
        reinterpret_cast<float4 *>(lr)[lane * T + wave] = make_float4(qt[0], qt[1], qt[2], qt[3]);
    }
    __syncthreads();
    // [64, 3T] and [64, 4T] row-major tiles: linear in LDS and in HBM, 16 bytes per thread and round
    const unsigned tid = threadIdx.x, nth = WAVE * (unsigned)T;
    float4 *gp = reinterpret_cast<float4 *>(pos + b0 * 3 * T), *gr = reinterpret_cast<float4 *>(quat + b0 * 4 * T);
    for (unsigned i = tid; i < (unsigned)(WAVE * 3 * T / 4); i += nth) store16_wt(gp + i, reinterpret_cast<const float4 *>(lp)[i]);
    for (unsigned i = tid; i < (unsigned)(WAVE * T); i += nth) store16_wt(gr + i, reinterpret_cast<const float4 *>(lr)[i]);
}

// The same fan-out with TWO SAMPLES PER LANE (rows l and l + 64 of a 128-row tile): every op of the pose chain is a full packed
// instruction (drm_sample.hpp fk_chain2_trig, the chain of the fused FK + RNEA kernel), so a block of T wavefronts covers 128
// samples for ~1.1x the instructions the one-sample form spends on 64.  Revolute and fixed ops only (DRM_WALK_NO_PRISMATIC);
// a fixed op is a joint at angle 0.  Launches of at least DRM_FAN2_MIN_TILES pairs of tiles (not BASELINE configuration 4's 65 536:
// see the measurement at the macro).
// LDS (static): four chain tables, then the block's [128, 3T] position and [128, 4T] quaternion tiles
constexpr int FAN2_TILE = 2 * WAVE;
template <int CAP, int USED>
__global__ void __launch_bounds__(WAVE * 4)
    fk_fan_chain2_kernel(const float *__restrict__ q, float *__restrict__ pos, float *__restrict__ quat, int T, int n, FanChains tab) {
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE;
    __shared__ __attribute__((aligned(16))) float smem[4 * C_FLOATS + FAN2_TILE * 12 + FAN2_TILE * 16];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int64_t b0 = (int64_t)blockIdx.x * FAN2_TILE;
    float *lc = smem + wave * C_FLOATS, *lp = smem + 4 * C_FLOATS, *lr = lp + FAN2_TILE * 12;
    const FanChain mine = tab.c[wave]; // (wave-uniform index into the kernel arguments: one scalar load)
    const float *ops_f = mine.ops_f;
    const ChainDofs &cd = mine.dofs;
    const int perm = mine.perm;
    int dof[CAP];
    unsigned pris;
    chain_dofs<CAP>(cd, dof, pris);
    chain_stage_table<CAP>(ops_f, lc, lane);
    // this lane's two rows, one dword load per op and row at a wave-uniform column (ops that do not move: column 0, value dropped)
    f2 qv[USED];
    {
        const char *qa = reinterpret_cast<const char *>(q + b0 * n), *qb = reinterpret_cast<const char *>(q + (b0 + WAVE) * n);
        const unsigned row_off = lane * (unsigned)n * 4u;
#pragma unroll
        for (int k = 0; k < USED; ++k) {
            const int c = (dof[k] < 0 ? 0 : dof[k]) * 4;
            const float a = *reinterpret_cast<const float *>(qa + c + row_off), b = *reinterpret_cast<const float *>(qb + c + row_off);
            qv[k] = dof[k] < 0 ? f2_bcast(0.0f) : f2_make(a, b);
        }
    }
    wave_lds_sync();
    f2 cs[USED], sn[USED];
    chain_trig2<USED>(qv, cs, sn);
    Pose2 ee;
    fk_chain2_trig<USED, USED>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, cs, sn, ee);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned r = lane + (unsigned)h * WAVE;
        float *p = lp + (r * T + wave) * 3;
        p[0] = ee.p[0][h]; p[1] = ee.p[1][h]; p[2] = ee.p[2][h];
        float R[9], qt[4];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = ee.R[i][h];
        unpermute(perm, R);
        quat_xyzw(R, qt);
        reinterpret_cast<float4 *>(lr)[r * T + wave] = make_float4(qt[0], qt[1], qt[2], qt[3]);
    }
    __syncthreads();
    const unsigned tid = threadIdx.x, nth = WAVE * (unsigned)T;
    float4 *gp = reinterpret_cast<float4 *>(pos + b0 * 3 * T), *gr = reinterpret_cast<float4 *>(quat + b0 * 4 * T);
    for (unsigned i = tid; i < (unsigned)(FAN2_TILE * 3 * T / 4); i += nth) store16_wt(gp + i, reinterpret_cast<const float4 *>(lp)[i]);
    for (unsigned i = tid; i < (unsigned)(FAN2_TILE * T); i += nth) store16_wt(gr + i, reinterpret_cast<const float4 *>(lr)[i]);
}
// measured (Allegro, four fingertips; us per launch, one / two samples per lane): 65 536: 6.0 / 6.3, 2^18: 15.5 / 13.8, 2^20: 54.8 / 47.8
// — half as many wavefronts hide less latency until there are enough of them: pairs of tiles from 2^18 samples on
#ifndef DRM_FAN2_MIN_TILES
#define DRM_FAN2_MIN_TILES 2048
#endif

// ---- launchers (called by drm_fk / drm_fk_jacobian / drm_fk_fanout); return the rows they covered (full tiles), 0 = not taken
static bool chain_ok(const drm_walk *w) {
    return (w->shape & DRM_WALK_SERIAL_CHAIN) && (w->shape & DRM_WALK_CHAIN_DOFS) && (w->capacity == 4 || w->capacity == 8 || w->capacity == 12 || w->capacity == 16) &&
           w->n_ops >= 1 &&
           w->n_slots == 0 && (((uintptr_t)w->ops_f) & 15u) == 0 && w->target_perm >= 0 && w->target_perm <= 5;
}

template <int CAP, int USED, bool JAC>
static void launch_chain_used(const drm_walk *w, const float *q, int n_tiles, float *pos, float *quat, float *lin, float *ang, hipStream_t s) {
    const int n = w->n_dofs;
    const size_t lds = sizeof(float) * (size_t)(CAP * DRM_OPF_STRIDE + WAVE * 3 + (JAC ? round4(WAVE * pad_odd(3 * n)) : 0));
    const bool nt = stream_past_llc((int64_t)n_tiles * WAVE * 4 * (7 + (JAC ? 6 * n : 0)));
    if (nt) {
        ensure_lds((chain_fk_kernel<CAP, USED, JAC, true>), lds);
        hipLaunchKernelGGL((chain_fk_kernel<CAP, USED, JAC, true>), dim3((unsigned)n_tiles), dim3(WAVE), lds, s, w->ops_f, chain_dofs_of(w), q,
                           n, (int)w->target_perm, w->dof_mask, pos, quat, lin, ang, div_magic(3 * n));
    } else {
        ensure_lds((chain_fk_kernel<CAP, USED, JAC, false>), lds);
        hipLaunchKernelGGL((chain_fk_kernel<CAP, USED, JAC, false>), dim3((unsigned)n_tiles), dim3(WAVE), lds, s, w->ops_f, chain_dofs_of(w), q,
                           n, (int)w->target_perm, w->dof_mask, pos, quat, lin, ang, div_magic(3 * n));
    }
}
// the instantiation whose walked ops (USED = CAP - 2 or CAP) cover the chain's n_ops
template <int CAP, bool JAC>
static void launch_chain(const drm_walk *w, const float *q, int n_tiles, float *pos, float *quat, float *lin, float *ang, hipStream_t s) {
    if (CAP > 4 && w->n_ops <= CAP - 2) launch_chain_used<CAP, (CAP > 4 ? CAP - 2 : CAP), JAC>(w, q, n_tiles, pos, quat, lin, ang, s);
    else launch_chain_used<CAP, CAP, JAC>(w, q, n_tiles, pos, quat, lin, ang, s);
}

int64_t launch_chain_fk_jacobian(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin, float *ang,
                                 hipStream_t s) {
#ifdef DRM_NO_CHAIN_KERNEL
    return 0;
#else
    const uint32_t al = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT) | al16(lin, AL_LIN) | al16(ang, AL_ANG);
    if (!chain_ok(w) || !pos || !quat || al != (AL_Q | AL_POS | AL_QUAT | AL_LIN | AL_ANG) || B < WAVE || B / WAVE >= 0x7fffffffLL ||
        w->n_dofs > 32)
        return 0;
    const int n_tiles = (int)(B / WAVE);
    if (w->capacity == 4) launch_chain<4, true>(w, q, n_tiles, pos, quat, lin, ang, s);
    else if (w->capacity == 8) launch_chain<8, true>(w, q, n_tiles, pos, quat, lin, ang, s);
    else if (w->capacity == 12) launch_chain<12, true>(w, q, n_tiles, pos, quat, lin, ang, s);
    else launch_chain<16, true>(w, q, n_tiles, pos, quat, lin, ang, s);
    return (int64_t)n_tiles * WAVE;
#endif
}

int64_t launch_chain_fk(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, hipStream_t s) {
#ifdef DRM_NO_CHAIN_KERNEL
    return 0;
#else
    const uint32_t al = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
    if (!chain_ok(w) || al != (AL_Q | AL_POS | AL_QUAT) || B < WAVE || B / WAVE >= 0x7fffffffLL) return 0;
    const int n_tiles = (int)(B / WAVE);
    if (w->capacity == 4) launch_chain<4, false>(w, q, n_tiles, pos, quat, nullptr, nullptr, s);
    else if (w->capacity == 8) launch_chain<8, false>(w, q, n_tiles, pos, quat, nullptr, nullptr, s);
    else if (w->capacity == 12) launch_chain<12, false>(w, q, n_tiles, pos, quat, nullptr, nullptr, s);
    else launch_chain<16, false>(w, q, n_tiles, pos, quat, nullptr, nullptr, s);
    return (int64_t)n_tiles * WAVE;
#endif
}

int64_t launch_fk_fan_chains(const drm_walk *chains, int T, const float *q, int64_t B, float *pos, float *quat, hipStream_t s, bool links) {
#ifdef DRM_NO_CHAIN_KERNEL
    return 0;
#else
    const uint32_t al = al16(q, AL_Q) | al16(pos, AL_POS) | al16(quat, AL_QUAT);
    if (T < 2 || T > 4 || al != (AL_Q | AL_POS | AL_QUAT) || B < WAVE || B / WAVE >= 0x7fffffffLL) return 0;
    FanChains tab;
    int longest = 0;
    const int cap = chains[0].capacity;
    bool revolute = true;
    for (int t = 0; t < 4; ++t) {
        const drm_walk *w = chains + (t < T ? t : 0);
        if (!chain_ok(w) || (cap != 4 && cap != 8) || w->capacity != cap) return 0;
        tab.c[t].ops_f = w->ops_f; tab.c[t].dofs = chain_dofs_of(w); tab.c[t].perm = w->target_perm;
        if (w->n_ops > longest) longest = w->n_ops;
        revolute = revolute && (w->shape & DRM_WALK_NO_PRISMATIC);
    }
    const int n = (int)chains[0].n_dofs;
    int n_tiles = (int)(B / WAVE);
    int64_t done = 0;
    // the instantiation that walks exactly the longest chain's ops (shorter chains of the same launch walk identity padding)
    if (links && (B & 3)) return 0; // (a link's [B, 3] array starts 12 B t bytes in)
    const int64_t Btot = B;
#ifndef DRM_NO_FAN2_KERNEL
    if (!links && revolute && n_tiles / 2 >= DRM_FAN2_MIN_TILES) { // pairs of tiles: two samples per lane
        const int n2 = n_tiles / 2;
#define FAN2(C, U) hipLaunchKernelGGL((fk_fan_chain2_kernel<C, U>), dim3((unsigned)n2), dim3(WAVE * T), 0, s, q, pos, quat, T, n, tab)
        if (cap == 4) FAN2(4, 4);
        else if (longest <= 5) FAN2(8, 5);
        else if (longest == 6) FAN2(8, 6);
        else if (longest == 7) FAN2(8, 7);
        else FAN2(8, 8);
#undef FAN2
        done = (int64_t)n2 * FAN2_TILE;
        n_tiles -= 2 * n2;
        if (n_tiles == 0) return done;
        q += done * n; pos += done * 3 * T; quat += done * 4 * T;
    }
#endif
    // ... and, where the last walked op of every chain is fixed (a fingertip frame; identity padding of a shorter chain), the form
    // that spends no sincos and no joint rotation on it
#define FAN_(C, U, TR, LK) hipLaunchKernelGGL((fk_fan_chain_kernel<C, U, TR, LK>), dim3((unsigned)n_tiles), dim3(WAVE * T), 0, s, q, pos, quat, T, n, Btot, tab)
#define FAN(C, U) do { if (links) { if (tail_fixed(U)) FAN_(C, U, U - 1, true); else FAN_(C, U, U, true); }                      \
                       else { if (tail_fixed(U)) FAN_(C, U, U - 1, false); else FAN_(C, U, U, false); } } while (0)
    auto tail_fixed = [&](int used) {
#ifdef DRM_NO_FAN_TAIL
        return false;
#endif
        for (int t = 0; t < T; ++t)
            if (chains[t].chain_dof1[used - 1]) return false;
        return true;
    };
    if (cap == 4) FAN(4, 4);
    else if (longest <= 5) FAN(8, 5);
    else if (longest == 6) FAN(8, 6);
    else if (longest == 7) FAN(8, 7);
    else FAN(8, 8);
#undef FAN
#undef FAN_
    return done + (int64_t)n_tiles * WAVE;
#endif
}

} // namespace drm
DRM_TL_READER(chain)
