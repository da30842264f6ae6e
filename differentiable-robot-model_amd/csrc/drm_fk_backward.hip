// drm_fk_backward.hip — K5: reverse-mode derivative of the multi-target FK (positions) with respect to the joint
// angles and the per-link constants R_fixed / trans.
//
// Replaces what torch autograd does for the reference when a loss on compute_forward_kinematics' position is
// back-propagated (robot_model.py:139-195, 223-248 with learnable link parameters robot_model.py:669-713;
// examples/learn_kinematics_of_iiwa.py:25-61, examples/run_kinematic_trajectory_opt.py): there, one backward node
// per tiny torch op of the per-link Python loop; here, two loops per sample over the links of the walk (drm_sample.hpp
// fk_backward_walk: the FK again, then one adjoint sweep), plus a fixed-order reduction of the per-link constant
// gradients over the batch.
//
// Per sample: in q[n], grad_pos[T,3]; out grad_q[n] (optional).                 n = 7, T = 1: 28 + 12 + 28 = 68 B
// Per launch: out grad_ops_f[cap, 32] (dL/dF and dL/dt in the FT block of every op selected by param_mask, zeros
//             elsewhere), reduced DETERMINISTICALLY: each wave strides over tiles and keeps one running sum per
//             (op, field), writes one row of `partials` (the arm kernel: one row per BLOCK, its wavefronts' sums added in
//             wavefront order through LDS), and a second tiny kernel adds the rows in a fixed order.
// LDS per wave: [ q : 64 (n|1) ][ grad_pos : 64 (3T|1) ][ grad_q : 64 (n|1) ][ constant-gradient sums : cap*12 ]
//               [ pose slots : n_slots*12*64 ][ adjoint slots : n_slots*12*64 ][ JAC: 2 x 64 (3n|1) ]
//               [ parked poses : cap*12*64 unless PARK_HBM ]
#include <type_traits>

#include "drm_common.hpp"
#include "drm_sample.hpp"

namespace drm {

constexpr int BWD_FIELDS = 12;          // dF (9) + dt (3) per op
constexpr int POSE_FLOATS = 12;         // a parked world pose: R (9) + p (3)

// One kernel for every walk: the sweeps loop over the n_ops links (drm_sample.hpp fk_backward_walk), so neither the code
// nor the register file grows with the robot; `cap` only fixes the row pitch of grad_ops_f / the partial sums.
// JAC: the walk is the root -> end-effector chain and the loss also depends on the geometric Jacobian
// (glin, gang [B, 3, n] = dL/d lin_jac, dL/d ang_jac; gpos may be NULL).
// PARK_HBM: the per-op poses of the forward sweep are parked in a slice of the caller's scratch buffer instead of LDS.
template <bool JAC, bool PARK_HBM>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    fk_backward_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int cap, int n_ops, int n,
                       int n_slots, int T, const float *__restrict__ q, const float *__restrict__ gpos,
                       const float *__restrict__ grot, const float *__restrict__ glin, const float *__restrict__ gang, int64_t B,
                       float *__restrict__ gq,
                       uint64_t param_mask, float *__restrict__ partials, float *__restrict__ park_hbm, uint32_t magic_q,
                       uint32_t magic_g, uint32_t magic_j, int lds_per_wave, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NV = cap * BWD_FIELDS;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const unsigned lane = threadIdx.x & 63u;
    const int64_t wave_id = (int64_t)blockIdx.x * wpb + wave_in_block;
    const int64_t n_waves = (int64_t)gridDim.x * wpb;
    const int64_t n_tiles = (B + WAVE - 1) / WAVE;

    const int Sq = pad_odd(n), Sg = pad_odd(3 * T), Sj = pad_odd(3 * n);
    float *lq = smem + wave_in_block * lds_per_wave;
    float *lg = lq + round4(WAVE * Sq);
    float *lgq = lg + round4(WAVE * Sg);
    float *lacc = lgq + round4(WAVE * Sq);             // this wave's running sums of the constant gradients
    float *lps = lacc + round4(NV);                     // pose slots    [slot][12][64]
    float *las = lps + n_slots * (12 * WAVE);           // adjoint slots [slot][12][64]
    float *ljl = las + n_slots * (12 * WAVE);           // JAC: dL/d lin_jac, dL/d ang_jac tiles, each 64 (3n|1)
    float *lja = ljl + (JAC ? round4(WAVE * Sj) : 0);
    float *lpk = lja + (JAC ? round4(WAVE * Sj) : 0);   // parked poses [op][12][64] unless PARK_HBM
    float *rec = (PARK_HBM ? park_hbm + wave_id * (int64_t)cap * (POSE_FLOATS * WAVE) : lpk) + lane;
    const int32_t *ctl = ops_i + DRM_OPI_CTRL * cap;

    for (int i = (int)lane; i < NV; i += WAVE) lacc[i] = 0.0f;

    for (int64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
        const int64_t b0 = tile * WAVE;
        const int64_t left = B - b0;
        const int rows = left < WAVE ? (int)left : WAVE;
        const bool full = rows == WAVE;
        const bool live = (int)lane < rows;
        wave_lds_sync(); // the previous tile's LDS reads are done before this tile overwrites
        tile_load<0>(q + b0 * n, rows, n, magic_q, lq, lane, full && (n & 1) && (align & AL_Q), full && (align & AL_Q));
        if (!JAC || gpos)
            tile_load<0>(gpos + b0 * 3 * T, rows, 3 * T, magic_g, lg, lane, full && ((3 * T) & 1) && (align & AL_POS),
                         full && (align & AL_POS));
        if (JAC) {
            tile_load<0>(glin + b0 * 3 * n, rows, 3 * n, magic_j, ljl, lane, full && ((3 * n) & 1) && (align & AL_LIN),
                         full && (align & AL_LIN));
            tile_load<0>(gang + b0 * 3 * n, rows, 3 * n, magic_j, lja, lane, full && ((3 * n) & 1) && (align & AL_ANG),
                         full && (align & AL_ANG));
        }
        for (int s = 0; s < n_slots * 12; ++s) las[s * WAVE + lane] = 0.0f;
        wave_lds_sync();

        const float *qrow = lq + lane * Sq;
        const float *grow = lg + lane * Sg;
        float *gqrow = lgq + lane * Sq;
        if (gq)
            for (int d = 0; d < n; ++d) gqrow[d] = 0.0f; // DoFs that are not on any target's chain
        auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; }; // zeros, not stale LDS, past a partial tile
        auto grad_in = [&](int t, float *G) {
            if (JAC && !gpos) return;
            G[0] += grow[t * 3 + 0]; G[1] += grow[t * 3 + 1]; G[2] += grow[t * 3 + 2];
        };
        const float *jlrow = ljl + lane * Sj, *jarow = lja + lane * Sj;
        auto jac_lin = [&](int d, float *v) { v[0] = jlrow[d]; v[1] = jlrow[n + d]; v[2] = jlrow[2 * n + d]; };
        auto jac_ang = [&](int d, float *v) { v[0] = jarow[d]; v[1] = jarow[n + d]; v[2] = jarow[2 * n + d]; };
        auto put_pose = [&](float *b, const Pose &P) {
#pragma unroll
            for (int i = 0; i < 9; ++i) b[i * WAVE] = P.R[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) b[(9 + i) * WAVE] = P.p[i];
        };
        auto get_pose = [&](const float *b, Pose &P) {
#pragma unroll
            for (int i = 0; i < 9; ++i) P.R[i] = b[i * WAVE];
#pragma unroll
            for (int i = 0; i < 3; ++i) P.p[i] = b[(9 + i) * WAVE];
        };
        auto pose_save = [&](int s, const Pose &P) { put_pose(lps + s * (12 * WAVE) + lane, P); };
        auto pose_load = [&](int s, Pose &P) { get_pose(lps + s * (12 * WAVE) + lane, P); };
        auto park = [&](int k, const Pose &P) { put_pose(rec + k * (POSE_FLOATS * WAVE), P); };
        auto unpark = [&](int k, Pose &P) { get_pose(rec + k * (POSE_FLOATS * WAVE), P); };
        auto adj_add = [&](int s, const Adjoint &A) {
            float *b = las + s * (12 * WAVE) + lane;
#pragma unroll
            for (int i = 0; i < 3; ++i) b[i * WAVE] += A.G[i];
#pragma unroll
            for (int i = 0; i < 9; ++i) b[(3 + i) * WAVE] += A.M[i];
        };
        auto adj_take = [&](int s, Adjoint &A) {
            const float *b = las + s * (12 * WAVE) + lane;
#pragma unroll
            for (int i = 0; i < 3; ++i) A.G[i] += b[i * WAVE];
#pragma unroll
            for (int i = 0; i < 9; ++i) A.M[i] += b[(3 + i) * WAVE];
        };
        auto gq_out = [&](int d, float v) { gqrow[d] = v; };
        auto param_out = [&](int k, const float *dF, const float *dt) { // wave-uniform: only for ops in param_mask
            wave_sums_lane63<BWD_FIELDS>(lane, [&](int j) { return live ? (j < 9 ? dF[j] : dt[j - 9]) : 0.0f; }, // lanes past a partial tile hold garbage
                                         [&](int j, float total) { lacc[k * BWD_FIELDS + j] += total; }); // tiles in this wave's fixed order
        };
        // dL/dR of a target (the loss reads its quaternion): 9 floats per (sample, target), read straight from HBM
        const float *rrow = grot ? grot + (b0 + (live ? lane : 0u)) * (int64_t)(9 * T) : nullptr;
        auto rot_in = [&](int t, float *Rb) -> bool {
            if (!grot) return false;
#pragma unroll
            for (int i = 0; i < 9; ++i) Rb[i] = live ? rrow[t * 9 + i] : 0.0f;
            return true;
        };
        fk_backward_walk<JAC>(ops_f, ctl, n_ops, param_mask, gq != nullptr, qf, grad_in, pose_save, pose_load, adj_add,
                              adj_take, gq_out, param_out, park, unpark, jac_lin, jac_ang, rot_in);
        if (gq) {
            wave_lds_sync();
            tile_store<0>(gq + b0 * n, rows, n, magic_q, lgq, lane, full && (n & 1) && (align & AL_TAU), full && (align & AL_TAU));
        }
    }
    wave_lds_sync();
    // one row of partial sums per wave (zeros for waves that had no tile)
    float *prow = partials + wave_id * NV;
    for (int i = (int)lane; i < NV; i += WAVE) prow[i] = lacc[i];
}

// grad_ops_f[k, FT field j] = column k * 12 + j of the partial rows summed in column_sum's order; the rest of the row zeroed
// (shared by fk_backward_reduce_kernel and the last-block form below)
__device__ __forceinline__ void store_fk_gradient(float *__restrict__ grad_ops_f, int e, float total) {
    const int k = e / BWD_FIELDS, j = e % BWD_FIELDS;
    const int at = j < 9 ? DRM_OPF_FIJ(j / 3, j % 3) : DRM_OPF_TI(j - 9); // dF row-major, then dt
    grad_ops_f[k * DRM_OPF_STRIDE + at] = total;
}

// one float of a row of partial sums, written THROUGH to where every XCD reads it (an agent-scope atomic store: `sc1`)
__device__ __forceinline__ void publish(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The tail of a backward kernel launched with a ticket word (ABI 11): every block publishes its rows of partial sums (agent-scope
// release), takes a ticket, and the block that takes the LAST one (all rows are visible to it: agent-scope acquire) reduces them —
// the same sums in the same order as fk_backward_reduce_kernel, whose REDUCE_WAVES wavefronts this block's waves stand in for, a
// few each — writes grad_ops_f (and the loss), and puts the ticket word back to zero for the next launch.
// Called by ALL threads of the block, outside divergent control flow.  `lds`: >= REDUCE_WAVES * WAVE floats of the block's LDS.
template <int CAP>
__device__ __forceinline__ void last_block_reduce(uint32_t *ticket, const float *partials, int n_rows, int pitch, float *grad_ops_f,
                                                  float *loss, float loss_scale, float *lds) {
    constexpr int NV = CAP * BWD_FIELDS;
    __shared__ uint32_t took;
    // No cache-wide fence on either side (an agent-scope release / acquire writes back / invalidates the whole L2 of the XCD: measured
    // 7.5 -> 43 us for drm_fk_mse at 16 384 rows): the rows were PUBLISHED with agent-scope atomic stores (write-through, `sc1`) by
    // publish_row() and are READ here with agent-scope atomic loads (which do not hit a line of this XCD's L2), so all the protocol
    // needs is program order: a wave's stores are acknowledged (s_waitcnt) before its block's ticket is taken.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) took = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (took != gridDim.x - 1) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    const unsigned lane = threadIdx.x & 63u;
    struct Rows {
        const float *p;
        __device__ float operator[](int64_t i) const { return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    } rows{partials};
    // wavefront `wave` stands in for wavefronts wave, wave + wpb, ... of column_sum — all of their loads of a round in flight together
    // (these loads go to memory: a round trip is ~0.7 us, and one wavefront after the other paid it VW times per 64 columns)
    constexpr int VW = REDUCE_WAVES / MAX_WAVES_PER_BLOCK;
    for (int e0 = 0; e0 < NV + (loss ? 1 : 0); e0 += WAVE) {
        const int e = e0 + (int)lane;
        const bool live = e < NV || (loss && e == NV);
        const int64_t col = live ? e : 0;
        float sum[VW][REDUCE_UNROLL];
#pragma unroll
        for (int j = 0; j < VW; ++j)
#pragma unroll
            for (int u = 0; u < REDUCE_UNROLL; ++u) sum[j][u] = 0.0f;
        for (int r0 = 0; r0 < n_rows; r0 += REDUCE_UNROLL * REDUCE_WAVES) {
            float v[VW][REDUCE_UNROLL];
#pragma unroll
            for (int j = 0; j < VW; ++j)
#pragma unroll
                for (int u = 0; u < REDUCE_UNROLL; ++u) {
                    const int r = r0 + wave + j * wpb + u * REDUCE_WAVES;
                    v[j][u] = r < n_rows ? rows[(int64_t)r * pitch + col] : 0.0f;
                }
#pragma unroll
            for (int j = 0; j < VW; ++j)
#pragma unroll
                for (int u = 0; u < REDUCE_UNROLL; ++u) sum[j][u] += v[j][u];
        }
#pragma unroll
        for (int j = 0; j < VW; ++j) {
#pragma unroll
            for (int w = REDUCE_UNROLL / 2; w >= 1; w >>= 1)
#pragma unroll
                for (int u = 0; u < w; ++u) sum[j][u] += sum[j][u + w];
            lds[(wave + j * wpb) * WAVE + lane] = live ? sum[j][0] : 0.0f;
        }
        __syncthreads();
        if (wave == 0) {
            float total = 0.0f;
#pragma unroll
            for (int w = 0; w < REDUCE_WAVES; ++w) total += lds[w * WAVE + lane];
            if (loss && e == NV) loss[0] = total * loss_scale;
            if (grad_ops_f && e < NV) store_fk_gradient(grad_ops_f, e, total);
        }
        __syncthreads();
    }
    if (grad_ops_f)      // the FT block holds exactly the 12 differentiated constants
        for (int i = (int)threadIdx.x; i < CAP * (DRM_OPF_STRIDE - BWD_FIELDS); i += (int)blockDim.x)
            grad_ops_f[(i / (DRM_OPF_STRIDE - BWD_FIELDS)) * DRM_OPF_STRIDE + BWD_FIELDS + i % (DRM_OPF_STRIDE - BWD_FIELDS)] = 0.0f;
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Serial-chain ("arm") specialisation for ONE target at the end of the chain, full tiles only (BASELINE config 5:
// iiwa, loss on the end-effector position): drm_sample.hpp fk_backward_chain — one packed chain FK per sample and
// closed-form adjoints instead of stored poses and an adjoint sweep; constants staged once per wave in LDS, the
// gradient tile staged over the dead q tile.  Same persistent-wave structure and batch reduction as the generic kernel.
// MSE (drm_fk_mse): `gpos` is the TARGET position [B, 3]; the kernel forms the loss sum((p_e - target)^2) (one more column of
// the partial rows, row pitch NV + 4) and its gradient g = g_scale (p_e - target), g_scale = 2 / (3 B), itself: forward
// kinematics, loss and backward of the reference's kinematics-learning step (examples/learn_kinematics_of_iiwa.py:47-55) in ONE
// pass over q — the chain is walked once, nothing per-sample is written unless grad_q is asked for.
// PRE (round 5): the register-resident table (96 VGPRs of FT blocks, 256 VGPR + AGPRs in all: ONE wavefront per SIMD) is the latency
// form of launches that put at most one tile on a wavefront; larger launches — the persistent grid walks many tiles per wavefront —
// read the rows from LDS op by op and fit several wavefronts per SIMD instead.
#ifndef DRM_FK_BWD_PRE_MAX_TILES
#define DRM_FK_BWD_PRE_MAX_TILES 1024 /* one tile per SIMD */
#endif
constexpr int FK_BWD_PRE_MAX_TILES = DRM_FK_BWD_PRE_MAX_TILES;
// LINKS (drm_fk_mse_links): the walk table is built INSIDE the launch.  `ops_f` is the table of the constant links gathered into walk
// order; the entries that come from a learnable link (sel[e] = slot * 32 + element of its link row, -1 otherwise) are rebuilt by every
// wavefront from that link's rot_angles / trans — drm_walk_table's arithmetic (link_row, then x gsign), without its launch.
struct LinkArgs {
    const float *rpy[DRM_FK_MSE_MAX_LINKS], *trans[DRM_FK_MSE_MAX_LINKS];
    const int32_t *sel;
    const float *gsign;
    int n_links;
};
// rot_angles (3) + trans (3) of every learnable link: lane i < 6 of the calling wavefront holds element i of link l in pv[l] (the
// loads are ISSUED here and waited for where pv is first used: callers put their other loads in between)
__device__ __forceinline__ void link_args_load(const LinkArgs &la, unsigned lane, float (&pv)[DRM_FK_MSE_MAX_LINKS]) {
#pragma unroll
    for (int l = 0; l < DRM_FK_MSE_MAX_LINKS; ++l) {      // (unrolled: every pointer is a scalar at a fixed offset of the argument block)
        pv[l] = 0.0f;
        if (l < la.n_links && lane < 6u) {
            const float *src = lane < 3u ? la.rpy[l] + lane : la.trans[l] + (lane - 3u);
            pv[l] = *src;
        }
    }
}
__device__ __forceinline__ void link_args_to_lds(const LinkArgs &la, unsigned lane, const float (&pv)[DRM_FK_MSE_MAX_LINKS], float *params) {
#pragma unroll
    for (int l = 0; l < DRM_FK_MSE_MAX_LINKS; ++l)
        if (l < la.n_links && lane < 6u) params[l * 6 + (int)lane] = pv[l];
}
template <int CAP, int NJ, bool MSE = false, bool PRE = true, bool LINKS = false>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    fk_backward_arm_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ gpos,
                           int n_tiles, uint64_t param_mask, float *__restrict__ gq, float *partials, float g_scale,
                           uint32_t *ticket, float *grad_ops_f, float *loss, float loss_scale, LinkArgs la) {
    static_assert(NJ & 1, "odd row widths only (linear LDS image)");
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, Q_FLOATS = round4(WAVE * NJ), G_FLOATS = round4(WAVE * 3);
    constexpr int NV = CAP * BWD_FIELDS, A_FLOATS = round4(NV + 1);      // (+ the loss column of the MSE form)
    constexpr int PER_WAVE = C_FLOATS + Q_FLOATS + G_FLOATS + A_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[MAX_WAVES_PER_BLOCK * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int wave_id = (int)blockIdx.x * MAX_WAVES_PER_BLOCK + wave, n_waves = (int)gridDim.x * MAX_WAVES_PER_BLOCK;
    float *lc = smem + wave * PER_WAVE;
    float *lq = lc + C_FLOATS, *lg = lq + Q_FLOATS;
    float *lacc = lg + G_FLOATS; // this wave's running sums of the constant gradients

    // LINKS: everything the table build waits for is requested first, then the first tile's rows (a lone wavefront would otherwise pay
    // the two round trips one after the other)
#if defined(DRM_TIMELINE) && DRM_TL_WHICH == 2
#define ARM_STAMP(slot) do { if constexpr (LINKS) DRM_STAMP(slot); } while (0)
#define ARM_STAMP_DRAINED(slot) do { if constexpr (LINKS) DRM_STAMP_DRAINED(slot); } while (0)
#else
#define ARM_STAMP(slot) ((void)0)
#define ARM_STAMP_DRAINED(slot) ((void)0)
#endif
    ARM_STAMP(0);
    float pv[DRM_FK_MSE_MAX_LINKS];
    int4 sel4;
    float4 sg4;
    if constexpr (LINKS) {
        link_args_load(la, lane, pv);
        sel4 = reinterpret_cast<const int4 *>(la.sel)[lane];
        sg4 = reinterpret_cast<const float4 *>(la.gsign)[lane];
    }
    float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];
    float qv[NJ], gv[3];
    auto load_rows = [&](int tile) {
        const int64_t b0 = (int64_t)tile * WAVE;
        const float *qrow = q + (b0 + lane) * NJ, *grow = gpos + (b0 + lane) * 3;
#pragma unroll
        for (int d = 0; d < NJ; ++d) qv[d] = qrow[d];
#pragma unroll
        for (int i = 0; i < 3; ++i) gv[i] = grow[i];
    };
    if constexpr (LINKS) {
        if (wave_id < n_tiles) load_rows(wave_id);
        static_assert(DRM_FK_MSE_MAX_LINKS * 12 <= Q_FLOATS && DRM_FK_MSE_MAX_LINKS * 6 <= G_FLOATS, "scratch of the table build");
        float *lrows = lq, *lpar = lg;          // (both tiles are written per tile further down)
        link_args_to_lds(la, lane, pv, lpar);
        wave_lds_sync();
        ARM_STAMP(1);
        if ((int)lane < la.n_links) {
            float p[LINK_PARAM_FLOATS], row[DRM_OPF_STRIDE];
#pragma unroll
            for (int i = 0; i < LINK_PARAM_FLOATS; ++i) p[i] = i < 6 ? lpar[lane * 6 + i] : 0.0f;
            link_row(p, row);                   // (only F and t are live: forward kinematics reads nothing else of a row)
#pragma unroll
            for (int i = 0; i < 12; ++i) lrows[lane * 12 + i] = row[i];
        }
        wave_lds_sync();
        auto entry = [&](int r, float sign, float constant) {
            const bool live = r >= 0 && (r & (DRM_OPF_STRIDE - 1)) < 12;
            const float v = lrows[live ? (r >> 5) * 12 + (r & (DRM_OPF_STRIDE - 1)) : 0];
            return live ? v * sign : constant;
        };
        cv.x = entry(sel4.x, sg4.x, cv.x); cv.y = entry(sel4.y, sg4.y, cv.y);
        cv.z = entry(sel4.z, sg4.z, cv.z); cv.w = entry(sel4.w, sg4.w, cv.w);
        wave_lds_sync();
        ARM_STAMP(2);
    }
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    for (int i = (int)lane; i < NV; i += WAVE) lacc[i] = 0.0f;
    float loss_acc = 0.0f; // (lane 63 of the wave: the squared errors of its tiles, in tile order)

    // Launches of at most one wave per SIMD (BASELINE configuration 5: 16 384 rows = 256 waves) are bound by what a lone wave
    // waits for: the FT blocks of the chain are read into registers in one burst (as in fk_jacobian_arm_kernel's PRE form), and
    // every lane reads its own rows of q / gpos straight from memory (28 + 12 contiguous bytes) instead of through an LDS tile.
    wave_lds_sync();
    float tabr[PRE ? CAP : 1][DRM_OPF_FT_FLOATS];
    if constexpr (PRE) {
#pragma unroll
        for (int k = 0; k < CAP; ++k)
#pragma unroll
            for (int i = 0; i < DRM_OPF_FT_FLOATS; ++i) tabr[k][i] = lc[k * DRM_OPF_STRIDE + i];
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int tile = wave_id; tile < n_tiles; tile += n_waves) {
        const int64_t b0 = (int64_t)tile * WAVE;
        float gqv[NJ];
        if (!LINKS || tile != wave_id) load_rows(tile);
        wave_lds_sync(); // the previous tile's staged gradients have left
#pragma unroll
        for (int d = 0; d < NJ; ++d) lq[lane * NJ + d] = qv[d]; // (the cold parameter loop reads an angle by run-time index)
        wave_lds_sync();
        fk_backward_chain_g<CAP, NJ>([&](int k) -> const float * { if constexpr (PRE) return tabr[k]; else return lc + k * DRM_OPF_STRIDE; },
                                   [&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, qv,
                                   [&](const float (&pe)[3], float (&g)[3]) {
                                       ARM_STAMP(4);      // (the forward sweep is done)
                                       if constexpr (MSE) {
                                           const float e[3] = {pe[0] - gv[0], pe[1] - gv[1], pe[2] - gv[2]};
                                           const float sq = wave_sum_lane63(e[0] * e[0] + (e[1] * e[1] + e[2] * e[2]));
                                           if (lane == 63u) loss_acc += sq;
                                           g[0] = g_scale * e[0]; g[1] = g_scale * e[1]; g[2] = g_scale * e[2];
                                       } else {
                                           g[0] = gv[0]; g[1] = gv[1]; g[2] = gv[2];
                                       }
                                   },
                                   param_mask, gqv, [&](int d) -> float { return lq[lane * NJ + d]; },
                                   [&](int k, const float *dF, const float *dt) {
                                       ARM_STAMP(6);      // (the adjoint sweep has reached a learnable op)
                                       // (PRE: one wavefront per SIMD, at most one tile per wavefront — all twelve sums at once, and
                                       // nothing to add to; the streaming form four at a time)
                                       wave_sums_lane63<BWD_FIELDS, PRE ? BWD_FIELDS : 4>(
                                           lane, [&](int j) { return j < 9 ? dF[j] : dt[j - 9]; },
                                           [&](int j, float total) {
                                               if constexpr (PRE)
                                                   lacc[k * BWD_FIELDS + j] = total;
                                               else
                                                   lacc[k * BWD_FIELDS + j] += total; // tiles in a fixed order
                                           });
                                   });
        ARM_STAMP(3);
        if (gq) {
            wave_lds_sync(); // every lane is done with its q row
#pragma unroll
            for (int d = 0; d < NJ; ++d) lq[lane * NJ + d] = gqv[d];
            wave_lds_sync();
            tile_store<NJ>(gq + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
        }
    }
    // ONE row of partial sums per BLOCK (round 6): the wavefronts' running sums are added in wavefront order through LDS before they
    // leave the CU.  The reduction launch then reads a quarter of the rows — at 16 384 rows 25.6 KB instead of 100 KB, which one CU
    // (drm_fk_mse_links' finish kernel) spent 0.8 us just pulling through its L1.
    constexpr int PITCH = MSE ? NV + 4 : NV, COLS = MSE ? NV + 1 : NV;
    if (MSE && lane == 63u) lacc[NV] = loss_acc;
    __syncthreads();
    if ((int)threadIdx.x < COLS) {
        float total = 0.0f;
#pragma unroll
        for (int w = 0; w < MAX_WAVES_PER_BLOCK; ++w) total += smem[w * PER_WAVE + (C_FLOATS + Q_FLOATS + G_FLOATS) + (int)threadIdx.x];
        publish(partials + (int64_t)blockIdx.x * PITCH + (int)threadIdx.x, total);
    }
    ARM_STAMP_DRAINED(5);
    // ABI 11 (drm_walk.special[DRM_WALK_TICKET]): ONE launch.  The block that takes the last ticket adds the rows of partial sums
    // itself — in column_sum's order, so the totals are the two-launch form's bit for bit — instead of a second kernel behind a
    // ~1.7 us kernel boundary (BASELINE configuration 5: 16 384 rows, where the boundary was a quarter of drm_fk_mse).
    if (ticket) last_block_reduce<CAP>(ticket, partials, (int)gridDim.x, PITCH, grad_ops_f, MSE ? loss : nullptr, loss_scale, smem);
}

// grad_ops_f[k, FT field j] = sum over the partial rows of column k * 12 + j, in a fixed order (drm_common.hpp
// column_sum); the rest of the row has no gradient here and is zeroed.
// `pitch` = floats per partial row (NV, or NV + 4 with the loss column of drm_fk_mse: loss[0] = loss_scale x its column sum);
// grad_ops_f may be NULL (drm_fk_mse without learnable links: only the loss is reduced)
__global__ void __launch_bounds__(WAVE *REDUCE_WAVES)
    fk_backward_reduce_kernel(const float *__restrict__ partials, int n_rows, int cap, float *__restrict__ grad_ops_f, int pitch,
                              float *__restrict__ loss, float loss_scale) {
    __shared__ float lds[REDUCE_WAVES][WAVE];
    const int NV = cap * BWD_FIELDS, e = (int)blockIdx.x * WAVE + (int)(threadIdx.x & 63u);
    const bool is_loss = loss && e == NV;
    const float total = column_sum(partials, n_rows, pitch, e, e < NV || is_loss, lds);
    if (threadIdx.x < WAVE && is_loss) loss[0] = total * loss_scale;
    if (!grad_ops_f) return;
    if (threadIdx.x < WAVE && e < NV) store_fk_gradient(grad_ops_f, e, total);
    // the FT block holds exactly the 12 differentiated constants
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < cap * (DRM_OPF_STRIDE - BWD_FIELDS);
         i += (int)(gridDim.x * blockDim.x))
        grad_ops_f[(i / (DRM_OPF_STRIDE - BWD_FIELDS)) * DRM_OPF_STRIDE + BWD_FIELDS + i % (DRM_OPF_STRIDE - BWD_FIELDS)] = 0.0f;
}

// The second launch of drm_fk_mse_links: the rows of partial sums -> d loss / d (F, t) of every op (fk_backward_reduce_kernel's sums,
// in its order) -> d loss / d (rot_angles, trans) of the learnable links (drm_walk_table_backward's sums, in its order, and
// link_row_backward), and the loss.  One block; capacity-8 walks.
// Latency-bound (one block, 25.6 KB read at 16 384 rows): everything that does not depend on the sums — the gather map of the 96 differentiated
// entries, the links' parameters, d F / d rpy of every learnable link — is requested / computed while the rows are in flight, both
// halves of the 97 columns are summed in ONE pass, and after the block's only barrier wavefront 0 finishes alone.
#ifndef DRM_FIN_WAVES
#define DRM_FIN_WAVES 8
#endif
#if defined(DRM_TIMELINE) && DRM_TL_WHICH == 1 /* development: stamps of the finish kernel (tools/timeline_links.py) */
#define FIN_STAMP(slot) DRM_STAMP(slot)
#define FIN_STAMP_DRAINED(slot) DRM_STAMP_DRAINED(slot)
#else
#define FIN_STAMP(slot) ((void)0)
#define FIN_STAMP_DRAINED(slot) ((void)0)
#endif
// (FIN_WAVES wavefronts stand in for column_sum's REDUCE_WAVES, VW each: 16 x 64 threads would cap a wavefront at 128 VGPRs, which the
// 32 rows in flight per column pair + the trigonometry under them do not fit)
constexpr int FIN_WAVES = DRM_FIN_WAVES, FIN_VW = REDUCE_WAVES / FIN_WAVES;
// One more wavefront than the FIN_WAVES that sum: it only turns the links' rot_angles into d F / d (roll, pitch, yaw) — 1.1 us of
// dependent arithmetic that ends with the sums instead of after them.
__global__ void __launch_bounds__(WAVE *(FIN_WAVES + 1))
    fk_mse_links_finish_kernel(const float *__restrict__ partials, int n_rows, LinkArgs la, float *__restrict__ grad_params,
                               float *__restrict__ loss, float loss_scale) {
    constexpr int CAP = 8, NV = CAP * BWD_FIELDS, PITCH = NV + 4, FT = 12, NT = DRM_FK_MSE_MAX_LINKS * FT;
    static_assert(REDUCE_WAVES % FIN_WAVES == 0, "whole virtual wavefronts per wavefront");
    __shared__ float lds[2][REDUCE_WAVES][WAVE];
    __shared__ float ge[CAP * FT], grows[NT], lpar[DRM_FK_MSE_MAX_LINKS * 6], dmat[DRM_FK_MSE_MAX_LINKS][27];
    __shared__ int se[CAP * FT];
    __shared__ unsigned owner[NT], count[NT];      // per element of a learnable link's (F, t): the first walk entry gathered from it, how many
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63u);
    FIN_STAMP(0);
    if (wave == FIN_WAVES) {      // the trigonometry wavefront
        float pv[DRM_FK_MSE_MAX_LINKS];
        link_args_load(la, (unsigned)lane, pv);
        for (int i = lane; i < NT; i += WAVE) { owner[i] = 0xffffffffu; count[i] = 0u; }
        link_args_to_lds(la, (unsigned)lane, pv, lpar);
        wave_lds_sync();
        if (lane < la.n_links) {
            float rpy[3], D[27];
#pragma unroll
            for (int i = 0; i < 3; ++i) rpy[i] = lpar[lane * 6 + i];
            rpy_jacobian(rpy, D, D + 9, D + 18);
#pragma unroll
            for (int i = 0; i < 27; ++i) dmat[lane][i] = D[i];
        }
        FIN_STAMP_DRAINED(3);
        __syncthreads();
        return;
    }
    // column e = op k, field j of (dF row-major, dt) -> entry `at` of the op's row (store_fk_gradient).  A lane holds columns 2 lane and
    // 2 lane + 1: ONE 8-byte load per lane and row (a row is 100 floats: 50 lanes), half the load instructions of one column per lane
    // — the block's 25 KB all go through one CU's address path, which is what this phase waits for.
    int at[2], sel[2] = {-1, -1};
    float sign[2] = {0.0f, 0.0f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = 2 * lane + h, k = e / BWD_FIELDS, j = e % BWD_FIELDS;
        const int a = j < 9 ? DRM_OPF_FIJ(j / 3, j % 3) : DRM_OPF_TI(j - 9);
        at[h] = k * FT + a;
        if (wave == 0 && e < NV) { sel[h] = la.sel[k * DRM_OPF_STRIDE + a]; sign[h] = la.gsign[k * DRM_OPF_STRIDE + a]; }
    }
    // column_sum's order for both columns; virtual wavefront vw = wave + j * FIN_WAVES adds rows vw, vw + 16, ...
    // (32-bit byte offsets from the scalar base: the load's own addressing mode, one add per load)
    const bool live0 = 2 * lane <= NV, live1 = 2 * lane + 1 <= NV;      // (column NV is the loss; NV + 1 .. NV + 3 pad the row)
    const unsigned c0 = (unsigned)(wave * PITCH + (live0 ? 2 * lane : 0));
    auto at_float2 = [&](unsigned index) -> float2 {
        return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(partials) + index * 4u);
    };
    float s0[FIN_VW][REDUCE_UNROLL], s1[FIN_VW][REDUCE_UNROLL];
#pragma unroll
    for (int j = 0; j < FIN_VW; ++j)
#pragma unroll
        for (int u = 0; u < REDUCE_UNROLL; ++u) s0[j][u] = s1[j][u] = 0.0f;
    int r = 0;
    for (; r + (REDUCE_WAVES - 1) + (REDUCE_UNROLL - 1) * REDUCE_WAVES < n_rows; r += REDUCE_UNROLL * REDUCE_WAVES) {
        float2 v[FIN_VW][REDUCE_UNROLL];
#pragma unroll
        for (int j = 0; j < FIN_VW; ++j)
#pragma unroll
            for (int u = 0; u < REDUCE_UNROLL; ++u) v[j][u] = at_float2((unsigned)(r + j * FIN_WAVES + u * REDUCE_WAVES) * PITCH + c0);
        FIN_STAMP(1);
#pragma unroll
        for (int j = 0; j < FIN_VW; ++j)
#pragma unroll
            for (int u = 0; u < REDUCE_UNROLL; ++u) { s0[j][u] += v[j][u].x; s1[j][u] += v[j][u].y; }
    }
    if (r < n_rows) {    // (uniform) the last, partial round: same slots, rows past the end add nothing.  BASELINE configuration 5:
        // 64 rows (one per block of the chain kernel), four per virtual wavefront.  Every load of the round is requested before the first
        // is added (clamped addresses instead of branches), for as many slots as have a row left: 2, 8 or all 16.
        auto round = [&](auto slots) {
            constexpr int U = decltype(slots)::value;
            float2 a[FIN_VW][U];
#pragma unroll
            for (int j = 0; j < FIN_VW; ++j)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool in = r + wave + j * FIN_WAVES + u * REDUCE_WAVES < n_rows;
                    a[j][u] = at_float2((unsigned)(in ? r + j * FIN_WAVES + u * REDUCE_WAVES : 0) * PITCH + c0);
                }
#pragma unroll
            for (int j = 0; j < FIN_VW; ++j)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool in = r + wave + j * FIN_WAVES + u * REDUCE_WAVES < n_rows;
                    s0[j][u] += in ? a[j][u].x : 0.0f;
                    s1[j][u] += in ? a[j][u].y : 0.0f;
                }
        };
        const int u_live = (n_rows - r + REDUCE_WAVES - 1) / REDUCE_WAVES;      // slots with any row left (uniform)
        if (u_live <= 2)
            round(std::integral_constant<int, 2>());
        else if (u_live <= 8)
            round(std::integral_constant<int, 8>());
        else
            round(std::integral_constant<int, REDUCE_UNROLL>());
    }
#pragma unroll
    for (int j = 0; j < FIN_VW; ++j) {
#pragma unroll
        for (int w = REDUCE_UNROLL / 2; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) { s0[j][u] += s0[j][u + w]; s1[j][u] += s1[j][u + w]; }
        lds[0][wave + j * FIN_WAVES][lane] = live0 ? s0[j][0] : 0.0f;
        lds[1][wave + j * FIN_WAVES][lane] = live1 ? s1[j][0] : 0.0f;
    }
    FIN_STAMP_DRAINED(3);
    __syncthreads();
    FIN_STAMP(4);
    if (wave != 0) return;
    float total[2] = {0.0f, 0.0f};
#pragma unroll
    for (int w = 0; w < REDUCE_WAVES; ++w) { total[0] += lds[0][w][lane]; total[1] += lds[1][w][lane]; }
    if (2 * lane == NV) loss[0] = total[0] * loss_scale;
    // d loss / d (element of a learnable link's row) = the walk entries gathered from it, added in entry order
    // (drm_walk_table_backward).  A link is one op of a walk, so nearly always ONE entry per element: every entry names itself at its
    // element (the lowest index wins) and is counted; an element with one entry takes it, one with several scans them in order.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = 2 * lane + h;
        if (e < NV) {
            ge[at[h]] = total[h] * sign[h];
            se[at[h]] = sel[h];
            const int col = sel[h] & (DRM_OPF_STRIDE - 1), slot = sel[h] >> 5;
            if (sel[h] >= 0 && col < FT && slot < la.n_links) {
                atomicMin(&owner[slot * FT + col], (unsigned)at[h]);
                atomicAdd(&count[slot * FT + col], 1u);
            }
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = h * WAVE + lane;
        if (t < la.n_links * FT) {
            const unsigned n = count[t];
            float s = 0.0f;
            if (n == 1u) {
                s += ge[owner[t]];
            } else if (n > 1u) {
                const int want = (t / FT) * DRM_OPF_STRIDE + t % FT;
                for (int e = 0; e < CAP * FT; ++e) s += se[e] == want ? ge[e] : 0.0f;
            }
            grows[t] = s;
        }
    }
    wave_lds_sync();
    FIN_STAMP_DRAINED(5);
    if (lane < la.n_links) {           // link_row_backward for the (F, t) block: its first six outputs; the other 14 are zeros
        float g[FT];
#pragma unroll
        for (int i = 0; i < FT; ++i) g[i] = grows[lane * FT + i];
        float *out = grad_params + lane * LINK_PARAM_FLOATS;
#pragma unroll
        for (int a = 0; a < 3; ++a) out[a] = dot9(&dmat[lane][a * 9], g);
#pragma unroll
        for (int i = 0; i < 3; ++i) out[3 + i] = g[9 + i];
#pragma unroll
        for (int i = 6; i < LINK_PARAM_FLOATS; ++i) out[i] = 0.0f;
    }
    FIN_STAMP_DRAINED(6);
}

} // namespace drm

using namespace drm;

// walks of more than FK_BACKWARD_LDS_OPS links park their per-op poses in HBM (3 KB per op and wave in LDS otherwise)
constexpr int FK_BACKWARD_LDS_OPS = 12;

extern "C" int64_t drm_fk_backward_scratch_floats(int64_t B, int32_t capacity) {
    if (B < 0 || capacity < 1 || capacity > DRM_MAX_OPS) return 0;
    // rows: waves of the launch rounded up to a full block of MAX_WAVES_PER_BLOCK (+ a ragged tail's rows)
    const int64_t waves = backward_waves(B, MAX_WAVES_PER_BLOCK);
    int64_t floats = (waves + MAX_WAVES_PER_BLOCK) * capacity * BWD_FIELDS;
    if (capacity > FK_BACKWARD_LDS_OPS) floats += (waves + MAX_WAVES_PER_BLOCK) * (int64_t)capacity * POSE_FLOATS * WAVE;
    return floats;
}

static int fk_backward_launch(const drm_walk *w, const float *q, int64_t B, int32_t n_targets, const float *grad_pos,
                              const float *grad_rot, const float *grad_lin, const float *grad_ang, uint64_t param_mask, float *grad_q,
                              float *grad_ops_f, float *scratch, void *stream) {
    const bool jac = grad_lin != nullptr;
    if (B < 0 || n_targets < 1) return fail(DRM_ERR_INVALID, "negative batch or no targets");
    if (n_targets > w->n_ops) return fail(DRM_ERR_INVALID, "more targets than ops in the walk");
    if ((param_mask != 0) != (grad_ops_f != nullptr))
        return fail(DRM_ERR_INVALID, "grad_ops_f must be given exactly when param_mask selects ops");
    if (!grad_q && !grad_ops_f) return fail(DRM_ERR_INVALID, "nothing to compute: grad_q and grad_ops_f are both NULL");
    if (!scratch) return fail(DRM_ERR_INVALID, "scratch must not be NULL (drm_fk_backward_scratch_floats)");
    if (w->capacity < 64 && (param_mask >> w->capacity)) return fail(DRM_ERR_INVALID, "param_mask selects ops beyond the walk's capacity");
    const int n = w->n_dofs, T = n_targets, cap = w->capacity;
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        if (grad_ops_f) {
            hipError_t e = hipMemsetAsync(grad_ops_f, 0, sizeof(float) * cap * DRM_OPF_STRIDE, s);
            if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
        return DRM_OK;
    }
    int rc;
    const int64_t B_total = B;
    int rows_done = 0;          // rows of partial sums already written by the arm kernel
    float *partials = scratch;
#ifndef DRM_NO_ARM_KERNEL
    {
        const uintptr_t ptrs = (uintptr_t)q | (uintptr_t)grad_pos | (uintptr_t)grad_q | (uintptr_t)w->ops_f;
        if (!jac && !grad_rot && T == 1 && (w->shape & DRM_WALK_ARM_CHAIN) && cap == 8 && n == 7 && (ptrs & 15u) == 0 && B >= WAVE &&
            B / WAVE < 0x7fffffffLL) {
            // 7-DoF arms, one target at the end of the chain: full tiles through the chain kernel, the ragged tail (if
            // any) through the generic kernel below with its rows of partial sums appended
            const int n_tiles = (int)(B / WAVE);
            const int64_t done = (int64_t)n_tiles * WAVE;
            const int waves_a = backward_waves(done, MAX_WAVES_PER_BLOCK);
            // ABI 11: with a ticket word on the walk and no ragged tail, the block that finishes last reduces (one launch)
            uint32_t *ticket = (done == B && grad_ops_f) ? (uint32_t *)w->special[DRM_WALK_TICKET] : nullptr;
            if (n_tiles <= FK_BWD_PRE_MAX_TILES)
                hipLaunchKernelGGL((fk_backward_arm_kernel<8, 7, false, true>), dim3((unsigned)(waves_a / MAX_WAVES_PER_BLOCK)),
                                   dim3(WAVE * MAX_WAVES_PER_BLOCK), 0, s, w->ops_f, q, grad_pos, n_tiles, param_mask, grad_q,
                                   partials, 0.0f, ticket, grad_ops_f, (float *)nullptr, 0.0f, LinkArgs{});
            else
                hipLaunchKernelGGL((fk_backward_arm_kernel<8, 7, false, false>), dim3((unsigned)(waves_a / MAX_WAVES_PER_BLOCK)),
                                   dim3(WAVE * MAX_WAVES_PER_BLOCK), 0, s, w->ops_f, q, grad_pos, n_tiles, param_mask, grad_q,
                                   partials, 0.0f, ticket, grad_ops_f, (float *)nullptr, 0.0f, LinkArgs{});
            rc = launched();
            if (rc) return rc;
            if (ticket) return DRM_OK;
            rows_done = waves_a / MAX_WAVES_PER_BLOCK;      // (one row per block)
            partials += (int64_t)rows_done * cap * BWD_FIELDS;
            q += done * n; grad_pos += done * 3;   // (grad_rot is NULL on this path)
            if (grad_q) grad_q += done * n;
            B -= done;
        }
    }
#endif
    int waves = 0;
    if (B > 0) {
        const bool park_hbm = cap > FK_BACKWARD_LDS_OPS;
        Geometry g;
        rc = make_geometry(B, 2 * round4(WAVE * pad_odd(n)) + round4(WAVE * pad_odd(3 * T)) + round4(cap * BWD_FIELDS) +
                                  w->n_slots * 24 * WAVE + (jac ? 2 * round4(WAVE * pad_odd(3 * n)) : 0) +
                                  (park_hbm ? 0 : cap * POSE_FLOATS * WAVE), g);
        if (rc) return rc;
        int wpb = (int)(g.block.x / WAVE);
        if (rows_done) { wpb = 1; g.block = dim3(WAVE); g.lds_bytes = (size_t)g.lds_per_wave * sizeof(float); } // a tail: one wave
        waves = backward_waves(B, wpb);
        g.grid = dim3((unsigned)(waves / wpb));
        // parked poses live behind ALL rows of partial sums (the arm kernel's and this launch's)
        float *park = scratch + (int64_t)(backward_waves(B_total, MAX_WAVES_PER_BLOCK) + MAX_WAVES_PER_BLOCK) * cap * BWD_FIELDS;
        const uint32_t align = al16(q, AL_Q) | al16(grad_pos, AL_POS) | al16(grad_q, AL_TAU) | al16(grad_lin, AL_LIN) |
                               al16(grad_ang, AL_ANG);
#define DRM_LAUNCH_FB(JAC, HBM)                                                                                        \
    {                                                                                                                  \
        rc = ensure_lds(fk_backward_kernel<JAC, HBM>, g.lds_bytes);                                                    \
        if (rc) return rc;                                                                                             \
        hipLaunchKernelGGL((fk_backward_kernel<JAC, HBM>), g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, cap,   \
                           (int)w->n_ops, n, (int)w->n_slots, T, q, grad_pos, grad_rot, grad_lin, grad_ang, B, grad_q, \
                           param_mask, partials, park, div_magic(n), div_magic(3 * T), div_magic(3 * n),               \
                           g.lds_per_wave, align);                                                                     \
    }
        if (jac) {
            if (park_hbm) DRM_LAUNCH_FB(true, true) else DRM_LAUNCH_FB(true, false)
        } else {
            if (park_hbm) DRM_LAUNCH_FB(false, true) else DRM_LAUNCH_FB(false, false)
        }
#undef DRM_LAUNCH_FB
        rc = launched();
        if (rc) return rc;
    }
    if (grad_ops_f) {
        hipLaunchKernelGGL(fk_backward_reduce_kernel, dim3((unsigned)((cap * BWD_FIELDS + WAVE - 1) / WAVE)),
                           dim3(WAVE * REDUCE_WAVES), 0, s, scratch, rows_done + waves, cap, grad_ops_f, cap * BWD_FIELDS,
                           (float *)nullptr, 0.0f);
        rc = launched();
        if (rc) return rc;
    }
    return DRM_OK;
}

extern "C" int drm_fk_backward(const drm_walk *w, const float *q, int64_t B, int32_t n_targets, const float *grad_pos,
                               const float *grad_rot, uint64_t param_mask, float *grad_q, float *grad_ops_f, float *scratch,
                               void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !grad_pos) return fail(DRM_ERR_INVALID, "q / grad_pos must not be NULL");
    if (w->n_slots > DRM_MAX_SLOTS_BACKWARD || w->capacity > DRM_MAX_OPS)
        return fail(DRM_ERR_UNSUPPORTED, "the backward kernels take walks of up to %s%ld links and %ld save slots", "", (long)DRM_MAX_OPS,
                    (long)DRM_MAX_SLOTS_BACKWARD);
    return fk_backward_launch(w, q, B, n_targets, grad_pos, grad_rot, nullptr, nullptr, param_mask, grad_q, grad_ops_f, scratch,
                              stream);
}

extern "C" int64_t drm_fk_mse_scratch_floats(int64_t B, int32_t capacity) {
    if (B < 0 || capacity < 1 || capacity > DRM_MAX_OPS) return 0;
    return (int64_t)(backward_waves(B, MAX_WAVES_PER_BLOCK) + MAX_WAVES_PER_BLOCK) * (capacity * BWD_FIELDS + 4);
}

// FK(end effector position) + mean squared error against `target` + the gradients of that loss, in two launches (the chain
// kernel, the fixed-order reduction): see include/drm_hip.h
extern "C" int drm_fk_mse(const drm_walk *w, const float *q, const float *target, int64_t B, uint64_t param_mask, float *loss,
                          float *grad_q, float *grad_ops_f, float *scratch, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !target || !loss || !scratch) return fail(DRM_ERR_INVALID, "q / target / loss / scratch must not be NULL");
    if ((param_mask != 0) != (grad_ops_f != nullptr))
        return fail(DRM_ERR_INVALID, "grad_ops_f must be given exactly when param_mask selects ops");
    const int cap = w->capacity;
    const uintptr_t ptrs = (uintptr_t)q | (uintptr_t)target | (uintptr_t)grad_q | (uintptr_t)w->ops_f;
    if (!((w->shape & DRM_WALK_ARM_CHAIN) && cap == 8 && w->n_dofs == 7 && (ptrs & 15u) == 0 && B >= WAVE && B % WAVE == 0 &&
          B / WAVE < 0x7fffffffLL))
        return fail(DRM_ERR_UNSUPPORTED, "drm_fk_mse takes 7-DoF arm chains (DRM_WALK_ARM_CHAIN, capacity 8), batches that are a "
                                         "multiple of 64 rows and 16-byte aligned pointers; compose drm_fk and drm_fk_backward otherwise%s", "");
    if (param_mask >> cap) return fail(DRM_ERR_INVALID, "param_mask selects ops beyond the walk's capacity");
    hipStream_t s = (hipStream_t)stream;
    const int n_tiles = (int)(B / WAVE);
    const int waves = backward_waves(B, MAX_WAVES_PER_BLOCK);
    uint32_t *ticket = (uint32_t *)w->special[DRM_WALK_TICKET];      // ABI 11: ONE launch, the last block reduces
    const float loss_scale = 1.0f / (3.0f * (float)B);
    if (n_tiles <= FK_BWD_PRE_MAX_TILES)
        hipLaunchKernelGGL((fk_backward_arm_kernel<8, 7, true, true>), dim3((unsigned)(waves / MAX_WAVES_PER_BLOCK)), dim3(WAVE * MAX_WAVES_PER_BLOCK),
                           0, s, w->ops_f, q, target, n_tiles, param_mask, grad_q, scratch, 2.0f / (3.0f * (float)B), ticket, grad_ops_f, loss,
                           loss_scale, LinkArgs{});
    else
        hipLaunchKernelGGL((fk_backward_arm_kernel<8, 7, true, false>), dim3((unsigned)(waves / MAX_WAVES_PER_BLOCK)), dim3(WAVE * MAX_WAVES_PER_BLOCK),
                           0, s, w->ops_f, q, target, n_tiles, param_mask, grad_q, scratch, 2.0f / (3.0f * (float)B), ticket, grad_ops_f, loss,
                           loss_scale, LinkArgs{});
    rc = launched();
    if (rc) return rc;
    if (ticket) return DRM_OK;
    hipLaunchKernelGGL(fk_backward_reduce_kernel, dim3((unsigned)((cap * BWD_FIELDS + 1 + WAVE - 1) / WAVE)), dim3(WAVE * REDUCE_WAVES), 0,
                       s, scratch, waves / MAX_WAVES_PER_BLOCK, cap, grad_ops_f, cap * BWD_FIELDS + 4, loss, 1.0f / (3.0f * (float)B));
    return launched();
}

// drm_fk_mse with the walk table built from the learnable links' parameters inside the first launch and the gradient taken back to
// them inside the second: see include/drm_hip.h
extern "C" int drm_fk_mse_links(const drm_walk *w, const int32_t *sel, const float *gsign, const drm_link_pieces *links, int32_t n_links,
                                const float *q, const float *target, int64_t B, uint64_t param_mask, float *loss, float *grad_q,
                                float *grad_params, float *scratch, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!sel || !gsign || !links || !q || !target || !loss || !grad_params || !scratch)
        return fail(DRM_ERR_INVALID, "drm_fk_mse_links: sel / gsign / links / q / target / loss / grad_params / scratch must not be NULL");
    if (n_links < 1 || n_links > DRM_FK_MSE_MAX_LINKS)
        return fail(DRM_ERR_UNSUPPORTED, "drm_fk_mse_links takes 1 .. %s%ld learnable links (%ld): compose drm_walk_table, drm_fk_mse, "
                                         "drm_walk_table_backward otherwise", "", (long)DRM_FK_MSE_MAX_LINKS, (long)n_links);
    const int cap = w->capacity;
    const uintptr_t ptrs = (uintptr_t)q | (uintptr_t)target | (uintptr_t)grad_q | (uintptr_t)w->ops_f | (uintptr_t)sel | (uintptr_t)gsign;
    if (!((w->shape & DRM_WALK_ARM_CHAIN) && cap == 8 && w->n_dofs == 7 && (ptrs & 15u) == 0 && B >= WAVE && B % WAVE == 0 &&
          B / WAVE < 0x7fffffffLL))
        return fail(DRM_ERR_UNSUPPORTED, "drm_fk_mse_links takes what drm_fk_mse takes: 7-DoF arm chains (capacity 8), batches that are a "
                                         "multiple of 64 rows, 16-byte aligned pointers%s", "");
    if (!param_mask || (param_mask >> cap)) return fail(DRM_ERR_INVALID, "param_mask must select ops of the walk (those of the learnable links)");
    LinkArgs la{};
    for (int l = 0; l < n_links; ++l) {
        if (!links[l].rot_angles || !links[l].trans) return fail(DRM_ERR_INVALID, "drm_fk_mse_links: rot_angles / trans of a link are NULL");
        la.rpy[l] = links[l].rot_angles;
        la.trans[l] = links[l].trans;
    }
    la.sel = sel; la.gsign = gsign; la.n_links = n_links;
    hipStream_t s = (hipStream_t)stream;
    const int n_tiles = (int)(B / WAVE);
    const int waves = backward_waves(B, MAX_WAVES_PER_BLOCK);
    const float loss_scale = 1.0f / (3.0f * (float)B), g_scale = 2.0f / (3.0f * (float)B);
    if (n_tiles <= FK_BWD_PRE_MAX_TILES)
        hipLaunchKernelGGL((fk_backward_arm_kernel<8, 7, true, true, true>), dim3((unsigned)(waves / MAX_WAVES_PER_BLOCK)),
                           dim3(WAVE * MAX_WAVES_PER_BLOCK), 0, s, w->ops_f, q, target, n_tiles, param_mask, grad_q, scratch, g_scale,
                           (uint32_t *)nullptr, (float *)nullptr, loss, loss_scale, la);
    else
        hipLaunchKernelGGL((fk_backward_arm_kernel<8, 7, true, false, true>), dim3((unsigned)(waves / MAX_WAVES_PER_BLOCK)),
                           dim3(WAVE * MAX_WAVES_PER_BLOCK), 0, s, w->ops_f, q, target, n_tiles, param_mask, grad_q, scratch, g_scale,
                           (uint32_t *)nullptr, (float *)nullptr, loss, loss_scale, la);
    rc = launched();
    if (rc) return rc;
    hipLaunchKernelGGL(fk_mse_links_finish_kernel, dim3(1), dim3(WAVE * (FIN_WAVES + 1)), 0, s, scratch, waves / MAX_WAVES_PER_BLOCK, la, grad_params, loss,
                       loss_scale);
    return launched();
}

extern "C" int drm_fk_jacobian_backward(const drm_walk *w, const float *q, int64_t B, const float *grad_pos,
                                        const float *grad_rot, const float *grad_lin_jac, const float *grad_ang_jac,
                                        uint64_t param_mask, float *grad_q, float *grad_ops_f, float *scratch, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !grad_lin_jac || !grad_ang_jac) return fail(DRM_ERR_INVALID, "q / grad_lin_jac / grad_ang_jac must not be NULL");
    if (w->n_slots != 0) return fail(DRM_ERR_INVALID, "the walk must be the root->link chain of the Jacobian's target (no branch points)");
    if (w->capacity > DRM_MAX_OPS) return fail(DRM_ERR_UNSUPPORTED, "the backward kernels take walks of up to %s%ld links", "", (long)DRM_MAX_OPS);
    return fk_backward_launch(w, q, B, 1, grad_pos, grad_rot, grad_lin_jac, grad_ang_jac, param_mask, grad_q, grad_ops_f, scratch,
                              stream);
}
DRM_TL_READER(fkb)
