// drm_forward_dynamics.hip — K8: joint accelerations from joint torques, qdd = H(q)^-1 (f - nle(q, qd)).
//
// Replaces DifferentiableRobotModel.compute_forward_dynamics (robot_model.py:487-624, Featherstone's
// articulated-body algorithm written as three Python loops over the links with 6x6 bmm's per link).  The
// articulated-body recursion is an O(n) elimination of the linear system H qdd = f - nle.  Where H is small — 7-DoF arm
// chains (forward_dynamics_arm_kernel) and robots whose segments are all short, the fingers of a hand
// (forward_dynamics_tree_kernel) — the kernels form the same system with the two walks they already have, the
// composite-rigid-body walk for H and RNEA with qdd = 0 for the bias torques nle, and solve it per sample in registers / LDS by
// an L^T D L factorisation taken from the leaves to the root, the elimination order of the recursion.  Every other robot (an
// arm carrying a gripper or a hand, a mobile manipulator) runs the recursion itself (forward_dynamics_aba_kernel): H of a
// 23-DoF robot would take a CU's LDS after one tile, and cond(H) times the rounding of its entries is what the result loses.
//
// Per sample: in q, qd, f [n] (12 n bytes), out qdd [n] (4 n bytes).          n = 7: 112 B
#include "drm_common.hpp"
#include "drm_sample.hpp"
#include "drm_tree_dev.hpp"

namespace drm {

constexpr int FD_SHORT_OPS = 6; // segments of up to this many ops in a row take crba_tree_walk_short

// Forward dynamics of a robot whose segments are all SHORT (the fingers of a hand, at most FD_SHORT_OPS ops each): one tile
// of 64 samples per block, one wavefront per segment.  Segments are independent (their joints share no link that moves),
// so H is block diagonal: every wavefront forms ITS block (drm_tree.hpp crba_tree_walk_short, packed lower triangle with
// segment-local DoF indices), the bias torques of its joints (rnea_tree_walk_short with qdd = 0, robot_model.py:377-400)
// and solves its block by the leaf-to-root L^T D L factorisation (drm_sample.hpp ltdl_factor_acc) — an Allegro hand is
// four 4 x 4 systems per sample, not one 16 x 16.  Both walks keep their per-op records in registers.
// LDS: [ table ][ f -> rhs -> qdd ][ q ][ qd ] shared, then per wavefront
//      [ slots : n_slots * 18 * 64 ][ triangle : 64 (nt|1) ]
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    forward_dynamics_tree_kernel(TreeArgs a, int flags, const float *__restrict__ q, const float *__restrict__ qd,
                                 const float *__restrict__ f, int64_t B, float *__restrict__ qdd, uint32_t magic_q, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n), region = round4(WAVE * Sq);
    float *lf = smem + table_lds_floats(a.n_ops), *lq = lf + region, *lqd = lq + region;
    const int first = a.seg_begin[wave], last = a.seg_begin[wave + 1];
    const int lo = a.seg_dof_lo[wave], cnt = a.seg_dof_cnt[wave], nt = cnt * (cnt + 1) / 2;
    float *lsl = smem + a.wave_off[wave]; // slots: inertia [slot][10][64], then motion [12] + force [6]
    float *lms = lsl, *lfs = lsl + a.n_slots * (12 * WAVE);
    float *ltri = lsl + a.n_slots * (18 * WAVE) + lane * pad_odd(nt); // this lane's packed triangle
    auto tri = [&](int i) -> float & { return ltri[i]; };

    const TableLds tab = stage_tree_table(a, smem);
    const bool fast = tc.full && (n & 1);
    if (wave == 0) tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, fast && (align & AL_Q), tc.full && (align & AL_Q));
    if (wave == (a.n_segments > 1 ? 1 : 0))
        tile_load<0>(qd + tc.b0 * n, tc.rows, n, magic_q, lqd, lane, fast && (align & AL_QD), tc.full && (align & AL_QD));
    if (wave == (a.n_segments > 2 ? 2 : 0))
        tile_load<0>(f + tc.b0 * n, tc.rows, n, magic_q, lf, lane, fast && (align & AL_QDD), tc.full && (align & AL_QDD));
    for (int i = 0; i < nt; ++i) tri(i) = 0.0f;
    __syncthreads();

    // lanes past a partial tile read zeros (see drm_fk.hip); their H is then a valid inertia matrix as well
    const bool live = (int)lane < tc.rows;
    const unsigned row = lane * Sq;
    const TableLds &ctl = tab;
    auto rowf = [&](int k) { return tab.row(k); };
    auto hput = [&](int di, int dj, float v) {
        if (di >= dj) tri(tri_index(di - lo, dj - lo)) = v;
    };
    auto qval = [&](int d) -> float { return live ? lq[row + d] : 0.0f; };
    // a serial segment (a finger): the unrolled walk with the joint transforms in registers; a short segment that branches
    // takes the loop, its cos / sin recomputed where the loop asks for them
    if (!crba_tree_walk_short<FD_SHORT_OPS>(first, last, ctl, rowf, qval, hput)) {
        for (int s = 0; s < a.n_slots * 10; ++s) lsl[s * WAVE + lane] = 0.0f;
        wave_lds_sync();
        crba_tree_walk(
            first, last, ctl, rowf,
            [&](int k, float &c, float &s, float &x) {
                int w0, w1;
                ctl_words(ctl, k, w0, w1);
                const OpCtl ct = decode_ctl(w0, w1);
                x = 0.0f; c = 1.0f; s = 0.0f;
                if (ct.dof >= 0) {
                    x = qval(ct.dof);
                    if (!ct.prismatic) sincos_one(x, s, c);
                }
            },
            [&](int s, const Inertia &I) { lds_add_inertia(lsl, s, lane, I); }, [&](int s, Inertia &I) { lds_take_inertia(lsl, s, lane, I); },
            hput);
        wave_lds_sync(); // the composite-inertia walk is done with the slot memory
    }
    for (int s = 0; s < a.n_slots * 6; ++s) lfs[s * WAVE + lane] = 0.0f;
    wave_lds_sync();
    // bias torques: RNEA with zero joint accelerations; rhs = f - nle, over f
    auto msave = [&](int s, const Motion &M) { lds_put_motion(lms, s, lane, M); };
    auto mload = [&](int s, Motion &M) { lds_get_motion(lms, s, lane, M); };
    auto fadd = [&](int s, const Force &F) { lds_add_force(lfs, s, lane, F); };
    auto ftake = [&](int s, Force &F) { lds_take_force(lfs, s, lane, F); };
    auto q_bias = [&](int d, float &x, float &v, float &acc) {
        x = qval(d);
        v = lqd[row + d];
        acc = 0.0f;
    };
    auto tau_bias = [&](int d, float v) { lf[row + d] -= v; };
    rnea_tree_walk_short<FD_SHORT_OPS>(a.prefix_end, first, last, ctl, rowf, flags, q_bias, tau_bias, msave, mload, fadd, ftake);
    ltdl_factor_acc(cnt, tri);
    ltdl_apply_acc(cnt, tri, lf + row + lo);
    __syncthreads();
    if (wave == 0)
        tile_store<0>(qdd + tc.b0 * n, tc.rows, n, magic_q, lf, lane, fast && (align & AL_TAU), tc.full && (align & AL_TAU));
}

// LDS bytes of that launch (0: does not apply — a segment is too long, or the hand does not fit)
static size_t fd_short_plan(const drm_walk *w, TreeArgs &a) {
    a = tree_args(w, false);
    if (a.max_seg_ops > FD_SHORT_OPS) return 0;
    const size_t shared = (size_t)table_lds_floats(a.n_ops) + 3 * (size_t)round4(WAVE * pad_odd(a.n));
    const size_t lds = sizeof(float) * layout_waves(a, shared, 0, a.n_slots * 18 * WAVE, [&](int s) {
        const int c = a.seg_dof_cnt[s];
        return round4(WAVE * pad_odd(c * (c + 1) / 2));
    });
    return lds <= (size_t)MAX_LDS_BYTES ? lds : 0;
}

// Every other robot: the articulated-body walk (drm_tree.hpp aba_tree_walk), one wavefront per segment, 64 samples per tile.
// What a sample needs between the sweeps (8 floats per link) would take a CU's LDS after two tiles of an arm with a hand, so
// those records live in HBM scratch, [op][8][64] per block: coalesced 256-byte accesses that the walk issues one op ahead of
// their use.  The grid is PERSISTENT — as many blocks as the chip holds at once, each looping over tiles — so the scratch is
// a few tens of MB whatever the batch and is re-read from L2 / Infinity Cache, never from HBM.
// LDS: [ table ][ q ][ qd ][ f -> qdd ] shared (staged per tile with coalesced 16-byte accesses), then per wavefront
//      [ slots : n_slots * (12 + 27) * 64 ]
constexpr int ABA_REC_FLOATS = 8;        // between the sweeps: the link velocity (6), then U (6), 1 / D, u
constexpr int ABA_BODY_FLOATS = 27;      // a branch point's accumulators: articulated inertia (21) + bias force (6)
constexpr int ABA_SLOT_FLOATS = 12 + ABA_BODY_FLOATS;

__device__ __forceinline__ void lds_add_body(float *slots, int s, unsigned lane, const ArtBody &a) {
    float *b = slots + s * (ABA_BODY_FLOATS * WAVE) + lane;
#pragma unroll
    for (int i = 0; i < 6; ++i) { b[(2 * i) * WAVE] += a.I.MA[i][0]; b[(2 * i + 1) * WAVE] += a.I.MA[i][1]; }
#pragma unroll
    for (int i = 0; i < 9; ++i) b[(12 + i) * WAVE] += a.I.B[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { b[(21 + 2 * i) * WAVE] += a.p.la[i][0]; b[(22 + 2 * i) * WAVE] += a.p.la[i][1]; }
}
__device__ __forceinline__ void lds_take_body(float *slots, int s, unsigned lane, ArtBody &a) {
    float *b = slots + s * (ABA_BODY_FLOATS * WAVE) + lane;
#pragma unroll
    for (int i = 0; i < 6; ++i) a.I.MA[i] += f2_make(b[(2 * i) * WAVE], b[(2 * i + 1) * WAVE]);
#pragma unroll
    for (int i = 0; i < 9; ++i) a.I.B[i] += b[(12 + i) * WAVE];
#pragma unroll
    for (int i = 0; i < 3; ++i) a.p.la[i] += f2_make(b[(21 + 2 * i) * WAVE], b[(22 + 2 * i) * WAVE]);
#pragma unroll
    for (int i = 0; i < ABA_BODY_FLOATS; ++i) b[i * WAVE] = 0.0f;
}

__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    forward_dynamics_aba_kernel(TreeArgs a, int flags, const float *__restrict__ q, const float *__restrict__ qd,
                                const float *__restrict__ f, int64_t B, int n_tiles, float *__restrict__ qdd, float *__restrict__ scratch,
                                uint32_t magic_q, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n), region = round4(WAVE * Sq);
    float *lq = smem + table_lds_floats(a.n_ops), *lqd = lq + region, *lf = lqd + region;
    const int first = a.seg_begin[wave], last = a.seg_begin[wave + 1];
    float *lms = smem + a.wave_off[wave], *lbs = lms + a.n_slots * (12 * WAVE);
    float *recs = scratch + ((int64_t)blockIdx.x * a.n_ops + first) * (ABA_REC_FLOATS * WAVE) + lane;
    const TableLds tab = stage_tree_table(a, smem);
    for (int s = 0; s < a.n_slots * ABA_BODY_FLOATS; ++s) lbs[s * WAVE + lane] = 0.0f; // (every take leaves its slot at zero again)
    const unsigned row = lane * Sq;
    auto rec = [&](int k) -> float * { return recs + (k - first) * (ABA_REC_FLOATS * WAVE); };

#pragma unroll 1
    for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
        __syncthreads(); // the table is staged / the previous tile's accelerations have left the f tile
        const int64_t b0 = (int64_t)tile * WAVE;
        const int rows = B - b0 < WAVE ? (int)(B - b0) : WAVE;
        const bool full = rows == WAVE, fast = full && (n & 1);
        if (wave == 0) tile_load<0>(q + b0 * n, rows, n, magic_q, lq, lane, fast && (align & AL_Q), full && (align & AL_Q));
        if (wave == (a.n_segments > 1 ? 1 : 0))
            tile_load<0>(qd + b0 * n, rows, n, magic_q, lqd, lane, fast && (align & AL_QD), full && (align & AL_QD));
        if (wave == (a.n_segments > 2 ? 2 : 0))
            tile_load<0>(f + b0 * n, rows, n, magic_q, lf, lane, fast && (align & AL_QDD), full && (align & AL_QDD));
        __syncthreads();
        // lanes past a partial tile walk a robot at rest with no torques (their LDS rows are never read or written)
        const bool live = (int)lane < rows;
        aba_tree_walk(
            a.prefix_end, first, last, tab, [&](int k) { return tab.row(k); }, flags,
            [&](int d, float &x, float &v) {
                x = live ? lq[row + d] : 0.0f;
                v = live ? lqd[row + d] : 0.0f;
            },
            [&](int d) -> float { return live ? lf[row + d] : 0.0f; },
            [&](int d, float v) {
                if (live) lf[row + d] = v;
            },
            [&](int k, const Motion &M) {
                float *b = rec(k);
#pragma unroll
                for (int i = 0; i < 3; ++i) { b[i * WAVE] = M.wa[i][0]; b[(3 + i) * WAVE] = M.va[i][0]; }
            },
            [&](int k, Motion &M) {
                const float *b = rec(k);
#pragma unroll
                for (int i = 0; i < 3; ++i) { M.wa[i] = f2_make(b[i * WAVE], 0.0f); M.va[i] = f2_make(b[(3 + i) * WAVE], 0.0f); }
            },
            [&](int k, const float *r) {
                float *b = rec(k);
#pragma unroll
                for (int i = 0; i < ABA_REC_FLOATS; ++i) b[i * WAVE] = r[i];
            },
            [&](int k, float *r) {
                const float *b = rec(k);
#pragma unroll
                for (int i = 0; i < ABA_REC_FLOATS; ++i) r[i] = b[i * WAVE];
            },
            [&](int s, const Motion &M) { lds_put_motion(lms, s, lane, M); }, [&](int s, Motion &M) { lds_get_motion(lms, s, lane, M); },
            [&](int s, const ArtBody &b) { lds_add_body(lbs, s, lane, b); }, [&](int s, ArtBody &b) { lds_take_body(lbs, s, lane, b); });
        __syncthreads();
        if (wave == 0) tile_store<0>(qdd + b0 * n, rows, n, magic_q, lf, lane, fast && (align & AL_TAU), full && (align & AL_TAU));
    }
}

// geometry of that launch: LDS bytes per block and the number of blocks the device holds at once (the persistent grid)
struct AbaPlan {
    TreeArgs a;
    size_t lds;
    int resident;
};
static int aba_plan(const drm_walk *w, AbaPlan &p) {
    // one wavefront per segment; all the segments through one wavefront when their save slots do not fit side by side
    for (int single = segments_worth_fanning_out(w) ? 0 : 1; single < 2; ++single) {
        p.a = tree_args(w, single != 0);
        const size_t shared = (size_t)table_lds_floats(p.a.n_ops) + 3 * (size_t)round4(WAVE * pad_odd(p.a.n));
        p.lds = sizeof(float) * layout_waves(p.a, shared, 0, p.a.n_slots * ABA_SLOT_FLOATS * WAVE, [](int) { return 0; });
        if (p.lds <= (size_t)MAX_LDS_BYTES || w->n_segments <= 1) break;
    }
    int rc = ensure_lds_tree(forward_dynamics_aba_kernel, p.lds);
    if (rc) return rc;
    return resident_blocks(forward_dynamics_aba_kernel, WAVE * p.a.n_segments, p.lds, p.resident);
}

// Serial-chain ("arm") specialisation, full tiles only: the chain forms of the two walks (drm_sample.hpp crba_chain,
// rnea_chain) with H's lower triangle and the right-hand side in REGISTERS and a fully unrolled L^T D L solve; constants
// staged once per wave in LDS, RNEA's body forces parked over the dead input tiles, qdd staged over them at the end.
// LINKS: the links the sweeps visit (NJ when the host folded the fixed tail into the last moving link, else CAP).
template <int CAP, int NJ, int LINKS>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    forward_dynamics_arm_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                                const float *__restrict__ f, int n_tiles, int flags, float *__restrict__ qdd) {
    static_assert(NJ & 1, "odd row widths only (linear LDS image)");
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, Q_FLOATS = round4(WAVE * NJ), F_FLOATS = (LINKS - DRM_RNEA_KEEP) * 6 * WAVE;
    static_assert(Q_FLOATS <= F_FLOATS, "the qdd tile fits under the parking area");
    constexpr int PER_WAVE = C_FLOATS + F_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[MAX_WAVES_PER_BLOCK * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = (int)blockIdx.x * MAX_WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * PER_WAVE;
    float *lq = lc + C_FLOATS;
    float *park = lq + lane; // body forces between RNEA's sweeps: [link][6][64]; qdd is staged over them at the end
    const int64_t b0 = (int64_t)tile * WAVE;

    // constant rows -> LDS (16 bytes per lane); every lane reads its own rows of q / qd / f straight into registers
    float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];
    float qv[NJ], qdv[NJ], rhs[NJ], zero[NJ], nle[NJ];
    {
        const int64_t r0 = (b0 + lane) * NJ;
#pragma unroll
        for (int d = 0; d < NJ; ++d) qv[d] = q[r0 + d];
#pragma unroll
        for (int d = 0; d < NJ; ++d) qdv[d] = qd[r0 + d];
#pragma unroll
        for (int d = 0; d < NJ; ++d) { rhs[d] = f[r0 + d]; zero[d] = 0.0f; }
    }
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    wave_lds_sync();
    auto row = [&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; };
    // bias torques first (RNEA with qdd = 0), H afterwards: the 28 floats of the triangle are not live across the RNEA
    // walk, whose own peak is what decides between one and two waves per SIMD; cos / sin are shared by the two walks
    float cs[NJ], sn[NJ];
    chain_trig<NJ>(qv, cs, sn);
    rnea_chain_trig<LINKS, NJ>(row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, cs, sn, qdv, zero, nle,
                             [&](int k, const Force &F) {
#pragma unroll
                                 for (int i = 0; i < 3; ++i) {
                                     park[(k * 6 + i) * WAVE] = F.la[i][0];
                                     park[(k * 6 + 3 + i) * WAVE] = F.la[i][1];
                                 }
                             },
                             [&](int k, Force &F) {
#pragma unroll
                                 for (int i = 0; i < 3; ++i) F.la[i] = f2_make(park[(k * 6 + i) * WAVE], park[(k * 6 + 3 + i) * WAVE]);
                             });
    float Ht[NJ * (NJ + 1) / 2];
    crba_chain_trig<LINKS, NJ>(row, cs, sn, [&](int i, int j, float v) {
        if (i >= j) Ht[tri_index(i, j)] = v;
    });
#pragma unroll
    for (int d = 0; d < NJ; ++d) rhs[d] -= nle[d];
    ltdl_solve_unrolled<NJ>(Ht, rhs);
    wave_lds_sync(); // every lane is done with the parking area before qdd is staged over it
#pragma unroll
    for (int d = 0; d < NJ; ++d) lq[lane * NJ + d] = rhs[d];
    wave_lds_sync();
    tile_store<NJ>(qdd + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
}

// A HAND (DRM_WALK_FINGERS: K serial chains of L revolute ops off the root): H is block diagonal, one L x L block per finger, and
// every finger is a short arm — so a finger takes the ARM kernel's three steps (bias torques by rnea_chain_trig with qdd = 0, its
// block of H by crba_chain_trig on the same cos / sin, the L^T D L solve in registers) on its own wavefront: a block of K
// wavefronts per 64-sample tile, per-lane accesses of the finger's L columns (16 bytes per sample and array at L = 4).
// The loop form (forward_dynamics_tree_kernel) issues 1 613 VALU + 964 SALU per finger and tile.
// LDS (static), per wavefront: [ table : L x 32 ][ parked body forces : L x 6 x 64 ]
template <int L>
__global__ void __launch_bounds__(WAVE * 4)
    forward_dynamics_fingers_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                                    const float *__restrict__ f, int n, int flags, float *__restrict__ qdd, int vec) {
    constexpr int C_FLOATS = L * DRM_OPF_STRIDE, P_FLOATS = L * 6 * WAVE; // (rnea_chain_trig parks every link's body force)
    __shared__ __attribute__((aligned(16))) float smem[4 * (C_FLOATS + P_FLOATS)];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * (C_FLOATS + P_FLOATS);
    float *park = lc + C_FLOATS + lane;
    if (lane < (unsigned)(L * (DRM_OPF_STRIDE / 4)))
        reinterpret_cast<float4 *>(lc)[lane] = reinterpret_cast<const float4 *>(ops_f + (size_t)wave * C_FLOATS)[lane];
    const int64_t r0 = ((int64_t)blockIdx.x * WAVE + lane) * n + wave * L;
    float qv[L], qdv[L], rhs[L], zero[L], nle[L];
    auto load = [&](const float *src, float (&dst)[L]) {
        if (L == 4 && vec) {
            const float4 a = *reinterpret_cast<const float4 *>(src + r0);
            dst[0] = a.x; dst[1] = a.y; dst[2 % L] = a.z; dst[3 % L] = a.w;
        } else {
#pragma unroll
            for (int d = 0; d < L; ++d) dst[d] = src[r0 + d];
        }
    };
    load(q, qv); load(qd, qdv); load(f, rhs);
#pragma unroll
    for (int d = 0; d < L; ++d) zero[d] = 0.0f;
    wave_lds_sync();
    auto row = [&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; };
    float cs[L], sn[L];
    chain_trig<L>(qv, cs, sn);
    rnea_chain_trig<L, L>(row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, cs, sn, qdv, zero, nle,
                          [&](int k, const Force &F) {
#pragma unroll
                              for (int i = 0; i < 3; ++i) { park[(k * 6 + i) * WAVE] = F.la[i][0]; park[(k * 6 + 3 + i) * WAVE] = F.la[i][1]; }
                          },
                          [&](int k, Force &F) {
#pragma unroll
                              for (int i = 0; i < 3; ++i) F.la[i] = f2_make(park[(k * 6 + i) * WAVE], park[(k * 6 + 3 + i) * WAVE]);
                          });
    float Ht[L * (L + 1) / 2];
    crba_chain_trig<L, L>(row, cs, sn, [&](int i, int j, float v) {
        if (i >= j) Ht[tri_index(i, j)] = v;
    });
#pragma unroll
    for (int d = 0; d < L; ++d) rhs[d] -= nle[d];
    ltdl_solve_unrolled<L>(Ht, rhs);
    if (L == 4 && vec) *reinterpret_cast<float4 *>(qdd + r0) = make_float4(rhs[0], rhs[1], rhs[2 % L], rhs[3 % L]);
    else {
#pragma unroll
        for (int d = 0; d < L; ++d) qdd[r0 + d] = rhs[d];
    }
}

// rows covered (full tiles), 0 = the call does not qualify
static int64_t launch_forward_dynamics_fingers(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int flags,
                                               float *qdd, hipStream_t s) {
#ifdef DRM_NO_FINGERS_KERNEL
    return 0;
#else
    if (!(w->shape & DRM_WALK_FINGERS) || B < WAVE || B / WAVE >= 0x7fffffffLL || (((uintptr_t)w->ops_f) & 15u) != 0) return 0;
    const int K = DRM_WALK_AH_K(w->shape), L = DRM_WALK_AH_L(w->shape), n = w->n_dofs;
    if (K * L != w->n_ops || n != w->n_ops || K < 2 || K > 4 || L < 2 || L > 4) return 0;
    const int n_tiles = (int)(B / WAVE);
    const int vec = (n % 4 == 0) && ((((uintptr_t)q | (uintptr_t)qd | (uintptr_t)f | (uintptr_t)qdd) & 15u) == 0);
#define X(l)                                                                                                                     \
    if (L == l)                                                                                                                  \
        hipLaunchKernelGGL((forward_dynamics_fingers_kernel<l>), dim3((unsigned)n_tiles), dim3(WAVE * K), 0, s, w->ops_f, q, qd, f, n, \
                           flags, qdd, vec);
    X(2) X(3) X(4)
#undef X
    return (int64_t)n_tiles * WAVE;
#endif
}

} // namespace drm

using namespace drm;

static int64_t drm_forward_dynamics_scratch_floats_impl(const drm_walk *w, int64_t B, bool aligned) {
    if (check_walk(w) || B <= 0 || !segments_ok(w)) return 0;
    // (full aligned tiles of these walks run straight-line kernels without scratch: sized for the ragged tail and for a misaligned
    // call, drm_common.hpp fast_path_scratch_tiles)
    if (w->special[DRM_SPECIAL_FD]) aligned = true;   // (the robot's own kernel takes any alignment: only a ragged tail is sized)
    const bool fast = ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && w->n_dofs == 7) || arm_hand_compiled(w) ||
                      w->special[DRM_SPECIAL_FD] != nullptr;
    TreeArgs a;
    if (fd_short_plan(w, a)) return 0;
    AbaPlan p;
    if (aba_plan(w, p)) return 0;
    const int64_t tiles = fast ? (aligned ? (B % WAVE ? 1 : 0) : fast_path_scratch_tiles(B)) : (B + WAVE - 1) / WAVE;
    return (tiles < p.resident ? tiles : (int64_t)p.resident) * p.a.n_ops * ABA_REC_FLOATS * WAVE;
}
extern "C" int64_t drm_forward_dynamics_scratch_floats(const drm_walk *w, int64_t B) { return drm_forward_dynamics_scratch_floats_impl(w, B, false); }
// ... for a caller that GUARANTEES 16-byte aligned q / qd / qdd (f) / outputs (both Python bindings do: they clone a misaligned
// slice): the full tiles of a 7-DoF arm / an arm with a hand then run straight-line kernels that need no scratch — only a ragged
// tail's one tile is sized
extern "C" int64_t drm_forward_dynamics_scratch_floats_aligned(const drm_walk *w, int64_t B) { return drm_forward_dynamics_scratch_floats_impl(w, B, true); }

extern "C" int drm_forward_dynamics(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B,
                                    int32_t flags, float *qdd, float *scratch, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !qd || !f || !qdd) return fail(DRM_ERR_INVALID, "q / qd / f / qdd must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs;
    if (w->special[DRM_SPECIAL_FD] && B >= WAVE && B / WAVE < 0x7fffffffLL && (((uintptr_t)w->ops_f) & 15u) == 0) {
        // the robot's own straight-line articulated-body kernel (csrc/drm_static.hpp aba_static_walk, built for exactly this walk):
        // full tiles, any pointer alignment, no scratch
        int n_tiles = (int)(B / WAVE), fl = (int)flags;
        uint32_t magic = div_magic(n), al = al16(q, AL_Q) | al16(qd, AL_QD) | al16(f, AL_QDD) | al16(qdd, AL_TAU);
        void *args[] = {(void *)&w->ops_f, (void *)&q, (void *)&qd, (void *)&f, (void *)&n_tiles, (void *)&fl, (void *)&qdd, (void *)&magic, (void *)&al};
        int grid = 0; // persistent wavefronts: what the device holds at once
        int rcg = resident_blocks_module((hipFunction_t)w->special[DRM_SPECIAL_FD], WAVE, grid);
        if (rcg) return rcg;
        if (grid > n_tiles || w->n_ops < STATIC_LONE_OPS) grid = n_tiles; // (small robots: one tile per block, drm_common.hpp STATIC_LONE_OPS)
        hipError_t e = hipModuleLaunchKernel((hipFunction_t)w->special[DRM_SPECIAL_FD], (unsigned)grid, 1, 1, WAVE, 1, 1, 0, (hipStream_t)stream, args, nullptr);
        if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_fd_static): %s", hipGetErrorString(e));
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done == B) return DRM_OK;
        drm_walk rest = *w;
        rest.special[DRM_SPECIAL_FD] = nullptr;
        return drm_forward_dynamics(&rest, q + done * n, qd + done * n, f + done * n, B - done, flags, qdd + done * n, scratch, stream);
    }
    {   // a hand (fingers off the root): full tiles through the per-finger arm form
        const int64_t done = launch_forward_dynamics_fingers(w, q, qd, f, B, (int)flags, qdd, (hipStream_t)stream);
        if (done > 0) {
            rc = launched();
            if (rc || done == B) return rc;
            drm_walk generic = *w;
            generic.shape &= ~DRM_WALK_FINGERS;
            return drm_forward_dynamics(&generic, q + done * n, qd + done * n, f + done * n, B - done, flags, qdd + done * n, scratch, stream);
        }
    }
#ifndef DRM_NO_ARM_KERNEL
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7 && B >= WAVE && B / WAVE < 0x7fffffffLL &&
        (((uintptr_t)q | (uintptr_t)qd | (uintptr_t)f | (uintptr_t)qdd | (uintptr_t)w->ops_f) & 15u) == 0) {
        // 7-DoF arms: full tiles through the register-resident chain kernel, ragged tail through the generic one
        int n_tiles = (int)(B / WAVE);
        if (w->special[DRM_SPECIAL_FD_ARM2] && w->special[DRM_SPECIAL_FD_ARM] && n_tiles / 2 >= DRM_ARM_STATIC_MIN_PAIRS) {
            // ABI 11: two samples per lane for the pairs of tiles of a large launch (issue-bound: a third fewer instructions per
            // sample); an odd last tile and the ragged tail follow through the forms below
            int n_pairs = n_tiles / 2, fl = (int)flags;
            void *args[] = {(void *)&q, (void *)&qd, (void *)&f, (void *)&n_pairs, (void *)&fl, (void *)&qdd};
            hipError_t e = hipModuleLaunchKernel((hipFunction_t)w->special[DRM_SPECIAL_FD_ARM2], (unsigned)n_pairs, 1, 1, WAVE, 1, 1, 0, (hipStream_t)stream, args, nullptr);
            if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_fd_arm2_static): %s", hipGetErrorString(e));
            const int64_t done2 = (int64_t)n_pairs * 2 * WAVE;
            if (done2 == B) return launched();
            rc = launched();
            if (rc) return rc;
            drm_walk rest = *w;
            rest.special[DRM_SPECIAL_FD_ARM2] = nullptr;
            return drm_forward_dynamics(&rest, q + done2 * n, qd + done2 * n, f + done2 * n, B - done2, flags, qdd + done2 * n, scratch, stream);
        }
        const dim3 grid((unsigned)((n_tiles + MAX_WAVES_PER_BLOCK - 1) / MAX_WAVES_PER_BLOCK)), block(WAVE * MAX_WAVES_PER_BLOCK);
        if (w->special[DRM_SPECIAL_FD_ARM]) {
            // this arm's own kernel, its constants folded into the instruction stream (csrc/drm_arm_static.hpp, specialize.py)
            int fl = (int)flags;
            void *args[] = {(void *)&q, (void *)&qd, (void *)&f, (void *)&n_tiles, (void *)&fl, (void *)&qdd};
            hipError_t e = hipModuleLaunchKernel((hipFunction_t)w->special[DRM_SPECIAL_FD_ARM], (unsigned)n_tiles, 1, 1, WAVE, 1, 1, 0, (hipStream_t)stream, args, nullptr);
            if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_fd_arm_static): %s", hipGetErrorString(e));
        } else if (arm_links(w) == 7)
            hipLaunchKernelGGL((forward_dynamics_arm_kernel<8, 7, 7>), grid, block, 0, (hipStream_t)stream, w->ops_f, q, qd, f,
                               n_tiles, (int)flags, qdd);
        else
            hipLaunchKernelGGL((forward_dynamics_arm_kernel<8, 7, 8>), grid, block, 0, (hipStream_t)stream, w->ops_f, q, qd, f,
                               n_tiles, (int)flags, qdd);
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done == B) return launched();
        rc = launched();
        if (rc) return rc;
        drm_walk generic = *w;
        generic.shape &= ~DRM_WALK_ARM_CHAIN;
        return drm_forward_dynamics(&generic, q + done * n, qd + done * n, f + done * n, B - done, flags, qdd + done * n,
                                    scratch, stream);
    }
#endif
    {   // an arm that carries a hand (Panda with gripper, Jaco, iiwa7 + Allegro): full tiles through the straight-line kernel
        const int64_t done = launch_forward_dynamics_arm_hand(w, q, qd, f, B, (int)flags, qdd, (hipStream_t)stream);
        if (done > 0) {
            rc = launched();
            if (rc || done == B) return rc;
            drm_walk generic = *w;
            generic.shape &= ~DRM_WALK_ARM_HAND;
            return drm_forward_dynamics(&generic, q + done * n, qd + done * n, f + done * n, B - done, flags, qdd + done * n, scratch,
                                        stream);
        }
    }
    // (as in drm_rnea: a misaligned call on a walk with a straight-line kernel, or its ragged tail, runs the loop kernel on at most
    // MISALIGNED_TILES blocks — what its scratch is sized for)
    const bool fast_walk = (((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7)) || arm_hand_compiled(w);
    if (!segments_ok(w)) return fail(DRM_ERR_INVALID, "walk segments are inconsistent");
    if ((((uintptr_t)w->ops_f) & 15u) != 0) return fail(DRM_ERR_INVALID, "ops_f must be 16-byte aligned");
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    hipStream_t s = (hipStream_t)stream;
    TreeArgs fingers;
    if (const size_t lds = fd_short_plan(w, fingers)) {
        const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(f, AL_QDD) | al16(qdd, AL_TAU);
        rc = ensure_lds_tree(forward_dynamics_tree_kernel, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(forward_dynamics_tree_kernel, dim3((unsigned)tiles), dim3(WAVE * fingers.n_segments), lds, s, fingers, (int)flags,
                           q, qd, f, B, qdd, div_magic(n), align);
        return launched();
    }
    AbaPlan p;
    rc = aba_plan(w, p);
    if (rc) return rc;
    if (!scratch)
        return fail(DRM_ERR_INVALID, "this robot runs the articulated-body kernel: pass drm_forward_dynamics_scratch_floats() floats of scratch");
    const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(f, AL_QDD) | al16(qdd, AL_TAU);
    int64_t grid = tiles < p.resident ? tiles : (int64_t)p.resident;
    if (fast_walk && grid > MISALIGNED_TILES) grid = MISALIGNED_TILES;
    hipLaunchKernelGGL(forward_dynamics_aba_kernel, dim3((unsigned)grid), dim3(WAVE * p.a.n_segments), p.lds, s, p.a, (int)flags, q, qd, f, B,
                       (int)tiles, qdd, scratch, div_magic(n), align);
    return launched();
}
