// drm_forward_dynamics.hip — K8: joint accelerations from joint torques, qdd = H(q)^-1 (f - nle(q, qd)).
//
// Replaces DifferentiableRobotModel.compute_forward_dynamics (robot_model.py:487-624, Featherstone's
// articulated-body algorithm written as three Python loops over the links with 6x6 bmm's per link).  The
// articulated-body recursion is an O(n) elimination of the linear system H qdd = f - nle; this kernel forms the same
// system with the two walks it already has — the composite-rigid-body walk for H, RNEA with qdd = 0 for the bias torques
// nle — and solves it per sample by an L^T D L factorisation taken from the leaves to the root, the elimination order of
// the articulated-body recursion (~n^3/3 FMAs, less than one of the walks).  Same result up to fp32 rounding amplified by
// the conditioning of the sub-trees, like the reference's own recursion (tolerances in tests/).
//
// Per sample: in q, qd, f [n] (12 n bytes), out qdd [n] (4 n bytes).          n = 7: 112 B
// 7-DoF arm chains run forward_dynamics_arm_kernel below; every other robot the loop-structured kernel.
#include "drm_common.hpp"
#include "drm_sample.hpp"
#include "drm_tree_dev.hpp"

namespace drm {

constexpr int FD_SHORT_OPS = 6; // segments of up to this many ops in a row take crba_tree_walk_short
constexpr int FD_FORCE_FLOATS = 6; // what the RNEA walks of this kernel park per op: the body force

// Loop-structured forward dynamics of any robot: one tile of 64 samples per block, one wavefront per segment of the walk.
// Segments are independent (their joints share no link that moves), so H is block diagonal: every wavefront forms ITS
// block (drm_tree.hpp crba_tree_walk, packed lower triangle with segment-local DoF indices), the bias torques of its
// joints (rnea_tree_walk with qdd = 0, robot_model.py:377-400) and solves its block by the leaf-to-root L^T D L
// factorisation (drm_sample.hpp ltdl_solve) — an Allegro hand is four 4 x 4 systems per sample, not one 16 x 16.
// LDS: [ table ][ q ][ qd ][ f -> rhs -> qdd ][ residual of the refinement step ] shared, then per wavefront
//      [ records : max_seg_ops * 9 * 64 (RNEA; CRBA's cos / sin / value first) ][ slots : n_slots * 18 * 64 ]
//      [ triangle : 64 (nt|1), nt = largest block's n (n + 1) / 2 — unless HBM ]
// HBM: the triangle lives in caller-provided scratch, [tile][segment][entry][64] (robots beyond ~30 DoF per segment).
template <bool HBM>
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    forward_dynamics_tree_kernel(TreeArgs a, int flags, int nt_max, const float *__restrict__ q, const float *__restrict__ qd,
                                 const float *__restrict__ f, int64_t B, float *__restrict__ qdd, float *__restrict__ scratch,
                                 uint32_t magic_q, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, Sq = pad_odd(n), region = round4(WAVE * Sq);
    // Every segment short (fingers): the four tiles of round 1, shared by the wavefronts.  Otherwise (an arm carrying a gripper
    // or a hand: LDS per sample is what bounds the wavefronts per CU) q and qd are read straight from global memory and only
    // the f -> rhs -> qdd tile and the residual tile are staged.
    const bool short_segments = a.max_seg_ops <= FD_SHORT_OPS;
    float *lf = smem + table_lds_floats(a.n_ops), *lres = lf + region, *lq = lres + region, *lqd = lq + region;
    const int first = a.seg_begin[wave], last = a.seg_begin[wave + 1];
    const int lo = a.seg_dof_lo[wave], cnt = a.seg_dof_cnt[wave], nt = cnt * (cnt + 1) / 2;
    float *park = smem + a.wave_off[wave];
    // per-op records: the body force of the RNEA walk (6 floats; cos / sin are recomputed on the way back), or just CRBA's 3
    // when every segment is short (the RNEA walk then keeps its own in registers)
    float *lsl = park + (last - first) * ((short_segments ? CRBA_PARK_FLOATS : FD_FORCE_FLOATS) * WAVE); // slots: inertia [slot][10][64], then motion [12] + force [6]
    float *lms = lsl, *lfs = lsl + a.n_slots * (12 * WAVE);
    float *ltri = lsl + a.n_slots * (18 * WAVE) + lane * pad_odd(nt); // this lane's packed triangle (LDS form)
    float *gtri = HBM ? scratch + ((int64_t)blockIdx.x * a.n_segments + wave) * (int64_t)nt_max * WAVE + lane : nullptr;
    auto tri = [&](int i) -> float & { return HBM ? gtri[(int64_t)i * WAVE] : ltri[i]; };

    const TableLds tab = stage_tree_table(a, smem);
    const bool fast = tc.full && (n & 1);
    if (short_segments) {
        if (wave == 0) tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, fast && (align & AL_Q), tc.full && (align & AL_Q));
        if (wave == (a.n_segments > 1 ? 1 : 0))
            tile_load<0>(qd + tc.b0 * n, tc.rows, n, magic_q, lqd, lane, fast && (align & AL_QD), tc.full && (align & AL_QD));
    }
    if (wave == (a.n_segments > 2 ? 2 : 0))
        tile_load<0>(f + tc.b0 * n, tc.rows, n, magic_q, lf, lane, fast && (align & AL_QDD), tc.full && (align & AL_QDD));
    for (int s = 0; s < a.n_slots * 10; ++s) lsl[s * WAVE + lane] = 0.0f;
    for (int i = 0; i < nt; ++i) tri(i) = 0.0f; // pairs of joints on different branches
    __syncthreads();

    // lanes past a partial tile read zeros (see drm_fk.hip); their H is then a valid inertia matrix as well
    const bool live = (int)lane < tc.rows;
    const unsigned row = lane * Sq;
    const int64_t grow = (tc.b0 + (live ? (int64_t)lane : 0)) * n;
    auto q_at = [&](int d) -> float { return !live ? 0.0f : short_segments ? lq[row + d] : q[grow + d]; };
    auto qd_at = [&](int d) -> float { return short_segments ? lqd[row + d] : (live ? qd[grow + d] : 0.0f); };
    const TableLds &ctl = tab;
    auto rowf = [&](int k) { return tab.row(k); };
    auto hput = [&](int di, int dj, float v) {
        if (di >= dj) tri(tri_index(di - lo, dj - lo)) = v;
    };
    auto qval = [&](int d) -> float { return q_at(d); };
    // a short serial segment (a finger): the unrolled walk with the joint transforms in registers; anything else: the loop
    if (!crba_tree_walk_short<FD_SHORT_OPS>(first, last, ctl, rowf, qval, hput)) {
        crba_prepare(first, last, ctl, qval, [&](int k, float c, float s, float x) {
            float *b = park + (k - first) * (CRBA_PARK_FLOATS * WAVE) + lane;
            b[0] = c; b[WAVE] = s; b[2 * WAVE] = x;
        });
        crba_tree_walk(
            first, last, ctl, rowf,
            [&](int k, float &c, float &s, float &x) {
                const float *b = park + (k - first) * (CRBA_PARK_FLOATS * WAVE) + lane;
                c = b[0]; s = b[WAVE]; x = b[2 * WAVE];
            },
            [&](int s, const Inertia &I) { lds_add_inertia(lsl, s, lane, I); }, [&](int s, Inertia &I) { lds_take_inertia(lsl, s, lane, I); },
            hput);
    }
    wave_lds_sync(); // the composite-inertia walk is done with the slot memory
    for (int s = 0; s < a.n_slots * 6; ++s) lfs[s * WAVE + lane] = 0.0f;
    wave_lds_sync();
    // bias torques: RNEA with zero joint accelerations; rhs = f - nle, over f.  Short segments (the fingers of a hand) take the
    // unrolled walk that keeps its per-op records in registers (drm_tree.hpp rnea_tree_walk_short), as drm_rnea does.
    auto park_f = [&](int k, const Force &F, float, float, float) {
        float *b = park + (k - first) * (FD_FORCE_FLOATS * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) { b[i * WAVE] = F.la[i][0]; b[(3 + i) * WAVE] = F.la[i][1]; }
    };
    auto unpark_f = [&](int k, Force &F, float &c, float &s, float &x) {
        const float *b = park + (k - first) * (FD_FORCE_FLOATS * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) F.la[i] = f2_make(b[i * WAVE], b[(3 + i) * WAVE]);
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        x = 0.0f; c = 1.0f; s = 0.0f;
        if (ct.dof >= 0) {
            x = q_at(ct.dof);
            if (!ct.prismatic) sincos_one(x, s, c);
        }
    };
    auto msave = [&](int s, const Motion &M) { lds_put_motion(lms, s, lane, M); };
    auto mload = [&](int s, Motion &M) { lds_get_motion(lms, s, lane, M); };
    auto fadd = [&](int s, const Force &F) { lds_add_force(lfs, s, lane, F); };
    auto ftake = [&](int s, Force &F) { lds_take_force(lfs, s, lane, F); };
    auto q_bias = [&](int d, float &x, float &v, float &acc) {
        x = q_at(d);
        v = qd_at(d);
        acc = 0.0f;
    };
    auto tau_bias = [&](int d, float v) { lf[row + d] -= v; };
    if (short_segments) rnea_tree_walk_short<FD_SHORT_OPS>(a.prefix_end, first, last, ctl, rowf, flags, q_bias, tau_bias, msave, mload, fadd, ftake);
    else rnea_tree_walk(a.prefix_end, first, last, ctl, rowf, flags, q_bias, tau_bias, park_f, unpark_f, msave, mload, fadd, ftake);
    ltdl_factor_acc(cnt, tri);
    ltdl_apply_acc(cnt, tri, lf + row + lo);
    if (flags & DRM_FD_REFINE) {
        // one step of iterative refinement: r = f - ID(q, qd, x0) by the inverse-dynamics walk (it never forms H, so its
        // rounding errors are those of the torques, not cond(H) times them), x1 = x0 + H^-1 r with the factors at hand;
        // the residual has a tile of its own (lres).
        wave_lds_sync();
        for (int s = 0; s < a.n_slots * 6; ++s) lfs[s * WAVE + lane] = 0.0f;
        wave_lds_sync();
        float *lr = lres + row; // residual tile (same layout as lf)
        auto q_res = [&](int d, float &x, float &v, float &acc) {
            x = q_at(d);
            v = qd_at(d);
            acc = (d >= lo && d < lo + cnt) ? lf[row + d] : 0.0f; // (prefix ops carry no DoF: every DoF read is this segment's)
        };
        auto tau_res = [&](int d, float v) { lr[d] = (live ? f[(tc.b0 + lane) * n + d] : 0.0f) - v; };
        if (short_segments) rnea_tree_walk_short<FD_SHORT_OPS>(a.prefix_end, first, last, ctl, rowf, flags, q_res, tau_res, msave, mload, fadd, ftake);
        else rnea_tree_walk(a.prefix_end, first, last, ctl, rowf, flags, q_res, tau_res, park_f, unpark_f, msave, mload, fadd, ftake);
        ltdl_apply_acc(cnt, tri, lr + lo);
        for (int d = lo; d < lo + cnt; ++d) lf[row + d] += lr[d];
    }
    __syncthreads();
    if (wave == 0)
        tile_store<0>(qdd + tc.b0 * n, tc.rows, n, magic_q, lf, lane, fast && (align & AL_TAU), tc.full && (align & AL_TAU));
}

// geometry of a launch: LDS bytes with the triangle in LDS, or (hbm = true) in scratch
struct FdPlan {
    TreeArgs a;
    int nt_max;
    bool hbm;
    size_t lds;
};
static FdPlan fd_plan(const drm_walk *w) {
    FdPlan p;
    auto lay = [&](bool single, bool hbm) {
        p.a = tree_args(w, single);
        p.nt_max = 1;
        for (int s = 0; s < p.a.n_segments; ++s) {
            const int c = p.a.seg_dof_cnt[s], nt = c * (c + 1) / 2;
            if (nt > p.nt_max) p.nt_max = nt;
        }
        const TreeArgs &a = p.a;
        const bool shorts = a.max_seg_ops <= FD_SHORT_OPS;
        // f -> qdd and the residual (+ q and qd when every segment is short: see the kernel)
        const size_t shared = (size_t)table_lds_floats(a.n_ops) + (shorts ? 4 : 2) * (size_t)round4(WAVE * pad_odd(a.n));
        p.hbm = hbm;
        const int per_op = (shorts ? CRBA_PARK_FLOATS : FD_FORCE_FLOATS) * WAVE;
        p.lds = sizeof(float) * layout_waves(p.a, shared, per_op, a.n_slots * 18 * WAVE, [&](int s) {
            const int c = a.seg_dof_cnt[s];
            return hbm ? 0 : round4(WAVE * pad_odd(c * (c + 1) / 2));
        });
        return p.lds <= (size_t)MAX_LDS_BYTES;
    };
    // in order of preference: fanned out with the triangles in LDS, one wavefront with its triangle in LDS, triangles in HBM
    if (lay(false, false)) return p;
    if (w->n_segments > 1 && lay(true, false)) return p;
    if (lay(false, true)) return p;
    lay(true, true);
    return p;
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
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, Q_FLOATS = round4(WAVE * NJ), F_FLOATS = CAP * 6 * WAVE;
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

} // namespace drm

using namespace drm;

extern "C" int64_t drm_forward_dynamics_scratch_floats(const drm_walk *w, int64_t B) {
    if (check_walk(w) || B <= 0 || !segments_ok(w)) return 0;
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && w->n_dofs == 7) return 0;
    const FdPlan p = fd_plan(w);
    if (!p.hbm) return 0;
    return ((B + WAVE - 1) / WAVE) * (int64_t)p.a.n_segments * p.nt_max * WAVE;
}

extern "C" int drm_forward_dynamics(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B,
                                    int32_t flags, float *qdd, float *scratch, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !qd || !f || !qdd) return fail(DRM_ERR_INVALID, "q / qd / f / qdd must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs;
#ifndef DRM_NO_ARM_KERNEL
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7 && B >= WAVE && B / WAVE < 0x7fffffffLL &&
        (((uintptr_t)q | (uintptr_t)qd | (uintptr_t)f | (uintptr_t)qdd | (uintptr_t)w->ops_f) & 15u) == 0) {
        // 7-DoF arms: full tiles through the register-resident chain kernel, ragged tail through the generic one
        const int n_tiles = (int)(B / WAVE);
        const dim3 grid((unsigned)((n_tiles + MAX_WAVES_PER_BLOCK - 1) / MAX_WAVES_PER_BLOCK)), block(WAVE * MAX_WAVES_PER_BLOCK);
        if (arm_links(w) == 7)
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
    if (!segments_ok(w)) return fail(DRM_ERR_INVALID, "walk segments are inconsistent");
    if ((((uintptr_t)w->ops_f) & 15u) != 0) return fail(DRM_ERR_INVALID, "ops_f must be 16-byte aligned");
    const FdPlan p = fd_plan(w);
    if (p.hbm && !scratch)
        return fail(DRM_ERR_INVALID, "this robot's inertia matrix does not fit in LDS: pass drm_forward_dynamics_scratch_floats() floats of scratch");
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    const uint32_t align = al16(q, AL_Q) | al16(qd, AL_QD) | al16(f, AL_QDD) | al16(qdd, AL_TAU);
    hipStream_t s = (hipStream_t)stream;
    if (p.hbm) {
        rc = ensure_lds_tree(forward_dynamics_tree_kernel<true>, p.lds);
        if (rc) return rc;
        hipLaunchKernelGGL(forward_dynamics_tree_kernel<true>, dim3((unsigned)tiles), dim3(WAVE * p.a.n_segments), p.lds, s, p.a, (int)flags,
                           p.nt_max, q, qd, f, B, qdd, scratch, div_magic(n), align);
    } else {
        rc = ensure_lds_tree(forward_dynamics_tree_kernel<false>, p.lds);
        if (rc) return rc;
        hipLaunchKernelGGL(forward_dynamics_tree_kernel<false>, dim3((unsigned)tiles), dim3(WAVE * p.a.n_segments), p.lds, s, p.a, (int)flags,
                           p.nt_max, q, qd, f, B, qdd, scratch, div_magic(n), align);
    }
    return launched();
}
