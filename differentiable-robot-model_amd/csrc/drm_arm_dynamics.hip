// drm_arm_dynamics.hip — the serial-chain ("arm") dynamics kernels: K3 RNEA and the fused FK + RNEA launch of BASELINE
// configuration 3.  Same design and build flags as drm_arm_kernels.hip (kernel-argument preload) except that this unit,
// like every dynamics unit, is compiled with -fno-slp-vectorize: the SLP vectoriser packs the scalar cross products of
// the body force into v_pk_* ops whose operand pairs it then has to assemble with v_mov (4 moves per packed FMA) —
// 1 608 -> 1 465 VALU per wave here, 50.5 -> 46.6 us at 2^20 (the FK + Jacobian kernel, whose pairs are laid out by hand,
// is 6 % shorter WITH it and stays in drm_arm_kernels.hip).
#include "drm_common.hpp"
#include "drm_sample.hpp"
#include "drm_arm_stream.hpp"

namespace drm {

// ---------------------------------------------------------------------------------------------------
// Serial-chain ("arm") specialisation, full tiles only (DRM_WALK_ARM_CHAIN walks, 16-byte aligned pointers,
// NJ odd): the same design as fk_jacobian_arm_kernel — constant rows staged once per wave in LDS (one 16-byte
// load per lane brings the whole 1 KB table) and read back as broadcast ds_reads, packed-FP32 sweeps
// (drm_sample.hpp rnea_chain), preloaded kernel arguments, one basic block; the per-link body forces are parked
// in LDS between the two sweeps (registers are what limits occupancy here).
// ---------------------------------------------------------------------------------------------------
// LINKS = the links the dynamics sweeps visit: CAP, or NJ when the walk has no ops behind its moving joints (the host
// folded the fixed tail into the last moving link, flatten.fold_link_table; the rest of the table is identity padding).
#ifdef DRM_RNEA_WAVES_FORCE /* development: tools/build_variants.sh */
#define DRM_RNEA_WAVES(LINKS, CAP) DRM_RNEA_WAVES_FORCE
#else
#define DRM_RNEA_WAVES(LINKS, CAP) ((LINKS) < (CAP) ? 3 : 2)
#endif
// LAT ("latency form"): this kernel only ever runs launches of at most 1 024 tiles (beyond that rnea_arm2_kernel takes over),
// i.e. ONE wave per SIMD, the whole register file to itself and nobody to hide its LDS round trips behind.  Op by op the
// forward sweep waited three times per link for constants it had just asked for (6 cycles per instruction instead of 4,
// tools/timeline.py: forward sweep 2.0 us for ~850 VALU).  LAT = the constants prefetched one link ahead (rnea_chain_trig's PREF)
// and every body force kept in registers (KEEP = LINKS): the sweeps wait for no LDS read and park nothing.
template <int CAP, int NJ, int LINKS, bool LAT = false>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(LAT ? 1 : DRM_RNEA_WAVES(LINKS, CAP), LAT ? 2 : DRM_RNEA_WAVES(LINKS, CAP))))
    rnea_arm_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                    const float *__restrict__ qdd, int n_tiles, int flags, float *__restrict__ tau) {
    static_assert(NJ & 1, "odd row widths only (linear LDS image)");
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    // body-force parking area between the sweeps (tau is staged over it at the end), one [6][64] record per PARKED link:
    // LINKS - DRM_RNEA_KEEP of them (the last links' forces stay in registers, drm_sample.hpp rnea_chain_trig)
    constexpr int KEEP = LAT ? LINKS : DRM_RNEA_KEEP;
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, Q_FLOATS = round4(WAVE * NJ),
                  F_FLOATS = (LINKS - KEEP) * 6 * WAVE > Q_FLOATS ? (LINKS - KEEP) * 6 * WAVE : Q_FLOATS; // (tau is staged over the parking area)
    constexpr int PER_WAVE = C_FLOATS + F_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[MAX_WAVES_PER_BLOCK * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = (int)blockIdx.x * MAX_WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * PER_WAVE;
    float *lq = lc + C_FLOATS;
    float *lf = lq + lane; // body forces between the sweeps: [link][6][64]
    const int64_t b0 = (int64_t)tile * WAVE;
    DRM_STAMP(0);

    // constant rows -> LDS (16 bytes per lane); every lane reads its own rows of q / qd / qdd straight into registers
    float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];
    float qv[NJ], qdv[NJ], qddv[NJ], tv[NJ];
    {
        const int64_t row = (b0 + lane) * NJ;
#pragma unroll
        for (int d = 0; d < NJ; ++d) qv[d] = q[row + d];
#pragma unroll
        for (int d = 0; d < NJ; ++d) qdv[d] = qd[row + d];
#pragma unroll
        for (int d = 0; d < NJ; ++d) qddv[d] = qdd ? qdd[row + d] : 0.0f;
    }
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    wave_lds_sync();
    DRM_STAMP_DRAINED(1);
    rnea_chain<LINKS, NJ, KEEP, LAT>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, flags & DRM_RNEA_GRAVITY,
                        flags & DRM_RNEA_DAMPING, qv, qdv, qddv, tv,
                        [&](int k, const Force &F) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                lf[(k * 6 + i) * WAVE] = F.la[i][0];
                                lf[(k * 6 + 3 + i) * WAVE] = F.la[i][1];
                            }
                        },
                        [&](int k, Force &F) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) F.la[i] = f2_make(lf[(k * 6 + i) * WAVE], lf[(k * 6 + 3 + i) * WAVE]);
                        });
    DRM_STAMP(3);
    wave_lds_sync(); // every lane is done with the parking area before tau is staged over it
#pragma unroll
    for (int d = 0; d < NJ; ++d) lq[lane * NJ + d] = tv[d];
    wave_lds_sync();
    tile_store<NJ>(tau + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
    DRM_STAMP(4);
    DRM_STAMP_DRAINED(5);
}


// ---------------------------------------------------------------------------------------------------
// The same launch with TWO SAMPLES PER LANE (drm_sample.hpp rnea_chain2_trig): a wavefront owns a tile of 128 consecutive
// samples, lane l the rows l and l + 64 of it.  Every instruction of the recursion is a full packed-FP32 op; a third
// fewer instructions per sample than rnea_arm_kernel.  One wavefront per block (the per-wave LDS is what bounds the
// number of resident waves: constants 1 KB + (LINKS - 1) parked forces x 3 KB).
//   q / qd / qdd: row l of the tile into the low halves, row l + 64 into the high halves of the register pairs (dword loads
//   that share cache lines across the wave, as in the one-sample kernels); tau leaves through a [128][NJ] LDS image.
// ---------------------------------------------------------------------------------------------------
constexpr int TILE2 = 2 * WAVE;
// Which form a launch takes.  Up to 1 024 64-sample tiles the one-sample kernels put one wave on every SIMD of the chip
// (256 CUs x 4) and nothing is gained by halving the number of waves (65 536 samples: 5.3 us against 5.8 us, fused 6.2
// against 7.3); beyond that the SIMDs hold several waves and the kernel with fewer instructions per sample wins
// (131 072: 6.4 against 7.5 us, fused 7.7 against 8.5; 2^20: 32 against 38.5 us, fused 41 against 48.5; profiles/r03_ab_rnea_two_samples.txt).
#ifndef DRM_TWO_SAMPLE_MIN_TILES
#define DRM_TWO_SAMPLE_MIN_TILES 1024
#endif
constexpr int TWO_SAMPLE_MIN_TILES = DRM_TWO_SAMPLE_MIN_TILES;
#ifndef DRM_LAT2_MAX_TILES
#define DRM_LAT2_MAX_TILES 1024 /* pairs of tiles: up to one two-sample wave per SIMD */
#endif
constexpr int LAT2_MAX_TILES = DRM_LAT2_MAX_TILES;
// (round 5) launches of at least this many pairs of tiles take the streaming form, arm2_stream_kernel below
#ifndef DRM_PIPE_MIN_TILES
#define DRM_PIPE_MIN_TILES 0x7fffffff /* off: measured 42.9 us against 41.0 us of the three-waves-per-SIMD form at 2^20 rows (the launch is VALU-issue bound, profiles/r05_ab_stream.txt) */
#endif
constexpr int PIPE_MIN_TILES = DRM_PIPE_MIN_TILES;
#ifndef DRM_PIPE_WAVES
#define DRM_PIPE_WAVES 2
#endif
#ifdef DRM_RNEA_NO_LAT
constexpr bool LAT1 = false;
#else
constexpr bool LAT1 = true; // the one-sample kernels never see more than 1 024 tiles
#endif
#ifndef DRM_RNEA2_WAVES
#define DRM_RNEA2_WAVES 3
#endif
// LAT: launches of at most 1 024 pairs of tiles (131 072 rows: one wave per SIMD) — constants prefetched one link ahead, no force
// parked (see rnea_arm_kernel)
template <int CAP, int NJ, int LINKS, bool LAT = false>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(LAT ? 1 : DRM_RNEA2_WAVES, LAT ? 2 : DRM_RNEA2_WAVES))) rnea_arm2_kernel(const float *__restrict__ ops_f, const float *__restrict__ q,
                                                         const float *__restrict__ qd, const float *__restrict__ qdd, int n_tiles,
                                                         int flags, float *__restrict__ tau) {
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int KEEP2 = LAT ? LINKS - 1 : DRM_RNEA2_KEEP;
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, P_FLOATS = (LINKS - 1 - KEEP2) * 6 * TILE2, T_FLOATS = round4(TILE2 * NJ),
                  F_FLOATS = P_FLOATS > T_FLOATS ? P_FLOATS : T_FLOATS; // tau is staged over the parking area at the end
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + F_FLOATS];
    const int tile = (int)blockIdx.x;
    const unsigned lane = threadIdx.x;
    float *lc = smem, *lt = smem + C_FLOATS;
    f2 *lf = reinterpret_cast<f2 *>(lt) + lane; // parked body forces: [link][6][64] pairs
    const int64_t b0 = (int64_t)tile * TILE2;

    float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];
    f2 qv[NJ], qdv[NJ], qddv[NJ], tv[NJ];
    {
        const int64_t ra = (b0 + lane) * NJ, rb = ra + (int64_t)WAVE * NJ;
#pragma unroll
        for (int d = 0; d < NJ; ++d) qv[d] = f2_make(q[ra + d], q[rb + d]);
#pragma unroll
        for (int d = 0; d < NJ; ++d) qdv[d] = f2_make(qd[ra + d], qd[rb + d]);
#pragma unroll
        for (int d = 0; d < NJ; ++d) qddv[d] = qdd ? f2_make(qdd[ra + d], qdd[rb + d]) : f2_bcast(0.0f);
    }
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    wave_lds_sync();
    f2 cs[NJ], sn[NJ];
    chain_trig2<NJ>(qv, cs, sn);
    rnea_chain2_trig<LINKS, NJ, KEEP2, LAT>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, flags & DRM_RNEA_GRAVITY,
                                flags & DRM_RNEA_DAMPING, cs, sn, qdv, qddv, tv,
                                [&](int k, const Force2 &F) {
#pragma unroll
                                    for (int i = 0; i < 3; ++i) {
                                        lf[(k * 6 + i) * WAVE] = F.f[i];
                                        lf[(k * 6 + 3 + i) * WAVE] = F.n[i];
                                    }
                                },
                                [&](int k, Force2 &F) {
#pragma unroll
                                    for (int i = 0; i < 3; ++i) {
                                        F.f[i] = lf[(k * 6 + i) * WAVE];
                                        F.n[i] = lf[(k * 6 + 3 + i) * WAVE];
                                    }
                                });
    wave_lds_sync(); // every lane is done with the parking area before tau is staged over it
#pragma unroll
    for (int d = 0; d < NJ; ++d) {
        lt[lane * NJ + d] = tv[d][0];
        lt[(WAVE + lane) * NJ + d] = tv[d][1];
    }
    wave_lds_sync();
    tile_store<2 * NJ>(tau + b0 * NJ, WAVE, 2 * NJ, 0u, lt, lane, true); // 128 rows of NJ floats = 64 "rows" of 2 NJ
}

// ---------------------------------------------------------------------------------------------------
// Inverse dynamics of a HAND (DRM_WALK_FINGERS: K serial chains of L revolute ops off the root — Allegro 4 x 4, TriFinger 3 x 3 —
// every fixed joint folded away on the host): each finger is a short arm, so it takes the arm recursion — two samples per lane,
// every instruction a full packed op (rnea_chain2_trig<L, L, L - 1>: all body forces in registers, nothing parked).  A block of
// K wavefronts owns 128 samples, wavefront w walks finger w: its L table rows in LDS, its L columns of q / qd / qdd read per lane
// (one 16-byte load per sample and array when L = 4 and the rows are 16-byte multiples), its L columns of tau written the same
// way — the K wavefronts of the block complete the rows between them.  The loop form (rnea_tree_kernel, one wavefront per
// finger and 64 samples, generic steps unrolled) issues 905 VALU per finger and 64 samples; this one ~450.
// ---------------------------------------------------------------------------------------------------
#ifndef DRM_FINGERS_WAVES4
#define DRM_FINGERS_WAVES4 2 /* waves per SIMD the L = 4 kernel is held to (L <= 3: three) */
#endif
#ifndef DRM_FINGERS_PARK
#define DRM_FINGERS_PARK 0 /* body forces of that many first links parked in LDS instead of registers */
#endif
// (L = 3, TriFinger: one parked link — with all three forces in registers the kernel sat 2 VGPRs over its three-wave budget and
// spilled them to scratch, profiles/r03_resource_usage.txt)
#define DRM_FINGERS_PARK_OF(L) ((L) == 3 && DRM_FINGERS_PARK == 0 ? 1 : DRM_FINGERS_PARK)
#define DRM_FINGERS_KEEP(L) ((L) - 1 - ((L) - 1 < DRM_FINGERS_PARK_OF(L) ? (L) - 1 : DRM_FINGERS_PARK_OF(L)))
template <int L>
__global__ void __launch_bounds__(WAVE * 4)
    __attribute__((amdgpu_waves_per_eu(L <= 3 ? 3 : DRM_FINGERS_WAVES4, L <= 3 ? 3 : DRM_FINGERS_WAVES4)))
    rnea_fingers2_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                         const float *__restrict__ qdd, int n, int flags, float *__restrict__ tau, int vec) {
    constexpr int C_FLOATS = L * DRM_OPF_STRIDE, KEEP = DRM_FINGERS_KEEP(L), PARK = (L - 1 - KEEP) * 6 * TILE2;
    __shared__ __attribute__((aligned(16))) float smem[4 * (C_FLOATS + PARK)];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * (C_FLOATS + PARK);
    f2 *lf = reinterpret_cast<f2 *>(lc + C_FLOATS) + lane; // parked body forces (if any): [link][6][64] pairs
    const int64_t b0 = (int64_t)blockIdx.x * TILE2;
    if (lane < (unsigned)(L * (DRM_OPF_STRIDE / 4)))
        reinterpret_cast<float4 *>(lc)[lane] = reinterpret_cast<const float4 *>(ops_f + (size_t)wave * C_FLOATS)[lane];
    const int64_t ra = (b0 + lane) * n + wave * L, rb = ra + (int64_t)WAVE * n;
    f2 qv[L], qdv[L], qddv[L], tv[L];
    auto load = [&](const float *src, f2 (&dst)[L]) {
        if (L == 4 && vec) {
            const float4 a = *reinterpret_cast<const float4 *>(src + ra), b = *reinterpret_cast<const float4 *>(src + rb);
            dst[0] = f2_make(a.x, b.x); dst[1] = f2_make(a.y, b.y); dst[2 % L] = f2_make(a.z, b.z); dst[3 % L] = f2_make(a.w, b.w);
        } else {
#pragma unroll
            for (int d = 0; d < L; ++d) dst[d] = f2_make(src[ra + d], src[rb + d]);
        }
    };
    load(q, qv);
    load(qd, qdv);
    if (qdd) load(qdd, qddv);
    else {
#pragma unroll
        for (int d = 0; d < L; ++d) qddv[d] = f2_bcast(0.0f);
    }
    wave_lds_sync();
    f2 cs[L], sn[L];
    chain_trig2<L>(qv, cs, sn);
    rnea_chain2_trig<L, L, KEEP>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, flags & DRM_RNEA_GRAVITY,
                                 flags & DRM_RNEA_DAMPING, cs, sn, qdv, qddv, tv,
                                 [&](int k, const Force2 &F) {
#pragma unroll
                                     for (int i = 0; i < 3; ++i) { lf[(k * 6 + i) * WAVE] = F.f[i]; lf[(k * 6 + 3 + i) * WAVE] = F.n[i]; }
                                 },
                                 [&](int k, Force2 &F) {
#pragma unroll
                                     for (int i = 0; i < 3; ++i) { F.f[i] = lf[(k * 6 + i) * WAVE]; F.n[i] = lf[(k * 6 + 3 + i) * WAVE]; }
                                 });
    if (L == 4 && vec) {
        *reinterpret_cast<float4 *>(tau + ra) = make_float4(tv[0][0], tv[1][0], tv[2 % L][0], tv[3 % L][0]);
        *reinterpret_cast<float4 *>(tau + rb) = make_float4(tv[0][1], tv[1][1], tv[2 % L][1], tv[3 % L][1]);
    } else {
#pragma unroll
        for (int d = 0; d < L; ++d) { tau[ra + d] = tv[d][0]; tau[rb + d] = tv[d][1]; }
    }
}

// rows covered (full 128-row tiles), 0 = the call does not qualify
int64_t launch_rnea_fingers(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau,
                            hipStream_t s) {
#ifdef DRM_NO_FINGERS_KERNEL
    return 0;
#else
    if (!(w->shape & DRM_WALK_FINGERS) || B < TILE2 || B / TILE2 >= 0x7fffffffLL || (((uintptr_t)w->ops_f) & 15u) != 0) return 0;
    const int K = DRM_WALK_AH_K(w->shape), L = DRM_WALK_AH_L(w->shape), n = w->n_dofs;
    if (K * L != w->n_ops || n != w->n_ops || K < 2 || K > 4 || L < 2 || L > 4) return 0;
    const int n2 = (int)(B / TILE2);
    const int vec = (n % 4 == 0) && ((((uintptr_t)q | (uintptr_t)qd | (uintptr_t)qdd | (uintptr_t)tau) & 15u) == 0);
#define X(l)                                                                                                                     \
    if (L == l) hipLaunchKernelGGL((rnea_fingers2_kernel<l>), dim3((unsigned)n2), dim3(WAVE * K), 0, s, w->ops_f, q, qd, qdd, n, flags, tau, vec);
    X(2) X(3) X(4)
#undef X
    return (int64_t)n2 * TILE2;
#endif
}

static void launch_rnea_arm2_stream(const float *ops_f, int links, const float *q, const float *qd, const float *qdd, int n2, int flags,
                                    float *tau, hipStream_t s);
void launch_rnea_arm(const float *ops_f, int links, const float *q, const float *qd, const float *qdd, int n_tiles, int flags,
                     float *tau, hipStream_t s) {
#ifndef DRM_RNEA_ONE_SAMPLE_PER_LANE
    // pairs of 64-sample tiles through the two-samples-per-lane kernel, an odd last tile through the one-sample kernel
    const int n2 = n_tiles > TWO_SAMPLE_MIN_TILES ? n_tiles / 2 : 0;
    if (n2 > 0) {
#define RNEA2(L, LATENCY) hipLaunchKernelGGL((rnea_arm2_kernel<8, 7, L, LATENCY>), dim3((unsigned)n2), dim3(WAVE), 0, s, ops_f, q, qd, qdd, n2, flags, tau)
        if (n2 >= PIPE_MIN_TILES) { // several tiles per resident wavefront: the streaming form (arm2_stream_kernel below)
            launch_rnea_arm2_stream(ops_f, links, q, qd, qdd, n2, flags, tau, s);
        } else if (n2 <= LAT2_MAX_TILES) { // one wave per SIMD: the latency form
            if (links == 7) RNEA2(7, true);
            else RNEA2(8, true);
        } else {
            if (links == 7) RNEA2(7, false);
            else RNEA2(8, false);
        }
#undef RNEA2
    }
    if (n2 > 0) {
        if (!(n_tiles & 1)) return;
        const int64_t done = (int64_t)n2 * TILE2 * 7;
        q += done; qd += done; qdd = qdd ? qdd + done : nullptr; tau += done;
        n_tiles = 1;
    }
#endif
    const dim3 grid((unsigned)((n_tiles + MAX_WAVES_PER_BLOCK - 1) / MAX_WAVES_PER_BLOCK)), block(WAVE * MAX_WAVES_PER_BLOCK);
    // (at most 1 024 tiles reach this kernel: one wave per SIMD — the latency form, see the kernel)
    if (links == 7) hipLaunchKernelGGL((rnea_arm_kernel<8, 7, 7, LAT1>), grid, block, 0, s, ops_f, q, qd, qdd, n_tiles, flags, tau);
    else hipLaunchKernelGGL((rnea_arm_kernel<8, 7, 8, LAT1>), grid, block, 0, s, ops_f, q, qd, qdd, n_tiles, flags, tau);
}

// ---------------------------------------------------------------------------------------------------
// Fused FK(target) + RNEA of a serial chain whose LAST link is the FK target (BASELINE configuration 3: Panda,
// q / qd / qdd -> tau, pos, quat; 140 B per evaluation instead of the 168 B and two launch floors of drm_fk +
// drm_rnea): one load of q, one sin/cos evaluation, one staged constant table; the FK chain runs first (its pose goes out
// while the dynamics sweeps run), then the RNEA sweeps of rnea_arm_kernel.  Arithmetic is that of the two separate
// kernels, bit for bit (drm_sample.hpp: fk_chain_pairs_trig / rnea_chain_trig on shared cos / sin).
// ---------------------------------------------------------------------------------------------------
// (LINKS as above: the FK chain always walks all CAP ops, the dynamics sweeps the first LINKS)
// amdgpu_waves_per_eu: the register allocator's occupancy goal follows the LDS footprint, and with the LINKS-sized parking
// area it would settle on 170 VGPRs, two short of a third wave per SIMD; the attribute holds it to <= 168
template <int CAP, int NJ, int LINKS, bool LAT = false>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(LAT ? 1 : (LINKS < CAP ? 3 : 2), LAT ? 2 : (LINKS < CAP ? 3 : 2))))
    fk_rnea_arm_kernel(const float *__restrict__ ops_f, const float *__restrict__ ops_tail, const float *__restrict__ q, const float *__restrict__ qd,
                       const float *__restrict__ qdd, int n_tiles, int flags, float *__restrict__ tau,
                       float *__restrict__ pos, float *__restrict__ quat) {
    static_assert(NJ & 1, "odd row widths only (linear LDS image)");
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    // parking area sized by the links actually parked (LINKS - DRM_RNEA_KEEP), not by the table capacity: with LINKS = 7
    // a four-wave block takes 44 KB instead of 56 KB and a CU holds three of them (164 VGPR allow three waves per SIMD)
    constexpr int KEEP = LAT ? LINKS : DRM_RNEA_KEEP; // (LAT: see rnea_arm_kernel)
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, Q_FLOATS = round4(WAVE * NJ), P_FLOATS = WAVE * 3,
                  F_FLOATS = (LINKS - KEEP) * 6 * WAVE > Q_FLOATS ? (LINKS - KEEP) * 6 * WAVE : Q_FLOATS; // (tau is staged over the parking area)
    constexpr int PER_WAVE = C_FLOATS + P_FLOATS + F_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[MAX_WAVES_PER_BLOCK * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = (int)blockIdx.x * MAX_WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * PER_WAVE;
    float *lp = lc + C_FLOATS, *lq = lp + P_FLOATS;
    float *lf = lq + lane;
    const int64_t b0 = (int64_t)tile * WAVE;

    // rows 0 .. LINKS-1 (the ops the dynamics sweeps visit, with whatever the host folded into them) come from the TREE walk's
    // table, rows LINKS .. CAP-1 (the fixed tail the FK chain still walks) from the CHAIN walk's: the two tables need not agree
    // on the inertial columns, and the FT blocks of rows 0 .. LINKS-1 are the same in both (folding never touches them)
    float4 cv = reinterpret_cast<const float4 *>(lane < LINKS * (DRM_OPF_STRIDE / 4) ? ops_f : ops_tail)[lane];
    float qv[NJ], qdv[NJ], qddv[NJ], tv[NJ];
    {
        const int64_t row = (b0 + lane) * NJ;
#pragma unroll
        for (int d = 0; d < NJ; ++d) qv[d] = q[row + d];
#pragma unroll
        for (int d = 0; d < NJ; ++d) qdv[d] = qd[row + d];
#pragma unroll
        for (int d = 0; d < NJ; ++d) qddv[d] = qdd ? qdd[row + d] : 0.0f;
    }
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    wave_lds_sync();

    float cs[NJ], sn[NJ];
    chain_trig<NJ>(qv, cs, sn);
    auto row = [&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; };
    {   // forward kinematics of the last link (robot_model.py:223-248)
        PoseP ee;
        f2 Bk[NJ][3];
        fk_chain_pairs_trig<CAP, NJ>(row, cs, sn, ee, Bk, [] {});
        lp[lane * 3 + 0] = ee.B[0][1];
        lp[lane * 3 + 1] = ee.B[1][1];
        lp[lane * 3 + 2] = ee.B[2][1];
        wave_lds_sync();
        tile_store<3>(pos + b0 * 3, WAVE, 3, 0u, lp, lane, true);
        Pose E;
        float qt[4];
        pose_from_pairs(ee, E);
        quat_xyzw(E.R, qt);
        store16_wt(quat + (b0 + lane) * 4, make_float4(qt[0], qt[1], qt[2], qt[3]));
    }
    // inverse dynamics (robot_model.py:305-375)
    rnea_chain_trig<LINKS, NJ, KEEP, LAT>(row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, cs, sn, qdv, qddv, tv,
                             [&](int k, const Force &F) {
#pragma unroll
                                 for (int i = 0; i < 3; ++i) {
                                     lf[(k * 6 + i) * WAVE] = F.la[i][0];
                                     lf[(k * 6 + 3 + i) * WAVE] = F.la[i][1];
                                 }
                             },
                             [&](int k, Force &F) {
#pragma unroll
                                 for (int i = 0; i < 3; ++i) F.la[i] = f2_make(lf[(k * 6 + i) * WAVE], lf[(k * 6 + 3 + i) * WAVE]);
                             });
    wave_lds_sync(); // every lane is done with the parking area before tau is staged over it
#pragma unroll
    for (int d = 0; d < NJ; ++d) lq[lane * NJ + d] = tv[d];
    wave_lds_sync();
    tile_store<NJ>(tau + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
}

// The fused launch with two samples per lane: FK chain of all CAP ops first (pose out while the dynamics sweeps run), then
// rnea_chain2_trig on the first LINKS ops; cos / sin shared.
template <int CAP, int NJ, int LINKS, bool LAT = false>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(LAT ? 1 : DRM_RNEA2_WAVES, LAT ? 2 : DRM_RNEA2_WAVES)))
    fk_rnea_arm2_kernel(const float *__restrict__ ops_f, const float *__restrict__ ops_tail, const float *__restrict__ q, const float *__restrict__ qd,
                        const float *__restrict__ qdd, int n_tiles, int flags, float *__restrict__ tau, float *__restrict__ pos,
                        float *__restrict__ quat) {
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int KEEP2 = LAT ? LINKS - 1 : DRM_RNEA2_KEEP;
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, P_FLOATS = (LINKS - 1 - KEEP2) * 6 * TILE2, T_FLOATS = round4(TILE2 * NJ),
                  F_FLOATS = P_FLOATS > T_FLOATS ? P_FLOATS : T_FLOATS;
    static_assert(TILE2 * 3 <= F_FLOATS, "the position tile fits into the staging area");
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + F_FLOATS];
    const int tile = (int)blockIdx.x;
    const unsigned lane = threadIdx.x;
    float *lc = smem, *lt = smem + C_FLOATS;
    f2 *lf = reinterpret_cast<f2 *>(lt) + lane;
    const int64_t b0 = (int64_t)tile * TILE2;

    float4 cv = reinterpret_cast<const float4 *>(lane < LINKS * (DRM_OPF_STRIDE / 4) ? ops_f : ops_tail)[lane]; // as in fk_rnea_arm_kernel
    f2 qv[NJ], qdv[NJ], qddv[NJ], tv[NJ];
    {
        const int64_t ra = (b0 + lane) * NJ, rb = ra + (int64_t)WAVE * NJ;
#pragma unroll
        for (int d = 0; d < NJ; ++d) qv[d] = f2_make(q[ra + d], q[rb + d]);
#pragma unroll
        for (int d = 0; d < NJ; ++d) qdv[d] = f2_make(qd[ra + d], qd[rb + d]);
#pragma unroll
        for (int d = 0; d < NJ; ++d) qddv[d] = qdd ? f2_make(qdd[ra + d], qdd[rb + d]) : f2_bcast(0.0f);
    }
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    wave_lds_sync();
    f2 cs[NJ], sn[NJ];
    chain_trig2<NJ>(qv, cs, sn);
    auto row = [&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; };
    {   // forward kinematics of the last link (robot_model.py:223-248)
        Pose2 ee;
        fk_chain2_trig<CAP, NJ>(row, cs, sn, ee); // (not prefetched: with its constants in registers the compiler re-associates the pose chain and the pose is no longer bit-identical to drm_fk's)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lt[lane * 3 + c] = ee.p[c][0];
            lt[(WAVE + lane) * 3 + c] = ee.p[c][1];
        }
        wave_lds_sync();
        tile_store<6>(pos + b0 * 3, WAVE, 6, 0u, lt, lane, true); // 128 rows of 3 floats
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float R[9], qt[4];
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = ee.R[i][h];
            quat_xyzw(R, qt);
            store16_wt(quat + (b0 + h * WAVE + lane) * 4, make_float4(qt[0], qt[1], qt[2], qt[3]));
        }
        wave_lds_sync(); // the position tile has left before the parking area is written
    }
    rnea_chain2_trig<LINKS, NJ, KEEP2, LAT>(row, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, cs, sn, qdv, qddv, tv,
                                [&](int k, const Force2 &F) {
#pragma unroll
                                    for (int i = 0; i < 3; ++i) {
                                        lf[(k * 6 + i) * WAVE] = F.f[i];
                                        lf[(k * 6 + 3 + i) * WAVE] = F.n[i];
                                    }
                                },
                                [&](int k, Force2 &F) {
#pragma unroll
                                    for (int i = 0; i < 3; ++i) {
                                        F.f[i] = lf[(k * 6 + i) * WAVE];
                                        F.n[i] = lf[(k * 6 + 3 + i) * WAVE];
                                    }
                                });
    wave_lds_sync();
#pragma unroll
    for (int d = 0; d < NJ; ++d) {
        lt[lane * NJ + d] = tv[d][0];
        lt[(WAVE + lane) * NJ + d] = tv[d][1];
    }
    wave_lds_sync();
    tile_store<2 * NJ>(tau + b0 * NJ, WAVE, 2 * NJ, 0u, lt, lane, true);
}

// ---------------------------------------------------------------------------------------------------
// Round 5: the STREAMING form of the two-samples-per-lane kernels (launches of at least PIPE_MIN_TILES 128-row tiles, e.g. BASELINE
// configuration 3 as one launch of 2^20 rows = 8 192 tiles).  fk_rnea_arm2_kernel<..., false> ran such a launch as 8 192 independent
// wavefronts, three per SIMD, each one load -> 2 000 VALU -> store with nothing of its own in flight; the waves of a SIMD start
// together and stay in phase, so the launch took t_HBM + t_VALU (18.4 + 22.9 us of floors, 40.6 us measured).
// Here the grid is what the device holds at two wavefronts per SIMD and every wavefront walks tiles blockIdx.x, + gridDim.x, ...:
//   * the NEXT tile's q / qd / qdd rows go from HBM straight into an LDS staging area (global_load_lds_dwordx4: 16 B per lane, a
//     contiguous 1 KB per instruction, no registers in between — 12 instructions per tile instead of 42 strided dword loads),
//     issued BEFORE this tile's trigonometry, pose chain and dynamics sweeps;
//   * this tile's pos / quat leave while its dynamics sweeps run, its tau while the next tile's trigonometry runs;
//   * the one wait (s_waitcnt vmcnt(0), just before the next rows are read back from LDS) finds loads that were issued ~2 000
//     VALU instructions earlier.
// Arithmetic: that of the latency form (KEEP2 = LINKS - 1: every body force in registers, constants prefetched one link ahead) —
// with two wavefronts per SIMD the register file has room for it (<= 256 VGPRs) and nothing is parked in LDS.
// LDS per wavefront: table 1 KB + tau / pos staging 3.5 KB + next rows 10.5 KB = 15 KB (eight wavefronts per CU: 120 KB).
// ---------------------------------------------------------------------------------------------------
#ifndef DRM_PIPE_PREF
#define DRM_PIPE_PREF true
#endif
template <int CAP, int NJ, int LINKS, bool FK>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(DRM_PIPE_WAVES, DRM_PIPE_WAVES)))
    arm2_stream_kernel(const float *__restrict__ ops_f, const float *__restrict__ ops_tail, const float *__restrict__ q, const float *__restrict__ qd,
                       const float *__restrict__ qdd, int n_tiles, int flags, float *__restrict__ tau, float *__restrict__ pos,
                       float *__restrict__ quat) {
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, T_FLOATS = round4(STREAM_TILE * NJ), ROWS_F = STREAM_TILE * NJ;
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + T_FLOATS + 3 * ROWS_F];
    float *lc = smem;
    const unsigned lane = threadIdx.x;
    arm2_stream_body<CAP, NJ, LINKS, FK, DRM_PIPE_PREF>(
        [&] { // rows 0 .. LINKS-1 from the tree walk's table, the fixed tail the pose chain still walks from the chain walk's (fk_rnea_arm_kernel)
            float4 cv = reinterpret_cast<const float4 *>(FK && lane >= LINKS * (DRM_OPF_STRIDE / 4) ? ops_tail : ops_f)[lane];
            pin(cv);
            reinterpret_cast<float4 *>(lc)[lane] = cv;
        },
        [&] {
            // the table's address is laundered once per tile: its rows are loop-invariant LDS reads that provably alias nothing the
            // loop writes, so the compiler would hoist all 200 of them out of the loop (and spill them).  An OFFSET is laundered, so
            // that the reads stay ds_reads: a laundered pointer loses its address space and reads flat.
            unsigned lcw = 0;
            asm volatile("" : "+v"(lcw));
            return [lc, lcw](int k) -> const float * { return lc + lcw + k * DRM_OPF_STRIDE; };
        },
        smem + C_FLOATS, q, qd, qdd, n_tiles, flags, tau, pos, quat);
}
// the grid of the streaming form: DRM_PIPE_WAVES wavefronts per SIMD on every CU of the current device (asked once per device)
int arm_stream_grid(int n_tiles) {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return n_tiles < 2048 ? n_tiles : 2048;
    int g = cached[dev];
    if (g == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        g = cus * 4 * DRM_PIPE_WAVES;
        cached[dev] = g;
    }
    return n_tiles < g ? n_tiles : g;
}

static void launch_rnea_arm2_stream(const float *ops_f, int links, const float *q, const float *qd, const float *qdd, int n2, int flags,
                                    float *tau, hipStream_t s) {
    const dim3 grid((unsigned)arm_stream_grid(n2)), block(WAVE);
    if (links == 7)
        hipLaunchKernelGGL((arm2_stream_kernel<8, 7, 7, false>), grid, block, 0, s, ops_f, ops_f, q, qd, qdd, n2, flags, tau, (float *)nullptr, (float *)nullptr);
    else
        hipLaunchKernelGGL((arm2_stream_kernel<8, 7, 8, false>), grid, block, 0, s, ops_f, ops_f, q, qd, qdd, n2, flags, tau, (float *)nullptr, (float *)nullptr);
}

void launch_fk_rnea_arm(const float *ops_f, const float *ops_tail, int links, const float *q, const float *qd, const float *qdd, int n_tiles, int flags,
                        float *tau, float *pos, float *quat, hipStream_t s) {
#ifndef DRM_RNEA_ONE_SAMPLE_PER_LANE
    const int n2 = n_tiles > TWO_SAMPLE_MIN_TILES ? n_tiles / 2 : 0; // pairs of 64-sample tiles: two samples per lane; an odd last tile: the one-sample kernel
    if (n2 > 0) {
#define FK_RNEA2_(L, LATENCY) hipLaunchKernelGGL((fk_rnea_arm2_kernel<8, 7, L, LATENCY>), dim3((unsigned)n2), dim3(WAVE), 0, s, ops_f, ops_tail, q, qd, qdd, n2, flags, tau, pos, quat)
#define FK_RNEA2S(L) hipLaunchKernelGGL((arm2_stream_kernel<8, 7, L, true>), dim3((unsigned)arm_stream_grid(n2)), dim3(WAVE), 0, s, ops_f, ops_tail, q, qd, qdd, n2, flags, tau, pos, quat)
#define FK_RNEA2(L) do { if (n2 >= PIPE_MIN_TILES) FK_RNEA2S(L); else if (n2 <= LAT2_MAX_TILES) FK_RNEA2_(L, true); else FK_RNEA2_(L, false); } while (0)
        if (links == 7)
            FK_RNEA2(7);
        else
            FK_RNEA2(8);
#undef FK_RNEA2
#undef FK_RNEA2S
#undef FK_RNEA2_
    }
    if (n2 > 0) {
        if (!(n_tiles & 1)) return;
        const int64_t rows = (int64_t)n2 * TILE2;
        q += rows * 7; qd += rows * 7; qdd = qdd ? qdd + rows * 7 : nullptr; tau += rows * 7; pos += rows * 3; quat += rows * 4;
        n_tiles = 1;
    }
#endif
    const dim3 grid((unsigned)((n_tiles + MAX_WAVES_PER_BLOCK - 1) / MAX_WAVES_PER_BLOCK)), block(WAVE * MAX_WAVES_PER_BLOCK);
    if (links == 7)
        hipLaunchKernelGGL((fk_rnea_arm_kernel<8, 7, 7, LAT1>), grid, block, 0, s, ops_f, ops_tail, q, qd, qdd, n_tiles, flags, tau, pos, quat);
    else
        hipLaunchKernelGGL((fk_rnea_arm_kernel<8, 7, 8, LAT1>), grid, block, 0, s, ops_f, ops_tail, q, qd, qdd, n_tiles, flags, tau, pos, quat);
}

} // namespace drm
DRM_TL_READER(dyn)
