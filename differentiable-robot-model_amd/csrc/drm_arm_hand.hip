// drm_arm_hand.hip — the dynamics kernels of robots shaped like "an arm that carries a hand" (round 3): P serial prefix ops and
// K serial sub-chains of L ops that all hang off the last prefix op (DRM_WALK_ARM_HAND, include/drm_hip.h) — K3 inverse
// dynamics, K8 forward dynamics, K6 the mass matrix, K7 the reverse mode of K3, in that order below.
// Franka Panda with its gripper (P, K, L) = (7, 2, 1), Kinova Jaco (6, 3, 2), KUKA iiwa7 + Allegro hand (7, 4, 4) once the host has
// folded the fixed joints away; compiled for every P = 5 .. 9 and L = 1 .. 4 (this file once per L, see DRM_ARM_HAND_SHAPES).
//
// These robots are ONE dynamics segment (everything hangs off the moving arm), so the loop-structured kernels walk a whole
// 64-sample tile with one wavefront, park per-link records (LDS or HBM scratch) and decode two control words per op.  Here the
// walks are straight-line code for the shape (drm_sample.hpp rnea_arm_hand / rnea_backward_arm_hand, drm_tree.hpp aba_arm_hand /
// crba_arm_hand_*): the prefix is the arm kernels' chain walk, every sub-chain is swept while the palm's state is live, so what
// is parked is a few floats per PREFIX op (or nothing).  Which DoF column an op drives comes from the walk's W0 words (scalar
// loads): fixed ops, sliding joints and arbitrary DoF numbering need no template parameters beyond (P, L); K is a run-time
// count.  Full 64-row tiles only; a ragged tail goes through the loop-structured kernels.
// Build flags as drm_arm_dynamics.hip (kernel-argument preload, no SLP vectoriser).
#include "drm_common.hpp"
#include "drm_sample.hpp"
#include "drm_tree.hpp"
#include "drm_tree_dev.hpp"
#include "drm_static.hpp"

namespace drm {

constexpr int AH_MAX_OPS = 32; // P + K L of any compiled shape
constexpr int AH_MAX_K = 4;    // sub-chains of a shape (two bits of drm_walk.shape): the LDS of a <P, L> kernel is sized for P + 4 L ops —
                               // 11.6 instead of 17.4 KB per wavefront for a 7-joint arm with a gripper, i.e. the three wavefronts per
                               // SIMD its registers allow instead of nine per CU

// LDS (static — with a dynamic allocation the compiler's occupancy guess goes wrong and the same code needs 300 registers):
// [ table : (P + 4 L) x 32 ][ torques of the sub-chain ops : 4 L x 64 ][ parking : (P - 1) x 6 x 64 | tau tile : 64 x (n | 1) ]
template <int P, int L>
__global__ void __launch_bounds__(WAVE)
    rnea_arm_hand_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, const float *__restrict__ q,
                         const float *__restrict__ qd, const float *__restrict__ qdd, int K, int cap, int n, int flags,
                         float *__restrict__ tau, uint32_t magic_n) {
    constexpr int OPS = P + AH_MAX_K * L;
    constexpr int C_FLOATS = OPS * DRM_OPF_STRIDE, H_FLOATS = AH_MAX_K * L * WAVE;
    constexpr int PARK = (P - 1) * 6 * WAVE, TILE = round4(WAVE * pad_odd(OPS));
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + H_FLOATS + (PARK > TILE ? PARK : TILE)];
    const unsigned lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * WAVE;
    const int n_ops = P + K * L;
    float *lc = smem, *lh = smem + C_FLOATS + lane, *lt = smem + C_FLOATS + H_FLOATS;
    float *lf = lt + lane; // parked body forces of the prefix: [op][6][64]
    const int32_t *w0 = ops_i + DRM_OPI_W0 * cap; // control words (wave-uniform scalar loads): bits 0..7 = DoF column + 1

    // DoF column of the prefix ops (-1: fixed)
    int dof[P];
#pragma unroll
    for (int k = 0; k < P; ++k) dof[k] = (w0[k] & 0xff) - 1;
    // constant rows -> LDS
    for (unsigned i = lane; i < (unsigned)n_ops * (DRM_OPF_STRIDE / 4); i += WAVE)
        reinterpret_cast<float4 *>(lc)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    // this lane's rows, one dword load per op and array at a wave-uniform column (fixed ops: column 0, value dropped):
    // uniform base (tile + column) + ONE 32-bit per-lane byte offset shared by every load
    const unsigned row_off = lane * (unsigned)n * 4u;
    const char *qb = reinterpret_cast<const char *>(q + b0 * n), *qdb = reinterpret_cast<const char *>(qd + b0 * n),
               *qddb = reinterpret_cast<const char *>(qdd ? qdd + b0 * n : q + b0 * n);
    const bool has_qdd = qdd != nullptr;
    auto joint_state = [&](int d, float &a, float &v, float &acc) {
        const int c = (d < 0 ? 0 : d) * 4;
        const float x = *reinterpret_cast<const float *>(qb + c + row_off), y = *reinterpret_cast<const float *>(qdb + c + row_off),
                    z = *reinterpret_cast<const float *>(qddb + c + row_off);
        a = d < 0 ? 0.0f : x; v = d < 0 ? 0.0f : y; acc = (d < 0 || !has_qdd) ? 0.0f : z;
    };
    float qv[P], qdv[P], qddv[P];
#pragma unroll
    for (int k = 0; k < P; ++k) joint_state(dof[k], qv[k], qdv[k], qddv[k]);
    wave_lds_sync();
    float cs[P], sn[P], tp[P];
    chain_trig<P>(qv, cs, sn);
    // bits of an op: 1 = moves, 2 = prismatic (W0 bit 26)
    auto kind = [&](int op) { const int w = w0[op]; return ((w & 0xff) ? 1 : 0) | (((w >> 26) & 1) << 1); };
    rnea_arm_hand<P, L>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, kind, K, flags & DRM_RNEA_GRAVITY,
                        flags & DRM_RNEA_DAMPING, qv, cs, sn, qdv, qddv,
                        [&](int j, int i, float &a, float &v, float &acc) { joint_state((w0[P + j * L + i] & 0xff) - 1, a, v, acc); }, tp,
                        [&](int j, int i, float t) { lh[(j * L + i) * WAVE] = t; },
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
    float *trow = lt + lane * pad_odd(n);
#pragma unroll
    for (int k = 0; k < P; ++k)
        if (dof[k] >= 0) trow[dof[k]] = tp[k];
    for (int k = 0; k < K * L; ++k) {
        const int d = (w0[P + k] & 0xff) - 1;
        if (d >= 0) trow[d] = lh[k * WAVE];
    }
    wave_lds_sync();
    tile_store<0>(tau + b0 * n, WAVE, n, magic_n, lt, lane, (n & 1) != 0, true);
}

// ---------------------------------------------------------------------------------------------------
// K8 (forward dynamics) of the same shapes: the articulated-body recursion of drm_tree.hpp aba_arm_hand — prefix records in
// one 8-float LDS slot per op (velocities, then U, 1/D, u), sub-chains visited twice instead of stored: no scratch, 12-20 KB
// of LDS per wave.  The loop form (forward_dynamics_aba_kernel) needs HBM scratch and holds five wavefronts per CU.
// LDS (static): [ table ][ slots : P x 8 x 64 | qdd tile : 64 x (n | 1) staged over them once sweep 3 of the prefix is done ]
// ---------------------------------------------------------------------------------------------------
template <int P, int L>
__global__ void __launch_bounds__(WAVE)
    forward_dynamics_arm_hand_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, const float *__restrict__ q,
                                     const float *__restrict__ qd, const float *__restrict__ f, int K, int cap, int n, int flags,
                                     float *__restrict__ qdd, uint32_t magic_n) {
    constexpr int OPS = P + AH_MAX_K * L;
    constexpr int C_FLOATS = OPS * DRM_OPF_STRIDE;
    constexpr int SLOTS = P * 8 * WAVE, TILE = round4(WAVE * pad_odd(OPS));
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + (SLOTS > TILE ? SLOTS : TILE)];
    const unsigned lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * WAVE;
    const int n_ops = P + K * L;
    float *lc = smem, *lt = smem + C_FLOATS;
    float *ls = lt + lane; // prefix slots: [op][8][64]
    const int32_t *w0 = ops_i + DRM_OPI_W0 * cap;

    int dof[P];
#pragma unroll
    for (int k = 0; k < P; ++k) dof[k] = (w0[k] & 0xff) - 1;
    for (unsigned i = lane; i < (unsigned)n_ops * (DRM_OPF_STRIDE / 4); i += WAVE)
        reinterpret_cast<float4 *>(lc)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    const unsigned row_off = lane * (unsigned)n * 4u;
    const char *qb = reinterpret_cast<const char *>(q + b0 * n), *qdb = reinterpret_cast<const char *>(qd + b0 * n),
               *fb = reinterpret_cast<const char *>(f + b0 * n);
    auto joint_state = [&](int d, float &a, float &v, float &t) {
        const int c = (d < 0 ? 0 : d) * 4;
        const float x = *reinterpret_cast<const float *>(qb + c + row_off), y = *reinterpret_cast<const float *>(qdb + c + row_off),
                    z = *reinterpret_cast<const float *>(fb + c + row_off);
        a = d < 0 ? 0.0f : x; v = d < 0 ? 0.0f : y; t = d < 0 ? 0.0f : z;
    };
    float qv[P], qdv[P], fv[P];
#pragma unroll
    for (int k = 0; k < P; ++k) joint_state(dof[k], qv[k], qdv[k], fv[k]);
    wave_lds_sync();
    float cs[P], sn[P], out[P];
    chain_trig<P>(qv, cs, sn);
    auto kind = [&](int op) { const int w = w0[op]; return ((w & 0xff) ? 1 : 0) | (((w >> 26) & 1) << 1); };
    float *trow = lt + lane * pad_odd(n);
    bool tile_open = false; // the slots turn into the qdd tile once the prefix's third sweep has read its last record
    aba_arm_hand<P, L>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, kind, K, flags & DRM_RNEA_GRAVITY,
                       flags & DRM_RNEA_DAMPING, qv, cs, sn, qdv, fv,
                       [&](int j, int i, float &a, float &v, float &t) { joint_state((w0[P + j * L + i] & 0xff) - 1, a, v, t); }, out,
                       [&](int j, int i, float a) {
                           if (!tile_open) { wave_lds_sync(); tile_open = true; }
                           trow[(w0[P + j * L + i] & 0xff) - 1] = a;
                       },
                       [&](int k, const float *s8) {
#pragma unroll
                           for (int c = 0; c < 8; ++c) ls[(k * 8 + c) * WAVE] = s8[c];
                       },
                       [&](int k, float *s8) {
#pragma unroll
                           for (int c = 0; c < 8; ++c) s8[c] = ls[(k * 8 + c) * WAVE];
                       });
    if (!tile_open) wave_lds_sync();
#pragma unroll
    for (int k = 0; k < P; ++k)
        if (dof[k] >= 0) trow[dof[k]] = out[k];
    wave_lds_sync();
    tile_store<0>(qdd + b0 * n, WAVE, n, magic_n, lt, lane, (n & 1) != 0, true);
}

// ---------------------------------------------------------------------------------------------------
// K6 (mass matrix) of the same shapes: drm_tree.hpp crba_arm_hand_sub / crba_arm_hand_prefix — a walk that parks nothing (the
// loop form, crba_rows_kernel, keeps the column forces of all the joints below an op in LDS: 35 KB per 64 samples on the arm
// with a hand, three wavefronts per CU) — and an output path without scratch.  H is 4 n^2 bytes per sample (2.1 KB at 23 DoF):
// the kernel is bound by writing it, so the entries must leave as whole cache lines, and a lane that keeps a sample cannot do
// that (its entries are 4 n^2 bytes apart; through a slice of HBM scratch and back the tile's triangle crossed the memory
// system three times: 1.8 ms per 2^20 samples, 0.33 ms of it the walk).  Here a BLOCK owns 64 samples and has one wavefront
// per sub-chain: wavefront j walks sub-chain j (its own entries, its L column forces up the prefix), the K composites meet in
// the palm through LDS, the prefix's columns are dealt round-robin to the wavefronts; every entry goes to an LDS triangle
// [slot][64 + 1] (the pad: the writers are 64 lanes of one slot, the readers 64 slots of one sample), and the block then writes
// the 64 matrices as one run of 16-byte stores, written through the L2 and — beyond the Infinity Cache — past it (bytes-only
// launches of this shape, tools/ubench/metric_lab floors: 2.3 GB take 723 us with plain stores, 403 us with sc1 nt).
// LDS (static): [ table ][ K composites : 10 x 64 each ][ triangle : slots x 65 ]
//               9 + 2x1 ops: 24 KB, 7 + 3x2: 31 KB, 8 + 4x4: 66 KB (two blocks = eight wavefronts per CU)
// ---------------------------------------------------------------------------------------------------
constexpr int AH_TRI_STRIDE = WAVE + 1;
// wavefronts (= sub-chains) the mass-matrix kernel of a shape reserves LDS for: a gripper's two fingers (and the tool frame
// that stays an op beside them when the hand is learnable), the Jaco's three, the Allegro's four; a robot of the shape with
// more of them takes the loop kernel
constexpr int crba_max_k(int L) { return L <= 2 ? 3 : 4; }
template <int P, int L>
struct AhTriangle { // slots of the pairs (oa <= ob) of ops on a common root path
    static constexpr int KMAX = crba_max_k(L), OPS = P + KMAX * L;
    static constexpr int PPN = P * (P + 1) / 2, SUBN = L * P + L * (L + 1) / 2, SLOTS = PPN + KMAX * SUBN;
    static DRM_HD int slot(int oa, int ob) {
        if (ob < P) return ob * (ob + 1) / 2 + oa;
        const int j = (ob - P) / L, i = ob - P - j * L;
        return PPN + j * SUBN + i * P + i * (i + 1) / 2 + (oa < P ? oa : oa - j * L);
    }
    static DRM_HD bool related(int oa, int ob) { return oa < P || (oa - P) / L == (ob - P) / L; }
};
template <int P, int L, bool NT>
__global__ void __launch_bounds__(WAVE *crba_max_k(L))
    crba_arm_hand_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, const float *__restrict__ q, int K, int cap,
                         int n, float *__restrict__ H) {
    using T = AhTriangle<P, L>;
    constexpr int C_FLOATS = T::OPS * DRM_OPF_STRIDE, X_FLOATS = T::KMAX * 10 * WAVE, ZERO_SLOT = T::SLOTS;
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + X_FLOATS + (T::SLOTS + 1) * AH_TRI_STRIDE];
    __shared__ int op_of_dof[T::OPS];
    __shared__ int slot_of[T::OPS * T::OPS]; // element (row, column) of a matrix -> its slot x stride (the zero slot: unrelated joints)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n_ops = P + K * L, nn = n * n;
    float *lc = smem, *lx = smem + C_FLOATS, *tri = lx + X_FLOATS;
    const int32_t *w0 = ops_i + DRM_OPI_W0 * cap;
    for (unsigned i = threadIdx.x; i < (unsigned)n_ops * (DRM_OPF_STRIDE / 4); i += blockDim.x)
        reinterpret_cast<float4 *>(lc)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    if ((int)threadIdx.x < n_ops) {
        const int d = (w0[threadIdx.x] & 0xff) - 1;
        if (d >= 0) op_of_dof[d] = (int)threadIdx.x;
    }
    if (threadIdx.x <= WAVE) tri[ZERO_SLOT * AH_TRI_STRIDE + threadIdx.x] = 0.0f;
    auto kind = [&](int op) { const int w = w0[op]; return ((w & 0xff) ? 1 : 0) | (((w >> 26) & 1) << 1); };
    auto dof_of = [&](int op) { return (w0[op] & 0xff) - 1; };
    const int64_t b0 = (int64_t)blockIdx.x * WAVE;
    const unsigned row_off = lane * (unsigned)n * 4u;
    const char *qb = reinterpret_cast<const char *>(q + b0 * n);
    auto q_of = [&](int d) {
        const float x = *reinterpret_cast<const float *>(qb + (d < 0 ? 0 : d) * 4 + row_off);
        return d < 0 ? 0.0f : x;
    };
    float qv[P], cs[P], sn[P];
#pragma unroll
    for (int k = 0; k < P; ++k) qv[k] = q_of(dof_of(k));
    __syncthreads(); // the table is staged
    {
        unsigned r = threadIdx.x / (unsigned)n, c = threadIdx.x - r * (unsigned)n;
        const unsigned step_r = blockDim.x / (unsigned)n, step_c = blockDim.x - step_r * (unsigned)n;
        for (int e = (int)threadIdx.x; e < nn; e += (int)blockDim.x) {
            const int a0 = op_of_dof[r], a1 = op_of_dof[c];
            const int oa = a0 < a1 ? a0 : a1, ob = a0 < a1 ? a1 : a0;
            slot_of[e] = (T::related(oa, ob) ? T::slot(oa, ob) : ZERO_SLOT) * AH_TRI_STRIDE;
            r += step_r; c += step_c;
            if (c >= (unsigned)n) { c -= (unsigned)n; ++r; }
        }
    }
    chain_trig<P>(qv, cs, sn);
    auto row = [&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; };
    auto hout = [&](int oa, int ob, float v) { tri[T::slot(oa, ob) * AH_TRI_STRIDE + lane] = v; };
    Force Fp[L];
    {
        Inertia part;
        crba_arm_hand_sub<P, L>(row, kind, wave, [&](int i) { return q_of(dof_of(P + wave * L + i)); }, hout, Fp, part);
        float *x = lx + wave * (10 * WAVE) + lane;
        x[0] = part.m;
#pragma unroll
        for (int i = 0; i < 3; ++i) x[(1 + i) * WAVE] = part.h[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) x[(4 + i) * WAVE] = part.I[i];
    }
    __syncthreads(); // what every sub-chain hands the palm
    {
        Inertia palm;
        inertia_zero(palm);
        for (int j = 0; j < K; ++j) { // (in sub-chain order on every wavefront: the same floats)
            const float *x = lx + j * (10 * WAVE) + lane;
            Inertia part;
            part.m = x[0];
#pragma unroll
            for (int i = 0; i < 3; ++i) part.h[i] = x[(1 + i) * WAVE];
#pragma unroll
            for (int i = 0; i < 6; ++i) part.I[i] = x[(4 + i) * WAVE];
            inertia_add(palm, part);
        }
        crba_arm_hand_prefix<P, L>(row, kind, wave, qv, cs, sn, palm, Fp, wave, K, hout);
    }
    __syncthreads(); // every entry of the 64 samples is in the triangle
    // The tile's 64 matrices are 64 n^2 consecutive floats, 16-byte aligned as a whole (a single matrix is not, n^2 being odd
    // for most robots): the block writes them as 16 n^2 float4, each of its four floats looked up by (sample, element) — the
    // element's slot from the table, the sample the row of the slot.
    const int n4 = 16 * nn, stride = 4 * (int)blockDim.x;
    const int inc_s = stride / nn, inc_e = stride - inc_s * nn;
    int sm = (4 * (int)threadIdx.x) / nn, e = 4 * (int)threadIdx.x - sm * nn;
    float *g = H + b0 * nn;
    for (int f = (int)threadIdx.x; f < n4; f += (int)blockDim.x) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool wrap = e + c >= nn;
            v[c] = tri[slot_of[wrap ? e + c - nn : e + c] + (wrap ? sm + 1 : sm)];
        }
        store16_wt<NT>(g + 4 * f, make_float4(v[0], v[1], v[2], v[3]));
        sm += inc_s; e += inc_e;
        if (e >= nn) { e -= nn; ++sm; }
    }
}

// ---------------------------------------------------------------------------------------------------
// K7 (reverse-mode inverse dynamics) of the same shapes: drm_sample.hpp rnea_backward_arm_hand — straight-line, nothing stored
// per link (parents recovered from their children on the way back), the palm's motion / force adjoint and the two sums of what
// the sub-chains hand it in 36 LDS floats per sample.  The loop form (rnea_backward_kernel) parks (cos, sin) per link, motion
// and force adjoint per leaf and 36 floats per branch point: 35 KB per wavefront on a Panda with gripper — one wavefront per
// SIMD, 408 us per 2^20 samples against 105 us for the arm alone.  Persistent wavefronts (one row of constant-gradient sums
// each, summed in a fixed order by rnea_backward_reduce_kernel), full 64-row tiles; a ragged tail takes the loop kernel.
// LDS (static): [ table ][ palm : 36 x 64 ][ grad_q | grad_qd | grad_qdd tiles : 64 x (n | 1) ]
// (The gradients leave through the tiles as coalesced stores.  Written straight to the lanes' rows — 4-byte stores 4 n bytes
// apart, which would free 18 KB of LDS at 23 DoF — the same launch took 2 005 instead of 760 us.)
// ---------------------------------------------------------------------------------------------------
// (P = 6, L = 4 alone does not fit two waves' worth of registers — 3 VGPRs went to scratch — and takes one wave per SIMD instead)
template <int P, int L>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu((P == 6 && L == 4) ? 1 : 2, 2)))
    rnea_backward_arm_hand_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, const float *__restrict__ q,
                                  const float *__restrict__ qd, const float *__restrict__ qdd, const float *__restrict__ gtau, int K,
                                  int cap, int n, int n_tiles, int flags, uint64_t param_mask, float *__restrict__ gq,
                                  float *__restrict__ gqd, float *__restrict__ gqdd, float *__restrict__ partials, uint32_t magic_n) {
    constexpr int OPS = P + 4 * L, C_FLOATS = OPS * DRM_OPF_STRIDE;
    constexpr int PALM = 36 * WAVE, TILE = round4(WAVE * pad_odd(OPS));
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + PALM + 3 * TILE];
    const unsigned lane = threadIdx.x;
    const int n_ops = P + K * L, NV = cap * DRM_OPF_STRIDE;
    float *lc = smem, *lp = lc + C_FLOATS + lane, *lt = lc + C_FLOATS + PALM;
    // this wavefront's row of constant-gradient sums lives in global memory (lane 63 alone adds to it, in tile order): the
    // 1.5 KB it would take in LDS is what keeps a Panda with gripper at seven instead of eight wavefronts per CU
    float *prow = partials + (int64_t)blockIdx.x * NV;
    const int Sq = pad_odd(n), region = round4(WAVE * Sq);
    float *lgq = lt, *lgqd = lt + region, *lgqdd = lgqd + region;
    const int32_t *w0 = ops_i + DRM_OPI_W0 * cap;
    int dof[P];
#pragma unroll
    for (int k = 0; k < P; ++k) dof[k] = (w0[k] & 0xff) - 1;
    for (unsigned i = lane; i < (unsigned)n_ops * (DRM_OPF_STRIDE / 4); i += WAVE)
        reinterpret_cast<float4 *>(lc)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    for (int i = (int)lane; i < NV; i += WAVE) prow[i] = 0.0f;
    __builtin_amdgcn_s_waitcnt(0); // (lane 63 adds to its own and the other lanes' zeros below: they must have landed)
    auto kind = [&](int op) { const int w = w0[op]; return ((w & 0xff) ? 1 : 0) | (((w >> 26) & 1) << 1); };
    const bool has_qdd = qdd != nullptr;
    const unsigned row_off = lane * (unsigned)n * 4u, trow = lane * (unsigned)Sq;
#pragma unroll 1
    for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
        const int64_t b0 = (int64_t)tile * WAVE;
        const char *qb = reinterpret_cast<const char *>(q + b0 * n), *qdb = reinterpret_cast<const char *>(qd + b0 * n),
                   *qddb = reinterpret_cast<const char *>(has_qdd ? qdd + b0 * n : q + b0 * n),
                   *gtb = reinterpret_cast<const char *>(gtau + b0 * n);
        // a joint's state where a sweep needs it: one dword load per array at a wave-uniform column (fixed ops: column 0, value
        // dropped) — the lines are in the L2 after the first touch.  (Staging the four input tiles in LDS first was slower:
        // 323 -> 340 us on the Panda with gripper; the wait is not on these loads.)
        auto joint_state = [&](int d, float &a, float &v, float &acc, float &g) {
            const int c = (d < 0 ? 0 : d) * 4;
            const float x = *reinterpret_cast<const float *>(qb + c + row_off), y = *reinterpret_cast<const float *>(qdb + c + row_off),
                        z = *reinterpret_cast<const float *>(qddb + c + row_off), u = *reinterpret_cast<const float *>(gtb + c + row_off);
            a = d < 0 ? 0.0f : x; v = d < 0 ? 0.0f : y; acc = (d < 0 || !has_qdd) ? 0.0f : z; g = d < 0 ? 0.0f : u;
        };
        wave_lds_sync(); // the table is staged; the previous tile's gradients have left the tiles
        rnea_backward_arm_hand<P, L>(
            [&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, kind, K, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING,
            param_mask, gq != nullptr, [&](int k, float &a, float &v, float &acc, float &g) { joint_state(dof[k], a, v, acc, g); },
            [&](int j, int i, float &a, float &v, float &acc, float &g) { joint_state((w0[P + j * L + i] & 0xff) - 1, a, v, acc, g); },
            [&](int op, float a, float v, float acc) {
                const int d = (w0[op] & 0xff) - 1;
                lgq[trow + d] = a; lgqd[trow + d] = v; lgqdd[trow + d] = acc;
            },
            [&](int k, const float *g) { // wave-uniform call: only for the ops param_mask selects
                wave_sums_lane63<DRM_OPF_DAMP + 1>(lane, [&](int j) { return g[j]; },
                                                   [&](int j, float total) { prow[k * DRM_OPF_STRIDE + j] += total; }); // tiles in this wavefront's fixed order
            },
            [&](int i, float x) { lp[i * WAVE] = x; }, [&](int i) { return lp[i * WAVE]; });
        if (gq) {
            wave_lds_sync();
            tile_store<0>(gq + b0 * n, WAVE, n, magic_n, lgq, lane, (n & 1) != 0, true);
            tile_store<0>(gqd + b0 * n, WAVE, n, magic_n, lgqd, lane, (n & 1) != 0, true);
            tile_store<0>(gqdd + b0 * n, WAVE, n, magic_n, lgqdd, lane, (n & 1) != 0, true);
        }
    }
}

// The (P, L) this library is compiled for: prefixes of 5 to 9 ops carrying 2 to 4 sub-chains of L = 1 .. 4 ops — 6- and 7-DoF arms
// with or without a fixed flange, with a two-finger gripper, a three-finger hand or a four-finger hand (the robots it ships:
// Panda with gripper 7 + 2 x 1, Jaco 6 + 3 x 2, iiwa7 + Allegro 7 + 4 x 4 once the host has folded every fixed joint, and
// 9 + 3 x 1, 7 + 3 x 2, 8 + 4 x 4 when learnable links keep the flange / palm an op of its own); anything else keeps the loop
// kernels.  This file is compiled once per L (-DDRM_AH_L=1 .. 4, four objects built in parallel; drm_arm_hand_dispatch.hip
// picks the object by the walk's L): its entry points carry the suffix _l<L>.
#ifndef DRM_AH_L
#define DRM_AH_L 1 /* (a bare `hipcc -c` of this file builds the L = 1 object; the Makefile passes -DDRM_AH_L=1 .. 4) */
#endif
#define DRM_ARM_HAND_SHAPES(X) X(5, DRM_AH_L) X(6, DRM_AH_L) X(7, DRM_AH_L) X(8, DRM_AH_L) X(9, DRM_AH_L)
#define DRM_AH_CAT2(a, b) a##b
#define DRM_AH_CAT(a, b) DRM_AH_CAT2(a, b)
#define DRM_AH_NAME(name) DRM_AH_CAT(DRM_AH_CAT(name, _l), DRM_AH_L)
#define arm_hand_compiled DRM_AH_NAME(arm_hand_compiled)
#define crba_arm_hand_applies DRM_AH_NAME(crba_arm_hand_applies)
#define launch_rnea_arm_hand DRM_AH_NAME(launch_rnea_arm_hand)
#define launch_forward_dynamics_arm_hand DRM_AH_NAME(launch_forward_dynamics_arm_hand)
#define launch_crba_arm_hand DRM_AH_NAME(launch_crba_arm_hand)
#define launch_rnea_backward_arm_hand DRM_AH_NAME(launch_rnea_backward_arm_hand)

static bool shape_of(const drm_walk *w, int &P, int &K, int &L) {
    if (!(w->shape & DRM_WALK_ARM_HAND)) return false;
    P = DRM_WALK_AH_P(w->shape); K = DRM_WALK_AH_K(w->shape); L = DRM_WALK_AH_L(w->shape);
    if (P + K * L != w->n_ops || w->n_ops > AH_MAX_OPS || K > AH_MAX_K || w->n_dofs > w->n_ops) return false;
#define X(p, l) if (P == p && L == l) return true;
    DRM_ARM_HAND_SHAPES(X)
#undef X
    return false;
}

bool arm_hand_compiled(const drm_walk *w) {
#ifdef DRM_NO_ARM_HAND_KERNEL
    return false;
#else
    int P, K, L;
    return shape_of(w, P, K, L);
#endif
}

// rows covered (full tiles), 0 = the call does not qualify
int64_t launch_rnea_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau,
                             hipStream_t s) {
    int P, K, L;
    if (!arm_hand_compiled(w) || !shape_of(w, P, K, L) || B < WAVE || B / WAVE >= 0x7fffffffLL || (((uintptr_t)w->ops_f) & 15u) != 0)
        return 0;
    const uint32_t al = al16(q, AL_Q) | al16(qd, AL_QD) | al16(tau, AL_TAU) | (qdd ? al16(qdd, AL_QDD) : AL_QDD);
    if (al != (AL_Q | AL_QD | AL_QDD | AL_TAU)) return 0;
    const int n_tiles = (int)(B / WAVE), n = w->n_dofs;
#define X(p, l)                                                                                                                   \
    if (P == p && L == l) {                                                                                                       \
        hipLaunchKernelGGL((rnea_arm_hand_kernel<p, l>), dim3((unsigned)n_tiles), dim3(WAVE), 0, s, w->ops_f, w->ops_i, q, qd, qdd, K, \
                           (int)w->capacity, n, flags, tau, div_magic(n));                                                        \
        return (int64_t)n_tiles * WAVE;                                                                                           \
    }
    DRM_ARM_HAND_SHAPES(X)
#undef X
    return 0;
}

// forward dynamics: rows covered (full tiles), 0 = the call does not qualify
int64_t launch_forward_dynamics_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int flags,
                                         float *qdd, hipStream_t s) {
#ifdef DRM_NO_ARM_HAND_FD
    return 0;
#else
    int P, K, L;
    if (!arm_hand_compiled(w) || !shape_of(w, P, K, L) || B < WAVE || B / WAVE >= 0x7fffffffLL || (((uintptr_t)w->ops_f) & 15u) != 0)
        return 0;
    const uint32_t al = al16(q, AL_Q) | al16(qd, AL_QD) | al16(f, AL_QDD) | al16(qdd, AL_TAU);
    if (al != (AL_Q | AL_QD | AL_QDD | AL_TAU)) return 0;
    const int n_tiles = (int)(B / WAVE), n = w->n_dofs;
#define X(p, l)                                                                                                                   \
    if (P == p && L == l) {                                                                                                       \
        hipLaunchKernelGGL((forward_dynamics_arm_hand_kernel<p, l>), dim3((unsigned)n_tiles), dim3(WAVE), 0, s, w->ops_f, w->ops_i, q, qd, \
                           f, K, (int)w->capacity, n, flags, qdd, div_magic(n));                                                  \
        return (int64_t)n_tiles * WAVE;                                                                                           \
    }
    DRM_ARM_HAND_SHAPES(X)
#undef X
    return 0;
#endif
}

// the THROUGHPUT form of the mass matrix for the small shapes (up to CRBA_TREE_MAX_OPS ops): one wavefront walks the whole tree of a
// tile (drm_static.hpp crba_shape_body), taken by launches of at least CRBA_TREE_MIN_TILES tiles
#ifndef DRM_CRBA_TREE_MIN_TILES
#define DRM_CRBA_TREE_MIN_TILES 2048 /* at 1 024 tiles (one per SIMD) the wavefront-per-sub-chain kernels are still ahead: 12.7 vs 13.1 us (Panda with gripper), 16.2 vs 18.4 (Jaco) */
#endif
constexpr int CRBA_TREE_MAX_OPS = 12;
template <int P, int K, int L, bool NT, bool PLAIN>
__global__ void __launch_bounds__(WAVE) crba_arm_hand_tree_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i,
                                                                  const float *__restrict__ q, int cap, int n, int n_tiles,
                                                                  float *__restrict__ H) {
    crba_shape_body<ShapeTree<P, K, L>, NT, PLAIN>(ops_f, ops_i, q, cap, n, n_tiles, H);
}
template <int P, int K, int L>
static bool launch_crba_tree(const drm_walk *w, const float *q, int n_tiles, bool nt, float *H, hipStream_t s) {
    if constexpr (P + K * L <= CRBA_TREE_MAX_OPS) {
        const bool plain = (w->shape & DRM_WALK_NO_PRISMATIC) && w->n_dofs == w->n_ops; // every op a revolute joint
#define DRM_TREE_LAUNCH(NT_, PLAIN_)                                                                                                \
    hipLaunchKernelGGL((crba_arm_hand_tree_kernel<P, K, L, NT_, PLAIN_>), dim3((unsigned)n_tiles), dim3(WAVE), 0, s, w->ops_f, w->ops_i, q, \
                       (int)w->capacity, (int)w->n_dofs, n_tiles, H)
        if (nt) { if (plain) DRM_TREE_LAUNCH(true, true); else DRM_TREE_LAUNCH(true, false); }
        else { if (plain) DRM_TREE_LAUNCH(false, true); else DRM_TREE_LAUNCH(false, false); }
#undef DRM_TREE_LAUNCH
        return true;
    }
    return false;
}

// the mass-matrix kernel has a wavefront per sub-chain
bool crba_arm_hand_applies(const drm_walk *w) {
#ifdef DRM_NO_ARM_HAND_CRBA
    return false;
#else
    int P, K, L;
    return arm_hand_compiled(w) && shape_of(w, P, K, L) && K >= 2 && K <= crba_max_k(L);
#endif
}
int64_t launch_crba_arm_hand(const drm_walk *w, const float *q, int64_t B, float *H, hipStream_t s) {
#ifdef DRM_NO_ARM_HAND_CRBA
    return 0;
#else
    int P, K, L;
    if (!crba_arm_hand_applies(w) || !shape_of(w, P, K, L) || B < WAVE || B / WAVE >= 0x7fffffffLL ||
        (((uintptr_t)w->ops_f | (uintptr_t)q | (uintptr_t)H) & 15u) != 0)
        return 0;
    const int n_tiles = (int)(B / WAVE);
    const bool nt = stream_past_llc((int64_t)n_tiles * WAVE * w->n_dofs * w->n_dofs * 4);
#ifndef DRM_NO_CRBA_TREE
    if (n_tiles >= DRM_CRBA_TREE_MIN_TILES && P + K * L <= CRBA_TREE_MAX_OPS) {
#ifdef DRM_CRBA_TREE_NEVER_NT
        const bool nt = false;
#endif
#define X(p, l)                                                                                                                  \
    if (P == p && L == l) {                                                                                                      \
        if ((K == 2 && launch_crba_tree<p, 2, l>(w, q, n_tiles, nt, H, s)) || (K == 3 && launch_crba_tree<p, 3, l>(w, q, n_tiles, nt, H, s)) || \
            (K == 4 && launch_crba_tree<p, 4, l>(w, q, n_tiles, nt, H, s)))                                                      \
            return (int64_t)n_tiles * WAVE;                                                                                      \
    }
        DRM_ARM_HAND_SHAPES(X)
#undef X
    }
#endif
#define X(p, l)                                                                                                                  \
    if (P == p && L == l) {                                                                                                      \
        if (nt)                                                                                                                  \
            hipLaunchKernelGGL((crba_arm_hand_kernel<p, l, true>), dim3((unsigned)n_tiles), dim3(WAVE * K), 0, s, w->ops_f,      \
                               w->ops_i, q, K, (int)w->capacity, (int)w->n_dofs, H);                                             \
        else                                                                                                                     \
            hipLaunchKernelGGL((crba_arm_hand_kernel<p, l, false>), dim3((unsigned)n_tiles), dim3(WAVE * K), 0, s, w->ops_f,     \
                               w->ops_i, q, K, (int)w->capacity, (int)w->n_dofs, H);                                             \
        return (int64_t)n_tiles * WAVE;                                                                                          \
    }
    DRM_ARM_HAND_SHAPES(X)
#undef X
    return 0;
#endif
}

// reverse-mode inverse dynamics: the full tiles of B; returns the rows done (0: not this walk) and the rows of partial sums
// the launch wrote (one per block) in `partial_rows`
int64_t launch_rnea_backward_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *qdd, const float *gtau,
                                      int64_t B, int flags, uint64_t param_mask, float *gq, float *gqd, float *gqdd, float *partials,
                                      int &partial_rows, hipStream_t s) {
#ifdef DRM_NO_ARM_HAND_BACKWARD
    return 0;
#else
    int P, K, L;
    const uintptr_t ptrs = (uintptr_t)q | (uintptr_t)qd | (uintptr_t)qdd | (uintptr_t)gtau | (uintptr_t)gq | (uintptr_t)gqd |
                           (uintptr_t)gqdd | (uintptr_t)w->ops_f;
    if (!arm_hand_compiled(w) || !shape_of(w, P, K, L) || K > 4 || B < WAVE || B / WAVE >= 0x7fffffffLL || (ptrs & 15u) != 0) return 0;
    const int n_tiles = (int)(B / WAVE);
    // as many one-wavefront blocks as the device holds at once (a grid of 2 048 on a chip that holds 1 792 runs two rounds)
    int resident = 0, rc = DRM_ERR_UNSUPPORTED;
#define X(p, l) if (P == p && L == l) rc = resident_blocks((rnea_backward_arm_hand_kernel<p, l>), WAVE, 0, resident);
    DRM_ARM_HAND_SHAPES(X)
#undef X
    if (rc) return 0;
    if (resident > BWD_MAX_WAVES) resident = BWD_MAX_WAVES;
    const int grid = n_tiles < resident ? n_tiles : resident;
#define X(p, l)                                                                                                                  \
    if (P == p && L == l) {                                                                                                      \
        hipLaunchKernelGGL((rnea_backward_arm_hand_kernel<p, l>), dim3((unsigned)grid), dim3(WAVE), 0, s, w->ops_f, w->ops_i, q, \
                           qd, qdd, gtau, K, (int)w->capacity, (int)w->n_dofs, n_tiles, flags, param_mask, gq, gqd, gqdd,        \
                           partials, div_magic(w->n_dofs));                                                                      \
        partial_rows = grid;                                                                                                     \
        return (int64_t)n_tiles * WAVE;                                                                                          \
    }
    DRM_ARM_HAND_SHAPES(X)
#undef X
    return 0;
#endif
}

} // namespace drm
