// drm_arm_hand.hip — K3 (inverse dynamics) of robots shaped like "an arm that carries a hand" (round 3): P serial prefix
// ops and K serial sub-chains of L ops that all hang off the last prefix op (DRM_WALK_ARM_HAND, include/drm_hip.h).
// Franka Panda with its gripper (P, K, L) = (9, 2, 1), Kinova Jaco (7, 3, 2), KUKA iiwa7 + Allegro hand (8, 4, 4).
//
// These robots are ONE dynamics segment (everything hangs off the moving arm), so the loop-structured kernel
// (drm_rnea.hip rnea_records_kernel) walks a whole 64-sample tile with one wavefront, parks every link's body force
// (LDS or HBM scratch) and decodes two control words per op: 183 / 256 / 437 us at 2^20 samples.  Here the walk is
// straight-line code for the shape (drm_sample.hpp rnea_arm_hand): the prefix is the arm kernels' chain walk, every
// sub-chain runs forward and backward while the palm's motion is in registers, so only the P - 1 prefix forces are parked
// (LDS, 1.5 KB per link and wave) and a CU holds 9-11 wavefronts.  Which DoF column an op drives comes from the walk's
// W0 words (scalar loads): fixed ops and arbitrary DoF numbering need no template parameters beyond (P, L); K is a
// run-time loop count.  Full 64-row tiles only; a ragged tail goes through the loop-structured kernel.
// Build flags as drm_arm_dynamics.hip (kernel-argument preload, no SLP vectoriser).
#include "drm_common.hpp"
#include "drm_sample.hpp"
#include "drm_tree.hpp"
#include "drm_tree_dev.hpp"

namespace drm {

constexpr int AH_MAX_OPS = 32; // P + K L of any compiled shape (sizes the torque tile)

// LDS (static — with a dynamic allocation the compiler's occupancy guess goes wrong and the same code needs 300 registers):
// [ table : AH_MAX_OPS x 32 ][ torques of the sub-chain ops : 16 x 64 ][ parking : (P - 1) x 6 x 64 | tau tile : 64 x (n | 1) ]
template <int P, int L>
__global__ void __launch_bounds__(WAVE)
    rnea_arm_hand_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, const float *__restrict__ q,
                         const float *__restrict__ qd, const float *__restrict__ qdd, int K, int cap, int n, int flags,
                         float *__restrict__ tau, uint32_t magic_n) {
    constexpr int C_FLOATS = AH_MAX_OPS * DRM_OPF_STRIDE, H_FLOATS = 16 * WAVE;
    constexpr int PARK = (P - 1) * 6 * WAVE, TILE = round4(WAVE * pad_odd(AH_MAX_OPS));
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
// one 8-float LDS slot per op (velocities, then U, 1/D, u), sub-chains visited twice instead of stored: no scratch, 14-18 KB
// of LDS per wave.  The loop form (forward_dynamics_aba_kernel) needs HBM scratch and holds five wavefronts per CU.
// LDS (static): [ table ][ slots : P x 8 x 64 | qdd tile : 64 x (n | 1) staged over them once sweep 3 of the prefix is done ]
// ---------------------------------------------------------------------------------------------------
template <int P, int L>
__global__ void __launch_bounds__(WAVE)
    forward_dynamics_arm_hand_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, const float *__restrict__ q,
                                     const float *__restrict__ qd, const float *__restrict__ f, int K, int cap, int n, int flags,
                                     float *__restrict__ qdd, uint32_t magic_n) {
    constexpr int C_FLOATS = AH_MAX_OPS * DRM_OPF_STRIDE;
    constexpr int SLOTS = P * 8 * WAVE, TILE = round4(WAVE * pad_odd(AH_MAX_OPS));
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
// K6 (mass matrix) of the same shapes: drm_tree.hpp crba_arm_hand — a walk that parks nothing (the loop form,
// crba_rows_kernel, keeps the column forces of all the joints below an op in LDS: 35 KB per 64 samples on the arm with a hand,
// three wavefronts per CU).  The output path is the loop form's: a lane scatters its lower-triangle entries into a
// sample-major slice of caller-owned HBM scratch (persistent grid: the slices stay cache-resident), the rows of H are then
// written sample by sample from coalesced reads through an LDS buffer.
// LDS (static): [ table ][ row buffer : 4096 floats ]
// ---------------------------------------------------------------------------------------------------
constexpr int AH_ROWBUF_FLOATS = 4096;
template <int P, int L>
__global__ void __launch_bounds__(WAVE)
    crba_arm_hand_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, const float *__restrict__ q, int K, int cap,
                         int n, int n_tiles, float *__restrict__ H, float *__restrict__ scratch) {
    constexpr int C_FLOATS = AH_MAX_OPS * DRM_OPF_STRIDE;
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + AH_ROWBUF_FLOATS];
    const unsigned lane = threadIdx.x;
    const int n_ops = P + K * L, nn = n * n, ntp = round4(n * (n + 1) / 2);
    float *lc = smem, *lrow = smem + C_FLOATS;
    const int32_t *w0 = ops_i + DRM_OPI_W0 * cap;
    float *tri = scratch + (int64_t)blockIdx.x * ntp * WAVE; // this wave's triangles, SAMPLE-major [64][ntp]
    int dof[P];
#pragma unroll
    for (int k = 0; k < P; ++k) dof[k] = (w0[k] & 0xff) - 1;
    for (unsigned i = lane; i < (unsigned)n_ops * (DRM_OPF_STRIDE / 4); i += WAVE)
        reinterpret_cast<float4 *>(lc)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    auto kind = [&](int op) { const int w = w0[op]; return ((w & 0xff) ? 1 : 0) | (((w >> 26) & 1) << 1); };
    auto dof_of = [&](int op) { return (w0[op] & 0xff) - 1; };
    const int G0 = (AH_ROWBUF_FLOATS / ntp) & ~3, G = G0 > WAVE ? WAVE : G0; // samples per assembly round
    const unsigned step_r = WAVE / (unsigned)n, step_c = WAVE - step_r * (unsigned)n; // 64 = step_r * n + step_c
#pragma unroll 1
    for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
        const int64_t b0 = (int64_t)tile * WAVE;
        for (int i = (int)lane; i < 16 * ntp; i += WAVE) // pairs of joints on different sub-chains stay zero
            reinterpret_cast<float4 *>(tri)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const unsigned row_off = lane * (unsigned)n * 4u;
        const char *qb = reinterpret_cast<const char *>(q + b0 * n);
        auto q_of = [&](int d) {
            const float x = *reinterpret_cast<const float *>(qb + (d < 0 ? 0 : d) * 4 + row_off);
            return d < 0 ? 0.0f : x;
        };
        float qv[P], cs[P], sn[P];
#pragma unroll
        for (int k = 0; k < P; ++k) qv[k] = q_of(dof[k]);
        __syncthreads(); // the zero fill has landed before this lane's entries follow it (and the table is staged)
        chain_trig<P>(qv, cs, sn);
        crba_arm_hand<P, L>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, kind, dof_of, K, qv, cs, sn,
                            [&](int j, int i) { return q_of(dof_of(P + j * L + i)); },
                            [&](int di, int dj, float v) {
                                const int hi = di > dj ? di : dj, lw = di > dj ? dj : di;
                                tri[lane * ntp + tri_index(hi, lw)] = v;
                            });
        __syncthreads(); // every entry of the tile is in scratch
        // the rows of H, G samples per round: their triangles come back with 16-byte loads (twelve in flight per lane) into the
        // row buffer, then every sample's n x n floats leave as 4-byte stores of consecutive lanes — consecutive addresses
        float *g = H + b0 * nn;
        for (int s0 = 0; s0 < WAVE; s0 += G) {
            const int gs = WAVE - s0 < G ? WAVE - s0 : G;
            const float4 *t4 = reinterpret_cast<const float4 *>(tri + s0 * ntp);
            float4 *l4 = reinterpret_cast<float4 *>(lrow);
            const int n4 = gs * (ntp >> 2);
            for (int i0 = 0; i0 < n4; i0 += 12 * WAVE) {
                float4 v[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int i = i0 + u * WAVE + (int)lane;
                    v[u] = i < n4 ? t4[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int i = i0 + u * WAVE + (int)lane;
                    if (i < n4) l4[i] = v[u];
                }
            }
            wave_lds_sync();
            unsigned r = lane / (unsigned)n, c = lane - r * (unsigned)n;
            for (int j = (int)lane; j < nn; j += WAVE) {
                const int hi = (int)(r > c ? r : c), lw = (int)(r > c ? c : r);
                const float *src = lrow + tri_index(hi, lw);
                float *dst = g + (int64_t)s0 * nn + j;
#pragma unroll 4
                for (int gi = 0; gi < gs; ++gi) dst[(int64_t)gi * nn] = src[gi * ntp];
                r += step_r; c += step_c;
                if (c >= (unsigned)n) { c -= (unsigned)n; ++r; }
            }
            wave_lds_sync();
        }
        __syncthreads(); // the slice is free for the next tile
    }
}

// the (P, L) this library is compiled for: those of the robots it ships (robot_data/); anything else keeps the loop kernel
#define DRM_ARM_HAND_SHAPES(X) X(9, 1) X(7, 2) X(8, 4)

static bool shape_of(const drm_walk *w, int &P, int &K, int &L) {
    if (!(w->shape & DRM_WALK_ARM_HAND)) return false;
    P = DRM_WALK_AH_P(w->shape); K = DRM_WALK_AH_K(w->shape); L = DRM_WALK_AH_L(w->shape);
    if (P + K * L != w->n_ops || w->n_ops > AH_MAX_OPS || K * L > 16 || w->n_dofs > w->n_ops) return false;
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

// mass matrix: blocks of the persistent grid for `tiles` full tiles, and the scratch floats they need (0: shape not compiled)
static int crba_arm_hand_grid(const drm_walk *w, int64_t tiles, int &grid) {
    int P, K, L, resident = 0;
    if (!shape_of(w, P, K, L)) return DRM_ERR_UNSUPPORTED;
    int rc = DRM_ERR_UNSUPPORTED;
#define X(p, l) if (P == p && L == l) rc = resident_blocks((crba_arm_hand_kernel<p, l>), WAVE, 0, resident);
    DRM_ARM_HAND_SHAPES(X)
#undef X
    if (rc) return rc;
    grid = (int)(tiles < resident ? tiles : (int64_t)resident);
    return DRM_OK;
}
int64_t crba_arm_hand_scratch_floats(const drm_walk *w, int64_t B) {
#ifdef DRM_NO_ARM_HAND_CRBA
    return 0;
#else
    int grid = 0;
    if (!arm_hand_compiled(w) || B < WAVE || crba_arm_hand_grid(w, B / WAVE, grid)) return 0;
    const int n = w->n_dofs;
    return (int64_t)grid * round4(n * (n + 1) / 2) * WAVE;
#endif
}
int64_t launch_crba_arm_hand(const drm_walk *w, const float *q, int64_t B, float *H, float *scratch, hipStream_t s) {
#ifdef DRM_NO_ARM_HAND_CRBA
    return 0;
#else
    int P, K, L, grid = 0;
    if (!arm_hand_compiled(w) || !shape_of(w, P, K, L) || B < WAVE || B / WAVE >= 0x7fffffffLL || !scratch ||
        (((uintptr_t)w->ops_f | (uintptr_t)q | (uintptr_t)H | (uintptr_t)scratch) & 15u) != 0)
        return 0;
    const int n_tiles = (int)(B / WAVE), n = w->n_dofs;
    if (crba_arm_hand_grid(w, n_tiles, grid)) return 0;
#define X(p, l)                                                                                                                  \
    if (P == p && L == l) {                                                                                                      \
        hipLaunchKernelGGL((crba_arm_hand_kernel<p, l>), dim3((unsigned)grid), dim3(WAVE), 0, s, w->ops_f, w->ops_i, q, K,       \
                           (int)w->capacity, n, n_tiles, H, scratch);                                                            \
        return (int64_t)n_tiles * WAVE;                                                                                          \
    }
    DRM_ARM_HAND_SHAPES(X)
#undef X
    return 0;
#endif
}

} // namespace drm
