// drm_static.hpp — per-robot STRAIGHT-LINE dynamics walks for arbitrary trees (round 4): the code a robot-specific
// translation unit instantiates (differentiable-robot-model_amd/specialize.py writes one per robot: a struct of constexpr
// functions describing the folded whole-tree walk — parent op, DoF column and joint kind of every op — plus the kernels below
// instantiated on it; hipcc turns it into a code object that drm_*'s C ABI launches through the walk's `special` handles).
//
// What the reference does with one Python loop over the links for every robot (robot_model.py:173-193, 262-301), the
// ahead-of-time library covers with straight-line code for a few shape families (7-DoF arms, serial chains, arm + hand,
// fingers) and with LOOP kernels for everything else: control words decoded per op, every per-op record parked in LDS or HBM.
// Here the tree itself is a compile-time constant: every op's parent is a known register set, a branch point keeps its motion
// alive exactly as long as it has children left (the register allocator sees the whole walk), nothing is decoded and nothing
// is parked.  Same per-op arithmetic as the loop kernels (drm_tree.hpp: joint_transform / motion_step / rnea_body_force /
// rnea_link_force_up), so the results agree with them to rounding.
#pragma once

#include <type_traits>
#include <utility>

#ifdef __HIPCC__
#include "drm_common.hpp"
#endif
#include "drm_tree.hpp"

namespace drm {

template <int... I, class F>
DRM_HD void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, k>) for k = 0 .. N-1, in order; every k is a compile-time constant inside f
template <int N, class F>
DRM_HD void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

// Inverse dynamics of the whole tree (robot_model.py:250-375), R = the robot's walk:
//   R::N ops in parent-before-child order, R::parent(k) (op index, -1 = the root link), R::dof(k) (column, -1 = fixed),
//   R::prismatic(k);   row(k) -> op k's constant row;   qf(d, q, qd, qdd);   tau_out(d, tau)
template <class R, class ROW, class QF, class TAU>
DRM_HD void rnea_static_walk(ROW row, int flags, QF qf, TAU tau_out) {
    constexpr int N = R::N;
    const float g = (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f;
    Motion mot[N];
    Force frc[N];
    float cc[N], ss[N], qq[N], qdv[N];
    static_for<N>([&](auto K) {
        constexpr int k = K, par = R::parent(k), dof = R::dof(k);
        constexpr bool pris = R::prismatic(k);
        DRM_RNEA_LINK_FENCE(); // (keeps the constant reads of later ops from being hoisted over this one: register pressure)
        const float *of = row(k);
        float wj = 0.0f, aj = 0.0f;
        qq[k] = 0.0f; cc[k] = 1.0f; ss[k] = 0.0f; qdv[k] = 0.0f;
        if constexpr (dof >= 0) {
            qf(dof, qq[k], wj, aj);
            qdv[k] = wj;
            if constexpr (!pris) sincos_one(qq[k], ss[k], cc[k]);
        }
        const OpFT o = load_ft(of);
        float J[9], t[3];
        joint_transform(o, dof >= 0, pris, qq[k], cc[k], ss[k], J, t);
        Motion from;
        if constexpr (par < 0) motion_root(from, g);
        else from = mot[par];
        motion_step(J, t, wj, aj, pris, from, mot[k]);
        rnea_body_force(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, mot[k], frc[k]);
    });
    static_for<N>([&](auto K) {
        constexpr int k = N - 1 - K, par = R::parent(k), dof = R::dof(k);
        constexpr bool pris = R::prismatic(k);
        DRM_RNEA_LINK_FENCE();
        const float *of = row(k);
        if constexpr (dof >= 0) {
            float tau = pris ? frc[k].la[2][0] : frc[k].la[2][1];
            if (flags & DRM_RNEA_DAMPING) tau += of[DRM_OPF_DAMP] * qdv[k];
            tau_out(dof, tau);
        }
        if constexpr (par >= 0) {
            const OpFT o = load_ft(of);
            float J[9], t[3];
            joint_transform(o, dof >= 0, pris, qq[k], cc[k], ss[k], J, t);
            Force up;
            rnea_link_force_up(J, t, frc[k], up);
#pragma unroll
            for (int i = 0; i < 3; ++i) frc[par].la[i] += up.la[i];
        }
    });
}

// Joint-space inertia matrix of the whole tree by the composite-rigid-body algorithm (robot_model.py:402-450; drm_tree.hpp
// crba_set_walk's organisation): ONE sweep from the leaves to the root.  When the sweep reaches op k, the forces F_c = Ic_c S_c of
// all moving ops c of k's sub-tree are already expressed in k's frame: H[k][c] = S_k . F_c is read off, then all of them (and
// k's own) are moved into the parent's frame by k's transform, which is built once.  The composite inertia of a sub-tree travels
// the same way.  R::below(c, k): op c lies in the sub-tree of op k (c != k);  hout(k, c, v): entry (row op k, column op c) and
// its mirror — the kernel maps the pair to its slot in the triangle (R::slot).
template <class R, class ROW, class QF, class HOUT>
DRM_HD void crba_static_walk(ROW row, QF qf, HOUT hout) {
    constexpr int N = R::N;
    Inertia up[N];   // a sub-tree's composite inertia in its PARENT's frame, from the step of its root until the parent's step
    Force F[N];      // F_c of every moving op, in the frame the sweep has carried it to
    static_for<N>([&](auto K) {
        constexpr int k = N - 1 - K, par = R::parent(k), dof = R::dof(k);
        constexpr bool pris = R::prismatic(k);
        DRM_RNEA_LINK_FENCE();
        const float *of = row(k);
        Inertia tot;
        tot.m = of[DRM_OPF_MASS];
#pragma unroll
        for (int i = 0; i < 3; ++i) tot.h[i] = of[DRM_OPF_MCOM + i];
        tot.I[0] = of[DRM_OPF_IO + 0]; tot.I[1] = of[DRM_OPF_IO + 1]; tot.I[2] = of[DRM_OPF_IO + 2];
        tot.I[3] = of[DRM_OPF_IO + 4]; tot.I[4] = of[DRM_OPF_IO + 5]; tot.I[5] = of[DRM_OPF_IO + 8];
        static_for<N>([&](auto C) { // children, in walk order (the loop kernels add them in the order the sweep meets them)
            constexpr int c = N - 1 - C;
            if constexpr (c > k && R::parent(c) == k) inertia_add(tot, up[c]);
        });
        float q = 0.0f, cs = 1.0f, sn = 0.0f;
        if constexpr (dof >= 0) {
            q = qf(dof);
            if constexpr (!pris) sincos_one(q, sn, cs);
        }
        const OpFT o = load_ft(of);
        float J[9], t[3];
        joint_transform(o, dof >= 0, pris, q, cs, sn, J, t);
        if constexpr (dof >= 0) {
            // F = Ic S_k.  revolute: f = -h x e_z, n = I e_z;  prismatic: f = m e_z, n = h x e_z  (drm_tree.hpp crba_tree_walk)
            if constexpr (!pris) {
                F[k].la[0] = f2_make(-tot.h[1], tot.I[2]);
                F[k].la[1] = f2_make(tot.h[0], tot.I[4]);
                F[k].la[2] = f2_make(0.0f, tot.I[5]);
                hout(K, K, tot.I[5]);
            } else {
                F[k].la[0] = f2_make(0.0f, tot.h[1]);
                F[k].la[1] = f2_make(0.0f, -tot.h[0]);
                F[k].la[2] = f2_make(tot.m, 0.0f);
                hout(K, K, tot.m);
            }
            static_for<N>([&](auto C) {
                constexpr int c = C;
                if constexpr (R::below(c, k) && R::dof(c) >= 0) hout(K, std::integral_constant<int, N - 1 - c>{}, pris ? F[c].la[2][0] : F[c].la[2][1]);
            });
        }
        if constexpr (par >= 0) {
            static_for<N>([&](auto C) {
                constexpr int c = C;
                if constexpr ((c == k || R::below(c, k)) && R::dof(c) >= 0) {
                    Force moved;
                    rnea_link_force_up(J, t, F[c], moved);
                    F[c] = moved;
                }
            });
            inertia_to_parent(J, t, tot, up[k]);
        }
    });
}

// Forward dynamics of the whole tree by the articulated-body algorithm, the reference's own (robot_model.py:487-624) — drm_tree.hpp
// aba_tree_walk's three sweeps with the tree as a compile-time constant: the branch-point motions and the articulated bodies
// travelling towards the root live in registers (a sub-tree's body from the step of its root to its parent's step); what every op
// needs again in a later sweep — its velocity (6 floats, sweep 1 -> 2) and U, 1 / D, u of its joint (8 floats, sweep 2 -> 3) — is
// parked by the caller (LDS).
//   qf(d, q, qd)   fj(d) joint torque   out(d, qdd)   vpark / vunpark(k, Motion)   rpark / runpark(k, rec[8])
template <class R, class ROW, class QF, class FJ, class OUT, class VPARK, class VUNPARK, class RPARK, class RUNPARK>
DRM_HD void aba_static_walk(ROW row, int flags, QF qf, FJ fj, OUT out, VPARK vpark, VUNPARK vunpark, RPARK rpark, RUNPARK runpark) {
    constexpr int N = R::N;
    const float g = (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f;
    auto joint = [&](auto K, float *J, float *t, float &qd) {
        constexpr int k = decltype(K)::value, dof = R::dof(k);
        constexpr bool pris = R::prismatic(k);
        float q = 0.0f, c = 1.0f, s = 0.0f;
        qd = 0.0f;
        if constexpr (dof >= 0) {
            qf(dof, q, qd);
            if constexpr (!pris) sincos_one(q, s, c);
        }
        const OpFT o = load_ft(row(k));
        joint_transform(o, dof >= 0, pris, q, c, s, J, t);
    };
    // ---- sweep 1: velocities ------------------------------------------------------------------------------------------
    {
        Motion mot[N];
        static_for<N>([&](auto K) {
            constexpr int k = K, par = R::parent(k);
            DRM_RNEA_LINK_FENCE();
            float J[9], t[3], qd;
            joint(K, J, t, qd);
            Motion from;
            if constexpr (par < 0) motion_root(from, 0.0f);
            else from = mot[par];
            motion_step(J, t, qd, 0.0f, R::prismatic(k), from, mot[k]);
            vpark(k, mot[k]);
        });
    }
    // ---- sweep 2: articulated inertias and bias forces, leaves -> root ------------------------------------------------------
    {
        ArtBody up[N];
        static_for<N>([&](auto KR) {
            constexpr int k = N - 1 - KR, par = R::parent(k), dof = R::dof(k);
            DRM_RNEA_LINK_FENCE();
            const float *of = row(k);
            Motion vel;
            vunpark(k, vel);      // (acceleration halves zero: the body force below is the bias force v x* (I v))
            ArtBody tot;
            art_from_link(of, tot.I);
            rnea_body_force(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, vel, tot.p);
            static_for<N>([&](auto C) { // children, in the order the leaves -> root sweep meets them
                constexpr int c = N - 1 - C;
                if constexpr (c > k && R::parent(c) == k) art_add(tot, up[c]);
            });
            float J[9], t[3], qd;
            joint(std::integral_constant<int, k>{}, J, t, qd);
            float fjoint = 0.0f;
            if constexpr (dof >= 0) {
                fjoint = fj(dof);
                if (flags & DRM_RNEA_DAMPING) fjoint -= of[DRM_OPF_DAMP] * qd;
            }
            float rec[8];
            aba_eliminate(dof >= 0, R::prismatic(k), J, t, qd, fjoint, vel, tot, rec, par >= 0, up[k]);
            if constexpr (dof >= 0) rpark(k, rec);
        });
    }
    // ---- sweep 3: accelerations, root -> leaves --------------------------------------------------------------------------
    {
        Motion mot[N];
        static_for<N>([&](auto K) {
            constexpr int k = K, par = R::parent(k), dof = R::dof(k);
            constexpr bool pris = R::prismatic(k);
            DRM_RNEA_LINK_FENCE();
            float J[9], t[3], qd;
            joint(K, J, t, qd);
            Motion from;
            if constexpr (par < 0) motion_root(from, g);
            else from = mot[par];
            motion_step(J, t, qd, 0.0f, pris, from, mot[k]); // acceleration halves: a' = X a_parent + c
            if constexpr (dof >= 0) {
                float rec[8];
                runpark(k, rec);
                float dot = 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i) dot += rec[i] * mot[k].va[i][1] + rec[3 + i] * mot[k].wa[i][1];
                const float qdd = (rec[7] - dot) * rec[6];
                out(dof, qdd);
                if constexpr (pris) mot[k].va[2][1] += qdd;
                else mot[k].wa[2][1] += qdd;
            }
        });
    }
}

} // namespace drm

#ifdef __HIPCC__
namespace drm {

// One wavefront per 64-row tile (full tiles; the C ABI sends a ragged tail to the loop kernels).  LDS: [ table : N x 32 ]
// [ tau tile : 64 x (n | 1) ].  Every lane reads its own rows of q / qd / qdd straight into registers (n contiguous floats each).
template <class R>
__device__ __forceinline__ void rnea_static_body(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                                                 const float *__restrict__ qdd, int n_tiles, int flags, float *__restrict__ tau,
                                                 uint32_t magic_n, uint32_t align) {
    constexpr int N = R::N, n = R::NDOF, Sq = pad_odd(n), C_FLOATS = N * DRM_OPF_STRIDE;
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + round4(WAVE * Sq)];
    const unsigned lane = threadIdx.x;
    const int tile = (int)blockIdx.x;
    if (tile >= n_tiles) return;
    float *lc = smem, *lt = smem + C_FLOATS;
    const int64_t b0 = (int64_t)tile * WAVE;
    float4 cv[(C_FLOATS / 4 + WAVE - 1) / WAVE];
#pragma unroll
    for (int it = 0; it < (C_FLOATS / 4 + WAVE - 1) / WAVE; ++it) {
        const int i = (int)lane + it * WAVE;
        cv[it] = reinterpret_cast<const float4 *>(ops_f)[i < C_FLOATS / 4 ? i : C_FLOATS / 4 - 1];
    }
    float qv[n], qdv[n], qddv[n];
    {
        const int64_t row = (b0 + lane) * n;
#pragma unroll
        for (int d = 0; d < n; ++d) qv[d] = q[row + d];
#pragma unroll
        for (int d = 0; d < n; ++d) qdv[d] = qd[row + d];
#pragma unroll
        for (int d = 0; d < n; ++d) qddv[d] = qdd ? qdd[row + d] : 0.0f;
    }
#pragma unroll
    for (int it = 0; it < (C_FLOATS / 4 + WAVE - 1) / WAVE; ++it) {
        const int i = (int)lane + it * WAVE;
        if (i < C_FLOATS / 4) reinterpret_cast<float4 *>(lc)[i] = cv[it];
    }
    wave_lds_sync();
    float *trow = lt + lane * Sq;
    rnea_static_walk<R>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, flags,
                        [&](int d, float &x, float &v, float &a) { x = qv[d]; v = qdv[d]; a = qddv[d]; },
                        [&](int d, float v) { trow[d] = v; });
    wave_lds_sync();
    tile_store<0>(tau + b0 * n, WAVE, n, magic_n, lt, lane, (n & 1) && (align & AL_TAU), (align & AL_TAU) != 0);
}


// The inertia matrices of a 64-row tile: the walk leaves every entry of the related (ancestor, descendant) pairs in a slot of an
// LDS triangle [slot][64 + 1]; the tile's 64 matrices (64 n^2 consecutive floats) then leave as 16-byte stores, every float looked
// up by (element -> slot, sample) — R::SLOT_OF[n^2] (the zero slot for pairs on different branches), as crba_arm_hand_kernel does.
template <class R>
__device__ __forceinline__ void crba_static_body(const float *__restrict__ ops_f, const float *__restrict__ q, int n_tiles,
                                                 float *__restrict__ H, const int *__restrict__ slot_of_global) {
    constexpr int N = R::N, n = R::NDOF, nn = n * n, C_FLOATS = N * DRM_OPF_STRIDE, TRI = WAVE + 1, ZERO = R::SLOTS;
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + (R::SLOTS + 1) * TRI];
    __shared__ int slot_of[nn];
    const unsigned lane = threadIdx.x;
    const int tile = (int)blockIdx.x;
    if (tile >= n_tiles) return;
    float *lc = smem, *tri = smem + C_FLOATS;
    const int64_t b0 = (int64_t)tile * WAVE;
    for (int i = (int)lane; i < C_FLOATS / 4; i += WAVE) reinterpret_cast<float4 *>(lc)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    float qv[n];
    {
        const int64_t row = (b0 + lane) * n;
#pragma unroll
        for (int d = 0; d < n; ++d) qv[d] = q[row + d];
    }
    for (int i = (int)lane; i < nn; i += WAVE) slot_of[i] = slot_of_global[i];
    tri[ZERO * TRI + lane] = 0.0f;
    if (lane == 0) tri[ZERO * TRI + WAVE] = 0.0f;
    wave_lds_sync();
    crba_static_walk<R>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, [&](int d) { return qv[d]; },
                        [&](auto KR, auto CR, float v) { // (reversed op indices as types: slot(k, c) is a compile-time constant)
                            constexpr int k = N - 1 - decltype(KR)::value, c = N - 1 - decltype(CR)::value;
                            tri[R::slot(k, c) * TRI + lane] = v;
                        });
    wave_lds_sync();
    const int n4 = 16 * nn;
    float *g = H + b0 * nn;
    for (int f = (int)lane; f < n4; f += WAVE) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int w = 4 * f + c, sm = w / nn, e = w - sm * nn;
            v[c] = tri[slot_of[e] * TRI + sm];
        }
        store16_wt(g + 4 * f, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// Forward dynamics of a 64-row tile.  LDS: [ table ][ velocities : N x 6 x 64 ][ records : N x 8 x 64 ] — the qdd tile is staged
// over the velocity area once sweep 2 is done with it.
template <class R>
__device__ __forceinline__ void aba_static_body(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                                                const float *__restrict__ f, int n_tiles, int flags, float *__restrict__ qdd,
                                                uint32_t magic_n, uint32_t align) {
    constexpr int N = R::N, n = R::NDOF, Sq = pad_odd(n), C_FLOATS = N * DRM_OPF_STRIDE;
    constexpr int V_FLOATS = N * 6 * WAVE > round4(WAVE * Sq) ? N * 6 * WAVE : round4(WAVE * Sq), R_FLOATS = N * 8 * WAVE;
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + V_FLOATS + R_FLOATS];
    const unsigned lane = threadIdx.x;
    const int tile = (int)blockIdx.x;
    if (tile >= n_tiles) return;
    float *lc = smem, *lv = smem + C_FLOATS, *lr = lv + V_FLOATS;
    const int64_t b0 = (int64_t)tile * WAVE;
    for (int i = (int)lane; i < C_FLOATS / 4; i += WAVE) reinterpret_cast<float4 *>(lc)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    float qv[n], qdv[n], fv[n];
    {
        const int64_t row = (b0 + lane) * n;
#pragma unroll
        for (int d = 0; d < n; ++d) qv[d] = q[row + d];
#pragma unroll
        for (int d = 0; d < n; ++d) qdv[d] = qd[row + d];
#pragma unroll
        for (int d = 0; d < n; ++d) fv[d] = f[row + d];
    }
    wave_lds_sync();
    float acc[n];
    aba_static_walk<R>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, flags,
                       [&](int d, float &x, float &v) { x = qv[d]; v = qdv[d]; }, [&](int d) { return fv[d]; },
                       [&](int d, float v) { acc[d] = v; },
                       [&](int k, const Motion &M) {
#pragma unroll
                           for (int i = 0; i < 3; ++i) { lv[((k * 6 + i) * WAVE) + lane] = M.wa[i][0]; lv[((k * 6 + 3 + i) * WAVE) + lane] = M.va[i][0]; }
                       },
                       [&](int k, Motion &M) {
#pragma unroll
                           for (int i = 0; i < 3; ++i) {
                               M.wa[i] = f2_make(lv[((k * 6 + i) * WAVE) + lane], 0.0f);
                               M.va[i] = f2_make(lv[((k * 6 + 3 + i) * WAVE) + lane], 0.0f);
                           }
                       },
                       [&](int k, const float *rec) {
#pragma unroll
                           for (int i = 0; i < 8; ++i) lr[((k * 8 + i) * WAVE) + lane] = rec[i];
                       },
                       [&](int k, float *rec) {
#pragma unroll
                           for (int i = 0; i < 8; ++i) rec[i] = lr[((k * 8 + i) * WAVE) + lane];
                       });
    wave_lds_sync(); // (the velocity area is free: sweep 2 has read it all)
    float *trow = lv + lane * Sq;
#pragma unroll
    for (int d = 0; d < n; ++d) trow[d] = acc[d];
    wave_lds_sync();
    tile_store<0>(qdd + b0 * n, WAVE, n, magic_n, lv, lane, (n & 1) && (align & AL_TAU), (align & AL_TAU) != 0);
}

} // namespace drm
#endif
