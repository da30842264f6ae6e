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

// The rows of the walk table, as the straight-line walks below read them.  A per-robot translation unit of a CONSTANT model (round 5,
// specialize.source(..., table=...)) defines `struct drm::RobotTable` — the table a constexpr array — and DRM_STATIC_CONST_TABLE before
// it includes this header: every row(k)[i] is then a compile-time constant, products with the robot's exact zeros and ones fold away
// (built with -fno-signed-zeros -ffinite-math-only) and the LDS copy of the table is never read.  Otherwise the rows come from LDS.
#ifdef DRM_STATIC_CONST_TABLE
#define DRM_STATIC_ROW(lc) [](int k) -> const float * { return ::drm::RobotTable::row(k); }
#else
#define DRM_STATIC_ROW(lc) [&](int k) -> const float * { return (lc) + k * DRM_OPF_STRIDE; }
#endif

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
//   PREF: the constants of the NEXT op are read (broadcast LDS reads) into a second register set before the arithmetic of the
//   current one, as in the arm kernels (drm_sample.hpp rnea_chain_trig): the kernels run one wavefront per SIMD, nothing else
//   hides the reads' latency
template <class R, bool PREF = false, class ROW, class QF, class TAU>
DRM_HD void rnea_static_walk(ROW row, int flags, QF qf, TAU tau_out) {
    constexpr int N = R::N;
    const float g = (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f;
    Motion mot[N];
    Force frc[N];
    float cc[N], ss[N], qq[N], qdv[N];
    float buf[2][PREF ? RNEA_ROW_FLOATS : 1];
    if constexpr (PREF) rnea_row_copy(row(0), buf[0]);
    static_for<N>([&](auto K) {
        constexpr int k = K, par = R::parent(k), dof = R::dof(k);
        constexpr bool pris = R::prismatic(k);
        DRM_RNEA_LINK_FENCE(); // (keeps the constant reads of later ops from being hoisted over this one: register pressure)
        if constexpr (PREF) {
            if constexpr (k + 1 < N) rnea_row_copy(row(k + 1), buf[(k + 1) & 1]);
            DRM_RNEA_LINK_FENCE();
        }
        const float *of = PREF ? buf[k & 1] : row(k);
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
    float back[2][PREF ? DRM_OPF_FT_FLOATS : 1], damp2[2];
    if constexpr (PREF) {
        rnea_row_copy(row(N - 1), back[(N - 1) & 1]);
        damp2[(N - 1) & 1] = row(N - 1)[DRM_OPF_DAMP];
    }
    static_for<N>([&](auto K) {
        constexpr int k = N - 1 - K, par = R::parent(k), dof = R::dof(k);
        constexpr bool pris = R::prismatic(k);
        DRM_RNEA_LINK_FENCE();
        if constexpr (PREF) {
            if constexpr (k > 0) {
                rnea_row_copy(row(k - 1), back[(k - 1) & 1]);
                damp2[(k - 1) & 1] = row(k - 1)[DRM_OPF_DAMP];
            }
            DRM_RNEA_LINK_FENCE();
        }
        const float *of = PREF ? back[k & 1] : row(k);
        if constexpr (dof >= 0) {
            float tau = pris ? frc[k].la[2][0] : frc[k].la[2][1];
            if (flags & DRM_RNEA_DAMPING) tau += (PREF ? damp2[k & 1] : of[DRM_OPF_DAMP]) * qdv[k];
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

// The same sweep for a tree that is a compile-time constant while the JOINTS are not: R::N, R::parent(k), R::below(c, k) as above;
// kind(k) -> bit 0: op k moves, bit 1: it slides (wave-uniform run-time values: the control words of the walk); qf(k) -> the joint
// value of op k.  For the ahead-of-time kernels of the SHAPE families (an arm that carries a hand, the fingers of a hand: the tree
// is the template's (P, K, L), which joints move or slide is the robot's).  hout(k, c, v) gets both ops as compile-time constants
// for EVERY related pair of ops; the caller drops the pairs of ops that do not move (kind).
template <class R, class ROW, class KIND, class QF, class HOUT>
DRM_HD void crba_static_walk_kinds(ROW row, KIND kind, QF qf, HOUT hout) {
    constexpr int N = R::N;
    Inertia up[N];
    Force F[N];
    static_for<N>([&](auto K) {
        constexpr int k = N - 1 - K, par = R::parent(k);
        DRM_RNEA_LINK_FENCE();
        const float *of = row(k);
        const int kd = kind(k);
        const bool moves = kd & 1, pris = kd & 2;
        Inertia tot;
        tot.m = of[DRM_OPF_MASS];
#pragma unroll
        for (int i = 0; i < 3; ++i) tot.h[i] = of[DRM_OPF_MCOM + i];
        tot.I[0] = of[DRM_OPF_IO + 0]; tot.I[1] = of[DRM_OPF_IO + 1]; tot.I[2] = of[DRM_OPF_IO + 2];
        tot.I[3] = of[DRM_OPF_IO + 4]; tot.I[4] = of[DRM_OPF_IO + 5]; tot.I[5] = of[DRM_OPF_IO + 8];
        static_for<N>([&](auto C) { // children, in the order the leaves -> root sweep meets them
            constexpr int c = N - 1 - C;
            if constexpr (c > k && R::parent(c) == k) inertia_add(tot, up[c]);
        });
        float q = 0.0f, cs = 1.0f, sn = 0.0f;
        if (moves) {
            q = qf(k);
            if (!pris) sincos_one(q, sn, cs);
        }
        const OpFT o = load_ft(of);
        float J[9], t[3];
        joint_transform(o, moves, pris, q, cs, sn, J, t);
        // F = Ic S_k.  revolute: f = -h x e_z, n = I e_z;  prismatic: f = m e_z, n = h x e_z  (zero for an op that does not move)
        F[k].la[0] = moves ? (pris ? f2_make(0.0f, tot.h[1]) : f2_make(-tot.h[1], tot.I[2])) : f2_bcast(0.0f);
        F[k].la[1] = moves ? (pris ? f2_make(0.0f, -tot.h[0]) : f2_make(tot.h[0], tot.I[4])) : f2_bcast(0.0f);
        F[k].la[2] = moves ? (pris ? f2_make(tot.m, 0.0f) : f2_make(0.0f, tot.I[5])) : f2_bcast(0.0f);
        hout(K, K, pris ? tot.m : tot.I[5]);
        static_for<N>([&](auto C) {
            constexpr int c = C;
            if constexpr (R::below(c, k)) hout(K, std::integral_constant<int, N - 1 - c>{}, pris ? F[c].la[2][0] : F[c].la[2][1]);
        });
        if constexpr (par >= 0) {
            static_for<N>([&](auto C) {
                constexpr int c = C;
                if constexpr (c == k || R::below(c, k)) {
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

// Reverse mode of the inverse dynamics of the whole tree (what torch autograd does for the reference when a loss on
// compute_inverse_dynamics' torques is back-propagated, robot_model.py:305-375, 669-713): drm_sample.hpp rnea_backward_walk with
// the tree as a compile-time constant — the same two sweeps (up: motions and force adjoints; down: the adjoints, every parent's
// motion / tbar recovered from its child's, the sub-tree's total force travelling with the walk), the same per-link arithmetic
// (rnea_link_adjoint_packed / rnea_link_adjoint / rnea_link_param_adjoint), nothing decoded.  What the loop walk parks per link
// (cos, sin) stays in registers; what it keeps in branch-point slots (a branch point's motion and tbar on the way up, the motion
// adjoints and forces its children hand it on the way down) are plain variables whose lifetimes the register allocator sees;
// only the LEAVES' (motion, tbar), which must survive from one sweep to the other, go through the caller (LDS).
//   qf(d, q, qd, qdd);  gtau(d) -> dL/dtau;  gout(d, gq, gqd, gqdd);  param_out(k, g[DRM_OPF_STRIDE]) for ops in param_mask
//   lpark / lunpark(leaf ordinal, Motion, tbar[3])
template <class R, class ROW, class QF, class GT, class GOUT, class PG, class LPARK, class LUNPARK>
DRM_HD void rnea_backward_static_walk(ROW row, int flags, uint64_t param_mask, bool want_gq, QF qf, GT gtau, GOUT gout, PG param_out,
                                      LPARK lpark, LUNPARK lunpark) {
    constexpr int N = R::N;
    const float g = (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f;
    const bool damping = flags & DRM_RNEA_DAMPING;
    float cc[N], ss[N], qq[N];
    // ---- up: motions and force adjoints -------------------------------------------------------------------------------------
    {
        Motion mot[N];
        f2 Tb[N][3];
        static_for<N>([&](auto K) {
            constexpr int k = K, par = R::parent(k), dof = R::dof(k);
            constexpr bool pris = R::prismatic(k);
            DRM_RNEA_LINK_FENCE();
            float wj = 0.0f, aj = 0.0f;
            qq[k] = 0.0f; cc[k] = 1.0f; ss[k] = 0.0f;
            if constexpr (dof >= 0) {
                qf(dof, qq[k], wj, aj);
                if constexpr (!pris) sincos_one(qq[k], ss[k], cc[k]);
            }
            const OpFT o = load_ft(row(k));
            float J[9], t[3];
            joint_transform(o, dof >= 0, pris, qq[k], cc[k], ss[k], J, t);
            if constexpr (par < 0) {
                Motion from;
                motion_root(from, g);
                motion_step(J, t, wj, aj, pris, from, mot[k]);
#pragma unroll
                for (int i = 0; i < 3; ++i) Tb[k][i] = f2_bcast(0.0f);
            } else {
                motion_step(J, t, wj, aj, pris, mot[par], mot[k]);
                tbar_child(J, t, Tb[par], Tb[k]);
            }
            if constexpr (dof >= 0) Tb[k][2][pris ? 0 : 1] += gtau(dof); // tau = S^T f: angular z (revolute), linear z (prismatic)
            if constexpr (R::leaf(k) >= 0) lpark(R::leaf(k), mot[k], Tb[k]);
        });
    }
    // ---- down: adjoints, with the total forces formed on the way ------------------------------------------------------------
    {
        Motion Pm[N], pbn[N];  // op k's parent's motion as k recovered it; the motion adjoint k hands to its parent
        f2 U[N][3];            // the parent's tbar, recovered
        Force up[N];           // the total force of k's sub-tree in the parent's frame
        static_for<N>([&](auto KR) {
            constexpr int k = N - 1 - KR, par = R::parent(k), dof = R::dof(k);
            constexpr bool pris = R::prismatic(k), has_parent = par >= 0;
            constexpr bool chained = k + 1 < N && R::parent(k + 1 < N ? k + 1 : k) == k; // op k + 1 is a child of this link
            DRM_RNEA_LINK_FENCE();
            const float *of = row(k);
            Motion M, B;
            f2 T[3];
            Force carry;
            if constexpr (chained) { // continue from what the child recovered / handed up
                M = Pm[k + 1]; B = pbn[k + 1]; carry = up[k + 1];
#pragma unroll
                for (int i = 0; i < 3; ++i) T[i] = U[k + 1][i];
            } else {                 // a leaf: its parked record, nothing below it
                lunpark(R::leaf(k), M, T);
#pragma unroll
                for (int i = 0; i < 3; ++i) { B.wa[i] = B.va[i] = carry.la[i] = f2_bcast(0.0f); }
            }
            { // the children that do not follow this link directly, summed in the order the sweep met them (the loop walk's slot)
                Motion sB;
                Force sF;
                bool any = false;
                static_for<N>([&](auto CR) {
                    constexpr int c = N - 1 - CR;
                    if constexpr (c > k + 1 && R::parent(c) == k) {
                        if (!any) {
                            sB = pbn[c]; sF = up[c];
                            any = true;
                        } else {
#pragma unroll
                            for (int i = 0; i < 3; ++i) { sB.wa[i] += pbn[c].wa[i]; sB.va[i] += pbn[c].va[i]; sF.la[i] += up[c].la[i]; }
                        }
                    }
                });
                if (any) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) { B.wa[i] += sB.wa[i]; B.va[i] += sB.va[i]; carry.la[i] += sF.la[i]; }
                }
            }
            float J[9], t[3], wj = 0.0f, aj = 0.0f, qdk = 0.0f;
            {
                const OpFT o = load_ft(of);
                joint_transform(o, dof >= 0, pris, qq[k], cc[k], ss[k], J, t);
            }
            if constexpr (dof >= 0) { float qv; qf(dof, qv, wj, aj); qdk = wj; }
            float gtk = 0.0f;
            if constexpr (dof >= 0) gtk = gtau(dof);
            if constexpr (has_parent) {
                motion_parent(J, t, wj, aj, pris, M, Pm[k]);
                f2 x[3] = {T[0], T[1], T[2]};
                x[2][pris ? 0 : 1] -= gtk;
                tbar_parent(J, t, x, U[k]);
            } else {
                motion_root(Pm[k], g);
#pragma unroll
                for (int i = 0; i < 3; ++i) U[k][i] = f2_bcast(0.0f);
            }
            Force tot;
            f2 hgl[3], hga[3];
            rnea_body_force_hg(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, M, tot, hgl, hga);
#pragma unroll
            for (int i = 0; i < 3; ++i) tot.la[i] += carry.la[i];
            const bool learn = (param_mask >> k) & 1u;
            float gq, wjb, ajb;
            auto tbar_to_floats = [](const f2 (&X)[3], float *v) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { v[i] = X[i][0]; v[3 + i] = X[i][1]; }
            };
            if constexpr (!pris) {
                LinkAdjointP A;
                rnea_link_adjoint_packed(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, wj, M, hgl, hga, T, tot, has_parent, B, A);
                gq = A.gq; wjb = A.wjb; ajb = A.ajb;
                pbn[k] = A.pb;
                if (learn) {
                    float gr[DRM_OPF_STRIDE], ub[6];
                    tbar_to_floats(U[k], ub);
                    rnea_link_param_adjoint(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, cc[k], ss[k], M, Pm[k], T, ub, tot,
                                            has_parent, B, A, gr);
                    gr[DRM_OPF_DAMP] = damping ? gtk * qdk : 0.0f;
                    param_out(k, gr);
                }
            } else { // a sliding joint: the scalar form (its motion subspace differs)
                float mo[12], mb[12], pr[12], fb[6], ub[6], tf[6];
                motion_to_floats(M, mo); motion_to_floats(B, mb); motion_to_floats(Pm[k], pr);
                tbar_to_floats(T, fb); tbar_to_floats(U[k], ub);
#pragma unroll
                for (int i = 0; i < 3; ++i) { tf[i] = tot.la[i][0]; tf[3 + i] = tot.la[i][1]; }
                LinkAdjoint A;
                rnea_link_adjoint(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, wj, mo, fb, pr, mb, ub, tf, has_parent, A, true, learn);
                gq = A.gq; wjb = A.wjb; ajb = A.ajb;
                motion_from_floats(A.pb, pbn[k]);
                if (learn) { // J = F and t = trans + F e_z q
                    float gr[DRM_OPF_STRIDE];
#pragma unroll
                    for (int i = 0; i < DRM_OPF_STRIDE; ++i) gr[i] = 0.0f;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        gr[DRM_OPF_FIJ(r, 0)] = A.Jb[r * 3 + 0];
                        gr[DRM_OPF_FIJ(r, 1)] = A.Jb[r * 3 + 1];
                        gr[DRM_OPF_FIJ(r, 2)] = A.Jb[r * 3 + 2] + A.tb[r] * qq[k];
                        gr[DRM_OPF_TI(r)] = A.tb[r];
                        gr[DRM_OPF_MCOM + r] = A.gmc[r];
                    }
                    gr[DRM_OPF_MASS] = A.gm;
#pragma unroll
                    for (int i = 0; i < 9; ++i) gr[DRM_OPF_IO + i] = A.gIo[i];
                    gr[DRM_OPF_DAMP] = damping ? gtk * qdk : 0.0f;
                    param_out(k, gr);
                }
            }
            if constexpr (dof >= 0) {
                if (want_gq) gout(dof, gq, wjb + (damping ? of[DRM_OPF_DAMP] * gtk : 0.0f), ajb);
            }
            if constexpr (has_parent) rnea_link_force_up(J, t, tot, up[k]); // the sub-tree's force in the parent's frame
        });
    }
}

} // namespace drm

#ifdef __HIPCC__
namespace drm {

// One wavefront per 64-row tile (full tiles; the C ABI sends a ragged tail to the loop kernels).  LDS: [ table : N x 32 ]
// [ tau tile : 64 x (n | 1) ].  Every lane reads its own rows of q / qd / qdd straight into registers (n contiguous floats each).
// rows of the NEXT tile straight from global memory into LDS (global_load_lds_dword: no registers in between), element d of the
// lanes' rows at stage[d][lane]; read back with rows_from_stage once the loads have landed (s_waitcnt vmcnt).
// `tile_base`: the tile's first row (wave-uniform), `lane_bytes` = lane * n * 4.  Inline assembly, not
// __builtin_amdgcn_global_load_lds: under this kernel's register pressure (256 VGPR + 256 AGPR) the compiler hands the builtin's
// address operand an AGPR and then rejects its own instruction ("Operand has incorrect register class", ROCm 7.2).
template <int n, int d = 0>
__device__ __forceinline__ void rows_to_stage(const float *__restrict__ tile_base, uint32_t lane_bytes, float *stage) {
    if constexpr (d < n) {
        const uint32_t lds = (uint32_t)(uintptr_t)(stage + d * WAVE); // (the low half of a generic LDS pointer is the LDS address)
        // No instruction offset: the ISA adds `offset:` to the LDS address as well as to the global one (LDS_ADDR = M0 base +
        // inst_offset + lane * 4), so element d rides on the scalar base (tile_base + d: one s_add_u32 / s_addc_u32) instead.
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" ::"v"(lane_bytes), "s"(tile_base + d), "s"(lds) : "memory", "m0");
        rows_to_stage<n, d + 1>(tile_base, lane_bytes, stage);
    }
}
// how many of a kernel's input arrays (n floats per row) to stage for the next tile: `all` of them while four wavefronts' worth of
// LDS (a CU's 160 KB) still holds the kernel's other `base_floats` four times, else `fewer`, else none
constexpr int staged_arrays(int base_floats, int n, int all, int fewer) {
    constexpr int BUDGET = 160 * 1024 / 4 / 4; // floats per wavefront at four wavefronts per CU
    return base_floats + all * n * WAVE + 4 <= BUDGET ? all : (base_floats + fewer * n * WAVE + 4 <= BUDGET ? fewer : 0);
}
template <int n>
__device__ __forceinline__ void rows_from_stage(const float *stage, unsigned lane, float (&v)[n]) {
#pragma unroll
    for (int d = 0; d < n; ++d) v[d] = stage[d * WAVE + lane];
}

template <class R>
__device__ __forceinline__ void rnea_static_body(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                                                 const float *__restrict__ qdd, int n_tiles, int flags, float *__restrict__ tau,
                                                 uint32_t magic_n, uint32_t align) {
    constexpr int N = R::N, n = R::NDOF, Sq = pad_odd(n), C_FLOATS = N * DRM_OPF_STRIDE, T_FLOATS = round4(WAVE * Sq);
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + T_FLOATS + (N >= STATIC_LONE_OPS ? 3 * n * WAVE : 4)];
    const unsigned lane = threadIdx.x;
    if ((int)blockIdx.x >= n_tiles) return;
    float *lc = smem, *lt = smem + C_FLOATS, *stage = lt + T_FLOATS;
    float4 cv[(C_FLOATS / 4 + WAVE - 1) / WAVE];
#pragma unroll
    for (int it = 0; it < (C_FLOATS / 4 + WAVE - 1) / WAVE; ++it) {
        const int i = (int)lane + it * WAVE;
        cv[it] = reinterpret_cast<const float4 *>(ops_f)[i < C_FLOATS / 4 ? i : C_FLOATS / 4 - 1];
    }
    // The grid is what the device holds at once (or one block per tile when that is fewer): a wavefront walks tiles blockIdx.x,
    // + gridDim.x, ... and, before it walks one, starts the NEXT tile's rows on their way from global memory into an LDS staging
    // area — with one wavefront per SIMD nothing else would overlap a tile's loads with arithmetic, and the register file
    // (256 VGPR + up to 256 AGPR here) has no room for a second set of inputs.
    float qv[n], qdv[n], qddv[n];
    {
        const int64_t row = ((int64_t)blockIdx.x * WAVE + lane) * n;
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
    float *trow = lt + lane * Sq;
#ifndef DRM_STATIC_PREF
#define DRM_STATIC_PREF 1
#endif
    constexpr bool LONE = N >= STATIC_LONE_OPS; // (drm_common.hpp: the latency forms, for walks that run one wavefront per SIMD anyway)
#pragma unroll 1
    for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
        const int next = LONE ? tile + (int)gridDim.x : n_tiles; // (small robots: one tile per block, launched as such)
        wave_lds_sync(); // (the table is in LDS; the previous tile's staged torques and next rows have been read)
        if (next < n_tiles) {
            const int64_t base = (int64_t)next * WAVE * n;
            const uint32_t off = lane * (uint32_t)(n * sizeof(float));
            rows_to_stage<n>(q + base, off, stage);
            rows_to_stage<n>(qd + base, off, stage + n * WAVE);
            if (qdd) rows_to_stage<n>(qdd + base, off, stage + 2 * n * WAVE);
        }
        rnea_static_walk<R, LONE && DRM_STATIC_PREF != 0>(DRM_STATIC_ROW(lc), flags,
                                                  [&](int d, float &x, float &v, float &a) { x = qv[d]; v = qdv[d]; a = qddv[d]; },
                                                  [&](int d, float v) { trow[d] = v; });
        if constexpr (LONE) __builtin_amdgcn_s_waitcnt(0); // the next rows have landed (issued a whole walk ago: nothing to wait for in practice)
        wave_lds_sync();
        if (next < n_tiles) {
            rows_from_stage<n>(stage, lane, qv);
            rows_from_stage<n>(stage + n * WAVE, lane, qdv);
            if (qdd) rows_from_stage<n>(stage + 2 * n * WAVE, lane, qddv);
        }
        tile_store<0>(tau + (int64_t)tile * WAVE * n, WAVE, n, magic_n, lt, lane, (n & 1) && (align & AL_TAU), (align & AL_TAU) != 0);
    }
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
    crba_static_walk<R>(DRM_STATIC_ROW(lc), [&](int d) { return qv[d]; },
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

// Forward dynamics of a 64-row tile.  LDS: [ table ][ per-op slots : N x 8 x 64 ] — a slot first holds the op's velocity (6 floats,
// sweep 1 -> 2) and, once sweep 2 has read it, the op's joint record (8 floats, sweep 2 -> 3) in its place; the qdd tile is staged
// over the slots when sweep 3 is done with them.  (Velocities and records side by side took 52 KB for Fetch: three wavefronts per CU.)
template <class R>
__device__ __forceinline__ void aba_static_body(const float *__restrict__ ops_f, const float *__restrict__ q, const float *__restrict__ qd,
                                                const float *__restrict__ f, int n_tiles, int flags, float *__restrict__ qdd,
                                                uint32_t magic_n, uint32_t align) {
    constexpr int N = R::N, n = R::NDOF, Sq = pad_odd(n), C_FLOATS = N * DRM_OPF_STRIDE;
    constexpr int S_FLOATS = N * 8 * WAVE > round4(WAVE * Sq) ? N * 8 * WAVE : round4(WAVE * Sq);
    // arrays of the next tile that are staged in LDS: all three while four wavefronts still fit a CU's LDS, else f is read at the
    // top of its own tile (the walk needs it in sweep 2 only), else nothing is staged
    constexpr int STAGED = N >= STATIC_LONE_OPS ? staged_arrays(C_FLOATS + S_FLOATS, n, 3, 2) : 0; // (small robots: one tile per block)
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + S_FLOATS + STAGED * n * WAVE + 4];
    const unsigned lane = threadIdx.x;
    if ((int)blockIdx.x >= n_tiles) return;
    float *lc = smem, *ls = smem + C_FLOATS, *stage = ls + S_FLOATS;
    for (int i = (int)lane; i < C_FLOATS / 4; i += WAVE) reinterpret_cast<float4 *>(lc)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    // persistent wavefronts, the next tile's rows staged in LDS while this one is walked (see rnea_static_body)
    float qv[n], qdv[n], fv[n];
    {
        const int64_t row = ((int64_t)blockIdx.x * WAVE + lane) * n;
#pragma unroll
        for (int d = 0; d < n; ++d) qv[d] = q[row + d];
#pragma unroll
        for (int d = 0; d < n; ++d) qdv[d] = qd[row + d];
#pragma unroll
        for (int d = 0; d < n; ++d) fv[d] = f[row + d];
    }
#pragma unroll 1
    for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
        const int next = tile + (int)gridDim.x;
        wave_lds_sync(); // (the table is in LDS; the previous tile's staged accelerations and next rows have been read)
        if (next < n_tiles && STAGED >= 2) {
            const int64_t base = (int64_t)next * WAVE * n;
            const uint32_t off = lane * (uint32_t)(n * sizeof(float));
            rows_to_stage<n>(q + base, off, stage);
            rows_to_stage<n>(qd + base, off, stage + n * WAVE);
            if constexpr (STAGED >= 3) rows_to_stage<n>(f + base, off, stage + 2 * n * WAVE);
        }
        if (STAGED < 3 && tile != (int)blockIdx.x) { // (the first tile's rows were read above)
            const int64_t row = ((int64_t)tile * WAVE + lane) * n;
            if constexpr (STAGED < 2) {
#pragma unroll
                for (int d = 0; d < n; ++d) qv[d] = q[row + d];
#pragma unroll
                for (int d = 0; d < n; ++d) qdv[d] = qd[row + d];
            }
#pragma unroll
            for (int d = 0; d < n; ++d) fv[d] = f[row + d];
        }
        float acc[n];
        aba_static_walk<R>(DRM_STATIC_ROW(lc), flags,
                           [&](int d, float &x, float &v) { x = qv[d]; v = qdv[d]; }, [&](int d) { return fv[d]; },
                           [&](int d, float v) { acc[d] = v; },
                           [&](int k, const Motion &M) {
#pragma unroll
                               for (int i = 0; i < 3; ++i) { ls[((k * 8 + i) * WAVE) + lane] = M.wa[i][0]; ls[((k * 8 + 3 + i) * WAVE) + lane] = M.va[i][0]; }
                           },
                           [&](int k, Motion &M) {
#pragma unroll
                               for (int i = 0; i < 3; ++i) {
                                   M.wa[i] = f2_make(ls[((k * 8 + i) * WAVE) + lane], 0.0f);
                                   M.va[i] = f2_make(ls[((k * 8 + 3 + i) * WAVE) + lane], 0.0f);
                               }
                           },
                           [&](int k, const float *rec) { // (op k's velocity has been read: sweep 2 unparks it before it eliminates the joint)
#pragma unroll
                               for (int i = 0; i < 8; ++i) ls[((k * 8 + i) * WAVE) + lane] = rec[i];
                           },
                           [&](int k, float *rec) {
#pragma unroll
                               for (int i = 0; i < 8; ++i) rec[i] = ls[((k * 8 + i) * WAVE) + lane];
                           });
        if constexpr (STAGED > 0) __builtin_amdgcn_s_waitcnt(0); // the next rows have landed (issued a whole walk ago)
        wave_lds_sync();               // (the slots are free: sweep 3 has read every record)
        float *trow = ls + lane * Sq;
#pragma unroll
        for (int d = 0; d < n; ++d) trow[d] = acc[d];
        if (next < n_tiles && STAGED >= 2) {
            rows_from_stage<n>(stage, lane, qv);
            rows_from_stage<n>(stage + n * WAVE, lane, qdv);
            if constexpr (STAGED >= 3) rows_from_stage<n>(stage + 2 * n * WAVE, lane, fv);
        }
        wave_lds_sync();
        tile_store<0>(qdd + (int64_t)tile * WAVE * n, WAVE, n, magic_n, ls, lane, (n & 1) && (align & AL_TAU), (align & AL_TAU) != 0);
    }
}

// Reverse-mode inverse dynamics of 64-row tiles: persistent wavefronts (one per block; the grid is what the device holds at
// once), each with its own row of constant-gradient sums — tiles added in the wavefront's fixed order, the rows reduced by
// rnea_backward_reduce_kernel afterwards, as for every backward kernel.  LDS: [ table : N x 32 ][ sums : CAP x 32 ]
// [ leaf records : LEAVES x 18 x 64 ].  q / qd / qdd / grad_tau and the three gradients of a sample stay in its lane's registers
// (n contiguous floats per lane and array on the way in and out, as in rnea_backward_arm_kernel).
template <class R, int CAP>
__device__ __forceinline__ void rnea_backward_static_body(const float *__restrict__ ops_f, const float *__restrict__ q,
                                                          const float *__restrict__ qd, const float *__restrict__ qdd,
                                                          const float *__restrict__ gtau, int n_tiles, int flags, uint64_t param_mask,
                                                          float *__restrict__ gq, float *__restrict__ gqd, float *__restrict__ gqdd,
                                                          float *__restrict__ partials) {
    constexpr int N = R::N, n = R::NDOF, C_FLOATS = N * DRM_OPF_STRIDE, NV = CAP * DRM_OPF_STRIDE, LEAF = 18;
    static_assert(CAP >= N, "the table's pitch covers the walk");
    // arrays of the next tile staged in LDS (see aba_static_body): q, qd, qdd, grad_tau — or without grad_tau — or none.
    // NONE by default: measured on Fetch, staging made this kernel 5 % SLOWER (260 -> 273 us at 2^20 rows; 256 VGPR + 246 AGPR,
    // the walk is 83 % of the launch already) where it gained 4-10 % on the forward kernels.  -DDRM_STATIC_STAGE_BACKWARD=1 for A/B runs.
#ifndef DRM_STATIC_STAGE_BACKWARD
#define DRM_STATIC_STAGE_BACKWARD 0
#endif
    constexpr int STAGED = DRM_STATIC_STAGE_BACKWARD ? staged_arrays(C_FLOATS + NV + R::LEAVES * LEAF * WAVE, n, 4, 3) : 0;
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + NV + R::LEAVES * LEAF * WAVE + STAGED * n * WAVE + 4];
    const unsigned lane = threadIdx.x;
    float *lc = smem, *lacc = smem + C_FLOATS, *lleaf = lacc + NV + lane, *stage = lacc + NV + R::LEAVES * LEAF * WAVE;
    for (int i = (int)lane; i < C_FLOATS / 4; i += WAVE) reinterpret_cast<float4 *>(lc)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    for (int i = (int)lane; i < NV; i += WAVE) lacc[i] = 0.0f;
    wave_lds_sync();
    // the next tile's rows are staged in LDS while this one is walked (see rnea_static_body)
    float qv[n], qdv[n], qddv[n], gtv[n];
    if (STAGED > 0 && (int)blockIdx.x < n_tiles) {
        const int64_t r0 = ((int64_t)blockIdx.x * WAVE + lane) * n;
#pragma unroll
        for (int d = 0; d < n; ++d) qv[d] = q[r0 + d];
#pragma unroll
        for (int d = 0; d < n; ++d) qdv[d] = qd[r0 + d];
#pragma unroll
        for (int d = 0; d < n; ++d) qddv[d] = qdd ? qdd[r0 + d] : 0.0f;
#pragma unroll
        for (int d = 0; d < n; ++d) gtv[d] = gtau[r0 + d];
    }
#pragma unroll 1
    for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
        const int64_t r0 = ((int64_t)tile * WAVE + lane) * n;
        const int next = tile + (int)gridDim.x;
        if constexpr (STAGED > 0) wave_lds_sync(); // (the staged rows of this tile have been read)
        if (next < n_tiles && STAGED >= 3) {
            const int64_t base = (int64_t)next * WAVE * n;
            const uint32_t off = lane * (uint32_t)(n * sizeof(float));
            rows_to_stage<n>(q + base, off, stage);
            rows_to_stage<n>(qd + base, off, stage + n * WAVE);
            if (qdd) rows_to_stage<n>(qdd + base, off, stage + 2 * n * WAVE);
            if constexpr (STAGED >= 4) rows_to_stage<n>(gtau + base, off, stage + 3 * n * WAVE);
        }
        if (STAGED == 0 || (STAGED < 4 && tile != (int)blockIdx.x)) { // (staging: the first tile's rows were read above)
            if constexpr (STAGED < 3) {
#pragma unroll
                for (int d = 0; d < n; ++d) qv[d] = q[r0 + d];
#pragma unroll
                for (int d = 0; d < n; ++d) qdv[d] = qd[r0 + d];
#pragma unroll
                for (int d = 0; d < n; ++d) qddv[d] = qdd ? qdd[r0 + d] : 0.0f;
            }
#pragma unroll
            for (int d = 0; d < n; ++d) gtv[d] = gtau[r0 + d];
        }
        float gqv[n], gqdv[n], gqddv[n];
#pragma unroll
        for (int d = 0; d < n; ++d) gqv[d] = gqdv[d] = gqddv[d] = 0.0f;
        rnea_backward_static_walk<R>(
            DRM_STATIC_ROW(lc), flags, param_mask, gq != nullptr,
            [&](int d, float &x, float &v, float &a) { x = qv[d]; v = qdv[d]; a = qddv[d]; }, [&](int d) { return gtv[d]; },
            [&](int d, float x, float v, float a) { gqv[d] = x; gqdv[d] = v; gqddv[d] = a; },
            [&](int k, const float *g) { // wave-uniform call: only for the ops param_mask selects
                wave_sums_lane63<DRM_OPF_DAMP + 1>(lane, [&](int j) { return g[j]; },
                                                   [&](int j, float total) { lacc[k * DRM_OPF_STRIDE + j] += total; }); // tiles in this wavefront's fixed order
            },
            [&](int leaf, const Motion &M, const f2 (&T)[3]) {
                float *r = lleaf + leaf * (LEAF * WAVE);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    r[i * WAVE] = M.wa[i][0]; r[(3 + i) * WAVE] = M.wa[i][1]; r[(6 + i) * WAVE] = M.va[i][0]; r[(9 + i) * WAVE] = M.va[i][1];
                    r[(12 + i) * WAVE] = T[i][0]; r[(15 + i) * WAVE] = T[i][1];
                }
            },
            [&](int leaf, Motion &M, f2 (&T)[3]) {
                const float *r = lleaf + leaf * (LEAF * WAVE);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    M.wa[i] = f2_make(r[i * WAVE], r[(3 + i) * WAVE]);
                    M.va[i] = f2_make(r[(6 + i) * WAVE], r[(9 + i) * WAVE]);
                    T[i] = f2_make(r[(12 + i) * WAVE], r[(15 + i) * WAVE]);
                }
            });
        if constexpr (STAGED > 0) {
            __builtin_amdgcn_s_waitcnt(0); // the next rows have landed (issued a whole walk ago)
            wave_lds_sync();
        }
        if (next < n_tiles && STAGED >= 3) {
            rows_from_stage<n>(stage, lane, qv);
            rows_from_stage<n>(stage + n * WAVE, lane, qdv);
            if (qdd) rows_from_stage<n>(stage + 2 * n * WAVE, lane, qddv);
            if constexpr (STAGED >= 4) rows_from_stage<n>(stage + 3 * n * WAVE, lane, gtv);
        }
        if (gq) {
#pragma unroll
            for (int d = 0; d < n; ++d) gq[r0 + d] = gqv[d];
#pragma unroll
            for (int d = 0; d < n; ++d) gqd[r0 + d] = gqdv[d];
#pragma unroll
            for (int d = 0; d < n; ++d) gqdd[r0 + d] = gqddv[d];
        }
    }
    wave_lds_sync();
    float *prow = partials + (int64_t)blockIdx.x * NV;
    for (int i = (int)lane; i < NV; i += WAVE) prow[i] = lacc[i];
}

// The tree of a SHAPE family as compile-time constants: ops 0 .. P-1 a serial chain (op 0 off the root; P = 0: none), then K serial
// sub-chains of L ops, each hanging off op P-1 (off the root when P = 0) — DRM_WALK_ARM_HAND / DRM_WALK_FINGERS of include/drm_hip.h.
// Which ops move or slide is the robot's (run-time kinds).  Triangle slots: every pair (oa <= ob) of ops on a common root path.
template <int P_, int K_, int L_>
struct ShapeTree {
    static constexpr int P = P_, K = K_, L = L_, N = P + K * L;
    static constexpr int parent(int k) { return k < P ? k - 1 : ((k - P) % L == 0 ? P - 1 : k - 1); }
    static constexpr bool below(int c, int k) { return c > k && (k < P || (c - P) / L == (k - P) / L); }
    static constexpr int PPN = P * (P + 1) / 2, SUBN = L * P + L * (L + 1) / 2, SLOTS = PPN + K * SUBN;
    static constexpr bool related(int oa, int ob) { return oa < P || (oa - P) / L == (ob - P) / L; } // (oa <= ob)
    static constexpr int slot(int oa, int ob) {
        if (ob < P) return ob * (ob + 1) / 2 + oa;
        const int j = (ob - P) / L, i = ob - P - j * L;
        return PPN + j * SUBN + i * P + i * (i + 1) / 2 + (oa < P ? oa : oa - j * L);
    }
    static constexpr int TRI = WAVE + 1;
    static constexpr size_t LDS_BYTES = sizeof(float) * (N * DRM_OPF_STRIDE + (SLOTS + 1) * TRI) + sizeof(int) * (N + N * N);
};

// Inertia matrices of 64-row tiles for a shape family, ONE wavefront per tile walking the whole tree (crba_static_walk_kinds): the
// throughput form.  (crba_arm_hand_kernel / crba_tree_kernel give every sub-chain a wavefront of its own, each replaying the prefix
// — the latency form: at 2^20 rows a Panda with gripper takes 133 us there and 86 us here, a TriFinger 2x.)  Entries go to the LDS
// triangle, the tile's 64 matrices leave as 16-byte stores looked up by (element -> slot, sample), as in crba_arm_hand_kernel.
// PLAIN: every op is a revolute joint (the host knows: DRM_WALK_NO_PRISMATIC and n_dofs == n_ops) — the kinds are the constant 1
// and the sliding / fixed forms of every op fold away (Panda with gripper, revolute fingers: 109 -> 9x us at 2^20 rows).
template <class T, bool NT, bool PLAIN = false>
__device__ __forceinline__ void crba_shape_body(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, const float *__restrict__ q,
                                                int cap, int n, int n_tiles, float *__restrict__ H) {
    constexpr int N = T::N, C_FLOATS = N * DRM_OPF_STRIDE, TRI = T::TRI, ZERO = T::SLOTS;
    __shared__ __attribute__((aligned(16))) float smem[C_FLOATS + (T::SLOTS + 1) * TRI];
    __shared__ int op_of_dof[N];
    __shared__ int slot_of[N * N];
    const unsigned lane = threadIdx.x;
    const int tile = (int)blockIdx.x;
    if (tile >= n_tiles) return;
    const int nn = n * n;
    float *lc = smem, *tri = smem + C_FLOATS;
    const int32_t *w0 = ops_i + DRM_OPI_W0 * cap;
    for (int i = (int)lane; i < C_FLOATS / 4; i += WAVE) reinterpret_cast<float4 *>(lc)[i] = reinterpret_cast<const float4 *>(ops_f)[i];
    if ((int)lane < N) {
        const int d = (w0[lane] & 0xff) - 1;
        if (d >= 0) op_of_dof[d] = (int)lane;
    }
    tri[ZERO * TRI + lane] = 0.0f;
    if (lane == 0) tri[ZERO * TRI + WAVE] = 0.0f;
    const int64_t b0 = (int64_t)tile * WAVE;
    const float *qrow = q + (b0 + lane) * n;
    int kinds[N];
    float qv[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { // (wave-uniform control words; a joint value per moving op)
        const int w = w0[k], d = (w & 0xff) - 1;
        kinds[k] = PLAIN ? 1 : ((d >= 0 ? 1 : 0) | (((w >> 26) & 1) << 1));
        qv[k] = qrow[d < 0 ? 0 : d];
    }
    wave_lds_sync();
    for (int e = (int)lane; e < nn; e += WAVE) {
        const int r = e / n, c = e - r * n;
        const int a0 = op_of_dof[r], a1 = op_of_dof[c];
        const int oa = a0 < a1 ? a0 : a1, ob = a0 < a1 ? a1 : a0;
        slot_of[e] = (T::related(oa, ob) ? T::slot(oa, ob) : ZERO) * TRI;
    }
    crba_static_walk_kinds<T>(DRM_STATIC_ROW(lc), [&](int k) { return PLAIN ? 1 : kinds[k]; },
                              [&](int k) { return qv[k]; },
                              [&](auto KR, auto CR, float v) { // (reversed op indices as types: the slot is a compile-time constant)
                                  constexpr int k = N - 1 - decltype(KR)::value, c = N - 1 - decltype(CR)::value;
                                  tri[T::slot(k, c) * TRI + lane] = v;
                              });
    wave_lds_sync();
    const int n4 = 16 * nn;
    float *g = H + b0 * nn;
    for (int f = (int)lane; f < n4; f += WAVE) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int w = 4 * f + c, sm = w / nn, e = w - sm * nn;
            v[c] = tri[slot_of[e] + sm];
        }
        store16_wt<NT>(g + 4 * f, make_float4(v[0], v[1], v[2], v[3]));
    }
}

} // namespace drm
#endif
