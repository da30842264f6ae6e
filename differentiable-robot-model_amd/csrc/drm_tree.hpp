// drm_tree.hpp — per-sample arithmetic of the LOOP-STRUCTURED forward walks: FK of many targets, FK + Jacobian of one
// chain, RNEA, the composite-rigid-body inertia matrix and forward dynamics of ANY robot (trees, hands, arms with
// grippers; revolute / continuous and prismatic joints; any number of links).
//
// One lane owns one sample; a walk is a LOOP over the n_ops links of the flattened tree (flatten.py build_walk), one
// pair of wide control words (DRM_OPI_W0 / W1, include/drm_hip.h) decoded per iteration, the per-link constants read
// from the op table (LDS in the kernels), nothing indexed by a compile-time op number.  The straight-line walks this
// replaces were compiled per capacity (4 .. 32 links): rnea<24> held 502 registers, crba<CAP> was O(CAP^2) code, the
// library took 2.5 minutes to build and stopped at 32 links.  State that outlives an iteration is handed to the caller
// through callbacks (LDS in the kernels, plain arrays in tests/host_emu):
//   save slots     the state of a branch point, read by its later children (one slot per open branch point)
//   per-op records what a second sweep needs from the first (body force + cos / sin of the joint angle)
// Every joint moves about / along +z of its stored frame (flatten.py folds the axis into the constants): a revolute
// joint turns it by q (rigid_body.py:146-156), a prismatic joint slides it by q (the reference models those as
// revolute, robot_model.py:122-126 — SURVEY.md §8 f4 asks for the correct model; reference_compat restores the other).
//
// Like drm_sample.hpp this header compiles with g++ for tests/host_emu (test infrastructure only).
#pragma once

#include "drm_sample.hpp"

namespace drm {

struct OpCtl {
    int dof, src, save, out, perm, parent; // DRM_OPI_* meanings; parent = op index of the parent link, -1 = root
    bool child_next, padding, prismatic;
};
DRM_HD OpCtl decode_ctl(int w0, int w1) {
    OpCtl c;
    c.dof = (w0 & 0xff) - 1;
    c.src = ((w0 >> 8) & 0xff) - 2;
    c.save = ((w0 >> 16) & 0xff) - 1;
    c.child_next = (w0 >> 24) & 1;
    c.padding = (w0 >> 25) & 1;
    c.prismatic = (w0 >> 26) & 1;
    c.perm = (w0 >> 27) & 7;
    c.out = (w1 & 0xffff) - 1;
    c.parent = ((w1 >> 16) & 0xffff) - 1;
    return c;
}

// control-word access of the walks: CTL provides raw(k, r0, r1) (request; plain loads) and uniform(r) (the wave-uniform
// value of a requested word); ctl_words() is the unpipelined form
template <class CTL>
DRM_HD void ctl_words(const CTL &ctl, int k, int &w0, int &w1) {
    int r0, r1;
    ctl.raw(k, r0, r1);
    w0 = ctl.uniform(r0);
    w1 = ctl.uniform(r1);
}

// one link of the pose chain on packed pairs; prismatic joints slide along the NEW z axis: p += R e_z q
DRM_HD void pose_step(const OpPairs &o, const OpCtl &ct, float q, bool from_root, PoseP &cur) {
    float c = 1.0f, s = 0.0f;
    if (ct.dof >= 0 && !ct.prismatic) sincos_one(q, s, c);
    f2 J01[3];
    joint_pairs(o, c, s, J01); // c = 1, s = 0 gives the F pairs back exactly
    if (from_root) compose_pairs_root(J01, o, cur);
    else compose_pairs(cur, J01, o, cur);
    if (ct.dof >= 0 && ct.prismatic) {
#pragma unroll
        for (int r = 0; r < 3; ++r) cur.B[r][1] += cur.B[r][0] * q;
    }
}

// ---------------------------------------------------------------------------
// FK of many targets over a (possibly branching) walk (robot_model.py:139-195 poses only, 223-248, 197-221).
//   ctl(k, w0, w1)  the two control words of op k (wave-uniform)      row(k)  op k's constant row
//   qf(d)           joint value of DoF d
//   slot_save(s, PoseP) / slot_load(s, PoseP&)   branch-point poses
//   emit(t, p[3], q[4])   world position and xyzw quaternion of target slot t
// ---------------------------------------------------------------------------
template <class CTL, class ROW, class QF, class SAVE, class LOAD, class EMIT>
DRM_HD void fk_tree_walk(int n_ops, CTL ctl, ROW row, QF qf, SAVE slot_save, LOAD slot_load, EMIT emit) {
    PoseP cur;
#pragma unroll
    for (int c = 0; c < 3; ++c) { cur.A[c] = f2_make(c == 0, c == 1); cur.B[c] = f2_make(c == 2, 0.0f); }
#pragma unroll 1
    for (int k = 0; k < n_ops; ++k) {
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        const OpPairs o = load_pairs(row(k));
        const float q = ct.dof >= 0 ? qf(ct.dof) : 0.0f;
        if (ct.src >= 0) slot_load(ct.src, cur);
        pose_step(o, ct, q, ct.src == DRM_SRC_ROOT, cur);
        if (ct.save >= 0) slot_save(ct.save, cur);
        if (ct.out >= 0) {
            Pose P;
            pose_from_pairs(cur, P);
            Rot9 R;
#pragma unroll
            for (int i = 0; i < 9; ++i) R.v[i] = P.R[i];
            const Quat4 qt = target_quaternion(R, ct.perm);
            emit(ct.out, P.p, qt.v);
        }
    }
}

// (A hand-pipelined form of this loop — control words two ops ahead, constants and joint value one op ahead — and a
// straight-line form for chains of up to 8 ops with paired sin / cos both measured SLOWER than the plain loop on the
// Allegro's four fingertips at 65 536 samples: 7.3 / 6.8 against 6.5 us.)

// ---------------------------------------------------------------------------
// FK + geometric Jacobian along one chain (robot_model.py:626-667): op 0 hangs off the root, op k off op k - 1.
//   column(d, z[3], p[3], prismatic)  called once per moving op: world joint axis and origin of DoF d; the caller forms
//                                     (lin, ang) = (z x (p_e - p), z) for a revolute and (z, 0) for a prismatic joint
// Returns the canonical pose of the last real op in `ee`.
// ---------------------------------------------------------------------------
template <class CTL, class ROW, class QF, class COL>
DRM_HD void fk_jacobian_tree_walk(int n_ops, CTL ctl, ROW row, QF qf, PoseP &ee, COL column) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { ee.A[c] = f2_make(c == 0, c == 1); ee.B[c] = f2_make(c == 2, 0.0f); }
#pragma unroll 1
    for (int k = 0; k < n_ops; ++k) {
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        const OpPairs o = load_pairs(row(k));
        const float q = ct.dof >= 0 ? qf(ct.dof) : 0.0f;
        pose_step(o, ct, q, k == 0, ee);
        if (ct.dof >= 0) {
            const float z[3] = {ee.B[0][0], ee.B[1][0], ee.B[2][0]};
            // a prismatic joint has moved the origin along z already; the column of a prismatic joint is (z, 0)
            const float p[3] = {ee.B[0][1], ee.B[1][1], ee.B[2][1]};
            column(ct.dof, z, p, ct.prismatic);
        }
    }
}

// ---------------------------------------------------------------------------
// RNEA over one segment of the tree (robot_model.py:250-375): the static prefix ops [0, p_end) are replayed for their
// motions, then ops [a, b) (a run of whole sub-trees): forward sweep (motions, body forces), backward sweep (forces
// towards the root, torques).  Body-frame Pluecker coordinates at the link origin, as in the reference.
//   qf(d, q, qd, qdd)  joint state of DoF d            tau_out(d, v)  torque of DoF d
//   park(k, F, c, s, q) / unpark(k, F, c, s, q)   per-op record between the sweeps
//   motion_save / motion_load(s, Motion)           branch-point motions
//   force_add(s, Force) / force_take(s, Force&)    branch-point force accumulators (take = add into F and reset)
// ---------------------------------------------------------------------------
// one op of the forward sweep: its motion from the parent's (cur is updated), its body force and what the backward sweep
// needs again (cos / sin / value of the joint)
template <class ROW, class QF, class MSAVE, class MLOAD>
DRM_HD void rnea_forward_step(int k, const OpCtl &ct, ROW row, float g, QF qf, MSAVE motion_save, MLOAD motion_load, Motion &cur,
                              bool want_force, Force &f, float &c, float &s, float &q) {
    const float *of = row(k);
    float wj = 0.0f, aj = 0.0f;
    q = 0.0f; c = 1.0f; s = 0.0f;
    if (ct.dof >= 0) {
        qf(ct.dof, q, wj, aj);
        if (!ct.prismatic) sincos_one(q, s, c);
    }
    const OpFT o = load_ft(of);
    float J[9], t[3];
    joint_transform(o, ct.dof >= 0, ct.prismatic, q, c, s, J, t);
    if (ct.src == DRM_SRC_ROOT) motion_root(cur, g);
    if (ct.src >= 0) motion_load(ct.src, cur);
    motion_step(J, t, wj, aj, ct.prismatic, cur, cur);
    if (ct.save >= 0) motion_save(ct.save, cur);
    if (want_force) rnea_body_force(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, cur, f);
}
// one op of the backward sweep: tot = this op's body force on entry, the total force of its sub-tree when the torque is
// taken; then moved into the parent's frame (carry / slot)
template <class ROW, class QF, class TAU, class FADD, class FTAKE>
DRM_HD void rnea_backward_step(int k, int a, const OpCtl &ct, ROW row, int flags, QF qf, TAU tau_out, FADD force_add,
                               FTAKE force_take, Force &tot, float c, float s, float q, Force &carry) {
    const float *of = row(k);
    if (ct.child_next) {
#pragma unroll
        for (int i = 0; i < 3; ++i) tot.la[i] += carry.la[i];
    }
    if (ct.save >= 0) force_take(ct.save, tot);
    if (ct.dof >= 0) {
        // tau = S^T f: the angular z component for a revolute joint (robot_model.py:353-373), the linear one for a
        // prismatic joint; + damping * qd
        float tau = ct.prismatic ? tot.la[2][0] : tot.la[2][1];
        if (flags & DRM_RNEA_DAMPING) {
            float qq, qd, qdd;
            qf(ct.dof, qq, qd, qdd);
            tau += of[DRM_OPF_DAMP] * qd;
        }
        tau_out(ct.dof, tau);
    }
    if (ct.src != DRM_SRC_ROOT && ct.parent >= a) { // (a parent in the static prefix takes no force)
        const OpFT o = load_ft(of);
        float J[9], t[3];
        joint_transform(o, ct.dof >= 0, ct.prismatic, q, c, s, J, t);
        Force up;
        rnea_link_force_up(J, t, tot, up);
        if (ct.src >= 0) force_add(ct.src, up);
        else carry = up;
    }
}

template <class CTL, class ROW, class QF, class TAU, class PARK, class UNPARK, class MSAVE, class MLOAD, class FADD, class FTAKE>
DRM_HD void rnea_tree_walk(int p_end, int a, int b, CTL ctl, ROW row, int flags, QF qf, TAU tau_out, PARK park,
                           UNPARK unpark, MSAVE motion_save, MLOAD motion_load, FADD force_add, FTAKE force_take) {
    const float g = (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f;
    Motion cur;
    motion_root(cur, g);
    // ---- forward sweep ------------------------------------------------------------------------------------
#pragma unroll 1
    for (int k = (p_end > 0 ? 0 : a); k < b; k = (k + 1 == p_end ? a : k + 1)) {
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        Force f;
        float c, s, q;
        rnea_forward_step(k, ct, row, g, qf, motion_save, motion_load, cur, k >= a, f, c, s, q);
        if (k >= a) park(k, f, c, s, q);
    }
    // ---- backward sweep -------------------------------------------------------------------------------------
    Force carry;
#pragma unroll
    for (int i = 0; i < 3; ++i) carry.la[i] = f2_bcast(0.0f);
#pragma unroll 1
    for (int k = b - 1; k >= a; --k) {
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        Force tot;
        float c, s, q;
        unpark(k, tot, c, s, q);
        rnea_backward_step(k, a, ct, row, flags, qf, tau_out, force_add, force_take, tot, c, s, q, carry);
    }
}

// The same walk for a SHORT segment (at most MAXOPS ops after the static prefix — a finger of a hand): both sweeps are
// unrolled, so the per-op records (body force, cos / sin / value) stay in REGISTERS instead of being parked in LDS.  With
// the records gone a 64-sample tile of a hand needs 16 KB of LDS instead of 66 KB and the launch is bounded by registers
// (four wavefronts per SIMD), not by LDS (two).
template <int MAXOPS, class CTL, class ROW, class QF, class TAU, class MSAVE, class MLOAD, class FADD, class FTAKE>
DRM_HD void rnea_tree_walk_short(int p_end, int a, int b, CTL ctl, ROW row, int flags, QF qf, TAU tau_out, MSAVE motion_save,
                                 MLOAD motion_load, FADD force_add, FTAKE force_take) {
    const float g = (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f;
    Motion cur;
    motion_root(cur, g);
#pragma unroll 1
    for (int k = 0; k < p_end; ++k) { // the static prefix: motions only
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        Force f;
        float c, s, q;
        rnea_forward_step(k, decode_ctl(w0, w1), row, g, qf, motion_save, motion_load, cur, false, f, c, s, q);
    }
    Force f[MAXOPS];
    float cc[MAXOPS], ss[MAXOPS], qq[MAXOPS];
    int w0s[MAXOPS], w1s[MAXOPS];
    const int len = b - a;
#pragma unroll
    for (int i = 0; i < MAXOPS; ++i) {
        if (i < len) {
            ctl_words(ctl, a + i, w0s[i], w1s[i]);
            rnea_forward_step(a + i, decode_ctl(w0s[i], w1s[i]), row, g, qf, motion_save, motion_load, cur, true, f[i], cc[i], ss[i], qq[i]);
        }
    }
    Force carry;
#pragma unroll
    for (int i = 0; i < 3; ++i) carry.la[i] = f2_bcast(0.0f);
#pragma unroll
    for (int i = MAXOPS - 1; i >= 0; --i) {
        if (i < len)
            rnea_backward_step(a + i, a, decode_ctl(w0s[i], w1s[i]), row, flags, qf, tau_out, force_add, force_take, f[i], cc[i], ss[i],
                               qq[i], carry);
    }
}

// ---------------------------------------------------------------------------
// Joint-space inertia matrix H(q) of one segment by the composite-rigid-body algorithm (what the reference builds from
// n + 1 inverse-dynamics passes, robot_model.py:402-450; see the note at `struct Inertia` in drm_sample.hpp).
// One sweep from the leaves to the root: the composite inertia of the sub-tree below op k is complete when the sweep
// reaches k; if k moves, F = Ic_k S_k is walked up the chain of k's ancestors (parent indices from W1) and leaves
// H[j][k] = S_j . F at every moving ancestor j.  S = (ang e_z, lin 0) for a revolute joint, (0, e_z) for a prismatic one.
//   trig(k, c, s, q)             cos / sin / value of op k's joint (the caller computes them once per op, see crba_prepare)
//   islot_add / islot_take       branch-point composite inertias (take = add into the argument and reset)
//   hout(di, dj, v)              H[di][dj] = v (called for both triangles)
// ---------------------------------------------------------------------------
template <class CTL, class ROW, class TRIG, class IADD, class ITAKE, class HOUT>
DRM_HD void crba_tree_walk(int a, int b, CTL ctl, ROW row, TRIG trig, IADD islot_add, ITAKE islot_take, HOUT hout) {
    Inertia carry;
    inertia_zero(carry);
#pragma unroll 1
    for (int k = b - 1; k >= a; --k) {
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        const float *of = row(k);
        Inertia tot;
        tot.m = of[DRM_OPF_MASS];
#pragma unroll
        for (int i = 0; i < 3; ++i) tot.h[i] = of[DRM_OPF_MCOM + i];
        tot.I[0] = of[DRM_OPF_IO + 0]; tot.I[1] = of[DRM_OPF_IO + 1]; tot.I[2] = of[DRM_OPF_IO + 2];
        tot.I[3] = of[DRM_OPF_IO + 4]; tot.I[4] = of[DRM_OPF_IO + 5]; tot.I[5] = of[DRM_OPF_IO + 8];
        if (ct.child_next) inertia_add(tot, carry);
        if (ct.save >= 0) islot_take(ct.save, tot);
        const OpFT o = load_ft(of);
        float J[9], t[3], c, s, q;
        trig(k, c, s, q);
        joint_transform(o, ct.dof >= 0, ct.prismatic, q, c, s, J, t);
        if (ct.dof >= 0) {
            // F = Ic S_k.  revolute: f = -h x e_z = (-h_y, h_x, 0), n = I e_z;  prismatic: f = m e_z, n = h x e_z = (h_y, -h_x, 0)
            Force F;
            if (!ct.prismatic) {
                F.la[0] = f2_make(-tot.h[1], tot.I[2]);
                F.la[1] = f2_make(tot.h[0], tot.I[4]);
                F.la[2] = f2_make(0.0f, tot.I[5]);
                hout(ct.dof, ct.dof, tot.I[5]);
            } else {
                F.la[0] = f2_make(0.0f, tot.h[1]);
                F.la[1] = f2_make(0.0f, -tot.h[0]);
                F.la[2] = f2_make(tot.m, 0.0f);
                hout(ct.dof, ct.dof, tot.m);
            }
            // up the ancestors: this op's own transform first, then its parent's, ...
            Force up;
            rnea_link_force_up(J, t, F, up);
            int anc = ct.parent;
#pragma unroll 1
            while (anc >= a) {
                int v0, v1;
                ctl_words(ctl, anc, v0, v1);
                const OpCtl ca = decode_ctl(v0, v1);
                if (ca.dof >= 0) {
                    const float v = ca.prismatic ? up.la[2][0] : up.la[2][1];
                    hout(ca.dof, ct.dof, v);
                    hout(ct.dof, ca.dof, v);
                }
                if (ca.parent < a) break;
                const OpFT oa = load_ft(row(anc));
                float Ja[9], ta[3], ca_c, ca_s, ca_q;
                trig(anc, ca_c, ca_s, ca_q);
                joint_transform(oa, ca.dof >= 0, ca.prismatic, ca_q, ca_c, ca_s, Ja, ta);
                Force nxt;
                rnea_link_force_up(Ja, ta, up, nxt);
                up = nxt;
                anc = ca.parent;
            }
        }
        if (ct.src != DRM_SRC_ROOT && ct.parent >= a) {
            Inertia upI;
            inertia_to_parent(J, t, tot, upI);
            if (ct.src >= 0) islot_add(ct.src, upI);
            else carry = upI;
        }
    }
}

// The same for a SHORT SERIAL segment (ops a .. a + len - 1, each the child of the one before it, at most MAXOPS of them — a
// finger of a hand, a leg): both loops unrolled, every op's joint transform built ONCE and kept in registers, nothing
// parked.  The generic walk above rebuilds an ancestor's transform (12 LDS reads, 12 multiply-adds, a control-word decode)
// for every (joint, ancestor) pair and parks cos / sin per op; here a 5-op finger costs about a third of its instructions.
// Returns false (nothing done) when the segment is not such a chain — the caller then runs crba_tree_walk.
//   qf(d) -> joint value of DoF d;   hout(di, dj, v) as above
template <int MAXOPS, class CTL, class ROW, class QF, class HOUT>
DRM_HD bool crba_tree_walk_short(int a, int b, CTL ctl, ROW row, QF qf, HOUT hout) {
    const int len = b - a;
    if (len < 1 || len > MAXOPS) return false;
    int dofs[MAXOPS];
    bool pris[MAXOPS];
    bool serial = true;
#pragma unroll
    for (int i = 0; i < MAXOPS; ++i) {
        dofs[i] = -1;
        pris[i] = false;
        if (i < len) {
            int w0, w1;
            ctl_words(ctl, a + i, w0, w1);
            const OpCtl ct = decode_ctl(w0, w1);
            dofs[i] = ct.dof;
            pris[i] = ct.prismatic;
            serial = serial && ct.save < 0 && (i == 0 || ct.parent == a + i - 1);
        }
    }
    if (!serial) return false;
    float J[MAXOPS][9], t[MAXOPS][3];
#pragma unroll
    for (int i = 0; i < MAXOPS; ++i) {
        if (i < len) {
            const OpFT o = load_ft(row(a + i));
            float q = 0.0f, c = 1.0f, s = 0.0f;
            if (dofs[i] >= 0) {
                q = qf(dofs[i]);
                if (!pris[i]) sincos_one(q, s, c);
            }
            joint_transform(o, dofs[i] >= 0, pris[i], q, c, s, J[i], t[i]);
        }
    }
    Inertia carry;
    inertia_zero(carry);
#pragma unroll
    for (int i = MAXOPS - 1; i >= 0; --i) {
        if (i < len) {
            const float *of = row(a + i);
            Inertia tot;
            tot.m = of[DRM_OPF_MASS];
#pragma unroll
            for (int x = 0; x < 3; ++x) tot.h[x] = of[DRM_OPF_MCOM + x];
            tot.I[0] = of[DRM_OPF_IO + 0]; tot.I[1] = of[DRM_OPF_IO + 1]; tot.I[2] = of[DRM_OPF_IO + 2];
            tot.I[3] = of[DRM_OPF_IO + 4]; tot.I[4] = of[DRM_OPF_IO + 5]; tot.I[5] = of[DRM_OPF_IO + 8];
            if (i < len - 1) inertia_add(tot, carry);
            if (dofs[i] >= 0) {
                Force F;
                if (!pris[i]) {
                    F.la[0] = f2_make(-tot.h[1], tot.I[2]);
                    F.la[1] = f2_make(tot.h[0], tot.I[4]);
                    F.la[2] = f2_make(0.0f, tot.I[5]);
                    hout(dofs[i], dofs[i], tot.I[5]);
                } else {
                    F.la[0] = f2_make(0.0f, tot.h[1]);
                    F.la[1] = f2_make(0.0f, -tot.h[0]);
                    F.la[2] = f2_make(tot.m, 0.0f);
                    hout(dofs[i], dofs[i], tot.m);
                }
#pragma unroll
                for (int j = i - 1; j >= 0; --j) { // into the frame of op a + j (through op a + j + 1's transform)
                    Force up;
                    rnea_link_force_up(J[j + 1], t[j + 1], F, up);
                    F = up;
                    if (dofs[j] >= 0) {
                        const float v = pris[j] ? F.la[2][0] : F.la[2][1];
                        hout(dofs[j], dofs[i], v);
                        hout(dofs[i], dofs[j], v);
                    }
                }
            }
            if (i > 0) inertia_to_parent(J[i], t[i], tot, carry);
        }
    }
    return true;
}

// cos / sin / value of the joint of every op of [a, b), handed to `put(k, c, s, q)` (parked by the caller for crba_tree_walk)
template <class CTL, class QF, class PUT>
DRM_HD void crba_prepare(int a, int b, CTL ctl, QF qf, PUT put) {
#pragma unroll 1
    for (int k = a; k < b; ++k) {
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        float q = 0.0f, c = 1.0f, s = 0.0f;
        if (ct.dof >= 0) {
            q = qf(ct.dof);
            if (!ct.prismatic) sincos_one(q, s, c);
        }
        put(k, c, s, q);
    }
}

} // namespace drm
