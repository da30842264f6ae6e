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

// The same walk over the ops [0, P) followed by [first, last) (P <= first): a wavefront of the fanned-out many-target kernel walks
// the part every sub-tree hangs off and then its own sub-trees.  emit(k, t, p, q) also gets the op.
template <class CTL, class ROW, class QF, class SAVE, class LOAD, class EMIT>
DRM_HD void fk_tree_walk_ranges(int P, int first, int last, CTL ctl, ROW row, QF qf, SAVE slot_save, LOAD slot_load, EMIT emit) {
    PoseP cur;
#pragma unroll
    for (int c = 0; c < 3; ++c) { cur.A[c] = f2_make(c == 0, c == 1); cur.B[c] = f2_make(c == 2, 0.0f); }
    const int total = P + (last - first);
#pragma unroll 1
    for (int i = 0; i < total; ++i) {
        const int k = i < P ? i : first + (i - P);
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        const OpPairs o = load_pairs(row(k));
        const float q = ct.dof >= 0 ? qf(ct.dof) : 0.0f;
        if (ct.src >= 0) slot_load(ct.src, cur);
        pose_step(o, ct, q, ct.src == DRM_SRC_ROOT, cur);
        if (ct.save >= 0) slot_save(ct.save, cur);
        if (ct.out >= 0) {
            Pose Q;
            pose_from_pairs(cur, Q);
            Rot9 R;
#pragma unroll
            for (int j = 0; j < 9; ++j) R.v[j] = Q.R[j];
            const Quat4 qt = target_quaternion(R, ct.perm);
            emit(k, ct.out, Q.p, qt.v);
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

// (unpark(k, F) fetches the body force parked for op k — the records may live in HBM, so the backward sweep asks for op k - 1's
// before it works on op k; trig(k, c, s, q) gives cos / sin / value of op k's joint again)
template <class CTL, class ROW, class QF, class TAU, class PARK, class UNPARK, class TRIG, class MSAVE, class MLOAD, class FADD, class FTAKE>
DRM_HD void rnea_tree_walk(int p_end, int a, int b, CTL ctl, ROW row, int flags, QF qf, TAU tau_out, PARK park,
                           UNPARK unpark, TRIG trig, MSAVE motion_save, MLOAD motion_load, FADD force_add, FTAKE force_take) {
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
    Force ahead;
    unpark(b - 1, ahead);
#pragma unroll 1
    for (int k = b - 1; k >= a; --k) {
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        Force tot = ahead;
        if (k > a) unpark(k - 1, ahead);
        float c, s, q;
        trig(k, c, s, q);
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
//   trig(k, c, s, q)             cos / sin / value of op k's joint
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

// The same matrix for a LONG segment (an arm carrying a gripper or a hand), organised so that nothing is walked twice.  The
// walk above climbs from every moving op to the root of its segment — for each (joint, ancestor) pair a control-word decode,
// the ancestor's constants and cos / sin out of LDS, its transform rebuilt, one force moved up: ~150 pairs on a 7-DoF arm
// with a 16-DoF hand, each a serial chain of LDS round trips that a lone wavefront cannot hide (measured: 30 cycles per
// instruction).  Here the forces F_c = Ic_c S_c of ALL the joints below an op travel up together: when the sweep reaches op
// i they are already expressed in i's frame, H[i][c] = S_i . F_c is read off for every c in the sub-tree, and all of them
// are moved into the parent's frame by i's transform, which is built ONCE.  Same multiply-adds in the same order per entry
// (bit-identical results), but the inner loop is over independent vectors.
//   sub-tree bookkeeping (wave-uniform, computed once per walk by crba_set_tables): the ops of a sub-tree are contiguous in the
//   walk, so are the slots of its moving ops —  slot_lo(k) = moving ops of the segment before op k;  slot_hi(k) = slot_lo of the
//   first op after k's sub-tree;  slot_dof(m) = DoF column of the m-th moving op
//   fget(m, Force&) / fput(m, Force)   the force of slot m (LDS in the kernel)
//   trig(k, c, s, q) as above, called once per op;  islot_add / islot_take / hout as above
template <class CTL, class ROW, class TRIG, class SLO, class SHI, class SDOF, class FGET, class FPUT, class IADD, class ITAKE, class HOUT>
DRM_HD void crba_set_walk(int a, int b, CTL ctl, ROW row, TRIG trig, SLO slot_lo, SHI slot_hi, SDOF slot_dof, FGET fget, FPUT fput,
                          IADD islot_add, ITAKE islot_take, HOUT hout) {
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
        const int m0 = slot_lo(k), m1 = slot_hi(k);
        const bool moving = ct.dof >= 0, up = ct.src != DRM_SRC_ROOT && ct.parent >= a;
        if (moving) {
            // F = Ic S_k.  revolute: f = -h x e_z = (-h_y, h_x, 0), n = I e_z;  prismatic: f = m e_z, n = h x e_z = (h_y, -h_x, 0)
            Force F;
            if (!ct.prismatic) {
                F.la[0] = f2_make(-tot.h[1], tot.I[2]);
                F.la[1] = f2_make(tot.h[0], tot.I[4]);
                F.la[2] = f2_make(0.0f, tot.I[5]);
            } else {
                F.la[0] = f2_make(0.0f, tot.h[1]);
                F.la[1] = f2_make(0.0f, -tot.h[0]);
                F.la[2] = f2_make(tot.m, 0.0f);
            }
            fput(m0, F);
        }
        // (three loops instead of one with the op's flags tested per vector)
        auto entry = [&](int m, const Force &F) { // H[k][c] = S_k . F_c: the z component, angular (revolute) or linear (prismatic)
            const float v = ct.prismatic ? F.la[2][0] : F.la[2][1];
            const int dj = slot_dof(m);
            hout(ct.dof, dj, v);
            if (dj != ct.dof) hout(dj, ct.dof, v);
        };
        if (moving && up) {
            for (int m = m0; m < m1; ++m) {
                Force F, nxt;
                fget(m, F);
                entry(m, F);
                rnea_link_force_up(J, t, F, nxt);
                fput(m, nxt);
            }
        } else if (moving) {
            for (int m = m0; m < m1; ++m) {
                Force F;
                fget(m, F);
                entry(m, F);
            }
        } else if (up) {
            for (int m = m0; m < m1; ++m) {
                Force F, nxt;
                fget(m, F);
                rnea_link_force_up(J, t, F, nxt);
                fput(m, nxt);
            }
        }
        if (up) {
            Inertia upI;
            inertia_to_parent(J, t, tot, upI);
            if (ct.src >= 0) islot_add(ct.src, upI);
            else carry = upI;
        }
    }
}
// the wave-uniform tables of crba_set_walk for ops [a, b): put_lo(k, v) for k in [a, b], put_hi(k, v) for k in [a, b),
// put_dof(m, dof); get_hi(k) reads back what put_hi stored (the caller's arrays: LDS in the kernel)
template <class CTL, class PLO, class PHI, class GHI, class PDOF>
DRM_HD void crba_set_tables(int a, int b, CTL ctl, PLO put_lo, PHI put_hi, GHI get_hi, PDOF put_dof) {
    int m = 0;
    for (int k = a; k < b; ++k) { // slots in walk order
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        put_lo(k, m);
        if (ct.dof >= 0) put_dof(m++, ct.dof);
    }
    put_lo(b, m);
    // last op of every sub-tree: a parent's sub-tree ends where its last child's does (children come after their parent)
    for (int k = a; k < b; ++k) put_hi(k, k);
    for (int k = b - 1; k >= a; --k) {
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        if (ct.parent >= a && get_hi(k) > get_hi(ct.parent)) put_hi(ct.parent, get_hi(k));
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

// ---------------------------------------------------------------------------
// Forward dynamics of one segment by the articulated-body algorithm, the reference's own algorithm
// (robot_model.py:487-624): O(ops) instead of the O(n^3) factorisation of H, nothing per sample but 8 floats per op
// between the sweeps — what robots with one long segment (an arm carrying a gripper or a hand) need, where the packed
// triangle of H in LDS left one wavefront per CU.
//   sweep 1, root -> leaves  (robot_model.py:539-549)  link velocities; parked (6 floats per op)
//   sweep 2, leaves -> root  (robot_model.py:551-604)  articulated inertia IA and bias force pA of every sub-tree; at a
//            moving joint  U = IA S,  D = S . U,  u = f - damping qd - S . pA  (parked: U, 1 / D, u),
//            Ia = IA - U U^T / D,  pa = pA + Ia c + U u / D,  both moved into the parent's frame
//   sweep 3, root -> leaves  (robot_model.py:611-629)  a' = X a_parent + c,  qdd = (u - U . a') / D,  a = a' + S qdd
// c = v x (S qd) is the velocity-product acceleration; sweep 3 recomputes the velocities next to the accelerations (the
// packed pairs of `Motion` carry both through the same instructions).  Fixed joints have S = 0: nothing is eliminated.
// Body-frame Pluecker coordinates at the link origin as everywhere else; S = (e_z, 0) revolute, (0, e_z) prismatic.
// ---------------------------------------------------------------------------
// A symmetric 6 x 6 inertia acting on a motion (w, v):  n = A w + B v,  f = B^T w + M v
struct ArtInertia {
    f2 MA[6];   // (M_ij, A_ij) for ij = 00 01 02 11 12 22: the two symmetric blocks go through the same rotations as pairs
    float B[9]; // row major
};
DRM_HD constexpr int sym_index(int i, int j) { return i <= j ? (i == 0 ? j : i + j + 1) : (j == 0 ? i : i + j + 1); }
static_assert(sym_index(0, 0) == 0 && sym_index(0, 2) == 2 && sym_index(1, 1) == 3 && sym_index(2, 1) == 4 && sym_index(2, 2) == 5, "packing");

struct ArtBody { // what travels towards the root: the articulated inertia and bias force of a sub-tree
    ArtInertia I;
    Force p; // (lin, ang) pairs
};
DRM_HD void art_zero(ArtBody &a) {
#pragma unroll
    for (int i = 0; i < 6; ++i) a.I.MA[i] = f2_bcast(0.0f);
#pragma unroll
    for (int i = 0; i < 9; ++i) a.I.B[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) a.p.la[i] = f2_bcast(0.0f);
}
DRM_HD void art_add(ArtBody &a, const ArtBody &b) {
#pragma unroll
    for (int i = 0; i < 6; ++i) a.I.MA[i] += b.I.MA[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) a.I.B[i] += b.I.B[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) a.p.la[i] += b.p.la[i];
}
// rigid-body inertia of a link row: A = I_o, M = m 1, B = skew(m c)   (n = I_o w + mc x v,  f = m v - mc x w)
DRM_HD void art_from_link(const float *of, ArtInertia &I) {
    const float m = of[DRM_OPF_MASS], *h = of + DRM_OPF_MCOM, *Io = of + DRM_OPF_IO;
    I.MA[0] = f2_make(m, Io[0]); I.MA[1] = f2_make(0.0f, Io[1]); I.MA[2] = f2_make(0.0f, Io[2]);
    I.MA[3] = f2_make(m, Io[4]); I.MA[4] = f2_make(0.0f, Io[5]); I.MA[5] = f2_make(m, Io[8]);
    I.B[0] = 0.0f;  I.B[1] = -h[2]; I.B[2] = h[1];
    I.B[3] = h[2];  I.B[4] = 0.0f;  I.B[5] = -h[0];
    I.B[6] = -h[1]; I.B[7] = h[0];  I.B[8] = 0.0f;
}
// X^T I X for the transform of a child into its parent's frame (x_p = J x_c + t; parent motion -> child motion is
// w_c = J^T w_p, v_c = J^T (v_p - t x w_p)).  With the blocks rotated first (A' = J A J^T, ...) and T = skew(t):
//   M_p = M' ;  B_p = B' + T M' ;  A_p = A' - B' T + T B_p^T       (robot_model.py:590-596 forms the 6 x 6 products)
DRM_HD void art_inertia_to_parent(const float *J, const float *t, const ArtInertia &c, ArtInertia &out) {
    f2 P[9], R[6];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            P[r * 3 + k] = f2_bcast(J[r * 3 + 0]) * c.MA[sym_index(0, k)] + f2_bcast(J[r * 3 + 1]) * c.MA[sym_index(1, k)] +
                           f2_bcast(J[r * 3 + 2]) * c.MA[sym_index(2, k)];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j)
            R[sym_index(i, j)] = P[i * 3 + 0] * f2_bcast(J[j * 3 + 0]) + P[i * 3 + 1] * f2_bcast(J[j * 3 + 1]) + P[i * 3 + 2] * f2_bcast(J[j * 3 + 2]);
    float Q[9], Br[9], Bp[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) Q[r * 3 + k] = J[r * 3 + 0] * c.B[0 * 3 + k] + J[r * 3 + 1] * c.B[1 * 3 + k] + J[r * 3 + 2] * c.B[2 * 3 + k];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Br[i * 3 + j] = Q[i * 3 + 0] * J[j * 3 + 0] + Q[i * 3 + 1] * J[j * 3 + 1] + Q[i * 3 + 2] * J[j * 3 + 2];
    // B_p = B' + T M': column j of T M' is t x (column j of M')
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float m0 = R[sym_index(0, j)][0], m1 = R[sym_index(1, j)][0], m2 = R[sym_index(2, j)][0];
        Bp[0 * 3 + j] = Br[0 * 3 + j] + (t[1] * m2 - t[2] * m1);
        Bp[1 * 3 + j] = Br[1 * 3 + j] + (t[2] * m0 - t[0] * m2);
        Bp[2 * 3 + j] = Br[2 * 3 + j] + (t[0] * m1 - t[1] * m0);
    }
    // A_p[i][j] = A'[i][j] - (row i of B' x t)[j] + (t x row j of B_p)[i]
    auto cross_c = [](const float *a, const float *b, int c) { // component c of a x b
        const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
        return a[c1] * b[c2] - a[c2] * b[c1];
    };
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            const float a = R[sym_index(i, j)][1] - cross_c(Br + 3 * i, t, j) + cross_c(t, Bp + 3 * j, i);
            out.MA[sym_index(i, j)] = f2_make(R[sym_index(i, j)][0], a);
        }
#pragma unroll
    for (int i = 0; i < 9; ++i) out.B[i] = Bp[i];
}

// one op of sweep 2.  `tot` = this op's own inertia + bias force + what its children handed up; on return `up` is the
// op's contribution to its parent (in the parent's frame).  U (lin, ang pairs), 1 / D and u go to rec[8].
DRM_HD void aba_eliminate(bool moving, bool prismatic, const float *J, const float *t, float qd, float f_joint, const Motion &vel,
                          ArtBody &tot, float *rec, bool want_up, ArtBody &up) {
    if (moving) {
        // U = IA S as (lin, ang) pairs: revolute = column "angular z" of IA, prismatic = column "linear z"
        f2 U[3];
        float D, sp;
        if (!prismatic) {
#pragma unroll
            for (int i = 0; i < 3; ++i) U[i] = f2_make(tot.I.B[2 * 3 + i], tot.I.MA[sym_index(i, 2)][1]);
            D = tot.I.MA[5][1];
            sp = tot.p.la[2][1];
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) U[i] = f2_make(tot.I.MA[sym_index(i, 2)][0], tot.I.B[i * 3 + 2]);
            D = tot.I.MA[5][0];
            sp = tot.p.la[2][0];
        }
        const float Dinv = 1.0f / D, u = f_joint - sp;
#pragma unroll
        for (int i = 0; i < 3; ++i) { rec[i] = U[i][0]; rec[3 + i] = U[i][1]; }
        rec[6] = Dinv; rec[7] = u;
        if (want_up) {
            // Ia = IA - U U^T / D
            f2 Us[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) Us[i] = U[i] * f2_bcast(Dinv);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = i; j < 3; ++j) tot.I.MA[sym_index(i, j)] -= Us[i] * U[j];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) tot.I.B[i * 3 + j] -= Us[i][1] * U[j][0];
            // c = v x (S qd) as (lin, ang) pairs: revolute (v x e_z, w x e_z) qd, prismatic (w x e_z, 0) qd;  x e_z = (y, -x, 0)
            const float w0 = vel.wa[0][0], w1 = vel.wa[1][0], v0 = vel.va[0][0], v1 = vel.va[1][0];
            f2 c0, c1;
            if (!prismatic) { c0 = f2_make(v1 * qd, w1 * qd); c1 = f2_make(-v0 * qd, -w0 * qd); }
            else { c0 = f2_make(w1 * qd, 0.0f); c1 = f2_make(-w0 * qd, 0.0f); }
            // pa = pA + Ia c + U u / D:   lin_i = M_ij cl_j + B_ji ca_j ,  ang_i = A_ij ca_j + B_ij cl_j   (j = 0, 1)
            const float ud = u * Dinv;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f2 acc = tot.I.MA[sym_index(i, 0)] * c0 + tot.I.MA[sym_index(i, 1)] * c1;
                acc += f2_make(tot.I.B[0 * 3 + i] * c0[1] + tot.I.B[1 * 3 + i] * c1[1], tot.I.B[i * 3 + 0] * c0[0] + tot.I.B[i * 3 + 1] * c1[0]);
                tot.p.la[i] += acc + U[i] * f2_bcast(ud);
            }
        }
    }
    if (want_up) {
        art_inertia_to_parent(J, t, tot.I, up.I);
        rnea_link_force_up(J, t, tot.p, up.p);
    }
}

//   qf(d, q, qd)         joint state of DoF d           fj(d)  joint torque (the caller takes damping off; robot_model.py:515-521)
//   out(d, v)            joint acceleration of DoF d
//   vpark(k, Motion) / vunpark(k, Motion&)   the velocity halves of op k's motion, sweep 1 -> 2
//   rpark(k, rec[8]) / runpark(k, rec[8])    U, 1 / D, u of a moving op, sweep 2 -> 3
//   motion_save / motion_load                branch-point motions (sweeps 1 and 3)
//   body_add(s, ArtBody) / body_take(s, ArtBody&)   branch-point accumulators of sweep 2 (take = add into the argument and reset)
template <class CTL, class ROW, class QF, class FJ, class OUT, class VPARK, class VUNPARK, class RPARK, class RUNPARK, class MSAVE,
          class MLOAD, class BADD, class BTAKE>
DRM_HD void aba_tree_walk(int p_end, int a, int b, CTL ctl, ROW row, int flags, QF qf, FJ fj, OUT out, VPARK vpark, VUNPARK vunpark,
                          RPARK rpark, RUNPARK runpark, MSAVE motion_save, MLOAD motion_load, BADD body_add, BTAKE body_take) {
    const float g = (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f;
    // the joint transform of op k and its joint state
    auto joint = [&](int k, const OpCtl &ct, float *J, float *t, float &qd) {
        float q = 0.0f, c = 1.0f, s = 0.0f;
        qd = 0.0f;
        if (ct.dof >= 0) {
            qf(ct.dof, q, qd);
            if (!ct.prismatic) sincos_one(q, s, c);
        }
        const OpFT o = load_ft(row(k));
        joint_transform(o, ct.dof >= 0, ct.prismatic, q, c, s, J, t);
    };
    // ---- sweeps 1 and 3 share their loop: the motion of every op from its parent's --------------------------------
    // (the records may live in HBM: every sweep asks for the record of the NEXT op before it works on the current one)
    auto down = [&](bool third) {
        Motion cur;
        motion_root(cur, third ? g : 0.0f);
        float ahead[8];
        if (third) runpark(a, ahead);
#pragma unroll 1
        for (int k = (p_end > 0 ? 0 : a); k < b; k = (k + 1 == p_end ? a : k + 1)) {
            int w0, w1;
            ctl_words(ctl, k, w0, w1);
            const OpCtl ct = decode_ctl(w0, w1);
            float rec[8];
            if (third && k >= a) {
#pragma unroll
                for (int i = 0; i < 8; ++i) rec[i] = ahead[i];
                if (k + 1 < b) runpark(k + 1, ahead);
            }
            float J[9], t[3], qd;
            joint(k, ct, J, t, qd);
            if (ct.src == DRM_SRC_ROOT) motion_root(cur, third ? g : 0.0f);
            if (ct.src >= 0) motion_load(ct.src, cur);
            motion_step(J, t, qd, 0.0f, ct.prismatic, cur, cur); // acceleration halves: a' = X a_parent + c
            if (third && ct.dof >= 0 && k >= a) {
                float dot = 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i) dot += rec[i] * cur.va[i][1] + rec[3 + i] * cur.wa[i][1];
                const float qdd = (rec[7] - dot) * rec[6];
                out(ct.dof, qdd);
                if (ct.prismatic) cur.va[2][1] += qdd;
                else cur.wa[2][1] += qdd;
            }
            if (ct.save >= 0) motion_save(ct.save, cur);
            if (!third && k >= a) vpark(k, cur);
        }
    };
    down(false);
    // ---- sweep 2 -------------------------------------------------------------------------------------------------
    ArtBody carry;
    art_zero(carry);
    Motion vahead;
    vunpark(b - 1, vahead);
#pragma unroll 1
    for (int k = b - 1; k >= a; --k) {
        int w0, w1;
        ctl_words(ctl, k, w0, w1);
        const OpCtl ct = decode_ctl(w0, w1);
        const float *of = row(k);
        const Motion vel = vahead; // acceleration halves zero: the body force below is then the bias force v x* (I v)
        if (k > a) vunpark(k - 1, vahead);
        ArtBody tot;
        art_from_link(of, tot.I);
        rnea_body_force(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, vel, tot.p);
        if (ct.child_next) art_add(tot, carry);
        if (ct.save >= 0) body_take(ct.save, tot);
        float J[9], t[3], qd;
        joint(k, ct, J, t, qd);
        float fjoint = 0.0f;
        if (ct.dof >= 0) {
            fjoint = fj(ct.dof);
            if (flags & DRM_RNEA_DAMPING) fjoint -= of[DRM_OPF_DAMP] * qd;
        }
        const bool want_up = ct.src != DRM_SRC_ROOT && ct.parent >= a; // (a parent in the static prefix takes nothing)
        float rec[8];
        ArtBody up;
        aba_eliminate(ct.dof >= 0, ct.prismatic, J, t, qd, fjoint, vel, tot, rec, want_up, up);
        if (ct.dof >= 0) rpark(k, rec);
        if (want_up) {
            if (ct.src >= 0) body_add(ct.src, up);
            else carry = up;
        }
    }
    down(true);
}

// ---------------------------------------------------------------------------
// Forward dynamics of "an arm that carries a hand" (DRM_WALK_ARM_HAND: P serial prefix ops, K serial sub-chains of L ops
// hanging off the last prefix op) by the articulated-body recursion, straight-line for the shape (round 3).
// The loop form above keeps 8 floats per op between its sweeps in HBM scratch and decodes control words per op; here
//   * the prefix keeps its velocities (sweep 1 -> 2) and then U, 1/D, u (sweep 2 -> 3) in ONE 8-float LDS slot per op;
//   * a sub-chain never stores anything: its sweeps 1 + 2 run once to hand the palm its articulated inertia and bias force,
//     and once more — when the palm's acceleration is known — to rebuild U, 1/D, u in registers for its sweep 3.  The second
//     evaluation costs the sub-chains' sweep 2 again (~half of the work of a hand) and buys a kernel without scratch whose
//     LDS footprint is the prefix alone.
//   row(op), kind(op) (1 = moves, 2 = prismatic)   as in rnea_arm_hand
//   q / cs / sn / qd / fj of the prefix ops (fj = joint torque);  hq(j, i, q, qd, f) of a sub-chain op
//   out_p[k], hout(j, i, qdd)   joint accelerations (the caller drops those of fixed ops)
//   sput(k, s[8]) / sget(k, s[8])   the prefix's per-op LDS slot
// ---------------------------------------------------------------------------
DRM_HD Motion velocity_halves(const Motion &M) { // (w, 0), (v, 0): what the bias force v x* (I v) and c = v x (S qd) are built from
    Motion V;
#pragma unroll
    for (int i = 0; i < 3; ++i) { V.wa[i] = f2_make(M.wa[i][0], 0.0f); V.va[i] = f2_make(M.va[i][0], 0.0f); }
    return V;
}
// qdd of a moving op from its record and the acceleration a' = X a_parent + c that sweep 3 has just formed in `cur`
DRM_HD float aba_joint_acceleration(const float *rec, bool prismatic, Motion &cur) {
    float dot = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) dot += rec[i] * cur.va[i][1] + rec[3 + i] * cur.wa[i][1];
    const float qdd = (rec[7] - dot) * rec[6];
    if (prismatic) cur.va[2][1] += qdd;
    else cur.wa[2][1] += qdd;
    return qdd;
}
template <int P, int L, class ROW, class KIND, class HQ, class HOUT, class SPUT, class SGET>
DRM_HD void aba_arm_hand(ROW row, KIND kind, int K, bool gravity, bool damping, const float (&q)[P], const float (&cs)[P],
                         const float (&sn)[P], const float (&qd)[P], const float (&fj)[P], HQ hq, float (&out_p)[P], HOUT hout,
                         SPUT sput, SGET sget) {
    auto prefix_joint = [&](int k, float *J, float *t) {
        const int kd = kind(k);
        joint_transform(load_ft(row(k)), kd & 1, kd & 2, q[k], cs[k], sn[k], J, t);
    };
    // ---- sweep 1 of the prefix: velocities --------------------------------------------------------------------------
    Motion cur;
    motion_root(cur, 0.0f);
#pragma unroll
    for (int k = 0; k < P; ++k) {
        DRM_RNEA_LINK_FENCE();
        float J[9], t[3];
        prefix_joint(k, J, t);
        motion_step(J, t, qd[k], 0.0f, kind(k) & 2, cur, cur);
        const float s[8] = {cur.wa[0][0], cur.wa[1][0], cur.wa[2][0], cur.va[0][0], cur.va[1][0], cur.va[2][0], 0.0f, 0.0f};
        sput(k, s);
    }
    // sweeps 1 + 2 of sub-chain j from the palm's motion; recs != nullptr: keep U, 1/D, u of its ops; returns what it hands up
    auto sub_chain_up = [&](int j, const Motion &palm, const float (&hqv)[L], const float (&hqd)[L], const float (&hf)[L],
                            const float (&hc)[L], const float (&hs)[L], float (*recs)[8], ArtBody &handed) {
        Motion M = palm, v[L];
#pragma unroll
        for (int i = 0; i < L; ++i) {
            DRM_RNEA_LINK_FENCE();
            const int op = P + j * L + i, kd = kind(op);
            float J[9], t[3];
            joint_transform(load_ft(row(op)), kd & 1, kd & 2, hqv[i], hc[i], hs[i], J, t);
            motion_step(J, t, hqd[i], 0.0f, kd & 2, M, M);
            v[i] = velocity_halves(M);
        }
        ArtBody carry;
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            DRM_RNEA_LINK_FENCE();
            const int op = P + j * L + i, kd = kind(op);
            const float *of = row(op);
            ArtBody tot;
            art_from_link(of, tot.I);
            rnea_body_force(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, v[i], tot.p);
            if (i < L - 1) art_add(tot, carry);
            float J[9], t[3];
            joint_transform(load_ft(of), kd & 1, kd & 2, hqv[i], hc[i], hs[i], J, t);
            // f - damping qd as ONE rounding: on a finger both terms are ~3 N m and their difference ~0.02, and 1 / D of a fingertip
            // is 6e4 — a separately rounded product costs two digits of the distal accelerations (the loop form's `fjoint -= d qd`
            // is contracted into the same FMA by the compiler; a select in between would keep it from doing so here)
            const float fjoint = !(kd & 1) ? 0.0f : (damping ? __builtin_fmaf(-of[DRM_OPF_DAMP], hqd[i], hf[i]) : hf[i]);
            float rec[8];
            ArtBody up;
            const bool want_up = (recs == nullptr) || i > 0;
            aba_eliminate(kd & 1, kd & 2, J, t, hqd[i], fjoint, v[i], tot, rec, want_up, up);
            if (recs != nullptr) {
#pragma unroll
                for (int c = 0; c < 8; ++c) recs[i][c] = rec[c];
            }
            if (want_up) carry = up;
        }
        handed = carry;
    };
    // ---- first visit of the sub-chains: what they hand to the palm ---------------------------------------------------
    const Motion palm_v = velocity_halves(cur);
    ArtBody acc;
    art_zero(acc);
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
        float hqv[L], hqd[L], hf[L], hc[L], hs[L];
#pragma unroll
        for (int i = 0; i < L; ++i) hq(j, i, hqv[i], hqd[i], hf[i]);
        chain_trig<L>(hqv, hc, hs);
        ArtBody up;
        sub_chain_up(j, palm_v, hqv, hqd, hf, hc, hs, nullptr, up);
        art_add(acc, up);
    }
    // ---- sweep 2 of the prefix ----------------------------------------------------------------------------------------
    {
        ArtBody carry = acc;
#pragma unroll
        for (int k = P - 1; k >= 0; --k) {
            DRM_RNEA_LINK_FENCE();
            const float *of = row(k);
            const int kd = kind(k);
            float s[8];
            sget(k, s);
            Motion vel;
#pragma unroll
            for (int i = 0; i < 3; ++i) { vel.wa[i] = f2_make(s[i], 0.0f); vel.va[i] = f2_make(s[3 + i], 0.0f); }
            ArtBody tot;
            art_from_link(of, tot.I);
            rnea_body_force(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, vel, tot.p);
            art_add(tot, carry);
            float J[9], t[3];
            prefix_joint(k, J, t);
            const float fjoint = !(kd & 1) ? 0.0f : (damping ? __builtin_fmaf(-of[DRM_OPF_DAMP], qd[k], fj[k]) : fj[k]);
            float rec[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            ArtBody up;
            aba_eliminate(kd & 1, kd & 2, J, t, qd[k], fjoint, vel, tot, rec, k > 0, up);
            sput(k, rec);
            if (k > 0) carry = up;
        }
    }
    // ---- sweep 3 of the prefix ----------------------------------------------------------------------------------------
    motion_root(cur, gravity ? 9.81f : 0.0f);
#pragma unroll
    for (int k = 0; k < P; ++k) {
        DRM_RNEA_LINK_FENCE();
        const int kd = kind(k);
        float J[9], t[3];
        prefix_joint(k, J, t);
        motion_step(J, t, qd[k], 0.0f, kd & 2, cur, cur); // acceleration halves: a' = X a_parent + c
        out_p[k] = 0.0f;
        if (kd & 1) {
            float rec[8];
            sget(k, rec);
            out_p[k] = aba_joint_acceleration(rec, kd & 2, cur);
        }
    }
    // ---- second visit of the sub-chains: U, 1/D, u again, then their sweep 3 from the palm's acceleration --------------
    const Motion palm_v2 = velocity_halves(cur);
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
        float hqv[L], hqd[L], hf[L], hc[L], hs[L], recs[L][8];
#pragma unroll
        for (int i = 0; i < L; ++i) hq(j, i, hqv[i], hqd[i], hf[i]);
        chain_trig<L>(hqv, hc, hs);
        ArtBody unused;
        sub_chain_up(j, palm_v2, hqv, hqd, hf, hc, hs, recs, unused);
        Motion M = cur;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            DRM_RNEA_LINK_FENCE();
            const int op = P + j * L + i, kd = kind(op);
            float J[9], t[3];
            joint_transform(load_ft(row(op)), kd & 1, kd & 2, hqv[i], hc[i], hs[i], J, t);
            motion_step(J, t, hqd[i], 0.0f, kd & 2, M, M);
            if (kd & 1) hout(j, i, aba_joint_acceleration(recs[i], kd & 2, M));
        }
    }
}

// ---------------------------------------------------------------------------
// Joint-space inertia matrix of "an arm that carries a hand" (DRM_WALK_ARM_HAND) by the composite-rigid-body algorithm,
// straight-line for the shape (round 3).  Nothing is parked: a sub-chain keeps its L joint transforms in registers, walks
// the column force F = Ic S of each of its joints up to the palm as it appears, then carries its L forces together up the
// prefix (one transform per prefix op and force, the prefix's joint transforms rebuilt from cos / sin on the way) and hands
// the palm its composite inertia; the prefix then runs the chain form (crba_chain_trig) with the palm's composites added.
// Two pieces, so that a block can give every sub-chain its own wavefront (drm_arm_hand.hip) and share the prefix's columns:
//   crba_arm_hand_sub<P, L>(.., j, .., Fp, palm)            sub-chain j: its own entries, its L column forces in the palm's
//                                                           frame, and the composite it hands the palm
//   crba_arm_hand_prefix<P, L>(.., palm, Fp, first, step)   see there
//   row(op), kind(op) (1 = moves, 2 = prismatic)   wave-uniform
//   q / cs / sn of the prefix ops;  hq(i) -> joint value of op i of the sub-chain
//   hout(oa, ob, v)   the entry of the joints of ops oa <= ob (ob's column force seen by oa's axis), once per pair of moving
//                     ops on a common root path (and per diagonal entry); pairs on different sub-chains are structurally zero
// ---------------------------------------------------------------------------
DRM_HD void inertia_from_row(const float *of, Inertia &I) {
    I.m = of[DRM_OPF_MASS];
#pragma unroll
    for (int i = 0; i < 3; ++i) I.h[i] = of[DRM_OPF_MCOM + i];
    I.I[0] = of[DRM_OPF_IO + 0]; I.I[1] = of[DRM_OPF_IO + 1]; I.I[2] = of[DRM_OPF_IO + 2];
    I.I[3] = of[DRM_OPF_IO + 4]; I.I[4] = of[DRM_OPF_IO + 5]; I.I[5] = of[DRM_OPF_IO + 8];
}
// F = Ic S for a joint about / along +z; returns S . F (the diagonal entry)
DRM_HD float crba_column_force(const Inertia &tot, bool prismatic, Force &F) {
    if (!prismatic) { // f = -h x e_z = (-h_y, h_x, 0), n = I e_z
        F.la[0] = f2_make(-tot.h[1], tot.I[2]);
        F.la[1] = f2_make(tot.h[0], tot.I[4]);
        F.la[2] = f2_make(0.0f, tot.I[5]);
        return tot.I[5];
    }
    F.la[0] = f2_make(0.0f, tot.h[1]); // f = m e_z, n = h x e_z = (h_y, -h_x, 0)
    F.la[1] = f2_make(0.0f, -tot.h[0]);
    F.la[2] = f2_make(tot.m, 0.0f);
    return tot.m;
}
template <int P, int L, class ROW, class KIND, class HQ, class HOUT>
DRM_HD void crba_arm_hand_sub(ROW row, KIND kind, int j, HQ hq, HOUT hout, Force (&Fp)[L], Inertia &palm) {
    auto along = [](const Force &F, bool prismatic) { return prismatic ? F.la[2][0] : F.la[2][1]; }; // S . F
    float hqv[L], hc[L], hs[L], Jf[L][9], tf[L][3];
#pragma unroll
    for (int i = 0; i < L; ++i) hqv[i] = hq(i);
    chain_trig<L>(hqv, hc, hs);
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int kd = kind(P + j * L + i);
        joint_transform(load_ft(row(P + j * L + i)), kd & 1, kd & 2, hqv[i], hc[i], hs[i], Jf[i], tf[i]);
    }
    Inertia carry;
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {
        DRM_RNEA_LINK_FENCE();
        const int op = P + j * L + i, kd = kind(op);
        Inertia tot;
        inertia_from_row(row(op), tot);
        if (i < L - 1) inertia_add(tot, carry);
        if (kd & 1) {
            Force F;
            hout(op, op, crba_column_force(tot, kd & 2, F));
#pragma unroll
            for (int a = i - 1; a >= 0; --a) { // up the sub-chain: into the frame of its op a
                Force up;
                rnea_link_force_up(Jf[a + 1], tf[a + 1], F, up);
                F = up;
                const int ka = kind(P + j * L + a);
                if (ka & 1) hout(P + j * L + a, op, along(F, ka & 2));
            }
            rnea_link_force_up(Jf[0], tf[0], F, Fp[i]); // ... and into the palm's
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) Fp[i].la[c] = f2_bcast(0.0f);
        }
        if (i > 0) inertia_to_parent(Jf[i], tf[i], tot, carry);
        else inertia_to_parent(Jf[i], tf[i], tot, palm);
    }
}
// The prefix, ONE sweep from the palm to the root for everything a caller carries: the composite inertias (of all the ops),
// the L column forces Fp of sub-chain j, and the column forces of the prefix ops P-1-first, P-1-first-step, ... (step >= 2: at
// most (P + 1) / 2 of them live at once); every op's joint transform is built once per sweep and moves all of them.
template <int P, int L, class ROW, class KIND, class HOUT>
DRM_HD void crba_arm_hand_prefix(ROW row, KIND kind, int j, const float (&q)[P], const float (&cs)[P], const float (&sn)[P],
                                 const Inertia &palm, Force (&Fp)[L], int first, int step, HOUT hout) {
    constexpr int LIVE = (P + 1) / 2;
    auto along = [](const Force &F, bool prismatic) { return prismatic ? F.la[2][0] : F.la[2][1]; };
    Force Fl[LIVE];
    int col[LIVE];    // the op whose column slot i carries (wave-uniform; below 0: none)
    bool live[LIVE];
#pragma unroll
    for (int i = 0; i < LIVE; ++i) {
        col[i] = P - 1 - first - i * step;
        live[i] = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) Fl[i].la[c] = f2_bcast(0.0f);
    }
    Inertia carry = palm;
#pragma unroll
    for (int k = P - 1; k >= 0; --k) {
        DRM_RNEA_LINK_FENCE();
        const int kd = kind(k);
        Inertia tot;
        inertia_from_row(row(k), tot);
        inertia_add(tot, carry);
        if (kd & 1) {
#pragma unroll
            for (int i = 0; i < L; ++i)
                if (kind(P + j * L + i) & 1) hout(k, P + j * L + i, along(Fp[i], kd & 2));
#pragma unroll
            for (int i = 0; i < LIVE; ++i) {
                if (live[i]) hout(k, col[i], along(Fl[i], kd & 2));
                if (col[i] == k) {
                    hout(k, k, crba_column_force(tot, kd & 2, Fl[i]));
                    live[i] = true;
                }
            }
        }
        if (k > 0) {
            float J[9], t[3];
            joint_transform(load_ft(row(k)), kd & 1, kd & 2, q[k], cs[k], sn[k], J, t);
#pragma unroll
            for (int i = 0; i < L; ++i) {
                Force up;
                rnea_link_force_up(J, t, Fp[i], up);
                Fp[i] = up;
            }
#pragma unroll
            for (int i = 0; i < LIVE; ++i)
                if (live[i]) {
                    Force up;
                    rnea_link_force_up(J, t, Fl[i], up);
                    Fl[i] = up;
                }
            inertia_to_parent(J, t, tot, carry);
        }
    }
}

} // namespace drm
