// drm_sample.hpp — per-sample arithmetic of the FK / Jacobian / RNEA walks.
//
// One lane of a wavefront owns one sample and runs these functions with the
// walk's constants (ops_f / ops_i, include/drm_hip.h) as wave-uniform scalar
// operands.  Walks are identity-padded to a compile-time capacity CAP and run
// as STRAIGHT-LINE code: every loop over ops is fully unrolled, per-op state
// (joint axes, origins, body forces) lives in registers with static indices,
// and the only wave-uniform branches left are the rare ones (fixed joint,
// branch point, target link).  Every joint rotates about its local +z axis by
// +q — the host folds x / y axes and negative axes into exact signed
// permutations of the constants (flatten.py "axis canonicalisation").
//
// The arithmetic restates, per sample, what the reference spreads over
// rigid_body.py:130-165, spatial_vector_algebra.py:14-136,175-338 and
// robot_model.py:139-195,250-375,626-667; the operation
// order is re-associated where that saves work (results agree to fp32 rounding,
// tolerances in tests/).
//
// The header has no HIP dependency beyond the DRM_HD qualifier so that
// tests/host_emu can compile the very same arithmetic with g++ and check it
// against the oracle on a machine without a GPU (test infrastructure only —
// the product never runs it on the CPU).
#pragma once

#include <math.h>
#include <stdint.h>

#include "../../include/drm_hip.h"

#if defined(__HIPCC__)
#define DRM_HD __host__ __device__ __forceinline__
#else
#define DRM_HD inline __attribute__((always_inline))
#endif

namespace drm {

// ops_i is stored FIELD-MAJOR, [DRM_OPI_STRIDE][CAP]: one scalar load fetches a field of many ops.
#define DRM_OPI(field, k) opi[(field) * CAP + (k)]

// sin / cos of a joint angle, branch-free.  Argument reduction k = rint(x 2/pi),
// r = x - k pi/2 is done in fp64 (two constants), which keeps r exact to fp32
// rounding for |x| < ~1e9 without a slow path; the kernels then use degree-9/10
// minimax polynomials on [-pi/4, pi/4] (coefficients of the classic fdlibm float
// kernels) — max error ~1 ulp, i.e. the same class as torch.sin/cos on the CPU
// (spatial_vector_algebra.py:14-53 evaluates them in fp32).
DRM_HD void sincos_f(float x, float &s, float &c) {
    const double xd = (double)x;
    const double kd = rint(xd * 0.63661977236758134308);   // 2/pi
    double rd = fma(-kd, 1.57079632679489655800e+00, xd);    // pi/2 hi
    rd = fma(-kd, 6.12323399573676603587e-17, rd);           // pi/2 lo
    const float r = (float)rd;
    const int q = (int)kd;
    const float z = r * r;
    float ps = fmaf(z, 2.7557314297e-06f, -1.9841270114e-04f);
    ps = fmaf(z, ps, 8.3333337680e-03f);
    ps = fmaf(z, ps, -1.6666667163e-01f);
    const float sr = fmaf(r * z, ps, r);
    float pc = fmaf(z, -2.7557314297e-07f, 2.4801587642e-05f);
    pc = fmaf(z, pc, -1.3888889225e-03f);
    pc = fmaf(z, pc, 4.1666667908e-02f);
    pc = fmaf(z, pc, -0.5f);
    const float cr = fmaf(z, pc, 1.0f);
    const bool swap = q & 1;
    const float s0 = swap ? cr : sr;
    const float c0 = swap ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

DRM_HD float rsqrt_f(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rsqrtf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}

// out = a x b
DRM_HD void cross3(const float *a, const float *b, float *out) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

// y = M x (M row-major 3x3)
DRM_HD void mat_vec(const float *M, const float *x, float *y) {
#pragma unroll
    for (int r = 0; r < 3; ++r) y[r] = M[r * 3 + 0] * x[0] + M[r * 3 + 1] * x[1] + M[r * 3 + 2] * x[2];
}

// y = M^T x
DRM_HD void matT_vec(const float *M, const float *x, float *y) {
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = M[0 * 3 + c] * x[0] + M[1 * 3 + c] * x[1] + M[2 * 3 + c] * x[2];
}

// J = F * Rot_z(q), c = cos(q), s = sin(q)
// (rigid_body.py:146-156, spatial_vector_algebra.py:42-53).  c = 1, s = 0 gives J == F exactly.
DRM_HD void joint_rot_z(const float *__restrict__ F, float c, float s, float *J) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        J[r * 3 + 0] = F[r * 3 + 0] * c + F[r * 3 + 1] * s;
        J[r * 3 + 1] = F[r * 3 + 1] * c - F[r * 3 + 0] * s;
        J[r * 3 + 2] = F[r * 3 + 2];
    }
}

struct Pose {
    float R[9];
    float p[3];
};

// world pose of a link from its parent's: R = Rp J, p = Rp t + pp
// (robot_model.py:186, spatial_vector_algebra.py:98-103)
DRM_HD void compose(const Pose &par, const float *J, const float *__restrict__ t, Pose &out) {
    Pose tmp;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            tmp.R[r * 3 + c] =
                par.R[r * 3 + 0] * J[0 * 3 + c] + par.R[r * 3 + 1] * J[1 * 3 + c] + par.R[r * 3 + 2] * J[2 * 3 + c];
        tmp.p[r] = par.R[r * 3 + 0] * t[0] + par.R[r * 3 + 1] * t[1] + par.R[r * 3 + 2] * t[2] + par.p[r];
    }
    out = tmp;
}

// child of the root link: the root pose is the identity, so R = J and p = t exactly
DRM_HD void compose_root(const float *J, const float *__restrict__ t, Pose &out) {
#pragma unroll
    for (int i = 0; i < 9; ++i) out.R[i] = J[i];
    out.p[0] = t[0]; out.p[1] = t[1]; out.p[2] = t[2];
}

DRM_HD void pose_identity(Pose &P) {
#pragma unroll
    for (int i = 0; i < 9; ++i) P.R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    P.p[0] = P.p[1] = P.p[2] = 0.0f;
}

// undo the axis canonicalisation of a stored frame R~ = R P, P = P_a D_s:  R[:, pi(c)] = d(c) R~[:, c]
// code = a + 3 * (s < 0);  a = 2: pi = identity; a = 0 (joint about x): pi = (1,2,0); a = 1 (about y): pi = (2,0,1);
// s < 0: d = (1,-1,-1)
DRM_HD void unpermute(int code, float *R) {
    if (code != 2) {
        const float d = code >= 3 ? -1.0f : 1.0f;
        const int perm = code >= 3 ? code - 3 : code;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float a = R[r * 3 + 0], b = d * R[r * 3 + 1], c = d * R[r * 3 + 2];
            if (perm == 0)      { R[r * 3 + 1] = a; R[r * 3 + 2] = b; R[r * 3 + 0] = c; }
            else if (perm == 1) { R[r * 3 + 2] = a; R[r * 3 + 0] = b; R[r * 3 + 1] = c; }
            else                { R[r * 3 + 1] = b; R[r * 3 + 2] = c; }
        }
    }
}

// rotation matrix -> quaternion (x, y, z, w), the reference's branch order
// (spatial_vector_algebra.py:108-136).
DRM_HD void quat_xyzw(const float *R, float *q) {
    const float m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7],
                m22 = R[8];
    float t = ((m00 + m11) + m22) + 1.0f;
    float x, y, z, w;
    if (t > 1.0f) {
        w = t;
        z = m10 - m01;
        y = m02 - m20;
        x = m21 - m12;
    } else {
        int i = 0;
        float mii = m00;
        if (m11 > m00) { i = 1; mii = m11; }
        if (m22 > mii) { i = 2; }
        if (i == 0) {
            t = m00 - (m11 + m22) + 1.0f;
            x = t; y = m01 + m10; z = m20 + m02; w = m21 - m12;
        } else if (i == 1) {
            t = m11 - (m22 + m00) + 1.0f;
            y = t; z = m12 + m21; x = m01 + m10; w = m02 - m20;
        } else {
            t = m22 - (m00 + m11) + 1.0f;
            z = t; x = m20 + m02; y = m12 + m21; w = m10 - m01;
        }
    }
    const float scale = 0.5f * rsqrt_f(t);
    q[0] = x * scale;
    q[1] = y * scale;
    q[2] = z * scale;
    q[3] = w * scale;
}

// cos / sin of every op's joint angle, computed up front so the transcendental work is
// off the serial pose chain.  Branch-free on purpose (one basic block lets the compiler issue
// every scalar load of the walk tables early): fixed joints and padding read DoF 0 and are
// masked to c = 1, s = 0.
template <int CAP, class QF>
DRM_HD void joint_trig(const int (&dof)[CAP], QF qf, float *cs, float *sn) {
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        float c_, s_;
        sincos_f(qf(dof[k] < 0 ? 0 : dof[k]), s_, c_);
        cs[k] = dof[k] < 0 ? 1.0f : c_;
        sn[k] = dof[k] < 0 ? 0.0f : s_;
    }
}

// one field of the (field-major) int table for all ops: a single wide scalar load
template <int CAP>
DRM_HD void load_field(const int32_t *__restrict__ opi, int field, int (&out)[CAP]) {
#pragma unroll
    for (int k = 0; k < CAP; ++k) out[k] = opi[field * CAP + k];
}

// ---------------------------------------------------------------------------
// FK + geometric Jacobian along one chain (robot_model.py:626-667).
// The walk is a chain: op 0 hangs off the root, op k off op k-1.  After the
// call `ee` is the (canonical) pose of the last op, and for every op k:
// z[k] = R~_k e_z (world joint axis), pj[k] = p_k (world joint origin).
// Column d = dof(k) of the Jacobian is (z[k] x (ee.p - pj[k]), z[k]).
// ---------------------------------------------------------------------------
template <int CAP, class QF>
DRM_HD void fk_chain(const float *__restrict__ opf, const int (&dof)[CAP], QF qf, Pose &ee, float (&z)[CAP][3],
                     float (&pj)[CAP][3]) {
    float cs[CAP], sn[CAP];
    joint_trig<CAP>(dof, qf, cs, sn);
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const float *of = opf + k * DRM_OPF_STRIDE;
        float J[9];
        joint_rot_z(of + DRM_OPF_F, cs[k], sn[k], J);
        if (k == 0) compose_root(J, of + DRM_OPF_T, ee);
        else compose(ee, J, of + DRM_OPF_T, ee);
        z[k][0] = ee.R[2]; z[k][1] = ee.R[5]; z[k][2] = ee.R[8];
        pj[k][0] = ee.p[0]; pj[k][1] = ee.p[1]; pj[k][2] = ee.p[2];
    }
}

// ---------------------------------------------------------------------------
// Multi-target FK walk over a (possibly branching) tree (robot_model.py:139-195
// poses only, 223-248).
//   qf(d)                  -> joint angle of DoF d for this sample
//   slot_save(s, Pose) / slot_load(s, Pose&)   -> branch-point poses (kept in LDS by the kernel)
//   emit(t, Pose)          -> called once per target slot t with the TRUE (un-permuted) pose
// ---------------------------------------------------------------------------
template <int CAP, class QF, class SAVE, class LOAD, class EMIT>
DRM_HD void fk_walk(const float *__restrict__ opf, const int32_t *__restrict__ opi, QF qf, SAVE slot_save,
                    LOAD slot_load, EMIT emit) {
    int dof[CAP];
    load_field<CAP>(opi, DRM_OPI_DOF, dof);
    float cs[CAP], sn[CAP];
    joint_trig<CAP>(dof, qf, cs, sn);
    Pose cur;
    pose_identity(cur);
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const float *of = opf + k * DRM_OPF_STRIDE;
        const int src = DRM_OPI(DRM_OPI_SRC, k), save = DRM_OPI(DRM_OPI_SAVE, k), out = DRM_OPI(DRM_OPI_OUT, k);
        float J[9];
        joint_rot_z(of + DRM_OPF_F, cs[k], sn[k], J);
        if (src >= 0) slot_load(src, cur);
        if (src == DRM_SRC_ROOT) compose_root(J, of + DRM_OPF_T, cur);
        else compose(cur, J, of + DRM_OPF_T, cur);
        if (save >= 0) slot_save(save, cur);
        if (out >= 0) {
            Pose P = cur;
            unpermute(DRM_OPI(DRM_OPI_PERM, k), P.R);
            emit(out, P);
        }
    }
}

// ---------------------------------------------------------------------------
// Reverse-mode derivative of the multi-target FK walk with respect to the joint
// angles and the per-link constants F (R_fixed) and t (trans), for a loss that
// depends on the target POSITIONS (the reference's quaternion is not
// differentiable, spatial_vector_algebra.py:108-136).  This is what torch
// autograd produces for the reference's robot_model.py:139-195 + 223-248 with
// learnable `trans` / `rot_angles` (robot_model.py:669-713,
// examples/learn_kinematics_of_iiwa.py:25-61), restated as one adjoint sweep:
//
//   for every op k (reverse order), with r = p_d - p_k over the targets d below k,
//     G_k = sum g_d                 (dL/dp_k, world frame)
//     M_k = sum g_d r^T             (first moment of the target gradients about p_k)
//     dL/dt_k = R_p^T G_k           dL/dF_k = (R_p^T M_k) (R_p F_k)
//     dL/dq_k = z_k . N_k,          z_k = R~_k e_z,  N_k = sum r x g_d  (antisymmetric part of M_k)
//   and (G, M) move to the parent as  G_p += G_k,  M_p += M_k + G_k (p_k - p_p)^T.
//
//   grad_in(t, G)          adds the loss gradient of target slot t to G[3]
//   pose_save / pose_load  branch-point poses (LDS); slots must be unique per branch point
//   adj_add / adj_take     branch-point adjoints (LDS), take = read-and-add
//   gq_out(d, v)           dL/dq of DoF d
//   param_out(k, dF, dt)   per-sample dL/dF (9), dL/dt (3) of op k, only for ops in param_mask
// ---------------------------------------------------------------------------
struct Adjoint {
    float G[3];
    float M[9];
};

template <int CAP, class QF, class GIN, class PSAVE, class PLOAD, class AADD, class ATAKE, class GQ, class PG>
DRM_HD void fk_backward_walk(const float *__restrict__ opf, const int32_t *__restrict__ opi, uint32_t param_mask,
                             bool want_gq, QF qf, GIN grad_in, PSAVE pose_save, PLOAD pose_load, AADD adj_add,
                             ATAKE adj_take, GQ gq_out, PG param_out) {
    int dof[CAP];
    load_field<CAP>(opi, DRM_OPI_DOF, dof);
    float cs[CAP], sn[CAP];
    joint_trig<CAP>(dof, qf, cs, sn);
    // ---- forward: world pose of every op, kept for the adjoint sweep -------
    Pose P[CAP];
    Pose cur;
    pose_identity(cur);
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const float *of = opf + k * DRM_OPF_STRIDE;
        const int src = DRM_OPI(DRM_OPI_SRC, k), save = DRM_OPI(DRM_OPI_SAVE, k);
        float J[9];
        joint_rot_z(of + DRM_OPF_F, cs[k], sn[k], J);
        if (src >= 0) pose_load(src, cur);
        if (src == DRM_SRC_ROOT) compose_root(J, of + DRM_OPF_T, cur);
        else compose(cur, J, of + DRM_OPF_T, cur);
        if (save >= 0) pose_save(save, cur);
        P[k] = cur;
    }
    // ---- adjoint sweep -------------------------------------------------------
    Adjoint carry;
#pragma unroll
    for (int i = 0; i < 3; ++i) carry.G[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) carry.M[i] = 0.0f;
#pragma unroll
    for (int k = CAP - 1; k >= 0; --k) {
        const float *of = opf + k * DRM_OPF_STRIDE;
        const int src = DRM_OPI(DRM_OPI_SRC, k), save = DRM_OPI(DRM_OPI_SAVE, k), out = DRM_OPI(DRM_OPI_OUT, k);
        Adjoint tot;
        if (DRM_OPI(DRM_OPI_FLAGS, k) & DRM_FLAG_CHILD_IS_NEXT) {
            tot = carry;
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) tot.G[i] = 0.0f;
#pragma unroll
            for (int i = 0; i < 9; ++i) tot.M[i] = 0.0f;
        }
        if (out >= 0) grad_in(out, tot.G);
        if (save >= 0) adj_take(save, tot);
        Pose par;
        if (src >= 0) pose_load(src, par);
        else if (src == DRM_SRC_ROOT || k == 0) pose_identity(par);
        else par = P[k > 0 ? k - 1 : 0];
        if (want_gq && dof[k] >= 0) {
            const float Nx = tot.M[7] - tot.M[5], Ny = tot.M[2] - tot.M[6], Nz = tot.M[3] - tot.M[1];
            gq_out(dof[k], P[k].R[2] * Nx + P[k].R[5] * Ny + P[k].R[8] * Nz);
        }
        if ((param_mask >> k) & 1u) {
            float dt[3], A[9], Bm[9], dF[9];
            matT_vec(par.R, tot.G, dt);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    A[r * 3 + c] = par.R[0 * 3 + r] * tot.M[0 * 3 + c] + par.R[1 * 3 + r] * tot.M[1 * 3 + c] +
                                   par.R[2 * 3 + r] * tot.M[2 * 3 + c];
                    Bm[r * 3 + c] = par.R[r * 3 + 0] * of[DRM_OPF_F + 0 * 3 + c] +
                                    par.R[r * 3 + 1] * of[DRM_OPF_F + 1 * 3 + c] +
                                    par.R[r * 3 + 2] * of[DRM_OPF_F + 2 * 3 + c];
                }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    dF[r * 3 + c] = A[r * 3 + 0] * Bm[0 * 3 + c] + A[r * 3 + 1] * Bm[1 * 3 + c] + A[r * 3 + 2] * Bm[2 * 3 + c];
            param_out(k, dF, dt);
        }
        if (src != DRM_SRC_ROOT) {
            const float r[3] = {P[k].p[0] - par.p[0], P[k].p[1] - par.p[1], P[k].p[2] - par.p[2]};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) tot.M[i * 3 + j] += tot.G[i] * r[j];
            if (src >= 0) adj_add(src, tot);
            else carry = tot;
        }
    }
}

// ---------------------------------------------------------------------------
// RNEA over the whole tree (robot_model.py:250-375).  Body-frame Pluecker
// coordinates at the link origin, as in the reference.
//   qf(d, q, qd, qdd)   -> joint state of DoF d
//   tau_out(d, value)   -> torque of DoF d
//   motion / force slots: branch-point state, kept in LDS by the kernel
// ---------------------------------------------------------------------------
struct Motion {
    float w[3];  // angular velocity
    float v[3];  // linear velocity
    float al[3]; // angular acceleration
    float a[3];  // linear acceleration
};

struct Force {
    float l[3]; // linear
    float a[3]; // angular
};

DRM_HD void motion_root(Motion &M, float g) {
#pragma unroll
    for (int i = 0; i < 3; ++i) M.w[i] = M.v[i] = M.al[i] = M.a[i] = 0.0f;
    M.a[2] = g; // robot_model.py:344-350: gravity enters as a base acceleration (0,0,+9.81)
}

template <int CAP, class QF, class TAU, class MSAVE, class MLOAD, class FADD, class FTAKE>
DRM_HD void rnea_walk(const float *__restrict__ opf, const int32_t *__restrict__ opi, int flags, QF qf, TAU tau_out,
                      MSAVE motion_save, MLOAD motion_load, FADD force_add, FTAKE force_take) {
    float cs[CAP], sn[CAP];
    Force f[CAP];
    const float g = (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f;
    Motion cur;
    motion_root(cur, g);

    // ---- forward sweep: velocities, accelerations, body forces -------------
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const float *of = opf + k * DRM_OPF_STRIDE;
        const int dof = DRM_OPI(DRM_OPI_DOF, k), src = DRM_OPI(DRM_OPI_SRC, k), save = DRM_OPI(DRM_OPI_SAVE, k);
        float c_ = 1.0f, s_ = 0.0f, wj = 0.0f, aj = 0.0f;
        if (dof >= 0) {
            float q;
            qf(dof, q, wj, aj); // joint velocity / acceleration along the (canonical +z) joint axis
            sincos_f(q, s_, c_); // (rigid_body.py:133-136, 159-165)
        }
        cs[k] = c_;
        sn[k] = s_;
        float J[9];
        joint_rot_z(of + DRM_OPF_F, c_, s_, J);
        if (src == DRM_SRC_ROOT) motion_root(cur, g);
        if (src >= 0) motion_load(src, cur);
        const float *t = of + DRM_OPF_T;
        // velocity (robot_model.py:189-193): w = J^T w_p + wj e_z ; v = J^T (v_p + w_p x t)
        float tmp[3], x[3];
        Motion N;
        matT_vec(J, cur.w, N.w);
        cross3(cur.w, t, x);
        tmp[0] = cur.v[0] + x[0]; tmp[1] = cur.v[1] + x[1]; tmp[2] = cur.v[2] + x[2];
        matT_vec(J, tmp, N.v);
        N.w[2] += wj;
        // acceleration (robot_model.py:269-277): al = J^T al_p + aj e_z + w x (wj e_z) ; a = J^T (a_p + al_p x t) + v x (wj e_z)
        matT_vec(J, cur.al, N.al);
        cross3(cur.al, t, x);
        tmp[0] = cur.a[0] + x[0]; tmp[1] = cur.a[1] + x[1]; tmp[2] = cur.a[2] + x[2];
        matT_vec(J, tmp, N.a);
        N.al[0] += N.w[1] * wj; N.al[1] -= N.w[0] * wj; N.al[2] += aj;
        N.a[0] += N.v[1] * wj;  N.a[1] -= N.v[0] * wj;
        cur = N;
        if (save >= 0) motion_save(save, cur);
        // body force f = I a + v x* (I v)  (robot_model.py:289-293, spatial_vector_algebra.py:321-338, 215-224)
        const float m = of[DRM_OPF_MASS];
        const float *mc = of + DRM_OPF_MCOM, *Io = of + DRM_OPF_IO;
        float hl[3], ha[3], gl[3], ga[3], y[3];
        cross3(mc, N.w, x);
        hl[0] = m * N.v[0] - x[0]; hl[1] = m * N.v[1] - x[1]; hl[2] = m * N.v[2] - x[2];
        mat_vec(Io, N.w, y);
        cross3(mc, N.v, x);
        ha[0] = y[0] + x[0]; ha[1] = y[1] + x[1]; ha[2] = y[2] + x[2];
        cross3(mc, N.al, x);
        gl[0] = m * N.a[0] - x[0]; gl[1] = m * N.a[1] - x[1]; gl[2] = m * N.a[2] - x[2];
        mat_vec(Io, N.al, y);
        cross3(mc, N.a, x);
        ga[0] = y[0] + x[0]; ga[1] = y[1] + x[1]; ga[2] = y[2] + x[2];
        cross3(N.w, hl, x);
        f[k].l[0] = gl[0] + x[0]; f[k].l[1] = gl[1] + x[1]; f[k].l[2] = gl[2] + x[2];
        cross3(N.w, ha, x);
        cross3(N.v, hl, y);
        f[k].a[0] = ga[0] + (x[0] + y[0]); f[k].a[1] = ga[1] + (x[1] + y[1]); f[k].a[2] = ga[2] + (x[2] + y[2]);
    }

    // ---- backward sweep: accumulate forces towards the root ----------------
    Force carry = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
#pragma unroll
    for (int k = CAP - 1; k >= 0; --k) {
        const float *of = opf + k * DRM_OPF_STRIDE;
        const int dof = DRM_OPI(DRM_OPI_DOF, k), src = DRM_OPI(DRM_OPI_SRC, k), save = DRM_OPI(DRM_OPI_SAVE, k);
        Force tot = f[k];
        if (DRM_OPI(DRM_OPI_FLAGS, k) & DRM_FLAG_CHILD_IS_NEXT) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { tot.l[i] += carry.l[i]; tot.a[i] += carry.a[i]; }
        }
        if (save >= 0) force_take(save, tot); // += children that hang off this branch point, slot reset to 0
        if (dof >= 0) {
            // tau = f.ang . axis (+ damping * qd)   (robot_model.py:353-373); the axis is +z of the stored frame
            float tau = tot.a[2];
            if (flags & DRM_RNEA_DAMPING) {
                float q, qd, qdd;
                qf(dof, q, qd, qdd);
                tau += of[DRM_OPF_DAMP] * qd;
            }
            tau_out(dof, tau);
        }
        if (src != DRM_SRC_ROOT) {
            // force.transform(joint_pose) (spatial_vector_algebra.py:281-291): lin = J f ; ang = t x (J f) + J n
            float J[9], x[3];
            joint_rot_z(of + DRM_OPF_F, cs[k], sn[k], J);
            Force up;
            mat_vec(J, tot.l, up.l);
            mat_vec(J, tot.a, up.a);
            cross3(of + DRM_OPF_T, up.l, x);
            up.a[0] += x[0]; up.a[1] += x[1]; up.a[2] += x[2];
            if (src >= 0) force_add(src, up);
            else carry = up;
        }
    }
}

} // namespace drm
