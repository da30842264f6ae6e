// drm_sample.hpp — per-sample arithmetic of the FK / Jacobian / RNEA walks.
//
// One lane of a wavefront owns one sample and runs these functions with the
// walk's constants (ops_f / ops_i, include/drm_hip.h) as wave-uniform scalar
// operands.  Every loop over ops is fully unrolled against the compile-time
// capacity CAP and guarded by the (uniform) `k < n_ops`, so per-op state lives
// in registers with compile-time indices and the only run-time-indexed
// accesses are LDS reads/writes done by the accessor functors.
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

DRM_HD void sincos_f(float x, float &s, float &c) {
#if defined(__HIP_DEVICE_COMPILE__)
    sincosf(x, &s, &c);
#else
    s = sinf(x);
    c = cosf(x);
#endif
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

// J = F * Rot_axis(theta), c = cos(theta), s = sin(theta) with theta = sign*q
// (rigid_body.py:146-156, spatial_vector_algebra.py:14-53).  For a fixed joint
// c = 1, s = 0 and J == F exactly.
DRM_HD void joint_rot(const float *F, int axis, float c, float s, float *J) {
    if (axis == 2) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            J[r * 3 + 0] = F[r * 3 + 0] * c + F[r * 3 + 1] * s;
            J[r * 3 + 1] = F[r * 3 + 1] * c - F[r * 3 + 0] * s;
            J[r * 3 + 2] = F[r * 3 + 2];
        }
    } else if (axis == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            J[r * 3 + 0] = F[r * 3 + 0];
            J[r * 3 + 1] = F[r * 3 + 1] * c + F[r * 3 + 2] * s;
            J[r * 3 + 2] = F[r * 3 + 2] * c - F[r * 3 + 1] * s;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            J[r * 3 + 0] = F[r * 3 + 0] * c - F[r * 3 + 2] * s;
            J[r * 3 + 1] = F[r * 3 + 1];
            J[r * 3 + 2] = F[r * 3 + 0] * s + F[r * 3 + 2] * c;
        }
    }
}

// world pose of a link from its parent's: R = Rp J, p = Rp t + pp
// (robot_model.py:186, spatial_vector_algebra.py:98-103)
DRM_HD void compose(const float *Rp, const float *pp, const float *J, const float *t, float *R, float *p) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            R[r * 3 + c] = Rp[r * 3 + 0] * J[0 * 3 + c] + Rp[r * 3 + 1] * J[1 * 3 + c] + Rp[r * 3 + 2] * J[2 * 3 + c];
        p[r] = Rp[r * 3 + 0] * t[0] + Rp[r * 3 + 1] * t[1] + Rp[r * 3 + 2] * t[2] + pp[r];
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

struct Pose {
    float R[9];
    float p[3];
};

DRM_HD void pose_identity(Pose &P) {
#pragma unroll
    for (int i = 0; i < 9; ++i) P.R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    P.p[0] = P.p[1] = P.p[2] = 0.0f;
}

// cos / sigma*sin of every op's joint angle, computed up front so the
// transcendental work is off the serial pose chain.
template <int CAP, class QF>
DRM_HD void joint_trig(const int32_t *__restrict__ opi, int n_ops, QF qf, float *cs, float *sn) {
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        cs[k] = 1.0f;
        sn[k] = 0.0f;
        if (k < n_ops) {
            const int d = opi[k * DRM_OPI_STRIDE + DRM_OPI_DOF];
            if (d >= 0) {
                float s_, c_;
                sincos_f(qf(d), s_, c_);
                cs[k] = c_;
                sn[k] = s_ * (float)opi[k * DRM_OPI_STRIDE + DRM_OPI_SIGN];
            }
        }
    }
}

// One FK step of op k: world pose of the op's link from the parent pose `par`
// (ignored when src == DRM_SRC_ROOT: the root pose is the identity, so R = J, p = t).
DRM_HD void fk_step(const float *__restrict__ F, const float *__restrict__ t, int dof, int axis, int src, float c,
                    float s, const Pose &par, Pose &out) {
    float J[9];
    if (dof >= 0) {
        joint_rot(F, axis, c, s, J);
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) J[i] = F[i];
    }
    if (src == DRM_SRC_ROOT) {
#pragma unroll
        for (int i = 0; i < 9; ++i) out.R[i] = J[i];
        out.p[0] = t[0]; out.p[1] = t[1]; out.p[2] = t[2];
    } else {
        Pose tmp;
        compose(par.R, par.p, J, t, tmp.R, tmp.p);
        out = tmp;
    }
}

// ---------------------------------------------------------------------------
// Multi-target FK walk (robot_model.py:139-195 poses only, 223-248).
//   qf(d)              -> joint angle of DoF d for this sample
//   emit(t, Pose)      -> called once per target slot t
// ---------------------------------------------------------------------------
template <int CAP, class QF, class EMIT>
DRM_HD void fk_walk(const float *__restrict__ opf, const int32_t *__restrict__ opi, int n_ops, QF qf, EMIT emit) {
    float cs[CAP], sn[CAP];
    joint_trig<CAP>(opi, n_ops, qf, cs, sn);
    Pose cur, s0, s1, s2, s3;
    pose_identity(cur);
    s0 = cur; s1 = cur; s2 = cur; s3 = cur;
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        if (k < n_ops) {
            const int32_t *oi = opi + k * DRM_OPI_STRIDE;
            const float *of = opf + k * DRM_OPF_STRIDE;
            const int src = oi[DRM_OPI_SRC], save = oi[DRM_OPI_SAVE], out = oi[DRM_OPI_OUT];
            Pose par = cur;
            if (src == 0) par = s0;
            else if (src == 1) par = s1;
            else if (src == 2) par = s2;
            else if (src == 3) par = s3;
            fk_step(of + DRM_OPF_F, of + DRM_OPF_T, oi[DRM_OPI_DOF], oi[DRM_OPI_AXIS], src, cs[k], sn[k], par, cur);
            if (save == 0) s0 = cur;
            else if (save == 1) s1 = cur;
            else if (save == 2) s2 = cur;
            else if (save == 3) s3 = cur;
            if (out >= 0) emit(out, cur);
        }
    }
}

// ---------------------------------------------------------------------------
// FK + geometric Jacobian along one chain (robot_model.py:626-667).
// After the call: ee = pose of the last op (the target link), and for every
// op k that drives a DoF: z[k] = R_k * (sign e_axis) (world joint axis),
// pj[k] = p_k (world joint origin).  Column d = dof(k) of the Jacobian is
// (z[k] x (ee.p - pj[k]), z[k]).
// ---------------------------------------------------------------------------
template <int CAP, class QF>
DRM_HD void fk_chain(const float *__restrict__ opf, const int32_t *__restrict__ opi, int n_ops, QF qf, Pose &ee,
                     float (&z)[CAP][3], float (&pj)[CAP][3]) {
    float cs[CAP], sn[CAP];
    joint_trig<CAP>(opi, n_ops, qf, cs, sn);
    pose_identity(ee);
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        z[k][0] = z[k][1] = z[k][2] = 0.0f;
        pj[k][0] = pj[k][1] = pj[k][2] = 0.0f;
        if (k < n_ops) {
            const int32_t *oi = opi + k * DRM_OPI_STRIDE;
            const float *of = opf + k * DRM_OPF_STRIDE;
            const int dof = oi[DRM_OPI_DOF], axis = oi[DRM_OPI_AXIS];
            fk_step(of + DRM_OPF_F, of + DRM_OPF_T, dof, axis, k == 0 ? DRM_SRC_ROOT : DRM_SRC_PREV, cs[k], sn[k], ee,
                    ee);
            if (dof >= 0) {
                const float sg = (float)oi[DRM_OPI_SIGN];
                const float c0 = axis == 0 ? ee.R[0] : (axis == 1 ? ee.R[1] : ee.R[2]);
                const float c1 = axis == 0 ? ee.R[3] : (axis == 1 ? ee.R[4] : ee.R[5]);
                const float c2 = axis == 0 ? ee.R[6] : (axis == 1 ? ee.R[7] : ee.R[8]);
                z[k][0] = c0 * sg; z[k][1] = c1 * sg; z[k][2] = c2 * sg;
                pj[k][0] = ee.p[0]; pj[k][1] = ee.p[1]; pj[k][2] = ee.p[2];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// RNEA over the whole tree (robot_model.py:250-375).  Body-frame Pluecker
// coordinates at the link origin, as in the reference.
//   qf(d, q, qd, qdd)   -> joint state of DoF d
//   tau_out(d, value)   -> torque of DoF d
// ---------------------------------------------------------------------------
struct Motion {
    float w[3];  // angular velocity
    float v[3];  // linear velocity
    float al[3]; // angular acceleration
    float a[3];  // linear acceleration
};

DRM_HD void motion_root(Motion &M, float g) {
#pragma unroll
    for (int i = 0; i < 3; ++i) M.w[i] = M.v[i] = M.al[i] = M.a[i] = 0.0f;
    M.a[2] = g; // robot_model.py:344-350: gravity enters as a base acceleration (0,0,+9.81)
}

template <int CAP, class QF, class TAU>
DRM_HD void rnea_walk(const float *__restrict__ opf, const int32_t *__restrict__ opi, int n_ops, int flags, QF qf,
                      TAU tau_out) {
    float cs[CAP], sn[CAP];
    float fl[CAP][3], fa[CAP][3];
    Motion cur, s0, s1, s2, s3, root;
    motion_root(root, (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f);
    cur = root; s0 = root; s1 = root; s2 = root; s3 = root;

    // ---- forward sweep: velocities, accelerations, body forces -------------
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        cs[k] = 1.0f; sn[k] = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) fl[k][i] = fa[k][i] = 0.0f;
        if (k < n_ops) {
            const int32_t *oi = opi + k * DRM_OPI_STRIDE;
            const float *of = opf + k * DRM_OPF_STRIDE;
            const int dof = oi[DRM_OPI_DOF], axis = oi[DRM_OPI_AXIS], src = oi[DRM_OPI_SRC], save = oi[DRM_OPI_SAVE];
            const float sg = (float)oi[DRM_OPI_SIGN];
            float q = 0.0f, qd = 0.0f, qdd = 0.0f;
            float J[9];
            if (dof >= 0) {
                qf(dof, q, qd, qdd);
                float s_, c_;
                sincos_f(q, s_, c_);
                cs[k] = c_;
                sn[k] = s_ * sg;
                joint_rot(of + DRM_OPF_F, axis, cs[k], sn[k], J);
            } else {
#pragma unroll
                for (int i = 0; i < 9; ++i) J[i] = of[DRM_OPF_F + i];
            }
            Motion P = cur;
            if (src == DRM_SRC_ROOT) P = root;
            else if (src == 0) P = s0;
            else if (src == 1) P = s1;
            else if (src == 2) P = s2;
            else if (src == 3) P = s3;
            const float *t = of + DRM_OPF_T;
            // joint velocity / acceleration along the joint axis (rigid_body.py:133-136, 159-165)
            const float wj = sg * qd, aj = sg * qdd;
            float jv[3] = {axis == 0 ? wj : 0.0f, axis == 1 ? wj : 0.0f, axis == 2 ? wj : 0.0f};
            float ja[3] = {axis == 0 ? aj : 0.0f, axis == 1 ? aj : 0.0f, axis == 2 ? aj : 0.0f};
            // velocity (robot_model.py:189-193): w = J^T w_p + jv ; v = J^T (v_p + w_p x t)
            float tmp[3], x[3];
            Motion N;
            matT_vec(J, P.w, N.w);
            cross3(P.w, t, x);
            tmp[0] = P.v[0] + x[0]; tmp[1] = P.v[1] + x[1]; tmp[2] = P.v[2] + x[2];
            matT_vec(J, tmp, N.v);
            N.w[0] += jv[0]; N.w[1] += jv[1]; N.w[2] += jv[2];
            // acceleration (robot_model.py:269-277): al = J^T al_p + ja + w x jv ; a = J^T (a_p + al_p x t) + v x jv
            matT_vec(J, P.al, N.al);
            cross3(P.al, t, x);
            tmp[0] = P.a[0] + x[0]; tmp[1] = P.a[1] + x[1]; tmp[2] = P.a[2] + x[2];
            matT_vec(J, tmp, N.a);
            cross3(N.w, jv, x);
            N.al[0] += ja[0] + x[0]; N.al[1] += ja[1] + x[1]; N.al[2] += ja[2] + x[2];
            cross3(N.v, jv, x);
            N.a[0] += x[0]; N.a[1] += x[1]; N.a[2] += x[2];
            cur = N;
            if (save == 0) s0 = cur;
            else if (save == 1) s1 = cur;
            else if (save == 2) s2 = cur;
            else if (save == 3) s3 = cur;
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
            fl[k][0] = gl[0] + x[0]; fl[k][1] = gl[1] + x[1]; fl[k][2] = gl[2] + x[2];
            cross3(N.w, ha, x);
            cross3(N.v, hl, y);
            fa[k][0] = ga[0] + (x[0] + y[0]); fa[k][1] = ga[1] + (x[1] + y[1]); fa[k][2] = ga[2] + (x[2] + y[2]);
        }
    }

    // ---- backward sweep: accumulate forces towards the root ----------------
    float cl[3] = {0, 0, 0}, ca[3] = {0, 0, 0};
    float l0[3] = {0, 0, 0}, a0[3] = {0, 0, 0}, l1[3] = {0, 0, 0}, a1[3] = {0, 0, 0};
    float l2[3] = {0, 0, 0}, a2[3] = {0, 0, 0}, l3[3] = {0, 0, 0}, a3[3] = {0, 0, 0};
#pragma unroll
    for (int k = CAP - 1; k >= 0; --k) {
        if (k < n_ops) {
            const int32_t *oi = opi + k * DRM_OPI_STRIDE;
            const float *of = opf + k * DRM_OPF_STRIDE;
            const int dof = oi[DRM_OPI_DOF], axis = oi[DRM_OPI_AXIS], src = oi[DRM_OPI_SRC], save = oi[DRM_OPI_SAVE];
            float tl[3] = {fl[k][0], fl[k][1], fl[k][2]}, ta[3] = {fa[k][0], fa[k][1], fa[k][2]};
            if (oi[DRM_OPI_FLAGS] & DRM_FLAG_CHILD_IS_NEXT) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { tl[i] += cl[i]; ta[i] += ca[i]; }
            }
#define DRM_TAKE_SLOT(L_, A_)                                                  \
    {                                                                          \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                        \
            tl[i] += L_[i]; ta[i] += A_[i]; L_[i] = 0.0f; A_[i] = 0.0f;        \
        }                                                                      \
    }
            if (save == 0) DRM_TAKE_SLOT(l0, a0)
            else if (save == 1) DRM_TAKE_SLOT(l1, a1)
            else if (save == 2) DRM_TAKE_SLOT(l2, a2)
            else if (save == 3) DRM_TAKE_SLOT(l3, a3)
#undef DRM_TAKE_SLOT
            float J[9];
            if (dof >= 0) {
                // tau = sign * f.ang[axis] (+ damping * qd)   (robot_model.py:353-373)
                float q, qd, qdd;
                qf(dof, q, qd, qdd);
                float tau = (float)oi[DRM_OPI_SIGN] * (axis == 0 ? ta[0] : (axis == 1 ? ta[1] : ta[2]));
                if (flags & DRM_RNEA_DAMPING) tau += of[DRM_OPF_DAMP] * qd;
                tau_out(dof, tau);
                joint_rot(of + DRM_OPF_F, axis, cs[k], sn[k], J);
            } else {
#pragma unroll
                for (int i = 0; i < 9; ++i) J[i] = of[DRM_OPF_F + i];
            }
            if (src != DRM_SRC_ROOT) {
                // force.transform(joint_pose) (spatial_vector_algebra.py:281-291): lin = J f ; ang = t x (J f) + J n
                float gl[3], ga[3], x[3];
                mat_vec(J, tl, gl);
                mat_vec(J, ta, ga);
                cross3(of + DRM_OPF_T, gl, x);
                ga[0] += x[0]; ga[1] += x[1]; ga[2] += x[2];
                if (src == DRM_SRC_PREV) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) { cl[i] = gl[i]; ca[i] = ga[i]; }
                }
#define DRM_ADD_SLOT(L_, A_)                                                   \
    {                                                                          \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) { L_[i] += gl[i]; A_[i] += ga[i]; } \
    }
                else if (src == 0) DRM_ADD_SLOT(l0, a0)
                else if (src == 1) DRM_ADD_SLOT(l1, a1)
                else if (src == 2) DRM_ADD_SLOT(l2, a2)
                else if (src == 3) DRM_ADD_SLOT(l3, a3)
#undef DRM_ADD_SLOT
            }
        }
    }
}

} // namespace drm
