/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of the reference's FK / geometric-Jacobian / RNEA arithmetic
 * (facebookresearch/differentiable-robot-model @ v1), one sample at a time,
 * plain C, following the reference's operation order.  This file is included
 * twice by drm_oracle.c, once with REAL=float (the reference computes in fp32:
 * urdf_utils.py:49,52,68,74,87,92) and once with REAL=double (to bound the
 * fp32 rounding noise of both the reference and the HIP kernels).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this.  Citations are relative to /root/reference/.
 *   rm.py  = differentiable_robot_model/robot_model.py
 *   rb.py  = differentiable_robot_model/rigid_body.py
 *   sva.py = differentiable_robot_model/spatial_vector_algebra.py
 *   ut.py  = differentiable_robot_model/utils.py
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* ---- 3x3 helpers (row-major) ------------------------------------------- */

/* C = A @ B, plain k=0..2 accumulation like a naive bmm */
static void FN(mat3_mul)(const REAL *A, const REAL *B, REAL *C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            REAL acc = A[r * 3 + 0] * B[0 * 3 + c];
            acc += A[r * 3 + 1] * B[1 * 3 + c];
            acc += A[r * 3 + 2] * B[2 * 3 + c];
            C[r * 3 + c] = acc;
        }
}

/* y = A @ x */
static void FN(mat3_vec)(const REAL *A, const REAL *x, REAL *y) {
    for (int r = 0; r < 3; ++r) {
        REAL acc = A[r * 3 + 0] * x[0];
        acc += A[r * 3 + 1] * x[1];
        acc += A[r * 3 + 2] * x[2];
        y[r] = acc;
    }
}

static void FN(mat3_transpose)(const REAL *A, REAL *T) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[c * 3 + r] = A[r * 3 + c];
}

/* ut.py:40-50 vector3_to_skew_symm_matrix */
static void FN(skew)(const REAL *v, REAL *S) {
    S[0] = 0;     S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2];  S[4] = 0;     S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0];  S[8] = 0;
}

/* ut.py:21-25 cross_product(a, b) = S(a) @ b (a matmul with explicit zeros) */
static void FN(cross)(const REAL *a, const REAL *b, REAL *out) {
    REAL S[9];
    FN(skew)(a, S);
    FN(mat3_vec)(S, b, out);
}

/* sva.py:14-53 x_rot / y_rot / z_rot */
static void FN(axis_rot)(int axis, REAL angle, REAL *R) {
    REAL c = COS(angle), s = SIN(angle);
    for (int i = 0; i < 9; ++i) R[i] = 0;
    if (axis == 0) {        /* x_rot sva.py:14-25 */
        R[0] = 1; R[4] = c; R[5] = -s; R[7] = s; R[8] = c;
    } else if (axis == 1) { /* y_rot sva.py:28-39 */
        R[0] = c; R[2] = s; R[4] = 1; R[6] = -s; R[8] = c;
    } else {                /* z_rot sva.py:42-53 */
        R[0] = c; R[1] = -s; R[3] = s; R[4] = c; R[8] = 1;
    }
}

/* rb.py:138-143: fixed_rotation = (z_rot(yaw) @ y_rot(pitch)) @ x_rot(roll) */
static void FN(fixed_rotation)(const REAL *rpy, REAL *F) {
    REAL Rx[9], Ry[9], Rz[9], T[9];
    FN(axis_rot)(0, rpy[0], Rx);
    FN(axis_rot)(1, rpy[1], Ry);
    FN(axis_rot)(2, rpy[2], Rz);
    FN(mat3_mul)(Rz, Ry, T);
    FN(mat3_mul)(T, Rx, F);
}

static REAL FN(sgn)(REAL x) { return (REAL)((x > 0) - (x < 0)); } /* torch.sign */

/*
 * rb.py:130-157 update_joint_state: joint_pose rotation = R_fixed @ Rot_k(sign*q),
 * axis chosen by |axis[0]|==1 -> x, elif |axis[1]|==1 -> y, else z (rb.py:149-154).
 * Fixed joints keep the pose built once in rb.py:64-67 with q = 0.
 */
/*
 * Joint models.  The reference knows ONE: a revolute joint about +-x / y / z (rb.py:149-154; every non-fixed joint,
 * prismatic ones included, rm.py:122-126) — that is what the code below does whenever sp->kind is NULL and the axis has
 * a single +-1 entry, line by line as in the reference.  EXTENSION (no reference behaviour exists to
 * restate, the reference mis-rotates such joints and cannot extract their torque, rm.py:356-358), textbook forms:
 *   - a revolute joint about ANY unit axis a: Rot_a(q) by Rodrigues' formula, joint velocity (ang a qd, lin 0),
 *     torque a . f_ang;
 *   - a prismatic joint (kind 2) along a: no rotation, translation trans + R_fixed a q, joint velocity (ang 0, lin a qd),
 *     torque a . f_lin, Jacobian column (lin R a, ang 0)          (Featherstone, Rigid Body Dynamics Algorithms, ch. 4).
 */
static int FN(is_prismatic)(const drm_oracle_spec *sp, int i) { return sp->kind && sp->kind[i] == 2; }
static int FN(axis_aligned)(const drm_oracle_spec *sp, int i) {
    int nz = 0, one = 0;
    for (int k = 0; k < 3; ++k) {
        const float a = sp->axis[i * 3 + k];
        nz += a != 0;
        one += (a == 1.0f || a == -1.0f);
    }
    return nz <= 1 && (nz == 0 || one == 1);
}
static void FN(unit_axis)(const drm_oracle_spec *sp, int i, REAL *a) {
    double x = sp->axis[i * 3], y = sp->axis[i * 3 + 1], z = sp->axis[i * 3 + 2];
    double nrm = sqrt(x * x + y * y + z * z);
    if (FN(axis_aligned)(sp, i) || nrm == 0) nrm = 1;   /* the reference uses the axis as written */
    a[0] = (REAL)(x / nrm); a[1] = (REAL)(y / nrm); a[2] = (REAL)(z / nrm);
}
static void FN(joint_rotation)(const drm_oracle_spec *sp, int i, REAL q, REAL *J) {
    REAL rpy[3] = {(REAL)sp->rpy[i * 3], (REAL)sp->rpy[i * 3 + 1], (REAL)sp->rpy[i * 3 + 2]};
    REAL ax[3] = {(REAL)sp->axis[i * 3], (REAL)sp->axis[i * 3 + 1], (REAL)sp->axis[i * 3 + 2]};
    REAL F[9], Rq[9];
    FN(fixed_rotation)(rpy, F);
    if (FN(is_prismatic)(sp, i)) {                      /* extension: a sliding joint does not rotate */
        for (int k = 0; k < 9; ++k) J[k] = F[k];
        return;
    }
    if (FN(axis_aligned)(sp, i)) {                      /* the reference, rb.py:149-154 */
        if (FABS(ax[0]) == 1)      FN(axis_rot)(0, FN(sgn)(ax[0]) * q, Rq);
        else if (FABS(ax[1]) == 1) FN(axis_rot)(1, FN(sgn)(ax[1]) * q, Rq);
        else                       FN(axis_rot)(2, FN(sgn)(ax[2]) * q, Rq);
    } else {                                            /* extension: Rodrigues, R = I + s K + (1 - c) K^2 */
        REAL a[3], K[9], KK[9];
        FN(unit_axis)(sp, i, a);
        FN(skew)(a, K);
        FN(mat3_mul)(K, K, KK);
        const REAL c = COS(q), s_ = SIN(q);
        for (int k = 0; k < 9; ++k) Rq[k] = ((k % 4 == 0) ? (REAL)1 : (REAL)0) + s_ * K[k] + ((REAL)1 - c) * KK[k];
    }
    FN(mat3_mul)(F, Rq, J);
}
/* joint origin in the parent frame: `trans` (rb.py:143-145), plus the slide of a prismatic joint (extension) */
static void FN(joint_trans)(const drm_oracle_spec *sp, int i, REAL q, REAL *t) {
    for (int k = 0; k < 3; ++k) t[k] = (REAL)sp->trans[i * 3 + k];
    if (FN(is_prismatic)(sp, i)) {
        REAL rpy[3] = {(REAL)sp->rpy[i * 3], (REAL)sp->rpy[i * 3 + 1], (REAL)sp->rpy[i * 3 + 2]};
        REAL F[9], a[3], Fa[3];
        FN(fixed_rotation)(rpy, F);
        FN(unit_axis)(sp, i, a);
        FN(mat3_vec)(F, a, Fa);
        for (int k = 0; k < 3; ++k) t[k] += Fa[k] * q;
    }
}
/* joint velocity / acceleration (lin, ang) for a joint rate r: rb.py:133-136, 159-165 = (0, r axis); prismatic (r axis, 0) */
static void FN(joint_rate)(const drm_oracle_spec *sp, int i, REAL r, REAL *lin, REAL *ang) {
    for (int k = 0; k < 3; ++k) { lin[k] = 0; ang[k] = 0; }
    if (FN(is_prismatic)(sp, i) || !FN(axis_aligned)(sp, i)) {
        REAL a[3];
        FN(unit_axis)(sp, i, a);
        for (int k = 0; k < 3; ++k) (FN(is_prismatic)(sp, i) ? lin : ang)[k] = r * a[k];
    } else {
        for (int k = 0; k < 3; ++k) ang[k] = r * (REAL)sp->axis[i * 3 + k];
    }
}

/*
 * rm.py:139-195 update_kinematic_state for ONE sample.
 *   R[L*9], p[L*3]  world pose of every link         (rm.py:186, sva.py:98-103)
 *   J[L*9]          joint (child->parent) rotations   (rb.py:146-156)
 *   vl/va[L*3]      body-frame spatial velocity       (rm.py:189-193, sva.py:226-236)
 * qd may be NULL (treated as 0, as compute_forward_kinematics does, rm.py:243).
 */
static void FN(kinematic_state)(const drm_oracle_spec *sp, const REAL *q, const REAL *qd,
                                REAL *R, REAL *p, REAL *J, REAL *vl, REAL *va) {
    const int L = sp->n_links;
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1 : 0; /* CoordinateTransform() default sva.py:60-72 */
    for (int i = 0; i < 3; ++i) { p[i] = 0; vl[i] = 0; va[i] = 0; }      /* rm.py:166-170 */
    for (int i = 0; i < 9; ++i) J[i] = (i % 4 == 0) ? 1 : 0;
    for (int i = 1; i < L; ++i) {
        const int par = sp->parent[i], d = sp->dof[i];
        REAL qi = (d >= 0) ? q[d] : 0, qdi = (d >= 0 && qd) ? qd[d] : 0;
        REAL t[3];
        FN(joint_trans)(sp, i, qi, t);
        REAL *Ji = J + i * 9;
        FN(joint_rotation)(sp, i, qi, Ji);
        /* body.pose = parent.pose.multiply_transform(joint_pose)  rm.py:186, sva.py:98-103 */
        FN(mat3_mul)(R + par * 9, Ji, R + i * 9);
        REAL Rt[3];
        FN(mat3_vec)(R + par * 9, t, Rt);
        for (int k = 0; k < 3; ++k) p[i * 3 + k] = Rt[k] + p[par * 3 + k];
        /* parentToChildT = joint_pose.inverse()  sva.py:92-96: (R^T, -(R^T t)) */
        REAL JT[9], it[3];
        FN(mat3_transpose)(Ji, JT);
        FN(mat3_vec)(JT, t, it);
        for (int k = 0; k < 3; ++k) it[k] = -it[k];
        /* new_vel = parent.vel.transform(parentToChildT)  sva.py:226-236 */
        REAL S[9], SR[9], na[3], nl[3], nl2[3];
        FN(mat3_vec)(JT, va + par * 3, na);
        FN(skew)(it, S);
        FN(mat3_mul)(S, JT, SR);                 /* trans_cross_rot sva.py:105-106 */
        FN(mat3_vec)(SR, va + par * 3, nl);
        FN(mat3_vec)(JT, vl + par * 3, nl2);
        /* body.vel = joint_vel.add_motion_vec(new_vel); joint_vel = (0, qd @ axis)  rb.py:133-136, rm.py:193 */
        REAL jvl[3], jva[3];
        FN(joint_rate)(sp, i, qdi, jvl, jva);
        for (int k = 0; k < 3; ++k) {
            vl[i * 3 + k] = jvl[k] + (nl[k] + nl2[k]);
            va[i * 3 + k] = jva[k] + na[k];
        }
    }
}

/*
 * sva.py:108-136 CoordinateTransform.get_quaternion (xyzw).  The reference
 * embeds (R, p) in a 4x4 M, so M[3][3] = 1 and trace(M) = tr(R) + 1.
 * Scale: q *= 0.5 / math.sqrt(t * M33)  (python double sqrt, then an fp32 multiply).
 */
static void FN(quaternion)(const REAL *Rm, REAL *quat) {
#define M(r, c) Rm[(r) * 3 + (c)]
    REAL t = ((M(0, 0) + M(1, 1)) + M(2, 2)) + (REAL)1; /* einsum("bii->b") over the 4x4 */
    REAL q[4];
    if (t > (REAL)1) {
        q[3] = t;
        q[2] = M(1, 0) - M(0, 1);
        q[1] = M(0, 2) - M(2, 0);
        q[0] = M(2, 1) - M(1, 2);
    } else {
        int i = 0, j = 1, k = 2;
        if (M(1, 1) > M(0, 0)) { i = 1; j = 2; k = 0; }
        if (M(2, 2) > M(i, i)) { i = 2; j = 0; k = 1; }
        t = M(i, i) - (M(j, j) + M(k, k)) + (REAL)1;
        q[i] = t;
        q[j] = M(i, j) + M(j, i);
        q[k] = M(k, i) + M(i, k);
        q[3] = M(k, j) - M(j, k);
    }
    REAL scale = (REAL)(0.5 / sqrt((double)t));
    for (int n = 0; n < 4; ++n) quat[n] = q[n] * scale;
#undef M
}

/* sva.py:321-338 DifferentiableSpatialRigidBodyInertia.multiply_motion_vec */
static void FN(inertia_mul)(const drm_oracle_spec *sp, int i, const REAL *lin, const REAL *ang,
                            REAL *flin, REAL *fang) {
    REAL mass = (REAL)sp->mass[i];
    REAL com[3], mcom[3], I[9], S[9], ST[9], SS[9], inertia[9];
    for (int k = 0; k < 3; ++k) { com[k] = (REAL)sp->com[i * 3 + k]; mcom[k] = com[k] * mass; }
    for (int k = 0; k < 9; ++k) I[k] = (REAL)sp->inertia[i * 9 + k];
    FN(skew)(com, S);
    FN(mat3_transpose)(S, ST);
    FN(mat3_mul)(S, ST, SS);
    for (int k = 0; k < 9; ++k) inertia[k] = I[k] + mass * SS[k];
    REAL c1[3], c2[3], Iw[3];
    FN(cross)(mcom, ang, c1);
    FN(cross)(mcom, lin, c2);
    FN(mat3_vec)(inertia, ang, Iw);
    for (int k = 0; k < 3; ++k) {
        flin[k] = mass * lin[k] - c1[k];
        fang[k] = Iw[k] + c2[k];
    }
}

/*
 * rm.py:305-375 compute_inverse_dynamics for ONE sample
 * (update_kinematic_state + update_joint_acc + iterative_newton_euler rm.py:250-303).
 * scratch: caller provides 45*L REALs.
 */
static void FN(rnea_sample)(const drm_oracle_spec *sp, const REAL *q, const REAL *qd, const REAL *qdd,
                            int include_gravity, int use_damping, REAL *tau, REAL *scratch) {
    const int L = sp->n_links;
    REAL *R = scratch, *p = R + 9 * L, *J = p + 3 * L, *vl = J + 9 * L, *va = vl + 3 * L;
    REAL *al = va + 3 * L, *aa = al + 3 * L, *fl = aa + 3 * L, *fa = fl + 3 * L;
    FN(kinematic_state)(sp, q, qd, R, p, J, vl, va);
    /* base acceleration rm.py:344-350 */
    al[0] = 0; al[1] = 0; al[2] = include_gravity ? (REAL)9.81 : 0;
    aa[0] = aa[1] = aa[2] = 0;
    /* forward pass rm.py:262-277 */
    for (int i = 1; i < L; ++i) {
        const int par = sp->parent[i], d = sp->dof[i];
        REAL qdi = (d >= 0) ? qd[d] : 0, qddi = (d >= 0) ? qdd[d] : 0;
        REAL t[3];
        FN(joint_trans)(sp, i, (d >= 0) ? q[d] : 0, t);
        REAL JT[9], it[3], S[9], SR[9];
        FN(mat3_transpose)(J + i * 9, JT);
        FN(mat3_vec)(JT, t, it);
        for (int k = 0; k < 3; ++k) it[k] = -it[k];
        FN(skew)(it, S);
        FN(mat3_mul)(S, JT, SR);
        REAL na[3], nl[3], nl2[3];
        FN(mat3_vec)(JT, aa + par * 3, na);
        FN(mat3_vec)(SR, aa + par * 3, nl);
        FN(mat3_vec)(JT, al + par * 3, nl2);
        /* joint_vel / joint_acc = (0, qd*axis) / (0, qdd*axis)  rb.py:133-136,159-165 */
        REAL jv[3], ja[3], jvl[3], jal[3];
        FN(joint_rate)(sp, i, qdi, jvl, jv);
        FN(joint_rate)(sp, i, qddi, jal, ja);
        /* tmp = body.vel.cross_motion_vec(body.joint_vel)  sva.py:204-213 */
        REAL ta[3], tl1[3], tl2[3];
        FN(cross)(va + i * 3, jv, ta);
        FN(cross)(va + i * 3, jvl, tl1);
        FN(cross)(vl + i * 3, jv, tl2);
        for (int k = 0; k < 3; ++k) {
            al[i * 3 + k] = ((nl[k] + nl2[k]) + jal[k]) + (tl1[k] + tl2[k]);
            aa[i * 3 + k] = (na[k] + ja[k]) + ta[k];
        }
    }
    for (int i = 0; i < 3 * L; ++i) { fl[i] = 0; fa[i] = 0; } /* rm.py:280-281 */
    /* backward pass rm.py:284-301 */
    for (int i = L - 1; i > 0; --i) {
        const int par = sp->parent[i];
        REAL ial[3], iaa[3], ivl[3], iva[3], c1[3], c2[3], c3[3];
        FN(inertia_mul)(sp, i, al + i * 3, aa + i * 3, ial, iaa);
        FN(inertia_mul)(sp, i, vl + i * 3, va + i * 3, ivl, iva);
        /* vel.cross_force_vec(icxvel)  sva.py:215-224 */
        FN(cross)(va + i * 3, iva, c1);
        FN(cross)(vl + i * 3, ivl, c2);
        FN(cross)(va + i * 3, ivl, c3);
        for (int k = 0; k < 3; ++k) {
            fl[i * 3 + k] = (fl[i * 3 + k] + ial[k]) + c3[k];
            fa[i * 3 + k] = (fa[i * 3 + k] + iaa[k]) + (c1[k] + c2[k]);
        }
        /* backprop_force = body.force.transform(joint_pose)  sva.py:281-291 */
        REAL t[3];
        FN(joint_trans)(sp, i, (sp->dof[i] >= 0) ? q[sp->dof[i]] : 0, t);
        REAL S[9], SR[9], bl[3], ba[3], ba2[3];
        FN(mat3_vec)(J + i * 9, fl + i * 3, bl);
        FN(skew)(t, S);
        FN(mat3_mul)(S, J + i * 9, SR);
        FN(mat3_vec)(SR, fl + i * 3, ba);
        FN(mat3_vec)(J + i * 9, fa + i * 3, ba2);
        for (int k = 0; k < 3; ++k) {
            fl[par * 3 + k] += bl[k];
            fa[par * 3 + k] += ba[k] + ba2[k];
        }
    }
    /* torque extraction rm.py:353-365 and damping rm.py:368-373 */
    for (int i = 1; i < L; ++i) {
        const int d = sp->dof[i];
        if (d < 0) continue;
        REAL f;
        if (FN(is_prismatic)(sp, i) || !FN(axis_aligned)(sp, i)) {   /* extension: tau = S^T f */
            REAL a[3];
            FN(unit_axis)(sp, i, a);
            const REAL *ff = FN(is_prismatic)(sp, i) ? fl + i * 3 : fa + i * 3;
            f = (a[0] * ff[0] + a[1] * ff[1]) + a[2] * ff[2];
        } else {
            int k = 0; /* int(torch.where(axis)[0]) */
            while (k < 2 && sp->axis[i * 3 + k] == 0) ++k;
            REAL sign = FN(sgn)((REAL)sp->axis[i * 3 + k]);
            f = (REAL)0 + sign * fa[i * 3 + k];
        }
        if (use_damping) f += (REAL)sp->damping[i] * qd[d];
        tau[d] = f;
    }
}

/* ---- batched entry points ---------------------------------------------- */

/* rm.py:223-248 compute_forward_kinematics (non-recursive) for T target links. */
static int FN(oracle_fk)(const drm_oracle_spec *sp, const IO_T *q, int64_t B, const int *targets, int T,
                         IO_T *pos, IO_T *quat) {
    const int L = sp->n_links, n = sp->n_dofs;
    int err = 0;
#pragma omp parallel
    {
        REAL *buf = (REAL *)malloc(sizeof(REAL) * (size_t)(27 * L + n + 8));
        if (!buf) {
#pragma omp atomic write
            err = 1;
        } else {
            REAL *R = buf, *p = R + 9 * L, *J = p + 3 * L, *vl = J + 9 * L, *va = vl + 3 * L, *qq = va + 3 * L;
#pragma omp for schedule(static)
            for (int64_t b = 0; b < B; ++b) {
                for (int d = 0; d < n; ++d) qq[d] = (REAL)q[b * n + d];
                FN(kinematic_state)(sp, qq, NULL, R, p, J, vl, va);
                for (int t = 0; t < T; ++t) {
                    REAL qt[4];
                    FN(quaternion)(R + targets[t] * 9, qt);
                    for (int k = 0; k < 3; ++k) pos[(b * T + t) * 3 + k] = (IO_T)p[targets[t] * 3 + k];
                    for (int k = 0; k < 4; ++k) quat[(b * T + t) * 4 + k] = (IO_T)qt[k];
                }
            }
            free(buf);
        }
    }
    return err;
}

/* world poses of ALL links: Rw[B,L,9], pw[B,L,3] (used to check body.pose state, rm.py:186) */
static int FN(oracle_fk_all_poses)(const drm_oracle_spec *sp, const IO_T *q, int64_t B, IO_T *Rw, IO_T *pw) {
    const int L = sp->n_links, n = sp->n_dofs;
    REAL *buf = (REAL *)malloc(sizeof(REAL) * (size_t)(27 * L + n + 8));
    if (!buf) return 1;
    REAL *R = buf, *p = R + 9 * L, *J = p + 3 * L, *vl = J + 9 * L, *va = vl + 3 * L, *qq = va + 3 * L;
    for (int64_t b = 0; b < B; ++b) {
        for (int d = 0; d < n; ++d) qq[d] = (REAL)q[b * n + d];
        FN(kinematic_state)(sp, qq, NULL, R, p, J, vl, va);
        for (int k = 0; k < 9 * L; ++k) Rw[b * 9 * L + k] = (IO_T)R[k];
        for (int k = 0; k < 3 * L; ++k) pw[b * 3 * L + k] = (IO_T)p[k];
    }
    free(buf);
    return 0;
}

/*
 * rm.py:626-667 compute_endeffector_jacobian; also returns the FK of the link
 * (the reference computes it at rm.py:641 and discards the quaternion).
 * pos/quat may be NULL.
 */
static int FN(oracle_fk_jacobian)(const drm_oracle_spec *sp, const IO_T *q, int64_t B, int link,
                                  IO_T *pos, IO_T *quat, IO_T *lin_jac, IO_T *ang_jac) {
    const int L = sp->n_links, n = sp->n_dofs;
    int err = 0;
#pragma omp parallel
    {
        REAL *buf = (REAL *)malloc(sizeof(REAL) * (size_t)(27 * L + n + 8));
        if (!buf) {
#pragma omp atomic write
            err = 1;
        } else {
            REAL *R = buf, *p = R + 9 * L, *J = p + 3 * L, *vl = J + 9 * L, *va = vl + 3 * L, *qq = va + 3 * L;
#pragma omp for schedule(static)
            for (int64_t b = 0; b < B; ++b) {
                for (int d = 0; d < n; ++d) qq[d] = (REAL)q[b * n + d];
                FN(kinematic_state)(sp, qq, NULL, R, p, J, vl, va);
                if (pos) for (int k = 0; k < 3; ++k) pos[b * 3 + k] = (IO_T)p[link * 3 + k];
                if (quat) {
                    REAL qt[4];
                    FN(quaternion)(R + link * 9, qt);
                    for (int k = 0; k < 4; ++k) quat[b * 4 + k] = (IO_T)qt[k];
                }
                IO_T *lj = lin_jac + b * 3 * n, *aj = ang_jac + b * 3 * n;
                for (int k = 0; k < 3 * n; ++k) { lj[k] = 0; aj[k] = 0; } /* rm.py:646-649 */
                /* walk link -> root  rm.py:651-665 */
                for (int i = link; i != 0; i = sp->parent[i]) {
                    const int d = sp->dof[i];
                    if (d < 0) continue;
                    REAL ax[3], z[3], dp[3], c[3];
                    FN(unit_axis)(sp, i, ax);
                    FN(mat3_vec)(R + i * 9, ax, z);                           /* rm.py:660 */
                    if (FN(is_prismatic)(sp, i)) {                            /* extension: column (z, 0) */
                        for (int k = 0; k < 3; ++k) { lj[k * n + d] = (IO_T)z[k]; aj[k * n + d] = 0; }
                        continue;
                    }
                    for (int k = 0; k < 3; ++k) dp[k] = p[link * 3 + k] - p[i * 3 + k];
                    c[0] = z[1] * dp[2] - z[2] * dp[1];                       /* torch.cross rm.py:661 */
                    c[1] = z[2] * dp[0] - z[0] * dp[2];
                    c[2] = z[0] * dp[1] - z[1] * dp[0];
                    for (int k = 0; k < 3; ++k) { lj[k * n + d] = (IO_T)c[k]; aj[k * n + d] = (IO_T)z[k]; }
                }
            }
            free(buf);
        }
    }
    return err;
}

/* rm.py:305-375 compute_inverse_dynamics */
static int FN(oracle_rnea)(const drm_oracle_spec *sp, const IO_T *q, const IO_T *qd, const IO_T *qdd, int64_t B,
                           int include_gravity, int use_damping, IO_T *tau) {
    const int L = sp->n_links, n = sp->n_dofs;
    int err = 0;
#pragma omp parallel
    {
        REAL *buf = (REAL *)malloc(sizeof(REAL) * (size_t)(45 * L + 4 * n + 8));
        if (!buf) {
#pragma omp atomic write
            err = 1;
        } else {
            REAL *qq = buf + 45 * L, *qv = qq + n, *qa = qv + n, *tt = qa + n;
#pragma omp for schedule(static)
            for (int64_t b = 0; b < B; ++b) {
                for (int d = 0; d < n; ++d) {
                    qq[d] = (REAL)q[b * n + d]; qv[d] = (REAL)qd[b * n + d]; qa[d] = (REAL)qdd[b * n + d];
                    tt[d] = 0;
                }
                FN(rnea_sample)(sp, qq, qv, qa, include_gravity, use_damping, tt, buf);
                for (int d = 0; d < n; ++d) tau[b * n + d] = (IO_T)tt[d];
            }
            free(buf);
        }
    }
    return err;
}

/* rm.py:402-450 compute_lagrangian_inertia_matrix: column j = ID(q, 0, e_j) - ID(q, 0, 0) (n + 1 inverse-dynamics
 * passes; the subtraction of the gravity / damping term is done in REAL like the reference does it in fp32) */
static int FN(oracle_mass_matrix)(const drm_oracle_spec *sp, const IO_T *q, int64_t B, int include_gravity,
                                  int use_damping, IO_T *H) {
    const int L = sp->n_links, n = sp->n_dofs;
    int err = 0;
#pragma omp parallel
    {
        REAL *buf = (REAL *)malloc(sizeof(REAL) * (size_t)(45 * L + 5 * n + 8));
        if (!buf) {
#pragma omp atomic write
            err = 1;
        } else {
            REAL *qq = buf + 45 * L, *zero = qq + n, *ej = zero + n, *t0 = ej + n, *tj = t0 + n;
#pragma omp for schedule(static)
            for (int64_t b = 0; b < B; ++b) {
                for (int d = 0; d < n; ++d) { qq[d] = (REAL)q[b * n + d]; zero[d] = 0; ej[d] = 0; t0[d] = 0; }
                if (include_gravity) FN(rnea_sample)(sp, qq, zero, zero, include_gravity, use_damping, t0, buf);
                for (int j = 0; j < n; ++j) {
                    ej[j] = 1;
                    for (int d = 0; d < n; ++d) tj[d] = 0;
                    FN(rnea_sample)(sp, qq, zero, ej, include_gravity, use_damping, tj, buf);
                    ej[j] = 0;
                    for (int i = 0; i < n; ++i) H[(b * n + i) * n + j] = (IO_T)(tj[i] - t0[i]);
                }
            }
            free(buf);
        }
    }
    return err;
}


/* ---- articulated-body algorithm ------------------------------------------------------------------- */
/* 6-vectors and 6x6 matrices use the reference's get_vector() order: [ang (3); lin (3)] (sva.py:236-237). */

/* DifferentiableSpatialRigidBodyInertia.get_spatial_mat  sva.py:340-372 */
static void FN(spatial_inertia_mat)(const drm_oracle_spec *sp, int i, REAL *M) {
    REAL m = (REAL)sp->mass[i];
    REAL c[3] = {(REAL)sp->com[i * 3], (REAL)sp->com[i * 3 + 1], (REAL)sp->com[i * 3 + 2]};
    REAL mc[3] = {m * c[0], m * c[1], m * c[2]};
    REAL S[9], ST[9], SS[9];
    FN(skew)(c, S);
    FN(mat3_transpose)(S, ST);
    FN(mat3_mul)(S, ST, SS);
    for (int k = 0; k < 36; ++k) M[k] = 0;
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) M[r * 6 + k] = (REAL)sp->inertia[i * 9 + r * 3 + k] + m * SS[r * 3 + k];
    M[3 * 6 + 1] = mc[2];  M[3 * 6 + 2] = -mc[1];
    M[4 * 6 + 0] = -mc[2]; M[4 * 6 + 2] = mc[0];
    M[5 * 6 + 0] = mc[1];  M[5 * 6 + 1] = -mc[0];
    M[0 * 6 + 4] = -mc[2]; M[0 * 6 + 5] = mc[1];
    M[1 * 6 + 3] = mc[2];  M[1 * 6 + 5] = -mc[0];
    M[2 * 6 + 3] = -mc[1]; M[2 * 6 + 4] = mc[0];
    M[3 * 6 + 3] = m; M[4 * 6 + 4] = m; M[5 * 6 + 5] = m;
}

/*
 * rm.py:487-624 compute_forward_dynamics for ONE sample (Featherstone's articulated-body algorithm as the
 * reference writes it, including its 1e-37 smoothing and the `parent_idx > 0` guard).  The reference subtracts the
 * damping torques from its INPUT tensor in place (rm.py:515-521); here the caller's array is left alone.
 * scratch: caller provides (27 + 36 + 6*5 + 2) * L REALs.
 */
static void FN(aba_sample)(const drm_oracle_spec *sp, const REAL *q, const REAL *qd, const REAL *f_in,
                           int include_gravity, int use_damping, REAL *qdd, REAL *scratch) {
    const int L = sp->n_links;
    REAL *R = scratch, *p = R + 9 * L, *J = p + 3 * L, *vl = J + 9 * L, *va = vl + 3 * L;
    REAL *IA = va + 3 * L;                 /* [L][36] */
    REAL *pA = IA + 36 * L;                /* [L][6]  (ang, lin) */
    REAL *cc = pA + 6 * L, *U = cc + 6 * L, *acc = U + 6 * L, *Sv = acc + 6 * L; /* [L][6] each */
    REAL *dd = Sv + 6 * L, *uu = dd + L;   /* [L] */
    FN(kinematic_state)(sp, q, qd, R, p, J, vl, va);
    for (int i = 1; i < L; ++i) { /* rm.py:539-549 */
        const int d = sp->dof[i];
        REAL qdi = (d >= 0) ? qd[d] : 0;
        REAL jv[3], jvl[3], ta[3], tl1[3], tl2[3];
        FN(joint_rate)(sp, i, qdi, jvl, jv);
        /* c = vel.cross_motion_vec(joint_vel)  sva.py:204-213 */
        FN(cross)(va + i * 3, jv, ta);
        FN(cross)(va + i * 3, jvl, tl1);
        FN(cross)(vl + i * 3, jv, tl2);
        for (int k = 0; k < 3; ++k) { cc[i * 6 + k] = ta[k]; cc[i * 6 + 3 + k] = tl1[k] + tl2[k]; }
        /* pA = vel.cross_force_vec(I vel)  sva.py:215-224 */
        REAL ivl[3], iva[3], c1[3], c2[3], c3[3];
        FN(inertia_mul)(sp, i, vl + i * 3, va + i * 3, ivl, iva);
        FN(cross)(va + i * 3, iva, c1);
        FN(cross)(vl + i * 3, ivl, c2);
        FN(cross)(va + i * 3, ivl, c3);
        for (int k = 0; k < 3; ++k) { pA[i * 6 + k] = c1[k] + c2[k]; pA[i * 6 + 3 + k] = c3[k]; }
        FN(spatial_inertia_mat)(sp, i, IA + i * 36);
    }
    for (int i = L - 1; i > 0; --i) { /* rm.py:551-604 */
        const int par = sp->parent[i], d = sp->dof[i];
        REAL *S = Sv + i * 6, *Ui = U + i * 6, *IAi = IA + i * 36;
        FN(joint_rate)(sp, i, (REAL)1, S + 3, S);   /* S = (ang, lin) of a unit joint rate; (axis, 0) in the reference */
        for (int r = 0; r < 6; ++r) {
            REAL a = 0;
            for (int k = 0; k < 6; ++k) a += IAi[r * 6 + k] * S[k];
            Ui[r] = a;
        }
        REAL dot = 0, pAS = 0;
        for (int k = 0; k < 3; ++k) dot += S[k] * Ui[k];          /* S.dot(U): ang part ... */
        { REAL l = 0; for (int k = 0; k < 3; ++k) l += S[3 + k] * Ui[3 + k]; dot = dot + l; }
        for (int k = 0; k < 3; ++k) pAS += pA[i * 6 + k] * S[k];
        { REAL l = 0; for (int k = 0; k < 3; ++k) l += pA[i * 6 + 3 + k] * S[3 + k]; pAS = pAS + l; }
        dd[i] = dot;
        REAL f_i = 0;
        if (d >= 0) f_i = f_in[d] - (use_damping ? (REAL)sp->damping[i] * qd[d] : 0);
        uu[i] = (d >= 0) ? f_i - pAS : -pAS;
        if (par > 0) {
            REAL Ud[6], IAn[36], tmp[6], pa[6];
            for (int k = 0; k < 6; ++k) Ud[k] = Ui[k] / (dd[i] + (REAL)1e-37);
            for (int r = 0; r < 6; ++r)
                for (int k = 0; k < 6; ++k) IAn[r * 6 + k] = IAi[r * 6 + k] - Ui[r] * Ud[k];
            for (int r = 0; r < 6; ++r) {
                REAL a = 0;
                for (int k = 0; k < 6; ++k) a += IAn[r * 6 + k] * cc[i * 6 + k];
                tmp[r] = a;
            }
            REAL ud = uu[i] / (dd[i] + (REAL)1e-37);
            for (int k = 0; k < 6; ++k) pa[k] = (pA[i * 6 + k] + tmp[k]) + Ui[k] * ud;
            /* X = joint_pose.to_matrix()  sva.py:138-154: [[R^T, 0], [-R^T S(t), R^T]] */
            REAL t[3];
            FN(joint_trans)(sp, i, (d >= 0) ? q[d] : 0, t);
            REAL JT[9], St[9], Erx[9], X[36], XtI[36];
            FN(mat3_transpose)(J + i * 9, JT);
            FN(skew)(t, St);
            FN(mat3_mul)(JT, St, Erx);
            for (int k = 0; k < 36; ++k) X[k] = 0;
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) {
                    X[r * 6 + k] = JT[r * 3 + k];
                    X[(3 + r) * 6 + k] = -Erx[r * 3 + k];
                    X[(3 + r) * 6 + 3 + k] = JT[r * 3 + k];
                }
            for (int r = 0; r < 6; ++r)
                for (int k = 0; k < 6; ++k) {
                    REAL a = 0;
                    for (int m = 0; m < 6; ++m) a += X[m * 6 + r] * IAn[m * 6 + k];
                    XtI[r * 6 + k] = a;
                }
            for (int r = 0; r < 6; ++r)
                for (int k = 0; k < 6; ++k) {
                    REAL a = 0;
                    for (int m = 0; m < 6; ++m) a += XtI[r * 6 + m] * X[m * 6 + k];
                    IA[par * 36 + r * 6 + k] += a;
                }
            /* parent.pA += pa.transform(joint_pose)  sva.py:281-291: lin = R f ; ang = S(t) R f + R n */
            REAL bl[3], SR[9], ba[3], ba2[3];
            FN(mat3_vec)(J + i * 9, pa + 3, bl);
            FN(mat3_mul)(St, J + i * 9, SR);
            FN(mat3_vec)(SR, pa + 3, ba);
            FN(mat3_vec)(J + i * 9, pa, ba2);
            for (int k = 0; k < 3; ++k) { pA[par * 6 + 3 + k] += bl[k]; pA[par * 6 + k] += ba[k] + ba2[k]; }
        }
    }
    for (int k = 0; k < 6; ++k) acc[k] = 0;
    acc[5] = include_gravity ? (REAL)9.81 : 0; /* base lin acc (0,0,9.81)  rm.py:527-533 */
    for (int i = 1; i < L; ++i) { /* rm.py:611-629 */
        const int par = sp->parent[i], d = sp->dof[i];
        REAL t[3];
        FN(joint_trans)(sp, i, (d >= 0) ? q[d] : 0, t);
        REAL JT[9], it[3], S3[9], SR[9], na[3], nl[3], nl2[3];
        FN(mat3_transpose)(J + i * 9, JT);
        FN(mat3_vec)(JT, t, it);
        for (int k = 0; k < 3; ++k) it[k] = -it[k];
        FN(skew)(it, S3);
        FN(mat3_mul)(S3, JT, SR);
        FN(mat3_vec)(JT, acc + par * 6, na);
        FN(mat3_vec)(SR, acc + par * 6, nl);
        FN(mat3_vec)(JT, acc + par * 6 + 3, nl2);
        for (int k = 0; k < 3; ++k) {
            acc[i * 6 + k] = na[k] + cc[i * 6 + k];
            acc[i * 6 + 3 + k] = (nl[k] + nl2[k]) + cc[i * 6 + 3 + k];
        }
        if (d >= 0) {
            REAL Ua = 0, Ul = 0;
            for (int k = 0; k < 3; ++k) { Ua += U[i * 6 + k] * acc[i * 6 + k]; Ul += U[i * 6 + 3 + k] * acc[i * 6 + 3 + k]; }
            qdd[d] = ((REAL)1 / dd[i]) * (uu[i] - (Ua + Ul));
            for (int k = 0; k < 6; ++k) acc[i * 6 + k] += Sv[i * 6 + k] * qdd[d];
        }
    }
}

static int FN(oracle_forward_dynamics)(const drm_oracle_spec *sp, const IO_T *q, const IO_T *qd, const IO_T *f, int64_t B,
                                       int include_gravity, int use_damping, IO_T *qdd) {
    const int L = sp->n_links, n = sp->n_dofs;
    int err = 0;
#pragma omp parallel
    {
        REAL *buf = (REAL *)malloc(sizeof(REAL) * (size_t)(95 * L + 4 * n + 8));
        if (!buf) {
#pragma omp atomic write
            err = 1;
        } else {
            REAL *qq = buf + 95 * L, *qv = qq + n, *ff = qv + n, *out = ff + n;
#pragma omp for schedule(static)
            for (int64_t b = 0; b < B; ++b) {
                for (int d = 0; d < n; ++d) {
                    qq[d] = (REAL)q[b * n + d]; qv[d] = (REAL)qd[b * n + d]; ff[d] = (REAL)f[b * n + d]; out[d] = 0;
                }
                FN(aba_sample)(sp, qq, qv, ff, include_gravity, use_damping, out, buf);
                for (int d = 0; d < n; ++d) qdd[b * n + d] = (IO_T)out[d];
            }
            free(buf);
        }
    }
    return err;
}

#undef FN
#undef CAT
#undef CAT_
