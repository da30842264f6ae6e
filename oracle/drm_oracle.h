/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see drm_oracle_impl.h).
 * C interface of the CPU restatement of the reference FK / Jacobian / RNEA path.
 * The robot is described per link in URDF <link> order exactly as the
 * reference's URDFRobotModel.get_body_parameters_from_urdf returns it
 * (reference differentiable_robot_model/urdf_utils.py:28-126).
 */
#ifndef DRM_ORACLE_H
#define DRM_ORACLE_H
#include <stdint.h>

typedef struct drm_oracle_spec {
    int32_t n_links;        /* L, link 0 = root */
    int32_t n_dofs;         /* n */
    const int32_t *parent;  /* [L]  parent link index, -1 for the root           */
    const int32_t *dof;     /* [L]  DoF column of the link's joint, -1 = fixed   */
    const float *rpy;       /* [L,3] joint origin rpy  (rot_angles)              */
    const float *trans;     /* [L,3] joint origin xyz  (trans)                   */
    const float *axis;      /* [L,3] joint axis (zeros for fixed joints)         */
    const float *damping;   /* [L]   joint damping (0 for root / fixed)          */
    const float *mass;      /* [L]                                               */
    const float *com;       /* [L,3] inertial origin xyz                         */
    const float *inertia;   /* [L,9] inertia about the com, row-major            */
    const int32_t *kind;    /* [L] or NULL: 0 fixed, 1 revolute / continuous, 2 prismatic.  NULL = the reference's
                               model (every moving joint revolute, robot_model.py:122-126).  EXTENSION beyond the
                               reference, see drm_oracle_impl.h "joint models"                 */
} drm_oracle_spec;

#ifdef __cplusplus
extern "C" {
#endif
/* fp32 arithmetic (what the reference does) */
int drm_oracle_fk_f32(const drm_oracle_spec *, const float *q, int64_t B, const int *targets, int T, float *pos, float *quat);
int drm_oracle_fk_all_poses_f32(const drm_oracle_spec *, const float *q, int64_t B, float *Rw, float *pw);
int drm_oracle_fk_jacobian_f32(const drm_oracle_spec *, const float *q, int64_t B, int link, float *pos, float *quat, float *lin_jac, float *ang_jac);
int drm_oracle_rnea_f32(const drm_oracle_spec *, const float *q, const float *qd, const float *qdd, int64_t B, int include_gravity, int use_damping, float *tau);
/* fp64 arithmetic on the same float32 constants (rounding-noise bound) */
int drm_oracle_fk_f64(const drm_oracle_spec *, const double *q, int64_t B, const int *targets, int T, double *pos, double *quat);
int drm_oracle_fk_all_poses_f64(const drm_oracle_spec *, const double *q, int64_t B, double *Rw, double *pw);
int drm_oracle_fk_jacobian_f64(const drm_oracle_spec *, const double *q, int64_t B, int link, double *pos, double *quat, double *lin_jac, double *ang_jac);
int drm_oracle_rnea_f64(const drm_oracle_spec *, const double *q, const double *qd, const double *qdd, int64_t B, int include_gravity, int use_damping, double *tau);
/* H [B, n, n] = compute_lagrangian_inertia_matrix (rm.py:402-450) */
int drm_oracle_mass_matrix_f32(const drm_oracle_spec *, const float *q, int64_t B, int include_gravity, int use_damping, float *H);
int drm_oracle_mass_matrix_f64(const drm_oracle_spec *, const double *q, int64_t B, int include_gravity, int use_damping, double *H);
/* qdd [B, n] = compute_forward_dynamics (articulated-body algorithm, rm.py:487-624) */
int drm_oracle_forward_dynamics_f32(const drm_oracle_spec *, const float *q, const float *qd, const float *f, int64_t B, int include_gravity, int use_damping, float *qdd);
int drm_oracle_forward_dynamics_f64(const drm_oracle_spec *, const double *q, const double *qd, const double *f, int64_t B, int include_gravity, int use_damping, double *qdd);
int drm_oracle_max_threads(void);
void drm_oracle_set_threads(int n);
#ifdef __cplusplus
}
#endif
#endif
