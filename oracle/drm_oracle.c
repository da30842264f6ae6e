/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  See drm_oracle_impl.h for the restatement
 * and its reference citations.  Build: `make -C oracle` (gcc, -ffp-contract=off
 * so that no FMA contraction changes the reference's fp32 operation order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "drm_oracle.h"

#define REAL float
#define IO_T float
#define SUFFIX _f32
#define COS cosf
#define SIN sinf
#define FABS fabsf
#include "drm_oracle_impl.h"
#undef REAL
#undef IO_T
#undef SUFFIX
#undef COS
#undef SIN
#undef FABS

#define REAL double
#define IO_T double
#define SUFFIX _f64
#define COS cos
#define SIN sin
#define FABS fabs
#include "drm_oracle_impl.h"
#undef REAL
#undef IO_T
#undef SUFFIX
#undef COS
#undef SIN
#undef FABS

int drm_oracle_fk_f32(const drm_oracle_spec *s, const float *q, int64_t B, const int *t, int T, float *pos, float *quat) { return oracle_fk_f32(s, q, B, t, T, pos, quat); }
int drm_oracle_fk_all_poses_f32(const drm_oracle_spec *s, const float *q, int64_t B, float *Rw, float *pw) { return oracle_fk_all_poses_f32(s, q, B, Rw, pw); }
int drm_oracle_fk_jacobian_f32(const drm_oracle_spec *s, const float *q, int64_t B, int link, float *pos, float *quat, float *lj, float *aj) { return oracle_fk_jacobian_f32(s, q, B, link, pos, quat, lj, aj); }
int drm_oracle_rnea_f32(const drm_oracle_spec *s, const float *q, const float *qd, const float *qdd, int64_t B, int g, int d, float *tau) { return oracle_rnea_f32(s, q, qd, qdd, B, g, d, tau); }
int drm_oracle_fk_f64(const drm_oracle_spec *s, const double *q, int64_t B, const int *t, int T, double *pos, double *quat) { return oracle_fk_f64(s, q, B, t, T, pos, quat); }
int drm_oracle_fk_all_poses_f64(const drm_oracle_spec *s, const double *q, int64_t B, double *Rw, double *pw) { return oracle_fk_all_poses_f64(s, q, B, Rw, pw); }
int drm_oracle_fk_jacobian_f64(const drm_oracle_spec *s, const double *q, int64_t B, int link, double *pos, double *quat, double *lj, double *aj) { return oracle_fk_jacobian_f64(s, q, B, link, pos, quat, lj, aj); }
int drm_oracle_rnea_f64(const drm_oracle_spec *s, const double *q, const double *qd, const double *qdd, int64_t B, int g, int d, double *tau) { return oracle_rnea_f64(s, q, qd, qdd, B, g, d, tau); }

int drm_oracle_mass_matrix_f32(const drm_oracle_spec *s, const float *q, int64_t B, int g, int d, float *H) { return oracle_mass_matrix_f32(s, q, B, g, d, H); }
int drm_oracle_mass_matrix_f64(const drm_oracle_spec *s, const double *q, int64_t B, int g, int d, double *H) { return oracle_mass_matrix_f64(s, q, B, g, d, H); }

int drm_oracle_forward_dynamics_f32(const drm_oracle_spec *s, const float *q, const float *qd, const float *f, int64_t B, int g, int d, float *qdd) { return oracle_forward_dynamics_f32(s, q, qd, f, B, g, d, qdd); }
int drm_oracle_forward_dynamics_f64(const drm_oracle_spec *s, const double *q, const double *qd, const double *f, int64_t B, int g, int d, double *qdd) { return oracle_forward_dynamics_f64(s, q, qd, f, B, g, d, qdd); }

int drm_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void drm_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
