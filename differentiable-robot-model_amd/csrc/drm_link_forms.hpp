// drm_link_forms.hpp — ABI 13 (include/drm_hip.h drm_walk_table_links): the forms a learnable link's pieces are stored in, shared by
// the device kernels (drm_host.hip) and the host build (drm_cpu.cpp).  Kept out of drm_sample.hpp on purpose: that header is part of
// the source key of every robot's own kernels (specialize._HEADERS), which do not read link parameters.
#pragma once
#include "drm_sample.hpp"

namespace drm {
// ---------------------------------------------------------------------------
// ABI 13: the FORM a learnable piece is stored in — the reference's parameter modules (rigid_body_params.py) evaluated where the
// table is built instead of by a chain of tiny torch kernels in front of it (and differentiated behind it):
//   scalars (mass, damping)   SQUARE_PLUS  value = l * l + c                              PositiveScalar, rigid_body_params.py:26-43
//   inertia_mat from l[6] = (three diagonal entries, then the strictly lower entries (1,0), (2,0), (2,1)):
//                             SYMM         the symmetric matrix with those entries        Symm3DInertiaMatrixNet, :387-404
//                             SPD          L L^T + c E, L lower triangular from l         SymmPosDef3DInertiaMatrixNet, :342-384
//                             COV          tr(S) E - S with S = L L^T + c E               CovParameterized3DInertiaMatrixNet, :252-339
// ---------------------------------------------------------------------------
DRM_HD float form_scalar(int form, float c, float raw) { return form == DRM_FORM_SQUARE_PLUS ? raw * raw + c : raw; }
DRM_HD float form_scalar_grad(int form, float raw, float g) { return form == DRM_FORM_SQUARE_PLUS ? 2.0f * (g * raw) : g; }
DRM_HD void form_lower(const float *l, float *L) {
    L[0] = l[0]; L[1] = 0.0f; L[2] = 0.0f;
    L[3] = l[3]; L[4] = l[1]; L[5] = 0.0f;
    L[6] = l[4]; L[7] = l[5]; L[8] = l[2];
}
// raw: 9 floats (PLAIN) or 6 (the other forms) -> I[9].  Branch-free: every form is evaluated and selected per element, so that every
// index is a compile-time constant and the arrays stay in registers (a first version with one branch per form spilled to scratch).
DRM_HD void form_inertia(int form, float c, const float *raw, float *I) {
    float L[9], S[9];
    form_lower(raw, L);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            S[r * 3 + k] = L[r * 3 + 0] * L[k * 3 + 0] + L[r * 3 + 1] * L[k * 3 + 1] + L[r * 3 + 2] * L[k * 3 + 2] + (r == k ? c : 0.0f);
    const float tr = S[0] + S[4] + S[8];
    const float symm[9] = {raw[0], raw[3], raw[4], raw[3], raw[1], raw[5], raw[4], raw[5], raw[2]};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = r * 3 + k;
            const float cov = (r == k ? tr : 0.0f) - S[i];
            I[i] = form == DRM_FORM_SYMM ? symm[i] : form == DRM_FORM_SPD ? S[i] : form == DRM_FORM_COV ? cov : raw[i];
        }
}
// gI[9] = d loss / d inertia_mat -> graw[9] = d loss / d raw (the entries past the form's raw size are zero)
DRM_HD void form_inertia_grad(int form, const float *raw, const float *gI, float *graw) {
    float L[9], G[9], M[9];
    form_lower(raw, L);
    const float tr = gI[0] + gI[4] + gI[8];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) G[r * 3 + k] = form == DRM_FORM_COV ? (r == k ? tr : 0.0f) - gI[r * 3 + k] : gI[r * 3 + k];
    // d / d L of <G, L L^T> = (G + G^T) L
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            M[r * 3 + k] = (G[r * 3 + 0] + G[0 * 3 + r]) * L[0 * 3 + k] + (G[r * 3 + 1] + G[1 * 3 + r]) * L[1 * 3 + k] +
                           (G[r * 3 + 2] + G[2 * 3 + r]) * L[2 * 3 + k];
    const float chol[9] = {M[0], M[4], M[8], M[3], M[6], M[7], 0.0f, 0.0f, 0.0f};
    const float symm[9] = {gI[0], gI[4], gI[8], gI[1] + gI[3], gI[2] + gI[6], gI[5] + gI[7], 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 9; ++i)
        graw[i] = form == DRM_FORM_SYMM ? symm[i] : (form == DRM_FORM_SPD || form == DRM_FORM_COV) ? chol[i] : gI[i];
}
// where element k of the 20 floats of a link lies: piece (rot_angles, trans, mass, com, inertia_mat, damping) and offset inside it
DRM_HD int link_piece_of(int k) { return k < 3 ? 0 : k < 6 ? 1 : k < 7 ? 2 : k < 10 ? 3 : k < 19 ? 4 : 5; }
DRM_HD int link_piece_offset(int k) { return k < 3 ? k : k < 6 ? k - 3 : k < 7 ? 0 : k < 10 ? k - 7 : k < 19 ? k - 10 : 0; }
// raw[20] (what lies at the pieces' addresses, zero-padded) -> p[20] (the URDF-level parameters link_row takes)
DRM_HD void link_forms_apply(const int32_t *form, const float *c, const float *raw, float *p) {
#pragma unroll
    for (int k = 0; k < LINK_PARAM_FLOATS; ++k) p[k] = raw[k];
    p[6] = form_scalar(form[0], c[0], raw[6]);
    form_inertia(form[1], c[1], raw + 10, p + 10);
    p[19] = form_scalar(form[2], c[2], raw[19]);
}
// gp[20] = d loss / d p  ->  in place d loss / d raw
DRM_HD void link_forms_grad(const int32_t *form, const float *raw, float *gp) {
    float gI[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gI[i] = gp[10 + i];
    gp[6] = form_scalar_grad(form[0], raw[6], gp[6]);
    form_inertia_grad(form[1], raw + 10, gI, gp + 10);
    gp[19] = form_scalar_grad(form[2], raw[19], gp[19]);
}

} // namespace drm
