// drm_host.hip — host-side pieces of the C ABI shared by all kernels: error string, walk checks, launch geometry.
#include <string.h>

#include "drm_common.hpp"
#include "drm_sample.hpp"
#include "drm_link_forms.hpp"

namespace drm {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *a, long b, long c) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

int check_walk(const drm_walk *w) {
    if (!w) return fail(DRM_ERR_INVALID, "walk is NULL");
    if (!w->ops_f || !w->ops_i) return fail(DRM_ERR_INVALID, "walk tables are NULL");
    const int c = w->capacity;
    if (c < 4 || (c & 3) || c > 0xffff)
        return fail(DRM_ERR_INVALID, "walk capacity %s%ld is not a multiple of 4 in [4, 65535]", "", c);
    if (w->n_ops < 0 || w->n_ops > c)
        return fail(DRM_ERR_INVALID, "walk has %s%ld ops but capacity %ld", "", w->n_ops, c);
    if (w->n_dofs < 1 || w->n_dofs > DRM_MAX_DOFS)
        return fail(DRM_ERR_UNSUPPORTED, "n_dofs %s%ld outside [1, %ld]", "", w->n_dofs, DRM_MAX_DOFS);
    if (w->n_slots < 0 || w->n_slots > DRM_MAX_SLOTS)
        return fail(DRM_ERR_UNSUPPORTED, "walk needs %s%ld save slots, kernels have %ld", "", w->n_slots, DRM_MAX_SLOTS);
    return DRM_OK;
}

int launched() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return DRM_OK;
}

int make_geometry(int64_t B, int lds_floats_per_wave, Geometry &g) {
    const size_t per_wave = (size_t)round4(lds_floats_per_wave) * sizeof(float);
    if (per_wave > (size_t)MAX_LDS_BYTES)
        return fail(DRM_ERR_UNSUPPORTED, "tile needs %s%ld bytes of LDS per wave (max %ld)", "", (long)per_wave,
                    (long)MAX_LDS_BYTES);
    int wpb = MAX_WAVES_PER_BLOCK;
    while (wpb > 1 && per_wave * wpb > (size_t)64 * 1024) wpb >>= 1;
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    const int64_t blocks = (tiles + wpb - 1) / wpb;
    if (blocks > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    g.grid = dim3((unsigned)blocks);
    g.block = dim3(WAVE * wpb);
    g.lds_per_wave = round4(lds_floats_per_wave);
    g.lds_bytes = per_wave * wpb;
    return DRM_OK;
}

const char *last_error() { return g_err; }

} // namespace drm

namespace drm {
__global__ void link_rows_kernel(const float *__restrict__ params, int n_links, float *__restrict__ rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_links) return;
    float p[LINK_PARAM_FLOATS], row[DRM_OPF_STRIDE];
#pragma unroll
    for (int k = 0; k < LINK_PARAM_FLOATS; ++k) p[k] = params[i * LINK_PARAM_FLOATS + k];
    link_row(p, row);
#pragma unroll
    for (int k = 0; k < DRM_OPF_STRIDE; ++k) rows[i * DRM_OPF_STRIDE + k] = row[k];
}
__global__ void link_rows_backward_kernel(const float *__restrict__ params, const float *__restrict__ grad_rows, int n_links,
                                          float *__restrict__ grad_params) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_links) return;
    float p[LINK_PARAM_FLOATS], g[DRM_OPF_STRIDE], gp[LINK_PARAM_FLOATS];
#pragma unroll
    for (int k = 0; k < LINK_PARAM_FLOATS; ++k) p[k] = params[i * LINK_PARAM_FLOATS + k];
#pragma unroll
    for (int k = 0; k < DRM_OPF_STRIDE; ++k) g[k] = grad_rows[i * DRM_OPF_STRIDE + k];
    link_row_backward(p, g, gp);
#pragma unroll
    for (int k = 0; k < LINK_PARAM_FLOATS; ++k) grad_params[i * LINK_PARAM_FLOATS + k] = gp[k];
}
// The walk table of a robot with learnable links in ONE launch (and its derivative in another): the rows of the learnable
// links from their URDF-level parameters, gathered into walk order (with the exact +-1 factors of the axis
// canonicalisation) over the table of the constant links.  One block; everything is < 8 KB.
constexpr int WALK_TABLE_THREADS = 256, WALK_TABLE_MAX_LINKS = 32;
// ABI 13: where the pieces of the learnable links lie and the forms they are stored in — passed by value (2.3 KB of kernel arguments)
struct LinkSources {
    const float *at[WALK_TABLE_MAX_LINKS][6];      // rot_angles, trans, mass, com, inertia_mat, damping
    int32_t form[WALK_TABLE_MAX_LINKS][3];         // of mass, inertia_mat, damping
    float c[WALK_TABLE_MAX_LINKS][3];
};
// raw[l * 20 + k]: what lies at the addresses, in drm_link_rows' layout (an inertia matrix in a six-number form: zeros behind them).
// One element per thread and pass; the caller's next barrier publishes it.
__device__ __forceinline__ void link_sources_load(const LinkSources &src, int n_links, int t, float *raw) {
    for (int i = t; i < n_links * LINK_PARAM_FLOATS; i += WALK_TABLE_THREADS) {
        const int l = i / LINK_PARAM_FLOATS, k = i - l * LINK_PARAM_FLOATS;
        const int piece = link_piece_of(k), off = link_piece_offset(k);
        const bool there = piece != 4 || src.form[l][1] == DRM_FORM_PLAIN || off < 6;
        raw[i] = there ? src.at[l][piece][off] : 0.0f;
    }
}
__global__ void __launch_bounds__(WALK_TABLE_THREADS)
    walk_table_links_kernel(LinkSources src, int n_links, const float *__restrict__ base, const int32_t *__restrict__ sel,
                            const float *__restrict__ gsign, int n_entries, float *__restrict__ ops_f) {
    __shared__ float rows[WALK_TABLE_MAX_LINKS * DRM_OPF_STRIDE], raw[WALK_TABLE_MAX_LINKS * LINK_PARAM_FLOATS];
    const int t = (int)threadIdx.x;
    // (requested before the parameters are waited for)
    constexpr int PER = DRM_MAX_OPS * DRM_OPF_STRIDE / WALK_TABLE_THREADS;
    int r_mine[PER];
    float s_mine[PER], b_mine[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = t + i * WALK_TABLE_THREADS;
        const bool in = e < n_entries;
        r_mine[i] = in ? sel[e] : -1;
        s_mine[i] = in ? gsign[e] : 0.0f;
        b_mine[i] = in ? base[e] : 0.0f;
    }
    link_sources_load(src, n_links, t, raw);
    __syncthreads();
    if (t < n_links) {
        float pr[LINK_PARAM_FLOATS], p[LINK_PARAM_FLOATS], row[DRM_OPF_STRIDE];
#pragma unroll
        for (int k = 0; k < LINK_PARAM_FLOATS; ++k) pr[k] = raw[t * LINK_PARAM_FLOATS + k];
        link_forms_apply(src.form[t], src.c[t], pr, p);
        link_row(p, row);
#pragma unroll
        for (int k = 0; k < DRM_OPF_STRIDE; ++k) rows[t * DRM_OPF_STRIDE + k] = row[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = t + i * WALK_TABLE_THREADS;
        if (e < n_entries) ops_f[e] = r_mine[i] >= 0 ? rows[r_mine[i]] * s_mine[i] : b_mine[i];
    }
}
__global__ void __launch_bounds__(WALK_TABLE_THREADS)
    walk_table_kernel(const float *__restrict__ params, int n_links, const float *__restrict__ base,
                      const int32_t *__restrict__ sel, const float *__restrict__ gsign, int n_entries,
                      float *__restrict__ ops_f) {
    __shared__ float rows[WALK_TABLE_MAX_LINKS * DRM_OPF_STRIDE];
    const int t = (int)threadIdx.x;
    if (t < n_links) {
        float p[LINK_PARAM_FLOATS], row[DRM_OPF_STRIDE];
#pragma unroll
        for (int k = 0; k < LINK_PARAM_FLOATS; ++k) p[k] = params[t * LINK_PARAM_FLOATS + k];
        link_row(p, row);
#pragma unroll
        for (int k = 0; k < DRM_OPF_STRIDE; ++k) rows[t * DRM_OPF_STRIDE + k] = row[k];
    }
    __syncthreads();
    for (int e = t; e < n_entries; e += WALK_TABLE_THREADS) {
        const int r = sel[e];
        ops_f[e] = r >= 0 ? rows[r] * gsign[e] : base[e];
    }
}
// Latency-bound (one block, a few KB): the last wavefront only turns every link's rot_angles into d F / d (roll, pitch, yaw)
// (rpy_jacobian: ~1.1 us of dependent arithmetic) while the others sort the walk entries to the row elements they were gathered from.
// A link is one op of a walk, so nearly always ONE entry per element: every entry names itself at its element (LDS atomicMin: the
// lowest index) and is counted; an element with one entry takes it, one with several adds them in entry order (deterministic, the
// host build's order) — instead of every element scanning the ops of the walk (round 4: 14.5 -> 6 us; now ~3).
// LINKS (ABI 13, drm_walk_table_links_backward): the parameters are read where the links' tensors lie, in their forms (LinkSources),
// and the gradient goes back to the raw parameters.
template <bool LINKS>
__global__ void __launch_bounds__(WALK_TABLE_THREADS)
    walk_table_backward_kernel(const float *__restrict__ params, int n_links, const float *__restrict__ grad_ops_f,
                               const int32_t *__restrict__ sel, const float *__restrict__ gsign, int n_entries,
                               float *__restrict__ grad_params, LinkSources src) {
    constexpr int NE = DRM_MAX_OPS * DRM_OPF_STRIDE, NR = WALK_TABLE_MAX_LINKS * DRM_OPF_STRIDE, WAVES = WALK_TABLE_THREADS / WAVE;
    __shared__ float grows[NR], ge[NE], dmat[WALK_TABLE_MAX_LINKS][27];
    __shared__ float raw[LINKS ? WALK_TABLE_MAX_LINKS * LINK_PARAM_FLOATS : 1];
    __shared__ int se[NE];
    __shared__ unsigned owner[NR], count[NR];
    const int t = (int)threadIdx.x;
    if constexpr (LINKS) {
        link_sources_load(src, n_links, t, raw);
        params = raw;      // (the three angles are plain in every form: the trigonometry wavefront reads them from here)
    }
    const bool trig_wave = t >= (WAVES - 1) * WAVE;
    const int n_rows = n_links * DRM_OPF_STRIDE;
    // (requested first: the entries this thread sorts below)
    constexpr int PER = NE / ((WAVES - 1) * WAVE) + 1;
    float g_mine[PER];
    int r_mine[PER];
    if (!trig_wave) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = t + i * (WAVES - 1) * WAVE;
            const bool in = e < n_entries;
            r_mine[i] = in ? sel[e] : -1;
            g_mine[i] = in ? grad_ops_f[e] * gsign[e] : 0.0f;
        }
    }
    for (int r = t; r < n_rows; r += WALK_TABLE_THREADS) { owner[r] = 0xffffffffu; count[r] = 0u; }
    __syncthreads();
    if (trig_wave) {
        const int l = t - (WAVES - 1) * WAVE;
        if (l < n_links) {
            float rpy[3], D[27];
#pragma unroll
            for (int i = 0; i < 3; ++i) rpy[i] = params[l * LINK_PARAM_FLOATS + i];
            rpy_jacobian(rpy, D, D + 9, D + 18);
#pragma unroll
            for (int i = 0; i < 27; ++i) dmat[l][i] = D[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = t + i * (WAVES - 1) * WAVE;
            if (e < n_entries) {
                ge[e] = g_mine[i];
                se[e] = r_mine[i];
                if (r_mine[i] >= 0 && r_mine[i] < n_rows) {
                    atomicMin(&owner[r_mine[i]], (unsigned)e);
                    atomicAdd(&count[r_mine[i]], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int r = t; r < n_rows; r += WALK_TABLE_THREADS) {
        const unsigned n = count[r];
        float s = 0.0f;
        if (n == 1u) {
            s += ge[owner[r]];
        } else if (n > 1u) {
            for (int e = 0; e < n_entries; ++e) s += se[e] == r ? ge[e] : 0.0f;
        }
        grows[r] = s;
    }
    __syncthreads();
    if (t < n_links) {
        float p[LINK_PARAM_FLOATS], g[DRM_OPF_STRIDE], gp[LINK_PARAM_FLOATS], pr[LINKS ? LINK_PARAM_FLOATS : 1];
        if constexpr (LINKS) {
#pragma unroll
            for (int k = 0; k < LINK_PARAM_FLOATS; ++k) pr[k] = raw[t * LINK_PARAM_FLOATS + k];
            link_forms_apply(src.form[t], src.c[t], pr, p);
        } else {
#pragma unroll
            for (int k = 0; k < LINK_PARAM_FLOATS; ++k) p[k] = params[t * LINK_PARAM_FLOATS + k];
        }
#pragma unroll
        for (int k = 0; k < DRM_OPF_STRIDE; ++k) g[k] = grows[t * DRM_OPF_STRIDE + k];
#pragma unroll
        for (int a = 0; a < 3; ++a) gp[a] = dot9(&dmat[t][a * 9], g);
        link_row_backward_rest(p, g, gp);
        if constexpr (LINKS) link_forms_grad(src.form[t], pr, gp);
#pragma unroll
        for (int k = 0; k < LINK_PARAM_FLOATS; ++k) grad_params[t * LINK_PARAM_FLOATS + k] = gp[k];
    }
}
} // namespace drm

extern "C" {
int drm_walk_table(const float *params, int32_t n_links, const float *base, const int32_t *sel, const float *gsign,
                   int32_t n_entries, float *ops_f, void *stream) {
    if (!params || !base || !sel || !gsign || !ops_f) return drm::fail(DRM_ERR_INVALID, "drm_walk_table: NULL argument");
    if (n_links < 1 || n_links > drm::WALK_TABLE_MAX_LINKS || n_entries < 1 || n_entries > DRM_MAX_OPS * DRM_OPF_STRIDE)
        return drm::fail(DRM_ERR_INVALID, "drm_walk_table: 1..32 learnable links and at most 32 x 32 walk entries");
    hipLaunchKernelGGL(drm::walk_table_kernel, dim3(1), dim3(drm::WALK_TABLE_THREADS), 0, (hipStream_t)stream, params,
                       (int)n_links, base, sel, gsign, (int)n_entries, ops_f);
    return drm::launched();
}
int drm_walk_table_backward(const float *params, int32_t n_links, const float *grad_ops_f, const int32_t *sel,
                            const float *gsign, int32_t n_entries, float *grad_params, void *stream) {
    if (!params || !grad_ops_f || !sel || !gsign || !grad_params)
        return drm::fail(DRM_ERR_INVALID, "drm_walk_table_backward: NULL argument");
    if (n_links < 1 || n_links > drm::WALK_TABLE_MAX_LINKS || n_entries < 1 || n_entries > DRM_MAX_OPS * DRM_OPF_STRIDE)
        return drm::fail(DRM_ERR_INVALID, "drm_walk_table_backward: 1..32 learnable links and at most 32 x 32 walk entries");
    hipLaunchKernelGGL(drm::walk_table_backward_kernel<false>, dim3(1), dim3(drm::WALK_TABLE_THREADS), 0, (hipStream_t)stream,
                       params, (int)n_links, grad_ops_f, sel, gsign, (int)n_entries, grad_params, drm::LinkSources{});
    return drm::launched();
}
static int link_sources(const char *who, const drm_link_pieces *links, const drm_link_forms *forms, int32_t n_links, int32_t n_entries,
                        drm::LinkSources &src) {
    if (!links) return drm::fail(DRM_ERR_INVALID, "%s: links must not be NULL", who);
    if (n_links < 1 || n_links > drm::WALK_TABLE_MAX_LINKS || n_entries < 1 || n_entries > DRM_MAX_OPS * DRM_OPF_STRIDE)
        return drm::fail(DRM_ERR_INVALID, "%s: 1..32 learnable links and at most 32 x 32 walk entries", who);
    for (int l = 0; l < n_links; ++l) {
        const float *at[6] = {links[l].rot_angles, links[l].trans, links[l].mass, links[l].com, links[l].inertia_mat, links[l].damping};
        for (int j = 0; j < 6; ++j) {
            if (!at[j]) return drm::fail(DRM_ERR_INVALID, "%s: a piece of a link is NULL", who);
            src.at[l][j] = at[j];
        }
        const drm_link_forms f = forms ? forms[l] : drm_link_forms{};
        const bool ok = (f.mass == DRM_FORM_PLAIN || f.mass == DRM_FORM_SQUARE_PLUS) &&
                        (f.damping == DRM_FORM_PLAIN || f.damping == DRM_FORM_SQUARE_PLUS) &&
                        (f.inertia_mat == DRM_FORM_PLAIN || (f.inertia_mat >= DRM_FORM_SYMM && f.inertia_mat <= DRM_FORM_COV));
        if (!ok) return drm::fail(DRM_ERR_INVALID, "%s: unknown form of a piece", who);
        src.form[l][0] = f.mass; src.form[l][1] = f.inertia_mat; src.form[l][2] = f.damping;
        src.c[l][0] = f.mass_c; src.c[l][1] = f.inertia_mat_c; src.c[l][2] = f.damping_c;
    }
    return DRM_OK;
}
int drm_walk_table_links(const drm_link_pieces *links, const drm_link_forms *forms, int32_t n_links, const float *base, const int32_t *sel,
                         const float *gsign, int32_t n_entries, float *ops_f, void *stream) {
    if (!base || !sel || !gsign || !ops_f) return drm::fail(DRM_ERR_INVALID, "drm_walk_table_links: NULL argument");
    drm::LinkSources src{};
    if (int rc = link_sources("drm_walk_table_links", links, forms, n_links, n_entries, src)) return rc;
    hipLaunchKernelGGL(drm::walk_table_links_kernel, dim3(1), dim3(drm::WALK_TABLE_THREADS), 0, (hipStream_t)stream, src, (int)n_links,
                       base, sel, gsign, (int)n_entries, ops_f);
    return drm::launched();
}
int drm_walk_table_links_backward(const drm_link_pieces *links, const drm_link_forms *forms, int32_t n_links, const float *grad_ops_f,
                                  const int32_t *sel, const float *gsign, int32_t n_entries, float *grad_params, void *stream) {
    if (!grad_ops_f || !sel || !gsign || !grad_params) return drm::fail(DRM_ERR_INVALID, "drm_walk_table_links_backward: NULL argument");
    drm::LinkSources src{};
    if (int rc = link_sources("drm_walk_table_links_backward", links, forms, n_links, n_entries, src)) return rc;
    hipLaunchKernelGGL(drm::walk_table_backward_kernel<true>, dim3(1), dim3(drm::WALK_TABLE_THREADS), 0, (hipStream_t)stream,
                       (const float *)nullptr, (int)n_links, grad_ops_f, sel, gsign, (int)n_entries, grad_params, src);
    return drm::launched();
}
int drm_link_rows(const float *params, int32_t n_links, float *rows, void *stream) {
    if (!params || !rows || n_links < 0) return drm::fail(DRM_ERR_INVALID, "params / rows must not be NULL, n_links >= 0");
    if (n_links == 0) return DRM_OK;
    hipLaunchKernelGGL(drm::link_rows_kernel, dim3((unsigned)((n_links + 63) / 64)), dim3(64), 0, (hipStream_t)stream, params,
                       (int)n_links, rows);
    return drm::launched();
}
int drm_link_rows_backward(const float *params, const float *grad_rows, int32_t n_links, float *grad_params, void *stream) {
    if (!params || !grad_rows || !grad_params || n_links < 0)
        return drm::fail(DRM_ERR_INVALID, "params / grad_rows / grad_params must not be NULL, n_links >= 0");
    if (n_links == 0) return DRM_OK;
    hipLaunchKernelGGL(drm::link_rows_backward_kernel, dim3((unsigned)((n_links + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       params, grad_rows, (int)n_links, grad_params);
    return drm::launched();
}
int drm_special_load(const char *code_object_path, const char *kernel_name, const void **function_out) {
    if (!code_object_path || !kernel_name || !function_out) return drm::fail(DRM_ERR_INVALID, "drm_special_load: NULL argument");
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    hipError_t e = hipModuleLoad(&mod, code_object_path);
    if (e != hipSuccess) return drm::fail(DRM_ERR_LAUNCH, "hipModuleLoad(%s) failed", code_object_path);
    e = hipModuleGetFunction(&fn, mod, kernel_name);
    if (e != hipSuccess) return drm::fail(DRM_ERR_LAUNCH, "hipModuleGetFunction(%s) failed", kernel_name);
    *function_out = (const void *)fn;
    return DRM_OK;
}
int drm_abi_version(void) { return DRM_ABI_VERSION; }
int drm_walk_sizeof(void) { return (int)sizeof(drm_walk); }
const char *drm_last_error(void) { return drm::last_error(); }
}
