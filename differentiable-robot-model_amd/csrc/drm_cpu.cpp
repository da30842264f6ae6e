// drm_cpu.cpp — the HOST build of the C ABI (include/drm_hip.h) → libdrm_cpu.so: what a model constructed with device="cpu"
// (the reference's default, robot_model.py:100-104; its whole test suite builds models there,
// tests/test_kinematics_dynamics.py:133-137) computes with.  The per-sample arithmetic is the kernels' own (drm_sample.hpp /
// drm_tree.hpp through the loops of drm_host_loops.hpp) compiled by g++; rows are processed in CHUNK-row pieces that a small
// pool of threads picks up.  Same entry points, same argument meaning and error codes as libdrm_hip.so, with HOST pointers;
// `stream` is ignored (calls return when the results are written).
//
// NOT a fallback of the HIP path: the Python binding picks the library by the DEVICE OF THE MODEL (backend.library_for) —
// tensors on a HIP device only ever reach libdrm_hip.so, which must exist (backend.load_library raises otherwise).
//
// Batch sums of the backward entry points (grad_ops_f, the loss of drm_fk_mse): one partial per CHUNK rows, added up in chunk
// order in double — the result does not depend on the number of threads.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "drm_host_loops.hpp"
#include "drm_link_forms.hpp"

namespace {
using namespace drm_host;

constexpr int64_t CHUNK = 256; // rows per work item (and per partial sum of the backward calls)
thread_local char g_err[512] = "";
std::atomic<int> g_threads{0}; // 0: hardware_concurrency (or DRM_CPU_THREADS)

int fail(int code, const char *fmt, long a = 0, long b = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}

int check_walk(const drm_walk *w) {
    if (!w) return fail(DRM_ERR_INVALID, "walk is NULL");
    if (!w->ops_f || !w->ops_i) return fail(DRM_ERR_INVALID, "walk tables are NULL");
    const int c = w->capacity;
    if (c < 4 || (c & 3) || c > 0xffff) return fail(DRM_ERR_INVALID, "walk capacity %ld is not a multiple of 4 in [4, 65535]", c);
    if (w->n_ops < 0 || w->n_ops > c) return fail(DRM_ERR_INVALID, "walk has %ld ops but capacity %ld", w->n_ops, c);
    if (w->n_dofs < 1 || w->n_dofs > DRM_MAX_DOFS) return fail(DRM_ERR_UNSUPPORTED, "n_dofs %ld outside [1, %ld]", w->n_dofs, DRM_MAX_DOFS);
    if (w->n_slots < 0 || w->n_slots > DRM_MAX_SLOTS)
        return fail(DRM_ERR_UNSUPPORTED, "walk needs %ld save slots, the walks have %ld", w->n_slots, DRM_MAX_SLOTS);
    if (w->n_segments < 1 || w->n_segments > DRM_MAX_SEGMENTS) return fail(DRM_ERR_INVALID, "walk has %ld segments", w->n_segments);
    return DRM_OK;
}

int check_backward_walk(const drm_walk *w, uint64_t mask) {
    if (int rc = check_walk(w)) return rc;
    if (w->capacity > DRM_MAX_OPS) return fail(DRM_ERR_UNSUPPORTED, "backward walks take at most %ld ops (capacity %ld)", DRM_MAX_OPS, w->capacity);
    if (w->n_slots > DRM_MAX_SLOTS_BACKWARD) return fail(DRM_ERR_UNSUPPORTED, "backward walks take at most %ld save slots (%ld)", DRM_MAX_SLOTS_BACKWARD, w->n_slots);
    if (w->capacity < 64 && (mask >> w->capacity)) return fail(DRM_ERR_INVALID, "param_mask selects ops beyond the walk's capacity");
    return DRM_OK;
}

int64_t n_chunks(int64_t B) { return (B + CHUNK - 1) / CHUNK; }

int pool_size(int64_t chunks) {
    int t = g_threads.load();
    if (t <= 0) {
        const char *e = getenv("DRM_CPU_THREADS");
        t = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    }
    if (t < 1) t = 1;
    if (t > 64) t = 64;
    return (int64_t)t > chunks ? (int)chunks : t;
}

// body(chunk, first_row, rows) for every chunk of B rows, on up to pool_size() threads (the calling thread is one of them)
template <class Body>
void for_chunks(int64_t B, Body body) {
    const int64_t chunks = n_chunks(B);
    if (chunks <= 0) return;
    const int threads = pool_size(chunks);
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t c = next.fetch_add(1);
            if (c >= chunks) return;
            const int64_t b0 = c * CHUNK;
            body(c, b0, B - b0 < CHUNK ? B - b0 : CHUNK);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(threads - 1);
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
}

// grad_ops_f = the per-chunk partials added in chunk order (double accumulation)
void reduce_partials(const float *partials, int64_t chunks, int entries, float *out) {
    for (int e = 0; e < entries; ++e) {
        double s = 0.0;
        for (int64_t c = 0; c < chunks; ++c) s += partials[c * entries + e];
        out[e] = (float)s;
    }
}

int fk_any(const drm_walk *w, const float *q, int64_t B, int T, float *pos, float *quat, bool link_major) {
    if (int rc = check_walk(w)) return rc;
    if (B < 0 || T < 1) return fail(DRM_ERR_INVALID, "bad batch (%ld) or target count (%ld)", (long)B, T);
    if (!q || !pos || !quat) return fail(DRM_ERR_INVALID, "q / pos / quat must not be NULL");
    const int n = w->n_dofs;
    for_chunks(B, [&](int64_t, int64_t b0, int64_t rows) {
        if (link_major)
            fk_loop(w, q + b0 * n, rows, T, pos + b0 * 3, quat + b0 * 4, 1, B);
        else
            fk_loop(w, q + b0 * n, rows, T, pos + b0 * T * 3, quat + b0 * T * 4);
    });
    return DRM_OK;
}

int fk_fan(const drm_walk *chains, int T, const float *q, int64_t B, float *pos, float *quat, bool link_major) {
    if (!chains || T < 2 || T > 4) return fail(DRM_ERR_INVALID, "drm_fk_fanout takes 2 .. 4 chains (%ld)", T);
    for (int t = 0; t < T; ++t)
        if (int rc = check_walk(chains + t)) return rc;
    if (B < 0 || !q || !pos || !quat) return fail(DRM_ERR_INVALID, "bad batch or NULL q / pos / quat");
    const int n = chains[0].n_dofs;
    for_chunks(B, [&](int64_t, int64_t b0, int64_t rows) {
        for (int t = 0; t < T; ++t) {
            if (link_major)
                fk_loop(chains + t, q + b0 * n, rows, 1, pos + (t * B + b0) * 3, quat + (t * B + b0) * 4);
            else
                fk_loop(chains + t, q + b0 * n, rows, 1, pos + (b0 * T + t) * 3, quat + (b0 * T + t) * 4, T, 0);
        }
    });
    return DRM_OK;
}

int rnea_checked(const drm_walk *w, const float *q, const float *qd, const float *tau, int64_t B) {
    if (int rc = check_walk(w)) return rc;
    if (B < 0 || !q || !qd || !tau) return fail(DRM_ERR_INVALID, "bad batch or NULL q / qd / output");
    return DRM_OK;
}
} // namespace

extern "C" {
int drm_abi_version(void) { return DRM_ABI_VERSION; }
int drm_walk_sizeof(void) { return (int)sizeof(drm_walk); }
const char *drm_last_error(void) { return g_err; }
// libdrm_cpu.so only: the number of threads the calls may use (0 = all cores / DRM_CPU_THREADS)
void drm_cpu_set_threads(int n) { g_threads.store(n); }
int drm_special_load(const char *, const char *, const void **) {
    return fail(DRM_ERR_UNSUPPORTED, "per-robot code objects are HIP kernels: not available in the host build");
}

int drm_fk(const drm_walk *w, const float *q, int64_t B, int32_t T, float *pos, float *quat, void *) {
    return fk_any(w, q, B, T, pos, quat, false);
}
int drm_fk_links(const drm_walk *w, const float *q, int64_t B, int32_t T, float *pos, float *quat, void *) {
    return fk_any(w, q, B, T, pos, quat, true);
}
int drm_fk_fanout(const drm_walk *chains, int32_t T, const float *q, int64_t B, float *pos, float *quat, void *) {
    return fk_fan(chains, T, q, B, pos, quat, false);
}
int drm_fk_fanout_links(const drm_walk *chains, int32_t T, const float *q, int64_t B, float *pos, float *quat, void *) {
    return fk_fan(chains, T, q, B, pos, quat, true);
}

int drm_fk_jacobian(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin, float *ang, void *) {
    if (int rc = check_walk(w)) return rc;
    if (B < 0 || !q || !lin || !ang) return fail(DRM_ERR_INVALID, "bad batch or NULL q / lin_jac / ang_jac");
    const int n = w->n_dofs;
    for_chunks(B, [&](int64_t, int64_t b0, int64_t rows) {
        jac_loop(w, q + b0 * n, rows, pos ? pos + b0 * 3 : nullptr, quat ? quat + b0 * 4 : nullptr, lin + b0 * 3 * n, ang + b0 * 3 * n);
    });
    return DRM_OK;
}

// no scratch anywhere in the host build: every thread keeps its per-link records on its own stack / heap
int64_t drm_rnea_scratch_floats(const drm_walk *, int64_t) { return 0; }
int64_t drm_rnea_scratch_floats_aligned(const drm_walk *, int64_t) { return 0; }
int64_t drm_crba_scratch_floats(const drm_walk *, int64_t) { return 0; }
int64_t drm_crba_scratch_floats_aligned(const drm_walk *, int64_t) { return 0; }
int64_t drm_forward_dynamics_scratch_floats(const drm_walk *, int64_t) { return 0; }
int64_t drm_forward_dynamics_scratch_floats_aligned(const drm_walk *, int64_t) { return 0; }

int drm_rnea(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags, float *tau, float *,
             void *) {
    if (int rc = rnea_checked(w, q, qd, tau, B)) return rc;
    const int n = w->n_dofs;
    for_chunks(B, [&](int64_t, int64_t b0, int64_t rows) {
        rnea_loop(w, q + b0 * n, qd + b0 * n, qdd ? qdd + b0 * n : nullptr, rows, flags, tau + b0 * n);
    });
    return DRM_OK;
}

int drm_fk_rnea(const drm_walk *tree, const drm_walk *chain, int32_t, const float *q, const float *qd, const float *qdd, int64_t B,
                int32_t flags, float *tau, float *pos, float *quat, float *, void *) {
    if (int rc = rnea_checked(tree, q, qd, tau, B)) return rc;
    if (int rc = check_walk(chain)) return rc;
    if (!pos || !quat) return fail(DRM_ERR_INVALID, "pos / quat must not be NULL");
    const int n = tree->n_dofs;
    for_chunks(B, [&](int64_t, int64_t b0, int64_t rows) {
        rnea_loop(tree, q + b0 * n, qd + b0 * n, qdd ? qdd + b0 * n : nullptr, rows, flags, tau + b0 * n);
        fk_loop(chain, q + b0 * n, rows, 1, pos + b0 * 3, quat + b0 * 4);
    });
    return DRM_OK;
}

// ABI 11: the host build of the one-sided gather — the destinations are host arrays (e.g. shared-memory tensors of the other ranks)
int drm_fk_rnea_put(const drm_walk *tree, const drm_walk *chain, int32_t target_op, const float *q, const float *qd, const float *qdd,
                    int64_t B, int32_t flags, float *tau, float *pos, float *quat, float *scratch, const drm_put *put, void *stream) {
    if (put && (put->n_peers < 0 || put->n_peers > DRM_MAX_PEERS || put->row_offset < 0))
        return fail(DRM_ERR_INVALID, "drm_put: n_peers must be 0 .. DRM_MAX_PEERS and row_offset >= 0");
    if (int rc = drm_fk_rnea(tree, chain, target_op, q, qd, qdd, B, flags, tau, pos, quat, scratch, stream)) return rc;
    if (!put || B == 0) return DRM_OK;
    const int n = tree->n_dofs;
    for (int p = 0; p < put->n_peers; ++p) {
        if (put->tau[p]) memcpy(put->tau[p] + put->row_offset * n, tau, sizeof(float) * (size_t)B * n);
        if (put->pos[p]) memcpy(put->pos[p] + put->row_offset * 3, pos, sizeof(float) * (size_t)B * 3);
        if (put->quat[p]) memcpy(put->quat[p] + put->row_offset * 4, quat, sizeof(float) * (size_t)B * 4);
    }
    return DRM_OK;
}

int drm_crba(const drm_walk *w, const float *q, int64_t B, float *H, float *, void *) {
    if (int rc = check_walk(w)) return rc;
    if (B < 0 || !q || !H) return fail(DRM_ERR_INVALID, "bad batch or NULL q / H");
    const int n = w->n_dofs;
    for_chunks(B, [&](int64_t, int64_t b0, int64_t rows) { crba_loop(w, q + b0 * n, rows, H + b0 * n * n); });
    return DRM_OK;
}

int drm_forward_dynamics(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int32_t flags, float *qdd,
                         float *, void *) {
    if (int rc = rnea_checked(w, q, qd, qdd, B)) return rc;
    if (!f) return fail(DRM_ERR_INVALID, "f must not be NULL");
    const int n = w->n_dofs;
    for_chunks(B, [&](int64_t, int64_t b0, int64_t rows) {
        fd_loop(w, q + b0 * n, qd + b0 * n, f + b0 * n, rows, flags, qdd + b0 * n);
    });
    return DRM_OK;
}

// backward scratch: one [capacity, DRM_OPF_STRIDE] partial of the constant gradients per chunk of rows
int64_t drm_fk_backward_scratch_floats(int64_t B, int32_t capacity) {
    if (B < 0 || capacity < 1 || capacity > DRM_MAX_OPS) return 0;
    return n_chunks(B) * capacity * DRM_OPF_STRIDE;
}
int64_t drm_fk_mse_scratch_floats(int64_t B, int32_t capacity) {
    if (B < 0 || capacity < 1 || capacity > DRM_MAX_OPS) return 0;
    return n_chunks(B) * ((int64_t)capacity * DRM_OPF_STRIDE + 2); // + the chunk's part of the loss (a double)
}
int64_t drm_rnea_backward_scratch_floats(int64_t B, int32_t capacity, int32_t, int32_t) {
    return drm_fk_backward_scratch_floats(B, capacity);
}

int drm_fk_backward(const drm_walk *w, const float *q, int64_t B, int32_t T, const float *gpos, const float *grot, uint64_t mask,
                    float *gq, float *gops, float *scratch, void *) {
    if (int rc = check_backward_walk(w, mask)) return rc;
    if (B < 0 || T < 1 || !q || !gpos) return fail(DRM_ERR_INVALID, "bad batch / target count or NULL q / grad_pos");
    if ((mask != 0) != (gops != nullptr)) return fail(DRM_ERR_INVALID, "grad_ops_f must be given iff param_mask != 0");
    if (mask && !scratch) return fail(DRM_ERR_INVALID, "scratch must not be NULL (drm_fk_backward_scratch_floats)");
    const int n = w->n_dofs, entries = w->capacity * DRM_OPF_STRIDE;
    for_chunks(B, [&](int64_t c, int64_t b0, int64_t rows) {
        fkb_t(w, q + b0 * n, rows, T, gpos + b0 * T * 3, nullptr, nullptr, mask, gq ? gq + b0 * n : nullptr,
              mask ? scratch + c * entries : nullptr, grot ? grot + b0 * T * 9 : nullptr);
    });
    if (mask) reduce_partials(scratch, n_chunks(B), entries, gops);
    return DRM_OK;
}

int drm_fk_jacobian_backward(const drm_walk *w, const float *q, int64_t B, const float *gpos, const float *grot, const float *glin,
                             const float *gang, uint64_t mask, float *gq, float *gops, float *scratch, void *) {
    if (int rc = check_backward_walk(w, mask)) return rc;
    if (B < 0 || !q || !glin || !gang) return fail(DRM_ERR_INVALID, "bad batch or NULL q / grad_lin_jac / grad_ang_jac");
    if ((mask != 0) != (gops != nullptr)) return fail(DRM_ERR_INVALID, "grad_ops_f must be given iff param_mask != 0");
    if (mask && !scratch) return fail(DRM_ERR_INVALID, "scratch must not be NULL (drm_fk_backward_scratch_floats)");
    const int n = w->n_dofs, entries = w->capacity * DRM_OPF_STRIDE;
    for_chunks(B, [&](int64_t c, int64_t b0, int64_t rows) {
        fkb_t(w, q + b0 * n, rows, 1, gpos ? gpos + b0 * 3 : nullptr, glin + b0 * 3 * n, gang + b0 * 3 * n, mask,
              gq ? gq + b0 * n : nullptr, mask ? scratch + c * entries : nullptr, grot ? grot + b0 * 9 : nullptr);
    });
    if (mask) reduce_partials(scratch, n_chunks(B), entries, gops);
    return DRM_OK;
}

// forward pose, loss and adjoints chunk by chunk (any single-target walk, any B: the host build has no tile shape to respect)
int drm_fk_mse(const drm_walk *w, const float *q, const float *target, int64_t B, uint64_t mask, float *loss, float *gq, float *gops,
               float *scratch, void *) {
    if (int rc = check_backward_walk(w, mask)) return rc;
    if (B < 1 || !q || !target || !loss || !scratch) return fail(DRM_ERR_INVALID, "bad batch or NULL q / target / loss / scratch");
    if ((mask != 0) != (gops != nullptr)) return fail(DRM_ERR_INVALID, "grad_ops_f must be given iff param_mask != 0");
    const int n = w->n_dofs, entries = w->capacity * DRM_OPF_STRIDE;
    const int64_t chunks = n_chunks(B), stride = entries + 2;
    const float g_scale = 2.0f / (3.0f * (float)B);
    for_chunks(B, [&](int64_t c, int64_t b0, int64_t rows) {
        float pos[CHUNK * 3], quat[CHUNK * 4], g[CHUNK * 3];
        fk_loop(w, q + b0 * n, rows, 1, pos, quat);
        double part = 0.0;
        for (int64_t i = 0; i < rows * 3; ++i) {
            const float d = pos[i] - target[b0 * 3 + i];
            part += (double)d * d;
            g[i] = d * g_scale;
        }
        memcpy(scratch + c * stride + entries, &part, sizeof(double));
        fkb_t(w, q + b0 * n, rows, 1, g, nullptr, nullptr, mask, gq ? gq + b0 * n : nullptr, mask ? scratch + c * stride : nullptr);
    });
    double total = 0.0;
    for (int64_t c = 0; c < chunks; ++c) {
        double part;
        memcpy(&part, scratch + c * stride + entries, sizeof(double));
        total += part;
    }
    loss[0] = (float)(total / (3.0 * (double)B));
    if (mask)
        for (int e = 0; e < entries; ++e) {
            double s = 0.0;
            for (int64_t c = 0; c < chunks; ++c) s += scratch[c * stride + e];
            gops[e] = (float)s;
        }
    return DRM_OK;
}

int drm_rnea_backward(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags, const float *gtau,
                      uint64_t mask, float *gq, float *gqd, float *gqdd, float *gops, float *scratch, void *) {
    if (int rc = check_backward_walk(w, mask)) return rc;
    if (B < 0 || !q || !qd || !gtau) return fail(DRM_ERR_INVALID, "bad batch or NULL q / qd / grad_tau");
    if ((gq != nullptr) != (gqd != nullptr) || (gq != nullptr) != (gqdd != nullptr))
        return fail(DRM_ERR_INVALID, "grad_q / grad_qd / grad_qdd: all three or none");
    if ((mask != 0) != (gops != nullptr)) return fail(DRM_ERR_INVALID, "grad_ops_f must be given iff param_mask != 0");
    if (mask && !scratch) return fail(DRM_ERR_INVALID, "scratch must not be NULL (drm_rnea_backward_scratch_floats)");
    const int n = w->n_dofs, entries = w->capacity * DRM_OPF_STRIDE;
    for_chunks(B, [&](int64_t c, int64_t b0, int64_t rows) {
        rneab_t(w, q + b0 * n, qd + b0 * n, qdd ? qdd + b0 * n : nullptr, rows, flags, gtau + b0 * n, mask, gq ? gq + b0 * n : nullptr,
                gq ? gqd + b0 * n : nullptr, gq ? gqdd + b0 * n : nullptr, mask ? scratch + c * entries : nullptr);
    });
    if (mask) reduce_partials(scratch, n_chunks(B), entries, gops);
    return DRM_OK;
}

int drm_link_rows(const float *params, int32_t n_links, float *rows, void *) {
    if (n_links < 0 || !params || !rows) return fail(DRM_ERR_INVALID, "bad link count or NULL params / rows");
    for (int i = 0; i < n_links; ++i) link_row(params + i * LINK_PARAM_FLOATS, rows + i * DRM_OPF_STRIDE);
    return DRM_OK;
}
int drm_link_rows_backward(const float *params, const float *grad_rows, int32_t n_links, float *grad_params, void *) {
    if (n_links < 0 || !params || !grad_rows || !grad_params) return fail(DRM_ERR_INVALID, "bad link count or NULL argument");
    for (int i = 0; i < n_links; ++i)
        link_row_backward(params + i * LINK_PARAM_FLOATS, grad_rows + i * DRM_OPF_STRIDE, grad_params + i * LINK_PARAM_FLOATS);
    return DRM_OK;
}

int drm_walk_table(const float *params, int32_t n_links, const float *base, const int32_t *sel, const float *gsign, int32_t n_entries,
                   float *ops_f, void *) {
    if (n_links < 1 || n_links > 32 || n_entries < 0 || n_entries > DRM_MAX_OPS * DRM_OPF_STRIDE)
        return fail(DRM_ERR_INVALID, "drm_walk_table takes 1 .. 32 links and at most %ld entries (%ld)", DRM_MAX_OPS * DRM_OPF_STRIDE, n_entries);
    if (!params || !base || !sel || !gsign || !ops_f) return fail(DRM_ERR_INVALID, "NULL argument");
    float rows[32 * DRM_OPF_STRIDE];
    for (int i = 0; i < n_links; ++i) link_row(params + i * LINK_PARAM_FLOATS, rows + i * DRM_OPF_STRIDE);
    for (int e = 0; e < n_entries; ++e) {
        const int r = sel[e];
        if (r >= n_links * DRM_OPF_STRIDE) return fail(DRM_ERR_INVALID, "sel[%ld] = %ld is beyond the learnable rows", e, r);
        ops_f[e] = r >= 0 ? rows[r] * gsign[e] : base[e];
    }
    return DRM_OK;
}
int drm_walk_table_backward(const float *params, int32_t n_links, const float *grad_ops_f, const int32_t *sel, const float *gsign,
                            int32_t n_entries, float *grad_params, void *) {
    if (n_links < 1 || n_links > 32 || n_entries < 0 || n_entries > DRM_MAX_OPS * DRM_OPF_STRIDE)
        return fail(DRM_ERR_INVALID, "drm_walk_table_backward takes 1 .. 32 links and at most %ld entries (%ld)", DRM_MAX_OPS * DRM_OPF_STRIDE, n_entries);
    if (!params || !grad_ops_f || !sel || !gsign || !grad_params) return fail(DRM_ERR_INVALID, "NULL argument");
    float grows[32 * DRM_OPF_STRIDE];
    for (int r = 0; r < n_links * DRM_OPF_STRIDE; ++r) grows[r] = 0.0f;
    for (int e = 0; e < n_entries; ++e) { // entry order, as the kernel adds them
        const int r = sel[e];
        if (r >= 0 && r < n_links * DRM_OPF_STRIDE) grows[r] += grad_ops_f[e] * gsign[e];
    }
    for (int i = 0; i < n_links; ++i)
        link_row_backward(params + i * LINK_PARAM_FLOATS, grows + i * DRM_OPF_STRIDE, grad_params + i * LINK_PARAM_FLOATS);
    return DRM_OK;
}

// ABI 13: the table from the links' parameter tensors where they lie, in the forms their modules store them, and the gradient back
// to those tensors (include/drm_hip.h)
static int links_raw(const char *, const drm_link_pieces *links, const drm_link_forms *forms, int32_t n_links, int32_t n_entries,
                     float *raw, int32_t (*form)[3], float (*c)[3]) {
    if (!links) return fail(DRM_ERR_INVALID, "drm_walk_table_links: links must not be NULL");
    if (n_links < 1 || n_links > 32 || n_entries < 1 || n_entries > DRM_MAX_OPS * DRM_OPF_STRIDE)
        return fail(DRM_ERR_INVALID, "drm_walk_table_links: 1..32 learnable links and at most 32 x 32 walk entries (%ld, %ld)", n_links, n_entries);
    for (int l = 0; l < n_links; ++l) {
        const float *at[6] = {links[l].rot_angles, links[l].trans, links[l].mass, links[l].com, links[l].inertia_mat, links[l].damping};
        for (int j = 0; j < 6; ++j)
            if (!at[j]) return fail(DRM_ERR_INVALID, "drm_walk_table_links: a piece of link %ld is NULL", l);
        const drm_link_forms f = forms ? forms[l] : drm_link_forms{};
        const bool ok = (f.mass == DRM_FORM_PLAIN || f.mass == DRM_FORM_SQUARE_PLUS) &&
                        (f.damping == DRM_FORM_PLAIN || f.damping == DRM_FORM_SQUARE_PLUS) &&
                        (f.inertia_mat == DRM_FORM_PLAIN || (f.inertia_mat >= DRM_FORM_SYMM && f.inertia_mat <= DRM_FORM_COV));
        if (!ok) return fail(DRM_ERR_INVALID, "drm_walk_table_links: unknown form of a piece of link %ld", l);
        form[l][0] = f.mass; form[l][1] = f.inertia_mat; form[l][2] = f.damping;
        c[l][0] = f.mass_c; c[l][1] = f.inertia_mat_c; c[l][2] = f.damping_c;
        for (int k = 0; k < LINK_PARAM_FLOATS; ++k) {
            const int piece = link_piece_of(k), off = link_piece_offset(k);
            const bool there = piece != 4 || f.inertia_mat == DRM_FORM_PLAIN || off < 6;
            raw[l * LINK_PARAM_FLOATS + k] = there ? at[piece][off] : 0.0f;
        }
    }
    return DRM_OK;
}
int drm_walk_table_links(const drm_link_pieces *links, const drm_link_forms *forms, int32_t n_links, const float *base, const int32_t *sel,
                         const float *gsign, int32_t n_entries, float *ops_f, void *) {
    float raw[32 * LINK_PARAM_FLOATS], params[32 * LINK_PARAM_FLOATS], c[32][3];
    int32_t form[32][3];
    if (int rc = links_raw("drm_walk_table_links", links, forms, n_links, n_entries, raw, form, c)) return rc;
    for (int l = 0; l < n_links; ++l) link_forms_apply(form[l], c[l], raw + l * LINK_PARAM_FLOATS, params + l * LINK_PARAM_FLOATS);
    return drm_walk_table(params, n_links, base, sel, gsign, n_entries, ops_f, nullptr);
}
int drm_walk_table_links_backward(const drm_link_pieces *links, const drm_link_forms *forms, int32_t n_links, const float *grad_ops_f,
                                  const int32_t *sel, const float *gsign, int32_t n_entries, float *grad_params, void *) {
    float raw[32 * LINK_PARAM_FLOATS], params[32 * LINK_PARAM_FLOATS], c[32][3];
    int32_t form[32][3];
    if (int rc = links_raw("drm_walk_table_links_backward", links, forms, n_links, n_entries, raw, form, c)) return rc;
    for (int l = 0; l < n_links; ++l) link_forms_apply(form[l], c[l], raw + l * LINK_PARAM_FLOATS, params + l * LINK_PARAM_FLOATS);
    if (int rc = drm_walk_table_backward(params, n_links, grad_ops_f, sel, gsign, n_entries, grad_params, nullptr)) return rc;
    for (int l = 0; l < n_links; ++l) link_forms_grad(form[l], raw + l * LINK_PARAM_FLOATS, grad_params + l * LINK_PARAM_FLOATS);
    return DRM_OK;
}

// ABI 12: the composition itself (the host build has no launches to save): table from the links' parameters, drm_fk_mse, and the
// gradient back through the table's map
int drm_fk_mse_links(const drm_walk *w, const int32_t *sel, const float *gsign, const drm_link_pieces *links, int32_t n_links, const float *q,
                     const float *target, int64_t B, uint64_t mask, float *loss, float *gq, float *grad_params, float *scratch, void *) {
    if (int rc = check_backward_walk(w, mask)) return rc;
    if (!sel || !gsign || !links || !grad_params) return fail(DRM_ERR_INVALID, "NULL sel / gsign / links / grad_params");
    if (n_links < 1 || n_links > DRM_FK_MSE_MAX_LINKS)
        return fail(DRM_ERR_UNSUPPORTED, "drm_fk_mse_links takes 1 .. %ld learnable links (%ld)", DRM_FK_MSE_MAX_LINKS, n_links);
    if (!mask) return fail(DRM_ERR_INVALID, "param_mask must select the ops of the learnable links");
    const int entries = w->capacity * DRM_OPF_STRIDE;
    float params[DRM_FK_MSE_MAX_LINKS * LINK_PARAM_FLOATS] = {0.0f};
    for (int l = 0; l < n_links; ++l) {
        if (!links[l].rot_angles || !links[l].trans) return fail(DRM_ERR_INVALID, "rot_angles / trans of a link are NULL");
        for (int i = 0; i < 3; ++i) {
            params[l * LINK_PARAM_FLOATS + i] = links[l].rot_angles[i];
            params[l * LINK_PARAM_FLOATS + 3 + i] = links[l].trans[i];
        }
    }
    std::vector<float> table(entries), gops(entries);
    if (int rc = drm_walk_table(params, n_links, w->ops_f, sel, gsign, entries, table.data(), nullptr)) return rc;
    drm_walk live = *w;
    live.ops_f = table.data();
    if (int rc = drm_fk_mse(&live, q, target, B, mask, loss, gq, gops.data(), scratch, nullptr)) return rc;
    for (int e = 0; e < entries; ++e)       // forward kinematics reads the FT block of a row only
        if ((e % DRM_OPF_STRIDE) >= DRM_OPF_FT_FLOATS) gops[e] = 0.0f;
    if (int rc = drm_walk_table_backward(params, n_links, gops.data(), sel, gsign, entries, grad_params, nullptr)) return rc;
    for (int l = 0; l < n_links; ++l)
        for (int i = 6; i < LINK_PARAM_FLOATS; ++i) grad_params[l * LINK_PARAM_FLOATS + i] = 0.0f;
    return DRM_OK;
}
} // extern "C"
