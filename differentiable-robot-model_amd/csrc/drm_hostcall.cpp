// drm_hostcall.cpp — the per-call HOST work of the hot public methods in C++ (a torch extension, no device code): what
// backend.fk / fk_jacobian / rnea / crba / forward_dynamics do in Python for a call that builds no autograd graph — check the inputs, make
// the outputs (one allocation, views into it), call the C ABI (include/drm_hip.h) on the caller's stream — takes ~17 us of
// interpreter time there and ~4 us here (measured on the CPU build with a one-row batch, where the C call itself is ~1 us).
// Replaces nothing of the kernels: the entry point is handed over as an address (the ctypes function pointer of libdrm_hip.so or,
// for CPU tensors, libdrm_cpu.so), so this module links against neither.  An input that is not already what the kernels take
// (fp32, contiguous, 16-byte aligned, [B, n]) is not converted here: the call returns NOT_CONFORMING and the Python path, which
// converts, serves it.  The reference does this bookkeeping in Python too (robot_model.py:25-84, 223-248, 305-375, 626-667); the
// point is what a drop-in caller pays per call next to a 4 us kernel.
#include <torch/extension.h>
#include <torch/csrc/autograd/python_variable.h>

#include <dlfcn.h>

#include <tuple>
#include <vector>

namespace {
constexpr int64_t NOT_CONFORMING = 1; // (the C ABI's own codes are <= 0)

typedef int (*fk_fn)(const void *, const float *, int64_t, int32_t, float *, float *, void *);
typedef int (*fk_jacobian_fn)(const void *, const float *, int64_t, float *, float *, float *, float *, void *);
typedef int (*rnea_fn)(const void *, const float *, const float *, const float *, int64_t, int32_t, float *, float *, void *);
typedef int64_t (*scratch_fn)(const void *, int64_t);
typedef int (*crba_fn)(const void *, const float *, int64_t, float *, float *, void *);
typedef int (*fk_rnea_fn)(const void *, const void *, int32_t, const float *, const float *, const float *, int64_t, int32_t, float *, float *,
                          float *, float *, void *);
typedef int (*fd_fn)(const void *, const float *, const float *, const float *, int64_t, int32_t, float *, float *, void *);
typedef int (*table_links_fn)(const void *, const void *, int32_t, const float *, const int32_t *, const float *, int32_t, float *, void *);

inline int64_t pad4(int64_t x) { return (x + 3) & ~int64_t(3); } // every output starts on a 16-byte boundary
inline bool conforms(const at::Tensor &t, int64_t cols) {
    return t.dim() == 2 && t.size(1) == cols && t.scalar_type() == at::kFloat && t.is_contiguous() &&
           (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15u) == 0;
}
} // namespace

// (pos, quat, rc): pos [B, T, 3], quat [B, T, 4] — [B, 3], [B, 4] with `squeeze` and one target
std::tuple<at::Tensor, at::Tensor, int64_t> fk(int64_t fn, int64_t walk, const at::Tensor &q, int64_t n, int64_t n_targets, bool squeeze,
                                               int64_t stream) {
    if (!conforms(q, n)) return {at::Tensor(), at::Tensor(), NOT_CONFORMING};
    const int64_t B = q.size(0), T = n_targets, o1 = pad4(B * T * 3);
    at::Tensor flat = at::empty({o1 + pad4(B * T * 4)}, q.options());
    at::Tensor pos, quat;
    if (squeeze && T == 1) {
        pos = flat.as_strided({B, 3}, {3, 1}, 0);
        quat = flat.as_strided({B, 4}, {4, 1}, o1);
    } else {
        pos = flat.as_strided({B, T, 3}, {3 * T, 3, 1}, 0);
        quat = flat.as_strided({B, T, 4}, {4 * T, 4, 1}, o1);
    }
    int64_t rc = 0;
    if (B > 0) {
        float *base = flat.data_ptr<float>();
        rc = reinterpret_cast<fk_fn>(fn)(reinterpret_cast<const void *>(walk), q.data_ptr<float>(), B, (int32_t)T, base, base + o1,
                                         reinterpret_cast<void *>(stream));
    }
    return {pos, quat, rc};
}

// (pos [B, 3], quat [B, 4], lin_jac [B, 3, n], ang_jac [B, 3, n], rc)
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, int64_t> fk_jacobian(int64_t fn, int64_t walk, const at::Tensor &q, int64_t n,
                                                                                 int64_t stream) {
    if (!conforms(q, n)) return {at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), NOT_CONFORMING};
    const int64_t B = q.size(0);
    const int64_t o1 = pad4(B * 3), o2 = o1 + pad4(B * 4), o3 = o2 + pad4(B * 3 * n), total = o3 + pad4(B * 3 * n);
    at::Tensor flat = at::empty({total}, q.options());
    at::Tensor pos = flat.as_strided({B, 3}, {3, 1}, 0), quat = flat.as_strided({B, 4}, {4, 1}, o1);
    at::Tensor lin = flat.as_strided({B, 3, n}, {3 * n, n, 1}, o2), ang = flat.as_strided({B, 3, n}, {3 * n, n, 1}, o3);
    int64_t rc = 0;
    if (B > 0) {
        float *base = flat.data_ptr<float>();
        rc = reinterpret_cast<fk_jacobian_fn>(fn)(reinterpret_cast<const void *>(walk), q.data_ptr<float>(), B, base, base + o1, base + o2,
                                                  base + o3, reinterpret_cast<void *>(stream));
    }
    return {pos, quat, lin, ang, rc};
}

// (tau [B, n], rc); qdd may be an undefined tensor (joint accelerations zero).  scratch: the `_aligned` query of the ABI
std::tuple<at::Tensor, int64_t> rnea(int64_t fn, int64_t scratch_query, int64_t walk, const at::Tensor &q, const at::Tensor &qd,
                                     const c10::optional<at::Tensor> &qdd, int64_t n, int64_t flags, int64_t stream) {
    const bool has_qdd = qdd.has_value() && qdd->defined();
    if (!conforms(q, n) || !conforms(qd, n) || (has_qdd && !conforms(*qdd, n))) return {at::Tensor(), NOT_CONFORMING};
    const int64_t B = q.size(0);
    if (qd.size(0) != B || (has_qdd && qdd->size(0) != B) || qd.device() != q.device() || (has_qdd && qdd->device() != q.device()))
        return {at::Tensor(), NOT_CONFORMING}; // (the Python path words the error)
    at::Tensor tau = at::empty({B, n}, q.options());
    int64_t rc = 0;
    if (B > 0) {
        const int64_t need = reinterpret_cast<scratch_fn>(scratch_query)(reinterpret_cast<const void *>(walk), B);
        at::Tensor scratch;
        if (need > 0) scratch = at::empty({need}, q.options());
        rc = reinterpret_cast<rnea_fn>(fn)(reinterpret_cast<const void *>(walk), q.data_ptr<float>(), qd.data_ptr<float>(),
                                           has_qdd ? qdd->data_ptr<float>() : nullptr, B, (int32_t)flags, tau.data_ptr<float>(),
                                           need > 0 ? scratch.data_ptr<float>() : nullptr, reinterpret_cast<void *>(stream));
    }
    return {tau, rc};
}

// (H [B, n, n], rc)
std::tuple<at::Tensor, int64_t> crba(int64_t fn, int64_t scratch_query, int64_t walk, const at::Tensor &q, int64_t n, int64_t stream) {
    if (!conforms(q, n)) return {at::Tensor(), NOT_CONFORMING};
    const int64_t B = q.size(0);
    at::Tensor H = at::empty({B, n, n}, q.options());
    int64_t rc = 0;
    if (B > 0) {
        const int64_t need = reinterpret_cast<scratch_fn>(scratch_query)(reinterpret_cast<const void *>(walk), B);
        at::Tensor scratch;
        if (need > 0) scratch = at::empty({need}, q.options());
        rc = reinterpret_cast<crba_fn>(fn)(reinterpret_cast<const void *>(walk), q.data_ptr<float>(), B, H.data_ptr<float>(),
                                           need > 0 ? scratch.data_ptr<float>() : nullptr, reinterpret_cast<void *>(stream));
    }
    return {H, rc};
}

// (qdd [B, n], rc)
std::tuple<at::Tensor, int64_t> forward_dynamics(int64_t fn, int64_t scratch_query, int64_t walk, const at::Tensor &q, const at::Tensor &qd,
                                                 const at::Tensor &f, int64_t n, int64_t flags, int64_t stream) {
    if (!conforms(q, n) || !conforms(qd, n) || !conforms(f, n)) return {at::Tensor(), NOT_CONFORMING};
    const int64_t B = q.size(0);
    if (qd.size(0) != B || f.size(0) != B || qd.device() != q.device() || f.device() != q.device()) return {at::Tensor(), NOT_CONFORMING};
    at::Tensor qdd = at::empty({B, n}, q.options());
    int64_t rc = 0;
    if (B > 0) {
        const int64_t need = reinterpret_cast<scratch_fn>(scratch_query)(reinterpret_cast<const void *>(walk), B);
        at::Tensor scratch;
        if (need > 0) scratch = at::empty({need}, q.options());
        rc = reinterpret_cast<fd_fn>(fn)(reinterpret_cast<const void *>(walk), q.data_ptr<float>(), qd.data_ptr<float>(), f.data_ptr<float>(), B,
                                         (int32_t)flags, qdd.data_ptr<float>(), need > 0 ? scratch.data_ptr<float>() : nullptr,
                                         reinterpret_cast<void *>(stream));
    }
    return {qdd, rc};
}

// (tau [B, n], pos [B, 3], quat [B, 4], rc): drm_fk_rnea — inverse dynamics and the pose of one link from one pass over q
std::tuple<at::Tensor, at::Tensor, at::Tensor, int64_t> fk_rnea(int64_t fn, int64_t scratch_query, int64_t tree, int64_t chain, int64_t target_op,
                                                               const at::Tensor &q, const at::Tensor &qd, const c10::optional<at::Tensor> &qdd,
                                                               int64_t n, int64_t flags, int64_t stream) {
    const bool has_qdd = qdd.has_value() && qdd->defined();
    if (!conforms(q, n) || !conforms(qd, n) || (has_qdd && !conforms(*qdd, n))) return {at::Tensor(), at::Tensor(), at::Tensor(), NOT_CONFORMING};
    const int64_t B = q.size(0);
    if (qd.size(0) != B || (has_qdd && qdd->size(0) != B) || qd.device() != q.device() || (has_qdd && qdd->device() != q.device()))
        return {at::Tensor(), at::Tensor(), at::Tensor(), NOT_CONFORMING};
    const int64_t o1 = pad4(B * n), o2 = o1 + pad4(B * 3);
    at::Tensor flat = at::empty({o2 + pad4(B * 4)}, q.options());
    at::Tensor tau = flat.as_strided({B, n}, {n, 1}, 0), pos = flat.as_strided({B, 3}, {3, 1}, o1), quat = flat.as_strided({B, 4}, {4, 1}, o2);
    int64_t rc = 0;
    if (B > 0) {
        const int64_t need = reinterpret_cast<scratch_fn>(scratch_query)(reinterpret_cast<const void *>(tree), B);
        at::Tensor scratch;
        if (need > 0) scratch = at::empty({need}, q.options());
        float *base = flat.data_ptr<float>();
        rc = reinterpret_cast<fk_rnea_fn>(fn)(reinterpret_cast<const void *>(tree), reinterpret_cast<const void *>(chain), (int32_t)target_op,
                                              q.data_ptr<float>(), qd.data_ptr<float>(), has_qdd ? qdd->data_ptr<float>() : nullptr, B,
                                              (int32_t)flags, base, base + o1, base + o2, need > 0 ? scratch.data_ptr<float>() : nullptr,
                                              reinterpret_cast<void *>(stream));
    }
    return {tau, pos, quat, rc};
}

// K launches of a PREPARED drm_fk_jacobian call (backend.FkJacobianPlan: fixed buffers, raw addresses), enqueued back to back from
// C++: what a compiled caller's loop does (a Python loop spends ~5.5 us per launch in the interpreter and ctypes, more than the
// kernel takes).  Returns the first non-zero return code, or 0.
int64_t repeat_fk_jacobian(int64_t fn, int64_t walk, int64_t q, int64_t B, int64_t pos, int64_t quat, int64_t lin, int64_t ang, int64_t stream,
                           int64_t K) {
    for (int64_t k = 0; k < K; ++k) {
        const int rc = reinterpret_cast<fk_jacobian_fn>(fn)(reinterpret_cast<const void *>(walk), reinterpret_cast<const float *>(q), B,
                                                            reinterpret_cast<float *>(pos), reinterpret_cast<float *>(quat),
                                                            reinterpret_cast<float *>(lin), reinterpret_cast<float *>(ang),
                                                            reinterpret_cast<void *>(stream));
        if (rc) return rc;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// FastCall (round 6): ONE prepared public-method call of a constant model — compute_forward_kinematics / compute_endeffector_jacobian /
// compute_fk_and_jacobian of a link, compute_inverse_dynamics — with EVERYTHING the Python method does per call on this side of the
// boundary: the reference's tensor_check (robot_model.py:25-84: device type, ndim 1 or 2, one batch shape, 1-D <-> [1, n] promotion),
// the shape asserts of the method, the "does this call build an autograd graph" test, the output allocation and the C-ABI call.
// A call it does not take AS IS — wrong device / dtype / layout, a 1-D / 2-D mix, gradients wanted, another current device —
// returns None and the Python method (which words the reference's AssertionErrors, converts, differentiates) serves it.
// The walk structs live in Python objects the FastCall keeps alive (`keep`); the Python side drops a FastCall whenever the walk's
// struct is rebuilt (own kernels attached, a parameter made learnable).
typedef int (*hip_get_device_fn)(int *);
static hip_get_device_fn hip_get_device() {
    static hip_get_device_fn fn = reinterpret_cast<hip_get_device_fn>(dlsym(RTLD_DEFAULT, "hipGetDevice"));
    return fn;
}

struct FastCall {
    int64_t fn = 0, scratch_query = 0, walk = 0, walk2 = 0, n = 0, target_op = -1;
    bool cuda = false;
    int64_t device_index = -1;
    pybind11::object keep;

    FastCall(int64_t fn_, int64_t scratch_query_, int64_t walk_, int64_t walk2_, int64_t n_, int64_t target_op_, bool cuda_,
             int64_t device_index_, pybind11::object keep_)
        : fn(fn_), scratch_query(scratch_query_), walk(walk_), walk2(walk2_), n(n_), target_op(target_op_), cuda(cuda_),
          device_index(device_index_), keep(std::move(keep_)) {}

    // ---- a model WITH learnable link parameters (round 6, ABI 13): its walk table is rebuilt from the parameter tensors in front of
    // every call (drm_walk_table_links: one small launch on the same stream) — when the call builds no autograd graph, i.e. under
    // torch.no_grad() or with every parameter frozen: a learned model in a control loop.  Every live piece is looked up in its
    // module's own parameter dictionary on every call (a parameter that was moved, replaced or re-typed is seen; one the kernels
    // cannot read as it lies sends the call to the Python path).  `walk` is then this call's OWN copy of the struct: its table
    // pointer is rewritten per call.
    int64_t table_fn = 0, forms = 0, n_links = 0, base = 0, sel = 0, gsign = 0, n_entries = 0;
    std::vector<const float *> pieces;                  // n_links x 6 (struct drm_link_pieces)
    std::vector<int64_t> live_slot, live_size;          // the slots of `pieces` that parameter modules supply, and their element counts
    std::vector<pybind11::object> live_dict, live_key;  // module._parameters, the parameter's name
    at::Tensor base_tensor;

    void set_table(int64_t table_fn_, int64_t links_, int64_t forms_, int64_t n_links_, const at::Tensor &base_, int64_t sel_, int64_t gsign_,
                   std::vector<int64_t> slots, std::vector<int64_t> sizes, std::vector<pybind11::object> dicts, std::vector<pybind11::object> keys) {
        TORCH_CHECK(slots.size() == sizes.size() && slots.size() == dicts.size() && slots.size() == keys.size(), "set_table: ragged lists");
        table_fn = table_fn_; forms = forms_; n_links = n_links_; sel = sel_; gsign = gsign_;
        base_tensor = base_; base = reinterpret_cast<int64_t>(base_.data_ptr()); n_entries = base_.numel();
        const float *const *src = reinterpret_cast<const float *const *>(links_);
        pieces.assign(src, src + n_links_ * 6);
        live_slot = std::move(slots); live_size = std::move(sizes); live_dict = std::move(dicts); live_key = std::move(keys);
    }
    // 1: the table is built (held keeps it alive until the launches are enqueued), 0: not servable here (the Python path), < 0: an error code
    int64_t table(int64_t stream, at::Tensor &held) {
        const bool grad = at::GradMode::is_enabled();
        for (size_t i = 0; i < live_slot.size(); ++i) {
            PyObject *o = PyDict_GetItemWithError(live_dict[i].ptr(), live_key[i].ptr());     // (borrowed)
            if (!o) { PyErr_Clear(); return 0; }
            if (!THPVariable_Check(o)) return 0;
            const at::Tensor &t = THPVariable_Unpack(o);
            if ((grad && t.requires_grad()) || t.scalar_type() != at::kFloat || !t.is_contiguous() || t.numel() != live_size[i] ||
                t.is_cuda() != cuda || (cuda && t.get_device() != device_index))
                return 0;
            pieces[(size_t)live_slot[i]] = t.data_ptr<float>();
        }
        held = at::empty_like(base_tensor);
        const int rc = reinterpret_cast<table_links_fn>(table_fn)(pieces.data(), reinterpret_cast<const void *>(forms), (int32_t)n_links,
                                                                   reinterpret_cast<const float *>(base), reinterpret_cast<const int32_t *>(sel),
                                                                   reinterpret_cast<const float *>(gsign), (int32_t)n_entries,
                                                                   held.data_ptr<float>(), reinterpret_cast<void *>(stream));
        if (rc) return rc;
        *reinterpret_cast<void **>(walk) = held.data_ptr();      // (drm_walk.ops_f is the struct's first member)
        return 1;
    }
#define DRM_FAST_TABLE(stream)                                                  \
    at::Tensor table_held;                                                      \
    if (table_fn) {                                                             \
        const int64_t trc = table(stream, table_held);                          \
        if (trc == 0) return pybind11::none();                                  \
        if (trc < 0) return pybind11::int_(trc);                                \
    }

    // the tensor behind a Python argument when it is EXACTLY a torch.Tensor (tensor_check tests `type(arg) is torch.Tensor`) that the
    // kernels take as it stands; rows = -1 for a 1-D tensor
    bool take(pybind11::handle h, const at::Tensor *&out, int64_t &rows) const {
        PyObject *o = h.ptr();
        if (!THPVariable_CheckExact(o)) return false;
        const at::Tensor &t = THPVariable_Unpack(o);
        const int64_t d = t.dim();
        if (d != 1 && d != 2) return false;
        if (t.size(d - 1) != n || t.scalar_type() != at::kFloat || !t.is_contiguous() ||
            (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15u) != 0)
            return false;
        if (t.is_cuda() != cuda || (cuda && t.get_device() != device_index) || (!cuda && !t.is_cpu())) return false;
        out = &t;
        rows = d == 1 ? -1 : t.size(0);
        return true;
    }
    bool device_is_current() const {
        if (!cuda) return true;
        int dev = -1;
        hip_get_device_fn get = hip_get_device();
        return get && get(&dev) == 0 && dev == device_index;
    }
    static bool graph_wanted(const at::Tensor &t) { return t.requires_grad() && at::GradMode::is_enabled(); }
    static pybind11::object wrap(const at::Tensor &t) { return pybind11::reinterpret_steal<pybind11::object>(THPVariable_Wrap(t)); }

    // kind 0: (pos, quat);  1: (lin_jac, ang_jac);  2: (pos, quat, lin_jac, ang_jac)
    pybind11::object kinematics(pybind11::handle hq, int64_t kind, int64_t stream) {
        const at::Tensor *q;
        int64_t rows;
        if (!take(hq, q, rows) || graph_wanted(*q) || !device_is_current()) return pybind11::none();
        DRM_FAST_TABLE(stream)
        const bool one = rows < 0;
        const int64_t B = one ? 1 : rows;
        const int64_t o1 = pad4(B * 3), o2 = o1 + pad4(B * 4), o3 = kind ? o2 + pad4(B * 3 * n) : o2, total = kind ? o3 + pad4(B * 3 * n) : o2;
        at::Tensor flat = at::empty({total}, q->options());
        int64_t rc = 0;
        if (B > 0) {
            float *base = flat.data_ptr<float>();
            if (kind == 0)
                rc = reinterpret_cast<fk_fn>(fn)(reinterpret_cast<const void *>(walk), q->data_ptr<float>(), B, 1, base, base + o1,
                                                 reinterpret_cast<void *>(stream));
            else
                rc = reinterpret_cast<fk_jacobian_fn>(fn)(reinterpret_cast<const void *>(walk), q->data_ptr<float>(), B, base, base + o1, base + o2,
                                                          base + o3, reinterpret_cast<void *>(stream));
        }
        if (rc) return pybind11::int_(rc);
        auto view = [&](int64_t off, std::initializer_list<int64_t> shape2, std::initializer_list<int64_t> stride2,
                        std::initializer_list<int64_t> shape1, std::initializer_list<int64_t> stride1) {
            return one ? flat.as_strided(shape1, stride1, off) : flat.as_strided(shape2, stride2, off);
        };
        if (kind == 0) return pybind11::make_tuple(wrap(view(0, {B, 3}, {3, 1}, {3}, {1})), wrap(view(o1, {B, 4}, {4, 1}, {4}, {1})));
        pybind11::object lin = wrap(view(o2, {B, 3, n}, {3 * n, n, 1}, {3, n}, {n, 1}));
        pybind11::object ang = wrap(view(o3, {B, 3, n}, {3 * n, n, 1}, {3, n}, {n, 1}));
        if (kind == 1) return pybind11::make_tuple(lin, ang);
        return pybind11::make_tuple(wrap(view(0, {B, 3}, {3, 1}, {3}, {1})), wrap(view(o1, {B, 4}, {4, 1}, {4}, {1})), lin, ang);
    }

    // compute_forward_kinematics_all_links (robot_model.py:197-221): the T targets of the walk through drm_fk_links' LINK-MAJOR outputs,
    // handed back as a list of T (pos [B, 3], quat [B, 4]) pairs — contiguous views of ONE allocation, in the walk's target order.
    // Batched q only (a 1-D q keeps the Python path, whose dictionary passes through tensor_check unsqueezed, as upstream's does).
    pybind11::object links(pybind11::handle hq, int64_t T, int64_t stream) {
        const at::Tensor *q;
        int64_t rows;
        if (!take(hq, q, rows) || rows < 0 || T < 1 || graph_wanted(*q) || !device_is_current()) return pybind11::none();
        DRM_FAST_TABLE(stream)
        const int64_t B = rows, o1 = pad4(T * B * 3), total = o1 + pad4(T * B * 4);
        at::Tensor flat = at::empty({total}, q->options());
        int64_t rc = 0;
        if (B > 0) {
            float *base = flat.data_ptr<float>();
            rc = reinterpret_cast<fk_fn>(fn)(reinterpret_cast<const void *>(walk), q->data_ptr<float>(), B, (int32_t)T, base, base + o1,
                                             reinterpret_cast<void *>(stream));
        }
        if (rc) return pybind11::int_(rc);
        pybind11::list out(T);
        for (int64_t t = 0; t < T; ++t)
            out[t] = pybind11::make_tuple(wrap(flat.as_strided({B, 3}, {3, 1}, t * B * 3)), wrap(flat.as_strided({B, 4}, {4, 1}, o1 + t * B * 4)));
        return out;
    }

    // tau [B, n] ([n] for 1-D inputs); hqdd may be None (the non-linear effects)
    pybind11::object inverse_dynamics(pybind11::handle hq, pybind11::handle hqd, pybind11::handle hqdd, int64_t flags, int64_t stream) {
        const at::Tensor *q, *qd, *qdd = nullptr;
        int64_t rows, rows_d, rows_dd;
        if (!take(hq, q, rows) || !take(hqd, qd, rows_d) || rows_d != rows) return pybind11::none();
        if (!hqdd.is_none() && (!take(hqdd, qdd, rows_dd) || rows_dd != rows)) return pybind11::none();
        if (graph_wanted(*q) || graph_wanted(*qd) || (qdd && graph_wanted(*qdd)) || !device_is_current()) return pybind11::none();
        DRM_FAST_TABLE(stream)
        const bool one = rows < 0;
        const int64_t B = one ? 1 : rows;
        at::Tensor tau = one ? at::empty({n}, q->options()) : at::empty({B, n}, q->options());
        int64_t rc = 0;
        if (B > 0) {
            const int64_t need = reinterpret_cast<scratch_fn>(scratch_query)(reinterpret_cast<const void *>(walk), B);
            at::Tensor scratch;
            if (need > 0) scratch = at::empty({need}, q->options());
            rc = reinterpret_cast<rnea_fn>(fn)(reinterpret_cast<const void *>(walk), q->data_ptr<float>(), qd->data_ptr<float>(),
                                               qdd ? qdd->data_ptr<float>() : nullptr, B, (int32_t)flags, tau.data_ptr<float>(),
                                               need > 0 ? scratch.data_ptr<float>() : nullptr, reinterpret_cast<void *>(stream));
        }
        if (rc) return pybind11::int_(rc);
        return wrap(tau);
    }

    // H [B, n, n] ([n, n] for a 1-D q): drm_crba
    pybind11::object inertia_matrix(pybind11::handle hq, int64_t stream) {
        const at::Tensor *q;
        int64_t rows;
        if (!take(hq, q, rows) || graph_wanted(*q) || !device_is_current()) return pybind11::none();
        DRM_FAST_TABLE(stream)
        const bool one = rows < 0;
        const int64_t B = one ? 1 : rows;
        at::Tensor H = one ? at::empty({n, n}, q->options()) : at::empty({B, n, n}, q->options());
        int64_t rc = 0;
        if (B > 0) {
            const int64_t need = reinterpret_cast<scratch_fn>(scratch_query)(reinterpret_cast<const void *>(walk), B);
            at::Tensor scratch;
            if (need > 0) scratch = at::empty({need}, q->options());
            rc = reinterpret_cast<crba_fn>(fn)(reinterpret_cast<const void *>(walk), q->data_ptr<float>(), B, H.data_ptr<float>(),
                                               need > 0 ? scratch.data_ptr<float>() : nullptr, reinterpret_cast<void *>(stream));
        }
        if (rc) return pybind11::int_(rc);
        return wrap(H);
    }

    // qdd [B, n]: drm_forward_dynamics
    pybind11::object forward_dynamics(pybind11::handle hq, pybind11::handle hqd, pybind11::handle hf, int64_t flags, int64_t stream) {
        const at::Tensor *q, *qd, *f;
        int64_t rows, rows_d, rows_f;
        if (!take(hq, q, rows) || !take(hqd, qd, rows_d) || !take(hf, f, rows_f) || rows_d != rows || rows_f != rows) return pybind11::none();
        if (graph_wanted(*q) || graph_wanted(*qd) || graph_wanted(*f) || !device_is_current()) return pybind11::none();
        DRM_FAST_TABLE(stream)
        const bool one = rows < 0;
        const int64_t B = one ? 1 : rows;
        at::Tensor qdd = one ? at::empty({n}, q->options()) : at::empty({B, n}, q->options());
        int64_t rc = 0;
        if (B > 0) {
            const int64_t need = reinterpret_cast<scratch_fn>(scratch_query)(reinterpret_cast<const void *>(walk), B);
            at::Tensor scratch;
            if (need > 0) scratch = at::empty({need}, q->options());
            rc = reinterpret_cast<fd_fn>(fn)(reinterpret_cast<const void *>(walk), q->data_ptr<float>(), qd->data_ptr<float>(), f->data_ptr<float>(),
                                             B, (int32_t)flags, qdd.data_ptr<float>(), need > 0 ? scratch.data_ptr<float>() : nullptr,
                                             reinterpret_cast<void *>(stream));
        }
        if (rc) return pybind11::int_(rc);
        return wrap(qdd);
    }

    // (tau [B, n], pos [B, 3], quat [B, 4]): drm_fk_rnea on (walk = the dynamics walk, walk2 = the target's chain walk)
    pybind11::object fk_inverse_dynamics(pybind11::handle hq, pybind11::handle hqd, pybind11::handle hqdd, int64_t flags, int64_t stream) const {
        const at::Tensor *q, *qd, *qdd;
        if (table_fn) return pybind11::none();      // (two walks, two tables: the Python path)
        int64_t rows, rows_d, rows_dd;
        if (!take(hq, q, rows) || !take(hqd, qd, rows_d) || !take(hqdd, qdd, rows_dd) || rows_d != rows || rows_dd != rows) return pybind11::none();
        if (!device_is_current()) return pybind11::none();      // (this entry point is not differentiable: no graph test)
        const bool one = rows < 0;
        const int64_t B = one ? 1 : rows;
        const int64_t o1 = pad4(B * n), o2 = o1 + pad4(B * 3);
        at::Tensor flat = at::empty({o2 + pad4(B * 4)}, q->options());
        int64_t rc = 0;
        if (B > 0) {
            const int64_t need = reinterpret_cast<scratch_fn>(scratch_query)(reinterpret_cast<const void *>(walk), B);
            at::Tensor scratch;
            if (need > 0) scratch = at::empty({need}, flat.options());
            float *base = flat.data_ptr<float>();
            rc = reinterpret_cast<fk_rnea_fn>(fn)(reinterpret_cast<const void *>(walk), reinterpret_cast<const void *>(walk2), (int32_t)target_op,
                                                  q->data_ptr<float>(), qd->data_ptr<float>(), qdd->data_ptr<float>(), B, (int32_t)flags, base,
                                                  base + o1, base + o2, need > 0 ? scratch.data_ptr<float>() : nullptr,
                                                  reinterpret_cast<void *>(stream));
        }
        if (rc) return pybind11::int_(rc);
        if (one)
            return pybind11::make_tuple(wrap(flat.as_strided({n}, {1}, 0)), wrap(flat.as_strided({3}, {1}, o1)), wrap(flat.as_strided({4}, {1}, o2)));
        return pybind11::make_tuple(wrap(flat.as_strided({B, n}, {n, 1}, 0)), wrap(flat.as_strided({B, 3}, {3, 1}, o1)),
                                    wrap(flat.as_strided({B, 4}, {4, 1}, o2)));
    }
};

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    pybind11::class_<FastCall>(m, "FastCall")
        .def(pybind11::init<int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, bool, int64_t, pybind11::object>())
        .def("set_table", &FastCall::set_table)
        .def("kinematics", &FastCall::kinematics)
        .def("links", &FastCall::links)
        .def("inverse_dynamics", &FastCall::inverse_dynamics)
        .def("inertia_matrix", &FastCall::inertia_matrix)
        .def("forward_dynamics", &FastCall::forward_dynamics)
        .def("fk_inverse_dynamics", &FastCall::fk_inverse_dynamics);

    m.def("repeat_fk_jacobian", &repeat_fk_jacobian, pybind11::call_guard<pybind11::gil_scoped_release>());
    m.def("crba", &crba);
    m.def("fk_rnea", &fk_rnea);
    m.def("forward_dynamics", &forward_dynamics);
    m.attr("NOT_CONFORMING") = NOT_CONFORMING;
    m.def("fk", &fk);
    m.def("fk_jacobian", &fk_jacobian);
    m.def("rnea", &rnea);
}
