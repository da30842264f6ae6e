"""ctypes binding of the C ABI in include/drm_hip.h: csrc/libdrm_hip.so for tensors on a HIP device, csrc/libdrm_cpu.so (the
same entry points compiled by g++ over host pointers, csrc/drm_cpu.cpp) for tensors on the CPU — the reference's default device
(robot_model.py:100-104).

PyTorch is plumbing here: it owns the buffers and the HIP stream; the kernels are launched on
``torch.cuda.current_stream()`` through plain pointers.  ``import torch`` must precede loading the HIP library so that both share
the HIP runtime that is already mapped into the process.

The library is chosen by the DEVICE OF THE TENSORS and by nothing else (`library_for`): there is no fallback from one to the
other — a HIP tensor without libdrm_hip.so raises NativeLibraryError, as does a CPU tensor without libdrm_cpu.so.
"""
import ctypes
import math
import os
import threading

import torch

from .flatten import MAX_SEGMENTS, OPI_PERM, WalkProgram

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdrm_hip.so")
CPU_LIB_PATH = os.path.join(_HERE, "csrc", "libdrm_cpu.so")
ABI_VERSION = 13

RNEA_GRAVITY, RNEA_DAMPING = 1, 2
SPECIAL_FK_FAN_LINKS = 9      # index of the fan-out FK kernel in drm_walk.special[] (include/drm_hip.h DRM_SPECIAL_FK_FAN_LINKS)
WALK_TICKET = 10               # ... and of the walk's ticket word (ABI 11, DRM_WALK_TICKET): one-launch backward reductions


class DrmWalk(ctypes.Structure):
    """Mirror of ``struct drm_walk`` (include/drm_hip.h)."""
    _fields_ = [("ops_f", ctypes.c_void_p), ("ops_i", ctypes.c_void_p),
                ("n_ops", ctypes.c_int32), ("capacity", ctypes.c_int32),
                ("n_dofs", ctypes.c_int32), ("n_slots", ctypes.c_int32),
                ("dof_mask", ctypes.c_uint64), ("target_perm", ctypes.c_int32), ("shape", ctypes.c_int32),
                ("n_segments", ctypes.c_int32), ("seg_begin", ctypes.c_int32 * (MAX_SEGMENTS + 1)),
                ("seg_dof_lo", ctypes.c_int32 * MAX_SEGMENTS), ("seg_dof_cnt", ctypes.c_int32 * MAX_SEGMENTS),
                ("prefix_end", ctypes.c_int32), ("seg_leaf_begin", ctypes.c_int32 * (MAX_SEGMENTS + 1)),
                ("chain_dof1", ctypes.c_uint8 * 16), ("chain_prismatic", ctypes.c_uint32), ("reserved0", ctypes.c_uint32),
                ("special", ctypes.c_void_p * 16)]      # per-robot straight-line kernels of this walk (specialize.py), or NULL


MAX_PEERS = 8


class DrmPut(ctypes.Structure):
    """Mirror of ``struct drm_put`` (include/drm_hip.h, ABI 11): the destination sets of drm_fk_rnea_put's one-sided gather."""
    _fields_ = [("n_peers", ctypes.c_int32), ("reserved", ctypes.c_int32), ("row_offset", ctypes.c_int64),
                ("tau", ctypes.c_void_p * MAX_PEERS), ("pos", ctypes.c_void_p * MAX_PEERS), ("quat", ctypes.c_void_p * MAX_PEERS)]


FK_MSE_MAX_LINKS = 8


class DrmLinkPieces(ctypes.Structure):
    """Mirror of ``struct drm_link_pieces`` (include/drm_hip.h, ABI 12): where the outputs of one learnable link's parameter modules lie."""
    _fields_ = [(name, ctypes.c_void_p) for name in ("rot_angles", "trans", "mass", "com", "inertia_mat", "damping")]


FORM_PLAIN, FORM_SQUARE_PLUS, FORM_SYMM, FORM_SPD, FORM_COV = 0, 1, 2, 3, 4      # DRM_FORM_* of include/drm_hip.h (ABI 13)


class DrmLinkForms(ctypes.Structure):
    """Mirror of ``struct drm_link_forms`` (include/drm_hip.h, ABI 13): the form mass / inertia_mat / damping of a learnable link are stored in."""
    _fields_ = [("mass", ctypes.c_int32), ("inertia_mat", ctypes.c_int32), ("damping", ctypes.c_int32),
                ("mass_c", ctypes.c_float), ("inertia_mat_c", ctypes.c_float), ("damping_c", ctypes.c_float)]


class NativeLibraryError(RuntimeError):
    pass


_libs = {}      # "hip" / "cpu" -> the loaded library
_lock = threading.Lock()

EXPORTS = ("drm_abi_version", "drm_walk_sizeof", "drm_last_error", "drm_fk", "drm_fk_jacobian", "drm_rnea", "drm_fk_backward",
           "drm_fk_backward_scratch_floats", "drm_crba", "drm_rnea_backward",
           "drm_rnea_backward_scratch_floats", "drm_forward_dynamics", "drm_link_rows",
           "drm_link_rows_backward", "drm_fk_fanout", "drm_fk_fanout_links", "drm_fk_links", "drm_fk_jacobian_backward", "drm_walk_table",
           "drm_walk_table_backward", "drm_fk_rnea", "drm_forward_dynamics_scratch_floats", "drm_crba_scratch_floats",
           "drm_rnea_scratch_floats", "drm_fk_mse", "drm_fk_mse_scratch_floats", "drm_rnea_scratch_floats_aligned",
           "drm_crba_scratch_floats_aligned", "drm_forward_dynamics_scratch_floats_aligned", "drm_special_load", "drm_fk_rnea_put",
           "drm_fk_mse_links", "drm_walk_table_links", "drm_walk_table_links_backward")


def library_for(device):
    """The library that computes for tensors on `device`: libdrm_hip.so for a HIP device, libdrm_cpu.so for the CPU."""
    kind = device.type
    lib = _libs.get(kind)
    if lib is not None:
        return lib
    if kind == "cuda":
        return load_library()
    if kind == "cpu":
        return load_library(kind="cpu")
    raise RuntimeError("tensors must live on a HIP device or on the CPU (got %s)" % device)


def load_library(path: str = None, kind: str = "cuda"):
    """Load csrc/libdrm_hip.so (kind "cuda": the library behind HIP tensors) or csrc/libdrm_cpu.so (kind "cpu": the host build of
    the same ABI, behind CPU tensors) once and declare the prototypes.  Raises NativeLibraryError if absent."""
    with _lock:
        if kind in _libs:
            return _libs[kind]
        if kind == "cuda":
            path = path or os.environ.get("DRM_HIP_LIBRARY", LIB_PATH)
            if not os.path.exists(path):
                raise NativeLibraryError(
                    "native HIP library not found at %s — build it with `python __graft_entry__.py build` "
                    "(or `make -C differentiable-robot-model_amd/csrc`); there is no CPU fallback for tensors on a HIP device" % path)
        else:
            path = path or os.environ.get("DRM_CPU_LIBRARY", CPU_LIB_PATH)
            if not os.path.exists(path):
                raise NativeLibraryError(
                    "host build of the library not found at %s — build it with `python __graft_entry__.py build` "
                    "(or `make -C differentiable-robot-model_amd/csrc libdrm_cpu.so`)" % path)
        try:
            lib = ctypes.CDLL(path)
        except OSError as err:
            raise NativeLibraryError("cannot load %s: %s" % (path, err))
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        wp = ctypes.POINTER(DrmWalk)
        lib.drm_abi_version.restype = ctypes.c_int
        lib.drm_abi_version.argtypes = []
        lib.drm_last_error.restype = ctypes.c_char_p
        lib.drm_last_error.argtypes = []
        lib.drm_fk.restype = ctypes.c_int
        lib.drm_fk.argtypes = [wp, vp, i64, i32, vp, vp, vp]
        lib.drm_fk_links.restype = ctypes.c_int
        lib.drm_fk_links.argtypes = [wp, vp, i64, i32, vp, vp, vp]
        lib.drm_fk_fanout.restype = ctypes.c_int
        lib.drm_fk_fanout.argtypes = [wp, i32, vp, i64, vp, vp, vp]
        lib.drm_fk_fanout_links.restype = ctypes.c_int
        lib.drm_fk_fanout_links.argtypes = [wp, i32, vp, i64, vp, vp, vp]
        lib.drm_fk_jacobian.restype = ctypes.c_int
        lib.drm_fk_jacobian.argtypes = [wp, vp, i64, vp, vp, vp, vp, vp]
        lib.drm_rnea.restype = ctypes.c_int
        lib.drm_rnea.argtypes = [wp, vp, vp, vp, i64, i32, vp, vp, vp]
        lib.drm_rnea_scratch_floats.restype = i64
        lib.drm_rnea_scratch_floats.argtypes = [wp, i64]
        lib.drm_fk_backward.restype = ctypes.c_int
        lib.drm_fk_backward.argtypes = [wp, vp, i64, i32, vp, vp, ctypes.c_uint64, vp, vp, vp, vp]
        lib.drm_fk_jacobian_backward.restype = ctypes.c_int
        lib.drm_fk_jacobian_backward.argtypes = [wp, vp, i64, vp, vp, vp, vp, ctypes.c_uint64, vp, vp, vp, vp]
        lib.drm_fk_backward_scratch_floats.restype = i64
        lib.drm_fk_backward_scratch_floats.argtypes = [i64, i32]
        lib.drm_rnea_backward.restype = ctypes.c_int
        lib.drm_rnea_backward.argtypes = [wp, vp, vp, vp, i64, i32, vp, ctypes.c_uint64, vp, vp, vp, vp, vp, vp]
        lib.drm_rnea_backward_scratch_floats.restype = i64
        lib.drm_rnea_backward_scratch_floats.argtypes = [i64, i32, i32, i32]
        lib.drm_link_rows.restype = ctypes.c_int
        lib.drm_link_rows.argtypes = [vp, i32, vp, vp]
        lib.drm_link_rows_backward.restype = ctypes.c_int
        lib.drm_link_rows_backward.argtypes = [vp, vp, i32, vp, vp]
        lib.drm_walk_table.restype = ctypes.c_int
        lib.drm_walk_table.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp]
        lib.drm_walk_table_backward.restype = ctypes.c_int
        lib.drm_walk_table_backward.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp]
        lib.drm_forward_dynamics.restype = ctypes.c_int
        lib.drm_forward_dynamics.argtypes = [wp, vp, vp, vp, i64, i32, vp, vp, vp]
        lib.drm_forward_dynamics_scratch_floats.restype = i64
        lib.drm_forward_dynamics_scratch_floats.argtypes = [wp, i64]
        lib.drm_crba.restype = ctypes.c_int
        lib.drm_crba.argtypes = [wp, vp, i64, vp, vp, vp]
        lib.drm_crba_scratch_floats.restype = i64
        lib.drm_crba_scratch_floats.argtypes = [wp, i64]
        lib.drm_fk_rnea.restype = ctypes.c_int
        lib.drm_fk_rnea.argtypes = [wp, wp, i32, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp]
        lib.drm_fk_rnea_put.restype = ctypes.c_int
        lib.drm_fk_rnea_put.argtypes = [wp, wp, i32, vp, vp, vp, i64, i32, vp, vp, vp, vp, ctypes.POINTER(DrmPut), vp]
        for name in ("drm_rnea_scratch_floats_aligned", "drm_crba_scratch_floats_aligned", "drm_forward_dynamics_scratch_floats_aligned"):
            getattr(lib, name).restype = i64
            getattr(lib, name).argtypes = [wp, i64]
        lib.drm_special_load.restype = ctypes.c_int
        lib.drm_special_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        lib.drm_fk_mse.restype = ctypes.c_int
        lib.drm_fk_mse.argtypes = [wp, vp, vp, i64, ctypes.c_uint64, vp, vp, vp, vp, vp]
        lib.drm_walk_table_links.restype = ctypes.c_int
        lib.drm_walk_table_links.argtypes = [ctypes.POINTER(DrmLinkPieces), ctypes.POINTER(DrmLinkForms), i32, vp, vp, vp, i32, vp, vp]
        lib.drm_walk_table_links_backward.restype = ctypes.c_int
        lib.drm_walk_table_links_backward.argtypes = [ctypes.POINTER(DrmLinkPieces), ctypes.POINTER(DrmLinkForms), i32, vp, vp, vp, i32, vp, vp]
        lib.drm_fk_mse_links.restype = ctypes.c_int
        lib.drm_fk_mse_links.argtypes = [wp, vp, vp, ctypes.POINTER(DrmLinkPieces), i32, vp, vp, i64, ctypes.c_uint64, vp, vp, vp, vp, vp]
        lib.drm_fk_mse_scratch_floats.restype = i64
        lib.drm_fk_mse_scratch_floats.argtypes = [i64, i32]
        if lib.drm_abi_version() != ABI_VERSION:
            raise NativeLibraryError("ABI version mismatch: library %d, binding %d" % (lib.drm_abi_version(), ABI_VERSION))
        if lib.drm_walk_sizeof() != ctypes.sizeof(DrmWalk):
            raise NativeLibraryError("struct drm_walk: library %d bytes, binding %d" % (lib.drm_walk_sizeof(), ctypes.sizeof(DrmWalk)))
        if kind == "cpu":
            lib.drm_cpu_set_threads.restype = None
            lib.drm_cpu_set_threads.argtypes = [ctypes.c_int]
            lib.drm_cpu_set_threads(torch.get_num_threads())      # (as many threads as torch's own CPU kernels use)
        _libs[kind] = lib
        return lib


HOSTCALL_PATH = os.path.join(_HERE, "csrc", "drm_hostcall.so")
_hostcall = False      # False: not tried yet;  None: unavailable;  else the module


def hostcall():
    """csrc/drm_hostcall.so — the per-call host work of fk / fk_jacobian / rnea in C++ (a torch extension without device code,
    built by `__graft_entry__.build()`), or None: the Python path below then does the same work, more slowly (11-14 instead of
    ~8 us per eager call).  It holds no kernels and no arithmetic: both paths call the same entry point of the same library.
    DRM_NO_HOSTCALL=1 switches it off (A/B runs)."""
    global _hostcall
    if _hostcall is False:
        _hostcall = None
        if os.environ.get("DRM_NO_HOSTCALL") != "1" and os.path.exists(HOSTCALL_PATH):
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("drm_hostcall", HOSTCALL_PATH)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _hostcall = mod
            except Exception as err:      # (built against another torch: the Python path serves)
                import warnings
                warnings.warn("csrc/drm_hostcall.so could not be loaded (%s); eager calls take the Python host path" % err)
    return _hostcall


def _fn_addr(lib, name: str) -> int:
    """Address of an entry point of `lib`, for drm_hostcall (cached on the library object)."""
    cache = lib.__dict__.setdefault("_drm_addr", {})
    addr = cache.get(name)
    if addr is None:
        addr = cache[name] = ctypes.cast(getattr(lib, name), ctypes.c_void_p).value
    return addr


def _stream_int(device) -> int:
    if device.type == "cpu":
        return 0
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


class KernelUnsupported(RuntimeError):
    """DRM_ERR_UNSUPPORTED: the request is valid but no compiled kernel takes it (e.g. an inertia matrix too large for LDS)."""


def _check(rc: int, lib=None):
    if rc != 0:
        msg = (lib if lib is not None else load_library()).drm_last_error()
        text = "drm_hip call failed (%d): %s" % (rc, msg.decode() if msg else "?")
        raise KernelUnsupported(text) if rc == -2 else RuntimeError(text)


def _lib_of(t, name: str, table=None):
    """The library for the device `t` lives on (`table`: a tensor of the model, which must live on the same kind of device).
    (Per-call path: `is_cuda` reads cost a third of `device.type`.)"""
    try:
        cuda = t.is_cuda
    except AttributeError:
        raise RuntimeError("%s must be a tensor (got %s)" % (name, type(t)))
    if table is not None and table.is_cuda is not cuda:
        raise RuntimeError("%s is on %s but the model's tables are on %s" % (name, t.device, table.device))
    lib = _libs.get("cuda" if cuda else "cpu") if cuda or t.is_cpu else None
    return lib if lib is not None else library_for(t.device)


def _plan_input(t: torch.Tensor, name: str, cols: int) -> torch.Tensor:
    """A plan launches on the caller's OWN buffers (an MPC loop writes the next state into them and replays): the tensor must
    already be what the kernels take — fp32, contiguous, 16-byte aligned — because a silent copy would detach the plan from
    the buffer the caller keeps writing to."""
    out = _dev_f32(t, name, cols)
    if out.data_ptr() != t.data_ptr():
        raise ValueError("%s of a plan must be a contiguous, 16-byte aligned fp32 tensor (got dtype %s, contiguous %s, data_ptr %% 16 = %d): "
                         "a plan launches on the caller's own buffer and cannot work on a copy"
                         % (name, t.dtype, t.is_contiguous(), t.data_ptr() & 15))
    return out


def _dev_f32(t: torch.Tensor, name: str, cols: int) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or t.device.type not in ("cuda", "cpu"):
        raise RuntimeError("%s must be a tensor on a HIP device or on the CPU (got %s)" % (name, getattr(t, "device", type(t))))
    if t.ndim != 2 or t.shape[1] != cols:
        raise ValueError("%s must be [B, %d], got %s" % (name, cols, tuple(t.shape)))
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    t = t.contiguous()
    if t.data_ptr() & 15:   # a row slice of a larger tensor: the fast kernels (and the scratch sizes) assume 16-byte alignment
        t = t.clone()
    return t


def _walk_struct(prog: WalkProgram, ops_f: torch.Tensor, ops_i: torch.Tensor, n_dofs: int) -> DrmWalk:
    """`struct drm_walk` of a walk on the device; the struct of the LAST (table, control words) pair is kept on the program
    (a constant model launches the same pair every call: filling the ~60 fields took 4-5 us of an eager call's ~20)."""
    key = (ops_f.data_ptr(), ops_i.data_ptr(), n_dofs, ops_f.shape[0])
    cached = getattr(prog, "_ws_cache", None)
    if cached is not None:
        if cached[0] == key:
            return cached[1]
        if cached[0][1:] == key[1:] and key[0] and cached[0][0]:
            # a model with learnable links: a NEW table on every call, everything else as before — the struct is read by the C side
            # while the call is made, so the one of the last call takes the new address
            w = cached[1]
            w.ops_f = key[0]
            cached[0] = key      # (the SAME cache entry: prepared calls of a learnable model — which own a copy of the struct — stay valid)
            return w
    w = _walk_struct_build(prog, ops_f, ops_i, n_dofs)
    prog._ws_cache = [key, w]
    return w


def _walk_struct_build(prog: WalkProgram, ops_f: torch.Tensor, ops_i: torch.Tensor, n_dofs: int) -> DrmWalk:
    rows = prog.capacity
    assert ops_f.device.type in ("cuda", "cpu") and ops_i.device == ops_f.device
    assert ops_f.dtype == torch.float32 and ops_f.is_contiguous() and ops_f.shape[0] == rows
    assert ops_i.dtype == torch.int32 and ops_i.is_contiguous() and ops_i.shape[1] == rows
    target_perm = getattr(prog, "_target_perm", None)
    if target_perm is None:   # un-permutation of the last op's frame, looked up once per walk
        target_perm = prog._target_perm = int(prog.ops_i[prog.n_ops - 1, OPI_PERM]) if prog.n_ops else 2
    return fill_walk_struct(DrmWalk, prog, ops_f.data_ptr(), ops_i.data_ptr(), n_dofs, target_perm)


def fill_walk_struct(cls, prog: WalkProgram, ops_f_ptr: int, ops_i_ptr: int, n_dofs: int, target_perm: int):
    """`struct drm_walk` of a WalkProgram whose tables live at the given addresses (device here, host in tests/host_emu)."""
    shape = int(prog.shape) & 0xffffffff   # the top byte carries (P, K, L) of DRM_WALK_ARM_HAND: bit 31 may be set
    w = cls(ops_f_ptr, ops_i_ptr, prog.n_ops, prog.capacity, n_dofs, prog.n_slots, prog.dof_mask, target_perm,
            shape - (1 << 32) if shape >> 31 else shape)
    w.n_segments = prog.n_segments
    for i, v in enumerate(prog.seg_begin):
        w.seg_begin[i] = int(v)
    for i, (lo, cnt) in enumerate(prog.seg_dof):
        w.seg_dof_lo[i], w.seg_dof_cnt[i] = int(lo), int(cnt)
    w.prefix_end = int(prog.prefix_end)
    for i, v in enumerate(prog.seg_leaf_begin):
        w.seg_leaf_begin[i] = int(v)
    for k, v in enumerate(prog.chain_dof1):      # (DRM_WALK_CHAIN_DOFS: what the control words say, as launch arguments)
        w.chain_dof1[k] = int(v)
    w.chain_prismatic = int(prog.chain_prismatic)
    for kind, handle in (getattr(prog, "_special", None) or {}).items():      # (specialize.attach: kernels built for this walk)
        w.special[kind] = handle
    if (getattr(prog, "_special", None) or {}).get(11):      # (specialize.attach_arm_param: the set of learnable blocks it was built for)
        w.reserved0 = int(getattr(prog, "_special_mask", 0))
    ticket = getattr(prog, "_ticket", None)
    if ticket is not None and ticket.is_cuda and ops_f_ptr:
        w.special[WALK_TICKET] = ticket.data_ptr()
    return w


def _ticket(prog: WalkProgram, device) -> None:
    """OPT-IN (DRM_TICKET=1): give the walk its TICKET word (ABI 11, include/drm_hip.h DRM_WALK_TICKET) before its first backward
    launch on a HIP device: one zeroed device word per walk, with which drm_fk_mse / drm_fk_backward reduce their partial sums in the
    block that finishes last instead of in a second launch.  Bit-identical results (tests/test_fk_backward.py) — and SLOWER on an
    MI355X, which is why it is not the default: drm_fk_mse at 16 384 rows 7.4 us as two launches, 43 us with agent-scope fences (each
    writes back / invalidates an XCD's whole L2), 13.2 us with write-through stores + agent-scope loads and no fence, 11.5 us with
    every load of the last block in flight at once (profiles/r06_ticket_ab.txt): store acknowledgement -> ticket -> loads are three
    DEPENDENT trips to memory, each dearer than the ~1.7 us kernel boundary they replace.
    The word belongs to the walk — i.e. to the model: backward launches of ONE model must not overlap on different streams.  Not
    created while a stream is capturing (the allocation would belong to the graph's pool)."""
    have = getattr(prog, "_ticket", None)
    if have is not None and have.device == device:
        return
    if device.type != "cuda" or os.environ.get("DRM_TICKET") != "1" or torch.cuda.is_current_stream_capturing():
        return
    with torch.cuda.device(device):
        prog._ticket = torch.zeros(1, dtype=torch.int32, device=device)
    prog._ws_cache = None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device) -> ctypes.c_void_p:
    """The caller's current HIP stream on `device` as a raw handle (the private accessor, when this torch has it, skips building
    a torch.cuda.Stream object per call)."""
    if device.type == "cpu":
        return None                                   # (libdrm_cpu.so ignores the argument: calls return when the results are written)
    if _raw_stream is not None and device.index is not None:
        return ctypes.c_void_p(_raw_stream(device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device(object):
    """`with torch.cuda.device(d)` only when `d` is not already the current device (the guard costs ~4 us per call)."""
    __slots__ = ("guard",)

    def __init__(self, device):
        self.guard = (None if device.type == "cpu" or device.index is None or torch.cuda.current_device() == device.index
                      else torch.cuda.device(device))

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
        return False


_output_plans = {}


def _outputs(device, *shapes):
    """The output tensors of one call as views of ONE allocation (each starting on a 16-byte boundary): one trip through the
    caching allocator instead of one per tensor.  (The block lives as long as any of its views does.)"""
    plan = _output_plans.get(shapes)
    if plan is None:
        views, total = [], 0
        for sh in shapes:
            views.append((sh, _contiguous_strides(sh), total))
            total += (math.prod(sh) + 3) & ~3          # every tensor starts on a 16-byte boundary
        if len(_output_plans) > 4096:                  # (shapes are (B, ...) tuples: bounded by the batch sizes a process uses)
            _output_plans.clear()
        plan = _output_plans[shapes] = (total, tuple(views))
    flat = torch.empty(plan[0], device=device, dtype=torch.float32)
    return [flat.as_strided(sh, st, off) for sh, st, off in plan[1]]


def _contiguous_strides(shape):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc)
        acc *= d
    return tuple(reversed(st))


def fk(prog: WalkProgram, ops_f, ops_i, q, n_targets: int, n_dofs: int, squeeze: bool = False):
    """pos [B, T, 3], quat [B, T, 4] of the walk's targets ([B, 3], [B, 4] for one target with ``squeeze``: the same
    memory without the T axis, so the single-link API returns it without a slicing op)."""
    lib = _lib_of(q, "q", ops_f)
    fast = hostcall()
    if fast is not None:
        walk = _walk_struct(prog, ops_f, ops_i, n_dofs)
        with _on_device(q.device):
            pos, quat, rc = fast.fk(_fn_addr(lib, "drm_fk"), ctypes.addressof(walk), q, n_dofs, n_targets, squeeze, _stream_int(q.device))
        if rc <= 0:
            if rc:
                _check(rc, lib)
            return pos, quat      # (rc == NOT_CONFORMING: a tensor the kernels do not take as it is — converted below)
    q = _dev_f32(q, "q", n_dofs)
    B = q.shape[0]
    if squeeze and n_targets == 1:
        pos, quat = _outputs(q.device, (B, 3), (B, 4))
    else:
        pos, quat = _outputs(q.device, (B, n_targets, 3), (B, n_targets, 4))
    if B == 0:
        return pos, quat
    walk = _walk_struct(prog, ops_f, ops_i, n_dofs)
    with _on_device(q.device):
        _check(lib.drm_fk(ctypes.byref(walk), q.data_ptr(), B, n_targets, pos.data_ptr(), quat.data_ptr(),
                          _stream(q.device)), lib)
    return pos, quat


def fk_links(prog: WalkProgram, ops_f, ops_i, q, n_targets: int, n_dofs: int):
    """pos [T, B, 3], quat [T, B, 4] of the walk's targets, LINK-major: pos[t] / quat[t] are contiguous [B, 3] / [B, 4] arrays
    (what compute_forward_kinematics_all_links returns per link; the kernel writes a link's 64 poses of a tile as one run)."""
    lib = _lib_of(q, "q", ops_f)
    q = _dev_f32(q, "q", n_dofs)
    B = q.shape[0]
    pos = torch.empty(n_targets, B, 3, device=q.device, dtype=torch.float32)
    quat = torch.empty(n_targets, B, 4, device=q.device, dtype=torch.float32)
    if B == 0:
        return pos, quat
    walk = _walk_struct(prog, ops_f.detach(), ops_i, n_dofs)
    with _on_device(q.device):
        _check(lib.drm_fk_links(ctypes.byref(walk), q.data_ptr(), B, n_targets, pos.data_ptr(), quat.data_ptr(),
                                _stream(q.device)), lib)
    return pos, quat


def fk_fanout(chains, q, n_dofs: int, link_major: bool = False, own=None):
    """FK of 2..4 targets with (nearly) disjoint chains: ``chains`` = [(prog, ops_f, ops_i)] per target.  pos [B, T, 3],
    quat [B, T, 4]; with ``link_major`` pos [T, B, 3], quat [T, B, 4] (every target's poses a contiguous array).
    ``own``: handle of the constant-folded kernel built for exactly THIS ordered set of chains (specialize.attach_fan), written
    into the structs of this call — drm_fk_fanout_links launches it when every chain carries the same handle; None: the library's."""
    lib = _lib_of(q, "q", chains[0][1])
    q = _dev_f32(q, "q", n_dofs)
    B, T = q.shape[0], len(chains)
    pos, quat = _outputs(q.device, (T, B, 3) if link_major else (B, T, 3), (T, B, 4) if link_major else (B, T, 4))
    if B == 0:
        return pos, quat
    # the array of the chains' structs is kept on the first chain's program, per ordered set of chains and own-kernel handle, while
    # the tables stay where they are (building it — four 248-byte structs through ctypes — took 20 us of a call whose kernel takes 4)
    key = (n_dofs, own) + tuple(f.data_ptr() for _, f, _ in chains) + tuple(id(p) for p, _, _ in chains)
    cache = chains[0][0].__dict__.setdefault("_fan_cache", {})
    walks = cache.get(key)
    if walks is None:
        walks = (DrmWalk * T)(*[_walk_struct(p, f, i, n_dofs) for p, f, i in chains])      # (copies of the programs' structs)
        for t in range(T):
            walks[t].special[SPECIAL_FK_FAN_LINKS] = own      # (None: NULL — the fan-out kernel never comes from a shared program)
        if len(cache) > 64:
            cache.clear()
        cache[key] = walks
    with _on_device(q.device):
        _check((lib.drm_fk_fanout_links if link_major else lib.drm_fk_fanout)(
            walks, T, q.data_ptr(), B, pos.data_ptr(), quat.data_ptr(), _stream(q.device)), lib)
    return pos, quat


def fk_jacobian(prog: WalkProgram, ops_f, ops_i, q, n_dofs: int):
    lib = _lib_of(q, "q", ops_f)
    fast = hostcall()
    if fast is not None:
        walk = _walk_struct(prog, ops_f, ops_i, n_dofs)
        with _on_device(q.device):
            pos, quat, lin, ang, rc = fast.fk_jacobian(_fn_addr(lib, "drm_fk_jacobian"), ctypes.addressof(walk), q, n_dofs, _stream_int(q.device))
        if rc <= 0:
            if rc:
                _check(rc, lib)
            return pos, quat, lin, ang
    q = _dev_f32(q, "q", n_dofs)
    B = q.shape[0]
    pos, quat, lin, ang = _outputs(q.device, (B, 3), (B, 4), (B, 3, n_dofs), (B, 3, n_dofs))
    if B == 0:
        return pos, quat, lin, ang
    walk = _walk_struct(prog, ops_f, ops_i, n_dofs)
    with _on_device(q.device):
        _check(lib.drm_fk_jacobian(ctypes.byref(walk), q.data_ptr(), B, pos.data_ptr(), quat.data_ptr(),
                                   lin.data_ptr(), ang.data_ptr(), _stream(q.device)), lib)
    return pos, quat, lin, ang


def rnea(prog: WalkProgram, ops_f, ops_i, q, qd, qdd, include_gravity: bool, use_damping: bool, n_dofs: int):
    lib = _lib_of(q, "q", ops_f)
    fast = hostcall()
    if fast is not None and isinstance(qd, torch.Tensor) and (qdd is None or isinstance(qdd, torch.Tensor)):
        walk = _walk_struct(prog, ops_f, ops_i, n_dofs)
        flags = (RNEA_GRAVITY if include_gravity else 0) | (RNEA_DAMPING if use_damping else 0)
        with _on_device(q.device):
            tau, rc = fast.rnea(_fn_addr(lib, "drm_rnea"), _fn_addr(lib, "drm_rnea_scratch_floats_aligned"), ctypes.addressof(walk), q, qd,
                                qdd, n_dofs, flags, _stream_int(q.device))
        if rc <= 0:
            if rc:
                _check(rc, lib)
            return tau
    q = _dev_f32(q, "q", n_dofs)
    qd = _dev_f32(qd, "qd", n_dofs)
    qdd = _dev_f32(qdd, "qdd", n_dofs) if qdd is not None else None
    B = q.shape[0]
    if qd.shape[0] != B or (qdd is not None and qdd.shape[0] != B):
        raise ValueError("q / qd / qdd batch sizes differ")
    tau = torch.empty(B, n_dofs, device=q.device, dtype=torch.float32)
    if B == 0:
        return tau
    flags = (RNEA_GRAVITY if include_gravity else 0) | (RNEA_DAMPING if use_damping else 0)
    walk = _walk_struct(prog, ops_f, ops_i, n_dofs)
    scratch = _rnea_scratch(lib, walk, B, q.device)
    with _on_device(q.device):
        _check(lib.drm_rnea(ctypes.byref(walk), q.data_ptr(), qd.data_ptr(),
                            qdd.data_ptr() if qdd is not None else None, B, flags, tau.data_ptr(),
                            scratch.data_ptr() if scratch is not None else None, _stream(q.device)), lib)
    return tau


def _rnea_scratch(lib, walk, B, device):
    """The body forces robots with a long segment keep between the two sweeps of drm_rnea (None: not needed)."""
    need = int(lib.drm_rnea_scratch_floats_aligned(ctypes.byref(walk), B))    # (_dev_f32 / _plan_input guarantee aligned pointers)
    return torch.empty(need, device=device, dtype=torch.float32) if need > 0 else None


def rnea_backward(prog: WalkProgram, ops_f, ops_i, q, qd, qdd, grad_tau, include_gravity: bool, use_damping: bool,
                  n_dofs: int, param_mask: int, want_grad_inputs: bool):
    """((grad_q, grad_qd, grad_qdd) or None, grad_ops_f [cap,32] or None) for a loss gradient on the torques."""
    lib = _lib_of(q, "q", ops_f)
    if not prog.slots_unique:
        raise RuntimeError("backward RNEA needs a walk whose branch points own their save slots")
    q = _dev_f32(q, "q", n_dofs)
    qd = _dev_f32(qd, "qd", n_dofs)
    qdd = _dev_f32(qdd, "qdd", n_dofs) if qdd is not None else None
    grad_tau = _dev_f32(grad_tau, "grad_tau", n_dofs)
    B, dev = q.shape[0], q.device
    gin = tuple(torch.empty(B, n_dofs, device=dev, dtype=torch.float32) for _ in range(3)) if want_grad_inputs else None
    grad_ops = torch.empty(prog.capacity, ops_f.shape[1], device=dev, dtype=torch.float32) if param_mask else None
    if gin is None and grad_ops is None:
        return None, None
    scratch = torch.empty(max(1, lib.drm_rnea_backward_scratch_floats(B, prog.capacity, n_dofs, prog.n_slots)),
                          device=dev, dtype=torch.float32)
    flags = (RNEA_GRAVITY if include_gravity else 0) | (RNEA_DAMPING if use_damping else 0)
    walk = _walk_struct(prog, ops_f.detach(), ops_i, n_dofs)
    ptr = lambda t: t.data_ptr() if t is not None else None
    with _on_device(dev):
        _check(lib.drm_rnea_backward(ctypes.byref(walk), q.data_ptr(), qd.data_ptr(), ptr(qdd), B, flags,
                                     grad_tau.data_ptr(), ctypes.c_uint64(param_mask),
                                     ptr(gin[0]) if gin else None, ptr(gin[1]) if gin else None,
                                     ptr(gin[2]) if gin else None, ptr(grad_ops), scratch.data_ptr(), _stream(dev)), lib)
    return gin, grad_ops


def link_rows_torch(params: torch.Tensor) -> torch.Tensor:
    """[L, 20] URDF-level link parameters (rpy, trans, mass, com, inertia_mat, damping) -> [L, 32] link-table rows with plain
    torch ops: the arithmetic of csrc/drm_sample.hpp link_row (= the reference's: R_fixed = (Rz(yaw) Ry(pitch)) Rx(roll),
    rigid_body.py:138-143, spatial_vector_algebra.py:14-53; mcom = com * mass, I_o = I_c + mass * S(com) S(com)^T,
    spatial_vector_algebra.py:321-327).  Differentiable any number of times: the constant snapshot of a model is built with it
    (on the host), and the one-kernel maps below fall back to it when a graph is being created THROUGH their backward
    (create_graph=True: second derivatives with respect to learnable link parameters)."""
    L, dev = params.shape[0], params.device
    rpy, trans, mass, com = params[:, 0:3], params[:, 3:6], params[:, 6:7], params[:, 7:10]
    inertia, damping = params[:, 10:19].reshape(L, 3, 3), params[:, 19:20]
    c, s = torch.cos(rpy), torch.sin(rpy)
    one, zero = torch.ones(L, device=dev, dtype=params.dtype), torch.zeros(L, device=dev, dtype=params.dtype)
    mat = lambda rows: torch.stack([torch.stack(r, dim=-1) for r in rows], dim=-2)
    Rx = mat([[one, zero, zero], [zero, c[:, 0], -s[:, 0]], [zero, s[:, 0], c[:, 0]]])
    Ry = mat([[c[:, 1], zero, s[:, 1]], [zero, one, zero], [-s[:, 1], zero, c[:, 1]]])
    Rz = mat([[c[:, 2], -s[:, 2], zero], [s[:, 2], c[:, 2], zero], [zero, zero, one]])
    F = (Rz @ Ry) @ Rx
    S = mat([[zero, -com[:, 2], com[:, 1]], [com[:, 2], zero, -com[:, 0]], [-com[:, 1], com[:, 0], zero]])
    Io = inertia + mass.reshape(L, 1, 1) * (S @ S.transpose(-2, -1))
    return torch.cat([F.reshape(L, 9), trans, mass, com * mass, Io.reshape(L, 9), damping,
                      torch.zeros(L, 32 - 26, device=dev, dtype=params.dtype)], dim=1)


class LinkRows(torch.autograd.Function):
    """[n, 20] URDF-level link parameters -> [n, 32] link-table rows, with a hand-written backward: two tiny
    kernels instead of ~150 eager torch kernels per training step (the reference rebuilds these quantities with one
    torch op per matrix entry on every call)."""

    @staticmethod
    def forward(ctx, params):
        lib = library_for(params.device)
        params = params.contiguous().to(torch.float32)
        rows = torch.empty(params.shape[0], 32, device=params.device, dtype=torch.float32)
        with _on_device(params.device):
            _check(lib.drm_link_rows(params.data_ptr(), params.shape[0], rows.data_ptr(), _stream(params.device)), lib)
        ctx.save_for_backward(params)
        return rows

    @staticmethod
    def backward(ctx, grad_rows):
        (params,) = ctx.saved_tensors
        if torch.is_grad_enabled():      # create_graph=True: the same map through torch ops, differentiable again (link_rows_torch)
            with torch.enable_grad():
                (grad,) = torch.autograd.grad(link_rows_torch(params), params, grad_rows, create_graph=True)
            return grad
        lib = library_for(params.device)
        grad_rows = grad_rows.contiguous().to(torch.float32)
        grad = torch.empty_like(params)
        with _on_device(params.device):
            _check(lib.drm_link_rows_backward(params.data_ptr(), grad_rows.data_ptr(), params.shape[0], grad.data_ptr(),
                                              _stream(params.device)), lib)
        return grad


class WalkTable(torch.autograd.Function):
    """The walk table of a robot with learnable links in ONE launch, and its derivative in another (drm_walk_table):
    ``pieces`` are the outputs of the links' parameter callables (rot_angles, trans, mass, com, inertia_mat, damping per
    learnable link, 20 floats per link); they are packed, turned into link-table rows and gathered into walk order over
    ``base`` inside the kernel.  The autograd graph holds this one node instead of a cat / stack / index_copy /
    index_select / mul chain, and the backward returns views of one [n_links, 20] gradient."""

    @staticmethod
    def forward(ctx, base, sel, gsign, n_links, *pieces):
        lib = library_for(base.device)
        dev = base.device
        params = torch.cat([p.reshape(-1).to(device=dev, dtype=torch.float32) for p in pieces])
        assert params.numel() == n_links * 20, "20 floats per learnable link"
        ops_f = torch.empty_like(base)
        with _on_device(dev):
            _check(lib.drm_walk_table(params.data_ptr(), n_links, base.data_ptr(), sel.data_ptr(), gsign.data_ptr(),
                                      base.numel(), ops_f.data_ptr(), _stream(dev)), lib)
        ctx.save_for_backward(params, sel, gsign, base, *pieces)
        ctx.n_links, ctx.shapes = n_links, [tuple(p.shape) for p in pieces]
        return ops_f

    @staticmethod
    def backward(ctx, grad_ops_f):
        params, sel, gsign, base = ctx.saved_tensors[:4]
        if torch.is_grad_enabled():
            # create_graph=True: the gradient has to be a differentiable function of the parameters and of grad_ops_f (second
            # derivatives with respect to learnable link parameters).  The same map, pieces -> link rows -> walk order, through
            # torch ops (link_rows_torch) and torch's own double backward; first-order training never comes here.
            pieces = ctx.saved_tensors[4:]
            with torch.enable_grad():
                packed = torch.cat([p.reshape(-1).to(device=base.device, dtype=torch.float32) for p in pieces]).reshape(ctx.n_links, 20)
                rows = link_rows_torch(packed).reshape(-1)
                live = sel >= 0
                table = torch.where(live, rows[sel.clamp_min(0).long()] * gsign, base)
                wanted = [i for i in range(len(pieces)) if ctx.needs_input_grad[4 + i]]
                got = torch.autograd.grad(table, [pieces[i] for i in wanted], grad_ops_f.reshape(table.shape), create_graph=True,
                                          allow_unused=True)
            out = [None] * len(pieces)
            for i, g in zip(wanted, got):
                out[i] = g if g is not None else torch.zeros_like(pieces[i])
            return (None, None, None, None) + tuple(out)
        lib = library_for(params.device)
        g = grad_ops_f.contiguous().to(torch.float32)
        grad = torch.empty_like(params)
        with _on_device(params.device):
            _check(lib.drm_walk_table_backward(params.data_ptr(), ctx.n_links, g.data_ptr(), sel.data_ptr(), gsign.data_ptr(),
                                               g.numel(), grad.data_ptr(), _stream(params.device)), lib)
        out, off = [], 0
        for i, shape in enumerate(ctx.shapes):
            n = 1
            for d in shape:
                n *= d
            out.append(grad[off:off + n].reshape(shape) if ctx.needs_input_grad[4 + i] else None)
            off += n
        return (None, None, None, None) + tuple(out)


PIECE_NAMES = ("rot_angles", "trans", "mass", "com", "inertia_mat", "damping")
PIECE_SIZES = (3, 3, 1, 3, 9, 1)
PIECE_OFFSETS = (0, 3, 6, 7, 10, 19)


class LinkSourcePlan(object):
    """What WalkTableLinks needs to know about the learnable links of a model besides the tensors that change: per link and piece the
    FORM it is stored in (FORM_*), the form's constant and — for a form other than plain — the module whose raw parameter the tensor
    is (its torch arithmetic is what second derivatives go through); ``fixed``: the tensor of every piece that is a CONSTANT of the
    model (six per link; None where a parameter module supplies the piece on every call — the LIVE pieces, in this order the
    ``sources`` of WalkTableLinks)."""

    def __init__(self, entries, fixed=None):
        self.entries = entries                      # [n_links][6] of (form, constant, module or None)
        self.n_links = len(entries)
        self.forms = (DrmLinkForms * self.n_links)()
        for l, link in enumerate(entries):
            for j, name in ((2, "mass"), (4, "inertia_mat"), (5, "damping")):
                setattr(self.forms[l], name, link[j][0])
                setattr(self.forms[l], name + "_c", float(link[j][1]))
        sizes = [6 if (j == 4 and link[j][0] != FORM_PLAIN) else PIECE_SIZES[j] for link in entries for j in range(6)]
        self.fixed = list(fixed) if fixed is not None else [None] * (6 * self.n_links)
        self.live = [i for i, t in enumerate(self.fixed) if t is None]
        self.sizes = [sizes[i] for i in self.live]
        self._links = (DrmLinkPieces * self.n_links)()
        for i, t in enumerate(self.fixed):
            if t is not None:
                if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != sizes[i]:
                    raise ValueError("constant piece %d of a learnable link: %d float32 values expected" % (i, sizes[i]))
                setattr(self._links[i // 6], PIECE_NAMES[i % 6], t.data_ptr())
        self._key = None
        self.checked = None        # data_ptr()s of the live sources WalkTableLinks last found on `device`, float32, contiguous, of the right sizes
        self.device = next((t.device for t in self.fixed if t is not None), None)

    def pieces(self, sources):
        """The drm_link_pieces array with the live pieces at these tensors (rewritten only when one of them has moved)."""
        key = tuple([t.data_ptr() for t in sources])
        if key != self._key:
            for i, ptr in zip(self.live, key):
                setattr(self._links[i // 6], PIECE_NAMES[i % 6], ptr)
            self._key = key
        return self._links

    def torch_pieces(self, sources):
        """All 6 x n_links pieces as differentiable functions of the live sources (the modules' own arithmetic on the raw tensors)."""
        out = list(self.fixed)
        for i, t in zip(self.live, sources):
            form, _, module = self.entries[i // 6][i % 6]
            out[i] = t if form == FORM_PLAIN else torch.func.functional_call(module, {"l": t}, ())
        return out


class WalkTableLinks(torch.autograd.Function):
    """The walk table of a robot with learnable links straight from the links' parameter tensors WHERE THEY LIE and in the form their
    modules store them (drm_walk_table_links, ABI 13), and its derivative back to those tensors: one launch each, no cat of the 6 x
    n_links pieces, no torch kernels for the modules the kernels know (PositiveScalar and the l[6] inertia-matrix modules of
    rigid_body_params).  ``sources``: the LIVE pieces of ``plan`` (LinkSourcePlan) — the raw parameter for a piece whose form is not
    plain, the module's output otherwise; the constant pieces of the learnable links are part of the plan."""

    @staticmethod
    def forward(ctx, base, sel, gsign, plan, *sources):
        lib = library_for(base.device)
        dev = base.device
        if tuple([t.data_ptr() for t in sources]) == plan.checked:      # (the tensors of the last call, where they were: checked then)
            held = sources
        else:
            held, as_given = [], True
            for t, size in zip(sources, plan.sizes):
                if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
                    t, as_given = t.detach().to(device=dev, dtype=torch.float32).contiguous(), False
                if t.numel() != size:
                    raise ValueError("a piece of a learnable link has %d elements, not %d" % (t.numel(), size))
                held.append(t)
            plan.checked = tuple([t.data_ptr() for t in sources]) if as_given and dev == plan.device else None
        ops_f = torch.empty_like(base)
        with _on_device(dev):
            _check(lib.drm_walk_table_links(plan.pieces(held), plan.forms, plan.n_links, base.data_ptr(), sel.data_ptr(), gsign.data_ptr(),
                                            base.numel(), ops_f.data_ptr(), _stream(dev)), lib)
        ctx.save_for_backward(sel, gsign, base, *sources)
        ctx.plan = plan
        return ops_f

    @staticmethod
    def backward(ctx, grad_ops_f):
        sel, gsign, base = ctx.saved_tensors[:3]
        sources, plan = ctx.saved_tensors[3:], ctx.plan
        if torch.is_grad_enabled():
            # create_graph=True: sources -> pieces -> link rows -> walk order through torch ops (WalkTable.backward's twin)
            with torch.enable_grad():
                pieces = plan.torch_pieces(sources)
                packed = torch.cat([p.reshape(-1).to(device=base.device, dtype=torch.float32) for p in pieces]).reshape(plan.n_links, 20)
                rows = link_rows_torch(packed).reshape(-1)
                table = torch.where(sel >= 0, rows[sel.clamp_min(0).long()] * gsign, base)
                wanted = [i for i in range(len(sources)) if ctx.needs_input_grad[4 + i]]
                got = torch.autograd.grad(table, [sources[i] for i in wanted], grad_ops_f.reshape(table.shape), create_graph=True,
                                          allow_unused=True)
            out = [None] * len(sources)
            for i, g in zip(wanted, got):
                out[i] = g if g is not None else torch.zeros_like(sources[i])
            return (None, None, None, None) + tuple(out)
        dev = base.device
        lib = library_for(dev)
        plain = tuple([t.data_ptr() for t in sources]) == plan.checked
        held = sources if plain else [t if (t.device == dev and t.dtype == torch.float32 and t.is_contiguous()) else
                                      t.detach().to(device=dev, dtype=torch.float32).contiguous() for t in sources]
        g = grad_ops_f.contiguous().to(torch.float32)
        grad = torch.empty(plan.n_links * 20, device=dev, dtype=torch.float32)
        with _on_device(dev):
            _check(lib.drm_walk_table_links_backward(plan.pieces(held), plan.forms, plan.n_links, g.data_ptr(), sel.data_ptr(),
                                                     gsign.data_ptr(), g.numel(), grad.data_ptr(), _stream(dev)), lib)
        out = []
        needs = ctx.needs_input_grad
        for k, (i, t, size) in enumerate(zip(plan.live, sources, plan.sizes)):
            if not needs[4 + k]:
                out.append(None)
                continue
            at = (i // 6) * 20 + PIECE_OFFSETS[i % 6]
            piece = grad[at:at + size].view(t.shape)
            out.append(piece if plain or (t.device == dev and t.dtype == torch.float32) else piece.to(device=t.device, dtype=t.dtype))
        return (None, None, None, None) + tuple(out)


def forward_dynamics(prog: WalkProgram, ops_f, ops_i, q, qd, f, include_gravity: bool, use_damping: bool, n_dofs: int):
    """qdd [B, n] produced by the joint torques f in state (q, qd)."""
    lib = _lib_of(q, "q", ops_f)
    fast = hostcall()
    if fast is not None and isinstance(qd, torch.Tensor) and isinstance(f, torch.Tensor):
        walk = _walk_struct(prog, ops_f, ops_i, n_dofs)      # (reads the table's address only)
        flags = (RNEA_GRAVITY if include_gravity else 0) | (RNEA_DAMPING if use_damping else 0)
        with _on_device(q.device):
            qdd, rc = fast.forward_dynamics(_fn_addr(lib, "drm_forward_dynamics"), _fn_addr(lib, "drm_forward_dynamics_scratch_floats_aligned"),
                                            ctypes.addressof(walk), q, qd, f, n_dofs, flags, _stream_int(q.device))
        if rc <= 0:
            if rc:
                _check(rc, lib)
            return qdd
    q, qd, f = _dev_f32(q, "q", n_dofs), _dev_f32(qd, "qd", n_dofs), _dev_f32(f, "f", n_dofs)
    B = q.shape[0]
    if qd.shape[0] != B or f.shape[0] != B:
        raise ValueError("q / qd / f batch sizes differ")
    qdd = torch.empty(B, n_dofs, device=q.device, dtype=torch.float32)
    if B == 0:
        return qdd
    flags = (RNEA_GRAVITY if include_gravity else 0) | (RNEA_DAMPING if use_damping else 0)
    walk = _walk_struct(prog, ops_f.detach(), ops_i, n_dofs)
    # the per-link records of the articulated-body sweeps, when the launch keeps them in HBM
    need = int(lib.drm_forward_dynamics_scratch_floats_aligned(ctypes.byref(walk), B))
    scratch = torch.empty(need, device=q.device, dtype=torch.float32) if need > 0 else None
    with _on_device(q.device):
        _check(lib.drm_forward_dynamics(ctypes.byref(walk), q.data_ptr(), qd.data_ptr(), f.data_ptr(), B, flags,
                                        qdd.data_ptr(), scratch.data_ptr() if scratch is not None else None,
                                        _stream(q.device)), lib)
    return qdd


def crba(prog: WalkProgram, ops_f, ops_i, q, n_dofs: int):
    """H [B, n, n] joint-space inertia matrix."""
    lib = _lib_of(q, "q", ops_f)
    fast = hostcall()
    if fast is not None:
        walk = _walk_struct(prog, ops_f, ops_i, n_dofs)
        with _on_device(q.device):
            H, rc = fast.crba(_fn_addr(lib, "drm_crba"), _fn_addr(lib, "drm_crba_scratch_floats_aligned"), ctypes.addressof(walk), q, n_dofs,
                              _stream_int(q.device))
        if rc <= 0:
            if rc:
                _check(rc, lib)
            return H
    q = _dev_f32(q, "q", n_dofs)
    B = q.shape[0]
    H = torch.empty(B, n_dofs, n_dofs, device=q.device, dtype=torch.float32)
    if B == 0:
        return H
    walk = _walk_struct(prog, ops_f.detach(), ops_i, n_dofs)
    # robots with a long segment collect the lower triangle of H in scratch before its rows are written
    need = int(lib.drm_crba_scratch_floats_aligned(ctypes.byref(walk), B))
    scratch = torch.empty(need, device=q.device, dtype=torch.float32) if need > 0 else None
    with _on_device(q.device):
        _check(lib.drm_crba(ctypes.byref(walk), q.data_ptr(), B, H.data_ptr(),
                            scratch.data_ptr() if scratch is not None else None, _stream(q.device)), lib)
    return H


def fk_backward(prog: WalkProgram, ops_f, ops_i, q, grad_pos, n_targets: int, n_dofs: int, param_mask: int,
                want_grad_q: bool, grad_rot=None):
    """(grad_q [B,n] or None, grad_ops_f [cap,32] or None) for a loss gradient on the target positions (grad_pos
    [B,T,3]) and, optionally, on the target rotation matrices (grad_rot [B,T,3,3])."""
    lib = _lib_of(q, "q", ops_f)
    if not prog.slots_unique:
        raise RuntimeError("backward FK needs a walk whose branch points own their save slots (more than %d "
                           "branch points in this tree)" % prog.n_slots)
    q = _dev_f32(q, "q", n_dofs)
    B = q.shape[0]
    grad_pos = _dev_f32(grad_pos.reshape(B, n_targets * 3), "grad_pos", n_targets * 3)
    if grad_rot is not None:
        grad_rot = _dev_f32(grad_rot.reshape(B, n_targets * 9), "grad_rot", n_targets * 9)
    dev = q.device
    grad_q = torch.empty(B, n_dofs, device=dev, dtype=torch.float32) if want_grad_q else None
    grad_ops = torch.empty(prog.capacity, ops_f.shape[1], device=dev, dtype=torch.float32) if param_mask else None
    if grad_q is None and grad_ops is None:
        return None, None
    scratch = torch.empty(max(1, lib.drm_fk_backward_scratch_floats(B, prog.capacity)), device=dev, dtype=torch.float32)
    if param_mask:
        _ticket(prog, dev)
    walk = _walk_struct(prog, ops_f.detach(), ops_i, n_dofs)
    with _on_device(dev):
        _check(lib.drm_fk_backward(ctypes.byref(walk), q.data_ptr(), B, n_targets, grad_pos.data_ptr(),
                                   grad_rot.data_ptr() if grad_rot is not None else None,
                                   ctypes.c_uint64(param_mask), grad_q.data_ptr() if want_grad_q else None,
                                   grad_ops.data_ptr() if param_mask else None, scratch.data_ptr(), _stream(dev)), lib)
    return grad_q, grad_ops


def fk_mse(prog: WalkProgram, ops_f, ops_i, q, target, n_dofs: int, param_mask: int, want_grad_q: bool):
    """(loss [], grad_q [B, n] or None, grad_ops_f [cap, 32] or None) of loss = mean((pos(q) - target)^2) for the walk's target
    (drm_fk_mse: forward kinematics, loss and backward in one pass).  Raises KernelUnsupported for walks / batches the fused
    kernel does not take (the caller composes fk + mse_loss then)."""
    lib = _lib_of(q, "q", ops_f)
    q = _dev_f32(q, "q", n_dofs)
    target = _dev_f32(target, "target", 3)
    B, dev = q.shape[0], q.device
    if target.shape[0] != B:
        raise ValueError("q and target batch sizes differ")
    loss = torch.empty((), device=dev, dtype=torch.float32)
    grad_q = torch.empty(B, n_dofs, device=dev, dtype=torch.float32) if want_grad_q else None
    grad_ops = torch.empty(prog.capacity, ops_f.shape[1], device=dev, dtype=torch.float32) if param_mask else None
    scratch = torch.empty(max(1, lib.drm_fk_mse_scratch_floats(B, prog.capacity)), device=dev, dtype=torch.float32)
    _ticket(prog, dev)
    walk = _walk_struct(prog, ops_f.detach(), ops_i, n_dofs)
    with _on_device(dev):
        _check(lib.drm_fk_mse(ctypes.byref(walk), q.data_ptr(), target.data_ptr(), B, ctypes.c_uint64(param_mask), loss.data_ptr(),
                              grad_q.data_ptr() if want_grad_q else None, grad_ops.data_ptr() if param_mask else None,
                              scratch.data_ptr(), _stream(dev)), lib)
    return loss, grad_q, grad_ops


def fk_mse_links(prog: WalkProgram, base, ops_i, sel, gsign, pieces, q, target, n_dofs: int, param_mask: int, want_grad_q: bool):
    """(loss [], grad_q [B, n] or None, grad_params [n_links, 20]) of loss = mean((pos(q) - target)^2) for a walk WITH learnable
    links, from the links' parameters to the gradients with respect to them (drm_fk_mse_links, ABI 12): ``base`` is the table of the
    constant links in walk order, ``sel`` / ``gsign`` as for WalkTable, ``pieces`` the six parameter tensors per learnable link
    (rot_angles, trans, mass, com, inertia_mat, damping) where they lie.  Two launches for what WalkTable -> fk_mse -> WalkTable's
    backward do in five (plus a cat).  Raises KernelUnsupported as fk_mse does."""
    lib = _lib_of(q, "q", base)
    q = _dev_f32(q, "q", n_dofs)
    target = _dev_f32(target, "target", 3)
    B, dev = q.shape[0], q.device
    if target.shape[0] != B:
        raise ValueError("q and target batch sizes differ")
    n_links, rest = divmod(len(pieces), 6)
    if rest or not 1 <= n_links <= FK_MSE_MAX_LINKS:
        raise KernelUnsupported("drm_fk_mse_links takes 1 .. %d learnable links" % FK_MSE_MAX_LINKS)
    links = (DrmLinkPieces * n_links)()
    keep = []
    for l in range(n_links):
        for j, (name, size) in enumerate((("rot_angles", 3), ("trans", 3), ("mass", 1), ("com", 3), ("inertia_mat", 9), ("damping", 1))):
            t = pieces[l * 6 + j]
            if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != size:
                t = t.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
                if t.numel() != size:
                    raise ValueError("%s of a learnable link has %d elements, not %d" % (name, t.numel(), size))
            keep.append(t)
            setattr(links[l], name, t.data_ptr())
    loss = torch.empty((), device=dev, dtype=torch.float32)
    grad_q = torch.empty(B, n_dofs, device=dev, dtype=torch.float32) if want_grad_q else None
    grad_params = torch.empty(n_links, 20, device=dev, dtype=torch.float32)
    scratch = torch.empty(max(1, lib.drm_fk_mse_scratch_floats(B, prog.capacity)), device=dev, dtype=torch.float32)
    table = base.detach().reshape(prog.capacity, -1)
    key = (table.data_ptr(), ops_i.data_ptr(), n_dofs)
    cached = getattr(prog, "_ws_links", None)      # (its own slot: _walk_struct keeps the struct of the LIVE table of this program)
    if cached is None or cached[0] != key:
        cached = prog._ws_links = (key, _walk_struct_build(prog, table, ops_i, n_dofs))
    walk = cached[1]
    with _on_device(dev):
        _check(lib.drm_fk_mse_links(ctypes.byref(walk), sel.data_ptr(), gsign.data_ptr(), links, n_links, q.data_ptr(), target.data_ptr(), B,
                                    ctypes.c_uint64(param_mask), loss.data_ptr(), grad_q.data_ptr() if want_grad_q else None,
                                    grad_params.data_ptr(), scratch.data_ptr(), _stream(dev)), lib)
    del keep
    return loss, grad_q, grad_params


def fk_jacobian_backward(prog: WalkProgram, ops_f, ops_i, q, grad_pos, grad_lin, grad_ang, n_dofs: int,
                         param_mask: int, want_grad_q: bool, grad_rot=None):
    """(grad_q [B,n] or None, grad_ops_f [cap,32] or None) for loss gradients on the Jacobian (and, optionally, the
    target position) of the chain walk ``prog``."""
    lib = _lib_of(q, "q", ops_f)
    q = _dev_f32(q, "q", n_dofs)
    B, dev = q.shape[0], q.device
    grad_lin = _dev_f32(grad_lin.reshape(B, 3 * n_dofs), "grad_lin_jac", 3 * n_dofs)
    grad_ang = _dev_f32(grad_ang.reshape(B, 3 * n_dofs), "grad_ang_jac", 3 * n_dofs)
    grad_pos = _dev_f32(grad_pos.reshape(B, 3), "grad_pos", 3) if grad_pos is not None else None
    grad_rot = _dev_f32(grad_rot.reshape(B, 9), "grad_rot", 9) if grad_rot is not None else None
    grad_q = torch.empty(B, n_dofs, device=dev, dtype=torch.float32) if want_grad_q else None
    grad_ops = torch.empty(prog.capacity, ops_f.shape[1], device=dev, dtype=torch.float32) if param_mask else None
    if grad_q is None and grad_ops is None:
        return None, None
    scratch = torch.empty(max(1, lib.drm_fk_backward_scratch_floats(B, prog.capacity)), device=dev, dtype=torch.float32)
    walk = _walk_struct(prog, ops_f.detach(), ops_i, n_dofs)
    with _on_device(dev):
        _check(lib.drm_fk_jacobian_backward(ctypes.byref(walk), q.data_ptr(), B,
                                            grad_pos.data_ptr() if grad_pos is not None else None,
                                            grad_rot.data_ptr() if grad_rot is not None else None, grad_lin.data_ptr(),
                                            grad_ang.data_ptr(), ctypes.c_uint64(param_mask),
                                            grad_q.data_ptr() if want_grad_q else None,
                                            grad_ops.data_ptr() if param_mask else None, scratch.data_ptr(), _stream(dev)), lib)
    return grad_q, grad_ops


def _plan_stream(stream, device):
    if device.type == "cpu":
        return None
    return ctypes.c_void_p((stream if stream is not None else torch.cuda.current_stream(device)).cuda_stream)


class FkJacobianPlan(object):
    """A prepared drm_fk_jacobian launch on fixed buffers (no per-call allocation or argument marshalling).

    Used by bench.py and by callers that evaluate the same batch shape repeatedly (MPC loops); works
    under HIP graph capture because it only enqueues one kernel on the current stream.
    """

    def __init__(self, prog: WalkProgram, ops_f, ops_i, q, n_dofs: int, want_pose: bool = True):
        self._lib = _lib_of(q, "q", ops_f)
        self.q = _plan_input(q, "q", n_dofs)
        B = self.q.shape[0]
        dev = self.q.device
        self.pos = torch.empty(B, 3, device=dev) if want_pose else None
        self.quat = torch.empty(B, 4, device=dev) if want_pose else None
        self.lin = torch.empty(B, 3, n_dofs, device=dev)
        self.ang = torch.empty(B, 3, n_dofs, device=dev)
        self._keep = (ops_f, ops_i)
        self._walk = _walk_struct(prog, ops_f.detach(), ops_i, n_dofs)
        self._args = (ctypes.byref(self._walk), self.q.data_ptr(), B,
                      self.pos.data_ptr() if want_pose else None, self.quat.data_ptr() if want_pose else None,
                      self.lin.data_ptr(), self.ang.data_ptr())
        self.batch = B
        self.device = dev

    def launch(self, stream=None):
        rc = self._lib.drm_fk_jacobian(*self._args, _plan_stream(stream, self.device))
        if rc != 0:
            _check(rc, self._lib)

    def launch_many(self, k: int, stream=None):
        """k back-to-back launches enqueued from C++ (csrc/drm_hostcall.cpp repeat_fk_jacobian: ~2 us of host time per launch, less
        than the kernel takes, so the device never waits for the host) — what a compiled caller's loop does; the Python loop over
        `launch` when the host-call module is not built."""
        fast = hostcall()
        if fast is None or self.pos is None:
            for _ in range(k):
                self.launch(stream)
            return
        st = 0 if self.device.type == "cpu" else (stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream
        rc = fast.repeat_fk_jacobian(_fn_addr(self._lib, "drm_fk_jacobian"), ctypes.addressof(self._walk), self.q.data_ptr(), self.batch,
                                     self.pos.data_ptr(), self.quat.data_ptr(), self.lin.data_ptr(), self.ang.data_ptr(), st, int(k))
        if rc != 0:
            _check(rc, self._lib)

    def outputs(self):
        return self.pos, self.quat, self.lin, self.ang


def fk_rnea(tree, chain, target_op: int, q, qd, qdd, include_gravity: bool, use_damping: bool, n_dofs: int):
    """(tau [B, n], pos [B, 3], quat [B, 4]) of ONE eager drm_fk_rnea call through the C++ host path, or None when that path is not
    there / the inputs are not what the kernels take as they are (the caller then goes through FkInverseDynamicsPlan)."""
    fast = hostcall()
    if fast is None or not hasattr(fast, "fk_rnea") or not isinstance(qd, torch.Tensor) or not (qdd is None or isinstance(qdd, torch.Tensor)):
        return None
    lib = _lib_of(q, "q", tree[1])
    wt, wc = _walk_struct(tree[0], tree[1], tree[2], n_dofs), _walk_struct(chain[0], chain[1], chain[2], n_dofs)
    flags = (RNEA_GRAVITY if include_gravity else 0) | (RNEA_DAMPING if use_damping else 0)
    with _on_device(q.device):
        tau, pos, quat, rc = fast.fk_rnea(_fn_addr(lib, "drm_fk_rnea"), _fn_addr(lib, "drm_rnea_scratch_floats_aligned"), ctypes.addressof(wt),
                                          ctypes.addressof(wc), int(target_op), q, qd, qdd, n_dofs, flags, _stream_int(q.device))
    if rc > 0:
        return None
    if rc:
        _check(rc, lib)
    return tau, pos, quat


class FkInverseDynamicsPlan(object):
    """A prepared drm_fk_rnea launch on fixed buffers: q, qd, qdd -> tau and the pose (pos, quat) of one link.  For a
    serial 7-DoF arm whose last link is the target (Franka Panda, KUKA iiwa) that is ONE fused kernel; BASELINE
    configuration 3 (bench.py --config 3) runs it on every GPU's shard."""

    def __init__(self, tree, chain, target_op: int, q, qd, qdd, include_gravity: bool, use_damping: bool, n_dofs: int,
                 outputs=None, put=None):
        # tree / chain: (WalkProgram, ops_f, ops_i) of the whole-tree walk and of the root -> link walk
        # put: a DrmPut (distributed.PeerGather.put()) — every launch also writes its rows into the peers' gathered arrays
        # (drm_fk_rnea_put: the one-sided gather of a batch sharded over the GPUs of a node)
        self._lib = _lib_of(q, "q", tree[1])
        self.q, self.qd = _plan_input(q, "q", n_dofs), _plan_input(qd, "qd", n_dofs)
        self.qdd = _plan_input(qdd, "qdd", n_dofs) if qdd is not None else None
        B, dev = self.q.shape[0], self.q.device
        if self.qd.shape[0] != B or (self.qdd is not None and self.qdd.shape[0] != B):
            raise ValueError("q / qd / qdd batch sizes differ")
        if outputs is None:
            self.tau = torch.empty(B, n_dofs, device=dev)
            self.pos = torch.empty(B, 3, device=dev)
            self.quat = torch.empty(B, 4, device=dev)
        else:   # caller-owned buffers (e.g. three views of ONE allocation that a collective then sends as it stands)
            self.tau, self.pos, self.quat = outputs
            for t, cols in ((self.tau, n_dofs), (self.pos, 3), (self.quat, 4)):
                if t.shape != (B, cols) or t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
                    raise ValueError("outputs must be contiguous float32 [B, %d] tensors on the inputs' device" % cols)
                if t.data_ptr() & 15:     # (the scratch below is sized by the *_aligned query, which assumes 16-byte aligned pointers)
                    raise ValueError("outputs must start on a 16-byte boundary (slice the shared buffer at multiples of 4 floats)")
        self._keep = (tree[1], tree[2], chain[1], chain[2])
        self._tree = _walk_struct(tree[0], tree[1].detach(), tree[2], n_dofs)
        self._chain = _walk_struct(chain[0], chain[1].detach(), chain[2], n_dofs)
        flags = (RNEA_GRAVITY if include_gravity else 0) | (RNEA_DAMPING if use_damping else 0)
        self._scratch = _rnea_scratch(self._lib, self._tree, B, dev)
        self._args = (ctypes.byref(self._tree), ctypes.byref(self._chain), int(target_op), self.q.data_ptr(),
                      self.qd.data_ptr(), self.qdd.data_ptr() if self.qdd is not None else None, B, flags,
                      self.tau.data_ptr(), self.pos.data_ptr(), self.quat.data_ptr(),
                      self._scratch.data_ptr() if self._scratch is not None else None)
        self.batch, self.device = B, dev
        self._put = put

    def launch(self, stream=None):
        if self._put is not None:
            rc = self._lib.drm_fk_rnea_put(*self._args, ctypes.byref(self._put), _plan_stream(stream, self.device))
        else:
            rc = self._lib.drm_fk_rnea(*self._args, _plan_stream(stream, self.device))
        if rc != 0:
            _check(rc, self._lib)

    def outputs(self):
        return self.tau, self.pos, self.quat


class InverseDynamicsPlan(object):
    """A prepared drm_rnea launch on fixed buffers (q, qd, qdd in, tau out): the allocation-free, graph-capturable form of
    compute_inverse_dynamics for loops that evaluate the same batch shape repeatedly (update q / qd / qdd in place)."""

    def __init__(self, prog: WalkProgram, ops_f, ops_i, q, qd, qdd, include_gravity: bool, use_damping: bool, n_dofs: int):
        self._lib = _lib_of(q, "q", ops_f)
        self.q, self.qd = _plan_input(q, "q", n_dofs), _plan_input(qd, "qd", n_dofs)
        self.qdd = _plan_input(qdd, "qdd", n_dofs) if qdd is not None else None
        B, dev = self.q.shape[0], self.q.device
        if self.qd.shape[0] != B or (self.qdd is not None and self.qdd.shape[0] != B):
            raise ValueError("q / qd / qdd batch sizes differ")
        self.tau = torch.empty(B, n_dofs, device=dev)
        self._keep = (ops_f, ops_i)
        self._walk = _walk_struct(prog, ops_f.detach(), ops_i, n_dofs)
        flags = (RNEA_GRAVITY if include_gravity else 0) | (RNEA_DAMPING if use_damping else 0)
        self._scratch = _rnea_scratch(self._lib, self._walk, B, dev)
        self._args = (ctypes.byref(self._walk), self.q.data_ptr(), self.qd.data_ptr(),
                      self.qdd.data_ptr() if self.qdd is not None else None, B, flags, self.tau.data_ptr(),
                      self._scratch.data_ptr() if self._scratch is not None else None)
        self.batch, self.device = B, dev

    def launch(self, stream=None):
        rc = self._lib.drm_rnea(*self._args, _plan_stream(stream, self.device))
        if rc != 0:
            _check(rc, self._lib)

    def outputs(self):
        return (self.tau,)
