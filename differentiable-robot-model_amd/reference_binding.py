"""INTEGRATION.md, stub B: the ctypes binding a maintainer of the reference would add to call `libdrm_hip.so` from the
reference's OWN `DifferentiableRobotModel`, as an executable module.

It needs nothing of this package's model class: its input is the list of per-link dicts that the reference's
`URDFRobotModel.get_body_parameters_from_urdf(i, link)` returns (urdf_utils.py:28-126; this package's `urdf_utils`
returns the same dicts, bit for bit) plus each link's parent name.  From those it

  1. flattens the tree once (`flatten.build_robot_spec` / `build_walk`: parent indices, DoF columns, axis folding),
  2. builds the per-link constant table with plain torch ops, as the reference computes them on every call
     (rigid_body.py:138-143 R_fixed = (Rz Ry) Rx; spatial_vector_algebra.py:321-327 mcom = m com, I_o = I_c + m S(c) S(c)^T),
  3. gathers it into walk order and calls the C ABI (`include/drm_hip.h`) on the current torch HIP stream.

In the reference, `robot_model.py` would construct one `HipBinding` in `DifferentiableRobotModel.__init__`
(robot_model.py:94-137) and replace the bodies of `compute_forward_kinematics` (223-248), `compute_endeffector_jacobian`
(626-667) and `compute_inverse_dynamics` (305-375) by calls to `fk` / `jacobian` / `inverse_dynamics` below; the
`tensor_check` decorator, asserts and signatures stay as they are.  tests/test_reference_binding.py runs exactly that wiring.
"""
import ctypes
import os
from typing import Optional, Sequence

import numpy as np
import torch

from .flatten import MAX_SEGMENTS, OPF_STRIDE, OPI_PERM, build_robot_spec, build_walk, identity_table_row, virtual_row_constants

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "csrc", "libdrm_hip.so")
RNEA_GRAVITY, RNEA_DAMPING = 1, 2
ABI_VERSION = 13      # DRM_ABI_VERSION of include/drm_hip.h this mirror of struct drm_walk follows


class DrmWalk(ctypes.Structure):
    """Mirror of `struct drm_walk` (include/drm_hip.h)."""
    _fields_ = [("ops_f", ctypes.c_void_p), ("ops_i", ctypes.c_void_p), ("n_ops", ctypes.c_int32),
                ("capacity", ctypes.c_int32), ("n_dofs", ctypes.c_int32), ("n_slots", ctypes.c_int32),
                ("dof_mask", ctypes.c_uint64), ("target_perm", ctypes.c_int32), ("shape", ctypes.c_int32),
                ("n_segments", ctypes.c_int32), ("seg_begin", ctypes.c_int32 * (MAX_SEGMENTS + 1)),
                ("seg_dof_lo", ctypes.c_int32 * MAX_SEGMENTS), ("seg_dof_cnt", ctypes.c_int32 * MAX_SEGMENTS),
                ("prefix_end", ctypes.c_int32), ("seg_leaf_begin", ctypes.c_int32 * (MAX_SEGMENTS + 1)),
                ("chain_dof1", ctypes.c_uint8 * 16), ("chain_prismatic", ctypes.c_uint32), ("reserved0", ctypes.c_uint32),
                ("special", ctypes.c_void_p * 16)]      # per-robot straight-line kernels of this walk (specialize.py), or NULL


def link_table(body_params: Sequence[dict], device, spec=None) -> torch.Tensor:
    """[L + 1 (+ virtual rows), 32] per-link constants F(9) t(3) m mcom(3) I_o(9) damping, row L = the identity op."""
    L = len(body_params)
    f = lambda key, shape, default=None: torch.stack([
        (bp[key] if bp[key] is not None else default).detach().reshape(shape).to(device=device, dtype=torch.float32)
        for bp in body_params])
    rpy, trans = f("rot_angles", (3,)), f("trans", (3,))
    mass, com, inertia = f("mass", (1,)), f("com", (3,)), f("inertia_mat", (3, 3))
    damping = f("joint_damping", (1,), torch.zeros(1))
    c, s = torch.cos(rpy), torch.sin(rpy)
    one, zero = torch.ones(L, device=device), torch.zeros(L, device=device)
    mat = lambda rows: torch.stack([torch.stack(r, dim=-1) for r in rows], dim=-2)
    Rx = mat([[one, zero, zero], [zero, c[:, 0], -s[:, 0]], [zero, s[:, 0], c[:, 0]]])
    Ry = mat([[c[:, 1], zero, s[:, 1]], [zero, one, zero], [-s[:, 1], zero, c[:, 1]]])
    Rz = mat([[c[:, 2], -s[:, 2], zero], [s[:, 2], c[:, 2], zero], [zero, zero, one]])
    F = (Rz @ Ry) @ Rx                                                     # rigid_body.py:138-143
    S = mat([[zero, -com[:, 2], com[:, 1]], [com[:, 2], zero, -com[:, 0]], [-com[:, 1], com[:, 0], zero]])
    Io = inertia + mass.reshape(L, 1, 1) * (S @ S.transpose(-2, -1))       # spatial_vector_algebra.py:324-327
    rows = torch.cat([F.reshape(L, 9), trans, mass, com * mass, Io.reshape(L, 9), damping,
                      torch.zeros(L, OPF_STRIDE - 26, device=device)], dim=1)
    table = torch.cat([rows, torch.from_numpy(identity_table_row()).to(device).reshape(1, OPF_STRIDE)], dim=0)
    if spec is not None and spec.skew.any():   # joints about a general axis: two virtual rows each (flatten.build_walk)
        links, Ra = virtual_row_constants(spec)
        Ra_t = torch.from_numpy(np.ascontiguousarray(Ra)).to(device)
        r = table[torch.tensor(links, device=device)]
        n = len(links)
        z = lambda k: torch.zeros(n, k, device=device)
        row_a = torch.cat([(r[:, 0:9].reshape(n, 3, 3) @ Ra_t).reshape(n, 9), r[:, 9:12], z(13), r[:, 25:26], z(OPF_STRIDE - 26)], 1)
        row_b = torch.cat([Ra_t.transpose(1, 2).reshape(n, 9), z(3), r[:, 12:25], z(OPF_STRIDE - 25)], 1)
        table = torch.cat([table, torch.stack([row_a, row_b], 1).reshape(2 * n, OPF_STRIDE)], 0)
    return table


class HipBinding(object):
    """`libdrm_hip.so` behind the three hot methods of the reference's model."""

    def __init__(self, body_params: Sequence[dict], parent_names: Sequence[Optional[str]], device, library: str = None,
                 reference_compat: bool = True):
        import torch  # noqa: F401  (before the library: both share one HIP runtime)
        self.device = torch.device(device)
        self.spec = build_robot_spec(body_params, parent_names, reference_compat=reference_compat)
        self.names = {bp["link_name"]: i for i, bp in enumerate(body_params)}
        self.n = self.spec.n_dofs
        self.table = link_table(body_params, self.device, self.spec)
        self._walks = {}
        self._lib = None
        self._library = library or os.environ.get("DRM_HIP_LIBRARY", _LIB_PATH)

    # -- plumbing ----------------------------------------------------------------------------------------------
    @property
    def lib(self):
        if self._lib is None:
            self._lib = ctypes.CDLL(self._library)
            self._lib.drm_last_error.restype = ctypes.c_char_p
            if self._lib.drm_abi_version() != ABI_VERSION or self._lib.drm_walk_sizeof() != ctypes.sizeof(DrmWalk):
                raise RuntimeError("libdrm_hip.so: ABI %d with a %d-byte drm_walk, this binding is written for ABI %d / %d bytes" % (
                    self._lib.drm_abi_version(), self._lib.drm_walk_sizeof(), ABI_VERSION, ctypes.sizeof(DrmWalk)))
        return self._lib

    def _check(self, rc):
        if rc:
            raise RuntimeError("drm_hip call failed (%d): %s" % (rc, self.lib.drm_last_error().decode()))

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def walk(self, key, **kw):
        """(struct drm_walk, the tensors it points to) of a target chain / the whole tree, built once."""
        if key not in self._walks:
            prog = build_walk(self.spec, **kw)
            gather = torch.from_numpy(prog.gather.reshape(-1)).to(self.device)     # walk order + axis folding
            gsign = torch.from_numpy(prog.gsign.reshape(-1)).to(self.device)       # its exact +-1 factors
            ops_f = (self.table.reshape(-1)[gather] * gsign).reshape(prog.capacity, OPF_STRIDE).contiguous()
            ops_i = torch.from_numpy(prog.ops_i_dev).to(self.device).contiguous()
            perm = int(prog.ops_i[prog.n_ops - 1, OPI_PERM]) if prog.n_ops else 2
            w = DrmWalk(ops_f.data_ptr(), ops_i.data_ptr(), prog.n_ops, prog.capacity, self.n, prog.n_slots, prog.dof_mask,
                        perm, prog.shape)
            w.n_segments = prog.n_segments
            for i, v in enumerate(prog.seg_begin):
                w.seg_begin[i] = int(v)
            for i, (lo, cnt) in enumerate(prog.seg_dof):
                w.seg_dof_lo[i], w.seg_dof_cnt[i] = int(lo), int(cnt)
            w.prefix_end = int(prog.prefix_end)
            for i, v in enumerate(prog.seg_leaf_begin):
                w.seg_leaf_begin[i] = int(v)
            for k, v in enumerate(prog.chain_dof1):          # (serial chains: the DoF columns as launch arguments)
                w.chain_dof1[k] = int(v)
            w.chain_prismatic = int(prog.chain_prismatic)
            self._walks[key] = (w, ops_f, ops_i, prog)
        return self._walks[key]

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def _in(self, t):
        assert t.device.type == "cuda" and t.ndim == 2 and t.shape[1] == self.n
        t = t.to(torch.float32).contiguous()
        if t.data_ptr() & 15:    # a row slice (q[1:] of a [B, 7] tensor starts 28 bytes in): drm_rnea / drm_forward_dynamics take the
            t = t.clone()        # aligned fast kernels of arm / arm + hand walks only on 16-byte aligned pointers (backend._dev_f32)
        return t

    # -- the three methods -------------------------------------------------------------------------------------
    def fk(self, q, link_name):
        """compute_forward_kinematics (robot_model.py:223-248): (pos [B,3], quat_xyzw [B,4])."""
        idx = self.names[link_name]
        q = self._in(q)
        B = q.shape[0]
        pos, quat = q.new_zeros(B, 3), q.new_zeros(B, 4)
        if idx == 0:                      # the root link: identity pose
            quat[:, 3] = 1.0
            return pos, quat
        w = self.walk(("chain", idx), targets=[idx])[0]
        self._check(self.lib.drm_fk(ctypes.byref(w), self._p(q), ctypes.c_int64(B), ctypes.c_int32(1), self._p(pos), self._p(quat),
                                    self._stream()))
        return pos, quat

    def fk_all_links(self, q):
        """compute_forward_kinematics_all_links (robot_model.py:197-221): {link name: (pos [B,3], quat_xyzw [B,4])} — one
        drm_fk_links launch; its link-major outputs make every dict entry a contiguous array."""
        q = self._in(q)
        B = q.shape[0]
        order = [i for i in self.spec.preorder() if i != 0]          # targets in walk order
        pos, quat = q.new_empty(len(order), B, 3), q.new_empty(len(order), B, 4)
        w = self.walk(("links",), targets=order)[0]
        self._check(self.lib.drm_fk_links(ctypes.byref(w), self._p(q), ctypes.c_int64(B), ctypes.c_int32(len(order)), self._p(pos),
                                          self._p(quat), self._stream()))
        root_quat = q.new_zeros(B, 4)
        root_quat[:, 3] = 1.0
        out = {name: (q.new_zeros(B, 3), root_quat) for name, i in self.names.items() if i == 0}
        slot = {i: k for k, i in enumerate(order)}
        out.update({name: (pos[slot[i]], quat[slot[i]]) for name, i in self.names.items() if i != 0})
        return out

    def jacobian(self, q, link_name):
        """compute_endeffector_jacobian (robot_model.py:626-667): (lin_jac [B,3,n], ang_jac [B,3,n])."""
        idx = self.names[link_name]
        q = self._in(q)
        B = q.shape[0]
        lin, ang = q.new_zeros(B, 3, self.n), q.new_zeros(B, 3, self.n)
        if idx == 0:
            return lin, ang
        w = self.walk(("chain", idx), targets=[idx])[0]
        self._check(self.lib.drm_fk_jacobian(ctypes.byref(w), self._p(q), ctypes.c_int64(B), None, None, self._p(lin), self._p(ang),
                                             self._stream()))
        return lin, ang

    def inverse_dynamics(self, q, qd, qdd, include_gravity=True, use_damping=True):
        """compute_inverse_dynamics (robot_model.py:305-375): tau [B,n]."""
        q, qd, qdd = self._in(q), self._in(qd), self._in(qdd)
        B = q.shape[0]
        tau = q.new_empty(B, self.n)
        w = self.walk(("tree",), whole_tree=True)[0]
        flags = (RNEA_GRAVITY if include_gravity else 0) | (RNEA_DAMPING if use_damping else 0)
        # (robots with a long segment keep their per-link body forces in caller-owned scratch between the two sweeps)
        self.lib.drm_rnea_scratch_floats.restype = ctypes.c_int64
        need = int(self.lib.drm_rnea_scratch_floats(ctypes.byref(w), ctypes.c_int64(B)))
        scratch = q.new_empty(need) if need > 0 else None
        self._check(self.lib.drm_rnea(ctypes.byref(w), self._p(q), self._p(qd), self._p(qdd), ctypes.c_int64(B), ctypes.c_int32(flags),
                                      self._p(tau), self._p(scratch) if scratch is not None else None, self._stream()))
        return tau


def bind_reference_model(model, **kw) -> HipBinding:
    """A HipBinding for an instance of the reference's `DifferentiableRobotModel` (or this package's): reads the per-link
    dicts through the model's own URDF loader, exactly as robot_model.py:107-137 does."""
    um = model._urdf_model
    params = [um.get_body_parameters_from_urdf(i, link) for i, link in enumerate(um.robot.links)]
    parents = [None] + [um.get_name_of_parent_body(link.name) for link in um.robot.links[1:]]
    return HipBinding(params, parents, model._device, **kw)
