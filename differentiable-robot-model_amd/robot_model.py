"""Drop-in ``DifferentiableRobotModel`` for the FK / Jacobian / RNEA path on MI355X.

Keeps the reference's Python method surface (reference
``differentiable_robot_model/robot_model.py:87-790``): same constructor, same
method names / argument meaning / return shapes, the ``tensor_check`` batching
rules (robot_model.py:25-84), ``AssertionError`` / ``AttributeError`` /
``KeyError`` behaviour, and the learnable-parameter mechanism
(robot_model.py:669-713).  Underneath, the reference's per-link Python loops
over tiny torch ops are replaced by:

  host (this file + flatten.py)   URDF -> RobotSpec -> depth-first walk tables, once;
  device (csrc/*.hip)             one fused hand-written HIP kernel per API call,
                                  reached through the C ABI of include/drm_hip.h.

There is NO CPU compute path: the compute methods raise if the model does not
live on a HIP device or if the native library is missing.
"""
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import backend
from .flatten import (OPF_STRIDE, RobotSpec, WalkProgram, build_robot_spec, build_walk, fold_link_table,
                      foldable_links, identity_table_row, virtual_row_constants)
from .autograd import (_FkJacobian, _FkMse, _FkMseLinks, _FkPositions, _ForwardDynamics, _InverseDynamics, _MassMatrix,  # noqa: F401
                       _quat_grad_to_rot)
from .rigid_body import DifferentiableRigidBody, LinkPose, LinkVelocity
from .urdf_utils import URDFRobotModel

robot_description_folder = os.path.join(os.path.dirname(os.path.abspath(__file__)), "robot_data")
_WARNED_URDFS = set()      # URDF files whose "joints modelled as the URDF says" warning has been given in this process


def tensor_check(function):
    """Input normalisation shared by all public compute methods.

    Behaviour of the reference decorator of the same name (robot_model.py:25-84):
    every ``torch.Tensor`` argument must be on the model's device type, have
    ndim 1 or 2 and share one batch shape; 1-D inputs are promoted to ``[1, n]``
    and Tensor / tuple results are stripped back (dict results pass through).
    """

    @dataclass
    class BatchInfo:
        shape: torch.Size = torch.Size([])
        init: bool = False

    def preprocess(arg, obj, info):
        if type(arg) is torch.Tensor:
            assert arg.device.type == obj._device.type, f"Input argument of different device as module: {arg}"
            assert arg.ndim in [1, 2], "Input tensors must have ndim of 1 or 2."
            if info.init:
                assert info.shape == arg.shape[:-1], "Batch size mismatch between input tensors."
            else:
                info.init = True
                info.shape = arg.shape[:-1]
            if len(info.shape) == 0:
                return arg.unsqueeze(0)
        return arg

    def postprocess(arg, info):
        if type(arg) is torch.Tensor and info.init and len(info.shape) == 0:
            return arg[0, ...]
        return arg

    def wrapper(self, *args, **kwargs):
        info = BatchInfo()
        p_args = [preprocess(a, self, info) for a in args]
        if kwargs:
            kwargs = {k: preprocess(v, self, info) for k, v in kwargs.items()}
        ret = function(self, *p_args, **kwargs)
        if not (info.init and len(info.shape) == 0):     # batched inputs: results pass through as they are
            return ret
        if type(ret) is torch.Tensor:
            return postprocess(ret, info)
        if type(ret) is tuple:
            return tuple(postprocess(r, info) for r in ret)
        return ret

    wrapper.__name__ = getattr(function, "__name__", "wrapper")
    wrapper.__doc__ = function.__doc__
    return wrapper


@dataclass
class _DeviceWalk:
    program: WalkProgram
    ops_i: torch.Tensor                 # int32 [8, cap] (field-major) on the model device
    gather: torch.Tensor                # int64 [cap * 32] flat indices into the [L+1, 32] link table
    gsign: torch.Tensor                 # float32 [cap * 32] +-1 factors of the gathered entries
    static_ops_f: Optional[torch.Tensor] = None
    folded: bool = False                # a dynamics walk without the foldable links, on the folded link table
    fold_key: Optional[tuple] = None    # ... folded for this set of kept (learnable) links (robot_model._fold_mask)
    learnable_plan: Optional[tuple] = None   # (learnable links, constant walk table, row selector) of _ops_f_learnable


class DifferentiableRobotModel(torch.nn.Module):
    """Batched FK / geometric Jacobian / RNEA on MI355X behind the reference API."""

    def __init__(self, urdf_path: str, name="", device=None, reference_compat: bool = True):
        """Same signature and defaults as the reference's constructor (robot_model.py:94-104): ``device=None`` is the CPU, and
        every non-fixed joint is an axis-aligned revolute joint (robot_model.py:122-126, rigid_body.py:149-154), so a model
        built the reference's way returns the reference's numbers for every URDF the reference accepts.
        ``reference_compat=False`` opts into what the URDF says instead: prismatic joints slide and joints turn about their
        true (possibly skew) axis."""
        super().__init__()
        self.name = name
        self._reference_compat = bool(reference_compat)
        if device is None:
            device = "cpu"      # robot_model.py:100-104 (pass device="cuda" for the HIP kernels; nothing falls back either way)
        self._device = torch.device(device)
        if self._device.type == "cuda" and self._device.index is None:
            self._device = torch.device("cuda", torch.cuda.current_device())

        self._urdf_model = URDFRobotModel(urdf_path=urdf_path, device=self._device)
        self._bodies = torch.nn.ModuleList()
        self._n_dofs = 0
        self._controlled_joints = []
        self._name_to_idx_map = dict()

        body_params, parent_names = [], []
        # the joint is folded into its child link, as in the reference (robot_model.py:107-131)
        for i, link in enumerate(self._urdf_model.robot.links):
            params = self._urdf_model.get_body_parameters_from_urdf(i, link)
            body = DifferentiableRigidBody(rigid_body_params=params, device=self._device)
            if params["joint_type"] != "fixed":
                body.joint_idx = self._n_dofs
                self._n_dofs += 1
                self._controlled_joints.append(i)
            self._bodies.append(body)
            self._name_to_idx_map[body.name] = i
            body_params.append(params)
            parent_names.append(None if i == 0 else self._urdf_model.get_name_of_parent_body(link.name))
        for i, body in enumerate(self._bodies):
            body._attach(self, i)
            if i == 0:
                continue
            parent = self._bodies[self._name_to_idx_map[parent_names[i]]]
            body.set_parent(parent)
            parent.add_child(body)
        self._zero1 = torch.zeros(1, device=self._device)
        self._table_links = os.environ.get("DRM_TABLE_LINKS", "1") != "0"        # (0: the table of a learnable model through a cat of the modules' outputs + drm_walk_table; A/B switch)
        self._source_plan = None
        self._learnable_version = 0                 # counts make_link_param_learnable: what is cached per set of learnable parameters looks at it
        self._fk_mse_links = os.environ.get("DRM_FK_MSE_LINKS", "1") != "0"      # (0: fk_mse_loss composes WalkTable + drm_fk_mse; A/B switch)
        self._kin_state = None   # (q, qd) of the last update_kinematic_state
        self._kin_cache = {}

        self._spec: RobotSpec = build_robot_spec(body_params, parent_names, reference_compat=self._reference_compat)
        if self._reference_compat:
            # the reference's joint model is the default; tell a user whose URDF has a sliding joint what that means (once per
            # URDF file and process: models are built in loops and on every rank), pointing at the USER's call
            sliding = [b.name for i, b in enumerate(self._bodies) if body_params[i]["joint_type"] == "prismatic"]
            if sliding and os.path.abspath(urdf_path) not in _WARNED_URDFS:
                _WARNED_URDFS.add(os.path.abspath(urdf_path))
                import warnings
                warnings.warn(
                    "%s: prismatic joint(s) of %s modelled as REVOLUTE joints, as the reference does (robot_model.py:122-126) "
                    "— upstream's numbers, not the mechanism's.  Pass reference_compat=False for joints that slide (and for "
                    "joint axes that are not +-x / y / z)." % (os.path.basename(urdf_path), ", ".join(sliding)),
                    stacklevel=2 if type(self) is DifferentiableRobotModel else 3)
        self._learnable = set()          # {(link_idx, parameter_name)}
        self._walks: Dict[tuple, _DeviceWalk] = {}
        self._chain_walks: Dict[int, _DeviceWalk] = {}          # link index -> _chain_walk's answer while nothing is learnable
        self._dyn_walk: Optional[_DeviceWalk] = None            # _dynamics_walk's answer while nothing is learnable
        self._fanout_plans: Dict[tuple, Optional[list]] = {}
        self._fan_handles: Dict[tuple, int] = {}               # fan-out plan key -> the constant-folded kernel of THAT ordered set
        # prepared eager calls of a constant model (csrc/drm_hostcall.cpp FastCall): link name -> (call, walk program, its struct cache)
        self._fast_fk: Dict[str, tuple] = {}
        self._fast_jac: Dict[str, tuple] = {}
        self._fast_id: Optional[tuple] = None
        self._fast_crba: Optional[tuple] = None
        self._fast_fd: Optional[tuple] = None
        self._fast_fkid: Dict[str, tuple] = {}                  # link name -> (call, dynamics walk program, its cache, chain program, its cache)
        self._stream_arg = (lambda: 0) if self._device.type != "cuda" or backend._raw_stream is None else \
            (lambda raw=backend._raw_stream, i=self._device.index: raw(i))
        self._own_kernels: Optional[str] = None                 # None: DRM_SPECIALIZE decides (default "auto"); "off" / "auto" / "build"
        self._static_table: Optional[torch.Tensor] = None      # snapshot of all rows (constants)
        self._static_folded: Dict[tuple, torch.Tensor] = {}    # ... with the links of a fold mask folded into their parents
        self._fold_masks: Dict[tuple, np.ndarray] = {}          # kept (learnable) links -> foldable_links(spec, keep)
        self._root_pose = None                                  # the root link's identity (pos [1,3], quat [1,4]), made on first use
        self._learnable_links: Optional[torch.Tensor] = None   # link indices whose rows are rebuilt per call

    # ------------------------------------------------------------------ copies
    # copy.deepcopy(model) works as it does for the reference's plain nn.Module (a ground-truth copy, a target network): parameters,
    # constants and the set of learnable parameters are copied; everything DERIVED — walks with their launch structs and kernel handles,
    # prepared calls, the source plan of the learnable links — is left behind and rebuilt by the copy on first use.
    _DERIVED = {"_walks": dict, "_chain_walks": dict, "_dyn_walk": lambda: None, "_fanout_plans": dict, "_fan_handles": dict,
                "_fast_fk": dict, "_fast_jac": dict, "_fast_id": lambda: None, "_fast_crba": lambda: None, "_fast_fd": lambda: None,
                "_fast_fkid": dict, "_source_plan": lambda: None, "_kin_cache": dict, "_kin_state": lambda: None,
                "_stream_arg": lambda: None, "_arm_specialized": lambda: False}
    _DERIVED_LAZY = ("_learnable_sorted", "_skew_any", "_dyn_walk_learnable", "_body_names", "_all_link_idxs", "_fast_links", "_fk_links_plans")

    def __getstate__(self):
        state = self.__dict__.copy()
        for name, make in self._DERIVED.items():
            if name in state:
                state[name] = make()
        for name in self._DERIVED_LAZY:
            state.pop(name, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._stream_arg = (lambda: 0) if self._device.type != "cuda" or backend._raw_stream is None else \
            (lambda raw=backend._raw_stream, i=self._device.index: raw(i))
        self._learnable_version = self.__dict__.get("_learnable_version", 0) + 1
        for i, body in enumerate(self._bodies):
            body._attach(self, i)       # (a body's pose / vel ask ITS model)

    # ------------------------------------------------------------------ constants
    def _link_rows(self, link_idxs, device=None) -> torch.Tensor:
        """[len(link_idxs), OPF_STRIDE] float32 rows of per-link constants on the model device.

        Built with torch ops from the bodies' parameter callables, so gradients reach learnable
        parametrisations.  The arithmetic mirrors the reference:
        R_fixed = (Rz(yaw) @ Ry(pitch)) @ Rx(roll)  (rigid_body.py:138-143, spatial_vector_algebra.py:14-53),
        mcom = com * mass, I_o = I_c + mass * S(com) S(com)^T  (spatial_vector_algebra.py:321-327).
        """
        dev = self._device if device is None else device
        rows = backend.link_rows_torch(self._link_params(link_idxs, device=dev).to(torch.float32))
        if OPF_STRIDE != rows.shape[1]:
            rows = torch.cat([rows, torch.zeros(rows.shape[0], OPF_STRIDE - rows.shape[1], device=dev)], dim=1)
        return rows.to(torch.float32)

    def _link_table(self, fold_key: Optional[tuple] = None) -> torch.Tensor:
        """[L + 1, OPF_STRIDE] table of per-link constants (row L = the identity op used to pad walks); ``fold_key``: with
        the links of that fold mask folded into their parents (_dynamics_walk).

        With learnable parameters only the rows of the links that carry them are rebuilt (differentiably) on
        top of a cached snapshot of the constant rows: the per-step host work is O(#learnable links), not O(L).
        """
        if self._static_table is None:
            with torch.no_grad():
                # the constant snapshot is computed on the HOST whatever the model's device (round 5): R_fixed = Rz Ry Rx through the
                # host's cos / sin is the reference's own CPU result bit for bit (a device's differ in the last bit for general
                # angles), the table — and with it the key of a robot's own kernels, specialize.build — is the same on every machine
                ident = torch.from_numpy(identity_table_row()).reshape(1, OPF_STRIDE)
                base = torch.cat([self._link_rows(range(len(self._bodies)), device=torch.device("cpu")), ident], dim=0)
                self._static_table = self._with_virtual_rows(base).to(self._device)
        if not self._learnable:
            return self._static_table if fold_key is None else self._with_virtual_rows(self._static_rows(fold_key))
        links = sorted({link for link, _ in self._learnable})
        if self._learnable_links is None or self._learnable_links.numel() != len(links):
            self._learnable_links = torch.tensor(links, dtype=torch.int64, device=self._device)
        if self._device.type == "cuda":
            # the rows of the learnable links through the fused parameter kernels (backend.LinkRows)
            rows = backend.LinkRows.apply(self._link_params(links))
        else:
            rows = self._link_rows(links)   # host-side tests of the table construction
        return self._with_virtual_rows(self._static_rows(fold_key).index_copy(0, self._learnable_links, rows))

    def _with_virtual_rows(self, base: torch.Tensor) -> torch.Tensor:
        """Append the two virtual rows of every link whose joint axis is not +-x / y / z (flatten.build_walk):
        A = (F R_a, t, massless, damping) — the joint itself, turning about +z — and B = (R_a^T, 0, mass, mcom, I_o, 0),
        the fixed step back into the link's own frame.  Differentiable in the link's row (torch ops)."""
        links, Ra = virtual_row_constants(self._spec)
        if not links:
            return base
        Ra_t = torch.from_numpy(np.ascontiguousarray(Ra)).to(base.device)
        rows = base[torch.tensor(links, device=base.device)]
        S = len(links)
        zeros = lambda k: torch.zeros(S, k, device=base.device)
        FA = (rows[:, 0:9].reshape(S, 3, 3) @ Ra_t).reshape(S, 9)
        row_a = torch.cat([FA, rows[:, 9:12], zeros(13), rows[:, 25:26], zeros(OPF_STRIDE - 26)], dim=1)
        row_b = torch.cat([Ra_t.transpose(1, 2).reshape(S, 9), zeros(3), rows[:, 12:25], zeros(OPF_STRIDE - 25)], dim=1)
        return torch.cat([base, torch.stack([row_a, row_b], dim=1).reshape(2 * S, OPF_STRIDE)], dim=0)

    def _link_params(self, link_idxs, device=None) -> torch.Tensor:
        """[len(link_idxs), 20] rpy, trans, mass, com, inertia_mat, damping of the given links (include/drm_hip.h
        drm_link_rows), read from the bodies' parameter callables so that gradients reach learnable modules."""
        dev = self._device if device is None else device
        zero1 = torch.zeros(1, device=dev)
        rows = []
        for i in link_idxs:
            b = self._bodies[i]
            damping = b.joint_damping()
            rows.append(torch.cat([t.reshape(-1).to(dev) for t in (
                b.rot_angles(), b.trans(), b.inertia.mass(), b.inertia.com(), b.inertia.inertia_mat(),
                damping if damping is not None else zero1)]))
        return torch.stack(rows)

    def _get_walk(self, key, targets=None, whole_tree=False, folded=False, fold_key=None) -> _DeviceWalk:
        dw = self._walks.get(key)
        if dw is None:
            prog = build_walk(self._spec, targets=targets, whole_tree=whole_tree, drop_folded=folded and whole_tree,
                              fold=self._fold_masks[fold_key] if folded else None)
            dw = _DeviceWalk(
                program=prog,
                ops_i=torch.from_numpy(prog.ops_i_dev).to(self._device).contiguous(),
                gather=torch.from_numpy(prog.gather.reshape(-1)).to(self._device),
                gsign=torch.from_numpy(prog.gsign.reshape(-1)).to(self._device),
                folded=folded, fold_key=fold_key if folded else None,
            )
            self._walks[key] = dw
        return dw

    def _own_kernel_mode(self) -> str:
        """How this model comes by its OWN kernels (its constants / its tree folded into the instruction stream, specialize.py):

          "auto"   (default, round 6) a code object that is ALREADY built — shipped next to the library (csrc/special_cache/, the
                   robots of the package), or in the run-time cache of an earlier `specialize()` — is attached on first use; nothing
                   is ever compiled on a call path, and a miss keeps the library's kernels (logged once per robot at INFO);
          "build"  compile what is missing with hipcc (2-7 s per kernel, cached): `model.specialize()`, `model.own_kernels = "build"`,
                   DRM_SPECIALIZE=1 (or =tune) in the environment;
          "off"    the library's kernels only: `model.own_kernels = "off"` or DRM_SPECIALIZE=0.
        Either way the results are the same to a few ulp (tests/test_specialize.py runs both)."""
        if self._own_kernels is not None:
            return self._own_kernels
        env = os.environ.get("DRM_SPECIALIZE")
        if env == "0":
            return "off"
        if env in ("1", "tune") or getattr(self, "_arm_specialized", False):
            return "build"
        return "auto"

    @property
    def own_kernels(self) -> Optional[str]:
        """None (DRM_SPECIALIZE decides; "auto" when unset), "off", "auto" or "build" — see _own_kernel_mode."""
        return self._own_kernels

    @own_kernels.setter
    def own_kernels(self, mode: Optional[str]) -> None:
        if mode not in (None, "off", "auto", "build"):
            raise ValueError('own_kernels must be None, "off", "auto" or "build"')
        self._own_kernels = mode
        # whatever was attached under the previous setting goes: the next call looks again under the new one
        for dw in list(self._walks.values()) + [w for plan in self._fanout_plans.values() if plan for w in plan]:
            prog = dw.program
            prog._special, prog._special_const, prog._special_tried, prog._ws_cache = {}, False, False, None
            prog._arm_special_failed = False
            prog.__dict__.pop("_fan_cache", None)
        self._fan_handles.clear()
        self._dyn_walk = None
        self.__dict__.pop("_own_kernel_missed", None)
        if mode in (None, "auto", "build"):
            for key, plan in self._fanout_plans.items():
                if plan:
                    self._fan_special(key, plan)

    def _own_kernel_miss(self, what: str, err: Exception) -> None:
        """One INFO line per model and kind of kernel when the default (`auto`) finds nothing built."""
        seen = self.__dict__.setdefault("_own_kernel_missed", set())
        if what not in seen:
            seen.add(what)
            import logging
            logging.getLogger("differentiable_robot_model_amd").info(
                "%s: no pre-built own kernel for %s (%s); the library's kernels serve it — model.specialize() builds one",
                self.name or "robot", what, str(err)[:160])

    def specialize(self, force: bool = False, tune: bool = False):
        """Build (hipcc, 3-7 s, cached) and attach this robot's OWN straight-line dynamics kernels — inverse dynamics and its
        reverse mode, the inertia matrix, forward dynamics (specialize.py, csrc/drm_static.hpp): for robots whose tree is none of the
        shapes the library ships straight-line kernels for (a mobile manipulator such as Fetch) the loop-structured kernels stop being
        the only choice (Fetch at 2^20 rows: 233 -> 70 us, 982 -> 206 us, 655 -> 170 us, 615 -> 150 us — round 5: a CONSTANT model's
        kernels also carry its walk table as compile-time constants, so products with the robot's zeros and ones are gone).  DRM_SPECIALIZE=1 in the environment does it on first use.  Returns True when a kernel was
        attached, False when the robot already runs a compiled straight-line kernel (7-DoF arms, arm + hand, hands).  Models with
        learnable link parameters specialise their full walk as well (the kernel reads the same table).  `force`: build the
        robot's own kernels even when the library has a straight-line kernel for its shape (measurements, tools/probe_special.py).
        `tune`: build them for ANY robot that is not a plain 7-DoF arm, time every entry point both ways on this device and keep the
        faster of the two per entry point (specialize.tune; returns its report): the library's kernels for "an arm that carries a
        hand" are generic in (P, K, L), a small robot's own walk can beat them (Panda with gripper: inertia matrix 133 -> 86 us,
        forward dynamics 167 -> 125, inverse dynamics 70 -> 58, its reverse mode 197 -> 149 at 2^20 rows)."""
        from . import specialize as sp
        if self._device.type != "cuda":
            raise RuntimeError("per-robot kernels are HIP code objects: the model must live on a HIP device (it is on %s)" % self._device)
        dw = self._dynamics_walk()
        from .flatten import SHAPE_ARM_CHAIN, SHAPE_ARM_HAND, SHAPE_FINGERS
        if not self._learnable:
            # constant models: the kernels that bake the robot's CONSTANTS (round 5) are built where they are first needed — the fused
            # FK + RNEA kernel of an arm's target link, the fan-out FK kernel of a set of fingertips (compute_forward_kinematics_links)
            self._arm_specialized = True
            for key, plan in self._fanout_plans.items():
                if plan:
                    self._fan_special(key, plan)
        if tune:
            if dw.program.shape & SHAPE_ARM_CHAIN or dw.program.n_ops > sp.MAX_STATIC_OPS:
                return {}
            dw.program._special = {k: h for k, h in (getattr(dw.program, "_special", None) or {}).items() if k not in sp.KERNELS}
            sp.attach(dw.program, self._spec, self._n_dofs, self._const_table(dw), ignore_tuned=True)
            return sp.tune(dw.program, self._ops_f(dw).detach(), dw.ops_i, self._n_dofs)      # (remembered: specialize.tuned_kinds)
        if not force and sp.arm_qualifies(dw.program, self._n_dofs):
            # a serial 7-DoF arm (Panda, iiwa): its OWN kernels too, of a different kind — the library's streaming walk with this
            # robot's constants folded into the instruction stream (specialize.attach_arm, csrc/drm_arm_stream.hpp).  Inverse
            # dynamics now; the fused FK + RNEA kernel of a target link when plan_fk_and_inverse_dynamics /
            # compute_fk_and_inverse_dynamics first meets it.  Constant models only (the kernels do not read the table).
            if self._learnable:      # the reverse-mode kernel of this set of learnable blocks (the constant blocks folded in)
                kin, dyn = self._learnable_block_masks(dw)
                sp.attach_arm_param(dw.program, self._ops_f(dw).detach().cpu().numpy(), self._n_dofs, kin, dyn)
                return True
            self._arm_specialized = True
            sp.attach_arm(dw.program, self._ops_f(dw).detach().cpu().numpy(), self._n_dofs)
            return True
        if not force and dw.program.shape & (SHAPE_ARM_CHAIN | SHAPE_ARM_HAND | SHAPE_FINGERS):
            return False
        sp.attach(dw.program, self._spec, self._n_dofs, self._const_table(dw))
        return True

    def _const_table(self, dw: "_DeviceWalk"):
        """The walk's table as a host array for kernels that carry it as compile-time constants (round 5) — None for a model with
        learnable parameters (its kernels keep reading the table: it changes every step)."""
        if self._learnable:
            return None
        return self._ops_f(dw).detach().cpu().numpy()

    def _arm_fused_special(self, tree: "_DeviceWalk", chain: "_DeviceWalk") -> None:
        """After specialize() (or under DRM_SPECIALIZE=1) on a serial 7-DoF arm: build and attach, once per target chain, the
        constant-folded fused FK + RNEA kernel of this (dynamics walk, chain walk) pair."""
        if self._learnable or self._device.type != "cuda":
            return
        mode = self._own_kernel_mode()
        if mode == "off":
            return
        from . import specialize as sp
        have = (getattr(tree.program, "_special", None) or {}).get(sp.SPECIAL_FK_RNEA_ARM)
        if have is not None and have == (getattr(chain.program, "_special", None) or {}).get(sp.SPECIAL_FK_RNEA_ARM):
            return
        if getattr(chain.program, "_arm_special_failed", False):
            return
        if not (sp.arm_qualifies(tree.program, self._n_dofs) and sp.arm_qualifies(chain.program, self._n_dofs) and chain.program.n_ops == 8):
            return
        try:
            sp.attach_arm(tree.program, self._ops_f(tree).detach().cpu().numpy(), self._n_dofs,
                          chain.program, self._ops_f(chain).detach().cpu().numpy(), cached_only=mode == "auto")
            if (tree.program._special or {}).get(sp.SPECIAL_FK_RNEA_ARM) is None:      # (auto: not built yet)
                chain.program._arm_special_failed = True
                self._own_kernel_miss("the fused FK + inverse-dynamics launch", sp.CacheMiss("; ".join(tree.program._special_missed)))
        except sp.SpecializeError:
            if getattr(self, "_arm_specialized", False):
                raise
            chain.program._arm_special_failed = True      # (environment opt-in on a machine without hipcc: the library's kernels)

    def _fold_key(self) -> tuple:
        """The links that must stay ops of their own (those with learnable parameters), as the key of a fold mask."""
        key = tuple(sorted({link for link, _ in self._learnable}))
        if key not in self._fold_masks:
            self._fold_masks[key] = foldable_links(self._spec, keep=key)
        return key

    def _dynamics_walk(self) -> _DeviceWalk:
        """The whole-tree walk of the dynamics kernels, forward and backward.  Links behind fixed joints (end-effector frames,
        fingertips, and the flanges / palms / plates between moving joints) are folded away: their inertia is added to the
        nearest moving ancestor's row and their transform composed into the rows of the links below them, once on the host
        (flatten.fold_link_table), and the walk leaves them out — the same torques / inertia matrix / accelerations and the
        same gradients from one op per DoF (Panda 8 -> 7, Panda with gripper 12 -> 9, Allegro 20 -> 16, Fetch 24 -> 14).  Links
        with learnable parameters stay ops of their own, and so does whatever would have been folded into them or would have
        had to carry a transform for them (flatten.foldable_links)."""
        if not self._learnable and self._dyn_walk is not None:      # (constant model: the walk never changes)
            return self._dyn_walk
        mode = self._own_kernel_mode()
        have = self.__dict__.get("_dyn_walk_learnable")     # (a model with learnable links asks on every call)
        if have is not None and have[0] == (self._learnable_version, mode) and (
                mode == "off" or self._device.type != "cuda" or getattr(have[1].program, "_special_tried", False)):
            return have[1]
        key = self._fold_key()
        if not self._fold_masks[key].any():
            dw = self._get_walk(("tree",), whole_tree=True)
        else:
            dw = self._get_walk(("tree", "folded", key), whole_tree=True, folded=True, fold_key=key)
        if mode != "off" and self._device.type == "cuda" and not getattr(dw.program, "_special_tried", False):
            # the robot's OWN kernels on first use.  "auto" (the default, round 6): whatever is already built — the code objects
            # shipped next to the library, the run-time cache of an earlier specialize() — is attached, nothing is compiled; "build"
            # (DRM_SPECIALIZE=1, model.own_kernels = "build"): every robot without a compiled straight-line shape builds its kernels
            # (specialize.py; ~2 s once per robot and machine; a machine without hipcc keeps the loop kernels).  DRM_SPECIALIZE=tune:
            # robots WITH a compiled shape (other than plain 7-DoF arms) build theirs too and keep, per entry point, the faster
            dw.program._special_tried = True
            from . import specialize as sp
            from .flatten import SHAPE_ARM_CHAIN, SHAPE_ARM_HAND, SHAPE_FINGERS
            auto = mode == "auto"
            try:
                if sp.arm_qualifies(dw.program, self._n_dofs):
                    if not self._learnable:
                        sp.attach_arm(dw.program, self._ops_f(dw).detach().cpu().numpy(), self._n_dofs, cached_only=auto)   # (constants folded in)
                        for what in dw.program._special_missed:
                            self._own_kernel_miss(what, sp.CacheMiss("not in the cache"))
                    else:
                        # learnable link parameters: the reverse-mode kernel of THIS set of learnable blocks — every other block of
                        # the table is still a constant of the robot and folds into the instruction stream (round 6)
                        kin, dyn = self._learnable_block_masks(dw)
                        sp.attach_arm_param(dw.program, self._ops_f(dw).detach().cpu().numpy(), self._n_dofs, kin, dyn, cached_only=auto)
                elif not dw.program.shape & (SHAPE_ARM_CHAIN | SHAPE_ARM_HAND | SHAPE_FINGERS):
                    if dw.program.n_ops <= sp.MAX_STATIC_OPS:
                        sp.attach(dw.program, self._spec, self._n_dofs, self._const_table(dw), cached_only=auto)
                elif (os.environ.get("DRM_SPECIALIZE") == "tune" and not dw.program.shape & SHAPE_ARM_CHAIN
                      and dw.program.n_ops <= sp.MAX_STATIC_OPS):
                    sp.attach(dw.program, self._spec, self._n_dofs, self._const_table(dw), ignore_tuned=True)
                    sp.tune(dw.program, self._ops_f(dw).detach(), dw.ops_i, self._n_dofs)
                elif auto and not dw.program.shape & SHAPE_ARM_CHAIN and dw.program.n_ops <= sp.MAX_STATIC_OPS and not self._learnable:
                    # a robot WITH a compiled shape in the library (an arm that carries a hand, a hand): its own kernels where a
                    # tuning record — this machine's `specialize(tune=True)`, or the one shipped with the package — says they are
                    # the faster ones, and only those entry points; silently none without a record
                    try:
                        sp.attach(dw.program, self._spec, self._n_dofs, self._const_table(dw), cached_only=True, tuned_only=True)
                    except sp.CacheMiss:
                        pass
            except sp.CacheMiss as err:
                self._own_kernel_miss("the dynamics walk", err)
            except sp.SpecializeError:
                pass
        if not self._learnable:
            self._dyn_walk = dw
        else:
            self.__dict__["_dyn_walk_learnable"] = ((self._learnable_version, mode), dw)
        return dw

    def _static_rows(self, fold_key: Optional[tuple]) -> torch.Tensor:
        """[L + 1, OPF_STRIDE] snapshot of the constant rows, with the links of a fold mask folded into their parents."""
        if self._static_table is None:
            self._link_table()
        L1 = len(self._bodies) + 1
        if fold_key is None:
            return self._static_table[:L1]
        if fold_key not in self._static_folded:
            with torch.no_grad():
                rows = fold_link_table(self._spec, self._static_table[:L1].cpu().numpy(), self._fold_masks[fold_key])
                self._static_folded[fold_key] = torch.from_numpy(rows.astype(np.float32)).to(self._device)
        return self._static_folded[fold_key]

    def _ops_f(self, dw: _DeviceWalk) -> torch.Tensor:
        """[cap, OPF_STRIDE] constants gathered (and axis-canonicalised) in walk order; ONE differentiable
        gather (+ exact sign flips) from the link table, cached while nothing is learnable."""
        if not self._learnable and dw.static_ops_f is not None:
            return dw.static_ops_f
        if self._learnable and len(self._learnable_link_list()) <= 32 and not self._has_skew():
            return self._ops_f_learnable(dw)      # (more learnable links than the fused kernel takes: the torch path below)
        table = self._link_table(dw.fold_key)
        ops_f = (table.reshape(-1).index_select(0, dw.gather) * dw.gsign).reshape(dw.program.capacity, OPF_STRIDE)
        if not self._learnable:
            dw.static_ops_f = ops_f
        return ops_f

    def _learnable_link_list(self) -> list:
        """The links with a learnable parameter, ascending (cached per set of learnable parameters)."""
        have = self.__dict__.get("_learnable_sorted")
        if have is None or have[0] != self._learnable_version:
            have = self.__dict__["_learnable_sorted"] = (self._learnable_version, sorted({link for link, _ in self._learnable}))
        return have[1]

    def _has_skew(self) -> bool:
        have = self.__dict__.get("_skew_any")
        if have is None:
            have = self.__dict__["_skew_any"] = bool(self._spec.skew.any())
        return have

    def _ops_f_learnable(self, dw: _DeviceWalk) -> torch.Tensor:
        """The walk table with learnable links through ONE fused kernel (backend.WalkTable): the constant entries come
        from a cached gather of the constant link table, the entries of the learnable links are rebuilt from their
        parameter callables inside the kernel, and the autograd graph holds a single node."""
        links, base, sel = self._learnable_plan(dw)
        if self._table_links:
            # ABI 13: from the parameter tensors where they lie, the known modules' arithmetic inside the kernel (backend.WalkTableLinks)
            plan, sources = self._learnable_sources(links)
            ops_f = backend.WalkTableLinks.apply(base, sel, dw.gsign, plan, *sources)
        else:
            ops_f = backend.WalkTable.apply(base, sel, dw.gsign, len(links), *self._learnable_pieces(links))
        return ops_f.reshape(dw.program.capacity, OPF_STRIDE)

    def _learnable_sources(self, links):
        """(backend.LinkSourcePlan, live tensors) of the given learnable links.  A piece that is a constant of the model is part of
        the plan; a piece supplied by a parameter module is LIVE: the module's RAW parameter and its form where the kernels know the
        module (exactly PositiveScalar for mass / damping, exactly one of the l[6] inertia-matrix modules for inertia_mat; exactly
        UnconstrainedTensor / UnconstrainedScalar: the parameter itself), the output of the module's call otherwise.  Cached per set
        of links and modules."""
        from . import rigid_body_params as rbp
        key = tuple(links)
        cached = self._source_plan
        if cached is None or cached[0] != key:
            entries, fixed, getters, refs = [], [], [], []      # refs: (module._parameters, name) of a live piece, None for a module's call
            for i in links:
                b = self._bodies[i]
                link = []
                for name, fn in (("rot_angles", b.rot_angles), ("trans", b.trans), ("mass", b.inertia.mass), ("com", b.inertia.com),
                                 ("inertia_mat", b.inertia.inertia_mat), ("damping", b.joint_damping)):
                    form, const = backend.FORM_PLAIN, 0.0
                    if not isinstance(fn, torch.nn.Module):
                        value = fn()
                        value = self._zero1 if value is None else value
                        fixed.append(value.detach().to(device=self._device, dtype=torch.float32).reshape(-1).contiguous())
                        link.append((form, const, None))
                        continue
                    fixed.append(None)
                    kind = type(fn)
                    if name in ("mass", "damping") and kind is rbp.PositiveScalar:
                        form, const = backend.FORM_SQUARE_PLUS, fn._min_val
                    elif name == "inertia_mat" and kind is rbp.Symm3DInertiaMatrixNet:
                        form = backend.FORM_SYMM
                    elif name == "inertia_mat" and kind is rbp.SymmPosDef3DInertiaMatrixNet:
                        form, const = backend.FORM_SPD, fn.spd_3d_inertia_mat_diag_bias
                    elif name == "inertia_mat" and kind is rbp.CovParameterized3DInertiaMatrixNet:
                        form, const = backend.FORM_COV, fn.spd_3d_cov_inertia_mat_diag_bias
                    link.append((form, const, fn if form != backend.FORM_PLAIN else None))
                    if form != backend.FORM_PLAIN:           # (the module's own dictionary: nn.Module.__getattr__ is three times slower)
                        getters.append(lambda d=fn._parameters: d["l"])
                        refs.append((fn._parameters, "l"))
                    elif kind in (rbp.UnconstrainedTensor, rbp.UnconstrainedScalar):
                        getters.append(lambda d=fn._parameters: d["param"])
                        refs.append((fn._parameters, "param"))
                    else:
                        getters.append(fn)
                        refs.append(None)
                entries.append(link)
            cached = self._source_plan = (key, backend.LinkSourcePlan(entries, fixed), getters, refs)
        return cached[1], [g() for g in cached[2]]

    def _learnable_plan(self, dw: _DeviceWalk):
        """(learnable links, base, sel) of a walk: the table of the CONSTANT links gathered into walk order, and for every entry of
        the table the element of a learnable link's row it comes from (slot * 32 + column; -1: constant).  Cached per set of links."""
        links = self._learnable_link_list()
        key = tuple(links)
        plan = dw.learnable_plan
        if plan is None or plan[0] != key:
            with torch.no_grad():
                static = self._with_virtual_rows(self._static_rows(dw.fold_key))
                base = (static.reshape(-1).index_select(0, dw.gather) * dw.gsign).contiguous()
            gather = dw.program.gather.reshape(-1)
            row_of = {link: i for i, link in enumerate(links)}
            sel = np.full(gather.shape, -1, np.int32)
            for e, flat in enumerate(gather):
                link, col = divmod(int(flat), OPF_STRIDE)
                if link in row_of:
                    sel[e] = row_of[link] * OPF_STRIDE + col
            plan = dw.learnable_plan = (key, base, torch.from_numpy(sel).to(self._device))
        return links, plan[1], plan[2]

    def _learnable_pieces(self, links) -> list:
        """The outputs of the parameter callables of the given links, six per link (backend.WalkTable's ``pieces``)."""
        zero1 = self._zero1
        pieces = []
        for i in links:
            b = self._bodies[i]
            damping = b.joint_damping()
            pieces += [b.rot_angles(), b.trans(), b.inertia.mass(), b.inertia.com(), b.inertia.inertia_mat(),
                       damping if damping is not None else zero1]
        return pieces

    def _chain_walk(self, idx: int) -> _DeviceWalk:
        """The root -> link chain walk of the kinematics calls that do NOT run under autograd.  Without learnable parameters the
        fixed links between the chain's moving joints are stepped over (their transforms composed into the next link's row of
        the folded table, flatten.fold_link_table; a fixed TARGET keeps its op, with the fixed links right before it composed into
        its row): the same pose and Jacobian from fewer ops (Panda with gripper: 10 -> 8 to a finger or to the tool frame, a
        capacity-8 walk; TriFinger fingertip: 6 -> 4).  With learnable parameters, and under autograd, the plain walk."""
        if idx == 0:
            return self._get_walk(("chain", idx), targets=[])
        if self._learnable:
            return self._get_walk(("chain", idx), targets=[idx])
        hit = self._chain_walks.get(idx)      # (constant model: the walk of a link never changes; cleared when a link turns learnable)
        if hit is not None:
            return hit
        dw = self._chain_walks[idx] = self._chain_walk_build(idx)
        return dw

    def _chain_walk_build(self, idx: int) -> _DeviceWalk:
        key = self._fold_key()
        fold = self._fold_masks[key]
        chain = self._spec.chain_to(idx)
        if fold[idx] and len(chain) > 1 and fold[chain[-2]]:
            # the target sits behind several fixed joints in a row (flange -> hand -> tool frame): it alone keeps an op, whose row
            # carries the others' transforms — a fold mask of its own, the target un-folded
            tkey = ("target", idx) + tuple(key)
            if tkey not in self._fold_masks:
                mask = fold.copy()
                mask[idx] = False
                self._fold_masks[tkey] = mask
            key, fold = tkey, self._fold_masks[tkey]
        if not any(fold[c] and not all(fold[d] for d in chain[n + 1:]) for n, c in enumerate(chain)):
            return self._get_walk(("chain", idx), targets=[idx])      # nothing to step over
        return self._get_walk(("chain", idx, "folded", key), targets=[idx], folded=True, fold_key=key)

    def _fanout_chains(self, targets, merged: _DeviceWalk):
        """Per-target chain walks for the fan-out FK kernel, or None when the merged walk is the better plan: 2..4
        targets whose chains overlap so little that walking them separately costs < 1.25x the ops of the merged
        walk (the fingertips of a hand: every finger hangs off the palm)."""
        key = ("fanout", tuple(targets))
        if key not in self._fanout_plans:
            plan = None
            if 2 <= len(targets) <= 4:
                lengths = [len(self._spec.chain_to(t)) for t in targets]
                if sum(lengths) <= 1.25 * merged.program.n_ops and merged.program.n_ops > 8:
                    if not self._learnable:     # the folded chain walks (fixed links stepped over), when they share a capacity
                        walks = [self._chain_walk(t) for t in targets]
                        if len({w.program.capacity for w in walks}) == 1:
                            self._fanout_plans[key] = walks
                            self._fan_special(key, walks)
                            return walks
                    cap = max(build_walk(self._spec, targets=[t]).capacity for t in targets)
                    plan = []
                    for t in targets:
                        prog = build_walk(self._spec, targets=[t], min_capacity=cap)
                        plan.append(_DeviceWalk(
                            program=prog,
                            ops_i=torch.from_numpy(prog.ops_i_dev).to(self._device).contiguous(),
                            gather=torch.from_numpy(prog.gather.reshape(-1)).to(self._device),
                            gsign=torch.from_numpy(prog.gsign.reshape(-1)).to(self._device)))
            self._fanout_plans[key] = plan
        return self._fanout_plans[key]

    def _fan_special(self, key, walks) -> None:
        """The fan-out FK call's own kernel — every chain's constants folded into the instruction stream (specialize.attach_fan) —
        for THIS ordered set of targets (`key` of _fanout_plans): built already (the default) or compiled now (specialize(),
        DRM_SPECIALIZE=1).  The handle stays with the plan (`_fan_handles`), never on the chain walks, which other sets share."""
        if self._learnable or self._device.type != "cuda" or key in self._fan_handles:
            return
        mode = self._own_kernel_mode()
        if mode == "off":
            return
        from . import specialize as sp
        try:
            handle = sp.attach_fan([w.program for w in walks], [self._ops_f(w).detach().cpu().numpy() for w in walks], self._n_dofs,
                                   cached_only=mode == "auto")
            if handle:
                self._fan_handles[key] = handle
        except sp.CacheMiss as err:
            self._own_kernel_miss("forward kinematics of links %s" % (list(key[1]),), err)
        except sp.SpecializeError:
            if getattr(self, "_arm_specialized", False):
                raise      # (the environment opt-in on a machine without hipcc keeps the library's kernels)

    def _fan_own(self, targets) -> Optional[int]:
        """The handle of the own fan-out kernel of this ordered set of targets, or None."""
        return self._fan_handles.get(("fanout", tuple(targets))) if self._own_kernel_mode() != "off" else None

    def _kinematic_param_mask(self, dw: _DeviceWalk) -> int:
        """bit k set <=> op k's R_fixed / trans come from a learnable parametrisation (needs a constant gradient)."""
        have = dw.__dict__.get("_kin_mask")
        if have is not None and have[0] == self._learnable_version:
            return have[1]
        links = {link for link, pname in self._learnable if pname in ("trans", "rot_angles")}
        mask = 0
        for k, link in enumerate(dw.program.links):
            if int(link) in links:
                mask |= 1 << k
        dw.__dict__["_kin_mask"] = (self._learnable_version, mask)
        return mask

    def _differentiable(self, dw: _DeviceWalk) -> None:
        """The backward kernels take walks of up to 64 links and 6 branch points (any joint model)."""
        if not dw.program.backward_ok:
            raise NotImplementedError("gradients need a walk of <= 64 links with <= 6 branch points (this one: %d links, "
                                      "%d slots)" % (dw.program.n_ops, dw.program.n_slots))

    def _require_device(self):
        """The library that computes on the model's device must be there: libdrm_hip.so for a HIP device, libdrm_cpu.so (the host
        build of the same C ABI) for the CPU, the reference's default device (robot_model.py:100-104).  One never stands in for
        the other (backend.library_for raises NativeLibraryError)."""
        backend.library_for(self._device)

    # ------------------------------------------------------------------ kinematic state (reference API)
    @tensor_check
    def update_kinematic_state(self, q: torch.Tensor, qd: torch.Tensor) -> None:
        """Record the joint state; ``body.pose`` / ``body.vel`` of every link then reflect it (robot_model.py:139-195).

        The reference walks the tree here and stores a pose and a velocity on every body; the kernels are
        stateless, so this only remembers (q, qd) and the per-link quantities are evaluated when read:
        poses by one all-links FK launch, a link's body-frame velocity from its Jacobian (R^T J qd).
        """
        assert q.ndim == 2
        assert qd.ndim == 2
        assert q.shape[1] == self._n_dofs
        assert qd.shape[1] == self._n_dofs
        self._kin_state = (q.detach(), qd.detach())
        self._kin_cache = {}

    def _all_poses(self):
        """pos [B, L', 3], quat [B, L', 4], rot [B, L', 3, 3] of every link of the recorded state, indexed by link (the kernel's
        outputs as they stand when the links are in walk order, else gathered once)."""
        if "poses" not in self._kin_cache:
            q = self._kin_state[0]
            with torch.no_grad():
                cols = self._fk_links(q, list(range(len(self._bodies))))
                pos = torch.stack([cols[i][0] for i in range(len(self._bodies))], dim=1)
                quat = torch.stack([cols[i][1] for i in range(len(self._bodies))], dim=1)
            x, y, z, w = quat.unbind(-1)
            rot = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                               2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                               2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1)
            self._kin_cache["poses"] = (pos, quat, rot.reshape(*quat.shape[:-1], 3, 3))
        return self._kin_cache["poses"]

    def _link_pose(self, idx: int) -> LinkPose:
        if self._kin_state is None:   # a fresh body sits at the identity (rigid_body.py:64-76)
            return LinkPose(torch.eye(3, device=self._device).unsqueeze(0), torch.zeros(1, 3, device=self._device),
                            torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=self._device))
        pos, quat, rot = self._all_poses()
        return LinkPose(rot[:, idx], pos[:, idx], quat[:, idx])

    def _link_velocity(self, idx: int) -> LinkVelocity:
        if self._kin_state is None:
            return LinkVelocity(torch.zeros(1, 3, device=self._device), torch.zeros(1, 3, device=self._device))
        key = ("vel", idx)
        if key not in self._kin_cache:
            q, qd = self._kin_state
            rot = self._all_poses()[2][:, idx]
            with torch.no_grad():
                lin, ang = self.compute_endeffector_jacobian(q, self._bodies[idx].name)
            to_body = lambda jac: torch.einsum("bji,bj->bi", rot, torch.einsum("bij,bj->bi", jac, qd))
            self._kin_cache[key] = LinkVelocity(to_body(lin), to_body(ang))
        return self._kin_cache[key]

    # ------------------------------------------------------------------ FK
    def _fk_links(self, q: torch.Tensor, link_idxs: List[int]) -> Dict[int, Tuple[torch.Tensor, torch.Tensor]]:
        """{link index: (pos [B,3], quat [B,4])} — column views of ONE launch whose targets are taken in WALK order (the many-target
        kernel then writes its outputs a group of consecutive slots at a time, DRM_WALK_TARGETS_ORDERED); the root's identity
        pose is two small tensors of its own."""
        out = {}
        B = q.shape[0]
        # (the walk order of a set of links is found once per set: the per-call Python work of a 13-link dictionary was ~10x the
        # kernel's time at 65 536 rows)
        plans = self.__dict__.setdefault("_fk_links_plans", {})
        key = tuple(link_idxs)
        plan = plans.get(key)
        if plan is None:
            wanted = set(int(i) for i in link_idxs if i != 0)
            ordered = [i for i in self._spec.preorder() if i in wanted]
            plan = plans[key] = (0 in link_idxs, ordered)
        has_root, ordered = plan
        if has_root:            # (read-only views of two constants: no kernel, no memory)
            if self._root_pose is None:
                self._root_pose = (torch.zeros(1, 3, device=self._device), torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=self._device))
            out[0] = (self._root_pose[0].expand(B, 3), self._root_pose[1].expand(B, 4))
        if ordered:
            dw = self._get_walk(("fk", tuple(ordered)), targets=ordered) if len(ordered) > 1 else None
            if dw is not None and not (torch.is_grad_enabled() and (q.requires_grad or self._ops_f(dw).requires_grad)):
                # no graph to build: link-major outputs — every link's poses a contiguous array.  A few links at the ends of
                # (nearly) disjoint chains, the fingertips of a hand: a wavefront per chain (drm_fk_fanout_links); else one walk
                # over all of them (drm_fk_links)
                self._require_device()
                fan = self._fanout_chains(ordered, dw)
                if fan is not None:
                    pos, quat = backend.fk_fanout([(c.program, self._ops_f(c), c.ops_i) for c in fan], q, self._n_dofs, link_major=True,
                                                  own=self._fan_own(ordered))
                else:
                    pos, quat = backend.fk_links(dw.program, self._ops_f(dw), dw.ops_i, q, len(ordered), self._n_dofs)
                out.update(zip(ordered, zip(pos.unbind(0), quat.unbind(0))))      # (one unbind per array instead of an index op per link)
                return out
            pos, quat = self._fk_targets(q, ordered)
            out.update(zip(ordered, zip(pos.unbind(1), quat.unbind(1))))
        return out

    def _fast_entry_learnable(self, fast, entry: str, scratch: Optional[str], dw: "_DeviceWalk", chain, target_op: int) -> Optional[tuple]:
        """The prepared call of a model WITH learnable link parameters (round 6): the C++ side rebuilds the walk table from the parameter
        tensors in front of every call (drm_walk_table_links, looked up in the modules' own parameter dictionaries each time) whenever
        the call builds no autograd graph — torch.no_grad(), or every parameter frozen: a learned model in a control loop costs what a
        constant one does plus one small launch.  None where the table does not come from the links path, a live piece is the output of
        a module the kernels do not know, or the call needs two walks."""
        import ctypes
        if (chain is not None or not hasattr(fast.FastCall, "set_table") or not self._table_links or self._has_skew()
                or len(self._learnable_link_list()) > 32):
            return None
        links, base, sel = self._learnable_plan(dw)
        plan, sources = self._learnable_sources(links)
        refs = self._source_plan[3]
        if any(r is None for r in refs) or any(t.device != self._device or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n
                                              for t, n in zip(sources, plan.sizes)):
            return None
        lib = backend.library_for(self._device)
        with torch.no_grad():
            ops_f = self._ops_f(dw)
        shared = backend._walk_struct(dw.program, ops_f.detach(), dw.ops_i, self._n_dofs)
        walk = backend.DrmWalk.from_buffer_copy(shared)      # (this call's own struct: the C++ side rewrites its table pointer per call)
        pieces = plan.pieces(sources)
        keep = [walk, dw.ops_i, lib, plan, base, sel, dw.gsign]
        call = fast.FastCall(backend._fn_addr(lib, entry), backend._fn_addr(lib, scratch) if scratch else 0, ctypes.addressof(walk), 0,
                             self._n_dofs, target_op, self._device.type == "cuda",
                             self._device.index if self._device.index is not None else -1, tuple(keep))
        call.set_table(backend._fn_addr(lib, "drm_walk_table_links"), ctypes.addressof(pieces), ctypes.addressof(plan.forms), plan.n_links,
                       base, sel.data_ptr(), dw.gsign.data_ptr(), list(plan.live), list(plan.sizes), [r[0] for r in refs], [r[1] for r in refs])
        return (call, dw.program, dw.program._ws_cache)

    def _fk_targets(self, q: torch.Tensor, link_idxs: List[int]) -> Tuple[torch.Tensor, torch.Tensor]:
        """pos [B,T,3], quat [B,T,4] of the given links (root targets filled with the identity pose)."""
        self._require_device()
        B = q.shape[0]
        non_root = [i for i in link_idxs if i != 0]
        pos = quat = None
        if len(non_root) != len(link_idxs):    # the root link's pose is the identity: no kernel needed for it
            pos = torch.zeros(B, len(link_idxs), 3, device=self._device)
            quat = torch.zeros(B, len(link_idxs), 4, device=self._device)
            quat[..., 3] = 1.0
        if non_root:
            dw = self._get_walk(("fk", tuple(non_root)), targets=non_root)
            ops_f = self._ops_f(dw)
            needs_grad = torch.is_grad_enabled() and (q.requires_grad or ops_f.requires_grad)
            fan = None if needs_grad else self._fanout_chains(non_root, dw)
            if fan is not None:
                p, r = backend.fk_fanout([(c.program, self._ops_f(c), c.ops_i) for c in fan], q, self._n_dofs)
            elif needs_grad and len(non_root) > 1 and not dw.program.backward_ok:
                # a many-target walk the backward kernels do not take (more than 6 branch points open at once, > 64 ops): the targets'
                # root -> link chains one by one — a chain has no branch point — as the reference's per-link autograd graph would
                parts = [self._fk_targets(q, [i]) for i in non_root]
                p, r = torch.cat([a for a, _ in parts], dim=1), torch.cat([b for _, b in parts], dim=1)
            elif needs_grad:
                self._differentiable(dw)
                p, r = _FkPositions.apply(q, ops_f, dw, len(non_root), self._n_dofs, self._kinematic_param_mask(dw))
            else:
                p, r = backend.fk(dw.program, ops_f, dw.ops_i, q, len(non_root), self._n_dofs)
            if len(non_root) == len(link_idxs):
                return p, r
            cols = [k for k, i in enumerate(link_idxs) if i != 0]
            if cols == list(range(cols[0], cols[0] + len(cols))):     # (a slice: no index tensor, capturable into a hipGraph)
                pos[:, cols[0]:cols[0] + len(cols)] = p
                quat[:, cols[0]:cols[0] + len(cols)] = r
            else:
                pos[:, cols] = p
                quat[:, cols] = r
        return pos, quat

    def compute_forward_kinematics_all_links(self, q: torch.Tensor) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
        """{link_name: (pos [B,3], quat_xyzw [B,4])} for every link (robot_model.py:197-221).  (Repeat calls of a constant model with a
        batched q: one C++ call over drm_fk_links' link-major outputs, see compute_forward_kinematics.)"""
        ent = self.__dict__.get("_fast_links")
        if ent is not None and ent[1]._ws_cache is ent[2]:
            pairs = ent[0].links(q, len(ent[3]), self._stream_arg())
            if pairs.__class__ is list:
                names, B = self.__dict__["_body_names"], q.shape[0]
                out = dict.fromkeys(names)
                out[names[0]] = (self._root_pose[0].expand(B, 3), self._root_pose[1].expand(B, 4))
                for i, pair in zip(ent[3], pairs):
                    out[names[i]] = pair
                return out
            if pairs is not None:
                backend._check(pairs, backend.library_for(self._device))
        return self._compute_forward_kinematics_all_links(q)

    @tensor_check
    def _compute_forward_kinematics_all_links(self, q: torch.Tensor) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
        assert q.ndim == 2
        assert q.shape[1] == self._n_dofs
        names = self.__dict__.get("_body_names")
        if names is None:       # (ModuleList indexing costs ~3 us per link and call)
            names = self.__dict__["_body_names"] = [b.name for b in self._bodies]
            self.__dict__["_all_link_idxs"] = list(range(len(names)))
        ent = self.__dict__.get("_fast_links")
        cols = self._fk_links(q, self.__dict__["_all_link_idxs"])
        if ent is None or ent[1]._ws_cache is not ent[2]:
            self.__dict__["_fast_links"] = self._fast_links_entry()
        return {name: cols[i] for i, name in enumerate(names)}

    def _fast_links_entry(self) -> Optional[tuple]:
        """(FastCall of drm_fk_links over every non-root link, its walk program, the program's struct cache, the links in the walk's
        target order) — None for models the prepared call does not serve (learnable parameters, a robot of one link, no C++ host path)."""
        plan = self.__dict__.get("_fk_links_plans", {}).get(tuple(self.__dict__["_all_link_idxs"]))
        if plan is None or not plan[0] or len(plan[1]) < 2 or self._root_pose is None:
            return None
        ordered = plan[1]
        dw = self._get_walk(("fk", tuple(ordered)), targets=ordered)
        if self._fanout_chains(ordered, dw) is not None:
            return None
        ent = self._fast_entry("drm_fk_links", None, dw)
        return None if ent is None else ent + (list(ordered),)

    @tensor_check
    def compute_forward_kinematics_links(self, q: torch.Tensor, link_names: List[str]) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
        """{link name: (pos [B, 3], quat xyzw [B, 4])} of the NAMED links in one launch — what calling
        compute_forward_kinematics (robot_model.py:223-248) once per link returns, e.g. for the fingertips of a hand (BASELINE
        configuration 4).  Not in the reference (which recomputes every link per call, robot_model.py:139-195).  Without a graph to
        build every tensor is a contiguous array of its own (link-major launch); under autograd they are columns of one
        [B, T, .] result, differentiable like compute_forward_kinematics."""
        idxs = [self._name_to_idx_map[name] for name in link_names]
        cols = self._fk_links(q, idxs)
        return {name: cols[i] for name, i in zip(link_names, idxs)}

    def _fast_entry(self, entry: str, scratch: Optional[str], dw: "_DeviceWalk", chain: Optional["_DeviceWalk"] = None,
                    target_op: int = -1) -> Optional[tuple]:
        """(FastCall, walk program, the program's struct cache) of one prepared eager call of this walk through the C ABI's `entry`
        — None when the C++ host path is not built, the model has learnable parameters (its table changes every call) or lives on
        a device kind the path does not serve.  The call stays valid while the program's struct cache does (the identity is checked
        per call): attaching own kernels or rebuilding the table retires it and the next call through the Python path renews it."""
        fast = backend.hostcall()
        if fast is None or not hasattr(fast, "FastCall") or self._device.type not in ("cuda", "cpu"):
            return None
        if self._learnable:
            return self._fast_entry_learnable(fast, entry, scratch, dw, chain, target_op)
        import ctypes
        lib = backend.library_for(self._device)
        ops_f = self._ops_f(dw)
        walk = backend._walk_struct(dw.program, ops_f, dw.ops_i, self._n_dofs)
        keep, walk2 = [walk, ops_f, dw.ops_i, lib], 0
        if chain is not None:      # (drm_fk_rnea: the dynamics walk and the target's chain walk)
            cf = self._ops_f(chain)
            w2 = backend._walk_struct(chain.program, cf, chain.ops_i, self._n_dofs)
            keep += [w2, cf, chain.ops_i]
            walk2 = ctypes.addressof(w2)
        call = fast.FastCall(backend._fn_addr(lib, entry), backend._fn_addr(lib, scratch) if scratch else 0, ctypes.addressof(walk), walk2,
                             self._n_dofs, target_op, self._device.type == "cuda",
                             self._device.index if self._device.index is not None else -1, tuple(keep))
        if chain is not None:
            return (call, dw.program, dw.program._ws_cache, chain.program, chain.program._ws_cache)
        return (call, dw.program, dw.program._ws_cache)

    def compute_forward_kinematics(self, q: torch.Tensor, link_name: str, recursive: bool = False
                                   ) -> Tuple[torch.Tensor, torch.Tensor]:
        """(pos [B,3], quat_xyzw [B,4]) of ``link_name`` (robot_model.py:223-248).

        ``recursive`` selects between two implementations in the reference that return the same
        pose on a fresh model; both map to the same kernel here.
        A constant model's repeat calls go through ONE C++ call (csrc/drm_hostcall.cpp FastCall: tensor_check, the asserts below,
        the allocation and the launch); whatever that does not take as it is — and every first call — comes here.
        """
        ent = self._fast_fk.get(link_name)
        if ent is not None and ent[1]._ws_cache is ent[2]:
            out = ent[0].kinematics(q, 0, self._stream_arg())
            if out.__class__ is tuple:
                return out
            if out is not None:
                backend._check(out, backend.library_for(self._device))
        return self._compute_forward_kinematics(q, link_name, recursive)

    @tensor_check
    def _compute_forward_kinematics(self, q: torch.Tensor, link_name: str, recursive: bool = False):
        assert q.ndim == 2
        assert q.shape[1] == self._n_dofs
        idx = self._name_to_idx_map[link_name]
        if idx != 0 and not (torch.is_grad_enabled() and (q.requires_grad or self._learnable)):
            # the common call: one kernel, outputs allocated in their final shape (no slicing ops on the way out)
            self._require_device()
            dw = self._chain_walk(idx)
            out = backend.fk(dw.program, self._ops_f(dw), dw.ops_i, q, 1, self._n_dofs, squeeze=True)
            if link_name not in self._fast_fk or self._fast_fk[link_name][1]._ws_cache is not self._fast_fk[link_name][2]:
                ent = self._fast_entry("drm_fk", None, dw)
                if ent is not None:
                    self._fast_fk[link_name] = ent
            return out
        pos, quat = self._fk_targets(q, [idx])
        return pos[:, 0], quat[:, 0]

    def fk_mse_loss(self, q: torch.Tensor, link_name: str, target: torch.Tensor) -> torch.Tensor:
        """``torch.nn.functional.mse_loss(self.compute_forward_kinematics(q, link_name)[0], target)`` — the loss of the reference's
        kinematics-learning loop (examples/learn_kinematics_of_iiwa.py:47-55) — as ONE differentiable node: for a serial 7-DoF arm
        (the iiwa of BASELINE configuration 5, the Panda) and a batch that is a multiple of 64 rows, forward kinematics, the loss
        and its gradients with respect to q and the learnable ``trans`` / ``rot_angles`` come from one pass over q (drm_fk_mse)
        instead of an FK launch, the loss kernels and a backward launch; every other robot / batch takes exactly the composition
        above.  Not in the reference."""
        assert q.ndim == 2 and q.shape[1] == self._n_dofs and target.shape == (q.shape[0], 3)
        self._require_device()
        idx = self._name_to_idx_map[link_name]
        if idx != 0 and q.shape[0] % 64 == 0 and q.shape[0] > 0:
            dw = self._get_walk(("fk", (idx,)), targets=[idx])
            if dw.program.shape & 1 and dw.program.capacity == 8 and self._n_dofs == 7:    # DRM_WALK_ARM_CHAIN
                if (self._learnable and torch.is_grad_enabled() and not self._spec.skew.any()
                        and len({link for link, _ in self._learnable}) <= backend.FK_MSE_MAX_LINKS and self._fk_mse_links):
                    # learnable links: from their parameter tensors to the gradients with respect to them in TWO launches and one
                    # autograd node (drm_fk_mse_links, ABI 12) — no table pass before, no table backward pass after
                    links, base, sel = self._learnable_plan(dw)
                    pieces = self._learnable_pieces(links)
                    mask = self._kinematic_param_mask(dw)
                    if mask and all(p.device == self._device for p in pieces):
                        try:
                            self._differentiable(dw)
                            return _FkMseLinks.apply(q, target, base, sel, dw.gsign, dw, self._n_dofs, mask, *pieces)
                        except backend.KernelUnsupported:
                            pass
                ops_f = self._ops_f(dw)
                try:
                    if torch.is_grad_enabled() and (q.requires_grad or ops_f.requires_grad):
                        self._differentiable(dw)
                        return _FkMse.apply(q, target, ops_f, dw, self._n_dofs, self._kinematic_param_mask(dw))
                    return backend.fk_mse(dw.program, ops_f, dw.ops_i, q, target, self._n_dofs, 0, False)[0]
                except backend.KernelUnsupported:
                    pass
        pos, _ = self.compute_forward_kinematics(q, link_name)
        return torch.nn.functional.mse_loss(pos, target)

    # ------------------------------------------------------------------ Jacobian
    def compute_endeffector_jacobian(self, q: torch.Tensor, link_name: str) -> Tuple[torch.Tensor, torch.Tensor]:
        """(lin_jac [B,3,n], ang_jac [B,3,n]) at the link origin, world frame (robot_model.py:626-667).  (Repeat calls of a constant
        model: one C++ call, see compute_forward_kinematics.)"""
        ent = self._fast_jac.get(link_name)
        if ent is not None and ent[1]._ws_cache is ent[2]:
            out = ent[0].kinematics(q, 1, self._stream_arg())
            if out.__class__ is tuple:
                return out
            if out is not None:
                backend._check(out, backend.library_for(self._device))
        return self._compute_endeffector_jacobian(q, link_name)

    @tensor_check
    def _compute_endeffector_jacobian(self, q: torch.Tensor, link_name: str):
        assert len(q.shape) == 2
        assert q.shape[1] == self._n_dofs
        _, _, lin, ang = self._fk_and_jacobian(q, link_name)
        return lin, ang

    def compute_fk_and_jacobian(self, q: torch.Tensor, link_name: str
                                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """(pos, quat, lin_jac, ang_jac) from ONE fused kernel launch.

        The reference's ``compute_endeffector_jacobian`` runs the full FK first and throws the pose
        away (robot_model.py:641); this returns it instead.
        """
        ent = self._fast_jac.get(link_name)
        if ent is not None and ent[1]._ws_cache is ent[2]:
            out = ent[0].kinematics(q, 2, self._stream_arg())
            if out.__class__ is tuple:
                return out
            if out is not None:
                backend._check(out, backend.library_for(self._device))
        return self._compute_fk_and_jacobian(q, link_name)

    @tensor_check
    def _compute_fk_and_jacobian(self, q: torch.Tensor, link_name: str):
        assert q.ndim == 2
        assert q.shape[1] == self._n_dofs
        return self._fk_and_jacobian(q, link_name)

    def _fk_and_jacobian(self, q: torch.Tensor, link_name: str):
        self._require_device()
        idx = self._name_to_idx_map[link_name]
        if idx != 0 and torch.is_grad_enabled() and (q.requires_grad or self._learnable):
            dw = self._get_walk(("chain", idx), targets=[idx])
            ops_f = self._ops_f(dw)
            if q.requires_grad or ops_f.requires_grad:
                self._differentiable(dw)
                return _FkJacobian.apply(q, ops_f, dw, self._n_dofs, self._kinematic_param_mask(dw))
            return backend.fk_jacobian(dw.program, ops_f, dw.ops_i, q, self._n_dofs)
        dw = self._chain_walk(idx)
        out = backend.fk_jacobian(dw.program, self._ops_f(dw), dw.ops_i, q, self._n_dofs)
        have = self._fast_jac.get(link_name)
        if idx != 0 and (have is None or have[1]._ws_cache is not have[2]):
            ent = self._fast_entry("drm_fk_jacobian", None, dw)
            if ent is not None:
                self._fast_jac[link_name] = ent
        return out

    def plan_fk_and_jacobian(self, q: torch.Tensor, link_name: str, want_pose: bool = True
                             ) -> "backend.FkJacobianPlan":
        """Prepared (allocation-free, graph-capturable) FK+Jacobian launch on fixed buffers."""
        self._require_device()
        assert q.ndim == 2 and q.shape[1] == self._n_dofs
        assert not self._learnable, "plans snapshot the constants; not available with learnable parameters"
        idx = self._name_to_idx_map[link_name]
        dw = self._chain_walk(idx)
        return backend.FkJacobianPlan(dw.program, self._ops_f(dw), dw.ops_i, q, self._n_dofs, want_pose)

    def plan_inverse_dynamics(self, q: torch.Tensor, qd: torch.Tensor, qdd_des: Optional[torch.Tensor],
                              include_gravity: bool = True, use_damping: bool = True) -> "backend.InverseDynamicsPlan":
        """Prepared (allocation-free, graph-capturable) inverse-dynamics launch on fixed buffers; ``qdd_des=None``
        gives the non-linear effects."""
        self._require_device()
        assert q.ndim == 2 and q.shape[1] == self._n_dofs
        assert not self._learnable, "plans snapshot the constants; not available with learnable parameters"
        dw = self._dynamics_walk()
        return backend.InverseDynamicsPlan(dw.program, self._ops_f(dw), dw.ops_i, q, qd, qdd_des, bool(include_gravity),
                                           bool(use_damping), self._n_dofs)

    def plan_fk_and_inverse_dynamics(self, q: torch.Tensor, qd: torch.Tensor, qdd_des: Optional[torch.Tensor],
                                     link_name: str, include_gravity: bool = True, use_damping: bool = True,
                                     outputs=None, put=None) -> "backend.FkInverseDynamicsPlan":
        """Prepared launch of inverse dynamics + the pose of ``link_name`` on fixed buffers (drm_fk_rnea): what the
        reference computes with compute_inverse_dynamics (robot_model.py:305-375) followed by
        compute_forward_kinematics (robot_model.py:223-248) on the same q.  One fused kernel for a serial 7-DoF arm
        whose last link is the target, the two walks back to back otherwise.  ``outputs``: caller-owned (tau [B,n], pos [B,3],
        quat [B,4]) buffers to write into.  ``put``: the destinations of a one-sided gather (distributed.PeerGather.put(): every
        launch also writes its rows into the other ranks' gathered arrays, drm_fk_rnea_put)."""
        self._require_device()
        assert q.ndim == 2 and q.shape[1] == self._n_dofs
        assert not self._learnable, "plans snapshot the constants; not available with learnable parameters"
        idx = self._name_to_idx_map[link_name]
        if idx == 0:
            raise ValueError("the root link has the identity pose; use plan_inverse_dynamics")
        tree = self._dynamics_walk()
        chain = self._get_walk(("chain", idx) + (("folded", tree.fold_key) if tree.folded else ()), targets=[idx],
                               folded=tree.folded, fold_key=tree.fold_key)
        self._arm_fused_special(tree, chain)
        return backend.FkInverseDynamicsPlan((tree.program, self._ops_f(tree), tree.ops_i),
                                             (chain.program, self._ops_f(chain), chain.ops_i),
                                             int(tree.program.op_of_link.get(idx, -1)),
                                             q, qd, qdd_des, bool(include_gravity), bool(use_damping), self._n_dofs,
                                             outputs=outputs, put=put)

    def compute_fk_and_inverse_dynamics(self, q: torch.Tensor, qd: torch.Tensor, qdd_des: torch.Tensor, link_name: str,
                                        include_gravity: Optional[bool] = True, use_damping: Optional[bool] = True
                                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(tau [B,n], pos [B,3], quat [B,4]) from one pass over q — see _compute_fk_and_inverse_dynamics; repeat calls of a constant
        model go through one prepared C++ call (csrc/drm_hostcall.cpp FastCall)."""
        ent = self._fast_fkid.get(link_name)
        if ent is not None and ent[1]._ws_cache is ent[2] and ent[3]._ws_cache is ent[4]:
            out = ent[0].fk_inverse_dynamics(q, qd, qdd_des, (1 if include_gravity else 0) | (2 if use_damping else 0), self._stream_arg())
            if out.__class__ is tuple:
                return out
            if out is not None:
                backend._check(out, backend.library_for(self._device))
        return self._compute_fk_and_inverse_dynamics(q, qd, qdd_des, link_name, include_gravity, use_damping)

    @tensor_check
    def _compute_fk_and_inverse_dynamics(self, q: torch.Tensor, qd: torch.Tensor, qdd_des: torch.Tensor, link_name: str,
                                         include_gravity: Optional[bool] = True, use_damping: Optional[bool] = True
                                         ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(tau [B,n], pos [B,3], quat [B,4]): compute_inverse_dynamics and compute_forward_kinematics of one link from
        a single pass over q (not differentiable; use the two separate methods under autograd)."""
        assert q.ndim == 2 and qd.ndim == 2 and qdd_des.ndim == 2
        assert q.shape[1] == self._n_dofs and qd.shape[1] == self._n_dofs and qdd_des.shape[1] == self._n_dofs
        idx = self._name_to_idx_map[link_name]
        if idx != 0 and not self._learnable:      # one eager call through the C++ host path (no plan object per call)
            self._require_device()
            tree = self._dynamics_walk()
            chain = self._get_walk(("chain", idx) + (("folded", tree.fold_key) if tree.folded else ()), targets=[idx],
                                   folded=tree.folded, fold_key=tree.fold_key)
            self._arm_fused_special(tree, chain)
            out = backend.fk_rnea((tree.program, self._ops_f(tree), tree.ops_i), (chain.program, self._ops_f(chain), chain.ops_i),
                                  int(tree.program.op_of_link.get(idx, -1)), q.detach(), qd.detach(), qdd_des.detach(),
                                  bool(include_gravity), bool(use_damping), self._n_dofs)
            if out is not None:
                have = self._fast_fkid.get(link_name)
                if have is None or have[1]._ws_cache is not have[2] or have[3]._ws_cache is not have[4]:
                    ent = self._fast_entry("drm_fk_rnea", "drm_rnea_scratch_floats_aligned", tree, chain,
                                           int(tree.program.op_of_link.get(idx, -1)))
                    if ent is not None:
                        self._fast_fkid[link_name] = ent
                return out
        plan = self.plan_fk_and_inverse_dynamics(q.detach(), qd.detach(), qdd_des.detach(), link_name,
                                                 bool(include_gravity), bool(use_damping))
        with backend._on_device(self._device):
            plan.launch()
        return plan.tau, plan.pos, plan.quat

    # ------------------------------------------------------------------ inverse dynamics
    def compute_inverse_dynamics(self, q: torch.Tensor, qd: torch.Tensor, qdd_des: torch.Tensor,
                                 include_gravity: Optional[bool] = True, use_damping: Optional[bool] = True
                                 ) -> torch.Tensor:
        """tau [B,n] achieving ``qdd_des`` (RNEA, robot_model.py:305-375).  (Repeat calls of a constant model: one C++ call, see
        compute_forward_kinematics.)"""
        ent = self._fast_id
        if ent is not None and ent[1]._ws_cache is ent[2]:
            out = ent[0].inverse_dynamics(q, qd, qdd_des, (1 if include_gravity else 0) | (2 if use_damping else 0), self._stream_arg())
            if out.__class__ is torch.Tensor:
                return out
            if out is not None:
                backend._check(out, backend.library_for(self._device))
        return self._compute_inverse_dynamics(q, qd, qdd_des, include_gravity, use_damping)

    @tensor_check
    def _compute_inverse_dynamics(self, q: torch.Tensor, qd: torch.Tensor, qdd_des: torch.Tensor,
                                  include_gravity: Optional[bool] = True, use_damping: Optional[bool] = True):
        assert q.ndim == 2
        assert qd.ndim == 2
        assert qdd_des.ndim == 2
        assert q.shape[1] == self._n_dofs
        assert qd.shape[1] == self._n_dofs
        assert qdd_des.shape[1] == self._n_dofs
        self._require_device()
        return self._inverse_dynamics(q, qd, qdd_des, bool(include_gravity), bool(use_damping))

    def compute_non_linear_effects(self, q: torch.Tensor, qd: torch.Tensor, include_gravity: Optional[bool] = True,
                                   use_damping: Optional[bool] = True) -> torch.Tensor:
        """Coriolis + centrifugal + gravity + damping torques = RNEA with qdd = 0 (robot_model.py:377-400)."""
        ent = self._fast_id
        if ent is not None and ent[1]._ws_cache is ent[2]:
            out = ent[0].inverse_dynamics(q, qd, None, (1 if include_gravity else 0) | (2 if use_damping else 0), self._stream_arg())
            if out.__class__ is torch.Tensor:
                return out
            if out is not None:
                backend._check(out, backend.library_for(self._device))
        return self._compute_non_linear_effects(q, qd, include_gravity, use_damping)

    @tensor_check
    def _compute_non_linear_effects(self, q: torch.Tensor, qd: torch.Tensor, include_gravity: Optional[bool] = True,
                                    use_damping: Optional[bool] = True):
        assert q.ndim == 2 and qd.ndim == 2
        assert q.shape[1] == self._n_dofs and qd.shape[1] == self._n_dofs
        self._require_device()
        return self._inverse_dynamics(q, qd, None, bool(include_gravity), bool(use_damping))

    def iterative_newton_euler(self, base_lin_acc: torch.Tensor, base_ang_acc: torch.Tensor) -> None:
        """The reference's RNEA sweeps over per-body state written by ``update_kinematic_state`` (robot_model.py:250-303):
        an internal step of its Python recursion.  Here both sweeps are one kernel behind ``compute_inverse_dynamics``."""
        raise NotImplementedError("iterative_newton_euler is an internal step of the reference's per-link recursion; "
                                  "call compute_inverse_dynamics / compute_non_linear_effects")

    def _learnable_block_masks(self, dw) -> Tuple[int, int]:
        """(kinematic, dynamic): bit k set <=> op k's F / t block (`trans`, `rot_angles`) / its mass - mcom - I_o - damping block
        (`mass`, `com`, `inertia_mat`, `joint_damping`) comes from a learnable parametrisation."""
        kin_links = {link for link, pname in self._learnable if pname in ("trans", "rot_angles")}
        dyn_links = {link for link, pname in self._learnable if pname not in ("trans", "rot_angles")}
        kin = dyn = 0
        for k, link in enumerate(dw.program.links):
            if int(link) in kin_links:
                kin |= 1 << k
            if int(link) in dyn_links:
                dyn |= 1 << k
        return kin, dyn

    def _learnable_op_mask(self, dw) -> int:
        """Bit k set <=> op k of the walk belongs to a link with a learnable parameter (param_mask of the backward kernels)."""
        have = dw.__dict__.get("_op_mask")
        if have is not None and have[0] == self._learnable_version:
            return have[1]
        links = {link for link, _ in self._learnable}
        mask = 0
        for k, link in enumerate(dw.program.links):
            if int(link) in links:
                mask |= 1 << k
        dw.__dict__["_op_mask"] = (self._learnable_version, mask)
        return mask

    def _inverse_dynamics(self, q, qd, qdd, gravity: bool, damping: bool) -> torch.Tensor:
        dw = self._dynamics_walk()    # (the full walk whenever a link parameter is learnable)
        ops_f = self._ops_f(dw)
        needs_grad = torch.is_grad_enabled() and (ops_f.requires_grad or any(
            t is not None and t.requires_grad for t in (q, qd, qdd)))
        if needs_grad:
            self._differentiable(dw)
            return _InverseDynamics.apply(q, qd, qdd, ops_f, dw, gravity, damping, self._n_dofs,
                                          self._learnable_op_mask(dw))
        out = backend.rnea(dw.program, ops_f, dw.ops_i, q, qd, qdd, gravity, damping, self._n_dofs)
        have = self._fast_id
        if have is None or have[1]._ws_cache is not have[2]:
            self._fast_id = self._fast_entry("drm_rnea", "drm_rnea_scratch_floats_aligned", dw)
        return out

    def compute_lagrangian_inertia_matrix(self, q: torch.Tensor, include_gravity: Optional[bool] = True,
                                          use_damping: Optional[bool] = True) -> torch.Tensor:
        """H(q) [B, n, n] (robot_model.py:402-450) — see _compute_lagrangian_inertia_matrix; repeat calls of a constant model go
        through one prepared C++ call (csrc/drm_hostcall.cpp FastCall)."""
        ent = self._fast_crba
        if ent is not None and ent[1]._ws_cache is ent[2]:
            out = ent[0].inertia_matrix(q, self._stream_arg())
            if out.__class__ is torch.Tensor:
                return out
            if out is not None:
                backend._check(out, backend.library_for(self._device))
        return self._compute_lagrangian_inertia_matrix(q, include_gravity, use_damping)

    @tensor_check
    def _compute_lagrangian_inertia_matrix(self, q: torch.Tensor, include_gravity: Optional[bool] = True,
                                           use_damping: Optional[bool] = True) -> torch.Tensor:
        """Joint-space inertia matrix H(q) [B, n, n] (robot_model.py:402-450).

        The reference assembles H from n + 1 inverse-dynamics passes, column j = ID(q, 0, e_j) - ID(q, 0, 0);
        ``include_gravity`` / ``use_damping`` cancel out of that difference (gravity is subtracted, damping
        multiplies qd = 0) and are accepted for signature compatibility only.  One fused
        composite-rigid-body kernel here; differentiable with respect to q and the learnable link parameters
        (n passes of the RNEA backward kernel, one per column).
        """
        assert q.ndim == 2
        assert q.shape[1] == self._n_dofs
        self._require_device()
        dw = self._dynamics_walk()
        ops_f = self._ops_f(dw)
        if torch.is_grad_enabled() and (ops_f.requires_grad or q.requires_grad):
            self._differentiable(dw)
            return _MassMatrix.apply(q, ops_f, dw, self._n_dofs, self._learnable_op_mask(dw))
        out = backend.crba(dw.program, ops_f, dw.ops_i, q, self._n_dofs)
        have = self._fast_crba
        if have is None or have[1]._ws_cache is not have[2]:
            self._fast_crba = self._fast_entry("drm_crba", "drm_crba_scratch_floats_aligned", dw)
        return out

    def compute_forward_dynamics(self, q: torch.Tensor, qd: torch.Tensor, f: torch.Tensor,
                                 include_gravity: Optional[bool] = True, use_damping: Optional[bool] = False
                                 ) -> torch.Tensor:
        """qdd [B, n] (robot_model.py:487-624) — see _compute_forward_dynamics; repeat calls of a constant model go through one
        prepared C++ call (csrc/drm_hostcall.cpp FastCall)."""
        ent = self._fast_fd
        if ent is not None and ent[1]._ws_cache is ent[2]:
            out = ent[0].forward_dynamics(q, qd, f, (1 if include_gravity else 0) | (2 if use_damping else 0), self._stream_arg())
            if out.__class__ is torch.Tensor:
                return out
            if out is not None:
                backend._check(out, backend.library_for(self._device))
        return self._compute_forward_dynamics(q, qd, f, include_gravity, use_damping)

    @tensor_check
    def _compute_forward_dynamics(self, q: torch.Tensor, qd: torch.Tensor, f: torch.Tensor,
                                  include_gravity: Optional[bool] = True, use_damping: Optional[bool] = False
                                  ) -> torch.Tensor:
        """qdd [B,n] that the joint torques ``f`` produce in state (q, qd) (robot_model.py:487-624).

        One kernel launch: Featherstone's articulated-body recursion, as in the reference, for robots with a long
        segment (an arm carrying a gripper or a hand); for 7-DoF arms and for hands (short independent fingers) the same
        linear system H(q) qdd = f - nle(q, qd) formed and solved in registers (composite-rigid-body H, RNEA bias torques,
        leaf-to-root L^T D L elimination).
        With ``use_damping`` the reference subtracts damping * qd from its ``f`` argument IN PLACE
        (robot_model.py:515-521); here ``f`` is left untouched.  Differentiable with respect to q, qd, f and the
        learnable link parameters (implicit differentiation: one more solve + the RNEA backward kernel).
        """
        assert q.ndim == 2
        assert qd.ndim == 2
        assert q.shape[1] == self._n_dofs
        assert qd.shape[1] == self._n_dofs
        self._require_device()
        dw = self._dynamics_walk()
        ops_f = self._ops_f(dw)
        if torch.is_grad_enabled() and (ops_f.requires_grad or any(t.requires_grad for t in (q, qd, f))):
            self._differentiable(dw)
            return _ForwardDynamics.apply(q, qd, f, ops_f, dw, bool(include_gravity), bool(use_damping), self._n_dofs,
                                          self._learnable_op_mask(dw))
        out = backend.forward_dynamics(dw.program, ops_f, dw.ops_i, q, qd, f, bool(include_gravity),
                                       bool(use_damping), self._n_dofs)
        have = self._fast_fd
        if have is None or have[1]._ws_cache is not have[2]:
            self._fast_fd = self._fast_entry("drm_forward_dynamics", "drm_forward_dynamics_scratch_floats_aligned", dw)
        return out

    def compute_forward_dynamics_old(self, q: torch.Tensor, qd: torch.Tensor, f: torch.Tensor,
                                     include_gravity: Optional[bool] = True, use_damping: Optional[bool] = True
                                     ) -> torch.Tensor:
        """qdd = H^-1 (f - nle) (robot_model.py:452-485; upstream it calls ``torch.solve``, which current torch no longer
        has).  The same linear system as ``compute_forward_dynamics`` — note the different default of ``use_damping``."""
        return self.compute_forward_dynamics(q, qd, f, include_gravity=include_gravity, use_damping=use_damping)

    # ------------------------------------------------------------------ learnable parameters
    def _get_parent_object_of_param(self, link_name: str, parameter_name: str):
        body_idx = self._name_to_idx_map[link_name]
        if parameter_name in ["trans", "rot_angles", "joint_damping"]:
            return self._bodies[body_idx]
        if parameter_name in ["mass", "inertia_mat", "com"]:
            return self._bodies[body_idx].inertia
        raise AttributeError(
            "Invalid parameter name. Accepted parameter names are: "
            "trans, rot_angles, joint_damping, mass, inertia_mat, com")

    def make_link_param_learnable(self, link_name: str, parameter_name: str, parametrization: torch.nn.Module):
        """Replace a URDF constant by a learnable module under the same attribute (robot_model.py:682-689)."""
        parent_object = self._get_parent_object_of_param(link_name, parameter_name)
        parent_object.__delattr__(parameter_name)
        parent_object.add_module(parameter_name, parametrization.to(self._device))
        self._learnable.add((self._name_to_idx_map[link_name], parameter_name))
        self._learnable_links = None
        self._source_plan = None
        self._learnable_version += 1
        for dw in self._walks.values():
            dw.static_ops_f = None
            special = getattr(dw.program, "_special", None) or {}
            # kernels that carry the walk table as compile-time constants (specialize.attach_arm / attach_fan / attach_arm_param;
            # attach(table=...)) bake the OLD constants or the OLD set of learnable blocks: dropped — the next call looks again
            baked = getattr(dw.program, "_special_const", False)
            dw.program._special = {} if baked else {k: v for k, v in special.items() if k < 4}
            dw.program._special_const = False
            dw.program._special_tried = False
            dw.program._special_mask = 0
            dw.program._ws_cache = None
        self._arm_specialized = False
        self._fast_fk.clear(); self._fast_jac.clear(); self._fast_fkid.clear()    # (prepared calls snapshot the constants)
        self.__dict__.pop("_fast_links", None)
        self._fast_id = self._fast_crba = self._fast_fd = None
        self._fanout_plans.clear()      # (they may hold folded chain walks, which are for models without learnable parameters)
        self._fan_handles.clear()       # (kernels that bake the OLD constants)
        self._chain_walks.clear()
        self._dyn_walk = None

    def _learnable_module(self, link_name: str, parameter_name: str):
        parent_object = self._get_parent_object_of_param(link_name, parameter_name)
        module = getattr(parent_object, parameter_name)
        assert isinstance(module, torch.nn.Module), f"{parameter_name} of {link_name} is not a learnable module."
        return module

    def freeze_learnable_link_param(self, link_name: str, parameter_name: str):
        for param in self._learnable_module(link_name, parameter_name).parameters():
            param.requires_grad = False

    def unfreeze_learnable_link_param(self, link_name: str, parameter_name: str):
        for param in self._learnable_module(link_name, parameter_name).parameters():
            param.requires_grad = True

    # ------------------------------------------------------------------ introspection
    def get_joint_limits(self) -> List[Dict[str, float]]:
        return [self._bodies[idx].get_joint_limits() for idx in self._controlled_joints]

    def get_link_names(self) -> List[str]:
        return [body.name for body in self._bodies]

    def print_link_names(self) -> None:
        for body in self._bodies:
            print(body.name)

    def print_learnable_params(self) -> None:
        for name, param in self.named_parameters():
            print(f"{name}: {param}")


def _robot_path(rel):
    return os.path.join(robot_description_folder, rel)


class DifferentiableKUKAiiwa(DifferentiableRobotModel):
    def __init__(self, device=None):
        self.urdf_path = _robot_path("iiwa7.urdf")
        self.learnable_rigid_body_config = None
        self.name = "differentiable_kuka_iiwa"
        super().__init__(self.urdf_path, self.name, device=device)


class DifferentiableFrankaPanda(DifferentiableRobotModel):
    def __init__(self, device=None):
        self.urdf_path = _robot_path("panda_no_gripper.urdf")
        self.learnable_rigid_body_config = None
        self.name = "differentiable_franka_panda"
        super().__init__(self.urdf_path, self.name, device=device)


class DifferentiableTwoLinkRobot(DifferentiableRobotModel):
    def __init__(self, device=None):
        self.urdf_path = _robot_path("2link_robot.urdf")
        self.learnable_rigid_body_config = None
        self.name = "diff_2d_robot"
        super().__init__(self.urdf_path, self.name, device=device)


class DifferentiableTrifingerEdu(DifferentiableRobotModel):
    def __init__(self, device=None):
        self.urdf_path = _robot_path("trifinger_edu.urdf")
        self.learnable_rigid_body_config = None
        self.name = "trifinger_edu"
        super().__init__(self.urdf_path, self.name, device=device)
