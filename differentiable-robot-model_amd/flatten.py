"""Flattening of the kinematic tree into the SoA tables the HIP kernels walk.

The reference keeps one ``DifferentiableRigidBody`` per link and loops over
them in Python (reference ``robot_model.py:173-193, 262-301, 651-665``),
re-discovering each parent by a linear name search inside the loop.  Here the
tree is flattened ONCE at model construction:

  RobotSpec      per-link topology in URDF <link> order (parent index, DoF
                 column, axis code/sign) + float32 snapshots of the constants.
  WalkProgram    a depth-first "walk program" over the sub-tree a kernel needs
                 (the root->target chain for FK/Jacobian, the union of chains
                 for multi-target FK, the whole tree for RNEA).  Every op is one
                 link; its parent pose comes either from the previous op
                 (``SRC_PREV``), from the identity root (``SRC_ROOT``) or from a
                 numbered save slot written by an earlier branch-point op.  All
                 of that is wave-uniform data the kernel reads through scalar
                 loads, so one compiled kernel serves every robot.

Device layout of one op (see include/drm_hip.h, DRM_OPF_* / DRM_OPI_*):
  ops_f[k, 0:32] float32: F(9) t(3) mass(1) mcom(3) Io(9) damping(1) pad(6)
  ops_i[k, 0:8 ] int32  : dof axis sign src save out link flags
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

# ---- constants shared with include/drm_hip.h --------------------------------
OPF_STRIDE = 32
OPF_F, OPF_T, OPF_MASS, OPF_MCOM, OPF_IO, OPF_DAMP = 0, 9, 12, 13, 16, 25
OPI_STRIDE = 8
OPI_DOF, OPI_AXIS, OPI_SIGN, OPI_SRC, OPI_SAVE, OPI_OUT, OPI_LINK, OPI_FLAGS = range(8)
SRC_PREV, SRC_ROOT = -1, -2          # >= 0: read parent state from that save slot
FLAG_CHILD_IS_NEXT = 1               # op k+1 is a child of op k (RNEA backward carry)
MAX_SLOTS = 4                        # save slots compiled into the kernels
MAX_OPS = 64                         # largest compiled capacity (ops per walk)
MAX_DOFS = 64                        # DoF columns addressable by one walk


class UnsupportedRobotError(ValueError):
    pass


@dataclass
class RobotSpec:
    link_names: List[str]
    parent: np.ndarray            # int32 [L], -1 for root
    dof: np.ndarray               # int32 [L], -1 = fixed joint
    axis_idx: np.ndarray          # int32 [L], 0/1/2 (2 for fixed joints, like rb.py:149-154's fall-through)
    axis_sign: np.ndarray         # int32 [L], +1/-1 (0 for fixed joints)
    controlled: List[int]         # link index of each DoF column (= reference _controlled_joints)
    # float32 snapshots of the URDF constants (used when nothing is learnable, and by the oracle)
    rpy: np.ndarray               # [L,3]
    trans: np.ndarray             # [L,3]
    axis: np.ndarray              # [L,3]
    damping: np.ndarray           # [L]
    mass: np.ndarray              # [L]
    com: np.ndarray               # [L,3]
    inertia: np.ndarray           # [L,9]
    children: List[List[int]] = field(default_factory=list)

    @property
    def n_links(self):
        return len(self.link_names)

    @property
    def n_dofs(self):
        return len(self.controlled)

    def chain_to(self, link: int) -> List[int]:
        """Links from the root's child down to ``link`` (root excluded)."""
        chain = []
        i = int(link)
        while i > 0:
            chain.append(i)
            i = int(self.parent[i])
        return chain[::-1]


def build_robot_spec(body_params: Sequence[dict], parent_names: Sequence[Optional[str]]) -> RobotSpec:
    """body_params[i] = dict from URDFRobotModel.get_body_parameters_from_urdf(i, link)."""
    L = len(body_params)
    names = [bp["link_name"] for bp in body_params]
    if len(set(names)) != L:
        raise UnsupportedRobotError("duplicate link names in URDF")
    idx = {n: i for i, n in enumerate(names)}
    parent = np.full(L, -1, np.int32)
    dof = np.full(L, -1, np.int32)
    axis_idx = np.full(L, 2, np.int32)
    axis_sign = np.zeros(L, np.int32)
    controlled = []
    f32 = lambda t, shape: np.asarray(t.detach().cpu().numpy(), np.float32).reshape(shape)
    rpy = np.zeros((L, 3), np.float32); trans = np.zeros((L, 3), np.float32)
    axis = np.zeros((L, 3), np.float32); damping = np.zeros(L, np.float32)
    mass = np.zeros(L, np.float32); com = np.zeros((L, 3), np.float32); inertia = np.zeros((L, 9), np.float32)
    for i, bp in enumerate(body_params):
        if i > 0:
            pname = parent_names[i]
            if pname not in idx:
                raise UnsupportedRobotError("link %s: parent link %r not found" % (names[i], pname))
            parent[i] = idx[pname]
            if parent[i] >= i:
                # the reference's loops silently assume parents precede children
                # (robot_model.py:173, 262, 284); make the assumption explicit
                raise UnsupportedRobotError(
                    "link %s appears before its parent %s in the URDF; the reference's sweeps require "
                    "parent-before-child <link> order" % (names[i], pname))
        rpy[i] = f32(bp["rot_angles"], 3); trans[i] = f32(bp["trans"], 3)
        axis[i] = f32(bp["joint_axis"], 3)
        if bp["joint_damping"] is not None:
            damping[i] = f32(bp["joint_damping"], 1)[0]
        mass[i] = f32(bp["mass"], 1)[0]; com[i] = f32(bp["com"], 3); inertia[i] = f32(bp["inertia_mat"], 9)
        if bp["joint_type"] != "fixed":
            dof[i] = len(controlled)
            controlled.append(i)
            a = axis[i]
            nz = np.nonzero(a)[0]
            if len(nz) != 1 or abs(a[nz[0]]) != 1.0:
                # reference: rotation picks x, then y, else z with sign(axis) (rigid_body.py:149-154) and the
                # torque extraction needs exactly one non-zero entry (robot_model.py:356-358)
                raise UnsupportedRobotError(
                    "joint of link %s has axis %s; only +-unit x/y/z axes are supported (the reference "
                    "silently mis-handles anything else)" % (names[i], a.tolist()))
            axis_idx[i] = nz[0]
            axis_sign[i] = 1 if a[nz[0]] > 0 else -1
    children = [[] for _ in range(L)]
    for i in range(1, L):
        children[parent[i]].append(i)
    return RobotSpec(names, parent, dof, axis_idx, axis_sign, controlled,
                     rpy, trans, axis, damping, mass, com, inertia, children)


@dataclass
class WalkProgram:
    links: np.ndarray        # int32 [nops]  link index of every op
    ops_i: np.ndarray        # int32 [nops_padded, OPI_STRIDE]
    n_ops: int
    n_slots: int
    capacity: int            # compiled kernel capacity that fits n_ops
    targets: List[int]       # link index per output slot
    dof_mask: int            # bit d set <=> DoF d is driven by some op of this walk


_CAPACITIES = (8, 16, 32, 64)


def _capacity_for(n_ops: int) -> int:
    for c in _CAPACITIES:
        if n_ops <= c:
            return c
    raise UnsupportedRobotError("walk of %d links exceeds the largest compiled capacity %d" % (n_ops, _CAPACITIES[-1]))


def build_walk(spec: RobotSpec, targets: Optional[Sequence[int]] = None, whole_tree: bool = False) -> WalkProgram:
    """Depth-first walk over the links needed to reach ``targets`` (or all links)."""
    L = spec.n_links
    needed = np.zeros(L, bool)
    if whole_tree:
        needed[1:] = True
    tlist = [int(t) for t in (targets or [])]
    for t in tlist:
        for i in spec.chain_to(t):
            needed[i] = True
    out_of = {}
    for slot, t in enumerate(tlist):
        out_of.setdefault(t, []).append(slot)
    for t, slots in out_of.items():
        if len(slots) > 1:
            raise ValueError("duplicate target link %s" % spec.link_names[t])

    ops = []      # rows of ops_i
    links = []
    free_slots = list(range(MAX_SLOTS))[::-1]
    max_used = 0

    def visit(i, src):
        nonlocal max_used
        kids = [c for c in spec.children[i] if needed[c]]
        save = -1
        if len(kids) > 1:
            if not free_slots:
                raise UnsupportedRobotError(
                    "tree needs more than %d nested branch points; not supported by the compiled kernels" % MAX_SLOTS)
            save = free_slots.pop()
            max_used = max(max_used, MAX_SLOTS - len(free_slots))
        flags = FLAG_CHILD_IS_NEXT if kids else 0
        out = out_of[i][0] if i in out_of else -1
        ops.append([int(spec.dof[i]), int(spec.axis_idx[i]), int(spec.axis_sign[i]), src, save, out, i, flags])
        links.append(i)
        for n, c in enumerate(kids):
            visit(c, SRC_PREV if n == 0 else save)
        if save >= 0:
            free_slots.append(save)

    # the root (link 0) has the identity pose and is never an op; its needed
    # children all read SRC_ROOT, so a root with several children costs no slot
    for c in spec.children[0]:
        if needed[c]:
            visit(c, SRC_ROOT)
    # a target that IS the root has no op: handled by the caller (identity pose)
    n_ops = len(ops)
    cap = _capacity_for(max(n_ops, 1))
    ops_i = np.zeros((cap + 1, OPI_STRIDE), np.int32)
    ops_i[:, OPI_DOF] = -1
    ops_i[:, OPI_SRC] = SRC_ROOT
    ops_i[:, OPI_SAVE] = -1
    ops_i[:, OPI_OUT] = -1
    if n_ops:
        ops_i[:n_ops] = np.asarray(ops, np.int32)
    mask = 0
    for row in ops:
        if row[OPI_DOF] >= 0:
            mask |= 1 << row[OPI_DOF]
    if spec.n_dofs > MAX_DOFS:
        raise UnsupportedRobotError("%d DoFs exceed the supported maximum %d" % (spec.n_dofs, MAX_DOFS))
    return WalkProgram(np.asarray(links, np.int32), ops_i, n_ops, max_used, cap, tlist, mask)
