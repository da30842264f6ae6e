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
                 link; its parent state comes from the previous op
                 (``SRC_PREV``), from the identity root (``SRC_ROOT``) or from a
                 numbered save slot written by an earlier branch-point op.  The
                 walk is padded with identity ops up to a compiled capacity so
                 the kernels run it as straight-line code.

Axis canonicalisation (exact, no rounding): the reference rotates a joint about
its local x, y or z axis by sign(axis) * q (rigid_body.py:149-154).  With the
signed permutation P = P_a D_s (P_a cyclic with P_a e_z = e_a; D_+ = I,
D_- = diag(1,-1,-1) = Rot_x(pi), so P e_z = s e_a) one has
Rot_a(s q) = P Rot_z(q) P^T, so storing each link frame as R~_i = R_i P_i turns
every joint into a rotation about local +z by +q:
    R~_i = R~_p (P_p^T F_i P_i) Rot_z(q),   p_i = R~_p (P_p^T t_i) + p_p.
Permuting / negating rows and columns of the constants is exact in floating
point, so the kernels compute the same products without any per-axis branch or
sign; body-frame quantities (com, inertia) are transformed the same way, and a
target link's columns are restored when its pose is emitted.

Device layout of one op (see include/drm_hip.h, DRM_OPF_* / DRM_OPI_*):
  ops_f[k, 0:32] float32: FT block(12: F and t interleaved in pairs) mass(1) mcom(3) Io(9) damping(1) pad(6)
  ops_i[0:8, k ] int32  : dof perm ctrl src save out link flags   (FIELD-MAJOR on the device; the kernels read
                          only `ctrl`, the packed control word, one wide scalar load for the whole walk)
``gather`` maps every ops_f entry to a flat index of the [L+1, 32] link table
(row L = the identity op) and ``gsign`` holds its +-1 factor, so the device table
is ONE differentiable gather and one multiply.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# ---- constants shared with include/drm_hip.h --------------------------------
OPF_STRIDE = 32
# link table ([L+1, 32], built by robot_model._link_table): F(9) t(3) mass mcom(3) Io(9) damping
OPF_F, OPF_T, OPF_MASS, OPF_MCOM, OPF_IO, OPF_DAMP = 0, 9, 12, 13, 16, 25
OPF_FT_FLOATS = 12


# op rows of a walk keep that layout except for the first 12 floats, which interleave F and t as the
# 8-byte pairs the packed-FP32 chain kernel consumes: (F00 F01)(F10 F11)(F20 F21)(F02 t0)(F12 t1)(F22 t2)
def opf_fij(i: int, j: int) -> int:
    return 2 * i + j if j < 2 else 6 + 2 * i


def opf_ti(i: int) -> int:
    return 7 + 2 * i


SHAPE_ARM_CHAIN = 1                  # drm_walk.shape bit, see include/drm_hip.h
SHAPE_SERIAL_CHAIN = 2               # DRM_WALK_SERIAL_CHAIN
SHAPE_ARM_HAND = 4                   # DRM_WALK_ARM_HAND (+ P, K, L in the top byte, DRM_WALK_AH_PACK)
SHAPE_TARGETS_ORDERED = 8            # DRM_WALK_TARGETS_ORDERED: output slots 0, 1, 2, ... in walk order
SHAPE_FINGERS = 16                   # DRM_WALK_FINGERS (+ K, L in the top byte)
SHAPE_NO_PRISMATIC = 32              # DRM_WALK_NO_PRISMATIC: no op of the walk slides
SHAPE_CHAIN_DOFS = 128               # DRM_WALK_CHAIN_DOFS: chain_dof1 / chain_prismatic of a serial chain of <= 16 ops are filled
SHAPE_FK_FAN = 64                    # DRM_WALK_FK_FAN: a many-target FK walk split behind a hub (prefix_end / seg_begin)
FK_FAN_WAVES = 4                     # wavefronts a fanned-out many-target FK block has at most


def fk_fan_partition(ops, parent_op, n_ops: int):
    """Where a many-target FK walk splits: (P, [begin_0, .., begin_K = n_ops]) such that the ops [0, P) are what every later op
    hangs off — they end at a HUB (the root itself: P = 0; a palm; a torso) whose sub-tree reaches to the end of the walk — and
    the hub's children's sub-trees, consecutive ones packed into K <= FK_FAN_WAVES runs, are [begin_j, begin_j+1).  Every run but
    the first starts at an op that reads its parent's pose from a save slot or the root (a later child of the hub), so a
    wavefront can walk [0, P) and then its run without hearing from the others.  None when no hub makes the longest wavefront
    (P + its run) at least 1.3x shorter than the walk, when the repeats of the shared part would more than double the block's
    work, or when two runs would write the same save slot."""
    if n_ops < 8:
        return None
    # the wavefronts of a block share the save slots without synchronising: every slot may be written by ONE op of the walk only
    # (a walk that reuses a slot once a sub-tree is done is only safe in sequence) — checked here, not left to the caller
    written = [ops[k][OPI_SAVE] for k in range(n_ops) if ops[k][OPI_SAVE] >= 0]
    if len(written) != len(set(written)):
        return None
    end = list(range(n_ops))                      # last op of every op's sub-tree (depth-first order: a contiguous range)
    for k in range(n_ops - 1, -1, -1):
        if parent_op[k] >= 0:
            end[parent_op[k]] = max(end[parent_op[k]], end[k])
    kids = {}
    for k in range(n_ops):
        kids.setdefault(parent_op[k], []).append(k)
    best = None
    for hub, ch in kids.items():
        if len(ch) < 2 or (hub >= 0 and end[hub] != n_ops - 1):
            continue
        P = hub + 1 if hub >= 0 else 0
        if ch[0] != P or any(ops[c][OPI_SRC] == SRC_PREV for c in ch[1:]):
            continue
        sizes = [end[c] - c + 1 for c in ch]
        limit = max(sizes)
        while True:
            begins, acc = [ch[0]], 0
            for c, sz in zip(ch, sizes):
                if acc and acc + sz > limit:
                    begins.append(c)
                    acc = 0
                acc += sz
            if len(begins) <= FK_FAN_WAVES:
                break
            limit += 1
        if len(begins) < 2:
            continue
        begins.append(n_ops)
        # the wavefronts share the save slots: a slot written inside one run must not be one another run (or the shared part)
        # writes too — the walk reuses slots once a sub-tree is done, which is only safe in sequence
        saves = [set(ops[k][OPI_SAVE] for k in range(a, b) if ops[k][OPI_SAVE] >= 0) for a, b in zip([0] + begins[:-1], [P] + begins[1:])]
        if any(saves[i] & saves[j] for i in range(len(saves)) for j in range(i + 1, len(saves))):
            continue
        # every wavefront repeats the shared part: not when that more than doubles the work of the block
        if len(begins[:-1]) * P + (n_ops - P) > 2 * n_ops:
            continue
        cost = P + max(b - a for a, b in zip(begins, begins[1:]))
        if best is None or cost < best[0]:
            best = (cost, P, begins)
    if best is None or best[0] * 1.3 > n_ops:
        return None
    return best[1], best[2]


def fingers_shape(ops, parent_op, n_ops: int, n_dofs: int, prismatic) -> int:
    """DRM_WALK_FINGERS | K, L when the walk is K (2..4) serial chains of L (2..4) revolute ops each, every chain hanging off
    the root, op k driving DoF column k — a hand whose fixed joints are folded away (Allegro 4 x 4, TriFinger 3 x 3); else 0."""
    if n_ops != n_dofs or n_ops < 4 or any(prismatic):
        return 0
    heads = [k for k in range(n_ops) if parent_op[k] == -1]
    K = len(heads)
    if not (2 <= K <= 4) or n_ops % K or heads[0] != 0:
        return 0
    L = n_ops // K
    if not (2 <= L <= 4) or heads != [j * L for j in range(K)]:
        return 0
    for k in range(n_ops):
        if ops[k][OPI_DOF] != k or (k % L and parent_op[k] != k - 1) or ops[k][OPI_SAVE] >= 0:
            return 0
    return SHAPE_FINGERS | ((K - 1) << 28) | ((L - 1) << 30)


def arm_hand_shape(parent_op, n_ops: int, prismatic) -> int:
    """DRM_WALK_AH_PACK(P, K, L) when the walk is a serial prefix of P ops carrying K >= 2 serial sub-chains of L ops each, all
    hanging off op P-1 (an arm with its gripper or hand; prismatic joints welcome), else 0."""
    if n_ops < 3 or parent_op[0] != -1:
        return 0
    breaks = [k for k in range(1, n_ops) if parent_op[k] != k - 1]
    if not breaks:
        return 0
    hub = parent_op[breaks[0]]
    if hub < 0 or any(parent_op[k] != hub for k in breaks):
        return 0
    P = hub + 1
    K = 1 + len(breaks)
    if (n_ops - P) % K:
        return 0
    L = (n_ops - P) // K
    if not (1 <= L <= 4 and 2 <= K <= 4 and P <= 15) or breaks != [P + j * L for j in range(1, K)]:
        return 0
    return SHAPE_ARM_HAND | (P << 24) | ((K - 1) << 28) | ((L - 1) << 30)
OPI_STRIDE = 10
OPI_DOF, OPI_PERM, OPI_CTRL, OPI_SRC, OPI_SAVE, OPI_OUT, OPI_LINK, OPI_FLAGS, OPI_W0, OPI_W1 = range(10)
SRC_PREV, SRC_ROOT = -1, -2          # >= 0: read parent state from that save slot
FLAG_CHILD_IS_NEXT = 1               # op k+1 is a child of op k (RNEA backward carry)
MAX_SLOTS = 16                       # save slots available to a walk (forward kernels)
MAX_SLOTS_BACKWARD = 6               # ... and to the backward kernels (the 3-bit source field of the packed control word addresses slots 0 .. 5)
MAX_OPS_BACKWARD = 64                # largest walk the backward kernels take (DRM_MAX_OPS)
MAX_SEGMENTS = 8                     # independent root-level sub-walks a dynamics launch fans out over (DRM_MAX_SEGMENTS)
MAX_DOFS = 64                        # DoF columns addressable by one walk
KIND_FIXED, KIND_REVOLUTE, KIND_PRISMATIC = 0, 1, 2

# _PERM[a][c] = index pi_a(c) with P_a e_c = e_{pi_a(c)};  (M P_a)[:, c] = M[:, pi_a(c)]
_PERM = {0: (1, 2, 0), 1: (2, 0, 1), 2: (0, 1, 2)}


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
    # float32 snapshots of the URDF constants (used by the oracle and by tests)
    rpy: np.ndarray               # [L,3]
    trans: np.ndarray             # [L,3]
    axis: np.ndarray              # [L,3]
    damping: np.ndarray           # [L]
    mass: np.ndarray              # [L]
    com: np.ndarray               # [L,3]
    inertia: np.ndarray           # [L,9]
    children: List[List[int]] = field(default_factory=list)
    kind: Optional[np.ndarray] = None       # int32 [L]: KIND_FIXED / KIND_REVOLUTE / KIND_PRISMATIC
    skew: Optional[np.ndarray] = None       # bool [L]: the joint axis is not +-x / y / z (axis_rot carries it)
    axis_rot: Optional[np.ndarray] = None   # float32 [L,3,3]: for skew axes, a rotation R_a with R_a e_z = axis

    def __post_init__(self):
        L = len(self.link_names)
        if self.kind is None:
            self.kind = np.where(self.dof >= 0, KIND_REVOLUTE, KIND_FIXED).astype(np.int32)
        if self.skew is None:
            self.skew = np.zeros(L, bool)
        if self.axis_rot is None:
            self.axis_rot = np.tile(np.eye(3, dtype=np.float32), (L, 1, 1))

    @property
    def skew_links(self) -> List[int]:
        return [int(i) for i in np.nonzero(self.skew)[0]]

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

    def preorder(self) -> List[int]:
        """Links in the depth-first order build_walk visits them (the root first)."""
        out, stack = [], [0]
        while stack:
            i = stack.pop()
            out.append(i)
            stack.extend(reversed(self.children[i]))
        return out

    def perm_of(self, link: int):
        """Signed column permutation (pi, d) of the stored frame of ``link``: (M P)[:, c] = d[c] * M[:, pi[c]]
        (identity for fixed joints / root)."""
        if link <= 0 or self.dof[link] < 0 or self.skew[link]:
            return _PERM[2], (1, 1, 1)   # (a skew-axis link hands its TRUE frame to its children: op B of build_walk)
        s = int(self.axis_sign[link])
        return _PERM[int(self.axis_idx[link])], (1, s, s)

    def perm_code(self, link: int) -> int:
        """DRM_OPI_PERM code of ``link``: axis index (2 for fixed joints) + 3 if the axis is negative."""
        if link <= 0 or self.dof[link] < 0 or self.skew[link]:
            return 2
        return int(self.axis_idx[link]) + (3 if self.axis_sign[link] < 0 else 0)


def axis_rotation(axis: np.ndarray) -> np.ndarray:
    """A rotation R_a (float64) with R_a e_z = axis / |axis|: the smallest one (about e_z x axis)."""
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    v = np.cross([0.0, 0.0, 1.0], a)
    c = a[2]
    if c < -1.0 + 1e-12:                       # axis = -e_z: half a turn about x
        return np.diag([1.0, -1.0, -1.0])
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + K + K @ K / (1.0 + c)


def build_robot_spec(body_params: Sequence[dict], parent_names: Sequence[Optional[str]],
                     reference_compat: bool = False) -> RobotSpec:
    """body_params[i] = dict from URDFRobotModel.get_body_parameters_from_urdf(i, link).

    Joint models: "revolute" / "continuous" turn about their axis, "prismatic" slide along it, any
    unit axis is accepted.  The reference models every non-fixed joint as a revolute one about +-x / y / z
    (robot_model.py:122-126, rigid_body.py:133,149-154) — wrong for the gripper fingers of panda.urdf;
    ``reference_compat=True`` reproduces that (prismatic = revolute) for parity runs against it."""
    L = len(body_params)
    names = [bp["link_name"] for bp in body_params]
    if len(set(names)) != L:
        raise UnsupportedRobotError("duplicate link names in URDF")
    idx = {n: i for i, n in enumerate(names)}
    parent = np.full(L, -1, np.int32)
    dof = np.full(L, -1, np.int32)
    axis_idx = np.full(L, 2, np.int32)
    axis_sign = np.zeros(L, np.int32)
    kind = np.zeros(L, np.int32)
    skew = np.zeros(L, bool)
    axis_rot = np.tile(np.eye(3, dtype=np.float32), (L, 1, 1))
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
        jtype = bp["joint_type"]
        if jtype != "fixed":
            if jtype not in ("revolute", "continuous", "prismatic") and not reference_compat:
                # (under reference_compat EVERY joint that is not `fixed` — floating and planar ones included — is one revolute DoF
                # about its <axis>, as upstream counts and moves them, robot_model.py:122-126, rigid_body.py:130-157)
                raise UnsupportedRobotError("joint of link %s has type %r; with reference_compat=False the supported types are "
                                            "fixed, revolute, continuous, prismatic" % (names[i], jtype))
            dof[i] = len(controlled)
            controlled.append(i)
            kind[i] = KIND_PRISMATIC if (jtype == "prismatic" and not reference_compat) else KIND_REVOLUTE
            a = axis[i]
            nz = np.nonzero(a)[0]
            if len(nz) == 1 and abs(a[nz[0]]) == 1.0:
                # +-x / y / z: folded into an exact signed permutation of the constants (see the module docstring)
                axis_idx[i] = nz[0]
                axis_sign[i] = 1 if a[nz[0]] > 0 else -1
            else:
                norm = float(np.linalg.norm(a.astype(np.float64)))
                if reference_compat or not np.isfinite(norm) or norm < 1e-6:
                    # the reference rotates such a joint about z by sign(axis_z) q and cannot extract its torque
                    # (rigid_body.py:149-154, robot_model.py:356-358): there is nothing meaningful to reproduce
                    raise UnsupportedRobotError(
                        "joint of link %s has axis %s; the reference only handles +-unit x/y/z axes%s"
                        % (names[i], a.tolist(), " (pass reference_compat=False to the model for joints about their true axis)"
                           if reference_compat and np.isfinite(norm) and norm >= 1e-6 else " and a zero axis has no direction"))
                skew[i] = True
                axis_rot[i] = axis_rotation(a).astype(np.float32)
                axis_idx[i] = 2
                axis_sign[i] = 1
    children = [[] for _ in range(L)]
    for i in range(1, L):
        children[parent[i]].append(i)
    return RobotSpec(names, parent, dof, axis_idx, axis_sign, controlled,
                     rpy, trans, axis, damping, mass, com, inertia, children, kind, skew, axis_rot)


@dataclass
class WalkProgram:
    links: np.ndarray        # int32 [n_ops]  link index of every op
    ops_i: np.ndarray        # int32 [capacity, OPI_STRIDE]   (identity-padded, op-major: host-side view)
    ops_i_dev: np.ndarray    # int32 [OPI_STRIDE, capacity]   field-major copy = the device layout
    gather: np.ndarray       # int64 [capacity, OPF_STRIDE]   flat indices into the [L+1, OPF_STRIDE] link table
    gsign: np.ndarray        # float32 [capacity, OPF_STRIDE] +-1 factor of every gathered entry
    n_ops: int
    n_slots: int
    capacity: int            # compiled kernel capacity that fits n_ops
    targets: List[int]       # link index per output slot
    dof_mask: int            # bit d set <=> DoF d is driven by some op of this walk
    slots_unique: bool = True  # every branch point owns its slot for the whole walk (needed by the backward walk)
    shape: int = 0           # SHAPE_* bits (drm_walk.shape)
    seg_begin: Sequence[int] = (0, 0)     # op ranges of the independent root-level sub-walks (drm_walk.seg_begin)
    seg_dof: Sequence[tuple] = ((0, 0),)  # (first DoF column, count) of every segment
    op_of_link: Optional[dict] = None     # link index -> op that carries the link's TRUE frame (targets, body forces)
    prefix_end: int = 0                   # ops [0, prefix_end) are static (fixed joints off the root): every segment replays them
    seg_leaf_begin: Sequence[int] = (0, 0)  # leaf ordinals of every segment (drm_walk.seg_leaf_begin)
    chain_dof1: Sequence[int] = (0,) * 16   # SHAPE_CHAIN_DOFS: 1 + DoF column of op k, 0 = does not move (drm_walk.chain_dof1)
    chain_prismatic: int = 0                # SHAPE_CHAIN_DOFS: bit k <=> op k slides

    @property
    def n_segments(self) -> int:
        return len(self.seg_begin) - 1

    @property
    def backward_ok(self) -> bool:
        """The backward kernels read the 3-bit fields of the packed control word and park one record per op."""
        return self.n_slots <= MAX_SLOTS_BACKWARD and self.capacity <= MAX_OPS_BACKWARD and len(self.targets) <= 126


def _capacity_for(n_ops: int) -> int:
    """Rows of the op tables: 4 or 8 for short walks (the 7-DoF arm kernels are compiled for 8), else n_ops rounded up to
    a multiple of 4 (16-byte rows of the field-major int table)."""
    if n_ops <= 4:
        return 4
    if n_ops <= 8:
        return 8
    return (n_ops + 3) // 4 * 4


def _gather_row(spec: RobotSpec, link: int, parent: Optional[int] = None):
    """(flat link-table indices, +-1 factors) of one op's constants, with the axis canonicalisation applied.  ``parent``: the
    link whose frame the row's F / t are expressed in (default: the link's parent)."""
    base = link * OPF_STRIDE
    pp, dp = spec.perm_of(int(spec.parent[link]) if parent is None else int(parent))   # rows of F / t follow the parent's stored frame
    pi, di = spec.perm_of(link)                     # columns of F and body-frame quantities follow this link's
    row = np.empty(OPF_STRIDE, np.int64)
    sgn = np.ones(OPF_STRIDE, np.float32)
    for r in range(3):
        for c in range(3):
            row[opf_fij(r, c)] = base + OPF_F + pp[r] * 3 + pi[c]
            sgn[opf_fij(r, c)] = dp[r] * di[c]
            row[OPF_IO + r * 3 + c] = base + OPF_IO + pi[r] * 3 + pi[c]
            sgn[OPF_IO + r * 3 + c] = di[r] * di[c]
        row[opf_ti(r)] = base + OPF_T + pp[r]
        sgn[opf_ti(r)] = dp[r]
        row[OPF_MCOM + r] = base + OPF_MCOM + pi[r]
        sgn[OPF_MCOM + r] = di[r]
    row[OPF_MASS] = base + OPF_MASS
    for k in range(OPF_DAMP, OPF_STRIDE):
        row[k] = base + k
    return row, sgn


def virtual_rows(spec: RobotSpec):
    """Link-table rows beyond the identity row L: two per skew-axis link, (L + 1 + 2 j, L + 2 + 2 j) for the j-th one."""
    return {link: (spec.n_links + 1 + 2 * j, spec.n_links + 2 + 2 * j) for j, link in enumerate(spec.skew_links)}


def _gather_plain(row_index: int):
    """An op whose constants are one link-table row as it stands (op-row layout, no permutation)."""
    base = row_index * OPF_STRIDE
    row = base + np.arange(OPF_STRIDE, dtype=np.int64)
    for r in range(3):
        for c in range(3):
            row[opf_fij(r, c)] = base + OPF_F + r * 3 + c
        row[opf_ti(r)] = base + OPF_T + r
    return row, np.ones(OPF_STRIDE, np.float32)


def _gather_axis_op(spec: RobotSpec, link: int, row_index: int, parent: Optional[int] = None):
    """Op A of a skew-axis link: virtual row `row_index` (F R_a, t), its rows following the parent's stored frame."""
    row, sgn = _gather_plain(row_index)
    base = row_index * OPF_STRIDE
    pp, dp = spec.perm_of(int(spec.parent[link]) if parent is None else int(parent))
    for r in range(3):
        for c in range(3):
            row[opf_fij(r, c)] = base + OPF_F + pp[r] * 3 + c
            sgn[opf_fij(r, c)] = dp[r]
        row[opf_ti(r)] = base + OPF_T + pp[r]
        sgn[opf_ti(r)] = dp[r]
    return row, sgn


def foldable_links(spec: RobotSpec, keep: Sequence[int] = ()) -> np.ndarray:
    """bool [L]: links behind a FIXED joint — end-effector frames, fingertips, sensor mounts, and the plates, flanges and palms
    between moving joints.  Such a link is rigidly attached to its nearest non-foldable ancestor (its fold target), so a
    dynamics walk may leave it out when the target's row carries its inertia as well and the rows of the links below it
    carry its transform (fold_link_table): same torques / inertia matrix / accelerations, one op less per such link
    (Panda: 8 -> 7 ops, Panda with gripper: 12 -> 9, Allegro: 20 -> 16, Fetch: 24 -> 14).
    ``keep``: links that must stay ops of their own — those with learnable parameters (their constants change every step
    and their gradients are theirs).  A link whose fold target is such a link stays as well (the target's row is rebuilt
    from its parameters alone), and so does the fixed parent of such a link (the link's own row is rebuilt from ITS
    parameters alone and could not carry the parent's transform)."""
    L = spec.n_links
    keep = set(int(i) for i in keep)
    while True:
        fold = np.zeros(L, bool)
        for i in range(1, L):
            fold[i] = spec.kind[i] == KIND_FIXED and i not in keep and not spec.skew[i]
        bad = set()
        for i in np.nonzero(fold)[0]:
            if effective_parent(spec, fold, int(i)) in keep and spec.parent[i] >= 0:
                bad.add(int(i))
        for i in keep:
            if 0 < i < L and spec.parent[i] > 0 and fold[spec.parent[i]]:
                bad.add(int(spec.parent[i]))
        if not bad:
            return fold
        keep |= bad


def effective_parent(spec: RobotSpec, fold: np.ndarray, link: int) -> int:
    """The nearest ancestor of ``link`` that is not folded (0: the root)."""
    t = int(spec.parent[link])
    while t > 0 and fold[t]:
        t = int(spec.parent[t])
    return t


def fold_link_table(spec: RobotSpec, table: np.ndarray, fold: Optional[np.ndarray] = None) -> np.ndarray:
    """The [L+1(+...), 32] link table with the inertia of every foldable link moved into its parent's row (composite
    rigid body, expressed in the parent link's frame): m' = m_p + m,  (mc)' = (mc)_p + F (mc) + m t,
    I_o' = I_o,p + F I_o F^T + m [(t.t) E - t t^T] + 2 (h.t) E - h t^T - t h^T  with h = F (mc), and with the transforms of
    the folded links between a link and its nearest non-folded ancestor composed into THAT link's F / t
    (x_anc = F_f (F_c x + t_c) + t_f).  fp64 on the host, once per robot.  The folded rows keep their own F / t (FK to such a
    link, walked through its folded ancestors, is unchanged); a foldable link hanging off the root loses its inertia."""
    out = np.array(table, np.float64, copy=True)
    if fold is None:
        fold = foldable_links(spec)
    orig = out.copy()
    for i in range(spec.n_links - 1, 0, -1):      # children come after their parents in URDF <link> order
        if not fold[i]:
            continue
        row = out[i]
        p = int(spec.parent[i])
        F, t = row[OPF_F:OPF_F + 9].reshape(3, 3), row[OPF_T:OPF_T + 3]
        m, mc, Io = row[OPF_MASS], row[OPF_MCOM:OPF_MCOM + 3].copy(), row[OPF_IO:OPF_IO + 9].reshape(3, 3).copy()
        if p > 0:
            h = F @ mc
            E = np.eye(3)
            out[p, OPF_MASS] += m
            out[p, OPF_MCOM:OPF_MCOM + 3] += h + m * t
            out[p, OPF_IO:OPF_IO + 9] += (F @ Io @ F.T + m * ((t @ t) * E - np.outer(t, t))
                                          + 2.0 * (h @ t) * E - np.outer(h, t) - np.outer(t, h)).reshape(9)
        row[OPF_MASS] = 0.0
        row[OPF_MCOM:OPF_MCOM + 3] = 0.0
        row[OPF_IO:OPF_IO + 9] = 0.0
    for i in range(1, spec.n_links):
        if fold[i]:
            continue
        F, t = orig[i, OPF_F:OPF_F + 9].reshape(3, 3).copy(), orig[i, OPF_T:OPF_T + 3].copy()
        p = int(spec.parent[i])
        while p > 0 and fold[p]:
            Fp, tp = orig[p, OPF_F:OPF_F + 9].reshape(3, 3), orig[p, OPF_T:OPF_T + 3]
            F, t = Fp @ F, Fp @ t + tp
            p = int(spec.parent[p])
        out[i, OPF_F:OPF_F + 9] = F.reshape(9)
        out[i, OPF_T:OPF_T + 3] = t
    return out


def build_walk(spec: RobotSpec, targets: Optional[Sequence[int]] = None, whole_tree: bool = False,
               min_capacity: int = 0, drop_folded: bool = False, fold: Optional[np.ndarray] = None) -> WalkProgram:
    """Depth-first walk over the links needed to reach ``targets`` (or all links); ``min_capacity`` pads it to at least
    that capacity (chains that are launched together share one).  ``drop_folded``: a whole-tree walk without the
    foldable links (``fold``, default foldable_links(spec); to be run on a table from fold_link_table with the same mask).

    A link whose joint axis is not +-x / y / z becomes TWO ops (Rot_a(q) = R_a Rot_z(q) R_a^T with R_a e_z = a):
      A  the joint: fixed part F R_a, trans t, moving about +z, massless (virtual link-table row);
      B  a fixed joint with F = R_a^T, t = 0 that carries the link's own frame, mass and inertia; the link's children
         and any output slot hang off B.
    Both are exact restatements of x_parent = F Rot_a(q) x_child + t, so no kernel knows about general axes."""
    L = spec.n_links
    needed = np.zeros(L, bool)
    if drop_folded and fold is None:
        fold = foldable_links(spec)
    if whole_tree:
        needed[1:] = ~fold[1:] if drop_folded else True
    tlist = [int(t) for t in (targets or [])]
    if fold is not None and not whole_tree and len(tlist) != 1:
        raise ValueError("a walk on a folded link table is the whole tree or the chain to one target")
    for t in tlist:
        chain = spec.chain_to(t)
        for n, i in enumerate(chain):
            # on a folded table a non-folded link's row carries the transforms of the folded links between it and its nearest
            # non-folded ancestor: those are not walked again (the folded links at the END of the chain kept their own rows)
            needed[i] = fold is None or not fold[i] or all(fold[c] for c in chain[n + 1:])
    dropped = np.zeros(L, bool) if fold is None else (fold & ~needed)   # folded links the walk steps over

    def kids_of(i):
        out = []
        for c in spec.children[i]:
            if needed[c]:
                out.append(c)
            elif dropped[c]:
                out += kids_of(c)
        return out

    def frame_parent(i):
        """The link whose frame row i's F / t are expressed in: the nearest ancestor that is not stepped over."""
        t = int(spec.parent[i])
        while t > 0 and dropped[t]:
            t = int(spec.parent[t])
        return t
    out_of = {}
    for slot, t in enumerate(tlist):
        if t in out_of:
            raise ValueError("duplicate target link %s" % spec.link_names[t])
        out_of[t] = slot
    vrows = virtual_rows(spec)

    # op record: [dof, perm, ctrl, src, save, out, link, flags, w0, w1] + (parent op, prismatic, gather kind)
    ops, links, parent_op, prismatic, gkind = [], [], [], [], []
    # slots are handed out fresh while there are any (so that a slot keeps its branch point's state for the
    # whole walk, which the backward walk relies on) and only recycled once all MAX_SLOTS have been used
    free_slots = []
    max_used = 0
    unique = True
    op_of_link = {}

    def emit(i, dof, perm, src, save, out, flags, par, pris, gk):
        ops.append([dof, perm, 0, src, save, out, i, flags, 0, 0])
        links.append(i); parent_op.append(par); prismatic.append(pris); gkind.append(gk)
        return len(ops) - 1

    def visit(i, src, par):
        nonlocal max_used, unique
        kids = kids_of(i)
        save = -1
        if len(kids) > 1:
            if max_used < MAX_SLOTS:
                save = max_used
                max_used += 1
            elif free_slots:
                save = free_slots.pop()
                unique = False
            else:
                raise UnsupportedRobotError(
                    "tree needs more than %d nested branch points; not supported by the kernels" % MAX_SLOTS)
        flags = FLAG_CHILD_IS_NEXT if kids else 0
        pris = int(spec.kind[i] == KIND_PRISMATIC)
        if spec.skew[i]:
            a = emit(i, int(spec.dof[i]), 2, src, -1, -1, FLAG_CHILD_IS_NEXT, par, pris, ("axis", vrows[i][0]))
            me = emit(i, -1, 2, SRC_PREV, save, out_of.get(i, -1), flags, a, 0, ("plain", vrows[i][1]))
        else:
            me = emit(i, int(spec.dof[i]), spec.perm_code(i), src, save, out_of.get(i, -1), flags, par, pris, ("link", i))
        op_of_link[i] = me
        for n, c in enumerate(kids):
            visit(c, SRC_PREV if n == 0 else save, me)
        if save >= 0:
            free_slots.append(save)

    # the root (link 0) has the identity pose and is never an op; its needed
    # children all read SRC_ROOT, so a root with several children costs no slot
    for c in kids_of(0):
        visit(c, SRC_ROOT, -1)
    # a target that IS the root has no op: handled by the caller (identity pose)
    n_ops = len(ops)
    if spec.n_dofs > MAX_DOFS:
        raise UnsupportedRobotError("%d DoFs exceed the supported maximum %d" % (spec.n_dofs, MAX_DOFS))
    if n_ops > 0xfffe or len(tlist) > 0xfffe:
        raise UnsupportedRobotError("walk of %d links / %d targets exceeds the 16-bit fields of the control words" % (n_ops, len(tlist)))
    cap = max(_capacity_for(max(n_ops, 1)), int(min_capacity))
    # identity padding: fixed joint, F = I, t = 0, mass-less, chained to the previous op
    ops_i = np.zeros((cap, OPI_STRIDE), np.int32)
    ops_i[:, OPI_DOF] = -1
    ops_i[:, OPI_PERM] = 2
    ops_i[:, OPI_SRC] = SRC_PREV
    ops_i[:, OPI_SAVE] = -1
    ops_i[:, OPI_OUT] = -1
    ops_i[:, OPI_LINK] = -1
    # identity padding rows gather the identity link-table row THROUGH the op-row layout
    ident, _ = _gather_plain(L)
    gather = np.tile(ident, (cap, 1))
    gsign = np.ones((cap, OPF_STRIDE), np.float32)
    par_arr = np.full(cap, -1, np.int64)
    pris_arr = np.zeros(cap, np.int64)
    if n_ops:
        ops_i[:n_ops] = np.asarray(ops, np.int32)
        par_arr[:n_ops] = parent_op
        pris_arr[:n_ops] = prismatic
        for k, (gk, arg) in enumerate(gkind):
            if gk == "link":
                gather[k], gsign[k] = _gather_row(spec, arg, frame_parent(arg))
            elif gk == "axis":
                gather[k], gsign[k] = _gather_axis_op(spec, links[k], arg, frame_parent(links[k]))
            else:
                gather[k], gsign[k] = _gather_plain(arg)
    else:
        ops_i[0, OPI_SRC] = SRC_ROOT
    padding = np.zeros(cap, np.int64)
    padding[n_ops:] = 1   # (a walk to the root itself has n_ops = 0: all padding)
    # the packed control word the backward and arm kernels read (DRM_OPI_CTRL_PACK in include/drm_hip.h) ...
    ops_i[:, OPI_CTRL] = (((ops_i[:, OPI_DOF] + 1) & 0x7f) | (((ops_i[:, OPI_SRC] + 2) & 7) << 7)
                          | (((ops_i[:, OPI_SAVE] + 1) & 7) << 10) | (((ops_i[:, OPI_OUT] + 1) & 0x7f) << 13)
                          | ((ops_i[:, OPI_PERM] & 7) << 20) | ((ops_i[:, OPI_FLAGS] & 1) << 23))
    ops_i[n_ops:, OPI_CTRL] |= 1 << 24
    ops_i[:, OPI_CTRL] |= (pris_arr << 25).astype(np.int32)
    # bits 26..31: ordinal of a LEAF op (no child follows it in the walk) among the leaves — the loop-structured RNEA backward
    # kernel parks a motion / force-adjoint record per leaf only (drm_sample.hpp rnea_backward_walk)
    n_leaves = 0
    leaf_ord = np.zeros(cap, np.int64)
    for k in range(n_ops):
        if not (ops_i[k, OPI_FLAGS] & FLAG_CHILD_IS_NEXT):
            leaf_ord[k] = n_leaves & 63
            n_leaves += 1
    ops_i[:, OPI_CTRL] = ((ops_i[:, OPI_CTRL].astype(np.int64) & 0x3ffffff) | (leaf_ord << 26)).astype(np.uint32).view(np.int32)
    # ... and the two wide control words of the loop-structured forward kernels (DRM_W0_PACK / DRM_W1_PACK)
    i64 = lambda col: ops_i[:, col].astype(np.int64)
    w0 = (((i64(OPI_DOF) + 1) & 0xff) | (((i64(OPI_SRC) + 2) & 0xff) << 8) | (((i64(OPI_SAVE) + 1) & 0xff) << 16)
          | ((i64(OPI_FLAGS) & 1) << 24) | (padding << 25) | (pris_arr << 26) | ((i64(OPI_PERM) & 7) << 27))
    w1 = ((i64(OPI_OUT) + 1) & 0xffff) | (((par_arr + 1) & 0xffff) << 16)
    ops_i[:, OPI_W0] = w0.astype(np.uint32).view(np.int32)
    ops_i[:, OPI_W1] = w1.astype(np.uint32).view(np.int32)
    mask = 0
    for row in ops:
        if row[OPI_DOF] >= 0:
            mask |= 1 << row[OPI_DOF]
    n = spec.n_dofs
    arm = (n_ops >= n and not any(prismatic)
           and all(row[OPI_DOF] == (k if k < n else -1) and row[OPI_SRC] == (SRC_ROOT if k == 0 else SRC_PREV)
                   for k, row in enumerate(ops)))
    # DRM_WALK_SERIAL_CHAIN: one root-to-target chain, whatever its joints (the straight-line chain kernels take it)
    serial = (n_ops >= 1 and max_used == 0
              and all(row[OPI_SRC] == (SRC_ROOT if k == 0 else SRC_PREV) and row[OPI_SAVE] < 0 for k, row in enumerate(ops))
              and all((row[OPI_OUT] >= 0) == (k == n_ops - 1) for k, row in enumerate(ops)))
    # bits 8..15 of shape: 1 + the largest op index that is a branch point (what per-ancestor slot records are sized by)
    branch_depth = min(255, max([k + 1 for k, row in enumerate(ops) if row[OPI_SAVE] >= 0], default=0))
    # DRM_WALK_TARGETS_ORDERED: the ops with an output slot carry slots 0, 1, 2, ... in walk order (multi-target FK then writes
    # its outputs a group of slots at a time)
    ordered = [row[OPI_OUT] for row in ops if row[OPI_OUT] >= 0] == list(range(len(tlist)))
    prefix_end, seg_begin, seg_dof = _segments(ops, parent_op, n_ops, n) if whole_tree else (0, [0, n_ops], [(0, n)])
    fk_fan = 0
    if not whole_tree and ordered and len(tlist) > 8 and unique:
        part = fk_fan_partition(ops, parent_op, n_ops)
        if part is not None:
            prefix_end, seg_begin = part
            seg_dof = [(0, 0)] * (len(seg_begin) - 1)
            fk_fan = SHAPE_FK_FAN
    # serial chains of up to 16 ops: the DoF columns and the prismatic bits also travel as launch arguments (what W0 says:
    # bits 0..7 DoF + 1, bit 25 padding, bit 26 prismatic)
    chain_dof1, chain_pris, chain_bit = [0] * 16, 0, 0
    if serial and n_ops <= 16:
        chain_bit = SHAPE_CHAIN_DOFS
        for k in range(n_ops):
            w = int(ops_i[k, OPI_W0]) & 0xffffffff
            d1 = 0 if (w >> 25) & 1 else w & 0xff
            chain_dof1[k] = d1
            if d1 and (w >> 26) & 1:
                chain_pris |= 1 << k
    is_leaf = [not (ops_i[k, OPI_FLAGS] & FLAG_CHILD_IS_NEXT) for k in range(n_ops)]
    seg_leaf_begin = [int(sum(is_leaf[:b])) for b in seg_begin]
    return WalkProgram(np.asarray(links, np.int32), ops_i, np.ascontiguousarray(ops_i.T), gather, gsign, n_ops,
                       max_used, cap, tlist, mask, unique,
                       (SHAPE_ARM_CHAIN if arm else 0) | (SHAPE_SERIAL_CHAIN if serial else 0) | (branch_depth << 8)
                       | (SHAPE_TARGETS_ORDERED if ordered and tlist else 0) | (0 if any(prismatic) else SHAPE_NO_PRISMATIC) | fk_fan | chain_bit
                       | (min(n_leaves, 255) << 16) | (arm_hand_shape(parent_op, n_ops, prismatic) if whole_tree else 0)
                       | (fingers_shape(ops, parent_op, n_ops, n, prismatic) if whole_tree else 0),
                       seg_begin, seg_dof, op_of_link, prefix_end, seg_leaf_begin, tuple(chain_dof1), chain_pris)


def _segments(ops, parent_op, n_ops: int, n_dofs: int):
    """Split a whole-tree walk into independent dynamics problems, one wavefront each (drm_walk.seg_begin):

    * the STATIC PREFIX ops [0, p): fixed joints whose parents are the root or other prefix ops (a mounting plate, the
      base link of a TriFinger) — they never move, so every wavefront replays them (forward sweeps only);
    * the sub-trees that hang off the root or off a prefix op are independent of each other; consecutive ones are packed
      into at most MAX_SEGMENTS runs no longer than the largest sub-tree (a block is as slow as its longest run).
    Returns (prefix_end, seg_begin, seg_dof); one segment covering everything when the DoF columns of a run are not
    contiguous or there is nothing to split."""
    whole = (0, [0, n_ops], [(0, n_dofs)])
    p = 0
    while p < n_ops and ops[p][OPI_DOF] < 0 and parent_op[p] < p and (ops[p][OPI_SRC] == SRC_ROOT or parent_op[p] >= 0):
        p += 1   # (parent_op[p] < p always holds in a depth-first walk: the prefix is closed under "parent of")
    if p == n_ops:
        return whole
    starts = [k for k in range(p, n_ops) if parent_op[k] < p]
    if len(starts) < 2 or starts[0] != p:
        return whole
    bounds = starts + [n_ops]
    sizes = [bounds[j + 1] - bounds[j] for j in range(len(starts))]
    limit = max(sizes)
    while True:
        cuts, acc = [p], 0
        for j, sz in enumerate(sizes):
            # a run may only start at a sub-tree whose first op does not continue from the previous op
            can_cut = ops[bounds[j]][OPI_SRC] != SRC_PREV
            if acc and acc + sz > limit and can_cut:
                cuts.append(bounds[j])
                acc = 0
            acc += sz
        cuts.append(n_ops)
        if len(cuts) - 1 <= MAX_SEGMENTS:
            break
        limit += 1
    if len(cuts) - 1 < 2:
        return whole
    seg_dof = []
    for a, b in zip(cuts, cuts[1:]):
        dofs = sorted(ops[k][OPI_DOF] for k in range(a, b) if ops[k][OPI_DOF] >= 0)
        if not dofs:
            seg_dof.append((0, 0))
        elif dofs[-1] - dofs[0] + 1 != len(dofs):
            return whole
        else:
            seg_dof.append((dofs[0], len(dofs)))
    return p, cuts, seg_dof


def virtual_row_constants(spec: RobotSpec):
    """(links, R_a [S,3,3] float32) of the skew-axis links, in virtual-row order (robot_model._link_table builds
    row A = (F R_a, t, massless, damping) and row B = (R_a^T, 0, mass, mcom, I_o, 0) from them)."""
    links = spec.skew_links
    return links, (spec.axis_rot[links] if links else np.zeros((0, 3, 3), np.float32))


def identity_table_row() -> np.ndarray:
    """Row L of the link table: the constants of an identity (padding) op."""
    row = np.zeros(OPF_STRIDE, np.float32)
    row[OPF_F + 0] = row[OPF_F + 4] = row[OPF_F + 8] = 1.0
    return row
