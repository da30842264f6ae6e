"""Per-robot straight-line dynamics kernels for ARBITRARY trees (round 4).

The reference walks every robot with the same Python loop (robot_model.py:173-193, 262-301).  The ahead-of-time library has
straight-line, register-resident kernels for a few shape families (7-DoF arms, serial chains, an arm that carries a hand, the
fingers of a hand) and loop-structured kernels for every other tree (a mobile manipulator: Fetch).  This module closes that gap
at run time: for a model's whole-tree dynamics walk it writes a ~30-line translation unit — the tree as constexpr functions
(parent op, DoF column, joint kind of every op) and the kernels of csrc/drm_static.hpp instantiated on it — compiles it with
hipcc (~2 s, cached by content hash under DRM_SPECIAL_CACHE or ~/.cache/drm_hip) and hands the kernel handles to the C ABI
through `drm_walk.special[]` (include/drm_hip.h).  `drm_rnea` then runs the robot's own straight-line kernel on full tiles.

    model = DifferentiableRobotModel("fetch.urdf", device="cuda")
    model.specialize()            # or DRM_SPECIALIZE=1 in the environment: every model without a compiled shape does it on first use

Needs hipcc on the machine (ROCm); without it `specialize()` raises and the loop kernels keep serving the robot.
"""
import ctypes
import hashlib
import os
import shutil
import subprocess
from typing import Dict, Optional


from .flatten import OPI_DOF, WalkProgram

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SPECIAL_RNEA, SPECIAL_CRBA, SPECIAL_FD, SPECIAL_RNEA_BACKWARD = 0, 1, 2, 3
KERNELS = {SPECIAL_RNEA: "drm_rnea_static", SPECIAL_CRBA: "drm_crba_static", SPECIAL_FD: "drm_fd_static",
           SPECIAL_RNEA_BACKWARD: "drm_rnea_backward_static"}
# (ABI 10) serial 7-DoF arms with the robot's constants folded into the instruction stream (csrc/drm_arm_stream.hpp)
SPECIAL_RNEA_ARM, SPECIAL_FK_RNEA_ARM, SPECIAL_CRBA_ARM, SPECIAL_FD_ARM, SPECIAL_RNEA_BACKWARD_ARM = 4, 5, 6, 7, 8
ARM_KERNELS = {SPECIAL_RNEA_ARM: "drm_rnea_arm_static", SPECIAL_FK_RNEA_ARM: "drm_fk_rnea_arm_static",
               SPECIAL_CRBA_ARM: "drm_crba_arm_static", SPECIAL_FD_ARM: "drm_fd_arm_static",
               SPECIAL_RNEA_BACKWARD_ARM: "drm_rnea_backward_arm_static"}
ARM_DYNAMICS = (SPECIAL_CRBA_ARM, SPECIAL_FD_ARM, SPECIAL_RNEA_BACKWARD_ARM)
SPECIAL_FK_FAN_LINKS = 9
FAN_KERNEL = "drm_fk_fan_links_static"
# (ABI 11) reverse-mode inverse dynamics of a serial arm WITH learnable link parameters: the constant blocks of the table folded in
SPECIAL_RNEA_BACKWARD_ARM_PARAM = 11
ARM_PARAM_KERNEL = "drm_rnea_backward_arm_param_static"
SPECIAL_FK_RNEA_ARM_PUT = 12          # (ABI 11) the fused FK + RNEA kernel with a one-sided gather of its outputs (drm_fk_rnea_put)
ARM_PUT_KERNEL = "drm_fk_rnea_arm_put_static"
SPECIAL_FD_ARM2 = 13                  # (ABI 11) forward dynamics of the arm, two samples per lane (launches of >= 1 024 tile pairs)
ARM_FD2_KERNEL = "drm_fd_arm2_static"
SPECIAL_RNEA_BACKWARD_ARM2 = 14       # (ABI 11) input gradients of inverse dynamics, two samples per lane
ARM_BWD2_KERNEL = "drm_rnea_backward_arm2_static"
FT_FLOATS, DAMP_INDEX = 12, 25        # include/drm_hip.h DRM_OPF_FT_FLOATS, DRM_OPF_DAMP: [0, 12) = F / t, [12, 26) = mass, mcom, I_o, damping
ARM_KINDS = tuple(ARM_KERNELS)
# (The FK + Jacobian metric kernel was built this way too and measured: 3.75 us either way at 65 536 rows — its pair-packed chain
# folds only a tenth of its instructions and the launch is bound by its loads and its store drain, profiles/r05_metric_static.txt.)
# what folds the constants: x * 0 -> 0 and x + 0 -> x need "no NaN / Inf / signed zero" (nothing else of -ffast-math: no reassociation)
ARM_FLAGS = ["-fno-signed-zeros", "-ffinite-math-only"]
MAX_STATIC_OPS = 24        # beyond this the straight-line walk no longer fits the register file (loop kernels keep serving)


class SpecializeError(RuntimeError):
    pass


class CacheMiss(SpecializeError):
    """`build(..., cached_only=True)`: no code object for this source in the run-time cache or next to the library — what a model's
    default (round 6: own kernels whenever they are already built, never a compile on the call path) treats as "keep the library's
    kernels"."""


def walk_tree(prog: WalkProgram, spec=None) -> Dict[str, list]:
    """(parent op, DoF column, prismatic) of every op of a whole-tree walk, parents before children — decoded from the packed
    control words W0 / W1 (include/drm_hip.h DRM_OPI_W0 / W1), i.e. exactly what the loop kernels read for the same walk
    (folded fixed links and the two ops of a skew-axis joint included)."""
    from .flatten import OPI_W0, OPI_W1
    parent, dof, pris = [], [], []
    for k in range(prog.n_ops):
        w0 = int(prog.ops_i[k, OPI_W0]) & 0xffffffff
        w1 = int(prog.ops_i[k, OPI_W1]) & 0xffffffff
        d = (w0 & 0xff) - 1
        par = ((w1 >> 16) & 0xffff) - 1
        if d != int(prog.ops_i[k, OPI_DOF]) or par >= k:
            raise SpecializeError("control words of op %d do not describe a parent-before-child walk" % k)
        parent.append(par)
        dof.append(d)
        pris.append(bool((w0 >> 26) & 1) and d >= 0)
    return {"parent": parent, "dof": dof, "prismatic": pris}


def tree_tables(tree: Dict[str, list], n_dofs: int):
    """Compile-time tables of the inertia-matrix walk: below[c][k] (op c in the sub-tree of op k), the slot of every related
    (ancestor-or-self k, descendant c) pair of MOVING ops in the triangle, and the slot of every matrix element."""
    parent, dof = tree["parent"], tree["dof"]
    n = len(parent)
    below = [[False] * n for _ in range(n)]
    for c in range(n):
        a = parent[c]
        while a >= 0:
            below[c][a] = True
            a = parent[a]
    slot, count = [[-1] * n for _ in range(n)], 0
    for k in range(n):
        if dof[k] < 0:
            continue
        for c in range(n):
            if dof[c] >= 0 and (c == k or below[c][k]):
                slot[k][c] = count
                count += 1
    op_of_dof = {d: k for k, d in enumerate(dof) if d >= 0}
    if sorted(op_of_dof) != list(range(n_dofs)):
        raise SpecializeError("the walk does not drive every DoF column exactly once")
    slot_of = []
    for di in range(n_dofs):
        for dj in range(n_dofs):
            a, b = op_of_dof[di], op_of_dof[dj]
            lo, hi = min(a, b), max(a, b)
            slot_of.append(slot[lo][hi] if (lo == hi or below[hi][lo]) else count)    # (count = the zero slot)
    return below, slot, count, slot_of


def leaf_ordinals(tree: Dict[str, list]) -> list:
    """Ordinal of every op that the next op of the walk does not hang off (what the backward walk calls a leaf: it starts its way
    down from its own record), -1 for the others."""
    parent = tree["parent"]
    n, out, count = len(parent), [], 0
    for k in range(n):
        if k + 1 < n and parent[k + 1] == k:
            out.append(-1)
        else:
            out.append(count)
            count += 1
    return out


def robot_struct(tree: Dict[str, list], n_dofs: int) -> str:
    """`struct drm::Robot` — the tree as constexpr functions — plus the element -> slot table of the inertia matrix: the text both
    the device translation unit (`source`) and the host harness of the tests (tests/host_emu/static_emu.hpp) are built from."""
    n = len(tree["parent"])
    arr = lambda xs: ", ".join(str(int(x)) for x in xs)
    below, slot, count, slot_of = tree_tables(tree, n_dofs)
    arr2 = lambda rows: ", ".join("{%s}" % arr(r) for r in rows)
    leaves = leaf_ordinals(tree)
    return """namespace drm {
struct Robot {
    static constexpr int N = %d, NDOF = %d, SLOTS = %d, LEAVES = %d;
    static constexpr int parent(int k) { constexpr int t[N] = {%s}; return t[k]; }
    static constexpr int dof(int k) { constexpr int t[N] = {%s}; return t[k]; }
    static constexpr bool prismatic(int k) { constexpr bool t[N] = {%s}; return t[k]; }
    static constexpr bool below(int c, int k) { constexpr bool t[N][N] = {%s}; return t[c][k]; }
    static constexpr int slot(int k, int c) { constexpr int t[N][N] = {%s}; return t[k][c]; }
    static constexpr int leaf(int k) { constexpr int t[N] = {%s}; return t[k]; }
};
}
#define DRM_SLOT_OF_INIT {%s}
""" % (n, n_dofs, count, max(leaves) + 1, arr(tree["parent"]), arr(tree["dof"]), arr(tree["prismatic"]), arr2(below), arr2(slot),
       arr(leaves), arr(slot_of))


def backward_lds_bytes(tree: Dict[str, list], capacity: int) -> int:
    """LDS of drm_rnea_backward_static per wavefront (csrc/drm_static.hpp rnea_backward_static_body): the table, the running sums
    of the constant gradients and 18 floats per leaf and sample."""
    n = len(tree["parent"])
    return 4 * (n * 32 + capacity * 32 + (max(leaf_ordinals(tree)) + 1) * 18 * 64 + 4)


def source(tree: Dict[str, list], n_dofs: int, capacity: int = 0, table=None) -> str:
    """`capacity` (rows of the walk's table = pitch of grad_ops_f): > 0 adds the reverse-mode kernel, when its LDS fits.
    `table` (round 5, constant models): the walk's [>= N, 32] float32 table, written into the translation unit as a constexpr array
    the walks read instead of LDS (csrc/drm_static.hpp DRM_STATIC_ROW) — the robot's constants fold into the instruction stream."""
    backward = ""
    if capacity > 0 and backward_lds_bytes(tree, capacity) <= 64 * 1024:
        backward = """extern "C" __global__ void __launch_bounds__(64) drm_rnea_backward_static(
    const float *ops_f, const float *q, const float *qd, const float *qdd, const float *gtau, int n_tiles, int flags, uint64_t param_mask,
    float *gq, float *gqd, float *gqdd, float *partials) {
    drm::rnea_backward_static_body<drm::Robot, %d>(ops_f, q, qd, qdd, gtau, n_tiles, flags, param_mask, gq, gqd, gqdd, partials);
}
""" % capacity
    const_table = ""
    if table is not None:
        import numpy as np
        n = len(tree["parent"])
        t = np.asarray(table, np.float32)
        if t.ndim != 2 or t.shape[0] < n or t.shape[1] != 32:
            raise SpecializeError("the walk table must be [>= %d, 32] floats" % n)
        rows = ",\n".join("    " + ", ".join(_literal(v) for v in r) for r in t[:n])
        const_table = """#include "drm_common.hpp"
namespace drm {
static __device__ constexpr float ROBOT_OPS[%d * DRM_OPF_STRIDE] = {
%s};
struct RobotTable {
    static __device__ const float *row(int k) { return ROBOT_OPS + k * DRM_OPF_STRIDE; }
};
}
#define DRM_STATIC_CONST_TABLE 1
""" % (n, rows)
    text = """// generated by differentiable-robot-model_amd/specialize.py — one robot's whole-tree dynamics walk as compile-time constants
%s#include "drm_static.hpp"
%s__device__ const int drm_slot_of[] = DRM_SLOT_OF_INIT;
extern "C" __global__ void __launch_bounds__(64) drm_rnea_static(const float *ops_f, const float *q, const float *qd, const float *qdd,
                                                                 int n_tiles, int flags, float *tau, uint32_t magic_n, uint32_t align) {
    drm::rnea_static_body<drm::Robot>(ops_f, q, qd, qdd, n_tiles, flags, tau, magic_n, align);
}
extern "C" __global__ void __launch_bounds__(64) drm_crba_static(const float *ops_f, const float *q, int n_tiles, float *H) {
    drm::crba_static_body<drm::Robot>(ops_f, q, n_tiles, H, drm_slot_of);
}
extern "C" __global__ void __launch_bounds__(64) drm_fd_static(const float *ops_f, const float *q, const float *qd, const float *f,
                                                               int n_tiles, int flags, float *qdd, uint32_t magic_n, uint32_t align) {
    drm::aba_static_body<drm::Robot>(ops_f, q, qd, f, n_tiles, flags, qdd, magic_n, align);
}
%s""" % (const_table, robot_struct(tree, n_dofs), backward)
    waves = os.environ.get("DRM_STATIC_WAVES")          # (experiments: force N wavefronts per SIMD on the forward kernels)
    if waves:
        for name in ("drm_rnea_static", "drm_fd_static"):
            text = text.replace("__launch_bounds__(64) %s(" % name,
                                "__launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(%s, %s))) %s(" % (waves, waves, name))
    return text


def host_source(tree: Dict[str, list], n_dofs: int, harness: str) -> str:
    """The same robot for the HOST harness of the tests (`harness` = path of tests/host_emu/static_emu.hpp)."""
    return """// generated by differentiable-robot-model_amd/specialize.py for the host harness of the tests
#include "%s"
%sstatic const int slot_of_host[] = DRM_SLOT_OF_INIT;
STATIC_EMU_EXPORTS(slot_of_host)
""" % (harness, robot_struct(tree, n_dofs))


def cache_dir() -> str:
    d = os.environ.get("DRM_SPECIAL_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "drm_hip")
    os.makedirs(d, exist_ok=True)
    return d


# code objects that ship WITH the package: `__graft_entry__.build()` pre-builds the shipped robots' own kernels here (next to
# libdrm_hip.so; *.hsaco is git-ignored like every built artefact), so a deployment finds them without running hipcc
SHIPPED_CACHE = os.path.join(CSRC, "special_cache")


def shipped_cache() -> str:
    """SHIPPED_CACHE, or DRM_SHIPPED_CACHE from the environment (tests point it at an empty directory)."""
    return os.environ.get("DRM_SHIPPED_CACHE") or SHIPPED_CACHE


# (the fingertip sets whose fan-out FK kernel is pre-built: BASELINE configuration 4's)
SHIPPED_FANS = {"allegro_left": ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]}


SHIPPED_TREES = ("fetch", "panda", "jaco")      # robots whose whole-tree dynamics kernels ship (entry points per *.tuned.json)


SHIPPED_ROBOTS = ("panda_no_gripper", "iiwa7", "allegro_left", "fetch_arm_no_gripper")
SHIPPED_LEARNABLE = ("panda_no_gripper", "iiwa7")      # arms whose learnable-parameter kernels ship (the examples' sets)


def prebuild_shipped(robots=SHIPPED_ROBOTS, trees=()) -> list:
    """Build, into SHIPPED_CACHE, the kernels `specialize()` would build at run time for the robots of the BASELINE configurations:
    the arms' inverse-dynamics / fused FK + RNEA (end-effector link) / inertia-matrix / forward-dynamics / input-gradient kernels and
    the Allegro's four-fingertip fan-out kernel; `trees`: the whole-tree straight-line dynamics kernels (`attach`) of robots that are
    not plain arms — a mobile manipulator (Fetch: no compiled shape in the library, every entry point 2-3.6x faster), arms with a
    gripper (Panda, Jaco: the entry points their shipped tuning record keeps).  Needs hipcc and no GPU (models are built on the CPU:
    the tables are device-independent).  Returns the file names."""
    import contextlib
    import io

    from .robot_model import DifferentiableRobotModel, robot_description_folder
    before = os.environ.get("DRM_SPECIAL_CACHE")
    os.environ["DRM_SPECIAL_CACHE"] = shipped_cache()
    built = []
    try:
        for robot in robots:
            with contextlib.redirect_stdout(io.StringIO()):
                m = DifferentiableRobotModel(os.path.join(robot_description_folder, robot + ".urdf"), device="cpu")
            dw = m._dynamics_walk()
            table = m._ops_f(dw).detach().numpy()
            if arm_qualifies(dw.program, m._n_dofs):
                links = dw.program.n_ops
                built.append(build(arm_source(table, links, False), ARM_FLAGS))
                built.append(build(arm_dynamics_source(table, links), ARM_FLAGS))
                ee = len(m._bodies) - 1                                   # (the arm's last link: the FK target of configuration 3)
                tree, chain = dw, m._get_walk(("chain", ee) + (("folded", dw.fold_key) if dw.folded else ()), targets=[ee],
                                              folded=dw.folded, fold_key=dw.fold_key)
                if arm_qualifies(chain.program, m._n_dofs) and chain.program.n_ops == 8:
                    both = table.copy()
                    both[links:] = m._ops_f(chain).detach().numpy()[links:]
                    built.append(build(arm_source(both, links, True), ARM_FLAGS))
            if arm_qualifies(dw.program, m._n_dofs) and robot in SHIPPED_LEARNABLE:
                # ... and the reverse-mode kernels of the arm WITH learnable dynamic parameters (round 6): mass, com and inertia_mat of
                # all moving links (examples/learn_dynamics_iiwa.py) and of every single one of them (identification of one link / a payload)
                moving = [m._bodies[i].name for i in m._controlled_joints]
                for names in [moving] + [[name] for name in moving]:
                    built.append(build(learnable_arm_source(robot, names, ("mass", "com", "inertia_mat")), ARM_FLAGS))
            tips = sorted(m._name_to_idx_map[name] for name in SHIPPED_FANS.get(robot, []))     # (_fk_links launches in link order)
            if 2 <= len(tips) <= 4:
                merged = m._get_walk(("fk", tuple(tips)), targets=tips)
                fan = m._fanout_chains(tips, merged)
                chains = [fan_chain(w.program, m._n_dofs) for w in fan] if fan else []
                if chains and all(chains):
                    built.append(build(fan_source(chains, [m._ops_f(w).detach().numpy() for w in fan], m._n_dofs), ARM_FLAGS))
        for robot in trees:
            with contextlib.redirect_stdout(io.StringIO()):
                m = DifferentiableRobotModel(os.path.join(robot_description_folder, robot + ".urdf"), device="cpu")
            dw = m._dynamics_walk()
            if arm_qualifies(dw.program, m._n_dofs) or dw.program.n_ops > MAX_STATIC_OPS:
                continue
            src = source(walk_tree(dw.program, m._spec), m._n_dofs, dw.program.capacity if dw.program.backward_ok else 0, m._const_table(dw))
            built.append(build(src, ARM_FLAGS))
    finally:
        if before is None:
            os.environ.pop("DRM_SPECIAL_CACHE", None)
        else:
            os.environ["DRM_SPECIAL_CACHE"] = before
    return [os.path.basename(b) for b in built]


def learnable_arm_source(robot: str, link_names, parameter_names) -> str:
    """The reverse-mode translation unit `model.specialize()` / the default look-up would ask for when the named parameters of the
    named links of a shipped arm are learnable: a host model with those parameters learnable, its dynamics walk, its block masks."""
    import contextlib
    import io

    import torch

    from .rigid_body_params import UnconstrainedTensor
    from .robot_model import DifferentiableRobotModel, robot_description_folder
    with contextlib.redirect_stdout(io.StringIO()):
        m = DifferentiableRobotModel(os.path.join(robot_description_folder, robot + ".urdf"), device="cpu")
    shapes = {"mass": (1, 1), "com": (1, 3), "inertia_mat": (3, 3), "trans": (1, 3), "rot_angles": (1, 3), "joint_damping": (1, 1)}
    for link in link_names:
        for pname in parameter_names:
            m.make_link_param_learnable(link, pname, UnconstrainedTensor(dim1=shapes[pname][0], dim2=shapes[pname][1]))
    dw = m._dynamics_walk()
    kin, dyn = m._learnable_block_masks(dw)
    with torch.no_grad():
        table = m._ops_f(dw).detach().numpy()
    return arm_param_backward_source(table, dw.program.n_ops, kin, dyn)


def hipcc() -> Optional[str]:
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


_CC_ID: Dict[str, str] = {}


def compiler_id(cc: str) -> str:
    """What identifies the compiler in the cache key: the text of `hipcc --version` (asked once per process)."""
    if cc not in _CC_ID:
        done = subprocess.run([cc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        _CC_ID[cc] = hashlib.sha256(done.stdout).hexdigest()[:8]
    return _CC_ID[cc]


def target_arch() -> str:
    """The ISA the code objects are built for: DRM_SPECIAL_ARCH, else the current HIP device's (gfx950 on an MI355X; also the
    answer on a machine without a device, e.g. when a cache is prepared for export)."""
    arch = os.environ.get("DRM_SPECIAL_ARCH")
    if arch:
        return arch
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName.split(":")[0]
    except Exception:       # noqa: BLE001  (no device, no torch: the product's one target)
        pass
    return "gfx950"


_HEADERS = ("drm_static.hpp", "drm_tree.hpp", "drm_sample.hpp", "drm_common.hpp", "drm_arm_stream.hpp", "drm_arm_static.hpp")
_HEADER_BYTES: Dict[tuple, bytes] = {}


def _headers() -> bytes:
    """The headers a generated translation unit includes, for the source key (read once per process and set of mtimes)."""
    stamp = tuple(os.path.getmtime(os.path.join(CSRC, name)) for name in _HEADERS)
    if stamp not in _HEADER_BYTES:
        _HEADER_BYTES.clear()
        parts = []
        for name in _HEADERS:
            with open(os.path.join(CSRC, name), "rb") as f:
                parts.append(f.read())
        _HEADER_BYTES[stamp] = b"".join(parts)
    return _HEADER_BYTES[stamp]


def build(src: str, flags=(), cached_only: bool = False) -> str:
    """Path of the code object of `src`: drm_special_<source key>_<compiler key>.hsaco in the cache directory.  The source key
    covers the generated text, the headers it includes, the flags and the target ISA; the compiler key `hipcc --version`.  A
    machine WITHOUT hipcc takes any code object with the right source key (a cache exported by `export_cache` from a build
    machine).  Several processes may build the same robot at once (the ranks of one node): each compiles to its own temporary
    name and publishes with an atomic rename, so nobody ever loads a half-written file.
    `cached_only` (round 6, what a model does by itself on first use): never run the compiler — a code object with the right source
    key from the run-time cache or from the shipped one, else CacheMiss."""
    import glob
    import tempfile
    extra = list(flags) + os.environ.get("DRM_SPECIAL_FLAGS", "").split()      # (experiments: e.g. -DDRM_STATIC_PREF=0)
    arch = target_arch()
    h = hashlib.sha256((src + " ".join(extra) + arch).encode())
    h.update(_headers())
    key = h.hexdigest()[:20]
    cc = None if cached_only else hipcc()
    if cc is None:
        shipped = sorted(glob.glob(os.path.join(cache_dir(), "drm_special_%s_*.hsaco" % key)) +
                         glob.glob(os.path.join(shipped_cache(), "drm_special_%s_*.hsaco" % key)))
        if shipped:
            return shipped[0]
        if cached_only:
            raise CacheMiss("no built code object drm_special_%s_*.hsaco in %s or %s" % (key, cache_dir(), shipped_cache()))
        raise SpecializeError("hipcc not found and %s holds no code object for this robot (drm_special_%s_*.hsaco): build it "
                              "on a machine with the ROCm compiler and copy the cache (specialize.export_cache)" % (cache_dir(), key))
    name = "drm_special_%s_%s.hsaco" % (key, compiler_id(cc))
    out = os.path.join(cache_dir(), name)
    if os.path.exists(out):
        return out
    if os.path.exists(os.path.join(shipped_cache(), name)):      # (built with the package by __graft_entry__.build(), same compiler)
        return os.path.join(shipped_cache(), name)
    fd, cpp = tempfile.mkstemp(prefix="drm_special_%s_" % key, suffix=".hip", dir=cache_dir())
    tmp = cpp[:-4] + ".hsaco.tmp"
    try:
        with os.fdopen(fd, "w") as f:
            f.write(src)
        cmd = [cc, "--genco", "--offload-arch=" + arch, "-O3", "-std=c++17", "-ffp-contract=fast", "-fno-slp-vectorize", "-w",
               "-I", CSRC, "-o", tmp, cpp] + extra
        done = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if done.returncode != 0 or not os.path.exists(tmp):
            raise SpecializeError("hipcc failed on the robot's kernels:\n%s" % done.stderr.decode()[-2000:])
        os.replace(tmp, out)          # (atomic; a concurrent builder's identical file may already be there: same content)
    finally:
        for leftover in (cpp, tmp):
            try:
                os.remove(leftover)
            except OSError:
                pass
    return out


def export_cache(dest: str) -> list:
    """Copy every code object of this machine's cache into `dest` (created if missing) — for deployment machines without the ROCm
    compiler: ship the directory and point DRM_SPECIAL_CACHE at it; `build` there finds the robots' kernels by their source key."""
    import glob
    os.makedirs(dest, exist_ok=True)
    copied = []
    for path in sorted(glob.glob(os.path.join(cache_dir(), "drm_special_*.hsaco"))):
        shutil.copy2(path, os.path.join(dest, os.path.basename(path)))
        copied.append(os.path.basename(path))
    return copied


def _literal(x) -> str:
    """A float32 as a C hex-float literal (exact round trip; zeros as the plain 0.0f the compiler folds)."""
    import numpy as np
    x = np.float32(x)
    if x == 0:
        return "0.0f"
    return "%sf" % float(x).hex()


def arm_source(table, links: int, fused: bool) -> str:
    """The translation unit of one serial 7-DoF arm: `table` = the [8, 32] float32 rows its streaming walk reads (rows < `links`
    from the dynamics walk's table, the fixed tail the pose chain still walks from the chain walk's — what
    csrc/drm_arm_dynamics.hip arm2_stream_kernel copies into LDS), written out as a constexpr array.  `fused`: the FK + RNEA
    kernel of this (dynamics walk, target chain) pair; else inverse dynamics alone."""
    import numpy as np
    table = np.asarray(table, np.float32)
    if table.shape != (8, 32) or links not in (7, 8):
        raise SpecializeError("arm kernels are built for walks of capacity 8 with 7 or 8 dynamics ops")
    rows = ",\n".join("    " + ", ".join(_literal(v) for v in r) for r in table)
    name = ARM_KERNELS[SPECIAL_FK_RNEA_ARM if fused else SPECIAL_RNEA_ARM]
    tail = ", float *pos, float *quat" if fused else ""
    text = """// generated by differentiable-robot-model_amd/specialize.py — one serial arm's walk table as compile-time constants
#include "drm_arm_stream.hpp"
namespace drm {
static __device__ constexpr float ROBOT_OPS[8 * DRM_OPF_STRIDE] = {
%s};
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
%s(const float *q, const float *qd, const float *qdd, int n_tiles, int flags, float *tau%s) {
    constexpr int NJ = 7, T_FLOATS = drm::round4(drm::STREAM_TILE * NJ), ROWS_F = drm::STREAM_TILE * NJ;
    __shared__ __attribute__((aligned(16))) float smem[T_FLOATS + 3 * ROWS_F];
    drm::arm2_stream_body<8, NJ, %d, %s, false>([] {}, [] { return [](int k) -> const float * { return drm::ROBOT_OPS + k * DRM_OPF_STRIDE; }; },
                                              smem, q, qd, qdd, n_tiles, flags, tau, %s);
}
""" % (rows, name, tail, links, "true" if fused else "false", "pos, quat" if fused else "nullptr, nullptr")
    if fused:      # (ABI 11) the same launch with a one-sided gather of its outputs (drm_fk_rnea_put): every tile also to the peers' arrays
        text += """extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
%s(const float *q, const float *qd, const float *qdd, int n_tiles, int flags, float *tau, float *pos, float *quat, drm_put put) {
    constexpr int NJ = 7, T_FLOATS = drm::round4(drm::STREAM_TILE * NJ), ROWS_F = drm::STREAM_TILE * NJ;
    __shared__ __attribute__((aligned(16))) float smem[T_FLOATS + 3 * ROWS_F];
    drm::arm2_stream_body<8, NJ, %d, true, false, true>([] {}, [] { return [](int k) -> const float * { return drm::ROBOT_OPS + k * DRM_OPF_STRIDE; }; },
                                                      smem, q, qd, qdd, n_tiles, flags, tau, pos, quat, &put);
}
""" % (ARM_PUT_KERNEL, links)
    return text


def arm_dynamics_source(table, links: int) -> str:
    """The inertia matrix, forward dynamics and the input gradients of inverse dynamics of one serial 7-DoF arm (the one-sample chain
    walks, csrc/drm_arm_static.hpp) on a constexpr copy of the dynamics walk's [8, 32] table."""
    import numpy as np
    table = np.asarray(table, np.float32)
    if table.shape != (8, 32) or links not in (7, 8):
        raise SpecializeError("arm kernels are built for walks of capacity 8 with 7 or 8 dynamics ops")
    rows = ",\n".join("    " + ", ".join(_literal(v) for v in r) for r in table)
    return """// generated by differentiable-robot-model_amd/specialize.py — one serial arm's walk table as compile-time constants
#include "drm_arm_static.hpp"
namespace drm {
static __device__ constexpr float ROBOT_OPS[8 * DRM_OPF_STRIDE] = {
%s};
struct RobotRow {
    __device__ const float *operator()(int k) const { return ROBOT_OPS + k * DRM_OPF_STRIDE; }
};
}
extern "C" __global__ void __launch_bounds__(64) drm_crba_arm_static(const float *q, int n_tiles, float *H) {
    drm::crba_arm_static_body<7, %d>(drm::RobotRow(), q, n_tiles, H);
}
extern "C" __global__ void __launch_bounds__(64) drm_fd_arm_static(const float *q, const float *qd, const float *f, int n_tiles, int flags, float *qdd) {
    drm::forward_dynamics_arm_static_body<7, %d>(drm::RobotRow(), q, qd, f, n_tiles, flags, qdd);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
drm_fd_arm2_static(const float *q, const float *qd, const float *f, int n_pairs, int flags, float *qdd) {
    drm::forward_dynamics_arm2_static_body<7, %d>(drm::RobotRow(), q, qd, f, n_pairs, flags, qdd);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
drm_rnea_backward_arm2_static(const float *q, const float *qd, const float *qdd, const float *gtau, int n_pairs, int flags, float *gq, float *gqd,
                              float *gqdd) {
    drm::rnea_backward_arm2_static_body<7, %d>(drm::RobotRow(), q, qd, qdd, gtau, n_pairs, flags, gq, gqd, gqdd);
}
extern "C" __global__ void __launch_bounds__(64) drm_rnea_backward_arm_static(const float *q, const float *qd, const float *qdd, const float *gtau,
                                                                              int n_tiles, int flags, float *gq, float *gqd, float *gqdd) {
    drm::rnea_backward_arm_static_body<7, %d>(drm::RobotRow(), q, qd, qdd, gtau, n_tiles, flags, gq, gqd, gqdd);
}
""" % (rows, links, links, links, links, links)


def arm_param_backward_source(table, links: int, mask_kin: int, mask_dyn: int) -> str:
    """The translation unit of one serial 7-DoF arm's reverse-mode inverse dynamics for ONE set of learnable blocks: `table` = the
    [8, 32] walk table of the model (the learnable blocks are written as zeros — their values come from the launch's ops_f, and a
    parameter update must not change the source key), bit k of `mask_kin` / `mask_dyn`: op k's F / t block / its mass, mcom, I_o,
    damping block is learnable (csrc/drm_arm_static.hpp rnea_backward_arm_param_static_body)."""
    import numpy as np
    table = np.array(table, np.float32, copy=True)
    if table.shape != (8, 32) or links not in (7, 8):
        raise SpecializeError("arm kernels are built for walks of capacity 8 with 7 or 8 dynamics ops")
    if not (mask_kin | mask_dyn) or (mask_kin | mask_dyn) >> links:
        raise SpecializeError("the learnable blocks must belong to the walk's %d ops" % links)
    for k in range(8):
        if (mask_kin >> k) & 1:
            table[k, :FT_FLOATS] = 0.0
        if (mask_dyn >> k) & 1:
            table[k, FT_FLOATS:DAMP_INDEX + 1] = 0.0
    rows = ",\n".join("    " + ", ".join(_literal(v) for v in r) for r in table)
    return """// generated by differentiable-robot-model_amd/specialize.py — a serial arm with learnable link parameters: the constant blocks of its walk
// table as compile-time constants, the learnable ones (kinematic 0x%x, dynamic 0x%x) read from the launch's table
#include "drm_arm_static.hpp"
namespace drm {
static __device__ constexpr float ROBOT_OPS[8 * DRM_OPF_STRIDE] = {
%s};
struct RobotRowMixed {
    const float *live;      // this launch's table (LDS)
    static constexpr uint32_t KIN = 0x%xu, DYN = 0x%xu;
    __device__ const float *operator()(int k) const { return ((DYN >> k) & 1u) ? live + k * DRM_OPF_STRIDE : ROBOT_OPS + k * DRM_OPF_STRIDE; }
    __device__ const float *ft(int k) const { return ((KIN >> k) & 1u) ? live + k * DRM_OPF_STRIDE : ROBOT_OPS + k * DRM_OPF_STRIDE; }
};
}
extern "C" __global__ void __launch_bounds__(256) %s(
    const float *ops_f, const float *q, const float *qd, const float *qdd, const float *gtau, int n_tiles, int flags, float *gq, float *gqd,
    float *gqdd, float *partials) {
    drm::rnea_backward_arm_param_static_body<7, %d, drm::RobotRowMixed::KIN, drm::RobotRowMixed::DYN, drm::RobotRowMixed>(
        ops_f, q, qd, qdd, gtau, n_tiles, flags, gq, gqd, gqdd, partials);
}
""" % (mask_kin, mask_dyn, rows, mask_kin, mask_dyn, ARM_PARAM_KERNEL, links)


def attach_arm_param(tree: WalkProgram, tree_table, n_dofs: int, mask_kin: int, mask_dyn: int, cached_only: bool = False) -> int:
    """Build (cached) and attach the reverse-mode kernel of a serial 7-DoF arm with THIS set of learnable blocks; the C ABI runs it
    when a call's param_mask equals `mask_kin | mask_dyn` (stored in drm_walk.reserved0).  Returns the handle."""
    if not arm_qualifies(tree, n_dofs):
        raise SpecializeError("not a serial 7-DoF arm walk of capacity 8")
    handle = _load(build(arm_param_backward_source(tree_table, tree.n_ops, mask_kin, mask_dyn), ARM_FLAGS, cached_only), ARM_PARAM_KERNEL)
    special = dict(getattr(tree, "_special", None) or {})
    special[SPECIAL_RNEA_BACKWARD_ARM_PARAM] = handle
    tree._special, tree._special_mask, tree._ws_cache = special, mask_kin | mask_dyn, None
    return handle


def fan_chain(prog: WalkProgram, n_dofs: int) -> Optional[dict]:
    """What the fan-out kernel needs to know about one chain walk, or None when the chain does not qualify: a serial chain of at
    most 8 ops without branch points or sliding joints; per op the DoF column it reads (-1: fixed)."""
    from .flatten import SHAPE_SERIAL_CHAIN
    if not (prog.shape & SHAPE_SERIAL_CHAIN) or prog.n_slots or not 1 <= prog.n_ops <= 8 or int(prog.chain_prismatic):
        return None
    dof = [int(v) - 1 for v in list(prog.chain_dof1)[:prog.n_ops]]
    moving = [d for d in dof if d >= 0]
    if not moving or len(set(moving)) != len(moving) or max(moving) >= n_dofs:
        return None
    perm = getattr(prog, "_target_perm", None)
    if perm is None:
        from .flatten import OPI_PERM
        perm = int(prog.ops_i[prog.n_ops - 1, OPI_PERM])
    q4 = (len(moving) == 4 and n_dofs % 4 == 0 and moving[0] % 4 == 0 and moving == list(range(moving[0], moving[0] + 4))
          and dof[:4] == moving)
    return {"used": prog.n_ops, "dof": dof, "moving": len(moving), "first": moving[0], "q4": q4, "perm": int(perm)}


def fan_source(chains, tables, n_dofs: int) -> str:
    """The translation unit of one fan-out FK call: `chains` = fan_chain() of its 2 .. 4 chain walks, `tables` their [capacity, 32]
    constant tables (the first `used` rows are written out)."""
    import numpy as np
    if not 2 <= len(chains) <= 4:
        raise SpecializeError("fan-out FK takes 2 to 4 chains")
    parts = []
    for i, (c, t) in enumerate(zip(chains, tables)):
        t = np.asarray(t, np.float32)[:c["used"]]
        rows = ",\n".join("    " + ", ".join(_literal(v) for v in r) for r in t)
        parts.append("""static __device__ constexpr float OPS%d[%d * DRM_OPF_STRIDE] = {
%s};
struct Chain%d {
    static constexpr int USED = %d, MOVING = %d, NDOFS = %d, FIRST = %d, PERM = %d;
    static constexpr bool Q4 = %s;
    static __device__ constexpr int dof(int k) {
        constexpr int D[%d] = {%s};
        return D[k];
    }
    static __device__ const float *row(int k) { return OPS%d + k * DRM_OPF_STRIDE; }
};
""" % (i, c["used"], rows, i, c["used"], c["moving"], n_dofs, c["first"], c["perm"], "true" if c["q4"] else "false", c["used"],
       ", ".join(str(d) for d in c["dof"]), i))
    cases = "\n".join("    case %d: drm::fk_fan_links_static_wave<drm::Chain%d>(q, pos, quat, B, %d, st + %d * 3 * drm::WAVE); break;" % (i, i, i, i)
                      for i in range(len(chains)))
    return """// generated by differentiable-robot-model_amd/specialize.py — the chains of one fan-out FK call as compile-time constants
#include "drm_arm_static.hpp"
namespace drm {
%s}
extern "C" __global__ void __launch_bounds__(%d) %s(const float *q, float *pos, float *quat, long long B) {
    __shared__ __attribute__((aligned(16))) float st[%d * 3 * drm::WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    switch (wave) {
%s
    default: break;
    }
}
""" % ("".join(parts), 64 * len(chains), FAN_KERNEL, len(chains), cases)


def attach_fan(progs, tables, n_dofs: int, cached_only: bool = False) -> Optional[int]:
    """Build (cached) the constant-folded fan-out FK kernel of these 2 .. 4 chain walks, IN THIS ORDER, and return its handle —
    None when a chain does not qualify (the library's fan-out kernel keeps serving the call).  The kernel belongs to the ordered
    set of chains, not to any one of them (the chain walk of a fingertip is shared by every set of targets that names it): the
    caller keeps the handle with its fan-out plan and hands it to backend.fk_fanout(own=...), which writes it into the structs of
    that call only (round 5 stored it on the shared programs: a subset of a specialised set of tips launched the wrong kernel)."""
    chains = [fan_chain(p, n_dofs) for p in progs]
    if any(c is None for c in chains) or not 2 <= len(chains) <= 4:
        return None
    return _load(build(fan_source(chains, tables, n_dofs), ARM_FLAGS, cached_only), FAN_KERNEL)


_LOADED: Dict[tuple, int] = {}


def _load(path: str, kernel: str) -> int:
    """Kernel handle of a code object (one hipModuleLoad per file and process)."""
    from . import backend
    key = (path, kernel)
    if key not in _LOADED:
        lib = backend.load_library()
        fn = ctypes.c_void_p()
        backend._check(lib.drm_special_load(path.encode(), kernel.encode(), ctypes.byref(fn)))
        _LOADED[key] = fn.value
    return _LOADED[key]


def arm_qualifies(prog: WalkProgram, n_dofs: int) -> bool:
    from .flatten import SHAPE_ARM_CHAIN
    return bool(prog.shape & SHAPE_ARM_CHAIN) and prog.capacity == 8 and n_dofs == 7 and prog.n_ops in (7, 8)


def attach_arm(tree: WalkProgram, tree_table, n_dofs: int, chain: Optional[WalkProgram] = None, chain_table=None,
               cached_only: bool = False) -> Dict[int, int]:
    """Build (hipcc ~2 s each, cached) and attach the constant-folded kernels of a serial 7-DoF arm: inverse dynamics, the inertia
    matrix, forward dynamics and the input gradients of inverse dynamics on the dynamics walk `tree` (its [8, 32] table as a host
    array), and — given the chain walk of an FK target whose last link is the
    arm's last link — the fused FK + RNEA kernel of the pair, the SAME handle stored on both programs (drm_fk_rnea checks that).
    A tree program keeps the fused kernel of the LAST chain it was paired with.  Constant models only: the kernels ignore ops_f.
    `cached_only`: attach whichever of them is already built (the run-time cache, the code objects shipped next to the library) and
    never compile; a piece that is missing stays with the library's kernel (`tree._special_missed` names it)."""
    import numpy as np
    if not arm_qualifies(tree, n_dofs):
        raise SpecializeError("not a serial 7-DoF arm walk of capacity 8")
    links = tree.n_ops
    table = np.array(tree_table, np.float32, copy=True)
    special = dict(getattr(tree, "_special", None) or {})
    missed = []

    def piece(what, fn):
        try:
            fn()
        except CacheMiss:
            missed.append(what)

    def rnea_arm():
        special[SPECIAL_RNEA_ARM] = _load(build(arm_source(table, links, False), ARM_FLAGS, cached_only), ARM_KERNELS[SPECIAL_RNEA_ARM])

    def dynamics():
        path = build(arm_dynamics_source(table, links), ARM_FLAGS, cached_only)
        for kind in ARM_DYNAMICS:
            special[kind] = _load(path, ARM_KERNELS[kind])
        special[SPECIAL_FD_ARM2] = _load(path, ARM_FD2_KERNEL)
        special[SPECIAL_RNEA_BACKWARD_ARM2] = _load(path, ARM_BWD2_KERNEL)

    def fused():
        both = table.copy()
        both[links:] = np.asarray(chain_table, np.float32)[links:]
        path = build(arm_source(both, links, True), ARM_FLAGS, cached_only)
        handle, put = _load(path, ARM_KERNELS[SPECIAL_FK_RNEA_ARM]), _load(path, ARM_PUT_KERNEL)
        special[SPECIAL_FK_RNEA_ARM], special[SPECIAL_FK_RNEA_ARM_PUT] = handle, put
        cs = dict(getattr(chain, "_special", None) or {})
        cs[SPECIAL_FK_RNEA_ARM], cs[SPECIAL_FK_RNEA_ARM_PUT] = handle, put
        chain._special, chain._ws_cache = cs, None

    if SPECIAL_RNEA_ARM not in special:
        piece(ARM_KERNELS[SPECIAL_RNEA_ARM], rnea_arm)
    if SPECIAL_CRBA_ARM not in special:
        piece("the inertia-matrix / forward-dynamics / input-gradient kernels", dynamics)
    if chain is not None:
        if not (arm_qualifies(chain, n_dofs) and chain.n_ops == 8):
            raise SpecializeError("the FK target's chain is not this arm's chain of 8 ops")
        piece(ARM_KERNELS[SPECIAL_FK_RNEA_ARM], fused)
    tree._special, tree._ws_cache = special, None
    tree._special_missed = missed
    return special


def _tuned_name(path: str) -> str:
    """drm_special_<source key>.tuned.json of a code object drm_special_<source key>_<compiler key>.hsaco (the choice belongs to the
    kernels' source, not to the compiler build that produced them)."""
    stem = os.path.basename(path)
    return "_".join(stem.split("_")[:3]) + ".tuned.json"


def tuned_kinds(path: str):
    """The entry points `tune` kept for this code object — on this machine (the run-time cache) or, failing that, as shipped with the
    package (csrc/special_cache/*.tuned.json, measured on an MI355X by tools/tune_shipped.py) — or None when it was never tuned."""
    import json
    for folder in (cache_dir(), shipped_cache()):
        try:
            with open(os.path.join(folder, _tuned_name(path))) as f:
                return set(int(k) for k in json.load(f)["kept"])
        except (OSError, ValueError, KeyError):
            continue
    return None


def attach(prog: WalkProgram, spec, n_dofs: int, table=None, cached_only: bool = False, tuned_only: bool = False,
           ignore_tuned: bool = False) -> Dict[int, int]:
    """Build (or fetch from the cache) and load the straight-line kernels of a whole-tree walk; the handles are stored on the
    program, from where backend._walk_struct copies them into every drm_walk built for it.  `table` (a CONSTANT model's walk table
    as a host array): the kernels carry it as compile-time constants (`source`) and no longer read ops_f — the host must drop them
    when a parameter becomes learnable (`prog._special_const`).
    A tuning record of this code object (`tuned_kinds`: written by `tune`, or shipped) limits the handles to the entry points it kept;
    `tuned_only`: attach nothing without one (robots with a compiled shape in the library: their own kernels win some entry points
    and lose others, only a measurement says which); `ignore_tuned`: every entry point (what `tune` itself starts from)."""
    from . import backend
    if any(k in (getattr(prog, "_special", None) or {}) for k in KERNELS):
        return prog._special
    if prog.n_ops > MAX_STATIC_OPS:
        raise SpecializeError("walk of %d ops: the straight-line form is built for up to %d" % (prog.n_ops, MAX_STATIC_OPS))
    tree = walk_tree(prog, spec)
    src = source(tree, n_dofs, prog.capacity if prog.backward_ok else 0, table)
    path = build(src, ARM_FLAGS if table is not None else (), cached_only)
    kept = None if ignore_tuned else tuned_kinds(path)
    if tuned_only and kept is None:
        raise CacheMiss("%s was never tuned" % os.path.basename(path))
    handles = dict(getattr(prog, "_special", None) or {})
    for kind, kernel in KERNELS.items():
        if kernel not in src:           # (the reverse-mode kernel of a walk the backward entry points do not take / whose leaves exceed LDS)
            continue
        if kept is not None and kind not in kept:
            continue
        handles[kind] = _load(path, kernel)
    prog._special = handles
    prog._special_path = path
    prog._special_const = table is not None
    prog._ws_cache = None          # (the cached drm_walk predates the handles)
    return handles


def tune(prog: WalkProgram, ops_f, ops_i, n_dofs: int, batch: int = 1 << 19, margin: float = 0.97) -> Dict[str, dict]:
    """Keep, per entry point, whichever is FASTER on this device: the robot's own straight-line kernel (attached by `attach`) or
    what the library would run without it (a shape-specialised kernel: an arm that carries a hand, the fingers of a hand — or
    the loop kernels).  Each entry point is timed both ways on `batch` random rows (HIP events around 10 eager launches, best of 3;
    the batch is large enough that a launch outlasts the ~13 us of host time an eager call costs, so the device is what is timed);
    a handle stays when its kernel takes < `margin` of the library's time.  Returns {kernel: {"own_us", "library_us", "kept"}}.
    Measured on an MI355X at 2^20 rows (profiles/r04_probe_special.txt): Panda with gripper keeps all four (inertia matrix 133 -> 86
    us), Jaco the inertia matrix and the reverse mode, a hand (Allegro) and an arm carrying one (iiwa7 + Allegro) none."""
    import torch
    from . import backend
    handles = dict(getattr(prog, "_special", None) or {})
    if not handles:
        return {}
    import copy
    shared, prog = prog, copy.copy(prog)     # timed on a PRIVATE copy: other threads / plans keep seeing the shared program's handles
    dev = ops_f.device
    g = torch.Generator(device="cpu").manual_seed(0)
    q, qd, x = ((torch.rand(batch, n_dofs, generator=g) - 0.5).to(dev) for _ in range(3))
    calls = {
        SPECIAL_RNEA: lambda: backend.rnea(prog, ops_f, ops_i, q, qd, x, True, True, n_dofs),
        SPECIAL_CRBA: lambda: backend.crba(prog, ops_f, ops_i, q, n_dofs),
        SPECIAL_FD: lambda: backend.forward_dynamics(prog, ops_f, ops_i, q, qd, x, True, True, n_dofs),
        SPECIAL_RNEA_BACKWARD: lambda: backend.rnea_backward(prog, ops_f, ops_i, q, qd, x, x, True, True, n_dofs, 0, True),
    }

    def timed(fn):
        best = float("inf")
        fn()
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                fn()
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b) * 100.0)      # us per launch
        return best

    def with_handles(h):
        prog._special = h
        prog._ws_cache = None

    report, kept = {}, {k: h for k, h in handles.items() if k not in calls}      # (kinds tune does not time stay as they are)
    try:
        for kind, handle in handles.items():
            if kind not in calls:
                continue
            with_handles({kind: handle})
            own = timed(calls[kind])
            with_handles({})
            lib = timed(calls[kind])
            keep = own < margin * lib
            report[KERNELS[kind]] = {"own_us": round(own, 2), "library_us": round(lib, 2), "kept": bool(keep)}
            if keep:
                kept[kind] = handle
    finally:
        shared._special = kept       # one assignment at the end (plans built before it keep the struct they snapshotted)
        shared._ws_cache = None
    path = getattr(shared, "_special_path", None)
    if path and report:      # remembered: a later process attaches exactly the kept entry points without measuring again
        import json
        try:
            with open(os.path.join(cache_dir(), _tuned_name(path)), "w") as f:
                json.dump({"kept": sorted(k for k in kept if k in calls), "batch": batch, "arch": target_arch(), "report": report}, f, indent=1)
        except OSError:
            pass
    return report
