/*
 * drm_hip.h — C ABI of the MI355X (gfx950) batched FK / Jacobian / RNEA engine.
 *
 * The reference (facebookresearch/differentiable-robot-model @ v1) has no FFI:
 * its boundary for this path is the Python method surface of
 * `DifferentiableRobotModel` (reference differentiable_robot_model/robot_model.py:87-754).
 * Each entry point below is what a binding for one of those methods calls;
 * the reference interface it replaces is cited per function.  INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every call returns 0 on
 *     success or a negative DRM_ERR_* code, drm_last_error() (thread-local)
 *     explains it.
 *   - all data pointers are DEVICE pointers to contiguous row-major float32 /
 *     int32 arrays owned by the caller (torch owns them in the Python host);
 *     the library never allocates, frees, copies or synchronises.
 *   - `stream` is a hipStream_t (NULL = the default stream); every call is an
 *     asynchronous launch on that stream and is re-entrant.
 *   - one wavefront (64 lanes) owns a tile of 64 consecutive samples; a lane
 *     owns one sample and walks the robot's flattened tree serially.
 *
 * HOST BUILD.  libdrm_cpu.so (csrc/drm_cpu.cpp) exports the same entry points over HOST pointers for models on
 * device="cpu", the reference's default (robot_model.py:100-104): same arguments, results and error codes; `stream` is
 * ignored and a call returns when its results are written; the forward scratch queries return 0; drm_special_load is
 * DRM_ERR_UNSUPPORTED; drm_fk_mse takes any single-target walk and batch size.  One extra symbol,
 * `drm_cpu_set_threads` (void, takes the thread count as an int).  A host picks the library by where the caller's arrays live — the two never stand
 * in for one another.
 *
 * The robot is handed over as a *walk*: the depth-first list of links a kernel
 * has to visit (flattened on the host once per robot / target set, see
 * differentiable-robot-model_amd/flatten.py) with their constants gathered in
 * walk order, so every per-link constant is a wave-uniform scalar load.
 */
#ifndef DRM_HIP_H
#define DRM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRM_ABI_VERSION 13

/* ---- layout of one op (= one link) of a walk ---------------------------- */
#define DRM_SPECIAL_KINDS 16 /* drm_walk.special[] (the kinds not named below are reserved and must be NULL): */
#define DRM_SPECIAL_RNEA 0   /*   inverse dynamics          kernel drm_rnea_static  */
#define DRM_SPECIAL_CRBA 1   /*   joint-space inertia matrix   kernel drm_crba_static  */
#define DRM_SPECIAL_FD 2     /*   forward dynamics             kernel drm_fd_static    */
#define DRM_SPECIAL_RNEA_BACKWARD 3 /* reverse-mode inverse dynamics  kernel drm_rnea_backward_static, built with the walk's capacity
                                       as the pitch of its rows of partial sums */
/* ABI 10: serial 7-DoF arms (DRM_WALK_ARM_CHAIN, capacity 8) with the robot's CONSTANTS folded into the instruction stream
 * (csrc/drm_arm_stream.hpp instantiated on a constexpr copy of the walk table: products with exact zeros and ones are gone,
 * no table in LDS).  They cover the 128-row tile pairs of a launch of at least DRM_ARM_STATIC_MIN_PAIRS pairs and ignore
 * ops_f: the host guarantees that the table the handle was built from is the walk's (constant models only). */
#define DRM_SPECIAL_RNEA_ARM 4    /* drm_rnea     kernel "drm_rnea_arm_static", arguments q, qd, qdd, n_pairs, flags, tau              */
#define DRM_SPECIAL_FK_RNEA_ARM 5 /* drm_fk_rnea  kernel "drm_fk_rnea_arm_static", arguments q, qd, qdd, n_pairs, flags, tau, pos, quat;
                                     the SAME handle on the tree walk and on the chain walk it was built for                */
/* ... and the one-sample-per-lane chain walks of the same arm (csrc/drm_arm_static.hpp), one wavefront per 64-row tile: */
#define DRM_SPECIAL_CRBA_ARM 6    /* drm_crba              kernel "drm_crba_arm_static", arguments q, n_tiles, H                       */
#define DRM_SPECIAL_FD_ARM 7      /* drm_forward_dynamics  kernel "drm_fd_arm_static", arguments q, qd, f, n_tiles, flags, qdd          */
#define DRM_SPECIAL_RNEA_BACKWARD_ARM 8 /* drm_rnea_backward with param_mask == 0 (input gradients of a constant model): kernel
                                     "drm_rnea_backward_arm_static", arguments q, qd, qdd, grad_tau, n_tiles, flags, grad_q, grad_qd, grad_qdd */
/* ... and on the 2 .. 4 CHAIN walks of a fan-out FK call (the fingertips of a hand): ONE kernel for all of them, a wavefront per
 * chain, every chain's constants folded in (scalar chain walk, csrc/drm_arm_static.hpp); the SAME handle on every chain walk. */
#define DRM_SPECIAL_FK_FAN_LINKS 9 /* drm_fk_fanout_links  kernel "drm_fk_fan_links_static", arguments q, pos, quat, B (int64)       */
/* ABI 11: NOT a kernel — a device `uint32_t` TICKET word owned by the host, one per walk: zero when a call is enqueued and zero again
 * when it has completed (allocate it zeroed once; calls that use it must not overlap on different streams).  With it the backward
 * entry points that end in a fixed-order reduction of per-wavefront partial sums (drm_fk_mse; drm_fk_backward of a 7-DoF arm
 * chain at a multiple of 64 rows) run as ONE launch: every block publishes its rows and takes a ticket, the block that takes
 * the last one adds the rows — same order, same bits as the separate reduction kernel — and resets the word.  NULL: two
 * launches, as before (the reference's counterpart is autograd's accumulation, robot_model.py:669-713 + examples/
 * learn_kinematics_of_iiwa.py:47-55). */
#define DRM_WALK_TICKET 10
/* ABI 11: a serial 7-DoF arm WITH learnable link parameters (the learn-dynamics workload): drm_rnea_backward's kernel for exactly
 * the set of learnable blocks `drm_walk.reserved0` names (bit k: op k has a learnable parameter; the host built the kernel for
 * that set and for the split into kinematic / dynamic blocks its model has) — the constant blocks of the table are compile-time
 * constants of the kernel, the learnable ones are read from ops_f.  Kernel "drm_rnea_backward_arm_param_static", arguments ops_f,
 * q, qd, qdd, grad_tau, n_tiles, flags, grad_q, grad_qd, grad_qdd, partials; launched with 256-thread blocks, the library's rows of
 * partial sums.  Used when the call's param_mask equals reserved0. */
#define DRM_SPECIAL_RNEA_BACKWARD_ARM_PARAM 11
/* ABI 11: drm_fk_rnea_put's form of DRM_SPECIAL_FK_RNEA_ARM (same code object, same pair of walks): kernel
 * "drm_fk_rnea_arm_put_static", arguments q, qd, qdd, n_pairs, flags, tau, pos, quat and the drm_put struct by value */
#define DRM_SPECIAL_FK_RNEA_ARM_PUT 12
/* ABI 11: DRM_SPECIAL_FD_ARM with TWO samples per lane (csrc/drm_arm_static.hpp forward_dynamics_arm2_static_body): covers the
 * 128-row tile pairs of a launch of at least DRM_ARM_STATIC_MIN_PAIRS pairs; kernel "drm_fd_arm2_static", arguments q, qd, f,
 * n_pairs, flags, qdd */
#define DRM_SPECIAL_FD_ARM2 13
/* ... and DRM_SPECIAL_RNEA_BACKWARD_ARM (input gradients, param_mask == 0) with two samples per lane: kernel
 * "drm_rnea_backward_arm2_static", arguments q, qd, qdd, grad_tau, n_pairs, flags, grad_q, grad_qd, grad_qdd */
#define DRM_SPECIAL_RNEA_BACKWARD_ARM2 14
#define DRM_OPF_STRIDE 32 /* floats per op in ops_f                                            */
/* [0..11] "FT block": R_fixed = Rz(yaw)Ry(pitch)Rx(roll) (rigid_body.py:138-143) and the joint origin xyz
 * ("trans", rigid_body.py:48) interleaved as the 8-byte pairs the packed-FP32 chain kernel multiplies with:
 *   (F00 F01) (F10 F11) (F20 F21) (F02 t0) (F12 t1) (F22 t2)                                          */
#define DRM_OPF_FIJ(i, j) ((j) < 2 ? 2 * (i) + (j) : 6 + 2 * (i)) /* entry (i, j) of R_fixed        */
#define DRM_OPF_TI(i) (7 + 2 * (i))                               /* component i of trans           */
#define DRM_OPF_FT_FLOATS 12
#define DRM_OPF_MASS 12   /* [1] link mass                                                     */
#define DRM_OPF_MCOM 13   /* [3] mass * com                (spatial_vector_algebra.py:323)      */
#define DRM_OPF_IO 16     /* [9] I_c + m S(c)S(c)^T        (spatial_vector_algebra.py:324-327)  */
#define DRM_OPF_DAMP 25   /* [1] joint damping             (robot_model.py:368-373)             */

#define DRM_OPI_STRIDE 10 /* int32 fields per op; ops_i is FIELD-MAJOR: ops_i[field * capacity + k] */
#define DRM_OPI_DOF 0     /* DoF column driven by this link's joint, -1 = fixed joint          */
#define DRM_OPI_PERM 1    /* axis canonicalisation of this link's stored frame, a + 3 * (axis negative):
                             a = 2 joint about local z (or fixed), 0 about local x, 1 about local y
                             (rigid_body.py:149-154).  The constants are pre-multiplied by the signed
                             permutation, the kernels rotate every joint about +z of the stored frame by
                             +q and undo the permutation when a target pose is emitted          */
#define DRM_OPI_CTRL 2    /* the op's control word: every field the kernels branch on, packed so that ONE
                             wide scalar load brings the control flow of the whole walk (a lone wave cannot
                             hide a scalar-load round trip per field and op):
                               bits  0..6  DoF column + 1 (0 = fixed joint)
                               bits  7..9  source + 2     (0 = root, 1 = previous op, 2.. = save slot)
                               bits 10..12 save slot + 1  (0 = none)
                               bits 13..19 output slot + 1 (0 = none)
                               bits 20..22 DRM_OPI_PERM code
                               bit  23     DRM_FLAG_CHILD_IS_NEXT
                               bit  24     identity padding op (kernels may skip it)
                               bit  25     prismatic joint (slides along +z of the stored frame by q)
                               bits 26..31 ordinal of a LEAF op (no child follows it in the walk) among the walk's leaves
                             The unpacked fields below stay in the table for hosts and debuggers       */
#define DRM_OPI_CTRL_PACK(dof, src, save, out, perm, flags)                                                       \
    ((((dof) + 1) & 0x7f) | ((((src) + 2) & 7) << 7) | ((((save) + 1) & 7) << 10) | ((((out) + 1) & 0x7f) << 13) | \
     (((perm) & 7) << 20) | (((flags) & 1) << 23))
#define DRM_OPI_SRC 3     /* parent state: DRM_SRC_PREV, DRM_SRC_ROOT, or a save-slot index    */
#define DRM_OPI_SAVE 4    /* save-slot this op's state is copied to (branch point), -1 = none  */
#define DRM_OPI_OUT 5     /* output slot (target index) this op's pose is written to, -1 = none */
#define DRM_OPI_LINK 6    /* link index in URDF <link> order (informational)                   */
#define DRM_OPI_FLAGS 7   /* DRM_FLAG_*                                                        */

#define DRM_OPI_W0 8      /* the two control words of the loop-structured forward kernels (drm_fk, drm_fk_jacobian,
                             drm_rnea, drm_crba, drm_forward_dynamics of robots that are not 7-DoF arm chains): wider
                             fields than DRM_OPI_CTRL, so walks may have any number of links, 16 save slots and
                             prismatic joints.  W0:
                               bits  0..7  DoF column + 1 (0 = fixed joint)
                               bits  8..15 source + 2     (0 = root, 1 = previous op, 2.. = save slot)
                               bits 16..23 save slot + 1  (0 = none)
                               bit  24     DRM_FLAG_CHILD_IS_NEXT
                               bit  25     identity padding op
                               bit  26     prismatic joint: the link slides along +z of its stored frame by q (URDF
                                           type "prismatic"; everything else that moves is revolute about +z)
                               bits 27..29 DRM_OPI_PERM code                                                    */
#define DRM_OPI_W1 9      /* W1: bits 0..15 output slot + 1 (0 = none), bits 16..31 op index of the parent + 1 (0 = root) */
#define DRM_W0_PACK(dof, src, save, child_next, padding, prismatic, perm)                                       \
    ((((dof) + 1) & 0xff) | ((((src) + 2) & 0xff) << 8) | ((((save) + 1) & 0xff) << 16) | (((child_next) & 1) << 24) | \
     (((padding) & 1) << 25) | (((prismatic) & 1) << 26) | (((perm) & 7) << 27))
#define DRM_W1_PACK(out, parent_op) ((((out) + 1) & 0xffff) | ((((parent_op) + 1) & 0xffff) << 16))

#define DRM_SRC_PREV (-1) /* parent = previous op of the walk                                  */
#define DRM_SRC_ROOT (-2) /* parent = the fixed root link (identity pose, zero velocity)       */
#define DRM_FLAG_CHILD_IS_NEXT 1 /* op k+1 is a child of op k                                  */
#define DRM_MAX_SLOTS 16  /* save slots a walk may use in the forward kernels (kept in LDS)    */
#define DRM_MAX_SLOTS_BACKWARD 6 /* ... and in the backward kernels (what the 3-bit source field of DRM_OPI_CTRL addresses: slots 0 .. 5) */
#define DRM_MAX_OPS 64    /* largest walk the BACKWARD kernels and drm_walk_table take; the forward kernels are bounded
                             by the LDS their per-link records need (DRM_ERR_UNSUPPORTED beyond that, ~70 links) */
#define DRM_MAX_SEGMENTS 8 /* independent sub-walks (sub-trees hanging off the fixed root) a dynamics launch fans out
                              over the wavefronts of a block                                      */
#define DRM_MAX_DOFS 64   /* largest supported number of DoF columns                           */

/* drm_walk.shape */
#define DRM_WALK_ARM_CHAIN 1 /* a serial chain: ops 0..n_dofs-1 are moving joints driving DoF columns
                                0..n_dofs-1 in order, every later op is a fixed joint or padding  */
#define DRM_WALK_SERIAL_CHAIN 2 /* a serial chain of ANY shape: op 0 hangs off the root, every other op off the previous one,
                                no save slots, the single target is the last op; revolute, prismatic and fixed ops
                                in any order, any DoF columns.  Walks of capacity 4 / 8 / 12 / 16 with this bit take the
                                straight-line chain kernels (drm_chain_kernels.hip) in drm_fk / drm_fk_jacobian / drm_fk_fanout */
#define DRM_WALK_ARM_HAND 4 /* "an arm that carries a hand": ops 0 .. P-1 form a serial chain (op 0 off the root) and the remaining
                              K * L ops are K serial sub-chains of L ops each, every one hanging off op P-1, in walk order.
                              P, K, L travel in the top byte of shape: */
#define DRM_WALK_AH_P(shape) ((int)(((uint32_t)(shape) >> 24) & 0xf))
#define DRM_WALK_AH_K(shape) ((int)(((uint32_t)(shape) >> 28) & 0x3) + 1)
#define DRM_WALK_AH_L(shape) ((int)(((uint32_t)(shape) >> 30) & 0x3) + 1)
#define DRM_WALK_AH_PACK(P, K, L) ((uint32_t)DRM_WALK_ARM_HAND | ((uint32_t)(P) << 24) | ((uint32_t)((K) - 1) << 28) | ((uint32_t)((L) - 1) << 30))
#define DRM_WALK_FINGERS 16 /* K serial chains of L revolute ops each, every chain hanging off the root, op k driving DoF column k
                              (the fingers of a hand once the fixed joints are folded: Allegro 4 x 4, TriFinger 3 x 3); K and L in
                              the top byte as for DRM_WALK_ARM_HAND (DRM_WALK_AH_K / DRM_WALK_AH_L; P = 0) */
#define DRM_WALK_NO_PRISMATIC 32 /* no op of the walk is a prismatic joint (the two-samples-per-lane fan-out FK kernel asks for it) */
#define DRM_WALK_CHAIN_DOFS 128 /* chain_dof1 / chain_prismatic below are filled (serial chains of <= 16 ops): the straight-line
                                  chain kernels take the DoF columns from the launch arguments instead of reading the control words
                                  first — one dependent memory round trip less before a wavefront's joint angles are requested */
#define DRM_WALK_FK_FAN 64 /* a many-target FK walk (DRM_WALK_TARGETS_ORDERED) that splits behind a hub: ops [0, prefix_end) are
                              what every sub-tree hangs off, seg_begin[j] .. seg_begin[j + 1] the ops of wavefront j's sub-trees (the
                              first op of every range but the first reads its parent's pose from a save slot or the root) */
#define DRM_WALK_TARGETS_ORDERED 8 /* the ops with an output slot carry slots 0, 1, 2, ... in walk order: drm_fk with more than
                                  eight targets then writes its outputs a group of eight consecutive slots at a time */
#define DRM_WALK_LEAVES(shape) (((shape) >> 16) & 0xff) /* number of leaf ops (ops no child follows), see DRM_OPI_CTRL */
#define DRM_WALK_BRANCH_DEPTH(shape) (((shape) >> 8) & 0xff) /* 1 + the largest op index that is a branch
                                point (0: none): sizes the per-ancestor slot records of drm_crba /
                                drm_forward_dynamics                                               */

/* flags of drm_rnea */
#define DRM_RNEA_GRAVITY 1 /* base acceleration (0,0,+9.81)   (robot_model.py:344-350)         */
#define DRM_RNEA_DAMPING 2 /* tau += damping * qd             (robot_model.py:368-373)         */

/* error codes */
#define DRM_OK 0
#define DRM_ERR_INVALID (-1)     /* bad argument (NULL pointer, negative size, capacity ...)   */
#define DRM_ERR_UNSUPPORTED (-2) /* walk larger than any compiled kernel                       */
#define DRM_ERR_LAUNCH (-3)      /* HIP reported a launch error                                */

/*
 * A walk, as produced by flatten.build_walk().  ops_f / ops_i hold `capacity`
 * rows: n_ops real ones followed by identity padding (fixed joint, F = I,
 * t = 0, mass 0, DRM_SRC_PREV); capacity is 4 or 8 for walks of up to 8 links (the
 * 7-DoF arm kernels are compiled for 8 and run the walk as straight-line code), n_ops rounded up to a
 * multiple of 4 beyond that (every other kernel loops over the n_ops links).
 */
typedef struct drm_walk {
    const float *ops_f;   /* device [capacity, DRM_OPF_STRIDE]                               */
    const int32_t *ops_i; /* device [DRM_OPI_STRIDE, capacity]  (field-major)                */
    int32_t n_ops;        /* links visited                                                   */
    int32_t capacity;     /* rows of ops_f / ops_i (>= n_ops, a multiple of 4)               */
    int32_t n_dofs;       /* n = row width of q / qd / qdd / tau and Jacobian column count   */
    int32_t n_slots;      /* save slots used (<= DRM_MAX_SLOTS; backward kernels: <= DRM_MAX_SLOTS_BACKWARD) */
    uint64_t dof_mask;    /* bit d set <=> DoF d is driven by an op of this walk             */
    int32_t target_perm;  /* drm_fk_jacobian: DRM_OPI_PERM code (0..5) of the target (last real) op */
    int32_t shape;        /* DRM_WALK_* bits describing the walk, so launchers can pick a specialised kernel */
    /* Segments: sub-trees that hang off the fixed root are independent dynamics problems (the fingers of a hand on a
     * fixed palm).  Ops seg_begin[s] .. seg_begin[s+1]-1 form segment s (a run of whole root-level sub-trees, in walk
     * order); its joints drive the DoF columns seg_dof_lo[s] .. seg_dof_lo[s] + seg_dof_cnt[s] - 1 and no others.
     * A block of n_segments wavefronts owns 64 samples, wavefront s walks segment s.  n_segments = 1: the whole walk. */
    int32_t n_segments;
    int32_t seg_begin[DRM_MAX_SEGMENTS + 1];
    int32_t seg_dof_lo[DRM_MAX_SEGMENTS];
    int32_t seg_dof_cnt[DRM_MAX_SEGMENTS];
    int32_t prefix_end;   /* ops 0 .. prefix_end-1 are STATIC (fixed joints hanging off the root: a mounting plate, the base
                             link of a TriFinger); every segment replays them before its own ops, forward sweeps only */
    int32_t seg_leaf_begin[DRM_MAX_SEGMENTS + 1]; /* leaf ordinals (DRM_OPI_CTRL bits 26..31) of segment s:
                             seg_leaf_begin[s] .. seg_leaf_begin[s+1]-1 (leaves are numbered in walk order, those of the
                             prefix first); read by the RNEA backward kernel when it fans the segments out over wavefronts */
    uint8_t chain_dof1[16];   /* DRM_WALK_CHAIN_DOFS: 1 + the DoF column op k reads, 0 for an op that does not move (fixed joint,
                                 padding) — what DRM_OPI_W0 says, for the first 16 ops of a serial chain */
    uint32_t chain_prismatic; /* DRM_WALK_CHAIN_DOFS: bit k set <=> op k slides */
    uint32_t reserved0;       /* ABI 11: the param_mask special[DRM_SPECIAL_RNEA_BACKWARD_ARM_PARAM] was built for (else 0) */
    /* ABI 9: per-robot STRAIGHT-LINE kernels for THIS walk (a whole-tree dynamics walk of any shape), or NULL.  Handles from
     * drm_special_load() of a code object the host built from csrc/drm_static.hpp instantiated on this walk's tree
     * (differentiable-robot-model_amd/specialize.py writes and compiles it: ~2 s with hipcc, cached).  When set, the full
     * 64-row tiles of drm_rnea / drm_crba / drm_forward_dynamics run it instead of the loop kernels (no scratch; drm_rnea at any pointer alignment); a ragged tail and every
     * walk without a handle behave as before.  The host guarantees that a handle was built for exactly this walk (n_ops,
     * parents, DoF columns, joint kinds).
     * ABI 10: 12 slots.  Kinds 4 .. 9 (DRM_SPECIAL_*_ARM, DRM_SPECIAL_FK_FAN_LINKS above) — and kinds 0 .. 3 when the host built
     * them with the table as compile-time constants — carry the walk's CONSTANTS in their instruction stream and do not read
     * ops_f: the host attaches them to constant models only and guarantees that the table they were built from is this walk's. */
    const void *special[DRM_SPECIAL_KINDS];
} drm_walk;

int drm_abi_version(void);
/* Load a code object (what `hipcc --genco --offload-arch=gfx950` writes) and look up one kernel: the handle for
 * drm_walk.special[].  The module stays loaded for the life of the process. */
int drm_special_load(const char *code_object_path, const char *kernel_name, const void **function_out);
int drm_walk_sizeof(void); /* sizeof(struct drm_walk) as the library was compiled: a binding checks its mirror of the struct against it */
const char *drm_last_error(void);

/*
 * Forward kinematics of T target links.
 * Replaces DifferentiableRobotModel.compute_forward_kinematics (robot_model.py:223-248:
 * update_kinematic_state 139-195 + CoordinateTransform.get_quaternion
 * spatial_vector_algebra.py:108-136) and, with T = all links,
 * compute_forward_kinematics_all_links (robot_model.py:197-221).
 *   q    [B, n]      joint angles
 *   pos  [B, T, 3]   world position of each target link frame
 *   quat [B, T, 4]   world orientation, xyzw
 * Target t is the op whose DRM_OPI_OUT == t.
 */
int drm_fk(const drm_walk *walk, const float *q, int64_t B, int32_t n_targets,
           float *pos, float *quat, void *stream);

/*
 * Forward kinematics of T target links with LINK-MAJOR outputs: every link's poses a contiguous array.
 * Replaces DifferentiableRobotModel.compute_forward_kinematics_all_links (robot_model.py:197-221), which hands out one
 * (pos [B, 3], quat [B, 4]) pair per link name.
 *   q    [B, n]
 *   pos  [T, B, 3]   link t's positions are pos + t * B * 3
 *   quat [T, B, 4]   xyzw
 * Target t is the op whose DRM_OPI_OUT == t (any order).  A walk with DRM_WALK_FK_FAN is fanned out over wavefronts.
 */
int drm_fk_links(const drm_walk *walk, const float *q, int64_t B, int32_t n_targets, float *pos, float *quat, void *stream);

/*
 * drm_fk for 2 .. 4 targets whose root->target chains are (nearly) disjoint — the fingertips of a hand: `chains[t]`
 * is the single-target walk of target t (all with the same capacity and n_dofs, no branch points); a block of T
 * wavefronts owns 64 samples and wavefront t walks only chain t (the "per-link fan-out" of the parent-index tree).
 * Same outputs as drm_fk with the merged walk: pos [B, T, 3], quat [B, T, 4].
 */
int drm_fk_fanout(const drm_walk *chains, int32_t n_chains, const float *q, int64_t B, float *pos, float *quat,
                  void *stream);

/* The same with LINK-MAJOR outputs, pos [T, B, 3] / quat [T, B, 4] (chain t's poses a contiguous array, as drm_fk_links): each
 * wavefront writes its own chain's arrays, the block shares no tile.  What a caller of compute_forward_kinematics for several
 * fingertips wants: T (pos [B, 3], quat [B, 4]) pairs (robot_model.py:223-248 called once per link). */
int drm_fk_fanout_links(const drm_walk *chains, int32_t n_chains, const float *q, int64_t B, float *pos, float *quat,
                        void *stream);

/*
 * FK + geometric Jacobian of ONE target link; the walk is the root->link chain
 * and its last op is the target.
 * Replaces DifferentiableRobotModel.compute_endeffector_jacobian (robot_model.py:626-667),
 * which itself runs compute_forward_kinematics first (robot_model.py:641).
 *   pos [B,3], quat [B,4] (either may be NULL), lin_jac / ang_jac [B, 3, n];
 *   columns of DoFs that are not on the chain are written as zeros.
 */
int drm_fk_jacobian(const drm_walk *walk, const float *q, int64_t B,
                    float *pos, float *quat, float *lin_jac, float *ang_jac, void *stream);

/*
 * Recursive Newton-Euler inverse dynamics over the whole tree.
 * Replaces DifferentiableRobotModel.compute_inverse_dynamics (robot_model.py:305-375:
 * update_kinematic_state + update_joint_acc rigid_body.py:159-165 +
 * iterative_newton_euler robot_model.py:250-303).  With qdd = NULL the joint
 * accelerations are zero: compute_non_linear_effects (robot_model.py:377-400).
 *   q, qd, qdd [B, n]  ->  tau [B, n];  flags = DRM_RNEA_GRAVITY | DRM_RNEA_DAMPING
 *   scratch   drm_rnea_scratch_floats(walk, B) floats owned by the caller: robots with a long segment (an arm carrying a
 *             gripper or a hand) keep the body force of every link there between the two sweeps,
 *             [resident block][link][6][64] — bounded by what the device holds at once, not by B.  0 for 7-DoF arms and
 *             for hands (short independent fingers): scratch is then never touched and may be NULL
 */
int64_t drm_rnea_scratch_floats(const drm_walk *walk, int64_t B);
int64_t drm_rnea_scratch_floats_aligned(const drm_walk *walk, int64_t B); /* the caller guarantees 16-byte aligned pointers (see Alignment) */
int drm_rnea(const drm_walk *walk, const float *q, const float *qd, const float *qdd, int64_t B,
             int32_t flags, float *tau, float *scratch, void *stream);

/*
 * Inverse dynamics AND the pose of one link in one call: what a caller of the reference gets from
 * compute_inverse_dynamics (robot_model.py:305-375) followed by compute_forward_kinematics (robot_model.py:223-248) on
 * the same q — the reference walks the tree twice, re-evaluating every joint transform (BASELINE configuration 3).
 *   tree      the whole-tree walk drm_rnea takes;   chain   the root->link walk drm_fk takes (one target)
 *   target_op index of the FK target among the ops of `tree`, or -1 if unknown: when the tree is a serial 7-DoF arm and
 *             the target is its last link, both results come from ONE fused launch (one read of q, one sin/cos
 *             evaluation, one constant table); otherwise the two walks are launched one after the other.
 *             FOLDED TREES: a host may leave links behind fixed leaf joints (an end-effector frame) out of `tree` after
 *             adding their inertia to their parents' rows (the dynamics are unchanged; fewer ops).  If `tree` then holds
 *             exactly the n moving joints of a serial 7-DoF arm and `chain` is that arm up to the target, the fused launch
 *             reads ONE table, chain->ops_f: the caller guarantees that its first n rows carry the same constants as
 *             tree->ops_f (both gathered from the same folded link table); target_op is ignored (-1).
 *   q, qd, qdd [B, n] (qdd may be NULL)  ->  tau [B, n], pos [B, 3], quat [B, 4];  flags as for drm_rnea.
 *   scratch   drm_rnea_scratch_floats(tree, B) floats, as for drm_rnea (0 for the fused launch)
 * Results are bit-identical to drm_rnea + drm_fk.
 */
int drm_fk_rnea(const drm_walk *tree, const drm_walk *chain, int32_t target_op, const float *q, const float *qd,
                const float *qdd, int64_t B, int32_t flags, float *tau, float *pos, float *quat, float *scratch, void *stream);

/*
 * ABI 11: drm_fk_rnea with a ONE-SIDED GATHER of its outputs — the exchange step of a batch sharded over the GPUs of a node
 * (BASELINE configuration 3; SURVEY.md §8e asked for direct peer writes instead of a collective after the kernel).  Besides its own
 * tau / pos / quat the call writes the same rows into up to DRM_MAX_PEERS destination sets at row `row_offset`: the gathered arrays
 * of the peers (their device memory, mapped into this process once by hipIpcOpenMemHandle — differentiable-robot-model_amd/
 * distributed.py PeerGather does the exchange), or of rank 0 only.  For a serial 7-DoF arm with its own fused kernel attached
 * (drm_walk.special[DRM_SPECIAL_FK_RNEA_ARM_PUT]) the kernel's epilogue stores every tile to all destinations as it leaves the
 * wavefront — the xGMI links are busy while the walk runs, no collective launch, no second pass over the data; every other walk,
 * and ragged tails, are computed by drm_fk_rnea and copied (hipMemcpyAsync on `stream`, one copy per destination and array).
 * A destination pointer may be NULL (that array is not wanted there, e.g. torques only).  Completion: stream order on THIS GPU —
 * a consumer on another GPU needs its own synchronisation with this rank (a barrier, a flag), as after any one-sided put.
 * The reference has no counterpart (one process, one device: robot_model.py:87-137); the rows are bit-identical to drm_fk_rnea's.
 */
#define DRM_MAX_PEERS 8
typedef struct drm_put {
    int32_t n_peers;             /* 0 .. DRM_MAX_PEERS destination sets                                              */
    int32_t reserved;
    int64_t row_offset;          /* this call's first row in the destinations' arrays (a multiple of 4 for the in-kernel form) */
    float *tau[DRM_MAX_PEERS];   /* [>= row_offset + B, n] each, or NULL                                              */
    float *pos[DRM_MAX_PEERS];   /* [>= row_offset + B, 3]                                                            */
    float *quat[DRM_MAX_PEERS];  /* [>= row_offset + B, 4]                                                            */
} drm_put;
int drm_fk_rnea_put(const drm_walk *tree, const drm_walk *chain, int32_t target_op, const float *q, const float *qd, const float *qdd,
                    int64_t B, int32_t flags, float *tau, float *pos, float *quat, float *scratch, const drm_put *put, void *stream);

/*
 * Joint-space inertia matrix over the whole tree (composite-rigid-body algorithm).
 * Replaces DifferentiableRobotModel.compute_lagrangian_inertia_matrix (robot_model.py:402-450: n + 1
 * compute_inverse_dynamics passes, column j = ID(q, 0, e_j) - ID(q, 0, 0)).  The gravity / damping flags of the
 * reference cancel out of that difference, so there are none here.
 *   q [B, n]  ->  H [B, n, n]   (symmetric; entries of joints on different branches are zero)
 *   scratch   drm_crba_scratch_floats(walk, B) floats owned by the caller: robots with a long segment (an arm carrying a
 *             gripper or a hand) collect the lower triangle of H there, [resident block][segment][entry][64] — bounded by
 *             what the device holds at once, not by B — before its rows are written.  0 for 7-DoF arms and for hands
 *             (short independent fingers): scratch is then never touched and may be NULL
 */
int64_t drm_crba_scratch_floats(const drm_walk *walk, int64_t B);
int64_t drm_crba_scratch_floats_aligned(const drm_walk *walk, int64_t B); /* the caller guarantees 16-byte aligned pointers (see Alignment) */
int drm_crba(const drm_walk *walk, const float *q, int64_t B, float *H, float *scratch, void *stream);

/*
 * Forward dynamics over the whole tree: joint accelerations produced by joint torques f in state (q, qd).
 * Replaces DifferentiableRobotModel.compute_forward_dynamics (robot_model.py:487-624, articulated-body
 * algorithm).  flags as for drm_rnea: DRM_RNEA_GRAVITY = base acceleration (0,0,+9.81) (robot_model.py:527-533),
 * DRM_RNEA_DAMPING = the damping torques damping * qd are taken off f first (robot_model.py:515-521; the
 * caller's f is NOT modified, unlike the reference, which subtracts in place).
 *   q, qd, f [B, n]  ->  qdd [B, n]
 * Three kernels: 7-DoF arm chains and robots whose segments are all short (the fingers of a hand) form H and the bias
 * torques and solve H qdd = f - nle by an L^T D L factorisation in registers / LDS (the same elimination, H is at most
 * 7 x 7 there); every other robot (an arm carrying a gripper or a hand, a mobile manipulator) runs the articulated-body
 * recursion itself, three sweeps over the walk with 8 floats per link and sample between them.
 *   scratch   drm_forward_dynamics_scratch_floats(walk, B) floats owned by the caller: the per-link records of the
 *             articulated-body sweeps, [resident block][link][8][64] — bounded by what the device holds at once (tens of
 *             MB), not by B.  0 for the arm and finger kernels: scratch is then never touched and may be NULL
 */
int64_t drm_forward_dynamics_scratch_floats(const drm_walk *walk, int64_t B);
int64_t drm_forward_dynamics_scratch_floats_aligned(const drm_walk *walk, int64_t B); /* the caller guarantees 16-byte aligned pointers (see Alignment) */
int drm_forward_dynamics(const drm_walk *walk, const float *q, const float *qd, const float *f, int64_t B,
                         int32_t flags, float *qdd, float *scratch, void *stream);

/*
 * Reverse-mode derivative of drm_fk: what torch autograd computes in the reference when a loss on
 * compute_forward_kinematics' outputs is back-propagated to q and to learnable `trans` / `rot_angles`
 * (robot_model.py:139-195, 223-248, 669-713; examples/learn_kinematics_of_iiwa.py:25-61).
 *   q          [B, n]        joint angles of the forward call
 *   grad_pos   [B, T, 3]     dL/dpos of every target
 *   grad_rot   [B, T, 3, 3]  dL/dR of every target's rotation matrix, or NULL.  The reference's quaternion is made of
 *                            entries of R copied inside autograd (spatial_vector_algebra.py:108-136; only its
 *                            normalisation is detached), so a loss on the quaternion is a loss on R: the host layer turns
 *                            dL/dquat into dL/dR (robot_model._quat_grad_to_rot), the kernel carries it through the chain
 *   param_mask               bit k set: produce the constant gradient of op k (its link is learnable)
 *   grad_q     [B, n]        dL/dq, or NULL
 *   grad_ops_f [capacity, DRM_OPF_STRIDE]  dL/dF at DRM_OPF_FIJ(i, j), dL/dt at DRM_OPF_TI(i), summed over the
 *                            batch in a fixed order (deterministic); zeros elsewhere.  NULL iff param_mask == 0.
 *   scratch    drm_fk_backward_scratch_floats(B, capacity) floats, owned by the caller
 * The walk must give every branch point its own save slot (flatten.WalkProgram.slots_unique).
 */
int64_t drm_fk_backward_scratch_floats(int64_t B, int32_t capacity);
int drm_fk_backward(const drm_walk *walk, const float *q, int64_t B, int32_t n_targets, const float *grad_pos,
                    const float *grad_rot, uint64_t param_mask, float *grad_q, float *grad_ops_f, float *scratch, void *stream);

/*
 * Reverse-mode derivative of drm_fk_jacobian: what torch autograd computes in the reference when a loss on
 * compute_endeffector_jacobian's outputs (and, optionally, on the target's position) is back-propagated to q and to
 * learnable `trans` / `rot_angles` (robot_model.py:626-667 on top of 139-195).  The walk is the root->link chain
 * drm_fk_jacobian takes (no branch points).
 *   grad_pos      [B, 3]      dL/dpos of the target, or NULL        grad_rot  [B, 3, 3]  dL/dR of the target, or NULL
 *   grad_lin_jac  [B, 3, n]   dL/dlin_jac      grad_ang_jac  [B, 3, n]   dL/dang_jac
 *   param_mask, grad_q, grad_ops_f, scratch (drm_fk_backward_scratch_floats)   as for drm_fk_backward
 */
int drm_fk_jacobian_backward(const drm_walk *walk, const float *q, int64_t B, const float *grad_pos, const float *grad_rot,
                             const float *grad_lin_jac, const float *grad_ang_jac, uint64_t param_mask, float *grad_q,
                             float *grad_ops_f, float *scratch, void *stream);

/*
 * One training step's worth of the reference's kinematics-learning loop in one pass over q
 * (examples/learn_kinematics_of_iiwa.py:47-55: compute_forward_kinematics (robot_model.py:223-248) -> torch.nn.MSELoss ->
 * backward through robot_model.py:139-195 with learnable `trans` / `rot_angles`, robot_model.py:669-713):
 *   loss[0]     = mean over the B x 3 entries of (pos(q) - target)^2, pos = position of the walk's target (its last op)
 *   grad_q      [B, n] or NULL   d loss / d q
 *   grad_ops_f  [capacity, DRM_OPF_STRIDE]  d loss / d (R_fixed, trans) of the ops in param_mask (as drm_fk_backward), NULL iff
 *               param_mask == 0
 *   target      [B, 3];  scratch: drm_fk_mse_scratch_floats(B, capacity) floats, owned by the caller
 * The chain is walked once per sample (the forward pose, the loss and the closed-form adjoints of drm_fk_backward's chain
 * kernel); the batch sums are reduced in a fixed order (bit-stable from run to run).  Two launches.
 * Takes serial 7-DoF arm chains (DRM_WALK_ARM_CHAIN, capacity 8: the iiwa of BASELINE configuration 5, the Panda), B a
 * multiple of 64, 16-byte aligned pointers; DRM_ERR_UNSUPPORTED otherwise (compose drm_fk + the loss + drm_fk_backward).
 */
int64_t drm_fk_mse_scratch_floats(int64_t B, int32_t capacity);
int drm_fk_mse(const drm_walk *walk, const float *q, const float *target, int64_t B, uint64_t param_mask, float *loss,
               float *grad_q, float *grad_ops_f, float *scratch, void *stream);

/*
 * ABI 12: drm_fk_mse for a walk WITH learnable links, from their URDF-level parameters to the gradients with respect to them —
 * what drm_walk_table -> drm_fk_mse -> drm_walk_table_backward compute (the whole forward + loss + backward of
 * examples/learn_kinematics_of_iiwa.py:47-55 with the parameter modules of rigid_body_params.py in the loop), in TWO launches
 * instead of five: every wavefront of the first launch rebuilds the rows of the learnable links itself (R_fixed = (Rz Ry) Rx of
 * rot_angles, rigid_body.py:138-143; trans), the second adds the rows of partial sums AND takes the result back through that map.
 *   walk->ops_f   the table of the CONSTANT links gathered into walk order (`base` of drm_walk_table)
 *   sel, gsign    [capacity * DRM_OPF_STRIDE] as for drm_walk_table: entry e of the table is element sel[e] % 32 of the link row of
 *                 learnable link sel[e] / 32, times gsign[e] (+-1: the axis canonicalisation), or the constant (sel[e] < 0)
 *   links         HOST array of n_links (1 .. DRM_FK_MSE_MAX_LINKS) structs of DEVICE pointers, one per learnable link: the outputs
 *                 of its parameter modules where they lie (no packing pass).  Forward kinematics reads rot_angles[3] and trans[3];
 *                 the other four may be NULL.
 *   grad_params   [n_links, 20] d loss / d (rot_angles, trans, mass, com, inertia_mat, damping) in drm_walk_table's layout (the last
 *                 14 of a link are zeros: forward kinematics does not depend on them)
 *   loss, grad_q, target, param_mask (!= 0: the ops of the learnable links), scratch (drm_fk_mse_scratch_floats): as for drm_fk_mse,
 *   and the same walks / batches / alignment; DRM_ERR_UNSUPPORTED otherwise (compose the three calls).  Sums in drm_fk_mse's and
 *   drm_walk_table_backward's order: the same bits as the composition.
 */
#define DRM_FK_MSE_MAX_LINKS 8
struct drm_link_pieces {
    const float *rot_angles, *trans, *mass, *com, *inertia_mat, *damping; /* 3, 3, 1, 3, 9, 1 floats */
};
int drm_fk_mse_links(const drm_walk *walk, const int32_t *sel, const float *gsign, const struct drm_link_pieces *links,
                     int32_t n_links, const float *q, const float *target, int64_t B, uint64_t param_mask, float *loss,
                     float *grad_q, float *grad_params, float *scratch, void *stream);

/*
 * Reverse-mode derivative of drm_rnea: what torch autograd computes in the reference when a loss on
 * compute_inverse_dynamics' torques is back-propagated to learnable link parameters and to q / qd / qdd
 * (robot_model.py:305-375, 669-713; examples/learn_dynamics_iiwa.py:49-96).
 *   q, qd, qdd [B, n], flags      the arguments of the forward call (qdd may be NULL = zeros)
 *   grad_tau   [B, n]             dL/dtau
 *   param_mask                    bit k set: produce the constant gradients of op k (its link is learnable)
 *   grad_q, grad_qd, grad_qdd [B, n]   all three or all NULL
 *   grad_ops_f [capacity, DRM_OPF_STRIDE]  gradient of every constant of the selected ops in the op-row layout
 *                                 (FT block, mass, mass*com, I_o, damping), summed over the batch in a fixed order;
 *                                 zeros elsewhere.  NULL iff param_mask == 0.
 *   scratch    drm_rnea_backward_scratch_floats(B, capacity, n_dofs, n_slots) floats, owned by the caller
 * The walk must give every branch point its own save slot (flatten.WalkProgram.slots_unique).
 */
int64_t drm_rnea_backward_scratch_floats(int64_t B, int32_t capacity, int32_t n_dofs, int32_t n_slots);
int drm_rnea_backward(const drm_walk *walk, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags,
                      const float *grad_tau, uint64_t param_mask, float *grad_q, float *grad_qd, float *grad_qdd,
                      float *grad_ops_f, float *scratch, void *stream);

/*
 * Link-table rows from URDF-level link parameters, and the derivative of that map, for the rows of LEARNABLE links
 * (what the reference recomputes from its parameter modules on every call: rigid_body.py:138-143 R_fixed,
 * spatial_vector_algebra.py:321-327 mcom and I_o, and differentiates with one autograd node per tiny op).
 *   params [n_links, 20]  rpy (3) trans (3) mass (1) com (3) inertia_mat (9, about the com) damping (1)
 *   rows   [n_links, 32]  F (9, row-major) t (3) mass mcom (3) I_o (9) damping 0...   (the host gathers these into
 *                         walk order; flatten._gather_row)
 */
int drm_link_rows(const float *params, int32_t n_links, float *rows, void *stream);
int drm_link_rows_backward(const float *params, const float *grad_rows, int32_t n_links, float *grad_params, void *stream);

/*
 * The walk table (ops_f) of a robot with learnable links in ONE launch, and the derivative of that map in another: what
 * the reference does implicitly by calling its parameter modules inside every per-link op (rigid_body.py:138-143,
 * spatial_vector_algebra.py:321-327) and differentiating through them.
 *   params   [n_links, 20]   URDF-level parameters of the learnable links (drm_link_rows layout), n_links <= 32
 *   base     [n_entries]     the walk table built from the constant links (entries of learnable links are ignored)
 *   sel      [n_entries]     int32: index into the flattened [n_links, 32] rows for entries that come from a learnable
 *                            link, -1 for entries taken from `base`
 *   gsign    [n_entries]     the exact +-1 factors of the axis canonicalisation (flatten.WalkProgram.gsign)
 *   ops_f    [n_entries]     out (n_entries = capacity * DRM_OPF_STRIDE)
 * drm_walk_table_backward: grad_ops_f [n_entries] -> grad_params [n_links, 20], summed in a fixed order.
 */
int drm_walk_table(const float *params, int32_t n_links, const float *base, const int32_t *sel, const float *gsign,
                   int32_t n_entries, float *ops_f, void *stream);
int drm_walk_table_backward(const float *params, int32_t n_links, const float *grad_ops_f, const int32_t *sel,
                            const float *gsign, int32_t n_entries, float *grad_params, void *stream);

/*
 * ABI 13: the walk table straight from the learnable links' parameter TENSORS, where they lie and in the form their modules store
 * them, and the derivative of that map back to those tensors — one launch each.  What the reference does by calling the parameter
 * modules inside every per-link op (rigid_body.py:138-143; rigid_body_params.py:26-43 PositiveScalar, :252-404 the inertia-matrix
 * modules) and differentiating through them with one autograd node per tiny op: a learn-dynamics step
 * (examples/learn_dynamics_iiwa.py:49-96) spends 14 + 1 + 21 launches there (l * l + min per link, the cat of 42 pieces, the
 * backward of the squares) around drm_walk_table's two.
 *   links  [n_links]   where rot_angles, trans, mass, com, inertia_mat, damping of a learnable link lie (device addresses; constant
 *                      pieces point at the constants).  A piece in a form other than DRM_FORM_PLAIN points at the module's RAW parameter:
 *                      one float for the scalars, l[6] = (diagonal entries, then (1,0), (2,0), (2,1)) for inertia_mat
 *   forms  [n_links]   the form of mass / inertia_mat / damping and its constant (min_val, bias); NULL = everything plain
 *   base, sel, gsign, n_entries, ops_f   as for drm_walk_table
 * drm_walk_table_links_backward: grad_ops_f [n_entries] -> grad_params [n_links, 20] in drm_link_rows' layout, with respect to what lies
 * AT THE ADDRESSES (the raw parameters; inertia_mat in a six-number form fills the first six of its nine places, zeros behind).
 * Sums in drm_walk_table_backward's order.  At most 32 links.
 */
#define DRM_FORM_PLAIN 0
#define DRM_FORM_SQUARE_PLUS 1 /* mass, damping: l * l + c */
#define DRM_FORM_SYMM 2        /* inertia_mat: symmetric from l[6] */
#define DRM_FORM_SPD 3         /* inertia_mat: L L^T + c E */
#define DRM_FORM_COV 4         /* inertia_mat: tr(S) E - S, S = L L^T + c E */
struct drm_link_forms {
    int32_t mass, inertia_mat, damping;
    float mass_c, inertia_mat_c, damping_c;
};
int drm_walk_table_links(const struct drm_link_pieces *links, const struct drm_link_forms *forms, int32_t n_links, const float *base,
                         const int32_t *sel, const float *gsign, int32_t n_entries, float *ops_f, void *stream);
int drm_walk_table_links_backward(const struct drm_link_pieces *links, const struct drm_link_forms *forms, int32_t n_links,
                                  const float *grad_ops_f, const int32_t *sel, const float *gsign, int32_t n_entries,
                                  float *grad_params, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DRM_HIP_H */
