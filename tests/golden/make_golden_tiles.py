#!/usr/bin/env python3
"""Generate tests/golden/golden_tiles_*.npz: THREE FULL 64-ROW TILES (192 joint states) per robot through the UNMODIFIED
reference on its CPU path — forward outputs (FK / Jacobian / inverse dynamics / inertia matrix / forward dynamics) and
torch-autograd gradients of the four learning losses (FK position, inverse dynamics, inertia matrix, forward dynamics)
with respect to the joint state AND the learnable link parameters.

Why: the straight-line production kernels of this repo (one wavefront = one tile of 64 rows) take only FULL tiles; a ragged
tail goes to the loop kernels.  The older fixtures hold 3-48 rows, so they pin the loop kernels; these pin the production
kernels — `rnea_backward_arm_kernel`, `rnea_backward_arm_hand_kernel`, `rnea_backward_fingers_kernel`, `fk_backward_arm_kernel`
and the straight-line forward kernels — to the reference itself (reference robot_model.py:305-375, 669-713;
tests/test_kinematics_dynamics.py:325-377).  Consumed by tests/test_golden_tiles.py.

Same case definitions and the same code as the small fixtures (make_golden_wide / _grad / _grad_dyn / _grad_mass / _grad_fd
`.generate`), only the batch differs.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_tiles.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_grad  # noqa: E402
import make_golden_grad_dyn  # noqa: E402
import make_golden_grad_fd  # noqa: E402
import make_golden_grad_mass  # noqa: E402
import make_golden_wide  # noqa: E402

ROWS = 192   # three full tiles

# FK-position loss: the four cases of make_golden_grad.py plus the arm + hand shapes
FK_EXTRA = [
    ("panda", "panda_description/urdf/panda.urdf", ["panda_hand", "panda_rightfinger"], ["panda_link4", "panda_link7"], ROWS),
    ("jaco", "kinova_description/urdf/jaco.urdf", ["j2n6s300_end_effector"], ["j2n6s300_link_3"], ROWS),
    ("iiwa7_allegro", "kuka_iiwa/urdf/iiwa7_allegro.urdf", ["link_3.0_tip", "link_15.0_tip"], ["iiwa_link_3", "link_13.0"], ROWS),
]
# inertia-matrix loss: make_golden_grad_mass.py's cases plus the hand alone (the "fingers" kernels)
MASS_EXTRA = [
    ("allegro_left", "allegro/urdf/allegro_hand_description_left.urdf",
     {"link_1.0": ["mass", "com", "trans"], "link_14.0": ["inertia_mat", "rot_angles"]}, 64),
]


def rows(cases):
    return [tuple(c[:-1]) + (ROWS,) for c in cases]


def main():
    save = lambda name, d: np.savez_compressed(os.path.join(HERE, name), **d)
    save("golden_tiles_fwd.npz", make_golden_wide.generate(ROWS, 19200))
    save("golden_tiles_grad.npz", make_golden_grad.generate(rows(make_golden_grad.CASES) + FK_EXTRA))
    save("golden_tiles_grad_dyn.npz", make_golden_grad_dyn.generate(rows(make_golden_grad_dyn.CASES)))
    # (the 16- and 23-DoF robots: ONE full tile — H, its target and the weight are [rows, n, n] each)
    mass = [c if c[0] != "iiwa7_allegro" else tuple(c[:-1]) + (64,) for c in rows(make_golden_grad_mass.CASES)]
    save("golden_tiles_grad_mass.npz", make_golden_grad_mass.generate(mass + MASS_EXTRA))
    # (no hand-alone case for forward dynamics: 0.1 rad on a thumb joint of the Allegro URDF — 1e-6 kg m^2 phalanges — turns
    # accelerations of 2 rad/s^2 into 4e4, where the reference's own fp32 recursion is 5 % noise; TriFinger covers the shape)
    save("golden_tiles_grad_fd.npz", make_golden_grad_fd.generate(rows(make_golden_grad_fd.CASES)))


if __name__ == "__main__":
    main()
