"""tests/golden/golden_tiles_*.npz: THREE FULL 64-row tiles (192 joint states) per robot through the UNMODIFIED reference
(tests/golden/make_golden_tiles.py) — forward outputs and torch-autograd gradients of the four learning losses.

The straight-line production kernels (one wavefront = one 64-row tile: `rnea_backward_arm_kernel`,
`rnea_backward_arm_hand_kernel`, `rnea_backward_fingers_kernel`, `fk_backward_arm_kernel`, the arm / arm + hand / fingers
forward kernels) take FULL tiles only and hand a ragged tail to the loop kernels, so the 3-48-row fixtures pin the loop
kernels; with 192 rows every row of these tests is computed by the production kernel and compared with the reference ITSELF
(input gradients AND parameter gradients), not with a sibling kernel (reference robot_model.py:305-375, 669-713;
tests/test_kinematics_dynamics.py:325-377).  Same checks and tolerances as the small fixtures' tests (the bodies are shared).

CPU (not gpu): the host emulation of the same kernel arithmetic and the oracle against the same fixtures.
"""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, GOLDEN_ROBOTS
from test_host_emu import emu  # noqa: F401  (fixture)
import test_fk_backward as fkb
import test_forward_dynamics as fdt
import test_golden_wide as wide
import test_mass_matrix as mmt
import test_rnea_backward as rbt


def tiles(kind):
    return np.load(os.path.join(GOLDEN_DIR, "golden_tiles_%s.npz" % kind), allow_pickle=False)


FK_CASES = ["iiwa7", "panda_no_gripper", "allegro_left", "trifinger_edu", "panda", "jaco", "iiwa7_allegro"]
DYN_CASES = rbt.CASES
MASS_CASES = mmt.H_GRAD_CASES + ["allegro_left"]
FD_CASES = fdt.FD_GRAD_CASES
# launches this large take the two-samples-per-lane arm kernels (rnea_arm2_kernel: beyond 1 024 tiles = 65 536 rows)
TWO_SAMPLE_REPEAT = 400      # 400 x 192 = 76 800 rows = 1 200 tiles


def test_fixtures_hold_full_tiles():
    fwd = tiles("fwd")
    for robot, links in GOLDEN_ROBOTS:
        assert fwd[robot + "/q"].shape[0] == 192 and list(fwd[robot + "/links"]) == links
    for kind, cases in (("grad", FK_CASES), ("grad_dyn", DYN_CASES), ("grad_mass", MASS_CASES), ("grad_fd", FD_CASES)):
        g = tiles(kind)
        for case in cases:
            assert g[case + "/q"].shape[0] % 64 == 0 and g[case + "/q"].shape[0] >= 64, (kind, case)


# ------------------------------------------------------------------------------------------------ CPU: oracle / host emulation
@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
def test_oracle_vs_reference_tiles(robot, links):
    wide.check_oracle_vs_reference(tiles("fwd"), robot, links, np.float32)


@pytest.mark.parametrize("case", FK_CASES)
def test_emu_fk_backward_tiles(emu, case):
    fkb.check_emu_backward_vs_reference_autograd(emu, tiles("grad"), case)


@pytest.mark.parametrize("case", DYN_CASES)
def test_emu_rnea_backward_tiles(emu, case):
    rbt.check_emu_backward_vs_reference_autograd(emu, tiles("grad_dyn"), case)


@pytest.mark.parametrize("case", ["panda_no_gripper", "trifinger_edu", "panda"])   # (n backward sweeps per case on one core)
def test_emu_mass_matrix_backward_tiles(emu, case):
    mmt.check_emu_mass_matrix_backward_vs_reference_autograd(emu, tiles("grad_mass"), case)


@pytest.mark.parametrize("case", FD_CASES)
def test_emu_forward_dynamics_backward_tiles(emu, case):
    fdt.check_emu_forward_dynamics_backward_vs_reference_autograd(emu, tiles("grad_fd"), case)


# ------------------------------------------------------------------------------------------------ GPU: the production kernels
@pytest.mark.gpu
@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
def test_gpu_forward_kernels_vs_reference_tiles(robot, links):
    wide.check_gpu_vs_reference(tiles("fwd"), robot, links)


@pytest.mark.gpu
@pytest.mark.parametrize("robot,links", [r for r in GOLDEN_ROBOTS if r[0] in ("panda_no_gripper", "iiwa7")])
def test_gpu_two_samples_per_lane_kernels_vs_reference_tiles(robot, links):
    wide.check_gpu_vs_reference(tiles("fwd"), robot, links, repeat=TWO_SAMPLE_REPEAT)


@pytest.mark.gpu
@pytest.mark.parametrize("case", FK_CASES)
def test_gpu_fk_backward_vs_reference_autograd_tiles(case):
    fkb.check_gpu_backward_vs_reference_autograd(tiles("grad"), case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["iiwa7", "panda_no_gripper"])
def test_gpu_fused_fk_mse_step_vs_reference_autograd_tiles(case):
    """fk_mse_loss (drm_fk_mse: forward kinematics, MSE and gradients from ONE pass over q) against the loss and the gradients
    torch autograd produced through the UNMODIFIED reference for compute_forward_kinematics -> MSELoss -> backward
    (examples/learn_kinematics_of_iiwa.py:47-55), and against this package's own composition of the three."""
    import torch
    g = tiles("grad")
    targets = [str(t) for t in g[case + "/targets"]]
    assert len(targets) == 1
    m = fkb.learnable_model(g, case, "cuda")
    q = torch.from_numpy(g[case + "/q"].copy()).cuda().requires_grad_(True)
    want = torch.from_numpy(g["%s/want/%s" % (case, targets[0])].copy()).cuda()
    loss = m.fk_mse_loss(q, targets[0], want)
    assert loss.grad_fn is not None and type(loss.grad_fn).__name__.startswith("_FkMse")     # (the fused node, not the fallback)
    loss.backward()
    assert abs(loss.item() - float(g[case + "/loss"])) < 1e-6
    assert fkb.close(q.grad.cpu().numpy(), g[case + "/grad_q"])
    grads = {}
    for link in g[case + "/learnable"]:
        body = m._bodies[m._name_to_idx_map[str(link)]]
        for pname in ("trans", "rot_angles"):
            grads[(str(link), pname)] = getattr(body, pname).param.grad.clone()
            assert fkb.close(grads[(str(link), pname)].cpu().numpy(), g["%s/grad/%s/%s" % (case, link, pname)]), (case, link, pname)
    # the composition it replaces: same loss and gradients to rounding
    m2 = fkb.learnable_model(g, case, "cuda")
    q2 = q.detach().clone().requires_grad_(True)
    loss2 = torch.nn.functional.mse_loss(m2.compute_forward_kinematics(q2, targets[0])[0], want)
    loss2.backward()
    assert abs(loss2.item() - loss.item()) < 1e-7 * max(1.0, abs(loss.item())) + 1e-9
    assert fkb.close(q.grad.cpu().numpy(), q2.grad.cpu().numpy(), 1e-5)
    for (link, pname), got in grads.items():
        ref = getattr(m2._bodies[m2._name_to_idx_map[link]], pname).param.grad
        assert fkb.close(got.cpu().numpy(), ref.cpu().numpy(), 1e-4), (link, pname)
    # no graph: the plain value; a ragged batch: the composed path, same number
    with torch.no_grad():
        assert abs(m.fk_mse_loss(q.detach(), targets[0], want).item() - loss.item()) < 1e-9
        rag = m.fk_mse_loss(q.detach()[:100], targets[0], want[:100])
        ref = torch.nn.functional.mse_loss(m.compute_forward_kinematics(q.detach()[:100], targets[0])[0], want[:100])
        assert abs(rag.item() - ref.item()) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("case", DYN_CASES)
def test_gpu_rnea_backward_vs_reference_autograd_tiles(case):
    rbt.check_gpu_backward_vs_reference_autograd(tiles("grad_dyn"), case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", MASS_CASES)
def test_gpu_mass_matrix_backward_vs_reference_autograd_tiles(case):
    mmt.check_gpu_mass_matrix_backward_vs_reference_autograd(tiles("grad_mass"), case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", FD_CASES)
def test_gpu_forward_dynamics_backward_vs_reference_autograd_tiles(case):
    fdt.check_gpu_forward_dynamics_backward_vs_reference_autograd(tiles("grad_fd"), case)
