"""RNEA backward (K7, csrc/drm_rnea_backward.hip + drm_sample.hpp::rnea_backward_walk).

CPU part (not gpu): the per-sample adjoint sweeps compiled with g++ / clang (tests/host_emu) against gradients
recorded from the UNMODIFIED reference through torch autograd (tests/golden/golden_grad_dyn.npz, made by
tests/golden/make_golden_grad_dyn.py; mirrors examples/learn_dynamics_iiwa.py:49-96).
GPU part (-m gpu): the real kernels through the public API + torch.autograd on the same fixtures, ragged batches
against the host emulation, determinism of the batch reduction.
Tolerance: gradients rtol 1e-3 of the largest entry (observed ~1e-5).
"""
import ctypes

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd.flatten import build_walk
from differentiable_robot_model_amd.rigid_body_params import PositiveScalar, UnconstrainedScalar, UnconstrainedTensor
from helpers import GOLDEN_DIR, load_model, sample_states
from test_host_emu import _ptr, emu, host_walk  # noqa: F401  (emu is a fixture)

import os

# the last four (round 3) are robots with one long segment: persistent backward kernels with HBM-parked records
CASES = ["iiwa7", "panda_no_gripper", "allegro_left", "trifinger_edu", "fetch", "jaco", "panda", "iiwa7_allegro"]
GRAD_RTOL = 1e-3


def load_golden_dyn():
    return np.load(os.path.join(GOLDEN_DIR, "golden_grad_dyn.npz"), allow_pickle=False)


def close(a, b, rtol=GRAD_RTOL):
    """|a - b| <= rtol * max|b|, with an absolute floor of 1e-5: a gradient that is (nearly) zero for a physical reason —
    the mass of a link that only spins about an axis through its origin — is a sum of O(1) per-sample terms that cancel,
    and fp32 cannot resolve it below ~1e-6 whichever way the sweeps are organised."""
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    return np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-2)


def parametrization(pname):
    if pname == "mass":
        return PositiveScalar()
    if pname == "joint_damping":
        return UnconstrainedScalar()
    if pname == "inertia_mat":
        return UnconstrainedTensor(dim1=3, dim2=3)
    return UnconstrainedTensor(dim1=1, dim2=3)


def learnable_model(g, case, device="cpu"):
    """Model with the case's learnable parameters initialised to the values the reference run started from."""
    m = load_model(case, device)
    params = {}
    for key in g[case + "/keys"]:
        link, pname, tensor_name = str(key).split("/")
        mod = parametrization(pname)
        m.make_link_param_learnable(link, pname, mod)
        p = dict(mod.named_parameters())[tensor_name]
        with torch.no_grad():
            p.copy_(torch.from_numpy(g["%s/init/%s" % (case, key)].copy()).reshape(p.shape).to(p.device))
        params[str(key)] = p
    return m, params


def dynamic_param_mask(m, prog):
    links = {link for link, _ in m._learnable}
    mask = 0
    for k, link in enumerate(prog.links):
        if int(link) in links:
            mask |= 1 << k
    return mask


def emu_loss_and_grads(emu, m, q, qd, qdd, want):
    prog = build_walk(m._spec, whole_tree=True)
    assert prog.slots_unique
    table = m._link_table()
    ops_f_t = (table.reshape(-1)[torch.from_numpy(prog.gather.reshape(-1))]
               * torch.from_numpy(prog.gsign.reshape(-1))).reshape(prog.capacity, 32)
    ops_f = np.ascontiguousarray(ops_f_t.detach().numpy(), np.float32)
    walk, _keep = host_walk(m, prog)
    walk.ops_f = ops_f.ctypes.data
    B, n = q.shape
    tau = np.zeros((B, n), np.float32)
    assert emu.emu_rnea(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(tau)) == 0
    loss = ((tau - want) ** 2).mean()
    gtau = np.ascontiguousarray(2.0 * (tau - want) / (B * n), np.float32)
    gq, gqd, gqdd = (np.full((B, n), np.nan, np.float32) for _ in range(3))
    gops = np.full((prog.capacity, 32), np.nan, np.float32)
    mask = dynamic_param_mask(m, prog)
    assert emu.emu_rnea_backward(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(gtau),
                                 ctypes.c_uint64(mask), _ptr(gq), _ptr(gqd), _ptr(gqdd), _ptr(gops)) == 0
    m.zero_grad()
    ops_f_t.backward(torch.from_numpy(gops))
    return loss, tau, gq, gqd, gqdd


@pytest.mark.parametrize("case", CASES)
def test_emu_backward_vs_reference_autograd(emu, case):
    check_emu_backward_vs_reference_autograd(emu, load_golden_dyn(), case)


def check_emu_backward_vs_reference_autograd(emu, g, case):
    m, params = learnable_model(g, case)
    q, qd, qdd = (np.ascontiguousarray(g["%s/%s" % (case, k)]) for k in ("q", "qd", "qdd"))
    loss, tau, gq, gqd, gqdd = emu_loss_and_grads(emu, m, q, qd, qdd, g[case + "/want"])
    assert np.allclose(tau, g[case + "/tau"], atol=2e-5, rtol=2e-5)
    assert abs(loss - float(g[case + "/loss"])) <= 1e-5 * max(1.0, float(g[case + "/loss"]))
    assert close(gq, g[case + "/grad_q"]), np.abs(gq - g[case + "/grad_q"]).max()
    assert close(gqd, g[case + "/grad_qd"]) and close(gqdd, g[case + "/grad_qdd"])
    for key, p in params.items():
        ref = g["%s/grad/%s" % (case, key)]
        assert close(p.grad.numpy(), ref), (case, key, p.grad.numpy().reshape(-1), ref.reshape(-1))


@pytest.mark.parametrize("robot", ["panda_no_gripper", "iiwa7", "fetch_arm_no_gripper_small_damping"])
@pytest.mark.parametrize("flags", [0, 3])
def test_emu_arm_chain_backward_equals_generic_walk(emu, robot, flags):
    """rnea_backward_chain (registers, motions recovered on the way back) against rnea_backward_walk (parked records),
    which the test above pins to the reference's autograd."""
    m = load_model(robot)
    n, B = m._n_dofs, 21
    q, qd, qdd = sample_states(m, B, seed=77)
    gtau = np.random.default_rng(1).standard_normal((B, n)).astype(np.float32)
    prog = build_walk(m._spec, whole_tree=True)
    assert prog.shape & 1
    walk, keep = host_walk(m, prog)
    mask = 0b10110101
    out = {}
    for name in ("emu_rnea_backward", "emu_rnea_backward_arm"):
        gq, gqd, gqdd = (np.full((B, n), np.nan, np.float32) for _ in range(3))
        gops = np.full((prog.capacity, 32), np.nan, np.float32)
        assert getattr(emu, name)(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), flags, _ptr(gtau),
                                  ctypes.c_uint64(mask), _ptr(gq), _ptr(gqd), _ptr(gqdd), _ptr(gops)) == 0
        out[name] = (gq, gqd, gqdd, gops)
    for a, b in zip(out["emu_rnea_backward"], out["emu_rnea_backward_arm"]):
        assert close(b, a, 2e-4), np.abs(a - b).max()


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_backward_vs_reference_autograd(case):
    check_gpu_backward_vs_reference_autograd(load_golden_dyn(), case)


def check_gpu_backward_vs_reference_autograd(g, case, device="cuda", prepare=None):
    m, params = learnable_model(g, case, device)
    if prepare is not None:
        prepare(m)
    q, qd, qdd = (torch.from_numpy(g["%s/%s" % (case, k)].copy()).to(device).requires_grad_(True) for k in ("q", "qd", "qdd"))
    want = torch.from_numpy(g[case + "/want"].copy()).to(device)
    tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    loss = torch.nn.functional.mse_loss(tau, want)
    loss.backward()
    assert abs(loss.item() - float(g[case + "/loss"])) <= 1e-5 * max(1.0, float(g[case + "/loss"]))
    assert close(q.grad.cpu().numpy(), g[case + "/grad_q"])
    assert close(qd.grad.cpu().numpy(), g[case + "/grad_qd"]) and close(qdd.grad.cpu().numpy(), g[case + "/grad_qdd"])
    for key, p in params.items():
        assert close(p.grad.cpu().numpy(), g["%s/grad/%s" % (case, key)]), (case, key)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 63, 64, 65, 700])
def test_gpu_backward_ragged_batches_vs_emu(emu, B):
    g = load_golden_dyn()
    case = "allegro_left"
    m, params = learnable_model(g, case, "cuda")
    mc, params_c = learnable_model(g, case, "cpu")
    q, qd, qdd = sample_states(m, B, seed=B)
    want = np.random.default_rng(B).standard_normal((B, m._n_dofs)).astype(np.float32)
    tq, tqd, tqdd = (torch.from_numpy(a).cuda().requires_grad_(True) for a in (q, qd, qdd))
    grads = []
    for _ in range(2):
        m.zero_grad()
        for t in (tq, tqd, tqdd):
            t.grad = None
        loss = torch.nn.functional.mse_loss(m.compute_inverse_dynamics(tq, tqd, tqdd), torch.from_numpy(want).cuda())
        loss.backward()
        grads.append([p.grad.clone() for p in params.values()])
    assert all(torch.equal(a, b) for a, b in zip(*grads)), "the batch reduction must be deterministic"
    _, _, gq, gqd, gqdd = emu_loss_and_grads(emu, mc, q, qd, qdd, want)
    assert close(tq.grad.cpu().numpy(), gq, 2e-4) and close(tqd.grad.cpu().numpy(), gqd, 2e-4)
    assert close(tqdd.grad.cpu().numpy(), gqdd, 2e-4)
    for key in params:
        assert close(params[key].grad.cpu().numpy(), params_c[key].grad.numpy(), 2e-4), key


@pytest.mark.gpu
def test_gpu_learn_dynamics_loop_lowers_the_loss():
    """The reference's dynamics-learning loop (examples/learn_dynamics_iiwa.py) on the GPU path."""
    torch.manual_seed(0)
    m = load_model("iiwa7", "cuda"); gt = load_model("iiwa7", "cuda")
    m.make_link_param_learnable("iiwa_link_1", "mass", PositiveScalar())
    m.make_link_param_learnable("iiwa_link_1", "inertia_mat", UnconstrainedTensor(dim1=3, dim2=3))
    m.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(dim1=1, dim2=3))
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(m, 4096, seed=3))
    with torch.no_grad():
        want = gt.compute_inverse_dynamics(q, qd, qdd)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    first = None
    for _ in range(40):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(m.compute_inverse_dynamics(q, qd, qdd), want)
        loss.backward()
        opt.step()
        first = first if first is not None else loss.item()
    assert loss.item() < 0.5 * first


@pytest.mark.gpu
@pytest.mark.parametrize("B", [64, 200, 4096])
def test_gpu_arm_chain_backward_vs_emu(emu, B):
    """7-DoF arms take rnea_backward_arm_kernel for the full tiles and the generic kernel for the ragged tail; both
    feed one reduction."""
    g = load_golden_dyn()
    case = "panda_no_gripper"
    m, params = learnable_model(g, case, "cuda")
    mc, params_c = learnable_model(g, case, "cpu")
    q, qd, qdd = sample_states(m, B, seed=B + 1)
    want = np.random.default_rng(B).standard_normal((B, 7)).astype(np.float32)
    tq, tqd, tqdd = (torch.from_numpy(a).cuda().requires_grad_(True) for a in (q, qd, qdd))
    grads = []
    for _ in range(2):
        m.zero_grad()
        for t in (tq, tqd, tqdd):
            t.grad = None
        loss = torch.nn.functional.mse_loss(m.compute_inverse_dynamics(tq, tqd, tqdd), torch.from_numpy(want).cuda())
        loss.backward()
        grads.append([p.grad.clone() for p in params.values()] + [tq.grad.clone()])
    assert all(torch.equal(a, b) for a, b in zip(*grads)), "deterministic"
    _, _, gq, gqd, gqdd = emu_loss_and_grads(emu, mc, q, qd, qdd, want)
    assert close(tq.grad.cpu().numpy(), gq, 3e-4) and close(tqd.grad.cpu().numpy(), gqd, 3e-4)
    assert close(tqdd.grad.cpu().numpy(), gqdd, 3e-4)
    for key in params:
        assert close(params[key].grad.cpu().numpy(), params_c[key].grad.numpy(), 3e-4), key


@pytest.mark.gpu
def test_gpu_every_link_learnable_matches_constant_model():
    """All 20 moving links of the Allegro hand learnable at once (20 rows through the fused walk-table kernel): with the
    parametrisations initialised to the URDF values the torques equal the constant model's, and every parameter gets a
    finite gradient."""
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
    ref = load_model("allegro_left", "cuda")
    m = load_model("allegro_left", "cuda")
    for i, body in enumerate(m._bodies):
        if i == 0:
            continue
        for pname in ("trans", "rot_angles"):
            init = getattr(body, pname)().detach().reshape(1, 3).clone()
            m.make_link_param_learnable(body.name, pname, UnconstrainedTensor(dim1=1, dim2=3, init_tensor=init))
        init = body.inertia.com().detach().reshape(1, 3).clone()
        m.make_link_param_learnable(body.name, "com", UnconstrainedTensor(dim1=1, dim2=3, init_tensor=init))
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(ref, 130, seed=8))
    want = ref.compute_inverse_dynamics(q, qd, qdd)
    tau = m.compute_inverse_dynamics(q, qd, qdd)
    assert torch.allclose(tau, want, atol=1e-6, rtol=1e-6)
    tau.square().sum().backward()
    grads = [p.grad for p in m.parameters()]
    assert len(grads) == 60 and all(g is not None and torch.isfinite(g).all() for g in grads)
    pos_ref, _ = ref.compute_forward_kinematics(q, "link_15.0_tip")
    pos, _ = m.compute_forward_kinematics(q, "link_15.0_tip")
    assert torch.allclose(pos, pos_ref, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("net", ["Symm3DInertiaMatrixNet", "SymmPosDef3DInertiaMatrixNet", "CovParameterized3DInertiaMatrixNet",
                                 "TriangParam3DInertiaMatrixNet"])
def test_gpu_training_step_with_inertia_parametrisation_is_graph_capturable(net):
    """Forward + backward of an inverse-dynamics loss with every inertia parametrisation, captured into a HIP graph and
    replayed: the same loss and gradients as the eager step (no host-side index tensors, no synchronising ops)."""
    from differentiable_robot_model_amd import rigid_body_params as rbp
    torch.manual_seed(0)
    gt = load_model("iiwa7", "cuda")
    m = load_model("iiwa7", "cuda")
    init = gt._bodies[3].inertia.inertia_mat().detach().reshape(3, 3).cpu()
    cls = getattr(rbp, net)
    mod = cls(bias=1e-7, init_param=init) if net == "TriangParam3DInertiaMatrixNet" else cls(init_param=init)
    m.make_link_param_learnable("iiwa_link_3", "inertia_mat", mod)
    m.make_link_param_learnable("iiwa_link_3", "mass", rbp.PositiveScalar())
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(gt, 200, seed=21))
    with torch.no_grad():
        want = gt.compute_inverse_dynamics(q, qd, qdd)

    def step():
        loss = torch.nn.functional.mse_loss(m.compute_inverse_dynamics(q, qd, qdd), want)
        loss.backward()
        return loss

    params = list(m.parameters())
    eager_loss = step().item()
    eager_grads = [p.grad.clone() for p in params]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            m.zero_grad(set_to_none=True)
            step()
    torch.cuda.current_stream().wait_stream(side)
    m.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = step()
    for p in params:
        p.grad.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert abs(static_loss.item() - eager_loss) <= 1e-6 * max(1.0, abs(eager_loss))
    for p, g in zip(params, eager_grads):
        assert torch.allclose(p.grad, g, rtol=1e-5, atol=1e-7), net
