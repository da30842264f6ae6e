"""FK backward (K5, csrc/drm_fk_backward.hip + drm_sample.hpp::fk_backward_walk).

CPU part (not gpu): the per-sample adjoint sweep compiled with g++ (tests/host_emu) against
  * gradients recorded from the UNMODIFIED reference through torch autograd (tests/golden/golden_grad.npz,
    made by tests/golden/make_golden_grad.py; mirrors examples/learn_kinematics_of_iiwa.py:25-61),
  * central differences of the fp64 oracle.
GPU part (-m gpu): the real kernels through the public API + torch.autograd, same fixtures, plus BASELINE
config 5 (iiwa, batch 16 384) against the host emulation and determinism of the batch reduction.
Tolerance: gradients rtol 1e-3 of the largest entry (observed ~1e-5).
"""
import ctypes

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd.flatten import build_walk
from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
from helpers import load_golden_grad, load_model, sample_states
from oracle import Oracle
from test_host_emu import _ptr, emu, host_walk  # noqa: F401  (emu is a fixture)

CASES = ["iiwa7", "panda_no_gripper", "allegro_left", "trifinger_edu"]
GRAD_RTOL = 1e-3


def close(a, b, rtol=GRAD_RTOL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-12)


def learnable_model(g, case, device="cpu"):
    m = load_model(case, device)
    for link in g[case + "/learnable"]:
        for pname in ("trans", "rot_angles"):
            init = torch.from_numpy(g["%s/init/%s/%s" % (case, link, pname)].copy())
            m.make_link_param_learnable(str(link), pname, UnconstrainedTensor(dim1=1, dim2=3, init_tensor=init))
    return m


def emu_loss_and_grads(emu, m, q, targets, wants):
    """Forward + backward through the host emulation and torch autograd of the constant table (CPU)."""
    idx = [m._name_to_idx_map[t] for t in targets]
    prog = build_walk(m._spec, targets=idx)
    assert prog.slots_unique
    table = m._link_table()
    ops_f_t = (table.reshape(-1)[torch.from_numpy(prog.gather.reshape(-1))]
               * torch.from_numpy(prog.gsign.reshape(-1))).reshape(prog.capacity, 32)
    ops_f = np.ascontiguousarray(ops_f_t.detach().numpy(), np.float32)
    walk, _keep = host_walk(m, prog)
    walk.ops_f = ops_f.ctypes.data
    B, T, n = q.shape[0], len(idx), m._n_dofs
    pos = np.zeros((B, T, 3), np.float32); quat = np.zeros((B, T, 4), np.float32)
    assert emu.emu_fk(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), T, _ptr(pos), _ptr(quat)) == 0
    want = np.stack(wants, axis=1)
    loss = sum(((pos[:, t] - want[:, t]) ** 2).mean() for t in range(T))
    gpos = np.ascontiguousarray(2.0 * (pos - want) / (B * 3), np.float32)
    gq = np.full((B, n), np.nan, np.float32); gops = np.full((prog.capacity, 32), np.nan, np.float32)
    mask = m._kinematic_param_mask(type("W", (), {"program": prog})())
    assert emu.emu_fk_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), T, _ptr(gpos), ctypes.c_uint64(mask),
                               _ptr(gq), _ptr(gops)) == 0
    m.zero_grad()
    ops_f_t.backward(torch.from_numpy(gops))
    return loss, pos, gq, gops, mask


@pytest.mark.parametrize("case", CASES)
def test_emu_backward_vs_reference_autograd(emu, case):
    check_emu_backward_vs_reference_autograd(emu, load_golden_grad(), case)


def check_emu_backward_vs_reference_autograd(emu, g, case):
    m = learnable_model(g, case)
    targets = [str(t) for t in g[case + "/targets"]]
    q = np.ascontiguousarray(g[case + "/q"])
    wants = [g["%s/want/%s" % (case, t)] for t in targets]
    loss, pos, gq, gops, mask = emu_loss_and_grads(emu, m, q, targets, wants)
    assert abs(loss - float(g[case + "/loss"])) < 1e-6
    for t_i, t in enumerate(targets):
        assert np.abs(pos[:, t_i] - g["%s/pos/%s" % (case, t)]).max() < 2e-6
    assert close(gq, g[case + "/grad_q"]), np.abs(gq - g[case + "/grad_q"]).max()
    for link in g[case + "/learnable"]:
        body = m._bodies[m._name_to_idx_map[str(link)]]
        for pname in ("trans", "rot_angles"):
            got = getattr(body, pname).param.grad.numpy()
            ref = g["%s/grad/%s/%s" % (case, link, pname)]
            assert close(got, ref), (case, link, pname, got, ref)
    # ops outside the mask (and the padding) get exactly zero constant gradient
    for k in range(gops.shape[0]):
        if not (mask >> k) & 1:
            assert not gops[k].any()


@pytest.mark.parametrize("robot", ["iiwa7", "allegro_left", "trifinger_edu", "panda", "jaco_clean"])
def test_emu_grad_q_vs_oracle_central_differences(emu, robot):
    m = load_model(robot)
    n, B = m._n_dofs, 5
    L = len(m._bodies)
    leaves = [i for i in range(1, L) if not m._spec.children[i]]
    prog = build_walk(m._spec, targets=leaves)
    walk, _keep = host_walk(m, prog)
    q, _, _ = sample_states(m, B, seed=11)
    T = len(leaves)
    rng = np.random.default_rng(5)
    gpos = rng.standard_normal((B, T, 3)).astype(np.float32)
    gq = np.full((B, n), np.nan, np.float32)
    assert emu.emu_fk_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), T, _ptr(gpos), ctypes.c_uint64(0),
                               _ptr(gq), None) == 0
    orc = Oracle(m._spec)
    h = 1e-6
    num = np.zeros((B, n))
    for d in range(n):
        qp = q.astype(np.float64); qm = q.astype(np.float64)
        qp[:, d] += h; qm[:, d] -= h
        pp, _ = orc.fk(qp, leaves, np.float64); pm, _ = orc.fk(qm, leaves, np.float64)
        num[:, d] = ((pp - pm) / (2 * h) * gpos).sum(axis=(1, 2))
    assert close(gq, num, 2e-5), np.abs(gq - num).max()


# ---------------------------------------------------------------------------------------------- GPU
def _api_loss(m, q, targets, wants):
    loss = 0.0
    for t, w in zip(targets, wants):
        pos, _ = m.compute_forward_kinematics(q, t)
        loss = loss + torch.nn.functional.mse_loss(pos, w)
    return loss


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_backward_vs_reference_autograd(case):
    check_gpu_backward_vs_reference_autograd(load_golden_grad(), case)


def check_gpu_backward_vs_reference_autograd(g, case, device="cuda"):
    m = learnable_model(g, case, device)
    targets = [str(t) for t in g[case + "/targets"]]
    q = torch.from_numpy(g[case + "/q"].copy()).to(device).requires_grad_(True)
    wants = [torch.from_numpy(g["%s/want/%s" % (case, t)].copy()).to(device) for t in targets]
    # one call per target, as the reference's loop would do it
    loss = _api_loss(m, q, targets, wants)
    loss.backward()
    assert abs(loss.item() - float(g[case + "/loss"])) < 1e-6
    assert close(q.grad.cpu().numpy(), g[case + "/grad_q"])
    for link in g[case + "/learnable"]:
        body = m._bodies[m._name_to_idx_map[str(link)]]
        for pname in ("trans", "rot_angles"):
            assert close(getattr(body, pname).param.grad.cpu().numpy(), g["%s/grad/%s/%s" % (case, link, pname)]), \
                (case, link, pname)
    # all targets from ONE multi-target walk (branch-point slots, adjoint slots) give the same gradients
    m2 = learnable_model(g, case, device)
    q2 = q.detach().clone().requires_grad_(True)
    poses = m2.compute_forward_kinematics_all_links(q2)
    loss2 = sum(torch.nn.functional.mse_loss(poses[t][0], w) for t, w in zip(targets, wants))
    loss2.backward()
    assert close(q2.grad.cpu().numpy(), g[case + "/grad_q"])
    for link in g[case + "/learnable"]:
        b1 = m._bodies[m._name_to_idx_map[str(link)]]; b2 = m2._bodies[m2._name_to_idx_map[str(link)]]
        for pname in ("trans", "rot_angles"):
            assert close(getattr(b2, pname).param.grad.cpu().numpy(), getattr(b1, pname).param.grad.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 63, 64, 65, 1000])
def test_gpu_backward_ragged_batches_vs_emu(emu, B):
    g = load_golden_grad()
    case = "allegro_left"
    targets = [str(t) for t in g[case + "/targets"]]
    m = learnable_model(g, case, "cuda"); mc = learnable_model(g, case, "cpu")
    q, _, _ = sample_states(m, B, seed=B)
    rng = np.random.default_rng(B)
    wants = [rng.standard_normal((B, 3)).astype(np.float32) * 0.1 for _ in targets]
    qt = torch.from_numpy(q).cuda().requires_grad_(True)
    poses = m.compute_forward_kinematics_all_links(qt)
    loss = sum(torch.nn.functional.mse_loss(poses[t][0], torch.from_numpy(w).cuda()) for t, w in zip(targets, wants))
    loss.backward()
    _, _, gq, _, _ = emu_loss_and_grads(emu, mc, q, targets, wants)
    assert close(qt.grad.cpu().numpy(), gq, 1e-4)
    for link in g[case + "/learnable"]:
        b1 = m._bodies[m._name_to_idx_map[str(link)]]; b2 = mc._bodies[mc._name_to_idx_map[str(link)]]
        for pname in ("trans", "rot_angles"):
            assert close(getattr(b1, pname).param.grad.cpu().numpy(), getattr(b2, pname).param.grad.numpy(), 1e-4)


@pytest.mark.gpu
def test_gpu_config5_iiwa_learnable_batch_16384(emu):
    """BASELINE config 5: iiwa, learnable trans + rot_angles of iiwa_link_1, batch 16 384, FK + backward."""
    torch.manual_seed(0)
    B = 16384
    m = load_model("iiwa7", "cuda"); mc = load_model("iiwa7", "cpu")
    init_t = torch.empty(1, 3).normal_(mean=0.0, std=0.1); init_r = torch.empty(1, 3).normal_(mean=0.0, std=0.1)
    for mm in (m, mc):
        mm.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(1, 3, init_tensor=init_t.clone()))
        mm.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(1, 3, init_tensor=init_r.clone()))
    gt = load_model("iiwa7", "cuda")
    q, _, _ = sample_states(m, B, seed=5)
    qt = torch.from_numpy(q).cuda()
    with torch.no_grad():
        want, _ = gt.compute_forward_kinematics(qt, "iiwa_link_ee")
    grads = []
    for _ in range(2):
        m.zero_grad()
        pos, _ = m.compute_forward_kinematics(qt, "iiwa_link_ee")
        loss = torch.nn.functional.mse_loss(pos, want)
        loss.backward()
        grads.append([p.grad.clone() for p in m.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*grads)), "the batch reduction must be deterministic"
    _, _, _, _, _ = emu_loss_and_grads(emu, mc, q, ["iiwa_link_ee"], [want.cpu().numpy()])
    for pg, pc in zip(m.parameters(), mc.parameters()):
        assert close(pg.grad.cpu().numpy(), pc.grad.numpy(), 1e-4), (pg.grad, pc.grad)
    # an Adam step on these gradients lowers the loss (the learning loop of the reference example)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    first = None
    for _ in range(20):
        opt.zero_grad()
        pos, _ = m.compute_forward_kinematics(qt, "iiwa_link_ee")
        loss = torch.nn.functional.mse_loss(pos, want)
        loss.backward()
        opt.step()
        first = first if first is not None else loss.item()
    assert loss.item() < 0.7 * first





@pytest.mark.gpu
def test_gpu_fk_backward_beyond_one_tile_per_wavefront():
    """Launches of more than 1 024 tiles take the table-in-LDS form of fk_backward_arm_kernel (round 5; the register-resident table is
    the latency form of smaller launches).  1 030 tiles + 9 rows of the iiwa with a learnable link offset: the whole batch in one call
    against the same rows in chunks of at most 1 024 tiles (the other form) — per-row input gradients to a few ulp, the parameter
    gradients (sums over the batch) to 1e-4 — for the autograd path (drm_fk_backward) and for fk_mse_loss (drm_fk_mse: full tiles)."""
    torch.manual_seed(0)
    m = load_model("iiwa7", "cuda")
    m.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(1, 3))
    m.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(1, 3))
    gt = load_model("iiwa7", "cuda")
    for B, fused in ((1030 * 64 + 9, False), (1030 * 64, True)):
        q = torch.from_numpy(sample_states(m, B, seed=15)[0]).cuda()
        with torch.no_grad():
            want, _ = gt.compute_forward_kinematics(q, "iiwa_link_ee")
            want = want + 0.01

        def grads(rows):
            m.zero_grad()
            x = q[rows].clone().requires_grad_(True)
            if fused:
                loss = m.fk_mse_loss(x, "iiwa_link_ee", want[rows]) * (x.shape[0] / B)
            else:
                loss = torch.nn.functional.mse_loss(m.compute_forward_kinematics(x, "iiwa_link_ee")[0], want[rows], reduction="sum") / (3 * B)
            loss.backward()
            return float(loss.detach()), x.grad, [p.grad.clone() for p in m.parameters()]

        whole = grads(slice(0, B))
        parts = [grads(slice(lo, min(B, lo + 512 * 64))) for lo in range(0, B, 512 * 64)]
        assert abs(whole[0] - sum(p[0] for p in parts)) <= 1e-5 * abs(whole[0])
        gq = torch.cat([p[1] for p in parts])
        assert float((whole[1] - gq).abs().max()) <= 1e-6 * max(1e-9, float(gq.abs().max())) + 1e-12
        for k, g in enumerate(whole[2]):
            ref = sum(p[2][k] for p in parts)
            assert float((g - ref).abs().max()) <= 1e-4 * max(1e-9, float(ref.abs().max())), (fused, k)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [64, 16384, 1030 * 64, 2500 * 64])
def test_gpu_one_launch_backward_reduction_equals_the_two_launch_form(B, monkeypatch):
    """ABI 11 (drm_walk.special[DRM_WALK_TICKET]): drm_fk_mse and drm_fk_backward of a 7-DoF arm chain reduce their per-wavefront
    partial sums in the block that finishes LAST instead of in a second launch — opt-in (DRM_TICKET=1): correct, bit-identical, and
    slower than the kernel boundary it removes (backend._ticket has the numbers).  Same sums in the same order: loss and parameter
    gradients BIT FOR BIT the two-launch form's,
    launch after launch (the ticket word resets itself), eagerly and replayed from a hipGraph; sizes on both sides of the
    register-table / LDS-table switch (1 024 tiles) and of the cap on the launch's wavefronts (2 048)."""
    from differentiable_robot_model_amd import backend
    torch.manual_seed(0)
    models = []
    for ticket in ("0", "1"):
        monkeypatch.setenv("DRM_TICKET", ticket)
        m = load_model("iiwa7", "cuda")
        torch.manual_seed(1)
        m.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(1, 3))
        m.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(1, 3))
        m._fk_mse_links = False      # (the composition WalkTable + drm_fk_mse: drm_fk_mse_links has no ticket form)
        models.append(m)
    two, one = models
    q = torch.from_numpy(sample_states(one, B, seed=B)[0]).cuda()
    with torch.no_grad():
        want = load_model("iiwa7", "cuda").compute_forward_kinematics(q, "iiwa_link_ee")[0] + 0.02

    def step(m, fused, ticket):
        monkeypatch.setenv("DRM_TICKET", ticket)
        m.zero_grad()
        x = q.clone().requires_grad_(True)
        loss = m.fk_mse_loss(x, "iiwa_link_ee", want) if fused else torch.nn.functional.mse_loss(m.compute_forward_kinematics(x, "iiwa_link_ee")[0], want)
        loss.backward()
        return [loss.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in m.parameters()]

    for fused in (True, False):
        ref = step(two, fused, "0")
        for rep in range(3):
            got = step(one, fused, "1")
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), (fused, rep)
    ee = one._name_to_idx_map["iiwa_link_ee"]
    prog = one._get_walk(("fk", (ee,)), targets=[ee]).program
    assert getattr(prog, "_ticket", None) is not None and int(prog._ticket.item()) == 0          # (back to zero after every launch)
    assert getattr(two._get_walk(("fk", (ee,)), targets=[ee]).program, "_ticket", None) is None
    # replayed from a hipGraph
    monkeypatch.setenv("DRM_TICKET", "1")
    dw = one._get_walk(("fk", (ee,)), targets=[ee])
    ops_f, mask = one._ops_f(dw).detach(), one._kinematic_param_mask(dw)
    eager = backend.fk_mse(dw.program, ops_f, dw.ops_i, q, want, 7, mask, True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = backend.fk_mse(dw.program, ops_f, dw.ops_i, q, want, 7, mask, True)
    for _ in range(3):
        for t in out:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, eager))
    assert int(prog._ticket.item()) == 0


@pytest.mark.gpu
def test_gpu_training_step_is_hipgraph_capturable():
    """Forward + loss + backward + Adam of the kinematics-learning loop captured ONCE into a hipGraph and replayed:
    no call on the path synchronises or touches host memory, so the launch-bound loop runs without the host."""
    torch.manual_seed(0)
    m = load_model("iiwa7", "cuda"); gt = load_model("iiwa7", "cuda")
    m.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(1, 3))
    m.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(1, 3))
    q = torch.from_numpy(sample_states(m, 4096, seed=2)[0]).cuda()
    with torch.no_grad():
        want, _ = gt.compute_forward_kinematics(q, "iiwa_link_ee")
    opt = torch.optim.Adam(m.parameters(), lr=5e-3, capturable=True)

    def train_step():
        pos, _ = m.compute_forward_kinematics(q, "iiwa_link_ee")
        loss = torch.nn.functional.mse_loss(pos, want)
        loss.backward()
        opt.step()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            train_step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss = train_step()
    graph.replay(); torch.cuda.synchronize()
    first = loss.item()
    before = [p.detach().clone() for p in m.parameters()]
    for _ in range(50):
        graph.replay()
    torch.cuda.synchronize()
    assert loss.item() < 0.6 * first
    assert all(not torch.equal(a, b) for a, b in zip(before, m.parameters()))


# ---------------------------------------------------------------------------------------------------------------
# Gradients of a loss on the geometric Jacobian (+ position): the JAC form of the same adjoint sweep (Jacobian-column
# gradients enter as pose adjoints of the chain links) against torch autograd through the reference's
# compute_endeffector_jacobian (tests/golden/golden_grad_jac.npz, made by tests/golden/make_golden_grad_jac.py).
# ---------------------------------------------------------------------------------------------------------------
JAC_CASES = ["iiwa7", "panda_no_gripper", "allegro_left"]


def load_golden_grad_jac():
    import os
    from helpers import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, "golden_grad_jac.npz"), allow_pickle=False)


def jac_learnable_model(g, case, device="cpu"):
    m = load_model(case, device)
    params = {}
    for key in g[case + "/keys"]:
        link, pname, tensor_name = str(key).split("/")
        init = torch.from_numpy(g["%s/init/%s" % (case, key)].copy())
        mod = UnconstrainedTensor(dim1=1, dim2=3, init_tensor=init)
        m.make_link_param_learnable(link, pname, mod)
        params[str(key)] = dict(mod.named_parameters())[tensor_name]
    return m, params


def jac_loss(lin, ang, pos, w_lin, w_ang, w_pos):
    return (w_lin * lin).sum() + (w_ang * ang).sum() + (lin ** 2).sum() + 0.5 * (ang ** 2 * w_ang).sum() + (w_pos * pos).sum()


@pytest.mark.parametrize("case", JAC_CASES)
def test_emu_jacobian_backward_vs_reference_autograd(emu, case):
    g = load_golden_grad_jac()
    m, params = jac_learnable_model(g, case)
    q = np.ascontiguousarray(g[case + "/q"])
    B, n = q.shape
    idx = m._name_to_idx_map[str(g[case + "/ee"])]
    prog = build_walk(m._spec, targets=[idx])
    table = m._link_table()
    ops_f_t = (table.reshape(-1)[torch.from_numpy(prog.gather.reshape(-1))]
               * torch.from_numpy(prog.gsign.reshape(-1))).reshape(prog.capacity, 32)
    ops_f = np.ascontiguousarray(ops_f_t.detach().numpy(), np.float32)
    walk, _keep = host_walk(m, prog)
    walk.ops_f = ops_f.ctypes.data
    pos, quat = np.zeros((B, 3), np.float32), np.zeros((B, 4), np.float32)
    lin, ang = np.zeros((B, 3, n), np.float32), np.zeros((B, 3, n), np.float32)
    assert emu.emu_fk_jacobian(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(pos), _ptr(quat), _ptr(lin), _ptr(ang)) == 0
    assert np.allclose(lin, g[case + "/lin"], atol=2e-6) and np.allclose(ang, g[case + "/ang"], atol=2e-6)
    w_lin, w_ang, w_pos = (g[case + "/" + k] for k in ("w_lin", "w_ang", "w_pos"))
    glin = np.ascontiguousarray(w_lin + 2.0 * lin, np.float32)
    gang = np.ascontiguousarray(w_ang + ang * w_ang, np.float32)
    gpos = np.ascontiguousarray(w_pos, np.float32)
    gq = np.full((B, n), np.nan, np.float32); gops = np.full((prog.capacity, 32), np.nan, np.float32)
    mask = m._kinematic_param_mask(type("W", (), {"program": prog})())
    assert emu.emu_fk_jacobian_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(gpos), _ptr(glin), _ptr(gang),
                                        ctypes.c_uint64(mask), _ptr(gq), _ptr(gops)) == 0
    assert close(gq, g[case + "/grad_q"]), np.abs(gq - g[case + "/grad_q"]).max()
    m.zero_grad()
    ops_f_t.backward(torch.from_numpy(gops))
    for key, p in params.items():
        ref = g["%s/grad/%s" % (case, key)]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(ref)
        assert np.abs(got - ref).max() <= GRAD_RTOL * max(np.abs(ref).max(), 1e-3), (case, key, got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", JAC_CASES)
def test_gpu_jacobian_backward_vs_reference_autograd(case):
    g = load_golden_grad_jac()
    m, params = jac_learnable_model(g, case, "cuda")
    ee = str(g[case + "/ee"])
    q = torch.from_numpy(g[case + "/q"].copy()).cuda().requires_grad_(True)
    w_lin, w_ang, w_pos = (torch.from_numpy(g[case + "/" + k].copy()).cuda() for k in ("w_lin", "w_ang", "w_pos"))
    lin, ang = m.compute_endeffector_jacobian(q, ee)
    pos, _ = m.compute_forward_kinematics(q, ee)
    loss = jac_loss(lin, ang, pos, w_lin, w_ang, w_pos)
    loss.backward()
    assert abs(loss.item() - float(g[case + "/loss"])) <= 1e-4 * max(1.0, abs(float(g[case + "/loss"])))
    assert close(q.grad.cpu().numpy(), g[case + "/grad_q"])
    for key, p in params.items():
        ref = g["%s/grad/%s" % (case, key)]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        assert np.abs(got - ref).max() <= GRAD_RTOL * max(np.abs(ref).max(), 1e-3), (case, key, got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 63, 64, 200])
def test_gpu_fused_fk_and_jacobian_backward_vs_emu(emu, B):
    """compute_fk_and_jacobian (pos + both Jacobians from one launch, one backward launch) on ragged batches."""
    m = load_model("panda_no_gripper", "cuda")
    mc = load_model("panda_no_gripper")
    q = sample_states(mc, B, seed=5)[0]
    rng = np.random.default_rng(1)
    gpos, glin, gang = (rng.standard_normal(s).astype(np.float32) for s in ((B, 3), (B, 3, 7), (B, 3, 7)))
    qt = torch.from_numpy(q).cuda().requires_grad_(True)
    pos, quat, lin, ang = m.compute_fk_and_jacobian(qt, "panda_virtual_ee_link")
    ((pos * torch.from_numpy(gpos).cuda()).sum() + (lin * torch.from_numpy(glin).cuda()).sum()
     + (ang * torch.from_numpy(gang).cuda()).sum()).backward()
    prog = build_walk(mc._spec, targets=[mc._name_to_idx_map["panda_virtual_ee_link"]])
    walk, _keep = host_walk(mc, prog)
    gq = np.full((B, 7), np.nan, np.float32)
    assert emu.emu_fk_jacobian_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(gpos), _ptr(glin), _ptr(gang),
                                        ctypes.c_uint64(0), _ptr(gq), None) == 0
    assert np.allclose(qt.grad.cpu().numpy(), gq, atol=2e-5, rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("robot", __import__("helpers").ALL_ROBOTS)
def test_gpu_jacobian_backward_every_robot_vs_emu(emu, robot):
    """The deepest chain of every shipped robot (all compiled capacities of the JAC kernels), ragged batch, constant
    gradients of two ops, against the host emulation of the same sweep."""
    from differentiable_robot_model_amd import backend
    mc = load_model(robot)
    m = load_model(robot, "cuda")
    depth = lambda i: 0 if i == 0 else 1 + depth(int(mc._spec.parent[i]))
    idx = max(range(len(mc._bodies)), key=depth)
    n, B = mc._n_dofs, 70
    prog = build_walk(mc._spec, targets=[idx])
    q = sample_states(mc, B, seed=11)[0]
    rng = np.random.default_rng(2)
    gpos, glin, gang = (rng.standard_normal(s).astype(np.float32) for s in ((B, 3), (B, 3, n), (B, 3, n)))
    mask = (1 << (prog.n_ops - 1)) | (1 << (prog.n_ops // 2))
    walk, _keep = host_walk(mc, prog)
    gq = np.full((B, n), np.nan, np.float32); gops = np.full((prog.capacity, 32), np.nan, np.float32)
    assert emu.emu_fk_jacobian_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(gpos), _ptr(glin), _ptr(gang),
                                        ctypes.c_uint64(mask), _ptr(gq), _ptr(gops)) == 0
    dw = m._get_walk(("chain", idx), targets=[idx])
    got_q, got_ops = backend.fk_jacobian_backward(dw.program, m._ops_f(dw), dw.ops_i, torch.from_numpy(q).cuda(),
                                                  torch.from_numpy(gpos).cuda(), torch.from_numpy(glin).cuda(),
                                                  torch.from_numpy(gang).cuda(), n, mask, True)
    assert dw.program.capacity == prog.capacity
    assert np.allclose(got_q.cpu().numpy(), gq, atol=3e-5, rtol=3e-5), (robot, np.abs(got_q.cpu().numpy() - gq).max())
    scale = max(np.abs(gops).max(), 1e-6)
    assert np.abs(got_ops.cpu().numpy() - gops).max() <= 1e-4 * scale, (robot, prog.capacity)


# ---------------------------------------------------------------------------------------------------------------
# Serial-chain specialisation (fk_backward_chain / fk_backward_arm_kernel<8, 7>): closed-form adjoints of one chain FK
# against the generic adjoint sweep, which the tests above pin to the reference's autograd.
# ---------------------------------------------------------------------------------------------------------------
ARM_ROBOTS = [("panda_no_gripper", "panda_virtual_ee_link"), ("iiwa7", "iiwa_link_ee"), ("fetch_arm_no_gripper", None)]


def _arm_walk(m, link):
    idx = m._name_to_idx_map[link] if link else len(m._bodies) - 1
    prog = build_walk(m._spec, targets=[idx])
    assert prog.shape & 1 and prog.capacity == 8
    return idx, prog


@pytest.mark.parametrize("robot,link", ARM_ROBOTS)
def test_emu_arm_chain_fk_backward_equals_generic_walk(emu, robot, link):
    m = load_model(robot)
    idx, prog = _arm_walk(m, link)
    B = 23
    q = sample_states(m, B, seed=9)[0]
    gpos = np.random.default_rng(4).standard_normal((B, 3)).astype(np.float32)
    walk, _keep = host_walk(m, prog)
    mask = 0b10100101 & ((1 << prog.n_ops) - 1)
    gq_a, gq_b = (np.full((B, 7), np.nan, np.float32) for _ in range(2))
    go_a, go_b = (np.full((8, 32), np.nan, np.float32) for _ in range(2))
    assert emu.emu_fk_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), 1, _ptr(gpos), ctypes.c_uint64(mask), _ptr(gq_a), _ptr(go_a)) == 0
    assert emu.emu_fk_backward_arm(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(gpos), ctypes.c_uint64(mask), _ptr(gq_b), _ptr(go_b)) == 0
    assert np.abs(gq_a - gq_b).max() <= 1e-5 * max(1.0, np.abs(gq_a).max())
    assert np.abs(go_a - go_b).max() <= 2e-5 * max(1.0, np.abs(go_a).max()), np.abs(go_a - go_b).max()


@pytest.mark.gpu
@pytest.mark.parametrize("robot,link", ARM_ROBOTS)
@pytest.mark.parametrize("B", [64, 200, 4096])
def test_gpu_arm_chain_fk_backward_vs_emu(emu, robot, link, B):
    """Full tiles through fk_backward_arm_kernel, the ragged tail through the generic kernel, one reduction over both."""
    from differentiable_robot_model_amd import backend
    mc, m = load_model(robot), load_model(robot, "cuda")
    idx, prog = _arm_walk(mc, link)
    q = sample_states(mc, B, seed=B)[0]
    gpos = np.random.default_rng(B).standard_normal((B, 3)).astype(np.float32)
    walk, _keep = host_walk(mc, prog)
    mask = 0b01001010 & ((1 << prog.n_ops) - 1)
    gq = np.full((B, 7), np.nan, np.float32); gops = np.full((8, 32), np.nan, np.float32)
    assert emu.emu_fk_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), 1, _ptr(gpos), ctypes.c_uint64(mask), _ptr(gq), _ptr(gops)) == 0
    dw = m._get_walk(("fk", (idx,)), targets=[idx])
    for want_q, pm in ((True, mask), (True, 0), (False, mask)):
        got_q, got_ops = backend.fk_backward(dw.program, m._ops_f(dw), dw.ops_i, torch.from_numpy(q).cuda(),
                                             torch.from_numpy(gpos).cuda().reshape(B, 1, 3), 1, 7, pm, want_q)
        if want_q:
            assert np.allclose(got_q.cpu().numpy(), gq, atol=2e-5, rtol=2e-5), np.abs(got_q.cpu().numpy() - gq).max()
        if pm:
            assert np.abs(got_ops.cpu().numpy() - gops).max() <= 1e-4 * max(np.abs(gops).max(), 1e-6) * max(1.0, B / 256)


@pytest.mark.gpu
def test_gpu_kinematic_trajectory_optimisation_lowers_the_cost():
    """examples/run_kinematic_trajectory_opt.py in miniature: an action sequence is optimised through
    compute_forward_kinematics (B = 1 calls, gradient with respect to q) so that the Panda's end effector approaches a
    goal position."""
    torch.manual_seed(0)
    m = load_model("panda_no_gripper", "cuda")
    link, horizon = "panda_virtual_ee_link", 8
    lim = m.get_joint_limits()
    lo = torch.tensor([j["lower"] for j in lim], device="cuda"); hi = torch.tensor([j["upper"] for j in lim], device="cuda")
    start = torch.tensor([0.0, 0.0, 0.0, -1.5, 0.0, 1.6, 0.0], device="cuda")
    with torch.no_grad():
        goal, _ = m.compute_forward_kinematics(torch.zeros(1, 7, device="cuda"), link)
    actions = torch.nn.Parameter(torch.zeros(horizon, 7, device="cuda"))
    opt = torch.optim.Adam([actions], lr=1e-2)

    def rollout():
        q, ee = start, []
        for t in range(horizon):
            q = torch.minimum(torch.maximum(q.detach() + actions[t], lo), hi)
            pos, _ = m.compute_forward_kinematics(q.reshape(1, 7), link)
            ee.append(pos.squeeze(0))
        return torch.stack(ee)

    costs = []
    for _ in range(40):
        opt.zero_grad()
        cost = ((100 * (rollout() - goal)) ** 2).mean()
        cost.backward()
        opt.step()
        costs.append(cost.item())
    assert actions.grad is not None and torch.isfinite(actions.grad).all()
    assert costs[-1] < 0.8 * costs[0], (costs[0], costs[-1])


# ---------------------------------------------------------------------------------------------------------------
# Gradients of losses that read the QUATERNION (tests/golden/golden_grad_quat.npz, made by
# tests/golden/make_golden_grad_quat.py from torch autograd through the unmodified reference): the reference assembles the
# quaternion from entries of R inside autograd (spatial_vector_algebra.py:108-136), so orientation losses have
# gradients; here dL/dquat becomes dL/dR on the host (autograd._quat_grad_to_rot) and enters the adjoint sweep.
# ---------------------------------------------------------------------------------------------------------------
QUAT_CASES = [("iiwa7", "quat"), ("iiwa7", "pose"), ("allegro_left", "quat"), ("allegro_left", "pose")]


def load_golden_grad_quat():
    import os
    from helpers import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, "golden_grad_quat.npz"), allow_pickle=False)


def learnable_model_quat(g, key, device="cpu"):
    m = load_model(key.split("/")[0], device)
    for link in g[key + "/learnable"]:
        for pname in ("trans", "rot_angles"):
            init = torch.from_numpy(g["%s/init/%s/%s" % (key, link, pname)].copy())
            m.make_link_param_learnable(str(link), pname, UnconstrainedTensor(dim1=1, dim2=3, init_tensor=init))
    return m


@pytest.mark.parametrize("case,mode", QUAT_CASES)
def test_emu_quaternion_backward_vs_reference_autograd(emu, case, mode):
    from differentiable_robot_model_amd.autograd import _quat_grad_to_rot
    g = load_golden_grad_quat()
    key = "%s/%s" % (case, mode)
    m = learnable_model_quat(g, key)
    targets = [str(t) for t in g[key + "/targets"]]
    idx = [m._name_to_idx_map[t] for t in targets]
    q = np.ascontiguousarray(g[key + "/q"])
    prog = build_walk(m._spec, targets=idx)
    table = m._link_table()
    ops_f_t = (table.reshape(-1)[torch.from_numpy(prog.gather.reshape(-1))]
               * torch.from_numpy(prog.gsign.reshape(-1))).reshape(prog.capacity, 32)
    ops_f = np.ascontiguousarray(ops_f_t.detach().numpy(), np.float32)
    walk, _keep = host_walk(m, prog)
    walk.ops_f = ops_f.ctypes.data
    B, T, n = q.shape[0], len(idx), m._n_dofs
    pos = np.zeros((B, T, 3), np.float32); quat = np.zeros((B, T, 4), np.float32)
    assert emu.emu_fk(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), T, _ptr(pos), _ptr(quat)) == 0
    wq = np.stack([g["%s/want_quat/%s" % (key, t)] for t in targets], 1)
    wp = np.stack([g["%s/want_pos/%s" % (key, t)] for t in targets], 1)
    for t_i, t in enumerate(targets):
        assert np.abs(quat[:, t_i] - g["%s/quat/%s" % (key, t)]).max() < 2e-6      # same sign as the reference's
    loss = sum(((quat[:, t] - wq[:, t]) ** 2).mean() for t in range(T))
    gquat = 2.0 * (quat - wq) / (B * 4)
    gpos = np.zeros((B, T, 3), np.float32)
    if mode == "pose":
        loss += sum(((pos[:, t] - wp[:, t]) ** 2).mean() for t in range(T))
        gpos = np.ascontiguousarray(2.0 * (pos - wp) / (B * 3), np.float32)
    assert abs(loss - float(g[key + "/loss"])) < 2e-6
    grot = np.ascontiguousarray(_quat_grad_to_rot(torch.from_numpy(quat), torch.from_numpy(gquat.astype(np.float32))).numpy(), np.float32)
    gq = np.full((B, n), np.nan, np.float32); gops = np.full((prog.capacity, 32), np.nan, np.float32)
    mask = m._kinematic_param_mask(type("W", (), {"program": prog})())
    assert emu.emu_fk_backward_rot(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), T, _ptr(gpos), _ptr(grot),
                                   ctypes.c_uint64(mask), _ptr(gq), _ptr(gops)) == 0
    m.zero_grad()
    ops_f_t.backward(torch.from_numpy(gops))
    assert close(gq, g[key + "/grad_q"]), np.abs(gq - g[key + "/grad_q"]).max()
    for link in g[key + "/learnable"]:
        body = m._bodies[m._name_to_idx_map[str(link)]]
        for pname in ("trans", "rot_angles"):
            got = getattr(body, pname).param.grad.numpy()
            ref = g["%s/grad/%s/%s" % (key, link, pname)]
            assert close(got, ref), (key, link, pname, got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case,mode", QUAT_CASES)
def test_gpu_quaternion_backward_vs_reference_autograd(case, mode):
    g = load_golden_grad_quat()
    key = "%s/%s" % (case, mode)
    m = learnable_model_quat(g, key, "cuda")
    targets = [str(t) for t in g[key + "/targets"]]
    q = torch.from_numpy(g[key + "/q"].copy()).cuda().requires_grad_(True)
    loss = 0.0
    for t in targets:
        pos, quat = m.compute_forward_kinematics(q, t)
        loss = loss + torch.nn.functional.mse_loss(quat, torch.from_numpy(g["%s/want_quat/%s" % (key, t)].copy()).cuda())
        if mode == "pose":
            loss = loss + torch.nn.functional.mse_loss(pos, torch.from_numpy(g["%s/want_pos/%s" % (key, t)].copy()).cuda())
    loss.backward()
    assert abs(loss.item() - float(g[key + "/loss"])) < 2e-6
    assert close(q.grad.cpu().numpy(), g[key + "/grad_q"])
    for link in g[key + "/learnable"]:
        body = m._bodies[m._name_to_idx_map[str(link)]]
        for pname in ("trans", "rot_angles"):
            assert close(getattr(body, pname).param.grad.cpu().numpy(), g["%s/grad/%s/%s" % (key, link, pname)]), (key, link, pname)
    # the fused FK + Jacobian call carries the same quaternion gradient
    q2 = torch.from_numpy(g[key + "/q"].copy()).cuda().requires_grad_(True)
    t = targets[0]
    _, quat2, _, _ = m.compute_fk_and_jacobian(q2, t)
    quat1 = m.compute_forward_kinematics(q, t)[1]
    w = torch.from_numpy(g["%s/want_quat/%s" % (key, t)].copy()).cuda()
    (g1,) = torch.autograd.grad(torch.nn.functional.mse_loss(quat1, w), q)
    (g2,) = torch.autograd.grad(torch.nn.functional.mse_loss(quat2, w), q2)
    assert close(g2.cpu().numpy(), g1.cpu().numpy(), 1e-4)


@pytest.mark.gpu
def test_gpu_config5_full_size_vs_reference_autograd():
    """BASELINE configuration 5 at its stated size against the REFERENCE (not the host emulation): iiwa7, iiwa_link_1.trans /
    .rot_angles learnable, batch 16 384, FK(EE position) MSE loss.  The golden holds the seed of q, the loss and the six
    parameter-gradient scalars torch autograd produced through the unmodified reference."""
    g = load_golden_grad_quat()
    B, seed = int(g["config5/batch"]), int(g["config5/seed"])
    m = load_model("iiwa7", "cuda")
    gt = load_model("iiwa7", "cuda")
    for pname in ("trans", "rot_angles"):
        init = torch.from_numpy(g["config5/init/" + pname].copy())
        m.make_link_param_learnable("iiwa_link_1", pname, UnconstrainedTensor(dim1=1, dim2=3, init_tensor=init))
    lim = m.get_joint_limits()
    lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
    qn = (lo + (hi - lo) * np.random.default_rng(seed).random((B, len(lim)))).astype(np.float32)
    assert abs(float(qn.astype(np.float64).sum()) - float(g["config5/q_checksum"])) < 1e-6      # the same batch
    q = torch.from_numpy(qn).cuda().requires_grad_(True)
    with torch.no_grad():
        want, _ = gt.compute_forward_kinematics(q.detach(), "iiwa_link_ee")
    pos, _ = m.compute_forward_kinematics(q, "iiwa_link_ee")
    loss = torch.nn.functional.mse_loss(pos, want)
    loss.backward()
    assert abs(loss.item() - float(g["config5/loss"])) < 1e-6 * max(1.0, float(g["config5/loss"]))
    assert np.abs(pos.detach().double().sum(0).cpu().numpy() - g["config5/pos_sum"]).max() < 2e-2   # 16 384 rows of ~1 m
    body = m._bodies[m._name_to_idx_map["iiwa_link_1"]]
    for pname in ("trans", "rot_angles"):
        assert close(getattr(body, pname).param.grad.cpu().numpy(), g["config5/grad/" + pname]), pname
    assert abs(float(q.grad.double().abs().sum()) - float(g["config5/grad_q_abs_sum"])) < 1e-3 * float(g["config5/grad_q_abs_sum"])


LINK_SETS = {
    "link1": [("iiwa_link_1", "trans"), ("iiwa_link_1", "rot_angles")],
    "rot_only": [("iiwa_link_3", "rot_angles")],
    "every_arm_link": [("iiwa_link_%d" % i, p) for i in range(1, 8) for p in ("trans", "rot_angles")],
    "with_dynamics": [("iiwa_link_2", "trans"), ("iiwa_link_5", "mass"), ("iiwa_link_5", "rot_angles"), ("iiwa_link_6", "com")],
}


def _learnable_pair(device, what, seed=3):
    """Two models with the same learnable parameters: fk_mse_loss through drm_fk_mse_links, and through the composition."""
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor as UT
    out = []
    for links_path in (True, False):
        m = load_model("iiwa7", device)
        torch.manual_seed(seed)
        for link, pname in LINK_SETS[what]:
            m.make_link_param_learnable(link, pname, UT(1, 1) if pname == "mass" else UT(1, 3))
        m._fk_mse_links = links_path
        m._table_links = False      # (the composition these tests hold drm_fk_mse_links to: cat + drm_walk_table + ... + drm_walk_table_backward;
        out.append(m)               #  ABI 13's drm_walk_table_links agrees with it to the last bits, tests/test_table_links.py)
    return out


def test_fk_mse_links_of_the_host_build_is_the_composition():
    """ABI 12, libdrm_cpu.so: drm_fk_mse_links (table from the links' parameter tensors, loss, gradients back to the parameters) against
    drm_walk_table -> drm_fk_mse -> drm_walk_table_backward called one by one."""
    from differentiable_robot_model_amd import backend
    for what in LINK_SETS:
        m = _learnable_pair("cpu", what)[0]
        ee = m._name_to_idx_map["iiwa_link_ee"]
        dw = m._get_walk(("fk", (ee,)), targets=[ee])
        links, base, sel = m._learnable_plan(dw)
        pieces = [p.detach() for p in m._learnable_pieces(links)]
        mask = m._kinematic_param_mask(dw)
        B = 192
        q = torch.from_numpy(sample_states(m, B, seed=2)[0])
        want = torch.randn(B, 3, generator=torch.Generator().manual_seed(1)) * 0.3
        loss, gq, gp = backend.fk_mse_links(dw.program, base, dw.ops_i, sel, dw.gsign, pieces, q, want, 7, mask, True)
        table = backend.WalkTable.apply(base, sel, dw.gsign, len(links), *pieces).reshape(dw.program.capacity, -1)
        loss2, gq2, gops = backend.fk_mse(dw.program, table, dw.ops_i, q, want, 7, mask, True)
        lib = backend.library_for(torch.device("cpu"))
        params = torch.cat([p.reshape(-1) for p in pieces]).contiguous()
        gp2 = torch.empty(len(links), 20)
        assert lib.drm_walk_table_backward(params.data_ptr(), len(links), gops.data_ptr(), sel.data_ptr(), dw.gsign.data_ptr(), gops.numel(),
                                           gp2.data_ptr(), None) == 0
        assert torch.equal(loss, loss2) and torch.equal(gq, gq2), what
        assert torch.equal(gp[:, :6], gp2[:, :6]) and not gp[:, 6:].any(), what
        assert float(gp[:, :6].abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("what", sorted(LINK_SETS))
@pytest.mark.parametrize("B", [64, 16384, 1030 * 64])
def test_gpu_fk_mse_links_is_the_composition_in_two_launches(what, B):
    """ABI 12 (drm_fk_mse_links): fk_mse_loss of a model with learnable links builds the walk table inside the chain kernel and takes
    the gradient back to the links' parameters inside the reduction kernel.  Loss, d loss / d q and every parameter's gradient BIT FOR
    BIT those of WalkTable -> drm_fk_mse -> WalkTable's backward (the same sums in the same order), for learnable sets of one link,
    rotation only, all seven arm links, and links that are learnable in their dynamics only; batch sizes on both sides of the
    register-table / LDS-table switch; with and without input gradients."""
    fused, composed = _learnable_pair("cuda", what)
    q = torch.from_numpy(sample_states(fused, B, seed=B)[0]).cuda()
    with torch.no_grad():
        want = load_model("iiwa7", "cuda").compute_forward_kinematics(q, "iiwa_link_ee")[0] + 0.02
    for want_q, scale in ((True, 1.0), (False, 1.0), (True, -1.7)):
        got = []
        for m in (fused, composed):
            m.zero_grad()
            x = q.clone().requires_grad_(want_q)
            loss = m.fk_mse_loss(x, "iiwa_link_ee", want)
            assert (loss.grad_fn.name().startswith("_FkMseLinks")) == (m is fused)
            (loss * scale).backward()
            got.append([loss.detach()] + ([x.grad] if want_q else []) + [p.grad for p in m.parameters()])
        assert len(got[0]) == len(got[1]) and len(got[0]) >= 2
        for k, (a, b) in enumerate(zip(*got)):
            assert (a is None) == (b is None)
            if a is None:
                continue
            if scale == 1.0:
                assert torch.equal(a, b), (what, B, want_q, k, float((a - b).abs().max()))
            else:       # (the incoming gradient multiplies the sums here, their terms there: the last bits differ)
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, (what, B, k)


@pytest.mark.gpu
def test_gpu_fk_mse_links_training_step_from_a_hip_graph(monkeypatch):
    """The training step of the reference's kinematics-learning loop (examples/learn_kinematics_of_iiwa.py:47-55) through fk_mse_loss's
    links form, captured into a hipGraph with fused Adam: the replayed steps follow the eager ones, and more learnable links than the
    kernel takes (nine: every link of the chain) fall back to the composition."""
    fused, composed = _learnable_pair("cuda", "link1")
    B = 16384
    q = torch.from_numpy(sample_states(fused, B, seed=4)[0]).cuda()
    with torch.no_grad():
        want = load_model("iiwa7", "cuda").compute_forward_kinematics(q, "iiwa_link_ee")[0]
    losses = []
    for m in (fused, composed):
        opt = torch.optim.Adam(m.parameters(), lr=1e-2, capturable=True, fused=True)

        def step():
            loss = m.fk_mse_loss(q, "iiwa_link_ee", want)
            loss.backward()
            opt.step()
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                step()
        torch.cuda.current_stream().wait_stream(side)
        opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = step()
        seen = []
        for _ in range(30):
            graph.replay()
            seen.append(float(loss.detach()))
        losses.append(seen)
    assert losses[0][-1] < 0.5 * losses[0][0]
    assert np.allclose(losses[0], losses[1], rtol=1e-5, atol=1e-9)
    # the kernel's maximum (eight learnable links: every link of the iiwa below its base) still goes through it, one more falls back
    from differentiable_robot_model_amd import backend
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor as UT
    pair = []
    for links_path in (True, False):
        many = load_model("iiwa7", "cuda")
        torch.manual_seed(5)
        for body in many._bodies[1:]:
            many.make_link_param_learnable(body.name, "trans", UT(1, 3))
        many._fk_mse_links = links_path
        pair.append(many)
    assert len({link for link, _ in pair[0]._learnable}) == backend.FK_MSE_MAX_LINKS == 8
    got = []
    for many in pair:
        loss = many.fk_mse_loss(q, "iiwa_link_ee", want)
        assert loss.grad_fn.name().startswith("_FkMseLinks") == (many is pair[0])
        loss.backward()
        got.append([loss.detach()] + [p.grad for p in many.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*got))
    monkeypatch.setattr(backend, "FK_MSE_MAX_LINKS", 7)
    loss = pair[0].fk_mse_loss(q, "iiwa_link_ee", want)
    assert loss.grad_fn.name().startswith("_FkMse") and not loss.grad_fn.name().startswith("_FkMseLinks")
    assert torch.equal(loss.detach(), got[0][0])
