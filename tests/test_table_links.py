"""ABI 13: the walk table of a model with learnable links straight from the links' parameter tensors, in the form their modules store
them (drm_walk_table_links / drm_walk_table_links_backward, backend.WalkTableLinks) — against the composition it replaces: the
modules' own torch arithmetic (the reference's, rigid_body_params.py:26-43, 252-404), a cat of their outputs, drm_walk_table and
torch autograd back through the modules.  Tolerances: 2e-6 relative to the largest entry (fp32; the kernel contracts l * l + c into
one fused multiply-add, torch rounds twice)."""
import ctypes

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd import DifferentiableKUKAiiwa, DifferentiableFrankaPanda, backend
from differentiable_robot_model_amd.rigid_body_params import (CovParameterized3DInertiaMatrixNet, PositiveScalar, Symm3DInertiaMatrixNet,
                                                              SymmPosDef3DInertiaMatrixNet, TriangParam3DInertiaMatrixNet,
                                                              UnconstrainedScalar, UnconstrainedTensor)

DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]
INERTIA = {
    "plain": lambda: UnconstrainedTensor(3, 3),
    "symm": lambda: Symm3DInertiaMatrixNet(init_param_std=0.1),
    "spd": lambda: SymmPosDef3DInertiaMatrixNet(bias=1e-3, init_param_std=0.3),
    "cov": lambda: CovParameterized3DInertiaMatrixNet(bias=1e-3, init_param_std=0.3),
    "triang": lambda: TriangParam3DInertiaMatrixNet(bias=1e-3),      # (a module the kernels do not know: its output is the piece)
}
RTOL = 2e-6      # of the largest entry; parameter gradients (sums with cancellation over the batch, fed by a mass that differs in its last bit): 5e-5


def learnable_iiwa(device, inertia, table_links, seed=0):
    torch.manual_seed(seed)
    m = DifferentiableKUKAiiwa(device=device)
    m._table_links = table_links
    for k in range(1, 8):
        link = "iiwa_link_%d" % k
        m.make_link_param_learnable(link, "mass", PositiveScalar(min_val=0.01))
        m.make_link_param_learnable(link, "com", UnconstrainedTensor(1, 3))
        m.make_link_param_learnable(link, "inertia_mat", INERTIA[inertia]())
        if k % 2:
            m.make_link_param_learnable(link, "trans", UnconstrainedTensor(1, 3))
        if k == 3:
            m.make_link_param_learnable(link, "joint_damping", PositiveScalar())
        if k == 4:
            m.make_link_param_learnable(link, "rot_angles", UnconstrainedTensor(1, 3))
        if k == 5:
            m.make_link_param_learnable(link, "joint_damping", UnconstrainedScalar())
    return m


def states(device, B=192, seed=1):
    g = torch.Generator().manual_seed(seed)
    return tuple((torch.rand(B, 7, generator=g) - 0.5).to(device) for _ in range(3))


def close(a, b, rtol=RTOL, floor=0.0):
    """max |a - b| <= rtol * max |b| (+ floor: for the gradient of ONE parameter tensor — a scalar that is a sum with cancellation over
    the batch — the noise scales with the largest gradient of the model, not with the scalar)."""
    scale = max(float(b.abs().max()), 1e-6)
    return float((a - b).abs().max()) <= rtol * scale + floor


def grad_floor(grads, rel=2e-6):
    return rel * max(float(g.abs().max()) for g in grads if g is not None)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("inertia", list(INERTIA))
def test_torques_and_parameter_gradients_equal_the_composition(device, inertia):
    got = []
    for table_links in (False, True):
        m = learnable_iiwa(device, inertia, table_links)
        q, qd, qdd = states(device)
        tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
        (tau.pow(2).mean() + m.compute_forward_kinematics(q, "iiwa_link_ee")[0].pow(2).mean()).backward()
        got.append((tau.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}))
    (tau0, g0), (tau1, g1) = got
    assert close(tau1, tau0)
    assert g0.keys() == g1.keys() and len(g0) >= 26
    floor = grad_floor(g0.values())
    for name in g0:
        assert g1[name].shape == g0[name].shape and close(g1[name], g0[name], 5e-5, floor), name


@pytest.mark.parametrize("device", DEVICES)
def test_one_autograd_node_and_no_module_kernels(device):
    """The table of the learnable model hangs on the raw parameters through ONE node: the modules' forward() is not called for the
    forms the kernel knows."""
    m = learnable_iiwa(device, "spd", True)
    m._ops_f(m._dynamics_walk())      # (the snapshot of the constant rows evaluates every module once)
    called = []
    for mod in m.modules():
        if type(mod) in (PositiveScalar, SymmPosDef3DInertiaMatrixNet):
            mod.register_forward_hook(lambda *a: called.append(1))
    dw = m._dynamics_walk()
    table = m._ops_f(dw)
    assert table.grad_fn.name().startswith("WalkTableLinks") or table.grad_fn.next_functions[0][0].name().startswith("WalkTableLinks")
    assert not called
    plan, sources = m._learnable_sources(sorted({l for l, _ in m._learnable}))
    forms = [plan.entries[l][j][0] for l in range(plan.n_links) for j in range(6)]
    assert forms.count(backend.FORM_SQUARE_PLUS) == 8 and forms.count(backend.FORM_SPD) == 7
    assert len(sources) == len(plan.live) == len(list(m.parameters())) and all(s.numel() == n for s, n in zip(sources, plan.sizes))
    assert all(any(s is p for p in m.parameters()) for s in sources)       # (the parameters themselves, where they lie)
    assert sum(t is not None for t in plan.fixed) == 6 * 7 - len(sources)


@pytest.mark.parametrize("device", DEVICES)
def test_frozen_and_shared_parameters(device):
    m0, m1 = learnable_iiwa(device, "cov", False), learnable_iiwa(device, "cov", True)
    for m in (m0, m1):
        m.freeze_learnable_link_param("iiwa_link_2", "mass")
        m.freeze_learnable_link_param("iiwa_link_6", "inertia_mat")
        q, qd, qdd = states(device, 64)
        m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True).pow(2).sum().backward()
    floor = grad_floor([p.grad for p in m0.parameters()])
    for (n0, p0), (n1, p1) in zip(m0.named_parameters(), m1.named_parameters()):
        assert n0 == n1 and (p0.grad is None) == (p1.grad is None), n0
        if p0.grad is not None:
            assert close(p1.grad, p0.grad, 5e-5, floor), n0
    assert sum(p.grad is None for p in m1.parameters()) == 2


@pytest.mark.parametrize("device", DEVICES)
def test_second_derivatives_go_through_the_modules(device):
    """create_graph=True: the backward of the table is the torch twin (the modules' arithmetic on the raw tensors), differentiable
    again — Hessian-vector products equal the composition's."""
    hv = []
    for table_links in (False, True):
        m = learnable_iiwa(device, "spd", table_links)
        q, qd, qdd = states(device, 64)
        params = [p for p in m.parameters()]
        loss = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True).pow(2).mean()
        grads = torch.autograd.grad(loss, params, create_graph=True)
        dot = sum((g * torch.ones_like(g)).sum() for g in grads)
        hv.append(torch.autograd.grad(dot, params, allow_unused=True))
    floor = grad_floor(hv[0], 2e-5)
    for a, b in zip(*hv):
        assert (a is None) == (b is None)
        if a is not None:
            assert close(b, a, 2e-4, floor)


def test_abi_errors_need_no_gpu(cpu_library):
    lib = cpu_library
    links = (backend.DrmLinkPieces * 1)()
    forms = (backend.DrmLinkForms * 1)()
    buf = np.zeros(64, np.float32)
    sel = np.full(32, -1, np.int32)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.drm_walk_table_links(links, forms, 1, ptr(buf), ptr(sel), ptr(buf), 32, ptr(buf), None) != 0      # NULL pieces
    assert b"NULL" in lib.drm_last_error()
    for name in backend.PIECE_NAMES:
        setattr(links[0], name, buf.ctypes.data)
    assert lib.drm_walk_table_links(links, forms, 1, ptr(buf), ptr(sel), ptr(buf), 32, ptr(buf), None) == 0
    forms[0].inertia_mat = 9
    assert lib.drm_walk_table_links(links, forms, 1, ptr(buf), ptr(sel), ptr(buf), 32, ptr(buf), None) != 0
    assert b"form" in lib.drm_last_error()
    forms[0].inertia_mat = backend.FORM_SQUARE_PLUS      # a scalar form on the matrix
    assert lib.drm_walk_table_links_backward(links, forms, 1, ptr(buf), ptr(sel), ptr(buf), 32, ptr(buf), None) != 0
    assert lib.drm_walk_table_links(links, None, 0, ptr(buf), ptr(sel), ptr(buf), 32, ptr(buf), None) != 0
    assert lib.drm_walk_table_links(links, None, 33, ptr(buf), ptr(sel), ptr(buf), 32, ptr(buf), None) != 0


@pytest.mark.parametrize("device", DEVICES)
def test_raw_gradient_layout_against_torch(device):
    """drm_walk_table_links_backward by itself: a random cotangent on the table, every form on one link — against torch autograd
    through the modules and link_rows_torch."""
    torch.manual_seed(3)
    lib = backend.library_for(torch.device(device))
    mods = [PositiveScalar(min_val=0.2), SymmPosDef3DInertiaMatrixNet(bias=1e-2, init_param_std=0.5), PositiveScalar(),
            CovParameterized3DInertiaMatrixNet(bias=1e-2, init_param_std=0.5), Symm3DInertiaMatrixNet(init_param_std=0.5)]
    mods = [m.to(device) for m in mods]
    entries, sources = [], []
    P, F = backend.FORM_PLAIN, backend
    rnd = lambda *s: torch.randn(*s, device=device).requires_grad_(True)
    for mass, inertia, form in ((mods[0], mods[1], F.FORM_SPD), (mods[2], mods[3], F.FORM_COV), (None, mods[4], F.FORM_SYMM)):
        link = [(P, 0.0, None), (P, 0.0, None)]
        srcs = [rnd(1, 3), rnd(1, 3)]
        if mass is None:
            link.append((P, 0.0, None)); srcs.append(rnd(1))
        else:
            link.append((F.FORM_SQUARE_PLUS, mass._min_val, mass)); srcs.append(mass.l)
        link.append((P, 0.0, None)); srcs.append(rnd(1, 3))
        bias = getattr(inertia, "spd_3d_inertia_mat_diag_bias", getattr(inertia, "spd_3d_cov_inertia_mat_diag_bias", 0.0))
        link.append((form, bias, inertia)); srcs.append(inertia.l)
        link.append((P, 0.0, None)); srcs.append(rnd(1))
        entries.append(link); sources += srcs
    plan = backend.LinkSourcePlan(entries)
    n_entries = 8 * 32
    base = torch.randn(n_entries, device=device)
    sel = torch.full((n_entries,), -1, dtype=torch.int32)
    sel[32:64] = torch.arange(32, dtype=torch.int32)
    sel[96:128] = torch.arange(32, 64, dtype=torch.int32)
    sel[160:192] = torch.arange(64, 96, dtype=torch.int32)
    sel[200:232] = torch.arange(32, dtype=torch.int32)          # (a link that two ops of the walk read)
    sel = sel.to(device)
    gsign = torch.where(torch.rand(n_entries, device=device) < 0.5, -1.0, 1.0)
    table = backend.WalkTableLinks.apply(base, sel, gsign, plan, *sources)
    cot = torch.randn(n_entries, device=device)
    got = torch.autograd.grad(table, sources, cot)
    pieces = plan.torch_pieces(sources)
    packed = torch.cat([p.reshape(-1) for p in pieces]).reshape(3, 20)
    rows = backend.link_rows_torch(packed).reshape(-1)
    want_table = torch.where(sel >= 0, rows[sel.clamp_min(0).long()] * gsign, base)
    want = torch.autograd.grad(want_table, sources, cot)
    assert close(table.detach(), want_table.detach())
    for g, w, s in zip(got, want, sources):
        assert g.shape == s.shape and close(g, w, 5e-6)


@pytest.mark.gpu
def test_learn_dynamics_step_under_a_hip_graph_has_no_module_kernels():
    """The reference's learn-dynamics step (examples/learn_dynamics_iiwa.py:49-96: PositiveScalar masses, free centres of mass and
    inertia matrices, Adam) captured into a hipGraph: replays equal the eager steps, and the parameters move."""
    def run(graph):
        torch.manual_seed(0)
        truth = DifferentiableKUKAiiwa(device="cuda")
        m = learnable_iiwa("cuda", "plain", True)
        q, qd, qdd = states("cuda", 256)
        with torch.no_grad():
            target = truth.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
        opt = torch.optim.Adam(m.parameters(), lr=1e-2, capturable=True)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.mse_loss(m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True), target)
            loss.backward()
            opt.step()
            return loss
        losses = []
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    losses.append(float(step()))
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = step()
            for _ in range(7):
                g.replay()
                losses.append(float(out))
        else:
            for _ in range(10):
                losses.append(float(step()))
        return losses
    eager, graphed = run(False), run(True)
    assert eager[-1] < eager[0]
    np.testing.assert_allclose(graphed, eager, rtol=1e-4)


@pytest.mark.gpu
def test_panda_learnable_arm_kernel_reads_the_links_table():
    """The arm's own kernel for a set of learnable blocks (specialize.attach_arm_param) takes the table WalkTableLinks built."""
    got = []
    for table_links in (False, True):
        torch.manual_seed(0)
        m = DifferentiableFrankaPanda(device="cuda")
        m._table_links = table_links
        m.make_link_param_learnable("panda_link4", "mass", PositiveScalar(min_val=0.01))
        m.make_link_param_learnable("panda_link4", "inertia_mat", SymmPosDef3DInertiaMatrixNet(bias=1e-3, init_param_std=0.3))
        q, qd, qdd = states("cuda", 65536)
        qg = q.clone().requires_grad_(True)
        m.compute_inverse_dynamics(qg, qd, qdd, include_gravity=True).pow(2).mean().backward()
        got.append([qg.grad] + [p.grad for p in m.parameters()])
    for a, b in zip(*got):
        assert close(b, a, 2e-5)


@pytest.mark.parametrize("device", DEVICES)
def test_deepcopy_gives_an_independent_model(device):
    """copy.deepcopy(model), as users of the reference's plain nn.Module do (a ground-truth copy, a target network): parameters and
    constants are copied, everything derived (walks, launch structs, prepared calls, the source plan of the learnable links) is
    rebuilt by the copy; the two models then learn independently and a body's pose asks its own model."""
    import copy
    m = learnable_iiwa(device, "spd", True)
    q, qd, qdd = states(device, 64)
    tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True).detach()
    pos = m.compute_forward_kinematics(q, "iiwa_link_ee")[0].detach()
    twin = copy.deepcopy(m)
    assert [n for n, _ in twin.named_parameters()] == [n for n, _ in m.named_parameters()]
    assert all(a is not b and torch.equal(a, b) for a, b in zip(twin.parameters(), m.parameters()))
    assert torch.equal(twin.compute_inverse_dynamics(q, qd, qdd, include_gravity=True).detach(), tau)
    with torch.no_grad():
        for p in twin.parameters():
            p.add_(0.05)
    assert torch.equal(m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True).detach(), tau)       # (the original is untouched)
    assert torch.equal(m.compute_forward_kinematics(q, "iiwa_link_ee")[0].detach(), pos)
    moved = twin.compute_inverse_dynamics(q, qd, qdd, include_gravity=True)
    assert not torch.equal(moved.detach(), tau)
    moved.pow(2).mean().backward()
    assert all(p.grad is not None for p in twin.parameters()) and all(p.grad is None for p in m.parameters())
    twin.update_kinematic_state(q, qd)
    m.update_kinematic_state(q, qd)
    assert not torch.equal(twin._bodies[-1].pose.translation(), m._bodies[-1].pose.translation())
    plain = DifferentiableKUKAiiwa(device=device)       # a constant model: prepared calls and walks are the copy's own
    want = plain.compute_endeffector_jacobian(q, "iiwa_link_ee")
    got = copy.deepcopy(plain).compute_endeffector_jacobian(q, "iiwa_link_ee")
    assert all(torch.equal(a, b) for a, b in zip(got, want))


FORM_CASES = ["iiwa7_spd", "iiwa7_cov", "panda_symm", "allegro_mixed"]
FORM_ROBOT = {"iiwa7_spd": "iiwa7", "iiwa7_cov": "iiwa7", "panda_symm": "panda_no_gripper", "allegro_mixed": "allegro_left"}


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("case", FORM_CASES)
def test_forms_against_the_references_own_modules(case, device):
    """tests/golden/golden_forms.npz (make_golden_forms.py): the UNMODIFIED reference with ITS parameter modules in the loop —
    PositiveScalar masses / dampings, symmetric / SPD / covariance inertia matrices, free tensors — torques, end-link positions and the
    gradient of a random linear functional of both with respect to every RAW parameter and the joint state, from torch autograd through
    the reference.  Here: the same raw values in this package's modules, the table built and differentiated inside the kernels (ABI 13).
    Tolerances as for the other reference-autograd fixtures: 2e-5 (outputs), 2e-4 (gradients), of the largest entry."""
    import os
    from helpers import load_model
    from differentiable_robot_model_amd import rigid_body_params as rbp
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_forms.npz"))
    m = load_model(FORM_ROBOT[case], device)
    assert m._table_links
    raws = []
    for key, kind, const in zip(G[case + "/keys"], G[case + "/kinds"], G[case + "/consts"]):
        lname, pname = str(key).split("/")
        raw = torch.from_numpy(G["%s/raw/%s" % (case, key)])
        if kind == "PositiveScalar":
            module = rbp.PositiveScalar(min_val=float(const))
        elif kind == "UnconstrainedTensor":
            module = rbp.UnconstrainedTensor(dim1=raw.shape[0], dim2=raw.shape[1])
        else:
            module = getattr(rbp, str(kind))(bias=float(const)) if kind != "Symm3DInertiaMatrixNet" else rbp.Symm3DInertiaMatrixNet()
        (param,) = list(module.parameters())
        with torch.no_grad():
            param.copy_(raw.reshape(param.shape))
        m.make_link_param_learnable(lname, pname, module)
        (param,) = list(module.parameters())        # (moved to the model's device)
        raws.append(param)
        holder = m._bodies[m._name_to_idx_map[lname]]
        holder = holder if pname in ("trans", "rot_angles", "joint_damping") else holder.inertia
        value = getattr(holder, pname)().detach().cpu().numpy()
        want = G["%s/value/%s" % (case, key)]
        assert np.abs(value.reshape(-1) - want.reshape(-1)).max() <= 1e-6 * max(1.0, np.abs(want).max()), key      # (the module's own forward)
    plan, _ = m._learnable_sources(m._learnable_link_list())
    kernel_forms = sum(e[0] != backend.FORM_PLAIN for link in plan.entries for e in link)
    assert kernel_forms == sum(str(k) != "UnconstrainedTensor" for k in G[case + "/kinds"])       # (every known module is a kernel form)
    t = lambda tag, grad=False: torch.from_numpy(G["%s/%s" % (case, tag)]).to(device).requires_grad_(grad)
    q, qd, qdd = t("q", True), t("qd", True), t("qdd", True)
    tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    pos, _ = m.compute_forward_kinematics(q, str(G[case + "/link"]))
    rel = lambda got, want: float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-6)
    assert rel(tau.detach().cpu().numpy(), G[case + "/tau"]) <= 2e-5 and rel(pos.detach().cpu().numpy(), G[case + "/pos"]) <= 2e-5
    L = (t("w_tau") * tau).sum() + (t("w_pos") * pos).sum()
    grads = torch.autograd.grad(L, raws + [q, qd, qdd], allow_unused=True)
    wants = [G["%s/grad/%s" % (case, key)] for key in G[case + "/keys"]] + [G[case + "/gq"], G[case + "/gqd"], G[case + "/gqdd"]]
    floor = 2e-5 * max(float(np.abs(w).max()) for w in wants[:len(raws)])
    for key, g, want in zip(list(G[case + "/keys"]) + ["q", "qd", "qdd"], grads, wants):
        got = np.zeros_like(want) if g is None else g.detach().cpu().numpy().reshape(want.shape)
        assert float(np.abs(got - want).max()) <= 2e-4 * float(np.abs(want).max()) + floor, (case, key)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("robot", ["panda_no_gripper", "allegro_left", "fetch"])
def test_repeat_all_links_calls_take_the_prepared_call(device, robot, hostcall_module):
    """compute_forward_kinematics_all_links of a constant model, second call on: ONE C++ call over drm_fk_links' link-major outputs
    (FastCall.links) — the same dictionary, bit for bit, as the first call through the Python path; another batch size, a 1-D q
    (Python path), gradients wanted (Python path) and a parameter made learnable afterwards (the prepared call of the constant model
    is dropped; the learned model gets its own, which serves the calls that build no graph)."""
    from helpers import load_model
    m = load_model(robot, device)
    n = m._n_dofs
    g = torch.Generator().manual_seed(5)
    q = (torch.rand(130, n, generator=g) - 0.5).to(device)
    first = m.compute_forward_kinematics_all_links(q)
    assert m.__dict__.get("_fast_links") is not None
    again = m.compute_forward_kinematics_all_links(q)
    assert list(first) == list(again) == [b.name for b in m._bodies]
    for name in first:
        assert torch.equal(first[name][0], again[name][0]) and torch.equal(first[name][1], again[name][1]), name
        assert again[name][0].shape == (130, 3) and again[name][1].shape == (130, 4)
    other = m.compute_forward_kinematics_all_links(q[:7])
    assert all(torch.equal(other[k][0], first[k][0][:7]) for k in first)
    one = m.compute_forward_kinematics_all_links(q[3])
    assert all(torch.equal(one[k][0][0], first[k][0][3]) for k in first)
    x = q.clone().requires_grad_(True)
    diff = m.compute_forward_kinematics_all_links(x)
    last = list(diff)[-1]
    assert diff[last][0].requires_grad and torch.allclose(diff[last][0].detach(), first[last][0], atol=1e-6)
    link = m._bodies[-1].name
    m.make_link_param_learnable(link, "trans", UnconstrainedTensor(1, 3))
    assert m.__dict__.get("_fast_links") is None
    moved = m.compute_forward_kinematics_all_links(q)
    assert not torch.equal(moved[link][0], first[link][0]) and moved[link][0].requires_grad       # (a graph: the Python path)
    with torch.no_grad():       # the learned model where no graph is built: the prepared call with the table rebuilt in front of it
        a = m.compute_forward_kinematics_all_links(q)
        b = m.compute_forward_kinematics_all_links(q)
    assert all(torch.equal(a[k][0], b[k][0]) and torch.equal(a[k][0], moved[k][0].detach()) and not b[k][0].requires_grad for k in a)


@pytest.mark.parametrize("device", DEVICES)
def test_a_learned_model_without_a_graph_takes_the_prepared_call(device, hostcall_module):
    """A model WITH learnable parameters called where no autograd graph is built (torch.no_grad(): a learned model in a control loop):
    repeat calls go through the prepared C++ call, which rebuilds the walk table from the parameter tensors in front of every launch
    (FastCall.set_table -> drm_walk_table_links).  Same bits as the Python path; parameters changed in place, REPLACED, or turned into
    something the kernels cannot read as it lies are seen on the next call; with gradients wanted the call builds its graph as before;
    a training step in between does not retire the prepared call."""
    m = learnable_iiwa(device, "spd", True)
    ee = "iiwa_link_ee"
    q, qd, qdd = states(device, 70)
    calls = {"id": lambda: m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True),
             "nle": lambda: m.compute_non_linear_effects(q, qd),
             "fk": lambda: torch.cat(m.compute_forward_kinematics(q, ee), dim=1),
             "jac": lambda: torch.cat(m.compute_endeffector_jacobian(q, ee), dim=1),
             "H": lambda: m.compute_lagrangian_inertia_matrix(q),
             "fd": lambda: m.compute_forward_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)}

    def python_path(name):
        saved = (m._fast_id, m._fast_crba, m._fast_fd, dict(m._fast_fk), dict(m._fast_jac))
        m._fast_id = m._fast_crba = m._fast_fd = None; m._fast_fk.clear(); m._fast_jac.clear()
        import differentiable_robot_model_amd.robot_model as rmod
        keep, rmod.DifferentiableRobotModel._fast_entry = rmod.DifferentiableRobotModel._fast_entry, lambda *a, **k: None
        try:
            return calls[name]()
        finally:
            rmod.DifferentiableRobotModel._fast_entry = keep
            m._fast_id, m._fast_crba, m._fast_fd = saved[:3]; m._fast_fk.update(saved[3]); m._fast_jac.update(saved[4])

    with torch.no_grad():
        first = {k: f() for k, f in calls.items()}
        assert m._fast_id is not None and m._fast_crba is not None and m._fast_fd is not None and ee in m._fast_fk and ee in m._fast_jac
        for k, f in calls.items():
            assert torch.equal(f(), first[k]) and torch.equal(python_path(k), first[k]), k
        for p in m.parameters():        # an optimizer step: in place
            p.add_(0.02)
        moved = {k: f() for k, f in calls.items()}
        for k in calls:
            assert not torch.equal(moved[k], first[k]) and torch.equal(python_path(k), moved[k]), k
        # a parameter REPLACED by a new object (the module's dictionary is read on every call) ...
        mass = m._bodies[3].inertia.mass
        mass.l = torch.nn.Parameter(mass.l.detach() * 1.5)
        again = calls["id"]()
        assert not torch.equal(again, moved["id"]) and torch.equal(python_path("id"), again)
        # ... and one the kernels cannot read as it lies (float64): the Python path serves the call
        com = m._bodies[5].inertia.com
        com.param.data = com.param.data.double()
        assert torch.allclose(calls["id"](), python_path("id"), atol=1e-6)
        com.param.data = com.param.data.float()
        assert torch.equal(calls["id"](), python_path("id"))
    entry = m._fast_id
    tau = calls["id"]()                 # gradients wanted: a graph, as before
    assert tau.requires_grad
    tau.pow(2).mean().backward()
    assert all(p.grad is not None for p in m.parameters())
    with torch.no_grad():
        assert torch.equal(calls["id"](), tau.detach()) and m._fast_id is entry       # (the training step did not retire the prepared call)
    # a module the kernels do not know: its output is a plain piece, no prepared call
    other = learnable_iiwa(device, "triang", True)
    with torch.no_grad():
        a = other.compute_inverse_dynamics(q, qd, qdd)
        assert other._fast_id is None and torch.equal(other.compute_inverse_dynamics(q, qd, qdd), a)
