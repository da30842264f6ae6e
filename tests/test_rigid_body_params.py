"""Inertia-matrix parametrisations (rigid_body_params.py): constructions, initialisation from a target matrix, and
equality with the reference's modules for the same parameter values (when /root/reference is present)."""
import numpy as np
import pytest
import torch

from differentiable_robot_model_amd.rigid_body_params import (CovParameterized3DInertiaMatrixNet, PositiveScalar,
                                                              Symm3DInertiaMatrixNet, SymmPosDef3DInertiaMatrixNet,
                                                              TriangParam3DInertiaMatrixNet)

try:
    import ref_import
    HAVE_REF = ref_import.reference_available()
except Exception:  # pragma: no cover
    HAVE_REF = False

TARGET = torch.tensor([[0.031, 0.002, -0.004], [0.002, 0.045, 0.003], [-0.004, 0.003, 0.052]])


@pytest.mark.parametrize("cls,kw", [(Symm3DInertiaMatrixNet, {}), (SymmPosDef3DInertiaMatrixNet, {"bias": 1e-7}),
                                    (CovParameterized3DInertiaMatrixNet, {"bias": 1e-7}),
                                    (TriangParam3DInertiaMatrixNet, {"bias": 1e-7})])
def test_initialised_from_a_matrix_reproduces_it(cls, kw):
    net = cls(init_param=TARGET.reshape(1, 3, 3), **kw)
    out = net()
    assert tuple(out.shape) == (3, 3)
    assert torch.allclose(out, TARGET, atol=2e-6)
    assert torch.allclose(out, out.t(), atol=1e-8)
    out.sum().backward()
    assert all(p.grad is not None for p in net.parameters())


@pytest.mark.parametrize("cls,kw", [(SymmPosDef3DInertiaMatrixNet, {"bias": 1e-5}),
                                    (CovParameterized3DInertiaMatrixNet, {"bias": 1e-5}),
                                    (TriangParam3DInertiaMatrixNet, {"bias": 1e-5})])
def test_random_parameters_give_admissible_inertias(cls, kw):
    torch.manual_seed(3)
    for _ in range(20):
        net = cls(init_param_std=0.3, **kw)
        ev = torch.linalg.eigvalsh(net().double())
        assert ev.min() > 0
        if cls is not SymmPosDef3DInertiaMatrixNet:   # principal moments obey the triangle inequalities
            assert ev[0] + ev[1] >= ev[2] * (1 - 1e-5)


def test_positive_scalar_starts_at_the_requested_value():
    p = PositiveScalar(min_val=0.5, init_param=torch.tensor(2.0))
    assert abs(p().item() - 2.0) < 1e-6


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present")
@pytest.mark.parametrize("name", ["Symm3DInertiaMatrixNet", "SymmPosDef3DInertiaMatrixNet",
                                  "CovParameterized3DInertiaMatrixNet"])
def test_same_parameters_same_matrix_as_the_reference(name):
    ref_import.import_reference()
    import differentiable_robot_model.rigid_body_params as ref_rbp
    import differentiable_robot_model_amd.rigid_body_params as my_rbp
    torch.manual_seed(1)
    kw = {} if name == "Symm3DInertiaMatrixNet" else {"bias": 1e-6}
    ref = getattr(ref_rbp, name)(init_param_std=0.2, **kw)
    mine = getattr(my_rbp, name)(**kw)
    with torch.no_grad():
        mine.l.copy_(ref.l)
    assert torch.allclose(mine(), ref(), atol=1e-7)
    # initialisation from a matrix stores the same six numbers
    ref2 = getattr(ref_rbp, name)(init_param=TARGET.reshape(1, 3, 3), **kw)
    mine2 = getattr(my_rbp, name)(init_param=TARGET.reshape(1, 3, 3), **kw)
    assert torch.allclose(mine2.l, ref2.l, atol=1e-6)


def test_batched_matrix_helpers_match_the_reference_layout():
    """SymmMatNet / CholeskyNet (reference rigid_body_params.py:59-132): diagonal first, then the strictly lower
    triangle in numpy.tril_indices order."""
    from differentiable_robot_model_amd.rigid_body_params import CholeskyNet, SymmMatNet
    rng = np.random.default_rng(0)
    for q in (1, 2, 3, 4):
        n = q * (q + 1) // 2
        l = torch.from_numpy(rng.standard_normal((5, n)).astype(np.float32))
        ii, jj = np.tril_indices(q, k=-1)
        want = np.zeros((5, q, q), np.float32)
        want[:, np.arange(q), np.arange(q)] = l[:, :q].numpy()
        want[:, ii, jj] = l[:, q:].numpy()
        L = CholeskyNet(q, 0.5).get_L(l)
        assert np.array_equal(L.numpy(), want)
        S = SymmMatNet(q)(l)
        assert np.array_equal(S.numpy(), want + np.transpose(np.tril(want, -1), (0, 2, 1)))
        spd, lb = CholeskyNet(q, 0.5).get_symm_pos_semi_def_matrix_and_l(l)
        assert np.allclose(lb[:, :q].numpy(), l[:, :q].numpy() + 0.5) and np.array_equal(lb[:, q:].numpy(), l[:, q:].numpy())
        Lb = want.copy(); Lb[:, np.arange(q), np.arange(q)] += 0.5
        assert np.allclose(spd.numpy(), Lb @ np.transpose(Lb, (0, 2, 1)), atol=1e-6)
