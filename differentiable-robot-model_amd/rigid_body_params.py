"""Learnable parametrisations that FEED the kernels' constant table.

The reference ships a family of nn.Modules for learnable link parameters
(reference ``differentiable_robot_model/rigid_body_params.py``).  They run once per
step on O(1) data upstream of the hot path, so they stay plain torch; any
``nn.Module`` whose ``forward()`` returns a tensor of the parameter's shape works
with ``make_link_param_learnable`` (the reference's own modules included).  The
three generic ones used by the reference's kinematics examples are provided here
under the same names and constructor arguments.
"""
import math

import torch


class UnconstrainedScalar(torch.nn.Module):
    """A free scalar; forward() -> the parameter as it was given ([1] by default, the shape of ``init_val`` otherwise, as
    in the reference, rigid_body_params.py:14-23: state_dicts carry over)."""

    def __init__(self, init_val=None):
        super().__init__()
        value = torch.rand(1) if init_val is None else torch.as_tensor(init_val, dtype=torch.float32)
        if value.numel() != 1:
            raise ValueError("UnconstrainedScalar takes one value, got shape %s" % (tuple(value.shape),))
        self.param = torch.nn.Parameter(value.clone())

    def forward(self):
        return self.param


class PositiveScalar(torch.nn.Module):
    """A scalar kept >= min_val by construction: min_val + l^2.  (reference rigid_body_params.py:26-43)"""

    def __init__(self, min_val=0.0, init_param_std=1.0, init_param=None):
        super().__init__()
        self._min_val = float(min_val)
        # `l` is 0-dim and forward() returns a 0-dim value, as in the reference (rigid_body_params.py:27-43): its
        # state_dicts (`...mass.l`) load here unchanged
        if init_param is None:
            value = torch.empty(1).normal_(mean=0.0, std=init_param_std).squeeze()
        else:  # init_param is the VALUE to start from, as in the reference: l = sqrt(value - min_val)
            value = torch.sqrt(torch.as_tensor(init_param, dtype=torch.float32).reshape(()) - self._min_val)
        self.l = torch.nn.Parameter(value.clone())

    def forward(self):
        return self.l * self.l + self._min_val


class UnconstrainedTensor(torch.nn.Module):
    """A free [dim1, dim2] tensor, N(0, init_std^2) initialised.  (reference rigid_body_params.py:46-56)"""

    def __init__(self, dim1, dim2, init_tensor=None, init_std=0.1):
        super().__init__()
        if init_tensor is None:
            value = torch.empty(dim1, dim2).normal_(mean=0.0, std=init_std)
        else:
            value = torch.as_tensor(init_tensor, dtype=torch.float32).reshape(dim1, dim2)
        self.param = torch.nn.Parameter(value.clone())

    def forward(self):
        return self.param


# ---------------------------------------------------------------------------------------------------------
# 3x3 inertia-matrix parametrisations (reference rigid_body_params.py:58-420).  Same class names, constructor
# arguments and parameter layout as the reference, so checkpoints and call sites carry over: the six free numbers
# live in ``self.l`` = [three diagonal entries, then the strictly-lower entries (1,0), (2,0), (2,1)].
# They run upstream of the kernels (once per step, 3x3), so they are plain torch.
# ---------------------------------------------------------------------------------------------------------
_DIAG = ([0, 1, 2], [0, 1, 2])
_LOWER = ([1, 2, 2], [0, 0, 1])


def _pack_diag_lower(mat: torch.Tensor) -> torch.Tensor:
    mat = torch.as_tensor(mat, dtype=torch.float32).reshape(3, 3)
    return torch.cat([mat[_DIAG], mat[_LOWER]])


def _placement(pairs) -> torch.Tensor:
    """[9, 6] 0/1 matrix M with (M @ l).reshape(3, 3)[r, c] = l[k] for every (r, c, k) in ``pairs``: the six free
    numbers are placed by ONE matrix-vector product with a constant that lives on the module's device (a registered
    buffer) — no host-side index tensors, so the parametrisations can be captured into a HIP graph."""
    M = torch.zeros(9, 6)
    for r, c, k in pairs:
        M[3 * r + c, k] = 1.0
    return M


_LOWER_PLACEMENT = [(0, 0, 0), (1, 1, 1), (2, 2, 2), (1, 0, 3), (2, 0, 4), (2, 1, 5)]
_SYMM_PLACEMENT = _LOWER_PLACEMENT + [(0, 1, 3), (0, 2, 4), (1, 2, 5)]


def _placement_q(qdim: int, symmetric: bool) -> torch.Tensor:
    """[q*q, q(q+1)/2] placement matrix of the batched helpers below: the first q numbers on the diagonal, the rest on the
    strictly lower triangle in row-major order (numpy.tril_indices(q, -1): (1,0), (2,0), (2,1), ...), mirrored if asked."""
    n = qdim * (qdim + 1) // 2
    M = torch.zeros(qdim * qdim, n)
    for i in range(qdim):
        M[i * qdim + i, i] = 1.0
    k = qdim
    for r in range(1, qdim):
        for c in range(r):
            M[r * qdim + c, k] = 1.0
            if symmetric:
                M[c * qdim + r, k] = 1.0
            k += 1
    return M


class SymmMatNet(torch.nn.Module):
    """Batched symmetric matrices from their diagonal + strictly-lower entries: ``forward(l [B, q(q+1)/2]) -> [B, q, q]``
    (reference rigid_body_params.py:59-83; same vector layout)."""

    def __init__(self, qdim):
        super().__init__()
        self._qdim = int(qdim)
        self.register_buffer("_place_q", _placement_q(self._qdim, True), persistent=False)

    def forward(self, l):
        return (l @ self._place_q.t()).reshape(l.shape[0], self._qdim, self._qdim)


class CholeskyNet(torch.nn.Module):
    """Batched symmetric positive (semi-)definite matrices L L^T from the entries of the lower-triangular factor, with a
    positive bias on the factor's diagonal (reference rigid_body_params.py:86-132; same vector layout and method names)."""

    def __init__(self, qdim, bias):
        super().__init__()
        self._qdim, self._bias = int(qdim), bias
        self.register_buffer("_place_L", _placement_q(self._qdim, False), persistent=False)

    def get_raw_l(self, raw_l_input):
        return raw_l_input

    def get_l(self, raw_l_input):
        raw_l = self.get_raw_l(raw_l_input)
        shift = torch.zeros(raw_l.shape[-1], device=raw_l.device, dtype=raw_l.dtype)
        shift[: self._qdim] = self._bias
        return raw_l + shift

    def get_L(self, l):
        return (l @ self._place_L.t()).reshape(l.shape[0], self._qdim, self._qdim)

    def get_symm_pos_semi_def_matrix_and_l(self, raw_l_input):
        l = self.get_l(raw_l_input)
        L = self.get_L(l)
        return L @ L.transpose(-2, -1), l


class Symm3DInertiaMatrixNet(torch.nn.Module):
    """Any symmetric 3x3 matrix: diag(l[:3]) + off-diagonals l[3:] mirrored.  (reference rigid_body_params.py:387-404)"""

    def __init__(self, init_param_std=0.01, init_param=None, is_initializing_params=True):
        super().__init__()
        if init_param is None or not is_initializing_params:
            value = torch.empty(6).normal_(mean=0.0, std=init_param_std)
        else:
            value = _pack_diag_lower(init_param)
        self.l = torch.nn.Parameter(value.clone())
        self.register_buffer("_place", _placement(_SYMM_PLACEMENT), persistent=False)

    def forward(self):
        return (self._place @ self.l).reshape(3, 3)


class SymmPosDef3DInertiaMatrixNet(torch.nn.Module):
    """Symmetric positive definite by construction: L L^T + bias I with L lower triangular from ``l``.
    (reference rigid_body_params.py:342-384)"""

    def __init__(self, bias=1e-7, init_param_std=0.01, init_param=None, is_initializing_params=True):
        super().__init__()
        self.spd_3d_inertia_mat_diag_bias = float(bias)
        if init_param is None or not is_initializing_params:
            value = torch.empty(6).normal_(mean=0.0, std=init_param_std)
        else:
            target = torch.as_tensor(init_param, dtype=torch.float64).reshape(3, 3) - bias * torch.eye(3, dtype=torch.float64)
            value = _pack_diag_lower(torch.linalg.cholesky(target).to(torch.float32))
        self.l = torch.nn.Parameter(value.clone())
        self.register_buffer("_place", _placement(_LOWER_PLACEMENT), persistent=False)

    def forward(self):
        L = (self._place @ self.l).reshape(3, 3)
        return L @ L.t() + self.spd_3d_inertia_mat_diag_bias * torch.eye(3, device=self.l.device)


class CovParameterized3DInertiaMatrixNet(torch.nn.Module):
    """Physically consistent inertia: the density-weighted covariance Sigma = L L^T + bias I is what is
    parametrised, and I = tr(Sigma) E - Sigma (Wensing et al. 2017, IV.A-B; reference rigid_body_params.py:252-339),
    which enforces the triangle inequalities of the principal moments."""

    def __init__(self, bias=1.0e-7, init_param_std=0.01, init_param=None, is_initializing_params=True):
        super().__init__()
        self.spd_3d_cov_inertia_mat_diag_bias = float(bias)
        if init_param is None or not is_initializing_params:
            value = torch.empty(6).normal_(mean=0.0, std=init_param_std)
        else:
            inertia = torch.as_tensor(init_param, dtype=torch.float64).reshape(3, 3)
            cov = 0.5 * torch.trace(inertia) * torch.eye(3, dtype=torch.float64) - inertia
            value = _pack_diag_lower(torch.linalg.cholesky(cov - bias * torch.eye(3, dtype=torch.float64)).to(torch.float32))
        self.l = torch.nn.Parameter(value.clone())
        self.register_buffer("_place", _placement(_LOWER_PLACEMENT), persistent=False)

    def forward(self):
        L = (self._place @ self.l).reshape(3, 3)
        eye = torch.eye(3, device=self.l.device)
        cov = L @ L.t() + self.spd_3d_cov_inertia_mat_diag_bias * eye
        return (cov * eye).sum() * eye - cov   # tr(Sigma) E - Sigma (torch.trace's backward is not graph-capturable)


def _exp_so3(omega: torch.Tensor) -> torch.Tensor:
    """Rotation matrix of an axis-angle vector (Rodrigues), well-behaved at zero angle."""
    theta2 = (omega * omega).sum()
    theta = torch.sqrt(theta2 + 1e-30)
    K = omega.new_zeros(3, 3)
    K[0, 1], K[0, 2], K[1, 0], K[1, 2], K[2, 0], K[2, 1] = -omega[2], omega[1], omega[2], -omega[0], -omega[1], omega[0]
    a = torch.where(theta2 > 1e-12, torch.sin(theta) / theta, 1.0 - theta2 / 6.0)
    b = torch.where(theta2 > 1e-12, (1.0 - torch.cos(theta)) / (theta2 + 1e-30), 0.5 - theta2 / 24.0)
    return torch.eye(3, device=omega.device) + a * K + b * (K @ K)


def _log_so3(R: torch.Tensor) -> torch.Tensor:
    """Axis-angle vector of a rotation matrix (float64 in, angle < pi)."""
    cos = ((torch.trace(R) - 1.0) / 2.0).clamp(-1.0, 1.0)
    theta = torch.acos(cos)
    w = torch.stack([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    scale = torch.where(theta > 1e-8, theta / (2.0 * torch.sin(theta) + 1e-30), torch.tensor(0.5, dtype=R.dtype))
    return w * scale


class TriangParam3DInertiaMatrixNet(torch.nn.Module):
    """I = R diag(J1, J2, J3) R^T with J3 = sqrt(J1^2 + J2^2 - 2 J1 J2 cos(alpha)), 0 < alpha < pi: principal moments
    that satisfy the triangle inequalities by construction, R = exp of an axis-angle parameter.
    (reference rigid_body_params.py:137-249; its constructor passes an argument its ``UnconstrainedTensor`` does not
    take, so it cannot be instantiated there — the construction it describes is implemented here.)"""

    def __init__(self, bias, init_param_std=0.01, init_param=None, is_initializing_params=True):
        super().__init__()
        self._bias = float(bias)
        axis_angle, j1, j2, alpha_param = torch.empty(3).normal_(mean=0.0, std=init_param_std), None, None, None
        if init_param is not None and is_initializing_params:
            inertia = torch.as_tensor(init_param, dtype=torch.float64).reshape(3, 3)
            R, J, _ = torch.linalg.svd(inertia)
            if torch.linalg.det(R) < 0:   # a member of SO(3), not just O(3)
                R = R.clone()
                R[:, 0] = -R[:, 0]
            axis_angle = _log_so3(R).to(torch.float32)
            j1, j2 = J[0].to(torch.float32), J[1].to(torch.float32)
            assert j1 > bias and j2 > bias, "Please set bias value smaller, such that this condition is satisfied!"
            alpha = torch.acos(((J[0] ** 2 + J[1] ** 2 - J[2] ** 2) / (2.0 * J[0] * J[1])).clamp(-1.0, 1.0)) / math.pi
            alpha_param = torch.log(alpha / (1.0 - alpha)).to(torch.float32).reshape(1, 1)   # inverse sigmoid
        self.inertia_ori_axis_angle = torch.nn.Parameter(axis_angle.clone())
        self.J1net = PositiveScalar(min_val=bias, init_param_std=0.1, init_param=j1)
        self.J2net = PositiveScalar(min_val=bias, init_param_std=0.1, init_param=j2)
        self.alpha_param_net = UnconstrainedTensor(dim1=1, dim2=1, init_tensor=alpha_param, init_std=init_param_std)

    def forward(self):
        alpha = math.pi * torch.sigmoid(self.alpha_param_net().reshape(()))
        j1, j2 = self.J1net().reshape(()), self.J2net().reshape(())
        j3 = torch.sqrt(j1 * j1 + j2 * j2 - 2.0 * j1 * j2 * torch.cos(alpha))
        R = _exp_so3(self.inertia_ori_axis_angle)
        return R @ torch.diag(torch.stack([j1, j2, j3])) @ R.t()
