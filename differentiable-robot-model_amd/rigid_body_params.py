"""Learnable parametrisations that FEED the kernels' constant table.

The reference ships a family of nn.Modules for learnable link parameters
(reference ``differentiable_robot_model/rigid_body_params.py``).  They run once per
step on O(1) data upstream of the hot path, so they stay plain torch; any
``nn.Module`` whose ``forward()`` returns a tensor of the parameter's shape works
with ``make_link_param_learnable`` (the reference's own modules included).  The
three generic ones used by the reference's kinematics examples are provided here
under the same names and constructor arguments.
"""
import torch


class UnconstrainedScalar(torch.nn.Module):
    """A free scalar; forward() -> [1].  (reference rigid_body_params.py:14-23)"""

    def __init__(self, init_val=None):
        super().__init__()
        value = torch.rand(1) if init_val is None else torch.as_tensor(init_val, dtype=torch.float32).reshape(1)
        self.param = torch.nn.Parameter(value.clone())

    def forward(self):
        return self.param


class PositiveScalar(torch.nn.Module):
    """A scalar kept >= min_val by construction: min_val + l^2.  (reference rigid_body_params.py:26-43)"""

    def __init__(self, min_val=0.0, init_param_std=1.0, init_param=None):
        super().__init__()
        self._min_val = float(min_val)
        if init_param is None:
            value = torch.empty(1).normal_(mean=0.0, std=init_param_std)
        else:  # init_param is the VALUE to start from, as in the reference: l = sqrt(value - min_val)
            value = torch.sqrt(torch.as_tensor(init_param, dtype=torch.float32).reshape(1) - self._min_val)
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
