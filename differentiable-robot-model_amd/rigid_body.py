"""Per-link parameter holder of the MI355X engine.

In the reference a ``DifferentiableRigidBody`` both stores a link's parameters
AND does the per-link math (joint transform, recursive FK; reference
``rigid_body.py:24-171``).  Here all math lives in the HIP kernels; this class
only keeps the *parameters* reachable under the reference's attribute names so
that user code, tests and the learnable-parameter mechanism keep working:

  body.trans() / body.rot_angles() / body.joint_damping()      (rigid_body.py:47-49)
  body.inertia.mass() / .com() / .inertia_mat()                 (spatial_vector_algebra.py:312-314)
  body.joint_axis, body.joint_limits, body.joint_id, body.joint_idx, body.name

Each parameter is a zero-argument callable: a lambda returning the URDF
constant, or — after ``make_link_param_learnable`` — an ``nn.Module`` registered
under the same attribute name (robot_model.py:682-689), whose output feeds the
kernels' constant table through differentiable torch ops.
"""
import weakref
from typing import List, Optional

import torch


class LinkPose(object):
    """World pose of a link after ``update_kinematic_state``: the accessors of the reference's
    ``CoordinateTransform`` that callers read (spatial_vector_algebra.py:56-136)."""

    def __init__(self, rot: torch.Tensor, trans: torch.Tensor, quat: Optional[torch.Tensor] = None):
        self._rot, self._trans, self._quat = rot, trans, quat

    def rotation(self):
        return self._rot

    def translation(self):
        return self._trans

    def get_quaternion(self):
        return self._quat


class LinkVelocity(object):
    """Body-frame spatial velocity of a link at its origin: ``.lin`` / ``.ang`` as in the reference's
    ``SpatialMotionVec`` (spatial_vector_algebra.py:175-250)."""

    def __init__(self, lin: torch.Tensor, ang: torch.Tensor):
        self.lin, self.ang = lin, ang

    def get_vector(self):
        return torch.cat([self.ang, self.lin], dim=1)


class SpatialRigidBodyInertiaParams(torch.nn.Module):
    """Holder of mass / com / inertia_mat (about the com); no arithmetic here."""

    def __init__(self, rigid_body_params, device="cpu"):
        super().__init__()
        self._device = torch.device(device)
        self.mass = lambda: rigid_body_params["mass"]
        self.com = lambda: rigid_body_params["com"]
        self.inertia_mat = lambda: rigid_body_params["inertia_mat"]

    def _get_parameter_values(self):
        return self.mass(), self.com(), self.inertia_mat()


class DifferentiableRigidBody(torch.nn.Module):
    _children: List["DifferentiableRigidBody"]

    def __init__(self, rigid_body_params, device="cpu"):
        super().__init__()
        self._parent: Optional["DifferentiableRigidBody"] = None
        self._children = []
        self._device = torch.device(device)
        self.joint_id = rigid_body_params["joint_id"]
        self.name = rigid_body_params["link_name"]
        self.joint_name = rigid_body_params["joint_name"]
        self.joint_type = rigid_body_params["joint_type"]
        self.joint_idx = None  # DoF column, set by the model for non-fixed joints

        # parameters that can be made learnable
        self.inertia = SpatialRigidBodyInertiaParams(rigid_body_params, device=self._device)
        self.joint_damping = lambda: rigid_body_params["joint_damping"]
        self.trans = lambda: rigid_body_params["trans"].reshape(1, 3)
        self.rot_angles = lambda: rigid_body_params["rot_angles"].reshape(1, 3)

        self.joint_axis = rigid_body_params["joint_axis"]
        self.joint_limits = rigid_body_params["joint_limits"]
        self._model_ref = None   # set by the model: (weakref to it, this link's index)

    # State written by ``update_kinematic_state`` in the reference (robot_model.py:186, 193).  The kernels keep
    # no per-link state; these are computed on first access from the (q, qd) the model was last updated with.
    @property
    def pose(self) -> LinkPose:
        model, idx = self._model_ref[0](), self._model_ref[1]
        return model._link_pose(idx)

    @property
    def vel(self) -> LinkVelocity:
        model, idx = self._model_ref[0](), self._model_ref[1]
        return model._link_velocity(idx)

    def _attach(self, model, idx: int):
        object.__setattr__(self, "_model_ref", (weakref.ref(model), idx))

    # kinematic tree bookkeeping (names as in rigid_body.py:78-82)
    def set_parent(self, link: "DifferentiableRigidBody"):
        # plain attribute, bypassing nn.Module registration (a registered parent would
        # make the module tree cyclic)
        object.__setattr__(self, "_parent", link)

    def add_child(self, link: "DifferentiableRigidBody"):
        self._children.append(link)

    # The reference advances its per-link Python recursion through these three methods (rigid_body.py:85-165); here the
    # whole sweep is one kernel launched from the model, so calling them on a single body has no meaning.
    def _per_link_step(self, name):
        raise NotImplementedError(
            "DifferentiableRigidBody.%s is a step of the reference's per-link Python loop; this engine runs the whole sweep in "
            "one kernel — call compute_forward_kinematics / update_kinematic_state / compute_inverse_dynamics on the model "
            "(body.pose and body.vel are filled from the last update_kinematic_state)" % name)

    def update_joint_state(self, q, qd):
        self._per_link_step("update_joint_state")

    def update_joint_acc(self, qdd):
        self._per_link_step("update_joint_acc")

    def forward_kinematics(self, q_dict):
        self._per_link_step("forward_kinematics")

    def get_joint_limits(self):
        return self.joint_limits

    def get_joint_damping_const(self):
        return self.joint_damping()
