"""Test-only stand-in for the third-party ``urdf_parser_py.urdf`` module.

The reference library imports ``urdf_parser_py.urdf.URDF`` (reference
``differentiable_robot_model/urdf_utils.py:9,14``); that package is not
installed in this container and cannot be fetched.  This file exposes exactly
the attributes the reference reads (``urdf_utils.py:17-26, 44-75, 85-108``) so
that the *unmodified* reference can be imported from ``/root/reference`` by
``tests/golden/make_golden.py`` and by the CPU parity tests.

It is NOT part of the product: the product has its own URDF ingest in
``differentiable-robot-model_amd/urdf_utils.py``.
"""
import xml.etree.ElementTree as ET


def _vec(text, default):
    if text is None:
        return list(default)
    return [float(v) for v in text.split()]


class _Pose:
    def __init__(self, node):
        if node is None:
            self.xyz, self.rpy = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
        else:
            self.xyz = _vec(node.get("xyz"), [0.0, 0.0, 0.0])
            self.rpy = _vec(node.get("rpy"), [0.0, 0.0, 0.0])

    @property
    def position(self):
        return self.xyz

    @property
    def rotation(self):
        return self.rpy


class _Inertia:
    def __init__(self, node):
        for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz"):
            setattr(self, k, float(node.get(k, 0.0)) if node is not None else 0.0)


class _Inertial:
    def __init__(self, node):
        mass = node.find("mass")
        self.mass = float(mass.get("value")) if mass is not None else 0.0
        self.origin = _Pose(node.find("origin"))
        self.inertia = _Inertia(node.find("inertia"))


class _Link:
    def __init__(self, node):
        self.name = node.get("name")
        inertial = node.find("inertial")
        self.inertial = _Inertial(inertial) if inertial is not None else None


class _Limit:
    def __init__(self, node):
        self.effort = float(node.get("effort", 0.0))
        self.lower = float(node.get("lower", 0.0))
        self.upper = float(node.get("upper", 0.0))
        self.velocity = float(node.get("velocity", 0.0))


class _Dynamics:
    def __init__(self, node):
        self.damping = float(node.get("damping", 0.0))
        self.friction = float(node.get("friction", 0.0))


class _Joint:
    def __init__(self, node):
        self.name = node.get("name")
        self.type = node.get("type")
        self.parent = node.find("parent").get("link")
        self.child = node.find("child").get("link")
        self.origin = _Pose(node.find("origin"))
        axis = node.find("axis")
        self.axis = _vec(axis.get("xyz"), [1.0, 0.0, 0.0]) if axis is not None else [1.0, 0.0, 0.0]
        limit = node.find("limit")
        self.limit = _Limit(limit) if limit is not None else None
        dyn = node.find("dynamics")
        self.dynamics = _Dynamics(dyn) if dyn is not None else None


class URDF:
    def __init__(self):
        self.links, self.joints, self.name = [], [], ""

    @classmethod
    def from_xml_file(cls, path):
        try:
            root = ET.parse(path).getroot()
        except ET.ParseError:
            # fetch.urdf carries an undeclared "sensor:" namespace prefix inside a <gazebo> block (the real urdf_parser_py
            # reads it through lxml's recovering parser); neutralise prefixes — <gazebo> content is never looked at
            import re
            with open(path) as f:
                root = ET.fromstring(re.sub(r"<(/?)([A-Za-z_][\w.-]*):", r"<\1\2_", f.read()))
        robot = cls()
        robot.name = root.get("name", "")
        robot.links = [_Link(n) for n in root.findall("link")]
        robot.joints = [_Joint(n) for n in root.findall("joint")]
        return robot
