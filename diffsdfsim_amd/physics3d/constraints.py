"""Constant-Jacobian joints (mirrors sdf_physics/physics3d/constraints.py:33-146; GripperJoint is out of scope)."""
import torch


class _Fixed:
    static = True
    rows = ()

    def __init__(self, body1):
        self.body1, self.body2 = body1, None
        self.num_constraints = len(self.rows)

    def J(self):
        return torch.eye(6, dtype=torch.float64)[list(self.rows)], None


class XConstraint(_Fixed):
    rows = (3,)


class YConstraint(_Fixed):
    rows = (4,)


class ZConstraint(_Fixed):
    rows = (5,)


class RotConstraint3D(_Fixed):
    rows = (0, 1, 2)


class TotalConstraint3D(_Fixed):
    rows = (0, 1, 2, 3, 4, 5)
