"""External forces (mirrors sdf_physics/physics3d/forces.py:48-85, lcp_physics/physics/forces.py:34-72)."""
import torch

from .utils import get_tensor


def down_force(t):
    return ExternalForce3D.DOWN


class ExternalForce3D:
    """force_func(t) -> 6-vector [torque, force], times ``multiplier``."""
    UP = get_tensor([0, 0, 0, 0, 1, 0]); DOWN = get_tensor([0, 0, 0, 0, -1, 0])
    RIGHT = get_tensor([0, 0, 0, 1, 0, 0]); LEFT = get_tensor([0, 0, 0, -1, 0, 0])
    FRONT = get_tensor([0, 0, 0, 0, 0, 1]); BACK = get_tensor([0, 0, 0, 0, 0, -1])
    ROTX = get_tensor([1, 0, 0, 0, 0, 0]); ROTY = get_tensor([0, 1, 0, 0, 0, 0]); ROTZ = get_tensor([0, 0, 1, 0, 0, 0])
    ZEROS = get_tensor([0, 0, 0, 0, 0, 0])

    def __init__(self, force_func=down_force, multiplier=1.0):
        self.multiplier = multiplier
        self.force_func = force_func
        self.body = None

    def force(self, t):
        return self.force_func(t) * self.multiplier

    def set_body(self, body):
        self.body = body
        self.multiplier = get_tensor(self.multiplier)


class Gravity3D(ExternalForce3D):
    """Constant [0,0,0, 0,-1,0] * mass * g (-y is down)."""

    def __init__(self, g=10.0):
        self.multiplier = g
        self.body = None
        self.cached_force = None

    def force(self, t):
        return self.cached_force

    def set_body(self, body):
        self.body = body
        self.cached_force = ExternalForce3D.DOWN.to(body.mass.dtype) * body.mass * self.multiplier
