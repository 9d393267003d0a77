"""`lcp_physics.physics.constraints`: `TotalConstraint` (constraints.py:196-211); the joints are not rebuilt."""
from diffsdfsim_amd.physics2d.world import TotalConstraint  # noqa: F401
