"""`lcp_physics.physics.forces`: `Gravity` (forces.py:55-72)."""
from diffsdfsim_amd.physics2d.world import Gravity  # noqa: F401
