"""`lcp_physics.physics.world`: `World`, `run_world` (world.py:40-139, 241-379, 513-587) on the device library."""
from diffsdfsim_amd.physics2d.world import World, run_world  # noqa: F401
