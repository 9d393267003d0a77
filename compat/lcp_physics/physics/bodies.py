"""`lcp_physics.physics.bodies`: the 2-D bodies of the reference (bodies.py:32-323) on the device library."""
from diffsdfsim_amd.physics2d.world import Body, Circle, Hull, Rect  # noqa: F401
