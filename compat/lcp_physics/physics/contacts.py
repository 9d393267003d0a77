"""`lcp_physics.physics.contacts`: the differentiable 2-D contact handler (contacts.py:55-357) on the device."""
from diffsdfsim_amd.physics2d import DiffContactHandler  # noqa: F401
