"""`lcp_physics.physics.utils`: the defaults and tensor helper the 3-D layer shares with it (utils.py:33-67, 270-283)."""
from diffsdfsim_amd.physics3d.utils import Defaults3D as Defaults, get_tensor  # noqa: F401
