"""`sdf_physics.physics3d.world` of the reference, served by the MI355X build (see compat/README.md)."""
from diffsdfsim_amd.physics3d.world import *  # noqa: F401,F403
from diffsdfsim_amd.physics3d import world as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
