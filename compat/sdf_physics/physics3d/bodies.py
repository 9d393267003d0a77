"""`sdf_physics.physics3d.bodies` of the reference, served by the MI355X build (see compat/README.md)."""
from diffsdfsim_amd.physics3d.bodies import *  # noqa: F401,F403
from diffsdfsim_amd.physics3d import bodies as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
