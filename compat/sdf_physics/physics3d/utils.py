"""`sdf_physics.physics3d.utils` of the reference, served by the MI355X build (see compat/README.md)."""
from diffsdfsim_amd.physics3d.utils import *  # noqa: F401,F403
from diffsdfsim_amd.physics3d import utils as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
