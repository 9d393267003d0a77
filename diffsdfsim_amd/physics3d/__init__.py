from .bodies import Body3D, SDF3D, SDFBowl, SDFBox, SDFBoxRounded, SDFBrick, SDFCylinder, SDFGrid3D, SDFSphere  # noqa: F401
from .constraints import RotConstraint3D, TotalConstraint3D, XConstraint, YConstraint, ZConstraint  # noqa: F401
from .forces import ExternalForce3D, Gravity3D  # noqa: F401
from .utils import Defaults3D  # noqa: F401
from .world import BatchWorld3D, World3D, run_world  # noqa: F401
from .engines import Engine, HipPdipmEngine, PdipmEngine  # noqa: F401
