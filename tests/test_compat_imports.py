"""CPU: the reference's module names resolve to this build when compat/ is on the path (north_star: the
sdf_physics.physics3d World/Body/run_world and lcp_physics.lcp.LCPFunction surfaces stay intact so that demos/ and
experiments/ keep their import blocks).  The import lines below are the union of what the reference's demos/ and experiments/
import from the two packages (grep of /root/reference)."""
import os
import subprocess
import sys

from helpers import ROOT

IMPORTS = """
from sdf_physics.physics3d.bodies import SDFBox, SDFSphere, SDFCylinder, SDFBoxRounded, SDF3D
from sdf_physics.physics3d.constraints import TotalConstraint3D, XConstraint, YConstraint, ZConstraint
from sdf_physics.physics3d.forces import Gravity3D, ExternalForce3D
from sdf_physics.physics3d.utils import get_tensor, Rx, Ry, Recorder3D, Defaults3D, load_igrnet, decode_igr
from sdf_physics.physics3d.world import World3D, run_world
from lcp_physics.lcp.lcp import LCPFunction
from lcp_physics.physics.bodies import Circle, Rect, Hull
from lcp_physics.physics.constraints import TotalConstraint
from lcp_physics.physics.forces import Gravity
from lcp_physics.physics.contacts import DiffContactHandler
from lcp_physics.physics.world import World, run_world as run_world_2d
import diffsdfsim_amd.physics2d as P2
assert World is P2.World and Circle is P2.Circle and DiffContactHandler is P2.DiffContactHandler
import diffsdfsim_amd.physics3d as P
assert World3D is P.World3D and SDF3D is P.SDF3D and run_world is P.run_world
assert Defaults3D.CUSTOM_MESH is False and Defaults3D.CUSTOM_INERTIA is False and Defaults3D.FRIC_DIRS == 8
import inspect
sig = inspect.signature(World3D.__init__)
for k in ("bodies", "constraints", "dt", "engine", "contact_callback", "eps", "tol", "fric_dirs", "post_stab",
          "strict_no_penetration", "time_of_contact_diff", "stop_contact_grad", "stop_friction_grad", "detach_contact_b2"):
    assert k in sig.parameters, k
sig = inspect.signature(SDF3D.__init__)
for k in ("pos", "scale", "sdf_func", "params", "grad_func", "vel", "mass", "restitution", "fric_coeff", "eps"):
    assert k in sig.parameters, k
print("ok")
"""


def test_reference_import_block_resolves_to_this_build():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT]))
    out = subprocess.run([sys.executable, "-c", IMPORTS], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
