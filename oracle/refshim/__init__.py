"""Stand-ins for the third-party modules the reference imports (TEST INFRASTRUCTURE ONLY).

The reference (`/root/reference`, read-only, never copied) depends on packages that are
not installed in this container: ``ode``, ``pytorch3d``, ``pygame``, ``pyrender``,
``trimesh``, ``cvxpy``, ``pyhocon``, ``ev_sdf_utils``.  To
generate golden vectors from the reference's own CPU path we insert small stand-ins
into ``sys.modules``:

* *functional* stand-ins restate the documented arithmetic of the symbols the physics
  path really calls (``pytorch3d.transforms`` quaternion / SO(3) helpers, the ODE
  broadphase callback loop, ``trimesh.creation.icosphere``, ``ev_sdf_utils.marching_cubes / grid_interp``);
* *inert* stand-ins only satisfy ``import`` (render / plotting / alternative solver).

Nothing here is imported by the product (``diffsdfsim_amd``); it is used by
``oracle/gen_golden.py`` in the build container only — ``/root/reference`` does not
exist on the GPU box.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DIFFSDFSIM_REFERENCE", "/root/reference")


class _Meta(type):
    def __getattr__(cls, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Anything


class _Anything(metaclass=_Meta):
    """Absorbs any construction, call or attribute access (render / plotting objects)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Anything()


def _inert(name, attrs=()):
    m = types.ModuleType(name)
    for a in attrs:
        setattr(m, a, _Anything)
    m.__getattr__ = lambda item: _Anything  # PEP 562
    return m


def install(device="cpu"):
    """Insert the stand-ins and make the reference importable.  Returns the reference root."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present at %s (goldens can only be generated "
                           "in the build container)" % REFERENCE_ROOT)
    from . import fake_ev_sdf_utils, fake_ode, fake_pytorch3d, fake_trimesh

    sys.modules.setdefault("ode", fake_ode)
    p3d = types.ModuleType("pytorch3d")
    p3d.transforms = fake_pytorch3d
    loss = _inert("pytorch3d.loss", ["chamfer_distance"])
    p3d.loss = loss
    sys.modules.setdefault("pytorch3d", p3d)
    sys.modules.setdefault("pytorch3d.transforms", fake_pytorch3d)
    sys.modules.setdefault("pytorch3d.loss", loss)
    sys.modules.setdefault("trimesh", fake_trimesh)
    sys.modules.setdefault("trimesh.creation", fake_trimesh.creation)
    sys.modules.setdefault("ev_sdf_utils", fake_ev_sdf_utils)
    for name in ("pygame", "cvxpy", "pyrender", "pyhocon"):
        sys.modules.setdefault(name, _inert(name))
    from . import fake_sacred      # experiment scripts (sacred decorators, tensorboard writer) import and stay callable
    sys.modules.setdefault("sacred", fake_sacred)
    sys.modules.setdefault("sacred.utils", fake_sacred.utils)
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        sys.modules["torch.utils.tensorboard"] = _inert("torch.utils.tensorboard", ["SummaryWriter"])
    exp = os.path.join(REFERENCE_ROOT, "experiments", "trajectory_fitting")
    if exp not in sys.path:
        sys.path.append(exp)
    os.environ.setdefault("IGR_PATH", "/tmp")
    os.environ.setdefault("PYOPENGL_PLATFORM", "egl")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    import warnings
    warnings.filterwarnings("ignore", category=UserWarning)
    warnings.filterwarnings("ignore", category=DeprecationWarning)

    import torch
    # circular import in the reference: physics must be imported before lcp.lcp
    import lcp_physics.physics  # noqa: F401
    import sdf_physics.physics3d.utils as u3
    u3.Defaults3D.DEVICE = torch.device(device)
    return REFERENCE_ROOT
