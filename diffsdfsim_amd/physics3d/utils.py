"""Defaults of the 3-D SDF layer (mirrors sdf_physics/physics3d/utils.py:41-62, lcp_physics/physics/utils.py:33-67)."""
import torch


class Defaults3D:
    DIM = 3
    EPSILON = 0.001          # contact detection band
    TOL = 1e-8               # penetration tolerance
    RESTITUTION = 0.5
    FRIC_COEFF = 0.9
    FRIC_DIRS = 8
    FPS = 30
    DT = 1.0 / FPS
    ENGINE = "HipPdipmEngine"
    CONTACT = "FWContactHandler"
    SOLVER = 1
    DTYPE = torch.double
    DEVICE = torch.device("cuda:0")
    POST_STABILIZATION = False
    CUSTOM_MESH = False      # as sdf_physics/physics3d/utils.py:56-57: level-set (marching cubes) mesh and mesh inertia unless
    CUSTOM_INERTIA = False   # the caller asks for the analytic ones


def get_tensor(x, base_tensor=None, **kw):
    """Wrap array or scalar in a float64 tensor (reference utils.py:270-283); host-side parameters live on the CPU
    unless the caller passes device tensors -- the engine copies them to the HIP device."""
    if isinstance(x, torch.Tensor):
        return x
    if base_tensor is not None:
        return base_tensor.new_tensor(x, **kw)
    return torch.tensor(x, dtype=Defaults3D.DTYPE, **kw)


def decode_igr(network):
    """`decode_igr` (sdf_physics/physics3d/utils.py:330-350): IGR ImplicitNet -> ``sdf(pts, latent)`` evaluated on the device;
    the result is what ``SDF3D(sdf_func=...)`` takes (diffsdfsim_amd/igr.py)."""
    from ..igr import decode_igr as _decode
    return _decode(network)
