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


def _rot4(rows):
    return torch.stack([torch.stack([torch.as_tensor(x, dtype=Defaults3D.DTYPE) for x in r]) for r in rows])


def Rx(theta):
    """Homogeneous 4x4 rotation about x (sdf_physics/physics3d/utils.py:183-188)."""
    t = get_tensor(theta); c, s = torch.cos(t), torch.sin(t)
    return _rot4([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]])


def Ry(theta):
    """About y, with the reference's sign convention (-sin in the first row; utils.py:191-196)."""
    t = get_tensor(theta); c, s = torch.cos(t), torch.sin(t)
    return _rot4([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]])


def Rz(theta):
    """About z (utils.py:199-204)."""
    t = get_tensor(theta); c, s = torch.cos(t), torch.sin(t)
    return _rot4([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])


class Recorder3D:
    """Constructor-compatible stand-in for the reference's pyrender frame recorder (utils.py:65-135).  Rendering is outside the
    hot path this package rebuilds: the recorder accepts the demo's arguments, keeps the time stamps it is asked to record and
    renders nothing."""

    def __init__(self, dt, scene=None, path=None, resolution=(640, 480), rotate=False, rotate_rate=0.0, rotate_axis=(0, 0, 1),
                 record_points=False, record_seg=False, noise_factor=0.0, save_to_disk=False):
        self.dt, self.scene, self.path, self.frame, self.prev_t, self.times = dt, scene, path, 0, 0.0, []

    def record(self, t, seg_node_map=None):
        if t - self.prev_t >= self.dt:
            self.times.append(float(t)); self.frame += 1; self.prev_t += self.dt
        return None, None, None, None, []


class _LoadedImplicitNet(torch.nn.Module):
    """lin0 .. lin{n-1} rebuilt from an IGR checkpoint's state dict: the shape decode_igr needs (the class itself lives in the
    external IGR repository, which the reference imports by file path, utils.py:300-308)."""

    def __init__(self, state):
        super().__init__()
        n = 0
        while "lin%d.weight" % n in state:
            W = state["lin%d.weight" % n]
            lin = torch.nn.Linear(W.shape[1], W.shape[0]).to(Defaults3D.DTYPE)
            with torch.no_grad():
                lin.weight.copy_(W); lin.bias.copy_(state["lin%d.bias" % n])
            setattr(self, "lin%d" % n, lin)
            n += 1
        self.num_layers = n + 1


def load_igrnet(experiment_dir, timestamp='latest', checkpoint='latest', run=None):
    """`load_igrnet` (utils.py:286-327): (network, latent codes) of a trained IGR run -- newest time stamp directory unless given,
    ``checkpoints/LatentCodes/<ckpt>.pth['latent_codes']`` and ``checkpoints/ModelParameters/<ckpt>.pth['model_state_dict']``.
    The layer sizes are read off the state dict (the reference reads them from exp.conf through pyhocon)."""
    import os
    if timestamp == 'latest':
        stamps = sorted(os.listdir(experiment_dir))
        if not stamps:
            raise FileNotFoundError('No timestamp directories found in {}'.format(experiment_dir))
        timestamp = stamps[-1]
    ck = os.path.join(experiment_dir, timestamp, 'checkpoints')
    lat = torch.load(os.path.join(ck, 'LatentCodes', str(checkpoint) + '.pth'), map_location='cpu')["latent_codes"]
    state = torch.load(os.path.join(ck, 'ModelParameters', str(checkpoint) + '.pth'), map_location='cpu')["model_state_dict"]
    net = _LoadedImplicitNet({k: v.detach().to(Defaults3D.DTYPE) for k, v in state.items()})
    net.eval()
    return net, lat.detach().to(Defaults3D.DTYPE)
