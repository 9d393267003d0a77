"""IGR neural SDF on the fp64 matrix cores (C ABI: dss_igr_query, csrc/igr_mlp.hip).

``pack_weights`` turns the nine Linear layers of the reference's ImplicitNet (PyTorch layout weight[out, in]) into
the operand set of the kernel: layer 0 and layer 8 stay dense, the seven 128x128 layers go into MFMA fragment order
``[tile t][k-step][lane] = W[16 t + (lane & 15)][4 ks + (lane >> 4)]`` so a wavefront fetches each B fragment with
one coalesced 512-byte load.  Layer 3 has 123 outputs (the skip concat appends the 5 inputs): its missing rows are
zero.  Real IGR checkpoints (`utils.py:310-320`) are not available offline; any state_dict with the same shapes works.
"""
import ctypes

import numpy as np
import torch

from . import _lib

H = 128


def pack_weights(Ws, bs, device="cuda"):
    Ws = [np.asarray(w, np.float64) for w in Ws]
    bs = [np.asarray(b, np.float64) for b in bs]
    assert Ws[0].shape == (H, 5) and Ws[8].shape == (1, H) and Ws[3].shape == (H - 5, H)
    packed = np.zeros((7, 8, 32, 64))
    bh = np.zeros((7, H))
    lane = np.arange(64)
    for l in range(1, 8):
        W = np.zeros((H, H)); W[: Ws[l].shape[0]] = Ws[l]
        bh[l - 1, : len(bs[l])] = bs[l]
        for t in range(8):
            for ks in range(32):
                packed[l - 1, t, ks] = W[16 * t + (lane & 15), 4 * ks + (lane >> 4)]
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=device)
    return dict(W0=t(Ws[0]), b0=t(bs[0]), Wp=t(packed), bh=t(bh), W8=t(Ws[8][0]), b8=t(bs[8]))


def igr_query(pts, latent, P, wrt="xyz"):
    """pts [n,3], latent [2] (float64, HIP device) -> sdf [n], d sdf / d xyz [n,3]
    (wrt="latent": [n,3] = d sdf / d latent_0, d sdf / d latent_1, 0)."""
    _lib.require_device(pts, latent)
    L = _lib.lib()
    n = pts.shape[0]
    sdf = torch.empty(n, dtype=torch.float64, device=pts.device)
    grad = torch.empty(n, 3, dtype=torch.float64, device=pts.device)
    fn = L.dss_igr_query_latent_grad if wrt == "latent" else L.dss_igr_query
    rc = fn(_lib.ptr(pts.contiguous()), _lib.ptr(latent.contiguous()), _lib.ptr(P["W0"]), _lib.ptr(P["b0"]),
                         _lib.ptr(P["Wp"]), _lib.ptr(P["bh"]), _lib.ptr(P["W8"]), _lib.ptr(P["b8"]), int(n), _lib.ptr(sdf),
                         _lib.ptr(grad), _lib.stream_ptr(pts.device))
    _lib.check(rc, "dss_igr_query")
    return sdf, grad


class _Net(ctypes.Structure):      # DssIgrNet (include/diffsdfsim_hip.h)
    _fields_ = [(k, ctypes.c_void_p) for k in ("W0", "b0", "Wp", "bh", "W8", "b8")]


def igr_values(pts, latent, P):
    """Values only (query_sdfs with return_grads=False): pts [n,3], latent [2] -> sdf [n].  A quarter of the matrix work of
    `igr_query` (no tangents) -- what sampling the 128^3 grid of a level-set mesh needs (bodies.py:657-664)."""
    _lib.require_device(pts, latent)
    L = _lib.lib()
    n = pts.shape[0]
    sdf = torch.empty(n, dtype=torch.float64, device=pts.device)
    net = _Net(*[P[k].data_ptr() for k in ("W0", "b0", "Wp", "bh", "W8", "b8")])
    lat = torch.zeros(3, dtype=torch.float64, device=pts.device)
    lat[:2] = latent.reshape(-1)[:2]
    rc = L.dss_igr_query_list(ctypes.byref(net), _lib.ptr(pts.contiguous()), None, _lib.ptr(lat), 3, None, int(n), 2, _lib.ptr(sdf), None,
                              _lib.stream_ptr(pts.device))
    _lib.check(rc, "dss_igr_query_list")
    return sdf


def weights_from_module(network):
    """(Ws, bs) of an IGR ``ImplicitNet``-like torch module: attributes ``lin0 .. lin8`` (torch.nn.Linear), as the
    external IGR repository defines it and the reference loads it (`utils.py:300-320`).  Only the bob_spot_setup shape
    (IGR_data/train_configs/bob_spot_setup.conf:38-45: input 2 + 3, eight hidden layers of 128, skip at layer 4) runs on
    the device kernels."""
    Ws, bs = [], []
    for l in range(9):
        lin = getattr(network, "lin%d" % l, None)
        if lin is None:
            raise ValueError("decode_igr needs an ImplicitNet with layers lin0..lin8 (got %r)" % type(network))
        Ws.append(lin.weight.detach().cpu().double().numpy())
        bs.append(lin.bias.detach().cpu().double().numpy())
    if getattr(network, "lin9", None) is not None or Ws[0].shape != (H, 5) or Ws[8].shape != (1, H) or Ws[3].shape != (H - 5, H):
        raise NotImplementedError("only the 5 -> 8 x 128 -> 1 network with a skip connection into layer 4 is built for the device")
    return Ws, bs


class IgrNet:
    """A network on the device: packed weights for the kernels + the plain layers (host) for reference."""

    def __init__(self, Ws, bs, device="cuda"):
        self.Ws, self.bs = Ws, bs
        self.packed = pack_weights(Ws, bs, device)

    @classmethod
    def from_module(cls, network, device="cuda"):
        hit = getattr(network, "_dss_igr_net", None)
        if hit is None:
            hit = cls(*weights_from_module(network), device=device)
            try:
                network._dss_igr_net = hit       # pack once per network object
            except Exception:
                pass
        return hit


def decode_igr(network):
    """`decode_igr` (`sdf_physics/physics3d/utils.py:330-350`): network -> ``sdf(pts, latent)``.  The returned function
    evaluates on the device (dss_igr_query) and carries the packed network (``.igr``), which is how ``SDF3D`` recognises a
    neural SDF it can hand to the stepper's kernels."""
    net = network if isinstance(network, IgrNet) else IgrNet.from_module(network)

    def sdf(pts, latent, max_batch=32 ** 3):
        dev = net.packed["W0"].device
        p = torch.as_tensor(pts, dtype=torch.float64).to(dev).contiguous()
        lat = torch.as_tensor(latent, dtype=torch.float64).detach().to(dev).contiguous()
        return igr_query(p, lat, net.packed)[0]

    sdf.igr = net
    return sdf
