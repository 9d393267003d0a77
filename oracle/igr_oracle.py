"""CPU oracle (TEST INFRASTRUCTURE) of the IGR neural SDF and its input gradient.

Restates ``decode_igr`` (`sdf_physics/physics3d/utils.py:330-350`) on the ImplicitNet the reference loads from the
external IGR repo (`utils.py:300-308`; architecture from `IGR_data/train_configs/bob_spot_setup.conf:38-45`:
dims 8 x 128, skip_in [4], Softplus(beta = 100), d_in 3 + latent 2).  The IGR repo and its trained weights are
not in the reference tree nor on this machine, so parity is UNPINNED against the reference: this
oracle checks the HIP kernel against plain numpy on seeded geometric-init weights.
"""
import numpy as np

H, DIN, SKIP = 128, 5, 4


def geometric_init(seed=0, radius_init=1.0):
    """IGR's geometric initialisation (Atzmon & Lipman 2020): last layer mean sqrt(pi)/sqrt(dim), bias -r."""
    r = np.random.default_rng(seed)
    dims = [DIN] + [H] * 8 + [1]
    Ws, bs = [], []
    for l in range(9):
        out = dims[l + 1] - DIN if l + 1 == SKIP else dims[l + 1]
        if l == 8:
            W = r.normal(np.sqrt(np.pi) / np.sqrt(dims[l]), 1e-5, (out, dims[l]))
            b = np.full(out, -radius_init)
        else:
            W = r.normal(0.0, np.sqrt(2) / np.sqrt(out), (out, dims[l]))
            b = np.zeros(out)
        Ws.append(W); bs.append(b)
    return Ws, bs


def softplus100(z):
    bz = 100.0 * z
    h = np.where(bz > 20.0, z, np.log1p(np.exp(np.minimum(bz, 20.0))) / 100.0)
    dh = np.where(bz > 20.0, 1.0, 1.0 / (1.0 + np.exp(-np.minimum(bz, 20.0))))
    return h, dh


def query(pts, latent, Ws, bs, wrt="xyz"):
    """-> sdf [n], grad [n,3] = d sdf / d xyz (what autograd returns in SDF3D.query_sdfs, bodies.py:730-745), or with
    wrt="latent" (d sdf / d latent_0, d sdf / d latent_1, 0): the d phi / d theta of the MeshSDF backward."""
    pts = np.asarray(pts, np.float64)
    n = len(pts)
    inp = np.concatenate([np.broadcast_to(latent, (n, 2)), pts], 1)
    x = inp
    J = np.zeros((n, DIN, 3))
    if wrt == "latent":
        J[:, 0, 0] = 1.0; J[:, 1, 1] = 1.0                  # d input / d latent
    else:
        J[:, 2:, :] = np.eye(3)                             # d input / d xyz
    Jx = J
    for l in range(9):
        if l == SKIP:
            x = np.concatenate([x, inp], 1) / np.sqrt(2)
            Jx = np.concatenate([Jx, J], 1) / np.sqrt(2)
        z = x @ Ws[l].T + bs[l]
        Jz = np.einsum("oi,nid->nod", Ws[l], Jx)
        if l < 8:
            x, dh = softplus100(z)
            Jx = dh[:, :, None] * Jz
        else:
            x, Jx = z, Jz
    return x[:, 0], Jx[:, 0, :]
