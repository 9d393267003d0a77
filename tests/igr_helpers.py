"""Build BatchEngine specs / worlds of the neural-SDF goldens (tests/golden/rollout_igr_*.npz, oracle/gen/gen_igr_golden.py)."""
import numpy as np

import rollout_helpers as R


def seeded_weights(g):
    """The network the golden was recorded with: numpy-seeded geometric initialisation (oracle/igr_oracle.py)."""
    from oracle import igr_oracle
    return igr_oracle.geometric_init(int(g["igr_seed"]), float(g["igr_radius"]))


def spec_from_golden(g, copies=1, packed=None):
    from diffsdfsim_amd import igr, meshes, meshsdf
    if packed is None:
        packed = igr.pack_weights(*seeded_weights(g))
    nb = len(g["mass"])
    kind = g["kind"]
    ms = []
    for i in range(nb):
        prm = g["shape_prm"][i]
        if "verts_%d" % i in g:          # the neural body's level-set mesh as the reference built it
            ms.append((g["verts_%d" % i], g["faces_%d" % i]))
        elif g["custom_mesh"][i]:
            v, f, _tie = meshes.box_mesh(prm)
            ms.append((v, f))
        else:                            # level-set mesh of a primitive, rebuilt with the device mesher (same case tables)
            scale = max(prm[0], prm[1] / 2) * 1.5 if kind[i] == 2 else prm.max() * 1.5 / 2
            v, f = meshsdf.primitive_mesh(int(kind[i]), np.concatenate([prm, [0.0]]) / scale, res=128)
            assert (len(v), len(f)) == tuple(g["meshsize_%d" % i]), "device marching cubes and the golden's mesh differ in size"
            ms.append(((v * scale).cpu().numpy(), f.cpu().numpy()))
    rep = lambda a: np.repeat(np.asarray(a)[None], copies, axis=0)
    Je = np.zeros((6 * len(g["fixed"]), 6 * nb))
    for k, b in enumerate(g["fixed"]):
        Je[6 * k:6 * k + 6, 6 * b:6 * b + 6] = np.eye(6)
    aux = np.where(kind == 6, float(g["igr_scale"]), 0.0)
    return dict(pose=rep(g["pose0"]), vel=rep(g["vel0"]), mass=rep(g["mass"]), inertia=rep(g["inertia"]),
                restitution=rep(g["restitution"]), fric=rep(g["fric"]), fext=rep(g["fext"]), shape_type=rep(kind.astype(np.int32)),
                shape_prm=rep(g["shape_prm"]), shape_aux=rep(aux), mesh_id=rep(np.arange(nb)), meshes=ms, Je=rep(Je),
                no_contact=np.asarray(g["no_contact"], np.uint8), igr_net=packed)


def engine_kwargs(g, **over):
    kw = dict(dt=float(g["dt"]), eps=float(g["eps"]), tol=float(g["tol"]), fric_dirs=int(g["fric_dirs"]), toc_diff=True,
              strict_no_pen=bool(g["strict_no_pen"]), maxc=64, max_cand=4096, max_pc=64)
    kw.update(over)
    return kw


def torch_network(g):
    """An ImplicitNet-shaped torch module (layers lin0..lin8, as the IGR repository's class the reference loads) holding the
    golden's seeded weights: what a caller hands to decode_igr."""
    import torch
    Ws, bs = seeded_weights(g)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for l, (W, b) in enumerate(zip(Ws, bs)):
                lin = torch.nn.Linear(W.shape[1], W.shape[0]).double()
                with torch.no_grad():
                    lin.weight.copy_(torch.tensor(W)); lin.bias.copy_(torch.tensor(b))
                setattr(self, "lin%d" % l, lin)
    return Net()
