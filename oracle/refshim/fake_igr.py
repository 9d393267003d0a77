"""Stand-in for the external IGR repository's ``model.network.ImplicitNet`` (TEST INFRASTRUCTURE, golden generation only).

The reference loads that class by file path from ``$IGR_PATH/code/model/network.py`` (`sdf_physics/physics3d/utils.py:300-308`)
and trained weights from a download (`README.md:41-42`); neither is on this machine.  This module restates the published
network (Gropp et al. 2020, "Implicit Geometric Regularization"): ``lin0 .. lin{n-1}`` Linear layers over
``dims = [d_in] + dims + [1]``, the layer before a skip index emits ``dims - d_in`` features, ``x = cat([x, input]) /
sqrt(2)`` at a skip layer, Softplus(beta) on all but the last layer.  The weights are the seeded geometric initialisation of
``oracle/igr_oracle.py`` (plain numpy, reproducible on the GPU box), so the rollout goldens generated through the
reference's own ``SDF3D.query_sdfs`` / ``World3D`` pin the STEPPER with a neural SDF body; the trained network itself stays
parity-unpinned.
"""
import numpy as np
import torch
from torch import nn

from .. import igr_oracle


class ImplicitNet(nn.Module):
    def __init__(self, d_in, dims, skip_in=(), geometric_init=True, radius_init=1, beta=100):
        super().__init__()
        dims = [d_in] + list(dims) + [1]
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.d_in = d_in
        for layer in range(self.num_layers - 1):
            out_dim = dims[layer + 1] - d_in if layer + 1 in self.skip_in else dims[layer + 1]
            lin = nn.Linear(dims[layer], out_dim)
            if geometric_init:
                if layer == self.num_layers - 2:
                    nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[layer]), std=1e-5)
                    nn.init.constant_(lin.bias, -radius_init)
                else:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            setattr(self, "lin%d" % layer, lin)
        self.activation = nn.Softplus(beta=beta) if beta > 0 else nn.ReLU()

    def forward(self, inp):
        x = inp
        for layer in range(self.num_layers - 1):
            if layer in self.skip_in:
                x = torch.cat([x, inp], -1) / np.sqrt(2)
            x = getattr(self, "lin%d" % layer)(x)
            if layer < self.num_layers - 2:
                x = self.activation(x)
        return x


def seeded_net(seed=0, radius_init=0.5):
    """The bob_spot_setup network shape (IGR_data/train_configs/bob_spot_setup.conf:38-45) with the numpy-seeded
    geometric-init weights of oracle/igr_oracle.geometric_init, float64."""
    Ws, bs = igr_oracle.geometric_init(seed, radius_init)
    net = ImplicitNet(d_in=5, dims=[128] * 8, skip_in=[4], geometric_init=False, beta=100).double()
    with torch.no_grad():
        for l in range(9):
            lin = getattr(net, "lin%d" % l)
            lin.weight.copy_(torch.tensor(Ws[l]))
            lin.bias.copy_(torch.tensor(bs[l]))
    net.eval()
    return net, Ws, bs
