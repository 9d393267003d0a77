"""Dev aid (GPU box): where the time of constructing a level-set primitive body goes."""
import cProfile
import pstats
import torch
from diffsdfsim_amd.physics3d import SDFSphere
SDFSphere([0, 0, 0], 0.8, custom_mesh=False, custom_inertia=False)      # warm-up (library load, allocator)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
r = torch.tensor(0.9, dtype=torch.float64, requires_grad=True)
b = SDFSphere([0, 0, 0], r, custom_mesh=False, custom_inertia=False)
b.ang_inertia.sum().backward()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
