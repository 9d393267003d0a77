"""CPU: stepper kernels (narrow phase, assembly, LCP, integration, accept/halve) through the fiber
emulator against rollouts recorded from the reference (tests/golden/rollout_*.npz)."""
import numpy as np
import pytest

import rollout_helpers as R
from emu import emu
from diffsdfsim_amd.engine import BatchEngine


@pytest.mark.parametrize("name", ["rollout_sphere", "rollout_stack1", "rollout_stack2"])
def test_initial_contacts_match_reference(name):
    g = R.load_rollout(name)
    E = BatchEngine(R.spec_from_golden(g), backend=emu.EmuBackend(), **R.engine_kwargs(g))
    R.check_contacts(E, 0, g["init_body"], g["init_geom"], len(g["init_body"]))


@pytest.mark.parametrize("name,nsteps", [("rollout_sphere", 24), ("rollout_stack1", 2), ("rollout_boxdrop", 12), ("rollout_cylinder", 10)])
def test_rollout_matches_reference(name, nsteps):
    g = R.load_rollout(name)
    E = BatchEngine(R.spec_from_golden(g), backend=emu.EmuBackend(), max_sub=64, **R.engine_kwargs(g))
    for k in range(nsteps):
        E.step()
    nsub = int(E.get("nsub")[0])
    t_ref = g["traj_t"]
    k_ref = int(np.searchsorted(t_ref, E.get("t")[0] - 1e-12)) - 1   # last recorded sub-step before t
    assert nsub == k_ref + 1, (nsub, k_ref)
    assert np.abs(E.get("pose")[0] - g["traj_p"][k_ref]).max() < 1e-7
    assert np.abs(E.get("vel")[0] - g["traj_v"][k_ref]).max() < 1e-7
    R.check_contacts(E, 0, g["traj_body"][k_ref], g["traj_geom"][k_ref], int(g["traj_nc"][k_ref]))
