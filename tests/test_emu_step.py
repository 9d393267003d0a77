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


def test_single_body_contact_free_branch_matches_reference():
    """One body, translation locked by three equality rows, constant torque, no contacts (engines.py:40-54;
    golden: oracle/gen/gen_config5_golden.py).  The torque of the golden is on for t < 0.3 = the first 9 steps."""
    import os
    from diffsdfsim_amd import meshes
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "config5_spin.npz"))
    dims = g["dims"]
    v, f, tie = meshes.box_mesh(dims)
    I = float(g["mass"]) * np.diag([dims[1] ** 2 + dims[2] ** 2, dims[0] ** 2 + dims[2] ** 2, dims[0] ** 2 + dims[1] ** 2]) / 12
    one = lambda a: np.asarray(a, np.float64)[None, None]
    spec = dict(pose=one([1.0, 0, 0, 0, 0, 0, 0]), vel=one(np.zeros(6)), mass=np.full((1, 1), float(g["mass"])), inertia=I[None, None],
                restitution=np.zeros((1, 1)), fric=np.zeros((1, 1)), fext=one(np.concatenate([float(g["mag"]) * g["dir"], np.zeros(3)])),
                shape_type=np.zeros((1, 1), np.int32), shape_prm=one(dims), mesh_id=np.zeros((1, 1), np.int32), meshes=[(v, f)],
                mesh_vgrad=[0.5 * tie], Je=np.concatenate([np.zeros((3, 3)), np.eye(3)], 1)[None], no_contact=np.zeros((1, 1), np.uint8))
    E = BatchEngine(spec, backend=emu.EmuBackend(), max_sub=16, dt=float(g["dt"]))
    for k in range(9):
        E.step()
        assert np.abs(E.get("pose")[0, 0] - g["traj_p"][k]).max() < 1e-10 and np.abs(E.get("vel")[0, 0] - g["traj_v"][k]).max() < 1e-10
    assert (E.get("nc") == 0).all()


@pytest.mark.parametrize("name,nsteps", [("rollout_sphere", 24), ("rollout_boxdrop", 12)])
def test_full_kernel_variants_agree_with_the_lean_ones(name, nsteps):
    """The same goldens through the full variants of the narrow phase and the contact adjoint (every primitive, level-set
    hull handling; selected with spec['full_kernels']): bit-identical state, gradients equal to rounding."""
    g = R.load_rollout(name)
    out = []
    for full in (False, True):
        spec = R.spec_from_golden(g)
        spec["full_kernels"] = full
        E = BatchEngine(spec, backend=emu.EmuBackend(), max_sub=64, **R.engine_kwargs(g))
        assert int(E.W.shape_rare) == int(full)
        R.rollout_and_sweep(E, nsteps)
        out.append((E.get("pose").copy(), E.get("vel").copy(), E.get("nsub").copy(), E.be.to_numpy(E.adj["g_prm"]).copy()))
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a, b)
    # the lean reverse sweep differentiates the contact geometry in reverse mode, the full one in forward mode: the same
    # derivative to rounding
    ga, gb = out[0][3], out[1][3]
    assert np.abs(ga - gb).max() < 3e-8 * np.abs(gb).max()      # (rounding differences of 1e-16 grow through 57 sub-steps)


def test_free_running_scenes_reproduce_lock_step_bit_for_bit():
    """BatchEngine.run(n) (DssWorld.steps_left: a scene that finishes an outer step starts its next one at once) against n calls of
    step() on two sphere drops that reach the floor at different times: state, times, sub-step counts and the whole tape are
    identical bit for bit, in no more attempt rounds."""
    g = R.load_rollout("rollout_sphere")
    out = []
    for free in (False, True):
        spec = R.spec_from_golden(g, 2)
        mv = [b for b in range(spec["pose"].shape[1]) if b not in g["fixed"]][0]
        up = int(np.argmax(np.abs(spec["fext"][0, mv, 3:]))) + 4          # the axis gravity acts along
        spec["pose"][1, mv, up] += 0.031
        E = BatchEngine(spec, backend=emu.EmuBackend(), max_sub=96, **R.engine_kwargs(g))
        rounds = E.run(24) if free else sum(E.step() for _ in range(24))
        out.append((rounds, {k: E.get(k).copy() for k in ("pose", "vel", "t", "nsub", "nc", "c_geom", "tp_pose", "tp_vel", "tp_dt", "tp_t", "tp_nc",
                                                           "tp_lam", "tp_geom", "tp_flags", "last_dt", "toc")}))
    (r0, a), (r1, b) = out
    assert len(set(a["nsub"])) > 1, "the scenes were meant to differ"
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert r1 <= r0, (r1, r0)      # (fewer with more scenes: tests/test_bench_scenes_gpu.py)
