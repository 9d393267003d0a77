"""The plug-in keywords of the product World3D (lcp_physics/physics/world.py:50-52: `engine`, `contact_callback` resolved by class
or class name).  One attempt of the device step is a single library call, so nothing but the device engine and the device
contact handler can sit behind them: anything else must be REFUSED, not silently replaced.  (The seams themselves are served
on the reference's side: tests/test_reference_dropin.py.)  The refusals come before any device work and run without a GPU; the
accepted spellings are exercised on the device."""
import pytest
import torch


def _bodies():
    from diffsdfsim_amd.physics3d import Gravity3D, SDFBox, TotalConstraint3D
    floor = SDFBox([0, -0.5, 0], [4.0, 1.0, 4.0], custom_mesh=True, custom_inertia=True)
    b = SDFBox([0.0, 0.3, 0.0], torch.tensor([0.5, 0.4, 0.3], dtype=torch.float64), custom_mesh=True, custom_inertia=True)
    b.add_force(Gravity3D())
    return [floor, b], [TotalConstraint3D(floor)]


def test_custom_engine_is_refused_loudly():
    from diffsdfsim_amd.physics3d import World3D
    from diffsdfsim_amd.physics3d.engines import Engine, HipPdipmEngine

    class MyEngine(Engine):
        def solve_dynamics(self, world, dt):
            return world.get_v()

    class Tweaked(HipPdipmEngine):            # a subclass that overrides the call the device loop could not honour
        def solve_dynamics(self, world, dt):
            return 0.5 * super().solve_dynamics(world, dt)
    bodies, joints = _bodies()
    for eng in (MyEngine, Tweaked):
        with pytest.raises(NotImplementedError, match="engine"):
            World3D(bodies, joints, engine=eng)
    with pytest.raises(ValueError, match="unknown engine"):
        World3D(bodies, joints, engine="NoSuchEngine")


def test_custom_contact_callback_is_refused_loudly():
    from diffsdfsim_amd.physics3d import World3D

    class MyHandler:
        def __call__(self, args, geom1, geom2):
            pass
    bodies, joints = _bodies()
    for cb in (MyHandler, MyHandler(), "OdeContactHandler", "DiffContactHandler"):
        with pytest.raises(NotImplementedError, match="contact_callback"):
            World3D(bodies, joints, contact_callback=cb)


@pytest.mark.gpu
def test_the_reference_spellings_of_the_defaults_are_accepted():
    from diffsdfsim_amd.physics3d import World3D
    from diffsdfsim_amd.physics3d.engines import HipPdipmEngine, PdipmEngine

    class FWContactHandler:                    # the class object spelling (physics3d/world.py:38 passes `.__class__`)
        pass
    for kw in (dict(), dict(engine="PdipmEngine"), dict(engine=PdipmEngine), dict(engine=HipPdipmEngine, contact_callback="FWContactHandler"),
               dict(contact_callback=FWContactHandler)):
        bodies, joints = _bodies()
        w = World3D(bodies, joints, **kw)
        w.step(fixed_dt=True)
        assert isinstance(w.engine_plugin, HipPdipmEngine)
