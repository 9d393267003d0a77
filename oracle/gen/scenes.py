"""Scene builders on top of the *reference's* classes (build container only; TEST INFRA).

Used by the golden generators.  Recipes follow SURVEY.md §8d (config 2: sphere drop,
config 3: box stack) at sizes that keep fixtures small.
"""
import torch


def box_stack(nbox=3, seed=0, floor_dims=(4.0, 1.0, 4.0), mu=0.5, rest=0.0, gap=5e-4, requires_grad=True,
              vel_scale=0.0, push=0.0, aligned=False, stacked=None):
    """aligned: unit cubes without yaw or lateral offsets (4 contacts per directed pair, the count SURVEY.md section 6 saw);
    stacked: only the first `stacked` boxes form the tower, the others rest on the floor next to it (needs a wide floor)."""
    from sdf_physics.physics3d.bodies import SDFBox
    from sdf_physics.physics3d.constraints import TotalConstraint3D
    from sdf_physics.physics3d.forces import Gravity3D

    g = torch.Generator().manual_seed(seed)
    floor = SDFBox([0, -floor_dims[1] / 2, 0], list(floor_dims), custom_mesh=True, custom_inertia=True,
                   restitution=rest, fric_coeff=mu)
    bodies, joints, params = [floor], [TotalConstraint3D(floor)], []
    y = 0.0
    for _ in range(nbox):
        dims = 0.9 + 0.2 * torch.rand(3, generator=g, dtype=torch.double)
        if requires_grad:
            dims.requires_grad_()
        off = -0.05 + 0.1 * torch.rand(2, generator=g, dtype=torch.double)
        yaw = 0.2 * torch.rand(1, generator=g, dtype=torch.double).item()
        if aligned:
            dims = torch.ones(3, dtype=torch.double, requires_grad=requires_grad)
            off, yaw = torch.zeros(2, dtype=torch.double), 0.0
        k = len(bodies) - 1
        if stacked is not None and k >= stacked:      # beside the tower, on the floor
            off = torch.tensor([1.5 * (k - stacked + 1) * (1 if k % 2 else -1), 0.0], dtype=torch.double)
            y = 0.0
        yc = y + gap + dims[1].item() / 2
        pos = torch.tensor([0, yaw, 0, off[0].item(), yc, off[1].item()], dtype=torch.double)
        vel = vel_scale * (torch.rand(6, generator=g, dtype=torch.double) - 0.5)
        vel[4] = -abs(vel[4])  # never start by flying apart
        vel[3] += push  # lateral shove so that friction saturates (sliding contacts)
        b = SDFBox(pos, dims, vel=vel, custom_mesh=True, custom_inertia=True, restitution=rest, fric_coeff=mu)
        b.add_force(Gravity3D())
        y += gap + dims[1].item()
        bodies.append(b)
        params.append(dims)
    return bodies, joints, params


def sphere_drop(seed=0, floor_dims=(4.0, 1.0, 4.0), rad=None, y0=None, vx=None, mu=0.25, rest=0.5,
                requires_grad=True):
    from sdf_physics.physics3d.bodies import SDFBox, SDFSphere
    from sdf_physics.physics3d.constraints import TotalConstraint3D
    from sdf_physics.physics3d.forces import Gravity3D

    g = torch.Generator().manual_seed(seed)
    r = 0.4 + 0.2 * torch.rand(1, generator=g, dtype=torch.double)
    y = 0.7 + 0.5 * torch.rand(1, generator=g, dtype=torch.double)
    v = torch.rand(1, generator=g, dtype=torch.double)
    rad_t = torch.tensor(float(r if rad is None else rad), dtype=torch.double, requires_grad=requires_grad)
    y0 = float(y if y0 is None else y0)
    vx = float(v if vx is None else vx)
    floor = SDFBox([0, -floor_dims[1] / 2, 0], list(floor_dims), custom_mesh=True, custom_inertia=True,
                   restitution=rest, fric_coeff=mu)
    ball = SDFSphere([0, y0, 0], rad_t, vel=[0, 0, 0, vx, 0, 0], custom_mesh=True, custom_inertia=True,
                     restitution=rest, fric_coeff=mu)
    ball.add_force(Gravity3D())
    return [floor, ball], [TotalConstraint3D(floor)], [rad_t]


def box_drop(seed=0, floor_dims=(4.0, 1.0, 4.0), height=0.25, mu=0.4, rest=0.3, requires_grad=True):
    """A tilted box dropped onto the floor: impact inside a step => rejected attempts, dt halving, a time-of-contact
    event and (after the bounce) sliding with friction."""
    from sdf_physics.physics3d.bodies import SDFBox
    from sdf_physics.physics3d.constraints import TotalConstraint3D
    from sdf_physics.physics3d.forces import Gravity3D

    g = torch.Generator().manual_seed(seed)
    floor = SDFBox([0, -floor_dims[1] / 2, 0], list(floor_dims), custom_mesh=True, custom_inertia=True,
                   restitution=rest, fric_coeff=mu)
    dims = 0.5 + 0.2 * torch.rand(3, generator=g, dtype=torch.double)
    if requires_grad:
        dims.requires_grad_()
    ang = 0.3 * (torch.rand(3, generator=g, dtype=torch.double) - 0.5)
    pos = torch.tensor([ang[0].item(), ang[1].item(), ang[2].item(), 0.0, height + 0.5 * dims.detach().max().item(), 0.0],
                       dtype=torch.double)
    vel = torch.tensor([0.2, -0.1, 0.3, 0.8, -0.5, 0.2], dtype=torch.double)
    b = SDFBox(pos, dims, vel=vel, custom_mesh=True, custom_inertia=True, restitution=rest, fric_coeff=mu)
    b.add_force(Gravity3D())
    return [floor, b], [TotalConstraint3D(floor)], [dims]


def cylinder_drop(seed=0, floor_dims=(4.0, 1.0, 4.0), mu=0.4, rest=0.3, requires_grad=True):
    """A tilted cylinder dropped onto the floor (curved side + flat caps + rim contacts)."""
    from sdf_physics.physics3d.bodies import SDFBox, SDFCylinder
    from sdf_physics.physics3d.constraints import TotalConstraint3D
    from sdf_physics.physics3d.forces import Gravity3D

    g = torch.Generator().manual_seed(seed)
    floor = SDFBox([0, -floor_dims[1] / 2, 0], list(floor_dims), custom_mesh=True, custom_inertia=True,
                   restitution=rest, fric_coeff=mu)
    rad = torch.tensor(0.25 + 0.1 * torch.rand(1, generator=g, dtype=torch.double).item(), dtype=torch.double, requires_grad=requires_grad)
    height = torch.tensor(0.6 + 0.2 * torch.rand(1, generator=g, dtype=torch.double).item(), dtype=torch.double, requires_grad=requires_grad)
    pos = torch.tensor([1.2, 0.3, 0.1, 0.0, 0.55, 0.0], dtype=torch.double)   # tipped by ~70 degrees about x
    vel = torch.tensor([0.0, 0.2, 0.5, 0.4, -0.3, 0.1], dtype=torch.double)
    c = SDFCylinder(pos, rad, height, vel=vel, custom_mesh=True, custom_inertia=True, restitution=rest, fric_coeff=mu)
    c.add_force(Gravity3D())
    return [floor, c], [TotalConstraint3D(floor)], [rad, height]


def big_box(requires_grad=True, gap=5e-4, mu=0.5, rest=0.0):
    """A wide flat box on the floor: ~800 contact candidates on its bottom face -- more than the wavefront-sized
    scratch of the narrow phase holds, so the pair is worked off by the deferred (workgroup) path."""
    from sdf_physics.physics3d.bodies import SDFBox
    from sdf_physics.physics3d.constraints import TotalConstraint3D
    from sdf_physics.physics3d.forces import Gravity3D

    floor = SDFBox([0, -0.5, 0], [6.0, 1.0, 6.0], custom_mesh=True, custom_inertia=True, restitution=rest, fric_coeff=mu)
    dims = torch.tensor([1.9, 0.5, 1.9], dtype=torch.double, requires_grad=requires_grad)
    pos = torch.tensor([0, 0.1, 0, 0.03, gap + 0.25, -0.02], dtype=torch.double)
    b = SDFBox(pos, dims, vel=[0, 0, 0, 0.6, -0.1, 0.2], custom_mesh=True, custom_inertia=True, restitution=rest, fric_coeff=mu)
    b.add_force(Gravity3D())
    return [floor, b], [TotalConstraint3D(floor)], [dims]


def rounded_drop(kind="rounded", mu=0.4, rest=0.3, requires_grad=False):
    """A tilted SDFBoxRounded / SDFBrick (level-set mesh from the marching-cubes stand-in, inertia from that mesh)
    dropped onto the floor."""
    from sdf_physics.physics3d.bodies import SDFBox, SDFBoxRounded, SDFBrick
    from sdf_physics.physics3d.constraints import TotalConstraint3D
    from sdf_physics.physics3d.forces import Gravity3D

    floor = SDFBox([0, -0.5, 0], [4.0, 1.0, 4.0], custom_mesh=True, custom_inertia=True, restitution=rest, fric_coeff=mu)
    pos = torch.tensor([0.25, 0.1, -0.2, 0.0, 0.5, 0.0], dtype=torch.double)
    vel = torch.tensor([0.3, -0.1, 0.2, 0.6, -0.4, 0.1], dtype=torch.double)
    dims = torch.tensor([0.6, 0.5, 0.7] if kind == "rounded" else [0.7, 0.6, 0.4], dtype=torch.double, requires_grad=requires_grad)
    if kind == "rounded":
        b = SDFBoxRounded(pos, dims, 0.15, vel=vel, restitution=rest, fric_coeff=mu)
    else:
        b = SDFBrick(pos, dims, 0.12, vel=vel, restitution=rest, fric_coeff=mu)
    b.add_force(Gravity3D())
    return [floor, b], [TotalConstraint3D(floor)], ([dims] if requires_grad else [])


# ---- shape-pair and bookkeeping cases that came out of random comparisons against the reference (tools/dbg_fuzz.py) -------
def _T(x, g=False):
    return torch.tensor(x, dtype=torch.double, requires_grad=g)


def _imports():
    from sdf_physics.physics3d.bodies import SDFBox, SDFCylinder, SDFSphere
    from sdf_physics.physics3d.constraints import TotalConstraint3D
    from sdf_physics.physics3d.forces import Gravity3D
    return SDFBox, SDFCylinder, SDFSphere, TotalConstraint3D, Gravity3D


def _floor(mu=0.4, rest=0.3):
    SDFBox = _imports()[0]
    return SDFBox([0, -0.5, 0], [4.0, 1.0, 4.0], custom_mesh=True, custom_inertia=True, restitution=rest, fric_coeff=mu)


def two_spheres():
    """Two spheres rolling into each other on the floor: sphere mesh against sphere SDF."""
    SDFBox, SDFCylinder, SDFSphere, TotalConstraint3D, Gravity3D = _imports()
    f = _floor(); r1, r2 = _T(0.3, True), _T(0.25, True)
    a = SDFSphere([0.0, 0.3005, 0.0], r1, vel=[0, 0, 0, 0.5, 0, 0], custom_mesh=True, custom_inertia=True, restitution=0.3, fric_coeff=0.4)
    b = SDFSphere([0.62, 0.2505, 0.05], r2, vel=[0, 0, 0, -0.8, 0, 0], custom_mesh=True, custom_inertia=True, restitution=0.3, fric_coeff=0.4)
    for x in (a, b):
        x.add_force(Gravity3D())
    return [f, a, b], [TotalConstraint3D(f)], [r1, r2]


def sphere_on_box():
    """A sphere dropped on a box that rests on the floor (three bodies, box-floor face contact with mid-edge candidates)."""
    SDFBox, SDFCylinder, SDFSphere, TotalConstraint3D, Gravity3D = _imports()
    f = _floor(); d = _T([0.8, 0.4, 0.7], True); r = _T(0.2, True)
    b = SDFBox([0.0, 0.2005, 0.0], d, custom_mesh=True, custom_inertia=True, restitution=0.2, fric_coeff=0.4)
    s = SDFSphere([0.1, 0.4 + 0.2 + 0.15, 0.05], r, vel=[0, 0, 0, 0.3, -0.5, 0.1], custom_mesh=True, custom_inertia=True, restitution=0.2, fric_coeff=0.4)
    for x in (b, s):
        x.add_force(Gravity3D())
    return [f, b, s], [TotalConstraint3D(f)], [d, r]


def floor_last():
    """The pinned body is not body 0 (the LCP's closed-form elimination of a pinned leading body does not apply)."""
    SDFBox, SDFCylinder, SDFSphere, TotalConstraint3D, Gravity3D = _imports()
    r = _T(0.3, True)
    s = SDFSphere([0.0, 0.5, 0.0], r, vel=[0, 0, 2.0, 0.6, 0, 0], custom_mesh=True, custom_inertia=True, restitution=0.4, fric_coeff=0.5)
    s.add_force(Gravity3D())
    f = _floor(0.5, 0.4)
    return [s, f], [TotalConstraint3D(f)], [r]


def no_contact_pair():
    """Two interpenetrating boxes that ignore each other (Body.add_no_contact), both resting on the floor."""
    SDFBox, SDFCylinder, SDFSphere, TotalConstraint3D, Gravity3D = _imports()
    f = _floor(); d1, d2 = _T([0.5, 0.4, 0.5], True), _T([0.4, 0.6, 0.4], True)
    a = SDFBox([0.0, 0.2005, 0.0], d1, vel=[0, 0, 0, 0.3, 0, 0], custom_mesh=True, custom_inertia=True, restitution=0.1, fric_coeff=0.4)
    b = SDFBox([0.1, 0.3005, 0.05], d2, vel=[0, 0, 0, -0.2, 0, 0.1], custom_mesh=True, custom_inertia=True, restitution=0.1, fric_coeff=0.4)
    a.add_no_contact(b)
    for x in (a, b):
        x.add_force(Gravity3D())
    return [f, a, b], [TotalConstraint3D(f)], [d1, d2]


def levelset_box(mu=0.4, rest=0.1, requires_grad=False):
    """A box with the reference's DEFAULT mesh and inertia (custom_mesh = custom_inertia = False: 128^3 marching cubes,
    volume integrals) set down flat on the floor with a small sideways velocity: every face of its bottom side is a
    contact candidate with the same normal (one cluster of thousands of points for the thinning stage)."""
    SDFBox, SDFCylinder, SDFSphere, TotalConstraint3D, Gravity3D = _imports()
    f = _floor(mu, rest)
    d = _T([0.6, 0.4, 0.5], requires_grad)
    b = SDFBox([0.0, 0.2 + 5e-4, 0.0], d, vel=[0, 0, 0, 0.4, 0, 0.1], restitution=rest, fric_coeff=mu)
    b.add_force(Gravity3D())
    return [f, b], [TotalConstraint3D(f)], ([d] if requires_grad else [])


def rounded_rest(mu=0.4, rest=0.1):
    """A level-set rounded box set down flat on the floor: one normal cluster of several thousand contact points that is
    almost, but not quite, flat (the resting face plus the first rows of its rounded rim) -- Qhull's 3-D hull keeps a
    few dozen of them."""
    from sdf_physics.physics3d.bodies import SDFBoxRounded
    SDFBox, SDFCylinder, SDFSphere, TotalConstraint3D, Gravity3D = _imports()
    f = _floor(mu, rest)
    b = SDFBoxRounded([0.0, 0.2 + 5e-4, 0.0], _T([0.8, 0.4, 0.7]), 0.1, vel=[0, 0.3, 0, 0.4, 0, 0.1], restitution=rest, fric_coeff=mu)
    b.add_force(Gravity3D())
    return [f, b], [TotalConstraint3D(f)], []


def levelset_sphere(requires_grad=True):
    """A sphere with the reference's default (level-set) mesh and integrated inertia dropped with spin: gradient w.r.t. its
    radius through the SDF, the mesh scale and the inertia."""
    SDFBox, SDFCylinder, SDFSphere, TotalConstraint3D, Gravity3D = _imports()
    f = _floor(); r = _T(0.3, requires_grad)
    s = SDFSphere([0.0, 0.45, 0.0], r, vel=[0, 0, 1.0, 0.5, -0.5, 0], restitution=0.3, fric_coeff=0.4)
    s.add_force(Gravity3D())
    return [f, s], [TotalConstraint3D(f)], [r]


def levelset_cylinder(requires_grad=True):
    """A level-set cylinder lying on the floor and rolling: line contact on a curved level-set surface (one normal cluster
    of ~850 contact points, ~340 distinct, in rows along the length of which the hull keeps the ends)."""
    SDFBox, SDFCylinder, SDFSphere, TotalConstraint3D, Gravity3D = _imports()
    f = _floor(); r, h = _T(0.25, requires_grad), _T(0.6, requires_grad)
    c = SDFCylinder([0.0, 0.2505, 0.0], r, h, vel=[0, 0, 1.0, 0.5, 0, 0], restitution=0.1, fric_coeff=0.3)
    c.add_force(Gravity3D())
    return [f, c], [TotalConstraint3D(f)], [r, h]


def fast_sphere(vy=-60.0, y0=1.2, rad=0.5, floor_dims=(4.0, 1.0, 4.0), mu=0.25, rest=0.5, requires_grad=True):
    """A sphere thrown at the floor so fast that no halving of dt lands it inside the contact band (eps = 1e-3) without
    penetrating (> tol): with strict_no_penetration=False the reference gives up halving once dt < dt/2^10 and goes on with
    the penetrating contacts as they are (world.py:345-347)."""
    from sdf_physics.physics3d.bodies import SDFBox, SDFSphere
    from sdf_physics.physics3d.constraints import TotalConstraint3D
    from sdf_physics.physics3d.forces import Gravity3D
    rad_t = torch.tensor(float(rad), dtype=torch.double, requires_grad=requires_grad)
    floor = SDFBox([0, -floor_dims[1] / 2, 0], list(floor_dims), custom_mesh=True, custom_inertia=True, restitution=rest, fric_coeff=mu)
    ball = SDFSphere([0, y0, 0], rad_t, vel=[0, 0, 0, 0.3, vy, 0], custom_mesh=True, custom_inertia=True, restitution=rest, fric_coeff=mu)
    ball.add_force(Gravity3D())
    return [floor, ball], [TotalConstraint3D(floor)], [rad_t]


def grid_body_drop(n=48, rad=0.5, scale=0.8, requires_grad=True):
    """A voxel-grid SDF body (SDFGrid3D, bodies.py:763-775): the samples of a sphere's SDF (radius `rad` in the unit cube)
    on n^3 points, dropped onto the floor with some spin and lateral speed.  The differentiable quantity is its start velocity
    (a grid has no shape parameter)."""
    from sdf_physics.physics3d.bodies import SDFBox, SDFGrid3D
    from sdf_physics.physics3d.constraints import TotalConstraint3D
    from sdf_physics.physics3d.forces import Gravity3D
    g = torch.linspace(-1.0, 1.0, n, dtype=torch.double)
    X, Y, Z = torch.meshgrid(g, g, g, indexing="ij")
    sdf = torch.sqrt(X * X + Y * Y + Z * Z) - rad
    floor = SDFBox([0, -0.5, 0], [4.0, 1.0, 4.0], custom_mesh=True, custom_inertia=True, restitution=0.3, fric_coeff=0.4)
    vel = torch.tensor([0.0, 0.0, 0.8, 0.5, -0.6, 0.0], dtype=torch.double, requires_grad=requires_grad)
    body = SDFGrid3D([0.0, rad * scale + 0.02, 0.0], scale, sdf, vel=vel, restitution=0.3, fric_coeff=0.4)
    body.add_force(Gravity3D())
    return [floor, body], [TotalConstraint3D(floor)], [vel]
