"""World3D / BatchWorld3D / run_world -- the reference's world surface on top of the HIP engine.

Mirrors sdf_physics/physics3d/world.py:32-205 and lcp_physics/physics/world.py:38-139 for what demos/ and
experiments/ touch: constructor keywords, ``step(fixed_dt)``, ``undo_step()``, ``t``, ``dt``, ``v``, ``set_v``,
``set_p``, ``bodies``, ``contacts``, ``trajectory``, ``observations``, ``run_world``.  One outer step is ONE torch.autograd node (``_StepFn``): its forward runs the
attempt loop of the device engine, its backward runs the reverse tape sweep of csrc/step_bwd.hip, so
``loss.backward()`` reaches body parameters (dims, rad, mass, fric_coeff, restitution, forces) and the initial
state exactly as it does through the reference's Python graph.  The adjoint of the contact geometry, which
crosses step boundaries (contacts found at the end of step k enter the LCP of step k+1), is carried inside the
engine between consecutive backward nodes.
"""
import time

import numpy as np
import torch

from .. import world_abi as abi
from ..engine import BatchEngine, TorchBackend
from .utils import Defaults3D

PARAMS = ("mass", "inertia", "restitution", "fric", "fext", "shape_prm")


class _StepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, world, fixed_dt, pose, vel, mass, inertia, rest, fric, fext, prm, verts=None, mask=None, nsteps=1):
        # verts: the mesh table's vertices [NV,3] as a differentiable input (level-set meshes: their shape gradient flows
        # through the vertex positions); the engine already holds their values, only the adjoint is produced
        E = world.engine
        ctx.has_verts = verts is not None
        # upload only what changed since the last step: parameters are usually the same tensors all along, and the
        # state handed in is usually the very tensor the previous step handed out (the engine still holds its values)
        seen = world.__dict__.setdefault("_uploaded", {})
        for name, t in zip(("pose", "vel") + PARAMS, (pose, vel, mass, inertia, rest, fric, fext, prm)):
            hit = seen.get(name)      # (tensor, version): the reference keeps the tensor alive, so `is` cannot be fooled by id reuse
            if hit is not None and hit[0] is t and hit[1] == t._version:
                continue
            E.arr[name].copy_(t.detach().reshape(E.arr[name].shape))
            seen[name] = (t, t._version)
        nsub0 = E.arr["nsub"].clone()
        # nsteps > 1: that many outer steps per scene in ONE node, the scenes free-running (BatchEngine.run)
        att = E.run(nsteps) if nsteps > 1 else (E.step(mask=mask) if fixed_dt else E.step_once())
        ctx.world, ctx.nsub0, ctx.att = world, nsub0, att
        ctx.nsub1 = E.arr["nsub"].clone()      # tape slots [nsub0, nsub1) belong to this node
        ctx.index = world._n_nodes
        world._n_nodes += 1
        out_pose, out_vel = E.arr["pose"].clone(), E.arr["vel"].clone()
        if not getattr(world, "record_substeps", False):
            ctx.nsubs = 0
            return out_pose, out_vel
        # The accepted sub-steps INSIDE this outer step (dt halving, time-of-contact events), as trajectory entries with a graph:
        # entry k of scene s = the state after its k-th sub-step of this call, read from the tape slot that records the START of
        # the next one; the last sub-step's entry is (out_pose, out_vel) itself.  sub_t = time at the start of each sub-step
        # (the reference stamps its trajectory entries before `self.t += dt`, world.py:373-379).
        cnt = ctx.nsub1 - nsub0                                           # [B] accepted sub-steps of this call
        kmax = max(int(cnt.max().item()) - 1, 0)
        ctx.nsubs = kmax
        B = E.B
        dev = out_pose.device
        ar = torch.arange(B, device=dev)
        ks = torch.arange(kmax + 1, device=dev)[:, None]                  # [kmax+1, 1]
        slot = (nsub0.long()[None, :] + ks).clamp(max=max(int(E.W.max_sub) - 1, 0))
        # (the time stamps are the world times the decide kernel recorded at the start of each sub-step -- the very sums
        # `self.t += dt` forms in the reference: the experiments pair entries with the target entry NEAREST IN TIME, and a stamp
        # midway between two target stamps is a tie that the last bit decides)
        t_start = E.arr["tp_t"][slot, ar[None, :]] if int(E.W.max_sub) > 0 else world._t_before.to(dev)[None, :].expand(kmax + 1, B)
        t_start = torch.where(ks < cnt.long()[None, :], t_start, world._t_before.to(dev)[None, :].expand(kmax + 1, B))
        valid = ks[:kmax] < (cnt.long() - 1)[None, :]                     # intermediate entries (not the last sub-step)
        nxt = (nsub0.long()[None, :] + ks[:kmax] + 1).clamp(max=max(int(E.W.max_sub) - 1, 0))
        if kmax > 0:
            sub_pose = E.arr["tp_pose"][nxt, ar[None, :]].clone()
            sub_vel = E.arr["tp_vel"][nxt, ar[None, :]].clone()
        else:
            sub_pose = out_pose.new_zeros((0,) + tuple(out_pose.shape)); sub_vel = out_vel.new_zeros((0,) + tuple(out_vel.shape))
        last_t = torch.gather(t_start, 0, (cnt.long() - 1).clamp(min=0)[None, :])[0]      # start time of the last sub-step
        t_sub = t_start[:kmax].clone()
        ctx.mark_non_differentiable(valid, t_sub, last_t)
        return out_pose, out_vel, sub_pose, sub_vel, t_sub, valid, last_t

    @staticmethod
    def backward(ctx, g_pose, g_vel, g_sub_pose=None, g_sub_vel=None, *_unused):
        w = ctx.world
        E = w.engine
        adj = E._adjoint()
        ctx_saved = None
        if ctx.index == w._n_nodes - 1 or w._bw_next != ctx.index:   # newest node: start a fresh reverse sweep
            adj["a_geom"].zero_()
            adj["a_last_dt"].zero_()
            # the sweep starts at THIS node's last tape slot: the loss may depend on an intermediate step only, with the
            # world stepped further before backward() (the reference's autograd handles that; later slots are not this
            # node's).  A tape that was rolled back and overwritten since (undo_step + new steps) cannot be differentiated.
            if bool((E.arr["nsub"] < ctx.nsub1).any()):
                raise RuntimeError("backward through a step whose tape slots were released by undo_step()")
            adj["cur_slot"].copy_(ctx.nsub1 - 1)
        else:
            # A scene for which this node is OUT OF THE CHAIN: its step was undone (undo_step) and redone by the next node, yet an
            # output of it is still part of the loss -- the reference's trajectory keeps the first entry of an undone step
            # (world.py:114-116), with its graph.  The sweep of such a scene has already passed these tape slots (the redo wrote
            # the same values into them); it goes over them once more for this node's own incoming gradient, with the contact /
            # time-of-contact adjoints the later steps left for the earlier ones set aside and added back afterwards.
            off = (adj["cur_slot"] != ctx.nsub1 - 1) & (ctx.nsub1 > ctx.nsub0)
            if bool(off.any()):
                ctx_saved = (off, adj["a_geom"][off].clone(), adj["a_last_dt"][off].clone())
                adj["a_geom"][off] = 0.0
                adj["a_last_dt"][off] = 0.0
                adj["cur_slot"][off] = (ctx.nsub1 - 1)[off]
            else:
                ctx_saved = None
        w._bw_next = ctx.index - 1
        for k in ("g_mass", "g_inertia", "g_rest", "g_fric", "g_fext", "g_prm", "g_verts"):
            adj[k].zero_()
        E.A.g_verts = E.be.ptr(adj["g_verts"]) if ctx.has_verts else None
        adj["a_pose"].copy_(g_pose)
        adj["a_vel"].copy_(g_vel)
        first = ctx.index == 0
        adj["lo_slot"].copy_(ctx.nsub0 - (1 if first else 0))   # slot -1: contacts found at construction
        if ctx.nsubs > 0 and g_sub_pose is not None:
            # adjoints of the intermediate entries join the sweep where it passes their state: after slot j has been undone
            # a_pose / a_vel belong to the state at the START of sub-step j = entry (j - 1 - nsub0) of this call
            n0 = ctx.nsub0.long()
            ar = torch.arange(E.B, device=adj["a_pose"].device)
            for _ in range(ctx.att + (1 if first else 0)):
                before = adj["cur_slot"].clone()
                E.backward_sweep(1)
                cur = adj["cur_slot"].long()
                k = cur - n0                                                 # entry whose state the adjoint now describes
                hit = (adj["cur_slot"] != before) & (k >= 0) & (k < ctx.nsubs)
                kk = k.clamp(0, ctx.nsubs - 1)
                adj["a_pose"] += torch.where(hit[:, None, None], g_sub_pose[kk, ar], torch.zeros_like(adj["a_pose"]))
                adj["a_vel"] += torch.where(hit[:, None, None], g_sub_vel[kk, ar], torch.zeros_like(adj["a_vel"]))
        else:
            E.backward_sweep(ctx.att + (1 if first else 0))
        if ctx_saved is not None:
            off, ag, al = ctx_saved
            adj["a_geom"][off] += ag
            adj["a_last_dt"][off] += al
        return (None, None, adj["a_pose"].clone(), adj["a_vel"].clone(), adj["g_mass"].clone(),
                adj["g_inertia"].reshape(E.B, E.nb, 3, 3).clone(), adj["g_rest"].clone(), adj["g_fric"].clone(),
                adj["g_fext"].clone(), adj["g_prm"].clone(), adj["g_verts"].clone() if ctx.has_verts else None, None, None)


class BatchWorld3D:
    """B independent scenes stepped in lock step on one HIP device (the capability BASELINE.json adds).

    ``spec``: BatchEngine spec (see diffsdfsim_amd.scenes).  ``params``: optional dict of torch tensors
    (mass [B,nb], inertia [B,nb,3,3], restitution, fric [B,nb], fext [B,nb,6], shape_prm [B,nb,3]; optionally verts
    [NV,3], the pooled mesh vertices, for shape gradients through level-set meshes) that may require grad; ``pose`` / ``vel`` are the current state tensors (autograd-connected)."""

    def __init__(self, spec, params=None, dt=Defaults3D.DT, eps=Defaults3D.EPSILON, tol=Defaults3D.TOL,
                 fric_dirs=Defaults3D.FRIC_DIRS, strict_no_penetration=True, time_of_contact_diff=True, device=None,
                 max_substeps=1024, maxc=96, max_cand=1024, max_pc=32, stop_contact_grad=False, stop_friction_grad=False,
                 detach_contact_b2=False):
        dev = torch.device(device) if device is not None else Defaults3D.DEVICE
        # the reference's gradient switches (physics3d/world.py:33-37): Jc / Jf built from detached contact geometry, the
        # contact point in body 2's frame a constant -- forward results are the same, the reverse sweep drops those paths
        flags = (1 if stop_contact_grad else 0) | (2 if stop_friction_grad else 0) | (4 if detach_contact_b2 else 0)
        self.engine = BatchEngine(spec, dt=dt, eps=eps, tol=tol, fric_dirs=fric_dirs, maxc=maxc, max_cand=max_cand,
                                  max_pc=max_pc, max_sub=max_substeps, strict_no_pen=strict_no_penetration,
                                  toc_diff=time_of_contact_diff, backend=TorchBackend(dev), grad_flags=flags)
        E = self.engine
        self.device, self.dt = dev, dt
        self.B, self.nb = E.B, E.nb
        self.params = {k: E.arr[k].clone().reshape((E.B, E.nb, 3, 3) if k == "inertia" else E.arr[k].shape) for k in PARAMS}
        for k, v in (params or {}).items():
            self.params[k] = v
        self.pose, self.vel = E.arr["pose"].clone(), E.arr["vel"].clone()
        self._n_nodes, self._bw_next = 0, -1
        self.trajectory = []

    @property
    def t(self):
        return self.engine.get("t")

    _UNDO_ARRAYS = ("t", "nc", "c_body", "c_face", "c_abc", "c_geom", "last_dt", "toc", "nsub")

    def step(self, fixed_dt=True, mask=None, keep_undo=True, nsteps=1):
        """One outer step for every scene, or with `mask` ([B] booleans) for the selected scenes only (the others keep their
        state and time).  Returns [B] booleans: contacts present during the step (World.step's `had_contacts`, world.py:119-139;
        False for scenes the mask left out).  keep_undo=False skips the snapshot `undo_step` needs (nine device copies per step,
        the contact arrays among them) for callers that never undo."""
        E = self.engine
        if nsteps > 1 and (mask is not None or not fixed_dt):
            raise ValueError("run(nsteps) steps every scene with fixed_dt=True: no mask, no single step_dt")
        self._start_batch = (self.pose, self.vel, {k: E.arr[k].clone() for k in self._UNDO_ARRAYS}) if keep_undo else None
        if mask is not None:
            mask = np.asarray(mask.detach().cpu() if torch.is_tensor(mask) else mask).astype(np.int32)
        P = self.params
        dev_cache = self.__dict__.setdefault("_dev_params", {})

        def to(name, x):      # device copy of a parameter, made once per (tensor, version)
            if x.device == self.device:
                return x
            hit = dev_cache.get(name)
            if hit is None or hit[0] is not x or hit[1] != x._version:
                hit = (x, x._version, x.to(self.device))
                dev_cache[name] = hit
            return hit[2]
        if getattr(self, "record_substeps", False):
            self._t_before = torch.as_tensor(E.get("t").copy())
        outs = _StepFn.apply(self, fixed_dt, to("pose", self.pose), to("vel", self.vel), to("mass", P["mass"]),
                             to("inertia", P["inertia"]), to("restitution", P["restitution"]), to("fric", P["fric"]),
                             to("fext", P["fext"]), to("shape_prm", P["shape_prm"]),
                             to("verts", P["verts"]) if P.get("verts") is not None else None, mask, nsteps)
        self.pose, self.vel = outs[0], outs[1]
        # (record_substeps: the entries of the sub-steps inside this call -- poses / velocities [K, B, ...] with their graph, start
        # times [K, B], validity [K, B] -- and the start time of the last sub-step, whose entry is (self.pose, self.vel))
        self.substeps = dict(pose=outs[2], vel=outs[3], t=outs[4], valid=outs[5], last_t=outs[6]) if len(outs) > 2 else None
        up = self.__dict__.setdefault("_uploaded", {})
        up["pose"], up["vel"] = (self.pose, self.pose._version), (self.vel, self.vel._version)
        had = self.engine.get("had_contacts") > 0
        return had if mask is None else had & (mask > 0)      # (the kernel clears the flag of stepped scenes only)

    def run(self, nsteps):
        """`nsteps` outer steps of length dt for every scene as ONE autograd node, the scenes going through their steps
        independently of each other (BatchEngine.run / DssWorld.steps_left: a scene that halves its dt at a bounce holds nobody
        up).  State, tape and gradients are those of `nsteps` calls of step(), bit for bit; what is given up is the per-step
        Python control between them (masks, undo_step, per-step `had_contacts`).  With `record_substeps` the entries of ALL
        accepted sub-steps of the call are in `self.substeps`."""
        nsteps = int(nsteps)
        if nsteps <= 0:
            return
        self.step(nsteps=nsteps, keep_undo=False)

    def undo_step(self, mask):
        """`World.undo_step` (lcp_physics/physics/world.py:106-116) for the scenes `mask` selects: time, poses, velocities
        (with their graph) and contacts go back to the start of the last step; the other scenes keep what they have."""
        if self._start_batch is None:
            raise RuntimeError("undo_step(): the last step was taken with keep_undo=False")
        pose0, vel0, arrs = self._start_batch
        m = torch.as_tensor(np.asarray(mask.detach().cpu() if torch.is_tensor(mask) else mask).astype(bool), device=self.device)
        E = self.engine
        for k, a in arrs.items():
            E.arr[k][m] = a[m]
        self.pose = torch.where(m[:, None, None], pose0.to(self.device), self.pose)
        self.vel = torch.where(m[:, None, None], vel0.to(self.device), self.vel)

    def detach_state(self, mask):
        """Cut the graph of the selected scenes' poses and velocities (optim_sphere.py:170-175: `detach_2nd_bounce`)."""
        m = torch.as_tensor(np.asarray(mask.detach().cpu() if torch.is_tensor(mask) else mask).astype(bool), device=self.device)
        self.pose = torch.where(m[:, None, None], self.pose.detach(), self.pose)
        self.vel = torch.where(m[:, None, None], self.vel.detach(), self.vel)

    def contact_pairs(self, s=0):
        nc = int(self.engine.get("nc")[s])
        return [tuple(r) for r in self.engine.get("c_body")[s][:, :nc].T]


class World3D(BatchWorld3D):
    """Drop-in for ``sdf_physics.physics3d.world.World3D`` (one scene = batch of one)."""

    def __init__(self, bodies, constraints=[], dt=Defaults3D.DT, engine=Defaults3D.ENGINE,
                 contact_callback=Defaults3D.CONTACT, eps=Defaults3D.EPSILON, tol=Defaults3D.TOL,
                 fric_dirs=Defaults3D.FRIC_DIRS, post_stab=Defaults3D.POST_STABILIZATION, strict_no_penetration=True,
                 time_of_contact_diff=True, stop_contact_grad=False, stop_friction_grad=False, detach_contact_b2=False,
                 device=None, max_substeps=1024, full_kernels=None):
        if post_stab:
            raise NotImplementedError("post_stab (engines.py:85-121) is not built on the HIP path")
        # The reference resolves `engine` / `contact_callback` by class or class name and calls them from inside step_dt
        # (lcp_physics/physics/world.py:50-52, 259, 399).  Here one attempt of that loop -- assembly, LCP, integration, contact
        # detection, accept / halve -- is ONE device call (dss_step_attempt), so only the device engine and the device contact
        # handler can sit behind these two keywords; anything else is refused loudly instead of being silently replaced.
        # (The plug-in seams themselves are served on the reference's side: its own World3D with engine=<a class calling
        # dss_solve_dynamics> / contact_callback=<a handler calling dss_find_contacts>, INTEGRATION.md sections 2-3.)
        from . import engines as engines_module
        cls = engine if isinstance(engine, type) else getattr(engines_module, str(engine), None)
        if cls is None:
            raise ValueError("unknown engine %r (known: PdipmEngine / HipPdipmEngine)" % (engine,))
        if not (issubclass(cls, engines_module.HipPdipmEngine) and cls.solve_dynamics is engines_module.HipPdipmEngine.solve_dynamics):
            raise NotImplementedError(
                "World3D(engine=%s): this world steps on the device engine, whose solve / move / detect / accept loop is one call per "
                "attempt; a custom Engine.solve_dynamics(world, dt) cannot be called from inside it.  Use the reference's World3D with "
                "an engine that wraps dss_solve_dynamics (INTEGRATION.md section 2) if a custom engine is needed." % getattr(cls, "__name__", cls))
        cb = contact_callback if isinstance(contact_callback, str) else getattr(contact_callback, "__name__", type(contact_callback).__name__)
        if cb != Defaults3D.CONTACT:
            raise NotImplementedError(
                "World3D(contact_callback=%s): contacts are found by the device narrow phase (the reference's FWContactHandler, "
                "contacts.py:27-272); another handler cannot be called from inside the device step.  Use the reference's World3D with "
                "a handler that wraps dss_find_contacts (INTEGRATION.md section 3)." % cb)
        self.engine_plugin = cls()
        self.bodies = bodies
        self.vec_len = 6
        nb = len(bodies)
        idx = {id(b): i for i, b in enumerate(bodies)}
        rows = []
        for j in constraints:
            J1, _ = j.J()
            blk = torch.zeros(J1.shape[0], 6 * nb, dtype=torch.float64)
            blk[:, 6 * idx[id(j.body1)]:6 * idx[id(j.body1)] + 6] = J1
            rows.append(blk)
        Je = torch.cat(rows).numpy()[None] if rows else np.zeros((1, 0, 6 * nb))
        nocon = np.zeros((nb, nb), np.uint8)
        for i, b in enumerate(bodies):
            for o in b.no_contact:
                nocon[i, idx[id(o)]] = 1
        st = lambda f: torch.stack([f(b) for b in bodies])[None]

        def verts_param():       # pooled mesh vertices as a differentiable input, if any body's mesh depends on a parameter
            vts = [getattr(b, "verts_t", None) for b in bodies]
            if not any(v is not None and v.requires_grad for v in vts):
                return {}
            dev = next(v.device for v in vts if v is not None)
            return dict(verts=torch.cat([v if v is not None else torch.as_tensor(b.verts_np, device=dev) for v, b in zip(vts, bodies)]))
        self._verts_param = verts_param
        self._ptensors = lambda t: dict(
            mass=st(lambda b: b.mass.reshape(())), inertia=st(lambda b: b.ang_inertia),
            restitution=st(lambda b: b.restitution.reshape(())), fric=st(lambda b: b.fric_coeff.reshape(())),
            fext=st(lambda b: b.apply_forces(t)), shape_prm=st(lambda b: b.shape_prm()), **self._verts_param())
        P = self._ptensors(0.0)
        npd = lambda x: x.detach().cpu().numpy()
        spec = dict(pose=npd(st(lambda b: b.p)), vel=npd(st(lambda b: b.v)), mass=npd(P["mass"]), inertia=npd(P["inertia"]),
                    restitution=npd(P["restitution"]), fric=npd(P["fric"]), fext=npd(P["fext"]),
                    shape_type=np.array([[b.shape_type for b in bodies]], np.int32), shape_prm=npd(P["shape_prm"]),
                    shape_aux=np.array([[b.shape_aux() for b in bodies]], np.float64),
                    mesh_id=np.arange(nb, dtype=np.int32)[None], meshes=[(b.verts_np, b.faces_np) for b in bodies],
                    mesh_vgrad=[b.vgrad_np for b in bodies], Je=Je, no_contact=nocon)
        if full_kernels is not None:       # (None: the engine picks the lean kernel variants when every body allows them)
            spec["full_kernels"] = bool(full_kernels)
        grids = [b for b in bodies if b.shape_type == abi.SHAPE_GRID]
        if grids:
            spec["grids"] = [b.sdf.detach().cpu().numpy() for b in grids]
            spec["grid_id"] = np.array([[grids.index(b) if b in grids else -1 for b in bodies]], np.int32)
        nets = {id(b.igr): b.igr for b in bodies if getattr(b, "igr", None) is not None}
        if len(nets) > 1:
            raise NotImplementedError("all neural SDF bodies of a world share one network (one decode_igr(network))")
        if nets:
            spec["igr_net"] = next(iter(nets.values())).packed
        maxc = 32 * max(1, nb - 1)
        # level-set meshes (128^3 marching cubes, triangles of ~1/64 of the body's size) put thousands of faces within
        # eps of a flat neighbour: give the Frank-Wolfe working set room for them
        pinned = {id(j.body1) for j in constraints if j.J()[0].shape[0] == 6}
        dense = any(len(b.faces_np) > 20000 and id(b) not in pinned for b in bodies)
        super().__init__(spec, None, dt, eps, tol, fric_dirs, strict_no_penetration, time_of_contact_diff, device,
                         max_substeps, maxc=4 * maxc if dense else maxc, max_cand=16384 if dense else 1024,
                         max_pc=128 if dense else 32, stop_contact_grad=stop_contact_grad, stop_friction_grad=stop_friction_grad,
                         detach_contact_b2=detach_contact_b2)
        self.pose = st(lambda b: b.p).to(self.device)      # keep the graph to leaf poses / velocities
        self.vel = st(lambda b: b.v).to(self.device)
        self.eps, self.tol, self.fric_dirs = eps, tol, fric_dirs
        self._t = 0.0
        self.observations = []      # filled by experiment code, as in the reference (world.py:66)
        self._start = None
        self._sync_bodies()

    t = property(lambda self: self._t)

    def _sync_bodies(self):
        self._views = []
        for i, b in enumerate(self.bodies):
            b.p, b.v = self.pose[0, i], self.vel[0, i]
            self._views.append((b.p, b.v))

    def _pull_bodies(self):
        """A body whose pose / velocity tensor was replaced from outside (Body3D.set_p, `body.v = ...`) since the last
        synchronisation wins over the world's copy, as in the reference where the bodies own the state."""
        if any(b.p is not vw[0] or b.v is not vw[1] for b, vw in zip(self.bodies, self._views)):
            self.pose = torch.stack([b.p.to(self.device) for b in self.bodies])[None]
            self.vel = torch.stack([b.v.to(self.device) for b in self.bodies])[None]

    @property
    def contacts(self):
        """[((normal, p1, p2, penetration), body1, body2)] as in the reference (values; detached)."""
        E = self.engine
        nc = int(E.get("nc")[0])
        g, b = E.get("c_geom")[0], E.get("c_body")[0]
        out = []
        for c in range(nc):
            tt = lambda a: torch.tensor(a.copy())
            out.append(((tt(g[0:3, c]), tt(g[3:6, c]), tt(g[6:9, c]), tt(g[9, c])), int(b[0, c]), int(b[1, c])))
        return out

    _UNDO_ARRAYS = ("t", "nc", "c_body", "c_face", "c_abc", "c_geom", "last_dt", "toc", "nsub")

    def step(self, fixed_dt=False):
        self.params = self._ptensors(self._t)      # forces may depend on time (ExternalForce3D.force_func)
        self._pull_bodies()
        E = self.engine
        self._start = (self._t, self.pose, self.vel, {k: E.arr[k].clone() for k in self._UNDO_ARRAYS}, self._n_nodes)
        n0, t0 = int(E.get("nsub")[0]), self._t
        had = bool(BatchWorld3D.step(self, fixed_dt, keep_undo=False)[0])      # (World3D keeps its own snapshot, above)
        self._t = float(self.engine.get("t")[0])
        self._sync_bodies()
        # (t, p, v, contacts, joint rotations) as lcp_physics/physics/world.py:373-379 appends them: one entry per ACCEPTED
        # sub-step (step(fixed_dt=True) loops step_dt until the full dt has passed, world.py:119-139), appended BEFORE
        # `self.t += dt` -- an entry carries the time at the START of its sub-step together with the state and contacts AFTER it.
        # The sub-steps inside one outer step are read back from the tape (values: only the state at the end of the call is
        # connected to the graph)
        n1 = int(E.get("nsub")[0])
        t = t0
        if n1 - n0 > 1 and n1 <= int(E.W.max_sub):
            tp, tv, tt = E.get("tp_pose"), E.get("tp_vel"), E.get("tp_t")
            tnc, tb, tg = E.get("tp_nc"), E.get("tp_body"), E.get("tp_geom")
            for j in range(n0, n1 - 1):          # the state after sub-step j = the start of sub-step j + 1
                cs = []
                for c in range(int(tnc[j + 1, 0])):
                    g_, b_ = tg[j + 1, 0], tb[j + 1, 0]
                    tt_ = lambda a: torch.tensor(a.copy())
                    cs.append(((tt_(g_[0:3, c]), tt_(g_[3:6, c]), tt_(g_[6:9, c]), tt_(g_[9, c])), int(b_[0, c]), int(b_[1, c])))
                self.trajectory.append((float(tt[j, 0]), torch.tensor(tp[j + 1, 0].reshape(-1)), torch.tensor(tv[j + 1, 0].reshape(-1)), cs, None))
            t = float(tt[n1 - 1, 0])             # start of the last sub-step (the kernel's own `t`, the sums `self.t += dt` forms)
        self.trajectory.append((t, self.pose[0].reshape(-1), self.vel[0].reshape(-1), self.contacts, None))
        return had

    def get_v(self):
        return self.vel[0].reshape(-1)

    v = property(get_v)

    def set_v(self, new_v):
        """lcp_physics/physics/world.py:384-387: flat [6 nb] generalized velocities."""
        self.vel = new_v.reshape(1, len(self.bodies), 6).to(self.device)
        self._sync_bodies()

    def set_p(self, new_p):
        """sdf_physics/physics3d/world.py:52-54: flat [7 nb] poses (quaternion wxyz + position per body).  As in the
        reference the contact list is not refreshed: it belongs to the pose at the end of the last step."""
        self.pose = new_p.reshape(1, len(self.bodies), 7).to(self.device)
        self._sync_bodies()

    def undo_step(self):
        """lcp_physics/physics/world.py:106-116: back to the state (time, poses, velocities, contacts) at the start
        of the last step; its trajectory entries are dropped."""
        if getattr(self, "_start", None) is None:
            return
        t0, pose0, vel0, arrs, nodes = self._start
        E = self.engine
        for k, a in arrs.items():
            E.arr[k].copy_(a)
        self._t, self.pose, self.vel, self._n_nodes = t0, pose0, vel0, nodes
        self._sync_bodies()
        while self.trajectory and self.trajectory[-1][0] > self._t:
            self.trajectory.pop()
        self._start = None


def run_world(world, fixed_dt=False, animation_dt=None, run_time=10, print_time=True, scene=None, recorder=None, **_render):
    """sdf_physics/physics3d/world.py:113-205 without the pyrender viewer (rendering is out of scope)."""
    start = time.time()
    while world.t < run_time:
        world.step(fixed_dt=fixed_dt)
        if print_time:
            print("\r {} / {} ".format(world.t, time.time() - start), end="")
