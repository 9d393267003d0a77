"""BatchEngine -- host driver of the batched stepper (C ABI section B3).

Owns the device arrays of a batch of B independent scenes (struct-of-arrays over the scene axis),
binds them into a ``DssWorld`` descriptor and runs the reference's ``World.step`` loop
(lcp_physics/physics/world.py:119-139, 241-379) for all scenes in lock step: every *attempt*
(solve -> integrate -> detect -> accept or halve dt) is one ``dss_step_attempt`` call; scenes that
reach the end of the outer step go inactive, the loop ends when none is active.  The only
host<->device traffic per attempt is the 4-byte active-scene counter.

Storage goes through a small backend object (``TorchBackend``: torch tensors on a HIP device, the
product path).  tests/emu supplies a numpy backend bound to the CPU emulation build of the same
kernels for logic tests in the GPU-less build container; nothing in this package refers to it.
"""
import ctypes

import numpy as np

from . import world_abi as abi


class TorchBackend:
    """Device arrays as torch tensors on a HIP device; no CPU fallback."""

    def __init__(self, device="cuda"):
        import torch
        from . import _lib
        self.torch = torch
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise _lib.HipLibraryError("BatchEngine needs a HIP device (got %s); there is no CPU fallback" % device)
        self.lib = _lib.lib()
        self._lib = _lib

    def zeros(self, shape, dtype):
        t = {np.float64: self.torch.float64, np.int32: self.torch.int32, np.uint8: self.torch.uint8}[dtype]
        return self.torch.zeros(shape, dtype=t, device=self.device)

    def from_numpy(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a)).to(self.device)

    def to_numpy(self, t):
        return t.detach().cpu().numpy()

    def ptr(self, t):
        return t.data_ptr()

    def stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def read_int(self, t):
        return int(t.item())


CHUNK = 256   # must equal NT of narrowphase.hip


def default_vgrad(shape_type, prm, verts):
    """d vertex / d shape parameter of the analytic meshes (bodies.py:803-813 box, :1004-1007 sphere).

    Box: only coordinates that ARE +-dims/2 carry a gradient (the reference re-ties the linspace ends),
    d v_k / d dims_k = sign/2.  Sphere: verts = unit * rad, d v / d rad = v / rad."""
    verts = np.asarray(verts, np.float64)
    if shape_type in (abi.SHAPE_BOX_ROUNDED, abi.SHAPE_BRICK, abi.SHAPE_BOWL, abi.SHAPE_IGR, abi.SHAPE_GRID):
        return np.zeros_like(verts)   # level-set / two-parameter meshes: no per-vertex parameter tangent in this layout
    if shape_type == abi.SHAPE_BOX:
        hd = np.asarray(prm, np.float64)[:3] / 2
        return np.where(np.abs(verts) == hd[None, :], 0.5 * np.sign(verts), 0.0)
    if shape_type == abi.SHAPE_CYLINDER:
        rho = np.maximum(np.hypot(verts[:, 0], verts[:, 1]), 1e-300)
        hh = float(prm[1]) / 2
        side = rho > 0.5 * float(prm[0])
        return np.stack([np.where(side, verts[:, 0] / rho, 0.0), np.where(side, verts[:, 1] / rho, 0.0),
                         np.where(verts[:, 2] == hh, 0.5, np.where(verts[:, 2] == -hh, -0.5, 0.0))], axis=1)
    return verts / float(prm[0])


def mesh_table(meshes, vgrads=None):
    """Concatenate (verts, faces) meshes; precompute pose-invariant face centroids and radii."""
    voff, nv, foff, nf, V, F, C, R = [], [], [], [], [], [], [], []
    fch, vch, fch_off, vch_off = [], [], [], []
    nfc = nvc = 0
    vo = fo = 0
    for verts, faces in meshes:
        verts = np.asarray(verts, np.float64); faces = np.asarray(faces, np.int64)
        voff.append(vo); nv.append(len(verts)); foff.append(fo); nf.append(len(faces))
        tri = verts[faces]
        cen = tri.mean(axis=1)
        rad = np.linalg.norm(cen[:, None, :] - tri, axis=2).max(axis=1)
        V.append(verts); F.append(faces.astype(np.int32)); C.append(cen); R.append(rad)
        vo += len(verts); fo += len(faces)
        # culling boxes of runs of 256 consecutive faces / vertices (narrowphase.hip)
        fch_off.append(nfc); vch_off.append(nvc)
        for k in range(0, len(faces), CHUNK):
            c, r = cen[k:k + CHUNK], rad[k:k + CHUNK, None]
            fch.append(np.concatenate([(c - r).min(0), (c + r).max(0)]))
        for k in range(0, len(verts), CHUNK):
            fv = verts[k:k + CHUNK]
            vch.append(np.concatenate([fv.min(0), fv.max(0)]))
        nfc, nvc = len(fch), len(vch)
    i32 = lambda x: np.asarray(x, np.int32)
    out = dict(mesh_voff=i32(voff), mesh_nv=i32(nv), mesh_foff=i32(foff), mesh_nf=i32(nf),
               fch_box=np.stack(fch), vch_box=np.stack(vch), mesh_fch_off=i32(fch_off), mesh_vch_off=i32(vch_off),
               verts=np.concatenate(V), faces=np.concatenate(F), fcent=np.concatenate(C), frad=np.concatenate(R))
    out["vgrad"] = np.concatenate([np.asarray(g, np.float64) for g in vgrads]) if vgrads is not None else np.zeros_like(out["verts"])
    return out


class BatchEngine:
    def __init__(self, spec, dt=1.0 / 30, eps=1e-3, tol=1e-8, fric_dirs=8, maxc=64, max_cand=1024, max_pc=32,
                 max_sub=0, strict_no_pen=True, toc_diff=True, lcp_max_iter=10, backend=None, grad_flags=0):
        """``spec``: numpy arrays pose [B,nb,7], vel [B,nb,6], mass, inertia [B,nb,3,3], restitution, fric,
        fext [B,nb,6], shape_type, shape_prm [B,nb,3], shape_aux [B,nb] (optional), mesh_id [B,nb], meshes [(verts, faces)...],
        no_contact [nb,nb] (optional), Je [B,neq,6nb] (optional)."""
        self.be = backend if backend is not None else TorchBackend()
        pose = np.asarray(spec["pose"], np.float64)
        B, nb = pose.shape[:2]
        Je = np.asarray(spec.get("Je", np.zeros((B, 0, 6 * nb))), np.float64)
        neq = Je.shape[1]
        self.B, self.nb, self.neq, self.maxc, self.fd = B, nb, neq, maxc, fric_dirs
        self.max_sub = max_sub
        vg = spec.get("mesh_vgrad")
        if vg is None:   # derive from the first body that uses each mesh
            mid = np.asarray(spec["mesh_id"]).reshape(B * nb)
            st = np.asarray(spec["shape_type"]).reshape(B * nb)
            sp = np.asarray(spec["shape_prm"], np.float64).reshape(B * nb, 3)
            vg = []
            for m, (verts, _f) in enumerate(spec["meshes"]):
                j = int(np.nonzero(mid == m)[0][0])
                vg.append(default_vgrad(int(st[j]), sp[j], verts))
        mt = mesh_table(spec["meshes"], vg)
        shapes = abi.array_shapes(B, nb, neq, maxc, fric_dirs, max_cand, max_pc, max_sub, len(spec["meshes"]),
                                  len(mt["verts"]), len(mt["faces"]), len(mt["fch_box"]), len(mt["vch_box"]))
        # neural SDF bodies (shape_type 6): network weights + the scratch of the round-based narrow phase
        st_all = np.asarray(spec["shape_type"]).reshape(B, nb)
        igr_b = st_all == abi.SHAPE_IGR
        self.igr_items_cap = self.igr_qcap = 0
        if igr_b.any():
            if spec.get("igr_net") is None:
                raise ValueError("a body of shape_type SHAPE_IGR needs spec['igr_net'] (diffsdfsim_amd.igr.pack_weights)")
            nocon = np.asarray(spec.get("no_contact", np.zeros((nb, nb))), bool)
            per_scene = [sum(2 for i in range(nb) for j in range(i + 1, nb) if (row[i] or row[j]) and not nocon[i, j]) for row in igr_b]
            self.igr_items_cap = max(1, int(sum(per_scene)))
            self.igr_qcap = int(spec.get("igr_qcap", min(self.igr_items_cap * 4096, 1 << 24)))
            shapes.update(abi.igr_shapes(self.igr_items_cap, self.igr_qcap, max_cand))
        kinds = dict(abi.FIELDS)
        self.arr = {}
        for name, shp in shapes.items():
            self.arr[name] = self.be.zeros(shp, abi.NP_DTYPE[kinds[name]])
        if igr_b.any():
            for k in ("W0", "b0", "Wp", "bh", "W8", "b8"):
                w = spec["igr_net"][k]
                self.arr["igr_" + k] = w if not isinstance(w, np.ndarray) and hasattr(w, "data_ptr") else self.be.from_numpy(np.asarray(w, np.float64))
        host = dict(mt)
        host.update(pose=pose, vel=spec["vel"], mass=spec["mass"], inertia=np.asarray(spec["inertia"]).reshape(B, nb, 9),
                    restitution=spec["restitution"], fric=spec["fric"], fext=spec["fext"], shape_type=spec["shape_type"],
                    shape_prm=spec["shape_prm"], mesh_id=spec["mesh_id"])
        if "shape_aux" in spec:
            host["shape_aux"] = spec["shape_aux"]
        host["no_contact"] = np.asarray(spec.get("no_contact", np.zeros((nb, nb))), np.uint8)
        if spec.get("grids"):      # voxel-grid SDF bodies: pooled table of their sample grids
            gs = [np.ascontiguousarray(np.asarray(g, np.float64)) for g in spec["grids"]]
            off = np.cumsum([0] + [g.size for g in gs[:-1]]).astype(np.int32)
            shapes.update(grid_id=(B, nb), grid_off=(len(gs),), grid_dims=(len(gs), 3), grid_data=(int(sum(g.size for g in gs)),))
            host.update(grid_id=np.asarray(spec["grid_id"], np.int32), grid_off=off,
                        grid_dims=np.array([g.shape for g in gs], np.int32), grid_data=np.concatenate([g.reshape(-1) for g in gs]))
        if neq:
            host["Je"] = Je
        for name, a in host.items():
            a = np.ascontiguousarray(np.asarray(a, abi.NP_DTYPE[kinds[name]]).reshape(shapes[name]))
            self.arr[name] = self.be.from_numpy(a)
        W = abi.DssWorld()
        W.B, W.nb, W.neq, W.maxc, W.fric_dirs = B, nb, neq, maxc, fric_dirs
        W.max_cand, W.max_pc, W.nmesh = max_cand, max_pc, len(spec["meshes"])
        W.strict_no_pen, W.toc_diff, W.lcp_max_iter = int(strict_no_pen), int(toc_diff), lcp_max_iter
        W.grad_flags = int(grad_flags)     # 1 stop_contact_grad | 2 stop_friction_grad | 4 detach_contact_b2 (reverse sweep only)
        W.eps, W.tol, W.dt = eps, tol, dt
        # Kernel variant (narrowphase.hip, step_bwd.hip): the lean one knows box / sphere / cylinder and thins contact
        # clusters in LDS; the full one has every primitive and is prepared for level-set meshes (coincident contact
        # points, clusters of thousands of contacts).  Full when a rare primitive or a dense mesh on a free body exists.
        full = spec.get("full_kernels")
        if full is None:
            full = bool((np.asarray(spec["shape_type"]) > abi.SHAPE_CYLINDER).any())      # (includes neural bodies)
            mid = np.asarray(spec["mesh_id"]).reshape(B, nb)
            for b in range(nb):
                pinned = neq >= 6 and (np.abs(Je[0][:, 6 * b:6 * b + 6]).sum(axis=1) > 0).sum() >= 6
                if not pinned and max(len(spec["meshes"][m][1]) for m in set(mid[:, b].tolist())) > 20000:
                    full = True
        W.shape_rare = int(bool(full))
        W.max_sub = max_sub
        W.igr_items_cap, W.igr_qcap, W.igr_rounds = self.igr_items_cap, self.igr_qcap, int(spec.get("igr_rounds", 0))
        for name, kind in abi.FIELDS:
            if kind in ("pd", "pi", "pb"):
                setattr(W, name, self.be.ptr(self.arr[name]) if name in self.arr else None)
        # neural narrow phase: what the previous detection found (host side), which sizes the grids of the next one
        self.igr_hint = None
        import os
        if self.igr_items_cap > 0 and not os.environ.get("DSS_NO_IGR_HINT"):
            self.igr_hint = np.full(2 * (abi.IGR_ROUNDS + 2), -1, np.int32)
            self.igr_hint[0] = self.igr_items_cap
            W.igr_hint = self.igr_hint.ctypes.data
        self.W = W
        L = self.be.lib
        L.dss_world_sizeof.restype = ctypes.c_size_t
        if L.dss_world_sizeof() != ctypes.sizeof(abi.DssWorld):
            raise RuntimeError("DssWorld layout mismatch: library %d bytes, python mirror %d bytes"
                               % (L.dss_world_sizeof(), ctypes.sizeof(abi.DssWorld)))
        if L.dss_np_slots(int(B), int(nb)) != abi.np_slots(int(B), int(nb)):
            raise RuntimeError("narrow-phase slot count mismatch between the library and its python mirror")
        L.dss_lcp_contact_workspace_bytes.restype = ctypes.c_size_t
        nbytes = L.dss_lcp_contact_workspace_bytes(B, nb, neq, maxc, fric_dirs)
        self.lcp_ws = self.be.zeros((nbytes,), np.uint8)
        self.lcp_ws_bytes = nbytes
        self.attempts = 0
        # World.__init__ (world.py:93-100): initial contacts + interpenetration check
        self._set_active(1)
        self._check(L.dss_find_contacts(ctypes.byref(W), self.be.stream()), "dss_find_contacts")
        self._set_active(0)
        if self.get("overflow").any():
            self._raise_overflow()
        if strict_no_pen:
            nc = self.get("nc")
            pen = self.get("c_geom")[:, 9, :]
            for s in range(B):
                if (pen[s, :nc[s]] > tol).any():
                    raise AssertionError("Interpenetration at start (scene %d)" % s)

    # -- helpers ---------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with code %d" % (what, rc))

    def _update_igr_hint(self, first_attempt, step_ends):
        """What the next detection should expect (DssWorld.igr_hint: it sizes launch grids, nothing else).  Within an outer
        step the set of active scenes only shrinks from attempt to attempt, so a retry expects what the attempt before it
        found; the first attempt of a step, with every scene active again, expects what the first attempt of the previous
        step found.  Two small device reads next to the one the attempt loop makes anyway."""
        if self.igr_hint is None:
            return
        now = self.get("igr_qn").astype(np.int32)
        now[0] = int(self.get("n_pairs")[6])
        if first_attempt:
            self._igr_first = now.copy()
        self.igr_hint[:] = self._igr_first if step_ends else now

    def _set_active(self, v):
        a = self.arr["active"]
        a[...] = v

    def get(self, name):
        return self.be.to_numpy(self.arr[name])

    # -- stepping --------------------------------------------------------------------------------
    def step(self, max_attempts=4096, mask=None):
        """One outer step of length dt for every scene (World.step(fixed_dt=True)); with `mask` ([B] of 0/1) only for the
        scenes it selects -- the others keep their state and time (a batch whose scenes are at different times)."""
        L, W = self.be.lib, self.W
        if mask is not None:
            if "step_mask" not in self.arr:
                self.arr["step_mask"] = self.be.zeros((self.B,), np.int32)
            self.arr["step_mask"][...] = self.be.from_numpy(np.asarray(mask, np.int32))
            W.step_mask = self.be.ptr(self.arr["step_mask"])
        else:
            W.step_mask = None
        self._check(L.dss_step_begin(ctypes.byref(W), self.be.stream()), "dss_step_begin")
        n = self.B if mask is None else int(np.asarray(mask, np.int32).sum())
        k = 0
        while n > 0:
            self._check(L.dss_step_attempt(ctypes.byref(W), ctypes.c_void_p(self.be.ptr(self.lcp_ws)),
                                           ctypes.c_size_t(self.lcp_ws_bytes), self.be.stream()), "dss_step_attempt")
            n = self.be.read_int(self.arr["n_active"])
            if n & abi.N_ACTIVE_OVERFLOW:
                self._raise_overflow()
            self._update_igr_hint(k == 0, n == 0)
            k += 1
            if k > max_attempts:
                raise RuntimeError("step did not finish within %d attempts" % max_attempts)
        self.attempts += k
        return k

    def run(self, nsteps, max_attempts=1 << 20):
        """`nsteps` outer steps of length dt for every scene, each scene going through its own `World.step()` calls without
        waiting for the others (DssWorld.steps_left): a scene that halves its dt at a bounce holds nobody up, so the number of
        attempt rounds is the largest per-scene total instead of the sum over steps of the per-step maximum.  State and tape
        are those of `nsteps` calls of step(), bit for bit.  Returns the number of attempt rounds."""
        if nsteps <= 0:
            return 0
        L, W = self.be.lib, self.W
        if "steps_left" not in self.arr:
            self.arr["steps_left"] = self.be.zeros((self.B,), np.int32)
        self.arr["steps_left"][...] = int(nsteps)
        W.step_mask = None
        W.steps_left = self.be.ptr(self.arr["steps_left"])
        try:
            self._check(L.dss_step_begin(ctypes.byref(W), self.be.stream()), "dss_step_begin")
            n, k = self.B, 0
            while n > 0:
                self._check(L.dss_step_attempt(ctypes.byref(W), ctypes.c_void_p(self.be.ptr(self.lcp_ws)),
                                               ctypes.c_size_t(self.lcp_ws_bytes), self.be.stream()), "dss_step_attempt")
                n = self.be.read_int(self.arr["n_active"])
                if n & abi.N_ACTIVE_OVERFLOW:
                    self._raise_overflow()
                self._update_igr_hint(k == 0, n == 0)
                k += 1
                if k > max_attempts:
                    raise RuntimeError("run did not finish within %d attempts" % max_attempts)
        finally:
            W.steps_left = None
        self.attempts += k
        return k

    def _raise_overflow(self):
        ov = self.get("overflow")
        s = int(np.nonzero(ov)[0][0])
        names = [n for b, n in ((1, "max_cand"), (2, "more than 1024 moving Frank-Wolfe candidates -- or, with the lean kernels (spec['full_kernels'] unset/False), "
                                         "contacts of one normal cluster -- in a body pair"), (4, "max_pc"), (8, "maxc"), (16, "max_sub (tape slots for the backward pass)"),
                                        (32, "igr_qcap (query list of the neural narrow phase)"), (64, "igr_rounds"),
                                        (128, "a contact cluster of more than 48 distinct non-coplanar points, whose hull only the full kernel variants take "
                                              "exactly: construct the engine with spec['full_kernels'] = True (World3D(..., full_kernels=True))")) if ov[s] & b]
        raise RuntimeError("contact detection exceeded a capacity in scene %d (%s): raise the limit when constructing "
                           "the engine / world -- contacts were dropped, the step is not valid" % (s, ", ".join(names)))

    def step_once(self, max_attempts=4096):
        """World.step(fixed_dt=False): a single step_dt -- attempts until the first accepted sub-step
        (world.py:136-139).  Scenes of a batch that accept early simply wait (exact for B = 1)."""
        L, W = self.be.lib, self.W
        W.step_mask = None
        self._check(L.dss_step_begin(ctypes.byref(W), self.be.stream()), "dss_step_begin")
        before = self.get("nsub").copy()
        k = 0
        while True:
            self._check(L.dss_step_attempt(ctypes.byref(W), ctypes.c_void_p(self.be.ptr(self.lcp_ws)),
                                           ctypes.c_size_t(self.lcp_ws_bytes), self.be.stream()), "dss_step_attempt")
            k += 1
            if self.get("overflow").any():
                self._raise_overflow()
            self._update_igr_hint(True, True)
            done = self.get("nsub") > before
            a = self.arr["active"]
            a[...] = self.be.from_numpy(np.where(done, 0, self.get("active")).astype(np.int32))
            if done.all():
                break
            if k > max_attempts:
                raise RuntimeError("step_once did not finish within %d attempts" % max_attempts)
        self.attempts += k
        return k

    # -- backward ----------------------------------------------------------------------------------
    def _adjoint(self):
        if getattr(self, "adj", None) is None:
            shp = abi.adjoint_shapes(self.B, self.nb, self.maxc, self.fd, int(self.arr["verts"].shape[0]), igr=self.igr_items_cap > 0)
            kinds = dict(abi.ADJ_FIELDS)
            self.adj = {n: self.be.zeros(s, abi.NP_DTYPE[kinds[n]]) for n, s in shp.items()}
            A = abi.DssAdjoint()
            for n, _k in abi.ADJ_FIELDS:
                setattr(A, n, self.be.ptr(self.adj[n]) if n in self.adj else None)
            self.A = A
            L = self.be.lib
            L.dss_adjoint_sizeof.restype = ctypes.c_size_t
            if L.dss_adjoint_sizeof() != ctypes.sizeof(abi.DssAdjoint):
                raise RuntimeError("DssAdjoint layout mismatch")
        return self.adj

    def backward_sweep(self, n_iter):
        """Run ``n_iter`` reverse sub-step sweeps (each undoes at most one sub-step per scene)."""
        self._adjoint()
        L = self.be.lib
        for _ in range(n_iter):
            self._check(L.dss_step_backward(ctypes.byref(self.W), ctypes.byref(self.A), self.be.stream()), "dss_step_backward")
