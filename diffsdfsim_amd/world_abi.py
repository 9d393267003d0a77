"""ctypes mirror of ``DssWorld`` (include/diffsdfsim_hip.h, section B3) and array allocation.

The field order below IS the struct layout; ``dss_world_sizeof()`` is checked against it when the
world is first bound to the library, so a mismatch fails loudly instead of corrupting memory.
"""
import ctypes
import os

import numpy as np

CAND_FIELDS = 28
N_ACTIVE_OVERFLOW = 1 << 30
SHAPE_BOX, SHAPE_SPHERE, SHAPE_CYLINDER, SHAPE_BOX_ROUNDED, SHAPE_BRICK, SHAPE_BOWL, SHAPE_IGR, SHAPE_GRID = 0, 1, 2, 3, 4, 5, 6, 7

_I, _D, _P = ctypes.c_int, ctypes.c_double, ctypes.c_void_p

# (name, kind) kind: 'i' int scalar, 'd' double scalar, 'pd' double*, 'pi' int*, 'pb' uint8*
FIELDS = [
    ("B", "i"), ("nb", "i"), ("neq", "i"), ("maxc", "i"), ("fric_dirs", "i"), ("max_cand", "i"), ("max_pc", "i"),
    ("nmesh", "i"), ("strict_no_pen", "i"), ("toc_diff", "i"), ("lcp_max_iter", "i"), ("shape_rare", "i"), ("grad_flags", "i"),
    ("eps", "d"), ("tol", "d"), ("dt", "d"),
    ("pose", "pd"), ("vel", "pd"),
    ("mass", "pd"), ("inertia", "pd"), ("restitution", "pd"), ("fric", "pd"), ("fext", "pd"),
    ("shape_type", "pi"), ("shape_prm", "pd"), ("shape_aux", "pd"), ("mesh_id", "pi"), ("no_contact", "pb"),
    ("grid_id", "pi"), ("grid_off", "pi"), ("grid_dims", "pi"), ("grid_data", "pd"),
    ("mesh_voff", "pi"), ("mesh_nv", "pi"), ("mesh_foff", "pi"), ("mesh_nf", "pi"),
    ("verts", "pd"), ("faces", "pi"), ("fcent", "pd"), ("frad", "pd"), ("vgrad", "pd"),
    ("fch_box", "pd"), ("vch_box", "pd"), ("mesh_fch_off", "pi"), ("mesh_vch_off", "pi"),
    ("Je", "pd"), ("b_eq", "pd"),
    ("t", "pd"), ("t_end", "pd"), ("dt_try", "pd"), ("last_dt", "pd"), ("dt_use", "pd"),
    ("active", "pi"), ("step_mask", "pi"), ("had_contacts", "pi"), ("steps_left", "pi"), ("toc", "pi"), ("nsub", "pi"), ("n_active", "pi"),
    ("nc", "pi"), ("c_body", "pi"), ("c_face", "pi"), ("c_abc", "pd"), ("c_geom", "pd"),
    ("n_nc", "pi"), ("n_body", "pi"), ("n_face", "pi"), ("n_abc", "pd"), ("n_geom", "pd"),
    ("pose0", "pd"), ("vel0", "pd"),
    ("Mblk", "pd"), ("pvec", "pd"), ("cop", "pd"), ("x", "pd"), ("lam", "pd"), ("slack", "pd"), ("nu", "pd"),
    ("cop_body", "pi"), ("lcp_iters", "pi"), ("lcp_status", "pi"),
    ("ovl", "pi"), ("pair_list", "pi"), ("n_pairs", "pi"), ("invalid", "pi"), ("overflow", "pi"),
    ("pc_count", "pi"), ("pc_stats", "pi"), ("pc_face", "pi"), ("pc_abc", "pd"), ("pc_geom", "pd"),
    ("cand_face", "pi"), ("cand_state", "pi"), ("cand_buf", "pd"),
    ("max_sub", "i"),
    ("tp_pose", "pd"), ("tp_vel", "pd"), ("tp_dt", "pd"), ("tp_x", "pd"), ("tp_lam", "pd"), ("tp_slack", "pd"),
    ("tp_nu", "pd"), ("tp_abc", "pd"), ("tp_geom", "pd"),
    ("tp_nc", "pi"), ("tp_body", "pi"), ("tp_face", "pi"), ("tp_flags", "pi"), ("tp_t", "pd"),
    ("ev_lcp_start", "ev"), ("ev_lcp_stop", "ev"), ("ev_np_start", "ev"), ("ev_np_stop", "ev"), ("dbg_stamps", "ev"),
    # neural SDF bodies: DssIgrNet (six pointers), capacities, the round-based narrow phase's item state and query lists
    ("igr_W0", "pd"), ("igr_b0", "pd"), ("igr_Wp", "pd"), ("igr_bh", "pd"), ("igr_W8", "pd"), ("igr_b8", "pd"),
    ("igr_items_cap", "i"), ("igr_qcap", "i"), ("igr_rounds", "i"),
    ("igr_list", "pi"), ("igr_hdr", "pi"), ("igr_cface", "pi"), ("igr_cstate", "pi"), ("igr_cbuf", "pd"),
    ("igr_qpts", "pd"), ("igr_qlat", "pi"), ("igr_qtag", "pi"), ("igr_qsdf", "pd"), ("igr_qgrad", "pd"), ("igr_qn", "pi"),
    ("igr_hint", "ev"), ("igr_ev", "ev"),
]
IGR_HDR, IGR_ROUNDS = 16, 42


class DssWorld(ctypes.Structure):
    _fields_ = [(n, {"i": _I, "d": _D}.get(k, _P)) for n, k in FIELDS]


NP_DTYPE = {"pd": np.float64, "pi": np.int32, "pb": np.uint8}


def np_slots(B, nb):
    """Scratch slots of the persistent narrow phase = wavefronts of its grid (mirrors dss_np_slots)."""
    return 4 * min(B * nb * (nb - 1), 256 * int(os.environ.get("DSS_NP_WAVES", 3)))     # (env: kernel experiments only)


def igr_shapes(items_cap, qcap, max_cand):
    """Arrays of the round-based narrow phase for neural SDF bodies (narrowphase_igr.hip)."""
    return {
        "igr_list": (items_cap,), "igr_hdr": (items_cap, IGR_HDR), "igr_cface": (items_cap, 3, max_cand),
        "igr_cstate": (items_cap, max_cand), "igr_cbuf": (items_cap, CAND_FIELDS, max_cand),
        "igr_qpts": (4, qcap, 3), "igr_qlat": (4, qcap), "igr_qtag": (4, qcap), "igr_qsdf": (4, qcap), "igr_qgrad": (2, qcap, 3),
        "igr_qn": (2 * (IGR_ROUNDS + 2),),
    }


def array_shapes(B, nb, neq, maxc, fd, max_cand, max_pc, max_sub, nmesh, NV, NF, NFC=1, NVC=1):
    """Shapes of every array the kernels touch (state, scratch, tape)."""
    npair = nb * (nb - 1)
    NR = fd + 2
    NFc = 3 * (1 + fd // 2) + 8
    nz = 6 * nb
    s = {
        "pose": (B, nb, 7), "vel": (B, nb, 6), "mass": (B, nb), "inertia": (B, nb, 9), "restitution": (B, nb),
        "fric": (B, nb), "fext": (B, nb, 6), "shape_type": (B, nb), "shape_prm": (B, nb, 3), "shape_aux": (B, nb), "mesh_id": (B, nb),
        "no_contact": (nb, nb), "mesh_voff": (nmesh,), "mesh_nv": (nmesh,), "mesh_foff": (nmesh,), "mesh_nf": (nmesh,),
        "verts": (NV, 3), "faces": (NF, 3), "fcent": (NF, 3), "frad": (NF,), "vgrad": (NV, 3),
        "fch_box": (NFC, 6), "vch_box": (NVC, 6), "mesh_fch_off": (nmesh,), "mesh_vch_off": (nmesh,),
        "Je": (B, max(neq, 1), nz), "b_eq": (B, max(neq, 1)),
        "t": (B,), "t_end": (B,), "dt_try": (B,), "last_dt": (B,), "dt_use": (B,),
        "active": (B,), "had_contacts": (B,), "toc": (B,), "nsub": (B,), "n_active": (1,),
        "nc": (B,), "c_body": (B, 2, maxc), "c_face": (B, maxc), "c_abc": (B, 3, maxc), "c_geom": (B, 10, maxc),
        "n_nc": (B,), "n_body": (B, 2, maxc), "n_face": (B, maxc), "n_abc": (B, 3, maxc), "n_geom": (B, 10, maxc),
        "pose0": (B, nb, 7), "vel0": (B, nb, 6),
        "Mblk": (B, nb, 6, 6), "pvec": (B, nz), "cop": (B, NFc, maxc), "x": (B, nz), "lam": (B, NR, maxc),
        "slack": (B, NR, maxc), "nu": (B, max(neq, 1)), "cop_body": (B, 2, maxc), "lcp_iters": (B,), "lcp_status": (B,),
        "ovl": (B, nb, nb), "pair_list": (3 * B * npair,), "n_pairs": (8,), "invalid": (B,), "overflow": (B,),
        "pc_count": (B, npair), "pc_stats": (B, npair, 2), "pc_face": (B, npair, max_pc), "pc_abc": (B, npair, 3, max_pc),
        "pc_geom": (B, npair, 10, max_pc),
        "cand_face": (np_slots(B, nb), 2, max_cand), "cand_state": (np_slots(B, nb), max_cand),
        "cand_buf": (np_slots(B, nb), CAND_FIELDS, max_cand),
    }
    if max_sub > 0:
        s.update({
            "tp_pose": (max_sub, B, nb, 7), "tp_vel": (max_sub, B, nb, 6), "tp_dt": (max_sub, B), "tp_x": (max_sub, B, nz),
            "tp_lam": (max_sub, B, NR, maxc), "tp_slack": (max_sub, B, NR, maxc), "tp_nu": (max_sub, B, max(neq, 1)),
            "tp_abc": (max_sub, B, 3, maxc), "tp_geom": (max_sub, B, 10, maxc),
            "tp_nc": (max_sub, B), "tp_body": (max_sub, B, 2, maxc), "tp_face": (max_sub, B, maxc), "tp_flags": (max_sub, B), "tp_t": (max_sub, B),
        })
    return s


ADJ_FIELDS = [
    ("a_pose", "pd"), ("a_vel", "pd"), ("a_geom", "pd"), ("a_last_dt", "pd"), ("a_dt", "pd"),
    ("g_mass", "pd"), ("g_inertia", "pd"), ("g_rest", "pd"), ("g_fric", "pd"), ("g_fext", "pd"), ("g_prm", "pd"), ("g_verts", "pd"),
    ("cur_slot", "pi"), ("lo_slot", "pi"), ("bw_active", "pi"),
    ("a_x", "pd"), ("dMblk", "pd"), ("dpvec", "pd"), ("dcop", "pd"), ("cscr", "pd"), ("bw_nc", "pi"),
    ("igr_bw_n", "pi"), ("igr_bw_idx", "pi"), ("igr_bw_pts", "pd"), ("igr_bw_lat", "pi"), ("igr_bw_sdf", "pd"), ("igr_bw_grad", "pd"),
]


class DssAdjoint(ctypes.Structure):
    _fields_ = [(n, _P) for n, k in ADJ_FIELDS]


def adjoint_shapes(B, nb, maxc, fd, NV=1, igr=False):
    NFc = 3 * (1 + fd // 2) + 8
    cap = B * 2 * maxc
    extra = {"igr_bw_n": (1,), "igr_bw_idx": (B, 2, maxc), "igr_bw_pts": (cap, 3), "igr_bw_lat": (cap,), "igr_bw_sdf": (2, cap),
             "igr_bw_grad": (2, cap, 3)} if igr else {}
    return {
        **extra,
        "a_pose": (B, nb, 7), "a_vel": (B, nb, 6), "a_geom": (B, 10, maxc), "a_last_dt": (B,), "a_dt": (B,),
        "g_mass": (B, nb), "g_inertia": (B, nb, 9), "g_rest": (B, nb), "g_fric": (B, nb), "g_fext": (B, nb, 6),
        "g_prm": (B, nb, 3), "g_verts": (NV, 3), "cur_slot": (B,), "lo_slot": (B,), "bw_active": (B,),
        "a_x": (B, 6 * nb), "dMblk": (B, nb, 36), "dpvec": (B, 6 * nb), "dcop": (B, NFc, maxc),
        "cscr": (B, 56, maxc), "bw_nc": (B,),
    }
