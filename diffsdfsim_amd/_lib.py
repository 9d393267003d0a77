"""Loader for the C-ABI HIP library (``include/diffsdfsim_hip.h``).

The product path has NO CPU fallback: if ``libdiffsdfsim_hip.so`` is missing or a tensor is
not on a HIP device the call raises.  ``build()`` compiles the library in-tree with hipcc
for gfx950 (cross-compiles without a GPU), so the ``.so`` travels with the source tree.
"""
import ctypes
import glob
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("DSS_LIB_PATH") or os.path.join(CSRC, "libdiffsdfsim_hip.so")    # (override: kernel experiments, tools/)
_LIB = None

ABI_VERSION = 1


class HipLibraryError(RuntimeError):
    pass


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


# Per-source flags.  The backward kernels differentiate with forward-mode dual numbers whose seeds are compile-time
# constants after unrolling; letting the compiler assume finite values and ignore the sign of zero is what allows it
# to fold the arithmetic on derivative slots that are identically zero (0 * x, x + 0).  Values are not reassociated.
_BWD_FLAGS = ["-fno-honor-nans", "-fno-honor-infinities", "-fno-signed-zeros"]
# Loop strength reduction off: the pass rewrites the kernels' address arithmetic into extra induction variables that live
# across their long loops, in kernels whose speed is set by register pressure (narrow phase 1.24 ->
# 1.20 ms, neural narrow phase and reverse sweep 1-2 %; integer address code only, results bit-identical).  The contact LCP
# keeps the default: it neither gains nor loses, and its register allocation is best left where it is.
_NO_LSR = ["-mllvm", "-disable-lsr"]
_DEFAULT_LLVM = {"lcp_contact.hip": ["-mllvm", "-amdgpu-load-store-vectorizer=0"]}      # (0.726 -> 0.716 ms)
PER_FILE_FLAGS = {"step_bwd.hip": _BWD_FLAGS, "step_bwd_all.hip": _BWD_FLAGS}


def file_flags(name):
    return _DEFAULT_LLVM.get(name, _NO_LSR) + PER_FILE_FLAGS.get(name, [])


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950: every csrc/*.hip -> csrc/_obj/*.o (rebuilt when it or a header is newer),
    linked into csrc/libdiffsdfsim_hip.so."""
    srcs = sources()
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(_HERE, "..", "include", "*.h"))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("DSS_HIPCC_FLAGS", "").split()
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "flags.txt")
    flagsig = " ".join(extra) + repr(sorted(PER_FILE_FLAGS.items())) + repr(_NO_LSR) + repr(sorted(_DEFAULT_LLVM.items())) + "cuid=stem"
    if not os.path.exists(stamp) or open(stamp).read() != flagsig:
        force = True
    hnew = max([os.path.getmtime(h) for h in hdrs] + [os.path.getmtime(__file__)])
    objs, procs = [], []
    for src in srcs:
        name = os.path.basename(src)
        obj = os.path.join(objdir, name[:-4] + ".o")
        objs.append(obj)
        dep = os.path.getmtime(src)
        if name.endswith("_all.hip"):      # second compilation of <base>.hip with every primitive SDF (see narrowphase.hip)
            dep = max(dep, os.path.getmtime(os.path.join(CSRC, name[:-8] + ".hip")))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(dep, hnew):
            continue
        # -cuid=<file stem>: the compilation-unit id is otherwise hashed from the source's absolute path; with a fixed one the
        # library (and the hash the counter files are stamped with) does not depend on where the tree is checked out
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-cuid=dss_" + name[:-4]] + extra + \
            file_flags(name) + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    if procs or not os.path.exists(LIB_PATH) or any(os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(flagsig)
    return LIB_PATH


def lib():
    """The loaded library; raises HipLibraryError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                "%s not found: run `python __graft_entry__.py build` (hipcc, gfx950). "
                "There is no CPU fallback for the product path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.dss_abi_version.restype = ctypes.c_int
        if L.dss_abi_version() != ABI_VERSION:
            raise HipLibraryError("ABI mismatch: library %d, python %d" % (L.dss_abi_version(), ABI_VERSION))
        L.dss_lcp_dense_workspace_bytes.restype = ctypes.c_size_t
        _LIB = L
    return _LIB


def require_device(*tensors):
    for t in tensors:
        if t is not None and t.numel() > 0 and not t.is_cuda:
            raise HipLibraryError("diffsdfsim_amd kernels need tensors on a HIP device (got %s); "
                                  "there is no CPU fallback" % t.device)


def ptr(t):
    """Device pointer of a contiguous tensor as c_void_p (None -> NULL)."""
    if t is None or t.numel() == 0:
        return ctypes.c_void_p(0)
    assert t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check(rc, what):
    if rc != 0:
        raise HipLibraryError("%s failed with code %d" % (what, rc))
