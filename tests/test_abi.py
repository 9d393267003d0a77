"""CPU: the C-ABI library builds, loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

from helpers import ROOT


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(dss_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from diffsdfsim_amd import _lib
    _lib.build()
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 4
    for s in syms:
        assert hasattr(L, s), "missing export %s" % s
    assert L.dss_abi_version() == _lib.ABI_VERSION


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    from diffsdfsim_amd import _lib
    from diffsdfsim_amd.lcp import LCPFunction
    Q = torch.eye(3, dtype=torch.double)[None]
    with pytest.raises(_lib.HipLibraryError):
        LCPFunction()(Q, torch.zeros(1, 3).double(), torch.ones(1, 2, 3).double(), torch.ones(1, 2).double(),
                      torch.tensor([]), torch.tensor([]), torch.zeros(1, 2, 2).double())
