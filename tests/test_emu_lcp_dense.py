"""CPU: kernel LOGIC of csrc/lcp_dense.hip run through the fiber emulator (tests/emu) vs goldens/oracle.

This is not the product path and proves nothing about the GPU build; the parity tests proper are
tests/test_lcp_dense_gpu.py.  It lets the CPU-only container catch indexing / algorithm errors.
"""
import numpy as np
import pytest

from helpers import lcp_goldens, load_lcp, random_lcp, rel
from emu import emu

# (the 400- and 560-row goldens lcp_stack4p3_0 / lcp_stack7_0 take half a minute each under the emulator: on the device only,
#  tests/test_lcp_dense_gpu.py)
SMALL = [p for p in lcp_goldens() if "stack3" not in p and "stack1" not in p and "stack7" not in p and "stack4p3" not in p]


@pytest.mark.parametrize("path", SMALL, ids=lambda p: p.split("/")[-1][:-4])
def test_emulated_kernel_matches_reference_golden(path):
    g = load_lcp(path)
    z, lam, s, nu, it, st = emu.lcp_dense_forward(g["Q"], g["p"], g["G"], g["h"], g["A"], g["b"], g["F"], max_iter=int(g["max_iter"]))
    assert (st == (4 if "inaccurate" in g else 0)).all()      # DSS_LCP_INACCURATE: the INACC_ERR condition (batch.py:165-167)
    assert rel(z, g["zhat"]) < 1e-9
    out = emu.lcp_dense_backward(g["Q"], g["G"], g["A"], g["F"], g["zhat"], g["lam"], g["slack"], g["nu"], g["dl_dz"])
    for name, got in zip("QpGhAbF", out):
        if g["d" + name].size:
            assert rel(got, g["d" + name]) < 1e-7, name


def test_emulated_kernel_matches_oracle_random():
    from oracle import lcp_oracle as O
    Q, p, G, h, A, b, F = random_lcp(3, 2, 8, 9, 2)
    z, lam, s, nu, it, st = emu.lcp_dense_forward(Q, p, G, h, A, b, F)
    zo, lo, so, nuo, ito, sto = O.forward(Q, p, G, h, A, b, F)
    assert (it == ito).all() and rel(z, zo) < 1e-10


def test_emulated_kernel_on_config1_calls():
    from helpers import config1_calls
    for c in config1_calls():
        z, lam, s, nu, it, st = emu.lcp_dense_forward(c["Q"], c["p"], c["G"], c["h"], c["A"], c["b"], c["F"], max_iter=c["max_iter"])
        assert rel(z, c["z"]) < 1e-10


@pytest.mark.parametrize("nB,nz,nineq,neq", [(11, 6, 4, 3), (9, 6, 8, 3), (10, 3, 4, 0), (8, 8, 8, 5), (5, 7, 5, 2)])
def test_eight_lanes_per_system_kernel_matches_oracle_and_the_wave_kernel(nB, nz, nineq, neq, monkeypatch):
    """csrc/lcp_dense_group.hip (nz, nineq, neq <= 8: eight systems per wavefront, matrices row-distributed in registers) against
    the C oracle on random systems -- batch sizes that are not a multiple of eight (padding groups), several groups with different
    iteration counts in one wavefront, no equalities, full 8 x 8 tiles -- forward (iterates, iteration counts, status) and the
    implicit backward; and against the wave-per-system kernel it replaces for these sizes (DSS_LCP_DENSE_WAVE=1)."""
    from oracle import lcp_oracle as O
    Q, p, G, h, A, b, F = random_lcp(11 + nz + nineq, nB, nz, nineq, neq)
    monkeypatch.delenv("DSS_LCP_DENSE_WAVE", raising=False)
    z, lam, s, nu, it, st = emu.lcp_dense_forward(Q, p, G, h, A, b, F)
    zo, lo, so, nuo, ito, sto = O.forward(Q, p, G, h, A, b, F)
    assert (st == sto).all() and (it == ito).all(), (it, ito)
    assert rel(z, zo) < 1e-9 and rel(lam, lo) < 1e-7 and rel(s, so) < 1e-7
    dl = np.random.default_rng(3).standard_normal((nB, nz))
    out = emu.lcp_dense_backward(Q, G, A, F, zo, lo, so, nuo, dl)
    ref = O.backward(Q, G, A, F, zo, lo, so, nuo, dl)
    for name, got, want in zip("QpGhAbF", out, ref):
        if want.size:
            assert rel(got, want) < 1e-8, name
    monkeypatch.setenv("DSS_LCP_DENSE_WAVE", "1")
    zw, lw, sw, nuw, itw, stw = emu.lcp_dense_forward(Q, p, G, h, A, b, F)
    assert (itw == it).all() and rel(zw, z) < 1e-9
