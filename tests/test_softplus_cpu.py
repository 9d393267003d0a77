"""CPU: the hand-written softplus / sigmoid of the network kernel (diffsdfsim_amd/csrc/igr_mlp.hip: softplus100, exp and log1p by
hand) against long double on eight million arguments (tests/emu/check_softplus.cpp compiles the function's own text)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_softplus_of_the_network_kernel_is_accurate_to_a_few_ulp():
    out = os.path.join(HERE, "emu", "_build")
    os.makedirs(out, exist_ok=True)
    src = open(os.path.join(HERE, "..", "diffsdfsim_amd", "csrc", "igr_mlp.hip")).read()
    i0 = src.index("__device__ inline void softplus100(double z, double &h, double &dh)")
    open(os.path.join(out, "softplus_extract.inc"), "w").write(src[i0:src.index("struct Query {")])
    exe = os.path.join(out, "check_softplus")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-I", os.path.join(HERE, "emu"), "-I", os.path.join(HERE, "..", "diffsdfsim_amd", "csrc"),
                           "-I", out, "-o", exe, os.path.join(HERE, "emu", "check_softplus.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
