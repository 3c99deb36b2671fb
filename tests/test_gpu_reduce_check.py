"""The workgroup reduction every kernel of the library ends with (device_util.hpp: block_reduce_store -- DPP
reduce-scatter inside the 16-lane rows of a wave, LDS across rows and waves; wave_sum / wave_min / wave_max)
against EXACT results: tests/reduce_check.hip feeds every slot count the library uses (K = 1 ... 192, sums |
minima | maxima in the combinations of its call sites) small integer-valued doubles, so that every sum is exact
whatever the order of the additions, and compares slot by slot, workgroup by workgroup with the host."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_reduction_epilogue_is_exact_on_integer_inputs():
    src = os.path.join(HERE, "reduce_check.hip")
    exe = os.path.join(HERE, "_build", "reduce_check")
    hdr = os.path.join(os.path.dirname(HERE), "lbfgsb_amd", "csrc", "device_util.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                               "-ffp-contract=off", src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "reduce_check ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
