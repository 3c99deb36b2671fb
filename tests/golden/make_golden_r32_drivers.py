#!/usr/bin/env python3
"""Transcripts of the reference's own test programs under its REAL32 build (-DREAL32,
src/lbfgsb_kinds_module.F90:29-37; README.md:23-35): the data the REAL32 Fortran face of the HIP
library is compared with (tests/test_gpu_fortran_drivers.py).

Builds test/driver{1,2,3}.f90 against the reference's own module from the sources where they lie
(`make -C oracle ref_drivers_r32`: outputs under oracle/_ref/, nothing of the reference is copied),
runs them in a scratch directory and stores what they printed:
    tests/golden/ref_outputs/output_r32_{1,2,3}      stdout
    tests/golden/ref_outputs/iterate_r32.dat         driver1's iteration file
Needs /root/reference and amdflang (this container; not the GPU box)."""
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "ref_outputs")


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_drivers_r32"])
    for k in (1, 2, 3):
        exe = os.path.join(ROOT, "oracle", "_ref", "driver%d_r32" % k)
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run([exe], cwd=td, capture_output=True, text=True, timeout=600, check=True)
            open(os.path.join(OUT, "output_r32_%d" % k), "w").write(r.stdout)
            itf = os.path.join(td, "driver1_output.txt")
            if k == 1 and os.path.exists(itf):
                shutil.copy(itf, os.path.join(OUT, "iterate_r32.dat"))
        print("driver%d (REAL32): %d lines" % (k, len(r.stdout.splitlines())))


if __name__ == "__main__":
    main()
