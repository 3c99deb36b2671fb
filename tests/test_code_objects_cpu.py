"""No scratch memory in any kernel of the library: a register that spills to scratch
inside a row loop is reloaded by a vector-memory instruction that returns in order behind the loads in
flight -- one spilled register cost the fp32 m = 20 update pass 3.17 -> 4.68 ms.  Read from
the code objects embedded in the built library (profiles/scripts/kernel_resources.py: the AMDHSA metadata
note of every kernel); runs on the CPU, right after the build.  The only kernels allowed a private segment
are rocPRIM's radix-sort kernels (library code behind the first iteration's full breakpoint sort)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resources():
    spec = importlib.util.spec_from_file_location(
        "kernel_resources", os.path.join(ROOT, "profiles", "scripts", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    return kr, kr.collect([os.path.join(ROOT, "lbfgsb_amd", "liblbfgsb_hip.so")])


def test_no_kernel_of_the_library_uses_scratch():
    kr, rows = _resources()
    assert len(rows) > 500, len(rows)          # the library's kernels were found
    own = [r for r in rows if "rocprim" not in r["kernel"]]
    bad = [(kr.short(r["kernel"]), r["scratch"]) for r in own if r["scratch"] != 0 or r["dyn_stack"] == "true"]
    assert not bad, "kernels with a private (scratch) segment: %s" % bad[:10]
    # the passes of the steady-state iteration, by name: present, gfx950, registers within the file
    hot = [r for r in own if kr.short(r["kernel"]).startswith(("update_scan_kernel<", "subsm_update_kernel<",
                                                                "cmprlb_wtv_kernel<", "cmprlb_wtv_pair_kernel<",
                                                                "wtv_kernel<"))]
    assert len(hot) > 100
    assert all(r["vgpr"] <= 512 for r in hot)


def test_hot_kernels_of_the_headline_config_keep_their_occupancy():
    """the two passes of the n = 1e8, m = 10, fp64 iteration: the storing pass at 3 waves per SIMD, the
    bare W'v kernel at 3 -- a register-count regression shows here before it shows in a bench line"""
    kr, rows = _resources()
    by = {kr.short(r["kernel"]): r for r in rows}
    assert by["subsm_update_kernel<double, 10, true, true, false, false>"]["waves_per_simd"] >= 3
    assert by["wtv_kernel<double, 10, true>"]["waves_per_simd"] >= 3
    assert by["update_scan_kernel<double, 10, true, true, true, false, false, false>"]["scratch"] == 0
    # ... and on the tile-local free-row layout of W (option compact_w): the storing pass and the pair-shared
    # update pass at two waves per SIMD, every accumulator of the latter in the VGPR file (no AGPR traffic)
    assert by["subsm_update_kernel<double, 10, true, true, false, true>"]["waves_per_simd"] >= 2
    up = by["update_scan_kernel<double, 10, true, false, true, true, true, true>"]
    assert up["waves_per_simd"] >= 2 and up["agpr"] == 0 and up["scratch"] == 0
