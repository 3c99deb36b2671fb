"""bench.py prints ONE compact JSON line as the LAST line of stdout (the driver keeps an 8 KB tail of stdout and parses
that line): <= bench.LINE_CAP bytes, with the contract's keys, `roofline` and `cpu_baseline`; everything else goes to
bench_detail.json / stderr.  Checked on a canned full record (round 5's own 23.8 KB line, which the driver could not
keep) -- no GPU needed."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANNED = os.path.join(ROOT, "profiles", "round5_final_bench_n1e8_m10.json")


@pytest.mark.skipif(not os.path.exists(CANNED), reason="canned bench record not in this tree")
def test_compact_line_fits_and_carries_the_contract():
    sys.path.insert(0, ROOT)
    import bench
    out = json.load(open(CANNED))
    assert len(json.dumps(out)) > 20000          # the record the driver lost
    txt = bench.compact_line(out)
    assert len(txt) <= bench.LINE_CAP <= 4096 and "\n" not in txt
    d = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_wtv", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == pytest.approx(out["value"], rel=1e-5)
    for k in ("workload", "n", "m", "entry", "parity_in_run", "rccl_nranks", "first_iteration_s", "legs"):
        assert k in d["config"], k
    assert len(json.dumps(d["config"]["legs"])) <= 1100
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
              "avg_launch_ms"):
        assert k in d["roofline"], k
    assert d["roofline"]["frac"] == pytest.approx(d["roofline"]["achieved"] / d["roofline"]["peak"], rel=1e-4)
    for k in ("achieved", "frac", "avg_launch_ms"):
        assert k in d["roofline_wtv"], k
    for k in ("value", "unit", "cores", "kind", "value_full_size_on_file", "host_cpu_model", "n_sample", "sample"):
        assert k in d["cpu_baseline"], k
    # a multi-GPU record explains itself: the extra keys appear with n_gpus > 1
    out2 = dict(out, n_gpus=8, collective_us=35.0)
    out2["config"] = dict(out["config"], first_iteration_s_per_rank=[0.2] * 8)
    d2 = json.loads(bench.compact_line(out2))
    for k in ("first_iteration_s_per_rank", "collective_us", "ms_per_step_rank_min", "ms_per_step_rank_max",
              "host_syncs_per_iter"):
        assert k in d2["config"], k
    # oversized legs are cut, never the line's cap exceeded
    out3 = dict(out)
    out3["config"] = dict(out["config"], legs={("leg%d" % i): "x" * 300 for i in range(20)})
    assert len(bench.compact_line(out3)) <= bench.LINE_CAP
