import sys, json
txt=sys.stdin.read()
d=json.loads(txt.split("bench.py detail: ",1)[1].split("\n")[0])
print("OPT", d["config"]["options"], "it/s %.1f ms %.3f syncs %.2f" % (d["value"], d["ms_per_step"], d["host_syncs_per_iter"]))
for r in [d["roofline"]]+d["roofline_other_w_passes"]:
    print("  ", r["kernel"][:40], "%.3f ms in run (%d)  %.3f b2b" % (r["avg_launch_ms"], r["launches_timed"], r["avg_launch_ms_back_to_back"]))
