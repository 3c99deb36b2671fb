#!/bin/bash
# plain (unprofiled) bench legs of the two latency-bound shapes + the headline, one JSON line each
#   bash profiles/scripts/bench_small_shapes.sh OUTDIR [extra bench args]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/${1:-gpurun_out/small}
shift || true
mkdir -p $O
cd $R
COMMON="--no-cpu-baseline --no-other-configs --no-live-traffic $*"
python3 bench.py --n 1000000 --steps 40 $COMMON > $O/bench_n1e6.json 2> $O/bench_n1e6.err
python3 bench.py --rows 12500000 --rccl-self --steps 40 $COMMON > $O/bench_n125e5.json 2> $O/bench_n125e5.err
if [ -z "$SKIP_HEADLINE" ]; then python3 bench.py --steps 20 $COMMON > $O/bench_n1e8.json 2> $O/bench_n1e8.err; fi
python3 - "$O" <<'PY'
import json, sys, os
for f in ("bench_n1e6.json", "bench_n125e5.json", "bench_n1e8.json"):
    if not os.path.exists(os.path.join(sys.argv[1], f)):
        continue
    d = json.loads(open(os.path.join(sys.argv[1], f)).read().strip().splitlines()[-1])
    ps = {d["roofline"]["kernel"][:12]: d["roofline"]["avg_launch_ms"]}
    for o in d.get("roofline_other_w_passes", []):
        ps[o["kernel"][:12]] = o["avg_launch_ms"]
    print("%-20s %8.1f it/s  %.4f ms (median %.4f)  syncs %.2f  launches %.2f  %s" % (
        f, d["value"], d["ms_per_step"], d["ms_per_step_median"], d["host_syncs_per_iter"],
        d["kernel_launches_per_iter"], " ".join("%s=%.4f" % kv for kv in ps.items())))
PY
