#!/bin/bash
# same-box A/B of one context option on the headline (n = 1e8) and the per-rank shape (1.25e7 rows, 1-rank RCCL):
#   bash profiles/scripts/ab_option.sh OUTDIR OPTION [values...]      e.g.  ab_option.sh gpurun_out/ab eager_patch 0 1 0 1
# (AB_ROSEN=1: configs[2], extended Rosenbrock n = 1e7, as well)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/$1; OPT=$2; shift 2
mkdir -p $O
cd $R
COMMON="--no-cpu-baseline --no-other-configs --no-live-traffic"
k=0
for v in "$@"; do
  k=$((k+1))
  python3 bench.py --steps 20 $COMMON --opt $OPT=$v > $O/n1e8_${OPT}${v}_$k.json 2> $O/n1e8_${OPT}${v}_$k.err
  python3 bench.py --rows 12500000 --rccl-self --steps 40 $COMMON --opt $OPT=$v > $O/n125e5_${OPT}${v}_$k.json 2> $O/n125e5_${OPT}${v}_$k.err
  if [ -n "$AB_ROSEN" ]; then
    python3 bench.py --rosenbrock --n 10000000 --steps 16 $COMMON --opt $OPT=$v > $O/rosen1e7_${OPT}${v}_$k.json 2> $O/rosen1e7_${OPT}${v}_$k.err
  fi
done
python3 - "$O" <<'PY'
import json, sys, os, glob
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "failed", e); continue
    ps = {d["roofline"]["kernel"][:12]: d["roofline"]["avg_launch_ms"]}
    for o in d.get("roofline_other_w_passes", []):
        ps[o["kernel"][:12]] = o["avg_launch_ms"]
    print("%-28s %8.1f it/s  %.4f ms (median %.4f)  syncs %.2f  launches %.2f  %s" % (
        os.path.basename(f), d["value"], d["ms_per_step"], d["ms_per_step_median"], d["host_syncs_per_iter"],
        d["kernel_launches_per_iter"], " ".join("%s=%.4f" % kv for kv in ps.items() if not kv[0].startswith("cmprlb"))))
PY
