#!/bin/bash
# rocprofv3 --kernel-trace --stats over one bench shape (no PMC):  bash profiles/scripts/prof_stats.sh OUTDIR [bench args]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/${1:-gpurun_out/stats}
shift || true
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
ARGS="--no-cpu-baseline --no-other-configs --no-live-traffic $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- "$PY" bench.py $ARGS > $O/bench_under_rocprof.json 2> $O/stats.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print('%-90s calls %6s  avg %10.1f us  total %8.2f ms  %5s%%' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
P
