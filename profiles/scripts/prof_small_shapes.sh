#!/bin/bash
# rocprofv3 kernel stats (+ traces) of the latency-bound shapes: BASELINE configs[1] (n = 1e6), configs[2]
# (extended Rosenbrock, n = 1e7) and the per-rank shape of configs[3] (1.25e7 rows, 1-rank RCCL communicator)
#   bash profiles/scripts/prof_small_shapes.sh OUTDIR
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/${1:-gpurun_out/small_prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
# the ELF interpreter itself goes after `--` (never a shim script: an exec hop behind the profiler's preload)
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
COMMON="--no-cpu-baseline --no-other-configs --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/n1e6 -- "$PY" bench.py --n 1000000 --steps 40 $COMMON > $O/bench_n1e6_under_rocprof.json 2> $O/n1e6.err
echo "n1e6 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rosen1e7 -- "$PY" bench.py --rosenbrock --n 10000000 --steps 16 $COMMON > $O/bench_rosen1e7_under_rocprof.json 2> $O/rosen1e7.err
echo "rosenbrock 1e7 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/n125e5 -- "$PY" bench.py --rows 12500000 --rccl-self --steps 40 $COMMON > $O/bench_n125e5_under_rocprof.json 2> $O/n125e5.err
echo "1.25e7 done"
find $O -name "*kernel_stats.csv"
