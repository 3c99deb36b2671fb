#!/bin/bash
# rocprofv3 passes over the headline config (n = 1e8, m = 10, fp64, ping-pong entry): kernel stats, then the
# two PMC passes (separate runs, --kernel-trace only) for the HBM traffic of its passes over W.
#   bash profiles/scripts/prof_headline.sh OUTDIR [extra bench args]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/${1:-gpurun_out/headline}
shift || true
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
# the ELF interpreter itself goes after `--` (never a shim script: an exec hop behind the profiler's preload)
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
ARGS="--steps 10 --warmup 12 --no-cpu-baseline --no-other-configs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- "$PY" bench.py $ARGS > $O/bench_under_rocprof.json 2> $O/stats.err
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- "$PY" bench.py $ARGS > $O/b_fetch.json 2> $O/fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- "$PY" bench.py $ARGS > $O/b_write.json 2> $O/write.err
echo done
find $O -name "*.csv" | head -20
