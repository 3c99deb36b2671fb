#!/bin/bash
# rocprofv3 passes over BASELINE.json configs[4] (n = 1e8, m = 20, REAL32): kernel stats, then the two
# PMC passes (separate runs, --kernel-trace only) for the HBM traffic of its passes over W.
#   bash profiles/scripts/prof_config5.sh OUTDIR
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/${1:-gpurun_out/config5}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
# the ELF interpreter itself goes after `--` (never a shim script: an exec hop behind the profiler's preload)
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
ARGS="--m 20 --real32 --steps 5 --warmup 22 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- "$PY" bench.py $ARGS > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- "$PY" bench.py $ARGS > $O/b_fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- "$PY" bench.py $ARGS > $O/b_write.json 2> $O/write.err
echo done
