#!/bin/bash
# rocprofv3 kernel traces of the two latency-bound shapes (n = 1e6; 1.25e7 rows with a 1-rank RCCL communicator)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
O=gpurun_out/r4a
mkdir -p $O
COMMON="--no-cpu-baseline --no-other-configs --no-live-traffic"
$PY bench.py --n 1000000 --steps 40 $COMMON > $O/bench_n1e6.json 2> $O/bench_n1e6.err
echo plain 1e6 done
$PY bench.py --rows 12500000 --rccl-self --steps 40 $COMMON > $O/bench_n125e5.json 2> $O/bench_n125e5.err
echo plain 1.25e7 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/n1e6 -- $PY bench.py --n 1000000 --steps 40 $COMMON > $O/bench_n1e6_prof.json 2> $O/n1e6.err
echo prof 1e6 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/n125e5 -- $PY bench.py --rows 12500000 --rccl-self --steps 40 $COMMON > $O/bench_n125e5_prof.json 2> $O/n125e5.err
echo prof 1.25e7 done
find $O -name "*.csv" | head
