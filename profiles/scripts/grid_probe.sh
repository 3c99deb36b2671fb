#!/bin/bash
# which grid the passes over W get from the runtime's occupancy query (LBFGSB_DEBUG prints one line per kernel)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/${1:-gpurun_out/grid}
mkdir -p $O
cd $R
LBFGSB_DEBUG=1 timeout -k 10 120 python3 bench.py --n 2000000 --steps 3 --no-cpu-baseline --no-other-configs --no-live-traffic > $O/bench.json 2> $O/bench.err
echo "rc=$?" > $O/rc.txt
grep "\[grid\]" $O/bench.err > $O/grid.txt
tail -3 $O/bench.err >> $O/rc.txt
cat $O/rc.txt $O/grid.txt
