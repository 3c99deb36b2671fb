#!/bin/bash
# kernel trace of the headline run, printed evaluation by evaluation (one block per objective call):
#   bash profiles/scripts/trace_iters.sh OUTDIR FIRST_NFG LAST_NFG [iter_timeline args]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/$1; A=$2; B=$3; shift 3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- "$PY" profiles/scripts/iter_timeline.py --iters 45 "$@" > $O/timeline.txt 2> $O/trace.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" $A $B <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
a,b=int(sys.argv[2]),int(sys.argv[3])
k=0; prev_end=None
for r in rows:
    nm=r['Kernel_Name']
    if 'obj_quadratic' in nm:
        k+=1
        if a<=k<=b: print('---- evaluation %d' % k)
    if a<=k<=b:
        s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
        gap=(s-prev_end)/1e3 if prev_end else 0.0
        print('  gap %7.1f us  run %8.1f us  %s' % (gap,(e-s)/1e3,nm[:70].replace('void lbk::','')))
    prev_end=int(r['End_Timestamp'])
P
