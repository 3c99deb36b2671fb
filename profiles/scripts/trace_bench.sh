#!/bin/bash
# kernel trace of a bench.py shape, printed evaluation by evaluation (one block per objective-kernel call):
#   bash profiles/scripts/trace_bench.sh OUTDIR FIRST_EVAL LAST_EVAL OBJ_KERNEL_SUBSTRING [bench args]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/$1; A=$2; B=$3; OBJ=$4; shift 4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- "$PY" bench.py --no-cpu-baseline --no-other-configs --no-live-traffic "$@" > $O/bench.json 2> $O/trace.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" $A $B "$OBJ" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
a,b,obj=int(sys.argv[2]),int(sys.argv[3]),sys.argv[4]
k=0; prev_end=None; t0=None
for r in rows:
    nm=r['Kernel_Name']
    if obj in nm:
        k+=1
        if a<=k<=b:
            s=int(r['Start_Timestamp'])
            print('---- evaluation %d   (+%.1f us since the previous one)' % (k, (s-t0)/1e3 if t0 else 0.0)); t0=s
        elif k==a-1: t0=int(r['Start_Timestamp'])
    if a<=k<=b:
        s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
        gap=(s-prev_end)/1e3 if prev_end else 0.0
        print('  gap %7.1f us  run %8.1f us  %s' % (gap,(e-s)/1e3,nm[:70].replace('void lbk::','')))
    prev_end=int(r['End_Timestamp'])
P
