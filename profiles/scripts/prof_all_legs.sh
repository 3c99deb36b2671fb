#!/bin/bash
# rocprofv3 --kernel-trace --stats for every shape the bench line quotes, on the build at hand:
# one kernel_stats.csv per leg, copied to OUT/<round>_<leg>_kernel_stats.csv; for the headline also the two PMC
# passes (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only).  The program itself goes after `--`.
#   bash profiles/scripts/prof_all_legs.sh gpurun_out/legs round6
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/${1:-gpurun_out/legs}
TAG=${2:-roundN}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
COMMON="--no-cpu-baseline --no-other-configs --no-live-traffic"
leg() {  # name, bench args
  local name=$1; shift
  rm -rf $O/tmp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -- "$PY" bench.py $COMMON "$@" > $O/${TAG}_${name}_bench_line.json 2> $O/${name}.err
  local f=$(find $O/tmp_$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${TAG}_${name}_kernel_stats.csv
  rm -rf $O/tmp_$name
  echo "== $name: $(python3 -c "import json,sys;d=json.load(open('$O/${TAG}_${name}_bench_line.json'));print('%.1f it/s, %.3f ms' % (d['value'], d['ms_per_step']))")"
  python3 - "$O/${TAG}_${name}_kernel_stats.csv" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print('   %-86s calls %5s  avg %9.1f us  %5s%%' % (r['Name'][:86], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
}
leg headline_n1e8_m10 --steps 20
leg headline_no_compact --steps 20 --no-compact
leg cfg1_n1e6 --rows 1000000 --steps 40
leg cfg2_rosen_n1e7 --rows 10000000 --rosenbrock --steps 16
leg cfg3_rank_shape --rows 12500000 --rccl-self --steps 60
leg cfg4_r32_m20 --m 20 --real32 --steps 16 --warmup 21
leg m48_n2e7 --rows 20000000 --m 48 --steps 10 --warmup 49
# HBM bytes of the headline's passes (PMC, separate runs)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/tmp_pmc
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/tmp_pmc -- "$PY" bench.py $COMMON --steps 6 > /dev/null 2> $O/pmc_$c.err
  f=$(find $O/tmp_pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c > $O/${TAG}_headline_pmc_$c.txt <<'P'
import csv,sys,statistics
per={}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"]==sys.argv[2]:
        per.setdefault(r["Kernel_Name"][:110],[]).append(float(r["Counter_Value"]))
for k,v in sorted(per.items(), key=lambda kv:-sum(kv[1]))[:8]:
    print("%-110s launches %4d  median %14.1f KiB  max %14.1f KiB" % (k,len(v),statistics.median(v),max(v)))
P
  rm -rf $O/tmp_pmc
  cat $O/${TAG}_headline_pmc_$c.txt | head -4
done
