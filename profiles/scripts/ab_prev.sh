#!/bin/bash
# A/B of the library in the tree against a second build (lbfgsb_amd/liblbfgsb_hip_prev.so: the previous commit's
# sources, built by hand) on ONE box: bench legs without cpu baseline / traffic / other configs, alternating.
# usage: ab_prev.sh OUTDIR  [bench args...]
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export LBFGSB_HIP_LIBRARY=$PWD/lbfgsb_amd/liblbfgsb_hip_prev.so; else unset LBFGSB_HIP_LIBRARY; fi
    python bench.py --no-cpu-baseline --no-other-configs --no-live-traffic "$@" > $out/${which}_$rep.json 2> $out/${which}_$rep.err || exit 1
  done
done
python - $out <<'P'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    d=json.loads(open(f).read().strip().split('\n')[-1])
    o=[x for x in d.get('roofline_other_w_passes',[]) if x.get('launches_timed')]
    print(os.path.basename(f), 'it/s %.2f ms %.3f | store pass %.3f ms %.3f | update pass %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], ' '.join('%.3f ms %.3f'%(x['avg_launch_ms'],x['frac']) for x in o)))
P
