for o in "" "--opt wgrid=256" "--opt wgrid=512" "--opt wgrid=1024" "--opt pipe=0" "--opt pipe=0 --opt wgrid=512" "--opt pipe=0 --opt wgrid=1536" "--opt nt=1" "--opt fold_finalize=0"; do
  python bench.py --n 1000000 --steps 60 --no-cpu-baseline --no-other-configs --no-live-traffic $o > /tmp/o.json 2>/dev/null
  python - "$o" <<'P'
import json,sys
d=json.loads(open('/tmp/o.json').read().strip().split('\n')[-1])
o=[x for x in d.get('roofline_other_w_passes',[]) if x.get('launches_timed')]
print('%-34s %8.1f it/s  median %.4f ms  store %.4f ms  update %s  syncs %.2f' % (sys.argv[1], d['value'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], ' '.join('%.4f'%x['avg_launch_ms'] for x in o), d['host_syncs_per_iter']))
P
done
