#!/bin/bash
# same-box A/B of the option compact_w (bench.py sets it; --no-compact leaves it off), two rounds, alternating
#   bash profiles/scripts/ab_compact.sh > gpurun_out/ab_compact.txt
B="python bench.py --no-other-configs --no-cpu-baseline --no-live-traffic"
for round in 1 2; do
  for cfg in "--rows 100000000" "--rows 12500000 --rccl-self --steps 60" "--rows 10000000 --rosenbrock --steps 16" "--rows 1000000 --steps 40" "--rows 100000000 --classic --no-defer"; do
    for c in "" "--no-compact"; do
      $B $cfg $c 2>&1 >/dev/null | python profiles/scripts/summ_detail.py | head -3 | tr '\n' ' '
      echo " [$cfg $c]"
    done
  done
done
