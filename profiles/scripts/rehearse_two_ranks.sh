#!/bin/bash
# bench.py with 2 ranks on ONE GPU (gloo group + the shared-memory stand-in for librccl of the tests):
# how long does the sharded first iteration take?   bash profiles/scripts/rehearse_two_ranks.sh ROWS
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import test_gpu_multirank as tm
print(tm._fake_rccl())
PY
FAKE=$R/tests/_build/libfake_rccl.so
LBFGSB_BENCH_SHARE_GPU=1 LBFGSB_RCCL_LIBRARY=$FAKE MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29877 bench.py --gpus 2 --rows ${1:-20000000} --steps 5 --warmup 12 --no-cpu-baseline 2>/dev/null | \
  python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        b=json.loads(ln); print('rows', b['config']['n'], 'first_iteration_s', b['first_iteration_s'], 'nseg', b['first_iteration_nseg'], 'ms/step', b['ms_per_step'])"
