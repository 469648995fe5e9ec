#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
pt() { echo "== pt $*"; env "$@" timeout 300 python tools/config5_bench.py ${STEPS:-20} ${LAD:-8} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s' % d['lml_evals_per_s'])"; }
c5() { echo "== cfg5/cfg2/cfg4 $*"; env "$@" timeout 300 python tools/config_bench.py cfg2 cfg4 cfg5 2>&1 | tail -6 | cut -c1-150; }
pt GPMI_STREAM_POOL=0
pt GPMI_STREAM_POOL=2
LAD=16 pt GPMI_STREAM_POOL=0
LAD=16 pt GPMI_STREAM_POOL=2
c5 GPMI_STREAM_POOL=0
c5 GPMI_STREAM_POOL=2
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-sharded > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "two ranks rc $?"
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_pool.json 2>gpurun_out/bench_pool.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_pool.json').read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], 'cfg3', d['sharded']['config3']['seconds'], 'cfg5', d['sharded']['config5']['lml_evals_per_s'])
PY
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
