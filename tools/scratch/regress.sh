#!/bin/bash
# the regression battery behind a change of the stream layout
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python tools/scratch/cfg3_alias.py keep 2>&1 | tail -1
timeout 300 python tools/grad_times.py 16384 2>&1 | tail -1
timeout 300 python tools/config_bench.py cfg2 cfg4 2>&1 | tail -2 | cut -c1-140
timeout 300 python tools/config5_bench.py 20 8 2>&1 | tail -1 | cut -c100-230
timeout 300 python tools/config5_bench.py 20 16 2>&1 | tail -1 | cut -c100-230
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-sharded > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "two ranks rc $?"
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_reg.json 2>gpurun_out/bench_reg.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_reg.json').read().strip().splitlines()[-1])
print('headline', round(d['ms_per_step'],3), 'cfg3', round(d['sharded']['config3']['seconds'],3), 'cfg5', round(d['sharded']['config5']['lml_evals_per_s']))
PY
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
