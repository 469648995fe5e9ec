#!/bin/bash
# stress: N ranks (default 8) of the headline bench on ONE device at the metric's own size (look-ahead regime, flag-ordered tail,
# chain_column_kernel all time-sliced between processes); stderr kept, every non-zero info and every warning counted
cd "$(dirname "$0")/.."
n=${1:-8}
mkdir -p gpurun_out/eight
d=$(mktemp -d)
BENCH_CFG3_POINTS=$n BENCH_CFG5_LADDERS=$n BENCH_CFG5_STEPS=4 GPMI_RDV_DIR=$d MASTER_PORT=29535 GPMI_DEBUG_INFO=1 \
  python bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/eight/full_out.json 2> gpurun_out/eight/full_err.txt
echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/eight/full_out.json').read().strip().splitlines()[-1])
print('n_gpus', d['n_gpus'], 'ms/step', round(d['ms_per_step'], 1), 'sharded:', d['sharded'].get('error', 'ok'),
      {k: round(v['seconds'], 2) for k, v in d['sharded'].items() if isinstance(v, dict)})
PY
echo "[gpmi] lines: $(grep -c '\[gpmi\]' gpurun_out/eight/full_err.txt); warnings: $(grep -ci 'warn' gpurun_out/eight/full_err.txt); timed out: $(grep -c 'timed out' gpurun_out/eight/full_err.txt)"
grep -h "\[gpmi\]\|timed out\|Warning" gpurun_out/eight/full_err.txt | sort | uniq -c | sort -rn | head -8
