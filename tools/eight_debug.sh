#!/bin/bash
# the 8-ranks-on-one-device bench command of tests/test_gpu_parity.py::test_bench_eight_ranks_on_one_device, stderr kept
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/eight
for i in $(seq 1 ${1:-3}); do
  d=$(mktemp -d)
  BENCH_CFG3_POINTS=8 BENCH_CFG5_LADDERS=8 BENCH_CFG5_STEPS=4 BENCH_CFG3_N=4096 GPMI_RDV_DIR=$d MASTER_PORT=29533 GPMI_DEBUG_INFO=1 \
    python bench.py --gpus 8 --steps 2 --warmup 1 --n 4096 --m 256 > gpurun_out/eight/out_$i.json 2> gpurun_out/eight/err_$i.txt
  echo "run $i rc=$? $(python -c "import json,sys; d=json.loads(open('gpurun_out/eight/out_$i.json').read().strip().splitlines()[-1]); print(d['sharded'].get('error','ok'))")"
  grep -h "\[gpmi\]" gpurun_out/eight/err_$i.txt | sort | uniq -c | head -5
done
