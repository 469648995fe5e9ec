#!/bin/bash
# Exercise bench.py's multi-rank path on a 1-GPU box: two ranks, both on device 0 (RCCL refuses duplicate
# devices, so the gather falls back to the rendezvous files - the code path the ranks share is the same).
# stdout and stderr are kept apart: rank 0's stdout must hold exactly one line (the JSON).
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 GPMI_RDV_KEY=sim$$
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/sim2_r$r.out 2> gpurun_out/sim2_r$r.err &
done
wait
echo "rank 0 stdout lines: $(wc -l < gpurun_out/sim2_r0.out), rank 1 stdout lines: $(wc -l < gpurun_out/sim2_r1.out)"
cut -c1-160 gpurun_out/sim2_r0.out; echo "--- rank 0 stderr:"; tail -6 gpurun_out/sim2_r0.err | cut -c1-200
