#!/bin/bash
# Kernel trace of the lockstep likelihood batch (config 5: 512 evaluations at N = 2048) with the two half-batches on two
# streams (default) and on one (GPMI_BATCH_SPLIT=0: every launch alone on the chip, clean per-launch durations).
# usage: tools/cfg5_trace.sh <outdir under gpurun_out>
out=${1:-gpurun_out/cfg5_trace}
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > $out/run.py <<'PY'
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
import workloads as wl
from inference_amd.gp import GpRegressor
x, y, e = wl.synthetic_dataset(5, 2048, 4)
th = wl.timing_theta(wl.SE, y, 4)
gp = GpRegressor(x, y, y_err=e, hyperpars=th)
thetas = th + 0.05 * np.random.default_rng(0).standard_normal((512, th.size))
gp.marginal_likelihood_batch(thetas)
t0 = time.perf_counter()
for _ in range(3): gp.marginal_likelihood_batch(thetas)
dt = (time.perf_counter() - t0) / 3
print(f"512 LML evaluations: {dt*1e3:.1f} ms = {512/dt:.0f} evals/s")
PY
for split in 1 0; do
  GPMI_BATCH_SPLIT=$split timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/split$split -- python3 $out/run.py > $out/split$split.txt 2>&1
  GPMI_BATCH_SPLIT=$split python3 $out/run.py > $out/split${split}_plain.txt 2>&1
done
find $out -name "*agent_info.csv" -delete
