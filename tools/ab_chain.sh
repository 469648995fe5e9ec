#!/bin/bash
# A/B of the chain launch of the flag-ordered tail on ONE box: GPMI_CHAIN_TILES=2 (potrf_diag + fused Tc/Uc launch, round 4)
# against 3 (one launch per column, round 5): parity first, then cfg2 / headline timings and the traced chain step.
cd "$(dirname "$0")/.."
out=gpurun_out/ab_chain; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "flow_tail_is_bit_identical or lost_flag or t32_fit_and_predict or headline_size_against or test_suite_detects" 2>&1 | tail -5 > $out/tests.log
tail -3 $out/tests.log
for rep in 1 2; do
  for mode in 2 3; do
    echo "== GPMI_CHAIN_TILES=$mode (rep $rep)"
    GPMI_CHAIN_TILES=$mode python tools/config_bench.py cfg2 2>&1 | tail -1
    GPMI_CHAIN_TILES=$mode python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['ms_per_step'],2), 'ms/step; flow tail', round(d['roofline']['flow_tail']['ms_per_step'],2), 'ms')"
  done
done
for mode in 2 3; do
  echo "== trace GPMI_CHAIN_TILES=$mode"
  GPMI_CHAIN_TILES=$mode N=8192 bash tools/flow_tr.sh $out/tr$mode 2>&1 | head -12
done
