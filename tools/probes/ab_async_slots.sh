# config 5 through the driver (64 ladders x 8 temperatures, 10 steps = bench.py's run) with larger asynchronous slots
for rep in 1 2; do
  echo -n "default (128 per slot, 6 GiB): "; python3 tools/config5_bench.py 10 64 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['lml_evals_per_s']), 'evals/s', d['lml_evaluations'], 'evaluations')"
  for sm in 256 512; do
    echo -n "GPMI_ASYNC_SLOT_MAX=$sm, 40 GiB, 1024 matrices: "; GPMI_ASYNC_SLOT_MAX=$sm GPMI_BATCH_GIB=40 GPMI_BATCH_MAX=1024 python3 tools/config5_bench.py 10 64 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['lml_evals_per_s']), 'evals/s', d['lml_evaluations'], 'evaluations')"
  done
done
echo -n "bare batch 512, default: "; python3 tools/lml_batch_time.py 512 | cut -c1-90
echo -n "bare batch 512, 40 GiB / 1024: "; GPMI_BATCH_GIB=40 GPMI_BATCH_MAX=1024 python3 tools/lml_batch_time.py 512 | cut -c1-90
