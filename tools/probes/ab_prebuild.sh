# NOTE: the GPMI_PREBUILD_INV2 mode switch this script drives was a temporary patch of enqueue_factor_and_forward (api.hip) and is
# not in the library; the script is kept as the record of the experiment in profiles/HISTORY.md R6.12.
# where the prediction's 512 x 512 inverse blocks are built (GPMI_PREBUILD_INV2: 1 beside the forward sweep on the update
# stream, 2 on the panel stream's 32 CUs, 3 beside the backward sweep, 4 not at all (first predict builds them), 0 not at
# all and the v.v / log det reduction in front of the backward sweep): fit alone, fit + predict, sweep kernel durations
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in 1 2 3 4; do
    echo -n "MODE=$v "; GPMI_PREBUILD_INV2=$v python3 tools/fit_timeline.py 8192 60 | tr '\n' ' '
    GPMI_PREBUILD_INV2=$v python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-configs --no-sharded 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('| headline', round(d['ms_per_step'],3), 'ms/step')"
  done
done
for v in 1 2 3 4; do
  for n in 16384; do
    GPMI_PREBUILD_INV2=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/pre_${v}_$n -o t -- python3 tools/fit_timeline.py $n 12 > /dev/null 2>&1
    f=$(find gpurun_out/r06/pre_${v}_$n -name '*kernel_stats.csv' | head -1)
    echo "MODE=$v N=$n: $(grep trsv_fwd $f | awk -F, '{print "fwd avg ns", $(NF-4)}') $(grep trsv_bwd $f | awk -F, '{print "bwd avg ns", $(NF-4)}')"
  done
done
