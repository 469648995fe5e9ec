cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cmd in "tools/config_bench.py cfg4" "tools/config_bench.py cfg2" "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sharded"; do
  for fl in 0 1; do
    GPMI_FLOW=$fl BENCH_NO_PROF=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_try/x -- python3 $cmd > gpurun_out/pmc_try.log 2>&1
    echo "flow=$fl cmd=$cmd rc=$? $(grep -c SIGSEGV gpurun_out/pmc_try.log) $(tail -1 gpurun_out/pmc_try.log | cut -c1-100)"
  done
done
