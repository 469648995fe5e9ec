#!/bin/bash
# One GPU session that produces every profile artefact of a round under gpurun_out/prof_<tag>/ ; copy into profiles/ with
#   python tools/make_profiles.py <tag>
# usage (on the GPU box, through gpurun):  tools/profile_round.sh r02
tag=${1:-r03}
out=gpurun_out/prof_$tag
rm -rf $GRAFT_REPO_ROOT/$out
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# 1. headline: kernel trace + stats of the bench command, then the bench line itself (un-profiled)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-sharded --no-configs > $out/bench_traced.json 2> $out/bench_traced.err
python3 tools/timeline_full.py $out/bench $out/bench_timeline.txt > /dev/null 2>&1
python3 tools/by_queue.py $out/bench $out/bench_kernel_stats_by_queue.csv > /dev/null 2>&1
timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
# 2. the other BASELINE configurations and the LML-gradient path: per-kernel stats of the same public calls
for cfg in cfg2 cfg3 cfg4 cfg5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$cfg -- python3 tools/config_bench.py $cfg > $out/$cfg.txt 2> $out/$cfg.err
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/lmlgrad -- python3 tools/grad_times.py 16384 > $out/lmlgrad.txt 2> $out/lmlgrad.err
# (the driver's rate un-traced - rocprofv3's per-launch cost lands on the host thread that the driver's bookkeeping shares -,
# the per-kernel stats from a traced run of the same command)
timeout 300 python3 tools/config5_bench.py 50 > $out/pt.json 2> $out/pt_plain.err
timeout 300 python3 tools/config5_bench.py 20 16 > $out/pt16.json 2>> $out/pt_plain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pt -- python3 tools/config5_bench.py 10 > $out/pt_traced.json 2> $out/pt.err
timeout 300 python3 tools/propose_bench.py 4096 32 > $out/propose.json 2> $out/propose.err
timeout 300 python3 tools/search_time.py > $out/search.json 2> $out/search.err
# keep the stats, drop the bulky traces (gpurun_out is capped at 64 MiB)
find $out -name "*kernel_trace.csv" -delete
find $out -name "*agent_info.csv" -delete
# 3. HBM traffic / MFMA counters: separate --pmc passes (tools/pmc_run.sh) of the headline command, of the LML gradient at
#    N = 16384 (the fused contraction, the k-skipped SYRK) and of config 5's lockstep batches (potrf_diag, the batched updates)
tools/pmc_run.sh $out/pmc_bench python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sharded --no-configs > $out/pmc_bench.json 2> $out/pmc_bench.err
tools/pmc_run.sh $out/pmc_grad python3 tools/grad_times.py 16384 > $out/pmc_grad.json 2> $out/pmc_grad.err
tools/pmc_run.sh $out/pmc_cfg5 python3 tools/config5_bench.py 4 > $out/pmc_cfg5.json 2> $out/pmc_cfg5.err
cp $out/pmc_bench/FETCH_SIZE.log $out/pmc_bench_line.txt 2>/dev/null
# 4. where the dominant kernel's issue slots go: SQ wait / LDS / MFMA counters by queue (tools/pmc_stalls.sh)
bash tools/pmc_stalls.sh $out/pmc_stalls > $out/pmc_stalls.log 2>&1
# 5. the flag-ordered tail: chain step and task timings from in-kernel stamps (tools/flow_tr.sh)
N=8192 bash tools/flow_tr.sh $out/flowt > $out/flow_trace_n8192.txt 2>&1
python3 tools/flow_curve.py $out/flowt/trace.bin > $out/flow_curve_n8192.txt 2>&1
rm -rf $out/flowt
# 6. the vendor's routines at the same shapes (tools only: never on the product path)
timeout 300 python3 tools/vendor_yardstick.py > $out/vendor.json 2> $out/vendor.err
find $out -name "*counter_collection.csv" -delete
find $out -name "*agent_info.csv" -delete
ls -R $out | head -80
