#!/bin/bash
# kernel timeline of the headline step under an environment variant: tools/scratch/trace_split.sh <name> VAR=val ...
name=$1; shift
out=gpurun_out/trace_$name
rm -rf $GRAFT_REPO_ROOT/$out; mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/bench -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sharded > $out/bench_traced.json 2> $out/bench_traced.err
python3 tools/timeline_full.py $out/bench $out/timeline.txt > /dev/null 2>&1
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
