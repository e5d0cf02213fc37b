#!/bin/bash
# Profile of the default bench command: kernel trace + the two PMC passes, summary -> profiles/<name>.md and
# profiles/traffic.json (stamped with the kernel source hash bench.py checks).
# usage (on the GPU box): tools/prof_bench.sh <name> [bench args...]
set -e
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
export TMPDIR=/tmp
A="bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-map --no-plain-leg --no-pmc --no-c4 --no-ranker --no-pinsage --no-e2e --no-topk $@"
rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 $A > $out/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch --output-format csv -- python3 $A > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write --output-format csv -- python3 $A > $out/write.log 2>&1
python3 tools/summarize_rocprof.py $out $out/$name --traffic-json $out/traffic.json > /dev/null
tail -3 $out/$name.md
find $out -name "*_kernel_trace.csv" -delete 2>/dev/null || true
find $out -name "*agent_info.csv" -delete 2>/dev/null || true
