#!/bin/bash
# Ranker iteration at BASELINE configs[2] scale: plain timing (serial + pipelined) and a kernel trace.
# usage: tools/prof_ranker.sh <tag> [extra bench_ranker args]
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
A="tools/bench_ranker.py --users 1371980 --items 105542 --edges 31800000 --batch 24 --device-sampler --steps 50 --warmup 10 $@"
python3 $A > $out/serial.json 2> $out/serial.err
python3 $A --pipelined > $out/pipelined.json 2> $out/pipelined.err
cat $out/serial.json $out/pipelined.json
rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 $A --pipelined > $out/kt.log 2>&1
python3 tools/summarize_rocprof.py $out $out/$tag > /dev/null
head -40 $out/$tag.md
find $out -name "*_kernel_trace.csv" -delete 2>/dev/null || true
find $out -name "*agent_info.csv" -delete 2>/dev/null || true
