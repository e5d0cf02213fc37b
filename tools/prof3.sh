#!/bin/bash
# Three rocprofv3 passes (kernel trace, FETCH_SIZE, WRITE_SIZE) of one command, summarised per spmm kernel.
# usage: tools/prof3.sh <tag> <python script and args...>   (outputs under gpurun_out/<tag>/)
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 "$@" > $out/plain.log 2>&1
tail -1 $out/plain.log
rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 "$@" > $out/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch --output-format csv -- python3 "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write --output-format csv -- python3 "$@" > $out/write.log 2>&1
python3 tools/summarize_rocprof.py $out $out/summary > /dev/null
grep -E "spmm|DENSE" $out/summary.md || true
# keep only the small csv files
find $out -name "*.db" -delete 2>/dev/null || true
find $out -name "*_kernel_trace.csv" -delete 2>/dev/null || true
find $out -name "*agent_info.csv" -delete 2>/dev/null || true
