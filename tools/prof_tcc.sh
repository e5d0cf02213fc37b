#!/bin/bash
# TCC counters of the spmm kernels: hits / misses / fabric read requests (tools/exp_locality.py args follow the tag)
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $out/hm --output-format csv -- python3 "$@" > $out/hm.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $out/rd --output-format csv -- python3 "$@" > $out/rd.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum -d $out/rq --output-format csv -- python3 "$@" > $out/rq.log 2>&1
for d in hm rd rq; do python3 tools/pmcsum.py $out/$d || true; done
find $out -name "*_kernel_trace.csv" -delete 2>/dev/null || true
find $out -name "*agent_info.csv" -delete 2>/dev/null || true
