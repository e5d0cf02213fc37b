#!/bin/bash
# tools/kt.sh TAG PATTERN script args...: rocprofv3 kernel stats of one python command, the lines matching PATTERN
tag=$1; pat=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 "$@" > $out/kt.log 2>&1
grep -E "ms per|plan items" $out/kt.log
f=$(find $out/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$pat" <<'EOF2'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>4} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
EOF2
find $out -name "*_kernel_trace.csv" -delete 2>/dev/null || true
find $out -name "*agent_info.csv" -delete 2>/dev/null || true
