#!/bin/bash
# SQ wave-state counters of the top-K prefilter kernel (one stream): where a wavefront's cycles go.
# WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles).
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r03_topk_sq; rm -rf $out; mkdir -p $out
export LAPLACE_TOPK_STREAMS=1
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 150 rocprofv3 --pmc $set -d $out/$tag --output-format csv -- python3 tools/bench_topk.py --full --pre-only --users 8192 > $out/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<EOF2
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "topk_prefilter_bf16" in n and "false>" in n[:80]:
            acc["VOTE"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    for c, x in sorted(v.items()):
        h = len(x) // 2
        print(f"{c:28s} k=12 chunks avg {sum(x[:h])/max(h,1):14.0f}   k=256 chunks avg {sum(x[h:])/max(len(x)-h,1):14.0f}")
EOF2
find $out -name "*agent_info.csv" -delete 2>/dev/null || true
