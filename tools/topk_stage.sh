#!/bin/bash
# stage timing of the fused top-K kernel: full-chunk dispatch durations (kernel trace) for the named library variants
# (liblaplace_hip_<v>.so built with -DMI_TOPK_STAGE=<n>; wrong results, timing only).  131 072 users = 48 chunks per k:
# the shader clock needs ~15 ms of load to settle after an idle start (809 -> 727 us per chunk), medians are of the settled half.
export TMPDIR=/tmp
P=laplace-gnn-recommendation_amd
# the variant is selected through LAPLACE_HIP_LIB (laplace_amd/_lib.py): the product library is never overwritten
for v in full "$@"; do
  if [ "$v" = full ]; then export LAPLACE_HIP_LIB=$PWD/$P/liblaplace_hip.so; else export LAPLACE_HIP_LIB=$PWD/$P/liblaplace_hip_$v.so; fi
  rm -rf /tmp/tk_$v
  timeout -k 10 150 rocprofv3 --kernel-trace -d /tmp/tk_$v --output-format csv -- python3 tools/bench_topk.py --full --users 131072 > /dev/null 2>&1
  python3 - <<EOF2
import csv, glob
f = glob.glob("/tmp/tk_$v/**/*kernel_trace.csv", recursive=True)[0]
d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f))
      if "topk_scores_filter" in r["Kernel_Name"] and r["Grid_Size_Y"] == "42"]
import statistics as st
h = len(d) // 2   # first half: k = 12, second half: k = 256; the clock takes ~20 chunks to settle after the idle start
print("$v", f"full chunks, settled: k=12 median {st.median(d[h // 2:h]):.0f} us, k=256 median {st.median(d[h + h // 2:]):.0f} us; first chunk {d[0]:.0f} us")
EOF2
done
