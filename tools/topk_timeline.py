"""Timeline of the LAST top-K call in a rocprofv3 kernel trace: every kernel with start / end relative to the call's first
kernel and its stream, plus how busy the chip was with the MFMA kernel.  usage: python3 tools/topk_timeline.py <dir> [n_last_kernels]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id"))) for r in csv.DictReader(open(f))]
rows.sort()
# the last call = the trailing run of kernels after the last gap > 2 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(0, i - 8):i]) > 2_000_000:
        cut = i
rows = rows[cut:]
t0 = rows[0][0]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:44]
streams = sorted(set(r[3] for r in rows))
for s, e, n, q in rows:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  s{streams.index(q)}  {short(n)}")
pre = [(s, e) for s, e, n, q in rows if "prefilter_bf16" in n]
span = rows[-1][1] - t0
print(f"span {span / 1e3:.1f} us; prefilter kernels {len(pre)}: busy {sum(e - s for s, e in pre) / 1e3:.1f} us")
