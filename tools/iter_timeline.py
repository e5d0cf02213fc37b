"""Average per-position timeline of one training iteration from a rocprofv3 kernel trace: the step stream's kernels
between two occurrences of a marker kernel (default adam_multi_kernel), averaged over the iterations that have the
modal number of kernels.  usage: python3 tools/iter_timeline.py <dir with *kernel_trace.csv> [marker]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_multi_kernel"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id"))) for r in csv.DictReader(open(f))]
rows.sort()
stream = collections.Counter(r[3] for r in rows if marker in r[2]).most_common(1)[0][0]
rows = [r for r in rows if r[3] == stream]
iters, cur = [], []
for r in rows:
    cur.append(r)
    if marker in r[2]:
        iters.append(cur); cur = []
iters = iters[len(iters) // 3:]
mode = collections.Counter(len(i) for i in iters).most_common(1)[0][0]
iters = [i for i in iters if len(i) == mode]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:60]
print(f"{len(iters)} iterations of {mode} kernels on stream {stream}")
tot_busy = tot_gap = 0.0
for p in range(mode):
    d = [i[p][1] - i[p][0] for i in iters]
    g = [i[p][0] - i[p - 1][1] for i in iters] if p else [0]
    names = collections.Counter(short(i[p][2]) for i in iters).most_common(1)[0][0]
    tot_busy += sum(d) / len(d); tot_gap += max(0.0, sum(g) / len(g))
    print(f"{p:3d} {names:60s} {sum(d)/len(d)/1e3:7.1f} us   gap before {sum(g)/len(g)/1e3:6.1f} us")
span = [i[-1][1] - i[0][0] for i in iters]
print(f"busy {tot_busy/1e3:.1f} us, gaps {tot_gap/1e3:.1f} us, first-start to last-end {sum(span)/len(span)/1e3:.1f} us")
