"""What runs in the GAP between two training iterations?  From a rocprofv3 kernel trace of the ranker loop: for the steady-state
iterations, the time between the last kernel of iteration i (adam_multi_kernel) and the first of iteration i + 1 on the step
stream, and the kernels of the OTHER streams (the sampler's side stream) that run in that window.
usage: python3 tools/iter_gap.py <dir with *kernel_trace.csv>"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id"))) for r in csv.DictReader(open(f))]
rows.sort()
marker = "adam_multi_kernel"
stream = collections.Counter(r[3] for r in rows if marker in r[2]).most_common(1)[0][0]
main = [r for r in rows if r[3] == stream]
other = [r for r in rows if r[3] != stream]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:48]
ends = [i for i, r in enumerate(main) if marker in r[2]]
ends = ends[len(ends) // 3: -1]
gaps, periods, busy_other, inside = [], [], [], collections.Counter()
for a, b in zip(ends[:-1], ends[1:]):
    periods.append(main[b][1] - main[a][1])
for e in ends:
    g0, g1 = main[e][1], main[e + 1][0]
    gaps.append(g1 - g0)
    for s, t, n, _ in other:
        if t > g0 and s < g1:
            inside[short(n)] += min(t, g1) - max(s, g0)
print(f"step stream {stream}: {len(ends)} iterations; period {sum(periods)/len(periods)/1e3:.1f} us; gap between iterations "
      f"avg {sum(gaps)/len(gaps)/1e3:.1f} us (min {min(gaps)/1e3:.1f}, max {max(gaps)/1e3:.1f})")
# time per iteration spent by the other streams' kernels, and how much of it overlaps step kernels
tot_other = collections.Counter()
lo, hi = main[ends[0]][1], main[ends[-1]][1]
for s, t, n, _ in other:
    if s >= lo and t <= hi:
        tot_other[short(n)] += t - s
n_it = len(ends) - 1
print("other streams' kernels, us per iteration (and us of it inside the gap):")
for k, v in tot_other.most_common(14):
    print(f"  {k:50s} {v/n_it/1e3:7.1f}   {inside[k]/len(ends)/1e3:6.1f}")
print(f"  total {sum(tot_other.values())/n_it/1e3:.1f} us per iteration")
# step kernels: busy per iteration
busy = sum(r[1] - r[0] for r in main if lo <= r[0] and r[1] <= hi) / n_it
print(f"step stream busy {busy/1e3:.1f} us per iteration")
