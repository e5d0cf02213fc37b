#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + separate FETCH_SIZE / WRITE_SIZE passes) into the
small summaries committed under profiles/.

usage: summarize_rocprof.py <prof_dir> <out_prefix> [--traffic-json profiles/traffic.json]
  <prof_dir>/kt/**/_kernel_stats.csv          rocprofv3 --kernel-trace --stats
  <prof_dir>/fetch/**/_counter_collection.csv rocprofv3 --pmc FETCH_SIZE
  <prof_dir>/write/**/_counter_collection.csv rocprofv3 --pmc WRITE_SIZE
HBM-side bytes follow MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of 16-B-per-lane reads (128-B requests tallied at 64 B) => bytes = (2*FETCH + WRITE)*1024.
The Adam kernel (known bytes: 4 reads + 3 writes of the table) serves as the calibration row.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:80]


def main():
    prof, out = sys.argv[1], sys.argv[2]
    lines = []
    stats = glob.glob(os.path.join(prof, "kt", "**", "*_kernel_stats.csv"), recursive=True)
    per_kernel = {}
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        lines.append("## rocprofv3 --kernel-trace --stats (per kernel)\n")
        lines.append("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
        for r in rows[:25]:
            k = short(r["Name"])
            per_kernel[k] = float(r["AverageNs"]) / 1e3
            lines.append(f"| `{k}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | "
                         f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    pmc = {}
    for name in ("fetch", "write"):
        files = glob.glob(os.path.join(prof, name, "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        pmc[name] = {k: sum(v) / len(v) for k, v in agg.items()}
    traffic = {}
    if pmc:
        lines.append("\n## PMC passes (separate runs): FETCH_SIZE, WRITE_SIZE — average per dispatch\n")
        lines.append("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | bytes = (2*F + W)*1024 |\n|---|---:|---:|---:|")
        keys = sorted(set(pmc.get("fetch", {})) | set(pmc.get("write", {})),
                      key=lambda k: -(pmc.get("fetch", {}).get(k, 0) + pmc.get("write", {}).get(k, 0)))
        for k in keys[:14]:
            f, w = pmc.get("fetch", {}).get(k, 0.0), pmc.get("write", {}).get(k, 0.0)
            b = (2 * f + w) * 1024
            traffic[k] = b
            lines.append(f"| `{k}` | {f:.0f} | {w:.0f} | {b/1e9:.3f} GB |")
    def dense(k):  # the plain dense launch: SPARSE=false and (rows / fix-up kernels) ADAM=false instantiations
        m = re.match(r"spmm_(rows|items|fixup|sweep)_kernel<([^>]*)>", k)
        if not m:
            return False
        a = [x.strip() for x in m.group(2).split(",")]
        flags = {"rows": a[4:6], "items": a[4:5], "fixup": a[2:4], "sweep": a[2:3]}[m.group(1)]
        return all(f == "false" for f in flags)
    spmm = sum(v for k, v in traffic.items() if dense(k))
    spmm_us = sum(v for k, v in per_kernel.items() if dense(k))
    if spmm:
        lines.append(f"\nOne plain DENSE propagate launch (sweep or items + rows + fixup kernels; SPARSE=false, no optimizer epilogue): {spmm/1e9:.3f} GB of L2-miss traffic"
                     + (f", {spmm_us:.1f} us summed kernel time => {spmm/spmm_us/1e6:.2f} TB/s" if spmm_us else ""))
    with open(out + ".md", "w") as fh:
        fh.write(f"# {os.path.basename(out)}\n\n" + "\n".join(lines) + "\n")
    if "--traffic-json" in sys.argv and spmm:
        path = sys.argv[sys.argv.index("--traffic-json") + 1]
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import kernel_source_hash
        json.dump({"spmm_hbm_bytes_per_launch": spmm, "kernel_source_hash": kernel_source_hash(), "per_kernel_bytes": traffic,
                   "per_kernel_avg_us": per_kernel, "source": os.path.basename(out) + ".md",
                   "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 correction per MI355X_MICROARCH.md"},
                  open(path, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
