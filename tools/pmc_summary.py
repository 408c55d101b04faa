#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs (one directory per PMC pass) per kernel.
Usage: pmc_summary.py <pmc_dir> [out.txt]"""
import csv, glob, os, sys, collections

def short(name):
    for k in ("k_bp", "k_scatter_box", "k_scatter_slab", "k_scatter_tile", "k_sweep_map", "k_depth", "k_traverse", "k_acc_combine"):
        if k in name:
            return k
    return None

def main():
    d = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in sorted(glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                if not k:
                    continue
                c = row["Counter_Name"]
                agg[k][c] += float(row["Counter_Value"])
                calls[k][c] += 1
    lines = []
    for k in sorted(agg):
        lines.append("== %s" % k)
        for c in sorted(agg[k]):
            n = calls[k][c]
            lines.append("   %-36s total %16.0f   per launch %14.0f   (%d launches)" % (c, agg[k][c], agg[k][c] / n, n))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)

if __name__ == "__main__":
    main()
