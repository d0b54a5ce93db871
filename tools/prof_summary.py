#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats run (kernel_stats.csv) into a per-step table.
usage: prof_summary.py <kernel_stats.csv> <steps_in_run> [title]"""
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?)I", n)
    if m:
        n = m.group(1) + "<...> " + ("bf16" if "DF16b" in n else "") + (" f32" if "EEf" in n or "IfE" in n else "")
    return n[:100]


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else path
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {title}\n")
    print(f"GPU busy time per step: {tot / steps / 1e6:.2f} ms ({steps:g} steps in the run, warm-up included)\n")
    print("| ms/step | % | calls/step | avg us | kernel |\n|---:|---:|---:|---:|---|")
    for r in rows[:30]:
        print(f"| {float(r['TotalDurationNs']) / steps / 1e6:.3f} | {float(r['Percentage']):.2f} | "
              f"{int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | `{short(r['Name'])}` |")


if __name__ == "__main__":
    main()
