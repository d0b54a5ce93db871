#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 --kernel-trace run kept as a rocpd database (<name>_results.db).
usage: db_summary.py <results.db> <steps_in_run> [title] [rows]"""
import sqlite3
import sys


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else path
    nrows = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end - start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# {title}\n")
    print(f"GPU busy time per step: {tot / steps / 1e6:.2f} ms ({steps:g} steps in the run, warm-up included)\n")
    print("| ms/step | % | calls/step | avg us | kernel |\n|---:|---:|---:|---:|---|")
    for n, c, t in rows[:nrows]:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"| {t / steps / 1e6:.3f} | {100 * t / tot:.2f} | {c / steps:.1f} | {t / c / 1e3:.1f} | `{n[:110]}` |")


if __name__ == "__main__":
    main()
