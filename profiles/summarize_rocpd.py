#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into the text summary kept in profiles/.

usage: python profiles/summarize_rocpd.py gpurun_out/prof_x/x_results.db > profiles/rNN_name.stats.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)            # drop the argument list
    name = re.sub(r"^void ", "", name)
    if "rocprim" in name:
        m = re.search(r"(radix_sort\w*|merge_sort\w*|onesweep\w*|block_sort\w*|scan\w*)", name)
        return "rocprim::" + (m.group(1) if m else "kernel")
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, duration from kernels").fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print(f"# source: {sys.argv[1]}  (rocprofv3 --kernel-trace --stats; durations in us)")
    print(f"{'kernel':<92}{'calls':>7}{'total_us':>13}{'avg_us':>11}{'min_us':>11}{'max_us':>11}{'pct':>7}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:<92}{a[0]:>7}{a[1]/1e3:>13.1f}{a[1]/a[0]/1e3:>11.2f}{a[2]/1e3:>11.2f}{a[3]/1e3:>11.2f}{100*a[1]/total:>7.2f}")


if __name__ == "__main__":
    main()
