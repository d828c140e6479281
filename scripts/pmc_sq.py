#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ counter passes per kernel (mean per dispatch).

    python scripts/pmc_sq.py <dir-with-*_counter_collection.csv> [kernel-substring ...]
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(bchmc::)?([A-Za-z0-9_]+(<[a-z, ]+>)?)", name)
    return m.group(2) if m else name[:40]


def main():
    pats = sys.argv[2:] or ["k_scatter_tile", "k_gather_tile", "k_bin", "k_reorder"]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = short(r["Kernel_Name"])
            if not any(p in n for p in pats):
                continue
            a = acc[n][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for n, cs in sorted(acc.items()):
        print(n)
        for c, (s, k) in sorted(cs.items()):
            print("   %-28s %16.0f  (mean of %d dispatches)" % (c, s / k, k))


if __name__ == "__main__":
    main()
