#!/usr/bin/env python3
"""Mean per-launch value of every collected counter for the kernels whose name matches a regex.

    python scripts/pmc_kernel.py '<regex>' <pmc-dir> [<pmc-dir> ...]

Each <pmc-dir> is the output of one `rocprofv3 --pmc ...` pass (counters never combined with tracing)."""
import csv
import glob
import re
import sys
from collections import defaultdict

rx = re.compile(sys.argv[1])
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"^void ", "", r["Kernel_Name"])
            if not rx.search(n):
                continue
            n = re.sub(r"\(.*", "", n)[:70]
            a = acc[n][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
for n, cs in sorted(acc.items()):
    print(n)
    for k, (s, c) in sorted(cs.items()):
        print("   %-44s %16.1f   (%d launches)" % (k, s / c, c))
