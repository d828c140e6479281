#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
HBM bytes per leapfrog step, per kernel and in total.

    python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps> [out.json [grid [fp64|fp32]]]

Counter units and gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md section HBM:
  * FETCH_SIZE / WRITE_SIZE are in KiB (bytes = value * 1024);
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read -> doubled here;
    WRITE_SIZE reads exactly for streaming stores and float atomics.
The doubling is calibrated for 16 B/lane streaming reads; our k-space kernels (double2 per lane) match that
pattern, 8 B/lane kernels are cross-checked against their known byte counts in the printed table.
Only the dispatches of the timed trajectory's step loop are counted: from the first k_kick_drift_za after the
last k_init_ctl (the initial force evaluation lies before it) up to the last k_step_boundary / k_assemble.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(bchmc::)?([A-Za-z0-9_]+(<[a-z0-9, ]+>)?)", name)
    return m.group(2) if m else name[:40]


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    start = max(i for i, n in enumerate(names) if n == "k_init_ctl")
    # first step: k_kick_drift_za<true>, or -- planes mode at the ends -- the SECOND BX_FIRST launch of
    # k_step_boundary_x after k_init_ctl (the first one is the Psi^ of the force evaluation before the first step)
    drift = re.compile(r"k_kick_drift_za<(double, |float, )?true>")
    bx_first = re.compile(r"k_step_boundary_x<.*, 1(, (false|true))?>")
    cand = [i for i in range(start, len(names)) if bx_first.fullmatch(names[i])]
    if len(cand) >= 2:
        first = cand[1]
    else:
        first = next(i for i in range(start, len(names)) if drift.fullmatch(names[i]))
    end = re.compile(r"k_step_boundary<.*>|k_step_boundary_x<.*, 2(, (false|true))?>|k_assemble<(double, |float, )?true>")
    last = max(i for i, n in enumerate(names) if end.fullmatch(n))
    per = defaultdict(lambda: [0.0, 0])
    for r, n in zip(rows[first:last + 1], names[first:last + 1]):
        per[n][0] += float(r["Counter_Value"]) * 1024.0
        per[n][1] += 1
    return per


def main():
    fetch_csv, write_csv, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, write = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    kernels = {}
    tot_r = tot_w = 0.0
    for n in sorted(set(fetch) | set(write)):
        r = 2.0 * fetch[n][0] / steps  # gfx950: FETCH_SIZE counts half the streamed bytes
        w = write[n][0] / steps
        calls = max(fetch[n][1], write[n][1]) / steps
        kernels[n] = dict(read_MB_per_step=round(r / 1e6, 1), write_MB_per_step=round(w / 1e6, 1),
                          launches_per_step=round(calls, 2))
        tot_r += r
        tot_w += w
    grid = int(sys.argv[5]) if len(sys.argv) > 5 else 256
    precision = sys.argv[6] if len(sys.argv) > 6 else "fp64"
    # Calibration of the x2 FETCH_SIZE correction on kernels whose streamed bytes are known exactly (8 B per lane
    # reads, not the 16 B per lane the microarch guide calibrated): R = one real array.
    R = grid ** 3 * (8 if precision == "fp64" else 4)
    known = {"k_bin_direct": (3 * R, None), "k_bin<double, true>": (3 * R, None), "k_bin<float, true>": (3 * R, None),
             "k_partial_like": (4 * R, 1 * R)}
    check = {}
    for n, k in kernels.items():
        for pref, (rd, wr) in known.items():
            if n.startswith(pref) and k["launches_per_step"] > 0:
                got_r = k["read_MB_per_step"] * 1e6 / k["launches_per_step"]
                check[n] = dict(read_expected_MB=round(rd / 1e6, 1), read_counter_x2_MB=round(got_r / 1e6, 1),
                                ratio=round(got_r / rd, 4))
                if wr:
                    got_w = k["write_MB_per_step"] * 1e6 / k["launches_per_step"]
                    check[n].update(write_expected_MB=round(wr / 1e6, 1), write_counter_MB=round(got_w / 1e6, 1))
    out = dict(grid=grid, precision=precision, steps=steps, hbm_read_bytes_per_step=tot_r, hbm_write_bytes_per_step=tot_w,
               hbm_bytes_per_step=tot_r + tot_w, fetch_size_correction=2.0, fetch_size_check=check, kernels=kernels)
    print("%-34s %10s %10s %8s" % ("kernel", "read MB", "write MB", "launches"))
    for n, k in kernels.items():
        print("%-34s %10.1f %10.1f %8.2f" % (n, k["read_MB_per_step"], k["write_MB_per_step"], k["launches_per_step"]))
    print("total per step: read %.1f MB, write %.1f MB, sum %.3f GB" % (tot_r / 1e6, tot_w / 1e6, (tot_r + tot_w) / 1e9))
    for n, c in check.items():
        print("FETCH_SIZE x2 check on %s: %s" % (n, c))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
