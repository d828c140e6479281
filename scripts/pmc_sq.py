#!/usr/bin/env python3
"""SQ-counter summary of the two particle-mesh kernels -> profiles/rNN_sq_tile81.json (bench.py's roofline.valu).

    python scripts/pmc_sq.py <kernel_stats.csv> <out.json> <pmc-dir> [<pmc-dir> ...]

Each <pmc-dir> holds the *counter_collection.csv of ONE `rocprofv3 --pmc ...` pass of the bench command (the SQ block
has 8 slots per pass on gfx950, so the counters come from several separate passes; never combined with tracing).
<kernel_stats.csv> is the `rocprofv3 --kernel-trace --stats` summary of the same command: it supplies the average
launch duration (a PMC pass runs at a lower clock, MI355X_MICROARCH.md "DVFS give-back" (2), so times are never
taken from one).

Method for the fp64 VALU roofline (peak 78.6 TFLOP/s = 256 CUs x 4 SIMDs x 16 fp64 lanes x 2 flop x 2.4 GHz):
  * flop_per_launch = 64 * lanes_active * (ADD_F64 + MUL_F64 + 2 * FMA_F64 + TRANS_F64) wave-instructions
    -- executed lane-operations, counting an FMA as 2 and v_rsq_f64 as 1; compares, selects, conversions, integer
    and address arithmetic count as zero flop although they occupy the same issue slots;
  * lanes_active = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)  (rocprofiler's VALUUtilization);
  * valu_busy = SQ_ACTIVE_INST_VALU / (n_CU * GRBM_GUI_ACTIVE / n_XCD)   (rocprofiler's VALUBusy; quad-cycles per CU
    over the dispatch's active cycles, GRBM_GUI_ACTIVE being summed over the 8 XCDs);
  * valu_insts_per_wave = SQ_INSTS_VALU / SQ_WAVES: the issue-slot view, which is what the kernel optimisations act on.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

PEAK_FP64_VECTOR_TFLOPS = 78.6
N_CU, N_XCD = 256, 8
KERNELS = ("k_scatter_tile81<double", "k_gather_tile81<double")


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(bchmc::)?([A-Za-z0-9_]+(<[a-z0-9, ]+>)?)", name)
    return m.group(2) if m else name[:40]


def main():
    stats_csv, out_json, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n = short(r["Kernel_Name"])
                if not any(n.startswith(k) for k in KERNELS):
                    continue
                a = acc[n][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    dur = {}
    for r in csv.DictReader(open(stats_csv)):
        n = short(r["Name"])
        if any(n.startswith(k) for k in KERNELS):
            dur[n] = (float(r["AverageNs"]), int(r["Calls"]))
    out = dict(peak_tflops=PEAK_FP64_VECTOR_TFLOPS, unit="TFLOP/s", bound="fp64 vector ALU (issue slots) + LDS atomics",
               method="flop = 64 x lanes_active x (ADD_F64 + MUL_F64 + 2 FMA_F64 + TRANS_F64) wave-instructions per "
                      "launch (SQ counters, separate rocprofv3 --pmc passes); time = rocprofv3 --kernel-trace --stats "
                      "average of the same command; compares / selects / address arithmetic occupy issue slots but "
                      "count as zero flop", kernels={})
    for n, cs in sorted(acc.items()):
        c = {k: s / max(cnt, 1) for k, (s, cnt) in cs.items()}
        k = dict(counters_mean_per_launch={kk: round(v, 1) for kk, v in sorted(c.items())},
                 dispatches_counted=max(cnt for _, cnt in cs.values()))
        act, thr = c.get("SQ_ACTIVE_INST_VALU"), c.get("SQ_THREAD_CYCLES_VALU")
        lanes = thr / (64.0 * act) if act and thr else None
        f64 = [c.get("SQ_INSTS_VALU_" + x) for x in ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64")]
        if lanes is not None:
            k["lanes_active"] = round(lanes, 4)
        gui = c.get("GRBM_GUI_ACTIVE")
        if act and gui:
            k["valu_busy"] = round(act / (N_CU * gui / N_XCD), 4)
        if all(v is not None for v in f64) and lanes is not None:
            add, mul, fma, trans = f64
            flop = 64.0 * lanes * (add + mul + 2.0 * fma + trans)
            k["f64_wave_insts_per_launch"] = dict(add=add, mul=mul, fma=fma, trans=trans)
            k["flop_per_launch"] = flop
            if c.get("SQ_INSTS_VALU_FLOPS_FP64") is not None:
                k["hw_flops_fp64_counter"] = c["SQ_INSTS_VALU_FLOPS_FP64"]
            if n in dur:
                ns, calls = dur[n]
                k["avg_launch_ms"] = round(ns / 1e6, 4)
                k["launches_timed"] = calls
                k["achieved_tflops"] = round(flop / (ns * 1e-9) / 1e12, 3)
                k["frac"] = round(flop / (ns * 1e-9) / 1e12 / PEAK_FP64_VECTOR_TFLOPS, 4)
        if c.get("SQ_INSTS_VALU") and c.get("SQ_WAVES"):
            k["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        if c.get("SQ_LDS_BANK_CONFLICT") is not None and c.get("SQ_LDS_IDX_ACTIVE"):
            k["lds_conflict_frac_of_lds_active"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        out["kernels"][n] = k
    for n, k in out["kernels"].items():
        print(n)
        for kk, v in k.items():
            if kk != "counters_mean_per_launch":
                print("   %-34s %s" % (kk, v))
        for kk, v in k["counters_mean_per_launch"].items():
            print("      %-30s %16.0f" % (kk, v))
    json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
