#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes of tools/pmc_mfma.sh per kernel (bench.py's kernel tags):

    python tools/pmc_mfma_summary.py <dir with p1/ p2/ p3/> [out.json]

Per kernel, averaged over its launches:
  mfma_busy_pct   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 256 CUs x 4 SIMDs)   (rocprofv3's MfmaUtil formula
                    with the GUI-active sum divided by the 8 XCDs it is summed over: MI355X_MICROARCH.md, DVFS give-back)
  wait_any_pct / wait_inst_pct / active_pct = share of wave-cycles parked at s_waitcnt or s_barrier / stalled at issue / issuing
  wait_lds_pct    = issue stalls on the LDS pipe; lds_conflict_pct = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  mfma_mops       = SQ_INSTS_VALU_MFMA_MOPS_BF16 (x 512 FLOP) -> executed MFMA FLOP per launch, padding included
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import tag_of  # noqa: E402


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                t = tag_of(r["Kernel_Name"])
                acc[t][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[t][r["Counter_Name"]] += 1
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    # duration of the same dispatch (ns): GRBM_GUI_ACTIVE / 8 / duration = the profiler's view of the clock
                    acc[t]["_dur_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    cnt[t]["_dur_ns"] += 1
    return {t: {c: v / cnt[t][c] for c, v in cs.items()} for t, cs in acc.items()}, {t: max(c.values()) for t, c in cnt.items()}


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(d, "pmc_mfma.json")
    avg, launches = load(d)
    res = {}
    for t, c in sorted(avg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * launches[kv[0]]):
        g = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        row = {"launches": launches[t]}
        if g > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            row["mfma_busy_pct"] = round(100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 256 * 4), 2)
            row["gui_active_cycles"] = round(g)
        if g > 0 and c.get("_dur_ns", 0) > 0:
            # reads HIGH on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back); the in-kernel clock is
            # s_memtime / s_memrealtime (tools/probes/clock_calib.hip calibrates both against the 16-cycle MFMA issue rate)
            row["dispatch_us"] = round(c["_dur_ns"] / 1e3, 1)
            row["gui_active_over_duration_ghz"] = round(g / c["_dur_ns"], 3)
        if wc > 0:
            for k, name in (("SQ_WAIT_ANY", "wait_any_pct"), ("SQ_WAIT_INST_ANY", "wait_inst_pct"), ("SQ_ACTIVE_INST_ANY", "active_pct")):
                if k in c:
                    row[name] = round(100.0 * c[k] / wc, 2)
        if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in c:
            row["mfma_gflop_executed"] = round(c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / 1e9, 3)
        if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
            row["lds_conflict_pct"] = round(100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 2)
        row["raw"] = {k: round(v, 1) for k, v in sorted(c.items()) if not k.startswith("_")}
        res[t] = row
    doc = {"collected": "tools/pmc_mfma.sh: rocprofv3 --kernel-trace --pmc <8 SQ counters per pass>, three passes, VDQN_NO_OVERLAP=1, "
                        "bench.py --steps 2 --warmup 1 --no-profile, batch 256 bf16; values are per-launch averages",
           "per_kernel": res}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"{'kernel':32s} {'n':>4s} {'mfma%':>7s} {'wait%':>7s} {'stall%':>7s} {'act%':>7s} {'ldsconf%':>8s} {'us':>8s} {'GUI GHz':>8s}")
    for t, r in res.items():
        print(f"{t:32s} {r['launches']:4d} {r.get('mfma_busy_pct', float('nan')):7.2f} {r.get('wait_any_pct', float('nan')):7.2f} "
              f"{r.get('wait_inst_pct', float('nan')):7.2f} {r.get('active_pct', float('nan')):7.2f} {r.get('lds_conflict_pct', float('nan')):8.2f} "
              f"{r.get('dispatch_us', float('nan')):8.1f} {r.get('gui_active_over_duration_ghz', float('nan')):8.3f}")


if __name__ == "__main__":
    main()
