#!/usr/bin/env python
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) into
profiles/pmc_latest.json: HBM-side bytes per launch for every libvdqn kernel, keyed by bench.py's kernel tags.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming reads, so read bytes = 2 * FETCH_SIZE * 1024.
"""
import collections
import csv
import json
import re
import sys


def tag_of(kernel_name: str):
    m = re.search(r"conv64_kernel<(\d+)[,>]", kernel_name)
    if m:
        return f"conv64<bf16,{('fwd', 'dgrad')[int(m[1])]}>"
    m = re.search(r"igemm_kernel<(unsigned short|float), (\d+), (\d+), (\d+), (\d+)(?:, \d+)?>", kernel_name)
    if m:
        dt = "bf16" if m[1] == "unsigned short" else "f32"
        bm, bn, mode = int(m[2]), int(m[3]), int(m[4])
        if mode == 3:
            return f"stem_conv_pool<{dt}>"
        size = "256x64" if bm == 256 else str(bn)
        return f"igemm<{dt},{size},{('fwd', 'dgrad', 'dgrad_s2')[mode]}>"
    m = re.search(r"skinny_kernel<(\d+), (\d+), (true|false)", kernel_name)
    if m:  # the Q-head's skinny GEMMs (forward and data-gradient launches share a symbol)
        return "skinny<bf16,conv>" if m[3] == "true" else f"skinny<bf16,{m[1]}x{m[2]}>"
    m = re.search(r"win9d_kernel<(\d+)", kernel_name)
    if m:  # plane-window kernel of the stride-2 data gradients
        return f"igemm_s2win<bf16,{32 * int(m[1])},dgrad>"
    m = re.search(r"win9sp?_kernel", kernel_name)
    if m:  # plane-window kernel of the stride-2 3x3 forward convolutions
        return "igemm_s2win<bf16,128,fwd>"
    m = re.search(r"win9[um]_kernel<(\d+)(?:, \d+)*>", kernel_name)
    if m:  # the unrolled nine-tap kernel (bf16 only), same tag
        return f"igemm_win<bf16,128,{('fwd', 'dgrad')[int(m[1])]}>"
    m = re.search(r"igemm_win9_kernel<(unsigned short|float), (\d+)>", kernel_name)
    if m:  # the nine-tap window variant runs under the same bench.py tag as the three-tap kernel it replaced
        return f"igemm_win<{'bf16' if m[1] == 'unsigned short' else 'f32'},128,{('fwd', 'dgrad')[int(m[2])]}>"
    m = re.search(r"igemm_win_kernel<(unsigned short|float), (\d+), (\d+)>", kernel_name)
    if m:
        return f"igemm_win<{'bf16' if m[1] == 'unsigned short' else 'f32'},{m[2]},{('fwd', 'dgrad')[int(m[3])]}>"
    m = re.search(r"wgrad_win_kernel<(\d+), (\d+), (\d+)>", kernel_name)
    if m:
        return "wgrad_win<bf16,128>" if m[3] == "8" else ("wgrad_win<bf16,128x64>" if m[1] == "128" else "wgrad_win<bf16,64>")
    m = re.search(r"wgrad_win_kernel<(\d+)(?:, \d+)?>", kernel_name)
    if m:
        return f"wgrad_win<bf16,{m[1]}>"
    if re.search(r"wgrad_win_kernel\b(?!<)", kernel_name):  # round 6: the one shipped tiling, no template arguments
        return "wgrad_win<bf16,64>"
    m = re.search(r"wgrad_kernel<(unsigned short|float), (\d+)>", kernel_name)
    if m:
        return f"wgrad<{'bf16' if m[1] == 'unsigned short' else 'f32'},{m[2]}>"
    if "stem_wgrad_pool_kernel" in kernel_name:
        return "wgrad_stem_pool<bf16>"
    if "stem_wgrad_kernel" in kernel_name:
        return "wgrad_stem<bf16>"
    if "pack_input" in kernel_name:
        return "pack_input"
    if "stem_kernel" in kernel_name:
        return "stem_conv_pool<bf16>"
    m = re.search(r"(\w+)_kernel", kernel_name)
    return m[1] if m else kernel_name[:40]


def per_kernel(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            t = tag_of(r["Kernel_Name"])
            tot[t] += float(r["Counter_Value"])
            cnt[t] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_latest.json"
    ft, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    wt, wc = per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    for t in sorted(set(ft) | set(wt)):
        n = max(fc[t], wc[t], 1)
        f_kb, w_kb = ft[t] / max(fc[t], 1), wt[t] / max(wc[t], 1)
        res[t] = {"launches": n, "FETCH_SIZE_KB": round(f_kb, 1), "WRITE_SIZE_KB": round(w_kb, 1),
                  "traffic_bytes": int(round(2 * f_kb * 1024 + w_kb * 1024))}
    doc = {"collected": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (no other tracing), "
                        "VDQN_NO_OVERLAP=1, bench.py --steps 2 --warmup 1 --no-profile --no-cpu-baseline, batch 256 bf16",
           "note": "FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads 1/2 of wide coalesced streaming reads "
                   "(MI355X_MICROARCH.md, HBM) so hbm_read = 2*FETCH_SIZE*1024; WRITE_SIZE exact. Infinity-Cache hits are included in FETCH_SIZE.",
           "per_launch": res}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for t, v in res.items():
        print(f"{t:32s} x{v['launches']:4d}  {v['traffic_bytes'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
