#!/usr/bin/env python3
"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, with --kernel-trace only)
into per-kernel HBM traffic per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes
for gfx950: FETCH_SIZE under-reports wide coalesced reads by exactly 2x (TCC_EA0_RDREQ tallied at 64 B for
128-B requests) -> doubled; WRITE_SIZE is used as reported (it matches the known output bytes of the tower
kernel exactly: 57,600 KiB = 720*1280*32*2 B).  Units in the CSV: KiB.

    python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [match] [pairs_per_launch] [exclude]
(exclude: kernels whose name contains it do not count as the dominant kernel, e.g. ", true>" = the tail form of the streamed block)
"""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    match = sys.argv[4] if len(sys.argv) > 4 else "k_ref_conv_f16"
    pairs_per_launch = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    exclude = sys.argv[6] if len(sys.argv) > 6 else None
    out = {"unit": "bytes per launch", "correction": "2 x FETCH_SIZE (gfx950) + WRITE_SIZE, KiB -> bytes", "kernels": {}}
    tot_b, tot_n = 0.0, 0
    for k in sorted(fetch):
        n, f = fetch[k]
        w = write.get(k, (0, 0.0))[1]
        b = (2.0 * f + w) * 1024.0
        short = k.split("(")[0].replace("void ", "")
        out["kernels"][short] = {"launches": n, "fetch_kib_raw": f, "write_kib": w, "hbm_bytes_per_launch": b}
        if match in k and not (exclude and exclude in k):
            tot_b += b * n
            tot_n += n
    out["dominant_match"] = match
    out["dominant_exclude"] = exclude
    out["pairs_per_launch"] = pairs_per_launch
    out["dominant_avg_hbm_bytes_per_launch"] = tot_b / tot_n if tot_n else None
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out["kernels"].items()}, indent=0))
    print("dominant avg MB/launch:", out["dominant_avg_hbm_bytes_per_launch"] / 1e6)


if __name__ == "__main__":
    main()
