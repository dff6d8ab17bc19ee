#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/...) into the small files
committed under profiles/.

    python tools/summarize_profile.py --stats DIR --fetch DIR --write DIR --tag r1_bench_c2_d010

stats dir : rocprofv3 --kernel-trace --stats --output-format csv
fetch dir : rocprofv3 --pmc FETCH_SIZE   (separate pass, MI355X_MICROARCH.md "rocprofv3 PMC slots")
write dir : rocprofv3 --pmc WRITE_SIZE   (separate pass)
Writes profiles/<tag>_kernel_stats.csv (this library's kernels + runtime fills)
and profiles/<tag>_traffic.json (per-launch means; FETCH_SIZE/WRITE_SIZE are in KiB).
"""
import argparse
import collections
import csv
import glob
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit(f"no file matches {pattern}")
    return max(files, key=os.path.getmtime)  # a directory may hold several runs: the newest


def short(name):
    name = name.replace("sputnik_hip::(anonymous namespace)::", "")
    return name.split("(")[0].replace("void ", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--kernel", default="spmm_flat_kernel")
    ap.add_argument("--bench-log", default=None,
                    help="stdout of the profiled bench.py run: its JSON line names the build "
                         "(library.build_id) the counters were collected on")
    ap.add_argument("--algorithmic-bytes", type=float, default=None)
    ap.add_argument("--traffic-name", default=None,
                    help="also write profiles/<name> (the file bench.py reads)")
    args = ap.parse_args()
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)

    if args.stats:
        src = one(os.path.join(args.stats, "**", "*_kernel_stats.csv"))
        dst = os.path.join(REPO, "profiles", args.tag + "_kernel_stats.csv")
        with open(src) as f, open(dst, "w", newline="") as g:
            rd = csv.DictReader(f)
            wr = csv.DictWriter(g, fieldnames=rd.fieldnames)
            wr.writeheader()
            for row in rd:
                if "sputnik_hip" in row["Name"] or "rocclr" in row["Name"]:
                    row["Name"] = short(row["Name"])
                    wr.writerow(row)
        print("wrote", dst)

    traffic = {"units": "FETCH_SIZE / WRITE_SIZE are KiB per launch (rocprofv3); bytes = KiB * 1024",
               "kernels": {}}
    if args.bench_log:
        for line in open(args.bench_log):
            if line.startswith("{"):
                try:
                    traffic["build_id"] = json.loads(line)["library"]["build_id"]
                except (ValueError, KeyError):
                    pass
    for label, d in (("FETCH_SIZE", args.fetch), ("WRITE_SIZE", args.write)):
        if not d:
            continue
        src = one(os.path.join(d, "**", "*_counter_collection.csv"))
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(src)):
            if "sputnik_hip" in row["Kernel_Name"] and row["Counter_Name"] == label:
                agg[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            traffic["kernels"].setdefault(k, {})[label + "_KiB_mean"] = sum(v) / len(v)
            traffic["kernels"][k][label + "_launches"] = len(v)
    dom = next((v for k, v in traffic["kernels"].items() if k.startswith(args.kernel)), None)
    if dom and "FETCH_SIZE_KiB_mean" in dom and "WRITE_SIZE_KiB_mean" in dom:
        fetch = dom["FETCH_SIZE_KiB_mean"] * 1024
        write = dom["WRITE_SIZE_KiB_mean"] * 1024
        traffic["dominant_kernel"] = args.kernel
        traffic["fetch_bytes_raw"] = fetch
        traffic["write_bytes"] = write
        # gfx950: FETCH_SIZE tallies the L2's 128-byte line fetches at 64 B, i.e. reports half
        # of the bytes read.  Calibrated (tools/fetch_calib.hip, profiles/r2_fetch_calibration.json)
        # on a known 512 MiB read in each access shape this kernel uses -- LDS-DMA dwordx4,
        # dwordx4 to registers, dword, and the 16-dword replicated entry window: the ratio is
        # 0.500 for all four, so the correction applies to the whole fetch.
        traffic["fetch_bytes_corrected_x2"] = 2 * fetch
        traffic["traffic_bytes_per_launch"] = 2 * fetch + write
        if args.algorithmic_bytes:
            traffic["algorithmic_bytes_per_launch"] = args.algorithmic_bytes
            traffic["traffic_over_algorithmic"] = (2 * fetch + write) / args.algorithmic_bytes
    for name in filter(None, (args.tag + "_traffic.json", args.traffic_name)):
        dst = os.path.join(REPO, "profiles", name)
        with open(dst, "w") as f:
            json.dump(traffic, f, indent=1)
        print("wrote", dst)


if __name__ == "__main__":
    main()
