#!/usr/bin/env python3
"""Condenses one rocprofv3 --pmc pass (tools/collect_pmc_script.sh / collect_pmc.sh: eight SQ
counters) into a small JSON for profiles/: per kernel of this library the mean counters per
launch (in millions, summed over all waves), the share of wave cycles parked at a wait and
the share in which a vector / an LDS instruction is being issued.

    python tools/summarize_pmc.py gpurun_out/pmc_<tag> profiles/<name>.json ["note"]
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = name.replace("sputnik_hip::(anonymous namespace)::", "").replace("sputnik_hip::", "")
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0] if "<" not in name.split("(")[0] else name[:name.find(">(") + 1] if ">(" in name else name
    name = re.sub(r"Geometry<(\d+), (\d+), (\d+), (\d+), (\d+)>", r"Geometry<\1,\2,\3,\4,\5>", name)
    return name[:110]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    f = max(glob.glob(src + "/**/*counter_collection.csv", recursive=True))
    total = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "sputnik_hip" not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        total[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    out = {}
    for k, v in sorted(total.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        n = max(1, len(launches[k]))
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1.0
        row = {c: round(x / n / 1e6, 3) for c, x in sorted(v.items())}
        row["launches"] = n
        row["wait_frac"] = round(v.get("SQ_WAIT_ANY", 0) / wc, 3)
        row["valu_active_frac_of_wave_cycles"] = round(v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3)
        row["lds_active_frac_of_wave_cycles"] = round(v.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3)
        out[k] = row
    json.dump({"note": note or "rocprofv3 --pmc (8 SQ counters); means per launch in MILLIONS summed over all waves",
               "kernels": out}, open(dst, "w"), indent=1)
    for k, row in out.items():
        print(f"{k[:80]:80s} n {row['launches']:4d} wait {row['wait_frac']:.2f} valu {row['valu_active_frac_of_wave_cycles']:.2f} "
              f"lds {row['lds_active_frac_of_wave_cycles']:.2f}")


if __name__ == "__main__":
    main()
