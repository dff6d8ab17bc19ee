#!/usr/bin/env python3
"""Renders the result tables of DESIGN.md section 8 from a bench.py JSON line.
    python tools/results_table.py profiles/r2b_bench_n1.json"""
import json
import sys

A100 = {0.5: 5267, 0.25: 4365, 0.2: 4532, 0.15: 4059, 0.1: 3416, 0.05: 2725}  # reference README.md:50-55


def main():
    d = json.load(open(sys.argv[1]))
    print("| density | nnz | ms (median of 100) | ms (min) | kernel ms | eff. GFLOP/s | alg. GB/s | frac of 8 TB/s | frac of 157.3 TF | A100 Sputnik GFLOP/s |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for s in d["sweep"]:
        b = "**" if s["density"] == 0.1 else ""
        print(f"| {b}{s['density']:.2f}{b} | {s['nnz']:,} | {b}{s['ms']:.3f}{b} | {s['ms_min']:.3f} | {s['kernel_ms']:.3f} | "
              f"{b}{s['gflops']:,.0f}{b} | {s['alg_gbs']:.0f} | {s['hbm_frac']:.3f} | {s['valu_frac']:.3f} | {A100[s['density']]:,} |".replace(",", " "))
    print()
    print(f"bench line: value {d['value']:.0f} GFLOP/s, {d['ms_per_step']:.4f} ms per step, kernel {d['roofline']['kernel_ms']:.4f} ms, "
          f"roofline.frac {d['roofline']['frac']:.4f}, roofline_valu.frac {d['roofline_valu']['frac']:.3f}, traffic {d['roofline']['traffic']}")
    o = d["other_ops"]
    for k in sorted(o):
        print(k, json.dumps(o[k]))
    print("per_gpu_share", d.get("per_gpu_share_of_multi_gpu_runs"))
    c = d["cpu_baseline"]
    print("cpu", c["value"], c["cores"], c["cpu_model"], c.get("dense_torch_matmul"))


if __name__ == "__main__":
    main()
