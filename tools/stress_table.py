#!/usr/bin/env python3
"""Markdown tables of a tools/stress_parity.py summary (profiles/rNN_stress_parity.json) for DESIGN.md 3.7."""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_stress_parity.json"))
cols = ["hip_default", "hip_direct", "oracle_1thr", "oracle_bs1"]
R = d["results"]
names = {"hip_default": "HIP default", "hip_direct": "HIP direct family", "oracle_1thr": "oracle, 1 thread", "oracle_bs1": "oracle, one frame per call"}
print(f"{d['frames']:,} frames, reference pass = {d['reference_pass']}\n")
print("| | " + " | ".join(names[c] for c in cols) + " |")
print("|---|" + "---:|" * len(cols))
print("| frames compared (logits) | " + " | ".join(f"{R[c]['frames']:,}" for c in cols) + " |")
print("| largest logit difference | " + " | ".join(f"{R[c]['max_abs_logit_diff']:.2e}" for c in cols) + " |")
print("| mean logit difference | " + " | ".join(f"{R[c]['mean_abs_logit_diff']:.2e}" for c in cols) + " |")
for key, title in (("loc", "`loc` arg-max decisions"), ("ids", "`ids` arg-max decisions"), ("heat", "heat-map arg-max decisions")):
    print(f"| {title} (decided / differ) | " + " | ".join(f"{sum(R[c]['histogram'][key]['decided']):,} / **{sum(R[c]['histogram'][key]['differs'])}**" for c in cols) + " |")
print("| cells whose final decision (fires / id / offset) differs | " + " | ".join(str(R[c]["cells_decided_differently"]) for c in cols) + " |")
print("| frames differing end to end (ids + cells + xy) | " + " | ".join(f"{R[c]['end_to_end']['mismatched_frames']} of {R[c]['end_to_end']['frames']:,}" for c in cols) + " |")
print()
print("| oracle top-2 margin | `loc` + `ids` cells decided | " + " | ".join(names[c] + " differs" for c in cols[:3]) + " |")
print("|---|---:|" + "---:|" * 3)
for i, lab in enumerate(d["buckets"]):
    dec = R["hip_default"]["histogram"]["loc"]["decided"][i] + R["hip_default"]["histogram"]["ids"]["decided"][i]
    print(f"| {lab} | {dec:,} | " + " | ".join(str(R[c]["histogram"]["loc"]["differs"][i] + R[c]["histogram"]["ids"]["differs"][i]) for c in cols[:3]) + " |")
c64 = [c for c in ("oracle_f32_vs_f64", "hip_default_vs_f64", "hip_direct_vs_f64") if c in R]
if c64:
    n64 = {"oracle_f32_vs_f64": "the reference pass (oracle fp32)", "hip_default_vs_f64": "HIP default", "hip_direct_vs_f64": "HIP direct family"}
    print("\nagainst EXACT arithmetic (the oracle's graph in float64 on the same fp32 inputs and weights):\n")
    print("| vs float64 | " + " | ".join(n64[c] for c in c64) + " |")
    print("|---|" + "---:|" * len(c64))
    print("| largest logit error | " + " | ".join(f"{R[c]['max_abs_logit_diff']:.2e}" for c in c64) + " |")
    print("| mean logit error | " + " | ".join(f"{R[c]['mean_abs_logit_diff']:.2e}" for c in c64) + " |")
    print("| arg-max decisions that differ from the exact ones (`loc` + `ids`) | " + " | ".join(
        str(sum(R[c]['histogram']['loc']['differs']) + sum(R[c]['histogram']['ids']['differs'])) for c in c64) + " |")
    print("| cells whose final decision differs from the exact one | " + " | ".join(str(R[c]["cells_decided_differently"]) for c in c64) + " |")
