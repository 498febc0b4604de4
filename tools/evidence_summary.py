#!/usr/bin/env python3
"""One JSON a reviewer can read without a script (round-4 review, item 7): per kernel family of the C2a step

    us_per_step, launches_per_step, avg_us, GFLOP per step and TFLOP/s (GEMM families), fabric bytes per launch from the PMC
    passes and bytes / duration in TB/s, MFMA-busy share (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs)

plus, for every bench line kept beside it, ms_per_step / value / roofline.frac / eager figures and the box (hostname, load
average at collection time).

    python tools/evidence_summary.py <dir with kernel_stats.csv [pmc_traffic.json] [pmc_sq_summary.txt] [bench_*.json]> <steps in the
        profiled run> <out.json>
"""
import csv
import glob
import json
import os
import re
import socket
import sys

FAMILIES = [   # (family, regex on the kernel name)
    ("nt_gemm", r"gemm_nt_ld_kernel|gemm_nt_ldp_kernel|gemm_nt_pp_kernel|gemm_nt_p8_kernel"),
    ("wgrad_layer", r"wgrad_ld_kernel|wgrad_pp_kernel|wgrad_p8_kernel"),
    ("attention_keep_bits", r"attn_keep_bits"),
    ("attention_fwd", r"attn_fwd"),
    ("attention_bwd", r"attn_bwd"),
    ("layernorm_fwd", r"ln_fwd"),
    ("layernorm_bwd", r"ln_bwd"),
    ("small_gemm", r"gemm_nt_h16_kernel|gemm_nt_group_kernel|gemm_nt_f32"),
    ("small_wgrad", r"wgrad_group_kernel|wgrad_group_reduce|wgrad_tn|slab_reduce"),
    ("weight_refresh", r"cast_weights"),
    ("torch_native", r"at::native|rocclr"),
]
# algorithmic FLOPs per C2a step (B = 64, M = 9920, E = 1024, FF = 2048, 6 layers): DESIGN.md section 4
M, E, FF, L = 9920, 1024, 2048, 6
GF = {"nt_gemm": 2 * 2.0 * M * (3 * E * E + E * E + 2 * E * FF) * L / 1e9,      # forward + input gradients
      "wgrad_layer": 2.0 * M * (3 * E * E + E * E + 2 * E * FF) * L / 1e9}


def fam_of(name):
    for f, rx in FAMILIES:
        if re.search(rx, name):
            return f
    return "row_kernels_and_rest"


def main():
    d, steps, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    fam = {}
    if steps <= 0:   # infer: one grouped weight-gradient launch per encoder layer (round 6: per TWO layers, wgrad_p8_kernel) and step
        with open(os.path.join(d, "kernel_stats.csv")) as f:
            steps = sum(int(r["calls"]) * (2 if "wgrad_p8_kernel" in r["name"] else 1) for r in csv.DictReader(f)
                        if "wgrad_ld_kernel" in r["name"] or "wgrad_p8_kernel" in r["name"]) / float(L)
    with open(os.path.join(d, "kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            if r["name"] == "TOTAL":
                continue
            k = fam.setdefault(fam_of(r["name"]), {"us": 0.0, "calls": 0, "kernels": {}})
            k["us"] += float(r["total_us"])
            k["calls"] += int(r["calls"])
            short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", r["name"])[:60]
            k["kernels"][short] = {"calls_per_step": round(int(r["calls"]) / steps, 2), "avg_us": float(r["avg_us"])}
    traffic = {}
    p = os.path.join(d, "pmc_traffic.json")
    if os.path.exists(p):
        traffic = json.load(open(p)).get("kernels", {})
    busy = {}
    p = os.path.join(d, "pmc_sq_summary.txt")
    if os.path.exists(p):
        cur, vals = None, {}
        for ln in open(p):
            if ln[:2] in ("a ", "b ", "c "):
                cur = ln.split()[1]
                vals[cur] = {}
            elif cur and "per launch" in ln:
                t = ln.split()
                vals[cur][t[0]] = float(t[1])
        for name, v in vals.items():
            if v.get("GRBM_GUI_ACTIVE"):
                share = (v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0) / (v["GRBM_GUI_ACTIVE"] / 8.0)
                busy[re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:48]] = round(share, 3)
    res = {"steps_profiled": steps, "families": {}}
    tot = sum(k["us"] for k in fam.values())
    for f, k in sorted(fam.items(), key=lambda kv: -kv[1]["us"]):
        e = {"us_per_step": round(k["us"] / steps, 1), "launches_per_step": round(k["calls"] / steps, 2),
             "share_of_kernel_time": round(k["us"] / tot, 4)}
        if f in GF:
            e["gflop_per_step"] = round(GF[f], 1)
            e["tflops_by_kernel_duration"] = round(GF[f] / (k["us"] / steps) * 1e3, 1)
            e["frac_of_2500_tflops"] = round(GF[f] / (k["us"] / steps) * 1e3 / 2500.0, 4)
        byts = [(n, t) for n, t in traffic.items() if fam_of(n) == f]
        if byts:
            e["fabric_bytes_per_launch"] = {n: t["bytes_per_launch"] for n, t in byts}
            bl = sum(t["bytes_per_launch"] * t["launches"] for _, t in byts) / max(1, sum(t["launches"] for _, t in byts))
            avg = k["us"] / max(1, k["calls"])
            e["fabric_TBps"] = round(bl / avg / 1e6, 2)
        e["kernels"] = k["kernels"]
        res["families"][f] = e
    res["kernel_time_us_per_step"] = round(tot / steps, 1)
    res["launches_per_step"] = round(sum(k["calls"] for k in fam.values()) / steps, 1)
    res["non_gemm_us_per_step"] = round((tot - fam.get("nt_gemm", {"us": 0})["us"] - fam.get("wgrad_layer", {"us": 0})["us"]) / steps, 1)
    res["mfma_busy_share"] = busy
    res["box"] = {"hostname": socket.gethostname(), "loadavg": list(os.getloadavg()), "cpus": os.cpu_count()}
    res["bench_lines"] = {}
    for p in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
        try:
            b = json.loads([ln for ln in open(p).read().splitlines() if ln.startswith("{")][-1])
        except Exception:  # noqa: BLE001
            continue
        res["bench_lines"][os.path.basename(p)] = {
            "ms_per_step": b.get("ms_per_step"), "value": b.get("value"), "step_mode": (b.get("step_mode") or "")[:32],
            "roofline_frac": (b.get("roofline") or {}).get("frac"), "whole_step_frac": (b.get("whole_step") or {}).get("frac_of_mfma_peak"),
            "eager": b.get("eager"), "forward_only": b.get("forward_only"), "parity": (b.get("parity") or {}).get("max_abs_logit_err"),
            "secondary_replay_ms": {k: (b[k].get("graph_replay") or {}).get("ms_per_step") for k in ("c2a_b8", "c2a_train", "c2b", "c1", "c3", "c4_train") if k in b},
            "repeat_ms": b.get("repeat_ms"), "non_gemm": b.get("non_gemm"), "launches_per_step": b.get("launches_per_step")}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("kernel_time_us_per_step", "launches_per_step", "non_gemm_us_per_step")}))


if __name__ == "__main__":
    main()
