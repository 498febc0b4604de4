#!/usr/bin/env python3
"""Which kernels of tim_amd/csrc spill registers?  (hipcc cross-compiles: runs in the build container, no GPU.)

    python tools/check_spills.py [file.hip ...]          default: every .hip of the library (report)
    python tools/check_spills.py --gate                  the budget `__graft_entry__.build()` enforces: exit 1 on a violation
                                                         (reads the remarks the Makefile's compile left in csrc/build/, compiles
                                                         a unit itself only when they are missing or stale)

The one-block-per-CU GEMM kernels run three waves per SIMD at 168 VGPRs: their epilogues sit at that limit, and an innocent
change (a loop around the kernel body, a value kept live across the epilogue) has made the compiler spill 20-132 registers
there without any warning - a 37 -> 54 us surprise that broke an A/B baseline twice.  The gate turns
that into a build failure: every kernel family named in BUDGET may spill at most the stated number of VGPRs in any of its
instances (the numbers are the spill counts of the measured, shipped kernels - an epilogue that scratch-stores a handful of
registers once per tile was measured and accepted; anything above it was not)."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tim_amd", "csrc")

# translation unit -> [(regex on the MANGLED kernel name, most VGPRs an instance may spill)]; first match decides; kernels no
# pattern matches are reported only.  EPI numbers: include/timhip.h (0 STORE_T, 1 RELU_T, 2 STORE_F32, 4 DROP_RES_F32, 5 ADD_F32,
# 11 GELU_DROP_G2, 12 MULAUX_T).
BUDGET = {
    "gemm.hip": [(r"gemm_nt_h16_kernel", 0), (r"gemm_nt_group_kernel", 0)],
    "gemm_pp.hip": [(r"gemm_nt_ld_kernel", 0), (r"gemm_nt_pp_kernel", 0),
                    (r"gemm_nt_p8_kernel", 0),   # round 6: 196 (256-row tile) / 239 (320-row tile) VGPRs of 256
                    # the tile walk's epilogues re-derive their lane arithmetic per tile and scratch-store a few registers once per
                    # tile (measured with these counts: the walk wins 0.5 % of the step); the residual
                    # epilogue (EPI 4) is not on any BASELINE config's path (its shapes run one round: gemm_nt_ld_kernel)
                    (r"gemm_nt_ldp_kernelIDF16[_b]Li4E", 35), (r"gemm_nt_ldp_kernelIDF16[_b]Li5E", 12), (r"gemm_nt_ldp_kernel", 8)],
    "wgrad_pp.hip": [(r"wgrad_ld_kernel", 0), (r"wgrad_pp_kernel", 0), (r"wgrad_p8_kernel", 0)],
    "wgrad.hip": [(r"wgrad_group_kernel", 0), (r"wgrad_tn_kernel", 0)],
    "attention_mfma.hip": [(r"attn_fwd_mfma", 0)],
    # the fp16 fused backward (C2a / C3 / C4), with the kernel's own draws and with the keep-bits of round 6
    "attention_bwd2.hip": [(r"attn_bwd_rowsIDF16_Li128ELi4ELb1E", 0), (r"attn_bwd_rowsIDF16[_b]Li128ELi4ELb1ELb1ELb1E", 0)],
    "rowops.hip": [(r"ln_fwd8", 0), (r"ln_bwd_kernel", 0)],
}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return out[:len(names)]
    except Exception:  # noqa: BLE001
        return names


def kernel_spills(src, extra=(), demangled=True, use_build=False):
    """-> [(kernel name, VGPRs spilled, VGPRs used)] of one translation unit"""
    remarks = os.path.join(CSRC, "build", src[:-4] + ".remarks")
    if use_build and os.path.exists(remarks) and os.path.getmtime(remarks) >= os.path.getmtime(os.path.join(CSRC, src)):
        txt = open(remarks).read()          # what the Makefile's own compile of this unit reported
    else:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
               "-Rpass-analysis=kernel-resource-usage", *extra]
        txt = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], {}
    for line in txt.split("\n"):
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            continue
        m = re.search(r"remark: .*?\bVGPRs: (\d+)", line)
        if m and cur:
            cur["vgprs"] = int(m.group(1))
        m = re.search(r"remark: .*?VGPRs Spill: (\d+)", line)
        if m and cur:
            rows.append((cur["name"], int(m.group(1)), cur.get("vgprs", -1)))
            cur = {}
    if not demangled:
        return rows
    names = demangle([r[0] for r in rows])
    return [(n, s, v) for n, (_, s, v) in zip(names, rows)]


def gate():
    bad = []
    for src, pats in BUDGET.items():
        rows = kernel_spills(src, demangled=False, use_build=True)
        if not rows:
            bad.append("%s: the compiler reported no kernels (did -Rpass-analysis change?)" % src)
            continue
        seen = set()
        for n, s, _ in rows:
            for i, (pat, limit) in enumerate(pats):
                if re.search(pat, n):
                    seen.add(i)
                    if s > limit:
                        bad.append("%s: %s spills %d VGPRs (budget %d)" % (src, n[:160], s, limit))
                    break
        for i, (pat, _) in enumerate(pats):
            if i not in seen:
                bad.append("%s: no kernel matches %r - update tools/check_spills.py:BUDGET" % (src, pat))
    for b in bad:
        print("check_spills: " + b, file=sys.stderr)
    return 1 if bad else 0


def report(files):
    for f in files:
        rows = kernel_spills(f)
        sp = [(n, s, v) for n, s, v in rows if s > 0]
        print("%s: %d of %d kernels spill VGPRs" % (f, len(sp), len(rows)))
        for n, s, v in sp:
            print("    %3d spilled (%3d used)  %s" % (s, v, n[:170]))


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--gate":
        sys.exit(gate())
    report(args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")))
