"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a bench.py run into per-kernel HBM-side bytes per launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/traffic/f -o p --output-format csv -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/traffic/w -o p --output-format csv -- python bench.py ...
    python tools/pmc_traffic.py gpurun_out/traffic profiles/r02_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so it is doubled
(MI355X_MICROARCH.md, section HBM).  Infinity-Cache hits are included (fabric-side counters).
"""
import collections, csv, glob, json, sys
root, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: {"fetch_kib": 0.0, "write_kib": 0.0, "n_f": 0, "n_w": 0})
def fam(name, row=None):
    # the encoder layers' launches: their NT GEMMs run the loader-wave ping-pong kernel (gemm_nt_ld_kernel; gemm_nt_pp_kernel with TIMHIP_GEMM_LD=0), their grouped weight gradient
    # wgrad_ld_kernel (wgrad_pp_kernel with TIMHIP_WGRAD_LD=0); the front end / heads / small models use gemm_nt_h16_kernel, gemm_nt_group_kernel, wgrad_group_kernel
    for k in ("gemm_nt_p8_kernel", "attn_keep_bits_kernel", "gemm_nt_ldp_kernel", "gemm_nt_ld_kernel", "gemm_nt_pp_kernel", "wgrad_p8_kernel", "wgrad_ld_kernel", "wgrad_pp_kernel", "gemm_nt_h16_kernel", "gemm_nt_group_kernel", "wgrad_group_kernel", "wgrad_tn_kernel",
              "attn_fwd_mfma", "attn_bwd_mfma", "ln_fwd_kernel", "ln_bwd_kernel", "wgrad_reduce_kernel", "cast_weights_kernel",
              "attn_bwd_rows", "attn_bwd_keys", "ln_fwd8_kernel", "split3_kernel", "grad_scale_kernel"):
        if k in name:
            return k
    return None
for sub, key, cnt in (("f", "fetch_kib", "n_f"), ("w", "write_kib", "n_w")):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            k = fam(r["Kernel_Name"], r)
            if k and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[k][key] += float(r["Counter_Value"]); acc[k][cnt] += 1
res = {}
for k, v in acc.items():
    fb = 2.0 * 1024 * v["fetch_kib"] / max(1, v["n_f"]); wb = 1024 * v["write_kib"] / max(1, v["n_w"])
    res[k] = {"launches": v["n_f"], "read_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
              "bytes_per_launch": round(fb + wb)}
json.dump({"note": "rocprofv3 FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per launch, bench.py C2a B=64", "kernels": res},
          open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
