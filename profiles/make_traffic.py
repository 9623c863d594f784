"""traffic.json (read by bench.py into roofline.traffic) from a PMC summary: HBM bytes per launch of the compositing
kernels = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950.
usage: python3 profiles/make_traffic.py profiles/r01_pmc_summary.json > profiles/traffic.json"""
import json, sys

d = json.load(open(sys.argv[1]))
out = {}
for name, k in (("render_backward", "render_bwd_kernel"), ("render_forward", "render_fwd_kernel")):
    v = d[k]
    out[name] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    out[name + "_raw"] = {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                          "TCC_EA0_ATOMIC_64B_requests": v.get("TCC_EA0_ATOMIC_sum", 0.0)}
print(json.dumps(out, indent=1))
