"""traffic.json (read by bench.py into roofline.traffic / roofline_valu_issue) from a PMC summary.
HBM bytes per launch of the compositing kernels = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, FETCH doubled as
MI355X_MICROARCH.md prescribes for gfx950; wave-level VALU instructions per launch (SQ_INSTS_VALU); and the counters'
own VALU-busy figure, rocprof's VALUBusy = 4 * SQ_ACTIVE_INST_VALU / (SIMDs * kernel cycles) with kernel cycles =
GRBM_GUI_ACTIVE / 8 XCDs (a wave holds its SIMD's VALU for one quad-cycle per instruction; values above 1 mean two waves'
instructions overlapping in the pipe, i.e. a saturated issue port).
The file records the hash of the device sources the measured library was built from (3dgs_amd/_lib.py
library_source_hash: the hash embedded in the binary); bench.py reports
the counter-derived figures only when that hash matches the library it runs.
usage: python3 profiles/make_traffic.py profiles/r02_pmc_summary.json > profiles/traffic.json"""
import importlib, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def entries(d):
    out = {}
    for name, k in (("render_backward", "render_bwd_kernel"), ("render_forward", "render_fwd_kernel")):
        v = d[k]
        out[name] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
        out[name + "_raw"] = {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                              "TCC_EA0_ATOMIC_64B_requests": v.get("TCC_EA0_ATOMIC_sum", 0.0)}
        if "SQ_INSTS_VALU" in v:
            out[name + "_valu_insts"] = int(v["SQ_INSTS_VALU"])
        if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
            cycles = v["GRBM_GUI_ACTIVE"] / 8.0
            out[name + "_valu_busy"] = 4.0 * v["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles)
            out[name + "_profiled_kernel_cycles"] = cycles
    return out


# usage: make_traffic.py <headline summary.json> [<workload>=<summary.json> ...]
# the hash the LOADED binary carries (gsplat_source_hash): what the counters were measured on
out = {"source_sha16": importlib.import_module("3dgs_amd._lib").library_source_hash()}
out.update(entries(json.load(open(sys.argv[1]))))
for extra in sys.argv[2:]:  # r05: the same counters on other workloads (tools/workload_stats.py <workload> under the PMC passes)
    name, path = extra.split("=", 1)
    out.setdefault("workloads", {})[name] = entries(json.load(open(path)))
print(json.dumps(out, indent=1))
