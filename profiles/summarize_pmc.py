"""Summarise rocprofv3 --pmc CSV output (one directory per pass) into per-kernel averages per dispatch."""
import csv, glob, json, os, sys, collections

def short(name):
    for k in ("render_bwd_kernel", "render_fwd_kernel", "preprocess_bwd_kernel", "preprocess_geom_kernel", "sh_colour_kernel", "preprocess_kernel", "tile_emit_kernel",
              "tile_ranges_kernel", "project_cull_kernel", "coarse_pairs_kernel", "pack_global_kernel", "bin_scatter_kernel",
              "bin_offsets_kernel", "loss_forward_kernel", "loss_backward_kernel", "tile_depth_sort_wave_kernel", "tile_depth_sort_kernel", "publish_counts_kernel"):
        if k in name:
            return k
    if "rocprim" in name:
        return "rocprim:" + ("radix" if "radix" in name or "onesweep" in name or "histogram" in name else
                             "scan" if "scan" in name else "other")
    return name[:40]

def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    out = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}
    print(json.dumps(out, indent=1, sort_keys=True))

if __name__ == "__main__":
    main(sys.argv[1])
