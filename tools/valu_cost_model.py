#!/usr/bin/env python3
"""VALU issue cost of render_bwd's trip loop, instruction by instruction: compiles gs_render.hip device-only, takes the
innermost loop of render_bwd_kernel<true,true,*> and weighs every VALU instruction with the per-SIMD issue cost measured
on MI355X (profiles/microbench/r01_valu_rate*.txt, cycles per wave64 instruction at 8 waves per SIMD, 2.4 GHz nominal).
Prints cycles per trip and, with the trip count of the benchmark scene, the share of the kernel's SIMD cycles.

    python tools/valu_cost_model.py [trips] [kernel_cycles]      defaults: 3.6e6 trips, 825e3 cycles (r02 profiles)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST = [  # (regex on the mnemonic + operands, cycles, class)
    (r"^v_(exp|rcp|rsq|sqrt|log|sin|cos)_f32", 8.3, "transcendental"),
    (r"^v_pk_", 5.3, "packed f32"),
    (r"_dpp|quad_perm|row_", 4.4, "DPP"),
    (r"^v_(cndmask|cmp|cmpx|min|max|med3|readlane|readfirstlane)", 4.4, "select / compare / min-max"),
    (r"^v_mad_u64_u32|^v_cvt_f64|^v_.*_f64", 8.0, "64-bit (assumed quarter rate)"),
    (r"^v_mov_b32", 2.5, "move"),
    (r"^v_", 2.9, "plain f32 / i32"),
]


def main():
    trips = float(sys.argv[1]) if len(sys.argv) > 1 else 3.6e6
    kernel_cycles = float(sys.argv[2]) if len(sys.argv) > 2 else 825e3
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "r.s")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-munsafe-fp-atomics",
                               "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "--cuda-device-only", "-S", "-o", out,
                               os.path.join(ROOT, "3dgs_amd", "csrc", "gs_render.hip")], stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith("_ZN2gs17render_bwd_kernelILb1ELb1"))
    body = []
    for l in text[start:]:
        body.append(l)
        if "s_endpgm" in l:
            break
    # the trip loop = the innermost loop that contains the f64 LDS atomic
    heads = [i for i, l in enumerate(body) if "Inner Loop Header: Depth=2" in l]
    atom = next(i for i, l in enumerate(body) if "ds_add_f64" in l)
    head = max(h for h in heads if h < atom)
    end = next(i for i in range(atom, len(body)) if re.match(r"^\.LBB\d+_\d+:", body[i]) and "Depth=2" not in body[i])
    total, by_class, n = 0.0, {}, 0
    for l in body[head:end]:
        ins = l.strip()
        if not ins.startswith("v_"):
            continue
        for rx, cyc, cls in COST:
            if re.search(rx, ins):
                total += cyc; n += 1
                c = by_class.setdefault(cls, [0, 0.0]); c[0] += 1; c[1] += cyc
                break
    print(f"trip loop: {n} VALU instructions, {total:.0f} issue cycles per trip when every instruction executes")
    for cls, (k, cyc) in sorted(by_class.items(), key=lambda kv: -kv[1][1]):
        print(f"  {cls:32s} {k:3d} instructions {cyc:6.1f} cycles")
    simd = 1024
    print(f"{trips:.3g} trips x {total:.0f} cycles / {simd} SIMDs = {trips * total / simd / 1e3:.0f} k cycles of VALU issue per SIMD "
          f"= {100 * trips * total / simd / kernel_cycles:.0f} % of the kernel's {kernel_cycles / 1e3:.0f} k cycles "
          f"(the loop only; staging, list building and flush add their own)")


if __name__ == "__main__":
    main()
