#!/usr/bin/env python3
"""Writes a training configuration in the flat `key: value` form gsplat_parse_config reads (include/gsplat_host.h): the
hyper-parameters BASELINE config 4 names (the reference's base schedule: 7 000 iterations on Mip-NeRF 360 "garden" at
1/4 resolution), from the defaults the host mirror already carries (3dgs_amd/trainer.py DEFAULT_CONFIG, optimizer.py
DEFAULT_LR) plus the dataset keys.  Every key of ConfigParameters is required by the parser.

    python tools/write_config.py garden.yaml [--extended] [key=value ...]      e.g. dataset_path=bicycle

--extended: the reference's 30 000-iteration schedule (print every 500, background / opacity reset / density control
until 10 000 / 15 000 / 15 000).
"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DATASET = dict(dataset_path="garden", downsample_factor=4, output_dir="splat_output", print_interval=100,
               test_eval_interval=500, test_split_ratio=8, initial_opacity=0.2, initial_scale_num_neighbors=3,
               initial_scale_factor=0.8, max_initial_scale=0.1, use_sh_precompute=True)
ORDER = ("dataset_path downsample_factor output_dir print_interval test_eval_interval test_split_ratio initial_opacity "
         "initial_scale_num_neighbors initial_scale_factor max_initial_scale near_thresh mh_dist cull_mask_padding num_iters "
         "ssim_frac base_lr xyz_lr_multiplier_init xyz_lr_multiplier_final quat_lr_multiplier scale_lr_multiplier "
         "opacity_lr_multiplier rgb_lr_multiplier sh_lr_multiplier use_background use_background_end use_sh_precompute "
         "max_sh_band add_sh_band_interval reset_opacity_interval reset_opacity_value reset_opacity_start reset_opacity_end "
         "use_split use_clone use_delete adaptive_control_start adaptive_control_end adaptive_control_interval max_gaussians "
         "delete_opacity_threshold uv_grad_threshold split_scale_factor").split()


def main():
    trainer = importlib.import_module("3dgs_amd.trainer")
    cfg = dict(trainer.DEFAULT_CONFIG, **DATASET)
    args = sys.argv[2:]
    if "--extended" in args:
        args.remove("--extended")
        cfg.update(print_interval=500, num_iters=30000, use_background_end=10000, reset_opacity_end=15000,
                   adaptive_control_end=15000)
    for kv in args:
        k, v = kv.split("=", 1)
        if k not in cfg:
            raise SystemExit(f"unknown key {k}")
        cfg[k] = v
    fmt = lambda v: ("true" if v else "false") if isinstance(v, bool) else str(v)
    with open(sys.argv[1], "w") as f:
        f.write("".join(f"{k}: {fmt(cfg[k])}\n" for k in ORDER))
    print(f"wrote {len(ORDER)} keys to {sys.argv[1]}")


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    main()
