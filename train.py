#!/usr/bin/env python3
"""python train.py <path_to_config_file.yaml> <path_to_root_directory>  -- the reference's `gsplat` binary
(src/main.cpp:10-98) on the MI355X-native library; see 3dgs_amd/app.py."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    sys.exit(importlib.import_module("3dgs_amd.app").main(sys.argv))
