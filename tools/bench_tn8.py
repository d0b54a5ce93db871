#!/usr/bin/env python3
"""bench.py with the eight-wave weight-gradient kernel selected (A/B against the default four-wave kernel)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-pytorch_amd"))
import vitk  # noqa: E402

vitk.lib.vitk_debug_tn_w4.argtypes = [ctypes.c_int]
vitk.lib.vitk_debug_tn_w4.restype = None
vitk.lib.vitk_debug_tn_w4(0)
import bench  # noqa: E402

bench.main()
