#!/usr/bin/env python3
"""Renders two frames of the capture bench's scene and saves the last recorded image as an .npy file (what
tools/probe_lz4 takes).  usage: python3 tools/dump_capture_frame.py OUT.npy"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np

import bench_capture

_, _, last = bench_capture.run_capture(frames=2, streams=2, out="/tmp/rh_cap_dump", warmup=0)
np.save(sys.argv[1], np.ascontiguousarray(last))
print(last.shape, last.dtype)
