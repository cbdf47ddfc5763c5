#!/usr/bin/env python3
"""tools/host_fed_probe.py -- the host-fed frame queue alone (bench.py host_fed), for a copy / kernel timeline: rocprofv3 --kernel-trace --memory-copy-trace -- python tools/host_fed_probe.py [batch depth steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import bench, cfhd_testlib as T
batch, depth, steps = (int(x) for x in (sys.argv[1:4] + ["128", "4", "16"][len(sys.argv) - 1:]))
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
print(bench.host_fed("1080p", frames, pitch, batch=batch, depth=depth, steps=steps, warmup=depth))
