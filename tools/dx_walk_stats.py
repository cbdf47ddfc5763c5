#!/usr/bin/env python3
"""tools/dx_walk_stats.py -- analysis (build container, needs oracle/_ref): lane statistics of k_dec_tiles' walk on the bench's Qbist samples.
   python tools/dx_walk_stats.py [nframes] [tile]"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfhd_testlib as T
so = os.path.join(ROOT, "tools", "_build", "libdx_walk_stats.so")
csrc = os.path.join(ROOT, "cineform-sdk_amd", "csrc")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ROOT, "tools", "dx_walk_stats.cpp")):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "tests", "hipemu"), "-I" + csrc, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "dx_walk_stats.cpp")] + [os.path.join(csrc, f) for f in ("cfhd_tables.cpp", "cfhd_bitstream.cpp", "cfhd_metadata.cpp")] + ["-o", so])
L = ctypes.CDLL(so)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
W, H = 1920, 1080
frames, pitch = T.qbist_frames(10, n, W, H)
out = (ctypes.c_uint64 * 40)(); hist = (ctypes.c_uint64 * 64)(); lens = (ctypes.c_uint64 * 192)()
for f in frames:
    s = T.ref_encode_frames([f], pitch, W, H)[0]
    rc = L.dx_walk_stats(s, ctypes.c_size_t(len(s)), tile, out, hist, lens)
    assert rc == 0, rc
names = ["tiles", "rounds", "lanes_inside", "lane_steps", "wave_steps", "wave_steps_long", "lane_steps_long", "pieces", "coefs", "bits", "flat_wave_steps", "flat_long", "nonzero", "short_wave_steps", "long_rounds", "flat_rounds"]
for g, nm in enumerate(("level 1", "levels 2+3")):
    o = {k: out[16 * g + i] / n for i, k in enumerate(names)}
    print(nm, {k: round(v, 1) for k, v in o.items()})
    print("   bits/coef %.3f  nonzero %.2f%%  pieces/tile %.1f  lanes/round %.1f  steps/piece %.2f  wave steps/round %.2f (long in %.2f)  lane utilisation %.2f  flat: wave steps/round %.2f  ideal wave steps %.0f vs %.0f vs flat %.0f"
          % (o["bits"] / o["coefs"], 100 * o["nonzero"] / o["coefs"], o["pieces"] / o["tiles"], o["lanes_inside"] / o["rounds"], o["lane_steps"] / o["lanes_inside"], o["wave_steps"] / o["rounds"],
             o["wave_steps_long"] / o["rounds"], o["lane_steps"] / (64 * o["wave_steps"]), o["flat_wave_steps"] / o["flat_rounds"], o["lane_steps"] / 64, o["wave_steps"], o["flat_wave_steps"]))
print("flat rounds, two phases: short wave steps %.0f, long wave steps %.0f, phases %.0f per frame" % (out[32] / n, out[33] / n, out[34] / n))
print("steps per piece histogram:", [hist[i] // n for i in range(40)])
for g, nm in enumerate(("level 1", "levels 2+3")):
    for k, kn in enumerate(("runs", "values")):
        h = [lens[(g * 3 + k) * 32 + i] for i in range(32)]; tot = sum(h) or 1
        print(nm, kn, "per frame %d; share by length incl. sign:" % (tot // n), {i: round(100.0 * h[i] / tot, 2) for i in range(32) if h[i]}, " > 11 bits: %.2f%%, > 12: %.2f%%, > 13: %.2f%%, > 18: %.3f%%" % tuple(100.0 * sum(h[j:]) / tot for j in (12, 13, 14, 19)))
