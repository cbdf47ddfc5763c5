#!/usr/bin/env python3
"""Regenerates tests/golden/ from the unmodified reference (oracle/_ref/libcfhd_ref.so must be built: `make -C oracle ref`).
Run from the repo root in the build container (the reference tree is not available on the GPU box)."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from cfhd_testlib import *

here = os.path.dirname(os.path.abspath(__file__))
g = {}
frames, pitch = qbist_frames(10, 1)
s = ref_encode_frames(frames, pitch, 1920, 1080)[0]
g["qbist_seed10_frame1_size"] = len(s)
g["qbist_seed10_frame1_masked_sha256"] = hashlib.sha256(mask_volatile_metadata(s)).hexdigest()
g["qbist_seed10_frame1_input_sha256"] = hashlib.sha256(frames[0].tobytes()).hexdigest()
w, h, seed = 192, 96, 17
f, p = synth_yuy2(w, h, seed)
s = ref_encode_frames([f], p, w, h)[0]
open(os.path.join(here, "ref_192x96_seed17.cfhd"), "wb").write(mask_volatile_metadata(s))
g["small"] = dict(width=w, height=h, seed=seed, sample_file="ref_192x96_seed17.cfhd", input_sha256=hashlib.sha256(f.tobytes()).hexdigest())
out, rp = ref_decode_sample(s, w, h)
g["small"]["ref_decode_psnr"] = round(float(psnr_yuy2(out.reshape(h, rp)[:, :w * 2], f.reshape(h, p))), 3)
json.dump(g, open(os.path.join(here, "golden.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(g, indent=1))
