import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from cfhd_testlib import *
w, h = 320, 240
f = synth_yuy2(w, h, 2)[0]
v = f.reshape(h, w * 2); v[1::2] = np.roll(v[1::2], 8, axis=1)
sample = amd_encode_frames([f], w * 2, w, h, PIX_YUY2, flags=1)[0]
plan = Plan(w, h, progressive=0)
coeffs = host_decode_pyramid(sample, plan)
lo = oracle_inverse_interlaced_yuv422(plan, coeffs, 0)[:h]
hi = oracle_inverse_interlaced_yuv422(plan, coeffs, 1)[:h]
for mode in ("gpu", "host"):
    if mode == "host": os.environ["CFHD_AMD_ENTROPY"] = "host"
    out, pitch, aw, ah = amd_decode_sample(sample, PIX_YUY2)
    img = out.reshape(ah, pitch)[:, : w * 2]
    ok = (img == lo) | (img == hi)
    print(mode, "bad bytes", (~ok).sum(), "of", ok.size, "psnr vs src", psnr_yuy2(img, v))
    if not ok.all():
        bad = ~ok
        print(" bad per row (first 8):", bad.sum(axis=1)[:8], " bad luma/chroma:", bad[:, 0::2].sum(), bad[:, 1::2].sum())
        print(" img row0[:16]", img[0, :16], "lo", lo[0, :16])
        print(" img row1[:16]", img[1, :16], "lo", lo[1, :16])
