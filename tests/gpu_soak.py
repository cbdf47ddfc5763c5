#!/usr/bin/env python3
"""GPU soak run (not collected by pytest): many Qbist frames through the synchronous encoder, every sample compared with
the reference encoder's (oracle/_ref) output for the same frame, volatile metadata masked.

    python tests/gpu_soak.py [frames_per_seed] [seed ...]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import cfhd_testlib as T


def main():
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seeds = [int(a) for a in sys.argv[2:]] or [1, 10]
    w, h = 1920, 1080
    bad = 0; total = 0
    for fmt, name in ((T.PIX_YUY2, "YUY2"), (T.PIX_2VUY, "2vuy")):
        for seed in seeds:
            frames, pitch = T.qbist_frames(seed, nfr, w, h, fmt)
            mine = T.amd_encode_frames(frames, pitch, w, h, fmt)
            refs = T.ref_encode_frames(frames, pitch, w, h, fmt)
            for i, (a, b) in enumerate(zip(mine, refs)):
                total += 1
                ma, mb = T.mask_volatile_metadata(a), T.mask_volatile_metadata(b)
                if ma != mb:
                    bad += 1
                    x = np.frombuffer(ma, np.uint8); y = np.frombuffer(mb, np.uint8)
                    n = min(len(x), len(y)); d = np.nonzero(x[:n] != y[:n])[0]
                    print("MISMATCH fmt=%s seed=%d frame=%d: %d vs reference %d bytes, first diff at %s, %d differing bytes"
                          % (name, seed, i, len(a), len(b), d[0] if len(d) else "tail", len(d)), flush=True)
    print("soak: %d samples, %d mismatches" % (total, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
