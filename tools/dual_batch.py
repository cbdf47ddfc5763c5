#!/usr/bin/env python3
"""Experiment: N batch objects of batch/N frames, each driven by its own host thread (own HIP streams), against one batch object of `batch` frames.
Does the GPU gain from kernels of different bounds (VALU-bound entropy kernels, HBM-bound transforms) running side by side?
  python tools/dual_batch.py [total_frames=512] [steps=20]"""
import ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import cfhd_testlib as T

def run(nobj, total, steps):
    L = bench.batch_api()
    W, H = 1920, 1080
    frames, pitch = T.qbist_frames(10, 32, W, H, T.PIX_YUY2)
    per = total // nobj
    objs = []
    for k in range(nobj):
        b = L.cfhd_amd_batch_create_ex(W, H, T.PIX_YUY2, 0, 0, T.QUALITY_FILMSCAN1, per, 8, 0)
        assert b, T.amd_last_error()
        for i in range(per):
            assert L.cfhd_amd_batch_upload(b, i, frames[(k * per + i) % 32].ctypes.data_as(ctypes.c_void_p), pitch) == 0
        objs.append(b)
    def work(b, n):
        for _ in range(n):
            assert L.cfhd_amd_batch_roundtrip(b) > 0
    for phase, n in (("warmup", 3), ("timed", steps)):
        ts = [threading.Thread(target=work, args=(b, n)) for b in objs]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        dt = time.perf_counter() - t0
    print("objects %d x %d frames: %.1f fps (%.3f ms per %d frames)" % (nobj, per, total * steps / dt, dt / steps * 1e3, total), flush=True)
    for b in objs: L.cfhd_amd_batch_destroy(b)

if __name__ == "__main__":
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    for nobj in (1, 2, 4):
        run(nobj, total, steps)
