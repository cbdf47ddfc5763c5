#!/usr/bin/env python3
"""PCIe-inclusive rates of the reference's own C ABI on one GPU (host frame in, host sample out / host sample in, host frame out):
synchronous CFHD_EncodeSample / CFHD_DecodeSample, and the asynchronous encoder pool.  These are the numbers a caller that keeps its
frames in host memory sees; bench.py's `value` is the device-resident batched rate.  Prints one JSON line."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_SDMA", "1")
import numpy as np
import cfhd_testlib as T

def main():
    w, h, n = 1920, 1080, 200
    frames = [T.synth_yuy2(w, h, 50 + i)[0] for i in range(8)]
    L = T.product()
    enc = ctypes.c_void_p(); assert L.CFHD_OpenEncoder(ctypes.byref(enc), None) == 0
    assert L.CFHD_PrepareToEncode(enc, w, h, T.PIX_YUY2, T.ENCODED_YUV422, 0, T.QUALITY_FILMSCAN1) == 0
    samples = []
    for warm in (True, False):
        t0 = time.perf_counter()
        for i in range(8 if warm else n):
            f = frames[i % 8]
            assert L.CFHD_EncodeSample(enc, f.ctypes.data_as(ctypes.c_void_p), w * 2) == 0
            p = ctypes.c_void_p(); sz = ctypes.c_size_t()
            assert L.CFHD_GetSampleData(enc, ctypes.byref(p), ctypes.byref(sz)) == 0
            if warm: samples.append(ctypes.string_at(p, sz.value))
        t_enc = time.perf_counter() - t0
    L.CFHD_CloseEncoder(enc)
    dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
    sbs = [ctypes.create_string_buffer(s, len(s)) for s in samples]
    aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
    assert L.CFHD_PrepareToDecode(dec, 0, 0, T.PIX_YUY2, 1, 0, sbs[0], 512, ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0
    out = np.zeros(w * 2 * h, np.uint8)
    for warm in (True, False):
        t0 = time.perf_counter()
        for i in range(8 if warm else n):
            assert L.CFHD_DecodeSample(dec, sbs[i % 8], len(samples[i % 8]), out.ctypes.data_as(ctypes.c_void_p), w * 2) == 0
        t_dec = time.perf_counter() - t0
    L.CFHD_CloseDecoder(dec)
    res = {"frame": "%dx%d YUY2" % (w, h), "sync_encode_fps": round(n / t_enc, 1), "sync_decode_fps": round(n / t_dec, 1)}
    for workers in (4, 16):
        pool = ctypes.c_void_p()
        assert L.CFHD_CreateEncoderPool(ctypes.byref(pool), workers, 2 * workers, None) == 0
        assert L.CFHD_PrepareEncoderPool(pool, w, h, T.PIX_YUY2, T.ENCODED_YUV422, 0, T.QUALITY_FILMSCAN1) == 0
        assert L.CFHD_StartEncoderPool(pool) == 0
        done = 0
        def collect(wait):
            num = ctypes.c_uint32(); sb = ctypes.c_void_p()
            rc = (L.CFHD_WaitForSample if wait else L.CFHD_TestForSample)(pool, ctypes.byref(num), ctypes.byref(sb))
            if rc == 0: L.CFHD_ReleaseSampleBuffer(pool, sb)
            return rc == 0
        total = 4 * n
        t0 = time.perf_counter()
        for i in range(total):
            while L.CFHD_EncodeAsyncSample(pool, i, frames[i % 8].ctypes.data_as(ctypes.c_void_p), w * 2, None) != 0:
                if collect(True): done += 1
            while collect(False): done += 1
        while done < total:
            if collect(True): done += 1
        res["pool_%d_workers_encode_fps" % workers] = round(total / (time.perf_counter() - t0), 1)
        L.CFHD_ReleaseEncoderPool(pool)
    print(json.dumps(res))

if __name__ == "__main__":
    main()
