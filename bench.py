#!/usr/bin/env python3
"""bench.py -- 1080p YUY2 4:2:2 encode+decode round trip on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
forward kernels -> entropy coding -> samples -> entropy decoding -> inverse kernels -> frames in HBM.
`value` is whole-job frames per second (all ranks), `roofline` is the longest kernel of the step against the
HBM peak (HIP events around every launch), `cpu_baseline` is the unmodified reference (oracle/_ref) timed on this box's host cores on a bounded sample.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse, ctypes, json, os, sys, time
os.environ.setdefault("HSA_ENABLE_SDMA", "1")   # D2H of the samples on the SDMA engines: blit-kernel copies stall the kernels they overlap with

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
W, H = 1920, 1080


def cpu_baseline(frames, pitch, seconds_budget=20.0):
    """Reference SSE2 path on the host cores: async pool encode (POOL_THREADS = cores) + decode of the same samples."""
    import cfhd_testlib as T
    if not T.have_ref():
        return None
    L = T.ref()
    cores = os.cpu_count() or 1
    nfr = len(frames)
    # encode: CFHD_CreateEncoderPool(threads = cores, queue = 1.5 * cores), as Example/TestCFHD.cpp:830-1026
    pool = ctypes.c_void_p()
    assert L.CFHD_CreateEncoderPool(ctypes.byref(pool), cores, max(2, cores * 3 // 2), None) == 0
    # TestCFHD attaches a metadata handle and refreshes timecode / unique frame number per frame (Example/TestCFHD.cpp:826-930);
    # the reference's pool workers never complete a job that was submitted with a NULL metadata handle.
    meta = ctypes.c_void_p()
    assert L.CFHD_MetadataOpen(ctypes.byref(meta)) == 0
    L.CFHD_AttachEncoderPoolMetadata(pool, meta)
    assert L.CFHD_PrepareEncoderPool(pool, W, H, T.PIX_YUY2, T.ENCODED_YUV422, 0, T.QUALITY_FILMSCAN1) == 0
    L.CFHD_AttachEncoderPoolMetadata(pool, meta)
    assert L.CFHD_StartEncoderPool(pool) == 0
    mtag = lambda t: ord(t[0]) | (ord(t[1]) << 8) | (ord(t[2]) << 16) | (ord(t[3]) << 24)
    samples = []
    def collect(wait):
        num = ctypes.c_uint32(); sb = ctypes.c_void_p()
        rc = (L.CFHD_WaitForSample if wait else L.CFHD_TestForSample)(pool, ctypes.byref(num), ctypes.byref(sb))
        if rc != 0:
            return False
        p = ctypes.c_void_p(); n = ctypes.c_size_t()
        L.CFHD_GetEncodedSample(sb, ctypes.byref(p), ctypes.byref(n))
        if len(samples) < nfr:
            samples.append(ctypes.string_at(p, n.value))
        else:
            samples.append(None)
        L.CFHD_ReleaseSampleBuffer(pool, sb)
        return True
    t0 = time.time(); sent = 0; target = 4 * nfr
    qlen = max(2, cores * 3 // 2)
    while True:
        # the job queue holds finished jobs until they are collected: never submit into a full queue (TestCFHD.cpp:903 does the same)
        while sent - len(samples) >= qlen:
            if not collect(True):
                return {"value": None, "unit": "fps", "cores": cores, "kind": "reference", "sample": "reference pool failed"}
        frms = 24 * 3600 + sent
        tc = ctypes.create_string_buffer(("%02d:%02d:%02d:%02d" % ((frms // 86400) % 24, (frms // 1440) % 60, (frms // 24) % 60, frms % 24)).encode(), 12)
        L.CFHD_MetadataAdd(meta, mtag("TIMC"), 1, 11, ctypes.cast(tc, ctypes.c_void_p), False)
        uf = ctypes.c_uint32(sent)
        L.CFHD_MetadataAdd(meta, mtag("UFRM"), 2, 4, ctypes.cast(ctypes.pointer(uf), ctypes.c_void_p), False)
        rc = L.CFHD_EncodeAsyncSample(pool, sent + 1, frames[sent % nfr].ctypes.data_as(ctypes.c_void_p), pitch, meta)
        if rc != 0:
            return {"value": None, "unit": "fps", "cores": cores, "kind": "reference", "sample": "reference pool returned error %d" % rc}
        sent += 1
        while collect(False):
            pass
        if sent >= target and time.time() - t0 > seconds_budget / 2:
            break
    while len(samples) < sent:
        if not collect(True):
            return {"value": None, "unit": "fps", "cores": cores, "kind": "reference", "sample": "reference pool failed while draining"}
    t_enc = time.time() - t0
    L.CFHD_ReleaseEncoderPool(pool)
    L.CFHD_MetadataClose(meta)
    enc_fps = sent / t_enc
    # decode: one decoder (it spawns its own worker threads, TAG_CPU_MAX unset = all cores)
    dec = ctypes.c_void_p(); L.CFHD_OpenDecoder(ctypes.byref(dec), None)
    aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
    sbuf = [ctypes.create_string_buffer(s, len(s)) for s in samples[:nfr]]
    L.CFHD_PrepareToDecode(dec, 0, 0, T.PIX_YUY2, 1, 0, sbuf[0], 512, ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af))
    import numpy as np
    out = np.zeros(W * 2 * H, dtype=np.uint8)
    t0 = time.time(); done = 0
    while True:
        s = sbuf[done % nfr]
        rc = L.CFHD_DecodeSample(dec, s, len(s), out.ctypes.data_as(ctypes.c_void_p), W * 2)
        if rc != 0:
            return {"value": None, "unit": "fps", "cores": cores, "kind": "reference", "sample": "reference decoder returned error %d" % rc}
        done += 1
        if done >= 2 * nfr and time.time() - t0 > seconds_budget / 2:
            break
    t_dec = time.time() - t0
    L.CFHD_CloseDecoder(dec)
    dec_fps = done / t_dec
    rt = 1.0 / (1.0 / enc_fps + 1.0 / dec_fps)
    return {"value": round(rt, 1), "unit": "fps", "cores": cores, "kind": "reference",
            "sample": "%d frames async-pool encode (%.1f fps, %d threads) + %d frames decode (%.1f fps) of 1920x1080 YUY2, reference SSE2 build"
                      % (sent, enc_fps, cores, done, dec_fps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="frames per step per GPU")
    ap.add_argument("--threads", type=int, default=0, help="host entropy threads per rank (0 = cores / ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("CFHD_AMD_DEVICE", str(local_rank))
    import numpy as np
    import torch
    import cfhd_testlib as T
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libcfhd_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    L = T.product()
    L.cfhd_amd_batch_create.restype = ctypes.c_void_p
    L.cfhd_amd_batch_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.cfhd_amd_batch_upload.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.cfhd_amd_batch_roundtrip.restype = ctypes.c_longlong
    L.cfhd_amd_batch_roundtrip.argtypes = [ctypes.c_void_p]
    L.cfhd_amd_batch_kernel_ms.restype = ctypes.c_float
    L.cfhd_amd_batch_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.cfhd_amd_batch_stage_seconds.restype = ctypes.c_double
    L.cfhd_amd_batch_stage_seconds.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.cfhd_amd_batch_destroy.argtypes = [ctypes.c_void_p]

    cores = os.cpu_count() or 1
    threads = args.threads or max(1, cores // world)
    nuniq = 8
    if T.have_ref():
        frames, pitch = T.qbist_frames(10 + rank, nuniq)            # Qbist seed 10 (BASELINE configs), QBIST_UNIQUE frames
        data = "synthetic Qbist 1920x1080 YUY2 (seed %d, %d unique frames per rank)" % (10, nuniq)
    else:
        frames = [T.synth_yuy2(W, H, 100 * rank + i)[0] for i in range(nuniq)]; pitch = W * 2
        data = "synthetic gradients+noise 1920x1080 YUY2"
    b = L.cfhd_amd_batch_create(W, H, T.PIX_YUY2, T.QUALITY_FILMSCAN1, args.batch, threads)
    if not b:
        raise SystemExit("cfhd_amd_batch_create failed: " + T.amd_last_error())
    for i in range(args.batch):
        assert L.cfhd_amd_batch_upload(b, i, frames[i % nuniq].ctypes.data_as(ctypes.c_void_p), pitch) == 0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        assert L.cfhd_amd_batch_roundtrip(b) > 0, T.amd_last_error()
    barrier()
    t0 = time.perf_counter()
    # kernel names as they appear in a rocprofv3 trace of this run: the register-strip kernels serve 1920x1080 unless an A/B switch asks for the tiled ones
    FWD1 = "k_fwd_yuv422" if os.environ.get("CFHD_AMD_FORWARD") == "tile" else "k_fwd_yuv422_strip"
    INV1 = "k_inv_yuv422" if os.environ.get("CFHD_AMD_INVERSE") == "tile" else "k_inv_yuv422_strip"
    PF, PI = ("k_fwd_plane", "k_inv_plane") if os.environ.get("CFHD_AMD_PLANES") == "tile" else ("k_fwd_plane_strip", "k_inv_plane_strip")
    KERNELS = [(FWD1, 0), (PF + "[L2]", 1), (PF + "[L3]", 2), ("k_ent_count", 8), ("k_ent_scan", 9), ("k_ent_layout", 10),
               ("k_ent_emit", 11), ("k_dec_parse", 12), ("k_dec_bands_par", 13), ("k_dec_lowpass", 14), (PI + "[L3]", 5), (PI + "[L2]", 4),
               (INV1, 3)]
    kms = {name: 0.0 for name, _ in KERNELS}; stage = [0.0] * 4; total_bytes = 0
    for _ in range(args.steps):
        n = L.cfhd_amd_batch_roundtrip(b)
        assert n > 0, T.amd_last_error()
        total_bytes = n
        for name, which in KERNELS:                    # HIP events recorded around each launch on the stream it runs on
            kms[name] += L.cfhd_amd_batch_kernel_ms(b, which)
        for k in range(4):
            stage[k] += L.cfhd_amd_batch_stage_seconds(b, k)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    import importlib.util
    spec = importlib.util.spec_from_file_location("frame_shards", os.path.join(ROOT, "cineform-sdk_amd", "host", "frame_shards.py"))
    shards = importlib.util.module_from_spec(spec); spec.loader.exec_module(shards)
    # weak scaling: every rank owns `batch` frames per step (rank r = frames [r*batch, (r+1)*batch) of each step's sequence), no data-path collective
    assert shards.shard_bounds(args.batch * world, rank, world) == (rank * args.batch, (rank + 1) * args.batch)
    fps = shards.whole_job_rate(args.batch * args.steps, world, elapsed)

    if rank == 0:
        kms = {k: v / args.steps for k, v in kms.items()}
        sample_bytes = total_bytes / args.batch
        # algorithmic bytes per frame of every kernel: samples are 8-bit in the packed frame, 16-bit in the pyramid
        S = W * ((H + 7) // 8 * 8) * 2                   # samples per 4:2:2 frame (luma + both chroma) = packed bytes
        coded = (S - S // 64) * 2                        # bytes of the 27 entropy-coded bands (everything but the three LL3 bands)
        algo = {FWD1: S + 2 * S, PF + "[L2]": S, PF + "[L3]": S // 4,                               # 12 441 600 B per 1080p frame
                "k_ent_count": coded, "k_ent_emit": coded + sample_bytes, "k_dec_bands_par": sample_bytes + coded,
                PI + "[L3]": S // 4, PI + "[L2]": S, INV1: 2 * S + S}
        dom = max(algo, key=lambda k: kms[k])            # the dominant kernel = the longest launch of the step
        ms = kms[dom]
        achieved = algo[dom] * args.batch / (ms * 1e-3) / 1e9
        traffic = None
        try:                                             # HBM bytes per launch from the committed PMC passes (profiles/, same batch size), else null
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            base = dom.split("[")[0]                      # the per-level launches of the plane kernels share one trace name
            if pmc.get("frames_per_launch") == args.batch and base in pmc["kernels"]:
                traffic = pmc["kernels"][base]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        handoff = os.environ.get("CFHD_AMD_HANDOFF", "device")
        ent = os.environ.get("CFHD_AMD_ENTROPY", "gpu")
        line = {
            "metric": "1080p YUY2 encode+decode fps", "value": round(fps, 1), "unit": "fps", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": data,
            "config": {"workload": "1920x1080 YUY2 4:2:2 FILMSCAN1 encode+decode round trip, frames resident in HBM", "frames_per_step_per_gpu": args.batch,
                       "entropy_stage": ("host, %d threads" % threads) if ent == "host" else "gpu (k_ent_* / k_dec_* kernels)",
                       "sample_handoff": "n/a" if ent == "host" else ("decoder reads the samples in HBM (k_dec_parse); host copy of every sample downloaded inside the step" if handoff != "host" else "samples cross PCIe to the host parser and back"),
                       "sample_bytes_per_frame": int(sample_bytes),
                       "stage_ms_per_step": {"submit": round(1000 * stage[0] / args.steps, 3), "encode_wait+sample_d2h": round(1000 * stage[1] / args.steps, 3),
                                             "decode_parse+stage": round(1000 * stage[2] / args.steps, 3), "decode_wait": round(1000 * stage[3] / args.steps, 3)},
                       "kernel_ms_per_step": {k: round(v, 4) for k, v in kms.items()}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "launch_ms": round(ms, 4),
                         "algorithmic_bytes_per_launch": int(algo[dom] * args.batch),
                         "other_kernels_gbs": {k: round(algo[k] * args.batch / (kms[k] * 1e-3) / 1e9, 1) for k in algo if kms[k] > 0 and k != dom}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(frames, pitch)
        print(json.dumps(line), flush=True)
    L.cfhd_amd_batch_destroy(b)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
