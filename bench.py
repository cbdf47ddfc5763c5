#!/usr/bin/env python3
"""bench.py -- CineForm encode+decode round trip on MI355X (BASELINE.json metric: 1080p YUY2 encode+decode fps, % of the HBM roofline,
bitstream-exact).

A step = one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
forward kernels -> entropy coding -> samples -> entropy decoding -> inverse kernels -> frames in HBM, and the host copy of every sample.
`value` is whole-job frames per second (all ranks); `roofline` is the longest kernel of the step against the HBM peak (HIP events around
every launch, on the stream it runs on); `cpu_baseline` is the unmodified reference (oracle/_ref) timed on this box's host cores on a
bounded sample.  After the timed region rank 0 checks what it timed -- eight frames spread over the last timed pass of every batch in flight: the
samples -- ALL of them, by hash -- against the reference encoder run on the same frames (sample 0 also against the golden hash), the decoded frames against the oracle's exact
reconstruction of their own samples (`config.parity`) -- and measures the same codec through the reference's own C ABI from host buffers
(`config.c_abi_fps`, PCIe inclusive, minimum of three runs, never `value`).  Several steps are in flight in the timed region (`--depth`, a HIP-stream frame
queue of batch objects: cfhd_amd_batch_submit / _wait) on the runtime's default hardware queues; the same with GPU_MAX_HW_QUEUES=16 is a side figure.

  python bench.py --gpus 1 --steps 20 --warmup 3 [--workload 1080p|2160p]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse, ctypes, hashlib, json, os, struct, sys, threading, time
os.environ.setdefault("HSA_ENABLE_SDMA", "1")   # D2H of the samples on the SDMA engines: blit-kernel copies stall the kernels they overlap with
# (Not set since round 6: GPU_MAX_HW_QUEUES.  Round 5's line ran with 16 hardware queues, which an application would have had to set itself; the library now shapes its streams
# for the runtime's default of 4, and `value` is what any application linking it gets.  The same run with 16 queues is a side figure: config.with_16_hardware_queues.)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
# BASELINE.json configs: [1] = "1080p" (the metric's configuration; "2160p" is the same path on the 3840x2160 Qbist frames north_star names),
# [2] = "rg48-2160p" (encode), [3] = "b64a-4320p" (encode + decode; the 8-GPU sharding is --gpus N), [4] = "byr4-2160p" (encode) and "1080i".
# fmt / enc / flags are the CFHD_PrepareToEncode arguments; mode 0 = encode + decode, 1 = encode only; bpp = bytes per pixel of the packed frame;
# comps = coefficient planes per frame in units of width x height samples (4:2:2: 1 + 1/2 + 1/2; Bayer: four quarter-size planes)
WORKLOADS = {
    "1080p":      dict(w=1920, h=1080, batch=512, unique=32, fmt="YUY2", enc=0, flags=0, mode=0, bpp=2, comps=2, label="YUY2 4:2:2"),
    "2160p":      dict(w=3840, h=2160, batch=128, unique=8,  fmt="YUY2", enc=0, flags=0, mode=0, bpp=2, comps=2, label="YUY2 4:2:2"),
    "rg48-2160p": dict(w=3840, h=2160, batch=48,  unique=4,  fmt="RG48", enc=1, flags=0, mode=1, bpp=6, comps=3, label="RG48 RGB 4:4:4 12-bit"),
    "b64a-4320p": dict(w=7680, h=4320, batch=8,   unique=2,  fmt="b64a", enc=2, flags=0, mode=0, bpp=8, comps=4, label="b64a RGBA 4:4:4:4"),
    "byr4-2160p": dict(w=3840, h=2160, batch=96,  unique=4,  fmt="BYR4", enc=3, flags=0, mode=1, bpp=2, comps=1, label="BYR4 12-bit Bayer RAW"),
    "1080i":      dict(w=1920, h=1080, batch=512, unique=32, fmt="YUY2", enc=0, flags=1, mode=0, bpp=2, comps=2, label="YUY2 4:2:2 interlaced (frame transform)"),
}


def cpu_baseline(frames, pitch, W, H, seconds_budget=20.0, fmt=None, enc=0, flags=0, decode=True, bpp=2, label="YUY2"):
    """Reference SSE2 path on the host cores: async pool encode (POOL_THREADS = cores) + decode of the same samples."""
    import cfhd_testlib as T
    import numpy as np
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
    fmt = fmt or T.PIX_YUY2
    assert L.CFHD_PrepareEncoderPool(pool, W, H, fmt, enc, flags, T.QUALITY_FILMSCAN1) == 0
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
        samples.append(ctypes.string_at(p, n.value) if len(samples) < nfr else None)
        L.CFHD_ReleaseSampleBuffer(pool, sb)
        return True
    fail = lambda why: {"value": None, "unit": "fps", "cores": cores, "kind": "reference", "sample": why}
    t0 = time.time(); sent = 0; target = 4 * nfr
    qlen = max(2, cores * 3 // 2)
    while True:
        # the job queue holds finished jobs until they are collected: never submit into a full queue (TestCFHD.cpp:903 does the same)
        while sent - len(samples) >= qlen:
            if not collect(True):
                return fail("reference pool failed")
        frms = 24 * 3600 + sent
        tc = ctypes.create_string_buffer(("%02d:%02d:%02d:%02d" % ((frms // 86400) % 24, (frms // 1440) % 60, (frms // 24) % 60, frms % 24)).encode(), 12)
        L.CFHD_MetadataAdd(meta, mtag("TIMC"), 1, 11, ctypes.cast(tc, ctypes.c_void_p), False)
        uf = ctypes.c_uint32(sent)
        L.CFHD_MetadataAdd(meta, mtag("UFRM"), 2, 4, ctypes.cast(ctypes.pointer(uf), ctypes.c_void_p), False)
        rc = L.CFHD_EncodeAsyncSample(pool, sent + 1, frames[sent % nfr].ctypes.data_as(ctypes.c_void_p), pitch, meta)
        if rc != 0:
            return fail("reference pool returned error %d" % rc)
        sent += 1
        while collect(False):
            pass
        if sent >= target and time.time() - t0 > seconds_budget / 2:
            break
    while len(samples) < sent:
        if not collect(True):
            return fail("reference pool failed while draining")
    t_enc = time.time() - t0
    L.CFHD_ReleaseEncoderPool(pool)
    L.CFHD_MetadataClose(meta)
    enc_fps = sent / t_enc
    if not decode:
        return {"value": round(enc_fps, 1), "unit": "fps", "cores": cores, "kind": "reference",
                "sample": "%d frames async-pool encode (%d threads) of %dx%d %s, reference SSE2 build" % (sent, cores, W, H, label)}
    # decode: one reference decoder handle per PROCESS, each configured as Example/TestCFHD.cpp:338-356 does, with TAG_CPU_MAX = 1.  Handles that
    # share a process hold each other up -- the 8-bit output path draws its dither from rand(), whose state sits behind one lock per process -- and the
    # more cores the worse: on the 256-core GPU host 64 one-thread handles in one process decoded 44 fps altogether, 16 handles x 16 threads 22 fps,
    # one handle with all cores 156 fps (8 cores, one process: 8 x 1 thread 375 fps).  Processes share nothing, so this is the reference's best arrangement.
    procs = max(1, min(64, cores))
    dec_seconds = max(2.0, seconds_budget / 2 - 2.0)
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    try:
        with ctx.Pool(procs) as pool_:
            res = pool_.map(_ref_decode_process, [(list(samples[:min(nfr, 4)]), fmt, W, H, bpp, dec_seconds, k) for k in range(procs)])
    except Exception as e:                               # noqa: BLE001 -- the baseline is reported, never allowed to take the bench line down
        return fail("reference decoder processes failed: %r" % (e,))
    if any(r[0] < 0 for r in res):
        return fail("reference decoder returned error %d" % min(r[0] for r in res))
    done = sum(r[0] for r in res)
    dec_fps = sum(r[0] / r[1] for r in res)             # every process timed its own decode loop; they ran side by side for dec_seconds
    rt = 1.0 / (1.0 / enc_fps + 1.0 / dec_fps)
    return {"value": round(rt, 1), "unit": "fps", "cores": cores, "kind": "reference",
            "sample": "%d frames async-pool encode (%.1f fps, %d threads) + %d frames decode by %d processes x 1 handle x 1 decoder thread, side by side for %.0f s (%.1f fps) of %dx%d %s, reference SSE2 build"
                      % (sent, enc_fps, cores, done, procs, dec_seconds, dec_fps, W, H, label)}


def _ref_decode_process(args):
    """One process of cpu_baseline's decode leg: its own copy of the reference library, one decoder handle limited to one thread, decoding the given
    samples in turn for `seconds`.  Returns (frames decoded or a negative error, seconds its loop took)."""
    samples, fmt, W, H, bpp, seconds, k = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import cfhd_testlib as T
    sbuf = [ctypes.create_string_buffer(s, len(s)) for s in samples]
    dec = T.RefDecoder(samples[0], fmt, 1, 1)
    out = np.zeros(W * bpp * H + 64, dtype=np.uint8)
    if dec.decode(sbuf[0], len(samples[0]), out, W * bpp) != 0:          # first call: allocations, tables
        return (-1, 1.0)
    n = 0; t0 = time.time()
    while time.time() - t0 < seconds:
        s = sbuf[(k + n) % len(sbuf)]
        rc = dec.decode(s, len(s), out, W * bpp)
        if rc != 0:
            return (-abs(rc), 1.0)
        n += 1
    el = time.time() - t0
    dec.close()
    return (n, el)


def c_abi_rates(frames, pitch, W, H, seconds=1.5, registered=False, decoders=8, workers=16, all_devices=False):
    """The product through the reference's own C ABI, host buffers in and out (PCIe inclusive): what a C/C++ caller of CFHD_* sees, measured
    by a plain C++ program (tools/cabi_bench.cpp, built against include/cfhd_amd.h and the library only).
    sync: one handle, one thread; pool: CFHD_*EncoderPool with `workers` HIP-stream workers; handles: N decoders on N host threads (calls
    that overlap share launches); round_trip: the pool encoding and N decoder threads decoding its samples at the same time, frames per
    second through both.  registered: the caller page-locked its frame and output buffers once (cfhd_amd_register_host_buffer, an optional
    extension), so the library DMAs between them and HBM without its staging copy."""
    import subprocess, tempfile
    tool = os.environ.get("CFHD_CABI_BENCH") or os.path.join(ROOT, "tools", "_build", "cabi_bench")      # (CFHD_CABI_BENCH: the same program linked against the emulated build, tests/test_frame_shards.py)
    if not os.path.exists(tool):
        return {"error": "tools/_build/cabi_bench is not built (run __graft_entry__.build())"}
    with tempfile.NamedTemporaryFile(suffix=".yuy2") as f:
        for fr in frames:
            f.write(fr.reshape(H, pitch)[:, : W * 2].tobytes())
        f.flush()
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)                # (this process's setting for its batches in flight; the many-thread C ABI case is faster with the runtime's default: profiles/r05_p_*)
        if all_devices:                                  # no pin: the pool's workers and the decoder handles spread over every GPU the process sees
            env.pop("CFHD_AMD_DEVICE", None); env.pop("LOCAL_RANK", None)
        else:                                            # one GPU, whatever the node has (an unpinned process deals its workers to all of them: INTEGRATION.md section 5)
            env.setdefault("CFHD_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0"))
        try:
            out = subprocess.run([tool, str(W), str(H), f.name, str(len(frames)), str(seconds), "1" if registered else "0", str(decoders), str(workers)],
                                 capture_output=True, text=True, timeout=300, env=env)
        except subprocess.TimeoutExpired:
            return {"error": "tools/_build/cabi_bench did not finish within 300 s"}
    if out.returncode != 0:
        return {"error": (out.stderr or out.stdout).strip()[-300:]}
    return json.loads(out.stdout.strip().splitlines()[-1])


DEFAULT_DEPTH = 4      # steps in flight in the timed region (round 5, 16 hardware queues: 55.4 / 58.2 / 59.1 / 57.0 / 57.3 k fps at depth 2 / 3 / 4 / 5 / 6, profiles/r05_o_*; round 6, the runtime's 4 queues: profiles/r06_*)


def normalise_counters(sample):
    """Frame number (tag 69, optional = negated) and the unique frame number tuple count per encoder call: set both to frame 1's."""
    b = bytearray(sample)
    k = bytes(b[:160]).find(struct.pack(">h", -69))
    if k >= 0: b[k + 2:k + 4] = b"\x00\x01"
    u = bytes(b[:1024]).find(b"UFRM")
    if u >= 0: b[u + 8:u + 12] = b"\0\0\0\0"
    return bytes(b)


def parity_check(L, b, frames, pitch, W, H, rank, wl, batch, nuniq, nchk=None, first=0, ref_cache=None):
    """What was timed is what the reference produces -- checked on frames spread over the batch of the last step (eight of them at 1080p, fewer for the larger
    formats: the checker is scalar C): every checked sample (frame / unique-frame counters set back to the first frame's) against the reference encoder run here on
    the same frame (sample 0 of the 1080p YUY2 workload also against the golden hash in tests/golden), and the decoded frame against the exact integer reconstruction
    of its own sample (oracle, test infrastructure, used here as the checker only): inside its dither interval for 8-bit 4:2:2 output, word for word for 16-bit output."""
    import numpy as np
    import cfhd_testlib as T
    fmt = getattr(T, "PIX_" + wl["fmt"].upper())
    nchk = nchk or (8 if W * H <= 1920 * 1080 else (2 if W * H <= 3840 * 2160 else 1))
    checked = sorted({(first + (k * batch) // nchk) % batch for k in range(nchk)})
    out = {"frames_checked": checked}
    bpp = wl["bpp"]
    psnr = []
    # every sample of the pass against the reference encoder's sample of its frame: the batch cycles through `nuniq` pictures, so the reference encodes each once
    # (ref_cache, shared by the batches in flight) and the comparison is a hash of the sample with its counters and clock-dependent metadata set aside
    ref_cache = {} if ref_cache is None else ref_cache
    def ref_digest(u):
        if u not in ref_cache:
            r = T.ref_encode_frames([frames[u]], pitch, W, H, fmt, encoded=wl["enc"], flags=wl["flags"])[0]
            ref_cache[u] = (len(r), hashlib.sha256(T.mask_volatile_metadata(normalise_counters(r))).digest())
        return ref_cache[u]
    for i in range(batch):
        p = ctypes.c_void_p(); sz = ctypes.c_size_t()
        assert L.cfhd_amd_batch_get_sample(b, i, ctypes.byref(p), ctypes.byref(sz)) == 0
        want_len, want = ref_digest(i % nuniq)
        if sz.value != want_len or hashlib.sha256(T.mask_volatile_metadata(normalise_counters(ctypes.string_at(p, sz.value)))).digest() != want:
            raise AssertionError("sample %d of the pass differs from the reference encoder's sample of its frame (%d vs %d bytes)" % (i, sz.value, want_len))
    out["samples_checked"] = batch
    for i in checked:
        frame = frames[i % nuniq]
        p = ctypes.c_void_p(); sz = ctypes.c_size_t()
        assert L.cfhd_amd_batch_get_sample(b, i, ctypes.byref(p), ctypes.byref(sz)) == 0
        sample = ctypes.string_at(p, sz.value)
        if i == 0 and rank == 0 and (W, H) == (1920, 1080) and wl["fmt"] == "YUY2" and not wl["flags"]:
            g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
            digest = hashlib.sha256(T.mask_volatile_metadata(normalise_counters(sample))).hexdigest()
            assert len(sample) == g["qbist_seed10_frame1_size"] and digest == g["qbist_seed10_frame1_masked_sha256"], "sample 0 differs from the reference encoder's golden sample"
            out["sample0_masked_sha256"] = digest
        ref_sample = T.ref_encode_frames([frame], pitch, W, H, fmt, encoded=wl["enc"], flags=wl["flags"])[0]
        ma, mb = T.mask_volatile_metadata(normalise_counters(sample)), T.mask_volatile_metadata(normalise_counters(ref_sample))
        if ma != mb:
            first = next((k for k in range(min(len(ma), len(mb))) if ma[k] != mb[k]), -1)
            raise AssertionError("sample %d differs from the reference encoder's: %d vs %d bytes, first difference at byte %d" % (i, len(ma), len(mb), first))
        if wl["mode"] != 0: continue
        img = np.zeros(H * W * bpp, dtype=np.uint8)
        assert L.cfhd_amd_batch_download_output(b, i, img.ctypes.data_as(ctypes.c_void_p), W * bpp) == 0
        if wl["fmt"] == "YUY2":
            img = img.reshape(H, W * 2)
            plan = T.Plan(W, H, progressive=0 if wl["flags"] & 1 else 1)
            deq = T.oracle_decode_pyramid(sample, plan)
            inverse = T.oracle_inverse_interlaced_yuv422 if wl["flags"] & 1 else T.oracle_inverse_yuv422
            lo = inverse(plan, deq, 0)[:H]; hi = inverse(plan, deq, 1)[:H]
            assert ((img == lo) | (img == hi)).all(), "decoded frame %d leaves the dither interval of the exact reconstruction" % i
            psnr.append(round(float(T.psnr_yuy2(img, frame.reshape(H, pitch)[:, : W * 2])), 2))
        else:
            b64a = wl["fmt"] == "b64a"
            plan = T.Plan(W, H, pixkind=T.PIXKIND[wl["fmt"]], enc=T.ENC["4444" if b64a else "444"])
            exact = T.oracle_inverse_rgb48(plan, T.oracle_decode_pyramid(sample, plan), b64a=b64a)[:H]
            got = np.frombuffer(img.tobytes(), np.uint16).reshape(H, W * bpp // 2)
            assert np.array_equal(got, exact), "decoded frame %d differs from the exact reconstruction" % i
    out["samples_equal_reference_encoder"] = True
    if wl["mode"] == 0:
        if wl["fmt"] == "YUY2": out["decoded_frames_in_dither_interval"] = True; out["psnr_db"] = psnr
        else: out["decoded_frames_equal_exact_reconstruction"] = True
    return out


_API_DECLARED = False


def batch_api():
    import cfhd_testlib as T
    global _API_DECLARED
    L = T.product()
    if not _API_DECLARED:
        L.cfhd_amd_batch_create_ex.restype = ctypes.c_void_p
        L.cfhd_amd_batch_create_ex.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.cfhd_amd_batch_upload.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.cfhd_amd_batch_roundtrip.restype = ctypes.c_longlong
        L.cfhd_amd_batch_roundtrip.argtypes = [ctypes.c_void_p]
        L.cfhd_amd_batch_submit.argtypes = [ctypes.c_void_p]
        L.cfhd_amd_batch_wait.restype = ctypes.c_longlong
        L.cfhd_amd_batch_wait.argtypes = [ctypes.c_void_p]
        L.cfhd_amd_batch_kernel_ms.restype = ctypes.c_float
        L.cfhd_amd_batch_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.cfhd_amd_batch_stage_seconds.restype = ctypes.c_double
        L.cfhd_amd_batch_stage_seconds.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.cfhd_amd_batch_get_sample.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        L.cfhd_amd_batch_download_output.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.cfhd_amd_batch_destroy.argtypes = [ctypes.c_void_p]
        L.cfhd_amd_batch_kernel_name.restype = ctypes.c_char_p
        L.cfhd_amd_batch_kernel_name.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _API_DECLARED = True
    return L


def measure(workload, steps, warmup, batch, unique, threads, rank, world, barrier, reduce_max, depth=1, geometry=None, all_gather=None):
    """One workload through the batched device-resident path: frames generated and uploaded, `warmup` untimed steps, `steps` timed ones between
    barriers, parity check of what was timed on rank 0.  Returns (line fields of this workload, frames, pitch).
    depth: batches in flight (the frame queue, cfhd_amd_batch_submit / _wait): step k + 1 is submitted while step k is still on the GPU; every step is still one
    complete pass over one batch of `batch` frames, and all `steps` of them start and finish inside the timed region."""
    import numpy as np
    import cfhd_testlib as T
    wl = dict(WORKLOADS[workload])
    probing = bool(os.environ.get("CFHD_BENCH_ENCODE_ONLY"))      # timing experiments on the encoder's kernels (tools/gpu_probe.sh): no decode, no parity check
    if probing: wl["mode"] = 1
    if geometry: wl["w"], wl["h"] = geometry            # tests/test_frame_shards.py: the same pass at a size the emulated product finishes in seconds
    W, H = wl["w"], wl["h"]
    L = batch_api()
    nuniq = min(unique or wl["unique"], batch)
    fmt = getattr(T, "PIX_" + wl["fmt"].upper())
    if wl["fmt"] == "BYR4":
        # TestCFHD has no Bayer generator (its Qbist writer would fill the buffer with 8-bit RGB bytes: noise as 16-bit photosites)
        frames = [T.synth_bayer(W, H, 10 + rank + i).reshape(-1).view(np.uint8).copy() for i in range(nuniq)]; pitch = W * 2
        data = "synthetic %dx%d Bayer mosaic, red-green order (tests/cfhd_testlib.py synth_bayer: smooth structure + texture + sensor noise; %d unique frames per rank, cycled through the batch)" % (W, H, nuniq)
    else:
        frames, pitch = T.qbist_frames(10 + rank, nuniq, W, H, fmt, alpha=1 if wl["fmt"] == "b64a" else 0)   # Qbist seed 10 (BASELINE configs), QBIST_UNIQUE frames
        data = "synthetic Qbist %dx%d %s (seed %d, %d unique frames per rank, cycled through the batch)" % (W, H, wl["fmt"], 10, nuniq)
    depth = max(1, min(depth, steps))
    slots = []
    for _ in range(depth):
        b = L.cfhd_amd_batch_create_ex(W, H, fmt, wl["enc"], wl["flags"], T.QUALITY_FILMSCAN1, batch, threads, wl["mode"])
        if not b:
            raise SystemExit("cfhd_amd_batch_create_ex failed: " + T.amd_last_error())
        for i in range(batch):
            assert L.cfhd_amd_batch_upload(b, i, frames[i % nuniq].ctypes.data_as(ctypes.c_void_p), pitch) == 0
        slots.append(b)
    b = slots[0]
    for _ in range(warmup):
        for q in slots: assert L.cfhd_amd_batch_submit(q) == 0
        for q in slots: assert L.cfhd_amd_batch_wait(q) > 0, T.amd_last_error()
    # kernel names as they appear in a rocprofv3 trace of this run (the library picks the register-strip or the LDS-tiled shape by geometry and batch size)
    kname = lambda which: L.cfhd_amd_batch_kernel_name(b, which).decode()
    FWD1, PF2, PF3 = kname(0), kname(1) + "[L2]", kname(2) + "[L3]"
    INV1, PI2, PI3 = (kname(3), kname(4) + "[L2]", kname(5) + "[L3]") if wl["mode"] == 0 else ("", "", "")
    TILES = "k_dec_tiles+k_dec_undiff" if wl["flags"] & 1 else "k_dec_tiles"
    old_dec = os.environ.get("CFHD_AMD_DEC") in ("par", "lane")
    DEC = [("k_dec_bands_par", 13)] if old_dec else [("k_dec_plan", 19), ("k_dec_index", 15), ("k_dec_chain", 16), (TILES, 17)]
    # the level-1 bands are counted on a second stream while levels 2 and 3 are transformed -- from the block lists the forward strip kernel leaves (k_ent_count_blocks)
    # where it runs, else from the dense bands (k_ent_count)
    COUNT1 = ("k_ent_count_blocks" if FWD1.endswith("_blocks") else "k_ent_count") + "[L1 bands, beside the L2 / L3 transforms]"
    KERNELS = [(FWD1, 0), (PF2, 1), (PF3, 2), (COUNT1, 18), ("k_ent_count", 8), ("k_ent_scan", 9), ("k_ent_layout", 10), ("k_ent_emit", 11)]
    if wl["mode"] == 0:
        KERNELS += [("k_dec_parse", 12)] + DEC + [("k_dec_lowpass", 14), (PI3, 5), (PI2, 4), (INV1, 3)]
    kms = {name: 0.0 for name, _ in KERNELS}; stage = [0.0] * 4; total_bytes = 0
    barrier()
    t0 = time.perf_counter()
    def collect(q):
        nonlocal total_bytes
        n = L.cfhd_amd_batch_wait(q)
        assert n > 0, T.amd_last_error()
        total_bytes = n
        for name, which in KERNELS:                    # HIP events recorded around each launch on the stream it runs on
            kms[name] += L.cfhd_amd_batch_kernel_ms(q, which)
        for k in range(4):
            stage[k] += L.cfhd_amd_batch_stage_seconds(q, k)
    for s in range(steps):                             # the frame queue: at most `depth` steps in flight, collected in submission order
        q = slots[s % depth]
        if s >= depth: collect(q)
        assert L.cfhd_amd_batch_submit(q) == 0
    for s in range(steps - depth, steps):
        collect(slots[s % depth])
    b = slots[(steps - 1) % depth]                      # the batch that ran the last step: what the parity check looks at
    barrier()
    mine = time.perf_counter() - t0
    elapsed = reduce_max(mine)
    # every rank's own time over the collective the job runs on (RCCL on the GPUs): rank 0's line says how many ranks took part and what each of them did --
    # the slowest sets `value` (all_gather: None for a single rank; a list of one float per rank, in rank order)
    per_rank = all_gather(mine) if all_gather is not None else [mine]
    # With several steps in flight the kernels of concurrent steps share the GPU, and the HIP events around a launch then time that sharing, not the kernel (as-run
    # times: config.kernel_ms_per_step).  The roofline of a kernel is a statement about the kernel: right behind the timed region the same pass runs ALONE_STEPS more
    # times one at a time on the batch of the last step, with the same events -- those launch times feed `roofline` (and agree with a rocprofv3 trace of --depth 1).
    # What was timed is what gets checked: the parity check reads the samples and pictures the LAST TIMED pass of every batch in flight left behind, before
    # anything runs again on them (a solo pass afterwards could hide corruption between concurrent passes).
    parity = None
    if rank == 0 and not probing and not os.environ.get("CFHD_BENCH_NO_PARITY"):      # (CFHD_BENCH_NO_PARITY: timing probes of builds that produce no valid output, tools/gpu_r05_f.sh; the line then says parity_checked false)
        last = (steps - 1) % depth
        ref_cache = {}
        parity = parity_check(L, slots[last], frames, pitch, W, H, rank, wl, batch, nuniq, ref_cache=ref_cache)
        for k, q in enumerate(slots):
            if k == last: continue
            other = parity_check(L, q, frames, pitch, W, H, rank, wl, batch, nuniq, nchk=2 if W * H <= 1920 * 1080 else 1, first=(k + 1) * batch // (depth + 1), ref_cache=ref_cache)
            parity.setdefault("other_batches_in_flight", []).append(other["frames_checked"])
            parity["samples_checked"] += other["samples_checked"]
    kms_alone = None
    if depth > 1:
        ALONE_STEPS = 8      # (three until round 6: the four longest kernels of a 1080p step are within 4 % of each other, and the longest one names the roofline)
        alone = {name: [] for name, _ in KERNELS}
        for _ in range(ALONE_STEPS):
            assert L.cfhd_amd_batch_roundtrip(b) > 0, T.amd_last_error()
            for name, which in KERNELS: alone[name].append(L.cfhd_amd_batch_kernel_ms(b, which))
        # the median of the passes: one pass in a hundred meets a hiccup of the box (a 10 ms gap inside one bracket: seen once in round 6 on a side workload), and a mean carries it
        kms_alone = {name: float(np.median(v)) for name, v in alone.items()}
    import importlib.util
    spec = importlib.util.spec_from_file_location("frame_shards", os.path.join(ROOT, "cineform-sdk_amd", "host", "frame_shards.py"))
    shards = importlib.util.module_from_spec(spec); spec.loader.exec_module(shards)
    # weak scaling: every rank owns `batch` frames per step (rank r = frames [r*batch, (r+1)*batch) of each step's sequence), no data-path collective
    assert shards.shard_bounds(batch * world, rank, world) == (rank * batch, (rank + 1) * batch)
    fps = shards.whole_job_rate(batch * steps, world, elapsed)

    dx_stats = None
    if os.environ.get("CFHD_AMD_DX_STATS"):               # convergence counters of the chunk-indexed entropy decoder (diagnostics, slows the kernels a little)
        st = (ctypes.c_uint32 * 16)()
        L.cfhd_amd_batch_dx_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        if L.cfhd_amd_batch_dx_stats(b, st) == 0:
            dx_stats = {"rounds": st[0], "chunks_indexed": st[1], "max_rounds": st[2], "chunks_repaired": st[3], "lanes_restarted_per_round": [st[4 + k] for k in range(12)]}
    line = None
    if rank == 0:
        kms_run = {k: v / steps for k, v in kms.items()}
        kms = dict(kms_alone) if kms_alone else dict(kms_run)      # what the roofline is computed from
        sample_bytes = total_bytes / batch
        # algorithmic bytes per frame of every kernel: samples are 8-bit in the packed frame, 16-bit in the pyramid
        Hp = (H + 7) // 8 * 8
        S = W * Hp * wl["comps"]                         # coefficients per frame (4:2:2: luma + both chroma = the packed bytes of the 8-bit frame)
        P = W * Hp * wl["bpp"]                           # bytes of the packed frame
        coded = (S - S // 64) * 2                        # bytes of the entropy-coded bands (everything but the LL3 bands)
        algo = {FWD1: P + 2 * S, PF2: S, PF3: S // 4,                               # 12 441 600 B per 1080p 4:2:2 frame
                "k_ent_count": coded, "k_ent_emit": coded // 2 + sample_bytes}          # emit reads the bit strings k_ent_count leaves (8 bytes per nonzero coefficient, about one in eight), not the pyramid
        if kms.get(COUNT1, 0.0) > 0.0:                   # the count is split: its level-1 part (three quarters of the coefficients) and the rest, each with its own bytes and time
            algo[COUNT1] = S * 3 // 4 * 2; algo["k_ent_count"] = coded - S * 3 // 4 * 2
        else:
            kms.pop(COUNT1, None); kms_run.pop(COUNT1, None)
        if wl["mode"] == 0:
            algo.update({PI3: S // 4, PI2: S, INV1: 2 * S + P})
            if old_dec: algo["k_dec_bands_par"] = sample_bytes + coded
            else: algo.update({"k_dec_index": sample_bytes, TILES: sample_bytes + coded})
        # the dominant kernel = the longest single launch of the step.  (The level-1 count is up to three launches on a second stream beside the transforms of levels 2 and 3: the
        # events around them time that sharing, and each of the launches is shorter than the longest launch of the step -- profiles/: rocprofv3 averages --; it is reported
        # among other_kernels_gbs with the time it takes as run.)
        dom = max((k for k in algo if k != COUNT1 or len(algo) == 1), key=lambda k: kms[k])
        ms = kms[dom] or 1e-9                             # (0 only on the emulated build of the CPU tests, whose events carry no time)
        achieved = algo[dom] * batch / (ms * 1e-3) / 1e9
        traffic = None; traffic_source = None
        try:                                             # HBM bytes per launch from the committed PMC passes of this command (profiles/, same batch size), else null
            pmc_file = "pmc_traffic.json" if workload == "1080p" else "pmc_traffic_%s.json" % workload
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            base = dom.split("[")[0].split("+")[-1]      # the per-level launches of the plane kernels share one trace name
            if pmc.get("frames_per_launch") == batch and pmc.get("workload", "1080p") == workload and base in pmc["kernels"]:
                traffic = pmc["kernels"][base]["hbm_bytes_per_launch"]
                traffic_source = "profiles/%s (rocprofv3 --pmc passes of this command; FETCH_SIZE x2 + WRITE_SIZE per the gfx950 correction)" % pmc_file
        except Exception:
            traffic = None
        handoff = os.environ.get("CFHD_AMD_HANDOFF", "device")
        ent = os.environ.get("CFHD_AMD_ENTROPY", "gpu")
        sum_kernels = sum(v for k, v in kms.items() if k != "k_dec_parse")
        round_trip_bytes = (2 if wl["mode"] == 0 else 1) * (P + 2 * S)     # encode S_in + 2 N_coef, decode 2 N_coef + S_out
        line = {
            "metric": "%s %s %s fps" % (workload.split("-")[-1], wl["fmt"], "encode+decode" if wl["mode"] == 0 else "encode"), "value": round(fps, 1), "unit": "fps", "n_gpus": world, "ranks_seen": len(per_rank), "per_rank_fps": [round(batch * steps / t, 1) for t in per_rank], "steps": steps,
            "warmup": warmup, "ms_per_step": round(1000.0 * elapsed / steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": data,
            "config": {"workload": "%dx%d %s FILMSCAN1 %s, frames resident in HBM" % (W, H, wl["label"], "encode+decode round trip" if wl["mode"] == 0 else "encode"), "frames_per_step_per_gpu": batch,
                       "steps_in_flight": depth,
                       "entropy_stage": ("host, %d threads" % threads) if ent == "host" else "gpu (k_ent_* / k_dec_* kernels)",
                       "sample_handoff": "n/a" if ent == "host" else ("decoder reads the samples in HBM (k_dec_parse); host copy of every sample downloaded inside the step" if handoff != "host" else "samples cross PCIe to the host parser and back"),
                       "sample_bytes_per_frame": int(sample_bytes),
                       "parity_checked": bool(parity), "parity": parity, **({"dx_stats": dx_stats} if dx_stats else {}),
                       "whole_path": {"algorithmic_bytes_per_frame": round_trip_bytes, "gbs": round(round_trip_bytes * batch / (1e6 * elapsed / steps) / 1e3, 1),
                                      "frac_of_hbm_peak": round(round_trip_bytes * batch / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, 4),
                                      "sum_of_kernels_ms": round(sum_kernels, 3)},
                       "stage_ms_per_step": {"submit": round(1000 * stage[0] / steps, 3), "encode_wait+sample_d2h": round(1000 * stage[1] / steps, 3),
                                             "decode_parse+stage": round(1000 * stage[2] / steps, 3), "decode_wait": round(1000 * stage[3] / steps, 3)},
                       "kernel_ms_per_step": {k: round(v, 4) for k, v in kms_run.items()},
                       **({"kernel_ms_one_step_at_a_time": {k: round(v, 4) for k, v in kms.items()}} if kms_alone else {})},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "launch_ms": round(ms, 4),
                         # `achieved` is ALGORITHMIC bytes per launch / launch time (what an ideal implementation must move); what the kernel really moved
                         # through HBM -- less, where the level-1 bands travel as block lists -- is the PMC traffic over the same launch time:
                         "hbm_gbs_from_pmc_traffic": round(traffic / (ms * 1e-3) / 1e9, 1) if traffic else None,
                         "hbm_frac_from_pmc_traffic": round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                         "launch_ms_measured": ("HIP events around the launch, average over the %d timed steps" % steps) if not kms_alone else
                                               ("HIP events around the launch, median of 8 passes run one at a time right behind the timed region (inside it %d steps are in flight and share the GPU: "
                                                "this kernel's event time there is %.4f ms)" % (depth, kms_run.get(dom, 0.0))),
                         "algorithmic_bytes_per_launch": int(algo[dom] * batch),
                         "other_kernels_gbs": {k: round(algo[k] * batch / (kms[k] * 1e-3) / 1e9, 1) for k in algo if kms[k] > 0 and k != dom}},
        }
    for q in slots: L.cfhd_amd_batch_destroy(q)
    return line, frames, pitch


def host_fed(workload, frames, pitch, batch=128, depth=6, steps=36, warmup=6, registered=True):
    """The frame queue fed from host memory (never `value`): every pass copies its `batch` frames from host memory into HBM, encodes, decodes and copies the decoded pictures
    back -- cfhd_amd_batch_submit_host / _wait, `depth` batches in flight so that the copies of one pass run beside the kernels of the others; all of it inside the timed
    region.  The pool semantics of the reference (EncoderSDK/EncoderPool.cpp:239-295, timed by Example/TestCFHD.cpp:1020-1023) for whole batches.  registered: the caller's
    buffers are page-locked (cfhd_amd_register_host_buffer: DMA straight from / into them); else plain memory, staged by the library.  SURVEY.md 8(d) puts the PCIe bound of
    this at ~13-14 k fps for 1080p YUY2 (4.1 MB each way per frame)."""
    import numpy as np
    import cfhd_testlib as T
    wl = WORKLOADS[workload]
    W, H = wl["w"], wl["h"]
    L = batch_api()
    L.cfhd_amd_batch_submit_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    L.cfhd_amd_register_host_buffer.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.cfhd_amd_unregister_host_buffer.argtypes = [ctypes.c_void_p]
    fmt = getattr(T, "PIX_" + wl["fmt"].upper())
    fbytes = pitch * H
    slots = []
    try:
        for _ in range(depth):
            b = L.cfhd_amd_batch_create_ex(W, H, fmt, wl["enc"], wl["flags"], T.QUALITY_FILMSCAN1, batch, 1, wl["mode"])
            if not b: return {"error": "cfhd_amd_batch_create_ex failed: " + T.amd_last_error()}
            src = np.empty(batch * fbytes, dtype=np.uint8); dst = np.zeros(batch * fbytes, dtype=np.uint8)
            for i in range(batch): src[i * fbytes:(i + 1) * fbytes] = frames[i % len(frames)][:fbytes]
            if registered:
                assert L.cfhd_amd_register_host_buffer(src.ctypes.data_as(ctypes.c_void_p), src.size) == 0 and L.cfhd_amd_register_host_buffer(dst.ctypes.data_as(ctypes.c_void_p), dst.size) == 0
            slots.append((b, src, dst))
        def submit(k):
            b, src, dst = slots[k]
            rc = L.cfhd_amd_batch_submit_host(b, src.ctypes.data_as(ctypes.c_void_p), fbytes, pitch, dst.ctypes.data_as(ctypes.c_void_p) if wl["mode"] == 0 else None, fbytes, pitch)
            assert rc == 0, "cfhd_amd_batch_submit_host: %d %s" % (rc, T.amd_last_error())
        def collect(k):
            n = L.cfhd_amd_batch_wait(slots[k][0]); assert n > 0, T.amd_last_error(); return n
        for _ in range(max(1, warmup // depth)):
            for k in range(depth): submit(k)
            for k in range(depth): collect(k)
        t0 = time.perf_counter(); nbytes = 0
        for s_ in range(steps):
            if s_ >= depth: nbytes = collect(s_ % depth)
            submit(s_ % depth)
        for s_ in range(steps - depth, steps): nbytes = collect(s_ % depth)
        el = time.perf_counter() - t0
        # what came back: every sample of the last pass against the reference encoder's (hash), and a decoded picture from the caller's buffer against its own sample
        b, src, dst = slots[(steps - 1) % depth]
        ref = {}
        for i in range(batch):
            p = ctypes.c_void_p(); sz = ctypes.c_size_t()
            assert L.cfhd_amd_batch_get_sample(b, i, ctypes.byref(p), ctypes.byref(sz)) == 0
            u = i % len(frames)
            if u not in ref:
                r = T.ref_encode_frames([frames[u]], pitch, W, H, fmt, encoded=wl["enc"], flags=wl["flags"])[0]
                ref[u] = hashlib.sha256(T.mask_volatile_metadata(normalise_counters(r))).digest()
            assert hashlib.sha256(T.mask_volatile_metadata(normalise_counters(ctypes.string_at(p, sz.value)))).digest() == ref[u], "host-fed sample %d differs from the reference encoder's" % i
        checked = {"samples_checked": batch}
        if wl["mode"] == 0 and wl["fmt"] == "YUY2":
            i = batch - 1
            p = ctypes.c_void_p(); sz = ctypes.c_size_t(); L.cfhd_amd_batch_get_sample(b, i, ctypes.byref(p), ctypes.byref(sz))
            sample = ctypes.string_at(p, sz.value)
            plan = T.Plan(W, H, progressive=0 if wl["flags"] & 1 else 1)
            deq = T.oracle_decode_pyramid(sample, plan)
            inverse = T.oracle_inverse_interlaced_yuv422 if wl["flags"] & 1 else T.oracle_inverse_yuv422
            img = dst[i * fbytes:(i + 1) * fbytes].reshape(H, pitch)[:, : W * 2]
            lo = inverse(plan, deq, 0)[:H]; hi = inverse(plan, deq, 1)[:H]
            assert ((img == lo) | (img == hi)).all(), "host-fed picture leaves the dither interval of the exact reconstruction"
            checked["picture_in_dither_interval"] = True
        per_frame = (fbytes * (2 if wl["mode"] == 0 else 1) + nbytes / batch)
        return {"fps": round(batch * steps / el, 1), "frames_per_pass": batch, "passes_in_flight": depth, "passes_timed": steps, "ms_per_pass": round(1000 * el / steps, 3),
                "host_buffers": "registered by the caller (page-locked): DMA straight from / into them" if registered else "plain memory, staged through the library's pinned buffers by the calling thread",
                "pcie_gbs_both_directions": round(per_frame * batch * steps / el / 1e9, 2), **checked}
    except (AssertionError, Exception) as e:                 # noqa: BLE001 -- a side figure: reported, never allowed to take the bench line down
        return {"error": str(e)[:300]}
    finally:
        for b, src, dst in slots:
            L.cfhd_amd_batch_destroy(b)
            if registered:
                L.cfhd_amd_unregister_host_buffer(src.ctypes.data_as(ctypes.c_void_p)); L.cfhd_amd_unregister_host_buffer(dst.ctypes.data_as(ctypes.c_void_p))


def launcher_command(gpus, argv):
    """`python bench.py --gpus N` on its own (no WORLD_SIZE in the environment): the command that starts the N ranks, one process per GPU, exactly as the
    driver's own command line does; rank 0 of that job prints the line."""
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)      # (a timed region of about a second at 1080p: 120 x 8.6 ms)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="1080p", help="1080p = BASELINE.json configs[1] (the metric's configuration)")
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (0 = the workload's default)")
    ap.add_argument("--unique", type=int, default=0, help="distinct Qbist frames per rank (0 = 32 at 1080p, 8 at 2160p); the batch cycles through them")
    ap.add_argument("--threads", type=int, default=0, help="host entropy threads per rank (0 = cores / ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c-abi", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the short runs of the other BASELINE configs behind the timed region")
    ap.add_argument("--depth", type=int, default=DEFAULT_DEPTH, help="steps in flight (frame queue of batch objects: cfhd_amd_batch_submit / _wait); 1 = one synchronous pass after the other")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        raise SystemExit(subprocess.call(launcher_command(args.gpus, sys.argv[1:])))
    wl = WORKLOADS[args.workload]
    W, H = wl["w"], wl["h"]
    batch = args.batch or wl["batch"]
    headline = wl["fmt"] == "YUY2" and not wl["flags"]     # the metric's own pixel format and transform
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus) and rank == 0:          # the launcher's rank count is what runs (n_gpus in the line says so)
        print("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE): measuring %d" % (args.gpus, world, world), file=sys.stderr)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("CFHD_AMD_DEVICE", str(local_rank))
    import torch
    import cfhd_testlib as T
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libcfhd_amd has no CPU fallback")
    if not T.have_ref():
        raise SystemExit("oracle/_ref/libcfhd_ref.so is missing: it holds the Qbist generator of the benchmark frames and the cpu_baseline (run __graft_entry__.build() where /root/reference exists)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(elapsed):
        if dist is None:
            return elapsed
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather(value):
        if dist is None:
            return [value]
        t = torch.tensor([value], device="cuda", dtype=torch.float64)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(x.item()) for x in out]

    cores = os.cpu_count() or 1
    threads = args.threads or max(1, cores // world)
    line, frames, pitch = measure(args.workload, args.steps, args.warmup, batch, args.unique, threads, rank, world, barrier, reduce_max, depth=args.depth, all_gather=all_gather)
    if rank == 0:
        if world == 1 and args.workload == "1080p" and not args.no_other_workloads:
            # the other BASELINE configs (and north_star's 3840x2160 frames) through the same path, a few steps each, behind the timed region of the
            # headline: their own fps, dominant kernel against the roofline and parity check in the driver-run record (never part of `value`)
            others = {}
            for name in ("2160p", "rg48-2160p", "b64a-4320p", "byr4-2160p", "1080i"):
                try:
                    ol, _, _ = measure(name, 2 * max(args.depth, 1) + 1, 2, WORKLOADS[name]["batch"], 0, threads, 0, 1, barrier, reduce_max, depth=args.depth)      # (the same frame queue as the headline)
                    others[name] = {"metric": ol["metric"], "value": ol["value"], "unit": "fps", "ms_per_step": ol["ms_per_step"], "frames_per_step": ol["config"]["frames_per_step_per_gpu"],
                                    "workload": ol["config"]["workload"], "data": ol["data"], "parity": ol["config"]["parity"], "roofline": {k: ol["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac", "launch_ms")},
                                    "whole_path": ol["config"]["whole_path"], "kernel_ms_per_step": ol["config"]["kernel_ms_per_step"], "steps": ol["steps"], "steps_in_flight": ol["config"].get("steps_in_flight"),
                                    "kernel_ms_one_step_at_a_time": ol["config"].get("kernel_ms_one_step_at_a_time")}
                except (Exception, SystemExit) as e:      # a failed side run is reported, it does not take the headline line with it
                    others[name] = {"error": str(e)[:300]}
            line["config"]["other_workloads"] = others
        if world == 1 and headline and not args.no_other_workloads and "GPU_MAX_HW_QUEUES" not in os.environ:
            # the same timed region in a child process that asks the runtime for 16 hardware queues (what round 5's headline ran with; an application sets it in its environment: INTEGRATION.md section 4)
            import subprocess
            try:
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(min(args.steps, 40)), "--warmup", str(args.warmup), "--depth", str(args.depth), "--batch", str(batch),
                                      "--no-cpu-baseline", "--no-c-abi", "--no-other-workloads"], capture_output=True, text=True, timeout=600, env=dict(os.environ, GPU_MAX_HW_QUEUES="16"))
                side = json.loads(out.stdout.strip().splitlines()[-1])
                line["config"]["with_16_hardware_queues"] = {"value": side["value"], "unit": "fps", "ms_per_step": side["ms_per_step"], "steps": side["steps"], "parity_checked": side["config"]["parity_checked"],
                                                             "environment": "GPU_MAX_HW_QUEUES=16 (read by the ROCm runtime when the process starts; the library then gives a pass three streams instead of two)"}
            except Exception as e:                        # noqa: BLE001 -- a side figure
                line["config"]["with_16_hardware_queues"] = {"error": str(e)[:300]}
        if world == 1 and not args.no_c_abi:
            # the frame queue fed from host memory: upload -> pass -> picture download inside the timed region (never `value`: frames resident in HBM is what the metric times)
            line["host_fed"] = {"what": "cfhd_amd_batch_submit_host / _wait: frames from host memory, samples and decoded pictures back to host memory, all inside the timed region",
                                "registered_buffers": host_fed(args.workload, frames, pitch, registered=True),
                                "plain_buffers": host_fed(args.workload, frames, pitch, depth=4, steps=12, registered=False)}
        if world == 1 and not args.no_c_abi and headline:
            # the host-fed figures swing with the scheduling of ~ 30 host threads: the plain-buffer configuration runs three times, every run is in the line and
            # the MINIMUM of each figure beside them (north_star's 4000 fps round trip is judged on that)
            runs = [c_abi_rates(frames[:8], pitch, W, H, decoders=8, workers=8) for _ in range(3)]
            good = [r for r in runs if "error" not in r]
            plain = dict(good[0]) if good else dict(runs[0])
            if good:
                plain = {k: min(r[k] for r in good) for k in good[0]}
                plain["min_of_runs"] = len(good); plain["runs"] = runs
            line["config"]["c_abi_fps"] = {"frame": "%dx%d YUY2, frames and samples in host memory (PCIe inclusive)" % (W, H),
                                           "plain_buffers": plain,
                                           "plain_buffers_16_threads": c_abi_rates(frames[:8], pitch, W, H, decoders=16, workers=16),
                                           "buffers_registered_by_the_caller_16_threads": c_abi_rates(frames[:8], pitch, W, H, registered=True, decoders=16, workers=16)}
            ngpu = torch.cuda.device_count()
            if ngpu > 1:                                 # host-fed, one process, every GPU of the node: pool workers and decoder handles dealt round robin (strong scaling of the C ABI)
                try:                                     # (first run on a multi-GPU node is the driver's: a failure here is reported, it does not take the line with it)
                    line["config"]["c_abi_fps"]["one_process_all_%d_gpus_plain_buffers" % ngpu] = c_abi_rates(frames[:8], pitch, W, H, decoders=4 * ngpu, workers=4 * ngpu, all_devices=True)
                except (Exception, SystemExit) as e:
                    line["config"]["c_abi_fps"]["one_process_all_%d_gpus_plain_buffers" % ngpu] = {"error": str(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            fmt = getattr(T, "PIX_" + wl["fmt"].upper())
            line["cpu_baseline"] = cpu_baseline(frames[:8], pitch, W, H, fmt=fmt, enc=wl["enc"], flags=wl["flags"], decode=wl["mode"] == 0, bpp=wl["bpp"], label=wl["fmt"])
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
