"""N>1 path on CPU: world_size-2 `gloo` run of the frame sharding bench.py uses (one process per GPU, no data-path collective).

Each rank encodes its shard with the CPU twin of the GPU path (oracle transform + the product's host sample writer, which
tests/test_host_bitstream.py pins byte for byte against the reference); the gathered samples must equal the unsharded run."""
import hashlib, importlib.util, os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _shards():
    spec = importlib.util.spec_from_file_location("frame_shards", os.path.join(ROOT, "cineform-sdk_amd", "host", "frame_shards.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


W, H, TOTAL = 96, 64, 7


def _encode(numbers):
    import cfhd_testlib as T
    plan = T.Plan(W, H)
    out = {}
    for n in numbers:
        frame, pitch = T.synth_yuy2(W, H, 1000 + n)
        coeffs = T.oracle_forward_yuv422(plan, frame, pitch)
        out[n] = hashlib.sha256(T.product_write_sample_host(plan, coeffs, n, meta_global=b"GUID\x10\x00\x00G" + bytes(16))).hexdigest()
    return out


def _encode_groups(numbers):
    """Two-frame groups (CFHD_ENCODING_FLAGS_YUV_2FRAME_GOP): the group that starts at frame number n (odd, 1-based) -> its sample's hash; oracle
    group transform + the product's host writer, which tests/test_gop.py pins byte for byte against the reference."""
    import cfhd_testlib as T
    gp = T.GopPlan(W, H)
    out = {}
    for n in numbers:
        if n % 2 == 0: continue                      # the second frame of a group travels in its group's sample
        f0, pitch = T.synth_yuy2(W, H, 1000 + n); f1, _ = T.synth_yuy2(W, H, 1001 + n)
        co = T.oracle_forward_gop(gp, f0, f1, pitch)
        out[n] = hashlib.sha256(T.product_write_gop_host(gp, 0, co, n, meta_global=b"GUID\x10\x00\x00G" + bytes(16))).hexdigest()
    return out


def _worker(rank, world, port, q, gop=1):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = _shards()
    mine = _encode(S.frame_numbers(TOTAL, rank, world)) if gop == 1 else _encode_groups(S.frame_numbers(TOTAL + 1, rank, world, gop_length=2))
    dist.barrier()                                   # the only cross-rank traffic of the data path: start/stop barriers ...
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)           # ... and, in this test only, the checksums for the comparison
    import torch
    t = torch.tensor([0.25 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)         # bench.py's max-over-ranks time
    if rank == 0:
        q.put((gathered, float(t.item())))
    dist.destroy_process_group()


def test_shard_bounds_cover_the_sequence_exactly():
    S = _shards()
    for total in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = S.shard_bounds(total, r, world)
                assert 0 <= a <= b <= total and b - a in (total // world, total // world + 1)
                seen += list(range(a, b))
            assert seen == list(range(total))
    with pytest.raises(ValueError):
        S.shard_bounds(4, 2, 2)
    assert S.whole_job_rate(256, 8, 0.5) == 4096.0
    # two-frame groups are never split over ranks; a trailing odd frame stays with the last group's rank
    for total in (0, 1, 2, 7, 8, 1001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = S.shard_bounds(total, r, world, gop_length=2)
                assert (a % 2 == 0 or a == total) and (b % 2 == 0 or b == total)
                seen += list(range(a, b))
            assert seen == list(range(total))
    assert S.frame_numbers(8, 1, 2, gop_length=2) == [5, 6, 7, 8]


def test_two_ranks_gloo_sharded_encode_equals_unsharded():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    gathered, slowest = q.get(timeout=180)
    for p in procs:
        p.join(60); assert p.exitcode == 0
    merged = {}
    for part in gathered:
        assert not (set(part) & set(merged)), "shards overlap"
        merged.update(part)
    assert merged == _encode(range(1, TOTAL + 1))
    assert slowest == 0.5


def test_two_ranks_gloo_sharded_group_encode_equals_unsharded():
    """The same for two-frame groups (8 frames = 4 groups, two per rank): group samples carry no state from the groups before them (fixed-table
    qualities), so cutting the sequence between groups changes no byte."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 2)) for r in range(2)]
    for p in procs: p.start()
    gathered, slowest = q.get(timeout=180)
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert [sorted(part) for part in gathered] == [[1, 3], [5, 7]]
    merged = {}
    for part in gathered: merged.update(part)
    assert merged == _encode_groups(range(1, TOTAL + 2))


def _product_samples(frame_numbers):
    """The frames with these (1-based) numbers through the PRODUCT's batched path -- the emulated build of the whole library (tests/cfhd_testlib.py product_emulated:
    C ABI, batch front end, job tables, unmodified kernels on the CPU) -- as bench.py drives it per rank: one batch of the rank's shard, one pass.  Returns
    {frame number: hash of its sample with the per-encoder counters (frame number, unique frame number) and the volatile metadata zeroed}."""
    import ctypes, struct
    import cfhd_testlib as T
    numbers = list(frame_numbers)
    out = {}
    with T.emulated_product() as L:
        L.cfhd_amd_batch_create.restype = ctypes.c_void_p
        L.cfhd_amd_batch_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.cfhd_amd_batch_upload.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.cfhd_amd_batch_roundtrip.restype = ctypes.c_longlong; L.cfhd_amd_batch_roundtrip.argtypes = [ctypes.c_void_p]
        L.cfhd_amd_batch_get_sample.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        L.cfhd_amd_batch_destroy.argtypes = [ctypes.c_void_p]
        L.cfhd_amd_device_count.restype = ctypes.c_int
        b = L.cfhd_amd_batch_create(W, H, T.PIX_YUY2, T.QUALITY_FILMSCAN1, len(numbers), 1)
        assert b, T.amd_last_error()
        for i, n in enumerate(numbers):
            frame, pitch = T.synth_yuy2(W, H, 1000 + n)
            assert L.cfhd_amd_batch_upload(b, i, frame.ctypes.data_as(ctypes.c_void_p), pitch) == 0
        assert L.cfhd_amd_batch_roundtrip(b) > 0, T.amd_last_error()
        for i, n in enumerate(numbers):
            p = ctypes.c_void_p(); sz = ctypes.c_size_t()
            assert L.cfhd_amd_batch_get_sample(b, i, ctypes.byref(p), ctypes.byref(sz)) == 0
            s = bytearray(T.mask_volatile_metadata(ctypes.string_at(p, sz.value)))
            k = bytes(s[:160]).find(struct.pack(">h", -69)); s[k + 2:k + 4] = b"\0\0"
            u = bytes(s[:1024]).find(b"UFRM"); s[u + 8:u + 12] = b"\0\0\0\0"
            out[n] = hashlib.sha256(bytes(s)).hexdigest()
        L.cfhd_amd_batch_destroy(b)
        ndev = L.cfhd_amd_device_count()
    return out, ndev


def _worker_product(rank, world, port, q):
    """One process per GPU as bench.py launches them: LOCAL_RANK pins the process to its own (emulated) device, the rank runs its shard through the product."""
    os.environ["LOCAL_RANK"] = str(rank); os.environ["HIPEMU_DEVICES"] = str(world)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = _shards()
    mine, ndev = _product_samples(S.frame_numbers(TOTAL, rank, world))
    dist.barrier()
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, ndev))
    if rank == 0: q.put(gathered)
    dist.destroy_process_group()


def test_two_ranks_gloo_product_batches_on_their_own_devices_equal_the_unsharded_run():
    """The product's own bookkeeping under the launch bench.py gets for N > 1: two processes, each pinned by LOCAL_RANK to its own emulated GPU (separate, protected
    heaps: tests/hipemu/hip/hip_runtime.h), each encoding + decoding its contiguous shard with cfhd_amd_batch_roundtrip; the samples, gathered over gloo, equal those
    of one process running the whole sequence."""
    import torch.multiprocessing as mp
    import cfhd_testlib as T
    if not T.have_ref(): pytest.skip("reference .so not built (the emulated product build is made beside it)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    T.product_emulated()                              # (built once here, not by two workers at the same time)
    procs = [ctx.Process(target=_worker_product, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    gathered = q.get(timeout=300)
    for p in procs:
        p.join(120); assert p.exitcode == 0
    merged = {}
    for part, ndev in gathered:
        assert ndev == 2
        assert not (set(part) & set(merged)), "shards overlap"
        merged.update(part)
    whole, _ = _product_samples(range(1, TOTAL + 1))
    assert merged == whole


def _worker_bench(rank, world, port, q):
    """bench.py's own measure() -- frame queue, max-over-ranks time, whole-job rate, parity check of what was timed -- as one rank of an N-rank job: the launch
    bench.py --gpus N gets (RANK / LOCAL_RANK / WORLD_SIZE in the environment), gloo in place of RCCL, the emulated product in place of the GPU."""
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "HIPEMU_DEVICES": str(world), "CFHD_AMD_DEVICE": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    import cfhd_testlib as T
    def reduce_max(elapsed):
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    def all_gather(value):
        out = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(out, torch.tensor([value], dtype=torch.float64))
        return [float(x.item()) for x in out]
    workload, geometry, batch = os.environ.get("CFHD_TEST_WORKLOAD", "1080p"), (192, 96), 3
    if workload == "b64a-4320p": geometry, batch = (256, 128), 2
    with T.emulated_product():
        line, frames, pitch = bench.measure(workload, 3, 1, batch, 2, 1, rank, world, dist.barrier, reduce_max, depth=2, geometry=geometry, all_gather=all_gather)
    gathered = [None] * world
    dist.all_gather_object(gathered, line)
    if rank == 0: q.put(gathered)
    dist.destroy_process_group()


def test_bench_measure_runs_as_two_ranks_and_reports_the_whole_job():
    """`python bench.py --gpus 2`: the launcher command starts two ranks of bench.py itself, and measure() run as those two ranks (gloo, two emulated GPUs)
    prints ONE line on rank 0 with n_gpus = 2, the frames of both ranks in `value`, and the parity check of what rank 0 timed."""
    import torch.multiprocessing as mp
    import cfhd_testlib as T
    if not T.have_ref(): pytest.skip("reference .so not built (Qbist frames, the emulated product build)")
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(2, ["--gpus", "2", "--steps", "5"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "2", "--steps", "5"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    T.product_emulated()
    procs = [ctx.Process(target=_worker_bench, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    lines = q.get(timeout=600)
    for p in procs:
        p.join(120); assert p.exitcode == 0
    line = lines[0]
    assert lines[1] is None                            # one line per job
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 3
    assert line["ranks_seen"] == 2 and len(line["per_rank_fps"]) == 2 and abs(min(line["per_rank_fps"]) * 2 - line["value"]) <= 0.25      # the slowest rank sets the whole-job figure (rates are rounded to 0.1)
    assert abs(line["value"] - 2 * 3 * 3 / (line["ms_per_step"] * 3e-3)) / line["value"] < 1e-2       # frames of both ranks / the slowest rank's time
    par = line["config"]["parity"]
    assert par["samples_equal_reference_encoder"] and par["decoded_frames_in_dither_interval"] and len(par["other_batches_in_flight"]) == 1


def test_bench_measure_runs_config_c_as_eight_ranks():
    """BASELINE.json configs[3] as the driver's 8-GPU run will start it -- `bench.py --gpus 8 --workload b64a-4320p` -- on eight emulated devices: eight ranks (gloo in place of
    RCCL), each pinned to its own device, each with its own shard of b64a frames through encode + decode; rank 0's line counts all eight (`ranks_seen`), carries every rank's
    rate and the parity check of what rank 0 timed."""
    import torch.multiprocessing as mp
    import cfhd_testlib as T
    if not T.have_ref(): pytest.skip("reference .so not built (Qbist frames, the emulated product build)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + (os.getpid() % 2000)
    T.product_emulated()
    os.environ["CFHD_TEST_WORKLOAD"] = "b64a-4320p"
    try:
        procs = [ctx.Process(target=_worker_bench, args=(r, 8, port, q)) for r in range(8)]
        for p in procs: p.start()
        lines = q.get(timeout=900)
        for p in procs:
            p.join(180); assert p.exitcode == 0
    finally:
        del os.environ["CFHD_TEST_WORKLOAD"]
    line = lines[0]
    assert all(l is None for l in lines[1:])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and len(line["per_rank_fps"]) == 8 and line["scaling"] == "weak"
    assert "b64a" in line["metric"] and line["config"]["frames_per_step_per_gpu"] == 2
    assert abs(line["value"] - 8 * 2 * 3 / (line["ms_per_step"] * 3e-3)) / line["value"] < 1e-2
    par = line["config"]["parity"]
    assert par["samples_equal_reference_encoder"] and par["decoded_frames_equal_exact_reconstruction"] and par["samples_checked"] == 4      # both batches in flight, two frames each


def test_c_abi_leg_of_the_bench_spreads_over_two_devices():
    """bench.py's `c_abi.one_process_all_N_gpus_plain_buffers` leg (one process, pool workers and decoder handles dealt to every GPU it sees: INTEGRATION.md section 5) on two
    emulated devices: tools/cabi_bench.cpp linked against the emulated build of the library, unpinned, must come back with all its figures."""
    import subprocess
    import cfhd_testlib as T
    if not T.have_ref(): pytest.skip("reference .so not built (the emulated product build is made beside it)")
    T.product_emulated()
    tool = os.path.join(ROOT, "tests", "_build", "cabi_bench_hipemu")
    src = os.path.join(ROOT, "tools", "cabi_bench.cpp")
    if not os.path.exists(tool) or os.path.getmtime(tool) < max(os.path.getmtime(src), os.path.getmtime(T.EMU_PRODUCT_SO)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", tool, T.EMU_PRODUCT_SO, "-Wl,-rpath," + os.path.dirname(T.EMU_PRODUCT_SO), "-lpthread"])
    sys.path.insert(0, ROOT)
    import bench
    w, h = 192, 96
    frames = [T.synth_yuy2(w, h, s)[0] for s in (1, 2, 3, 4)]
    os.environ["CFHD_CABI_BENCH"] = tool; os.environ["HIPEMU_DEVICES"] = "2"
    try:
        r = bench.c_abi_rates(frames, w * 2, w, h, seconds=0.2, decoders=4, workers=4, all_devices=True)
    finally:
        del os.environ["CFHD_CABI_BENCH"]; del os.environ["HIPEMU_DEVICES"]
    assert "error" not in r, r
    assert r["sync_encode_fps"] > 0 and r["sync_decode_fps"] > 0 and r["decode_fps_4_handles"] > 0 and r["pool_encode_fps_4_workers"] > 0 and r["round_trip_fps_pool4_plus_4_decoders"] >= 0, r      # (the round trip's window opens with warm decoders: it can stay empty at the emulator's pace)
