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
