"""Frame sharding across ranks for the N>1 path (one process per GPU).

Intra frames (gop_length == 1) carry no cross-frame state (every frame is its own sample), so the
sequence is cut into contiguous shards, one per rank, and nothing but the start/stop barrier and the max-over-ranks
time crosses ranks -- no data-path collective.  bench.py and tests/test_frame_shards.py both use these helpers.
"""


def shard_bounds(total_frames, rank, world):
    """[first, last) of the frames rank `rank` of `world` processes: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world) or total_frames < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total_frames, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def frame_numbers(total_frames, rank, world, first_number=1):
    """Frame numbers (the sample header's FRAME_NUMBER, 1-based like the reference's encoder) of this rank's shard."""
    first, last = shard_bounds(total_frames, rank, world)
    return list(range(first_number + first, first_number + last))


def whole_job_rate(frames_per_rank, world, slowest_rank_seconds):
    """bench.py's `value`: every rank's frames over the slowest rank's time."""
    return frames_per_rank * world / slowest_rank_seconds
