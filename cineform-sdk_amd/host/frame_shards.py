"""Frame sharding across ranks for the N>1 path (one process per GPU).

Intra frames (gop_length == 1) carry no cross-frame state (every frame is its own sample), so the
sequence is cut into contiguous shards, one per rank, and nothing but the start/stop barrier and the max-over-ranks
time crosses ranks -- no data-path collective.  bench.py and tests/test_frame_shards.py both use these helpers.
"""


def shard_bounds(total_frames, rank, world, gop_length=1):
    """[first, last) of the frames rank `rank` of `world` processes: contiguous, sizes differ by at most one *group*.
    gop_length 2 (CFHD_ENCODING_FLAGS_YUV_2FRAME_GOP): the two frames of a group share one sample, so shards are cut between groups; a
    trailing odd frame stays with the last group's rank (the reference's encoder holds it until the next frame arrives, encoder.c:2927)."""
    if not (0 <= rank < world) or total_frames < 0 or gop_length < 1:
        raise ValueError("bad shard request")
    groups = (total_frames + gop_length - 1) // gop_length
    base, extra = divmod(groups, world)
    first = rank * base + min(rank, extra)
    last = first + base + (1 if rank < extra else 0)
    return min(first * gop_length, total_frames), min(last * gop_length, total_frames)


def frame_numbers(total_frames, rank, world, first_number=1, gop_length=1):
    """Frame numbers (the sample header's FRAME_NUMBER, 1-based like the reference's encoder) of this rank's shard."""
    first, last = shard_bounds(total_frames, rank, world, gop_length)
    return list(range(first_number + first, first_number + last))


def whole_job_rate(frames_per_rank, world, slowest_rank_seconds):
    """bench.py's `value`: every rank's frames over the slowest rank's time."""
    return frames_per_rank * world / slowest_rank_seconds
