import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libcfhd_ref.so (the unmodified reference built by oracle/Makefile)")


# Order of the GPU suite: the driver runs `pytest -x`, so what BASELINE.json's north_star and configs name runs first -- the kernels bench.py
# times (forced strips, bench-size batches), config A through the C ABI, configs B / C / D, the reference's own harness -- and the format sweep
#  after it.  Tests not listed keep their file order behind the listed ones.
GPU_FIRST = [
    "test_yuv422_strip_kernels_equal_reference",
    "test_batched_round_trip_at_bench_sizes_equals_reference",
    "test_encode_bitstream_identical_qbist_1080p",
    "test_encode_matches_golden_small_fixture",
    "test_encode_bitstream_identical_synthetic",
    "test_decode_reference_samples",
    "test_round_trip_psnr_1080p",
    "test_batched_device_resident_round_trip",
    "test_packed16_strip_kernels_equal_reference",
    "test_batched_path_of_the_other_configurations_equals_reference",
    "test_rg48_encode_bitstream_identical",
    "test_rg48_decode_equals_reference_exactly",
    "test_b64a_encode_bitstream_identical",
    "test_b64a_decode_equals_reference",
    "test_b64a_8k_config_c_round_trip",
    "test_byr4_encode_bitstream_identical",
    "test_interlaced_encode_bitstream_identical",
    "test_interlaced_encode_peak_table_frames",
    "test_interlaced_decode_reference_samples",
    "test_interlaced_decode_peak_table_frames",
    "test_yuy2_4k_two_segments",
    "test_encoder_pool_is_fifo_and_matches_sync",
    "test_concurrent_decoders_share_launches_and_stay_exact",
    "test_frame_queue_of_batches_equals_synchronous_passes",
    "test_gop_encode_bitstream_identical",
    "test_gop_decode_reference_samples",
    "test_gop_round_trip_of_the_product_alone",
    "test_gop_gates",
    "test_encoder_pool_and_decoders_over_several_devices_keep_order_and_bytes",
]
# ... and what runs last: the reference's own harness linked against the library (minutes of Qbist drawing on one core; compared with a committed fixture, no live reference).
GPU_LAST = ["test_reference_harness_links_unchanged_and_prints_same_numbers", "test_zz_every_reference_route_agreed"]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: k for k, name in enumerate(GPU_FIRST)}
    def key(item):
        if "test_gpu_" not in item.nodeid: return (0, 0)            # the other files keep their place in front
        name = getattr(item, "originalname", None) or item.name.split("[")[0]
        return (1, len(GPU_FIRST) + 1 + GPU_LAST.index(name) if name in GPU_LAST else rank.get(name, len(GPU_FIRST)))
    items.sort(key=key)                                               # stable: file order inside one rank


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The live-reference legs of the run (cfhd_testlib.reference_leg), route by route; the counts go last so that the tail of the output shows them."""
    import cfhd_testlib
    R = cfhd_testlib.REFERENCE_ROUTES
    if not R: return
    terminalreporter.section("live reference legs")
    for name, what, detail in cfhd_testlib.REFERENCE_DISAGREEMENTS:
        terminalreporter.write_line("never agreed: %s: %s %s" % (name, what, detail or ""))
    for what, (ok, fresh, never) in sorted(R.items()):
        if fresh or never: terminalreporter.write_line("%-48s agreed %d, only in a fresh process %d, never %d" % (what, ok, fresh, never))
    terminalreporter.write_line("live reference legs: %d agreed, %d agreed only with the reference in a fresh process, %d never agreed (%d routes, %d of them without any agreement)"
                                % (sum(v[0] for v in R.values()), sum(v[1] for v in R.values()), sum(v[2] for v in R.values()), len(R), sum(1 for v in R.values() if v[2] and not (v[0] or v[1]))))
