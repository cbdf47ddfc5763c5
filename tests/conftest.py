import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "cineform-sdk_amd", "python"))


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
    "test_reference_harness_links_unchanged_and_prints_same_numbers",
    "test_encoder_pool_is_fifo_and_matches_sync",
    "test_concurrent_decoders_share_launches_and_stay_exact",
]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: k for k, name in enumerate(GPU_FIRST)}
    def key(item):
        if "test_gpu_" not in item.nodeid: return (0, 0)            # the other files keep their place in front
        return (1, rank.get(getattr(item, "originalname", None) or item.name.split("[")[0], len(GPU_FIRST)))
    items.sort(key=key)                                               # stable: file order inside one rank
