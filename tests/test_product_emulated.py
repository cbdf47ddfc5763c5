"""The GPU parity tests, run on the CPU against the emulated build of the whole product library (cfhd_testlib.product_emulated: the product's sources over
tests/hipemu/hip/hip_runtime.h + hip_emu.h -- same C ABI code, same job builders, same kernel source, kernels executed by fibers).

Why: the kernel-level emulation (test_kernels_emulated.py) fills its job structures itself; the tables the *product* builds between the C ABI and the kernels were
only ever exercised on hardware, and three bugs of this round lived exactly there.  This file takes every test of tests/test_gpu_parity.py and tests/test_gpu_gop.py
(they go through CFHD_* / cfhd_amd_batch_* only) and runs it here at the parameter sets that are small enough for the emulation, so a change to the device layer is
checked against the reference encoder / decoder and the oracle before it ever reaches a GPU.  The GPU suite itself is unchanged and remains the parity gate.

Test infrastructure only: nothing here is a product path (the product has no CPU path and fails loudly without a HIP device)."""
import itertools, os
import pytest
from cfhd_testlib import *
import test_gpu_parity, test_gpu_gop

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libcfhd_ref.so is not built")

# per case: frames x pixels up to this run by default (the whole file in about two minutes on one core); CFHD_EMU_MAX_PIXELS=2088960 adds the 1080p cases (seven minutes more)
MAX_PIXELS = int(os.environ.get("CFHD_EMU_MAX_PIXELS", 720 * 486))
# not for the emulation: launch sizes made for the hardware (bench-size batches, 8K, forced shapes over dozens of 1080p / 4K frames), the reference's harness
# binary (links libcfhd_amd.so itself)
SKIP = {"test_batched_round_trip_at_bench_sizes_equals_reference", "test_b64a_8k_config_c_round_trip", "test_reference_harness_links_unchanged_and_prints_same_numbers",
        "test_yuy2_4k_two_segments"}
# always: the register-strip kernels bench.py times, forced on the smallest batches of the GPU suite (their job tables and launch shapes come from the product)
ALWAYS = {("test_yuv422_strip_kernels_equal_reference", 1952, 250), ("test_yuv422_strip_kernels_equal_reference", 2304, 72), ("test_bayer_strip_kernel_equals_reference", 1008, 244), ("test_interlaced_strip_kernels_equal_reference", 2048, 120), ("test_interlaced_strip_kernels_equal_reference", 4000, 64), ("test_bayer_strip_kernel_equals_reference", 2032, 120), ("test_packed16_strip_kernels_equal_reference", 1016, 304), ("test_packed16_strip_kernels_equal_reference", 504, 242),
          ("test_packed16_strip_kernels_equal_reference", 136, 120)}


def _cases(module):
    out = []
    for name in sorted(dir(module)):
        fn = getattr(module, name)
        if not name.startswith("test_") or not callable(fn) or name in SKIP: continue
        marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
        axes = []
        for m in marks:
            names = [n.strip() for n in m.args[0].split(",")]
            axes.append([dict(zip(names, v if len(names) > 1 else (v,))) for v in m.args[1]])
        for combo in itertools.product(*axes):
            kw = {}
            for d in combo: kw.update(d)
            if "w" in kw and "h" in kw and kw["w"] * kw["h"] * kw.get("n", 1) > MAX_PIXELS and (name, kw["w"], kw["h"]) not in ALWAYS: continue
            ident = "-".join(str(v) for v in kw.values())
            out.append(pytest.param(module, name, kw, id="%s[%s]" % (name, ident) if kw else name))
    return out


@pytest.mark.parametrize("module,name,kw", _cases(test_gpu_parity) + _cases(test_gpu_gop))
def test_on_the_emulated_product(module, name, kw):
    with emulated_product():
        getattr(module, name)(**kw)
