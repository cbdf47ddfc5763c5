"""tests/hipemu/run_asan.py -- TEST INFRASTRUCTURE ONLY.

The emulated build of the whole product library (cfhd_testlib.product_emulated) once more with -fsanitize=address, and every case of tests/test_product_emulated.py run
against it: "device" memory is host memory there, so an out-of-bounds read or write of any kernel or job table is an AddressSanitizer report (the run stops at the
first one).  The fiber switch of hip_emu.h is a plain register swap; ASan needs no annotation for it (detect_stack_use_after_return stays off).

    python tests/hipemu/run_asan.py [substring of a test name ...]        # re-executes itself under LD_PRELOAD=libasan.so
"""
import os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(ROOT, "tests", "_build", "libcfhd_amd_hipemu_asan.so")


def build():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cfhd_testlib as T
    T.product_emulated()                                  # (leaves the translated sources under tests/_build/hipemu_product)
    csrc = os.path.join(ROOT, "cineform-sdk_amd", "csrc"); hipemu = os.path.join(ROOT, "tests", "hipemu"); gen = os.path.join(ROOT, "tests", "_build", "hipemu_product")
    srcs = [os.path.join(hipemu, "emu_runtime.cpp")] + [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".cpp")] + [os.path.join(gen, f) for f in sorted(os.listdir(gen)) if f.endswith(".cpp") and f.count(".") == 1]      # (not the <name>.<pid>.cpp leftovers of an interrupted build)
    deps = srcs + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-w", "-fsanitize=address", "-fno-omit-frame-pointer", "-std=c++17", "-fPIC", "-shared", "-pthread",
                               "-I" + hipemu, "-I" + csrc, "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", SO])


if __name__ == "__main__":
    if os.environ.get("CFHD_ASAN_CHILD") != "1":
        build()
        asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
        env = dict(os.environ, CFHD_ASAN_CHILD="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1")
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import cfhd_testlib as T
    L = ctypes.CDLL(SO); T.declare_cfhd_api(L); T._emu_product = L
    import test_gpu_parity, test_gpu_gop, test_product_emulated as E
    failed = 0
    for case in E._cases(test_gpu_parity) + E._cases(test_gpu_gop):
        module, name, kw = case.values
        if sys.argv[1:] and not any(a in name for a in sys.argv[1:]): continue
        t = time.time()
        try:
            with T.emulated_product(): getattr(module, name)(**kw)
            verdict = "ok"
        except BaseException as e:                       # (an ASan report ends the process; this is an ordinary test failure)
            verdict = "FAILED " + repr(e)[:200]; failed += 1
        print("%-80s %-40s %5.1fs %s" % (name, kw, time.time() - t, verdict), flush=True)
    sys.exit(1 if failed else 0)
