"""The drop-in boundary: libcfhd_amd.so loads without a GPU and exports every function include/cfhd_amd.h declares (and nothing from the
test infrastructure: oracle, emulator, hooks)."""
import ctypes, os, re, subprocess
from cfhd_testlib import ROOT, PRODUCT_DIR

HEADER = os.path.join(ROOT, "include", "cfhd_amd.h")
LIB = os.path.join(PRODUCT_DIR, "libcfhd_amd.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    names = set()
    for stmt in text.split(";"):
        if "typedef" in stmt or "(" not in stmt: continue
        m = re.search(r"\b((?:CFHD_|cfhd_amd_)\w+)\s*\(", stmt)
        if m: names.add(m.group(1))
    return names


def exported_functions():
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_library_exports_every_declared_entry_point():
    decl = declared_functions()
    assert len(decl) > 40 and "CFHD_EncodeSample" in decl and "CFHD_DecodeSample" in decl and "cfhd_amd_batch_create_ex" in decl
    exp = exported_functions()
    missing = sorted(decl - exp)
    assert not missing, "declared in include/cfhd_amd.h but not exported: %s" % missing


def test_library_loads_without_a_gpu_and_resolves_the_entry_points():
    L = ctypes.CDLL(LIB)
    for name in declared_functions():
        assert getattr(L, name) is not None


def test_no_test_infrastructure_in_the_product_library():
    exp = exported_functions()
    bad = sorted(n for n in exp if n.startswith(("orc_", "emu_", "hipemu")) or "hook" in n.lower())
    assert not bad, bad
    undocumented = sorted(n for n in exp if n.startswith(("CFHD_", "cfhd_amd_")) and n not in declared_functions())
    assert not undocumented, "exported but not declared in the header: %s" % undocumented


def test_library_exports_exactly_the_declared_c_abi():
    """cineform-sdk_amd/exports.map: nothing but the CFHD_* / cfhd_amd_* entry points leaves the library -- no mangled C++ (cfhd::EncodeBatch::...), no kernel host
    stubs (__device_stub__k_*), no data symbols: a drop-in for libCFHDCodec does not leak its object model."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    extra = sorted(n for n in names if n not in declared_functions())
    assert not extra, "exported beyond the C ABI: %s" % extra[:20]
