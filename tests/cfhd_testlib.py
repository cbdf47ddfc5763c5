"""Shared helpers for the tests: load the oracle (oracle/libcfhd_oracle.so), the compiled reference
(oracle/_ref/libcfhd_ref.so, when present) and the product (cineform-sdk_amd/libcfhd_amd.so) via ctypes.

oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
"""
import ctypes, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libcfhd_ref.so")
ORACLE_SO = os.path.join(ORACLE_DIR, "libcfhd_oracle.so")

c_i16p = ctypes.POINTER(ctypes.c_int16)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_intp = ctypes.POINTER(ctypes.c_int)


def fourcc(s):
    return (ord(s[0]) << 24) | (ord(s[1]) << 16) | (ord(s[2]) << 8) | ord(s[3])


PIX_YUY2 = fourcc("YUY2")
PIX_2VUY = fourcc("2vuy")
PIX_RG48 = fourcc("RG48")
PIX_B64A = fourcc("b64a")
PIX_BYR4 = fourcc("BYR4")
PIX_YU64 = fourcc("YU64")
PIX_V210 = fourcc("v210")
PIX_RG24 = fourcc("RG24")
PIX_BGRA = fourcc("BGRA")
PIX_BGRa = fourcc("BGRa")
# 10-bit RGB in 32-bit words: name -> (byte order of the word, bit positions of R, G, B, COLOR_FORMAT code)
RGB10_FORMATS = {"r210": (">", (20, 10, 0), 123), "DPX0": (">", (22, 12, 2), 128), "AB10": ("<", (0, 10, 20), 125), "AR10": ("<", (20, 10, 0), 124)}
ENCODED_BAYER = 3       # CFHD_ENCODED_FORMAT_BAYER
COLOR_FORMAT_BYR4 = 104 # Codec/color.h
ENCODED_RGBA4444 = 2    # CFHD_ENCODED_FORMAT_RGBA_4444
COLOR_FORMAT_B64A = 30  # COLOR_FORMAT_BGRA64, Codec/color.h
ENCODED_RGB444 = 1      # CFHD_ENCODED_FORMAT_RGB_444
COLOR_FORMAT_RG48 = 120 # Codec/color.h
ENCODED_YUV422 = 0      # CFHD_ENCODED_FORMAT_YUV_422
QUALITY_FILMSCAN1 = 4   # CFHD_ENCODING_QUALITY_FILMSCAN1
COLOR_FORMAT_UYVY = 1   # Codec/color.h:64
COLOR_FORMAT_YUYV = 2   # Codec/color.h:65


def p16(a):
    assert a.dtype == np.int16 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_i16p)


def _build_once(target, cmd, deps):
    """Compile `target` under a file lock (pytest-xdist workers find the same stale library at the same time) and move it into place whole."""
    import fcntl
    with open(target + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps):      # (a worker in front of us may have built it)
            tmp = "%s.%d.tmp" % (target, os.getpid())
            subprocess.check_call(cmd + ["-o", tmp])
            os.replace(tmp, target)
        fcntl.flock(lock, fcntl.LOCK_UN)


def p8(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_u8p)


def iarr(vals):
    return (ctypes.c_int * len(vals))(*vals)


def declare_cfhd_api(L):
    """ctypes prototypes of the CFHD_* C ABI (identical for the reference build and for libcfhd_amd.so)."""
    vp, vpp, i, u32, sz = ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t
    protos = {
        "CFHD_OpenEncoder": [vpp, vp], "CFHD_PrepareToEncode": [vp, i, i, u32, i, u32, i], "CFHD_EncodeSample": [vp, vp, i],
        "CFHD_GetSampleData": [vp, vpp, ctypes.POINTER(sz)], "CFHD_CloseEncoder": [vp],
        "CFHD_MetadataOpen": [vpp], "CFHD_MetadataAdd": [vp, u32, i, sz, vp, ctypes.c_bool], "CFHD_MetadataAttach": [vp, vp], "CFHD_MetadataClose": [vp],
        "CFHD_CreateEncoderPool": [vpp, i, i, vp], "CFHD_PrepareEncoderPool": [vp, ctypes.c_uint16, ctypes.c_uint16, u32, i, u32, i],
        "CFHD_AttachEncoderPoolMetadata": [vp, vp], "CFHD_StartEncoderPool": [vp], "CFHD_StopEncoderPool": [vp],
        "CFHD_EncodeAsyncSample": [vp, u32, vp, ctypes.c_ssize_t, vp], "CFHD_WaitForSample": [vp, ctypes.POINTER(u32), vpp],
        "CFHD_TestForSample": [vp, ctypes.POINTER(u32), vpp], "CFHD_GetEncodedSample": [vp, vpp, ctypes.POINTER(sz)],
        "CFHD_ReleaseSampleBuffer": [vp, vp], "CFHD_ReleaseEncoderPool": [vp],
        "CFHD_OpenDecoder": [vpp, vp], "CFHD_GetSampleInfo": [vp, vp, sz, i, vp, sz],
        "CFHD_PrepareToDecode": [vp, i, i, u32, i, u32, vp, sz, ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(u32)],
        "CFHD_GetPixelSize": [u32, ctypes.POINTER(u32)], "CFHD_GetImagePitch": [u32, u32, ctypes.POINTER(ctypes.c_int32)],
        "CFHD_GetImageSize": [u32, u32, u32, i, i, ctypes.POINTER(u32)], "CFHD_DecodeSample": [vp, vp, sz, vp, ctypes.c_int32],
        "CFHD_SetActiveMetadata": [vp, vp, ctypes.c_uint, i, vp, ctypes.c_uint], "CFHD_CloseDecoder": [vp],
        "CFHD_OpenMetadata": [vpp], "CFHD_InitSampleMetadata": [vp, i, vp, sz], "CFHD_CloseMetadata": [vp],
    }
    for name, args in protos.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = ctypes.c_int


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"])
        _oracle = ctypes.CDLL(ORACLE_SO)
        _oracle.orc_vlc_encode_band.restype = ctypes.c_size_t
        _oracle.orc_vlc_encode_band.argtypes = [c_i16p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_u8p, ctypes.c_size_t]
        _oracle.orc_vlc_decode_band.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i16p]
    return _oracle


_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        _ref = ctypes.CDLL(REF_SO)
        _ref.ref_psnr.restype = ctypes.c_float
        declare_cfhd_api(_ref)
    return _ref


class OrcQuant(ctypes.Structure):
    _fields_ = [("num_channels", ctypes.c_int), ("prescale", ctypes.c_int * 8),
                ("quant", ((ctypes.c_int * 4) * 3) * 4), ("scale", ((ctypes.c_int * 4) * 3) * 4),
                ("midpoint_prequant", ctypes.c_int)]


def orc_quant_tables(quality=QUALITY_FILMSCAN1, precision=10, chroma_full=0, channels=3, progressive=1):
    q = OrcQuant()
    oracle().orc_quant_tables(quality, precision, chroma_full, channels, progressive, ctypes.byref(q))
    return q


def qbist_frames(seed, count, width=1920, height=1080, pixfmt=PIX_YUY2, alpha=0):
    """Frames exactly as Example/TestCFHD.cpp:1149-1150,1208,1216-1220 generates them (QBIST_UNIQUE)."""
    L = ref()
    pitch = L.ref_frame_pitch(pixfmt, width)
    L.ref_qbist_reset(seed)
    buf = np.zeros(width * height * 8, dtype=np.uint8)
    frames = []
    for _ in range(count):
        L.ref_qbist_frame(width, height, pitch, pixfmt, alpha, buf.ctypes.data_as(ctypes.c_void_p))
        frames.append(buf[: pitch * height].copy())
    return frames, pitch


def synth_yuy2(width, height, seed):
    """Deterministic synthetic 4:2:2 frame that does not need the reference (smooth gradients + texture + noise)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width]
    luma = 128 + 90 * np.sin(x / 37.0 + seed) * np.cos(y / 23.0) + 20 * np.sin((x * y) / 5000.0) + rng.normal(0, 3, (height, width))
    cb = 128 + 60 * np.sin(x / 91.0) + rng.normal(0, 2, (height, width))
    cr = 128 + 60 * np.cos(y / 67.0) + rng.normal(0, 2, (height, width))
    f = np.zeros((height, width * 2), dtype=np.uint8)
    f[:, 0::2] = np.clip(luma, 0, 255).astype(np.uint8)
    f[:, 1::4] = np.clip(cb[:, 0::2], 0, 255).astype(np.uint8)
    f[:, 3::4] = np.clip(cr[:, 0::2], 0, 255).astype(np.uint8)
    return f.reshape(-1).copy(), width * 2


def feedback_test_frames(w=640, h=360, n=6):
    """Frames whose compressed size changes from one to the next (noise of varying strength on the gradients), so that the rate
    feedback of FILMSCAN2/3 and the bit-rate limiter have something to react to; the samples stay below the sample buffer size."""
    rng = np.random.default_rng(3)
    frames = []
    for i in range(n):
        f = synth_yuy2(w, h, 40 + i)[0].reshape(h, w * 2).astype(np.int32)
        f += rng.integers(-9, 10, f.shape) * (1 + i % 3)
        frames.append(np.clip(f, 0, 255).astype(np.uint8).reshape(-1).copy())
    return frames


def ref_encode_frames(frames, pitch, width, height, pixfmt=PIX_YUY2, encoded=ENCODED_YUV422, quality=QUALITY_FILMSCAN1, flags=0):
    """Encode through the reference's own C ABI (CFHD_OpenEncoder ... CFHD_GetSampleData); returns list of bytes."""
    L = ref()
    enc = ctypes.c_void_p()
    assert L.CFHD_OpenEncoder(ctypes.byref(enc), None) == 0
    assert L.CFHD_PrepareToEncode(enc, width, height, pixfmt, encoded, flags, quality) == 0
    out = []
    for f in frames:
        assert L.CFHD_EncodeSample(enc, f.ctypes.data_as(ctypes.c_void_p), pitch) == 0
        p = ctypes.c_void_p(); n = ctypes.c_size_t()
        assert L.CFHD_GetSampleData(enc, ctypes.byref(p), ctypes.byref(n)) == 0
        out.append(ctypes.string_at(p, n.value))
    L.CFHD_CloseEncoder(enc)
    return out


TAG_CPU_MAX = fourcc("CPUM")        # Common/CFHDMetadataTags.h:271
TAG_PROCESS_PATH = fourcc("PRCS")   # Common/CFHDMetadataTags.h:232
METADATATYPE_UINT32 = 2             # Common/CFHDTypes.h:311


class RefDecoder:
    """One decoder handle of the reference, opened and configured exactly as Example/TestCFHD.cpp:218-437 does, including its
    CFHD_SetActiveMetadata(TAG_PROCESS_PATH / TAG_CPU_MAX) calls (:338-356).  The harness caps the decoder's worker threads at 16; the
    parity tests use ONE: without a cap the reference starts a worker per core (256 on the GPU host), and its workers race on
    decoder->frame.alpha_Companded (Codec/bayer.c:13871 written by whichever finishes first, read at :16034) and have been seen to damage
    frames -- with a single worker the reference decoder is deterministic (except for its rand() dither)."""

    def __init__(self, sample, pixfmt=PIX_YUY2, resolution=1, cpus=1):
        L = self.L = ref()
        self.dec = ctypes.c_void_p(); self.md = ctypes.c_void_p()
        assert L.CFHD_OpenDecoder(ctypes.byref(self.dec), None) == 0
        aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
        sb = ctypes.create_string_buffer(sample, len(sample))
        assert L.CFHD_PrepareToDecode(self.dec, 0, 0, pixfmt, resolution, 0, sb, 512, ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0
        pitch = ctypes.c_int32()
        assert L.CFHD_GetImagePitch(aw.value, af.value, ctypes.byref(pitch)) == 0
        self.width, self.height, self.pitch = aw.value, ah.value, pitch.value
        if cpus:
            assert L.CFHD_OpenMetadata(ctypes.byref(self.md)) == 0
            assert L.CFHD_InitSampleMetadata(self.md, 0, sb, len(sample)) == 0           # METADATATYPE_ORIGINAL
            for tag, value in ((TAG_PROCESS_PATH, 0xffff), (TAG_CPU_MAX, cpus)):          # PROCESSING_ALL_ON, as the harness
                v = ctypes.c_uint32(value)
                assert L.CFHD_SetActiveMetadata(self.dec, self.md, tag, METADATATYPE_UINT32, ctypes.byref(v), 4) == 0

    def decode(self, sample_buffer, size, out, pitch=None):
        """sample_buffer: ctypes string buffer; out: numpy uint8 array of at least pitch * height bytes.  Returns the CFHD_Error."""
        return self.L.CFHD_DecodeSample(self.dec, sample_buffer, size, out.ctypes.data_as(ctypes.c_void_p), pitch or self.pitch)

    def close(self):
        self.L.CFHD_CloseDecoder(self.dec)
        if self.md: self.L.CFHD_CloseMetadata(self.md)


def ref_decode_sample(sample, width, height, pixfmt=PIX_YUY2, resolution=1, cpus=1):
    """Decode through the reference's C ABI (resolution 1 = full, 2 = half) with `cpus` decoder worker threads (RefDecoder)."""
    if _REF_FRESH_PROCESS and not _IN_FRESH_CHILD:             # reference_leg escalated: this process's reference state is suspect, ask a fresh one
        return ref_decode_sample_fresh_process(sample, width, height, pixfmt, resolution, cpus)
    d = RefDecoder(sample, pixfmt, resolution, cpus)
    sb = ctypes.create_string_buffer(sample, len(sample))
    out = np.zeros(d.pitch * d.height + 64, dtype=np.uint8)
    assert d.decode(sb, len(sample), out) == 0
    d.close()
    return out[: d.pitch * d.height].copy(), d.pitch


_REF_FRESH_PROCESS = False       # set by reference_leg() while it repeats a leg with the reference in a fresh process
_IN_FRESH_CHILD = bool(os.environ.get("CFHD_TEST_FRESH_CHILD"))


def ref_decode_sample_fresh_process(sample, width, height, pixfmt=PIX_YUY2, resolution=1, cpus=1, env_pad=None):
    """ref_decode_sample in a child process of its own.  For routes of the reference whose output depends on what the process did before (its half-resolution
    RG48 decode of RGBA 4:4:4:4 samples returns other words in one colour component once `import torch` has run in the process -- uninitialised state
    somewhere in its planar rows; a fresh process gives the same words every time)."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "sample"), "wb").write(sample)
        code = ("import sys; sys.path.insert(0, %r); import numpy as np; from cfhd_testlib import *; "
                "out, pitch = ref_decode_sample(open(%r, 'rb').read(), %d, %d, %d, %d, %d); open(%r, 'wb').write(out.tobytes()); print(pitch)"
                % (os.path.join(ROOT, "tests"), os.path.join(d, "sample"), width, height, pixfmt, resolution, cpus, os.path.join(d, "out")))
        # env_pad: run the child with a minimal environment + a variable of that many bytes.  (Found in round 4: the answer of that route also depends on the size of the
        # process's environment block -- an extra variable in os.environ flipped it -- i.e. on stack contents the reference never initialised; a caller that pins
        # arithmetic on such a route tries a few sizes.)
        env = dict(os.environ, CFHD_TEST_FRESH_CHILD="1") if env_pad is None else {"PATH": os.environ.get("PATH", "/usr/bin:/bin"), "HOME": os.environ.get("HOME", "/tmp"), "CFHD_TEST_PAD": "x" * env_pad, "CFHD_TEST_FRESH_CHILD": "1"}
        pitch = int(subprocess.check_output([sys.executable, "-c", code], env=env).split()[-1])
        return np.frombuffer(open(os.path.join(d, "out"), "rb").read(), np.uint8).copy(), pitch


REFERENCE_DISAGREEMENTS = []      # (test, route, detail) of every live-reference leg that never agreed, not even with the reference in a fresh process
REFERENCE_ROUTES = {}             # route ("what") -> [legs that agreed in this process, legs that agreed only with the reference in a fresh process, legs that never agreed]


def reference_leg(agree, attempts=6, what="", racy=False):
    """The live-reference leg of a GPU test.  The *gate* of such a test is product == oracle (the oracle's model of the route is pinned on the reference by
    tests/test_oracle_vs_ref.py on the CPU, where the reference runs with one worker on a quiet 8-core host); this leg runs the reference decoder once more on
    the GPU box beside it as a witness.  `agree()` runs the reference and returns True (or None) when its output agrees with what the test holds, False or a
    string (the detail) otherwise; an exception it raises counts as a disagreement with the exception as its detail.

    What is tolerated, and what is not (round 5; the advisor's finding on round 4's version, which swallowed everything):
      * The reference decoder is a threaded third party with a rand() dither: a leg gets `attempts` runs in this process.
      * Some of its routes answer differently depending on what the process did before (uninitialised rows: the same sample decodes to the exact words in a
        fresh process and to other words late in a long pytest process -- reproduced on the build container, the 16-bit routes' "empty detail" entries of round 4).
        A leg that never agreed here is therefore repeated with the reference in a FRESH process (ref_decode_sample does that while _REF_FRESH_PROCESS is set);
        agreement there is recorded as "fresh process only" and is not a failure.
      * A leg that does not agree in a fresh process either is a finding about the product or the oracle: recorded, written to gpurun_out/reference_disagreements.log,
        and an assertion on the spot -- strict is the default since round 6 (the fresh-process escalation above absorbs the reference's history dependence; rounds 4
        and 5 ran lenient by default and failed only per route at the end of the run, so a route that agreed at 320x240 and never at 1080p would have passed).
        CFHD_REFERENCE_LEG=lenient: the suite goes on, tests/test_gpu_parity.py::test_zz_every_reference_route_agreed still fails the run at its end when a ROUTE never
        agreed on any of its legs.  racy=True: a leg over one of the reference's two documented races (the alpha flag a worker thread sets while others still convert
        rows, bayer.c:13871 / :16034; rows below a height that is no multiple of 8) never asserts by itself.  The last lines of the pytest output carry the counts (tests/conftest.py)."""
    global _REF_FRESH_PROCESS
    route = REFERENCE_ROUTES.setdefault(what, [0, 0, 0])
    def once():
        try:
            r = agree()
        except Exception as e:                      # noqa: BLE001 -- an error code or an exception of the reference is a finding too; it is reported, never swallowed
            return "reference leg raised %s: %s" % (type(e).__name__, e)
        if r is None or (not isinstance(r, str) and bool(r)): return None            # (numpy booleans are not `True`)
        return r if isinstance(r, str) else ""
    detail = None
    for attempt in range(attempts):
        detail = once()
        if detail is None: route[0] += 1; return True
    here = detail
    _REF_FRESH_PROCESS = True
    try:
        for attempt in range(2):
            detail = once()
            if detail is None: break
    finally:
        _REF_FRESH_PROCESS = False
    name = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    import warnings
    if detail is None:
        route[1] += 1
        warnings.warn("live reference agreed only in a fresh process (in this process: %s): %s %s" % (here or "different output", name, what))
        return True
    route[2] += 1
    REFERENCE_DISAGREEMENTS.append((name, what, detail))
    warnings.warn("live reference never agreed in %d attempts + 2 in fresh processes: %s %s %s" % (attempts, name, what, detail))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "reference_disagreements.log"), "a") as f:
            f.write("%s\t%s\t%s\n" % (name, what, detail))
    except OSError:
        pass
    assert racy or os.environ.get("CFHD_REFERENCE_LEG", "strict") != "strict", "live reference disagrees on %s: %s" % (what, detail)
    return False


def _metadata_chunks(sample):
    """(offset, size) of the payload of every metadata chunk in the header of a sample: the tag / value walk of Codec/decoder.c UpdateCodecState as far as the first
    lowpass marker -- an optional (negated) tag with bit 0x4000 carries a payload of `value` (bit 0x2000: (tag & 0xff) << 16 | value) longwords; 0x4002 = TAG_METADATA."""
    out = []; pos = 0; n = len(sample)
    while pos + 4 <= n:
        tag = (sample[pos] << 8) | sample[pos + 1]; val = (sample[pos + 2] << 8) | sample[pos + 3]
        pos += 4
        if tag & 0x8000: tag = (-tag) & 0xffff
        if tag == 4 and val in (0x1A4A, 0x0D0D): break              # TAG_MARKER lowpass / highpass start: the header is behind us
        if tag & 0x4000:
            size = (((tag & 0xff) << 16) | val) * 4 if tag & 0x2000 else val * 4
            if (tag & 0xff00) == 0x4000 and (tag & 0xff) == 0x02 or tag == 0x4002: out.append((pos, min(size, n - pos)))
            pos += size
        elif tag == 2: pos += 4 * val                             # TAG_INDEX: the channel size entries
    return out


def mask_volatile_metadata(sample):
    """Zero the bytes of a sample that legitimately differ between two encoders: the payloads of the GUID / DATE / TIME / TIMC tuples of its metadata chunks
    (EncoderSDK/SampleEncoder.cpp:764,786-787,806-814) -- wherever in the header they sit: a large user block in front of them pushes them far behind the first
    kilobyte, and two encoders that run on either side of a full second write different TIME strings (a one-in-twenty flake of the large-metadata test until round 4)."""
    b = bytearray(sample)
    regions = _metadata_chunks(bytes(sample)) or [(0, min(len(b), 1024))]
    for lo, size in regions:
        chunk = bytes(b[lo: lo + size]); at = 0
        while at + 8 <= len(chunk):                                  # tuples: FourCC, 24-bit little-endian size + type byte, payload padded to a longword
            tag = chunk[at: at + 4]; tsize = chunk[at + 4] | (chunk[at + 5] << 8) | (chunk[at + 6] << 16)
            if tag in (b"GUID", b"DATE", b"TIME", b"TIMC"):
                for k in range(min(tsize, len(chunk) - at - 8)): b[lo + at + 8 + k] = 0
            at += 8 + (tsize + 3) // 4 * 4
    return bytes(b)


# ------------------------------------------------------------------------------------------
# product library (cineform-sdk_amd/libcfhd_amd.so)
# ------------------------------------------------------------------------------------------
PRODUCT_DIR = os.path.join(ROOT, "cineform-sdk_amd")
PRODUCT_SO = os.environ.get("CFHD_AMD_LIB") or os.path.join(PRODUCT_DIR, "libcfhd_amd.so")      # (CFHD_AMD_LIB: A/B runs of another build of the library, tools/gpu_r06_*.sh)
PIXKIND = {"YUY2": 1, "2vuy": 2, "RG48": 3, "b64a": 4, "BYR4": 5, "YU64": 6, "v210": 7, "RG24": 8, "BGRA": 9, "BGRa": 10, "r210": 11, "DPX0": 12, "AB10": 13, "AR10": 14, "RG64": 15, "BYR5": 16}
ENC = {"422": 1, "bayer": 2, "444": 3, "4444": 4}
_product = None


def product():
    """libcfhd_amd.so: the CFHD_* C ABI and the cfhd_amd_batch_* extension (needs a GPU for anything that computes).
    Inside `with emulated_product():` the same sources built over the CPU stand-in for the HIP runtime (product_emulated)."""
    global _product
    if _use_emulated_product:
        return product_emulated()
    if _product is None:
        if not os.path.exists(PRODUCT_SO):
            subprocess.check_call(["make", "-C", PRODUCT_DIR])
        L = ctypes.CDLL(PRODUCT_SO)
        declare_cfhd_api(L)
        _product = L
    return _product


EMU_PRODUCT_SO = os.path.join(ROOT, "tests", "_build", "libcfhd_amd_hipemu.so")
_emu_product = None
_use_emulated_product = False


def product_emulated():
    """TEST INFRASTRUCTURE: the whole product library -- C ABI, batch front end, job builders, entropy drivers, unmodified kernel source -- compiled by g++ over
    tests/hipemu/hip/hip_runtime.h (host memory, synchronous copies) and hip_emu.h (fibers for GPU threads); kernel launches rewritten by
    tests/hipemu/translate_launches.py.  Lets the CPU suite drive the C ABI end to end at small frame sizes, in particular the job tables between the ABI and
    the kernels, which the kernel-level emulation (emu()) never saw.  Never part of libcfhd_amd.so; the product has no CPU path."""
    global _emu_product
    if _emu_product is None:
        csrc = os.path.join(PRODUCT_DIR, "csrc"); hipemu = os.path.join(ROOT, "tests", "hipemu")
        gen = os.path.join(ROOT, "tests", "_build", "hipemu_product")
        deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(hipemu, f) for f in ("hip_emu.h", "cfhd_gfx950.h", "emu_runtime.cpp", "translate_launches.py")] + [
            os.path.join(hipemu, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "cfhd_amd.h")]
        if not os.path.exists(EMU_PRODUCT_SO) or any(os.path.getmtime(d) > os.path.getmtime(EMU_PRODUCT_SO) for d in deps):
            os.makedirs(gen, exist_ok=True)
            sys.path.insert(0, hipemu)
            from translate_launches import translate
            sys.path.pop(0)
            srcs = [os.path.join(hipemu, "emu_runtime.cpp")] + [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".cpp")]
            for f in sorted(os.listdir(csrc)):
                if f.endswith(".hip"):
                    out = os.path.join(gen, f[:-4] + ".%d.cpp" % os.getpid())
                    with open(out, "w") as fh: fh.write("// generated from cineform-sdk_amd/csrc/%s by tests/hipemu/translate_launches.py -- test infrastructure\n" % f + translate(open(os.path.join(csrc, f)).read()))
                    srcs.append(out)
            _build_once(EMU_PRODUCT_SO, ["g++", "-O1", "-w", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + hipemu, "-I" + csrc, "-I" + os.path.join(ROOT, "include")] + srcs, deps)
            for f in srcs:
                if f.startswith(gen): os.replace(f, os.path.join(gen, os.path.basename(f).split(".")[0] + ".cpp"))      # (kept for reading: what the compiler saw)
        # Four emulated GPUs with separate, protected heaps (tests/hipemu/hip/hip_runtime.h): an unpinned process deals its pool workers and decoder handles over all
        # of them (cfhd_core.h unit_device), so every pool / handle test of the emulated suite is a several-device run in which a pointer, a stream or an event that
        # crosses devices -- or host code that dereferences device memory -- ends the test loudly.  (Read when the library makes its first HIP call.)
        os.environ.setdefault("HIPEMU_DEVICES", "4")
        os.environ.setdefault("CFHD_AMD_STAGE_MIN_BYTES", "4096")      # plain host buffers are staged in pieces (cfhd_device.hip upload_frame / download_frame): at every frame size here, not only from 1 MB on
        L = ctypes.CDLL(EMU_PRODUCT_SO)
        declare_cfhd_api(L)
        _emu_product = L
    return _emu_product


class emulated_product:
    """with emulated_product(): amd_encode_frames / amd_decode_sample / product() address the emulated build of the product library."""
    def __enter__(self):
        global _use_emulated_product
        self.before = _use_emulated_product; _use_emulated_product = True
        return product_emulated()
    def __exit__(self, *a):
        global _use_emulated_product
        _use_emulated_product = self.before


HOOKS_SO = os.path.join(ROOT, "tests", "_build", "libcfhd_hooks.so")
_hooks = None


def hooks():
    """Test-only library: tests/hooks/cfhd_hooks.cpp + the product's host-only sources (plan geometry, quantizer derivation, sample
    writer / parser, host VLC), compiled with g++.  Not part of libcfhd_amd.so."""
    global _hooks
    if _hooks is None:
        csrc = os.path.join(PRODUCT_DIR, "csrc")
        srcs = [os.path.join(ROOT, "tests", "hooks", "cfhd_hooks.cpp")] + [os.path.join(csrc, f) for f in ("cfhd_tables.cpp", "cfhd_bitstream.cpp", "cfhd_metadata.cpp", "cfhd_gop.cpp")]
        deps = srcs + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
        if not os.path.exists(HOOKS_SO) or any(os.path.getmtime(d) > os.path.getmtime(HOOKS_SO) for d in deps):
            os.makedirs(os.path.dirname(HOOKS_SO), exist_ok=True)
            _build_once(HOOKS_SO, ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + csrc, "-I" + os.path.join(ROOT, "include")] + srcs, deps)
        L = ctypes.CDLL(HOOKS_SO)
        L.cfhd_amd_write_sample_host.restype = ctypes.c_size_t
        L.cfhd_amd_write_sample_host.argtypes = [ctypes.c_int] * 8 + [ctypes.c_uint, c_i16p, c_u8p, ctypes.c_size_t, c_u8p, ctypes.c_size_t, c_u8p, ctypes.c_size_t]
        L.cfhd_amd_decode_bands_host.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_int, c_i16p, ctypes.c_size_t, c_intp, ctypes.c_int]
        L.cfhd_amd_quant_sequence.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, c_intp]
        L.cfhd_amd_sample_quants.argtypes = [c_u8p, ctypes.c_size_t, c_intp]
        _hooks = L
    return _hooks


class Plan:
    """Python view of cfhd::FramePlan (geometry + quantizer) as the product derives it."""

    def __init__(self, width, height, pixkind=1, enc=1, quality=QUALITY_FILMSCAN1, progressive=1):
        buf = (ctypes.c_int * 512)()
        n = hooks().cfhd_amd_plan_info(width, height, pixkind, enc, quality, progressive, buf)
        assert n > 0, "plan_info failed"
        v = list(buf[:n])
        self.coeff_elems, self.final_elems, self.num_channels, self.precision, self.mpq = v[0:5]
        self.prescale = v[5:8]
        self.band = {}
        i = 8
        for c in range(self.num_channels):
            for lv in range(3):
                for b in range(4):
                    w, h, pitch, off, quant, scale = v[i:i + 6]; i += 6
                    self.band[(c, lv, b)] = dict(width=w, height=h, pitch=pitch, offset=off, quant=quant, scale=scale)
        self.width, self.height, self.pixkind, self.enc, self.quality = width, height, pixkind, enc, quality

    def view(self, coeffs, c, lv, b):
        d = self.band[(c, lv, b)]
        return coeffs[d["offset"]: d["offset"] + d["pitch"] * d["height"]].reshape(d["height"], d["pitch"])


def pad_yuv422_rows(plan, frame, pitch):
    """The codec works on a height rounded up to a multiple of 8; the encoder fills the extra rows of a packed 4:2:2 frame with 0x80
    bytes (encoder.c:2442-2478).  Returns (frame, encoded height)."""
    H = plan.band[(0, 0, 0)]["height"] * 2
    if H == plan.height: return frame, H
    padded = np.full(H * pitch, 0x80, dtype=np.uint8)
    padded[: plan.height * pitch] = np.asarray(frame).reshape(-1)[: plan.height * pitch]
    return padded, H


def oracle_forward_yuv422(plan, frame, pitch, uyvy=0):
    """Whole forward path of one 4:2:2 frame with the oracle, written into the product's pyramid layout."""
    O = oracle()
    coeffs = np.zeros(plan.coeff_elems, dtype=np.int16)
    frame, H = pad_yuv422_rows(plan, frame, pitch)
    for c in range(3):
        cw = plan.width if c == 0 else plan.width // 2
        # level 1 straight from the packed frame
        q = [plan.band[(c, 0, b)]["quant"] for b in range(4)]
        outs = [plan.view(coeffs, c, 0, b) for b in range(4)]
        assert len({o.shape[1] for o in outs}) == 1
        bands = (c_i16p * 4)(*[o.ctypes.data_as(c_i16p) for o in outs])
        O.orc_fwd_spatial_yuv422(p8(frame), pitch, cw, H, c, plan.precision - 8, uyvy, iarr(q), plan.mpq, bands, outs[0].shape[1])
        for lv in (1, 2):
            src = plan.view(coeffs, c, lv - 1, 0)
            d = plan.band[(c, lv - 1, 0)]
            q = [plan.band[(c, lv, b)]["quant"] for b in range(4)]
            outs = [plan.view(coeffs, c, lv, b) for b in range(4)]
            bands = (c_i16p * 4)(*[o.ctypes.data_as(c_i16p) for o in outs])
            O.orc_fwd_spatial(src.ctypes.data_as(c_i16p), d["pitch"], d["width"], d["height"], plan.prescale[lv], iarr(q), plan.mpq, bands, outs[0].shape[1])
    return coeffs


def rg48_planes(frame, pitch, w, h):
    """G, R, B planes (12-bit, value >> 4) of an RG48 frame, the order and scaling of ConvertRGB48ToFrame16s (frame.c:6128-6157)."""
    px = np.frombuffer(frame.tobytes(), dtype=np.uint16).reshape(h, pitch // 2)[:, : w * 3].reshape(h, w, 3)
    return [(px[:, :, k] >> 4).astype(np.int16) for k in (1, 0, 2)]


def b64a_planes(frame, pitch, w, h):
    """G, R, B, A planes of a b64a frame (words A, R, G, B; value >> 4; alpha companded for 0 < a < 4095), as
    ConvertBGRA64ToFrame_4444_16s builds them (frame.c:6676-6707)."""
    px = np.frombuffer(frame.tobytes(), dtype=np.uint16).reshape(h, pitch // 2)[:, : w * 4].reshape(h, w, 4)
    a = (px[:, :, 0] >> 4).astype(np.int32)
    a = np.where((a > 0) & (a < 4095), ((a * 223 + 128) >> 8) + 256, a)
    return [(px[:, :, 2] >> 4).astype(np.int16), (px[:, :, 1] >> 4).astype(np.int16), (px[:, :, 3] >> 4).astype(np.int16), a.astype(np.int16)]


def synth_bayer(width, height, seed):
    """Deterministic 16-bit Bayer mosaic (red-green order) with smooth structure, texture and noise; TestCFHD has no BYR4 generator."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width]
    lum = (np.sin(x / 37.0 + seed) * np.cos(y / 23.0) * 0.35 + 0.45) * 52000 + (x * y % 4099) * 2.0 + rng.normal(0, 120, (height, width))
    gain = np.where((y % 2 == 0) & (x % 2 == 0), 0.8, np.where((y % 2 == 1) & (x % 2 == 1), 0.6, 1.0))       # R, B darker than the greens
    return np.clip(lum * gain, 0, 65535).astype(np.uint16)


def synth_v210(width, height, seed):
    """A 10-bit 4:2:2 picture and its v210 packing (six pixels in four little-endian 32-bit words: Cb0 Y0 Cr0 | Y1 Cb1 Y2 | Cr1 Y3 Cb2 |
    Y4 Cr2 Y5; rows padded to 48 pixels = 128 bytes).  TestCFHD has no v210 generator.  Returns (frame bytes, pitch, Y, Cb, Cr)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width]
    Y = np.clip(512 + 300 * np.sin(xx / 23.0 + seed) * np.cos(yy / 17.0) + rng.normal(0, 6, (height, width)), 4, 1019).astype(np.uint32)
    Cb = np.clip(512 + 200 * np.sin(xx[:, ::2] / 41.0) + rng.normal(0, 4, (height, width // 2)), 4, 1019).astype(np.uint32)
    Cr = np.clip(512 + 200 * np.cos(yy[:, ::2] / 31.0 + seed) + rng.normal(0, 4, (height, width // 2)), 4, 1019).astype(np.uint32)
    wp = (width + 47) // 48 * 48
    Yp = np.zeros((height, wp), np.uint32); Yp[:, :width] = Y
    Cbp = np.zeros((height, wp // 2), np.uint32); Cbp[:, : width // 2] = Cb
    Crp = np.zeros((height, wp // 2), np.uint32); Crp[:, : width // 2] = Cr
    out = np.zeros((height, wp // 6 * 4), np.uint32)
    out[:, 0::4] = Cbp[:, 0::3] | (Yp[:, 0::6] << 10) | (Crp[:, 0::3] << 20)
    out[:, 1::4] = Yp[:, 1::6] | (Cbp[:, 1::3] << 10) | (Yp[:, 2::6] << 20)
    out[:, 2::4] = Crp[:, 1::3] | (Yp[:, 3::6] << 10) | (Cbp[:, 2::3] << 20)
    out[:, 3::4] = Yp[:, 4::6] | (Crp[:, 2::3] << 10) | (Yp[:, 5::6] << 20)
    return out.reshape(-1).view(np.uint8).copy(), out.shape[1] * 4, Y, Cb, Cr


def v210_planes(plan, Y, Cb, Cr):
    """The three planes the reference's v210 unpack produces (Codec/convert.c:3968): channel 1 = Cr, channel 2 = Cb, rows below the picture
    zero; behind the last whole 48 pixels its scalar loop stores every group's first Cr twice (Cr0 Cr0 Cr1 instead of Cr0 Cr1 Cr2)."""
    h, w = Y.shape
    H = 2 * plan.band[(0, 0, 0)]["height"]
    cr = Cr.astype(np.int64).copy()
    t = (w - w % 48) // 2
    k = np.arange(t, w // 2)
    src = np.where((k - t) % 3 == 0, k, k - 1)           # local 0 -> 0, 1 -> 0, 2 -> 1 of the group
    cr[:, t:] = Cr[:, src]
    planes = []
    for pl in (Y, cr, Cb):
        p = np.zeros((H, pl.shape[1]), np.int16); p[:h] = pl
        planes.append(p)
    return planes


def byr4_planes(mosaic):
    """G, R-G, B-G, G1-G2 planes of a Bayer mosaic through the oracle's restatement of ConvertBYR4ToFrame16s (default log-90 curve)."""
    O = oracle()
    curve = np.zeros(1 << 14, np.uint16)
    O.orc_byr4_log90_curve.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    O.orc_byr4_log90_curve(12, 14, curve.ctypes.data_as(ctypes.c_void_p))
    O.orc_byr4_unpack_row.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 5
    ph, pw = mosaic.shape[0] // 2, mosaic.shape[1] // 2
    planes = [np.zeros((ph, pw), np.int16) for _ in range(4)]
    for r in range(ph):
        O.orc_byr4_unpack_row(mosaic[2 * r].ctypes.data_as(ctypes.c_void_p), mosaic[2 * r + 1].ctypes.data_as(ctypes.c_void_p), pw, 12, 14,
                              curve.ctypes.data_as(ctypes.c_void_p), *[p[r].ctypes.data_as(ctypes.c_void_p) for p in planes])
    return planes


def oracle_forward_planes(plan, planes):
    """Forward path of a 4:4:4(:4) frame with the oracle from its component planes (rows below the picture repeat the last row,
    frame.c:6020-6024), written into the product's pyramid layout."""
    O = oracle()
    coeffs = np.zeros(plan.coeff_elems, dtype=np.int16)
    for c, pl in enumerate(planes):
        H, w = 2 * plan.band[(c, 0, 0)]["height"], 2 * plan.band[(c, 0, 0)]["width"]     # component plane (half the mosaic for Bayer)
        src = np.zeros((H, w), np.int16); src[: pl.shape[0]] = pl; src[pl.shape[0]:] = pl[-1]
        for lv in (0, 1, 2):
            if lv:
                d = plan.band[(c, lv - 1, 0)]; v = plan.view(coeffs, c, lv - 1, 0)
                sp, sp_pitch, sw, sh = v.ctypes.data_as(c_i16p), d["pitch"], d["width"], d["height"]
            else:
                sp, sp_pitch, sw, sh = src.ctypes.data_as(c_i16p), w, w, H
            q = [plan.band[(c, lv, b)]["quant"] for b in range(4)]
            outs = [plan.view(coeffs, c, lv, b) for b in range(4)]
            bands = (c_i16p * 4)(*[o.ctypes.data_as(c_i16p) for o in outs])
            O.orc_fwd_spatial(sp, sp_pitch, sw, sh, plan.prescale[lv], iarr(q), plan.mpq, bands, outs[0].shape[1])
    return coeffs


def oracle_inverse_rgb48(plan, coeffs, b64a=False):
    """Whole inverse path with the oracle from a dequantized RGB 4:4:4 pyramid to packed RG48 words (display rows only);
    b64a: RGBA 4:4:4:4 pyramid to packed A,R,G,B words with the alpha expansion."""
    O = oracle()
    O.orc_inv_spatial_to_rgb48.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    O.orc_inv_spatial_to_b64a.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    work = coeffs.copy()
    nch = plan.num_channels
    for c in range(nch):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    d = plan.band[(0, 0, 0)]
    flat = [plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(nch) for b in range(4)]
    out = np.zeros((2 * d["height"], 2 * d["width"] * nch), np.uint16)
    if b64a:
        assert nch == 4
        O.orc_inv_spatial_to_b64a((c_i16p * 16)(*flat), d["pitch"], d["width"], d["height"], plan.precision, out.ctypes.data_as(ctypes.c_void_p), 2 * d["width"] * nch)
        return out
    O.orc_inv_spatial_to_rgb48((c_i16p * 16)(*(flat + [None] * (16 - len(flat)))), d["pitch"], d["width"], d["height"], plan.precision, nch,
                               out.ctypes.data_as(ctypes.c_void_p), 2 * d["width"] * nch)
    return out


def half_resolution_model(Y, V, U, uyvy=0):
    """Half-resolution picture of a 4:2:2 sample from its level-1 lowpass planes: SATURATE_8U(value >> 4), no dither
    (frame.c:11742 ConvertLowpass16s10bitToYUV, scalar loop)."""
    rows, w = Y.shape
    out = np.zeros((rows, 2 * w), np.uint8)
    yo, co = (1, 0) if uyvy else (0, 1)
    out[:, yo::2] = np.clip(Y.astype(np.int32) >> 4, 0, 255)
    out[:, co::4] = np.clip(U.astype(np.int32) >> 4, 0, 255)
    out[:, co + 2::4] = np.clip(V.astype(np.int32) >> 4, 0, 255)
    return out


def half16_equal(img, want, raw, nch):
    """Reference output == model; for b64a a row may keep its companded alpha (`raw`): the reference marks the alpha as expanded from a
    worker thread while others still convert rows (bayer.c:13871 / :16034)."""
    if nch == 3: return np.array_equal(img, want)
    if not all(np.array_equal(img[:, k::4], want[:, k::4]) for k in (1, 2, 3)): return False
    rows = (img[:, 0::4] == want[:, 0::4]).all(axis=1) | (img[:, 0::4] == raw[:, 0::4]).all(axis=1)
    return bool(rows.all())


def half_resolution_model16(planes, b64a=False, expand_alpha=True):
    """Half-resolution picture of an RGB 4:4:4 (RG48 out) or RGBA 4:4:4:4 (b64a out) sample from its level-1 lowpass planes G, R, B[, A]:
    value << 2 saturated to 16 bits (frame.c:7256 ConvertLowpassRGB444ToRGB48); b64a: words A, R, G, B, alpha expanded as at full resolution."""
    rows, w = planes[0].shape
    if b64a: v = [np.clip(p.astype(np.int64), 0, 16383) << 2 for p in planes]      # the planar-row route clamps to 14 bits first (bayer.c:12921-12934)
    else: v = [np.clip(p.astype(np.int64) << 2, 0, 65535) for p in planes]
    if b64a:
        a = np.clip((((v[3] >> 4) - 256) * 8 * 9400) >> 12, 0, 65535) if expand_alpha else v[3]
        order = [a, v[1], v[0], v[2]]
    else:
        order = [v[1], v[0], v[2]]
    out = np.zeros((rows, w * len(order)), np.uint16)
    for k, p in enumerate(order): out[:, k::len(order)] = p
    return out


def oracle_half_resolution16(plan, coeffs, b64a=False, expand_alpha=True):
    O = oracle()
    work = coeffs.copy()
    nch = plan.num_channels
    for c in range(nch):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    P = [plan.view(work, c, 0, 0)[: plan.height // 2, : plan.band[(c, 0, 0)]["width"]] for c in range(nch)]
    return half_resolution_model16(P, b64a, expand_alpha)


def oracle_half_resolution_rgba8(plan, coeffs):
    """Half-resolution picture of an RGBA 4:4:4:4 sample as BGRa bytes (top row first; BGRA is the same upside down): the level-1 lowpass planes of a pyramid with the
    lowpass bias 8 of the 8-bit RGB outputs, clamped to 14 bits; colour = >> 6, alpha = the 12-bit value (>> 2) through the alpha expansion of codec.h:164-165 -- the
    planar-row route of the full-resolution decode (orc_inv_spatial_to_rgba8) fed with the lowpass planes; no dither.  Pinned on the reference decoder."""
    O = oracle()
    work = with_lowpass_bias(plan, coeffs, 8)
    for c in range(4):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    rows = plan.height // 2
    G, R, B, A = [np.clip(plan.view(work, c, 0, 0)[:rows, : plan.band[(c, 0, 0)]["width"]].astype(np.int64), 0, 16383) for c in range(4)]
    out = np.zeros((rows, G.shape[1], 4), np.uint8)
    out[:, :, 0] = B >> 6; out[:, :, 1] = G >> 6; out[:, :, 2] = R >> 6
    a = np.maximum((A >> 2) - 256, 0)
    out[:, :, 3] = np.clip((((a << 3) * 9400) >> 16) >> 4, 0, 255)
    return out.reshape(rows, -1)


def oracle_half_resolution_yu64(plan, coeffs):
    """Half-resolution picture of a 4:2:2 sample as YU64 (frame.c:11146 ConvertLowpass16sToYUV64, 10-bit branch): the level-1 lowpass planes clamped to [0, 4095], << 4,
    words Y0 C1 Y1 C2.  coeffs: decoded with the YU64 lowpass bias (Plan(..., pixkind=PIXKIND["YU64"]))."""
    O = oracle()
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    rows = plan.height // 2
    Y, C1, C2 = [np.clip(plan.view(work, c, 0, 0)[:rows, : plan.band[(c, 0, 0)]["width"]].astype(np.int64), 0, 4095) << 4 for c in range(3)]
    out = np.zeros((rows, 2 * Y.shape[1]), np.uint16)
    out[:, 0::2] = Y; out[:, 1::4] = C1; out[:, 3::4] = C2
    return out


def oracle_half_resolution_v210(plan, coeffs):
    """Half-resolution picture of a 4:2:2 sample as v210 (frame.c:12139 ConvertLowpass16s10bitToV210): the level-1 lowpass planes >> 2 clamped to 10 bits -- the words of
    oracle_half_resolution_yu64 >> 6 --, groups of six pixels in four words (Cb Y Cr | Y Cb Y | Cr Y Cb | Y Cr Y, low bits first), Cb = plane 2, Cr = plane 1.  coeffs: decoded
    with the lowpass bias 4 of the 10-bit 4:2:2 outputs (Plan(..., pixkind=PIXKIND["v210"]) or ["YU64"])."""
    yu = oracle_half_resolution_yu64(plan, coeffs).astype(np.uint32) >> 6
    rows, n = yu.shape[0], yu.shape[1] // 2                # n pixels per row
    g = n // 6
    Y = yu[:, 0::2][:, : 6 * g].reshape(rows, g, 6); Cr = yu[:, 1::4][:, : 3 * g].reshape(rows, g, 3); Cb = yu[:, 3::4][:, : 3 * g].reshape(rows, g, 3)
    out = np.zeros((rows, g, 4), np.uint32)
    out[:, :, 0] = Cb[:, :, 0] | (Y[:, :, 0] << 10) | (Cr[:, :, 0] << 20)
    out[:, :, 1] = Y[:, :, 1] | (Cb[:, :, 1] << 10) | (Y[:, :, 2] << 20)
    out[:, :, 2] = Cr[:, :, 1] | (Y[:, :, 3] << 10) | (Cb[:, :, 2] << 20)
    out[:, :, 3] = Y[:, :, 4] | (Cr[:, :, 2] << 10) | (Y[:, :, 5] << 20)
    return out.reshape(rows, 4 * g)


def oracle_half_resolution_rgb24_of_yuv422(plan, coeffs, color_space=2):
    """Half-resolution picture of a 4:2:2 sample as RG24 (bottom row first), restated from the scalar loop of frame.c:9153 (ConvertLowpass16sToRGBNoIPPFast; its vector code is
    compiled out): lowpass planes >> 4, Y = ((Y - y_offset) * ymult) >> 7, R = (Y + r_vmult V) >> 7, G = (2 Y - g_umult U - g_vmult V) >> 8, B = (Y + 2 b_umult U) >> 7, no dither.
    coeffs: decoded with the lowpass bias of RG24 output (Plan(..., pixkind=PIXKIND["RG24"])).  color_space: 2 = 709 (the default), 1 = 601 (computer-systems range both)."""
    O = oracle()
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    rows = plan.height // 2
    Yp, C1, C2 = [plan.view(work, c, 0, 0)[:rows, : plan.band[(c, 0, 0)]["width"]].astype(np.int64) for c in range(3)]
    yo, ym, rv, gv, gu, bu = (16, 128 * 149, 204, 208, 100, 129) if color_space == 1 else (16, 128 * 149, 230, 137, 55, 135)
    Y = (((Yp >> 4) - yo) * ym) >> 7
    V = np.repeat(C1 >> 4, 2, axis=1) - 128; U = np.repeat(C2 >> 4, 2, axis=1) - 128
    out = np.zeros((rows, Y.shape[1], 3), np.uint8)
    out[:, :, 2] = np.clip((Y + rv * V) >> 7, 0, 255); out[:, :, 1] = np.clip((2 * Y - gu * U - gv * V) >> 8, 0, 255); out[:, :, 0] = np.clip((Y + 2 * bu * U) >> 7, 0, 255)
    return out[::-1].reshape(rows, -1)


def _half_lowpass_planes_of_yuv422(plan, coeffs):
    O = oracle()
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    rows = plan.height // 2
    return [plan.view(work, c, 0, 0)[:rows, : plan.band[(c, 0, 0)]["width"]].astype(np.int64) for c in range(3)]


def oracle_half_resolution_rgb32_of_yuv422(plan, coeffs, bottom_up, color_space=2):
    """Half-resolution picture of a 4:2:2 sample as BGRA (bottom row first) / BGRa, restated from the SSE2 loop of the RGB32 branch of frame.c:8504
    ConvertLowpass16sToRGBNoIPPFast (:9270-9478; half widths that are multiples of 16: the loop serves every column): lowpass planes >> 4 packed to unsigned bytes,
    (Y - 16) << 7 mulhi 128 * 149 << 1, chroma products in wrapping 16-bit arithmetic shifted to six fraction bits, saturating sums, >> 6 WITHOUT the rounding term of
    the full-resolution routine, packus; bytes B, G, R, 255.  coeffs: decoded with the lowpass bias of the output format (Plan(..., pixkind=PIXKIND["BGRA"] / ["BGRa"], enc=ENC["422"]))."""
    Yp, C1, C2 = _half_lowpass_planes_of_yuv422(plan, coeffs)
    ym, rv, gv, gu, bu = (128 * 149, 204, 208, 100, 129) if color_space == 1 else (128 * 149, 230, 137, 55, 135)
    s16 = lambda x: ((x + 0x8000) % 0x10000) - 0x8000
    sat = lambda x: np.clip(x, -32768, 32767)
    Y = np.clip(Yp >> 4, 0, 255) - 16
    V = np.repeat(np.clip(C1 >> 4, 0, 255), 2, axis=1) - 128; U = np.repeat(np.clip(C2 >> 4, 0, 255), 2, axis=1) - 128
    Y = s16(Y << 7); Y = (Y * ym) >> 16; Y = s16(Y << 1)
    R = sat(Y + (s16(V * rv) >> 1)) >> 6
    G = sat(sat(Y - (s16(V * gv) >> 2)) - (s16(U * gu) >> 2)) >> 6
    B = sat(Y + s16(U * bu)) >> 6
    out = np.zeros((Y.shape[0], Y.shape[1], 4), np.uint8)
    out[:, :, 0] = np.clip(B, 0, 255); out[:, :, 1] = np.clip(G, 0, 255); out[:, :, 2] = np.clip(R, 0, 255); out[:, :, 3] = 255
    return (out[::-1] if bottom_up else out).reshape(Y.shape[0], -1)


def oracle_half_resolution_rgb16_of_yuv422(plan, coeffs, b64a, color_space=2):
    """Half-resolution picture of a 4:2:2 sample as RG48 words (R, G, B) / b64a words (0xffff, R, G, B), restated from frame.c:9567 ConvertLowpass16sYUVtoRGB48 (a scalar
    loop): the lowpass planes read as UNSIGNED 16-bit words << 4, Y = ((Y - 16 * 256) * ymult) >> 7, R = (Y + r_vmult V) >> 7, G = (2 Y - g_umult U - g_vmult V) >> 8,
    B = (Y + 2 b_umult U) >> 7 with U, V - 32768, saturated to 16 bits.  coeffs: decoded with the bias of the output format (Plan(..., pixkind=PIXKIND["RG48"] / ["b64a"], enc=ENC["422"]))."""
    Yp, C1, C2 = _half_lowpass_planes_of_yuv422(plan, coeffs)
    ym, rv, gv, gu, bu = (128 * 149, 204, 208, 100, 129) if color_space == 1 else (128 * 149, 230, 137, 55, 135)
    u16 = lambda x: x & 0xffff
    i32 = lambda x: ((x + 2 ** 31) % 2 ** 32) - 2 ** 31              # (the reference computes in int: wraps where a lowpass word is far out of range)
    Y = i32(i32((u16(Yp) << 4) - (16 << 8)) * ym) >> 7
    V = np.repeat(u16(C1) << 4, 2, axis=1) - 32768; U = np.repeat(u16(C2) << 4, 2, axis=1) - 32768
    R = i32(Y + rv * V) >> 7; G = i32(2 * Y - gu * U - gv * V) >> 8; B = i32(Y + 2 * bu * U) >> 7
    nw = 4 if b64a else 3
    out = np.zeros((Y.shape[0], Y.shape[1], nw), np.uint16)
    if b64a: out[:, :, 0] = 0xffff
    out[:, :, nw - 3] = np.clip(R, 0, 65535); out[:, :, nw - 2] = np.clip(G, 0, 65535); out[:, :, nw - 1] = np.clip(B, 0, 65535)
    return out.reshape(Y.shape[0], -1)


def oracle_half_resolution_rgb(plan, coeffs, name, r=0):
    """Half-resolution picture of an RGB 4:4:4 sample in the 8-bit (RG24 / BGRA / BGRa), 10-bit (r210 / DPX0 / AB10 / AR10) and b64a output formats, restated from
    frame.c:7150 ConvertLowpassRGB444ToRGB: the level-1 lowpass planes G, R, B of a pyramid that carries the lowpass bias of the output format (decoder.c:12290-12312:
    8 for 8-bit RGB, 6 for 10-bit RGB: with_lowpass_bias), then 8 bit: (v + 9 + r) clamped to 14 bits >> 6 with r = rand() & 31 per pixel
    (convert.c:6151, shift 6; the caller passes r = 0 / 31 for the two ends), RG24 / BGRA bottom row first; 10 bit: (v << 2) saturated >> 6 (frame.c:7662);
    b64a: (v << 2) saturated, alpha 65535 (frame.c:7494).  Returns rows of bytes / 32-bit words / 16-bit words."""
    O = oracle()
    work = with_lowpass_bias(plan, coeffs, 8 if name in ("RG24", "BGRA", "BGRa") else (0 if name == "b64a" else 6))
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    rows = plan.height // 2
    G, R, B = [plan.view(work, c, 0, 0)[:rows, : plan.band[(c, 0, 0)]["width"]].astype(np.int64) for c in range(3)]
    if name in ("RG24", "BGRA", "BGRa"):
        bpp = 3 if name == "RG24" else 4
        out = np.full((rows, G.shape[1], bpp), 255, np.uint8)
        for byte, pl in ((0, B), (1, G), (2, R)):
            out[:, :, byte] = np.clip(pl + 9 + r, 0, 16383) >> 6
        return out.reshape(rows, -1)
    if name == "b64a":
        out = np.zeros((rows, G.shape[1], 4), np.uint16)
        out[:, :, 0] = 65535
        for word, pl in ((1, R), (2, G), (3, B)): out[:, :, word] = np.clip(pl << 2, 0, 65535)
        return out.reshape(rows, -1)
    shifts = {"r210": (20, 10, 0), "DPX0": (22, 12, 2), "AB10": (0, 10, 20), "AR10": (20, 10, 0), "RG30": (0, 10, 20)}[name]
    words = sum((np.clip(pl << 2, 0, 65535) >> 6) << sh for sh, pl in zip(shifts, (R, G, B))).astype(np.uint32)
    return words.byteswap() if name in ("r210", "DPX0") else words


def oracle_half_resolution(plan, coeffs, uyvy=0):
    """Levels 3 -> 2 -> 1 with the oracle, then the half-resolution model; rows = display height / 2."""
    O = oracle()
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    P = [plan.view(work, c, 0, 0)[:, : plan.band[(c, 0, 0)]["width"]] for c in range(3)]
    return half_resolution_model(P[0], P[1], P[2], uyvy)[: plan.height // 2]


def field_flicker_frame(w, h):
    """Interlaced torture picture: the two fields differ by nearly the full range and swap sign at vertical edges, so the
    difference-coded HL1 band holds quantized steps beyond +-250 (peak values)."""
    f = np.zeros((h, w * 2), np.uint8)
    f[:, 1::2] = 128
    x = np.arange(w)
    band = (x // 37) % 2
    f[0::2, 0::2] = np.where(band, 255, 0)
    f[1::2, 0::2] = np.where(band, 0, 255)
    return f.reshape(-1).copy(), w * 2


def oracle_forward_interlaced_yuv422(plan, frame, pitch, uyvy=0):
    """Forward path of one interlaced 4:2:2 frame with the oracle: "frame" wavelet at level 1 (temporal pair + horizontal 2/6,
    difference-coded HL band), spatial wavelets at levels 2 and 3; product pyramid layout."""
    O = oracle()
    coeffs = np.zeros(plan.coeff_elems, dtype=np.int16)
    frame, H = pad_yuv422_rows(plan, frame, pitch)
    for c in range(3):
        cw = plan.width if c == 0 else plan.width // 2
        q = [plan.band[(c, 0, b)]["quant"] for b in range(4)]
        outs = [plan.view(coeffs, c, 0, b) for b in range(4)]
        bands = (c_i16p * 4)(*[o.ctypes.data_as(c_i16p) for o in outs])
        O.orc_fwd_frame_yuv422(p8(frame), pitch, cw, H, c, plan.precision - 8, uyvy, iarr(q), plan.mpq, bands, outs[0].shape[1])
        for lv in (1, 2):
            src = plan.view(coeffs, c, lv - 1, 0)
            d = plan.band[(c, lv - 1, 0)]
            q = [plan.band[(c, lv, b)]["quant"] for b in range(4)]
            outs = [plan.view(coeffs, c, lv, b) for b in range(4)]
            bands = (c_i16p * 4)(*[o.ctypes.data_as(c_i16p) for o in outs])
            O.orc_fwd_spatial(src.ctypes.data_as(c_i16p), d["pitch"], d["width"], d["height"], plan.prescale[lv], iarr(q), plan.mpq, bands, outs[0].shape[1])
    return coeffs


def product_write_sample_host(plan, coeffs, frame_number, meta_global=b"", meta_local=b"", input_format=COLOR_FORMAT_YUYV, color_space=2, progressive=1):
    out = np.zeros(plan.width * plan.height * 4 + 65536, dtype=np.uint8)
    mg = np.frombuffer(meta_global, dtype=np.uint8).copy() if meta_global else np.zeros(4, np.uint8)
    ml = np.frombuffer(meta_local, dtype=np.uint8).copy() if meta_local else np.zeros(4, np.uint8)
    n = hooks().cfhd_amd_write_sample_host(plan.width, plan.height, plan.pixkind, plan.enc, plan.quality, progressive, input_format, color_space,
                                             frame_number, p16(coeffs), p8(mg), len(meta_global), p8(ml), len(meta_local), p8(out), out.size)
    assert n > 0
    return bytes(out[:n])


def first_metadata_chunk(sample):
    """(offset, size) of the payload of the first CODEC_TAG_METADATA chunk in a sample."""
    import struct
    pos = 0
    while pos + 4 <= len(sample):
        tag, val = struct.unpack(">hH", sample[pos:pos + 4])
        t = -tag if tag < 0 else tag
        pos += 4
        if t == 2:
            pos += 4 * val
        elif t == 0x4002:
            return pos, val * 4
        elif t & 0x4000:
            pos += val * 4
    return None


# ------------------------------------------------------------------------------------------
# CPU emulation of the HIP kernels (tests/hipemu): same kernel source, executed workgroup by workgroup
# ------------------------------------------------------------------------------------------
EMU_SO = os.path.join(ROOT, "tests", "_build", "libcfhd_emu.so")
_emu = None


def emu():
    global _emu
    if _emu is None:
        src = os.path.join(ROOT, "tests", "hipemu", "emu_kernels.cpp")
        deps = [src, os.path.join(ROOT, "tests", "hipemu", "hip_emu.h")] + [
            os.path.join(PRODUCT_DIR, "csrc", f) for f in os.listdir(os.path.join(PRODUCT_DIR, "csrc")) if f.endswith((".h", ".cpp"))]
        if not os.path.exists(EMU_SO) or any(os.path.getmtime(d) > os.path.getmtime(EMU_SO) for d in deps):
            os.makedirs(os.path.dirname(EMU_SO), exist_ok=True)
            csrc = os.path.join(PRODUCT_DIR, "csrc")
            host = [os.path.join(csrc, f) for f in ("cfhd_tables.cpp", "cfhd_bitstream.cpp", "cfhd_gop.cpp")]      # host-only product sources the entropy emulation needs
            _build_once(EMU_SO, ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.dirname(src), "-I" + csrc, src] + host, deps)
        _emu = ctypes.CDLL(EMU_SO)
    return _emu


# ------------------------------------------------------------------------------------------
# product C ABI (CFHD_* entry points of libcfhd_amd.so) -- needs a GPU
# ------------------------------------------------------------------------------------------
def amd_encode_frames(frames, pitch, width, height, pixfmt=PIX_YUY2, encoded=ENCODED_YUV422, quality=QUALITY_FILMSCAN1, flags=0):
    L = product()
    enc = ctypes.c_void_p()
    assert L.CFHD_OpenEncoder(ctypes.byref(enc), None) == 0
    rc = L.CFHD_PrepareToEncode(enc, width, height, pixfmt, encoded, flags, quality)
    assert rc == 0, "CFHD_PrepareToEncode -> %d" % rc
    out = []
    for f in frames:
        rc = L.CFHD_EncodeSample(enc, f.ctypes.data_as(ctypes.c_void_p), pitch)
        assert rc == 0, "CFHD_EncodeSample -> %d (%s)" % (rc, amd_last_error())
        p = ctypes.c_void_p(); n = ctypes.c_size_t()
        assert L.CFHD_GetSampleData(enc, ctypes.byref(p), ctypes.byref(n)) == 0
        out.append(ctypes.string_at(p, n.value))
    L.CFHD_CloseEncoder(enc)
    return out


def amd_last_error():
    L = product()
    L.cfhd_amd_last_error.restype = ctypes.c_char_p
    return (L.cfhd_amd_last_error() or b"").decode()


def amd_decode_sample(sample, pixfmt=PIX_YUY2, pitch=None, decoder=None, resolution=1):
    L = product()
    dec = decoder or ctypes.c_void_p()
    if decoder is None:
        assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
    aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
    sb = ctypes.create_string_buffer(sample, len(sample))
    rc = L.CFHD_PrepareToDecode(dec, 0, 0, pixfmt, resolution, 0, sb, 512, ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af))
    assert rc == 0, "CFHD_PrepareToDecode -> %d" % rc
    p = ctypes.c_int32()
    assert L.CFHD_GetImagePitch(aw.value, af.value, ctypes.byref(p)) == 0
    pitch = pitch or p.value
    out = np.zeros(pitch * ah.value, dtype=np.uint8)
    rc = L.CFHD_DecodeSample(dec, sb, len(sample), out.ctypes.data_as(ctypes.c_void_p), pitch)
    assert rc == 0, "CFHD_DecodeSample -> %d (%s)" % (rc, amd_last_error())
    if decoder is None:
        L.CFHD_CloseDecoder(dec)
    return out, pitch, aw.value, ah.value


def host_decode_pyramid(sample, plan, lowpass_offset=1):
    """Dequantized coefficient pyramid of a sample via the product's host parser + VLC decoder (CPU only).
    lowpass_offset=1 applies the bias the reference decoder adds to the lowpass band (Codec/decoder.c:12240-12290,
    :12468-12545: +24 / +5 for even / odd lowpass widths of 10-bit intra frames); pass 0 for the raw coefficients."""
    out = np.zeros(plan.coeff_elems, dtype=np.int16)
    info = (ctypes.c_int * 8)()
    s = np.frombuffer(sample, dtype=np.uint8).copy()
    rc = hooks().cfhd_amd_decode_bands_host(p8(s), len(sample), plan.pixkind, p16(out), out.size, info, lowpass_offset)
    assert rc == 0, rc
    return out


def oracle_lowpass_bias(precision, lowpass_width, out_pixkind, channel):
    """What the reference decoder adds to every word of the raw lowpass band before the inverse transform (Codec/decoder.c:12240-12312 even widths, :12468-12545 the
    bit-serial path of odd widths): 8-bit sources 32; 10-bit sources 24 (odd width: 5) for the 8-bit outputs, 4 (5) for the deep 4:2:2 outputs YU64 / v210, and on the
    odd-width path 8 (luma) / 4 (chroma) less for the bottom-up 8-bit RGB outputs (:12500-12508); 12-bit sources 8 for the 8-bit RGB outputs, 6 for the 10-bit RGB
    words, nothing for the 16-bit outputs and Bayer (:12290-12316)."""
    even = (lowpass_width & 1) == 0
    K = PIXKIND
    if precision == 8: return 32
    if precision == 10:
        if out_pixkind in (K["YU64"], K["v210"]): return 4 if even else 5
        if not even and out_pixkind in (K["RG24"], K["BGRA"]): return 5 - 8 if channel == 0 else 5 - 4
        return 24 if even else 5
    if precision == 12:
        if out_pixkind in (K["RG24"], K["BGRA"], K["BGRa"]): return 8
        if K["r210"] <= out_pixkind <= K["AR10"]: return 6
    return 0


def oracle_decode_pyramid(sample, plan, lowpass_offset=1, out_pixkind=None):
    """Dequantized coefficient pyramid of an intra-frame sample, in the product's pyramid layout, by the ORACLE alone (oracle/cfhd_oracle_ent.c orc_decode_sample: its own
    tag-value walk and bit-serial decoder of both code sets, peak tables and difference coding -- nothing of the product's parser or VLC decoder, which the GPU entropy
    kernels share tables and job builders with).  This is what the decode gates of the GPU tests, bench.py's parity check and smoke() feed the oracle's inverse transform
    with.  lowpass_offset=1 adds the reference decoder's lowpass bias for the output format (oracle_lowpass_bias; out_pixkind defaults to the plan's pixel kind)."""
    O = oracle()
    out = np.zeros(plan.coeff_elems, dtype=np.int16)
    P16 = ctypes.POINTER(ctypes.c_int16)
    dst = (P16 * 4 * 3 * 4)(); pitch = (ctypes.c_int * 4 * 3 * 4)(); dims = (ctypes.c_int * 2 * 4 * 3 * 4)()
    for (c, lv, b), d in plan.band.items():
        if b == 0 and lv != 2: continue                                  # (the level-1 / level-2 lowpass planes are intermediates of the transform, not coded)
        v = plan.view(out, c, lv, b)
        dst[c][lv][b] = v.ctypes.data_as(P16); pitch[c][lv][b] = d["pitch"]; dims[c][lv][b][0] = d["width"]; dims[c][lv][b][1] = d["height"]
    info = (ctypes.c_int32 * 8)()
    s = np.frombuffer(sample, dtype=np.uint8).copy()
    O.orc_decode_sample.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = O.orc_decode_sample(p8(s), len(sample), ctypes.byref(dst), ctypes.byref(pitch), ctypes.byref(dims), ctypes.byref(info))
    assert rc == 0, "oracle sample walk failed: %d" % rc
    assert info[3] == plan.num_channels and info[7] == 10 * plan.num_channels, "sample has %d channels, %d bands decoded" % (info[3], info[7])
    if lowpass_offset:
        kind = plan.pixkind if out_pixkind is None else out_pixkind
        for c in range(plan.num_channels):
            d = plan.band[(c, 2, 0)]
            ll = plan.view(out, c, 2, 0)[:, : d["width"]]
            bias = oracle_lowpass_bias(info[4] or 8, d["width"], kind, c)
            ll[:] = np.minimum(ll.astype(np.int32) + bias, 0x7fff).astype(np.int16)
    return out


def oracle_inverse_yuv422(plan, coeffs, dither, uyvy=0):
    """Whole inverse path with the oracle from a dequantized pyramid (product layout) to packed 4:2:2."""
    O = oracle()
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    ptrs = (c_i16p * 12)(*[plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(3) for b in range(4)])
    pitches = [plan.band[(c, 0, 0)]["pitch"] for c in range(3)]
    w = plan.band[(0, 0, 0)]["width"]; h = plan.band[(0, 0, 0)]["height"]
    out = np.zeros((2 * h, 4 * w), np.uint8)
    O.orc_inv_spatial_to_yuv422(ptrs, iarr(pitches), w, h, plan.precision, uyvy, dither, p8(out), 4 * w)
    return out


def oracle_rgb16_to_yuv422_planes(words, words_per_pixel, r_word, w, h, color_space=0):
    """Deep RGB pixels (uint16 array h x w*words_per_pixel, r at word r_word of every pixel, then g, b) -> the three 10-bit planes Y,
    channel 1, channel 2 of a 4:2:2 frame with the oracle (Codec/frame.c:6731 ConvertAnyDeep444to422)."""
    O = oracle()
    O.orc_rgb16_to_yuv422.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    src = np.ascontiguousarray(words[:, r_word:])
    flat = np.zeros(h * w * words_per_pixel + 8, np.uint16); flat[: words.size - r_word] = words.reshape(-1)[r_word:]
    Y = np.zeros((h, w), np.int16); C1 = np.zeros((h, w // 2), np.int16); C2 = np.zeros((h, w // 2), np.int16)
    O.orc_rgb16_to_yuv422(flat.ctypes.data_as(ctypes.c_void_p), w * words_per_pixel, words_per_pixel, w, h, h, color_space, p16(Y), w, p16(C1), p16(C2), w // 2)
    return [Y, C1, C2]


def with_lowpass_bias(plan, coeffs, want):
    """A copy of a pyramid from host_decode_pyramid(sample, plan) whose lowpass bands carry the bias `want` of the output format in question (decoder.c:12290-12312:
    8 for the 8-bit RGB outputs of 12-bit samples, 6 for the 10-bit RGB words, 0 for the 16-bit ones) whatever output kind the plan was made for."""
    applied = 0
    if plan.precision == 12:
        applied = 8 if plan.pixkind in (PIXKIND["RG24"], PIXKIND["BGRA"], PIXKIND["BGRa"]) else (6 if plan.pixkind in (PIXKIND["r210"], PIXKIND["DPX0"], PIXKIND["AB10"], PIXKIND["AR10"]) else 0)
    out = coeffs.copy()
    if want != applied:
        for c in range(plan.num_channels): plan.view(out, c, 2, 0)[:] += want - applied
    return out


def oracle_inverse_rgb10(plan, coeffs, name):
    """Whole inverse path with the oracle from a dequantized RGB 4:4:4 pyramid to the 32-bit words of r210 / DPX0 / AB10 / AR10 (as they lie in memory)."""
    O = oracle()
    work = with_lowpass_bias(plan, coeffs, 6)
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    d = plan.band[(0, 0, 0)]
    flat = [plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(3) for b in range(4)]
    order, shifts, code = RGB10_FORMATS[name]
    out = np.zeros((plan.height, 2 * d["width"]), np.uint32)
    O.orc_inv_spatial_to_rgb10.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_int]
    O.orc_inv_spatial_to_rgb10((c_i16p * 16)(*(flat + [None] * 4)), d["pitch"], d["width"], d["height"], plan.height, shifts[0], shifts[1], shifts[2], int(order == ">"),
                               out.ctypes.data_as(ctypes.c_void_p), out.shape[1])
    return out


def oracle_inverse_rgb8(plan, coeffs, bytes_per_pixel, bottom_up, r):
    """Whole inverse path with the oracle from a dequantized RGB 4:4:4 pyramid to 8-bit B, G, R(, A) pixels with the dither value r (0..127)."""
    O = oracle()
    work = with_lowpass_bias(plan, coeffs, 8)
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    d = plan.band[(0, 0, 0)]
    flat = [plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(3) for b in range(4)]
    out = np.zeros((plan.height, 2 * d["width"] * bytes_per_pixel), np.uint8)
    O.orc_inv_spatial_to_rgb8.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_int]
    O.orc_inv_spatial_to_rgb8((c_i16p * 16)(*(flat + [None] * 4)), d["pitch"], d["width"], d["height"], plan.precision, plan.height, bytes_per_pixel, int(bottom_up), r,
                              out.ctypes.data_as(ctypes.c_void_p), out.shape[1])
    return out


def oracle_inverse_yu64(plan, coeffs):
    """Whole inverse path with the oracle from a dequantized 4:2:2 pyramid to YU64 words (Y0 C1 Y1 C2, 16 bits each; no dither)."""
    O = oracle()
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    ptrs = (c_i16p * 12)(*[plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(3) for b in range(4)])
    pitches = [plan.band[(c, 0, 0)]["pitch"] for c in range(3)]
    w = plan.band[(0, 0, 0)]["width"]; h = plan.band[(0, 0, 0)]["height"]
    out = np.zeros((2 * h, 4 * w), np.uint16)
    O.orc_inv_spatial_to_yu64.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    O.orc_inv_spatial_to_yu64(ptrs, iarr(pitches), w, h, plan.precision, out.ctypes.data_as(ctypes.c_void_p), 4 * w)
    return out


def oracle_inverse_v210(plan, coeffs, width):
    """Whole inverse path with the oracle from a dequantized 4:2:2 pyramid (Plan(..., pixkind=PIXKIND["YU64"]): lowpass bias 4) to v210 words: the YU64 words >> 6,
    three to a 32-bit word, Cb from channel 2, Cr from channel 1 (orc_inv_spatial_to_v210, pinned on the reference by test_reference_v210_decode_equals_oracle)."""
    O = oracle()
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    ptrs = (c_i16p * 12)(*[plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(3) for b in range(4)])
    pitches = [plan.band[(c, 0, 0)]["pitch"] for c in range(3)]
    bw = plan.band[(0, 0, 0)]["width"]; bh = plan.band[(0, 0, 0)]["height"]
    nwords = (width // 6) * 4
    out = np.zeros((2 * bh, nwords), np.uint32)
    O.orc_inv_spatial_to_v210.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    O.orc_inv_spatial_to_v210(ptrs, iarr(pitches), bw, bh, plan.precision, out.ctypes.data_as(ctypes.c_void_p), nwords)
    return out


def oracle_inverse_interlaced_yuv422(plan, coeffs, dither, uyvy=0):
    """Inverse path of an interlaced 4:2:2 sample with the oracle: spatial levels 3 and 2, then the inverse "frame" transform
    (horizontal synthesis of the temporal low / high rows, temporal pair, 10 -> 8 bits)."""
    O = oracle()
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    ptrs = (c_i16p * 12)(*[plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(3) for b in range(4)])
    pitches = [plan.band[(c, 0, 0)]["pitch"] for c in range(3)]
    w = plan.band[(0, 0, 0)]["width"]; h = plan.band[(0, 0, 0)]["height"]
    out = np.zeros((2 * h, 4 * w), np.uint8)
    O.orc_inv_frame_to_yuv422(ptrs, iarr(pitches), w, h, plan.precision, uyvy, dither, p8(out), 4 * w)
    return out


def psnr_yuy2(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = np.mean(d * d)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 * 255.0 / mse)


# ------------------------------------------------------------------------------------------
# Two-frame group (CFHD_ENCODING_FLAGS_YUV_2FRAME_GOP; cineform-sdk_amd/csrc/cfhd_gop.h)
# ------------------------------------------------------------------------------------------
ENCODING_FLAGS_2FRAME_GOP = 2       # Common/CFHDTypes.h:254


class GopPlan:
    """Python view of cfhd::GopPlan (six wavelets per channel) as the product derives it."""

    def __init__(self, width, height, pixkind=1, quality=QUALITY_FILMSCAN1, interlaced=0):
        L = hooks()
        L.cfhd_amd_gop_plan_info.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_longlong)]
        buf = (ctypes.c_longlong * 512)()
        self.interlaced = int(bool(interlaced))            # CFHD_ENCODING_FLAGS_YUV_INTERLACED on top of the group flag: frame transform at level 1 of both frames
        n = L.cfhd_amd_gop_plan_info(width, height, pixkind | (self.interlaced << 8), quality, buf)
        assert n > 0, "gop_plan_info failed"
        v = list(buf[:n])
        self.coeff_elems, self.enc_height, self.mpq = v[0:3]
        self.width, self.height, self.pixkind, self.quality = width, height, pixkind, quality
        self.w = {}
        i = 3
        for c in range(3):
            for k in range(6):
                d = dict(zip(("type", "level", "nbands", "width", "height", "pitch", "prescale"), v[i:i + 7])); i += 7
                d["offset"] = []; d["quant"] = []; d["scale"] = []
                for b in range(4):
                    d["offset"].append(v[i]); d["quant"].append(v[i + 1]); d["scale"].append(v[i + 2]); i += 3
                self.w[(c, k)] = d

    def view(self, coeffs, c, k, b):
        d = self.w[(c, k)]
        o = d["offset"][b]
        return coeffs[o: o + d["pitch"] * d["height"]].reshape(d["height"], d["pitch"])


def oracle_forward_gop(gp, frame0, frame1, pitch, uyvy=0):
    """The group transform with the oracle (oracle/cfhd_oracle_fwd.c for the spatial steps; the temporal step is a saturating sum /
    difference, Codec/temporal.c:498), written into the product's group pyramid layout."""
    O = oracle()
    coeffs = np.zeros(gp.coeff_elems, dtype=np.int16)
    H = gp.enc_height
    for f, frame in enumerate((frame0, frame1)):
        if H != gp.height:
            padded = np.full(H * pitch, 0x80, dtype=np.uint8); padded[: gp.height * pitch] = np.asarray(frame).reshape(-1)[: gp.height * pitch]; frame = padded
        for c in range(3):
            d = gp.w[(c, f)]
            outs = [gp.view(coeffs, c, f, b) for b in range(4)]
            bands = (c_i16p * 4)(*[o.ctypes.data_as(c_i16p) for o in outs])
            cw = gp.width if c == 0 else gp.width // 2
            # level 1: the spatial transform, or -- interlaced groups -- the frame transform of interlaced intra frames (Codec/encoder.c:2950-2979)
            (O.orc_fwd_frame_yuv422 if gp.interlaced else O.orc_fwd_spatial_yuv422)(p8(np.ascontiguousarray(frame)), pitch, cw, H, c, 2, uyvy, iarr(d["quant"]), gp.mpq, bands, d["pitch"])
    def spatial(c, src, dst):
        d = gp.w[(c, dst)]
        outs = [gp.view(coeffs, c, dst, b) for b in range(4)]
        bands = (c_i16p * 4)(*[o.ctypes.data_as(c_i16p) for o in outs])
        srcc = np.ascontiguousarray(src)
        O.orc_fwd_spatial(srcc.ctypes.data_as(c_i16p), srcc.shape[1], 2 * d["width"], 2 * d["height"], d["prescale"], iarr(d["quant"]), gp.mpq, bands, d["pitch"])
    for c in range(3):
        a = gp.view(coeffs, c, 0, 0).astype(np.int32); b = gp.view(coeffs, c, 1, 0).astype(np.int32)
        gp.view(coeffs, c, 2, 0)[:] = np.clip(a + b, -32768, 32767).astype(np.int16)
        gp.view(coeffs, c, 2, 1)[:] = np.clip(b - a, -32768, 32767).astype(np.int16)
        # (pad columns: zero in both inputs, zero in both outputs)
        spatial(c, gp.view(coeffs, c, 2, 1), 3)
        spatial(c, gp.view(coeffs, c, 2, 0), 4)
        spatial(c, gp.view(coeffs, c, 4, 0), 5)
    return coeffs


def product_write_gop_host(gp, kind, coeffs=None, frame_number=1, meta_global=b""):
    """kind 0: group sample from a group pyramid; 1: sequence header; 2: P-frame sample -- the product's host writer."""
    L = hooks()
    L.cfhd_amd_write_gop_host.restype = ctypes.c_size_t
    L.cfhd_amd_write_gop_host.argtypes = [ctypes.c_int] * 5 + [ctypes.c_uint, c_i16p, c_u8p, ctypes.c_size_t, c_u8p, ctypes.c_size_t]
    out = np.zeros(gp.width * gp.enc_height * 4 + 65536, dtype=np.uint8)
    mg = np.frombuffer(meta_global, dtype=np.uint8).copy() if meta_global else np.zeros(4, np.uint8)
    cf = coeffs if coeffs is not None else np.zeros(8, np.int16)
    n = L.cfhd_amd_write_gop_host(kind, gp.width, gp.height, gp.pixkind | (gp.interlaced << 8), gp.quality, frame_number, p16(cf), p8(mg), len(meta_global), p8(out), out.size)
    assert n > 0
    return out[:n].tobytes()


def host_decode_group(sample, gp):
    """Dequantized group pyramid of a group sample via the product's host parser + VLC decoder (CPU only)."""
    L = hooks()
    L.cfhd_amd_decode_group_host.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_int, c_i16p, ctypes.c_size_t]
    out = np.zeros(gp.coeff_elems, dtype=np.int16)
    sb = np.frombuffer(sample, dtype=np.uint8).copy()
    rc = L.cfhd_amd_decode_group_host(p8(sb), len(sample), gp.pixkind, p16(out), out.size)
    assert rc == 0, "decode_group_host -> %d" % rc
    return out


def oracle_decode_group(sample, gp, lowpass_offset=1):
    """Dequantized group pyramid of a group sample, in the product's group pyramid layout, by the ORACLE alone (oracle/cfhd_oracle_ent.c orc_decode_group: its own tag-value
    walk, bit-serial decoder of both code sets with peak tables and difference coding, raw 16-bit bands) -- the group twin of oracle_decode_pyramid, and what the decode
    gates of the group tests feed the oracle's inverse with.  lowpass_offset=1 adds the reference decoder's lowpass bias of a group for 8-bit output: twice the intra
    frame's (Codec/decoder.c:12265 `num_frames == 2 ? 48 : 24`; the odd-width path likewise), on words read as unsigned where the band's width is odd (:12468-12545)."""
    O = oracle()
    out = np.zeros(gp.coeff_elems, dtype=np.int16)
    P16 = ctypes.POINTER(ctypes.c_int16)
    dst = (P16 * 4 * 6 * 3)(); pitch = (ctypes.c_int * 4 * 6 * 3)(); dims = (ctypes.c_int * 2 * 4 * 6 * 3)()
    for c in range(3):
        for k in (5, 4, 3, 1, 0):
            d = gp.w[(c, k)]
            for b in range(4):
                if b == 0 and k not in (5, 3): continue                 # (band 0 is coded for the top wavelet -- raw, behind the coefficient marker -- and for the temporal highpass's wavelet)
                v = gp.view(out, c, k, b)
                dst[c][k][b] = v.ctypes.data_as(P16); pitch[c][k][b] = d["pitch"]; dims[c][k][b][0] = d["width"]; dims[c][k][b][1] = d["height"]
    info = (ctypes.c_int32 * 8)()
    s = np.frombuffer(sample, dtype=np.uint8).copy()
    O.orc_decode_group.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = O.orc_decode_group(p8(s), len(sample), ctypes.byref(dst), ctypes.byref(pitch), ctypes.byref(dims), ctypes.byref(info))
    assert rc == 0, "oracle group walk failed: %d" % rc
    assert info[3] == 3 and info[7] == 3 * 17, "group sample has %d channels, %d bands decoded" % (info[3], info[7])
    if lowpass_offset:
        for c in range(3):
            d = gp.w[(c, 5)]
            ll = gp.view(out, c, 5, 0)[:, : d["width"]]
            words = ll.view(np.uint16).astype(np.int32) if d["width"] & 1 else ll.astype(np.int32)
            bias = 2 * oracle_lowpass_bias(10, d["width"], gp.pixkind, c)
            ll[:] = np.minimum(words + bias, 0x7fff).astype(np.int16)
    return out


def oracle_inverse_gop(gp, coeffs, dither, uyvy=0, reference_defect=True):
    """The inverse group transform with the oracle from a dequantized group pyramid: spatial synthesis of w[5], w[4], w[3] (orc_inv_spatial: the descale variant where
    the encoder prescaled; orc_inv_spatial_overflow_protected -- the routine the reference's group decoder runs, defect of its last row included -- where it did not;
    reference_defect=False: the filter as meant, 5 dB better than the reference's own pictures), the temporal step of Codec/wavelet.c TransformInverseTemporal (frame 0 = sat(low - high)
    >> 1, frame 1 = sat(low + high) >> 1; the width % 8 tail columns divide towards zero), the last level of both frames.  Returns two
    packed 8-bit 4:2:2 pictures."""
    O = oracle()
    work = coeffs.copy()
    def inv(c, k, dst_k, dst_b):
        d = gp.w[(c, k)]
        bands = (c_i16p * 4)(*[gp.view(work, c, k, b).ctypes.data_as(c_i16p) for b in range(4)])
        dst = gp.view(work, c, dst_k, dst_b)
        if d["prescale"] == 0 and reference_defect:
            # the wavelets the reference's group decoder sends through InvertSpatialQuantOverflowProtected16s (no prescale: w[5] at level 4, wavelet.c:5759, and the
            # temporal-highpass wavelet w[3], wavelet.c:5886), whose last coefficient row reads the lowpass band one row too high (spatial.c:21770-21830)
            O.orc_inv_spatial_overflow_protected(bands, d["pitch"], d["width"], d["height"], dst.ctypes.data_as(c_i16p), gp.w[(c, dst_k)]["pitch"])
        else:
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], d["prescale"], dst.ctypes.data_as(c_i16p), gp.w[(c, dst_k)]["pitch"])
    for c in range(3):
        inv(c, 5, 4, 0); inv(c, 4, 2, 0); inv(c, 3, 2, 1)
        d = gp.w[(c, 2)]
        lo = gp.view(work, c, 2, 0).astype(np.int32); hi = gp.view(work, c, 2, 1).astype(np.int32)
        even = np.clip(lo - hi, -32768, 32767) >> 1; odd = np.clip(lo + hi, -32768, 32767) >> 1
        tail = d["width"] - d["width"] % 8
        if tail < d["width"]:
            tz = lambda v: np.where(v < 0, -((-v) // 2), v // 2)
            even[:, tail:] = tz(lo[:, tail:] - hi[:, tail:]); odd[:, tail:] = tz(lo[:, tail:] + hi[:, tail:])
        gp.view(work, c, 0, 0)[:] = even.astype(np.int16); gp.view(work, c, 1, 0)[:] = odd.astype(np.int16)
    outs = []
    for f in range(2):
        ptrs = (c_i16p * 12)(*[gp.view(work, c, f, b).ctypes.data_as(c_i16p) for c in range(3) for b in range(4)])
        pitches = [gp.w[(c, f)]["pitch"] for c in range(3)]
        w = gp.w[(0, f)]["width"]; h = gp.w[(0, f)]["height"]
        out = np.zeros((2 * h, 4 * w), np.uint8)
        # the last level: spatial synthesis, or -- interlaced groups -- the inverse frame transform of interlaced intra frames
        (O.orc_inv_frame_to_yuv422 if gp.interlaced else O.orc_inv_spatial_to_yuv422)(ptrs, iarr(pitches), w, h, 10, uyvy, dither, p8(out), 4 * w)
        outs.append(out)
    return outs


def ref_decode_group_frames(samples, width, height, pixfmt=PIX_YUY2):
    """The pictures the reference decoder gives for a stream of two-frame groups (samples: [sequence header,] group, P-frame header, group, ...): a list of
    (frame 0, frame 1) per group, cropped to width x height.  How the reference has to be driven (probed; Codec/decoder.c:11180 DecodeSampleGroup, :11482
    DecodeSampleFrame): the handle is prepared on the first GROUP sample (the 40-byte sequence header carries the coded height only), one worker thread
    (TAG_CPU_MAX = 1, as every RefDecoder here), and every group sample is decoded TWICE -- the first CFHD_DecodeSample of a new group returns a picture put
    together from the previous group's wavelets (its entropy decode runs behind the reconstruction of the first frame; noise for the first group of a stream,
    a 20 dB ghost of the previous group afterwards), a later call on the same sample returns the group's first frame; the P-frame sample behind it then
    returns the second."""
    groups = [k for k, s in enumerate(samples) if len(s) > 64]
    d = RefDecoder(samples[groups[0]], pixfmt, 1, 1)
    def dec(s):
        sb = ctypes.create_string_buffer(s, len(s)); out = np.zeros(d.pitch * d.height + 64, np.uint8)
        assert d.decode(sb, len(s), out) == 0
        return out[: d.pitch * d.height].reshape(d.height, d.pitch)[:height, : width * 2].copy()
    frames = []
    for k in groups:
        # ... and its second call is not always complete either (seen once in three runs at 1080p: the chroma channels still the previous call's -- the worker that
        # decodes them races the reconstruction): the group sample is decoded until two consecutive calls return the same picture up to the dither bit
        prev = dec(samples[k]); f0 = dec(samples[k])
        for _ in range(6):
            if (np.abs(f0.astype(np.int16) - prev) <= 1).all(): break
            prev = f0; f0 = dec(samples[k])
        f1 = dec(samples[k + 1]) if k + 1 < len(samples) and len(samples[k + 1]) <= 64 else None
        frames.append((f0, f1))
    d.close()
    return frames


def oracle_rgb8_to_yuv422_planes(frame, pitch, bytes_per_pixel, top_down, w, h, enc_height, color_space=0):
    """8-bit RGB(A) frame (bytes B, G, R(, A)) -> the three 10-bit planes Y, channel 1, channel 2 of a 4:2:2 frame at the coded height, with the
    oracle (Codec/frame.c:378 ConvertRGB32to10bitYUVFrame)."""
    O = oracle()
    O.orc_rgb8_to_yuv422.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    src = np.ascontiguousarray(np.frombuffer(frame.tobytes(), np.uint8))
    Y = np.zeros((enc_height, w), np.int16); C1 = np.zeros((enc_height, w // 2), np.int16); C2 = np.zeros((enc_height, w // 2), np.int16)
    O.orc_rgb8_to_yuv422(src.ctypes.data_as(ctypes.c_void_p), pitch, bytes_per_pixel, top_down, w, h, enc_height, color_space, p16(Y), w, p16(C1), p16(C2), w // 2)
    return [Y, C1, C2]


def oracle_inverse_rgba8(plan, coeffs, bottom_up):
    """Whole inverse path with the oracle from a dequantized RGBA 4:4:4:4 pyramid to 8-bit B, G, R, A pixels (no dither on this route).  Returns (bytes, the
    alternative alpha bytes of a row on which the reference lost its alpha_Companded race: the companded value rounded the same way)."""
    O = oracle()
    O.orc_inv_spatial_to_rgba8.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_int]
    work = with_lowpass_bias(plan, coeffs, 8)
    biased = work.copy()
    for c in range(4):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    d = plan.band[(0, 0, 0)]
    flat = [plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(4) for b in range(4)]
    out = np.zeros((plan.height, 2 * d["width"] * 4), np.uint8)
    O.orc_inv_spatial_to_rgba8((c_i16p * 16)(*flat), d["pitch"], d["width"], d["height"], plan.precision, plan.height, int(bottom_up), out.ctypes.data_as(ctypes.c_void_p), out.shape[1])
    raw = oracle_inverse_rgb48(plan, biased, b64a=False)[: plan.height].reshape(plan.height, -1, 4)[:, :, 3].astype(np.int64)
    alt = np.minimum((raw >> 4) >> 4, 255).astype(np.uint8)
    return out, (alt[::-1] if bottom_up else alt)


def oracle_inverse_b64a_of_rgb444(plan, coeffs):
    """Whole inverse path with the oracle from a dequantized RGB 4:4:4 pyramid to b64a words A, R, G, B (alpha 0xfff0): orc_inv_spatial_to_b64a_of_rgb444."""
    O = oracle()
    O.orc_inv_spatial_to_b64a_of_rgb444.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    d = plan.band[(0, 0, 0)]
    flat = [plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(3) for b in range(4)]
    out = np.zeros((2 * d["height"], 2 * d["width"] * 4), np.uint16)
    O.orc_inv_spatial_to_b64a_of_rgb444((c_i16p * 16)(*(flat + [None] * 4)), d["pitch"], d["width"], d["height"], plan.precision, out.ctypes.data_as(ctypes.c_void_p), out.shape[1])
    return out


def oracle_inverse_rgb24_of_yuv422(plan, coeffs, d, color_space=2):
    """Whole inverse path with the oracle from a dequantized 4:2:2 pyramid to RG24 bytes (bottom row first) with the 15-bit dither value d (0 .. 32767)."""
    O = oracle()
    O.orc_inv_spatial_to_rgb24_of_yuv422.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_int]
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            dsc = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, dsc["pitch"], dsc["width"], dsc["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    ptrs = (c_i16p * 12)(*[plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(3) for b in range(4)])
    pitches = [plan.band[(c, 0, 0)]["pitch"] for c in range(3)]
    w = plan.band[(0, 0, 0)]["width"]; h = plan.band[(0, 0, 0)]["height"]
    out = np.zeros((plan.height, 2 * w * 3), np.uint8)
    O.orc_inv_spatial_to_rgb24_of_yuv422(ptrs, iarr(pitches), w, h, plan.precision, plan.height, color_space, d, out.ctypes.data_as(ctypes.c_void_p), out.shape[1])
    return out


def _oracle_levels_3_2_of_yuv422(plan, coeffs):
    """The upper two inverse levels of a 4:2:2 pyramid with the oracle; returns (work pyramid, band pointers of level 1, pitches, band width, band height)."""
    O = oracle()
    work = coeffs.copy()
    for c in range(3):
        for lv in (2, 1):
            dsc = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, dsc["pitch"], dsc["width"], dsc["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    ptrs = (c_i16p * 12)(*[plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(3) for b in range(4)])
    return work, ptrs, [plan.band[(c, 0, 0)]["pitch"] for c in range(3)], plan.band[(0, 0, 0)]["width"], plan.band[(0, 0, 0)]["height"]


def oracle_inverse_rgb16_of_yuv422(plan, coeffs, b64a, color_space=2):
    """Whole inverse path with the oracle from a dequantized 4:2:2 pyramid (Plan(..., pixkind=PIXKIND["RG48"] or ["b64a"], enc=ENC["422"]): lowpass bias 24 / 5) to
    RG48 words (R, G, B) or b64a words (0xffff, R, G, B): orc_inv_spatial_to_rgb16_of_yuv422, pinned on the reference decoder."""
    O = oracle()
    O.orc_inv_spatial_to_rgb16_of_yuv422.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_int]
    work, ptrs, pitches, w, h = _oracle_levels_3_2_of_yuv422(plan, coeffs)
    nw = 4 if b64a else 3
    out = np.zeros((plan.height, 2 * w * nw), np.uint16)
    O.orc_inv_spatial_to_rgb16_of_yuv422(ptrs, iarr(pitches), w, h, plan.precision, plan.height, color_space, int(bool(b64a)), out.ctypes.data_as(ctypes.c_void_p), out.shape[1])
    return out


def oracle_inverse_rgb32_of_yuv422(plan, coeffs, bottom_up, color_space=2):
    """Whole inverse path with the oracle from a dequantized 4:2:2 pyramid (Plan(..., pixkind=PIXKIND["BGRA"] or ["BGRa"], enc=ENC["422"]): the bias of these outputs,
    which for odd lowpass widths differs between the two) to bytes B, G, R, 255 -- BGRA: bottom row first, BGRa: top row first: orc_inv_spatial_to_rgb32_of_yuv422,
    pinned on the reference decoder (no dither on this route)."""
    O = oracle()
    O.orc_inv_spatial_to_rgb32_of_yuv422.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_int]
    work, ptrs, pitches, w, h = _oracle_levels_3_2_of_yuv422(plan, coeffs)
    out = np.zeros((plan.height, 2 * w * 4), np.uint8)
    O.orc_inv_spatial_to_rgb32_of_yuv422(ptrs, iarr(pitches), w, h, plan.precision, plan.height, color_space, int(bool(bottom_up)), out.ctypes.data_as(ctypes.c_void_p), out.shape[1])
    return out


def oracle_inverse_byr4(plan, coeffs, curve=True, quad_rows=None):
    """Whole inverse path with the oracle from a dequantized Bayer pyramid (planes G, R-G, B-G, G1-G2) to the BYR4 mosaic (rows r g1 / g2 b), through the
    reference's linear-restore table (orc_byr4_linear_restore_curve) unless curve is False: orc_inv_spatial_to_byr4."""
    O = oracle()
    O.orc_inv_spatial_to_byr4.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    work = coeffs.copy()
    for c in range(4):
        for lv in (2, 1):
            d = plan.band[(c, lv, 0)]
            bands = (c_i16p * 4)(*[plan.view(work, c, lv, b).ctypes.data_as(c_i16p) for b in range(4)])
            dst = plan.view(work, c, lv - 1, 0)
            O.orc_inv_spatial(bands, d["pitch"], d["width"], d["height"], plan.prescale[lv], dst.ctypes.data_as(c_i16p), plan.band[(c, lv - 1, 0)]["pitch"])
    d = plan.band[(0, 0, 0)]
    flat = [plan.view(work, c, 0, b).ctypes.data_as(c_i16p) for c in range(4) for b in range(4)]
    lut = np.zeros(16384, np.uint16)
    O.orc_byr4_linear_restore_curve(lut.ctypes.data_as(ctypes.c_void_p))
    quad_rows = quad_rows or 2 * d["height"]
    out = np.zeros((2 * quad_rows, 4 * d["width"]), np.uint16)
    O.orc_inv_spatial_to_byr4((c_i16p * 16)(*flat), d["pitch"], d["width"], d["height"], plan.precision, quad_rows, lut.ctypes.data_as(ctypes.c_void_p) if curve else None,
                              out.ctypes.data_as(ctypes.c_void_p), out.shape[1])
    return out


def pack_byr5(mosaic):
    """A 16-bit Bayer mosaic (red-green order) as a BYR5 frame: 12-bit values (>> 4), per row pair the four components R, G1, G2, B as runs of high bytes, then
    their low nibbles two to a byte, the even sample's in the low half (CFHDTypes.h: "packed line of 8-bit then line a 4-bit reminder")."""
    H, W = mosaic.shape
    v = (mosaic >> 4).astype(np.uint16)
    rows = []
    for r in range(H // 2):
        comp = np.concatenate([v[2 * r, 0::2], v[2 * r, 1::2], v[2 * r + 1, 0::2], v[2 * r + 1, 1::2]])
        lo = (comp & 15).astype(np.uint8)
        rows.append(np.concatenate([(comp >> 4).astype(np.uint8), (lo[0::2] | (lo[1::2] << 4)).astype(np.uint8)]))
    return np.concatenate(rows)


def byr5_planes(frame, w, h):
    """G, R-G, B-G, G1-G2 planes (w x h each) of a BYR5 frame through the oracle's restatement of ConvertBYR5ToFrame16s."""
    O = oracle()
    O.orc_byr5_unpack_row.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
    rows = np.ascontiguousarray(np.asarray(frame, np.uint8).reshape(h, 6 * w))
    planes = [np.zeros((h, w), np.int16) for _ in range(4)]
    for r in range(h):
        O.orc_byr5_unpack_row(rows[r].ctypes.data_as(ctypes.c_void_p), w, *[p[r].ctypes.data_as(ctypes.c_void_p) for p in planes])
    return planes


def rg64_frame(seed, w, h):
    """An RG64 frame (16-bit words R, G, B, A) with the picture and alpha of the harness's b64a Qbist frame: (bytes, pitch, words h x w x 4)."""
    fb, pb = qbist_frames(seed, 1, w, h, PIX_B64A, alpha=1)
    b = np.frombuffer(fb[0].tobytes(), np.uint16).reshape(h, pb // 2)[:, : w * 4].reshape(h, w, 4)
    words = np.ascontiguousarray(b[:, :, [1, 2, 3, 0]])
    return np.frombuffer(words.tobytes(), np.uint8).copy(), w * 8, words
