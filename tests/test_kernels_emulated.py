"""The product's HIP kernel source (cineform-sdk_amd/csrc/cfhd_kernels.h) executed on the CPU by tests/hipemu
(one OS thread per GPU thread, real barriers) and compared bit for bit with the oracle.  This is the
no-GPU half of the kernel parity tests; tests/test_gpu_parity.py repeats them on the MI355X."""
import ctypes
import numpy as np
import pytest
from cfhd_testlib import *


def rand_plane(rng, w, h, bits, signed=False):
    lo = -(1 << (bits - 1)) if signed else 0
    hi = (1 << (bits - 1)) - 1 if signed else (1 << bits) - 1
    return rng.integers(lo, hi + 1, size=(h, w), dtype=np.int64).astype(np.int16)


@pytest.mark.parametrize("w,h", [(16, 8), (64, 16), (90, 34), (136, 70), (240, 134), (480, 66)])
@pytest.mark.parametrize("prescale", [0, 2])
def test_fwd_plane(w, h, prescale):
    rng = np.random.default_rng(w * 11 + h + prescale)
    x = rand_plane(rng, w, h, 12 if prescale == 0 else 14)
    quant = [1, 24, 12, 36] if prescale == 0 else [1, 6, 6, 3]
    hw, hh = w // 2, h // 2
    pitch = (hw + 7) // 8 * 8
    o = [np.zeros((hh, hw), np.int16) for _ in range(4)]
    e = [np.full((hh, pitch), 77, np.int16) for _ in range(4)]
    bands = (c_i16p * 4)(*[p16(a) for a in o])
    oracle().orc_fwd_spatial(p16(x), w, w, h, prescale, iarr(quant), 2, bands, hw)
    emu().emu_fwd_plane(p16(x), w, w, h, prescale, iarr(quant), 2, p16(e[0]), p16(e[1]), p16(e[2]), p16(e[3]), pitch)
    for k in range(4):
        assert np.array_equal(e[k][:, :hw], o[k]), "band %d" % k
        if hw & 1:
            assert np.all(e[k][:, hw] == 0)     # the pad column next to an odd width is written as zero


@pytest.mark.parametrize("w,h,dh", [(64, 16, 16), (128, 48, 48), (192, 40, 34), (720, 64, 64)])
@pytest.mark.parametrize("uyvy", [0, 1])
def test_fwd_yuv422(w, h, dh, uyvy):
    rng = np.random.default_rng(w + h + uyvy)
    frame = rng.integers(0, 256, size=(dh, w * 2), dtype=np.int64).astype(np.uint8)
    padded = np.full((h, w * 2), 128, np.uint8); padded[:dh] = frame
    quant = [1, 24, 24, 36, 1, 24, 24, 48, 1, 24, 24, 48]
    outs_e, outs_o, pitches = [], [], []
    for ch in range(3):
        cw = (w if ch == 0 else w // 2) // 2
        pitches.append((cw + 7) // 8 * 8)
        outs_e.append([np.zeros((h // 2, pitches[-1]), np.int16) for _ in range(4)])
        outs_o.append([np.zeros((h // 2, cw), np.int16) for _ in range(4)])
        bands = (c_i16p * 4)(*[p16(a) for a in outs_o[-1]])
        oracle().orc_fwd_spatial_yuv422(p8(padded), w * 2, cw * 2, h, ch, 2, uyvy, iarr(quant[4 * ch:4 * ch + 4]), 2, bands, cw)
    ptrs = (c_i16p * 12)(*[p16(a) for ch in range(3) for a in outs_e[ch]])
    emu().emu_fwd_yuv422(p8(frame), w * 2, w, h, dh, uyvy, 2, iarr(quant), 2, ptrs, iarr(pitches))
    for ch in range(3):
        for k in range(4):
            assert np.array_equal(outs_e[ch][k][:, :outs_o[ch][k].shape[1]], outs_o[ch][k]), (ch, k)


@pytest.mark.parametrize("w,h,dh", [(32, 8, 8), (64, 16, 13), (192, 40, 34), (960, 72, 72), (1984, 8, 8), (2016, 16, 16), (2048, 8, 8), (3840, 16, 14), (48, 8, 8)])
@pytest.mark.parametrize("uyvy", [0, 1])
def test_fwd_yuv422_strip_kernel(w, h, dh, uyvy):
    """k_fwd_yuv422_strip (register windows + lane exchange, 16-byte accesses) = oracle, incl. full-range samples, strips that start
    inside the picture (h > 32), rows below the display height and frames of several segments (1984 pixels each)."""
    rng = np.random.default_rng(w + 7 * h + uyvy)
    frame = rng.integers(0, 256, size=(dh, w * 2), dtype=np.int64).astype(np.uint8)
    frame[: dh // 2, : w] = rng.choice([0, 255], size=(dh // 2, w))
    padded = np.full((h, w * 2), 128, np.uint8); padded[:dh] = frame
    quant = [1, 24, 24, 36, 1, 24, 24, 48, 1, 3, 1, 48]
    outs_e, outs_o, pitches = [], [], []
    for ch in range(3):
        cw = (w if ch == 0 else w // 2) // 2
        pitches.append((cw + 7) // 8 * 8 + (8 if ch == 2 else 0))
        outs_e.append([np.full((h // 2, pitches[-1]), 77, np.int16) for _ in range(4)])
        outs_o.append([np.zeros((h // 2, cw), np.int16) for _ in range(4)])
    E = emu()
    ptrs = (c_i16p * 12)(*[p16(a) for ch in range(3) for a in outs_e[ch]])
    rc = E.emu_fwd_yuv422_strip(p8(frame), w * 2, w, h, dh, uyvy, 2, iarr(quant), 2, ptrs, iarr(pitches))
    if w % 32:
        assert rc == -1
        return
    assert rc == 0
    for ch in range(3):
        cw = (w if ch == 0 else w // 2) // 2
        bands = (c_i16p * 4)(*[p16(a) for a in outs_o[ch]])
        oracle().orc_fwd_spatial_yuv422(p8(padded), w * 2, cw * 2, h, ch, 2, uyvy, iarr(quant[4 * ch:4 * ch + 4]), 2, bands, cw)
        for k in range(4):
            assert np.array_equal(outs_e[ch][k][:, :cw], outs_o[ch][k]), (ch, k)
            assert np.all(outs_e[ch][k][:, cw:] == 77)                       # nothing written beyond the band


@pytest.mark.parametrize("w,h", [(8, 4), (45, 17), (64, 8), (120, 135), (130, 20), (66, 33), (129, 18), (64, 16), (200, 35), (3, 3)])
@pytest.mark.parametrize("descale", [0, 2])
def test_inv_plane(w, h, descale):
    rng = np.random.default_rng(w * 5 + h + descale)
    pitch = (w + 7) // 8 * 8
    b = [np.zeros((h, pitch), np.int16) for _ in range(4)]
    b[0][:, :w] = rand_plane(rng, w, h, 13)
    for k in range(1, 4): b[k][:, :w] = rand_plane(rng, w, h, 11, signed=True)
    o = np.zeros((2 * h, 2 * w), np.int16); e = np.zeros((2 * h, 2 * pitch), np.int16)
    bands = (c_i16p * 4)(*[p16(a) for a in b])
    oracle().orc_inv_spatial(bands, pitch, w, h, descale, p16(o), 2 * w)
    emu().emu_inv_plane(p16(b[0]), p16(b[1]), p16(b[2]), p16(b[3]), pitch, w, h, descale, p16(e), 2 * pitch)
    assert np.array_equal(e[:, :2 * w], o)


@pytest.mark.parametrize("w,h", [(8, 4), (45, 17), (64, 16), (130, 18), (66, 33), (129, 34), (40, 5), (3, 3)])
def test_inv_plane_with_the_reference_group_decoders_last_row(w, h):
    """k_inv_plane with InvPlaneJob::ll_bottom_row_high = the oracle's restatement of the reference's InvertSpatialQuantOverflowProtected16s (Codec/spatial.c:21114: the
    routine its group decoder runs for the unprescaled wavelets, whose last coefficient row reads the LL band one row too high) -- last tiles of one, two and more rows,
    a band of three rows (too short for the shift: the plain filter)."""
    rng = np.random.default_rng(w * 7 + h)
    pitch = (w + 7) // 8 * 8
    b = [np.zeros((h, pitch), np.int16) for _ in range(4)]
    b[0][:, :w] = rand_plane(rng, w, h, 13)
    for k in range(1, 4): b[k][:, :w] = rand_plane(rng, w, h, 11, signed=True)
    o = np.zeros((2 * h, 2 * w), np.int16); e = np.zeros((2 * h, 2 * pitch), np.int16); plain = np.zeros((2 * h, 2 * w), np.int16)
    bands = (c_i16p * 4)(*[p16(a) for a in b])
    oracle().orc_inv_spatial_overflow_protected(bands, pitch, w, h, p16(o), 2 * w)
    oracle().orc_inv_spatial(bands, pitch, w, h, 0, p16(plain), 2 * w)
    emu().emu_inv_plane_ex(p16(b[0]), p16(b[1]), p16(b[2]), p16(b[3]), pitch, w, h, 0, p16(e), 2 * pitch, 1)
    assert np.array_equal(e[:, :2 * w], o)
    assert np.array_equal(o[:-2], plain[:-2]) and (h < 4 or not np.array_equal(o[-2:], plain[-2:]))      # only the last coefficient row differs from the filter as meant


@pytest.mark.parametrize("w,rows,b64a", [(8, 2, 0), (160, 3, 1), (960, 2, 0), (2056, 2, 1)])
def test_half_resolution_output_kernel_16bit(w, rows, b64a):
    """k_half_packed16 = the model of the reference's half-resolution RG48 / b64a output (pinned in tests/test_oracle_vs_ref.py)."""
    rng = np.random.default_rng(w + rows + b64a)
    nch = 4 if b64a else 3
    pitch = (w + 7) // 8 * 8 + 8
    planes = [rng.integers(-200, 17000, size=(rows, pitch)).astype(np.int16) for _ in range(nch)]
    want = half_resolution_model16([p[:, :w] for p in planes], bool(b64a))
    words = [2, 1, 3, 0] if b64a else [1, 0, 2]
    out = np.full((rows, w * nch + 8), 9, np.uint16)
    E = emu()
    E.emu_half_packed16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    E.emu_half_packed16((c_i16p * nch)(*[p16(p) for p in planes]), nch, pitch, w, rows, iarr(words), 2, b64a, out.ctypes.data_as(ctypes.c_void_p), (w * nch + 8) * 2)
    assert np.array_equal(out[:, : w * nch], want)
    assert np.all(out[:, w * nch:] == 9)
    assert (want == 0).any() and (want == 65535).any()


@pytest.mark.parametrize("w,rows,uyvy", [(8, 3, 0), (168, 5, 1), (960, 4, 0), (2056, 2, 1)])
def test_half_resolution_output_kernel(w, rows, uyvy):
    """k_half_yuv422: SATURATE_8U(lowpass >> 4) of the level-1 lowpass planes, interleaved Y U Y V / U Y V Y (pinned against the reference's
    half-resolution decode in the GPU tests and, as this model, on the CPU: tests/test_oracle_vs_ref.py)."""
    rng = np.random.default_rng(w + rows + uyvy)
    planes, pitches = [], []
    for c in range(3):
        cw = w if c == 0 else w // 2
        pitch = (cw + 7) // 8 * 8 + 8
        p = rng.integers(-300, 4800, size=(rows, pitch)).astype(np.int16)
        planes.append(p); pitches.append(pitch)
    out = np.full((rows, 2 * w + 16), 9, np.uint8)
    emu().emu_half_yuv422((c_i16p * 3)(*[p16(p) for p in planes]), iarr(pitches), w, rows, uyvy, p8(out), 2 * w + 16)
    want = half_resolution_model(planes[0][:, :w], planes[1][:, :w // 2], planes[2][:, :w // 2], uyvy)
    assert np.array_equal(out[:, :2 * w], want)
    assert np.all(out[:, 2 * w:] == 9)
    assert (want == 0).any() and (want == 255).any()


@pytest.mark.parametrize("w,h,nplanes", [(16, 8, 9), (64, 16, 5), (240, 134, 3), (480, 66, 3), (960, 40, 2), (1024, 8, 1), (1040, 8, 1), (1056, 36, 3), (1920, 40, 2), (2000, 12, 1), (3840, 10, 1)])
@pytest.mark.parametrize("prescale", [0, 2])
def test_fwd_plane_strip_kernel(w, h, nplanes, prescale):
    """k_fwd_plane_strip (levels 2 / 3: several planes per wave, six-row register window, packed quantizer) = oracle, plane by plane."""
    rng = np.random.default_rng(w * 13 + h + prescale)
    quant = [1, 24, 12, 36] if prescale == 0 else [1, 6, 6, 3]
    hw, hh = w // 2, h // 2
    pitch = (hw + 7) // 8 * 8 + 8
    ins, want, got = [], [], []
    for k in range(nplanes):
        x = np.zeros((h, w + 8), np.int16); x[:, w:] = 1234
        x[:, :w] = rand_plane(rng, w, h, 12 if prescale == 0 else 14)
        if k == 1 and prescale == 0: x[:, :w] = rng.choice(np.array([-32768, 32767, 0], np.int16), size=(h, w))       # saturating arithmetic and the quantizer's wrap
        ins.append(x)
        o = [np.zeros((hh, hw), np.int16) for _ in range(4)]
        oracle().orc_fwd_spatial(p16(x), w + 8, w, h, prescale, iarr(quant), 2, (c_i16p * 4)(*[p16(a) for a in o]), hw)
        want.append(o)
        got.append([np.full((hh, pitch), 77, np.int16) for _ in range(4)])
    E = emu()
    E.emu_fwd_plane_strip.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    rc = E.emu_fwd_plane_strip((c_i16p * nplanes)(*[p16(x) for x in ins]), nplanes, w + 8, w, h, prescale, iarr(quant), 2,
                               (c_i16p * (4 * nplanes))(*[p16(a) for g in got for a in g]), pitch)
    assert rc == 0                                       # (planes of more than 64 blocks: one plane per wave in segments of 62 blocks)
    for k in range(nplanes):
        for b in range(4):
            assert np.array_equal(got[k][b][:, :hw], want[k][b]), (k, b)
            assert np.all(got[k][b][:, hw:] == 77)


@pytest.mark.parametrize("w,h,nplanes", [(8, 4, 9), (64, 8, 3), (120, 135, 5), (240, 33, 3), (480, 18, 2), (512, 16, 1), (520, 8, 1), (528, 20, 3), (960, 18, 2), (1000, 6, 1), (1920, 5, 1)])
@pytest.mark.parametrize("descale", [0, 2])
def test_inv_plane_strip_kernel(w, h, nplanes, descale):
    """k_inv_plane_strip (several planes side by side in one wave, neighbours by lane exchange, 16-byte accesses) = oracle, plane by plane."""
    rng = np.random.default_rng(w * 7 + h + descale)
    pitch = (w + 7) // 8 * 8 + 8
    planes, outs, want = [], [], []
    for k in range(nplanes):
        b = [rng.integers(-9, 9, size=(h, pitch)).astype(np.int16) for _ in range(4)]               # junk in the pad columns
        b[0][:, :w] = rand_plane(rng, w, h, 13)
        for i in range(1, 4): b[i][:, :w] = rand_plane(rng, w, h, 11, signed=True)
        planes.append(b)
        outs.append(np.full((2 * h, 2 * pitch), 55, np.int16))
        o = np.zeros((2 * h, 2 * w), np.int16)
        oracle().orc_inv_spatial((c_i16p * 4)(*[p16(a) for a in b]), pitch, w, h, descale, p16(o), 2 * w)
        want.append(o)
    E = emu()
    E.emu_inv_plane_strip.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    rc = E.emu_inv_plane_strip((c_i16p * (4 * nplanes))(*[p16(a) for b in planes for a in b]), nplanes, pitch, w, h, descale,
                               (c_i16p * nplanes)(*[p16(o) for o in outs]), 2 * pitch)
    assert rc == 0                                       # (more than 64 blocks: segments of 62)
    for k in range(nplanes):
        assert np.array_equal(outs[k][:, :2 * w], want[k]), k
        assert np.all(outs[k][:, 2 * w:] == 55)


@pytest.mark.parametrize("w,h,dh", [(32, 8, 16), (96, 20, 40), (360, 30, 58), (128, 17, 34), (132, 33, 66), (64, 16, 32), (260, 19, 37)])
@pytest.mark.parametrize("uyvy", [0, 1])
def test_inv_yuv422(w, h, dh, uyvy):
    """Emulated last-level kernel vs oracle: every output byte must equal the oracle with dither 0 or with dither 1."""
    rng = np.random.default_rng(w + h + uyvy)
    bands, pitches = [], []
    for ch in range(3):
        cw = w if ch == 0 else w // 2
        pitch = (cw + 7) // 8 * 8; pitches.append(pitch)
        bs = [np.zeros((h, pitch), np.int16) for _ in range(4)]
        bs[0][:, :cw] = rand_plane(rng, cw, h, 11)              # LL1 of 10-bit video: <= 4*1020
        for k in range(1, 4): bs[k][:, :cw] = rand_plane(rng, cw, h, 9, signed=True)
        bands.append(bs)
    ptrs = (c_i16p * 12)(*[p16(a) for ch in range(3) for a in bands[ch]])
    outs = []
    for dither in (0, 1):
        o = np.zeros((2 * h, 4 * w), np.uint8)
        oracle().orc_inv_spatial_to_yuv422(ptrs, iarr(pitches), w, h, 10, uyvy, dither, p8(o), 4 * w)
        outs.append(o[:dh])
    e = np.zeros((dh, 4 * w), np.uint8)
    emu().emu_inv_yuv422(ptrs, iarr(pitches), w, h, dh, uyvy, 2, 1234, p8(e), 4 * w)
    assert np.all((e == outs[0]) | (e == outs[1]))
    assert np.any(e != outs[0]) and np.any(e != outs[1])        # the dither really toggles


@pytest.mark.parametrize("w,h,dh", [(16, 8, 16), (32, 17, 33), (96, 20, 40), (480, 35, 70), (992, 16, 31), (1008, 16, 31), (1024, 8, 16), (1920, 8, 15), (24, 8, 16)])
@pytest.mark.parametrize("uyvy", [0, 1])
def test_inv_yuv422_strip_kernel(w, h, dh, uyvy):
    """k_inv_yuv422_strip (register windows + lane exchange instead of LDS tiles) vs the oracle, and byte for byte equal to
    k_inv_yuv422 with the same dither seed: the two kernels are interchangeable.  Bands wider than 992 columns (124 blocks) take several
    segments (1008: a second segment of two blocks; 1920: the 3840-pixel frame); 24 is not a multiple of 16 and is refused."""
    rng = np.random.default_rng(w + 3 * h + uyvy)
    bands, pitches = [], []
    for ch in range(3):
        cw = w if ch == 0 else w // 2
        pitch = (cw + 7) // 8 * 8 + (8 if ch == 1 else 0); pitches.append(pitch)
        bs = [rng.integers(-50, 50, size=(h, pitch)).astype(np.int16) for _ in range(4)]          # the pad columns hold junk: they must not matter
        bs[0][:, :cw] = rand_plane(rng, cw, h, 11)
        for k in range(1, 4): bs[k][:, :cw] = rand_plane(rng, cw, h, 9, signed=True)
        bands.append(bs)
    ptrs = (c_i16p * 12)(*[p16(a) for ch in range(3) for a in bands[ch]])
    E = emu()
    got = np.full((dh, 4 * w + 16), 7, np.uint8)
    rc = E.emu_inv_yuv422_strip(ptrs, iarr(pitches), w, h, dh, uyvy, 2, 99, p8(got), 4 * w + 16)
    if w % 16:
        assert rc == -1
        return
    assert rc == 0
    assert np.all(got[:, 4 * w:] == 7)
    outs = []
    for dither in (0, 1):
        o = np.zeros((2 * h, 4 * w), np.uint8)
        oracle().orc_inv_spatial_to_yuv422(ptrs, iarr(pitches), w, h, 10, uyvy, dither, p8(o), 4 * w)
        outs.append(o[:dh])
    e = got[:, :4 * w]
    assert np.all((e == outs[0]) | (e == outs[1]))
    assert np.any(e != outs[0]) and np.any(e != outs[1])
    tile = np.zeros((dh, 4 * w), np.uint8)
    E.emu_inv_yuv422(ptrs, iarr(pitches), w, h, dh, uyvy, 2, 99, p8(tile), 4 * w)
    assert np.array_equal(tile, e)


def _emu_entropy(plan, coeffs, frame_number, meta):
    out = np.zeros(plan.width * plan.height * 4 + 65536, dtype=np.uint8)
    m = np.frombuffer(meta, dtype=np.uint8).copy()
    E = emu()
    E.emu_entropy_encode.restype = ctypes.c_long
    E.emu_entropy_encode.argtypes = [ctypes.c_int] * 4 + [ctypes.c_uint, c_i16p, c_u8p, ctypes.c_size_t, c_u8p, ctypes.c_size_t]
    n = E.emu_entropy_encode(plan.width, plan.height, plan.pixkind, plan.quality, frame_number, p16(coeffs), p8(m), len(meta), p8(out), out.size)
    assert n > 0, n
    return bytes(out[:n])


@pytest.mark.parametrize("w,h,seed", [(192, 96, 1), (336, 252, 3), (720, 480, 4)])
def test_gpu_entropy_stage_emulated_equals_host_writer(w, h, seed):
    """The four entropy kernels (count / scan / layout / emit) under emulation produce the complete sample byte for byte
    as the product's host writer does (which test_host_bitstream pins against the reference encoder)."""
    frame, pitch = synth_yuy2(w, h, seed)
    plan = Plan(w, h)
    coeffs = oracle_forward_yuv422(plan, frame, pitch)
    meta = b"GUID\x10\x00\x00G" + bytes(range(16)) + b"UFRM\x04\x00\x00L" + (7).to_bytes(4, "little")
    want = product_write_sample_host(plan, coeffs, 5, meta_global=meta)
    got = _emu_entropy(plan, coeffs, 5, meta)
    assert len(got) == len(want)
    if got != want:
        first = next(k for k in range(len(got)) if got[k] != want[k])
        raise AssertionError("first difference at byte %d of %d" % (first, len(got)))


def test_gpu_entropy_stage_emulated_extreme_bands():
    """All-zero frame (runs of hundreds of thousands of zeros spanning every segment), saturated values (clamp to +-1023),
    dense noise (no zeros at all), single tokens behind very long runs."""
    w, h = 416, 208
    plan = Plan(w, h)
    rng = np.random.default_rng(5)
    meta = b"GUID\x10\x00\x00G" + bytes(16)
    for mode in ("zero", "sparse", "lonely", "dense", "huge", "ladder"):
        coeffs = np.zeros(plan.coeff_elems, dtype=np.int16)
        first_gap = 0
        for c in range(3):
            plan.view(coeffs, c, 2, 0)[:, : plan.band[(c, 2, 0)]["width"]] = rng.integers(0, 16000, size=(plan.band[(c, 2, 0)]["height"], plan.band[(c, 2, 0)]["width"]))
            for lv in range(3):
                for b in (1, 2, 3):
                    d = plan.band[(c, lv, b)]; v = plan.view(coeffs, c, lv, b)[:, : d["width"]]
                    if mode == "sparse": v[rng.random(v.shape) < 0.0005] = 3
                    elif mode == "lonely": v[-1, -1] = -7; v[d["height"] // 2, 3] = 2         # runs of several thousand zeros in front of a token: many copies of the longest run code
                    elif mode == "dense": v[:] = rng.integers(1, 40, size=v.shape) * rng.choice([-1, 1], size=v.shape)
                    elif mode == "huge": v[:] = rng.choice([0, 0, 0, 5000, -5000, 1023, -1024, 1], size=v.shape)
                    elif mode == "ladder":
                        # zero runs of every length from 0 to beyond a segment, one after the other (band by band): every composite run code, and
                        # the runs one code does not cover (the bit strings k_ent_count marks for k_ent_emit's table walk)
                        flat = np.zeros(v.size, dtype=np.int16); at = 0; gap = first_gap
                        while at + gap < flat.size:
                            at += gap; flat[at] = 1 + gap % 7; at += 1; gap += 1
                        v[:] = flat.reshape(v.shape)                       # (runs continue through the zero pad columns: the effective lengths shift a little)
                        first_gap = gap if gap < 1200 else 0
        want = product_write_sample_host(plan, coeffs, 1, meta_global=meta)
        got = _emu_entropy(plan, coeffs, 1, meta)
        assert got == want, mode


def _emu_entropy_interlaced(plan, coeffs, frame_number, meta):
    E = emu()
    out = np.zeros(plan.width * plan.height * 4 + 65536, dtype=np.uint8)
    m = np.frombuffer(meta, dtype=np.uint8).copy()
    E.emu_entropy_encode2.restype = ctypes.c_long
    E.emu_entropy_encode2.argtypes = [ctypes.c_int] * 4 + [ctypes.c_uint, c_i16p, c_u8p, ctypes.c_size_t, c_u8p, ctypes.c_size_t, ctypes.c_int]
    n = E.emu_entropy_encode2(plan.width, plan.height, plan.pixkind, plan.quality, frame_number, p16(coeffs), p8(m), len(meta), p8(out), out.size, 1)
    return n, bytes(out[:max(n, 0)])


@pytest.mark.parametrize("w,h,seed", [(192, 96, 1), (336, 252, 3), (720, 480, 4)])
def test_gpu_entropy_stage_emulated_interlaced(w, h, seed):
    """Interlaced frames: the field-difference band (subband 8 of every channel) goes through the second entropy table (code set 18) and is coded with
    peaks: values beyond +-250 steps enter the stream as +-251 and their products with the divisor fill the table behind the band (k_ent_count / k_ent_scan /
    k_ent_layout / k_ent_peaks).  The sample equals the host writer's, which test_host_bitstream pins against the reference: without peaks, with single
    peaks in odd places (last coefficient of a band, first of a segment, an odd count), and with peaks all over the band."""
    frame, pitch = synth_yuy2(w, h, seed)
    plan = Plan(w, h, progressive=0)
    coeffs = oracle_forward_interlaced_yuv422(plan, frame, pitch)
    for c in range(3): np.clip(plan.view(coeffs, c, 0, 2), -250, 250, out=plan.view(coeffs, c, 0, 2))
    meta = b"GUID\x10\x00\x00G" + bytes(range(16))

    def check(cf, what):
        want = product_write_sample_host(plan, cf, 3, meta_global=meta, progressive=0)
        n, got = _emu_entropy_interlaced(plan, cf, 3, meta)
        assert n > 0, (what, n)
        assert len(got) == len(want), (what, len(got), len(want))
        if got != want:
            first = next(k for k in range(len(got)) if got[k] != want[k])
            raise AssertionError("%s: first difference at byte %d of %d" % (what, first, len(got)))
        return got

    def peak_levels(sample):         # the three optional tags in front of a band coded with peaks: TAG_PEAK_TABLE_OFFSET_L / _H, TAG_PEAK_LEVEL
        return [int.from_bytes(sample[i + 10:i + 12], "big") for i in range(0, len(sample) - 12, 4) if sample[i:i + 2] == b"\xff\xb5" and sample[i + 4:i + 6] == b"\xff\xb4" and sample[i + 8:i + 10] == b"\xff\xb6"]

    assert not any(peak_levels(check(coeffs, "no peaks")))
    for c, v in ((2, 251), (0, -251)):
        peaky = coeffs.copy()
        d = plan.band[(c, 0, 2)]
        plan.view(peaky, c, 0, 2)[d["height"] - 1, d["width"] - 1] = v
        assert sum(1 for l in peak_levels(check(peaky, "one peak in channel %d" % c)) if l) == 1
    ok = coeffs.copy()
    plan.view(ok, 0, 0, 1)[0, 0] = 900                # other bands may hold anything: no table for them
    plan.view(ok, 0, 0, 2)[0, 0] = 250
    assert not any(peak_levels(check(ok, "at the threshold")))
    rng = np.random.default_rng(seed)
    many = coeffs.copy()
    for c in range(3):
        v = plan.view(many, c, 0, 2)[:, :plan.band[(c, 0, 2)]["width"]]      # (the columns beyond the band's width are row padding: zero)
        n = max(3, v.size // 40) | 1                                        # an odd number of draws (the table is padded to a longword)
        v[rng.integers(0, v.shape[0], size=n), rng.integers(0, v.shape[1], size=n)] = rng.integers(251, 1024, size=n) * rng.choice([-1, 1], size=n)
        v[0, 0] = -1023; v[-1, -1] = 777                                    # first and last coefficient of the band
    assert sum(1 for l in peak_levels(check(many, "peaks all over")) if l) == 3
    dense = coeffs.copy()
    v = plan.view(dense, 1, 0, 2)[:, :plan.band[(1, 0, 2)]["width"]]
    v[:] = np.where(rng.integers(0, 2, size=v.shape) > 0, 300, -4000).astype(np.int16)      # every coefficient of one band a peak
    check(dense, "a band of peaks only")


def test_gpu_entropy_stage_emulated_more_peaks_than_a_table_holds():
    """A band with more than 131070 peaks (2 x MAX_CHUNK_SIZE, codec.h:195): the reference writes no table and leaves the band's three tags zero (encoder.c:6557; pinned on the
    reference in test_host_bitstream, "allpeaks") -- k_ent_layout / k_ent_peaks and the host writer do the same; the other channels' bands keep their tables."""
    w, h = 1024, 528
    frame, pitch = synth_yuy2(w, h, 5)
    plan = Plan(w, h, progressive=0)
    coeffs = oracle_forward_interlaced_yuv422(plan, frame, pitch)
    rng = np.random.default_rng(9)
    for c in range(3):
        v = plan.view(coeffs, c, 0, 2)[:, :plan.band[(c, 0, 2)]["width"]]
        v[:] = np.where(rng.integers(0, 2, size=v.shape) > 0, 300, -4000).astype(np.int16)
    assert plan.band[(0, 0, 2)]["width"] * plan.band[(0, 0, 2)]["height"] > 131070 > plan.band[(1, 0, 2)]["width"] * plan.band[(1, 0, 2)]["height"]
    meta = b"GUID\x10\x00\x00G" + bytes(range(16))
    want = product_write_sample_host(plan, coeffs, 3, meta_global=meta, progressive=0)
    n, got = _emu_entropy_interlaced(plan, coeffs, 3, meta)
    assert n > 0 and got == want
    levels = [int.from_bytes(got[i + 10:i + 12], "big") for i in range(0, len(got) - 12, 4) if got[i:i + 2] == b"\xff\xb5" and got[i + 4:i + 6] == b"\xff\xb4" and got[i + 8:i + 10] == b"\xff\xb6"]
    assert len(levels) == 3 and levels[0] == 0 and levels[1] and levels[2]


@pytest.mark.parametrize("parallel", [0, 1, 2, 3])
@pytest.mark.parametrize("w,h,seed", [(192, 96, 1), (336, 252, 3), (720, 480, 4)])
def test_gpu_entropy_decoder_emulated_equals_host_decoder(w, h, seed, parallel):
    """k_dec_bands (one lane per band), k_dec_bands_par (one workgroup per band; 2: fed by the GPU parser k_dec_parse; 3: the low-latency shape k_dec_bands_par_ll) + k_dec_lowpass under emulation reproduce the ORACLE's decoder (oracle_decode_pyramid since round 5; the product's host VLC decoder is tied to it in test_oracle_vs_ref) (dequantized pyramid incl. lowpass bias)."""
    frame, pitch = synth_yuy2(w, h, seed)
    plan = Plan(w, h)
    coeffs = oracle_forward_yuv422(plan, frame, pitch)
    if seed == 3:                                   # long code words: values up to the +-1023 clamp
        v = plan.view(coeffs, 0, 0, 1); v[::7, ::5] = 1023; v[1::9, 2::11] = -1023; v[3::5, 1::13] = 300
    sample = product_write_sample_host(plan, coeffs, 1, meta_global=b"GUID\x10\x00\x00G" + bytes(16))
    want = oracle_decode_pyramid(sample, plan)
    got = np.full(plan.coeff_elems, 99, dtype=np.int16)
    E = emu()
    E.emu_entropy_decode.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_int, c_i16p, ctypes.c_size_t, ctypes.c_int]
    s = np.frombuffer(sample, dtype=np.uint8).copy()
    assert E.emu_entropy_decode(p8(s), len(sample), 1, p16(got), got.size, parallel) == 0
    for (c, lv, b) in plan.band:                    # every coded band incl. its pitch padding (the gaps between bands are nobody's)
        if b == 0 and lv != 2: continue
        cols = plan.band[(c, lv, b)]["width"] if b == 0 else None      # k_dec_lowpass writes the columns it has
        assert np.array_equal(plan.view(got, c, lv, b)[:, :cols], plan.view(want, c, lv, b)[:, :cols]), (c, lv, b)


@pytest.mark.parametrize("parallel", [1, 2, 3])
def test_gpu_entropy_decoder_emulated_survives_damaged_samples(parallel):
    """Truncated samples are refused; garbage inside the code words never writes outside the band nor hangs (error flag or wrong values, no crash)."""
    w, h = 336, 252
    frame, pitch = synth_yuy2(w, h, 5)
    plan = Plan(w, h)
    coeffs = oracle_forward_yuv422(plan, frame, pitch)
    sample = product_write_sample_host(plan, coeffs, 1, meta_global=b"GUID\x10\x00\x00G" + bytes(16))
    E = emu()
    E.emu_entropy_decode.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_int, c_i16p, ctypes.c_size_t, ctypes.c_int]
    guard = 4096
    got = np.full(plan.coeff_elems + guard, 99, dtype=np.int16)
    s = np.frombuffer(sample, dtype=np.uint8).copy()
    assert E.emu_entropy_decode(p8(s), len(sample) // 2 & ~3, 1, p16(got), plan.coeff_elems, parallel) < 0
    rng = np.random.default_rng(11)
    for trial in range(4):
        t = s.copy()
        lo = len(t) // 3 + trial * 1000
        t[lo: lo + 600] = rng.integers(0, 256, 600, dtype=np.uint8)
        E.emu_entropy_decode(p8(t), len(t), 1, p16(got), plan.coeff_elems, parallel)
        assert np.all(got[plan.coeff_elems:] == 99)


@pytest.mark.parametrize("w,h", [(66, 33), (130, 20)])
@pytest.mark.parametrize("descale", [0, 2])
def test_inv_plane_full_range_saturation(w, h, descale):
    """Full-range int16 bands: every saturating step of the packed kernel must clip exactly where the oracle (= the reference's
    _mm_adds_epi16 / _mm_subs_epi16 chain, interior, and its 32-bit border arithmetic) clips."""
    rng = np.random.default_rng(w + 3 * h + descale)
    pitch = (w + 7) // 8 * 8
    b = [np.zeros((h, pitch), np.int16) for _ in range(4)]
    for k in range(4):
        b[k][:, :w] = rng.choice(np.array([-32768, -32767, -20000, -1, 0, 1, 9000, 32767], dtype=np.int16), size=(h, w))
    o = np.zeros((2 * h, 2 * w), np.int16); e = np.zeros((2 * h, 2 * pitch), np.int16)
    bands = (c_i16p * 4)(*[p16(a) for a in b])
    oracle().orc_inv_spatial(bands, pitch, w, h, descale, p16(o), 2 * w)
    emu().emu_inv_plane(p16(b[0]), p16(b[1]), p16(b[2]), p16(b[3]), pitch, w, h, descale, p16(e), 2 * pitch)
    assert np.array_equal(e[:, :2 * w], o)


@pytest.mark.parametrize("w,h,dh,nch", [(32, 16, 16, 3), (136, 40, 37, 3), (192, 64, 64, 4), (320, 48, 48, 3)])
def test_fwd_packed16_level1_of_444_formats(w, h, dh, nch):
    """k_fwd_packed16 (level 1 straight from interleaved 16-bit pixels, RG48 / b64a style) = unpack >> 4 + oracle plane transform per
    component, rows below the display height repeating the last picture row (frame.c:6020-6024)."""
    rng = np.random.default_rng(w + h + nch)
    px = rng.integers(0, 65536, size=(dh, w, nch), dtype=np.int64).astype(np.uint16)
    words = [1, 0, 2, 3][:nch] if nch == 3 else [2, 1, 3, 0]      # RG48: planes G, R, B in R, G, B pixels; b64a: planes G, R, B, A in A, R, G, B pixels
    quant = [1, 12, 12, 24] * nch
    pitch = (w // 2 + 7) // 8 * 8
    outs = [np.zeros((h // 2, pitch), np.int16) for _ in range(4 * nch)]
    ptrs = (c_i16p * (4 * nch))(*[p16(o) for o in outs])
    E = emu()
    E.emu_fwd_packed16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    if nch == 4: px[::3, ::5, 0] = 0; px[1::4, 2::7, 0] = 65535          # alpha extremes stay uncompanded
    E.emu_fwd_packed16(px.ctypes.data_as(ctypes.c_void_p), w * nch, w, h, dh, nch, 4, iarr(words), 3 if nch == 4 else -1, iarr(quant), 2, ptrs, pitch)
    for c in range(nch):
        plane = np.zeros((h, w), np.int16)
        comp = (px[:, :, words[c]] >> 4).astype(np.int32)
        if nch == 4 and c == 3: comp = np.where((comp > 0) & (comp < 4095), ((comp * 223 + 128) >> 8) + 256, comp)     # frame.c:6696-6707
        plane[:dh] = comp.astype(np.int16)
        plane[dh:] = plane[dh - 1]
        want = [np.zeros((h // 2, pitch), np.int16) for _ in range(4)]
        bands = (c_i16p * 4)(*[p16(o) for o in want])
        oracle().orc_fwd_spatial(p16(plane), w, w, h, 0, iarr(quant[:4]), 2, bands, pitch)
        for b in range(4):
            assert np.array_equal(outs[4 * c + b][:, :w // 2], want[b][:, :w // 2]), (c, b)


@pytest.mark.parametrize("w,h,dh,wpp,nch", [(8, 8, 8, 3, 3), (16, 16, 16, 4, 4), (72, 24, 21, 3, 3), (496, 16, 16, 3, 3), (504, 72, 70, 4, 4), (1000, 40, 40, 4, 3),
                                            (1024, 136, 131, 3, 3), (136, 8, 8, 4, 4)])
def test_fwd_packed16_strip_equals_the_tiled_kernel(w, h, dh, wpp, nch):
    """k_fwd_packed16_strip (register strips: a lane = 8 pixels of all planes, segments of 62 blocks, strips of 32 band rows) writes the
    same bands as k_fwd_packed16 (checked against the oracle above): one / several segments and strips, the lanes at the borders, rows
    below the display height, companded alpha, b64a pixels with three planes."""
    rng = np.random.default_rng(w * 7 + h + wpp + nch)
    px = rng.integers(0, 65536, size=(dh, w, wpp), dtype=np.int64).astype(np.uint16)
    if wpp == 4: px[::3, ::5, 0] = 0; px[1::4, 2::7, 0] = 65535; px[2::5, 1::3, 0] >>= 6
    quant = [1, 12, 12, 24, 1, 8, 24, 36, 1, 12, 6, 48, 1, 24, 12, 12][:4 * nch]
    pitch = (w // 2 + 7) // 8 * 8
    E = emu()
    E.emu_fwd_packed16_shapes.argtypes = [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    got = []
    for which in (0, 1):
        outs = [np.full((h // 2, pitch), -7, np.int16) for _ in range(4 * nch)]
        ptrs = (c_i16p * (4 * nch))(*[p16(o) for o in outs])
        E.emu_fwd_packed16_shapes(which, px.ctypes.data_as(ctypes.c_void_p), w * wpp, w, h, dh, wpp, nch, 4, int(nch == 4), iarr(quant), 2, ptrs, pitch)
        got.append(outs)
    for k in range(4 * nch):
        assert np.array_equal(got[0][k][:, :w // 2], got[1][k][:, :w // 2]), (k // 4, k % 4)
        assert (got[1][k][:, w // 2:] == -7).all()            # nothing written beside the band


@pytest.mark.parametrize("w,h,dh,nch,b64a", [(16, 8, 16, 3, 0), (68, 20, 37, 3, 0), (96, 33, 66, 4, 0), (160, 17, 34, 3, 0), (96, 33, 66, 4, 1), (40, 9, 17, 4, 1),
                                             (252, 40, 77, 3, 0), (500, 18, 36, 4, 1), (248, 16, 32, 3, 0)])
def test_inv_packed16_last_level_of_444_formats(w, h, dh, nch, b64a):
    """k_inv_packed16 = oracle RG48 / RG64 / b64a reconstruction (pinned against the reference decoder in test_oracle_vs_ref): exact, incl.
    the 65535-vs-65520 saturation difference between the reference's vector columns and its scalar tail columns and, for b64a, the
    expansion of the companded alpha plane."""
    rng = np.random.default_rng(w * 3 + h + nch)
    pitch = (w + 7) // 8 * 8
    bands = []
    for c in range(nch):
        bs = [np.zeros((h, pitch), np.int16) for _ in range(4)]
        bs[0][:, :w] = rand_plane(rng, w, h, 14)               # LL1 of 12-bit components: up to 4 * 4095, here beyond it to hit the clamps
        for k in range(1, 4): bs[k][:, :w] = rand_plane(rng, w, h, 11, signed=True)
        bands.append(bs)
    words = [2, 1, 3, 0] if b64a else [1, 0, 2, 3][:nch]
    flat = [p16(a) for c in range(nch) for a in bands[c]]
    O = oracle()
    O.orc_inv_spatial_to_rgb48.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    O.orc_inv_spatial_to_b64a.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    want = np.zeros((2 * h, 2 * w * nch), np.uint16)
    if b64a: O.orc_inv_spatial_to_b64a((c_i16p * 16)(*flat), pitch, w, h, 12, want.ctypes.data_as(ctypes.c_void_p), 2 * w * nch)
    else: O.orc_inv_spatial_to_rgb48((c_i16p * 16)(*(flat + [None] * (16 - len(flat)))), pitch, w, h, 12, nch, want.ctypes.data_as(ctypes.c_void_p), 2 * w * nch)
    got = np.full((dh, 2 * w * nch), 7, np.uint16)
    E = emu()
    E.emu_inv_packed16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    E.emu_inv_packed16((c_i16p * len(flat))(*flat), pitch, w, h, dh, nch, 12, iarr(words), got.ctypes.data_as(ctypes.c_void_p), 2 * w * nch, 3 if b64a else -1)
    assert np.array_equal(got, want[:dh])
    if w % 4 == 0 and (b64a or nch == 3):
        # k_inv_packed16_strip (register strips: a lane = 4 band columns of all planes, segments of 62 blocks, strips of 16 band rows): the same words
        got2 = np.full((dh, 2 * w * nch), 7, np.uint16)
        E.emu_inv_packed16_use_strip(1)
        try: E.emu_inv_packed16((c_i16p * len(flat))(*flat), pitch, w, h, dh, nch, 12, iarr(words), got2.ctypes.data_as(ctypes.c_void_p), 2 * w * nch, 3 if b64a else -1)
        finally: E.emu_inv_packed16_use_strip(0)
        assert np.array_equal(got2, want[:dh])
    if b64a:
        a = want[:, 0::4]
        assert (a == 0).any() and (a == 65535).any() and ((a > 0) & (a < 65535)).any()
    assert (want == 65535).any() and (want == 65520).any() and (want == 0).any()


@pytest.mark.parametrize("w,h,dh", [(32, 8, 16), (64, 16, 32), (168, 20, 37), (360, 33, 66), (200, 17, 30)])
def test_inv_yu64_last_level_equals_oracle(w, h, dh):
    """k_inv_packed16 with per-plane widths and word strides (YU64 output of 4:2:2 samples: words Y0 C1 Y1 C2) = oracle restatement of the
    reference's planar 16-bit row route (pinned against the reference decoder in test_oracle_vs_ref), incl. the 65535-vs-(1023 << 6)
    saturation difference between its vector columns and its scalar tail columns, per plane."""
    rng = np.random.default_rng(w * 5 + h)
    bands, pitches = [], []
    for c in range(3):
        cw = w // 2 if c else w
        pitch = (cw + 7) // 8 * 8
        bs = [np.zeros((h, pitch), np.int16) for _ in range(4)]
        bs[0][:, :cw] = rand_plane(rng, cw, h, 12)              # LL1 of 10-bit components: up to 4 * 1023, here beyond it to hit the clamps
        for k in range(1, 4): bs[k][:, :cw] = rand_plane(rng, cw, h, 9, signed=True)
        bands.append(bs); pitches.append(pitch)
    flat = [p16(a) for c in range(3) for a in bands[c]]
    O = oracle()
    O.orc_inv_spatial_to_yu64.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    want = np.zeros((2 * h, 4 * w), np.uint16)
    O.orc_inv_spatial_to_yu64((c_i16p * 12)(*flat), iarr(pitches), w, h, 10, want.ctypes.data_as(ctypes.c_void_p), 4 * w)
    got = np.full((dh, 4 * w), 7, np.uint16)
    E = emu()
    E.emu_inv_yu64.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    E.emu_inv_yu64((c_i16p * 12)(*flat), iarr(pitches), w, h, dh, 10, got.ctypes.data_as(ctypes.c_void_p), 4 * w)
    assert np.array_equal(got, want[:dh])
    assert (want == 65535).any() and (want == 1023 << 6).any() and (want == 0).any()


@pytest.mark.parametrize("w,h,dh,bpp,bottom_up", [(16, 8, 16, 3, 1), (68, 20, 37, 4, 1), (160, 17, 34, 4, 0), (250, 33, 66, 3, 1), (96, 16, 30, 4, 1)])
def test_inv_rgb8_last_level_lies_in_oracle_interval(w, h, dh, bpp, bottom_up):
    """k_inv_packed16 in its byte mode (RG24 / BGRA / BGRa output of RGB 4:4:4 samples): every byte between the oracle's reconstruction with the
    dither value 0 and with 15 (the model pinned on the reference decoder in test_oracle_vs_ref), both ends hit about equally often, alpha
    255, bottom-up row order, nothing written beside the picture."""
    rng = np.random.default_rng(w * 3 + h + bpp)
    pitch = (w + 7) // 8 * 8
    bands = []
    for c in range(3):
        bs = [np.zeros((h, pitch), np.int16) for _ in range(4)]
        bs[0][:, :w] = rand_plane(rng, w, h, 14)
        for k in range(1, 4): bs[k][:, :w] = rand_plane(rng, w, h, 11, signed=True)
        bands.append(bs)
    flat = [p16(a) for c in range(3) for a in bands[c]]
    O = oracle()
    O.orc_inv_spatial_to_rgb8.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_int]
    ends = []
    for r in (0, 127):
        o = np.zeros((dh, 2 * w * bpp), np.uint8)
        O.orc_inv_spatial_to_rgb8((c_i16p * 16)(*(flat + [None] * 4)), pitch, w, h, 12, dh, bpp, bottom_up, r, o.ctypes.data_as(ctypes.c_void_p), 2 * w * bpp)
        ends.append(o)
    lo, hi = ends
    opitch = 2 * w * bpp + 16
    got = np.full((dh, opitch), 7, np.uint8)
    E = emu()
    E.emu_inv_rgb8.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
    E.emu_inv_rgb8((c_i16p * 12)(*flat), pitch, w, h, dh, 12, bpp, bottom_up, 77, got.ctypes.data_as(ctypes.c_void_p), opitch)
    img = got[:, : 2 * w * bpp]
    assert ((img >= lo) & (img <= hi)).all()
    differ = lo != hi
    assert 0.4 < (img[differ] == hi[differ]).mean() < 0.6
    assert (got[:, 2 * w * bpp:] == 7).all()
    if bpp == 4: assert (img[:, 3::4] == 255).all()
    assert (img == 255).any() and (img == 0).any()


@pytest.mark.parametrize("w,rows,matrix,cs", [(16, 3, 0, 2), (130, 7, 2, 1), (338, 5, 1, 6), (64, 4, 3, 5)])
def test_yu64_to_rgb24_lies_in_oracle_interval(w, rows, matrix, cs):
    """k_yu64_to_rgb24 (4:2:2 samples decoded to RG24, second step): every byte between the oracle's conversion with dither 0 and with 32767 (the scalar loop
    of the reference restated, pinned on the reference decoder in test_oracle_vs_ref), both ends about equally often,
    three bytes never sit at opposite ends unless one of them cannot move), all four matrices, full-range words, bottom row first, nothing beside the picture."""
    rng = np.random.default_rng(w + rows)
    yu = rng.integers(0, 65536, size=(rows, 2 * w), dtype=np.int64).astype(np.uint16)
    yu[0, :8] = 65535; yu[0, 8:16] = 0
    O = oracle()
    O.orc_yu64_to_rgb24.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int]
    ends = []
    for d in (0, 32767):
        o = np.zeros((rows, 3 * w), np.uint8)
        O.orc_yu64_to_rgb24(yu.ctypes.data_as(ctypes.c_void_p), 2 * w, w, rows, cs, d, o.ctypes.data_as(ctypes.c_void_p), 3 * w)
        ends.append(o)
    lo, hi = ends
    opitch = 3 * w + 5
    got = np.full((rows, opitch), 7, np.uint8)
    E = emu()
    E.emu_yu64_to_rgb24.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
    E.emu_yu64_to_rgb24(yu.ctypes.data_as(ctypes.c_void_p), 2 * w, w, rows, matrix, 99, got.ctypes.data_as(ctypes.c_void_p), opitch)
    img = got[:, : 3 * w]
    assert ((img >= lo) & (img <= hi)).all()
    differ = lo != hi
    assert 0.35 < (img[differ] == hi[differ]).mean() < 0.65
    assert (got[:, 3 * w:] == 7).all() and (img == 255).any() and (img == 0).any()


@pytest.mark.parametrize("w,h,dh", [(16, 8, 16), (68, 20, 37), (160, 17, 34), (250, 33, 66)])
def test_inv_b64a_of_rgb444_last_level_equals_oracle(w, h, dh):
    """RGB 4:4:4 samples decoded to b64a (k_inv_packed16 with three planes in four-word pixels): the oracle model pinned on the reference decoder in
    test_oracle_vs_ref -- RG48 words with the scalar-tail clamp in the last band column only, constant alpha word 0xfff0; nothing beside the picture."""
    rng = np.random.default_rng(w * 7 + h)
    pitch = (w + 7) // 8 * 8
    bands = []
    for c in range(3):
        bs = [np.zeros((h, pitch), np.int16) for _ in range(4)]
        bs[0][:, :w] = rand_plane(rng, w, h, 14)
        for k in range(1, 4): bs[k][:, :w] = rand_plane(rng, w, h, 11, signed=True)
        bands.append(bs)
    flat = [p16(a) for c in range(3) for a in bands[c]]
    O = oracle()
    O.orc_inv_spatial_to_b64a_of_rgb444.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    full = np.zeros((2 * h, 2 * w * 4), np.uint16)
    O.orc_inv_spatial_to_b64a_of_rgb444((c_i16p * 16)(*(flat + [None] * 4)), pitch, w, h, 12, full.ctypes.data_as(ctypes.c_void_p), 2 * w * 4)
    want = full[:dh]
    assert (want == 65535).any() and (want == 0).any() and (want[:, : 2 * w * 4 - 8] <= 0xfff0).all()
    opitch = 2 * w * 4 + 8
    got = np.full((dh, opitch), 7, np.uint16)
    E = emu()
    E.emu_inv_b64a_of_444.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    E.emu_inv_b64a_of_444((c_i16p * 12)(*flat), pitch, w, h, dh, got.ctypes.data_as(ctypes.c_void_p), opitch)
    assert np.array_equal(got[:, : 2 * w * 4], want)
    assert (got[:, 2 * w * 4:] == 7).all()


@pytest.mark.parametrize("w,h,dh,bottom_up", [(16, 8, 16, 1), (68, 20, 37, 0), (160, 17, 34, 1), (250, 33, 66, 0)])
def test_inv_rgba8_last_level_equals_oracle(w, h, dh, bottom_up):
    """k_inv_packed16's byte mode for RGBA 4:4:4:4 samples (BGRA / BGRa output): no dither -- equal to the oracle model pinned on the reference decoder in
    test_oracle_vs_ref ((12-bit component + 2) >> 4, alpha expanded from the rounded value), both clamps, both row orders, nothing beside the picture."""
    rng = np.random.default_rng(w * 5 + h)
    pitch = (w + 7) // 8 * 8
    bands = []
    for c in range(4):
        bs = [np.zeros((h, pitch), np.int16) for _ in range(4)]
        bs[0][:, :w] = rand_plane(rng, w, h, 14)
        for k in range(1, 4): bs[k][:, :w] = rand_plane(rng, w, h, 11, signed=True)
        bands.append(bs)
    flat = [p16(a) for c in range(4) for a in bands[c]]
    O = oracle()
    O.orc_inv_spatial_to_rgba8.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_int]
    want = np.zeros((dh, 2 * w * 4), np.uint8)
    O.orc_inv_spatial_to_rgba8((c_i16p * 16)(*flat), pitch, w, h, 12, dh, bottom_up, want.ctypes.data_as(ctypes.c_void_p), 2 * w * 4)
    opitch = 2 * w * 4 + 16
    got = np.full((dh, opitch), 7, np.uint8)
    E = emu()
    E.emu_inv_rgba8.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_int]
    E.emu_inv_rgba8((c_i16p * 16)(*flat), pitch, w, h, dh, 12, bottom_up, got.ctypes.data_as(ctypes.c_void_p), opitch)
    assert np.array_equal(got[:, : 2 * w * 4], want)
    assert (got[:, 2 * w * 4:] == 7).all()
    assert (want[:, 3::4] == 255).any() and (want[:, 3::4] == 0).any() and (want[:, 0::4] == 255).any() and (want[:, 0::4] == 0).any()


@pytest.mark.parametrize("w,h,dh,name", [(16, 8, 16, "r210"), (68, 20, 37, "DPX0"), (160, 17, 34, "AB10"), (250, 33, 66, "AR10")])
def test_inv_rgb10_last_level_equals_oracle(w, h, dh, name):
    """k_inv_rgb10 (r210 / DPX0 / AB10 / AR10 output of RGB 4:4:4 samples: one 32-bit word per pixel, big- or little-endian) = oracle model
    pinned on the reference decoder in test_oracle_vs_ref: (value before the final >> 1, + 3) >> 3 per component, clamped at both ends."""
    rng = np.random.default_rng(w * 3 + h)
    pitch = (w + 7) // 8 * 8
    bands = []
    for c in range(3):
        bs = [np.zeros((h, pitch), np.int16) for _ in range(4)]
        bs[0][:, :w] = rand_plane(rng, w, h, 14)
        for k in range(1, 4): bs[k][:, :w] = rand_plane(rng, w, h, 11, signed=True)
        bands.append(bs)
    flat = [p16(a) for c in range(3) for a in bands[c]]
    order, shifts, code = RGB10_FORMATS[name]
    O = oracle()
    O.orc_inv_spatial_to_rgb10.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_int]
    want = np.zeros((dh, 2 * w), np.uint32)
    O.orc_inv_spatial_to_rgb10((c_i16p * 16)(*(flat + [None] * 4)), pitch, w, h, dh, shifts[0], shifts[1], shifts[2], int(order == ">"), want.ctypes.data_as(ctypes.c_void_p), 2 * w)
    got = np.full((dh, 2 * w + 4), 7, np.uint32)
    E = emu()
    E.emu_inv_rgb10.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_int]
    E.emu_inv_rgb10((c_i16p * 12)(*flat), pitch, w, h, dh, shifts[0], shifts[1], shifts[2], int(order == ">"), got.ctypes.data_as(ctypes.c_void_p), 2 * w + 4)
    assert np.array_equal(got[:, : 2 * w], want)
    assert (got[:, 2 * w:] == 7).all()
    words = want.byteswap() if order == ">" else want
    comp = (words >> shifts[1]) & 0x3ff
    assert (comp == 1023).any() and (comp == 0).any()


@pytest.mark.parametrize("w,h,dh", [(40, 8, 8), (300, 24, 21)])
def test_unpack_byr4_equals_oracle(w, h, dh):
    """k_unpack_byr4 (Bayer mosaic -> G, R-G, B-G, G1-G2 planes through the log-90 curve) = oracle restatement of ConvertBYR4ToFrame16s
    (pinned against the reference encoder's samples in test_host_bitstream); rows below the picture repeat the last quad row."""
    rng = np.random.default_rng(w + h)
    mosaic = rng.integers(0, 65536, size=(2 * dh, 2 * w), dtype=np.int64).astype(np.uint16)
    O = oracle()
    curve = np.zeros(1 << 14, np.uint16)
    O.orc_byr4_log90_curve.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    O.orc_byr4_log90_curve(12, 14, curve.ctypes.data_as(ctypes.c_void_p))
    assert curve[-1] == 4094 and curve[1] > 0 and np.all(np.diff(curve.astype(np.int32)) >= 0)
    O.orc_byr4_unpack_row.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 5
    pitch = (w + 15) // 16 * 16
    want = [np.zeros((h, pitch), np.int16) for _ in range(4)]
    for r in range(h):
        sr = min(r, dh - 1)
        O.orc_byr4_unpack_row(mosaic[2 * sr].ctypes.data_as(ctypes.c_void_p), mosaic[2 * sr + 1].ctypes.data_as(ctypes.c_void_p), w, 12, 14,
                              curve.ctypes.data_as(ctypes.c_void_p), *[p[r].ctypes.data_as(ctypes.c_void_p) for p in want])
    got = [np.zeros((h, pitch), np.int16) for _ in range(4)]
    E = emu()
    E.emu_unpack_byr4.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    E.emu_unpack_byr4(mosaic.ctypes.data_as(ctypes.c_void_p), 2 * w, w, h, dh, curve.ctypes.data_as(ctypes.c_void_p), 0, 12, (c_i16p * 4)(*[p16(g) for g in got]), pitch)
    for c in range(4):
        assert np.array_equal(got[c], want[c]), c


@pytest.mark.parametrize("w,h,dh,packed12", [(40, 8, 8, 0), (304, 24, 21, 0), (48, 16, 13, 1), (304, 24, 24, 1)])
def test_fwd_bayer_level1_without_planes(w, h, dh, packed12):
    """Bayer level 1 straight from the mosaic (k_fwd_packed16, loader layouts 10 / 11: what the product runs) = the oracle's unpack (BYR4 through the log-90
    curve, BYR5 without) + the oracle's plane transform of the four component planes; rows below the picture repeat the last row pair."""
    rng = np.random.default_rng(w + h + packed12)
    O = oracle()
    pitch = (w // 2 + 7) // 8 * 8
    planes = [np.zeros((h, w), np.int16) for _ in range(4)]
    curve = np.zeros(1 << 14, np.uint16)
    if packed12:
        frame = rng.integers(0, 256, size=(dh, 6 * w), dtype=np.int64).astype(np.uint8)
        O.orc_byr5_unpack_row.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
        for r in range(h):
            O.orc_byr5_unpack_row(frame[min(r, dh - 1)].ctypes.data_as(ctypes.c_void_p), w, *[p[r].ctypes.data_as(ctypes.c_void_p) for p in planes])
        in_pitch = 3 * w
    else:
        frame = rng.integers(0, 65536, size=(2 * dh, 2 * w), dtype=np.int64).astype(np.uint16)
        O.orc_byr4_log90_curve.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        O.orc_byr4_log90_curve(12, 14, curve.ctypes.data_as(ctypes.c_void_p))
        O.orc_byr4_unpack_row.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 5
        for r in range(h):
            sr = min(r, dh - 1)
            O.orc_byr4_unpack_row(frame[2 * sr].ctypes.data_as(ctypes.c_void_p), frame[2 * sr + 1].ctypes.data_as(ctypes.c_void_p), w, 12, 14,
                                  curve.ctypes.data_as(ctypes.c_void_p), *[p[r].ctypes.data_as(ctypes.c_void_p) for p in planes])
        in_pitch = 2 * w
    quant = [1, 24, 24, 12, 1, 36, 36, 18, 1, 36, 36, 18, 1, 48, 48, 24]
    outs = [np.zeros((h // 2, pitch), np.int16) for _ in range(16)]
    ptrs = (c_i16p * 16)(*[p16(o) for o in outs])
    E = emu()
    E.emu_fwd_bayer.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    E.emu_fwd_bayer(frame.ctypes.data_as(ctypes.c_void_p), in_pitch, w, h, dh, packed12, curve.ctypes.data_as(ctypes.c_void_p), 0, iarr(quant), 2, ptrs, pitch)
    for c in range(4):
        want = [np.zeros((h // 2, pitch), np.int16) for _ in range(4)]
        bands = (c_i16p * 4)(*[p16(o) for o in want])
        O.orc_fwd_spatial(p16(planes[c]), w, w, h, 0, iarr(quant[4 * c: 4 * c + 4]), 2, bands, pitch)
        for b in range(4):
            assert np.array_equal(outs[4 * c + b][:, : w // 2], want[b][:, : w // 2]), (c, b)


@pytest.mark.parametrize("w,h,dh", [(40, 8, 8), (300, 24, 21)])
def test_unpack_byr5_equals_oracle(w, h, dh):
    """k_unpack_byr4 on BYR5 input (12-bit samples: runs of high bytes, then low nibbles; no curve) = oracle restatement of ConvertBYR5ToFrame16s (pinned
    against the reference encoder's samples in test_host_bitstream); all 4096 values of every component; rows below the picture repeat the last row pair."""
    rng = np.random.default_rng(w + h)
    frame = rng.integers(0, 256, size=(dh, 6 * w), dtype=np.int64).astype(np.uint8)
    O = oracle()
    O.orc_byr5_unpack_row.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
    pitch = (w + 15) // 16 * 16
    want = [np.zeros((h, pitch), np.int16) for _ in range(4)]
    for r in range(h):
        O.orc_byr5_unpack_row(frame[min(r, dh - 1)].ctypes.data_as(ctypes.c_void_p), w, *[p[r].ctypes.data_as(ctypes.c_void_p) for p in want])
    got = [np.zeros((h, pitch), np.int16) for _ in range(4)]
    E = emu()
    E.emu_unpack_byr5.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    E.emu_unpack_byr5(frame.ctypes.data_as(ctypes.c_void_p), w, h, dh, 0, (c_i16p * 4)(*[p16(g) for g in got]), pitch)
    for c in range(4):
        assert np.array_equal(got[c], want[c]), c
    assert want[0].max() > 3900 and want[3].min() < 300


@pytest.mark.parametrize("w,h,dh", [(64, 16, 16), (272, 24, 21), (720, 16, 16), (1920, 8, 8)])
@pytest.mark.parametrize("uyvy", [0, 1])
def test_fwd_frame_yuv422_interlaced_level1(w, h, dh, uyvy):
    """k_fwd_frame_yuv422 = oracle restatement of TransformForwardFrameYUV (pinned on reference coefficients in test_oracle_vs_ref and,
    through the host writer, on whole reference samples), incl. full-range field differences that quantize beyond the peak level."""
    rng = np.random.default_rng(w + h + uyvy)
    frame = rng.integers(0, 256, size=(dh, w * 2), dtype=np.int64).astype(np.uint8)
    frame[0::2, : w] = rng.choice([0, 255], size=(len(frame[0::2]), w))          # strong field flicker on the left half
    quant = [1, 36, 16, 36, 1, 36, 16, 48, 1, 36, 16, 48]
    outs_o, outs_e, pitches = [], [], []
    for c in range(3):
        cw = (w if c == 0 else w // 2) // 2
        pitch = (cw + 7) // 8 * 8; pitches.append(pitch)
        outs_o.append([np.zeros((h // 2, pitch), np.int16) for _ in range(4)])
        outs_e.append([np.zeros((h // 2, pitch), np.int16) for _ in range(4)])
    padded = np.full((h, w * 2), 0x80, np.uint8); padded[:dh] = frame
    O = oracle()
    for c in range(3):
        bands = (c_i16p * 4)(*[p16(a) for a in outs_o[c]])
        O.orc_fwd_frame_yuv422(p8(padded), w * 2, w if c == 0 else w // 2, h, c, 2, uyvy, iarr(quant[4 * c: 4 * c + 4]), 2, bands, pitches[c])
    ptrs = (c_i16p * 12)(*[p16(a) for c in range(3) for a in outs_e[c]])
    emu().emu_fwd_frame_yuv422(p8(frame), w * 2, w, h, dh, uyvy, 2, iarr(quant), 2, ptrs, iarr(pitches))
    for c in range(3):
        cw = (w if c == 0 else w // 2) // 2
        for b in range(4):
            assert np.array_equal(outs_e[c][b][:, :cw], outs_o[c][b][:, :cw]), (c, b)


# ---------------------------------------------------------------------------------------------------------------
# Round-2 decoder (cfhd_dec_kernels.h: k_dec_plan / k_dec_index / k_dec_chain / k_dec_tiles), same kernel source under emulation
# ---------------------------------------------------------------------------------------------------------------
DX_ARRANGEMENT = 0      # bits added to the mode of emu_entropy_decode_dx; set per test by the `arrangement` fixture


@pytest.fixture(params=[0, 16, 24], ids=["index+tiles", "one-wave-workgroups", "one-wave-workgroups-full-tiles"])
def arrangement(request):
    """The arrangements of the chunk-indexed decoder that run every test of this section: k_dec_tiles with the product's four waves per workgroup and tiles of 1536 coefficients, with
    one wave per workgroup (a tile then takes several rounds of 64 pieces), and that with tiles of the product's size (hundreds of pieces per tile in the dense bands).  (One decoder since round 6: the single-pass experiment of round 5 -- index walk that
    logs its steps + a scatter pass -- was measured four times, never adopted and has left the tree; profiles/r05_a..e_*.)"""
    global DX_ARRANGEMENT
    DX_ARRANGEMENT = request.param
    yield request.param
    DX_ARRANGEMENT = 0


def _dx_decode(sample, plan, mode, grid, size=None, guard=0):
    mode |= DX_ARRANGEMENT
    E = emu()
    E.emu_entropy_decode_dx.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_int, c_i16p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
    got = np.full(plan.coeff_elems + guard, 99, dtype=np.int16)
    s = np.frombuffer(sample, dtype=np.uint8).copy() if not isinstance(sample, np.ndarray) else sample
    rc = E.emu_entropy_decode_dx(p8(s), size if size is not None else len(s), 1, p16(got), plan.coeff_elems, mode, grid)
    return rc, got


@pytest.mark.parametrize("mode,grid", [(0, 3), (1, 2), (2, 1), (0, 64), (8, 2), (10, 3)])      # + 8: tiles of the product's size (else 1536 coefficients: bands of many tiles)
@pytest.mark.parametrize("w,h,seed", [(192, 96, 1), (336, 252, 3), (720, 480, 4)])
def test_dx_decoder_emulated_equals_host_decoder(w, h, seed, mode, grid, arrangement):
    """The chunk-indexed decoder reproduces the oracle's decoder (oracle_decode_pyramid: its own sample walk and bit-serial decoder) coefficient for coefficient, every element of every band incl.
    its pitch padding written by the tile kernel itself.  mode 1 switches the run-in speculation off, so every chunk but a band's first
    assumes a wrong start and k_dec_chain has to repair it; mode 2 parses two copies of the sample with k_dec_parse / k_dec_plan."""
    frame, pitch = synth_yuy2(w, h, seed)
    if seed == 4:                                   # busy picture: long payloads, several chunks per band
        rng = np.random.default_rng(9)
        f = frame.reshape(h, pitch).astype(np.int32) + rng.integers(-30, 31, (h, pitch))
        frame = np.clip(f, 0, 255).astype(np.uint8).reshape(-1).copy()
    plan = Plan(w, h)
    coeffs = oracle_forward_yuv422(plan, frame, pitch)
    if seed == 3:                                   # long code words: values up to the +-1023 clamp
        v = plan.view(coeffs, 0, 0, 1); v[::7, ::5] = 1023; v[1::9, 2::11] = -1023; v[3::5, 1::13] = 300
    sample = product_write_sample_host(plan, coeffs, 1, meta_global=b"GUID\x10\x00\x00G" + bytes(16))
    want = oracle_decode_pyramid(sample, plan)
    rc, got = _dx_decode(sample, plan, mode, grid)
    assert rc == 0
    for (c, lv, b) in plan.band:
        if b == 0 and lv != 2: continue
        cols = plan.band[(c, lv, b)]["width"] if b == 0 else None
        assert np.array_equal(plan.view(got, c, lv, b)[:, :cols], plan.view(want, c, lv, b)[:, :cols]), (c, lv, b)


def test_dx_decoder_emulated_sparse_and_dense_bands(arrangement):
    """Extremes of the code: a band that is one single zero run (the tiles behind the first are never touched by a code word), a band with a
    value in every position (three bits per coefficient and more: many pieces per tile), values at the very first and very last position."""
    w, h = 336, 252
    plan = Plan(w, h)
    frame, pitch = synth_yuy2(w, h, 8)
    coeffs = oracle_forward_yuv422(plan, frame, pitch)
    rng = np.random.default_rng(2)
    plan.view(coeffs, 0, 0, 1)[:] = 0                                           # all zero (pitch padding too)
    d = plan.band[(0, 0, 2)]
    dense = plan.view(coeffs, 0, 0, 2); dense[:] = 0
    dense[:, : d["width"]] = rng.integers(1, 40, (d["height"], d["width"])) * rng.choice([-1, 1], (d["height"], d["width"]))
    e = plan.view(coeffs, 0, 0, 3); e[:] = 0; e[0, 0] = -5; e[-1, plan.band[(0, 0, 3)]["width"] - 1] = 7
    sample = product_write_sample_host(plan, coeffs, 1, meta_global=b"GUID\x10\x00\x00G" + bytes(16))
    want = oracle_decode_pyramid(sample, plan)
    for mode, grid in ((0, 5), (1, 1)):
        rc, got = _dx_decode(sample, plan, mode, grid)
        assert rc == 0
        for (c, lv, b) in plan.band:
            if b == 0: continue
            assert np.array_equal(plan.view(got, c, lv, b), plan.view(want, c, lv, b)), (mode, c, lv, b)


def test_dx_decoder_emulated_code_without_unique_alignment(arrangement):
    """A smooth gradient gives a band the same value in every position: the same code word over and over, a bit pattern that parses
    consistently at several alignments, so no lane can find its phase on its own and the true starts travel through a chunk lane by lane
    (and from chunk to chunk through k_dec_chain's repair path).  The result must still be exact."""
    w, h = 720, 480
    plan = Plan(w, h)
    frame, pitch = synth_yuy2(w, h, 8)
    coeffs = oracle_forward_yuv422(plan, frame, pitch)
    for (c, lv, b), val in (((0, 0, 2), 3), ((1, 0, 1), -7), ((0, 1, 2), 21)):
        d = plan.band[(c, lv, b)]
        v = plan.view(coeffs, c, lv, b); v[:] = 0; v[:, : d["width"]] = val
    sample = product_write_sample_host(plan, coeffs, 1, meta_global=b"GUID\x10\x00\x00G" + bytes(16))
    want = oracle_decode_pyramid(sample, plan)
    for mode, grid in ((0, 4), (1, 3)):
        rc, got = _dx_decode(sample, plan, mode, grid)
        assert rc == 0
        for (c, lv, b) in plan.band:
            if b == 0: continue
            assert np.array_equal(plan.view(got, c, lv, b), plan.view(want, c, lv, b)), (mode, c, lv, b)
    E = emu(); E.emu_dx_stats.restype = ctypes.POINTER(ctypes.c_uint32)
    rc, got = _dx_decode(sample, plan, 0, 4)
    st = E.emu_dx_stats()
    assert st[2] >= 20, "the constant bands were expected to need many rounds (%d): the test does not exercise what it is for" % st[2]
    # chunks inside such a stretch are indexed for every candidate start and the chain picks the true one: none is left to the serial repair
    assert (st[3] >> 16) > 0 and st[13] > 0 and st[12] == 0, "candidates %d, re-indexed %d, bands repaired serially %d" % (st[3] >> 16, st[13], st[12])


@pytest.mark.parametrize("mode", [0, 2])
def test_dx_decoder_emulated_survives_damaged_samples(mode, arrangement):
    """Truncated samples are refused; garbage inside the code words never writes outside the pyramid nor hangs (error flag or wrong values, no crash)."""
    w, h = 336, 252
    frame, pitch = synth_yuy2(w, h, 5)
    plan = Plan(w, h)
    coeffs = oracle_forward_yuv422(plan, frame, pitch)
    sample = product_write_sample_host(plan, coeffs, 1, meta_global=b"GUID\x10\x00\x00G" + bytes(16))
    s = np.frombuffer(sample, dtype=np.uint8).copy()
    rc, _ = _dx_decode(s, plan, mode, 2, size=len(sample) // 2 & ~3)
    assert rc < 0
    rng = np.random.default_rng(11)
    flagged = 0
    for trial in range(6):
        t = s.copy()
        lo = len(t) // 3 + trial * 1000
        t[lo: lo + 600] = rng.integers(0, 256, 600, dtype=np.uint8)
        rc, got = _dx_decode(t, plan, mode, 2, guard=4096)
        flagged += rc != 0
        assert np.all(got[plan.coeff_elems:] == 99)
    assert flagged >= 1


# ---------------------------------------------------------------------------------------------------------------
# Interlaced samples: code set 18 + difference coding + peak tables (k_dec_parse / k_dec_tiles / k_dec_undiff), inverse frame transform
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,seed,peaks", [(192, 96, 1, 0), (336, 252, 3, 1), (720, 480, 4, 1)])
def test_dx_decoder_emulated_interlaced_samples(w, h, seed, peaks, arrangement):
    """The field-difference band of every channel arrives in the second code set, difference coded along the row and -- when a value lies
    beyond the peak threshold -- with its large values in a peak table behind the band.  The emulated kernels must rebuild exactly the
    pyramid of the oracle's decoder (code set 18, peak tables, running sums restated in oracle/cfhd_oracle_ent.c; tied to the product's host decoder in test_oracle_vs_ref)."""
    frame, pitch = field_flicker_frame(w, h) if peaks else synth_yuy2(w, h, seed)       # field flicker: quantized steps beyond +-250 in the difference band
    if peaks:
        rng = np.random.default_rng(seed)
        frame = np.clip(frame.astype(np.int32) + rng.integers(-6, 7, frame.shape), 0, 255).astype(np.uint8)
    plan = Plan(w, h, progressive=0)
    coeffs = oracle_forward_interlaced_yuv422(plan, frame, pitch)
    sample = product_write_sample_host(plan, coeffs, 1, meta_global=b"GUID\x10\x00\x00G" + bytes(16), progressive=0)
    levels = [int.from_bytes(sample[i + 2:i + 4], "big") for i in range(0, len(sample) - 4, 4) if sample[i:i + 2] == b"\xff\xb6"]      # TAG_PEAK_LEVEL (optional)
    assert any(levels) == bool(peaks)
    want = oracle_decode_pyramid(sample, plan)
    for mode, grid in ((0, 3), (2, 2)):
        rc, got = _dx_decode(sample, plan, mode, grid)
        assert rc == 0, (mode, rc)
        for (c, lv, b) in plan.band:
            if b == 0 and lv != 2: continue
            cols = plan.band[(c, lv, b)]["width"]
            assert np.array_equal(plan.view(got, c, lv, b)[:, :cols], plan.view(want, c, lv, b)[:, :cols]), (mode, c, lv, b)


@pytest.mark.parametrize("w,h,dh", [(32, 8, 16), (96, 20, 40), (360, 30, 58), (128, 17, 34), (132, 33, 66), (260, 19, 37), (960, 6, 12), (1032, 4, 8), (8, 4, 8)])
@pytest.mark.parametrize("uyvy", [0, 1])
def test_inv_frame_yuv422(w, h, dh, uyvy):
    """Emulated inverse frame transform vs the oracle (itself pinned against the reference decoder): every output byte must equal the
    oracle's with dither 0 or with dither 1, and rows beyond the display height stay untouched."""
    rng = np.random.default_rng(w + h + uyvy)
    bands, pitches = [], []
    for ch in range(3):
        cw = w if ch == 0 else w // 2
        pitch = (cw + 7) // 8 * 8 + (8 if ch == 2 else 0); pitches.append(pitch)
        bs = [rng.integers(-50, 50, size=(h, pitch)).astype(np.int16) for _ in range(4)]
        bs[0][:, :cw] = rand_plane(rng, cw, h, 11)
        for k in range(1, 4): bs[k][:, :cw] = rand_plane(rng, cw, h, 9, signed=True)
        bs[2][:, :cw] *= 3                                      # temporal highpass beyond the lowpass: low - high below zero, low + high above the range
        bands.append(bs)
    ptrs = (c_i16p * 12)(*[p16(a) for ch in range(3) for a in bands[ch]])
    outs = []
    for dither in (0, 1):
        o = np.zeros((2 * h, 4 * w), np.uint8)
        oracle().orc_inv_frame_to_yuv422(ptrs, iarr(pitches), w, h, 10, uyvy, dither, p8(o), 4 * w)
        outs.append(o[:dh])
    e = np.full((2 * h, 4 * w + 16), 7, np.uint8)
    emu().emu_inv_frame_yuv422(ptrs, iarr(pitches), w, h, dh, uyvy, 2, 1234, p8(e), 4 * w + 16)
    assert np.all(e[:, 4 * w:] == 7) and np.all(e[dh:] == 7)
    # four band columns per thread (8-byte loads, neighbours from the adjacent lanes, 16-byte stores): byte for byte the same picture
    q = np.full((2 * h, 4 * w + 16), 7, np.uint8)
    assert emu().emu_inv_frame_yuv422_quad(ptrs, iarr(pitches), w, h, dh, uyvy, 2, 1234, p8(q), 4 * w + 16) == 0
    assert np.array_equal(q, e)
    # the register-strip kernel (what large launches take; luma band widths that are multiples of 16): the same picture too -- same arithmetic, and since round 6 the same
    # dither bit for every sample (dither422_block / dither422_column)
    E = emu()
    E.emu_inv_frame_yuv422_strip.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_uint, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    st = np.full((2 * h, 4 * w + 16), 7, np.uint8)
    rc = E.emu_inv_frame_yuv422_strip(ptrs, iarr(pitches), w, h, dh, uyvy, 2, 1234, p8(st), 4 * w + 16, None, None)
    if w % 16 == 0:
        assert rc == 0 and np.array_equal(st, e)
    e = e[:dh, :4 * w]
    assert np.all((e == outs[0]) | (e == outs[1]))
    assert np.any(e != outs[0]) and np.any(e != outs[1])


def test_gpu_entropy_stage_emulated_long_trailers():
    """Empty bands of a larger frame: the trailer of a band is dozens of copies of the longest run code, more than one piece of k_ent_layout's
    fill (the emulated build uses pieces of 16 words), written in closed form by the lanes; a single value in the middle of one band splits
    its zeros into a run for k_ent_emit and a trailer."""
    w, h = 1280, 720
    plan = Plan(w, h)
    meta = b"GUID\x10\x00\x00G" + bytes(16)
    coeffs = np.zeros(plan.coeff_elems, dtype=np.int16)
    for c in range(3):
        d = plan.band[(c, 2, 0)]
        plan.view(coeffs, c, 2, 0)[:, : d["width"]] = 4000
    d = plan.band[(0, 0, 1)]
    plan.view(coeffs, 0, 0, 1)[d["height"] // 3, 7] = -9
    want = product_write_sample_host(plan, coeffs, 1, meta_global=meta)
    got = _emu_entropy(plan, coeffs, 1, meta)
    assert got == want


@pytest.mark.parametrize("w,h,dh", [(32, 16, 16), (144, 40, 37), (336, 24, 24)])
def test_fwd_packed16_level1_of_yu64(w, h, dh):
    """YU64 input (16-bit words Y0 C1 Y1 C2, each >> 6 to 10 bits; Codec/frame.c:1556): k_fwd_packed16 with a first word, a stride and a
    width per channel = the oracle's plane transform of the three unpacked planes (rows below the picture repeat the last row)."""
    rng = np.random.default_rng(w + h)
    words = rng.integers(0, 65536, size=(dh, 2 * w), dtype=np.int64).astype(np.uint16)
    quant = [1, 24, 24, 12] * 3
    pitches = [(cw // 2 + 7) // 8 * 8 for cw in (w, w // 2, w // 2)]
    outs = [np.zeros((h // 2, pitches[c]), np.int16) for c in range(3) for _ in range(4)]
    ptrs = (c_i16p * 12)(*[p16(o) for o in outs])
    E = emu()
    E.emu_fwd_yu64.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    E.emu_fwd_yu64(words.ctypes.data_as(ctypes.c_void_p), 2 * w, w, h, dh, iarr(quant), 2, ptrs, iarr(pitches))
    planes = [words[:, 0::2] >> 6, words[:, 1::4] >> 6, words[:, 3::4] >> 6]
    for c in range(3):
        cw = w if c == 0 else w // 2
        plane = np.zeros((h, cw), np.int16); plane[:dh] = planes[c].astype(np.int16); plane[dh:] = plane[dh - 1]
        want = [np.zeros((h // 2, pitches[c]), np.int16) for _ in range(4)]
        bands = (c_i16p * 4)(*[p16(o) for o in want])
        oracle().orc_fwd_spatial(p16(plane), cw, cw, h, 0, iarr(quant[:4]), 2, bands, pitches[c])
        for b in range(4):
            assert np.array_equal(outs[4 * c + b][:, :cw // 2], want[b][:, :cw // 2]), (c, b)


@pytest.mark.parametrize("w,h,dh,wpp,cs", [(32, 16, 16, 3, 0), (144, 40, 37, 4, 0), (336, 24, 24, 3, 1), (208, 16, 13, 4, 2), (64, 8, 8, 3, 3)])
def test_fwd_packed16_level1_of_deep_rgb_as_yuv422(w, h, dh, wpp, cs):
    """RG48 / b64a encoded as YUV 4:2:2: the loader of k_fwd_packed16 converts the pixels on the way in (layout 7) = the oracle's conversion
    (pinned on reference samples in test_host_bitstream) + the oracle's plane transform of the three planes; all four matrices, clamps at
    both ends of the 10-bit range (saturated primaries), rows below the picture repeating the last one."""
    rng = np.random.default_rng(w + h + wpp)
    px = rng.integers(0, 65536, size=(dh, w, wpp), dtype=np.int64).astype(np.uint16)
    r_word = 0 if wpp == 3 else 1
    px[::3, ::4, r_word:r_word + 3] = [65535, 0, 0]; px[1::3, 1::4, r_word:r_word + 3] = [0, 0, 65535]; px[2::5, 2::6, r_word:r_word + 3] = [65535, 65535, 65535]; px[::7, 3::5, r_word:r_word + 3] = 0
    words = px.reshape(dh, w * wpp)
    quant = [1, 24, 24, 12] * 3
    pitches = [(cw // 2 + 7) // 8 * 8 for cw in (w, w // 2, w // 2)]
    outs = [np.zeros((h // 2, pitches[c]), np.int16) for c in range(3) for _ in range(4)]
    ptrs = (c_i16p * 12)(*[p16(o) for o in outs])
    E = emu()
    E.emu_fwd_rgb16_to_yuv422.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    E.emu_fwd_rgb16_to_yuv422(words.ctypes.data_as(ctypes.c_void_p), w * wpp, wpp, r_word, w, h, dh, cs, iarr(quant), 2, ptrs, iarr(pitches))
    planes = oracle_rgb16_to_yuv422_planes(words, wpp, r_word, w, dh, cs)
    assert planes[0].max() == 1023 or cs in (0, 2)
    for c in range(3):
        cw = w if c == 0 else w // 2
        plane = np.zeros((h, cw), np.int16); plane[:dh] = planes[c]; plane[dh:] = plane[dh - 1]
        want = [np.zeros((h // 2, pitches[c]), np.int16) for _ in range(4)]
        bands = (c_i16p * 4)(*[p16(o) for o in want])
        oracle().orc_fwd_spatial(p16(plane), cw, cw, h, 0, iarr(quant[:4]), 2, bands, pitches[c])
        for b in range(4):
            assert np.array_equal(outs[4 * c + b][:, :cw // 2], want[b][:, :cw // 2]), (c, b)


@pytest.mark.parametrize("w,h,dh,bpp,top_down,cs", [(32, 16, 16, 3, 0, 0), (144, 40, 37, 4, 0, 0), (336, 24, 24, 4, 1, 1), (208, 16, 13, 3, 0, 2), (64, 8, 8, 4, 1, 3)])
def test_fwd_packed16_level1_of_rgb8_as_yuv422(w, h, dh, bpp, top_down, cs):
    """RG24 / BGRA / BGRa encoded as YUV 4:2:2: the loader of k_fwd_packed16 converts the pixels on the way in (layout 8 / 9) = the oracle's
    conversion (pinned on reference samples in test_host_bitstream) + the oracle's plane transform of the three planes; all four matrices,
    saturated primaries, both row orders, the constant rows (Y 64, chroma 512) below the picture."""
    rng = np.random.default_rng(w + h + bpp)
    px = rng.integers(0, 256, size=(dh, w, bpp), dtype=np.int64).astype(np.uint8)
    px[::3, ::4, :3] = [255, 0, 0]; px[1::3, 1::4, :3] = [0, 0, 255]; px[2::5, 2::6, :3] = [255, 255, 255]; px[::7, 3::5, :3] = 0; px[1::4, 2::7, :3] = [0, 255, 0]
    frame = np.ascontiguousarray(px.reshape(dh, w * bpp))
    quant = [1, 24, 24, 12] * 3
    pitches = [(cw // 2 + 7) // 8 * 8 for cw in (w, w // 2, w // 2)]
    outs = [np.zeros((h // 2, pitches[c]), np.int16) for c in range(3) for _ in range(4)]
    ptrs = (c_i16p * 12)(*[p16(o) for o in outs])
    E = emu()
    E.emu_fwd_rgb8_to_yuv422.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    E.emu_fwd_rgb8_to_yuv422(frame.ctypes.data_as(ctypes.c_void_p), w * bpp, bpp, top_down, w, h, dh, cs, iarr(quant), 2, ptrs, iarr(pitches))
    planes = oracle_rgb8_to_yuv422_planes(frame, w * bpp, bpp, top_down, w, dh, h, cs)
    assert planes[1].max() > 850 and planes[2].min() < 200 and planes[0].max() >= 930
    for c in range(3):
        cw = w if c == 0 else w // 2
        want = [np.zeros((h // 2, pitches[c]), np.int16) for _ in range(4)]
        bands = (c_i16p * 4)(*[p16(o) for o in want])
        oracle().orc_fwd_spatial(p16(np.ascontiguousarray(planes[c])), cw, cw, h, 0, iarr(quant[:4]), 2, bands, pitches[c])
        for b in range(4):
            assert np.array_equal(outs[4 * c + b][:, :cw // 2], want[b][:, :cw // 2]), (c, b)


@pytest.mark.parametrize("w,h,dh", [(48, 16, 16), (144, 40, 37), (320, 24, 24), (400, 16, 13)])
def test_fwd_packed16_level1_of_v210(w, h, dh):
    """v210 input: the loader of k_fwd_packed16 picks the 10-bit fields out of the 32-bit words = the oracle's plane transform of the planes
    the reference's unpack produces (v210_planes: zero rows below the picture, the repeated Cr of the scalar tail)."""
    frame, pitch, Y, Cb, Cr = synth_v210(w, dh, w + h)
    plan_like = type("P", (), {"band": {(0, 0, 0): {"height": h // 2}}})()
    planes = v210_planes(plan_like, Y, Cb, Cr)
    quant = [1, 24, 24, 12] * 3
    pitches = [(cw // 2 + 7) // 8 * 8 for cw in (w, w // 2, w // 2)]
    outs = [np.zeros((h // 2, pitches[c]), np.int16) for c in range(3) for _ in range(4)]
    ptrs = (c_i16p * 12)(*[p16(o) for o in outs])
    E = emu()
    E.emu_fwd_v210.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    E.emu_fwd_v210(frame.ctypes.data_as(ctypes.c_void_p), pitch, w, h, dh, iarr(quant), 2, ptrs, iarr(pitches))
    for c in range(3):
        cw = w if c == 0 else w // 2
        plane = np.ascontiguousarray(planes[c][:h])
        want = [np.zeros((h // 2, pitches[c]), np.int16) for _ in range(4)]
        bands = (c_i16p * 4)(*[p16(o) for o in want])
        oracle().orc_fwd_spatial(p16(plane), cw, cw, h, 0, iarr(quant[:4]), 2, bands, pitches[c])
        for b in range(4):
            assert np.array_equal(outs[4 * c + b][:, :cw // 2], want[b][:, :cw // 2]), (c, b)


@pytest.mark.parametrize("interlaced", [0, 1])
def test_dx_decoder_emulated_fuzzed_samples(interlaced, arrangement):
    """Random damage anywhere in a sample -- headers, size fields, peak table tags, code words, single bit flips, truncation: the kernels
    (device parser included) either flag an error or decode something, never write outside the pyramid and never hang."""
    w, h = 192, 96
    plan = Plan(w, h, progressive=0 if interlaced else 1)
    if interlaced:
        frame, pitch = field_flicker_frame(w, h)
        coeffs = oracle_forward_interlaced_yuv422(plan, frame, pitch)
    else:
        frame, pitch = synth_yuy2(w, h, 5)
        coeffs = oracle_forward_yuv422(plan, frame, pitch)
    sample = product_write_sample_host(plan, coeffs, 1, meta_global=b"GUID\x10\x00\x00G" + bytes(16), progressive=0 if interlaced else 1)
    s = np.frombuffer(sample, dtype=np.uint8).copy()
    rng = np.random.default_rng(23 + interlaced)
    outcomes = {0: 0}
    for trial in range(24):
        t = s.copy()
        kind = trial % 4
        if kind == 0:                                   # a burst of garbage at a random place (the header area in every other trial of this kind)
            lo = int(rng.integers(0, 600)) if trial % 8 == 0 else int(rng.integers(0, len(t) - 64))
            n = int(rng.integers(1, 64)); t[lo: lo + n] = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1:                                 # single bit flips
            for _ in range(int(rng.integers(1, 6))):
                i = int(rng.integers(0, len(t))); t[i] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 2:                                 # a size / tag word overwritten with a large value
            i = int(rng.integers(0, len(t) // 4)) * 4; t[i: i + 4] = [0x20 | int(rng.integers(0, 32)), int(rng.integers(0, 256)), 0xff, 0xff]
        size = len(t) if kind != 3 else int(rng.integers(16, len(t))) & ~3
        for mode in (0, 2):
            rc, got = _dx_decode(t, plan, mode, 2, size=size, guard=4096)
            outcomes[rc != 0] = outcomes.get(rc != 0, 0) + 1
            assert np.all(got[plan.coeff_elems:] == 99), (trial, mode)
    assert outcomes.get(True, 0) >= 4                  # (some damage must have been noticed)


@pytest.mark.parametrize("w,h,dh", [(32, 16, 16), (136, 40, 37), (320, 48, 48)])
def test_fwd_packed16_level1_of_rg24(w, h, dh):
    """RG24 input (8-bit B, G, R, bottom row first; Codec/frame.c:6173): the loader of k_fwd_packed16 lifts the bytes to 12 bits and flips the
    rows = the oracle's plane transform of the planes G, R, B (rows below the picture zero)."""
    rng = np.random.default_rng(w + h)
    pitch = w * 3 + 5
    buf = rng.integers(0, 256, size=(dh, pitch), dtype=np.int64).astype(np.uint8)
    quant = [1, 12, 12, 24] * 3
    opitch = (w // 2 + 7) // 8 * 8
    outs = [np.zeros((h // 2, opitch), np.int16) for _ in range(12)]
    ptrs = (c_i16p * 12)(*[p16(o) for o in outs])
    E = emu()
    E.emu_fwd_rg24.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    E.emu_fwd_rg24(buf.ctypes.data_as(ctypes.c_void_p), pitch, w, h, dh, iarr(quant), 2, ptrs, opitch)
    px = buf[:, : w * 3].reshape(dh, w, 3)[::-1]
    for c, byte in enumerate((1, 2, 0)):
        plane = np.zeros((h, w), np.int16); plane[:dh] = px[:, :, byte].astype(np.int16) << 4
        want = [np.zeros((h // 2, opitch), np.int16) for _ in range(4)]
        bands = (c_i16p * 4)(*[p16(o) for o in want])
        oracle().orc_fwd_spatial(p16(plane), w, w, h, 0, iarr(quant[:4]), 2, bands, opitch)
        for b in range(4):
            assert np.array_equal(outs[4 * c + b][:, :w // 2], want[b][:, :w // 2]), (c, b)


@pytest.mark.parametrize("w,h,dh,top_down", [(32, 16, 16, 0), (136, 40, 37, 1), (320, 24, 24, 0)])
def test_fwd_packed16_level1_of_rgba8(w, h, dh, top_down):
    """BGRA / BGRa encoded as RGBA 4:4:4:4: planes G, R, B = byte << 4, alpha = byte << 4 curved inside the open interval (0, 255 << 4) (frame.c:6415
    ConvertRGBAtoRGBA64; every alpha byte value occurs), zero rows below the picture."""
    rng = np.random.default_rng(w + h)
    px = rng.integers(0, 256, size=(dh, w, 4), dtype=np.int64).astype(np.uint8)
    px.reshape(-1, 4)[:256, 3] = np.arange(256)
    buf = np.ascontiguousarray(px.reshape(dh, w * 4))
    quant = [1, 12, 12, 24] * 4
    opitch = (w // 2 + 7) // 8 * 8
    outs = [np.zeros((h // 2, opitch), np.int16) for _ in range(16)]
    ptrs = (c_i16p * 16)(*[p16(o) for o in outs])
    E = emu()
    E.emu_fwd_rgba8.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    E.emu_fwd_rgba8(buf.ctypes.data_as(ctypes.c_void_p), w * 4, top_down, w, h, dh, iarr(quant), 2, ptrs, opitch)
    src = px if top_down else px[::-1]
    for c, byte in enumerate((1, 2, 0, 3)):
        v = src[:, :, byte].astype(np.int32) << 4
        if c == 3: v = np.where((v > 0) & (v < 4080), ((v * 223 + 128) >> 8) + 256, v)
        plane = np.zeros((h, w), np.int16); plane[:dh] = v.astype(np.int16)
        want = [np.zeros((h // 2, opitch), np.int16) for _ in range(4)]
        bands = (c_i16p * 4)(*[p16(o) for o in want])
        oracle().orc_fwd_spatial(p16(plane), w, w, h, 0, iarr(quant[:4]), 2, bands, opitch)
        for b in range(4):
            assert np.array_equal(outs[4 * c + b][:, :w // 2], want[b][:, :w // 2]), (c, b)


@pytest.mark.parametrize("w,h,dh,big_endian", [(32, 16, 16, 1), (136, 40, 37, 0), (320, 48, 48, 1)])
def test_fwd_packed16_level1_of_rgb10(w, h, dh, big_endian):
    """10-bit RGB fields of a 32-bit word per pixel (r210 / DPX0 big-endian, AB10 / AR10 little-endian): the loader of k_fwd_packed16 takes a byte
    order and a bit position per plane = the oracle's plane transform of the fields << 2 (rows below the picture repeat the last row)."""
    rng = np.random.default_rng(w + h)
    words = rng.integers(0, 1 << 32, size=(dh, w + 3), dtype=np.uint64).astype(np.uint32)
    buf = words.astype(">u4" if big_endian else "<u4")
    shifts = [12, 22, 2] if big_endian else [10, 0, 20]
    quant = [1, 12, 12, 24] * 3
    opitch = (w // 2 + 7) // 8 * 8
    outs = [np.zeros((h // 2, opitch), np.int16) for _ in range(12)]
    ptrs = (c_i16p * 12)(*[p16(o) for o in outs])
    E = emu()
    E.emu_fwd_rgb10.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    E.emu_fwd_rgb10(buf.ctypes.data_as(ctypes.c_void_p), (w + 3) * 4, w, h, dh, big_endian, iarr(shifts), iarr(quant), 2, ptrs, opitch)
    for c in range(3):
        plane = np.zeros((h, w), np.int16); plane[:dh] = (((words[:, :w] >> shifts[c]) & 0x3ff) << 2).astype(np.int16); plane[dh:] = plane[dh - 1]
        want = [np.zeros((h // 2, opitch), np.int16) for _ in range(4)]
        bands = (c_i16p * 4)(*[p16(o) for o in want])
        oracle().orc_fwd_spatial(p16(plane), w, w, h, 0, iarr(quant[:4]), 2, bands, opitch)
        for b in range(4):
            assert np.array_equal(outs[4 * c + b][:, :w // 2], want[b][:, :w // 2]), (c, b)


# ---------------------------------------------------------------------------------------------------------------
# Round 4: the register-strip kernels of config D, the block lists of the decode side, the row-per-wave running sums
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,dh", [(64, 16, 16), (736, 24, 21), (1920, 8, 8), (2112, 36, 33), (4000, 8, 8)])
@pytest.mark.parametrize("uyvy", [0, 1])
def test_fwd_frame_yuv422_strip_kernel(w, h, dh, uyvy):
    """k_fwd_frame_yuv422_strip = the oracle's TransformForwardFrameYUV (as test_fwd_frame_yuv422_interlaced_level1 for the tiled kernel): full-range bytes with strong field
    flicker, one to three segments of 1984 pixels with a partial last one, pad rows below the picture, more than one workgroup of row pairs."""
    rng = np.random.default_rng(w + h + uyvy)
    frame = rng.integers(0, 256, size=(dh, w * 2), dtype=np.int64).astype(np.uint8)
    frame[0::2, : w] = rng.choice([0, 255], size=(len(frame[0::2]), w))
    quant = [1, 36, 16, 36, 1, 36, 16, 48, 1, 36, 16, 48]
    outs_o, outs_e, pitches = [], [], []
    for c in range(3):
        cw = (w if c == 0 else w // 2) // 2
        pitch = (cw + 7) // 8 * 8; pitches.append(pitch)
        outs_o.append([np.zeros((h // 2, pitch), np.int16) for _ in range(4)])
        outs_e.append([np.full((h // 2, pitch), 77, np.int16) for _ in range(4)])
    padded = np.full((h, w * 2), 0x80, np.uint8); padded[:dh] = frame
    O = oracle()
    for c in range(3):
        bands = (c_i16p * 4)(*[p16(a) for a in outs_o[c]])
        O.orc_fwd_frame_yuv422(p8(padded), w * 2, w if c == 0 else w // 2, h, c, 2, uyvy, iarr(quant[4 * c: 4 * c + 4]), 2, bands, pitches[c])
    ptrs = (c_i16p * 12)(*[p16(a) for c in range(3) for a in outs_e[c]])
    assert emu().emu_fwd_frame_yuv422_strip(p8(frame), w * 2, w, h, dh, uyvy, 2, iarr(quant), 2, ptrs, iarr(pitches)) == 0
    for c in range(3):
        cw = (w if c == 0 else w // 2) // 2
        for b in range(4):
            assert np.array_equal(outs_e[c][b][:, :cw], outs_o[c][b][:, :cw]), (c, b)


@pytest.mark.parametrize("w,h,dh,order", [(40, 8, 8, 0), (304, 24, 21, 0), (1000, 72, 70, 0), (504, 16, 16, 1), (504, 16, 13, 2), (72, 40, 40, 3)])
def test_fwd_bayer_strip_kernel(w, h, dh, order):
    """k_fwd_bayer_strip (level 1 straight from the BYR4 mosaic, curve LUT in LDS, four planes from one pass) = k_unpack_byr4 (pinned on the oracle above) followed by the
    oracle's plane transform of the four component planes: full-range photosites, all four pixel orders, one to three segments of 62 blocks, pad rows, several strips."""
    rng = np.random.default_rng(w + h + order)
    O = oracle()
    mosaic = rng.integers(0, 65536, size=(2 * dh, 2 * w), dtype=np.int64).astype(np.uint16)
    curve = np.zeros(1 << 14, np.uint16)
    O.orc_byr4_log90_curve.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    O.orc_byr4_log90_curve(12, 14, curve.ctypes.data_as(ctypes.c_void_p))
    E = emu()
    ppitch = (w + 15) // 16 * 16
    planes = [np.zeros((h, ppitch), np.int16) for _ in range(4)]
    E.emu_unpack_byr4.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    E.emu_unpack_byr4(mosaic.ctypes.data_as(ctypes.c_void_p), 2 * w, w, h, dh, curve.ctypes.data_as(ctypes.c_void_p), order, 12, (c_i16p * 4)(*[p16(g) for g in planes]), ppitch)
    quant = [1, 24, 24, 12, 1, 36, 36, 18, 1, 36, 36, 18, 1, 48, 48, 24]
    pitch = (w // 2 + 7) // 8 * 8
    want = []
    for c in range(4):
        o = [np.zeros((h // 2, w // 2), np.int16) for _ in range(4)]
        O.orc_fwd_spatial(p16(planes[c]), ppitch, w, h, 0, iarr(quant[4 * c: 4 * c + 4]), 2, (c_i16p * 4)(*[p16(a) for a in o]), w // 2)
        want.append(o)
    outs = [np.full((h // 2, pitch), 77, np.int16) for _ in range(16)]
    E.emu_fwd_bayer_strip.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    assert E.emu_fwd_bayer_strip(mosaic.ctypes.data_as(ctypes.c_void_p), 2 * w, w, h, dh, curve.ctypes.data_as(ctypes.c_void_p), order, iarr(quant), 2, (c_i16p * 16)(*[p16(o) for o in outs]), pitch) == 0
    for c in range(4):
        for b in range(4):
            assert np.array_equal(outs[4 * c + b][:, : w // 2], want[c][b]), (c, b)
            assert np.all(outs[4 * c + b][:, w // 2:] == 77)


def _as_block_lists(band, pitch, masks, mask_at):
    """A dense band (rows of `pitch` coefficients) -> the block-list form of cfhd_core.h dec_block_list_layout in place: per chunk of 64 blocks of the flat raster the
    nonzero blocks compacted at the chunk's first places, junk behind them, the chunk's mask in masks[mask_at + chunk]."""
    flat = band.reshape(-1).copy()
    blocks = flat.reshape(-1, 8)
    out = np.full_like(blocks, 0x0bad)
    for k in range((len(blocks) + 63) // 64):
        chunk = blocks[64 * k: 64 * k + 64]
        nz = np.any(chunk != 0, axis=1)
        m = 0
        for i, flag in enumerate(nz):
            if flag: m |= 1 << i
        masks[mask_at + k] = m
        out[64 * k: 64 * k + int(nz.sum())] = chunk[nz]
    return out.reshape(band.shape)


@pytest.mark.parametrize("w,h,dh,interlaced", [(32, 8, 16, 0), (96, 20, 40, 0), (1008, 21, 41, 0), (128, 17, 34, 1), (2000, 6, 12, 1), (1040, 33, 66, 1)])
@pytest.mark.parametrize("uyvy", [0, 1])
def test_inverse_strip_kernels_gather_block_lists(w, h, dh, interlaced, uyvy):
    """k_inv_yuv422_strip_blocks / k_inv_frame_yuv422_strip_blocks: the level-1 highpass bands as block lists (all three for progressive frames, LH and HH for interlaced
    ones) give byte for byte the picture of the same kernels reading the dense bands -- sparse bands (two thirds of the blocks empty), chunks that straddle rows, a last
    chunk that is partial, segments of 124 blocks with a partial last one."""
    rng = np.random.default_rng(w + h + uyvy + interlaced)
    E = emu()
    bands, pitches, mask_base, at = [], [], [], 0
    for ch in range(3):
        cw = w if ch == 0 else w // 2
        pitch = (cw + 7) // 8 * 8; pitches.append(pitch)
        bs = [np.zeros((h, pitch), np.int16) for _ in range(4)]
        bs[0][:, :cw] = rand_plane(rng, cw, h, 11)
        for k in range(1, 4):
            v = rand_plane(rng, cw, h, 9, signed=True)
            keep = np.repeat(rng.random((h, (cw + 7) // 8)) < 0.35, 8, axis=1)[:, :cw]      # a third of the blocks hold anything
            bs[k][:, :cw] = np.where(keep, v, 0)
        bands.append(bs)
        for b in range(4):
            mask_base.append(at if b else -1)
            if b: at += (h * pitch + 511) // 512
    ptrs = (c_i16p * 12)(*[p16(a) for ch in range(3) for a in bands[ch]])
    dense = np.full((2 * h, 4 * w + 16), 7, np.uint8)
    if interlaced:
        E.emu_inv_frame_yuv422_strip.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_uint, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        assert E.emu_inv_frame_yuv422_strip(ptrs, iarr(pitches), w, h, dh, uyvy, 2, 1234, p8(dense), 4 * w + 16, None, None) == 0
        # (the dense strip kernel against the oracle: every byte with dither 0 or with dither 1)
        outs = []
        for dither in (0, 1):
            o = np.zeros((2 * h, 4 * w), np.uint8)
            oracle().orc_inv_frame_to_yuv422(ptrs, iarr(pitches), w, h, 10, uyvy, dither, p8(o), 4 * w)
            outs.append(o[:dh])
        e = dense[:dh, :4 * w]
        assert np.all((e == outs[0]) | (e == outs[1]))
    else:
        assert E.emu_inv_yuv422_strip(ptrs, iarr(pitches), w, h, dh, uyvy, 2, 1234, p8(dense), 4 * w + 16) == 0
    masks = np.full(at + 8, 0x5555555555555555, np.uint64)
    listed = []
    for ch in range(3):
        bs = list(bands[ch])
        for b in ((1, 3) if interlaced else (1, 2, 3)):
            bs[b] = _as_block_lists(bands[ch][b], pitches[ch], masks, mask_base[4 * ch + b])
        listed.append(bs)
    lptrs = (c_i16p * 12)(*[p16(a) for ch in range(3) for a in listed[ch]])
    got = np.full((2 * h, 4 * w + 16), 7, np.uint8)
    fn = E.emu_inv_frame_yuv422_strip if interlaced else E.emu_inv_yuv422_strip_blocks
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_uint, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert fn(lptrs, iarr(pitches), w, h, dh, uyvy, 2, 1234, p8(got), 4 * w + 16, masks.ctypes.data_as(ctypes.c_void_p), iarr(mask_base)) == 0
    assert np.array_equal(got, dense)
    assert np.all(got[:, 4 * w:] == 7) and np.all(got[dh:] == 7)


@pytest.mark.parametrize("w,h", [(5, 3), (100, 7), (960, 13), (1030, 4), (2050, 9), (16, 70)])
def test_dec_undiff_rows_kernel(w, h):
    """k_dec_undiff_rows: every row of a difference-coded band becomes its running sum (16-bit wrap), the pitch padding stays zero, rows of more than 1024 columns in
    pieces with a carry, widths that are no multiple of the lanes' 16 columns."""
    rng = np.random.default_rng(w * 3 + h)
    pitch = (w + 7) // 8 * 8
    band = np.zeros((h, pitch), np.int16)
    band[:, :w] = rng.integers(-3000, 3000, size=(h, w)).astype(np.int16)
    band[0, :w] = 32767 if w > 2 else 1                                   # wraps
    want = np.zeros_like(band)
    want[:, :w] = np.cumsum(band[:, :w].astype(np.int64), axis=1).astype(np.int16)
    got = band.copy()
    E = emu()
    E.emu_dec_undiff_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    E.emu_dec_undiff_rows(got.ctypes.data_as(ctypes.c_void_p), w, h, pitch)
    assert np.array_equal(got, want)
