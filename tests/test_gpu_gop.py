"""Two-frame groups on the GPU through the C ABI: CFHD_EncodeSample with CFHD_ENCODING_FLAGS_YUV_2FRAME_GOP gives
the reference encoder's samples byte for byte (sequence header, groups, P-frame headers); CFHD_DecodeSample decodes the reference's group
samples to pictures inside the dither interval of the exact reconstruction (the oracle's inverse model, which tests/test_gop.py pins on the reference's own group
decoder)."""
import ctypes, os
import numpy as np
import pytest
from cfhd_testlib import *

pytestmark = pytest.mark.gpu


def _frames(w, h, n, fmt):
    fr = [synth_yuy2(w, h, 70 + i)[0] for i in range(n)]
    if fmt == PIX_2VUY: fr = [f.reshape(-1, 2)[:, ::-1].reshape(-1).copy() for f in fr]
    return fr


@pytest.mark.parametrize("w,h,fmt", [(320, 240, PIX_YUY2), (720, 486, PIX_2VUY), (1920, 1080, PIX_YUY2)])
def test_gop_encode_bitstream_identical(w, h, fmt):
    assert have_ref(), "oracle/_ref/libcfhd_ref.so is missing"
    frames = _frames(w, h, 6, fmt) if (w, h) != (1920, 1080) else qbist_frames(10, 6, w, h)[0]
    mine = amd_encode_frames(frames, w * 2, w, h, fmt, flags=ENCODING_FLAGS_2FRAME_GOP)
    refs = ref_encode_frames(frames, w * 2, w, h, pixfmt=fmt, flags=ENCODING_FLAGS_2FRAME_GOP)
    assert [len(s) for s in mine] == [len(s) for s in refs]
    for i, (a, b) in enumerate(zip(mine, refs)):
        assert mask_volatile_metadata(a) == mask_volatile_metadata(b), "sample %d differs from the reference" % i


@pytest.mark.parametrize("w,h,n,quality", [(320, 180, 6, 5), (320, 180, 6, 6), (640, 360, 8, 5), (640, 360, 8, 6), (1920, 1080, 6, 2), (1920, 1080, 6, 3)])
def test_gop_rate_feedback_bitstream_identical(w, h, n, quality):
    """Groups whose tables follow the size of the last key sample (encoder.c:2880-2905, :3414): FILMSCAN2 / FILMSCAN3 move their limiter on every call, MEDIUM / HIGH
    at 1080p run into the bit-rate limiter (two noisy 1080p frames are far beyond 130 Mbit/s), and every group here fills more than 80% of the reference's sample
    buffer somewhere in the frame wavelets, behind which the reference codes the remaining bands as zeros (encoder.c:8332).  Byte for byte the reference's samples."""
    assert have_ref(), "oracle/_ref/libcfhd_ref.so is missing"
    frames = feedback_test_frames(w, h, n)
    mine = amd_encode_frames(frames, w * 2, w, h, PIX_YUY2, flags=ENCODING_FLAGS_2FRAME_GOP, quality=quality)
    refs = ref_encode_frames(frames, w * 2, w, h, pixfmt=PIX_YUY2, flags=ENCODING_FLAGS_2FRAME_GOP, quality=quality)
    assert [len(s) for s in mine] == [len(s) for s in refs]
    for i, (a, b) in enumerate(zip(mine, refs)):
        assert mask_volatile_metadata(a) == mask_volatile_metadata(b), "sample %d differs from the reference" % i
    assert len(set(len(s) for s in mine[1::2])) > 1      # (the groups do differ in size: the feedback had something to follow)


def test_gop_gates():
    L = product()
    enc = ctypes.c_void_p(); assert L.CFHD_OpenEncoder(ctypes.byref(enc), None) == 0
    assert L.CFHD_PrepareToEncode(enc, 320, 240, PIX_YUY2, ENCODED_YUV422, ENCODING_FLAGS_2FRAME_GOP | 1, QUALITY_FILMSCAN1) == 0     # interlaced groups (round 5)
    assert L.CFHD_PrepareToEncode(enc, 320, 240, PIX_RG48, ENCODED_RGB444, ENCODING_FLAGS_2FRAME_GOP, QUALITY_FILMSCAN1) == 3        # 4:2:2 only
    assert L.CFHD_PrepareToEncode(enc, 320, 240, PIX_YUY2, ENCODED_YUV422, ENCODING_FLAGS_2FRAME_GOP, 5) == 0                       # FILMSCAN2 (rate feedback: below)
    assert L.CFHD_PrepareToEncode(enc, 320, 240, PIX_YUY2, ENCODED_YUV422, ENCODING_FLAGS_2FRAME_GOP, QUALITY_FILMSCAN1) == 0
    L.CFHD_CloseEncoder(enc)
    pool = ctypes.c_void_p(); assert L.CFHD_CreateEncoderPool(ctypes.byref(pool), 2, 2, None) == 0
    assert L.CFHD_PrepareEncoderPool(pool, 320, 240, PIX_YUY2, ENCODED_YUV422, ENCODING_FLAGS_2FRAME_GOP, QUALITY_FILMSCAN1) == 3
    L.CFHD_ReleaseEncoderPool(pool)


@pytest.mark.parametrize("w,h,fmt", [(320, 240, PIX_YUY2), (336, 252, PIX_YUY2), (720, 480, PIX_2VUY), (1920, 1080, PIX_YUY2)])
def test_gop_decode_reference_samples(w, h, fmt):
    assert have_ref(), "oracle/_ref/libcfhd_ref.so is missing"
    kind = 2 if fmt == PIX_2VUY else 1
    frames = _frames(w, h, 5, fmt)               # sequence header, group, P-frame header, group, P-frame header
    samples = ref_encode_frames(frames, w * 2, w, h, pixfmt=fmt, flags=ENCODING_FLAGS_2FRAME_GOP)
    gp = GopPlan(w, h, pixkind=kind)
    L = product()
    dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
    aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
    sb = ctypes.create_string_buffer(samples[0], len(samples[0]))
    assert L.CFHD_PrepareToDecode(dec, 0, 0, fmt, 1, 0, sb, len(samples[0]), ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0
    assert (aw.value, ah.value) == (w, (h + 7) // 8 * 8)               # as the reference: the sequence header carries the coded height
    H = ah.value
    outs = []
    for s in samples:
        sb = ctypes.create_string_buffer(s, len(s)); out = np.full(w * 2 * H, 7, np.uint8)
        assert L.CFHD_DecodeSample(dec, sb, len(s), out.ctypes.data_as(ctypes.c_void_p), w * 2) == 0, amd_last_error()
        outs.append(out.reshape(H, w * 2))
    L.CFHD_CloseDecoder(dec)
    assert (outs[0] == 7).all()                                          # the sequence header decodes to nothing
    # the gate: the oracle's group inverse, which tests/test_gop.py pins on the reference's group decoder byte for byte inside the dither interval (defect of its
    # last wavelet row included: cineform-sdk_amd InvPlaneJob::ll_bottom_row_high); the reference decoder itself runs beside it as a witness
    intervals = {}
    for g in range(2):
        co = oracle_decode_group(samples[2 * g + 1], gp)
        lo = oracle_inverse_gop(gp, co, 0, uyvy=int(fmt == PIX_2VUY)); hi = oracle_inverse_gop(gp, co, 1, uyvy=int(fmt == PIX_2VUY))
        for f in range(2):
            img = outs[2 * g + 1 + f][:h]
            ok = (img == lo[f][:h]) | (img == hi[f][:h])
            assert ok.all(), "group %d frame %d: %d bytes outside the dither interval" % (g, f, (~ok).sum())
            assert psnr_yuy2(img, frames[2 * g + f].reshape(h, w * 2)) > 38.0
            intervals[(g, f)] = (lo[f][:h], hi[f][:h], img)
    def leg():
        got = ref_decode_group_frames(samples, w, h, fmt)
        for (g, f), (lo_f, hi_f, img) in intervals.items():
            r = got[g][f]
            if r is None: continue
            if not ((r == lo_f) | (r == hi_f)).all(): return "group %d frame %d: the reference decoder's picture leaves the interval" % (g, f)
            src = frames[2 * g + f].reshape(h, w * 2)
            if abs(psnr_yuy2(r, src) - psnr_yuy2(img, src)) >= 0.1: return "group %d frame %d: PSNR %.2f vs reference %.2f" % (g, f, psnr_yuy2(img, src), psnr_yuy2(r, src))
        return True
    reference_leg(leg, 2, "two-frame groups -> 8-bit 4:2:2")


def test_group_lowpass_words_beyond_int16_decode_alike_on_both_stages():
    """The raw lowpass band of a group's top wavelet: a band of odd width is read 16 unsigned bits at a time (decoder.c:12240-12290, GetBits), one of even width as
    signed words.  720 pixels give 45 chroma lowpass columns: with words of 0x8000 and above in the band (no encoder writes them) the device stage (k_dec_lowpass) and
    the host coder must still return the same pictures -- the host path read them as signed until round 5 (advisor finding)."""
    assert have_ref(), "oracle/_ref/libcfhd_ref.so is missing"
    w, h, fmt = 720, 240, PIX_YUY2
    frames = _frames(w, h, 2, fmt)
    samples = ref_encode_frames(frames, w * 2, w, h, pixfmt=fmt, flags=ENCODING_FLAGS_2FRAME_GOP)
    group = bytearray(samples[1])
    gp = GopPlan(w, h, pixkind=1)
    d5 = gp.w[(1, 5)]
    assert d5["width"] % 2 == 1
    # where the band lies in the sample: its decoded values are the sample's big-endian words plus a constant (the decoder's lowpass bias)
    co = oracle_decode_group(bytes(group), gp)
    top = gp.view(co, 1, 5, 0)[0, :8].astype(np.int64)
    words = np.frombuffer(bytes(group[: len(group) // 2 * 2]), dtype=">u2").astype(np.int64)
    hits = [i for i in range(0, min(len(words), 65536) - 8, 2) if np.array_equal(words[i:i + 8] - words[i], top - top[0]) and words[i] > 256]
    assert len(hits) == 1, hits
    at = 2 * hits[0]
    for k, v in ((0, 0x8000), (3, 0x9abc), (6, 0xffff)): group[at + 2 * k: at + 2 * k + 2] = v.to_bytes(2, "big")
    L = product()
    pictures = {}
    old = os.environ.get("CFHD_AMD_ENTROPY")
    try:
        for stage in ("device", "host"):
            os.environ["CFHD_AMD_ENTROPY"] = stage
            dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
            aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
            sb = ctypes.create_string_buffer(samples[0], len(samples[0]))
            assert L.CFHD_PrepareToDecode(dec, 0, 0, fmt, 1, 0, sb, len(samples[0]), ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0
            sb = ctypes.create_string_buffer(bytes(group), len(group)); out = np.full(w * 2 * ah.value, 7, np.uint8)
            assert L.CFHD_DecodeSample(dec, sb, len(group), out.ctypes.data_as(ctypes.c_void_p), w * 2) == 0, amd_last_error()
            L.CFHD_CloseDecoder(dec)
            pictures[stage] = out.copy()
    finally:
        if old is None: os.environ.pop("CFHD_AMD_ENTROPY", None)
        else: os.environ["CFHD_AMD_ENTROPY"] = old
    # (the 8-bit output draws one dither bit per sample from the call counter: both handles decoded their first picture, so the bits agree)
    assert np.array_equal(pictures["device"], pictures["host"])


def test_gop_round_trip_of_the_product_alone():
    """Encode and decode with the product only, entered at a group (no sequence header): every frame comes back at intra-like quality."""
    w, h = 640, 360
    frames = _frames(w, h, 4, PIX_YUY2)
    samples = amd_encode_frames(frames, w * 2, w, h, PIX_YUY2, flags=ENCODING_FLAGS_2FRAME_GOP)
    L = product()
    dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
    aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
    sb = ctypes.create_string_buffer(samples[1], len(samples[1]))
    assert L.CFHD_PrepareToDecode(dec, 0, 0, PIX_YUY2, 1, 0, sb, 512, ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0
    assert (aw.value, ah.value) == (w, h)
    for i, s in enumerate(samples[1:]):
        sb = ctypes.create_string_buffer(s, len(s)); out = np.zeros(w * 2 * h, np.uint8)
        assert L.CFHD_DecodeSample(dec, sb, len(s), out.ctypes.data_as(ctypes.c_void_p), w * 2) == 0
        assert psnr_yuy2(out.reshape(h, w * 2), frames[i].reshape(h, w * 2)) > 40.0, i
    L.CFHD_CloseDecoder(dec)


def _interlaced_frames(w, h, n, fmt, flicker=False):
    """Frames whose fields disagree (tests/test_gop.py: the second field shifted sideways; flicker: field differences beyond the peak threshold)."""
    import test_gop
    return test_gop._interlaced_frames(w, h, n, fmt, flicker)


@pytest.mark.parametrize("w,h,fmt,flicker", [(320, 240, PIX_YUY2, 0), (336, 252, PIX_YUY2, 1), (720, 486, PIX_2VUY, 0), (1920, 1080, PIX_YUY2, 0)])
def test_interlaced_gop_encode_bitstream_identical(w, h, fmt, flicker):
    """CFHD_ENCODING_FLAGS_YUV_INTERLACED | _2FRAME_GOP (round 5): frame transform at level 1 of both frames (k_fwd_frame_yuv422 on the group's job table), subbands 12
    and 15 difference coded in code set 18 with peaks on the GPU entropy stage -- the flicker frames carry peak tables, written on the device as well (k_ent_peaks;
    CFHD_AMD_ENTROPY=device makes a sample handed to the host writer fail the call).  Byte for byte the reference's samples."""
    assert have_ref(), "oracle/_ref/libcfhd_ref.so is missing"
    frames = _interlaced_frames(w, h, 4, fmt, bool(flicker))
    old = os.environ.get("CFHD_AMD_ENTROPY")
    os.environ["CFHD_AMD_ENTROPY"] = "device"
    try:
        mine = amd_encode_frames(frames, w * 2, w, h, fmt, flags=ENCODING_FLAGS_2FRAME_GOP | 1)
    finally:
        if old is None: os.environ.pop("CFHD_AMD_ENTROPY")
        else: os.environ["CFHD_AMD_ENTROPY"] = old
    refs = ref_encode_frames(frames, w * 2, w, h, pixfmt=fmt, flags=ENCODING_FLAGS_2FRAME_GOP | 1)
    assert [len(s) for s in mine] == [len(s) for s in refs]
    for i, (a, b) in enumerate(zip(mine, refs)):
        assert mask_volatile_metadata(a) == mask_volatile_metadata(b), "sample %d differs from the reference" % i


@pytest.mark.parametrize("w,h,fmt,flicker", [(320, 240, PIX_YUY2, 0), (336, 252, PIX_YUY2, 1), (720, 480, PIX_2VUY, 0), (1920, 1080, PIX_YUY2, 0)])
def test_interlaced_gop_decode_reference_samples(w, h, fmt, flicker):
    """A reference-encoded group of interlaced frames carries no SAMPLE_FLAGS tag: `progressive` stays at the reference's default 0 (codec.c:263, decoder.c:13397).
    Round 4 refused such samples (CFHD_ERROR_BADFORMAT, zero-filled picture); now they decode on the device: the difference-coded bands with the tables of code set 18,
    k_dec_undiff for peak values and running sums (CFHD_AMD_ENTROPY=device: a sample handed to the host coder fails the call), the group's inverse with the inverse
    frame transform as its last level.  Gate: the oracle's group inverse, which
    tests/test_gop.py pins on the reference's own group decoder byte for byte inside the dither interval; the reference decoder runs beside it as a witness."""
    assert have_ref(), "oracle/_ref/libcfhd_ref.so is missing"
    kind = 2 if fmt == PIX_2VUY else 1
    frames = _interlaced_frames(w, h, 4, fmt, bool(flicker))
    samples = ref_encode_frames(frames, w * 2, w, h, pixfmt=fmt, flags=ENCODING_FLAGS_2FRAME_GOP | 1)
    gp = GopPlan(w, h, pixkind=kind, interlaced=1)
    L = product()
    dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
    aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
    sb = ctypes.create_string_buffer(samples[1], len(samples[1]))
    assert L.CFHD_PrepareToDecode(dec, 0, 0, fmt, 1, 0, sb, min(512, len(samples[1])), ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0
    H = ah.value
    outs = []
    old = os.environ.get("CFHD_AMD_ENTROPY")
    os.environ["CFHD_AMD_ENTROPY"] = "device"
    try:
        for s in samples[1:]:
            sb = ctypes.create_string_buffer(s, len(s)); out = np.full(w * 2 * H, 7, np.uint8)
            assert L.CFHD_DecodeSample(dec, sb, len(s), out.ctypes.data_as(ctypes.c_void_p), w * 2) == 0, amd_last_error()
            outs.append(out.reshape(H, w * 2))
    finally:
        if old is None: os.environ.pop("CFHD_AMD_ENTROPY")
        else: os.environ["CFHD_AMD_ENTROPY"] = old
    L.CFHD_CloseDecoder(dec)
    intervals = {}
    for g in range(2):
        co = oracle_decode_group(samples[2 * g + 1], gp)
        lo = oracle_inverse_gop(gp, co, 0, uyvy=int(kind == 2)); hi = oracle_inverse_gop(gp, co, 1, uyvy=int(kind == 2))
        for f in range(2):
            if 2 * g + f >= len(outs): continue
            img = outs[2 * g + f][:h]
            ok = (img == lo[f][:h]) | (img == hi[f][:h])
            assert ok.all(), "group %d frame %d: %d bytes outside the dither interval" % (g, f, (~ok).sum())
            intervals[(g, f)] = (lo[f][:h], hi[f][:h], img)
    def leg():
        got = ref_decode_group_frames(samples, w, h, fmt)
        for (g, f), (lo_f, hi_f, img) in intervals.items():
            r = got[g][f]
            if r is None: continue
            if not ((r == lo_f) | (r == hi_f)).all(): return "group %d frame %d: the reference decoder's picture leaves the interval" % (g, f)
        return True
    reference_leg(leg, 2, "interlaced two-frame groups -> 8-bit 4:2:2")


def test_interlaced_gop_round_trip_of_the_product_alone():
    """Encode and decode interlaced groups with the product only: every frame comes back at intra-like quality."""
    w, h = 640, 360
    frames = _interlaced_frames(w, h, 4, PIX_YUY2)
    samples = amd_encode_frames(frames, w * 2, w, h, PIX_YUY2, flags=ENCODING_FLAGS_2FRAME_GOP | 1)
    L = product()
    dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
    aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
    sb = ctypes.create_string_buffer(samples[1], len(samples[1]))
    assert L.CFHD_PrepareToDecode(dec, 0, 0, PIX_YUY2, 1, 0, sb, 512, ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0
    for i, s in enumerate(samples[1:]):
        sb = ctypes.create_string_buffer(s, len(s)); out = np.zeros(w * 2 * h, np.uint8)
        assert L.CFHD_DecodeSample(dec, sb, len(s), out.ctypes.data_as(ctypes.c_void_p), w * 2) == 0
        assert psnr_yuy2(out.reshape(h, w * 2), frames[i].reshape(h, w * 2)) > 38.0, i
    L.CFHD_CloseDecoder(dec)
