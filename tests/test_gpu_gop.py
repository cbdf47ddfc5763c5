"""Two-frame groups on the GPU through the C ABI: CFHD_EncodeSample with CFHD_ENCODING_FLAGS_YUV_2FRAME_GOP gives
the reference encoder's samples byte for byte (sequence header, groups, P-frame headers); CFHD_DecodeSample decodes the reference's group
samples to pictures inside the dither interval of the exact reconstruction (the oracle's inverse model, which tests/test_gop.py pins on the reference's own group
decoder)."""
import ctypes
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
    assert L.CFHD_PrepareToEncode(enc, 320, 240, PIX_YUY2, ENCODED_YUV422, ENCODING_FLAGS_2FRAME_GOP | 1, QUALITY_FILMSCAN1) == 3     # interlaced groups: not built
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
        co = host_decode_group(samples[2 * g + 1], gp)
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


def test_interlaced_group_samples_are_refused_not_misdecoded():
    """A reference-encoded group of interlaced frames (YUV_INTERLACED | 2FRAME_GOP) carries no SAMPLE_FLAGS tag: `progressive` stays at the reference's default 0
    (codec.c:263, decoder.c:13397).  The field transform of groups is not built, so the decoder must answer CFHD_ERROR_BADFORMAT with a zero-filled picture -- not run
    the progressive inverse over it and return a wrong picture with CFHD_ERROR_OKAY (advisor finding, round 3)."""
    assert have_ref(), "oracle/_ref/libcfhd_ref.so is missing"
    w, h = 320, 240
    frames = _frames(w, h, 3, PIX_YUY2)
    samples = ref_encode_frames(frames, w * 2, w, h, flags=ENCODING_FLAGS_2FRAME_GOP | 1)
    L = product()
    dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
    aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
    sb = ctypes.create_string_buffer(samples[1], len(samples[1]))
    assert L.CFHD_PrepareToDecode(dec, 0, 0, PIX_YUY2, 1, 0, sb, min(512, len(samples[1])), ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0
    out = np.full(w * 2 * ah.value, 7, np.uint8)
    assert L.CFHD_DecodeSample(dec, sb, len(samples[1]), out.ctypes.data_as(ctypes.c_void_p), w * 2) == 3          # CFHD_ERROR_BADFORMAT
    assert not out.reshape(ah.value, w * 2)[:h].any()
    L.CFHD_CloseDecoder(dec)


@pytest.mark.parametrize("stage", ["device", "host"])
def test_group_entropy_stage_on_the_gpu_and_on_the_host(stage):
    """Groups through the GPU entropy stage (the default: 17 subbands per channel through k_ent_count / scan / layout / emit with the two raw 16-bit bands of a channel as
    lowpass holes; decode: 45 band jobs of k_dec_bands_par_ll + 6 of k_dec_lowpass) and through the host coder (CFHD_AMD_ENTROPY=host).  CFHD_AMD_ENTROPY=device makes a
    silent hand-over to the host coder an error, so that leg proves the device stage served every sample.  Both legs: the reference encoder's bytes, pictures inside the
    dither interval of the exact reconstruction (the checks of the two tests above, odd chroma lowpass width and rate feedback included)."""
    import os
    old = os.environ.get("CFHD_AMD_ENTROPY")
    os.environ["CFHD_AMD_ENTROPY"] = stage
    try:
        test_gop_encode_bitstream_identical(320, 240, PIX_YUY2)
        test_gop_rate_feedback_bitstream_identical(320, 180, 6, 5)
        test_gop_decode_reference_samples(336, 252, PIX_YUY2)
    finally:
        if old is None: del os.environ["CFHD_AMD_ENTROPY"]
        else: os.environ["CFHD_AMD_ENTROPY"] = old


@pytest.mark.parametrize("stage", ["default", "host"])
def test_group_decoder_survives_fuzzed_samples(stage):
    """Damaged group samples through CFHD_DecodeSample (GPU entropy stage: GpuGroupEntropyDecoder's job table + k_dec_bands_par_ll; and the host coder): bursts of garbage,
    bit flips, oversized size fields, truncation.  Every call returns (OKAY with some picture, BADSAMPLE / BADFORMAT with a zero-filled one), nothing is written behind the
    output buffer, and the handle decodes the intact group correctly afterwards."""
    import os
    assert have_ref(), "oracle/_ref/libcfhd_ref.so is missing"
    w, h = 320, 240
    frames = _frames(w, h, 2, PIX_YUY2)
    samples = ref_encode_frames(frames, w * 2, w, h, pixfmt=PIX_YUY2, flags=ENCODING_FLAGS_2FRAME_GOP)
    group = np.frombuffer(samples[1], np.uint8).copy()
    old = os.environ.get("CFHD_AMD_ENTROPY")
    if stage == "host": os.environ["CFHD_AMD_ENTROPY"] = "host"
    try:
        L = product()
        dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
        aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
        sb = ctypes.create_string_buffer(samples[0], len(samples[0]))
        assert L.CFHD_PrepareToDecode(dec, 0, 0, PIX_YUY2, 1, 0, sb, len(samples[0]), ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0
        H = ah.value
        out = np.zeros(H * w * 2 + 4096, np.uint8)
        rng = np.random.default_rng(77)
        codes = {}
        for trial in range(24):
            t = group.copy()
            kind = trial % 4
            if kind == 0:
                lo = int(rng.integers(700, len(t) - 128)); n = int(rng.integers(1, 128)); t[lo: lo + n] = rng.integers(0, 256, n, dtype=np.uint8)
            elif kind == 1:
                for _ in range(int(rng.integers(1, 6))):
                    i = int(rng.integers(700, len(t))); t[i] ^= np.uint8(1 << int(rng.integers(0, 8)))
            elif kind == 2:
                i = int(rng.integers(180, len(t) // 4)) * 4; t[i: i + 4] = [0x20 | int(rng.integers(0, 32)), int(rng.integers(0, 256)), 0xff, 0xff]
            size = len(t) if kind != 3 else int(rng.integers(1024, len(t))) & ~3
            out[:] = 7
            tb = ctypes.create_string_buffer(t.tobytes(), len(t))
            rc = L.CFHD_DecodeSample(dec, tb, size, out.ctypes.data_as(ctypes.c_void_p), w * 2)
            codes[rc] = codes.get(rc, 0) + 1
            assert rc in (0, 3, 5), (trial, rc, amd_last_error())
            assert np.all(out[H * w * 2:] == 7), "trial %d wrote behind the output buffer" % trial
            if rc: assert not out[: h * w * 2].any()
        assert sum(v for k, v in codes.items() if k) >= 3, codes
        # the intact group afterwards: both frames at intra-like quality
        tb = ctypes.create_string_buffer(samples[1], len(samples[1])); out[:] = 0
        assert L.CFHD_DecodeSample(dec, tb, len(samples[1]), out.ctypes.data_as(ctypes.c_void_p), w * 2) == 0
        assert psnr_yuy2(out[: h * w * 2].reshape(h, w * 2), frames[0].reshape(h, w * 2)) > 38.0
        L.CFHD_CloseDecoder(dec)
    finally:
        if stage == "host":
            if old is None: del os.environ["CFHD_AMD_ENTROPY"]
            else: os.environ["CFHD_AMD_ENTROPY"] = old
