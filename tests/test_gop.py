"""Two-frame groups (CFHD_ENCODING_FLAGS_YUV_2FRAME_GOP; cineform-sdk_amd/csrc/cfhd_gop.h), CPU side: the oracle's
restatement of the group transform (oracle spatial steps + the temporal sum / difference of Codec/temporal.c:498) written with the product's
sample syntax equals the reference encoder's samples byte for byte -- sequence header, groups, P-frame headers -- and the inverse model
(Codec/wavelet.c TransformInverseTemporal + the oracle's spatial synthesis) reconstructs the frames as well as the intra path does."""
import struct
import numpy as np
import pytest
from cfhd_testlib import *

pytestmark = pytest.mark.skipif(not have_ref(), reason="reference .so not built")


def _frames(w, h, n, fmt):
    fr = [synth_yuy2(w, h, 70 + i)[0] for i in range(n)]
    if fmt == PIX_2VUY: fr = [f.reshape(-1, 2)[:, ::-1].reshape(-1).copy() for f in fr]
    return fr


def _frame_number(sample):
    k = sample[:160].find(struct.pack(">h", -69))
    return struct.unpack(">H", sample[k + 2:k + 4])[0]


@pytest.mark.parametrize("w,h,fmt", [(320, 240, PIX_YUY2), (720, 486, PIX_2VUY), (336, 252, PIX_YUY2), (400, 120, PIX_YUY2), (1920, 1080, PIX_YUY2)])
def test_group_samples_equal_reference(w, h, fmt):
    kind = 2 if fmt == PIX_2VUY else 1
    frames = _frames(w, h, 6, fmt)
    ref = ref_encode_frames(frames, w * 2, w, h, pixfmt=fmt, flags=ENCODING_FLAGS_2FRAME_GOP)
    gp = GopPlan(w, h, pixkind=kind)
    assert [len(s) for s in ref[0::2]] == [40, 24, 24]                     # sequence header, then the header of every group's second frame
    assert product_write_gop_host(gp, 1) == ref[0]
    for g in range(3):
        s = ref[2 * g + 1]
        assert _frame_number(s) == 2 * g + 1
        co = oracle_forward_gop(gp, frames[2 * g], frames[2 * g + 1], w * 2, uyvy=int(fmt == PIX_2VUY))
        off, n = first_metadata_chunk(s)
        mine = product_write_gop_host(gp, 0, co, 2 * g + 1, meta_global=s[off:off + n])
        assert len(mine) == len(s)
        assert mine == s, "group %d differs at byte %d" % (g, next(i for i in range(len(s)) if mine[i] != s[i]))
        if g < 2: assert product_write_gop_host(gp, 2, frame_number=2 * g + 1) == ref[2 * g + 2]


def test_group_quantizer_known_answers_and_gates():
    """Band divisors / scales of a FILMSCAN1 group as the reference's band headers carry them (probe: 320x240 luma / chroma); geometries the group transform does not serve."""
    gp = GopPlan(320, 240)
    assert [gp.w[(0, k)]["quant"][1:] for k in (5, 4, 3, 1, 0)] == [[48, 48, 24], [12, 12, 6], [48, 48, 24], [24, 24, 36], [24, 24, 36]]
    assert [gp.w[(0, k)]["scale"] for k in range(6)] == [[4, 2, 2, 1], [4, 2, 2, 1], [8, 4, 0, 0], [16, 8, 8, 4], [32, 16, 16, 8], [128, 64, 64, 32]]
    assert gp.w[(1, 0)]["quant"][3] == 48 and gp.w[(1, 1)]["quant"][3] == 48      # chroma: the coarser HH divisor of the 4:2:2 chroma tables
    assert [gp.w[(0, k)]["prescale"] for k in range(6)] == [0, 0, 0, 0, 2, 0]
    L = hooks()
    buf = (ctypes.c_longlong * 512)()
    L.cfhd_amd_gop_plan_info.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_longlong)]
    assert L.cfhd_amd_gop_plan_info(320, 240, 1, 5, buf) > 0            # FILMSCAN2, MEDIUM: the tables of a sequence's first group (the later ones follow the size
    assert L.cfhd_amd_gop_plan_info(320, 240, 1, 2, buf) > 0            # of the last key sample: tests/test_gpu_gop.py test_gop_rate_feedback_bitstream_identical)
    assert L.cfhd_amd_gop_plan_info(328, 240, 1, 4, buf) == -1          # chroma would not halve on whole pairs


@pytest.mark.parametrize("w,h,fmt", [(320, 240, PIX_YUY2), (336, 252, PIX_YUY2), (720, 480, PIX_2VUY), (400, 120, PIX_YUY2), (1920, 1080, PIX_YUY2)])
def test_group_inverse_model_equals_the_reference_group_decoder(w, h, fmt):
    """Pins oracle_inverse_gop on the reference's own group decoder (Codec/decoder.c:11180 DecodeSampleGroup + :11482 DecodeSampleFrame, driven as
    cfhd_testlib.ref_decode_group_frames documents): reference group samples -> host parser / VLC decoder -> oracle inverse (lowpass bias 48 for groups,
    decoder.c:12265; the unprescaled wavelets through the restated InvertSpatialQuantOverflowProtected16s, spatial.c:21114, whose last coefficient row reads the LL
    band one row too high) -- every byte of both frames of every group of the reference decoder's output lies inside the oracle's dither interval, the PSNR against
    the source agrees to 0.1 dB.  The same inverse without that defect reconstructs the frames better than the reference does (the bottom 16 rows), which is why the
    defect is part of the model: the reference's pictures are the bar, not the best picture."""
    kind = 2 if fmt == PIX_2VUY else 1
    frames = _frames(w, h, 6 if w < 1920 else 2, fmt)
    samples = ref_encode_frames(frames, w * 2, w, h, pixfmt=fmt, flags=ENCODING_FLAGS_2FRAME_GOP)
    gp = GopPlan(w, h, pixkind=kind)
    ngroups = len(frames) // 2
    models = []
    for g in range(ngroups):
        co = oracle_decode_group(samples[2 * g + 1], gp)
        models.append((oracle_inverse_gop(gp, co, 0, uyvy=int(kind == 2)), oracle_inverse_gop(gp, co, 1, uyvy=int(kind == 2)), oracle_inverse_gop(gp, co, 0, uyvy=int(kind == 2), reference_defect=False)))
    def outside(got):
        return [(g, f, int((~((img == models[g][0][f][:h]) | (img == models[g][1][f][:h]))).sum())) for g, pair in enumerate(got) for f, img in enumerate(pair) if img is not None]
    for attempt in range(3):                            # (the reference's group decoder races its own worker: a run in which a picture did not settle gets another decoder)
        got = ref_decode_group_frames(samples, w, h, fmt)
        assert len(got) == ngroups
        if not any(n for _, _, n in outside(got)): break
    for g, pair in enumerate(got):
        lo, hi, better = models[g]
        for f, img in enumerate(pair):
            if img is None: continue                     # (the last group of the stream has no P-frame sample behind it)
            ok = (img == lo[f][:h]) | (img == hi[f][:h])
            assert ok.all(), "group %d frame %d: %d of %d bytes of the reference decoder's picture lie outside the oracle's dither interval" % (g, f, (~ok).sum(), ok.size)
            src = frames[2 * g + f].reshape(h, w * 2)
            if kind == 2: src = src.reshape(-1, 2)[:, ::-1].reshape(h, w * 2); img = img.reshape(-1, 2)[:, ::-1].reshape(h, w * 2); lo_f = lo[f][:h].reshape(-1, 2)[:, ::-1].reshape(h, w * 2); b_f = better[f][:h].reshape(-1, 2)[:, ::-1].reshape(h, w * 2)
            else: lo_f = lo[f][:h]; b_f = better[f][:h]
            assert abs(psnr_yuy2(img, src) - psnr_yuy2(lo_f, src)) < 0.1
            assert psnr_yuy2(b_f, src) > psnr_yuy2(img, src) + 1.0


def _interlaced_frames(w, h, n, fmt, flicker=False):
    """Frames whose two fields disagree (the second field shifted sideways; `flicker`: field flicker strong enough for quantized field differences beyond the peak
    threshold, i.e. peak tables behind the difference-coded bands)."""
    out = []
    for i in range(n):
        f = (field_flicker_frame(w, h)[0] if flicker else synth_yuy2(w, h, 90 + i)[0]).reshape(h, w * 2).copy()
        f[1::2] = np.roll(f[1::2], 4 + 2 * (i % 3), axis=1)
        if flicker: f = np.clip(f.astype(np.int32) + np.random.default_rng(i).integers(-5, 6, f.shape), 0, 255).astype(np.uint8)
        if fmt == PIX_2VUY: f = f.reshape(h, w, 2)[:, :, ::-1].reshape(h, w * 2)
        out.append(np.ascontiguousarray(f).reshape(-1))
    return out


@pytest.mark.parametrize("w,h,fmt,flicker", [(320, 240, PIX_YUY2, 0), (720, 486, PIX_2VUY, 0), (336, 252, PIX_YUY2, 1), (400, 120, PIX_YUY2, 0), (1920, 1080, PIX_YUY2, 0), (720, 480, PIX_YUY2, 1)])
def test_interlaced_group_samples_equal_reference(w, h, fmt, flicker):
    """CFHD_ENCODING_FLAGS_YUV_INTERLACED | _2FRAME_GOP: level 1 of both frames is the frame transform of interlaced intra frames (Codec/encoder.c:2950-2979), the
    horizontal-lowpass / temporal-highpass band of both frame wavelets -- subbands 12 and 15 -- is difference coded in code set 18 with a peak table where needed
    (encoder.c:6143-6154), the quantizer tables are the interlaced ones (quantize.c:492).  Oracle transform + the product's host writer = the reference's samples."""
    kind = 2 if fmt == PIX_2VUY else 1
    frames = _interlaced_frames(w, h, 4, fmt, bool(flicker))
    ref = ref_encode_frames(frames, w * 2, w, h, pixfmt=fmt, flags=ENCODING_FLAGS_2FRAME_GOP | 1)
    gp = GopPlan(w, h, pixkind=kind, interlaced=1)
    assert [len(s) for s in ref[0::2]] == [40, 24]
    assert product_write_gop_host(gp, 1) == ref[0]
    peaks = 0
    for g in range(2):
        s = ref[2 * g + 1]
        co = oracle_forward_gop(gp, frames[2 * g], frames[2 * g + 1], w * 2, uyvy=int(fmt == PIX_2VUY))
        off, n = first_metadata_chunk(s)
        mine = product_write_gop_host(gp, 0, co, 2 * g + 1, meta_global=s[off:off + n])
        assert len(mine) == len(s), (len(mine), len(s))
        assert mine == s, "group %d differs at byte %d" % (g, next(i for i in range(len(s)) if mine[i] != s[i]))
        peaks += sum(int.from_bytes(s[i + 2:i + 4], "big") != 0 for i in range(0, len(s) - 4, 4) if s[i:i + 2] == b"\xff\xb6")      # TAG_PEAK_LEVEL (optional)
    if flicker: assert peaks, "the flicker frames were expected to carry peak tables"


@pytest.mark.parametrize("w,h,fmt,flicker", [(320, 240, PIX_YUY2, 0), (336, 252, PIX_YUY2, 1), (720, 480, PIX_2VUY, 0), (1920, 1080, PIX_YUY2, 0)])
def test_interlaced_group_inverse_model_equals_the_reference_group_decoder(w, h, fmt, flicker):
    """The same pin for interlaced groups: reference samples -> host parser / decoder (code set 18, peak tables, running sums along the rows of subbands 12 and 15)
    -> the group's inverse with the inverse FRAME transform as its last level -- every byte of both frames the reference's group decoder returns lies inside the
    oracle's dither interval, and the PSNR agrees to 0.1 dB."""
    kind = 2 if fmt == PIX_2VUY else 1
    frames = _interlaced_frames(w, h, 4 if w < 1920 else 2, fmt, bool(flicker))
    samples = ref_encode_frames(frames, w * 2, w, h, pixfmt=fmt, flags=ENCODING_FLAGS_2FRAME_GOP | 1)
    gp = GopPlan(w, h, pixkind=kind, interlaced=1)
    ngroups = len(frames) // 2
    models = []
    for g in range(ngroups):
        co = oracle_decode_group(samples[2 * g + 1], gp)
        models.append((oracle_inverse_gop(gp, co, 0, uyvy=int(kind == 2)), oracle_inverse_gop(gp, co, 1, uyvy=int(kind == 2))))
    def outside(got):
        return [(g, f, int((~((img == models[g][0][f][:h]) | (img == models[g][1][f][:h]))).sum())) for g, pair in enumerate(got) for f, img in enumerate(pair) if img is not None]
    for attempt in range(3):
        got = ref_decode_group_frames(samples, w, h, fmt)
        assert len(got) == ngroups
        if not any(n for _, _, n in outside(got)): break
    for g, f, n in outside(got): assert n == 0, "group %d frame %d: %d bytes of the reference decoder's picture lie outside the oracle's dither interval" % (g, f, n)
    for g, pair in enumerate(got):
        for f, img in enumerate(pair):
            if img is None: continue
            src = frames[2 * g + f].reshape(h, w * 2); lo_f = models[g][0][f][:h]; hi_f = models[g][1][f][:h]
            if kind == 2: sw = lambda a: a.reshape(-1, 2)[:, ::-1].reshape(h, w * 2); src, img, lo_f, hi_f = sw(src), sw(img), sw(lo_f), sw(hi_f)
            ends = (psnr_yuy2(lo_f, src), psnr_yuy2(hi_f, src))      # any picture inside the interval lies between its two ends (to the printed 0.1 dB)
            assert min(ends) - 0.1 < psnr_yuy2(img, src) < max(ends) + 0.1


@pytest.mark.parametrize("w,h,fmt,interlaced,flicker", [(320, 240, PIX_YUY2, 0, 0), (720, 486, PIX_2VUY, 0, 0), (336, 252, PIX_YUY2, 1, 1), (720, 480, PIX_2VUY, 1, 0),
                                                         (1920, 1080, PIX_YUY2, 0, 0), (1920, 1080, PIX_YUY2, 1, 0)])
def test_oracle_group_walk_equals_product_host_decoder(w, h, fmt, interlaced, flicker):
    """The decode gates of the group tests feed the oracle's inverse with coefficients decoded by the oracle alone (oracle/cfhd_oracle_ent.c orc_decode_group: its own walk over
    the group sample -- six wavelets, the empty band of the temporal wavelet, the raw 16-bit band of the temporal highpass with the band end code behind it, code sets 17 / 18, peak
    tables, difference coding; no size field is consulted).  This test ties that decoder to the product's host parser + VLC decoder on reference-encoded groups of every kind
    the tests use: both must give the same pyramid, coefficient for coefficient -- so a defect shared by the product's host and GPU group decoders cannot hide behind a gate."""
    assert have_ref(), "oracle/_ref/libcfhd_ref.so is missing"
    kind = 2 if fmt == PIX_2VUY else 1
    if interlaced: frames = _interlaced_frames(w, h, 4, fmt, bool(flicker))
    elif w >= 1920: frames = qbist_frames(10, 4, w, h, fmt)[0]
    else:
        frames = [synth_yuy2(w, h, 40 + i)[0] for i in range(4)]
        if fmt == PIX_2VUY: frames = [f.reshape(-1, 2)[:, ::-1].reshape(-1).copy() for f in frames]
    samples = ref_encode_frames(frames, w * 2, w, h, pixfmt=fmt, flags=ENCODING_FLAGS_2FRAME_GOP | interlaced)
    gp = GopPlan(w, h, pixkind=kind, interlaced=interlaced)
    for g in (1, 3):
        a = host_decode_group(samples[g], gp); b = oracle_decode_group(samples[g], gp)
        assert np.array_equal(a, b), "group sample %d: %d coefficients differ" % (g, int((a != b).sum()))
        if flicker: assert b"\xff\xb5" in samples[g]

