"""Product host layer (plan geometry, quantizer derivation, sample writer, host VLC, parser) checked on CPU:
oracle forward transform -> product sample writer must reproduce the reference encoder's sample byte for byte."""
import ctypes
import numpy as np
import pytest
from cfhd_testlib import *

needs_ref = [pytest.mark.ref, pytest.mark.skipif(not have_ref(), reason="reference .so not built")]


@pytest.mark.parametrize("w,h", [(320, 240), (640, 480), (336, 252), (1920, 1080)])
def test_sample_bytes_equal_reference_synthetic(w, h):
    for m in needs_ref: pass
    if not have_ref(): pytest.skip("reference .so not built")
    frame, pitch = synth_yuy2(w, h, 3)
    refs = ref_encode_frames([frame, frame], pitch, w, h)
    plan = Plan(w, h)
    coeffs = oracle_forward_yuv422(plan, frame, pitch)
    for i, rs in enumerate(refs):
        off, n = first_metadata_chunk(rs)
        mine = product_write_sample_host(plan, coeffs, i + 1, meta_global=rs[off:off + n])
        assert len(mine) == len(rs)
        assert mine == rs


def test_sample_bytes_equal_reference_qbist_1080p():
    if not have_ref(): pytest.skip("reference .so not built")
    frames, pitch = qbist_frames(10, 2)
    refs = ref_encode_frames(frames, pitch, 1920, 1080)
    assert len(refs[0]) == 310392          # SURVEY.md section 6 [probe]
    plan = Plan(1920, 1080)
    for i, (f, rs) in enumerate(zip(frames, refs)):
        coeffs = oracle_forward_yuv422(plan, f, pitch)
        off, n = first_metadata_chunk(rs)
        mine = product_write_sample_host(plan, coeffs, i + 1, meta_global=rs[off:off + n])
        assert mine == rs


def test_quant_tables_known_answers():
    # SURVEY.md section 8 (a14) [probe]: 1080p YUY2 FILMSCAN1
    plan = Plan(1920, 1080)
    luma = [plan.band[(0, lv, b)]["quant"] for lv in (2, 1, 0) for b in (1, 2, 3)]
    chroma = [plan.band[(1, lv, b)]["quant"] for lv in (2, 1, 0) for b in (1, 2, 3)]
    scale = [plan.band[(0, lv, b)]["scale"] for lv in (2, 1, 0) for b in (1, 2, 3)]
    assert luma == [24, 24, 12, 6, 6, 3, 24, 24, 36]
    assert chroma == [24, 24, 12, 6, 6, 3, 24, 24, 48]
    assert scale == [32, 32, 16, 8, 8, 4, 2, 2, 1]
    assert plan.prescale == [0, 2, 0] and plan.mpq == 2
    assert (plan.band[(0, 2, 0)]["width"], plan.band[(0, 2, 0)]["height"]) == (240, 135)
    assert (plan.band[(1, 2, 0)]["width"], plan.band[(1, 2, 0)]["height"]) == (120, 135)


@pytest.mark.parametrize("w,h", [(320, 240), (1920, 1080)])
def test_parse_and_host_decode_roundtrip(w, h):
    """write_sample -> parse_sample + host VLC decode gives back sign*expand(compand(|q|))*quant for every band."""
    frame, pitch = synth_yuy2(w, h, 5)
    plan = Plan(w, h)
    coeffs = oracle_forward_yuv422(plan, frame, pitch)
    sample = product_write_sample_host(plan, coeffs, 1)
    out = np.zeros(plan.coeff_elems, dtype=np.int16)
    info = (ctypes.c_int * 8)()
    s = np.frombuffer(sample, dtype=np.uint8).copy()
    rc = hooks().cfhd_amd_decode_bands_host(p8(s), len(sample), 1, p16(out), out.size, info, 0)
    assert rc == 0
    assert list(info)[:5] == [w, plan.height, h, 3, 10]
    # expected: companding curve applied by the encoder LUT, expanded and dequantized by the decoder
    idx = np.arange(256)
    expand = idx + ((idx.astype(np.int64) ** 3 * 768) >> 24)
    inv = np.zeros(1025, dtype=np.int64)
    inv[np.minimum(expand[1:], 1023)] = idx[1:]
    inv = np.maximum.accumulate(inv)
    for c in range(3):
        assert np.array_equal(plan.view(out, c, 2, 0), plan.view(coeffs, c, 2, 0))
        for lv in range(3):
            for b in (1, 2, 3):
                q = plan.view(coeffs, c, lv, b).astype(np.int64)
                mag = np.minimum(np.abs(q), 1023)
                want = np.sign(q) * expand[inv[mag]] * plan.band[(c, lv, b)]["quant"]
                assert np.array_equal(plan.view(out, c, lv, b), want.astype(np.int16)), (c, lv, b)


@pytest.mark.parametrize("w,h", [(192, 96), (320, 240), (1920, 1080)])
def test_rg48_rgb444_sample_bytes_equal_reference(w, h):
    """RG48 -> RGB 4:4:4 12-bit.  Unpack (>> 4, planes G, R, B) + oracle transform (prescale table {0,2,2}) +
    product quantizer tables and sample writer = the reference encoder's sample, byte for byte."""
    if not have_ref(): pytest.skip("reference .so not built")
    frames, pitch = qbist_frames(10, 1, w, h, PIX_RG48)
    rs = ref_encode_frames(frames, pitch, w, h, PIX_RG48, encoded=ENCODED_RGB444)[0]
    plan = Plan(w, h, pixkind=PIXKIND["RG48"], enc=3)
    assert plan.precision == 12 and plan.prescale[:3] == [0, 2, 2] and plan.num_channels == 3
    coeffs = oracle_forward_planes(plan, rg48_planes(frames[0], pitch, w, h))
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, coeffs, 1, meta_global=rs[off:off + n], input_format=COLOR_FORMAT_RG48, color_space=0)
    assert mine == rs


@pytest.mark.parametrize("w,h", [(192, 96), (640, 360)])
def test_b64a_rgba4444_sample_bytes_equal_reference(w, h):
    """SURVEY 8a9 / config C (encode side): b64a -> RGBA 4:4:4:4 12-bit with the companded alpha plane; R, B and A take the chroma
    quantizer tables because b64a's colour format code is below COLOR_FORMAT_BAYER (encoder.c:1141), and the quality word carries
    the "4444 instead of 444" mark 0x20000000 (SampleEncoder.cpp:250-257)."""
    if not have_ref(): pytest.skip("reference .so not built")
    frames, pitch = qbist_frames(10, 1, w, h, PIX_B64A, alpha=1)
    px = np.frombuffer(frames[0].tobytes(), dtype=np.uint16).reshape(h, pitch // 2).copy()
    px[:, 0: w * 4: 4] = ((np.arange(h)[:, None] * 523 + np.arange(w)[None, :] * 97) % 65536).astype(np.uint16)   # a real alpha ramp (Qbist's is opaque)
    frame = px.reshape(-1).view(np.uint8).copy()
    rs = ref_encode_frames([frame], pitch, w, h, PIX_B64A, encoded=ENCODED_RGBA4444)[0]
    plan = Plan(w, h, pixkind=PIXKIND["b64a"], enc=4, quality=QUALITY_FILMSCAN1 | 0x20000000)
    assert plan.num_channels == 4 and plan.precision == 12
    coeffs = oracle_forward_planes(plan, b64a_planes(frame, pitch, w, h))
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, coeffs, 1, meta_global=rs[off:off + n], input_format=COLOR_FORMAT_B64A, color_space=0)
    assert mine == rs


@pytest.mark.parametrize("w,h", [(192, 96), (640, 352), (1920, 1080)])
def test_byr4_bayer_sample_bytes_equal_reference(w, h):
    """SURVEY 8a9 / config D (Bayer half): BYR4 mosaic -> four half-resolution planes through the log-90 encode curve -> Bayer sample
    (RGB quality bits pinned for the quantizer only, encoder.c:2638).  Pins orc_byr4_* and the product's tables/writer on the reference."""
    if not have_ref(): pytest.skip("reference .so not built")
    mosaic = synth_bayer(w, h, 3)
    frame = mosaic.reshape(-1).view(np.uint8).copy()
    rs = ref_encode_frames([frame], w * 2, w, h, PIX_BYR4, encoded=ENCODED_BAYER)[0]
    plan = Plan(w, h, pixkind=PIXKIND["BYR4"], enc=2)
    assert plan.num_channels == 4 and plan.precision == 12
    coeffs = oracle_forward_planes(plan, byr4_planes(mosaic))
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, coeffs, 1, meta_global=rs[off:off + n], input_format=COLOR_FORMAT_BYR4, color_space=0)
    assert mine == rs


@pytest.mark.parametrize("w,h,encoded", [(320, 240, ENCODED_RGBA4444), (336, 252, ENCODED_RGB444), (320, 240, ENCODED_YUV422), (1920, 1080, ENCODED_RGBA4444)])
def test_rg64_sample_bytes_equal_reference(w, h, encoded):
    """RG64 (16-bit words R, G, B, A; frame.c ConvertRGBA64ToFrame16s) to all three encoded formats the reference offers for it: planes G, R, B = word >> 4, alpha
    curved as b64a's, rows below the picture repeat the last one; input format 121 -- above COLOR_FORMAT_BAYER, so every plane takes the full-resolution
    quantizer tables (RG48's, not b64a's) --, quality marks as b64a's (0x2000 for 4:4:4:4, 0x0800 for 4:2:2, none for RGB 4:4:4)."""
    if not have_ref(): pytest.skip("reference .so not built")
    frame, pitch, words = rg64_frame(10, w, h)
    rs = ref_encode_frames([frame], pitch, w, h, fourcc("RG64"), encoded=encoded)[0]
    if encoded == ENCODED_YUV422:
        plan = Plan(w, h, pixkind=PIXKIND["RG64"], enc=1)
        planes = oracle_rgb16_to_yuv422_planes(words.reshape(h, w * 4), 4, 0, w, h)
    else:
        plan = Plan(w, h, pixkind=PIXKIND["RG64"], enc=4 if encoded == ENCODED_RGBA4444 else 3)
        a = words[:, :, 3].astype(np.int32) >> 4
        a = np.where((a > 0) & (a < 4095), ((a * 223 + 128) >> 8) + 256, a)
        planes = [(words[:, :, 1] >> 4).astype(np.int16), (words[:, :, 0] >> 4).astype(np.int16), (words[:, :, 2] >> 4).astype(np.int16), a.astype(np.int16)][: plan.num_channels]
        assert plan.band[(1, 0, 3)]["quant"] == Plan(w, h, pixkind=PIXKIND["RG48"], enc=3).band[(1, 0, 3)]["quant"]
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, planes), 1, meta_global=rs[off:off + n], input_format=121, color_space=2 if encoded == ENCODED_YUV422 else 0)
    assert len(mine) == len(rs)
    diff = [i for i, (x, y) in enumerate(zip(mine, rs)) if x != y]
    mark = {ENCODED_RGBA4444: 0x20, ENCODED_YUV422: 0x08}.get(encoded)
    assert diff == ([90] if mark else []) and (not mark or rs[90] == mark)      # (the host writer of this test passes no quality word)


@pytest.mark.parametrize("w,h", [(320, 240), (640, 360), (1920, 1080)])
def test_byr5_sample_bytes_equal_reference(w, h):
    """BYR5 (12-bit Bayer, CFHDTypes.h: "packed line of 8-bit then line a 4-bit reminder") -> CFHD_ENCODED_FORMAT_BAYER: the reference unpacks the four
    components of every row pair without an encode curve (frame.c:5473 ConvertBYR5ToFrame16s) and goes on as for BYR4; input format 105 in the header.
    Oracle unpack + oracle plane transform + product syntax = reference sample, byte for byte (1080: the last row pair repeated below the picture)."""
    if not have_ref(): pytest.skip("reference .so not built")
    frame = pack_byr5(synth_bayer(w, h, 3))
    rs = ref_encode_frames([frame], w * 2, w, h, fourcc("BYR5"), encoded=ENCODED_BAYER)[0]
    plan = Plan(w, h, pixkind=PIXKIND["BYR5"], enc=2)
    assert plan.num_channels == 4 and plan.precision == 12
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, byr5_planes(frame, w // 2, h // 2)), 1, meta_global=rs[off:off + n], input_format=105, color_space=0)
    assert mine == rs


@pytest.mark.parametrize("w,h", [(320, 240), (720, 486)])
def test_yu64_sample_bytes_equal_reference(w, h):
    """YU64 (16-bit 4:2:2 words Y0 C1 Y1 C2) -> YUV 4:2:2 sample: the reference unpacks every word >> 6 into three planes (channel 1 = the
    second word of a pixel pair, channel 2 = the fourth; frame.c:1556, convert.c:3345 / :14375) and runs the plane transform on them;
    input format 12 in the header.  Pins that reading on the reference: oracle plane transform + product syntax = reference sample."""
    if not have_ref(): pytest.skip("reference .so not built")
    frames, pitch = qbist_frames(10, 1, w, h, PIX_YU64)
    words = np.frombuffer(frames[0].tobytes(), np.uint16).reshape(h, pitch // 2)[:, : w * 2]
    rs = ref_encode_frames(frames, pitch, w, h, PIX_YU64, encoded=ENCODED_YUV422)[0]
    plan = Plan(w, h, pixkind=PIXKIND["YU64"], enc=1)
    planes = [(words[:, 0::2] >> 6).astype(np.int16), (words[:, 1::4] >> 6).astype(np.int16), (words[:, 3::4] >> 6).astype(np.int16)]
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, planes), 1, meta_global=rs[off:off + n], input_format=12, color_space=2)
    assert mine == rs


@pytest.mark.parametrize("w,h,name", [(192, 96, "RG48"), (320, 240, "b64a"), (336, 252, "RG48"), (720, 486, "b64a"), (1920, 1080, "RG48")])
def test_deep_rgb_to_yuv422_sample_bytes_equal_reference(w, h, name):
    """RG48 / b64a encoded as YUV 4:2:2 (rows of TestCFHD's format table): the reference converts every pixel pair with its integer
    709 matrix (Codec/frame.c:6731 ConvertAnyDeep444to422: 10-bit Y per pixel, chroma of the pair averaged, alpha dropped) and goes on as for
    YU64; the quality word carries 0x0800 in its upper half (an encoded format other than the input's default, SampleEncoder.cpp:218) and the
    header the input's format code.  Oracle conversion + oracle plane transform + product syntax = reference sample, byte for byte, heights
    that are not multiples of 8 included (the conversion repeats the last picture row, :7176)."""
    if not have_ref(): pytest.skip("reference .so not built")
    fmt = PIX_RG48 if name == "RG48" else PIX_B64A
    wpp = 3 if name == "RG48" else 4
    frames, pitch = qbist_frames(10, 1, w, h, fmt, alpha=int(name == "b64a"))
    words = np.frombuffer(frames[0].tobytes(), np.uint16).reshape(h, pitch // 2)[:, : w * wpp]
    rs = ref_encode_frames(frames, pitch, w, h, fmt, encoded=ENCODED_YUV422)[0]
    plan = Plan(w, h, pixkind=PIXKIND[name], enc=1)          # the product's own plan for this combination: 4:2:2 geometry AND 4:2:2 quantizer tables
    yu64 = Plan(w, h, pixkind=PIXKIND["YU64"], enc=1)        # (round 3's first hardware run: RG48 took the full-resolution chroma tables of RGB 4:4:4)
    assert all(plan.band[k]["quant"] == yu64.band[k]["quant"] for k in plan.band)
    planes = oracle_rgb16_to_yuv422_planes(words, wpp, 0 if name == "RG48" else 1, w, h)
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, planes), 1, meta_global=rs[off:off + n], input_format=120 if name == "RG48" else 30, color_space=2)
    assert len(mine) == len(rs)
    diff = [i for i, (a, b) in enumerate(zip(mine, rs)) if a != b]
    assert diff == [90] and rs[88:92] == bytes([0xff, 0xaf, 0x08, 0x00])      # the quality mark (the host writer of this test passes no quality word)


@pytest.mark.parametrize("w,h,name,flags", [(320, 240, "RG24", 0), (336, 252, "BGRA", 0), (208, 120, "BGRa", 0), (336, 252, "BGRa", 4), (320, 240, "BGRA", 0x100),
                                              (320, 240, "RG24", 0x104), (1920, 1080, "RG24", 0)])
def test_rgb8_to_yuv422_sample_bytes_equal_reference(w, h, name, flags):
    """RG24 / BGRA / BGRa encoded as YUV 4:2:2 -- the default encoded format of these inputs and three rows of TestCFHD's format table.  The reference
    converts row by row (frame.c:378 ConvertRGB32to10bitYUVFrame: 13-bit matrix on 15-bit samples, the EVEN pixel's chroma kept, Y 64 / chroma 512
    in the rows below the picture) and goes on as for YU64; the quality word carries 0x01a0 in its upper half (encoder.c:2351).  Oracle conversion +
    oracle plane transform + product syntax = reference sample, byte for byte: both row orders, the four matrices (flags 4 = 601, 0x100 = video
    range), a height that is not a multiple of 8."""
    if not have_ref(): pytest.skip("reference .so not built")
    fmt = {"RG24": PIX_RG24, "BGRA": PIX_BGRA, "BGRa": PIX_BGRa}[name]
    frames, pitch = qbist_frames(10, 1, w, h, fmt)
    rs = ref_encode_frames(frames, pitch, w, h, fmt, encoded=ENCODED_YUV422, flags=flags)[0]
    plan = Plan(w, h, pixkind=PIXKIND[name], enc=1)
    yu64 = Plan(w, h, pixkind=PIXKIND["YU64"], enc=1)
    assert all(plan.band[k]["quant"] == yu64.band[k]["quant"] for k in plan.band)
    cs = (1 if flags & 0x100 else 0) + (2 if flags & 4 else 0)
    planes = oracle_rgb8_to_yuv422_planes(frames[0], pitch, 3 if name == "RG24" else 4, int(name == "BGRa"), w, h, 2 * plan.band[(0, 0, 0)]["height"], cs)
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, planes), 1, meta_global=rs[off:off + n], input_format={"RG24": 7, "BGRA": 32, "BGRa": 9}[name],
                                     color_space=(1 if flags & 4 else 2) | (4 if flags & 0x100 else 0))
    assert len(mine) == len(rs)
    diff = [i for i, (a, b) in enumerate(zip(mine, rs)) if a != b]
    assert diff == [90, 91] and rs[88:92] == bytes([0xff, 0xaf, 0x01, 0xa0])      # the quality mark (the host writer of this test passes no quality word)


@pytest.mark.parametrize("w,h,name", [(320, 240, "BGRA"), (336, 256, "BGRa"), (720, 480, "BGRa"), (1920, 1088, "BGRA")])
def test_rgba8_to_rgba4444_sample_bytes_equal_reference(w, h, name):
    """BGRA / BGRa encoded as RGBA 4:4:4:4 (two rows of TestCFHD's table): planes G, R, B = byte << 4 and the alpha byte << 4 curved like b64a's, but
    with the interval ending at 255 << 4 (frame.c:6415 ConvertRGBAtoRGBA64); the frame goes on as COLOR_FORMAT_RG64 -- above COLOR_FORMAT_BAYER, so all
    four planes take the full-resolution quantizer tables (b64a's R, B, A planes take the chroma tables; encoder.c:1141) -- and the quality word reads
    0x21a0.  The rows below a picture whose height is not a multiple of 8 are never written by the reference: zeros from a fresh allocation, whatever the
    heap held otherwise (336 x 252 gives samples of different sizes from call to call) -- parity is claimed for heights that are multiples of 8; the
    library writes zeros there."""
    if not have_ref(): pytest.skip("reference .so not built")
    fmt = {"BGRA": PIX_BGRA, "BGRa": PIX_BGRa}[name]
    frames, pitch = qbist_frames(10, 1, w, h, fmt, alpha=1)
    rs = ref_encode_frames(frames, pitch, w, h, fmt, encoded=ENCODED_RGBA4444)[0]
    px = np.frombuffer(frames[0].tobytes(), np.uint8).reshape(h, pitch)[:, : w * 4].reshape(h, w, 4).astype(np.int32)
    if name == "BGRA": px = px[::-1]
    a = px[:, :, 3] << 4
    a = np.where((a > 0) & (a < 4080), ((a * 223 + 128) >> 8) + 256, a)
    plan = Plan(w, h, pixkind=PIXKIND[name], enc=4)
    b64a = Plan(w, h, pixkind=PIXKIND["b64a"], enc=4)
    assert plan.num_channels == 4 and plan.band[(1, 0, 3)]["quant"] < b64a.band[(1, 0, 3)]["quant"]
    eh = 2 * plan.band[(0, 0, 0)]["height"]
    planes = [np.vstack([p.astype(np.int16), np.zeros((eh - h, w), np.int16)]) for p in ((px[:, :, 1] << 4), (px[:, :, 2] << 4), (px[:, :, 0] << 4), a)]
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, planes), 1, meta_global=rs[off:off + n], input_format={"BGRA": 32, "BGRa": 9}[name], color_space=0)
    assert len(mine) == len(rs)
    diff = [i for i, (x, y) in enumerate(zip(mine, rs)) if x != y]
    assert diff == [90, 91] and rs[88:92] == bytes([0xff, 0xaf, 0x21, 0xa0])


@pytest.mark.parametrize("w,h", [(192, 96), (320, 240), (400, 120), (720, 480)])
def test_v210_sample_bytes_equal_reference(w, h):
    """v210 (10-bit 4:2:2) -> YUV 4:2:2 sample, input format 10.  Pins the reading of the reference's unpack (convert.c:3968), including its
    oddity that the scalar loop behind the last whole 48 pixels repeats a Cr sample (widths 320 and 400 have such a tail).  Heights that are
    not multiples of 8 are left out: the reference never writes the rows below the display height (frame.c:1481 stops there) and transforms
    whatever its allocation holds -- the same frame gives samples of different sizes from call to call (test below)."""
    if not have_ref(): pytest.skip("reference .so not built")
    frame, pitch, Y, Cb, Cr = synth_v210(w, h, 5)
    rs = ref_encode_frames([frame], pitch, w, h, PIX_V210, encoded=ENCODED_YUV422)[0]
    plan = Plan(w, h, pixkind=PIXKIND["v210"], enc=1)
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, v210_planes(plan, Y, Cb, Cr)), 1, meta_global=rs[off:off + n], input_format=10, color_space=2)
    assert mine == rs


def test_v210_reference_is_not_reproducible_for_padded_heights():
    """720 x 486 v210: the reference's sample depends on what its heap held before (uninitialised rows 486..487 of the planes).  Documents
    why parity for v210 is claimed for heights that are multiples of 8 only; the product zeroes those rows."""
    if not have_ref(): pytest.skip("reference .so not built")
    w, h = 720, 486
    frame, pitch, Y, Cb, Cr = synth_v210(w, h, 5)
    sizes = set()
    for k in range(4):
        sizes.add(len(ref_encode_frames([frame], pitch, w, h, PIX_V210, encoded=ENCODED_YUV422)[0]))
        ref_encode_frames([synth_yuy2(640 + 64 * k, 360, k)[0]], (640 + 64 * k) * 2, 640 + 64 * k, 360)      # stir the heap
    plan = Plan(w, h, pixkind=PIXKIND["v210"], enc=1)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, v210_planes(plan, Y, Cb, Cr)), 1, input_format=10, color_space=2)
    assert len(mine) > 0
    if len(sizes) == 1: pytest.skip("the reference happened to see the same memory four times")


@pytest.mark.parametrize("w,h", [(320, 240), (720, 480)])
def test_rg24_rgb444_sample_bytes_equal_reference(w, h):
    """RG24 (8-bit B, G, R bytes, bottom row first) -> RGB 4:4:4: the reference lifts every byte to 12 bits (<< 4) into planes G, R, B
    (frame.c:6173 ConvertRGBtoRGB48) and goes on as for RG48; input format 7, the quality word marked 0x09a0 in its upper half.  Heights
    that are multiples of 8 only: the reference's conversion stops at the display height and transforms uninitialised rows below it."""
    if not have_ref(): pytest.skip("reference .so not built")
    frames, pitch = qbist_frames(10, 1, w, h, PIX_RG24)
    rs = ref_encode_frames(frames, pitch, w, h, PIX_RG24, encoded=ENCODED_RGB444)[0]
    px = frames[0].reshape(h, pitch)[:, : w * 3].reshape(h, w, 3)[::-1]
    plan = Plan(w, h, pixkind=PIXKIND["RG24"], enc=3, quality=QUALITY_FILMSCAN1 | 0x09a00000)
    planes = [px[:, :, 1].astype(np.int16) << 4, px[:, :, 2].astype(np.int16) << 4, px[:, :, 0].astype(np.int16) << 4]
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, planes), 1, meta_global=rs[off:off + n], input_format=7, color_space=0)
    assert mine == rs


@pytest.mark.parametrize("name,flip,code", [("BGRA", 1, 32), ("BGRa", 0, 9)])
def test_bgra_rgb444_sample_bytes_equal_reference(name, flip, code):
    """8-bit B, G, R, A pixels -> RGB 4:4:4 (frame.c:6286 ConvertRGBAtoRGB48): as RG24 with four bytes per pixel, alpha dropped; 'BGRA' frames
    are stored bottom row first, 'BGRa' top row first; input format codes 32 / 9."""
    if not have_ref(): pytest.skip("reference .so not built")
    w, h = 320, 240
    fmt = fourcc(name)
    frames, pitch = qbist_frames(10, 1, w, h, fmt, alpha=1)
    rs = ref_encode_frames(frames, pitch, w, h, fmt, encoded=ENCODED_RGB444)[0]
    px = frames[0].reshape(h, pitch)[:, : w * 4].reshape(h, w, 4)
    if flip: px = px[::-1]
    plan = Plan(w, h, pixkind=PIXKIND[name], enc=3, quality=QUALITY_FILMSCAN1 | 0x09a00000)
    planes = [px[:, :, 1].astype(np.int16) << 4, px[:, :, 2].astype(np.int16) << 4, px[:, :, 0].astype(np.int16) << 4]
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, planes), 1, meta_global=rs[off:off + n], input_format=code, color_space=0)
    assert mine == rs


@pytest.mark.parametrize("name", sorted(RGB10_FORMATS))
def test_rgb10_rgb444_sample_bytes_equal_reference(name):
    """r210 / DPX0 / AB10 / AR10 (10-bit RGB in one 32-bit word) -> RGB 4:4:4: every field << 2 into planes G, R, B, rows top-down, input
    format codes 123 / 128 / 125 / 124 (the reference runs a fused unpack + transform, wavelet.c:3595; its result for a height that is a
    multiple of 8 is the plane transform of these planes)."""
    if not have_ref(): pytest.skip("reference .so not built")
    w, h = 320, 240
    order, shifts, code = RGB10_FORMATS[name]
    fmt = fourcc(name)
    frames, pitch = qbist_frames(10, 1, w, h, fmt)
    rs = ref_encode_frames(frames, pitch, w, h, fmt, encoded=ENCODED_RGB444)[0]
    words = np.frombuffer(frames[0].tobytes(), order + "u4").reshape(h, pitch // 4)[:, :w]
    r, g, b = [((words >> s) & 0x3ff).astype(np.int16) << 2 for s in shifts]
    plan = Plan(w, h, pixkind=PIXKIND[name], enc=3)
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, [g, r, b]), 1, meta_global=rs[off:off + n], input_format=code, color_space=0)
    assert mine == rs


def test_b64a_rgb444_sample_bytes_equal_reference():
    """b64a -> RGB 4:4:4 (the reference's default for b64a input): alpha dropped, planes G, R, B = words >> 4, with the quantizer tables the
    colour format code 30 selects (the chroma tables for R and B, as for 4:4:4:4) and no mark in the quality word."""
    if not have_ref(): pytest.skip("reference .so not built")
    w, h = 320, 240
    frames, pitch = qbist_frames(10, 1, w, h, PIX_B64A, alpha=1)
    rs = ref_encode_frames(frames, pitch, w, h, PIX_B64A, encoded=ENCODED_RGB444)[0]
    px = np.frombuffer(frames[0].tobytes(), np.uint16).reshape(h, pitch // 2)[:, : w * 4].reshape(h, w, 4)
    plan = Plan(w, h, pixkind=PIXKIND["b64a"], enc=3)
    planes = [(px[:, :, 2] >> 4).astype(np.int16), (px[:, :, 1] >> 4).astype(np.int16), (px[:, :, 3] >> 4).astype(np.int16)]
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, oracle_forward_planes(plan, planes), 1, meta_global=rs[off:off + n], input_format=30, color_space=0)
    assert mine == rs


def test_byr4_pitch_is_ignored_by_the_reference():
    """The reference's BYR4 unpack (frame.c:5376) walks the mosaic as tightly packed rows whatever pitch the caller passes.  A drop-in has to
    read the same bytes: the product does (EncodeBatch::upload_frame), this pins the behaviour on the reference itself."""
    if not have_ref(): pytest.skip("reference .so not built")
    w, h, pitch = 192, 96, 192 * 2 + 48
    rng = np.random.default_rng(4)
    buf = rng.integers(0, 256, pitch * h).astype(np.uint8)
    rs = ref_encode_frames([buf], pitch, w, h, PIX_BYR4, encoded=ENCODED_BAYER)[0]
    plan = Plan(w, h, pixkind=PIXKIND["BYR4"], enc=2)
    off, n = first_metadata_chunk(rs)
    tight = np.ascontiguousarray(buf[: 2 * w * h]).view(np.uint16).reshape(h, w)
    strided = np.ascontiguousarray(buf.reshape(h, pitch)[:, : 2 * w]).view(np.uint16)
    write = lambda m: product_write_sample_host(plan, oracle_forward_planes(plan, byr4_planes(m)), 1, meta_global=rs[off:off + n], input_format=COLOR_FORMAT_BYR4, color_space=0)
    assert write(tight) == rs
    assert write(strided) != rs


def test_bayer_curve_table_equals_oracle():
    want = np.zeros(1 << 14, np.uint16); got = np.zeros(1 << 14, np.uint16)
    oracle().orc_byr4_log90_curve.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    oracle().orc_byr4_log90_curve(12, 14, want.ctypes.data_as(ctypes.c_void_p))
    hooks().cfhd_amd_bayer_curve.argtypes = [ctypes.c_int, ctypes.c_void_p]
    hooks().cfhd_amd_bayer_curve(12, got.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(got, want)


def every_column_flicker_frame(w, h):
    """Field flicker that changes sign every second luma pixel: every step between neighbouring coefficients of the luma field-difference band is a peak."""
    f = np.zeros((h, w * 2), np.uint8)
    f[:, 1::2] = 128
    phase = (np.arange(w) // 2) % 2
    f[0::2, 0::2] = np.where(phase, 255, 0)
    f[1::2, 0::2] = np.where(phase, 0, 255)
    return f.reshape(-1).copy(), w * 2


@pytest.mark.parametrize("w,h,kind", [(192, 96, "smooth"), (720, 480, "smooth"), (720, 486, "smooth"), (1920, 1080, "qbist"), (320, 64, "peaks"), (1024, 528, "allpeaks")])
def test_interlaced_sample_bytes_equal_reference(w, h, kind):
    """SURVEY 8a8 (encode side, host twin): CFHD_ENCODING_FLAGS_YUV_INTERLACED.  Oracle frame transform + product quantizer tables
    (interlaced variants) + product sample writer (no SAMPLE_FLAGS tag, HL1 coded with code set 18 + difference flag, peak tags and,
    for coefficients beyond +-250, the peak table) = reference sample."""
    if not have_ref(): pytest.skip("reference .so not built")
    if kind == "qbist": frames, pitch = qbist_frames(10, 1, w, h); frame = frames[0]
    elif kind == "peaks": frame, pitch = field_flicker_frame(w, h)
    elif kind == "allpeaks": frame, pitch = every_column_flicker_frame(w, h)
    else: frame, pitch = synth_yuy2(w, h, 3)
    rs = ref_encode_frames([frame], pitch, w, h, PIX_YUY2, encoded=ENCODED_YUV422, flags=1)[0]
    plan = Plan(w, h, progressive=0)
    coeffs = oracle_forward_interlaced_yuv422(plan, frame, pitch)
    if kind == "peaks": assert np.abs(plan.view(coeffs, 0, 0, 2)).max() > 250
    if kind == "allpeaks":
        # more peaks than the table's chunk header can count (2 x MAX_CHUNK_SIZE = 131070 values, encoder.c:6557): the reference writes no table for that band and
        # leaves the three tags in front of it zero (advisor finding of round 5: the product wrote a table behind a truncated chunk length)
        assert (np.abs(plan.view(coeffs, 0, 0, 2)) > 250).sum() > 131070
    off, n = first_metadata_chunk(rs)
    mine = product_write_sample_host(plan, coeffs, 1, meta_global=rs[off:off + n], progressive=0)
    assert len(mine) == len(rs)
    assert mine == rs


def _thumbnail(L, sample):
    dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
    sb = ctypes.create_string_buffer(sample, len(sample))
    out = np.zeros(512 * 512 * 4, np.uint8)
    w = ctypes.c_size_t(); h = ctypes.c_size_t(); n = ctypes.c_size_t()
    L.CFHD_GetThumbnail.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                    ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    rc = L.CFHD_GetThumbnail(dec, sb, len(sample), out.ctypes.data_as(ctypes.c_void_p), out.size, 0, ctypes.byref(w), ctypes.byref(h), ctypes.byref(n))
    L.CFHD_CloseDecoder(dec)
    return rc, w.value, h.value, bytes(out[: n.value])


@pytest.mark.parametrize("w,h,fmt,enc", [(320, 240, PIX_YUY2, ENCODED_YUV422), (1920, 1080, PIX_YUY2, ENCODED_YUV422), (336, 252, PIX_2VUY, ENCODED_YUV422),
                                         (320, 240, PIX_RG48, ENCODED_RGB444), (320, 240, PIX_B64A, ENCODED_RGBA4444)])
def test_thumbnail_equals_reference(w, h, fmt, enc):
    """CFHD_GetThumbnail (host code, no GPU): the 1/8 x 1/8 10-bit RGB picture the reference cuts out of the raw lowpass bands
    (Codec/thumbnail.c:65), byte for byte."""
    if not have_ref(): pytest.skip("reference .so not built")
    if fmt in (PIX_YUY2, PIX_2VUY): frames, pitch = [synth_yuy2(w, h, 5)[0]], w * 2
    else: frames, pitch = qbist_frames(10, 1, w, h, fmt, alpha=1) if fmt == PIX_B64A else qbist_frames(10, 1, w, h, fmt)
    sample = ref_encode_frames(frames, pitch, w, h, fmt, encoded=enc)[0]
    want = _thumbnail(ref(), sample)
    got = _thumbnail(product(), sample)
    assert want[0] == 0 and got[0] == 0
    assert got[1:3] == want[1:3] == (w // 8, (h + 7) // 8)
    assert got[3] == want[3]
    assert len(set(got[3])) > 16                      # a picture, not a constant


def test_thumbnail_and_output_formats_argument_handling():
    """No GPU involved: argument checks of CFHD_GetThumbnail and the sample-aware CFHD_GetOutputFormats."""
    L = product()
    w, h = 320, 240
    frame, pitch = synth_yuy2(w, h, 5)
    plan = Plan(w, h)
    sample = product_write_sample_host(plan, oracle_forward_yuv422(plan, frame, pitch), 1)
    dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
    sb = ctypes.create_string_buffer(sample, len(sample))
    small = np.zeros(16, np.uint8)
    L.CFHD_GetThumbnail.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                    ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    assert L.CFHD_GetThumbnail(dec, sb, len(sample), small.ctypes.data_as(ctypes.c_void_p), small.size, 0, None, None, None) == 1   # buffer too small
    assert L.CFHD_GetThumbnail(dec, None, 0, small.ctypes.data_as(ctypes.c_void_p), small.size, 0, None, None, None) == 1
    big = np.zeros(40 * 30 * 4, np.uint8)
    assert L.CFHD_GetThumbnail(dec, sb, 200, big.ctypes.data_as(ctypes.c_void_p), big.size, 0, None, None, None) == 5               # truncated sample
    assert L.CFHD_GetThumbnail(dec, sb, len(sample), big.ctypes.data_as(ctypes.c_void_p), big.size, 0, None, None, None) == 0
    fmts = (ctypes.c_uint32 * 8)(); n = ctypes.c_int()
    L.CFHD_GetOutputFormats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    assert L.CFHD_GetOutputFormats(dec, sb, len(sample), fmts, 8, ctypes.byref(n)) == 0
    assert [fmts[i] for i in range(n.value)] == [PIX_YUY2, PIX_2VUY, fourcc("YU64"), fourcc("v210"), PIX_RG24]
    assert L.CFHD_GetOutputFormats(dec, None, 0, fmts, 8, ctypes.byref(n)) == 0 and n.value == 8
    many = (ctypes.c_uint32 * 32)()
    assert L.CFHD_GetOutputFormats(dec, None, 0, many, 32, ctypes.byref(n)) == 0 and n.value == 15 and len(set(many[: n.value])) == 15      # every format once, RG30 and BYR4 among them
    L.CFHD_CloseDecoder(dec)



def test_deep_rgb_as_yuv422_and_rgb10_outputs_are_accepted():
    """RG48 / b64a -> YUV 4:2:2 and the 10-bit RGB decoder outputs are open (they sat behind CFHD_AMD_UNVERIFIED=1 until their first hardware run in
    round 3): CFHD_PrepareToEncode goes on to the GPU (any answer but BADFORMAT here, where there is none); CFHD_PrepareToDecode is host code."""
    L = product()
    enc = ctypes.c_void_p(); assert L.CFHD_OpenEncoder(ctypes.byref(enc), None) == 0
    for fmt in (PIX_RG48, PIX_B64A, PIX_RG24, PIX_BGRA, PIX_BGRa):           # (8-bit RGB -> 4:2:2: added later in round 3)
        assert L.CFHD_PrepareToEncode(enc, 320, 240, fmt, ENCODED_YUV422, 0, QUALITY_FILMSCAN1) != 3
    assert L.CFHD_PrepareToEncode(enc, 328, 240, PIX_BGRA, ENCODED_YUV422, 0, QUALITY_FILMSCAN1) == 3          # chroma width must divide by 8
    for fmt in (PIX_BGRA, PIX_BGRa): assert L.CFHD_PrepareToEncode(enc, 320, 240, fmt, ENCODED_RGBA4444, 0, QUALITY_FILMSCAN1) != 3
    assert L.CFHD_PrepareToEncode(enc, 320, 240, PIX_RG24, ENCODED_RGBA4444, 0, QUALITY_FILMSCAN1) == 3           # no alpha to encode
    if have_ref():
        frgb, prgb = qbist_frames(10, 1, 320, 240, PIX_RG48)
        sample = ref_encode_frames(frgb, prgb, 320, 240, PIX_RG48, encoded=ENCODED_RGB444)[0]
        dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
        aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
        sb = ctypes.create_string_buffer(sample, len(sample))
        for name in sorted(RGB10_FORMATS):
            assert L.CFHD_PrepareToDecode(dec, 0, 0, fourcc(name), 1, 0, sb, 512, ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0
            assert L.CFHD_PrepareToDecode(dec, 0, 0, fourcc(name), 2, 0, sb, 512, ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 0 and (aw.value, ah.value) == (160, 120)      # half resolution: k_half_rgb
        L.CFHD_CloseDecoder(dec)
    L.CFHD_CloseEncoder(enc)


def test_decoder_output_format_gates():
    """CFHD_PrepareToDecode (host code: no GPU involved) accepts exactly the (encoded format, output format, resolution) combinations the
    library decodes and answers CFHD_ERROR_BADFORMAT (3) for the rest: 4:2:2 -> YUY2 / 2vuy (full, half), YU64 (full), BGRA / BGRa / RG48 / b64a (full, half); RGB 4:4:4 -> RG48
    (full, half), RG24 / BGRA / BGRa / 10-bit RGB / b64a (full, half); RGBA 4:4:4:4 -> b64a and RG48 (full, half), BGRA / BGRa (full)."""
    if not have_ref(): pytest.skip("reference .so not built")
    L = product()
    w, h = 320, 240
    f422, p422 = synth_yuy2(w, h, 3)
    frgb, prgb = qbist_frames(10, 1, w, h, PIX_RG48)
    fa, pa = qbist_frames(10, 1, w, h, PIX_B64A, alpha=1)
    samples = {"422": ref_encode_frames([f422], p422, w, h, PIX_YUY2)[0],
               "444": ref_encode_frames(frgb, prgb, w, h, PIX_RG48, encoded=ENCODED_RGB444)[0],
               "4444": ref_encode_frames(fa, pa, w, h, PIX_B64A, encoded=ENCODED_RGBA4444)[0]}
    accepted = {("422", "YUY2", 1), ("422", "YUY2", 2), ("422", "2vuy", 1), ("422", "2vuy", 2), ("422", "YU64", 1), ("422", "YU64", 2), ("422", "RG24", 2),
                ("444", "RG48", 1), ("444", "RG48", 2), ("444", "RG24", 1), ("444", "BGRA", 1), ("444", "BGRa", 1), ("444", "r210", 1),
                ("4444", "b64a", 1), ("4444", "b64a", 2), ("4444", "BGRA", 1), ("4444", "BGRa", 1), ("444", "b64a", 1), ("422", "RG24", 1), ("4444", "RG48", 1), ("4444", "RG48", 2),
                ("444", "RG24", 2), ("444", "BGRA", 2), ("444", "BGRa", 2), ("444", "r210", 2), ("444", "b64a", 2), ("4444", "BGRA", 2), ("4444", "BGRa", 2),
                ("422", "BGRA", 1), ("422", "BGRa", 1), ("422", "RG48", 1), ("422", "b64a", 1),       # (round 4: the last four rows of TestCFHD's table)
                ("422", "BGRA", 2), ("422", "BGRa", 2), ("422", "RG48", 2), ("422", "b64a", 2)}
    dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
    aw = ctypes.c_int(); ah = ctypes.c_int(); af = ctypes.c_uint32()
    for enc, sample in samples.items():
        sb = ctypes.create_string_buffer(sample, len(sample))
        for name in ("YUY2", "2vuy", "YU64", "v210", "RG48", "RG24", "BGRA", "BGRa", "b64a", "r210", "BYR4"):
            for res in (1, 2):
                rc = L.CFHD_PrepareToDecode(dec, 0, 0, fourcc(name), res, 0, sb, 512, ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af))
                if (enc, name, res) in accepted:
                    assert rc == 0, (enc, name, res, rc)
                    assert (aw.value, ah.value, af.value) == (w // res, h // res, fourcc(name))
                else:
                    assert rc == 3, (enc, name, res, rc)
        assert L.CFHD_PrepareToDecode(dec, 0, 0, PIX_YUY2 if enc == "422" else (PIX_RG48 if enc == "444" else PIX_B64A), 3, 0, sb, 512, ctypes.byref(aw), ctypes.byref(ah), ctypes.byref(af)) == 11   # quarter: CFHD_ERROR_BAD_RESOLUTION
    L.CFHD_CloseDecoder(dec)


@pytest.mark.parametrize("w,h,fmt,enc,flags", [(320, 240, PIX_YUY2, ENCODED_YUV422, 0), (336, 252, PIX_YUY2, ENCODED_YUV422, 1), (320, 240, PIX_RG48, ENCODED_RGB444, 0),
                                               (320, 240, PIX_B64A, ENCODED_RGBA4444, 0)])
def test_obsolete_header_parser_and_encoder_side_thumbnail_equal_reference(w, h, fmt, enc, flags):
    """CFHD_ParseSampleHeader (four ints: encoded format, field type, width, height) and CFHD_GetEncodeThumbnail, host code, against the
    reference's on the same sample."""
    if not have_ref(): pytest.skip("reference .so not built")
    if fmt == PIX_YUY2: frames, pitch = [synth_yuy2(w, h, 5)[0]], w * 2
    else: frames, pitch = qbist_frames(10, 1, w, h, fmt, alpha=1) if fmt == PIX_B64A else qbist_frames(10, 1, w, h, fmt)
    sample = ref_encode_frames(frames, pitch, w, h, fmt, encoded=enc, flags=flags)[0]
    sb = ctypes.create_string_buffer(sample, len(sample))
    got = []
    for L in (ref(), product()):
        hdr = (ctypes.c_int * 4)(-1, -1, -1, -1)
        assert L.CFHD_ParseSampleHeader(sb, ctypes.c_size_t(len(sample)), hdr) == 0
        enc_ref = ctypes.c_void_p(); assert L.CFHD_OpenEncoder(ctypes.byref(enc_ref), None) == 0
        out = np.zeros(64 * 64 * 4, np.uint8)
        tw = ctypes.c_size_t(); th = ctypes.c_size_t(); tn = ctypes.c_size_t()
        L.CFHD_GetEncodeThumbnail.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                              ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
        assert L.CFHD_GetEncodeThumbnail(enc_ref, sb, len(sample), out.ctypes.data_as(ctypes.c_void_p), out.size, 0, ctypes.byref(tw), ctypes.byref(th), ctypes.byref(tn)) == 0
        L.CFHD_CloseEncoder(enc_ref)
        got.append((list(hdr), tw.value, th.value, bytes(out[: tn.value])))
    assert got[0] == got[1]
    assert got[0][0][2:] == [w, h]


@pytest.mark.parametrize("w,h,fmt,enc", [(320, 240, PIX_YUY2, ENCODED_YUV422), (320, 240, PIX_RG48, ENCODED_RGB444), (320, 240, PIX_B64A, ENCODED_RGBA4444),
                                         (320, 240, PIX_BYR4, 3)])
def test_sample_info_equals_reference(w, h, fmt, enc):
    """CFHD_GetSampleInfo, every tag, against the reference on the same sample -- CFHD_SAMPLE_ENCODED_FORMAT in particular is the public
    CFHD_EncodedFormat value (SampleDecoder.cpp:821-840), not the bitstream's code."""
    if not have_ref(): pytest.skip("reference .so not built")
    if fmt == PIX_YUY2: frames, pitch = [synth_yuy2(w, h, 5)[0]], w * 2
    elif fmt == PIX_BYR4:
        mosaic = synth_bayer(w, h, 3); frames, pitch = [mosaic.reshape(-1).view(np.uint8).copy()], w * 2
    else: frames, pitch = qbist_frames(10, 1, w, h, fmt, alpha=1) if fmt == PIX_B64A else qbist_frames(10, 1, w, h, fmt)
    sample = ref_encode_frames(frames, pitch, w, h, fmt, encoded=enc)[0]
    sb = ctypes.create_string_buffer(sample, len(sample))
    got = []
    for L in (ref(), product()):
        dec = ctypes.c_void_p(); assert L.CFHD_OpenDecoder(ctypes.byref(dec), None) == 0
        vals = []
        for tag in (1, 2, 3, 4, 5):                  # display width / height, key frame, progressive, encoded format
            v = ctypes.c_int32(-7)
            rc = L.CFHD_GetSampleInfo(dec, sb, ctypes.c_size_t(len(sample)), tag, ctypes.byref(v), ctypes.c_size_t(4))
            vals.append((rc, v.value))
        L.CFHD_CloseDecoder(dec)
        got.append(vals)
    assert got[0] == got[1]
    assert got[1][4] == (0, enc)


@pytest.mark.parametrize("quality", [5, 6, 2])
def test_rate_feedback_quantizers_follow_the_reference(quality):
    """Multi-frame encodes at FILMSCAN2 / FILMSCAN3 (and MEDIUM, whose bit-rate limiter looks at the previous sample) re-derive the
    quantizer tables every frame from the size of the previous sample (encoder.c:9442, :9911; quantize.c:186-300, :2994).  The product's
    derivation, fed with the reference's own sample sizes, must produce the divisors the reference wrote into its band headers."""
    if not have_ref(): pytest.skip("reference .so not built")
    w, h, n = 640, 360, 6
    frames = feedback_test_frames(w, h, n)
    samples = ref_encode_frames(frames, w * 2, w, h, quality=quality)
    sizes = (ctypes.c_longlong * n)(*[len(s) for s in samples])
    mine = (ctypes.c_int * (n * 27))()
    assert hooks().cfhd_amd_quant_sequence(w, h, 1, 1, quality, 1, sizes, n, mine) == n * 27
    moved = False
    for f, s in enumerate(samples):
        theirs = (ctypes.c_int * 64)()
        a = np.frombuffer(s, np.uint8).copy()
        assert hooks().cfhd_amd_sample_quants(p8(a), len(s), theirs) == 27
        assert list(theirs[:27]) == list(mine[27 * f: 27 * f + 27]), "frame %d" % f
        moved |= f > 0 and list(mine[27 * f: 27 * f + 27]) != list(mine[:27])
    if quality >= 5:
        assert moved, "the test frames never moved the limiter: nothing was tested"
