// cfhd_device.hip -- HIP runtime glue: device selection, HBM/pinned buffers, job tables and kernel launches.
//
// Replaces the reference's per-encoder scratch + thread pool (EncoderSDK/AsyncEncoder.cpp, Codec/thread.c) with
// HIP streams: a batch owns one stream; H2D copy, the three forward (or inverse) launches and the D2H copy are
// queued back to back on it, batches on different streams overlap.
#include "cfhd_device.h"
#include "cfhd_kernels.h"
#include <hip/hip_runtime.h>
#include <string.h>
#include <stddef.h>
#include <vector>
#include <stdlib.h>
#include <stdio.h>
#include <mutex>
#include <thread>
#include <atomic>
#include <string>
#include <utility>

namespace cfhd {

namespace {
// Text of the last failure: written from whatever thread failed (the chunks of a batch run on threads of their own), read through cfhd_amd_last_error()
std::string g_err_text; std::mutex g_err_mutex;
thread_local std::string t_err_copy;
struct ErrSlot { ErrSlot &operator=(const char *t) { std::lock_guard<std::mutex> l(g_err_mutex); g_err_text = t; return *this; } } g_err;
std::once_flag g_init_once;
int g_init_rc = -1;
int g_device = 0;

int fail(hipError_t e, const char *what)
{
	char buf[256];
	snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
	g_err = buf;
	return (int)e ? (int)e : -1;
}
#define HIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(_e, #expr); } while (0)

dev::QuantParam make_q(int divisor, int mpq)
{
	dev::QuantParam q; q.divisor = divisor; q.mid = 0; q.mult = 0;
	if (divisor > 1) {
		if (mpq >= 2 && mpq < 9) { q.mid = divisor / mpq; if (mpq == 2 && q.mid) q.mid--; }   // quantize.c:1415-1427
		q.mult = ((1u << 16) / (unsigned)divisor) & 0xffffu;
	}
	return q;
}

struct EncJobs {            // layout of the job table buffer of an EncodeBatch
	dev::FwdYuvJob *yuv;    // [n]         level 1 of the packed 4:2:2 formats
	dev::FwdPlaneJob *l2;   // [n * nch]
	dev::FwdPlaneJob *l3;   // [n * nch]
	dev::FwdPlaneJob *l1;   // [n * nch]   level 1 of the interleaved 16-bit 4:4:4(:4) formats (k_fwd_packed16) and of the Bayer planes (k_fwd_plane)
	dev::BayerJob *bayer;   // [n]         Bayer mosaic -> component planes (k_unpack_byr4)
};
EncJobs enc_jobs_at(void *base, int n, int nch)
{
	EncJobs j;
	j.yuv = (dev::FwdYuvJob *)base;
	j.l2 = (dev::FwdPlaneJob *)(j.yuv + n);
	j.l3 = j.l2 + (size_t)n * nch;
	j.l1 = j.l3 + (size_t)n * nch;
	j.bayer = (dev::BayerJob *)(j.l1 + (size_t)n * nch);
	return j;
}
size_t enc_jobs_bytes(int n, int nch) { return (size_t)n * sizeof(dev::FwdYuvJob) + 3 * (size_t)n * nch * sizeof(dev::FwdPlaneJob) + (size_t)n * sizeof(dev::BayerJob); }

struct DecJobs { dev::InvPlaneJob *l3, *l2; dev::InvYuvJob *yuv; dev::InvPlaneJob *l1; /* last level of the 4:4:4(:4) formats (k_inv_packed16) */ dev::HalfYuvJob *half; /* [n] half-resolution output */ dev::HalfPackedJob *halfp; /* [n] the same for the 4:4:4(:4) formats */ };
DecJobs dec_jobs_at(void *base, int n, int nch)
{
	DecJobs j;
	j.l3 = (dev::InvPlaneJob *)base;
	j.l2 = j.l3 + (size_t)n * nch;
	j.yuv = (dev::InvYuvJob *)(j.l2 + (size_t)n * nch);
	j.l1 = (dev::InvPlaneJob *)(j.yuv + n);
	j.half = (dev::HalfYuvJob *)(j.l1 + (size_t)n * nch);
	j.halfp = (dev::HalfPackedJob *)(j.half + n);
	return j;
}
size_t dec_jobs_bytes(int n, int nch) { return 3 * (size_t)n * nch * sizeof(dev::InvPlaneJob) + (size_t)n * sizeof(dev::InvYuvJob) + (size_t)n * sizeof(dev::HalfYuvJob) + (size_t)n * sizeof(dev::HalfPackedJob); }

// Word of component plane c inside an interleaved 16-bit pixel: planes are G, R, B(, A) (frame.c:6128-6157, convert.c:6750-6752),
// RG48 pixels are R, G, B; b64a pixels are A, R, G, B (frame.c:6676-6683).
int packed_word_of_channel(int pixel_kind, int c)
{
	static const int rg48[4] = { 1, 0, 2, 3 }, b64a[4] = { 2, 1, 3, 0 };
	return (pixel_kind == PIX_B64A ? b64a : rg48)[c & 3];
}
bool is_packed16(int pixel_kind) { return pixel_kind == PIX_RG48 || pixel_kind == PIX_B64A; }
// decoder outputs made of 16-bit words that k_inv_packed16 writes plane by plane: the interleaved RGB(A) pixels, and YU64 (words Y0 C1 Y1 C2:
// luma every second word, the half-width chroma planes every fourth; the reference decodes 4:2:2 samples to YU64 through the same planar
// 16-bit rows as RGB 4:4:4 to RG48, oracle/cfhd_oracle_inv.c orc_inv_spatial_to_yu64)
// ... and the 8-bit RGB pixels B, G, R(, A) of RGB 4:4:4 samples (RG24, BGRA: bottom row first; BGRa: top row first), which the same kernel writes
// in its byte mode: the 12-bit component doubled + 9 + a four-bit dither, >> 5 (orc_inv_spatial_to_rgb8: the reference's model)
static bool dec_rgb8(int out_kind) { return out_kind == PIX_RG24 || out_kind == PIX_BGRA || out_kind == PIX_BGRa; }
static int rgb8_bytes(int out_kind) { return out_kind == PIX_RG24 ? 3 : 4; }
// ... and the 10-bit RGB words (r210, DPX0: big-endian; AB10, AR10: little-endian) through k_inv_rgb10 (orc_inv_spatial_to_rgb10)
static bool dec_rgb10(int out_kind) { return out_kind >= PIX_R210 && out_kind <= PIX_AR10; }
// planes of the sample that reach the output pixel: an RGBA 4:4:4:4 sample decoded to RG48 leaves its alpha plane behind (the reference's RG48 route on planes G, R, B;
// pinned on eight geometries, tests/test_oracle_vs_ref.py)
static int dec_out_channels(int out_kind, const FramePlan &plan) { return out_kind == PIX_RG48 && plan.encoded_format == ENC_RGBA4444 ? 3 : plan.num_channels; }
static bool dec_planes16(int out_kind) { return is_packed16(out_kind) || out_kind == PIX_YU64 || dec_rgb8(out_kind) || dec_rgb10(out_kind); }
// position of plane c inside the pixel: 16-bit word, or byte for the 8-bit formats (planes G, R, B(, A) -> bytes 1, 2, 0(, 3))
static int dec_word_of_channel(int out_kind, int c) { return dec_rgb10(out_kind) ? 0 : out_kind == PIX_YU64 ? (c == 0 ? 0 : (c == 1 ? 1 : 3)) : (dec_rgb8(out_kind) ? (c == 0 ? 1 : (c == 1 ? 2 : (c == 2 ? 0 : 3))) : packed_word_of_channel(out_kind, c)); }
static int dec_stride_of_channel(int out_kind, int c, int nch) { return out_kind == PIX_YU64 ? (c == 0 ? 2 : 4) : (dec_rgb8(out_kind) ? rgb8_bytes(out_kind) : (dec_rgb10(out_kind) ? 3 : (out_kind == PIX_B64A ? 4 : nch))); }     // (b64a from RGB 4:4:4: three planes, four words)
static int dec_words_per_position(int out_kind, int nch) { return out_kind == PIX_YU64 ? 2 : (dec_rgb8(out_kind) ? rgb8_bytes(out_kind) : (out_kind == PIX_B64A ? 4 : nch)); }
static int16_t *dec_plane_out(void *frame, int out_kind, int c) { return dec_rgb8(out_kind) ? (int16_t *)((uint8_t *)frame + dec_word_of_channel(out_kind, c)) : (int16_t *)((uint16_t *)frame + dec_word_of_channel(out_kind, c)); }
// encoder input made of 16-bit words that k_fwd_packed16 picks apart (per channel: first word, words from sample to sample, right shift).
// YU64 (Codec/frame.c:1556 ConvertYU64ToFrame16s + convert.c:3345, :14375): words Y0 C1 Y1 C2, every word >> 6 to 10 bits, channel 1 = C1, 2 = C2.
// v210 (frame.c:1431 ConvertV210ToFrame16s): three 10-bit samples per 32-bit word, FwdPlaneJob::layout tells the loader which component to pick.
// RG24 (frame.c:6173 ConvertRGBtoRGB48): bytes B, G, R, bottom row first, byte << 4; planes G, R, B.
static bool enc_packed16(int pixel_kind) { return is_packed16(pixel_kind) || pixel_kind == PIX_RG64 || pixel_kind == PIX_YU64 || pixel_kind == PIX_V210 || pixel_kind == PIX_RG24 || pixel_kind == PIX_BGRA || pixel_kind == PIX_BGRa || (pixel_kind >= PIX_R210 && pixel_kind <= PIX_AR10); }
static bool enc_bytes8(int pixel_kind) { return pixel_kind == PIX_RG24 || pixel_kind == PIX_BGRA || pixel_kind == PIX_BGRa; }
static bool enc_rgb10(int pixel_kind) { return pixel_kind >= PIX_R210 && pixel_kind <= PIX_AR10; }
// bit position of plane c (G, R, B) inside the pixel word of the 10-bit RGB formats
static int rgb10_shift(int pixel_kind, int c) { const int r = pixel_kind == PIX_DPX0 ? 22 : (pixel_kind == PIX_AB10 ? 0 : 20), g = pixel_kind == PIX_DPX0 ? 12 : 10, b = pixel_kind == PIX_DPX0 ? 2 : (pixel_kind == PIX_AB10 ? 20 : 0); return c == 0 ? g : (c == 1 ? r : b); }
// RG48 / b64a encoded as YUV 4:2:2: the loader of k_fwd_packed16 converts the pixels (FwdPlaneJob::layout 7); every plane reads from the R word
static bool enc_rgb_as_422(const FramePlan &plan) { return (is_packed16(plan.pixel_kind) || plan.pixel_kind == PIX_RG64) && plan.encoded_format == ENC_YUV422; }
static int enc_word_of_channel(int pixel_kind, int c) { return pixel_kind == PIX_V210 || enc_bytes8(pixel_kind) || enc_rgb10(pixel_kind) ? 0 : (pixel_kind == PIX_YU64 ? (c == 0 ? 0 : (c == 1 ? 1 : 3)) : packed_word_of_channel(pixel_kind, c)); }
static int enc_stride_of_channel(int pixel_kind, int c, int nch) { return pixel_kind == PIX_YU64 ? (c == 0 ? 2 : 4) : (pixel_kind == PIX_B64A || pixel_kind == PIX_RG64 ? 4 : nch); }     // (b64a / RG64 to RGB 4:4:4 have three planes of four-word pixels)
} // namespace

const char *device_last_error() { std::lock_guard<std::mutex> l(g_err_mutex); t_err_copy = g_err_text; return t_err_copy.c_str(); }

namespace {
std::mutex g_pins_mutex;
std::vector<std::pair<uintptr_t, size_t>> g_pins;
}
int host_buffer_register(void *p, size_t bytes)
{
	if (!p || !bytes) return -1;
	int rc = device_init();
	if (rc) return rc;
	std::lock_guard<std::mutex> lk(g_pins_mutex);
	for (auto &e : g_pins) if (e.first == (uintptr_t)p) return e.second >= bytes ? 0 : -1;
	HIPCHK(hipHostRegister(p, bytes, hipHostRegisterPortable));
	g_pins.push_back({ (uintptr_t)p, bytes });
	return 0;
}
int host_buffer_unregister(void *p)
{
	std::lock_guard<std::mutex> lk(g_pins_mutex);
	for (size_t i = 0; i < g_pins.size(); i++) if (g_pins[i].first == (uintptr_t)p) {
		g_pins.erase(g_pins.begin() + (ptrdiff_t)i);
		HIPCHK(hipHostUnregister(p));
		return 0;
	}
	return -1;
}
bool host_buffer_is_registered(const void *p, size_t bytes)
{
	std::lock_guard<std::mutex> lk(g_pins_mutex);
	for (auto &e : g_pins) if ((uintptr_t)p >= e.first && (uintptr_t)p + bytes <= e.first + e.second) return true;
	return false;
}

// ---- streams (cfhd_device.h)
static thread_local bool t_scope = false;
static thread_local void *t_scope_stream = nullptr;
static std::mutex g_shared_mutex;
static std::vector<void *> g_shared_streams;
int device_stream_create(void **stream)
{
	if (t_scope && t_scope_stream) { *stream = t_scope_stream; return 0; }
	hipStream_t s;
	const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	if (e != hipSuccess) return (int)e;
	*stream = (void *)s;
	if (t_scope) { t_scope_stream = (void *)s; std::lock_guard<std::mutex> lk(g_shared_mutex); g_shared_streams.push_back((void *)s); }
	return 0;
}
void device_stream_destroy(void *stream)
{
	if (!stream) return;
	{ std::lock_guard<std::mutex> lk(g_shared_mutex); for (void *p : g_shared_streams) if (p == stream) return; }      // a scope's stream: its owner releases it
	(void)hipStreamDestroy((hipStream_t)stream);
}
void device_stream_release(void *stream)
{
	if (!stream) return;
	{ std::lock_guard<std::mutex> lk(g_shared_mutex); for (size_t i = 0; i < g_shared_streams.size(); i++) if (g_shared_streams[i] == stream) { g_shared_streams.erase(g_shared_streams.begin() + (long)i); break; } }
	(void)hipStreamDestroy((hipStream_t)stream);
}
static thread_local bool t_lean = false;
void device_streams_lean(bool on) { t_lean = on; }
bool device_streams_are_lean() { return t_lean; }
StreamScope::StreamScope() { t_scope = true; t_scope_stream = nullptr; }
StreamScope::StreamScope(void *preset) { t_scope = true; t_scope_stream = preset; }
StreamScope::~StreamScope() { t_scope = false; t_scope_stream = nullptr; }
void *StreamScope::stream() const { return t_scope_stream; }

int device_count()
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// The device the calling thread prepares its objects on: the process default (CFHD_AMD_DEVICE / LOCAL_RANK / 0) unless the thread asked for
// another one -- a worker of an encoder pool that spreads over the node's GPUs, a decoder handle that was dealt one (cfhd_api.cpp).
static thread_local int t_device = -1;
int device_select(int dev)
{
	t_device = dev;
	const int rc = device_init();
	return rc ? -1 : device_current();
}
int device_current() { return t_device >= 0 ? t_device : g_device; }

namespace {
struct StageOrder { std::mutex m; hipEvent_t ev[2] = { nullptr, nullptr }; bool recorded[2] = { false, false }; };
StageOrder *stage_order_of(int device)
{
	static std::mutex m; static std::vector<StageOrder *> all;       // (never freed: events of a device live as long as the process)
	std::lock_guard<std::mutex> lk(m);
	if (device < 0) return nullptr;
	if ((size_t)device >= all.size()) all.resize((size_t)device + 1, nullptr);
	if (!all[device]) all[device] = new StageOrder;
	return all[device];
}
}
int stage_order_wait(int device, int stage, void *stream)
{
	StageOrder *o = stage_order_of(device);
	if (!o || stage < 0 || stage > 1) return -1;
	std::lock_guard<std::mutex> lk(o->m);
	if (o->recorded[stage]) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, o->ev[stage], 0));
	return 0;
}
int stage_order_done(int device, int stage, void *stream)
{
	StageOrder *o = stage_order_of(device);
	if (!o || stage < 0 || stage > 1) return -1;
	std::lock_guard<std::mutex> lk(o->m);
	(void)hipSetDevice(device);
	if (!o->ev[stage]) HIPCHK(hipEventCreateWithFlags(&o->ev[stage], hipEventDisableTiming));
	HIPCHK(hipEventRecord(o->ev[stage], (hipStream_t)stream));
	o->recorded[stage] = true;
	return 0;
}
int device_caller_save() { int d = -1; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return -1; } return d; }
void device_caller_restore(int dev) { if (dev >= 0) { int now = -1; if (hipGetDevice(&now) == hipSuccess && now != dev) (void)hipSetDevice(dev); } }

int device_init()
{
	std::call_once(g_init_once, [] {
		int n = 0;
		hipError_t e = hipGetDeviceCount(&n);
		if (e != hipSuccess || n <= 0) { g_err = "no HIP device available (libcfhd_amd has no CPU fallback)"; g_init_rc = e ? (int)e : -1; return; }
		const char *env = getenv("CFHD_AMD_DEVICE");
		if (!env) env = getenv("LOCAL_RANK");
		g_device = env ? atoi(env) % n : 0;
		e = hipSetDevice(g_device);
		if (e != hipSuccess) { fail(e, "hipSetDevice"); g_init_rc = (int)e; return; }
		g_init_rc = 0;
	});
	if (g_init_rc == 0) {
		int n = 1; (void)hipGetDeviceCount(&n);
		if (t_device >= n) t_device = t_device % (n > 0 ? n : 1);
		hipSetDevice(device_current());              // per calling thread
	}
	return g_init_rc;
}

int packed_frame_pitch(int pixel_kind, int width)
{
	switch (pixel_kind) {
	case PIX_YUY2: case PIX_2VUY: return width * 2;
	case PIX_RG48: return width * 6;
	case PIX_B64A: case PIX_RG64: return width * 8;
	case PIX_BYR4: return width * 2;
	case PIX_BYR5: return width * 3;       // per row PAIR of the mosaic: 4 x width / 2 samples of 12 bits (the unit the frame is laid out in)
	case PIX_YU64: return width * 4;
	case PIX_RG24: return width * 3;
	case PIX_BGRA: case PIX_BGRa: case PIX_R210: case PIX_DPX0: case PIX_AB10: case PIX_AR10: return width * 4;
	case PIX_V210: return (width + 47) / 48 * 128;      // six pixels in 16 bytes, rows padded to 48 pixels (Example/utils.cpp:84-90)
	default: return 0;
	}
}

// =============================================================================================
// EncodeBatch
// =============================================================================================
EncodeBatch::EncodeBatch() {}
EncodeBatch::~EncodeBatch() { release(); }

void EncodeBatch::release()
{
	(void)hipSetDevice(device_);
	if (stream_) hipStreamSynchronize((hipStream_t)stream_);
	ent_ready_ = false;
	if (d_in_) hipFree(d_in_);
	if (h_in_) hipHostFree(h_in_);
	if (d_coeff_) hipFree(d_coeff_);
	if (h_coeff_) hipHostFree(h_coeff_);
	if (d_jobs_) hipFree(d_jobs_);
	if (h_jobs_) hipHostFree(h_jobs_);
	if (d_planes_) hipFree(d_planes_);
	if (d_curve_) hipFree(d_curve_);
	d_planes_ = nullptr; d_curve_ = nullptr;
	if (ev0_) hipEventDestroy((hipEvent_t)ev0_);
	if (ev1_) hipEventDestroy((hipEvent_t)ev1_);
	for (int k = 0; k < 2; k++) if (evl_[k]) { hipEventDestroy((hipEvent_t)evl_[k]); evl_[k] = nullptr; }
	if (stream_) device_stream_destroy(stream_);
	d_in_ = h_in_ = nullptr; d_coeff_ = h_coeff_ = nullptr; d_jobs_ = h_jobs_ = nullptr; stream_ = ev0_ = ev1_ = nullptr; n_ = 0;
}

int EncodeBatch::prepare(const FramePlan &plan, int nframes, bool own_input)
{
	int rc = device_init();
	if (rc) return rc;
	release();
	device_ = device_current(); (void)hipSetDevice(device_);      // (release() went to the device of the buffers it freed)
	const bool bayer = plan.pixel_kind == PIX_BYR4 || plan.pixel_kind == PIX_BYR5;
	if (plan.pixel_kind != PIX_YUY2 && plan.pixel_kind != PIX_2VUY && !enc_packed16(plan.pixel_kind) && !bayer) { g_err = "pixel format not supported by the GPU path yet"; return -2; }
	plan_ = plan; n_ = nframes; own_input_ = own_input;
	bayer_fused_ = false;      // (level 1 straight from the mosaic through the tiled kernel: measured 5.1 ms against 2.2 for 96 4K frames in round 3; k_fwd_bayer_strip is the fused kernel that pays)
	HIPCHK((hipError_t)device_stream_create(&stream_));
	HIPCHK(hipEventCreate((hipEvent_t *)&ev0_));
	HIPCHK(hipEventCreate((hipEvent_t *)&ev1_));
	for (int k = 0; k < 2; k++) HIPCHK(hipEventCreate((hipEvent_t *)&evl_[k]));
	// Bayer: the plan describes the component planes (half the mosaic in both directions)
	in_pitch_ = packed_frame_pitch(plan.pixel_kind, bayer ? 2 * plan.width : plan.width);
	in_rows_ = plan.pixel_kind == PIX_BYR4 ? 2 * plan.display_height : plan.display_height;      // (BYR5: one packed row per row pair)
	frame_bytes_ = (size_t)in_pitch_ * in_rows_;
	if (bayer) {
		plane_elems_ = (size_t)plan.ch[0].band[0][0].pitch * 2 * plan.height;           // plane pitch = 2 x the level-1 band pitch (multiple of 16)
		HIPCHK(hipMalloc((void **)&d_planes_, plane_elems_ * 2 * 4 * n_));
		std::vector<uint16_t> curve((size_t)1 << kBayerCurveBits);
		build_bayer_log90_curve(plan.precision, curve.data());
		HIPCHK(hipMalloc((void **)&d_curve_, curve.size() * 2));
		HIPCHK(hipMemcpy(d_curve_, curve.data(), curve.size() * 2, hipMemcpyHostToDevice));
	}
	if (own_input) {
		HIPCHK(hipMalloc((void **)&d_in_, frame_bytes_ * n_));
		HIPCHK(hipHostMalloc((void **)&h_in_, frame_bytes_ * n_, hipHostMallocPortable));
	}
	HIPCHK(hipMalloc((void **)&d_coeff_, (size_t)plan.coeff_elems * 2 * n_));
	HIPCHK(hipMemsetAsync(d_coeff_, 0, (size_t)plan.coeff_elems * 2 * n_, (hipStream_t)stream_));   // pad columns stay zero forever
	HIPCHK(hipHostMalloc((void **)&h_coeff_, (size_t)plan.final_elems * 2 * n_, hipHostMallocPortable));
	jobs_bytes_ = enc_jobs_bytes(n_, plan.num_channels);
	HIPCHK(hipMalloc(&d_jobs_, jobs_bytes_));
	HIPCHK(hipHostMalloc(&h_jobs_, jobs_bytes_, hipHostMallocPortable));
	memset(h_jobs_, 0, jobs_bytes_);

	fill_jobs();
	return 0;
}

// (Re)writes the job tables from plan_: geometry, band addresses and quantizer parameters.  Input frame pointers set with
// set_device_frame() are kept.
void EncodeBatch::fill_jobs()
{
	const FramePlan &plan = plan_;
	const bool own_input = own_input_;
	const bool bayer = plan.pixel_kind == PIX_BYR4 || plan.pixel_kind == PIX_BYR5;
	const int nch = plan.num_channels, mpq = plan.midpoint_prequant;
	EncJobs j = enc_jobs_at(h_jobs_, n_, nch);
	for (int i = 0; i < n_; i++) {
		int16_t *base = d_coeff_ + (size_t)i * plan.coeff_elems;
		dev::FwdYuvJob &y = j.yuv[i];
		y.in = own_input ? d_in_ + frame_bytes_ * i : nullptr; y.in_pitch = in_pitch_;
		y.width = plan.width; y.height = plan.height; y.display_height = plan.display_height;
		y.uyvy = plan.pixel_kind == PIX_2VUY; y.shift = plan.precision - 8;
		for (int c = 0; c < 3; c++) {
			y.out_pitch[c] = plan.ch[c].band[0][0].pitch;
			for (int b = 0; b < 4; b++) { y.out[c][b] = base + plan.ch[c].band[0][b].offset; y.q[c][b] = make_q(plan.ch[c].band[0][b].quant, mpq); }
			// interlaced: the difference-coded band is quantized inside the horizontal filter, midpoint = divisor / prequant without the decrement
			const int dq = plan.ch[c].band[0][2].quant;
			if (plan.interlaced && dq > 1 && mpq >= 2 && mpq < 9) y.q[c][2].mid = dq / mpq;
		}
		if (bayer) {
			const int ppitch = plan.ch[0].band[0][0].pitch * 2;
			dev::BayerJob &bj = j.bayer[i];
			bj.in = own_input ? (const uint16_t *)(d_in_ + frame_bytes_ * i) : nullptr; bj.in_pitch = in_pitch_ / 2;
			bj.width = plan.width; bj.height = plan.height; bj.display_height = plan.display_height;
			bj.out_pitch = ppitch; bj.curve = d_curve_; bj.order = 0; bj.precision = plan.precision; bj.packed12 = plan.pixel_kind == PIX_BYR5;
			for (int c = 0; c < 4; c++) {
				bj.out[c] = d_planes_ + ((size_t)i * 4 + c) * plane_elems_;
				dev::FwdPlaneJob &p = j.l1[(size_t)i * nch + c];
				p.in = bj.out[c]; p.in_pitch = ppitch; p.width = plan.ch[c].width; p.height = plan.ch[c].height; p.prescale = plan.prescale[0];
				p.xstride = 1; p.shift = 0; p.display_height = plan.ch[c].height; p.compand = 0;
				if (bayer_fused_) {                          // the planes are never written: level 1 reads the mosaic
					p.in = (const int16_t *)bj.in; p.in_pitch = bj.in_pitch; p.layout = plan.pixel_kind == PIX_BYR5 ? 11 : 10; p.tail_from = c; p.xstride = bj.order;
					p.shift = plan.precision; p.display_height = plan.display_height; p.curve = d_curve_;
				}
				p.out_pitch = plan.ch[c].band[0][0].pitch;
				for (int b = 0; b < 4; b++) { p.out[b] = base + plan.ch[c].band[0][b].offset; p.q[b] = make_q(plan.ch[c].band[0][b].quant, mpq); }
			}
		}
		if (enc_packed16(plan.pixel_kind))
			for (int c = 0; c < nch; c++) {
				dev::FwdPlaneJob &p = j.l1[(size_t)i * nch + c];
				const uint16_t *frame = own_input ? (const uint16_t *)(d_in_ + frame_bytes_ * i) : nullptr;
				p.in = frame ? (const int16_t *)(frame + enc_word_of_channel(plan.pixel_kind, c)) : nullptr; p.in_pitch = in_pitch_ / 2;
				p.width = plan.ch[c].width; p.height = plan.ch[c].height; p.prescale = plan.prescale[0];
				p.xstride = enc_stride_of_channel(plan.pixel_kind, c, nch); p.shift = 16 - plan.precision; p.display_height = plan.display_height;
				p.compand = (plan.pixel_kind == PIX_B64A || plan.pixel_kind == PIX_RG64) && c == 3;
				p.layout = plan.pixel_kind == PIX_V210 ? c + 1 : 0; p.tail_from = (plan.width - plan.width % 48) / 2;
				if (enc_bytes8(plan.pixel_kind)) { p.layout = plan.pixel_kind == PIX_BGRa ? 5 : 4; p.in_pitch = in_pitch_; p.xstride = plan.pixel_kind == PIX_RG24 ? 3 : 4; p.tail_from = c == 0 ? 1 : (c == 1 ? 2 : (c == 2 ? 0 : 3)); p.compand = c == 3; }     // planes G, R, B(, A) of bytes B, G, R(, A)
				if (enc_rgb10(plan.pixel_kind)) { p.layout = 6; p.in_pitch = in_pitch_ / 4; p.xstride = plan.pixel_kind == PIX_R210 || plan.pixel_kind == PIX_DPX0; p.tail_from = rgb10_shift(plan.pixel_kind, c); }
				if (enc_rgb_as_422(plan)) {
					p.in = frame ? (const int16_t *)(frame + (plan.pixel_kind == PIX_B64A ? 1 : 0)) : nullptr;
					p.layout = 7; p.xstride = plan.pixel_kind == PIX_RG48 ? 3 : 4; p.tail_from = c; p.shift = plan.color_matrix; p.compand = 0;
				}
				if (enc_bytes8(plan.pixel_kind) && plan.encoded_format == ENC_YUV422) { p.layout = plan.pixel_kind == PIX_BGRa ? 9 : 8; p.tail_from = c; p.shift = plan.color_matrix; }
				p.out_pitch = plan.ch[c].band[0][0].pitch;
				for (int b = 0; b < 4; b++) { p.out[b] = base + plan.ch[c].band[0][b].offset; p.q[b] = make_q(plan.ch[c].band[0][b].quant, mpq); }
			}
		for (int lv = 1; lv < 3; lv++)
			for (int c = 0; c < nch; c++) {
				dev::FwdPlaneJob &p = (lv == 1 ? j.l2 : j.l3)[(size_t)i * nch + c];
				const BandDesc &src = plan.ch[c].band[lv - 1][0];
				p.in = base + src.offset; p.in_pitch = src.pitch; p.width = src.width; p.height = src.height; p.prescale = plan.prescale[lv];
				p.xstride = 1; p.shift = 0; p.display_height = src.height; p.compand = 0;
				p.out_pitch = plan.ch[c].band[lv][0].pitch;
				for (int b = 0; b < 4; b++) { p.out[b] = base + plan.ch[c].band[lv][b].offset; p.q[b] = make_q(plan.ch[c].band[lv][b].quant, mpq); }
			}
	}
	jobs_dirty_ = true;
}

// New quantizer tables for the frames encoded from now on (rate feedback re-derives them per frame: quantize.c:186, :2865): the band
// geometry is unchanged, only the divisors in the job tables and in the sample headers move.
int EncodeBatch::update_quant(const FramePlan &plan)
{
	(void)hipSetDevice(device_);
	if (plan.coeff_elems != plan_.coeff_elems || plan.num_channels != plan_.num_channels) return -1;
	if (stream_) HIPCHK(hipStreamSynchronize((hipStream_t)stream_));      // the pinned job table may still be in flight
	EncJobs j = enc_jobs_at(h_jobs_, n_, plan_.num_channels);
	std::vector<const void *> keep_yuv(n_), keep_bayer(n_), keep_l1((size_t)n_ * plan_.num_channels);
	std::vector<int> keep_pitch(n_), keep_bpitch(n_), keep_l1pitch((size_t)n_ * plan_.num_channels);
	for (int i = 0; i < n_; i++) {
		keep_yuv[i] = j.yuv[i].in; keep_pitch[i] = j.yuv[i].in_pitch; keep_bayer[i] = j.bayer[i].in; keep_bpitch[i] = j.bayer[i].in_pitch;
		for (int c = 0; c < plan_.num_channels; c++) { keep_l1[(size_t)i * plan_.num_channels + c] = j.l1[(size_t)i * plan_.num_channels + c].in; keep_l1pitch[(size_t)i * plan_.num_channels + c] = j.l1[(size_t)i * plan_.num_channels + c].in_pitch; }
	}
	plan_ = plan;
	fill_jobs();
	if (!own_input_) for (int i = 0; i < n_; i++) {
		j.yuv[i].in = (const uint8_t *)keep_yuv[i]; j.yuv[i].in_pitch = keep_pitch[i];
		if (plan_.pixel_kind == PIX_BYR4 || plan_.pixel_kind == PIX_BYR5) {
			j.bayer[i].in = (const uint16_t *)keep_bayer[i]; j.bayer[i].in_pitch = keep_bpitch[i];
			if (bayer_fused_) for (int c = 0; c < 4; c++) { j.l1[(size_t)i * 4 + c].in = (const int16_t *)keep_bayer[i]; j.l1[(size_t)i * 4 + c].in_pitch = keep_bpitch[i]; }
		}
		if (enc_packed16(plan_.pixel_kind)) for (int c = 0; c < plan_.num_channels; c++) { j.l1[(size_t)i * plan_.num_channels + c].in = (const int16_t *)keep_l1[(size_t)i * plan_.num_channels + c]; j.l1[(size_t)i * plan_.num_channels + c].in_pitch = keep_l1pitch[(size_t)i * plan_.num_channels + c]; }
	}
	if (ent_ready_) ent_.set_plan(plan);
	return 0;
}

int EncodeBatch::prepare_entropy(size_t sample_cap)
{
	int rc = ent_.prepare(plan_, n_, d_coeff_, plan_.coeff_elems, sample_cap, stream_);
	ent_ready_ = rc == 0;
	if (ent_ready_) fill_block_lists();
	// the level-1 bands can be counted while levels 2 and 3 are still being transformed (measured in round 3 against one launch behind level 3: + 2.8 %)
	if (ent_ready_) ent_.set_level1_event(evl_[0]);
	return rc;
}

// Where the forward strip kernel leaves the block lists of the level-1 bands (the entropy stage owns the buffers: GpuEntropyEncoder::prepare).
void EncodeBatch::fill_block_lists()
{
	EncJobs j = enc_jobs_at(h_jobs_, n_, plan_.num_channels);
	int mask_base[kMaxChannels][kNumBands];
	block_list_layout(plan_, mask_base);
	static_assert((int)kBlockChunkCols == (int)dev::FWD_CHUNK_COLS, "one chunk geometry");
	for (int i = 0; i < n_; i++) {
		dev::FwdBlockLists &l = j.yuv[i].lists;
		const bool on = ent_ready_ && ent_.block_slots();
		l.blocks = on ? (uint4 *)ent_.block_slots() : nullptr;
		l.masks = on ? ent_.block_masks(i) : nullptr;
		l.base = d_coeff_;                               // (block slots are numbered over the whole batch's pyramids, as the entropy stage's segment jobs address them)
		for (int c = 0; c < 3; c++) for (int b = 0; b < 4; b++) l.mask_base[c][b] = mask_base[c][b];
	}
	jobs_dirty_ = true;
}

int EncodeBatch::sync_jobs()
{
	(void)hipSetDevice(device_);
	if (!jobs_dirty_) return 0;
	HIPCHK(hipMemcpyAsync(d_jobs_, h_jobs_, jobs_bytes_, hipMemcpyHostToDevice, (hipStream_t)stream_));
	jobs_dirty_ = false;
	return 0;
}

// frames from this size on are staged in pieces (below it the extra calls cost more than the overlap gives); CFHD_AMD_STAGE_MIN_BYTES: tests lower it
static size_t stage_piece_min_bytes() { static const size_t v = [] { const char *e = getenv("CFHD_AMD_STAGE_MIN_BYTES") /* test hook: pieces at the small frames of the CPU suite too */; return e ? (size_t)atoll(e) : ((size_t)1 << 20); }(); return v; }

int EncodeBatch::upload_frame(int i, const void *frame, int pitch)
{
	(void)hipSetDevice(device_);
	if (!own_input_ || i < 0 || i >= n_) return -1;
	const uint8_t *src = (const uint8_t *)frame;
	if (plan_.pixel_kind == PIX_BYR4) {
		// The reference reads a BYR4 frame as tightly packed rows whatever pitch it was given (frame.c:5376-5377: line1 = data + row * width * 4,
		// line2 = line1 + width * 2, in 16-bit words of the component plane width); only the sign of the pitch moves the start (encoder.c:1957,
		// with the pitch doubled by SampleEncoder.cpp:494 and the display height of the component planes).  Same bytes here.
		if (pitch < 0) src += (ptrdiff_t)(plan_.display_height - 1) * 2 * pitch;
		pitch = in_pitch_;
	}
	if (plan_.pixel_kind == PIX_BYR5) pitch = in_pitch_;      // frame.c:5515 walks the frame as tightly packed row pairs of width * 4 * 3 / 2 bytes, whatever the pitch
	if (pitch < 0) { src += (ptrdiff_t)(in_rows_ - 1) * pitch; pitch = -pitch; }     // encoder.c:1957
	if (pitch >= in_pitch_ && host_buffer_is_registered(src, (size_t)pitch * (in_rows_ - 1) + in_pitch_)) {
		// a buffer the caller registered: DMA straight out of it (the frame is borrowed until the encode completes, as in the reference)
		HIPCHK(hipMemcpy2DAsync(d_in_ + frame_bytes_ * i, (size_t)in_pitch_, src, (size_t)pitch, (size_t)in_pitch_, (size_t)in_rows_, hipMemcpyHostToDevice, (hipStream_t)stream_));
		return 0;
	}
	// a plain buffer: staged through pinned memory, in a few pieces so that the DMA of a piece runs beside the CPU copy of the next (a 1080p frame: 4 MB, 0.2 ms of
	// memcpy and as much of PCIe -- one after the other they were two thirds of a synchronous CFHD_EncodeSample call: 2350 -> 2820 fps).  Only where one caller waits for one
	// frame (set_stage_pieces: the handles of CFHD_EncodeSample / CFHD_DecodeSample): pool workers and gathered decoders run many such copies side by side already, and
	// the extra runtime calls cost them 15-20 % (measured, profiles/r04_j_*).
	const int pieces = frame_bytes_ >= stage_piece_min_bytes() ? stage_pieces_ : 1;
	uint8_t *dst = h_in_ + frame_bytes_ * i;
	for (int k = 0; k < pieces; k++) {
		const int r0 = (int)((long long)in_rows_ * k / pieces), r1 = (int)((long long)in_rows_ * (k + 1) / pieces);
		if (r1 <= r0) continue;
		const size_t off = (size_t)r0 * in_pitch_, bytes = (size_t)(r1 - r0) * in_pitch_;
		if (pitch == in_pitch_) memcpy(dst + off, src + (size_t)r0 * pitch, bytes);
		else for (int r = r0; r < r1; r++) memcpy(dst + (size_t)r * in_pitch_, src + (size_t)r * pitch, (size_t)in_pitch_);
		HIPCHK(hipMemcpyAsync(d_in_ + frame_bytes_ * i + off, dst + off, bytes, hipMemcpyHostToDevice, (hipStream_t)stream_));
	}
	return 0;
}

// The whole batch from host memory, asynchronous on the batch's stream (the frame queue fed from the host: cfhd_amd_batch_submit_host).
int EncodeBatch::upload_frames(const void *frames, size_t frame_stride, int pitch)
{
	(void)hipSetDevice(device_);
	if (!own_input_ || !frames) return -1;
	if (pitch == in_pitch_ && frame_stride == frame_bytes_ && plan_.pixel_kind != PIX_BYR4 && plan_.pixel_kind != PIX_BYR5 && host_buffer_is_registered(frames, frame_bytes_ * (size_t)n_)) {
		HIPCHK(hipMemcpyAsync(d_in_, frames, frame_bytes_ * (size_t)n_, hipMemcpyHostToDevice, (hipStream_t)stream_));
		return 0;
	}
	if (n_ > 8 && pitch >= in_pitch_ && plan_.pixel_kind != PIX_BYR4 && plan_.pixel_kind != PIX_BYR5 && !host_buffer_is_registered(frames, 1)) {
		// plain memory, many frames: staged into the batch's pinned memory by a few threads side by side (one thread copies 4 MB in ~0.3 ms: 128 frames one after the
		// other were 40 ms of a pass), then ONE copy to the device
		const int nt = n_ < 8 ? n_ : 8;
		std::atomic<int> next(0);
		auto work = [&] {
			for (int i; (i = next.fetch_add(1)) < n_;) {
				const uint8_t *src = (const uint8_t *)frames + frame_stride * (size_t)i; uint8_t *dst = h_in_ + frame_bytes_ * (size_t)i;
				if (pitch == in_pitch_) memcpy(dst, src, frame_bytes_);
				else for (int r = 0; r < in_rows_; r++) memcpy(dst + (size_t)r * in_pitch_, src + (size_t)r * pitch, (size_t)in_pitch_);
			}
		};
		std::vector<std::thread> pool;
		for (int k = 1; k < nt; k++) pool.emplace_back(work);
		work();
		for (auto &t : pool) t.join();
		HIPCHK(hipMemcpyAsync(d_in_, h_in_, frame_bytes_ * (size_t)n_, hipMemcpyHostToDevice, (hipStream_t)stream_));
		return 0;
	}
	for (int i = 0; i < n_; i++) { const int rc = upload_frame(i, (const uint8_t *)frames + frame_stride * (size_t)i, pitch); if (rc) return rc; }
	return 0;
}

int EncodeBatch::set_device_frame(int i, const void *d_frame, int pitch)
{
	if (i < 0 || i >= n_) return -1;
	EncJobs j = enc_jobs_at(h_jobs_, n_, plan_.num_channels);
	if (plan_.pixel_kind == PIX_BYR4 || plan_.pixel_kind == PIX_BYR5) {
		j.bayer[i].in = (const uint16_t *)d_frame; j.bayer[i].in_pitch = pitch / 2;
		if (bayer_fused_) for (int c = 0; c < 4; c++) { j.l1[(size_t)i * 4 + c].in = (const int16_t *)d_frame; j.l1[(size_t)i * 4 + c].in_pitch = pitch / 2; }
		jobs_dirty_ = true; return 0;
	}
	if (enc_packed16(plan_.pixel_kind)) {
		for (int c = 0; c < plan_.num_channels; c++) {
			dev::FwdPlaneJob &p = j.l1[(size_t)i * plan_.num_channels + c];
			p.in = (const int16_t *)((const uint16_t *)d_frame + enc_word_of_channel(plan_.pixel_kind, c)); p.in_pitch = enc_bytes8(plan_.pixel_kind) ? pitch : (enc_rgb10(plan_.pixel_kind) ? pitch / 4 : pitch / 2);
			if (enc_rgb_as_422(plan_)) p.in = (const int16_t *)((const uint16_t *)d_frame + (plan_.pixel_kind == PIX_B64A ? 1 : 0));
		}
		jobs_dirty_ = true;
		return 0;
	}
	if (j.yuv[i].in != d_frame || j.yuv[i].in_pitch != pitch) { j.yuv[i].in = (const uint8_t *)d_frame; j.yuv[i].in_pitch = pitch; jobs_dirty_ = true; }
	return 0;
}

// Levels 2 and 3 in the register-strip kernels: one launch per run of equally wide channels (luma | both chroma planes of 4:2:2 | all planes
// of 4:4:4 / Bayer).  false: some channel's geometry is outside what the strip kernels serve (or CFHD_AMD_PLANES=tile) -> tiled kernel.
// The register-strip kernels are the throughput shape: a wave walks down its strip of the plane row by row, so a launch lasts at least one
// such walk (about 80 us at 1080p) however few frames it covers, and fills the chip only with a few hundred frames' worth of strips.  The
// LDS-tiled kernels are many short workgroups: 8x faster on a single frame, slower from a hundred-odd frames on.  Measured crossovers at
// 1080p (tools/small_batch_sweep.sh; frames per launch): plane levels 140-230, forward level 1 about 32, inverse level 1 about 12; larger
// frames count in proportion to their area.  CFHD_AMD_PLANES / _FORWARD / _INVERSE = tile | strip force one shape (A/B runs).
static double frames_1080p_equivalent(const FramePlan &plan, int frames) { return (double)frames * plan.width * plan.height / (1920.0 * 1080.0); }
static int shape_override(const char *name)           // 0: by size, 1: tile, 2: strip (read at every launch: tests switch shapes within one process)
{
	const char *e = getenv(name);
	return !e ? 0 : (strcmp(e, "tile") == 0 ? 1 : (strcmp(e, "strip") == 0 ? 2 : 0));
}
static bool planes_as_strips(const FramePlan &plan, int lv /* wavelet index whose bands are produced / consumed */, int frames)
{
	const int forced = shape_override("CFHD_AMD_PLANES");
	// (the crossover was measured on 4:2:2 frames, whose planes add up to twice the picture; the planes of RGB, RGBA and Bayer frames add up to 3, 4 and 4 times plan.width x plan.height)
	double samples = 0.0;
	for (int c = 0; c < plan.num_channels; c++) samples += (double)plan.ch[c].width * plan.ch[c].height;
	if (forced == 1 || (forced == 0 && frames * samples / (2.0 * 1920.0 * 1080.0) < 160.0)) return false;
	for (int c = 0; c < plan.num_channels; c++) {
		const BandDesc &b = plan.ch[c].band[lv][0];
		if (b.width % dev::SBLK || plan.ch[c].band[lv - 1][0].width != 2 * b.width || plan.ch[c].band[lv - 1][0].height != 2 * b.height ||
		    (plan.ch[c].band[lv - 1][0].pitch & 7) || (b.pitch & 7)) return false;
	}
	return true;
}
template <typename F> static void for_channel_runs(const FramePlan &plan, int lv, F f)
{
	for (int c0 = 0; c0 < plan.num_channels;) {
		int nc = 1;
		while (c0 + nc < plan.num_channels && plan.ch[c0 + nc].band[lv][0].width == plan.ch[c0].band[lv][0].width &&
		       plan.ch[c0 + nc].band[lv][0].height == plan.ch[c0].band[lv][0].height) nc++;
		const BandDesc &b = plan.ch[c0].band[lv][0];
		int glog = 0; while ((1 << glog) < b.width / dev::SBLK) glog++;
		// more than 64 blocks of 8 columns: one plane per wave in segments of PLSTEP blocks (the lanes at a segment's ends feed their neighbours)
		const int nblk = b.width / dev::SBLK, nseg = nblk > 64 ? (nblk + dev::PLSTEP - 1) / dev::PLSTEP : 1;
		if (nseg > 1) glog = 6;
		f(c0, nc, glog, b, nseg);
		c0 += nc;
	}
}

// k_fwd_yuv422_strip serves progressive 4:2:2 frames of whole 32-pixel blocks whose rows are 16-byte aligned;
// everything else (and CFHD_AMD_FORWARD=tile, for A/B runs) takes the LDS-tiled k_fwd_yuv422.  Both produce the same coefficients.
bool EncodeBatch::strip_forward() const
{
	const int forced = shape_override("CFHD_AMD_FORWARD");
	const int act = active_ > 0 && active_ < n_ ? active_ : n_;
	if (forced == 1 || (forced == 0 && frames_1080p_equivalent(plan_, act) < 32.0)) return false;
	if (plan_.interlaced || plan_.encoded_format != ENC_YUV422 || plan_.width % 32) return false;
	EncJobs j = enc_jobs_at(h_jobs_, n_, plan_.num_channels);
	for (int i = 0; i < n_; i++) if (((uintptr_t)j.yuv[i].in & 15) || (j.yuv[i].in_pitch & 15)) return false;
	return true;
}

// k_fwd_frame_yuv422_strip: the same geometry rule for interlaced 4:2:2 frames (else the LDS-tiled k_fwd_frame_yuv422)
bool EncodeBatch::strip_forward_frame() const
{
	const int forced = shape_override("CFHD_AMD_FORWARD");
	const int act = active_ > 0 && active_ < n_ ? active_ : n_;
	if (forced == 1 || (forced == 0 && frames_1080p_equivalent(plan_, act) < 32.0)) return false;
	if (!plan_.interlaced || plan_.encoded_format != ENC_YUV422 || plan_.width % 32 || !(plan_.pixel_kind == PIX_YUY2 || plan_.pixel_kind == PIX_2VUY)) return false;
	EncJobs j = enc_jobs_at(h_jobs_, n_, plan_.num_channels);
	for (int i = 0; i < n_; i++) if (((uintptr_t)j.yuv[i].in & 15) || (j.yuv[i].in_pitch & 15)) return false;
	return true;
}

// k_fwd_packed16_strip serves RG48 / b64a frames of whole 8-pixel blocks whose rows are 16-byte aligned, from the launch size on at which
// the strip kernels pay (as strip_forward()); everything else takes the LDS-tiled k_fwd_packed16.
bool EncodeBatch::strip_forward_packed16() const
{
	const int forced = shape_override("CFHD_AMD_FORWARD");
	const int act = active_ > 0 && active_ < n_ ? active_ : n_;
	if (forced == 1 || (forced == 0 && frames_1080p_equivalent(plan_, act) < 32.0)) return false;
	if (!is_packed16(plan_.pixel_kind) || plan_.encoded_format == ENC_YUV422 || plan_.width % 8 || plan_.num_channels < 3) return false;
	EncJobs j = enc_jobs_at(h_jobs_, n_, plan_.num_channels);
	for (int i = 0; i < n_; i++) {
		const dev::FwdPlaneJob &p = j.l1[(size_t)i * plan_.num_channels];
		const uintptr_t frame = (uintptr_t)((const uint16_t *)p.in - packed_word_of_channel(plan_.pixel_kind, 0));
		if ((frame & 15) || ((p.in_pitch * 2) & 15) || p.layout != 0 || p.width % 8) return false;
	}
	return true;
}

// k_fwd_bayer_strip serves BYR4 mosaics whose component planes are whole 8-column blocks wide and whose rows are 16-byte aligned, from the launch size on at which
// the strip kernels pay; BYR5, small launches and CFHD_AMD_FORWARD=tile take k_unpack_byr4 + k_fwd_plane.
bool EncodeBatch::strip_forward_bayer() const
{
	const int forced = shape_override("CFHD_AMD_FORWARD");
	const int act = active_ > 0 && active_ < n_ ? active_ : n_;
	if (forced == 1 || (forced == 0 && frames_1080p_equivalent(plan_, act) < 32.0)) return false;
	if (plan_.pixel_kind != PIX_BYR4 || bayer_fused_ || plan_.width % 8 || plan_.num_channels != 4) return false;
	EncJobs j = enc_jobs_at(h_jobs_, n_, plan_.num_channels);
	for (int i = 0; i < n_; i++) if (((uintptr_t)j.bayer[i].in & 15) || ((j.bayer[i].in_pitch * 2) & 15) || j.bayer[i].order != j.bayer[0].order) return false;
	return true;
}

const char *EncodeBatch::level_kernel(int level) const
{
	const int act = active_ > 0 && active_ < n_ ? active_ : n_;
	if (level > 0) return planes_as_strips(plan_, level, act) ? "k_fwd_plane_strip" : "k_fwd_plane";
	if (strip_forward_bayer()) return "k_fwd_bayer_strip";
	if (plan_.pixel_kind == PIX_BYR4 || plan_.pixel_kind == PIX_BYR5) return bayer_fused_ ? "k_fwd_packed16" : "k_unpack_byr4+k_fwd_plane";
	if (strip_forward_packed16()) return "k_fwd_packed16_strip";
	if (enc_packed16(plan_.pixel_kind)) return "k_fwd_packed16";
	if (plan_.interlaced) return strip_forward_frame() ? "k_fwd_frame_yuv422_strip" : "k_fwd_frame_yuv422";
	return strip_forward() ? (block_lists_forward() ? "k_fwd_yuv422_strip_blocks" : "k_fwd_yuv422_strip") : "k_fwd_yuv422";
}

// Level-1 bands as block lists for the GPU entropy stage (k_fwd_yuv422_strip_blocks -> k_ent_count_blocks): wherever the 4:2:2 strip kernel runs in front of the
// GPU entropy stage.  CFHD_AMD_BLOCKS=0: dense bands and k_ent_count (A/B runs).
bool EncodeBatch::block_lists_forward() const
{
	static const int blocks_env = [] { const char *e = getenv("CFHD_AMD_BLOCKS"); return e ? atoi(e) : 1; }();
	return blocks_env && ent_ready_ && ent_.block_slots() && (plan_.pixel_kind == PIX_YUY2 || plan_.pixel_kind == PIX_2VUY) && strip_forward();
}

int EncodeBatch::launch_forward(bool coeffs_needed)
{
	(void)hipSetDevice(device_);
	const int dense_env = 0;
	const bool blocks = block_lists_forward();
	if (ent_ready_) ent_.set_block_lists(blocks);
	int rc = sync_jobs();
	if (rc) return rc;
	hipStream_t st = (hipStream_t)stream_;
	const int nch = plan_.num_channels;
	const int act = active_ > 0 && active_ < n_ ? active_ : n_;      // frames 0 .. act-1 of the batch hold frames (set_active)
	EncJobs j = enc_jobs_at(d_jobs_, n_, nch);
	(void)hipGetLastError();                            // drop stale sticky errors: the check below is for these launches only
	timed_ = true;
	HIPCHK(hipEventRecord((hipEvent_t)ev0_, st));
	if ((plan_.pixel_kind == PIX_BYR4 || plan_.pixel_kind == PIX_BYR5) && bayer_fused_) {
		// level 1 straight from the mosaic: every component plane's loader computes its samples from the photosite quads (the planes k_unpack_byr4 would
		// write -- 8 bytes per quad out, 8 back in -- never exist)
		dim3 grid(((plan_.width / 2 + dev::TW - 1) / dev::TW) * nch, (plan_.height / 2 + dev::TH - 1) / dev::TH, act);
		dev::k_fwd_packed16<<<grid, dev::NTHREADS, 0, st>>>(j.l1, nch);
	} else if (strip_forward_bayer()) {
		// level 1 straight from the mosaic, every photosite read and curved once, all four component planes from one pass (cfhd_kernels.h k_fwd_bayer_strip)
		const int nseg = (plan_.width / 8 + dev::PSTEP - 1) / dev::PSTEP, nstrips = (plan_.height / 2 + dev::PSR - 1) / dev::PSR, waves = act * nseg * nstrips;
		dev::k_fwd_bayer_strip<<<(waves + 3) / 4, dev::NTHREADS, 0, st>>>(j.l1, j.bayer, act, nseg, nstrips);
	} else if (plan_.pixel_kind == PIX_BYR4 || plan_.pixel_kind == PIX_BYR5) {
		dev::k_unpack_byr4<<<dim3((plan_.width / 2 + dev::NTHREADS - 1) / dev::NTHREADS, plan_.height, act), dev::NTHREADS, 0, st>>>(j.bayer);      // two quads per thread
		dim3 grid((plan_.width / 2 + dev::TW - 1) / dev::TW, (plan_.height / 2 + dev::TH - 1) / dev::TH, act * nch);
		dev::k_fwd_plane<<<grid, dev::NTHREADS, 0, st>>>(j.l1);
	} else if (strip_forward_packed16()) {
		const int nseg = (plan_.width / 8 + dev::PSTEP - 1) / dev::PSTEP, nstrips = (plan_.height / 2 + dev::PSR - 1) / dev::PSR, waves = act * nseg * nstrips;
		const dim3 grid((waves + 3) / 4);
		if (plan_.pixel_kind == PIX_RG48) dev::k_fwd_packed16_strip<3, 3><<<grid, dev::NTHREADS, 0, st>>>(j.l1, act, nseg, nstrips);
		else if (nch == 4) dev::k_fwd_packed16_strip<4, 4><<<grid, dev::NTHREADS, 0, st>>>(j.l1, act, nseg, nstrips);
		else dev::k_fwd_packed16_strip<4, 3><<<grid, dev::NTHREADS, 0, st>>>(j.l1, act, nseg, nstrips);
	} else if (enc_packed16(plan_.pixel_kind)) {
		dim3 grid(((plan_.width / 2 + dev::TW - 1) / dev::TW) * nch, (plan_.height / 2 + dev::TH - 1) / dev::TH, act);
		dev::k_fwd_packed16<<<grid, dev::NTHREADS, 0, st>>>(j.l1, nch);
	} else if (strip_forward_frame()) {
		static_assert(sizeof(dev::FwdFrameJob) == sizeof(dev::FwdYuvJob) && offsetof(dev::FwdFrameJob, q) == offsetof(dev::FwdYuvJob, q), "the two level-1 jobs share one table");
		const int nseg = (plan_.width / 16 + dev::SSEG - 1) / dev::SSEG;      // segments of 124 luma blocks (1984 pixels)
		dev::k_fwd_frame_yuv422_strip<<<dim3(nseg, (plan_.height / 2 + dev::SRI - 1) / dev::SRI, act), dev::NTHREADS, 0, st>>>((const dev::FwdFrameJob *)j.yuv);
	} else if (plan_.interlaced) {
		static_assert(sizeof(dev::FwdFrameJob) == sizeof(dev::FwdYuvJob) && offsetof(dev::FwdFrameJob, q) == offsetof(dev::FwdYuvJob, q), "the two level-1 jobs share one table");
		dim3 grid((plan_.width / 2 + dev::FTW - 1) / dev::FTW, (plan_.height / 2 + dev::FRW - 1) / dev::FRW, act);
		dev::k_fwd_frame_yuv422<<<grid, dev::NTHREADS, 0, st>>>((const dev::FwdFrameJob *)j.yuv);
	} else if (strip_forward()) {
		const int nseg = (plan_.width / 16 + dev::SSEG - 1) / dev::SSEG;      // segments of 124 luma blocks (1984 pixels)
		const dim3 grid(nseg, (plan_.height / 2 + dev::SRF - 1) / dev::SRF, act);
		if (blocks && (coeffs_needed || dense_env)) dev::k_fwd_yuv422_strip_blocks_dense<<<grid, dev::NTHREADS, 0, st>>>(j.yuv);
		else if (blocks) dev::k_fwd_yuv422_strip_blocks<<<grid, dev::NTHREADS, 0, st>>>(j.yuv);
		else dev::k_fwd_yuv422_strip<<<grid, dev::NTHREADS, 0, st>>>(j.yuv);
	} else {
		dim3 grid((plan_.width / 2 + dev::TW - 1) / dev::TW, (plan_.height / 2 + dev::TH - 1) / dev::TH, act);
		dev::k_fwd_yuv422<<<grid, dev::NTHREADS, 0, st>>>(j.yuv);
	}
	for (int lv = 1; lv < 3; lv++) {
		HIPCHK(hipEventRecord((hipEvent_t)evl_[lv - 1], st));
		const BandDesc &src = plan_.ch[0].band[lv - 1][0];     // luma is the widest plane of the level
		const dev::FwdPlaneJob *jobs = lv == 1 ? j.l2 : j.l3;
		if (planes_as_strips(plan_, lv, act)) {
			const int n = act;
			for_channel_runs(plan_, lv, [&](int c0, int nc, int glog, const BandDesc &b, int nseg) {
				const int nstrips = (b.height + dev::SRP - 1) / dev::SRP, per_wave = nseg > 1 ? 1 : 64 >> glog, waves = ((n * nc + per_wave - 1) / per_wave) * nstrips * nseg;
				dev::k_fwd_plane_strip<<<(waves + 3) / 4, dev::NTHREADS, 0, st>>>(jobs, n, nch, c0, nc, glog, nstrips, 2 * b.width, 2 * b.height, nseg);
			});
			continue;
		}
		dim3 grid((src.width / 2 + dev::TW - 1) / dev::TW, (src.height / 2 + dev::TH - 1) / dev::TH, act * nch);
		dev::k_fwd_plane<<<grid, dev::NTHREADS, 0, st>>>(jobs);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord((hipEvent_t)ev1_, st));
	return 0;
}

int EncodeBatch::download_coeffs()
{
	(void)hipSetDevice(device_);
	HIPCHK(hipMemcpy2DAsync(h_coeff_, (size_t)plan_.final_elems * 2, d_coeff_, (size_t)plan_.coeff_elems * 2,
	                        (size_t)plan_.final_elems * 2, n_, hipMemcpyDeviceToHost, (hipStream_t)stream_));
	return 0;
}

int EncodeBatch::wait()
{
	(void)hipSetDevice(device_);
	HIPCHK(hipStreamSynchronize((hipStream_t)stream_));
	float ms = 0;
	if (!timed_) return 0;
	timed_ = false;
	if (hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)ev1_) == hipSuccess) kernel_ms_ = ms;
	if (hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)evl_[0]) == hipSuccess) level_ms_[0] = ms;
	if (hipEventElapsedTime(&ms, (hipEvent_t)evl_[0], (hipEvent_t)evl_[1]) == hipSuccess) level_ms_[1] = ms;
	if (hipEventElapsedTime(&ms, (hipEvent_t)evl_[1], (hipEvent_t)ev1_) == hipSuccess) level_ms_[2] = ms;
	return 0;
}

// =============================================================================================
// DecodeBatch
// =============================================================================================
DecodeBatch::DecodeBatch() {}
DecodeBatch::~DecodeBatch() { release(); }

void DecodeBatch::release()
{
	(void)hipSetDevice(device_);
	if (stream_) hipStreamSynchronize((hipStream_t)stream_);
	ent_ready_ = false;
	if (d_out_) hipFree(d_out_);
	if (d_tmp_) { hipFree(d_tmp_); d_tmp_ = nullptr; }
	if (d_restore_) { hipFree(d_restore_); d_restore_ = nullptr; }
	if (h_out_) hipHostFree(h_out_);
	if (d_coeff_) hipFree(d_coeff_);
	if (h_coeff_) hipHostFree(h_coeff_);
	if (d_jobs_) hipFree(d_jobs_);
	if (h_jobs_) hipHostFree(h_jobs_);
	if (ev0_) hipEventDestroy((hipEvent_t)ev0_);
	if (ev1_) hipEventDestroy((hipEvent_t)ev1_);
	for (int k = 0; k < 2; k++) if (evl_[k]) { hipEventDestroy((hipEvent_t)evl_[k]); evl_[k] = nullptr; }
	if (evdep_) { hipEventDestroy((hipEvent_t)evdep_); evdep_ = nullptr; }
	for (void *e : piece_ev_) if (e) hipEventDestroy((hipEvent_t)e);
	piece_ev_.clear(); out_pieces_.clear();
	for (int k = 0; k < 3; k++) if (ev2_[k]) { hipEventDestroy((hipEvent_t)ev2_[k]); ev2_[k] = nullptr; }
	if (stream2_) { device_stream_destroy(stream2_); stream2_ = nullptr; }
	if (stream_) device_stream_destroy(stream_);
	d_out_ = h_out_ = nullptr; d_coeff_ = h_coeff_ = nullptr; d_jobs_ = h_jobs_ = nullptr; stream_ = ev0_ = ev1_ = nullptr; n_ = 0;
}

int DecodeBatch::prepare(const FramePlan &plan, int nframes, int out_kind, bool own_output, bool half)
{
	int rc = device_init();
	if (rc) return rc;
	release();
	device_ = device_current(); (void)hipSetDevice(device_);      // (release() went to the device of the buffers it freed)
	half_ = half;
	// v210 output (10-bit 4:2:2, three samples per 32-bit word): the reference's samples are its YU64 words >> 6 (oracle/cfhd_oracle_inv.c
	// orc_inv_spatial_to_v210, pinned on the reference decoder for widths that are multiples of six) -- the frames are computed as YU64 rows into a
	// scratch buffer and k_yu64_to_v210 packs them into the output
	v210_ = out_kind == PIX_V210;
	if (v210_) { if ((half ? plan.width / 2 : plan.width) % 6 || !own_output) { g_err = "v210 output: widths that are multiples of 6"; return -2; } out_kind = PIX_YU64; }      // (half resolution: k_half_yu64 feeds the same repack)
	// RG24 output of 4:2:2 samples: the reference computes the three planes as 16-bit rows (the YU64 route) and converts them pixel by pixel
	// (convert.c:11392 ConvertRow16uToDitheredRGB, oracle orc_inv_spatial_to_rgb24_of_yuv422): the same two steps here
	lowpass_kind_ = out_kind;
	rgb24_of_422_ = out_kind == PIX_RG24 && plan.encoded_format == ENC_YUV422;
	if (rgb24_of_422_) { if (!own_output) { g_err = "RG24 output of 4:2:2 samples: into the library's own output frames"; return -2; } if (!half) out_kind = PIX_YU64; }
	const bool rgb24_half = rgb24_of_422_ && half;      // (half resolution: k_half_rgb24 straight from the lowpass planes, no scratch frame)
	if (rgb24_half) rgb24_of_422_ = false;
	// RG48 / b64a output of 4:2:2 samples (bayer.c:11916 Row16uFull2OutputFormat: the 16-bit rows through RGB2YUV.c:1308 / :1760): the YU64 route into the scratch frame,
	// then k_yu64_to_rgb16.  BGRA / BGRa output of 4:2:2 samples: the last level with the reference's fused colour conversion (k_inv_yuv422_rgb32).
	rgb16_of_422_ = (out_kind == PIX_RG48 || out_kind == PIX_B64A) && plan.encoded_format == ENC_YUV422; rgb16_b64a_ = out_kind == PIX_B64A;
	const int final_kind = out_kind;
	if (rgb16_of_422_) { if (!own_output) { g_err = "RG48 / b64a output of 4:2:2 samples: into the library's own output frames"; return -2; } if (!half) out_kind = PIX_YU64; }
	rgb32_of_422_ = (out_kind == PIX_BGRA || out_kind == PIX_BGRa) && plan.encoded_format == ENC_YUV422;
	if (rgb32_of_422_ && (!own_output || (half && (plan.width / 2) % 16))) { g_err = "BGRA / BGRa output of 4:2:2 samples: into the library's own output frames; half widths that are multiples of 16"; return -2; }
	// (half resolution of the four: k_half_rgb24's other modes straight from the lowpass planes -- frame.c:8504 RGB32 branch, frame.c:9567 -- no scratch frame, no last level)
	const bool rgb_half_of_422 = half && (rgb16_of_422_ || rgb32_of_422_);
	if (rgb_half_of_422) { rgb16_of_422_ = false; rgb32_of_422_ = false; }
	// BYR4 output of Bayer samples (decoder.c:14738 + bayer.c:13233 GenerateBYR2): the four component planes as 16-bit rows -- the RG48 route with four planes,
	// four words per photosite quad -- then k_bayer_to_byr4
	byr4_ = out_kind == PIX_BYR4;
	if (byr4_) { if (plan.encoded_format != ENC_BAYER || !own_output || half) { g_err = "BYR4 output: Bayer samples, full resolution"; return -2; } out_kind = PIX_RG48; }
	const bool repack = v210_ || rgb24_of_422_ || byr4_ || rgb16_of_422_;

	const bool yuv_ok = (out_kind == PIX_YUY2 || out_kind == PIX_2VUY || rgb32_of_422_) && plan.encoded_format == ENC_YUV422;
	// (b64a from an RGB 4:4:4 sample: the three colour planes and a constant alpha word, full resolution)
	const bool rgb_ok = ((out_kind == PIX_RG48 && plan.encoded_format == ENC_RGB444) || (out_kind == PIX_B64A && plan.encoded_format == ENC_RGBA4444) ||
	                     (out_kind == PIX_B64A && plan.encoded_format == ENC_RGB444) || (out_kind == PIX_RG48 && plan.encoded_format == ENC_RGBA4444) || byr4_) &&
	                    plan.ch[0].band[0][0].width >= 16;   // k_inv_packed16's tail-column rule assumes the reference's vector path
	const bool yu64_ok = out_kind == PIX_YU64 && plan.encoded_format == ENC_YUV422 && plan.ch[1].band[0][0].width >= 16;
	const bool rgb8_ok = dec_rgb8(out_kind) && (plan.encoded_format == ENC_RGB444 || (plan.encoded_format == ENC_RGBA4444 && out_kind != PIX_RG24)) && plan.ch[0].band[0][0].width >= 16 && plan.ch[0].band[0][0].width % 2 == 0;
	const bool rgb10_ok = dec_rgb10(out_kind) && plan.encoded_format == ENC_RGB444 && plan.ch[0].band[0][0].width >= 16;
	if (!yuv_ok && !rgb_ok && !yu64_ok && !rgb8_ok && !rgb10_ok && !rgb24_half && !rgb_half_of_422) { g_err = "output format not supported by the GPU path yet"; return -2; }
	plan_ = plan; n_ = nframes; out_kind_ = out_kind; own_output_ = own_output;
	HIPCHK((hipError_t)device_stream_create(&stream_));
	HIPCHK(hipEventCreate((hipEvent_t *)&ev0_));
	HIPCHK(hipEventCreate((hipEvent_t *)&ev1_));
	for (int k = 0; k < 2; k++) HIPCHK(hipEventCreate((hipEvent_t *)&evl_[k]));
	for (int k = 0; k < 3; k++) HIPCHK(hipEventCreate((hipEvent_t *)&ev2_[k]));
	// (a batch of the frame queue creates its second stream only for the arrangement that uses it -- CFHD_AMD_TILES_SPLIT=1, launch_inverse --: a stream that is never used
	// still takes its turn when the runtime deals its 4 hardware queues to the streams in creation order)
	{ const char *se = getenv("CFHD_AMD_TILES_SPLIT"); if ((se && se[0] == '1') || !device_streams_are_lean()) HIPCHK((hipError_t)device_stream_create(&stream2_)); }      // (not lean: the C ABI's handles, cfhd_entropy_gpu.h device_streams_lean)
	out_rows_ = half ? plan.display_height / 2 : plan.display_height;
	out_pitch_ = byr4_ ? plan.width * 8 : packed_frame_pitch(out_kind, half ? plan.width / 2 : plan.width);      // (BYR4: the scratch rows hold four words per quad)
	frame_bytes_ = (size_t)out_pitch_ * out_rows_;
	uint8_t *job_out = nullptr; size_t job_frame_bytes = frame_bytes_;      // where the last-level kernel writes frame i: the output, or the YU64 scratch of v210 output
	if (repack) {
		HIPCHK(hipMalloc((void **)&d_tmp_, frame_bytes_ * n_));
		job_out = d_tmp_; tmp_pitch_ = out_pitch_; tmp_frame_bytes_ = frame_bytes_;
		if (byr4_) {
			out_rows_ = 2 * plan.display_height; out_pitch_ = packed_frame_pitch(PIX_BYR4, 2 * plan.width);      // (the plan counts quads)
			std::vector<uint16_t> curve((size_t)1 << kBayerCurveBits);
			build_bayer_linear_restore_curve(curve.data());
			HIPCHK(hipMalloc((void **)&d_restore_, curve.size() * 2));
			HIPCHK(hipMemcpy(d_restore_, curve.data(), curve.size() * 2, hipMemcpyHostToDevice));
		} else out_pitch_ = packed_frame_pitch(v210_ ? PIX_V210 : (rgb16_of_422_ ? final_kind : PIX_RG24), half ? plan.width / 2 : plan.width);
		frame_bytes_ = (size_t)out_pitch_ * out_rows_;
	}
	if (own_output) {
		HIPCHK(hipMalloc((void **)&d_out_, frame_bytes_ * n_));
		HIPCHK(hipHostMalloc((void **)&h_out_, frame_bytes_ * n_, hipHostMallocPortable));
		if (v210_) HIPCHK(hipMemsetAsync(d_out_, 0, frame_bytes_ * n_, (hipStream_t)stream_));      // (row padding beyond the last whole group of 48 pixels stays zero)
	}
	if (!repack) job_out = d_out_;
	const int job_pitch = repack ? tmp_pitch_ : out_pitch_;
	HIPCHK(hipMalloc((void **)&d_coeff_, (size_t)plan.coeff_elems * 2 * n_));
	HIPCHK(hipMemsetAsync(d_coeff_, 0, (size_t)plan.coeff_elems * 2 * n_, (hipStream_t)stream_));
	HIPCHK(hipHostMalloc((void **)&h_coeff_, (size_t)plan.final_elems * 2 * n_, hipHostMallocPortable));
	memset(h_coeff_, 0, (size_t)plan.final_elems * 2 * n_);
	jobs_bytes_ = dec_jobs_bytes(n_, plan.num_channels);
	HIPCHK(hipMalloc(&d_jobs_, jobs_bytes_));
	HIPCHK(hipHostMalloc(&h_jobs_, jobs_bytes_, hipHostMallocPortable));
	memset(h_jobs_, 0, jobs_bytes_);

	const int nch = plan.num_channels, onch = dec_out_channels(out_kind, plan);
	DecJobs j = dec_jobs_at(h_jobs_, n_, nch);
	for (int i = 0; i < n_; i++) {
		int16_t *base = d_coeff_ + (size_t)i * plan.coeff_elems;
		for (int lv = 2; lv >= 1; lv--)
			for (int c = 0; c < nch; c++) {
				dev::InvPlaneJob &p = (lv == 2 ? j.l3 : j.l2)[(size_t)i * nch + c];
				for (int b = 0; b < 4; b++) p.band[b] = base + plan.ch[c].band[lv][b].offset;
				p.band_pitch = plan.ch[c].band[lv][0].pitch;
				p.width = plan.ch[c].band[lv][0].width; p.height = plan.ch[c].band[lv][0].height;
				p.descale = plan.prescale[lv];                                  // wavelet.c:5685: prescaled levels use the Descale variant
				p.out = base + plan.ch[c].band[lv - 1][0].offset; p.out_pitch = plan.ch[c].band[lv - 1][0].pitch;
				p.xstride = 1; p.precision = 0; p.display_height = 2 * p.height;
			}
		if (half && plan.encoded_format == ENC_YUV422 && (dec_rgb8(out_kind) || out_kind == PIX_RG48 || out_kind == PIX_B64A)) {      // k_half_rgb24: RG24, BGRA / BGRa, RG48, b64a of a 4:2:2 sample
			dev::HalfYuvJob &hj = j.half[i];
			for (int c = 0; c < 3; c++) { hj.ll[c] = base + plan.ch[c].band[0][0].offset; hj.pitch[c] = plan.ch[c].band[0][0].pitch; }
			hj.width = plan.ch[0].band[0][0].width; hj.rows = out_rows_; hj.uyvy = 0; hj.matrix = plan.color_matrix;
			hj.mode = out_kind == PIX_RG24 ? 0 : (out_kind == PIX_RG48 ? 2 : (out_kind == PIX_B64A ? 3 : 1)); hj.bottom_up = out_kind == PIX_BGRA;
			hj.out = own_output ? d_out_ + frame_bytes_ * i : nullptr; hj.out_pitch = out_pitch_;
			continue;
		}
		if (half && plan.encoded_format != ENC_YUV422 && (dec_rgb8(out_kind) || dec_rgb10(out_kind) || (out_kind == PIX_B64A && nch == 3))) {      // k_half_rgb
			dev::HalfPackedJob &hp = j.halfp[i];
			for (int c = 0; c < 3; c++) { hp.ll[c] = base + plan.ch[c].band[0][0].offset; hp.word[c] = dec_rgb10(out_kind) ? rgb10_shift(out_kind, c) : 0; }
			hp.pitch = plan.ch[0].band[0][0].pitch; hp.width = plan.ch[0].band[0][0].width; hp.rows = out_rows_; hp.nch = dec_rgb8(out_kind) && nch == 4 ? 4 : 3;
			if (hp.nch == 4) hp.ll[3] = base + plan.ch[3].band[0][0].offset;
			hp.mode = dec_rgb8(out_kind) ? 1 : (dec_rgb10(out_kind) ? 2 : 3);
			hp.bytes = dec_rgb8(out_kind) ? rgb8_bytes(out_kind) : 0; hp.bottom_up = out_kind == PIX_RG24 || out_kind == PIX_BGRA;
			hp.big_endian = out_kind == PIX_R210 || out_kind == PIX_DPX0; hp.dither_seed = 0x9E3779B9u * (uint32_t)(i + 1);
			hp.out = own_output ? (uint16_t *)(d_out_ + frame_bytes_ * i) : nullptr; hp.out_pitch = out_pitch_;
			continue;
		}
		if (half && is_packed16(out_kind)) {
			dev::HalfPackedJob &hp = j.halfp[i];
			for (int c = 0; c < onch; c++) { hp.ll[c] = base + plan.ch[c].band[0][0].offset; hp.word[c] = packed_word_of_channel(out_kind, c); }
			hp.pitch = plan.ch[0].band[0][0].pitch; hp.width = plan.ch[0].band[0][0].width; hp.rows = out_rows_; hp.nch = onch;
			hp.shift = 16 - plan.precision - 2; hp.alpha = out_kind == PIX_B64A;
			hp.out = own_output ? (uint16_t *)(d_out_ + frame_bytes_ * i) : nullptr; hp.out_pitch = out_pitch_;
		}
		if (half && out_kind == PIX_RG24 && plan.encoded_format == ENC_YUV422) {      // k_half_rgb24
			dev::HalfYuvJob &hj = j.half[i];
			for (int c = 0; c < 3; c++) { hj.ll[c] = base + plan.ch[c].band[0][0].offset; hj.pitch[c] = plan.ch[c].band[0][0].pitch; }
			hj.width = plan.ch[0].band[0][0].width; hj.rows = out_rows_; hj.uyvy = 0; hj.matrix = plan.color_matrix;
			hj.out = own_output ? d_out_ + frame_bytes_ * i : nullptr; hj.out_pitch = out_pitch_;
			continue;
		}
		if (half && out_kind == PIX_YU64) {                 // k_half_yu64
			dev::HalfYuvJob &hj = j.half[i];
			for (int c = 0; c < 3; c++) { hj.ll[c] = base + plan.ch[c].band[0][0].offset; hj.pitch[c] = plan.ch[c].band[0][0].pitch; }
			hj.width = plan.ch[0].band[0][0].width; hj.rows = out_rows_; hj.uyvy = 0;
			hj.out = own_output ? job_out + job_frame_bytes * i : nullptr; hj.out_pitch = job_pitch;      // (v210 output: the scratch frame k_yu64_to_v210 reads)
			continue;
		}
		if (dec_planes16(out_kind) && !rgb32_of_422_) {
			for (int c = 0; c < onch; c++) {
				dev::InvPlaneJob &p = j.l1[(size_t)i * onch + c];
				for (int b = 0; b < 4; b++) p.band[b] = base + plan.ch[c].band[0][b].offset;
				p.band_pitch = plan.ch[c].band[0][0].pitch;
				p.width = plan.ch[c].band[0][0].width; p.height = plan.ch[c].band[0][0].height; p.descale = 0;
				uint16_t *frame = own_output ? (uint16_t *)(job_out + job_frame_bytes * i) : nullptr;
				p.out = frame ? dec_plane_out(frame, out_kind, c) : nullptr; p.out_pitch = dec_rgb8(out_kind) ? job_pitch : job_pitch / 2;
				p.xstride = dec_stride_of_channel(out_kind, c, onch); p.precision = plan.precision; p.display_height = plan.display_height;
				p.alpha = (out_kind == PIX_B64A || dec_rgb8(out_kind)) && c == 3;
				p.alpha_const = out_kind == PIX_B64A && nch == 3 ? 0xfff0 : 0;
				p.bytes8 = dec_rgb8(out_kind) ? (nch == 4 ? 2 : 1) : 0;        // 2: BGRA / BGRa of an RGBA 4:4:4:4 sample (alpha from the fourth plane, no dither)
				p.bottom_up = out_kind == PIX_RG24 || out_kind == PIX_BGRA; p.dither_seed = 0x9E3779B9u * (uint32_t)(i + 1);
				if (dec_rgb10(out_kind)) { p.out = (int16_t *)frame; p.out_pitch = job_pitch / 4; p.bit_shift = rgb10_shift(out_kind, c); p.big_endian = out_kind == PIX_R210 || out_kind == PIX_DPX0; }
			}
			continue;
		}
		if (half && !is_packed16(out_kind)) {
			dev::HalfYuvJob &hj = j.half[i];
			for (int c = 0; c < 3; c++) { hj.ll[c] = base + plan.ch[c].band[0][0].offset; hj.pitch[c] = plan.ch[c].band[0][0].pitch; }
			hj.width = plan.ch[0].band[0][0].width; hj.rows = out_rows_; hj.uyvy = out_kind == PIX_2VUY;
			hj.out = own_output ? d_out_ + frame_bytes_ * i : nullptr; hj.out_pitch = out_pitch_;
		}
		dev::InvYuvJob &y = j.yuv[i];
		for (int c = 0; c < 3; c++) { y.band_pitch[c] = plan.ch[c].band[0][0].pitch; for (int b = 0; b < 4; b++) y.band[c][b] = base + plan.ch[c].band[0][b].offset; }
		y.width = plan.ch[0].band[0][0].width; y.height = plan.ch[0].band[0][0].height; y.display_height = plan.display_height;
		y.uyvy = out_kind == PIX_2VUY; y.shift = plan.precision - 8; y.dither_seed = 0x9E3779B9u * (uint32_t)(i + 1);
		y.out = own_output ? d_out_ + frame_bytes_ * i : nullptr; y.out_pitch = out_pitch_;
		y.bottom_up = out_kind == PIX_BGRA; y.matrix_601 = plan.color_matrix >= 2;       // (k_inv_yuv422_rgb32)
		y.masks = nullptr;                               // (block lists: prepare_entropy() knows the mask buffer)
		{ int mb[kMaxChannels][kNumBands]; dec_block_list_layout(plan, mb); for (int c = 0; c < 3; c++) for (int b = 0; b < 4; b++) y.mask_base[c][b] = mb[c][b]; }
	}
	jobs_dirty_ = true;
	return 0;
}

int DecodeBatch::prepare_entropy(size_t sample_cap)
{
	ent_.set_skip_level1(half_);                         // half resolution never looks at the level-1 highpass bands
	int rc = ent_.prepare(plan_, n_, d_coeff_, plan_.coeff_elems, sample_cap, lowpass_kind_, stream_);
	ent_ready_ = rc == 0;
	if (ent_ready_ && h_jobs_) {
		DecJobs j = dec_jobs_at(h_jobs_, n_, plan_.num_channels);
		for (int i = 0; i < n_; i++) j.yuv[i].masks = ent_.block_masks(i);
		jobs_dirty_ = true;
	}
	return rc;
}

// The level-1 highpass bands as block lists between the entropy decoder's tile pass and k_inv_yuv422_strip_blocks (cfhd_core.h dec_block_list_layout): wherever the
// progressive 4:2:2 strip kernel writes 8-bit 4:2:2 pictures behind the GPU entropy stage.  CFHD_AMD_DEC_BLOCKS=0: dense bands (A/B runs).
bool DecodeBatch::block_lists_inverse() const
{
	const int blocks_env = [] { const char *e = getenv("CFHD_AMD_DEC_BLOCKS"); return e ? atoi(e) : 1; }();      // (read at every launch: tests switch within one process)
	if (!blocks_env || !ent_ready_ || !ent_.block_masks(0) || !ent_.chunk_indexed()) return false;
	if (half_ || dec_planes16(out_kind_) || rgb32_of_422_ || rgb16_of_422_ || rgb24_of_422_ || v210_ || byr4_) return false;
	if (!(out_kind_ == PIX_YUY2 || out_kind_ == PIX_2VUY) || plan_.encoded_format != ENC_YUV422) return false;
	return interlaced_ ? frame_inverse_strips() : strip_inverse();      // (interlaced: LH and HH as lists, the difference-coded HL dense)
}

int DecodeBatch::launch_entropy()
{
	ent_.set_block_lists(block_lists_inverse());
	return ent_.launch();
}

int DecodeBatch::sync_jobs()
{
	(void)hipSetDevice(device_);
	if (!jobs_dirty_) return 0;
	HIPCHK(hipMemcpyAsync(d_jobs_, h_jobs_, jobs_bytes_, hipMemcpyHostToDevice, (hipStream_t)stream_));
	jobs_dirty_ = false;
	return 0;
}

void DecodeBatch::clear_host_coeffs(int i) { memset(h_coeff_ + (size_t)i * plan_.final_elems, 0, (size_t)plan_.final_elems * 2); }

int DecodeBatch::upload_coeffs()
{
	(void)hipSetDevice(device_);
	ent_.dense_pyramid_uploaded();                        // (whatever form the last GPU entropy pass left the level-1 bands in: they are dense rows now)
	HIPCHK(hipMemcpy2DAsync(d_coeff_, (size_t)plan_.coeff_elems * 2, h_coeff_, (size_t)plan_.final_elems * 2,
	                        (size_t)plan_.final_elems * 2, n_, hipMemcpyHostToDevice, (hipStream_t)stream_));
	return 0;
}

int DecodeBatch::set_device_output(int i, void *d_out, int pitch)
{
	if (i < 0 || i >= n_) return -1;
	DecJobs j = dec_jobs_at(h_jobs_, n_, plan_.num_channels);
	if (half_ && is_packed16(out_kind_)) { j.halfp[i].out = (uint16_t *)d_out; j.halfp[i].out_pitch = pitch; jobs_dirty_ = true; return 0; }
	if (dec_planes16(out_kind_)) {
		for (int c = 0; c < plan_.num_channels; c++) {
			dev::InvPlaneJob &p = j.l1[(size_t)i * plan_.num_channels + c];
			p.out = dec_plane_out(d_out, out_kind_, c); p.out_pitch = dec_rgb8(out_kind_) ? pitch : (dec_rgb10(out_kind_) ? pitch / 4 : pitch / 2);
		}
		jobs_dirty_ = true;
		return 0;
	}
	if (half_) { j.half[i].out = (uint8_t *)d_out; j.half[i].out_pitch = pitch; jobs_dirty_ = true; return 0; }
	if (j.yuv[i].out != d_out || j.yuv[i].out_pitch != pitch) { j.yuv[i].out = (uint8_t *)d_out; j.yuv[i].out_pitch = pitch; jobs_dirty_ = true; }
	return 0;
}

// k_inv_yuv422_strip serves luma bands of whole 16-column blocks and writes 16-byte words; everything else (and
// CFHD_AMD_INVERSE=tile, for A/B runs) takes the LDS-tiled k_inv_yuv422.  Both produce the same bytes.
bool DecodeBatch::strip_inverse() const
{
	const int forced = shape_override("CFHD_AMD_INVERSE");
	const int bw = plan_.ch[0].band[0][0].width;
	const int act = active_ > 0 && active_ < n_ ? active_ : n_;
	if (forced == 1 || (forced == 0 && frames_1080p_equivalent(plan_, act) < 12.0)) return false;
	if (dec_planes16(out_kind_) || bw % 16) return false;
	DecJobs j = dec_jobs_at(h_jobs_, n_, plan_.num_channels);
	for (int i = 0; i < n_; i++) if (((uintptr_t)j.yuv[i].out & 15) || (j.yuv[i].out_pitch & 15)) return false;
	return true;
}

// k_inv_packed16_strip serves RG48 / b64a output of whole 8-pixel blocks whose rows are 16-byte aligned, from the launch size on at which the
// strip kernels pay; everything else takes the LDS-tiled k_inv_packed16.
bool DecodeBatch::strip_inverse_packed16() const
{
	const int forced = shape_override("CFHD_AMD_INVERSE");
	const int act = active_ > 0 && active_ < n_ ? active_ : n_;
	if (forced == 1 || (forced == 0 && frames_1080p_equivalent(plan_, act) < 12.0)) return false;
	if (!is_packed16(out_kind_) || half_ || plan_.ch[0].band[0][0].width % 4 || (out_kind_ == PIX_B64A && plan_.num_channels == 3) || byr4_) return false;      // (the strip kernel knows the RG48 and b64a pixels only)
	DecJobs j = dec_jobs_at(h_jobs_, n_, plan_.num_channels);
	const int onch = dec_out_channels(out_kind_, plan_);
	for (int i = 0; i < n_; i++) {
		const dev::InvPlaneJob &p = j.l1[(size_t)i * onch];
		const uintptr_t frame = (uintptr_t)((const uint16_t *)p.out - packed_word_of_channel(out_kind_, 0));
		if ((frame & 15) || ((p.out_pitch * 2) & 15) || (p.band_pitch & 3)) return false;
	}
	return true;
}

// k_inv_frame_yuv422_strip: the interlaced last level in the shape of k_inv_yuv422_strip, under the same conditions
bool DecodeBatch::frame_inverse_strips() const
{
	if (!interlaced_ || half_) return false;
	for (int c = 0; c < 3; c++) if (plan_.ch[c].band[0][0].pitch % 8) return false;
	return strip_inverse();
}

// k_inv_frame_yuv422_quad: four band columns per thread with 8-byte loads and 16-byte stores (CFHD_AMD_INVERSE=tile: the one-column kernel)
bool DecodeBatch::frame_inverse_quads() const
{
	const int forced = shape_override("CFHD_AMD_INVERSE");
	const BandDesc &b = plan_.ch[0].band[0][0];
	if (forced == 1 || b.width % 4 || b.width < 8) return false;
	for (int c = 0; c < 3; c++) if (plan_.ch[c].band[0][0].pitch % 4) return false;
	DecJobs j = dec_jobs_at(h_jobs_, n_, plan_.num_channels);
	for (int i = 0; i < n_; i++) if (((uintptr_t)j.yuv[i].out & 15) || (j.yuv[i].out_pitch & 15)) return false;
	return true;
}

const char *DecodeBatch::level_kernel(int level) const
{
	const int act = active_ > 0 && active_ < n_ ? active_ : n_;
	if (level > 0) return planes_as_strips(plan_, level, act) ? "k_inv_plane_strip" : "k_inv_plane";
	if (half_) return is_packed16(out_kind_) ? "k_half_packed16" : "k_half_yuv422";
	if (dec_rgb10(out_kind_)) return "k_inv_rgb10";
	if (dec_planes16(out_kind_)) return strip_inverse_packed16() ? "k_inv_packed16_strip" : "k_inv_packed16";
	if (interlaced_) return frame_inverse_strips() ? (block_lists_inverse() ? "k_inv_frame_yuv422_strip_blocks" : "k_inv_frame_yuv422_strip") : (frame_inverse_quads() ? "k_inv_frame_yuv422_quad" : "k_inv_frame_yuv422");
	return strip_inverse() ? (block_lists_inverse() ? "k_inv_yuv422_strip_blocks" : "k_inv_yuv422_strip") : "k_inv_yuv422";
}

int DecodeBatch::launch_inverse(uint32_t dither_seed)
{
	// (the level-1 bands of the last entropy pass are block lists: only the kernel that gathers them may run behind it)
	if (ent_ready_ && ent_.level1_as_block_lists() && !block_lists_inverse()) { g_err = "the level-1 bands are block lists but the inverse would read them as dense rows"; return -1; }
	(void)hipSetDevice(device_);
	const bool jobs_uploaded_now = jobs_dirty_;          // (the upload is queued on `st`: a second stream must not read the tables before it)
	int rc = sync_jobs();
	if (rc) return rc;
	hipStream_t st = (hipStream_t)stream_;
	const int nch = plan_.num_channels;
	const int act = active_ > 0 && active_ < n_ ? active_ : n_;      // frames 0 .. act-1 carry pyramids (set_active)
	DecJobs j = dec_jobs_at(d_jobs_, n_, nch);
	(void)hipGetLastError();
	timed_ = true;
	// The entropy decoder may have finished the bands of levels 3 and 2 ahead of the level-1 bands (its tile pass over those is still queued on `st`): then the
	// inverse transforms of levels 3 and 2 run on a second stream beside it, and level 1 waits for both.
	void *l23 = ent_ready_ ? ent_.levels23_event() : nullptr;
	inv_split_ = l23 && stream2_ && !jobs_uploaded_now;
	hipStream_t sl = inv_split_ ? (hipStream_t)stream2_ : st;      // the stream of levels 3 and 2
	if (inv_split_) { HIPCHK(hipStreamWaitEvent(sl, (hipEvent_t)l23, 0)); HIPCHK(hipEventRecord((hipEvent_t)ev2_[0], sl)); }
	HIPCHK(hipEventRecord((hipEvent_t)ev0_, st));
	for (int lv = 2; lv >= 1; lv--) {
		const BandDesc &b = plan_.ch[0].band[lv][0];
		const dev::InvPlaneJob *jobs = lv == 2 ? j.l3 : j.l2;
		if (planes_as_strips(plan_, lv, act)) {
			const int n = act;
			for_channel_runs(plan_, lv, [&](int c0, int nc, int glog, const BandDesc &cb, int nseg) {
				const int nstrips = (cb.height + dev::SRP - 1) / dev::SRP, per_wave = nseg > 1 ? 1 : 64 >> glog, waves = ((n * nc + per_wave - 1) / per_wave) * nstrips * nseg;
				dev::k_inv_plane_strip<<<(waves + 3) / 4, dev::NTHREADS, 0, sl>>>(jobs, n, nch, c0, nc, glog, nstrips, cb.width, cb.height, nseg);
			});
			HIPCHK(hipEventRecord((hipEvent_t)(inv_split_ ? ev2_[3 - lv] : evl_[2 - lv]), sl));
			continue;
		}
		dim3 grid((b.width + dev::ITW - 1) / dev::ITW, (b.height + dev::ITH - 1) / dev::ITH, act * nch);
		dev::k_inv_plane<<<grid, dev::NTHREADS, 0, sl>>>(jobs);
		HIPCHK(hipEventRecord((hipEvent_t)(inv_split_ ? ev2_[3 - lv] : evl_[2 - lv]), sl));
	}
	if (inv_split_) {                                   // level 1 behind the level-1 tiles (stream order) and behind levels 3 and 2 (this wait)
		HIPCHK(hipStreamWaitEvent(st, (hipEvent_t)ev2_[2], 0));
		HIPCHK(hipEventRecord((hipEvent_t)evl_[1], st));
	}
	if (interlaced_ && !half_ && dec_planes16(out_kind_)) return -1;
	if (half_ && plan_.encoded_format == ENC_YUV422 && (dec_rgb8(out_kind_) || out_kind_ == PIX_RG48 || out_kind_ == PIX_B64A)) {
		const BandDesc &b = plan_.ch[0].band[0][0];
		dev::k_half_rgb24<<<dim3((b.width / 2 + dev::NTHREADS - 1) / dev::NTHREADS, out_rows_, act), dev::NTHREADS, 0, st>>>(j.half);
	} else if (half_ && (dec_rgb8(out_kind_) || dec_rgb10(out_kind_) || (out_kind_ == PIX_B64A && nch == 3))) {
		const BandDesc &b = plan_.ch[0].band[0][0];
		dev::k_half_rgb<<<dim3((b.width + dev::NTHREADS - 1) / dev::NTHREADS, out_rows_, act), dev::NTHREADS, 0, st>>>(j.halfp, dither_seed);
	} else if (half_ && out_kind_ == PIX_YU64) {
		const BandDesc &b = plan_.ch[0].band[0][0];
		dev::k_half_yu64<<<dim3((b.width / 2 + dev::NTHREADS - 1) / dev::NTHREADS, out_rows_, act), dev::NTHREADS, 0, st>>>(j.half);
	} else if (half_ && is_packed16(out_kind_)) {
		const BandDesc &b = plan_.ch[0].band[0][0];
		dev::k_half_packed16<<<dim3((b.width / 8 + dev::NTHREADS - 1) / dev::NTHREADS, out_rows_, act), dev::NTHREADS, 0, st>>>(j.halfp);
	} else if (half_) {
		const BandDesc &b = plan_.ch[0].band[0][0];
		dev::k_half_yuv422<<<dim3((b.width / 8 + dev::NTHREADS - 1) / dev::NTHREADS, out_rows_, act), dev::NTHREADS, 0, st>>>(j.half);
	} else if (strip_inverse_packed16()) {
		const BandDesc &b = plan_.ch[0].band[0][0];
		const int nseg = (b.width / 4 + dev::PSTEP - 1) / dev::PSTEP, nstrips = (b.height + dev::QSR - 1) / dev::QSR, waves = act * nseg * nstrips;
		if (dec_out_channels(out_kind_, plan_) == 4) dev::k_inv_packed16_strip<4><<<(waves + 3) / 4, dev::NTHREADS, 0, st>>>(j.l1, act, nseg, nstrips);
		else dev::k_inv_packed16_strip<3><<<(waves + 3) / 4, dev::NTHREADS, 0, st>>>(j.l1, act, nseg, nstrips);
	} else if (rgb32_of_422_) {
		const BandDesc &b = plan_.ch[0].band[0][0];
		dim3 grid((b.width + dev::ITW - 1) / dev::ITW, (b.height + dev::ITH - 1) / dev::ITH, act);
		dev::k_inv_yuv422_rgb32<<<grid, dev::NTHREADS, 0, st>>>(j.yuv);
	} else if (dec_planes16(out_kind_)) {
		const BandDesc &b = plan_.ch[0].band[0][0];
		dim3 grid((b.width + dev::ITW - 1) / dev::ITW, (b.height + dev::ITH - 1) / dev::ITH, act);      // one workgroup per tile, all components
		if (dec_rgb10(out_kind_)) dev::k_inv_rgb10<<<grid, dev::NTHREADS, 0, st>>>(j.l1);
		else { const int onch = dec_out_channels(out_kind_, plan_); dev::k_inv_packed16<<<grid, dev::NTHREADS, 0, st>>>(j.l1, onch, dec_words_per_position(out_kind_, onch), dither_seed); }
	} else if (interlaced_) {                           // (half resolution was served above: the level-1 lowpass planes need no inverse frame transform)
		const BandDesc &b = plan_.ch[0].band[0][0];
		if (frame_inverse_strips()) {
			const int nseg = (b.width / dev::SBLK + dev::SSEG - 1) / dev::SSEG;
			if (ent_ready_ && ent_.level1_as_block_lists()) dev::k_inv_frame_yuv422_strip_blocks<<<dim3(nseg, (b.height + dev::SRI - 1) / dev::SRI, act), dev::NTHREADS, 0, st>>>(j.yuv, dither_seed);
			else dev::k_inv_frame_yuv422_strip<<<dim3(nseg, (b.height + dev::SRI - 1) / dev::SRI, act), dev::NTHREADS, 0, st>>>(j.yuv, dither_seed);
		} else if (frame_inverse_quads()) dev::k_inv_frame_yuv422_quad<<<dim3((b.width / 4 + dev::NTHREADS - 1) / dev::NTHREADS, b.height, act), dev::NTHREADS, 0, st>>>(j.yuv, dither_seed);
		else dev::k_inv_frame_yuv422<<<dim3((b.width / 2 + dev::NTHREADS - 1) / dev::NTHREADS, b.height, act), dev::NTHREADS, 0, st>>>(j.yuv, dither_seed);
	} else if (strip_inverse()) {
		const BandDesc &b = plan_.ch[0].band[0][0];
		const int nseg = (b.width / dev::SBLK + dev::SSEG - 1) / dev::SSEG;
		if (ent_ready_ && ent_.level1_as_block_lists()) dev::k_inv_yuv422_strip_blocks<<<dim3(nseg, (b.height + dev::SR - 1) / dev::SR, act), dev::NTHREADS, 0, st>>>(j.yuv, dither_seed);
		else dev::k_inv_yuv422_strip<<<dim3(nseg, (b.height + dev::SR - 1) / dev::SR, act), dev::NTHREADS, 0, st>>>(j.yuv, dither_seed);
	} else {
		const BandDesc &b = plan_.ch[0].band[0][0];
		dim3 grid((b.width + dev::ITW - 1) / dev::ITW, (b.height + dev::ITH - 1) / dev::ITH, act);
		dev::k_inv_yuv422<<<grid, dev::NTHREADS, 0, st>>>(j.yuv, dither_seed);
	}
	if (rgb24_of_422_) {
		const int pairs = plan_.width / 2;
		dev::k_yu64_to_rgb24<<<dim3((unsigned)((pairs + dev::NTHREADS - 1) / dev::NTHREADS), (unsigned)out_rows_, (unsigned)act), dev::NTHREADS, 0, st>>>(
			(const uint16_t *)d_tmp_, tmp_pitch_ / 2, tmp_frame_bytes_ / 2, d_out_, out_pitch_, frame_bytes_, pairs, out_rows_, plan_.color_matrix, dither_seed);
	}
	if (rgb16_of_422_) {
		const int pairs = plan_.width / 2;
		dev::k_yu64_to_rgb16<<<dim3((unsigned)((pairs + dev::NTHREADS - 1) / dev::NTHREADS), (unsigned)out_rows_, (unsigned)act), dev::NTHREADS, 0, st>>>(
			(const uint16_t *)d_tmp_, tmp_pitch_ / 2, tmp_frame_bytes_ / 2, (uint16_t *)d_out_, out_pitch_ / 2, frame_bytes_ / 2, pairs, plan_.color_matrix >= 2, rgb16_b64a_ ? 1 : 0);
	}
	if (byr4_) {
		const int quads = plan_.width;
		dev::k_bayer_to_byr4<<<dim3((unsigned)((quads + dev::NTHREADS - 1) / dev::NTHREADS), (unsigned)plan_.display_height, (unsigned)act), dev::NTHREADS, 0, st>>>(
			(const uint16_t *)d_tmp_, tmp_pitch_ / 2, tmp_frame_bytes_ / 2, (uint16_t *)d_out_, out_pitch_ / 2, frame_bytes_ / 2, quads, d_restore_);
	}
	if (v210_) {
		const int groups = (half_ ? plan_.width / 2 : plan_.width) / 6;
		dev::k_yu64_to_v210<<<dim3((unsigned)((groups + dev::NTHREADS - 1) / dev::NTHREADS), (unsigned)out_rows_, (unsigned)act), dev::NTHREADS, 0, st>>>(
			(const uint16_t *)d_tmp_, tmp_pitch_ / 2, tmp_frame_bytes_ / 2, (uint32_t *)d_out_, out_pitch_ / 4, frame_bytes_ / 4, groups);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord((hipEvent_t)ev1_, st));
	return 0;
}

int DecodeBatch::download_frame(int i, void *out, int pitch)
{
	(void)hipSetDevice(device_);
	if (!own_output_ || i < 0 || i >= n_) return -1;
	if (direct_.size() != (size_t)n_) direct_.assign((size_t)n_, 0);
	direct_[i] = 0;
	if (out && pitch >= out_pitch_ && host_buffer_is_registered(out, (size_t)pitch * (out_rows_ - 1) + out_pitch_)) {
		HIPCHK(hipMemcpy2DAsync(out, (size_t)pitch, d_out_ + frame_bytes_ * i, (size_t)out_pitch_, (size_t)out_pitch_, (size_t)out_rows_, hipMemcpyDeviceToHost, (hipStream_t)stream_));
		direct_[i] = 1;
		return 0;
	}
	// a plain buffer: staged through pinned memory; the few frames of a C ABI call come in pieces with an event behind each, so that finish_frame() can copy a piece
	// into the caller's buffer while the DMA of the next is still running (as EncodeBatch::upload_frame does on the way in)
	const int pieces = (n_ <= 8 && frame_bytes_ >= stage_piece_min_bytes()) ? (stage_pieces_ > kMaxOutPieces ? (int)kMaxOutPieces : stage_pieces_) : 1;
	if (out_pieces_.size() != (size_t)n_) out_pieces_.assign((size_t)n_, 0);
	out_pieces_[i] = 0;
	if (stage_pieces_ > 1) {                             // (a batch that stages in pieces puts an event behind every frame it stages, also behind one that is too small to cut: finish_frame() may run before wait())
		if (piece_ev_.size() != (size_t)n_ * kMaxOutPieces) {
			piece_ev_.assign((size_t)n_ * kMaxOutPieces, nullptr);
			for (void *&e : piece_ev_) { hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); e = ev; }
		}
		for (int k = 0; k < pieces; k++) {
			const int r0 = (int)((long long)out_rows_ * k / pieces), r1 = (int)((long long)out_rows_ * (k + 1) / pieces);
			const size_t off = (size_t)r0 * out_pitch_, bytes = (size_t)(r1 - r0) * out_pitch_;
			if (bytes) HIPCHK(hipMemcpyAsync(h_out_ + frame_bytes_ * i + off, d_out_ + frame_bytes_ * i + off, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream_));
			HIPCHK(hipEventRecord((hipEvent_t)piece_ev_[(size_t)i * kMaxOutPieces + k], (hipStream_t)stream_));
		}
		out_pieces_[i] = pieces;
		return 0;
	}
	HIPCHK(hipMemcpyAsync(h_out_ + frame_bytes_ * i, d_out_ + frame_bytes_ * i, frame_bytes_, hipMemcpyDeviceToHost, (hipStream_t)stream_));
	return 0;
}

// The whole batch to host memory, asynchronous on the batch's stream; behind wait() the caller runs finish_frame() for every frame (nothing to do for frames that went
// straight into a registered buffer).
int DecodeBatch::download_frames(void *out, size_t frame_stride, int pitch)
{
	(void)hipSetDevice(device_);
	if (!own_output_ || !out) return -1;
	if (pitch == out_pitch_ && frame_stride == frame_bytes_ && host_buffer_is_registered(out, frame_bytes_ * (size_t)n_)) {
		HIPCHK(hipMemcpyAsync(out, d_out_, frame_bytes_ * (size_t)n_, hipMemcpyDeviceToHost, (hipStream_t)stream_));
		direct_.assign((size_t)n_, 1);
		return 0;
	}
	if (n_ > 8 && !host_buffer_is_registered(out, 1)) {
		// plain memory, many frames: ONE copy into the batch's pinned memory; finish_frame() copies every frame out behind wait() (the caller runs them on several threads)
		HIPCHK(hipMemcpyAsync(h_out_, d_out_, frame_bytes_ * (size_t)n_, hipMemcpyDeviceToHost, (hipStream_t)stream_));
		direct_.assign((size_t)n_, 0);
		out_pieces_.assign((size_t)n_, 0);
		return 0;
	}
	for (int i = 0; i < n_; i++) { const int rc = download_frame(i, (uint8_t *)out + frame_stride * (size_t)i, pitch); if (rc) return rc; }
	return 0;
}

// Orders everything queued on this batch's stream from now on behind what the producer stream holds at this moment.
int DecodeBatch::after(void *producer_stream)
{
	(void)hipSetDevice(device_);
	if (!evdep_) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); evdep_ = e; }
	HIPCHK(hipEventRecord((hipEvent_t)evdep_, (hipStream_t)producer_stream));
	HIPCHK(hipStreamWaitEvent((hipStream_t)stream_, (hipEvent_t)evdep_, 0));
	return 0;
}

int DecodeBatch::wait()
{
	(void)hipSetDevice(device_);
	HIPCHK(hipStreamSynchronize((hipStream_t)stream_));
	float ms = 0;
	if (!timed_) return 0;
	timed_ = false;
	if (hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)ev1_) == hipSuccess) kernel_ms_ = ms;
	if (inv_split_) {                                   // levels 3 and 2 ran on the second stream: their own events
		if (hipEventElapsedTime(&ms, (hipEvent_t)ev2_[0], (hipEvent_t)ev2_[1]) == hipSuccess) level_ms_[2] = ms;
		if (hipEventElapsedTime(&ms, (hipEvent_t)ev2_[1], (hipEvent_t)ev2_[2]) == hipSuccess) level_ms_[1] = ms;
	} else {
		if (hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)evl_[0]) == hipSuccess) level_ms_[2] = ms;
		if (hipEventElapsedTime(&ms, (hipEvent_t)evl_[0], (hipEvent_t)evl_[1]) == hipSuccess) level_ms_[1] = ms;
	}
	if (hipEventElapsedTime(&ms, (hipEvent_t)evl_[1], (hipEvent_t)ev1_) == hipSuccess) level_ms_[0] = ms;
	return 0;
}

int DecodeBatch::finish_frame(int i, void *out, int pitch)
{
	if (!own_output_ || i < 0 || i >= n_) return -1;
	if (direct_.size() == (size_t)n_ && direct_[i]) return 0;            // already in the caller's buffer
	const uint8_t *src = h_out_ + frame_bytes_ * i;
	uint8_t *dst = (uint8_t *)out;
	const int pieces = out_pieces_.size() == (size_t)n_ ? out_pieces_[i] : 0;
	if (pieces >= 1) {
		// the frame came down in pieces (download_frame): every piece is copied out as soon as its DMA has finished, beside the DMA of the next one.  (May be
		// called before wait(): the events order it; after wait() they have all fired.)
		(void)hipSetDevice(device_);
		for (int k = 0; k < pieces; k++) {
			const int r0 = (int)((long long)out_rows_ * k / pieces), r1 = (int)((long long)out_rows_ * (k + 1) / pieces);
			HIPCHK(hipEventSynchronize((hipEvent_t)piece_ev_[(size_t)i * kMaxOutPieces + k]));
			if (pitch == out_pitch_) memcpy(dst + (size_t)r0 * out_pitch_, src + (size_t)r0 * out_pitch_, (size_t)(r1 - r0) * out_pitch_);
			else for (int r = r0; r < r1; r++) memcpy(dst + (ptrdiff_t)r * pitch, src + (size_t)r * out_pitch_, (size_t)out_pitch_);
		}
		return 0;
	}
	if (pitch == out_pitch_) memcpy(dst, src, frame_bytes_);
	else for (int r = 0; r < out_rows_; r++) memcpy(dst + (ptrdiff_t)r * pitch, src + (size_t)r * out_pitch_, (size_t)out_pitch_);
	return 0;
}

// =============================================================================================
// GopBatch
// =============================================================================================
namespace {
struct GopJobs {            // layout of a GopBatch's job table
	dev::FwdYuvJob *yuv;        // [2]      level 1 of the two frames
	dev::GopTemporalJob *temp;  // [3]      per channel
	dev::FwdPlaneJob *mid;      // [6]      w[3] (from the temporal highpass band) and w[4] (from the lowpass band), per channel
	dev::FwdPlaneJob *top;      // [3]      w[5]
	dev::InvPlaneJob *itop;     // [3]      w[5] -> lowpass band of w[4]
	dev::InvPlaneJob *imid;     // [6]      w[4] -> temporal lowpass, w[3] -> temporal highpass
	dev::InvYuvJob *iyuv;       // [2]
};
GopJobs gop_jobs_at(void *base)
{
	GopJobs j;
	j.yuv = (dev::FwdYuvJob *)base; j.temp = (dev::GopTemporalJob *)(j.yuv + 2); j.mid = (dev::FwdPlaneJob *)(j.temp + 3); j.top = j.mid + 6;
	j.itop = (dev::InvPlaneJob *)(j.top + 3); j.imid = j.itop + 3; j.iyuv = (dev::InvYuvJob *)(j.imid + 6);
	return j;
}
size_t gop_jobs_bytes() { return 2 * sizeof(dev::FwdYuvJob) + 3 * sizeof(dev::GopTemporalJob) + 9 * sizeof(dev::FwdPlaneJob) + 9 * sizeof(dev::InvPlaneJob) + 2 * sizeof(dev::InvYuvJob); }
}

GopBatch::GopBatch() {}
GopBatch::~GopBatch() { release(); }

void GopBatch::release()
{
	(void)hipSetDevice(device_);
	if (stream_) hipStreamSynchronize((hipStream_t)stream_);
	ent_ready_ = false;                              // (its buffers go with the object or with the next prepare_group(); it is not used before prepare_entropy() ran again)
	if (dec_ready_) { dec_.release(); dec_ready_ = false; }
	if (d_frames_) hipFree(d_frames_);
	if (h_frames_) hipHostFree(h_frames_);
	if (d_coeff_) hipFree(d_coeff_);
	if (h_coeff_) hipHostFree(h_coeff_);
	if (d_jobs_) hipFree(d_jobs_);
	if (h_jobs_) hipHostFree(h_jobs_);
	if (stream_) device_stream_destroy(stream_);
	d_frames_ = h_frames_ = nullptr; d_coeff_ = h_coeff_ = nullptr; d_jobs_ = h_jobs_ = nullptr; stream_ = nullptr;
}

int GopBatch::prepare(const GopPlan &plan, bool decode, int out_pixel_kind)
{
	int rc = device_init();
	if (rc) return rc;
	release();
	device_ = device_current(); (void)hipSetDevice(device_);
	plan_ = plan; decode_ = decode; out_kind_ = out_pixel_kind;
	HIPCHK((hipError_t)device_stream_create(&stream_));
	pitch_ = packed_frame_pitch(decode ? out_pixel_kind : plan.pixel_kind, plan.width); rows_ = plan.display_height;
	frame_bytes_ = (size_t)pitch_ * rows_;
	HIPCHK(hipMalloc((void **)&d_frames_, 2 * frame_bytes_));
	HIPCHK(hipHostMalloc((void **)&h_frames_, 2 * frame_bytes_, hipHostMallocPortable));
	HIPCHK(hipMalloc((void **)&d_coeff_, plan.coeff_elems * 2));
	HIPCHK(hipMemsetAsync(d_coeff_, 0, plan.coeff_elems * 2, (hipStream_t)stream_));      // pad columns stay zero forever
	HIPCHK(hipHostMalloc((void **)&h_coeff_, plan.coeff_elems * 2, hipHostMallocPortable));
	memset(h_coeff_, 0, plan.coeff_elems * 2);
	jobs_bytes_ = gop_jobs_bytes();
	HIPCHK(hipMalloc(&d_jobs_, jobs_bytes_));
	HIPCHK(hipHostMalloc(&h_jobs_, jobs_bytes_, hipHostMallocPortable));
	memset(h_jobs_, 0, jobs_bytes_);
	fill_jobs();
	return 0;
}

void GopBatch::set_plan(const GopPlan &plan)
{
	if (stream_) (void)hipStreamSynchronize((hipStream_t)stream_);
	plan_ = plan;
	if (ent_ready_) ent_.set_group_plan(plan);
	fill_jobs();
}

int GopBatch::prepare_entropy(size_t sample_cap)
{
	(void)hipSetDevice(device_);
	ent_ready_ = false;
	if (decode_ || !d_coeff_) return -1;
	const int rc = ent_.prepare_group(plan_, 1, d_coeff_, plan_.coeff_elems, sample_cap, stream_);
	ent_ready_ = rc == 0;
	return rc;
}

int GopBatch::launch_entropy_decode(const uint8_t *sample, size_t size, const ParsedGroup &pg, size_t sample_cap)
{
	(void)hipSetDevice(device_);
	if (!decode_ || !d_coeff_) return -1;
	if (!dec_ready_) {
		if (dec_.prepare(plan_, d_coeff_, sample_cap, out_kind_, stream_, device_)) return -1;
		dec_ready_ = true;
	}
	return dec_.launch(sample, size, pg);
}

void GopBatch::fill_jobs()
{
	const GopPlan &plan = plan_;
	GopJobs j = gop_jobs_at(h_jobs_);
	const int mpq = plan.midpoint_prequant;
	int16_t *base = d_coeff_;
	for (int f = 0; f < 2; f++) {
		dev::FwdYuvJob &y = j.yuv[f];
		y.in = d_frames_ + frame_bytes_ * f; y.in_pitch = pitch_;
		y.width = plan.width; y.height = plan.height; y.display_height = plan.display_height;
		y.uyvy = plan.pixel_kind == PIX_2VUY; y.shift = plan.precision - 8;
		dev::InvYuvJob &iy = j.iyuv[f];
		for (int c = 0; c < 3; c++) {
			const GopWavelet &w = plan.ch[c].w[f];
			y.out_pitch[c] = w.pitch; iy.band_pitch[c] = w.pitch;
			for (int b = 0; b < 4; b++) { y.out[c][b] = base + w.offset[b]; y.q[c][b] = make_q(w.quant[b], mpq); iy.band[c][b] = base + w.offset[b]; }
			// interlaced groups: the difference-coded band is quantized inside the horizontal filter, midpoint = divisor / prequant without the decrement (spatial.c:5360-5363)
			if (plan.interlaced && w.quant[2] > 1 && mpq >= 2 && mpq < 9) y.q[c][2].mid = w.quant[2] / mpq;
		}
		iy.width = plan.ch[0].w[f].width; iy.height = plan.ch[0].w[f].height; iy.display_height = plan.display_height;
		iy.uyvy = out_kind_ == PIX_2VUY; iy.shift = plan.precision - 8; iy.dither_seed = 0x9E3779B9u * (uint32_t)(f + 1);
		iy.out = d_frames_ + frame_bytes_ * f; iy.out_pitch = pitch_;
	}
	auto fwd = [&](dev::FwdPlaneJob &p, const int16_t *in, const GopWavelet &src, const GopWavelet &dst) {
		p.in = in; p.in_pitch = src.pitch; p.width = src.width; p.height = src.height; p.prescale = dst.prescale;
		p.xstride = 1; p.shift = 0; p.display_height = src.height; p.compand = 0; p.layout = 0; p.tail_from = 0;
		p.out_pitch = dst.pitch;
		for (int b = 0; b < 4; b++) { p.out[b] = base + dst.offset[b]; p.q[b] = make_q(dst.quant[b], mpq); }
	};
	auto inv = [&](dev::InvPlaneJob &p, const GopWavelet &src, int16_t *out, int out_pitch) {
		memset(&p, 0, sizeof(p));
		for (int b = 0; b < 4; b++) p.band[b] = base + src.offset[b];
		p.band_pitch = src.pitch; p.width = src.width; p.height = src.height; p.descale = src.prescale;
		p.out = out; p.out_pitch = out_pitch; p.xstride = 1; p.precision = 0; p.display_height = 2 * src.height;
		// the unprescaled wavelets of a group (w[5] and w[3]) go through the reference's InvertSpatialQuantOverflowProtected16s and inherit the defect of its last
		// row (InvPlaneJob::ll_bottom_row_high): the reference decoder's pictures are the parity bar.  CFHD_AMD_GOP_BOTTOM_ROWS=fixed: the filter as meant (+5 dB).
		static const bool fixed = [] { const char *e = getenv("CFHD_AMD_GOP_BOTTOM_ROWS"); return e && strcmp(e, "fixed") == 0; }();
		p.ll_bottom_row_high = (src.prescale == 0 && !fixed) ? 1 : 0;
	};
	for (int c = 0; c < 3; c++) {
		const GopChannel &ch = plan.ch[c];
		dev::GopTemporalJob &t = j.temp[c];
		t.pitch = ch.w[2].pitch; t.height = ch.w[2].height; t.width = ch.w[2].width;
		if (!decode_) { t.a = base + ch.w[0].offset[0]; t.b = base + ch.w[1].offset[0]; t.x = base + ch.w[2].offset[0]; t.y = base + ch.w[2].offset[1]; }
		else { t.a = base + ch.w[2].offset[0]; t.b = base + ch.w[2].offset[1]; t.x = base + ch.w[0].offset[0]; t.y = base + ch.w[1].offset[0]; }
		fwd(j.mid[2 * c], base + ch.w[2].offset[1], ch.w[2], ch.w[3]);      // the temporal highpass band
		fwd(j.mid[2 * c + 1], base + ch.w[2].offset[0], ch.w[2], ch.w[4]);  // the temporal lowpass band
		fwd(j.top[c], base + ch.w[4].offset[0], ch.w[4], ch.w[5]);
		inv(j.itop[c], ch.w[5], base + ch.w[4].offset[0], ch.w[4].pitch);
		inv(j.imid[2 * c], ch.w[4], base + ch.w[2].offset[0], ch.w[2].pitch);
		inv(j.imid[2 * c + 1], ch.w[3], base + ch.w[2].offset[1], ch.w[2].pitch);
	}
	jobs_dirty_ = true;
}

int GopBatch::upload_frame(int f, const void *frame, int pitch)
{
	(void)hipSetDevice(device_);
	if (decode_ || f < 0 || f > 1) return -1;
	const uint8_t *src = (const uint8_t *)frame;
	if (pitch < 0) { src += (ptrdiff_t)(rows_ - 1) * pitch; pitch = -pitch; }     // encoder.c:1957
	uint8_t *dst = h_frames_ + frame_bytes_ * f;
	if (pitch == pitch_) memcpy(dst, src, frame_bytes_);
	else for (int r = 0; r < rows_; r++) memcpy(dst + (size_t)r * pitch_, src + (size_t)r * pitch, (size_t)pitch_);
	HIPCHK(hipMemcpyAsync(d_frames_ + frame_bytes_ * f, dst, frame_bytes_, hipMemcpyHostToDevice, (hipStream_t)stream_));
	return 0;
}

int GopBatch::launch_forward()
{
	(void)hipSetDevice(device_);
	hipStream_t st = (hipStream_t)stream_;
	if (jobs_dirty_) { HIPCHK(hipMemcpyAsync(d_jobs_, h_jobs_, jobs_bytes_, hipMemcpyHostToDevice, st)); jobs_dirty_ = false; }
	GopJobs j = gop_jobs_at(d_jobs_);
	(void)hipGetLastError();
	// level 1 of both frames: the spatial transform, or -- interlaced groups -- the frame transform of interlaced intra frames (the two kernels share the job table)
	if (plan_.interlaced) dev::k_fwd_frame_yuv422<<<dim3((plan_.width / 2 + dev::FTW - 1) / dev::FTW, (plan_.height / 2 + dev::FRW - 1) / dev::FRW, 2), dev::NTHREADS, 0, st>>>((const dev::FwdFrameJob *)j.yuv);
	else dev::k_fwd_yuv422<<<dim3((plan_.width / 2 + dev::TW - 1) / dev::TW, (plan_.height / 2 + dev::TH - 1) / dev::TH, 2), dev::NTHREADS, 0, st>>>(j.yuv);
	const GopWavelet &t = plan_.ch[0].w[2];
	dev::k_gop_temporal_fwd<<<dim3((unsigned)((t.pitch * t.height / 2 + dev::NTHREADS - 1) / dev::NTHREADS), 3), dev::NTHREADS, 0, st>>>(j.temp);
	dev::k_fwd_plane<<<dim3((t.width / 2 + dev::TW - 1) / dev::TW, (t.height / 2 + dev::TH - 1) / dev::TH, 6), dev::NTHREADS, 0, st>>>(j.mid);
	const GopWavelet &m = plan_.ch[0].w[4];
	dev::k_fwd_plane<<<dim3((m.width / 2 + dev::TW - 1) / dev::TW, (m.height / 2 + dev::TH - 1) / dev::TH, 3), dev::NTHREADS, 0, st>>>(j.top);
	HIPCHK(hipGetLastError());
	return 0;
}

int GopBatch::download_coeffs()
{
	(void)hipSetDevice(device_);
	HIPCHK(hipMemcpyAsync(h_coeff_, d_coeff_, plan_.coeff_elems * 2, hipMemcpyDeviceToHost, (hipStream_t)stream_));
	return 0;
}

int GopBatch::launch_inverse(uint32_t dither_seed, bool coeffs_on_device)
{
	(void)hipSetDevice(device_);
	hipStream_t st = (hipStream_t)stream_;
	if (jobs_dirty_) { HIPCHK(hipMemcpyAsync(d_jobs_, h_jobs_, jobs_bytes_, hipMemcpyHostToDevice, st)); jobs_dirty_ = false; }
	if (!coeffs_on_device) HIPCHK(hipMemcpyAsync(d_coeff_, h_coeff_, plan_.coeff_elems * 2, hipMemcpyHostToDevice, st));
	GopJobs j = gop_jobs_at(d_jobs_);
	(void)hipGetLastError();
	const GopWavelet &top = plan_.ch[0].w[5], &mid = plan_.ch[0].w[4], &t = plan_.ch[0].w[2], &l1 = plan_.ch[0].w[0];
	dev::k_inv_plane<<<dim3((top.width + dev::ITW - 1) / dev::ITW, (top.height + dev::ITH - 1) / dev::ITH, 3), dev::NTHREADS, 0, st>>>(j.itop);
	dev::k_inv_plane<<<dim3((mid.width + dev::ITW - 1) / dev::ITW, (mid.height + dev::ITH - 1) / dev::ITH, 6), dev::NTHREADS, 0, st>>>(j.imid);
	dev::k_gop_temporal_inv<<<dim3((unsigned)((t.pitch * t.height / 2 + dev::NTHREADS - 1) / dev::NTHREADS), 3), dev::NTHREADS, 0, st>>>(j.temp);
	if (plan_.interlaced) dev::k_inv_frame_yuv422<<<dim3((l1.width / 2 + dev::NTHREADS - 1) / dev::NTHREADS, l1.height, 2), dev::NTHREADS, 0, st>>>(j.iyuv, dither_seed);
	else dev::k_inv_yuv422<<<dim3((l1.width + dev::ITW - 1) / dev::ITW, (l1.height + dev::ITH - 1) / dev::ITH, 2), dev::NTHREADS, 0, st>>>(j.iyuv, dither_seed);
	HIPCHK(hipGetLastError());
	return 0;
}

int GopBatch::download_frame(int f, void *, int)
{
	(void)hipSetDevice(device_);
	if (!decode_ || f < 0 || f > 1) return -1;
	HIPCHK(hipMemcpyAsync(h_frames_ + frame_bytes_ * f, d_frames_ + frame_bytes_ * f, frame_bytes_, hipMemcpyDeviceToHost, (hipStream_t)stream_));
	return 0;
}

int GopBatch::finish_frame(int f, void *out, int pitch)
{
	if (!decode_ || f < 0 || f > 1) return -1;
	const uint8_t *src = h_frames_ + frame_bytes_ * f;
	if (pitch == pitch_) memcpy(out, src, frame_bytes_);
	else for (int r = 0; r < rows_; r++) memcpy((uint8_t *)out + (ptrdiff_t)r * pitch, src + (size_t)r * pitch_, (size_t)pitch_);
	return 0;
}

int GopBatch::wait()
{
	(void)hipSetDevice(device_);
	HIPCHK(hipStreamSynchronize((hipStream_t)stream_));
	return 0;
}

} // namespace cfhd
