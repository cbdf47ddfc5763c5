// cfhd_device.h -- GPU side of the codec: batches of frames in flight (HIP stream + HBM buffers + job tables)
// and the launch sequences for encode (forward transform) and decode (inverse transform).
#pragma once
#include "cfhd_core.h"
#include "cfhd_bitstream.h"
#include "cfhd_entropy_gpu.h"
#include "cfhd_gop.h"
#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace cfhd {

// 0 on success; otherwise a hipError_t value.  The product has no CPU fallback: callers turn a failure into
// CFHD_ERROR_INTERNAL and the message is available from device_last_error().
int device_init();                 // picks the device from CFHD_AMD_DEVICE, else LOCAL_RANK, else 0
int device_count();
int device_select(int dev);        // this thread prepares its batches on device `dev` from now on (-1: the process default again); returns the device in effect, -1 on failure
int device_current();              // the device device_init() puts this thread on
// Passes that are queued as a whole (cfhd_amd_batch_submit) can take turns per stage on a device: the encode half of a pass starts behind the encode half of the pass queued
// before it on that device (the default for encode-only passes: cfhd_batch.cpp batch_launch says why) and, with CFHD_AMD_QUEUE=ordered, the decode half behind that pass's
// decode half.  stage 0: encode, 1: decode.  stage_order_wait: work queued on
// `stream` from now on waits for the last stage_order_done of that stage on `device`; stage_order_done: the work `stream` holds now is that stage's latest.
int stage_order_wait(int device, int stage, void *stream);
int stage_order_done(int device, int stage, void *stream);
// The HIP device the calling thread had current when it entered the library, put back when it leaves: a handle that was dealt another GPU (unit_device)
// must not leave the application's thread on that GPU -- its own HIP / PyTorch work would land there.  Every public entry point that can select a device holds one.
int device_caller_save();          // the caller's current device, -1 when there is none to restore
void device_caller_restore(int dev);
struct CallerDevice { int dev; CallerDevice() : dev(device_caller_save()) {} ~CallerDevice() { device_caller_restore(dev); } CallerDevice(const CallerDevice &) = delete; };
const char *device_last_error();

// N frames that travel through the forward path together: one launch per wavelet level covers every channel of
// every frame (blockIdx.z walks the job table).  N = 1 is the synchronous CFHD_EncodeSample path.
class EncodeBatch {
public:
	EncodeBatch();
	~EncodeBatch();
	// own_input: allocate HBM (and pinned host staging) for the packed frames; otherwise frames are supplied as device pointers.
	int prepare(const FramePlan &plan, int nframes, bool own_input);
	int nframes() const { return n_; }
	const FramePlan &plan() const { return plan_; }
	// Stage one host frame (any pitch, negative allowed as in Codec/encoder.c:1957) and start its H2D copy.
	int upload_frame(int i, const void *frame, int pitch_bytes);
	int upload_frames(const void *frames, size_t frame_stride, int pitch_bytes);      // all n frames, asynchronous on the batch's stream; ONE copy when the frames lie back to back in a registered buffer
	// Use frames that already live in HBM (bench / device-resident callers).
	int set_device_frame(int i, const void *d_frame, int pitch_bytes);
	// the next launches (forward transform, entropy coder, sample download) cover frames 0 .. k-1 only (0 = all)
	void set_active(int k) { active_ = k; ent_.set_active(k); }
	// async: all levels, all frames.  coeffs_needed = false: nothing but the GPU entropy stage will read this launch's coefficients -- where the level-1 bands
	// leave as block lists (k_fwd_yuv422_strip_blocks, cfhd_kernels.h FwdBlockLists) their dense rows are then not written at all.
	int launch_forward(bool coeffs_needed = true);
	int update_quant(const FramePlan &plan);           // same geometry, new quantizer tables (per-frame rate feedback)
	// GPU entropy stage (cfhd_entropy_kernels.h): complete samples are produced in HBM after launch_forward().
	int prepare_entropy(size_t sample_cap);
	GpuEntropyEncoder &entropy() { return ent_; }
	void set_stage_pieces(int k) { stage_pieces_ = k < 1 ? 1 : (k > 16 ? 16 : k); }      // plain input frames staged in k pieces (upload_frame)
	bool has_entropy() const { return ent_ready_; }
	bool strip_forward() const;                     // level 1 of 4:2:2 runs as k_fwd_yuv422_strip (else k_fwd_yuv422)
	bool block_lists_forward() const;               // ... and leaves the quantized level-1 bands as block lists for k_ent_count_blocks (k_fwd_yuv422_strip_blocks)
	bool strip_forward_packed16() const;
	bool strip_forward_bayer() const;
	bool strip_forward_frame() const;            // level 1 of RG48 / b64a runs as k_fwd_packed16_strip (else k_fwd_packed16)
	const char *level_kernel(int level) const;      // name of the kernel the next launch_forward() uses for level 0 / 1 / 2 (as a profiler shows it)
	int download_coeffs();                             // async: final (entropy coded) region of every frame -> pinned host
	int wait();
	const int16_t *host_coeffs(int i) const { return h_coeff_ + (size_t)i * plan_.final_elems; }
	int16_t *device_coeffs(int i) { return d_coeff_ + (size_t)i * plan_.coeff_elems; }
	void *stream() { return stream_; }
	int device() const { return device_; }
	float last_kernel_ms() const { return kernel_ms_; }   // forward kernels of the last launch (HIP events on this stream)
	float last_level_ms(int level) const { return level_ms_[level]; }   // level 0 = k_fwd_yuv422, 1/2 = k_fwd_plane launches
	void release();                                    // frees every device / pinned buffer of the batch (the destructor's work; prepare() starts with it)
private:
	int sync_jobs();
	void fill_jobs();
	void fill_block_lists();
	FramePlan plan_;
	int n_ = 0, active_ = 0, device_ = 0; bool own_input_ = false, jobs_dirty_ = true;
	void *stream_ = nullptr, *ev0_ = nullptr, *ev1_ = nullptr, *evl_[2] = {nullptr, nullptr};
	float level_ms_[3] = {0, 0, 0};
	uint8_t *d_in_ = nullptr, *h_in_ = nullptr; size_t frame_bytes_ = 0; int in_pitch_ = 0, in_rows_ = 0;
	int16_t *d_coeff_ = nullptr, *h_coeff_ = nullptr;
	void *d_jobs_ = nullptr, *h_jobs_ = nullptr; size_t jobs_bytes_ = 0;
	int16_t *d_planes_ = nullptr; uint16_t *d_curve_ = nullptr; size_t plane_elems_ = 0;   // Bayer input: component planes + encode curve LUT
	// Bayer: k_unpack_byr4 writes the component planes and k_fwd_plane transforms them (small launches; large ones take k_fwd_bayer_strip).  bayer_fused_ (tests only, no
	// switch since round 5): level 1 computes the planes in its loader (k_fwd_packed16 layout 10 / 11) and never writes them -- same bytes, but 5.1 ms instead of 2.2 for
	// 96 4K frames (round 3): every plane's loader pays the four curve gathers of a photosite quad again, and the tile's halo columns on top.
	bool bayer_fused_ = false;
	float kernel_ms_ = 0;
	bool timed_ = false;
	GpuEntropyEncoder ent_; bool ent_ready_ = false;
	int stage_pieces_ = 1;
};

class DecodeBatch {
public:
	DecodeBatch();
	~DecodeBatch();
	// half: CFHD_DECODED_RESOLUTION_HALF of 4:2:2 samples -- the last wavelet level is not run, the level-1 lowpass planes are the picture
	int prepare(const FramePlan &plan, int nframes, int out_pixel_kind, bool own_output, bool half = false);
	// interlaced 4:2:2 samples: the last level is the inverse frame transform (k_inv_frame_yuv422); 8-bit 4:2:2 output, full resolution
	void set_interlaced(bool on) { interlaced_ = on; ent_.set_interlaced(on); }
	bool interlaced() const { return interlaced_; }
	// the next launches (entropy decoder with host-parsed samples, inverse transform) cover frames 0 .. k-1 only (0 = all)
	void set_active(int k) { active_ = k; ent_.set_active(k); }
	int nframes() const { return n_; }
	const FramePlan &plan() const { return plan_; }
	int16_t *host_coeffs(int i) { return h_coeff_ + (size_t)i * plan_.final_elems; }   // host entropy decoder writes here
	int16_t *device_coeffs(int i) { return d_coeff_ + (size_t)i * plan_.coeff_elems; }
	void clear_host_coeffs(int i);
	int upload_coeffs();                               // async: pinned host -> HBM (final region of every frame)
	// GPU entropy decoder (k_dec_bands): samples in, dequantized pyramid built in HBM (replaces host_coeffs()/upload_coeffs()).
	int prepare_entropy(size_t sample_cap);
	GpuEntropyDecoder &entropy() { return ent_; }
	int launch_entropy();                            // entropy().launch() with the level-1 bands as block lists where the inverse gathers them (block_lists_inverse())
	bool has_entropy() const { return ent_ready_; }
	bool strip_inverse() const;                     // the last level of 4:2:2 runs as k_inv_yuv422_strip (else k_inv_yuv422)
	bool strip_inverse_packed16() const;            // the last level of RG48 / b64a output runs as k_inv_packed16_strip (else k_inv_packed16)
	bool frame_inverse_quads() const;
	bool frame_inverse_strips() const;
	bool block_lists_inverse() const;               // interlaced samples: k_inv_frame_yuv422_quad (else k_inv_frame_yuv422)
	const char *level_kernel(int level) const;      // name of the kernel the next launch_inverse() uses for level 0 / 1 / 2
	int set_device_output(int i, void *d_out, int pitch_bytes);
	int launch_inverse(uint32_t dither_seed);          // async
	int download_frame(int i, void *out, int pitch_bytes);   // async D2H into pinned staging, then row copy after wait
	int download_frames(void *out, size_t frame_stride, int pitch_bytes);      // all n frames (finish_frame() for each behind wait()); ONE copy into a registered buffer that takes them back to back
	int after(void *producer_stream);                  // async: later work on this batch's stream waits for what the producer stream holds now
	int wait();
	int finish_frame(int i, void *out, int pitch_bytes);      // after wait(): copy the staged frame to the caller's buffer
	void *stream() { return stream_; }
	float last_kernel_ms() const { return kernel_ms_; }
	float last_level_ms(int level) const { return level_ms_[level]; }   // level 0 = k_inv_yuv422, 1/2 = k_inv_plane launches (level index)
	void release();                                    // frees every device / pinned buffer of the batch (the destructor's work; prepare() starts with it)
private:
	int sync_jobs();
	FramePlan plan_;
	int n_ = 0, out_kind_ = 0, device_ = 0; bool own_output_ = false, jobs_dirty_ = true, half_ = false, interlaced_ = false; int active_ = 0;
	void *stream_ = nullptr, *ev0_ = nullptr, *ev1_ = nullptr, *evl_[2] = {nullptr, nullptr};
	float level_ms_[3] = {0, 0, 0};
	int16_t *d_coeff_ = nullptr, *h_coeff_ = nullptr;
	uint8_t *d_out_ = nullptr, *h_out_ = nullptr; size_t frame_bytes_ = 0; int out_pitch_ = 0, out_rows_ = 0;
	bool v210_ = false; uint8_t *d_tmp_ = nullptr; int tmp_pitch_ = 0; size_t tmp_frame_bytes_ = 0;      // v210 output: YU64 rows first, packed by k_yu64_to_v210
	// levels 3 and 2 of the inverse transform on a stream of their own when the entropy decoder finished their bands ahead of the level-1 bands (GpuEntropyDecoder::levels23_event)
	void *stream2_ = nullptr, *ev2_[3] = {nullptr, nullptr, nullptr}; bool inv_split_ = false;
	int lowpass_kind_ = 0;             // the output format as the lowpass bias rule sees it (lowpass_bias(): RG24 of a 4:2:2 sample is not YU64 there)
	bool byr4_ = false; uint16_t *d_restore_ = nullptr;   // BYR4 output of Bayer samples: the four planes as 16-bit words per quad first, turned into the mosaic by k_bayer_to_byr4 (linear-restore table in HBM)
	bool rgb16_of_422_ = false, rgb16_b64a_ = false;   // RG48 / b64a output of 4:2:2 samples: YU64 rows first, converted by k_yu64_to_rgb16 (bayer.c:11916 + RGB2YUV.c:1760)
	bool rgb32_of_422_ = false;        // BGRA / BGRa output of 4:2:2 samples: the last level as k_inv_yuv422_rgb32 (spatial.c:29577)
	bool rgb24_of_422_ = false;        // RG24 output of 4:2:2 samples: YU64 rows first, converted by k_yu64_to_rgb24 (the reference's route: 16-bit rows, then colour conversion)
	std::vector<char> direct_;                      // frame i went straight to the caller's (registered) buffer: finish_frame has nothing to copy
	enum { kMaxOutPieces = 8 };
	int stage_pieces_ = 1;
public:
	// plain host buffers staged in k pieces (download_frame / finish_frame: the CPU copy of a piece beside the DMA of the next); then finish_frame() may be called before wait()
	void set_stage_pieces(int k) { stage_pieces_ = k < 1 ? 1 : (k > kMaxOutPieces ? (int)kMaxOutPieces : k); }
	bool staged_in_pieces() const { return stage_pieces_ > 1; }
private:
	std::vector<void *> piece_ev_; std::vector<int> out_pieces_;     // frames staged in pieces (download_frame / finish_frame): an event behind every piece's DMA
	void *d_jobs_ = nullptr, *h_jobs_ = nullptr; size_t jobs_bytes_ = 0;
	float kernel_ms_ = 0;
	bool timed_ = false;
	void *evdep_ = nullptr;
	GpuEntropyDecoder ent_; bool ent_ready_ = false;
};

// One two-frame group on the GPU (cfhd_gop.h): the kernels of the intra path (level 1 of both frames, the plane transforms of the three spatial
// wavelets) around the temporal step, forward for the encoder and inverse for the decoder.  The run-length / VLC stage of a group stays on the
// host (write_group_sample / vlc_decode_band): the coefficient pyramid crosses PCIe, as BASELINE.json's north_star arranges the codec.
class GopBatch {
public:
	GopBatch();
	~GopBatch();
	int prepare(const GopPlan &plan, bool decode, int out_pixel_kind);
	const GopPlan &plan() const { return plan_; }
	void set_plan(const GopPlan &plan);              // same geometry, new quantizer tables
	// encoder
	int upload_frame(int f, const void *frame, int pitch_bytes);      // f = 0, 1: stage one frame of the pair and start its H2D copy
	int launch_forward();                            // async: level 1 of both frames, temporal step, three spatial transforms per channel
	int download_coeffs();                           // async: the group pyramid -> pinned host
	// GPU entropy stage for the group sample (GpuEntropyEncoder::prepare_group): entropy().set_frame_header(0, hdr), launch_forward(), entropy().launch(),
	// entropy().download(), wait() -> entropy().host_sample(0)
	int prepare_entropy(size_t sample_cap);
	bool has_entropy() const { return ent_ready_; }
	GpuEntropyEncoder &entropy() { return ent_; }
	const int16_t *host_coeffs() const { return h_coeff_; }
	// decoder
	int16_t *host_coeffs_rw() { return h_coeff_; }   // the host entropy decoder writes the dequantized bands here
	// GPU entropy decode of a parsed group sample into the pyramid in HBM (GpuGroupEntropyDecoder); < 0: not served / malformed, decode on the host instead
	int launch_entropy_decode(const uint8_t *sample, size_t size, const ParsedGroup &pg, size_t sample_cap);
	int entropy_decode_errors() { return dec_.check(); }                    // after wait()
	int launch_inverse(uint32_t dither_seed, bool coeffs_on_device = false);      // async: (pyramid H2D,) three inverse spatial transforms, temporal step, last level of both frames
	int download_frame(int f, void *out, int pitch_bytes);
	int finish_frame(int f, void *out, int pitch_bytes);
	int wait();
	void release();
private:
	void fill_jobs();
	GopPlan plan_; bool decode_ = false; int out_kind_ = 0, device_ = 0;
	void *stream_ = nullptr;
	uint8_t *d_frames_ = nullptr, *h_frames_ = nullptr; size_t frame_bytes_ = 0; int pitch_ = 0, rows_ = 0;
	int16_t *d_coeff_ = nullptr, *h_coeff_ = nullptr;
	void *d_jobs_ = nullptr, *h_jobs_ = nullptr; size_t jobs_bytes_ = 0; bool jobs_dirty_ = true;
	GpuEntropyEncoder ent_; bool ent_ready_ = false;
	GpuGroupEntropyDecoder dec_; bool dec_ready_ = false;
};

int packed_frame_pitch(int pixel_kind, int width);     // bytes per row of a tightly packed frame

// Host buffers the caller promised to keep alive (cfhd_amd_register_host_buffer): page-locked once, then frames and samples travel between
// them and HBM without the staging copy through the library's own pinned memory.  Everything else is staged.
int host_buffer_register(void *p, size_t bytes);       // 0, or a hipError_t
int host_buffer_unregister(void *p);
bool host_buffer_is_registered(const void *p, size_t bytes);


} // namespace cfhd
