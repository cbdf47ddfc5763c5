// cfhd_entropy_gpu.h -- host driver of the GPU entropy stage (cfhd_entropy_kernels.h): builds the per-frame sample
// templates, the band/segment job tables and runs the four launches on the batch's stream.
#pragma once
#include "cfhd_core.h"
#include "cfhd_bitstream.h"
#include "cfhd_gop.h"
#include <vector>

namespace cfhd {

// HIP streams of the device-side objects.  Every object creates its stream(s) here; inside a StreamScope (thread-local) all of them get ONE stream, which belongs to the
// scope's owner: the objects' own device_stream_destroy() leaves it alone, the owner ends its life with device_stream_release() after the objects are gone.  Why: the runtime
// maps a process's streams onto 4 hardware queues (GPU_MAX_HW_QUEUES) in the order they were created, and streams that share a queue wait for each other -- a batch whose
// pass lives on one stream takes one queue, so four passes in flight run beside each other without the application setting anything (cfhd_batch.cpp).
int device_stream_create(void **stream);
void device_stream_destroy(void *stream);
void device_stream_release(void *stream);
// Lean streams (thread-local, set by cfhd_batch.cpp while it prepares a batch of the frame queue): an object creates only the streams it will launch on.  Outside it -- the
// handles of the C ABI: pool workers, decoders -- every encoder and decoder also creates the second stream of round 5's arrangement although single frames never use it:
// measured, not explained (profiles/r06_j_*: five runs each on one box) -- with 8 pool workers + 8 decoder threads the round trip through the C ABI runs at 5.0 k fps with
// those idle streams in place and at 3.8-4.1 k without them; the runtime deals its 4 hardware queues to streams in creation order, and the many-thread case likes its
// active streams on FEW queues (round 5 found the same from the other side: 16 hardware queues cost that case 10-25 %).
void device_streams_lean(bool on);
bool device_streams_are_lean();
struct StreamScope { StreamScope(); explicit StreamScope(void *preset /* a stream an earlier scope created */); ~StreamScope(); void *stream() const; };


class GpuEntropyEncoder {
public:
	GpuEntropyEncoder();
	~GpuEntropyEncoder();
	// coeffs: device pyramid of frame 0; frames are coeff_stride elements apart.
	int prepare(const FramePlan &plan, int nframes, int16_t *d_coeffs, size_t coeff_stride_elems, size_t sample_cap, void *stream);
	int set_frame_header(int i, const SampleHeaderInfo &hdr);        // header fields / metadata of frame i's sample
	void set_plan(const FramePlan &plan) { plan_ = plan; }          // same geometry, new quantizer values: the headers written from now on carry them
	// Two-frame groups: unit i is a group sample (17 subbands per channel, two raw 16-bit bands), its pyramid laid out by GopPlan; set_frame_header() writes group headers.
	// A sample beyond gop_sample_may_zero_bands() is not valid (the reference zeroes bands there, encoder.c:8332): the caller writes it on the host.
	int prepare_group(const GopPlan &plan, int ngroups, int16_t *d_coeffs, size_t coeff_stride_elems, size_t sample_cap, void *stream);
	void set_group_plan(const GopPlan &plan) { gplan_ = plan; }
	int launch();                                                    // async: templates H2D + 4 kernels
	int download();                                                  // sizes + packed offsets -> sync -> one async copy of all sample bytes (wait on the stream afterwards)
	int download_queue();                                            // the two halves of download(): what can be queued behind launch() without the host ...
	int download_finish();                                           // ... and what needs the sizes on the host (synchronises the stream, queues the copy of the sample bytes)
	int fetch_sizes();                                               // sizes only (device-resident consumers); synchronises the stream
	const uint32_t *device_sizes() const { return d_sizes_; }
	const uint8_t *host_sample(int i) const { return h_samples_ + h_offsets_[i]; }   // after download() + stream wait
	uint8_t *device_sample(int i) { return d_samples_ + (size_t)i * cap_; }
	uint32_t sample_bytes(int i) const { return h_sizes_[i]; }
	// the next launch() / download() cover frames 0 .. k-1 of the batch (0 = all)
	void set_active(int k) { active_ = k; }
	// passes queued as a whole (the batch's frame queue): download_queue() sends the sample bytes on their way at once, sized by the previous pass
	void set_speculative_download(bool on) { speculative_download_ = on; if (!on) expect_bytes_ = 0; }
	// Interlaced frames and groups: the difference-coded bands are coded with peaks (encoder.c:4802) and their tables written on the device (k_ent_peaks).
	// needs_peak_table(i): a band of sample i has more peaks than the device's positions hold (2 million): that sample is not valid, the caller writes it on the host.
	bool needs_peak_table(int i) const { return peak_flags_in_use() && (h_sizes_[n_ + i] & 2u) != 0; }     // (the flags are only cleared and written for interlaced plans)
	bool has_peak_table(int i) const { return peak_flags_in_use() && (h_sizes_[n_ + i] & 1u) != 0; }       // statistics / tests: the sample carries a peak table written on the device
	bool peak_flags_in_use() const { return group_ ? gplan_.interlaced : plan_.interlaced; }
	size_t sample_cap() const { return cap_; }
	int total_segments() const { return total_segs_; }
	// HIP-event time of kernel k of the last launch() (0 k_ent_count -- when the level-1 bands are counted on the second stream: the launches on the main stream
	// only --, 1 k_ent_scan, 2 k_ent_layout, 3 k_ent_emit, 4 the level-1 part of k_ent_count on the second stream, 0 when there is none); valid once the stream was synchronised
	float kernel_ms(int k);
	// Events of the last launch() on the encoder's stream: every sample's header, size fields and raw lowpass bands are in place
	// (k_ent_layout done) / the samples are complete (k_ent_emit done).  A consumer on another stream can parse behind the first.
	// Optional: the event behind the level-1 transform of the frames about to be coded (on the stream given to prepare()).  With it launch() counts the level-1
	// bands on a stream of its own, beside the level-2 / level-3 transforms that are still queued in front of it on the main stream.
	void set_level1_event(void *ev) { ev_level1_ = ev; }
	// Block lists of the level-1 bands (cfhd_kernels.h FwdBlockLists): the buffers the forward strip kernel fills (null when the geometry has none), and whether the
	// next launch() counts the level-1 bands from them (k_ent_count_blocks) instead of reading the dense bands (k_ent_count).
	void *block_slots() const { return d_blocks_; }
	unsigned long long *block_masks(int frame) const { return d_masks_ ? d_masks_ + (size_t)frame * masks_per_frame_ : nullptr; }
	void set_block_lists(bool on) { use_blocks_ = on && d_blocks_; }
	bool block_lists() const { return use_blocks_; }
	void *headers_event() const { return ev_[3]; }
	void *samples_event() const { return ev_[4]; }
private:
	struct Host; Host *host_;           // host mirrors of the job tables (types live in the kernel headers)
	void release();
	int prepare_units(int n, int16_t *d_coeffs, size_t coeff_stride_elems, size_t sample_cap, void *stream);
	void build_template(const SampleHeaderInfo &hdr, SampleTemplate *t) const;
	bool group_ = false; GopPlan gplan_;
	FramePlan plan_; int n_ = 0, active_ = 0, device_ = 0 /* the GPU prepare() ran on */; size_t cap_ = 0; void *stream_ = nullptr;
	int active_frames() const { return active_ > 0 && active_ < n_ ? active_ : n_; }
	int nbands_ = 0, total_segs_ = 0;
	std::vector<SampleTemplate> tmpl_;
	uint8_t *d_samples_ = nullptr, *h_samples_ = nullptr;
	uint32_t *d_sizes_ = nullptr, *h_sizes_ = nullptr;
	uint8_t *d_packed_ = nullptr; uint32_t *d_offsets_ = nullptr, *h_offsets_ = nullptr;   // dense copy of the samples for the D2H transfer
	size_t expect_bytes_ = 0, copied_ahead_ = 0; int expect_frames_ = 0; bool speculative_download_ = false;      // download_queue(): the copy sized by the last pass (set_speculative_download)
	void *d_tables_ = nullptr, *d_bands_ = nullptr, *d_segband_ = nullptr, *d_segs_ = nullptr, *d_bandstate_ = nullptr, *d_frames_ = nullptr, *d_tokens_ = nullptr;
	uint8_t *d_tmpl_ = nullptr, *h_tmpl_ = nullptr;
	bool dirty_ = true;
	void *ev_[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; bool timed_ = false;
	bool split_ = false, serial_split_ = false; void *ev_level1_ = nullptr, *stream2_ = nullptr, *ev2_[3] = {nullptr, nullptr, nullptr};      // the level-1 part of k_ent_count on its own stream
	int16_t *d_coeffs_ = nullptr; size_t coeff_stride_ = 0;
	void *d_blocks_ = nullptr; unsigned long long *d_masks_ = nullptr; size_t masks_per_frame_ = 0; bool use_blocks_ = false;
};


// Decoder counterpart: parses each sample on the host (tag walk only), ships the sample bytes to HBM (unless they already live
// there) and lets one GPU lane per coded band rebuild the dequantized coefficient pyramid.
class GpuEntropyDecoder {
public:
	GpuEntropyDecoder();
	~GpuEntropyDecoder();
	// before the samples are set: interlaced samples (frame transform; one band per channel in code set 18, difference coded, maybe with a peak table)
	void set_interlaced(bool on) { interlaced_ = on; }
	// host-parsed samples only: the next launch() covers frames 0 .. k-1 of the batch (0 = all; a batch that gathers concurrent callers is rarely full)
	void set_active(int k) { active_ = k; }
	void set_skip_level1(bool skip) { skip_level1_ = skip; }            // before prepare(): host-parsed samples decode levels 2 and 3 only
	int prepare(const FramePlan &plan, int nframes, int16_t *d_coeffs, size_t coeff_stride_elems, size_t sample_cap, int out_pixel_kind, void *stream);
	// Sample bytes on the host: staged through pinned memory and copied to HBM by launch().
	int set_sample_host(int i, const uint8_t *sample, size_t size);
	// Sample bytes already in HBM at d_sample; host_copy (same bytes) is only parsed for the band offsets.
	int set_sample_device(int i, const uint8_t *d_sample, const uint8_t *host_copy, size_t size);
	// All n samples already in HBM (sample i at d_samples + i * stride_bytes, its size in d_sizes[i]): nothing touches the host,
	// k_dec_parse walks the tag streams on the GPU.  Stays in force until a set_sample_host()/set_sample_device() call.
	int set_samples_device(const uint8_t *d_samples, size_t stride_bytes, const uint32_t *d_sizes);
	// Optional, for device-resident samples still being written by another stream: the next launch() lets k_dec_parse wait for
	// `headers` only (it reads the tag stream, not the coded payloads) and the band decoder for `payloads`.
	void set_producer_events(void *headers, void *payloads) { ev_headers_ = headers; ev_payloads_ = payloads; }
	int launch();                        // async: (H2D samples, job tables | k_dec_parse), k_dec_bands_par + k_dec_lowpass
	int check();                         // after the stream was synchronised: 0 when every band decoded cleanly
	float kernel_ms(int k);              // last launch(): 0 k_dec_parse (device-resident samples only), 1 band decoder (all its kernels), 2 k_dec_lowpass, 3 k_dec_plan + k_dec_index, 4 k_dec_chain + k_dec_tile_index, 5 k_dec_tiles
	bool chunk_indexed() const { return dx_; }
	// Set by launch() when it decoded the bands of levels 2 and 3 (and the lowpass bands) first and recorded this event behind them: the inverse transforms of
	// those levels may start there, beside the tile pass over the level-1 bands that is still queued on the decoder's stream.  Null otherwise.
	void *levels23_event() const { return l23_split_ ? ev_l23_ : nullptr; }
	// Block lists of the level-1 bands (cfhd_core.h dec_block_list_layout): with set_block_lists(true) the next launch() of the chunk-indexed decoder leaves the level-1
	// highpass bands of a progressive 4:2:2 sample compacted in their own places in the pyramid + one occupancy mask per chunk of 64 blocks, for the inverse level-1 strip
	// kernel that gathers them; level1_as_block_lists(): what the last launch() did (the inverse has to match it)
	void set_block_lists(bool on) { use_blocks_ = on && d_masks_; }
	bool level1_as_block_lists() const { return blocks_written_; }
	void dense_pyramid_uploaded() { blocks_written_ = false; }          // the caller replaced the pyramid by dense coefficients from the host (host entropy decode)
	const unsigned long long *block_masks(int frame) const { return d_masks_ ? d_masks_ + (size_t)frame * masks_per_frame_ : nullptr; }
	int stats(uint32_t out[16]);         // CFHD_AMD_DX_STATS=1: convergence counters of the chunk index (see the .hip)
private:
	struct Host; Host *host_;
	void release();
	FramePlan plan_; int n_ = 0, out_kind_ = 0; size_t cap_ = 0, coeff_stride_ = 0; void *stream_ = nullptr;
	int16_t *d_coeffs_ = nullptr;
	uint8_t *d_samples_ = nullptr, *h_samples_ = nullptr;
	void *d_tables_ = nullptr, *d_bandjobs_ = nullptr, *d_lowjobs_ = nullptr, *d_plan_ = nullptr;
	void *ev_headers_ = nullptr, *ev_payloads_ = nullptr;
	bool skip_level1_ = false, interlaced_ = false; void *d_diffjobs_ = nullptr;
	int active_ = 0;
	int active_frames() const { return active_ > 0 && active_ < n_ ? active_ : n_; }
	const uint8_t *ext_samples_ = nullptr; size_t ext_stride_ = 0; const uint32_t *ext_sizes_ = nullptr;   // set_samples_device()
	int *d_errors_ = nullptr, *h_errors_ = nullptr;
	bool lane_kernel_ = false;
	// cfhd_dec_kernels.h (default): chunk index + tile decode.  CFHD_AMD_DEC=par / lane select the round-1 kernels for A/B runs.
	bool dx_ = true;
	void *d_tile_start_ = nullptr, *d_stats_ = nullptr, *d_repair_ = nullptr, *d_alts_ = nullptr, *d_reindex_ = nullptr, *d_alt_entries_ = nullptr; uint32_t alt_slots_ = 0;
	void *d_idx_tables_ = nullptr, *d_entries_ = nullptr, *d_recs_ = nullptr, *d_chunk_base_ = nullptr, *d_chunk_job_ = nullptr, *d_sums_ = nullptr, *d_counters_ = nullptr;
	uint32_t max_chunks_ = 0, *h_counters_ = nullptr; void *h_chunk_job_ = nullptr;
	int grid_index_ = 0, grid_tiles_ = 0;
	int device_ = 0;                       // the GPU prepare() ran on: every launch selects it for the calling thread
	int launch_dx(bool device_jobs, int njobs, uint32_t host_chunks, int lowpass_jobs);
	void *ev_l23_ = nullptr, *ev_low_ = nullptr; bool l23_split_ = false;      // ev_low_: in front of k_dec_lowpass when it runs between the two tile passes      // recorded behind the tiles of the level-2 / level-3 bands and the lowpass bands when the tile pass is split
	void *ev_[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; bool timed_ = false;    // [4]: end of k_dec_parse when the band decoder waits for a second event behind it; [5], [6]: behind k_dec_index / k_dec_chain; [7]: behind k_dec_plan (in front of k_dec_index)
	bool parse_end_ = false;
	unsigned long long *d_masks_ = nullptr; size_t masks_per_frame_ = 0; bool use_blocks_ = false, blocks_written_ = false;
};

// Group samples (cfhd_gop.h): parsed on the host (parse_group_sample), every coded band of the 17 subbands per channel decoded by one workgroup of
// k_dec_bands_par_ll (the flat job list of the round-1 decoder: a single group is a latency problem, not a throughput one), the two raw 16-bit bands of a channel
// by k_dec_lowpass.  Interlaced groups: the two difference-coded bands of a channel through the same kernel with the tables of code set 18, then k_dec_undiff
// (peak values, running sums along the rows: decoder.c:19809, :20822).  The dequantized pyramid is left in HBM for GopBatch::launch_inverse.
class GpuGroupEntropyDecoder {
public:
	GpuGroupEntropyDecoder() {}
	~GpuGroupEntropyDecoder() { release(); }
	int prepare(const GopPlan &plan, int16_t *d_coeffs, size_t sample_cap, int out_pixel_kind, void *stream, int device = -1);
	// async: sample H2D, job tables, the two kernels.  < 0: the sample does not fit the plan (nothing was launched)
	int launch(const uint8_t *sample, size_t size, const ParsedGroup &pg);
	int check();                         // after the stream was synchronised: 0 when every band decoded cleanly
	void release();
private:
	GopPlan plan_; int out_kind_ = 0, device_ = 0; size_t cap_ = 0; void *stream_ = nullptr;
	int16_t *d_coeffs_ = nullptr;
	uint8_t *d_sample_ = nullptr, *h_sample_ = nullptr;
	void *d_tables_ = nullptr, *d_bandjobs_ = nullptr, *d_lowjobs_ = nullptr, *h_bandjobs_ = nullptr, *h_lowjobs_ = nullptr;
	void *d_tables18_ = nullptr, *d_diffjobs_ = nullptr, *h_diffjobs_ = nullptr;      // interlaced groups: code set 18, the difference-coded bands (subbands 12 and 15 of every channel)
	int *d_errors_ = nullptr, *h_errors_ = nullptr;
};

} // namespace cfhd
