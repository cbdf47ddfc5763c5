// cfhd_entropy_gpu.hip -- see cfhd_entropy_gpu.h
#include "cfhd_entropy_gpu.h"
#include "cfhd_entropy_jobs.h"
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

namespace cfhd {

int device_init();
int device_current();
namespace { int g_fail(hipError_t e, const char *what) { fprintf(stderr, "[cfhd_amd] %s: %s\n", what, hipGetErrorString(e)); return (int)e ? (int)e : -1; } }
#define HIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return g_fail(_e, #expr); } while (0)

struct GpuEntropyEncoder::Host { EntHostJobs jobs; std::vector<EntHoleGeom> geom; dev::EntFrameJob *frames = nullptr; /* pinned: async copies must not stall the host */ };

GpuEntropyEncoder::GpuEntropyEncoder() : host_(new Host) {}
GpuEntropyEncoder::~GpuEntropyEncoder() { release(); delete host_; }

void GpuEntropyEncoder::release()
{
	(void)hipSetDevice(device_);
	void *dev[] = { d_samples_, d_sizes_, d_tables_, d_bands_, d_segband_, d_segs_, d_bandstate_, d_frames_, d_tmpl_, d_packed_, d_offsets_, d_tokens_, d_blocks_, d_masks_ };
	for (void *p : dev) if (p) (void)hipFree(p);
	if (h_samples_) (void)hipHostFree(h_samples_);
	if (h_sizes_) (void)hipHostFree(h_sizes_);
	if (h_offsets_) (void)hipHostFree(h_offsets_);
	d_packed_ = nullptr; d_offsets_ = h_offsets_ = nullptr;
	if (h_tmpl_) (void)hipHostFree(h_tmpl_);
	if (host_->frames) { (void)hipHostFree(host_->frames); host_->frames = nullptr; }
	for (void *&e : ev_) if (e) { (void)hipEventDestroy((hipEvent_t)e); e = nullptr; }
	for (void *&e : ev2_) if (e) { (void)hipEventDestroy((hipEvent_t)e); e = nullptr; }
	if (stream2_) { device_stream_destroy(stream2_); stream2_ = nullptr; }
	timed_ = false;
	d_samples_ = h_samples_ = nullptr; d_sizes_ = h_sizes_ = nullptr; d_tables_ = d_bands_ = d_segband_ = d_segs_ = d_bandstate_ = d_frames_ = d_tokens_ = nullptr;
	d_tmpl_ = h_tmpl_ = nullptr; n_ = 0;
	d_blocks_ = nullptr; d_masks_ = nullptr; masks_per_frame_ = 0; use_blocks_ = false;
}

int GpuEntropyEncoder::prepare(const FramePlan &plan, int nframes, int16_t *d_coeffs, size_t stride, size_t sample_cap, void *stream)
{
	group_ = false; plan_ = plan;
	return prepare_units(nframes, d_coeffs, stride, sample_cap, stream);
}

// Two-frame groups (cfhd_gop.h): one group per unit, 17 subbands per channel -- the same kernels over the group's template and band table.
int GpuEntropyEncoder::prepare_group(const GopPlan &plan, int ngroups, int16_t *d_coeffs, size_t stride, size_t sample_cap, void *stream)
{
	group_ = true; gplan_ = plan; plan_ = FramePlan(); plan_.interlaced = false;
	return prepare_units(ngroups, d_coeffs, stride, sample_cap, stream);
}

void GpuEntropyEncoder::build_template(const SampleHeaderInfo &hdr, SampleTemplate *t) const
{
	if (group_) build_group_template(gplan_, hdr, t); else build_sample_template(plan_, hdr, t);
}

int GpuEntropyEncoder::prepare_units(int nframes, int16_t *d_coeffs, size_t stride, size_t sample_cap, void *stream)
{
	int rc = device_init();
	if (rc) return rc;
	release();
	device_ = device_current(); (void)hipSetDevice(device_);
	const FramePlan &plan = plan_;
	n_ = nframes; cap_ = (sample_cap + 255) & ~(size_t)255; stream_ = stream; d_coeffs_ = d_coeffs; coeff_stride_ = stride;
	if (cap_ * (size_t)n_ >= ((size_t)1 << 32)) { fprintf(stderr, "[cfhd_amd] batch of %d frames exceeds the 4 GiB sample arena\n", n_); return -5; }   // packed offsets are 32-bit
	{
		std::vector<dev::EntTables> h(2);
		ent_build_tables(&h[0], 1); ent_build_tables(&h[1], 2);
		HIPCHK(hipMalloc(&d_tables_, 2 * sizeof(dev::EntTables)));
		HIPCHK(hipMemcpy(d_tables_, h.data(), 2 * sizeof(dev::EntTables), hipMemcpyHostToDevice));
	}
	SampleHeaderInfo hdr0 = { 1, 2, 2, 4, !plan.interlaced, nullptr, 0, nullptr, 0 };
	tmpl_.assign(n_, SampleTemplate());
	build_template(hdr0, &tmpl_[0]);
	EntHostJobs &jobs = host_->jobs;
	host_->geom = group_ ? ent_hole_geometry(gplan_, tmpl_[0]) : ent_hole_geometry(plan, tmpl_[0]);
	if (!ent_build_band_jobs(host_->geom, tmpl_[0], n_, d_coeffs, stride, &jobs)) return -3;
	nbands_ = jobs.nbands; total_segs_ = (int)jobs.segjobs.size();
	HIPCHK(hipMalloc(&d_bands_, jobs.bands.size() * sizeof(dev::EntBandJob)));
	HIPCHK(hipMemcpy(d_bands_, jobs.bands.data(), jobs.bands.size() * sizeof(dev::EntBandJob), hipMemcpyHostToDevice));
	HIPCHK(hipMalloc(&d_segband_, jobs.segjobs.size() * sizeof(dev::EntSegJob)));
	HIPCHK(hipMemcpy(d_segband_, jobs.segjobs.data(), jobs.segjobs.size() * sizeof(dev::EntSegJob), hipMemcpyHostToDevice));
	HIPCHK(hipMalloc(&d_segs_, jobs.segjobs.size() * sizeof(dev::EntSegState)));
	HIPCHK(hipMalloc(&d_tokens_, jobs.segjobs.size() * (size_t)dev::ENT_TOK_STRIDE * sizeof(uint32_t)));      // token lists and finished bit strings: worst case one per coefficient, only the used part is ever touched
	HIPCHK(hipMalloc(&d_bandstate_, jobs.bands.size() * sizeof(dev::EntBandState)));
	// block lists of the level-1 bands, for the geometries k_fwd_yuv422_strip_blocks serves (EncodeBatch::strip_forward): one 16-byte slot per block of the
	// pyramid (only the slots of listed blocks are ever touched), one mask per chunk
	static_assert((int)kBlockChunkCols == (int)dev::FWD_CHUNK_COLS_ENT, "one chunk geometry");
	if (!group_ && !plan.interlaced && plan.encoded_format == ENC_YUV422 && (plan.pixel_kind == PIX_YUY2 || plan.pixel_kind == PIX_2VUY) && plan.width % 32 == 0) {
		int mask_base[kMaxChannels][kNumBands];
		masks_per_frame_ = (size_t)block_list_layout(plan, mask_base);
		HIPCHK(hipMalloc(&d_blocks_, stride * 2 * (size_t)n_));
		HIPCHK(hipMalloc((void **)&d_masks_, masks_per_frame_ * 8 * (size_t)n_ + 64));      // (+ spare entries: the inverse kernel fetches the masks of two chunks at a time)
		HIPCHK(hipMemset(d_masks_, 0, masks_per_frame_ * 8 * (size_t)n_ + 64));
	}
	HIPCHK(hipMalloc((void **)&d_samples_, cap_ * n_));
	HIPCHK(hipHostMalloc((void **)&h_samples_, cap_ * n_, hipHostMallocPortable));
	HIPCHK(hipMalloc((void **)&d_sizes_, sizeof(uint32_t) * 2 * n_));                   // [n] sample sizes, [n] peak flags
	HIPCHK(hipHostMalloc((void **)&h_sizes_, sizeof(uint32_t) * 2 * n_, hipHostMallocPortable));
	memset(h_sizes_, 0, sizeof(uint32_t) * 2 * n_);
	HIPCHK(hipMalloc((void **)&d_packed_, cap_ * n_));
	HIPCHK(hipMalloc((void **)&d_offsets_, sizeof(uint32_t) * (n_ + 1)));
	HIPCHK(hipHostMalloc((void **)&h_offsets_, sizeof(uint32_t) * (n_ + 1), hipHostMallocPortable));
	memset(h_offsets_, 0, sizeof(uint32_t) * (n_ + 1));
	HIPCHK(hipMalloc((void **)&d_tmpl_, (size_t)kEntTmplStride * n_));
	HIPCHK(hipHostMalloc((void **)&h_tmpl_, (size_t)kEntTmplStride * n_, hipHostMallocPortable));
	memset(h_tmpl_, 0, (size_t)kEntTmplStride * n_);
	HIPCHK(hipMalloc(&d_frames_, n_ * sizeof(dev::EntFrameJob)));
	HIPCHK(hipHostMalloc((void **)&host_->frames, n_ * sizeof(dev::EntFrameJob), hipHostMallocPortable));
	for (int f = 0; f < n_; f++) { SampleHeaderInfo h = hdr0; h.frame_number = (uint32_t)f + 1; if ((rc = set_frame_header(f, h))) return rc; }
	for (void *&e : ev_) HIPCHK(hipEventCreate((hipEvent_t *)&e));
	for (void *&e : ev2_) HIPCHK(hipEventCreate((hipEvent_t *)&e));
	// (the second stream: used with eight frames or more -- the level-1 count beside the level-2 / 3 transforms --; the C ABI's handles create it nevertheless:
	// cfhd_entropy_gpu.h device_streams_lean)
	if (n_ >= 8 || !device_streams_are_lean()) HIPCHK((hipError_t)device_stream_create(&stream2_));
	return 0;
}

int GpuEntropyEncoder::set_frame_header(int f, const SampleHeaderInfo &hdr)
{
	if (f < 0 || f >= n_) return -1;
	SampleTemplate &t = tmpl_[f];
	build_template(hdr, &t);
	if (!ent_fill_frame_block(host_->geom, t, f, host_->jobs, d_coeffs_ + (size_t)f * coeff_stride_, h_tmpl_ + (size_t)kEntTmplStride * f)) return -4;
	host_->frames[f] = ent_frame_job(t, d_tmpl_ + (size_t)kEntTmplStride * f, d_samples_ + cap_ * f, (uint32_t)cap_, d_sizes_ + f, d_sizes_ + n_ + f);
	dirty_ = true;
	return 0;
}

int GpuEntropyEncoder::launch()
{
	(void)hipSetDevice(device_);
	hipStream_t st = (hipStream_t)stream_;
	if (dirty_) {
		HIPCHK(hipMemcpyAsync(d_tmpl_, h_tmpl_, (size_t)kEntTmplStride * n_, hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(d_frames_, host_->frames, n_ * sizeof(dev::EntFrameJob), hipMemcpyHostToDevice, st));
		dirty_ = false;
	}
	const dev::EntTables *T = (const dev::EntTables *)d_tables_;
	const dev::EntBatchGeom geom = { total_segs_ / n_, nbands_, coeff_stride_ };
	const int act = active_frames(), total_segs = total_segs_ / n_ * act;      // frames 0 .. act-1 (set_active)
	(void)hipGetLastError();
	HIPCHK(hipEventRecord((hipEvent_t)ev_[0], st));
#ifdef CFHD_AMD_PROBES
	const int count_probe = []{ const char *e = getenv("CFHD_AMD_COUNT_PROBE"); return e ? atoi(e) : 0; }();
#else
	const int count_probe = 0;
#endif
	const dev::EntBlockLists lists = { (const uint4 *)d_blocks_, d_masks_, d_coeffs_, masks_per_frame_ };
	auto count_range = [&](hipStream_t s, int lo, int n, bool level1 = false) {
		const int total = n * act;
		if (level1 && use_blocks_)
			dev::k_ent_count_blocks<<<(total + dev::ENT_WAVES - 1) / dev::ENT_WAVES, dev::ENT_THREADS, 0, s>>>((const dev::EntSegJob *)d_segband_, geom, total, (dev::EntSegState *)d_segs_, T,
			                                                                                                 d_sizes_ + n_, (uint32_t *)d_tokens_, lo, n, lists);
		else
			dev::k_ent_count<<<(total + dev::ENT_WAVES - 1) / dev::ENT_WAVES, dev::ENT_THREADS, 0, s>>>((const dev::EntSegJob *)d_segband_, geom, total, (dev::EntSegState *)d_segs_, T,
			                                                                                          d_sizes_ + n_, (uint32_t *)d_tokens_, lo, n, count_probe);
	};
	// CFHD_AMD_COUNT_SPLIT=0: (A/B) the level-1 bands counted on the main stream behind the level-2 / level-3 transforms instead of beside them
	static const bool split_on = [] { const char *e = getenv("CFHD_AMD_COUNT_SPLIT"); return !(e && atoi(e) == 0); }();
	split_ = split_on && ev_level1_ && stream2_ && !host_->jobs.ranges_l1.empty() && act >= 8;      // (a single frame gains nothing from six launches instead of one)
	// the peak flags are raised by the difference-coded band only, a level-1 band: cleared on the stream that counts it
	if (peak_flags_in_use() && !split_) HIPCHK(hipMemsetAsync(d_sizes_ + n_, 0, sizeof(uint32_t) * n_, st));
	serial_split_ = split_ && stream2_ == stream_;      // (a pass whose streams are one, cfhd_batch.cpp StreamScope: the same launches in a row, timed apart)
	if (serial_split_) {
		if (peak_flags_in_use()) HIPCHK(hipMemsetAsync(d_sizes_ + n_, 0, sizeof(uint32_t) * n_, st));
		HIPCHK(hipEventRecord((hipEvent_t)ev2_[0], st));
		for (const auto &r : host_->jobs.ranges_l1) count_range(st, r.first, r.second, true);
		HIPCHK(hipEventRecord((hipEvent_t)ev2_[1], st));
		for (const auto &r : host_->jobs.ranges_rest) count_range(st, r.first, r.second);
		HIPCHK(hipEventRecord((hipEvent_t)ev2_[2], st));
	} else if (split_) {
		// the level-1 bands on the second stream as soon as the level-1 transform is done; the bands of levels 2 and 3 here, behind their transforms; the scan waits for both
		hipStream_t s2 = (hipStream_t)stream2_;
		HIPCHK(hipStreamWaitEvent(s2, (hipEvent_t)ev_level1_, 0));
		if (peak_flags_in_use()) HIPCHK(hipMemsetAsync(d_sizes_ + n_, 0, sizeof(uint32_t) * n_, s2));
		HIPCHK(hipEventRecord((hipEvent_t)ev2_[0], s2));
		for (const auto &r : host_->jobs.ranges_l1) count_range(s2, r.first, r.second, true);
		HIPCHK(hipEventRecord((hipEvent_t)ev2_[1], s2));
		for (const auto &r : host_->jobs.ranges_rest) count_range(st, r.first, r.second);
		HIPCHK(hipEventRecord((hipEvent_t)ev2_[2], st));
		HIPCHK(hipStreamWaitEvent(st, (hipEvent_t)ev2_[1], 0));
	} else if (use_blocks_) {
		for (const auto &r : host_->jobs.ranges_l1) count_range(st, r.first, r.second, true);
		for (const auto &r : host_->jobs.ranges_rest) count_range(st, r.first, r.second);
	} else count_range(st, 0, total_segs_ / n_);
	HIPCHK(hipEventRecord((hipEvent_t)ev_[1], st));
	dev::k_ent_scan<<<nbands_ * act, dev::ENT_THREADS, 0, st>>>((const dev::EntBandJob *)d_bands_, (dev::EntSegState *)d_segs_, (dev::EntBandState *)d_bandstate_, T);
	HIPCHK(hipEventRecord((hipEvent_t)ev_[2], st));
	// workgroups per frame: enough to fill the chip for short batches of large frames, at least the 8 that 512 1080p frames were tuned with
	const unsigned layout_parts = act >= 256 ? 8u : (unsigned)((2048 + act - 1) / act > 256 ? 256 : (2048 + act - 1) / act);
	dev::k_ent_layout<<<dim3((unsigned)act, layout_parts), dev::ENT_THREADS, 0, st>>>((const dev::EntFrameJob *)d_frames_, (const dev::EntBandJob *)d_bands_, (dev::EntSegState *)d_segs_,
	                                                  (dev::EntBandState *)d_bandstate_, T);
	HIPCHK(hipEventRecord((hipEvent_t)ev_[3], st));
	// the values of the peak tables (interlaced plans: the difference-coded bands; a wave looks at a segment's record and leaves unless the segment has peaks)
	{
		dev::EntPeakHoles which; which.n = 0;
		for (size_t h = 0; h < tmpl_[0].holes.size() && which.n < 7; h++) if (tmpl_[0].holes[h].kind == 2) which.hole[which.n++] = (int)h;
		if (which.n) dev::k_ent_peaks<<<dim3(act >= 64 ? 4u : 32u, (unsigned)which.n, (unsigned)act), dev::ENT_THREADS, 0, st>>>((const dev::EntFrameJob *)d_frames_, which, (const dev::EntBandJob *)d_bands_,
		                                                  (const dev::EntSegJob *)d_segband_, geom, (const dev::EntSegState *)d_segs_, (const dev::EntBandState *)d_bandstate_);
	}
#ifdef CFHD_AMD_PROBES
	const int emit_probe = []{ const char *e = getenv("CFHD_AMD_EMIT_PROBE"); return e ? atoi(e) : 0; }();
#else
	const int emit_probe = 0;
#endif
	dev::k_ent_emit<<<(total_segs + dev::ENT_WAVES * dev::ENT_EMIT_SEGS - 1) / (dev::ENT_WAVES * dev::ENT_EMIT_SEGS), dev::ENT_THREADS, 0, st>>>(total_segs, (const dev::EntSegState *)d_segs_,
	                                                          T, (const uint32_t *)d_tokens_, emit_probe);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord((hipEvent_t)ev_[4], st));
	timed_ = true;
	return 0;
}

float GpuEntropyEncoder::kernel_ms(int k)
{
	float ms = 0;
	if (!timed_ || k < 0 || k > 4) return 0;
	if (k == 4) {                                        // the level-1 part of k_ent_count on the second stream (it runs beside the level-2 / level-3 transforms: its own events)
		if (!split_ || hipEventElapsedTime(&ms, (hipEvent_t)ev2_[0], (hipEvent_t)ev2_[1]) != hipSuccess) { (void)hipGetLastError(); return 0; }
		return ms;
	}
	void *end = (k == 0 && split_) ? ev2_[2] : ev_[k + 1];      // (the main stream's own launches, not its wait for the second stream)
	void *begin = (k == 0 && serial_split_) ? ev2_[1] : ev_[k];   // (one stream: the level-1 count ran in front, between ev2_[0] and ev2_[1])
	if (hipEventElapsedTime(&ms, (hipEvent_t)begin, (hipEvent_t)end) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return ms;
}

int GpuEntropyEncoder::fetch_sizes()
{
	(void)hipSetDevice(device_);
	hipStream_t st = (hipStream_t)stream_;
	HIPCHK(hipMemcpyAsync(h_sizes_, d_sizes_, sizeof(uint32_t) * 2 * n_, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	return 0;
}

int GpuEntropyEncoder::download() { const int rc = download_queue(); return rc ? rc : download_finish(); }

// The part of download() that can be queued behind the kernels without the host: the dense copy of the samples in HBM, sizes and offsets on their way to the host.
int GpuEntropyEncoder::download_queue()
{
	(void)hipSetDevice(device_);
	hipStream_t st = (hipStream_t)stream_;
	// pack in HBM, then one copy (SDMA engine when the runtime has it enabled)
	const bool direct = false;      // (k_ent_pack storing straight into the pinned host buffer: measured slower than pack + one SDMA copy in round 2)
	(void)hipGetLastError();
	const int act = active_frames();
	dev::k_ent_pack_offsets<<<1, dev::ENT_THREADS, 0, st>>>(d_sizes_, act, d_offsets_);
	dev::k_ent_pack<<<dim3(direct ? 2 : 8, (unsigned)act), dev::ENT_THREADS, 0, st>>>(d_samples_, cap_, d_sizes_, d_offsets_, direct ? h_samples_ : d_packed_);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(h_sizes_, d_sizes_, sizeof(uint32_t) * 2 * n_, hipMemcpyDeviceToHost, st));
	HIPCHK(hipMemcpyAsync(h_offsets_, d_offsets_, sizeof(uint32_t) * (act + 1), hipMemcpyDeviceToHost, st));
	// The copy of the sample bytes needs their number, which the host learns when the sizes arrive (download_finish()).  A pass that is queued as a whole
	// (cfhd_amd_batch_submit) would start that copy only when its result is collected, a step or two later; so the copy goes out now, sized by what the last pass of this
	// many frames produced plus a margin, and download_finish() adds the remainder if this pass turned out larger (consecutive passes of a sequence differ by a few percent).
	copied_ahead_ = 0;
	if (!direct && expect_bytes_ && expect_frames_ == act) {
		copied_ahead_ = expect_bytes_ < cap_ * (size_t)act ? expect_bytes_ : cap_ * (size_t)act;
		HIPCHK(hipMemcpyAsync(h_samples_, d_packed_, copied_ahead_, hipMemcpyDeviceToHost, st));
	}
	return 0;
}

// ... and the part that needs the sizes: waits for the stream, then queues the one copy of all sample bytes (wait on the stream afterwards).
int GpuEntropyEncoder::download_finish()
{
	(void)hipSetDevice(device_);
	hipStream_t st = (hipStream_t)stream_;
	const bool direct = false;
	const int act = active_frames();
	HIPCHK(hipStreamSynchronize(st));
	const size_t total = h_offsets_[act];
	if (!direct && total > copied_ahead_) HIPCHK(hipMemcpyAsync(h_samples_ + copied_ahead_, d_packed_ + copied_ahead_, total - copied_ahead_, hipMemcpyDeviceToHost, st));
	expect_bytes_ = speculative_download_ ? ((total + total / 32 + 4095) & ~(size_t)4095) : 0; expect_frames_ = act;
	return 0;
}

struct GpuEntropyDecoder::Host {
	std::vector<std::vector<dev::DecBandJob>> bands;      // per frame
	std::vector<std::vector<dev::DecLowpassJob>> lows;
	std::vector<size_t> host_bytes;                       // bytes to copy H2D per frame (0: sample already in HBM)
	std::vector<std::vector<dev::DecDiffJob>> diffs;     // per frame: the difference-coded band of every channel (interlaced samples)
	dev::DecBandJob *flat_bands = nullptr; dev::DecLowpassJob *flat_lows = nullptr; dev::DecDiffJob *flat_diffs = nullptr;   // pinned
};

GpuEntropyDecoder::GpuEntropyDecoder() : host_(new Host) {}
GpuEntropyDecoder::~GpuEntropyDecoder() { release(); delete host_; }

void GpuEntropyDecoder::release()
{
	(void)hipSetDevice(device_);
	void *dev[] = { d_samples_, d_tables_, d_bandjobs_, d_lowjobs_, d_errors_, d_plan_, d_idx_tables_, d_entries_, d_recs_, d_chunk_base_, d_chunk_job_, d_sums_, d_counters_, d_tile_start_, d_stats_, d_repair_, d_alts_, d_reindex_, d_diffjobs_, d_alt_entries_, d_masks_ };
	d_masks_ = nullptr; masks_per_frame_ = 0; use_blocks_ = false; blocks_written_ = false;
	for (void *p : dev) if (p) (void)hipFree(p);
	d_idx_tables_ = d_entries_ = d_recs_ = d_chunk_base_ = d_chunk_job_ = d_sums_ = d_counters_ = d_tile_start_ = d_stats_ = d_repair_ = d_alts_ = d_reindex_ = d_diffjobs_ = d_alt_entries_ = nullptr;
	if (h_chunk_job_) (void)hipHostFree(h_chunk_job_);
	if (h_counters_) (void)hipHostFree(h_counters_);
	h_chunk_job_ = nullptr; h_counters_ = nullptr;
	if (h_samples_) (void)hipHostFree(h_samples_);
	if (h_errors_) (void)hipHostFree(h_errors_);
	if (host_->flat_bands) { (void)hipHostFree(host_->flat_bands); host_->flat_bands = nullptr; }
	if (host_->flat_lows) { (void)hipHostFree(host_->flat_lows); host_->flat_lows = nullptr; }
	if (host_->flat_diffs) { (void)hipHostFree(host_->flat_diffs); host_->flat_diffs = nullptr; }
	for (void *&e : ev_) if (e) { (void)hipEventDestroy((hipEvent_t)e); e = nullptr; }
	if (ev_l23_) { (void)hipEventDestroy((hipEvent_t)ev_l23_); ev_l23_ = nullptr; }
	if (ev_low_) { (void)hipEventDestroy((hipEvent_t)ev_low_); ev_low_ = nullptr; }
	timed_ = false;
	d_samples_ = h_samples_ = nullptr; d_tables_ = d_bandjobs_ = d_lowjobs_ = d_plan_ = nullptr; d_errors_ = h_errors_ = nullptr; n_ = 0; ext_samples_ = nullptr;
}

int GpuEntropyDecoder::prepare(const FramePlan &plan, int nframes, int16_t *d_coeffs, size_t stride, size_t sample_cap, int out_kind, void *stream)
{
	int rc = device_init();
	if (rc) return rc;
	release();
	device_ = device_current(); (void)hipSetDevice(device_);
	plan_ = plan; n_ = nframes; cap_ = (sample_cap + 255) & ~(size_t)255; stream_ = stream; d_coeffs_ = d_coeffs; coeff_stride_ = stride; out_kind_ = out_kind;
	std::vector<uint32_t> t = build_dec_tables(1);
	HIPCHK(hipMalloc(&d_tables_, t.size() * 4));
	HIPCHK(hipMemcpy(d_tables_, t.data(), t.size() * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMalloc((void **)&d_samples_, cap_ * n_));
	HIPCHK(hipHostMalloc((void **)&h_samples_, cap_ * n_, hipHostMallocPortable));
	const size_t max_bands = (size_t)n_ * kMaxChannels * 9, max_lows = (size_t)n_ * kMaxChannels;
	HIPCHK(hipMalloc(&d_bandjobs_, max_bands * sizeof(dev::DecBandJob)));
	HIPCHK(hipMalloc(&d_lowjobs_, max_lows * sizeof(dev::DecLowpassJob)));
	HIPCHK(hipHostMalloc((void **)&host_->flat_bands, max_bands * sizeof(dev::DecBandJob), hipHostMallocPortable));
	HIPCHK(hipHostMalloc((void **)&host_->flat_lows, max_lows * sizeof(dev::DecLowpassJob), hipHostMallocPortable));
	HIPCHK(hipMalloc(&d_diffjobs_, max_lows * sizeof(dev::DecDiffJob)));
	HIPCHK(hipHostMalloc((void **)&host_->flat_diffs, max_lows * sizeof(dev::DecDiffJob), hipHostMallocPortable));
	HIPCHK(hipMalloc((void **)&d_errors_, sizeof(int)));
	HIPCHK(hipHostMalloc((void **)&h_errors_, sizeof(int), hipHostMallocPortable));
	*h_errors_ = 0;
	{ const char *e = getenv("CFHD_AMD_DEC"); lane_kernel_ = e && strcmp(e, "lane") == 0; dx_ = !(e && (strcmp(e, "lane") == 0 || strcmp(e, "par") == 0)); }   // A/B switch: the round-1 kernels
	if (dx_) {
		dev::DecIdxTables *it = new dev::DecIdxTables;
		const bool ok = build_dec_index_tables(1, it);
		if (ok) { hipError_t e = hipMalloc(&d_idx_tables_, sizeof(*it)); if (e == hipSuccess) e = hipMemcpy(d_idx_tables_, it, sizeof(*it), hipMemcpyHostToDevice); if (e != hipSuccess) { delete it; return g_fail(e, "decoder tables"); } }
		delete it;
		if (!ok) return -6;
		// chunk arrays for the worst case: every frame's sample as long as its slot, every band with a partial chunk
		const size_t per_frame = cap_ / dev::DX_CHUNK_BYTES + (size_t)kMaxChannels * 9 + 1;
		if (per_frame * (size_t)n_ >= ((size_t)1 << 31) / dev::DX_ENTRY_STRIDE * 8) return -5;
		max_chunks_ = (uint32_t)(per_frame * (size_t)n_);
		HIPCHK(hipMalloc(&d_entries_, (size_t)max_chunks_ * dev::DX_ENTRY_STRIDE * 4));
		HIPCHK(hipMalloc(&d_recs_, (size_t)max_chunks_ * sizeof(dev::DxChunkRec)));
		HIPCHK(hipMalloc(&d_chunk_base_, (size_t)max_chunks_ * 4));
		HIPCHK(hipMalloc(&d_chunk_job_, (size_t)max_chunks_ * sizeof(dev::DxChunkDesc)));
		HIPCHK(hipMalloc(&d_sums_, max_bands * sizeof(dev::DxBandSum)));
		HIPCHK(hipMalloc(&d_counters_, 32));          // [0] chunks, [1] bands to repair, [2] chunks to re-index, [3] alternate-entry slots taken, [4] next chunk of k_dec_index
		HIPCHK(hipMalloc(&d_repair_, max_bands * 4));
		HIPCHK(hipMalloc(&d_alts_, (size_t)max_chunks_ * sizeof(dev::DxChunkAlt)));
		HIPCHK(hipMalloc(&d_reindex_, (size_t)max_chunks_ * sizeof(dev::DxReindex)));
		alt_slots_ = max_chunks_ / 8 > 64u ? max_chunks_ / 8 : 64u;          // entries of the extra candidates of chunks without a unique alignment (a few per cent of the chunks)
		HIPCHK(hipMalloc(&d_alt_entries_, (size_t)alt_slots_ * dev::DX_ENTRY_STRIDE * 4));
		{
			dev::DecPlan dp0; dec_build_plan(plan, out_kind, &dp0);
			const dev::DxTilePlan tp0 = dx_tile_plan(plan, dp0, n_, false);
			HIPCHK(hipMalloc(&d_tile_start_, ((size_t)tp0.total + 1) * sizeof(dev::DxTileDesc)));      // one record per tile (k_dec_tile_index)
			const char *se = getenv("CFHD_AMD_DX_STATS");
			if (se && atoi(se)) { HIPCHK(hipMalloc(&d_stats_, 64)); HIPCHK(hipMemset(d_stats_, 0, 64)); }
		}
		HIPCHK(hipHostMalloc((void **)&h_chunk_job_, (size_t)max_chunks_ * sizeof(dev::DxChunkDesc), hipHostMallocPortable));
		HIPCHK(hipHostMalloc((void **)&h_counters_, 32, hipHostMallocPortable));
		int cus = 256;
		(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_current());
		// workgroups that fit a CU at once: k_dec_index 31 KB of LDS each, k_dec_tiles its tables (22.5 KB) + the image of a tile
		grid_index_ = cus * 5; grid_tiles_ = cus * (int)((160 * 1024) / (sizeof(uint2) * (1 << dev::DX_KM) + sizeof(uint32_t) * (dev::DX_LONG11_MAX + dev::DX_TILE_WORDS)));
		if (const char *e = getenv("CFHD_AMD_DX_GRID_TILES")) if (atoi(e) > 0) grid_tiles_ = atoi(e);      // (sweeps: tools/dx_tile_sweep.sh)
		if (grid_index_ < 1) grid_index_ = 1;
		if (grid_tiles_ < 1) grid_tiles_ = 1;
	}
	{
		dev::DecPlan dp;
		dec_build_plan(plan, out_kind, &dp);
		HIPCHK(hipMalloc(&d_plan_, sizeof(dp)));
		HIPCHK(hipMemcpy(d_plan_, &dp, sizeof(dp), hipMemcpyHostToDevice));
	}
	// chunk masks of the level-1 bands as block lists (cfhd_core.h dec_block_list_layout): for 4:2:2 frames, whose last level has a strip kernel that gathers them
	if (dx_ && plan.encoded_format == ENC_YUV422 && plan.num_channels == 3) {
		int mask_base[kMaxChannels][kNumBands];
		masks_per_frame_ = (size_t)dec_block_list_layout(plan, mask_base);
		HIPCHK(hipMalloc((void **)&d_masks_, masks_per_frame_ * 8 * (size_t)n_));
	}
	host_->bands.assign(n_, {}); host_->lows.assign(n_, {}); host_->diffs.assign(n_, {}); host_->host_bytes.assign(n_, 0);
	for (void *&e : ev_) HIPCHK(hipEventCreate((hipEvent_t *)&e));
	HIPCHK(hipEventCreate((hipEvent_t *)&ev_l23_));
	HIPCHK(hipEventCreate((hipEvent_t *)&ev_low_));
	return 0;
}

int GpuEntropyDecoder::set_samples_device(const uint8_t *d_samples, size_t stride_bytes, const uint32_t *d_sizes)
{
	if (!d_samples || !d_sizes || (stride_bytes & 255) || ((uintptr_t)d_samples & 255)) return -1;   // k_dec_parse reads aligned 256-byte windows (a slot must extend to the window that holds its last tag)
	ext_samples_ = d_samples; ext_stride_ = stride_bytes; ext_sizes_ = d_sizes;
	return 0;
}

int GpuEntropyDecoder::set_sample_host(int i, const uint8_t *sample, size_t size)
{
	if (i < 0 || i >= n_ || size > cap_) return -1;
	memcpy(h_samples_ + cap_ * i, sample, size);
	int rc = set_sample_device(i, d_samples_ + cap_ * i, sample, size);
	if (rc == 0) host_->host_bytes[i] = (size + 3) & ~(size_t)3;
	return rc;
}

int GpuEntropyDecoder::set_sample_device(int i, const uint8_t *d_sample, const uint8_t *host_copy, size_t size)
{
	if (i < 0 || i >= n_) return -1;
	ext_samples_ = nullptr;
	ParsedSample ps;
	if (parse_sample(host_copy, size, &ps) != 0) return -2;
	if (ps.width != plan_.width || ps.display_height != plan_.display_height || ps.encoded_format != plan_.encoded_format || ps.num_channels != plan_.num_channels) return -3;
	host_->bands[i].clear(); host_->lows[i].clear(); host_->host_bytes[i] = 0;
	if (dx_) {
		// rows of the [slot][frame] job table, by slot
		dev::DecPlan dp; dec_build_plan(plan_, out_kind_, &dp);
		host_->bands[i].assign((size_t)dp.bands_per_frame, dev::DecBandJob());
		host_->lows[i].assign((size_t)plan_.num_channels, dev::DecLowpassJob());
		host_->diffs[i].assign((size_t)plan_.num_channels, dev::DecDiffJob());
		if (!dx_build_jobs(ps, plan_, dp, d_sample, d_coeffs_ + (size_t)i * coeff_stride_, out_kind_, 0, 1, host_->bands[i].data(), host_->lows[i].data(), skip_level1_,
		                   interlaced_ ? host_->diffs[i].data() : nullptr)) return -4;
		return 0;
	}
	if (interlaced_) return -4;                        // the round-1 kernels know one code set
	if (!dec_build_jobs(ps, plan_, d_sample, d_coeffs_ + (size_t)i * coeff_stride_, out_kind_, &host_->bands[i], &host_->lows[i], skip_level1_)) return -4;
	return 0;
}

enum { kLowLatencyFrames = 32 };     // up to here k_dec_bands_par_ll: measured 0.22 vs 0.37 ms for one 1080p frame, break-even near 64 frames

int GpuEntropyDecoder::launch()
{
	(void)hipSetDevice(device_);
	hipStream_t st = (hipStream_t)stream_;
	l23_split_ = false; blocks_written_ = false;
	if (ext_samples_) {
		// device-resident samples: parse on the GPU; the job tables have one row per band type (largest first), nframes wide
		const int nch = plan_.num_channels, nb = n_ * nch * 9;
		HIPCHK(hipMemsetAsync(d_errors_, 0, sizeof(int), st));
		(void)hipGetLastError();
		// CFHD_AMD_PARSE_EARLY=0: (A/B) the parser behind the finished payloads instead of beside k_ent_emit
		static const bool parse_early = [] { const char *e = getenv("CFHD_AMD_PARSE_EARLY"); return !(e && atoi(e) == 0); }();
		if (!parse_early && ev_payloads_) ev_headers_ = ev_payloads_;
		if (ev_headers_) HIPCHK(hipStreamWaitEvent(st, (hipEvent_t)ev_headers_, 0));
		HIPCHK(hipEventRecord((hipEvent_t)ev_[0], st));
		dev::k_dec_parse<<<n_, dev::DEC_PARSE_THREADS, 0, st>>>(ext_samples_, ext_stride_, ext_sizes_, n_,
			(const dev::DecPlan *)d_plan_, d_coeffs_, coeff_stride_, (dev::DecBandJob *)d_bandjobs_, (dev::DecLowpassJob *)d_lowjobs_, d_errors_,
			interlaced_ && dx_ ? (dev::DecDiffJob *)d_diffjobs_ : nullptr);
		parse_end_ = ev_payloads_ != nullptr;
		if (parse_end_) { HIPCHK(hipEventRecord((hipEvent_t)ev_[4], st)); HIPCHK(hipStreamWaitEvent(st, (hipEvent_t)ev_payloads_, 0)); }
		ev_headers_ = ev_payloads_ = nullptr;
		HIPCHK(hipEventRecord((hipEvent_t)ev_[1], st));
		// few frames: the latency shape (the launch lasts as long as the longest band's serial steps); many: the throughput shape
		l23_split_ = false;
		if (dx_) { const int rc_dx = launch_dx(true, nb, 0u, n_ * nch); if (rc_dx) return rc_dx; }
		else if (n_ <= kLowLatencyFrames) dev::k_dec_bands_par_ll<<<nb, dev::DECP_LL_THREADS, 0, st>>>((const dev::DecBandJob *)d_bandjobs_, (const dev::DecTables *)d_tables_, d_errors_);
		else dev::k_dec_bands_par<<<nb, dev::DECP_THREADS, 0, st>>>((const dev::DecBandJob *)d_bandjobs_, (const dev::DecTables *)d_tables_, d_errors_);
		HIPCHK(hipEventRecord((hipEvent_t)ev_[2], st));
		if (!l23_split_) dev::k_dec_lowpass<<<dim3(8, (unsigned)(n_ * nch)), 256, 0, st>>>((const dev::DecLowpassJob *)d_lowjobs_);      // (split: launched between the two tile passes)
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord((hipEvent_t)ev_[3], st));
		timed_ = true;
		HIPCHK(hipMemcpyAsync(h_errors_, d_errors_, sizeof(int), hipMemcpyDeviceToHost, st));
		return 0;
	}
	parse_end_ = false;
	if (dx_) {
		// host-parsed samples: the [slot][frame] job table and the chunk numbering come from the host
		dev::DecPlan dp; dec_build_plan(plan_, out_kind_, &dp);
		const int nch = plan_.num_channels, act = active_frames(), nb = dp.bands_per_frame * act;      // frames 0 .. act-1 of the batch carry samples
		HIPCHK(hipStreamSynchronize(st));                                     // the pinned tables of the previous launch may still be in flight
		for (int f = 0; f < act; f++) {
			if ((int)host_->bands[f].size() != dp.bands_per_frame || (int)host_->lows[f].size() != nch) return -1;
			for (int s = 0; s < dp.bands_per_frame; s++) host_->flat_bands[(size_t)s * act + f] = host_->bands[f][s];
			for (int c = 0; c < nch; c++) host_->flat_lows[(size_t)f * nch + c] = host_->lows[f][c];
			if (interlaced_) { if ((int)host_->diffs[f].size() != nch) return -1; for (int c = 0; c < nch; c++) host_->flat_diffs[(size_t)f * nch + c] = host_->diffs[f][c]; }
		}
		std::vector<dev::DxChunkDesc> cj;
		const uint32_t nchunks = dx_number_chunks(host_->flat_bands, nb, &cj);
		if (nchunks > max_chunks_) return -5;
		memcpy(h_chunk_job_, cj.data(), cj.size() * sizeof(dev::DxChunkDesc));
		h_counters_[0] = nchunks; h_counters_[1] = 0; h_counters_[2] = 0; h_counters_[3] = 0; h_counters_[4] = 0;
		HIPCHK(hipMemsetAsync(d_errors_, 0, sizeof(int), st));
		for (int f = 0; f < act; f++)
			if (host_->host_bytes[f]) HIPCHK(hipMemcpyAsync(d_samples_ + cap_ * f, h_samples_ + cap_ * f, host_->host_bytes[f], hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(d_bandjobs_, host_->flat_bands, (size_t)nb * sizeof(dev::DecBandJob), hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(d_lowjobs_, host_->flat_lows, (size_t)act * nch * sizeof(dev::DecLowpassJob), hipMemcpyHostToDevice, st));
		if (interlaced_) HIPCHK(hipMemcpyAsync(d_diffjobs_, host_->flat_diffs, (size_t)act * nch * sizeof(dev::DecDiffJob), hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(d_chunk_job_, h_chunk_job_, (size_t)nchunks * sizeof(dev::DxChunkDesc), hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(d_counters_, h_counters_, 20, hipMemcpyHostToDevice, st));
		(void)hipGetLastError();
		HIPCHK(hipEventRecord((hipEvent_t)ev_[0], st));
		HIPCHK(hipEventRecord((hipEvent_t)ev_[1], st));
		l23_split_ = false;
		int rc = launch_dx(false, nb, nchunks, act * nch);
		if (rc) return rc;
		HIPCHK(hipEventRecord((hipEvent_t)ev_[2], st));
		if (!l23_split_) dev::k_dec_lowpass<<<dim3(8, (unsigned)(act * nch)), 256, 0, st>>>((const dev::DecLowpassJob *)d_lowjobs_);
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord((hipEvent_t)ev_[3], st));
		timed_ = true;
		HIPCHK(hipMemcpyAsync(h_errors_, d_errors_, sizeof(int), hipMemcpyDeviceToHost, st));
		return 0;
	}
	const bool lane_kernel = lane_kernel_;
	dev::DecBandJob *fb = host_->flat_bands; dev::DecLowpassJob *fl = host_->flat_lows;
	size_t nfb = 0, nfl = 0, per_frame = 0;
	// band-type major: the 64 lanes of a wave (lane kernel) decode bands of similar length, and the workgroups of the parallel
	// kernel start with the long bands (level-1 luma first) so that the short ones fill the tail of the launch
	for (int f = 0; f < n_; f++) per_frame = host_->bands[f].size() > per_frame ? host_->bands[f].size() : per_frame;
	for (size_t k = 0; k < per_frame; k++) for (int f = 0; f < n_; f++) if (k < host_->bands[f].size()) fb[nfb++] = host_->bands[f][k];
	for (int f = 0; f < n_; f++) for (const dev::DecLowpassJob &j : host_->lows[f]) fl[nfl++] = j;
	if (!nfb) return -1;
	if (!lane_kernel) std::stable_sort(fb, fb + nfb, [](const dev::DecBandJob &a, const dev::DecBandJob &b) { return a.bytes > b.bytes; });
	else HIPCHK(hipMemset2DAsync(d_coeffs_, coeff_stride_ * 2, 0, (size_t)plan_.final_elems * 2, n_, st));   // the parallel kernel clears its own bands
	HIPCHK(hipMemsetAsync(d_errors_, 0, sizeof(int), st));
	for (int f = 0; f < n_; f++)
		if (host_->host_bytes[f]) HIPCHK(hipMemcpyAsync(d_samples_ + cap_ * f, h_samples_ + cap_ * f, host_->host_bytes[f], hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_bandjobs_, fb, nfb * sizeof(dev::DecBandJob), hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_lowjobs_, fl, nfl * sizeof(dev::DecLowpassJob), hipMemcpyHostToDevice, st));
	(void)hipGetLastError();
	const int nb = (int)nfb;
	HIPCHK(hipEventRecord((hipEvent_t)ev_[0], st));
	HIPCHK(hipEventRecord((hipEvent_t)ev_[1], st));
	if (lane_kernel) dev::k_dec_bands<<<(nb + dev::DEC_THREADS - 1) / dev::DEC_THREADS, dev::DEC_THREADS, 0, st>>>((const dev::DecBandJob *)d_bandjobs_, nb, (const dev::DecTables *)d_tables_, d_errors_);
	else if (n_ <= kLowLatencyFrames) dev::k_dec_bands_par_ll<<<nb, dev::DECP_LL_THREADS, 0, st>>>((const dev::DecBandJob *)d_bandjobs_, (const dev::DecTables *)d_tables_, d_errors_);
	else dev::k_dec_bands_par<<<nb, dev::DECP_THREADS, 0, st>>>((const dev::DecBandJob *)d_bandjobs_, (const dev::DecTables *)d_tables_, d_errors_);
	HIPCHK(hipEventRecord((hipEvent_t)ev_[2], st));
	dev::k_dec_lowpass<<<dim3(8, (unsigned)nfl), 256, 0, st>>>((const dev::DecLowpassJob *)d_lowjobs_);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord((hipEvent_t)ev_[3], st));
	timed_ = true;
	HIPCHK(hipMemcpyAsync(h_errors_, d_errors_, sizeof(int), hipMemcpyDeviceToHost, st));
	return 0;
}

// The chunk-indexed decoder on the batch's stream.  device_jobs: the job table was filled by k_dec_parse (chunks are numbered on the device).
int GpuEntropyDecoder::launch_dx(bool device_jobs, int njobs, uint32_t host_chunks, int lowpass_jobs)
{
	hipStream_t st = (hipStream_t)stream_;
	dev::DecBandJob *jobs = (dev::DecBandJob *)d_bandjobs_;
	const dev::DecIdxTables *T = (const dev::DecIdxTables *)d_idx_tables_;
	dev::DecPlan dp; dec_build_plan(plan_, out_kind_, &dp);
	const int frames = device_jobs ? n_ : active_frames();
	const bool lists = use_blocks_ && d_masks_ && !skip_level1_;
	const dev::DxTilePlan tp = dx_tile_plan(plan_, dp, frames, skip_level1_, lists, interlaced_);
	unsigned long long *const tmasks = lists ? d_masks_ : nullptr;
	blocks_written_ = lists;
	const char *spec_env = getenv("CFHD_AMD_DX_SPECULATE");
	const bool speculate = !(spec_env && atoi(spec_env) == 0);            // 0: every chunk goes through the repair path (tests)
	if (device_jobs) HIPCHK(hipMemsetAsync((uint32_t *)d_counters_ + 1, 0, 16, st));     // the repair and re-index lists, the alternate-entry slots and the chunk counter start at zero (the host path uploads zeroed counters)
	if (device_jobs) {
		dev::k_dec_plan<<<1, 1024, 0, st>>>(jobs, njobs, max_chunks_, (uint32_t *)d_counters_, d_errors_);
		dev::k_dec_plan_fill<<<(njobs + dev::DX_WAVES - 1) / dev::DX_WAVES, dev::DX_THREADS, 0, st>>>(jobs, njobs, (dev::DxChunkDesc *)d_chunk_job_, (const uint32_t *)d_counters_);
	}
	HIPCHK(hipEventRecord((hipEvent_t)ev_[7], st));         // (k_dec_index is timed by itself: kernel_ms(3); the plan in front of it: kernel_ms(6))
	// grid-stride kernels: as many workgroups as the chip holds at once, fewer when there is less work
	const uint32_t chunk_bound = device_jobs ? max_chunks_ : host_chunks;
	int g1 = grid_index_, g3 = grid_tiles_;
	if ((uint32_t)g1 * dev::DX_WAVES > chunk_bound) g1 = (int)((chunk_bound + dev::DX_WAVES - 1) / dev::DX_WAVES);
	const uint32_t tile_waves = 1u;                       // (a workgroup per tile)
	if ((uint32_t)g3 > tp.total) g3 = (int)tp.total;
	if (g1 < 1) g1 = 1;
	if (g3 < 1) g3 = 1;
	dev::k_dec_index<<<g1, dev::DX_THREADS, 0, st>>>((const dev::DxChunkDesc *)d_chunk_job_, (const uint32_t *)d_counters_, T, (uint32_t *)d_entries_, (dev::DxChunkRec *)d_recs_, (dev::DxChunkAlt *)d_alts_, speculate ? 1 : 0, (uint32_t *)d_stats_,
	                                                 (uint32_t *)d_alt_entries_, alt_slots_, (uint32_t *)d_counters_ + 3, (uint32_t *)d_counters_ + 4);
	HIPCHK(hipEventRecord((hipEvent_t)ev_[5], st));
	dev::k_dec_chain<<<(njobs + dev::DX_WAVES - 1) / dev::DX_WAVES, dev::DX_THREADS, 0, st>>>(jobs, njobs, (const dev::DxChunkRec *)d_recs_, (const dev::DxChunkAlt *)d_alts_, (uint32_t *)d_chunk_base_,
	                                                                                            (dev::DxBandSum *)d_sums_, d_errors_, (uint32_t *)d_repair_, (dev::DxReindex *)d_reindex_, (uint32_t *)d_counters_);
	const int small_grid = njobs < 256 ? (njobs + dev::DX_WAVES - 1) / dev::DX_WAVES : 64;
	dev::k_dec_repair<<<small_grid, dev::DX_THREADS, 0, st>>>(jobs, T, (uint32_t *)d_entries_, (dev::DxChunkRec *)d_recs_, (const dev::DxChunkAlt *)d_alts_, (uint32_t *)d_chunk_base_, (dev::DxBandSum *)d_sums_,
	                                                         d_errors_, (const uint32_t *)d_repair_, (dev::DxReindex *)d_reindex_, (uint32_t *)d_counters_, (uint32_t *)d_stats_);
	dev::k_dec_reindex<<<g1, dev::DX_THREADS, 0, st>>>(jobs, T, (uint32_t *)d_entries_, (const dev::DxReindex *)d_reindex_, (const uint32_t *)d_counters_, (uint32_t *)d_stats_,
	                                                   (const dev::DxChunkAlt *)d_alts_, (const uint32_t *)d_alt_entries_);
	dev::k_dec_tile_index<<<(tp.total + dev::DX_TILE_INDEX_TILES - 1) / dev::DX_TILE_INDEX_TILES, dev::DX_THREADS, 0, st>>>(jobs, tp, (const uint32_t *)d_entries_, (const uint32_t *)d_chunk_base_, (const dev::DxBandSum *)d_sums_,
	                                                                                                  (dev::DxTileDesc *)d_tile_start_, tmasks, (uint32_t)masks_per_frame_);
	HIPCHK(hipEventRecord((hipEvent_t)ev_[6], st));
	// Many frames: the tiles of the level-2 / level-3 bands (a quarter of them) and the lowpass bands first, an event behind them, then the level-1 tiles -- the caller
	// may run the inverse transforms of levels 3 and 2 on another stream beside the second launch (DecodeBatch::launch_inverse).  Off unless CFHD_AMD_TILES_SPLIT=1:
	// measured in round 3 (1080p, 512 frames), the step does not get shorter (48.1 k fps either way) -- k_dec_tiles is bound by instruction issue and owns its CUs' LDS,
	// the plane kernels beside it only stretch it from 1.67 to 2.25 ms.  (The same idea pays on the encoder side, where the kernel that shares the chip waits on memory.)
	const char *split_env = getenv("CFHD_AMD_TILES_SPLIT");
	l23_split_ = frames >= 8 && !skip_level1_ && tp.split > 0 && tp.split < tp.total && split_env && split_env[0] == '1';
	auto tile_pass = [&](const dev::DxTilePlan &p, int g) {
		dev::k_dec_tiles<dev::DX_TILE_THREADS><<<g < 1 ? 1 : g, dev::DX_TILE_THREADS, 0, st>>>((const dev::DxTileDesc *)d_tile_start_, p.first, p.total, T, (const uint32_t *)d_entries_, (const uint32_t *)d_chunk_base_);
	};
	if (l23_split_) {
		dev::DxTilePlan ta = tp, tb = tp;
		ta.total = tp.split; tb.first = tp.split;
		int ga = g3, gb = g3;
		if ((uint32_t)ga * tile_waves > ta.total) ga = (int)((ta.total + tile_waves - 1) / tile_waves);
		if ((uint32_t)gb * tile_waves > tb.total - tb.first) gb = (int)((tb.total - tb.first + tile_waves - 1) / tile_waves);
		tile_pass(ta, ga);
		HIPCHK(hipEventRecord((hipEvent_t)ev_low_, st));
		dev::k_dec_lowpass<<<dim3(8, (unsigned)lowpass_jobs), 256, 0, st>>>((const dev::DecLowpassJob *)d_lowjobs_);
		HIPCHK(hipEventRecord((hipEvent_t)ev_l23_, st));
		tile_pass(tb, gb);
	} else tile_pass(tp, g3);
	if (interlaced_) {
		// the difference-coded band of every channel back to coefficients: a wave per row for the bands without a peak table, the workgroup-per-band
		// kernel of round 3 for the few that have one (measured against that kernel for all of them: profiles/r04_o_*)
		const bool rows = true;
		if (rows) dev::k_dec_undiff_rows<<<dim3((unsigned)(frames * plan_.num_channels), 32), 64 * dev::DXR_WAVES, 0, st>>>((const dev::DecDiffJob *)d_diffjobs_);
		dev::k_dec_undiff<<<dim3((unsigned)(frames * plan_.num_channels), dev::DXU_SPLIT), dev::DXU_THREADS, 0, st>>>((const dev::DecDiffJob *)d_diffjobs_, d_errors_, rows ? 1 : 0);
	}
	HIPCHK(hipGetLastError());
	return 0;
}

int GpuEntropyDecoder::check() { return *h_errors_ ? -1 : 0; }

// CFHD_AMD_DX_STATS=1: counters of k_dec_index / k_dec_chain since prepare(): [0] rounds, [1] chunks indexed, [2] most rounds of a chunk, [3] chunks repaired,
// [4..15] lanes that restarted in round 0..10, 11+.  Synchronises the stream.
int GpuEntropyDecoder::stats(uint32_t out[16])
{
	(void)hipSetDevice(device_);
	memset(out, 0, 64);
	if (!d_stats_) return -1;
	HIPCHK(hipStreamSynchronize((hipStream_t)stream_));
	HIPCHK(hipMemcpy(out, d_stats_, 64, hipMemcpyDeviceToHost));
	return 0;
}

float GpuEntropyDecoder::kernel_ms(int k)
{
	float ms = 0;
	if (k >= 3) {                                                // the chunk-indexed decoder's own kernels
		if (!timed_ || !dx_ || k > 6) return 0;
		// 3: k_dec_index, 4: k_dec_chain (+ repair, reindex, tile index), 5: k_dec_tiles, 6: k_dec_plan + k_dec_plan_fill (one workgroup that numbers the chunks: latency, not work)
		void *a = k == 3 ? ev_[7] : (k == 4 ? ev_[5] : (k == 5 ? ev_[6] : ev_[1])), *b = k == 3 ? ev_[5] : (k == 4 ? ev_[6] : (k == 5 ? ev_[2] : ev_[7]));
		if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) { (void)hipGetLastError(); return 0; }
		if (k == 5 && l23_split_) {                          // k_dec_lowpass ran between the two tile passes: not part of k_dec_tiles
			float low = 0;
			if (hipEventElapsedTime(&low, (hipEvent_t)ev_low_, (hipEvent_t)ev_l23_) == hipSuccess) ms -= low; else (void)hipGetLastError();
		}
		return ms;
	}
	if (k == 2 && l23_split_ && timed_) {                        // k_dec_lowpass between the two tile passes
		if (hipEventElapsedTime(&ms, (hipEvent_t)ev_low_, (hipEvent_t)ev_l23_) != hipSuccess) { (void)hipGetLastError(); return 0; }
		return ms;
	}
	void *end = (k == 0 && parse_end_) ? ev_[4] : ev_[k + 1];      // the parser's own end, not the start of the band decoder that waited for the payloads
	if (!timed_ || k < 0 || k > 2 || hipEventElapsedTime(&ms, (hipEvent_t)ev_[k], (hipEvent_t)end) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return ms;
}

// =============================================================================================
// GpuGroupEntropyDecoder
// =============================================================================================
void GpuGroupEntropyDecoder::release()
{
	(void)hipSetDevice(device_);
	void *dev[] = { d_sample_, d_tables_, d_tables18_, d_bandjobs_, d_lowjobs_, d_diffjobs_, d_errors_ };
	for (void *p : dev) if (p) (void)hipFree(p);
	void *host[] = { h_sample_, h_bandjobs_, h_lowjobs_, h_diffjobs_, h_errors_ };
	for (void *p : host) if (p) (void)hipHostFree(p);
	d_sample_ = h_sample_ = nullptr; d_tables_ = d_tables18_ = d_bandjobs_ = d_lowjobs_ = d_diffjobs_ = h_bandjobs_ = h_lowjobs_ = h_diffjobs_ = nullptr; d_errors_ = h_errors_ = nullptr;
}

enum { kGroupBandJobs = 3 * 15, kGroupRawJobs = 3 * 2, kGroupDiffJobs = 3 * 2 };

int GpuGroupEntropyDecoder::prepare(const GopPlan &plan, int16_t *d_coeffs, size_t sample_cap, int out_kind, void *stream, int device)
{
	int rc = device_init();
	if (rc) return rc;
	release();
	device_ = device >= 0 ? device : device_current(); (void)hipSetDevice(device_);      // (the GPU the pyramid and the stream live on)
	plan_ = plan; d_coeffs_ = d_coeffs; cap_ = (sample_cap + 255) & ~(size_t)255; out_kind_ = out_kind; stream_ = stream;
	std::vector<uint32_t> t = build_dec_tables(1);
	HIPCHK(hipMalloc(&d_tables_, t.size() * 4));
	HIPCHK(hipMemcpy(d_tables_, t.data(), t.size() * 4, hipMemcpyHostToDevice));
	t = build_dec_tables(2);
	HIPCHK(hipMalloc(&d_tables18_, t.size() * 4));
	HIPCHK(hipMemcpy(d_tables18_, t.data(), t.size() * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMalloc(&d_diffjobs_, kGroupDiffJobs * sizeof(dev::DecDiffJob)));
	HIPCHK(hipHostMalloc(&h_diffjobs_, kGroupDiffJobs * sizeof(dev::DecDiffJob), hipHostMallocPortable));
	HIPCHK(hipMalloc((void **)&d_sample_, cap_));
	HIPCHK(hipHostMalloc((void **)&h_sample_, cap_, hipHostMallocPortable));
	HIPCHK(hipMalloc(&d_bandjobs_, kGroupBandJobs * sizeof(dev::DecBandJob)));
	HIPCHK(hipMalloc(&d_lowjobs_, kGroupRawJobs * sizeof(dev::DecLowpassJob)));
	HIPCHK(hipHostMalloc(&h_bandjobs_, kGroupBandJobs * sizeof(dev::DecBandJob), hipHostMallocPortable));
	HIPCHK(hipHostMalloc(&h_lowjobs_, kGroupRawJobs * sizeof(dev::DecLowpassJob), hipHostMallocPortable));
	HIPCHK(hipMalloc((void **)&d_errors_, sizeof(int)));
	HIPCHK(hipHostMalloc((void **)&h_errors_, sizeof(int), hipHostMallocPortable));
	*h_errors_ = 0;
	return 0;
}

int GpuGroupEntropyDecoder::launch(const uint8_t *sample, size_t size, const ParsedGroup &pg)
{
	(void)hipSetDevice(device_);
	hipStream_t st = (hipStream_t)stream_;
	if (size > cap_) return -1;
	HIPCHK(hipStreamSynchronize(st));                                       // the pinned sample / tables of the previous launch may still be in flight
	dev::DecBandJob *bj = (dev::DecBandJob *)h_bandjobs_; dev::DecLowpassJob *lj = (dev::DecLowpassJob *)h_lowjobs_; dev::DecDiffJob *dj = (dev::DecDiffJob *)h_diffjobs_;
	int nb = 0, nl = 0, nd = 0;
	for (int c = 0; c < 3; c++) {
		const GopChannel &ch = plan_.ch[c];
		const ParsedBand &lp = pg.lowpass[c];
		const GopWavelet &top = ch.w[5];
		if (!lp.present || lp.width != top.width || lp.height != top.height || (size_t)lp.offset + (size_t)top.width * top.height * 2 > size) return -2;
		// the bias the reference adds to the lowpass band while unpacking it: twice the intra frame's for a group (decoder.c:12265 `num_frames == 2 ? 48 : 24`)
		lj[nl++] = dev::DecLowpassJob{ d_sample_ + lp.offset, d_coeffs_ + top.offset[0], top.width, top.height, top.pitch, 2 * lowpass_bias(10, top.width, out_kind_) };
		static const int coded[5] = { 5, 4, 3, 1, 0 };
		for (int k : coded) {
			const GopWavelet &wv = ch.w[k];
			for (int b = (k == 3 ? 0 : 1); b < 4; b++) {
				const ParsedBand &pb = pg.band[c][k][b];
				if (!pb.present || pb.width != wv.width || pb.height != wv.height || (pb.offset & 3) || (size_t)pb.offset + pb.bytes > size) return -2;
				if (pb.codebook < 0) {                                       // raw 16-bit words (the lowpass band of the temporal highpass wavelet): signed, no bias
					if ((size_t)pb.bytes < (size_t)wv.width * wv.height * 2 || pb.quant != 1 || (wv.width & 1)) return -3;
					lj[nl++] = dev::DecLowpassJob{ d_sample_ + pb.offset, d_coeffs_ + wv.offset[0], wv.width, wv.height, wv.pitch, 0 };
					continue;
				}
				// the difference-coded bands of an interlaced group (band 2 of the two frame wavelets: subbands 12 and 15) come in code set 18, maybe with a peak table
				const bool diff = gop_band_is_difference_coded(plan_, k, b);
				if (pb.codebook != (diff ? 2 : 1) || pb.difference != diff || (wv.offset[b] & 7) || (wv.pitch & 7)) return -3;
				if (diff) {
					if (nd >= kGroupDiffJobs || wv.width > 16 * 256 || (pb.peak_level && (size_t)pb.peak_offset + 2 > size)) return -3;      // (k_dec_undiff: rows of up to DXU_MAX x DXU_THREADS coefficients)
					dj[nd++] = dev::DecDiffJob{ d_coeffs_ + wv.offset[b], wv.width, wv.height, wv.pitch, pb.peak_level ? d_sample_ + pb.peak_offset : nullptr,
					                            pb.peak_level ? (uint32_t)(size - pb.peak_offset) : 0u, pb.peak_level };
				}
				bj[nb++] = dev::DecBandJob{ d_sample_ + pb.offset, pb.bytes, d_coeffs_ + wv.offset[b], wv.height * wv.pitch, pb.quant, 0u, diff ? 1 : 0 };
			}
		}
	}
	if (nb != kGroupBandJobs || nl != kGroupRawJobs || nd != (plan_.interlaced ? kGroupDiffJobs : 0)) return -2;
	// the bands of code set 17 in front, those of code set 18 behind them (one launch each: the kernel reads one set of tables); in either part the long bands start first
	std::stable_sort(bj, bj + nb, [](const dev::DecBandJob &a, const dev::DecBandJob &b) { return a.table != b.table ? a.table < b.table : a.bytes > b.bytes; });
	const int nb18 = nd, nb17 = nb - nb18;
	memcpy(h_sample_, sample, size);
	HIPCHK(hipMemsetAsync(d_errors_, 0, sizeof(int), st));
	HIPCHK(hipMemcpyAsync(d_sample_, h_sample_, (size + 3) & ~(size_t)3, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_bandjobs_, bj, nb * sizeof(dev::DecBandJob), hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(d_lowjobs_, lj, nl * sizeof(dev::DecLowpassJob), hipMemcpyHostToDevice, st));
	(void)hipGetLastError();
	dev::k_dec_bands_par_ll<<<nb17, dev::DECP_LL_THREADS, 0, st>>>((const dev::DecBandJob *)d_bandjobs_, (const dev::DecTables *)d_tables_, d_errors_);
	if (nb18) {
		HIPCHK(hipMemcpyAsync(d_diffjobs_, dj, nd * sizeof(dev::DecDiffJob), hipMemcpyHostToDevice, st));
		dev::k_dec_bands_par_ll<<<nb18, dev::DECP_LL_THREADS, 0, st>>>((const dev::DecBandJob *)d_bandjobs_ + nb17, (const dev::DecTables *)d_tables18_, d_errors_);
		dev::k_dec_undiff<<<dim3((unsigned)nd, dev::DXU_SPLIT), dev::DXU_THREADS, 0, st>>>((const dev::DecDiffJob *)d_diffjobs_, d_errors_, 0);
	}
	dev::k_dec_lowpass<<<dim3(8, (unsigned)nl), 256, 0, st>>>((const dev::DecLowpassJob *)d_lowjobs_);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(h_errors_, d_errors_, sizeof(int), hipMemcpyDeviceToHost, st));
	return 0;
}

int GpuGroupEntropyDecoder::check() { return h_errors_ ? *h_errors_ : -1; }

} // namespace cfhd
