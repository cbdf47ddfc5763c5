// cfhd_batch.cpp -- device-resident batched round trip (extension API, cfhd_amd_batch_*), used by bench.py and by
// callers that keep frames in HBM: N frames -> forward kernels -> entropy coding -> N samples -> entropy decoding
// -> inverse kernels -> N frames in HBM.  One launch per stage covers the whole batch.
//
// Default: every stage on the GPU; the decoder reads the samples where the encoder left them in HBM (k_dec_parse) while their
// host copy travels beside it.  CFHD_AMD_HANDOFF=host sends the samples through the host parser and back;
// CFHD_AMD_ENTROPY=host keeps the run-length/VLC stage on host threads (the reference's own arrangement, Codec/encoder.c:5386 /
// decoder.c:19534), fed by one D2H copy of the quantized bands and followed by one H2D copy of the dequantized bands.
#include "../../include/cfhd_amd.h"
#include "cfhd_core.h"
#include "cfhd_bitstream.h"
#include "cfhd_device.h"
#include "cfhd_metadata.h"
#include "cfhd_params.h"
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>
#include <memory>

using namespace cfhd;

struct cfhd_amd_chunk { EncodeBatch enc; DecodeBatch dec; int first = 0, n = 0; };

struct cfhd_amd_batch {
	FramePlan plan;
	int n = 0, nthreads = 1, quality = 4, pixel_kind = PIX_YUY2;
	int color_format = 2, color_space = 2; bool progressive = true;
	bool decode = true;                // false: encode only (cfhd_amd_batch_create_ex mode 1; Bayer samples, which this library does not decode)
	// The batch is cut into chunks with their own HIP streams: while one chunk's entropy decode (latency bound: one lane per
	// band) is in flight, the next chunk's bandwidth-bound kernels and PCIe copies run beside it.
	std::vector<std::unique_ptr<cfhd_amd_chunk>> chunks;
	cfhd_amd_chunk &chunk_of(int i, int *local) { for (auto &c : chunks) if (i < c->first + c->n) { *local = i - c->first; return *c; } *local = 0; return *chunks[0]; }
	std::vector<std::vector<uint8_t>> samples;
	std::vector<size_t> sample_size;
	MetaState meta;                    // the metadata a synchronous encoder attaches on its own (SampleEncoder.cpp:744-939): frame i of the batch = the (steps * n + i + 1)-th CFHD_EncodeSample call
	std::vector<MetaBlock> frame_meta;  // per frame: the global block as it stood when the frame was "submitted"
	uint32_t steps = 0;
	bool gpu_entropy = true;
	bool device_handoff = true;        // the decoder reads the samples where the encoder left them in HBM (k_dec_parse); the host copy arrives beside it
	double t_fwd = 0, t_entropy_enc = 0, t_entropy_dec = 0, t_inv = 0;   // wall seconds of the last round trip
	// the frame queue (cfhd_amd_batch_submit / _wait): the pass in flight runs on a thread of its own, on the batch's own HIP streams
	// (cfhd_amd_batch_submit / _wait).  The default arrangement -- everything on the GPU, one chunk -- needs no thread: submit queues the whole pass on the batch's streams
	// (the decoder's stream waits for the encoder's events), wait fetches the sizes, queues the copy of the samples and waits for both streams.  The other arrangements
	// (host hand-off, host entropy, several chunks) have host work in the middle of a pass: those run the blocking pass on a thread of their own.
	std::thread worker; bool in_flight = false, queued = false; long long pending = -1;
	// cfhd_amd_batch_submit_host: the pass in flight takes its frames from / leaves its pictures in the caller's memory
	const uint8_t *host_in = nullptr; size_t host_in_stride = 0; int host_in_pitch = 0;
	uint8_t *host_out = nullptr; size_t host_out_stride = 0; int host_out_pitch = 0;
	double t_launch0 = 0, t_launched = 0;
	std::vector<void *> shared_streams;      // streams several objects of the batch share (StreamScope): released behind the objects
	~cfhd_amd_batch() { if (worker.joinable()) worker.join(); chunks.clear(); for (void *s : shared_streams) device_stream_release(s); }
};

namespace {
template <typename F> void parallel_for(int n, int nthreads, F f)
{
	if (nthreads <= 1 || n <= 1) { for (int i = 0; i < n; i++) f(i); return; }
	std::atomic<int> next(0);
	std::vector<std::thread> pool;
	int t = nthreads < n ? nthreads : n;
	for (int k = 0; k < t; k++) pool.emplace_back([&] { for (int i; (i = next.fetch_add(1)) < n;) f(i); });
	for (auto &th : pool) th.join();
}
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The default pass (all stages on the GPU, samples handed over in HBM, one chunk) in two halves with no host wait inside either.
// batch_launch: every launch of the pass queued -- forward transform, entropy coder, dense copy of the samples + their sizes on the way to the host on the encoder's
// stream; parser, entropy decoder and inverse transform on the decoder's stream behind the encoder's events (headers final / payloads final).  Returns 0 or < 0.
int batch_launch(cfhd_amd_batch *b)
{
	cfhd_amd_chunk *c = b->chunks[0].get();
	b->t_launch0 = now();
	const uint32_t base_number = b->steps * (uint32_t)b->n;
	// every frame gets the metadata CFHD_EncodeSample would give it: GUID, encode date / time, a timecode and a unique frame number that advance per frame
	b->frame_meta.resize(b->n);
	for (int i = 0; i < b->n; i++) { b->meta.handle(); b->frame_meta[i] = b->meta.global; meta_remove_hidden(b->frame_meta[i]); }
	const uint32_t seed = 0xA511E9B3u * (b->steps + 1);
	// Encode-only passes in flight on one device take turns (cfhd_device.h stage_order_wait): this pass's forward transform starts behind the entropy coder of the pass
	// queued before it.  Passes that start together otherwise run in lock step -- all encode, then all copy their samples to the host, and the copies (the PCIe link) and
	// the kernels never overlap: byr4-2160p 18.9 k fps with turns, 11.9 k without; rg48-2160p 15.9 / 12.4 k.  Round-trip passes run free: their decode halves fill the
	// gaps, and turns -- for the encode halves alone or for both -- cost 6 % (54.6 vs 58.1 k fps at 1080p, 16.5 vs 17.9 k at 2160p, 51.4 vs 55.8 k at 1080i; 16 hardware
	// queues, three steps in flight: profiles/r05_o_*).  CFHD_AMD_QUEUE=ordered: (A/B) turns for both halves of every pass; =free: for none.
	static const int forced = [] { const char *e = getenv("CFHD_AMD_QUEUE"); return e && strcmp(e, "ordered") == 0 ? 2 : (e && strcmp(e, "free") == 0 ? 0 : -1); }();
	const int turns = forced >= 0 ? forced : (b->decode ? 0 : 1);
	const bool ordered = turns >= 1, ordered_decode = turns >= 2;
	if (ordered && stage_order_wait(c->enc.device(), 0, c->enc.stream())) return -2;
	// fed from the host: the frames' copies first, on the encoder's stream (passes in flight on other batches run beside them)
	if (b->host_in && c->enc.upload_frames(b->host_in, b->host_in_stride, b->host_in_pitch)) return -2;
	// the transform kernels start first: the host serialises the sample headers (0.5 ms per 256) while they run
	if (c->enc.launch_forward(false)) return -2;         // (nothing but the entropy stage reads these coefficients)
	for (int l = 0; l < c->n; l++) {
		SampleHeaderInfo h = { base_number + (uint32_t)(c->first + l) + 1, b->color_format, b->color_space, b->quality, b->progressive, b->frame_meta[c->first + l].data(), b->frame_meta[c->first + l].size(), nullptr, 0 };
		if (c->enc.entropy().set_frame_header(l, h)) return -6;
	}
	if (c->enc.entropy().launch()) return -2;
	if (ordered && stage_order_done(c->enc.device(), 0, c->enc.stream())) return -2;      // (in front of the copies to the host: the next pass's encode does not wait for PCIe)
	if (b->decode) {
		if (ordered_decode && stage_order_wait(c->enc.device(), 1, c->dec.stream())) return -5;
		// the parser only needs the headers and size fields (k_ent_layout): it runs beside k_ent_emit, the band decoder waits for the payloads
		c->dec.entropy().set_producer_events(c->enc.entropy().headers_event(), c->enc.entropy().samples_event());
		if (c->dec.entropy().set_samples_device(c->enc.entropy().device_sample(0), c->enc.entropy().sample_cap(), c->enc.entropy().device_sizes())) return -4;
		if (c->dec.launch_entropy() || c->dec.launch_inverse(seed + (uint32_t)c->first)) return -5;
		if (ordered_decode && stage_order_done(c->enc.device(), 1, c->dec.stream())) return -5;
		if (b->host_out && c->dec.download_frames(b->host_out, b->host_out_stride, b->host_out_pitch)) return -5;      // the pictures' copies behind the inverse transform, on the decoder's stream
	}
	if (c->enc.entropy().download_queue()) return -2;
	b->t_launched = now();
	return 0;
}
// batch_finish: the sizes arrive, the copy of the sample bytes is queued, both streams drain.  Returns the total number of sample bytes or < 0.
long long batch_finish(cfhd_amd_batch *b)
{
	cfhd_amd_chunk *c = b->chunks[0].get();
	if (c->enc.entropy().download_finish() || c->enc.wait()) return -2;
	for (int l = 0; l < c->n; l++) {
		size_t n = c->enc.entropy().sample_bytes(l); if (!n) return -3; b->sample_size[c->first + l] = n;
		if (c->enc.entropy().needs_peak_table(l)) return -8;      // a band with more peak values than the entropy stage's positions hold (two million): only CFHD_EncodeSample writes such a sample (host writer)
	}
	const double t_enc = now();
	if (b->decode) { if (c->dec.wait()) return -5; if (c->dec.entropy().check()) return -7; }
	if (b->decode && b->host_out) {                  // (pictures staged through pinned memory -- a plain buffer -- leave it on a few threads side by side; nothing to do for a registered one)
		std::atomic<int> bad(0);
		parallel_for(c->n, c->n > 8 ? 8 : 1, [&](int l) { if (c->dec.finish_frame(l, b->host_out + b->host_out_stride * (size_t)l, b->host_out_pitch)) bad.store(1); });
		if (bad.load()) return -5;
	}
	const double t4 = now();
	b->t_fwd = b->t_launched - b->t_launch0; b->t_entropy_enc = t_enc - b->t_launched; b->t_entropy_dec = 0; b->t_inv = t4 - t_enc;
	b->steps++;
	long long total = 0;
	for (int i = 0; i < b->n; i++) total += (long long)b->sample_size[i];
	return total;
}
}

extern "C" {
long long cfhd_amd_batch_wait(cfhd_amd_batch *b);

// width x height frames of `pixel_format` encoded as `encoded_format` with `encoding_flags` (the CFHD_PrepareToEncode arguments: YUY2 / 2vuy ->
// 4:2:2, optionally interlaced; RG48 -> RGB 4:4:4; b64a -> RGBA 4:4:4:4; BYR4 -> Bayer).  mode 0: encode + decode back to the same pixel
// format; mode 1: encode only (the only mode for BYR4).
cfhd_amd_batch *cfhd_amd_batch_create_ex(int width, int height, uint32_t pixel_format, int encoded_format, uint32_t encoding_flags, int quality, int nframes, int nthreads, int mode)
{
	CallerDevice caller_device;
	FrontEndParams fp;
	if (nframes < 1 || front_end_params(width, height, pixel_format, encoded_format, encoding_flags, quality, &fp)) return nullptr;
	// A batch encodes its frames in one launch with one set of quantizer tables.  Qualities whose tables follow the size of the previous sample
	// (FILMSCAN2/3, LOW..HIGH up to 1080p: encoder.c:3414 -> quantize.c:2869) need frame i's sample before frame i + 1 can start -- those
	// sequences go through CFHD_EncodeSample, which applies the feedback per frame; here they are refused instead of encoded differently.
	if (!fp.static_quantizer) return nullptr;
	const int kind = fp.pixel_kind;
	const bool yuv = kind == PIX_YUY2 || kind == PIX_2VUY;
	cfhd_amd_batch *b = new (std::nothrow) cfhd_amd_batch;
	if (!b) return nullptr;
	b->n = nframes; b->nthreads = nthreads > 0 ? nthreads : 1; b->quality = fp.quality; b->pixel_kind = kind;
	b->color_format = fp.color_format; b->color_space = fp.color_space; b->progressive = fp.progressive;
	b->decode = mode == 0;
	b->plan = fp.plan;
	if (b->decode && (kind == PIX_BYR4 || kind == PIX_BYR5)) { delete b; return nullptr; }
	const char *e = getenv("CFHD_AMD_ENTROPY");
	b->gpu_entropy = !(e && strcmp(e, "host") == 0);
	if (!b->gpu_entropy && (!yuv || !b->progressive || !b->decode)) { delete b; return nullptr; }      // the host-entropy arrangement is kept for the headline workload only
	const char *ho = getenv("CFHD_AMD_HANDOFF");
	b->device_handoff = b->gpu_entropy && !(ho && strcmp(ho, "host") == 0);
	const char *cs = getenv("CFHD_AMD_CHUNK");
	int chunk = cs ? atoi(cs) : 0;                     // 0 = whole batch in one chunk (measured fastest: launches are already batch-wide)
	if (chunk <= 0 || !b->gpu_entropy) chunk = nframes;
	const size_t cap = (size_t)width * height * fp.pixel_bytes + 65536;        // SampleEncoder.cpp:387
	// Streams of a pass.  The runtime deals its hardware queues (GPU_MAX_HW_QUEUES, 4 unless the application's environment says otherwise) to a process's streams in the
	// order of their creation, and streams on one queue take turns (cfhd_entropy_gpu.h device_stream_create).  Measured with four passes in flight (profiles/r06_d_*):
	//   4 queues:  one stream per pass 55.5 k fps, two (encoder | decoder) 57.9 k, three (transforms | level-1 count | decoder) 56.9 k   [round 5: four, one unused: 50.4-50.8 k]
	//   16 queues: 45.3 k / 59.8 k / 61.2 k
	// so a pass takes two streams unless the environment gives the process eight queues or more -- read, never set here.  CFHD_AMD_STREAMS=1|2|3 overrides (A/B).
	const char *sp = getenv("CFHD_AMD_STREAMS"), *hq = getenv("GPU_MAX_HW_QUEUES");
	const int streams = sp && atoi(sp) >= 1 && atoi(sp) <= 3 ? atoi(sp) : (hq && atoi(hq) >= 8 ? 3 : 2);
	struct Lean { Lean() { device_streams_lean(true); } ~Lean() { device_streams_lean(false); } } lean;      // (a batch creates only the streams it launches on)
	for (int first = 0; first < nframes; first += chunk) {
		std::unique_ptr<cfhd_amd_chunk> c(new cfhd_amd_chunk);
		c->first = first; c->n = nframes - first < chunk ? nframes - first : chunk;
		bool ok = true;
		auto encoder = [&] { return !c->enc.prepare(b->plan, c->n, true) && !(b->gpu_entropy && c->enc.prepare_entropy(cap)); };
		auto decoder = [&] { if (!b->decode) return true; c->dec.set_interlaced(!b->progressive); return !c->dec.prepare(b->plan, c->n, kind, true) && !(b->gpu_entropy && c->dec.prepare_entropy(cap)); };
		if (streams == 1) { StreamScope scope; ok = encoder() && decoder(); if (scope.stream()) b->shared_streams.push_back(scope.stream()); }
		else if (streams == 2) {
			// (CFHD_AMD_STREAM_ORDER=alt, A/B: every second batch of the process creates its decoder's stream first -- with four queues dealt in creation order the encoder of
			// one pass then shares its queue with the decoder of another instead of with another encoder)
			static std::atomic<unsigned> created(0);
			static const bool alternate = [] { const char *e = getenv("CFHD_AMD_STREAM_ORDER"); return e && strcmp(e, "alt") == 0; }();
			void *dec_stream = nullptr;
			if (alternate && b->decode && (created.fetch_add(1) & 1u)) { StreamScope scope; void *x = nullptr; if (device_stream_create(&x) == 0) dec_stream = x; }
			{ StreamScope scope; ok = encoder(); if (scope.stream()) b->shared_streams.push_back(scope.stream()); }
			if (ok) { StreamScope scope(dec_stream); ok = decoder(); if (scope.stream()) b->shared_streams.push_back(scope.stream()); }
			else if (dec_stream) b->shared_streams.push_back(dec_stream);
		} else ok = encoder() && decoder();
		b->chunks.push_back(std::move(c));                // (also when it failed: the batch's destructor releases what was prepared, then the scopes' streams)
		if (!ok) { delete b; return nullptr; }
	}
	b->samples.resize(nframes); b->sample_size.assign(nframes, 0);
	if (!b->gpu_entropy) for (auto &s : b->samples) s.resize(cap);
	return b;
}

cfhd_amd_batch *cfhd_amd_batch_create(int width, int height, uint32_t pixel_format, int quality, int nframes, int nthreads)
{
	return cfhd_amd_batch_create_ex(width, height, pixel_format == 0x32767579u /* '2vuy' */ ? pixel_format : 0x59555932u /* 'YUY2' */, 0, 0, quality, nframes, nthreads, 0);
}

void cfhd_amd_batch_destroy(cfhd_amd_batch *b) { CallerDevice caller_device; if (b && b->in_flight) (void)cfhd_amd_batch_wait(b); delete b; }

// Puts frame i into HBM (outside the timed region of the benchmark).
int cfhd_amd_batch_upload(cfhd_amd_batch *b, int i, const void *frame, int pitch)
{
	CallerDevice caller_device;
	if (!b || b->in_flight || i < 0 || i >= b->n) return -1;
	int l; cfhd_amd_chunk &c = b->chunk_of(i, &l);
	int rc = c.enc.upload_frame(l, frame, pitch);
	if (rc) return rc;
	return c.enc.wait();
}

// One step of the hot path over the whole batch.  Returns the total number of sample bytes, or < 0.
static long long cfhd_amd_batch_roundtrip_locked(cfhd_amd_batch *b);
// (A batch with a pass in flight -- cfhd_amd_batch_submit without its _wait -- refuses every entry point that would touch its streams, job tables or pinned buffers: -1.)
long long cfhd_amd_batch_roundtrip(cfhd_amd_batch *b)
{
	if (!b || b->in_flight) return -1;
	return cfhd_amd_batch_roundtrip_locked(b);
}
static long long cfhd_amd_batch_roundtrip_locked(cfhd_amd_batch *b)
{
	CallerDevice caller_device;
	const FramePlan &plan = b->plan;
	double t0 = now(), t1, t2, t3;
	std::atomic<int> bad(0);
	const uint32_t base_number = b->steps * (uint32_t)b->n;
	// every frame gets the metadata CFHD_EncodeSample would give it: GUID, encode date / time, a timecode and a unique frame number that advance per frame
	b->frame_meta.resize(b->n);
	bool meta_ready = false;
	auto prepare_meta = [&] { if (meta_ready) return; meta_ready = true; for (int i = 0; i < b->n; i++) { b->meta.handle(); b->frame_meta[i] = b->meta.global; meta_remove_hidden(b->frame_meta[i]); } };
	auto header = [&](int i) { SampleHeaderInfo h = { base_number + (uint32_t)i + 1, b->color_format, b->color_space, b->quality, b->progressive, b->frame_meta[i].data(), b->frame_meta[i].size(), nullptr, 0 }; return h; };
	const uint32_t seed = 0xA511E9B3u * (b->steps + 1);
	if (b->gpu_entropy && b->device_handoff && b->chunks.size() == 1) {
		const int rc = batch_launch(b);
		if (rc) return rc;
		return batch_finish(b);
	} else if (b->gpu_entropy && b->device_handoff) {
		// Samples stay in HBM between the encoder and the decoder: the decoder's stream waits for the encoder's kernels, parses
		// the samples on the GPU and decodes, while the finished samples travel to the host (the encoder's product) on the
		// encoder's stream beside it.
		// Every chunk is driven by its own host thread from its first launch to its last wait, on its own streams.  (Queued from one thread, four
		// chunks of 128 ran at 36.5 k fps where one chunk of 512 runs at 46 k: the chunks did not overlap.  Four threads with 128 frames each: +10 %
		// over the single chunk -- the VALU-bound entropy kernels of one chunk run beside the HBM-bound transforms of another.  gpurun_out/r03i, r03j.)
		prepare_meta();
		std::atomic<int> err(0);
		std::vector<double> t_sub(b->chunks.size(), 0.0), t_enc(b->chunks.size(), 0.0);
		parallel_for((int)b->chunks.size(), (int)b->chunks.size(), [&](int k) {
			cfhd_amd_chunk *c = b->chunks[k].get();
			auto fail = [&](int code) { int z = 0; err.compare_exchange_strong(z, code); };
			// the transform kernels start first: the host serialises the sample headers (0.5 ms per 256) while they run
			if (c->enc.launch_forward(false)) return fail(-2);      // (nothing but the entropy stage reads these coefficients)
			for (int l = 0; l < c->n; l++) if (c->enc.entropy().set_frame_header(l, header(c->first + l))) return fail(-6);
			if (c->enc.entropy().launch()) return fail(-2);
			if (b->decode) {
				// the parser only needs the headers and size fields (k_ent_layout): it runs beside k_ent_emit, the band decoder waits for the payloads
				c->dec.entropy().set_producer_events(c->enc.entropy().headers_event(), c->enc.entropy().samples_event());
				if (c->dec.entropy().set_samples_device(c->enc.entropy().device_sample(0), c->enc.entropy().sample_cap(), c->enc.entropy().device_sizes())) return fail(-4);
				if (c->dec.launch_entropy() || c->dec.launch_inverse(seed + (uint32_t)c->first)) return fail(-5);
			}
			t_sub[k] = now();
			if (c->enc.entropy().download() || c->enc.wait()) return fail(-2);
			for (int l = 0; l < c->n; l++) {
				size_t n = c->enc.entropy().sample_bytes(l); if (!n) return fail(-3); b->sample_size[c->first + l] = n;
				if (c->enc.entropy().needs_peak_table(l)) return fail(-8);      // a band with more peak values than the entropy stage's positions hold (two million): only CFHD_EncodeSample writes such a sample (host writer)
			}
			t_enc[k] = now();
			if (b->decode) { if (c->dec.wait()) return fail(-5); if (c->dec.entropy().check()) return fail(-7); }
		});
		if (err.load()) return err.load();
		t1 = t0; t2 = t0;
		for (size_t k = 0; k < b->chunks.size(); k++) { if (t_sub[k] > t1) t1 = t_sub[k]; if (t_enc[k] > t2) t2 = t_enc[k]; }
		if (t2 < t1) t2 = t1;
		t3 = t2;
	} else if (b->gpu_entropy) {
		// 1. every chunk: forward transform + entropy coding, queued on the chunk's own stream
		prepare_meta();
		for (auto &c : b->chunks) {
			for (int l = 0; l < c->n; l++) if (c->enc.entropy().set_frame_header(l, header(c->first + l))) return -6;
			if (c->enc.launch_forward(false) || c->enc.entropy().launch()) return -2;
		}
		t1 = now();
		// 2. as each chunk's samples arrive on the host (the encoder's product), hand them to the decoder and queue its work
		double wait_s = 0, stage_s = 0;
		for (auto &c : b->chunks) {
			double a = now();
			if (c->enc.entropy().download() || c->enc.wait()) return -2;
			for (int l = 0; l < c->n; l++) {
				size_t n = c->enc.entropy().sample_bytes(l); if (!n) return -3; b->sample_size[c->first + l] = n;
				if (c->enc.entropy().needs_peak_table(l)) return -8;
			}
			double m = now();
			if (!b->decode) { wait_s += m - a; continue; }
			cfhd_amd_chunk *cp = c.get();
			parallel_for(c->n, b->nthreads < 16 ? b->nthreads : 16, [&, cp](int l) {
				if (cp->dec.entropy().set_sample_host(l, cp->enc.entropy().host_sample(l), b->sample_size[cp->first + l])) bad++; });
			if (bad) return -4;
			if (c->dec.launch_entropy() || c->dec.launch_inverse(seed + (uint32_t)c->first)) return -5;
			wait_s += m - a; stage_s += now() - m;
		}
		t2 = t1 + wait_s; t3 = t2 + stage_s;
		// 3. drain
		if (b->decode) for (auto &c : b->chunks) { if (c->dec.wait()) return -5; if (c->dec.entropy().check()) return -7; }
	} else {
		cfhd_amd_chunk &c = *b->chunks[0];
		prepare_meta();
		if (c.enc.launch_forward() || c.enc.download_coeffs() || c.enc.wait()) return -2;
		t1 = now();
		parallel_for(b->n, b->nthreads, [&](int i) {
			SampleHeaderInfo hdr = header(i);
			BandSource src; src.coeffs = c.enc.host_coeffs(i);
			size_t n = write_sample(plan, hdr, src, b->samples[i].data(), b->samples[i].size());
			if (!n) bad++;
			b->sample_size[i] = n;
		});
		t2 = now();
		if (bad) return -3;
		parallel_for(b->n, b->nthreads, [&](int i) {
			const uint8_t *s = b->samples[i].data();
			ParsedSample ps;
			if (parse_sample(s, b->sample_size[i], &ps) != 0) { bad++; return; }
			c.dec.clear_host_coeffs(i);
			int16_t *coeffs = c.dec.host_coeffs(i);
			for (int ch = 0; ch < plan.num_channels; ch++) {
				const ParsedBand &lp = ps.lowpass[ch];
				const BandDesc &ll = plan.ch[ch].band[2][0];
				const int bias = lowpass_bias(plan.precision, ll.width, b->pixel_kind, ch);
				for (int r = 0; r < ll.height; r++) {
					const uint8_t *p = s + lp.offset + (size_t)r * ll.width * 2;
					int16_t *dst = coeffs + ll.offset + (size_t)r * ll.pitch;
					for (int x = 0; x < ll.width; x++) { int v = (int16_t)((p[2 * x] << 8) | p[2 * x + 1]); v += bias; dst[x] = (int16_t)(v > 0x7fff ? 0x7fff : v); }
				}
				for (int lv = 0; lv < kNumLevels; lv++)
					for (int k = 1; k < 4; k++) {
						const ParsedBand &pb = ps.high[ch][lv][k];
						const BandDesc &bd = plan.ch[ch].band[lv][k];
						if (!pb.present || vlc_decode_band(s + pb.offset, pb.bytes, bd.width, bd.height, bd.pitch, pb.quant, pb.codebook, coeffs + bd.offset)) { bad++; return; }
					}
			}
		});
		t3 = now();
		if (bad) return -4;
		if (c.dec.upload_coeffs() || c.dec.launch_inverse(seed) || c.dec.wait()) return -5;
	}
	double t4 = now();
	b->t_fwd = t1 - t0; b->t_entropy_enc = t2 - t1; b->t_entropy_dec = t3 - t2; b->t_inv = t4 - t3;
	b->steps++;
	long long total = 0;
	for (int i = 0; i < b->n; i++) total += (long long)b->sample_size[i];
	return total;
}

// The streaming form (north_star: "the EncoderPool / AsyncEncoder thread pool is replaced by a HIP-stream frame queue"; semantics of the reference's queue:
// EncoderSDK/EncoderPool.cpp:239-380 -- submit returns at once, results are collected in submission order).  A batch object is one slot of the queue: submit starts
// its pass (all launches asynchronous on the batch's own streams, driven by a thread of its own), wait collects it.  Several batch objects in flight overlap: while
// one pass is in its instruction-bound entropy kernels another's bandwidth-bound transforms run beside them, which a single synchronous pass cannot have
// (tools/dual_batch.py).  The caller keeps FIFO order by waiting in the order it submitted.
int cfhd_amd_batch_submit(cfhd_amd_batch *b)
{
	CallerDevice caller_device;
	if (!b || b->in_flight) return -1;
	b->pending = -1;
	// CFHD_AMD_QUEUE=thread: (A/B) the blocking pass on a thread of its own for the default arrangement too, as in round 4.  (Also measured: the launches of a queued
	// pass -- 2-3 ms of host time with the serialising of 512 sample headers -- on a short-lived thread instead of the caller's: no difference on any line, profiles/r05_q_*.)
	static const bool threaded = [] { const char *e = getenv("CFHD_AMD_QUEUE"); return e && strcmp(e, "thread") == 0; }();
	if (b->gpu_entropy && b->device_handoff && b->chunks.size() == 1 && !threaded) {
		b->chunks[0]->enc.entropy().set_speculative_download(true);
		const int rc = batch_launch(b);                   // the whole pass is on the batch's streams when this returns; nothing waits
		if (rc) {                                         // a launch that failed midway: drain what it queued, leave the batch idle (advisor, round 5)
			(void)b->chunks[0]->enc.wait(); if (b->decode) (void)b->chunks[0]->dec.wait();
			b->chunks[0]->enc.entropy().set_speculative_download(false);
			return rc;
		}
		b->in_flight = true; b->queued = true;
		return 0;
	}
	b->in_flight = true; b->queued = false;
	b->worker = std::thread([b] { b->pending = cfhd_amd_batch_roundtrip_locked(b); });
	return 0;
}

// The frame queue fed from host memory (what EncoderSDK/EncoderPool.cpp:239-295 is to the reference: frames in, samples out, nothing resident): the pass copies its n frames
// from `frames` (frame i at frames + i * frame_stride, rows `pitch` bytes apart) into HBM on its own stream, runs, and copies the n decoded pictures to `pictures` (likewise;
// null: none) -- all queued behind one another, so that with several batches in flight the copies of one pass run beside the kernels of the others.  Both buffers are
// borrowed until cfhd_amd_batch_wait returns; buffers registered with cfhd_amd_register_host_buffer are DMA sources / targets as they are (one copy each way when the frames
// lie back to back at the batch's own pitch), plain buffers are staged through pinned memory by the calling thread.  The samples arrive as for cfhd_amd_batch_submit.
int cfhd_amd_batch_submit_host(cfhd_amd_batch *b, const void *frames, size_t frame_stride, int pitch, void *pictures, size_t picture_stride, int picture_pitch)
{
	CallerDevice caller_device;
	if (!b || b->in_flight || !frames || pitch <= 0 || (pictures && (!b->decode || picture_pitch <= 0))) return -1;
	if (!(b->gpu_entropy && b->device_handoff && b->chunks.size() == 1)) return -9;      // (the arrangements with host work in the middle of a pass are not queued: cfhd_amd_batch_upload + _submit)
	b->pending = -1;
	b->host_in = (const uint8_t *)frames; b->host_in_stride = frame_stride; b->host_in_pitch = pitch;
	b->host_out = (uint8_t *)pictures; b->host_out_stride = picture_stride; b->host_out_pitch = picture_pitch;
	b->chunks[0]->enc.entropy().set_speculative_download(true);
	const int rc = batch_launch(b);
	if (rc) { b->host_in = nullptr; b->host_out = nullptr; (void)b->chunks[0]->enc.wait(); if (b->decode) (void)b->chunks[0]->dec.wait(); b->chunks[0]->enc.entropy().set_speculative_download(false); return rc; }
	b->in_flight = true; b->queued = true;
	return 0;
}

long long cfhd_amd_batch_wait(cfhd_amd_batch *b)
{
	CallerDevice caller_device;
	if (!b || !b->in_flight) return -1;
	if (b->queued) {
		b->pending = batch_finish(b);
		if (b->pending < 0) { cfhd_amd_chunk *c = b->chunks[0].get(); (void)c->enc.wait(); if (b->decode) (void)c->dec.wait(); }      // an error exit must not leave work on the batch's streams (advisor, round 5)
		b->host_in = nullptr; b->host_out = nullptr;
	}
	else b->worker.join();
	b->in_flight = false; b->queued = false;
	return b->pending;
}

// which: 0..2 forward level launches (0 = k_fwd_yuv422), 3..5 inverse level launches (3 = k_inv_yuv422), 6 forward total, 7 inverse total,
// 8..11 k_ent_count / k_ent_scan / k_ent_layout / k_ent_emit, 12..14 k_dec_parse / band decoder (all its kernels) / k_dec_lowpass,
// 15..17 k_dec_index / k_dec_chain / k_dec_tiles, 18 the level-1 part of k_ent_count on its own stream (then 8 is the rest of it; 0 when the count is one
// launch), 19 k_dec_plan (the single workgroup that numbers the chunks in front of k_dec_index) (ms, HIP events on the launch streams)
float cfhd_amd_batch_kernel_ms(cfhd_amd_batch *b, int which)
{
	if (!b || b->in_flight) return 0;
	float ms = 0;                                        // summed over the chunks (each chunk times its own launches with HIP events on its stream)
	for (auto &c : b->chunks) {
		if (which >= 3 && which != 6 && which != 18 && !(which >= 8 && which < 12) && !b->decode) continue;
		if (which < 3) ms += c->enc.last_level_ms(which);
		else if (which < 6) ms += c->dec.last_level_ms(which - 3);
		else if (which >= 8 && which < 12) ms += b->gpu_entropy ? c->enc.entropy().kernel_ms(which - 8) : 0.0f;
		else if (which >= 12 && which < 18) ms += b->gpu_entropy ? c->dec.entropy().kernel_ms(which - 12) : 0.0f;
		else if (which == 18) ms += b->gpu_entropy ? c->enc.entropy().kernel_ms(4) : 0.0f;
		else if (which == 19) ms += b->gpu_entropy ? c->dec.entropy().kernel_ms(6) : 0.0f;
		else ms += which == 6 ? c->enc.last_kernel_ms() : c->dec.last_kernel_ms();
	}
	return ms;
}

// which as in cfhd_amd_batch_kernel_ms, 0..5: the name of the transform kernel behind that number (the shape depends on geometry and batch size)
const char *cfhd_amd_batch_kernel_name(cfhd_amd_batch *b, int which)
{
	if (!b || which < 0 || which > 5 || b->chunks.empty() || (which >= 3 && !b->decode)) return "";
	return which < 3 ? b->chunks[0]->enc.level_kernel(which) : b->chunks[0]->dec.level_kernel(which - 3);
}

// which: 0 forward (kernels + D2H), 1 host entropy encode + syntax, 2 host parse + entropy decode, 3 H2D + inverse kernels (wall seconds)
double cfhd_amd_batch_stage_seconds(cfhd_amd_batch *b, int which)
{
	if (!b) return 0;
	switch (which) { case 0: return b->t_fwd; case 1: return b->t_entropy_enc; case 2: return b->t_entropy_dec; default: return b->t_inv; }
}

// CFHD_AMD_DX_STATS=1: convergence counters of the chunk-indexed entropy decoder, summed over the chunks (16 words; GpuEntropyDecoder::stats)
int cfhd_amd_batch_dx_stats(cfhd_amd_batch *b, uint32_t *out)
{
	if (!b || b->in_flight || !out) return -1;
	for (int k = 0; k < 16; k++) out[k] = 0;
	int rc = -1;
	if (b->decode) for (auto &c : b->chunks) { uint32_t s[16]; if (c->dec.entropy().stats(s) == 0) { rc = 0; for (int k = 0; k < 16; k++) out[k] = k == 2 ? (s[k] > out[k] ? s[k] : out[k]) : out[k] + s[k]; } }
	return rc;
}

int cfhd_amd_batch_get_sample(cfhd_amd_batch *b, int i, const void **data, size_t *size)
{
	if (!b || b->in_flight || i < 0 || i >= b->n) return -1;
	int l; cfhd_amd_chunk &c = b->chunk_of(i, &l);
	*data = b->gpu_entropy ? (const void *)c.enc.entropy().host_sample(l) : (const void *)b->samples[i].data(); *size = b->sample_size[i];
	return 0;
}

int cfhd_amd_batch_download_output(cfhd_amd_batch *b, int i, void *out, int pitch)
{
	CallerDevice caller_device;
	if (!b || b->in_flight || i < 0 || i >= b->n || !b->decode) return -1;
	int l; cfhd_amd_chunk &c = b->chunk_of(i, &l);
	if (c.dec.download_frame(l, out, pitch) || c.dec.wait()) return -2;
	return c.dec.finish_frame(l, out, pitch);
}

} // extern "C"
