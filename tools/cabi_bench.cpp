// tools/cabi_bench.cpp -- what a C/C++ application sees through the reference's own C ABI (CFHD_*), host buffers in and out, PCIe inclusive.
// Built against include/cfhd_amd.h + libcfhd_amd.so by __graft_entry__.build(); run by bench.py after its timed region (config.c_abi_fps).
//   cabi_bench <width> <height> <frames.yuy2> <nframes> <seconds> <registered 0|1> <decoder threads> <pool workers>
// Prints one JSON object: synchronous encode / decode, N decoder handles on N threads, the asynchronous encoder pool, and the round trip
// (the pool encoding while N decoder threads decode its samples; frames counted when decoded).
#include "../include/cfhd_amd.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(x) do { int rc_ = (int)(x); if (rc_ != 0) { fprintf(stderr, "cabi_bench: %s -> %d (line %d)\n", #x, rc_, __LINE__); exit(2); } } while (0)
static const CFHD_PixelFormat YUY2 = (CFHD_PixelFormat)0x59555932;

int main(int argc, char **argv)
{
	if (argc < 9) { fprintf(stderr, "usage: cabi_bench W H frames.yuy2 nframes seconds registered decoders workers\n"); return 1; }
	const int W = atoi(argv[1]), H = atoi(argv[2]), nfr = atoi(argv[4]), registered = atoi(argv[6]), handles = atoi(argv[7]), workers = atoi(argv[8]);
	const double seconds = atof(argv[5]);
	const size_t frame_bytes = (size_t)W * 2 * H; const int pitch = W * 2;
	std::vector<std::vector<uint8_t>> frames(nfr, std::vector<uint8_t>(frame_bytes));
	FILE *f = fopen(argv[3], "rb");
	if (!f) { perror(argv[3]); return 1; }
	for (auto &fr : frames) if (fread(fr.data(), 1, frame_bytes, f) != frame_bytes) { fprintf(stderr, "short read\n"); return 1; }
	fclose(f);
	if (registered) for (auto &fr : frames) CHECK(cfhd_amd_register_host_buffer(fr.data(), fr.size()));
	auto new_output = [&] { uint8_t *p = (uint8_t *)aligned_alloc(4096, (frame_bytes + 4095) & ~(size_t)4095); memset(p, 0, frame_bytes); if (registered) CHECK(cfhd_amd_register_host_buffer(p, frame_bytes)); return p; };
	auto drop_output = [&](uint8_t *p) { if (registered) cfhd_amd_unregister_host_buffer(p); free(p); };

	// --- synchronous encoder; keep its samples for the decoders
	std::vector<std::vector<uint8_t>> samples;
	double sync_enc;
	{
		CFHD_EncoderRef enc; CHECK(CFHD_OpenEncoder(&enc, nullptr));
		CHECK(CFHD_PrepareToEncode(enc, W, H, YUY2, (CFHD_EncodedFormat)0, 0, (CFHD_EncodingQuality)4));
		for (int i = 0; i < nfr; i++) {
			CHECK(CFHD_EncodeSample(enc, frames[i].data(), pitch));
			void *p; size_t n; CHECK(CFHD_GetSampleData(enc, &p, &n));
			samples.emplace_back((uint8_t *)p, (uint8_t *)p + n);
		}
		const double t0 = now(); long n = 0;
		while (now() - t0 < seconds) { CHECK(CFHD_EncodeSample(enc, frames[n % nfr].data(), pitch)); n++; }
		sync_enc = n / (now() - t0);
		CFHD_CloseEncoder(enc);
	}
	auto open_decoder = [&] {
		CFHD_DecoderRef dec; CHECK(CFHD_OpenDecoder(&dec, nullptr));
		int aw, ah; CFHD_PixelFormat af;
		CHECK(CFHD_PrepareToDecode(dec, 0, 0, YUY2, (CFHD_DecodedResolution)1, 0, samples[0].data(), 512, &aw, &ah, &af));
		return dec;
	};
	// --- N decoder handles on N threads
	auto decode_rate = [&](int nthreads) {
		std::atomic<bool> stop(false); std::atomic<long> done(0); std::atomic<int> warm(0);
		std::vector<std::thread> th;
		for (int k = 0; k < nthreads; k++) th.emplace_back([&, k] {
			CFHD_DecoderRef dec = open_decoder(); uint8_t *out = new_output();
			long i = k, mine = 0;
			while (!stop) { CHECK(CFHD_DecodeSample(dec, samples[i % nfr].data(), samples[i % nfr].size(), out, pitch)); i++; done++; if (++mine == 3) warm++; }
			CFHD_CloseDecoder(dec); drop_output(out);
		});
		while (warm < nthreads) std::this_thread::sleep_for(std::chrono::milliseconds(2));
		const long a = done; const double t0 = now();
		std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
		const long b = done; const double t1 = now();
		stop = true;
		for (auto &t : th) t.join();
		return (b - a) / (t1 - t0);
	};
	const double sync_dec = decode_rate(1), multi_dec = decode_rate(handles);

	// --- the asynchronous pool (one submitting thread, as the reference's TestCFHD drives it), optionally feeding decoder threads
	struct Queue { std::mutex m; std::condition_variable cv; std::deque<std::vector<uint8_t>> q; bool closed = false; size_t cap; } Q; Q.cap = 4 * (size_t)handles;
	auto pool_run = [&](bool feed) {
		CFHD_EncoderPoolRef pool; CHECK(CFHD_CreateEncoderPool(&pool, workers, 2 * workers, nullptr));
		CHECK(CFHD_PrepareEncoderPool(pool, W, H, YUY2, (CFHD_EncodedFormat)0, 0, (CFHD_EncodingQuality)4));
		CHECK(CFHD_StartEncoderPool(pool));
		long sent = 0, got = 0;
		auto collect = [&](bool wait) {
			uint32_t num; CFHD_SampleBufferRef sb;
			const int rc = wait ? CFHD_WaitForSample(pool, &num, &sb) : CFHD_TestForSample(pool, &num, &sb);
			if (rc != 0) return false;
			if (feed) {
				void *p; size_t n; CHECK(CFHD_GetEncodedSample(sb, &p, &n));
				std::unique_lock<std::mutex> lk(Q.m);
				Q.cv.wait(lk, [&] { return Q.q.size() < Q.cap; });
				Q.q.emplace_back((uint8_t *)p, (uint8_t *)p + n);
				Q.cv.notify_all();
			}
			CFHD_ReleaseSampleBuffer(pool, sb);
			return true;
		};
		// throughput, not set-up: the clock starts when the first 4 * workers frames are through (pool creation, the workers' first launches and allocations lie in front of it)
		double t0 = now(); long sent0 = 0; bool warm = false;
		while (now() - t0 < seconds || !warm) {
			CHECK(CFHD_EncodeAsyncSample(pool, (uint32_t)sent, frames[sent % nfr].data(), pitch, nullptr));
			sent++;
			while (collect(false)) got++;
			if (!warm && got >= 4 * workers) { warm = true; t0 = now(); sent0 = sent; }
		}
		const long sent1 = sent; const double dt = now() - t0;
		while (got < sent) if (collect(true)) got++;
		CFHD_ReleaseEncoderPool(pool);
		return (sent1 - sent0) / dt;
	};
	const double pool_enc = pool_run(false);
	std::atomic<long> decoded(0);
	std::vector<std::thread> th;
	for (int k = 0; k < handles; k++) th.emplace_back([&] {
		CFHD_DecoderRef dec = open_decoder(); uint8_t *out = new_output();
		for (;;) {
			std::vector<uint8_t> s;
			{
				std::unique_lock<std::mutex> lk(Q.m);
				Q.cv.wait(lk, [&] { return !Q.q.empty() || Q.closed; });
				if (Q.q.empty()) break;
				s = std::move(Q.q.front()); Q.q.pop_front();
				Q.cv.notify_all();
			}
			CHECK(CFHD_DecodeSample(dec, s.data(), s.size(), out, pitch));
			decoded++;
		}
		CFHD_CloseDecoder(dec); drop_output(out);
	});
	// (likewise: frames decoded per second from the moment the first 4 * handles frames have come out of the decoders to the moment pool_run returns -- the pool has stopped
	// submitting, collected every outstanding sample and been released: the window holds the steady state, the drain of the pool's queue and its teardown; advisor, round 5)
	std::atomic<bool> rt_done(false);
	double rt_t0 = 0, rt_t1 = 0; long rt_n0 = 0, rt_n1 = 0;
	std::thread watcher([&] {
		while (decoded < 4 * handles && !rt_done) std::this_thread::sleep_for(std::chrono::microseconds(200));
		rt_t0 = now(); rt_n0 = decoded;
	});
	pool_run(true);
	rt_t1 = now(); rt_n1 = decoded;
	rt_done = true;
	{ std::lock_guard<std::mutex> lk(Q.m); Q.closed = true; Q.cv.notify_all(); }
	for (auto &t : th) t.join();
	watcher.join();
	const double round_trip = rt_t1 > rt_t0 && rt_n1 > rt_n0 ? (rt_n1 - rt_n0) / (rt_t1 - rt_t0) : 0.0;
	if (registered) for (auto &fr : frames) cfhd_amd_unregister_host_buffer(fr.data());
	printf("{\"sync_encode_fps\": %.1f, \"sync_decode_fps\": %.1f, \"decode_fps_%d_handles\": %.1f, \"pool_encode_fps_%d_workers\": %.1f, \"round_trip_fps_pool%d_plus_%d_decoders\": %.1f}\n",
	       sync_enc, sync_dec, handles, multi_dec, workers, pool_enc, workers, handles, round_trip);
	return 0;
}
