// cfhd_api_gather.h -- part of cfhd_api.cpp (one translation unit: the parts are #included there in this order, they share the handle types of an unnamed namespace).
// Calls that overlap share launches: the gatherer behind CFHD_DecodeSample on several handles and behind the pool's workers, and the decoders-at-work signal.

// ---- calls that overlap share launches -----------------------------------------------------------------------------------------------
// The reference's decoder is synchronous per handle and its encoder pool runs one CPU encoder per thread; applications get throughput by
// working on several frames from several threads.  On the GPU one frame per launch sequence leaves the chip mostly idle (a dozen launches
// of kernels that see a single frame), so calls of the same geometry that are in flight at the same time are gathered: every caller stages
// its frame or sample into a slot of a shared batch, one of two dispatcher threads (one per batch, each with its own HIP stream) launches
// whatever has gathered as one multi-frame pass, and every caller copies its own result out.  A lone caller never comes here (the handle's
// own batch of one frame serves it).  CFHD_AMD_DECODE_BATCH=n (default 8) / CFHD_AMD_ENCODE_BATCH=n (default 0 = off) set the slots per batch.
template <class BatchT> struct Gatherer {
	struct Pass {
		BatchT batch;
		int claimed = 0, ready = 0, released = 0, state = 0 /* 0 collecting, 1 running, 2 done */, rc = 0;
		bool bad = false; uint32_t gen = 0;
		std::vector<void *> ptr; std::vector<int> num;      // per slot: what the pass needs from the caller (output buffer + pitch of a decode)
		std::thread worker;
	};
	int slots = 8, device = -1; bool ok = false, dead = false;      // device: the GPU the two batches live on (-1: the process default)
	std::mutex m; std::condition_variable cv_callers, cv_workers;
	Pass g[2];
	std::atomic<int> inflight{0};
	uint32_t launches = 0;
	int (*run_pass)(Pass &, int n, uint32_t launch) = nullptr;

	void start_workers()
	{
		for (Pass &x : g) { x.ptr.assign((size_t)slots, nullptr); x.num.assign((size_t)slots, 0); }
		for (int k = 0; k < 2; k++) g[k].worker = std::thread([this, k] { run(g[k]); });
		ok = true;
	}
	void run(Pass &x)
	{
		(void)device_select(device);                       // the device is selected per thread
		std::unique_lock<std::mutex> lk(m);
		for (;;) {
			cv_workers.wait(lk, [&] { return x.state == 0 && x.claimed > 0 && x.ready == x.claimed; });
			x.state = 1;
			const int n = x.claimed; const bool bad = x.bad; const uint32_t launch = ++launches;
			lk.unlock();
			const int rc = bad ? 0 : run_pass(x, n, launch);
			lk.lock();
			x.rc = rc; x.state = 2;
			cv_callers.notify_all();
		}
	}
	// stage(batch, slot, pass) != 0: this caller's input cannot go through a gathered pass (nobody of the pass is served here then);
	// finish(batch, slot) copies the caller's result out.  Returns 0 when served, 1 when the caller has to take its own path.
	template <class Stage, class Finish> int submit(Stage stage, Finish finish)
	{
		std::unique_lock<std::mutex> lk(m);
		Pass *x = nullptr;
		cv_callers.wait(lk, [&] {
			// join the pass that is gathering; else open one on a free batch
			for (Pass &c : g) if (c.state == 0 && c.claimed > 0 && c.claimed < slots) { x = &c; return true; }
			for (Pass &c : g) if (c.state == 0 && c.claimed == 0) { x = &c; return true; }
			return false;
		});
		const int i = x->claimed++; const uint32_t gen = x->gen;
		lk.unlock();
		const int staged = stage(x->batch, i, *x);            // beside the other callers: parse / copy into the slot's pinned memory
		lk.lock();
		if (staged) x->bad = true;
		x->ready++;
		cv_workers.notify_all();
		cv_callers.wait(lk, [&] { return x->state == 2 && x->gen == gen; });
		const bool failed = x->bad || x->rc != 0;
		lk.unlock();
		if (!failed) finish(x->batch, i);
		lk.lock();
		if (++x->released == x->claimed) { x->claimed = x->ready = x->released = 0; x->bad = false; x->rc = 0; x->state = 0; x->gen++; cv_callers.notify_all(); cv_workers.notify_all(); }
		return failed ? 1 : 0;
	}
};
int gather_slots(const char *env, int dflt) { const char *e = getenv(env); int v = e ? atoi(e) : dflt; return v < 0 ? 0 : (v > 64 ? 64 : v); }


// The encoder side: workers of a pool (or several pools) that encode at the same time.  Only where no frame depends on the previous one:
// qualities whose quantizer follows the size of the last sample (rate feedback, encode_one) keep one launch sequence per frame.
struct EncodeServiceKey {
	int width, height, pixel_kind, encoded_format, quality, color_space; uint32_t flags;
	int device;                                // the GPU the pool's workers were dealt (two pools on different GPUs do not share a service)
	bool operator==(const EncodeServiceKey &o) const { return memcmp(this, &o, sizeof(*this)) == 0; }
};
struct EncodeService : Gatherer<EncodeBatch> {
	EncodeServiceKey key;
	bool start(const EncodeParams &p, int nslots)
	{
		slots = nslots;
		device = key.device;
		struct OnDevice { OnDevice(int d) { device_select(d); } ~OnDevice() { device_select(-1); } } on(device);      // the shared batches live on the GPU of the workers they serve
		for (Pass &x : g) if (x.batch.prepare(p.plan, slots, true) || x.batch.prepare_entropy(sample_capacity(p))) { for (Pass &y : g) y.batch.release(); return false; }   // (a service that cannot be set up holds no HBM)
		run_pass = [](Pass &x, int n, uint32_t) {
			x.batch.set_active(n);
			int rc = x.batch.launch_forward();
			if (!rc) rc = x.batch.entropy().launch();
			if (!rc) rc = x.batch.entropy().download();
			if (!rc) rc = x.batch.wait(); else (void)x.batch.wait();
			for (int i = 0; i < n && !rc; i++) if (!x.batch.entropy().sample_bytes(i) || x.batch.entropy().needs_peak_table(i)) rc = 1;   // overflow / more peak values than the stage places: every caller takes its own path
			return rc;
		};
		start_workers();
		return true;
	}
	int encode(const SampleHeaderInfo &hdr, const void *frame, int pitch, uint8_t *out, size_t cap, size_t *size_out)
	{
		return submit([&](EncodeBatch &b, int i, Pass &) { int rc = b.upload_frame(i, frame, pitch); if (!rc) rc = b.entropy().set_frame_header(i, hdr); return rc; },
		              [&](EncodeBatch &b, int i) { const size_t n = b.entropy().sample_bytes(i); if (n <= cap) { memcpy(out, b.entropy().host_sample(i), n); *size_out = n; } else *size_out = 0; });
	}
};
struct EncodeServices {
	std::mutex m;
	std::vector<EncodeService *> list;         // never freed (see DecodeServices)
	EncodeService *find(const EncodeServiceKey &key)
	{
		std::lock_guard<std::mutex> lk(m);
		for (EncodeService *s : list) if (s->key == key) return s;
		if (list.size() >= 8) return nullptr;
		EncodeService *s = new EncodeService; s->key = key; list.push_back(s);
		return s;
	}
};
EncodeServices &encode_services() { static EncodeServices *s = new EncodeServices; return *s; }
// Pool workers gather their frames into shared passes WHILE DECODERS ARE AT WORK on the GPU, and only then.  Measured on one MI355X at 1080p (profiles/r05_e_*): a pool
// alone runs faster with every worker on its own stream (7.6 k against 6.8 k fps gathered: a shared pass keeps its callers in lock step), but a pool beside eight decoder
// threads -- the round trip through the C ABI -- runs at 2.4-2.9 k fps ungathered and 4.1-4.2 k gathered: three dozen launches per frame from sixteen threads queue up in
// the runtime, a pass of eight frames makes them a dozen.  CFHD_AMD_ENCODE_BATCH=n forces n slots whatever the decoders do (0: never gather).
// The signal is kept per GPU (advisor, round 5: decoders on another GPU pushed a pool on this one into lock-step passes): slot 0 = the process default device (-1), slot
// d + 1 = device d; a pool looks at the slot of the device its service lives on.
enum { kDecodeSignalSlots = 17 };
std::atomic<int> g_decodes_in_flight[kDecodeSignalSlots];
std::atomic<long long> g_last_decode_ns[kDecodeSignalSlots];
long long mono_ns() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec; }
// (a handle that was not dealt a device runs on the process default: the slot is that of the real device, so that a decoder's -1 and a pool's device number meet)
int decode_signal_slot(int device) { if (device < 0) device = device_current(); return device < 0 || device + 1 >= (int)kDecodeSignalSlots ? 0 : device + 1; }
struct DecodeInFlight {
	int slot;
	explicit DecodeInFlight(int device) : slot(decode_signal_slot(device)) { g_decodes_in_flight[slot].fetch_add(1); }
	~DecodeInFlight() { g_last_decode_ns[slot].store(mono_ns()); g_decodes_in_flight[slot].fetch_sub(1); }
};
bool encode_gather_forced() { static const bool f = getenv("CFHD_AMD_ENCODE_BATCH") != nullptr; return f; }
// a decode call running on that GPU, or one that ended there within the last 5 ms (CFHD_AMD_ENCODE_BATCH is the deterministic override: the tests use it)
bool decoders_at_work(int device) { const int k = decode_signal_slot(device); return g_decodes_in_flight[k].load() > 0 || mono_ns() - g_last_decode_ns[k].load() < 5000000ll; }
int encode_gather_slots() { return gather_slots("CFHD_AMD_ENCODE_BATCH", 8); }        // (read when a pool starts)
// true when the quantizer tables of a sequence never move: FILMSCAN1 (and anything above 1080p for LOW..HIGH) -- decided by asking the
// derivation itself whether a large previous sample would change them
bool quantizer_is_static(const EncodeParams &p)
{
	FramePlan probe = p.plan; QuantState st = p.qstate;
	st.lastgopbitcount = (int64_t)p.width * p.height * 64;               // an absurdly large previous sample
	derive_quantization(&probe, p.quality, p.progressive, 0.0f, &st);
	for (int c = 0; c < probe.num_channels; c++)
		for (int lv = 0; lv < kNumLevels; lv++)
			for (int b = 0; b < kNumBands; b++) if (probe.ch[c].band[lv][b].quant != p.plan.ch[c].band[lv][b].quant) return false;
	return true;
}

// encode_one for a caller that may share its launches with others encoding the same geometry right now
int encode_one_gathered(EncodeBatch &own, EncodeParams &p, const void *frame, int pitch, uint32_t frame_number,
                        MetaBlock global, MetaBlock local, uint8_t *out, size_t cap, size_t *size_out, EncodeService *svc)
{
	if (svc) {
		struct InFlight { std::atomic<int> &n; int before; InFlight(std::atomic<int> &c) : n(c), before(c.fetch_add(1)) {} ~InFlight() { n.fetch_sub(1); } } mark(svc->inflight);
		if (mark.before > 0 && (encode_gather_forced() || decoders_at_work(svc->key.device))) {
			bool usable;
			{
				std::lock_guard<std::mutex> lk(svc->m);
				if (!svc->ok && !svc->dead) { if (!svc->start(p, encode_gather_slots())) svc->dead = true; }
				usable = svc->ok;
			}
			MetaBlock g2 = global, l2 = local;
			meta_remove_hidden(g2); meta_remove_hidden(l2);
			SampleHeaderInfo hdr = { frame_number, p.pixel_format == FMT_RG30 ? 122 : color_format_of(p.pixel_kind), p.color_space, p.quality, p.progressive, g2.data(), g2.size(), l2.data(), l2.size() };
			uint32_t sz; unsigned char ty;
			const uint32_t VCHN = CFHD_FOURCC('V', 'C', 'H', 'N');
			const bool vchn = meta_find(g2.data(), g2.size(), VCHN, &sz, &ty) || meta_find(l2.data(), l2.size(), VCHN, &sz, &ty);       // rare syntax switch: encode_one knows it
			if (usable && !vchn && svc->encode(hdr, frame, pitch, out, cap, size_out) == 0 && *size_out) return ERR_OKAY;
		}
		return encode_one(own, p, frame, pitch, frame_number, global, local, out, cap, size_out);
	}
	return encode_one(own, p, frame, pitch, frame_number, global, local, out, cap, size_out);
}
