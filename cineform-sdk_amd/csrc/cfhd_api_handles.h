// cfhd_api_handles.h -- part of cfhd_api.cpp (one translation unit: the parts are #included there in this order, they share the handle types of an unnamed namespace).
// What the opaque references point to: Encoder, EncoderPool (workers = HIP-stream slots, FIFO completion), SampleBuffer, Decoder, DecMetadata; the dealing of units to devices.

// Handles carry their kind in their first word: the reference's own harness closes an encoder POOL with CFHD_CloseEncoder on its error path (Example/TestCFHD.cpp:1044)
enum : uint32_t { kEncoderMagic = 0x43464845u /* 'CFHE' */, kPoolMagic = 0x43464850u /* 'CFHP' */ };
struct Encoder {
	uint32_t magic = kEncoderMagic;
	EncodeParams params;
	MetaState meta;
	EncodeBatch batch;
	bool batch_ready = false;
	GopBatch gop_batch; bool gop_ready = false; uint32_t gop_calls = 0;      // two-frame groups: calls since CFHD_PrepareToEncode
	uint32_t frame_number = 0;
	std::vector<uint8_t> sample; size_t sample_size = 0;
	StageProfile prof;
};

// ---- devices ----
// Worker i of an encoder pool / the i-th decoder handle of the process on a node with several GPUs (cfhd_core.h unit_device): -1 = the process default.
int device_of_unit(int i)
{
	const char *pinned = getenv("CFHD_AMD_DEVICE");
	if (!pinned) pinned = getenv("LOCAL_RANK");
	return unit_device(i, device_count(), pinned, getenv("CFHD_AMD_POOL_DEVICES"));
}

// ---- async pool ----
struct SampleBuffer { std::vector<uint8_t> data; size_t size = 0; };

struct PoolJob {
	uint32_t frame_number = 0;
	const void *frame = nullptr; intptr_t pitch = 0;
	MetaBlock global, local;
	std::unique_ptr<SampleBuffer> sample;
	int error = 0;
	bool finished = false;
};

struct PoolWorker {
	int device = -1;                          // the GPU this worker's batch lives on (-1: the process default)
	EncodeBatch batch;
	EncodeParams params;                      // this worker's encoder state (quantizer feedback is per encoder: each CAsyncEncoder owns an ENCODER)
	uint32_t encoded = 0;                     // per-"encoder" frame counter (the reference numbers frames per CAsyncEncoder)
	std::thread thread;
	std::deque<std::shared_ptr<PoolJob>> inbox;
};

struct EncoderPool {
	uint32_t magic = kPoolMagic;
	int nworkers = 1, queue_len = 1;
	EncodeParams params;
	MetaState meta;                           // pool-wide metadata (attached with CFHD_AttachEncoderPoolMetadata)
	std::vector<std::unique_ptr<PoolWorker>> workers;
	std::mutex m; std::condition_variable cv_work, cv_done;
	std::deque<std::shared_ptr<PoolJob>> fifo;               // submission order
	bool started = false, stopping = false;
	int next_worker = 0;
	EncodeService *service = nullptr;          // set by CFHD_StartEncoderPool when the sequence has no frame-to-frame dependency

	void worker_loop(PoolWorker *w)
	{
		device_select(w->device);                  // (per thread: everything this worker launches goes to its own GPU)
		for (;;) {
			std::shared_ptr<PoolJob> job;
			{
				std::unique_lock<std::mutex> lk(m);
				cv_work.wait(lk, [&] { return stopping || !w->inbox.empty(); });
				if (w->inbox.empty()) return;
				job = w->inbox.front(); w->inbox.pop_front();
			}
			job->sample.reset(new SampleBuffer);
			job->sample->data.resize(sample_capacity(params));
			job->error = encode_one_gathered(w->batch, w->params, job->frame, (int)job->pitch, ++w->encoded, job->global, job->local,
			                                 job->sample->data.data(), job->sample->data.size(), &job->sample->size, service);
			{
				std::lock_guard<std::mutex> lk(m);
				job->finished = true;
			}
			cv_done.notify_all();
		}
	}
	void stop()
	{
		{ std::lock_guard<std::mutex> lk(m); stopping = true; }
		cv_work.notify_all();
		for (auto &w : workers) if (w->thread.joinable()) w->thread.join();
		started = false;
	}
};

// ---- decoder ----
struct Decoder {
	int device = -1;                          // the GPU this handle decodes on (dealt round robin when the process owns several, -1: the process default)
	ParsedSample header; bool prepared = false, half = false;
	uint32_t out_format = 0; int out_kind = 0;
	FramePlan plan;
	DecodeBatch batch; bool batch_ready = false;
	uint32_t frames_decoded = 0;
	StageProfile prof;
	struct DecodeService *service = nullptr; bool service_interlaced = false;
	// samples of two-frame groups (cfhd_gop.h): the group decodes both frames, the P-frame sample behind it hands out the second one
	bool gop = false, gop_ready = false, gop_second = false; GopPlan gplan; GopBatch gop_batch;
};

struct DecMetadata { std::vector<uint8_t> block; size_t cursor = 0; };


struct DecodeServiceKey {
	int width, height, display_height, encoded_format, precision, prescale[3], out_kind, device; bool half, interlaced;
	bool operator==(const DecodeServiceKey &o) const
	{
		return device == o.device && width == o.width && height == o.height && display_height == o.display_height && encoded_format == o.encoded_format && precision == o.precision &&
		       prescale[0] == o.prescale[0] && prescale[1] == o.prescale[1] && prescale[2] == o.prescale[2] && out_kind == o.out_kind && half == o.half && interlaced == o.interlaced;
	}
};
struct DecodeService : Gatherer<DecodeBatch> {
	DecodeServiceKey key;
	bool start(const FramePlan &plan, int out_kind, bool half, bool interlaced, int nslots)
	{
		slots = nslots;
		const size_t cap = (size_t)plan.width * plan.height * pixel_bytes_of(out_kind) + 65536;
		device = key.device;
		struct OnDevice { OnDevice(int d) { device_select(d); } ~OnDevice() { device_select(-1); } } on(device);      // the batches are prepared on the service's GPU
		for (Pass &x : g) {
			x.batch.set_interlaced(interlaced);
			if (x.batch.prepare(plan, slots, out_kind, true, half) || x.batch.prepare_entropy(cap) || !x.batch.entropy().chunk_indexed()) { for (Pass &y : g) y.batch.release(); return false; }
		}
		run_pass = [](Pass &x, int n, uint32_t launch) {
			x.batch.set_active(n);
			int rc = x.batch.launch_entropy();
			if (!rc) rc = x.batch.launch_inverse(0x2545F491u * launch);
			for (int i = 0; i < n && !rc; i++) rc = x.batch.download_frame(i, x.ptr[i], x.num[i]);
			if (!rc) rc = x.batch.wait(); else (void)x.batch.wait();
			if (!rc && x.batch.entropy().check()) rc = 1;      // a damaged sample somewhere in the pass: every caller decodes alone and gets its own verdict
			return rc;
		};
		start_workers();
		return true;
	}
	int decode(const uint8_t *sample, size_t size, void *out, int pitch)
	{
		return submit([&](DecodeBatch &b, int i, Pass &x) { x.ptr[i] = out; x.num[i] = pitch; return b.entropy().set_sample_host(i, sample, size); },
		              [&](DecodeBatch &b, int i) { b.finish_frame(i, out, pitch); });       // every caller copies its own frame out of the pinned staging
	}
};
struct DecodeServices {
	std::mutex m;
	std::vector<DecodeService *> list;         // never freed: the dispatcher threads and their HIP objects live as long as the process
	DecodeService *find(const DecodeServiceKey &key)
	{
		std::lock_guard<std::mutex> lk(m);
		for (DecodeService *s : list) if (s->key == key) return s;
		if (list.size() >= 8) return nullptr;          // (a service holds two batches in HBM and pinned memory for good: a process that decodes more geometries than this at once does without)
		DecodeService *s = new DecodeService; s->key = key; list.push_back(s);
		return s;
	}
};
DecodeServices &decode_services() { static DecodeServices *s = new DecodeServices; return *s; }
int decode_gather_slots() { static const int n = gather_slots("CFHD_AMD_DECODE_BATCH", 8); return n; }

void plan_from_sample(const ParsedSample &ps, int out_kind, FramePlan *plan, bool *ok)
{
	const int quad = ps.encoded_format == ENC_BAYER ? 2 : 1;      // (build_frame_plan takes the mosaic's size)
	*ok = build_frame_plan(plan, quad * ps.width, quad * ps.display_height, out_kind, ps.encoded_format);
	if (!*ok) return;
	plan->precision = ps.precision;
	// for outputs that convert YUV to RGB: 601 or 709 by the sample's colour space tag, always the computer-systems range -- probed on the reference decoder: a
	// sample encoded with CFHD_ENCODING_FLAGS_YUV_VSRGB decodes with the CG matrix all the same (26 dB against its source instead of 31)
	plan->color_matrix = (ps.color_space & 3) == 1 ? 2 : 0;
	if (ps.prescale_table) for (int i = 0; i < kNumLevels; i++) plan->prescale[i] = (ps.prescale_table >> (14 - 2 * i)) & 3;
	else { plan->prescale[0] = 0; plan->prescale[1] = ps.precision >= 10 ? 2 : 0; plan->prescale[2] = ps.precision == 12 ? 2 : 0; }
	if (plan->height != ps.height) *ok = false;
}
