// tests/hipemu/hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny CPU stand-in for the subset of the HIP device model our kernels use, so that the *same kernel
// source* (cineform-sdk_amd/csrc/cfhd_kernels.h) can be executed block by block on the host in the
// `-m "not gpu"` test suite: one OS thread per GPU thread of a workgroup, a real barrier for
// __syncthreads(), `__shared__` mapped to block-shared static storage (workgroups run one after another).
// It exists because this container has no GPU; it is never linked into the product library and is not a
// fallback path -- the product fails loudly without a HIP device.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <thread>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <functional>

#define CFHD_HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };

namespace hipemu {
struct Barrier {
	std::mutex m; std::condition_variable cv; unsigned count = 0, waiting = 0, generation = 0;
	void reset(unsigned n) { count = n; waiting = 0; }
	void wait() {
		std::unique_lock<std::mutex> lk(m);
		unsigned gen = generation;
		if (++waiting == count) { waiting = 0; generation++; cv.notify_all(); }
		else cv.wait(lk, [&] { return gen != generation; });
	}
};
inline Barrier &barrier() { static Barrier b; return b; }
}

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
inline void __syncthreads() { hipemu::barrier().wait(); }
using std::min; using std::max;

namespace hipemu {
// Runs kernel(args...) for every workgroup of the grid; workgroups are executed sequentially, the threads
// of one workgroup concurrently (so barriers and shared-memory hand-offs behave as on the device).
template <typename F>
void launch(dim3 grid, dim3 block, F body)
{
	const unsigned nthreads = block.x * block.y * block.z;
	Barrier &bar = barrier();
	Barrier block_done;     // separates workgroups: static __shared__ storage is reused
	bar.reset(nthreads);
	block_done.reset(nthreads);
	std::vector<std::thread> pool;
	for (unsigned t = 0; t < nthreads; t++) {
		pool.emplace_back([=, &block_done]() {
			threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
			blockDim = block; gridDim = grid;
			for (unsigned bz = 0; bz < grid.z; bz++)
				for (unsigned by = 0; by < grid.y; by++)
					for (unsigned bx = 0; bx < grid.x; bx++) {
						blockIdx = dim3(bx, by, bz);
						body();
						block_done.wait();
					}
		});
	}
	for (auto &th : pool) th.join();
}
}
