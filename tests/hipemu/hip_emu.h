// tests/hipemu/hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny CPU stand-in for the subset of the HIP device model our kernels use, so that the *same kernel
// source* (cineform-sdk_amd/csrc/cfhd_kernels.h) can be executed block by block on the host in the
// `-m "not gpu"` test suite: one fiber per GPU thread of a workgroup, __syncthreads() as a scheduling point,
// `__shared__` mapped to block-shared static storage (workgroups run one after another).
// It exists because this container has no GPU; it is never linked into the product library and is not a
// fallback path -- the product fails loudly without a HIP device.
#pragma once
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <ucontext.h>
#include <algorithm>
#include <vector>
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

// One fiber (ucontext) per GPU thread of the workgroup, all on the calling OS thread: __syncthreads() yields to the
// scheduler, which resumes the fibers round-robin, so every fiber reaches barrier k before any passes it -- the same
// guarantee the hardware gives -- and `__shared__` (block-shared static storage) needs no locking.  Workgroups run one
// after another.  Thread-uniform early returns are fine (a finished fiber is simply skipped).
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipemu {
struct Fiber { ucontext_t ctx; char *stack; bool done; dim3 tid; };
struct Sched { ucontext_t main; Fiber *current; std::function<void()> *body; };
inline Sched &sched() { static Sched s; return s; }
inline void fiber_entry()
{
	Sched &s = sched();
	(*s.body)();
	s.current->done = true;
	swapcontext(&s.current->ctx, &s.main);
}
}

inline void __syncthreads()
{
	hipemu::Sched &s = hipemu::sched();
	hipemu::Fiber *me = s.current;
	swapcontext(&me->ctx, &s.main);      // back to the scheduler; resumed after every other fiber reached this barrier
	threadIdx = me->tid;
}
using std::min; using std::max;

namespace hipemu {
template <typename F>
void launch(dim3 grid, dim3 block, F body_fn)
{
	const unsigned nthreads = block.x * block.y * block.z;
	const size_t stack_bytes = 256 * 1024;
	Sched &s = sched();
	std::function<void()> body = body_fn;
	s.body = &body;
	std::vector<Fiber> fibers(nthreads);
	for (auto &f : fibers) f.stack = (char *)malloc(stack_bytes);
	blockDim = block; gridDim = grid;
	for (unsigned bz = 0; bz < grid.z; bz++)
		for (unsigned by = 0; by < grid.y; by++)
			for (unsigned bx = 0; bx < grid.x; bx++) {
				blockIdx = dim3(bx, by, bz);
				for (unsigned t = 0; t < nthreads; t++) {
					Fiber &f = fibers[t];
					f.done = false;
					f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
					getcontext(&f.ctx);
					f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = stack_bytes; f.ctx.uc_link = &s.main;
					makecontext(&f.ctx, (void (*)())fiber_entry, 0);
				}
				for (bool any = true; any;) {
					any = false;
					for (unsigned t = 0; t < nthreads; t++) {
						Fiber &f = fibers[t];
						if (f.done) continue;
						s.current = &f; threadIdx = f.tid;
						swapcontext(&s.main, &f.ctx);
						if (!f.done) any = true;
					}
				}
			}
	for (auto &f : fibers) free(f.stack);
}
}
