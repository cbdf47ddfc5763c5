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
// Switching fibers.  glibc's swapcontext() saves and restores the signal mask with a system call on every switch -- a barrier among 256 fibers is
// hundreds of switches --, so on x86-64 a switch is done by hand: push the callee-saved registers, exchange the stack pointers, pop, return.  Other
// architectures keep ucontext.
#if defined(__x86_64__)
#define HIPEMU_FAST_SWITCH 1
struct Context { void *sp; };
__attribute__((naked, noinline)) inline void switch_context(Context * /* save: rdi */, Context * /* load: rsi */)
{
	asm volatile("pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
	             "movq %rsp, (%rdi)\n\tmovq (%rsi), %rsp\n\t"
	             "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret");
}
inline void make_context(Context *c, char *stack, size_t bytes, void (*entry)())
{
	// a frame switch_context() can "return" into: six register slots, the entry point as the return address; the entry function then sees the
	// stack as after a call (rsp = 16 n + 8) and never returns (fiber_entry switches back to the scheduler)
	uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
	void **sp = (void **)(top - 64);
	for (int k = 0; k < 6; k++) sp[k] = nullptr;
	sp[6] = (void *)entry; sp[7] = nullptr;
	c->sp = sp;
}
#else
#define HIPEMU_FAST_SWITCH 0
struct Context { ucontext_t uc; };
inline void switch_context(Context *save, Context *load) { swapcontext(&save->uc, &load->uc); }
#endif
struct Fiber { Context ctx; char *stack; bool done; dim3 tid; unsigned lin; };
enum { kWave = 64, kMaxWaves = 16 };
struct Sched {
	Context main; Fiber *current; std::function<void()> *body;
	// counting barriers: a barrier completes when every fiber of its scope that is still running has arrived (finished fibers
	// drop out of the count), so scopes may execute different numbers of barriers (a wave that left early, wave-level exchanges)
	unsigned block_live, block_arrived, block_gen;
	unsigned wave_live[kMaxWaves], wave_arrived[kMaxWaves], wave_gen[kMaxWaves];
	uint64_t exchange[kWave * kMaxWaves];     // one slot per thread for the wave-level data exchanges below
};
inline Sched &sched() { static Sched s; return s; }
inline void yield()
{
	Sched &s = sched();
	Fiber *me = s.current;
	switch_context(&me->ctx, &s.main);
	threadIdx = me->tid;
}
inline void fiber_entry()
{
	Sched &s = sched();
	(*s.body)();
	s.current->done = true;
	s.block_live--; s.wave_live[s.current->lin / kWave]--;
	switch_context(&s.current->ctx, &s.main);
	__builtin_trap();                                    // (a finished fiber is never resumed)
}
// barrier among the running fibers of this thread's wave (the hardware executes a wave in lock step; fibers need the rendezvous)
inline void wave_sync()
{
	Sched &s = sched();
	const unsigned w = s.current->lin / kWave;
	const unsigned gen = s.wave_gen[w];
	s.wave_arrived[w]++;
	while (s.wave_gen[w] == gen) {
		if (s.wave_arrived[w] >= s.wave_live[w]) { s.wave_gen[w]++; s.wave_arrived[w] = 0; break; }
		yield();
	}
}
inline unsigned lane_id() { return sched().current->lin % kWave; }
template <typename T> inline T exchange(T v, int src_lane)      // value of `v` in lane src_lane of the same wave (own value if out of range)
{
	Sched &s = sched();
	const unsigned lin = s.current->lin, w = lin / kWave;
	uint64_t bits = 0; memcpy(&bits, &v, sizeof(T));
	s.exchange[lin] = bits;
	wave_sync();
	T r = v;
	if (src_lane >= 0 && src_lane < (int)kWave) { const uint64_t b = s.exchange[w * kWave + (unsigned)src_lane]; memcpy(&r, &b, sizeof(T)); }
	wave_sync();
	return r;
}
}

inline void __syncthreads()
{
	hipemu::Sched &s = hipemu::sched();
	const unsigned gen = s.block_gen;
	s.block_arrived++;
	while (s.block_gen == gen) {
		if (s.block_arrived >= s.block_live) { s.block_gen++; s.block_arrived = 0; break; }
		hipemu::yield();
	}
}
// wave-level intrinsics (wave64).  Lanes that already returned do not take part; their slot keeps its last value.
template <typename T> inline T __shfl(T v, int lane) { return hipemu::exchange(v, lane); }
template <typename T> inline T __shfl_up(T v, unsigned d) { const int l = (int)hipemu::lane_id(); return hipemu::exchange(v, l - (int)d >= 0 ? l - (int)d : -1); }
template <typename T> inline T __shfl_down(T v, unsigned d) { const int l = (int)hipemu::lane_id(); return hipemu::exchange(v, l + (int)d < (int)hipemu::kWave ? l + (int)d : -1); }
template <typename T> inline T __shfl_xor(T v, int m) { return hipemu::exchange(v, (int)hipemu::lane_id() ^ m); }
inline unsigned long long __ballot(int pred)
{
	hipemu::Sched &s = hipemu::sched();
	const unsigned lin = s.current->lin, w = lin / hipemu::kWave;
	s.exchange[lin] = pred ? 1u : 0u;
	hipemu::wave_sync();
	unsigned long long m = 0;
	for (unsigned l = 0; l < hipemu::kWave; l++) if (s.exchange[w * hipemu::kWave + l] & 1u) m |= 1ull << l;
	hipemu::wave_sync();
	return m;
}
using std::min; using std::max;
#define __popcll(x) __builtin_popcountll(x)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_mbcnt_lo(mask, base) (base)      // lanes below this one within `mask`: the kernels only use it with mask 0
template <typename T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }

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
	static std::vector<char *> stacks;                    // kept from launch to launch (launches never overlap: hip_runtime.h's launch_sync holds a lock, emu_kernels.cpp is single-threaded)
	while (stacks.size() < nthreads) stacks.push_back((char *)malloc(stack_bytes));
	for (unsigned t = 0; t < nthreads; t++) fibers[t].stack = stacks[t];
	blockDim = block; gridDim = grid;
	for (unsigned bz = 0; bz < grid.z; bz++)
		for (unsigned by = 0; by < grid.y; by++)
			for (unsigned bx = 0; bx < grid.x; bx++) {
				blockIdx = dim3(bx, by, bz);
				s.block_live = nthreads; s.block_arrived = 0; s.block_gen = 0;
				for (unsigned wv = 0; wv < kMaxWaves; wv++) { s.wave_live[wv] = 0; s.wave_arrived[wv] = 0; s.wave_gen[wv] = 0; }
				for (unsigned t = 0; t < nthreads; t++) s.wave_live[t / kWave]++;
				memset(s.exchange, 0, sizeof(s.exchange));
				for (unsigned t = 0; t < nthreads; t++) {
					Fiber &f = fibers[t];
					f.done = false; f.lin = t;
					f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
#if HIPEMU_FAST_SWITCH
					make_context(&f.ctx, f.stack, stack_bytes, fiber_entry);
#else
					getcontext(&f.ctx.uc);
					f.ctx.uc.uc_stack.ss_sp = f.stack; f.ctx.uc.uc_stack.ss_size = stack_bytes; f.ctx.uc.uc_link = &s.main.uc;
					makecontext(&f.ctx.uc, (void (*)())fiber_entry, 0);
#endif
				}
				for (bool any = true; any;) {
					any = false;
					for (unsigned t = 0; t < nthreads; t++) {
						Fiber &f = fibers[t];
						if (f.done) continue;
						s.current = &f; threadIdx = f.tid;
						switch_context(&s.main, &f.ctx);
						if (!f.done) any = true;
					}
				}
			}
}
}
