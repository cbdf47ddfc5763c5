// tests/hipemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A host-memory stand-in for the part of the HIP runtime the product's device layer calls (cineform-sdk_amd/csrc/cfhd_device.hip,
// cfhd_entropy_gpu.hip), so that the `-m "not gpu"` suite can run the WHOLE product library -- CFHD_* C ABI, batch front end, job builders,
// entropy drivers and the unmodified kernel source -- on the CPU (tests/_build/libcfhd_amd_hipemu.so, built by cfhd_testlib.product_emulated()).
// "Device" memory is host memory, streams and events are tokens, every copy and every kernel launch completes before the call returns
// (kernels run through hip_emu.h: one fiber per GPU thread, workgroups one after another, launches of different host threads serialised).
// It exists because the layer between the C ABI and the kernels -- the job tables -- is where bugs were found only on hardware; it is never
// linked into libcfhd_amd.so and is not a fallback: the product fails loudly without a HIP device.
#pragma once
#include "../hip_emu.h"
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <mutex>
#include <stdio.h>
#include <time.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct hipemuStream_ *hipStream_t;
typedef struct hipemuEvent_ *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipHostMallocPortable = 1, hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipHostRegisterPortable = 1, hipEventDisableTiming = 2, hipEventDefault = 0 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 };

// ---- devices -------------------------------------------------------------------------------------------------------------------------------------
// HIPEMU_DEVICES=n (default 1) emulated GPUs with separate heaps.  What a real node enforces and a single host heap would not is enforced here:
//   * a stream (and an event) belongs to the device that was current when it was created; launching or copying on it from a thread whose current device is
//     another one is an error (sticky: hipGetLastError reports it, the product's HIPCHK turns it into a failed call);
//   * "device" memory (hipMalloc) is mapped PROT_NONE while nothing runs on it: it is readable and writable only inside a kernel launch on its own device
//     and inside a copy whose stream (or, for synchronous copies, whose calling thread) is on its own device.  A kernel that is handed another device's
//     pointer, a copy queued on the wrong device and host code that dereferences a device pointer all end in a segmentation fault at the culprit.
// Pinned host memory (hipHostMalloc) is ordinary memory, as on the hardware.
#include <sys/mman.h>
#include <map>
#include <unistd.h>
struct hipemuStream_ { int device; };
struct hipemuEvent_ { int device; };
namespace hipemu {
struct Alloc { size_t bytes; int device; };
struct DeviceState {
	std::recursive_mutex m;                         // allocations, protection changes, launches and copies: one at a time, process wide
	std::map<uintptr_t, Alloc> allocs;              // device memory by address
	int ndevices = 1; hipError_t sticky = 0; int open_depth = 0, open_device = -1;
	DeviceState() { const char *e = getenv("HIPEMU_DEVICES"); ndevices = e && atoi(e) > 0 ? atoi(e) : 1; }
};
inline DeviceState &devices() { static DeviceState *s = new DeviceState; return *s; }
inline int &current_device() { static thread_local int d = 0; return d; }
inline hipError_t violation(const char *what, int a, int b)
{
	fprintf(stderr, "[hipemu] cross-device use: %s (device %d vs %d)\n", what, a, b);
	devices().sticky = 1;
	return 1;
}
// Opens the heap of `device` (and closes it again): nested for a launch that copies.  Called with the state's mutex held.
struct HeapOpen {
	DeviceState &st; bool outer;
	HeapOpen(int device) : st(devices()), outer(false)
	{
		st.m.lock();
		if (st.open_depth++ == 0) { outer = true; st.open_device = device; for (auto &a : st.allocs) if (a.second.device == device) mprotect((void *)a.first, a.second.bytes, PROT_READ | PROT_WRITE); }
	}
	~HeapOpen()
	{
		if (--st.open_depth == 0) { for (auto &a : st.allocs) if (a.second.device == st.open_device) mprotect((void *)a.first, a.second.bytes, PROT_NONE); st.open_device = -1; }
		st.m.unlock();
	}
};
inline int device_of_pointer(const void *p)          // -1: not device memory (host memory, pinned or not)
{
	DeviceState &st = devices();
	std::lock_guard<std::recursive_mutex> lk(st.m);
	auto it = st.allocs.upper_bound((uintptr_t)p);
	if (it == st.allocs.begin()) return -1;
	--it;
	return (uintptr_t)p < it->first + it->second.bytes ? it->second.device : -1;
}
}

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : (e == hipErrorOutOfMemory ? "hipErrorOutOfMemory (emulated)" : "hipErrorInvalidValue (emulated)"); }
inline hipError_t hipGetLastError() { hipemu::DeviceState &st = hipemu::devices(); std::lock_guard<std::recursive_mutex> lk(st.m); const hipError_t e = st.sticky; st.sticky = hipSuccess; return e; }
inline hipError_t hipGetDeviceCount(int *n) { *n = hipemu::devices().ndevices; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= hipemu::devices().ndevices) return hipErrorInvalidValue; hipemu::current_device() = d; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = hipemu::current_device(); return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }      // two "compute units": persistent kernels size their grids by this

inline hipError_t hipemu_alloc(void **p, size_t bytes) { *p = nullptr; if (posix_memalign(p, 256, bytes ? bytes : 256)) return hipErrorOutOfMemory; memset(*p, 0xA5, bytes); return hipSuccess; }      // (hipMalloc hands out uninitialised memory: a pattern finds code that relies on zeros)
inline hipError_t hipemu_device_alloc(void **p, size_t bytes)
{
	hipemu::DeviceState &st = hipemu::devices();
	const size_t page = (size_t)sysconf(_SC_PAGESIZE), n = ((bytes ? bytes : 1) + page - 1) / page * page;
	void *m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (m == MAP_FAILED) { *p = nullptr; return hipErrorOutOfMemory; }
	memset(m, 0xA5, n);
	std::lock_guard<std::recursive_mutex> lk(st.m);
	st.allocs[(uintptr_t)m] = hipemu::Alloc{ n, hipemu::current_device() };
	if (!(st.open_depth && st.open_device == hipemu::current_device())) mprotect(m, n, PROT_NONE);
	*p = m;
	return hipSuccess;
}
template <typename T> inline hipError_t hipMalloc(T **p, size_t bytes) { return hipemu_device_alloc((void **)p, bytes); }
template <typename T> inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned = 0) { return hipemu_alloc((void **)p, bytes); }
inline hipError_t hipFree(void *p)
{
	if (!p) return hipSuccess;
	hipemu::DeviceState &st = hipemu::devices();
	std::lock_guard<std::recursive_mutex> lk(st.m);
	auto it = st.allocs.find((uintptr_t)p);
	if (it == st.allocs.end()) return hipErrorInvalidValue;
	munmap(p, it->second.bytes); st.allocs.erase(it);
	return hipSuccess;
}
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void *) { return hipSuccess; }

// A copy runs on the device of its stream (synchronous copies: of the calling thread); device memory on either side has to be that device's.
inline hipError_t hipemu_copy_device(const void *d, const void *s, hipStream_t st, int *dev)
{
	*dev = st ? st->device : hipemu::current_device();
	if (st && st->device != hipemu::current_device()) return hipemu::violation("copy queued on a stream of another device than the thread's current one", st->device, hipemu::current_device());
	const int dd = hipemu::device_of_pointer(d), sd = hipemu::device_of_pointer(s);
	if (dd >= 0 && dd != *dev) return hipemu::violation("copy into memory of another device", dd, *dev);
	if (sd >= 0 && sd != *dev) return hipemu::violation("copy out of memory of another device", sd, *dev);
	return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st = nullptr)
{
	int dev; if (hipemu_copy_device(d, s, st, &dev)) return hipErrorInvalidValue;
	hipemu::HeapOpen open(dev); memmove(d, s, n); return hipSuccess;
}
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) { return hipMemcpyAsync(d, s, n, k, nullptr); }
inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t st = nullptr)
{
	int dev; if (hipemu_copy_device(d, s, st, &dev)) return hipErrorInvalidValue;
	hipemu::HeapOpen open(dev);
	for (size_t r = 0; r < height; r++) memmove((uint8_t *)d + r * dpitch, (const uint8_t *)s + r * spitch, width);
	return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = nullptr)
{
	int dev; if (hipemu_copy_device(d, d, st, &dev)) return hipErrorInvalidValue;
	hipemu::HeapOpen open(dev); memset(d, v, n); return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) { return hipMemsetAsync(d, v, n, nullptr); }
inline hipError_t hipMemset2DAsync(void *d, size_t pitch, int v, size_t width, size_t height, hipStream_t st = nullptr)
{
	int dev; if (hipemu_copy_device(d, d, st, &dev)) return hipErrorInvalidValue;
	hipemu::HeapOpen open(dev);
	for (size_t r = 0; r < height; r++) memset((uint8_t *)d + r * pitch, v, width);
	return hipSuccess;
}

inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new hipemuStream_{ hipemu::current_device() }; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemuEvent_{ hipemu::current_device() }; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr)
{
	if (e && s && e->device != s->device) { hipemu::violation("event recorded on a stream of another device", e->device, s->device); return hipErrorInvalidValue; }
	return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

namespace hipemu {
// One launch at a time, process wide: the emulator's scheduler, threadIdx / blockIdx and the kernels' `__shared__` statics are global state.
inline std::mutex &launch_mutex() { static std::mutex m; return m; }
template <typename F> inline void launch_sync(const char *kernel, dim3 grid, dim3 block, hipStream_t stream, F body)
{
	std::lock_guard<std::mutex> lk(launch_mutex());
	const int dev = stream ? stream->device : current_device();
	if (stream && stream->device != current_device()) { violation(kernel, stream->device, current_device()); return; }      // (the hardware refuses a launch on another device's stream)
	HeapOpen open(dev);                                 // the kernel sees its own device's memory and nothing else
	static const bool trace = getenv("HIPEMU_TRACE") != nullptr;      // one line per launch on stderr: which kernel, which grid, how long the emulation took
	timespec t0, t1;
	if (trace) clock_gettime(CLOCK_MONOTONIC, &t0);
	launch(grid, block, body);
	if (trace) { clock_gettime(CLOCK_MONOTONIC, &t1); fprintf(stderr, "[hipemu] %-28s grid %u x %u x %u  block %u  dev %d  %.1f ms\n", kernel, grid.x, grid.y, grid.z, block.x * block.y * block.z, dev, (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6); }
}
}
