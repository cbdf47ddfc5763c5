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

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : (e == hipErrorOutOfMemory ? "hipErrorOutOfMemory (emulated)" : "hipErrorInvalidValue (emulated)"); }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }      // two "compute units": persistent kernels size their grids by this

inline hipError_t hipemu_alloc(void **p, size_t bytes) { *p = nullptr; if (posix_memalign(p, 256, bytes ? bytes : 256)) return hipErrorOutOfMemory; memset(*p, 0xA5, bytes); return hipSuccess; }      // (hipMalloc hands out uninitialised memory: a pattern finds code that relies on zeros)
template <typename T> inline hipError_t hipMalloc(T **p, size_t bytes) { return hipemu_alloc((void **)p, bytes); }
template <typename T> inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned = 0) { return hipemu_alloc((void **)p, bytes); }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void *) { return hipSuccess; }

inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t = nullptr)
{
	for (size_t r = 0; r < height; r++) memmove((uint8_t *)d + r * dpitch, (const uint8_t *)s + r * spitch, width);
	return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset2DAsync(void *d, size_t pitch, int v, size_t width, size_t height, hipStream_t = nullptr)
{
	for (size_t r = 0; r < height; r++) memset((uint8_t *)d + r * pitch, v, width);
	return hipSuccess;
}

inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

namespace hipemu {
// One launch at a time, process wide: the emulator's scheduler, threadIdx / blockIdx and the kernels' `__shared__` statics are global state.
inline std::mutex &launch_mutex() { static std::mutex m; return m; }
template <typename F> inline void launch_sync(const char *kernel, dim3 grid, dim3 block, F body)
{
	std::lock_guard<std::mutex> lk(launch_mutex());
	static const bool trace = getenv("HIPEMU_TRACE") != nullptr;      // one line per launch on stderr: which kernel, which grid, how long the emulation took
	timespec t0, t1;
	if (trace) clock_gettime(CLOCK_MONOTONIC, &t0);
	launch(grid, block, body);
	if (trace) { clock_gettime(CLOCK_MONOTONIC, &t1); fprintf(stderr, "[hipemu] %-28s grid %u x %u x %u  block %u  %.1f ms\n", kernel, grid.x, grid.y, grid.z, block.x * block.y * block.z, (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6); }
}
}
