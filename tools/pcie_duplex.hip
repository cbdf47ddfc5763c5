// tools/pcie_duplex.hip -- what the host link of the GPU box gives the host-fed frame queue (bench.py host_fed): pinned host memory <-> HBM, each direction alone and both
// at once on two streams, with hipHostMalloc'ed and with hipHostRegister'ed host memory, 512 MB per copy.   hipcc --offload-arch=gfx950 -O2 tools/pcie_duplex.hip -o /tmp/pcie_duplex
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const size_t N = (size_t)512 << 20; const int reps = 8;
	void *d_a, *d_b; CHK(hipMalloc(&d_a, N)); CHK(hipMalloc(&d_b, N));
	hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
	for (int kind = 0; kind < 2; kind++) {
		void *h_a, *h_b;
		if (kind == 0) { CHK(hipHostMalloc(&h_a, N, hipHostMallocPortable)); CHK(hipHostMalloc(&h_b, N, hipHostMallocPortable)); }
		else { h_a = aligned_alloc(4096, N); h_b = aligned_alloc(4096, N); memset(h_a, 1, N); memset(h_b, 2, N); CHK(hipHostRegister(h_a, N, hipHostRegisterDefault)); CHK(hipHostRegister(h_b, N, hipHostRegisterDefault)); }
		memset(h_a, 3, N);
		for (int mode = 0; mode < 3; mode++) {      // 0: host -> device alone, 1: device -> host alone, 2: both at once
			CHK(hipDeviceSynchronize());
			const double t0 = now();
			for (int r = 0; r < reps; r++) {
				if (mode != 1) CHK(hipMemcpyAsync(d_a, h_a, N, hipMemcpyHostToDevice, s1));
				if (mode != 0) CHK(hipMemcpyAsync(h_b, d_b, N, hipMemcpyDeviceToHost, s2));
			}
			CHK(hipStreamSynchronize(s1)); CHK(hipStreamSynchronize(s2));
			const double dt = now() - t0;
			printf("%-22s %-30s %6.1f GB/s per direction\n", kind ? "hipHostRegister'ed" : "hipHostMalloc'ed", mode == 0 ? "host -> device alone" : (mode == 1 ? "device -> host alone" : "both directions at once"), N * (double)reps / dt / 1e9);
		}
		if (kind == 0) { CHK(hipHostFree(h_a)); CHK(hipHostFree(h_b)); } else { CHK(hipHostUnregister(h_a)); CHK(hipHostUnregister(h_b)); free(h_a); free(h_b); }
	}
	return 0;
}
