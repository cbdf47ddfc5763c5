// tools/microbench_level1.hip -- development aid (not part of the library, not used by the tests): times k_inv_yuv422 against
// stripped variants of itself on a synthetic 1080p batch, to tell which resource bounds it:
//   tile        the LDS-tiled product kernel k_inv_yuv422
//   strip       the register-strip product kernel k_inv_yuv422_strip
//   access      the same dword loads and 8-byte stores, no LDS, no arithmetic  (floor of this access pattern)
//   wide        16-byte loads of the tile interior, 16-byte stores              (floor of a wide access pattern)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Icineform-sdk_amd/csrc tools/microbench_level1.hip -o tools/_build/mb_level1
#include "cfhd_kernels.h"
#include <stdio.h>
#include <vector>
using namespace cfhd::dev;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(NTHREADS) mb_access(const InvYuvJob *jobs)
{
	const TileId tile = xcd_tile();
	__shared__ InvYuvJob s_job;
	stage_job(&s_job, &jobs[tile.z]);
	const InvYuvJob &job = s_job;
	const int w = job.width, h = job.height, c0 = tile.x * ITW, r0 = tile.y * ITH, cw = w >> 1, cc0 = c0 >> 1;
	enum { CDW = ITW / 2 + 4, CLD = CDW / 2 };
	if (!((c0 < w) && (r0 < h))) return;
	const int tid = threadIdx.x, rs = inv_tile_first_row(r0, h);
	enum { NL = (2 * ILROWS * IDW + NTHREADS - 1) / NTHREADS, NH = (2 * ITH * IDW + NTHREADS - 1) / NTHREADS,
	       NCL = (2 * ILROWS * CLD + NTHREADS - 1) / NTHREADS, NCH = (2 * ITH * CLD + NTHREADS - 1) / NTHREADS };
	uint32_t yl[NL], yh[NH], cl[2][NCL], ch[2][NCH];
	const int dw0 = (c0 >> 1) - 1, wdw = (w + 1) >> 1, cdw0 = (cc0 >> 1) - 1, cwdw = (cw + 1) >> 1;
	inv_stage_load<ILROWS, IDW>(yl, job.band[0][0], job.band[0][1], job.band_pitch[0], rs, h, dw0, wdw);
	inv_stage_load<ITH, IDW>(yh, job.band[0][2], job.band[0][3], job.band_pitch[0], r0, h, dw0, wdw);
#pragma unroll
	for (int x = 0; x < 2; x++) {
		inv_stage_load<ILROWS, CLD>(cl[x], job.band[1 + x][0], job.band[1 + x][1], job.band_pitch[1 + x], rs, h, cdw0, cwdw);
		inv_stage_load<ITH, CLD>(ch[x], job.band[1 + x][2], job.band[1 + x][3], job.band_pitch[1 + x], r0, h, cdw0, cwdw);
	}
	uint32_t acc = 0;
#pragma unroll
	for (int k = 0; k < NL; k++) acc ^= yl[k];
#pragma unroll
	for (int k = 0; k < NH; k++) acc ^= yh[k];
#pragma unroll
	for (int x = 0; x < 2; x++) {
#pragma unroll
		for (int k = 0; k < NCL; k++) acc ^= cl[x][k];
#pragma unroll
		for (int k = 0; k < NCH; k++) acc ^= ch[x][k];
	}
	for (int i = tid; i < 2 * ITH * (ITW / 2); i += NTHREADS) {
		const int par = i / (ITH * (ITW / 2)), rem = i - par * (ITH * (ITW / 2)), rl = rem / (ITW / 2), p = rem - rl * (ITW / 2);
		const int r = r0 + rl, cc = cc0 + p, orow = 2 * r + par;
		if (r >= h || cc >= cw || orow >= job.display_height) continue;
		uint2 o2; o2.x = acc; o2.y = acc ^ (uint32_t)i;
		*(uint2 *)(job.out + (size_t)orow * job.out_pitch + 8 * (size_t)cc) = o2;
	}
}

__global__ void __launch_bounds__(NTHREADS) mb_wide(const InvYuvJob *jobs)
{
	const TileId tile = xcd_tile();
	__shared__ InvYuvJob s_job;
	stage_job(&s_job, &jobs[tile.z]);
	const InvYuvJob &job = s_job;
	const int w = job.width, h = job.height, c0 = tile.x * ITW, r0 = tile.y * ITH, cc0 = c0 >> 1;
	if (!((c0 < w) && (r0 < h))) return;
	const int tid = threadIdx.x, rs = inv_tile_first_row(r0, h);
	uint4 acc = { 0, 0, 0, 0 };
	enum { YROWS = 2 * (ILROWS + ITH), YQ = ITW / 8, CQ = ITW / 16 };        // band rows of the tile (two bands each for low and high), uint4 per row
	uint4 v[3 + 3];
#pragma unroll
	for (int k = 0; k < 3; k++) {
		const int i = tid + k * NTHREADS;
		v[k] = acc;
		if (i < YROWS * YQ) {
			const int rr = i / YQ, q = i - rr * YQ;
			const int band = rr < 2 * ILROWS ? rr / ILROWS : 2 + (rr - 2 * ILROWS) / ITH;
			int row = rr < 2 * ILROWS ? rs + rr % ILROWS : r0 + (rr - 2 * ILROWS) % ITH;
			row = row < h ? row : h - 1;
			int col = c0 + 8 * q; col = col + 8 <= w ? col : w - 8;
			v[k] = *(const uint4 *)(job.band[0][band] + (size_t)row * job.band_pitch[0] + col);
		}
	}
#pragma unroll
	for (int k = 0; k < 3; k++) {
		const int i = tid + k * NTHREADS;
		v[3 + k] = acc;
		if (i < 2 * YROWS * CQ) {
			const int x = i / (YROWS * CQ), i2 = i - x * (YROWS * CQ), rr = i2 / CQ, q = i2 - rr * CQ;
			const int band = rr < 2 * ILROWS ? rr / ILROWS : 2 + (rr - 2 * ILROWS) / ITH;
			int row = rr < 2 * ILROWS ? rs + rr % ILROWS : r0 + (rr - 2 * ILROWS) % ITH;
			row = row < h ? row : h - 1;
			int col = cc0 + 8 * q; col = col + 8 <= (w >> 1) ? col : (w >> 1) - 8;
			v[3 + k] = *(const uint4 *)(job.band[1 + x][band] + (size_t)row * job.band_pitch[1 + x] + col);
		}
	}
#pragma unroll
	for (int k = 0; k < 6; k++) { acc.x ^= v[k].x; acc.y ^= v[k].y; acc.z ^= v[k].z; acc.w ^= v[k].w; }
	// output tile: 2 * ITH rows of ITW luma columns * 2 samples * 2 bytes = 4 * ITW bytes = ITW / 4 uint4 per row
	for (int i = tid; i < 2 * ITH * (ITW / 4); i += NTHREADS) {
		const int orl = i / (ITW / 4), q = i - orl * (ITW / 4);
		const int orow = 2 * r0 + orl;
		const size_t byte = (size_t)4 * c0 + 16 * (size_t)q;
		if (orow >= job.display_height || byte + 16 > (size_t)4 * w) continue;
		uint4 o = acc; o.x ^= (uint32_t)i;
		*(uint4 *)(job.out + (size_t)orow * job.out_pitch + byte) = o;
	}
}

int main(int argc, char **argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 256, reps = 20;
	const int w = 960, h = 540, cw = 480;                 // level-1 band dimensions of 1080p 4:2:2
	const size_t ybytes = (size_t)w * h * 2, cbytes = (size_t)cw * h * 2, frame_in = 4 * ybytes + 8 * cbytes, frame_out = (size_t)1920 * 2 * 1080;
	uint8_t *d_in, *d_out; InvYuvJob *d_jobs;
	CK(hipMalloc(&d_in, frame_in * n)); CK(hipMalloc(&d_out, frame_out * n)); CK(hipMalloc(&d_jobs, sizeof(InvYuvJob) * n));
	{
		std::vector<int16_t> host(frame_in / 2);
		uint32_t s = 12345;
		for (auto &x : host) { s = s * 1664525u + 1013904223u; x = (int16_t)((s >> 20) % 61) - 30; }
		for (int i = 0; i < n; i++) CK(hipMemcpy(d_in + frame_in * i, host.data(), frame_in, hipMemcpyHostToDevice));
	}
	std::vector<InvYuvJob> jobs(n);
	for (int i = 0; i < n; i++) {
		InvYuvJob &j = jobs[i];
		uint8_t *p = d_in + frame_in * i;
		for (int b = 0; b < 4; b++) { j.band[0][b] = (const int16_t *)p; p += ybytes; }
		for (int c = 1; c < 3; c++) for (int b = 0; b < 4; b++) { j.band[c][b] = (const int16_t *)p; p += cbytes; }
		j.band_pitch[0] = w; j.band_pitch[1] = j.band_pitch[2] = cw;
		j.width = w; j.height = h; j.display_height = 1080; j.uyvy = 0; j.shift = 2; j.dither_seed = 77u * (i + 1);
		j.out = d_out + frame_out * i; j.out_pitch = 3840;
	}
	CK(hipMemcpy(d_jobs, jobs.data(), sizeof(InvYuvJob) * n, hipMemcpyHostToDevice));
	hipStream_t st; CK(hipStreamCreate(&st));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	dim3 grid((w + ITW - 1) / ITW, (h + ITH - 1) / ITH, n);
	const double bytes = (double)(frame_in + frame_out) * n;
	for (int variant = 0; variant < 4; variant++) {
		const char *name = variant == 0 ? "tile" : variant == 1 ? "access" : variant == 2 ? "wide" : "strip";
		float best = 1e9f, sum = 0;
		for (int r = 0; r < reps + 3; r++) {
			CK(hipEventRecord(e0, st));
			if (variant == 0) k_inv_yuv422<<<grid, NTHREADS, 0, st>>>(d_jobs, 1u);
			else if (variant == 1) mb_access<<<grid, NTHREADS, 0, st>>>(d_jobs);
			else if (variant == 2) mb_wide<<<grid, NTHREADS, 0, st>>>(d_jobs);
			else k_inv_yuv422_strip<<<dim3((w / SBLK + SSEG - 1) / SSEG, (h + SR - 1) / SR, n), NTHREADS, 0, st>>>(d_jobs, 1u);
			CK(hipEventRecord(e1, st));
			CK(hipStreamSynchronize(st));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (r >= 3) { sum += ms; if (ms < best) best = ms; }
		}
		printf("inv %-7s frames %d  avg %.3f ms  best %.3f ms  %.0f GB/s (avg)\n", name, n, sum / reps, best, bytes / (sum / reps) * 1e-6);
	}
	// ---- forward level 1: k_fwd_yuv422 (LDS tiles) vs k_fwd_yuv422_strip
	{
		const int W = 1920, H = 1080;
		const size_t fbytes = (size_t)W * 2 * H, obytes = 4 * ybytes + 8 * cbytes;
		uint8_t *d_f; int16_t *d_o; FwdYuvJob *d_fj;
		CK(hipMalloc(&d_f, fbytes * n)); CK(hipMalloc(&d_o, obytes * n)); CK(hipMalloc(&d_fj, sizeof(FwdYuvJob) * n));
		{
			std::vector<uint8_t> host(fbytes);
			uint32_t s = 777;
			for (size_t i = 0; i < fbytes; i++) { s = s * 1664525u + 1013904223u; host[i] = (uint8_t)(128 + ((s >> 24) % 9) - 4 + (i / 64) % 50); }
			for (int i = 0; i < n; i++) CK(hipMemcpy(d_f + fbytes * i, host.data(), fbytes, hipMemcpyHostToDevice));
		}
		std::vector<FwdYuvJob> fj(n);
		const int quant[3][4] = { {1, 36, 24, 24}, {1, 48, 24, 24}, {1, 48, 24, 24} };
		for (int i = 0; i < n; i++) {
			FwdYuvJob &j = fj[i];
			j.in = d_f + fbytes * i; j.in_pitch = W * 2; j.width = W; j.height = 1080 + 0; j.display_height = H; j.uyvy = 0; j.shift = 2;
			j.height = 1080;
			uint8_t *p = (uint8_t *)d_o + obytes * i;
			for (int b = 0; b < 4; b++) { j.out[0][b] = (int16_t *)p; p += ybytes; }
			for (int c = 1; c < 3; c++) for (int b = 0; b < 4; b++) { j.out[c][b] = (int16_t *)p; p += cbytes; }
			j.out_pitch[0] = w; j.out_pitch[1] = j.out_pitch[2] = cw;
			for (int c = 0; c < 3; c++) for (int b = 0; b < 4; b++) { QuantParam q; q.divisor = quant[c][b]; q.mid = q.divisor > 1 ? q.divisor / 2 - 1 : 0; q.mult = q.divisor > 1 ? (65536u / q.divisor) & 0xffffu : 0; j.q[c][b] = q; }
		}
		CK(hipMemcpy(d_fj, fj.data(), sizeof(FwdYuvJob) * n, hipMemcpyHostToDevice));
		const double fb = (double)(fbytes + obytes) * n;
		for (int variant = 0; variant < 2; variant++) {
			float best = 1e9f, sum = 0;
			for (int r = 0; r < reps + 3; r++) {
				CK(hipEventRecord(e0, st));
				if (variant == 0) k_fwd_yuv422<<<dim3((W / 2 + TW - 1) / TW, (1080 / 2 + TH - 1) / TH, n), NTHREADS, 0, st>>>(d_fj);
				else k_fwd_yuv422_strip<<<dim3((W / 16 + SSEG - 1) / SSEG, (1080 / 2 + SRF - 1) / SRF, n), NTHREADS, 0, st>>>(d_fj);
				CK(hipEventRecord(e1, st));
				CK(hipStreamSynchronize(st));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				if (r >= 3) { sum += ms; if (ms < best) best = ms; }
			}
			printf("fwd %-7s frames %d  avg %.3f ms  best %.3f ms  %.0f GB/s (avg)\n", variant ? "strip" : "tile", n, sum / reps, best, fb / (sum / reps) * 1e-6);
		}
	}
	return 0;
}
