/* oracle/ref_glue.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin C entry points (prefix ref_) around the unmodified reference so that tests and the
 * cpu_baseline leg of bench.py can drive it through ctypes:
 *   - Qbist synthetic frames exactly as Example/TestCFHD.cpp:1149-1150,1208 generates them
 *   - PSNR exactly as Example/utils.cpp:471 computes it
 * The CFHD_* C ABI of the reference itself is exported by the same .so and is called directly.
 */
#include <stdint.h>
#include <string.h>
#include "CFHDTypes.h"
#include "qbist.h"
#include "utils.h"

extern "C" {

void ref_qbist_reset(unsigned int seed)
{
	GetRand(seed);
	initBaseTransform();
}

/* buf must hold width*height*8 bytes: Qbist renders RGBA16 in place (TestCFHD.cpp:1178). */
void ref_qbist_frame(int width, int height, int pitch, unsigned int pixelFormat, int alpha, unsigned char *buf)
{
	RunQBist(width, height, pitch, (CFHD_PixelFormat)pixelFormat, alpha, buf);
}

int ref_frame_pitch(unsigned int pixelFormat, int width)
{
	return FramePitch4PixelFormat((CFHD_PixelFormat)pixelFormat, width);
}

float ref_psnr(void *a, void *b, int width, int height, unsigned int pixelFormat, int scale)
{
	return PSNR(a, b, width, height, (CFHD_PixelFormat)pixelFormat, scale);
}

} // extern "C"
