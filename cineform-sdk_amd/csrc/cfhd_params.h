// cfhd_params.h -- what CFHD_PrepareToEncode derives from its arguments, for the other front ends of the same encoder (cfhd_batch.cpp).
#pragma once
#include "cfhd_core.h"
#include <stdint.h>

namespace cfhd {

struct FrontEndParams {
	int pixel_kind = 0, encoded_format = 0, pixel_bytes = 2;
	int color_format = 2, color_space = 2, quality = 0;      // as the sample header carries them (quality incl. the 4:4:4:4 marker)
	bool progressive = true;
	bool static_quantizer = true;                             // false: the quality re-derives its tables from the size of the previous sample (rate feedback)
	FramePlan plan;                                           // geometry + first-frame quantizer
};
// Same checks and derivations as CFHD_PrepareToEncode (cfhd_api.cpp make_params).  Returns a CFHD_Error value (0 = OK).
int front_end_params(int width, int height, uint32_t pixel_format, int encoded_format, uint32_t encoding_flags, int quality, FrontEndParams *out);

} // namespace cfhd
