/* oracle/shim/uuid/uuid.h -- TEST INFRASTRUCTURE ONLY.
 * Minimal stand-in for libuuid's header so the reference EncoderSDK
 * (EncoderSDK/SampleEncoder.cpp:26,763-764) builds without an external library and
 * emits a *deterministic* TAG_CLIP_GUID, which makes whole-sample bitstream diffs exact.
 */
#ifndef ORACLE_SHIM_UUID_H
#define ORACLE_SHIM_UUID_H
#ifdef __cplusplus
extern "C" {
#endif
typedef unsigned char uuid_t[16];
void uuid_generate(uuid_t out);
#ifdef __cplusplus
}
#endif
#endif
