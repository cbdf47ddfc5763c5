/* include/cfhd_amd.h -- C ABI of libcfhd_amd.so, the MI355X-native CineForm encode/decode core.
 *
 * The entry points are, symbol for symbol and argument for argument, the ones the reference SDK
 * exports for this path, so an application (or the reference's own Example/TestCFHD.cpp) that was
 * compiled against the reference headers links against this library unchanged.  Each declaration
 * cites the reference interface it replaces.  Handles are opaque; every function returns a
 * CFHD error code (0 = CFHD_ERROR_OKAY, Common/CFHDError.h:25-82).
 *
 * Scope: intra-frame encode of YUY2 / 2vuy (progressive and interlaced), YU64 and v210 -> YUV 4:2:2 10-bit, RG48 / RG24 / BGRA / BGRa /
 * r210 / DPX0 / AB10 / AR10 / b64a -> RGB 4:4:4 12-bit, b64a -> RGBA 4:4:4:4 12-bit, BYR4 -> Bayer 12-bit; decode of 4:2:2 samples to
 * YUY2 / 2vuy (full and half resolution, interlaced samples too) and YU64 (full resolution), RGB 4:4:4 samples to RG48 (and RG24 / BGRA / BGRa
 * at full resolution) and RGBA 4:4:4:4
 * samples to b64a at full and half resolution (DESIGN.md section 1 lists what each round added).  Anything else
 * returns CFHD_ERROR_BADFORMAT (3) / CFHD_ERROR_BAD_RESOLUTION (11).
 * There is no CPU fallback: without a HIP device the encode/decode calls return CFHD_ERROR_INTERNAL (6).
 */
#ifndef CFHD_AMD_H
#define CFHD_AMD_H
#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int CFHD_Error;                    /* Common/CFHDError.h:25 (enum, int sized) */
typedef uint32_t CFHD_PixelFormat;         /* Common/CFHDTypes.h:41  four character codes, e.g. 'YUY2' */
typedef int CFHD_EncodedFormat;            /* Common/CFHDTypes.h:231 0 = YUV 4:2:2, 1 = RGB 4:4:4, 2 = RGBA, 3 = Bayer */
typedef uint32_t CFHD_EncodingFlags;       /* Common/CFHDTypes.h:282 */
typedef int CFHD_EncodingQuality;          /* Common/CFHDTypes.h:200 1 = LOW .. 4 = FILMSCAN1 .. 6 = FILMSCAN3 */
typedef int CFHD_DecodedResolution;        /* Common/CFHDTypes.h:451 1 = full */
typedef uint32_t CFHD_DecodingFlags;       /* Common/CFHDTypes.h:489 */
typedef int CFHD_MetadataType;             /* Common/CFHDTypes.h:307 */
typedef int CFHD_MetadataTrack;            /* Common/CFHDTypes.h:405 */
typedef int CFHD_SampleInfoTag;            /* Common/CFHDTypes.h:182 */
typedef int CFHD_VideoSelect;              /* Common/CFHDTypes.h:417 */
typedef int CFHD_Stereo3DType;             /* Common/CFHDTypes.h:425 */
typedef int32_t CFHD_MetadataSize;
typedef struct cfhd_allocator CFHD_ALLOCATOR;   /* Common/CFHDAllocator.h (optional, may be NULL; unused here) */

typedef void *CFHD_EncoderRef;             /* Common/CFHDEncoder.h:54-57 */
typedef void *CFHD_MetadataRef;
typedef void *CFHD_EncoderPoolRef;
typedef void *CFHD_SampleBufferRef;
typedef void *CFHD_DecoderRef;             /* Common/CFHDDecoder.h:51 */

/* ---------------- synchronous encoder: EncoderSDK/CFHDEncoder.cpp ---------------- */
CFHD_Error CFHD_OpenEncoder(CFHD_EncoderRef *encoderRefOut, CFHD_ALLOCATOR *allocator);                 /* CFHDEncoder.h:255, .cpp:150 */
CFHD_Error CFHD_GetInputFormats(CFHD_EncoderRef encoderRef, CFHD_PixelFormat *inputFormatArray,
                                int inputFormatArrayLength, int *actualInputFormatCountOut);             /* CFHDEncoder.h:259 */
CFHD_Error CFHD_PrepareToEncode(CFHD_EncoderRef encoderRef, int frameWidth, int frameHeight,
                                CFHD_PixelFormat pixelFormat, CFHD_EncodedFormat encodedFormat,
                                CFHD_EncodingFlags encodingFlags, CFHD_EncodingQuality encodingQuality); /* CFHDEncoder.h:265, .cpp:261 */
CFHD_Error CFHD_SetEncodeLicense(CFHD_EncoderRef encoderRef, unsigned char *licenseKey);                 /* CFHDEncoder.h:274 (no-op) */
CFHD_Error CFHD_SetEncodeLicense2(CFHD_EncoderRef encoderRef, unsigned char *licenseKey, uint32_t *level);
CFHD_Error CFHD_EncodeSample(CFHD_EncoderRef encoderRef, void *frameBuffer, int framePitch);             /* CFHDEncoder.h:284, .cpp:319 */
CFHD_Error CFHD_GetSampleData(CFHD_EncoderRef encoderRef, void **sampleDataOut, size_t *sampleSizeOut);  /* CFHDEncoder.h:289, .cpp:382 */
CFHD_Error CFHD_CloseEncoder(CFHD_EncoderRef encoderRef);                                                /* CFHDEncoder.h:294 */

/* ---------------- encoder metadata: EncoderSDK/CFHDEncoderMetadata.cpp ---------------- */
CFHD_Error CFHD_MetadataOpen(CFHD_MetadataRef *metadataRefOut);                                          /* CFHDEncoder.h:313 */
CFHD_Error CFHD_MetadataAdd(CFHD_MetadataRef metadataRef, uint32_t tag, CFHD_MetadataType type,
                            size_t size, uint32_t *data, bool temporary);                                /* CFHDEncoder.h:316 */
CFHD_Error CFHD_MetadataAttach(CFHD_EncoderRef encoderRef, CFHD_MetadataRef metadataRef);                /* CFHDEncoder.h:324 */
CFHD_Error CFHD_MetadataClose(CFHD_MetadataRef metadataRef);                                             /* CFHDEncoder.h:327 */

/* ---------------- asynchronous encoder pool: EncoderSDK/CFHDEncoderPool.cpp ----------------
 * encoderThreadCount -> HIP streams (frames in flight on the GPU), jobQueueLength -> queued frames.
 * Samples come back in submission order (EncoderSDK/EncoderPool.cpp:297-380). */
CFHD_Error CFHD_CreateEncoderPool(CFHD_EncoderPoolRef *encoderPoolRefOut, int encoderThreadCount,
                                  int jobQueueLength, CFHD_ALLOCATOR *allocator);                        /* CFHDEncoder.h:338, Pool.cpp:103 */
CFHD_Error CFHD_GetAsyncInputFormats(CFHD_EncoderPoolRef encoderPoolRef, CFHD_PixelFormat *inputFormatArray,
                                     int inputFormatArrayLength, int *actualInputFormatCountOut);
CFHD_Error CFHD_PrepareEncoderPool(CFHD_EncoderPoolRef encoderPoolRef, uint_least16_t frameWidth, uint_least16_t frameHeight,
                                   CFHD_PixelFormat pixelFormat, CFHD_EncodedFormat encodedFormat,
                                   CFHD_EncodingFlags encodingFlags, CFHD_EncodingQuality encodingQuality); /* Pool.cpp:176 */
CFHD_Error CFHD_SetEncoderPoolLicense(CFHD_EncoderPoolRef encoderPoolRef, unsigned char *licenseKey);
CFHD_Error CFHD_SetEncoderPoolLicense2(CFHD_EncoderPoolRef encoderPoolRef, unsigned char *licenseKey, uint32_t *level);
CFHD_Error CFHD_AttachEncoderPoolMetadata(CFHD_EncoderPoolRef encoderPoolRef, CFHD_MetadataRef metadataRef);
CFHD_Error CFHD_StartEncoderPool(CFHD_EncoderPoolRef encoderPoolRef);                                    /* Pool.cpp:360 */
CFHD_Error CFHD_StopEncoderPool(CFHD_EncoderPoolRef encoderPoolRef);
CFHD_Error CFHD_EncodeAsyncSample(CFHD_EncoderPoolRef encoderPoolRef, uint32_t frameNumber, void *frameBuffer,
                                  intptr_t framePitch, CFHD_MetadataRef metadataRef);                    /* Pool.cpp:436 */
CFHD_Error CFHD_WaitForSample(CFHD_EncoderPoolRef encoderPoolRef, uint32_t *frameNumberOut,
                              CFHD_SampleBufferRef *sampleBufferRefOut);                                 /* Pool.cpp:475 */
CFHD_Error CFHD_TestForSample(CFHD_EncoderPoolRef encoderPoolRef, uint32_t *frameNumberOut,
                              CFHD_SampleBufferRef *sampleBufferRefOut);                                 /* Pool.cpp:520 */
CFHD_Error CFHD_GetEncodedSample(CFHD_SampleBufferRef sampleBufferRef, void **sampleDataOut, size_t *sampleSizeOut); /* Pool.cpp:557 */
CFHD_Error CFHD_ReleaseSampleBuffer(CFHD_EncoderPoolRef encoderPoolRef, CFHD_SampleBufferRef sampleBufferRef);       /* Pool.cpp:726 */
CFHD_Error CFHD_ReleaseEncoderPool(CFHD_EncoderPoolRef encoderPoolRef);

/* ---------------- decoder: DecoderSDK/CFHDDecoder.cpp ---------------- */
CFHD_Error CFHD_OpenDecoder(CFHD_DecoderRef *decoderRefOut, CFHD_ALLOCATOR *allocator);                  /* CFHDDecoder.h:203 */
CFHD_Error CFHD_GetOutputFormats(CFHD_DecoderRef decoderRef, void *samplePtr, size_t sampleSize,
                                 CFHD_PixelFormat *outputFormatArray, int outputFormatArrayLength, int *actualOutputFormatCountOut);
CFHD_Error CFHD_GetSampleInfo(CFHD_DecoderRef decoderRef, void *samplePtr, size_t sampleSize,
                              CFHD_SampleInfoTag tag, void *value, size_t buffer_size);                  /* CFHDDecoder.h:216 */
CFHD_Error CFHD_PrepareToDecode(CFHD_DecoderRef decoderRef, int outputWidth, int outputHeight, CFHD_PixelFormat outputFormat,
                                CFHD_DecodedResolution decodedResolution, CFHD_DecodingFlags decodingFlags,
                                void *samplePtr, size_t sampleSize, int *actualWidthOut, int *actualHeightOut,
                                CFHD_PixelFormat *actualFormatOut);                                      /* CFHDDecoder.h:224 */
CFHD_Error CFHD_GetPixelSize(CFHD_PixelFormat pixelFormat, uint32_t *pixelSizeOut);                      /* CFHDDecoder.h:244 */
CFHD_Error CFHD_GetImagePitch(uint32_t imageWidth, CFHD_PixelFormat pixelFormat, int32_t *imagePitchOut);/* CFHDDecoder.h:247 */
CFHD_Error CFHD_GetImageSize(uint32_t imageWidth, uint32_t imageHeight, CFHD_PixelFormat pixelFormat,
                             CFHD_VideoSelect videoselect, CFHD_Stereo3DType stereotype, uint32_t *imageSizeOut); /* CFHDDecoder.h:250 */
CFHD_Error CFHD_DecodeSample(CFHD_DecoderRef decoderRef, void *samplePtr, size_t sampleSize,
                             void *outputBuffer, int32_t outputPitch);                                   /* CFHDDecoder.h:262, .cpp:716 */
CFHD_Error CFHD_SetLicense(CFHD_DecoderRef decoderRef, const unsigned char *licenseKey);
CFHD_Error CFHD_SetActiveMetadata(CFHD_DecoderRef decoderRef, CFHD_MetadataRef metadataRef, unsigned int tag,
                                  CFHD_MetadataType type, void *data, unsigned int size);                /* CFHDDecoder.h:272 (accepted, ignored) */
CFHD_Error CFHD_ClearActiveMetadata(CFHD_DecoderRef decoderRef, CFHD_MetadataRef metadataRef);
CFHD_Error CFHD_GetThumbnail(CFHD_DecoderRef decoderRef, void *samplePtr, size_t sampleSize, void *outputBuffer,
                             size_t outputBufferSize, uint32_t flags, size_t *retWidth, size_t *retHeight, size_t *retSize); /* CFHDDecoder.cpp:1512 */
/* the same 1/8 x 1/8 10-bit RGB thumbnail through the encoder-side handles */
CFHD_Error CFHD_GetEncodeThumbnail(CFHD_EncoderRef encoderRef, void *samplePtr, size_t sampleSize, void *outputBuffer,
                                   size_t outputBufferSize, uint32_t flags, size_t *retWidth, size_t *retHeight, size_t *retSize); /* CFHDEncoder.cpp:593 */
CFHD_Error CFHD_GetSampleThumbnail(CFHD_SampleBufferRef sampleBufferRef, void *thumbnailBuffer, size_t bufferSize, uint32_t flags,
                                   uint_least16_t *actualWidthOut, uint_least16_t *actualHeightOut, CFHD_PixelFormat *pixelFormatOut,
                                   size_t *actualSizeOut);                                              /* CFHDEncoderPool.cpp:620 */
/* obsolete in the reference (use CFHD_GetSampleInfo); layout of Common/CFHDSampleHeader.h:32 */
typedef struct CFHD_SampleHeader { int encoded_format; int field_type; int width; int height; } CFHD_SampleHeader;
CFHD_Error CFHD_ParseSampleHeader(void *samplePtr, size_t sampleSize, CFHD_SampleHeader *sampleHeader);   /* CFHDDecoder.cpp:443 */
CFHD_Error CFHD_CloseDecoder(CFHD_DecoderRef decoderRef);                                                /* CFHDDecoder.h:300 */

/* ---------------- decoder-side metadata: DecoderSDK/CFHDMetadata.cpp ---------------- */
CFHD_Error CFHD_OpenMetadata(CFHD_MetadataRef *metadataRefOut);                                          /* CFHDMetadata.cpp:115 */
CFHD_Error CFHD_InitSampleMetadata(CFHD_MetadataRef metadataRef, CFHD_MetadataTrack track, void *sampleData, size_t sampleSize); /* :157 */
CFHD_Error CFHD_ReadMetadata(CFHD_MetadataRef metadataRef, unsigned int *tag, CFHD_MetadataType *type, void **data, CFHD_MetadataSize *size);
CFHD_Error CFHD_FindMetadata(CFHD_MetadataRef metadataRef, unsigned int tag, CFHD_MetadataType *type, void **data, CFHD_MetadataSize *size);
CFHD_Error CFHD_CloseMetadata(CFHD_MetadataRef metadataRef);                                             /* :1371 */

/* ---------------- extensions of this library (not in the reference ABI) ---------------- */
/* Batched, device-resident round trip used by bench.py: frames already in HBM -> samples -> frames in HBM. */
typedef struct cfhd_amd_batch cfhd_amd_batch;
/* nframes frames of one geometry travel through every stage together, one launch per stage (cfhd_batch.cpp).  The arguments are those of
 * CFHD_PrepareToEncode; mode 0: encode + decode back to the same pixel format, mode 1: encode only (the only mode for BYR4).
 * cfhd_amd_batch_create: 4:2:2 progressive round trip (YUY2 / 2vuy), kept for callers of the first release. */
cfhd_amd_batch *cfhd_amd_batch_create_ex(int width, int height, uint32_t pixel_format, int encoded_format, uint32_t encoding_flags,
                                         int quality, int nframes, int nthreads, int mode);
cfhd_amd_batch *cfhd_amd_batch_create(int width, int height, uint32_t pixel_format, int quality, int nframes, int nthreads);
void cfhd_amd_batch_destroy(cfhd_amd_batch *batch);
int  cfhd_amd_batch_upload(cfhd_amd_batch *batch, int frame, const void *pixels, int pitch);   /* host frame -> HBM (outside any timed region) */
long long cfhd_amd_batch_roundtrip(cfhd_amd_batch *batch);                                     /* one pass; total sample bytes, or < 0 */
/* The same pass as a slot of a frame queue (replaces the reference's EncoderPool job queue, EncoderSDK/EncoderPool.cpp:239-380): submit returns at once -- the whole
 * pass is queued on the batch's HIP streams, the decoder's stream waiting for the encoder's events; no host thread, no host wait inside the pass --, wait returns what
 * cfhd_amd_batch_roundtrip would have; batches in flight at the same time overlap on the GPU.  One pass per batch at a time: between submit and wait every other entry
 * point on that batch (roundtrip, upload, get_sample, download_output, kernel_ms, dx_stats) returns its error value without touching the batch; destroy waits first.
 * Encode-only batches in flight on one device take turns (their host copies hide behind the next batch's kernels), round-trip batches run free.  A process that keeps
 * several batches in flight should run with GPU_MAX_HW_QUEUES=16 in its environment (ROCm runtime, read at start-up: by default all HIP streams of a process share 4
 * hardware queues and the streams of several passes wait for each other). */
int  cfhd_amd_batch_submit(cfhd_amd_batch *batch);
/* The same pass fed from host memory: frame i at frames + i * frame_stride (rows `pitch` bytes apart) goes to HBM on the pass's own stream, the decoded pictures come back to
 * pictures + i * picture_stride (NULL: none) behind it; both buffers are borrowed until cfhd_amd_batch_wait returns.  Buffers registered with
 * cfhd_amd_register_host_buffer are copied by DMA as they are, plain ones are staged by the calling thread.  This is the pool semantics of the reference
 * (EncoderSDK/EncoderPool.cpp:239-295: frames in, samples out, nothing resident) for whole batches; bench.py's `host_fed` figure times it. */
int  cfhd_amd_batch_submit_host(cfhd_amd_batch *batch, const void *frames, size_t frame_stride, int pitch, void *pictures, size_t picture_stride, int picture_pitch);
long long cfhd_amd_batch_wait(cfhd_amd_batch *batch);
int  cfhd_amd_batch_get_sample(cfhd_amd_batch *batch, int frame, const void **data, size_t *size);
int  cfhd_amd_batch_download_output(cfhd_amd_batch *batch, int frame, void *out, int pitch);
float cfhd_amd_batch_kernel_ms(cfhd_amd_batch *batch, int which);                              /* HIP-event time of the kernels of the last pass */
const char *cfhd_amd_batch_kernel_name(cfhd_amd_batch *batch, int which);                      /* which 0..5: the transform kernel behind that time */
double cfhd_amd_batch_stage_seconds(cfhd_amd_batch *batch, int which);
int  cfhd_amd_batch_dx_stats(cfhd_amd_batch *batch, uint32_t *out16);
int  cfhd_amd_device_count(void);
/* Text of the last HIP / device failure behind a CFHD_ERROR_INTERNAL (the library has no CPU fallback: without a gfx950 device every
 * compute call fails and says why here). */
const char *cfhd_amd_last_error(void);
/* Fixes the otherwise random clip GUID that every new encoder stamps into its samples (16 bytes), for bit-exact diffs. */
void cfhd_amd_set_clip_guid(const unsigned char guid[16]);
/* Optional: page-lock a frame / output buffer the caller reuses, so that CFHD_EncodeSample, the encoder pool and CFHD_DecodeSample
 * move it over PCIe without a staging copy.  The caller keeps it alive and unregisters it before freeing it.  Returns 0 on success. */
int  cfhd_amd_register_host_buffer(void *buffer, size_t bytes);
int  cfhd_amd_unregister_host_buffer(void *buffer);

#ifdef __cplusplus
}
#endif
#endif
