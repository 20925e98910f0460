/*
 * transform360-mi355x: the four C entry points the ffmpeg filter binds.
 *
 * Drop-in for the reference header of the same name
 *   reference: Transform360/Library/VideoFrameTransformHandler.h:22-47
 *   bodies   : Transform360/Library/VideoFrameTransformHandler.cpp:18-64
 * and called, unchanged, by Transform360/vf_transform360.c:
 *   _new                  vf_transform360.c:141   (ctx is a stack local -> copied)
 *   _generateMapForPlane  vf_transform360.c:157   (index 0: luma dims, index 1: chroma dims)
 *   _transformFramePlane  vf_transform360.c:383   (once per plane per frame)
 *   _delete               vf_transform360.c:334   (NULL must be accepted)
 *
 * Conventions kept from the reference:
 *   - int results are C++ bools: 1 = success, 0 = failure; diagnostics go to stdout.
 *   - no exception crosses this boundary; _new reports failure with NULL.
 *   - "...WidthWithPadding" is the row stride in BYTES (AVFrame.linesize).
 *   - frame buffers are borrowed for the duration of the call only.
 *
 * Attribution: the declarations below (function names and signatures / enumerator names and values,
 * the field names, order and types of FrameTransformContext) reproduce the public interface of
 * facebook/transform360, Copyright (c) 2015-present, Facebook, Inc., released under the BSD license
 * in that project's LICENSE file.  They are repeated here only because binary compatibility with
 * that interface is the purpose of this library; everything behind them is an independent
 * implementation.
 *
 * What is different behind the boundary: maps, low-pass filtering and the gather
 * run as HIP kernels on an MI355X.  inputData / outputData may be host pointers
 * (staged over PCIe, synchronous like the reference) or device pointers (used in
 * place; the call first waits for work already queued on the device unless the caller
 * took over ordering with T360_setStream -- see Transform360/t360_device.h for the
 * stream / batch additions).
 */
#ifndef TRANSFORM360_VIDEOFRAMETRANSFORMHANDLER_H
#define TRANSFORM360_VIDEOFRAMETRANSFORMHANDLER_H

#include "VideoFrameTransformHelper.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque to C callers; the C++ class lives inside the library.  The reference spells this
 * `typedef class VideoFrameTransform VideoFrameTransform;` for both languages (Handler.h:22),
 * which a C compiler rejects; the C branch below is the same incomplete type. */
#ifdef __cplusplus
typedef class VideoFrameTransform VideoFrameTransform;
#else
typedef struct VideoFrameTransform VideoFrameTransform;
#endif

/* reference Handler.h:24 / Handler.cpp:18-20 */
extern VideoFrameTransform* VideoFrameTransform_new(FrameTransformContext* ctx);

/* reference Handler.h:26 / Handler.cpp:22-24 */
extern void VideoFrameTransform_delete(VideoFrameTransform* transform);

/* reference Handler.h:28-34 / Handler.cpp:26-39 -> VideoFrameTransform.cpp:504-576 */
extern int VideoFrameTransform_generateMapForPlane(
    VideoFrameTransform* transform,
    int inputWidth,
    int inputHeight,
    int outputWidth,
    int outputHeight,
    int transformMatPlaneIndex);

/* reference Handler.h:36-47 / Handler.cpp:41-64 -> VideoFrameTransform.cpp:1319-1351 */
extern int VideoFrameTransform_transformFramePlane(
    VideoFrameTransform* transform,
    uint8_t* inputData,
    uint8_t* outputData,
    int inputWidth,
    int inputHeight,
    int inputWidthWithPadding,
    int outputWidth,
    int outputHeight,
    int outputWidthWithPadding,
    int transformMatPlaneIndex,
    int imagePlaneIndex);

#ifdef __cplusplus
}
#endif

#endif /* TRANSFORM360_VIDEOFRAMETRANSFORMHANDLER_H */
