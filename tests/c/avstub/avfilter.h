/*
 * tests/c/avstub -- TEST-ONLY stand-ins for the handful of libavfilter / libavutil declarations the reference's
 * ffmpeg filter (Transform360/vf_transform360.c) uses, so that the UNMODIFIED file can be compiled and linked
 * against libTransform360.so on a machine without ffmpeg (tests/test_filter_link.py).  Written from the public
 * FFmpeg API (struct and field names the filter touches); nothing here ships.
 */
#ifndef T360_AVSTUB_AVFILTER_H
#define T360_AVSTUB_AVFILTER_H

#include <errno.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define LIBAVUTIL_VERSION_INT ((56 << 16) | (70 << 8) | 100)
#define AVERROR(e) (-(e))
#define av_cold
#define NULL_IF_CONFIG_SMALL(x) x
#define FF_CEIL_RSHIFT(a, b) (-((-(a)) >> (b)))
#define FFSWAP(type, a, b) do { type SWAP_tmp = b; b = a; a = SWAP_tmp; } while (0)

enum { AV_LOG_ERROR = 16, AV_LOG_INFO = 32, AV_LOG_VERBOSE = 40 };
enum AVMediaType { AVMEDIA_TYPE_VIDEO = 0 };
enum { AV_CLASS_CATEGORY_FILTER = 4 };
enum AVPixelFormat { AV_PIX_FMT_YUV420P = 0, AV_PIX_FMT_GRAY8 = 8, AV_PIX_FMT_YUV444P = 5, AV_PIX_FMT_YUV422P = 4, AV_PIX_FMT_YUVA420P = 33 };

enum AVOptionType {
  AV_OPT_TYPE_FLAGS, AV_OPT_TYPE_INT, AV_OPT_TYPE_INT64, AV_OPT_TYPE_DOUBLE, AV_OPT_TYPE_FLOAT, AV_OPT_TYPE_STRING,
  AV_OPT_TYPE_RATIONAL, AV_OPT_TYPE_BINARY, AV_OPT_TYPE_DICT, AV_OPT_TYPE_UINT64, AV_OPT_TYPE_CONST, AV_OPT_TYPE_BOOL = 32
};
#define AV_OPT_FLAG_VIDEO_PARAM 16
#define AV_OPT_FLAG_FILTERING_PARAM (1 << 16)

typedef struct AVOption {
  const char* name;
  const char* help;
  int offset;
  enum AVOptionType type;
  union {
    int64_t i64;
    double dbl;
    const char* str;
  } default_val;
  double min, max;
  int flags;
  const char* unit;
} AVOption;

typedef struct AVClass {
  const char* class_name;
  const char* (*item_name)(void* ctx);
  const AVOption* option;
  int version;
  int category;
} AVClass;

typedef struct AVDictionary AVDictionary;

typedef struct AVPixFmtDescriptor {
  const char* name;
  uint8_t nb_components;
  uint8_t log2_chroma_w, log2_chroma_h;
} AVPixFmtDescriptor;

typedef struct AVFrame {
  uint8_t* data[8];
  int linesize[8];
  int width, height, format;
  int64_t pts;
} AVFrame;

struct AVFilterContext;
struct AVFilterLink;

typedef struct AVFilterPad {
  const char* name;
  enum AVMediaType type;
  int (*filter_frame)(struct AVFilterLink* link, AVFrame* frame);
  int (*config_props)(struct AVFilterLink* link);
} AVFilterPad;

typedef struct AVFilter {
  const char* name;
  const char* description;
  const AVFilterPad* inputs;
  const AVFilterPad* outputs;
  const AVClass* priv_class;
  int (*init_dict)(struct AVFilterContext* ctx, AVDictionary** options);
  void (*uninit)(struct AVFilterContext* ctx);
  int priv_size;
} AVFilter;

typedef struct AVFilterLink {
  struct AVFilterContext* src;
  struct AVFilterContext* dst;
  int w, h, format;
} AVFilterLink;

typedef struct AVFilterContext {
  const AVClass* av_class;
  const AVFilter* filter;
  AVFilterLink** inputs;
  AVFilterLink** outputs;
  void* priv;
} AVFilterContext;

void av_log(void* avcl, int level, const char* fmt, ...);
const char* av_default_item_name(void* ctx);
const AVPixFmtDescriptor* av_pix_fmt_desc_get(int pix_fmt);
int av_pix_fmt_count_planes(int pix_fmt);
int av_expr_parse_and_eval(double* res, const char* s, const char* const* const_names, const double* const_values,
                           const char* const* func1_names, double (*const* funcs1)(void*, double),
                           const char* const* func2_names, double (*const* funcs2)(void*, double, double), void* opaque,
                           int log_offset, void* log_ctx);
void av_dict_free(AVDictionary** m);
void av_frame_free(AVFrame** frame);
int av_frame_copy_props(AVFrame* dst, const AVFrame* src);
AVFrame* ff_get_video_buffer(AVFilterLink* link, int w, int h);
int ff_filter_frame(AVFilterLink* link, AVFrame* frame);
#define av_assert1(cond) ((void)0)

#endif
