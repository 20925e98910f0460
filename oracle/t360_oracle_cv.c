/*
 * oracle/t360_oracle_cv.c -- CPU ORACLE (test infrastructure only; see t360_oracle.h)
 *
 * ** PARITY UNPINNED at this boundary. **
 * The per-frame arithmetic of the reference lives in OpenCV (cv::remap at
 * VideoFrameTransform.cpp:748-754, cv::sepFilter2D at :189-197), an un-vendored dependency
 * with no version pin (CMakeLists.txt:11) that is not installed in this image; the reference
 * ships no tests or golden images.  This file restates the published OpenCV 4.x algorithms
 * (modules/imgproc/src/imgwarp.cpp: initInterTab1D/2D, RemapInvoker, remapNearest/Bilinear/
 * Bicubic/Lanczos4; modules/imgproc/src/filter.dispatch.cpp + filter.simd.hpp: getKernelType,
 * createSeparableLinearFilter, FilterEngine ROI rules) following SURVEY.md Appendix A.
 * Known residual uncertainty (each <= 1 LSB, on exact ties only): OpenCV's SIMD column pass
 * of the fixed-point separable filter rounds half-to-even where this integer form rounds
 * half-up; the float filter path's tap pairing order; the scalar tail of the 2 x 2 INTER_AREA
 * fast path.  t360o_set_cv_variant() switches the first and the last to what a SIMD build of
 * OpenCV 4.x computes (SymmColumnVec_32s8u; ResizeAreaFastVec's scalar remainder), so that the
 * uncertainty can be COUNTED on real frames (tests/test_oracle_variants.py).
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "t360_oracle.h"

#define INTER_BITS 5
#define INTER_TAB_SIZE 32
#define INTER_TAB_SIZE2 (INTER_TAB_SIZE * INTER_TAB_SIZE)
#define INTER_REMAP_COEF_BITS 15
#define INTER_REMAP_COEF_SCALE (1 << INTER_REMAP_COEF_BITS)

#ifndef CV_PI
#define CV_PI 3.1415926535897932384626433832795
#endif

/* cvRound: round half to even under the default rounding mode */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline short sat_s16(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
/* saturate_cast<short>(float) = saturate(cvRound(v)) */
static inline short sat_s16_f(float v) { return sat_s16(cv_round_f(v)); }

/* cv::borderInterpolate (core/src/copy.cpp) */
int t360o_border_interpolate(int p, int len, int borderType) {
  if ((unsigned)p < (unsigned)len) return p;
  if (borderType == T360O_BORDER_REPLICATE) {
    p = p < 0 ? 0 : len - 1;
  } else if (borderType == T360O_BORDER_REFLECT || borderType == T360O_BORDER_REFLECT_101) {
    int delta = borderType == T360O_BORDER_REFLECT_101;
    if (len == 1) return 0;
    do {
      if (p < 0)
        p = -p - 1 + delta;
      else
        p = len - 1 - (p - len) - delta;
    } while ((unsigned)p >= (unsigned)len);
  } else if (borderType == T360O_BORDER_WRAP) {
    if (p < 0) p -= ((p - len + 1) / len) * len;
    if (p >= len) p %= len;
  } else if (borderType == T360O_BORDER_CONSTANT) {
    p = -1;
  }
  return p;
}

/* ---- 1-D coefficient generators (imgwarp.cpp) ---- */
static void interpolate_linear(float x, float* c) {
  c[0] = 1.f - x;
  c[1] = x;
}
static void interpolate_cubic(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}
static void interpolate_lanczos4(float x, float* c) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[][2] = {{1, 0},  {-s45, -s45}, {0, 1},  {s45, -s45},
                                 {-1, 0}, {s45, s45},   {0, -1}, {-s45, s45}};
  if (x < FLT_EPSILON) {
    for (int i = 0; i < 8; i++) c[i] = 0;
    c[3] = 1;
    return;
  }
  float sum = 0;
  double y0 = -(x + 3) * CV_PI * 0.25, s0 = sin(y0), c0 = cos(y0);
  for (int i = 0; i < 8; i++) {
    double y = -(x + 3 - i) * CV_PI * 0.25;
    c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += c[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; i++) c[i] *= sum;
}

/* ---- 2-D fixed-point table (initInterTab2D) ---- */
static int16_t g_tab_linear[INTER_TAB_SIZE2 * 4];
static int16_t g_tab_cubic[INTER_TAB_SIZE2 * 16];
static int16_t g_tab_lanczos[INTER_TAB_SIZE2 * 64];
static int g_tab_ready[5];
static pthread_mutex_t g_tab_mu = PTHREAD_MUTEX_INITIALIZER;

static void build_tab(int interp, int16_t* itab_base, int ksize) {
  float tab1[8 * INTER_TAB_SIZE];
  float scale = 1.f / INTER_TAB_SIZE;
  for (int i = 0; i < INTER_TAB_SIZE; i++) {
    float* t = tab1 + i * ksize;
    if (interp == LINEAR)
      interpolate_linear(i * scale, t);
    else if (interp == CUBIC)
      interpolate_cubic(i * scale, t);
    else
      interpolate_lanczos4(i * scale, t);
  }
  for (int i = 0; i < INTER_TAB_SIZE; i++)
    for (int j = 0; j < INTER_TAB_SIZE; j++) {
      int16_t* itab = itab_base + (size_t)(i * INTER_TAB_SIZE + j) * ksize * ksize;
      int isum = 0;
      for (int k1 = 0; k1 < ksize; k1++) {
        float vy = tab1[i * ksize + k1];
        for (int k2 = 0; k2 < ksize; k2++) {
          float v = vy * tab1[j * ksize + k2];
          isum += itab[k1 * ksize + k2] = sat_s16_f(v * INTER_REMAP_COEF_SCALE);
        }
      }
      if (isum != INTER_REMAP_COEF_SCALE) {
        int diff = isum - INTER_REMAP_COEF_SCALE;
        int ksize2 = ksize / 2, Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
        /* For ksize == 2 OpenCV's scan indexes past the 2x2 block; that case cannot occur
         * because bilinear products (k/32)*(m/32)*32768 are integers summing to 32768. */
        for (int k1 = ksize2; k1 < ksize2 + 2; k1++)
          for (int k2 = ksize2; k2 < ksize2 + 2; k2++) {
            if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2])
              mk1 = k1, mk2 = k2;
            else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2])
              Mk1 = k1, Mk2 = k2;
          }
        if (diff < 0)
          itab[Mk1 * ksize + Mk2] = (short)(itab[Mk1 * ksize + Mk2] - diff);
        else
          itab[mk1 * ksize + mk2] = (short)(itab[mk1 * ksize + mk2] - diff);
      }
    }
}

const int16_t* t360o_inter_tab(int interp, int* ksize) {
  int16_t* tab;
  int ks;
  switch (interp) {
    case LINEAR: tab = g_tab_linear; ks = 2; break;
    case CUBIC: tab = g_tab_cubic; ks = 4; break;
    case LANCZOS4: tab = g_tab_lanczos; ks = 8; break;
    default: return NULL;
  }
  if (!__atomic_load_n(&g_tab_ready[interp], __ATOMIC_ACQUIRE)) {
    /* remap stripes run on several threads: build each table exactly once */
    pthread_mutex_lock(&g_tab_mu);
    if (!g_tab_ready[interp]) {
      build_tab(interp, tab, ks);
      __atomic_store_n(&g_tab_ready[interp], 1, __ATOMIC_RELEASE);
    }
    pthread_mutex_unlock(&g_tab_mu);
  }
  if (ksize) *ksize = ks;
  return tab;
}

/* ---- cv::remap, CV_8UC1 source, CV_32FC2 map, no second map ---- */
void t360o_remap_rows(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw,
                      int dh, size_t dstep, const float* map, int interp, int borderType,
                      int row0, int row1) {
  (void)dh;
  if (interp == NEAREST) {
    /* RemapInvoker: XY = saturate_cast<short>(map); remapNearest */
    for (int dy = row0; dy < row1; dy++) {
      const float* m = map + (size_t)dy * dw * 2;
      uint8_t* D = dst + (size_t)dy * dstep;
      for (int dx = 0; dx < dw; dx++) {
        int sx = sat_s16_f(m[dx * 2]);
        int sy = sat_s16_f(m[dx * 2 + 1]);
        if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) {
          D[dx] = src[(size_t)sy * sstep + sx];
        } else if (borderType == T360O_BORDER_REPLICATE) {
          sx = sx < 0 ? 0 : (sx >= sw ? sw - 1 : sx);
          sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
          D[dx] = src[(size_t)sy * sstep + sx];
        } else if (borderType == T360O_BORDER_CONSTANT) {
          D[dx] = 0;
        } else if (borderType != T360O_BORDER_TRANSPARENT) {
          sx = t360o_border_interpolate(sx, sw, borderType);
          sy = t360o_border_interpolate(sy, sh, borderType);
          D[dx] = src[(size_t)sy * sstep + sx];
        }
      }
    }
    return;
  }

  int ksize;
  const int16_t* wtab = t360o_inter_tab(interp, &ksize);
  if (!wtab) return;
  const int half = ksize / 2 - 1; /* footprint origin = (ix - half, iy - half) */
  const int borderType1 = borderType != T360O_BORDER_TRANSPARENT ? borderType : T360O_BORDER_REFLECT_101;
  /* fast-path extent: remapBilinear uses (w-1,h-1) on the top-left tap, bicubic (w-3,h-3),
   * lanczos4 (w-7,h-7): footprint entirely inside */
  const unsigned width1 = (unsigned)(sw - (ksize - 1) > 0 ? sw - (ksize - 1) : 0);
  const unsigned height1 = (unsigned)(sh - (ksize - 1) > 0 ? sh - (ksize - 1) : 0);

  for (int dy = row0; dy < row1; dy++) {
    const float* m = map + (size_t)dy * dw * 2;
    uint8_t* D = dst + (size_t)dy * dstep;
    for (int dx = 0; dx < dw; dx++) {
      /* RemapInvoker, CV_32FC2 branch */
      int sxq = cv_round_f(m[dx * 2] * INTER_TAB_SIZE);
      int syq = cv_round_f(m[dx * 2 + 1] * INTER_TAB_SIZE);
      int fxy = (syq & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sxq & (INTER_TAB_SIZE - 1));
      int cx = sat_s16(sxq >> INTER_BITS);
      int cy = sat_s16(syq >> INTER_BITS);
      const int16_t* w = wtab + (size_t)fxy * ksize * ksize;
      int sx = cx - half, sy = cy - half;
      int sum = 0;
      if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        const uint8_t* S = src + (size_t)sy * sstep + sx;
        for (int r = 0; r < ksize; r++, S += sstep, w += ksize)
          for (int c = 0; c < ksize; c++) sum += S[c] * w[c];
      } else {
        if (borderType == T360O_BORDER_TRANSPARENT) {
          if (ksize == 2) continue; /* remapBilinear skips every outlier for cn != 3 */
          if ((unsigned)cx >= (unsigned)sw || (unsigned)cy >= (unsigned)sh) continue;
        }
        if (borderType1 == T360O_BORDER_CONSTANT &&
            (sx >= sw || sx + ksize <= 0 || sy >= sh || sy + ksize <= 0)) {
          D[dx] = 0;
          continue;
        }
        int xi[8], yi[8];
        for (int i = 0; i < ksize; i++) {
          xi[i] = t360o_border_interpolate(sx + i, sw, borderType1);
          yi[i] = t360o_border_interpolate(sy + i, sh, borderType1);
        }
        for (int r = 0; r < ksize; r++, w += ksize) {
          if (yi[r] < 0) continue;
          const uint8_t* S = src + (size_t)yi[r] * sstep;
          for (int c = 0; c < ksize; c++)
            if (xi[c] >= 0) sum += S[xi[c]] * w[c];
        }
      }
      /* FixedPtCast<int, uchar, INTER_REMAP_COEF_BITS> */
      D[dx] = sat_u8((sum + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS);
    }
  }
}

/* ---- cv::getKernelType, anchor = kernel centre ---- */
enum { K_SYMMETRICAL = 1, K_ASYMMETRICAL = 2, K_SMOOTH = 4, K_INTEGER = 8 };
int t360o_kernel_type(const float* k, int len) {
  double sum = 0;
  int type = K_SMOOTH + K_INTEGER;
  /* 1-D kernel with anchor*2+1 == len (len is always odd here) */
  if ((len & 1) == 1) type |= (K_SYMMETRICAL + K_ASYMMETRICAL);
  for (int i = 0; i < len; i++) {
    double a = k[i], b = k[len - i - 1];
    if (a != b) type &= ~K_SYMMETRICAL;
    if (a != -b) type &= ~K_ASYMMETRICAL;
    if (a < 0) type &= ~K_SMOOTH;
    if (a != (double)cv_round_d(a)) type &= ~K_INTEGER;
    sum += a;
  }
  if (fabs(sum - 1) > FLT_EPSILON * (fabs(sum) + 1)) type &= ~K_SMOOTH;
  return type;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Which of two published evaluation orders the restatement follows where a SIMD build of OpenCV 4.x and its scalar
 * code differ on exact ties (process-wide; test infrastructure).
 *   column_simd_lanes > 0: the fixed-point column pass of cv::sepFilter2D as SymmColumnVec_32s8u does it for the first
 *     width - width % lanes pixels of every filtered ROI row (filter.simd.hpp: the int32 row sums converted to float, the
 *     Q8 taps as float(k / 65536), v_muladd, v_round = round half to EVEN, saturating pack) and as FixedPtCastEx does it
 *     for the remaining pixels ((s + 32768) >> 16: round half UP).  v_uint8::nlanes = 16 for the 128-bit baseline
 *     (SSE2 / NEON), 32 for an AVX2 baseline.  0 (default): the integer form everywhere.
 *   area_tail_lanes > 0: the 2 x 2 INTER_AREA fast path as ResizeAreaFastVec_SIMD_8u + its scalar remainder: the first
 *     dw - dw % lanes pixels of a row (sum + 2) >> 2, the rest saturate_cast<uchar>(sum * 0.25f) = round half to even
 *     (resize.cpp; v_uint16::nlanes = 8 for the 128-bit baseline).  0 (default): (sum + 2) >> 2 everywhere. */
static int g_column_simd_lanes = 0, g_area_tail_lanes = 0;
void t360o_set_cv_variant(int column_simd_lanes, int area_tail_lanes) {
  g_column_simd_lanes = column_simd_lanes;
  g_area_tail_lanes = area_tail_lanes;
}

/* ---- cv::sepFilter2D on an ROI of a parent image (no BORDER_ISOLATED) ---- */
int t360o_sepfilter_roi(const uint8_t* parent, int pw, int ph, size_t pstep, uint8_t* dparent,
                        size_t dstep, int left, int top, int width, int height, const float* kx,
                        int kx_len, const float* ky, int ky_len) {
  /* cv::Mat::operator()(Rect) asserts the ROI is inside the matrix */
  if (left < 0 || top < 0 || width < 0 || height < 0 || left + width > pw || top + height > ph)
    return -1;
  if (width == 0 || height == 0) return 1;
  const int rx = kx_len / 2, ry = ky_len / 2; /* anchor (-1,-1) -> centre */
  const int rtype = t360o_kernel_type(kx, kx_len);
  const int ctype = t360o_kernel_type(ky, ky_len);
  const int fixedpt = rtype == (K_SMOOTH + K_SYMMETRICAL) && ctype == (K_SMOOTH + K_SYMMETRICAL);
  const int nrows = height + 2 * ry;

  if (fixedpt) {
    int* kxi = (int*)malloc(sizeof(int) * (size_t)kx_len);
    int* kyi = (int*)malloc(sizeof(int) * (size_t)ky_len);
    /* Mat::convertTo(CV_32S, 256): saturate_cast<int>(v * 256) in double -> cvRound */
    for (int i = 0; i < kx_len; i++) kxi[i] = cv_round_d((double)kx[i] * 256.0);
    for (int i = 0; i < ky_len; i++) kyi[i] = cv_round_d((double)ky[i] * 256.0);
    int* rows = (int*)malloc(sizeof(int) * (size_t)nrows * (size_t)width);
    for (int r = 0; r < nrows; r++) {
      const uint8_t* S = parent + (size_t)clampi(top - ry + r, 0, ph - 1) * pstep;
      int* R = rows + (size_t)r * width;
      for (int x = 0; x < width; x++) {
        int s = 0;
        for (int k = 0; k < kx_len; k++) s += kxi[k] * S[clampi(left + x - rx + k, 0, pw - 1)];
        R[x] = s;
      }
    }
    /* SymmColumnVec_32s8u: symmetric kernels only (the fixed-point path implies it), ksize >= 3 */
    const int simd_w = (g_column_simd_lanes > 0 && ky_len >= 3) ? width - width % g_column_simd_lanes : 0;
    for (int y = 0; y < height; y++) {
      uint8_t* D = dparent + (size_t)(top + y) * dstep + left;
      for (int x = 0; x < simd_w; x++) {
        /* kernel.convertTo(CV_32F, 1. / (1 << 16)); s = S[0] * ky[0] + delta; s += (S[k] + S[-k]) * ky[k], k = 1..ry */
        float sf = (float)rows[(size_t)(y + ry) * width + x] * (float)((double)kyi[ry] * (1.0 / 65536.0)) + 0.f;
        for (int k = 1; k <= ry; k++)
          sf = (float)(rows[(size_t)(y + ry + k) * width + x] + rows[(size_t)(y + ry - k) * width + x]) *
                   (float)((double)kyi[ry + k] * (1.0 / 65536.0)) + sf;
        D[x] = sat_u8(cv_round_f(sf));
      }
      for (int x = simd_w; x < width; x++) {
        int s = 0;
        for (int k = 0; k < ky_len; k++) s += kyi[k] * rows[(size_t)(y + k) * width + x];
        /* FixedPtCastEx<int, uchar>(16) */
        D[x] = sat_u8((s + (1 << 15)) >> 16);
      }
    }
    free(rows);
    free(kxi);
    free(kyi);
    return 1;
  }

  /* float path: RowFilter<uchar,float> then (Symm)ColumnFilter<Cast<float,uchar>> */
  float* rows = (float*)malloc(sizeof(float) * (size_t)nrows * (size_t)width);
  for (int r = 0; r < nrows; r++) {
    const uint8_t* S = parent + (size_t)clampi(top - ry + r, 0, ph - 1) * pstep;
    float* R = rows + (size_t)r * width;
    for (int x = 0; x < width; x++) {
      float s = kx[0] * S[clampi(left + x - rx, 0, pw - 1)];
      for (int k = 1; k < kx_len; k++) s += kx[k] * S[clampi(left + x - rx + k, 0, pw - 1)];
      R[x] = s;
    }
  }
  const int symmetric = (ctype & K_SYMMETRICAL) != 0;
  for (int y = 0; y < height; y++) {
    uint8_t* D = dparent + (size_t)(top + y) * dstep + left;
    for (int x = 0; x < width; x++) {
      float s;
      if (symmetric) {
        s = ky[ry] * rows[(size_t)(y + ry) * width + x];
        for (int k = 1; k <= ry; k++)
          s += ky[ry + k] * (rows[(size_t)(y + ry + k) * width + x] + rows[(size_t)(y + ry - k) * width + x]);
      } else {
        s = ky[0] * rows[(size_t)y * width + x];
        for (int k = 1; k < ky_len; k++) s += ky[k] * rows[(size_t)(y + k) * width + x];
      }
      D[x] = sat_u8(cv_round_f(s));
    }
  }
  free(rows);
  return 0;
}


/* ---------------------------------------------------------------------------------------------
 * cv::resize(src, dst, dsize, 0, 0, INTER_AREA) for CV_8UC1, the shrinking case the reference
 * uses for its supersample-then-decimate antialiasing (VideoFrameTransform.cpp:770-776).
 * Restated from OpenCV 4.x modules/imgproc/src/resize.cpp (PARITY UNPINNED, like the rest of
 * this file):
 *   scale_x = 1 / ((double)dw / sw), scale_y likewise; only scale >= 1 on both axes is restated
 *   (INTER_AREA enlargement is a different, bilinear-like code path; returns 0 = unsupported).
 *   a) both scales integers (|scale - (int)scale| < DBL_EPSILON), "ResizeAreaFast":
 *        2 x 2      : (a + b + c + d + 2) >> 2                       (ResizeAreaFastVec / its scalar tail)
 *        otherwise  : saturate_cast<uchar>(int_sum * (1.f / area))   float multiply, cvRound (half-even)
 *      the source is an exact multiple of the destination here, so the partial-cell tail of the
 *      OpenCV loop is never reached.
 *   b) otherwise, "ResizeArea": per axis a DecimateAlpha table (computeResizeAreaTab);
 *        buf[dx] = SUM_k S[si_k] * alpha_k  (float, table order),  sum[dx] = SUM_j beta_j * buf_j[dx]
 *        (float, row order, first term assigned), D = saturate_cast<uchar>(sum).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int si, di;
  float alpha;
} DecimateAlpha;

static int resize_area_tab(int ssize, int dsize, double scale, DecimateAlpha* tab) {
  int k = 0;
  for (int dx = 0; dx < dsize; dx++) {
    double fsx1 = dx * scale;
    double fsx2 = fsx1 + scale;
    double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1; /* std::min(scale, ssize - fsx1) */
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    if (sx1 - fsx1 > 1e-3) {
      tab[k].di = dx;
      tab[k].si = sx1 - 1;
      tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth);
    }
    for (int sx = sx1; sx < sx2; sx++) {
      tab[k].di = dx;
      tab[k].si = sx;
      tab[k++].alpha = (float)(1.0 / cellWidth);
    }
    if (fsx2 - sx2 > 1e-3) {
      double a = fsx2 - sx2;
      a = a < 1.0 ? a : 1.0;
      a = a < cellWidth ? a : cellWidth;
      tab[k].di = dx;
      tab[k].si = sx2;
      tab[k++].alpha = (float)(a / cellWidth);
    }
  }
  return k;
}

/* cv::resize(INTER_AREA) when either factor ENLARGES (scale < 1): OpenCV has true area interpolation only for
 * scale_x >= 1 && scale_y >= 1 and otherwise "emulates it using some variant of bilinear interpolation"
 * (resize.cpp, the general ksize = 2 path with area_mode coefficients): per destination column
 *     sx = floor(dx * scale_x),  fx = (float)((dx + 1) - (sx + 1) * inv_scale_x),  fx = fx <= 0 ? 0 : fx - floor(fx)
 * (same for rows), 11-bit fixed-point coefficients saturate_cast<short>(c * 2048), then the 8-bit bilinear kernels
 *     HResizeLinear:  D = S[sx] * a0 + S[sx + 1] * a1      (dx < xmax;  D = S[sx] * 2048 beyond it)
 *     VResizeLinear:  dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2
 * with source rows clamped to the plane.  Restated from the published OpenCV 4.x algorithm; PARITY UNPINNED like
 * the rest of this file (no OpenCV in this image). */
static int resize_area_as_linear(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh,
                                 size_t dstep) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1.0 / inv_scale_x, scale_y = 1.0 / inv_scale_y;
  int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
  short* ialpha = (short*)malloc(sizeof(short) * (size_t)dw * 2);
  int* hrow[2];
  hrow[0] = (int*)malloc(sizeof(int) * (size_t)dw);
  hrow[1] = (int*)malloc(sizeof(int) * (size_t)dw);
  if (!xofs || !ialpha || !hrow[0] || !hrow[1]) {
    free(xofs); free(ialpha); free(hrow[0]); free(hrow[1]);
    return 0;
  }
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    int sx = (int)floor(dx * scale_x);
    float fx = (float)((dx + 1) - (sx + 1) * inv_scale_x);
    fx = fx <= 0 ? 0.f : fx - floorf(fx);
    if (sx + 1 >= sw) { /* ksize2 = 1 */
      if (dx < xmax) xmax = dx;
      if (sx >= sw - 1) fx = 0, sx = sw - 1;
    }
    xofs[dx] = sx;
    const float c0 = 1.f - fx, c1 = fx;
    long a0 = lrintf(c0 * 2048.f), a1 = lrintf(c1 * 2048.f);
    ialpha[2 * dx] = (short)(a0 > 32767 ? 32767 : a0);
    ialpha[2 * dx + 1] = (short)(a1 > 32767 ? 32767 : a1);
  }
  for (int dy = 0; dy < dh; dy++) {
    const int sy = (int)floor(dy * scale_y);
    float fy = (float)((dy + 1) - (sy + 1) * inv_scale_y);
    fy = fy <= 0 ? 0.f : fy - floorf(fy);
    const int b0 = (int)lrintf((1.f - fy) * 2048.f), b1 = (int)lrintf(fy * 2048.f);
    for (int k = 0; k < 2; k++) {
      int r = sy + k; /* clip(sy - ksize2 + 1 + k, 0, sh) */
      r = r >= 0 ? (r < sh ? r : sh - 1) : 0;
      const uint8_t* S = src + (size_t)r * sstep;
      int* D = hrow[k];
      int dx = 0;
      for (; dx < xmax; dx++) D[dx] = S[xofs[dx]] * ialpha[2 * dx] + S[xofs[dx] + 1] * ialpha[2 * dx + 1];
      for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
    }
    uint8_t* D = dst + (size_t)dy * dstep;
    for (int dx = 0; dx < dw; dx++)
      D[dx] = (uint8_t)((((b0 * (hrow[0][dx] >> 4)) >> 16) + ((b1 * (hrow[1][dx] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs); free(ialpha); free(hrow[0]); free(hrow[1]);
  return 1;
}

int t360o_resize_area(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh,
                      size_t dstep) {
  if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return 0;
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1.0 / inv_scale_x, scale_y = 1.0 / inv_scale_y;
  if (!(scale_x >= 1 && scale_y >= 1)) return resize_area_as_linear(src, sw, sh, sstep, dst, dw, dh, dstep);
  const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y); /* saturate_cast<int>(double) */
  const int fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
  if (fast && (size_t)iscale_x * dw == (size_t)sw && (size_t)iscale_y * dh == (size_t)sh) {
    const int area = iscale_x * iscale_y;
    const float scale = 1.f / area;
    for (int dy = 0; dy < dh; dy++) {
      uint8_t* D = dst + (size_t)dy * dstep;
      for (int dx = 0; dx < dw; dx++) {
        int sum = 0;
        for (int sy = 0; sy < iscale_y; sy++) {
          const uint8_t* S = src + (size_t)(dy * iscale_y + sy) * sstep + (size_t)dx * iscale_x;
          for (int sx = 0; sx < iscale_x; sx++) sum += S[sx];
        }
        if (iscale_x == 2 && iscale_y == 2 && !(g_area_tail_lanes > 0 && dx >= dw - dw % g_area_tail_lanes)) {
          D[dx] = (uint8_t)((sum + 2) >> 2);
        } else {
          long v = lrintf((float)sum * scale);
          D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
      }
    }
    return 1;
  }
  DecimateAlpha* xt = (DecimateAlpha*)malloc(sizeof(DecimateAlpha) * (size_t)sw * 2);
  DecimateAlpha* yt = (DecimateAlpha*)malloc(sizeof(DecimateAlpha) * (size_t)sh * 2);
  float* buf = (float*)malloc(sizeof(float) * (size_t)dw * 2);
  if (!xt || !yt || !buf) {
    free(xt);
    free(yt);
    free(buf);
    return 0;
  }
  float* sum = buf + dw;
  const int nx = resize_area_tab(sw, dw, scale_x, xt);
  const int ny = resize_area_tab(sh, dh, scale_y, yt);
  int prev_dy = yt[0].di;
  for (int dx = 0; dx < dw; dx++) sum[dx] = 0.f;
  for (int j = 0; j < ny; j++) {
    const float beta = yt[j].alpha;
    const int dy = yt[j].di;
    const uint8_t* S = src + (size_t)yt[j].si * sstep;
    for (int dx = 0; dx < dw; dx++) buf[dx] = 0.f;
    for (int k = 0; k < nx; k++) buf[xt[k].di] += S[xt[k].si] * xt[k].alpha;
    if (dy != prev_dy) {
      uint8_t* D = dst + (size_t)prev_dy * dstep;
      for (int dx = 0; dx < dw; dx++) {
        long v = lrintf(sum[dx]);
        D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        sum[dx] = beta * buf[dx];
      }
      prev_dy = dy;
    } else {
      for (int dx = 0; dx < dw; dx++) sum[dx] += beta * buf[dx];
    }
  }
  {
    uint8_t* D = dst + (size_t)prev_dy * dstep;
    for (int dx = 0; dx < dw; dx++) {
      long v = lrintf(sum[dx]);
      D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
  free(xt);
  free(yt);
  free(buf);
  return 1;
}
