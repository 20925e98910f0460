/*
 * oracle/t360_oracle_frame.c -- CPU ORACLE (test infrastructure only; see t360_oracle.h)
 *
 * The reference's per-frame orchestration with its call protocol and threading structure:
 *   transformFramePlane  VideoFrameTransform.cpp:1319-1351
 *   transformPlane       :707-794   (optional low-pass, then remap with BORDER_WRAP, or
 *                                    BORDER_TRANSPARENT for the barrel layouts)
 *   filterPlane          :621-704   (zeroed plane; segments applied per eye for LR/TB input)
 *   runFiltering         :579-618   (one std::thread per segment when enable_multi_threading)
 *   filterSegment        :173-204   (sepFilter2D on an ROI of the whole plane)
 * cv::remap parallelises over destination row stripes (parallel_for_); here: row stripes over
 * a pthread pool of `threads` workers.  The low-pass runs one task per segment on the same
 * pool (the reference creates one thread per segment per call).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "t360_oracle.h"

#define T360O_MAX_MAPS 8

typedef struct PlaneState {
  int valid;
  int in_w, in_h, out_w, out_h; /* as passed to generateMapForPlane */
  int map_w, map_h;             /* scaled size = warp map size */
  float* map;
  T360OFilterConfig segs;
} PlaneState;

struct T360Oracle {
  FrameTransformContext ctx;
  int threads;
  PlaneState plane[T360O_MAX_MAPS];
};

T360Oracle* t360o_new(const FrameTransformContext* ctx) {
  T360Oracle* o = (T360Oracle*)calloc(1, sizeof(T360Oracle));
  if (!o) return NULL;
  memcpy(&o->ctx, ctx, sizeof(*ctx)); /* VideoFrameTransform.cpp:206-208 */
  o->threads = 0;
  return o;
}

void t360o_delete(T360Oracle* o) {
  if (!o) return;
  for (int i = 0; i < T360O_MAX_MAPS; i++) {
    free(o->plane[i].map);
    t360o_filter_config_free(&o->plane[i].segs);
  }
  free(o);
}

void t360o_set_threads(T360Oracle* o, int threads) { o->threads = threads; }

static int resolve_threads(const T360Oracle* o) {
  if (o->threads > 0) return o->threads;
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  return n > 0 ? (int)n : 1;
}

/* VideoFrameTransform.cpp:504-576 */
int t360o_generateMapForPlane(T360Oracle* o, int inW, int inH, int outW, int outH, int mapIdx) {
  if (mapIdx < 0 || mapIdx >= T360O_MAX_MAPS) return 0;
  PlaneState* p = &o->plane[mapIdx];
  int sw, sh;
  if (!t360o_scaled_size(&o->ctx, outW, outH, &sw, &sh)) return 0;
  float* map = (float*)malloc(sizeof(float) * 2 * (size_t)sw * (size_t)sh);
  if (!map) return 0;
  if (!t360o_generate_map(&o->ctx, inW, inH, outW, outH, map)) {
    free(map);
    return 0;
  }
  free(p->map); /* warpMats_[idx] = warpMat replaces (:556) */
  p->map = map;
  p->map_w = sw;
  p->map_h = sh;
  p->in_w = inW;
  p->in_h = inH;
  p->out_w = outW;
  p->out_h = outH;
  p->valid = 1;
  if (o->ctx.enable_low_pass_filter) {
    /* the reference APPENDS on a repeated call (emplace_back, :237/:290-294); the duplicates
     * recompute identical output, so replacing is result-identical */
    t360o_filter_config_free(&p->segs);
    if (!t360o_filter_config(&o->ctx, inW, inH, sw, sh, &p->segs)) {
      /* the reference has stored the warp map by now (:556) and fails in calcualteFilteringConfig (:571-576) */
      printf("Could not generate map for plane %d. Error: kernel of negative length\n", mapIdx);
      return 0;
    }
  }
  return 1;
}

const float* t360o_map(const T360Oracle* o, int mapIdx, int* w, int* h) {
  const PlaneState* p = &o->plane[mapIdx];
  if (!p->valid) return NULL;
  if (w) *w = p->map_w;
  if (h) *h = p->map_h;
  return p->map;
}
const T360OFilterConfig* t360o_segments(const T360Oracle* o, int mapIdx) {
  return &o->plane[mapIdx].segs;
}

/* ---- a minimal task pool: tasks are claimed with an atomic counter ---- */
typedef struct Job {
  void (*fn)(void* arg, int task);
  void* arg;
  int ntasks;
  int next;
} Job;

static void* job_worker(void* a) {
  Job* j = (Job*)a;
  for (;;) {
    int t = __sync_fetch_and_add(&j->next, 1);
    if (t >= j->ntasks) break;
    j->fn(j->arg, t);
  }
  return NULL;
}

/* Persistent worker pool (OpenCV's parallel_for_ also keeps its workers alive between calls).
 * One job at a time; callers are serialised by run_mu. */
#define T360O_MAX_WORKERS 512
static struct {
  pthread_mutex_t mu;
  pthread_cond_t cv_work, cv_done;
  pthread_mutex_t run_mu;
  pthread_t th[T360O_MAX_WORKERS];
  int nworkers;
  Job* job;
  unsigned generation;
  int want;    /* workers with id < want take part in the current job */
  int pending; /* participants that have not finished yet */
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER,
            PTHREAD_MUTEX_INITIALIZER, {0}, 0, NULL, 0, 0, 0};

static void* pool_worker(void* a) {
  const int id = (int)(intptr_t)a;
  unsigned seen = 0;
  pthread_mutex_lock(&g_pool.mu);
  for (;;) {
    while (g_pool.generation == seen) pthread_cond_wait(&g_pool.cv_work, &g_pool.mu);
    seen = g_pool.generation;
    if (id >= g_pool.want) continue;
    Job* j = g_pool.job;
    pthread_mutex_unlock(&g_pool.mu);
    job_worker(j);
    pthread_mutex_lock(&g_pool.mu);
    if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.cv_done);
  }
  return NULL;
}

static void run_tasks(int threads, int ntasks, void (*fn)(void*, int), void* arg) {
  Job j = {fn, arg, ntasks, 0};
  if (threads <= 1 || ntasks <= 1) {
    job_worker(&j);
    return;
  }
  int helpers = (threads < ntasks ? threads : ntasks) - 1; /* the caller works too */
  if (helpers > T360O_MAX_WORKERS) helpers = T360O_MAX_WORKERS;
  pthread_mutex_lock(&g_pool.run_mu);
  pthread_mutex_lock(&g_pool.mu);
  while (g_pool.nworkers < helpers) {
    if (pthread_create(&g_pool.th[g_pool.nworkers], NULL, pool_worker, (void*)(intptr_t)g_pool.nworkers) != 0) break;
    pthread_detach(g_pool.th[g_pool.nworkers]);
    g_pool.nworkers++;
  }
  if (helpers > g_pool.nworkers) helpers = g_pool.nworkers;
  g_pool.job = &j;
  g_pool.want = helpers;
  g_pool.pending = helpers;
  g_pool.generation++;
  pthread_cond_broadcast(&g_pool.cv_work);
  pthread_mutex_unlock(&g_pool.mu);
  job_worker(&j);
  pthread_mutex_lock(&g_pool.mu);
  while (g_pool.pending > 0) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
  pthread_mutex_unlock(&g_pool.mu);
  pthread_mutex_unlock(&g_pool.run_mu);
}

/* ---- low-pass ---- */
typedef struct LpfArgs {
  const uint8_t* in;
  int w, h;
  size_t instep;
  uint8_t* out;
  size_t outstep;
  const T360OFilterConfig* segs;
  int left_off, top_off;
  int image_plane;
} LpfArgs;

static void lpf_task(void* a, int t) {
  const LpfArgs* g = (const LpfArgs*)a;
  const T360OSegment* s = &g->segs->seg[t];
  int r = t360o_sepfilter_roi(g->in, g->w, g->h, g->instep, g->out, g->outstep,
                              s->left + g->left_off, s->top + g->top_off, s->width, s->height,
                              s->kx, s->kx_len, s->ky, s->ky_len);
  if (r < 0) /* cv::Exception caught and printed by filterSegment (:198-203) */
    printf("Could not filter segment for the plane %d. Error: roi outside the plane\n",
           g->image_plane);
}

/* filterPlane, VideoFrameTransform.cpp:621-704 */
static void filter_plane(T360Oracle* o, const uint8_t* in, int w, int h, size_t instep,
                         uint8_t* out, size_t outstep, int mapIdx, int imagePlane) {
  for (int y = 0; y < h; y++) memset(out + (size_t)y * outstep, 0, (size_t)w); /* Mat::zeros */
  const T360OFilterConfig* segs = &o->plane[mapIdx].segs;
  int threads = o->ctx.enable_multi_threading ? resolve_threads(o) : 1;
  LpfArgs a = {in, w, h, instep, out, outstep, segs, 0, 0, imagePlane};
  switch (o->ctx.input_stereo_format) {
    case STEREO_FORMAT_LR:
      run_tasks(threads, segs->count, lpf_task, &a);
      a.left_off = (int)(0.5 * w);
      run_tasks(threads, segs->count, lpf_task, &a);
      break;
    case STEREO_FORMAT_TB:
      run_tasks(threads, segs->count, lpf_task, &a);
      a.top_off = (int)(0.5 * h);
      run_tasks(threads, segs->count, lpf_task, &a);
      break;
    default:
      run_tasks(threads, segs->count, lpf_task, &a);
      break;
  }
}

int t360o_filterPlane(T360Oracle* o, const uint8_t* in, int inW, int inH, int inStride,
                      uint8_t* dst, int dstStride, int mapIdx) {
  if (mapIdx < 0 || mapIdx >= T360O_MAX_MAPS || !o->plane[mapIdx].valid) return 0;
  filter_plane(o, in, inW, inH, (size_t)inStride, dst, (size_t)dstStride, mapIdx, mapIdx);
  return 1;
}

/* ---- remap over row stripes ---- */
typedef struct RemapArgs {
  const uint8_t* src;
  int sw, sh;
  size_t sstep;
  uint8_t* dst;
  int dw, dh;
  size_t dstep;
  const float* map;
  int interp, border, rows_per_task;
} RemapArgs;

static void remap_task(void* a, int t) {
  const RemapArgs* g = (const RemapArgs*)a;
  int r0 = t * g->rows_per_task;
  int r1 = r0 + g->rows_per_task;
  if (r1 > g->dh) r1 = g->dh;
  t360o_remap_rows(g->src, g->sw, g->sh, g->sstep, g->dst, g->dw, g->dh, g->dstep, g->map,
                   g->interp, g->border, r0, r1);
}

/* transformFramePlane -> transformPlane, VideoFrameTransform.cpp:1319-1351, 707-794 */
int t360o_transformFramePlane(T360Oracle* o, const uint8_t* in, uint8_t* out, int inW, int inH,
                              int inStride, int outW, int outH, int outStride, int mapIdx,
                              int imagePlaneIdx) {
  if (mapIdx < 0 || mapIdx >= T360O_MAX_MAPS || !o->plane[mapIdx].valid) return 0;
  PlaneState* p = &o->plane[mapIdx];
  const int barrel =
      o->ctx.output_layout == LAYOUT_BARREL || o->ctx.output_layout == LAYOUT_BARREL_SPLIT;
  const int border = barrel ? T360O_BORDER_TRANSPARENT : T360O_BORDER_WRAP;
  const int interp = (int)o->ctx.interpolation_alg;
  if (!(interp == NEAREST || interp == LINEAR || interp == CUBIC || interp == LANCZOS4)) {
    printf("Could not find interpolation algorithm for plane %d", imagePlaneIdx); /* :780-783 */
    return 1;
  }

  const uint8_t* src = in;
  size_t sstep = (size_t)inStride;
  uint8_t* blurred = NULL;
  if (o->ctx.enable_low_pass_filter) {
    blurred = (uint8_t*)malloc((size_t)inW * (size_t)inH);
    if (!blurred) return 0;
    filter_plane(o, in, inW, inH, (size_t)inStride, blurred, (size_t)inW, mapIdx, imagePlaneIdx);
    src = blurred;
    sstep = (size_t)inW;
  }

  const int needResize = (outH != p->map_h || outW != p->map_w);
  int threads = resolve_threads(o);
  if (!needResize) {
    if (mapIdx && barrel) /* :743-747 */
      for (int y = 0; y < outH; y++) memset(out + (size_t)y * outStride, 128, (size_t)outW);
    RemapArgs a = {src, inW, inH, sstep, out, outW, outH, (size_t)outStride, p->map, interp, border, 0};
    int stripes = threads > 1 ? threads * 4 : 1;
    if (stripes > outH) stripes = outH;
    a.rows_per_task = (outH + stripes - 1) / stripes;
    int ntasks = (outH + a.rows_per_task - 1) / a.rows_per_task;
    run_tasks(threads, ntasks, remap_task, &a);
  } else {
    /* supersample + cv::resize(INTER_AREA) (:759-776): remap into a map-sized image that starts
     * as Scalar(mapIdx ? 128 : 0), then shrink it into the output plane */
    uint8_t* scaled = (uint8_t*)malloc((size_t)p->map_w * (size_t)p->map_h);
    if (!scaled) {
      free(blurred);
      return 0;
    }
    memset(scaled, mapIdx ? 128 : 0, (size_t)p->map_w * (size_t)p->map_h);
    RemapArgs a = {src, inW, inH, sstep, scaled, p->map_w, p->map_h, (size_t)p->map_w, p->map, interp, border, 0};
    int stripes = threads > 1 ? threads * 4 : 1;
    if (stripes > p->map_h) stripes = p->map_h;
    a.rows_per_task = (p->map_h + stripes - 1) / stripes;
    int ntasks = (p->map_h + a.rows_per_task - 1) / a.rows_per_task;
    run_tasks(threads, ntasks, remap_task, &a);
    int ok = t360o_resize_area(scaled, p->map_w, p->map_h, (size_t)p->map_w, out, outW, outH, (size_t)outStride);
    free(scaled);
    if (!ok) {
      /* the reference would run cv::resize's enlargement path here, which is not restated */
      printf("Could not transform the plane %d. Error: INTER_AREA enlargement is not restated by the oracle\n",
             imagePlaneIdx);
      free(blurred);
      return 0;
    }
  }
  free(blurred);
  return 1;
}
