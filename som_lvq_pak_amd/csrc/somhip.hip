// somhip.hip -- C ABI (include/somhip.h) of the MI355X SOM/LVQ engine: host control
// of the gfx950 kernels in kernels.hpp.  No CPU fallback: every entry point runs on
// the GPU or fails with a message.
#include "../../include/somhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "schedule.hpp"

using namespace somhip;

// ---------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}
#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define CHK(expr)          \
  do {                     \
    int rc_ = (expr);      \
    if (rc_) return rc_;   \
  } while (0)

extern "C" const char *somhip_last_error(void) { return g_err.c_str(); }
extern "C" int somhip_version(void) { return SOMHIP_VERSION; }

// ---------------------------------------------------------------------------------
// kernel ids for the timing table
// ---------------------------------------------------------------------------------
enum KernelId {
  KID_SCAN_EXACT = 0,
  KID_SOM_UPDATE_RUN,
  KID_SOM_ONLINE_STEP,
  KID_LVQ_ONLINE_STEP,
  KID_PACK_SAMPLES,
  KID_MERGE_TOPK,
  KID_SCAN_MASKED,
  KID_LAYOUT,
  KID_DECODE,
  KID_DIST_MFMA,
  KID_RERANK,
  KID_NORMS,
  KID_MEMBERS,
  KID_RERANK_SELECT,
  KID_RERANK_PAIRS,
  KID_DIST_MFMA_BF16,
  KID_LVQ_BATCH_APPLY,
  KID_COUNT
};
static const char *kKernelNames[KID_COUNT] = {
    "k_scan_exact", "k_som_update_run", "k_som_online_step", "k_lvq_online_step",
    "k_pack_samples", "k_merge_topk", "k_scan_masked", "k_layout", "k_decode_winners",
    "k_dist_mfma", "k_rerank", "k_norms_tau", "k_som_members",
    "k_rerank_select", "k_rerank_pairs", "k_dist_mfma_bf16", "k_lvq_batch_apply"};
extern "C" int somhip_kernel_count(void) { return KID_COUNT; }
extern "C" const char *somhip_kernel_name(int i) { return (i >= 0 && i < KID_COUNT) ? kKernelNames[i] : ""; }

constexpr int64_t ONLINE_CHUNK = 1024;   // iterations per captured graph

struct OnlineGraphKey {
  const void *tiles; int64_t n; int d; int patch_w; int64_t row_offset; int xdim; int topol;
  const void *rows; const void *mask; const void *slot; const void *sc; const void *rowidx; bool G, M;
  bool operator==(const OnlineGraphKey &o) const {
    return tiles == o.tiles && n == o.n && d == o.d && patch_w == o.patch_w && row_offset == o.row_offset &&
           xdim == o.xdim && topol == o.topol && rows == o.rows && mask == o.mask && slot == o.slot && sc == o.sc &&
           rowidx == o.rowidx && G == o.G && M == o.M;
  }
};

struct somhip_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  bool timing = false;
  uint64_t timing_mask = ~0ull;              // which kernel ids get events when timing is on
  int scan_mode = SOMHIP_SCAN_MFMA_BF16;
  double tau_scale = 1.0;                    // >= 1: widen the pre-filter window (experiments only)
  unsigned long long *d_stats = nullptr;     // [4] re-rank statistics (device)
  uint64_t samples_searched = 0;
  uint64_t lvq_batches = 0, lvq_samples = 0;   // exact batched LVQ: rescans and samples
  uint64_t lvq_stop_list = 0, lvq_stop_cache = 0, lvq_cycles[4] = {0, 0, 0, 0};   // batches ended by an exhausted candidate list / a full cache
  // ring of pinned host staging buffers for per-batch scalars (H2D without a host sync)
  void *pin_buf[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t pin_bytes[4] = {0, 0, 0, 0};
  hipEvent_t pin_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  int pin_next = 0;
  hipGraphExec_t online_graph_exec = nullptr;   // one chunk of k_som_online_step launches
  OnlineGraphKey online_graph_key{};
  struct Pending { int kid; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
  int64_t launches[KID_COUNT] = {0};
  double total_ms[KID_COUNT] = {0};
  // reusable device scratch
  void *scratch[20] = {nullptr};
  size_t scratch_bytes[20] = {0};
};

static int engine_scratch(somhip_engine *e, int slot, size_t bytes, void **out) {
  if (e->scratch_bytes[slot] < bytes) {
    if (e->scratch[slot]) HIPCHK(hipFree(e->scratch[slot]));
    e->scratch[slot] = nullptr;
    e->scratch_bytes[slot] = 0;
    size_t want = std::max(bytes, (size_t)4096);
    HIPCHK(hipMalloc(&e->scratch[slot], want));
    e->scratch_bytes[slot] = want;
  }
  *out = e->scratch[slot];
  return 0;
}

// next pinned staging buffer of the ring (waits only if its previous copy is still in flight)
static int pin_acquire(somhip_engine *e, size_t bytes, void **host, int *slot) {
  int i = e->pin_next;
  e->pin_next = (i + 1) & 3;
  if (!e->pin_ev[i]) HIPCHK(hipEventCreateWithFlags(&e->pin_ev[i], hipEventDisableTiming));
  else HIPCHK(hipEventSynchronize(e->pin_ev[i]));
  if (e->pin_bytes[i] < bytes) {
    if (e->pin_buf[i]) HIPCHK(hipHostFree(e->pin_buf[i]));
    e->pin_buf[i] = nullptr; e->pin_bytes[i] = 0;
    size_t want = std::max(bytes, (size_t)65536);
    HIPCHK(hipHostMalloc(&e->pin_buf[i], want, hipHostMallocDefault));
    e->pin_bytes[i] = want;
  }
  *host = e->pin_buf[i];
  *slot = i;
  return 0;
}
static int pin_upload(somhip_engine *e, int slot, void *dev, size_t bytes) {
  HIPCHK(hipMemcpyAsync(dev, e->pin_buf[slot], bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipEventRecord(e->pin_ev[slot], e->stream));
  return 0;
}

static int timing_flush(somhip_engine *e) {
  if (e->pending.empty()) return 0;
  HIPCHK(hipStreamSynchronize(e->stream));
  for (auto &p : e->pending) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, p.a, p.b));
    e->launches[p.kid]++;
    e->total_ms[p.kid] += ms;
    e->pool.push_back(p.a);
    e->pool.push_back(p.b);
  }
  e->pending.clear();
  return 0;
}

struct LaunchTimer {   // HIP events on the engine's own stream around one launch
  somhip_engine *e;
  int kid;
  hipEvent_t a = nullptr, b = nullptr;
  bool on = false;
  LaunchTimer(somhip_engine *e_, int kid_) : e(e_), kid(kid_) {
    if (!e->timing || !((e->timing_mask >> kid_) & 1ull)) return;
    auto get = [&](hipEvent_t *ev) {
      if (!e->pool.empty()) { *ev = e->pool.back(); e->pool.pop_back(); return true; }
      return hipEventCreate(ev) == hipSuccess;
    };
    if (get(&a) && get(&b)) { on = true; (void)hipEventRecord(a, e->stream); }
  }
  ~LaunchTimer() {
    if (!on) return;
    (void)hipEventRecord(b, e->stream);
    e->pending.push_back({kid, a, b});
    if (e->pending.size() >= 2048) (void)timing_flush(e);
  }
};

// ---------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------
extern "C" int somhip_engine_create(int device, somhip_engine **out) {
  if (!out) return fail("somhip_engine_create: null out");
  int ndev = 0;
  hipError_t er = hipGetDeviceCount(&ndev);
  if (er != hipSuccess || ndev <= 0)
    return fail("somhip_engine_create: no HIP device available (%s) -- this engine has no CPU path",
                er == hipSuccess ? "device count 0" : hipGetErrorString(er));
  if (device < 0 || device >= ndev) return fail("somhip_engine_create: device %d out of range (%d)", device, ndev);
  HIPCHK(hipSetDevice(device));
  somhip_engine *e = new somhip_engine();
  e->device = device;
  HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  if (const char *ts = getenv("SOMHIP_TAU_SCALE")) { double v = atof(ts); if (v >= 1.0) e->tau_scale = v; }
  HIPCHK(hipMalloc((void **)&e->d_stats, (8 + 128) * sizeof(unsigned long long)));   // + 64 {rows, pairs} update counters
  HIPCHK(hipMemset(e->d_stats, 0, (8 + 128) * sizeof(unsigned long long)));
  *out = e;
  return 0;
}
extern "C" void somhip_engine_destroy(somhip_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  for (auto &p : e->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto ev : e->pool) (void)hipEventDestroy(ev);
  for (int i = 0; i < 20; i++) if (e->scratch[i]) (void)hipFree(e->scratch[i]);
  if (e->d_stats) (void)hipFree(e->d_stats);
  if (e->online_graph_exec) (void)hipGraphExecDestroy(e->online_graph_exec);
  for (int i = 0; i < 4; i++) {
    if (e->pin_buf[i]) (void)hipHostFree(e->pin_buf[i]);
    if (e->pin_ev[i]) (void)hipEventDestroy(e->pin_ev[i]);
  }
  (void)hipStreamDestroy(e->stream);
  delete e;
}
extern "C" void *somhip_engine_stream(somhip_engine *e) { return e ? (void *)e->stream : nullptr; }
extern "C" int somhip_engine_sync(somhip_engine *e) {
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}
extern "C" int somhip_engine_set_scan_mode(somhip_engine *e, int mode) {
  if (mode != SOMHIP_SCAN_DIRECT && mode != SOMHIP_SCAN_MFMA && mode != SOMHIP_SCAN_MFMA_BF16) return fail("somhip_engine_set_scan_mode: bad mode %d", mode);
  e->scan_mode = mode;
  return 0;
}
extern "C" int somhip_scan_stats(somhip_engine *e, uint64_t out[6]) {
  unsigned long long h[8 + 128];
  HIPCHK(hipMemcpyAsync(h, e->d_stats, sizeof h, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int k = 0; k < 64; k++) { h[3] += h[8 + 2 * k]; h[4] += h[8 + 2 * k + 1]; }
  out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[3] = e->samples_searched; out[4] = h[3]; out[5] = h[4];
  return 0;
}
extern "C" int somhip_lvq_stats(somhip_engine *e, uint64_t out[8]) {
  if (!e || !out) return fail("somhip_lvq_stats: null argument");
  out[0] = e->lvq_batches; out[1] = e->lvq_samples; out[2] = e->lvq_stop_list; out[3] = e->lvq_stop_cache;
  for (int k = 0; k < 4; k++) out[4 + k] = e->lvq_cycles[k];
  return 0;
}
extern "C" int somhip_timing_enable(somhip_engine *e, int on) { CHK(timing_flush(e)); e->timing = on != 0; return 0; }
extern "C" int somhip_timing_select(somhip_engine *e, uint64_t kernel_mask) { e->timing_mask = kernel_mask; return 0; }
extern "C" int somhip_timing_reset(somhip_engine *e) {
  CHK(timing_flush(e));
  for (int i = 0; i < KID_COUNT; i++) { e->launches[i] = 0; e->total_ms[i] = 0; }
  return 0;
}
extern "C" int somhip_timing_get(somhip_engine *e, int k, int64_t *launches, double *total_ms) {
  if (k < 0 || k >= KID_COUNT) return fail("somhip_timing_get: bad kernel id %d", k);
  CHK(timing_flush(e));
  if (launches) *launches = e->launches[k];
  if (total_ms) *total_ms = e->total_ms[k];
  return 0;
}
extern "C" int somhip_device_alloc(somhip_engine *e, int64_t bytes, void **p) {
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMalloc(p, (size_t)std::max<int64_t>(bytes, 16)));
  return 0;
}
extern "C" int somhip_device_free(somhip_engine *e, void *p) {
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipFree(p));
  return 0;
}
extern "C" int somhip_copy_to_host(somhip_engine *e, void *dst, const void *src, int64_t bytes) {
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}
extern "C" int somhip_copy_to_device(somhip_engine *e, void *dst, const void *src, int64_t bytes) {
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

// ---------------------------------------------------------------------------------
// codebook / dataset mirrors
// ---------------------------------------------------------------------------------
struct somhip_codebook {
  somhip_engine *e = nullptr;
  CbView v{};
  int ydim = 0;
  int64_t n_global = 0;
  int32_t *d_labels = nullptr;     // [n] local rows
  float *d_talpha = nullptr;       // [n] OLVQ1 rates
  float *d_cn = nullptr;           // [ngroups*64] squared row norms (MFMA pre-filter)
  unsigned int *d_cnmax = nullptr; // bits of max squared norm
  uint4 *d_chi = nullptr, *d_clo = nullptr;   // bf16 hi/lo tiles [ngroups][d8][64] (bf16 pre-filter)
  bool prep_current = false;       // one-shot: the caller has just brought d_cn/d_chi/d_clo up to date itself
};
struct somhip_dataset {
  somhip_engine *e = nullptr;
  const float *d_rows = nullptr;
  bool owns_rows = false;
  int64_t n = 0;
  int d = 0;
  uint8_t *d_mask = nullptr;
  std::vector<uint8_t> all_masked;   // host: 1 if every component of the row is masked
  std::vector<int32_t> labels;       // host copies of the per-row scalars
  std::vector<int16_t> weight;
  std::vector<int16_t> fixed_xy;
};

static int upload_rows(somhip_codebook *cb, const float *rows) {
  somhip_engine *e = cb->e;
  cb->prep_current = false;
  void *stage;
  size_t bytes = sizeof(float) * (size_t)cb->v.n * cb->v.d;
  CHK(engine_scratch(e, 0, bytes, &stage));
  HIPCHK(hipMemcpyAsync(stage, rows, bytes, hipMemcpyHostToDevice, e->stream));
  {
    LaunchTimer t(e, KID_LAYOUT);
    hipLaunchKernelGGL(k_rows_to_tiles, dim3((unsigned)cb->v.ngroups), dim3(256), 0, e->stream,
                       (const float *)stage, cb->v);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int somhip_codebook_create(somhip_engine *e, const float *rows, const int32_t *labels,
                                      int64_t n_rows, int dim, int topol, int neigh, int xdim,
                                      int ydim, int64_t row_offset, int64_t n_global,
                                      somhip_codebook **out) {
  if (!e || !rows || !out) return fail("somhip_codebook_create: null argument");
  if (n_rows <= 0 || dim <= 0) return fail("somhip_codebook_create: empty codebook (%lld x %d)", (long long)n_rows, dim);
  if (n_global < row_offset + n_rows) return fail("somhip_codebook_create: shard [%lld,%lld) outside %lld rows",
                                                  (long long)row_offset, (long long)(row_offset + n_rows), (long long)n_global);
  if (n_global >= 0xFFFFFFFFll) return fail("somhip_codebook_create: more than 2^32-2 rows");
  if (topol >= SOMHIP_TOPOL_HEXA && (xdim <= 0 || ydim <= 0 || (int64_t)xdim * ydim != n_global))
    return fail("somhip_codebook_create: map %dx%d does not have %lld units", xdim, ydim, (long long)n_global);
  HIPCHK(hipSetDevice(e->device));
  somhip_codebook *cb = new somhip_codebook();
  cb->e = e;
  cb->v.n = n_rows;
  cb->v.ngroups = (n_rows + WAVE - 1) / WAVE;
  cb->v.d = dim;
  cb->v.d4 = (dim + 3) / 4;
  cb->v.row_offset = row_offset;
  cb->v.xdim = xdim > 0 ? xdim : 1;
  cb->v.topol = topol;
  cb->v.neigh = neigh;
  cb->v.patch_w = 0;
  if (topol >= SOMHIP_TOPOL_HEXA && xdim % 8 == 0 && ydim % 8 == 0 && row_offset % (8 * (int64_t)xdim) == 0 &&
      n_rows % (8 * (int64_t)xdim) == 0 && !getenv("SOMHIP_LINEAR_ROWS"))
    cb->v.patch_w = xdim / 8;                     // 8x8-unit row groups (kernels.hpp CbView)
  cb->ydim = ydim;
  cb->n_global = n_global;
  size_t tile_bytes = (size_t)cb->v.ngroups * cb->v.d4 * WAVE * 4 * sizeof(float);
  HIPCHK(hipMalloc((void **)&cb->v.tiles, tile_bytes));
  int rc = upload_rows(cb, rows);
  if (rc) { somhip_codebook_destroy(cb); return rc; }
  if (labels) {
    HIPCHK(hipMalloc((void **)&cb->d_labels, sizeof(int32_t) * (size_t)n_rows));
    HIPCHK(hipMemcpy(cb->d_labels, labels, sizeof(int32_t) * (size_t)n_rows, hipMemcpyHostToDevice));
  }
  *out = cb;
  return 0;
}
extern "C" int somhip_codebook_upload(somhip_codebook *cb, const float *rows) {
  HIPCHK(hipSetDevice(cb->e->device));
  return upload_rows(cb, rows);
}
extern "C" int somhip_codebook_download(somhip_codebook *cb, float *rows) {
  somhip_engine *e = cb->e;
  HIPCHK(hipSetDevice(e->device));
  void *stage;
  size_t bytes = sizeof(float) * (size_t)cb->v.n * cb->v.d;
  CHK(engine_scratch(e, 0, bytes, &stage));
  {
    LaunchTimer t(e, KID_LAYOUT);
    hipLaunchKernelGGL(k_tiles_to_rows, dim3((unsigned)cb->v.ngroups), dim3(256), 0, e->stream,
                       (float *)stage, cb->v);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(rows, stage, bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}
extern "C" void somhip_codebook_destroy(somhip_codebook *cb) {
  if (!cb) return;
  (void)hipSetDevice(cb->e->device);
  (void)hipStreamSynchronize(cb->e->stream);
  if (cb->v.tiles) (void)hipFree(cb->v.tiles);
  if (cb->d_labels) (void)hipFree(cb->d_labels);
  if (cb->d_talpha) (void)hipFree(cb->d_talpha);
  if (cb->d_cn) (void)hipFree(cb->d_cn);
  if (cb->d_cnmax) (void)hipFree(cb->d_cnmax);
  if (cb->d_chi) (void)hipFree(cb->d_chi);
  if (cb->d_clo) (void)hipFree(cb->d_clo);
  delete cb;
}

extern "C" int somhip_dataset_create(somhip_engine *e, const float *rows, int64_t n_rows, int dim,
                                     const uint8_t *mask, const int32_t *labels,
                                     const int16_t *weight, const int16_t *fixed_xy,
                                     somhip_dataset **out) {
  if (!e || !rows || !out) return fail("somhip_dataset_create: null argument");
  if (n_rows <= 0 || dim <= 0) return fail("somhip_dataset_create: empty data set");
  HIPCHK(hipSetDevice(e->device));
  somhip_dataset *ds = new somhip_dataset();
  ds->e = e; ds->n = n_rows; ds->d = dim; ds->owns_rows = true;
  size_t bytes = sizeof(float) * (size_t)n_rows * dim;
  float *dr = nullptr;
  // 16 spare bytes: wave-uniform float4 reads of the last row never leave the buffer
  HIPCHK(hipMalloc((void **)&dr, bytes + 16));
  HIPCHK(hipMemcpy(dr, rows, bytes, hipMemcpyHostToDevice));
  ds->d_rows = dr;
  if (mask) {
    bool any = false;
    ds->all_masked.assign((size_t)n_rows, 0);
    for (int64_t r = 0; r < n_rows; r++) {
      int cnt = 0;
      for (int i = 0; i < dim; i++) cnt += mask[r * dim + i] != 0;
      any |= cnt > 0;
      ds->all_masked[(size_t)r] = cnt == dim;
    }
    if (any) {
      HIPCHK(hipMalloc((void **)&ds->d_mask, (size_t)n_rows * dim));
      HIPCHK(hipMemcpy(ds->d_mask, mask, (size_t)n_rows * dim, hipMemcpyHostToDevice));
    } else {
      ds->all_masked.clear();
    }
  }
  if (labels) ds->labels.assign(labels, labels + n_rows);
  if (weight) ds->weight.assign(weight, weight + n_rows);
  if (fixed_xy) ds->fixed_xy.assign(fixed_xy, fixed_xy + 2 * n_rows);
  *out = ds;
  return 0;
}
extern "C" int somhip_dataset_wrap_device(somhip_engine *e, const float *dev_rows, int64_t n_rows,
                                          int dim, somhip_dataset **out) {
  if (!e || !dev_rows || !out) return fail("somhip_dataset_wrap_device: null argument");
  somhip_dataset *ds = new somhip_dataset();
  ds->e = e; ds->n = n_rows; ds->d = dim; ds->d_rows = dev_rows; ds->owns_rows = false;
  *out = ds;
  return 0;
}
extern "C" void somhip_dataset_destroy(somhip_dataset *ds) {
  if (!ds) return;
  (void)hipSetDevice(ds->e->device);
  (void)hipStreamSynchronize(ds->e->stream);
  if (ds->owns_rows && ds->d_rows) (void)hipFree((void *)ds->d_rows);
  if (ds->d_mask) (void)hipFree(ds->d_mask);
  delete ds;
}

// ---------------------------------------------------------------------------------
// winner scans
// ---------------------------------------------------------------------------------
constexpr int SCAN_S = 32;     // samples per workgroup tile

static int check_pair(const somhip_codebook *cb, const somhip_dataset *ds, const char *who) {
  if (!cb || !ds) return fail("%s: null handle", who);
  if (cb->e != ds->e) return fail("%s: codebook and data belong to different engines", who);
  if (cb->v.d != ds->d)
    return fail("%s: code dimension (%d) != data dimension (%d)", who, cb->v.d, ds->d);   // som_rout.c:591-596
  return 0;
}

constexpr int64_t MFMA_MIN_SAMPLES = 32;

// Bound on |s~ + ||x||^2 - d| / (||x|| + ||c||)^2: fp32 MFMA GEMM form vs the
// reference's direct form, u = 2^-24, gamma_k = k u / (1 - k u):  2 * gamma_{d+2}.
// Split-bf16 form (kernels.hpp K2b): the dot product loses at most 3.1 * 2^-16 ||x|| ||c|| to the
// dropped lo*lo / residual terms and accumulates 3d exact products in fp32 -- bounded here with
// a factor 2 on the accumulation (no assumption on the matrix pipe's internal summation order or
// rounding mode beyond "error of a sum of k terms <= 2 gamma_k * sum |terms|"), sum |terms| <=
// 1.02 ||x|| ||c||.  The cn term and the direct-form term are as in the fp32 case.
static double prefilter_err_coeff(const somhip_engine *e, int d) {
  const double u = 5.9604644775390625e-08;
  const double k = (d + 2) * u;
  const double gam = k / (1.0 - k);
  if (e->scan_mode == SOMHIP_SCAN_MFMA_BF16) {
    const double k3 = 3.0 * (d + 2) * u;
    const double dot = 2.0 * (k3 / (1.0 - k3)) * 1.02 + 3.1 / 65536.0 + u;
    return (std::max(dot, gam) + gam) * e->tau_scale;
  }
  return 2.0 * gam * e->tau_scale;
}

// MFMA pre-filter + exact re-rank (kernels.hpp K2/K2b/K2s/K2p/K2r)
static int scan_keys_mfma(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count,
                          int64_t nsb, uint64_t *d_keys, bool prefilter_only = false,
                          float **out_wmin = nullptr, float **out_tau = nullptr) {
  somhip_engine *e = cb->e;
  const int64_t bpad = nsb * SCAN_S;
  const bool bf16 = e->scan_mode == SOMHIP_SCAN_MFMA_BF16;
  const int d8 = (cb->v.d4 + 1) / 2;
  if (!cb->d_cn) {
    HIPCHK(hipMalloc((void **)&cb->d_cn, sizeof(float) * (size_t)cb->v.ngroups * WAVE));
    HIPCHK(hipMalloc((void **)&cb->d_cnmax, sizeof(unsigned int)));
  }
  if (bf16 && !cb->d_chi) {
    HIPCHK(hipMalloc((void **)&cb->d_chi, sizeof(uint4) * (size_t)cb->v.ngroups * d8 * WAVE));
    HIPCHK(hipMalloc((void **)&cb->d_clo, sizeof(uint4) * (size_t)cb->v.ngroups * d8 * WAVE));
  }
  void *dtau, *dwmin, *dwmask, *xt;
  CHK(engine_scratch(e, 5, sizeof(float) * (size_t)bpad, &dtau));
  CHK(engine_scratch(e, 6, sizeof(float) * (size_t)cb->v.ngroups * bpad, &dwmin));
  CHK(engine_scratch(e, 7, sizeof(uint64_t) * (size_t)cb->v.ngroups * bpad, &dwmask));
  // sample tiles: fp32 xt[sb][q][32][4], or bf16 hi | lo xt[sb][kb][32][8]
  const size_t xt_bytes = bf16 ? 2 * sizeof(uint4) * (size_t)nsb * d8 * 32 : sizeof(float4) * (size_t)nsb * cb->v.d4 * SCAN_S;
  CHK(engine_scratch(e, 1, xt_bytes, &xt));
  uint4 *xhi = (uint4 *)xt, *xlo = xhi + (size_t)nsb * d8 * 32;
  {
    LaunchTimer t(e, KID_PACK_SAMPLES);
    if (bf16)
      hipLaunchKernelGGL(k_pack_samples_bf16, dim3((unsigned)nsb), dim3(256), 0, e->stream, ds->d_rows, ds->n, ds->d,
                         d8, first, count, xhi, xlo, cb->d_cnmax);
    else
      hipLaunchKernelGGL(k_pack_samples<SCAN_S>, dim3((unsigned)nsb), dim3(256), 0, e->stream, ds->d_rows, ds->n,
                         ds->d, cb->v.d4, first, count, (float4 *)xt);
  }
  HIPCHK(hipGetLastError());
  const bool prep_was_current = cb->prep_current;
  cb->prep_current = false;                              // one-shot
  if (!bf16) HIPCHK(hipMemsetAsync(cb->d_cnmax, 0, sizeof(unsigned int), e->stream));   // (the bf16 pack kernel zeroes it)
  // scratch of the re-rank, preset by k_sample_tau
  const uint32_t ncols = (uint32_t)(bpad / 32);
  void *dg;
  CHK(engine_scratch(e, 12, sizeof(uint32_t) * (2 * (size_t)bpad + 4 * (size_t)ncols), &dg));
  uint32_t *dgmin = (uint32_t *)dg, *dgcount = dgmin + bpad, *dcolcount = dgcount + bpad;
  uint32_t *d_paircount = reinterpret_cast<uint32_t *>(e->d_stats + 6);   // stays 0 unless a segment overflows
  RerankInit rinit = {prefilter_only ? nullptr : d_keys, dgmin, d_paircount, bpad, (int)ncols};
  {
    LaunchTimer t(e, KID_NORMS);
    if (bf16 && prep_was_current) {
      // the LVQ engine re-split exactly the rows its last batch corrected: only the maximum is due
      hipLaunchKernelGGL(k_max_norm, dim3(64), dim3(256), 0, e->stream, cb->v, (const float *)cb->d_cn, cb->d_cnmax);
    } else if (bf16)
      hipLaunchKernelGGL(k_prep_codes_bf16, dim3((unsigned)((cb->v.ngroups + 3) / 4)), dim3(256), 0, e->stream,
                         cb->v, d8, cb->d_cn, cb->d_cnmax, cb->d_chi, cb->d_clo);
    else
      hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((cb->v.ngroups + 3) / 4)), dim3(256), 0, e->stream,
                         cb->v, cb->d_cn, cb->d_cnmax);
    hipLaunchKernelGGL(k_sample_tau, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, e->stream,
                       ds->d_rows, ds->n, ds->d, first, count, (const unsigned int *)cb->d_cnmax,
                       prefilter_err_coeff(e, ds->d), (float *)dtau, rinit);
  }
  HIPCHK(hipGetLastError());
  {
    LaunchTimer t(e, bf16 ? KID_DIST_MFMA_BF16 : KID_DIST_MFMA);
    dim3 grid((unsigned)((nsb + 3) / 4), (unsigned)((cb->v.ngroups + 1) / 2));
#ifndef SOMHIP_DMA_KB
#define SOMHIP_DMA_KB 4
#endif
#ifndef SOMHIP_DMA_MINB
#define SOMHIP_DMA_MINB 2
#endif
    if (bf16 && (d8 % 2) == 0 && nsb >= 8 && !getenv("SOMHIP_NO_LDS_DMA") && !getenv("SOMHIP_DIST_NARROW")) {
      dim3 gridw((unsigned)((nsb + 7) / 8), (unsigned)((cb->v.ngroups + 1) / 2));
      hipLaunchKernelGGL((k_dist_mfma_bf16_wide<2>), gridw, dim3(256), 0, e->stream, cb->v, d8,
                         (const uint4 *)cb->d_chi, (const uint4 *)cb->d_clo, (const uint4 *)xhi, (const uint4 *)xlo,
                         (const float *)cb->d_cn, (const float *)dtau, count, bpad, (float *)dwmin, (uint64_t *)dwmask);
    } else if (bf16 && (d8 % SOMHIP_DMA_KB) == 0 && !getenv("SOMHIP_NO_LDS_DMA"))
      hipLaunchKernelGGL((k_dist_mfma_bf16_dma<SOMHIP_DMA_KB, SOMHIP_DMA_MINB>), grid, dim3(256), 0, e->stream, cb->v, d8,
                         (const uint4 *)cb->d_chi, (const uint4 *)cb->d_clo, (const uint4 *)xhi, (const uint4 *)xlo,
                         (const float *)cb->d_cn, (const float *)dtau, count, bpad, (float *)dwmin, (uint64_t *)dwmask);
    else if (bf16)
      hipLaunchKernelGGL(k_dist_mfma_bf16, grid, dim3(256), 0, e->stream, cb->v, d8, (const uint4 *)cb->d_chi,
                         (const uint4 *)cb->d_clo, (const uint4 *)xhi, (const uint4 *)xlo, (const float *)cb->d_cn,
                         (const float *)dtau, count, bpad, (float *)dwmin, (uint64_t *)dwmask);
    else
      hipLaunchKernelGGL(k_dist_mfma, grid, dim3(256), 0, e->stream, cb->v, (const float4 *)xt, (const float *)cb->d_cn,
                         (const float *)dtau, count, bpad, (float *)dwmin, (uint64_t *)dwmask);
  }
  HIPCHK(hipGetLastError());
  if (prefilter_only) { *out_wmin = (float *)dwmin; *out_tau = (float *)dtau; return 0; }
  // exact re-rank: row-granular pair path for the usual few candidates, group-granular
  // k_rerank for flagged samples (too many candidates / list full)
  // pair list: one segment per 32-sample column
  if (ncols > (uint32_t)PAIR_MAX_COLS) return fail("winner search: more than %d samples in one run", PAIR_MAX_COLS * 32);
  const uint32_t cap_col = 16384;                    // 512 per sample on average; a full segment -> K2r
  const uint32_t cap = (uint32_t)std::min<int64_t>((int64_t)ncols * cap_col, 0x7FFFFFF0);
  void *dpairs;
  CHK(engine_scratch(e, 11, sizeof(uint2) * (size_t)ncols * cap_col + 16, &dpairs));
  {
    // global minimum of the group minima per sample, then the candidate pairs; both over a grid of
    // (32-sample columns) x (chunks of row groups)
    const int64_t nchunks = std::max<int64_t>(1, std::min<int64_t>(64, (cb->v.ngroups + 63) / 64));
    const int64_t chunk = ((cb->v.ngroups + nchunks - 1) / nchunks + 7) / 8 * 8;
    const dim3 sgrid((unsigned)(bpad / 32), (unsigned)((cb->v.ngroups + chunk - 1) / chunk));
    LaunchTimer t(e, KID_RERANK_SELECT);
    hipLaunchKernelGGL(k_group_min, sgrid, dim3(256), 0, e->stream, cb->v.ngroups, bpad, chunk, (const float *)dwmin, dgmin);
    hipLaunchKernelGGL(k_rerank_select, sgrid, dim3(256), 0, e->stream, cb->v, count, bpad, chunk,
                       (const float *)dwmin, (const uint64_t *)dwmask, (const float *)dtau, (const uint32_t *)dgmin,
                       dgcount, cap, cap_col, (uint2 *)dpairs, dcolcount, d_paircount, e->d_stats);
  }
  {
    LaunchTimer t(e, KID_RERANK_PAIRS);
    hipLaunchKernelGGL(k_rerank_pairs, dim3(512), dim3(256), 0, e->stream, cb->v, ds->d_rows,
                       ds->n, first, cap, cap_col, (int)ncols, (const uint2 *)dpairs, (const uint32_t *)dcolcount,
                       (const uint32_t *)d_paircount, d_keys, e->d_stats);
  }
  {
    LaunchTimer t(e, KID_RERANK);
    hipLaunchKernelGGL(k_rerank, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, e->stream, cb->v,
                       ds->d_rows, ds->n, first, count, bpad, (const float *)dwmin,
                       (const uint64_t *)dwmask, (const float *)dtau, (const uint32_t *)d_paircount, cap, d_keys,
                       e->d_stats);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// keys[count] <- exact nearest row per sample, FIRST tie rule, local shard
static int scan_keys_top1(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count,
                          uint64_t *d_keys) {
  somhip_engine *e = cb->e;
  const bool use_mfma = !ds->d_mask && e->scan_mode != SOMHIP_SCAN_DIRECT && count >= MFMA_MIN_SAMPLES && cb->v.n >= 64 &&
                        count <= (int64_t)PAIR_MAX_COLS * 32;      // longer runs: the direct scan below
  if (!use_mfma) HIPCHK(hipMemsetAsync(d_keys, 0xFF, sizeof(uint64_t) * (size_t)count, e->stream));   // (else k_sample_tau presets them)
  if (ds->d_mask) {
    for (int64_t off = 0; off < count; off += 32768) {      // grid.y limit
      int64_t c = std::min<int64_t>(32768, count - off);
      LaunchTimer t(e, KID_SCAN_MASKED);
      dim3 grid((unsigned)((cb->v.ngroups + 3) / 4), (unsigned)c);
      hipLaunchKernelGGL(k_scan_masked, grid, dim3(256), 0, e->stream, cb->v, ds->d_rows, ds->d_mask,
                         ds->n, (first + off) % ds->n, c, 0, d_keys + off);
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  int64_t nsb = (count + SCAN_S - 1) / SCAN_S;
  e->samples_searched += (uint64_t)count;
  if (use_mfma) return scan_keys_mfma(cb, ds, first, count, nsb, d_keys);
  void *xt;
  CHK(engine_scratch(e, 1, sizeof(float4) * (size_t)nsb * cb->v.d4 * SCAN_S, &xt));
  {
    LaunchTimer t(e, KID_PACK_SAMPLES);
    hipLaunchKernelGGL(k_pack_samples<SCAN_S>, dim3((unsigned)nsb), dim3(256), 0, e->stream,
                       ds->d_rows, ds->n, ds->d, cb->v.d4, first, count, (float4 *)xt);
  }
  HIPCHK(hipGetLastError());
  {
    LaunchTimer t(e, KID_SCAN_EXACT);
    dim3 grid((unsigned)nsb, (unsigned)((cb->v.ngroups + 3) / 4));
    hipLaunchKernelGGL((k_scan_exact<SCAN_S, 1, 1>), grid, dim3(256), 0, e->stream, cb->v,
                       (const float4 *)xt, count, 0, d_keys, (uint64_t *)nullptr);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

template <int K>
static int scan_keys_topk(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count,
                          uint64_t *d_keys /*[count][K]*/, int tie_knn = 1) {
  somhip_engine *e = cb->e;
  // big codebooks: bf16 pre-filter + exact re-rank of the surviving row groups (kernels.hpp K2k)
  const bool force_mfma = getenv("SOMHIP_TOPK_MFMA") != nullptr, no_mfma = getenv("SOMHIP_TOPK_DIRECT") != nullptr;
  if (!no_mfma && e->scan_mode == SOMHIP_SCAN_MFMA_BF16 && !ds->d_mask && count >= MFMA_MIN_SAMPLES &&
      cb->v.n >= (force_mfma ? 64 : 4096) && count <= (int64_t)PAIR_MAX_COLS * 32) {
    const int64_t nsb = (count + SCAN_S - 1) / SCAN_S;
    float *dw = nullptr, *dt = nullptr;
    CHK(scan_keys_mfma(cb, ds, first, count, nsb, nullptr, true, &dw, &dt));
    const int64_t bpad = nsb * SCAN_S;
    // pair-parallel re-rank (three launches); the one-wave-per-sample kernel only if the list overflows
    const uint32_t cap = (uint32_t)std::min<int64_t>(count * 128 + 4096, 0x3FFFFFF0);
    void *dpairs, *dspan, *dpart, *dcnt;
    CHK(engine_scratch(e, 11, sizeof(uint2) * (size_t)cap, &dpairs));
    CHK(engine_scratch(e, 2, sizeof(uint64_t) * (size_t)cap * K, &dpart));
    CHK(engine_scratch(e, 15, sizeof(TopkSpan) * (size_t)count + 16, &dspan));
    uint32_t *dcounter = reinterpret_cast<uint32_t *>(e->d_stats + 7);
    (void)dcnt;
    HIPCHK(hipMemsetAsync(dcounter, 0, 2 * sizeof(uint32_t), e->stream));
    LaunchTimer t(e, KID_RERANK);
    hipLaunchKernelGGL(k_topk_select<K>, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, e->stream, cb->v, count, bpad,
                       (const float *)dw, (const float *)dt, cap, (uint2 *)dpairs, (TopkSpan *)dspan, dcounter);
    hipLaunchKernelGGL(k_topk_pairs<K>, dim3(1024), dim3(256), 0, e->stream, cb->v, ds->d_rows, ds->n, first, tie_knn,
                       (const uint2 *)dpairs, (const uint32_t *)dcounter, (uint64_t *)dpart);
    hipLaunchKernelGGL(k_topk_merge<K>, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, e->stream, count,
                       (const TopkSpan *)dspan, (const uint64_t *)dpart, (const uint32_t *)dcounter, d_keys);
    HIPCHK(hipGetLastError());
    uint32_t hc[2];
    HIPCHK(hipMemcpyAsync(hc, dcounter, sizeof hc, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (hc[1]) {                                         // list full: every sample through the one-wave kernel
      hipLaunchKernelGGL(k_rerank_topk<K>, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, e->stream, cb->v, ds->d_rows,
                         ds->n, first, count, bpad, (const float *)dw, (const float *)dt, tie_knn, d_keys);
      HIPCHK(hipGetLastError());
    }
    return 0;
  }
  int64_t nsb = (count + SCAN_S - 1) / SCAN_S;
  int nblk = (int)((cb->v.ngroups + 3) / 4);
  void *xt, *part;
  CHK(engine_scratch(e, 1, sizeof(float4) * (size_t)nsb * cb->v.d4 * SCAN_S, &xt));
  CHK(engine_scratch(e, 2, sizeof(uint64_t) * (size_t)count * nblk * K, &part));
  {
    LaunchTimer t(e, KID_PACK_SAMPLES);
    hipLaunchKernelGGL(k_pack_samples<SCAN_S>, dim3((unsigned)nsb), dim3(256), 0, e->stream,
                       ds->d_rows, ds->n, ds->d, cb->v.d4, first, count, (float4 *)xt);
  }
  HIPCHK(hipGetLastError());
  {
    LaunchTimer t(e, KID_SCAN_EXACT);
    dim3 grid((unsigned)nsb, (unsigned)nblk);
    hipLaunchKernelGGL((k_scan_exact<SCAN_S, 1, K>), grid, dim3(256), 0, e->stream, cb->v,
                       (const float4 *)xt, count, tie_knn, (uint64_t *)nullptr, (uint64_t *)part);
  }
  HIPCHK(hipGetLastError());
  {
    LaunchTimer t(e, KID_MERGE_TOPK);
    hipLaunchKernelGGL(k_merge_topk<K>, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, e->stream,
                       (const uint64_t *)part, nblk, count, d_keys);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int somhip_debug_prefilter(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count,
                                      float *wmin, float *tau, int64_t *bpad) {
  CHK(check_pair(cb, ds, "somhip_debug_prefilter"));
  somhip_engine *e = cb->e;
  if (e->scan_mode == SOMHIP_SCAN_DIRECT || ds->d_mask) return fail("somhip_debug_prefilter: no pre-filter in this mode");
  HIPCHK(hipSetDevice(e->device));
  int64_t nsb = (count + SCAN_S - 1) / SCAN_S;
  float *dw = nullptr, *dt = nullptr;
  CHK(scan_keys_mfma(cb, ds, first, count, nsb, nullptr, true, &dw, &dt));
  HIPCHK(hipMemcpyAsync(wmin, dw, sizeof(float) * (size_t)cb->v.ngroups * nsb * SCAN_S, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(tau, dt, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (bpad) *bpad = nsb * SCAN_S;
  return 0;
}

extern "C" int somhip_batch_winner_keys(somhip_codebook *cb, somhip_dataset *ds, int64_t first,
                                        int64_t count, uint64_t *dev_keys) {
  CHK(check_pair(cb, ds, "somhip_batch_winner_keys"));
  if (count <= 0) return 0;
  HIPCHK(hipSetDevice(cb->e->device));
  CHK(scan_keys_top1(cb, ds, first, count, dev_keys));
  hipLaunchKernelGGL(k_clamp_keys, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, cb->e->stream, dev_keys, count);
  HIPCHK(hipGetLastError());
  return 0;
}

// X2: this shard's k best rows per sample as packed keys, ascending; a host all-gathers
// the shards' lists and keeps the k smallest per sample (keys are unique: tag = global row, or its
// complement for the k-NN tie order, so the merge IS find_winner_knn over the whole codebook).
extern "C" int somhip_batch_topk_keys(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count,
                                      int knn, int tie, uint64_t *dev_keys) {
  CHK(check_pair(cb, ds, "somhip_batch_topk_keys"));
  if (knn < 1 || knn > 8) return fail("somhip_batch_topk_keys: knn %d not in 1..8", knn);
  if (ds->d_mask) return fail("somhip_batch_topk_keys: masked samples are not supported");
  if (count <= 0) return 0;
  HIPCHK(hipSetDevice(cb->e->device));
  const int t = tie == SOMHIP_TIE_KNN ? 1 : 0;
  if (knn == 1) {
    if (t) return fail("somhip_batch_topk_keys: knn 1 is find_winner_euc (SOMHIP_TIE_FIRST)");
    return somhip_batch_winner_keys(cb, ds, first, count, dev_keys);
  }
  if (knn == 2) return scan_keys_topk<2>(cb, ds, first, count, dev_keys, t);
  if (knn <= 4) {
    if (knn != 4) return fail("somhip_batch_topk_keys: knn must be 1, 2, 4 or 8");
    return scan_keys_topk<4>(cb, ds, first, count, dev_keys, t);
  }
  if (knn != 8) return fail("somhip_batch_topk_keys: knn must be 1, 2, 4 or 8");
  return scan_keys_topk<8>(cb, ds, first, count, dev_keys, t);
}

static void decode_key(uint64_t k, bool inverted, int32_t *index, float *diff) {
  uint32_t bits = (uint32_t)(k >> 32);
  uint32_t tag = (uint32_t)k;
  if (bits >= FLT_MAX_BITS) { *index = -1; *diff = -1.0f; return; }   // nothing beat FLT_MAX (lvq_pak.c:56)
  *index = (int32_t)(inverted ? ~tag : tag);
  memcpy(diff, &bits, 4);
}

extern "C" int somhip_find_winners(somhip_codebook *cb, somhip_dataset *ds, int64_t first,
                                   int64_t count, int knn, int tie, int32_t *index, float *diff,
                                   int32_t *ret) {
  CHK(check_pair(cb, ds, "somhip_find_winners"));
  if (knn < 1 || knn > 8) return fail("somhip_find_winners: knn %d not in 1..8", knn);
  if (!index || !diff) return fail("somhip_find_winners: null output");
  if (count <= 0) return 0;
  somhip_engine *e = cb->e;
  HIPCHK(hipSetDevice(e->device));
  // find_winner_knn(knn == 1) IS find_winner_euc (lvq_pak.c:160-161)
  const bool knn_rule = (tie == SOMHIP_TIE_KNN) && knn >= 2;
  if (!knn_rule && knn != 1) return fail("somhip_find_winners: knn > 1 needs SOMHIP_TIE_KNN");
  if (ds->d_mask && knn_rule) return fail("somhip_find_winners: k-NN with masked samples is not implemented");
  const int64_t CH = 4096;
  const int KK = knn == 1 ? 1 : knn == 2 ? 2 : knn <= 4 ? 4 : 8;
  void *dk;
  CHK(engine_scratch(e, 3, sizeof(uint64_t) * (size_t)std::min(CH, count) * KK, &dk));
  std::vector<uint64_t> hk((size_t)std::min(CH, count) * KK);
  for (int64_t off = 0; off < count; off += CH) {
    int64_t c = std::min(CH, count - off);
    int64_t f = (first + off) % ds->n;
    if (!knn_rule) CHK(scan_keys_top1(cb, ds, f, c, (uint64_t *)dk));
    else if (KK == 2) CHK(scan_keys_topk<2>(cb, ds, f, c, (uint64_t *)dk));
    else if (KK == 4) CHK(scan_keys_topk<4>(cb, ds, f, c, (uint64_t *)dk));
    else CHK(scan_keys_topk<8>(cb, ds, f, c, (uint64_t *)dk));
    HIPCHK(hipMemcpyAsync(hk.data(), dk, sizeof(uint64_t) * (size_t)c * KK, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int64_t i = 0; i < c; i++) {
      int64_t r = (f + i) % ds->n;
      bool empty = !ds->all_masked.empty() && ds->all_masked[(size_t)r];
      for (int k = 0; k < knn; k++) {
        int32_t *pi = index + (off + i) * knn + k;
        float *pd = diff + (off + i) * knn + k;
        if (empty) { *pi = -2; *pd = -1.0f; }
        else decode_key(hk[(size_t)i * KK + k], knn_rule, pi, pd);
      }
      if (ret) ret[off + i] = empty ? 0 : knn;
    }
  }
  return 0;
}

// lininit's data passes (find_eigenvectors, som_rout.c:211-289): per-component sums / counts over the
// unmasked entries, then the upper triangle (j >= i) of sum_r (x_ri - mean_i)(x_rj - mean_j); every
// element accumulated over the rows in file order, in fp32, like the reference.
extern "C" int somhip_column_sums(somhip_dataset *ds, float *sum, int64_t *count) {
  if (!ds || !sum || !count) return fail("somhip_column_sums: null argument");
  somhip_engine *e = ds->e;
  HIPCHK(hipSetDevice(e->device));
  void *dsum, *dcnt;
  CHK(engine_scratch(e, 3, sizeof(float) * (size_t)ds->d, &dsum));
  CHK(engine_scratch(e, 4, sizeof(unsigned long long) * (size_t)ds->d, &dcnt));
  hipLaunchKernelGGL(k_column_sums, dim3((unsigned)((ds->d + 255) / 256)), dim3(256), 0, e->stream, ds->d_rows,
                     (const uint8_t *)ds->d_mask, ds->n, ds->d, (float *)dsum, (unsigned long long *)dcnt);
  HIPCHK(hipGetLastError());
  std::vector<unsigned long long> hc((size_t)ds->d);
  HIPCHK(hipMemcpyAsync(sum, dsum, sizeof(float) * (size_t)ds->d, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(hc.data(), dcnt, sizeof(unsigned long long) * (size_t)ds->d, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int i = 0; i < ds->d; i++) count[i] = (int64_t)hc[(size_t)i];
  return 0;
}

extern "C" int somhip_centered_products(somhip_dataset *ds, const float *mean, float *r) {
  if (!ds || !mean || !r) return fail("somhip_centered_products: null argument");
  somhip_engine *e = ds->e;
  HIPCHK(hipSetDevice(e->device));
  const size_t dd = (size_t)ds->d * ds->d;
  void *dmean, *dr;
  CHK(engine_scratch(e, 3, sizeof(float) * (size_t)ds->d, &dmean));
  CHK(engine_scratch(e, 4, sizeof(float) * dd, &dr));
  HIPCHK(hipMemcpyAsync(dmean, mean, sizeof(float) * (size_t)ds->d, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemsetAsync(dr, 0, sizeof(float) * dd, e->stream));
  const unsigned nb = (unsigned)((ds->d + 15) / 16);
  hipLaunchKernelGGL(k_centered_products, dim3(nb, nb), dim3(256), 0, e->stream, ds->d_rows,
                     (const uint8_t *)ds->d_mask, ds->n, ds->d, (const float *)dmean, (float *)dr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(r, dr, sizeof(float) * dd, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

// find_qerror2 (som_rout.c:823-885): out[i] = the neighbourhood-weighted error of sample first+i
// (0 where the sample has no winner); the caller adds them in data order, as the reference does.
extern "C" int somhip_qerror2(somhip_codebook *cb, somhip_dataset *ds, float radius, int64_t first,
                              int64_t count, float *out, int32_t *ret) {
  CHK(check_pair(cb, ds, "somhip_qerror2"));
  if (!out) return fail("somhip_qerror2: null output");
  if (cb->v.topol != SOMHIP_TOPOL_HEXA && cb->v.topol != SOMHIP_TOPOL_RECT) return fail("somhip_qerror2: can't set SOM parameters");
  if (cb->v.row_offset != 0 || cb->n_global != cb->v.n) return fail("somhip_qerror2: sharded codebook not supported");
  if (count <= 0) return 0;
  somhip_engine *e = cb->e;
  HIPCHK(hipSetDevice(e->device));
  const bool gauss = cb->v.neigh == SOMHIP_NEIGH_GAUSSIAN;
  const float thresh = gauss ? 0.0f : bubble_threshold(radius);
  double reach = radius > 0.0f ? (double)radius / (cb->v.topol == SOMHIP_TOPOL_RECT ? 1.0 : 0.8660254037844386) + 1.0 : 1.0;
  const int ireach = reach > 1e6 ? 1000000 : (int)reach;
  const int64_t CH = 4096;
  void *dk, *dq;
  CHK(engine_scratch(e, 3, sizeof(uint64_t) * (size_t)std::min(CH, count), &dk));
  CHK(engine_scratch(e, 4, sizeof(float) * (size_t)std::min(CH, count), &dq));
  const size_t dyn = (size_t)cb->v.d * 5 + 16;
  for (int64_t off = 0; off < count; off += CH) {
    const int64_t c = std::min(CH, count - off);
    const int64_t f = (first + off) % ds->n;
    CHK(scan_keys_top1(cb, ds, f, c, (uint64_t *)dk));
    if (gauss)
      hipLaunchKernelGGL(k_qerror2<true>, dim3((unsigned)c), dim3(256), dyn, e->stream, cb->v, cb->ydim, ds->d_rows,
                         ds->d_mask, ds->n, f, (const uint64_t *)dk, radius, thresh, ireach, (float *)dq);
    else
      hipLaunchKernelGGL(k_qerror2<false>, dim3((unsigned)c), dim3(256), dyn, e->stream, cb->v, cb->ydim, ds->d_rows,
                         ds->d_mask, ds->n, f, (const uint64_t *)dk, radius, thresh, ireach, (float *)dq);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out + off, dq, sizeof(float) * (size_t)c, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (ret)
      for (int64_t i = 0; i < c; i++) {
        const int64_t r = (f + i) % ds->n;
        ret[off + i] = (!ds->all_masked.empty() && ds->all_masked[(size_t)r]) ? 0 : 1;
      }
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// som_training
// ---------------------------------------------------------------------------------
// Bubble threshold for maps with both sides <= 1024: every squared lattice distance is then an exact
// multiple of 1/4, so only K = floor(4 T) matters (T = bubble_threshold(radius)) and K/4 is an
// equivalent threshold.  K is constant while the radius stays inside [f(K/4), f((K+1)/4)), f(r) =
// (float)sqrt((double)r) -- the schedule moves the radius by ~1e-4 per iteration, so the exact
// search runs only when a lattice distance is crossed.
struct ThreshCache {
  bool valid = false;
  float lo = 0.f, hi = 0.f, thresh = -1.f;
  float get(float radius) {
    if (!(radius >= 0.0f)) return -1.0f;
    if (valid && radius >= lo && radius < hi) return thresh;
    const float T = bubble_threshold(radius);
    const double k = std::floor(4.0 * (double)T);
    thresh = (float)(k / 4.0);
    lo = (float)std::sqrt(k / 4.0);
    hi = (float)std::sqrt((k + 1.0) / 4.0);
    valid = hi > lo;
    return thresh;
  }
};

static int som_scalars(const somhip_codebook *cb, const somhip_dataset *ds, const somhip_som_params *p,
                       int64_t it0, int64_t cnt, int64_t row0, StepScalars *out) {
  const bool gauss = cb->v.neigh == SOMHIP_NEIGH_GAUSSIAN;
  const bool small_map = cb->v.xdim <= 1024 && cb->ydim <= 1024;
  ThreshCache tc;
  for (int64_t j = 0; j < cnt; j++) {
    int64_t le = it0 + j, r = (row0 + j) % ds->n;
    float trad = radius_at(le, p->length, p->radius);
    float talp = alpha_at(p->alpha_type, le, p->length, p->alpha);
    float w = ds->weight.empty() ? 0.0f : (float)ds->weight[(size_t)r];
    if (w > 0.0f && p->use_weights) talp = weighted_alpha(talp, w);
    StepScalars s;
    s.alpha = talp;
    s.thresh = gauss ? trad : (small_map ? tc.get(trad) : bubble_threshold(trad));
    s.fixed = -1;
    // lattice rows a neighbourhood of this radius can span: hexa rows are sqrt(0.75)
    // apart (som_rout.c:451), rect rows 1 apart; +1 keeps it conservative
    double reach = gauss ? 1e9 : (trad > 0.0f ? (double)trad / (cb->v.topol == SOMHIP_TOPOL_RECT ? 1.0 : 0.8660254037844386) + 1.0 : 1.0);
    s.reach = reach > 1e6 ? 1000000 : (int32_t)reach;
    if (!ds->all_masked.empty() && ds->all_masked[(size_t)r]) s.reach = -1;
    if (p->use_fixed && !ds->fixed_xy.empty() && ds->fixed_xy[(size_t)(2 * r)] >= 0) {
      int fx = ds->fixed_xy[(size_t)(2 * r)], fy = ds->fixed_xy[(size_t)(2 * r + 1)];
      s.fixed = fy * cb->v.xdim + fx;      // inverse of som_rout.c:641-642
      if (s.reach < 0) s.reach = reach > 1e6 ? 1000000 : (int32_t)reach;
    }
    out[j] = s;
  }
  return 0;
}

constexpr int ONLINE_U = 8;    // chunks (KiB) per register buffer; two buffers per wave
template <bool G, bool M>
static void launch_online(somhip_engine *e, const somhip_codebook *cb, const somhip_dataset *ds,
                          const int64_t *prev_row, const int64_t *cur_row, int has_prev, int has_cur,
                          const uint64_t *prev_slot, uint64_t *cur_slot, const StepScalars *prev_sc,
                          const StepScalars *cur_sc) {
  LaunchTimer t(e, KID_SOM_ONLINE_STEP);
  hipLaunchKernelGGL((k_som_online_step<G, M, ONLINE_U>), dim3((unsigned)((cb->v.ngroups + 3) / 4)), dim3(256), 0,
                     e->stream, cb->v, ds->d_rows, (const uint8_t *)ds->d_mask, prev_row, cur_row,
                     has_prev, has_cur, prev_slot, cur_slot, prev_sc, cur_sc);
}
static void launch_online_any(somhip_engine *e, const somhip_codebook *cb, const somhip_dataset *ds, bool G, bool M,
                              const int64_t *prev_row, const int64_t *cur_row, int has_prev, int has_cur,
                              const uint64_t *prev_slot, uint64_t *cur_slot, const StepScalars *prev_sc,
                              const StepScalars *cur_sc) {
#define GO(GG, MM) launch_online<GG, MM>(e, cb, ds, prev_row, cur_row, has_prev, has_cur, prev_slot, cur_slot, prev_sc, cur_sc)
  if (G && M) GO(true, true); else if (G) GO(true, false); else if (M) GO(false, true); else GO(false, false);
#undef GO
}

// The online algorithm is one small launch per iteration; a full chunk of them is captured
// once into a hipGraph (all per-iteration inputs live in device arrays the host refreshes) and
// replayed, which removes the per-launch host cost that bounds small maps.
static int som_train_online(somhip_codebook *cb, somhip_dataset *ds, const somhip_som_params *p,
                            int32_t *trace_index, float *trace_diff) {
  somhip_engine *e = cb->e;
  cb->prep_current = false;
  const int64_t CH = ONLINE_CHUNK;
  const bool G = cb->v.neigh == SOMHIP_NEIGH_GAUSSIAN, M = ds->d_mask != nullptr;
  void *dslot, *dsc, *drow;
  // entry 0 of the arrays carries the last iteration of the previous chunk
  CHK(engine_scratch(e, 3, sizeof(uint64_t) * (size_t)(CH + 1), &dslot));
  CHK(engine_scratch(e, 4, sizeof(StepScalars) * (size_t)(CH + 1), &dsc));
  CHK(engine_scratch(e, 2, sizeof(int64_t) * (size_t)(CH + 1), &drow));
  uint64_t *slot = (uint64_t *)dslot;
  StepScalars *sc = (StepScalars *)dsc;
  int64_t *rowidx = (int64_t *)drow;
  std::vector<StepScalars> hsc((size_t)CH + 1);
  std::vector<uint64_t> hslot((size_t)CH + 1);
  std::vector<int64_t> hrow((size_t)CH + 1);
  // entry 0 before the first iteration: "teaches nothing" (reach < 0), so has_prev can always be 1
  hsc[0].alpha = 0.f; hsc[0].thresh = -1.f; hsc[0].fixed = -1; hsc[0].reach = -1;
  hrow[0] = 0;
  HIPCHK(hipMemcpyAsync(sc, hsc.data(), sizeof(StepScalars), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(rowidx, hrow.data(), sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemsetAsync(slot, 0xFF, sizeof(uint64_t), e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));

  // graph of one full chunk, cached per engine while its arguments stay the same
  OnlineGraphKey key{cb->v.tiles, cb->v.n, cb->v.d, cb->v.patch_w, cb->v.row_offset, cb->v.xdim, cb->v.topol,
                     ds->d_rows, ds->d_mask, slot, sc, rowidx, G, M};
  const bool want_graph = !e->timing && p->count >= CH && !getenv("SOMHIP_NO_GRAPH");
  if (want_graph && !(e->online_graph_exec && e->online_graph_key == key)) {
    if (e->online_graph_exec) { (void)hipGraphExecDestroy(e->online_graph_exec); e->online_graph_exec = nullptr; }
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    for (int64_t j = 0; j < CH; j++)
      launch_online_any(e, cb, ds, G, M, rowidx + j, rowidx + j + 1, 1, 1, slot + j, slot + j + 1, sc + j, sc + j + 1);
    HIPCHK(hipStreamEndCapture(e->stream, &graph));
    HIPCHK(hipGraphInstantiate(&e->online_graph_exec, graph, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(graph));
    e->online_graph_key = key;
  }

  bool have_prev = false;
  int64_t last_row = 0;
  for (int64_t off = 0; off < p->count; off += CH) {
    int64_t c = std::min(CH, p->count - off);
    int64_t it0 = p->start_iter + off, row0 = (p->data_first + off) % ds->n;
    CHK(som_scalars(cb, ds, p, it0, c, row0, hsc.data() + 1));
    for (int64_t j = 0; j < c; j++) hrow[(size_t)j + 1] = (row0 + j) % ds->n;
    HIPCHK(hipMemcpyAsync(sc + 1, hsc.data() + 1, sizeof(StepScalars) * (size_t)c, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(rowidx + 1, hrow.data() + 1, sizeof(int64_t) * (size_t)c, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemsetAsync(slot + 1, 0xFF, sizeof(uint64_t) * (size_t)c, e->stream));
    if (want_graph && c == CH) {
      HIPCHK(hipGraphLaunch(e->online_graph_exec, e->stream));
    } else {
      for (int64_t j = 0; j < c; j++)
        launch_online_any(e, cb, ds, G, M, rowidx + j, rowidx + j + 1, 1, 1, slot + j, slot + j + 1, sc + j, sc + j + 1);
    }
    HIPCHK(hipGetLastError());
    have_prev = true;
    last_row = hrow[(size_t)c];
    if (trace_index || trace_diff) {
      HIPCHK(hipMemcpyAsync(hslot.data(), slot + 1, sizeof(uint64_t) * (size_t)c, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      for (int64_t j = 0; j < c; j++) {
        int32_t idx; float df;
        const StepScalars &s = hsc[(size_t)j + 1];
        if (s.fixed >= 0) { idx = -3; df = -1.0f; }
        else if (s.reach < 0) { idx = -2; df = -1.0f; }
        else decode_key(hslot[(size_t)j], false, &idx, &df);
        if (trace_index) trace_index[off + j] = idx;
        if (trace_diff) trace_diff[off + j] = df;
      }
    }
    // carry the last iteration's slot + scalars + row into entry 0 for the next chunk / the flush
    HIPCHK(hipMemcpyAsync(slot, slot + c, sizeof(uint64_t), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(sc, sc + c, sizeof(StepScalars), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(rowidx, rowidx + c, sizeof(int64_t), hipMemcpyDeviceToDevice, e->stream));
    // the host staging vectors are reused by the next chunk
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  (void)last_row;
  if (have_prev) {   // flush: apply the last iteration's update
    launch_online_any(e, cb, ds, G, M, rowidx, rowidx, 1, 0, slot, slot, sc, sc);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

static int som_update_run(somhip_codebook *cb, somhip_dataset *ds, int64_t data_first, int64_t count,
                          const uint64_t *d_keys, const StepScalars *d_sc) {
  somhip_engine *e = cb->e;
  cb->prep_current = false;
  const bool G = cb->v.neigh == SOMHIP_NEIGH_GAUSSIAN, M = ds->d_mask != nullptr;
#ifndef SOMHIP_UPD_QW
#define SOMHIP_UPD_QW 8
#endif
#ifndef SOMHIP_UPD_TB
#define SOMHIP_UPD_TB 32
#endif
  constexpr int QW = SOMHIP_UPD_QW, TB = SOMHIP_UPD_TB;
  void *dbxy, *dcnt, *dent;
  CHK(engine_scratch(e, 8, sizeof(int2) * (size_t)count, &dbxy));
  CHK(engine_scratch(e, 9, sizeof(uint32_t) * (size_t)cb->v.ngroups, &dcnt));
  CHK(engine_scratch(e, 10, sizeof(MemberEntry) * (size_t)cb->v.ngroups * (size_t)count, &dent));
  if (G) {                                                // the gaussian update needs the decoded winners itself
    LaunchTimer t(e, KID_DECODE);
    hipLaunchKernelGGL(k_decode_winners, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, e->stream,
                       d_keys, d_sc, count, cb->v.xdim, (int2 *)dbxy);
    HIPCHK(hipGetLastError());
  }
  {
    LaunchTimer t(e, KID_MEMBERS);
    if (G) hipLaunchKernelGGL(k_som_members<true>, dim3((unsigned)cb->v.ngroups), dim3(256), 0, e->stream, cb->v, count,
                              (const int2 *)dbxy, (const uint64_t *)nullptr, d_sc, (uint32_t *)dcnt, (MemberEntry *)dent, e->d_stats);
    else hipLaunchKernelGGL(k_som_members<false>, dim3((unsigned)cb->v.ngroups), dim3(256), 0, e->stream, cb->v, count,
                            (const int2 *)nullptr, d_keys, d_sc, (uint32_t *)dcnt, (MemberEntry *)dent, e->d_stats);
  }
  HIPCHK(hipGetLastError());
  uint32_t *dorder = nullptr;
  if (cb->v.ngroups <= 8192 && !getenv("SOMHIP_NO_ORDER")) {
    void *p_;
    CHK(engine_scratch(e, 0, sizeof(uint32_t) * (size_t)cb->v.ngroups, &p_));
    dorder = (uint32_t *)p_;
    LaunchTimer t(e, KID_DECODE);
    hipLaunchKernelGGL(k_order_groups, dim3((unsigned)((cb->v.ngroups * 8 + 255) / 256)), dim3(256), 0, e->stream,
                       (const uint32_t *)dcnt, (int)cb->v.ngroups, dorder);
    HIPCHK(hipGetLastError());
  }
  dim3 grid((unsigned)cb->v.ngroups, (unsigned)((cb->v.d4 + 4 * QW - 1) / (4 * QW)));
  LaunchTimer t(e, KID_SOM_UPDATE_RUN);
#define GO(GG, MM)                                                                                   \
  hipLaunchKernelGGL((k_som_update_run<QW, TB, GG, MM>), grid, dim3(256), 0, e->stream, cb->v, ds->d_rows, \
                     (const uint8_t *)ds->d_mask, ds->n, data_first, count, (const int2 *)dbxy, d_sc,   \
                     (const uint32_t *)dcnt, (const MemberEntry *)dent, (const uint32_t *)dorder)
  if (G && M) GO(true, true); else if (G) GO(true, false); else if (M) GO(false, true); else GO(false, false);
#undef GO
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int somhip_som_batch_update(somhip_codebook *cb, somhip_dataset *ds,
                                       const somhip_som_params *p, int64_t batch_start_iter,
                                       int64_t count, int64_t data_first, const uint64_t *dev_keys) {
  CHK(check_pair(cb, ds, "somhip_som_batch_update"));
  if (cb->v.topol < SOMHIP_TOPOL_HEXA) return fail("somhip_som_batch_update: codebook is not a map");
  if (count <= 0) return 0;
  somhip_engine *e = cb->e;
  HIPCHK(hipSetDevice(e->device));
  void *dsc, *hsc;
  int slot;
  CHK(engine_scratch(e, 4, sizeof(StepScalars) * (size_t)count, &dsc));
  CHK(pin_acquire(e, sizeof(StepScalars) * (size_t)count, &hsc, &slot));
  CHK(som_scalars(cb, ds, p, batch_start_iter, count, data_first % ds->n, (StepScalars *)hsc));
  CHK(pin_upload(e, slot, dsc, sizeof(StepScalars) * (size_t)count));
  return som_update_run(cb, ds, data_first % ds->n, count, dev_keys, (const StepScalars *)dsc);   // asynchronous
}

static int som_train_batched(somhip_codebook *cb, somhip_dataset *ds, const somhip_som_params *p,
                             int32_t *trace_index, float *trace_diff) {
  somhip_engine *e = cb->e;
  const int64_t B = p->batch;
  void *dkeys, *dsc;
  CHK(engine_scratch(e, 3, sizeof(uint64_t) * (size_t)B, &dkeys));
  CHK(engine_scratch(e, 4, sizeof(StepScalars) * (size_t)B, &dsc));
  std::vector<uint64_t> hk((size_t)B);
  const bool trace = trace_index || trace_diff;
  // batches are aligned to the schedule (iteration 0, B, 2B, ...), as in the oracle; the host runs
  // ahead of the GPU (scalars go through a ring of pinned buffers) unless a trace is wanted
  for (int64_t off = 0; off < p->count;) {
    int64_t it0 = p->start_iter + off;
    int64_t c = std::min(B - (it0 % B), p->count - off);
    int64_t row0 = (p->data_first + off) % ds->n;
    void *hscv;
    int slot;
    CHK(pin_acquire(e, sizeof(StepScalars) * (size_t)c, &hscv, &slot));
    StepScalars *hsc = (StepScalars *)hscv;
    CHK(som_scalars(cb, ds, p, it0, c, row0, hsc));
    CHK(pin_upload(e, slot, dsc, sizeof(StepScalars) * (size_t)c));
    CHK(scan_keys_top1(cb, ds, row0, c, (uint64_t *)dkeys));
    CHK(som_update_run(cb, ds, row0, c, (const uint64_t *)dkeys, (const StepScalars *)dsc));
    if (trace) {
      HIPCHK(hipMemcpyAsync(hk.data(), dkeys, sizeof(uint64_t) * (size_t)c, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      for (int64_t j = 0; j < c; j++) {
        int32_t idx; float df;
        if (hsc[j].fixed >= 0) { idx = -3; df = -1.0f; }
        else if (hsc[j].reach < 0) { idx = -2; df = -1.0f; }
        else decode_key(hk[(size_t)j], false, &idx, &df);
        if (trace_index) trace_index[off + j] = idx;
        if (trace_diff) trace_diff[off + j] = df;
      }
    }
    off += c;
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int somhip_som_train(somhip_codebook *cb, somhip_dataset *ds, const somhip_som_params *p,
                                int32_t *trace_index, float *trace_diff) {
  CHK(check_pair(cb, ds, "somhip_som_train"));
  if (!p) return fail("somhip_som_train: null params");
  if (cb->v.topol < SOMHIP_TOPOL_HEXA || (cb->v.neigh != SOMHIP_NEIGH_BUBBLE && cb->v.neigh != SOMHIP_NEIGH_GAUSSIAN))
    return fail("som_training: can't set SOM parameters");                 // som_rout.c:576-580
  if (p->length <= 0 || p->count < 0 || p->start_iter < 0 || p->start_iter + p->count > p->length)
    return fail("somhip_som_train: iterations [%lld,%lld) outside schedule of %lld",
                (long long)p->start_iter, (long long)(p->start_iter + p->count), (long long)p->length);
  if (cb->v.row_offset != 0 || cb->n_global != cb->v.n)
    return fail("somhip_som_train: sharded codebook -- use somhip_batch_winner_keys + somhip_som_batch_update");
  if (p->count == 0) return 0;
  HIPCHK(hipSetDevice(cb->e->device));
  if (p->batch <= 1) return som_train_online(cb, ds, p, trace_index, trace_diff);
  return som_train_batched(cb, ds, p, trace_index, trace_diff);
}

// ---------------------------------------------------------------------------------
// lvq*_training
// ---------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------
// lvq1/olvq1/lvq2/lvq3_training, exact batched form (kernels.hpp K6): per batch one
// frozen-codebook top-8 scan, then k_lvq_batch_apply walks the samples in order.  The
// device tells how many samples it could certify (ctl.consumed); the next batch starts
// there.  Bit-identical to the online loop.
// ---------------------------------------------------------------------------------
static int lvq_cache_slots(int d4) {
  const int64_t budget = 152 * 1024;                 // dynamic LDS; ~6.5 KiB static on top (160 KiB per workgroup)
  int64_t s = (budget - 3 * (int64_t)d4 * 16) / ((int64_t)d4 * 16);
  return (int)std::min<int64_t>(s, LVQ_BT);
}

static int lvq_train_batched(somhip_codebook *cb, somhip_dataset *ds, const somhip_lvq_params *p, int knn,
                             float *talpha, int32_t *trace_index, float *trace_diff) {
  somhip_engine *e = cb->e;
  const int slots = lvq_cache_slots(cb->v.d4);
  const size_t dyn = ((size_t)cb->v.d4 * slots + 3 * (size_t)cb->v.d4) * sizeof(float4);
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(hipFuncSetAttribute((const void *)k_lvq_batch_apply, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    attr_set = true;
  }
  const int64_t BMAX = 1024;
  void *dcand, *dfin, *dst, *dctl;
  CHK(engine_scratch(e, 3, sizeof(uint64_t) * (size_t)BMAX * LVQ_K0, &dcand));
  CHK(engine_scratch(e, 4, sizeof(LvqStep) * (size_t)BMAX, &dst));
  CHK(engine_scratch(e, 13, sizeof(uint64_t) * (size_t)BMAX * 2, &dfin));    // (5..7, 11, 12 belong to the pre-filter)
  CHK(engine_scratch(e, 14, sizeof(LvqBatchCtl), &dctl));
  void *dmod;
  CHK(engine_scratch(e, 16, sizeof(int32_t) * LVQ_BT, &dmod));
  cb->prep_current = false;
  std::vector<LvqStep> hst((size_t)BMAX);
  std::vector<uint64_t> hfin((size_t)BMAX * 2);
  const float ratio = (1 - p->winlen) / (1 + p->winlen);                  // lvq_rout.c:770, fp32
  const bool want_trace = trace_index || trace_diff;
  int64_t B = std::min<int64_t>(BMAX, std::max<int64_t>(32, slots));
  int64_t off = 0;
  uint64_t n_batches = 0;
  while (off < p->count) {
    const int64_t c = std::min(B, p->count - off);
    const int64_t it0 = p->start_iter + off, row0 = (p->data_first + off) % ds->n;
    for (int64_t j = 0; j < c; j++) {
      LvqStep s;
      s.kind = p->kind;
      s.alpha = alpha_at(p->alpha_type, it0 + j, p->length, p->alpha);
      s.alpha_clamp = p->alpha;
      s.win_ratio = ratio;
      s.epsilon = p->epsilon;
      s.label = ds->labels[(size_t)((row0 + j) % ds->n)];
      hst[(size_t)j] = s;
    }
    HIPCHK(hipMemcpyAsync(dst, hst.data(), sizeof(LvqStep) * (size_t)c, hipMemcpyHostToDevice, e->stream));
    CHK(scan_keys_topk<LVQ_K0>(cb, ds, row0, c, (uint64_t *)dcand, knn == 2 ? 1 : 0));
    {
      LaunchTimer t(e, KID_LVQ_BATCH_APPLY);
      hipLaunchKernelGGL(k_lvq_batch_apply, dim3(1), dim3(LVQ_BT), dyn, e->stream, cb->v, ds->d_rows, ds->n, row0,
                         (int)c, (const int32_t *)cb->d_labels, p->kind == SOMHIP_OLVQ1 ? cb->d_talpha : nullptr,
                         (const uint64_t *)dcand, (const LvqStep *)dst, knn, slots, (uint64_t *)dfin,
                         (int32_t *)dmod, (LvqBatchCtl *)dctl);
    }
    HIPCHK(hipGetLastError());
    LvqBatchCtl ctl;
    HIPCHK(hipMemcpyAsync(&ctl, dctl, sizeof(ctl), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (ctl.consumed < 0 || ctl.consumed > c) return fail("somhip_lvq_train: bad batch control block");
    if (ctl.consumed == 0) {
      // cannot happen: with an empty cache every winner comes from the frozen list and one
      // or two slots always fit; refuse to spin
      return fail("somhip_lvq_train: batch made no progress (reason %d)", ctl.reason);
    }
    if (want_trace) {
      HIPCHK(hipMemcpy(hfin.data(), dfin, sizeof(uint64_t) * 2 * (size_t)ctl.consumed, hipMemcpyDeviceToHost));
      for (int64_t j = 0; j < ctl.consumed; j++)
        for (int k = 0; k < knn; k++) {
          int32_t idx; float df;
          decode_key(hfin[(size_t)(2 * j + k)], knn == 2, &idx, &df);
          if (trace_index) trace_index[(off + j) * knn + k] = idx;
          if (trace_diff) trace_diff[(off + j) * knn + k] = df;
        }
    }
    off += ctl.consumed;
    n_batches++;
    // the bf16 copies / norms of exactly the rows this batch corrected, so that the next batch's
    // pre-filter (big codebooks, scan_keys_topk) needs no pass over the whole codebook
    if (cb->d_chi && cb->d_cn && e->scan_mode == SOMHIP_SCAN_MFMA_BF16 && !getenv("SOMHIP_ALWAYS_PREP")) {
      if (ctl.nmod > 0)
        hipLaunchKernelGGL(k_prep_rows_bf16, dim3((unsigned)((ctl.nmod + 3) / 4)), dim3(256), 0, e->stream, cb->v,
                           (cb->v.d4 + 1) / 2, (const int32_t *)dmod, (int)ctl.nmod, cb->d_cn, cb->d_chi, cb->d_clo);
      HIPCHK(hipGetLastError());
      cb->prep_current = true;
    }
    if (ctl.reason == 1) e->lvq_stop_list++;
    if (ctl.reason == 2) e->lvq_stop_cache++;
    for (int k = 0; k < 4; k++) e->lvq_cycles[k] += (uint64_t)ctl.cycles[k];
    // next batch: a little more than what this one managed (the scan of samples that were not
    // certified is wasted), never below 32
    B = std::min<int64_t>(BMAX, std::max<int64_t>(32, (int64_t)ctl.consumed + ctl.consumed / 4 + 8));
  }
  e->lvq_batches += n_batches;
  e->lvq_samples += (uint64_t)p->count;
  return 0;
}

extern "C" int somhip_lvq_train(somhip_codebook *cb, somhip_dataset *ds, const somhip_lvq_params *p,
                                float *talpha, int32_t *trace_index, float *trace_diff) {
  CHK(check_pair(cb, ds, "somhip_lvq_train"));
  cb->prep_current = false;
  if (!p) return fail("somhip_lvq_train: null params");
  if (p->kind < SOMHIP_LVQ1 || p->kind > SOMHIP_LVQ3) return fail("Unknown LVQ type %d", p->kind);
  if (!cb->d_labels) return fail("somhip_lvq_train: codebook has no labels");
  if (ds->labels.empty()) return fail("somhip_lvq_train: data has no labels");
  if (ds->d_mask) return fail("somhip_lvq_train: masked samples are not supported by the LVQ loops");
  if (p->kind == SOMHIP_OLVQ1 && !talpha) return fail("somhip_lvq_train: OLVQ1 needs talpha");
  if (p->length <= 0 || p->count < 0 || p->start_iter + p->count > p->length)
    return fail("somhip_lvq_train: iterations outside schedule");
  if (cb->v.row_offset != 0 || cb->n_global != cb->v.n) return fail("somhip_lvq_train: sharded codebook not supported");
  const int knn = (p->kind >= SOMHIP_LVQ2) ? 2 : 1;
  if (knn == 2 && cb->v.n < 2) return fail("somhip_lvq_train: LVQ2/LVQ3 need at least two code rows");
  if (p->count == 0) return 0;
  somhip_engine *e = cb->e;
  HIPCHK(hipSetDevice(e->device));
  if (p->kind == SOMHIP_OLVQ1) {
    if (!cb->d_talpha) HIPCHK(hipMalloc((void **)&cb->d_talpha, sizeof(float) * (size_t)cb->v.n));
    HIPCHK(hipMemcpyAsync(cb->d_talpha, talpha, sizeof(float) * (size_t)cb->v.n, hipMemcpyHostToDevice, e->stream));
  }
  // exact batched engine unless asked otherwise (SOMHIP_LVQ_ONLINE=1), or the row does not fit the cache
  if (!getenv("SOMHIP_LVQ_ONLINE") && cb->v.patch_w == 0 && cb->v.d4 <= LVQ_BT && lvq_cache_slots(cb->v.d4) >= 8) {
    int rc = lvq_train_batched(cb, ds, p, knn, talpha, trace_index, trace_diff);
    if (rc) return rc;
    if (p->kind == SOMHIP_OLVQ1)
      HIPCHK(hipMemcpyAsync(talpha, cb->d_talpha, sizeof(float) * (size_t)cb->v.n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
  }
  const int64_t CH = 4096;
  const int nblk = (int)((cb->v.ngroups + 3) / 4);
  void *dpart, *dfinal, *dst;
  CHK(engine_scratch(e, 2, sizeof(uint64_t) * (size_t)nblk * 2 * 2, &dpart));
  CHK(engine_scratch(e, 3, sizeof(uint64_t) * (size_t)(CH + 1) * 2, &dfinal));
  CHK(engine_scratch(e, 4, sizeof(LvqStep) * (size_t)(CH + 1), &dst));
  uint64_t *part[2] = {(uint64_t *)dpart, (uint64_t *)dpart + (size_t)nblk * 2};
  uint64_t *fin = (uint64_t *)dfinal;
  LvqStep *st = (LvqStep *)dst;
  std::vector<LvqStep> hst((size_t)CH + 1);
  std::vector<uint64_t> hfin((size_t)(CH + 1) * 2);
  const float ratio = (1 - p->winlen) / (1 + p->winlen);                  // lvq_rout.c:770, fp32
  int64_t prev_row = 0;
  bool have_prev = false;
  int flip = 0;
  // fin[j] receives the merged winners of chunk-iteration j-1 when iteration j launches;
  // the last one of a chunk lands in fin[c] when the next chunk's first launch (or the
  // flush) runs, so traces are read one launch late.
  int64_t pending_trace = -1;   // global offset of the iteration whose winners arrive next
  for (int64_t off = 0; off < p->count; off += CH) {
    int64_t c = std::min(CH, p->count - off);
    int64_t it0 = p->start_iter + off, row0 = (p->data_first + off) % ds->n;
    for (int64_t j = 0; j < c; j++) {
      int64_t r = (row0 + j) % ds->n;
      LvqStep s;
      s.kind = p->kind;
      s.alpha = alpha_at(p->alpha_type, it0 + j, p->length, p->alpha);
      s.alpha_clamp = p->alpha;
      s.win_ratio = ratio;
      s.epsilon = p->epsilon;
      s.label = ds->labels[(size_t)r];
      hst[(size_t)j + 1] = s;
    }
    HIPCHK(hipMemcpyAsync(st + 1, hst.data() + 1, sizeof(LvqStep) * (size_t)c, hipMemcpyHostToDevice, e->stream));
    for (int64_t j = 0; j < c; j++) {
      int64_t cur_row = (row0 + j) % ds->n;
      {
        LaunchTimer t(e, KID_LVQ_ONLINE_STEP);
        hipLaunchKernelGGL(k_lvq_online_step, dim3((unsigned)nblk), dim3(256), 0, e->stream, cb->v,
                           ds->d_rows, (const int32_t *)cb->d_labels, cb->d_talpha, prev_row, cur_row,
                           have_prev ? 1 : 0, 1, knn, (const uint64_t *)part[flip], nblk, part[flip ^ 1],
                           fin + 2 * j, (const LvqStep *)(st + j));
      }
      flip ^= 1;
      prev_row = cur_row;
      have_prev = true;
    }
    HIPCHK(hipGetLastError());
    // winners of iterations (off-1 .. off+c-2) are now in fin[0..c-1]
    if (trace_index || trace_diff) {
      HIPCHK(hipMemcpyAsync(hfin.data(), fin, sizeof(uint64_t) * 2 * (size_t)c, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      for (int64_t j = 0; j < c; j++) {
        int64_t it = off + j - 1;
        if (it < 0) continue;
        for (int k = 0; k < knn; k++) {
          int32_t idx; float df;
          decode_key(hfin[(size_t)(2 * j + k)], knn == 2, &idx, &df);
          if (trace_index) trace_index[it * knn + k] = idx;
          if (trace_diff) trace_diff[it * knn + k] = df;
        }
      }
    }
    HIPCHK(hipMemcpyAsync(st, st + c, sizeof(LvqStep), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    (void)pending_trace;
  }
  // flush: apply the last iteration's correction; its winners land in fin[0]
  {
    LaunchTimer t(e, KID_LVQ_ONLINE_STEP);
    hipLaunchKernelGGL(k_lvq_online_step, dim3((unsigned)nblk), dim3(256), 0, e->stream, cb->v, ds->d_rows,
                       (const int32_t *)cb->d_labels, cb->d_talpha, prev_row, prev_row, 1, 0, knn,
                       (const uint64_t *)part[flip], nblk, part[flip ^ 1], fin, (const LvqStep *)st);
  }
  HIPCHK(hipGetLastError());
  if (trace_index || trace_diff) {
    HIPCHK(hipMemcpyAsync(hfin.data(), fin, sizeof(uint64_t) * 2, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    int64_t it = p->count - 1;
    for (int k = 0; k < knn; k++) {
      int32_t idx; float df;
      decode_key(hfin[(size_t)k], knn == 2, &idx, &df);
      if (trace_index) trace_index[it * knn + k] = idx;
      if (trace_diff) trace_diff[it * knn + k] = df;
    }
  }
  if (p->kind == SOMHIP_OLVQ1)
    HIPCHK(hipMemcpyAsync(talpha, cb->d_talpha, sizeof(float) * (size_t)cb->v.n, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}
