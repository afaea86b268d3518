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
#include <exception>
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

// No C++ exception may cross the C ABI (the header promises "nothing here aborts"): every extern "C" body is a
// function-try-block that ends in one of these.  The HIP runtime itself throws on some misuse (a destroyed
// stream handle gave std::bad_variant_access out of hipStreamSynchronize).
static int abi_caught(const char *who) {
  try { throw; }
  catch (const std::exception &ex) { return fail("%s: C++ exception: %s", who, ex.what()); }
  catch (...) { return fail("%s: unknown C++ exception", who); }
}
#define ABI_CATCH(name) catch (...) { return abi_caught(#name); }
#define ABI_CATCH_VOID(name) catch (...) { (void)abi_caught(#name); }

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
  KID_SOM_UPDATE_BUBBLE_S,
  KID_LVQ_COMPONENTS,
  KID_SOM_UPDATE_GEMM,
  KID_DIST_L2,
  KID_L2_SELECT,
  KID_COUNT
};
static const char *kKernelNames[KID_COUNT] = {
    "k_scan_exact", "k_som_update_run", "k_som_online_step", "k_lvq_online_step",
    "k_pack_samples", "k_merge_topk", "k_scan_masked", "k_layout", "k_decode_winners",
    "k_dist_mfma", "k_rerank", "k_norms_tau", "k_som_members",
    "k_rerank_select", "k_rerank_pairs", "k_dist_mfma_bf16", "k_lvq_batch_apply", "k_som_update_bubble_s", "k_lvq_components", "k_som_update_gemm", "k_dist_l2", "k_l2_select"};
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
  int update_mode = SOMHIP_UPDATE_EXACT;
  double tau_scale = 1.0;                    // >= 1: widen the pre-filter window (experiments only)
  unsigned long long *d_stats = nullptr;     // [4] re-rank statistics (device)
  uint64_t samples_searched = 0;
  // somhip_shard_winner_begin / _refine / _finish: which search is under way on this engine (0 = none)
  int xc_phase = 0;
  const void *xc_cb = nullptr, *xc_ds = nullptr;
  int64_t xc_first = 0, xc_count = 0;
  uint64_t lvq_batches = 0, lvq_samples = 0;   // exact batched LVQ: rescans and samples
  uint64_t lvq_stop_list = 0, lvq_stop_cache = 0, lvq_cycles[4] = {0, 0, 0, 0};   // batches ended by an exhausted candidate list / a full cache
  int64_t lvq_batch_hint = 256;                   // batch size the exact LVQ engine starts its next call with
  uint64_t lvq_components = 0, lvq_largest = 0;   // independent components walked, and the sum of the largest one's size per batch
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
  void *scratch[32] = {nullptr};
  size_t scratch_bytes[32] = {0};
  // the mirrors created on this engine: destroying the engine first releases their device memory and orphans
  // them (their own destroy then only frees the host struct; any other call on them fails with a message)
  std::vector<somhip_codebook *> codebooks;
  std::vector<somhip_dataset *> datasets;
  bool lvq_apply_attr_set = false;             // hipFuncSetAttribute(k_lvq_batch_apply, ...) done on this device
  bool l2_lds_attr_set = false;                // ... and for k_dist_l2_lds
  bool l1r_attr_set = false;                   // ... and for k_dist_mfma_bf16_l1r
  int n_cus = 0;                               // compute units of the device (grid of the persistent kernels)
  int online_u = 8;                            // register-buffer depth of the online step kernel in use (SOMHIP_ONLINE_U)
  LvqCtl *lvq_hctl = nullptr;                  // pinned: read-backs of the LVQ batch loop's control block, one per batch in flight
  hipEvent_t lvq_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
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
static void codebook_release(somhip_codebook *cb);   // device memory of a mirror (defined with the mirrors below)
static void dataset_release(somhip_dataset *ds);

extern "C" int somhip_engine_create(int device, somhip_engine **out) try {
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
  auto init = [&]() -> int {
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    if (const char *ts = getenv("SOMHIP_TAU_SCALE")) { double v = atof(ts); if (v >= 1.0) e->tau_scale = v; }
    HIPCHK(hipMalloc((void **)&e->d_stats, (8 + 128 + 8 + 2) * sizeof(unsigned long long)));   // + 64 {rows, pairs} update counters + 8 gemm-walk counters + level-2 pairs + top-k pairs
    HIPCHK(hipMemset(e->d_stats, 0, (8 + 128 + 8 + 2) * sizeof(unsigned long long)));
    if (const char *um = getenv("SOMHIP_UPDATE_MODE")) e->update_mode = strcmp(um, "gemm") == 0 ? SOMHIP_UPDATE_GEMM : SOMHIP_UPDATE_EXACT;
    return 0;
  };
  if (int rc = init()) { somhip_engine_destroy(e); return rc; }
  *out = e;
  return 0;
} ABI_CATCH(somhip_engine_create)
extern "C" void somhip_engine_destroy(somhip_engine *e) try {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  // mirrors that outlive their engine become orphans: their device memory goes now, their handles stay valid
  // for somhip_codebook_destroy / somhip_dataset_destroy (any order of the destroy calls is fine)
  for (auto *cb : e->codebooks) codebook_release(cb);     // (these also clear the mirror's engine pointer)
  for (auto *ds : e->datasets) dataset_release(ds);
  for (auto &p : e->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto ev : e->pool) (void)hipEventDestroy(ev);
  for (int i = 0; i < 32; i++) if (e->scratch[i]) (void)hipFree(e->scratch[i]);
  if (e->d_stats) (void)hipFree(e->d_stats);
  if (e->online_graph_exec) (void)hipGraphExecDestroy(e->online_graph_exec);
  for (int i = 0; i < 4; i++) {
    if (e->pin_buf[i]) (void)hipHostFree(e->pin_buf[i]);
    if (e->pin_ev[i]) (void)hipEventDestroy(e->pin_ev[i]);
  }
  if (e->lvq_hctl) (void)hipHostFree(e->lvq_hctl);
  for (int i = 0; i < 8; i++) if (e->lvq_ev[i]) (void)hipEventDestroy(e->lvq_ev[i]);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
} ABI_CATCH_VOID(somhip_engine_destroy)
extern "C" int somhip_device_count(int *count) try {
  if (!count) return fail("somhip_device_count: null argument");
  int n = 0;
  const hipError_t er = hipGetDeviceCount(&n);
  if (er != hipSuccess) { *count = 0; return fail("somhip_device_count: %s", hipGetErrorString(er)); }
  *count = n;
  return 0;
} ABI_CATCH(somhip_device_count)
extern "C" void *somhip_engine_stream(somhip_engine *e) { return e ? (void *)e->stream : nullptr; }
extern "C" int somhip_engine_sync(somhip_engine *e) try {
  if (!e) return fail("somhip_engine_sync: null engine");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
} ABI_CATCH(somhip_engine_sync)
extern "C" int somhip_engine_set_scan_mode(somhip_engine *e, int mode) try {
  if (mode != SOMHIP_SCAN_DIRECT && mode != SOMHIP_SCAN_MFMA && mode != SOMHIP_SCAN_MFMA_BF16) return fail("somhip_engine_set_scan_mode: bad mode %d", mode);
  e->scan_mode = mode;
  return 0;
} ABI_CATCH(somhip_engine_set_scan_mode)
extern "C" int somhip_engine_set_update_mode(somhip_engine *e, int mode) try {
  if (!e) return fail("somhip_engine_set_update_mode: null engine");
  if (mode != SOMHIP_UPDATE_EXACT && mode != SOMHIP_UPDATE_GEMM) return fail("somhip_engine_set_update_mode: bad mode %d", mode);
  e->update_mode = mode;
  return 0;
} ABI_CATCH(somhip_engine_set_update_mode)
extern "C" int somhip_scan_stats(somhip_engine *e, uint64_t out[8]) try {
  unsigned long long h[8 + 128 + 8 + 1];
  HIPCHK(hipMemcpyAsync(h, e->d_stats, sizeof h, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int k = 0; k < 64; k++) { h[3] += h[8 + 2 * k]; h[4] += h[8 + 2 * k + 1]; }
  out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[3] = e->samples_searched; out[4] = h[3]; out[5] = h[4];
  out[6] = 0;
  for (int k = 0; k < 8; k++) out[6] += h[8 + 128 + k];
  out[7] = h[8 + 128 + 8];
  return 0;
} ABI_CATCH(somhip_scan_stats)
extern "C" int somhip_lvq_stats(somhip_engine *e, uint64_t out[12]) try {
  if (!e || !out) return fail("somhip_lvq_stats: null argument");
  out[0] = e->lvq_batches; out[1] = e->lvq_samples; out[2] = e->lvq_stop_list; out[3] = e->lvq_stop_cache;
  for (int k = 0; k < 4; k++) out[4 + k] = e->lvq_cycles[k];
  unsigned long long pairs = 0;                            // counted on the device (the top-k search does not wait for the host)
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(&pairs, e->d_stats + 8 + 128 + 8 + 1, sizeof pairs, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  out[8] = e->lvq_components; out[9] = e->lvq_largest; out[10] = pairs; out[11] = 0;
  return 0;
} ABI_CATCH(somhip_lvq_stats)
extern "C" int somhip_timing_enable(somhip_engine *e, int on) try {
  CHK(timing_flush(e));
  e->timing = on != 0;
  return 0;
} ABI_CATCH(somhip_timing_enable)
extern "C" int somhip_timing_select(somhip_engine *e, uint64_t kernel_mask) { e->timing_mask = kernel_mask; return 0; }
extern "C" int somhip_timing_reset(somhip_engine *e) try {
  CHK(timing_flush(e));
  for (int i = 0; i < KID_COUNT; i++) { e->launches[i] = 0; e->total_ms[i] = 0; }
  return 0;
} ABI_CATCH(somhip_timing_reset)
extern "C" int somhip_timing_get(somhip_engine *e, int k, int64_t *launches, double *total_ms) try {
  if (k < 0 || k >= KID_COUNT) return fail("somhip_timing_get: bad kernel id %d", k);
  CHK(timing_flush(e));
  if (launches) *launches = e->launches[k];
  if (total_ms) *total_ms = e->total_ms[k];
  return 0;
} ABI_CATCH(somhip_timing_get)
extern "C" int somhip_device_alloc(somhip_engine *e, int64_t bytes, void **p) try {
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMalloc(p, (size_t)std::max<int64_t>(bytes, 16)));
  return 0;
} ABI_CATCH(somhip_device_alloc)
extern "C" int somhip_device_free(somhip_engine *e, void *p) try {
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipFree(p));
  return 0;
} ABI_CATCH(somhip_device_free)
extern "C" int somhip_copy_to_host(somhip_engine *e, void *dst, const void *src, int64_t bytes) try {
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
} ABI_CATCH(somhip_copy_to_host)
extern "C" int somhip_copy_to_device(somhip_engine *e, void *dst, const void *src, int64_t bytes) try {
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
} ABI_CATCH(somhip_copy_to_device)

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
  bool prep_valid = false;         // d_cn / d_chi / d_clo describe the rows as they are now (set by a full k_prep_codes_bf16,
                                   // kept by the LVQ engine when it re-splits exactly the rows it corrected, cleared by every other writer)
  float *d_rowmajor = nullptr;     // [ngroups*64][d] fp32 rows, row-major: what the exact re-rank of single rows gathers from (a row is
                                   // 4 d contiguous bytes here, 16 bytes per KiB in the tiles); written by the full k_prep_codes_bf16
  bool rowmajor_valid = false;     // ... and current only together with prep_valid and until a partial re-split (LVQ) clears it
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
  cb->prep_valid = false;
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

static int codebook_create(somhip_engine *e, const float *rows, const int32_t *labels, int64_t n_rows, int dim,
                           int topol, int neigh, int xdim, int ydim, int64_t row_offset, int64_t n_global,
                           int patch_stride, int patch_phase, somhip_codebook **out) {
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
  e->codebooks.push_back(cb);
  cb->v.n = n_rows;
  cb->v.ngroups = (n_rows + WAVE - 1) / WAVE;
  cb->v.d = dim;
  cb->v.d4 = (dim + 3) / 4;
  cb->v.row_offset = row_offset;
  cb->v.xdim = xdim > 0 ? xdim : 1;
  cb->v.topol = topol;
  cb->v.neigh = neigh;
  cb->v.patch_w = 0;
  cb->v.patch_stride = 1;
  cb->v.patch_phase = 0;
  if (patch_stride > 1) {                         // interleaved shard: patch order is part of its definition
    cb->v.patch_w = xdim / 8;
    cb->v.patch_stride = patch_stride;
    cb->v.patch_phase = patch_phase;
  } else if (topol >= SOMHIP_TOPOL_HEXA && xdim % 8 == 0 && ydim % 8 == 0 && row_offset % (8 * (int64_t)xdim) == 0 &&
      n_rows % (8 * (int64_t)xdim) == 0 && !getenv("SOMHIP_LINEAR_ROWS"))
    cb->v.patch_w = xdim / 8;                     // 8x8-unit row groups (kernels.hpp CbView)
  cb->ydim = ydim;
  cb->n_global = n_global;
  size_t tile_bytes = (size_t)cb->v.ngroups * cb->v.d4 * WAVE * 4 * sizeof(float);
  auto fill = [&]() -> int {
    HIPCHK(hipMalloc((void **)&cb->v.tiles, tile_bytes));
    CHK(upload_rows(cb, rows));
    if (labels) {
      HIPCHK(hipMalloc((void **)&cb->d_labels, sizeof(int32_t) * (size_t)n_rows));
      HIPCHK(hipMemcpy(cb->d_labels, labels, sizeof(int32_t) * (size_t)n_rows, hipMemcpyHostToDevice));
    }
    return 0;
  };
  if (int rc = fill()) { somhip_codebook_destroy(cb); return rc; }
  *out = cb;
  return 0;
}

extern "C" int somhip_codebook_create(somhip_engine *e, const float *rows, const int32_t *labels,
                                      int64_t n_rows, int dim, int topol, int neigh, int xdim,
                                      int ydim, int64_t row_offset, int64_t n_global,
                                      somhip_codebook **out) try {
  return codebook_create(e, rows, labels, n_rows, dim, topol, neigh, xdim, ydim, row_offset, n_global, 1, 0, out);
} ABI_CATCH(somhip_codebook_create)

// Interleaved shards of a map: the map is cut into 8x8-unit patches, numbered row-major; shard s of S owns
// patches s, s+S, s+2S, ...  -- every shard sees every region of the map, so the neighbourhood updates of a
// batch are spread evenly over the ranks whatever the radius and wherever the winners fall (contiguous row
// blocks are not: the middle of the map is inside more neighbourhoods than its edges).
static int shard_patch_count(int xdim, int ydim, int shard_index, int shard_count, int64_t *count) {
  if (xdim <= 0 || ydim <= 0 || xdim % 8 || ydim % 8) return fail("interleaved shards need map sides that are multiples of 8 (%dx%d)", xdim, ydim);
  if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count) return fail("shard %d of %d", shard_index, shard_count);
  const int64_t patches = (int64_t)(xdim / 8) * (ydim / 8);
  *count = patches > shard_index ? (patches - shard_index + shard_count - 1) / shard_count : 0;
  return 0;
}
extern "C" int somhip_shard_units(int xdim, int ydim, int shard_index, int shard_count, int64_t *units,
                                  int64_t *n_units) try {
  int64_t np = 0;
  CHK(shard_patch_count(xdim, ydim, shard_index, shard_count, &np));
  if (n_units) *n_units = np * 64;
  if (!units) return 0;
  CbView v{};
  v.xdim = xdim; v.patch_w = xdim / 8; v.patch_stride = shard_count; v.patch_phase = shard_index;
  for (int64_t r = 0; r < np * 64; r++) units[r] = shard_count == 1 ? r : (int64_t)unit_of_row(v, r);   // one shard = the whole map, unit order
  return 0;
} ABI_CATCH(somhip_shard_units)
extern "C" int somhip_codebook_create_interleaved(somhip_engine *e, const float *rows, int64_t n_rows, int dim,
                                                  int topol, int neigh, int xdim, int ydim, int shard_index,
                                                  int shard_count, somhip_codebook **out) try {
  if (topol < SOMHIP_TOPOL_HEXA) return fail("somhip_codebook_create_interleaved: maps only");
  int64_t np = 0;
  CHK(shard_patch_count(xdim, ydim, shard_index, shard_count, &np));
  if (n_rows != np * 64) return fail("somhip_codebook_create_interleaved: shard %d of %d of a %dx%d map has %lld units, not %lld",
                                     shard_index, shard_count, xdim, ydim, (long long)(np * 64), (long long)n_rows);
  return codebook_create(e, rows, nullptr, n_rows, dim, topol, neigh, xdim, ydim, 0, (int64_t)xdim * ydim,
                         shard_count > 1 ? shard_count : 1, shard_count > 1 ? shard_index : 0, out);
} ABI_CATCH(somhip_codebook_create_interleaved)
extern "C" int somhip_codebook_upload(somhip_codebook *cb, const float *rows) try {
  if (!cb || !rows) return fail("somhip_codebook_upload: null argument");
  if (!cb->e) return fail("somhip_codebook_upload: the engine of this codebook was destroyed");
  HIPCHK(hipSetDevice(cb->e->device));
  return upload_rows(cb, rows);
} ABI_CATCH(somhip_codebook_upload)
extern "C" int somhip_codebook_download(somhip_codebook *cb, float *rows) try {
  if (!cb || !rows) return fail("somhip_codebook_download: null argument");
  if (!cb->e) return fail("somhip_codebook_download: the engine of this codebook was destroyed");
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
} ABI_CATCH(somhip_codebook_download)
static void codebook_release(somhip_codebook *cb) {
  if (cb->v.tiles) (void)hipFree(cb->v.tiles);
  if (cb->d_labels) (void)hipFree(cb->d_labels);
  if (cb->d_talpha) (void)hipFree(cb->d_talpha);
  if (cb->d_cn) (void)hipFree(cb->d_cn);
  if (cb->d_cnmax) (void)hipFree(cb->d_cnmax);
  if (cb->d_rowmajor) (void)hipFree(cb->d_rowmajor);
  cb->d_rowmajor = nullptr;
  if (cb->d_chi) (void)hipFree(cb->d_chi);
  if (cb->d_clo) (void)hipFree(cb->d_clo);
  cb->v.tiles = nullptr;
  cb->d_labels = nullptr; cb->d_talpha = nullptr; cb->d_cn = nullptr; cb->d_cnmax = nullptr;
  cb->d_chi = cb->d_clo = nullptr;
  cb->e = nullptr;
}
extern "C" void somhip_codebook_destroy(somhip_codebook *cb) try {
  if (!cb) return;
  if (somhip_engine *e = cb->e) {                       // an orphan (engine destroyed first) has nothing left on the device
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    codebook_release(cb);
    e->codebooks.erase(std::remove(e->codebooks.begin(), e->codebooks.end(), cb), e->codebooks.end());
  }
  delete cb;
} ABI_CATCH_VOID(somhip_codebook_destroy)

extern "C" int somhip_dataset_create(somhip_engine *e, const float *rows, int64_t n_rows, int dim,
                                     const uint8_t *mask, const int32_t *labels,
                                     const int16_t *weight, const int16_t *fixed_xy,
                                     somhip_dataset **out) try {
  if (!e || !rows || !out) return fail("somhip_dataset_create: null argument");
  if (n_rows <= 0 || dim <= 0) return fail("somhip_dataset_create: empty data set");
  HIPCHK(hipSetDevice(e->device));
  somhip_dataset *ds = new somhip_dataset();
  ds->e = e; ds->n = n_rows; ds->d = dim; ds->owns_rows = true;
  e->datasets.push_back(ds);
  auto fill = [&]() -> int {
  size_t bytes = sizeof(float) * (size_t)n_rows * dim;
  float *dr = nullptr;
  // 16 spare bytes: wave-uniform float4 reads of the last row never leave the buffer
  HIPCHK(hipMalloc((void **)&dr, bytes + 16));
  ds->d_rows = dr;
  HIPCHK(hipMemcpy(dr, rows, bytes, hipMemcpyHostToDevice));
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
  if (fixed_xy) {
    // -1,-1 = no fixed point; a negative coordinate is how "none" is written, so it cannot also be a position
    for (int64_t r = 0; r < n_rows; r++)
      if ((fixed_xy[2 * r] < 0) != (fixed_xy[2 * r + 1] < 0) || fixed_xy[2 * r] < -1 || fixed_xy[2 * r + 1] < -1)
        return fail("somhip_dataset_create: fixed point (%d,%d) of row %lld: negative coordinates are not supported",
                    fixed_xy[2 * r], fixed_xy[2 * r + 1], (long long)r);
    ds->fixed_xy.assign(fixed_xy, fixed_xy + 2 * n_rows);
  }
  return 0;
  };
  if (int rc = fill()) { somhip_dataset_destroy(ds); return rc; }
  *out = ds;
  return 0;
} ABI_CATCH(somhip_dataset_create)
extern "C" int somhip_dataset_wrap_device(somhip_engine *e, const float *dev_rows, int64_t n_rows,
                                          int dim, somhip_dataset **out) try {
  if (!e || !dev_rows || !out) return fail("somhip_dataset_wrap_device: null argument");
  if (n_rows <= 0 || dim <= 0) return fail("somhip_dataset_wrap_device: empty data set");
  somhip_dataset *ds = new somhip_dataset();
  ds->e = e; ds->n = n_rows; ds->d = dim; ds->d_rows = dev_rows; ds->owns_rows = false;
  e->datasets.push_back(ds);
  *out = ds;
  return 0;
} ABI_CATCH(somhip_dataset_wrap_device)
extern "C" int somhip_dataset_generate(somhip_engine *e, uint64_t seed, int k_centres, int dim, int64_t first_row,
                                       int64_t n_rows, int32_t *centres, somhip_dataset **out) try {
  if (!e || !out) return fail("somhip_dataset_generate: null argument");
  if (k_centres <= 0 || dim <= 0 || n_rows <= 0 || first_row < 0) return fail("somhip_dataset_generate: bad shape");
  HIPCHK(hipSetDevice(e->device));
  somhip_dataset *ds = new somhip_dataset();
  ds->e = e; ds->n = n_rows; ds->d = dim; ds->owns_rows = true;
  e->datasets.push_back(ds);
  int32_t *dcen = nullptr;
  auto fill = [&]() -> int {
  float *rows = nullptr;
  HIPCHK(hipMalloc((void **)&rows, sizeof(float) * (size_t)n_rows * dim + 16));   // same 16 spare bytes as dataset_create
  ds->d_rows = rows;
  if (centres) HIPCHK(hipMalloc((void **)&dcen, sizeof(int32_t) * (size_t)n_rows));
  const int64_t total = n_rows * dim;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65536);
  {
    LaunchTimer t(e, KID_LAYOUT);
    hipLaunchKernelGGL(k_gen_mixture, dim3(blocks), dim3(256), 0, e->stream, seed, k_centres, dim, first_row, n_rows, rows, dcen);
  }
  HIPCHK(hipGetLastError());
  if (centres) {
    HIPCHK(hipMemcpyAsync(centres, dcen, sizeof(int32_t) * (size_t)n_rows, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    ds->labels.assign(centres, centres + n_rows);
  }
  return 0;
  };
  const int rc = fill();
  if (dcen) (void)hipFree(dcen);
  if (rc) { somhip_dataset_destroy(ds); return rc; }
  *out = ds;
  return 0;
} ABI_CATCH(somhip_dataset_generate)
extern "C" int somhip_dataset_download_rows(somhip_dataset *ds, int64_t first, int64_t count, float *rows) try {
  if (!ds || !rows) return fail("somhip_dataset_download_rows: null argument");
  if (!ds->e) return fail("somhip_dataset_download_rows: the engine of this data set was destroyed");
  if (first < 0 || count < 0 || first + count > ds->n) return fail("somhip_dataset_download_rows: rows [%lld,%lld) outside %lld",
                                                                 (long long)first, (long long)(first + count), (long long)ds->n);
  if (count == 0) return 0;
  HIPCHK(hipSetDevice(ds->e->device));
  HIPCHK(hipMemcpyAsync(rows, ds->d_rows + first * ds->d, sizeof(float) * (size_t)count * ds->d, hipMemcpyDeviceToHost, ds->e->stream));
  HIPCHK(hipStreamSynchronize(ds->e->stream));
  return 0;
} ABI_CATCH(somhip_dataset_download_rows)
static void dataset_release(somhip_dataset *ds) {
  if (ds->owns_rows && ds->d_rows) (void)hipFree((void *)ds->d_rows);
  if (ds->d_mask) (void)hipFree(ds->d_mask);
  ds->d_rows = nullptr;
  ds->d_mask = nullptr;
  ds->e = nullptr;
}
extern "C" void somhip_dataset_destroy(somhip_dataset *ds) try {
  if (!ds) return;
  if (somhip_engine *e = ds->e) {
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    dataset_release(ds);
    e->datasets.erase(std::remove(e->datasets.begin(), e->datasets.end(), ds), e->datasets.end());
  }
  delete ds;
} ABI_CATCH_VOID(somhip_dataset_destroy)

// the rest of the host side, by stage (same translation unit)
#include "host_scan.inc"
#include "host_som.inc"
#include "host_lvq.inc"
#include "host_comm.inc"
