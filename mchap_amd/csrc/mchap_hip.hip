// libmchap_hip.so -- host side of the C ABI declared in include/mchap_hip.h (gfx950 only).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/mchap_hip.h"
#include "denovo_kernel.hpp"
#include "denovo_simt_kernel.hpp"
#include "denovo_spec_kernel.hpp"
#include "denovo_fillw_kernel.hpp"
#ifdef MCHAP_TEST_KERNELS
#include "denovo_fill_kernel.hpp"
#include "denovo_lane_kernel.hpp"
#endif
#include "exact_kernel.hpp"
#include "call_mcmc_kernel.hpp"
#include "posterior_kernel.hpp"

// ---- sampler objects ------------------------------------------------------------------------------------------
// Every instantiation of a sampler kernel is its own object file (spec_inst.hip, simt_inst.hip, ...: they compile in
// parallel) with two internal entry points: init (its copy of the constant log tables) and launch.
#define SPEC_LIST(X) X(2, 16) X(2, 32) X(2, 64) X(3, 16) X(3, 32) X(3, 64) X(4, 16) X(4, 32) X(4, 64) X(5, 32) X(5, 64) X(6, 32) X(6, 64) X(7, 64) X(8, 64)
// phased form: one chain per wavefront at every ploidy (the narrower groups were measured slower; they
// stay in the test library so that the parity suite keeps covering hand-over with several chains per wave)
#ifdef MCHAP_TEST_KERNELS
#define SPECP_LIST(X) X(2, 64) X(3, 64) X(4, 64) X(5, 64) X(6, 64) X(7, 64) X(8, 64) X(2, 16) X(3, 16) X(4, 16) X(4, 32) X(5, 32) X(6, 32)
#else
#define SPECP_LIST(X) X(2, 64) X(3, 64) X(4, 64) X(5, 64) X(6, 64) X(7, 64) X(8, 64)
#endif
// the phased form with the side-by-side evaluation of shallow units compiled in (MCHAP_SPEC_SBS, denovo_spec_kernel.hpp)
#define SPECS_LIST(X) X(2, 64) X(3, 64) X(4, 64) X(5, 64) X(6, 64) X(7, 64) X(8, 64)
#define SIMT_LIST(X) X(0) X(2) X(4) X(6) X(8)
#define V1_LIST(X) X(1) X(2) X(4) X(8) X(16)
#define LANE_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)

typedef int (*init_fn)(const double *, const double *);
typedef int (*simt_launch_fn)(const mchap::SimtParams *, unsigned, size_t, hipStream_t);
#define DECL_SPEC(k, g)                                                    \
  extern "C" int mchap_spec_init_##k##_##g(const double *, const double *); \
  extern "C" int mchap_spec_launch_##k##_##g(const mchap::SimtParams *, unsigned, size_t, hipStream_t);
#define DECL_SPECP(k, g)                                                    \
  extern "C" int mchap_specp_init_##k##_##g(const double *, const double *); \
  extern "C" int mchap_specp_launch_##k##_##g(const mchap::SimtParams *, unsigned, size_t, hipStream_t);
#define DECL_SIMT(k)                                                \
  extern "C" int mchap_simt_init_##k(const double *, const double *); \
  extern "C" int mchap_simt_launch_##k(const mchap::SimtParams *, unsigned, size_t, hipStream_t);
#define DECL_SPECS(k, g)                                                    \
  extern "C" int mchap_specs_init_##k##_##g(const double *, const double *); \
  extern "C" int mchap_specs_launch_##k##_##g(const mchap::SimtParams *, unsigned, size_t, hipStream_t); \
  extern "C" int mchap_specd_init_##k##_##g(const double *, const double *); \
  extern "C" int mchap_specd_launch_##k##_##g(const mchap::SimtParams *, unsigned, size_t, hipStream_t);
SPEC_LIST(DECL_SPEC)
SPECP_LIST(DECL_SPECP)
SPECS_LIST(DECL_SPECS)
SIMT_LIST(DECL_SIMT)
// the phased form's instantiations with decision contexts per genotype (one chain per wavefront; plain / side by side / deep)
#define DECL_SPECC(k)                                                                                        \
  extern "C" int mchap_specp_launchc_##k##_64(const mchap::SimtParams *, unsigned, size_t, hipStream_t); \
  extern "C" int mchap_specs_launchc_##k##_64(const mchap::SimtParams *, unsigned, size_t, hipStream_t); \
  extern "C" int mchap_specd_launchc_##k##_64(const mchap::SimtParams *, unsigned, size_t, hipStream_t);
DECL_SPECC(2) DECL_SPECC(3) DECL_SPECC(4) DECL_SPECC(5) DECL_SPECC(6) DECL_SPECC(7) DECL_SPECC(8)
// ... and its instantiation with 128-bit haplotype words (simt_inst.hip -DSIMT_WIDE): the general fallback for wide targets
extern "C" int mchap_simt_init_w(const double *, const double *);
extern "C" int mchap_simt_launch_w(const mchap::SimtParams *, unsigned, size_t, hipStream_t);
extern "C" int mchap_coast_launch(const mchap::SimtParams *, unsigned, int, hipStream_t);
// table completion of the phased sampler, one workgroup per chain and one wavefront per request (denovo_fillw_kernel.hpp): shipped
#define FILLW_LIST(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#define DECL_FILLW(k)                                                    \
  extern "C" int mchap_fillw_init_##k(const double *, const double *); \
  extern "C" int mchap_fillw_launch_##k(const mchap::SimtParams *, unsigned, size_t, hipStream_t);
FILLW_LIST(DECL_FILLW)
#ifdef MCHAP_TEST_KERNELS
#define FILL_LIST(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#define DECL_FILL(k)                                                \
  extern "C" int mchap_fill_init_##k(const double *, const double *); \
  extern "C" int mchap_fill_launch_##k(const mchap::SimtParams *, unsigned, size_t, hipStream_t);
FILL_LIST(DECL_FILL)
#define DECL_V1(r)                                                \
  extern "C" int mchap_v1_init_##r(const double *, const double *); \
  extern "C" int mchap_v1_launch_##r(const mchap::DenovoParams *, unsigned, unsigned, unsigned, size_t, hipStream_t);
#define DECL_LANE(k)                                                \
  extern "C" int mchap_lane_init_##k(const double *, const double *); \
  extern "C" int mchap_lane_launch_##k(const mchap::SimtParams *, int, int, unsigned, size_t, hipStream_t);
V1_LIST(DECL_V1)
LANE_LIST(DECL_LANE)
#endif
#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
#define DECL_SPEC_STATS(k, g) extern "C" int mchap_spec_stats_##k##_##g(unsigned long long *, int);
#define DECL_SPECP_STATS(k, g) extern "C" int mchap_specp_stats_##k##_##g(unsigned long long *, int);
#define DECL_SPECS_STATS(k, g) extern "C" int mchap_specs_stats_##k##_##g(unsigned long long *, int); extern "C" int mchap_specd_stats_##k##_##g(unsigned long long *, int);
SPEC_LIST(DECL_SPEC_STATS)
SPECP_LIST(DECL_SPECP_STATS)
SPECS_LIST(DECL_SPECS_STATS)
#ifdef MCHAP_TEST_KERNELS
extern "C" int mchap_lane_stats_4(unsigned long long *, int);
#endif
#endif

namespace {

thread_local char g_err[512] = "";
// Epochs of the likelihood caches: a fit whose packed genotypes leave the upper half of a cache tag free (ploidy x SNVs x bits per
// allele <= 32: configs[1]) writes its epoch there instead of clearing the chains' tables first (0.33 GB of stores per 10 000 loci);
// what an earlier call -- of any batch, on any stream -- left in a workspace carries another epoch and never matches.  The one
// piece of process-wide state in the library: a counter, never read back by anything but the next fit -- and only used when the
// caller does not name the call's epoch itself (mchap_denovo_cfg.cache_epoch).  After 2^31 - 2 fits the tables are cleared again
// as before.
std::atomic<uint64_t> g_cache_epoch{0};

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(MCHAP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

std::mutex g_init_mu;
bool g_init_done[64] = {false};

// A caller-owned pair of events (mchap_timer_create): the only profiling state there is, and it travels with the call.
struct Timer {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool recorded = false;
};
struct SamplerTimer {
  Timer *t;
  hipStream_t stream;
  SamplerTimer(void *timer, hipStream_t s) : t(reinterpret_cast<Timer *>(timer)), stream(s) {
    if (t) (void)hipEventRecord(t->e0, stream);
  }
  ~SamplerTimer() {
    if (!t) return;
    (void)hipEventRecord(t->e1, stream);
    t->recorded = true;
  }
};

struct SpecInst {
  int K, G;
  init_fn init;
  simt_launch_fn launch;
};
#define ROW_SPEC(k, g) {k, g, mchap_spec_init_##k##_##g, mchap_spec_launch_##k##_##g},
#define ROW_SPECP(k, g) {k, g, mchap_specp_init_##k##_##g, mchap_specp_launch_##k##_##g},
const SpecInst SPEC_INSTS[] = {SPEC_LIST(ROW_SPEC)};
const SpecInst SPECP_INSTS[] = {SPECP_LIST(ROW_SPECP)};
#define ROW_SPECS(k, g) {k, g, mchap_specs_init_##k##_##g, mchap_specs_launch_##k##_##g},
const SpecInst SPECS_INSTS[] = {SPECS_LIST(ROW_SPECS)};
#define ROW_SPECD(k, g) {k, g, mchap_specd_init_##k##_##g, mchap_specd_launch_##k##_##g},
const SpecInst SPECD_INSTS[] = {SPECS_LIST(ROW_SPECD)};
const SpecInst *find_inst(const SpecInst *tab, size_t n, int K, int G) {
  for (size_t i = 0; i < n; i++)
    if (tab[i].K == K && tab[i].G == G) return &tab[i];
  return nullptr;
}
#define FIND_SPEC(K, G) find_inst(SPEC_INSTS, sizeof(SPEC_INSTS) / sizeof(SPEC_INSTS[0]), K, G)
#define FIND_SPECP(K, G) find_inst(SPECP_INSTS, sizeof(SPECP_INSTS) / sizeof(SPECP_INSTS[0]), K, G)
#define FIND_SPECS(K, G) find_inst(SPECS_INSTS, sizeof(SPECS_INSTS) / sizeof(SPECS_INSTS[0]), K, G)
#define FIND_SPECD(K, G) find_inst(SPECD_INSTS, sizeof(SPECD_INSTS) / sizeof(SPECD_INSTS[0]), K, G)
// ... and their launchers with decision contexts, by ploidy - 2: {plain, side by side, deep}
#define ROW_SPECC(k) {mchap_specp_launchc_##k##_64, mchap_specs_launchc_##k##_64, mchap_specd_launchc_##k##_64},
const simt_launch_fn SPECC_LAUNCH[7][3] = {ROW_SPECC(2) ROW_SPECC(3) ROW_SPECC(4) ROW_SPECC(5) ROW_SPECC(6) ROW_SPECC(7) ROW_SPECC(8)};

int ensure_init() {
  int dev = 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(MCHAP_ERR_NO_DEVICE, "no HIP device visible: the MCHap kernels need an MI355X (gfx950)");
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (dev < 64 && g_init_done[dev]) return MCHAP_OK;
  double ln[260], ln_inv[260];
  for (int i = 0; i < 260; i++) {
    ln[i] = std::log((double)i);
    ln_inv[i] = std::log(1.0 / (double)i);
  }
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln), ln, sizeof(ln)));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln_inv), ln_inv, sizeof(ln_inv)));
  for (const SpecInst &i : SPEC_INSTS)
    if (i.init(ln, ln_inv) != 0) return fail(MCHAP_ERR_HIP, "constant tables of the speculative sampler <%d, %d>", i.K, i.G);
  for (const SpecInst &i : SPECP_INSTS)
    if (i.init(ln, ln_inv) != 0) return fail(MCHAP_ERR_HIP, "constant tables of the phased sampler <%d, %d>", i.K, i.G);
  for (const SpecInst &i : SPECS_INSTS)
    if (i.init(ln, ln_inv) != 0) return fail(MCHAP_ERR_HIP, "constant tables of the phased sampler <%d, %d> (side by side)", i.K, i.G);
  for (const SpecInst &i : SPECD_INSTS)
    if (i.init(ln, ln_inv) != 0) return fail(MCHAP_ERR_HIP, "constant tables of the phased sampler <%d, %d> (deep)", i.K, i.G);
  {
#define ROW_SIMT_INIT(k) mchap_simt_init_##k,
#define ROW_V1_INIT(r) mchap_v1_init_##r,
#define ROW_LANE_INIT(k) mchap_lane_init_##k,
#define ROW_FILL_INIT(k) mchap_fill_init_##k,
#define ROW_FILLW_INIT(k) mchap_fillw_init_##k,
    const init_fn inits[] = {SIMT_LIST(ROW_SIMT_INIT) mchap_simt_init_w, FILLW_LIST(ROW_FILLW_INIT)
#ifdef MCHAP_TEST_KERNELS
                                 FILL_LIST(ROW_FILL_INIT) V1_LIST(ROW_V1_INIT) LANE_LIST(ROW_LANE_INIT)
#endif
    };
    for (init_fn f : inits)
      if (f(ln, ln_inv) != 0) return fail(MCHAP_ERR_HIP, "constant tables of a sampler object");
  }
  if (dev < 64) g_init_done[dev] = true;
  return MCHAP_OK;
}

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  template <class T>
  T *as() const { return reinterpret_cast<T *>(p); }
};

// A host-pointer entry point works on a stream of its own (non-blocking: it neither waits for nor stalls the caller's
// other streams) and synchronises that stream only.
struct HostCall {
  hipStream_t stream = nullptr;
  int open() {
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    return MCHAP_OK;
  }
  ~HostCall() {
    if (stream) (void)hipStreamDestroy(stream);
  }
  int up(void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
    return MCHAP_OK;
  }
  int down(void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
    return MCHAP_OK;
  }
  int sync() {
    HIP_TRY(hipStreamSynchronize(stream));
    return MCHAP_OK;
  }
};
#define MCHAP_TRY(expr)       \
  do {                        \
    const int rc_ = (expr);   \
    if (rc_) return rc_;      \
  } while (0)

size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
// one device allocation for a host-pointer call: pieces handed out 256-byte aligned
struct DevArena {
  DevBuf buf;
  size_t cap = 0, off = 0;
  int reserve(size_t bytes) {
    cap = bytes + 4096;
    if (hipMalloc(&buf.p, cap) != hipSuccess) return fail(MCHAP_ERR_HIP, "hipMalloc of %zu bytes", cap);
    return MCHAP_OK;
  }
  template <class T>
  T *take(size_t n) {
    T *p = reinterpret_cast<T *>(reinterpret_cast<unsigned char *>(buf.p) + off);
    off += up256(n * sizeof(T));
    return off <= cap ? p : nullptr;
  }
};

// bytes per lane and row of the coded table: 1, 2, or a multiple of 4 (the sampler loads 1 / 2 / 4 bytes at a time)
int code_stride(int rpl) { return rpl <= 2 ? rpl : (rpl + 3) & ~3; }

int rpl_for(int max_reads) {
  const int need = (max_reads + 63) / 64;
  for (int r : {1, 2, 4, 8, 16})
    if (need <= r) return r;
  return -1;
}

int validate_cfg(const mchap_denovo_cfg *cfg) {
  if (!cfg) return fail(MCHAP_ERR_BAD_ARG, "cfg is NULL");
  if (cfg->steps < 1 || cfg->chains < 1) return fail(MCHAP_ERR_BAD_ARG, "steps and chains must be >= 1");
  if (cfg->chains > 65535) return fail(MCHAP_ERR_LIMIT, "chains > 65535");
  if (cfg->n_temps < 1 || cfg->n_temps > MCHAP_MAX_TEMPS) return fail(MCHAP_ERR_LIMIT, "n_temps out of range");
  // assemble/mcmc.py:224-226
  for (int t = 1; t < cfg->n_temps; t++)
    if (cfg->temperatures[t] < cfg->temperatures[t - 1]) return fail(MCHAP_ERR_BAD_ARG, "temperatures must be ascending");
  if (cfg->temperatures[0] < 0.0) return fail(MCHAP_ERR_BAD_ARG, "temperatures must be >= 0");
  if (cfg->temperatures[cfg->n_temps - 1] != 1.0) return fail(MCHAP_ERR_BAD_ARG, "last temperature must be 1.0");
  if (cfg->n_intervals == 0 && !cfg->break_table) return fail(MCHAP_ERR_BAD_ARG, "break_table required when n_intervals is None");
#ifndef MCHAP_TEST_KERNELS
  if (cfg->kernel == 1 || cfg->kernel == 4)
    return fail(MCHAP_ERR_BAD_ARG, "kernel %d is only built into libmchap_hip_test.so (make test-kernels)", cfg->kernel);
#endif
  if (cfg->kernel < 0 || cfg->kernel > 5) return fail(MCHAP_ERR_BAD_ARG, "kernel %d: not one of 0..5", cfg->kernel);
  return MCHAP_OK;
}

// The knobs of mchap_denovo_tuning with their defaults filled in.
struct Tune {
  int cache_slots = 1024, flags = 0, spec_group = 0, pipe_first = 0, pipe_resume = 8, pipe_rounds = 2, pipe_max = 64, pipe_parts = 8;
  int pipe_group = 64;
  size_t prep_lds_limit = 8 * 1024;
  int pipe_stop = 0;
  bool cache_auto = true;  // cache_slots not named by the caller: cache_slots_of() sizes the tables by the batch
};
Tune tune_of(const mchap_denovo_cfg *cfg) {
  Tune t;
  const mchap_denovo_tuning *u = cfg->tuning;
  if (!u) return t;
  if (u->cache_slots >= 64 && u->cache_slots <= 65536 && (u->cache_slots & (u->cache_slots - 1)) == 0) {
    t.cache_slots = u->cache_slots;
    t.cache_auto = false;
  }
  t.flags = u->flags;
  if (u->spec_group == 16 || u->spec_group == 32 || u->spec_group == 64) t.spec_group = u->spec_group;
  if (u->pipe_first > 0) t.pipe_first = u->pipe_first;
  if (u->pipe_resume > 0) t.pipe_resume = u->pipe_resume;
  if (u->pipe_rounds > 0) t.pipe_rounds = u->pipe_rounds - 1;
  if (u->pipe_max > 0) t.pipe_max = u->pipe_max;
  if (u->pipe_parts > 0 && u->pipe_parts <= 64) t.pipe_parts = u->pipe_parts;
  if (u->prep_lds_limit > 0) t.prep_lds_limit = (size_t)u->prep_lds_limit;
  t.pipe_stop = u->pipe_stop;
#ifdef MCHAP_TEST_KERNELS
  if (u->reserved[0] == 16 || u->reserved[0] == 32) t.pipe_group = u->reserved[0];  // narrower phased groups: test library only
#endif
  return t;
}

// The break table of a call lives at the head of the CALLER's workspace (no library-owned device state: two fits
// on different streams, threads or devices never share a buffer).
size_t break_table_bytes(const mchap_denovo_cfg *cfg) {
  const size_t mp = cfg->max_pos > 0 ? (size_t)cfg->max_pos : 1;
  return ((mp + 1) * mp * sizeof(double) + 255) & ~(size_t)255;
}

struct BatchDims {
  int max_reads = 1, max_pos = 1, max_allele = 1, max_ploidy = 1, max_ma = 1, max_ugens = 1;
  int min_reads = 1 << 30;  // the shallowest unit (units of at most 64 reads select the side-by-side instantiation)
  int uniform_ploidy = -1;  // the ploidy shared by all units, 0 if mixed
};

bool use_simt(const mchap_denovo_cfg *cfg);
int batch_dims(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_host, BatchDims &B) {
  for (int u = 0; u < n_units; u++) {
    const mchap_unit &U = units_host[u];
    const int kmax = use_simt(cfg) ? MCHAP_MAX_PLOIDY_DENOVO : MCHAP_MAX_PLOIDY;
    if (U.ploidy < 1 || U.ploidy > kmax) return fail(MCHAP_ERR_LIMIT, "unit %d: ploidy %d not in 1..%d", u, U.ploidy, kmax);
    if (U.max_allele < 1 || U.max_allele > MCHAP_MAX_ALLELE) return fail(MCHAP_ERR_LIMIT, "unit %d: max_allele %d not in 1..%d", u, U.max_allele, MCHAP_MAX_ALLELE);
    if (U.n_reads < 1 || U.n_reads > MCHAP_MAX_READS) return fail(MCHAP_ERR_LIMIT, "unit %d: n_reads %d not in 1..%d (zero reads are mocked by the caller as one NaN read)", u, U.n_reads, MCHAP_MAX_READS);
    if (U.n_pos < 1 || U.n_pos > MCHAP_MAX_POS) return fail(MCHAP_ERR_LIMIT, "unit %d: n_pos %d not in 1..%d", u, U.n_pos, MCHAP_MAX_POS);
    if (cfg->n_intervals == 0 && U.n_pos > cfg->max_pos) return fail(MCHAP_ERR_BAD_ARG, "unit %d: n_pos exceeds break_table", u);
    B.max_reads = std::max(B.max_reads, U.n_reads);
    B.min_reads = std::min(B.min_reads, U.n_reads);
    B.max_pos = std::max(B.max_pos, U.n_pos);
    B.max_allele = std::max(B.max_allele, U.max_allele);
    B.max_ploidy = std::max(B.max_ploidy, U.ploidy);
    B.uniform_ploidy = (B.uniform_ploidy < 0 || B.uniform_ploidy == U.ploidy) ? U.ploidy : 0;
    B.max_ma = std::max(B.max_ma, U.n_pos * U.max_allele);
    B.max_ugens = std::max(B.max_ugens, mchap::snv_genotypes(U.max_allele, U.ploidy));
  }
  return MCHAP_OK;
}

// lanes per chain for the speculative sampler: every option of an interval step (<= K(K-1)) and half the
// sub-steps of a mutation step (K * n_pos) must fit -- a third of them with one chain per wavefront; 0 if the
// shape is not supported by it
int spec_group(const Tune &T, int K, int max_pos) {
  if (K < 2 || K > 8) return 0;
  const int n = K * max_pos;
  int g = T.spec_group ? T.spec_group : 16;
  while (g < 64 && (g < K * (K - 1) || 2 * g < n)) g *= 2;
  const int slots = (g == 64) ? 3 : 2;   // sub-steps per lane the instantiation supports (denovo_spec_kernel.hpp: NS)
  if (g < K * (K - 1) || slots * g < n) return 0;
  if ((K == 5 || K == 6) && g < 32) g = 32;  // instantiated group sizes: 2..4: 16/32/64, 5..6: 32/64, 7..8: 64
  if (K >= 7) g = 64;
  return g;
}

// The steady-state sampler (kernel 4, test library): any ploidy 1..8 with a single temperature
bool lane_supported(int K, int max_pos, int n_temps) {
#ifdef MCHAP_TEST_KERNELS
  if (n_temps != 1 || K < 1 || K > 8) return false;
  const int slots = (K == 8) ? 3 : 2;
  return K * max_pos <= slots * 64 && K * (K - 1) <= 64;
#else
  return false;
#endif
}

// The phased sampler (kernel 5): one chain per wavefront at every ploidy -- it spends its time in likelihood
// evaluations, which a wave serves one after the other whatever the group size; wider groups need fewer rounds per table
// completion and leave a shorter tail (MI355X, 10 000 loci of config #2's shape: K = 4: 16.4 ms with 16 lanes, 14.5 with
// 32, 13.7 with 64; K = 6: 87 -> 67 ms).  Needs the interval memo (single temperature, tables in LDS) and a mutation
// step whose draws fit the staged window.
int pipe_group(const Tune &T, int K) {
  if (K < 2 || K > 8) return 0;
  if (T.pipe_group != 64 && FIND_SPECP(K, T.pipe_group)) return T.pipe_group;
  return 64;
}
bool pipe_supported(const mchap_denovo_cfg *cfg, const Tune &T, int K, int max_pos) {
  if ((cfg->kernel != 5 && cfg->kernel != 0) || cfg->n_temps != 1 || K < 2 || K > 8) return false;
  if (T.spec_group || (T.flags & 3)) return false;  // measurement settings of kernel 3 / memos switched off
  const int g = spec_group(T, K, max_pos);
  if (g == 0 || g > pipe_group(T, K)) return false;
  if (mchap::spec_memo_bytes(max_pos, 1, g) == 0) return false;
  return K * max_pos <= mchap::spec_draws(K, max_pos);
}

// Read chunks (of 64) per unit for the prepare pass and the sampler behind it.  The speculative sampler takes any
// count up to 8, then 12, 16, 24, 32, 48, 64 (4096 reads: the prepare pass is instantiated per count); the
// lanes-over-chains kernel and kernel 1 are instantiated for powers of two up to 16 (1024 reads).
int simt_rpl(const mchap_denovo_cfg *cfg, const Tune &T, int uniform_ploidy, int max_pos, int max_reads) {
  const bool lane = cfg->kernel == 4 && uniform_ploidy > 0 && lane_supported(uniform_ploidy, max_pos, cfg->n_temps);
  const bool spec = lane || (cfg->kernel != 1 && cfg->kernel != 2 && uniform_ploidy > 0 && spec_group(T, uniform_ploidy, max_pos) != 0);
  if (!spec) return rpl_for(max_reads);
  const int need = (max_reads + 63) / 64;
  if (need <= 8) return need < 1 ? 1 : need;
  for (int r : {12, 16, 24, 32, 48, 64})  // (the sampler takes its chunks four at a time: any count works there)
    if (need <= r) return r;
  return -1;
}

bool use_simt(const mchap_denovo_cfg *cfg) { return cfg->kernel != 1; }

// Which sampler a batch runs on (mchap_denovo_cfg.kernel, the shapes, the ploidies present).
enum SamplerKind { SAMPLER_V1 = 1, SAMPLER_SIMT = 2, SAMPLER_SPEC = 3, SAMPLER_LANE = 4, SAMPLER_PIPE = 5 };
struct Plan {
  int kind = 0, K = 0, G = 0, rpl = 0;
  bool wide = false;  // 128-bit haplotype words: denovo_simt_kernel<0, u128>
};
// A batch whose units may need more than 64 bits of sampled alleles per haplotype, or hold more than 62 SNVs (the 64-bit
// samplers keep a unit's interval end points in one word): the lanes-over-chains sampler with 128-bit words takes it.
// The same kernel takes ploidies 9 to 15 (packs of sixteen nibbles).
bool wide_batch(const BatchDims &B) {
  return mchap::allele_bits(B.max_allele) * B.max_pos > 64 || B.max_pos > 62 || B.max_ploidy > MCHAP_MAX_PLOIDY;
}
int plan_sampler(const mchap_denovo_cfg *cfg, const Tune &T, const BatchDims &B, Plan &pl) {
  pl.K = B.uniform_ploidy;
  if (use_simt(cfg) && wide_batch(B)) {
    pl.rpl = rpl_for(B.max_reads);
    if (pl.rpl < 0) return fail(MCHAP_ERR_LIMIT, "n_reads %d: the sampler for targets wider than 64 bits takes 1..1024 reads", B.max_reads);
    pl.kind = SAMPLER_SIMT;
    pl.K = 0;
    pl.wide = true;
    return MCHAP_OK;
  }
  pl.rpl = use_simt(cfg) ? simt_rpl(cfg, T, B.uniform_ploidy, B.max_pos, B.max_reads) : rpl_for(B.max_reads);
  if (pl.rpl < 0)
    return fail(MCHAP_ERR_LIMIT, "n_reads %d: the sampler kernel for this batch takes 1..%d reads", B.max_reads,
                (use_simt(cfg) && B.uniform_ploidy > 0 && spec_group(T, B.uniform_ploidy, B.max_pos)) ? MCHAP_MAX_READS : 1024);
  if (!use_simt(cfg)) {
    pl.kind = SAMPLER_V1;
  } else if (cfg->kernel == 4 && B.uniform_ploidy > 0 && lane_supported(B.uniform_ploidy, B.max_pos, cfg->n_temps)) {
    pl.kind = SAMPLER_LANE;
  } else if (pipe_supported(cfg, T, B.uniform_ploidy, B.max_pos)) {
    pl.kind = SAMPLER_PIPE;
    pl.G = pipe_group(T, pl.K);
  } else if (cfg->kernel != 2 && B.uniform_ploidy > 0 && spec_group(T, B.uniform_ploidy, B.max_pos)) {
    pl.kind = SAMPLER_SPEC;
    pl.G = spec_group(T, B.uniform_ploidy, B.max_pos);
    // a temperature ladder runs its replicas side by side, one wavefront each: that form is one chain per wavefront
    if (cfg->n_temps > 1 && cfg->n_temps <= mchap::SPEC_TW_MAX && !T.spec_group && !(T.flags & 16384)) pl.G = 64;
  } else {
    pl.kind = SAMPLER_SIMT;
    pl.K = (pl.K == 2 || pl.K == 4 || pl.K == 6 || pl.K == 8) ? pl.K : 0;  // (specialised for even ploidies, else generic)
  }
  return MCHAP_OK;
}
void plan_name(const Plan &pl, char *out, size_t n) {
  switch (pl.kind) {
    case SAMPLER_V1: snprintf(out, n, "denovo_mcmc_kernel<%d>", pl.rpl); break;
    case SAMPLER_SIMT:
      if (pl.wide) snprintf(out, n, "denovo_simt_kernel<0, u128>");
      else snprintf(out, n, "denovo_simt_kernel<%d>", pl.K);
      break;
    case SAMPLER_SPEC: snprintf(out, n, "denovo_spec_kernel<%d, %d>", pl.K, pl.G); break;
    case SAMPLER_LANE: snprintf(out, n, "denovo_settle_kernel<%d> + denovo_steady_kernel<%d>", pl.K, pl.K); break;
    default: snprintf(out, n, "denovo_spec_kernel<%d, %d, phased> + denovo_coast_kernel", pl.K, pl.G); break;
  }
}

struct SimtCarve {
  size_t cache = 0, ckeys = 0, rt = 0, cntw = 0, codes = 0, dict = 0, meta_i = 0, meta_f = 0, lane_state = 0, lane_memo = 0, total = 0;
  size_t pipe_state = 0, pipe_memo = 0, pipe_lists = 0, pipe_counts = 0;
  int key_words = 0;
  size_t gbp = 0;
  bool has_gbp = false;
  size_t ctx = 0;  // decision contexts per genotype of the phased sampler's resumed chains: the LAST piece (ctx_n slots per chain)
  int ctx_n = 0;
};
constexpr int PIPE_MAX_ROUNDS = 6;  // resume rounds of the phased sampler (counters in the workspace)

// Entries per chain of the likelihood cache.  The reference's cache is unbounded (assemble/mcmc.py: a dictionary per chain); a
// chain that never settles -- phase-ambiguous samples of real pileups: docs/example locus015 requests ~100 likelihoods per step
// and 95 % of them are genotypes it has seen -- re-evaluates what a small table has dropped: at 1024 entries three quarters of
// such a chain's time were evaluations (24 per step against the 5 an unbounded table leaves; profiles/r04d_stats_example.txt).
// Unless the caller names a size, the tables therefore take what CACHE_BUDGET bytes over the batch's chains allow, between 1024
// and 65536 entries: 880 chains of wide genotypes get 65536 each, 20 000 chains of configs[1] 8192.
constexpr size_t CACHE_BUDGET = (size_t)4 << 30;
// (a budget never takes more than 1/frac of the memory the device has free right now: several passes in flight on a shared or
// partly used device each ask for theirs; the tables only save work, any size gives the same results)
size_t budget_on_device(size_t cap, size_t frac) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / frac < cap) return free_b / frac;
  return cap;
}
int cache_slots_of(const mchap_denovo_cfg *cfg, const Tune &T, const Plan &pl, int n_units, const BatchDims &B) {
  if (!cfg->llk_cache) return 0;
  if (!T.cache_auto) return T.cache_slots;
  const size_t ncc = (size_t)n_units * cfg->chains * (size_t)(pl.kind == SAMPLER_SPEC ? cfg->n_temps : 1);
  const bool keyed = B.max_ploidy * mchap::allele_bits(B.max_allele) * B.max_pos > 63;
  const size_t entry = 16 + (keyed ? (size_t)B.max_ploidy * (pl.wide ? 2 : 1) * 8 : 0);
  const size_t budget = budget_on_device(CACHE_BUDGET, 8);
  int slots = 1024;
  while (slots > 32 && ncc * (size_t)slots * entry > budget) slots >>= 1;  // (a nearly full device: below the usual floor)
  while (slots < 65536 && ncc * (size_t)(2 * slots) * entry <= budget) slots *= 2;
  return slots;
}

// Decision contexts per genotype (denovo_spec_kernel.hpp): slots per chain -- what CTX_BUDGET bytes (and an eighth of the
// device's free memory) over the batch's chains allow, at most 64; fewer than 8 are not worth the look-ups.  The piece sits at the
// end of the workspace and a fit takes as many slots as the workspace it was handed holds.
constexpr size_t CTX_BUDGET = (size_t)6 << 30;
bool ctx_shape(const Plan &pl, const BatchDims &B, const Tune &T) {
  return pl.kind == SAMPLER_PIPE && pl.G == 64 && !(T.flags & 524288) && B.max_ploidy * B.max_pos <= 192;
}
size_t ctx_slot_bytes(const Plan &pl, const BatchDims &B) { return (size_t)mchap::spec_ctx_words(pl.K, B.max_pos) * 8; }
int ctx_slots_for(size_t bytes, size_t n_chains, size_t slot_bytes) {
  const size_t n = bytes / (n_chains * slot_bytes);
  return n >= (size_t)mchap::SPEC_CTX_MAX ? mchap::SPEC_CTX_MAX : (n < 8 ? 0 : (int)n);
}
int ctx_slots_of(const mchap_denovo_cfg *cfg, const Tune &T, const Plan &pl, int n_units, const BatchDims &B) {
  if (!ctx_shape(pl, B, T)) return 0;
  return ctx_slots_for(budget_on_device(CTX_BUDGET, 8), (size_t)n_units * cfg->chains, ctx_slot_bytes(pl, B));
}

SimtCarve simt_carve(const mchap_denovo_cfg *cfg, const Plan &pl, int n_units, const BatchDims &B, int rpad, int cache_slots, int ctx_n = 0) {
  SimtCarve c;
  size_t o = 0;
  const size_t nc = (size_t)n_units * cfg->chains;
  // (a temperature ladder on the speculative sampler: a likelihood cache per REPLICA -- they run side by side, one wavefront each)
  const size_t ncc = nc * (size_t)(pl.kind == SAMPLER_SPEC ? cfg->n_temps : 1);
  c.cache = o; o += up256(ncc * cache_slots * 16);
  // genotypes of more than 63 bits: their words beside the (hashed) tags, so that a cache hit is always exact
  if (cache_slots > 0 && B.max_ploidy * mchap::allele_bits(B.max_allele) * B.max_pos > 63) {
    c.key_words = B.max_ploidy * (pl.wide ? 2 : 1);  // (uint64 words per entry: K haplotype words of 64 or 128 bits)
    c.ckeys = o; o += up256(ncc * cache_slots * c.key_words * 8);
  }
  c.rt = o; o += up256((size_t)n_units * B.max_ma * rpad * 8);
  c.cntw = o; o += up256((size_t)n_units * rpad * 8);
  c.codes = o; o += up256((size_t)n_units * B.max_ma * 64 * code_stride(rpad / 64));
  c.dict = o; o += up256((size_t)n_units * mchap::DICT_MAX * 8);
  c.meta_i = o; o += up256((size_t)n_units * mchap::meta_i_stride(B.max_pos) * 4);
  c.meta_f = o; o += up256((size_t)n_units * mchap::meta_f_stride(B.max_ploidy, B.max_pos, B.max_allele) * 8);
#ifdef MCHAP_TEST_KERNELS
  if (pl.kind == SAMPLER_LANE) {  // hand-over records of the steady-state pipeline
    c.lane_state = o; o += up256(nc * sizeof(mchap::LaneState));
    c.lane_memo = o; o += up256(nc * 2 * mchap::spec_memo_entries(B.max_pos) * 4);
  }
#endif
  // deep units (more than four chunks of 64 reads): the haplotype products of every chain's current genotype for the
  // chunks beyond the fourth (denovo_spec_kernel.hpp BaseProductsG), [chain][ploidy][rpad] doubles
  if ((pl.kind == SAMPLER_PIPE || pl.kind == SAMPLER_SPEC) && rpad > 4 * 64) {
    c.gbp = o; o += up256(nc * (size_t)(pl.kind == SAMPLER_SPEC ? cfg->n_temps : 1) * B.max_ploidy * rpad * 8);  // (a set per replica)
    c.has_gbp = true;
  }
  if (pl.kind == SAMPLER_PIPE) {  // hand-over records of the phased sampler
    c.pipe_state = o; o += up256(nc * sizeof(mchap::PipeState));
    c.pipe_memo = o; o += up256(nc * 2 * mchap::spec_memo_entries(B.max_pos) * 8);
    c.pipe_lists = o; o += up256(nc * 4) * 2;
    c.pipe_counts = o; o += up256((PIPE_MAX_ROUNDS + 2) * 4);
    c.ctx = o;
    c.ctx_n = ctx_n;
    o += up256(nc * (size_t)ctx_n * ctx_slot_bytes(pl, B));
  }
  c.total = o;
  return c;
}

template <int RPL>
int launch_prepare(const mchap::SimtParams &P, int n_units, size_t lds_prep, hipStream_t stream) {
  auto kp = mchap::denovo_prepare_kernel<RPL>;
  if (lds_prep > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep));
  hipLaunchKernelGGL(kp, dim3(n_units), dim3(64), lds_prep, stream, P);
  HIP_TRY(hipGetLastError());
  return MCHAP_OK;
}

int launch_simt(int KT, const mchap::SimtParams &P, int n_units, int chains, size_t lds_simt, void *timer, hipStream_t stream,
                bool wide = false) {
  simt_launch_fn launch = wide ? mchap_simt_launch_w :
      KT == 2 ? mchap_simt_launch_2 : KT == 4 ? mchap_simt_launch_4 : KT == 6 ? mchap_simt_launch_6 : KT == 8 ? mchap_simt_launch_8 : mchap_simt_launch_0;
  const long long n_chains = (long long)n_units * chains;
  SamplerTimer tm(timer, stream);
  const int e = launch(&P, (unsigned)((n_chains + 63) / 64), lds_simt, stream);
  if (e != 0) return fail(MCHAP_ERR_HIP, "launch of denovo_simt_kernel<%d>: %s", KT, hipGetErrorString((hipError_t)e));
  return MCHAP_OK;
}

// One chain per wave (G = 64): the haplotype products of the chain's current genotype live in an LDS cache behind the
// sampler's LDS (8 KB at K = 4, 16 KB at K = 8) instead of 64-128 VGPRs
int bp_cache_fits(const Tune &T, int G, size_t lds, int K) {
  if (G != 64 || (T.flags & 16)) return 0;
  return ((lds + 15) & ~(size_t)15) + mchap::spec_bp_cache_bytes(K) <= 160 * 1024 ? 1 : 0;
}

#define DECL_SPEC_TW(k) extern "C" int mchap_spec_launchtw_##k##_64(const mchap::SimtParams *, unsigned, size_t, hipStream_t);
DECL_SPEC_TW(2) DECL_SPEC_TW(3) DECL_SPEC_TW(4) DECL_SPEC_TW(5) DECL_SPEC_TW(6) DECL_SPEC_TW(7) DECL_SPEC_TW(8)

int launch_spec(const Tune &T, int K, int G, const mchap::SimtParams &P, int n_units, int chains, int n_temps, void *timer, hipStream_t stream) {
  const SpecInst *inst = FIND_SPEC(K, G);
  if (!inst) return fail(MCHAP_ERR_LIMIT, "speculative sampler: no instantiation for ploidy %d with %d lanes per chain", K, G);
  size_t lds = mchap::spec_lds_bytes(K, P.max_pos, P.max_allele, n_temps, G);
  if (lds > 160 * 1024) return fail(MCHAP_ERR_LIMIT, "speculative sampler needs %zu bytes of LDS", lds);
  mchap::SimtParams Q = P;
  Q.bp_cache = bp_cache_fits(T, G, lds, K);
  if (Q.bp_cache) lds = ((lds + 15) & ~(size_t)15) + mchap::spec_bp_cache_bytes(K);
  const long long n_chains = (long long)n_units * chains;
  // A temperature ladder with one chain per wavefront: a workgroup of n_temps wavefronts per chain, one replica each
  // (denovo_spec_kernel<.., TW>; tuning flag 16384: the replicas one after the other on one wavefront, as before round 4)
  if (G == 64 && n_temps > 1 && n_temps <= mchap::SPEC_TW_MAX && !(T.flags & 16384) && K >= 2 && K <= 8) {
    const size_t per_wave = (lds + 63) & ~(size_t)63;
    const size_t lds_tw = mchap::spec_tw_exchange_bytes(K, n_temps) + per_wave * n_temps;
    if (lds_tw <= 160 * 1024) {
      const simt_launch_fn tw[] = {mchap_spec_launchtw_2_64, mchap_spec_launchtw_3_64, mchap_spec_launchtw_4_64, mchap_spec_launchtw_5_64,
                                   mchap_spec_launchtw_6_64, mchap_spec_launchtw_7_64, mchap_spec_launchtw_8_64};
      Q.tw_lds = (int)per_wave;
      SamplerTimer tm(timer, stream);
      const int e = tw[K - 2](&Q, (unsigned)n_chains, lds_tw, stream);
      if (e != 0) return fail(MCHAP_ERR_HIP, "launch of denovo_spec_kernel<%d, 64, replicas side by side>: %s", K, hipGetErrorString((hipError_t)e));
      return MCHAP_OK;
    }
  }
  const int per_wave = 64 / G;
  SamplerTimer tm(timer, stream);
  const int e = inst->launch(&Q, (unsigned)((n_chains + per_wave - 1) / per_wave), lds, stream);
  if (e != 0) return fail(MCHAP_ERR_HIP, "launch of denovo_spec_kernel<%d, %d>: %s", K, G, hipGetErrorString((hipError_t)e));
  return MCHAP_OK;
}

#ifdef MCHAP_TEST_KERNELS
int launch_denovo(int rpl, const mchap::DenovoParams &P, int n_units, int chains, size_t lds, void *timer, hipStream_t stream) {
  int (*launch)(const mchap::DenovoParams *, unsigned, unsigned, unsigned, size_t, hipStream_t) =
      rpl == 1 ? mchap_v1_launch_1 : rpl == 2 ? mchap_v1_launch_2 : rpl == 4 ? mchap_v1_launch_4 : rpl == 8 ? mchap_v1_launch_8 : mchap_v1_launch_16;
  const int cpb = chains < mchap::CHAINS_PER_BLOCK ? chains : mchap::CHAINS_PER_BLOCK;
  SamplerTimer tm(timer, stream);
  const int e = launch(&P, (unsigned)n_units, (unsigned)((chains + cpb - 1) / cpb), (unsigned)(64 * cpb), lds, stream);
  if (e != 0) return fail(MCHAP_ERR_HIP, "launch of denovo_mcmc_kernel<%d>: %s", rpl, hipGetErrorString((hipError_t)e));
  return MCHAP_OK;
}

// Lanes per chain of the settling kernel (it runs one wave per SIMD -- its serving code needs the registers -- so a
// wave may use a quarter of a CU's LDS: the unit tables of its chains live there).  -1: does not fit.
int lane_shift(int K, int max_pos, int max_allele, int tab_bytes, int rpad) {
  int s = 4;
  while (s < 6 && mchap::lane_lds_bytes(K, max_pos, max_allele, 1 << s, tab_bytes, rpad) > 40 * 1024) s++;
  if (mchap::lane_lds_bytes(K, max_pos, max_allele, 1 << s, tab_bytes, rpad) > 160 * 1024) return -1;
  return s;
}

int launch_lane(const Tune &T, int K, const mchap::SimtParams &P, int n_units, int chains, void *timer, hipStream_t stream) {
#define ROW_LANE_LAUNCH(k) mchap_lane_launch_##k,
  int (*const launches[])(const mchap::SimtParams *, int, int, unsigned, size_t, hipStream_t) = {LANE_LIST(ROW_LANE_LAUNCH)};
  auto launch = launches[K - 1];
  const long long n_chains = (long long)n_units * chains;
  const int tab_bytes = P.max_ma * 64 * P.cstride;
  const int lsh = lane_shift(K, P.max_pos, P.max_allele, tab_bytes, P.d.rpad);
  if (lsh < 0) return fail(MCHAP_ERR_LIMIT, "steady-state sampler: the unit tables do not fit the LDS");
  const size_t lds = mchap::lane_lds_bytes(K, P.max_pos, P.max_allele, 1 << lsh, tab_bytes, P.d.rpad);
  const int per_wave = 64 >> lsh;
  // the steady kernel's own geometry: it is lean, so more lanes per chain (more waves) only help the SIMDs' issue rate
  int fsh = 0;
  while (fsh < 4 && n_chains * (1ll << fsh) / 64 < 4096) fsh++;
  const size_t lds_f = mchap::steady_lds_bytes(P.max_pos, 1 << fsh);
  const int per_wave_f = 64 >> fsh;
  const int rounds = T.pipe_rounds;
  SamplerTimer tm(timer, stream);
  const unsigned grid_w = (unsigned)((n_chains + per_wave - 1) / per_wave), grid_f = (unsigned)((n_chains + per_wave_f - 1) / per_wave_f);
  // settle (park the chains whose thresholds are complete) -> steady -> settle the ones handed back -> ... -> finish
  int e = launch(&P, lsh, rounds > 0 ? mchap::LANE_MODE_PARK : 0, grid_w, lds, stream);
  for (int r = 0; r < rounds && e == 0; r++) {
    e = launch(&P, fsh, -1, grid_f, lds_f, stream);
    if (e == 0) e = launch(&P, lsh, mchap::LANE_MODE_RESUME | (r + 1 < rounds ? mchap::LANE_MODE_PARK : 0), grid_w, lds, stream);
  }
  if (e != 0) return fail(MCHAP_ERR_HIP, "launch of the settle / steady pipeline <%d>: %s", K, hipGetErrorString((hipError_t)e));
  return MCHAP_OK;
}
#endif  // MCHAP_TEST_KERNELS

// The phased sampler (kernel 5): denovo_spec_kernel<K, G, true> for the first steps, denovo_coast_kernel for the
// chains' long no-move stretches, denovo_spec_kernel again for the chains handed back.
int launch_pipe(const Tune &T, int K, int G, mchap::SimtParams P, int n_units, int chains, int32_t *lists, int32_t *counts, void *timer,
                hipStream_t stream, bool shallow_units, int max_reads) {
  // deep units (product rows in the workspace) or more than 128 (haplotype, position) pairs: the "deep" instantiation; else a
  // batch with a unit of at most 64 reads: the one that evaluates such a unit's requests side by side; else the plain one
  const SpecInst *inst = nullptr;
  int variant = 0;  // 0 plain, 1 side by side, 2 deep
  if (G == 64 && (P.gbp != nullptr || (K * P.max_pos > 128 && !(T.flags & 512)))) inst = FIND_SPECD(K, G), variant = 2;
  else if (shallow_units && G == 64 && !(T.flags & 256)) inst = FIND_SPECS(K, G), variant = 1;
  if (!inst) inst = FIND_SPECP(K, G), variant = 0;
  if (!inst) return fail(MCHAP_ERR_LIMIT, "phased sampler: no instantiation for ploidy %d with %d lanes per chain", K, G);
  simt_launch_fn launch = inst->launch;
  size_t lds = mchap::spec_lds_bytes(K, P.max_pos, P.max_allele, 1, G);
  if (lds > 160 * 1024) return fail(MCHAP_ERR_LIMIT, "speculative sampler needs %zu bytes of LDS", lds);
  P.bp_cache = bp_cache_fits(T, G, lds, K);
  if (P.bp_cache) lds = ((lds + 15) & ~(size_t)15) + mchap::spec_bp_cache_bytes(K);
  // The chain's likelihood cache in LDS (denovo_spec_kernel.hpp spec_eval): exact tags only (packed genotype of at most 63
  // bits), and only where its 4 KB do not cost the launch a resident wavefront per CU (two per SIMD: eight per CU at most).
  // Tuning flag 8192: never (the table in the workspace is probed, as before round 4: same traces).
  auto per_cu = [](size_t b) { const size_t w = (160 * 1024) / b; return w > 8 ? (size_t)8 : w; };
  const size_t lds_nolc = lds;
  const bool lc_shape = P.bp_cache && P.d.cache_slots > 0 && !(T.flags & 8192) && K * mchap::allele_bits(P.max_allele) * P.max_pos <= 63;
  if (lc_shape) {
    const size_t with = ((lds + 15) & ~(size_t)15) + mchap::spec_lc_bytes();
    if (with <= 160 * 1024 && per_cu(with) >= per_cu(lds)) {
      P.bp_cache |= 2;
      lds = with;
    }
  }
  // The launches over handed-back chains keep decision contexts per genotype (denovo_spec_kernel<.., CTX>): its own instantiation
  // and LDS size, so that the first launch -- every chain, three steps -- stays what it was.  The first launch gets no region.
  // (a unit with more than two alleles at a position runs without them: the kernel's own check)
  simt_launch_fn launch_c = launch;
  size_t lds_c = lds;
  int bp_cache_c = P.bp_cache;
  uint64_t *ctx_region = nullptr;
  int ctx_n = 0;
  if (G == 64 && P.ctx != nullptr && P.ctx_n > 0 && K >= 2 && K <= 8) {
    // the front cache beside the contexts: whole, halved or not at all -- the largest that keeps the launch's wavefronts per CU
    // (the contexts answer most of what the front cache did); if nothing does, as in the other launches
    const size_t cx = mchap::spec_ctx_lds_bytes(K, P.max_pos);
    size_t with = ((lds + 15) & ~(size_t)15) + cx;
    if ((P.bp_cache & 1) && mchap::spec_ctx_in_bpc(K, P.max_pos, P.max_allele, max_reads)) {
      with = lds;  // (the product cache's free chunk slots hold them: nothing added, the front cache whole)
      bp_cache_c = P.bp_cache | 8;
    } else if (P.bp_cache & 2) {
      for (int mode = 0; mode < 3; mode++) {  // 256 entries, 128, none
        const size_t base = mode == 2 ? lds_nolc : ((lds_nolc + 15) & ~(size_t)15) + mchap::spec_lc_bytes(mode == 0 ? mchap::SPEC_LC_ENTRIES : mchap::SPEC_LC_ENTRIES / 2);
        const size_t t = ((base + 15) & ~(size_t)15) + cx;
        if (per_cu(t) >= per_cu(lds)) {
          with = t;
          bp_cache_c = mode == 0 ? P.bp_cache : (mode == 1 ? (P.bp_cache | 4) : (P.bp_cache & ~2));
          break;
        }
      }
    }
    if (with <= 160 * 1024) {
      launch_c = SPECC_LAUNCH[K - 2][variant];
      lds_c = with;
      ctx_region = P.ctx;
      ctx_n = P.ctx_n;
    } else {
      bp_cache_c = P.bp_cache;
    }
  }
  P.ctx = nullptr;
  P.ctx_n = 0;
  const long long n_chains = (long long)n_units * chains;
  // steps before the first hand-over: a chain handed over before it has settled comes back and has its tables
  // completed a second time, which costs more the more sub-steps and intervals a step has (config #2: 32 sub-steps,
  // best at 3 since the listed coasting launches run four wavefronts per chain -- 1.060 / 1.080 / 1.081 M loci/s at 4 / 3 / 2,
  // one pass at a time 860 / 870 / 854 k --, at 4 before; config #5: 160 sub-steps, 1251 / 912 / 570 / 632 ms at 4 / 8 / 16 / 32)
  const int n_sub = K * P.max_pos;
  const int s0 = T.pipe_first > 0 ? T.pipe_first : (n_sub / 10 < 3 ? 3 : (n_sub / 10 > 32 ? 32 : n_sub / 10));
  const int nr = T.pipe_resume;   // steps a handed-back chain runs before the next hand-over
  const int rounds = T.pipe_rounds < PIPE_MAX_ROUNDS ? T.pipe_rounds : PIPE_MAX_ROUNDS;
  P.pipe_iters_max = T.pipe_max;  // ... extended to while a chain of the wave is unsettled
  P.pipe_parts = T.pipe_parts;    // wavefronts per chain completing tables when chains are few
  const unsigned grid_f = (unsigned)((mchap::PIPE_FILL_SLOTS + 64 / G - 1) / (64 / G));
  const unsigned grid_s = (unsigned)((n_chains + 64 / G - 1) / (64 / G)), grid_c = (unsigned)n_chains;
  const size_t list_stride = up256((size_t)n_chains * 4) / 4;
  HIP_TRY(hipMemsetAsync(counts, 0, (PIPE_MAX_ROUNDS + 2) * 4, stream));
  SamplerTimer tm(timer, stream);
  // every chain: first steps from scratch, complete tables, records
  // Table completion: inside the exporting launch (spread over several wavefronts per chain by a PIPE_FILLONLY launch when
  // the chains are few).  The parity suite's library also carries denovo_fill_kernel -- the same tables from one lane per
  // request -- behind tuning flag 64: bit-identical (tests/test_gpu_fill.py), measured slower, not shipped.
  bool lpr = false;
  size_t lds_fill = 0;
  // The shipped completion (round 4): denovo_fillw_kernel -- its own launch after every exporting launch, a workgroup of four
  // wavefronts per chain, one wavefront per distinct request, 128 VGPRs -- for genotypes that pack into 64 bits (tuning flag 1024:
  // the in-kernel completion instead; the tables are the same bit for bit: tests/test_gpu_fillw.py)
  const bool fillw = G == 64 && !(T.flags & (64 | 1024 | 32)) && mchap::fillw_takes(K, P.max_pos, P.max_allele);
  const int fillw_rows = mchap::fillw_tab_rows(P.max_pos, P.max_allele, P.d.rpad);
  const bool fillw_wide = mchap::fillw_wide(K, P.max_pos, P.max_allele);
  const int fillw_memo = (T.flags & 32768) ? 0 : mchap::fillw_memo_cap(K, P.max_pos, fillw_rows, fillw_wide);  // (flag 32768: no memo across chunks)
  const size_t lds_fillw = mchap::fillw_lds_bytes(K, P.max_pos, fillw_rows, P.d.rpad, fillw_wide, fillw_memo);
  if (fillw) {
    P.fill_lt = fillw_rows;
    // (the instantiation: keyed by changed words / deep units: fillw_inst.hip; bits 8..: entries / 256 of the memo across chunks)
    P.fill_kw = (fillw_wide ? 1 : 0) | (P.gbp != nullptr ? 2 : 0) | ((fillw_memo / 256) << 8);
  }
#define ROW_FILLW_LAUNCH(k) mchap_fillw_launch_##k,
  const simt_launch_fn fillw_launches[] = {FILLW_LIST(ROW_FILLW_LAUNCH)};
#ifdef MCHAP_TEST_KERNELS
  if (T.flags & 64) {
    P.fill_lt = mchap::fill_geometry(K, P.max_pos, P.max_allele, P.d.rpad, &lds_fill);
    P.fill_kw = mchap::fill_key_words(K, P.max_pos, P.max_allele);
    lpr = P.fill_lt > 0;
  }
#define ROW_FILL_LAUNCH(k) mchap_fill_launch_##k,
  const simt_launch_fn fill_launches[] = {FILL_LIST(ROW_FILL_LAUNCH)};
#endif
  const int export_mode = mchap::PIPE_EXPORT | ((lpr || fillw) ? mchap::PIPE_NOFILL : 0);
  P.pipe_list = nullptr;
  P.pipe_count = nullptr;
  P.pipe_iters = s0;
  P.pipe_mode = export_mode;
  int e = launch(&P, grid_s, lds, stream);
  auto fill_launch = [&]() {
#ifdef MCHAP_TEST_KERNELS
    if (lpr) return fill_launches[K - 2](&P, (unsigned)n_chains, lds_fill, stream);
#endif
    if (fillw) return fillw_launches[K - 2](&P, (unsigned)n_chains, lds_fillw, stream);
    if (P.pipe_parts <= 1) return 0;  // (a no-op unless the chains of the list are few: pipe_parts_eff)
    mchap::SimtParams F = P;
    F.pipe_mode = mchap::PIPE_RESUME | mchap::PIPE_FILLONLY;
    return launch(&F, grid_f, lds, stream);
  };
  if (e == 0) e = fill_launch();
  // coast; then rounds of (resume the chains handed back for a few steps, coast again); the rest runs to the end
  P.pipe_out = lists;
  P.pipe_out_count = counts;
  if (e == 0) e = mchap_coast_launch(&P, grid_c, 0, stream);
  for (int r = 0; r <= rounds && e == 0 && T.pipe_stop != 1; r++) {
    P.pipe_list = lists + (size_t)(r & 1) * list_stride;
    P.pipe_count = counts + r;
    const bool last = r == rounds;
    P.pipe_iters = last ? 0 : nr;
    P.pipe_mode = mchap::PIPE_RESUME | (last ? 0 : export_mode);
    {
      mchap::SimtParams R = P;  // (this launch's list, counters and mode; the contexts' region)
      R.ctx = ctx_region;
      R.ctx_n = ctx_n;
      R.bp_cache = bp_cache_c;
      e = launch_c(&R, grid_s, lds_c, stream);
    }
    if (last || e != 0) break;
    e = fill_launch();
    if (e != 0) break;
    P.pipe_out = lists + (size_t)((r + 1) & 1) * list_stride;
    P.pipe_out_count = counts + r + 1;
    if (T.pipe_stop == -(r + 2)) break;  // (debugging: stop before this round's coasting launch)
    // (the chains handed back are few -- 23 of 20 000 at configs[1] --: four wavefronts per chain, 256 steps per sweep; tuning
    // flag 131072: one, as the launch over every chain)
    e = mchap_coast_launch(&P, grid_c, (T.flags & 131072) ? 0 : 1, stream);
    if (T.pipe_stop == r + 2) break;  // (measurement / debugging: stop after this round's coasting launch)
  }
  if (e != 0) return fail(MCHAP_ERR_HIP, "launch of the phased sampler <%d, %d>: %s", K, G, hipGetErrorString((hipError_t)e));
  return MCHAP_OK;
}

}  // namespace

extern "C" {

const char *mchap_version(void) { return "mchap-hip 0.1 (gfx950; restates MCHap v0.11.1 assemble + calling.exact)"; }
const char *mchap_last_error(void) { return g_err; }

int mchap_timer_create(void **timer) {
  if (!timer) return fail(MCHAP_ERR_BAD_ARG, "timer is NULL");
  int rc = ensure_init();
  if (rc) return rc;
  Timer *t = new Timer();
  if (hipEventCreate(&t->e0) != hipSuccess || hipEventCreate(&t->e1) != hipSuccess) {
    delete t;
    return fail(MCHAP_ERR_HIP, "hipEventCreate");
  }
  *timer = t;
  return MCHAP_OK;
}

double mchap_timer_ms(void *timer) {
  Timer *t = reinterpret_cast<Timer *>(timer);
  if (!t || !t->recorded) return -1.0;
  if (hipEventSynchronize(t->e1) != hipSuccess) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, t->e0, t->e1) != hipSuccess) return -1.0;
  return (double)ms;
}

int mchap_timer_destroy(void *timer) {
  Timer *t = reinterpret_cast<Timer *>(timer);
  if (!t) return MCHAP_OK;
  if (t->e0) (void)hipEventDestroy(t->e0);
  if (t->e1) (void)hipEventDestroy(t->e1);
  delete t;
  return MCHAP_OK;
}

int mchap_denovo_sampler_name(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_host, char *out, int out_len) {
  int rc = validate_cfg(cfg);
  if (rc) return rc;
  if (!out || out_len < 1 || !units_host || n_units < 1) return fail(MCHAP_ERR_BAD_ARG, "NULL buffer");
  BatchDims B;
  rc = batch_dims(cfg, n_units, units_host, B);
  if (rc) return rc;
  Plan pl;
  rc = plan_sampler(cfg, tune_of(cfg), B, pl);
  if (rc) return rc;
  plan_name(pl, out, (size_t)out_len);
  return MCHAP_OK;
}

int mchap_denovo_trace_words_per_haplotype(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_host) {
  if (validate_cfg(cfg) || !units_host || n_units < 1) return -1;
  BatchDims B;
  if (batch_dims(cfg, n_units, units_host, B)) return -1;
  Plan pl;
  if (plan_sampler(cfg, tune_of(cfg), B, pl)) return -1;
  return pl.wide ? 2 : 1;
}

#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
/* profiling builds only (make stats / make phases): [0] likelihood requests, [1] cache misses, [2] probe slots, ... */
int mchap_debug_stats(unsigned long long *out, int reset) {
  unsigned long long z[mchap::N_STATS] = {0};
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(mchap::g_stats), sizeof(z)));
  if (reset) HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mchap::g_stats), z, sizeof(z)));
  // plus the copies of the speculative sampler's object files
#define ROW_SPEC_STATS(k, g) mchap_spec_stats_##k##_##g,
#define ROW_SPECP_STATS(k, g) mchap_specp_stats_##k##_##g,
#define ROW_SPECS_STATS(k, g) mchap_specs_stats_##k##_##g, mchap_specd_stats_##k##_##g,
  int (*fs[])(unsigned long long *, int) = {SPEC_LIST(ROW_SPEC_STATS) SPECP_LIST(ROW_SPECP_STATS) SPECS_LIST(ROW_SPECS_STATS)};
  for (auto f : fs) {
    unsigned long long t[mchap::N_STATS];
    if (f(t, reset) != 0) return fail(MCHAP_ERR_HIP, "reading the counters of a sampler object");
    for (int i = 0; i < mchap::N_STATS; i++) out[i] += t[i];
  }
  return MCHAP_OK;
}
/* counters of the table-completion kernel (denovo_fill_kernel.hpp FPH / FCNT) summed over its objects: ticks in
 * [0] listing, [1] de-duplication, [2] staging, [3] evaluation, [4] probabilities + totals; [6] chunks, [7] distinct requests,
 * [8] option slots, [9] stagings, [10] batch x tile evaluations, [11] chains */
#ifdef MCHAP_TEST_KERNELS
#define DECL_FILL_STATS(k) extern "C" int mchap_fill_stats_##k(unsigned long long *, int);
FILL_LIST(DECL_FILL_STATS)
extern "C" int mchap_debug_fill_stats(unsigned long long *out, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < mchap::N_STATS; i++) out[i] = 0;
#define ROW_FILL_STATS(k) mchap_fill_stats_##k,
  int (*fs[])(unsigned long long *, int) = {FILL_LIST(ROW_FILL_STATS)};
  for (auto f : fs) {
    unsigned long long t[mchap::N_STATS];
    if (f(t, reset) != 0) return fail(MCHAP_ERR_HIP, "reading the counters of a table-completion object");
    for (int i = 0; i < mchap::N_STATS; i++) out[i] += t[i];
  }
  return MCHAP_OK;
}
/* counters of the steady-state sampler's K = 4 object (its own layout: denovo_lane_kernel.hpp LPH / LCNT) */
int mchap_debug_lane_stats(unsigned long long *out, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  if (mchap_lane_stats_4(out, reset) != 0) return fail(MCHAP_ERR_HIP, "reading the counters of the lane sampler");
  return MCHAP_OK;
}
#endif
#endif

#ifdef MCHAP_TEST_KERNELS
/* test library only (tools/pipe_records.py, tests): the phased sampler's hand-over records and round counters as the
 * last fit on this workspace left them.  records: [n_units * chains] PipeState (128 bytes each), counts: [8] int32. */
int mchap_debug_pipe_records(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_host, const void *workspace,
                             void *records, int32_t *counts) {
  BatchDims B;
  if (batch_dims(cfg, n_units, units_host, B)) return MCHAP_ERR_BAD_ARG;
  const Tune T = tune_of(cfg);
  Plan pl;
  if (plan_sampler(cfg, T, B, pl) || pl.kind != SAMPLER_PIPE) return fail(MCHAP_ERR_BAD_ARG, "not a phased-sampler batch");
  const SimtCarve cv = simt_carve(cfg, pl, n_units, B, 64 * pl.rpl, cache_slots_of(cfg, T, pl, n_units, B));
  const unsigned char *ws = reinterpret_cast<const unsigned char *>(workspace) + break_table_bytes(cfg);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(records, ws + cv.pipe_state, (size_t)n_units * cfg->chains * sizeof(mchap::PipeState), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(counts, ws + cv.pipe_counts, (PIPE_MAX_ROUNDS + 2) * 4, hipMemcpyDeviceToHost));
  return MCHAP_OK;
}
/* ... and the chains' interval tables [n_units * chains][2][max_pos (max_pos + 1) / 2] float64 (NaN: not evaluated,
 * -1: the step has no options, else the total move probability) */
int mchap_debug_pipe_memo(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_host, const void *workspace, double *memo) {
  BatchDims B;
  if (batch_dims(cfg, n_units, units_host, B)) return MCHAP_ERR_BAD_ARG;
  const Tune T = tune_of(cfg);
  Plan pl;
  if (plan_sampler(cfg, T, B, pl) || pl.kind != SAMPLER_PIPE) return fail(MCHAP_ERR_BAD_ARG, "not a phased-sampler batch");
  const SimtCarve cv = simt_carve(cfg, pl, n_units, B, 64 * pl.rpl, cache_slots_of(cfg, T, pl, n_units, B));
  const unsigned char *ws = reinterpret_cast<const unsigned char *>(workspace) + break_table_bytes(cfg);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(memo, ws + cv.pipe_memo, (size_t)n_units * cfg->chains * 2 * mchap::spec_memo_entries(B.max_pos) * 8, hipMemcpyDeviceToHost));
  return MCHAP_OK;
}
#endif

int mchap_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int64_t mchap_denovo_lds_bytes(int n_reads, int n_pos, int max_allele, int ploidy, int chains, int n_temps) {
  const int rpl = rpl_for(n_reads < 1 ? 1 : n_reads);
  if (rpl < 0 || ploidy > MCHAP_MAX_PLOIDY || max_allele > MCHAP_MAX_ALLELE || n_pos < 1) return -1;
  const mchap::WaveLayout L = mchap::wave_layout(ploidy, n_pos, max_allele, n_temps);
  const int cpb = chains < mchap::CHAINS_PER_BLOCK ? chains : mchap::CHAINS_PER_BLOCK;
  return (int64_t)n_pos * max_allele * 64 * rpl * 8 + (int64_t)cpb * L.total;
}

int64_t mchap_denovo_workspace_bytes(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_host) {
  if (!cfg || n_units <= 0) return 0;
  const Tune T = tune_of(cfg);
  const int64_t bt = (int64_t)break_table_bytes(cfg);
  if (!use_simt(cfg)) return bt + (int64_t)n_units * cfg->chains * (cfg->llk_cache ? T.cache_slots : 0) * 16;
  if (!units_host) return -1;
  BatchDims B;
  if (batch_dims(cfg, n_units, units_host, B)) return -1;
  Plan pl;
  if (plan_sampler(cfg, T, B, pl)) return -1;
  return bt + (int64_t)simt_carve(cfg, pl, n_units, B, 64 * pl.rpl, cache_slots_of(cfg, T, pl, n_units, B), ctx_slots_of(cfg, T, pl, n_units, B)).total;
}

static int fit_batch_device_impl(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_dev,
                                 const mchap_unit *units_host, const double *reads, const int8_t *calls, const int16_t *quals,
                                 const double *qual_prob, int qual_prob_len, const int64_t *read_counts,
                                 const int8_t *n_alleles, const int8_t *initial, uint64_t *trace_words, double *llks,
                                 int8_t *fixed_alleles, int32_t *status, void *workspace, int64_t workspace_bytes,
                                 void *stream_);

int mchap_denovo_fit_batch_device(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_dev,
                                  const mchap_unit *units_host, const double *reads, const int64_t *read_counts,
                                  const int8_t *n_alleles, const int8_t *initial, uint64_t *trace_words, double *llks,
                                  int8_t *fixed_alleles, int32_t *status, void *workspace, int64_t workspace_bytes,
                                  void *stream_) {
  if (n_units > 0 && !reads) return fail(MCHAP_ERR_BAD_ARG, "NULL buffer");
  return fit_batch_device_impl(cfg, n_units, units_dev, units_host, reads, nullptr, nullptr, nullptr, 0, read_counts, n_alleles,
                               initial, trace_words, llks, fixed_alleles, status, workspace, workspace_bytes, stream_);
}

int mchap_denovo_fit_batch_calls_device(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_dev,
                                        const mchap_unit *units_host, const int8_t *calls, const int16_t *quals,
                                        const double *qual_prob, int qual_prob_len, const int64_t *read_counts,
                                        const int8_t *n_alleles, const int8_t *initial, uint64_t *trace_words, double *llks,
                                        int8_t *fixed_alleles, int32_t *status, void *workspace, int64_t workspace_bytes,
                                        void *stream_) {
  if (n_units > 0 && (!calls || !qual_prob || qual_prob_len < 1)) return fail(MCHAP_ERR_BAD_ARG, "NULL buffer");
  if (cfg && cfg->kernel == 1)
    return fail(MCHAP_ERR_BAD_ARG, "allele-call input is read by the prepare pass of kernels 2 and 3; kernel 1 has none");
  return fit_batch_device_impl(cfg, n_units, units_dev, units_host, nullptr, calls, quals, qual_prob, qual_prob_len, read_counts,
                               n_alleles, initial, trace_words, llks, fixed_alleles, status, workspace, workspace_bytes, stream_);
}

static int fit_batch_device_impl(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_dev,
                                 const mchap_unit *units_host, const double *reads, const int8_t *calls, const int16_t *quals,
                                 const double *qual_prob, int qual_prob_len, const int64_t *read_counts,
                                 const int8_t *n_alleles, const int8_t *initial, uint64_t *trace_words, double *llks,
                                 int8_t *fixed_alleles, int32_t *status, void *workspace, int64_t workspace_bytes,
                                 void *stream_) {
  int rc = validate_cfg(cfg);
  if (rc) return rc;
  if (n_units <= 0) return MCHAP_OK;
  if (!units_dev || !units_host || !n_alleles || !trace_words || !llks || !fixed_alleles || !status)
    return fail(MCHAP_ERR_BAD_ARG, "NULL buffer");
  rc = ensure_init();
  if (rc) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  BatchDims B;
  rc = batch_dims(cfg, n_units, units_host, B);
  if (rc) return rc;
  const Tune T = tune_of(cfg);
  Plan pl;
  rc = plan_sampler(cfg, T, B, pl);
  if (rc) return rc;
  const int rpl = pl.rpl;
  const int rpad = 64 * rpl;

  mchap::SimtParams SP;
  std::memset(&SP, 0, sizeof(SP));
  mchap::DenovoParams &P = SP.d;
  P.units = units_dev;
  P.reads = reads;
  P.calls = calls;
  P.quals = quals;
  P.qual_prob = qual_prob;
  P.qual_prob_len = qual_prob_len;
  P.counts = read_counts;
  P.n_alleles = n_alleles;
  P.initial = initial;
  P.trace = trace_words;
  P.llks = llks;
  P.fixed = fixed_alleles;
  P.status = status;
  P.steps = cfg->steps;
  P.chains = cfg->chains;
  P.n_temps = cfg->n_temps;
  P.n_intervals = cfg->n_intervals;
  for (int t = 0; t < cfg->n_temps; t++) P.temps[t] = cfg->temperatures[t];
  P.fix_hom = cfg->fix_homozygous;
  P.p_recomb = cfg->p_recomb;
  P.p_partial = cfg->p_partial_dosage;
  P.p_dosage = cfg->p_dosage;
  P.seed = cfg->seed;
  P.rpad = rpad;
  P.max_pos = cfg->max_pos > 0 ? cfg->max_pos : 1;
  {
    // head of the caller's workspace: this call's break table, copied on this call's stream
    const size_t n = (size_t)(P.max_pos + 1) * P.max_pos;
    const size_t bt_bytes = break_table_bytes(cfg);
    if (!workspace || workspace_bytes < (int64_t)bt_bytes)
      return fail(MCHAP_ERR_BAD_ARG, "workspace of %lld bytes is too small (mchap_denovo_workspace_bytes)", (long long)workspace_bytes);
    double *bt_dev = reinterpret_cast<double *>(workspace);
    if (cfg->break_table)
      HIP_TRY(hipMemcpyAsync(bt_dev, cfg->break_table, n * sizeof(double), hipMemcpyHostToDevice, stream));
    else
      HIP_TRY(hipMemsetAsync(bt_dev, 0, n * sizeof(double), stream));
    P.break_table = bt_dev;
    workspace = reinterpret_cast<unsigned char *>(workspace) + bt_bytes;
    workspace_bytes -= (int64_t)bt_bytes;
  }
  HIP_TRY(hipMemsetAsync(status, 0, sizeof(int32_t) * n_units, stream));
  P.cache = nullptr;
  P.cache_slots = 0;

  if (use_simt(cfg)) {
    const int want_slots = cache_slots_of(cfg, T, pl, n_units, B);
    int slots = want_slots;
    SimtCarve cv = simt_carve(cfg, pl, n_units, B, rpad, slots);
    while (slots >= 32 && (int64_t)cv.total > workspace_bytes) {
      slots >>= 1;
      cv = simt_carve(cfg, pl, n_units, B, rpad, slots);
    }
    if (slots < 32 && want_slots) {
      slots = 0;
      cv = simt_carve(cfg, pl, n_units, B, rpad, 0);
    }
    if (!workspace || (int64_t)cv.total > workspace_bytes)
      return fail(MCHAP_ERR_BAD_ARG, "workspace of %lld bytes is too small: kernel %d needs at least %zu (mchap_denovo_workspace_bytes)",
                  (long long)workspace_bytes, cfg->kernel, cv.total);
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    // decision contexts: as many slots per chain as the rest of the workspace holds (mchap_denovo_workspace_bytes sized it by its budget)
    if (ctx_shape(pl, B, T) && workspace_bytes > (int64_t)cv.total) {
      const int cn = ctx_slots_for((size_t)(workspace_bytes - (int64_t)cv.total), (size_t)n_units * cfg->chains, ctx_slot_bytes(pl, B));
      if (cn > 0) {
        SP.ctx = reinterpret_cast<uint64_t *>(ws + cv.ctx);
        SP.ctx_n = cn;
      }
    }
    if (slots > 0) {
      P.cache = reinterpret_cast<uint64_t *>(ws + cv.cache);
      P.cache_slots = slots;
      // (the words of wide genotypes need no clearing: they are only read behind a matching tag)
      // Narrow genotypes on the speculative / phased sampler: tagged with this call's epoch, nothing cleared (tuning flag 65536:
      // clear as before; flag 64, the parity suite's lane-per-request completion, forms its own tags)
      uint64_t epoch = 0;
      if ((pl.kind == SAMPLER_PIPE || pl.kind == SAMPLER_SPEC) && !(T.flags & (64 | 65536)) &&
          B.max_ploidy * mchap::allele_bits(B.max_allele) * B.max_pos <= 32) {
        // the caller's number (a property of the call: the Python host draws from one sequence for every library copy in the
        // process -- two copies counting on their own could tag different data alike), else this copy's counter
        epoch = cfg->cache_epoch > 0 ? (uint64_t)cfg->cache_epoch : g_cache_epoch.fetch_add(1) + 1;
        if (epoch >= (1ull << 31) - 1) epoch = 0;
      }
      SP.cache_epoch = epoch << 33;
      if (!epoch)
        HIP_TRY(hipMemsetAsync(ws + cv.cache, 0, (size_t)n_units * cfg->chains * (pl.kind == SAMPLER_SPEC ? cfg->n_temps : 1) * slots * 16, stream));
      if (cv.key_words) {
        P.cache_keys = reinterpret_cast<uint64_t *>(ws + cv.ckeys);
        P.cache_key_words = cv.key_words;
      }
    }
    SP.rt = reinterpret_cast<double *>(ws + cv.rt);
    SP.cntw = reinterpret_cast<double *>(ws + cv.cntw);
    SP.codes = ws + cv.codes;
    SP.dict = reinterpret_cast<double *>(ws + cv.dict);
    SP.gbp = (cv.has_gbp && !(T.flags & 512)) ? reinterpret_cast<double *>(ws + cv.gbp) : nullptr;
    SP.meta_i = reinterpret_cast<int32_t *>(ws + cv.meta_i);
    SP.meta_f = reinterpret_cast<double *>(ws + cv.meta_f);
    SP.lane_state = ws + cv.lane_state;
    SP.lane_memo = ws + cv.lane_memo;
    SP.pipe_state = ws + cv.pipe_state;
    SP.pipe_memo = reinterpret_cast<double *>(ws + cv.pipe_memo);
    SP.n_units = n_units;
    SP.max_pos = B.max_pos;
    SP.max_allele = B.max_allele;
    SP.max_ploidy = B.max_ploidy;
    SP.max_ma = B.max_ma;
    SP.cstride = code_stride(rpl);
    SP.flags = T.flags & (63 | 128 | 2048 | 4096 | 262144 | (1 << 20));  // (1 << 20: no grouped logarithms -- read_log_sum, META_I_W01)  // (2048 / 4096: denovo_fillw_kernel without the cache probe / the LDS table)  // (256, 512: host only -- never the side-by-side instantiation / no deep-chunk product rows)
    // the prepare pass keeps the transposed table in LDS when it fits, else it re-reads its own global copy
    SP.max_ugens_pad = (B.max_ugens + 8) & ~7;
    const size_t lds_dict = (size_t)mchap::DICT_HASH * (8 + 2) + 64;  // hash set of the dictionary pass
    size_t lds_prep = (size_t)B.max_ma * rpad * 8 + (size_t)SP.max_ugens_pad * 8 + lds_dict;
    // (an LDS copy of more than a few KB costs the prepare pass its occupancy: one wavefront per workgroup)
    SP.prep_rows_off = 0;
    if (lds_prep > 160 * 1024 || (size_t)B.max_ma * rpad * 8 > T.prep_lds_limit) {
      lds_prep = (size_t)SP.max_ugens_pad * 8 + lds_dict;
      lds_prep = (lds_prep + 15) & ~(size_t)15;
      SP.prep_rows_off = (int)lds_prep;  // the rows of one position: [max_allele][rpad] float64
      lds_prep += (size_t)B.max_allele * rpad * 8;
      SP.flags |= mchap::SIMT_FLAG_PREP_GLOBAL;
    }
    const size_t lds_simt = mchap::simt_lds_bytes(B.max_ploidy, B.max_pos, cfg->n_temps, pl.wide ? 16 : 8);
    SP.word_bits = pl.wide ? 128 : 64;
    if (lds_prep > 160 * 1024 || lds_simt > 160 * 1024)
      return fail(MCHAP_ERR_LIMIT, "a unit needs %zu / %zu bytes of LDS (> 160 KiB)", lds_prep, lds_simt);
    auto prepare = [&](const mchap::SimtParams &Q, int n, hipStream_t st) {
      switch (rpl) {
        case 1: return launch_prepare<1>(Q, n, lds_prep, st);
        case 2: return launch_prepare<2>(Q, n, lds_prep, st);
        case 3: return launch_prepare<3>(Q, n, lds_prep, st);
        case 4: return launch_prepare<4>(Q, n, lds_prep, st);
        case 5: return launch_prepare<5>(Q, n, lds_prep, st);
        case 6: return launch_prepare<6>(Q, n, lds_prep, st);
        case 7: return launch_prepare<7>(Q, n, lds_prep, st);
        case 8: return launch_prepare<8>(Q, n, lds_prep, st);
        case 12: return launch_prepare<12>(Q, n, lds_prep, st);
        case 16: return launch_prepare<16>(Q, n, lds_prep, st);
        case 24: return launch_prepare<24>(Q, n, lds_prep, st);
        case 32: return launch_prepare<32>(Q, n, lds_prep, st);
        case 48: return launch_prepare<48>(Q, n, lds_prep, st);
        default: return launch_prepare<64>(Q, n, lds_prep, st);
      }
    };
    rc = prepare(SP, n_units, stream);
    if (rc) return rc;
    switch (pl.kind) {
#ifdef MCHAP_TEST_KERNELS
      case SAMPLER_LANE: return launch_lane(T, pl.K, SP, n_units, cfg->chains, cfg->timer, stream);
#endif
      case SAMPLER_PIPE:
        return launch_pipe(T, pl.K, pl.G, SP, n_units, cfg->chains, reinterpret_cast<int32_t *>(ws + cv.pipe_lists),
                           reinterpret_cast<int32_t *>(ws + cv.pipe_counts), cfg->timer, stream, B.min_reads <= 64, B.max_reads);
      case SAMPLER_SPEC: return launch_spec(T, pl.K, pl.G, SP, n_units, cfg->chains, cfg->n_temps, cfg->timer, stream);
      default: return launch_simt(pl.K, SP, n_units, cfg->chains, lds_simt, cfg->timer, stream, pl.wide);  // lanes over chains
    }
  }

#ifdef MCHAP_TEST_KERNELS
  // ---- kernel 1: wavefront per chain, reads staged in LDS ----
  size_t lds = 0;
  for (int u = 0; u < n_units; u++) {
    const mchap_unit &U = units_host[u];
    const mchap::WaveLayout L = mchap::wave_layout(U.ploidy, U.n_pos, U.max_allele, cfg->n_temps);
    const int cpb = cfg->chains < mchap::CHAINS_PER_BLOCK ? cfg->chains : mchap::CHAINS_PER_BLOCK;
    const size_t need = (size_t)U.n_pos * U.max_allele * rpad * 8 + (size_t)cpb * L.total;
    if (need > lds) lds = need;
  }
  if (lds > 160 * 1024)
    return fail(MCHAP_ERR_LIMIT, "a unit needs %zu bytes of LDS (> 160 KiB): dense float64 staging does not fit", lds);
  if (cfg->llk_cache && workspace && workspace_bytes > 0) {
    const int64_t rows = (int64_t)n_units * cfg->chains;
    int slots = T.cache_slots;
    while (slots >= 16 && rows * slots * 16 > workspace_bytes) slots >>= 1;
    if (slots >= 16) {
      P.cache = reinterpret_cast<uint64_t *>(workspace);
      P.cache_slots = slots;
      HIP_TRY(hipMemsetAsync(workspace, 0, (size_t)(rows * slots * 16), stream));
    }
  }
  return launch_denovo(rpl, P, n_units, cfg->chains, lds, cfg->timer, stream);
#else
  return fail(MCHAP_ERR_BAD_ARG, "kernel 1 is only built into libmchap_hip_test.so");
#endif
}

int mchap_denovo_fit_batch(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units, const double *reads,
                           int64_t reads_len, const int64_t *read_counts, int64_t counts_len, const int8_t *n_alleles,
                           int64_t nalleles_len, const int8_t *initial, int64_t initial_len, uint64_t *trace_words,
                           int64_t trace_len, double *llks, int64_t llks_len, int8_t *fixed_alleles, int64_t fixed_len,
                           int32_t *status) {
  int rc = validate_cfg(cfg);
  if (rc) return rc;
  rc = ensure_init();
  if (rc) return rc;
  if (n_units <= 0) return MCHAP_OK;
  DevBuf d_units, d_reads, d_counts, d_nal, d_init, d_trace, d_llk, d_fixed, d_status, d_ws;
  const int64_t ws_bytes = mchap_denovo_workspace_bytes(cfg, n_units, units);
  if (ws_bytes < 0) return fail(MCHAP_ERR_LIMIT, "unsupported unit shape");
  HostCall hc;
  MCHAP_TRY(hc.open());
  if (ws_bytes > 0) HIP_TRY(hipMalloc(&d_ws.p, (size_t)ws_bytes));
  HIP_TRY(hipMalloc(&d_units.p, sizeof(mchap_unit) * n_units));
  HIP_TRY(hipMalloc(&d_reads.p, sizeof(double) * (size_t)reads_len));
  HIP_TRY(hipMalloc(&d_nal.p, (size_t)nalleles_len));
  HIP_TRY(hipMalloc(&d_trace.p, sizeof(uint64_t) * (size_t)trace_len));
  HIP_TRY(hipMalloc(&d_llk.p, sizeof(double) * (size_t)llks_len));
  HIP_TRY(hipMalloc(&d_fixed.p, (size_t)fixed_len));
  HIP_TRY(hipMalloc(&d_status.p, sizeof(int32_t) * n_units));
  MCHAP_TRY(hc.up(d_units.p, units, sizeof(mchap_unit) * n_units));
  MCHAP_TRY(hc.up(d_reads.p, reads, sizeof(double) * (size_t)reads_len));
  MCHAP_TRY(hc.up(d_nal.p, n_alleles, (size_t)nalleles_len));
  if (read_counts && counts_len > 0) {
    HIP_TRY(hipMalloc(&d_counts.p, sizeof(int64_t) * (size_t)counts_len));
    MCHAP_TRY(hc.up(d_counts.p, read_counts, sizeof(int64_t) * (size_t)counts_len));
  }
  if (initial && initial_len > 0) {
    HIP_TRY(hipMalloc(&d_init.p, (size_t)initial_len));
    MCHAP_TRY(hc.up(d_init.p, initial, (size_t)initial_len));
  }
  rc = mchap_denovo_fit_batch_device(cfg, n_units, d_units.as<mchap_unit>(), units, d_reads.as<double>(),
                                     d_counts.as<int64_t>(), d_nal.as<int8_t>(), d_init.as<int8_t>(),
                                     d_trace.as<uint64_t>(), d_llk.as<double>(), d_fixed.as<int8_t>(),
                                     d_status.as<int32_t>(), d_ws.p, ws_bytes, hc.stream);
  if (rc) {
    (void)hipStreamSynchronize(hc.stream);
    return rc;
  }
  MCHAP_TRY(hc.down(trace_words, d_trace.p, sizeof(uint64_t) * (size_t)trace_len));
  MCHAP_TRY(hc.down(llks, d_llk.p, sizeof(double) * (size_t)llks_len));
  MCHAP_TRY(hc.down(fixed_alleles, d_fixed.p, (size_t)fixed_len));
  MCHAP_TRY(hc.down(status, d_status.p, sizeof(int32_t) * n_units));
  return hc.sync();
}

int mchap_log_likelihood_batch(const double *reads, int n_reads, int n_pos, int max_allele, const int64_t *read_counts,
                               const int8_t *genotypes, int n_genotypes, int ploidy, double *llks_out) {
  int rc = ensure_init();
  if (rc) return rc;
  if (n_genotypes <= 0) return MCHAP_OK;
  const int rpl = rpl_for(n_reads);
  if (rpl < 0 || n_reads < 1) return fail(MCHAP_ERR_LIMIT, "n_reads %d not in 1..%d", n_reads, MCHAP_MAX_READS);
  if (ploidy < 1 || ploidy > MCHAP_MAX_PLOIDY) return fail(MCHAP_ERR_LIMIT, "ploidy %d not supported", ploidy);
  if (n_pos * mchap::allele_bits(max_allele) > 64) return fail(MCHAP_ERR_LIMIT, "n_pos * bits > 64");
  const size_t lds = (size_t)n_pos * max_allele * 64 * rpl * 8 + 4 * (8 * ploidy + 4 * n_pos + 64);
  if (lds > 160 * 1024) return fail(MCHAP_ERR_LIMIT, "unit needs %zu bytes of LDS", lds);
  DevBuf d_reads, d_counts, d_g, d_out;
  HostCall hc;
  MCHAP_TRY(hc.open());
  const size_t nr = (size_t)n_reads * n_pos * max_allele;
  HIP_TRY(hipMalloc(&d_reads.p, nr * 8));
  MCHAP_TRY(hc.up(d_reads.p, reads, nr * 8));
  if (read_counts) {
    HIP_TRY(hipMalloc(&d_counts.p, (size_t)n_reads * 8));
    MCHAP_TRY(hc.up(d_counts.p, read_counts, (size_t)n_reads * 8));
  }
  const size_t ng = (size_t)n_genotypes * ploidy * n_pos;
  HIP_TRY(hipMalloc(&d_g.p, ng));
  MCHAP_TRY(hc.up(d_g.p, genotypes, ng));
  HIP_TRY(hipMalloc(&d_out.p, (size_t)n_genotypes * 8));
  const int blocks = n_genotypes < 1024 ? (n_genotypes + 3) / 4 : 256;
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, hc.stream, d_reads.as<double>(), n_reads, n_pos, max_allele,
                       d_counts.as<int64_t>(), d_g.as<int8_t>(), n_genotypes, ploidy, 64 * rpl, d_out.as<double>());
    HIP_TRY(hipGetLastError());
    return MCHAP_OK;
  };
  switch (rpl) {
    case 1: rc = go(mchap::llk_batch_kernel<1>); break;
    case 2: rc = go(mchap::llk_batch_kernel<2>); break;
    case 4: rc = go(mchap::llk_batch_kernel<4>); break;
    case 8: rc = go(mchap::llk_batch_kernel<8>); break;
    default: rc = go(mchap::llk_batch_kernel<16>); break;
  }
  if (rc) return rc;
  MCHAP_TRY(hc.down(llks_out, d_out.p, (size_t)n_genotypes * 8));
  return hc.sync();
}

/* Test hook: read_log (csrc/read_log.hpp), the logarithm every likelihood kernel takes of its per-read terms, for n
 * arguments.  Host pointers. */
int mchap_read_log_batch(const double *x, int64_t n, double *out) {
  int rc = ensure_init();
  if (rc) return rc;
  if (n <= 0) return MCHAP_OK;
  if (!x || !out) return fail(MCHAP_ERR_BAD_ARG, "NULL buffer");
  DevBuf d_x, d_o;
  HostCall hc;
  MCHAP_TRY(hc.open());
  HIP_TRY(hipMalloc(&d_x.p, (size_t)n * 8));
  HIP_TRY(hipMalloc(&d_o.p, (size_t)n * 8));
  MCHAP_TRY(hc.up(d_x.p, x, (size_t)n * 8));
  hipLaunchKernelGGL(mchap::read_log_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, hc.stream,
                     d_x.as<double>(), (long long)n, d_o.as<double>());
  HIP_TRY(hipGetLastError());
  MCHAP_TRY(hc.down(out, d_o.p, (size_t)n * 8));
  return hc.sync();
}

namespace mchap {
static __global__ void read_log_product_kernel(const double *x, long long n, int group, double *out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double *g = x + i * group;
    if (group == 4) {
      const double y[4] = {g[0], g[1], g[2], g[3]};
      out[i] = read_log_product<4>(y);
    } else if (group == 3) {
      const double y[3] = {g[0], g[1], g[2]};
      out[i] = read_log_product<3>(y);
    } else if (group == 2) {
      const double y[2] = {g[0], g[1]};
      out[i] = read_log_product<2>(y);
    } else {
      out[i] = read_log(g[0]);
    }
  }
}
}  // namespace mchap

/* Test hook: read_log_product (csrc/read_log.hpp, round 5): out[i] = the logarithm of the product of x[i * group .. + group - 1]
 * (group = 1..4) as the likelihood kernels form it for the reads of one lane where the read weights are 0 / 1.  Host pointers. */
int mchap_read_log_product_batch(const double *x, int64_t n, int group, double *out) {
  int rc = ensure_init();
  if (rc) return rc;
  if (n <= 0) return MCHAP_OK;
  if (!x || !out || group < 1 || group > 4) return fail(MCHAP_ERR_BAD_ARG, "NULL buffer or group not in 1..4");
  DevBuf d_x, d_o;
  HostCall hc;
  MCHAP_TRY(hc.open());
  HIP_TRY(hipMalloc(&d_x.p, (size_t)n * group * 8));
  HIP_TRY(hipMalloc(&d_o.p, (size_t)n * 8));
  MCHAP_TRY(hc.up(d_x.p, x, (size_t)n * group * 8));
  hipLaunchKernelGGL(mchap::read_log_product_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, hc.stream,
                     d_x.as<double>(), (long long)n, group, d_o.as<double>());
  HIP_TRY(hipGetLastError());
  MCHAP_TRY(hc.down(out, d_o.p, (size_t)n * 8));
  return hc.sync();
}

/* Test hook: wave_sum (csrc/denovo_kernel.hpp), the 64-lane sum every likelihood goes through, for n_waves x 64 values: out[w] =
 * the sum of x[64 w .. 64 w + 63] as lane 0 holds it.  Host pointers. */
int mchap_wave_sum_batch(const double *x, int64_t n_waves, double *out) {
  int rc = ensure_init();
  if (rc) return rc;
  if (n_waves <= 0) return MCHAP_OK;
  if (!x || !out) return fail(MCHAP_ERR_BAD_ARG, "NULL buffer");
  DevBuf d_x, d_o;
  HostCall hc;
  MCHAP_TRY(hc.open());
  HIP_TRY(hipMalloc(&d_x.p, (size_t)n_waves * 64 * 8));
  HIP_TRY(hipMalloc(&d_o.p, (size_t)n_waves * 8));
  MCHAP_TRY(hc.up(d_x.p, x, (size_t)n_waves * 64 * 8));
  hipLaunchKernelGGL(mchap::wave_sum_kernel, dim3((unsigned)n_waves), dim3(64), 0, hc.stream, d_x.as<double>(), d_o.as<double>());
  HIP_TRY(hipGetLastError());
  MCHAP_TRY(hc.down(out, d_o.p, (size_t)n_waves * 8));
  return hc.sync();
}

#include "api_posterior_exact.inc"

}  // extern "C"
