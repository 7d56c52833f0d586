// mhlik.hip -- C-ABI of the MI355X MH likelihood library (see include/mhlik.h).
//
// Host side of the hot path: context + HBM layout + kernel launches.  There is
// NO CPU fallback in here: every evaluation entry point launches HIP kernels
// or fails with an error string.
#include "mhlik.h"

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mhlik_kernels.hpp"
#include "mhlik_batched.hpp"
#include "mhlik_batched_x3.hpp"
#include "mhlik_mp.hpp"
#include "mhlik_mp_batched.hpp"
#include "mhlik_chain.hpp"
#include "mhlik_kchain.hpp"
#include "mhlik_stats.hpp"
#include "mhlik_synth.h"

using namespace mhlik;

// Scratch of the allocation-free entry point (mhlik_loglik_f64_noalloc): one double per row tile.  Every GPU the
// library runs on holds its own 2 MiB copy of this array, at its own address, until the process ends.  In host code
// the NAME is only a stand-in for the symbol: the device address is looked up once per device in mhlik_create
// (hipGetSymbolAddress, kept in g_scratch_ptr[device]) and only that pointer is ever handed to a kernel.
#define MHLIK_NOALLOC_MAX_TILES_ (1 << 18)
static_assert(MHLIK_NOALLOC_MAX_TILES_ == MHLIK_NOALLOC_MAX_TILES, "mhlik.h states the scratch size");
static __device__ __align__(256) double scratch[MHLIK_NOALLOC_MAX_TILES_];

namespace {

thread_local std::string g_create_error;

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
double *g_scratch_ptr[kMaxDevices]; // device address of `scratch` on each device, nullptr until looked up

// Once per device, with that device current.
hipError_t scratch_lookup(int dev)
{
  if (dev < 0 || dev >= kMaxDevices) return hipSuccess; // such a device simply has no allocation-free path
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[dev]) return hipSuccess;
  void *p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e == hipSuccess) g_scratch_ptr[dev] = (double *)p;
  return e;
}

// ------------------------------------------------------------------ RCCL
// Bound at run time (dlopen) so that single-GPU use needs no librccl and so
// that a process that already loaded torch's RCCL shares that copy.
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
static_assert(sizeof(ncclUniqueId) == MHLIK_COMM_ID_BYTES, "nccl id size");
static_assert(mhlik::kMaxSegmentsDev == MHLIK_MAX_SEGMENTS, "segment slots of the chain step kernel");
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommCount)(const ncclComm_t, int *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};
RcclApi g_rccl;

bool rccl_load(std::string &err)
{
  if (g_rccl.ok) return true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) {
    err = std::string("cannot load librccl: ") + dlerror();
    return false;
  }
#define RSYM(field, name)                                                         \
  *(void **)(&g_rccl.field) = dlsym(g_rccl.lib, name);                            \
  if (!g_rccl.field) {                                                            \
    err = std::string("librccl lacks ") + name;                                   \
    return false;                                                                 \
  }
  RSYM(GetUniqueId, "ncclGetUniqueId")
  RSYM(CommInitRank, "ncclCommInitRank")
  RSYM(CommDestroy, "ncclCommDestroy")
  RSYM(CommCount, "ncclCommCount")
  RSYM(AllReduce, "ncclAllReduce")
  RSYM(GetErrorString, "ncclGetErrorString")
#undef RSYM
  g_rccl.ok = true;
  return true;
}
constexpr int kNcclFloat64 = 8; // ncclDouble
constexpr int kNcclSum = 0;

// ------------------------------------------------------------------ layout

struct Segment {
  int64_t row_begin = 0; // global
  int64_t rows = 0;
  int global_index = 0;
};

// One precision's copy of the shard.
struct Store {
  void *X = nullptr;
  size_t elem = 0;        // sizeof(T)
  int tile_rows = 0;      // rows per tile (kBlock * VEC * R)
  int64_t ntiles = 0;
  int64_t tile_stride = 0, col_stride = 0; // in elements
  int32_t *tile_valid = nullptr;           // device [ntiles]
  int64_t *tile_row0 = nullptr;            // device [ntiles], global row of the tile's first row
  SegDesc *segs = nullptr;                 // device [nseg_local]
  PersistJob *pjobs = nullptr;             // device [npjobs]: the (segment, lane) jobs of the persistent chain kernel
  int npjobs = 0;
  std::vector<PersistJob> h_pjobs;
  std::vector<int32_t> h_valid;
  std::vector<int64_t> h_row0;
  std::vector<SegDesc> h_segs;
  bool has_data = false;
  int R = 1, U = 16;
  bool nt = true;
  size_t bytes = 0;
};

// Device-resident MH chain (mhlik_chain.hpp): the state, and two batch slots so that a second batch can be
// submitted while the first one runs (mhlik_chain_submit / _collect).  A slot owns one upload image and one download
// image on the device and their page-locked host mirrors; everything is allocated by begin / resume (for at least
// kChainMinCap iterations) and only ever grows, so a run that follows a burn-in allocates nothing.
struct ChainSlot {
  unsigned char *dev_up = nullptr, *dev_down = nullptr; // device images
  unsigned char *h_up = nullptr, *h_down = nullptr;     // page-locked host mirrors
  int64_t cap_n = 0;                                    // iterations the images hold (8-byte elements)
  hipEvent_t done = nullptr;                            // recorded after the batch's download
  std::vector<hipEvent_t> ev;                           // timing pairs, created with the slot
  int64_t n = 0;                                        // the batch in flight
  int pairs = 0;
  bool want_samples = false;
  bool persist = false;                                 // ran as ONE persistent launch: pairs == 1 brackets all n iterations
};
struct ChainHost {
  ChainHdr *hdr = nullptr;
  void *cur = nullptr, *prop = nullptr;
  size_t state_bytes = 0;
  ChainSlot slot[2];
  int head = 0, inflight = 0; // slot[head] holds the oldest batch in flight
  double pdf_norm = 0, abs_sigma = 1;
  int prec = 0;
  // mhlik_chain_resume keeps the caller's state here; the next batch carries it to the device (load_state).
  // After a collect that leaves nothing in flight it is the device-side state again (host_state_valid).
  bool load_state = false, host_state_valid = false;
  ChainHdr host_hdr;
  std::vector<unsigned char> host_cur;
  unsigned char *persist_words = nullptr; // device: [abort word, 16 B][2][2 * kPersistMaxJobs] granules
};

// K lock-step chains on the device (mhlik_kchain.hpp): states + one batch of buffers.
struct KChainHost {
  int mode = 0, K = 0;
  double *cpost = nullptr, *cur = nullptr, *prop = nullptr;
  int state_K = 0;
  void *image = nullptr; // Theta in the batched pass's operand order, built on the device from prop
  size_t image_cap = 0;
  double *mp_res = nullptr; // [K][4] results of the mixed-precision evaluation (device)
  double *inc = nullptr, *logu = nullptr, *samples = nullptr, *lik = nullptr, *ratio = nullptr;
  unsigned char *acc = nullptr;
  MpbChainArgs *mp_args = nullptr; // [cap_n] phases and seeds of the batch's evaluations
  int64_t cap_n = 0;
  int cap_K = 0;
  double pdf_norm = 0, abs_sigma = 1;
  double ms_sum = 0; // sampled pass times since begin / resume
  int ms_n = 0;
  std::vector<hipEvent_t> ev;
};

} // namespace

struct mhlik_ctx {
  mhlik_config cfg;
  int device = 0;
  int compute_units = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int nseg = 8;
  std::vector<Segment> segs; // local
  int64_t local_begin = 0, local_rows = 0;
  int64_t seg_rows = 0; // rows per fixed segment (the last non-empty one may hold fewer)
  int layout = MHLIK_LAYOUT_TILED;
  Store s64, s32;

  double *partials = nullptr;   // device [max tiles * MHLIK_MAX_BATCH]
  int64_t partials_cap = 0;
  double *slots_send = nullptr; // device [MAX_BATCH * nseg], zero outside own segments
  double *slots_recv = nullptr; // device [MAX_BATCH * nseg]
  double *slots_host = nullptr; // pinned host [MAX_BATCH * nseg]
  double *slots_host_dev = nullptr; // device view of slots_host
  unsigned long long *done_flag = nullptr;     // pinned host word the last kernel writes
  unsigned long long *done_flag_dev = nullptr; // its device view
  unsigned long long done_seq = 0;
  double *theta_dev = nullptr;  // device staging for d > kMaxKargCols and for batched Theta
  void *theta_pin = nullptr;    // pinned host staging
  size_t theta_cap = 0;
  double *rowdots = nullptr;
  size_t rowdots_cap = 0;
  int batched_f32_mode = MHLIK_BATCHED_F32_SPLIT; // mhlik_set_batched_f32_mode
  int batched_f64_mode = MHLIK_BATCHED_F64_FAST;  // mhlik_set_batched_f64_mode
  // an evaluation begun with mhlik_*_begin and not yet collected with the matching _end
  int pending_kind = 0; // 0 none, 1 batched, 2 mixed-precision batched
  int pending_nvec = 0;
  unsigned long long pending_seq = 0;
  bool pending_published = false;

  // mixed precision
  int8_t *mp_z = nullptr;
  int32_t mp_blocks = 0, mp_block_rows = 0;
  double *mp_out = nullptr; // device scratch
  size_t mp_out_cap = 0;
  std::vector<double> mp_last_slots; // per-segment sums of the last mixed-precision evaluation
  // mixed precision, K chains (mhlik_mp_batched.hpp)
  int8_t *mpb_z = nullptr;      // [blocks][64]
  int32_t mpb_blocks = 0, mpb_block_rows = 0, mpb_K = 0;
  double *mpb_acc = nullptr;    // [kMpbFixWaves][64] bright sums
  int32_t *mpb_cnt = nullptr;   // [kMpbFixWaves][64] bright counts
  double *mpb_theta = nullptr;  // [64][d] row-major
  float *mpb_thetaf = nullptr;
  double *mpb_res = nullptr, *mpb_res_dev = nullptr; // pinned host [64][4] and its device view
  void *mpb_segtab = nullptr;   // device MpbSeg[local segments]: each segment's data blocks
  double *mpb_slots_send = nullptr, *mpb_slots_recv = nullptr; // device [64*4][nseg], zero outside the local segments
  int mpb_seg_groups = 0;

  void *stats_buf = nullptr; // chain statistics scratch (mhlik_stats.hpp)
  size_t stats_cap = 0;

  ncclComm_t comm = nullptr;
  ChainHost chain;
  KChainHost kchain;

  float last_ms = 0.f;
  int64_t last_bytes = 0;
  char last_kernel[96] = ""; // instantiation of the last streaming kernel launched, as a profiler prints it
  std::vector<double> last_seg_sums;
  size_t hbm_bytes = 0;
  std::string err;
};

namespace {

int fail(mhlik_ctx *c, int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c)
    c->err = buf;
  else
    g_create_error = buf;
  return code;
}

#define HIPCHK(c, expr)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return fail((c), e_ == hipErrorOutOfMemory ? MHLIK_ENOMEM : MHLIK_EHIP, "%s failed: %s (%s:%d)", \
                  #expr, hipGetErrorString(e_), __FILE__, __LINE__);                                 \
  } while (0)

void chain_free(ChainHost &h);
void kchain_free(KChainHost &h);
void mpb_free(mhlik_ctx *c);
void mpb_clear_slots(mhlik_ctx *c);

struct VariantDef {
  int R, U;
  bool nt;
};
// Kernel variants (config.variant).  0 picks kDefaultVariant.
const VariantDef kVariants[] = {
    {0, 0, false},  // 0: placeholder
    {1, 16, true},  // 1
    {2, 8, true},   // 2
    {4, 4, true},   // 3
    {1, 8, true},   // 4
    {2, 16, true},  // 5
    {4, 8, true},   // 6
    {1, 32, true},  // 7
    {1, 16, false}, // 8
    {2, 8, false},  // 9
    {4, 4, false},  // 10
    {2, 4, true},   // 11
    {8, 2, true},   // 12
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
constexpr int kDefaultVariant = 1;

template <typename T> Store &store_of(mhlik_ctx *c);
template <> Store &store_of<double>(mhlik_ctx *c) { return c->s64; }
template <> Store &store_of<float>(mhlik_ctx *c) { return c->s32; }

int64_t tile_offset(const Store &s, int64_t t) { return t * s.tile_stride; }

template <typename T> int build_store(mhlik_ctx *c)
{
  Store &s = store_of<T>(c);
  if (s.X) return MHLIK_OK;
  const int variant = c->cfg.variant ? c->cfg.variant : kDefaultVariant;
  const VariantDef v = kVariants[variant];
  s.elem = sizeof(T);
  s.R = v.R;
  s.U = v.U;
  s.nt = v.nt;
  s.tile_rows = kBlock * Vec16<T>::N * v.R;
  const int d = c->cfg.n_cols;

  s.h_valid.clear();
  s.h_row0.clear();
  s.h_segs.clear();
  int64_t t = 0;
  for (const Segment &sg : c->segs) {
    SegDesc sd;
    sd.tile_begin = t;
    sd.tile_count = (sg.rows + s.tile_rows - 1) / s.tile_rows;
    sd.global_index = sg.global_index;
    sd.pad = 0;
    for (int64_t k = 0; k < sd.tile_count; k++) {
      int64_t left = sg.rows - k * s.tile_rows;
      s.h_valid.push_back((int32_t)(left < s.tile_rows ? left : s.tile_rows));
      s.h_row0.push_back(sg.row_begin + k * s.tile_rows);
    }
    t += sd.tile_count;
    s.h_segs.push_back(sd);
  }
  s.ntiles = t;
  if (c->layout == MHLIK_LAYOUT_TILED) {
    s.tile_stride = (int64_t)s.tile_rows * d;
    s.col_stride = s.tile_rows;
  } else {
    s.tile_stride = s.tile_rows;
    s.col_stride = s.ntiles * s.tile_rows;
  }
  s.bytes = (size_t)s.ntiles * s.tile_rows * d * sizeof(T);

  hipError_t e;
  if (s.ntiles > 0) {
    e = hipMalloc(&s.X, s.bytes);
    if (e != hipSuccess)
      return fail(c, MHLIK_ENOMEM, "hipMalloc of %zu bytes for the design matrix failed: %s", s.bytes,
                  hipGetErrorString(e));
    HIPCHK(c, hipMemsetAsync(s.X, 0, s.bytes, c->stream));
    HIPCHK(c, hipMalloc((void **)&s.tile_valid, sizeof(int32_t) * s.ntiles));
    HIPCHK(c, hipMalloc((void **)&s.tile_row0, sizeof(int64_t) * s.ntiles));
    HIPCHK(c, hipMemcpy(s.tile_valid, s.h_valid.data(), sizeof(int32_t) * s.ntiles, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(s.tile_row0, s.h_row0.data(), sizeof(int64_t) * s.ntiles, hipMemcpyHostToDevice));
  }
  if (!s.h_segs.empty()) {
    HIPCHK(c, hipMalloc((void **)&s.segs, sizeof(SegDesc) * s.h_segs.size()));
    HIPCHK(c, hipMemcpy(s.segs, s.h_segs.data(), sizeof(SegDesc) * s.h_segs.size(), hipMemcpyHostToDevice));
  }
  // jobs of the persistent chain kernel: lane i of segment g's sum owns that segment's tiles i, i+64, ...
  // (ordered lane-major, so that a grid of W workgroups taking jobs w, w+W, ... spreads every segment)
  s.npjobs = 0;
  if ((int)c->segs.size() == c->nseg && c->nseg <= kPersistMaxSeg && s.ntiles > 0) {
    std::vector<PersistJob> jobs;
    for (int lane = 0; lane < kWave; lane++)
      for (const SegDesc &sd : s.h_segs)
        if (sd.tile_count > lane) {
          PersistJob j;
          j.first_tile = sd.tile_begin + lane;
          j.ntiles = (int32_t)((sd.tile_count - lane + kWave - 1) / kWave);
          j.pub_idx = sd.global_index * kWave + lane;
          jobs.push_back(j);
        }
    HIPCHK(c, hipMalloc((void **)&s.pjobs, sizeof(PersistJob) * jobs.size()));
    HIPCHK(c, hipMemcpy(s.pjobs, jobs.data(), sizeof(PersistJob) * jobs.size(), hipMemcpyHostToDevice));
    s.npjobs = (int)jobs.size();
    s.h_pjobs = jobs;
  }
  c->hbm_bytes += s.bytes;

  // partial-sum scratch: one double per tile per chain
  const int64_t need = (s.ntiles > 0 ? s.ntiles : 1) * MHLIK_MAX_BATCH;
  if (need > c->partials_cap) {
    if (c->partials) (void)hipFree(c->partials);
    HIPCHK(c, hipMalloc((void **)&c->partials, sizeof(double) * need));
    c->partials_cap = need;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MHLIK_OK;
}

void free_store(Store &s)
{
  if (s.X) (void)hipFree(s.X);
  if (s.tile_valid) (void)hipFree(s.tile_valid);
  if (s.tile_row0) (void)hipFree(s.tile_row0);
  if (s.segs) (void)hipFree(s.segs);
  if (s.pjobs) (void)hipFree(s.pjobs);
  s = Store();
}

// ------------------------------------------------------------------ kernels

template <typename T>
__global__ __launch_bounds__(kBlock) void synth_fill_kernel(T *__restrict__ X, TileGeom g,
                                                            const int32_t *__restrict__ tile_valid,
                                                            const int64_t *__restrict__ tile_row0, uint64_t seed)
{
  const int64_t tile = blockIdx.x;
  const int valid = tile_valid[tile];
  const int64_t row0 = tile_row0[tile];
  T *tbase = X + tile * g.tile_stride;
  for (int r = threadIdx.x; r < valid; r += kBlock) {
    const int64_t row = row0 + r;
    const double y = (double)mhlik_synth_label(seed, row, g.d);
    for (int j = 0; j < g.d; j++) {
      const double v = -y * mhlik_synth_feature(seed, row, j);
      tbase[(int64_t)j * g.col_stride + r] = (T)v;
    }
  }
}

// Upload staging -> tiled layout: workgroup b moves tile (tile0 + b) out of a
// column-major chunk (leading dimension stage_ld, the chunk's first row is the
// tile's row 0 for b = 0).  Rows past the tile's valid count stay zero.
template <typename T>
__global__ __launch_bounds__(kBlock) void retile_kernel(const T *__restrict__ stage, int64_t stage_ld, T *__restrict__ X,
                                                        TileGeom g, int64_t tile0, const int32_t *__restrict__ tile_valid,
                                                        int tile_rows)
{
  const int64_t tile = tile0 + blockIdx.x;
  const int valid = tile_valid[tile];
  const T *src = stage + (int64_t)blockIdx.x * tile_rows;
  T *dst = X + tile * g.tile_stride;
  for (int j = 0; j < g.d; j++)
    for (int r = threadIdx.x; r < valid; r += kBlock) dst[(int64_t)j * g.col_stride + r] = src[(int64_t)j * stage_ld + r];
}

template <typename T, int R, int U, bool NT, bool KARG, bool ROWDOTS>
void launch_tile(mhlik_ctx *c, const Store &s, const ThetaArg<T> &th, const T *theta_g, double *partials, T *rowdots)
{
  TileGeom g{s.tile_stride, s.col_stride, c->cfg.n_cols};
  snprintf(c->last_kernel, sizeof c->last_kernel, "lik_tile_kernel<%s, %d, %d, %s, %s, %s>", sizeof(T) == 8 ? "double" : "float",
           R, U, NT ? "true" : "false", KARG ? "true" : "false", ROWDOTS ? "true" : "false");
  hipLaunchKernelGGL((lik_tile_kernel<T, R, U, NT, KARG, ROWDOTS>), dim3((unsigned)s.ntiles), dim3(kBlock), 0,
                     c->stream, (const T *)s.X, g, s.tile_valid, th, theta_g, partials, rowdots);
}

template <typename T>
int dispatch_tile(mhlik_ctx *c, const Store &s, const ThetaArg<T> &th, const T *theta_g, bool karg, bool want_rowdots,
                  T *rowdots, double *P /* one partial per tile: the context's buffer or the static scratch */)
{
  if (want_rowdots) {
    // debug path: one fixed shape per R so the tile geometry matches the store
#define RD(Rv, Uv)                                                                         \
  if (s.R == Rv) {                                                                         \
    if (karg)                                                                              \
      launch_tile<T, Rv, Uv, true, true, true>(c, s, th, theta_g, P, rowdots);             \
    else                                                                                   \
      launch_tile<T, Rv, Uv, true, false, true>(c, s, th, theta_g, P, rowdots);            \
    return MHLIK_OK;                                                                       \
  }
    RD(1, 8) RD(2, 8) RD(4, 4) RD(8, 2)
#undef RD
    return fail(c, MHLIK_EINVAL, "no rowdots kernel for R=%d", s.R);
  }
#define V(Rv, Uv, NTv)                                                                     \
  if (s.R == Rv && s.U == Uv && s.nt == NTv) {                                             \
    if (karg)                                                                              \
      launch_tile<T, Rv, Uv, NTv, true, false>(c, s, th, theta_g, P, rowdots);             \
    else                                                                                   \
      launch_tile<T, Rv, Uv, NTv, false, false>(c, s, th, theta_g, P, rowdots);            \
    return MHLIK_OK;                                                                       \
  }
  V(1, 16, true) V(2, 8, true) V(4, 4, true) V(1, 8, true) V(2, 16, true) V(4, 8, true) V(1, 32, true)
  V(1, 16, false) V(2, 8, false) V(4, 4, false) V(2, 4, true) V(8, 2, true)
#undef V
  return fail(c, MHLIK_EINVAL, "no kernel for variant R=%d U=%d nt=%d", s.R, s.U, (int)s.nt);
}

int ensure_theta_staging(mhlik_ctx *c, size_t bytes)
{
  if (bytes <= c->theta_cap) return MHLIK_OK;
  if (c->theta_dev) (void)hipFree(c->theta_dev);
  if (c->theta_pin) (void)hipHostFree(c->theta_pin);
  c->theta_dev = nullptr;
  c->theta_pin = nullptr;
  c->theta_cap = 0;
  HIPCHK(c, hipMalloc((void **)&c->theta_dev, bytes));
  HIPCHK(c, hipHostMalloc(&c->theta_pin, bytes, hipHostMallocDefault));
  c->theta_cap = bytes;
  return MHLIK_OK;
}

// Wait for the completion word the last kernel of an evaluation writes into
// pinned host memory.  Polling a host word is a few microseconds cheaper per
// MH iteration than hipStreamSynchronize; the stream is still queried now and
// then so that a failed launch cannot spin forever.
int wait_published(mhlik_ctx *c, unsigned long long seq)
{
  volatile unsigned long long *flag = c->done_flag;
  for (unsigned long long spins = 0;; spins++) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return MHLIK_OK;
    if ((spins & 0xFFF) == 0xFFF) {
      hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return MHLIK_OK;
        return fail(c, MHLIK_EHIP, "stream drained but the completion word was not written");
      }
      if (q != hipErrorNotReady) return fail(c, MHLIK_EHIP, "stream error while waiting: %s", hipGetErrorString(q));
    }
  }
}

// Add the per-segment sums of `nvec` results (tile partials -> segment slots
// -> all shards -> host), leaving out[k] = sum over segments in index order.
int combine_launch(mhlik_ctx *c, const Store &s, int nvec, const double *partials = nullptr)
{
  if (!partials) partials = c->partials;
  const int nseg = c->nseg, nloc = (int)c->segs.size();
  const bool dist = c->comm != nullptr;
  const int jobs = nloc * nvec;
  // which reduction shape: fixed by the data set's geometry alone (tiles per
  // segment), never by the GPU count, so the bits stay world-invariant
  const int64_t global_tiles_per_seg = ((c->cfg.n_rows + nseg - 1) / nseg + s.tile_rows - 1) / s.tile_rows;
  const bool wide = global_tiles_per_seg > 8192;
  const bool one_group = !wide && jobs <= kSegWaves;
  const unsigned long long seq = ++c->done_seq;
  bool published = false;
  double *seg_out = dist ? c->slots_send : c->slots_host_dev;
  if (wide) {
    hipLaunchKernelGGL(segment_sum_wide_kernel, dim3(jobs), dim3(kSegBlock), 0, c->stream, partials,
                       (int64_t)s.ntiles, s.segs, nloc, seg_out, nseg);
  } else {
    const int grid = one_group ? 1 : (jobs + kSegWaves - 1) / kSegWaves;
    const bool flag = one_group && !dist;
    hipLaunchKernelGGL(segment_sum_kernel, dim3(grid), dim3(kSegBlock), 0, c->stream, partials, (int64_t)s.ntiles,
                       s.segs, nloc, nvec, seg_out, nseg, flag ? c->done_flag_dev : (unsigned long long *)nullptr,
                       flag ? seq : 0ull);
    published = flag;
  }
  if (dist) {
    int rc = g_rccl.AllReduce(c->slots_send, c->slots_recv, (size_t)nvec * nseg, kNcclFloat64, kNcclSum, c->comm,
                              c->stream);
    if (rc != 0) return fail(c, MHLIK_ECOMM, "ncclAllReduce failed: %s", g_rccl.GetErrorString(rc));
    hipLaunchKernelGGL(publish_slots_kernel, dim3(1), dim3(kWave), 0, c->stream, c->slots_recv, c->slots_host_dev,
                       nvec * nseg, c->done_flag_dev, seq);
    published = true;
  }
  HIPCHK(c, hipGetLastError());
  c->pending_nvec = nvec;
  c->pending_seq = seq;
  c->pending_published = published;
  return MHLIK_OK;
}

// Second half: wait for what combine_launch queued and add the slots on the host.
int combine_wait(mhlik_ctx *c, double *out)
{
  const int nseg = c->nseg, nvec = c->pending_nvec;
  if (c->pending_published) {
    int rc = wait_published(c, c->pending_seq);
    if (rc) return rc;
  } else {
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (c->cfg.timing) {
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
  }

  c->last_seg_sums.assign(c->slots_host, c->slots_host + nseg);
  for (int k = 0; k < nvec; k++) {
    double total = 0.0;
    for (int g = 0; g < nseg; g++) total += c->slots_host[(size_t)k * nseg + g];
    out[k] = total;
  }
  return MHLIK_OK;
}

int combine(mhlik_ctx *c, const Store &s, int nvec, double *out, const double *partials = nullptr)
{
  int rc = combine_launch(c, s, nvec, partials);
  return rc ? rc : combine_wait(c, out);
}

// Evaluation entry points refuse to start while a begun evaluation has not been collected: they share the
// context's staging buffers, partials and completion word.
#define NOT_PENDING(c)                                                                                              \
  do {                                                                                                              \
    if ((c)->pending_kind)                                                                                          \
      return fail((c), MHLIK_ESTATE, "an evaluation begun with mhlik_*_begin is pending: collect it with its _end call first"); \
  } while (0)

// `static_partials`: the allocation-free entry point passes this device's static scratch array; then nothing in
// here may allocate (wide theta would need the staging buffers, so the caller has rejected it).
template <typename T> int loglik_impl(mhlik_ctx *c, const T *theta, double *out, T *rowdots_dev,
                                      double *static_partials = nullptr)
{
  if (!c) return MHLIK_EINVAL;
  if (!theta || !out) return fail(c, MHLIK_EINVAL, "theta and out must not be NULL");
  NOT_PENDING(c);
  Store &s = store_of<T>(c);
  if (!s.has_data)
    return fail(c, MHLIK_ESTATE, "no %s design matrix on the device: upload or fill first",
                sizeof(T) == 8 ? "float64" : "float32");
  HIPCHK(c, hipSetDevice(c->device));
  const int d = c->cfg.n_cols;
  ThetaArg<T> th;
  const T *theta_g = nullptr;
  const bool karg = d <= kMaxKargCols;
  if (karg) {
    memset(&th, 0, sizeof th);
    memcpy(th.v, theta, sizeof(T) * d);
  } else {
    int rc = ensure_theta_staging(c, sizeof(T) * d);
    if (rc) return rc;
    memcpy(c->theta_pin, theta, sizeof(T) * d);
    HIPCHK(c, hipMemcpyAsync(c->theta_dev, c->theta_pin, sizeof(T) * d, hipMemcpyHostToDevice, c->stream));
    theta_g = (const T *)c->theta_dev;
  }
  if (c->cfg.timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  if (s.ntiles > 0) {
    int rc = dispatch_tile<T>(c, s, th, theta_g, karg, rowdots_dev != nullptr, rowdots_dev,
                              static_partials ? static_partials : c->partials);
    if (rc) return rc;
  }
  // the events bracket the streaming kernel alone (the roofline's kernel)
  if (c->cfg.timing) HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  c->last_bytes = c->local_rows * (int64_t)d * (int64_t)sizeof(T);
  return combine(c, s, 1, out, static_partials);
}

template <typename T> int upload_impl(mhlik_ctx *c, const T *X, int64_t ld, int64_t row0)
{
  if (!c) return MHLIK_EINVAL;
  NOT_PENDING(c);
  if (!X) return fail(c, MHLIK_EINVAL, "X must not be NULL");
  if (!(c->cfg.precision & (sizeof(T) == 8 ? MHLIK_F64 : MHLIK_F32)))
    return fail(c, MHLIK_EINVAL, "context was not created with this precision");
  if (row0 < 0 || row0 > c->local_begin)
    return fail(c, MHLIK_EINVAL, "row0=%lld must be in [0, first local row %lld]", (long long)row0,
                (long long)c->local_begin);
  if (ld < c->local_begin + c->local_rows - row0)
    return fail(c, MHLIK_EINVAL, "ld=%lld is smaller than the rows addressed (%lld)", (long long)ld,
                (long long)(c->local_begin + c->local_rows - row0));
  HIPCHK(c, hipSetDevice(c->device));
  int rc = build_store<T>(c);
  if (rc) return rc;
  Store &s = store_of<T>(c);
  const int d = c->cfg.n_cols;
  // Large strided pieces over PCIe into a column-major staging buffer (one
  // 2-D copy per chunk of rows: d pieces of chunk*sizeof(T) bytes), then a
  // device kernel moves the chunk into the tiled layout at HBM speed.  A
  // per-tile 2-D copy (d pieces of 4 KiB each) runs at < 0.5 GB/s.
  const int64_t chunk_tiles = s.ntiles < 2048 ? s.ntiles : 2048; // <= 1M rows (f64) per chunk
  T *stage = nullptr;
  if (chunk_tiles > 0) HIPCHK(c, hipMalloc((void **)&stage, sizeof(T) * (size_t)chunk_tiles * s.tile_rows * d));
  TileGeom g{s.tile_stride, s.col_stride, d};
  for (size_t si = 0; si < s.h_segs.size(); si++) {
    const SegDesc &sd = s.h_segs[si];
    const Segment &sg = c->segs[si];
    for (int64_t t0 = 0; t0 < sd.tile_count; t0 += chunk_tiles) {
      const int64_t nt = sd.tile_count - t0 < chunk_tiles ? sd.tile_count - t0 : chunk_tiles;
      const int64_t first = sg.row_begin + t0 * s.tile_rows; // global row of the chunk's first row
      int64_t rows = sg.rows - t0 * s.tile_rows;
      if (rows > nt * s.tile_rows) rows = nt * s.tile_rows;
      const int64_t stage_ld = nt * s.tile_rows;
      hipError_t e = hipMemcpy2DAsync(stage, (size_t)stage_ld * sizeof(T), X + (first - row0), (size_t)ld * sizeof(T),
                                      (size_t)rows * sizeof(T), (size_t)d, hipMemcpyHostToDevice, c->stream);
      if (e != hipSuccess) {
        (void)hipFree(stage);
        return fail(c, MHLIK_EHIP, "hipMemcpy2DAsync (upload) failed: %s", hipGetErrorString(e));
      }
      hipLaunchKernelGGL(retile_kernel<T>, dim3((unsigned)nt), dim3(kBlock), 0, c->stream, (const T *)stage, stage_ld,
                         (T *)s.X, g, sd.tile_begin + t0, s.tile_valid, s.tile_rows);
    }
  }
  hipError_t e2 = hipStreamSynchronize(c->stream);
  if (stage) (void)hipFree(stage);
  if (e2 != hipSuccess) return fail(c, MHLIK_EHIP, "upload failed: %s", hipGetErrorString(e2));
  HIPCHK(c, hipGetLastError());
  s.has_data = true;
  return MHLIK_OK;
}

template <typename T> int download_impl(mhlik_ctx *c, T *X, int64_t ld)
{
  if (!c) return MHLIK_EINVAL;
  NOT_PENDING(c);
  if (!X) return fail(c, MHLIK_EINVAL, "X must not be NULL");
  Store &s = store_of<T>(c);
  if (!s.has_data) return fail(c, MHLIK_ESTATE, "nothing to download");
  if (ld < c->local_rows) return fail(c, MHLIK_EINVAL, "ld smaller than the local row count");
  HIPCHK(c, hipSetDevice(c->device));
  const int d = c->cfg.n_cols;
  for (int64_t t = 0; t < s.ntiles; t++) {
    T *dst = X + (s.h_row0[t] - c->local_begin);
    const T *src = (const T *)s.X + tile_offset(s, t);
    HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)ld * sizeof(T), src, (size_t)s.col_stride * sizeof(T),
                               (size_t)s.h_valid[t] * sizeof(T), (size_t)d, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MHLIK_OK;
}

template <typename T> int fill_impl(mhlik_ctx *c, uint64_t seed)
{
  int rc = build_store<T>(c);
  if (rc) return rc;
  Store &s = store_of<T>(c);
  if (s.ntiles > 0) {
    TileGeom g{s.tile_stride, s.col_stride, c->cfg.n_cols};
    hipLaunchKernelGGL(synth_fill_kernel<T>, dim3((unsigned)s.ntiles), dim3(kBlock), 0, c->stream, (T *)s.X, g,
                       s.tile_valid, s.tile_row0, seed);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  s.has_data = true;
  return MHLIK_OK;
}

} // namespace

// ===================================================================== ABI

extern "C" {

const char *mhlik_version(void) { return "mhlik 0.1 (gfx950)"; }

const char *mhlik_last_error(const mhlik_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mhlik_create(const mhlik_config *cfg, mhlik_ctx **out)
{
  if (out) *out = nullptr;
  if (!cfg || !out) return fail(nullptr, MHLIK_EINVAL, "config and out must not be NULL");
  if (cfg->n_rows < 0 || cfg->n_cols < 1) return fail(nullptr, MHLIK_EINVAL, "need n_rows >= 0 and n_cols >= 1");
  if (!(cfg->precision & (MHLIK_F64 | MHLIK_F32)) || (cfg->precision & ~(MHLIK_F64 | MHLIK_F32)))
    return fail(nullptr, MHLIK_EINVAL, "precision must be MHLIK_F64, MHLIK_F32 or both");
  const int world = cfg->world > 0 ? cfg->world : 1;
  if (cfg->rank < 0 || cfg->rank >= world) return fail(nullptr, MHLIK_EINVAL, "rank %d outside world %d", cfg->rank, world);
  const int nseg = cfg->segments > 0 ? cfg->segments : 8;
  if (nseg > MHLIK_MAX_SEGMENTS || nseg % world != 0)
    return fail(nullptr, MHLIK_EINVAL, "segments=%d must be <= %d and a multiple of world=%d", nseg,
                MHLIK_MAX_SEGMENTS, world);
  if (cfg->variant < 0 || cfg->variant >= kNumVariants) return fail(nullptr, MHLIK_EINVAL, "unknown variant %d", cfg->variant);
  if (cfg->layout < 0 || cfg->layout > MHLIK_LAYOUT_COLMAJOR) return fail(nullptr, MHLIK_EINVAL, "unknown layout %d", cfg->layout);
  if (cfg->segment_align < 0) return fail(nullptr, MHLIK_EINVAL, "segment_align must be >= 0");

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return fail(nullptr, MHLIK_EHIP, "no HIP device available (%s): the MI355X path has no CPU fallback",
                hipGetErrorString(e));
  int dev = cfg->device;
  if (dev < 0) {
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(nullptr, MHLIK_EHIP, "hipGetDevice: %s", hipGetErrorString(e));
  }
  if (dev >= ndev) return fail(nullptr, MHLIK_EINVAL, "device %d of %d", dev, ndev);

  mhlik_ctx *c = new (std::nothrow) mhlik_ctx();
  if (!c) return fail(nullptr, MHLIK_ENOMEM, "out of host memory");
  c->cfg = *cfg;
  c->cfg.world = world;
  c->cfg.segments = nseg;
  c->device = dev;
  c->nseg = nseg;
  c->layout = cfg->layout == MHLIK_LAYOUT_DEFAULT ? MHLIK_LAYOUT_TILED : cfg->layout;

#define CREATE_CHK(expr)                                                                    \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      fail(nullptr, MHLIK_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));            \
      mhlik_destroy(c);                                                                     \
      return MHLIK_EHIP;                                                                    \
    }                                                                                       \
  } while (0)
  CREATE_CHK(hipSetDevice(dev));
  CREATE_CHK(scratch_lookup(dev));
  c->last_seg_sums.reserve(MHLIK_MAX_SEGMENTS); // so that no evaluation call grows it
  c->mp_last_slots.reserve(3 * MHLIK_MAX_SEGMENTS);
  hipDeviceProp_t prop;
  CREATE_CHK(hipGetDeviceProperties(&prop, dev));
  c->compute_units = prop.multiProcessorCount;
  if (const char *e = getenv("MHLIK_BATCHED_F32")) // "exact" | "split": default of mhlik_set_batched_f32_mode
    c->batched_f32_mode = strcmp(e, "exact") == 0 ? MHLIK_BATCHED_F32_EXACT : MHLIK_BATCHED_F32_SPLIT;
  if (const char *e = getenv("MHLIK_BATCHED_F64")) // "exact" | "fast": default of mhlik_set_batched_f64_mode
    c->batched_f64_mode = strcmp(e, "exact") == 0 ? MHLIK_BATCHED_F64_EXACT : MHLIK_BATCHED_F64_FAST;
  CREATE_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  CREATE_CHK(hipEventCreate(&c->ev0));
  CREATE_CHK(hipEventCreate(&c->ev1));

  // fixed row segments: the summation is grouped by them whatever the GPU count
  const int64_t N = cfg->n_rows;
  int64_t seg_rows = (N + nseg - 1) / nseg;
  if (cfg->segment_align > 1) seg_rows = (seg_rows + cfg->segment_align - 1) / cfg->segment_align * cfg->segment_align;
  c->seg_rows = seg_rows;
  const int per = nseg / world;
  for (int g = cfg->rank * per; g < (cfg->rank + 1) * per; g++) {
    Segment sg;
    sg.global_index = g;
    sg.row_begin = (int64_t)g * seg_rows < N ? (int64_t)g * seg_rows : N;
    int64_t end = (int64_t)(g + 1) * seg_rows < N ? (int64_t)(g + 1) * seg_rows : N;
    sg.rows = end - sg.row_begin;
    c->segs.push_back(sg);
  }
  c->local_begin = c->segs.front().row_begin;
  c->local_rows = 0;
  for (const Segment &sg : c->segs) c->local_rows += sg.rows;

  // room for the slots plus one completion word at the end
  const size_t slot_bytes = sizeof(double) * MHLIK_MAX_BATCH * nseg;
  CREATE_CHK(hipMalloc((void **)&c->slots_send, slot_bytes));
  CREATE_CHK(hipMalloc((void **)&c->slots_recv, slot_bytes));
  CREATE_CHK(hipMemset(c->slots_send, 0, slot_bytes));
  CREATE_CHK(hipMemset(c->slots_recv, 0, slot_bytes));
  CREATE_CHK(hipHostMalloc((void **)&c->slots_host, slot_bytes + 64, hipHostMallocMapped | hipHostMallocCoherent));
  memset(c->slots_host, 0, slot_bytes + 64);
  CREATE_CHK(hipHostGetDevicePointer((void **)&c->slots_host_dev, c->slots_host, 0));
  c->done_flag = (unsigned long long *)(c->slots_host + (size_t)MHLIK_MAX_BATCH * nseg);
  c->done_flag_dev = (unsigned long long *)(c->slots_host_dev + (size_t)MHLIK_MAX_BATCH * nseg);
#undef CREATE_CHK
  *out = c;
  return MHLIK_OK;
}

void mhlik_destroy(mhlik_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.ok) g_rccl.CommDestroy(c->comm);
  chain_free(c->chain);
  kchain_free(c->kchain);
  free_store(c->s64);
  free_store(c->s32);
  if (c->partials) (void)hipFree(c->partials);
  if (c->slots_send) (void)hipFree(c->slots_send);
  if (c->slots_recv) (void)hipFree(c->slots_recv);
  if (c->slots_host) (void)hipHostFree(c->slots_host);
  if (c->theta_dev) (void)hipFree(c->theta_dev);
  if (c->theta_pin) (void)hipHostFree(c->theta_pin);
  if (c->rowdots) (void)hipFree(c->rowdots);
  if (c->mp_z) (void)hipFree(c->mp_z);
  if (c->mp_out) (void)hipFree(c->mp_out);
  mpb_free(c);
  if (c->stats_buf) (void)hipFree(c->stats_buf);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int mhlik_local_rows(const mhlik_ctx *c, int64_t *begin, int64_t *count)
{
  if (!c) return MHLIK_EINVAL;
  if (begin) *begin = c->local_begin;
  if (count) *count = c->local_rows;
  return MHLIK_OK;
}

int mhlik_upload_f64(mhlik_ctx *c, const double *X, int64_t ld, int64_t row0) { return upload_impl<double>(c, X, ld, row0); }
int mhlik_upload_f32(mhlik_ctx *c, const float *X, int64_t ld, int64_t row0) { return upload_impl<float>(c, X, ld, row0); }
int mhlik_download_f64(mhlik_ctx *c, double *X, int64_t ld) { return download_impl<double>(c, X, ld); }
int mhlik_download_f32(mhlik_ctx *c, float *X, int64_t ld) { return download_impl<float>(c, X, ld); }

int mhlik_fill_synthetic(mhlik_ctx *c, uint64_t seed)
{
  if (!c) return MHLIK_EINVAL;
  NOT_PENDING(c);
  HIPCHK(c, hipSetDevice(c->device));
  if (c->cfg.precision & MHLIK_F64) {
    int rc = fill_impl<double>(c, seed);
    if (rc) return rc;
  }
  if (c->cfg.precision & MHLIK_F32) {
    int rc = fill_impl<float>(c, seed);
    if (rc) return rc;
  }
  return MHLIK_OK;
}

int mhlik_synthetic_host_f64(uint64_t seed, int64_t n_rows, int32_t n_cols, int64_t row0, int64_t rows, double *X,
                             int64_t ld)
{
  if (!X || n_cols < 1 || row0 < 0 || rows < 0 || row0 + rows > n_rows || ld < rows) return MHLIK_EINVAL;
  unsigned nt = std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > 16) nt = 16;
  if (rows < 65536) nt = 1;
  auto work = [=](int64_t lo, int64_t hi) {
    for (int64_t r = lo; r < hi; r++) {
      const int64_t row = row0 + r;
      const double y = (double)mhlik_synth_label(seed, row, n_cols);
      for (int32_t j = 0; j < n_cols; j++) X[(int64_t)j * ld + r] = -y * mhlik_synth_feature(seed, row, j);
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nt; t++) pool.emplace_back(work, rows * t / nt, rows * (t + 1) / nt);
  work(0, rows / nt);
  for (auto &th : pool) th.join();
  return MHLIK_OK;
}

int mhlik_synthetic_beta(uint64_t seed, int32_t n_cols, int32_t *beta)
{
  if (!beta || n_cols < 1) return MHLIK_EINVAL;
  for (int32_t j = 0; j < n_cols; j++) beta[j] = mhlik_synth_beta_j(seed, j);
  return MHLIK_OK;
}

int mhlik_loglik_f64(mhlik_ctx *c, const double *theta, double *out) { return loglik_impl<double>(c, theta, out, nullptr); }

// The double-precision evaluation with the tile partials in the static scratch array: the same kernels and the same
// bits as mhlik_loglik_f64, and no hipMalloc / hipFree / allocating routine on the way.
int mhlik_loglik_f64_noalloc(mhlik_ctx *c, const double *theta, double *out)
{
  if (!c) return MHLIK_EINVAL;
  if (c->comm) return fail(c, MHLIK_ESTATE, "mhlik_loglik_f64_noalloc: single-GPU contexts only (this one has an RCCL communicator)");
  if (c->cfg.n_cols > kMaxKargCols)
    return fail(c, MHLIK_EINVAL, "mhlik_loglik_f64_noalloc: n_cols=%d exceeds %d (wider theta needs a staging buffer)",
                c->cfg.n_cols, kMaxKargCols);
  if (c->s64.ntiles > MHLIK_NOALLOC_MAX_TILES)
    return fail(c, MHLIK_EINVAL, "mhlik_loglik_f64_noalloc: %lld row tiles exceed the static scratch (%d tiles of %d rows)",
                (long long)c->s64.ntiles, MHLIK_NOALLOC_MAX_TILES, c->s64.tile_rows);
  double *p = (c->device >= 0 && c->device < kMaxDevices) ? g_scratch_ptr[c->device] : nullptr;
  if (!p) return fail(c, MHLIK_ESTATE, "mhlik_loglik_f64_noalloc: no scratch address for device %d", c->device);
  return loglik_impl<double>(c, theta, out, nullptr, p);
}

int mhlik_loglik_f32(mhlik_ctx *c, const float *theta, float *out)
{
  if (!out) return c ? fail(c, MHLIK_EINVAL, "out must not be NULL") : MHLIK_EINVAL;
  double v = 0.0;
  int rc = loglik_impl<float>(c, theta, &v, nullptr);
  if (rc == MHLIK_OK) *out = (float)v;
  return rc;
}

int mhlik_rowdots_f64(mhlik_ctx *c, const double *theta, double *out)
{
  if (!c) return MHLIK_EINVAL;
  NOT_PENDING(c);
  if (!out) return fail(c, MHLIK_EINVAL, "out must not be NULL");
  Store &s = c->s64;
  if (!s.has_data) return fail(c, MHLIK_ESTATE, "no float64 design matrix on the device");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t need = sizeof(double) * (size_t)(s.ntiles > 0 ? s.ntiles : 1) * s.tile_rows;
  if (need > c->rowdots_cap) {
    if (c->rowdots) (void)hipFree(c->rowdots);
    c->rowdots = nullptr;
    c->rowdots_cap = 0;
    HIPCHK(c, hipMalloc((void **)&c->rowdots, need));
    c->rowdots_cap = need;
  }
  double L = 0.0;
  int rc = loglik_impl<double>(c, theta, &L, c->rowdots);
  if (rc) return rc;
  for (size_t i = 0; i < s.h_segs.size(); i++) {
    const SegDesc &sd = s.h_segs[i];
    const Segment &sg = c->segs[i];
    if (sg.rows == 0) continue;
    HIPCHK(c, hipMemcpy(out + (sg.row_begin - c->local_begin), c->rowdots + sd.tile_begin * s.tile_rows,
                        sizeof(double) * sg.rows, hipMemcpyDeviceToHost));
  }
  return MHLIK_OK;
}

int mhlik_last_segment_sums(const mhlik_ctx *c, double *seg_sums)
{
  if (!c || !seg_sums) return MHLIK_EINVAL;
  for (int g = 0; g < c->nseg; g++) seg_sums[g] = g < (int)c->last_seg_sums.size() ? c->last_seg_sums[g] : 0.0;
  return MHLIK_OK;
}

int mhlik_last_kernel_ms(const mhlik_ctx *c, float *ms, int64_t *bytes)
{
  if (!c) return MHLIK_EINVAL;
  if (ms) *ms = c->last_ms;
  if (bytes) *bytes = c->last_bytes;
  return MHLIK_OK;
}

const char *mhlik_last_kernel_name(const mhlik_ctx *c) { return c ? c->last_kernel : ""; }

int mhlik_get_info(const mhlik_ctx *c, mhlik_info *info)
{
  if (!c || !info) return MHLIK_EINVAL;
  memset(info, 0, sizeof *info);
  info->n_rows = c->cfg.n_rows;
  info->n_cols = c->cfg.n_cols;
  info->precision = c->cfg.precision;
  info->rank = c->cfg.rank;
  info->world = c->cfg.world;
  info->local_rows = c->local_rows;
  info->tiles_f64 = c->s64.ntiles;
  info->tiles_f32 = c->s32.ntiles;
  info->tile_rows_f64 = c->s64.tile_rows;
  info->tile_rows_f32 = c->s32.tile_rows;
  info->layout = c->layout;
  info->variant = c->cfg.variant ? c->cfg.variant : kDefaultVariant;
  info->compute_units = c->compute_units;
  info->hbm_bytes = (int64_t)c->hbm_bytes;
  return MHLIK_OK;
}

// ---------------------------------------------------------------- multi-GPU

int mhlik_comm_make_id(uint8_t id[MHLIK_COMM_ID_BYTES])
{
  std::string err;
  if (!id) return MHLIK_EINVAL;
  if (!rccl_load(err)) return fail(nullptr, MHLIK_ECOMM, "%s", err.c_str());
  ncclUniqueId uid;
  int rc = g_rccl.GetUniqueId(&uid);
  if (rc != 0) return fail(nullptr, MHLIK_ECOMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString(rc));
  memcpy(id, &uid, MHLIK_COMM_ID_BYTES);
  return MHLIK_OK;
}

int mhlik_comm_init(mhlik_ctx *c, const uint8_t id[MHLIK_COMM_ID_BYTES])
{
  if (!c) return MHLIK_EINVAL;
  if (!id) return fail(c, MHLIK_EINVAL, "id must not be NULL");
  // A one-rank communicator is pointless in production and skipped, unless
  // MHLIK_FORCE_COMM=1 asks for it: that lets a single-GPU box exercise the
  // whole RCCL path (dlopen, id, init, all-reduce on the stream, publish).
  const char *force = getenv("MHLIK_FORCE_COMM");
  if (c->cfg.world <= 1 && !(force && force[0] == '1')) return MHLIK_OK;
  std::string err;
  if (!rccl_load(err)) return fail(c, MHLIK_ECOMM, "%s", err.c_str());
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId uid;
  memcpy(&uid, id, MHLIK_COMM_ID_BYTES);
  int rc = g_rccl.CommInitRank(&c->comm, c->cfg.world, uid, c->cfg.rank);
  if (rc != 0) {
    c->comm = nullptr;
    return fail(c, MHLIK_ECOMM, "ncclCommInitRank: %s", g_rccl.GetErrorString(rc));
  }
  return MHLIK_OK;
}

int mhlik_comm_destroy(mhlik_ctx *c)
{
  if (!c) return MHLIK_EINVAL;
  if (!c->comm) return MHLIK_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int rc = g_rccl.ok ? g_rccl.CommDestroy(c->comm) : 0;
  c->comm = nullptr;
  // From here on an evaluation writes this rank's segments only: what the last all-reduce left in the other
  // segments' slots (device and pinned host copies) must not be added to it.
  const size_t slot_bytes = sizeof(double) * MHLIK_MAX_BATCH * c->nseg;
  memset(c->slots_host, 0, slot_bytes);
  HIPCHK(c, hipMemset(c->slots_send, 0, slot_bytes));
  HIPCHK(c, hipMemset(c->slots_recv, 0, slot_bytes));
  mpb_clear_slots(c);
  c->last_seg_sums.clear();
  c->mp_last_slots.clear();
  if (rc != 0) return fail(c, MHLIK_ECOMM, "ncclCommDestroy: %s", g_rccl.GetErrorString(rc));
  return MHLIK_OK;
}

int mhlik_comm_ranks(const mhlik_ctx *c, int32_t *ranks)
{
  if (!c || !ranks) return MHLIK_EINVAL;
  *ranks = 0;
  if (!c->comm || !g_rccl.ok) return MHLIK_OK;
  int n = 0;
  if (g_rccl.CommCount(c->comm, &n) != 0) return MHLIK_ECOMM;
  *ranks = n;
  return MHLIK_OK;
}

} // extern "C"

#include "mhlik_batched_abi.inc"
#include "mhlik_mp_abi.inc"
#include "mhlik_mp_batched_abi.inc"
#include "mhlik_chain_abi.inc"
#include "mhlik_kchain_abi.inc"
#include "mhlik_stats_abi.inc"
