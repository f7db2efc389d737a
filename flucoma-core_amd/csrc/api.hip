// api.hip -- the C ABI of libflucoma_hip.so (include/flucoma_hip.h) over the gfx950 kernels.
//
// Host-side control flow mirrors the reference call sites it replaces:
//   algorithm::STFT::process / magnitude   include/flucoma/algorithms/public/STFT.hpp:90-108,61-66
//   algorithm::NMF::process                include/flucoma/algorithms/public/NMF.hpp:91-134
//   NMF::multiplicativeUpdates             include/flucoma/algorithms/public/NMF.hpp:144-183
//   bufnmf::NMFClient::process write-back  include/flucoma/clients/nrt/NMFClient.hpp:277-300
// There is no CPU fallback anywhere in this file: every compute path launches HIP kernels and
// fails with FLUHIP_ERROR when the device is unusable.
#include "../../include/flucoma_hip.h"
#include "fluhip_kernels.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <random>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

using namespace fluhip;

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
struct ProfRec
{
  int cls;
  hipEvent_t start, stop;
};

struct fluhip_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copyStream = nullptr; // host -> device audio uploads run beside the compute stream (created on first use)
  std::string err;
  std::map<std::tuple<int64_t, int64_t, int>, double*> windows; // (win, fft, type) -> device table
  std::map<int, double*> twiddles;                // fft -> device table
  bool prof = false;
  std::vector<ProfRec> profRecs;
  std::vector<hipEvent_t> eventPool;
  hipDeviceProp_t props;
  int progressLag = 8;        // iterations the device may run ahead of the last progress report (fluhip_ctx_set_progress_lag)
  void* bigFft = nullptr;     // workspace of the global-memory FFT passes (fft > 8192), grown on demand
  size_t bigFftBytes = 0;
  void* stage[2] = {nullptr, nullptr}; // pinned staging blocks of large device -> host copies (copy_to_host)
  hipEvent_t stageEv[2] = {nullptr, nullptr};
};

static int fail(fluhip_ctx* ctx, const std::string& msg, int status = FLUHIP_ERROR);
// workspace for transforms whose frame does not fit the LDS; null (with the error set) when it cannot be had
static double* big_fft_scratch(fluhip_ctx* ctx, int64_t win, int64_t fft, int64_t frames)
{
  if (!stft_needs_scratch(win, fft)) return nullptr;
  const size_t need = (size_t) big_fft_scratch_bytes(fft, frames, nullptr);
  if (need > ctx->bigFftBytes)
  {
    (void) hipStreamSynchronize(ctx->stream);
    if (ctx->bigFft) (void) hipFree(ctx->bigFft);
    ctx->bigFft = nullptr;
    ctx->bigFftBytes = 0;
    if (hipMalloc(&ctx->bigFft, need) != hipSuccess) { fail(ctx, "out of device memory for the FFT workspace"); return nullptr; }
    ctx->bigFftBytes = need;
  }
  return static_cast<double*>(ctx->bigFft);
}

static int fail(fluhip_ctx* ctx, const std::string& msg, int status)
{
  if (ctx) ctx->err = msg;
  return status;
}

#define HIPCHK(ctx, expr)                                                                        \
  do                                                                                             \
  {                                                                                              \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess)                                                                       \
      return fail(ctx, std::string("HIP error: ") + hipGetErrorString(e__) + " in " #expr);      \
  } while (0)

// Large results to pageable host memory: a plain hipMemcpy stages them through the runtime's small pinned buffers (measured
// 4.7 - 5.3 GB/s: 85 - 95 ms for the 451 MB of an 8-channel x 32-component resynthesis).  Here: two pinned blocks of 8 MiB,
// the DMA of block i + 1 running while the host copies block i to its place.  `rows` rows of `width` bytes, source rows
// spitch and destination rows dpitch bytes apart (a contiguous copy: rows = 1).  Work queued on `s` before the call is
// complete when it returns.  Small copies take the plain path.
static int copy_to_host(fluhip_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows,
                        hipStream_t s)
{
  constexpr size_t kStage = (size_t) 8 << 20;
  const size_t total = width * rows;
  static const int off = [] { const char* e = fluhip::ab_getenv("FLUHIP_PINNED_D2H"); return e && std::atoi(e) == 0 ? 1 : 0; }();
  if (off || total < 2 * kStage || width > kStage)
  {
    HIPCHK(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return FLUHIP_OK;
  }
  for (int i = 0; i < 2; i++)
  {
    if (!ctx->stage[i]) HIPCHK(ctx, hipHostMalloc(&ctx->stage[i], kStage, hipHostMallocDefault));
    if (!ctx->stageEv[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->stageEv[i], hipEventDisableTiming));
  }
  // units: whole rows per block when there are several rows, byte ranges of the one row otherwise
  const bool byRows = rows > 1;
  const size_t unit = byRows ? width : 1;
  const size_t unitsPerBlock = kStage / unit;
  const size_t units = byRows ? rows : width;
  auto issue = [&](size_t u0, int slot) -> hipError_t {
    const size_t nu = std::min(unitsPerBlock, units - u0);
    hipError_t e = byRows ? hipMemcpy2DAsync(ctx->stage[slot], width, static_cast<const char*>(src) + u0 * spitch, spitch, width, nu,
                                             hipMemcpyDeviceToHost, s)
                          : hipMemcpyAsync(ctx->stage[slot], static_cast<const char*>(src) + u0, nu, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    return hipEventRecord(ctx->stageEv[slot], s);
  };
  size_t u = 0;
  int slot = 0;
  HIPCHK(ctx, issue(0, 0));
  while (u < units)
  {
    const size_t nu = std::min(unitsPerBlock, units - u);
    const size_t next = u + nu;
    if (next < units) HIPCHK(ctx, issue(next, slot ^ 1));
    HIPCHK(ctx, hipEventSynchronize(ctx->stageEv[slot]));
    if (byRows)
      for (size_t r = 0; r < nu; r++)
        std::memcpy(static_cast<char*>(dst) + (u + r) * dpitch, static_cast<const char*>(ctx->stage[slot]) + r * width, width);
    else
      std::memcpy(static_cast<char*>(dst) + u, ctx->stage[slot], nu);
    u = next;
    slot ^= 1;
  }
  return FLUHIP_OK;
}


// Device allocations go through a small caching pool: a BufNMF call allocates and frees a dozen buffers, and
// hipMalloc / the device-synchronising hipFree each time were a millisecond or two of a 2-15 ms call.  Freed blocks
// are kept per device (up to kPoolCap bytes) and handed out again to requests of about their size; a block is
// returned to the pool only after the stream it was used on has drained.  (HIP's own stream-ordered pool --
// hipMallocAsync -- was tried first and returned corrupted tails of result buffers from the third call of a
// shape on; not pursued.)  FLUHIP_NO_POOL=1 goes back to plain hipMalloc / hipFree.
struct BlockPool
{
  static constexpr size_t kPoolCap = (size_t) 8 << 30;
  std::mutex m;
  std::multimap<size_t, void*> freeBlocks[16];
  size_t cached[16] = {}; // per device
  static bool enabled()
  {
    static const bool on = [] { const char* e = std::getenv("FLUHIP_NO_POOL"); return !(e && std::atoi(e)); }();
    return on;
  }
  void* take(int dev, size_t n, size_t* got)
  {
    std::lock_guard<std::mutex> g(m);
    auto& f = freeBlocks[dev & 15];
    auto it = f.lower_bound(n);
    if (it == f.end() || it->first > 2 * n + ((size_t) 1 << 20)) return nullptr;
    void* p = it->second;
    *got = it->first;
    cached[dev & 15] -= it->first;
    f.erase(it);
    return p;
  }
  bool give(int dev, size_t n, void* p)
  {
    std::lock_guard<std::mutex> g(m);
    if (cached[dev & 15] + n > kPoolCap) return false;
    freeBlocks[dev & 15].emplace(n, p);
    cached[dev & 15] += n;
    return true;
  }
  void trim(int dev)
  {
    std::lock_guard<std::mutex> g(m);
    for (auto& kv : freeBlocks[dev & 15]) { (void) hipFree(kv.second); cached[dev & 15] -= kv.first; }
    freeBlocks[dev & 15].clear();
  }
};
static BlockPool g_pool;

struct DevBuf
{
  void* p = nullptr;
  size_t bytes = 0;     // requested
  size_t capacity = 0;  // of the block behind it
  int dev = 0;
  hipStream_t owner = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  // FLUHIP_CANARY=1 (debugging): every buffer gets a 64 KiB guard band behind the requested bytes, filled with a
  // pattern at allocation and checked when the buffer is released; a kernel that writes past its buffer aborts
  // the process with the size of the buffer it trampled.
  static bool canary()
  {
    static const bool on = [] { const char* e = std::getenv("FLUHIP_CANARY"); return e && std::atoi(e); }();
    return on;
  }
  void check_canary()
  {
    if (!canary() || !p) return;
    std::vector<unsigned char> h(65536);
    (void) hipStreamSynchronize(owner);
    if (hipMemcpy(h.data(), static_cast<char*>(p) + bytes, 65536, hipMemcpyDeviceToHost) != hipSuccess) return;
    for (size_t i = 0; i < h.size(); i++)
      if (h[i] != 0xA5)
      {
        std::fprintf(stderr, "fluhip: write past the end of a %zu-byte device buffer (offset +%zu)\n", bytes, i);
        std::abort();
      }
  }
  void release()
  {
    if (p)
    {
      check_canary();
      bool kept = false;
      if (BlockPool::enabled() && hipStreamSynchronize(owner) == hipSuccess) kept = g_pool.give(dev, capacity, p);
      if (!kept) (void) hipFree(p);
    }
    p = nullptr;
    bytes = capacity = 0;
  }
  hipError_t alloc(size_t n, bool zero, hipStream_t s)
  {
    release();
    if (n == 0) n = 16;
    const size_t want = ((n + 65535) & ~(size_t) 65535) + (canary() ? 131072 : 0); // 64 KiB granules: near-equal requests share blocks
    (void) hipGetDevice(&dev);
    hipError_t e = hipSuccess;
    capacity = want;
    if (BlockPool::enabled()) p = g_pool.take(dev, want, &capacity);
    if (!p)
    {
      e = hipMalloc(&p, want);
      if (e != hipSuccess && BlockPool::enabled())
      {
        g_pool.trim(dev); // the cache may be what stands in the way
        e = hipMalloc(&p, want);
      }
      if (e != hipSuccess) { p = nullptr; return e; }
    }
    bytes = n;
    owner = s;
    if (canary()) (void) hipMemsetAsync(static_cast<char*>(p) + n, 0xA5, 65536, s);
    if (zero) e = hipMemsetAsync(p, 0, n, s);
    return e;
  }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};

static hipEvent_t take_event(fluhip_ctx* ctx)
{
  if (!ctx->eventPool.empty())
  {
    hipEvent_t e = ctx->eventPool.back();
    ctx->eventPool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void) hipEventCreate(&e);
  return e;
}

struct ProfScope
{
  fluhip_ctx* ctx;
  ProfRec rec;
  bool on;
  ProfScope(fluhip_ctx* c, int cls) : ctx(c), on(c->prof)
  {
    if (!on) return;
    rec.cls = cls;
    rec.start = take_event(ctx);
    rec.stop = take_event(ctx);
    (void) hipEventRecord(rec.start, ctx->stream);
  }
  ~ProfScope()
  {
    if (!on) return;
    (void) hipEventRecord(rec.stop, ctx->stream);
    ctx->profRecs.push_back(rec);
  }
};

// Strided host view -> contiguous device copy (clients/nrt/NMFClient.hpp:240 `tmp <<= samps(...)`).
// A contiguous source is one plain copy; a strided one (a channel of a frame-interleaved host buffer) is gathered
// on the host first -- a 2-D copy with element-sized rows would be issued row by row.
static hipError_t upload_strided(void* dst, const void* src, size_t n, size_t stride, size_t esz, hipStream_t s)
{
  if (stride == 1)
  {
    hipError_t e = hipMemcpyAsync(dst, src, n * esz, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);
  }
  std::vector<char> tmp(n * esz);
  const char* p = static_cast<const char*>(src);
  if (esz == 4)
    for (size_t i = 0; i < n; i++) reinterpret_cast<float*>(tmp.data())[i] = reinterpret_cast<const float*>(p)[i * stride];
  else
    for (size_t i = 0; i < n; i++) reinterpret_cast<double*>(tmp.data())[i] = reinterpret_cast<const double*>(p)[i * stride];
  hipError_t e = hipMemcpyAsync(dst, tmp.data(), n * esz, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s); // tmp goes out of scope
}

// ---------------------------------------------------------------------------------------
// tables: window (alg/WindowFuncs.hpp:38-72) and FFT twiddles, computed on the host in f64
// ---------------------------------------------------------------------------------------
static bool make_window(int type, int64_t size, std::vector<double>& out)
{
  const double pi = M_PI; // util/AlgorithmUtils.hpp:21
  out.resize((size_t) size);
  switch (type)
  {
  case FLUHIP_WINDOW_HANN: // alg/WindowFuncs.hpp:41-45
    for (int64_t i = 0; i < size; i++) out[(size_t) i] = 0.5 - 0.5 * std::cos((pi * 2 * i) / size);
    return true;
  case FLUHIP_WINDOW_HANND: // :46-51
  {
    double norm = pi / size;
    for (int64_t i = 0; i < size; i++) out[(size_t) i] = norm * std::sin((2 * pi * i) / size);
    return true;
  }
  case FLUHIP_WINDOW_HAMMING: // :52-56
    for (int64_t i = 0; i < size; i++) out[(size_t) i] = 0.54 - 0.46 * std::cos((pi * 2 * i) / size);
    return true;
  case FLUHIP_WINDOW_BLACKMANHARRIS: // :57-65 (all three cosines share one argument, as written there)
    for (int64_t i = 0; i < size; i++)
      out[(size_t) i] = 0.35875 - 0.48829 * std::cos((pi * 2 * i) / size) +
                        0.14128 * std::cos((pi * 2 * i) / size) +
                        0.01168 * std::cos((pi * 2 * i) / size);
    return true;
  case FLUHIP_WINDOW_GAUSSIAN: // :66-72 (requires odd size; sigma = size / 3 in integer arithmetic)
  {
    if (size % 2 == 0) return false;
    double sigma = (double) (size / 3);
    int64_t h = (size - 1) / 2;
    for (int64_t i = -h; i <= h; i++) out[(size_t) (i + h)] = std::exp(-i * i / (2 * sigma * sigma));
    return true;
  }
  default: return false;
  }
}

// device table of `fft` doubles: the window followed by zeros (a frame shorter than the transform
// is zero-padded at its tail, util/FFT.hpp:97-98)
static int get_window(fluhip_ctx* ctx, int64_t win, int64_t fft, int type, const double** out)
{
  auto key = std::make_tuple(win, fft, type);
  auto it = ctx->windows.find(key);
  if (it == ctx->windows.end())
  {
    std::vector<double> w;
    if (!make_window(type, win, w)) return fail(ctx, "unsupported window type / size");
    w.resize((size_t) std::max(win, fft), 0.0);
    double* d = nullptr;
    HIPCHK(ctx, hipMalloc(&d, w.size() * sizeof(double)));
    HIPCHK(ctx, hipMemcpy(d, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
    it = ctx->windows.emplace(key, d).first;
  }
  *out = it->second;
  return FLUHIP_OK;
}

static int get_twiddle(fluhip_ctx* ctx, int64_t fft, const double** out)
{
  auto it = ctx->twiddles.find((int) fft);
  if (it == ctx->twiddles.end())
  {
    const size_t nc = (size_t) fft / 2;
    std::vector<double> t(2 * nc);
    for (size_t j = 0; j < nc; j++)
    {
      const double ang = -2.0 * M_PI * (double) j / (double) fft;
      t[2 * j] = std::cos(ang);
      t[2 * j + 1] = std::sin(ang);
    }
    double* d = nullptr;
    HIPCHK(ctx, hipMalloc(&d, std::max<size_t>(16, t.size() * sizeof(double))));
    HIPCHK(ctx, hipMemcpy(d, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    it = ctx->twiddles.emplace((int) fft, d).first;
  }
  *out = it->second;
  return FLUHIP_OK;
}

// ---------------------------------------------------------------------------------------
// corpus
// ---------------------------------------------------------------------------------------
struct fluhip_corpus
{
  fluhip_ctx* ctx = nullptr;
  int64_t B = 0, n = 0, win = 0, fft = 0, hop = 0, K = 0;
  int64_t T = 0, F = 0, Tp = 0, Fp = 0, Kp = 0;
  int windowType = FLUHIP_WINDOW_HANN;
  bool keepSpec = false;
  const float* audioDev = nullptr; // borrowed or owned (audioOwn)
  DevBuf audioOwn, mag, magT, Wf, H1, spec, part, dpart, stage, hmax, normScratch;
  int nsplitW = 1, nsplitH = 1;
  // H update in two launches (plan_tail): the first tailStripsH strips of every buffer (tailColsH frames) as whole
  // contractions, the rest (tailRestH strips) with the contraction cut into tailSplitH pieces; 0 = one launch
  int tailSplitH = 0, tailStripsH = 0, tailRestH = 0, tailColsH = 0;
  // deferred normalisation of W inside the iteration loop (fluhip_kernels.h UpdateArgs::nrm)
  bool lazy = false;     // the shape takes the two-launch-per-factor fast path
  bool sideW = false;    // ... with the Nyquist bin of the W update as a side column
  bool wPending = false; // W in memory is W' = W diag(wnorm)
  int stripsW = 0;       // wavefronts per buffer of the W update (statistics partials)
  DevBuf wnorm, wscratch, csumScratch, wideScratch;
  DevBuf clk; // UpdateArgs::clk: 4 words for the W update's launches, 4 for the H update's
  // frame-strip schedule of a single large buffer at rank <= 16 (kernels_nmf_strip.hip)
  bool strip = false;
  bool stripReady = false;     // the numerator partials of the next W update are in stripPart
  bool stripNormFresh = false; // wnorm holds the column norms of the W' in memory
  bool stripStatsValid = false; // the column-statistics records of generation stripGen describe the W in memory
  int stripGen = 0;
  DevBuf stripPart;
  bool haveMag = false, haveFactors = false;
  bool touched = false; // work that reads the audio has been enqueued on the compute stream
  // Seed / Fixed factors of the batched form (fluhip_corpus_set_factors): host copies, [B][K][F] and [B][K][T] floats
  std::vector<float> seedW32, seedH32;
  // ragged corpus (fluhip_corpus_create_ragged): buffers of different lengths in ONE set of launches.  n / T are those of
  // the longest buffer (the strides of every array); frames past a buffer's own count are zero padding that stays zero.
  // The factor updates run kernels_nmf5.hip in work-list mode: one WaveDesc per wavefront, dealt by work.
  bool ragged = false;
  bool useLists = false; // the factor updates run from work lists (ragged corpora; small equal-length ones)
  // window of buffers the next enqueue_iteration works on (0 buffers = all): corpora of several rounds of wavefronts
  // run their iterations round by round (corpus_iterate_loop)
  int64_t winB0 = 0, winB = 0;
  int winStripsW = 0, winStripsH = 0;
  std::vector<int64_t> nOf; // samples per buffer
  std::vector<int> tOf;     // frames per buffer
  DevBuf nTab, tTab;        // the same on the device
  struct WorkList
  {
    DevBuf list, splitTab;
    int wgs = 0, ng = 0, partial = 0, maxSplit = 1;
    int64_t nPartials = 0;
    int statParts = 0; // column-statistics parts per buffer (W update)
  } listW, listH;
  int64_t device_bytes() const
  {
    return (int64_t) (audioOwn.bytes + mag.bytes + magT.bytes + Wf.bytes + H1.bytes + spec.bytes +
                      part.bytes + dpart.bytes + stage.bytes + hmax.bytes + normScratch.bytes);
  }
};

// which factor-update path runs:
//   5 = v_mfma_f64_4x4x4_4b + LDS-DMA operand streaming (kernels_nmf5.hip): every rank up to 128, padded to 16 / 32 / 64 / 128
//   0 = un-fused, over a materialised ratio matrix (kernels_nmf_wide.hip): any rank, used above 128
//       (FLUHIP_NMF_KERNEL=-1 forces it: an independent second implementation for the tests)
static int update_variant(int Kp)
{
  if (Kp > 128) return 0;
  static const int forced = [] {
    const char* e = fluhip::ab_getenv("FLUHIP_NMF_KERNEL");
    return e ? std::atoi(e) : 0;
  }();
  if (forced == -1) return 0;
  return 5;
}
// padded rank: the 4x4x4 kernel is built for 16 / 32 / 64 / 128 (components up to the padded rank are zero and stay zero);
// above 128 the any-rank path takes multiples of 16
static int64_t padded_rank(int64_t K)
{
  if (K <= 16) return 16;
  if (K <= 32) return 32;
  if (K <= 64) return 64;
  if (K <= 128) return 128;
  return round_up(K, 16);
}

static int choose_split4(int64_t B, int C, int R, int Kp)
{
  static const int forceS = [] { const char* e = fluhip::ab_getenv("FLUHIP_PLAN_SPLIT"); return e ? std::atoi(e) : 0; }();
  const int64_t nSteps = (R + 3) / 4;
  const int64_t smax = std::max<int64_t>(1, std::min<int64_t>(64, nSteps / 12)); // >= 12 steps per wavefront
  if (forceS > 0) return (int) std::min<int64_t>(std::min<int64_t>(forceS, 64), std::max<int64_t>(1, nSteps / 2)); // the finalize kernel sums at most 64 splits
  const int64_t waves = B * nmf_update5_waves_per_buffer(C, Kp, (int) B);
  if (waves >= 768) return 1;
  // one wavefront per SIMD (the kernel's register footprint allows no more): never exceed 1024 in
  // total, a 1025th wavefront would wait for a whole pass of the others
  const int64_t s = 1024 / waves;
  return (int) std::max<int64_t>(1, std::min(s, smax));
}

// workspaces of the factor updates: split-contraction partials, denominators, column-sum pre-pass
static int alloc_update_scratch(fluhip_ctx* ctx, fluhip_corpus* c)
{
  hipStream_t s = ctx->stream;
  const size_t B = (size_t) c->B;
  const int ns = std::max(std::max(c->nsplitW, c->nsplitH), c->tailSplitH);
  if (ns > 1 && !c->strip)
  {
    // partial numerators [B][pieces][rows][Kp]: the W update's rows are the bins, the H update's the frames (of the tail launch:
    // the frames behind the whole-contraction ones); the launches set UpdateArgs::Cp to their own row count
    const size_t rowsW = c->nsplitW > 1 ? (size_t) c->nsplitW * c->Fp : 0;
    const size_t rowsH = c->nsplitH > 1 ? (size_t) c->nsplitH * c->Tp : 0;
    const size_t rowsT = c->tailSplitH > 1 ? (size_t) c->tailSplitH * round_up(c->T - c->tailColsH, 32) : 0;
    HIPCHK(ctx, c->part.alloc(B * std::max(std::max(rowsW, rowsH), rowsT) * c->Kp * sizeof(double), true, s));
  }
  // (the tail launch keeps its denominator slots behind the first launch's B x Kp)
  HIPCHK(ctx, c->dpart.alloc(std::max<size_t>(256, B * std::max(ns, 1 + c->tailSplitH) * c->Kp * sizeof(double)), true, s));
  if (c->Kp > 64)
    HIPCHK(ctx, c->csumScratch.alloc((size_t) colsum_scratch_doubles((int) std::max(c->T, c->F), (int) c->Kp, (int) B) *
                                         sizeof(double), false, s));
  return FLUHIP_OK;
}

// A whole-contraction update whose wavefronts need a last, poorly filled round of the 1024 SIMDs (config 3's H update: 2 x 808
// strips = 1.58 rounds, paid as 2) goes out as TWO launches: the strips that fill whole rounds as before, then the remaining
// strips with their contraction cut into `split` pieces (uniform split schedule: partials + finalize), which deals the tail
// over the chip in short rounds -- 513 + 2 x 171 steps instead of 2 x 513.  Costs in 4-row steps of this strip width:
// ~20 k cycles of prologue + epilogue per wavefront, partials written and read back at ~3 TB/s plus the finalize launch.
// Returns the split (0: one launch pays) and the strips per buffer of the first launch.  FLUHIP_TAIL_SPLIT: 0 off, n forces n
// pieces; FLUHIP_TAIL_SLOTS: the wavefronts of a round (tests: small corpora take the path).
static int plan_tail(int64_t B, int C, int R, int Kp, int* stripsA)
{
  static const int forceS = [] { const char* e = fluhip::ab_getenv("FLUHIP_TAIL_SPLIT"); return e ? std::atoi(e) : -1; }();
  static const int slotsEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_TAIL_SLOTS"); return e ? std::atoi(e) : 0; }();
  *stripsA = 0;
  if (forceS == 0) return 0;
  const int64_t slots = slotsEnv > 0 ? slotsEnv : 1024;
  const int G = (C + 15) / 16;
  const int w = nmf_update5_strips(C, Kp, (int) B);
  const int ng = (G + w - 1) / w;
  const int64_t total = B * w;
  const int64_t roundsA = total / slots;
  if (roundsA < 1 || roundsA > 3 || total % slots == 0) return 0;
  // (a workgroup is four wavefronts of ONE buffer and one workgroup fills a CU: strips are dealt in fours, and a launch
  //  occupies round_up(strips, 4) slots per buffer -- 200 buffers x 5 strips are 400 workgroups, two rounds, not 1000 slots)
  int wA = (int) std::min<int64_t>(w - 1, roundsA * slots / B);
  wA -= wA % 4;
  if (wA < 4) return 0;
  const int64_t nRest = B * round_up(w - wA, 4);
  const int64_t nSteps = (R + 3) / 4;
  const double cycles = 400.0 + 270.0 * ng * (Kp / 32.0); // per step (measured: rank 32 400 + 270 NG; rank 128, 2 groups 2 376)
  const double ovh = 20000.0 / cycles, stepUs = cycles / 2300.0;
  const double cur = (double) ((B * round_up(w, 4) + slots - 1) / slots) * (nSteps + ovh);
  const int64_t restCols = C - (int64_t) wA * ng * 16;
  int best = 0;
  double bestCost = 0.93 * cur; // a clear win only
  const int smax = (int) std::min<int64_t>(8, nSteps / 12);
  for (int sp = 2; sp <= smax; sp++)
  {
    if (forceS > 0 && sp != std::min(forceS, smax)) continue;
    const double finUs = 2.0 * sp * B * restCols * Kp * 8.0 / 3.0e6 + 8.0;
    const double cost = (double) ((B * wA + slots - 1) / slots) * (nSteps + ovh) +
                        (double) ((nRest * sp + slots - 1) / slots) * ((nSteps + sp - 1) / sp + ovh) + finUs / stepUs;
    // (more pieces only for a clear gain: they cost memory, and config 3 measures the same with 3 and 5)
    if ((best == 0 ? cost < bestCost : cost < 0.97 * bestCost) || (forceS > 0 && best == 0)) { best = sp; bestCost = cost; }
  }
  if (best) *stripsA = wA;
  return best;
}

static int plan_lists(fluhip_ctx* ctx, fluhip_corpus* c);
// where the work-list form beats the uniform split schedule for equal-length corpora (tools/batch_timing.py with
// FLUHIP_LIST_PLAN=0|1, profiles/r03/small_batches_*.jsonl)
struct PlanShape { int64_t B, T, F, Kp; }; // what the choice of schedule depends on (also reachable without a corpus: fluhip_debug_plan_kind)
static bool list_plan_pays(const PlanShape* c)
{
  // measured at rank 32, 10 s buffers (us per iteration, uniform schedule -> lists): 1 buffer 47 -> 51, 2: 53 -> 57, 4: 73 -> 65,
  // 8: 92 -> 73, 16: 135 -> 109, 24: 196 -> 168, 32: 250 -> 202, 48: 346 -> 289, 64: 370 -> 345, 96: 758 -> 511, 112: 878 -> 564,
  // 128: 619 = 621 (the same schedule either way); ranks 16 / 64 / 128 at 4 and 16 buffers likewise (49 -> 49, 89 -> 76;
  // 130 -> 98, 227 -> 193; 211 -> 160, 486 -> 476).  Between one and two rounds (profiles/r03/midsize_ab.txt, two rounds of
  // the uniform schedule = ~1 250 us): 144 buffers 729, 176: 854, 200: 1 007, 232: 1 102, 250: 1 153.  So: from three buffers on,
  // while whole contractions at the widest strips do not fill the chip in whole rounds.
  const int maxNG = nmf_update5_max_groups((int) c->Kp);
  const int G = ((int) c->F - 1 + 15) / 16;
  const int64_t w0 = c->B * ((G + maxNG - 1) / maxNG);
  // Beyond one round the lists win as well, whole rounds or not (profiles/r03/midsize_ab.txt parts 3 - 4, uniform schedule in
  // round-major windows -> lists: 288 buffers 1 723 -> 1 475 us, 400: 2 300 -> 2 015, 520: 2 864 -> 2 589, 900: 4 568 -> 4 320,
  // 1000: 4 586 -> 4 424; exact multiples of a round 256: 1 222 -> 1 206, 512: 2 346 -> 2 285, 1024: 4 611 -> 4 485, and with a
  // progress callback -- iteration-major launches of several rounds -- 512: 2 427 -> 2 234, 1024: 5 157 -> 4 452): several
  // thousand workgroups handed out as CUs free up keep the chip busy across what the uniform launch runs as lock-step rounds.
  // Exactly one round (the bench shard: 128 buffers) is the same schedule either way and stays on the uniform kernel.
  // (Rank 128 apart: the list kernel's instantiation there -- grouped refill, column sums in the kernel -- runs 3 580 cycles per
  //  step against the uniform one's 2 376.)
  // Rank 128 (profiles/r03/plan_regimes.txt): 4 buffers 211 -> 160 on the lists, but 16: 452 -> 500, 24: 581 -> 820, 32: 639 -> 818,
  // 40: 902 (uniform, two-launch H update) -> 1 330 -- lists only while whole contractions fill less than half a round.
  // (few buffers, 10 s each: 3: 177 -> 167, 4: 168 -> 163, 8: 240 -> 273, 12: 516 -> 405; long ones lose: 4 x 60 s 558 -> 753)
  if (c->Kp > 64) return c->B >= 3 && w0 < 512 && c->T <= 2048;
  // Rank 64 follows rank 32 (100 x 10 s 1 333 -> 1 043, 300: 2 999 -> 2 771, 128 x 2 s 330 -> 317, 1 x 60 s 131 -> 106, 1 x 300 s
  // 417 -> 263, 2 x 300 s 800 -> 614; 40 x 10 s 456 -> 473 the one loss).
  // One or two buffers (rank <= 64; a single buffer of rank <= 16 that fits the frame-strip schedule never gets here): the
  // uniform split schedule up to ~45 s of frames in all (1 x 30 s 62.5 us per iteration against 66.7, 2 x 10 s 53 against 57),
  // lists beyond (1 x 60 s 80.5 -> 72.8, 1 x 300 s 253 -> 163, 2 x 30 s 79.3 -> 71.3, 2 x 300 s 454 -> 340).
  if (c->B <= 2) return c->B * c->T * c->F >= 4200000;
  return w0 != 1024;
}

// how the factor updates of this shape are scheduled (splits, deferred normalisation, side column) and their
// workspaces; needs B, T, F, Tp, Fp, Kp
static int plan_updates(fluhip_ctx* ctx, fluhip_corpus* c)
{
  hipStream_t s = ctx->stream;
  const size_t B = (size_t) c->B;
  c->tailSplitH = 0;
  if (update_variant((int) c->Kp) == 0)
  {
    c->nsplitW = c->nsplitH = 1;
    c->lazy = c->sideW = false;
    const int64_t nd = std::max(nmf_update_wide_scratch_doubles((int) c->T, (int) c->F, (int) c->Kp, (int) B),
                                nmf_update_wide_scratch_doubles((int) c->F, (int) c->T, (int) c->Kp, (int) B));
    HIPCHK(ctx, c->wideScratch.alloc((size_t) nd * sizeof(double), false, s));
  }
  else
  {
    c->nsplitW = choose_split4(c->B, (int) c->F, (int) c->T, (int) c->Kp);
    c->nsplitH = choose_split4(c->B, (int) c->T, (int) c->F, (int) c->Kp);
    // Fast path: W stays un-normalised in memory during the loop (UpdateArgs::nrm), the column statistics
    // come out of the update kernel's epilogue and the Nyquist bin is a side column when that shortens the
    // widest strip of the MFMA kernel (fluhip_kernels.h SideColumn).
    static const int lazyOff = [] { const char* e = fluhip::ab_getenv("FLUHIP_NO_LAZY"); return e ? std::atoi(e) : 0; }();
    static const int sideOff = [] { const char* e = fluhip::ab_getenv("FLUHIP_NO_SIDE"); return e ? std::atoi(e) : 0; }();
    c->lazy = !lazyOff && update_variant((int) c->Kp) == 5;
    c->sideW = false;
    if (c->lazy && !sideOff && c->nsplitW == 1 && nmf_side_column_supported((int) c->T, (int) c->F, (int) c->Kp) &&
        choose_split4(c->B, (int) c->F - 1, (int) c->T, (int) c->Kp) == 1)
    {
      const int G = ((int) c->F + 15) / 16, G1 = G - 1;
      const int w = nmf_update5_strips((int) c->F, (int) c->Kp, (int) c->B);
      const int w1 = nmf_update5_strips((int) c->F - 1, (int) c->Kp, (int) c->B);
      // worth it when the widest strip gets shorter, or when the launch needs fewer passes over the 1024 SIMDs
      const int64_t passes = (c->B * w + 1023) / 1024, passes1 = (c->B * w1 + 1023) / 1024;
      c->sideW = w1 <= w && ((G1 + w1 - 1) / w1 < (G + w - 1) / w || passes1 < passes);
    }
    else if (c->lazy && !sideOff && c->nsplitW > 1 && nmf_side_column_supported((int) c->T, (int) c->F, (int) c->Kp))
    {
      // Split contraction (few buffers): without the 16 m + 1-th bin the strips deal evenly and the pieces get shorter --
      // config 3 (2 x 2049 bins, rank 128): 130 strips x 7 pieces of 923 steps (910 wavefronts) -> 128 strips x 8 pieces
      // of 808 steps (1024 wavefronts).  Taken when the longest piece shrinks.
      const int s1 = choose_split4(c->B, (int) c->F - 1, (int) c->T, (int) c->Kp);
      const int64_t nSteps = (c->T + 3) / 4;
      const int64_t w = c->B * nmf_update5_waves_per_buffer((int) c->F, (int) c->Kp, (int) c->B);
      const int64_t w1 = c->B * nmf_update5_waves_per_buffer((int) c->F - 1, (int) c->Kp, (int) c->B);
      if (s1 > 1 && w1 * s1 <= 1024 && w * c->nsplitW <= 1024 && (nSteps + s1 - 1) / s1 < (nSteps + c->nsplitW - 1) / c->nsplitW)
      {
        c->sideW = true;
        c->nsplitW = s1;
      }
    }
    // A single buffer of rank <= 16 runs the frame-strip schedule while one round of workgroups covers it (at most 6 frame
    // quads per CU: 71 s at hop 512): two launches per iteration instead of five and V read once.  Longer buffers and
    // batches stay with the split / batched kernels, which win there (tools/strip_vs_split.py).  FLUHIP_STRIP=0 off,
    // =1 wherever the kernel supports the shape.
    static const int stripEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_STRIP"); return e ? std::atoi(e) : -1; }();
    c->strip = c->lazy && stripEnv != 0 && nmf_strip_supported((int) c->F, (int) c->T, (int) c->Kp) &&
               (stripEnv == 1 || (c->B == 1 && nmf_strip_workgroups((int) c->T) <= 512));
    if (c->strip)
    {
      c->sideW = false;
      HIPCHK(ctx, c->stripPart.alloc((size_t) nmf_strip_part_doubles((int) c->F, (int) c->T, (int) B) * sizeof(double), false, s));
    }
    // Equal-length corpora too small to fill the chip with whole contractions: the work-list form (plan_lists) instead of
    // the uniform split schedule -- narrow strips, the pieces of a contraction added up inside a workgroup, few or no
    // partials in memory.  FLUHIP_LIST_PLAN=0 keeps the uniform split schedule, =1 takes the lists whenever something is split.
    static const int listEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_LIST_PLAN"); return e ? std::atoi(e) : -1; }();
    if (!c->strip && c->lazy && listEnv != 0 && (listEnv == 1 || [&] { const PlanShape ps{c->B, c->T, c->F, c->Kp}; return list_plan_pays(&ps); }()))
    {
      c->tOf.assign(B, (int) c->T);
      c->useLists = true;
      return plan_lists(ctx, c);
    }
    // statistics partials of the W update: one per wavefront of a buffer, or one per 64-row chunk from the
    // finalize kernel when the contraction is split
    c->stripsW = c->nsplitW > 1 ? update_finalize_parts((int) c->F - (c->sideW ? 1 : 0), (int) c->Kp)
                                : nmf_update5_strips((int) c->F - (c->sideW ? 1 : 0), (int) c->Kp, (int) c->B);
    c->tailSplitH = 0;
    if (!c->strip && c->lazy && c->nsplitH == 1)
    {
      int wA = 0;
      const int sp = plan_tail(c->B, (int) c->T, (int) c->F, (int) c->Kp, &wA);
      if (sp > 1)
      {
        const int G = ((int) c->T + 15) / 16, w = nmf_update5_strips((int) c->T, (int) c->Kp, (int) c->B);
        c->tailSplitH = sp;
        c->tailStripsH = wA;
        c->tailRestH = w - wA;
        c->tailColsH = wA * ((G + w - 1) / w) * 16;
      }
    }
  }
  if (int rc = alloc_update_scratch(ctx, c)) return rc;
  HIPCHK(ctx, c->clk.alloc(8 * sizeof(long long), true, s));
  if (c->lazy)
  {
    HIPCHK(ctx, c->wnorm.alloc(B * c->Kp * sizeof(double), false, s));
    launch_fill_ones(c->wnorm.as<double>(), (int64_t) (B * c->Kp), s);
    HIPCHK(ctx, c->wscratch.alloc((size_t) wnorm_scratch_doubles((int) c->Kp, (int) B, c->stripsW) * sizeof(double), true, s));
  }
  return FLUHIP_OK;
}

// ---------------------------------------------------------------------------------------
// ragged corpora: work lists of the two factor updates
// ---------------------------------------------------------------------------------------
namespace {
// up to four wavefronts that share a strip and split its contraction (WaveDesc::grp); they stay together in one workgroup
struct WaveGroup
{
  std::vector<WaveDesc> waves; // waves[0] is the leader; grp holds (rank << 4 | size << 8) until the packing
  int64_t work;                // what the group takes: its longest member
  int buf;
};

// groups -> workgroups of 4 wavefronts: longest first (the hardware hands the next workgroup to whichever CU frees up),
// first fit (a later, shorter group fills the slots an earlier workgroup has left).  Workgroup i runs on XCD i & 7 and a
// launch of up to 256 workgroups is ONE round only if every XCD gets at most 32 of them, so the workgroups are numbered
// in list order -- 8 consecutive ones on 8 different XCDs -- rather than by buffer.
void pack_groups(std::vector<WaveGroup>& groups, std::vector<WaveDesc>& out, int* wgs)
{
  std::stable_sort(groups.begin(), groups.end(), [](const WaveGroup& a, const WaveGroup& b) { return a.work > b.work; });
  std::vector<std::vector<WaveDesc>> q; // workgroups of up to 4 descriptors
  size_t firstOpen = 0;
  for (const auto& g : groups)
  {
    bool placed = false;
    while (firstOpen < q.size() && q[firstOpen].size() >= 4) firstOpen++;
    for (size_t i = std::max(firstOpen, q.size() > 256 ? q.size() - 256 : 0); i < q.size(); i++) // (a bounded look-back: linear time)
      if (q[i].size() + g.waves.size() <= 4) { q[i].insert(q[i].end(), g.waves.begin(), g.waves.end()); placed = true; break; }
    if (!placed) q.push_back(g.waves);
  }
  out.assign(q.size() * 4, WaveDesc{0, 0, 0, 0, 0, -1, 0, -1, 0, 0, 0, 0});
  for (size_t i = 0; i < q.size(); i++)
  {
    auto& wgp = q[i];
    bool barrier = false;
    for (const auto& d : wgp) barrier = barrier || ((d.grp >> 8) & 15) > 1;
    size_t leaderAt = 0; // the leader's wavefront index is known only now
    for (size_t w = 0; w < wgp.size(); w++)
    {
      WaveDesc d = wgp[w];
      const int rank = (d.grp >> 4) & 15, size = (d.grp >> 8) & 15;
      if (rank == 0) leaderAt = w;
      d.grp = (int) leaderAt | (rank << 4) | (size << 8) | (barrier ? (1 << 16) : 0);
      out[i * 4 + w] = d;
    }
  }
  *wgs = (int) q.size();
}

int upload_list(fluhip_ctx* ctx, DevBuf& dst, const void* src, size_t bytes)
{
  HIPCHK(ctx, dst.alloc(bytes, false, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // the host image may go out of scope
  return FLUHIP_OK;
}
} // namespace

// Work lists of the two factor updates over buffers with their own frame counts tOf[b]: ragged corpora, and equal-length
// corpora too small to fill the chip with whole contractions (the intra-workgroup reduction is what this form has over the
// uniform split schedule: fewer or no partials in memory, no finalize launch).
namespace {
struct ListSide
{
  std::vector<WaveDesc> list;
  std::vector<int> splitTab;
  int wgs = 0, ng = 0, partial = 0, maxSplit = 1, pieces = 1, statParts = 0;
  int64_t nPartials = 0;
};
struct ListPlanHost
{
  ListSide W, H;
  bool sideW = false;
};
} // namespace
// pure host code (no device, no context): also reachable through fluhip_debug_plan_lists for the CPU tests
static void build_list_plan(const std::vector<int>& tOf, int Tmax, int F, int Kp, ListPlanHost& out)
{
  const int B = (int) tOf.size();
  const int maxNG = nmf_update5_max_groups(Kp);
  // ---- W update: strips over the bins (the same for every buffer), the contraction over a buffer's own frames ------------
  {
    auto envInt = [](const char* name, int dflt) { const char* e = fluhip::ab_getenv(name); return e ? std::atoi(e) : dflt; };
    std::vector<int> steps((size_t) B);
    for (int b = 0; b < B; b++) steps[(size_t) b] = (tOf[(size_t) b] + 3) / 4;
    // the Nyquist bin as a side column (fluhip_kernels.h SideColumn): every power-of-two transform has 16 m + 1 bins, and a
    // wavefront runs the loop of the widest strip of the launch -- 65 column groups never deal evenly
    static const int sideOff = [] { const char* e = fluhip::ab_getenv("FLUHIP_NO_SIDE"); return e ? std::atoi(e) : 0; }();
    out.sideW = !sideOff && nmf_side_column_supported(Tmax, F, Kp);
    const int C = F - (out.sideW ? 1 : 0);
    const int G = (C + 15) / 16;
    // One candidate schedule per strip width.  Contractions are cut into pieces when one round of wavefronts (1024 SIMDs, one
    // workgroup of four per CU) would stay part empty: the wavefront budget goes to whichever buffer has the longest pieces
    // (down to 12 steps); the pieces of one strip go to up to four wavefronts of one workgroup (added up through the LDS:
    // WaveDesc::grp) and only what is left beyond four becomes partials in memory for the finalize launch.  Cost model as
    // for the H update below: ~400 + 270 NG cycles per step of a strip of NG column groups, ~40 k per round, ~60 k per
    // finalize launch -- narrow strips cost little per group, so few buffers take many narrow strips and few partials.
    struct Cand
    {
      std::vector<WaveDesc> list;
      std::vector<int> splitTab;
      bool anyPartial = false;
      int64_t pbase = 0;
      int maxSplit = 1, maxPieces = 1, wgs = 0, wW = 0, ngW = 0;
      double cost = 1e300;
    };
    auto build = [&](int wW, int forcedBudget) -> Cand {
      Cand r;
      r.wW = wW;
      r.ngW = (G + wW - 1) / wW;
      r.splitTab.assign((size_t) B * 2, 0);
      const int gmax = std::min(envInt("FLUHIP_RG_GMAX", 4), nmf_update5_groups_fit(Kp, r.ngW) ? 4 : 1);
      const int64_t waves0 = (int64_t) B * wW;
      const int64_t wgs0 = (waves0 + 3) / 4;
      const int64_t roundsWanted = wgs0 <= 384 ? std::max<int64_t>(1, (wgs0 + 255) / 256) : 0; // 0: many rounds, no cutting
      std::vector<int> nsOf((size_t) B, 1);
      for (int64_t budgetWaves = forcedBudget > 0 ? forcedBudget : 1024 * std::max<int64_t>(roundsWanted, 1);; budgetWaves -= 32)
      {
        std::fill(nsOf.begin(), nsOf.end(), 1);
        if (roundsWanted > 0 && budgetWaves > waves0)
        {
          const int64_t budget = budgetWaves / wW;
          auto piece = [&](int b) { return (steps[(size_t) b] + nsOf[(size_t) b] - 1) / nsOf[(size_t) b]; };
          std::vector<std::pair<int, int>> heap; // (piece, buffer)
          for (int b = 0; b < B; b++) heap.emplace_back(piece(b), b);
          std::make_heap(heap.begin(), heap.end());
          for (int64_t units = B; units < budget; units++)
          {
            std::pop_heap(heap.begin(), heap.end());
            const int b = heap.back().second;
            if (heap.back().first <= 12 || nsOf[(size_t) b] >= 64) break; // the longest piece cannot get shorter
            nsOf[(size_t) b]++;
            heap.back() = std::make_pair(piece(b), b);
            std::push_heap(heap.begin(), heap.end());
          }
        }
        r.anyPartial = false;
        for (int b = 0; b < B; b++) r.anyPartial = r.anyPartial || nsOf[(size_t) b] > gmax;
        std::vector<WaveGroup> groups;
        r.pbase = 0;
        r.maxSplit = 1;
        r.maxPieces = 1;
        int64_t longest = 0;
        for (int b = 0; b < B; b++)
        {
          const int ns = nsOf[(size_t) b];
          const int per = (steps[(size_t) b] + ns - 1) / ns;
          const int np = (ns + gmax - 1) / gmax; // partials in memory
          r.splitTab[(size_t) b * 2] = (int) r.pbase;
          r.splitTab[(size_t) b * 2 + 1] = np;
          r.maxSplit = std::max(r.maxSplit, np);
          r.maxPieces = std::max(r.maxPieces, ns);
          longest = std::max<int64_t>(longest, per);
          int piece = 0;
          for (int q = 0; q < np; q++)
          {
            const int size = ns / np + (q < ns % np ? 1 : 0);
            const int base = G / wW, rem = G % wW;
            int g0 = 0;
            for (int st = 0; st < wW; st++)
            {
              const int ng = base + (st < rem ? 1 : 0);
              WaveGroup grp;
              grp.buf = b; grp.work = 0;
              for (int rk = 0; rk < size; rk++)
              {
                const int s0 = std::min(steps[(size_t) b], (piece + rk) * per), s1 = std::min(steps[(size_t) b], s0 + per); // (an empty tail piece: no steps, zeros)
                WaveDesc d{};
                d.buf = b; d.g0 = g0; d.ng = ng; d.s0 = s0; d.s1 = std::max(s0, s1);
                d.partIdx = (r.anyPartial && rk == 0) ? (int) r.pbase + q : -1;
                d.statIdx = b * wW + st;
                d.dIdx = (r.anyPartial && st == 0 && rk == 0) ? d.partIdx : -1;
                d.grp = (rk << 4) | (size << 8);
                grp.work = std::max<int64_t>(grp.work, d.s1 - d.s0);
                grp.waves.push_back(d);
              }
              if (ng > 0) groups.push_back(std::move(grp));
              g0 += ng;
            }
            piece += size;
          }
          r.pbase += np;
        }
        pack_groups(groups, r.list, &r.wgs);
        const double rounds = roundsWanted == 0 ? (double) r.wgs / 256.0 : (double) ((r.wgs + 255) / 256);
        r.cost = rounds * ((double) longest * (400.0 + 270.0 * r.ngW) + 40000.0) + (r.maxPieces > 1 ? 6000.0 : 0.0) +
                 (r.anyPartial ? 60000.0 + 3000.0 * r.maxSplit : 0.0);
        if (roundsWanted == 0 || r.wgs <= 256 * roundsWanted || budgetWaves <= waves0 || forcedBudget > 0) break;
      }
      return r;
    };
    Cand best;
    const int forcedNG = envInt("FLUHIP_RG_WNG", 0), forcedBudget = envInt("FLUHIP_RG_WBUDGET", 0);
    int lastW = -1;
    for (int ngc = maxNG; ngc >= 1; ngc--)
    {
      if (forcedNG > 0 && ngc != std::min(forcedNG, maxNG)) continue;
      const int wW = (G + ngc - 1) / ngc;
      if (wW == lastW) continue; // the same strips as the wider candidate
      lastW = wW;
      // wide strips re-use the moving factor's rows over more columns: with enough buffers to fill the chip without cutting
      // anything the widest form is what the batched kernel was tuned for -- narrower candidates only while the chip is not full
      if (forcedNG == 0 && (int64_t) B * ((G + maxNG - 1) / maxNG) >= 1536 && ngc < maxNG) break;
      Cand cnd = build(wW, forcedBudget);
      if (cnd.cost < best.cost - 1.0) best = std::move(cnd);
    }
    out.W.wgs = best.wgs;
    out.W.ng = best.ngW; out.W.partial = best.anyPartial ? 1 : 0; out.W.maxSplit = best.maxSplit;
    out.W.nPartials = best.anyPartial ? best.pbase : 0;
    out.W.statParts = best.anyPartial ? update_finalize_parts(C, Kp) : best.wW;
    out.W.pieces = best.maxPieces;
    out.W.list = std::move(best.list);
    out.W.splitTab = std::move(best.splitTab);
  }
  // ---- H update: strips over a buffer's own frames, the contraction over the bins (the same for every buffer) ------------
  {
    std::vector<int> groupsOf((size_t) B);
    for (int b = 0; b < B; b++) groupsOf[(size_t) b] = (tOf[(size_t) b] + 15) / 16;
    // Strip width NG, pieces of the bin contraction that share a workgroup (gsz: added up through the LDS) and partials in
    // memory (ns: a finalize launch) by a small cost model in shader cycles, fitted to tools/ragged_sweep.sh
    // and tools/batch_timing.py: a wavefront runs the loop of the launch's widest strip at ~400 + 270 NG cycles per 4-row step
    // (measured: 1 group 695, 4: 1 370, 6: 1 780, 7: 2 250, 8: 2 620, 9: 2 775), a round of up to 256 workgroups pays ~40 k
    // cycles of prologue, epilogue and launch, a finalize launch ~60 k.
    const int steps = (F + 3) / 4;
    int NG = maxNG, gsz = 1, ns = 1;
    double bestCost = 1e300;
    for (int cand = maxNG; cand >= 1; cand--)
    {
      int64_t strips = 0;
      int widestStrip = 1;
      for (int b = 0; b < B; b++)
      {
        const int w = (groupsOf[(size_t) b] + cand - 1) / cand;
        strips += w;
        widestStrip = std::max(widestStrip, (groupsOf[(size_t) b] + w - 1) / std::max(w, 1));
      }
      const bool fit = nmf_update5_groups_fit(Kp, widestStrip);
      for (int g : {1, 2, 4})
      {
        if (g > 1 && !fit) continue;
        for (int m : {1, 2, 3, 4, 6, 8, 12, 16})
        {
          const int pieces = g * m;
          if (pieces > 1 && steps / pieces < 12) continue;
          const int64_t wgs = (strips * m + (4 / g) - 1) / (4 / g);
          const double rounds = (double) ((wgs + 255) / 256);
          const double perWave = (double) ((steps + pieces - 1) / pieces) * (400.0 + 270.0 * widestStrip) + 40000.0;
          const double cost = rounds * perWave + (g > 1 ? 6000.0 : 0.0) + (m > 1 ? 60000.0 + 3000.0 * m : 0.0);
          if (cost < bestCost - 1.0) { bestCost = cost; NG = cand; gsz = g; ns = m; }
        }
      }
    }
    {
      auto envInt = [](const char* name, int dflt) { const char* e = fluhip::ab_getenv(name); return e ? std::atoi(e) : dflt; };
      NG = std::max(1, std::min(maxNG, envInt("FLUHIP_RG_HNG", NG)));
      gsz = envInt("FLUHIP_RG_HG", gsz);
      ns = envInt("FLUHIP_RG_HM", ns);
    }
    int ngH = 1;
    for (int b = 0; b < B; b++)
    {
      const int w = (groupsOf[(size_t) b] + NG - 1) / NG;
      ngH = std::max(ngH, (groupsOf[(size_t) b] + w - 1) / std::max(w, 1));
    }
    if (!nmf_update5_groups_fit(Kp, ngH)) gsz = 1;
    const int pieces = gsz * ns;
    const int per = (steps + pieces - 1) / pieces;
    std::vector<WaveGroup> groups;
    std::vector<int> splitTab((size_t) B * 2);
    for (int b = 0; b < B; b++) { splitTab[(size_t) b * 2] = b * ns; splitTab[(size_t) b * 2 + 1] = ns; }
    for (int b = 0; b < B; b++)
    {
      const int G = groupsOf[(size_t) b];
      const int w = (G + NG - 1) / NG;
      for (int j = 0; j < ns; j++)
      {
        const int base = G / w, rem = G % w;
        int g0 = 0;
        for (int st = 0; st < w; st++)
        {
          const int ng = base + (st < rem ? 1 : 0);
          WaveGroup grp;
          grp.buf = b; grp.work = 0;
          for (int r = 0; r < gsz; r++)
          {
            WaveDesc d{};
            d.buf = b; d.g0 = g0; d.ng = ng;
            d.s0 = std::min(steps, (j * gsz + r) * per); d.s1 = std::max(d.s0, std::min(steps, d.s0 + per));
            d.partIdx = (ns > 1 && r == 0) ? b * ns + j : -1;
            d.statIdx = 0;
            d.dIdx = (ns > 1 && st == 0 && r == 0) ? d.partIdx : -1;
            d.grp = (r << 4) | (gsz << 8);
            grp.work = std::max<int64_t>(grp.work, (int64_t) (d.s1 - d.s0) * ng);
            grp.waves.push_back(d);
          }
          if (ng > 0) groups.push_back(std::move(grp));
          g0 += ng;
        }
      }
    }
    pack_groups(groups, out.H.list, &out.H.wgs);
    out.H.ng = ngH; out.H.partial = ns > 1 ? 1 : 0; out.H.maxSplit = ns; out.H.nPartials = ns > 1 ? (int64_t) B * ns : 0;
    out.H.pieces = pieces;
    out.H.splitTab = std::move(splitTab);
  }
}

static int plan_lists(fluhip_ctx* ctx, fluhip_corpus* c)
{
  hipStream_t s = ctx->stream;
  const int B = (int) c->B, Kp = (int) c->Kp;
  c->lazy = true; c->strip = false;
  ListPlanHost plan;
  build_list_plan(c->tOf, (int) c->T, (int) c->F, Kp, plan);
  c->sideW = plan.sideW;
  auto take = [&](fluhip_corpus::WorkList& dst, const ListSide& src) -> int {
    dst.wgs = src.wgs; dst.ng = src.ng; dst.partial = src.partial; dst.maxSplit = src.maxSplit; dst.nPartials = src.nPartials;
    dst.statParts = src.statParts;
    if (int rc = upload_list(ctx, dst.list, src.list.data(), src.list.size() * sizeof(WaveDesc))) return rc;
    return upload_list(ctx, dst.splitTab, src.splitTab.data(), src.splitTab.size() * sizeof(int));
  };
  if (int rc = take(c->listW, plan.W)) return rc;
  if (int rc = take(c->listH, plan.H)) return rc;
  c->stripsW = plan.W.statParts;
  c->nsplitW = plan.W.pieces;
  c->nsplitH = plan.H.pieces;

  // workspaces
  const int64_t nPart = std::max(c->listW.nPartials, c->listH.nPartials);
  const size_t Cp = (size_t) std::max(c->Fp, c->Tp);
  if (nPart > 0) HIPCHK(ctx, c->part.alloc((size_t) nPart * Cp * Kp * sizeof(double), true, s));
  HIPCHK(ctx, c->dpart.alloc(std::max<size_t>(256, (size_t) std::max<int64_t>(nPart, B) * Kp * sizeof(double)), true, s));
  HIPCHK(ctx, c->clk.alloc(8 * sizeof(long long), true, s));
  HIPCHK(ctx, c->wnorm.alloc((size_t) B * Kp * sizeof(double), false, s));
  launch_fill_ones(c->wnorm.as<double>(), (int64_t) B * Kp, s);
  HIPCHK(ctx, c->wscratch.alloc((size_t) wnorm_scratch_doubles(Kp, B, c->stripsW) * sizeof(double), true, s));
  return FLUHIP_OK;
}

static int corpus_alloc(fluhip_ctx* ctx, fluhip_corpus* c)
{
  hipStream_t s = ctx->stream;
  c->T = (c->n + c->hop) / c->hop; // alg/STFT.hpp:98-99; nrt/NMFClient.hpp:111-112
  c->F = c->fft / 2 + 1;
  c->Tp = round_up(c->T, 32);
  c->Fp = round_up(c->F, 32);
  c->Kp = padded_rank(c->K);
  const size_t B = (size_t) c->B;
  HIPCHK(ctx, c->mag.alloc(B * c->Tp * c->Fp * sizeof(double), true, s));
  HIPCHK(ctx, c->magT.alloc(B * c->Fp * c->Tp * sizeof(double), true, s));
  HIPCHK(ctx, c->Wf.alloc(B * c->Fp * c->Kp * sizeof(double), true, s));
  HIPCHK(ctx, c->H1.alloc(B * c->Tp * c->Kp * sizeof(double), true, s));
  HIPCHK(ctx, c->hmax.alloc(B * sizeof(double), true, s));
  if (c->ragged)
  {
    HIPCHK(ctx, c->nTab.alloc(B * sizeof(int64_t), false, s));
    HIPCHK(ctx, c->tTab.alloc(B * sizeof(int), false, s));
    HIPCHK(ctx, hipMemcpyAsync(c->nTab.p, c->nOf.data(), B * sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(c->tTab.p, c->tOf.data(), B * sizeof(int), hipMemcpyHostToDevice, s));
    c->useLists = true;
    return plan_lists(ctx, c);
  }
  if (int rc = plan_updates(ctx, c)) return rc;
  return FLUHIP_OK;
}

// launch-geometry limits of the factor-update paths, refused up front with a message instead of surfacing as an
// "invalid configuration" launch error: the normalisation kernels put the padded rank in one workgroup (<= 1024
// threads), and the any-rank path (rank above 128, kernels_nmf_wide.hip) puts frames / bins in gridDim.y (<= 65535)
static int check_rank(fluhip_ctx* ctx, int64_t T, int64_t F, int64_t K)
{
  if (padded_rank(K) > 1024) return fail(ctx, "ranks above 1024 are not supported");
  if (padded_rank(K) > 128 && std::max(T, F) > 65535)
    return fail(ctx, "ranks above 128 are limited to 65535 frames and bins");
  return FLUHIP_OK;
}

static int check_shape(fluhip_ctx* ctx, int64_t n, int64_t win, int64_t fft, int64_t hop, int64_t K)
{
  if (n <= 0) return fail(ctx, "not enough frames");
  if (win < 1 || hop < 1) return fail(ctx, "window and hop sizes must be positive");
  if (fft < 4 || (fft & (fft - 1)) || fft < win)
    return fail(ctx, "fft size must be a power of two >= window size");
  if (!stft_supported(win, fft))
    return fail(ctx, "fft sizes above 65536 are not supported");
  if (K < 1) return fail(ctx, "rank must be >= 1");
  if ((n + hop) / hop > 2000000000LL / 16) return fail(ctx, "too many frames");
  return check_rank(ctx, (n + hop) / hop, fft / 2 + 1, K);
}

static int corpus_stft(fluhip_corpus* c, const float* a32, const double* a64, int64_t audioStride)
{
  fluhip_ctx* ctx = c->ctx;
  const double *wtab = nullptr, *ttab = nullptr;
  int rc = get_window(ctx, c->win, c->fft, c->windowType, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, c->fft, &ttab);
  if (rc) return rc;
  if (c->keepSpec && !c->spec.p)
    HIPCHK(ctx, c->spec.alloc((size_t) c->B * c->T * c->F * 2 * sizeof(double), false, ctx->stream));
  StftArgs a;
  a.audio = a32; a.audio64 = a64; a.n = c->n; a.audioStride = audioStride;
  a.win = (int) c->win; a.fft = (int) c->fft; a.hop = (int) c->hop;
  a.T = (int) c->T; a.F = (int) c->F; a.B = (int) c->B;
  a.window = wtab; a.twiddle = ttab;
  a.mag = c->mag.as<double>(); a.magStride = c->Tp * c->Fp; a.ldMag = c->Fp;
  a.spec = c->keepSpec ? c->spec.as<double>() : nullptr; a.specStride = c->T * c->F * 2;
  a.frameOffset = 0;
  a.nTab = c->ragged ? c->nTab.as<int64_t>() : nullptr;
  a.bigScratch = big_fft_scratch(ctx, c->win, c->fft, c->B * c->T);
  if (stft_needs_scratch(c->win, c->fft) && !a.bigScratch) return FLUHIP_ERROR;
  // V is kept in both layouts (frame-major for the W update, bin-major for the H update).  The block form of K1
  // writes both in one pass; shapes it does not cover take the wave / generic kernel and a transposing copy.
  bool both = false;
  {
    ProfScope p(ctx, 0);
    if (!stft_needs_scratch(c->win, c->fft))
      both = launch_stft_block(a, c->magT.as<double>(), c->Fp * c->Tp, c->Tp, ctx->stream);
    if (!both && c->ragged) return fail(ctx, "ragged corpora need an STFT shape with a block form (fft 1024 / 2048 / 4096, even window)");
    if (!both) launch_stft(a, ctx->stream);
  }
  if (!both)
  {
    ProfScope p(ctx, 4);
    launch_transpose(c->mag.as<double>(), c->Fp, c->Tp * c->Fp, c->magT.as<double>(), c->Tp,
                     c->Fp * c->Tp, (int) c->T, (int) c->F, (int) c->B, ctx->stream);
  }
  HIPCHK(ctx, hipGetLastError());
  c->haveMag = true;
  c->touched = true;
  return FLUHIP_OK;
}

// util/EigenRandom.hpp:73-101: std::mt19937_64 g{seed ? *seed : rd()} +
// std::uniform_real_distribution<double>{0, 1}; one draw per coefficient in Eigen's column-major
// linear order.  libstdc++'s <random> is used verbatim, exactly as the reference does.
static void draw_uniform(int64_t seed, size_t count, std::vector<double>& out)
{
  std::random_device rd;
  std::mt19937_64 g{seed >= 0 ? (size_t) seed : (size_t) rd()};
  std::uniform_real_distribution<double> d{0.0, 1.0};
  out.resize(count);
  for (size_t i = 0; i < count; i++) out[i] = d(g);
}

struct FactorInit
{
  // device sources already in the padded layout are marked by null here
  const double* W0host = nullptr; // [B or 1][K][F] f64
  const double* H0host = nullptr; // [B or 1][T][K] f64
  const float* W0f32 = nullptr;   // [B][K][F] f32 channel-major seeds
  const float* H0f32 = nullptr;   // [B][K][T] f32 channel-major seeds
  bool sharedW = false, sharedH = false;
};

static int corpus_init_factors(fluhip_corpus* c, int64_t seed, const int64_t* seeds,
                               const FactorInit& fi)
{
  fluhip_ctx* ctx = c->ctx;
  hipStream_t s = ctx->stream;
  const size_t FK = (size_t) c->F * c->K, TK = (size_t) c->T * c->K;
  const int B = (int) c->B;
  // The draws run on the calling thread while the device is still busy with whatever was enqueued before (the STFT
  // of this job); the host images stay alive to the single synchronisation at the end, so the two uploads and the
  // scatter kernels queue up behind each other without a host round trip in between.
  std::vector<double> hostW, hostH;
  DevBuf stageH;
  // --- W ---
  if (fi.W0f32)
  {
    HIPCHK(ctx, c->stage.alloc((size_t) B * FK * sizeof(float), false, s));
    HIPCHK(ctx, hipMemcpyAsync(c->stage.p, fi.W0f32, (size_t) B * FK * sizeof(float), hipMemcpyHostToDevice, s));
    launch_scatter_factor_f32(c->stage.as<float>(), (int64_t) FK, c->Wf.as<double>(), c->Fp * c->Kp,
                              (int) c->F, (int) c->K, (int) c->Kp, B, s);
  }
  else
  {
    std::vector<double>& host = hostW;
    const double* src = fi.W0host;
    int nsrc = fi.W0host ? (fi.sharedW ? 1 : B) : 1;
    if (!src)
    {
      if (seeds)
      {
        nsrc = B;
        host.resize((size_t) B * FK);
        std::vector<double> tmp;
        std::map<int64_t, int> seen;
        for (int b = 0; b < B; b++)
        {
          auto it = seeds[b] >= 0 ? seen.find(seeds[b]) : seen.end();
          if (it != seen.end())
            std::memcpy(&host[(size_t) b * FK], &host[(size_t) it->second * FK], FK * sizeof(double));
          else
          {
            draw_uniform(seeds[b], FK, tmp);
            std::memcpy(&host[(size_t) b * FK], tmp.data(), FK * sizeof(double));
            if (seeds[b] >= 0) seen[seeds[b]] = b;
          }
        }
      }
      else
        draw_uniform(seed, FK, host); // alg/NMF.hpp:104-105
      src = host.data();
    }
    HIPCHK(ctx, c->stage.alloc((size_t) nsrc * FK * sizeof(double), false, s));
    HIPCHK(ctx, hipMemcpyAsync(c->stage.p, src, (size_t) nsrc * FK * sizeof(double), hipMemcpyHostToDevice, s));
    // K x F row-major source (random: column-major F x K fill; seeded: W0 transposed, :102-112)
    launch_scatter_factor(c->stage.as<double>(), nsrc == 1 ? 0 : (int64_t) FK, c->Wf.as<double>(),
                          c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, B, true, s);
  }
  // --- H ---
  if (fi.H0f32)
  {
    HIPCHK(ctx, stageH.alloc((size_t) B * TK * sizeof(float), false, s));
    HIPCHK(ctx, hipMemcpyAsync(stageH.p, fi.H0f32, (size_t) B * TK * sizeof(float), hipMemcpyHostToDevice, s));
    launch_scatter_factor_f32(stageH.as<float>(), (int64_t) TK, c->H1.as<double>(), c->Tp * c->Kp,
                              (int) c->T, (int) c->K, (int) c->Kp, B, s, c->ragged ? c->tTab.as<int>() : nullptr);
  }
  else
  {
    std::vector<double>& host = hostH;
    const double* src = fi.H0host;
    int nsrc = fi.H0host ? (fi.sharedH ? 1 : B) : 1;
    if (!src)
    {
      if (seeds)
      {
        nsrc = B;
        host.resize((size_t) B * TK);
        std::vector<double> tmp;
        std::map<int64_t, int> seen;
        for (int b = 0; b < B; b++)
        {
          auto it = seeds[b] >= 0 ? seen.find(seeds[b]) : seen.end();
          if (it != seen.end())
            std::memcpy(&host[(size_t) b * TK], &host[(size_t) it->second * TK], TK * sizeof(double));
          else
          {
            draw_uniform(seeds[b], TK, tmp);
            std::memcpy(&host[(size_t) b * TK], tmp.data(), TK * sizeof(double));
            if (seeds[b] >= 0) seen[seeds[b]] = b;
          }
        }
      }
      else
        draw_uniform(seed, TK, host); // alg/NMF.hpp:116-117 (a fresh generator from the same seed)
      src = host.data();
    }
    HIPCHK(ctx, stageH.alloc((size_t) nsrc * TK * sizeof(double), false, s));
    HIPCHK(ctx, hipMemcpyAsync(stageH.p, src, (size_t) nsrc * TK * sizeof(double), hipMemcpyHostToDevice, s));
    // T x K row-major source (random: column-major K x T fill; seeded: H0 transposed, :113-124)
    launch_scatter_factor(stageH.as<double>(), nsrc == 1 ? 0 : (int64_t) TK, c->H1.as<double>(),
                          c->Tp * c->Kp, (int) c->T, (int) c->K, (int) c->Kp, B, false, s,
                          c->ragged ? c->tTab.as<int>() : nullptr); // (K x T_b column-major = the first T_b K draws)
  }
  // alg/NMF.hpp:150-153: clamp both to eps, normalise columns of W and rows of H (= columns of H1)
  if (!c->normScratch.p)
  {
    const size_t nd = (size_t) colnorm_scratch_doubles((int) std::max(c->T, c->F), (int) c->Kp, B);
    HIPCHK(ctx, c->normScratch.alloc(nd * sizeof(double), false, s));
  }
  launch_colnorm(c->H1.as<double>(), c->Tp * c->Kp, (int) c->T, (int) c->K, (int) c->Kp, B, true, false,
                 c->normScratch.as<double>(), s, c->ragged ? c->tTab.as<int>() : nullptr);
  launch_colnorm(c->Wf.as<double>(), c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, B, true, false,
                 c->normScratch.as<double>(), s);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(s)); // the host images and the second staging buffer go out of scope
  c->haveFactors = true;
  c->stripStatsValid = false;
  return FLUHIP_OK;
}

static void enqueue_iteration(fluhip_corpus* c, bool updateW, bool updateH, bool last)
{
  fluhip_ctx* ctx = c->ctx;
  hipStream_t s = ctx->stream;
  const int B = (int) c->B;
  if (c->strip)
  {
    StripArgs a;
    a.V = c->mag.as<double>(); a.strideV = c->Tp * c->Fp; a.ldv = c->Fp;
    a.W = c->Wf.as<double>(); a.strideW = c->Fp * c->Kp;
    a.H = c->H1.as<double>(); a.strideH = c->Tp * c->Kp;
    a.part = c->stripPart.as<double>(); a.nrm = c->wnorm.as<double>();
    a.F = (int) c->F; a.T = (int) c->T; a.K = (int) c->K; a.B = B;
    a.doH = a.doW = 0; a.wPend = c->wPending ? 1 : 0;
    if (!c->stripStatsValid)
    {
      // W was written by something else (initialisation, the normalisation at the end of the last call)
      a.statGen = c->stripGen;
      launch_nmf_strip_wstats(a, s);
      c->stripStatsValid = true;
    }
    if (updateW)
    {
      // alg/NMF.hpp:158-161; :162 is implicit in the next staging of W'
      if (!c->stripReady)
      {
        a.doH = 0; a.doW = 1; a.wPend = c->wPending ? 1 : 0; a.statGen = c->stripGen;
        ProfScope p(ctx, 1);
        launch_nmf_strip(a, s);
      }
      {
        a.wPend = c->wPending ? 1 : 0; a.statGen = c->stripGen;
        ProfScope p(ctx, 3);
        launch_nmf_strip_reduce(a, s);
        c->stripGen ^= 1;
      }
      c->wPending = true;
      c->stripReady = false;
      c->stripNormFresh = false;
    }
    if (updateH)
    {
      // :165-170, and behind it the numerator of the next iteration's W update while the new H is at hand
      a.doH = 1; a.doW = (updateW && !last) ? 1 : 0; a.wPend = c->wPending ? 1 : 0; a.statGen = c->stripGen;
      ProfScope p(ctx, 1);
      launch_nmf_strip(a, s);
      c->stripReady = a.doW != 0;
      c->stripNormFresh = true;
    }
    return;
  }
  // (a window of the corpus: every per-buffer array starts b0 buffers in; the scratch of the small kernels is reused)
  const int64_t b0 = c->winB ? c->winB0 : 0;
  const int Bw = c->winB ? (int) c->winB : B;
  double* const magW = c->mag.as<double>() + b0 * c->Tp * c->Fp;
  double* const magTW = c->magT.as<double>() + b0 * c->Fp * c->Tp;
  double* const WfW = c->Wf.as<double>() + b0 * c->Fp * c->Kp;
  double* const H1W = c->H1.as<double>() + b0 * c->Tp * c->Kp;
  double* const wnormW = c->wnorm.p ? c->wnorm.as<double>() + b0 * c->Kp : nullptr;
  if (updateW)
  {
    // alg/NMF.hpp:158-161
    UpdateArgs a;
    a.V = magW; a.ldv = c->Fp; a.strideV = c->Tp * c->Fp;
    a.Mv = H1W; a.strideM = c->Tp * c->Kp;
    a.S = WfW; a.strideS = c->Fp * c->Kp;
    a.R = (int) c->T; a.C = (int) c->F - (c->sideW ? 1 : 0); a.B = Bw; a.Kp = (int) c->Kp;
    if (c->winB) a.stripsOverride = c->winStripsW;
    a.nsplit = c->nsplitW; a.part = c->part.as<double>(); a.dpart = c->dpart.as<double>();
    a.Cp = c->useLists ? std::max(c->Fp, c->Tp) : c->Fp; a.colsumScratch = c->csumScratch.as<double>();
    a.clk = c->clk.as<long long>();
    if (c->useLists)
    {
      a.list = c->listW.list.as<WaveDesc>(); a.listWGs = c->listW.wgs; a.listNG = c->listW.ng; a.listPartial = c->listW.partial;
    }
    if (c->lazy)
    {
      // W' = W diag(wnorm) in memory: the kernel divides its stationary rows by wnorm, writes the new W'
      // and its per-wavefront column statistics; [side column ->] new wnorm.  alg/NMF.hpp:162 is then
      // implicit in every later use of (W', wnorm).
      a.nrm = wnormW; a.nrmMode = 1; a.statPart = c->wscratch.as<double>();
      {
        ProfScope p(ctx, 1);
        launch_nmf_update5(a, s);
        if (c->useLists && c->listW.partial)
          launch_update_finalize(a.S, a.strideS, a.part, a.dpart, a.C, a.Kp, a.Cp, c->listW.maxSplit, a.B, s, a.nrm, a.nrmMode,
                                 a.statPart, c->listW.splitTab.as<int>());
      }
      ProfScope p(ctx, 3);
      SideColumn sc{magTW + (c->F - 1) * c->Tp, c->Fp * c->Tp, H1W, c->Tp * c->Kp, (int) c->T};
      launch_wnorm_combine(WfW, c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, Bw, c->stripsW,
                           c->wscratch.as<double>(), wnormW, c->sideW ? &sc : nullptr, s);
      c->wPending = true;
    }
    else
    {
      {
        ProfScope p(ctx, 1);
        if (update_variant(a.Kp) == 5) launch_nmf_update5(a, s);
        else launch_nmf_update_wide(a, c->wideScratch.as<double>(), s);
      }
      // :162  if (W.maxCoeff() > epsilon) W.colwise().normalize()
      launch_colnorm(WfW, c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, Bw, false, true,
                     c->normScratch.as<double>(), s);
    }
  }
  if (updateH)
  {
    // alg/NMF.hpp:165-170 (V2 is formed from the already updated W)
    UpdateArgs a;
    a.V = magTW; a.ldv = c->Tp; a.strideV = c->Fp * c->Tp;
    a.Mv = WfW; a.strideM = c->Fp * c->Kp;
    a.S = H1W; a.strideS = c->Tp * c->Kp;
    a.R = (int) c->F; a.C = (int) c->T; a.B = Bw; a.Kp = (int) c->Kp;
    if (c->winB) a.stripsOverride = c->winStripsH;
    a.nsplit = c->nsplitH; a.part = c->part.as<double>(); a.dpart = c->dpart.as<double>();
    a.Cp = c->useLists ? std::max(c->Fp, c->Tp) : c->Tp; a.colsumScratch = c->csumScratch.as<double>();
    a.clk = c->clk.as<long long>() + 4;
    if (c->wPending) { a.nrm = wnormW; a.nrmMode = 2; }
    ProfScope p(ctx, 1);
    const int uv = update_variant(a.Kp);
    if (c->useLists)
    {
      a.list = c->listH.list.as<WaveDesc>(); a.listWGs = c->listH.wgs; a.listNG = c->listH.ng; a.listPartial = c->listH.partial;
      launch_nmf_update5(a, s);
      if (c->listH.partial)
        launch_update_finalize(a.S, a.strideS, a.part, a.dpart, a.C, a.Kp, a.Cp, c->listH.maxSplit, a.B, s, a.nrm, a.nrmMode,
                               nullptr, c->listH.splitTab.as<int>());
    }
    else if (uv == 5 && c->tailSplitH > 1 && !c->winB)
    {
      // two launches (plan_tail): whole contractions for the frames that fill whole rounds, split ones for the rest
      UpdateArgs a1 = a;
      a1.C = c->tailColsH; a1.stripsOverride = c->tailStripsH; a1.nsplit = 1;
      launch_nmf_update5(a1, s);
      UpdateArgs a2 = a;
      a2.V = a.V + c->tailColsH; a2.S = a.S + (int64_t) c->tailColsH * c->Kp; a2.C = a.C - c->tailColsH;
      a2.stripsOverride = c->tailRestH; a2.nsplit = c->tailSplitH; a2.Cp = round_up(a2.C, 32);
      a2.dpart = a.dpart + (int64_t) Bw * c->Kp;
      if (c->Kp > 64) a2.colsumGiven = a.dpart; // (the first launch's pre-pass left the column sums of W there)
      a2.clk = nullptr;                         // the clock stamps stay those of the whole-contraction wavefront
      launch_nmf_update5(a2, s);
    }
    else if (uv == 5) launch_nmf_update5(a, s);
    else launch_nmf_update_wide(a, c->wideScratch.as<double>(), s);
  }
}

// alg/NMF.hpp:154-181 loop + :175-176 callbacks.  Iterations are enqueued back to back; with a progress callback
// every iteration is followed by an event and the callback is invoked once per completed iteration, in order, on
// the calling thread, with a bounded run-ahead.
static int corpus_iterate_loop(fluhip_corpus* c, int64_t iters, bool updateW, bool updateH,
                               fluhip_progress_fn progress, void* user)
{
  fluhip_ctx* ctx = c->ctx;
  if (!progress)
  {
    // Corpora of several rounds of wavefronts (more than 1024 / strips buffers) run ROUND-MAJOR: all iterations of the
    // first 128 buffers (at 8 strips), then of the next 128, ...  Buffers are independent jobs (nrt/NMFClient.hpp:233),
    // so the order is free, and a round's factor matrices (63 MB at the bench shape) stay in the 256 MB last-level cache
    // from launch to launch, which the whole corpus's (500 MB at 1024 buffers) do not: 8-round launches measured 0.304 -
    // 0.319 ms per round against 0.271 - 0.278 for one-round launches (profiles/r03/bench_v*_1024_buffers_one_gpu.json).
    // With a progress callback the iterations stay outermost: "iteration i" means every buffer has passed it.
    // FLUHIP_ROUND_MAJOR=0: iteration-major as before.
    static const int roundEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_ROUND_MAJOR"); return e ? std::atoi(e) : 1; }();
    int64_t chunk = 0;
    if (roundEnv && !c->strip && !c->useLists && c->lazy && c->nsplitW == 1 && c->nsplitH == 1 && update_variant((int) c->Kp) == 5)
    {
      const int CW = (int) c->F - (c->sideW ? 1 : 0);
      const int wW = nmf_update5_strips(CW, (int) c->Kp, (int) c->B), wH = nmf_update5_strips((int) c->T, (int) c->Kp, (int) c->B);
      const int64_t per = 1024 / std::max(wW, wH);
      if (per >= 8 && c->B >= 2 * per)
      {
        // the schedule of ONE round of `per` buffers, kept for every window (the update kernel would otherwise deal its
        // strips from the number of buffers it is launched on); the W update's strips also fix the layout of its column
        // statistics, which was planned for the whole corpus
        const int wWc = nmf_update5_strips(CW, (int) c->Kp, (int) per), wHc = nmf_update5_strips((int) c->T, (int) c->Kp, (int) per);
        // (both updates must fill most of a round at that window: short buffers at a wide rank -- 128 x 1.6 s, rank 128: 32
        //  strips over the bins, 5 over the frames -- ran H updates of 160 wavefronts per window: 1 510 us per iteration against
        //  608 iteration-major, tools/wide_list_ab.sh)
        if (wWc == wW && per * std::max(wWc, wHc) <= 1024 && per * std::min(wWc, wHc) >= 768)
        {
          chunk = per; c->winStripsW = wWc; c->winStripsH = wHc;
        }
      }
    }
    if (chunk > 0)
    {
      for (int64_t b0 = 0; b0 < c->B; b0 += chunk)
      {
        c->winB0 = b0;
        c->winB = std::min(chunk, c->B - b0);
        for (int64_t i = 0; i < iters; i++) enqueue_iteration(c, updateW, updateH, i + 1 == iters);
      }
      c->winB0 = c->winB = 0;
    }
    else
    {
      // FLUHIP_GRAPH_ITERS=n (experiment, default off -- DESIGN.md "hipGraph"): after the first iteration, runs of n iterations
      // are captured once as a hipGraph and replayed; the last iterations (fewer than n + 1) are enqueued as usual.  n even: the
      // frame-strip schedule alternates two statistics buffers.  Not with the profiler's events in the stream.
      static const int graphN = [] { const char* e = fluhip::ab_getenv("FLUHIP_GRAPH_ITERS"); return e ? std::atoi(e) & ~1 : 0; }();
      int64_t i = 0;
      if (graphN >= 2 && !ctx->prof && iters >= 2 * (int64_t) graphN + 2)
      {
        enqueue_iteration(c, updateW, updateH, false);
        i = 1;
        // whatever fails in here, the stream leaves capture mode and the graph objects are released (an early
        // return between Begin- and EndCapture left the context's stream capturing, and every later call on it failed)
        struct Capture
        {
          hipStream_t s;
          hipGraph_t graph = nullptr;
          hipGraphExec_t exec = nullptr;
          bool capturing = false;
          ~Capture()
          {
            if (capturing) (void) hipStreamEndCapture(s, &graph);
            if (exec) (void) hipGraphExecDestroy(exec);
            if (graph) (void) hipGraphDestroy(graph);
          }
        } cap{ctx->stream};
        HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        cap.capturing = true;
        for (int g = 0; g < graphN; g++) enqueue_iteration(c, updateW, updateH, false);
        cap.capturing = false;
        HIPCHK(ctx, hipStreamEndCapture(ctx->stream, &cap.graph));
        HIPCHK(ctx, hipGraphInstantiate(&cap.exec, cap.graph, nullptr, nullptr, 0));
        for (; i + graphN < iters; i += graphN) HIPCHK(ctx, hipGraphLaunch(cap.exec, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // (experiment: the graph is dropped right away)
      }
      for (; i < iters; i++) enqueue_iteration(c, updateW, updateH, i + 1 == iters);
    }
    HIPCHK(ctx, hipGetLastError());
    return FLUHIP_OK;
  }
  // One event per iteration; callbacks are delivered in order as the events complete, and the host never runs more
  // than kLag iterations ahead of the last one it has reported: a cancel at iteration i stops the device after at
  // most kLag - 1 further iterations (alg/NMF.hpp:175-176 stops at i exactly; the client above never looks at the
  // factors of a cancelled job, clients/nrt/NMFClient.hpp:273-274).
  constexpr int kMaxLag = 64;
  const int kLag = std::max(1, std::min(kMaxLag, ctx->progressLag));
  hipEvent_t ev[kMaxLag];
  for (int i = 0; i < kLag; i++) ev[i] = take_event(ctx);
  auto give_back = [&] { for (int i = 0; i < kLag; i++) ctx->eventPool.push_back(ev[i]); };
  int64_t reported = 0;
  for (int64_t i = 0; i < iters; i++)
  {
    enqueue_iteration(c, updateW, updateH, i + 1 == iters);
    if (hipEventRecord(ev[i % kLag], ctx->stream) != hipSuccess) { give_back(); return fail(ctx, "HIP error: hipEventRecord"); }
    const int64_t enq = i + 1;
    while (reported < enq)
    {
      hipEvent_t e = ev[reported % kLag];
      const bool mustWait = enq - reported >= kLag || enq == iters;
      hipError_t q = mustWait ? hipEventSynchronize(e) : hipEventQuery(e);
      if (q == hipErrorNotReady) break;
      if (q != hipSuccess) { give_back(); return fail(ctx, std::string("HIP error: ") + hipGetErrorString(q) + " in the iteration loop"); }
      reported++;
      if (!progress(reported, user))
      {
        (void) hipStreamSynchronize(ctx->stream);
        give_back();
        return fail(ctx, "cancelled", FLUHIP_CANCELLED);
      }
    }
  }
  give_back();
  return FLUHIP_OK;
}

static int corpus_iterate(fluhip_corpus* c, int64_t iters, bool updateW, bool updateH,
                          fluhip_progress_fn progress, void* user)
{
  const int rc = corpus_iterate_loop(c, iters, updateW, updateH, progress, user);
  if (c->strip)
  {
    c->stripReady = false; // H may change before the next call
    if (c->wPending && !c->stripNormFresh)
    {
      // the last launch was a reduce: one workgroup per buffer recomputes the column norms of W'
      StripArgs a;
      a.V = c->mag.as<double>(); a.strideV = c->Tp * c->Fp; a.ldv = c->Fp;
      a.W = c->Wf.as<double>(); a.strideW = c->Fp * c->Kp;
      a.H = c->H1.as<double>(); a.strideH = c->Tp * c->Kp;
      a.part = c->stripPart.as<double>(); a.nrm = c->wnorm.as<double>();
      a.F = (int) c->F; a.T = (int) c->T; a.K = (int) c->K; a.B = (int) c->B;
      a.doH = a.doW = 0; a.wPend = 1; a.statGen = c->stripGen;
      launch_nmf_strip(a, c->ctx->stream);
      c->stripNormFresh = true;
    }
    if (c->wPending) c->stripStatsValid = false; // the normalisation below rewrites W
  }
  if (c->wPending)
  {
    // leave the deferred form on every exit (also a cancelled run hands back W, alg/NMF.hpp:175-176)
    launch_wnorm_apply(c->Wf.as<double>(), c->Fp * c->Kp, (int) c->F, (int) c->Kp, (int) c->B, c->wnorm.as<double>(),
                       c->ctx->stream);
    c->wPending = false;
  }
  return rc;
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int fluhip_abi_version(void) { return FLUHIP_ABI_VERSION; }

int fluhip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fluhip_ctx_create(int device, fluhip_ctx** out)
{
  if (!out) return FLUHIP_ERROR;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return FLUHIP_ERROR;
  std::unique_ptr<fluhip_ctx> ctx(new fluhip_ctx);
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) return FLUHIP_ERROR;
  if (hipGetDeviceProperties(&ctx->props, device) != hipSuccess) return FLUHIP_ERROR;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return FLUHIP_ERROR;
  *out = ctx.release();
  return FLUHIP_OK;
}

void fluhip_ctx_destroy(fluhip_ctx* ctx)
{
  if (!ctx) return;
  (void) hipSetDevice(ctx->device);
  if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
  if (ctx->bigFft) (void) hipFree(ctx->bigFft);
  for (int i = 0; i < 2; i++)
  {
    if (ctx->stage[i]) (void) hipHostFree(ctx->stage[i]);
    if (ctx->stageEv[i]) (void) hipEventDestroy(ctx->stageEv[i]);
  }
  for (auto& kv : ctx->windows) (void) hipFree(kv.second);
  for (auto& kv : ctx->twiddles) (void) hipFree(kv.second);
  for (auto& r : ctx->profRecs) { (void) hipEventDestroy(r.start); (void) hipEventDestroy(r.stop); }
  for (auto e : ctx->eventPool) (void) hipEventDestroy(e);
  if (ctx->copyStream) (void) hipStreamDestroy(ctx->copyStream);
  if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
  g_pool.trim(ctx->device); // cached device blocks go with the context
  delete ctx;
}

const char* fluhip_last_error(const fluhip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int fluhip_ctx_device_info(const fluhip_ctx* ctx, char* name, int name_len, char* arch,
                           int arch_len, int* compute_units)
{
  if (!ctx) return FLUHIP_ERROR;
  if (name && name_len > 0) { std::snprintf(name, (size_t) name_len, "%s", ctx->props.name); }
  if (arch && arch_len > 0) { std::snprintf(arch, (size_t) arch_len, "%s", ctx->props.gcnArchName); }
  if (compute_units) *compute_units = ctx->props.multiProcessorCount;
  return FLUHIP_OK;
}

void* fluhip_ctx_stream(const fluhip_ctx* ctx) { return ctx ? (void*) ctx->stream : nullptr; }

int fluhip_ctx_trim(fluhip_ctx* ctx)
{
  if (!ctx) return FLUHIP_ERROR;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  g_pool.trim(ctx->device);
  return FLUHIP_OK;
}

int fluhip_ctx_set_progress_lag(fluhip_ctx* ctx, int lag)
{
  if (!ctx) return FLUHIP_ERROR;
  if (lag < 1 || lag > 64) return fail(ctx, "progress lag must be 1 .. 64");
  ctx->progressLag = lag;
  return FLUHIP_OK;
}

int fluhip_ctx_synchronize(fluhip_ctx* ctx)
{
  if (!ctx) return FLUHIP_ERROR;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLUHIP_OK;
}

int fluhip_fft_params(int64_t win, int64_t hop, int64_t fft, int64_t* win_out, int64_t* hop_out,
                      int64_t* fft_out, int64_t* bins_out)
{
  // clients/common/ParameterTypes.hpp:295-312
  if (win < 4) return FLUHIP_ERROR;
  int64_t h = hop > 0 ? hop : win >> 1;
  int64_t f = fft;
  if (f < 0)
  {
    f = 1;
    while (f < win) f <<= 1; // nextPow2(win, up)
  }
  if ((f & (f - 1)) || f < win) return FLUHIP_ERROR;
  if (win_out) *win_out = win;
  if (hop_out) *hop_out = h;
  if (fft_out) *fft_out = f;
  if (bins_out) *bins_out = (f >> 1) + 1;
  return FLUHIP_OK;
}

int64_t fluhip_stft_num_frames(int64_t n, int64_t win, int64_t hop)
{
  (void) win;
  return hop > 0 ? (n + hop) / hop : 0;
}

// ---- corpus ---------------------------------------------------------------------------
int fluhip_corpus_create(fluhip_ctx* ctx, int64_t count, int64_t n, int64_t win, int64_t fft,
                         int64_t hop, int64_t K, fluhip_corpus** out)
{
  if (!ctx || !out) return FLUHIP_ERROR;
  *out = nullptr;
  if (count < 1) return fail(ctx, "corpus must hold at least one buffer");
  if (count > 65535) return fail(ctx, "a corpus holds at most 65535 buffers (split larger corpora into several)");
  int rc = check_shape(ctx, n, win, fft, hop, K);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<fluhip_corpus> c(new fluhip_corpus);
  c->ctx = ctx; c->B = count; c->n = n; c->win = win; c->fft = fft; c->hop = hop; c->K = K;
  rc = corpus_alloc(ctx, c.get());
  if (rc) return rc;
  *out = c.release();
  return FLUHIP_OK;
}

int fluhip_corpus_create_ragged(fluhip_ctx* ctx, int64_t count, const int64_t* n, int64_t win, int64_t fft, int64_t hop,
                                int64_t K, fluhip_corpus** out)
{
  if (!ctx || !out) return FLUHIP_ERROR;
  *out = nullptr;
  if (!n || count < 1) return fail(ctx, "corpus must hold at least one buffer");
  if (count > 65535) return fail(ctx, "a corpus holds at most 65535 buffers (split larger corpora into several)");
  int64_t nmax = 0;
  for (int64_t i = 0; i < count; i++)
  {
    if (n[i] < 1) return fail(ctx, "buffer " + std::to_string(i) + ": not enough frames");
    nmax = std::max(nmax, n[i]);
  }
  int rc = check_shape(ctx, nmax, win, fft, hop, K);
  if (rc) return rc;
  // one set of launches over buffers of different lengths needs the work-list form of the factor-update kernel (padded
  // rank 16 / 32 / 64 / 128) and the block form of the STFT (it takes per-buffer lengths)
  if (update_variant((int) padded_rank(K)) != 5) return fail(ctx, "ragged corpora support ranks up to 128");
  if (!(fft == 1024 || fft == 2048 || fft == 4096) || (win % 2) != 0 || win > fft)
    return fail(ctx, "ragged corpora need an STFT shape with a block form (fft 1024 / 2048 / 4096, even window)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<fluhip_corpus> c(new fluhip_corpus);
  c->ctx = ctx; c->B = count; c->n = nmax; c->win = win; c->fft = fft; c->hop = hop; c->K = K;
  c->ragged = true;
  c->nOf.assign(n, n + count);
  c->tOf.resize((size_t) count);
  for (int64_t i = 0; i < count; i++) c->tOf[(size_t) i] = (int) ((n[i] + hop) / hop); // alg/STFT.hpp:98-99 per buffer
  rc = corpus_alloc(ctx, c.get());
  if (rc) return rc;
  *out = c.release();
  return FLUHIP_OK;
}

int64_t fluhip_corpus_frames_of(const fluhip_corpus* c, int64_t i)
{
  if (!c || i < 0 || i >= c->B) return 0;
  return c->ragged ? c->tOf[(size_t) i] : c->T;
}

int fluhip_corpus_set_audio_ragged_host(fluhip_corpus* c, const float* const* audio)
{
  if (!c || !audio) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (!c->ragged) return fail(ctx, "not a ragged corpus");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t) c->B * c->n * sizeof(float);
  if (c->audioOwn.bytes < bytes) HIPCHK(ctx, c->audioOwn.alloc(bytes, false, ctx->stream));
  // every buffer at its slot of the longest buffer's stride; the kernel never reads past a buffer's own length
  for (int64_t i = 0; i < c->B; i++)
  {
    if (!audio[i]) return fail(ctx, "buffer " + std::to_string(i) + ": null audio");
    HIPCHK(ctx, hipMemcpyAsync(c->audioOwn.as<float>() + i * c->n, audio[i], (size_t) c->nOf[(size_t) i] * sizeof(float),
                               hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  c->audioDev = c->audioOwn.as<float>();
  return FLUHIP_OK;
}

int fluhip_corpus_writeback_ragged_host(fluhip_corpus* c, float* const* bases, float* const* acts)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf db, da;
  const size_t nb = (size_t) c->B * c->K * c->F * sizeof(float), na = (size_t) c->B * c->K * c->T * sizeof(float);
  if (bases) HIPCHK(ctx, db.alloc(nb, false, ctx->stream));
  if (acts) HIPCHK(ctx, da.alloc(na, false, ctx->stream));
  int rc = fluhip_corpus_writeback_dev(c, bases ? db.as<float>() : nullptr, acts ? da.as<float>() : nullptr);
  if (rc) return rc;
  for (int64_t i = 0; i < c->B; i++)
  {
    const int64_t Ti = fluhip_corpus_frames_of(c, i);
    if (bases && bases[i])
      HIPCHK(ctx, hipMemcpyAsync(bases[i], db.as<float>() + i * c->K * c->F, (size_t) c->K * c->F * sizeof(float),
                                 hipMemcpyDeviceToHost, ctx->stream));
    if (acts && acts[i]) // K rows of the buffer's own T_i frames out of rows of the longest buffer's length
      HIPCHK(ctx, hipMemcpy2DAsync(acts[i], (size_t) Ti * sizeof(float), da.as<float>() + i * c->K * c->T,
                                   (size_t) c->T * sizeof(float), (size_t) Ti * sizeof(float), (size_t) c->K,
                                   hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLUHIP_OK;
}

void fluhip_corpus_destroy(fluhip_corpus* c)
{
  if (!c) return;
  (void) hipSetDevice(c->ctx->device);
  (void) hipStreamSynchronize(c->ctx->stream);
  delete c;
}

int64_t fluhip_corpus_frames(const fluhip_corpus* c) { return c ? c->T : 0; }
int64_t fluhip_corpus_bins(const fluhip_corpus* c) { return c ? c->F : 0; }
int64_t fluhip_corpus_device_bytes(const fluhip_corpus* c) { return c ? c->device_bytes() : 0; }

int fluhip_corpus_set_audio_host(fluhip_corpus* c, const float* audio)
{
  if (!c || !audio) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t) c->B * c->n * sizeof(float);
  if (c->audioOwn.bytes < bytes) HIPCHK(ctx, c->audioOwn.alloc(bytes, false, ctx->stream));
  // The copy runs on the context's COPY stream: whatever another corpus of this context has in flight on the compute stream
  // (the previous slice of a pool job: its iterations) goes on beside it -- SURVEY section 7 step 5's double-buffered upload.
  // A corpus whose own audio may still be read by enqueued work drains the compute stream first.
  if (c->touched) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (!ctx->copyStream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copyStream, hipStreamNonBlocking));
  HIPCHK(ctx, hipMemcpyAsync(c->audioOwn.p, audio, bytes, hipMemcpyHostToDevice, ctx->copyStream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->copyStream)); // the caller's buffer is free again; the device copy is complete
  c->audioDev = c->audioOwn.as<float>();
  return FLUHIP_OK;
}

int fluhip_corpus_set_audio_dev(fluhip_corpus* c, const float* audio_dev)
{
  if (!c || !audio_dev) return FLUHIP_ERROR;
  c->audioDev = audio_dev;
  return FLUHIP_OK;
}

int fluhip_corpus_stft(fluhip_corpus* c)
{
  if (!c) return FLUHIP_ERROR;
  if (!c->audioDev) return fail(c->ctx, "corpus has no audio");
  HIPCHK(c->ctx, hipSetDevice(c->ctx->device));
  return corpus_stft(c, c->audioDev, nullptr, c->n);
}

int fluhip_corpus_nmf(fluhip_corpus* c, int64_t iters, int update_w, int update_h, int64_t seed,
                      const int64_t* seeds, fluhip_progress_fn progress, void* user)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (!c->haveMag) return fail(ctx, "corpus has no spectrogram: call fluhip_corpus_stft first");
  if (iters < 0) return fail(ctx, "negative iteration count");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  FactorInit fi;
  if (!c->seedW32.empty()) fi.W0f32 = c->seedW32.data(); // nrt/NMFClient.hpp:246-258 -> alg/NMF.hpp:102-112
  if (!c->seedH32.empty()) fi.H0f32 = c->seedH32.data(); // :113-124
  int rc = corpus_init_factors(c, seed, seeds, fi);
  if (rc) return rc;
  return corpus_iterate(c, iters, update_w != 0, update_h != 0, progress, user);
}

int fluhip_corpus_set_factors(fluhip_corpus* c, const float* bases_seed, const float* acts_seed)
{
  if (!c) return FLUHIP_ERROR;
  const size_t nw = (size_t) c->B * c->K * c->F, nh = (size_t) c->B * c->K * c->T;
  if (bases_seed) c->seedW32.assign(bases_seed, bases_seed + nw);
  else std::vector<float>().swap(c->seedW32);
  if (acts_seed) c->seedH32.assign(acts_seed, acts_seed + nh);
  else std::vector<float>().swap(c->seedH32);
  return FLUHIP_OK;
}

int fluhip_corpus_writeback_dev(fluhip_corpus* c, float* bases_dev, float* acts_dev)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (!c->haveFactors) return fail(ctx, "corpus has no factors: call fluhip_corpus_nmf first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (bases_dev) // clients/nrt/NMFClient.hpp:277-283
    launch_gather_w_f32(c->Wf.as<double>(), c->Fp * c->Kp, bases_dev, c->K * c->F, (int) c->F,
                        (int) c->K, (int) c->Kp, (int) c->B, s);
  if (acts_dev) // :286-300
    launch_acts_f32(c->H1.as<double>(), c->Tp * c->Kp, acts_dev, c->K * c->T, (int) c->T, (int) c->K,
                    (int) c->Kp, (int) c->B, c->hmax.as<double>(), s);
  HIPCHK(ctx, hipGetLastError());
  return FLUHIP_OK;
}

// clients/nrt/NMFClient.hpp:302-334 for every buffer of the corpus: estimate -> ratio mask -> ISTFT per component
int fluhip_corpus_keep_spectrum(fluhip_corpus* c, int on)
{
  if (!c) return FLUHIP_ERROR;
  c->keepSpec = on != 0;
  if (!c->keepSpec) c->spec.release();
  c->haveMag = c->haveMag && !(c->keepSpec && !c->spec.p); // a later resynthesis needs the STFT to run again
  return FLUHIP_OK;
}

int fluhip_corpus_resynth_dev(fluhip_corpus* c, float* out_dev)
{
  if (!c || !out_dev) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (!c->haveFactors) return fail(ctx, "corpus has no factors: call fluhip_corpus_nmf first");
  if (!c->keepSpec || !c->spec.p)
    return fail(ctx, "resynthesis needs the complex spectrogram: fluhip_corpus_keep_spectrum(c, 1) before fluhip_corpus_stft");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const double *wtab = nullptr, *ttab = nullptr;
  int rc = get_window(ctx, c->win, c->fft, c->windowType, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, c->fft, &ttab);
  if (rc) return rc;
  // Batched form (kernels_stft2.hip resynth_seq_kernel): every component of a group of buffers in one launch, the
  // overlap-add in registers -- no windowed frames in memory (the per-buffer path below writes and re-reads K T win doubles
  // per buffer: 58 GB each way on the bench shard, 85 ms against 113 ms for 200 iterations).  FLUHIP_RESYNTH_BATCH=0 off.
  static const int batchEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_RESYNTH_BATCH"); return e ? std::atoi(e) : 1; }();
  if (batchEnv != 0 && resynth_batch_supported((int) c->win, (int) c->fft, (int) c->hop))
  {
    // buffers per launch: the reciprocal V-hat of a group within ~1 GiB
    const int64_t perBuf = c->T * c->F * (int64_t) sizeof(double);
    const int64_t group = std::max<int64_t>(1, std::min<int64_t>(c->B, ((int64_t) 1 << 30) / perBuf));
    DevBuf mult, wt, nrm;
    HIPCHK(ctx, mult.alloc((size_t) (group * perBuf), false, s));
    HIPCHK(ctx, wt.alloc((size_t) (group * c->Kp * c->F) * sizeof(double), false, s));
    HIPCHK(ctx, nrm.alloc((size_t) c->hop * sizeof(double), false, s));
    launch_resynth_normaliser(wtab, (int) c->win, (int) c->hop, nrm.as<double>(), s);
    for (int64_t b0 = 0; b0 < c->B; b0 += group)
    {
      const int nb = (int) std::min(group, c->B - b0);
      const double* Wb = c->Wf.as<double>() + b0 * c->Fp * c->Kp;
      const double* Hb = c->H1.as<double>() + b0 * c->Tp * c->Kp;
      launch_resynth_mult(Wb, c->Fp * c->Kp, Hb, c->Tp * c->Kp, wt.as<double>(), mult.as<double>(), (int) c->T, (int) c->F,
                          (int) c->K, (int) c->Kp, nb, s);
      ResynthBatchArgs ra;
      ra.spec = c->spec.as<double>() + b0 * c->T * c->F * 2; ra.specStride = c->T * c->F * 2;
      ra.mult = mult.as<double>(); ra.multStride = c->T * c->F;
      ra.Wt = wt.as<double>(); ra.wtStride = c->Kp * c->F;
      ra.H1 = Hb; ra.hStride = c->Tp * c->Kp;
      ra.Kp = (int) c->Kp; ra.K = (int) c->K;
      ra.win = (int) c->win; ra.fft = (int) c->fft; ra.hop = (int) c->hop; ra.T = (int) c->T; ra.F = (int) c->F; ra.B = nb;
      ra.window = wtab; ra.twiddle = ttab; ra.nrmTab = nrm.as<double>();
      ra.out32 = out_dev + b0 * c->K * c->n; ra.n = c->n; ra.outStride = c->n; ra.trim = c->win / 2;
      ra.nTab = c->ragged ? c->nTab.as<int64_t>() + b0 : nullptr;
      if (!launch_resynth_batch(ra, s)) return fail(ctx, "batched resynthesis refused a shape it had accepted");
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(s)); // the workspaces go out of scope
    return FLUHIP_OK;
  }
  DevBuf vhat, frames;
  HIPCHK(ctx, vhat.alloc((size_t) c->T * c->F * sizeof(double), false, s));
  const int64_t compsPerLaunch = std::max<int64_t>(1, std::min<int64_t>(c->K, ((int64_t) 1 << 30) / (c->T * c->win * 8)));
  HIPCHK(ctx, frames.alloc((size_t) compsPerLaunch * c->T * c->win * sizeof(double), false, s));
  for (int64_t b = 0; b < c->B; b++)
  {
    const double* Wb = c->Wf.as<double>() + b * c->Fp * c->Kp;
    const double* Hb = c->H1.as<double>() + b * c->Tp * c->Kp;
    // a ragged corpus: the buffer's own frames and samples (arrays are strided by the longest buffer's)
    const int Tb = c->ragged ? c->tOf[(size_t) b] : (int) c->T;
    const int64_t nb = c->ragged ? c->nOf[(size_t) b] : c->n;
    launch_vhat(Wb, 0, Hb, 0, vhat.as<double>(), c->F, 0, Tb, (int) c->F, (int) c->Kp, 1, s);
    ResynthArgs ra;
    ra.spec = c->spec.as<double>() + b * c->T * c->F * 2; ra.Wf = Wb; ra.H1 = Hb;
    ra.Vhat = vhat.as<double>(); ra.ldV = c->F; ra.Kp = (int) c->Kp;
    ra.win = (int) c->win; ra.fft = (int) c->fft; ra.hop = (int) c->hop; ra.T = Tb; ra.F = (int) c->F;
    ra.window = wtab; ra.twiddle = ttab; ra.frames = frames.as<double>(); ra.out = nullptr; ra.n = nb; ra.outStride = c->n;
    ra.trim = c->win / 2;
    for (int64_t k = 0; k < c->K; k += compsPerLaunch)
    {
      ra.k = (int) k;
      ra.nComp = (int) std::min(compsPerLaunch, c->K - k);
      ra.out32 = out_dev + (b * c->K + k) * c->n;
      ra.bigScratch = big_fft_scratch(ctx, ra.win, ra.fft, ra.T);
      if (stft_needs_scratch(ra.win, ra.fft) && !ra.bigScratch) return FLUHIP_ERROR;
      launch_resynth(ra, s);
    }
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(s)); // vhat / frames go out of scope
  return FLUHIP_OK;
}

int fluhip_corpus_resynth_host(fluhip_corpus* c, float* out)
{
  if (!c || !out) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf d;
  const size_t nb = (size_t) c->B * c->K * c->n * sizeof(float);
  HIPCHK(ctx, d.alloc(nb, false, ctx->stream));
  int rc = fluhip_corpus_resynth_dev(c, d.as<float>());
  if (rc) return rc;
  return copy_to_host(ctx, out, nb, d.p, nb, nb, 1, ctx->stream);
}

// The same, written the way an interleaved host buffer holds it (MemoryBufferAdaptor, SuperCollider and Max buffers:
// frames x channels): out[t * frame_stride + b * K + k] = component k of buffer b at sample t -- the layout of
// resynth.samps(i * rank + j) in clients/nrt/NMFClient.hpp:321-326 when the buffer's channels are interleaved.  The
// transposition happens on the device; the host sees one streaming copy instead of count x K strided passes over its buffer.
int fluhip_corpus_resynth_interleaved_host(fluhip_corpus* c, float* out, int64_t frame_stride)
{
  if (!c || !out) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (c->ragged) return fail(ctx, "interleaved resynthesis needs equal-length buffers");
  const int64_t chans = c->B * c->K;
  if (frame_stride < chans) return fail(ctx, "frame stride below count x K");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  DevBuf d, dt;
  const size_t nb = (size_t) chans * c->n * sizeof(float);
  HIPCHK(ctx, d.alloc(nb, false, s));
  HIPCHK(ctx, dt.alloc(nb, false, s));
  int rc = fluhip_corpus_resynth_dev(c, d.as<float>());
  if (rc) return rc;
  launch_transpose_f32(d.as<float>(), c->n, dt.as<float>(), chans, (int) chans, c->n, s); // [chans][n] -> [n][chans]
  HIPCHK(ctx, hipGetLastError());
  const size_t width = (size_t) chans * sizeof(float);
  if (frame_stride == chans) return copy_to_host(ctx, out, nb, dt.p, nb, nb, 1, s);
  return copy_to_host(ctx, out, (size_t) frame_stride * sizeof(float), dt.p, width, width, (size_t) c->n, s);
}

int fluhip_corpus_resynth_ragged_host(fluhip_corpus* c, float* const* out)
{
  if (!c || !out) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf d;
  const size_t nb = (size_t) c->B * c->K * c->n * sizeof(float);
  HIPCHK(ctx, d.alloc(nb, true, ctx->stream));
  int rc = fluhip_corpus_resynth_dev(c, d.as<float>());
  if (rc) return rc;
  for (int64_t i = 0; i < c->B; i++)
  {
    if (!out[i]) continue;
    const int64_t ni = c->ragged ? c->nOf[(size_t) i] : c->n; // K rows of the buffer's own samples out of rows of the longest
    HIPCHK(ctx, hipMemcpy2DAsync(out[i], (size_t) ni * sizeof(float), d.as<float>() + i * c->K * c->n, (size_t) c->n * sizeof(float),
                                 (size_t) ni * sizeof(float), (size_t) c->K, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLUHIP_OK;
}

int fluhip_corpus_writeback_host(fluhip_corpus* c, float* bases, float* acts)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf db, da;
  const size_t nb = (size_t) c->B * c->K * c->F * sizeof(float), na = (size_t) c->B * c->K * c->T * sizeof(float);
  if (bases) HIPCHK(ctx, db.alloc(nb, false, ctx->stream));
  if (acts) HIPCHK(ctx, da.alloc(na, false, ctx->stream));
  int rc = fluhip_corpus_writeback_dev(c, bases ? db.as<float>() : nullptr, acts ? da.as<float>() : nullptr);
  if (rc) return rc;
  // (large corpora: through the pinned staging blocks, copy_to_host)
  if (bases && (rc = copy_to_host(ctx, bases, nb, db.p, nb, nb, 1, ctx->stream))) return rc;
  if (acts && (rc = copy_to_host(ctx, acts, na, da.p, na, na, 1, ctx->stream))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLUHIP_OK;
}

int fluhip_corpus_read_f64(fluhip_corpus* c, double* mag, double* W1, double* H1)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (mag)
  {
    if (!c->haveMag) return fail(ctx, "corpus has no spectrogram");
    for (int64_t b = 0; b < c->B; b++)
      HIPCHK(ctx, hipMemcpy2DAsync(mag + b * c->T * c->F, (size_t) c->F * sizeof(double),
                                   c->mag.as<double>() + b * c->Tp * c->Fp, (size_t) c->Fp * sizeof(double),
                                   (size_t) c->F * sizeof(double), (size_t) c->T, hipMemcpyDeviceToHost, s));
  }
  DevBuf dw, dh;
  if (W1 || H1)
    if (!c->haveFactors) return fail(ctx, "corpus has no factors");
  if (W1)
  {
    const size_t bytes = (size_t) c->B * c->K * c->F * sizeof(double);
    HIPCHK(ctx, dw.alloc(bytes, false, s));
    launch_gather_w_f64(c->Wf.as<double>(), c->Fp * c->Kp, dw.as<double>(), c->K * c->F, (int) c->F,
                        (int) c->K, (int) c->Kp, (int) c->B, s);
    HIPCHK(ctx, hipMemcpyAsync(W1, dw.p, bytes, hipMemcpyDeviceToHost, s));
  }
  if (H1)
  {
    const size_t bytes = (size_t) c->B * c->T * c->K * sizeof(double);
    HIPCHK(ctx, dh.alloc(bytes, false, s));
    launch_gather_h_f64(c->H1.as<double>(), c->Tp * c->Kp, dh.as<double>(), c->T * c->K, (int) c->T,
                        (int) c->K, (int) c->Kp, (int) c->B, s);
    HIPCHK(ctx, hipMemcpyAsync(H1, dh.p, bytes, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(ctx, hipStreamSynchronize(s));
  return FLUHIP_OK;
}

// ---- algorithm-level single-buffer entry points -----------------------------------------
static int stft_common(fluhip_ctx* ctx, const float* a32, const double* a64, int64_t n, int64_t stride,
                       int64_t win, int64_t fft, int64_t hop, int window_type, double* spec,
                       double* mag, int64_t* frames_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!a32 && !a64) return fail(ctx, "null audio");
  if (stride < 1) return fail(ctx, "stride must be >= 1");
  int rc = check_shape(ctx, n, win, fft, hop, 1);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.n = n; c.win = win; c.fft = fft; c.hop = hop; c.K = 1;
  c.windowType = window_type;
  c.keepSpec = spec != nullptr;
  rc = corpus_alloc(ctx, &c);
  if (rc) return rc;
  // strided host view -> contiguous device copy (clients/nrt/NMFClient.hpp:240 `tmp <<= samps(...)`)
  DevBuf in;
  const size_t esz = a32 ? sizeof(float) : sizeof(double);
  HIPCHK(ctx, in.alloc((size_t) n * esz, false, ctx->stream));
  HIPCHK(ctx, upload_strided(in.p, a32 ? (const void*) a32 : (const void*) a64, (size_t) n, (size_t) stride, esz,
                             ctx->stream));
  rc = corpus_stft(&c, a32 ? in.as<float>() : nullptr, a32 ? nullptr : in.as<double>(), n);
  if (rc) return rc;
  if (frames_out) *frames_out = c.T;
  // (long buffers: through the pinned staging blocks -- a minute of audio at fft 2048 is 42 + 85 MB)
  if (mag && (rc = copy_to_host(ctx, mag, (size_t) c.F * sizeof(double), c.mag.p, (size_t) c.Fp * sizeof(double),
                                (size_t) c.F * sizeof(double), (size_t) c.T, ctx->stream)))
    return rc;
  if (spec)
  {
    const size_t nb = (size_t) c.T * c.F * 2 * sizeof(double);
    if ((rc = copy_to_host(ctx, spec, nb, c.spec.p, nb, nb, 1, ctx->stream))) return rc;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLUHIP_OK;
}

int fluhip_stft_f64(fluhip_ctx* ctx, const double* audio, int64_t n, int64_t stride, int64_t win,
                    int64_t fft, int64_t hop, int window_type, double* spec, double* mag,
                    int64_t* frames_out)
{
  return stft_common(ctx, nullptr, audio, n, stride, win, fft, hop, window_type, spec, mag, frames_out);
}

int fluhip_stft_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride, int64_t win,
                    int64_t fft, int64_t hop, int window_type, double* spec, double* mag,
                    int64_t* frames_out)
{
  return stft_common(ctx, audio, nullptr, n, stride, win, fft, hop, window_type, spec, mag, frames_out);
}

// ---- two-stride views at the algorithm boundary (data/FluidTensor_Support.hpp:260-420, util/FluidEigenMappings.hpp:35-225)
} // extern "C" (the helpers below are C++)
namespace {
inline bool view_empty(const fluhip_matrix_view* v) { return !v || !v->data || v->rows == 0 || v->cols == 0; }
// contiguous row-major host copy of a view (small matrices: seeds)
std::vector<double> view_gather(const fluhip_matrix_view& v)
{
  std::vector<double> out((size_t) (v.rows * v.cols));
  for (int64_t r = 0; r < v.rows; r++)
    for (int64_t c = 0; c < v.cols; c++) out[(size_t) (r * v.cols + c)] = v.data[r * v.row_stride + c * v.col_stride];
  return out;
}
void view_scatter(const fluhip_matrix_view& v, const double* src)
{
  for (int64_t r = 0; r < v.rows; r++)
    for (int64_t c = 0; c < v.cols; c++) v.data[r * v.row_stride + c * v.col_stride] = src[r * v.cols + c];
}
} // namespace
extern "C" {

int fluhip_nmf_process_views_f64(fluhip_ctx* ctx, const fluhip_matrix_view* Xv, int64_t K, int64_t iters, int update_w,
                                 int update_h, int64_t seed, const fluhip_matrix_view* W0v, const fluhip_matrix_view* H0v,
                                 const fluhip_matrix_view* W1v, const fluhip_matrix_view* H1v, const fluhip_matrix_view* V1v,
                                 fluhip_progress_fn progress, void* user)
{
  if (!ctx) return FLUHIP_ERROR;
  if (view_empty(Xv)) return fail(ctx, "bad input matrix");
  const int64_t T = Xv->rows, F = Xv->cols;
  if (K < 1) return fail(ctx, "rank must be >= 1");
  if (iters < 0) return fail(ctx, "negative iteration count");
  if (int rcr = check_rank(ctx, T, F, K)) return rcr;
  if ((Xv->col_stride == 1 && Xv->row_stride < F) || (Xv->row_stride == 1 && Xv->col_stride < T && Xv->col_stride != 1))
    return fail(ctx, "bad input matrix");
  // alg/NMF.hpp:109-110, 121-122 assert these shapes
  if (!view_empty(W0v) && (W0v->rows != K || W0v->cols != F)) return fail(ctx, "W0 must be rank x bins");
  if (!view_empty(H0v) && (H0v->rows != T || H0v->cols != K)) return fail(ctx, "H0 must be frames x rank");
  if (!view_empty(W1v) && (W1v->rows != K || W1v->cols != F)) return fail(ctx, "W1 must be rank x bins");
  if (!view_empty(H1v) && (H1v->rows != T || H1v->cols != K)) return fail(ctx, "H1 must be frames x rank");
  if (!view_empty(V1v) && (V1v->rows != T || V1v->cols != F)) return fail(ctx, "V1 must be frames x bins");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.K = K;
  // shape the corpus directly from the matrix extents (no audio behind it)
  c.hop = 1; c.n = T - 1; c.fft = (F - 1) * 2; c.win = c.fft;
  c.T = T; c.F = F;
  c.Tp = round_up(T, 32); c.Fp = round_up(F, 32); c.Kp = padded_rank(K);
  {
    HIPCHK(ctx, c.mag.alloc((size_t) c.Tp * c.Fp * sizeof(double), true, s));
    HIPCHK(ctx, c.magT.alloc((size_t) c.Fp * c.Tp * sizeof(double), true, s));
    HIPCHK(ctx, c.Wf.alloc((size_t) c.Fp * c.Kp * sizeof(double), true, s));
    HIPCHK(ctx, c.H1.alloc((size_t) c.Tp * c.Kp * sizeof(double), true, s));
    HIPCHK(ctx, c.hmax.alloc(sizeof(double), true, s));
    if (int rc2 = plan_updates(ctx, &c)) return rc2;
  }
  // alg/NMF.hpp:125  V = X^T.  A view with unit column stride is the T x F row-major image the frame-major copy wants;
  // a view with unit ROW stride (FluidTensorView::transpose() of an F x T matrix) is byte for byte the bin-major copy:
  // either goes up as one strided 2-D copy and the other layout is made on the device.  Anything else (both strides
  // non-unit) is gathered on the host first.
  // (a one-row view with a non-unit column stride is NOT a contiguous row: the frame-major path needs unit column stride
  //  or a single column; such a view is byte for byte a bin-major image of one frame and takes the second path)
  std::vector<double> xtmp;
  if (Xv->col_stride == 1 || F == 1)
  {
    HIPCHK(ctx, hipMemcpy2DAsync(c.mag.p, (size_t) c.Fp * sizeof(double), Xv->data, (size_t) Xv->row_stride * sizeof(double),
                                 (size_t) F * sizeof(double), (size_t) T, hipMemcpyHostToDevice, s));
    launch_transpose(c.mag.as<double>(), c.Fp, c.Tp * c.Fp, c.magT.as<double>(), c.Tp, c.Fp * c.Tp, (int) T, (int) F, 1, s);
  }
  else if (Xv->row_stride == 1 || T == 1)
  {
    HIPCHK(ctx, hipMemcpy2DAsync(c.magT.p, (size_t) c.Tp * sizeof(double), Xv->data, (size_t) Xv->col_stride * sizeof(double),
                                 (size_t) T * sizeof(double), (size_t) F, hipMemcpyHostToDevice, s));
    launch_transpose(c.magT.as<double>(), c.Tp, c.Fp * c.Tp, c.mag.as<double>(), c.Fp, c.Tp * c.Fp, (int) F, (int) T, 1, s);
  }
  else
  {
    xtmp = view_gather(*Xv);
    HIPCHK(ctx, hipMemcpy2DAsync(c.mag.p, (size_t) c.Fp * sizeof(double), xtmp.data(), (size_t) F * sizeof(double),
                                 (size_t) F * sizeof(double), (size_t) T, hipMemcpyHostToDevice, s));
    launch_transpose(c.mag.as<double>(), c.Fp, c.Tp * c.Fp, c.magT.as<double>(), c.Tp, c.Fp * c.Tp, (int) T, (int) F, 1, s);
  }
  c.haveMag = true;
  // seeds are small (K x F, T x K): contiguous host images whatever their strides
  std::vector<double> w0tmp, h0tmp;
  FactorInit fi;
  fi.sharedW = fi.sharedH = true;
  if (!view_empty(W0v))
  {
    if (W0v->col_stride == 1 && W0v->row_stride == F) fi.W0host = W0v->data;
    else { w0tmp = view_gather(*W0v); fi.W0host = w0tmp.data(); }
  }
  if (!view_empty(H0v))
  {
    if (H0v->col_stride == 1 && H0v->row_stride == K) fi.H0host = H0v->data;
    else { h0tmp = view_gather(*H0v); fi.H0host = h0tmp.data(); }
  }
  int rc = corpus_init_factors(&c, seed, nullptr, fi);
  if (rc) return rc;
  rc = corpus_iterate(&c, iters, update_w != 0, update_h != 0, progress, user);
  if (rc != FLUHIP_OK && rc != FLUHIP_CANCELLED) return rc;
  const bool cancelled = rc == FLUHIP_CANCELLED;
  // alg/NMF.hpp:127-133 outputs; :182 V = W*H only when the loop ran to completion
  DevBuf dw, dh, dv, dvt;
  std::vector<double> w1tmp, h1tmp, v1tmp;
  if (!view_empty(W1v))
  {
    const bool direct = W1v->col_stride == 1 && W1v->row_stride == F;
    if (!direct) w1tmp.resize((size_t) (K * F));
    HIPCHK(ctx, dw.alloc((size_t) K * F * sizeof(double), false, s));
    launch_gather_w_f64(c.Wf.as<double>(), 0, dw.as<double>(), 0, (int) F, (int) K, (int) c.Kp, 1, s);
    HIPCHK(ctx, hipMemcpyAsync(direct ? W1v->data : w1tmp.data(), dw.p, (size_t) K * F * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (!view_empty(H1v))
  {
    const bool direct = H1v->col_stride == 1 && H1v->row_stride == K;
    if (!direct) h1tmp.resize((size_t) (T * K));
    HIPCHK(ctx, dh.alloc((size_t) T * K * sizeof(double), false, s));
    launch_gather_h_f64(c.H1.as<double>(), 0, dh.as<double>(), 0, (int) T, (int) K, (int) c.Kp, 1, s);
    HIPCHK(ctx, hipMemcpyAsync(direct ? H1v->data : h1tmp.data(), dh.p, (size_t) T * K * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (!view_empty(V1v) && !cancelled)
  {
    HIPCHK(ctx, dv.alloc((size_t) T * F * sizeof(double), false, s));
    launch_vhat(c.Wf.as<double>(), 0, c.H1.as<double>(), 0, dv.as<double>(), F, 0, (int) T, (int) F,
                (int) c.Kp, 1, s);
    if (V1v->col_stride == 1 || F == 1)
      HIPCHK(ctx, hipMemcpy2DAsync(V1v->data, (size_t) V1v->row_stride * sizeof(double), dv.p, (size_t) F * sizeof(double),
                                   (size_t) F * sizeof(double), (size_t) T, hipMemcpyDeviceToHost, s));
    else if (V1v->row_stride == 1 || T == 1)
    {
      // a transposed view: the F x T image, made on the device
      HIPCHK(ctx, dvt.alloc((size_t) F * T * sizeof(double), false, s));
      launch_transpose(dv.as<double>(), F, 0, dvt.as<double>(), T, 0, (int) T, (int) F, 1, s);
      HIPCHK(ctx, hipMemcpy2DAsync(V1v->data, (size_t) V1v->col_stride * sizeof(double), dvt.p, (size_t) T * sizeof(double),
                                   (size_t) T * sizeof(double), (size_t) F, hipMemcpyDeviceToHost, s));
    }
    else
    {
      v1tmp.resize((size_t) (T * F));
      HIPCHK(ctx, hipMemcpyAsync(v1tmp.data(), dv.p, (size_t) T * F * sizeof(double), hipMemcpyDeviceToHost, s));
    }
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(s));
  if (!w1tmp.empty()) view_scatter(*W1v, w1tmp.data());
  if (!h1tmp.empty()) view_scatter(*H1v, h1tmp.data());
  if (!v1tmp.empty()) view_scatter(*V1v, v1tmp.data());
  return cancelled ? FLUHIP_CANCELLED : FLUHIP_OK;
}

int fluhip_nmf_process_f64(fluhip_ctx* ctx, const double* X, int64_t T, int64_t F, int64_t ldx,
                           int64_t K, int64_t iters, int update_w, int update_h, int64_t seed,
                           const double* W0, const double* H0, double* W1, double* H1,
                           double* V1, fluhip_progress_fn progress, void* user)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!X || T < 1 || F < 1 || ldx < F) return fail(ctx, "bad input matrix");
  const fluhip_matrix_view xv{const_cast<double*>(X), T, F, ldx, 1};
  const fluhip_matrix_view w0{const_cast<double*>(W0), K, F, F, 1}, h0{const_cast<double*>(H0), T, K, K, 1};
  const fluhip_matrix_view w1{W1, K, F, F, 1}, h1{H1, T, K, K, 1}, v1{V1, T, F, F, 1};
  return fluhip_nmf_process_views_f64(ctx, &xv, K, iters, update_w, update_h, seed, W0 ? &w0 : nullptr, H0 ? &h0 : nullptr,
                                      W1 ? &w1 : nullptr, H1 ? &h1 : nullptr, V1 ? &v1 : nullptr, progress, user);
}

int fluhip_nmf_process_frames_f64(fluhip_ctx* ctx, const double* X, int64_t T, int64_t F, int64_t ldx,
                                  const double* W0, int64_t K, int64_t iters, int64_t seed, double* H, double* V)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!X || T < 1 || F < 1 || ldx < F) return fail(ctx, "bad input matrix");
  if (!W0 || K < 1) return fail(ctx, "bad dictionary");
  if (iters < 0) return fail(ctx, "negative iteration count");
  if (int rcr = check_rank(ctx, T, F, K)) return rcr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.K = K;
  c.hop = 1; c.n = T - 1; c.fft = (F - 1) * 2; c.win = c.fft;
  c.T = T; c.F = F;
  c.Tp = round_up(T, 32); c.Fp = round_up(F, 32); c.Kp = padded_rank(K);
  HIPCHK(ctx, c.mag.alloc((size_t) c.Tp * c.Fp * sizeof(double), true, s));
  HIPCHK(ctx, c.magT.alloc((size_t) c.Fp * c.Tp * sizeof(double), true, s));
  HIPCHK(ctx, c.Wf.alloc((size_t) c.Fp * c.Kp * sizeof(double), true, s));
  HIPCHK(ctx, c.H1.alloc((size_t) c.Tp * c.Kp * sizeof(double), true, s));
  if (int rc2 = plan_updates(ctx, &c)) return rc2;
  // :57-58, 61  v0 = max(x, eps)
  HIPCHK(ctx, hipMemcpy2DAsync(c.mag.p, (size_t) c.Fp * sizeof(double), X, (size_t) ldx * sizeof(double),
                               (size_t) F * sizeof(double), (size_t) T, hipMemcpyHostToDevice, s));
  launch_clamp_eps(c.mag.as<double>(), c.Fp, 0, (int) T, (int) F, 1, s);
  launch_transpose(c.mag.as<double>(), c.Fp, c.Tp * c.Fp, c.magT.as<double>(), c.Tp, c.Fp * c.Tp, (int) T,
                   (int) F, 1, s);
  // :59, 64-65  W = max(W, eps), every component divided by its L2 norm over the bins
  HIPCHK(ctx, c.stage.alloc((size_t) K * F * sizeof(double), false, s));
  HIPCHK(ctx, hipMemcpyAsync(c.stage.p, W0, (size_t) K * F * sizeof(double), hipMemcpyHostToDevice, s));
  launch_scatter_factor(c.stage.as<double>(), 0, c.Wf.as<double>(), c.Fp * c.Kp, (int) F, (int) K, (int) c.Kp, 1,
                        true, s);
  HIPCHK(ctx, c.normScratch.alloc((size_t) colnorm_scratch_doubles((int) F, (int) c.Kp, 1) * sizeof(double), false, s));
  launch_colnorm(c.Wf.as<double>(), c.Fp * c.Kp, (int) F, (int) K, (int) c.Kp, 1, true, false,
                 c.normScratch.as<double>(), s);
  // :55-56, 60  h = max(uniform(0,1)^K, eps): the same K draws for every frame when seeded
  std::vector<double> h0, rows((size_t) T * K);
  if (seed >= 0)
  {
    draw_uniform(seed, (size_t) K, h0);
    for (int64_t t = 0; t < T; t++) std::memcpy(&rows[(size_t) t * K], h0.data(), (size_t) K * sizeof(double));
  }
  else
    draw_uniform(seed, (size_t) T * K, rows);
  DevBuf hs;
  HIPCHK(ctx, hs.alloc((size_t) T * K * sizeof(double), false, s));
  HIPCHK(ctx, hipMemcpyAsync(hs.p, rows.data(), (size_t) T * K * sizeof(double), hipMemcpyHostToDevice, s));
  launch_scatter_factor(hs.as<double>(), 0, c.H1.as<double>(), c.Tp * c.Kp, (int) T, (int) K, (int) c.Kp, 1, false, s);
  launch_clamp_eps(c.H1.as<double>(), c.Kp, 0, (int) T, (int) K, 1, s);
  HIPCHK(ctx, hipStreamSynchronize(s)); // host staging vectors go out of use
  c.haveMag = c.haveFactors = true;
  // :71-79  nIterations of the H update
  int rc = corpus_iterate(&c, iters, false, true, nullptr, nullptr);
  if (rc != FLUHIP_OK) return rc;
  DevBuf dh, dv;
  if (H)
  {
    HIPCHK(ctx, dh.alloc((size_t) T * K * sizeof(double), false, s));
    launch_gather_h_f64(c.H1.as<double>(), 0, dh.as<double>(), 0, (int) T, (int) K, (int) c.Kp, 1, s);
    HIPCHK(ctx, hipMemcpyAsync(H, dh.p, (size_t) T * K * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (V) // :87  v = W^T h
  {
    HIPCHK(ctx, dv.alloc((size_t) T * F * sizeof(double), false, s));
    launch_vhat(c.Wf.as<double>(), 0, c.H1.as<double>(), 0, dv.as<double>(), F, 0, (int) T, (int) F, (int) c.Kp, 1, s);
    HIPCHK(ctx, hipMemcpyAsync(V, dv.p, (size_t) T * F * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(s));
  return FLUHIP_OK;
}


// ---------------------------------------------------------------------------------------
// NNDSVD (alg/NNDSVD.hpp) -- the SVD by one-sided Jacobi on the device (kernels_svd.hip), the O(k (F + T))
// construction on the host
// ---------------------------------------------------------------------------------------
// G: device [F][ldg], row f = bin f over the T frames (the transposed magnitude copy; destroyed).  Top factors
// to the host: s [min(F,T)] descending, U [k][F] (row j = u_j), VT [k][T]; k from the coverage rule.
static int nndsvd_device(fluhip_ctx* ctx, double* G, int64_t F, int64_t T, int64_t ldg, int64_t minRank, int64_t maxRank,
                         double amount, std::vector<double>& s, std::vector<double>& U, std::vector<double>& VT,
                         int64_t* kOut)
{
  hipStream_t st = ctx->stream;
  DevBuf dJ, dN, dFlag;
  HIPCHK(ctx, dJ.alloc((size_t) F * F * sizeof(double), false, st));
  HIPCHK(ctx, dN.alloc((size_t) F * sizeof(double), false, st));
  HIPCHK(ctx, dFlag.alloc(sizeof(unsigned), true, st));
  const int sweeps = launch_jacobi_svd(G, ldg, (int) F, (int) T, dJ.as<double>(), dN.as<double>(), dFlag.as<unsigned>(),
                                       40, st);
  HIPCHK(ctx, hipGetLastError());
  if (sweeps < 0) return fail(ctx, "the SVD did not converge");
  std::vector<double> norms((size_t) F);
  HIPCHK(ctx, hipMemcpyAsync(norms.data(), dN.p, (size_t) F * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  std::vector<int64_t> order((size_t) F);
  for (int64_t i = 0; i < F; i++) order[(size_t) i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return norms[(size_t) a] > norms[(size_t) b]; });
  const int64_t r = std::min(F, T);
  s.resize((size_t) r);
  for (int64_t i = 0; i < r; i++) s[(size_t) i] = norms[(size_t) order[(size_t) i]];
  // alg/NNDSVD.hpp:47-58
  int64_t k = 0;
  if (amount == 0) k = minRank;
  else
  {
    double current = 0, total = 0;
    for (double v : s) total += v;
    while ((current / total) < amount && k < r) current += s[(size_t) k++];
  }
  if (k < minRank) k = minRank;
  if (k > maxRank) k = maxRank;
  if (k > r) return fail(ctx, "rank above min(bins, frames)");
  *kOut = k;
  U.assign((size_t) std::max<int64_t>(k, 1) * F, 0.0);
  VT.assign((size_t) std::max<int64_t>(k, 1) * T, 0.0);
  for (int64_t j = 0; j < k; j++)
  {
    const int64_t row = order[(size_t) j];
    HIPCHK(ctx, hipMemcpyAsync(&U[(size_t) j * F], dJ.as<double>() + row * F, (size_t) F * sizeof(double),
                               hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(&VT[(size_t) j * T], G + row * ldg, (size_t) T * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(ctx, hipStreamSynchronize(st));
  for (int64_t j = 0; j < k; j++)
  {
    const double sj = s[(size_t) j];
    if (sj > 0)
      for (int64_t t = 0; t < T; t++) VT[(size_t) j * T + t] /= sj;
  }
  return FLUHIP_OK;
}

// alg/NNDSVD.hpp:60-129 from (U, s, V^T).  W: wRows x F row-major, H: T x wRows row-major.
static void nndsvd_construct(const std::vector<double>& s, const std::vector<double>& U, const std::vector<double>& VT,
                             int64_t F, int64_t T, int64_t k, int64_t wRows, int method, int64_t seed, double mean,
                             double* W, double* H)
{
  const double eps = kEpsilon;
  std::fill(W, W + wRows * F, 0.0);
  std::fill(H, H + T * wRows, 0.0);
  auto u = [&](int64_t j) { return &U[(size_t) j * F]; };
  auto v = [&](int64_t j) { return &VT[(size_t) j * T]; };
  if (method == 0)
  {
    for (int64_t j = 0; j < k; j++)
    {
      for (int64_t f = 0; f < F; f++) W[j * F + f] = std::fabs(u(j)[f]);
      for (int64_t t = 0; t < T; t++) H[t * wRows + j] = std::fabs(s[(size_t) j] * v(j)[t]);
    }
    return;
  }
  if (k > 0)
  {
    for (int64_t f = 0; f < F; f++) W[f] = std::fabs(u(0)[f]);                                  // :68
    const double sq = std::sqrt(s[0]);
    for (int64_t t = 0; t < T; t++) H[t * wRows] = sq * std::fabs(v(0)[t]);                       // :69
  }
  for (int64_t j = 1; j < k; j++)
  {
    double xP = 0, yP = 0, xN = 0;
    for (int64_t f = 0; f < F; f++) { const double x = u(j)[f]; if (x > 0) xP += x * x; else xN += x * x; }
    for (int64_t t = 0; t < T; t++) { const double y = v(j)[t]; if (y > 0) yP += y * y; }
    const double xPn = std::sqrt(xP), yPn = std::sqrt(yP), xNn = std::sqrt(xN);
    const double yNn = xNn;                                                                       // :85 as written
    const double mP = xPn * yPn, mN = xNn * yNn;
    const bool pos = mP > mN;
    const double sigma = pos ? mP : mN;
    const double lbd = std::sqrt(s[(size_t) j] * sigma);
    const double xn = pos ? xPn : xNn, yn = pos ? yPn : yNn; // :90-100 (yNn is ||xN||, see above)
    for (int64_t f = 0; f < F; f++)
    {
      const double x = u(j)[f];
      W[j * F + f] = (pos ? std::max(x, 0.0) : std::fabs(std::min(x, 0.0))) / xn;
    }
    for (int64_t t = 0; t < T; t++)
    {
      const double y = v(j)[t];
      H[t * wRows + j] = lbd * ((pos ? std::max(y, 0.0) : std::fabs(std::min(y, 0.0))) / yn);
    }
  }
  if (method == 1)
  {
    // :107-116: the lazily evaluated random matrix is only sampled where the condition holds, in the assignment's
    // column-major traversal (WT is F x wRows, HT is wRows x T); a fresh generator of the same seed for each
    std::random_device rd;
    const double lo = eps, hi = mean * 0.001;
    {
      std::mt19937_64 g{seed >= 0 ? (size_t) seed : (size_t) rd()};
      std::uniform_real_distribution<double> d{lo, hi};
      for (int64_t j = 0; j < wRows; j++)
        for (int64_t f = 0; f < F; f++)
          if (W[j * F + f] < eps) W[j * F + f] = d(g);
    }
    {
      std::mt19937_64 g{seed >= 0 ? (size_t) seed : (size_t) rd()};
      std::uniform_real_distribution<double> d{lo, hi};
      for (int64_t t = 0; t < T; t++)
        for (int64_t j = 0; j < wRows; j++)
          if (H[t * wRows + j] < eps) H[t * wRows + j] = d(g);
    }
  }
  else if (method == 2)
  {
    for (int64_t i = 0; i < wRows * F; i++) if (W[i] < eps) W[i] = mean;
    for (int64_t i = 0; i < T * wRows; i++) if (H[i] < eps) H[i] = mean;
  }
}

int fluhip_nndsvd_f64(fluhip_ctx* ctx, const double* X, int64_t T, int64_t F, int64_t ldx, int64_t w_rows,
                      int64_t min_rank, int64_t max_rank, double amount, int method, int64_t seed, double* W,
                      double* H, int64_t* rank_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!X || !W || !H || T < 1 || F < 1 || ldx < F) return fail(ctx, "bad matrix arguments");
  if (method < 0 || method > 3) return fail(ctx, "method must be 0..3");
  if (!(amount > 0 || min_rank > 0)) return fail(ctx, "coverage or minimum rank must be positive"); // :40 assert
  if (amount > 1) return fail(ctx, "coverage must be <= 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf A, G;
  HIPCHK(ctx, A.alloc((size_t) T * F * sizeof(double), false, st));
  HIPCHK(ctx, G.alloc((size_t) F * T * sizeof(double), false, st));
  HIPCHK(ctx, hipMemcpy2DAsync(A.p, (size_t) F * sizeof(double), X, (size_t) ldx * sizeof(double),
                               (size_t) F * sizeof(double), (size_t) T, hipMemcpyHostToDevice, st));
  launch_transpose(A.as<double>(), F, 0, G.as<double>(), T, 0, (int) T, (int) F, 1, st); // one bin per row
  std::vector<double> s, U, VT;
  int64_t k = 0;
  if (int rc = nndsvd_device(ctx, G.as<double>(), F, T, T, min_rank, max_rank, amount, s, U, VT, &k)) return rc;
  if (k > w_rows) return fail(ctx, "rank exceeds the rows of W");
  double mean = 0;
  for (int64_t t = 0; t < T; t++)
    for (int64_t f = 0; f < F; f++) mean += X[t * ldx + f];
  mean /= (double) (T * F);
  nndsvd_construct(s, U, VT, F, T, k, w_rows, method, seed, mean, W, H);
  if (rank_out) *rank_out = k;
  return FLUHIP_OK;
}

int fluhip_bufnmfseed_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride, int64_t win,
                          int64_t fft, int64_t hop, int64_t min_rank, int64_t max_rank, double coverage,
                          int method, int64_t seed, float* bases_out, float* acts_out, int64_t* rank_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!audio) return fail(ctx, "null audio");
  if (stride < 1) return fail(ctx, "stride must be >= 1");
  if (max_rank < 1) return fail(ctx, "maximum rank must be >= 1");
  int rc = check_shape(ctx, n, win, fft, hop, 1);
  if (rc) return rc;
  if (method < 0 || method > 3) return fail(ctx, "method must be 0..3");
  if (!(coverage > 0 || min_rank > 0)) return fail(ctx, "coverage or minimum rank must be positive");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.n = n; c.win = win; c.fft = fft; c.hop = hop; c.K = 1;
  rc = corpus_alloc(ctx, &c);
  if (rc) return rc;
  DevBuf in;
  HIPCHK(ctx, in.alloc((size_t) n * sizeof(float), false, st));
  HIPCHK(ctx, upload_strided(in.p, audio, (size_t) n, (size_t) stride, sizeof(float), st));
  rc = corpus_stft(&c, in.as<float>(), nullptr, n); // NMFSeedClient.hpp:97-98
  if (rc) return rc;
  const int64_t T = c.T, F = c.F;
  // mean of the magnitudes for methods 1 and 2 (alg/NNDSVD.hpp:105): on the host from a copy of the
  // spectrogram (it is small next to the SVD)
  std::vector<double> mag((size_t) T * F);
  HIPCHK(ctx, hipMemcpy2DAsync(mag.data(), (size_t) F * sizeof(double), c.mag.p, (size_t) c.Fp * sizeof(double),
                               (size_t) F * sizeof(double), (size_t) T, hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  double mean = 0;
  for (double v : mag) mean += v;
  mean /= (double) (T * F);
  std::vector<double> s, U, VT;
  int64_t k = 0;
  // the transposed copy holds one bin per row, as the Jacobi kernels want it; they work in place
  rc = nndsvd_device(ctx, c.magT.as<double>(), F, T, c.Tp, min_rank, max_rank, coverage, s, U, VT, &k);
  if (rc) return rc;
  std::vector<double> W((size_t) max_rank * F), H((size_t) T * max_rank);
  nndsvd_construct(s, U, VT, F, T, k, max_rank, method, seed, mean, W.data(), H.data());
  // NMFSeedClient.hpp:108-128
  if (bases_out)
    for (int64_t i = 0; i < max_rank * F; i++) bases_out[i] = i < k * F ? (float) W[(size_t) i] : 0.f;
  if (acts_out)
  {
    double maxH = H[0];
    for (double v : H) maxH = std::max(maxH, v);
    const float scale = (float) (1.0 / maxH);
    for (int64_t j = 0; j < max_rank; j++)
      for (int64_t t = 0; t < T; t++)
        acts_out[j * T + t] = j < k ? (float) H[(size_t) t * max_rank + j] * scale : 0.f;
  }
  if (rank_out) *rank_out = k;
  return FLUHIP_OK;
}

int fluhip_bufnmf_channel_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride,
                              int64_t win, int64_t fft, int64_t hop, int64_t K, int64_t iters,
                              int update_w, int update_h, int64_t seed, const float* bases_seed,
                              const float* acts_seed, float* bases_out, float* acts_out,
                              float* resynth_out, fluhip_progress_fn progress, void* user)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!audio) return fail(ctx, "null audio");
  if (stride < 1) return fail(ctx, "stride must be >= 1");
  int rc = check_shape(ctx, n, win, fft, hop, K);
  if (rc) return rc;
  if (iters < 0) return fail(ctx, "negative iteration count");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.n = n; c.win = win; c.fft = fft; c.hop = hop; c.K = K;
  c.keepSpec = resynth_out != nullptr; // the complex spectrogram is only needed for resynthesis
  rc = corpus_alloc(ctx, &c);
  if (rc) return rc;
  DevBuf in;
  HIPCHK(ctx, in.alloc((size_t) n * sizeof(float), false, s));
  HIPCHK(ctx, upload_strided(in.p, audio, (size_t) n, (size_t) stride, sizeof(float), s));
  rc = corpus_stft(&c, in.as<float>(), nullptr, n); // nrt/NMFClient.hpp:240-242
  if (rc) return rc;
  FactorInit fi;
  fi.W0f32 = bases_seed; // :246-258 seeds gathered channel by channel
  fi.H0f32 = acts_seed;
  rc = corpus_init_factors(&c, seed, nullptr, fi);
  if (rc) return rc;
  rc = corpus_iterate(&c, iters, update_w != 0, update_h != 0, progress, user); // :268-271
  if (rc) return rc;                                                             // :273-274
  rc = fluhip_corpus_writeback_host(&c, bases_out, acts_out);                    // :277-300
  if (rc) return rc;
  if (resynth_out) // :302-334  estimate -> ratio mask -> ISTFT per component (the corpus form, one buffer)
  {
    DevBuf out32;
    HIPCHK(ctx, out32.alloc((size_t) K * n * sizeof(float), false, s));
    c.haveFactors = true;
    rc = fluhip_corpus_resynth_dev(&c, out32.as<float>());
    if (rc) return rc;
    const size_t nbytes = (size_t) K * n * sizeof(float);
    rc = copy_to_host(ctx, resynth_out, nbytes, out32.p, nbytes, nbytes, 1, s);
    if (rc) return rc;
  }
  return FLUHIP_OK;
}

// ---- BufSTFT ------------------------------------------------------------------
static int64_t bufstft_padding(int64_t win, int64_t hop, int mode)
{
  return mode == 0 ? 0 : (mode == 1 ? win >> 1 : win - hop); // cc/ParameterTypes.hpp:315-323
}

int fluhip_bufstft_forward_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride, int64_t win,
                               int64_t fft, int64_t hop, int padding_mode, float* mag, float* phase,
                               int64_t* hops_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!audio) return fail(ctx, "No input buffer supplied");
  if (!mag && !phase) return fail(ctx, "Neither magnitude nor phase buffer supplied");
  if (padding_mode < 0 || padding_mode > 2) return fail(ctx, "padding mode must be 0, 1 or 2");
  if (stride < 1) return fail(ctx, "stride must be >= 1");
  int rc = check_shape(ctx, n, win, fft, hop, 1);
  if (rc) return rc;
  if (fft / 2 + 1 >= 65536) // nrt/BufSTFTClient.hpp:135-138
    return fail(ctx, "Can produce up to 65536 channels. Split your data up and try again");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int64_t F = fft / 2 + 1, pad = bufstft_padding(win, hop, padding_mode);
  int64_t padded = n + 2 * pad;                                      // :121-124
  if (padding_mode == 2) padded = ((padded + hop - 1) / hop) * hop;   // :125-127
  if (padded < win) return fail(ctx, "not enough frames");
  const int64_t T = 1 + (padded - win) / hop;                         // :129-130
  if (hops_out) *hops_out = T;
  const double *wtab = nullptr, *ttab = nullptr;
  rc = get_window(ctx, win, fft, FLUHIP_WINDOW_HANN, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, fft, &ttab);
  if (rc) return rc;
  DevBuf in, spec, dm, dp;
  HIPCHK(ctx, in.alloc((size_t) n * sizeof(float), false, s));
  HIPCHK(ctx, upload_strided(in.p, audio, (size_t) n, (size_t) stride, sizeof(float), s));
  HIPCHK(ctx, spec.alloc((size_t) T * F * 2 * sizeof(double), false, s));
  if (mag) HIPCHK(ctx, dm.alloc((size_t) T * F * sizeof(float), false, s));
  if (phase) HIPCHK(ctx, dp.alloc((size_t) T * F * sizeof(float), false, s));
  StftArgs sa;
  sa.audio = in.as<float>(); sa.audio64 = nullptr; sa.n = n; sa.audioStride = n;
  sa.win = (int) win; sa.fft = (int) fft; sa.hop = (int) hop; sa.T = (int) T; sa.F = (int) F; sa.B = 1;
  sa.window = wtab; sa.twiddle = ttab; sa.mag = nullptr; sa.magStride = 0; sa.ldMag = 0;
  sa.spec = spec.as<double>(); sa.specStride = 0;
  sa.frameOffset = (int) (win / 2 - pad); // frame i starts at sample i*hop - padding (:151-162)
  sa.bigScratch = big_fft_scratch(ctx, win, fft, T);
  if (stft_needs_scratch(win, fft) && !sa.bigScratch) return FLUHIP_ERROR;
  launch_stft(sa, s);
  launch_spec_to_magphase(spec.as<double>(), (int) T, (int) F, mag ? dm.as<float>() : nullptr,
                          phase ? dp.as<float>() : nullptr, s);
  HIPCHK(ctx, hipGetLastError());
  if (mag) HIPCHK(ctx, hipMemcpyAsync(mag, dm.p, (size_t) T * F * sizeof(float), hipMemcpyDeviceToHost, s));
  if (phase) HIPCHK(ctx, hipMemcpyAsync(phase, dp.p, (size_t) T * F * sizeof(float), hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipStreamSynchronize(s));
  return FLUHIP_OK;
}

int fluhip_bufstft_inverse_f32(fluhip_ctx* ctx, const float* mag, const float* phase, int64_t hops, int64_t win,
                               int64_t fft, int64_t hop, int padding_mode, float* out, int64_t* n_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!mag || !phase) return fail(ctx, "Need both magnutude and phase buffers for inverse transform");
  if (padding_mode < 0 || padding_mode > 2) return fail(ctx, "padding mode must be 0, 1 or 2");
  if (hops < 1) return fail(ctx, "not enough frames");
  int rc = check_shape(ctx, 1, win, fft, hop, 1);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int64_t F = fft / 2 + 1, T = hops, pad = bufstft_padding(win, hop, padding_mode);
  const int64_t paddedOut = (T - 1) * hop + win; // nrt/BufSTFTClient.hpp:233
  const int64_t finalOut = paddedOut - pad;      // :234
  if (n_out) *n_out = finalOut;
  if (!out) return FLUHIP_OK;                    // size query
  const double *wtab = nullptr, *ttab = nullptr;
  rc = get_window(ctx, win, fft, FLUHIP_WINDOW_HANN, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, fft, &ttab);
  if (rc) return rc;
  DevBuf dm, dp, spec, frames, dout;
  HIPCHK(ctx, dm.alloc((size_t) T * F * sizeof(float), false, s));
  HIPCHK(ctx, dp.alloc((size_t) T * F * sizeof(float), false, s));
  HIPCHK(ctx, spec.alloc((size_t) T * F * 2 * sizeof(double), false, s));
  HIPCHK(ctx, frames.alloc((size_t) T * win * sizeof(double), false, s));
  HIPCHK(ctx, dout.alloc((size_t) finalOut * sizeof(float), false, s));
  HIPCHK(ctx, hipMemcpyAsync(dm.p, mag, (size_t) T * F * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(dp.p, phase, (size_t) T * F * sizeof(float), hipMemcpyHostToDevice, s));
  launch_polar_to_spec(dm.as<float>(), dp.as<float>(), (int) T, (int) F, spec.as<double>(), s);
  ResynthArgs ra;
  ra.spec = spec.as<double>(); ra.Wf = nullptr; ra.H1 = nullptr; ra.Vhat = nullptr; ra.ldV = 0; ra.Kp = 0; ra.k = 0;
  ra.win = (int) win; ra.fft = (int) fft; ra.hop = (int) hop; ra.T = (int) T; ra.F = (int) F;
  ra.window = wtab; ra.twiddle = ttab; ra.frames = frames.as<double>(); ra.out = nullptr;
  ra.out32 = dout.as<float>(); ra.n = finalOut; ra.trim = pad;
  ra.bigScratch = big_fft_scratch(ctx, ra.win, ra.fft, ra.T);
  if (stft_needs_scratch(ra.win, ra.fft) && !ra.bigScratch) return FLUHIP_ERROR;
  launch_resynth(ra, s);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout.p, (size_t) finalOut * sizeof(float), hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipStreamSynchronize(s));
  return FLUHIP_OK;
}

// ---- feature pipeline -------------------------------------------------------
static int features_common(fluhip_ctx* ctx, bool mfcc, const float* audio, int64_t count, int64_t n, int64_t win,
                           int64_t fft, int64_t hop, int64_t nBands, int64_t nCoefs, int64_t startCoeff,
                           double minFreq, double maxFreq, double sampleRate, int normalize, int scaleDb,
                           int paddingMode, float* out, int64_t* frames_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!audio || !out) return fail(ctx, "null buffer");
  if (paddingMode < 0 || paddingMode > 2) return fail(ctx, "padding mode must be 0 (None), 1 (Default) or 2 (Full)");
  if (count < 1) return fail(ctx, "need at least one buffer");
  int rc = check_shape(ctx, n, win, fft, hop, 1);
  if (rc) return rc;
  if (nBands < 2 || nBands > fft / 2 + 1) return fail(ctx, "numBands must be in [2, fft/2 + 1]");
  if (!(maxFreq > minFreq)) return fail(ctx, "maxFreq must be above minFreq");
  if (mfcc && (nCoefs < 2 || nCoefs > nBands || startCoeff < 0 || startCoeff > 1))
    return fail(ctx, "numCoeffs must be in [2, numBands] and startCoeff in [0, 1]");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int64_t F = fft / 2 + 1;
  // StreamingControl bookkeeping (cc/FluidNRTClientWrapper.hpp:564-579, 642-644)
  // userPadding.first = FFTParams::padding (cc/ParameterTypes.hpp:315-323): 0 / win/2 / win - hop; the input sits that
  // far into the padded signal, the client's latency (= win) is added in front of the analysis, the padded length is
  // rounded up to whole hops in Full mode (:572-574), and the first latency / hop output frames are dropped (:643-656)
  const int64_t latencyHops = win / hop;
  const int64_t userPad = paddingMode == 0 ? 0 : paddingMode == 1 ? win / 2 : win - hop;
  int64_t paddedLength = n + win + 2 * userPad;
  if (paddingMode == 2) paddedLength = ((paddedLength + hop - 1) / hop) * hop;
  const int64_t T = 1 + (paddedLength - win) / hop - latencyHops;
  // kept frame k starts at sample latencyHops hop - win - userPad + k hop; the kernels place frame t at
  // t hop - win/2 + frameOffset
  const int64_t frameOffset = latencyHops * hop - win + win / 2 - userPad;
  if (T < 1) return fail(ctx, "not enough frames");
  if (frames_out) *frames_out = T;
  const int64_t Tp = round_up(T, 32), Fp = round_up(F, 32);
  const int64_t bandsPad = round_up(nBands, 64);
  // mel filter bank (alg/MelBands.hpp:53-73), bin-major and zero padded; f64 on the host like the reference
  std::vector<double> filtT((size_t) F * bandsPad, 0.0);
  {
    auto hz2mel = [](double x) { return 1127.01048 * std::log(x / 700.0 + 1.0); };
    const int64_t nc = nBands + 2;
    std::vector<double> centres((size_t) nc);
    const double mlo = hz2mel(minFreq), mhi = hz2mel(maxFreq);
    for (int64_t i = 0; i < nc; i++)
      centres[(size_t) i] = 700.0 * (std::exp((mlo + (double) i * (mhi - mlo) / (double) (nc - 1)) / 1127.01048) - 1.0);
    for (int64_t b = 0; b < nBands; b++)
    {
      const double d0 = std::fabs(centres[(size_t) b] - centres[(size_t) b + 1]);
      const double d1 = std::fabs(centres[(size_t) b + 1] - centres[(size_t) b + 2]);
      for (int64_t f = 0; f < F; f++)
      {
        const double hz = (double) f * (sampleRate / 2.0) / (double) (F - 1);
        const double lower = -(centres[(size_t) b] - hz) / d0, upper = (centres[(size_t) b + 2] - hz) / d1;
        filtT[(size_t) (f * bandsPad + b)] = std::max(0.0, std::min(lower, upper));
      }
    }
  }
  // the filter bank over each band's support only (kernels_feat.hip): first non-zero bin and packed weights
  std::vector<int> bandLo((size_t) bandsPad, 0);
  int64_t maxLen = 1;
  {
    std::vector<int64_t> hi((size_t) bandsPad, -1);
    for (int64_t b = 0; b < nBands; b++)
    {
      int64_t lo = -1;
      for (int64_t f = 0; f < F; f++)
        if (filtT[(size_t) (f * bandsPad + b)] != 0.0) { if (lo < 0) lo = f; hi[(size_t) b] = f; }
      bandLo[(size_t) b] = (int) std::max<int64_t>(lo, 0);
      if (lo >= 0) maxLen = std::max(maxLen, hi[(size_t) b] - lo + 1);
    }
  }
  std::vector<double> wpack((size_t) maxLen * bandsPad, 0.0);
  for (int64_t b = 0; b < nBands; b++)
    for (int64_t j = 0; j < maxLen; j++)
    {
      const int64_t f = bandLo[(size_t) b] + j;
      if (f < F) wpack[(size_t) (j * bandsPad + b)] = filtT[(size_t) (f * bandsPad + b)];
    }
  const int64_t nDct = mfcc ? std::min(nCoefs + startCoeff, nBands) : 0; // rt/MFCCClient.hpp:104-105
  std::vector<double> dct((size_t) std::max<int64_t>(1, nDct * nBands));
  for (int64_t i = 0; i < nDct; i++) // alg/DCT.hpp:53-61
  {
    const double scale = i == 0 ? 1.0 / std::sqrt((double) nBands) : std::sqrt(2.0 / (double) nBands);
    for (int64_t j = 0; j < nBands; j++)
      dct[(size_t) (i * nBands + j)] = std::cos((M_PI / (double) nBands) * (double) i * (0.5 + (double) j)) * scale;
  }
  const int64_t nOut = mfcc ? nCoefs : nBands;
  const double *wtab = nullptr, *ttab = nullptr;
  rc = get_window(ctx, win, fft, FLUHIP_WINDOW_HANN, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, fft, &ttab);
  if (rc) return rc;
  DevBuf dFilt, dDct, dAudio, dMag, dOut, dLo, dPack;
  HIPCHK(ctx, dLo.alloc(bandLo.size() * sizeof(int), false, s));
  HIPCHK(ctx, dPack.alloc(wpack.size() * sizeof(double), false, s));
  HIPCHK(ctx, hipMemcpyAsync(dLo.p, bandLo.data(), bandLo.size() * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(dPack.p, wpack.data(), wpack.size() * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, dFilt.alloc(filtT.size() * sizeof(double), false, s));
  HIPCHK(ctx, dDct.alloc(dct.size() * sizeof(double), false, s));
  HIPCHK(ctx, hipMemcpyAsync(dFilt.p, filtT.data(), filtT.size() * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(dDct.p, dct.data(), dct.size() * sizeof(double), hipMemcpyHostToDevice, s));
  // ---- fused form (kernels_stft2.hip stft_feat_kernel): the magnitudes never leave the chip --------------------------
  // Every bin must lie on the rising edge of at most one band and the falling edge of the band below it, with the
  // bins of each edge contiguous: then band b = (sum of up[f] m[f] over interval b) + (sum of dn[f] m[f] over interval
  // b + 1), interval s = the bins between centres s and s + 1.  True of any filter bank whose triangles are wider than
  // a bin; checked here against the dense matrix, coefficient by coefficient, and anything else takes the two-kernel path.
  {
    const int CH = stft_features_bins_per_lane((int) fft);
    std::vector<double> up((size_t) 64 * CH, 0.0), dn((size_t) 64 * CH, 0.0);
    std::vector<short> slot((size_t) 64 * CH, (short) -1);
    std::vector<int64_t> interval((size_t) F, -1); // interval of bin f, -1: no band touches it
    bool ok = nBands <= 64 && (!mfcc || nDct * nBands <= 4096) && !stft_needs_scratch(win, fft) && (fft == 1024 || fft == 2048) &&
              (win % 2) == 0;
    if (const char* e = fluhip::ab_getenv("FLUHIP_FEAT_FUSED")) // A/B and tests: 0 forces the two-kernel form
      if (std::atoi(e) == 0) ok = false;
    std::vector<int64_t> peak((size_t) nBands, 0);
    for (int64_t b = 0; ok && b < nBands; b++)
    {
      double best = -1.0;
      for (int64_t f = 0; f < F; f++)
        if (filtT[(size_t) (f * bandsPad + b)] > best) { best = filtT[(size_t) (f * bandsPad + b)]; peak[(size_t) b] = f; }
      if (best <= 0.0) ok = false; // a band no bin falls into
    }
    for (int64_t f = 0; ok && f < F; f++)
    {
      int64_t b1 = -1, b2 = -1, cnt = 0;
      for (int64_t b = 0; b < nBands; b++)
        if (filtT[(size_t) (f * bandsPad + b)] != 0.0) { if (cnt == 0) b1 = b; else b2 = b; cnt++; }
      if (cnt == 0) continue;
      if (cnt > 2 || (cnt == 2 && b2 != b1 + 1)) { ok = false; break; }
      if (cnt == 2)
      {
        interval[(size_t) f] = b2;
        up[(size_t) f] = filtT[(size_t) (f * bandsPad + b2)];
        dn[(size_t) f] = filtT[(size_t) (f * bandsPad + b1)];
      }
      else if (f <= peak[(size_t) b1]) { interval[(size_t) f] = b1; up[(size_t) f] = filtT[(size_t) (f * bandsPad + b1)]; }
      else { interval[(size_t) f] = b1 + 1; dn[(size_t) f] = filtT[(size_t) (f * bandsPad + b1)]; }
    }
    // interval s starts at bin g[s].  The touched bins must be one contiguous run whose intervals ascend one at a time
    // from some i0 up to nBands (the falling edge of the last band); intervals below i0 are empty and the running
    // sums are still 0 at their boundaries, which therefore publish nothing.
    std::vector<int64_t> g((size_t) nBands + 2, -1);
    if (ok)
    {
      int64_t prev = -1, first = -1, last = -1;
      bool ended = false;
      for (int64_t f = 0; f < F && ok; f++)
      {
        const int64_t iv = interval[(size_t) f];
        if (iv < 0) { if (first >= 0) ended = true; continue; }
        if (ended) { ok = false; break; }              // touched bins are not one contiguous run
        if (first < 0) first = f;
        else if (iv != prev && iv != prev + 1) { ok = false; break; }
        if (iv != prev) g[(size_t) iv] = f;
        prev = iv;
        last = f;
      }
      if (first < 0 || prev != nBands) ok = false;
      if (ok)
      {
        int64_t i0 = 0;
        while (g[(size_t) i0] < 0) i0++;
        for (int64_t sI = 0; sI < i0; sI++) g[(size_t) sI] = first;
        g[(size_t) nBands + 1] = last + 1;
        for (int64_t sI = i0 + 1; sI <= nBands + 1; sI++) slot[(size_t) (g[(size_t) sI] - 1)] = (short) sI;
      }
      // reconstruction: the segment sums must give back the dense matrix exactly
      for (int64_t b = 0; ok && b < nBands; b++)
        for (int64_t f = 0; f < F; f++)
        {
          double w = 0.0;
          if (f >= g[(size_t) b] && f < g[(size_t) b + 1]) w += up[(size_t) f];
          if (f >= g[(size_t) b + 1] && f < g[(size_t) b + 2]) w += dn[(size_t) f];
          if (w != filtT[(size_t) (f * bandsPad + b)]) { ok = false; break; }
        }
    }
    if (ok)
    {
      DevBuf dUp, dDn, dSlot, dDct2, dAud, dOutF;
      HIPCHK(ctx, dUp.alloc(up.size() * sizeof(double), false, s));
      HIPCHK(ctx, dDn.alloc(dn.size() * sizeof(double), false, s));
      HIPCHK(ctx, dSlot.alloc(slot.size() * sizeof(short), false, s));
      HIPCHK(ctx, dDct2.alloc(dct.size() * sizeof(double), false, s));
      HIPCHK(ctx, hipMemcpyAsync(dUp.p, up.data(), up.size() * sizeof(double), hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(dDn.p, dn.data(), dn.size() * sizeof(double), hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(dSlot.p, slot.data(), slot.size() * sizeof(short), hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(dDct2.p, dct.data(), dct.size() * sizeof(double), hipMemcpyHostToDevice, s));
      // device-resident audio / output are used in place; host buffers go through staging chunks of bounded size
      hipPointerAttribute_t pa;
      const bool audDev = hipPointerGetAttributes(&pa, audio) == hipSuccess && pa.type == hipMemoryTypeDevice;
      const bool outDev = hipPointerGetAttributes(&pa, out) == hipSuccess && pa.type == hipMemoryTypeDevice;
      (void) hipGetLastError();
      int64_t chunkBytes = (int64_t) 1 << 31;
      if (const char* e = fluhip::ab_getenv("FLUHIP_FEAT_CHUNK_BYTES")) chunkBytes = std::max<int64_t>(1, std::atoll(e)); // tests: force several chunks
      const int64_t chunkB = (audDev && outDev) ? count
                                                : std::max<int64_t>(1, std::min<int64_t>(count, chunkBytes / (n * (int64_t) sizeof(float))));
      if (!audDev) HIPCHK(ctx, dAud.alloc((size_t) chunkB * n * sizeof(float), false, s));
      if (!outDev) HIPCHK(ctx, dOutF.alloc((size_t) chunkB * nOut * T * sizeof(float), false, s));
      for (int64_t b0 = 0; b0 < count; b0 += chunkB)
      {
        const int64_t nb = std::min(chunkB, count - b0);
        const float* aPtr = audio + b0 * n;
        if (!audDev)
        {
          HIPCHK(ctx, hipMemcpyAsync(dAud.p, aPtr, (size_t) nb * n * sizeof(float), hipMemcpyDefault, s));
          aPtr = dAud.as<float>();
        }
        float* oPtr = outDev ? out + b0 * nOut * T : dOutF.as<float>();
        StftArgs sa;
        sa.audio = aPtr; sa.audio64 = nullptr; sa.n = n; sa.audioStride = n;
        sa.win = (int) win; sa.fft = (int) fft; sa.hop = (int) hop; sa.T = (int) T; sa.F = (int) F; sa.B = (int) nb;
        sa.window = wtab; sa.twiddle = ttab;
        sa.mag = nullptr; sa.magStride = 0; sa.ldMag = 0; sa.spec = nullptr; sa.specStride = 0;
        sa.frameOffset = (int) frameOffset; sa.bigScratch = nullptr;
        FeatArgs fa;
        fa.mag = nullptr; fa.magStride = 0; fa.ldMag = 0;
        fa.T = (int) T; fa.F = (int) F; fa.B = (int) nb; fa.win = (int) win;
        fa.filtT = nullptr; fa.nBands = (int) nBands; fa.bandsPad = (int) bandsPad;
        fa.bandLo = nullptr; fa.wpack = nullptr; fa.maxLen = 0;
        fa.magNorm = mfcc ? 0 : (normalize ? 1 : 0); fa.usePower = 0; fa.logOutput = mfcc ? 1 : (scaleDb ? 1 : 0);
        fa.dct = mfcc ? dDct2.as<double>() : nullptr; fa.nDct = (int) nDct; fa.startCoeff = (int) startCoeff;
        fa.nOut = (int) nOut; fa.out = oPtr;
        bool launched;
        {
          ProfScope p(ctx, 2);
          launched = launch_stft_features(sa, fa, dUp.as<double>(), dDn.as<double>(), dSlot.as<short>(), s);
        }
        if (!launched) { ok = false; break; }
        HIPCHK(ctx, hipGetLastError());
        if (!outDev)
          HIPCHK(ctx, hipMemcpyAsync(out + b0 * nOut * T, oPtr, (size_t) nb * nOut * T * sizeof(float), hipMemcpyDefault, s));
        if (!audDev || !outDev) HIPCHK(ctx, hipStreamSynchronize(s));   // the staging buffers are reused by the next chunk
      }
      if (ok)
      {
        HIPCHK(ctx, hipStreamSynchronize(s));
        return FLUHIP_OK;
      }
    }
  }
  // ---- two-kernel form: magnitudes through HBM, any filter bank / fft size -------------------------------------------
  // buffers are processed in chunks that keep the magnitude scratch around 2 GiB
  const int64_t perBuf = Tp * Fp * (int64_t) sizeof(double);
  int64_t scratchBytes = 2LL << 30;
  if (const char* e = fluhip::ab_getenv("FLUHIP_FEAT_CHUNK_BYTES")) scratchBytes = std::max<int64_t>(1, std::atoll(e)); // tests: force several chunks
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(count, 65535), scratchBytes / perBuf));
  HIPCHK(ctx, dAudio.alloc((size_t) chunk * n * sizeof(float), false, s));
  HIPCHK(ctx, dMag.alloc((size_t) chunk * perBuf, true, s));
  HIPCHK(ctx, dOut.alloc((size_t) chunk * nOut * T * sizeof(float), false, s));
  for (int64_t b0 = 0; b0 < count; b0 += chunk)
  {
    const int64_t nb = std::min(chunk, count - b0);
    // hipMemcpyDefault: `audio` and `out` may be host or device pointers (a corpus already resident in HBM skips PCIe)
    HIPCHK(ctx, hipMemcpyAsync(dAudio.p, audio + b0 * n, (size_t) nb * n * sizeof(float), hipMemcpyDefault, s));
    StftArgs sa;
    sa.audio = dAudio.as<float>(); sa.audio64 = nullptr; sa.n = n; sa.audioStride = n;
    sa.win = (int) win; sa.fft = (int) fft; sa.hop = (int) hop; sa.T = (int) T; sa.F = (int) F; sa.B = (int) nb;
    sa.window = wtab; sa.twiddle = ttab;
    sa.mag = dMag.as<double>(); sa.magStride = Tp * Fp; sa.ldMag = Fp;
    sa.spec = nullptr; sa.specStride = 0; sa.frameOffset = (int) frameOffset;
    sa.bigScratch = big_fft_scratch(ctx, win, fft, nb * T);
    if (stft_needs_scratch(win, fft) && !sa.bigScratch) return FLUHIP_ERROR;
    {
      ProfScope p(ctx, 0);
      launch_stft(sa, s);
    }
    FeatArgs fa;
    fa.mag = dMag.as<double>(); fa.magStride = Tp * Fp; fa.ldMag = Fp;
    fa.T = (int) T; fa.F = (int) F; fa.B = (int) nb; fa.win = (int) win;
    fa.filtT = dFilt.as<double>(); fa.nBands = (int) nBands; fa.bandsPad = (int) bandsPad;
    fa.bandLo = dLo.as<int>(); fa.wpack = dPack.as<double>(); fa.maxLen = (int) maxLen;
    // rt/MFCCClient.hpp:123-124 (false, false, true); rt/MelBandsClient.hpp:106-108 (normalize, false, scale == dB)
    fa.magNorm = mfcc ? 0 : (normalize ? 1 : 0); fa.usePower = 0; fa.logOutput = mfcc ? 1 : (scaleDb ? 1 : 0);
    fa.dct = mfcc ? dDct.as<double>() : nullptr; fa.nDct = (int) nDct; fa.startCoeff = (int) startCoeff;
    fa.nOut = (int) nOut; fa.out = dOut.as<float>();
    {
      ProfScope p(ctx, 2);
      launch_features(fa, s);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out + b0 * nOut * T, dOut.p, (size_t) nb * nOut * T * sizeof(float),
                               hipMemcpyDefault, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
  }
  return FLUHIP_OK;
}

int fluhip_bufmelbands_padded_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win,
                                  int64_t fft, int64_t hop, int64_t n_bands, double min_freq, double max_freq,
                                  double sample_rate, int normalize, int scale_db, int padding_mode, float* out,
                                  int64_t* frames_out)
{
  return features_common(ctx, false, audio, count, n, win, fft, hop, n_bands, 0, 0, min_freq, max_freq,
                         sample_rate, normalize, scale_db, padding_mode, out, frames_out);
}
int fluhip_bufmelbands_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win,
                           int64_t fft, int64_t hop, int64_t n_bands, double min_freq, double max_freq,
                           double sample_rate, int normalize, int scale_db, float* out, int64_t* frames_out)
{
  return fluhip_bufmelbands_padded_f32(ctx, audio, count, n, win, fft, hop, n_bands, min_freq, max_freq, sample_rate,
                                       normalize, scale_db, 1, out, frames_out);
}

int fluhip_bufmfcc_padded_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                              int64_t hop, int64_t n_bands, int64_t n_coefs, int64_t start_coeff, double min_freq,
                              double max_freq, double sample_rate, int padding_mode, float* out, int64_t* frames_out)
{
  return features_common(ctx, true, audio, count, n, win, fft, hop, n_bands, n_coefs, start_coeff, min_freq,
                         max_freq, sample_rate, 0, 0, padding_mode, out, frames_out);
}
int fluhip_bufmfcc_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                       int64_t hop, int64_t n_bands, int64_t n_coefs, int64_t start_coeff, double min_freq,
                       double max_freq, double sample_rate, float* out, int64_t* frames_out)
{
  return fluhip_bufmfcc_padded_f32(ctx, audio, count, n, win, fft, hop, n_bands, n_coefs, start_coeff, min_freq,
                                   max_freq, sample_rate, 1, out, frames_out);
}

int64_t fluhip_debug_plan_lists(int64_t count, const int64_t* frames, int64_t bins, int64_t K, int which, int32_t* desc,
                                int64_t cap, int32_t* info8)
{
  if (!frames || count < 1 || bins < 1 || K < 1 || K > 128) return -1;
  std::vector<int> tOf((size_t) count);
  int tmax = 1;
  for (int64_t i = 0; i < count; i++)
  {
    if (frames[i] < 1) return -1;
    tOf[(size_t) i] = (int) frames[i];
    tmax = std::max(tmax, tOf[(size_t) i]);
  }
  ListPlanHost plan;
  build_list_plan(tOf, tmax, (int) bins, (int) padded_rank(K), plan);
  const ListSide& sd = which ? plan.H : plan.W;
  if (info8)
  {
    info8[0] = sd.wgs; info8[1] = sd.ng; info8[2] = sd.partial; info8[3] = sd.maxSplit; info8[4] = sd.pieces;
    info8[5] = (int32_t) sd.nPartials; info8[6] = plan.sideW ? 1 : 0; info8[7] = sd.statParts;
  }
  const int64_t n = (int64_t) sd.list.size();
  static_assert(sizeof(WaveDesc) == 12 * sizeof(int32_t), "descriptor layout");
  if (desc) std::memcpy(desc, sd.list.data(), (size_t) std::min(n, cap) * sizeof(WaveDesc));
  return n;
}

int fluhip_debug_plan_kind(int64_t count, int64_t frames, int64_t bins, int64_t K)
{
  if (count < 1 || frames < 1 || bins < 1 || K < 1) return -1;
  const int Kp = (int) padded_rank(K);
  if (update_variant(Kp) != 5) return 0;
  const PlanShape ps{count, frames, bins, Kp};
  return list_plan_pays(&ps) ? 1 : 0;
}

int fluhip_debug_plan_tail(int64_t count, int64_t frames, int64_t bins, int64_t K, int64_t* out4)
{
  if (count < 1 || frames < 1 || bins < 1 || K < 1 || !out4) return FLUHIP_ERROR;
  const int Kp = (int) padded_rank(K);
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  if (update_variant(Kp) != 5) return FLUHIP_OK;
  int wA = 0;
  const int sp = plan_tail(count, (int) frames, (int) bins, Kp, &wA);
  if (sp > 1)
  {
    const int G = ((int) frames + 15) / 16, w = nmf_update5_strips((int) frames, Kp, (int) count);
    out4[0] = sp; out4[1] = wA; out4[2] = w - wA; out4[3] = (int64_t) wA * ((G + w - 1) / w) * 16;
  }
  return FLUHIP_OK;
}

int fluhip_corpus_plan(const fluhip_corpus* c, int64_t* out8)
{
  if (!c || !out8) return FLUHIP_ERROR;
  out8[0] = update_variant((int) c->Kp);
  out8[1] = c->nsplitW;
  out8[2] = c->nsplitH | ((int64_t) c->tailSplitH << 16); // (tail split of the two-launch H update in the high half)
  out8[3] = c->lazy ? 1 : 0;
  out8[4] = c->sideW ? 1 : 0;
  out8[5] = c->stripsW;
  out8[6] = c->Kp;
  out8[7] = c->strip ? 1 : 0;
  return FLUHIP_OK;
}

// ---- profiling ------------------------------------------------------------------------
// debugging aid for FLUHIP_K5_INSTR: first 8 words of the split-denominator scratch
int fluhip_corpus_debug_words(fluhip_corpus* c, int64_t* out32)
{
  if (!c || !out32 || !c->dpart.p) return FLUHIP_ERROR;
  HIPCHK(c->ctx, hipStreamSynchronize(c->ctx->stream));
  if (c->strip)
  {
    // FLUHIP_STRIP_INSTR: the 32 words behind the partials (kernels_nmf_strip.hip STRIP_STAMP)
    std::memset(out32, 0, 32 * sizeof(int64_t));
    const int64_t off = nmf_strip_part_doubles((int) c->F, (int) c->T, (int) c->B) - 32;
    HIPCHK(c->ctx, hipMemcpy(out32, c->stripPart.as<double>() + off, 32 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return FLUHIP_OK;
  }
  HIPCHK(c->ctx, hipMemcpy(out32, c->dpart.p, 32 * sizeof(int64_t), hipMemcpyDeviceToHost));
  return FLUHIP_OK;
}

int fluhip_corpus_update_clocks(fluhip_corpus* c, int64_t* out8, int reset)
{
  if (!c || !c->clk.p) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (out8) HIPCHK(ctx, hipMemcpy(out8, c->clk.p, 8 * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (reset) HIPCHK(ctx, hipMemset(c->clk.p, 0, 8 * sizeof(int64_t)));
  return FLUHIP_OK;
}

int fluhip_prof_enable(fluhip_ctx* ctx, int on)
{
  if (!ctx) return FLUHIP_ERROR;
  ctx->prof = on != 0;
  return FLUHIP_OK;
}

int fluhip_prof_reset(fluhip_ctx* ctx)
{
  if (!ctx) return FLUHIP_ERROR;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& r : ctx->profRecs)
  {
    ctx->eventPool.push_back(r.start);
    ctx->eventPool.push_back(r.stop);
  }
  ctx->profRecs.clear();
  return FLUHIP_OK;
}

int fluhip_prof_read(fluhip_ctx* ctx, int kernel_class, int64_t* launches, double* total_ms)
{
  if (!ctx) return FLUHIP_ERROR;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int64_t n = 0;
  double tot = 0.0;
  for (auto& r : ctx->profRecs)
  {
    if (r.cls != kernel_class) continue;
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, r.start, r.stop));
    tot += ms;
    n++;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = tot;
  return FLUHIP_OK;
}

} // extern "C"
