// api_internal.h -- what the translation units of the C-ABI layer share (api_core / api_corpus / api_algorithms /
// api_features / api_frames .hip): the context and corpus structures, the caching device buffer, and the helpers one
// unit defines for the others.  Not installed; nothing here is part of the ABI (hidden visibility).
#pragma once

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

#pragma GCC visibility push(hidden)

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
  hipStream_t sideStream = nullptr; // the W update's side column runs beside the update launch (created on first use)
  hipEvent_t sideEv[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // fork / join pairs, in rotation
  unsigned sideTurn = 0;
  std::string err;
  bool errOom = false; // the last failure was an allocation the device (or the host) could not serve (fluhip_last_error_is_out_of_memory)
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

int fail(fluhip_ctx* ctx, const std::string& msg, int status = FLUHIP_ERROR);
int fail_oom(fluhip_ctx* ctx, const std::string& msg); // fail() + the out-of-memory classification (by code, not by text)
int fail_hip(fluhip_ctx* ctx, hipError_t e, const char* what);

#define HIPCHK(ctx, expr)                                                                        \
  do                                                                                             \
  {                                                                                              \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess) return fail_hip(ctx, e__, #expr);                                     \
  } while (0)

// api_core.hip
double* big_fft_scratch(fluhip_ctx* ctx, int64_t win, int64_t fft, int64_t frames);
int copy_to_host(fluhip_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, hipStream_t s);
hipEvent_t take_event(fluhip_ctx* ctx);
hipError_t upload_strided(void* dst, const void* src, size_t n, size_t stride, size_t esz, hipStream_t s);
bool make_window(int type, int64_t size, std::vector<double>& out);
int get_window(fluhip_ctx* ctx, int64_t win, int64_t fft, int type, const double** out);
int get_twiddle(fluhip_ctx* ctx, int64_t fft, const double** out);

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
extern BlockPool g_pool;

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

// ---------------------------------------------------------------------------------------
// corpus
// ---------------------------------------------------------------------------------------
// the schedule of the factor updates of a shape as data: api_corpus.hip decide_update_plan (round 6)
struct UpdatePlan
{
  int variant = 5;      // 5: kernels_nmf5.hip (ranks up to 128), 0: the any-rank path
  int64_t Kp = 0;       // rank the arrays are laid out for
  int Kc = 0;           // rank the factor updates compute (off-size ranks)
  int nsplitW = 1, nsplitH = 1;
  bool lazy = false, sideW = false, strip = false, stripBin = false, stripTile = false, useLists = false;
  int stripsW = 0;      // statistics records per buffer of a W update (its wavefronts, or the finalize's chunks when split)
  int stripsH = 0;      // wavefronts per buffer of a uniform H update (0: lists / frame strips / any-rank)
  int tailSplitH = 0, tailStripsH = 0, tailRestH = 0, tailColsH = 0;
  // workspaces in doubles: what the launches of this plan index (plan_updates allocates exactly these)
  int64_t partDoubles = 0, dpartDoubles = 0, csumDoubles = 0, wscratchDoubles = 0, colPartDoubles = 0, stripPartDoubles = 0,
          wideDoubles = 0;
};
void decide_update_plan(int64_t B, int64_t T, int64_t F, int64_t K, UpdatePlan& p);

struct fluhip_corpus
{
  fluhip_ctx* ctx = nullptr;
  int64_t B = 0, n = 0, win = 0, fft = 0, hop = 0, K = 0;
  int64_t T = 0, F = 0, Tp = 0, Fp = 0, Kp = 0;
  int windowType = FLUHIP_WINDOW_HANN;
  bool keepSpec = false;
  bool stftOnly = false; // spectrogram-only use (fluhip_stft_*): one layout of the magnitudes, no factor workspaces
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
  bool sideFromH = false; // the H update enqueued last left the next W update's side-column partials (sideFromHSlices per buffer)
  bool colsumWInPlace = false; // the W update enqueued last left the column sums of the new W' in the H update's denominator slots (launch_wnorm_combine)
  int sideFromHSlices = 0;
  int sideGen = 0;        // which of the two side-partial areas of wscratch holds them (an H update reads one and fills the other)
  bool normDue = false;   // the W update enqueued last left its norm combine to the H update behind it
  bool planError = false; // a launch did not take the form its dry run announced (reported by corpus_iterate)
  int stripsW = 0;       // wavefronts per buffer of the W update (statistics partials)
  DevBuf wnorm, wscratch, csumScratch, wideScratch;
  DevBuf colPart;            // [B][stripsW][Kp]: column sums of the rows of W' each wavefront of the last W update wrote (UpdateArgs::colOut)
  bool colPartValid = false; // ... and whether that launch filled it (the H update behind then takes its column sums from there)
  DevBuf clk; // UpdateArgs::clk: 4 words for the W update's launches, 4 for the H update's
  // events on the context stream around the last iteration loop (behind the host-side initialisation, in front of
  // nothing but the loop's own launches): fluhip_corpus_last_loop_ms -- device time of the loop for tools/perf_matrix.py
  hipEvent_t loopEv0 = nullptr, loopEv1 = nullptr;
  bool loopTimed = false;
  // frame-strip schedule of a single large buffer at rank <= 16 (kernels_nmf_strip.hip)
  bool strip = false;
  bool stripReady = false;     // the numerator partials of the next W update are in stripPart
  bool stripNormFresh = false; // wnorm holds the column norms of the W' in memory
  bool stripStatsValid = false; // the column-statistics records of generation stripGen describe the W in memory
  int stripGen = 0;
  DevBuf stripPart;
  // ... with the W update as its own launch over bin strips (kernels_nmf_strip.hip nmf_binstrip_kernel; round 4): no
  // numerator partials of the whole matrix, no reduce launch, the strip kernel runs its H phase only
  bool stripBin = false;
  DevBuf binWork;
  int Kc = 0;                  // compute rank of the factor updates (off-size ranks: 24 on arrays of rank 32; 40 / 48 / 56 on 64; 72 .. 112 in eights on 128), else Kp
  bool stripTile = false;      // round 5: the W update as the bin-tiled launch (kernels_nmf_bintile.hip)
  bool stripSideReady = false; // the Nyquist bin's numerator partials of the next W update are in tileWork
  DevBuf tileWork;
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

struct FactorInit
{
  // device sources already in the padded layout are marked by null here
  const double* W0host = nullptr; // [B or 1][K][F] f64
  const double* H0host = nullptr; // [B or 1][T][K] f64
  const float* W0f32 = nullptr;   // [B][K][F] f32 channel-major seeds
  const float* H0f32 = nullptr;   // [B][K][T] f32 channel-major seeds
  bool sharedW = false, sharedH = false;
};

// api_corpus.hip
int update_variant(int Kp);
int64_t padded_rank(int64_t K);
int plan_updates(fluhip_ctx* ctx, fluhip_corpus* c);
int corpus_alloc(fluhip_ctx* ctx, fluhip_corpus* c);
int check_rank(fluhip_ctx* ctx, int64_t T, int64_t F, int64_t K);
int check_shape(fluhip_ctx* ctx, int64_t n, int64_t win, int64_t fft, int64_t hop, int64_t K);
int corpus_stft(fluhip_corpus* c, const float* a32, const double* a64, int64_t audioStride, bool magOnly = false);
void draw_uniform(int64_t seed, size_t count, std::vector<double>& out);
int corpus_init_factors(fluhip_corpus* c, int64_t seed, const int64_t* seeds, const FactorInit& fi);
int corpus_iterate(fluhip_corpus* c, int64_t iters, bool updateW, bool updateH, fluhip_progress_fn progress, void* user);

// api_algorithms.hip
int process_frames_on_device(fluhip_ctx* ctx, fluhip_corpus& c, const double* W0host, int64_t iters, int64_t seed);

#pragma GCC visibility pop
