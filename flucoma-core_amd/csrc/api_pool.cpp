// api_pool.cpp -- several devices behind the C ABI (include/flucoma_hip.h "device pool").
//
// The reference's deployment unit is one std::thread per job (include/flucoma/clients/common/
// FluidNRTClientWrapper.hpp:1042-1048) and a corpus of independent buffers has no cross-buffer state
// (clients/nrt/NMFClient.hpp:233 loop body): the pool keeps one context and, per call, one host thread per
// device, deals the buffers in contiguous blocks (the same arithmetic as flucoma-core_amd/sharding.py) and lets
// every device write its share of the result straight into the caller's arrays -- no collective is needed when
// one host process owns all devices ("D2H per GPU is equally valid").  Built on the public corpus entry
// points only, so it is also a usage example of them.  Plain C++17, no HIP in this file.
#include "../../include/flucoma_hip.h"
#include "fluhip_env.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

struct fluhip_pool
{
  std::vector<fluhip_ctx*> ctx;
  std::vector<int> device;
  std::string err;
};

extern "C" {

void fluhip_shard_range(int64_t n_items, int world, int rank, int64_t* begin, int64_t* end)
{
  if (world < 1) world = 1;
  const int64_t base = n_items / world, rem = n_items % world;
  const int64_t b = rank * base + std::min<int64_t>(rank, rem);
  if (begin) *begin = b;
  if (end) *end = b + base + (rank < rem ? 1 : 0);
}

int fluhip_balanced_assignment(const double* costs, int64_t n, int world, int32_t* rank_of_item)
{
  if (!costs || !rank_of_item || n < 0 || world < 1) return FLUHIP_ERROR;
  std::vector<int64_t> order((size_t) n);
  for (int64_t i = 0; i < n; i++) order[(size_t) i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return costs[a] > costs[b]; }); // ties: lower index first
  std::vector<double> load((size_t) world, 0.0);
  for (int64_t i : order)
  {
    int best = 0;
    for (int r = 1; r < world; r++)
      if (load[(size_t) r] < load[(size_t) best]) best = r; // ties: lower rank
    rank_of_item[i] = best;
    load[(size_t) best] += costs[i];
  }
  return FLUHIP_OK;
}

int fluhip_pool_create(const int* devices, int n_devices, fluhip_pool** out)
{
  if (!out) return FLUHIP_ERROR;
  *out = nullptr;
  std::vector<int> dev;
  if (devices && n_devices > 0) dev.assign(devices, devices + n_devices);
  else
    for (int d = 0; d < fluhip_device_count(); d++) dev.push_back(d);
  if (dev.empty()) return FLUHIP_ERROR;
  fluhip_pool* p = new fluhip_pool;
  for (int d : dev)
  {
    fluhip_ctx* c = nullptr;
    if (fluhip_ctx_create(d, &c) != FLUHIP_OK)
    {
      for (fluhip_ctx* q : p->ctx) fluhip_ctx_destroy(q);
      delete p;
      return FLUHIP_ERROR;
    }
    p->ctx.push_back(c);
    p->device.push_back(d);
  }
  *out = p;
  return FLUHIP_OK;
}

void fluhip_pool_destroy(fluhip_pool* p)
{
  if (!p) return;
  for (fluhip_ctx* c : p->ctx) fluhip_ctx_destroy(c);
  delete p;
}

int fluhip_pool_size(const fluhip_pool* p) { return p ? (int) p->ctx.size() : 0; }
int fluhip_pool_device(const fluhip_pool* p, int member) { return (p && member >= 0 && member < (int) p->device.size()) ? p->device[(size_t) member] : -1; }
const char* fluhip_pool_last_error(const fluhip_pool* p) { return p ? p->err.c_str() : "null pool"; }

namespace {
struct Worker
{
  std::atomic<int64_t> done{0};
  std::atomic<bool>* cancel = nullptr;
  std::atomic<int> rc{FLUHIP_OK};
  std::string err; // written before rc is stored (release), read after rc is seen (acquire) or after the join
};
int worker_progress(int64_t it, void* u)
{
  Worker* w = static_cast<Worker*>(u);
  w->done.store(it, std::memory_order_release);
  return w->cancel->load(std::memory_order_acquire) ? 0 : 1;
}
} // namespace

int fluhip_pool_bufnmf_job_f32(fluhip_pool* p, const fluhip_bufnmf_job* job, fluhip_progress_fn progress, void* user)
{
  if (!p) return FLUHIP_ERROR;
  p->err.clear();
  if (!job || !job->audio || job->count < 1 || job->n < 1) { p->err = "null / empty corpus"; return FLUHIP_ERROR; }
  const fluhip_bufnmf_job j = *job;
  const int world = (int) p->ctx.size();
  const int64_t F = j.fft / 2 + 1, T = fluhip_stft_num_frames(j.n, j.win, j.hop);
  std::vector<Worker> ws((size_t) world);
  std::atomic<bool> cancel{false};
  std::vector<std::thread> th;
  std::vector<int> active;
  for (int r = 0; r < world; r++)
  {
    int64_t b0, b1;
    fluhip_shard_range(j.count, world, r, &b0, &b1);
    if (b1 <= b0) { ws[(size_t) r].done.store(j.iters); continue; }
    active.push_back(r);
    ws[(size_t) r].cancel = &cancel;
    th.emplace_back([=, &ws] {
      Worker& w = ws[(size_t) r];
      fluhip_ctx* ctx = p->ctx[(size_t) r];
      // A share goes in slices: a corpus handle holds at most 65535 buffers, and without a progress callback (whose
      // "iteration i" spans the whole share) large shares are cut into slices of 256 buffers so that slice i + 1 uploads --
      // on the context's copy stream -- while slice i iterates: the factor updates are enqueued asynchronously, the upload
      // of the next slice blocks only this host thread, the write-back of slice i then waits for its iterations
      // (SURVEY section 7 step 5).  Buffers are independent jobs: slicing changes no result.
      static const bool slicesOff = [] { const char* e = fluhip::ab_getenv("FLUHIP_POOL_SLICES"); return e && std::atoi(e) == 0; }(); // A/B
      const int64_t sliceMax = (!progress && !slicesOff && b1 - b0 >= 512) ? 256 : 65535;
      auto prepare = [&](int64_t s0, int64_t nb, fluhip_corpus** out) -> int {
        int rc = fluhip_corpus_create(ctx, nb, j.n, j.win, j.fft, j.hop, j.K, out);
        if (rc == FLUHIP_OK && j.resynth) rc = fluhip_corpus_keep_spectrum(*out, 1);
        if (rc == FLUHIP_OK) rc = fluhip_corpus_set_audio_host(*out, j.audio + s0 * j.n);
        return rc;
      };
      fluhip_corpus* c = nullptr;
      int rc = prepare(b0, std::min<int64_t>(sliceMax, b1 - b0), &c);
      for (int64_t s0 = b0; s0 < b1 && rc == FLUHIP_OK && !w.cancel->load(std::memory_order_acquire); s0 += sliceMax)
      {
        rc = fluhip_corpus_stft(c);
        if (rc == FLUHIP_OK && (j.bases_seed || j.acts_seed)) // clients/nrt/NMFClient.hpp:246-258
          rc = fluhip_corpus_set_factors(c, j.bases_seed ? j.bases_seed + s0 * j.K * F : nullptr,
                                         j.acts_seed ? j.acts_seed + s0 * j.K * T : nullptr);
        if (rc == FLUHIP_OK)
          rc = fluhip_corpus_nmf(c, j.iters, j.update_w, j.update_h, j.seed, j.seeds ? j.seeds + s0 : nullptr,
                                 progress ? worker_progress : nullptr, progress ? &w : nullptr);
        fluhip_corpus* next = nullptr;
        if (rc == FLUHIP_OK && s0 + sliceMax < b1) // the next slice's upload, beside this slice's iterations
          rc = prepare(s0 + sliceMax, std::min<int64_t>(sliceMax, b1 - s0 - sliceMax), &next);
        if (rc == FLUHIP_OK)
          rc = fluhip_corpus_writeback_host(c, j.bases ? j.bases + s0 * j.K * F : nullptr, j.acts ? j.acts + s0 * j.K * T : nullptr);
        if (rc == FLUHIP_OK && j.resynth) rc = fluhip_corpus_resynth_host(c, j.resynth + s0 * j.K * j.n); // :302-334
        if (rc != FLUHIP_OK) w.err = fluhip_last_error(ctx);
        if (c) fluhip_corpus_destroy(c);
        c = next;
      }
      if (rc != FLUHIP_OK && w.err.empty()) w.err = fluhip_last_error(ctx);
      if (c) fluhip_corpus_destroy(c);
      if (rc != FLUHIP_OK)
      {
        w.rc.store(rc, std::memory_order_release);
        if (rc != FLUHIP_CANCELLED) w.cancel->store(true, std::memory_order_release); // a failed share stops the others
      }
      if (w.rc.load() == FLUHIP_OK) w.done.store(j.iters, std::memory_order_release);
    });
  }
  // progress of the whole job = the slowest device; reported in order from the calling thread (alg/NMF.hpp:175-176).  A share
  // that failed reports nothing further: the job ends with its error instead of counting up to `iters`.
  int64_t reported = 0;
  if (progress)
  {
    bool running = true;
    while (running)
    {
      int64_t m = j.iters;
      bool failed = false;
      for (int r : active)
      {
        m = std::min(m, ws[(size_t) r].done.load(std::memory_order_acquire));
        failed = failed || ws[(size_t) r].rc.load(std::memory_order_acquire) != FLUHIP_OK;
      }
      for (; reported < m && !failed; reported++)
        if (!cancel.load() && !progress(reported + 1, user)) cancel.store(true, std::memory_order_release);
      running = reported < j.iters && !cancel.load() && !failed;
      if (running) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
  for (auto& t : th) t.join();
  int rc = FLUHIP_OK;
  for (int r : active)
  {
    const int wrc = ws[(size_t) r].rc.load();
    if (wrc != FLUHIP_OK && rc != FLUHIP_ERROR)
    {
      rc = wrc;
      if (rc == FLUHIP_ERROR) p->err = "device " + std::to_string(p->device[(size_t) r]) + ": " + ws[(size_t) r].err;
    }
  }
  if (rc == FLUHIP_OK && cancel.load()) rc = FLUHIP_CANCELLED;
  if (rc == FLUHIP_CANCELLED && p->err.empty()) p->err = "cancelled";
  return rc;
}

int fluhip_pool_bufnmf_f32(fluhip_pool* p, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                           int64_t hop, int64_t K, int64_t iters, int update_w, int update_h, int64_t seed,
                           const int64_t* seeds, float* bases, float* acts, fluhip_progress_fn progress, void* user)
{
  fluhip_bufnmf_job j{};
  j.count = count; j.n = n; j.win = win; j.fft = fft; j.hop = hop; j.K = K; j.iters = iters;
  j.update_w = update_w; j.update_h = update_h; j.seed = seed; j.seeds = seeds;
  j.audio = audio; j.bases = bases; j.acts = acts;
  return fluhip_pool_bufnmf_job_f32(p, &j, progress, user);
}

// Ragged corpus: buffers of different lengths.  Dealt by the greedy longest-processing-time rule over cost = frames (the
// work of a buffer is ~ T F K, F and K being common); every device runs ITS buffers as one ragged corpus
// (fluhip_corpus_create_ragged: one STFT launch, one set of factor-update launches per iteration, the work dealt per
// wavefront by each buffer's own length).  Shapes the ragged form does not cover (ranks above 128, fft sizes without a
// block STFT) fall back to runs of equal length as equal-length corpora.
namespace {
// the fallback: runs of equal length as one corpus each, the others one by one
int ragged_by_groups(fluhip_ctx* ctx, const std::vector<int64_t>& mine, const float* const* audio, const int64_t* n, int64_t win,
                     int64_t fft, int64_t hop, int64_t K, int64_t iters, int update_w, int update_h, int64_t seed,
                     const int64_t* seeds, float* const* bases, float* const* acts, std::atomic<int64_t>& buffersDone,
                     std::atomic<bool>& cancel, std::string& err)
{
  const int64_t F = fft / 2 + 1;
  std::vector<float> packed, gb, ga;
  std::vector<int64_t> gseeds;
  for (size_t i0 = 0; i0 < mine.size() && !cancel.load(std::memory_order_acquire);)
  {
    size_t i1 = i0;
    while (i1 < mine.size() && n[mine[i1]] == n[mine[i0]] && i1 - i0 < 65535) i1++;
    const int64_t nb = (int64_t) (i1 - i0), len = n[mine[i0]];
    const int64_t T = fluhip_stft_num_frames(len, win, hop);
    const float* src = audio[mine[i0]];
    if (nb > 1)
    {
      packed.resize((size_t) nb * (size_t) len);
      for (int64_t j = 0; j < nb; j++)
        std::copy(audio[mine[i0 + (size_t) j]], audio[mine[i0 + (size_t) j]] + len, packed.begin() + (size_t) j * (size_t) len);
      src = packed.data();
    }
    if (seeds)
    {
      gseeds.resize((size_t) nb);
      for (int64_t j = 0; j < nb; j++) gseeds[(size_t) j] = seeds[mine[i0 + (size_t) j]];
    }
    fluhip_corpus* c = nullptr;
    int rc = fluhip_corpus_create(ctx, nb, len, win, fft, hop, K, &c);
    if (rc == FLUHIP_OK) rc = fluhip_corpus_set_audio_host(c, src);
    if (rc == FLUHIP_OK) rc = fluhip_corpus_stft(c);
    if (rc == FLUHIP_OK) rc = fluhip_corpus_nmf(c, iters, update_w, update_h, seed, seeds ? gseeds.data() : nullptr, nullptr, nullptr);
    if (rc == FLUHIP_OK)
    {
      float* bdst = nullptr;
      float* adst = nullptr;
      if (nb == 1) { bdst = bases ? bases[mine[i0]] : nullptr; adst = acts ? acts[mine[i0]] : nullptr; }
      else
      {
        if (bases) { gb.resize((size_t) (nb * K * F)); bdst = gb.data(); }
        if (acts) { ga.resize((size_t) (nb * K * T)); adst = ga.data(); }
      }
      rc = fluhip_corpus_writeback_host(c, bdst, adst);
      if (rc == FLUHIP_OK && nb > 1)
        for (int64_t j = 0; j < nb; j++)
        {
          const int64_t g = mine[i0 + (size_t) j];
          if (bases && bases[g]) std::copy(gb.begin() + (size_t) (j * K * F), gb.begin() + (size_t) ((j + 1) * K * F), bases[g]);
          if (acts && acts[g]) std::copy(ga.begin() + (size_t) (j * K * T), ga.begin() + (size_t) ((j + 1) * K * T), acts[g]);
        }
    }
    if (rc != FLUHIP_OK) err = fluhip_last_error(ctx);
    if (c) fluhip_corpus_destroy(c);
    if (rc != FLUHIP_OK) return rc;
    buffersDone.fetch_add(nb, std::memory_order_release);
    i0 = i1;
  }
  return FLUHIP_OK;
}

struct RaggedWorker
{
  std::atomic<int64_t> done{0}; // iterations completed by this device's corpus
  std::atomic<bool>* cancel = nullptr;
};
int ragged_progress(int64_t it, void* u)
{
  RaggedWorker* w = static_cast<RaggedWorker*>(u);
  w->done.store(it, std::memory_order_release);
  return w->cancel->load(std::memory_order_acquire) ? 0 : 1;
}
} // namespace

int fluhip_pool_bufnmf_ragged_f32(fluhip_pool* p, const float* const* audio, const int64_t* n, int64_t count, int64_t win,
                                  int64_t fft, int64_t hop, int64_t K, int64_t iters, int update_w, int update_h, int64_t seed,
                                  const int64_t* seeds, float* const* bases, float* const* acts, fluhip_progress_fn progress,
                                  void* user)
{
  if (!p) return FLUHIP_ERROR;
  p->err.clear();
  if (!audio || !n || count < 1) { p->err = "null / empty corpus"; return FLUHIP_ERROR; }
  for (int64_t i = 0; i < count; i++)
    if (!audio[i] || n[i] < 1) { p->err = "buffer " + std::to_string(i) + ": null or empty"; return FLUHIP_ERROR; }
  const int world = (int) p->ctx.size();
  std::vector<double> cost((size_t) count);
  for (int64_t i = 0; i < count; i++) cost[(size_t) i] = (double) fluhip_stft_num_frames(n[i], win, hop);
  std::vector<int32_t> owner((size_t) count);
  if (fluhip_balanced_assignment(cost.data(), count, world, owner.data()) != FLUHIP_OK) return FLUHIP_ERROR;

  std::atomic<int64_t> buffersDone{0};
  std::atomic<bool> cancel{false};
  std::vector<std::atomic<int>> rcs((size_t) world); // polled by the calling thread while the workers run
  for (auto& x : rcs) x.store(FLUHIP_OK);
  std::vector<std::string> errs((size_t) world);
  std::vector<std::thread> th;
  for (int r = 0; r < world; r++)
  {
    std::vector<int64_t> mine;
    for (int64_t i = 0; i < count; i++)
      if (owner[(size_t) i] == r) mine.push_back(i);
    if (mine.empty()) continue;
    // longest first; ties in corpus order (deterministic)
    std::stable_sort(mine.begin(), mine.end(), [&](int64_t a, int64_t b) { return n[a] > n[b]; });
    th.emplace_back([=, &rcs, &errs, &buffersDone, &cancel] {
      fluhip_ctx* ctx = p->ctx[(size_t) r];
      int rc = FLUHIP_OK;
      // ---- the whole share as ONE ragged corpus (slices of 65535 buffers) ---------------------------------------
      bool fallback = false;
      for (size_t i0 = 0; i0 < mine.size() && rc == FLUHIP_OK && !fallback && !cancel.load(std::memory_order_acquire); i0 += 65535)
      {
        const size_t nb = std::min<size_t>(65535, mine.size() - i0);
        std::vector<int64_t> lens(nb), sd;
        std::vector<const float*> ap(nb);
        std::vector<float*> bp(nb, nullptr), cp(nb, nullptr);
        for (size_t j = 0; j < nb; j++)
        {
          const int64_t g = mine[i0 + j];
          lens[j] = n[g]; ap[j] = audio[g];
          if (bases) bp[j] = bases[g];
          if (acts) cp[j] = acts[g];
        }
        if (seeds) { sd.resize(nb); for (size_t j = 0; j < nb; j++) sd[j] = seeds[mine[i0 + j]]; }
        fluhip_corpus* c = nullptr;
        rc = fluhip_corpus_create_ragged(ctx, (int64_t) nb, lens.data(), win, fft, hop, K, &c);
        if (rc != FLUHIP_OK && i0 == 0) { fallback = true; rc = FLUHIP_OK; break; } // a shape the ragged form does not cover
        if (rc == FLUHIP_OK) rc = fluhip_corpus_set_audio_ragged_host(c, ap.data());
        if (rc == FLUHIP_OK) rc = fluhip_corpus_stft(c);
        RaggedWorker w;
        w.cancel = &cancel;
        if (rc == FLUHIP_OK)
          // the per-iteration callback (one host round trip per iteration, no round-major / graph scheduling) only where a
          // refusal can arrive: without a caller's callback the cancel flag can never be set
          rc = fluhip_corpus_nmf(c, iters, update_w, update_h, seed, seeds ? sd.data() : nullptr,
                                 progress ? ragged_progress : nullptr, progress ? &w : nullptr);
        if (rc == FLUHIP_OK) rc = fluhip_corpus_writeback_ragged_host(c, bases ? bp.data() : nullptr, acts ? cp.data() : nullptr);
        if (rc != FLUHIP_OK) errs[(size_t) r] = fluhip_last_error(ctx);
        if (c) fluhip_corpus_destroy(c);
        if (rc == FLUHIP_OK) buffersDone.fetch_add((int64_t) nb, std::memory_order_release);
      }
      if (fallback)
        rc = ragged_by_groups(ctx, mine, audio, n, win, fft, hop, K, iters, update_w, update_h, seed, seeds, bases, acts,
                              buffersDone, cancel, errs[(size_t) r]);
      if (rc != FLUHIP_OK && rc != FLUHIP_CANCELLED) rcs[(size_t) r].store(rc, std::memory_order_release);
    });
  }
  // progress = buffers finished so far (1 .. count), from the calling thread; a refusal stops every device (at its next
  // iteration in the ragged form, after the group it is working on in the fallback)
  if (progress)
  {
    int64_t reported = 0;
    for (;;)
    {
      const int64_t d = buffersDone.load(std::memory_order_acquire);
      bool stop = false;
      for (; reported < d && !stop; reported++)
        if (!progress(reported + 1, user)) { cancel.store(true, std::memory_order_release); stop = true; }
      if (stop || reported >= count) break;
      bool failed = false;
      for (int r = 0; r < world; r++) failed = failed || rcs[(size_t) r].load(std::memory_order_acquire) != FLUHIP_OK;
      if (failed) break;
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
  for (auto& t : th) t.join();
  int rc = FLUHIP_OK;
  for (int r = 0; r < world; r++)
    if (rcs[(size_t) r].load() != FLUHIP_OK && rc == FLUHIP_OK)
    {
      rc = rcs[(size_t) r].load();
      p->err = "device " + std::to_string(p->device[(size_t) r]) + ": " + errs[(size_t) r];
    }
  if (rc == FLUHIP_OK && cancel.load()) { rc = FLUHIP_CANCELLED; p->err = "cancelled"; }
  return rc;
}


// ---- feature pipeline over the pool (BASELINE config 5: "... over 8192 x 2 s slices ... 1 -> 8 GPU scaling") ----------------
// The slices are independent analyses (one StreamingControl run each, cc/FluidNRTClientWrapper.hpp:598-630): contiguous
// blocks of slices per device, every device reads its block of the caller's audio and writes its block of the features.
static int pool_features(fluhip_pool* p, bool mfcc, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                         int64_t hop, int64_t n_bands, int64_t n_coefs, int64_t start_coeff, double min_freq, double max_freq,
                         double sample_rate, int normalize, int scale_db, int padding_mode, float* out, int64_t* frames_out)
{
  if (!p) return FLUHIP_ERROR;
  p->err.clear();
  if (!audio || !out || count < 1 || n < 1 || win < 1 || hop < 1) { p->err = "null / empty corpus"; return FLUHIP_ERROR; }
  if (padding_mode < 0 || padding_mode > 2) { p->err = "padding mode must be 0 (None), 1 (Default) or 2 (Full)"; return FLUHIP_ERROR; }
  const int world = (int) p->ctx.size();
  // frames per slice, as the single-device entry points count them (include/flucoma_hip.h)
  const int64_t userPad = padding_mode == 0 ? 0 : padding_mode == 1 ? win / 2 : win - hop;
  int64_t padded = n + win + 2 * userPad;
  if (padding_mode == 2) padded = ((padded + hop - 1) / hop) * hop;
  const int64_t T = 1 + (padded - win) / hop - win / hop;
  if (T < 1) { p->err = "not enough frames"; return FLUHIP_ERROR; }
  const int64_t nOut = mfcc ? n_coefs : n_bands;
  std::vector<int> rcs((size_t) world, FLUHIP_OK);
  std::vector<std::string> errs((size_t) world);
  std::vector<std::thread> th;
  for (int r = 0; r < world; r++)
  {
    int64_t b0, b1;
    fluhip_shard_range(count, world, r, &b0, &b1);
    if (b1 <= b0) continue;
    th.emplace_back([=, &rcs, &errs] {
      fluhip_ctx* ctx = p->ctx[(size_t) r];
      int64_t Tr = 0;
      const float* a = audio + b0 * n;
      float* o = out + b0 * nOut * T;
      const int rc = mfcc ? fluhip_bufmfcc_padded_f32(ctx, a, b1 - b0, n, win, fft, hop, n_bands, n_coefs, start_coeff, min_freq,
                                                      max_freq, sample_rate, padding_mode, o, &Tr)
                          : fluhip_bufmelbands_padded_f32(ctx, a, b1 - b0, n, win, fft, hop, n_bands, min_freq, max_freq,
                                                          sample_rate, normalize, scale_db, padding_mode, o, &Tr);
      rcs[(size_t) r] = rc;
      if (rc != FLUHIP_OK) errs[(size_t) r] = fluhip_last_error(ctx);
      else if (Tr != T) { rcs[(size_t) r] = FLUHIP_ERROR; errs[(size_t) r] = "frame count differs between the pool and the device path"; }
    });
  }
  for (auto& t : th) t.join();
  for (int r = 0; r < world; r++)
    if (rcs[(size_t) r] != FLUHIP_OK)
    {
      p->err = "device " + std::to_string(p->device[(size_t) r]) + ": " + errs[(size_t) r];
      return rcs[(size_t) r];
    }
  if (frames_out) *frames_out = T;
  return FLUHIP_OK;
}

int fluhip_pool_bufmfcc_f32(fluhip_pool* p, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft, int64_t hop,
                            int64_t n_bands, int64_t n_coefs, int64_t start_coeff, double min_freq, double max_freq,
                            double sample_rate, int padding_mode, float* out, int64_t* frames_out)
{
  return pool_features(p, true, audio, count, n, win, fft, hop, n_bands, n_coefs, start_coeff, min_freq, max_freq, sample_rate, 0, 0,
                       padding_mode, out, frames_out);
}

int fluhip_pool_bufmelbands_f32(fluhip_pool* p, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                                int64_t hop, int64_t n_bands, double min_freq, double max_freq, double sample_rate, int normalize,
                                int scale_db, int padding_mode, float* out, int64_t* frames_out)
{
  return pool_features(p, false, audio, count, n, win, fft, hop, n_bands, 0, 0, min_freq, max_freq, sample_rate, normalize, scale_db,
                       padding_mode, out, frames_out);
}

} // extern "C"
