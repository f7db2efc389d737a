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

#include <algorithm>
#include <atomic>
#include <chrono>
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
  int rc = FLUHIP_OK;
  std::string err;
};
int worker_progress(int64_t it, void* u)
{
  Worker* w = static_cast<Worker*>(u);
  w->done.store(it, std::memory_order_release);
  return w->cancel->load(std::memory_order_acquire) ? 0 : 1;
}
} // namespace

int fluhip_pool_bufnmf_f32(fluhip_pool* p, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                           int64_t hop, int64_t K, int64_t iters, int update_w, int update_h, int64_t seed,
                           const int64_t* seeds, float* bases, float* acts, fluhip_progress_fn progress, void* user)
{
  if (!p) return FLUHIP_ERROR;
  p->err.clear();
  if (!audio || count < 1 || n < 1) { p->err = "null / empty corpus"; return FLUHIP_ERROR; }
  const int world = (int) p->ctx.size();
  const int64_t F = fft / 2 + 1, T = fluhip_stft_num_frames(n, win, hop);
  std::vector<Worker> ws((size_t) world);
  std::atomic<bool> cancel{false};
  std::vector<std::thread> th;
  std::vector<int> active;
  for (int r = 0; r < world; r++)
  {
    int64_t b0, b1;
    fluhip_shard_range(count, world, r, &b0, &b1);
    if (b1 <= b0) { ws[(size_t) r].done.store(iters); continue; }
    active.push_back(r);
    ws[(size_t) r].cancel = &cancel;
    th.emplace_back([=, &ws] {
      Worker& w = ws[(size_t) r];
      fluhip_ctx* ctx = p->ctx[(size_t) r];
      // a corpus handle holds at most 65535 buffers: larger shares go in slices
      for (int64_t s0 = b0; s0 < b1 && w.rc == FLUHIP_OK; s0 += 65535)
      {
        const int64_t nb = std::min<int64_t>(65535, b1 - s0);
        fluhip_corpus* c = nullptr;
        int rc = fluhip_corpus_create(ctx, nb, n, win, fft, hop, K, &c);
        if (rc == FLUHIP_OK) rc = fluhip_corpus_set_audio_host(c, audio + s0 * n);
        if (rc == FLUHIP_OK) rc = fluhip_corpus_stft(c);
        if (rc == FLUHIP_OK)
          rc = fluhip_corpus_nmf(c, iters, update_w, update_h, seed, seeds ? seeds + s0 : nullptr,
                                 progress ? worker_progress : nullptr, progress ? &w : nullptr);
        if (rc == FLUHIP_OK)
          rc = fluhip_corpus_writeback_host(c, bases ? bases + s0 * K * F : nullptr, acts ? acts + s0 * K * T : nullptr);
        if (rc != FLUHIP_OK) { w.rc = rc; w.err = fluhip_last_error(ctx); }
        if (c) fluhip_corpus_destroy(c);
      }
      w.done.store(iters, std::memory_order_release);
    });
  }
  // progress of the whole job = the slowest device; reported in order from the calling thread (alg/NMF.hpp:175-176)
  int64_t reported = 0;
  if (progress)
  {
    bool running = true;
    while (running)
    {
      int64_t m = iters;
      for (int r : active) m = std::min(m, ws[(size_t) r].done.load(std::memory_order_acquire));
      for (; reported < m; reported++)
        if (!cancel.load() && !progress(reported + 1, user)) cancel.store(true, std::memory_order_release);
      running = reported < iters && !cancel.load();
      if (running) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
  for (auto& t : th) t.join();
  int rc = FLUHIP_OK;
  for (int r : active)
    if (ws[(size_t) r].rc != FLUHIP_OK && rc != FLUHIP_ERROR)
    {
      rc = ws[(size_t) r].rc;
      if (rc == FLUHIP_ERROR) p->err = "device " + std::to_string(p->device[(size_t) r]) + ": " + ws[(size_t) r].err;
    }
  if (rc == FLUHIP_OK && cancel.load()) rc = FLUHIP_CANCELLED;
  if (rc == FLUHIP_CANCELLED && p->err.empty()) p->err = "cancelled";
  return rc;
}

} // extern "C"
