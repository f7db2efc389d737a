// NMFClient.hpp -- BufNMF client over the MI355X C ABI (include/flucoma_hip.h).
//
// Mirrors client::bufnmf::NMFClient, include/flucoma/clients/nrt/NMFClient.hpp:36-343:
//   parameter table   :36-71     -> NMFParams (plain struct, same names / defaults / constraints)
//   process<T>()      :96-337    -> same order of checks, same Result codes and messages, same
//                                   output buffer shapes and sample rates; the per-channel body
//                                   (:240-300: STFT -> magnitude -> NMF -> float write-back) is one
//                                   call into libflucoma_hip.so instead of algorithm::STFT / NMF.
// There is no CPU path: if the library cannot create a context on the requested device the job
// returns kError.
#pragma once

#include "../flucoma_hip.h"
#include "BufferAdaptor.hpp"
#include "ParamDescriptors.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace fluhip {
namespace bufnmf {

// nrt/NMFClient.hpp:36-52
enum NMFParamIndex {
  kSource, kOffset, kNumFrames, kStartChan, kNumChans, kResynth, kResynthMode, kFilters, kFiltersUpdate,
  kEnvelopes, kEnvelopesUpdate, kRank, kIterations, kRandomSeed, kFFT
};

// nrt/NMFClient.hpp:54-71 (names, defaults and constraints of defineParameters(...))
struct NMFParams
{
  std::shared_ptr<const BufferAdaptor> source;            // "source"
  index                                startFrame{0};     // Min(0)
  index                                numFrames{-1};
  index                                startChan{0};      // Min(0)
  index                                numChans{-1};
  std::shared_ptr<BufferAdaptor>       resynth;           // "resynth"
  index                                resynthMode{0};    // 0..1
  std::shared_ptr<BufferAdaptor>       bases;             // "bases"
  index                                basesMode{0};      // None, Seed, Fixed
  std::shared_ptr<BufferAdaptor>       activations;       // "activations"
  index                                actMode{0};        // None, Seed, Fixed
  index                                components{1};     // Min(1)
  index                                iterations{100};   // Min(1)
  index                                seed{-1};
  FFTParams                            fftSettings{1024, -1, -1};

  // the buffer parameters, for the job layer's deep copies and copy-back (NRTThreadingAdaptor.hpp)
  template <class In, class Out>
  void forEachBuffer(In&& in, Out&& out)
  {
    forEachBuffer(in, out, out);
  }
  // (third visitor: buffers process() never reads -- it resizes the resynthesis buffer and fills every sample)
  template <class In, class Out, class OutOnly>
  void forEachBuffer(In&& in, Out&& out, OutOnly&& outOnly)
  {
    in(source);
    outOnly(resynth);
    out(bases);
    out(activations);
  }

  // the clamping the reference's constraints apply when a value is set
  void constrain()
  {
    startFrame = std::max<index>(0, startFrame);
    startChan = std::max<index>(0, startChan);
    resynthMode = std::min<index>(1, std::max<index>(0, resynthMode));
    basesMode = std::min<index>(2, std::max<index>(0, basesMode));
    actMode = std::min<index>(2, std::max<index>(0, actMode));
    components = std::max<index>(1, components);
    iterations = std::max<index>(1, iterations);
    // cc/ParameterTypes.hpp:371-393: win >= 4, fft a power of two >= nextPow2(win)
    fftSettings.win = std::max<index>(4, fftSettings.win);
    if (fftSettings.fft >= 0)
    {
      index p = 1;
      while (p < std::max(fftSettings.fft, fftSettings.win)) p <<= 1;
      fftSettings.fft = p;
    }
  }
};

class NMFClient
{
public:
  using ParamSetViewType = NMFParams;
  // the parameter table a host enumerates (nrt/NMFClient.hpp:81; ParamDescriptors.hpp)
  static constexpr ParamDescriptorList getParameterDescriptors() { return paramdesc::list(paramdesc::kBufNMF); }

  NMFClient(NMFParams& p, FluidContext&) : mParams(&p) {}
  ~NMFClient()
  {
    if (mCtx) fluhip_ctx_destroy(mCtx);
    for (auto& pc : mPool)
      if (pc.second) fluhip_ctx_destroy(pc.second);
  }
  NMFClient(const NMFClient&) = delete;
  NMFClient& operator=(const NMFClient&) = delete;

  void setParams(NMFParams& p) { mParams = &p; }

  template <typename T>
  Result process(FluidContext& c)
  {
    using S = Result::Status;
    struct WholeCall // FLUHIP_CLIENT_TIMING=1: the whole of process(), destructors of its locals included
    {
      std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      ~WholeCall()
      {
        if (std::getenv("FLUHIP_CLIENT_TIMING"))
          std::fprintf(stderr, "  client %-22s %8.3f ms\n", "process(), whole",
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
    } wholeCall;
    const NMFParams& P = *mParams;
    index            nFrames = P.numFrames;
    index            nChannels = P.numChans;
    Result           rangeCheck = bufferRangeCheck(P.source.get(), P.startFrame, nFrames, P.startChan, nChannels);
    if (!rangeCheck.ok()) return rangeCheck;

    BufferAdaptor::ReadAccess source(P.source.get());
    const double              sampleRate = source.sampleRate();
    const FFTParams           fftParams = P.fftSettings;
    const index               hop = fftParams.hopSize();
    const index               nWindows = (nFrames + hop) / hop; // :111-112
    const index               nBins = fftParams.frameSize();
    const index               rank = P.components;

    bool       hasFilters = false;
    const bool seedFilters = P.basesMode > 0;
    const bool fixFilters = P.basesMode == 2;
    const bool shouldResynth = P.resynthMode != 0;

    if (P.bases)
    {
      BufferAdaptor::Access buf(P.bases.get());
      if (!buf.exists()) return {S::kError, "Bases Buffer Supplied But Invalid"};
      if (P.basesMode > 0 && (!buf.valid() || buf.numFrames() != nBins || buf.numChans() != rank * nChannels))
        return {S::kError, "Supplied bases buffer for seeding must be [(FFTSize / 2) + "
                           "1] frames long, and have [rank] * [channels] channels"};
      hasFilters = true;
    }
    else if (P.basesMode > 0)
      return {S::kError, "Bases Mode set to Seed or Fix , but no Bases Buffer supplied"};

    bool       hasEnvelopes = false;
    const bool seedEnvelopes = P.actMode > 0;
    const bool fixEnvelopes = P.actMode == 2;
    const bool needsAnalysis = !(fixEnvelopes && fixFilters);

    if (!needsAnalysis && !shouldResynth)
      return {S::kWarning, "Bases and Activations buffers both fixed, but resynthesis disabled: no work to do"};

    if (P.activations)
    {
      BufferAdaptor::Access buf(P.activations.get());
      if (!buf.exists()) return {S::kError, "Activations Buffer Supplied But Invalid"};
      if (P.actMode > 0 &&
          (!buf.valid() || buf.numFrames() != (nFrames / hop) + 1 || buf.numChans() != rank * nChannels))
        return {S::kError, "Supplied activations buffer for seeding must be [(num samples / hop "
                           "size)  + 1] frames long, and have [rank] * [channels] channels"};
      hasEnvelopes = true;
    }
    else if (P.actMode > 0)
      return {S::kError, "Activations Mode set to Seed or Fix , but no Activations Buffer supplied"};

    bool hasResynth = false;
    if (shouldResynth)
    {
      if (!P.resynth) return {S::kError, "Resynthesis requested but no buffer supplied"};
      BufferAdaptor::Access buf(P.resynth.get());
      if (!buf.exists()) return {S::kError, "Resynthesis Buffer Supplied But Invalid"};
      hasResynth = true;
      Result r = buf.resize(nFrames, nChannels * rank, sampleRate);
      if (!r.ok()) return r;
    }
    if (hasFilters && P.basesMode == 0)
    {
      Result r = BufferAdaptor::Access(P.bases.get()).resize(nBins, nChannels * rank, sampleRate / fftParams.fftSize());
      if (!r.ok()) return r;
    }
    if (hasEnvelopes && P.actMode == 0)
    {
      Result r = BufferAdaptor::Access(P.activations.get()).resize((nFrames / hop) + 1, nChannels * rank, sampleRate / hop);
      if (!r.ok()) return r;
    }

    // device context (one per client, like the per-client algorithm objects of the reference)
    if (!mCtx || mDevice != c.device())
    {
      if (mCtx) fluhip_ctx_destroy(mCtx);
      mCtx = nullptr;
      if (fluhip_ctx_create(c.device(), &mCtx) != FLUHIP_OK)
        return {S::kError, "BufNMF: no usable MI355X device ", c.device(), " (libflucoma_hip has no CPU fallback)"};
      mDevice = c.device();
    }

    // scratch of the channel-by-channel loop (sized where that loop starts: the batched path does not use it, and
    // rank x nFrames zero-filled floats were 9 ms of an 8-channel x 10 s job)
    std::vector<float> mono, seedW, seedH, outW, outH, outR;

    const double progressTotal =
        static_cast<double>((needsAnalysis ? P.iterations : 0) + ((shouldResynth && hasResynth) ? 3 * rank : 0)); // :229-230

    struct Prog
    {
      FluidContext* c;
      int           count;
      double        total;
      double        floor = 0.0; // least count to report (a job that fell back from the batched block does not report less than it had)
    };
    // fraction of the whole job the batched block had reported when it handed the job to the channel loop (0: it never ran)
    double progressFloor = 0.0;

    // ---- several devices: the channels run concurrently, one host thread + context per listed device --------------
    // (the reference runs them one after the other on its single thread, :233; they share no state -- a fresh
    //  algorithm::NMF per channel, :260 -- so the results are the same floats in the same buffer channels)
    if (c.devices().size() > 1 && nChannels > 1)
    {
      const size_t nDev = c.devices().size();
      if (mPool.size() != nDev) { for (auto& pc : mPool) if (pc.second) fluhip_ctx_destroy(pc.second); mPool.assign(nDev, {-1, nullptr}); }
      for (size_t d = 0; d < nDev; ++d)
        if (!mPool[d].second || mPool[d].first != c.devices()[d])
        {
          if (mPool[d].second) fluhip_ctx_destroy(mPool[d].second);
          mPool[d] = {c.devices()[d], nullptr};
          if (fluhip_ctx_create(c.devices()[d], &mPool[d].second) != FLUHIP_OK)
            return {S::kError, "BufNMF: no usable MI355X device ", c.devices()[d], " (libflucoma_hip has no CPU fallback)"};
        }
      const size_t nc = static_cast<size_t>(nChannels);
      std::vector<std::vector<float>> monoC(nc), seedWC(nc), seedHC(nc), outWC(nc), outHC(nc), outRC(nc);
      for (index i = 0; i < nChannels; ++i) // all host-buffer reads on this thread, before the workers start
      {
        const size_t ci = static_cast<size_t>(i);
        monoC[ci].resize(static_cast<size_t>(nFrames));
        VectorView<float>(monoC[ci].data(), nFrames) <<= source.samps(P.startFrame, nFrames, P.startChan + i);
        if (seedFilters) seedWC[ci].resize(static_cast<size_t>(rank * nBins));
        if (seedEnvelopes) seedHC[ci].resize(static_cast<size_t>(rank * nWindows));
        for (index j = 0; j < rank; ++j)
        {
          if (seedFilters)
            VectorView<float>(seedWC[ci].data() + j * nBins, nBins) <<=
                VectorView<const float>(BufferAdaptor::Access(P.bases.get()).samps(i * rank + j));
          if (seedEnvelopes)
            VectorView<float>(seedHC[ci].data() + j * nWindows, nWindows) <<=
                VectorView<const float>(BufferAdaptor::Access(P.activations.get()).samps(i * rank + j));
        }
        if (hasFilters && !fixFilters) outWC[ci].resize(static_cast<size_t>(rank * nBins));
        if (hasEnvelopes && !fixEnvelopes) outHC[ci].resize(static_cast<size_t>(rank * nWindows));
        if (shouldResynth && hasResynth) outRC[ci].resize(static_cast<size_t>(rank * nFrames));
      }
      struct Shared
      {
        FluidContext* c;
        std::mutex    m;
        double        count{0};
        double        total;
      } shared{&c, {}, 0, progressTotal * static_cast<double>(nChannels)};
      auto cbShared = [](int64_t, void* u) -> int {
        auto* sh = static_cast<Shared*>(u);
        if (!sh->c->task()) return 1;
        std::lock_guard<std::mutex> g(sh->m);
        sh->count += 1;
        return sh->c->task()->processUpdate(sh->count, sh->total) ? 1 : 0;
      };
      std::vector<int>         rcs(nc, FLUHIP_OK);
      std::vector<std::string> errs(nc);
      std::vector<std::thread> workers;
      const index              iters = needsAnalysis ? P.iterations : 0;
      for (size_t d = 0; d < nDev; ++d)
        workers.emplace_back([&, d] {
          for (size_t ci = d; ci < nc; ci += nDev)
          {
            rcs[ci] = fluhip_bufnmf_channel_f32(
                mPool[d].second, monoC[ci].data(), nFrames, 1, fftParams.winSize(), fftParams.fftSize(), hop, rank, iters,
                !fixFilters, !fixEnvelopes, P.seed, seedFilters ? seedWC[ci].data() : nullptr,
                seedEnvelopes ? seedHC[ci].data() : nullptr, outWC[ci].empty() ? nullptr : outWC[ci].data(),
                outHC[ci].empty() ? nullptr : outHC[ci].data(), outRC[ci].empty() ? nullptr : outRC[ci].data(), cbShared, &shared);
            if (rcs[ci] != FLUHIP_OK) { errs[ci] = fluhip_last_error(mPool[d].second); break; }
          }
        });
      for (auto& w : workers) w.join();
      for (size_t ci = 0; ci < nc; ++ci)
      {
        if (rcs[ci] == FLUHIP_CANCELLED || (c.task() && c.task()->cancelled())) return {S::kCancelled, ""};
        if (rcs[ci] != FLUHIP_OK) return {S::kError, "BufNMF: ", errs[ci]};
      }
      for (index i = 0; i < nChannels; ++i) // all host-buffer writes on this thread, in channel order (:277-334)
      {
        const size_t ci = static_cast<size_t>(i);
        if (hasFilters && !fixFilters)
        {
          BufferAdaptor::Access filters(P.bases.get());
          for (index j = 0; j < rank; ++j)
            filters.samps(i * rank + j) <<= VectorView<const float>(outWC[ci].data() + j * nBins, nBins);
        }
        if (hasEnvelopes && !fixEnvelopes)
        {
          BufferAdaptor::Access envelopes(P.activations.get());
          for (index j = 0; j < rank; ++j)
            envelopes.samps(i * rank + j) <<= VectorView<const float>(outHC[ci].data() + j * nWindows, nWindows);
        }
        if (shouldResynth && hasResynth)
        {
          BufferAdaptor::Access resynth(P.resynth.get());
          scatterChannels(resynth, i * rank, rank, outRC[ci].data(), nFrames);
          for (index j = 0; j < rank; ++j)
          {
            // the 3 steps per component of :321-332 are part of `total`: reported here, so that progress reaches 1
            for (int step = 0; step < 3; ++step)
            {
              shared.count += 1;
              if (c.task() && !c.task()->processUpdate(shared.count, shared.total)) return {S::kCancelled, ""};
            }
          }
        }
      }
      return {S::kOk, ""};
    }

    // ---- one device, several channels: ONE corpus on the batched kernels ----------------------------------------------
    // The channels of a buffer are equal-shape independent jobs by construction (:233; a fresh algorithm::NMF per
    // channel, :260), i.e. exactly what the corpus entry points batch: one STFT launch, every factor-update launch over
    // all channels, Seed / Fixed factors through fluhip_corpus_set_factors.  Same floats per channel as the sequential
    // loop below up to the summation order of the schedule the batch gets.  Progress: an iteration of the batch is one
    // iteration of every channel, so it counts nChannels of the progressTotal * nChannels steps (the arithmetic of
    // :261-267 summed over the channel loop).  FLUHIP_CLIENT_SEQUENTIAL=1 keeps the channel-by-channel loop (A/B, tests);
    // a corpus that does not fit the device falls back to it as well.
    if (!sequentialForced()) // (a mono job is a corpus of one: the same path, and its resynthesis reaches an interleaved buffer in place)
    {
      const size_t nc = static_cast<size_t>(nChannels);
      // FLUHIP_CLIENT_TIMING=1: wall time of the phases of the batched path on stderr (measurement aid)
      const bool timing = std::getenv("FLUHIP_CLIENT_TIMING") != nullptr;
      auto       tPrev = wholeCall.t0;
      auto       lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  client %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - tPrev).count());
        tPrev = now;
      };
      lap("checks + resizes");
      std::vector<float> audioAll(nc * static_cast<size_t>(nFrames));
      std::vector<float> seedWAll, seedHAll;
      if (seedFilters) seedWAll.resize(nc * static_cast<size_t>(rank * nBins));
      if (seedEnvelopes) seedHAll.resize(nc * static_cast<size_t>(rank * nWindows));
      {
        // :240 for every channel.  The channels of a host buffer are usually interleaved (MemoryBufferAdaptor: frames x
        // channels), so a channel at a time would pull the whole buffer through the cache once per channel: the frames go
        // in blocks that stay cached across the channel views instead.
        std::vector<VectorView<const float>> chan;
        for (index i = 0; i < nChannels; ++i) chan.push_back(source.samps(P.startFrame, nFrames, P.startChan + i));
        constexpr index kBlock = 4096;
        for (index t0 = 0; t0 < nFrames; t0 += kBlock)
        {
          const index nb = std::min<index>(kBlock, nFrames - t0);
          for (index i = 0; i < nChannels; ++i)
          {
            const float* src = chan[static_cast<size_t>(i)].data() + t0 * chan[static_cast<size_t>(i)].stride;
            const index  st = chan[static_cast<size_t>(i)].stride;
            float*       dst = audioAll.data() + i * nFrames + t0;
            for (index t = 0; t < nb; ++t) dst[t] = src[t * st];
          }
        }
      }
      for (index i = 0; i < nChannels; ++i)
      {
        for (index j = 0; j < rank; ++j)
        {
          if (seedFilters)
            VectorView<float>(seedWAll.data() + (i * rank + j) * nBins, nBins) <<=
                VectorView<const float>(BufferAdaptor::Access(P.bases.get()).samps(i * rank + j));
          if (seedEnvelopes)
            VectorView<float>(seedHAll.data() + (i * rank + j) * nWindows, nWindows) <<=
                VectorView<const float>(BufferAdaptor::Access(P.activations.get()).samps(i * rank + j));
        }
      }
      lap("gather channels");
      fluhip_corpus* cor = nullptr;
      fluhip_clear_error(mCtx);
      if (fluhip_corpus_create(mCtx, nChannels, nFrames, fftParams.winSize(), fftParams.fftSize(), hop, rank, &cor) == FLUHIP_OK)
      {
        struct Guard
        {
          fluhip_corpus* c;
          ~Guard() { fluhip_corpus_destroy(c); }
        } guard{cor};
        lap("corpus create");
        if (c.task() && !c.task()->iterationUpdate(0.0, 1.0)) return {S::kCancelled, ""};
        const bool wantResynth = shouldResynth && hasResynth;
        int        rc = FLUHIP_OK;
        if (wantResynth) rc = fluhip_corpus_keep_spectrum(cor, 1);
        if (rc == FLUHIP_OK) rc = fluhip_corpus_set_audio_host(cor, audioAll.data());
        if (rc == FLUHIP_OK) rc = fluhip_corpus_stft(cor);                                            // :240-242, all channels
        lap("upload + stft");
        if (rc == FLUHIP_OK)
          rc = fluhip_corpus_set_factors(cor, seedFilters ? seedWAll.data() : nullptr, seedEnvelopes ? seedHAll.data() : nullptr);
        struct BatchProg
        {
          FluidContext* c;
          double        count, total, perIteration;
        } prog{&c, 0.0, progressTotal * static_cast<double>(nChannels), static_cast<double>(nChannels)};
        auto cb = [](int64_t, void* u) -> int {
          auto* p = static_cast<BatchProg*>(u);
          p->count += p->perIteration;
          return p->c->task() ? (p->c->task()->processUpdate(p->count, p->total) ? 1 : 0) : 1;
        };
        // a negative seed means a fresh std::random_device draw per NMF object (:260, util/EigenRandom.hpp:87): one per channel
        std::vector<int64_t> perChannelSeeds;
        if (P.seed < 0) perChannelSeeds.assign(nc, -1);
        if (rc == FLUHIP_OK)
          rc = fluhip_corpus_nmf(cor, needsAnalysis ? P.iterations : 0, !fixFilters, !fixEnvelopes, P.seed,
                                 perChannelSeeds.empty() ? nullptr : perChannelSeeds.data(), cb, &prog);      // :268-271
        lap("nmf");
        if (rc == FLUHIP_CANCELLED || (c.task() && c.task()->cancelled())) return {S::kCancelled, ""};        // :273-274
        bool injectedAllocFailure = false;
#ifdef FLUHIP_AB_SWITCHES // tests: a failure of the batched block behind the iterations -- "1": as an out-of-memory allocation of
                          // the resynthesis would be (falls back); "other": any other error (reported at once)
        if (rc == FLUHIP_OK)
          if (const char* inj = std::getenv("FLUHIP_CLIENT_FAIL_BATCHED"))
          {
            rc = FLUHIP_ERROR;
            injectedAllocFailure = std::strcmp(inj, "other") != 0;
          }
#endif
        std::vector<float> outWAll, outHAll, outRAll;
        if (hasFilters && !fixFilters) outWAll.resize(nc * static_cast<size_t>(rank * nBins));
        if (hasEnvelopes && !fixEnvelopes) outHAll.resize(nc * static_cast<size_t>(rank * nWindows));
        if (rc == FLUHIP_OK)
          rc = fluhip_corpus_writeback_host(cor, outWAll.empty() ? nullptr : outWAll.data(),
                                            outHAll.empty() ? nullptr : outHAll.data());                    // :277-300
        bool resynthInPlace = false;
        if (rc == FLUHIP_OK && wantResynth)
        {
          // :302-334.  An interleaved host buffer (frames x channels) takes the result as the device wrote it, in one
          // streaming copy; any other layout gets it channel-major and is filled block by block below
          BufferAdaptor::Access resynth(P.resynth.get());
          index                 frameStride = 0;
          float*                base = resynth.numChans() == nChannels * rank && resynth.numFrames() == nFrames
                                           ? interleavedBase(resynth, frameStride)
                                           : nullptr;
          if (base)
          {
            rc = fluhip_corpus_resynth_interleaved_host(cor, base, frameStride);
            resynthInPlace = rc == FLUHIP_OK;
          }
          else
          {
            outRAll.resize(nc * static_cast<size_t>(rank * nFrames));
            rc = fluhip_corpus_resynth_host(cor, outRAll.data());
          }
        }
        // Any failure of the batched block other than a cancellation -- the kept spectrum (as large as both magnitude
        // copies), the resynthesis output and its transposed copy (2 x channels x nFrames floats), the mask workspace: all
        // sized for every channel at once -- sends the job to the channel-by-channel loop below, which needs one channel's
        // worth of each and rewrites every output buffer from scratch (nothing of this block's results has been handed
        // to the host's buffers as a finished channel yet).  If one channel does not fit either, that loop reports it.
        if (rc == FLUHIP_OK)
        {
        lap("write-back");
        for (index i = 0; i < nChannels; ++i) // buffer writes in channel order, as the reference's loop leaves them
        {
          if (!outWAll.empty())
          {
            BufferAdaptor::Access filters(P.bases.get());
            for (index j = 0; j < rank; ++j)
              filters.samps(i * rank + j) <<= VectorView<const float>(outWAll.data() + (i * rank + j) * nBins, nBins);
          }
          if (!outHAll.empty())
          {
            BufferAdaptor::Access envelopes(P.activations.get());
            for (index j = 0; j < rank; ++j)
              envelopes.samps(i * rank + j) <<= VectorView<const float>(outHAll.data() + (i * rank + j) * nWindows, nWindows);
          }
          if (wantResynth)
          {
            BufferAdaptor::Access resynth(P.resynth.get());
            if (!resynthInPlace) scatterChannels(resynth, i * rank, rank, outRAll.data() + i * rank * nFrames, nFrames);
            for (index j = 0; j < rank; ++j)
            {
              for (int step = 0; step < 3; ++step)
              {
                prog.count += 1;
                if (c.task() && !c.task()->processUpdate(prog.count, prog.total)) return {S::kCancelled, ""};
              }
            }
          }
        }
        lap("scatter to buffers");
        return {S::kOk, ""};
        }
        // only an ALLOCATION failure is worth a second attempt with one channel's worth of memory; a device fault or
        // an argument error would fail again after a whole second job, with the first message lost and the progress restarted
        {
          // (classified by the library at the point of failure -- hipErrorOutOfMemory, the FFT workspace, a host
          //  bad_alloc -- not by message text; the flag was cleared in front of the block, so nothing stale is read)
          const char* why = fluhip_last_error(mCtx);
          const bool  alloc = injectedAllocFailure || fluhip_last_error_is_out_of_memory(mCtx) != 0;
          if (!alloc) return {S::kError, "BufNMF: ", why ? why : "the batched job failed"};
        }
        batchedFallbacks() += 1; // (read by tests: the batched block failed and the sequential loop took the job)
        progressFloor = prog.count / std::max(1.0, prog.total); // (the task's progress does not go backwards: see the loop below)
      }
      // the corpus could not be created, or a later allocation of the batched block failed (device memory): the
      // channel-by-channel loop below needs one channel at a time
    }

    mono.resize(static_cast<size_t>(nFrames));
    if (seedFilters) seedW.resize(static_cast<size_t>(rank * nBins));
    if (seedEnvelopes) seedH.resize(static_cast<size_t>(rank * nWindows));
    if (hasFilters && !fixFilters) outW.resize(static_cast<size_t>(rank * nBins));
    if (hasEnvelopes && !fixEnvelopes) outH.resize(static_cast<size_t>(rank * nWindows));
    if (shouldResynth && hasResynth) outR.resize(static_cast<size_t>(rank * nFrames));
    for (index i = 0; i < nChannels; ++i)
    {
      if (c.task() && !c.task()->iterationUpdate(static_cast<double>(i), static_cast<double>(nChannels)))
        return {S::kCancelled, ""};
      // :240  tmp <<= source.samps(offset, nFrames, startChan + i)   (float stays float: the
      //        float -> double conversion happens on the device)
      VectorView<float>(mono.data(), nFrames) <<= source.samps(P.startFrame, nFrames, P.startChan + i);
      // :246-258 seeds, gathered channel by channel
      for (index j = 0; j < rank; ++j)
      {
        if (seedFilters)
          VectorView<float>(seedW.data() + j * nBins, nBins) <<=
              VectorView<const float>(BufferAdaptor::Access(P.bases.get()).samps(i * rank + j));
        if (seedEnvelopes)
          VectorView<float>(seedH.data() + j * nWindows, nWindows) <<=
              VectorView<const float>(BufferAdaptor::Access(P.activations.get()).samps(i * rank + j));
      }
      // (channel i covers the fractions [i, i + 1) / nChannels of the job: what the batched block had already reported is the floor)
      Prog prog{&c, 0, progressTotal,
                std::min(progressTotal, std::max(0.0, (progressFloor * static_cast<double>(nChannels) - static_cast<double>(i)) * progressTotal))};
      auto cb = [](int64_t, void* u) -> int { // :261-267
        auto* p = static_cast<Prog*>(u);
        ++p->count;
        return p->c->task() ? (p->c->task()->processUpdate(std::max(static_cast<double>(p->count), p->floor), p->total) ? 1 : 0) : 1;
      };
      const int rc = fluhip_bufnmf_channel_f32(
          mCtx, mono.data(), nFrames, 1, fftParams.winSize(), fftParams.fftSize(), hop, rank,
          needsAnalysis ? P.iterations : 0, !fixFilters, !fixEnvelopes, P.seed, seedFilters ? seedW.data() : nullptr,
          seedEnvelopes ? seedH.data() : nullptr, outW.empty() ? nullptr : outW.data(),
          outH.empty() ? nullptr : outH.data(), outR.empty() ? nullptr : outR.data(), cb, &prog); // :268-271
      if (rc == FLUHIP_CANCELLED || (c.task() && c.task()->cancelled())) return {S::kCancelled, ""}; // :273-274
      if (rc != FLUHIP_OK) return {S::kError, "BufNMF: ", fluhip_last_error(mCtx)};

      if (hasFilters && !fixFilters) // :277-283
      {
        BufferAdaptor::Access filters(P.bases.get());
        for (index j = 0; j < rank; ++j)
          filters.samps(i * rank + j) <<= VectorView<const float>(outW.data() + j * nBins, nBins);
      }
      if (hasEnvelopes && !fixEnvelopes) // :286-300 (scaling by 1/max(H) already applied on the device)
      {
        BufferAdaptor::Access envelopes(P.activations.get());
        for (index j = 0; j < rank; ++j)
          envelopes.samps(i * rank + j) <<= VectorView<const float>(outH.data() + j * nWindows, nWindows);
      }
      if (shouldResynth && hasResynth) // :302-334
      {
        BufferAdaptor::Access resynth(P.resynth.get());
        scatterChannels(resynth, i * rank, rank, outR.data(), nFrames);
        for (index j = 0; j < rank; ++j)
        {
          for (int step = 0; step < 3; ++step)
            if (c.task() && !c.task()->processUpdate(std::max(static_cast<double>(++prog.count), prog.floor), progressTotal)) return {S::kCancelled, ""};
        }
      }
    }
    return {S::kOk, ""};
  }

private:
  static bool sequentialForced()
  {
#ifdef FLUHIP_AB_SWITCHES // test / measurement builds of a host: the channel-by-channel loop on request
    const char* e = std::getenv("FLUHIP_CLIENT_SEQUENTIAL");
    return e && std::atoi(e) != 0;
#else
    return false;
#endif
  }
public:
  // jobs (of this process) whose batched block failed and that the channel-by-channel loop then ran
  static std::atomic<int>& batchedFallbacks()
  {
    static std::atomic<int> n{0};
    return n;
  }

private:
  NMFParams*  mParams;
  fluhip_ctx* mCtx{nullptr};
  int         mDevice{-1};
  std::vector<std::pair<int, fluhip_ctx*>> mPool; // (device, context) per entry of FluidContext::devices()
};

} // namespace bufnmf
} // namespace fluhip
