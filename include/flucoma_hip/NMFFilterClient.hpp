// NMFFilterClient.hpp -- NMFFilter over a whole buffer, on the MI355X C ABI (include/flucoma_hip.h).
//
// client::nmffilter::NMFFilterClient, include/flucoma/clients/rt/NMFFilterClient.hpp, is a real-time client (AudioIn,
// AudioOut with one output per component; FlucomaClients.cmake:127 registers it as RTNMFFilterClient only): per frame
// NMF::processFrame against the dictionary in `bases` (kIterations, :102-105), the estimate W^T h as the ratio mask's
// denominator (:106), every component's rank-one estimate through the mask and the inverse transform (:107-113), the
// BufferedProcess overlap-adding the frames and normalising by the overlap-added squared window.  Offline the reference's
// wrapper for audio-rate clients, impl::NRTClientWrapper<Streaming, ...> (clients/common/FluidNRTClientWrapper.hpp:
// 466-547), feeds such a client 64-sample host vectors, input padded by the client's latency (one window), and drops that
// many output samples; this header is the client driven that way, with
//   parameters   the wrapper's source / startFrame / numFrames / startChan / numChans in front of the client's table
//                (:26-38): bases (input buffer) / maxComponents (20, Min 1) / iterations (10, Min 1) / seed (-1) /
//                fftSettings (1024, -1, -1)
//   output       ONE buffer `resynth` of numFrames x (channels * rank), component j of channel i in buffer channel
//                i * rank + j at the source's sample rate -- the layout BufNMF's resynthesis buffer has
//                (clients/nrt/NMFClient.hpp:196-198, 302-334).  (Streaming itself gives every audio output of the client
//                its own buffer, i.e. `maxComponents` buffer parameters: the reason no BufNMFFilter exists upstream.)
//   rank = min(channels of `bases`, maxComponents) (:91); without a valid bases buffer of fft/2 + 1 frames every call
//   returns at :83-93 and the outputs stay zero.
// The whole job -- every channel, every frame, every component -- is one call, fluhip_nmffilter_f32.
#pragma once

#include "BufferAdaptor.hpp"
#include "ParamDescriptors.hpp"
#include "DeviceContext.hpp"
#include "NRTControlAdaptor.hpp"
#include "NRTThreadingAdaptor.hpp"

namespace fluhip {
namespace nmffilter {

enum NMFFilterIndex { kFilterbuf, kMaxRank, kIterations, kRandomSeed, kFFT }; // rt/NMFFilterClient.hpp:26-32

struct NRTNMFFilterParams
{
  // cc/FluidNRTClientWrapper.hpp:33-39 (the wrapper's input-buffer parameters) and the output buffer
  std::shared_ptr<const BufferAdaptor> source;
  index                                startFrame{0}; // Min(0)
  index                                numFrames{-1};
  index                                startChan{0};  // Min(0)
  index                                numChans{-1};
  std::shared_ptr<BufferAdaptor>       resynth;
  // rt/NMFFilterClient.hpp:34-38
  std::shared_ptr<const BufferAdaptor> bases;
  index                                maxComponents{20}; // Min(1)
  index                                iterations{10};    // Min(1)
  index                                seed{-1};
  FFTParams                            fftSettings{1024, -1, -1};

  template <class In, class Out>
  void forEachBuffer(In&& in, Out&& out)
  {
    forEachBuffer(in, out, out);
  }
  template <class In, class Out, class OutOnly>
  void forEachBuffer(In&& in, Out&&, OutOnly&& outOnly)
  {
    in(source);
    in(bases);
    outOnly(resynth); // resized, every sample written
  }
  void constrain()
  {
    startFrame = std::max<index>(0, startFrame);
    startChan = std::max<index>(0, startChan);
    impl::constrainFFT(fftSettings);
    maxComponents = std::max<index>(1, maxComponents);
    iterations = std::max<index>(1, iterations);
  }
};
} // namespace nmffilter

class NRTNMFFilterClient
{
public:
  using ParamSetViewType = nmffilter::NRTNMFFilterParams;
  // the parameter table a host enumerates (the offline wrapper's parameters in front of rt/NMFFilterClient.hpp:34-38; ParamDescriptors.hpp)
  static constexpr ParamDescriptorList getParameterDescriptors() { return paramdesc::list(paramdesc::kBufNMFFilter); }

  NRTNMFFilterClient(ParamSetViewType& p, FluidContext&) : mParams(&p) {}
  void setParams(ParamSetViewType& p) { mParams = &p; }

  template <typename T>
  Result process(FluidContext& c)
  {
    using S = Result::Status;
    const ParamSetViewType& P = *mParams;
    const FFTParams         f = P.fftSettings;
    // NRTClientWrapper::process, cc/FluidNRTClientWrapper.hpp:298-353
    index  nFrames = P.numFrames, nChans = P.numChans;
    Result rangeCheck = bufferRangeCheck(P.source.get(), P.startFrame, nFrames, P.startChan, nChans);
    if (!rangeCheck.ok()) return rangeCheck;
    if (!P.resynth || !BufferAdaptor::Access(P.resynth.get()).exists()) return {S::kError, "No valid output has been set"};

    // rt/NMFFilterClient.hpp:83-113: the filter buffer as every process() call reads it
    std::vector<float> bases;
    index              rank = 0;
    if (P.bases)
    {
      BufferAdaptor::ReadAccess filterBuffer(P.bases.get());
      if (filterBuffer.exists() && filterBuffer.valid() && filterBuffer.numFrames() == f.frameSize())
      {
        rank = std::min<index>(filterBuffer.numChans(), P.maxComponents); // :91
        bases.resize(static_cast<size_t>(rank * f.frameSize()));
        for (index i = 0; i < rank; ++i) // :103-104
          VectorView<float>(bases.data() + i * f.frameSize(), f.frameSize()) <<= filterBuffer.samps(i);
      }
    }
    // audioChannelsOut() is maxComponents as constructed and never narrows (rt/NMFFilterClient.hpp:57-59; only NMFMatch narrows
    // its control outputs to the bases' rank): channels past the rank stay silent.  ADVICE r04.
    const index nOut = P.maxComponents;

    BufferAdaptor::ReadAccess source(P.source.get());
    const double              sampleRate = source.sampleRate();
    std::vector<float>        out(static_cast<size_t>(nChans * nOut * nFrames), 0.0f);
    if (rank > 0)
    {
      Result dev = mDevice.ensure(c.device());
      if (!dev.ok()) return dev;
      std::vector<float> audio(static_cast<size_t>(nChans * nFrames));
      for (index i = 0; i < nChans; ++i) // :499-509
        VectorView<float>(audio.data() + i * nFrames, nFrames) <<= source.samps(P.startFrame, nFrames, P.startChan + i);
      if (c.task() && !c.task()->iterationUpdate(0.0, 1.0)) return {S::kCancelled, ""};
      // the device writes channels x rank rows; the output buffer has maxComponents rows per channel
      std::vector<float> dense;
      float*             dst = out.data();
      if (rank != nOut) { dense.resize(static_cast<size_t>(nChans * rank * nFrames)); dst = dense.data(); }
      const int rc = fluhip_nmffilter_f32(mDevice.get(), audio.data(), nChans, nFrames, f.winSize(), f.fftSize(), f.hopSize(),
                                          bases.data(), rank, P.iterations, P.seed, dst);
      if (rc != FLUHIP_OK) return mDevice.result(rc);
      if (rank != nOut)
        for (index i = 0; i < nChans; ++i)
          for (index j = 0; j < rank; ++j)
            std::memcpy(out.data() + (i * nOut + j) * nFrames, dense.data() + (i * rank + j) * nFrames, sizeof(float) * static_cast<size_t>(nFrames));
    }
    // :526-530 reports progress per host vector; the batch is one step
    if (FluidTask* task = c.task())
      if (!task->processUpdate(1.0, 1.0)) return {S::kCancelled, ""};

    BufferAdaptor::Access thisOutput(P.resynth.get());
    Result                r = thisOutput.resize(nFrames, nChans * nOut, sampleRate); // :536-538 per output; here one buffer
    if (!r.ok()) return r;
    scatterChannels(thisOutput, 0, nChans * nOut, out.data(), nFrames);
    return {};
  }

private:
  ParamSetViewType* mParams;
  DeviceContext     mDevice;
};

using NRTThreadedNMFFilterClient = NRTThreadingAdaptor<NRTNMFFilterClient>;

} // namespace fluhip
