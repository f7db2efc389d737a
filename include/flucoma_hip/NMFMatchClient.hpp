// NMFMatchClient.hpp -- NMFMatch over a whole buffer, on the MI355X C ABI (include/flucoma_hip.h).
//
// client::nmfmatch::NMFMatchClient, include/flucoma/clients/rt/NMFMatchClient.hpp, is a real-time client (AudioIn,
// ControlOut; FlucomaClients.cmake:128 registers it as RTNMFMatchClient only): per host vector it writes the current
// activations to its control output (:104-105) and then solves NMF::processFrame -- ten iterations, :113-116 -- for every
// frame the vector completes.  This header is that client behind the reference's own offline wrapper for control-rate
// clients, impl::NRTClientWrapper<StreamingControl, ...> (clients/common/FluidNRTClientWrapper.hpp:551-660; the wrapper the
// analysis clients BufMFCC / BufMelBands are made of): the wrapper's parameters in front of the client's
//   parameter table  :24-38   bases (input buffer) / maxComponents (20, Min 1) / iterations (10, Min 1) / seed (-1) /
//                             fftSettings (1024, -1, -1)
//   process          :76-118  rank = min(channels of `bases`, maxComponents); nothing happens (the output stays zero, the
//                             feature count stays maxComponents) without a valid bases buffer of fft/2 + 1 frames
// The whole job -- every channel, every frame -- is one call, fluhip_nmfmatch_f32.
#pragma once

#include "NRTControlAdaptor.hpp"
#include "NRTThreadingAdaptor.hpp"
#include "ParamDescriptors.hpp"

namespace fluhip {
namespace nmfmatch {

enum NMFMatchParamIndex { kFilterbuf, kMaxRank, kIterations, kRandomSeed, kFFT }; // rt/NMFMatchClient.hpp:24-30

struct NRTNMFMatchParams : NRTControlParams
{
  std::shared_ptr<const BufferAdaptor> bases;                // "bases"
  index                                maxComponents{20};    // Min(1)
  index                                iterations{10};       // Min(1); read by nothing (:113-116 passes the literal 10)
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
    outOnly(features);
  }
  void constrain()
  {
    constrainWrapper();
    impl::constrainFFT(fftSettings);
    maxComponents = std::max<index>(1, maxComponents);
    iterations = std::max<index>(1, iterations);
  }
};
} // namespace nmfmatch

class NRTNMFMatchClient
{
public:
  using ParamSetViewType = nmfmatch::NRTNMFMatchParams;
  // the parameter table a host enumerates (the offline wrapper's parameters in front of rt/NMFMatchClient.hpp:32-38; ParamDescriptors.hpp)
  static constexpr ParamDescriptorList getParameterDescriptors() { return paramdesc::list(paramdesc::kBufNMFMatch); }

  NRTNMFMatchClient(ParamSetViewType& p, FluidContext&) : mParams(&p) {}
  void setParams(ParamSetViewType& p) { mParams = &p; }

  template <typename T>
  Result process(FluidContext& c)
  {
    const ParamSetViewType& P = *mParams;
    const FFTParams         f = P.fftSettings;
    // rt/NMFMatchClient.hpp:84-98: the filter buffer, as every process() call reads it
    std::vector<float> bases;
    index              rank = 0;
    if (P.bases)
    {
      BufferAdaptor::ReadAccess filterBuffer(P.bases.get());
      if (filterBuffer.exists() && filterBuffer.valid() && filterBuffer.numFrames() == f.frameSize())
      {
        rank = std::min<index>(filterBuffer.numChans(), P.maxComponents); // :93
        bases.resize(static_cast<size_t>(rank * f.frameSize()));
        for (index i = 0; i < rank; ++i) // :100-101
          VectorView<float>(bases.data() + i * f.frameSize(), f.frameSize()) <<= filterBuffer.samps(i);
      }
    }
    // controlChannelsOut(): {1, maxComponents} as constructed (:65), {1, rank} once a call has seen the filters (:95-98)
    const index nFeatures = rank > 0 ? rank : P.maxComponents;
    return impl::streamingControl(P, f, nFeatures, mDevice, c,
                                  [&](fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int padding, float* out,
                                      int64_t* frames) {
                                    if (rank > 0)
                                      return fluhip_nmfmatch_f32(ctx, audio, count, n, f.winSize(), f.fftSize(), f.hopSize(),
                                                                 bases.data(), rank, P.seed, padding, out, frames);
                                    // no usable filters: every call returns at :82-91, the output columns stay zero
                                    int rc = fluhip_nmfmatch_f32(ctx, audio, count, n, f.winSize(), f.fftSize(), f.hopSize(),
                                                                 nullptr, 1, P.seed, padding, nullptr, frames);
                                    if (rc == FLUHIP_OK) std::fill(out, out + count * nFeatures * (*frames), 0.0f);
                                    return rc;
                                  });
  }

private:
  ParamSetViewType* mParams;
  DeviceContext     mDevice;
};

using NRTThreadedNMFMatchClient = NRTThreadingAdaptor<NRTNMFMatchClient>;

} // namespace fluhip
