// NMFSeedClient.hpp -- BufNMFSeed client over the MI355X C ABI (include/flucoma_hip.h).
//
// Mirrors client::nndsvd::NMFSeedClient, include/flucoma/clients/nrt/NMFSeedClient.hpp:26-140:
//   parameter table   :26-52    -> NMFSeedParams (plain struct, same names / defaults / constraints)
//   process<T>()      :73-131   -> same checks, messages, output shapes and sample rates; STFT -> magnitude -> NNDSVD ->
//                                  float bases / activations scaled by 1 / max H is one call, fluhip_bufnmfseed_f32
// There is no CPU path: if the library cannot create a context on the requested device the job returns kError.
#pragma once

#include "BufferAdaptor.hpp"
#include "ParamDescriptors.hpp"
#include "DeviceContext.hpp"
#include "NRTThreadingAdaptor.hpp"

#include <algorithm>
#include <memory>
#include <vector>

namespace fluhip {
namespace nndsvd {

// nrt/NMFSeedClient.hpp:26-36
enum NMFSeedParamIndex { kSource, kFilters, kEnvelopes, kMinRank, kMaxRank, kCoverage, kMethod, kRandomSeed, kFFT };

// nrt/NMFSeedClient.hpp:38-52
struct NMFSeedParams
{
  std::shared_ptr<const BufferAdaptor> source;             // "source"
  std::shared_ptr<BufferAdaptor>       bases;              // "bases"
  std::shared_ptr<BufferAdaptor>       activations;        // "activations"
  index                                minComponents{1};   // Min(1), UpperLimit<maxComponents>
  index                                maxComponents{200}; // Min(1), LowerLimit<minComponents>
  double                               coverage{0.5};      // 0..1
  index                                method{0};          // NMF-SVD, NNDSVDar, NNDSVDa, NNDSVD
  index                                seed{-1};
  FFTParams                            fftSettings{1024, -1, -1};

  template <class In, class Out>
  void forEachBuffer(In&& in, Out&& out)
  {
    in(source);
    out(bases);
    out(activations);
  }

  void constrain()
  {
    minComponents = std::max<index>(1, minComponents);
    maxComponents = std::max<index>(1, maxComponents);
    minComponents = std::min(minComponents, maxComponents);
    coverage = std::min(1.0, std::max(0.0, coverage));
    method = std::min<index>(3, std::max<index>(0, method));
    fftSettings.win = std::max<index>(4, fftSettings.win);
    if (fftSettings.fft >= 0)
    {
      index p = 1;
      while (p < std::max(fftSettings.fft, fftSettings.win)) p <<= 1;
      fftSettings.fft = p;
    }
  }
};

class NMFSeedClient
{
public:
  using ParamSetViewType = NMFSeedParams;
  // the parameter table a host enumerates (nrt/NMFSeedClient.hpp; ParamDescriptors.hpp)
  static constexpr ParamDescriptorList getParameterDescriptors() { return paramdesc::list(paramdesc::kBufNMFSeed); }

  NMFSeedClient(NMFSeedParams& p, FluidContext&) : mParams(&p) {}
  void setParams(NMFSeedParams& p) { mParams = &p; }

  template <typename T>
  Result process(FluidContext& c)
  {
    using S = Result::Status;
    const NMFSeedParams&      P = *mParams;
    BufferAdaptor::ReadAccess source(P.source.get());
    if (!source.exists()) return {S::kError, "Source Buffer Supplied But Invalid"};

    const double    sampleRate = source.sampleRate();
    const index     nFrames = source.numFrames();
    const FFTParams fftParams = P.fftSettings;
    const index     hop = fftParams.hopSize();
    const index     nWindows = (nFrames + hop) / hop; // :83-84
    const index     nBins = fftParams.frameSize();
    if (source.numChans() > 1) return {S::kError, "Only one channel supported"};

    Result dev = mDevice.ensure(c.device());
    if (!dev.ok()) return dev;

    const index        maxRank = P.maxComponents;
    auto               input = source.samps(0, nFrames, 0);
    std::vector<float> basesOut((size_t) (maxRank * nBins)), actsOut((size_t) (maxRank * nWindows));
    int64_t            rank = 0;
    const int rc = fluhip_bufnmfseed_f32(mDevice.get(), input.data(), nFrames, input.stride, fftParams.winSize(),
                                         fftParams.fftSize(), hop, P.minComponents, maxRank, P.coverage, (int) P.method,
                                         P.seed, basesOut.data(), actsOut.data(), &rank);
    if (rc != FLUHIP_OK) return mDevice.result(rc);

    // :108-128  both buffers are resized to the rank found, not to maxComponents
    BufferAdaptor::Access filters(P.bases.get());
    Result                resizeResult = filters.resize(nBins, rank, sampleRate / fftParams.fftSize());
    if (!resizeResult.ok()) return resizeResult;
    for (index j = 0; j < rank; ++j) filters.samps(j) <<= VectorView<const float>(basesOut.data() + j * nBins, nBins);

    BufferAdaptor::Access envelopes(P.activations.get());
    resizeResult = envelopes.resize((nFrames / hop) + 1, rank, sampleRate / hop);
    if (!resizeResult.ok()) return resizeResult;
    for (index j = 0; j < rank; ++j)
      envelopes.samps(j) <<= VectorView<const float>(actsOut.data() + j * nWindows, nWindows);
    return {};
  }

private:
  NMFSeedParams* mParams;
  DeviceContext  mDevice;
};
} // namespace nndsvd

using NRTThreadedNMFSeedClient = NRTThreadingAdaptor<nndsvd::NMFSeedClient>; // nrt/NMFSeedClient.hpp:135-136

} // namespace fluhip
