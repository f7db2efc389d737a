// MFCCClient.hpp -- BufMFCC client over the MI355X C ABI (include/flucoma_hip.h).
//
// Mirrors the offline form of client::mfcc::MFCCClient, include/flucoma/clients/rt/MFCCClient.hpp:
//   parameter table  :28-49     numCoeffs / numBands / startCoeff / minFreq / maxFreq / fftSettings, behind the
//                               wrapper's source / startFrame / numFrames / startChan / numChans / features / padding
//                               (:167-169 makeNRTParams)
//   process          :86-131    STFT magnitude -> MelBands::processFrame(bands, false, false, true) -> DCT rows
//                               startCoeff .. startCoeff + numCoeffs - 1; reset() :140-152 (filter bank of the buffer's
//                               sample rate, DCT of min(numCoeffs + startCoeff, numBands) rows)
//   NRTMFCCClient / NRTThreadedMFCCClient  :171-175
// The whole job -- every channel, every frame -- is one call, fluhip_bufmfcc_padded_f32.
#pragma once

#include "NRTControlAdaptor.hpp"
#include "NRTThreadingAdaptor.hpp"
#include "ParamDescriptors.hpp"

namespace fluhip {
namespace mfcc {

enum MFCCParamIndex { kNCoefs, kNBands, kDrop0, kMinFreq, kMaxFreq, kFFT }; // rt/MFCCClient.hpp:28-35

struct NRTMFCCParams : NRTControlParams
{
  index     numCoeffs{13};  // Min(2), UpperLimit<numBands>
  index     numBands{40};   // Min(2), FrameSizeUpperLimit<fftSettings>, LowerLimit<numCoeffs>
  index     startCoeff{0};  // 0..1
  double    minFreq{20};    // Min(0)
  double    maxFreq{20000}; // Min(0)
  FFTParams fftSettings{1024, -1, -1};

  void constrain()
  {
    constrainWrapper();
    impl::constrainFFT(fftSettings);
    numBands = std::min(std::max<index>(2, numBands), fftSettings.frameSize());
    numCoeffs = std::min(std::max<index>(2, numCoeffs), numBands);
    startCoeff = std::min<index>(1, std::max<index>(0, startCoeff));
    minFreq = std::max(0.0, minFreq);
    maxFreq = std::max(0.0, maxFreq);
  }
};
} // namespace mfcc

class NRTMFCCClient
{
public:
  using ParamSetViewType = mfcc::NRTMFCCParams;
  // the parameter table a host enumerates (rt/MFCCClient.hpp:171-175; ParamDescriptors.hpp)
  static constexpr ParamDescriptorList getParameterDescriptors() { return paramdesc::list(paramdesc::kBufMFCC); }

  NRTMFCCClient(ParamSetViewType& p, FluidContext&) : mParams(&p) {}
  void setParams(ParamSetViewType& p) { mParams = &p; }

  template <typename T>
  Result process(FluidContext& c)
  {
    const ParamSetViewType& P = *mParams;
    const FFTParams         f = P.fftSettings;
    const double            sampleRate = P.source ? BufferAdaptor::ReadAccess(P.source.get()).sampleRate() : 0.0;
    return impl::streamingControl(P, f, P.numCoeffs, mDevice, c,
                                  [&](fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int padding, float* out,
                                      int64_t* frames) {
                                    return fluhip_bufmfcc_padded_f32(ctx, audio, count, n, f.winSize(), f.fftSize(), f.hopSize(),
                                                                     P.numBands, P.numCoeffs, P.startCoeff, P.minFreq, P.maxFreq,
                                                                     sampleRate, padding, out, frames);
                                  });
  }

private:
  ParamSetViewType* mParams;
  DeviceContext     mDevice;
};

using NRTThreadedMFCCClient = NRTThreadingAdaptor<NRTMFCCClient>; // rt/MFCCClient.hpp:175

} // namespace fluhip
