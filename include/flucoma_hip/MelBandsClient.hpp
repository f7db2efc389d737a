// MelBandsClient.hpp -- BufMelBands client over the MI355X C ABI (include/flucoma_hip.h).
//
// Mirrors the offline form of client::melbands::MelBandsClient, include/flucoma/clients/rt/MelBandsClient.hpp:
//   parameter table  :28-43     numBands / minFreq / maxFreq / normalize / scale / fftSettings, behind the wrapper's
//                               source / startFrame / numFrames / startChan / numChans / features / padding
//   process          :77-119    STFT magnitude -> MelBands::processFrame(bands, normalize == 1, false, scale == 1);
//                               reset() :123-129 (filter bank of the buffer's sample rate)
//   NRTMelBandsClient / NRTThreadedMelBandsClient  (end of the file)
// The whole job -- every channel, every frame -- is one call, fluhip_bufmelbands_padded_f32.
#pragma once

#include "NRTControlAdaptor.hpp"
#include "NRTThreadingAdaptor.hpp"
#include "ParamDescriptors.hpp"

namespace fluhip {
namespace melbands {

enum MelBandsParamIndex { kNBands, kMinFreq, kMaxFreq, kNormalize, kScale, kFFT }; // rt/MelBandsClient.hpp:28-35

struct NRTMelBandsParams : NRTControlParams
{
  index     numBands{40};   // Min(2)
  double    minFreq{20};    // Min(0)
  double    maxFreq{20000}; // Min(0)
  index     normalize{1};   // No, Yes
  index     scale{0};       // Linear, dB
  FFTParams fftSettings{1024, -1, -1};

  void constrain()
  {
    constrainWrapper();
    impl::constrainFFT(fftSettings);
    numBands = std::max<index>(2, numBands);
    minFreq = std::max(0.0, minFreq);
    maxFreq = std::max(0.0, maxFreq);
    normalize = std::min<index>(1, std::max<index>(0, normalize));
    scale = std::min<index>(1, std::max<index>(0, scale));
  }
};
} // namespace melbands

class NRTMelBandsClient
{
public:
  using ParamSetViewType = melbands::NRTMelBandsParams;
  // the parameter table a host enumerates (rt/MelBandsClient.hpp:151-155; ParamDescriptors.hpp)
  static constexpr ParamDescriptorList getParameterDescriptors() { return paramdesc::list(paramdesc::kBufMelBands); }

  NRTMelBandsClient(ParamSetViewType& p, FluidContext&) : mParams(&p) {}
  void setParams(ParamSetViewType& p) { mParams = &p; }

  template <typename T>
  Result process(FluidContext& c)
  {
    const ParamSetViewType& P = *mParams;
    const FFTParams         f = P.fftSettings;
    const double            sampleRate = P.source ? BufferAdaptor::ReadAccess(P.source.get()).sampleRate() : 0.0;
    return impl::streamingControl(P, f, P.numBands, mDevice, c,
                                  [&](fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int padding, float* out,
                                      int64_t* frames) {
                                    return fluhip_bufmelbands_padded_f32(ctx, audio, count, n, f.winSize(), f.fftSize(),
                                                                         f.hopSize(), P.numBands, P.minFreq, P.maxFreq, sampleRate,
                                                                         (int) P.normalize, (int) P.scale, padding, out, frames);
                                  });
  }

private:
  ParamSetViewType* mParams;
  DeviceContext     mDevice;
};

using NRTThreadedMelBandsClient = NRTThreadingAdaptor<NRTMelBandsClient>;

} // namespace fluhip
