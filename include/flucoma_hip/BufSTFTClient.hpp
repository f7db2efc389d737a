// BufSTFTClient.hpp -- BufSTFT client over the MI355X C ABI (include/flucoma_hip.h).
//
// Mirrors client::bufstft::BufferSTFTClient, include/flucoma/clients/nrt/BufSTFTClient.hpp:23-287:
//   parameter table   :23-47    -> BufSTFTParams (plain struct, same names / defaults / constraints)
//   processFwd        :81-184   -> same order of checks, Result codes and messages, buffer shapes and sample rates;
//                                  padding + framing + FFT + magnitude / phase are one call, fluhip_bufstft_forward_f32
//   processInverse    :186-276  -> likewise; std::polar + inverse FFT + overlap-add / window^2 normaliser are
//                                  fluhip_bufstft_inverse_f32
// There is no CPU path: if the library cannot create a context on the requested device the job returns kError.
#pragma once

#include "BufferAdaptor.hpp"
#include "ParamDescriptors.hpp"
#include "DeviceContext.hpp"
#include "NRTThreadingAdaptor.hpp"

#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

namespace fluhip {
namespace bufstft {

// nrt/BufSTFTClient.hpp:23-34
enum BufferSTFTParamIndex { kSource, kOffset, kNumFrames, kStartChan, kMag, kPhase, kResynth, kInvert, kPadding, kFFT };

// nrt/BufSTFTClient.hpp:36-47
struct BufSTFTParams
{
  std::shared_ptr<const BufferAdaptor> source;          // "source"
  index                                startFrame{0};   // Min(0)
  index                                numFrames{-1};
  index                                startChan{0};    // Min(0)
  std::shared_ptr<BufferAdaptor>       magnitude;       // "magnitude"
  std::shared_ptr<BufferAdaptor>       phase;           // "phase"
  std::shared_ptr<BufferAdaptor>       resynth;         // "resynth"
  index                                inverse{0};      // 0..1
  index                                padding{1};      // None, Default, Full
  FFTParams                            fftSettings{1024, -1, -1};

  template <class In, class Out>
  void forEachBuffer(In&& in, Out&& out)
  {
    forEachBuffer(in, out, out);
  }
  // (third visitor: buffers the client only writes -- magnitude / phase going forward, the resynthesis going back)
  template <class In, class Out, class OutOnly>
  void forEachBuffer(In&& in, Out&& out, OutOnly&& outOnly)
  {
    in(source);
    if (inverse == 0) { outOnly(magnitude); outOnly(phase); out(resynth); }
    else { out(magnitude); out(phase); outOnly(resynth); }
  }

  void constrain()
  {
    startFrame = std::max<index>(0, startFrame);
    startChan = std::max<index>(0, startChan);
    inverse = std::min<index>(1, std::max<index>(0, inverse));
    padding = std::min<index>(2, std::max<index>(0, padding));
    fftSettings.win = std::max<index>(4, fftSettings.win); // cc/ParameterTypes.hpp:371-393
    if (fftSettings.fft >= 0)
    {
      index p = 1;
      while (p < std::max(fftSettings.fft, fftSettings.win)) p <<= 1;
      fftSettings.fft = p;
    }
  }
};

// FFTParams::padding, cc/ParameterTypes.hpp:315-323
inline index fftPadding(const FFTParams& f, index option)
{
  return option == 0 ? 0 : option == 1 ? f.winSize() >> 1 : f.winSize() - f.hopSize();
}

class BufferSTFTClient
{
public:
  using ParamSetViewType = BufSTFTParams;
  // the parameter table a host enumerates (nrt/BufSTFTClient.hpp; ParamDescriptors.hpp)
  static constexpr ParamDescriptorList getParameterDescriptors() { return paramdesc::list(paramdesc::kBufSTFT); }

  BufferSTFTClient(BufSTFTParams& p, FluidContext&) : mParams(&p) {}
  void setParams(BufSTFTParams& p) { mParams = &p; }

  template <typename T>
  Result process(FluidContext& c)
  {
    return mParams->inverse == 0 ? processFwd(c) : processInverse(c); // :74-79
  }

private:
  using S = Result::Status;

  Result processFwd(FluidContext& c)
  {
    const BufSTFTParams& P = *mParams;
    if (!P.source) return {S::kError, "No input buffer supplied"};
    const bool haveMag = P.magnitude != nullptr, havePhase = P.phase != nullptr;
    if (!haveMag && !havePhase) return {S::kError, "Neither magnitude nor phase buffer supplied"};

    BufferAdaptor::Access mags(P.magnitude.get());
    BufferAdaptor::Access phases(P.phase.get());

    index  offset = P.startFrame;
    index  numFrames = P.numFrames;
    index  numChans = 1;
    Result rangeOK = bufferRangeCheck(P.source.get(), offset, numFrames, P.startChan, numChans);
    if (!rangeOK.ok()) return rangeOK;

    BufferAdaptor::ReadAccess source(P.source.get());
    if (haveMag && !mags.exists()) return {S::kError, "Magnitude buffer not found"};
    if (havePhase && !phases.exists()) return {S::kError, "Phase buffer not found"};

    const index fftSize = P.fftSettings.fftSize(), winSize = P.fftSettings.winSize(), hopSize = P.fftSettings.hopSize();
    const index padding = fftPadding(P.fftSettings, P.padding);
    index       paddedLength = numFrames + (padding << 1); // :121-128
    if (P.padding == 2) paddedLength = static_cast<index>(std::ceil(double(paddedLength) / hopSize) * hopSize);
    const index numHops = 1 + (paddedLength - winSize) / hopSize;
    const index numBins = (fftSize >> 1) + 1;

    if (numChans * numBins >= 65536) // :135-138
      return {S::kError, "Can produce up to 65536 channels. Split your data up and try again"};

    if (haveMag)
    {
      Result r = mags.resize(numHops, numBins * numChans, source.sampleRate() / hopSize);
      if (!r.ok()) return r;
    }
    if (havePhase)
    {
      Result r = phases.resize(numHops, numBins * numChans, source.sampleRate() / hopSize);
      if (!r.ok()) return r;
    }

    Result dev = mDevice.ensure(c.device());
    if (!dev.ok()) return dev;

    // :152 reads source.samps(0): channel 0 whatever startChan says (startChan only takes part in the range check);
    // kept, so that the two implementations return the same buffers
    auto               input = source.samps(offset, numFrames, 0);
    std::vector<float> magOut(haveMag ? (size_t) (numBins * numHops) : 0), phaseOut(havePhase ? (size_t) (numBins * numHops) : 0);
    int64_t            hops = 0;
    const int rc = fluhip_bufstft_forward_f32(mDevice.get(), input.data(), numFrames, input.stride, winSize, fftSize, hopSize,
                                              (int) P.padding, haveMag ? magOut.data() : nullptr,
                                              havePhase ? phaseOut.data() : nullptr, &hops);
    if (rc != FLUHIP_OK) return mDevice.result(rc);
    if (hops != numHops) return {S::kError, "frame count of the device path differs from the client's: ", hops, " vs ", numHops};

    // :168-178  buffer channel = bin, buffer frame = hop (mags.allFrames().transpose() <<= tmpMags)
    // (block by block of frames: 1025 strided passes over an interleaved buffer otherwise)
    if (haveMag) scatterChannels(mags, 0, numBins, magOut.data(), numHops);
    if (havePhase) scatterChannels(phases, 0, numBins, phaseOut.data(), numHops);
    return {};
  }

  Result processInverse(FluidContext& c)
  {
    const BufSTFTParams& P = *mParams;
    const bool           haveMag = P.magnitude != nullptr, havePhase = P.phase != nullptr;
    if (!haveMag || !havePhase) return {S::kError, "Need both magnutude and phase buffers for inverse transform"};
    if (!P.resynth) return {S::kError, "No resynthesis buffer supplied"};

    BufferAdaptor::ReadAccess mags(P.magnitude.get());
    BufferAdaptor::ReadAccess phases(P.phase.get());
    if (mags.numFrames() != phases.numFrames() || mags.numChans() != phases.numChans())
      return {S::kError, "Magnitude and Phase buffer sizes don't match"};

    const index fftSize = P.fftSettings.fftSize(), winSize = P.fftSettings.winSize(), hopSize = P.fftSettings.hopSize();
    if (mags.numChans() != (fftSize >> 1) + 1)
      return {S::kError, "Wrong number of channels for FFT sizee of ", fftSize, " got ", mags.numChans(), " expected ",
              (fftSize >> 1) + 1};

    BufferAdaptor::Access resynth(P.resynth.get());
    const index           numFrames = mags.numFrames();
    const index           padding = fftPadding(P.fftSettings, P.padding);
    const index           paddedOutputSize = (numFrames - 1) * hopSize + winSize; // :228-231
    const index           finalOutputSize = paddedOutputSize - padding;
    Result resizeResult = resynth.resize(finalOutputSize, 1, mags.sampleRate() * hopSize);
    if (!resizeResult.ok()) return resizeResult;

    Result dev = mDevice.ensure(c.device());
    if (!dev.ok()) return dev;

    const index        numBins = mags.numChans();
    std::vector<float> magIn((size_t) (numBins * numFrames)), phaseIn((size_t) (numBins * numFrames));
    gatherChannels(mags, 0, numBins, magIn.data(), numFrames);
    gatherChannels(phases, 0, numBins, phaseIn.data(), numFrames);
    int64_t nOut = 0;
    int     rc = fluhip_bufstft_inverse_f32(mDevice.get(), magIn.data(), phaseIn.data(), numFrames, winSize, fftSize, hopSize,
                                            (int) P.padding, nullptr, &nOut);
    if (rc != FLUHIP_OK) return mDevice.result(rc);
    if (nOut != finalOutputSize)
      return {S::kError, "output length of the device path differs from the client's: ", nOut, " vs ", finalOutputSize};
    std::vector<float> out((size_t) std::max<index>(nOut, 1));
    rc = fluhip_bufstft_inverse_f32(mDevice.get(), magIn.data(), phaseIn.data(), numFrames, winSize, fftSize, hopSize,
                                    (int) P.padding, out.data(), &nOut);
    if (rc != FLUHIP_OK) return mDevice.result(rc);
    resynth.samps(0) <<= VectorView<const float>(out.data(), finalOutputSize); // :272
    return {};
  }

  BufSTFTParams* mParams;
  DeviceContext  mDevice;
};
} // namespace bufstft

using NRTThreadedBufferSTFTClient = NRTThreadingAdaptor<bufstft::BufferSTFTClient>; // nrt/BufSTFTClient.hpp:279-280

} // namespace fluhip
