// NRTControlAdaptor.hpp -- the offline wrapper of the frame-rate analysis clients (BufMFCC, BufMelBands), over the C ABI.
//
// Mirrors impl::NRTClientWrapper<StreamingControl, ...>, include/flucoma/clients/common/FluidNRTClientWrapper.hpp:
//   wrapper parameters  :33-39, :747-763   source / startFrame / numFrames / startChan / numChans, the output buffer,
//                                          "padding" {None, Default, Full}; the wrapped client's parameters follow
//   process<T>()        :298-353           range check of the input, "No valid output has been set", then
//   StreamingControl    :551-660           padded copy of every channel, one client call per hop, the first
//                                          latency / hop frames dropped, output resized to frames x (channels * features)
//                                          at sampleRate / hop, feature i of channel j in buffer channel i + j * features
// The reference pushes hop samples at a time through the real-time client; here the channels of the job are one batch of
// equal-length buffers for the fused STFT -> mel -> [DCT] kernel (kernels_stft2.hip: stft_feat_kernel), whose frame
// positions are the closed form of that bookkeeping (tests/test_oracle.py derives it from the FluidSource model).
#pragma once

#include "BufferAdaptor.hpp"
#include "DeviceContext.hpp"

#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

namespace fluhip {

// the wrapper's own parameters (cc/FluidNRTClientWrapper.hpp:33-39, :761); a client's parameter struct derives from this
struct NRTControlParams
{
  std::shared_ptr<const BufferAdaptor> source;        // "source"
  index                                startFrame{0}; // Min(0)
  index                                numFrames{-1};
  index                                startChan{0};  // Min(0)
  index                                numChans{-1};
  std::shared_ptr<BufferAdaptor>       features;      // "features"
  index                                padding{1};    // None, Default, Full

  template <class In, class Out>
  void forEachBuffer(In&& in, Out&& out)
  {
    forEachBuffer(in, out, out);
  }
  template <class In, class Out, class OutOnly>
  void forEachBuffer(In&& in, Out&&, OutOnly&& outOnly)
  {
    in(source);
    outOnly(features); // resized and filled, never read
  }
  void constrainWrapper()
  {
    startFrame = std::max<index>(0, startFrame);
    startChan = std::max<index>(0, startChan);
    padding = std::min<index>(2, std::max<index>(0, padding));
  }
};

namespace impl {

// Launch: int(fluhip_ctx*, const float* audio /*channels x n*/, int64_t channels, int64_t n, int padding,
//             float* out /*channels x nFeatures x frames*/, int64_t* frames)
template <class Launch>
Result streamingControl(const NRTControlParams& P, const FFTParams& fft, index nFeatures, DeviceContext& device,
                        FluidContext& c, Launch&& launch)
{
  using S = Result::Status;
  // NRTClientWrapper::process, :298-353
  index  nFrames = P.numFrames, nChans = P.numChans;
  Result rangeCheck = bufferRangeCheck(P.source.get(), P.startFrame, nFrames, P.startChan, nChans);
  if (!rangeCheck.ok()) return rangeCheck;
  if (!P.features || !BufferAdaptor::Access(P.features.get()).exists()) return {S::kError, "No valid output has been set"};

  // StreamingControl::process, :557-579
  const index win = fft.winSize(), hop = fft.hopSize();
  const index userPadding = P.padding == 0 ? 0 : P.padding == 1 ? win >> 1 : win - hop; // FFTParams::padding
  const index latency = win;                                                            // rt/MFCCClient.hpp:138
  index       paddedLength = nFrames + latency + 2 * userPadding;
  if (P.padding == 2) paddedLength = static_cast<index>(std::ceil(double(paddedLength) / hop) * hop);
  const index nAnalysisFrames = 1 + (paddedLength - win) / hop;
  const index keepHops = nAnalysisFrames - latency / hop; // :643-644

  Result dev = device.ensure(c.device());
  if (!dev.ok()) return dev;

  BufferAdaptor::ReadAccess source(P.source.get());
  const double              sampleRate = source.sampleRate();
  std::vector<float>        audio((size_t) (nChans * nFrames));
  for (index i = 0; i < nChans; ++i) // :586-596
    VectorView<float>(audio.data() + i * nFrames, nFrames) <<= source.samps(P.startFrame, nFrames, P.startChan + i);

  std::vector<float> out((size_t) std::max<index>(1, nChans * nFeatures * keepHops));
  int64_t            frames = 0;
  const int          rc = launch(device.get(), audio.data(), (int64_t) nChans, (int64_t) nFrames, (int) P.padding, out.data(), &frames);
  if (rc != FLUHIP_OK) return device.result(rc);
  if (frames != keepHops) return {S::kError, "frame count of the device path differs from the client's: ", frames, " vs ", keepHops};
  // :624-628 reports progress per analysis frame; the batch is one step
  if (FluidTask* task = c.task())
    task->processUpdate(static_cast<double>(nAnalysisFrames * nChans), static_cast<double>(nAnalysisFrames * nChans));

  // :636-656
  BufferAdaptor::Access thisOutput(P.features.get());
  Result                resizeResult = thisOutput.resize(keepHops, nChans * nFeatures, sampleRate / hop);
  if (!resizeResult.ok()) return resizeResult;
  for (index i = 0; i < nFeatures; ++i)
    for (index j = 0; j < nChans; ++j)
      thisOutput.samps(i + j * nFeatures) <<= VectorView<const float>(out.data() + (j * nFeatures + i) * keepHops, keepHops);
  return {};
}

inline void constrainFFT(FFTParams& f)
{
  f.win = std::max<index>(4, f.win); // cc/ParameterTypes.hpp:371-393
  if (f.fft >= 0)
  {
    index p = 1;
    while (p < std::max(f.fft, f.win)) p <<= 1;
    f.fft = p;
  }
}
} // namespace impl
} // namespace fluhip
