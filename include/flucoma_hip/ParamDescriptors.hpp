// ParamDescriptors.hpp -- the parameter tables of the mirrored clients as data a host can enumerate.
//
// Upstream a host wrapper (Max / Pd / SuperCollider / CLI) never names a client's parameters itself: it walks
// `Client::getParameterDescriptors()` -- the tuple `defineParameters(...)` builds (clients/common/ParameterSet.hpp,
// clients/common/ParameterTypes.hpp) -- and creates one attribute per entry from its name, display name, type, default
// and constraints; `NRTThreadingAdaptor` forwards the call to the wrapped client
// (clients/common/FluidNRTClientWrapper.hpp:792-809).  The mirrors keep their parameters in plain structs
// (NMFParams, BufSTFTParams, ...), so the same information is restated here as one array of ParamDescriptor per client,
// in the order of the reference's table -- which is also the order of the clients' parameter index enums -- and every
// client class and NRTThreadingAdaptor expose it as `getParameterDescriptors()`.  Names, display names, defaults, bounds
// and enum strings are the host-visible contract and are held against the reference's own tables by
// tests/test_client.py (fixture tests/golden/param_descriptors.json, minted from the reference headers by
// tools/make_param_descriptor_fixture.py).
//
//   nrt/NMFClient.hpp:54-71           nrt/NMFSeedClient.hpp:38-52       nrt/BufSTFTClient.hpp:36-47
//   rt/MFCCClient.hpp:37-50, :171-173 rt/MelBandsClient.hpp:37-44, :151-153
//   rt/NMFFilterClient.hpp:34-38      rt/NMFMatchClient.hpp:32-38       (the two real-time clients behind the offline
//                                                                        wrapper's parameters, as NMFFilterClient.hpp /
//                                                                        NMFMatchClient.hpp here describe)
//   wrapper parameters: clients/common/FluidNRTClientWrapper.hpp:33-39 (source offsets), :747-763 ("padding")
#pragma once

#include <cstddef>

namespace fluhip {

enum class ParamKind { kInputBuffer, kBuffer, kLong, kFloat, kEnum, kFFT };

struct ParamDescriptor
{
  const char*        name;
  const char*        displayName;
  ParamKind          kind;
  double             defaultValue;  // Long / Float / Enum (the index); FFT: the window size
  bool               hasMin;
  double             min;
  bool               hasMax;
  double             max;
  const char* const* enumStrings;   // Enum: the choices, else nullptr
  int                numEnumStrings;
  long               fftHop, fftSize; // FFT: the other two defaults (-1 = derived: hop = win / 2, fft = nextPow2(win))
  const char*        relational;    // constraints against other parameters, as the reference spells them, or nullptr
};

struct ParamDescriptorList
{
  const ParamDescriptor* data;
  std::size_t            count;
  constexpr const ParamDescriptor* begin() const { return data; }
  constexpr const ParamDescriptor* end() const { return data + count; }
  constexpr std::size_t            size() const { return count; }
  constexpr const ParamDescriptor& operator[](std::size_t i) const { return data[i]; }
};

namespace paramdesc {

constexpr ParamDescriptor inputBuffer(const char* n, const char* d)
{
  return {n, d, ParamKind::kInputBuffer, 0, false, 0, false, 0, nullptr, 0, 0, 0, nullptr};
}
constexpr ParamDescriptor buffer(const char* n, const char* d)
{
  return {n, d, ParamKind::kBuffer, 0, false, 0, false, 0, nullptr, 0, 0, 0, nullptr};
}
constexpr ParamDescriptor longParam(const char* n, const char* d, double def, const char* rel = nullptr)
{
  return {n, d, ParamKind::kLong, def, false, 0, false, 0, nullptr, 0, 0, 0, rel};
}
constexpr ParamDescriptor longMin(const char* n, const char* d, double def, double lo, const char* rel = nullptr)
{
  return {n, d, ParamKind::kLong, def, true, lo, false, 0, nullptr, 0, 0, 0, rel};
}
constexpr ParamDescriptor longMinMax(const char* n, const char* d, double def, double lo, double hi)
{
  return {n, d, ParamKind::kLong, def, true, lo, true, hi, nullptr, 0, 0, 0, nullptr};
}
constexpr ParamDescriptor floatMin(const char* n, const char* d, double def, double lo)
{
  return {n, d, ParamKind::kFloat, def, true, lo, false, 0, nullptr, 0, 0, 0, nullptr};
}
constexpr ParamDescriptor floatMinMax(const char* n, const char* d, double def, double lo, double hi)
{
  return {n, d, ParamKind::kFloat, def, true, lo, true, hi, nullptr, 0, 0, 0, nullptr};
}
template <int N>
constexpr ParamDescriptor enumParam(const char* n, const char* d, double def, const char* const (&s)[N])
{
  return {n, d, ParamKind::kEnum, def, true, 0, true, N - 1, s, N, 0, 0, nullptr};
}
constexpr ParamDescriptor fft(const char* n, const char* d, long win, long hop, long size)
{
  return {n, d, ParamKind::kFFT, static_cast<double>(win), false, 0, false, 0, nullptr, 0, hop, size, nullptr};
}

inline constexpr const char* kUpdateModes[] = {"None", "Seed", "Fixed"};
inline constexpr const char* kPaddingModes[] = {"None", "Default", "Full"};
inline constexpr const char* kSeedMethods[] = {"NMF-SVD", "NNDSVDar", "NNDSVDa", "NNDSVD"};
inline constexpr const char* kNoYes[] = {"No", "Yes"};
inline constexpr const char* kAmpScale[] = {"Linear", "dB"};

// nrt/NMFClient.hpp:54-71
inline constexpr ParamDescriptor kBufNMF[] = {
    inputBuffer("source", "Source Buffer"),
    longMin("startFrame", "Source Offset", 0, 0),
    longParam("numFrames", "Number of Frames", -1),
    longMin("startChan", "Start Channel", 0, 0),
    longParam("numChans", "Number Channels", -1),
    buffer("resynth", "Resynthesis Buffer"),
    longMinMax("resynthMode", "Resynthesise components", 0, 0, 1),
    buffer("bases", "Bases Buffer"),
    enumParam("basesMode", "Bases Buffer Update Mode", 0, kUpdateModes),
    buffer("activations", "Activations Buffer"),
    enumParam("actMode", "Activations Buffer Update Mode", 0, kUpdateModes),
    longMin("components", "Number of Components", 1, 1),
    longMin("iterations", "Number of Iterations", 100, 1),
    longParam("seed", "Random Seed", -1),
    fft("fftSettings", "FFT Settings", 1024, -1, -1)};

// nrt/NMFSeedClient.hpp:38-52
inline constexpr ParamDescriptor kBufNMFSeed[] = {
    inputBuffer("source", "Source Buffer"),
    buffer("bases", "Bases Buffer"),
    buffer("activations", "Activations Buffer"),
    longMin("minComponents", "Minimum Number of Components", 1, 1, "UpperLimit<maxComponents>"),
    longMin("maxComponents", "Maximum Number of Components", 200, 1, "LowerLimit<minComponents>"),
    floatMinMax("coverage", "Coverage", 0.5, 0, 1),
    enumParam("method", "Initialization Method", 0, kSeedMethods),
    longParam("seed", "Random Seed", -1),
    fft("fftSettings", "FFT Settings", 1024, -1, -1)};

// nrt/BufSTFTClient.hpp:36-47
inline constexpr ParamDescriptor kBufSTFT[] = {
    inputBuffer("source", "Source Buffer"),
    longMin("startFrame", "Source Offset", 0, 0),
    longParam("numFrames", "Number of Frames", -1),
    longMin("startChan", "Start Channel", 0, 0),
    buffer("magnitude", "Magnitude Buffer"),
    buffer("phase", "Phase Buffer"),
    buffer("resynth", "Resynthesis Buffer"),
    longMinMax("inverse", "Inverse Transform", 0, 0, 1),
    enumParam("padding", "Added Padding", 1, kPaddingModes),
    fft("fftSettings", "FFT Settings", 1024, -1, -1)};

// cc/FluidNRTClientWrapper.hpp:33-39 + the output buffer + :747-763 "padding", then rt/MFCCClient.hpp:37-50
inline constexpr ParamDescriptor kBufMFCC[] = {
    inputBuffer("source", "Source Buffer"),
    longMin("startFrame", "Source Offset", 0, 0),
    longParam("numFrames", "Number of Frames", -1),
    longMin("startChan", "Start Channel", 0, 0),
    longParam("numChans", "Number of Channels", -1),
    buffer("features", "Output Buffer"),
    enumParam("padding", "Added Padding", 1, kPaddingModes),
    longMin("numCoeffs", "Number of Cepstral Coefficients", 13, 2, "UpperLimit<numBands>"),
    longMin("numBands", "Number of Bands", 40, 2, "FrameSizeUpperLimit<fftSettings>, LowerLimit<numCoeffs>"),
    longMinMax("startCoeff", "Output Coefficient Offset", 0, 0, 1),
    floatMin("minFreq", "Low Frequency Bound", 20, 0),
    floatMin("maxFreq", "High Frequency Bound", 20000, 0),
    fft("fftSettings", "FFT Settings", 1024, -1, -1)};

// the same wrapper parameters, then rt/MelBandsClient.hpp:37-44
inline constexpr ParamDescriptor kBufMelBands[] = {
    inputBuffer("source", "Source Buffer"),
    longMin("startFrame", "Source Offset", 0, 0),
    longParam("numFrames", "Number of Frames", -1),
    longMin("startChan", "Start Channel", 0, 0),
    longParam("numChans", "Number of Channels", -1),
    buffer("features", "Output Buffer"),
    enumParam("padding", "Added Padding", 1, kPaddingModes),
    longMin("numBands", "Number of Bands", 40, 2),
    floatMin("minFreq", "Low Frequency Bound", 20, 0),
    floatMin("maxFreq", "High Frequency Bound", 20000, 0),
    enumParam("normalize", "Normalize", 1, kNoYes),
    enumParam("scale", "Amplitude Scale", 0, kAmpScale),
    fft("fftSettings", "FFT Settings", 1024, -1, -1)};

// the wrapper's source parameters and ONE output buffer (NMFFilterClient.hpp here), then rt/NMFFilterClient.hpp:34-38
inline constexpr ParamDescriptor kBufNMFFilter[] = {
    inputBuffer("source", "Source Buffer"),
    longMin("startFrame", "Source Offset", 0, 0),
    longParam("numFrames", "Number of Frames", -1),
    longMin("startChan", "Start Channel", 0, 0),
    longParam("numChans", "Number of Channels", -1),
    buffer("resynth", "Resynthesis Buffer"),
    inputBuffer("bases", "Bases Buffer"),
    longMin("maxComponents", "Maximum Number of Components", 20, 1),
    longMin("iterations", "Number of Iterations", 10, 1),
    longParam("seed", "Random Seed", -1),
    fft("fftSettings", "FFT Settings", 1024, -1, -1)};

// the control wrapper's parameters, then rt/NMFMatchClient.hpp:32-38
inline constexpr ParamDescriptor kBufNMFMatch[] = {
    inputBuffer("source", "Source Buffer"),
    longMin("startFrame", "Source Offset", 0, 0),
    longParam("numFrames", "Number of Frames", -1),
    longMin("startChan", "Start Channel", 0, 0),
    longParam("numChans", "Number of Channels", -1),
    buffer("features", "Output Buffer"),
    enumParam("padding", "Added Padding", 1, kPaddingModes),
    inputBuffer("bases", "Bases Buffer"),
    longMin("maxComponents", "Maximum Number of Components", 20, 1),
    longMin("iterations", "Number of Iterations", 10, 1),
    longParam("seed", "Random Seed", -1),
    fft("fftSettings", "FFT Settings", 1024, -1, -1)};

template <std::size_t N>
constexpr ParamDescriptorList list(const ParamDescriptor (&a)[N])
{
  return {a, N};
}

} // namespace paramdesc
} // namespace fluhip
