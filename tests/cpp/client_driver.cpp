// client_driver.cpp -- exercises the host-side BufNMF client (include/flucoma_hip/*.hpp) the way a
// host wrapper would: MemoryBufferAdaptor buffers, NMFParams, NRTThreadedNMFClient sync/async,
// progress polling and cancellation.  Driven by tests/test_client.py, which checks the outputs
// against the oracle.
//
//   client_driver errors
//   client_driver run <in.f32> <frames> <chans> <win> <hop> <fft> <rank> <iters> <seed>
//                     <basesMode> <actMode> <async> <startFrame> <numFrames> <startChan> <numChans>
//                     <outprefix> [<bases_seed.f32> <acts_seed.f32>]
//   client_driver cancel <frames>
//   client_driver pool <count> <frames>      (fluhip_pool_* from a C++ host: two contexts on device 0 vs one)
// and the clients of SURVEY 8 (f): BufSTFT, BufNMFSeed, BufMFCC, BufMelBands
//   client_driver errors2
//   client_driver stft <in.f32> <frames> <chans> <win> <hop> <fft> <padding> <startFrame> <numFrames> <async> <outprefix>
//                      (forward into magnitude / phase buffers, then the inverse from those buffers)
//   client_driver seed <in.f32> <frames> <win> <hop> <fft> <minRank> <maxRank> <coverage> <method> <seed> <async> <outprefix>
//   client_driver mfcc <in.f32> <frames> <chans> <win> <hop> <fft> <padding> <nBands> <nCoefs> <startCoeff>
//                      <startFrame> <numFrames> <startChan> <numChans> <async> <outprefix>
//   client_driver melbands <in.f32> <frames> <chans> <win> <hop> <fft> <padding> <nBands> <normalize> <scale>
//                      <startFrame> <numFrames> <startChan> <numChans> <async> <outprefix>
//   client_driver nmfmatch <in.f32> <frames> <chans> <win> <hop> <fft> <padding> <maxComponents> <seed> <bases.f32|-> <K>
//                      <async> <outprefix>          (bases file: K x (fft/2 + 1) floats, channel-major; "-" = no buffer)
//   client_driver nmffilter <in.f32> <frames> <chans> <win> <hop> <fft> <maxComponents> <iterations> <seed> <bases.f32|->
//                      <K> <async> <outprefix>
#include "../../include/flucoma_hip/BufSTFTClient.hpp"
#include "../../include/flucoma_hip/NMFFilterClient.hpp"
#include "../../include/flucoma_hip/NMFMatchClient.hpp"
#include "../../include/flucoma_hip/MFCCClient.hpp"
#include "../../include/flucoma_hip/MelBandsClient.hpp"
#include "../../include/flucoma_hip/NMFSeedClient.hpp"
#include "../../include/flucoma_hip/NRTThreadingAdaptor.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>

using fluhip::BufferAdaptor; using fluhip::FFTParams; using fluhip::FluidContext; using fluhip::MemoryBufferAdaptor;
using fluhip::NRTThreadedNMFClient; using fluhip::ProcessState; using fluhip::Result; using fluhip::kProcessing;
namespace bufnmf = fluhip::bufnmf;
using idx = fluhip::index; // (::index is a POSIX function)

static std::vector<float> readFile(const char* path)
{
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(3); }
  size_t bytes = (size_t) f.tellg();
  f.seekg(0);
  std::vector<float> v(bytes / sizeof(float));
  f.read(reinterpret_cast<char*>(v.data()), (std::streamsize) bytes);
  return v;
}

static void writeBuffer(const std::string& path, const std::shared_ptr<MemoryBufferAdaptor>& b)
{
  BufferAdaptor::ReadAccess a(b.get());
  std::ofstream f(path, std::ios::binary);
  int64_t hdr[2] = {a.numFrames(), a.numChans()};
  double  sr = a.sampleRate();
  f.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
  f.write(reinterpret_cast<const char*>(&sr), sizeof(sr));
  for (idx c = 0; c < a.numChans(); ++c) // channel-major dump
  {
    auto v = a.samps(c);
    for (idx i = 0; i < v.size(); ++i) { float x = v(i); f.write(reinterpret_cast<const char*>(&x), 4); }
  }
}

static void report(const char* tag, const Result& r)
{
  std::printf("%s|%d|%s\n", tag, (int) r.status(), r.message().c_str());
}

static std::shared_ptr<MemoryBufferAdaptor> makeBuffer(idx chans, idx frames, const float* interleaved = nullptr)
{
  auto b = std::make_shared<MemoryBufferAdaptor>(chans, frames, 44100.0);
  if (interleaved) std::memcpy(b->raw(), interleaved, sizeof(float) * (size_t) (chans * frames));
  return b;
}

static int runErrors()
{
  // every validation branch of nrt/NMFClient.hpp:100-185 that needs no device
  FluidContext          ctx;
  bufnmf::NMFParams     p;
  bufnmf::NMFClient     client(p, ctx);
  report("no_source", client.process<float>(ctx));
  auto src = makeBuffer(2, 4096);
  p.source = src;
  p.startFrame = 5000;
  report("bad_start_frame", client.process<float>(ctx));
  p.startFrame = 0;
  p.startChan = 2;
  report("bad_start_chan", client.process<float>(ctx));
  p.startChan = 0;
  p.numFrames = 5000;
  report("too_many_frames", client.process<float>(ctx));
  p.numFrames = -1;
  p.numChans = 3;
  report("too_many_chans", client.process<float>(ctx));
  p.numChans = -1;
  p.basesMode = 1;
  report("seed_no_bases", client.process<float>(ctx));
  p.bases = makeBuffer(1, 10);
  report("seed_bad_bases_shape", client.process<float>(ctx));
  p.basesMode = 0;
  p.bases.reset();
  p.actMode = 2;
  report("fix_no_acts", client.process<float>(ctx));
  p.activations = makeBuffer(1, 10);
  report("fix_bad_acts_shape", client.process<float>(ctx));
  // both fixed, no resynthesis -> warning, no work
  p.components = 2;
  p.bases = makeBuffer(2 * 2, 513);
  p.activations = makeBuffer(2 * 2, 4096 / 512 + 1);
  p.basesMode = 2;
  p.actMode = 2;
  report("both_fixed", client.process<float>(ctx));
  p.basesMode = 0;
  p.actMode = 0;
  p.resynthMode = 1;
  report("resynth_no_buffer", client.process<float>(ctx));
  // threading adaptor bookkeeping
  bufnmf::NMFParams    q;
  NRTThreadedNMFClient adaptor(q);
  report("empty_queue", adaptor.process());
  return 0;
}

// one job through a threading adaptor, on the caller's thread or on the adaptor's own with progress polling
template <class Adaptor, class Params>
static Result runJob(Params& p, bool async)
{
  Adaptor adaptor(p);
  Result  r;
  adaptor.enqueue(p);
  if (!async)
  {
    adaptor.setSynchronous(true);
    return adaptor.process();
  }
  report("process", adaptor.process());
  ProcessState st = kProcessing;
  while (st == kProcessing)
  {
    st = adaptor.checkProgress(r);
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  return r;
}

static int runErrors2()
{
  // the validation branches of nrt/BufSTFTClient.hpp:84-107,189-216, nrt/NMFSeedClient.hpp:75-88 and
  // cc/FluidNRTClientWrapper.hpp:313-328 that need no device
  FluidContext ctx;
  {
    // the wrapper's checks in front of NMFMatch / NMFFilter (cc/FluidNRTClientWrapper.hpp:313-328)
    fluhip::nmfmatch::NRTNMFMatchParams p;
    fluhip::NRTNMFMatchClient           client(p, ctx);
    report("match_no_source", client.process<float>(ctx));
    p.source = makeBuffer(1, 4096);
    report("match_no_output", client.process<float>(ctx));
    p.maxComponents = 0; p.iterations = 0; p.startFrame = -3; p.padding = 7;
    p.constrain();
    std::printf("match_constraints|%d|\n", (p.maxComponents == 1 && p.iterations == 1 && p.startFrame == 0 && p.padding == 2) ? 1 : 0);
    fluhip::nmffilter::NRTNMFFilterParams q;
    fluhip::NRTNMFFilterClient            filt(q, ctx);
    report("filter_no_source", filt.process<float>(ctx));
    q.source = makeBuffer(2, 4096);
    q.startChan = 2;
    report("filter_bad_start_chan", filt.process<float>(ctx));
    q.startChan = 0;
    report("filter_no_output", filt.process<float>(ctx));
  }
  {
    fluhip::bufstft::BufSTFTParams    p;
    fluhip::bufstft::BufferSTFTClient client(p, ctx);
    report("stft_no_source", client.process<float>(ctx));
    p.source = makeBuffer(1, 4096);
    report("stft_no_outputs", client.process<float>(ctx));
    p.magnitude = makeBuffer(1, 1);
    p.startFrame = 5000;
    report("stft_bad_start_frame", client.process<float>(ctx));
    p.startFrame = 0;
    p.numFrames = 5000;
    report("stft_too_many_frames", client.process<float>(ctx));
    p.numFrames = -1;
    p.fftSettings = FFTParams(1024, 512, 131072); // 65537 bins
    report("stft_too_many_bins", client.process<float>(ctx));
    p.fftSettings = FFTParams(1024, 512, 1024);
    p.inverse = 1;
    report("istft_needs_both", client.process<float>(ctx));
    p.phase = makeBuffer(513, 9);
    report("istft_no_resynth", client.process<float>(ctx));
    p.resynth = makeBuffer(1, 1);
    p.magnitude = makeBuffer(513, 8);
    report("istft_size_mismatch", client.process<float>(ctx));
    p.magnitude = makeBuffer(512, 9);
    p.phase = makeBuffer(512, 9);
    report("istft_wrong_channels", client.process<float>(ctx));
  }
  {
    fluhip::nndsvd::NMFSeedParams p;
    fluhip::nndsvd::NMFSeedClient client(p, ctx);
    report("seed_no_source", client.process<float>(ctx));
    p.source = makeBuffer(2, 4096);
    report("seed_two_channels", client.process<float>(ctx));
  }
  {
    fluhip::mfcc::NRTMFCCParams p;
    fluhip::NRTMFCCClient       client(p, ctx);
    report("mfcc_no_source", client.process<float>(ctx));
    p.source = makeBuffer(2, 4096);
    p.startChan = 2;
    report("mfcc_bad_start_chan", client.process<float>(ctx));
    p.startChan = 0;
    report("mfcc_no_output", client.process<float>(ctx));
    fluhip::melbands::NRTMelBandsParams q;
    fluhip::NRTMelBandsClient           mel(q, ctx);
    q.source = makeBuffer(1, 4096);
    q.numChans = 2;
    report("melbands_too_many_chans", mel.process<float>(ctx));
    q.numChans = -1;
    report("melbands_no_output", mel.process<float>(ctx));
    // constraints of the parameter tables (rt/MFCCClient.hpp:38-49)
    p.numCoeffs = 50;
    p.numBands = 1000;
    p.startCoeff = 7;
    p.constrain();
    std::printf("mfcc_constraints|%d|%ld %ld %ld\n", (p.numBands == 513 && p.numCoeffs == 50 && p.startCoeff == 1) ? 1 : 0,
                (long) p.numBands, (long) p.numCoeffs, (long) p.startCoeff);
  }
  return 0;
}

// the parameter tables as JSON, one object per client, through the adaptor the host would hold
template <class Adaptor>
static void printDescriptors(const char* client, bool last)
{
  constexpr auto list = Adaptor::getParameterDescriptors();
  static const char* kinds[] = {"InputBuffer", "Buffer", "Long", "Float", "Enum", "FFT"};
  std::printf("\"%s\": [", client);
  for (std::size_t i = 0; i < list.size(); i++)
  {
    const fluhip::ParamDescriptor& d = list[i];
    std::printf("%s{\"name\": \"%s\", \"display\": \"%s\", \"kind\": \"%s\"", i ? ", " : "", d.name, d.displayName,
                kinds[static_cast<int>(d.kind)]);
    if (d.kind == fluhip::ParamKind::kLong || d.kind == fluhip::ParamKind::kFloat || d.kind == fluhip::ParamKind::kEnum)
      std::printf(", \"default\": %.17g", d.defaultValue);
    if (d.kind == fluhip::ParamKind::kFFT) std::printf(", \"default\": [%ld, %ld, %ld]", (long) d.defaultValue, d.fftHop, d.fftSize);
    if (d.kind != fluhip::ParamKind::kEnum && d.hasMin) std::printf(", \"min\": %.17g", d.min);
    if (d.kind != fluhip::ParamKind::kEnum && d.hasMax) std::printf(", \"max\": %.17g", d.max);
    if (d.kind == fluhip::ParamKind::kEnum)
    {
      std::printf(", \"strings\": [");
      for (int j = 0; j < d.numEnumStrings; j++) std::printf("%s\"%s\"", j ? ", " : "", d.enumStrings[j]);
      std::printf("]");
    }
    if (d.relational) std::printf(", \"relational\": \"%s\"", d.relational);
    std::printf("}");
  }
  std::printf("]%s\n", last ? "" : ",");
}
// the adaptor's connection counts (cc/FluidNRTClientWrapper.hpp:811-822): "client|buffersIn buffersOut audioIn audioOut controlIn controlOut.count|messages"
template <class Adaptor, class Params>
static void printConnections(const char* client)
{
  Params       p;
  FluidContext ctx;
  Adaptor      a(p, ctx);
  std::printf("%s|%ld %ld %ld %ld %ld %ld|%ld\n", client, (long) a.audioBuffersIn(), (long) a.audioBuffersOut(), (long) a.audioChannelsIn(),
              (long) a.audioChannelsOut(), (long) a.controlChannelsIn(), (long) a.controlChannelsOut().count,
              (long) Adaptor::getMessageDescriptors().size());
}

int main(int argc, char** argv)
{
  if (argc < 2) return 2;
  const std::string mode = argv[1];
  if (mode == "descriptors")
  {
    std::printf("{\n");
    printDescriptors<NRTThreadedNMFClient>("BufNMF", false);
    printDescriptors<fluhip::NRTThreadedNMFSeedClient>("BufNMFSeed", false);
    printDescriptors<fluhip::NRTThreadedBufferSTFTClient>("BufSTFT", false);
    printDescriptors<fluhip::NRTThreadedMFCCClient>("BufMFCC", false);
    printDescriptors<fluhip::NRTThreadedMelBandsClient>("BufMelBands", false);
    printDescriptors<fluhip::NRTThreadedNMFFilterClient>("BufNMFFilter", false);
    printDescriptors<fluhip::NRTThreadedNMFMatchClient>("BufNMFMatch", true);
    std::printf("}\n");
    return 0;
  }
  if (mode == "connections")
  {
    printConnections<NRTThreadedNMFClient, bufnmf::NMFParams>("BufNMF");
    printConnections<fluhip::NRTThreadedNMFSeedClient, fluhip::nndsvd::NMFSeedParams>("BufNMFSeed");
    printConnections<fluhip::NRTThreadedBufferSTFTClient, fluhip::bufstft::BufSTFTParams>("BufSTFT");
    printConnections<fluhip::NRTThreadedMFCCClient, fluhip::mfcc::NRTMFCCParams>("BufMFCC");
    printConnections<fluhip::NRTThreadedMelBandsClient, fluhip::melbands::NRTMelBandsParams>("BufMelBands");
    printConnections<fluhip::NRTThreadedNMFFilterClient, fluhip::nmffilter::NRTNMFFilterParams>("BufNMFFilter");
    printConnections<fluhip::NRTThreadedNMFMatchClient, fluhip::nmfmatch::NRTNMFMatchParams>("BufNMFMatch");
    return 0;
  }
  if (mode == "errors") return runErrors();
  if (mode == "errors2") return runErrors2();

  if (mode == "stft")
  {
    if (argc < 13) return 2;
    auto      in = readFile(argv[2]);
    const idx frames = std::atol(argv[3]), chans = std::atol(argv[4]);
    fluhip::bufstft::BufSTFTParams p;
    p.source = makeBuffer(chans, frames, in.data());
    p.fftSettings = FFTParams(std::atol(argv[5]), std::atol(argv[6]), std::atol(argv[7]));
    p.padding = std::atol(argv[8]);
    p.startFrame = std::atol(argv[9]);
    p.numFrames = std::atol(argv[10]);
    const bool        async = std::atoi(argv[11]) != 0;
    const std::string prefix = argv[12];
    p.constrain();
    auto mag = makeBuffer(1, 1), phase = makeBuffer(1, 1), resynth = makeBuffer(1, 1);
    p.magnitude = mag;
    p.phase = phase;
    report("forward", runJob<fluhip::NRTThreadedBufferSTFTClient>(p, async));
    writeBuffer(prefix + "_mag.bin", mag);
    writeBuffer(prefix + "_phase.bin", phase);
    p.inverse = 1;
    p.resynth = resynth;
    report("inverse", runJob<fluhip::NRTThreadedBufferSTFTClient>(p, async));
    writeBuffer(prefix + "_resynth.bin", resynth);
    return 0;
  }

  if (mode == "seed")
  {
    if (argc < 14) return 2;
    auto      in = readFile(argv[2]);
    const idx frames = std::atol(argv[3]);
    fluhip::nndsvd::NMFSeedParams p;
    p.source = makeBuffer(1, frames, in.data());
    p.fftSettings = FFTParams(std::atol(argv[4]), std::atol(argv[5]), std::atol(argv[6]));
    p.minComponents = std::atol(argv[7]);
    p.maxComponents = std::atol(argv[8]);
    p.coverage = std::atof(argv[9]);
    p.method = std::atol(argv[10]);
    p.seed = std::atol(argv[11]);
    const bool        async = std::atoi(argv[12]) != 0;
    const std::string prefix = argv[13];
    p.constrain();
    auto bases = makeBuffer(1, 1), acts = makeBuffer(1, 1);
    p.bases = bases;
    p.activations = acts;
    report("result", runJob<fluhip::NRTThreadedNMFSeedClient>(p, async));
    writeBuffer(prefix + "_bases.bin", bases);
    writeBuffer(prefix + "_acts.bin", acts);
    return 0;
  }

  if (mode == "mfcc" || mode == "melbands")
  {
    if (argc < 18) return 2;
    auto      in = readFile(argv[2]);
    const idx frames = std::atol(argv[3]), chans = std::atol(argv[4]);
    auto      features = makeBuffer(1, 1);
    auto      wrapper = [&](fluhip::NRTControlParams& w) {
      w.source = makeBuffer(chans, frames, in.data());
      w.padding = std::atol(argv[8]);
      w.startFrame = std::atol(argv[12]);
      w.numFrames = std::atol(argv[13]);
      w.startChan = std::atol(argv[14]);
      w.numChans = std::atol(argv[15]);
      w.features = features;
    };
    const FFTParams   fft(std::atol(argv[5]), std::atol(argv[6]), std::atol(argv[7]));
    const bool        async = std::atoi(argv[16]) != 0;
    const std::string prefix = argv[17];
    Result            r;
    if (mode == "mfcc")
    {
      fluhip::mfcc::NRTMFCCParams p;
      wrapper(p);
      p.fftSettings = fft;
      p.numBands = std::atol(argv[9]);
      p.numCoeffs = std::atol(argv[10]);
      p.startCoeff = std::atol(argv[11]);
      p.constrain();
      r = runJob<fluhip::NRTThreadedMFCCClient>(p, async);
    }
    else
    {
      fluhip::melbands::NRTMelBandsParams p;
      wrapper(p);
      p.fftSettings = fft;
      p.numBands = std::atol(argv[9]);
      p.normalize = std::atol(argv[10]);
      p.scale = std::atol(argv[11]);
      p.constrain();
      r = runJob<fluhip::NRTThreadedMelBandsClient>(p, async);
    }
    report("result", r);
    writeBuffer(prefix + "_features.bin", features);
    return 0;
  }

  if (mode == "nmfmatch" || mode == "nmffilter")
  {
    if (argc < 15) return 2;
    auto      in = readFile(argv[2]);
    const idx frames = std::atol(argv[3]), chans = std::atol(argv[4]);
    const FFTParams fft(std::atol(argv[5]), std::atol(argv[6]), std::atol(argv[7]));
    auto      output = makeBuffer(1, 1);
    auto      loadBases = [&](const char* path, idx K) -> std::shared_ptr<MemoryBufferAdaptor> {
      if (std::string(path) == "-") return nullptr;
      auto      sb = readFile(path);
      const idx F = (idx) sb.size() / K;      // (a wrong frame count is one of the cases: take what the file holds)
      auto      b = makeBuffer(K, F);
      for (idx c = 0; c < K; ++c)
        for (idx f = 0; f < F; ++f) b->raw()[f * K + c] = sb[(size_t) (c * F + f)];
      return b;
    };
    Result r;
    if (mode == "nmfmatch")
    {
      fluhip::nmfmatch::NRTNMFMatchParams p;
      p.source = makeBuffer(chans, frames, in.data());
      p.features = output;
      p.fftSettings = fft;
      p.padding = std::atol(argv[8]);
      p.maxComponents = std::atol(argv[9]);
      p.seed = std::atol(argv[10]);
      p.bases = loadBases(argv[11], std::atol(argv[12]));
      p.constrain();
      r = runJob<fluhip::NRTThreadedNMFMatchClient>(p, std::atoi(argv[13]) != 0);
      report("result", r);
      writeBuffer(std::string(argv[14]) + "_features.bin", output);
    }
    else
    {
      fluhip::nmffilter::NRTNMFFilterParams p;
      p.source = makeBuffer(chans, frames, in.data());
      p.resynth = output;
      p.fftSettings = fft;
      p.maxComponents = std::atol(argv[8]);
      p.iterations = std::atol(argv[9]);
      p.seed = std::atol(argv[10]);
      p.bases = loadBases(argv[11], std::atol(argv[12]));
      p.constrain();
      r = runJob<fluhip::NRTThreadedNMFFilterClient>(p, std::atoi(argv[13]) != 0);
      report("result", r);
      writeBuffer(std::string(argv[14]) + "_resynth.bin", output);
    }
    return 0;
  }

  if (mode == "cancel")
  {
    const idx frames = std::atol(argv[2]);
    std::vector<float> x((size_t) frames);
    for (idx i = 0; i < frames; ++i) x[(size_t) i] = 0.5f * std::sin(0.05f * i) + 0.25f * std::sin(0.31f * i);
    bufnmf::NMFParams p;
    p.source = makeBuffer(1, frames, x.data());
    p.bases = makeBuffer(1, 1);
    p.activations = makeBuffer(1, 1);
    p.components = 16;
    p.iterations = 100000; // far more than can finish before the cancel lands
    p.seed = 42;
    p.fftSettings = FFTParams(2048, 512, 2048);
    NRTThreadedNMFClient adaptor(p);
    adaptor.enqueue(p);
    report("process", adaptor.process());
    Result r;
    double lastProgress = 0;
    for (int i = 0; i < 50 && adaptor.progress() <= 0.0; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(20));
    lastProgress = adaptor.progress();
    adaptor.cancel();
    ProcessState st = kProcessing;
    for (int i = 0; i < 2000 && st == kProcessing; ++i)
    {
      st = adaptor.checkProgress(r);
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    std::printf("progress_before_cancel|%d|%g\n", lastProgress > 0.0 && lastProgress < 1.0 ? 1 : 0, lastProgress);
    report("cancelled", r);
    return 0;
  }

  if (mode == "pool")
  {
    const idx count = std::atol(argv[2]), frames = std::atol(argv[3]);
    std::vector<float> x((size_t) (count * frames));
    for (idx b = 0; b < count; ++b)
      for (idx i = 0; i < frames; ++i)
        x[(size_t) (b * frames + i)] = 0.5f * std::sin(0.01f * (float) (b + 3) * (float) i) + 0.25f * std::sin(0.31f * (float) i + (float) b);
    const idx win = 1024, fft = 1024, hop = 256, K = 4, iters = 10, F = fft / 2 + 1, T = fluhip_stft_num_frames(frames, win, hop);
    auto runPool = [&](std::vector<int> devs, std::vector<float>& bases, std::vector<float>& acts) -> int {
      fluhip_pool* pool = nullptr;
      if (fluhip_pool_create(devs.data(), (int) devs.size(), &pool) != FLUHIP_OK) return -1;
      bases.assign((size_t) (count * K * F), 0.f);
      acts.assign((size_t) (count * K * T), 0.f);
      const int rc = fluhip_pool_bufnmf_f32(pool, x.data(), count, frames, win, fft, hop, K, iters, 1, 1, 42, nullptr, bases.data(),
                                            acts.data(), nullptr, nullptr);
      if (rc != FLUHIP_OK) std::fprintf(stderr, "pool: %s\n", fluhip_pool_last_error(pool));
      fluhip_pool_destroy(pool);
      return rc;
    };
    std::vector<float> b1, a1, b2, a2;
    const int rc1 = runPool({0}, b1, a1), rc2 = runPool({0, 0}, b2, a2);
    double maxd = 0, maxv = 0;
    for (size_t i = 0; i < b1.size(); ++i) { maxd = std::max(maxd, (double) std::fabs(b1[i] - b2[i])); maxv = std::max(maxv, (double) std::fabs(b1[i])); }
    for (size_t i = 0; i < a1.size(); ++i) maxd = std::max(maxd, (double) std::fabs(a1[i] - a2[i]));
    std::printf("pool_rc|%d|%d\n", rc1, rc2);
    std::printf("pool_match|%d|%g %g\n", (maxd <= 1e-6 && maxv > 0) ? 1 : 0, maxd, maxv);
    return 0;
  }

  if (mode == "run")
  {
    if (argc < 19) return 2;
    auto        in = readFile(argv[2]);
    const idx frames = std::atol(argv[3]), chans = std::atol(argv[4]);
    bufnmf::NMFParams p;
    p.source = makeBuffer(chans, frames, in.data());
    p.fftSettings = FFTParams(std::atol(argv[5]), std::atol(argv[6]), std::atol(argv[7]));
    p.components = std::atol(argv[8]);
    p.iterations = std::atol(argv[9]);
    p.seed = std::atol(argv[10]);
    p.basesMode = std::atol(argv[11]);
    p.actMode = std::atol(argv[12]);
    const bool async = std::atoi(argv[13]) != 0;
    p.startFrame = std::atol(argv[14]);
    p.numFrames = std::atol(argv[15]);
    p.startChan = std::atol(argv[16]);
    p.numChans = std::atol(argv[17]);
    const std::string prefix = argv[18];
    p.constrain();
    idx nCh = p.numChans < 0 ? chans - p.startChan : p.numChans;
    idx nFr = p.numFrames < 0 ? frames - p.startFrame : p.numFrames;
    auto  bases = makeBuffer(1, 1);
    auto  acts = makeBuffer(1, 1);
    if (argc >= 21)
    {
      auto        sb = readFile(argv[19]);
      auto        sa = readFile(argv[20]);
      const idx F = p.fftSettings.frameSize(), T = nFr / p.fftSettings.hopSize() + 1, KC = p.components * nCh;
      bases = makeBuffer(KC, F);
      acts = makeBuffer(KC, T);
      // seed files are channel-major [KC][frames]
      for (idx c = 0; c < KC; ++c)
      {
        for (idx f = 0; f < F; ++f) bases->raw()[f * KC + c] = sb[(size_t) (c * F + f)];
        for (idx t = 0; t < T; ++t) acts->raw()[t * KC + c] = sa[(size_t) (c * T + t)];
      }
    }
    p.bases = bases;
    p.activations = acts;
    auto resynth = makeBuffer(1, 1);
    if (std::getenv("CLIENT_RESYNTH"))
    {
      p.resynth = resynth;
      p.resynthMode = 1;
    }
    // CLIENT_DEVICES=0,0 : the channels of the job dealt over several device contexts (FluidContext::devices)
    FluidContext hostCtx;
    if (const char* e = std::getenv("CLIENT_DEVICES"))
    {
      std::vector<int> devs;
      for (const char* q = e; *q;)
      {
        devs.push_back(std::atoi(q));
        while (*q && *q != ',') ++q;
        if (*q == ',') ++q;
      }
      hostCtx.devices(devs);
    }
    Result r;
    if (!async)
    {
      // CLIENT_REPEAT=n: the same job n times on one client (the first pays for the context and the code objects);
      // elapsed_ms is the last one's wall time around process()
      const int repeat = std::getenv("CLIENT_REPEAT") ? std::max(1, std::atoi(std::getenv("CLIENT_REPEAT"))) : 1;
      NRTThreadedNMFClient adaptor(p, hostCtx);
      adaptor.setSynchronous(true);
      double ms = 0;
      for (int rep = 0; rep < repeat; ++rep)
      {
        adaptor.enqueue(p);
        const auto t0 = std::chrono::steady_clock::now();
        r = adaptor.process();
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (std::getenv("CLIENT_REPEAT_PRINT")) std::fprintf(stderr, "repeat %d: %.3f ms\n", rep, ms);
      }
      std::printf("elapsed_ms|1|%.3f\n", ms);
    }
    else
    {
      NRTThreadedNMFClient adaptor(p, hostCtx);
      const int repeat = std::getenv("CLIENT_REPEAT") ? std::max(1, std::atoi(std::getenv("CLIENT_REPEAT"))) : 1;
      double    maxProgress = 0, ms = 0;
      for (int rep = 0; rep < repeat; ++rep)
      {
        const auto t0 = std::chrono::steady_clock::now();
        adaptor.enqueue(p);
        Result pr = adaptor.process();
        if (rep == 0) report("process", pr);
        ProcessState st = kProcessing;
        while (st == kProcessing)
        {
          maxProgress = std::max(maxProgress, adaptor.progress());
          st = adaptor.checkProgress(r);
          if (st == kProcessing) std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (std::getenv("CLIENT_REPEAT_PRINT")) std::fprintf(stderr, "repeat %d: %.3f ms\n", rep, ms);
      }
      std::printf("async_wall_ms|1|%.3f\n", ms);
      std::printf("max_progress|%d|%g\n", maxProgress <= 1.0 ? 1 : 0, maxProgress);
    }
    report("result", r);
    std::printf("fallbacks|%d|\n", fluhip::bufnmf::NMFClient::batchedFallbacks().load());
    writeBuffer(prefix + "_bases.bin", bases);
    writeBuffer(prefix + "_acts.bin", acts);
    if (p.resynthMode) writeBuffer(prefix + "_resynth.bin", resynth);
    return 0;
  }
  return 2;
}
