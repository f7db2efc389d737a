// host_buffers_check.cpp -- CPU-only checks of the host-side buffer plumbing (include/flucoma_hip/BufferAdaptor.hpp,
// NRTThreadingAdaptor.hpp): block-wise scatter / gather against the channel-by-channel loops they replace, on an
// interleaved buffer and on a planar one; interleaved-layout detection; MemoryBufferAdaptor deep copies (memcpy path and
// block path), copy-back with resize, refilling a kept copy, shape-only copies; the job layer with a client that needs no
// device (copies in, runs on a worker thread, copies back on the polling thread, write-only buffers, copies reused by the
// next job).  Prints "ok <name>" per check; tests/test_oracle_ref.py builds it with -fsanitize=address,undefined.
#include "../../include/flucoma_hip/NRTThreadingAdaptor.hpp"

#include <cstdio>
#include <numeric>

namespace chk {
using namespace fluhip;
using fluhip::index; // (::index is a POSIX function)

// a planar (channel-major) host buffer: samps(c) is contiguous, the opposite of MemoryBufferAdaptor
class PlanarBuffer : public BufferAdaptor
{
public:
  PlanarBuffer(index chans, index frames) : mData((size_t) (chans * frames), 0.f), mFrames(frames), mChans(chans) {}
  bool        acquire() const override { return true; }
  void        release() const override {}
  bool        valid() const override { return true; }
  bool        exists() const override { return true; }
  std::string asString() const override { return "planar"; }
  Result      resize(index frames, index channels, double sr) override
  {
    mFrames = frames; mChans = channels; mSR = sr;
    mData.assign((size_t) (frames * channels), 0.f);
    return {};
  }
  VectorView<float>       samps(index c) override { return {mData.data() + c * mFrames, mFrames, 1}; }
  VectorView<float>       samps(index off, index n, index c) override { return {mData.data() + c * mFrames + off, n, 1}; }
  VectorView<const float> samps(index c) const override { return {mData.data() + c * mFrames, mFrames, 1}; }
  VectorView<const float> samps(index off, index n, index c) const override { return {mData.data() + c * mFrames + off, n, 1}; }
  MatrixView<float>       allFrames() override { return MatrixView<float>(mData.data(), mChans, mFrames); }
  MatrixView<const float> allFrames() const override { return MatrixView<const float>(mData.data(), mChans, mFrames); }
  index                   numFrames() const override { return mFrames; }
  index                   numChans() const override { return mChans; }
  double                  sampleRate() const override { return mSR; }
  std::vector<float>      mData;
  index                   mFrames, mChans;
  double                  mSR{44100};
};

static int fails = 0;
static void check(bool ok, const char* name)
{
  std::printf("%s %s\n", ok ? "ok" : "FAILED", name);
  if (!ok) ++fails;
}

template <class Buf>
static bool sameAsChannelLoop(Buf& b, index chans, index frames)
{
  std::vector<float> src((size_t) (chans * frames));
  for (size_t i = 0; i < src.size(); ++i) src[i] = (float) (i % 9973) * 0.5f - 17.f;
  {
    BufferAdaptor::Access a(&b);
    scatterChannels(a, 0, chans, src.data(), frames);
  }
  bool ok = true;
  {
    BufferAdaptor::ReadAccess r(&b);
    for (index c = 0; c < chans && ok; ++c)
      for (index t = 0; t < frames; ++t)
        if (r.samps(c)(t) != src[(size_t) (c * frames + t)]) { ok = false; break; }
    std::vector<float> back(src.size(), -1.f);
    gatherChannels(r, 0, chans, back.data(), frames);
    ok = ok && back == src;
    // a sub-range of channels
    std::vector<float> part((size_t) (2 * frames), 0.f);
    gatherChannels(r, 1, 2, part.data(), frames);
    for (index t = 0; t < frames && ok; ++t) ok = part[(size_t) t] == src[(size_t) (frames + t)] && part[(size_t) (frames + t)] == src[(size_t) (2 * frames + t)];
  }
  return ok;
}

// a client that needs no device: out[c][t] = 2 in[c][t] + c, into a write-only buffer; `state` is read AND written
struct ToyParams
{
  std::shared_ptr<const BufferAdaptor> source;
  std::shared_ptr<BufferAdaptor>       out, state;
  template <class In, class Out>
  void forEachBuffer(In&& in, Out&& o) { forEachBuffer(in, o, o); }
  template <class In, class Out, class OutOnly>
  void forEachBuffer(In&& in, Out&& o, OutOnly&& oo) { in(source); oo(out); o(state); }
};
struct ToyClient
{
  using ParamSetViewType = ToyParams;
  ToyClient(ToyParams& p, FluidContext&) : mP(&p) {}
  void setParams(ToyParams& p) { mP = &p; }
  template <typename T>
  Result process(FluidContext& c)
  {
    BufferAdaptor::ReadAccess src(mP->source.get());
    BufferAdaptor::Access     out(mP->out.get()), st(mP->state.get());
    Result r = out.resize(src.numFrames(), src.numChans(), src.sampleRate());
    if (!r.ok()) return r;
    for (index ch = 0; ch < src.numChans(); ++ch)
      for (index t = 0; t < src.numFrames(); ++t) out.samps(ch)(t) = 2.f * src.samps(ch)(t) + (float) ch;
    st.samps(0)(0) += 1.f; // reads what the host buffer held
    if (c.task()) c.task()->processUpdate(1, 1);
    return {};
  }
  ToyParams* mP;
};

static int run()
{
  const index chans = 5, frames = 1000;
  {
    MemoryBufferAdaptor inter(chans, frames);
    PlanarBuffer        planar(chans, frames);
    check(sameAsChannelLoop(inter, chans, frames), "scatter_gather_interleaved");
    check(sameAsChannelLoop(planar, chans, frames), "scatter_gather_planar");
    index stride = 0;
    {
      BufferAdaptor::Access a(&inter);
      float*                base = interleavedBase(a, stride);
      check(base == inter.raw() && stride == chans, "interleaved_detected");
    }
    {
      BufferAdaptor::Access a(&planar);
      check(interleavedBase(a, stride) == nullptr, "planar_not_interleaved");
    }
    MemoryBufferAdaptor one(1, frames);
    {
      BufferAdaptor::Access a(&one);
      check(interleavedBase(a, stride) == one.raw() && stride == 1, "mono_is_interleaved");
    }
  }
  {
    // deep copies: memcpy path (interleaved origin) and block path (planar origin), copy-back with a resize in between
    auto fill = [](BufferAdaptor& b) {
      BufferAdaptor::Access a(&b);
      for (index c = 0; c < a.numChans(); ++c)
        for (index t = 0; t < a.numFrames(); ++t) a.samps(c)(t) = (float) (c * 10000 + t);
    };
    auto equal = [](const BufferAdaptor& x, const BufferAdaptor& y) {
      BufferAdaptor::ReadAccess a(&x), b(&y);
      if (a.numChans() != b.numChans() || a.numFrames() != b.numFrames()) return false;
      for (index c = 0; c < a.numChans(); ++c)
        for (index t = 0; t < a.numFrames(); ++t)
          if (a.samps(c)(t) != b.samps(c)(t)) return false;
      return true;
    };
    std::shared_ptr<BufferAdaptor> inter = std::make_shared<MemoryBufferAdaptor>(chans, frames);
    std::shared_ptr<BufferAdaptor> planar = std::make_shared<PlanarBuffer>(chans, frames);
    fill(*inter); fill(*planar);
    MemoryBufferAdaptor c1(inter), c2(planar);
    check(equal(c1, *inter) && equal(c2, *planar), "deep_copy_both_layouts");
    Result r;
    BufferAdaptor::Access(&c1).resize(frames + 7, chans + 1, 48000); fill(c1);
    BufferAdaptor::Access(&c2).resize(frames + 7, chans + 1, 48000); fill(c2);
    c1.copyToOrigin(r); c2.copyToOrigin(r);
    check(r.ok() && equal(c1, *inter) && equal(c2, *planar) && BufferAdaptor::ReadAccess(planar.get()).numChans() == chans + 1, "copy_back_resizes_origin");
    // a kept copy refilled from its origin; a shape-only copy
    fill(*inter);
    { BufferAdaptor::Access a(inter.get()); a.samps(2)(5) = -1234.f; }
    c1.rebind(inter, true);
    check(equal(c1, *inter), "rebind_refills");
    MemoryBufferAdaptor shape(planar, false);
    {
      BufferAdaptor::ReadAccess sh(&shape);
      check(sh.numFrames() == frames + 7 && sh.numChans() == chans + 1 && sh.exists() && sh.valid(), "shape_only_copy");
    }
  }
  {
    // the job layer without a device
    auto src = std::make_shared<MemoryBufferAdaptor>(3, 500);
    { BufferAdaptor::Access a(src.get()); for (index c = 0; c < 3; ++c) for (index t = 0; t < 500; ++t) a.samps(c)(t) = (float) (t - 7 * c); }
    auto out = std::make_shared<PlanarBuffer>(1, 1);
    auto state = std::make_shared<MemoryBufferAdaptor>(1, 4);
    ToyParams p;
    p.source = src; p.out = out; p.state = state;
    NRTThreadingAdaptor<ToyClient> adaptor(p);
    bool ok = true;
    for (int job = 0; job < 3 && ok; ++job) // the second and third jobs refill the copies the first one made
    {
      adaptor.enqueue(p);
      Result pr = adaptor.process();
      ok = ok && pr.ok();
      Result       r;
      ProcessState st = kProcessing;
      while (st == kProcessing) { st = adaptor.checkProgress(r); std::this_thread::yield(); }
      ok = ok && r.ok() && out->numFrames() == 500 && out->numChans() == 3;
      for (index c = 0; c < 3 && ok; ++c)
        for (index t = 0; t < 500; ++t)
          if (out->samps(c)(t) != 2.f * (float) (t - 7 * c) + (float) c) { ok = false; break; }
      ok = ok && BufferAdaptor::ReadAccess(state.get()).samps(0)(0) == (float) (job + 1); // read-modify-write through the copy
    }
    check(ok, "threaded_jobs_copy_in_run_copy_back");
    adaptor.setSynchronous(true);
    adaptor.enqueue(p);
    Result r = adaptor.process();
    check(r.ok() && BufferAdaptor::ReadAccess(state.get()).samps(0)(0) == 4.f, "synchronous_job_on_the_host_buffers");
  }
  return fails == 0 ? 0 : 1;
}
} // namespace chk

int main() { return chk::run(); }
