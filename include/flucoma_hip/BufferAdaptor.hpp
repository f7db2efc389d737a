// BufferAdaptor.hpp -- the host-buffer contract of the BufNMF drop-in.
//
//   client::BufferAdaptor (+ ReadAccess / Access)  include/flucoma/clients/common/BufferAdaptor.hpp:18-208
//   client::MemoryBufferAdaptor                    include/flucoma/clients/common/MemoryBufferAdaptor.hpp:21-140
//   client::bufferRangeCheck                       include/flucoma/clients/common/BufferAdaptor.hpp:175-208
//
// The virtual set, the RAII access objects and the range-check messages are those of the
// reference, so a host wrapper's existing BufferAdaptor subclasses port by changing the base
// class.  Element type is float, a buffer is frames x channels, views may be strided.
#pragma once
#include <algorithm>
#include <cstring>

#include "Types.hpp"

#include <memory>
#include <ostream>
#include <vector>

namespace fluhip {

class BufferAdaptor
{
public:
  class ReadAccess
  {
  public:
    explicit ReadAccess(const BufferAdaptor* a) : mAdaptor(a && a->acquire() ? a : nullptr) {}
    ~ReadAccess()
    {
      if (mAdaptor) mAdaptor->release();
    }
    ReadAccess(const ReadAccess&) = delete;
    ReadAccess& operator=(const ReadAccess&) = delete;
    ReadAccess(ReadAccess&& o) noexcept : mAdaptor(o.mAdaptor) { o.mAdaptor = nullptr; }

    bool valid() const { return mAdaptor ? mAdaptor->valid() : false; }
    bool exists() const { return mAdaptor ? mAdaptor->exists() : false; }
    MatrixView<const float> allFrames() const { return mAdaptor->allFrames(); }
    VectorView<const float> samps(index channel) const { return mAdaptor->samps(channel); }
    VectorView<const float> samps(index offset, index nframes, index chan) const
    {
      return mAdaptor->samps(offset, nframes, chan);
    }
    index  numFrames() const { return mAdaptor ? mAdaptor->numFrames() : 0; }
    index  numChans() const { return mAdaptor ? mAdaptor->numChans() : 0; }
    double sampleRate() const { return mAdaptor ? mAdaptor->sampleRate() : 0; }

  protected:
    const BufferAdaptor* mAdaptor;
  };

  class Access : public ReadAccess
  {
  public:
    explicit Access(BufferAdaptor* a) : ReadAccess(a), mMutable(a) {}
    ~Access()
    {
      if (mMutable) mMutable->refresh(); // cc/BufferAdaptor.hpp:87-90
    }
    MatrixView<float> allFrames() { return mMutable->allFrames(); }
    VectorView<float> samps(index channel) { return mMutable->samps(channel); }
    VectorView<float> samps(index offset, index nframes, index chan) { return mMutable->samps(offset, nframes, chan); }
    Result            resize(index frames, index channels, double sampleRate)
    {
      return mMutable ? mMutable->resize(frames, channels, sampleRate)
                      : Result{Result::Status::kError, "Trying to resize null buffer"};
    }
    void refresh()
    {
      if (mMutable) mMutable->refresh();
    }

  private:
    BufferAdaptor* mMutable;
  };

  BufferAdaptor() = default;
  virtual ~BufferAdaptor() = default;

  virtual std::string asString() const = 0;

private:
  friend class ReadAccess;
  friend class Access;
  // cc/BufferAdaptor.hpp:144-166
  virtual bool                    acquire() const = 0;
  virtual void                    release() const = 0;
  virtual bool                    valid() const = 0;
  virtual bool                    exists() const = 0;
  virtual Result                  resize(index frames, index channels, double sampleRate) = 0;
  virtual VectorView<float>       samps(index channel) = 0;
  virtual VectorView<float>       samps(index offset, index nframes, index chanoffset) = 0;
  virtual VectorView<const float> samps(index channel) const = 0;
  virtual VectorView<const float> samps(index offset, index nframes, index chanoffset) const = 0;
  virtual MatrixView<float>       allFrames() = 0;
  virtual MatrixView<const float> allFrames() const = 0;
  virtual index                   numFrames() const = 0;
  virtual index                   numChans() const = 0;
  virtual double                  sampleRate() const = 0;
  virtual void                    refresh() {}
};

inline std::ostream& operator<<(std::ostream& os, const BufferAdaptor* b) { return os << b->asString(); }

// cc/BufferAdaptor.hpp:175-208 -- same order of checks, same messages, resolves -1 counts in place
inline Result bufferRangeCheck(const BufferAdaptor* b, index startFrame, index& nFrames, index startChan,
                               index& nChans)
{
  using S = Result::Status;
  if (!b) return {S::kError, "Input buffer not set"};
  BufferAdaptor::ReadAccess in(b);
  if (!in.exists()) return {S::kError, "Input buffer ", b, " not found."};
  if (!in.valid()) return {S::kError, "Input buffer ", b, " invalid (possibly zero-size?)"};
  if (startFrame >= in.numFrames() || startFrame < 0)
    return {S::kError, "Input buffer ", b, " invalid start frame ", startFrame};
  if (startChan >= in.numChans() || startChan < 0)
    return {S::kError, "Input buffer ", b, " invalid start channel ", startChan};
  nFrames = nFrames < 0 ? in.numFrames() - startFrame : nFrames;
  if (nFrames <= 0 || nFrames > in.numFrames() - startFrame)
    return {S::kError, "Input buffer ", b, ": not enough frames"};
  nChans = nChans < 0 ? in.numChans() - startChan : nChans;
  if (nChans <= 0 || nChans > in.numChans() - startChan)
    return {S::kError, "Input buffer ", b, ": not enough channels"};
  return {S::kOk, ""};
}

// In-memory buffer, frames-major like the reference's (frame f, channel c at data[f * chans + c]),
// so samps(channel) is a strided view (cc/MemoryBufferAdaptor.hpp:24-26, 96-115).
// Channel-major floats [nch][n] into the channels [ch0, ch0 + nch) of a buffer (the `samps(c) <<= row` loops of
// nrt/NMFClient.hpp:281-282, 295-298, 321-326), block by block of frames: an interleaved destination is then written a few
// cache lines at a time instead of in nch strided passes over the whole buffer (8 channels x 32 components x 441 000 samples:
// 576 ms of strided passes).  Same values in the same places.
inline void scatterChannels(BufferAdaptor::Access& dst, index ch0, index nch, const float* src, index n)
{
  constexpr index kBlock = 256;
  for (index t0 = 0; t0 < n; t0 += kBlock)
  {
    const index len = std::min(kBlock, n - t0);
    for (index j = 0; j < nch; ++j) dst.samps(t0, len, ch0 + j) <<= VectorView<const float>(src + j * n + t0, len);
  }
}
// base pointer and frame stride of a buffer whose channels are interleaved in one array (frame-major, like
// MemoryBufferAdaptor, clients/common/MemoryBufferAdaptor.hpp:96-100); nullptr for any other layout
inline float* interleavedBase(BufferAdaptor::Access& b, index& frameStride)
{
  const index nch = b.numChans();
  if (nch < 1 || b.numFrames() < 1) return nullptr;
  auto v0 = b.samps(0);
  if (v0.stride < nch) return nullptr;
  for (index c = 1; c < nch; c = (c == nch - 1 ? nch : nch - 1)) // the second and the last channel
  {
    auto v = b.samps(c);
    if (v.data() != v0.data() + c || v.stride != v0.stride) return nullptr;
  }
  frameStride = v0.stride;
  return v0.data();
}

inline const float* interleavedBase(const BufferAdaptor::ReadAccess& b, index& frameStride)
{
  const index nch = b.numChans();
  if (nch < 1 || b.numFrames() < 1) return nullptr;
  auto v0 = b.samps(0);
  if (v0.stride < nch) return nullptr;
  for (index c = 1; c < nch; c = (c == nch - 1 ? nch : nch - 1))
  {
    auto v = b.samps(c);
    if (v.data() != v0.data() + c || v.stride != v0.stride) return nullptr;
  }
  frameStride = v0.stride;
  return v0.data();
}
// the other direction: the channels [ch0, ch0 + nch) of a buffer into channel-major floats [nch][n], block by block of frames
inline void gatherChannels(const BufferAdaptor::ReadAccess& src, index ch0, index nch, float* dst, index n)
{
  constexpr index kBlock = 256;
  for (index t0 = 0; t0 < n; t0 += kBlock)
  {
    const index len = std::min(kBlock, n - t0);
    for (index j = 0; j < nch; ++j) VectorView<float>(dst + j * n + t0, len) <<= src.samps(t0, len, ch0 + j);
  }
}

class MemoryBufferAdaptor : public BufferAdaptor
{
public:
  MemoryBufferAdaptor(index chans, index frames, double sampleRate = 44100)
      : mData(static_cast<size_t>(frames * chans), 0.f), mFrames(frames), mChans(chans), mSampleRate(sampleRate)
  {}
  // deep copy of any adaptor: the thread-isolation step of cc/FluidNRTClientWrapper.hpp:1045-1046
  explicit MemoryBufferAdaptor(const std::shared_ptr<BufferAdaptor>& other, bool contents = true) { rebind(other, contents); }
  explicit MemoryBufferAdaptor(const std::shared_ptr<const BufferAdaptor>& other) { copyFrom(other.get(), true); }

  // the copy taken again from `other` into the storage this object already has (the job layer keeps the copies of one job
  // for the next: a fresh 451 MB resynthesis buffer per job is ~100 ms of page faults).  contents == false: shape, sample
  // rate and flags only -- for a buffer the client only ever writes (it resizes it and fills every sample)
  void rebind(const std::shared_ptr<BufferAdaptor>& other, bool contents = true)
  {
    copyFrom(other.get(), contents);
    mOrigin = other;
  }
  const std::shared_ptr<BufferAdaptor>& origin() const { return mOrigin; }

  // cc/MemoryBufferAdaptor.hpp:51-67
  void copyToOrigin(Result& r)
  {
    mCacheTrusted = r.status() == Result::Status::kOk; // (what this copy holds is what its origin holds only behind a clean job)
    if (!mWrite || !mOrigin) return;
    BufferAdaptor::Access dst(mOrigin.get());
    if (!dst.exists()) return;
    if (numChans() != dst.numChans() || numFrames() != dst.numFrames())
      r = dst.resize(numFrames(), numChans(), mSampleRate);
    if (r.ok() && dst.valid())
    {
      index  stride = 0;
      float* base = interleavedBase(dst, stride);
      if (base && stride == numChans())
        std::memcpy(base, mData.data(), sizeof(float) * static_cast<size_t>(numFrames() * numChans()));
      else
      {
        constexpr index kBlock = 256; // (as in copyFrom: block by block of frames)
        for (index t0 = 0; t0 < numFrames(); t0 += kBlock)
        {
          const index len = std::min(kBlock, numFrames() - t0);
          for (index c = 0; c < numChans(); ++c) dst.samps(t0, len, c) <<= VectorView<const float>(samps(t0, len, c));
        }
      }
    }
  }

  std::string asString() const override { return ""; }
  float*      raw() { return mData.data(); }
  const float* raw() const { return mData.data(); }

private:
  void copyFrom(const BufferAdaptor* other, bool contents)
  {
    BufferAdaptor::ReadAccess src(other);
    mExists = src.exists();
    mValid = src.valid();
    mSampleRate = src.sampleRate();
    mWrite = false;
    if (mValid)
    {
      const bool sameShape = mFrames == src.numFrames() && mChans == src.numChans() &&
                             mData.size() == static_cast<size_t>(mFrames * mChans);
      mFrames = src.numFrames();
      mChans = src.numChans();
      // block by block of frames (cc/MemoryBufferAdaptor.hpp:125-131 copies channel by channel: on two interleaved buffers
      // that is numChans strided passes over both)
      index        stride = 0;
      const float* base = contents ? interleavedBase(src, stride) : nullptr;
      if (!contents) // shape only (a buffer the job writes and never reads): zeros unless the cached copy already has the shape
      {
        // (a cached copy of the same shape holds what the last job copied back to this very origin -- the host's own content --
        //  IF that job ended kOk: every client writes every sample of an output it sized on success.  After a warning, an
        //  error or a cancellation the cache may hold samples nobody wrote this time, and a later job that takes a warning
        //  path itself would hand them to the host: then it starts from zeros.  ADVICE r04.)
        if (!sameShape || !mCacheTrusted) mData.assign(static_cast<size_t>(mFrames * mChans), 0.0f);
        mCacheTrusted = false; // until the job that starts now has copied back with kOk
      }
      else if (base && stride == mChans) // the same layout as this one: one pass, allocation and copy together
        mData.assign(base, base + mFrames * mChans);
      else
      {
        mData.resize(static_cast<size_t>(mFrames * mChans));
        constexpr index kBlock = 256;
        for (index t0 = 0; t0 < mFrames; t0 += kBlock)
        {
          const index len = std::min(kBlock, mFrames - t0);
          for (index c = 0; c < mChans; ++c) samps(t0, len, c) <<= src.samps(t0, len, c);
        }
      }
    }
  }

  bool   acquire() const override { return true; }
  void   release() const override {}
  bool   valid() const override { return mValid; }
  bool   exists() const override { return mExists; }
  Result resize(index frames, index channels, double sampleRate) override
  {
    mWrite = true;
    mSampleRate = sampleRate;
    // (FluidTensor::resize, data/FluidTensor.hpp: the container is resized, not cleared -- a buffer that already has the
    //  shape keeps its memory untouched; zero-filling 451 MB of resynthesis buffer per job was 40 ms of an 8-channel job.)
    // A CHANGE of shape starts from zeros: the old samples would sit at other (frame, channel) positions, and a client
    // that resizes an output without writing every sample of it (a warning path, fewer frames written than sized) must not
    // hand the previous job's data back to the host.  Same for a shape-only copy, copyFrom(contents = false).
    if (frames != mFrames || channels != mChans) mData.assign(static_cast<size_t>(frames * channels), 0.0f);
    mFrames = frames;
    mChans = channels;
    return {};
  }
  VectorView<float>       samps(index c) override { return {mData.data() + c, mFrames, mChans}; }
  VectorView<float>       samps(index off, index n, index c) override { return {mData.data() + off * mChans + c, n, mChans}; }
  VectorView<const float> samps(index c) const override { return {mData.data() + c, mFrames, mChans}; }
  VectorView<const float> samps(index off, index n, index c) const override
  {
    return {mData.data() + off * mChans + c, n, mChans};
  }
  MatrixView<float>       allFrames() override { return MatrixView<float>(mData.data(), mFrames, mChans).transpose(); }
  MatrixView<const float> allFrames() const override
  {
    return MatrixView<const float>(mData.data(), mFrames, mChans).transpose();
  }
  index  numFrames() const override { return mFrames; }
  index  numChans() const override { return mChans; }
  double sampleRate() const override { return mSampleRate; }
  void   refresh() override { mWrite = true; }

  std::shared_ptr<BufferAdaptor> mOrigin;
  std::vector<float>             mData;
  index                          mFrames{0}, mChans{0};
  double                         mSampleRate{44100};
  bool                           mValid{true}, mExists{true}, mWrite{false};
  bool                           mCacheTrusted{false}; // the samples are those of the origin (see copyFrom, contents = false)
};

} // namespace fluhip
