// Types.hpp -- host-side value types of the BufNMF drop-in, restated in plain C++17 so a host
// wrapper can be built without Eigen / HISSTools / foonathan-memory.
//
// Mirrors (semantics, not text) of the reference types that appear in the path's signatures:
//   fluid::index                     include/flucoma/data/FluidIndex.hpp:9
//   FluidTensorView<T,N> (N = 1, 2)  include/flucoma/data/FluidTensor.hpp:447-741 (pointer + extents + strides,
//                                    row-major, transposed views, element-wise converting copy `<<=`)
//   client::Result                   include/flucoma/clients/common/Result.hpp:21-81
//   FluidTask                        include/flucoma/clients/common/FluidTask.hpp:17-49
//   client::FluidContext             include/flucoma/clients/common/FluidContext.hpp:23-52
//   client::FFTParams                include/flucoma/clients/common/ParameterTypes.hpp:260-439 (arithmetic only)
#pragma once
#include <vector>

#include <atomic>
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <sstream>
#include <string>
#include <utility>

namespace fluhip {

using index = std::ptrdiff_t; // data/FluidIndex.hpp:9

// ---- strided non-owning views --------------------------------------------------------------
template <typename T>
struct VectorView
{
  T*    ptr{nullptr};
  index n{0};
  index stride{1};

  VectorView() = default;
  VectorView(T* p, index size, index st = 1) : ptr(p), n(size), stride(st) {}
  template <typename U, typename = std::enable_if_t<std::is_same_v<const U, T>>>
  VectorView(const VectorView<U>& o) : ptr(o.ptr), n(o.n), stride(o.stride)
  {}

  index size() const { return n; }
  T&    operator()(index i) const { assert(i >= 0 && i < n); return ptr[i * stride]; }
  T&    operator[](index i) const { return (*this)(i); }
  T*    data() const { return ptr; }
  bool  contiguous() const { return stride == 1; }

  // element-wise converting copy, the `<<=` of data/FluidTensor.hpp:203-212
  template <typename U>
  const VectorView& operator<<=(const VectorView<U>& src) const
  {
    assert(src.size() == n && "mismatched extents in converting copy");
    for (index i = 0; i < n; ++i) (*this)(i) = static_cast<std::remove_const_t<T>>(src(i));
    return *this;
  }
  template <typename F>
  void apply(F&& f) const
  {
    for (index i = 0; i < n; ++i) f((*this)(i));
  }
};

template <typename T>
struct MatrixView
{
  T*    ptr{nullptr};
  index nrows{0}, ncols{0};
  index rstride{0}, cstride{1};

  MatrixView() = default;
  MatrixView(T* p, index r, index c) : ptr(p), nrows(r), ncols(c), rstride(c), cstride(1) {}
  MatrixView(T* p, index r, index c, index rs, index cs) : ptr(p), nrows(r), ncols(c), rstride(rs), cstride(cs) {}

  index         rows() const { return nrows; }
  index         cols() const { return ncols; }
  index         extent(int d) const { return d == 0 ? nrows : ncols; }
  T&            operator()(index r, index c) const { return ptr[r * rstride + c * cstride]; }
  VectorView<T> row(index r) const { return {ptr + r * rstride, ncols, cstride}; }
  VectorView<T> col(index c) const { return {ptr + c * cstride, nrows, rstride}; }
  MatrixView    transpose() const { return {ptr, ncols, nrows, cstride, rstride}; } // data/FluidTensor_Support.hpp:386-393
  // {data, rows, cols, row stride, col stride}: the field order of fluhip_matrix_view (include/flucoma_hip.h), so a
  // MatrixView<double> -- transposed or not -- goes to fluhip_nmf_process_views_f64 as it is
  template <typename V>
  V as() const { return V{const_cast<std::remove_const_t<T>*>(ptr), nrows, ncols, rstride, cstride}; }
  T*            data() const { return ptr; }
};

// ---- Result ----------------------------------------------------------------------------------
class Result
{
public:
  enum class Status { kOk, kWarning, kError, kCancelled }; // cc/Result.hpp:24; == fluhip_status

  Result() = default;
  Result(Status s, std::string msg) : mStatus(s), mMsg(std::move(msg)) {}
  template <typename... Args>
  Result(Status s, Args&&... args) : mStatus(s)
  {
    std::ostringstream os;
    (void) std::initializer_list<int>{((os << args), 0)...};
    mMsg = os.str();
  }

  bool               ok() const noexcept { return mStatus == Status::kOk; }
  Status             status() const noexcept { return mStatus; }
  const std::string& message() const noexcept { return mMsg; }
  void               set(Status s) noexcept { mStatus = s; }
  template <typename... Ts>
  void addMessage(Ts&&... args)
  {
    std::ostringstream os;
    (void) std::initializer_list<int>{((os << args), 0)...};
    mMsg += os.str();
  }
  void reset()
  {
    mStatus = Status::kOk;
    mMsg.clear();
  }

private:
  Status      mStatus{Status::kOk};
  std::string mMsg;
};

// ---- FluidTask / FluidContext ------------------------------------------------------------------
// Same progress arithmetic as cc/FluidTask.hpp:22-34.  The cancel flag is an atomic here: the
// reference writes a plain bool from the host thread and reads it on the worker.
class FluidTask
{
public:
  bool processUpdate(double samplesDone, double taskLength)
  {
    mProgress = (samplesDone / (taskLength * mTotalIterations)) + (mIteration / mTotalIterations);
    return !mCancel.load(std::memory_order_relaxed);
  }
  bool iterationUpdate(double iterationsDone, double totalIterations)
  {
    mIteration = iterationsDone;
    mTotalIterations = totalIterations;
    return !mCancel.load(std::memory_order_relaxed);
  }
  void   cancel() { mCancel = true; }
  void   reset() { mCancel = false; }
  double progress() const { return mProgress; }
  bool   cancelled() const { return mCancel; }

private:
  std::atomic<double> mProgress{0.0};
  std::atomic<bool>   mCancel{false};
  double              mTotalIterations{1};
  double              mIteration{0};
};

class FluidContext
{
public:
  FluidContext() = default;
  explicit FluidContext(FluidTask& t) : mTask(&t) {}
  FluidContext(FluidTask& t, int device) : mTask(&t), mDevice(device) {}
  FluidTask* task() const { return mTask; }
  void       task(FluidTask* t) { mTask = t; }
  int        device() const { return mDevice; } // which GPU the job runs on (no reference analogue)
  void       device(int d) { mDevice = d; }
  // several GPUs for one job: the channels of a multichannel BufNMF are independent factorisations
  // (clients/nrt/NMFClient.hpp:233) and are dealt round-robin over this list, one host thread per entry.
  // Empty (the default): everything on device().  An id may be listed twice (two contexts on one GPU).
  const std::vector<int>& devices() const { return mDevices; }
  void                    devices(std::vector<int> d) { mDevices = std::move(d); }

private:
  FluidTask*       mTask{nullptr};
  int              mDevice{0};
  std::vector<int> mDevices;
};

// ---- FFTParams (arithmetic of cc/ParameterTypes.hpp:295-312) -------------------------------------
struct FFTParams
{
  index win{1024}, hop{-1}, fft{-1};

  constexpr FFTParams() = default;
  constexpr FFTParams(index w, index h, index f) : win(w), hop(h), fft(f) {}

  index winSize() const noexcept { return win; }
  index hopSize() const noexcept { return hop > 0 ? hop : win >> 1; }
  index fftSize() const noexcept
  {
    if (fft >= 0) return fft;
    index p = 1;
    while (p < win) p <<= 1; // nextPow2(win, up)
    return p;
  }
  index frameSize() const noexcept { return (fftSize() >> 1) + 1; }
};

} // namespace fluhip
