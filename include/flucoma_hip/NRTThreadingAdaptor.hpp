// NRTThreadingAdaptor.hpp -- job layer of the buffer-processing drop-ins (BufNMF; BufSTFT, BufNMFSeed, BufMFCC and
// BufMelBands in their own headers).
//
// Mirrors client::NRTThreadingAdaptor + ThreadedTask,
// include/flucoma/clients/common/FluidNRTClientWrapper.hpp:788-1132: a queue of parameter sets,
// synchronous execution on the caller thread or one std::thread per job, deep copies of every
// buffer parameter before the worker starts (:1045-1046) and copy-back to the host's buffers from
// the thread that polls checkProgress (:1089-1100), progress and cancellation through FluidTask.
// Generic over the client, like the reference's template: a client names its parameter set `ParamSetViewType`, and the
// parameter set tells the adaptor which of its members are buffers --
//   template <class In, class Out> void forEachBuffer(In&& in, Out&& out)
// calls in(std::shared_ptr<const BufferAdaptor>&) for every InputBufferParam and out(std::shared_ptr<BufferAdaptor>&)
// for every BufferParam (the reference walks its parameter tuple for the same two types, :1045-1046, :1093-1097).
#pragma once

#include "NMFClient.hpp"
#include "ParamDescriptors.hpp"

#include <deque>
#include <functional>
#include <future>
#include <thread>

namespace fluhip {

enum ProcessState { kNoProcess, kProcessing, kDone, kDoneStillProcessing }; // cc/FluidBaseClient.hpp:32
struct ControlChannel { index count{0}; index size{-1}; index max{-1}; };          // cc/FluidBaseClient.hpp:34-38

namespace impl {
// parameter sets with write-only buffers offer forEachBuffer(in, out, outOnly); the others forEachBuffer(in, out)
template <class P, class In, class Out, class OutOnly>
auto forEachBufferImpl(P& p, In&& in, Out&& out, OutOnly&& outOnly, int) -> decltype(p.forEachBuffer(in, out, outOnly), void())
{
  p.forEachBuffer(in, out, outOnly);
}
template <class P, class In, class Out, class OutOnly>
void forEachBufferImpl(P& p, In&& in, Out&& out, OutOnly&&, long)
{
  p.forEachBuffer(in, out);
}
} // namespace impl
template <class P, class In, class Out, class OutOnly>
void forEachBufferOf(P& p, In&& in, Out&& out, OutOnly&& outOnly)
{
  impl::forEachBufferImpl(p, in, out, outOnly, 0);
}

template <class NRTClient>
class NRTThreadingAdaptor
{
public:
  using Client = NRTClient;
  using ParamSetType = typename NRTClient::ParamSetViewType;

  // cc/FluidNRTClientWrapper.hpp:801-804: the wrapped client's table, for the host wrapper that builds its attributes from it
  static constexpr auto getParameterDescriptors() { return NRTClient::getParameterDescriptors(); }
  // :806-809 (none of the mirrored clients defines a message: an empty list)
  static constexpr ParamDescriptorList getMessageDescriptors() { return {nullptr, 0}; }

  // :811-822: an offline object has no audio or control connections; its inputs and outputs are its buffer parameters
  index          audioChannelsIn() const noexcept { return 0; }
  index          audioChannelsOut() const noexcept { return 0; }
  index          controlChannelsIn() const noexcept { return 0; }
  ControlChannel controlChannelsOut() const noexcept { return {0, 0}; }
  index          audioBuffersIn() const noexcept { return countKind(ParamKind::kInputBuffer); }
  index          audioBuffersOut() const noexcept { return countKind(ParamKind::kBuffer); }

  // ONE client for the adaptor's lifetime, handed to every task (:831 mClient{new NRTClient{mHostParams, c}}, :883): the
  // client's device context -- and with it the cached device blocks and loaded code objects -- outlives a job
  explicit NRTThreadingAdaptor(ParamSetType& p, FluidContext c = {})
      : mHostParams(p), mContext(c), mClient(std::make_shared<Client>(mHostParams, mContext))
  {}
  ~NRTThreadingAdaptor()
  {
    mQueue.clear();
    if (mTask)
    {
      mTask->cancel();
      mTask->join();
    }
  }

  void   setParams(ParamSetType& p) { mHostParams = p; }
  Result enqueue(ParamSetType& p, std::function<void()> callback = {})
  {
    if (mTask && (mSynchronous || !mQueueEnabled)) return {Result::Status::kError, "already processing"};
    mQueue.push_back({p, std::move(callback)});
    return {};
  }

  Result process()
  {
    if (mTask && (mSynchronous || !mQueueEnabled)) return {Result::Status::kError, "already processing"};
    if (mTask) return {};
    if (mQueue.empty()) return {Result::Status::kWarning, "Process() called on empty queue"};
    if (mSynchronous) mSynchronousDone = false;
    mTask = std::make_unique<ThreadedTask>(mClient, mQueue.front(), mContext, mSynchronous, mCopyCache);
    mQueue.pop_front();
    Result result;
    if (mSynchronous)
    {
      result = mTask->result();
      mTask.reset();
      mSynchronousDone = true;
    }
    return result;
  }

  ProcessState checkProgress(Result& result)
  {
    if (!mTask) return kNoProcess;
    ProcessState state = mTask->checkProgress(result);
    if (state == kDone)
    {
      if (!mQueue.empty())
      {
        mTask = std::make_unique<ThreadedTask>(mClient, mQueue.front(), mContext, false, mCopyCache);
        mQueue.pop_front();
        state = kDoneStillProcessing;
      }
      else
        mTask.reset();
    }
    return state;
  }

  bool   synchronous() const { return mSynchronous; }
  void   setSynchronous(bool s) { mSynchronous = s; }
  void   setQueueEnabled(bool q) { mQueueEnabled = q; }
  double progress() { return mTask ? mTask->progress() : 0.0; }
  void   cancel()
  {
    mQueue.clear();
    if (mTask) mTask->cancel();
  }
  bool done() const { return mTask ? mTask->done() : (mSynchronous && mSynchronousDone); }
  ProcessState state() const { return mTask ? mTask->state() : kNoProcess; }

private:
  static constexpr index countKind(ParamKind k)
  {
    index n = 0;
    for (const ParamDescriptor& d : NRTClient::getParameterDescriptors()) n += d.kind == k ? 1 : 0;
    return n;
  }
  struct NRTJob
  {
    ParamSetType          params;
    std::function<void()> callback;
  };

  class ThreadedTask
  {
  public:
    ThreadedTask(std::shared_ptr<Client> client, NRTJob& job, const FluidContext& host, bool synchronous,
                 std::vector<std::shared_ptr<MemoryBufferAdaptor>>& cache)
        : mJob(job), mContext(mTaskState, host.device()), mClient(std::move(client)), mCache(cache)
    {
      mContext.devices(host.devices());
      mState = kProcessing;
      if (synchronous)
      {
        mClient->setParams(mJob.params); // :1038
        mResult = mClient->template process<float>(mContext);
        mState = kDone;
        mDetached = true;
        return;
      }
      // deep copies: the worker only ever touches MemoryBufferAdaptors
      // (a copy the adaptor still holds from its last job for the same host buffer is refilled instead of allocated anew;
      //  a parameter set may name buffers the client only writes -- forEachBuffer's optional third visitor -- and those
      //  copies take shape and flags only)
      auto isolate = [this](std::shared_ptr<BufferAdaptor>& b, bool contents) {
        if (!b) return;
        std::shared_ptr<MemoryBufferAdaptor> copy;
        for (auto& old : mCache)
          if (old && old->origin() == b && old.use_count() == 1) { copy = old; break; }
        if (copy) copy->rebind(b, contents);
        else copy = std::make_shared<MemoryBufferAdaptor>(b, contents);
        mOutputCopies.push_back(copy);
        b = copy;
      };
      forEachBufferOf(
          mJob.params,
          [this](std::shared_ptr<const BufferAdaptor>& b) {
            if (!b) return;
            mInputCopies.push_back(std::make_shared<MemoryBufferAdaptor>(b));
            b = mInputCopies.back();
          },
          [&](std::shared_ptr<BufferAdaptor>& b) { isolate(b, true); }, [&](std::shared_ptr<BufferAdaptor>& b) { isolate(b, false); });
      mCache = mOutputCopies; // what the next job may reuse
      mClient->setParams(mJob.params);
      mFuture = mPromise.get_future();
      mThread = std::thread([this] {
        Result r = mClient->template process<float>(mContext);
        mState = kDone;
        mPromise.set_value(r);
        if (mJob.callback) mJob.callback();
      });
    }
    ~ThreadedTask() { join(); }

    Result result() const { return mResult; }
    double progress() { return mTaskState.progress(); }
    void   cancel() { mTaskState.cancel(); }
    bool   done() const { return mState == kDone || mState == kDoneStillProcessing; }
    ProcessState state() const { return mState; }
    void   join()
    {
      if (mThread.joinable()) mThread.join();
    }

    ProcessState checkProgress(Result& result)
    {
      if (mDetached) { result = mResult; return kDone; }
      if (mState != kDone) return kProcessing;
      if (mFuture.valid())
      {
        mResult = mFuture.get();
        join();
        if (mResult.ok() || mResult.status() == Result::Status::kWarning)
        {
          // copy-back happens on the polling (host) thread, like :1089-1100
          for (auto& b : mOutputCopies) b->copyToOrigin(mResult);
        }
      }
      result = mResult;
      return kDone;
    }

  private:
    NRTJob                               mJob;
    FluidTask                            mTaskState;
    FluidContext                         mContext;
    std::shared_ptr<Client>              mClient;
    std::vector<std::shared_ptr<MemoryBufferAdaptor>> mInputCopies, mOutputCopies;
    std::vector<std::shared_ptr<MemoryBufferAdaptor>>& mCache;
    std::promise<Result>                 mPromise;
    std::future<Result>                  mFuture;
    std::thread                          mThread;
    Result                               mResult;
    std::atomic<ProcessState>            mState{kNoProcess};
    bool                                 mDetached{false};
  };

  ParamSetType                  mHostParams;
  FluidContext                  mContext;
  std::shared_ptr<Client>       mClient;
  std::vector<std::shared_ptr<MemoryBufferAdaptor>> mCopyCache; // the buffer copies of the last job (storage reused by the next)
  std::deque<NRTJob>            mQueue;
  std::unique_ptr<ThreadedTask> mTask;
  bool                          mSynchronous{false};
  bool                          mQueueEnabled{false};
  bool                          mSynchronousDone{false};
};

// FlucomaClients.cmake:101 (BufNMF ... CLASS NRTThreadedNMFClient), clients/nrt/NMFClient.hpp:341-342
using NRTThreadedNMFClient = NRTThreadingAdaptor<bufnmf::NMFClient>;

} // namespace fluhip
