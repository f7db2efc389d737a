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

#include <deque>
#include <functional>
#include <future>
#include <thread>

namespace fluhip {

enum ProcessState { kNoProcess, kProcessing, kDone, kDoneStillProcessing }; // cc/FluidBaseClient.hpp:32

template <class NRTClient>
class NRTThreadingAdaptor
{
public:
  using Client = NRTClient;
  using ParamSetType = typename NRTClient::ParamSetViewType;

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
    mTask = std::make_unique<ThreadedTask>(mClient, mQueue.front(), mContext, mSynchronous);
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
        mTask = std::make_unique<ThreadedTask>(mClient, mQueue.front(), mContext, false);
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
  struct NRTJob
  {
    ParamSetType          params;
    std::function<void()> callback;
  };

  class ThreadedTask
  {
  public:
    ThreadedTask(std::shared_ptr<Client> client, NRTJob& job, const FluidContext& host, bool synchronous)
        : mJob(job), mContext(mTaskState, host.device()), mClient(std::move(client))
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
      mJob.params.forEachBuffer(
          [this](std::shared_ptr<const BufferAdaptor>& b) {
            if (!b) return;
            mInputCopies.push_back(std::make_shared<MemoryBufferAdaptor>(b));
            b = mInputCopies.back();
          },
          [this](std::shared_ptr<BufferAdaptor>& b) {
            if (!b) return;
            mOutputCopies.push_back(std::make_shared<MemoryBufferAdaptor>(b));
            b = mOutputCopies.back();
          });
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
  std::deque<NRTJob>            mQueue;
  std::unique_ptr<ThreadedTask> mTask;
  bool                          mSynchronous{false};
  bool                          mQueueEnabled{false};
  bool                          mSynchronousDone{false};
};

// FlucomaClients.cmake:101 (BufNMF ... CLASS NRTThreadedNMFClient), clients/nrt/NMFClient.hpp:341-342
using NRTThreadedNMFClient = NRTThreadingAdaptor<bufnmf::NMFClient>;

} // namespace fluhip
