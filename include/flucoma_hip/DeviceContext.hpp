// DeviceContext.hpp -- the library context a host-side client keeps for its lifetime (one per client, like the
// per-client algorithm objects of the reference: the context's cached device blocks, window / twiddle tables and
// loaded code objects outlive a job).  There is no CPU path: a client whose context cannot be created returns kError.
#pragma once

#include "../flucoma_hip.h"
#include "Types.hpp"

#include <string>

namespace fluhip {

class DeviceContext
{
public:
  DeviceContext() = default;
  ~DeviceContext()
  {
    if (mCtx) fluhip_ctx_destroy(mCtx);
  }
  DeviceContext(const DeviceContext&) = delete;
  DeviceContext& operator=(const DeviceContext&) = delete;

  // the context on `device`, (re)created when the job asks for another GPU than the last one did
  Result ensure(int device)
  {
    if (mCtx && mDevice == device) return {};
    if (mCtx) fluhip_ctx_destroy(mCtx);
    mCtx = nullptr;
    if (fluhip_ctx_create(device, &mCtx) != FLUHIP_OK)
    {
      mCtx = nullptr;
      return {Result::Status::kError, "MI355X path unavailable: could not create a HIP context on device ", device};
    }
    mDevice = device;
    return {};
  }
  fluhip_ctx* get() const { return mCtx; }
  // the library's status codes are Result::Status (include/flucoma_hip.h: fluhip_status), its message the last error
  Result result(int rc) const
  {
    if (rc == FLUHIP_OK) return {};
    return {static_cast<Result::Status>(rc), std::string(fluhip_last_error(mCtx))};
  }

private:
  fluhip_ctx* mCtx{nullptr};
  int         mDevice{-1};
};

} // namespace fluhip
