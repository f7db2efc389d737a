// fluhip_env.h -- the experiment switches of DESIGN section 6b.
//
// Every FLUHIP_* environment variable that selects a kernel form, a schedule or a test aid is read through
// ab_getenv(), which only looks at the environment in a build with -DFLUHIP_AB_SWITCHES (flucoma-core_amd/lib_ab/
// libflucoma_hip_ab.so: what tests/test_gpu_variants.py and the tools/ A/B scripts load through FLUHIP_LIB).  The
// production library answers "unset" for all of them -- it has one schedule per shape, the planner's -- and reads
// the environment only for its two allocator debugging aids (FLUHIP_NO_POOL, FLUHIP_CANARY; api_internal.h).
#pragma once

#include <cstdlib>

namespace fluhip {

inline const char* ab_getenv(const char* name)
{
#ifdef FLUHIP_AB_SWITCHES
  return std::getenv(name);
#else
  (void) name;
  return nullptr;
#endif
}

constexpr bool kAbSwitches =
#ifdef FLUHIP_AB_SWITCHES
    true;
#else
    false;
#endif

} // namespace fluhip
