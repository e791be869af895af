// meas_env.h -- the one place the library decides which environment variables it reads.
#pragma once
#include <cstdlib>

// Environment variables.  The PRODUCT library reads three, each a deployment matter and named in include/beatrice_batch.h:
// BEATRICE_HIP_DEBUG (print HIP errors), BEATRICE_HIP_CUMASK (CU masks of the stage-pipelining streams), BEATRICE_HIP_HOP_GRAPH (the
// 1-stream calls replayed as hipGraphs).  Every A/B switch and trace of the measurements (profiles/r0*_notes.md) goes through meas_env()
// and exists only in MEASUREMENT BUILDS (tools/debug/build_variant.sh <name> -DBEATRICE_HIP_MEASUREMENT_BUILD): in the product a stray
// environment variable cannot select an untested configuration.
namespace bhip {
inline const char* meas_env(const char* name) {
#ifdef BEATRICE_HIP_MEASUREMENT_BUILD
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
}  // namespace bhip

