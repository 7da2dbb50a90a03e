// knobs.hpp — the ONE gate in front of every experiment / diagnostic environment variable of the library.
//
// Rounds 1-3 grew some thirty variables that select kernel variants, launch shapes and bookkeeping paths (DEMI_K2_MODE,
// DEMI_JIT_K1_HOT, DEMI_DPOR_HOST_BOOKKEEPING, ...) or print diagnostics (DEMI_K1_VERBOSE, DEMI_DPOR_TIMING, ...).  Every
// setting gives the same verdicts - the test suites run the variants against each other - but a host process that merely
// INHERITS one of them (a JVM started from a developer's shell) would silently run another, usually slower, engine.  So the
// library reads them only when DEMI_EXPERIMENT=1 is set as well; without it `knob()` answers "unset" for every one of them.
// What is read unconditionally is listed in include/demi_gpu.h ("Environment"): DEMI_HIPRTC_LIB and DEMI_RCCL_LIB, which name
// the libraries to dlopen and select no engine.
#pragma once
#include <cstdlib>

namespace demi_host {
inline const char* knob(const char* name) {
  const char* e = getenv("DEMI_EXPERIMENT");       // (read every time: a test process switches it on before it sets a knob)
  return (e && e[0] == '1' && e[1] == 0) ? getenv(name) : nullptr;
}
}  // namespace demi_host
