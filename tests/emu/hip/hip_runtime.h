// <hip/hip_runtime.h> of the wave64 emulator - TEST INFRASTRUCTURE (see ../wave64_emu.hpp).
// The host API the library uses, over plain host memory: "device" pointers are host pointers, a launch runs to completion
// inside the call, streams and events order nothing (everything already is in order).
#pragma once

// every standard header the emulated sources use, BEFORE the keyword games at the end of this file
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>
#include <dlfcn.h>

#include "../wave64_emu.hpp"

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotFound = 500, hipErrorLaunchFailure = 719 };
typedef void* hipStream_t;
struct w64_event { double t_ms; };
typedef w64_event* hipEvent_t;
typedef void* hipModule_t;
typedef void* hipFunction_t;      // -> a w64_kernel_entry of a module compiled by the emulator's hiprtc (tests/emu/hiprtc_emu.cpp)
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum : unsigned { hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
struct hipDeviceProp_t {
  char name[256];
  char gcnArchName[256];
  size_t totalGlobalMem;
  int multiProcessorCount, clockRate, warpSize;
  size_t sharedMemPerBlock;
};

extern "C" {
// w64rt.cpp
void w64_launch(dim3 grid, dim3 block, size_t lds_bytes, void (*body)(void*), void* arg);
int w64_num_cu();
int w64_occupancy();
}

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof *p);
  snprintf(p->name, sizeof p->name, "wave64 emulator");
  snprintf(p->gcnArchName, sizeof p->gcnArchName, "w64emu");
  p->totalGlobalMem = (size_t)8 << 30; p->multiProcessorCount = w64_num_cu(); p->clockRate = 1000000; p->warpSize = 64;
  p->sharedMemPerBlock = 160 * 1024;
  return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)2 << 30; *total_b = (size_t)8 << 30; return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) {
  void* q = nullptr;
  if (posix_memalign(&q, 256, n ? n : 1) != 0) return hipErrorOutOfMemory;
  memset(q, 0xA5, n);          // (fresh device memory holds no promises either)
  *p = static_cast<T*>(q);
  return hipSuccess;
}
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
#define hipStreamNonBlocking 1u
// (every launch of the emulator is synchronous: a stream is a name)
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static char names[64]; static int next = 0; *s = &names[next++ & 63]; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new w64_event{0.0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
  e->t_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <class F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t lds) {
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  *n = w64_occupancy();
  return hipSuccess;
}
static inline hipError_t hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(int* n, hipFunction_t, int, size_t lds) {
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  *n = w64_occupancy();
  return hipSuccess;
}

// ---- modules: the "code object" of the emulator's hiprtc is the path of a shared object, a kernel an entry that takes the
// parameter array of hipModuleLaunchKernel
typedef void (*w64_kernel_entry)(void** params);
static inline hipError_t hipModuleLoadData(hipModule_t* m, const void* image) {
  *m = dlopen(static_cast<const char*>(image), RTLD_NOW | RTLD_LOCAL);
  if (!*m) { fprintf(stderr, "[w64] %s\n", dlerror()); return hipErrorInvalidValue; }
  return hipSuccess;
}
static inline hipError_t hipModuleUnload(hipModule_t m) { if (m) dlclose(m); return hipSuccess; }
static inline hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char* name) {
  *f = dlsym(m, name);
  return *f ? hipSuccess : hipErrorNotFound;
}
struct w64_module_call { w64_kernel_entry fn; void** params; };
static inline void w64_module_body(void* p) { const w64_module_call* c = static_cast<const w64_module_call*>(p); c->fn(c->params); }
static inline hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                                               unsigned lds, hipStream_t, void** params, void**) {
  w64_module_call c{reinterpret_cast<w64_kernel_entry>(f), params};
  w64_launch(dim3(gx, gy, gz), dim3(bx, by, bz), lds, &w64_module_body, &c);
  return hipSuccess;
}

// ---- launches of the kernels compiled into the library
template <class F> static inline void w64_body_thunk(void* p) { (*static_cast<F*>(p))(); }
template <class F> static inline void w64_launch_lambda(dim3 grid, dim3 block, size_t lds, F f) { w64_launch(grid, block, lds, &w64_body_thunk<F>, &f); }
#define hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, ...) \
  w64_launch_lambda(dim3(GRID), dim3(BLOCK), (LDS), [&]() { (KERNEL)(__VA_ARGS__); })

// ---- the keyword games (wave64_emu.hpp, last paragraph): `asm volatile ( ... );` -> `;`
#define asm
#define volatile(...)
