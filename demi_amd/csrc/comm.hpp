// comm.hpp — the exchange steps of the multi-GPU paths behind one small interface (host code, no device code).
//
// The reference is a single JVM with one execution at a time (Instrumenter.scala:1289-1296); sharding is new here
// (SURVEY 8e): K1 schedule ranges, K2 frontier candidates and K3 rounds are split over the GPUs of one node, one process
// per GPU, and what is exchanged is always "every rank's fixed-size block to every rank" - an all-gather:
//   * RcclComm: ncclAllGather over xGMI, on device buffers, on the caller's stream.  RCCL is resolved with dlopen next
//     to the HIP runtime already in the process (PyTorch-ROCm ships its own librccl: one RCCL per process), so the
//     library has no link-time dependency on it and single-GPU use never loads it.
//   * LocalComm: the ranks are threads of one process and the buffers host memory (the CPU test harness of the sharded
//     bookkeeping; also what world = 1 degenerates to).
#pragma once

#include <dlfcn.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace demi_comm {

struct Comm {
  int rank = 0, world = 1;
  virtual ~Comm() {}
  // every rank contributes `bytes` from `send`; `recv` receives world * bytes, rank r's block at r * bytes.
  // Device pointers + stream for RcclComm, host pointers for LocalComm.  Returns 0 or a negative demi_status.
  virtual int allgather(const void* send, void* recv, size_t bytes, void* stream) = 0;
  virtual const char* last_error() const { return ""; }
};

// ------------------------------------------------------------------ RCCL through dlopen
struct RcclId { char internal[128]; };       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value

struct RcclApi {
  void* lib = nullptr;
  int (*get_unique_id)(RcclId*) = nullptr;
  int (*comm_init_rank)(void**, int, RcclId, int) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, void*, void*) = nullptr;
  const char* (*get_error_string)(int) = nullptr;

  bool open(std::string& err, const void* hip_symbol) {
    if (lib) return true;
    std::vector<std::string> cands;
    if (const char* only = getenv("DEMI_RCCL_LIB")) cands.push_back(only);
    cands.push_back("librccl.so");               // already mapped when torch.distributed is in the process
    cands.push_back("librccl.so.1");
    Dl_info info;
    if (hip_symbol && dladdr(hip_symbol, &info) && info.dli_fname) {
      std::string p(info.dli_fname);
      const size_t slash = p.rfind('/');
      if (slash != std::string::npos) cands.insert(cands.begin() + (getenv("DEMI_RCCL_LIB") ? 1 : 0), p.substr(0, slash) + "/librccl.so");
    }
    for (const std::string& c : cands) {
      lib = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { err = "RCCL not found (dlopen librccl.so failed)"; return false; }
    get_unique_id = reinterpret_cast<decltype(get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
    comm_init_rank = reinterpret_cast<decltype(comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
    comm_destroy = reinterpret_cast<decltype(comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
    all_gather = reinterpret_cast<decltype(all_gather)>(dlsym(lib, "ncclAllGather"));
    get_error_string = reinterpret_cast<decltype(get_error_string)>(dlsym(lib, "ncclGetErrorString"));
    if (!get_unique_id || !comm_init_rank || !comm_destroy || !all_gather) {
      err = "RCCL entry points missing";
      lib = nullptr;
      return false;
    }
    return true;
  }
};

struct RcclComm : Comm {
  RcclApi* api;
  void* comm = nullptr;
  std::string err;
  explicit RcclComm(RcclApi* a) : api(a) {}
  ~RcclComm() override { if (comm) (void)api->comm_destroy(comm); }
  int init(const RcclId& id, int rank_, int world_) {
    rank = rank_; world = world_;
    const int rc = api->comm_init_rank(&comm, world_, id, rank_);
    if (rc != 0) { err = std::string("ncclCommInitRank: ") + (api->get_error_string ? api->get_error_string(rc) : "error"); comm = nullptr; return -6; }
    return 0;
  }
  int allgather(const void* send, void* recv, size_t bytes, void* stream) override {
    const int rc = api->all_gather(send, recv, bytes, /* ncclInt8 */ 0, comm, stream);
    if (rc != 0) { err = std::string("ncclAllGather: ") + (api->get_error_string ? api->get_error_string(rc) : "error"); return -6; }
    return 0;
  }
  const char* last_error() const override { return err.c_str(); }
};

// ------------------------------------------------------------------ ranks as threads of one process (tests)
struct LocalGroup {
  int world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  std::vector<const void*> send;
  explicit LocalGroup(int w) : world(w), send((size_t)w, nullptr) {}
};

struct LocalComm : Comm {
  LocalGroup* g;
  LocalComm(LocalGroup* group, int rank_) : g(group) { rank = rank_; world = group->world; }
  int allgather(const void* send, void* recv, size_t bytes, void*) override {
    std::unique_lock<std::mutex> lk(g->mu);
    g->send[(size_t)rank] = send;
    const uint64_t gen = g->generation;
    if (++g->arrived == g->world) { g->arrived = 0; g->generation++; g->cv.notify_all(); }
    else g->cv.wait(lk, [&] { return g->generation != gen; });
    // everyone has published its block: copy them out, then meet again before the blocks may be reused
    std::vector<const void*> src = g->send;
    lk.unlock();
    for (int r = 0; r < world; r++) memcpy(static_cast<unsigned char*>(recv) + (size_t)r * bytes, src[(size_t)r], bytes);
    lk.lock();
    const uint64_t gen2 = g->generation;
    if (++g->arrived == g->world) { g->arrived = 0; g->generation++; g->cv.notify_all(); }
    else g->cv.wait(lk, [&] { return g->generation != gen2; });
    return 0;
  }
};

}  // namespace demi_comm
