// w64rt.cpp - the runtime of the wave64 emulator (TEST INFRASTRUCTURE, see wave64_emu.hpp).
//
// A launch runs its workgroups on a few OS threads (W64_THREADS, default 4; global-memory atomics are real atomics, so
// workgroups interleave as they would on different CUs).  A workgroup is a set of fibers on ONE thread: the scheduler below
// resumes the running lanes of a wave one after the other until each waits at its next cross-lane operation - one "pass" is one
// lock-step interval of the wave - then checks that they all wait at the same operation of the same source line, and goes on to
// the next wave.  A wave waiting at a workgroup barrier is skipped until every wave of the workgroup waits there.
//
// Deposits of a rendezvous are double-buffered by the parity of the wave's operation count: a lane that has passed operation n
// can deposit for n + 1 before the slowest lane has read n, never for n + 2.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <atomic>
#include <thread>
#include <vector>

#include "wave64_emu.hpp"

#if !defined(__x86_64__)
#error "the fiber switch below is x86-64 System V"
#endif

namespace demi { alignas(16) thread_local unsigned char smem[160 * 1024]; }
// The users see `extern thread_local`, so g++ makes them call the variable's "TLS init function" when that weak symbol
// resolves - and inside a shared object it tests the PLT entry, which is never null.  Give it something to call.
extern "C" void w64_smem_tls_init() asm("_ZTHN4demi4smemE");
extern "C" void w64_smem_tls_init() {}

extern "C" void w64_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl w64_switch
.type w64_switch,@function
w64_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size w64_switch,.-w64_switch
)");

namespace {

constexpr size_t STACK_BYTES = 512 * 1024;

struct Wave;
struct Lane {
  w64_item item;
  void* sp = nullptr;
  unsigned char* stack = nullptr;
  Wave* wave = nullptr;
  unsigned lane = 0;
  bool done = false, cleared = false;
  int wait_kind = 0, wait_line = 0;
};
struct Wave {
  uint64_t val[2][64];
  uint64_t seq = 0;           // operations completed by the slowest lane
  uint64_t lane_seq[64];      // operations each lane has arrived at
  Lane* lanes[64];
  unsigned n = 0;
  bool at_barrier = false, done = false;
  uint64_t shuffle_state = 0;
};
struct Group {
  std::vector<Lane> lanes;
  std::vector<Wave> waves;
  void* sched_sp = nullptr;
  void (*body)(void*) = nullptr;
  void* arg = nullptr;
};
thread_local Group* g_group = nullptr;
bool g_reverse = false;
uint64_t g_shuffle = 0;         // W64_LANE_ORDER=shuffle[:seed]
uint64_t g_watch = 0;          // W64_WATCH=n: report where a wave is every n rendezvous (finding a loop that never ends)
thread_local Lane* g_cur = nullptr;
struct StackPool {
  std::vector<unsigned char*> v;
  ~StackPool() { for (unsigned char* p : v) munmap(p, STACK_BYTES); }
};
thread_local StackPool g_stacks;

[[noreturn]] void die(const char* what, int a = 0, int b = 0) {
  fprintf(stderr, "[w64] %s (%d, %d)\n", what, a, b);
  abort();
}

void yield_to_scheduler() {
  Lane* me = g_cur;
  w64_switch(&me->sp, g_group->sched_sp);
}

void lane_entry() {
  Group* g = g_group;
  g->body(g->arg);
  Lane* me = g_cur;
  me->done = true;
  // a finished lane holds nothing a readlane could see - from the wave's NEXT operation on: the lanes resumed after this one in
  // the same interval still have to read what it deposited for the operation it has just passed (the scheduler clears that slot
  // once the interval is over)
  me->wave->val[(me->wave->lane_seq[me->lane] + 1) & 1u][me->lane] = 0;
  yield_to_scheduler();
  die("a finished lane was resumed");
}

unsigned char* stack_get(size_t i) {
  while (g_stacks.v.size() <= i) {
    void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) die("mmap of a fiber stack failed");
    mprotect(p, 4096, PROT_NONE);          // guard page below the stack
    g_stacks.v.push_back(static_cast<unsigned char*>(p));
  }
  return g_stacks.v[i];
}

void run_group(dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz, void (*body)(void*), void* arg) {
  const unsigned n = block.x * block.y * block.z;
  Group g;
  g.body = body; g.arg = arg;
  g.lanes.resize(n);
  g.waves.resize((n + 63) / 64);
  for (Wave& w : g.waves) {
    w.shuffle_state = g_shuffle * 0x9E3779B97F4A7C15ULL + bx * 1315423911ull + (uint64_t)(&w - &g.waves[0]);
    memset(w.val, 0, sizeof w.val);
    memset(w.lane_seq, 0, sizeof w.lane_seq);
  }
  for (unsigned t = 0; t < n; t++) {
    Lane& l = g.lanes[t];
    l.item.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    l.item.bid = dim3(bx, by, bz); l.item.bdim = block; l.item.gdim = grid;
    Wave& w = g.waves[t / 64];
    l.wave = &w; l.lane = t % 64;
    w.lanes[w.n++] = &l;
    l.stack = stack_get(t);
    // initial frame: six callee-saved registers, then the entry as the return address, then a slot that keeps the ABI's
    // alignment at function entry (rsp + 8 is a multiple of 16)
    uintptr_t top = (reinterpret_cast<uintptr_t>(l.stack) + STACK_BYTES) & ~(uintptr_t)15;
    uint64_t* sp = reinterpret_cast<uint64_t*>(top);
    *--sp = 0;                                              // (a return address lane_entry never uses)
    *--sp = reinterpret_cast<uint64_t>(&lane_entry);
    for (int k = 0; k < 6; k++) *--sp = 0;
    l.sp = sp;
  }
  Group* const prev = g_group;
  g_group = &g;
  for (;;) {
    bool any = false;
    for (Wave& w : g.waves) {
      if (w.done) continue;
      if (w.at_barrier) { any = true; continue; }
      // one pass: every running lane up to its next rendezvous
      bool running = false;
      unsigned char order[64];
      if (g_shuffle) {                          // W64_LANE_ORDER=shuffle[:seed]: a new permutation of the lanes for every interval
        for (unsigned k = 0; k < w.n; k++) order[k] = (unsigned char)k;
        for (unsigned k = w.n; k > 1; k--) {
          w.shuffle_state = w.shuffle_state * 6364136223846793005ULL + 1442695040888963407ULL;
          const unsigned j = (unsigned)((w.shuffle_state >> 33) % k);
          const unsigned char t = order[k - 1]; order[k - 1] = order[j]; order[j] = t;
        }
      }
      for (unsigned k0 = 0; k0 < w.n; k0++) {
        // W64_LANE_ORDER=reverse resumes the lanes of an interval from the last to the first: a kernel whose result depends on
        // which lane's stores another lane sees WITHOUT a cross-lane operation in between computes something else then
        Lane* l = w.lanes[g_shuffle ? order[k0] : g_reverse ? w.n - 1 - k0 : k0];
        if (l->done) continue;
        g_cur = l;
        w64_switch(&g.sched_sp, l->sp);
        running |= !l->done;
      }
      g_cur = nullptr;
      for (unsigned k = 0; k < w.n; k++) {
        Lane* l = w.lanes[k];
        if (l->done && !l->cleared) { w.val[0][l->lane] = 0; w.val[1][l->lane] = 0; l->cleared = true; }
      }
      if (!running) { w.done = true; continue; }
      any = true;
      // the lanes still running wait somewhere: it has to be ONE place
      int kind = 0, line = 0;
      for (unsigned k = 0; k < w.n; k++) {
        const Lane* l = w.lanes[k];
        if (l->done) continue;
        if (!kind) { kind = l->wait_kind; line = l->wait_line; }
        else if (kind != l->wait_kind || line != l->wait_line) {
          fprintf(stderr, "[w64] divergent rendezvous in workgroup (%u, %u, %u): a lane waits at line %d (operation %d), lane %u at line %d "
                          "(operation %d) - a cross-lane operation under divergent control flow is outside this emulator\n",
                  bx, by, bz, line, kind, l->lane, l->wait_line, l->wait_kind);
          abort();
        }
      }
      w.seq++;
      if (g_watch && w.seq % g_watch == 0)
        fprintf(stderr, "[w64 watch] workgroup %u wave %u: %llu rendezvous so far, now at line %d (operation %d)\n", bx,
                (unsigned)(&w - &g.waves[0]), (unsigned long long)w.seq, line, kind);
      if (kind == W64_SYNCTHREADS) w.at_barrier = true;
    }
    if (!any) break;
    // release the workgroup barrier once every wave that still runs waits at it
    bool release = true;
    for (const Wave& w : g.waves) if (!w.done && !w.at_barrier) release = false;
    if (release) for (Wave& w : g.waves) w.at_barrier = false;
  }
  g_group = prev;
}

int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  if (!e) return dflt;
  const long v = strtol(e, nullptr, 10);
  return v < lo ? lo : v > hi ? hi : (int)v;
}

}  // namespace

extern "C" const w64_item* w64_self() {
  if (!g_cur) die("threadIdx / blockIdx outside a kernel");
  return &g_cur->item;
}

extern "C" w64_xchg w64_exchange(uint64_t value, int kind, int line) {
  Lane* me = g_cur;
  if (!me) die("cross-lane operation outside a kernel", kind, line);
  Wave* w = me->wave;
  const uint64_t n = ++w->lane_seq[me->lane];      // this lane's n-th rendezvous
  const unsigned par = (unsigned)(n & 1u);
  w->val[par][me->lane] = value;
  me->wait_kind = kind; me->wait_line = line;
  yield_to_scheduler();
  // resumed: every running lane of the wave has deposited
  uint64_t live = 0;
  for (unsigned k = 0; k < w->n; k++) {
    const Lane* l = w->lanes[k];
    if (w->lane_seq[l->lane] >= n) live |= 1ull << l->lane;      // (a lane that finished earlier never arrived at n)
  }
  w64_xchg r;
  r.live = live; r.v = w->val[par]; r.lane = me->lane;
  return r;
}

extern "C" void w64_block_barrier(int line) {
  Lane* me = g_cur;
  if (!me) die("__syncthreads outside a kernel", line);
  me->wave->lane_seq[me->lane]++;
  me->wait_kind = W64_SYNCTHREADS; me->wait_line = line;
  yield_to_scheduler();
}

extern "C" int w64_num_cu() { return env_int("W64_NUM_CU", 2, 1, 64); }
extern "C" int w64_occupancy() { return env_int("W64_OCCUPANCY", 2, 1, 8); }

extern "C" void w64_launch(dim3 grid, dim3 block, size_t lds_bytes, void (*body)(void*), void* arg) {
  if (lds_bytes > sizeof demi::smem) die("dynamic LDS beyond 160 KB", (int)lds_bytes);
  const uint64_t n_groups = (uint64_t)grid.x * grid.y * grid.z;
  if (block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024) die("workgroup size", (int)block.x);
  { const char* o = getenv("W64_LANE_ORDER"); g_reverse = o && o[0] == 'r';
    g_shuffle = (o && o[0] == 's') ? (strchr(o, ':') ? strtoull(strchr(o, ':') + 1, nullptr, 10) : 1) | 1ull : 0; }
  { const char* o = getenv("W64_WATCH"); g_watch = o ? strtoull(o, nullptr, 10) : 0; }
  if (getenv("W64_TRACE")) fprintf(stderr, "[w64 launch] grid %u block %u lds %zu\n", grid.x, block.x, lds_bytes);
  const int n_thr = (int)std::min<uint64_t>((uint64_t)env_int("W64_THREADS", 4, 1, 64), n_groups);
  std::atomic<uint64_t> next{0};
  auto work = [&]() {
    for (;;) {
      const uint64_t i = next.fetch_add(1);
      if (i >= n_groups) break;
      memset(demi::smem, 0xCD, lds_bytes);     // LDS starts undefined
      run_group(grid, block, (unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((uint64_t)grid.x * grid.y)), body, arg);
    }
  };
  if (n_thr <= 1) { work(); return; }
  std::vector<std::thread> pool;
  for (int t = 1; t < n_thr; t++) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
}
