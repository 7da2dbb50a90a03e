// jit.hpp — host side: compile the loaded transition table to native gfx950 code (hiprtc).
//
// The generic kernels interpret the table row by row: every lane decodes its own row and computes every
// result class with selects (sim_core.hpp vm_run, ~150 VALU instructions per row) so that lanes sitting in
// different handlers do not serialise.  Once a model is loaded its rows are constants, so the same handlers
// can be emitted as straight-line code over 16 named byte registers with real branches: a row becomes 1-4
// instructions, and a wave pays only for the rows some lane actually takes.  demi_model_specialize()
// generates that source (generate_vm below), compiles the K1 kernel with it through hiprtc and the launch
// path then uses the module's kernel; everything else (scheduling step, pending set, effects, verdict) is
// the same code as the generic kernel, and the results stay bit-identical (tests run both).
//
// hiprtc is resolved with dlopen at specialisation time, next to the HIP runtime this process already uses
// (PyTorch bundles its own), so the library has no link-time dependency on it and keeps working without it.
#pragma once
#include "knobs.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sim_core.hpp"

namespace demi_jit {

// ------------------------------------------------------------------ static effect-slot schedule (K1)
// The generic kernels record a delivery's effect rows into a queue and apply entry k of every lane together; at entry k
// some lane of the wave has a SEND, another a TCANCEL, a third a TSET, so the wave pays for every body at every entry.
// For a loaded table the effect rows are known: each one gets a FIXED slot in a short schedule of effect classes
// (all SEND / BCAST rows are one class; timer rows are a class per (op, timer type)) such that along every control-flow
// path of every handler the slots increase - program order is kept - and slot j of all lanes is the same class: the
// apply phase becomes one straight pass over the schedule, each body once, with the class constants folded in.
// The schedule is a common supersequence of the handlers' effect-class sequences (majority merge over the distinct paths,
// then one pass over the row DAG that fixes every row's slot and extends the schedule where the paths of one row disagree).
// Not applicable (the dynamic queue is kept) when some path holds more than DEMI_FX_CAP effect rows - the queue overflow
// is then a possible verdict - or the schedule needs more than DEMI_FX_CAP slots.
struct FxSchedule {
  bool ok = false;
  std::vector<uint32_t> cls;        // per slot: kind << 16 | op << 8 | type   (kind 0: send, op / type unused)
  std::vector<int32_t> slot;        // per row: its slot, -1 for rows that are not effect rows
};
enum : uint32_t { FXK_SEND = 0, FXK_CANCEL = 1, FXK_TSET = 2, FXK_CRASH = 3 };
inline uint32_t fx_class_of(uint32_t row) {
  const uint32_t op = row & 0x3Fu, type = (row >> 17) & 0x7Fu;
  if (op == DEMI_OP_SEND || op == DEMI_OP_BCAST) return FXK_SEND << 16;
  if (op == DEMI_OP_TCANCEL) return (FXK_CANCEL << 16) | (op << 8) | type;
  if (op == DEMI_OP_TSET || op == DEMI_OP_TREP) return (FXK_TSET << 16) | (op << 8) | type;
  return FXK_CRASH << 16;
}
inline FxSchedule fx_schedule(const demi::DevModel& h) {
  using namespace demi;
  FxSchedule S;
  const uint32_t n = h.code_len;
  S.slot.assign(n, -1);
  // successors of a row (n = the handler is done)
  auto succ = [&](uint32_t pc, uint32_t out[2]) -> int {
    const uint32_t row = h.code[pc], cw = op_control(row & 0x3Fu);
    auto clip = [&](uint32_t t) { return t < n ? t : n; };
    if (cw & CW_HALT) return 0;                                   // HALT, CRASH
    if (cw & CW_IF) { out[0] = clip(pc + 1); out[1] = clip(pc + 1 + ((row >> 17) & 0x7Fu)); return 2; }
    if (cw & (CW_SKIPZ | CW_SKIPNZ)) { out[0] = clip(pc + 1); out[1] = clip(pc + 1 + (row >> 24)); return 2; }
    if (cw & CW_SKIP) { out[0] = clip(pc + 1 + (row >> 24)); return 1; }
    out[0] = clip(pc + 1);
    return 1;
  };
  auto is_fx = [&](uint32_t pc) { return (op_control(h.code[pc] & 0x3Fu) & CW_FX) != 0; };
  std::vector<uint32_t> starts;
  for (uint32_t i = 0; i < h.n_classes * h.n_msg_types; i++) {
    const uint32_t st = h.handler_start[i];
    if (st == 0xFFFF || st >= n) continue;
    if (std::find(starts.begin(), starts.end(), st) == starts.end()) starts.push_back(st);
  }
  // most effect rows on any path from a row (the DAG has forward edges only)
  std::vector<uint32_t> most(n + 1, 0);
  for (uint32_t pc = n; pc-- > 0;) {
    uint32_t o[2], m = 0;
    const int k = succ(pc, o);
    for (int i = 0; i < k; i++) m = std::max(m, most[o[i]]);
    most[pc] = m + (is_fx(pc) ? 1u : 0u);
  }
  for (uint32_t st : starts) if (most[st] > DEMI_FX_CAP) return S;
  // the distinct effect-class sequences of the handlers' paths (bounded enumeration; beyond the bound the DAG pass below
  // still produces a valid schedule, only a longer one)
  std::vector<std::vector<uint32_t>> seqs;
  {
    size_t budget = 20000;
    std::vector<uint32_t> cur;
    struct Frame { uint32_t pc; int next; size_t depth; };
    for (uint32_t st : starts) {
      std::vector<Frame> stack;
      stack.push_back({st, 0, 0});
      cur.clear();
      while (!stack.empty() && budget) {
        Frame& f = stack.back();
        if (f.pc >= n) {
          if (!cur.empty() && std::find(seqs.begin(), seqs.end(), cur) == seqs.end()) seqs.push_back(cur);
          budget--;
          stack.pop_back();
          continue;
        }
        uint32_t o[2];
        const int k = succ(f.pc, o);
        if (f.next == 0) {
          f.depth = cur.size();
          if (is_fx(f.pc)) cur.push_back(fx_class_of(h.code[f.pc]));
          if (k == 0) {
            if (!cur.empty() && std::find(seqs.begin(), seqs.end(), cur) == seqs.end()) seqs.push_back(cur);
            budget--;
          }
        }
        if (f.next < k) {
          const uint32_t to = o[f.next];
          f.next++;
          cur.resize(f.depth + (is_fx(f.pc) ? 1 : 0));
          stack.push_back({to, 0, 0});
        } else {
          cur.resize(f.depth);
          stack.pop_back();
        }
      }
    }
  }
  // a cheap common supersequence of those sequences.  First guess: majority merge (the class at the front of the most
  // remaining sequences comes next); then a bounded branch-and-bound over "which front class next" looks for a cheaper one
  // (cost = the bodies the kernel will execute per delivery: a send or a cancel body is about twice a timer-set body)
  auto weight = [](uint32_t c) -> uint32_t { const uint32_t k = c >> 16; return k == FXK_TSET ? 2u : (k == FXK_CRASH ? 1u : 4u); };
  {
    std::vector<size_t> at(seqs.size(), 0);
    for (;;) {
      std::vector<std::pair<uint32_t, uint32_t>> votes;          // class, count
      for (size_t i = 0; i < seqs.size(); i++) {
        if (at[i] >= seqs[i].size()) continue;
        const uint32_t c = seqs[i][at[i]];
        bool hit = false;
        for (auto& v : votes) if (v.first == c) { v.second++; hit = true; }
        if (!hit) votes.push_back({c, 1u});
      }
      if (votes.empty()) break;
      uint32_t best = 0;
      for (uint32_t i = 1; i < votes.size(); i++) if (votes[i].second > votes[best].second) best = i;
      const uint32_t c = votes[best].first;
      S.cls.push_back(c);
      for (size_t i = 0; i < seqs.size(); i++) if (at[i] < seqs[i].size() && seqs[i][at[i]] == c) at[i]++;
      if (S.cls.size() > 4 * DEMI_FX_CAP) return S;
    }
    uint32_t best_cost = 0;
    for (uint32_t c : S.cls) best_cost += weight(c);
    size_t nodes = 200000;
    std::vector<uint32_t> cur;
    std::vector<size_t> pos(seqs.size(), 0);
    // (plain recursion: the depth is at most the length of the first guess)
    struct Rec {
      const std::vector<std::vector<uint32_t>>& seqs; std::vector<uint32_t>& best; uint32_t& best_cost; size_t& nodes;
      decltype(weight)& w;
      void go(std::vector<size_t>& pos, std::vector<uint32_t>& cur, uint32_t cost) {
        if (nodes == 0) return;
        nodes--;
        std::vector<uint32_t> fronts;
        uint32_t need = 0;                                       // lower bound: the most expensive remaining single sequence
        for (size_t i = 0; i < seqs.size(); i++) {
          if (pos[i] >= seqs[i].size()) continue;
          if (std::find(fronts.begin(), fronts.end(), seqs[i][pos[i]]) == fronts.end()) fronts.push_back(seqs[i][pos[i]]);
          uint32_t rest = 0;
          for (size_t k = pos[i]; k < seqs[i].size(); k++) rest += w(seqs[i][k]);
          need = std::max(need, rest);
        }
        if (fronts.empty()) {
          if (cost < best_cost || (cost == best_cost && cur.size() < best.size())) { best = cur; best_cost = cost; }
          return;
        }
        if (cost + need > best_cost || (cost + need == best_cost && cur.size() + 1 >= best.size())) return;
        for (uint32_t c : fronts) {
          std::vector<size_t> saved = pos;
          for (size_t i = 0; i < seqs.size(); i++) if (pos[i] < seqs[i].size() && seqs[i][pos[i]] == c) pos[i]++;
          cur.push_back(c);
          go(pos, cur, cost + w(c));
          cur.pop_back();
          pos = saved;
        }
      }
    } rec{seqs, S.cls, best_cost, nodes, weight};
    rec.go(pos, cur, 0);
  }
  // fix every row's slot: the first slot of its class after the slots of every effect row that can precede it
  std::vector<int32_t> lo(n + 1, -1);                            // -1: not reachable
  for (uint32_t st : starts) lo[st] = 0;
  for (uint32_t pc = 0; pc < n; pc++) {
    if (lo[pc] < 0) continue;
    int32_t next = lo[pc];
    if (is_fx(pc)) {
      const uint32_t c = fx_class_of(h.code[pc]);
      int32_t j = next;
      while (j < (int32_t)S.cls.size() && S.cls[j] != c) j++;
      if (j >= (int32_t)S.cls.size()) { S.cls.push_back(c); j = (int32_t)S.cls.size() - 1; }
      S.slot[pc] = j;
      next = j + 1;
    }
    uint32_t o[2];
    const int k = succ(pc, o);
    for (int i = 0; i < k; i++) if (o[i] < n) lo[o[i]] = std::max(lo[o[i]], next);
  }
  // slots nobody uses (the merge saw a path the DAG pass placed elsewhere) are dropped
  {
    std::vector<int32_t> used(S.cls.size(), 0), remap(S.cls.size(), -1);
    for (uint32_t pc = 0; pc < n; pc++) if (S.slot[pc] >= 0) used[S.slot[pc]] = 1;
    std::vector<uint32_t> cls2;
    for (size_t j = 0; j < S.cls.size(); j++) if (used[j]) { remap[j] = (int32_t)cls2.size(); cls2.push_back(S.cls[j]); }
    for (uint32_t pc = 0; pc < n; pc++) if (S.slot[pc] >= 0) S.slot[pc] = remap[S.slot[pc]];
    S.cls.swap(cls2);
  }
  S.ok = S.cls.size() <= DEMI_FX_CAP;
  return S;
}

// effect-queue entries the scheduled K1 of a table needs (its SEND / BCAST slots), DEMI_FX_CAP without a schedule
inline uint32_t k1_fxq_slots(const demi::DevModel& h, bool sched) {
  if (!sched) return DEMI_FX_CAP;
  const FxSchedule fxs = fx_schedule(h);
  if (!fxs.ok) return DEMI_FX_CAP;
  uint32_t nq = 0;
  for (uint32_t c : fxs.cls) nq += (c >> 16) == FXK_SEND;
  return nq ? nq : 1u;
}

// ------------------------------------------------------------------ code generation
// One statement block per row; forward skips become gotos (the table has no backward edges, validation
// guarantees it), handler entry is a switch over the distinct handler starts.
// `sched`: K1's flavour with the static effect-slot schedule above (vm_run_jit then returns the mask of filled slots);
// without it, or when no schedule exists, effect rows are queued in program order as vm_run does (K2 / K3).
inline std::string generate_vm(const demi::DevModel& h, bool sched = false) {
  using namespace demi;
  std::string s;
  char buf[512];
  auto emit = [&](const char* fmt, auto... a) { snprintf(buf, sizeof buf, fmt, a...); s += buf; };
  FxSchedule fxs;
  std::vector<uint32_t> fxq_of_slot;
  if (sched) fxs = fx_schedule(h);
  if (fxs.ok) {
    // the schedule for the kernel's apply phase: one DEMI_FX_SLOT(slot, kind, op, type, timer index) per slot, in order
    // (only the SEND / BCAST slots carry data: they get the entries 0, 1, ... of the effect queue, DEMI_JIT_FXQ_SLOTS in all)
    s += "#define DEMI_JIT_FX_SCHED 1\n#define DEMI_JIT_FX_APPLY";
    uint32_t nq = 0;
    for (size_t j = 0; j < fxs.cls.size(); j++) {
      const uint32_t c = fxs.cls[j], type = c & 0xFFu;
      const bool snd = (c >> 16) == FXK_SEND;
      emit(" DEMI_FX_SLOT(%zu, %uu, %uu, %uu, %uu, %uu)", j, c >> 16, (c >> 8) & 0xFFu, type, snd ? 0u : (h.meta[type & 31u] >> 8), snd ? nq : 0u);
      fxq_of_slot.push_back(snd ? nq : 0u);
      if (snd) nq++;
    }
    emit("\n#define DEMI_JIT_FXQ_SLOTS %uu\n", nq ? nq : 1u);
  }
  s += "namespace demi {\n";
  // (PAY: the two payload operands of the row - or, for a table with DEMI_MODEL_PAYLOADS, the payload area with the staged
  // fields P2.. packed in: fx_pack_area, sim_core.hpp)
  const bool long_msg = h.npay > 2;
  const char* fx_pack_fn = long_msg ? "fx_pack_area" : "fx_pack";
  if (fxs.ok)
    // an effect row fills its own slot of the schedule; only SEND / BCAST rows carry data (a timer row IS its slot)
    s += std::string("#define DEMI_FX_AT(SLOT, Q, OP, TYPE, TGT, ...) { mem.fxq[(Q) * 64] = ") + fx_pack_fn + "(OP, TYPE, TGT, __VA_ARGS__); nfx |= 1u << (SLOT); }\n"
         "#define DEMI_FX_MARK(SLOT) { nfx |= 1u << (SLOT); }\n";
  else
    // effect rows are recorded into the LDS effect queue in program order, exactly as vm_run does
    s += std::string("#define DEMI_FX(OP, TYPE, TGT, ...) { if (nfx >= DEMI_FX_CAP) { flags |= DEMI_V_QUEUE_OVF; goto done; } "
         "mem.fxq[nfx * 64] = ") + fx_pack_fn + "(OP, TYPE, TGT, __VA_ARGS__); nfx++; }\n";
  // A wide table (DevModel::wide) has 16-bit registers: the same statements with the masks of the wider window, the
  // state in two words, and 64-bit message / effect words (word_t of a -DDEMI_WIDE translation unit).
  const bool wide = h.wide != 0;
  const unsigned RM = wide ? 65535u : 255u, SM = wide ? 15u : 7u;
  emit("__device__ inline uint32_t vm_run_jit(const Tables& t, const LaneMem& mem, %s w, uint32_t& flags, uint64_t& app_rng) {\n",
       wide ? "word_t" : "uint32_t");
  s += "  const uint32_t type = w_type(w), me = w_dst(w);\n";
  s += "  const uint32_t entry = t.hs[((t.ac_packed >> (4 * me)) & 15u) * t.NT + type];\n";
  s += "  if (entry == 0xFFFFu) return 0;\n";
  if (wide) {
    s += "  const uint64_t st0 = mem.st[(ST_WORDS * me) * 64], st1 = mem.st[(ST_WORDS * me + 1) * 64];\n";
    s += "  uint32_t r0 = (uint32_t)st0 & 65535u, r1 = (uint32_t)(st0 >> 16) & 65535u, r2 = (uint32_t)(st0 >> 32) & 65535u, "
         "r3 = (uint32_t)(st0 >> 48) & 65535u,\n           r4 = (uint32_t)st1 & 65535u, r5 = (uint32_t)(st1 >> 16) & 65535u, "
         "r6 = (uint32_t)(st1 >> 32) & 65535u, r7 = (uint32_t)(st1 >> 48) & 65535u;\n";
  } else {
    s += "  const uint64_t st0 = mem.st[(ST_WORDS * me) * 64];\n";
    s += "  uint32_t r0 = (uint32_t)st0 & 255u, r1 = (uint32_t)(st0 >> 8) & 255u, r2 = (uint32_t)(st0 >> 16) & 255u, "
         "r3 = (uint32_t)(st0 >> 24) & 255u,\n           r4 = (uint32_t)(st0 >> 32) & 255u, r5 = (uint32_t)(st0 >> 40) & 255u, "
         "r6 = (uint32_t)(st0 >> 48) & 255u, r7 = (uint32_t)(st0 >> 56) & 255u;\n";
  }
  s += "  uint32_t r8 = 0, r9 = 0, r10 = 0, r11 = 0, r12 = w_p0(w), r13 = w_p1(w), r14 = w_src(w), r15 = me;\n";
  s += "  uint32_t nfx = 0;\n";
  s += "  (void)r8; (void)r9; (void)r10; (void)r11; (void)r12; (void)r13; (void)r14; (void)r15;\n";
  if (long_msg) s += "  uint32_t q2 = 0, q3 = 0, q4 = 0, q5 = 0;   // DEMI_OP_PSET: the staged payload fields of the messages sent next\n"
                     "  (void)q2; (void)q3; (void)q4; (void)q5;\n";
  // distinct handler starts
  std::vector<uint32_t> starts;
  for (uint32_t i = 0; i < h.n_classes * h.n_msg_types; i++) {
    const uint32_t st = h.handler_start[i];
    if (st == 0xFFFF || st >= h.code_len) continue;
    bool seen = false;
    for (uint32_t x : starts) seen |= (x == st);
    if (!seen) starts.push_back(st);
  }
  s += "  switch (entry) {\n";
  for (uint32_t st : starts) emit("    case %uu: goto L%u;\n", st, st);
  s += "    default: goto done;\n  }\n";
  auto target = [&](uint32_t pc) -> std::string {
    if (pc >= h.code_len) return "done";
    return "L" + std::to_string(pc);
  };
  const char* relop[6] = {"==", "!=", "<", ">=", "<=", ">"};     // EQ NE LT GE LE GT
  // Optional if-conversion (DEMI_JIT_IFCONVERT = longest guarded run, 0 / unset = off): a fused guard over a short
  // run of pure ALU rows, none of which is a jump target or a handler entry, becomes selects instead of a branch
  // (fewer exec-mask manipulations on the scalar unit).  Same semantics: a skipped ALU row leaves its register alone.
  uint32_t ifconv = 0;
  if (const char* e = demi_host::knob("DEMI_JIT_IFCONVERT")) { const long x = strtol(e, nullptr, 10); ifconv = x > 0 && x < 16 ? (uint32_t)x : 0; }
  std::vector<uint8_t> is_target(h.code_len + 1, 0);
  for (uint32_t st : starts) is_target[st] = 1;
  for (uint32_t pc = 0; pc < h.code_len; pc++) {
    const uint32_t row = h.code[pc], cw = op_control(row & 0x3Fu);
    uint32_t t = h.code_len;
    if (cw & CW_IF) t = pc + 1 + ((row >> 17) & 0x7Fu);
    else if (cw & (CW_SKIPZ | CW_SKIPNZ | CW_SKIP)) t = pc + 1 + (row >> 24);
    if (t < h.code_len) is_target[t] = 1;
  }
  std::vector<int32_t> pred_of(h.code_len, -1);      // row -> the guard row whose predicate it runs under, if converted
  for (uint32_t pc = 0; ifconv && pc < h.code_len; pc++) {
    const uint32_t row = h.code[pc];
    if (!(op_control(row & 0x3Fu) & CW_IF) || pred_of[pc] >= 0) continue;
    const uint32_t len = (row >> 17) & 0x7Fu;
    if (len == 0 || len > ifconv || pc + 1 + len > h.code_len) continue;
    bool ok = true;
    for (uint32_t q = pc + 1; q <= pc + len && ok; q++)
      ok = (op_control(h.code[q] & 0x3Fu) & CW_ALU) && !(op_control(h.code[q] & 0x3Fu) & CW_RND) && !is_target[q] && pred_of[q] < 0;
    if (!ok) continue;
    for (uint32_t q = pc + 1; q <= pc + len; q++) pred_of[q] = (int32_t)pc;
  }
  // the predicates are declared before the dispatch switch: no goto may cross an initialisation
  std::string preds;
  for (uint32_t pc = 0; pc + 1 < h.code_len; pc++)
    if (pred_of[pc + 1] == (int32_t)pc) preds += "  bool c" + std::to_string(pc) + " = false;\n";
  if (!preds.empty()) {
    const size_t at = s.find("  switch (entry) {");
    s.insert(at, preds);
  }
  // the value an ALU row assigns, as an expression over the register names a, b
  const char* arr_of = "mem.st, me";      // (whose array an LDX reads: the receiving actor's; the invariant program's actor below)
  auto alu_value = [&](uint32_t op, const char* a, const char* b, char* val, size_t cap) {
    switch (op) {
      case DEMI_OP_LDX: snprintf(val, cap, "arr_load(%s, %s)", arr_of, b); break;      // DEMI_MODEL_ARRAY (demi_device.hpp)
      case DEMI_OP_MOV: snprintf(val, cap, "%s & %uu", b, RM); break;
      case DEMI_OP_MOVHI: snprintf(val, cap, "((%s & 255u) | (%s << 8)) & %uu", a, b, RM); break;   // (wide tables only)
      case DEMI_OP_ADD: snprintf(val, cap, "(%s + %s) & %uu", a, b, RM); break;
      case DEMI_OP_SUB: snprintf(val, cap, "(%s - %s) & %uu", a, b, RM); break;
      case DEMI_OP_AND: snprintf(val, cap, "%s & %s & %uu", a, b, RM); break;
      case DEMI_OP_OR: snprintf(val, cap, "(%s | %s) & %uu", a, b, RM); break;
      case DEMI_OP_XOR: snprintf(val, cap, "(%s ^ %s) & %uu", a, b, RM); break;
      case DEMI_OP_SHL: snprintf(val, cap, "(%s << (%s & %uu)) & %uu", a, b, SM, RM); break;
      case DEMI_OP_SHR: snprintf(val, cap, "(%s >> (%s & %uu)) & %uu", a, b, SM, RM); break;
      case DEMI_OP_BITSET: snprintf(val, cap, "(%s | (1u << (%s & %uu))) & %uu", a, b, SM, RM); break;
      case DEMI_OP_POPC: snprintf(val, cap, "(uint32_t)__popc(%s)", b); break;
      case DEMI_OP_MIN: snprintf(val, cap, "%s < %s ? %s : %s", a, b, a, b); break;
      case DEMI_OP_MAX: snprintf(val, cap, "%s < %s ? %s : %s", a, b, b, a); break;
      case DEMI_OP_PEER: snprintf(val, cap, "0u"); break;     // (invariant programs only: their rows are emitted by the loop further down)
      case DEMI_OP_LDP: snprintf(val, cap, "w_pay(w, %s)", b); break;       // DEMI_MODEL_PAYLOADS (demi_device.hpp)
      case DEMI_OP_RND:      // (a wide table's bound is b & 0xFF: the magics cover 1..256)
        if (wide) snprintf(val, cap, "app_next_int(app_rng, %s & 255u, t.gmagic)", b);
        else snprintf(val, cap, "app_next_int(app_rng, %s, t.gmagic)", b);
        break;
      default: snprintf(val, cap, "(%s %s %s) ? 1u : 0u", a, relop[op - DEMI_OP_EQ], b); break;   // EQ .. GT
    }
  };
  for (uint32_t pc = 0; pc < h.code_len; pc++) {
    const uint32_t row = h.code[pc];
    const uint32_t op = row & 0x3Fu, dsti = (row >> 8) & 15u, ai = (row >> 12) & 15u, aux = (row >> 17) & 0x7Fu,
                   braw = row >> 24;
    const bool bimm = (row & 0x10000u) != 0;
    char a[8], b[16], d[8];
    snprintf(a, sizeof a, "r%u", ai);
    snprintf(d, sizeof d, "r%u", dsti);
    if (bimm) snprintf(b, sizeof b, "%uu", braw); else snprintf(b, sizeof b, "r%u", braw & 15u);
    emit("  L%u: ", pc);
    const uint32_t cw = op_control(op);
    if ((cw & CW_HALT) && !(cw & CW_FX)) {
      s += "goto done;\n";
      continue;
    }
    if (cw & CW_ALU) {
      char val[96];
      alu_value(op, a, b, val, sizeof val);
      if (pred_of[pc] >= 0) emit("%s = c%d ? (%s) : %s;\n", d, pred_of[pc], val, d);
      else emit("%s = %s;\n", d, val);
    } else if (cw & CW_IF) {
      if (pc + 1 < h.code_len && pred_of[pc + 1] == (int32_t)pc)
        emit("c%u = (%s %s %s);\n", pc, a, relop[op - DEMI_OP_IFEQ], b);
      else
        emit("if (!(%s %s %s)) goto %s;\n", a, relop[op - DEMI_OP_IFEQ], b, target(pc + 1 + aux).c_str());
    } else if (cw & CW_SKIPZ) {
      emit("if (%s == 0u) goto %s;\n", a, target(pc + 1 + braw).c_str());
    } else if (cw & CW_SKIPNZ) {
      emit("if (%s != 0u) goto %s;\n", a, target(pc + 1 + braw).c_str());
    } else if (cw & CW_SKIP) {
      emit("goto %s;\n", target(pc + 1 + braw).c_str());
    } else if (cw & CW_STX) {
      emit("arr_store(mem.st, me, %s, %s);\n", b, a);
    } else if (cw & CW_PSET) {
      emit("q%u = %s;\n", aux >= 2 && aux <= 5 ? aux : 2u, b);
    } else {   // CW_FX: recorded now, applied after the rows have run (same record as vm_run)
      char pay[96];
      if (long_msg) snprintf(pay, sizeof pay, "pay_area(%s, %s, q2, q3, q4, q5)", d, b);
      else snprintf(pay, sizeof pay, "%s, %s", d, b);
      if (fxs.ok && (fxs.cls[fxs.slot[pc]] >> 16) != FXK_SEND)
        emit("DEMI_FX_MARK(%d)%s\n", fxs.slot[pc], (cw & CW_HALT) ? " goto done;" : "");
      else if (fxs.ok)
        emit("DEMI_FX_AT(%d, %uu, %uu, %uu, %s > FX_NOBODY ? FX_NOBODY : %s, %s)\n", fxs.slot[pc], fxq_of_slot[fxs.slot[pc]], row & 0xFFu, aux, a, a, pay);
      else      // (a target beyond the field - sim_core.hpp FX_NOBODY: 15, 31 in the BIG layout - is nobody)
        emit("DEMI_FX(%uu, %uu, %s > FX_NOBODY ? FX_NOBODY : %s, %s)%s\n", row & 0xFFu, aux, a, a, pay, (cw & CW_HALT) ? " goto done;" : "");
    }
  }
  s += "  done:\n";
  if (wide)
    s += "  mem.st[(ST_WORDS * me) * 64] = (uint64_t)(r0 | (r1 << 16)) | ((uint64_t)(r2 | (r3 << 16)) << 32);\n"
         "  mem.st[(ST_WORDS * me + 1) * 64] = (uint64_t)(r4 | (r5 << 16)) | ((uint64_t)(r6 | (r7 << 16)) << 32);\n";
  else
    s += "  mem.st[(ST_WORDS * me) * 64] = (uint64_t)(r0 | (r1 << 8) | (r2 << 16) | (r3 << 24)) | "
         "((uint64_t)(r4 | (r5 << 8) | (r6 << 16) | (r7 << 24)) << 32);\n";
  s += "  return nfx;\n}\n";
  if (h.inv_kind & DEMI_INV_PROGRAM) {
    // the invariant's per-actor program (include/demi_gpu.h DEMI_INV_PROGRAM): the rows from inv_fa on, run on one actor's
    // state - r0..r7 its fields, r15 its id, everything else 0; T0 = hit, T1 = key; what it writes to the fields is discarded
    // (jit_source defines DEMI_JIT_INV_PROG ahead of sim_core.hpp, whose inv_prog() then calls this)
    s += "__device__ inline uint32_t inv_prog_jit(const uint64_t* st, uint32_t actor, uint32_t& key, uint32_t exists, uint32_t n_actors) {\n"
         "  (void)exists; (void)n_actors;\n";
    if (wide) {
      s += "  const uint64_t st0 = st[(ST_WORDS * actor) * 64], st1 = st[(ST_WORDS * actor + 1) * 64];\n";
      s += "  uint32_t r0 = (uint32_t)st0 & 65535u, r1 = (uint32_t)(st0 >> 16) & 65535u, r2 = (uint32_t)(st0 >> 32) & 65535u, "
           "r3 = (uint32_t)(st0 >> 48) & 65535u,\n           r4 = (uint32_t)st1 & 65535u, r5 = (uint32_t)(st1 >> 16) & 65535u, "
           "r6 = (uint32_t)(st1 >> 32) & 65535u, r7 = (uint32_t)(st1 >> 48) & 65535u;\n";
    } else {
      s += "  const uint64_t st0 = st[(ST_WORDS * actor) * 64];\n";
      s += "  uint32_t r0 = (uint32_t)st0 & 255u, r1 = (uint32_t)(st0 >> 8) & 255u, r2 = (uint32_t)(st0 >> 16) & 255u, "
           "r3 = (uint32_t)(st0 >> 24) & 255u,\n           r4 = (uint32_t)(st0 >> 32) & 255u, r5 = (uint32_t)(st0 >> 40) & 255u, "
           "r6 = (uint32_t)(st0 >> 48) & 255u, r7 = (uint32_t)(st0 >> 56) & 255u;\n";
    }
    s += "  uint32_t r8 = 0, r9 = 0, r10 = 0, r11 = 0, r12 = 0, r13 = 0, r14 = 0, r15 = actor;\n";
    s += "  (void)r0; (void)r1; (void)r2; (void)r3; (void)r4; (void)r5; (void)r6; (void)r7; (void)r10; (void)r11; (void)r12; (void)r13; (void)r14; (void)r15;\n";
    auto itarget = [&](uint32_t pc) -> std::string { return pc >= h.code_len ? "idone" : "I" + std::to_string(pc); };
    arr_of = "st, actor";
    for (uint32_t pc = h.inv_fa; pc < h.code_len; pc++) {
      const uint32_t row = h.code[pc];
      const uint32_t op = row & 0x3Fu, dsti = (row >> 8) & 15u, ai = (row >> 12) & 15u, aux = (row >> 17) & 0x7Fu, braw = row >> 24;
      const bool bimm = (row & 0x10000u) != 0;
      char a[8], b[16], d[8];
      snprintf(a, sizeof a, "r%u", ai);
      snprintf(d, sizeof d, "r%u", dsti);
      if (bimm) snprintf(b, sizeof b, "%uu", braw); else snprintf(b, sizeof b, "r%u", braw & 15u);
      emit("  I%u: ", pc);
      const uint32_t cw = op_control(op);
      if (cw & CW_HALT) { s += "goto idone;\n"; continue; }
      if (cw & CW_PEER) {
        emit("%s = peer_field(st, %s, %uu, exists, n_actors);\n", d, a, aux);
      } else if ((cw & CW_ALU) && !(cw & CW_RND)) {
        char val[96];
        alu_value(op, a, b, val, sizeof val);
        emit("%s = %s;\n", d, val);
      } else if (cw & CW_IF) {
        emit("if (!(%s %s %s)) goto %s;\n", a, relop[op - DEMI_OP_IFEQ], b, itarget(pc + 1 + aux).c_str());
      } else if (cw & CW_SKIPZ) {
        emit("if (%s == 0u) goto %s;\n", a, itarget(pc + 1 + braw).c_str());
      } else if (cw & CW_SKIPNZ) {
        emit("if (%s != 0u) goto %s;\n", a, itarget(pc + 1 + braw).c_str());
      } else if (cw & CW_SKIP) {
        emit("goto %s;\n", itarget(pc + 1 + braw).c_str());
      } else {
        s += ";   /* (refused at load: an invariant program has no effects) */\n";
      }
    }
    s += "  idone:\n  key = r9;\n  return r8;\n}\n";
  }
  s += "}  // namespace demi\n";
  return s;
}

// ------------------------------------------------------------------ hiprtc through dlopen
struct Rtc {
  void* lib = nullptr;
  int (*create)(void**, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*compile)(void*, int, const char* const*) = nullptr;
  int (*log_size)(void*, size_t*) = nullptr;
  int (*log)(void*, char*) = nullptr;
  int (*code_size)(void*, size_t*) = nullptr;
  int (*code)(void*, char*) = nullptr;
  int (*destroy)(void**) = nullptr;
  int (*add_name)(void*, const char*) = nullptr;
  int (*lowered)(void*, const char*, const char**) = nullptr;

  bool open(std::string& err) {
    if (lib) return true;
    // prefer the hiprtc that ships next to the HIP runtime already mapped into this process
    std::vector<std::string> cands;
    if (const char* only = getenv("DEMI_HIPRTC_LIB")) {      // an explicit library (tests use a bogus path to see the fallback)
      lib = dlopen(only, RTLD_NOW | RTLD_LOCAL);
      if (!lib) { err = std::string("hiprtc not found (DEMI_HIPRTC_LIB=") + only + ")"; return false; }
    }
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
      std::string p(info.dli_fname);
      const size_t slash = p.rfind('/');
      if (slash != std::string::npos) cands.push_back(p.substr(0, slash) + "/libhiprtc.so");
    }
    cands.push_back("libhiprtc.so.7");
    cands.push_back("libhiprtc.so");
    for (const std::string& c : cands) {
      if (lib) break;
      lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    }
    if (!lib) { err = "hiprtc not found (dlopen libhiprtc.so failed)"; return false; }
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    create = reinterpret_cast<decltype(create)>(sym("hiprtcCreateProgram"));
    compile = reinterpret_cast<decltype(compile)>(sym("hiprtcCompileProgram"));
    log_size = reinterpret_cast<decltype(log_size)>(sym("hiprtcGetProgramLogSize"));
    log = reinterpret_cast<decltype(log)>(sym("hiprtcGetProgramLog"));
    code_size = reinterpret_cast<decltype(code_size)>(sym("hiprtcGetCodeSize"));
    code = reinterpret_cast<decltype(code)>(sym("hiprtcGetCode"));
    destroy = reinterpret_cast<decltype(destroy)>(sym("hiprtcDestroyProgram"));
    add_name = reinterpret_cast<decltype(add_name)>(sym("hiprtcAddNameExpression"));
    lowered = reinterpret_cast<decltype(lowered)>(sym("hiprtcGetLoweredName"));
    if (!create || !compile || !log_size || !log || !code_size || !code || !destroy || !add_name || !lowered) {
      err = "hiprtc entry points missing";
      dlclose(lib); lib = nullptr;
      return false;
    }
    return true;
  }
};

struct Header { const char* name; const char* text; };

// Compiles `source` (with the embedded headers) for gfx950; on success `image` is the code object and
// `lowered_names[i]` the mangled name of name_exprs[i].
inline bool compile(Rtc& rtc, const std::string& source, const std::vector<Header>& headers,
                    const std::vector<std::string>& name_exprs, std::vector<char>& image,
                    std::vector<std::string>& lowered_names, std::string& err, const char* opt_level = "-O3") {
  if (!rtc.open(err)) return false;
  std::vector<const char*> hn, ht;
  for (const Header& h : headers) { hn.push_back(h.name); ht.push_back(h.text); }
  void* prog = nullptr;
  if (rtc.create(&prog, source.c_str(), "demi_k1_jit.hip", (int)headers.size(), ht.data(), hn.data()) != 0) {
    err = "hiprtcCreateProgram failed";
    return false;
  }
  for (const std::string& n : name_exprs) rtc.add_name(prog, n.c_str());
  std::vector<const char*> opts = {"--offload-arch=gfx950", opt_level, "-std=c++17", "-Wno-unused-label"};
  std::vector<std::string> extra;                            // experiment knob: DEMI_JIT_FLAGS = extra compiler options, space separated
  if (const char* f = demi_host::knob("DEMI_JIT_FLAGS")) {
    std::string item;
    for (const char* c = f;; c++) {
      if (*c == ' ' || *c == '\0') { if (!item.empty()) extra.push_back(item); item.clear(); if (*c == '\0') break; }
      else item += *c;
    }
  }
  for (const std::string& x : extra) opts.push_back(x.c_str());
  const int rc = rtc.compile(prog, (int)opts.size(), opts.data());
  if (rc != 0) {
    size_t ls = 0;
    rtc.log_size(prog, &ls);
    std::string lg(ls + 1, '\0');
    if (ls) rtc.log(prog, &lg[0]);
    err = "hiprtcCompileProgram failed: " + lg.substr(0, 1500);
    rtc.destroy(&prog);
    return false;
  }
  lowered_names.clear();
  for (const std::string& n : name_exprs) {
    const char* ln = nullptr;
    if (rtc.lowered(prog, n.c_str(), &ln) != 0 || !ln) { err = "hiprtcGetLoweredName failed for " + n; rtc.destroy(&prog); return false; }
    lowered_names.push_back(ln);
  }
  size_t cs = 0;
  rtc.code_size(prog, &cs);
  image.resize(cs);
  if (cs == 0 || rtc.code(prog, image.data()) != 0) { err = "hiprtcGetCode failed"; rtc.destroy(&prog); return false; }
  rtc.destroy(&prog);
  return true;
}

}  // namespace demi_jit
