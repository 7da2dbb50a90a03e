// hiprtc_emu.cpp - the run-time compiler of the wave64 emulator (TEST INFRASTRUCTURE, see wave64_emu.hpp).
//
// demi_model_specialize reaches hiprtc through dlopen (demi_amd/csrc/jit.hpp, DEMI_HIPRTC_LIB names the library).  This one has
// the same entry points and compiles the SAME translation unit - the kernel headers as the library embeds them plus the code
// generated for the loaded table - with g++ against the emulator's <hip/hip_runtime.h>.  The "code object" it returns is the
// path of a shared object; the lowered name of a kernel is an entry that takes hipModuleLaunchKernel's parameter array.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

#ifndef W64_EMU_DIR
#error "build with -DW64_EMU_DIR=\"<repo>/tests/emu\""
#endif

namespace {
struct Program {
  std::string source;
  std::vector<std::pair<std::string, std::string>> headers;   // (include name, text)
  std::vector<std::string> names, lowered;
  std::string log, so_path;
};
uint64_t fnv(uint64_t h, const std::string& s) {
  for (unsigned char c : s) h = (h ^ c) * 0x100000001B3ULL;
  return (h ^ 0xFF) * 0x100000001B3ULL;
}
void mkdirs(const std::string& path) {      // every directory of `path` (a file name)
  for (size_t i = 1; i < path.size(); i++)
    if (path[i] == '/') mkdir(path.substr(0, i).c_str(), 0700);
}
bool write_file(const std::string& path, const std::string& text) {
  mkdirs(path);
  FILE* f = fopen(path.c_str(), "w");
  if (!f) return false;
  const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
  return fclose(f) == 0 && ok;
}
}  // namespace

extern "C" {

int hiprtcCreateProgram(void** prog, const char* src, const char*, int n_headers, const char* const* texts, const char* const* names) {
  Program* p = new Program();
  p->source = src;
  for (int i = 0; i < n_headers; i++) p->headers.emplace_back(names[i], texts[i]);
  *prog = p;
  return 0;
}
int hiprtcAddNameExpression(void* prog, const char* expr) { static_cast<Program*>(prog)->names.push_back(expr); return 0; }

int hiprtcCompileProgram(void* prog, int n_opts, const char* const* opts) {
  Program* p = static_cast<Program*>(prog);
  std::string flags;
  for (int i = 0; i < n_opts; i++) {
    const std::string o = opts[i];
    if (o.rfind("-D", 0) == 0 || o.rfind("-U", 0) == 0) flags += " '" + o + "'";      // (macro definitions are the options that mean the same here)
  }
  std::string tu = p->source;
  tu += "\n// ---- entries (hiprtc_emu.cpp)\n#include <utility>\n#include <type_traits>\nnamespace w64jit {\n"
        "template <class... A, size_t... I> static void call(void (*f)(A...), void** p, std::index_sequence<I...>) {\n"
        "  f(*static_cast<std::remove_cv_t<std::remove_reference_t<A>>*>(p[I])...); }\n"
        "template <class... A> static void invoke(void (*f)(A...), void** p) { call(f, p, std::index_sequence_for<A...>{}); }\n}\n";
  p->lowered.clear();
  for (size_t i = 0; i < p->names.size(); i++) {
    const std::string entry = "w64k_" + std::to_string(i);
    tu += "extern \"C\" void " + entry + "(void** p) { w64jit::invoke(&" + p->names[i] + ", p); }\n";
    p->lowered.push_back(entry);
  }
  uint64_t h = fnv(0xCBF29CE484222325ULL, tu);
  for (const auto& hd : p->headers) { h = fnv(h, hd.first); h = fnv(h, hd.second); }
  h = fnv(h, flags);
  const char* base = getenv("W64_JIT_CACHE");
  char dir[512];
  snprintf(dir, sizeof dir, "%s/w64jit-%u/%016llx", base ? base : "/tmp", (unsigned)getuid(), (unsigned long long)h);
  // the sources two levels down, so that "../../include/demi_gpu.h" of the embedded headers stays inside the directory
  const std::string src_dir = std::string(dir) + "/demi_amd/csrc/";
  p->so_path = std::string(dir) + "/kernels.so";
  if (access(p->so_path.c_str(), R_OK) == 0) return 0;          // compiled before (same sources, same flags)
  if (!write_file(src_dir + "tu.cpp", tu)) { p->log = "cannot write " + src_dir; return 1; }
  for (const auto& hd : p->headers)
    if (!write_file(src_dir + hd.first, hd.second)) { p->log = "cannot write header " + hd.first; return 1; }
  const std::string tmp_so = p->so_path + "." + std::to_string((long)getpid());
  const std::string log_path = std::string(dir) + "/log." + std::to_string((long)getpid());
  const std::string cmd = std::string("g++ -O1 -std=c++17 -shared -fPIC -w -I'") + W64_EMU_DIR + "'" + flags + " -o '" + tmp_so + "' '" + src_dir +
                          "tu.cpp' -L'" + W64_EMU_DIR + "/_build' -lw64rt -Wl,-rpath,'" + W64_EMU_DIR + "/_build' > '" + log_path + "' 2>&1";
  const int rc = system(cmd.c_str());
  if (rc != 0) {
    if (FILE* f = fopen(log_path.c_str(), "r")) {
      char buf[4096];
      const size_t n = fread(buf, 1, sizeof buf - 1, f);
      buf[n] = 0;
      p->log = buf;
      fclose(f);
    }
    p->log = "g++ failed (" + cmd + "): " + p->log;
    return 1;
  }
  unlink(log_path.c_str());
  if (rename(tmp_so.c_str(), p->so_path.c_str()) != 0) { p->log = "rename failed"; return 1; }
  return 0;
}

int hiprtcGetProgramLogSize(void* prog, size_t* n) { *n = static_cast<Program*>(prog)->log.size() + 1; return 0; }
int hiprtcGetProgramLog(void* prog, char* out) { const std::string& l = static_cast<Program*>(prog)->log; memcpy(out, l.c_str(), l.size() + 1); return 0; }
int hiprtcGetCodeSize(void* prog, size_t* n) { *n = static_cast<Program*>(prog)->so_path.size() + 1; return 0; }
int hiprtcGetCode(void* prog, char* out) { const std::string& s = static_cast<Program*>(prog)->so_path; memcpy(out, s.c_str(), s.size() + 1); return 0; }
int hiprtcGetLoweredName(void* prog, const char* expr, const char** out) {
  Program* p = static_cast<Program*>(prog);
  for (size_t i = 0; i < p->names.size(); i++)
    if (p->names[i] == expr && i < p->lowered.size()) { *out = p->lowered[i].c_str(); return 0; }
  return 1;
}
int hiprtcDestroyProgram(void** prog) { delete static_cast<Program*>(*prog); *prog = nullptr; return 0; }

}  // extern "C"
