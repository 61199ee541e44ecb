/* jit.cpp -- topology-specialised render kernels, compiled at run time.
 *
 * For a scene of <= RT_JIT_MAX_NODES nodes the pre-order sweep can be unrolled along the scene's own
 * tree (rt_sweep_static in rt_core.h) once the node kinds and subtree ends are compile-time constants.
 * They are only known when a scene has been committed, so the kernel is generated then: this file writes
 * a ten-line translation unit (the topology as constexpr arrays + one extern "C" kernel around
 * rt_render_sorted_body), hands it and the library's own headers (embedded at build time,
 * jit_headers.inc) to hiprtc, and keeps the code object in a cache keyed by a hash of everything that
 * went into the compiler.  The arithmetic of a path is the same source as in the generic kernels; only
 * control flow is resolved earlier, so results are bit-identical (tests/test_gpu_parity.py).
 *
 * libhiprtc is opened with dlopen: the library has no link-time dependency on it, and a host without
 * it simply keeps the generic kernels.
 */
#include "jit.h"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "jit_headers.inc"

namespace rt1w {

namespace {

/* ---- the slice of the hiprtc API that is used (resolved with dlsym) ---- */
typedef struct _hiprtcProgram* hiprtcProgram;
typedef int hiprtcResult;
struct Hiprtc {
    void* lib = nullptr;
    hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char* const*) = nullptr;
    hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetCode)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*DestroyProgram)(hiprtcProgram*) = nullptr;
    hiprtcResult (*Version)(int*, int*) = nullptr; /* optional */
    bool tried = false, ok = false;
};
Hiprtc g_rtc;

bool load_hiprtc(std::string& err) {
    if (g_rtc.tried) { if (!g_rtc.ok) err = "libhiprtc.so is not available"; return g_rtc.ok; }
    g_rtc.tried = true;
    const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    if (!std::getenv("RT1W_NO_HIPRTC")) /* tests: a host without the run-time compiler */
        for (const char* n : names) { g_rtc.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_rtc.lib) break; }
    if (!g_rtc.lib) { err = "libhiprtc.so is not available"; return false; }
#define RT_SYM(field, name) *(void**)(&g_rtc.field) = dlsym(g_rtc.lib, name); if (!g_rtc.field) { err = std::string("libhiprtc.so lacks ") + name; return false; }
    RT_SYM(CreateProgram, "hiprtcCreateProgram")
    RT_SYM(CompileProgram, "hiprtcCompileProgram")
    RT_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize")
    RT_SYM(GetProgramLog, "hiprtcGetProgramLog")
    RT_SYM(GetCodeSize, "hiprtcGetCodeSize")
    RT_SYM(GetCode, "hiprtcGetCode")
    RT_SYM(DestroyProgram, "hiprtcDestroyProgram")
#undef RT_SYM
    *(void**)(&g_rtc.Version) = dlsym(g_rtc.lib, "hiprtcVersion");
    g_rtc.ok = true;
    return true;
}

uint64_t fnv1a(uint64_t h, const void* data, size_t n) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
uint64_t fnv1a(uint64_t h, const std::string& s) { return fnv1a(h, s.data(), s.size()); }

/* compiler options: fixed, plus -DRT_STAMPS=1 with RT1W_JIT_STAMPS in the environment (per-phase cycle counters,
 * tools/stamps.py), plus the space-separated words of RT1W_JIT_EXTRA_OPTS (compiler-flag experiments; never anything that
 * changes floating-point semantics).  All of them go into the cache key. */
std::vector<std::string> options() {
    /* -structurizecfg-skip-uniform-regions: the unrolled sweep is a chain of wave-uniform branches ("is any lane of the wave at this
     * node?") around divergent bodies; left alone, StructurizeCFG rewrites the uniform ones as well and they cost exec-mask
     * bookkeeping.  Measured on the Cornell kernel: 155 VGPRs instead of 158, +2.0 % (profiles/r04_jit_options.txt: 39 backend
     * options and their combinations; nothing else is worth a flag).  Codegen only: the frames are the same bits (every GPU parity
     * test runs these kernels). */
    std::vector<std::string> o = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DRT_JIT=1",
                                  "-mllvm", "-structurizecfg-skip-uniform-regions=1",
                                  /* the hipcc-built kernels' other options (csrc/Makefile: KOPTS): neutral at three waves per SIMD,
                                   * but they take the Cornell kernel from 155 to 143 VGPRs -- what makes the fourth wave pay (jit_source) */
                                  "-mllvm", "-spec-exec-max-speculation-cost=0", "-mllvm", "-simplifycfg-hoist-common=false",
                                  "-mllvm", "-amdgpu-sdwa-peephole=0",
                                  /* no partial-redundancy elimination in GVN: Cornell +0.8 % (twice measured), the other arms +-0.5 % */
                                  "-mllvm", "-enable-pre=0"};
    if (std::getenv("RT1W_JIT_STAMPS")) o.push_back("-DRT_STAMPS=1");
    if (const char* e = std::getenv("RT1W_JIT_EXTRA_OPTS")) {
        std::string w;
        for (const char* p = e;; ++p) {
            if (*p == ' ' || *p == 0) { if (!w.empty()) o.push_back(w); w.clear(); if (!*p) break; }
            else w.push_back(*p);
        }
    }
    return o;
}

/* the compiler's identity.  It names the files of the USER cache (a code object another hiprtc / ROCm release left there is not
 * picked up); it is NOT part of a kernel's key: the kernels under <libdir>/kernels were compiled by the build's toolchain for
 * this very library (same source, headers, options = same key) and are valid whatever hiprtc the host has, or none -- a code
 * object the driver refuses is simply skipped (hipModuleLoadData fails, the generic kernels run). */
std::string compiler_id() {
    std::string err;
    if (!load_hiprtc(err)) return "hiprtc-absent";
    int major = 0, minor = 0;
    if (g_rtc.Version && g_rtc.Version(&major, &minor) == 0) return "hiprtc-" + std::to_string(major) + "." + std::to_string(minor);
    return "hiprtc-unknown";
}

bool read_file(const std::string& path, std::vector<char>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n <= 0) { std::fclose(f); return false; }
    out.resize((size_t)n);
    size_t got = std::fread(out.data(), 1, (size_t)n, f);
    std::fclose(f);
    return got == (size_t)n;
}
bool write_file_atomic(const std::string& path, const std::vector<char>& data) {
    std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    size_t put = std::fwrite(data.data(), 1, data.size(), f);
    std::fclose(f);
    if (put != data.size() || std::rename(tmp.c_str(), path.c_str()) != 0) { std::remove(tmp.c_str()); return false; }
    return true;
}
void mkdirs(const std::string& dir) {
    std::string cur;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || dir[i] == '/') { if (!cur.empty()) (void)mkdir(cur.c_str(), 0755); }
        if (i < dir.size()) cur.push_back(dir[i]);
    }
}

/* <directory of librt1w.so>/kernels: filled by the build (precompiled reference scenes), read-only at run time */
std::string install_cache_dir() {
    Dl_info info;
    if (!dladdr((const void*)&install_cache_dir, &info) || !info.dli_fname) return std::string();
    std::string p = info.dli_fname;
    size_t k = p.rfind('/');
    return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/kernels";
}
std::string user_cache_tag() {
    std::string id = compiler_id();
    for (char& ch : id) if (!((ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || ch == '.')) ch = '_';
    return id;
}
/* RT1W_KERNEL_CACHE, or ~/.cache/rt1w: where kernels compiled at run time are kept */
std::string user_cache_dir() {
    if (const char* e = std::getenv("RT1W_KERNEL_CACHE")) return e;
    if (const char* h = std::getenv("HOME")) return std::string(h) + "/.cache/rt1w";
    return std::string();
}

} // namespace

bool jit_eligible(const rt1w_scene& s) {
    return s.committed && !s.flat_nodes.empty() && s.flat_nodes.size() <= RT_JIT_MAX_NODES;
}

std::string jit_source(const rt1w_scene& s, bool f32) {
    const std::vector<RtNode>& N = s.flat_nodes;
    std::string src;
    src += "/* generated by librt1w (jit.cpp) for one scene topology */\n";
    if (f32) /* the single-precision build of the same kernel: context_f32.hip's switch, here for the run-time compiler */
        src += "#include <stdint.h>\n#include <type_traits>\ntypedef double rt_f64;\n#define RT_F32 1\n#define double float\n";
    else {
        /* Four waves per SIMD (128 VGPRs) and the exchange in two rounds (27 KB of LDS per workgroup, four workgroups per CU) for the
         * scenes whose shading fits: no Perlin-noise and no checker texture.  With the options below the Cornell kernel needs 143 VGPRs
         * at three waves, so the fourth wave costs few spills and hides what three cannot: Cornell 2362 -> 2620 Mpaths/s (+11 %),
         * cornel_smoke +12 %, earth +8 %; the noise / checker scenes lose 2-6 % that way and keep three waves and one round
         * (profiles/r04_jit_options.txt) */
        bool heavy_tex = false;
        for (const RtTexture& t : s.textures) if (t.kind == RT_TEX_NOISE || t.kind == RT_TEX_CHECKER) heavy_tex = true;
        if (!heavy_tex) /* (a -DRT_SORT_WAVES_OVERRIDE=3 -DRT_XCH_PARTS=1 in RT1W_JIT_EXTRA_OPTS builds the three-wave form: A/B tools) */
            src += "#ifndef RT_SORT_WAVES_OVERRIDE\n#define RT_XCH_PARTS 2\n#define RT_SORT_WAVES_OVERRIDE 4\n#endif\n";
        src += "#include \"rt1w_num.h\"\n"; /* declares rt_f64 */
    }
    src += "#include \"rt_kernel_sorted.h\"\n";
    src += "struct TopoJit {\n";
    src += "    static constexpr uint32_t n = " + std::to_string(N.size()) + "u, root = " + std::to_string(s.flat_root) + "u;\n";
    src += "    static constexpr uint32_t kind[" + std::to_string(N.size()) + "] = {";
    for (size_t i = 0; i < N.size(); ++i) src += (i ? ", " : "") + std::to_string(N[i].kind) + "u";
    src += "};\n";
    src += "    static constexpr uint32_t skip[" + std::to_string(N.size()) + "] = {";
    for (size_t i = 0; i < N.size(); ++i) src += (i ? ", " : "") + std::to_string(N[i].skip) + "u";
    src += "};\n};\n";
    /* the features the scene actually has (code for the others is not generated, rt_flat.h RtCfg) */
    uint32_t depth = s.scope_depth < 2u ? 2u : s.scope_depth;
    src += std::string("typedef RtCfg<") + (s.has_media ? "true" : "false") + ", " + (s.has_tex ? "true" : "false") + ", " +
           (s.has_msphere ? "true" : "false") + ", true, " + std::to_string(depth) + ", TopoJit> CfgJit;\n";
    src += "extern \"C\" __global__ __launch_bounds__(RT_SORT_BLOCK, RT_SORT_WAVES(CfgJit)) void rt_jit_sorted(\n"
           "    RtSceneView sc, RtFrame f, rt_f64* __restrict__ partial, unsigned long long* __restrict__ counters) {\n"
           "    rt_render_sorted_body<CfgJit>(sc, f, partial, counters);\n}\n";
    return src;
}

std::string jit_key(const std::string& source) {
    uint64_t h = 0xcbf29ce484222325ull;
    h = fnv1a(h, source);
    for (int i = 0; i < RT_JIT_N_HEADERS; ++i) { h = fnv1a(h, rt_jit_header_names[i], std::strlen(rt_jit_header_names[i])); h = fnv1a(h, rt_jit_header_texts[i], std::strlen(rt_jit_header_texts[i])); }
    for (const std::string& o : options()) h = fnv1a(h, o);
    char buf[32];
    std::snprintf(buf, sizeof buf, "%016llx", (unsigned long long)h);
    return buf;
}

int jit_compile(const std::string& source, std::vector<char>& code, std::string& log) {
    static std::mutex mu; /* one compile at a time per process (contexts may be created from several host threads) */
    std::lock_guard<std::mutex> lock(mu);
    std::string err;
    if (!load_hiprtc(err)) { log = err; return RT1W_ERR_UNSUPPORTED; }
    hiprtcProgram prog = nullptr;
    if (g_rtc.CreateProgram(&prog, source.c_str(), "rt_jit_sorted.hip", RT_JIT_N_HEADERS, rt_jit_header_texts, rt_jit_header_names) != 0) {
        log = "hiprtcCreateProgram failed"; return RT1W_ERR_DEVICE;
    }
    const std::vector<std::string> opts = options();
    std::vector<const char*> optv;
    for (const std::string& o : opts) optv.push_back(o.c_str());
    hiprtcResult rc = g_rtc.CompileProgram(prog, (int)optv.size(), optv.data());
    size_t ls = 0;
    g_rtc.GetProgramLogSize(prog, &ls);
    log.assign(ls, '\0');
    if (ls) g_rtc.GetProgramLog(prog, &log[0]);
    if (rc != 0) { g_rtc.DestroyProgram(&prog); if (log.empty()) log = "hiprtcCompileProgram failed"; return RT1W_ERR_DEVICE; }
    size_t cs = 0;
    g_rtc.GetCodeSize(prog, &cs);
    code.resize(cs);
    if (cs) g_rtc.GetCode(prog, code.data());
    g_rtc.DestroyProgram(&prog);
    if (!cs) { log = "hiprtc produced no code"; return RT1W_ERR_DEVICE; }
    return RT1W_OK;
}

/* a cached code object the driver refused (truncated file, foreign toolchain): drop it from the user cache so that the next
 * request compiles again; files under <libdir>/kernels belong to the installation and are left alone */
void jit_invalidate(const JitInfo& info) {
    const std::string u = user_cache_dir();
    if (!info.path.empty() && !u.empty() && info.path.compare(0, u.size(), u) == 0) std::remove(info.path.c_str());
}

int jit_get_code(const std::string& src, bool allow_compile, std::vector<char>& code, JitInfo& info, bool ignore_cache) {
    info = JitInfo();
    if (src.empty()) { info.message = "scene is not eligible (more than RT_JIT_MAX_NODES nodes)"; return RT1W_ERR_UNSUPPORTED; }
    info.key = jit_key(src);
    /* 1. the installation's kernels (written by the build, rt1w_precompile): keyed by source + headers + options only */
    const std::string inst = install_cache_dir();
    if (!inst.empty() && !ignore_cache && read_file(inst + "/sweep_" + info.key + ".hsaco", code)) {
        info.from_cache = true; info.path = inst + "/sweep_" + info.key + ".hsaco"; return RT1W_OK;
    }
    /* 2. the user cache: files carry the compiler that made them (this is the first point that opens libhiprtc) */
    const std::string user = user_cache_dir();
    const std::string uname = "/sweep_" + info.key + "-" + user_cache_tag() + ".hsaco";
    if (!user.empty() && !ignore_cache && read_file(user + uname, code)) { info.from_cache = true; info.path = user + uname; return RT1W_OK; }
    if (!allow_compile) { info.message = "no cached kernel for this topology"; return RT1W_ERR_STATE; }
    auto t0 = std::chrono::steady_clock::now();
    std::string log;
    int rc = jit_compile(src, code, log);
    info.compile_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc < 0) { info.message = log; return rc; }
    if (!user.empty()) {
        mkdirs(user);
        if (write_file_atomic(user + uname, code)) info.path = user + uname;
    }
    return RT1W_OK;
}

int jit_precompile_to(const rt1w_scene& s, const std::string& dir, JitInfo& info, bool f32) {
    info = JitInfo();
    if (!jit_eligible(s)) { info.message = "scene is not eligible"; return RT1W_ERR_UNSUPPORTED; }
    const std::string src = jit_source(s, f32);
    info.key = jit_key(src);
    const std::string path = dir + "/sweep_" + info.key + ".hsaco";
    std::vector<char> code;
    if (read_file(path, code)) { info.from_cache = true; info.path = path; return RT1W_OK; }
    auto t0 = std::chrono::steady_clock::now();
    std::string log;
    int rc = jit_compile(src, code, log);
    info.compile_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc < 0) { info.message = log; return rc; }
    mkdirs(dir);
    if (!write_file_atomic(path, code)) { info.message = "cannot write " + path; return RT1W_ERR_STATE; }
    info.path = path;
    /* which compiler made the directory's kernels: for the record only, no lookup reads it */
    const std::string note = compiler_id() + "\n";
    (void)write_file_atomic(dir + "/COMPILER", std::vector<char>(note.begin(), note.end()));
    return RT1W_OK;
}

} // namespace rt1w

extern "C" int rt1w_scene_kernel_key(const rt1w_scene* s, char out[24]) {
    if (!s || !out) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    if (!s->committed) { rt1w::set_error("scene not committed"); return RT1W_ERR_STATE; }
    if (!rt1w::jit_eligible(*s)) { rt1w::set_error("scene has more than RT_JIT_MAX_NODES nodes: no specialised kernel"); return RT1W_ERR_UNSUPPORTED; }
    std::snprintf(out, 24, "%s", rt1w::jit_key(rt1w::jit_source(*s)).c_str());
    return RT1W_OK;
}
