/* jit.h -- topology-specialised render kernels (jit.cpp): source generation, hiprtc, kernel cache. */
#ifndef RT1W_JIT_H
#define RT1W_JIT_H

#include <string>
#include <vector>

#include "scene.h"

namespace rt1w {

struct JitInfo {
    std::string key;      /* hash of source + embedded headers + compiler options + hiprtc version */
    std::string path;     /* cache file the code came from / went to ("" if none) */
    std::string message;  /* why it failed (compiler log, ...) */
    bool from_cache = false;
    double compile_ms = 0.0;
};

bool jit_eligible(const rt1w_scene& s);               /* committed, <= RT_SWEEP_MAX_NODES nodes */
std::string jit_source(const rt1w_scene& s, bool f32 = false); /* the generated translation unit (f32: the RT_F32 build of it) */
std::string jit_key(const std::string& source);
int jit_compile(const std::string& source, std::vector<char>& code, std::string& log); /* hiprtc; RT1W_ERR_UNSUPPORTED without libhiprtc */
/* code object for a generated source: <libdir>/kernels, then the user cache, then (if allowed) the compiler */
int jit_get_code(const std::string& source, bool allow_compile, std::vector<char>& code, JitInfo& info, bool ignore_cache = false);
void jit_invalidate(const JitInfo& info); /* unlink a cached object the driver refused (user cache only) */
/* build step: compile into `dir` unless already there */
int jit_precompile_to(const rt1w_scene& s, const std::string& dir, JitInfo& info, bool f32 = false);

} // namespace rt1w

#endif
