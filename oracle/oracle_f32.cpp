/* oracle_f32.cpp -- the literal oracle (oracle.cpp) with the reference's own precision switch thrown: `type Float = f64;` -> f32
 * (/root/reference/src/main.rs:1; mirrored at oracle.cpp `typedef double Float`).  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * It is the expected side of the product's RT1W_PRECISION_F32 mode (raytracing-1w_amd/csrc/context_f32.hip), whose parity is
 * statistical by nature.  How the switch is thrown: every `double` of oracle.cpp and include/rt1w_num.h becomes `float` (the same
 * preprocessor switch the device build uses, RT_F32), so every vector, ray, hit record, box, camera, colour and pdf of the restatement
 * is f32.  What stays 64-bit, stated so that nobody mistakes this for a bit-level model of a Rust f32 build:
 *   - generator word -> number conversions: the f64 draw of rand 0.8's shapes (rt_gen_f64, rt_gen_range), rounded to f32 where it is
 *     assigned -- exactly what context_f32.hip does (rand's own f32 shapes, 24-bit draws from ONE u32, would consume the word stream
 *     differently and the f32 frame would stop sharing its paths with the f64 frame);
 *   - the internals of sin / cos / tan / acos / atan2 / ln (include/rt1w_num.h, spelled rt_f64), rounded once;
 *   - untyped literals are C++ doubles: a subexpression that contains one is evaluated in f64 and rounded on assignment (a single
 *     + - * / of f32 operands evaluated in f64 and rounded is the correctly rounded f32 result; longer chains can differ in the last
 *     f32 ulp from pure f32 evaluation);
 * The per-pixel sums (`pixel_color`, main.rs:966-989) are f32 here, as the reference's switch makes them; the product accumulates in
 * f64 ("f32 traversal and shading, f64 accumulation") -- a difference of ~1e-5 relative at 1000 spp, far inside the test's bounds.
 * The C entry points keep their names; their `double` parameters are `float` in this library (tests/orc.py binds them so).
 * Build: make -C oracle liborc_f32.so.  Loaded only by tests/ (tests/orc.py: OracleScene(f32=True)) and tests/golden/make_golden.py. */
#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
#include <stdint.h>

typedef double rt_f64; /* what stays 64-bit (include/rt1w_num.h) */
#define RT_F32 1
#define ORC_F32 1
#define double float
#include "oracle.cpp"
