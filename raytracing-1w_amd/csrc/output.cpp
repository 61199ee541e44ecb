/* output.cpp -- the output side of the reference on the host: Color::into_sampled
 * (src/color.rs:14-21), Display for SampledColor (src/color.rs:56-65) and the P3
 * writer of src/main.rs:953,1003-1007.  Integer/text work, kept off the GPU
 * (SURVEY.md section 8f ranks a device-side quantiser as "next"). */
#include <cstdio>
#include <cstring>

#include "rt1w.h"
#include "rt_core.h"
#include "scene.h"

extern "C" {

int rt1w_resolve(const double* sums, uint64_t n_pixels, uint32_t spp, double* means) {
    if (!sums || !means || spp == 0) { rt1w::set_error("bad argument"); return RT1W_ERR_INVALID; }
    for (uint64_t i = 0; i < n_pixels; ++i) {
        RtV3 m = rt_into_sampled(rt_v3(sums[i * 3], sums[i * 3 + 1], sums[i * 3 + 2]), spp);
        means[i * 3] = m.x; means[i * 3 + 1] = m.y; means[i * 3 + 2] = m.z;
    }
    return RT1W_OK;
}

int rt1w_quantize(const double* means, uint64_t n_values, uint8_t* out) {
    if (!means || !out) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    for (uint64_t i = 0; i < n_values; ++i) out[i] = (uint8_t)rt_quantize(means[i]);
    return RT1W_OK;
}

int64_t rt1w_format_ppm(const double* means, uint32_t width, uint32_t height, char* buf, uint64_t cap) {
    if (!means) { rt1w::set_error("null argument"); return RT1W_ERR_INVALID; }
    char line[64];
    /* println!("P3\n{} {}\n255", image_width, image_height) main.rs:953 */
    int n = std::snprintf(line, sizeof line, "P3\n%u %u\n255\n", width, height);
    uint64_t pos = 0;
    auto put = [&](const char* s, int len) {
        if (buf && pos + (uint64_t)len <= cap) std::memcpy(buf + pos, s, (size_t)len);
        pos += (uint64_t)len;
    };
    put(line, n);
    /* rows are collected from (0..H).rev(): j = H-1 first (main.rs:957-960,1003-1007) */
    for (uint32_t r = 0; r < height; ++r) {
        uint32_t j = height - 1u - r;
        for (uint32_t i = 0; i < width; ++i) {
            const double* c = means + ((uint64_t)j * width + i) * 3u;
            n = std::snprintf(line, sizeof line, "%u %u %u\n", rt_quantize(c[0]), rt_quantize(c[1]), rt_quantize(c[2]));
            put(line, n);
        }
    }
    if (buf) {
        if (pos + 1 > cap) { rt1w::set_error("buffer too small"); return RT1W_ERR_INVALID; }
        buf[pos] = '\0';
    }
    return (int64_t)pos;
}

} /* extern "C" */
