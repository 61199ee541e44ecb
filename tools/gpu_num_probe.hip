// Standalone probe: are f64 / sqrt and the rt1w_num.h functions bit-identical on
// gfx950 and on the host?  (Both sides compiled with -ffp-contract=off.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "rt1w_num.h"

#define NOUT 12
RT_HD void eval(double a, double b, uint32_t i, double* o) {
    o[0] = a / b;
    o[1] = rt_sqrt(rt_abs(a));
    o[2] = rt_sin(a);
    o[3] = rt_cos(a);
    o[4] = rt_acos(a / (rt_abs(a) + 1.0));
    o[5] = rt_atan2(a, b);
    o[6] = rt_log(rt_abs(b));
    o[7] = (a * b + a) - b * 3.0;           // contraction canary
    RtRng r = rt_rng_pixel_sample(i, i * 7u + 1u, 5u);
    o[8] = rt_gen_f64(r);
    o[9] = rt_gen_range(r, -1.0, 1.0);
    o[10] = (double)rt_gen_below(r, 3u) + (rt_gen_bool(r) ? 0.5 : 0.0);
    o[11] = rt_floor(a * 1000.0) + 1.0 / a;
}
__global__ void k(const double* a, const double* b, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) eval(a[i], b[i], (uint32_t)i, o + (size_t)i * NOUT);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> a(n), b(n), ho((size_t)n * NOUT), go((size_t)n * NOUT);
    uint64_t s = 88172645463325252ull;
    auto nx = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < n; i++) {
        int m = i & 7;
        double u = (double)(nx() >> 11) * (1.0 / 9007199254740992.0);
        double v = (double)(nx() >> 11) * (1.0 / 9007199254740992.0);
        double sc = m < 2 ? 1.0 : m < 4 ? 10.0 : m < 6 ? 1e3 : 1e5;
        a[i] = (u * 2 - 1) * sc; b[i] = (v * 2 - 1) * (m & 1 ? 1e-3 : 555.0);
        if (i % 100003 == 0) { a[i] = rt_u2d(nx()); b[i] = rt_u2d(nx()); }  // raw bit patterns
    }
    for (int i = 0; i < n; i++) eval(a[i], b[i], (uint32_t)i, &ho[(size_t)i * NOUT]);
    double *da, *db, *dout;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, (size_t)n * NOUT * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(da, db, dout, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(go.data(), dout, (size_t)n * NOUT * 8, hipMemcpyDeviceToHost);
    long bad[NOUT] = {0};
    for (size_t i = 0; i < (size_t)n * NOUT; i++) {
        uint64_t x = rt_d2u(ho[i]), y = rt_d2u(go[i]);
        bool bothnan = (ho[i] != ho[i]) && (go[i] != go[i]);
        if (x != y && !bothnan) {
            if (bad[i % NOUT]++ < 3) printf("mismatch out%zu idx %zu a=%a b=%a host=%a gpu=%a\n", i % NOUT, i / NOUT, a[i / NOUT], b[i / NOUT], ho[i], go[i]);
        }
    }
    long tot = 0;
    for (int j = 0; j < NOUT; j++) { printf("out%d mismatches %ld / %d\n", j, bad[j], n); tot += bad[j]; }
    printf(tot == 0 ? "PROBE OK: bit-identical\n" : "PROBE FAIL\n");
    return tot ? 1 : 0;
}
