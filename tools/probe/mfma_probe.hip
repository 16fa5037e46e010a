// Probe of v_mfma_f32_32x32x16_f16's internal arithmetic on gfx950: how are the 16
// products and C accumulated (exact? sequential fp32? alignment width? rounding?).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_probe.hip -o tools/probe/mfma_probe
// Output is quoted in DESIGN.md and asserted by tests/test_gpu_parity.py::test_mfma_f16_accumulation_model.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// a[16], b[16]: the k-vectors of output element (0,0); c: its C input.  Every row of A gets a[],
// every column of B gets b[] (so all 1024 outputs are the same number).
__global__ void probe(const float *a, const float *b, float c, float *out)
{
    const int lane = threadIdx.x, half = lane >> 5;
    f16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)a[half * 8 + j]; fb[j] = (_Float16)b[half * 8 + j]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
    if (lane == 37) out[1] = acc[5];
}

static float run(const float *a, const float *b, float c)
{
    float *da, *db, *dout, h[2];
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 8);
    hipMemcpy(da, a, 64, hipMemcpyHostToDevice); hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, c, dout);
    hipMemcpy(h, dout, 8, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    if (h[0] != h[1]) printf("  (!! elements differ: %.9g vs %.9g)\n", h[0], h[1]);
    return h[0];
}

static void fill(float *v, float x) { for (int i = 0; i < 16; ++i) v[i] = x; }

int main()
{
    float a[16], b[16];
    // T1: 2^24 first, then fifteen 1s
    fill(a, 1.f); fill(b, 1.f); a[0] = 4096.f; b[0] = 4096.f;
    printf("T1 big-first  [2^24, 1 x15] + 0        = %.1f   (exact 16777231; seq-RNE 16777216; exact+RNE 16777232; exact+trunc 16777230)\n", run(a, b, 0.f));
    // T2: big last
    fill(a, 1.f); fill(b, 1.f); a[15] = 4096.f; b[15] = 4096.f;
    printf("T2 big-last   [1 x15, 2^24] + 0        = %.1f\n", run(a, b, 0.f));
    // T3: C big, sixteen 1s
    fill(a, 1.f); fill(b, 1.f);
    printf("T3 C=2^24     [1 x16] + 2^24           = %.1f   (exact 16777232; seq-RNE 16777216)\n", run(a, b, 16777216.f));
    // T3b: C = 2^24, three 1s (exact 2^24+3: RNE -> +4, trunc -> +2)
    fill(a, 0.f); fill(b, 0.f); a[0] = a[1] = a[2] = 1.f; b[0] = b[1] = b[2] = 1.f;
    printf("T3b C=2^24    [1 x3] + 2^24            = %.1f   (exact 16777219; RNE 16777220; trunc 16777218)\n", run(a, b, 16777216.f));
    // T3c: C = 2^24, one 1 (tie: RNE-even -> 2^24; round-half-up -> +2)
    fill(a, 0.f); fill(b, 0.f); a[0] = 1.f; b[0] = 1.f;
    printf("T3c C=2^24    [1] + 2^24               = %.1f   (tie: RNE 16777216)\n", run(a, b, 16777216.f));
    // T4: cancellation: 2^24 - 2^24 + fourteen small terms of 2^-s: how far below the largest exponent do addends survive?
    for (int s = 0; s <= 30; s += (s < 6 ? 1 : 6)) {
        fill(a, 1.f); fill(b, 1.f);
        a[0] = 4096.f; b[0] = 4096.f; a[1] = 4096.f; b[1] = -4096.f;
        const float small = ldexpf(1.f, -(s / 2)), small2 = ldexpf(1.f, -(s - s / 2));
        for (int k = 2; k < 16; ++k) { a[k] = small; b[k] = small2; }
        printf("T4 s=%2d       [2^24, -2^24, 2^-%d x14] + 0 = %.9g   (exact %.9g)\n", s, s, run(a, b, 0.f), 14.0 * ldexp(1.0, -s));
    }
    // T5: same with the big pair in C and a product: C = 2^24, product -2^24, small terms
    for (int s = 0; s <= 30; s += 6) {
        fill(a, 1.f); fill(b, 1.f);
        a[0] = 4096.f; b[0] = -4096.f;
        const float small = ldexpf(1.f, -(s / 2)), small2 = ldexpf(1.f, -(s - s / 2));
        for (int k = 1; k < 16; ++k) { a[k] = small; b[k] = small2; }
        printf("T5 s=%2d       [-2^24, 2^-%d x15] + 2^24    = %.9g   (exact %.9g)\n", s, s, run(a, b, 16777216.f), 15.0 * ldexp(1.0, -s));
    }
    // T7: one big term + ONE small term in the same 8-group, no cancellation: what survives into the rounding?
    //     2^24 + 2^0 * 1 + (2^-1 ...): sticky information.  [2^24, 1, 2^-s] -> exact 2^24 + 1 + 2^-s: RNE gives +2 if the
    //     2^-s is seen (above the tie), +0 (tie to even) if it was truncated away before the rounding
    for (int s = 1; s <= 12; s += (s < 4 ? 1 : 4)) {
        fill(a, 0.f); fill(b, 0.f);
        a[0] = 4096.f; b[0] = 4096.f; a[1] = 1.f; b[1] = 1.f; a[2] = ldexpf(1.f, -s); b[2] = 1.f;
        printf("T7 s=%2d       [2^24, 1, 2^-%d] + 0          = %.1f   (sticky kept: 16777218; lost: 16777216)\n", s, s, run(a, b, 0.f));
    }
    // T8: negative small terms under cancellation (direction of the truncation)
    for (int s = 0; s <= 3; ++s) {
        fill(a, 1.f); fill(b, -1.f);
        a[0] = 4096.f; b[0] = 4096.f; a[1] = 4096.f; b[1] = -4096.f;
        for (int k = 2; k < 16; ++k) { a[k] = ldexpf(1.f, -s); b[k] = -1.f; }
        printf("T8 s=%2d       [2^24, -2^24, -2^-%d x14] + 0 = %.9g   (exact %.9g)\n", s, s, run(a, b, 0.f), -14.0 * ldexp(1.0, -s));
    }
    // T9: terms in the OTHER 8-group than the big pair: are the two groups aligned to a common exponent?
    for (int s = 0; s <= 6; s += 3) {
        fill(a, 0.f); fill(b, 0.f);
        a[0] = 4096.f; b[0] = 4096.f;                       // 2^24 in group 0, nothing cancels it
        for (int k = 8; k < 16; ++k) { a[k] = ldexpf(1.f, -s); b[k] = 1.f; }   // 8 * 2^-s in group 1
        printf("T9 s=%2d       [2^24 | 2^-%d x8] + -2^24     = %.9g   (exact %.9g)\n", s, s, run(a, b, -16777216.f), 8.0 * ldexp(1.0, -s));
    }
    // T10: is the final (group 0 + group 1 + C) addition exact before the single rounding?  C = 2^24, group 0 = 1,
    //      group 1 = 2^-s: exact 2^24 + 1 + 2^-s rounds (RNE) to +2 iff the 2^-s is still there
    for (int s = 1; s <= 25; s += (s < 3 ? 1 : 11)) {
        fill(a, 0.f); fill(b, 0.f);
        a[0] = 1.f; b[0] = 1.f; a[8] = ldexpf(1.f, -(s / 2)); b[8] = ldexpf(1.f, -(s - s / 2));
        printf("T10 s=%2d      [1 | 2^-%d] + 2^24            = %.1f   (exact-then-RNE: 16777218; addends cut at 2^0: 16777216)\n", s, s, run(a, b, 16777216.f));
    }
    // T11: same but the small term comes from C: group 0 = 2^24, group 1 = 1, C = 2^-s
    for (int s = 1; s <= 25; s += 12) {
        fill(a, 0.f); fill(b, 0.f);
        a[0] = 4096.f; b[0] = 4096.f; a[8] = 1.f; b[8] = 1.f;
        printf("T11 s=%2d      [2^24 | 1] + 2^-%d             = %.1f   (exact-then-RNE: 16777218)\n", s, s, run(a, b, ldexpf(1.f, -s)));
    }
    // T6: f16 subnormal inputs: flushed?
    fill(a, 0.f); fill(b, 0.f); a[0] = ldexpf(1.f, -20); b[0] = 1024.f;   // 2^-20 is an f16 subnormal
    printf("T6 subnormal  [2^-20 * 2^10] + 0        = %.9g   (exact %.9g; 0 if f16 subnormals are flushed)\n", run(a, b, 0.f), ldexp(1.0, -10));
    return 0;
}
