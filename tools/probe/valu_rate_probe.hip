// VALU issue-rate probe on gfx950: how many lane-operations per second do v_sad_u16 / v_sad_u8 / v_add_f32 /
// v_pk_add_f32 / v_add_f32 |x| sustain?  (Decides whether an integer sum-of-absolute-differences prefilter can
// beat the fp32 L1 broadcast-subtract kernel, which is bound by 2 VALU ops per (pair, k).)
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/valu_rate_probe.hip -o tools/probe/valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP 4096
template <int OP>
__global__ void probe(unsigned *out, unsigned a0, unsigned b0)
{
    unsigned acc[8];
    unsigned a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = i;
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {       // 8 independent chains: issue bound, not latency bound
            if (OP == 0) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            else if (OP == 1) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            else if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
            else if (OP == 3) asm volatile("v_add_f32 %0, %0, |%1|" : "+v"(acc[i]) : "v"(a));
            else if (OP == 4) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            else if (OP == 5) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void probe_pk(float *out, float a0)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc[8];
    f2 a = {a0 + threadIdx.x, a0 - threadIdx.x};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f2){(float)i, 1.0f};
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// OP: 0 v_pk_fma_f32 acc += a*a ; 1 v_pk_mul_f32 ; 2 v_fma_f32 (scalar, for reference) ; 3 v_pk_fma_f32 with op_sel broadcast + neg
template <int OP>
__global__ void probe_pk2(float *out, float a0)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc[8];
    f2 a = {a0 + threadIdx.x * 1e-9f, a0 - threadIdx.x * 1e-9f};
    float sacc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = (f2){(float)i, 1.0f}; sacc[i] = (float)i; }
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(a));
            else if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
            else if (OP == 2) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(sacc[i]) : "v"(a.x));
            else if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %1, %0 op_sel:[0,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(acc[i]) : "v"(a));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + sacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_it(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e-3;
}

int main()
{
    const int blocks = 256 * 8, threads = 256;
    unsigned *out; hipMalloc(&out, (size_t)blocks * threads * 4);
    const double lane_ops = (double)blocks * threads * REP * 8;
    const char *names[] = {"v_sad_u16 (2 elem/op)", "v_sad_u8 (4 elem/op)", "v_add_f32", "v_add_f32 |x|", "v_sad_u32", "v_msad_u8"};
    double t;
    t = time_it([&] { hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u); }); printf("%-24s %.2f T lane-ops/s\n", names[0], lane_ops / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u); }); printf("%-24s %.2f T lane-ops/s\n", names[1], lane_ops / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u); }); printf("%-24s %.2f T lane-ops/s\n", names[2], lane_ops / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u); }); printf("%-24s %.2f T lane-ops/s\n", names[3], lane_ops / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u); }); printf("%-24s %.2f T lane-ops/s\n", names[4], lane_ops / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u); }); printf("%-24s %.2f T lane-ops/s\n", names[5], lane_ops / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(probe_pk, dim3(blocks), dim3(threads), 0, 0, (float *)out, 1.5f); }); printf("%-24s %.2f T lane-ops/s (2 elem/op)\n", "v_pk_add_f32", lane_ops / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(probe_pk2<0>, dim3(blocks), dim3(threads), 0, 0, (float *)out, 1.0f); }); printf("%-24s %.2f T lane-ops/s (2 fma/op)\n", "v_pk_fma_f32", lane_ops / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(probe_pk2<1>, dim3(blocks), dim3(threads), 0, 0, (float *)out, 1.0f); }); printf("%-24s %.2f T lane-ops/s (2 mul/op)\n", "v_pk_mul_f32", lane_ops / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(probe_pk2<2>, dim3(blocks), dim3(threads), 0, 0, (float *)out, 1.0f); }); printf("%-24s %.2f T lane-ops/s\n", "v_fma_f32", lane_ops / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(probe_pk2<3>, dim3(blocks), dim3(threads), 0, 0, (float *)out, 1.0f); }); printf("%-24s %.2f T lane-ops/s (2 fma/op)\n", "v_pk_fma_f32 op_sel+neg", lane_ops / t / 1e12);
    return 0;
}
