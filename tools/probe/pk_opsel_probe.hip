// (r06) Is  v_pk_fma_f32 D, A, B, B op_sel:[0,1,0] op_sel_hi:[1,1,0]  -- D = fma(A, B.hi, B.lo) on both lanes, the form hipcc's
// SLP vectoriser emits for  fmaf(x, z, p)  when (p, z) sit in ONE register pair -- reliable while ANOTHER wave of the same
// SIMD keeps the matrix pipe busy?  tools/probe/slp_bisect.sh traced the rare, launch-to-launch different miscounts of the
// TransH / TransD epilogue of lp_hi_stream.hip (SLP build) to exactly these instructions; this is the stand-alone check.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probe/pk_opsel_probe.hip -o tools/probe/pk_opsel_probe && tools/probe/pk_opsel_probe
//
// Per workgroup 8 waves: waves 0..3 (one per SIMD) run the packed instruction on fresh operands every iteration and compare
// with two scalar v_fma_f32 of the same operands; waves 4..7 (their SIMD partners) run back-to-back v_mfma_f32_32x32x16_f16
// (mode 1), plain VALU work (mode 2) or nothing (mode 0).  FORM 0: the op_sel form, (p, z) from a ds_read_b64 as in the
// kernel; FORM 1: the same with (p, z) already in registers; FORM 2: the plain form on materialised {z, z} / {p, p} pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FORM>
__global__ __launch_bounds__(512, 2) void probe(unsigned long long *bad, float *sink, int iters, int partner_mode,
                                                 const float4 *stream, int stream_n)
{
    __shared__ float2 pz[64 * 4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (wid >= 4) {                                 // the SIMD partners
        if ((partner_mode & 3) == 1) {
            f16x8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(lane * 0.01f + e); b[e] = (_Float16)(1.0f - e * 0.1f); }
            f32x16 acc;
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int i = 0; i < iters * 2; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
            }
            sink[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[7];
        } else if ((partner_mode & 3) == 2) {
            float v = lane * 0.5f, w = 1.0001f;
            for (int i = 0; i < iters * 16; ++i) { v = fmaf(v, w, 0.25f); w = fmaf(w, 0.9999f, 1e-4f); }
            sink[blockIdx.x * 512 + threadIdx.x] = v + w;
        }
        return;
    }
    unsigned long long nbad = 0;
    float x0 = lane * 0.37f + wid, x1 = -lane * 0.11f + 2.0f;
    float4 ring[4] = {};            // mode >= 4: global loads in flight around the packed instruction (VMEM write-backs into OTHER registers)
    float keep = 0.f;
    const int vm = partner_mode >= 4;
    for (int i = 0; i < iters; ++i) {
        if (vm) {
            keep += ring[i & 3].x;
            ring[i & 3] = stream[((size_t)blockIdx.x * 8191 + (size_t)i * 257 + threadIdx.x) % stream_n];
        }
        // fresh (p, z) per iteration and lane, through LDS like pthr[] of the kernel (FORM 0) or straight from registers
        const float p = x0 * 0.5f + i * 1e-3f, z = x1 * 0.25f - 1.0f;
        f32x2 b = {p, z};
        if (FORM == 0) {
            pz[wid * 64 + lane] = make_float2(p, z);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const float2 t = pz[wid * 64 + lane];
            b = (f32x2){t.x, t.y};
        }
        f32x2 a = {x0, x1}, d;
        if (FORM == 2) {
            f32x2 zz = {b.y, b.y}, pp = {b.x, b.x};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(zz), "v"(pp));
        } else {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b));
        }
        float r0, r1;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a.x), "v"(b.y), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(a.y), "v"(b.y), "v"(b.x));
        nbad += (__float_as_uint(d.x) != __float_as_uint(r0)) + (__float_as_uint(d.y) != __float_as_uint(r1));
        x0 = fmaf(x0, 1.0003f, 0.01f);
        x1 = fmaf(x1, 0.9997f, -0.02f);
    }
    if (vm) sink[blockIdx.x * 512 + threadIdx.x] = keep + ring[0].y + ring[1].y + ring[2].y + ring[3].y;
    if (nbad) atomicAdd(bad, nbad);
}

template <int FORM>
static unsigned long long run(int blocks, int iters, int mode)
{
    unsigned long long *bad, h = 0;
    float *sink;
    static float4 *stream = nullptr;
    const int stream_n = 1 << 24;       // 256 MiB: the loads miss the L2
    if (!stream) { hipMalloc(&stream, (size_t)stream_n * 16); hipMemset(stream, 0, (size_t)stream_n * 16); }
    hipMalloc(&bad, 8);
    hipMalloc(&sink, (size_t)blocks * 512 * 4);
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(probe<FORM>, dim3(blocks), dim3(512), 0, 0, bad, sink, iters, mode, stream, stream_n);
    hipDeviceSynchronize();
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    hipFree(bad); hipFree(sink);
    return h;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200000, blocks = 256;
    const char *forms[3] = {"op_sel form, (p, z) via ds_read_b64", "op_sel form, (p, z) in registers", "plain form on {z,z} / {p,p}"};
    const char *modes[8] = {"partner idle", "partner runs MFMAs", "partner runs VALU", "", "idle + own VMEM in flight", "MFMAs + own VMEM in flight",
                            "VALU + own VMEM in flight", ""};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 7; ++mode) {
            if (mode == 3) continue;
            const unsigned long long e0 = run<0>(blocks, iters, mode), e1 = run<1>(blocks, iters, mode), e2 = run<2>(blocks, iters, mode);
            const double n = 2.0 * iters * 256.0 * blocks;
            printf("%-28s | %-38s wrong %llu of %.2e | %-34s wrong %llu | %-28s wrong %llu\n", modes[mode], forms[0], e0, n, forms[1],
                   e1, forms[2], e2);
        }
    return 0;
}
