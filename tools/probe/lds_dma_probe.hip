// LDS-DMA vs register staging: how many bytes per clock and CU does the global -> LDS path sustain on gfx950?
//   MODE 0: global_load_lds_dwordx4 (LDS-DMA, 16 B per lane, 1 KiB per wave instruction)
//   MODE 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   MODE 2: global_load_lds_dword (4 B per lane)
// 8 waves per block, one block per CU, each "stage" moves 56 KiB (56 wave instructions, 7 per wave) out of an L2-resident
// table with the split kernel's row geometry (128-byte row segments, row stride 832 B), then waits (vmcnt / lgkmcnt) and
// barriers -- the data-movement skeleton of lp_split_count_kernel's stage loop without its MFMAs.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/lds_dma_probe.hip -o tools/probe/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define STAGE_BYTES (56 * 1024)
template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(const char *tab, long row_bytes, int rows, int stages, float *out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int srow = tid >> 3, sch = tid & 7;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int s = 0; s < stages; ++s) {
        const int buf = s & 1;
        // 448 rows x 128 B per stage; the block walks its own window of the table
        const long r0 = ((long)blockIdx.x * 448 + (long)(s % 16) * 7168) % (rows - 448);
        const char *g = tab + (r0 + srow) * row_bytes + (s % 6) * 128 + sch * 16;
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const char *gj = g + (long)j * 64 * row_bytes;
                const unsigned l = lds0 + buf * STAGE_BYTES + wid * 1024 + j * 8192;
                if (MODE == 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gj,
                                                     (__attribute__((address_space(3))) void *)(smem + buf * STAGE_BYTES + wid * 1024 + j * 8192), 16, 0, 0);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)   // 4 dword instructions for the same 1 KiB
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (long)j * 64 * row_bytes - sch * 16 + (lane & 7) * 4 + q * 32),
                                                         (__attribute__((address_space(3))) void *)(smem + buf * STAGE_BYTES + wid * 1024 + j * 8192 + q * 256), 4, 0, 0);
                }
                (void)l;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = *reinterpret_cast<const uint4 *>(g + (long)j * 64 * row_bytes);
#pragma unroll
            for (int j = 0; j < 7; ++j)
                *reinterpret_cast<uint4 *>(smem + buf * STAGE_BYTES + j * 8192 + tid * 16) = v[j];
        }
        __syncthreads();
        // touch the stage so that nothing is optimised away (one b128 read per thread)
        const uint4 t = *reinterpret_cast<const uint4 *>(smem + buf * STAGE_BYTES + ((tid * 16 + s * 64) % STAGE_BYTES));
        acc.x ^= t.x; acc.y += t.y; acc.z ^= t.z; acc.w += t.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1.f;
}

int main()
{
    const long row_bytes = 832;
    const int rows = 256 * 1024;                      // 213 MB table: Infinity-Cache resident, mostly L2 misses at first touch
    char *tab; float *out;
    hipMalloc(&tab, rows * row_bytes + 4096); hipMalloc(&out, 4096 * sizeof(float));
    hipMemset(tab, 1, rows * row_bytes + 4096);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, stages = 4000;
    auto run = [&](int mode, int small) {
        const int r = small ? 16384 : rows;           // 13.6 MB window: L2 / MALL hits
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) hipEventRecord(e0);
            if (mode == 0) { hipFuncSetAttribute((const void *)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES); hipLaunchKernelGGL(probe<0>, dim3(cus), dim3(512), 2 * STAGE_BYTES, 0, tab, row_bytes, r, stages, out); }
            if (mode == 1) { hipFuncSetAttribute((const void *)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES); hipLaunchKernelGGL(probe<1>, dim3(cus), dim3(512), 2 * STAGE_BYTES, 0, tab, row_bytes, r, stages, out); }
            if (mode == 2) { hipFuncSetAttribute((const void *)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES); hipLaunchKernelGGL(probe<2>, dim3(cus), dim3(512), 2 * STAGE_BYTES, 0, tab, row_bytes, r, stages, out); }
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)cus * stages * STAGE_BYTES;
        printf("mode %d (%s) window %s: %.3f ms, %.2f TB/s aggregate, %.1f GB/s per CU, %.1f us per 56-KiB stage\n", mode,
               mode == 0 ? "LDS-DMA b128" : mode == 1 ? "load b128 -> VGPR -> ds_write_b128" : "LDS-DMA b32", small ? "13.6 MB" : "213 MB",
               ms, bytes / ms / 1e9, bytes / ms / 1e6 / cus, ms * 1e3 / stages);
    };
    for (int small = 1; small >= 0; --small)
        for (int mode = 0; mode < 3; ++mode) run(mode, small);
    return 0;
}
