// Micro-benchmark: how fast can one CU pull L2-resident bytes into LDS?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: global_load_dwordx4 -> VGPR only (no LDS write)
//   mode 3: LDS-DMA with a workgroup barrier per iteration (the GEMM ring's synchronisation, no MFMAs)
// One workgroup per CU (LDS-limited), NW waves; every wave streams `iters` x PIECES KiB from a buffer of
// `span` bytes that all workgroups share (L2-resident when small) or that is private per workgroup (HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PIECES>
__global__ __launch_bounds__(512) void k(const char* src, unsigned span, size_t wg_stride, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * wg_stride;
    char* lds = smem + wave * PIECES * 1024 * 2;  // two slots per wave
    u32x4 acc = {0, 0, 0, 0};
    unsigned off = wave * PIECES * 1024;  // 32-bit offsets and power-of-two spans: no 64-bit modulo on the streaming path
    for (int it = 0; it < iters; ++it) {
        char* dst = lds + (it & 1) * PIECES * 1024;
        u32x4 r[PIECES];
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const char* g = base + ((off + p * 1024) & (span - 1)) + lane * 16;
            if (MODE == 0 || MODE == 3) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
            } else {
                r[p] = *(const u32x4*)g;
            }
        }
        if (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");  // previous iteration's pieces landed
        } else if (MODE == 3) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 1) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) *(u32x4*)(dst + p * 1024 + lane * 16) = r[p];
        } else {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) acc ^= r[p];
        }
        off += nw * PIECES * 1024;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    acc ^= *(u32x4*)(smem + threadIdx.x * 16);
    if (acc[0] == 0x12345678u) sink[0] = acc[1] + acc[2] + acc[3];
}

template <int MODE, int PIECES>
static void run(const char* name, int nw, unsigned span, bool priv, const char* buf, unsigned* sink) {
    const int iters = 2000, grid = 256;
    const size_t lds = 100 * 1024;
    hipFuncSetAttribute((const void*)k<MODE, PIECES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t stride = priv ? span : 0;
    k<MODE, PIECES><<<grid, nw * 64, lds>>>(buf, span, stride, 50, sink);
    hipEventRecord(e0);
    k<MODE, PIECES><<<grid, nw * 64, lds>>>(buf, span, stride, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * nw * iters * PIECES * 1024;
    printf("%-28s waves %d pieces %d span %6u KiB %s: %7.2f TB/s  (%.1f B/clk/CU at 2.0 GHz)\n", name, nw, PIECES, span >> 10,
           priv ? "private" : "shared ", bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.0e9 / 1e0 / 1e0 * 1e-0 / 1.0 / 1.0 * 1.0 / 1e0);
}

int main() {
    char* buf; unsigned* sink;
    const size_t total = (size_t)256 * 8 << 20;  // 2 GiB: 8 MiB private per workgroup
    hipMalloc(&buf, total); hipMemset(buf, 1, total); hipMalloc(&sink, 64);
    for (int nw : {4, 8}) {
        run<0, 6>("lds-dma", nw, 1 << 20, false, buf, sink);
        run<1, 6>("vgpr + ds_write", nw, 1 << 20, false, buf, sink);
        run<2, 6>("vgpr only", nw, 1 << 20, false, buf, sink);
        run<0, 6>("lds-dma", nw, 8 << 20, true, buf, sink);
        run<2, 6>("vgpr only", nw, 8 << 20, true, buf, sink);
    }
    run<3, 5>("lds-dma + barrier", 8, 1 << 20, false, buf, sink);
    run<3, 5>("lds-dma + barrier", 8, 4 << 20, false, buf, sink);
    run<0, 5>("lds-dma", 8, 4 << 20, false, buf, sink);
    run<3, 6>("lds-dma + barrier", 4, 1 << 20, false, buf, sink);
    run<0, 2>("lds-dma", 4, 1 << 20, false, buf, sink);
    run<0, 12>("lds-dma", 4, 1 << 20, false, buf, sink);
    run<2, 12>("vgpr only", 4, 1 << 20, false, buf, sink);
    return 0;
}
