// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS library's access patterns: three kernels
// that move a known number of bytes exactly once from / to a 2 GiB buffer (nothing can come from L2 or the Infinity Cache):
//   calib_dma    every workgroup streams its private 8 MiB by LDS-DMA (buffer_load ... lds, 16 B per lane: the GEMM operand path)
//   calib_vgpr   the same bytes by buffer_load_b128 into registers (the tail's weight / the epilogues' row path)
//   calib_store  every workgroup writes its private 8 MiB with 16-byte stores (the epilogues' row stores)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); tools/fetch_calib.py divides the
// counters by the known byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static constexpr size_t WG_BYTES = 8u << 20;

__global__ __launch_bounds__(256) void calib_dma(const char* src, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* base = src + (size_t)blockIdx.x * WG_BYTES;
    for (size_t off = (size_t)wave * 1024; off < WG_BYTES; off += 4 * 1024) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + ((off >> 10) & 31) * 1024), 16, 0, 0);
        if (((off >> 12) & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (*(unsigned*)(smem + threadIdx.x * 4) == 0x12345678u) sink[0] = 1;
}
__global__ __launch_bounds__(256) void calib_vgpr(const char* src, unsigned* sink) {
    const char* base = src + (size_t)blockIdx.x * WG_BYTES;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t off = (size_t)threadIdx.x * 16; off < WG_BYTES; off += 256 * 16) acc ^= *(const u32x4*)(base + off);
    if (acc[0] == 0x12345678u && acc[1] == 1) sink[0] = acc[2] + acc[3];
}
__global__ __launch_bounds__(256) void calib_store(char* dst) {
    char* base = dst + (size_t)blockIdx.x * WG_BYTES;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t off = (size_t)threadIdx.x * 16; off < WG_BYTES; off += 256 * 16) *(u32x4*)(base + off) = v;
}

int main() {
    char* buf; unsigned* sink;
    const int grid = 256;
    (void)hipMalloc(&buf, grid * WG_BYTES); (void)hipMemset(buf, 1, grid * WG_BYTES); (void)hipMalloc(&sink, 64);
    (void)hipFuncSetAttribute((const void*)calib_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        calib_dma<<<grid, 256, 32 * 1024>>>(buf, sink);
        calib_vgpr<<<grid, 256>>>(buf, sink);
        calib_store<<<grid, 256>>>(buf);
    }
    (void)hipDeviceSynchronize();
    printf("bytes per launch: %zu\n", (size_t)grid * WG_BYTES);
    return 0;
}
