// LDS read bandwidth of ds_read_b128 in the access pattern of conv_mma8.hip (lane -> row lane%32 of a 128-byte-row image, swizzled chunk):
// W waves per workgroup (one workgroup per CU), each issuing R reads per s_waitcnt lgkmcnt(0), T rounds.  Prints bytes/clock/CU.
// hipcc --offload-arch=gfx950 -O3 tools/probe/lds_read_probe.hip -o gpurun_out/lds_probe && gpurun_out/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int R>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int rounds, int waves_reading) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const int lr = lane & 31, hh = lane >> 5;
    const int base = lr * 128 + ((hh ^ ((lr >> 1) & 7)) << 4) + (wave & 7) * 4096;
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < waves_reading) {
        for (int t = 0; t < rounds; ++t) {
            u32x4 v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = *reinterpret_cast<const u32x4*>(smem + ((base + r * 32768 / R * 0 + ((r & 3) << 5) + (r >> 2) * 4096 * 8) & 131071));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < R; ++r) acc += v[r];
            asm volatile("" : "+v"(acc));
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc[0] == 0x12345678) out[1000] = acc[1];
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8 * 2048);
    const int rounds = 2000;
    for (int waves : {4, 8, 16}) for (int reading : {waves / 2, waves}) {
        for (int R : {6, 12, 16}) {
            hipFuncSetAttribute((const void*)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute((const void*)k<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute((const void*)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            for (int rep = 0; rep < 2; ++rep) {
                if (R == 6) k<6><<<256, waves * 64, 131072>>>(d, rounds, reading);
                else if (R == 12) k<12><<<256, waves * 64, 131072>>>(d, rounds, reading);
                else k<16><<<256, waves * 64, 131072>>>(d, rounds, reading);
                hipDeviceSynchronize();
            }
            unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double clk = 0; for (int i = 0; i < 256; ++i) clk += h[i]; clk /= 256;
            const double bytes = (double)reading * R * 1024.0 * rounds;
            printf("waves/WG %2d reading %2d reads/wait %2d: %8.0f clk, %6.1f B/clk/CU (%5.1f clk per wave-round)\n", waves, reading, R, clk, bytes / clk, clk / rounds);
        }
    }
    return 0;
}
