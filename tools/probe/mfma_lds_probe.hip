// Do ds_read_b128 returns overlap MFMAs on the same SIMD?  8 waves per workgroup (2 per SIMD), one workgroup per CU.
//   mode 0: waves 0-3 issue NMF MFMAs (32x32x16 bf16, 8 independent accumulators), waves 4-7 idle
//   mode 1: waves 4-7 issue NRD ds_read_b128 (6 per s_waitcnt), waves 0-3 idle
//   mode 2: both at once (the ping-pong situation: one wave of a SIMD multiplies, the other reads)
//   mode 3: every wave interleaves 8 MFMAs with 6 reads of the NEXT fragments (the free-running situation), half the rounds each
// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_lds_probe.hip -o tools/probe/bin/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
//   mode 4: mode 3 + two LDS-DMA instructions (1 KiB each, L2-resident source) per round between the MFMAs, awaited every 4th round
//   mode 5: mode 4 + one s_barrier every 4th round (the k tile of conv_mma8's free-running schedule)
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int rounds, int mode, const char* gsrc) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    const int lr = lane & 31, hh = lane >> 5;
    const int base = lr * 128 + ((hh ^ ((lr >> 1) & 7)) << 4) + (wave & 3) * 16384;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 fa[2], fb[4];
    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(smem + base + i * 4096);
    for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const bf16x8*>(smem + base + 8192 + i * 2048);
    u32x4 racc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const bool do_mf = mode == 0 || (mode == 2 && wave < 4), do_rd = mode == 1 || (mode == 2 && wave >= 4);
    //   mode 6: mode 5 with the younger half (waves 4-7) at s_setprio 1 for the whole loop;  mode 7: mode 5 with s_setprio 1 around every MFMA cluster
    //   mode 8: mode 5 with the barrier every 8th round (two k tiles per barrier)
    if (mode >= 4) {
        if (mode == 6 && wave >= 4) __builtin_amdgcn_s_setprio(1);
        const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
        const char* src = gsrc + (size_t)blockIdx.x * 65536;
        for (int t = 0; t < rounds / 2; ++t) {
            u32x4 v[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) v[r] = *reinterpret_cast<const u32x4*>(smem + base + ((t & 3) << 5) + r * 2048);
            __builtin_amdgcn_sched_barrier(0);
            if (mode == 7) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i >> 2], fb[i & 3], acc[i], 0, 0, 0);
                if (i == 2 || i == 5) {
                    __builtin_amdgcn_sched_barrier(0);
                    glds16(src, (unsigned)(((t & 3) * 2 + (i == 5)) * 8192 + wave * 1024 + lane * 16), lds0 + 65536 + (unsigned)(((t & 3) * 2 + (i == 5)) * 8192 + wave * 1024));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (mode == 7) __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((t & 3) == 3) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((mode >= 5 && mode <= 7) || (mode == 8 && (t & 7) == 7)) __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) racc += v[r];
            asm volatile("" : "+v"(racc));
        }
    } else if (mode == 3) {
        for (int t = 0; t < rounds / 2; ++t) {
            u32x4 v[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) v[r] = *reinterpret_cast<const u32x4*>(smem + base + ((t & 3) << 5) + r * 2048);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i >> 2], fb[i & 3], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 6; ++r) racc += v[r];
            asm volatile("" : "+v"(racc));
        }
    } else if (do_mf && wave < 4 || (mode == 0 && wave < 4)) {
        for (int t = 0; t < rounds; ++t) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i >> 2], fb[i & 3], acc[i], 0, 0, 0);
        }
    } else if (do_rd && wave >= 4 || (mode == 1 && wave >= 4)) {
        for (int t = 0; t < rounds; ++t) {
            u32x4 v[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) v[r] = *reinterpret_cast<const u32x4*>(smem + base + ((t & 3) << 5) + r * 2048);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 6; ++r) racc += v[r];
            asm volatile("" : "+v"(racc));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (threadIdx.x == 0) out[4096 + blockIdx.x] = t2 - t0;
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
    if (s == 123.25f || racc[0] == 0x1234567u) sink[0] = s + racc[1];
}
//   k16: 16 waves per workgroup (4 per SIMD), 64 x 64 wave tiles: per round 4 reads + 4 MFMAs + 1 DMA; barrier every 4th round (prio: setprio around MFMAs)
__global__ __launch_bounds__(1024) void k16(unsigned long long* out, float* sink, int rounds, int prio, int bar, const char* gsrc) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    const int lr = lane & 31, hh = lane >> 5;
    const int base = lr * 128 + ((hh ^ ((lr >> 1) & 7)) << 4) + (wave & 3) * 16384;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 fa[2], fb[2];
    for (int i = 0; i < 2; ++i) { fa[i] = *reinterpret_cast<const bf16x8*>(smem + base + i * 4096); fb[i] = *reinterpret_cast<const bf16x8*>(smem + base + 8192 + i * 2048); }
    u32x4 racc = {0, 0, 0, 0};
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const char* src = gsrc + (size_t)blockIdx.x * 65536;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < rounds; ++t) {
        u32x4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const u32x4*>(smem + base + ((t & 3) << 5) + r * 2048);
        __builtin_amdgcn_sched_barrier(0);
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i >> 1], fb[i & 1], acc[i], 0, 0, 0);
            if (i == 1) {
                __builtin_amdgcn_sched_barrier(0);
                glds16(src, (unsigned)((t & 3) * 16384 + wave * 1024 + lane * 16), lds0 + 65536 + (unsigned)((t & 3) * 16384 + wave * 1024));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (prio) __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if ((t & 3) == 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (bar) __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) racc += v[r];
        asm volatile("" : "+v"(racc));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = t2 - t0; out[4096 + blockIdx.x] = t1 - t0; }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][lane & 15];
    if (s == 123.25f || racc[0] == 0x1234567u) sink[0] = s + racc[1];
}
//   k4: FOUR waves per workgroup (one per SIMD), TWO workgroups per CU (64 KiB of LDS each): per round 8 MFMAs + 6 reads + NDMA DMA instructions,
//   s_barrier (4 waves) every `bar_every` rounds -- the two waves of a SIMD belong to different workgroups: one's barrier stall is the other's MFMA time
template <int NDMA>
__global__ __launch_bounds__(256) void k4(unsigned long long* out, float* sink, int rounds, int prio, int bar_every, const char* gsrc) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    const int lr = lane & 31, hh = lane >> 5;
    const int base = lr * 128 + ((hh ^ ((lr >> 1) & 7)) << 4) + (wave & 1) * 16384;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 fa[2], fb[4];
    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(smem + base + i * 4096);
    for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const bf16x8*>(smem + base + 8192 + i * 2048);
    u32x4 racc = {0, 0, 0, 0};
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const char* src = gsrc + (size_t)(blockIdx.x & 255) * 65536;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < rounds; ++t) {
        u32x4 v[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) v[r] = *reinterpret_cast<const u32x4*>(smem + base + ((t & 3) << 5) + (r & 3) * 2048 + (r >> 2) * 4096);
        __builtin_amdgcn_sched_barrier(0);
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i >> 2], fb[i & 3], acc[i], 0, 0, 0);
            if ((NDMA >= 1 && i == 1) || (NDMA >= 2 && i == 4) || (NDMA >= 3 && i == 6)) {
                __builtin_amdgcn_sched_barrier(0);
                glds16(src, (unsigned)((t & 3) * 12288 + i * 4096 / 2 + wave * 1024 + lane * 16) & 65535u, lds0 + 32768 + (unsigned)(((t & 1) * 3 + (i > 1) + (i > 4)) * 4096 + wave * 1024));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (prio) __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (bar_every && (t % bar_every) == bar_every - 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) racc += v[r];
        asm volatile("" : "+v"(racc));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
    if (s == 123.25f || racc[0] == 0x1234567u) sink[0] = s + racc[1];
}
int main() {
    unsigned long long* d; float* sink; char* g; hipMalloc(&d, 8 * 8192); hipMalloc(&sink, 64); hipMalloc(&g, 256 * 65536 + 4096); hipMemset(g, 0, 256 * 65536 + 4096);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int rounds = 2000;
    for (int mode = 5; mode < 8; mode += 2) {
        for (int rep = 0; rep < 2; ++rep) { k<<<256, 512, 131072>>>(d, sink, rounds, mode, g); hipDeviceSynchronize(); }
        static unsigned long long h[8192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double mf = 0, rd = 0, tot = 0;
        for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) { mf += h[b * 8 + w]; rd += h[b * 8 + 4 + w]; } tot += h[4096 + b]; }
        mf /= 1024; rd /= 1024; tot /= 256;
        printf("mode %d: waves 0-3 %8.0f clk (%6.1f / round), waves 4-7 %8.0f clk (%6.1f / round), workgroup %8.0f clk (%6.1f / round)\n", mode, mf, mf / rounds, rd, rd / rounds, tot, tot / rounds);
    }
    hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int prio = 0; prio < 2; ++prio) for (int bar = 0; bar < 2; ++bar) {
        for (int rep = 0; rep < 2; ++rep) { k16<<<256, 1024, 131072>>>(d, sink, rounds / 2, prio, bar, g); hipDeviceSynchronize(); }
        static unsigned long long h[8192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double tot = 0; for (int b = 0; b < 256; ++b) tot += h[b]; tot /= 256;
        printf("k16 (16 waves, 4 MFMA + 4 reads + 1 DMA per round) prio %d barrier %d: workgroup %8.0f clk = %6.1f per 16 MFMAs per SIMD (ideal 512)\n", prio, bar, tot, tot / (rounds / 2));
    }
    hipFuncSetAttribute((const void*)k4<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k4<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int ndma = 2; ndma <= 3; ++ndma) for (int prio = 0; prio < 2; ++prio) for (int bar : {0, 2, 4}) {
        for (int rep = 0; rep < 2; ++rep) {
            if (ndma == 2) k4<2><<<512, 256, 65536>>>(d, sink, rounds / 2, prio, bar, g); else k4<3><<<512, 256, 65536>>>(d, sink, rounds / 2, prio, bar, g);
            hipDeviceSynchronize();
        }
        static unsigned long long h[8192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double tot = 0; for (int b = 0; b < 512; ++b) tot += h[b]; tot /= 512;
        printf("k4 (2 workgroups of 4 waves per CU, 8 MFMA + 6 reads + %d DMA per round) prio %d barrier every %d: %8.0f clk = %6.1f per 16 MFMAs per SIMD (ideal 512)\n", ndma, prio, bar, tot, tot / (rounds / 2));
    }
    printf("(8 MFMAs of 32 clk = 256 clk per round; 6 reads alone on a SIMD ~176 clk per round)\n");
    return 0;
}
