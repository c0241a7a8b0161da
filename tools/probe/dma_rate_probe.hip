// probe: what bounds the operand stream of a short implicit-GEMM workgroup on gfx950?
// Per-CU delivered bandwidth of (a) LDS-DMA (global_load_lds_dwordx4) and (b) global_load_dwordx4 -> VGPR -> ds_write_b128,
// for an L2-resident shared region (filter-like: every workgroup reads the same bytes) and a private streaming region
// (activation-like), at 1 / 2 / 4 workgroups per CU and several in-flight depths.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// MODE 0: LDS-DMA; MODE 1: VGPR + ds_write.  Each wave moves `iters` x DEPTH KiB; region of `region_bytes` per workgroup
// starting at base + wg_stride * blockIdx.x (wg_stride = 0: shared region).
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char* base, size_t wg_stride, unsigned region_bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const char* src = base + wg_stride * blockIdx.x;
    asm volatile("" : "+s"(src));
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    unsigned off = (unsigned)(wave * 1024 + lane * 16);        // wave w covers KiB w, w+4, ... of the region
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                glds16_s(src, off, lds0 + (unsigned)((wave * DEPTH + d) * 1024));
                off += 4096; if (off >= region_bytes) off -= region_bytes;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                v[d] = *reinterpret_cast<const u32x4*>(src + off);
                off += 4096; if (off >= region_bytes) off -= region_bytes;
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) *reinterpret_cast<u32x4*>(smem + (wave * DEPTH + d) * 1024 + lane * 16) = v[d];
        }
    }
    __syncthreads();
    acc = *reinterpret_cast<unsigned*>(smem + t * 4);
    if (acc == 0x12345) sink[0] = acc;
}

// Segmented gather (what an implicit-GEMM operand tile looks like): one DMA instruction = (1024 / SEG) rows x SEG contiguous bytes,
// rows `pitch` bytes apart; successive instructions walk along the rows (k direction), wrapping inside the region.
template <int SEG, int DEPTH>
__global__ __launch_bounds__(256) void seg_kernel(const char* base, size_t wg_stride, unsigned region_bytes, unsigned pitch, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const char* src = base + wg_stride * blockIdx.x;
    asm volatile("" : "+s"(src));
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    constexpr int LPR = SEG / 16;                       // lanes per row
    const unsigned row = (unsigned)(wave * (64 / LPR) + lane / LPR);      // 4 waves cover 4 * 1024 / SEG rows
    unsigned koff = (unsigned)(lane % LPR) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            unsigned off = row * pitch + koff;
            if (off >= region_bytes) off %= region_bytes;
            glds16_s(src, off, lds0 + (unsigned)((wave * DEPTH + d) * 1024));
            koff += SEG; if (koff >= pitch) koff -= pitch;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (*reinterpret_cast<unsigned*>(smem + t * 4) == 0x12345) sink[0] = 1;
}

template <int SEG, int DEPTH>
float run_seg(const char* buf, size_t wg_stride, unsigned region, unsigned pitch, int nwg, int iters) {
    unsigned* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 4 * DEPTH * 1024;
    hipFuncSetAttribute((const void*)seg_kernel<SEG, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    seg_kernel<SEG, DEPTH><<<nwg, 256, lds>>>(buf, wg_stride, region, pitch, iters, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    seg_kernel<SEG, DEPTH><<<nwg, 256, lds>>>(buf, wg_stride, region, pitch, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(sink);
    return ms;
}

template <int MODE, int DEPTH>
float run(const char* buf, size_t wg_stride, unsigned region, int nwg, int iters) {
    unsigned* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 4 * DEPTH * 1024;
    hipFuncSetAttribute((const void*)stream_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    stream_kernel<MODE, DEPTH><<<nwg, 256, lds>>>(buf, wg_stride, region, iters, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    stream_kernel<MODE, DEPTH><<<nwg, 256, lds>>>(buf, wg_stride, region, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(sink);
    return ms;
}

int main() {
    const size_t total = (size_t)1 << 30;
    char* buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
    printf("%-8s %-10s %6s %5s %9s %10s %12s\n", "path", "region", "wgs", "depth", "us", "TB/s chip", "GB/s per CU");
    struct Case { const char* name; size_t stride; unsigned region; };
    const Case cases[] = {{"shared128K", 0, 128 << 10}, {"shared1M", 0, 1 << 20}, {"shared16M", 0, 16 << 20}, {"private1M", (size_t)1 << 20, 1 << 20}};
    for (const Case& c : cases)
        for (int nwg : {256, 512, 1024}) {
            if (c.stride && (size_t)nwg * c.stride > total) continue;
            const int iters = 64;
#define ROW(MODE, DEPTH, label) { float ms = run<MODE, DEPTH>(buf, c.stride, c.region, nwg, iters); double bytes = (double)nwg * 4 * DEPTH * 1024.0 * iters; \
            printf("%-8s %-10s %6d %5d %9.1f %10.2f %12.1f\n", label, c.name, nwg, DEPTH, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / 256.0); }
            ROW(0, 2, "ldsdma") ROW(0, 4, "ldsdma") ROW(0, 8, "ldsdma") ROW(0, 16, "ldsdma")
            ROW(1, 2, "vgpr") ROW(1, 4, "vgpr") ROW(1, 8, "vgpr")
        }
    printf("\nsegmented gather from a shared, L2-resident 1 MB region (rows 2 KB apart) and from private 1 MB regions\n");
    printf("%-8s %-10s %6s %5s %9s %10s %12s\n", "seg", "region", "wgs", "depth", "us", "TB/s chip", "GB/s per CU");
    for (int priv = 0; priv < 2; ++priv)
        for (int nwg : {256, 512}) {
            const size_t stride = priv ? ((size_t)1 << 20) : 0;
            const int iters = 64;
#define SROW(SEG, DEPTH) { float ms = run_seg<SEG, DEPTH>(buf, stride, 1 << 20, 2048, nwg, iters); double bytes = (double)nwg * 4 * DEPTH * 1024.0 * iters; \
            printf("%-8d %-10s %6d %5d %9.1f %10.2f %12.1f\n", SEG, priv ? "private1M" : "shared1M", nwg, DEPTH, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / 256.0); }
            SROW(64, 4) SROW(64, 8) SROW(128, 4) SROW(128, 8) SROW(256, 4) SROW(256, 8) SROW(1024, 8)
        }
    return 0;
}
