// probe: semantics of __builtin_amdgcn_global_load_lds (16 B) on gfx950: lane-linear LDS placement,
// per-lane source, zero-page redirection, counted vmcnt across a raw barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ src, const unsigned* __restrict__ zero, unsigned* out, int n) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 4096];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // 4 "tiles", each 4 KB: wave w writes 1 KB at tile*4096 + w*1024; lane reads chunk (lane ^ 5) reversed to prove per-lane source
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        const int idx = tile * 256 + wave * 64 + (lane ^ 5);        // source chunk index (16 B units)
        const unsigned* g = (idx % 7 == 3) ? zero : src + (size_t)idx * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(smem + tile * 4096 + wave * 1024), 16, 0, 0);
    }
    // wait for tile 0 only (3 newer loads may stay in flight), then barrier, read tile 0
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    u32x4 v0 = *reinterpret_cast<const u32x4*>(smem + t * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    u32x4 acc = v0;
#pragma unroll
    for (int tile = 1; tile < 4; ++tile) {
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + tile * 4096 + t * 16);
        acc += v;
    }
    *reinterpret_cast<u32x4*>(out + (size_t)t * 4) = acc;
    *reinterpret_cast<u32x4*>(out + 1024 + (size_t)t * 4) = v0;
}

int main() {
    const int n = 4 * 256 * 4;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = i * 3 + 1;
    unsigned *d, *z, *o;
    hipMalloc(&d, n * 4); hipMalloc(&z, 64); hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(z, 0, 64);
    probe<<<1, 256>>>(d, z, o, n);
    std::vector<unsigned> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const int wave = t >> 6, lane = t & 63;
        for (int e = 0; e < 4; ++e) {
            unsigned want = 0, want0 = 0;
            for (int tile = 0; tile < 4; ++tile) {
                const int idx = tile * 256 + wave * 64 + (lane ^ 5);
                const unsigned v = (idx % 7 == 3) ? 0u : h[(size_t)idx * 4 + e];
                want += v;
                if (tile == 0) want0 = v;
            }
            if (r[t * 4 + e] != want || r[1024 + t * 4 + e] != want0) { if (bad < 5) printf("mismatch t=%d e=%d got %u/%u want %u/%u\n", t, e, r[t*4+e], r[1024+t*4+e], want, want0); ++bad; }
        }
    }
    printf("glds probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
