// Probe the semantics of ds_read_b64_tr_b16 on gfx950: LDS holds a [64 rows][16 cols] u16 matrix M[r][c] = r*256 + c
// (row pitch 32 B).  Hypothesis: in each 16-lane group, lane i supplies the address of 4 contiguous elements
// M[row0 + i/4][(i%4)*4 ..], and receives M[row0 + j][i] for j = 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(uint16_t* out, int variant) {
    __shared__ __attribute__((aligned(16))) uint16_t M[64 * 16];
    for (int i = threadIdx.x; i < 64 * 16; i += 64) M[i] = (uint16_t)((i / 16) * 256 + (i % 16));
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    int row, col;
    if (variant == 0) { row = g * 4 + i / 4; col = (i % 4) * 4; }          // hypothesis
    else if (variant == 1) { row = g * 4 + (i % 4); col = (i / 4) * 4; }   // alternative lane->piece mapping
    else { row = (3 - g) * 8 + i / 4; col = (i % 4) * 4; }                 // arbitrary per-group row blocks
    const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) uint16_t*)M) + (row * 16 + col) * 2;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int variant = 0; variant < 3; ++variant) {
        probe<<<1, 64>>>(d, variant); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("variant %d\n", variant);
        for (int l = 0; l < 64; l += (l % 16 == 3 ? 13 : 1)) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (r%2d,c%2d)", h[l * 4 + j] >> 8, h[l * 4 + j] & 255);
            printf("\n");
        }
    }
    return 0;
}
