// Probe the semantics of ds_read_b64_tr_b8 on gfx950: LDS holds a [32 rows][16 cols] byte matrix M[r][c] = (r << 4) | c (row pitch 16 B).
// Hypothesis (the b16 form's rule with 8 rows): in each 16-lane group, lane i supplies the address of 8 contiguous bytes
// M[row0 + i/2][(i%2)*8 ..] and receives M[row0 + j][i] for j = 0..7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(uint8_t* out, int variant) {
    __shared__ __attribute__((aligned(16))) uint8_t M[64 * 16];
    for (int i = threadIdx.x; i < 64 * 16; i += 64) M[i] = (uint8_t)((((i / 16) & 15) << 4) | (i % 16));
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    int row, col;
    if (variant == 0) { row = g * 8 + i / 2; col = (i % 2) * 8; }          // hypothesis
    else if (variant == 1) { row = g * 8 + (i % 8); col = (i / 8) * 8; }   // alternative lane->piece mapping
    else { row = ((3 - g) * 8 + i / 2) & 15; col = (i % 2) * 8; }          // other row blocks per group
    const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) uint8_t*)M) + (row * 16 + col);
    u32x2 v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) { out[l * 8 + j] = (v[0] >> (8 * j)) & 0xff; out[l * 8 + 4 + j] = (v[1] >> (8 * j)) & 0xff; }
}
int main() {
    uint8_t* d; hipMalloc(&d, 64 * 8);
    uint8_t h[512];
    for (int variant = 0; variant < 3; ++variant) {
        probe<<<1, 64>>>(d, variant); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("variant %d\n", variant);
        for (int l = 0; l < 64; l += (l % 16 == 3 ? 13 : 1)) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 8; ++j) printf(" (r%2d,c%2d)", h[l * 8 + j] >> 4, h[l * 8 + j] & 15);
            printf("\n");
        }
    }
    return 0;
}
