// probe: multi-stream (fork/join) stream capture into a hipGraph on ROCm 7.2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void add(float* p, float v, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += v; }
int main() {
    float *a, *b; const int n = 1 << 20;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork, e1, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
    add<<<n / 256, 256, 0, s0>>>(a, 1.f, n);
    add<<<n / 256, 256, 0, s1>>>(b, 2.f, n);
    CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s0, e1, 0));
    add<<<n / 256, 256, 0, s0>>>(a, 3.f, n);
    add<<<n / 256, 256, 0, s1>>>(b, 4.f, n);
    CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
    hipGraph_t g; CK(hipStreamEndCapture(s0, &g));
    hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ex, s0));
    CK(hipStreamSynchronize(s0));
    float ha, hb; CK(hipMemcpy(&ha, a, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, b, 4, hipMemcpyDeviceToHost));
    printf("capture probe: a=%g (want 12) b=%g (want 18)\n", ha, hb);
    return 0;
}
