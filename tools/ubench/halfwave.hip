// micro-benchmark: does a wave64 VALU instruction with only 32 active lanes issue faster on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void chain(float *out, int active, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    if ((int)(threadIdx.x & 63) < active) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { a = fmaf(a, b, c); d = fmaf(d, b, a); c = fmaf(c, b, d); b = fmaf(b, 0.999f, 1e-7f); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main() {
    float *out; hipMalloc(&out, 1 << 24);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int wpb : {1, 2}) for (int active : {64, 32, 16}) {
        // 1024 blocks (4 per CU) of wpb waves: 1 or 2 waves per SIMD
        int blocks = 1024;
        chain<<<blocks, 64 * wpb, 0, 0>>>(out, active, 10);
        hipDeviceSynchronize();
        hipEventRecord(s); chain<<<blocks, 64 * wpb, 0, 0>>>(out, active, 2000); hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        printf("waves/SIMD=%d active_lanes=%d  %.1f us  (%.2f cycles@2.4GHz per dependent-ish FMA)\n", wpb, active, ms * 1e3, ms * 1e-3 * 2.4e9 / (2000.0 * 64));
    }
    return 0;
}
