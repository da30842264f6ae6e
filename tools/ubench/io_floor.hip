// micro-benchmark: what does a launch that only MOVES the metric workload's bytes cost on this GPU, and how do
// instruction issue and the store drain add up when there is exactly one wave per SIMD (batch 65 536)?
//   empty     : 256 blocks x 256 threads, no work                      -> launch + completion floor
//   io        : per wave, read the q tile (64 x 7 floats) and write pos / quat / lin_jac / ang_jac tiles
//               (64 x 49 floats) with the same 16-byte coalesced accesses as fk_jacobian_full_tiles_kernel
//   rd / wr   : only the loads / only the stores of `io`
//   spin<N,S> : loads, then N dependent VALU FMAs, the 14 stores issued in S equal instalments spread over
//               the FMAs (S = 1: all at the end, like the real kernel)
// K launches are captured into one hipGraph and replayed (what bench.py does); HIP events around the replay.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define ARGS const float *__restrict__ q, int n_tiles, float *__restrict__ pos, float *__restrict__ quat, \
             float *__restrict__ lin, float *__restrict__ ang

__global__ void __launch_bounds__(256) k_empty(ARGS) {}

template <bool NT>
__device__ __forceinline__ void st4(float4 *p, float4 v) {
    if (NT) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<v4 *>(p));
    } else {
        *p = v;
    }
}

// one 16-byte store per lane and nothing else: is the cost of writing a fixed end-of-kernel cost or a drain?
__global__ void __launch_bounds__(256) k_wr_small(ARGS) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    reinterpret_cast<float4 *>(quat + (size_t)tile * 256)[threadIdx.x & 63u] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <bool RD, bool WR, int N, int S, bool NT = false>
__global__ void __launch_bounds__(256) k_model(ARGS) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x & 63u;
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    if (RD) {
        const float4 *q4 = reinterpret_cast<const float4 *>(q + (size_t)tile * 448);
        float4 a = q4[lane], b = q4[lane < 48 ? 64 + lane : 111];
        v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    float4 *p4 = reinterpret_cast<float4 *>(pos + (size_t)tile * 192);
    float4 *r4 = reinterpret_cast<float4 *>(quat + (size_t)tile * 256);
    float4 *l4 = reinterpret_cast<float4 *>(lin + (size_t)tile * 1344);
    float4 *a4 = reinterpret_cast<float4 *>(ang + (size_t)tile * 1344);
    auto store = [&](int i) { // 14 store instructions, i = 0..13
        if (!WR) { if (v.x == 12345.678f) r4[lane] = v; return; }
        if (i == 0) { if (lane < 48) st4<NT>(p4 + lane, v); }
        else if (i == 1) st4<NT>(r4 + lane, v);
        else if (i < 7) st4<NT>(l4 + lane + 64 * (i - 2), v);
        else if (i == 7) { if (lane < 16) st4<NT>(l4 + lane + 320, v); }
        else if (i < 13) st4<NT>(a4 + lane + 64 * (i - 8), v);
        else { if (lane < 16) st4<NT>(a4 + lane + 320, v); }
    };
    int next = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll 1
        for (int i = 0; i < N / S / 4; ++i) { // 4 dependent FMAs per iteration + loop overhead (2 SALU)
            v.x = fmaf(v.x, 1.0001f, v.y); v.y = fmaf(v.y, 0.9999f, v.z); v.z = fmaf(v.z, 1.0001f, v.w); v.w = fmaf(v.w, 0.9999f, v.x);
        }
        const int upto = 14 * (s + 1) / S;
#pragma unroll
        for (int i = 0; i < 14; ++i) if (i >= next && i < upto) store(i);
        next = upto;
    }
}

// issue-rate probes: 4 independent FMA chains, N instructions in total, either as one straight-line block
// (every instruction fetched once, like the real kernels) or as a 16-instruction loop body (fits any fetch buffer)
template <int N, bool LOOP>
__global__ void __launch_bounds__(256) k_issue(ARGS) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    float a = threadIdx.x * 1e-3f, b = 1.5f, c = 0.5f, d = 0.25f;
    if (LOOP) {
#pragma unroll 1
        for (int i = 0; i < N / 16; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { a = fmaf(a, 1.0001f, 0.5f); b = fmaf(b, 0.9999f, 0.25f); c = fmaf(c, 1.0002f, 0.125f); d = fmaf(d, 0.9998f, 0.0625f); }
        }
    } else {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { a = fmaf(a, 1.0001f, 0.5f); b = fmaf(b, 0.9999f, 0.25f); c = fmaf(c, 1.0002f, 0.125f); d = fmaf(d, 0.9998f, 0.0625f); }
    }
    if (a + b + c + d == 12345.678f) quat[threadIdx.x] = a;
}

template <class K>
static float run(K kernel, int B, int launches, const float *q, float *pos, float *quat, float *lin, float *ang) {
    hipStream_t s; (void)hipStreamCreate(&s);
    const int n_tiles = B / 64;
    dim3 grid((n_tiles + 3) / 4), block(256);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, s, q, n_tiles, pos, quat, lin, ang);
    (void)hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, s, q, n_tiles, pos, quat, lin, ang);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0, s); (void)hipGraphLaunch(ge, s); (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f / launches; // us per launch
}

int main(int argc, char **argv) {
    const int launches = 200;
    for (int B : {65536, 1 << 20}) {
        float *q, *pos, *quat, *lin, *ang;
        (void)hipMalloc(&q, (size_t)B * 7 * 4); (void)hipMalloc(&pos, (size_t)B * 3 * 4); (void)hipMalloc(&quat, (size_t)B * 4 * 4);
        (void)hipMalloc(&lin, (size_t)B * 21 * 4); (void)hipMalloc(&ang, (size_t)B * 21 * 4);
        (void)hipMemset(q, 0, (size_t)B * 7 * 4);
#define RUN(name, K) printf("%s B=%d %-28s %6.2f us/launch\n", argv[0], B, name, run(K, B, launches, q, pos, quat, lin, ang));
        RUN("empty", k_empty)
        RUN("io (rd+wr)", (k_model<true, true, 0, 1>))
        RUN("rd only", (k_model<true, false, 0, 1>))
        RUN("wr only", (k_model<false, true, 0, 1>))
        RUN("wr only, nontemporal", (k_model<false, true, 0, 1, true>))
        RUN("wr 16 B per lane only", k_wr_small)
        RUN("rd + 800 fma + wr at end, nt", (k_model<true, true, 800, 1, true>))
        RUN("rd + 800 fma + wr in 7, nt", (k_model<true, true, 800, 7, true>))
        RUN("800 indep fma, straight-line", (k_issue<800, false>))
        RUN("800 indep fma, loop of 16", (k_issue<800, true>))
        RUN("1600 indep fma, straight-line", (k_issue<1600, false>))
        RUN("1600 indep fma, loop of 16", (k_issue<1600, true>))
        RUN("rd + 400 fma, no wr", (k_model<true, false, 400, 1>))
        RUN("rd + 800 fma, no wr", (k_model<true, false, 800, 1>))
        RUN("rd + 1200 fma, no wr", (k_model<true, false, 1200, 1>))
        RUN("rd + 400 fma + wr at end", (k_model<true, true, 400, 1>))
        RUN("rd + 800 fma + wr at end", (k_model<true, true, 800, 1>))
        RUN("rd + 1200 fma + wr at end", (k_model<true, true, 1200, 1>))
        RUN("rd + 800 fma + wr in 2", (k_model<true, true, 800, 2>))
        RUN("rd + 800 fma + wr in 7", (k_model<true, true, 800, 7>))
        RUN("rd + 1200 fma + wr in 7", (k_model<true, true, 1200, 7>))
        (void)hipFree(q); (void)hipFree(pos); (void)hipFree(quat); (void)hipFree(lin); (void)hipFree(ang);
    }
    return 0;
}
