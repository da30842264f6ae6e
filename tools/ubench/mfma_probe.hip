// mfma_probe.hip — does the matrix pipe pay for this path's products?  (DESIGN.md §4/§6 "MFMA")
//
// The path's contractions are per-sample 3x3 (x) 3x3 / 3x3 (x) 3-vector products in a serial chain (pose composition, the
// CRBA / RNEA transforms): every sample has its own operands on at least one side.  gfx950's f32-input MFMAs run at the
// vector FP32 rate (64 flop/clk/SIMD, MI355X_MICROARCH.md "Matrix cores"), so they can only win by *not wasting* slots.
// Three ways of doing the same chained product  R <- R J_k  (J_k per sample, k = 1..STEPS) for 64 samples per wave:
//
//   pk        lane per sample, rows as packed pairs: 9 v_pk_fma_f32 + 9 v_fma_f32 per step       (what the kernels do)
//   mfma4     v_mfma_f32_4x4x1_16b_f32: 16 samples per instruction as 16 independent 4x4 blocks (3x3 padded), 3 rank-1
//             updates per product, 4 accumulator sets per wave for 64 samples.  D comes back as (lane = column, reg = row)
//             but the next A operand wants (lane = row): the per-step quad transpose (DPP + selects) is included.
//   mfma4_raw the same MFMAs without the transpose (the chain is then arithmetically meaningless): the matrix pipe's
//             upper bound for this shape.
//   mfma16    v_mfma_f32_16x16x4_f32 with a wave-uniform 3x3 constant on the A side (padded to 16x4) and 16 samples'
//             3-vectors as B columns — the "constant x batch" shape of R_fixed / inertia products: 144 useful MACs of 1024.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/ubench/mfma_probe tools/ubench/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int STEPS = 1024;

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// cheap per-sample, per-step operand: a rotation-like matrix from two running values (no memory traffic)
struct Jgen {
    float c, s;
    __device__ void next() { const float t = c * 0.999f - s * 0.04f; s = s * 0.999f + c * 0.04f; c = t; }
};

__global__ void __launch_bounds__(256) k_pk(float *out, float seed) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    Jgen g{1.0f - 1e-6f * (tid & 1023u), seed + 1e-6f * (tid & 1023u)};
    f2 A[3] = {{1, 0}, {0, 1}, {0, 0}};   // (R_c0, R_c1)
    float Z[3] = {0, 0, 1};               // R_c2
#pragma unroll 4
    for (int k = 0; k < STEPS; ++k) {
        g.next();
        // J = [[c, -s, s*c], [s, c, -s*s], [-s, 0.5 s, c]]  (general 3x3, values irrelevant)
        const f2 j0 = {g.c, -g.s}, j1 = {g.s, g.c}, j2 = {-g.s, 0.5f * g.s};
        const float j02 = g.s * g.c, j12 = -g.s * g.s, j22 = g.c;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f2 r0 = {A[c].x, A[c].x}, r1 = {A[c].y, A[c].y}, r2 = {Z[c], Z[c]};
            const f2 n = pk_fma(r2, j2, pk_fma(r1, j1, r0 * j0));
            const float z = fmaf(Z[c], j22, fmaf(A[c].y, j12, A[c].x * j02));
            A[c] = n;
            Z[c] = z;
        }
    }
    out[tid] = A[0].x + A[1].y + Z[2] + A[2].x + Z[0];
}

template <bool TRANSPOSE>
__global__ void __launch_bounds__(256) k_mfma4(float *out, float seed) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u, r = lane & 3u;   // lane = (block = sample, row / column index)
    Jgen g{1.0f - 1e-6f * (tid & 1023u), seed + 1e-6f * (tid & 1023u)};
    // four accumulator sets = 4 x 16 samples per wave.  a[s][k]: column k of R in A layout (lane = row)
    float a[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < 3; ++k) a[s][k] = (r == (unsigned)k) ? 1.0f : 0.0f;
#pragma unroll 2
    for (int k = 0; k < STEPS; ++k) {
        g.next();
        // row kk of J in B layout (lane = column): pick by this lane's column index
        const float b0 = r == 0 ? g.c : r == 1 ? -g.s : r == 2 ? g.s * g.c : 0.0f;
        const float b1 = r == 0 ? g.s : r == 1 ? g.c : r == 2 ? -g.s * g.s : 0.0f;
        const float b2 = r == 0 ? -g.s : r == 1 ? 0.5f * g.s : r == 2 ? g.c : 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f4 d = {0, 0, 0, 0};
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s][0], b0, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s][1], b1, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s][2], b2, d, 0, 0, 0);
            if (TRANSPOSE) {
                // d: lane (b, j) holds column j, reg = row i.  Next A_k wants lane (b, i) = D[i][k] = reg i of lane (b, k).
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    float t[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int v = __builtin_bit_cast(int, d[i]);
                        const int w = kk == 0   ? __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true)    // quad_perm [0,0,0,0]
                                      : kk == 1 ? __builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true)    // [1,1,1,1]
                                                : __builtin_amdgcn_mov_dpp(v, 0xaa, 0xf, 0xf, true);   // [2,2,2,2]
                        t[i] = __builtin_bit_cast(float, w);
                    }
                    a[s][kk] = r == 0 ? t[0] : r == 1 ? t[1] : r == 2 ? t[2] : t[3];
                }
            } else {
                a[s][0] = d[0]; a[s][1] = d[1]; a[s][2] = d[2];
            }
        }
    }
    float acc = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc += a[s][0] + a[s][1] + a[s][2];
    out[tid] = acc;
}

// constant (A side, wave-uniform 3x3 padded to 16x4) x 16 samples' 3-vectors (B columns): v <- F v per step
__global__ void __launch_bounds__(256) k_mfma16(float *out, float seed) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u, i = lane & 15u, kq = lane >> 4;
    const float fa = (i < 3 && kq < 3) ? (i == kq ? 0.999f : 0.01f * seed) : 0.0f;   // A[i][k]
    float b[4];                                                                      // 4 x 16 samples per wave
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = kq < 3 ? 1.0f + 1e-6f * (tid & 255u) : 0.0f;   // B[k][j]
#pragma unroll 2
    for (int k = 0; k < STEPS; ++k) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f4 d = {0, 0, 0, 0};
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, b[s], d, 0, 0, 0);
            b[s] = d[0];   // (the real chain would need the same row->k relayout as above; left out: upper bound)
        }
    }
    out[tid] = b[0] + b[1] + b[2] + b[3];
}

template <class F>
static double time_ms(F launch, hipStream_t s) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a, s)); launch(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int blocks = 256 * 16, threads = 256;   // 16 waves per SIMD in total, 4 resident blocks per CU at a time
    float *out; CK(hipMalloc(&out, sizeof(float) * blocks * threads));
    const double samples = (double)blocks * threads, prods = samples * STEPS;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; %d steps of R <- R J per sample, %.0f samples per launch\n", prop.gcnArchName,
           prop.multiProcessorCount, STEPS, samples);
    struct { const char *name; double ms; double useful_flops; } rows[4];
    rows[0] = {"pk        (v_pk_fma_f32, lane per sample)", time_ms([&] { hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(threads), 0, s, out, 0.01f); }, s), 54.0};
    rows[1] = {"mfma4     (4x4x1_16b + quad transpose)", time_ms([&] { hipLaunchKernelGGL(k_mfma4<true>, dim3(blocks), dim3(threads), 0, s, out, 0.01f); }, s), 54.0};
    rows[2] = {"mfma4_raw (4x4x1_16b, no relayout: bound)", time_ms([&] { hipLaunchKernelGGL(k_mfma4<false>, dim3(blocks), dim3(threads), 0, s, out, 0.01f); }, s), 54.0};
    rows[3] = {"mfma16    (16x16x4, constant x 16 vectors; F v only: 18 flop)", time_ms([&] { hipLaunchKernelGGL(k_mfma16, dim3(blocks), dim3(threads), 0, s, out, 0.01f); }, s), 18.0};
    for (auto &r : rows)
        printf("%-64s %8.3f ms  %7.2f ps per sample-step  %6.1f useful TFLOP/s\n", r.name, r.ms, r.ms * 1e9 / prods,
               prods * r.useful_flops / (r.ms * 1e-3) / 1e12);
    printf("3x3 (x) 3-vector by v_pk_fma_f32 for comparison: 1/3 of the pk row's time per step (9 of its 27 MACs)\n");
    return 0;
}
