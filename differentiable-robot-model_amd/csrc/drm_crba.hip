// drm_crba.hip — K6: joint-space inertia matrix H(q) by the composite-rigid-body algorithm.
//
// Replaces DifferentiableRobotModel.compute_lagrangian_inertia_matrix (robot_model.py:402-450), which runs n + 1
// full inverse-dynamics passes (each ~11 k tiny torch ops) and subtracts the gravity pass; see drm_sample.hpp
// crba_walk for why the composite-rigid-body form computes the same matrix.
//
// Per sample: in q[n] (4 n bytes), out H[n, n] (4 n^2 bytes).          n = 7: 28 + 196 = 224 B, ~2 kflop
// LDS per wave: [ q : 64 (n|1) ][ H : 64 (n^2|1) ][ inertia slots : n_slots*10*64 ][ axis slots : n_slots*depth*6*64 ]
// (depth = DRM_WALK_BRANCH_DEPTH: a slot holds the axes of the branch point and of the ops above it)
// When the H tile does not fit in LDS (n > ~20) lanes store their entries straight to HBM (uncoalesced, rare).
#include "drm_common.hpp"
#include "drm_sample.hpp"

namespace drm {

template <int CAP, bool DIRECT>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    crba_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i, int n, int n_slots, int sdepth,
                int zero_fill,
                const float *__restrict__ q, int64_t B, float *__restrict__ H, uint32_t magic_q, uint32_t magic_h,
                int lds_per_wave, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    WaveCtx cx;
    if (!wave_begin(B, lds_per_wave, smem, cx)) return;
    const unsigned lane = cx.lane;
    const int nn = n * n;
    const int Sq = pad_odd(n), Sh = DIRECT ? 0 : pad_odd(nn);
    float *lq = cx.lds;
    float *lh = lq + round4(WAVE * Sq);
    float *lis = lh + round4(WAVE * Sh);              // inertia slots [slot][10][64]
    float *lss = lis + n_slots * (10 * WAVE);         // axis slots    [slot][op < sdepth][6][64]

    tile_load<0>(q + cx.b0 * n, cx.rows, n, magic_q, lq, lane, cx.full && (n & 1) && (align & AL_Q), cx.full && (align & AL_Q));
    for (int s = 0; s < n_slots * 10; ++s) lis[s * WAVE + lane] = 0.0f;
    float *hrow = lh + lane * Sh;
    if (!DIRECT && zero_fill)
        for (int i = 0; i < nn; ++i) hrow[i] = 0.0f; // pairs of joints on different branches
    wave_lds_sync();

    const bool live = (int)lane < cx.rows;
    const float *qrow = lq + lane * Sq; // lanes past a partial tile's last row compute garbage, never stored
    float *hdst = H + (cx.b0 + lane) * nn;
    // lanes past a partial tile read zeros (not stale LDS): their angles must not be able to push the wave onto
    // the rare large-angle sincos path, which would change the rounding of the live lanes from run to run
    auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; };
    auto islot_add = [&](int s, const Inertia &a) {
        float *b = lis + s * (10 * WAVE) + lane;
        b[0] += a.m;
#pragma unroll
        for (int i = 0; i < 3; ++i) b[(1 + i) * WAVE] += a.h[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) b[(4 + i) * WAVE] += a.I[i];
    };
    auto islot_take = [&](int s, Inertia &a) {
        float *b = lis + s * (10 * WAVE) + lane;
        a.m += b[0]; b[0] = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { a.h[i] += b[(1 + i) * WAVE]; b[(1 + i) * WAVE] = 0.0f; }
#pragma unroll
        for (int i = 0; i < 6; ++i) { a.I[i] += b[(4 + i) * WAVE]; b[(4 + i) * WAVE] = 0.0f; }
    };
    auto sslot_save = [&](int s, int j, const Axis &a) {
        float *b = lss + (s * sdepth + j) * (6 * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) { b[i * WAVE] = a.ang[i]; b[(3 + i) * WAVE] = a.lin[i]; }
    };
    auto sslot_load = [&](int s, int j, Axis &a) {
        const float *b = lss + (s * sdepth + j) * (6 * WAVE) + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) { a.ang[i] = b[i * WAVE]; a.lin[i] = b[(3 + i) * WAVE]; }
    };
    auto hout = [&](int di, int dj, float v) {
        if (DIRECT) {
            if (live) hdst[di * n + dj] = v;
        } else {
            hrow[di * n + dj] = v;
        }
    };
    crba_walk<CAP>(ops_f, ops_i, qf, islot_add, islot_take, sslot_save, sslot_load, hout);
    if (!DIRECT) {
        wave_lds_sync();
        tile_store<0>(H + cx.b0 * nn, cx.rows, nn, magic_h, lh, lane, cx.full && (nn & 1) && (align & AL_TAU), cx.full && (align & AL_TAU));
    }
}

// Serial-chain ("arm") specialisation, full tiles only — the design of fk_jacobian_arm_kernel: constant rows staged
// once per wave in LDS, packed-FP32 sweeps without the int table (drm_sample.hpp crba_chain), preloaded kernel
// arguments, one basic block, H staged as a linear LDS image (n^2 = 49 is odd).
template <int CAP, int NJ>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    crba_arm_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, int n_tiles, float *__restrict__ H) {
    static_assert((NJ & 1) && ((NJ * NJ) & 1), "odd row widths only (linear LDS images)");
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int NN = NJ * NJ;
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, Q_FLOATS = round4(WAVE * NJ), H_FLOATS = round4(WAVE * NN);
    constexpr int PER_WAVE = C_FLOATS + Q_FLOATS + H_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[MAX_WAVES_PER_BLOCK * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = (int)blockIdx.x * MAX_WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * PER_WAVE;
    float *lq = lc + C_FLOATS, *lh = lq + Q_FLOATS;
    const int64_t b0 = (int64_t)tile * WAVE;

    float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];
    tile_load<NJ>(q + b0 * NJ, WAVE, NJ, 0u, lq, lane, true);
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    wave_lds_sync();

    float qv[NJ];
#pragma unroll
    for (int d = 0; d < NJ; ++d) qv[d] = lq[lane * NJ + d];
    float *hrow = lh + lane * NN;
    crba_chain<CAP, NJ>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, qv,
                        [&](int i, int j, float v) { hrow[i * NJ + j] = v; });
    wave_lds_sync();
    tile_store<NN>(H + b0 * NN, WAVE, NN, 0u, lh, lane, true);
}

} // namespace drm

using namespace drm;

extern "C" int drm_crba(const drm_walk *w, const float *q, int64_t B, float *H, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !H) return fail(DRM_ERR_INVALID, "q / H must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs, nn = n * n;
    hipStream_t s = (hipStream_t)stream;
#ifndef DRM_NO_ARM_KERNEL
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7 && B >= WAVE && B / WAVE < 0x7fffffffLL &&
        (((uintptr_t)q | (uintptr_t)H | (uintptr_t)w->ops_f) & 15u) == 0) {
        // 7-DoF arms: full tiles through the packed-FP32 chain kernel, ragged tail through the generic one
        const int n_tiles = (int)(B / WAVE);
        hipLaunchKernelGGL((crba_arm_kernel<8, 7>),
                           dim3((unsigned)((n_tiles + MAX_WAVES_PER_BLOCK - 1) / MAX_WAVES_PER_BLOCK)),
                           dim3(WAVE * MAX_WAVES_PER_BLOCK), 0, s, w->ops_f, q, n_tiles, H);
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done == B) return launched();
        rc = launched();
        if (rc) return rc;
        drm_walk generic = *w;
        generic.shape &= ~DRM_WALK_ARM_CHAIN;
        return drm_crba(&generic, q + done * n, B - done, H + done * nn, stream);
    }
#endif
    const int sdepth = DRM_WALK_BRANCH_DEPTH(w->shape);
    if (w->n_slots > 0 && sdepth == 0) return fail(DRM_ERR_INVALID, "walk has save slots but no branch depth in shape");
    const int base = round4(WAVE * pad_odd(n)) + w->n_slots * (10 + sdepth * 6) * WAVE;
    // the H tile of a wave is 256 n^2 bytes: beyond 32 KB (n >= 12) it decides how many waves a CU can hold, and a
    // branching robot's H is mostly structural zeros — write the entries straight to HBM over a memset instead
    // (Allegro, n = 16, 65 536 samples: see profiles/r01_kernel_times.txt)
    const bool direct = (size_t)round4(WAVE * pad_odd(nn)) * sizeof(float) > (size_t)32 * 1024 ||
                        (size_t)(base + round4(WAVE * pad_odd(nn))) * sizeof(float) > (size_t)MAX_LDS_BYTES;
    Geometry g;
    rc = make_geometry(B, base + (direct ? 0 : round4(WAVE * pad_odd(nn))), g);
    if (rc) return rc;
    const int zero_fill = (w->shape & DRM_WALK_ARM_CHAIN) ? 0 : 1; // a chain over all DoFs writes every entry
    if (direct && zero_fill) {
        hipError_t e = hipMemsetAsync(H, 0, sizeof(float) * (size_t)B * nn, s);
        if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    const uint32_t align = al16(q, AL_Q) | al16(H, AL_TAU);
    DRM_DISPATCH_CAP(w->capacity, {
        if (direct) {
            rc = ensure_lds(crba_kernel<C, true>, g.lds_bytes);
            if (rc) return rc;
            hipLaunchKernelGGL((crba_kernel<C, true>), g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, n,
                               (int)w->n_slots, sdepth, zero_fill, q, B, H, div_magic(n), div_magic(nn), g.lds_per_wave, align);
        } else {
            rc = ensure_lds(crba_kernel<C, false>, g.lds_bytes);
            if (rc) return rc;
            hipLaunchKernelGGL((crba_kernel<C, false>), g.grid, g.block, g.lds_bytes, s, w->ops_f, w->ops_i, n,
                               (int)w->n_slots, sdepth, zero_fill, q, B, H, div_magic(n), div_magic(nn), g.lds_per_wave, align);
        }
    })
    return launched();
}
