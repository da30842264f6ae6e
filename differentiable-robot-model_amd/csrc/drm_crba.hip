// drm_crba.hip — K6: joint-space inertia matrix H(q) by the composite-rigid-body algorithm.
//
// Replaces DifferentiableRobotModel.compute_lagrangian_inertia_matrix (robot_model.py:402-450), which runs n + 1
// full inverse-dynamics passes (each ~11 k tiny torch ops) and subtracts the gravity pass; see drm_sample.hpp
// ("Joint-space inertia matrix") for why the composite-rigid-body form computes the same matrix.
//
// Per sample: in q[n] (4 n bytes), out H[n, n] (4 n^2 bytes).          n = 7: 28 + 196 = 224 B, ~2 kflop
// 7-DoF arm chains run crba_arm_kernel below; every other robot the loop-structured kernel.
#include "drm_common.hpp"
#include "drm_sample.hpp"
#include "drm_tree_dev.hpp"

namespace drm {

constexpr int CRBA_BLOCKS = 0, CRBA_TRIANGLE = 1, CRBA_DIRECT = 2; // how a segment's part of H reaches HBM (crba_tree_kernel<MODE>)
constexpr int CRBA_SHORT_OPS = 6; // segments of up to this many ops in a row take crba_tree_walk_short

// Loop-structured composite-rigid-body algorithm of any robot (drm_tree.hpp crba_tree_walk): one tile of 64 samples per
// block, one wavefront per segment.  The sub-trees off the fixed root give the diagonal blocks of H and everything between
// two of them is a structural zero (an Allegro hand: four 4 x 4 blocks in a 16 x 16 matrix), so a wavefront keeps only ITS
// block (cnt x cnt floats per sample) in LDS; once all are done the whole block of threads assembles the [64, n, n] rows —
// block entries where a row and a column belong to the same segment, zeros elsewhere — and writes them with coalesced
// 16-byte stores.  (Staging the full 256 n^2-byte tile instead allowed one block per CU for n = 16: 1 330 -> 697 us at
// 2^20 samples; this form: see profiles/.)
// LDS: [ table ][ q : 64 (n|1) ][ segment map : dof -> (segment's first dof, its dof count, its block's LDS offset) ]
//      shared, then per wavefront [ cos / sin / value per op : ops * 3 * 64 ][ inertia slots : n_slots * 10 * 64 ]
//      [ block : 64 (cnt^2 | 1), or (CRBA_TRIANGLE) 64 (cnt (cnt + 1) / 2 | 1), or (CRBA_DIRECT) nothing: lanes store their
//        entries straight to HBM over a zeroed H ]
template <int MODE>
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    crba_tree_kernel(TreeArgs a, const float *__restrict__ q, int64_t B, float *__restrict__ H, uint32_t magic_q, uint32_t align) {
    constexpr bool DIRECT = MODE == CRBA_DIRECT, TRI = MODE == CRBA_TRIANGLE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, nn = n * n;
    const int Sq = pad_odd(n);
    float *lq = smem + table_lds_floats(a.n_ops);
    int *lmap = reinterpret_cast<int *>(lq + round4(WAVE * Sq)); // [3][n]: segment lo, cnt, block offset (floats) per DoF
    const int first = a.seg_begin[wave], last = a.seg_begin[wave + 1];
    // staged per segment: the cnt x cnt block, or (TRI: one big segment) the lower triangle of the symmetric block
    const int lo = a.seg_dof_lo[wave], cnt = a.seg_dof_cnt[wave], Sb = pad_odd(TRI ? cnt * (cnt + 1) / 2 : cnt * cnt);
    float *ltr = smem + a.wave_off[wave];                          // [op - first][3][64]
    float *lis = ltr + (last - first) * (CRBA_PARK_FLOATS * WAVE); // inertia slots [slot][10][64]
    float *lb = lis + a.n_slots * (10 * WAVE);                     // this segment's block of H: [64][cnt^2 | 1]

    const TableLds tab = stage_tree_table(a, smem);
    if (wave == 0) tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, tc.full && (n & 1) && (align & AL_Q), tc.full && (align & AL_Q));
    for (int s = 0; s < a.n_slots * 10; ++s) lis[s * WAVE + lane] = 0.0f;
    if (!DIRECT) {
        for (int i = (int)lane; i < WAVE * Sb; i += WAVE) lb[i] = 0.0f; // pairs of joints on different branches of the segment
        for (int d = (int)lane; d < cnt; d += WAVE) {
            lmap[lo + d] = lo; lmap[n + lo + d] = cnt; lmap[2 * n + lo + d] = (int)(lb - smem);
        }
    }
    __syncthreads();

    const bool live = (int)lane < tc.rows;
    const float *qrow = lq + lane * Sq; // lanes past a partial tile's last row read zero angles, their H is never stored
    float *brow = lb + lane * Sb;
    float *hdst = H + (tc.b0 + lane) * nn;
    const TableLds &ctl = tab;
    auto hout = [&](int di, int dj, float v) {
        if (DIRECT) {
            if (live) hdst[di * n + dj] = v;
        } else if (TRI) {
            if (di >= dj) brow[tri_index(di - lo, dj - lo)] = v;
        } else {
            brow[(di - lo) * cnt + (dj - lo)] = v;
        }
    };
    auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; };
    // a short serial segment (a finger): the unrolled walk with the joint transforms in registers; anything else: the loop
    if (!crba_tree_walk_short<CRBA_SHORT_OPS>(first, last, ctl, [&](int k) { return tab.row(k); }, qf, hout)) {
        crba_prepare(first, last, ctl, qf, [&](int k, float c, float s, float x) {
            float *b = ltr + (k - first) * (CRBA_PARK_FLOATS * WAVE) + lane;
            b[0] = c; b[WAVE] = s; b[2 * WAVE] = x;
        });
        crba_tree_walk(
            first, last, ctl, [&](int k) { return tab.row(k); },
            [&](int k, float &c, float &s, float &x) {
                const float *b = ltr + (k - first) * (CRBA_PARK_FLOATS * WAVE) + lane;
                c = b[0]; s = b[WAVE]; x = b[2 * WAVE];
            },
            [&](int s, const Inertia &I) { lds_add_inertia(lis, s, lane, I); }, [&](int s, Inertia &I) { lds_take_inertia(lis, s, lane, I); },
            hout);
    }
    if (!DIRECT) {
        __syncthreads();
        // assembly: element (r, c) of sample b is the block entry when r and c belong to the same segment, else 0
        auto entry = [&](unsigned b, unsigned r, unsigned c) -> float {
            const int slo = lmap[r], scnt = lmap[n + r];
            const unsigned cc = c - (unsigned)slo, rr = r - (unsigned)slo;
            if (cc >= (unsigned)scnt) return 0.0f;
            if (!TRI) return smem[lmap[2 * n + r] + b * (unsigned)pad_odd(scnt * scnt) + rr * (unsigned)scnt + cc];
            const unsigned hi = rr > cc ? rr : cc, lw = rr > cc ? cc : rr;
            return smem[lmap[2 * n + r] + b * (unsigned)pad_odd(scnt * (scnt + 1) / 2) + hi * (hi + 1u) / 2u + lw];
        };
        float *g = H + tc.b0 * nn;
        if (tc.full && !(n & 3) && (align & AL_TAU)) {
            // one 16-byte store per thread and round, linear in the tile: thread i writes float4 i.  The (sample, row, column)
            // of a float4 is kept incrementally (no integer divisions in the loop) and its row's segment is looked up once.
            const unsigned per_row = (unsigned)n >> 2, per_sample = (unsigned)nn >> 2, total = WAVE * per_sample;
            const unsigned row_magic = per_row > 1u ? 0xffffffffu / per_row + 1u : 0u; // j / per_row for j < 2^16 (per_row = 1: j)
            unsigned b = threadIdx.x / per_sample, j = threadIdx.x - b * per_sample;
            for (unsigned i = threadIdx.x; i < total; i += blockDim.x) {
                const unsigned r = per_row > 1u ? __umulhi(j, row_magic) : j, c = (j - r * per_row) * 4u;
                const int slo = lmap[r], scnt = lmap[n + r];
                const unsigned rr = r - (unsigned)slo, rbase = TRI ? rr * (rr + 1u) / 2u : rr * (unsigned)scnt;
                const float *src = smem + lmap[2 * n + r] + b * (unsigned)pad_odd(TRI ? scnt * (scnt + 1) / 2 : scnt * scnt);
                float v[4];
#pragma unroll
                for (unsigned e = 0; e < 4u; ++e) {
                    const unsigned cc = c + e - (unsigned)slo;
                    if (TRI) // entry (rr, cc) of the symmetric block from its packed lower triangle
                        v[e] = cc >= (unsigned)scnt ? 0.0f : cc <= rr ? src[rbase + cc] : src[cc * (cc + 1u) / 2u + rr];
                    else
                        v[e] = cc < (unsigned)scnt ? src[rbase + cc] : 0.0f;
                }
                store16_wt(g + 4u * i, make_float4(v[0], v[1], v[2], v[3]));
                j += blockDim.x;
                while (j >= per_sample) { j -= per_sample; ++b; }
            }
        } else {
            const unsigned total = (unsigned)tc.rows * (unsigned)nn;
            for (unsigned i = threadIdx.x; i < total; i += blockDim.x) {
                const unsigned b = i / (unsigned)nn, j = i - b * (unsigned)nn, r = j / (unsigned)n;
                g[i] = entry(b, r, j - r * (unsigned)n);
            }
        }
    }
}

// Serial-chain ("arm") specialisation, full tiles only — the design of fk_jacobian_arm_kernel: constant rows staged
// once per wave in LDS, packed-FP32 sweeps without the int table (drm_sample.hpp crba_chain), preloaded kernel
// arguments, one basic block, H staged as a linear LDS image (n^2 = 49 is odd).
// LINKS: the links the sweep visits (NJ when the host folded the fixed tail into the last moving link, else CAP).
template <int CAP, int NJ, int LINKS>
__global__ void __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK)
    crba_arm_kernel(const float *__restrict__ ops_f, const float *__restrict__ q, int n_tiles, float *__restrict__ H) {
    static_assert((NJ & 1) && ((NJ * NJ) & 1), "odd row widths only (linear LDS images)");
    static_assert(CAP * DRM_OPF_STRIDE == 4 * WAVE, "one float4 per lane copies the constant table");
    constexpr int NN = NJ * NJ;
    constexpr int C_FLOATS = CAP * DRM_OPF_STRIDE, H_FLOATS = round4(WAVE * NN);
    constexpr int PER_WAVE = C_FLOATS + H_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[MAX_WAVES_PER_BLOCK * PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = (int)blockIdx.x * MAX_WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    const unsigned lane = threadIdx.x & 63u;
    float *lc = smem + wave * PER_WAVE;
    float *lh = lc + C_FLOATS;
    const int64_t b0 = (int64_t)tile * WAVE;

    float4 cv = reinterpret_cast<const float4 *>(ops_f)[lane];
    float qv[NJ];
    {
        const float *qrow = q + (b0 + lane) * NJ; // this lane's own row, straight into registers
#pragma unroll
        for (int d = 0; d < NJ; ++d) qv[d] = qrow[d];
    }
    pin(cv);
    reinterpret_cast<float4 *>(lc)[lane] = cv;
    wave_lds_sync();
    float *hrow = lh + lane * NN;
    crba_chain<LINKS, NJ>([&](int k) -> const float * { return lc + k * DRM_OPF_STRIDE; }, qv,
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
        const dim3 grid((unsigned)((n_tiles + MAX_WAVES_PER_BLOCK - 1) / MAX_WAVES_PER_BLOCK)), block(WAVE * MAX_WAVES_PER_BLOCK);
        if (arm_links(w) == 7) hipLaunchKernelGGL((crba_arm_kernel<8, 7, 7>), grid, block, 0, s, w->ops_f, q, n_tiles, H);
        else hipLaunchKernelGGL((crba_arm_kernel<8, 7, 8>), grid, block, 0, s, w->ops_f, q, n_tiles, H);
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done == B) return launched();
        rc = launched();
        if (rc) return rc;
        drm_walk generic = *w;
        generic.shape &= ~DRM_WALK_ARM_CHAIN;
        return drm_crba(&generic, q + done * n, B - done, H + done * nn, stream);
    }
#endif
    if (!segments_ok(w)) return fail(DRM_ERR_INVALID, "walk segments are inconsistent");
    if ((((uintptr_t)w->ops_f) & 15u) != 0) return fail(DRM_ERR_INVALID, "ops_f must be 16-byte aligned");
    TreeArgs a = tree_args(w);
    // how a segment's part of H is staged: its cnt x cnt block (small segments: the fingers of a hand), the lower
    // triangle of its symmetric block (a segment of more than 8 DoF: half the LDS), or not at all (DIRECT: lanes store their entries
    // straight to HBM over a memset) when staging would leave fewer than two blocks per CU — a big walk (an arm carrying a
    // hand, 23 DoF) is bound by the latency of its serial ancestor walks, so wavefronts per CU matter more to it than
    // coalesced stores (measured at 2^20: 3.7 ms direct against 11 ms with one staged wavefront per CU).
    auto plan = [&](TreeArgs &t, int mode) {
        const size_t shared = (size_t)table_lds_floats(t.n_ops) + round4(WAVE * pad_odd(n)) + round4(3 * n);
        return sizeof(float) * layout_waves(t, shared, CRBA_PARK_FLOATS * WAVE, t.n_slots * 10 * WAVE, [&](int sg) {
            const int c = t.seg_dof_cnt[sg];
            return mode == CRBA_DIRECT ? 0 : round4(WAVE * pad_odd(mode == CRBA_TRIANGLE ? c * (c + 1) / 2 : c * c));
        });
    };
    int max_cnt = 0;
    for (int sg = 0; sg < a.n_segments; ++sg) max_cnt = a.seg_dof_cnt[sg] > max_cnt ? a.seg_dof_cnt[sg] : max_cnt;
    int mode = max_cnt <= 8 ? CRBA_BLOCKS : CRBA_TRIANGLE; // (small blocks: the plain indexing of a full block is cheaper to assemble)
    if (plan(a, mode) > (size_t)MAX_LDS_BYTES && a.n_segments > 1) {
        a = tree_args(w, true);
        mode = CRBA_TRIANGLE;
    }
    if (plan(a, mode) > (size_t)MAX_LDS_BYTES / 2) mode = CRBA_DIRECT;
    const size_t lds = plan(a, mode);
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    const uint32_t align = al16(q, AL_Q) | al16(H, AL_TAU);
    if (mode == CRBA_DIRECT) {
        hipError_t e = hipMemsetAsync(H, 0, sizeof(float) * (size_t)B * nn, s);
        if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
        rc = ensure_lds_tree(crba_tree_kernel<CRBA_DIRECT>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(crba_tree_kernel<CRBA_DIRECT>, dim3((unsigned)tiles), dim3(WAVE * a.n_segments), lds, s, a, q, B, H, div_magic(n), align);
    } else if (mode == CRBA_TRIANGLE) {
        rc = ensure_lds_tree(crba_tree_kernel<CRBA_TRIANGLE>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(crba_tree_kernel<CRBA_TRIANGLE>, dim3((unsigned)tiles), dim3(WAVE * a.n_segments), lds, s, a, q, B, H, div_magic(n), align);
    } else {
        rc = ensure_lds_tree(crba_tree_kernel<CRBA_BLOCKS>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(crba_tree_kernel<CRBA_BLOCKS>, dim3((unsigned)tiles), dim3(WAVE * a.n_segments), lds, s, a, q, B, H, div_magic(n), align);
    }
    return launched();
}
