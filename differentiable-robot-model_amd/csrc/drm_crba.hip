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
#include "drm_static.hpp"

namespace drm {

constexpr int CRBA_SHORT_OPS = 6; // segments of up to this many ops in a row take crba_tree_walk_short

// Robots whose segments are all SHORT (the fingers of a hand): one tile of 64 samples per block, one wavefront per segment.
// The sub-trees off the fixed root give the diagonal blocks of H and everything between two of them is a structural zero (an
// Allegro hand: four 4 x 4 blocks in a 16 x 16 matrix), so a wavefront keeps only ITS block (cnt x cnt floats per sample) in
// LDS; once all are done the whole block of threads assembles the [64, n, n] rows — block entries where a row and a column
// belong to the same segment, zeros elsewhere — and writes them with coalesced 16-byte stores.  (Staging the full
// 256 n^2-byte tile instead allowed one block per CU for n = 16: 1 330 -> 697 us at 2^20 samples; this form: see profiles/.)
// LDS: [ table ][ q : 64 (n|1) ][ segment map : dof -> (segment's first dof, its dof count, its block's LDS offset) ]
//      shared, then per wavefront [ inertia slots : n_slots * 10 * 64 ][ block : 64 (cnt^2 | 1) ]
template <bool NT>
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    crba_tree_kernel(TreeArgs a, const float *__restrict__ q, int64_t B, float *__restrict__ H, uint32_t magic_q, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileCtx tc = tile_begin(B);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, nn = n * n;
    const int Sq = pad_odd(n);
    float *lq = smem + table_lds_floats(a.n_ops);
    int *lmap = reinterpret_cast<int *>(lq + round4(WAVE * Sq)); // [3][n]: segment lo, cnt, block offset (floats) per DoF
    const int first = a.seg_begin[wave], last = a.seg_begin[wave + 1];
    const int lo = a.seg_dof_lo[wave], cnt = a.seg_dof_cnt[wave], Sb = pad_odd(cnt * cnt);
    float *lis = smem + a.wave_off[wave];        // inertia slots [slot][10][64]
    float *lb = lis + a.n_slots * (10 * WAVE);   // this segment's block of H: [64][cnt^2 | 1]

    const TableLds tab = stage_tree_table(a, smem);
    if (wave == 0) tile_load<0>(q + tc.b0 * n, tc.rows, n, magic_q, lq, lane, tc.full && (n & 1) && (align & AL_Q), tc.full && (align & AL_Q));
    for (int s = 0; s < a.n_slots * 10; ++s) lis[s * WAVE + lane] = 0.0f;
    for (int i = (int)lane; i < WAVE * Sb; i += WAVE) lb[i] = 0.0f; // pairs of joints on different branches of the segment
    for (int d = (int)lane; d < cnt; d += WAVE) {
        lmap[lo + d] = lo; lmap[n + lo + d] = cnt; lmap[2 * n + lo + d] = (int)(lb - smem);
    }
    __syncthreads();

    const bool live = (int)lane < tc.rows;
    const float *qrow = lq + lane * Sq; // lanes past a partial tile's last row read zero angles, their H is never stored
    float *brow = lb + lane * Sb;
    const TableLds &ctl = tab;
    auto hout = [&](int di, int dj, float v) { brow[(di - lo) * cnt + (dj - lo)] = v; };
    auto qf = [&](int d) -> float { return live ? qrow[d] : 0.0f; };
    // a serial segment (a finger): the unrolled walk with the joint transforms in registers; a short segment that branches
    // takes the loop, its cos / sin recomputed where the loop asks for them
    if (!crba_tree_walk_short<CRBA_SHORT_OPS>(first, last, ctl, [&](int k) { return tab.row(k); }, qf, hout)) {
        crba_tree_walk(
            first, last, ctl, [&](int k) { return tab.row(k); },
            [&](int k, float &c, float &s, float &x) {
                int w0, w1;
                ctl_words(ctl, k, w0, w1);
                const OpCtl ct = decode_ctl(w0, w1);
                x = 0.0f; c = 1.0f; s = 0.0f;
                if (ct.dof >= 0) {
                    x = qf(ct.dof);
                    if (!ct.prismatic) sincos_one(x, s, c);
                }
            },
            [&](int s, const Inertia &I) { lds_add_inertia(lis, s, lane, I); }, [&](int s, Inertia &I) { lds_take_inertia(lis, s, lane, I); },
            hout);
    }
    __syncthreads();
    // assembly: element j = (r, c) of a matrix is a block entry when r and c belong to the same segment, else 0 — where sample
    // 0's value sits in LDS (-1: structural zero) and the per-sample stride of its block, worked out once per element
    int *loff = lmap + 3 * n, *lstr = loff + nn;
    for (int j = (int)threadIdx.x; j < nn; j += (int)blockDim.x) {
        const int r = j / n, c = j - r * n;
        const int slo = lmap[r], scnt = lmap[n + r];
        const unsigned cc = (unsigned)(c - slo);
        loff[j] = cc < (unsigned)scnt ? lmap[2 * n + r] + (r - slo) * scnt + (int)cc : -1;
        lstr[j] = pad_odd(scnt * scnt);
    }
    __syncthreads();
    float *g = H + tc.b0 * nn;
    if (tc.full && (align & AL_TAU)) {
        // the tile's 64 matrices are one run of 64 n^2 floats, 16-byte aligned as a whole (a single matrix is not when n^2 is
        // odd): 16 n^2 float4, thread i writes float4 i, i + threads, ...; the (sample, element) of its first float is kept
        // incrementally, the other three follow with one wrap test each
        const int total4 = 16 * nn, stride = 4 * (int)blockDim.x;
        const int inc_b = stride / nn, inc_j = stride - inc_b * nn;
        int b = (4 * (int)threadIdx.x) / nn, j = 4 * (int)threadIdx.x - b * nn;
        for (int i = (int)threadIdx.x; i < total4; i += (int)blockDim.x) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (a float4 spans up to four matrices when n = 1, nn = 1: round 5 — found by the random-tree goldens, a 1-DoF
                // robot's rows 2 and 3 of every four came out zero; one wrap test was enough only for nn >= 4)
                int jj = j + e, bb = b;
                while (jj >= nn) { jj -= nn; ++bb; }
                const int o = loff[jj];
                v[e] = o >= 0 ? smem[o + bb * lstr[jj]] : 0.0f;
            }
            store16_wt<NT>(g + 4 * i, make_float4(v[0], v[1], v[2], v[3]));
            b += inc_b; j += inc_j;
            if (j >= nn) { j -= nn; ++b; }
        }
    } else {
        const int total = tc.rows * nn, stride = (int)blockDim.x;
        const int inc_b = stride / nn, inc_j = stride - inc_b * nn;
        int b = (int)threadIdx.x / nn, j = (int)threadIdx.x - b * nn;
        for (int i = (int)threadIdx.x; i < total; i += stride) {
            const int o = loff[j];
            g[i] = o >= 0 ? smem[o + b * lstr[j]] : 0.0f;
            b += inc_b; j += inc_j;
            if (j >= nn) { j -= nn; ++b; }
        }
    }
}

// LDS bytes of that launch (0: does not apply — a segment is too long, or the blocks do not fit)
static size_t crba_short_plan(const drm_walk *w, TreeArgs &a) {
    a = tree_args(w, false);
    if (a.max_seg_ops > CRBA_SHORT_OPS) return 0;
    const int n = a.n;
    const size_t shared = (size_t)table_lds_floats(a.n_ops) + round4(WAVE * pad_odd(n)) + round4(3 * n + 2 * n * n);
    const size_t lds = sizeof(float) * layout_waves(a, shared, 0, a.n_slots * 10 * WAVE, [&](int sg) {
        const int c = a.seg_dof_cnt[sg];
        return round4(WAVE * pad_odd(c * c));
    });
    return lds <= (size_t)MAX_LDS_BYTES ? lds : 0;
}

// Every other robot (an arm carrying a gripper or a hand, a mobile manipulator: a segment of more than CRBA_SHORT_OPS ops).
// The walk is drm_tree.hpp crba_set_walk: the forces of all the joints below an op travel up the tree together (LDS, 6 floats
// per joint and sample), every op's transform is built once.
// A 23-DoF segment's part of H is 2 KB per sample: staged in LDS, even as a triangle, it left one wavefront per CU, and lanes
// storing their entries straight to H — 4 bytes each, 2 KB apart, 64 cache lines per instruction — kept the address path
// busy instead.  Here the lower triangle of a segment's block goes to HBM scratch, [entry][64] per block and segment
// (coalesced; the grid is PERSISTENT, so the scratch is sized by what the chip holds at once and stays in L2), and after the
// walk every wavefront writes the rows of ITS segment: a row's entries come back from scratch with coalesced loads, are turned
// through a [64][n|1] LDS buffer and leave as runs of n floats per sample.
// LDS: [ table ][ q : 64 (n|1) ] shared, then per wavefront
//      [ inertia slots : n_slots * 10 * 64 ][ forces : cnt * 6 * 64, later the row buffer : 64 (n|1) ][ sub-tree tables ]
__global__ void __launch_bounds__(WAVE *DRM_MAX_SEGMENTS)
    crba_rows_kernel(TreeArgs a, int nt_max, const float *__restrict__ q, int64_t B, int n_tiles, float *__restrict__ H,
                     float *__restrict__ scratch, uint32_t magic_q, uint32_t align) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const int n = a.n, nn = n * n, Sq = pad_odd(n);
    float *lq = smem + table_lds_floats(a.n_ops);
    const int first = a.seg_begin[wave], last = a.seg_begin[wave + 1], n_seg_ops = last - first;
    const int lo = a.seg_dof_lo[wave], cnt = a.seg_dof_cnt[wave], nt = cnt * (cnt + 1) / 2;
    float *lis = smem + a.wave_off[wave];          // inertia slots [slot][10][64]
    float *lfo = lis + a.n_slots * (10 * WAVE);    // forces [slot of a moving op][6][64]; after the walk: one row of H, [64][n|1]
    float *lrow = lfo;
    const int fo_floats = cnt * (6 * WAVE) > round4(WAVE * Sq) ? cnt * (6 * WAVE) : round4(WAVE * Sq);
    int *t_lo = reinterpret_cast<int *>(lfo + fo_floats), *t_end = t_lo + n_seg_ops + 1, *t_dof = t_end + n_seg_ops;
    // this wavefront's triangles in scratch, SAMPLE-major [64][ntp]: a lane scatters its own entries during the walk (one store
    // per entry), the rows are then written sample by sample from coalesced reads
    const int ntp = round4(nt);
    float *tri = scratch + ((int64_t)blockIdx.x * a.n_segments + wave) * (int64_t)nt_max * WAVE;
    const TableLds tab = stage_tree_table(a, smem);
    for (int s = 0; s < a.n_slots * 10; ++s) lis[s * WAVE + lane] = 0.0f; // (every take leaves its slot at zero again)
    __syncthreads();
    const TableLds &ctl = tab;
    // (every lane writes the same wave-uniform values)
    crba_set_tables(first, last, ctl, [&](int k, int v) { t_lo[k - first] = v; }, [&](int k, int v) { t_end[k - first] = v; },
                    [&](int k) { return __builtin_amdgcn_readfirstlane(t_end[k - first]); }, [&](int m, int d) { t_dof[m] = d; });

#pragma unroll 1
    for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
        __syncthreads(); // every wavefront is done with the previous tile's q
        const int64_t b0 = (int64_t)tile * WAVE;
        const int rows = B - b0 < WAVE ? (int)(B - b0) : WAVE;
        const bool full = rows == WAVE;
        if (wave == 0) tile_load<0>(q + b0 * n, rows, n, magic_q, lq, lane, full && (n & 1) && (align & AL_Q), full && (align & AL_Q));
        for (int i = (int)lane; i < 16 * ntp; i += WAVE) // pairs of joints on different branches of the segment stay zero
            reinterpret_cast<float4 *>(tri)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        __syncthreads();
        const bool live = (int)lane < rows;
        const float *qrow = lq + lane * Sq; // lanes past a partial tile's last row read zero angles, their H is never stored
        crba_set_walk(
            first, last, ctl, [&](int k) { return tab.row(k); },
            [&](int k, float &c, float &s, float &x) {
                int w0, w1;
                ctl_words(ctl, k, w0, w1);
                const OpCtl ct = decode_ctl(w0, w1);
                x = 0.0f; c = 1.0f; s = 0.0f;
                if (ct.dof >= 0) {
                    x = live ? qrow[ct.dof] : 0.0f;
                    if (!ct.prismatic) sincos_one(x, s, c);
                }
            },
            [&](int k) { return __builtin_amdgcn_readfirstlane(t_lo[k - first]); },
            [&](int k) { return __builtin_amdgcn_readfirstlane(t_lo[__builtin_amdgcn_readfirstlane(t_end[k - first]) + 1 - first]); },
            [&](int m) { return __builtin_amdgcn_readfirstlane(t_dof[m]); },
            [&](int m, Force &F) {
                const float *b = lfo + m * (6 * WAVE) + lane;
#pragma unroll
                for (int i = 0; i < 3; ++i) F.la[i] = f2_make(b[i * WAVE], b[(3 + i) * WAVE]);
            },
            [&](int m, const Force &F) {
                float *b = lfo + m * (6 * WAVE) + lane;
#pragma unroll
                for (int i = 0; i < 3; ++i) { b[i * WAVE] = F.la[i][0]; b[(3 + i) * WAVE] = F.la[i][1]; }
            },
            [&](int s, const Inertia &I) { lds_add_inertia(lis, s, lane, I); }, [&](int s, Inertia &I) { lds_take_inertia(lis, s, lane, I); },
            [&](int di, int dj, float v) {
                if (di >= dj) tri[lane * ntp + tri_index(di - lo, dj - lo)] = v;
            });
        wave_lds_sync(); // the forces are dead: their LDS becomes the row buffer
        // rows lo .. lo + cnt - 1 of H for the tile's samples: block entries inside the segment's columns, zeros outside.
        // G samples per round: their triangles come back with 16-byte loads, twelve in flight per lane (one at a time, each
        // would wait out a round trip to L2 / Infinity Cache), into the LDS the forces have left; then every sample's cnt x n
        // floats leave as 4-byte stores of consecutive lanes — consecutive addresses.
        float *g = H + b0 * nn;
        // (a multiple of 4 keeps the float4 copies aligned; a segment without joints — a camera on a fixed mount — has no rows)
        const int G0 = ntp == 0 ? WAVE : fo_floats / ntp < WAVE ? (fo_floats / ntp) & ~3 : WAVE;
        const int G = G0 > 0 ? G0 : 1, per_sample = cnt * n;
        const unsigned step_r = WAVE / (unsigned)n, step_c = WAVE - step_r * (unsigned)n; // 64 = step_r * n + step_c
        for (int s0 = 0; s0 < rows && cnt > 0; s0 += G) {
            const int gs = rows - s0 < G ? rows - s0 : G;
            const float4 *t4 = reinterpret_cast<const float4 *>(tri + s0 * ntp);
            float4 *l4 = reinterpret_cast<float4 *>(lrow);
            const int n4 = gs * (ntp >> 2);
            for (int i0 = 0; i0 < n4; i0 += 12 * WAVE) {
                float4 v[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int i = i0 + u * WAVE + (int)lane;
                    v[u] = i < n4 ? t4[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int i = i0 + u * WAVE + (int)lane;
                    if (i < n4) l4[i] = v[u];
                }
            }
            wave_lds_sync();
            // element j of a sample's cnt x n floats = (row j / n, column j % n), kept incrementally; which triangle entry it is
            // does not depend on the sample, so it is worked out once and the samples are an inner loop of independent copies
            unsigned r = lane / (unsigned)n, c = lane - r * (unsigned)n;
            for (int j = (int)lane; j < per_sample; j += WAVE) {
                const int cc = (int)c - lo;
                const bool inside = cc >= 0 && cc < cnt;
                const int hi = (int)r > cc ? (int)r : cc, lw = (int)r > cc ? cc : (int)r;
                const float *src = lrow + (inside ? tri_index(hi, lw) : 0);
                float *dst = g + (int64_t)s0 * nn + lo * n + j;
#pragma unroll 8
                for (int gi = 0; gi < gs; ++gi) dst[(int64_t)gi * nn] = inside ? src[gi * ntp] : 0.0f;
                r += step_r; c += step_c;
                if (c >= (unsigned)n) { c -= (unsigned)n; ++r; }
            }
            wave_lds_sync();
        }
    }
}

struct CrbaRowsPlan {
    TreeArgs a;
    int nt_max, resident;
    size_t lds;
};
static int crba_rows_plan(const drm_walk *w, CrbaRowsPlan &p) {
    // one wavefront per segment; all the segments through one wavefront when their private areas do not fit side by side
    for (int single = 0; single < 2; ++single) {
        p.a = tree_args(w, single != 0);
        const int n = p.a.n;
        p.nt_max = 1;
        for (int sg = 0; sg < p.a.n_segments; ++sg) {
            const int c = p.a.seg_dof_cnt[sg];
            if (c * (c + 1) / 2 > p.nt_max) p.nt_max = c * (c + 1) / 2;
        }
        p.nt_max = round4(p.nt_max); // (the triangles are copied with 16-byte accesses)
        const size_t shared = (size_t)table_lds_floats(p.a.n_ops) + round4(WAVE * pad_odd(n));
        p.lds = sizeof(float) * layout_waves(p.a, shared, 0, p.a.n_slots * 10 * WAVE, [&](int sg) {
            const int forces = p.a.seg_dof_cnt[sg] * 6 * WAVE, row = round4(WAVE * pad_odd(n));
            const int ops = p.a.seg_begin[sg + 1] - p.a.seg_begin[sg];
            return (forces > row ? forces : row) + round4(2 * ops + 1 + p.a.seg_dof_cnt[sg]);
        });
        if (p.lds <= (size_t)MAX_LDS_BYTES || w->n_segments <= 1) break;
    }
    int rc = ensure_lds_tree(crba_rows_kernel, p.lds);
    if (rc) return rc;
    return resident_blocks(crba_rows_kernel, WAVE * p.a.n_segments, p.lds, p.resident);
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

// The fingers of a hand (DRM_WALK_FINGERS: K serial chains of L ops off the root), small ones, large launches: one wavefront walks
// the whole hand of a tile (drm_static.hpp crba_shape_body with P = 0) instead of a wavefront per finger (crba_tree_kernel) —
// TriFinger at 2^19 rows: 60 -> 31 us.
#ifndef DRM_CRBA_TREE_MIN_TILES
#define DRM_CRBA_TREE_MIN_TILES 2048 /* at 1 024 tiles (one per SIMD) the wavefront-per-sub-chain kernels are still ahead: 12.7 vs 13.1 us (Panda with gripper), 16.2 vs 18.4 (Jaco) */
#endif
// (the ops of a DRM_WALK_FINGERS walk are all revolute joints, op k driving DoF column k: the PLAIN form)
template <int K, int L, bool NT>
__global__ void __launch_bounds__(WAVE) crba_fingers_tree_kernel(const float *__restrict__ ops_f, const int32_t *__restrict__ ops_i,
                                                                 const float *__restrict__ q, int cap, int n, int n_tiles,
                                                                 float *__restrict__ H) {
    crba_shape_body<ShapeTree<0, K, L>, NT, true>(ops_f, ops_i, q, cap, n, n_tiles, H);
}
// rows covered (full tiles), 0: not this walk / launch
static int64_t launch_crba_fingers_tree(const drm_walk *w, const float *q, int64_t B, float *H, hipStream_t s) {
#ifdef DRM_NO_CRBA_TREE
    return 0;
#else
    if (!(w->shape & DRM_WALK_FINGERS) || B / WAVE < DRM_CRBA_TREE_MIN_TILES || B / WAVE >= 0x7fffffffLL ||
        (((uintptr_t)w->ops_f | (uintptr_t)H) & 15u) != 0)
        return 0;
    const int K = DRM_WALK_AH_K(w->shape), L = DRM_WALK_AH_L(w->shape);
    if (K * L != w->n_ops || w->n_dofs != w->n_ops || K * L > 12) return 0;
    const int n_tiles = (int)(B / WAVE);
#ifdef DRM_CRBA_TREE_NEVER_NT
    const bool nt = false;
#else
    const bool nt = stream_past_llc((int64_t)n_tiles * WAVE * w->n_dofs * w->n_dofs * 4);
#endif
#define X(k, l)                                                                                                                  \
    if (K == k && L == l) {                                                                                                      \
        if (nt)                                                                                                                  \
            hipLaunchKernelGGL((crba_fingers_tree_kernel<k, l, true>), dim3((unsigned)n_tiles), dim3(WAVE), 0, s, w->ops_f, w->ops_i, q,  \
                               (int)w->capacity, (int)w->n_dofs, n_tiles, H);                                                    \
        else                                                                                                                     \
            hipLaunchKernelGGL((crba_fingers_tree_kernel<k, l, false>), dim3((unsigned)n_tiles), dim3(WAVE), 0, s, w->ops_f, w->ops_i, q, \
                               (int)w->capacity, (int)w->n_dofs, n_tiles, H);                                                    \
        return (int64_t)n_tiles * WAVE;                                                                                          \
    }
    X(2, 2) X(3, 2) X(4, 2) X(2, 3) X(3, 3) X(4, 3) X(2, 4) X(3, 4)
#undef X
    return 0;
#endif
}

} // namespace drm

using namespace drm;

static int64_t drm_crba_scratch_floats_impl(const drm_walk *w, int64_t B, bool aligned) {
    if (check_walk(w) || B <= 0 || !segments_ok(w)) return 0;
    // (full aligned tiles of these walks run straight-line kernels without scratch: sized for the ragged tail and for a misaligned
    // call, drm_common.hpp fast_path_scratch_tiles)
    const bool fast = ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && w->n_dofs == 7) || crba_arm_hand_applies(w) ||
                      w->special[DRM_SPECIAL_CRBA] != nullptr;
    TreeArgs a;
    if (crba_short_plan(w, a)) return 0;
    CrbaRowsPlan p;
    if (crba_rows_plan(w, p)) return 0;
    const int64_t tiles = fast ? (aligned ? (B % WAVE ? 1 : 0) : fast_path_scratch_tiles(B)) : (B + WAVE - 1) / WAVE;
    return (tiles < p.resident ? tiles : (int64_t)p.resident) * p.a.n_segments * p.nt_max * WAVE;
}
extern "C" int64_t drm_crba_scratch_floats(const drm_walk *w, int64_t B) { return drm_crba_scratch_floats_impl(w, B, false); }
// ... for a caller that GUARANTEES 16-byte aligned q / qd / qdd (f) / outputs (both Python bindings do: they clone a misaligned
// slice): the full tiles of a 7-DoF arm / an arm with a hand then run straight-line kernels that need no scratch — only a ragged
// tail's one tile is sized
extern "C" int64_t drm_crba_scratch_floats_aligned(const drm_walk *w, int64_t B) { return drm_crba_scratch_floats_impl(w, B, true); }

extern "C" int drm_crba(const drm_walk *w, const float *q, int64_t B, float *H, float *scratch, void *stream) {
    int rc = check_walk(w);
    if (rc) return rc;
    if (!q || !H) return fail(DRM_ERR_INVALID, "q / H must not be NULL");
    if (B < 0) return fail(DRM_ERR_INVALID, "negative batch");
    if (B == 0) return DRM_OK;
    const int n = w->n_dofs, nn = n * n;
    hipStream_t s = (hipStream_t)stream;
    if (w->special[DRM_SPECIAL_CRBA] && B >= WAVE && B / WAVE < 0x7fffffffLL && (((uintptr_t)H | (uintptr_t)w->ops_f) & 15u) == 0) {
        // the robot's own straight-line kernel (csrc/drm_static.hpp crba_static_walk, built for exactly this walk): full tiles,
        // no scratch; q at any alignment, H 16-byte aligned (the tile's matrices leave as 16-byte stores)
        int n_tiles = (int)(B / WAVE);
        void *args[] = {(void *)&w->ops_f, (void *)&q, (void *)&n_tiles, (void *)&H};
        hipError_t e = hipModuleLaunchKernel((hipFunction_t)w->special[DRM_SPECIAL_CRBA], (unsigned)n_tiles, 1, 1, WAVE, 1, 1, 0, s, args, nullptr);
        if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_crba_static): %s", hipGetErrorString(e));
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done == B) return DRM_OK;
        drm_walk rest = *w;
        rest.special[DRM_SPECIAL_CRBA] = nullptr;
        return drm_crba(&rest, q + done * n, B - done, H + done * nn, scratch, stream);
    }
#ifndef DRM_NO_ARM_KERNEL
    if ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7 && B >= WAVE && B / WAVE < 0x7fffffffLL &&
        (((uintptr_t)q | (uintptr_t)H | (uintptr_t)w->ops_f) & 15u) == 0) {
        // 7-DoF arms: full tiles through the packed-FP32 chain kernel, ragged tail through the generic one
        int n_tiles = (int)(B / WAVE);
        if (w->special[DRM_SPECIAL_CRBA_ARM]) {
            // this arm's own kernel, its constants folded into the instruction stream (csrc/drm_arm_static.hpp, specialize.py)
            void *args[] = {(void *)&q, (void *)&n_tiles, (void *)&H};
            hipError_t e = hipModuleLaunchKernel((hipFunction_t)w->special[DRM_SPECIAL_CRBA_ARM], (unsigned)n_tiles, 1, 1, WAVE, 1, 1, 0, s, args, nullptr);
            if (e != hipSuccess) return fail(DRM_ERR_LAUNCH, "hipModuleLaunchKernel(drm_crba_arm_static): %s", hipGetErrorString(e));
        } else {
            const dim3 grid((unsigned)((n_tiles + MAX_WAVES_PER_BLOCK - 1) / MAX_WAVES_PER_BLOCK)), block(WAVE * MAX_WAVES_PER_BLOCK);
            if (arm_links(w) == 7) hipLaunchKernelGGL((crba_arm_kernel<8, 7, 7>), grid, block, 0, s, w->ops_f, q, n_tiles, H);
            else hipLaunchKernelGGL((crba_arm_kernel<8, 7, 8>), grid, block, 0, s, w->ops_f, q, n_tiles, H);
        }
        const int64_t done = (int64_t)n_tiles * WAVE;
        if (done == B) return launched();
        rc = launched();
        if (rc) return rc;
        drm_walk generic = *w;
        generic.shape &= ~DRM_WALK_ARM_CHAIN;
        return drm_crba(&generic, q + done * n, B - done, H + done * nn, scratch, stream);
    }
#endif
    {   // an arm that carries a hand (Panda with gripper, Jaco, iiwa7 + Allegro): full tiles through the straight-line kernel
        const int64_t done = launch_crba_arm_hand(w, q, B, H, s);
        if (done > 0) {
            rc = launched();
            if (rc || done == B) return rc;
            drm_walk generic = *w;
            generic.shape &= ~DRM_WALK_ARM_HAND;
            return drm_crba(&generic, q + done * n, B - done, H + done * nn, scratch, stream);
        }
    }
    {   // a small hand, a large launch: one wavefront per tile walks all the fingers
        const int64_t done = launch_crba_fingers_tree(w, q, B, H, s);
        if (done > 0) {
            rc = launched();
            if (rc || done == B) return rc;
            return drm_crba(w, q + done * n, B - done, H + done * nn, scratch, stream); // (< 64 rows: the kernels below)
        }
    }
    // (a misaligned call on a walk with a straight-line kernel, or its ragged tail: the loop kernel on at most MISALIGNED_TILES blocks)
    const bool fast_walk = ((w->shape & DRM_WALK_ARM_CHAIN) && w->capacity == 8 && n == 7) || crba_arm_hand_applies(w) ||
                           w->special[DRM_SPECIAL_CRBA] != nullptr;
    if (!segments_ok(w)) return fail(DRM_ERR_INVALID, "walk segments are inconsistent");
    if ((((uintptr_t)w->ops_f) & 15u) != 0) return fail(DRM_ERR_INVALID, "ops_f must be 16-byte aligned");
    const int64_t tiles = (B + WAVE - 1) / WAVE;
    if (tiles > 0x7fffffffLL) return fail(DRM_ERR_UNSUPPORTED, "batch too large");
    const uint32_t align = al16(q, AL_Q) | al16(H, AL_TAU);
    TreeArgs fingers;
    if (const size_t lds = crba_short_plan(w, fingers)) {
        if (stream_past_llc(B * nn * (int64_t)sizeof(float))) { // beyond the Infinity Cache: sc1 nt stores
            rc = ensure_lds_tree(crba_tree_kernel<true>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(crba_tree_kernel<true>, dim3((unsigned)tiles), dim3(WAVE * fingers.n_segments), lds, s, fingers, q, B, H, div_magic(n), align);
        } else {
            rc = ensure_lds_tree(crba_tree_kernel<false>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(crba_tree_kernel<false>, dim3((unsigned)tiles), dim3(WAVE * fingers.n_segments), lds, s, fingers, q, B, H, div_magic(n), align);
        }
        return launched();
    }
    CrbaRowsPlan p;
    rc = crba_rows_plan(w, p);
    if (rc) return rc;
    if (!scratch || ((uintptr_t)scratch & 15u))
        return fail(DRM_ERR_INVALID, "this robot's inertia matrix is assembled through scratch: pass drm_crba_scratch_floats() floats, 16-byte aligned");
    int64_t grid = tiles < p.resident ? tiles : (int64_t)p.resident;
    if (fast_walk && grid > MISALIGNED_TILES) grid = MISALIGNED_TILES;
    hipLaunchKernelGGL(crba_rows_kernel, dim3((unsigned)grid), dim3(WAVE * p.a.n_segments), p.lds, s, p.a, p.nt_max, q, B, (int)tiles, H, scratch,
                       div_magic(n), align);
    return launched();
}
