// drm_host_loops.hpp — the per-sample arithmetic of the kernels (drm_sample.hpp, drm_tree.hpp) run over HOST arrays, one sample at a
// time: the generic walks of every entry point of the C ABI.  Two users:
//   * csrc/drm_cpu.cpp — the HOST build of the C ABI (libdrm_cpu.so) behind `device="cpu"` models (the reference's default device,
//     robot_model.py:100-104): the same headers compiled by g++, samples spread over OpenMP threads;
//   * tests/host_emu/host_emu.cpp — test infrastructure: the same loops plus the shape-specialised variants, against the oracle.
// Not a fallback of the HIP path: a model on a HIP device never reaches this code (backend.load_library raises without
// libdrm_hip.so).
#pragma once
#include <stdint.h>

#include <cmath>
#include <vector>

#include "drm_sample.hpp"
#include "drm_tree.hpp"

namespace drm_host {
using namespace drm;


// the control words of a walk for the loop-structured walks (drm_tree.hpp)
struct Ctl {
    const int32_t *w0, *w1;
    explicit Ctl(const drm_walk *w) : w0(w->ops_i + DRM_OPI_W0 * w->capacity), w1(w->ops_i + DRM_OPI_W1 * w->capacity) {}
    void raw(int k, int &a, int &b) const { a = w0[k]; b = w1[k]; }
    int uniform(int r) const { return r; }
};

// pose of target t of sample b at pos + (b * sb + t * st) * 3 (quat: * 4): sample-major [B, T, .] by default (sb = T, st = 1),
// link-major [T, B_total, .] with sb = 1, st = B_total
inline void fk_loop(const drm_walk *w, const float *q, int64_t B, int T, float *pos, float *quat, int64_t sb = -1, int64_t st = 1) {
    const int n = w->n_dofs;
    if (sb < 0) sb = T;
    const Ctl ctl(w);
    for (int64_t b = 0; b < B; ++b) {
        PoseP slots[DRM_MAX_SLOTS];
        auto row = [&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; };
        auto qf = [&](int d) { return q[b * n + d]; };
        auto save = [&](int s, const PoseP &P) { slots[s] = P; };
        auto load = [&](int s, PoseP &P) { P = slots[s]; };
        auto emit = [&](int t, const float *p, const float *qt) {
            for (int i = 0; i < 3; ++i) pos[(b * sb + t * st) * 3 + i] = p[i];
            for (int i = 0; i < 4; ++i) quat[(b * sb + t * st) * 4 + i] = qt[i];
        };
        if ((w->shape & DRM_WALK_FK_FAN) && w->n_segments >= 2) {
            // like fk_tree_fan_kernel: every wavefront walks the shared part and its own run — here with save slots of its
            // own that start as NaN, so a run that leaned on a slot another run wrote would show
            for (int j = 0; j < w->n_segments; ++j) {
                for (auto &sl : slots)
                    for (int c = 0; c < 3; ++c) { sl.A[c] = f2_make(NAN, NAN); sl.B[c] = f2_make(NAN, NAN); }
                fk_tree_walk_ranges(w->prefix_end, w->seg_begin[j], w->seg_begin[j + 1], ctl, row, qf, save, load,
                                    [&](int k, int t, const float *p, const float *qt) {
                                        if (k < w->prefix_end && j != 0) return;
                                        emit(t, p, qt);
                                    });
            }
            continue;
        }
        fk_tree_walk(w->n_ops, ctl, row, qf, save, load, emit);
    }
}

inline void jac_loop(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin, float *ang) {
    const int n = w->n_dofs;
    const Ctl ctl(w);
    for (int64_t b = 0; b < B; ++b) {
        auto row = [&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; };
        auto qf = [&](int d) { return q[b * n + d]; };
        for (int i = 0; i < 3 * n; ++i) { lin[b * 3 * n + i] = 0.f; ang[b * 3 * n + i] = 0.f; }
        PoseP ee;
        // like the kernel: the walk leaves (z, p) of every moving joint in the two tiles, a second pass forms the columns
        std::vector<char> pris(n, 0), on(n, 0);
        fk_jacobian_tree_walk(w->n_ops, ctl, row, qf, ee, [&](int d, const float *z, const float *p, bool prismatic) {
            for (int r = 0; r < 3; ++r) { ang[(b * 3 + r) * n + d] = z[r]; lin[(b * 3 + r) * n + d] = p[r]; }
            pris[d] = prismatic; on[d] = 1;
        });
        Pose E;
        pose_from_pairs(ee, E);
        for (int d = 0; d < n; ++d) {
            if (!on[d]) continue;
            float z[3], p[3], c[3];
            for (int r = 0; r < 3; ++r) { z[r] = ang[(b * 3 + r) * n + d]; p[r] = lin[(b * 3 + r) * n + d]; }
            if (pris[d]) {
                for (int r = 0; r < 3; ++r) { lin[(b * 3 + r) * n + d] = z[r]; ang[(b * 3 + r) * n + d] = 0.f; }
            } else {
                const float dp[3] = {E.p[0] - p[0], E.p[1] - p[1], E.p[2] - p[2]};
                cross3(z, dp, c);
                for (int r = 0; r < 3; ++r) lin[(b * 3 + r) * n + d] = c[r];
            }
        }
        if (pos) for (int i = 0; i < 3; ++i) pos[b * 3 + i] = E.p[i];
        if (quat) {
            unpermute(w->target_perm, E.R);
            quat_xyzw(E.R, quat + b * 4);
        }
    }
}


struct ParkRec { Force f; float c, s, q; };

inline void rnea_loop(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau) {
    const int n = w->n_dofs;
    const Ctl ctl(w);
    std::vector<ParkRec> rec(w->capacity);
    for (int64_t b = 0; b < B; ++b) {
        for (int seg = 0; seg < w->n_segments; ++seg) {
            Motion ms[DRM_MAX_SLOTS];
            Force fs[DRM_MAX_SLOTS];
            for (auto &F : fs) for (int i = 0; i < 3; ++i) F.la[i] = f2_bcast(0.f);
            auto row = [&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; };
            auto qf = [&](int d, float &a, float &v, float &acc) {
                a = q[b * n + d]; v = qd[b * n + d]; acc = qdd ? qdd[b * n + d] : 0.f;
            };
            auto out = [&](int d, float v) { tau[b * n + d] = v; };
            auto park = [&](int k, const Force &F, float c, float s, float qq) { rec[k] = ParkRec{F, c, s, qq}; };
            auto unpark = [&](int k, Force &F) { F = rec[k].f; };
            auto trig = [&](int k, float &c, float &s, float &qq) { c = rec[k].c; s = rec[k].s; qq = rec[k].q; };
            auto msave = [&](int s, const Motion &M) { ms[s] = M; };
            auto mload = [&](int s, Motion &M) { M = ms[s]; };
            auto fadd = [&](int s, const Force &F) { for (int i = 0; i < 3; ++i) fs[s].la[i] += F.la[i]; };
            auto ftake = [&](int s, Force &F) {
                for (int i = 0; i < 3; ++i) { F.la[i] += fs[s].la[i]; fs[s].la[i] = f2_bcast(0.f); }
            };
            rnea_tree_walk(w->prefix_end, w->seg_begin[seg], w->seg_begin[seg + 1], ctl, row, flags, qf, out, park, unpark,
                           trig, msave, mload, fadd, ftake);
        }
    }
}

// reverse-mode FK: per-sample adjoint sweep, constant gradients summed over the batch in double
inline void fkb_t(const drm_walk *w, const float *q, int64_t B, int T, const float *gpos, const float *glin, const float *gang,
           uint64_t mask, float *gq, float *gops, const float *grot = nullptr) {
    const int n = w->n_dofs, CAP = w->capacity;
    std::vector<double> sum((size_t)CAP * 12, 0.0);
    std::vector<Pose> parked(CAP);
    for (int64_t b = 0; b < B; ++b) {
        Pose ps[DRM_MAX_SLOTS];
        Adjoint as[DRM_MAX_SLOTS] = {};
        auto park = [&](int k, const Pose &P) { parked[k] = P; };
        auto unpark = [&](int k, Pose &P) { P = parked[k]; };
        if (gq) for (int d = 0; d < n; ++d) gq[b * n + d] = 0.f;
        auto qf = [&](int d) { return q[b * n + d]; };
        auto gin = [&](int t, float *G) { if (gpos) for (int i = 0; i < 3; ++i) G[i] += gpos[(b * T + t) * 3 + i]; };
        auto jl = [&](int d, float *v) { for (int i = 0; i < 3; ++i) v[i] = glin[(b * 3 + i) * n + d]; };
        auto ja = [&](int d, float *v) { for (int i = 0; i < 3; ++i) v[i] = gang[(b * 3 + i) * n + d]; };
        auto psave = [&](int s, const Pose &P) { ps[s] = P; };
        auto pload = [&](int s, Pose &P) { P = ps[s]; };
        auto aadd = [&](int s, const Adjoint &A) {
            for (int i = 0; i < 3; ++i) as[s].G[i] += A.G[i];
            for (int i = 0; i < 9; ++i) as[s].M[i] += A.M[i];
        };
        auto atake = [&](int s, Adjoint &A) {
            for (int i = 0; i < 3; ++i) A.G[i] += as[s].G[i];
            for (int i = 0; i < 9; ++i) A.M[i] += as[s].M[i];
        };
        auto gqo = [&](int d, float v) { gq[b * n + d] = v; };
        auto pout = [&](int k, const float *dF, const float *dt) {
            for (int j = 0; j < 9; ++j) sum[k * 12 + j] += dF[j];
            for (int j = 0; j < 3; ++j) sum[k * 12 + 9 + j] += dt[j];
        };
        const int32_t *ctl = w->ops_i + DRM_OPI_CTRL * CAP;
        auto rot = [&](int t, float *Rb) {
            if (!grot) return false;
            for (int i = 0; i < 9; ++i) Rb[i] = grot[(b * T + t) * 9 + i];
            return true;
        };
        if (glin)
            fk_backward_walk<true>(w->ops_f, ctl, w->n_ops, mask, gq != nullptr, qf, gin, psave, pload, aadd, atake, gqo, pout,
                                   park, unpark, jl, ja, rot);
        else
            fk_backward_walk<false>(w->ops_f, ctl, w->n_ops, mask, gq != nullptr, qf, gin, psave, pload, aadd, atake, gqo, pout,
                                    park, unpark, NoJacobianGrad(), NoJacobianGrad(), rot);
    }
    if (gops)
        for (int k = 0; k < CAP; ++k) {
            for (int j = 0; j < DRM_OPF_STRIDE; ++j) gops[k * DRM_OPF_STRIDE + j] = 0.f;
            for (int j = 0; j < 12; ++j)
                gops[k * DRM_OPF_STRIDE + (j < 9 ? DRM_OPF_FIJ(j / 3, j % 3) : DRM_OPF_TI(j - 9))] = (float)sum[k * 12 + j];
        }
}

// reverse-mode RNEA: per-sample adjoint sweeps, constant gradients summed over the batch in double
inline void rneab_t(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, const float *gtau,
             uint64_t mask, float *gq, float *gqd, float *gqdd, float *gops) {
    const int n = w->n_dofs, CAP = w->capacity;
    std::vector<double> sum((size_t)CAP * DRM_OPF_STRIDE, 0.0);
    std::vector<float> recv((size_t)CAP * 26);
    float (*rec)[26] = reinterpret_cast<float (*)[26]>(recv.data());
    for (int64_t b = 0; b < B; ++b) {
        float slots[DRM_MAX_SLOTS][36];
        for (auto &s : slots) for (float &x : s) x = 0.f;
        if (gq) for (int d = 0; d < n; ++d) { gq[b * n + d] = 0.f; gqd[b * n + d] = 0.f; gqdd[b * n + d] = 0.f; }
        auto qf = [&](int d, float &a, float &v, float &acc) { a = q[b * n + d]; v = qd[b * n + d]; acc = qdd ? qdd[b * n + d] : 0.f; };
        auto gt = [&](int d) { return gtau[b * n + d]; };
        auto park = [&](int k, int off, const float *v, int cnt) { for (int i = 0; i < cnt; ++i) rec[k][off + i] = v[i]; };
        auto unpark = [&](int k, int off, float *v, int cnt) { for (int i = 0; i < cnt; ++i) v[i] = rec[k][off + i]; };
        auto sput = [&](int s, int off, const float *v, int cnt) { for (int i = 0; i < cnt; ++i) slots[s][off + i] = v[i]; };
        auto sget = [&](int s, int off, float *v, int cnt) { for (int i = 0; i < cnt; ++i) v[i] = slots[s][off + i]; };
        auto sadd = [&](int s, int off, const float *v, int cnt) { for (int i = 0; i < cnt; ++i) slots[s][off + i] += v[i]; };
        auto stake = [&](int s, int off, float *v, int cnt) { for (int i = 0; i < cnt; ++i) { v[i] = slots[s][off + i]; slots[s][off + i] = 0.f; } };
        auto gout = [&](int d, float a, float v, float acc) { gq[b * n + d] = a; gqd[b * n + d] = v; gqdd[b * n + d] = acc; };
        auto pout = [&](int k, const float *g) { for (int j = 0; j < DRM_OPF_STRIDE; ++j) sum[k * DRM_OPF_STRIDE + j] += g[j]; };
        // segment by segment, as the fanned-out kernel does it (one wavefront each there); a walk with learnable prefix ops
        // in one go
        const uint64_t prefix_mask = w->prefix_end >= 64 ? ~0ull : ((1ull << w->prefix_end) - 1ull);
        if (w->n_segments > 1 && !(mask & prefix_mask)) {
            for (int seg = 0; seg < w->n_segments; ++seg) {
                for (auto &s : slots) for (float &x : s) x = 0.f;
                if (!rnea_backward_walk_short<6>(w->ops_f, w->ops_i + DRM_OPI_CTRL * CAP, w->prefix_end, w->seg_begin[seg],
                                                 w->seg_begin[seg + 1], flags, mask, gq != nullptr, qf, gt, sput, sget, gout, pout))
                    rnea_backward_walk(w->ops_f, w->ops_i + DRM_OPI_CTRL * CAP, w->prefix_end, w->seg_begin[seg], w->seg_begin[seg + 1],
                                       flags, mask, gq != nullptr, qf, gt, park, unpark, sput, sget, sadd, stake, gout, pout);
            }
        } else {
            rnea_backward_walk(w->ops_f, w->ops_i + DRM_OPI_CTRL * CAP, 0, 0, w->n_ops, flags, mask, gq != nullptr, qf, gt, park, unpark,
                               sput, sget, sadd, stake, gout, pout);
        }
    }
    if (gops) for (int i = 0; i < CAP * DRM_OPF_STRIDE; ++i) gops[i] = (float)sum[i];
}


struct TrigRec { float c, s, q; };

inline void crba_loop(const drm_walk *w, const float *q, int64_t B, float *H) {
    const int n = w->n_dofs;
    const Ctl ctl(w);
    std::vector<TrigRec> tr(w->capacity);
    bool long_segments = false;
    for (int s = 0; s < w->n_segments; ++s) long_segments = long_segments || w->seg_begin[s + 1] - w->seg_begin[s] > 6;
    for (int64_t b = 0; b < B; ++b) {
        for (int i = 0; i < n * n; ++i) H[b * n * n + i] = 0.f;
        for (int seg = 0; seg < w->n_segments; ++seg) {
            Inertia is[DRM_MAX_SLOTS];
            for (auto &a : is) inertia_zero(a);
            const int a0 = w->seg_begin[seg], b0 = w->seg_begin[seg + 1];
            auto row = [&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; };
            auto qf = [&](int d) { return q[b * n + d]; };
            for (int k = a0; k < b0; ++k) { // cos / sin / value of every op's joint
                int w0, w1;
                ctl_words(ctl, k, w0, w1);
                const OpCtl ct = decode_ctl(w0, w1);
                TrigRec t{1.f, 0.f, 0.f};
                if (ct.dof >= 0) {
                    t.q = qf(ct.dof);
                    if (!ct.prismatic) sincos_one(t.q, t.s, t.c);
                }
                tr[k] = t;
            }
            auto trig = [&](int k, float &c, float &s, float &qq) { c = tr[k].c; s = tr[k].s; qq = tr[k].q; };
            auto iadd = [&](int s, const Inertia &a) { inertia_add(is[s], a); };
            auto itake = [&](int s, Inertia &a) { inertia_add(a, is[s]); inertia_zero(is[s]); };
            auto hout = [&](int di, int dj, float v) { H[(b * n + di) * n + dj] = v; };
            // as the kernels: robots with a segment of more than 6 ops take the walk that moves all the forces of a sub-tree up
            // together; otherwise the unrolled short-serial form where it applies, the loop where it does not
            if (long_segments) {
                std::vector<int> t_lo(b0 - a0 + 1), t_end(b0 - a0), t_dof(n);
                std::vector<Force> fs(n);
                crba_set_tables(a0, b0, ctl, [&](int k, int v) { t_lo[k - a0] = v; }, [&](int k, int v) { t_end[k - a0] = v; },
                                [&](int k) { return t_end[k - a0]; }, [&](int m, int d) { t_dof[m] = d; });
                crba_set_walk(a0, b0, ctl, row, trig, [&](int k) { return t_lo[k - a0]; }, [&](int k) { return t_lo[t_end[k - a0] + 1 - a0]; },
                              [&](int m) { return t_dof[m]; }, [&](int m, Force &F) { F = fs[m]; }, [&](int m, const Force &F) { fs[m] = F; },
                              iadd, itake, hout);
            } else if (!crba_tree_walk_short<6>(a0, b0, ctl, row, qf, hout)) {
                crba_tree_walk(a0, b0, ctl, row, trig, iadd, itake, hout);
            }
        }
    }
}

// forward dynamics: per segment, H (packed lower triangle of the segment's block) + bias torques + L^T D L solve
// ... by the articulated-body walk, segment by segment (what the kernel runs when a segment is longer than 6 ops)
inline void aba_loop(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int flags, float *qdd) {
    const int n = w->n_dofs;
    const Ctl ctl(w);
    std::vector<Motion> vel(w->capacity);
    std::vector<float> recs((size_t)w->capacity * 8);
    for (int64_t b = 0; b < B; ++b)
        for (int seg = 0; seg < w->n_segments; ++seg) {
            Motion ms[DRM_MAX_SLOTS];
            ArtBody bs[DRM_MAX_SLOTS];
            for (auto &a : bs) art_zero(a);
            aba_tree_walk(
                w->n_segments > 1 ? w->prefix_end : 0, w->seg_begin[seg], w->seg_begin[seg + 1], ctl,
                [&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, flags,
                [&](int d, float &x, float &v) { x = q[b * n + d]; v = qd[b * n + d]; }, [&](int d) { return f[b * n + d]; },
                [&](int d, float v) { qdd[b * n + d] = v; },
                [&](int k, const Motion &M) { vel[k] = M; },
                [&](int k, Motion &M) {
                    for (int i = 0; i < 3; ++i) { M.wa[i] = f2_make(vel[k].wa[i][0], 0.f); M.va[i] = f2_make(vel[k].va[i][0], 0.f); }
                },
                [&](int k, const float *r) { for (int i = 0; i < 8; ++i) recs[k * 8 + i] = r[i]; },
                [&](int k, float *r) { for (int i = 0; i < 8; ++i) r[i] = recs[k * 8 + i]; },
                [&](int sl, const Motion &M) { ms[sl] = M; }, [&](int sl, Motion &M) { M = ms[sl]; },
                [&](int sl, const ArtBody &a) { art_add(bs[sl], a); },
                [&](int sl, ArtBody &a) { art_add(a, bs[sl]); art_zero(bs[sl]); });
        }
}

inline bool short_segments(const drm_walk *w) {
    for (int s = 0; s < w->n_segments; ++s)
        if (w->seg_begin[s + 1] - w->seg_begin[s] > 6) return false;
    return true;
}

inline void fd_loop(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int flags, float *qdd) {
    const int n = w->n_dofs;
    if (!short_segments(w)) return aba_loop(w, q, qd, f, B, flags, qdd);
    std::vector<float> H((size_t)n * n), T((size_t)n * (n + 1) / 2), x(n);
    for (int64_t b = 0; b < B; ++b) {
        crba_loop(w, q + b * n, 1, H.data());
        rnea_loop(w, q + b * n, qd + b * n, nullptr, 1, flags, x.data());
        for (int seg = 0; seg < w->n_segments; ++seg) {
            const int lo = w->seg_dof_lo[seg], cnt = w->seg_dof_cnt[seg];
            for (int i = 0; i < cnt; ++i)
                for (int j = 0; j <= i; ++j) T[tri_index(i, j)] = H[(lo + i) * n + lo + j];
            float *r = qdd + b * n + lo;
            for (int d = 0; d < cnt; ++d) r[d] = f[b * n + lo + d] - x[lo + d];
            ltdl_solve(cnt, T.data(), r);
        }
    }
}

} // namespace drm_host
