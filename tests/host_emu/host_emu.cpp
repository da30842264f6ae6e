// TEST INFRASTRUCTURE ONLY — never loaded by the product package.
//
// Compiles the per-sample arithmetic of the HIP kernels (csrc/drm_sample.hpp)
// with g++ and runs it one sample at a time over HOST arrays, so that the
// kernel arithmetic and the walk encoding can be checked against the oracle
// in the GPU-less build container.  The tile I/O and launch code of
// drm_kernels.hip are NOT covered here; the `-m gpu` tests cover the real thing.
#include <stdint.h>

#include <cmath>
#include <vector>

#include "../../differentiable-robot-model_amd/csrc/drm_sample.hpp"
#include "../../differentiable-robot-model_amd/csrc/drm_tree.hpp"

using namespace drm;

namespace {

// the control words of a walk for the loop-structured walks (drm_tree.hpp)
struct Ctl {
    const int32_t *w0, *w1;
    explicit Ctl(const drm_walk *w) : w0(w->ops_i + DRM_OPI_W0 * w->capacity), w1(w->ops_i + DRM_OPI_W1 * w->capacity) {}
    void raw(int k, int &a, int &b) const { a = w0[k]; b = w1[k]; }
    int uniform(int r) const { return r; }
};

void fk_loop(const drm_walk *w, const float *q, int64_t B, int T, float *pos, float *quat) {
    const int n = w->n_dofs;
    const Ctl ctl(w);
    for (int64_t b = 0; b < B; ++b) {
        PoseP slots[DRM_MAX_SLOTS];
        auto row = [&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; };
        auto qf = [&](int d) { return q[b * n + d]; };
        auto save = [&](int s, const PoseP &P) { slots[s] = P; };
        auto load = [&](int s, PoseP &P) { P = slots[s]; };
        auto emit = [&](int t, const float *p, const float *qt) {
            for (int i = 0; i < 3; ++i) pos[(b * T + t) * 3 + i] = p[i];
            for (int i = 0; i < 4; ++i) quat[(b * T + t) * 4 + i] = qt[i];
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

void jac_loop(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin, float *ang) {
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

// packed-FP32 chain arithmetic of the arm kernel (fk_jacobian_arm_kernel<8, 7>)
void jac_arm_8_7(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin, float *ang) {
    constexpr int CAP = 8, NJ = 7;
    for (int64_t b = 0; b < B; ++b) {
        float qv[NJ];
        for (int d = 0; d < NJ; ++d) qv[d] = q[b * NJ + d];
        PoseP ee;
        f2 Bk[NJ][3];
        fk_chain_pairs<CAP, NJ>([&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, qv, ee, Bk, [] {});
        Pose E;
        pose_from_pairs(ee, E);
        for (int i = 0; i < 3; ++i) pos[b * 3 + i] = E.p[i];
        unpermute(w->target_perm, E.R);
        quat_xyzw(E.R, quat + b * 4);
        for (int k = 0; k < NJ; ++k) {
            const float z[3] = {Bk[k][0][0], Bk[k][1][0], Bk[k][2][0]};
            const float dp[3] = {E.p[0] - Bk[k][0][1], E.p[1] - Bk[k][1][1], E.p[2] - Bk[k][2][1]};
            float c[3];
            cross3(z, dp, c);
            for (int r = 0; r < 3; ++r) { lin[(b * 3 + r) * NJ + k] = c[r]; ang[(b * 3 + r) * NJ + k] = z[r]; }
        }
    }
}

void rnea_arm_8_7(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau) {
    constexpr int CAP = 8, NJ = 7;
    for (int64_t b = 0; b < B; ++b) {
        float qv[NJ], qdv[NJ], qddv[NJ], tv[NJ];
        for (int d = 0; d < NJ; ++d) { qv[d] = q[b * NJ + d]; qdv[d] = qd[b * NJ + d]; qddv[d] = qdd ? qdd[b * NJ + d] : 0.f; }
        Force park[CAP];
        rnea_chain<CAP, NJ>([&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, flags & DRM_RNEA_GRAVITY,
                            flags & DRM_RNEA_DAMPING, qv, qdv, qddv, tv, [&](int k, const Force &F) { park[k] = F; },
                            [&](int k, Force &F) { F = park[k]; });
        for (int d = 0; d < NJ; ++d) tau[b * NJ + d] = tv[d];
    }
}

// the arithmetic of rnea_arm2_kernel: samples (b, b + 1) share a lane as (A, B); LINKS = 8 (whole table) or 7 (tail folded)
template <int LINKS>
void rnea_arm2_8_7(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau) {
    constexpr int NJ = 7;
    for (int64_t b = 0; b < B; b += 2) {
        const int64_t b1 = b + 1 < B ? b + 1 : b;
        f2 qv[NJ], qdv[NJ], qddv[NJ], tv[NJ], cs[NJ], sn[NJ];
        for (int d = 0; d < NJ; ++d) {
            qv[d] = f2_make(q[b * NJ + d], q[b1 * NJ + d]);
            qdv[d] = f2_make(qd[b * NJ + d], qd[b1 * NJ + d]);
            qddv[d] = qdd ? f2_make(qdd[b * NJ + d], qdd[b1 * NJ + d]) : f2_bcast(0.f);
        }
        chain_trig2<NJ>(qv, cs, sn);
        Force2 park[LINKS];
        rnea_chain2_trig<LINKS, NJ>([&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, flags & DRM_RNEA_GRAVITY,
                                    flags & DRM_RNEA_DAMPING, cs, sn, qdv, qddv, tv, [&](int k, const Force2 &F) { park[k] = F; },
                                    [&](int k, Force2 &F) { F = park[k]; });
        for (int d = 0; d < NJ; ++d) { tau[b * NJ + d] = tv[d][0]; tau[b1 * NJ + d] = tv[d][1]; }
    }
}

// the arithmetic of rnea_fingers2_kernel<L>: finger w = ops w L .. w L + L - 1 = DoF columns of the same numbers; two samples share
// a lane; all body forces kept (nothing parked)
template <int L>
void rnea_fingers2_emu(const drm_walk *w, int K, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau) {
    const int n = w->n_dofs;
    for (int64_t b = 0; b < B; b += 2) {
        const int64_t b1 = b + 1 < B ? b + 1 : b;
        for (int f = 0; f < K; ++f) {
            f2 qv[L], qdv[L], qddv[L], tv[L], cs[L], sn[L];
            for (int d = 0; d < L; ++d) {
                const int c = f * L + d;
                qv[d] = f2_make(q[b * n + c], q[b1 * n + c]);
                qdv[d] = f2_make(qd[b * n + c], qd[b1 * n + c]);
                qddv[d] = qdd ? f2_make(qdd[b * n + c], qdd[b1 * n + c]) : f2_bcast(0.f);
            }
            chain_trig2<L>(qv, cs, sn);
            rnea_chain2_trig<L, L, L - 1>([&](int k) { return w->ops_f + (f * L + k) * DRM_OPF_STRIDE; }, flags & DRM_RNEA_GRAVITY,
                                          flags & DRM_RNEA_DAMPING, cs, sn, qdv, qddv, tv, [](int, const Force2 &) {}, [](int, Force2 &) {});
            for (int d = 0; d < L; ++d) { tau[b * n + f * L + d] = tv[d][0]; tau[b1 * n + f * L + d] = tv[d][1]; }
        }
    }
}

// the arithmetic of rnea_arm_hand_kernel<P, L> (one sample per lane; K sub-chains of L ops behind a prefix of P ops)
template <int P, int L>
void rnea_arm_hand_emu(const drm_walk *w, int K, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau) {
    const int n = w->n_dofs;
    const int32_t *w0 = w->ops_i + DRM_OPI_W0 * w->capacity;
    auto dof_of = [&](int k) { return (w0[k] & 0xff) - 1; };
    for (int64_t b = 0; b < B; ++b) {
        auto state = [&](int d, float &a, float &v, float &acc) {
            a = d < 0 ? 0.f : q[b * n + d]; v = d < 0 ? 0.f : qd[b * n + d]; acc = (d < 0 || !qdd) ? 0.f : qdd[b * n + d];
        };
        float qv[P], qdv[P], qddv[P], cs[P], sn[P], tp[P];
        for (int k = 0; k < P; ++k) state(dof_of(k), qv[k], qdv[k], qddv[k]);
        chain_trig<P>(qv, cs, sn);
        Force park[P];
        auto kind = [&](int op) { const int x = w0[op]; return ((x & 0xff) ? 1 : 0) | (((x >> 26) & 1) << 1); };
        rnea_arm_hand<P, L>([&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, kind, K, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING,
                            qv, cs, sn, qdv, qddv, [&](int j, int i, float &a, float &v, float &acc) { state(dof_of(P + j * L + i), a, v, acc); },
                            tp, [&](int j, int i, float t) { const int d = dof_of(P + j * L + i); if (d >= 0) tau[b * n + d] = t; },
                            [&](int k, const Force &F) { park[k] = F; }, [&](int k, Force &F) { F = park[k]; });
        for (int k = 0; k < P; ++k) if (dof_of(k) >= 0) tau[b * n + dof_of(k)] = tp[k];
    }
}

// the arithmetic of forward_dynamics_arm_hand_kernel<P, L>
template <int P, int L>
void fd_arm_hand_emu(const drm_walk *w, int K, const float *q, const float *qd, const float *f, int64_t B, int flags, float *qdd) {
    const int n = w->n_dofs;
    const int32_t *w0 = w->ops_i + DRM_OPI_W0 * w->capacity;
    auto dof_of = [&](int k) { return (w0[k] & 0xff) - 1; };
    auto kind = [&](int op) { const int x = w0[op]; return ((x & 0xff) ? 1 : 0) | (((x >> 26) & 1) << 1); };
    for (int64_t b = 0; b < B; ++b) {
        auto state = [&](int d, float &a, float &v, float &t) {
            a = d < 0 ? 0.f : q[b * n + d]; v = d < 0 ? 0.f : qd[b * n + d]; t = d < 0 ? 0.f : f[b * n + d];
        };
        float qv[P], qdv[P], fv[P], cs[P], sn[P], out[P], slot[P][8];
        for (int k = 0; k < P; ++k) state(dof_of(k), qv[k], qdv[k], fv[k]);
        chain_trig<P>(qv, cs, sn);
        aba_arm_hand<P, L>([&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, kind, K, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING,
                           qv, cs, sn, qdv, fv, [&](int j, int i, float &a, float &v, float &t) { state(dof_of(P + j * L + i), a, v, t); }, out,
                           [&](int j, int i, float a) { const int d = dof_of(P + j * L + i); if (d >= 0) qdd[b * n + d] = a; },
                           [&](int k, const float *s8) { for (int c = 0; c < 8; ++c) slot[k][c] = s8[c]; },
                           [&](int k, float *s8) { for (int c = 0; c < 8; ++c) s8[c] = slot[k][c]; });
        for (int k = 0; k < P; ++k) if (dof_of(k) >= 0) qdd[b * n + dof_of(k)] = out[k];
    }
}

// the arithmetic of crba_arm_hand_kernel<P, L>
template <int P, int L>
void crba_arm_hand_emu(const drm_walk *w, int K, const float *q, int64_t B, float *H) {
    const int n = w->n_dofs;
    const int32_t *w0 = w->ops_i + DRM_OPI_W0 * w->capacity;
    auto dof_of = [&](int k) { return (w0[k] & 0xff) - 1; };
    auto kind = [&](int op) { const int x = w0[op]; return ((x & 0xff) ? 1 : 0) | (((x >> 26) & 1) << 1); };
    for (int64_t b = 0; b < B; ++b) {
        for (int i = 0; i < n * n; ++i) H[b * n * n + i] = 0.f;
        float qv[P], cs[P], sn[P];
        for (int k = 0; k < P; ++k) qv[k] = dof_of(k) < 0 ? 0.f : q[b * n + dof_of(k)];
        chain_trig<P>(qv, cs, sn);
        auto row = [&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; };
        auto hq = [&](int j, int i) { const int d = dof_of(P + j * L + i); return d < 0 ? 0.f : q[b * n + d]; };
        auto hout = [&](int oa, int ob, float v) {
            const int di = dof_of(oa), dj = dof_of(ob);
            H[(b * n + di) * n + dj] = v; H[(b * n + dj) * n + di] = v;
        };
        // as the kernel runs it: a wavefront per sub-chain, the palm's composites summed in sub-chain order, every wavefront
        // sweeps the prefix with its sub-chain's forces and its share of the prefix's columns
        Inertia palm, part[4];
        Force Fp[4][L];
        inertia_zero(palm);
        for (int j = 0; j < K; ++j) {
            crba_arm_hand_sub<P, L>(row, kind, j, [&](int i) { return hq(j, i); }, hout, Fp[j], part[j]);
            inertia_add(palm, part[j]);
        }
        for (int j = 0; j < K; ++j) crba_arm_hand_prefix<P, L>(row, kind, j, qv, cs, sn, palm, Fp[j], j, K, hout);
    }
}

struct ParkRec { Force f; float c, s, q; };

void rnea_loop(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, float *tau) {
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
void fkb_t(const drm_walk *w, const float *q, int64_t B, int T, const float *gpos, const float *glin, const float *gang,
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
void rneab_t(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int flags, const float *gtau,
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

// the arithmetic of rnea_backward_arm_hand_kernel<P, L>
template <int P, int L>
void rneab_arm_hand_emu(const drm_walk *w, int K, const float *q, const float *qd, const float *qdd, int64_t B, int flags,
                        const float *gtau, uint64_t mask, float *gq, float *gqd, float *gqdd, float *gops) {
    const int n = w->n_dofs, CAP = w->capacity;
    const int32_t *w0 = w->ops_i + DRM_OPI_W0 * w->capacity;
    auto dof_of = [&](int k) { return (w0[k] & 0xff) - 1; };
    auto kind = [&](int op) { const int x = w0[op]; return ((x & 0xff) ? 1 : 0) | (((x >> 26) & 1) << 1); };
    std::vector<double> sum((size_t)CAP * DRM_OPF_STRIDE, 0.0);
    for (int64_t b = 0; b < B; ++b) {
        auto at = [&](const float *a, int d) { return (d < 0 || !a) ? 0.f : a[b * n + d]; };
        float palm[36];
        rnea_backward_arm_hand<P, L>(
            [&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, kind, K, flags & DRM_RNEA_GRAVITY, flags & DRM_RNEA_DAMPING, mask,
            gq != nullptr,
            [&](int k, float &a, float &v, float &acc, float &g) {
                const int d = dof_of(k);
                a = at(q, d); v = at(qd, d); acc = at(qdd, d); g = at(gtau, d);
            },
            [&](int j, int i, float &a, float &v, float &acc, float &g) {
                const int d = dof_of(P + j * L + i);
                a = at(q, d); v = at(qd, d); acc = at(qdd, d); g = at(gtau, d);
            },
            [&](int op, float a, float v, float acc) { const int d = dof_of(op); gq[b * n + d] = a; gqd[b * n + d] = v; gqdd[b * n + d] = acc; },
            [&](int k, const float *g) { for (int j = 0; j < DRM_OPF_STRIDE; ++j) sum[k * DRM_OPF_STRIDE + j] += g[j]; },
            [&](int i, float x) { palm[i] = x; }, [&](int i) { return palm[i]; });
    }
    if (gops) for (int i = 0; i < CAP * DRM_OPF_STRIDE; ++i) gops[i] = (float)sum[i];
}

struct TrigRec { float c, s, q; };

void crba_loop(const drm_walk *w, const float *q, int64_t B, float *H) {
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
void aba_loop(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int flags, float *qdd) {
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

static bool short_segments(const drm_walk *w) {
    for (int s = 0; s < w->n_segments; ++s)
        if (w->seg_begin[s + 1] - w->seg_begin[s] > 6) return false;
    return true;
}

void fd_loop(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int flags, float *qdd) {
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

} // namespace

extern "C" {
int emu_fk(const drm_walk *w, const float *q, int64_t B, int32_t T, float *pos, float *quat) {
    fk_loop(w, q, B, T, pos, quat);
    return 0;
}
int emu_fk_jacobian(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin, float *ang) {
    jac_loop(w, q, B, pos, quat, lin, ang);
    return 0;
}
int emu_fk_jacobian_arm(const drm_walk *w, const float *q, int64_t B, float *pos, float *quat, float *lin, float *ang) {
    if (!(w->shape & DRM_WALK_ARM_CHAIN) || w->capacity != 8 || w->n_dofs != 7) return -2;
    jac_arm_8_7(w, q, B, pos, quat, lin, ang);
    return 0;
}
int emu_rnea_arm(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags, float *tau) {
    if (!(w->shape & DRM_WALK_ARM_CHAIN) || w->capacity != 8 || w->n_dofs != 7) return -2;
    rnea_arm_8_7(w, q, qd, qdd, B, flags, tau);
    return 0;
}
int emu_rnea_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags, float *tau) {
    if (!(w->shape & DRM_WALK_ARM_HAND)) return -2;
    const int P = DRM_WALK_AH_P(w->shape), K = DRM_WALK_AH_K(w->shape), L = DRM_WALK_AH_L(w->shape);
    if (P + K * L != w->n_ops) return -1;
    if (P == 7 && L == 1) { rnea_arm_hand_emu<7, 1>(w, K, q, qd, qdd, B, flags, tau); return 0; }
    if (P == 6 && L == 2) { rnea_arm_hand_emu<6, 2>(w, K, q, qd, qdd, B, flags, tau); return 0; }
    if (P == 7 && L == 4) { rnea_arm_hand_emu<7, 4>(w, K, q, qd, qdd, B, flags, tau); return 0; }
    if (P == 9 && L == 1) rnea_arm_hand_emu<9, 1>(w, K, q, qd, qdd, B, flags, tau);
    else if (P == 7 && L == 2) rnea_arm_hand_emu<7, 2>(w, K, q, qd, qdd, B, flags, tau);
    else if (P == 8 && L == 4) rnea_arm_hand_emu<8, 4>(w, K, q, qd, qdd, B, flags, tau);
    else return -2;
    return 0;
}
int emu_forward_dynamics_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int32_t flags, float *qdd) {
    if (!(w->shape & DRM_WALK_ARM_HAND)) return -2;
    const int P = DRM_WALK_AH_P(w->shape), K = DRM_WALK_AH_K(w->shape), L = DRM_WALK_AH_L(w->shape);
    if (P + K * L != w->n_ops) return -1;
    if (P == 7 && L == 1) { fd_arm_hand_emu<7, 1>(w, K, q, qd, f, B, flags, qdd); return 0; }
    if (P == 6 && L == 2) { fd_arm_hand_emu<6, 2>(w, K, q, qd, f, B, flags, qdd); return 0; }
    if (P == 7 && L == 4) { fd_arm_hand_emu<7, 4>(w, K, q, qd, f, B, flags, qdd); return 0; }
    if (P == 9 && L == 1) fd_arm_hand_emu<9, 1>(w, K, q, qd, f, B, flags, qdd);
    else if (P == 7 && L == 2) fd_arm_hand_emu<7, 2>(w, K, q, qd, f, B, flags, qdd);
    else if (P == 8 && L == 4) fd_arm_hand_emu<8, 4>(w, K, q, qd, f, B, flags, qdd);
    else return -2;
    return 0;
}
int emu_crba_arm_hand(const drm_walk *w, const float *q, int64_t B, float *H) {
    if (!(w->shape & DRM_WALK_ARM_HAND)) return -2;
    const int P = DRM_WALK_AH_P(w->shape), K = DRM_WALK_AH_K(w->shape), L = DRM_WALK_AH_L(w->shape);
    if (P + K * L != w->n_ops || K < 2 || K > 4) return -1;
    if (P == 7 && L == 1) { crba_arm_hand_emu<7, 1>(w, K, q, B, H); return 0; }
    if (P == 6 && L == 2) { crba_arm_hand_emu<6, 2>(w, K, q, B, H); return 0; }
    if (P == 7 && L == 4) { crba_arm_hand_emu<7, 4>(w, K, q, B, H); return 0; }
    if (P == 9 && L == 1) crba_arm_hand_emu<9, 1>(w, K, q, B, H);
    else if (P == 7 && L == 2) crba_arm_hand_emu<7, 2>(w, K, q, B, H);
    else if (P == 8 && L == 4) crba_arm_hand_emu<8, 4>(w, K, q, B, H);
    else return -2;
    return 0;
}
int emu_rnea_fingers(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags, float *tau) {
    if (!(w->shape & DRM_WALK_FINGERS)) return -2;
    const int K = DRM_WALK_AH_K(w->shape), L = DRM_WALK_AH_L(w->shape);
    if (K * L != w->n_ops || w->n_ops != w->n_dofs) return -1;
    if (L == 2) rnea_fingers2_emu<2>(w, K, q, qd, qdd, B, flags, tau);
    else if (L == 3) rnea_fingers2_emu<3>(w, K, q, qd, qdd, B, flags, tau);
    else if (L == 4) rnea_fingers2_emu<4>(w, K, q, qd, qdd, B, flags, tau);
    else return -2;
    return 0;
}
int emu_rnea_arm2(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags, float *tau) {
    if (!(w->shape & DRM_WALK_ARM_CHAIN) || w->capacity != 8 || w->n_dofs != 7) return -2;
    if (w->n_ops == 7) rnea_arm2_8_7<7>(w, q, qd, qdd, B, flags, tau);
    else rnea_arm2_8_7<8>(w, q, qd, qdd, B, flags, tau);
    return 0;
}
int emu_rnea_backward(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags,
                      const float *gtau, uint64_t mask, float *gq, float *gqd, float *gqdd, float *gops) {
    rneab_t(w, q, qd, qdd, B, flags, gtau, mask, gq, gqd, gqdd, gops);
    return 0;
}
int emu_forward_dynamics(const drm_walk *w, const float *q, const float *qd, const float *f, int64_t B, int32_t flags,
                         float *qdd) {
    fd_loop(w, q, qd, f, B, flags, qdd);
    return 0;
}
int emu_link_rows(const float *params, int32_t n, float *rows) {
    for (int i = 0; i < n; ++i) link_row(params + i * LINK_PARAM_FLOATS, rows + i * DRM_OPF_STRIDE);
    return 0;
}
int emu_link_rows_backward(const float *params, const float *grad_rows, int32_t n, float *grad_params) {
    for (int i = 0; i < n; ++i)
        link_row_backward(params + i * LINK_PARAM_FLOATS, grad_rows + i * DRM_OPF_STRIDE, grad_params + i * LINK_PARAM_FLOATS);
    return 0;
}
int emu_rnea_backward_arm_hand(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags,
                               const float *gtau, uint64_t mask, float *gq, float *gqd, float *gqdd, float *gops) {
    if (!(w->shape & DRM_WALK_ARM_HAND)) return -2;
    const int P = DRM_WALK_AH_P(w->shape), K = DRM_WALK_AH_K(w->shape), L = DRM_WALK_AH_L(w->shape);
    if (P + K * L != w->n_ops) return -1;
#define X(p, l) if (P == p && L == l) { rneab_arm_hand_emu<p, l>(w, K, q, qd, qdd, B, flags, gtau, mask, gq, gqd, gqdd, gops); return 0; }
    X(7, 1) X(6, 2) X(7, 4) X(9, 1) X(7, 2) X(8, 4)
#undef X
    return -2;
}
int emu_rnea_backward_arm(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags,
                          const float *gtau, uint64_t mask, float *gq, float *gqd, float *gqdd, float *gops) {
    if (!(w->shape & DRM_WALK_ARM_CHAIN) || w->capacity != 8 || w->n_dofs != 7) return -2;
    constexpr int CAP = 8, NJ = 7;
    static thread_local double sum[CAP * DRM_OPF_STRIDE];
    for (int i = 0; i < CAP * DRM_OPF_STRIDE; ++i) sum[i] = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        float qv[NJ], qdv[NJ], qddv[NJ], gt[NJ];
        for (int d = 0; d < NJ; ++d) {
            qv[d] = q[b * NJ + d]; qdv[d] = qd[b * NJ + d]; qddv[d] = qdd ? qdd[b * NJ + d] : 0.f; gt[d] = gtau[b * NJ + d];
        }
        rnea_backward_chain<CAP, NJ>([&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, flags & DRM_RNEA_GRAVITY,
                                     flags & DRM_RNEA_DAMPING, mask, gq != nullptr, qv, qdv, qddv, gt,
                                     [&](int d, float a, float v, float c) { gq[b * NJ + d] = a; gqd[b * NJ + d] = v; gqdd[b * NJ + d] = c; },
                                     [&](int k, const float *g) { for (int j = 0; j < DRM_OPF_STRIDE; ++j) sum[k * DRM_OPF_STRIDE + j] += g[j]; });
    }
    if (gops) for (int i = 0; i < CAP * DRM_OPF_STRIDE; ++i) gops[i] = (float)sum[i];
    return 0;
}
int emu_fk_backward_arm(const drm_walk *w, const float *q, int64_t B, const float *gpos, uint64_t mask, float *gq, float *gops) {
    if (!(w->shape & DRM_WALK_ARM_CHAIN) || w->capacity != 8 || w->n_dofs != 7) return -2;
    constexpr int CAP = 8, NJ = 7;
    static thread_local double sum[CAP * 12];
    for (int i = 0; i < CAP * 12; ++i) sum[i] = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        float qv[NJ], gv[3], out[NJ];
        for (int d = 0; d < NJ; ++d) qv[d] = q[b * NJ + d];
        for (int i = 0; i < 3; ++i) gv[i] = gpos[b * 3 + i];
        fk_backward_chain<CAP, NJ>([&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, qv, gv, mask, out,
                                   [&](int d) { return qv[d]; },
                                   [&](int k, const float *dF, const float *dt) {
                                       for (int j = 0; j < 9; ++j) sum[k * 12 + j] += dF[j];
                                       for (int j = 0; j < 3; ++j) sum[k * 12 + 9 + j] += dt[j];
                                   });
        if (gq) for (int d = 0; d < NJ; ++d) gq[b * NJ + d] = out[d];
    }
    if (gops)
        for (int k = 0; k < CAP; ++k) {
            for (int j = 0; j < DRM_OPF_STRIDE; ++j) gops[k * DRM_OPF_STRIDE + j] = 0.f;
            for (int j = 0; j < 12; ++j)
                gops[k * DRM_OPF_STRIDE + (j < 9 ? DRM_OPF_FIJ(j / 3, j % 3) : DRM_OPF_TI(j - 9))] = (float)sum[k * 12 + j];
        }
    return 0;
}
int emu_crba_arm(const drm_walk *w, const float *q, int64_t B, float *H) {
    if (!(w->shape & DRM_WALK_ARM_CHAIN) || w->capacity != 8 || w->n_dofs != 7) return -2;
    for (int64_t b = 0; b < B; ++b) {
        float qv[7];
        for (int d = 0; d < 7; ++d) qv[d] = q[b * 7 + d];
        crba_chain<8, 7>([&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, qv,
                         [&](int i, int j, float v) { H[(b * 7 + i) * 7 + j] = v; });
    }
    return 0;
}
int emu_crba(const drm_walk *w, const float *q, int64_t B, float *H) {
    crba_loop(w, q, B, H);
    return 0;
}
int emu_fk_backward(const drm_walk *w, const float *q, int64_t B, int32_t T, const float *gpos, uint64_t mask, float *gq,
                    float *gops) {
    fkb_t(w, q, B, T, gpos, nullptr, nullptr, mask, gq, gops);
    return 0;
}
int emu_fk_backward_rot(const drm_walk *w, const float *q, int64_t B, int32_t T, const float *gpos, const float *grot,
                        uint64_t mask, float *gq, float *gops) {
    fkb_t(w, q, B, T, gpos, nullptr, nullptr, mask, gq, gops, grot);
    return 0;
}
int emu_fk_jacobian_backward(const drm_walk *w, const float *q, int64_t B, const float *gpos, const float *glin,
                             const float *gang, uint64_t mask, float *gq, float *gops) {
    fkb_t(w, q, B, 1, gpos, glin, gang, mask, gq, gops);
    return 0;
}
int emu_rnea_short(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags, float *tau) {
    // the register-parked form of short segments (drm_tree.hpp rnea_tree_walk_short<6>)
    const int n = w->n_dofs;
    const Ctl ctl(w);
    for (int s = 0; s < w->n_segments; ++s)
        if (w->seg_begin[s + 1] - w->seg_begin[s] > 6) return -2;
    for (int64_t b = 0; b < B; ++b)
        for (int seg = 0; seg < w->n_segments; ++seg) {
            Motion ms[DRM_MAX_SLOTS];
            Force fs[DRM_MAX_SLOTS];
            for (auto &F : fs) for (int i = 0; i < 3; ++i) F.la[i] = f2_bcast(0.f);
            rnea_tree_walk_short<6>(
                w->prefix_end, w->seg_begin[seg], w->seg_begin[seg + 1], ctl, [&](int k) { return w->ops_f + k * DRM_OPF_STRIDE; }, flags,
                [&](int d, float &a, float &v, float &acc) { a = q[b * n + d]; v = qd[b * n + d]; acc = qdd ? qdd[b * n + d] : 0.f; },
                [&](int d, float v) { tau[b * n + d] = v; }, [&](int sl, const Motion &M) { ms[sl] = M; },
                [&](int sl, Motion &M) { M = ms[sl]; },
                [&](int sl, const Force &F) { for (int i = 0; i < 3; ++i) fs[sl].la[i] += F.la[i]; },
                [&](int sl, Force &F) { for (int i = 0; i < 3; ++i) { F.la[i] += fs[sl].la[i]; fs[sl].la[i] = f2_bcast(0.f); } });
        }
    return 0;
}
int emu_rnea(const drm_walk *w, const float *q, const float *qd, const float *qdd, int64_t B, int32_t flags, float *tau) {
    rnea_loop(w, q, qd, qdd, B, flags, tau);
    return 0;
}
}
