// drm_sample.hpp — per-sample arithmetic of the FK / Jacobian / RNEA walks.
//
// One lane of a wavefront owns one sample and runs these functions with the
// walk's constants (ops_f / ops_i, include/drm_hip.h) as wave-uniform scalar
// operands.  The 7-DoF arm chains (Franka Panda, KUKA iiwa) run their walks as
// STRAIGHT-LINE code over 8 ops (every loop fully unrolled, per-op state in
// registers with static indices); every other robot goes through the
// loop-structured walks of drm_tree.hpp.  Every joint rotates about its local +z axis by
// +q — the host folds x / y axes and negative axes into exact signed
// permutations of the constants (flatten.py "axis canonicalisation").
//
// The arithmetic restates, per sample, what the reference spreads over
// rigid_body.py:130-165, spatial_vector_algebra.py:14-136,175-338 and
// robot_model.py:139-195,250-375,626-667; the operation
// order is re-associated where that saves work (results agree to fp32 rounding,
// tolerances in tests/).
//
// The header has no HIP dependency beyond the DRM_HD qualifier so that
// tests/host_emu can compile the very same arithmetic with g++ and check it
// against the oracle on a machine without a GPU (test infrastructure only —
// the product never runs it on the CPU).
#pragma once

#include <math.h>
#include <stdint.h>

#include "../../include/drm_hip.h"

#if defined(__HIPCC__)
#define DRM_HD __host__ __device__ __forceinline__
#else
#define DRM_HD inline __attribute__((always_inline))
#endif

// true if the predicate holds in any lane of the wavefront (the host emulation runs one sample at a time)
#if defined(__HIP_DEVICE_COMPILE__)
#define DRM_WAVE_ANY(pred) (__builtin_amdgcn_ballot_w64(pred) != 0ull)
#else
#define DRM_WAVE_ANY(pred) (pred)
#endif

namespace drm {

// 8-byte pair of floats: what v_pk_fma_f32 / v_pk_mul_f32 operate on (see "Packed-FP32 form" below);
// vector_size(8) is understood by hipcc and by g++ (tests/host_emu)
typedef float f2 __attribute__((vector_size(8)));
DRM_HD f2 f2_make(float a, float b) { f2 v = {a, b}; return v; }
DRM_HD f2 f2_bcast(float a) { f2 v = {a, a}; return v; }
// a * b + c with ONE rounding per lane, guaranteed (not left to the compiler's contraction): the Cody-Waite argument
// reduction below is only accurate with true FMAs
DRM_HD f2 f2_fma(f2 a, f2 b, f2 c) {
#if defined(__clang__)
    return __builtin_elementwise_fma(a, b, c);
#else
    f2 v = {__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1])};
    return v;
#endif
}

// keeps a pair out of a later contraction: the value is rounded here (host: a volatile-free identity the optimiser cannot see through)
DRM_HD void pin2(f2 &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#else
    asm volatile("" : "+x"(v));
#endif
}

// ops_i is stored FIELD-MAJOR, [DRM_OPI_STRIDE][capacity].  The backward walks read a single field, the packed control
// word (DRM_OPI_CTRL, include/drm_hip.h), and decode the fields they branch on with scalar bit-field extracts; the
// loop-structured forward walks (drm_tree.hpp) read the two wider words DRM_OPI_W0 / W1.
DRM_HD int ctl_field(int ctl, int field) {
    return field == DRM_OPI_DOF     ? (ctl & 0x7f) - 1
           : field == DRM_OPI_SRC   ? ((ctl >> 7) & 7) - 2
           : field == DRM_OPI_SAVE  ? ((ctl >> 10) & 7) - 1
           : field == DRM_OPI_OUT   ? ((ctl >> 13) & 0x7f) - 1
           : field == DRM_OPI_PERM  ? (ctl >> 20) & 7
                                    : (ctl >> 23) & 1; // DRM_OPI_FLAGS
}
DRM_HD bool ctl_prismatic(int ctl) { return (ctl >> 25) & 1; }
// sin / cos of a joint angle, branch-free.  Argument reduction k = rint(x 2/pi),
// r = x - k pi/2 is done in fp64 (two constants), which keeps r exact to fp32
// rounding for |x| < ~1e9 without a slow path; the kernels then use degree-9/10
// minimax polynomials on [-pi/4, pi/4] (coefficients of the classic fdlibm float
// kernels) — max error ~1 ulp, i.e. the same class as torch.sin/cos on the CPU
// (spatial_vector_algebra.py:14-53 evaluates them in fp32).
// Domain (round 6): |x| <= 1e9.  Beyond it — and for +-Inf / NaN — the quadrant no longer fits an int and the reduced argument
// is noise: sin and cos are NaN, on every path (this is where sincos_one / chain_trig send their large arguments).  The test
// and the select are INTEGER operations on the bit patterns, so they survive -ffinite-math-only (the robots' own kernels,
// specialize.ARM_FLAGS), which may fold a floating-point isnan / select-of-NaN away.
DRM_HD void sincos_f(float x, float &s, float &c) {
    const bool outside = (__builtin_bit_cast(uint32_t, x) & 0x7fffffffu) > 0x4e6e6b28u;   // |x| > 1e9f, Inf, NaN
    if (outside) x = 0.0f;
    const double xd = (double)x;
    const double kd = rint(xd * 0.63661977236758134308);   // 2/pi
    double rd = fma(-kd, 1.57079632679489655800e+00, xd);    // pi/2 hi
    rd = fma(-kd, 6.12323399573676603587e-17, rd);           // pi/2 lo
    const float r = (float)rd;
    const int q = (int)kd;
    const float z = r * r;
    float ps = fmaf(z, 2.7557314297e-06f, -1.9841270114e-04f);
    ps = fmaf(z, ps, 8.3333337680e-03f);
    ps = fmaf(z, ps, -1.6666667163e-01f);
    const float sr = fmaf(r * z, ps, r);
    float pc = fmaf(z, -2.7557314297e-07f, 2.4801587642e-05f);
    pc = fmaf(z, pc, -1.3888889225e-03f);
    pc = fmaf(z, pc, 4.1666667908e-02f);
    pc = fmaf(z, pc, -0.5f);
    const float cr = fmaf(z, pc, 1.0f);
    const bool swap = q & 1;
    const float s0 = swap ? cr : sr;
    const float c0 = swap ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
    s = __builtin_bit_cast(float, outside ? 0x7fc00000u : __builtin_bit_cast(uint32_t, s));
    c = __builtin_bit_cast(float, outside ? 0x7fc00000u : __builtin_bit_cast(uint32_t, c));
}

DRM_HD float rsqrt_f(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rsqrtf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}

// out = a x b
DRM_HD void cross3(const float *a, const float *b, float *out) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

// y = M x (M row-major 3x3)
DRM_HD void mat_vec(const float *M, const float *x, float *y) {
#pragma unroll
    for (int r = 0; r < 3; ++r) y[r] = M[r * 3 + 0] * x[0] + M[r * 3 + 1] * x[1] + M[r * 3 + 2] * x[2];
}

// y = M^T x
DRM_HD void matT_vec(const float *M, const float *x, float *y) {
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = M[0 * 3 + c] * x[0] + M[1 * 3 + c] * x[1] + M[2 * 3 + c] * x[2];
}

// J = F * Rot_z(q), c = cos(q), s = sin(q)
// (rigid_body.py:146-156, spatial_vector_algebra.py:42-53).  c = 1, s = 0 gives J == F exactly.
DRM_HD void joint_rot_z(const float *__restrict__ F, float c, float s, float *J) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        J[r * 3 + 0] = F[r * 3 + 0] * c + F[r * 3 + 1] * s;
        J[r * 3 + 1] = F[r * 3 + 1] * c - F[r * 3 + 0] * s;
        J[r * 3 + 2] = F[r * 3 + 2];
    }
}

struct Pose {
    float R[9];
    float p[3];
};

// R_fixed (row-major) and trans of one op, read out of the row's interleaved FT block (include/drm_hip.h)
struct OpFT {
    float F[9];
    float t[3];
};
DRM_HD OpFT load_ft(const float *__restrict__ of) {
    OpFT o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) o.F[i * 3 + j] = of[DRM_OPF_FIJ(i, j)];
        o.t[i] = of[DRM_OPF_TI(i)];
    }
    return o;
}

// world pose of a link from its parent's: R = Rp J, p = Rp t + pp
// (robot_model.py:186, spatial_vector_algebra.py:98-103)
DRM_HD void compose(const Pose &par, const float *J, const float *__restrict__ t, Pose &out) {
    Pose tmp;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            tmp.R[r * 3 + c] =
                par.R[r * 3 + 0] * J[0 * 3 + c] + par.R[r * 3 + 1] * J[1 * 3 + c] + par.R[r * 3 + 2] * J[2 * 3 + c];
        tmp.p[r] = par.R[r * 3 + 0] * t[0] + par.R[r * 3 + 1] * t[1] + par.R[r * 3 + 2] * t[2] + par.p[r];
    }
    out = tmp;
}

// child of the root link: the root pose is the identity, so R = J and p = t exactly
DRM_HD void compose_root(const float *J, const float *__restrict__ t, Pose &out) {
#pragma unroll
    for (int i = 0; i < 9; ++i) out.R[i] = J[i];
    out.p[0] = t[0]; out.p[1] = t[1]; out.p[2] = t[2];
}

DRM_HD void pose_identity(Pose &P) {
#pragma unroll
    for (int i = 0; i < 9; ++i) P.R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    P.p[0] = P.p[1] = P.p[2] = 0.0f;
}

// undo the axis canonicalisation of a stored frame R~ = R P, P = P_a D_s:  R[:, pi(c)] = d(c) R~[:, c]
// code = a + 3 * (s < 0);  a = 2: pi = identity; a = 0 (joint about x): pi = (1,2,0); a = 1 (about y): pi = (2,0,1);
// s < 0: d = (1,-1,-1)
DRM_HD void unpermute(int code, float *R) {
    if (code != 2) {
        const float d = code >= 3 ? -1.0f : 1.0f;
        const int perm = code >= 3 ? code - 3 : code;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float a = R[r * 3 + 0], b = d * R[r * 3 + 1], c = d * R[r * 3 + 2];
            if (perm == 0)      { R[r * 3 + 1] = a; R[r * 3 + 2] = b; R[r * 3 + 0] = c; }
            else if (perm == 1) { R[r * 3 + 2] = a; R[r * 3 + 0] = b; R[r * 3 + 1] = c; }
            else                { R[r * 3 + 1] = b; R[r * 3 + 2] = c; }
        }
    }
}

// rotation matrix -> quaternion (x, y, z, w): the reference's algorithm and case order
// (spatial_vector_algebra.py:108-136: t = trace + 1 > 1 -> "w" case, else the largest diagonal entry picks
// the x / y / z case, ties resolved as there), written with selects instead of branches so that a wave whose
// lanes fall into different cases does not execute four masked code paths.  Each case's arithmetic is the
// reference's, so results are identical to the branchy form.
DRM_HD void quat_xyzw(const float *R, float *q) {
    const float m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7],
                m22 = R[8];
    const float d21 = m21 - m12, d02 = m02 - m20, d10 = m10 - m01;
    const float s01 = m01 + m10, s20 = m20 + m02, s12 = m12 + m21;
    const float tW = ((m00 + m11) + m22) + 1.0f;
    const float tX = m00 - (m11 + m22) + 1.0f, tY = m11 - (m22 + m00) + 1.0f, tZ = m22 - (m00 + m11) + 1.0f;
    const bool isW = tW > 1.0f;
    const bool yx = m11 > m00;                         // i = 1 beats i = 0
    const bool isZ = !isW && (m22 > (yx ? m11 : m00)); // i = 2 beats the winner of the first test
    const bool isY = !isW && !isZ && yx;
    const float t = isW ? tW : (isZ ? tZ : (isY ? tY : tX));
    const float x = isW ? d21 : (isZ ? s20 : (isY ? s01 : tX));
    const float y = isW ? d02 : (isZ ? s12 : (isY ? tY : s01));
    const float z = isW ? d10 : (isZ ? tZ : (isY ? s12 : s20));
    const float w = isW ? tW : (isZ ? d10 : (isY ? d02 : d21));
    const float scale = 0.5f * rsqrt_f(t);
    q[0] = x * scale;
    q[1] = y * scale;
    q[2] = z * scale;
    q[3] = w * scale;
}

// Orientation of an emitted target: undo the axis canonicalisation, then the quaternion.  Deliberately NOT inlined:
// the multi-target walk is unrolled over up to 32 links and any of them may be a target, so an inlined copy per
// link multiplies the code of the kernel (tens of KB of straight-line code that every wave has to fetch once, which
// is what a small launch then spends its time on); one shared copy costs a call per target instead.
struct Quat4 {
    float v[4];
};
struct Rot9 {
    float v[9];
};
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__
#else
static __attribute__((noinline))
#endif
Quat4 target_quaternion(Rot9 R, int perm_code) {
    Quat4 q;
    unpermute(perm_code, R.v);
    quat_xyzw(R.v, q.v);
    return q;
}


// ---------------------------------------------------------------------------
// Packed-FP32 form of the chain FK (the metric kernel's arithmetic).
//
// gfx950 executes v_pk_fma_f32 / v_pk_mul_f32 (two fp32 lanes per VGPR pair) at the rate of a scalar
// FMA, so the chain is written on 8-byte pairs throughout.  The three rows of a pose evolve
// independently:   row c of R' = (row c of R) J,   p'_c = p_c + (row c of R) . t,   J = F Rot_z(q),
// so the state of row c is two pairs  A_c = (R_c0, R_c1),  B_c = (R_c2, p_c)  and one link costs
//     J01_i = (F_i0, F_i1) cos + (F_i1, F_i0) (sin, -sin)                      6 packed ops (off the chain)
//     A'_c  = R_c0 J01_0 + R_c1 J01_1 + R_c2 J01_2                              9 packed ops
//     B'_c  = R_c0 (F_02, t_0) + R_c1 (F_12, t_1) + R_c2 (F_22, t_2) + (0, p_c) 9 packed ops + 3 adds
// instead of 48 scalar ones; the (F_i0 F_i1) / (F_i2 t_i) pairs are exactly the FT block of an op row
// (include/drm_hip.h).  After link k, B_c = (z_c, p_c): the world joint axis and origin the Jacobian needs.
// Same products as joint_rot_z + compose (the sums are associated differently; fp32 rounding only).
// vector_size(8) is understood by hipcc and by g++ (tests/host_emu).
// ---------------------------------------------------------------------------

struct PoseP {
    f2 A[3]; // A[c] = (R_c0, R_c1)
    f2 B[3]; // B[c] = (R_c2, p_c)
};
struct OpPairs {
    f2 f01[3]; // (F_i0, F_i1)
    f2 f2t[3]; // (F_i2, t_i)
};
DRM_HD OpPairs load_pairs(const float *__restrict__ ft) { // ft = the 12 floats of an FT block (8-byte aligned)
    OpPairs o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.f01[i] = f2_make(ft[2 * i], ft[2 * i + 1]);
        o.f2t[i] = f2_make(ft[6 + 2 * i], ft[7 + 2 * i]);
    }
    return o;
}
// J01_i of a moving joint
DRM_HD void joint_pairs(const OpPairs &o, float c, float s, f2 (&J01)[3]) {
    const f2 cc = f2_bcast(c), sm = f2_make(s, -s);
#pragma unroll
    for (int i = 0; i < 3; ++i) J01[i] = o.f01[i] * cc + f2_make(o.f01[i][1], o.f01[i][0]) * sm;
}
// first link of a chain: the parent is the identity root, so R = J and p = t exactly
DRM_HD void compose_pairs_root(const f2 (&J01)[3], const OpPairs &o, PoseP &out) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { out.A[c] = J01[c]; out.B[c] = o.f2t[c]; }
}
DRM_HD void compose_pairs(const PoseP &P, const f2 (&J01)[3], const OpPairs &o, PoseP &out) {
    PoseP n;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const f2 r0 = f2_bcast(P.A[c][0]), r1 = f2_bcast(P.A[c][1]), r2 = f2_bcast(P.B[c][0]);
        n.A[c] = r0 * J01[0] + r1 * J01[1] + r2 * J01[2];
        f2 b = r0 * o.f2t[0] + r1 * o.f2t[1] + r2 * o.f2t[2];
        b[1] += P.B[c][1];
        n.B[c] = b;
    }
    out = n;
}
DRM_HD void pose_from_pairs(const PoseP &P, Pose &out) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        out.R[c * 3 + 0] = P.A[c][0]; out.R[c * 3 + 1] = P.A[c][1]; out.R[c * 3 + 2] = P.B[c][0];
        out.p[c] = P.B[c][1];
    }
}

// sin / cos of TWO joint angles at once on packed pairs, for |x| <= 1e5 (the caller routes larger arguments
// to sincos_f's fp64 reduction).  Written for instruction count, the metric kernel's bottleneck:
//   * k = rint(x / pi) by the add-magic-constant trick (1.5 * 2^23): no v_rndne / v_cvt, and the parity of k
//     is bit 0 of the biased sum;
//   * r = x - k pi by a three-constant Cody-Waite reduction in fp32 FMAs (pi split into three floats, the
//     classic single-precision scheme), r in [-pi/2, pi/2];
//   * sin x = (-1)^k sin r, cos x = (-1)^k cos r: ONE shared sign flip (xor with parity << 31), no quadrant
//     swap / select logic;
//   * sin r = r + r z S(z), cos r = 1 + z C(z), z = r^2: near-minimax fits on [-pi/2, pi/2] (fit error 1e-10;
//     evaluated in fp32 the error is rounding-dominated, <= 1.2e-7 absolute, i.e. <= 2 ulp near 1).
DRM_HD void sincos_pair(f2 x, f2 &s, f2 &c) {
    const f2 magic = f2_bcast(12582912.0f);                       // 1.5 * 2^23
    const f2 kb = f2_fma(x, f2_bcast(0.318309886f), magic);       // low mantissa bits = rint(x / pi)
    const f2 kf = kb - magic;
    f2 r = f2_fma(kf, f2_bcast(-3.14159202e+00f), x);
    r = f2_fma(kf, f2_bcast(-6.27832947e-07f), r);
    r = f2_fma(kf, f2_bcast(-1.07806051e-14f), r);
    const f2 z = r * r;
    f2 ps = z * f2_bcast(-2.3776610902e-08f) + f2_bcast(2.7522166874e-06f);
    ps = z * ps + f2_bcast(-1.9840880122e-04f);
    ps = z * ps + f2_bcast(8.3333319053e-03f);
    ps = z * ps + f2_bcast(-1.6666667163e-01f);
    const f2 sr = (r * z) * ps + r;
    f2 pc = z * f2_bcast(1.6759177379e-09f) + f2_bcast(-2.7332046670e-07f);
    pc = z * pc + f2_bcast(2.4796934667e-05f);
    pc = z * pc + f2_bcast(-1.3888848480e-03f);
    pc = z * pc + f2_bcast(4.1666664183e-02f);
    pc = z * pc + f2_bcast(-0.5f);
    const f2 cr = z * pc + f2_bcast(1.0f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // copy the elements into scalars first: clang's __builtin_bit_cast of a vector-element lvalue reads
        // element 0 whatever the subscript is
        const float kbi = kb[i], sri = sr[i], cri = cr[i];
        const uint32_t flip = __builtin_bit_cast(uint32_t, kbi) << 31;
        s[i] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, sri) ^ flip);
        c[i] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, cri) ^ flip);
    }
}
constexpr float SINCOS_PAIR_MAX_ARG = 1.0e5f;

// FK of a serial chain whose first NJ links are moving joints driving DoF columns 0..NJ-1 and whose
// remaining CAP - NJ links are fixed joints or identity padding (DRM_WALK_ARM_CHAIN).
//   ft(k)  -> pointer to the FT block of op k (12 floats)
//   q[d]   -> joint angles of this sample
//   joints_done()  -> called once, right after the last MOVING joint: every B[k] is final from here on
//                     (the kernel starts storing the angular Jacobian while the fixed tail is still computed)
// Out: B[k][c] = (z_c, p_c) of every moving joint k, and the end pose.
// sin / cos of one joint angle: sincos_pair's algorithm on scalars (same constants, same operation order), with the
// same wave-uniform escape to the fp64 reduction for |x| > 1e5
DRM_HD void sincos_one(float x, float &s, float &c) {
    if (DRM_WAVE_ANY(!(fabsf(x) <= SINCOS_PAIR_MAX_ARG))) {
        sincos_f(x, s, c);
        return;
    }
    const float magic = 12582912.0f;
    const float kb = __builtin_fmaf(x, 0.318309886f, magic);
    const float kf = kb - magic;
    float r = __builtin_fmaf(kf, -3.14159202e+00f, x);
    r = __builtin_fmaf(kf, -6.27832947e-07f, r);
    r = __builtin_fmaf(kf, -1.07806051e-14f, r);
    const float z = r * r;
    float ps = z * -2.3776610902e-08f + 2.7522166874e-06f;
    ps = z * ps + -1.9840880122e-04f;
    ps = z * ps + 8.3333319053e-03f;
    ps = z * ps + -1.6666667163e-01f;
    const float sr = (r * z) * ps + r;
    float pc = z * 1.6759177379e-09f + -2.7332046670e-07f;
    pc = z * pc + 2.4796934667e-05f;
    pc = z * pc + -1.3888848480e-03f;
    pc = z * pc + 4.1666664183e-02f;
    pc = z * pc + -0.5f;
    const float cr = z * pc + 1.0f;
    const uint32_t flip = __builtin_bit_cast(uint32_t, kb) << 31;
    s = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, sr) ^ flip);
    c = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, cr) ^ flip);
}

// cos / sin of the NJ joint angles of a chain, two joints per packed evaluation
template <int NJ>
DRM_HD void chain_trig(const float (&q)[NJ], float (&cs)[NJ], float (&sn)[NJ]) {
    bool big = false;
#pragma unroll
    for (int d = 0; d < NJ; ++d) big = big || !(fabsf(q[d]) <= SINCOS_PAIR_MAX_ARG);
    if (DRM_WAVE_ANY(big)) { // rare; wave-uniform, so the common path carries no execution-mask juggling
#pragma unroll
        for (int d = 0; d < NJ; ++d) sincos_f(q[d], sn[d], cs[d]);
    } else {
#pragma unroll
        for (int d = 0; d < NJ; d += 2) {
            f2 s2, c2;
            sincos_pair(f2_make(q[d], q[d + 1 < NJ ? d + 1 : d]), s2, c2);
            sn[d] = s2[0]; cs[d] = c2[0];
            if (d + 1 < NJ) { sn[d + 1] = s2[1]; cs[d + 1] = c2[1]; }
        }
    }
}
// the chain itself, given cos / sin of the joint angles (the fused FK + RNEA kernel shares them between its two walks)
template <int CAP, int NJ, class FT, class DONE>
DRM_HD void fk_chain_pairs_trig(FT ft, const float (&cs)[NJ], const float (&sn)[NJ], PoseP &ee, f2 (&B)[NJ][3],
                                DONE joints_done) {
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const OpPairs o = load_pairs(ft(k));
        f2 J01[3];
        if (k < NJ) {
            joint_pairs(o, cs[k], sn[k], J01);
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) J01[i] = o.f01[i];
        }
        if (k == 0) compose_pairs_root(J01, o, ee);
        else compose_pairs(ee, J01, o, ee);
        if (k < NJ) {
#pragma unroll
            for (int c = 0; c < 3; ++c) B[k][c] = ee.B[c];
        }
        if (k == NJ - 1) joints_done();
    }
}
template <int CAP, int NJ, class FT, class DONE>
DRM_HD void fk_chain_pairs(FT ft, const float (&q)[NJ], PoseP &ee, f2 (&B)[NJ][3], DONE joints_done) {
    float cs[NJ], sn[NJ];
    chain_trig<NJ>(q, cs, sn);
    fk_chain_pairs_trig<CAP, NJ>(ft, cs, sn, ee, B, joints_done);
}

// ---------------------------------------------------------------------------
// Reverse-mode derivative of the multi-target FK walk with respect to the joint
// angles and the per-link constants F (R_fixed) and t (trans), for a loss that
// depends on the target POSITIONS (the reference's quaternion is not
// differentiable, spatial_vector_algebra.py:108-136).  This is what torch
// autograd produces for the reference's robot_model.py:139-195 + 223-248 with
// learnable `trans` / `rot_angles` (robot_model.py:669-713,
// examples/learn_kinematics_of_iiwa.py:25-61), restated as one adjoint sweep:
//
//   for every op k (reverse order), with r = p_d - p_k over the targets d below k,
//     G_k = sum g_d                 (dL/dp_k, world frame)
//     M_k = sum g_d r^T             (first moment of the target gradients about p_k)
//     dL/dt_k = R_p^T G_k           dL/dF_k = (R_p^T M_k) (R_p F_k)
//     dL/dq_k = z_k . N_k,          z_k = R~_k e_z,  N_k = sum r x g_d  (antisymmetric part of M_k)
//   and (G, M) move to the parent as  G_p += G_k,  M_p += M_k + G_k (p_k - p_p)^T.
//
//   grad_in(t, G)          adds the loss gradient of target slot t to G[3]
//   pose_save / pose_load  branch-point poses (LDS); slots must be unique per branch point
//   adj_add / adj_take     branch-point adjoints (LDS), take = read-and-add
//   gq_out(d, v)           dL/dq of DoF d
//   param_out(k, dF, dt)   per-sample dL/dF (9), dL/dt (3) of op k, only for ops in param_mask
// ---------------------------------------------------------------------------
struct Adjoint {
    float G[3];
    float M[9];
};

struct NoJacobianGrad { // fk_backward_walk without Jacobian gradients
    DRM_HD void operator()(int, float *) const {}
};
struct NoRotationGrad { // fk_backward_walk without gradients on the targets' rotations
    DRM_HD bool operator()(int, float *) const { return false; }
};
// The two sweeps are LOOPS over the n_ops links of the walk (one control word decoded per iteration, no identity
// padding, nothing indexed by a compile-time op number), the world pose of every op is parked by the caller between
// them (LDS, or HBM for walks that do not fit):
//   ctl = the control-word field of the int table (DRM_OPI_CTRL);   park(k, Pose) / unpark(k, Pose&)
//   grot(t, Rbar[9]) -> true if the loss depends on the ROTATION R_t of target t (the quaternion output: the reference
//   copies entries of R into it, spatial_vector_algebra.py:108-136, so torch autograd differentiates through them), with
//   Rbar = dL/dR_t, row-major, for the target's TRUE frame (after the axis canonicalisation is undone).  A rotation
//   adjoint enters the sweep as  M_t += Rbar R_t^T  (M = Rbar R^T in general; see the JAC note below).
template <bool JAC = false, class QF, class GIN, class PSAVE, class PLOAD, class AADD, class ATAKE, class GQ, class PG,
          class PARK, class UNPARK, class GL = NoJacobianGrad, class GA = NoJacobianGrad, class GR = NoRotationGrad>
DRM_HD void fk_backward_walk(const float *__restrict__ opf, const int32_t *__restrict__ ctl, int n_ops, uint64_t param_mask,
                             bool want_gq, QF qf, GIN grad_in, PSAVE pose_save, PLOAD pose_load, AADD adj_add,
                             ATAKE adj_take, GQ gq_out, PG param_out, PARK park, UNPARK unpark, GL glin = GL(),
                             GA gang = GA(), GR grot = GR()) {
    // ---- forward: world pose of every op, parked for the adjoint sweep -------
    // JAC (the walk is the root -> end-effector chain, its last op the target): column d(k) of the geometric Jacobian
    // is (lin, ang) = (z_k x (p_e - p_k), z_k), so loss gradients (l_k, a_k) on the columns are gradients on the
    // chain's poses:  dL/dz_k = a_k + (p_e - p_k) x l_k  enters M_k as (dL/dz_k) z_k^T (M = Rbar R^T in general: the
    // position-only case above is its special case), dL/dp_k = -(l_k x z_k), and dL/dp_e = sum_k l_k x z_k.
    float Se[3] = {0.0f, 0.0f, 0.0f};
    Pose cur;
    pose_identity(cur);
#pragma unroll 1
    for (int k = 0; k < n_ops; ++k) {
        const float *of = opf + k * DRM_OPF_STRIDE;
        const int c = ctl[k];
        const int dof = ctl_field(c, DRM_OPI_DOF), src = ctl_field(c, DRM_OPI_SRC), save = ctl_field(c, DRM_OPI_SAVE);
        const OpFT o = load_ft(of);
        const bool pris = ctl_prismatic(c);
        float J[9], cs = 1.0f, sn = 0.0f;
        if (dof >= 0 && !pris) sincos_f(qf(dof), sn, cs);
        joint_rot_z(o.F, cs, sn, J);
        if (src >= 0) pose_load(src, cur);
        if (src == DRM_SRC_ROOT) compose_root(J, o.t, cur);
        else compose(cur, J, o.t, cur);
        if (dof >= 0 && pris) { // a prismatic joint slides the frame along its own z axis: p += R e_z q
            const float d = qf(dof);
            cur.p[0] += cur.R[2] * d; cur.p[1] += cur.R[5] * d; cur.p[2] += cur.R[8] * d;
        }
        if (save >= 0) pose_save(save, cur);
        park(k, cur);
        if (JAC && dof >= 0 && !pris) { // (the column of a prismatic joint, (z, 0), does not depend on p_e)
            float l[3];
            glin(dof, l);
            const float z[3] = {cur.R[2], cur.R[5], cur.R[8]};
            Se[0] += l[1] * z[2] - l[2] * z[1];
            Se[1] += l[2] * z[0] - l[0] * z[2];
            Se[2] += l[0] * z[1] - l[1] * z[0];
        }
    }
    const float pe[3] = {cur.p[0], cur.p[1], cur.p[2]}; // JAC: the last op is the target
    // ---- adjoint sweep -------------------------------------------------------
    Adjoint carry;
#pragma unroll
    for (int i = 0; i < 3; ++i) carry.G[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) carry.M[i] = 0.0f;
#pragma unroll 1
    for (int k = n_ops - 1; k >= 0; --k) {
        const float *of = opf + k * DRM_OPF_STRIDE;
        const int c = ctl[k];
        const int dof = ctl_field(c, DRM_OPI_DOF), src = ctl_field(c, DRM_OPI_SRC), save = ctl_field(c, DRM_OPI_SAVE),
                  out = ctl_field(c, DRM_OPI_OUT);
        Adjoint tot;
        if (ctl_field(c, DRM_OPI_FLAGS) & DRM_FLAG_CHILD_IS_NEXT) {
            tot = carry;
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) tot.G[i] = 0.0f;
#pragma unroll
            for (int i = 0; i < 9; ++i) tot.M[i] = 0.0f;
        }
        Pose Pk;
        unpark(k, Pk);
        if (out >= 0) {
            grad_in(out, tot.G);
            float Rb[9];
            if (grot(out, Rb)) {
                float Rt[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) Rt[i] = Pk.R[i];
                unpermute(ctl_field(c, DRM_OPI_PERM), Rt); // the frame the quaternion was taken from
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        tot.M[i * 3 + j] += Rb[i * 3 + 0] * Rt[j * 3 + 0] + Rb[i * 3 + 1] * Rt[j * 3 + 1] + Rb[i * 3 + 2] * Rt[j * 3 + 2];
            }
        }
        const bool pris = ctl_prismatic(c);
        if (JAC) {
            if (out >= 0) { tot.G[0] += Se[0]; tot.G[1] += Se[1]; tot.G[2] += Se[2]; }
            if (dof >= 0 && pris) { // column (lin, ang) = (z_k, 0):  dL/dz_k = l_k
                float l[3];
                glin(dof, l);
                const float z[3] = {Pk.R[2], Pk.R[5], Pk.R[8]};
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) tot.M[i * 3 + j] += l[i] * z[j];
            } else if (dof >= 0) {
                float l[3], a[3];
                glin(dof, l);
                gang(dof, a);
                const float z[3] = {Pk.R[2], Pk.R[5], Pk.R[8]};
                const float r[3] = {pe[0] - Pk.p[0], pe[1] - Pk.p[1], pe[2] - Pk.p[2]};
                const float zb[3] = {a[0] + (r[1] * l[2] - r[2] * l[1]), a[1] + (r[2] * l[0] - r[0] * l[2]),
                                     a[2] + (r[0] * l[1] - r[1] * l[0])};
                tot.G[0] -= l[1] * z[2] - l[2] * z[1];
                tot.G[1] -= l[2] * z[0] - l[0] * z[2];
                tot.G[2] -= l[0] * z[1] - l[1] * z[0];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) tot.M[i * 3 + j] += zb[i] * z[j];
            }
        }
        if (save >= 0) adj_take(save, tot);
        Pose par;
        if (src >= 0) pose_load(src, par);
        else if (src == DRM_SRC_ROOT || k == 0) pose_identity(par);
        else unpark(k - 1, par);
        if (want_gq && dof >= 0) {
            if (pris) { // everything below the joint translates along z_k with q
                gq_out(dof, Pk.R[2] * tot.G[0] + Pk.R[5] * tot.G[1] + Pk.R[8] * tot.G[2]);
            } else {
                const float Nx = tot.M[7] - tot.M[5], Ny = tot.M[2] - tot.M[6], Nz = tot.M[3] - tot.M[1];
                gq_out(dof, Pk.R[2] * Nx + Pk.R[5] * Ny + Pk.R[8] * Nz);
            }
        }
        if ((param_mask >> k) & 1u) {
            float dt[3], A[9], Bm[9], dF[9];
            matT_vec(par.R, tot.G, dt);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    A[r * 3 + cc] = par.R[0 * 3 + r] * tot.M[0 * 3 + cc] + par.R[1 * 3 + r] * tot.M[1 * 3 + cc] +
                                    par.R[2 * 3 + r] * tot.M[2 * 3 + cc];
                    Bm[r * 3 + cc] = par.R[r * 3 + 0] * of[DRM_OPF_FIJ(0, cc)] + par.R[r * 3 + 1] * of[DRM_OPF_FIJ(1, cc)] +
                                     par.R[r * 3 + 2] * of[DRM_OPF_FIJ(2, cc)];
                }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    dF[r * 3 + cc] = A[r * 3 + 0] * Bm[0 * 3 + cc] + A[r * 3 + 1] * Bm[1 * 3 + cc] + A[r * 3 + 2] * Bm[2 * 3 + cc];
            if (dof >= 0 && pris) { // p_k also depends on F through the slide F e_z q
                const float d = qf(dof);
                dF[2] += dt[0] * d; dF[5] += dt[1] * d; dF[8] += dt[2] * d;
            }
            param_out(k, dF, dt);
        }
        if (src != DRM_SRC_ROOT) {
            const float r[3] = {Pk.p[0] - par.p[0], Pk.p[1] - par.p[1], Pk.p[2] - par.p[2]};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) tot.M[i * 3 + j] += tot.G[i] * r[j];
            if (src >= 0) adj_add(src, tot);
            else carry = tot;
        }
    }
}

// Reverse-mode FK of a serial chain (DRM_WALK_ARM_CHAIN, one target = the end of the chain) for a loss on the
// target's POSITION: fk_backward_walk without the int table, the stored poses and the adjoint sweep.  With one
// target the adjoints are closed-form in the forward quantities (g = dL/dp_e, r_k = p_e - p_k):
//     dL/dq_k = g . (z_k x r_k)                  (the linear-Jacobian column: fk_chain_pairs already yields z_k, p_k)
//     dL/dt_k = R_p^T g            dL/dF_k = (R_p^T g) (F_k^T R_p^T r_k)^T        (R_p = rotation of op k's parent)
// so the sample costs ONE packed chain FK + 7 cross/dot products; the constant gradients of the (few) ops in
// param_mask re-walk the chain to their parent in a LOOP (wave-uniform, cold, and it must not cost the hot path
// registers: op numbers are run-time values there, angles come back through q_at).
//   ft(k) -> FT block of op k;   gq[d] <- dL/dq_d;   q_at(d) -> q[d] for a run-time d;
//   param_out(k, dF[9], dt[3]) for ops in param_mask (run-time k)
//   g_of(pe, g): dL/dp_e once the end position is known — a given gradient (drm_fk_backward), or the gradient of a loss
//   evaluated inside the kernel (drm_fk_mse: g = 2 (p_e - target) / (3 B))
//   ft_walk(k): the FT block of op k for the (unrolled) chain walk — k is a compile-time constant there, so a kernel may hand out
//   registers it filled ahead of time; ft(k): the same for the cold parameter loop, where k is a run-time value (LDS)
template <int CAP, int NJ, class FTW, class FT, class GOF, class QAT, class PG>
DRM_HD void fk_backward_chain_g(FTW ft_walk, FT ft, const float (&q)[NJ], GOF g_of, uint64_t param_mask, float (&gq)[NJ], QAT q_at, PG param_out);
template <int CAP, int NJ, class FT, class QAT, class PG>
DRM_HD void fk_backward_chain(FT ft, const float (&q)[NJ], const float (&g)[3], uint64_t param_mask, float (&gq)[NJ],
                              QAT q_at, PG param_out) {
    fk_backward_chain_g<CAP, NJ>(ft, ft, q, [&](const float (&)[3], float (&go)[3]) { go[0] = g[0]; go[1] = g[1]; go[2] = g[2]; },
                                 param_mask, gq, q_at, param_out);
}
template <int CAP, int NJ, class FTW, class FT, class GOF, class QAT, class PG>
DRM_HD void fk_backward_chain_g(FTW ft_walk, FT ft, const float (&q)[NJ], GOF g_of, uint64_t param_mask, float (&gq)[NJ], QAT q_at, PG param_out) {
    PoseP ee;
    f2 B[NJ][3];
    fk_chain_pairs<CAP, NJ>(ft_walk, q, ee, B, [] {});
    const float pe[3] = {ee.B[0][1], ee.B[1][1], ee.B[2][1]};
    float g[3];
    g_of(pe, g);
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        const float z[3] = {B[k][0][0], B[k][1][0], B[k][2][0]};
        const float r[3] = {pe[0] - B[k][0][1], pe[1] - B[k][1][1], pe[2] - B[k][2][1]};
        gq[k] = g[0] * (z[1] * r[2] - z[2] * r[1]) + g[1] * (z[2] * r[0] - z[0] * r[2]) + g[2] * (z[0] * r[1] - z[1] * r[0]);
    }
#pragma unroll 1
    for (int k = 0; k < CAP; ++k) {
        if (!((param_mask >> k) & 1u)) continue;
        Pose par;
        pose_identity(par);
#pragma unroll 1
        for (int j = 0; j < k; ++j) {
            const OpFT o = load_ft(ft(j));
            float J[9], c = 1.0f, sn = 0.0f;
            if (j < NJ) sincos_f(q_at(j), sn, c);
            joint_rot_z(o.F, c, sn, J);
            compose(par, J, o.t, par);
        }
        const OpFT o = load_ft(ft(k));
        float a[3], pk[3], r[3], u[3], c[3], dF[9];
        matT_vec(par.R, g, a);                        // dL/dt_k = R_p^T g
        mat_vec(par.R, o.t, pk);
#pragma unroll
        for (int i = 0; i < 3; ++i) r[i] = pe[i] - (par.p[i] + pk[i]);
        matT_vec(par.R, r, u);
        matT_vec(o.F, u, c);                          // c = F_k^T R_p^T r_k
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) dF[i * 3 + j] = a[i] * c[j];
        param_out(k, dF, a);
    }
}

// ---------------------------------------------------------------------------
// RNEA over the whole tree (robot_model.py:250-375).  Body-frame Pluecker
// coordinates at the link origin, as in the reference.
//   qf(d, q, qd, qdd)   -> joint state of DoF d
//   tau_out(d, value)   -> torque of DoF d
//   motion / force slots: branch-point state, kept in LDS by the kernel
// ---------------------------------------------------------------------------
// Written on packed FP32 pairs like the chain FK: velocity-like and acceleration-like quantities go through the
// same linear maps (J^T x, x cross t, I x), so they travel as pairs (w_i, alpha_i), (v_i, a_i) and, on the way
// back, (f_i, n_i); one v_pk_fma_f32 then does the work of two scalar FMAs.
struct Motion {
    f2 wa[3]; // (angular velocity, angular acceleration)
    f2 va[3]; // (linear velocity, linear acceleration)
};

struct Force {
    f2 la[3]; // (linear, angular)
};

DRM_HD void motion_root(Motion &M, float g) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { M.wa[i] = f2_bcast(0.0f); M.va[i] = f2_bcast(0.0f); }
    M.va[2] = f2_make(0.0f, g); // robot_model.py:344-350: gravity enters as a base acceleration (0,0,+9.81)
}

// out = a x b on pairs (element-wise in the pair dimension); b may be a broadcast scalar vector
DRM_HD void cross3_pp(const f2 *a, const f2 *b, f2 *out) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}
DRM_HD void cross3_sp(const float *a, const f2 *b, f2 *out) { // scalar vector x pair vector
    out[0] = f2_bcast(a[1]) * b[2] - f2_bcast(a[2]) * b[1];
    out[1] = f2_bcast(a[2]) * b[0] - f2_bcast(a[0]) * b[2];
    out[2] = f2_bcast(a[0]) * b[1] - f2_bcast(a[1]) * b[0];
}
DRM_HD void cross3_ps(const f2 *a, const float *b, f2 *out) { // pair vector x scalar vector
    out[0] = a[1] * f2_bcast(b[2]) - a[2] * f2_bcast(b[1]);
    out[1] = a[2] * f2_bcast(b[0]) - a[0] * f2_bcast(b[2]);
    out[2] = a[0] * f2_bcast(b[1]) - a[1] * f2_bcast(b[0]);
}
// y = M x, y = M^T x on pairs (M row-major 3x3 scalars)
DRM_HD void mat_vec_p(const float *M, const f2 *x, f2 *y) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
        y[r] = f2_bcast(M[r * 3 + 0]) * x[0] + f2_bcast(M[r * 3 + 1]) * x[1] + f2_bcast(M[r * 3 + 2]) * x[2];
}
DRM_HD void matT_vec_p(const float *M, const f2 *x, f2 *y) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
        y[c] = f2_bcast(M[0 * 3 + c]) * x[0] + f2_bcast(M[1 * 3 + c]) * x[1] + f2_bcast(M[2 * 3 + c]) * x[2];
}

// one link of the forward sweep: motion of the link from its parent's (both in their own body frames)
//   velocity (robot_model.py:189-193):       w = J^T w_p + wj e_z ;          v = J^T (v_p + w_p x t)
//   acceleration (robot_model.py:269-277):  al = J^T al_p + aj e_z + w x (wj e_z) ;  a = J^T (a_p + al_p x t) + v x (wj e_z)
DRM_HD void rnea_link_motion(const float *J, const float *t, float wj, float aj, const Motion &par, Motion &out) {
    f2 x[3], tmp[3];
    Motion N;
    matT_vec_p(J, par.wa, N.wa);
    cross3_ps(par.wa, t, x);
#pragma unroll
    for (int i = 0; i < 3; ++i) tmp[i] = par.va[i] + x[i];
    matT_vec_p(J, tmp, N.va);
    N.wa[2] += f2_make(wj, aj);
    N.wa[0][1] += N.wa[1][0] * wj; N.wa[1][1] -= N.wa[0][0] * wj;
    N.va[0][1] += N.va[1][0] * wj; N.va[1][1] -= N.va[0][0] * wj;
    out = N;
}
// joint transform of one op (child -> parent: x_p = J x_c + t) from its constants and joint value:
//   revolute   J = F Rot_z(q), t = trans          prismatic   J = F, t = trans + F e_z q          fixed  J = F, t = trans
DRM_HD void joint_transform(const OpFT &o, bool moving, bool prismatic, float q, float c, float s, float *J, float *t) {
    if (moving && !prismatic) {
        joint_rot_z(o.F, c, s, J);
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) J[i] = o.F[i];
    }
    const float d = (moving && prismatic) ? q : 0.0f;
    t[0] = o.t[0] + o.F[2] * d;
    t[1] = o.t[1] + o.F[5] * d;
    t[2] = o.t[2] + o.F[8] * d;
}

// the same for a prismatic joint (joint velocity (0, e_z qd)); used by the loop-structured walks (drm_tree.hpp) and the
// backward walk below
DRM_HD void motion_step(const float *J, const float *t, float wj, float aj, bool prismatic, const Motion &par, Motion &out) {
    if (!prismatic) {
        rnea_link_motion(J, t, wj, aj, par, out);
        return;
    }
    // v = J^T (v_p + w_p x t) + e_z qd ;  a = J^T (a_p + al_p x t) + e_z qdd + w x (e_z qd)   (joint velocity (0, e_z qd))
    f2 x[3], tmp[3];
    Motion N;
    matT_vec_p(J, par.wa, N.wa);
    cross3_ps(par.wa, t, x);
#pragma unroll
    for (int i = 0; i < 3; ++i) tmp[i] = par.va[i] + x[i];
    matT_vec_p(J, tmp, N.va);
    N.va[2] += f2_make(wj, aj);
    N.va[0][1] += N.wa[1][0] * wj;
    N.va[1][1] -= N.wa[0][0] * wj;
    out = N;
}

// body force f = I a + v x* (I v)  (robot_model.py:289-293, spatial_vector_algebra.py:321-338, 215-224)
// pairs: (h, g) = (I v, I a):  lin = m (v, a) - mc x (w, al) ;  ang = Io (w, al) + mc x (v, a)
DRM_HD void rnea_body_force_hg(float m, const float *mc, const float *Io, const Motion &N, Force &out, f2 (&hgl)[3],
                               f2 (&hga)[3]);
DRM_HD void rnea_body_force(float m, const float *mc, const float *Io, const Motion &N, Force &out) {
    f2 hgl[3], hga[3];
    rnea_body_force_hg(m, mc, Io, N, out, hgl, hga);
}
// ... also handing out the momentum-like pairs (h, g).lin = hgl, (h, g).ang = hga (the packed adjoint needs h again)
DRM_HD void rnea_body_force_hg(float m, const float *mc, const float *Io, const Motion &N, Force &out, f2 (&hgl)[3],
                               f2 (&hga)[3]) {
    f2 x[3], y[3];
    cross3_sp(mc, N.wa, x);
#pragma unroll
    for (int i = 0; i < 3; ++i) hgl[i] = f2_bcast(m) * N.va[i] - x[i];
    mat_vec_p(Io, N.wa, y);
    cross3_sp(mc, N.va, x);
#pragma unroll
    for (int i = 0; i < 3; ++i) hga[i] = y[i] + x[i];
    // f.lin = g.lin + w x h.lin ;  f.ang = g.ang + (w x h.ang + v x h.lin)
    const float w[3] = {N.wa[0][0], N.wa[1][0], N.wa[2][0]}, v[3] = {N.va[0][0], N.va[1][0], N.va[2][0]};
    const float hl[3] = {hgl[0][0], hgl[1][0], hgl[2][0]}, ha[3] = {hga[0][0], hga[1][0], hga[2][0]};
    float xl[3], xa[3], ya[3];
    cross3(w, hl, xl);
    cross3(w, ha, xa);
    cross3(v, hl, ya);
#pragma unroll
    for (int i = 0; i < 3; ++i) out.la[i] = f2_make(hgl[i][1] + xl[i], hga[i][1] + (xa[i] + ya[i]));
}
// one link of the backward sweep: force.transform(joint_pose) (spatial_vector_algebra.py:281-291):
// lin = J f ; ang = t x (J f) + J n
DRM_HD void rnea_link_force_up(const float *J, const float *t, const Force &tot, Force &up) {
    mat_vec_p(J, tot.la, up.la);
    const float l[3] = {up.la[0][0], up.la[1][0], up.la[2][0]};
    up.la[0][1] += t[1] * l[2] - t[2] * l[1];
    up.la[1][1] += t[2] * l[0] - t[0] * l[2];
    up.la[2][1] += t[0] * l[1] - t[1] * l[0];
}

// RNEA of a serial chain (DRM_WALK_ARM_CHAIN: NJ moving joints driving DoF columns 0..NJ-1, then CAP - NJ fixed
// links or identity padding): the straight-line form of drm_tree.hpp rnea_tree_walk without control words, with the joint
// transforms kept in registers between the sweeps and two joints per sincos evaluation.
//   row(k) -> pointer to op k's constant row (DRM_OPF_* layout)
//   fput(k, Force) / fget(k, Force&) -> body force of link k, parked between the sweeps (LDS in the kernel: the
//   48 floats would otherwise be the registers that keep a second wave off the SIMD)
//   KEEP: the body forces of the last KEEP links never leave the registers (only links 0 .. CAP-KEEP-1 are parked; fput /
//   fget see no others).  Measured and left at 0: with KEEP = 1 or 2 the register allocator loses the LDS stores as anchors
//   and needs 194-220 VGPRs for the same arithmetic (24-41 spilled under a three-wave bound), profiles/r03_resource_usage.txt
#ifndef DRM_RNEA_KEEP
#define DRM_RNEA_KEEP 0
#endif
// A scheduling barrier between the links of the straight-line sweeps (device code only).  Without it the compiler hoists
// the constant reads (broadcast ds_read_b128) of later links over the current one and the live ranges pile up: 164-170 VGPRs;
// with it 136, no spills, and the kernels are 3-7 % faster (profiles/r03_ab_rnea_fence.txt).  -DDRM_RNEA_NO_FENCE for A/B runs.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DRM_RNEA_NO_FENCE)
#define DRM_RNEA_LINK_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define DRM_RNEA_LINK_FENCE() ((void)0)
#endif
// PREF: software prefetch of the constants, one link ahead.  The sweeps read a link's 26 constants from LDS (broadcast reads of the
// staged table) and, op by op, waited for reads they had issued a few instructions earlier — three times per link in the
// forward sweep — because a read cannot be hoisted over the parking stores of the link before it (same LDS array: may alias) nor
// over the link fence.  With PREF the reads of link k + 1 (k - 1 on the way back, together with its parked force) are issued
// BEFORE the arithmetic of link k, into a second set of registers (+26 VGPRs), so they land while link k computes.
template <int CAP, int NJ, int KEEP = DRM_RNEA_KEEP, bool PREF = false, class ROW, class FPUT, class FGET>
DRM_HD void rnea_chain_trig(ROW row, bool gravity, bool damping, const float (&cs)[NJ], const float (&sn)[NJ],
                            const float (&qd)[NJ], const float (&qdd)[NJ], float (&tau)[NJ], FPUT fput, FGET fget);
template <int CAP, int NJ, int KEEP = DRM_RNEA_KEEP, bool PREF = false, class ROW, class FPUT, class FGET>
DRM_HD void rnea_chain(ROW row, bool gravity, bool damping, const float (&q)[NJ], const float (&qd)[NJ],
                       const float (&qdd)[NJ], float (&tau)[NJ], FPUT fput, FGET fget) {
    float cs[NJ], sn[NJ];
    chain_trig<NJ>(q, cs, sn);
    rnea_chain_trig<CAP, NJ, KEEP, PREF>(row, gravity, damping, cs, sn, qd, qdd, tau, fput, fget);
}
#ifndef DRM_STAMP
#define DRM_STAMP(slot) ((void)0) /* (drm_common.hpp's development time stamps; this header is also compiled for the host) */
#endif
constexpr int RNEA_ROW_FLOATS = 28; // floats 0 .. 25 of an op row: FT block, mass, m c, I_o, damping (16-byte multiples)
template <int N>
DRM_HD void rnea_row_copy(const float *__restrict__ of, float (&r)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = of[i];
}
template <int CAP, int NJ, int KEEP, bool PREF, class ROW, class FPUT, class FGET>
DRM_HD void rnea_chain_trig(ROW row, bool gravity, bool damping, const float (&cs)[NJ], const float (&sn)[NJ],
                            const float (&qd)[NJ], const float (&qdd)[NJ], float (&tau)[NJ], FPUT fput, FGET fget) {
    static_assert(KEEP >= 0 && KEEP <= CAP, "KEEP == CAP: nothing is parked, fput / fget are never called");
    Force kept[KEEP > 0 ? KEEP : 1];
    // The joint transforms are rebuilt in the backward sweep (12 VALU ops + three broadcast LDS reads per link)
    // instead of being kept: 72 fewer live registers.
    Motion cur;
    motion_root(cur, gravity ? 9.81f : 0.0f);
    float buf[2][PREF ? RNEA_ROW_FLOATS : 1]; // two register sets, used alternately (the loop is unrolled: no copies)
    if constexpr (PREF) rnea_row_copy(row(0), buf[0]);
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        DRM_RNEA_LINK_FENCE();
        if constexpr (PREF) {
            if (k + 1 < CAP) rnea_row_copy(row(k + 1), buf[(k + 1) & 1]);
            DRM_RNEA_LINK_FENCE(); // the reads of link k + 1 are issued here, ahead of link k's arithmetic
        }
        const float *of = PREF ? buf[k & 1] : row(k);
        const OpFT o = load_ft(of);
        float J[9];
        if (k < NJ) {
            joint_rot_z(o.F, cs[k], sn[k], J);
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) J[i] = o.F[i];
        }
        rnea_link_motion(J, o.t, k < NJ ? qd[k] : 0.0f, k < NJ ? qdd[k] : 0.0f, cur, cur);
        Force fk;
        rnea_body_force(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, cur, fk);
        if (k >= CAP - KEEP) kept[k - (CAP - KEEP)] = fk;
        else fput(k, fk);
    }
    DRM_STAMP(2); // forward sweep done
    Force tot;
#pragma unroll
    for (int i = 0; i < 3; ++i) tot.la[i] = f2_bcast(0.0f);
    // the way back needs a link's FT block and damping again, and its parked force
    constexpr int BACK_FLOATS = DRM_OPF_FT_FLOATS;
    float back[2][PREF ? BACK_FLOATS : 1], damp2[2];
    Force f2buf[PREF ? 2 : 1];
    if constexpr (PREF) {
        rnea_row_copy(row(CAP - 1), back[(CAP - 1) & 1]);
        damp2[(CAP - 1) & 1] = row(CAP - 1)[DRM_OPF_DAMP];
        if (CAP - 1 < CAP - KEEP) fget(CAP - 1, f2buf[(CAP - 1) & 1]);
    }
#pragma unroll
    for (int k = CAP - 1; k >= 0; --k) {
        DRM_RNEA_LINK_FENCE();
        Force fk;
        if constexpr (PREF) {
            if (k > 0) {
                rnea_row_copy(row(k - 1), back[(k - 1) & 1]);
                damp2[(k - 1) & 1] = row(k - 1)[DRM_OPF_DAMP];
                if (k - 1 < CAP - KEEP) fget(k - 1, f2buf[(k - 1) & 1]);
            }
            DRM_RNEA_LINK_FENCE();
            if (k < CAP - KEEP) fk = f2buf[k & 1];
        }
        const float *of = PREF ? back[k & 1] : row(k);
        if (k >= CAP - KEEP) fk = kept[k - (CAP - KEEP)];
        else if constexpr (!PREF) fget(k, fk);
#pragma unroll
        for (int i = 0; i < 3; ++i) tot.la[i] += fk.la[i];
        if (k < NJ) {
            if constexpr (PREF) tau[k] = tot.la[2][1] + (damping ? damp2[k & 1] * qd[k] : 0.0f);
            else tau[k] = tot.la[2][1] + (damping ? of[DRM_OPF_DAMP] * qd[k] : 0.0f);
        }
        if (k > 0) {
            const OpFT o = load_ft(of);
            float J[9];
            if (k < NJ) {
                joint_rot_z(o.F, cs[k], sn[k], J);
            } else {
#pragma unroll
                for (int i = 0; i < 9; ++i) J[i] = o.F[i];
            }
            Force up;
            rnea_link_force_up(J, o.t, tot, up);
            tot = up;
        }
    }
}

// ---------------------------------------------------------------------------
// RNEA of "an arm that carries a hand": P serial prefix ops (the arm, with whatever fixed links sit in it: a base, a
// flange, a palm) and K serial sub-chains of L ops each that all hang off the LAST prefix op (the fingers of a gripper
// or a hand; Franka Panda with its gripper: P = 9, K = 2, L = 1; Kinova Jaco: 7, 3, 2; KUKA iiwa7 + Allegro: 8, 4, 4).
// Straight-line form of drm_tree.hpp rnea_tree_walk for this shape: the prefix is rnea_chain_trig's walk; a sub-chain
// runs its forward AND backward sweep while the palm's motion is still in registers, so its body forces never leave the
// registers and only the prefix's are parked (the loop form parks every link: 35 KB of LDS per 64 samples on the arm with
// a hand, three wavefronts per CU).  Every op is walked as a joint about +z: a fixed op has angle 0, cos = 1, sin = 0
// (J == F exactly) and zero joint velocity / acceleration, so the shape needs no flags.
//   row(k)      -> op k's constant row (walk order: prefix, then sub-chain 0, sub-chain 1, ...)
//   cs / sn / qd / qdd -> of the prefix ops; hq(j, i, q, qd, qdd) -> joint state of op i of sub-chain j, fetched when the
//                  sub-chain starts (0 for a fixed op)
//   tau_p[k]    -> generalised force at prefix op k (+ damping); htau(j, i, value) the same for a sub-chain op; the
//                  caller drops those of fixed ops
//   fput / fget -> body forces of prefix ops 0 .. P-2 between the sweeps
// The sub-chains are a ROLLED loop (K is a run-time count): unrolled, the register allocator needed 250 (K = 3, L = 2) to 600
// (K = 4, L = 4) registers for what is the same 130-register body K times.
// ---------------------------------------------------------------------------
//   kind(op)    -> wave-uniform bits of op (walk order): 1 = the joint moves, 2 = it is prismatic (slides along +z by q;
//                  its torque is the z FORCE at the op).  q[] of the prefix ops is only read for prismatic joints.
template <int P, int L, class ROW, class KIND, class HQ, class HTAU, class FPUT, class FGET>
DRM_HD void rnea_arm_hand(ROW row, KIND kind, int K, bool gravity, bool damping, const float (&q)[P], const float (&cs)[P],
                          const float (&sn)[P], const float (&qd)[P], const float (&qdd)[P], HQ hq, float (&tau_p)[P], HTAU htau,
                          FPUT fput, FGET fget) {
    Motion cur;
    Force tot;
    motion_root(cur, gravity ? 9.81f : 0.0f);
#pragma unroll
    for (int k = 0; k < P; ++k) {
        DRM_RNEA_LINK_FENCE();
        const float *of = row(k);
        const int kd = kind(k);
        float J[9], t[3];
        joint_transform(load_ft(of), kd & 1, kd & 2, q[k], cs[k], sn[k], J, t);
        motion_step(J, t, qd[k], qdd[k], kd & 2, cur, cur);
        Force fk;
        rnea_body_force(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, cur, fk);
        if (k < P - 1) fput(k, fk);
        else tot = fk;
    }
    // the sub-chains: forward and backward sweep of each while `cur` (the motion of the last prefix op) stays live
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
        float hqv[L], hqd[L], hqdd[L], hc[L], hs[L];
#pragma unroll
        for (int i = 0; i < L; ++i) hq(j, i, hqv[i], hqd[i], hqdd[i]);
        chain_trig<L>(hqv, hc, hs);
        Motion M = cur;
        Force fs[L];
#pragma unroll
        for (int i = 0; i < L; ++i) {
            DRM_RNEA_LINK_FENCE();
            const float *of = row(P + j * L + i);
            const int kd = kind(P + j * L + i);
            float J[9], t[3];
            joint_transform(load_ft(of), kd & 1, kd & 2, hqv[i], hc[i], hs[i], J, t);
            motion_step(J, t, hqd[i], hqdd[i], kd & 2, M, M);
            rnea_body_force(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, M, fs[i]);
        }
        Force ft = fs[L - 1];
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            DRM_RNEA_LINK_FENCE();
            const float *of = row(P + j * L + i);
            const int kd = kind(P + j * L + i);
            if (i < L - 1) {
#pragma unroll
                for (int c = 0; c < 3; ++c) ft.la[c] += fs[i].la[c];
            }
            htau(j, i, ((kd & 2) ? ft.la[2][0] : ft.la[2][1]) + (damping ? of[DRM_OPF_DAMP] * hqd[i] : 0.0f));
            float J[9], t[3];
            joint_transform(load_ft(of), kd & 1, kd & 2, hqv[i], hc[i], hs[i], J, t);
            Force up;
            rnea_link_force_up(J, t, ft, up); // after i = 0: in the frame of the last prefix op
            ft = up;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) tot.la[c] += ft.la[c];
    }
#pragma unroll
    for (int k = P - 1; k >= 0; --k) {
        DRM_RNEA_LINK_FENCE();
        const float *of = row(k);
        const int kd = kind(k);
        if (k < P - 1) {
            Force fk;
            fget(k, fk);
#pragma unroll
            for (int c = 0; c < 3; ++c) tot.la[c] += fk.la[c];
        }
        tau_p[k] = ((kd & 2) ? tot.la[2][0] : tot.la[2][1]) + (damping ? of[DRM_OPF_DAMP] * qd[k] : 0.0f);
        if (k > 0) {
            float J[9], t[3];
            joint_transform(load_ft(of), kd & 1, kd & 2, q[k], cs[k], sn[k], J, t);
            Force up;
            rnea_link_force_up(J, t, tot, up);
            tot = up;
        }
    }
}

// ---------------------------------------------------------------------------
// RNEA of a serial chain, TWO SAMPLES PER LANE.
//
// The form above pairs velocity- with acceleration-like quantities of ONE sample, which leaves the genuinely scalar
// parts of the recursion (the joint rotation, the v x* (I v) cross products, a third of the motion update, the sincos
// of an odd joint count) on half-empty instructions: 1 141 VALU instructions per 64 samples, 512 of them packed.  Here a
// pair is the SAME quantity of two samples (rows b and b + 64 of a 128-sample tile), so every instruction of the
// per-sample recursion is a full v_pk_*_f32: the instruction count per sample drops by a third (profiles/r03_*), the
// wave-uniform link constants enter as broadcast operands (op_sel), and two dword loads of consecutive tiles land in
// the two halves of a register pair without a move.  The price is registers (two samples of state per lane: two waves per
// SIMD) — which is what an issue-bound kernel can afford.
// Arithmetic per sample: the reference's formulas in the association order written below; same
// results as rnea_chain_trig up to fp32 rounding (the sums are associated differently).
// ---------------------------------------------------------------------------
struct Motion2 {
    f2 w[3], al[3], v[3], a[3]; // angular / linear velocity and acceleration, body frame; [i] = (sample A, sample B)
};
struct Force2 {
    f2 f[3], n[3]; // linear, angular
};
// the first two columns of J = F Rot_z(q) per sample; the third column is the constant F[:, 2]
struct Joint2 {
    f2 c0[3], c1[3];
};
DRM_HD void joint2_moving(const float *F, f2 c, f2 s, Joint2 &J) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        J.c0[r] = f2_bcast(F[r * 3 + 0]) * c + f2_bcast(F[r * 3 + 1]) * s;
        J.c1[r] = f2_bcast(F[r * 3 + 1]) * c - f2_bcast(F[r * 3 + 0]) * s;
    }
}
DRM_HD void joint2_fixed(const float *F, Joint2 &J) {
#pragma unroll
    for (int r = 0; r < 3; ++r) { J.c0[r] = f2_bcast(F[r * 3 + 0]); J.c1[r] = f2_bcast(F[r * 3 + 1]); }
}
DRM_HD void joint2_T(const Joint2 &J, const float *F, const f2 *x, f2 *y) { // y = J^T x
    y[0] = J.c0[0] * x[0] + J.c0[1] * x[1] + J.c0[2] * x[2];
    y[1] = J.c1[0] * x[0] + J.c1[1] * x[1] + J.c1[2] * x[2];
    y[2] = f2_bcast(F[2]) * x[0] + f2_bcast(F[5]) * x[1] + f2_bcast(F[8]) * x[2];
}
DRM_HD void joint2_N(const Joint2 &J, const float *F, const f2 *x, f2 *y) { // y = J x
#pragma unroll
    for (int r = 0; r < 3; ++r) y[r] = J.c0[r] * x[0] + J.c1[r] * x[1] + f2_bcast(F[r * 3 + 2]) * x[2];
}
DRM_HD void cross2_vc(const f2 *a, const float *b, f2 *out) { // per-sample vector x constant vector
    out[0] = a[1] * f2_bcast(b[2]) - a[2] * f2_bcast(b[1]);
    out[1] = a[2] * f2_bcast(b[0]) - a[0] * f2_bcast(b[2]);
    out[2] = a[0] * f2_bcast(b[1]) - a[1] * f2_bcast(b[0]);
}
DRM_HD void cross2_cv(const float *a, const f2 *b, f2 *out) { // constant vector x per-sample vector
    out[0] = f2_bcast(a[1]) * b[2] - f2_bcast(a[2]) * b[1];
    out[1] = f2_bcast(a[2]) * b[0] - f2_bcast(a[0]) * b[2];
    out[2] = f2_bcast(a[0]) * b[1] - f2_bcast(a[1]) * b[0];
}
DRM_HD void cross2_vv(const f2 *a, const f2 *b, f2 *out) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}
// motion of a link from its parent's (rnea_link_motion on sample pairs); wj / aj = joint velocity / acceleration (0 for a fixed link)
DRM_HD void rnea2_link_motion(const Joint2 &J, const float *F, const float *t, f2 wj, f2 aj, const Motion2 &par, Motion2 &out) {
    Motion2 N;
    f2 x[3], tmp[3];
    joint2_T(J, F, par.w, N.w);
    joint2_T(J, F, par.al, N.al);
    cross2_vc(par.w, t, x);
#pragma unroll
    for (int i = 0; i < 3; ++i) tmp[i] = par.v[i] + x[i];
    joint2_T(J, F, tmp, N.v);
    cross2_vc(par.al, t, x);
#pragma unroll
    for (int i = 0; i < 3; ++i) tmp[i] = par.a[i] + x[i];
    joint2_T(J, F, tmp, N.a);
    N.w[2] += wj;
    N.al[2] += aj;
    N.al[0] += N.w[1] * wj; N.al[1] -= N.w[0] * wj; // w x (wj e_z)
    N.a[0] += N.v[1] * wj;  N.a[1] -= N.v[0] * wj;  // v x (wj e_z)
    out = N;
}
// the first link of a chain: its parent is the resting root (w = al = v = 0, a = (0, 0, g))
DRM_HD void rnea2_first_motion(const Joint2 &J, const float *F, float g, f2 wj, f2 aj, Motion2 &out) {
    const f2 z = f2_bcast(0.0f);
    out.w[0] = z; out.w[1] = z; out.w[2] = wj;
    out.al[0] = z; out.al[1] = z; out.al[2] = aj;
    out.v[0] = z; out.v[1] = z; out.v[2] = z;
    out.a[0] = J.c0[2] * f2_bcast(g); out.a[1] = J.c1[2] * f2_bcast(g); out.a[2] = f2_bcast(F[8] * g);
}
// f = I a + v x* (I v) (rnea_body_force on sample pairs)
DRM_HD void rnea2_body_force(float m, const float *mc, const float *Io, const Motion2 &N, Force2 &out) {
    f2 hl[3], ha[3], gl[3], ga[3], x[3], y[3];
    cross2_cv(mc, N.w, x);
#pragma unroll
    for (int i = 0; i < 3; ++i) hl[i] = f2_bcast(m) * N.v[i] - x[i];
    cross2_cv(mc, N.v, x);
#pragma unroll
    for (int r = 0; r < 3; ++r)
        ha[r] = f2_bcast(Io[r * 3 + 0]) * N.w[0] + f2_bcast(Io[r * 3 + 1]) * N.w[1] + f2_bcast(Io[r * 3 + 2]) * N.w[2] + x[r];
    cross2_cv(mc, N.al, x);
#pragma unroll
    for (int i = 0; i < 3; ++i) gl[i] = f2_bcast(m) * N.a[i] - x[i];
    cross2_cv(mc, N.a, x);
#pragma unroll
    for (int r = 0; r < 3; ++r)
        ga[r] = f2_bcast(Io[r * 3 + 0]) * N.al[0] + f2_bcast(Io[r * 3 + 1]) * N.al[1] + f2_bcast(Io[r * 3 + 2]) * N.al[2] + x[r];
    cross2_vv(N.w, hl, x);
#pragma unroll
    for (int i = 0; i < 3; ++i) out.f[i] = gl[i] + x[i];
    cross2_vv(N.w, ha, x);
    cross2_vv(N.v, hl, y);
#pragma unroll
    for (int i = 0; i < 3; ++i) out.n[i] = ga[i] + (x[i] + y[i]);
}
//   row(k) -> op k's constant row;  cs / sn / qd / qdd / tau: [d] = (sample A, sample B)
//   fput(k, Force2) / fget(k, Force2&): body forces of links 0 .. LINKS-2 between the sweeps (the last link's is consumed
//   the moment it exists and never parked)
//   KEEP2: that many more of the last links keep their forces in registers (parked: links 0 .. LINKS-2-KEEP2)
#ifndef DRM_RNEA2_KEEP
#define DRM_RNEA2_KEEP 5
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DRM_RNEA2_NO_FENCE)
#define DRM_RNEA2_LINK_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define DRM_RNEA2_LINK_FENCE() ((void)0)
#endif
// PREF: the constants of the next link are read one link ahead into a second register set (see rnea_chain_trig)
template <int LINKS, int NJ, int KEEP2 = DRM_RNEA2_KEEP, bool PREF = false, class ROW, class FPUT, class FGET>
DRM_HD void rnea_chain2_trig(ROW row, bool gravity, bool damping, const f2 (&cs)[NJ], const f2 (&sn)[NJ], const f2 (&qd)[NJ],
                             const f2 (&qdd)[NJ], f2 (&tau)[NJ], FPUT fput, FGET fget) {
    static_assert(LINKS >= NJ && LINKS >= 1 + KEEP2, "moving joints first, then fixed links"); // (KEEP2 = LINKS - 1: nothing parked)
    constexpr int PARKED = LINKS - 1 - KEEP2;
    Motion2 cur;
    Force2 tot;
    Force2 kept[KEEP2 > 0 ? KEEP2 : 1];
    float buf[2][PREF ? RNEA_ROW_FLOATS : 1];
    if constexpr (PREF) rnea_row_copy(row(0), buf[0]);
#pragma unroll
    for (int k = 0; k < LINKS; ++k) {
        DRM_RNEA2_LINK_FENCE();
        if constexpr (PREF) {
            if (k + 1 < LINKS) rnea_row_copy(row(k + 1), buf[(k + 1) & 1]);
            DRM_RNEA2_LINK_FENCE();
        }
        const float *of = PREF ? buf[k & 1] : row(k);
        const OpFT o = load_ft(of);
        Joint2 J;
        if (k < NJ) joint2_moving(o.F, cs[k], sn[k], J);
        else joint2_fixed(o.F, J);
        const f2 wj = k < NJ ? qd[k] : f2_bcast(0.0f), aj = k < NJ ? qdd[k] : f2_bcast(0.0f);
        if (k == 0) rnea2_first_motion(J, o.F, gravity ? 9.81f : 0.0f, wj, aj, cur);
        else rnea2_link_motion(J, o.F, o.t, wj, aj, cur, cur);
        Force2 fk;
        rnea2_body_force(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, cur, fk);
        if (k < PARKED) fput(k, fk);
        else if (k < LINKS - 1) kept[k - PARKED] = fk;
        else tot = fk;
    }
    float back[2][PREF ? DRM_OPF_FT_FLOATS : 1], damp2[2];
    Force2 f2buf[PREF ? 2 : 1];
    if constexpr (PREF) {
        rnea_row_copy(row(LINKS - 1), back[(LINKS - 1) & 1]);
        damp2[(LINKS - 1) & 1] = row(LINKS - 1)[DRM_OPF_DAMP];
    }
#pragma unroll
    for (int k = LINKS - 1; k >= 0; --k) {
        DRM_RNEA2_LINK_FENCE();
        if constexpr (PREF) {
            if (k > 0) {
                rnea_row_copy(row(k - 1), back[(k - 1) & 1]);
                damp2[(k - 1) & 1] = row(k - 1)[DRM_OPF_DAMP];
                if (k - 1 < PARKED) fget(k - 1, f2buf[(k - 1) & 1]);
            }
            DRM_RNEA2_LINK_FENCE();
        }
        const float *of = PREF ? back[k & 1] : row(k);
        if (k < LINKS - 1) {
            Force2 fk;
            if (k < PARKED) {
                if constexpr (PREF) fk = f2buf[k & 1];
                else fget(k, fk);
            } else {
                fk = kept[k - PARKED];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) { tot.f[i] += fk.f[i]; tot.n[i] += fk.n[i]; }
        }
        if (k < NJ) {
            if constexpr (PREF) tau[k] = damping ? tot.n[2] + f2_bcast(damp2[k & 1]) * qd[k] : tot.n[2];
            else tau[k] = damping ? tot.n[2] + f2_bcast(of[DRM_OPF_DAMP]) * qd[k] : tot.n[2];
        }
        if (k > 0) {
            const OpFT o = load_ft(of);
            Joint2 J;
            if (k < NJ) joint2_moving(o.F, cs[k], sn[k], J);
            else joint2_fixed(o.F, J);
            Force2 up;
            joint2_N(J, o.F, tot.f, up.f);
            joint2_N(J, o.F, tot.n, up.n);
            f2 x[3];
            cross2_cv(o.t, up.f, x);
#pragma unroll
            for (int i = 0; i < 3; ++i) up.n[i] += x[i];
            tot = up;
        }
    }
}
// world pose of the last link of a chain, two samples per lane: R = R_p J, p = R_p t + p_p per link (robot_model.py:186,
// spatial_vector_algebra.py:98-103), in the association order of the one-sample pair form (fk_chain_pairs_trig /
// compose_pairs) so that the fused FK + RNEA launch returns what compute_forward_kinematics returns, bit for bit.
struct Pose2 {
    f2 R[9], p[3];
};
template <int CAP, int NJ, bool PREF = false, class ROW>
DRM_HD void fk_chain2_trig(ROW row, const f2 (&cs)[NJ], const f2 (&sn)[NJ], Pose2 &ee) {
    float buf[2][PREF ? DRM_OPF_FT_FLOATS : 1];
    if (PREF) rnea_row_copy(row(0), buf[0]);
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        DRM_RNEA2_LINK_FENCE();
        if (PREF) {
            if (k + 1 < CAP) rnea_row_copy(row(k + 1), buf[(k + 1) & 1]);
            DRM_RNEA2_LINK_FENCE();
        }
        const OpFT o = load_ft(PREF ? buf[k & 1] : row(k));
        f2 J0[3], J1[3]; // columns 0 and 1 of J = F Rot_z(q) (joint_pairs: f01 c + (f1, f0) (s, -s)); column 2 is F[:, 2]
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (k < NJ) {
                J0[i] = f2_bcast(o.F[i * 3 + 0]) * cs[k] + f2_bcast(o.F[i * 3 + 1]) * sn[k];
                J1[i] = f2_bcast(o.F[i * 3 + 1]) * cs[k] + f2_bcast(o.F[i * 3 + 0]) * -sn[k];
            } else {
                J0[i] = f2_bcast(o.F[i * 3 + 0]);
                J1[i] = f2_bcast(o.F[i * 3 + 1]);
            }
        }
        if (k == 0) { // the parent is the identity root: R = J, p = t exactly
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ee.R[c * 3 + 0] = J0[c]; ee.R[c * 3 + 1] = J1[c]; ee.R[c * 3 + 2] = f2_bcast(o.F[c * 3 + 2]);
                ee.p[c] = f2_bcast(o.t[c]);
            }
        } else {
            Pose2 n;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f2 r0 = ee.R[c * 3 + 0], r1 = ee.R[c * 3 + 1], r2 = ee.R[c * 3 + 2];
                n.R[c * 3 + 0] = r0 * J0[0] + r1 * J0[1] + r2 * J0[2];
                n.R[c * 3 + 1] = r0 * J1[0] + r1 * J1[1] + r2 * J1[2];
                n.R[c * 3 + 2] = r0 * f2_bcast(o.F[2]) + r1 * f2_bcast(o.F[5]) + r2 * f2_bcast(o.F[8]);
                f2 b = r0 * f2_bcast(o.t[0]) + r1 * f2_bcast(o.t[1]) + r2 * f2_bcast(o.t[2]);
                pin2(b); // the translation joins AFTER the rounded sum, as in compose_pairs (b[1] += p)
                n.p[c] = b + ee.p[c];
            }
            ee = n;
        }
    }
}
// cos / sin of the NJ joint angles of two samples
template <int NJ>
DRM_HD void chain_trig2(const f2 (&q)[NJ], f2 (&cs)[NJ], f2 (&sn)[NJ]) {
    bool big = false;
#pragma unroll
    for (int d = 0; d < NJ; ++d) big = big || !(fabsf(q[d][0]) <= SINCOS_PAIR_MAX_ARG) || !(fabsf(q[d][1]) <= SINCOS_PAIR_MAX_ARG);
    if (DRM_WAVE_ANY(big)) { // rare; wave-uniform
#pragma unroll
        for (int d = 0; d < NJ; ++d) {
            float s0, c0, s1, c1;
            sincos_f(q[d][0], s0, c0);
            sincos_f(q[d][1], s1, c1);
            sn[d] = f2_make(s0, s1); cs[d] = f2_make(c0, c1);
        }
    } else {
#pragma unroll
        for (int d = 0; d < NJ; ++d) sincos_pair(q[d], sn[d], cs[d]);
    }
}

// ---------------------------------------------------------------------------
// Reverse-mode derivative of the RNEA walk: what torch autograd computes for the reference when a loss on
// compute_inverse_dynamics' torques is back-propagated to the learnable link parameters (robot_model.py:305-375
// with robot_model.py:669-713; examples/learn_dynamics_iiwa.py:49-96) and to q / qd / qdd.
//
// Mathematically four sweeps over the walk
//   A  k up    motion (w, al, v, a) of every link and its body force                      [= the RNEA forward sweep]
//   B  k down  total force tot_k = f_k + sum of the children's forces moved up             [= the RNEA backward sweep]
//   C  k up    adjoint of B:  tbar_k = J^T-transformed tbar_parent + gtau_k e_(ang z)
//   D  k down  adjoint of A:  motion adjoints from the body force and the children, then the adjoints of the
//              joint transform (J, t) from both sweeps, of the constants (m, mc, Io, damping) and of q, qd, qdd
// run as TWO: A and C together on the way up; D on the way down with B folded in (the body force is recomputed from the
// motion, the sub-tree's force travels with the walk) and every parent's motion / tbar recovered from its child's — see
// rnea_backward_walk (any tree) and rnea_backward_chain (serial arms) below.
// Layout of a parked link record (floats): 0..11 motion (w, v, al, a) and 18..23 tbar of LEAF links, 24..25 (cos q, sin q)
// of every link;  of a slot record: 0..11 motion, 12..17 total-force accumulator, 18..23 tbar, 24..35 motion-adjoint accumulator.
//   park(k, off, v, n) / unpark(k, off, v, n)                       per-link records
//   slot_put / slot_get / slot_add / slot_take(s, off, v, n)        branch-point records (take = read and zero)
//   gtau(d) -> dL/dtau of DoF d;   gout(d, gq, gqd, gqdd);   param_out(k, g[DRM_OPF_STRIDE]) for ops in param_mask
// ---------------------------------------------------------------------------
DRM_HD void motion_to_floats(const Motion &M, float *v) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { v[i] = M.wa[i][0]; v[3 + i] = M.va[i][0]; v[6 + i] = M.wa[i][1]; v[9 + i] = M.va[i][1]; }
}
DRM_HD void motion_from_floats(const float *v, Motion &M) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { M.wa[i] = f2_make(v[i], v[6 + i]); M.va[i] = f2_make(v[3 + i], v[9 + i]); }
}
DRM_HD void add_cross(float *acc, const float *a, const float *b) { // acc += a x b
    acc[0] += a[1] * b[2] - a[2] * b[1];
    acc[1] += a[2] * b[0] - a[0] * b[2];
    acc[2] += a[0] * b[1] - a[1] * b[0];
}
DRM_HD void sub_cross(float *acc, const float *a, const float *b) { // acc -= a x b
    acc[0] -= a[1] * b[2] - a[2] * b[1];
    acc[1] -= a[2] * b[0] - a[0] * b[2];
    acc[2] -= a[0] * b[1] - a[1] * b[0];
}

// Adjoint of ONE link of the RNEA: given the link's motion `mo` (w, v, al, a), its parent's motion `par`, the adjoint
// of its total force `fb` (= tbar_k), the motion adjoint `mb` arriving from its children (updated in place to the
// link's full motion adjoint), and for the force transform of the backward sweep the parent's tbar `ub` and the
// link's total force `tot`: the adjoints of the parent's motion (pb), of the joint value / rate / acceleration
// (gq, wjb, ajb) and — only when `want_params`, i.e. for the few links with learnable constants — of the joint transform
// (Jb, tb) and of the inertial constants (gm, gmc, gIo).
//
// gq without Jb.  With J = F Rot_z(q), dL/dq = sum_r Jb[r][0] J[r][1] - Jb[r][1] J[r][0], and Jb is a sum of outer
// products y (x) x; each contributes x[0] (J^T y)[1] - x[1] (J^T y)[0], and every J^T y is a quantity the forward sweep
// produced: J^T (Pa + Pal x t) = a - (v_y wj, -v_x wj, 0), J^T Pal = al - (w_y wj, -w_x wj, aj), J^T (Pv + Pw x t) = v,
// J^T Pw = w - wj e_z, J^T (ubar.lin + ubar.ang x t) = tbar.lin, J^T ubar.ang = tbar.ang (- gtau e_z).  Eight dot-product
// terms of two factors instead of 63 multiply-adds for Jb — on every link, learnable or not.  (A prismatic joint's gq is
// tb . F e_z, so tb is formed for it as well.)
struct LinkAdjoint {
    float pb[12], Jb[9], tb[3], gm, gmc[3], gIo[9], wjb, ajb, gq;
};
DRM_HD void rnea_link_adjoint(float m, const float *mc, const float *Io, const float *J, const float *t, float wj,
                              const float *mo, const float *fb, const float *par, float *mb, const float *ub,
                              const float *tot, bool has_parent, LinkAdjoint &out, bool prismatic = false,
                              bool want_params = true) {
    float *wb = mb, *vb = mb + 3, *alb = mb + 6, *ab = mb + 9;
    const float *w = mo, *v = mo + 3, *al = mo + 6, *a = mo + 9;
    const float *fl = fb, *fa = fb + 3; // adjoint of this link's body force = tbar_k
    // body force: hl = m v - mc x w, ha = Io w + mc x v, gl = m a - mc x al, ga = Io al + mc x a,
    //             f.lin = gl + w x hl,  f.ang = ga + w x ha + v x hl
    float hl[3], ha[3], x[3];
    cross3(mc, w, x);
    hl[0] = m * v[0] - x[0]; hl[1] = m * v[1] - x[1]; hl[2] = m * v[2] - x[2];
    mat_vec(Io, w, ha);
    add_cross(ha, mc, v);
    float hlb[3] = {0, 0, 0}, hab[3] = {0, 0, 0};
    add_cross(hlb, fl, w); add_cross(hlb, fa, v);
    add_cross(hab, fa, w);
    add_cross(wb, hl, fl); add_cross(wb, ha, fa);
    add_cross(vb, hl, fa);
    vb[0] += m * hlb[0]; vb[1] += m * hlb[1]; vb[2] += m * hlb[2];
    sub_cross(wb, hlb, mc);
    matT_vec(Io, hab, x);
    wb[0] += x[0]; wb[1] += x[1]; wb[2] += x[2];
    add_cross(vb, hab, mc);
    ab[0] += m * fl[0]; ab[1] += m * fl[1]; ab[2] += m * fl[2];
    sub_cross(alb, fl, mc);
    matT_vec(Io, fa, x);
    alb[0] += x[0]; alb[1] += x[1]; alb[2] += x[2];
    add_cross(ab, fa, mc);
    if (want_params) {
        float gm = hlb[0] * v[0] + hlb[1] * v[1] + hlb[2] * v[2];
        gm += fl[0] * a[0] + fl[1] * a[1] + fl[2] * a[2];
        out.gm = gm;
        float *gmc = out.gmc, *gIo = out.gIo;
        gmc[0] = gmc[1] = gmc[2] = 0.0f;
        sub_cross(gmc, w, hlb);
        add_cross(gmc, v, hab);
        sub_cross(gmc, al, fl);
        add_cross(gmc, a, fa);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) gIo[r * 3 + c] = hab[r] * w[c] + fa[r] * al[c];
    }

    // link motion from the parent's: adjoint
    const float *Pw = par, *Pv = par + 3, *Pal = par + 6, *Pa = par + 9;
    float *pb = out.pb, *Jb = out.Jb, *tbr = out.tb, ya[3], yv[3], yb[3];
    float wjb = 0.0f, ajb = 0.0f, gq = 0.0f;
    const bool want_tb = want_params || prismatic;
    float *Pwb = pb, *Pvb = pb + 3, *Palb = pb + 6, *Pab = pb + 9;
    // a = J^T (Pa + Pal x t) + (v_y wj, -v_x wj, 0)            [prismatic: ... + aj e_z + (w_y wj, -w_x wj, 0)]
    mat_vec(J, ab, yb);
    if (prismatic) {
        ajb += ab[2];
        wb[1] += ab[0] * wj; wb[0] -= ab[1] * wj;
        wjb += ab[0] * w[1] - ab[1] * w[0];
    } else {
        gq += ab[0] * (a[1] + v[0] * wj) - ab[1] * (a[0] - v[1] * wj);
        vb[1] += ab[0] * wj; vb[0] -= ab[1] * wj;
        wjb += ab[0] * v[1] - ab[1] * v[0];
    }
    Pab[0] = yb[0]; Pab[1] = yb[1]; Pab[2] = yb[2];
    cross3(t, yb, Palb);
    if (want_tb) cross3(yb, Pal, tbr);
    // al = J^T Pal + aj e_z + (w_y wj, -w_x wj, 0)             [prismatic: al = J^T Pal]
    mat_vec(J, alb, x);
    Palb[0] += x[0]; Palb[1] += x[1]; Palb[2] += x[2];
    if (!prismatic) {
        gq += alb[0] * (al[1] + w[0] * wj) - alb[1] * (al[0] - w[1] * wj);
        ajb += alb[2];
        wb[1] += alb[0] * wj; wb[0] -= alb[1] * wj;
        wjb += alb[0] * w[1] - alb[1] * w[0];
    }
    // v = J^T (Pv + Pw x t)                                    [prismatic: ... + wj e_z]
    mat_vec(J, vb, yb);
    if (prismatic) wjb += vb[2];
    else gq += vb[0] * v[1] - vb[1] * v[0];
    Pvb[0] = yb[0]; Pvb[1] = yb[1]; Pvb[2] = yb[2];
    cross3(t, yb, Pwb);
    if (want_tb) add_cross(tbr, yb, Pw);
    // w = J^T Pw + wj e_z                                      [prismatic: w = J^T Pw]
    mat_vec(J, wb, x);
    Pwb[0] += x[0]; Pwb[1] += x[1]; Pwb[2] += x[2];
    if (!prismatic) {
        gq += wb[0] * w[1] - wb[1] * w[0];
        wjb += wb[2];
    }
    // the force transform of the backward sweep: up.lin = J tot.lin, up.ang = J tot.ang + t x (J tot.lin)
    if (has_parent && !prismatic) gq += tot[0] * fb[1] - tot[1] * fb[0] + tot[3] * fb[4] - tot[4] * fb[3];
    if (has_parent && want_tb) {
        float L[3];
        mat_vec(J, tot, L);
        add_cross(tbr, L, ub + 3);
    }
    if (want_params) {
        // Jb = (Pa + Pal x t) (x) ab + Pal (x) alb + (Pv + Pw x t) (x) vb + Pw (x) wb  [+ the force transform's terms]
        ya[0] = Pa[0]; ya[1] = Pa[1]; ya[2] = Pa[2];
        add_cross(ya, Pal, t);
        yv[0] = Pv[0]; yv[1] = Pv[1]; yv[2] = Pv[2];
        add_cross(yv, Pw, t);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Jb[r * 3 + c] = ya[r] * ab[c] + Pal[r] * alb[c] + yv[r] * vb[c] + Pw[r] * wb[c];
        if (has_parent) {
            float Lb[3];
            Lb[0] = ub[0]; Lb[1] = ub[1]; Lb[2] = ub[2];
            add_cross(Lb, ub + 3, t);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Jb[r * 3 + c] += Lb[r] * tot[c] + ub[3 + r] * tot[3 + c];
        }
    }
    if (prismatic) gq = tbr[0] * J[2] + tbr[1] * J[5] + tbr[2] * J[8]; // t = trans + F e_z q
    out.wjb = wjb;
    out.ajb = ajb;
    out.gq = gq;
}

// ---------------------------------------------------------------------------
// The same adjoint on PACKED pairs, for the straight-line arm form (rnea_backward_chain): the forward sweeps pair
// velocity- with acceleration-like quantities, (w, al), (v, a), because both go through the same linear maps — and so do
// their adjoints, (wb, alb), (vb, ab), and the adjoints of the momentum-like pairs (h, g): X = (hlb, fl), Y = (hab, fa).
// Every product with J, Io, mc x and t x below handles both halves at once; what stays scalar are the products of two
// per-sample vectors (the crosses with w, v, fl, fa) and the closed-form dL/dq terms.
//   M   the link's motion,  Pm  its parent's,  T[i] = (tbar.lin_i, tbar.ang_i),  tot = the sub-tree's total force
//   hgl / hga   the (h, g) pairs of the link's body force (rnea_body_force_hg)
//   B   in: motion adjoint arriving from the children; out: the link's full motion adjoint (the extras read it)
//   pb  motion adjoint handed to the parent;  YV = (J vb, J ab) pairs, kept for the extras
// ---------------------------------------------------------------------------
struct LinkAdjointP {
    Motion pb;
    f2 YV[3];
    float hlb[3], hab[3], wjb, ajb, gq;
};
DRM_HD void rnea_link_adjoint_packed(float m, const float *mc, const float *Io, const float *J, const float *t, float wj,
                                     const Motion &M, const f2 (&hgl)[3], const f2 (&hga)[3], const f2 (&T)[3],
                                     const Force &tot, bool has_parent, Motion &B, LinkAdjointP &out) {
    const float w[3] = {M.wa[0][0], M.wa[1][0], M.wa[2][0]}, v[3] = {M.va[0][0], M.va[1][0], M.va[2][0]};
    const float al[3] = {M.wa[0][1], M.wa[1][1], M.wa[2][1]}, a[3] = {M.va[0][1], M.va[1][1], M.va[2][1]};
    const float fl[3] = {T[0][0], T[1][0], T[2][0]}, fa[3] = {T[0][1], T[1][1], T[2][1]};
    const float hl[3] = {hgl[0][0], hgl[1][0], hgl[2][0]}, ha[3] = {hga[0][0], hga[1][0], hga[2][0]};
    // f.lin = gl + w x hl,  f.ang = ga + w x ha + v x hl:  adjoints of h, and of (w, v) through the crosses
    float *hlb = out.hlb, *hab = out.hab, xw[3] = {0, 0, 0}, xv[3];
    cross3(fl, w, hlb); add_cross(hlb, fa, v);
    cross3(fa, w, hab);
    add_cross(xw, hl, fl); add_cross(xw, ha, fa);
    cross3(hl, fa, xv);
#pragma unroll
    for (int i = 0; i < 3; ++i) { B.wa[i][0] += xw[i]; B.va[i][0] += xv[i]; }
    // (hl, gl) = m (v, a) - mc x (w, al);  (ha, ga) = Io (w, al) + mc x (v, a)
    f2 X[3], Y[3], c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { X[i] = f2_make(hlb[i], fl[i]); Y[i] = f2_make(hab[i], fa[i]); }
#pragma unroll
    for (int i = 0; i < 3; ++i) B.va[i] += f2_bcast(m) * X[i];
    cross3_ps(X, mc, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) B.wa[i] -= c[i];
    matT_vec_p(Io, Y, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) B.wa[i] += c[i];
    cross3_ps(Y, mc, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) B.va[i] += c[i];
    // the link's motion from its parent's (revolute joint about +z):
    //   (w, al) = J^T (Pw, Pal) + (wj, aj) e_z + (0, (w_y wj, -w_x wj, 0));  (v, a) = J^T ((Pv, Pa) + (Pw, Pal) x t) + (0, (v_y wj, -v_x wj, 0))
    const float ab0 = B.va[0][1], ab1 = B.va[1][1], alb0 = B.wa[0][1], alb1 = B.wa[1][1];
    float gq = ab0 * (a[1] + v[0] * wj) - ab1 * (a[0] - v[1] * wj) + alb0 * (al[1] + w[0] * wj) - alb1 * (al[0] - w[1] * wj);
    float wjb = ab0 * v[1] - ab1 * v[0] + alb0 * w[1] - alb1 * w[0];
    B.va[1][0] += ab0 * wj; B.va[0][0] -= ab1 * wj;
    B.wa[1][0] += alb0 * wj; B.wa[0][0] -= alb1 * wj;
    gq += B.va[0][0] * v[1] - B.va[1][0] * v[0] + B.wa[0][0] * w[1] - B.wa[1][0] * w[0];
    wjb += B.wa[2][0];
    out.ajb = B.wa[2][1];
    mat_vec_p(J, B.va, out.YV);                 // (J vb, J ab) = (Pvb, Pab)
    cross3_sp(t, out.YV, out.pb.wa);            // (t x J vb, t x J ab) into (Pwb, Palb)
    mat_vec_p(J, B.wa, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) { out.pb.wa[i] += c[i]; out.pb.va[i] = out.YV[i]; }
    if (has_parent) { // the force transform of the backward sweep (see rnea_link_adjoint)
        const f2 d = tot.la[0] * T[1] - tot.la[1] * T[0];
        gq += d[0] + d[1];
    }
    out.wjb = wjb;
    out.gq = gq;
}
// The adjoints of a LEARNABLE link's constants, from what rnea_link_adjoint_packed left (cold path: a few links at most):
// the op row's gradient gr[DRM_OPF_STRIDE] apart from the damping entry.  ub = the parent's tbar (lin, ang).
DRM_HD void rnea_link_param_adjoint(float m, const float *mc, const float *Io, const float *J, const float *t, float c_, float s_,
                                    const Motion &M, const Motion &Pm, const f2 (&T)[3], const float *ub, const Force &tot,
                                    bool has_parent, const Motion &B, const LinkAdjointP &A, float *gr) {
    const float w[3] = {M.wa[0][0], M.wa[1][0], M.wa[2][0]}, v[3] = {M.va[0][0], M.va[1][0], M.va[2][0]};
    const float al[3] = {M.wa[0][1], M.wa[1][1], M.wa[2][1]}, a[3] = {M.va[0][1], M.va[1][1], M.va[2][1]};
    const float Pw[3] = {Pm.wa[0][0], Pm.wa[1][0], Pm.wa[2][0]}, Pv[3] = {Pm.va[0][0], Pm.va[1][0], Pm.va[2][0]};
    const float Pal[3] = {Pm.wa[0][1], Pm.wa[1][1], Pm.wa[2][1]}, Pa[3] = {Pm.va[0][1], Pm.va[1][1], Pm.va[2][1]};
    const float fl[3] = {T[0][0], T[1][0], T[2][0]}, fa[3] = {T[0][1], T[1][1], T[2][1]};
    const float wb[3] = {B.wa[0][0], B.wa[1][0], B.wa[2][0]}, vb[3] = {B.va[0][0], B.va[1][0], B.va[2][0]};
    const float alb[3] = {B.wa[0][1], B.wa[1][1], B.wa[2][1]}, ab[3] = {B.va[0][1], B.va[1][1], B.va[2][1]};
    const float yv[3] = {A.YV[0][0], A.YV[1][0], A.YV[2][0]}, ya[3] = {A.YV[0][1], A.YV[1][1], A.YV[2][1]}; // J vb, J ab
    const float tl[3] = {tot.la[0][0], tot.la[1][0], tot.la[2][0]}, ta[3] = {tot.la[0][1], tot.la[1][1], tot.la[2][1]};
    const float *hlb = A.hlb, *hab = A.hab;
#pragma unroll
    for (int i = 0; i < DRM_OPF_STRIDE; ++i) gr[i] = 0.0f;
    gr[DRM_OPF_MASS] = hlb[0] * v[0] + hlb[1] * v[1] + hlb[2] * v[2] + fl[0] * a[0] + fl[1] * a[1] + fl[2] * a[2];
    float gmc[3] = {0, 0, 0};
    sub_cross(gmc, w, hlb); add_cross(gmc, v, hab); sub_cross(gmc, al, fl); add_cross(gmc, a, fa);
    float Jb[9], tbr[3], y1[3], y2[3];
    y1[0] = Pa[0]; y1[1] = Pa[1]; y1[2] = Pa[2];
    add_cross(y1, Pal, t);
    y2[0] = Pv[0]; y2[1] = Pv[1]; y2[2] = Pv[2];
    add_cross(y2, Pw, t);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) Jb[r * 3 + cc] = y1[r] * ab[cc] + Pal[r] * alb[cc] + y2[r] * vb[cc] + Pw[r] * wb[cc];
    cross3(ya, Pal, tbr);
    add_cross(tbr, yv, Pw);
    if (has_parent) {
        float Lb[3], L[3];
        Lb[0] = ub[0]; Lb[1] = ub[1]; Lb[2] = ub[2];
        add_cross(Lb, ub + 3, t);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) Jb[r * 3 + cc] += Lb[r] * tl[cc] + ub[3 + r] * ta[cc];
        mat_vec(J, tl, L);
        add_cross(tbr, L, ub + 3);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        gr[DRM_OPF_FIJ(r, 0)] = Jb[r * 3 + 0] * c_ - Jb[r * 3 + 1] * s_;
        gr[DRM_OPF_FIJ(r, 1)] = Jb[r * 3 + 0] * s_ + Jb[r * 3 + 1] * c_;
        gr[DRM_OPF_FIJ(r, 2)] = Jb[r * 3 + 2];
        gr[DRM_OPF_TI(r)] = tbr[r];
        gr[DRM_OPF_MCOM + r] = gmc[r];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) gr[DRM_OPF_IO + r * 3 + cc] = hab[r] * w[cc] + fa[r] * al[cc];
    }
}

// Parent's motion from a link's own (the inverse of rnea_link_motion / motion_step; J is orthogonal):
//   revolute   (w_p, al_p) = J ((w, al) - (wj, aj) e_z - (0, (w_y wj, -w_x wj, 0))),   (v_p, a_p) = J ((v, a) - (0, (v_y wj, -v_x wj, 0))) - (w_p, al_p) x t
//   prismatic  (w_p, al_p) = J (w, al),   (v_p, a_p) = J ((v, a) - (wj, aj) e_z - (0, (w_y wj, -w_x wj, 0))) - (w_p, al_p) x t
DRM_HD void motion_parent(const float *J, const float *t, float wj, float aj, bool prismatic, const Motion &M, Motion &P) {
    const float w0 = M.wa[0][0], w1 = M.wa[1][0], v0 = M.va[0][0], v1 = M.va[1][0];
    f2 xw[3] = {M.wa[0], M.wa[1], M.wa[2]}, xv[3] = {M.va[0], M.va[1], M.va[2]}, y[3], c[3];
    if (prismatic) {
        xv[0][1] -= w1 * wj; xv[1][1] += w0 * wj; xv[2] -= f2_make(wj, aj);
    } else {
        xw[0][1] -= w1 * wj; xw[1][1] += w0 * wj; xw[2] -= f2_make(wj, aj);
        xv[0][1] -= v1 * wj; xv[1][1] += v0 * wj;
    }
    mat_vec_p(J, xw, P.wa);
    mat_vec_p(J, xv, y);
    cross3_ps(P.wa, t, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) P.va[i] = y[i] - c[i];
}
// tbar of a link from its parent's, T[i] = (lin_i, ang_i):  tbar = (J^T (ubar.lin + ubar.ang x t), J^T ubar.ang)  [+ the joint's
// own dL/dtau on its torque component, added by the caller] — and the inverse, the parent's from the link's (torque
// component removed by the caller first)
DRM_HD void tbar_child(const float *J, const float *t, const f2 (&U)[3], f2 (&T)[3]) {
    const float ua[3] = {U[0][1], U[1][1], U[2][1]};
    f2 x[3] = {U[0], U[1], U[2]};
    x[0][0] += ua[1] * t[2] - ua[2] * t[1];
    x[1][0] += ua[2] * t[0] - ua[0] * t[2];
    x[2][0] += ua[0] * t[1] - ua[1] * t[0];
    matT_vec_p(J, x, T);
}
DRM_HD void tbar_parent(const float *J, const float *t, const f2 (&T)[3], f2 (&U)[3]) {
    mat_vec_p(J, T, U);
    const float ua[3] = {U[0][1], U[1][1], U[2][1]};
    U[0][0] -= ua[1] * t[2] - ua[2] * t[1];
    U[1][0] -= ua[2] * t[0] - ua[0] * t[2];
    U[2][0] -= ua[0] * t[1] - ua[1] * t[0];
}

// The walk over ANY tree: two LOOPS over the n_ops links (no identity padding, one control word decoded per iteration,
// nothing indexed by a compile-time op number — the adjoint of one link is several hundred instructions, so a
// straight-line walk of 24 or 32 links fits neither the instruction cache nor the register file).
//   up    motions (A) and force adjoints (C) together; what is kept per link is (cos, sin) of its joint angle, and for
//         LEAF links (no child follows them in the walk) their motion and tbar.  Branch points save both in their slot.
//   down  D with B folded in, as in rnea_backward_chain: a link that was just handed its child's view of it (CHILD_IS_NEXT)
//         continues from the parent motion / tbar / motion adjoint / force that child recovered; a leaf starts from its
//         parked record.  The body force is recomputed from the motion, the sub-tree's total force travels down with the
//         walk (through the slots at branch points).
// Per link only 2 floats are parked, per leaf 18 more (the earlier four-sweep form parked 26 per link and read most of
// them three times — for a hand that traffic, in HBM because it did not fit LDS, was the kernel's whole run time).
//   ctl = the control-word field of the int table (DRM_OPI_CTRL)
//   ops: the static prefix [0, p_end) is replayed on the way up (motions only matter), then ops [a, b) — one SEGMENT of the
//        walk (independent sub-trees hanging off the prefix; a wavefront each in the kernel); the way down visits [a, b)
//        only: nothing is handed to a prefix op (they have no DoF; the caller must not ask for their constants' gradients).
//        The whole walk in one go: p_end = 0, a = 0, b = n_ops.
//   park / unpark(k, off, v, n): record of link k — floats 0..11 motion (w, v, al, a), 18..23 tbar (leaves only), 24..25 trig
//   slot records: 0..11 motion, 12..17 total-force accumulator, 18..23 tbar, 24..35 motion-adjoint accumulator
template <class QF, class GT, class PARK, class UNPARK, class SPUT, class SGET, class SADD, class STAKE, class GOUT, class PG>
DRM_HD void rnea_backward_walk(const float *__restrict__ opf, const int32_t *__restrict__ ctl, int p_end, int a, int b, int flags,
                               uint64_t param_mask, bool want_gq, QF qf, GT gtau, PARK park, UNPARK unpark, SPUT slot_put,
                               SGET slot_get, SADD slot_add, STAKE slot_take, GOUT gout, PG param_out) {
    const float g = (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f;
    const bool damping = flags & DRM_RNEA_DAMPING;
    uint32_t prefix_slots = 0; // slots owned by prefix ops: read on the way up, never added to on the way down
    auto tbar_to_floats = [](const f2 (&T)[3], float *v) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[i] = T[i][0]; v[3 + i] = T[i][1]; }
    };
    auto tbar_from_floats = [](const float *v, f2 (&T)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) T[i] = f2_make(v[i], v[3 + i]);
    };
    // ---- up: motions and force adjoints ---------------------------------------------------------------------------
    {
        Motion cur;
        f2 T[3];
        motion_root(cur, g);
#pragma unroll
        for (int i = 0; i < 3; ++i) T[i] = f2_bcast(0.0f);
#pragma unroll 1
        for (int k = (p_end > 0 ? 0 : a); k < b; k = (k + 1 == p_end ? a : k + 1)) {
            const float *of = opf + k * DRM_OPF_STRIDE;
            const int c = ctl[k];
            const int dof = ctl_field(c, DRM_OPI_DOF), src = ctl_field(c, DRM_OPI_SRC), save = ctl_field(c, DRM_OPI_SAVE);
            const bool pris = ctl_prismatic(c);
            const bool mine = k >= a; // (a prefix op: nothing is parked for it)
            if (!mine && save >= 0) prefix_slots |= 1u << save;
            float wj = 0.0f, aj = 0.0f, J[9], t[3], rec[12], trig[2] = {1.0f, 0.0f};
            if (dof >= 0) {
                float q;
                qf(dof, q, wj, aj);
                if (pris) trig[1] = q;
                else sincos_one(q, trig[1], trig[0]);
            }
            if (mine) park(k, 24, trig, 2);
            const OpFT o = load_ft(of);
            joint_transform(o, dof >= 0, pris, trig[1], trig[0], trig[1], J, t);
            if (src == DRM_SRC_ROOT) {
                motion_root(cur, g);
#pragma unroll
                for (int i = 0; i < 3; ++i) T[i] = f2_bcast(0.0f);
            }
            if (src >= 0) {
                slot_get(src, 0, rec, 12);
                motion_from_floats(rec, cur);
                slot_get(src, 18, rec, 6);
                tbar_from_floats(rec, T);
            }
            motion_step(J, t, wj, aj, pris, cur, cur);
            if (src != DRM_SRC_ROOT) {
                f2 Tn[3];
                tbar_child(J, t, T, Tn);
#pragma unroll
                for (int i = 0; i < 3; ++i) T[i] = Tn[i];
            }
            if (dof >= 0) T[2][pris ? 0 : 1] += gtau(dof); // tau = S^T f: angular z (revolute), linear z (prismatic)
            const bool leaf = mine && !(ctl_field(c, DRM_OPI_FLAGS) & DRM_FLAG_CHILD_IS_NEXT);
            if (save >= 0 || leaf) {
                float tb[6];
                motion_to_floats(cur, rec);
                tbar_to_floats(T, tb);
                if (save >= 0) { slot_put(save, 0, rec, 12); slot_put(save, 18, tb, 6); }
                if (leaf) { park(k, 0, rec, 12); park(k, 18, tb, 6); }
            }
        }
    }
    // ---- down: adjoints, with the total forces formed on the way ------------------------------------------------------
    {
        Motion M, B, Pm, pbn;       // this link's motion and motion adjoint; its parent's motion; the adjoint handed up
        f2 T[3], U[3];              // this link's tbar, its parent's
        Force carry, up;            // total force of the links below (this link's frame); this sub-tree's, moved up
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Pm.wa[i] = Pm.va[i] = pbn.wa[i] = pbn.va[i] = U[i] = up.la[i] = f2_bcast(0.0f);
        }
#pragma unroll 1
        for (int k = b - 1; k >= a; --k) {
            const float *of = opf + k * DRM_OPF_STRIDE;
            const int c = ctl[k];
            const int dof = ctl_field(c, DRM_OPI_DOF), src = ctl_field(c, DRM_OPI_SRC), save = ctl_field(c, DRM_OPI_SAVE);
            const bool pris = ctl_prismatic(c);
            const bool chained = ctl_field(c, DRM_OPI_FLAGS) & DRM_FLAG_CHILD_IS_NEXT;
            if (chained) { // op k + 1 is a child of this link: continue from what it recovered / handed up
                M = Pm;
                B = pbn;
                carry = up;
#pragma unroll
                for (int i = 0; i < 3; ++i) T[i] = U[i];
            } else {       // a leaf: its parked record, nothing below it
                float rec[12], tb[6];
                unpark(k, 0, rec, 12);
                motion_from_floats(rec, M);
                unpark(k, 18, tb, 6);
                tbar_from_floats(tb, T);
#pragma unroll
                for (int i = 0; i < 3; ++i) { B.wa[i] = B.va[i] = carry.la[i] = f2_bcast(0.0f); }
            }
            if (save >= 0) { // what the children that do not follow this link directly left in its slot
                float x12[12], x6[6];
                slot_take(save, 24, x12, 12);
                slot_take(save, 12, x6, 6);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    B.wa[i] += f2_make(x12[i], x12[6 + i]);
                    B.va[i] += f2_make(x12[3 + i], x12[9 + i]);
                    carry.la[i] += f2_make(x6[i], x6[3 + i]);
                }
            }
            float J[9], t[3], trig[2], wj = 0.0f, aj = 0.0f, qdk = 0.0f;
            {
                const OpFT o = load_ft(of);
                unpark(k, 24, trig, 2);
                joint_transform(o, dof >= 0, pris, trig[1], trig[0], trig[1], J, t);
            }
            if (dof >= 0) { float q; qf(dof, q, wj, aj); qdk = wj; }
            const float gtk = dof >= 0 ? gtau(dof) : 0.0f;
            const bool has_parent = src != DRM_SRC_ROOT;
            if (has_parent) {
                motion_parent(J, t, wj, aj, pris, M, Pm);
                f2 x[3] = {T[0], T[1], T[2]};
                x[2][pris ? 0 : 1] -= gtk;
                tbar_parent(J, t, x, U);
            } else {
                motion_root(Pm, g);
#pragma unroll
                for (int i = 0; i < 3; ++i) U[i] = f2_bcast(0.0f);
            }
            Force tot;
            f2 hgl[3], hga[3];
            rnea_body_force_hg(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, M, tot, hgl, hga);
#pragma unroll
            for (int i = 0; i < 3; ++i) tot.la[i] += carry.la[i];
            const bool learn = (param_mask >> k) & 1u;
            float gq, wjb, ajb;
            if (!pris) {
                LinkAdjointP A;
                rnea_link_adjoint_packed(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, wj, M, hgl, hga, T, tot,
                                         has_parent, B, A);
                gq = A.gq; wjb = A.wjb; ajb = A.ajb;
                pbn = A.pb;
                if (learn) {
                    float gr[DRM_OPF_STRIDE], ub[6];
                    tbar_to_floats(U, ub);
                    rnea_link_param_adjoint(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, trig[0], trig[1], M, Pm, T, ub,
                                            tot, has_parent, B, A, gr);
                    gr[DRM_OPF_DAMP] = damping ? gtk * qdk : 0.0f;
                    param_out(k, gr);
                }
            } else { // a sliding joint: the scalar form (its motion subspace differs; rare)
                float mo[12], mb[12], par[12], fb[6], ub[6], tf[6];
                motion_to_floats(M, mo); motion_to_floats(B, mb); motion_to_floats(Pm, par);
                tbar_to_floats(T, fb); tbar_to_floats(U, ub);
#pragma unroll
                for (int i = 0; i < 3; ++i) { tf[i] = tot.la[i][0]; tf[3 + i] = tot.la[i][1]; }
                LinkAdjoint A;
                rnea_link_adjoint(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, wj, mo, fb, par, mb, ub, tf, has_parent,
                                  A, true, learn);
                gq = A.gq; wjb = A.wjb; ajb = A.ajb;
                motion_from_floats(A.pb, pbn);
                if (learn) { // J = F and t = trans + F e_z q (trig = (1, q))
                    float gr[DRM_OPF_STRIDE];
#pragma unroll
                    for (int i = 0; i < DRM_OPF_STRIDE; ++i) gr[i] = 0.0f;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        gr[DRM_OPF_FIJ(r, 0)] = A.Jb[r * 3 + 0];
                        gr[DRM_OPF_FIJ(r, 1)] = A.Jb[r * 3 + 1];
                        gr[DRM_OPF_FIJ(r, 2)] = A.Jb[r * 3 + 2] + A.tb[r] * trig[1];
                        gr[DRM_OPF_TI(r)] = A.tb[r];
                        gr[DRM_OPF_MCOM + r] = A.gmc[r];
                    }
                    gr[DRM_OPF_MASS] = A.gm;
#pragma unroll
                    for (int i = 0; i < 9; ++i) gr[DRM_OPF_IO + i] = A.gIo[i];
                    gr[DRM_OPF_DAMP] = damping ? gtk * qdk : 0.0f;
                    param_out(k, gr);
                }
            }
            if (want_gq && dof >= 0) gout(dof, gq, wjb + (damping ? of[DRM_OPF_DAMP] * gtk : 0.0f), ajb);
            if (has_parent) {
                rnea_link_force_up(J, t, tot, up); // the sub-tree's force in the parent's frame
                if (src >= 0 && !((prefix_slots >> src) & 1u)) {
                    float x12[12], x6[6];
                    motion_to_floats(pbn, x12);
#pragma unroll
                    for (int i = 0; i < 3; ++i) { x6[i] = up.la[i][0]; x6[3 + i] = up.la[i][1]; }
                    slot_add(src, 24, x12, 12);
                    slot_add(src, 12, x6, 6);
                }
            }
        }
    }
}

// The same walk for a SHORT SERIAL segment with revolute / fixed joints only (ops a .. b - 1, each the child of the one before
// it, at most MAXOPS of them, none a branch point — a finger of a hand): both sweeps unrolled, cos / sin of the joints in
// registers, NOTHING parked: like rnea_backward_chain, the way down starts from the tip's motion and force adjoint and recovers
// every parent's.  The static prefix (fixed joints by construction) is replayed in a loop for its motions and branch-point slots.
// Returns false (nothing done) when the segment is not of that shape — the caller then runs rnea_backward_walk.
template <int MAXOPS, class QF, class GT, class SPUT, class SGET, class GOUT, class PG>
DRM_HD bool rnea_backward_walk_short(const float *__restrict__ opf, const int32_t *__restrict__ ctl, int p_end, int a, int b,
                                     int flags, uint64_t param_mask, bool want_gq, QF qf, GT gtau, SPUT slot_put, SGET slot_get,
                                     GOUT gout, PG param_out) {
    const int len = b - a;
    if (len < 1 || len > MAXOPS) return false;
    int dofs[MAXOPS], src0 = DRM_SRC_ROOT;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < MAXOPS; ++i) {
        dofs[i] = -1;
        if (i < len) {
            const int c = ctl[a + i];
            dofs[i] = ctl_field(c, DRM_OPI_DOF);
            const bool next = ctl_field(c, DRM_OPI_FLAGS) & DRM_FLAG_CHILD_IS_NEXT;
            ok = ok && !ctl_prismatic(c) && ctl_field(c, DRM_OPI_SAVE) < 0 && next == (i + 1 < len) &&
                 (i == 0 || ctl_field(c, DRM_OPI_SRC) == DRM_SRC_PREV);
            if (i == 0) src0 = ctl_field(c, DRM_OPI_SRC);
        }
    }
    if (!ok || (src0 == DRM_SRC_PREV && a != p_end)) return false; // (PREV at the head: the op before it must be the prefix tip)
    const float g = (flags & DRM_RNEA_GRAVITY) ? 9.81f : 0.0f;
    const bool damping = flags & DRM_RNEA_DAMPING;
    // ---- the prefix: motions of its (fixed) links, branch points into their slots (tbar of a prefix link is zero) ----
    Motion M;
    motion_root(M, g);
#pragma unroll 1
    for (int k = 0; k < p_end; ++k) {
        const int c = ctl[k];
        const int src = ctl_field(c, DRM_OPI_SRC), save = ctl_field(c, DRM_OPI_SAVE);
        const OpFT o = load_ft(opf + k * DRM_OPF_STRIDE);
        float rec[12];
        if (src == DRM_SRC_ROOT) motion_root(M, g);
        if (src >= 0) { slot_get(src, 0, rec, 12); motion_from_floats(rec, M); }
        rnea_link_motion(o.F, o.t, 0.0f, 0.0f, M, M);
        if (save >= 0) {
            const float zero6[6] = {0, 0, 0, 0, 0, 0};
            motion_to_floats(M, rec);
            slot_put(save, 0, rec, 12);
            slot_put(save, 18, zero6, 6);
        }
    }
    if (src0 == DRM_SRC_ROOT) motion_root(M, g);
    if (src0 >= 0) { float rec[12]; slot_get(src0, 0, rec, 12); motion_from_floats(rec, M); }
    const bool rooted = src0 == DRM_SRC_ROOT; // the head's parent is the world: no force adjoint comes from above, none goes up
    // ---- up ---------------------------------------------------------------------------------------------------------
    float cs[MAXOPS], sn[MAXOPS], wjs[MAXOPS], ajs[MAXOPS];
    f2 T[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = f2_bcast(0.0f);
    auto joint = [&](int i, float *J, float *t) {
        const OpFT o = load_ft(opf + (a + i) * DRM_OPF_STRIDE);
        joint_rot_z(o.F, cs[i], sn[i], J); // (cos, sin) = (1, 0) for a fixed joint: J == F exactly
        t[0] = o.t[0]; t[1] = o.t[1]; t[2] = o.t[2];
    };
#pragma unroll
    for (int i = 0; i < MAXOPS; ++i) {
        cs[i] = 1.0f; sn[i] = 0.0f; wjs[i] = 0.0f; ajs[i] = 0.0f;
        if (i < len) {
            if (dofs[i] >= 0) {
                float q;
                qf(dofs[i], q, wjs[i], ajs[i]);
                sincos_one(q, sn[i], cs[i]);
            }
            float J[9], t[3];
            joint(i, J, t);
            rnea_link_motion(J, t, wjs[i], ajs[i], M, M);
            if (i > 0 || !rooted) {
                f2 Tn[3];
                tbar_child(J, t, T, Tn);
#pragma unroll
                for (int x = 0; x < 3; ++x) T[x] = Tn[x];
            }
            if (dofs[i] >= 0) T[2][1] += gtau(dofs[i]);
        }
    }
    // ---- down -------------------------------------------------------------------------------------------------------
    Motion B;
    Force carry;
#pragma unroll
    for (int x = 0; x < 3; ++x) { B.wa[x] = B.va[x] = carry.la[x] = f2_bcast(0.0f); }
#pragma unroll
    for (int i = MAXOPS - 1; i >= 0; --i) {
        if (i < len) {
            const int k = a + i;
            const float *of = opf + k * DRM_OPF_STRIDE;
            float J[9], t[3];
            joint(i, J, t);
            const float gtk = dofs[i] >= 0 ? gtau(dofs[i]) : 0.0f;
            const bool has_parent = i > 0 || !rooted;
            Motion Pm;
            f2 U[3];
            if (has_parent) {
                motion_parent(J, t, wjs[i], ajs[i], false, M, Pm);
                f2 x[3] = {T[0], T[1], T[2]};
                x[2][1] -= gtk;
                tbar_parent(J, t, x, U);
            } else {
                motion_root(Pm, g);
#pragma unroll
                for (int x = 0; x < 3; ++x) U[x] = f2_bcast(0.0f);
            }
            Force tot;
            f2 hgl[3], hga[3];
            rnea_body_force_hg(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, M, tot, hgl, hga);
#pragma unroll
            for (int x = 0; x < 3; ++x) tot.la[x] += carry.la[x];
            LinkAdjointP A;
            rnea_link_adjoint_packed(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, wjs[i], M, hgl, hga, T, tot, has_parent,
                                     B, A);
            if (want_gq && dofs[i] >= 0) gout(dofs[i], A.gq, A.wjb + (damping ? of[DRM_OPF_DAMP] * gtk : 0.0f), A.ajb);
            if ((param_mask >> k) & 1u) {
                float gr[DRM_OPF_STRIDE];
                const float ub[6] = {U[0][0], U[1][0], U[2][0], U[0][1], U[1][1], U[2][1]};
                rnea_link_param_adjoint(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, cs[i], sn[i], M, Pm, T, ub, tot,
                                        has_parent, B, A, gr);
                gr[DRM_OPF_DAMP] = damping ? gtk * wjs[i] : 0.0f;
                param_out(k, gr);
            }
            if (i > 0) rnea_link_force_up(J, t, tot, carry);
            B = A.pb;
            M = Pm;
#pragma unroll
            for (int x = 0; x < 3; ++x) T[x] = U[x];
        }
    }
    return true;
}

// Reverse-mode RNEA of a serial chain (DRM_WALK_ARM_CHAIN): the sweeps of rnea_backward_walk without the int table, slots
// or parked records — and without ANY per-link storage.  Every joint transform is orthogonal, so what the forward sweeps
// (A: motions, C: force adjoints) leave at the tip of the chain is enough: sweep D recovers each parent's motion and
// force adjoint from its child's while it walks back to the root
//     w_p = J (w - wj e_z),  v_p = J v - w_p x t, ...        ubar.ang = J tbar.ang,  ubar.lin = J tbar.lin - ubar.ang x t
// recomputes the body force of the link it is at from that motion, and carries the total force of the sub-tree below it
// along (sweep B folded into D).  Live state: one motion, one force adjoint, one total force, one motion adjoint — the
// kernel runs at two or three waves per SIMD instead of one (the earlier form kept tot[CAP][6] and tbar[CAP][6] in registers:
// 256 VGPR + 180 AGPR).
//   row(k) -> op k's constant row;   gout(d, gq, gqd, gqdd);   param_out(k, g[DRM_OPF_STRIDE]) for ops in param_mask
// op k's F / t block: `row.ft(k)` when the row source tells kinematic constants from dynamic ones (the per-robot kernels of a model
// with learnable DYNAMIC parameters, csrc/drm_arm_static.hpp: F / t stay compile-time constants while mass / mcom / I_o come from
// this launch's table), else the start of `row(k)`
template <class ROW> DRM_HD auto row_ft_of(ROW &row, int k, int) -> decltype(row.ft(k)) { return row.ft(k); }
template <class ROW> DRM_HD const float *row_ft_of(ROW &row, int k, long) { return row(k); }
template <class ROW> DRM_HD const float *row_ft(ROW &row, int k) { return row_ft_of(row, k, 0); }

template <int CAP, int NJ, class ROW, class GOUT, class PG>
DRM_HD void rnea_backward_chain(ROW row, bool gravity, bool damping, uint64_t param_mask, bool want_gq,
                                const float (&q)[NJ], const float (&qd)[NJ], const float (&qdd)[NJ],
                                const float (&gtau)[NJ], GOUT gout, PG param_out) {
    float cs[NJ], sn[NJ];
    chain_trig<NJ>(q, cs, sn);
    const float g = gravity ? 9.81f : 0.0f;
    auto joint = [&](int k, float *J, float *t) {
        const OpFT o = load_ft(row_ft(row, k));
        if (k < NJ) {
            joint_rot_z(o.F, cs[k], sn[k], J);
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) J[i] = o.F[i];
        }
        t[0] = o.t[0]; t[1] = o.t[1]; t[2] = o.t[2];
    };
    // ---- A: motions up the chain (only the tip's is kept);  C: adjoints of the total forces, likewise ------------
    // T[i] = (tbar.lin_i, tbar.ang_i): both halves go through J^T together
    Motion M;
    f2 T[3];
    motion_root(M, g);
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = f2_bcast(0.0f);
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        float J[9], t[3];
        joint(k, J, t);
        rnea_link_motion(J, t, k < NJ ? qd[k] : 0.0f, k < NJ ? qdd[k] : 0.0f, M, M);
        if (k > 0) { // tbar_k = (J^T (ubar.lin + ubar.ang x t), J^T ubar.ang), ubar = tbar_(k-1)
            const float ua[3] = {T[0][1], T[1][1], T[2][1]};
            f2 x[3] = {T[0], T[1], T[2]}, y[3];
            x[0][0] += ua[1] * t[2] - ua[2] * t[1];
            x[1][0] += ua[2] * t[0] - ua[0] * t[2];
            x[2][0] += ua[0] * t[1] - ua[1] * t[0];
            matT_vec_p(J, x, y);
#pragma unroll
            for (int i = 0; i < 3; ++i) T[i] = y[i];
        }
        if (k < NJ) T[2][1] += gtau[k];
    }
    // ---- D (with B): back to the root ----------------------------------------------------------------------------
    Motion B;      // motion adjoint: B.wa = (wb, alb), B.va = (vb, ab)
    Force carry;   // total force of the links below, in this link's frame
#pragma unroll
    for (int i = 0; i < 3; ++i) { B.wa[i] = f2_bcast(0.0f); B.va[i] = f2_bcast(0.0f); carry.la[i] = f2_bcast(0.0f); }
#pragma unroll
    for (int k = CAP - 1; k >= 0; --k) {
        const float *of = row(k);
        float J[9], t[3];
        joint(k, J, t);
        const float wj = k < NJ ? qd[k] : 0.0f, aj = k < NJ ? qdd[k] : 0.0f;
        // parent's motion and force adjoint from this link's (inverses of rnea_link_motion / of sweep C; J is orthogonal)
        Motion Pm;
        f2 U[3];
        if (k == 0) {
            motion_root(Pm, g);
#pragma unroll
            for (int i = 0; i < 3; ++i) U[i] = f2_bcast(0.0f);
        } else {
            const float w0 = M.wa[0][0], w1 = M.wa[1][0], v0 = M.va[0][0], v1 = M.va[1][0];
            f2 x[3] = {M.wa[0], M.wa[1], M.wa[2]}, y[3], c[3];
            x[0][1] -= w1 * wj; x[1][1] += w0 * wj; x[2] -= f2_make(wj, aj);
            mat_vec_p(J, x, Pm.wa);                               // (w_p, al_p) = J ((w, al) - joint terms)
            x[0] = M.va[0]; x[1] = M.va[1]; x[2] = M.va[2];
            x[0][1] -= v1 * wj; x[1][1] += v0 * wj;
            mat_vec_p(J, x, y);
            cross3_ps(Pm.wa, t, c);
#pragma unroll
            for (int i = 0; i < 3; ++i) Pm.va[i] = y[i] - c[i];   // (v_p, a_p) = J (...) - (w_p, al_p) x t
            x[0] = T[0]; x[1] = T[1]; x[2] = T[2];
            if (k < NJ) x[2][1] -= gtau[k];
            mat_vec_p(J, x, U);                                   // (J tbar.lin, ubar.ang)
            const float ua[3] = {U[0][1], U[1][1], U[2][1]};
            U[0][0] -= ua[1] * t[2] - ua[2] * t[1];               // ubar.lin = J tbar.lin - ubar.ang x t
            U[1][0] -= ua[2] * t[0] - ua[0] * t[2];
            U[2][0] -= ua[0] * t[1] - ua[1] * t[0];
        }
        // total force of the sub-tree at this link: its own body force (from the recovered motion) + what came up
        Force tot;
        f2 hgl[3], hga[3];
        rnea_body_force_hg(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, M, tot, hgl, hga);
#pragma unroll
        for (int i = 0; i < 3; ++i) tot.la[i] += carry.la[i];
        LinkAdjointP A;
        rnea_link_adjoint_packed(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, wj, M, hgl, hga, T, tot, k > 0, B, A);
        if (k < NJ && want_gq) gout(k, A.gq, A.wjb + (damping ? of[DRM_OPF_DAMP] * gtau[k] : 0.0f), A.ajb);
        if ((param_mask >> k) & 1u) {
            float gr[DRM_OPF_STRIDE];
            const float ub[6] = {U[0][0], U[1][0], U[2][0], U[0][1], U[1][1], U[2][1]};
            rnea_link_param_adjoint(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, k < NJ ? cs[k] : 1.0f,
                                    k < NJ ? sn[k] : 0.0f, M, Pm, T, ub, tot, k > 0, B, A, gr);
            gr[DRM_OPF_DAMP] = (damping && k < NJ) ? gtau[k < NJ ? k : 0] * qd[k < NJ ? k : 0] : 0.0f;
            param_out(k, gr);
        }
        if (k > 0) rnea_link_force_up(J, t, tot, carry); // the sub-tree's force in the parent's frame
        B = A.pb;
        M = Pm;
#pragma unroll
        for (int i = 0; i < 3; ++i) T[i] = U[i];
    }
}

// ---------------------------------------------------------------------------
// Reverse-mode RNEA of "an arm that carries a hand" (DRM_WALK_ARM_HAND: P serial prefix ops, K serial sub-chains of L ops off
// the last prefix op), straight-line like rnea_backward_chain and with the same idea: nothing is stored per link, every
// parent's motion and force adjoint are recovered from its child's on the way back.  What a tree adds is the palm (op P - 1):
// its motion and force adjoint are put away once (36 floats with the two sums below; put / get), every sub-chain starts its
// way up from them, walks back down to them and ADDS what it hands the palm — its total force and its motion adjoint — to the
// sums; the prefix then walks back from the palm with the sums as if they had come from a single child.
//   one link of the way down (sweep D with B folded in), any joint kind:
//     in : M, T = the link's motion and force adjoint; B = motion adjoint from below; carry = force from below (link frame)
//     out: M, T = the PARENT's (recovered); B = the adjoint handed to the parent; carry = the sub-tree's force in the parent's
//          frame; gq / gqd / gqdd of the joint; param_out(op, row gradient) when `learn`
// ---------------------------------------------------------------------------
template <class PG>
DRM_HD void rnea_backward_link_down(const float *of, int op, bool moving, bool pris, const float *J, const float *t, float c, float s,
                                    float qv, float wj, float aj, float gtk, bool has_parent, float g, bool damping, bool learn,
                                    Motion &M, f2 (&T)[3], Motion &B, Force &carry, float &gq, float &gqd, float &gqdd,
                                    PG param_out) {
    Motion Pm, pbn;
    f2 U[3];
    if (has_parent) {
        motion_parent(J, t, wj, aj, pris, M, Pm);
        f2 x[3] = {T[0], T[1], T[2]};
        x[2][pris ? 0 : 1] -= gtk;
        tbar_parent(J, t, x, U);
    } else {
        motion_root(Pm, g);
#pragma unroll
        for (int i = 0; i < 3; ++i) U[i] = f2_bcast(0.0f);
    }
    Force tot;
    f2 hgl[3], hga[3];
    rnea_body_force_hg(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, M, tot, hgl, hga);
#pragma unroll
    for (int i = 0; i < 3; ++i) tot.la[i] += carry.la[i];
    float wjb, ajb;
    if (!pris) {
        LinkAdjointP A;
        rnea_link_adjoint_packed(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, wj, M, hgl, hga, T, tot, has_parent, B, A);
        gq = A.gq; wjb = A.wjb; ajb = A.ajb;
        pbn = A.pb;
        if (learn) {
            float gr[DRM_OPF_STRIDE];
            const float ub[6] = {U[0][0], U[1][0], U[2][0], U[0][1], U[1][1], U[2][1]};
            rnea_link_param_adjoint(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, c, s, M, Pm, T, ub, tot, has_parent, B, A, gr);
            gr[DRM_OPF_DAMP] = (damping && moving) ? gtk * wj : 0.0f;
            param_out(op, gr);
        }
    } else { // a sliding joint: the scalar form (its motion subspace differs; rare)
        float mo[12], mb[12], par[12], fb[6], ub[6], tf[6];
        motion_to_floats(M, mo); motion_to_floats(B, mb); motion_to_floats(Pm, par);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            fb[i] = T[i][0]; fb[3 + i] = T[i][1]; ub[i] = U[i][0]; ub[3 + i] = U[i][1];
            tf[i] = tot.la[i][0]; tf[3 + i] = tot.la[i][1];
        }
        LinkAdjoint A;
        rnea_link_adjoint(of[DRM_OPF_MASS], of + DRM_OPF_MCOM, of + DRM_OPF_IO, J, t, wj, mo, fb, par, mb, ub, tf, has_parent, A, true, learn);
        gq = A.gq; wjb = A.wjb; ajb = A.ajb;
        motion_from_floats(A.pb, pbn);
        if (learn) { // J = F and t = trans + F e_z q
            float gr[DRM_OPF_STRIDE];
#pragma unroll
            for (int i = 0; i < DRM_OPF_STRIDE; ++i) gr[i] = 0.0f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                gr[DRM_OPF_FIJ(r, 0)] = A.Jb[r * 3 + 0];
                gr[DRM_OPF_FIJ(r, 1)] = A.Jb[r * 3 + 1];
                gr[DRM_OPF_FIJ(r, 2)] = A.Jb[r * 3 + 2] + A.tb[r] * qv;
                gr[DRM_OPF_TI(r)] = A.tb[r];
                gr[DRM_OPF_MCOM + r] = A.gmc[r];
            }
            gr[DRM_OPF_MASS] = A.gm;
#pragma unroll
            for (int i = 0; i < 9; ++i) gr[DRM_OPF_IO + i] = A.gIo[i];
            gr[DRM_OPF_DAMP] = damping ? gtk * wj : 0.0f;
            param_out(op, gr);
        }
    }
    gqd = wjb + ((damping && moving) ? of[DRM_OPF_DAMP] * gtk : 0.0f);
    gqdd = ajb;
    if (has_parent) {
        Force up;
        rnea_link_force_up(J, t, tot, up); // the sub-tree's force in the parent's frame
        carry = up;
    }
    B = pbn;
    M = Pm;
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = U[i];
}

//   row(op), kind(op) (1 = moves, 2 = prismatic)        wave-uniform
//   pq(k, q, qd, qdd, gtau) joint state and dL/dtau of prefix op k (0 where fixed);  hq(j, i, q, qd, qdd, gtau) of op i of sub-chain j
//   gout(op, gq, gqd, gqdd) for every moving op;  param_out(op, g[DRM_OPF_STRIDE]) for ops in param_mask
//   put(i, x) / get(i) -> x    36 lane-private floats (the palm's motion 0..11 and force adjoint 12..17, the sums 18..35)
template <int P, int L, class ROW, class KIND, class PQ, class HQ, class GOUT, class PG, class PUT, class GET>
DRM_HD void rnea_backward_arm_hand(ROW row, KIND kind, int K, bool gravity, bool damping, uint64_t param_mask, bool want_gq, PQ pq, HQ hq,
                                   GOUT gout, PG param_out, PUT put, GET get) {
    const float g = gravity ? 9.81f : 0.0f;
    // a prefix op's joint state is fetched (and its cos / sin formed) where each sweep needs it: nothing of the prefix stays in
    // registers while the sub-chains run
    auto prefix_joint = [&](int k, float *J, float *t, float &qk, float &wj, float &aj, float &gtk, float &c, float &s) {
        const int kd = kind(k);
        pq(k, qk, wj, aj, gtk);
        c = 1.0f; s = 0.0f;
        if ((kd & 1) && !(kd & 2)) sincos_one(qk, s, c);
        joint_transform(load_ft(row(k)), kd & 1, kd & 2, qk, c, s, J, t);
    };
    // ---- up the prefix: motions (A) and force adjoints (C); only the palm's are kept --------------------------------------
    {
        Motion M;
        f2 T[3];
        motion_root(M, g);
#pragma unroll
        for (int i = 0; i < 3; ++i) T[i] = f2_bcast(0.0f);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int kd = kind(k);
            float J[9], t[3], qk, wj, aj, gtk, c, sn_;
            prefix_joint(k, J, t, qk, wj, aj, gtk, c, sn_);
            motion_step(J, t, wj, aj, kd & 2, M, M);
            if (k > 0) {
                f2 Tn[3];
                tbar_child(J, t, T, Tn);
#pragma unroll
                for (int i = 0; i < 3; ++i) T[i] = Tn[i];
            }
            if (kd & 1) T[2][(kd & 2) ? 0 : 1] += gtk;
        }
        float rec[12];
        motion_to_floats(M, rec);
#pragma unroll
        for (int i = 0; i < 12; ++i) put(i, rec[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) { put(12 + i, T[i][0]); put(15 + i, T[i][1]); }
#pragma unroll
        for (int i = 18; i < 36; ++i) put(i, 0.0f);
    }
    // ---- every sub-chain: up from the palm, back down to it, its force and motion adjoint into the sums -------------------
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
        float hqv[L], hqd[L], hqdd[L], hgt[L], hc[L], hs[L];
#pragma unroll
        for (int i = 0; i < L; ++i) hq(j, i, hqv[i], hqd[i], hqdd[i], hgt[i]);
        chain_trig<L>(hqv, hc, hs);
        auto joint = [&](int i, float *J, float *t) {
            const int kd = kind(P + j * L + i);
            joint_transform(load_ft(row(P + j * L + i)), kd & 1, kd & 2, hqv[i], hc[i], hs[i], J, t);
        };
        Motion M;
        f2 T[3];
        {
            float rec[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) rec[i] = get(i);
            motion_from_floats(rec, M);
#pragma unroll
            for (int i = 0; i < 3; ++i) T[i] = f2_make(get(12 + i), get(15 + i));
        }
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int kd = kind(P + j * L + i);
            float J[9], t[3];
            joint(i, J, t);
            motion_step(J, t, hqd[i], hqdd[i], kd & 2, M, M);
            f2 Tn[3];
            tbar_child(J, t, T, Tn);
#pragma unroll
            for (int x = 0; x < 3; ++x) T[x] = Tn[x];
            if (kd & 1) T[2][(kd & 2) ? 0 : 1] += hgt[i];
        }
        Motion B;
        Force carry;
#pragma unroll
        for (int x = 0; x < 3; ++x) { B.wa[x] = B.va[x] = carry.la[x] = f2_bcast(0.0f); }
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            const int op = P + j * L + i, kd = kind(op);
            float J[9], t[3], gq, gqd, gqdd;
            joint(i, J, t);
            rnea_backward_link_down(row(op), op, kd & 1, kd & 2, J, t, hc[i], hs[i], hqv[i], hqd[i], hqdd[i], (kd & 1) ? hgt[i] : 0.0f,
                                    true, g, damping, (param_mask >> op) & 1u, M, T, B, carry, gq, gqd, gqdd, param_out);
            if (want_gq && (kd & 1)) gout(op, gq, gqd, gqdd);
        }
        float rec[12];
        motion_to_floats(B, rec);
#pragma unroll
        for (int i = 0; i < 12; ++i) put(18 + i, get(18 + i) + rec[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            put(30 + i, get(30 + i) + carry.la[i][0]);
            put(33 + i, get(33 + i) + carry.la[i][1]);
        }
    }
    // ---- down the prefix from the palm, the sums standing in for a single child --------------------------------------------
    Motion M, B;
    f2 T[3];
    Force carry;
    {
        float rec[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) rec[i] = get(i);
        motion_from_floats(rec, M);
#pragma unroll
        for (int i = 0; i < 12; ++i) rec[i] = get(18 + i);
        motion_from_floats(rec, B);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T[i] = f2_make(get(12 + i), get(15 + i));
            carry.la[i] = f2_make(get(30 + i), get(33 + i));
        }
    }
#pragma unroll
    for (int k = P - 1; k >= 0; --k) {
        const int kd = kind(k);
        float J[9], t[3], gq, gqd, gqdd, qk, wj, aj, gtk, c, sn_;
        prefix_joint(k, J, t, qk, wj, aj, gtk, c, sn_);
        rnea_backward_link_down(row(k), k, kd & 1, kd & 2, J, t, c, sn_, qk, wj, aj, (kd & 1) ? gtk : 0.0f, k > 0, g, damping,
                                (param_mask >> k) & 1u, M, T, B, carry, gq, gqd, gqdd, param_out);
        if (want_gq && (kd & 1)) gout(k, gq, gqd, gqdd);
    }
}

// ---------------------------------------------------------------------------
// Link-table rows from URDF-level link parameters, and the reverse-mode derivative of that map: what the
// reference recomputes with ~60 tiny torch ops per link on every call (rigid_body.py:138-143 R_fixed = (Rz Ry) Rx;
// spatial_vector_algebra.py:321-327 mcom = m com, I_o = I_c + m S(com) S(com)^T) and differentiates with as many
// autograd nodes.  One link per thread; used for the rows of LEARNABLE links only (constant rows are built once).
//   p[20]  : rpy (3), trans (3), mass (1), com (3), inertia_mat (9, about the com), damping (1)
//   row[32]: link-table layout  F (9) t (3) m (1) mcom (3) I_o (9) damping (1) 0...
// ---------------------------------------------------------------------------
constexpr int LINK_PARAM_FLOATS = 20;
DRM_HD void rpy_factors(const float *rpy, float *Rx, float *Ry, float *Rz, float *cs) {
    float sr, cr, sp, cp, sy, cy;
    sincos_f(rpy[0], sr, cr);
    sincos_f(rpy[1], sp, cp);
    sincos_f(rpy[2], sy, cy);
    const float rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr}, ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp},
                rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1};
#pragma unroll
    for (int i = 0; i < 9; ++i) { Rx[i] = rx[i]; Ry[i] = ry[i]; Rz[i] = rz[i]; }
    cs[0] = sr; cs[1] = cr; cs[2] = sp; cs[3] = cp; cs[4] = sy; cs[5] = cy;
}
DRM_HD void mat3_mul(const float *A, const float *B, float *C) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + A[r * 3 + 1] * B[1 * 3 + c] + A[r * 3 + 2] * B[2 * 3 + c];
}
DRM_HD void link_row(const float *p, float *row) {
    float Rx[9], Ry[9], Rz[9], cs[6], T[9];
    rpy_factors(p, Rx, Ry, Rz, cs);
    mat3_mul(Rz, Ry, T);
    mat3_mul(T, Rx, row);
    const float m = p[6];
    const float *t = p + 3, *c = p + 7, *I = p + 10;
    const float c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { row[9 + i] = t[i]; row[13 + i] = c[i] * m; }
    row[12] = m;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) row[16 + r * 3 + k] = I[r * 3 + k] + m * ((r == k ? c2 : 0.0f) - c[r] * c[k]);
    row[25] = p[19];
#pragma unroll
    for (int i = 26; i < DRM_OPF_STRIDE; ++i) row[i] = 0.0f;
}
// d F / d (roll, pitch, yaw) of F = (Rz Ry) Rx, each a 3 x 3 matrix (row-major)
DRM_HD void rpy_jacobian(const float *rpy, float *D0, float *D1, float *D2) {
    float Rx[9], Ry[9], Rz[9], cs[6], T[9], E[9];
    rpy_factors(rpy, Rx, Ry, Rz, cs);
    const float sr = cs[0], cr = cs[1], sp = cs[2], cp = cs[3], sy = cs[4], cy = cs[5];
    const float dRx[9] = {0, 0, 0, 0, -sr, -cr, 0, cr, -sr}, dRy[9] = {-sp, 0, cp, 0, 0, 0, -cp, 0, -sp},
                dRz[9] = {-sy, -cy, 0, cy, -sy, 0, 0, 0, 0};
    mat3_mul(Rz, Ry, T);
    mat3_mul(T, dRx, D0);
    mat3_mul(Rz, dRy, E);
    mat3_mul(E, Rx, D1);
    mat3_mul(dRz, Ry, E);
    mat3_mul(E, Rx, D2);
}
DRM_HD float dot9(const float *A, const float *g) {
    float a = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) a += A[i] * g[i];
    return a;
}
DRM_HD void link_row_backward_rest(const float *p, const float *g, float *gp);
DRM_HD void link_row_backward(const float *p, const float *g, float *gp) {
    float D0[9], D1[9], D2[9];
    rpy_jacobian(p, D0, D1, D2);
    gp[0] = dot9(D0, g);
    gp[1] = dot9(D1, g);
    gp[2] = dot9(D2, g);
    link_row_backward_rest(p, g, gp);
}
// gp[3 .. 19]: everything but the three angles (trans, mass, com, inertia_mat, damping)
DRM_HD void link_row_backward_rest(const float *p, const float *g, float *gp) {
    const float m = p[6];
    const float *c = p + 7;
    const float *gI = g + 16;
    const float c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2], trG = gI[0] + gI[4] + gI[8];
    float gm = g[12], cGc = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { gp[3 + i] = g[9 + i]; gm += c[i] * g[13 + i]; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) cGc += gI[r * 3 + k] * c[r] * c[k];
    gp[6] = gm + (c2 * trG - cGc);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float Gc = 0.0f; // ((G + G^T) c)_i
#pragma unroll
        for (int k = 0; k < 3; ++k) Gc += (gI[i * 3 + k] + gI[k * 3 + i]) * c[k];
        gp[7 + i] = m * g[13 + i] + m * (2.0f * c[i] * trG - Gc);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) gp[10 + i] = gI[i];
    gp[19] = g[25];
}

// ---------------------------------------------------------------------------
// Joint-space inertia matrix H(q) by the composite-rigid-body algorithm.
//
// The reference builds H column by column from n + 1 inverse-dynamics passes (robot_model.py:402-450:
// H[:, j] = ID(q, 0, e_j) - ID(q, 0, 0)); with qd = 0 inverse dynamics is linear in qdd, so H is the matrix of
// that linear map, which the composite-rigid-body algorithm evaluates directly (same quantity, ~4x fewer
// flops, exactly symmetric, and without the cancellation against the gravity term):
//   backward sweep   Ic_k = I_k + sum over children c of (Ic_c moved into frame k)          composite inertias
//   forward sweep    every ancestor joint axis S_j is carried down the tree in current-frame coordinates;
//                    at a moving link k:  F = Ic_k S_k,  H[k][k] = S_k . F,  H[k][j] = H[j][k] = F . S_j.
// Same body-frame Pluecker conventions as the RNEA walk (joint about +z of the stored frame, child -> parent
// transform x_p = J x_c + t, spatial inertia f = m v - h x w, n = I w + h x v with h = m c).
// (the tree form of the walk is drm_tree.hpp crba_tree_walk, the chain form crba_chain below)
// ---------------------------------------------------------------------------
struct Inertia {
    float m;
    float h[3]; // first moment m c
    float I[6]; // xx xy xz yy yz zz, about the link origin
};
DRM_HD void inertia_zero(Inertia &a) {
    a.m = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) a.h[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) a.I[i] = 0.0f;
}
DRM_HD void inertia_add(Inertia &a, const Inertia &b) {
    a.m += b.m;
#pragma unroll
    for (int i = 0; i < 3; ++i) a.h[i] += b.h[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) a.I[i] += b.I[i];
}
// inertia c (child frame) expressed in the parent frame, x_p = J x_c + t:
//   h' = J h + m t,   I' = J I J^T + 2 (w . t) E - w t^T - t w^T   with  w = J h + (m / 2) t
DRM_HD void inertia_to_parent(const float *J, const float *t, const Inertia &c, Inertia &out) {
    float g[3], M[9], w[3];
    mat_vec(J, c.h, g);
    const float If[9] = {c.I[0], c.I[1], c.I[2], c.I[1], c.I[3], c.I[4], c.I[2], c.I[4], c.I[5]};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            M[r * 3 + k] = J[r * 3 + 0] * If[0 * 3 + k] + J[r * 3 + 1] * If[1 * 3 + k] + J[r * 3 + 2] * If[2 * 3 + k];
    auto R = [&](int r, int k) { return M[r * 3 + 0] * J[k * 3 + 0] + M[r * 3 + 1] * J[k * 3 + 1] + M[r * 3 + 2] * J[k * 3 + 2]; };
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = g[i] + 0.5f * c.m * t[i];
    const float wt2 = 2.0f * (w[0] * t[0] + w[1] * t[1] + w[2] * t[2]);
    out.I[0] = R(0, 0) + (wt2 - 2.0f * w[0] * t[0]);
    out.I[1] = R(0, 1) - (w[0] * t[1] + t[0] * w[1]);
    out.I[2] = R(0, 2) - (w[0] * t[2] + t[0] * w[2]);
    out.I[3] = R(1, 1) + (wt2 - 2.0f * w[1] * t[1]);
    out.I[4] = R(1, 2) - (w[1] * t[2] + t[1] * w[2]);
    out.I[5] = R(2, 2) + (wt2 - 2.0f * w[2] * t[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) out.h[i] = g[i] + c.m * t[i];
    out.m = c.m;
}
// Joint-space inertia matrix of a serial chain (DRM_WALK_ARM_CHAIN: NJ moving joints driving DoF columns 0..NJ-1,
// then CAP - NJ fixed links or identity padding): the composite-rigid-body walk without control words and branch points,
// joint axes carried as packed (ang_i, lin_i) pairs, two joints per sincos evaluation.
//   row(k) -> pointer to op k's constant row;   hout(i, j, v) -> H[i][j] = v (called for both triangles)
template <int CAP, int NJ, class ROW, class HOUT>
DRM_HD void crba_chain_trig(ROW row, const float (&cs)[NJ], const float (&sn)[NJ], HOUT hout);
template <int CAP, int NJ, class ROW, class HOUT>
DRM_HD void crba_chain(ROW row, const float (&q)[NJ], HOUT hout) {
    float cs[NJ], sn[NJ];
    chain_trig<NJ>(q, cs, sn);
    crba_chain_trig<CAP, NJ>(row, cs, sn, hout);
}
// ... given cos / sin of the joint angles (the forward-dynamics arm kernel shares them with its RNEA walk)
template <int CAP, int NJ, class ROW, class HOUT>
DRM_HD void crba_chain_trig(ROW row, const float (&cs)[NJ], const float (&sn)[NJ], HOUT hout) {
    // all joint transforms first (kept in registers: both sweeps below read them)
    float J[CAP][9];
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const OpFT o = load_ft(row(k));
        if (k < NJ) {
            joint_rot_z(o.F, cs[k], sn[k], J[k]);
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) J[k][i] = o.F[i];
        }
    }
    // one sweep from the tip to the root: the composite inertia is a running value (nothing is stored per link),
    // and at every moving link k the force F = Ic_k S_k is walked up the chain as a packed (f_i, n_i) pair vector,
    // leaving H[j][k] = S_j . F = n_z at every joint j above — the classic column-by-column form of the algorithm.
    Inertia carry;
    inertia_zero(carry);
#pragma unroll
    for (int k = CAP - 1; k >= 0; --k) {
        const float *of = row(k);
        Inertia tot;
        tot.m = of[DRM_OPF_MASS];
#pragma unroll
        for (int i = 0; i < 3; ++i) tot.h[i] = of[DRM_OPF_MCOM + i];
        tot.I[0] = of[DRM_OPF_IO + 0]; tot.I[1] = of[DRM_OPF_IO + 1]; tot.I[2] = of[DRM_OPF_IO + 2];
        tot.I[3] = of[DRM_OPF_IO + 4]; tot.I[4] = of[DRM_OPF_IO + 5]; tot.I[5] = of[DRM_OPF_IO + 8];
        if (k < CAP - 1) inertia_add(tot, carry);
        if (k < NJ) {
            // F = Ic S_k with S_k = (ang e_z, lin 0):  f = -h x e_z = (-h_y, h_x, 0),  n = I e_z
            Force F;
            F.la[0] = f2_make(-tot.h[1], tot.I[2]);
            F.la[1] = f2_make(tot.h[0], tot.I[4]);
            F.la[2] = f2_make(0.0f, tot.I[5]);
            hout(k, k, tot.I[5]);
#pragma unroll
            for (int j = k - 1; j >= 0; --j) {
                const float *oc = row(j + 1);
                const float t[3] = {oc[DRM_OPF_TI(0)], oc[DRM_OPF_TI(1)], oc[DRM_OPF_TI(2)]};
                Force up;
                rnea_link_force_up(J[j + 1], t, F, up); // into the frame of link j (spatial_vector_algebra.py:281-291)
                F = up;
                hout(j, k, F.la[2][1]);
                hout(k, j, F.la[2][1]);
            }
        }
        if (k > 0) {
            const float t[3] = {of[DRM_OPF_TI(0)], of[DRM_OPF_TI(1)], of[DRM_OPF_TI(2)]};
            inertia_to_parent(J[k], t, tot, carry);
        }
    }
}

// ---------------------------------------------------------------------------
// Solve H x = b for one sample, H symmetric positive definite (the joint-space inertia matrix), by the factorisation
// H = L^T D L (L unit lower triangular) taken from the LAST joint towards the first, followed by the three solves.
// Eliminating the distal joints first is what the reference's articulated-body recursion does (robot_model.py:487-624
// works from the leaves to the root and divides joint by joint): the Schur complements then only ADD the small
// inertias of the finger / wrist links to the large entries of the joints above them.  A Cholesky factorisation from
// joint 0 down subtracts kilogram-scale terms from gram-scale blocks instead and loses cond(H) * eps there.  Same flops as Cholesky.
// H is the lane's PACKED LOWER TRIANGLE (entry (i, j), i >= j, at i (i + 1) / 2 + j; LDS in the kernel); b is
// overwritten with x.  Used by the forward-dynamics kernels: qdd = H^-1 (f - nle).
// ---------------------------------------------------------------------------
DRM_HD constexpr int tri_index(int i, int j) { return i * (i + 1) / 2 + j; } // i >= j
DRM_HD float recip_f(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __frcp_rn(x);
#else
    return 1.0f / x;
#endif
}
// H(i) -> reference to element i of the packed triangle (an LDS row in the kernel)
template <class HA>
DRM_HD void ltdl_factor_acc(int n, HA H) {
    for (int k = n - 1; k >= 0; --k) {
        const int rk = tri_index(k, 0);
        const float inv = recip_f(H(rk + k));
        H(rk + k) = inv; // the diagonal keeps 1 / D_k
        for (int i = 0; i < k; ++i) {
            const float hki = H(rk + i);
            const float a = hki * inv;
            const int ri = tri_index(i, 0);
            for (int j = 0; j < i; ++j) H(ri + j) -= hki * H(rk + j);  // H(rk + j), j < i, already holds L[k][j] = H[k][j] / D_k
            H(ri + i) -= hki * a;
            H(rk + i) = a;                                             // L[k][i]
        }
    }
}
// b <- (L^T D L)^-1 b with the factors left by ltdl_factor_acc
template <class HA>
DRM_HD void ltdl_apply_acc(int n, HA H, float *b) {
    for (int i = n - 1; i >= 0; --i) {       // y = L^-T b
        const int ri = tri_index(i, 0);
        const float bi = b[i];
        for (int j = 0; j < i; ++j) b[j] -= H(ri + j) * bi;
    }
    for (int i = 0; i < n; ++i) {            // z = D^-1 y,  x = L^-1 z
        const int ri = tri_index(i, 0);
        float t = b[i] * H(ri + i);
        for (int j = 0; j < i; ++j) t -= H(ri + j) * b[j];
        b[i] = t;
    }
}
template <class HA>
DRM_HD void ltdl_solve_acc(int n, HA H, float *b) {
    ltdl_factor_acc(n, H);
    ltdl_apply_acc(n, H, b);
}
DRM_HD void ltdl_solve(int n, float *H, float *b) {
    ltdl_solve_acc(n, [H](int i) -> float & { return H[i]; }, b);
}

// The same factorisation and solves for a compile-time size, fully unrolled: H (packed lower triangle) and b live
// in registers (the arm kernels: n = 7, 28 + 7 floats).
DRM_HD f2 recip_f(f2 x) { return f2_make(recip_f(x[0]), recip_f(x[1])); }      // (two samples per lane)
template <int N, class T = float>
DRM_HD void ltdl_solve_unrolled(T (&H)[N * (N + 1) / 2], T (&b)[N]) {
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        const T inv = recip_f(H[tri_index(k, k)]);
        H[tri_index(k, k)] = inv;
#pragma unroll
        for (int i = 0; i < k; ++i) {
            const T hki = H[tri_index(k, i)];
            const T a = hki * inv;
#pragma unroll
            for (int j = 0; j < i; ++j) H[tri_index(i, j)] -= hki * H[tri_index(k, j)];
            H[tri_index(i, i)] -= hki * a;
            H[tri_index(k, i)] = a;
        }
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
#pragma unroll
        for (int j = 0; j < i; ++j) b[j] -= H[tri_index(i, j)] * b[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        T t = b[i] * H[tri_index(i, i)];
#pragma unroll
        for (int j = 0; j < i; ++j) t -= H[tri_index(i, j)] * b[j];
        b[i] = t;
    }
}

// ---------------------------------------------------------------------------
// Joint-space inertia matrix of a serial chain, TWO SAMPLES PER LANE (round 6): crba_chain_trig on sample pairs — a pair is the
// same quantity of rows b and b + 64 of a 128-row tile, the link constants enter as broadcast operands, every instruction of the
// per-sample walk is a full v_pk_*_f32 (the form that took inverse dynamics from 1 126 to 780 VALU per 64 rows; forward dynamics =
// this + rnea_chain2_trig + the L^T D L solve on pairs).  Same formulas, same association order as the one-sample walk.
//   row(k) -> op k's constant row;   cs / sn: [d] = (sample A, sample B);   hout(i, j, f2) for i >= j (the lower triangle)
// ---------------------------------------------------------------------------
struct Inertia2 {
    float m;     // the sub-tree's mass: a constant of the robot, the same for both samples
    f2 h[3];     // first moment
    f2 I[6];     // xx xy xz yy yz zz about the link origin
};
// inertia_to_parent on sample pairs: x_p = J x_c + t with J = (c0 | c1 | F[:, 2])
DRM_HD void inertia2_to_parent(const Joint2 &J, const float *F, const float *t, const Inertia2 &c, Inertia2 &out) {
    f2 g[3], M[9], w[3];
    joint2_N(J, F, c.h, g);
    const f2 If[9] = {c.I[0], c.I[1], c.I[2], c.I[1], c.I[3], c.I[4], c.I[2], c.I[4], c.I[5]};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) M[r * 3 + k] = J.c0[r] * If[0 * 3 + k] + J.c1[r] * If[1 * 3 + k] + f2_bcast(F[r * 3 + 2]) * If[2 * 3 + k];
    auto R = [&](int r, int k) { return M[r * 3 + 0] * J.c0[k] + M[r * 3 + 1] * J.c1[k] + M[r * 3 + 2] * f2_bcast(F[k * 3 + 2]); };
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = g[i] + f2_bcast(0.5f * c.m * t[i]);
    const f2 wt2 = f2_bcast(2.0f) * (w[0] * f2_bcast(t[0]) + w[1] * f2_bcast(t[1]) + w[2] * f2_bcast(t[2]));
    out.I[0] = R(0, 0) + (wt2 - f2_bcast(2.0f * t[0]) * w[0]);
    out.I[1] = R(0, 1) - (w[0] * f2_bcast(t[1]) + f2_bcast(t[0]) * w[1]);
    out.I[2] = R(0, 2) - (w[0] * f2_bcast(t[2]) + f2_bcast(t[0]) * w[2]);
    out.I[3] = R(1, 1) + (wt2 - f2_bcast(2.0f * t[1]) * w[1]);
    out.I[4] = R(1, 2) - (w[1] * f2_bcast(t[2]) + f2_bcast(t[1]) * w[2]);
    out.I[5] = R(2, 2) + (wt2 - f2_bcast(2.0f * t[2]) * w[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) out.h[i] = g[i] + f2_bcast(c.m * t[i]);
    out.m = c.m;
}
template <int CAP, int NJ, class ROW, class HOUT>
DRM_HD void crba_chain2_trig(ROW row, const f2 (&cs)[NJ], const f2 (&sn)[NJ], HOUT hout) {
    Joint2 J[CAP];
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const OpFT o = load_ft(row(k));
        if (k < NJ) joint2_moving(o.F, cs[k], sn[k], J[k]);
        else joint2_fixed(o.F, J[k]);
    }
    Inertia2 carry;
    carry.m = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) carry.h[i] = f2_bcast(0.0f);
#pragma unroll
    for (int i = 0; i < 6; ++i) carry.I[i] = f2_bcast(0.0f);
#pragma unroll
    for (int k = CAP - 1; k >= 0; --k) {
        const float *of = row(k);
        Inertia2 tot;
        tot.m = of[DRM_OPF_MASS];
#pragma unroll
        for (int i = 0; i < 3; ++i) tot.h[i] = f2_bcast(of[DRM_OPF_MCOM + i]);
        const int at[6] = {0, 1, 2, 4, 5, 8};
#pragma unroll
        for (int i = 0; i < 6; ++i) tot.I[i] = f2_bcast(of[DRM_OPF_IO + at[i]]);
        if (k < CAP - 1) {
            tot.m += carry.m;
#pragma unroll
            for (int i = 0; i < 3; ++i) tot.h[i] += carry.h[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) tot.I[i] += carry.I[i];
        }
        if (k < NJ) {      // F = Ic S_k with S_k = (ang e_z, lin 0):  f = (-h_y, h_x, 0),  n = I e_z
            Force2 F;
            F.f[0] = f2_bcast(0.0f) - tot.h[1]; F.f[1] = tot.h[0]; F.f[2] = f2_bcast(0.0f);
            F.n[0] = tot.I[2]; F.n[1] = tot.I[4]; F.n[2] = tot.I[5];
            hout(k, k, tot.I[5]);
#pragma unroll
            for (int j = k - 1; j >= 0; --j) {
                const OpFT oc = load_ft(row(j + 1));
                Force2 up;
                joint2_N(J[j + 1], oc.F, F.f, up.f);
                joint2_N(J[j + 1], oc.F, F.n, up.n);
                f2 x[3];
                cross2_cv(oc.t, up.f, x);
#pragma unroll
                for (int i = 0; i < 3; ++i) up.n[i] += x[i];
                F = up;
                hout(k, j, F.n[2]);
            }
        }
        if (k > 0) {
            const OpFT o = load_ft(of);
            inertia2_to_parent(J[k], o.F, o.t, tot, carry);
        }
    }
}

// ---------------------------------------------------------------------------
// Reverse-mode inverse dynamics of a serial chain, INPUT gradients, TWO SAMPLES PER LANE (round 6): rnea_backward_chain's three
// sweeps (A motions up, C force adjoints up, D back to the root with B folded in; nothing stored per link, every parent's motion and
// force adjoint recovered from its child's) with a pair = the same quantity of rows b and b + 64, on the helpers of the two-sample
// RNEA (Joint2, Motion2, Force2).  The one-sample form packs (w, al) / (v, a) of ONE sample and leaves its scalar third — crosses of
// two per-sample vectors, the joint rotations, the closed-form dL/dq terms — on half-empty instructions; here every instruction is
// a full v_pk_*_f32.  The link adjoint is rnea_link_adjoint's (revolute joint about +z, no parameter gradients) written out on pairs.
//   gout(d, gq, gqd, gqdd): f2 each
// ---------------------------------------------------------------------------
DRM_HD void acc2_cross_vv(f2 *acc, const f2 *a, const f2 *b) { // acc += a x b
    acc[0] += a[1] * b[2] - a[2] * b[1];
    acc[1] += a[2] * b[0] - a[0] * b[2];
    acc[2] += a[0] * b[1] - a[1] * b[0];
}
DRM_HD void acc2_cross_vc(f2 *acc, const f2 *a, const float *b, float sign) { // acc += sign (a x b), b a constant of the robot
    acc[0] += f2_bcast(sign) * (a[1] * f2_bcast(b[2]) - a[2] * f2_bcast(b[1]));
    acc[1] += f2_bcast(sign) * (a[2] * f2_bcast(b[0]) - a[0] * f2_bcast(b[2]));
    acc[2] += f2_bcast(sign) * (a[0] * f2_bcast(b[1]) - a[1] * f2_bcast(b[0]));
}
DRM_HD void acc2_matT_c(f2 *acc, const float *M, const f2 *x) { // acc += M^T x, M a constant 3x3 (row-major)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += f2_bcast(M[0 * 3 + c]) * x[0] + f2_bcast(M[1 * 3 + c]) * x[1] + f2_bcast(M[2 * 3 + c]) * x[2];
}
template <int CAP, int NJ, class ROW, class GOUT>
DRM_HD void rnea_backward_chain2(ROW row, bool gravity, bool damping, const f2 (&q)[NJ], const f2 (&qd)[NJ], const f2 (&qdd)[NJ],
                                 const f2 (&gtau)[NJ], GOUT gout) {
    f2 cs[NJ], sn[NJ];
    chain_trig2<NJ>(q, cs, sn);
    const float g = gravity ? 9.81f : 0.0f;
    const f2 zero = f2_bcast(0.0f);
    auto joint = [&](int k, Joint2 &J) {
        const OpFT o = load_ft(row_ft(row, k));
        if (k < NJ) joint2_moving(o.F, cs[k], sn[k], J);
        else joint2_fixed(o.F, J);
    };
    // ---- A: motions up the chain (only the tip's is kept);  C: adjoints of the total forces (tbar), likewise ------------
    Motion2 M;
    f2 Tl[3] = {zero, zero, zero}, Ta[3] = {zero, zero, zero};
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        const OpFT o = load_ft(row_ft(row, k));
        Joint2 J;
        joint(k, J);
        const f2 wj = k < NJ ? qd[k] : zero, aj = k < NJ ? qdd[k] : zero;
        if (k == 0) rnea2_first_motion(J, o.F, g, wj, aj, M);
        else rnea2_link_motion(J, o.F, o.t, wj, aj, M, M);
        if (k > 0) { // tbar_k = (J^T (ubar.lin + ubar.ang x t), J^T ubar.ang), ubar = tbar_(k-1)
            f2 x[3], y[3];
            cross2_vc(Ta, o.t, x);
#pragma unroll
            for (int i = 0; i < 3; ++i) x[i] += Tl[i];
            joint2_T(J, o.F, x, y);
            joint2_T(J, o.F, Ta, x);
#pragma unroll
            for (int i = 0; i < 3; ++i) { Tl[i] = y[i]; Ta[i] = x[i]; }
        }
        if (k < NJ) Ta[2] += gtau[k];
    }
    // ---- D (with B): back to the root ----------------------------------------------------------------------------
    f2 wb[3] = {zero, zero, zero}, vb[3] = {zero, zero, zero}, alb[3] = {zero, zero, zero}, ab[3] = {zero, zero, zero};
    Force2 carry;
#pragma unroll
    for (int i = 0; i < 3; ++i) { carry.f[i] = zero; carry.n[i] = zero; }
#pragma unroll
    for (int k = CAP - 1; k >= 0; --k) {
        const float *of = row(k);
        const OpFT o = load_ft(row_ft(row, k));
        const float m = of[DRM_OPF_MASS], *mc = of + DRM_OPF_MCOM, *Io = of + DRM_OPF_IO;
        Joint2 J;
        joint(k, J);
        const f2 wj = k < NJ ? qd[k] : zero, aj = k < NJ ? qdd[k] : zero;
        // parent's motion and force adjoint from this link's (inverses of rnea2_link_motion / of sweep C; J is orthogonal)
        Motion2 P;
        f2 Ul[3], Ua[3];
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { P.w[i] = zero; P.al[i] = zero; P.v[i] = zero; P.a[i] = zero; Ul[i] = zero; Ua[i] = zero; }
            P.a[2] = f2_bcast(g);
        } else {
            f2 x[3], y[3], c[3];
            x[0] = M.w[0]; x[1] = M.w[1]; x[2] = M.w[2] - wj;
            joint2_N(J, o.F, x, P.w);
            x[0] = M.al[0] - M.w[1] * wj; x[1] = M.al[1] + M.w[0] * wj; x[2] = M.al[2] - aj;
            joint2_N(J, o.F, x, P.al);
            joint2_N(J, o.F, M.v, y);
            cross2_vc(P.w, o.t, c);
#pragma unroll
            for (int i = 0; i < 3; ++i) P.v[i] = y[i] - c[i];
            x[0] = M.a[0] - M.v[1] * wj; x[1] = M.a[1] + M.v[0] * wj; x[2] = M.a[2];
            joint2_N(J, o.F, x, y);
            cross2_vc(P.al, o.t, c);
#pragma unroll
            for (int i = 0; i < 3; ++i) P.a[i] = y[i] - c[i];
            x[0] = Ta[0]; x[1] = Ta[1]; x[2] = k < NJ ? Ta[2] - gtau[k < NJ ? k : 0] : Ta[2];
            joint2_N(J, o.F, Tl, Ul);
            joint2_N(J, o.F, x, Ua);
            cross2_vc(Ua, o.t, c);
#pragma unroll
            for (int i = 0; i < 3; ++i) Ul[i] -= c[i];            // ubar.lin = J tbar.lin - ubar.ang x t
        }
        // body force of the link from the recovered motion (+ what came up): hl = m v - mc x w, ha = Io w + mc x v, likewise (gl, ga)
        f2 hl[3], ha[3], x[3], y[3];
        Force2 tot;
        cross2_cv(mc, M.w, x);
#pragma unroll
        for (int i = 0; i < 3; ++i) hl[i] = f2_bcast(m) * M.v[i] - x[i];
        cross2_cv(mc, M.v, x);
#pragma unroll
        for (int r = 0; r < 3; ++r)
            ha[r] = f2_bcast(Io[r * 3 + 0]) * M.w[0] + f2_bcast(Io[r * 3 + 1]) * M.w[1] + f2_bcast(Io[r * 3 + 2]) * M.w[2] + x[r];
        {
            f2 gl[3], ga[3];
            cross2_cv(mc, M.al, x);
#pragma unroll
            for (int i = 0; i < 3; ++i) gl[i] = f2_bcast(m) * M.a[i] - x[i];
            cross2_cv(mc, M.a, x);
#pragma unroll
            for (int r = 0; r < 3; ++r)
                ga[r] = f2_bcast(Io[r * 3 + 0]) * M.al[0] + f2_bcast(Io[r * 3 + 1]) * M.al[1] + f2_bcast(Io[r * 3 + 2]) * M.al[2] + x[r];
            cross2_vv(M.w, hl, x);
#pragma unroll
            for (int i = 0; i < 3; ++i) tot.f[i] = gl[i] + x[i] + carry.f[i];
            cross2_vv(M.w, ha, x);
            cross2_vv(M.v, hl, y);
#pragma unroll
            for (int i = 0; i < 3; ++i) tot.n[i] = ga[i] + (x[i] + y[i]) + carry.n[i];
        }
        // the link's adjoint (rnea_link_adjoint: fl = tbar.lin, fa = tbar.ang; wb / vb / alb / ab arrive from the child)
        f2 hlb[3] = {zero, zero, zero}, hab[3] = {zero, zero, zero};
        acc2_cross_vv(hlb, Tl, M.w); acc2_cross_vv(hlb, Ta, M.v);
        acc2_cross_vv(hab, Ta, M.w);
        acc2_cross_vv(wb, hl, Tl); acc2_cross_vv(wb, ha, Ta);
        acc2_cross_vv(vb, hl, Ta);
#pragma unroll
        for (int i = 0; i < 3; ++i) vb[i] += f2_bcast(m) * hlb[i];
        acc2_cross_vc(wb, hlb, mc, -1.0f);
        acc2_matT_c(wb, Io, hab);
        acc2_cross_vc(vb, hab, mc, 1.0f);
#pragma unroll
        for (int i = 0; i < 3; ++i) ab[i] += f2_bcast(m) * Tl[i];
        acc2_cross_vc(alb, Tl, mc, -1.0f);
        acc2_matT_c(alb, Io, Ta);
        acc2_cross_vc(ab, Ta, mc, 1.0f);
        // the link's motion from its parent's (revolute joint about +z): adjoint
        f2 Pwb[3], Pvb[3], Palb[3], Pab[3];
        f2 gq = ab[0] * (M.a[1] + M.v[0] * wj) - ab[1] * (M.a[0] - M.v[1] * wj);
        f2 wjb = ab[0] * M.v[1] - ab[1] * M.v[0];
        joint2_N(J, o.F, ab, Pab);
        vb[1] += ab[0] * wj; vb[0] -= ab[1] * wj;
        cross2_cv(o.t, Pab, Palb);
        joint2_N(J, o.F, alb, x);
#pragma unroll
        for (int i = 0; i < 3; ++i) Palb[i] += x[i];
        gq += alb[0] * (M.al[1] + M.w[0] * wj) - alb[1] * (M.al[0] - M.w[1] * wj);
        const f2 ajb = alb[2];
        wb[1] += alb[0] * wj; wb[0] -= alb[1] * wj;
        wjb += alb[0] * M.w[1] - alb[1] * M.w[0];
        joint2_N(J, o.F, vb, Pvb);
        gq += vb[0] * M.v[1] - vb[1] * M.v[0];
        cross2_cv(o.t, Pvb, Pwb);
        joint2_N(J, o.F, wb, x);
#pragma unroll
        for (int i = 0; i < 3; ++i) Pwb[i] += x[i];
        gq += wb[0] * M.w[1] - wb[1] * M.w[0];
        wjb += wb[2];
        if (k > 0) gq += tot.f[0] * Tl[1] - tot.f[1] * Tl[0] + tot.n[0] * Ta[1] - tot.n[1] * Ta[0];
        if (k < NJ) gout(k, gq, damping ? wjb + f2_bcast(of[DRM_OPF_DAMP]) * gtau[k < NJ ? k : 0] : wjb, ajb);
        if (k > 0) { // the sub-tree's force in the parent's frame
            Force2 up;
            joint2_N(J, o.F, tot.f, up.f);
            joint2_N(J, o.F, tot.n, up.n);
            cross2_cv(o.t, up.f, x);
#pragma unroll
            for (int i = 0; i < 3; ++i) { carry.f[i] = up.f[i]; carry.n[i] = up.n[i] + x[i]; }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            wb[i] = Pwb[i]; vb[i] = Pvb[i]; alb[i] = Palb[i]; ab[i] = Pab[i];
            Tl[i] = Ul[i]; Ta[i] = Ua[i];
        }
        M = P;
    }
}

} // namespace drm
