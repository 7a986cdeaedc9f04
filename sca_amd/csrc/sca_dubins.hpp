// sca_dubins.hpp -- native host-side replacement of SCA's preferred-velocity tracker.
//
// The reference computes v_pref for SCAPolicy / RVO3dDubinsPolicy with a per-agent, stateful tracker over a sampled 3-D
// Dubins path (mamp/policies/sca/scaPolicy.py:92-104,243-338), planned by dubinsmaneuver3d.py:34-162 on top of the 2-D
// planner dubinsmaneuver2d.py:33-218,260-297.  This file is a C++ restatement that follows the Python statement by
// statement and is compiled twice from the same text:
//   * for the host (sca_tracker_*): thread-parallel over agents, libm through function pointers so that the compiler cannot
//     fold pow(x, 2) or fuse sin/cos -- bit-exact against the reference on the fixtures (tests/test_tracker.py);
//   * for gfx950 (sca_device_tracker_*, kernels in sca_tracker.hip.h): one lane per agent, state resident in HBM.  The
//     device's sin / cos / atan2 / acos are not glibc's (they differ in the last bit in a few percent of the calls), so
//     the device tracker is the reference's algorithm to within rounding noise, not bit for bit: see DESIGN.md.
// generate_course (dubinsmaneuver2d.py:221-257) is not reproduced: the 3-D planner never reads its output.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "sca_glibc_math.h"

#if defined(__HIP_DEVICE_COMPILE__)
// device: everything inlined into its kernel whatever the number of kernels that use it (with two callers the inliner left
// the planner as a function: 31 more registers and scratch in k_replan)
#define SCA_DHD __host__ __device__ __attribute__((always_inline))
#elif defined(__HIPCC__)
#define SCA_DHD __host__ __device__
#else
#define SCA_DHD
#endif

namespace sca_dubins {

// The libm.  The reference's math.sin / cos / atan2 / acos and ** 2 are glibc's; sca_glibc_math.h restates those five functions
// operation for operation (validated against the running glibc bit for bit), so the tracker computes the reference's bits on
// the host AND on the device -- the device library's own functions (used until round 2) are within an ulp of glibc's, which a
// search that ends on comparisons of nearly equal path lengths turns into another v_pref every few hundred plans.
// Host: through volatile pointers (no folding across calls; tests swap them, sca_selftest_libm_noise).
static double h_pow_impl(double x, double y) { return y == 2.0 ? sca_gm::g_pow2(x) : std::pow(x, y); }
static double (*volatile h_pow)(double, double) = h_pow_impl;
static double (*volatile h_sin)(double) = sca_gm::g_sin;
static double (*volatile h_cos)(double) = sca_gm::g_cos;
static double (*volatile h_acos)(double) = sca_gm::g_acos;
static double (*volatile h_atan2)(double, double) = sca_gm::g_atan2;
#if defined(__HIP_DEVICE_COMPILE__)
SCA_DHD static inline double m_pow(double x, double) { return sca_gm::g_pow2(x); }   // only ever called with exponent 2
SCA_DHD static inline double m_sin(double x) { return sca_gm::g_sin(x); }
SCA_DHD static inline double m_cos(double x) { return sca_gm::g_cos(x); }
SCA_DHD static inline double m_acos(double x) { return sca_gm::g_acos(x); }
SCA_DHD static inline double m_atan2(double y, double x) { return sca_gm::g_atan2(y, x); }
SCA_DHD static inline void m_sincos(double x, double &s, double &c) { sca_gm::g_sincos(x, s, c); }
#else
SCA_DHD static inline double m_pow(double x, double y) { return h_pow(x, y); }
SCA_DHD static inline double m_sin(double x) { return h_sin(x); }
SCA_DHD static inline double m_cos(double x) { return h_cos(x); }
SCA_DHD static inline double m_acos(double x) { return h_acos(x); }
SCA_DHD static inline double m_atan2(double y, double x) { return h_atan2(y, x); }
SCA_DHD static inline void m_sincos(double x, double &s, double &c) { s = h_sin(x); c = h_cos(x); }
#endif

static const double PI = 3.141592653589793;
SCA_DHD static inline double fma3(const double *a, const double *b) { return std::fma(a[2], b[2], std::fma(a[1], b[1], a[0] * b[0])); }
// util.py:113  theta - 2.0 * pi * floor(theta / 2.0 / pi), literally (the planner's search calls it ~20 times per candidate
// radius; a division-free floor was tried on the device and dropped)
#if defined(__HIP_DEVICE_COMPILE__)
// Device: the quotient fl(fl(t / 2) / pi) by the constant-divisor sequence instead of the division macro (a quarter-rate
// reciprocal, two Newton steps, scale and fix-up: a quarter of the planner's arithmetic at ~20 calls per candidate radius):
//     q0 = x * fl(1/pi);   r = fma(-q0, pi, x)  (the residual of a quotient good to an ulp is exact);   q1 = fma(r, fl(1/pi), q0)
// q1 is the real quotient rounded once, after a relative perturbation below 2^-104: it is the correctly rounded quotient
// unless that lies within 2^-53 ulp of a rounding midpoint, and even then its floor is the same unless the midpoint sits
// just below an integer -- no such input exists at a rate that matters (< 1e-30 per call).  No branch: a guarded form that fell
// back to the division for a whole wavefront cost 68 spilled registers and 15 % of the kernel (measured, round 2).
SCA_DHD static inline double mod2pi(double t) {
    const double x = t * 0.5;                                            // exact
    const double q0 = x * 0.3183098861837907;
    const double r = std::fma(-q0, PI, x);
    const double q1 = std::fma(r, 0.3183098861837907, q0);
    return t - 2.0 * PI * std::floor(q1);
}
#else
SCA_DHD static inline double mod2pi(double t) { return t - 2.0 * PI * std::floor(t / 2.0 / PI); }
#endif
// Python round(x, 5): correctly rounded (see sca_core.h round5_py)
SCA_DHD static inline double round5_py(double x) {
    const double y = x * 100000.0;
    const double e = std::fma(x, 100000.0, -y);
    double r = std::rint(y);
    const double d = y - r;
    if (d == 0.5) { if (e > 0.0) r += 1.0; }
    else if (d == -0.5) { if (e < 0.0) r -= 1.0; }
    return r / 100000.0;
}
SCA_DHD static inline double round5_np(double x) { return std::rint(x * 100000.0) / 100000.0; }
SCA_DHD static inline double trunc5(double x) { double t = std::trunc(x * 100000.0); if (t == 0.0) t = 0.0; return t / 100000.0; }
SCA_DHD static inline double l3norm(const double *a, const double *b) {                                              // util.py:104
    return round5_py(std::sqrt(m_pow(a[0] - b[0], 2.0) + m_pow(a[1] - b[1], 2.0) + m_pow(a[2] - b[2], 2.0)));
}
// one 2-D maneuver: what the 3-D planner and the sampler read of dubinsmaneuver2d's result (start yaw, radius, the first two
// segment lengths, total length, word)
struct Maneuver2D { double yaw, r_min, t, p, length; char mode[3]; bool ok; };

// dubinsmaneuver2d.py:33-145: one candidate word
// sa..c_ab are the five trig values every planner of the reference recomputes (same arguments, same libm: same bits)
SCA_DHD static bool word(int which, double alpha, double beta, double d, double sa, double sb, double ca, double cb, double c_ab,
                 double &t, double &p, double &q, char mode[3]) {
    switch (which) {
    case 0: {                                                                                                // LSL :33-51
        mode[0] = 'L'; mode[1] = 'S'; mode[2] = 'L';
        const double tmp0 = d + sa - sb;
        const double p2 = 2 + (d * d) - (2 * c_ab) + (2 * d * (sa - sb));
        if (p2 < 0) return false;
        const double tmp1 = m_atan2((cb - ca), tmp0);
        t = mod2pi(-alpha + tmp1); p = std::sqrt(p2); q = mod2pi(beta - tmp1);
        return true;
    }
    case 1: {                                                                                                // RSR :54-71
        mode[0] = 'R'; mode[1] = 'S'; mode[2] = 'R';
        const double tmp0 = d - sa + sb;
        const double p2 = 2 + (d * d) - (2 * c_ab) + (2 * d * (sb - sa));
        if (p2 < 0) return false;
        const double tmp1 = m_atan2((ca - cb), tmp0);
        t = mod2pi(alpha - tmp1); p = std::sqrt(p2); q = mod2pi(-beta + tmp1);
        return true;
    }
    case 2: {                                                                                                // LSR :74-90
        mode[0] = 'L'; mode[1] = 'S'; mode[2] = 'R';
        const double p2 = -2 + (d * d) + (2 * c_ab) + (2 * d * (sa + sb));
        if (p2 < 0) return false;
        p = std::sqrt(p2);
        const double tmp2 = m_atan2((-ca - cb), (d + sa + sb)) - m_atan2(-2.0, p);
        t = mod2pi(-alpha + tmp2); q = mod2pi(-mod2pi(beta) + tmp2);
        return true;
    }
    case 3: {                                                                                                // RSL :93-109
        mode[0] = 'R'; mode[1] = 'S'; mode[2] = 'L';
        const double p2 = (d * d) - 2 + (2 * c_ab) - (2 * d * (sa + sb));
        if (p2 < 0) return false;
        p = std::sqrt(p2);
        const double tmp2 = m_atan2((ca + cb), (d - sa - sb)) - m_atan2(2.0, p);
        t = mod2pi(alpha - tmp2); q = mod2pi(beta - tmp2);
        return true;
    }
    case 4: {                                                                                                // RLR :112-127
        mode[0] = 'R'; mode[1] = 'L'; mode[2] = 'R';
        const double tmp = (6.0 - d * d + 2.0 * c_ab + 2.0 * d * (sa - sb)) / 8.0;
        if (std::fabs(tmp) > 1.0) return false;
        p = mod2pi(2 * PI - m_acos(tmp));
        t = mod2pi(alpha - m_atan2(ca - cb, d - sa + sb) + mod2pi(p / 2.0));
        q = mod2pi(alpha - beta - t + mod2pi(p));
        return true;
    }
    default: {                                                                                               // LRL :130-145
        mode[0] = 'L'; mode[1] = 'R'; mode[2] = 'L';
        const double tmp = (6. - d * d + 2 * c_ab + 2 * d * (-sa + sb)) / 8.;
        if (std::fabs(tmp) > 1) return false;
        p = mod2pi(2 * PI - m_acos(tmp));
        t = mod2pi(-alpha - m_atan2(ca - cb, d + sa - sb) + p / 2.);
        q = mod2pi(mod2pi(beta) - alpha - t + mod2pi(p));
        return true;
    }
    }
}

// The four CSC words (LSL, RSR, LSR, RSL = word 0..3 above) as ONE instruction stream with per-word signs, for the device's
// four-lanes-per-plan kernel (sca_tracker.hip.h: a lane evaluates the word of its index).  Only exact transformations of the
// literal statements are used -- negation commutes with rounding, a - b == a + (-b), a + b == b + a, x - (+0) == x -- so the
// result equals word(w, ...) bit for bit on any IEEE machine (checked on the host: sca_selftest_dubins_words).
//   S = (+-sa) + (+-sb) is the bracket of p2 and of atan2's x;  y = (+-ca) + (+-cb);  the second atan2 (LSR / RSL only)
//   becomes atan2(0, p) = +0 for LSL / RSR.
// Returns false (word not feasible) exactly when word() does; t, p, q are computed unconditionally (nan when infeasible), so
// that lanes of one plan never diverge.
// W_KNOWN: w is a compile-time constant at the call (the lane-per-plan planner unrolls the four words): LSL / RSR then skip
// their second arctangent, which is atan2(+0, p) = +0 for every p >= 0 (p2 = -0 cannot occur: 2 + d^2 > 0 and an exact
// cancellation rounds to +0), and A1 - (+0) == A1 bit for bit.
template <bool W_KNOWN = false>
SCA_DHD static bool csc_word_uniform(int w, double alpha, double beta, double mbeta /* mod2pi(beta) */, double d, double sa,
                                     double sb, double ca, double cb, double c_ab, double &t, double &p, double &q) {
    const bool cross = w >= 2;                         // LSR, RSL
    const bool rfirst = (w & 1) != 0;                  // RSR, RSL start with a right turn
    const double u = rfirst ? -sa : sa;                // LSL(+,-) RSR(-,+) LSR(+,+) RSL(-,-)
    const double v = (w == 0 || w == 3) ? -sb : sb;
    const double S = u + v;
    const double k2 = cross ? -2.0 : 2.0;
    const double cab2 = 2 * c_ab;
    const double p2 = ((k2 + (d * d)) + (cross ? cab2 : -cab2)) + ((2 * d) * S);
    const double x = (d + u) + v;
    const double ya = (w == 0 || w == 2) ? -ca : ca;   // LSL(-,+) RSR(+,-) LSR(-,-) RSL(+,+)
    const double yb = (w == 1 || w == 2) ? -cb : cb;
    const double y = ya + yb;
    p = sca_gm::sqrt_(p2);
    const double A1 = m_atan2(y, x);
    double tmp;
    if (W_KNOWN && !cross) tmp = A1;
    else {
        const double A2 = m_atan2(cross ? (w == 2 ? -2.0 : 2.0) : 0.0, p);
        tmp = A1 - A2;
    }
    const double ta = tmp - alpha;
    t = mod2pi(rfirst ? -ta : ta);
    const double qa = (w == 2 ? mbeta : beta) - tmp;
    q = mod2pi((w == 1 || w == 2) ? -qa : qa);
    return !(p2 < 0);
}

// dubins_path_planning (:179-218) + dubins_path_planning_from_origin (:148-176), in two parts: the frame depends on the end
// points only, so the 3-D planner's search over the horizontal radius (dubinsmaneuver3d.py:52-100, ~50-100 calls with the same
// end points) computes it once -- same arguments, same library functions, same bits as recomputing it every time
struct Frame2D { double D, alpha, beta, sa, sb, ca, cb, c_ab; };
SCA_DHD static Frame2D frame2d(const double start[3], const double end[3]) {
    Frame2D F;
    const double ex = end[0] - start[0], ey = end[1] - start[1];
    const double syaw = start[2], eyaw = end[2];
    F.D = std::sqrt(m_pow(ex, 2.0) + m_pow(ey, 2.0));
    const double theta = mod2pi(m_atan2(ey, ex));
    F.alpha = mod2pi(syaw - theta);
    F.beta = mod2pi(eyaw - theta);
    m_sincos(F.alpha, F.sa, F.ca);
    m_sincos(F.beta, F.sb, F.cb);
    F.c_ab = m_cos(F.alpha - F.beta);
    return F;
}
#if defined(__HIP_DEVICE_COMPILE__)
// The four CSC words of one 2-D plan at once, for the lane-per-plan kernels: csc_word_uniform<true> for w = 0 .. 3 with its six
// arctangents taken out of the words and evaluated side by side through the branch-free form of sca_glibc_math.h -- one
// straight-line block the scheduler can interleave, and ONE branch behind it for arguments outside that form's domain
// (non-finite, denormal, beyond 2^+-500) instead of one inside every arctangent.  Same expressions, same values, same bits.
SCA_DHD static void csc_words4(const Frame2D &F, double mbeta, double d, double t4[4], double p4[4], double q4[4], bool ok4[4]) {
    const double cab2 = 2 * F.c_ab, d2 = d * d, dd = 2 * d;
    bool dom = true;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const bool cross = w >= 2, rfirst = (w & 1) != 0;
        const double u = rfirst ? -F.sa : F.sa;
        const double v = (w == 0 || w == 3) ? -F.sb : F.sb;
        const double S = u + v;
        const double k2 = cross ? -2.0 : 2.0;
        const double p2 = ((k2 + d2) + (cross ? cab2 : -cab2)) + (dd * S);
        const double x = (d + u) + v;
        const double ya = (w == 0 || w == 2) ? -F.ca : F.ca;
        const double yb = (w == 1 || w == 2) ? -F.cb : F.cb;
        const double y = ya + yb;
        ok4[w] = !(p2 < 0);
        p4[w] = sca_gm::sqrt_(p2);
        double tmp = sca_gm::atan2_core(y, x, dom);
        // LSL / RSR: their second arctangent is atan2(+0, p) = +0 and A1 - (+0) == A1; an infeasible word's is never read
        if (cross) tmp = tmp - sca_gm::atan2_core(w == 2 ? -2.0 : 2.0, ok4[w] ? p4[w] : 1.0, dom);
        const double ta = tmp - F.alpha;
        t4[w] = mod2pi(rfirst ? -ta : ta);
        const double qa = (w == 2 ? mbeta : F.beta) - tmp;
        q4[w] = mod2pi((w == 1 || w == 2) ? -qa : qa);
#ifndef SCA_WORDS_PER_GROUP
#define SCA_WORDS_PER_GROUP 2
#endif
        // how many words the scheduler may interleave (all four: it runs out of registers and spills)
        if (((w + 1) % SCA_WORDS_PER_GROUP) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    if (!dom) {                                                             // (cold) an argument outside the branch-free form's domain
#pragma unroll
        for (int w = 0; w < 4; w++) ok4[w] = csc_word_uniform<true>(w, F.alpha, F.beta, mbeta, d, F.sa, F.sb, F.ca, F.cb, F.c_ab, t4[w], p4[w], q4[w]);
    }
}
#endif

SCA_DHD static Maneuver2D plan2d(const Frame2D &F, double yaw, double c) {
    Maneuver2D m;
    m.yaw = yaw;
    m.r_min = c; m.t = m.p = -1.0; m.length = INFINITY; m.ok = false;
    m.mode[0] = m.mode[1] = m.mode[2] = 0;
    const double d = F.D / c;
    double bcost = INFINITY;
#if defined(__HIP_DEVICE_COMPILE__)
    // device: the four CSC words through the branch-free uniform form (bit-identical to word(0..3)), so that the four
    // independent chains can be interleaved by the instruction scheduler
    {
        const double mbeta = mod2pi(F.beta);
        double t4[4], p4[4], q4[4]; bool ok4[4];
#if defined(SCA_V_WORDS) && SCA_V_WORDS == 1
        csc_words4(F, mbeta, d, t4, p4, q4, ok4);
#else
#pragma unroll
        for (int w = 0; w < 4; w++) ok4[w] = csc_word_uniform<true>(w, F.alpha, F.beta, mbeta, d, F.sa, F.sb, F.ca, F.cb, F.c_ab, t4[w], p4[w], q4[w]);
#endif
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const double cost = c * (std::fabs(t4[w]) + std::fabs(p4[w]) + std::fabs(q4[w]));
            if (ok4[w] && bcost > cost) {
                m.t = t4[w]; m.p = p4[w]; bcost = cost; m.ok = true;
                m.mode[0] = (w & 1) ? 'R' : 'L'; m.mode[1] = 'S'; m.mode[2] = (w == 0 || w == 3) ? 'L' : 'R';
            }
        }
    }
    for (int w = 4; w < 6; w++) {
#else
    for (int w = 0; w < 6; w++) {                                     // planners = [LSL, RSR, LSR, RSL, RLR, LRL]
#endif
        double t, p, q; char mode[3];
        if (!word(w, F.alpha, F.beta, d, F.sa, F.sb, F.ca, F.cb, F.c_ab, t, p, q, mode)) continue;
        const double cost = c * (std::fabs(t) + std::fabs(p) + std::fabs(q));
        if (bcost > cost) { m.t = t; m.p = p; m.mode[0] = mode[0]; m.mode[1] = mode[1]; m.mode[2] = mode[2]; bcost = cost; m.ok = true; }
    }
    m.length = bcost;
    return m;
}

// get_position_in_segment (:283-297) / get_coordinates (:260-280).  The reference recomputes the two segment boundaries of
// the maneuver (and the sin / cos of every segment's start angle) for every sample; they are functions of the maneuver only,
// so they are computed once per plan (PathFrame) -- same arguments, same library functions, same bits.
SCA_DHD static void seg(double offset, const double qi[3], double s0, double c0, char mode, double q[3]) {   // s0, c0 = sin, cos of qi[2]
    q[0] = q[1] = q[2] = 0.0;
    if (mode == 'L') {
        double s1, c1;
        m_sincos(qi[2] + offset, s1, c1);
        q[0] = qi[0] + s1 - s0;
        q[1] = qi[1] - c1 + c0;
        q[2] = qi[2] + offset;
    } else if (mode == 'R') {
        double s1, c1;
        m_sincos(qi[2] - offset, s1, c1);
        q[0] = qi[0] - s1 + s0;
        q[1] = qi[1] + c1 - c0;
        q[2] = qi[2] - offset;
    } else if (mode == 'S') {
        q[0] = qi[0] + c0 * offset;
        q[1] = qi[1] + s0 * offset;
        q[2] = qi[2];
    }
}
struct PathFrame { double s0, c0, q1[3], s1, c1, q2[3], s2, c2; };   // start of segment 1 / 2 / 3: state and sin / cos of its angle
SCA_DHD static PathFrame path_frame(const Maneuver2D &m) {
    PathFrame F;
    const double qi[3] = {0., 0., m.yaw};
    m_sincos(qi[2], F.s0, F.c0);
    seg(m.t, qi, F.s0, F.c0, m.mode[0], F.q1);
    m_sincos(F.q1[2], F.s1, F.c1);
    seg(m.p, F.q1, F.s1, F.c1, m.mode[1], F.q2);
    m_sincos(F.q2[2], F.s2, F.c2);
    return F;
}
SCA_DHD static void get_coordinates(const Maneuver2D &m, const PathFrame &F, double offset, double q[3]) {
    const double noffset = offset / m.r_min;
    const double qi[3] = {0., 0., m.yaw};
    const double l1 = m.t, l2 = m.p;
    if (noffset < l1) seg(noffset, qi, F.s0, F.c0, m.mode[0], q);
    else if (noffset < (l1 + l2)) seg(noffset - l1, F.q1, F.s1, F.c1, m.mode[1], q);
    else seg(noffset - l1 - l2, F.q2, F.s2, F.c2, m.mode[2], q);
    q[0] = q[0] * m.r_min + qi[0];
    q[1] = q[1] * m.r_min + qi[1];
    q[2] = mod2pi(q[2]);
}

// sample i of a path made of a horizontal and a vertical maneuver, [x, y, z, psi, gamma] (compute_sampling, dubinsmaneuver3d.py:116-132)
SCA_DHD static void sample_path(const Maneuver2D &h, const PathFrame &fh, const Maneuver2D &v, const PathFrame &fv, const double qi[5],
                                double ran, double s[5]) {
    double qSZ[3], qXY[3];
    get_coordinates(v, fv, ran, qSZ);
    get_coordinates(h, fh, qSZ[0], qXY);
    s[0] = qXY[0] + qi[0]; s[1] = qXY[1] + qi[1]; s[2] = qSZ[1] + qi[2]; s[3] = qXY[2]; s[4] = qSZ[2];
}
struct Plan3D {
    Maneuver2D h{}, v{};
    PathFrame fh{}, fv{};
    double length = -1.0, sampling_size = 0.1;
    double qi[5] = {0, 0, 0, 0, 0};
    char mode[7] = {0};
    bool ok = false;
    int iters = 0;                       // candidate radii the search tried (statistics; sca_*_tracker_debug)
    long count = 0;                      // number of samples compute_sampling (dubinsmaneuver3d.py:116-132) would produce
    // sample i of the path: a pure function of i, so the tracker evaluates samples on demand
    // (the reference materialises all ~1000 of them at every re-plan and then discards most)
    SCA_DHD void sample(long i, double s[5]) const { sample_path(h, fh, v, fv, qi, (double)i * sampling_size, s); }
};
// the tail of dubinsmaneuver3d (:102-113) + compute_sampling's grid (:116-132) for the winning pair of maneuvers
SCA_DHD static void finish_plan(Plan3D &P, const Maneuver2D &fbh, const Maneuver2D &fbv, const double qi[5]) {
    P.h = fbh; P.v = fbv; P.length = fbv.length; P.ok = true;
    P.fh = path_frame(fbh); P.fv = path_frame(fbv);
    for (int k = 0; k < 3; k++) { P.mode[k] = fbh.mode[k]; P.mode[3 + k] = fbv.mode[k]; }
    P.mode[6] = 0;
    double ss = 0.1;
    if (P.length > 100) ss = P.length / 1000;
    P.sampling_size = ss;
    for (int k = 0; k < 5; k++) P.qi[k] = qi[k];
    const double stop = P.length + ss;
    const long cnt = (long)std::ceil(stop / ss);                        // len(np.arange(0, stop, ss))
    P.count = cnt > 0 ? cnt : 0;
}

// try_to_construct (dubinsmaneuver3d.py:135-162); returns the number of maneuvers (0 or 2)
// H = frame2d of the horizontal end points (qi[0,1,3] -> qf[0,1,3]), the same for every radius
// what try_to_construct computes from its arguments alone, the same for every candidate radius of a search: 1 / Rmin^2 of the
// vertical curvature (:143) and the squared height difference of the vertical frame (dubins_path_planning :184) -- same
// arguments, same functions, same bits as recomputing them per candidate
struct SearchConst { double inv_rmin2, dz2; };
SCA_DHD static SearchConst search_const(const double qi[5], const double qf[5], double Rmin) {
    SearchConst K;
    K.inv_rmin2 = 1.0 / m_pow(Rmin, 2.0);
    K.dz2 = m_pow(qf[2] - qi[2], 2.0);
    return K;
}
// frame2d for the vertical plane: start (0, z_i, pitch_i), end (len, z_f, pitch_f); dz2 = pow(z_f - z_i, 2) from SearchConst
SCA_DHD static Frame2D frame2d_vertical(double len, double dz, double dz2, double spitch, double epitch) {
    Frame2D F;
    const double ex = len - 0.0;
#if defined(__HIP_DEVICE_COMPILE__) && (!defined(SCA_V_FRAME) || SCA_V_FRAME == 1)
    // the branch-free forms with one domain flag for the whole frame (see csc_words4)
    bool dom = true;
    F.D = sca_gm::sqrt_(sca_gm::pow2_core(ex, dom) + dz2);
    const double theta = mod2pi(sca_gm::atan2_core(dz, ex, dom));
    F.alpha = mod2pi(spitch - theta);
    F.beta = mod2pi(epitch - theta);
    sca_gm::sincos_core(F.alpha, F.sa, F.ca, dom);
    sca_gm::sincos_core(F.beta, F.sb, F.cb, dom);
    F.c_ab = sca_gm::cos_core(F.alpha - F.beta, dom);
    if (dom) return F;
#endif
    F.D = std::sqrt(m_pow(ex, 2.0) + dz2);
    const double theta2 = mod2pi(m_atan2(dz, ex));
    F.alpha = mod2pi(spitch - theta2);
    F.beta = mod2pi(epitch - theta2);
    m_sincos(F.alpha, F.sa, F.ca);
    m_sincos(F.beta, F.sb, F.cb);
    F.c_ab = m_cos(F.alpha - F.beta);
    return F;
}
SCA_DHD static int try_to_construct(const Frame2D &H, const SearchConst &K, const double qi[5], const double qf[5], double Rmin,
                                    const double pitchlims[2], double hr, Maneuver2D &mh, Maneuver2D &mv) {
    // the vertical curvature first: when it fails (:146-147: a candidate clamped to Rmin, `c < 1 -> c = 1`, does every time) the
    // horizontal maneuver computed before it (:140) is never read by the caller, so it is not computed here
    (void)Rmin;
    const double vc = sca_gm::sqrt_(K.inv_rmin2 - 1.0 / m_pow(hr, 2.0));
    if (vc < 1e-5) return 0;
    mh = plan2d(H, qi[3], hr);
    const double vr = 1.0 / vc;
    mv = plan2d(frame2d_vertical(mh.length, qf[2] - qi[2], K.dz2, qi[4], qf[4]), qi[4], vr);
    if (mv.mode[0] == 'R' && mv.mode[1] == 'L' && mv.mode[2] == 'R') return 0;
    if (mv.mode[0] == 'R') { if (qi[4] - mv.t < pitchlims[0]) return 0; }
    else { if (qi[4] + mv.t > pitchlims[1]) return 0; }
    return 2;
}

// dubinsmaneuver3d (dubinsmaneuver3d.py:34-113) + compute_sampling (:116-132)
SCA_DHD static Plan3D plan3d(const double qi[5], const double qf[5], double Rmin, const double pitchlims[2]) {
    Plan3D P;
    double b = 1.0;
    Maneuver2D fbh, fbv, fch, fcv;
    const double qi2D[3] = {qi[0], qi[1], qi[3]}, qf2D[3] = {qf[0], qf[1], qf[3]};
    const Frame2D H = frame2d(qi2D, qf2D);
    const SearchConst K = search_const(qi, qf, Rmin);
    int nfb = try_to_construct(H, K, qi, qf, Rmin, pitchlims, Rmin * b, fbh, fbv);
    int guard = 0;
    P.iters = 1;
    while (nfb < 2) {
        b *= 2.0;
        nfb = try_to_construct(H, K, qi, qf, Rmin, pitchlims, Rmin * b, fbh, fbv);
        P.iters++;
        if (++guard > 200) return P;                                   // the reference would loop forever
    }
    double step = 0.1;
    while (std::fabs(step) > 1e-10) {                                      // the local search (:86-100)
        double c = b + step;
        if (c < 1.0) c = 1.0;
        P.iters++;
        const int nfc = try_to_construct(H, K, qi, qf, Rmin, pitchlims, Rmin * c, fch, fcv);
        if (nfc > 0 && fcv.length < fbv.length) { b = c; fbh = fch; fbv = fcv; step *= 2.; continue; }
        step *= -0.1;
    }
    finish_plan(P, fbh, fbv, qi);
    return P;
}

// ---- the tracker (scaPolicy.py:243-338) ------------------------------------------------------------------------------
struct AgentTrack {
    bool is_use_dubins = false;
    Plan3D plan;                        // agent.dubins_path == samples [next, plan.count) of this plan, popped in path order
    long next = 0;
    double now_goal[3] = {0, 0, 0};
    double sampling_size = 0.1;
    double v_pref[3] = {0, 0, 0};       // agent.v_pref (not truncated), read by is_parallel on the next call
    int replans = 0;
};

// what the tracker reads of the agents (agent.py:13-36): plain pointers, host or device
struct TrackView {
    const double *goal;           // [n*3] goal_global_frame
    const double *goal_heading;   // [n*3] goal_heading_frame
    const double *pref_speed;     // [n]
    const uint8_t *zaxis;         // [n] is_zAxis of scaPolicy.py:188-190 (condition_dist :300)
    double turning_radius, pitch_lo, pitch_hi, neighbor_dist;
};

struct Pool;
struct Tracker {
    int n = 0;
    Pool *pool = nullptr;
    std::vector<double> goal, goal_heading, pref_speed;
    std::vector<uint8_t> zaxis;
    double turning_radius = 1.5, pitchlims[2] = {-PI / 4, PI / 4}, neighbor_dist = 10.0;
    std::vector<AgentTrack> st;
    TrackView view() const {
        return TrackView{goal.data(), goal_heading.data(), pref_speed.data(), zaxis.data(), turning_radius, pitchlims[0], pitchlims[1],
                         neighbor_dist};
    }
};

// the end points of compute_dubins (scaPolicy.py:92-104): qi = pose now, qf = goal pose
SCA_DHD static void dubins_endpoints(TrackView T, int i, const double *pos, const double *heading, double qi[5], double qf[5]) {
    qi[0] = pos[0]; qi[1] = pos[1]; qi[2] = pos[2]; qi[3] = heading[0]; qi[4] = heading[1];
    qf[0] = T.goal[3 * i]; qf[1] = T.goal[3 * i + 1]; qf[2] = T.goal[3 * i + 2];
    qf[3] = T.goal_heading[3 * i]; qf[4] = T.goal_heading[3 * i + 1];
}
SCA_DHD static void adopt_plan(AgentTrack &a, const Plan3D &P) {
    a.plan = P;
    a.sampling_size = a.plan.sampling_size;
    a.next = 0;
    a.replans++;
}
SCA_DHD static void compute_dubins(TrackView T, AgentTrack &a, int i, const double *pos, const double *heading) {  // :92-104
    double qi[5], qf[5];
    dubins_endpoints(T, i, pos, heading, qi, qf);
    const double pl[2] = {T.pitch_lo, T.pitch_hi};
    adopt_plan(a, plan3d(qi, qf, T.turning_radius, pl));
}
SCA_DHD static bool path_empty(const AgentTrack &a) { return a.next >= a.plan.count; }
SCA_DHD static bool path_pop(AgentTrack &a, double out[3]) {
    if (path_empty(a)) return false;
    double s5[5];
    a.plan.sample(a.next++, s5);
    out[0] = s5[0]; out[1] = s5[1]; out[2] = s5[2];
    return true;
}
SCA_DHD static void node_pop4(AgentTrack &a) {                                                                    // :253-261
    // the four popped nodes are discarded: only the cursor moves (pop past the end is a no-op, as list.pop guarded by `if`)
    const long left = a.plan.count - a.next;
    a.next += left < 4 ? (left > 0 ? left : 0) : 4;
}
SCA_DHD static void update_dubins(TrackView T, AgentTrack &a, int i, const double *pos) {                         // :243-250
    const double dis = l3norm(pos, a.now_goal);
    if (dis < a.sampling_size * 2) {
        if (!path_pop(a, a.now_goal)) { a.now_goal[0] = T.goal[3 * i]; a.now_goal[1] = T.goal[3 * i + 1]; a.now_goal[2] = T.goal[3 * i + 2]; }
    }
}
// util.py:125-137 is_parallel(vA float32, v_pref float64)
SCA_DHD static bool is_parallel(const float *vA, const double *vp) {
    const float n1 = std::sqrt((float)((double)(float)(vA[0] * vA[0]) + (double)(float)(vA[1] * vA[1]) + (double)(float)(vA[2] * vA[2])));
    const double n2 = std::sqrt(fma3(vp, vp));
    const float v1[3] = {vA[0] / n1, vA[1] / n1, vA[2] / n1};
    const double v2[3] = {vp[0] / n2, vp[1] / n2, vp[2] / n2};
    if (n1 <= (float)1e-5 || n2 <= 1e-5) return true;
    const double v1d[3] = {(double)v1[0], (double)v1[1], (double)v1[2]};
    const double rv = 1.0 - std::fabs(fma3(v1d, v2));
    return round5_np(rv) < 3e-3;
}

// compute_v_pref (scaPolicy.py:264-338) for one agent, in three parts so that the device can run the (rare, long) re-plans
// of a step in a kernel of their own: decide -> [replan] -> finish.  nbr0_dsq < 0 means agent.neighbors is empty.
// track_decide: everything up to the choice between following the path and re-planning; returns true when the agent must
// re-plan (first call :283-287, off-track :322-327), otherwise dif is the vector to the tracked node.
SCA_DHD static bool track_decide(TrackView T, AgentTrack &a, int i, const double *pos, const float *vel, double nbr0_dsq,
                                 double dif[3]) {
    const double *goal = &T.goal[3 * i];
    const double dis_goal = l3norm(pos, goal);
    const double k = 3.0 * T.turning_radius;
    if (!a.is_use_dubins) {
        a.is_use_dubins = true;
        return true;
    }
    update_dubins(T, a, i, pos);
    const double dis = l3norm(pos, a.now_goal);
    const double max_size = round5_py(6 * a.sampling_size);
    const double pApG[3] = {goal[0] - pos[0], goal[1] - pos[1], goal[2] - pos[2]};
    const double vA64[3] = {(double)vel[0], (double)vel[1], (double)vel[2]};
    const float nvA = std::sqrt((float)((double)(float)(vel[0] * vel[0]) + (double)(float)(vel[1] * vel[1]) + (double)(float)(vel[2] * vel[2])));
    double cs = fma3(vA64, pApG) / ((double)nvA * std::sqrt(fma3(pApG, pApG)));
    if (!(cs < 1.0)) cs = 1.0;                                       // min(x, 1.0); nan -> 1.0 as Python's min does here
    if (cs < -1.0) cs = -1.0;                                        // the reference would raise; clamp
    const double theta = round5_py(m_acos(cs));
    const double deg100 = round5_np(100.0 * (PI / 180.0));
    const double min_dist_ob = nbr0_dsq >= 0 ? round5_py(std::sqrt(nbr0_dsq)) : std::rint(T.neighbor_dist);
    const bool condition_dist = T.zaxis[i] ? (min_dist_ob >= 2.0 * T.turning_radius) : false;
    if (((is_parallel(vel, a.v_pref) || dis_goal <= k) && dis < max_size) || (theta >= deg100) || condition_dist) {
        update_dubins(T, a, i, pos);
        if (!path_empty(a)) for (int q = 0; q < 3; q++) dif[q] = a.now_goal[q] - pos[q];
        else for (int q = 0; q < 3; q++) dif[q] = goal[q] - pos[q];
        return false;
    }
    return true;
}
// compute_dubins + dubins_path_node_pop + the first tracked node (:284-287, :323-327)
SCA_DHD static void track_replan(TrackView T, AgentTrack &a, int i, const double *pos, const double *heading, double dif[3]) {
    compute_dubins(T, a, i, pos, heading);
    node_pop4(a);
    path_pop(a, a.now_goal);
    for (int q = 0; q < 3; q++) dif[q] = a.now_goal[q] - pos[q];
}
// the same with a plan computed elsewhere (the device's four-lane planner)
SCA_DHD static void track_adopt(AgentTrack &a, const Plan3D &P, const double *pos, double dif[3]) {
    adopt_plan(a, P);
    node_pop4(a);
    path_pop(a, a.now_goal);
    for (int q = 0; q < 3; q++) dif[q] = a.now_goal[q] - pos[q];
}
// :329-338
SCA_DHD static void track_finish(TrackView T, AgentTrack &a, int i, const double *pos, const double dif[3], double *V_des) {
    const double *goal = &T.goal[3 * i];
    const double zero[3] = {0, 0, 0};
    const double norm = l3norm(dif, zero);
    double v[3];
    for (int q = 0; q < 3; q++) v[q] = dif[q] * T.pref_speed[i] / norm;
    if (l3norm(goal, pos) < 0.2) v[0] = v[1] = v[2] = 0.0;               // util.reached, bound 0.2
    for (int q = 0; q < 3; q++) { a.v_pref[q] = v[q]; V_des[q] = trunc5(v[q]); }
}
static void compute_v_pref(TrackView T, AgentTrack &a, int i, const double *pos, const float *vel, const double *heading,
                           double nbr0_dsq, double *V_des) {
    double dif[3];
    if (track_decide(T, a, i, pos, vel, nbr0_dsq, dif)) track_replan(T, a, i, pos, heading, dif);
    track_finish(T, a, i, pos, dif, V_des);
}

// persistent worker threads (spawning 64 threads per step cost more than the tracking itself at N = 1024)
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::function<void(int)> job;
    int n = 0, gen = 0, pending = 0;
    bool stop = false;
    void ensure(int want) {
        if ((int)th.size() == want) return;
        shutdown();
        n = want; stop = false;
        // a re-sized pool must not mistake the previous generations for work: a fresh thread that started from 0 ran the stale
        // `job` of the last run (found by ThreadSanitizer, tests/test_sanitizers.py)
        const int born = gen;
        for (int t = 0; t < want; t++) th.emplace_back([this, t, born] {
            int seen = born;
            for (;;) {
                std::function<void(int)> f;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || gen != seen; });
                    if (stop) return;
                    seen = gen; f = job;
                }
                f(t);
                { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) done_cv.notify_all(); }
            }
        });
    }
    void run(const std::function<void(int)> &f) {
        { std::lock_guard<std::mutex> lk(mu); job = f; pending = n; gen++; }
        cv.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return pending == 0; });
    }
    void shutdown() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &x : th) x.join();
        th.clear();
    }
    ~Pool() { shutdown(); }
};

static void step_all(Tracker &T, const double *pos, const float *vel, const double *heading, const uint8_t *active,
                     const double *nbr0_dsq, double *vpref_out, int nthreads) {
    const TrackView V = T.view();
    auto work = [&](int lo, int hi) {
        for (int i = lo; i < hi; i++)
            if (active[i]) compute_v_pref(V, T.st[i], i, pos + 3 * i, vel + 3 * i, heading + 3 * i, nbr0_dsq[i], vpref_out + 3 * i);
    };
    if (nthreads <= 1 || T.n < 32) { work(0, T.n); return; }
    if (!T.pool) T.pool = new Pool();
    T.pool->ensure(nthreads);
    // interleaved blocks of 16 agents: re-plans cluster in id ranges, contiguous chunks would be unbalanced
    std::atomic<int> next{0};
    const int blk = 16;
    T.pool->run([&](int) {
        for (;;) {
            const int lo = next.fetch_add(blk);
            if (lo >= T.n) return;
            work(lo, std::min(T.n, lo + blk));
        }
    });
}

}  // namespace sca_dubins
