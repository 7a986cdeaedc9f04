// sca_dubins.hpp -- native host-side replacement of SCA's preferred-velocity tracker.
//
// The reference computes v_pref for SCAPolicy / RVO3dDubinsPolicy with a per-agent, stateful tracker over a sampled 3-D
// Dubins path (mamp/policies/sca/scaPolicy.py:92-104,243-338), planned by dubinsmaneuver3d.py:34-162 on top of the 2-D
// planner dubinsmaneuver2d.py:33-218,260-297.  This file is a C++ restatement that follows the Python statement by
// statement and is compiled twice from the same text:
//   * for the host (sca_tracker_*): thread-parallel over agents -- bit-exact against the reference on the fixtures
//     (tests/test_tracker.py);
//   * for gfx950 (sca_device_tracker_*, kernels in sca_tracker.hip.h): one lane up to one wavefront per plan, state resident in
//     HBM.
// Both builds call the SAME libm: sca_glibc_math.h, glibc 2.35's sin / cos / atan2 / acos / pow(x, 2) restated operation for
// operation (the ROCm device library's differ from glibc's in the last bit in a few per cent of the calls, which a discrete radius
// search does not forgive).  Since round 3 the device tracker therefore equals the host tracker -- and the reference -- bit for
// bit: every v_pref, every follow-or-re-plan decision, every plan (tests/test_gpu_tracker.py, tests/test_gpu_value_parity.py).
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
// (t / 2 is never formed: scalings by two are exact, so q0 = t * (fl(1/pi) / 2), the residual against 2 pi is twice the one
// against pi, and the halved reciprocal takes it back: the same q1 bit for bit, one multiplication less)
SCA_DHD static inline double mod2pi(double t) {
    const double q0 = t * 0.15915494309189535;
    const double r = std::fma(-q0, 2.0 * PI, t);
    const double q1 = std::fma(r, 0.15915494309189535, q0);
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
#pragma unroll
        for (int w = 0; w < 4; w++) ok4[w] = csc_word_uniform<true>(w, F.alpha, F.beta, mbeta, d, F.sa, F.sb, F.ca, F.cb, F.c_ab, t4[w], p4[w], q4[w]);
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
    int rounds = 0;                      // rounds of the speculative search that tried them (k_replan_group<16 / 32 / 64>: several steps per round;
                                         // 0: a form that takes one step at a time).  Statistics only (sits in the record's padding).
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
#if defined(__HIP_DEVICE_COMPILE__)
    // the branch-free forms with one domain flag for the whole frame
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

#if !defined(SCA_PLAN_LEAN) || SCA_PLAN_LEAN == 1
// ---- the lane-per-plan search, lean (device; compiled for the host as well, where tests compare it with plan3d) --------------------------------------------------------------------------------
// plan3d above, for the kernels that give a plan one lane: the SAME candidates in the same order with the same verdicts, but
//   * a candidate is evaluated for what the search reads of it -- feasible or not, and the length -- and the winning radius is
//     constructed once more at the end for its maneuvers (the evaluation is a pure function of the radius): the four Maneuver2D
//     the literal loop carries (best / candidate x horizontal / vertical: 48 registers) are gone from the loop, which is what
//     lets the scheduler keep the four words of a 2-D plan in flight together instead of spilling;
//   * when every lane of the wavefront is FAR -- both 2-D problems with d = D / radius >= 7 (the benchmark circle's d are 10^4, and
//     26 for its large-radius plans) -- a candidate is one straight-line block.  For d >= 7 every arctangent is glibc's case (i)
//     (|y| <= 2 against x >= d - 2 >= 5, or 2 against p >= sqrt(d^2 - 4 d - 4) > 4): the polynomial piece when the quotient is
//     below 1/16, else the table piece, decided per row group with a ballot and selected per lane (atan_far_n: the same
//     operations on the same values as sca_gm::atan2_core takes for such arguments, incl. glibc's early return for exponents 57
//     or more apart); every word is feasible (p^2 >= 17); both CCC words are infeasible (their tmp <= -1.6).  Anything else -- a
//     lane that is not far, a quotient below 2^-1000 (the lean division has no scaling), a non-finite value -- sends the whole
//     wavefront's candidate through try_to_construct above (a called function, cold).
namespace lean {
using sca_gm::fma_;
static long long g_host_fast = 0, g_host_literal = 0;      // candidates by the lean block / by try_to_construct (host self-test only)
#if !defined(__HIP_DEVICE_COMPILE__)
#define SCA_LEAN_HOST_COUNT(x) (++(x))
#else
#define SCA_LEAN_HOST_COUNT(x) ((void)0)
#endif
// does any lane of the wavefront say so?  (host: the one caller)
SCA_DHD static inline bool any_says(bool b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(b) != 0;
#else
    return b;
#endif
}
#if defined(SCA_LEAN_STATS)
#define SCA_LEAN_STAT(i) atomicAdd(&::g_lean_stats[i], 1ull)
#else
#define SCA_LEAN_STAT(i) ((void)0)
#endif
#if defined(SCA_LEAN_TIMING)
#define SCA_LEAN_T(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#define SCA_LEAN_TACC(i, a, b) if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0 || true) { const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true); if ((int)(threadIdx.x & 63) == __builtin_ctzll(m_)) atomicAdd(&::g_lean_stats[i], (b) - (a)); }
#else
#define SCA_LEAN_T(v) ((void)0)
#define SCA_LEAN_TACC(i, a, b) ((void)0)
#endif
// sqrt for finite x >= 2^-767, x > 0 (sca_gm::sqrt_ without its select for +-0 / inf)
SCA_DHD static inline double sqrt_pos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y, h0 = y * 0.5;
    const double r0 = fma_(-h0, s0, 0.5);
    const double s1 = fma_(s0, r0, s0), h1 = fma_(h0, r0, h0);
    const double d0 = fma_(-s1, s1, x);
    const double s2 = fma_(d0, h1, s1);
    const double d1 = fma_(-s2, s2, x);
    return fma_(d1, h1, s2);
#else
    return std::sqrt(x);
#endif
}
// atan_far_n below: atan2(y, x) for 2^-400 < x < 2^100, 16 |y| < x -- case (i) of e_atan2.c with u < 1/16: the polynomial piece
SCA_DHD static inline bool far_d(double d) { return d >= 7.0 && d < 1.2676506002282294e30; }
// ---- lock-step forms -----------------------------------------------------------------------------------------------------
// A dependent fp64 fma on gfx950 returns after ~13 cycles while the SIMD could issue one every 4 (measured: a chain of fmas
// 5.7 ns per link, four interleaved chains 2.05 ns per fma, scratch/mb2/ilp.hip) -- and the compiler's scheduler, which believes
// the latency is the issue time, leaves Horner chains and Newton steps back to back.  So the N independent evaluations a 2-D
// problem consists of (four words: four square roots, six arctangents, eight mod2pi) are written as rows: statement k of
// all N evaluations, then statement k + 1 of all N ...  Same operations on the same values; only the order in the instruction
// stream differs.  SCA_LEAN_ROW (optional) pins the rows with scheduling barriers.
#if defined(SCA_LEAN_SB)
#define SCA_LEAN_ROW() __builtin_amdgcn_sched_barrier(0)
#else
#define SCA_LEAN_ROW() ((void)0)
#endif
#define SCA_ROW(N, ...) { _Pragma("unroll") for (int k = 0; k < (N); k++) { __VA_ARGS__; } SCA_LEAN_ROW(); }
template <int N> SCA_DHD static inline void sqrt_pos_n(const double (&x)[N], double (&out)[N]) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y[N], s[N], h[N], r[N], d[N];
    SCA_ROW(N, y[k] = __builtin_amdgcn_rsq(x[k]))
    SCA_ROW(N, s[k] = x[k] * y[k]; h[k] = y[k] * 0.5)
    SCA_ROW(N, r[k] = fma_(-h[k], s[k], 0.5))
    SCA_ROW(N, s[k] = fma_(s[k], r[k], s[k]); h[k] = fma_(h[k], r[k], h[k]))
    SCA_ROW(N, d[k] = fma_(-s[k], s[k], x[k]))
    SCA_ROW(N, s[k] = fma_(d[k], h[k], s[k]))
    SCA_ROW(N, d[k] = fma_(-s[k], s[k], x[k]))
    SCA_ROW(N, out[k] = fma_(d[k], h[k], s[k]))
#else
    SCA_ROW(N, out[k] = std::sqrt(x[k]))
#endif
}
// atan_far for N argument pairs (see atan_far: recip_of / div_by twice, the polynomial, the tail).  kmin: the running minimum of
// hiword(u) - 1 (unsigned) over the first NK pairs -- a quotient 0 < u < 2^-1000 shows as a key below 0x016fffff, u == 0 (y == 0)
// wraps to the top: one comparison per candidate
template <int N, int NK> SCA_DHD static inline void atan_far_n(const double (&y)[N], const double (&x)[N], double (&out)[N], uint32_t &kmin) {
    using namespace sca_gm;                                              // (the table macros name its arrays)
    double ay[N], e[N], u[N], v0[N], vv[N], w[N], du[N], vs[N], ps[N], uv[N];
#if defined(__HIP_DEVICE_COMPILE__)
    double r[N], q[N];
    SCA_ROW(N, ay[k] = sca_gm::fabs_(y[k]); r[k] = __builtin_amdgcn_rcp(x[k]))
    SCA_ROW(N, e[k] = fma_(-x[k], r[k], 1.0))
    SCA_ROW(N, r[k] = fma_(e[k], r[k], r[k]))
    SCA_ROW(N, e[k] = fma_(-x[k], r[k], 1.0))
    SCA_ROW(N, r[k] = fma_(e[k], r[k], r[k]))
    SCA_ROW(N, u[k] = ay[k] * r[k])
    SCA_ROW(N, e[k] = fma_(-x[k], u[k], ay[k]))
    SCA_ROW(N, u[k] = fma_(e[k], r[k], u[k]))
    // from here two chains per pair: the quotient's low part (du) and the polynomial in u^2
    SCA_ROW(N, v0[k] = x[k] * u[k]; vs[k] = u[k] * u[k])
    SCA_ROW(N, vv[k] = fma_(x[k], u[k], -v0[k]); w[k] = ay[k] - v0[k]; ps[k] = sca_gm::fma_k(vs[k], sca_gm::dbl(0x3fb375f08b31cbceull), sca_gm::dbl(0xbfb7458022b13c25ull)); uv[k] = u[k] * vs[k])
    SCA_ROW(N, w[k] = w[k] - vv[k]; ps[k] = sca_gm::fma_k(vs[k], ps[k], sca_gm::dbl(0x3fbc71c6e5129a3bull)))
    SCA_ROW(N, q[k] = w[k] * r[k]; ps[k] = sca_gm::fma_k(vs[k], ps[k], sca_gm::dbl(0xbfc24924923f7603ull)))
    SCA_ROW(N, e[k] = fma_(-x[k], q[k], w[k]); ps[k] = sca_gm::fma_k(vs[k], ps[k], sca_gm::dbl(0x3fc99999999997fdull)))
    SCA_ROW(N, du[k] = fma_(e[k], r[k], q[k]); ps[k] = sca_gm::fma_k(vs[k], ps[k], sca_gm::dbl(0xbfd5555555555555ull)))
#else
    // host: the two quotients by the machine's division (what the device's reciprocal-and-correction sequences round to)
    SCA_ROW(N, ay[k] = sca_gm::fabs_(y[k]); u[k] = ay[k] / x[k])
    SCA_ROW(N, v0[k] = x[k] * u[k]; vs[k] = u[k] * u[k]; uv[k] = u[k] * vs[k])
    SCA_ROW(N, vv[k] = fma_(x[k], u[k], -v0[k]); w[k] = (ay[k] - v0[k]) - vv[k]; du[k] = w[k] / x[k]; ps[k] = sca_gm::atan_poly_k(vs[k]))
#endif
    SCA_ROW(N, e[k] = fma_(uv[k], ps[k], du[k]))
    SCA_ROW(N, e[k] = u[k] + e[k])
    // u >= 1/16 in some lane: the table piece (cij) for everybody, selected per pair
    bool small[N], all_small = true;
#pragma unroll
    for (int k = 0; k < N; k++) { small[k] = u[k] < 0.0625; all_small = all_small && small[k]; }
    if (lean::any_says(!all_small)) {
        // (in groups of at most three pairs: seven coefficients per pair are 14 registers each)
        const double two52 = sca_gm::dbl(0x4330000000000000ull);
        constexpr int G = N <= 3 ? N : (N % 3 == 0 ? 3 : 2);          // a divisor of N (instantiated: N = 1, 2, 6)
        static_assert(N % G == 0, "the groups must tile the pairs (N = 4 with groups of three read two pairs that do not exist)");
#pragma unroll
        for (int g = 0; g < N; g += G) {
            double c0[G], c1[G], c2[G], c3[G], c4[G], c5[G], c6[G], t3[G], v[G], p3[G], bb[G], dv[G], ww[G]; int idx[G];
            SCA_ROW(G, t3[k] = fma_(u[g + k], 256.0, two52))
            SCA_ROW(G, idx[k] = (int)(t3[k] - two52) - 16; idx[k] = idx[k] < 0 ? 0 : (idx[k] > 240 ? 240 : idx[k]))
            SCA_ROW(G, const uint64_t *c = SCA_GM_ATAN_TAB + 7 * idx[k]; c0[k] = tab(c, 0); c1[k] = tab(c, 1); c2[k] = tab(c, 2);
                       c3[k] = tab(c, 3); c4[k] = tab(c, 4); c5[k] = tab(c, 5); c6[k] = tab(c, 6))
            SCA_ROW(G, t3[k] = u[g + k] - c0[k])
            SCA_ROW(G, v[k] = t3[k] + du[g + k])
            SCA_ROW(G, p3[k] = fma_(v[k], c6[k], c5[k]); bb[k] = v[k] - t3[k])
            SCA_ROW(G, p3[k] = fma_(v[k], p3[k], c4[k]); dv[k] = t3[k] - (v[k] - bb[k]); bb[k] = du[g + k] - bb[k])
            SCA_ROW(G, p3[k] = fma_(v[k], p3[k], c3[k]); dv[k] = dv[k] + bb[k]; ww[k] = v[k] * v[k])
            SCA_ROW(G, ww[k] = ww[k] * p3[k])
            SCA_ROW(G, ww[k] = fma_(dv[k], c2[k], ww[k]))
            SCA_ROW(G, ww[k] = fma_(v[k], c2[k], ww[k]))
            SCA_ROW(G, e[g + k] = small[g + k] ? e[g + k] : ww[k] + c1[k])
        }
    }
    // glibc's early return for exponent fields 57 or more apart (e_atan2.c: `de < -57`, x > 0): the quotient itself.  Only the
    // first NK pairs can be that lopsided (the others are 2 / p)
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int32_t de = (hiword(ay[k]) & 0x7ff00000) - (hiword(x[k]) & 0x7ff00000);
        e[k] = de < (int32_t)0xfc700001 ? u[k] : e[k];
        const uint32_t key = (uint32_t)hiword(u[k]) - 1u;
        kmin = key < kmin ? key : kmin;
    }
    SCA_ROW(N, out[k] = sca_gm::copysign_(e[k], y[k]))
}
// (the lean division has no scaling for quotients near the bottom of the exponent range: 0 < u < 2^-1000 goes the literal way)
SCA_DHD static inline bool keys_odd(uint32_t kmin) { return kmin < 0x016fffffu; }
// mod2pi for N arguments: the device form above with its exact scalings by two folded into the constants (x = t / 2 is never
// formed: q0 = t * fl(1 / 2 pi), the residual against 2 pi is twice the residual against pi, and half of fl(1 / pi) takes it back)
template <int N> SCA_DHD static inline void mod2pi_n(const double (&t)[N], double (&out)[N]) {
    double q[N], r[N];
    SCA_ROW(N, q[k] = t[k] * 0.15915494309189535)
    SCA_ROW(N, r[k] = std::fma(-q[k], 2.0 * PI, t[k]))
    SCA_ROW(N, q[k] = std::fma(r[k], 0.15915494309189535, q[k]))
    SCA_ROW(N, q[k] = std::floor(q[k]))
    SCA_ROW(N, q[k] = 2.0 * PI * q[k])
    SCA_ROW(N, out[k] = t[k] - q[k])
}
// the four CSC words of a far 2-D problem: the shortest word's cost, its first segment and whether it starts with a right turn
template <bool WINNER>
SCA_DHD static inline double words_far(const Frame2D &F, double mbeta, double d, double c, uint32_t &kmin, double &bt, bool &bright) {
    const double cab2 = 2 * F.c_ab, d2 = d * d, dd = 2 * d;
    double p2[4], y6[6], x6[6], p[4], A[6];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const bool cross = w >= 2, rfirst = (w & 1) != 0;
        const double u = rfirst ? -F.sa : F.sa;                // LSL(+,-) RSR(-,+) LSR(+,+) RSL(-,-)
        const double v = (w == 0 || w == 3) ? -F.sb : F.sb;
        const double S = u + v;
        const double k2 = cross ? -2.0 : 2.0;
        p2[w] = ((k2 + d2) + (cross ? cab2 : -cab2)) + (dd * S);
        x6[w] = (d + u) + v;
        const double ya = (w == 0 || w == 2) ? -F.ca : F.ca;
        const double yb = (w == 1 || w == 2) ? -F.cb : F.cb;
        y6[w] = ya + yb;
    }
    SCA_LEAN_ROW();
    sqrt_pos_n<4>(p2, p);
    y6[4] = -2.0; x6[4] = p[2];                                 // LSR's and RSL's second arctangent
    y6[5] = 2.0; x6[5] = p[3];
    atan_far_n<6, 4>(y6, x6, A, kmin);                          // (the quotients 2 / p of the last two cannot be tiny)
    double arg[8], m[8];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const bool rfirst = (w & 1) != 0;
        const double tmp = w >= 2 ? A[w] - A[w + 2] : A[w];     // LSL / RSR: their second arctangent is atan2(+0, p) = +0
        const double ta = tmp - F.alpha;
        arg[w] = rfirst ? -ta : ta;
        const double qa = (w == 2 ? mbeta : F.beta) - tmp;
        arg[4 + w] = (w == 1 || w == 2) ? -qa : qa;
    }
    SCA_LEAN_ROW();
    mod2pi_n<8>(arg, m);
    double cost[4];
    SCA_ROW(4, cost[k] = std::fabs(m[k]) + std::fabs(p[k]))
    SCA_ROW(4, cost[k] = cost[k] + std::fabs(m[4 + k]))
    SCA_ROW(4, cost[k] = c * cost[k])
    double bcost = INFINITY;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        bcost = sca_gm::min_(bcost, cost[w]);
    }
    if (WINNER) {                                              // the first word that attains the minimum (`if bcost > cost` in planner order)
        const bool e0 = cost[0] == bcost, e1 = cost[1] == bcost, e2 = cost[2] == bcost;
        bt = e0 ? m[0] : (e1 ? m[1] : (e2 ? m[2] : m[3]));
        bright = e0 ? false : (e1 ? true : (e2 ? false : true));
    }
    return bcost;
}
// sin and cos of N (three: the lane-per-plan search; one: a lane of the quad search) arguments below 105414350 in magnitude, in rows: sca_gm::sincos_bf<3> statement for statement (one do_sin
// and one do_cos per argument; see there), with ONE wavefront-uniform decision for all three -- whether any do_sin argument
// needs the table piece -- instead of two inside every call.  sn[2] is computed and not used by the caller (cos(alpha - beta)).
template <int N> SCA_DHD static inline void sincos_n(const double (&x)[N], double (&sn)[N], double (&cs)[N]) {
    using namespace sca_gm;
    const double hp0 = dbl(0x3ff921fb54442d18ull), hp1 = dbl(0x3c91a62633145c07ull);
    const double toint = dbl(0x4338000000000000ull), big = dbl(0x42c8000000000000ull);
    const double mp1 = dbl(0x3ff921fb58000000ull), mp2 = dbl(0xbe4dde973c000000ull);
    const double pp3 = dbl(0xbc8cb3b398000000ull), pp4 = dbl(0xbacd747f23e32ed7ull);
    int32_t hx[N]; bool r1[N], r2[N]; uint32_t n[N];
    double t[N], xn[N], y[N], t2[N], d1[N], a3[N], d2[N], da3[N], tt[N], a2[N], da2[N], aS[N], dS[N], aC[N], dC[N], S[N], C[N];
    SCA_ROW(N, hx[k] = hiword(x[k]) & 0x7fffffff; r1[k] = hx[k] < 0x3feb6000; r2[k] = hx[k] < 0x400368fd)
    // reduce_sincos (range 3) and the range-2 arguments
    SCA_ROW(N, t[k] = fma_(x[k], dbl(0x3fe45f306dc9c883ull), toint); tt[k] = hp0 - fabs_(x[k]))
    SCA_ROW(N, xn[k] = t[k] - toint; n[k] = loword(t[k]); a2[k] = tt[k] + hp1)
    SCA_ROW(N, y[k] = fma_(-xn[k], mp1, x[k]); da2[k] = (tt[k] - a2[k]) + hp1)
    SCA_ROW(N, y[k] = fma_(-xn[k], mp2, y[k]))
    SCA_ROW(N, t2[k] = fma_(-xn[k], pp3, y[k]))
    SCA_ROW(N, d1[k] = fma_(-pp3, xn[k], y[k] - t2[k]); a3[k] = fma_(-xn[k], pp4, t2[k]))
    SCA_ROW(N, d2[k] = fma_(-xn[k], pp4, t2[k] - a3[k]))
    SCA_ROW(N, da3[k] = d1[k] + d2[k])
    SCA_ROW(N, aS[k] = sel(r1[k], x[k], sel(r2[k], a2[k], a3[k])); dS[k] = sel(r1[k], 0.0, sel(r2[k], da2[k], da3[k]));
               aC[k] = sel(r1[k], x[k], sel(r2[k], tt[k], a3[k])); dC[k] = sel(r1[k], 0.0, sel(r2[k], hp1, da3[k])))
    // do_cos of (aC, dC): the table piece
    {
        double db[N], aa[N], u[N], xx[N], xr[N], p[N], s[N], q[N], c[N], cor[N]; int ki[N];
        SCA_ROW(N, db[k] = flip(dC[k], aC[k] < 0); aa[k] = fabs_(aC[k]))
        SCA_ROW(N, u[k] = aa[k] + big)
        SCA_ROW(N, xr[k] = (aa[k] - (u[k] - big)) + db[k]; ki[k] = (int)(loword(u[k]) << 2); ki[k] = ki[k] < 0 ? 0 : (ki[k] > 436 ? 436 : ki[k]))
        SCA_ROW(N, xx[k] = xr[k] * xr[k])
        SCA_ROW(N, p[k] = fma_k(xx[k], dbl(0x3f811110e829872full), dbl(0xbfc5555555555515ull)); q[k] = fma_k(xx[k], dbl(0x3f56c16bedd9e239ull), dbl(0xbfa5555555555535ull)); s[k] = xr[k] * xx[k])
        SCA_ROW(N, s[k] = fma_(s[k], p[k], xr[k]); q[k] = fma_k(xx[k], q[k], 0.5))
        SCA_ROW(N, c[k] = xx[k] * q[k])
        SCA_ROW(N, cor[k] = fma_(-s[k], tab(SCA_GM_SINCOS_TAB, ki[k] + 1), tab(SCA_GM_SINCOS_TAB, ki[k] + 3)))
        SCA_ROW(N, cor[k] = fma_(-c[k], tab(SCA_GM_SINCOS_TAB, ki[k] + 2), cor[k]))
        SCA_ROW(N, cor[k] = fma_(-s[k], tab(SCA_GM_SINCOS_TAB, ki[k]), cor[k]))
        SCA_ROW(N, C[k] = tab(SCA_GM_SINCOS_TAB, ki[k] + 2) + cor[k])
    }
    // do_sin of (aS, dS): TAYLOR_SIN, and the table piece when some lane's argument is 0.126 or more
    bool tay[N];
    {
        double xxt[N], pt[N], w[N];
        SCA_ROW(N, tay[k] = fabs_(aS[k]) < 0.126; xxt[k] = aS[k] * aS[k])
        SCA_ROW(N, pt[k] = fma_k(xxt[k], dbl(0xbe5addffc2fcdf59ull), dbl(0x3ec71de27b9a7ed9ull)))
        SCA_ROW(N, pt[k] = fma_k(xxt[k], pt[k], dbl(0xbf2a01a019db08b8ull)))
        SCA_ROW(N, pt[k] = fma_k(xxt[k], pt[k], dbl(0x3f81111111110eceull)))
        SCA_ROW(N, pt[k] = fma_k(xxt[k], pt[k], dbl(0xbfc5555555555555ull)))
        SCA_ROW(N, w[k] = fma_(pt[k], aS[k], -(dS[k] * 0.5)))
        SCA_ROW(N, w[k] = fma_(xxt[k], w[k], dS[k]))
        SCA_ROW(N, S[k] = w[k] + aS[k])
    }
    bool all_tay = true;
#pragma unroll
    for (int k = 0; k < N; k++) all_tay = all_tay && tay[k];
    if (lean::any_says(!all_tay)) {
        double db[N], aa[N], u[N], xx[N], xr[N], p[N], s[N], q[N], c[N], cor[N]; int ki[N];
        SCA_ROW(N, db[k] = flip(dS[k], aS[k] <= 0); aa[k] = fabs_(aS[k]))
        SCA_ROW(N, u[k] = aa[k] + big)
        SCA_ROW(N, xr[k] = aa[k] - (u[k] - big); ki[k] = (int)(loword(u[k]) << 2); ki[k] = ki[k] < 0 ? 0 : (ki[k] > 436 ? 436 : ki[k]))
        SCA_ROW(N, xx[k] = xr[k] * xr[k])
        SCA_ROW(N, p[k] = fma_k(xx[k], dbl(0x3f811110e829872full), dbl(0xbfc5555555555515ull)); q[k] = fma_k(xx[k], dbl(0x3f56c16bedd9e239ull), dbl(0xbfa5555555555535ull)); s[k] = xr[k] * xx[k])
        SCA_ROW(N, s[k] = fma_(s[k], p[k], db[k]); q[k] = fma_k(xx[k], q[k], 0.5))
        SCA_ROW(N, s[k] = xr[k] + s[k]; c[k] = xx[k] * q[k])
        SCA_ROW(N, c[k] = fma_(xr[k], db[k], c[k]))
        SCA_ROW(N, cor[k] = fma_(s[k], tab(SCA_GM_SINCOS_TAB, ki[k] + 3), tab(SCA_GM_SINCOS_TAB, ki[k] + 1)))
        SCA_ROW(N, cor[k] = fma_(-c[k], tab(SCA_GM_SINCOS_TAB, ki[k]), cor[k]))
        SCA_ROW(N, cor[k] = fma_(s[k], tab(SCA_GM_SINCOS_TAB, ki[k] + 2), cor[k]))
        SCA_ROW(N, S[k] = sel(tay[k], S[k], copysign_(tab(SCA_GM_SINCOS_TAB, ki[k]) + cor[k], aS[k])))
    }
    // sin: r1 do_sin; r2 copysign(do_cos, x); r3 (n & 1 ? do_cos : do_sin), negated when n & 2.  cos: r1 do_cos; r2 do_sin; r3 with n + 1.
    // |x| < 2^-26: sin(x) = x; |x| < 2^-27: cos(x) = 1
    SCA_ROW(N, const bool odd = (n[k] & 1) != 0; const uint32_t m = n[k] + 1;
               const double s3 = flip(sel(odd, C[k], S[k]), (n[k] & 2) != 0), c3 = flip(sel((m & 1) != 0, C[k], S[k]), (m & 2) != 0);
               const double sv = sel(r1[k], S[k], sel(r2[k], copysign_(C[k], x[k]), s3)), cv = sel(r1[k], C[k], sel(r2[k], S[k], c3));
               sn[k] = hx[k] < 0x3e500000 ? x[k] : sv; cs[k] = hx[k] < 0x3e400000 ? 1.0 : cv)
}
// a candidate by the literal construction: a function of its own (called, cold) so that the search loop's code stays small
#if defined(SCA_LEAN_GENERAL_INLINE)
SCA_DHD static inline
#else
SCA_DHD static __attribute__((noinline))
#endif
bool candidate_general(const Frame2D &H, const SearchConst &K, const double qi[5], const double qf[5], double Rmin,
                       const double pitchlims[2], double hr, double &len) {
    Maneuver2D mh, mv;
    const int n = try_to_construct(H, K, qi, qf, Rmin, pitchlims, hr, mh, mv);
    len = mv.length;
    return n > 0;
}
// one candidate radius: feasible?  len = the 3-D path's length.  (try_to_construct + what the search reads of its result)
#if defined(SCA_LEAN_CAND_NOINLINE)
SCA_DHD static __attribute__((noinline))
#else
SCA_DHD static inline
#endif
bool candidate(bool fast_ok, const Frame2D &H, double mbetaH, const SearchConst &K, const double qi[5], const double qf[5], double Rmin,
               const double pitchlims[2], double hr, double &len) {
#if defined(SCA_LEAN_WAVEITERS)
    { const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true); if ((int)(threadIdx.x & 63) == __builtin_ctzll(m_)) atomicAdd(&::g_wave_iters[threadIdx.x >> 6], 1); }
#endif
    // the radius Rmin itself (every search's first candidate, and every candidate the search clamps to c = 1): the vertical
    // curvature is sqrt(x - x) = 0 for a finite Rmin, i.e. try_to_construct leaves at :146-147 -- when that holds for the whole
    // wavefront nothing is evaluated
    if (fast_ok && !lean::any_says(hr != Rmin)) { len = 0.0; return false; }
    const double dH = H.D / hr;                                                       // (plan2d)
    if (fast_ok && !lean::any_says(!far_d(dH))) {
        uint32_t kmin = 0xffffffffu;
        SCA_LEAN_T(t0);
        const double vc = sca_gm::sqrt_(K.inv_rmin2 - 1.0 / sca_gm::g_pow2_main(hr));    // (hr = Rmin c: inside pow2's main range, plan3d_lean checks Rmin)
        const bool flat = vc < 1e-5;                                                  // :146-147 (a lane that leaves here computes on, unread)
        SCA_LEAN_T(t1);
        double dummy_t = 0.0; bool dummy_r = false;
        const double lenH = words_far<false>(H, mbetaH, dH, hr, kmin, dummy_t, dummy_r);
        SCA_LEAN_T(t2);
        const double vr = 1.0 / (flat ? 1.0 : vc);
        // frame2d_vertical
        const double dz = qf[2] - qi[2];
        Frame2D F;
        F.D = sqrt_pos(sca_gm::g_pow2_main(lenH) + K.dz2);                            // (36 hr <= lenH < 2^100 when nothing below objects)
        const bool far_theta = lenH > std::fabs(dz) && lenH < 1.2676506002282294e30;
        // near in the vertical plane (vr grows without bound towards Rmin): known here, before the vertical frame's trigonometry and
        // words -- the literal way at once instead of after a whole evaluation that would be thrown away (round 4)
        if (!lean::any_says(!flat && !far_d(F.D / vr))) {
        const double y1[1] = {dz}, x1[1] = {lenH};
        double th[1];
        atan_far_n<1, 1>(y1, x1, th, kmin);
        const double theta = mod2pi(th[0]);
        F.alpha = mod2pi(qi[4] - theta);
        F.beta = mod2pi(qf[4] - theta);
        SCA_LEAN_T(t3);
        const double x3[3] = {F.alpha, F.beta, F.alpha - F.beta};
        double s3[3], c3[3];
        sincos_n<3>(x3, s3, c3);
        F.sa = s3[0]; F.ca = c3[0]; F.sb = s3[1]; F.cb = c3[1]; F.c_ab = c3[2];
        const double dV = F.D / vr;
        SCA_LEAN_T(t4);
        double t = 0.0; bool right = false;
        const double lenV = words_far<true>(F, mod2pi(F.beta), dV, vr, kmin, t, right);
        const bool ok = !flat && !(right ? (qi[4] - t < pitchlims[0]) : (qi[4] + t > pitchlims[1]));
        SCA_LEAN_T(t5);
        SCA_LEAN_TACC(0, t0, t1); SCA_LEAN_TACC(1, t1, t2); SCA_LEAN_TACC(2, t2, t3); SCA_LEAN_TACC(3, t3, t4); SCA_LEAN_TACC(4, t4, t5); SCA_LEAN_TACC(5, 0ull, 1ull);
        if (!lean::any_says(!flat && (keys_odd(kmin) || !far_theta || !far_d(dV)))) { SCA_LEAN_STAT(0); SCA_LEAN_HOST_COUNT(g_host_fast); len = lenV; return ok; }
        }
        SCA_LEAN_STAT(1);
    }
    SCA_LEAN_STAT(2);
    SCA_LEAN_HOST_COUNT(g_host_literal);
    return candidate_general(H, K, qi, qf, Rmin, pitchlims, hr, len);
}
}  // namespace lean
SCA_DHD static Plan3D plan3d_lean(const double qi[5], const double qf[5], double Rmin, const double pitchlims[2]) {
    Plan3D P;
#if defined(SCA_LEAN_TIMING)
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
    const double qi2D[3] = {qi[0], qi[1], qi[3]}, qf2D[3] = {qf[0], qf[1], qf[3]};
    const Frame2D H = frame2d(qi2D, qf2D);
    const double mbetaH = mod2pi(H.beta);
    const SearchConst K = search_const(qi, qf, Rmin);
    // the lean form squares Rmin c without pow's range checks: c <= 2^201 (the doubling stage's guard), so
    const bool fast_ok = !lean::any_says(!(Rmin >= 1e-40 && Rmin <= 1e40));
    double b = 1.0, best = 0.0;
    bool fb = lean::candidate(fast_ok, H, mbetaH, K, qi, qf, Rmin, pitchlims, Rmin * b, best);
    int guard = 0;
    P.iters = 1;
    while (!fb) {
        b *= 2.0;
        fb = lean::candidate(fast_ok, H, mbetaH, K, qi, qf, Rmin, pitchlims, Rmin * b, best);
        P.iters++;
        if (++guard > 200) return P;
    }
    double step = 0.1;
    while (std::fabs(step) > 1e-10) {
        double c = b + step;
        if (c < 1.0) c = 1.0;
        P.iters++;
#if !defined(SCA_LEAN_NO_CLAMP_SKIP)
        // a candidate clamped to c = 1 is the radius Rmin itself: never feasible (see lean::candidate), so the lane takes the
        // verdict without an evaluation and goes on to its next candidate within the same trip of the wavefront's loop
        if (fast_ok && Rmin * c == Rmin) { step *= -0.1; continue; }
#endif
        double lc;
        const bool fc = lean::candidate(fast_ok, H, mbetaH, K, qi, qf, Rmin, pitchlims, Rmin * c, lc);
        if (fc && lc < best) { b = c; best = lc; step *= 2.; continue; }
        step *= -0.1;
    }
#if defined(SCA_LEAN_TIMING)
    { const unsigned long long tl = __builtin_readcyclecounter(); atomicMax(&::g_lean_stats[6], tl - t_begin); if ((threadIdx.x & 63) == 0) atomicAdd(&::g_lean_stats[7], tl - t_begin); }
#endif
    Maneuver2D fbh, fbv;
    try_to_construct(H, K, qi, qf, Rmin, pitchlims, Rmin * b, fbh, fbv);      // the winner's maneuvers (the candidate evaluated again)
    finish_plan(P, fbh, fbv, qi);
    return P;
}
#endif
#if defined(__HIP_DEVICE_COMPILE__) && (!defined(SCA_PLAN_LEAN) || SCA_PLAN_LEAN == 1)
#define SCA_PLAN3D_LANE plan3d_lean            // the device's lane-per-plan kernels
#else
#define SCA_PLAN3D_LANE plan3d
#endif

// ---- the tracker (scaPolicy.py:243-338) ------------------------------------------------------------------------------
#if defined(SCA_KT_TIMING) && defined(__HIPCC__)   // debug builds: workgroup 0 / thread 0's clock between the pieces of track_decide
__device__ int g_td_ticks[32];
__device__ long long g_td_last;
#endif
#if defined(SCA_KT_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define SCA_TD_MARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = wall_clock64(); g_td_ticks[k] = (int)(t_ - g_td_last); g_td_last = t_; } } while (0)
#else
#define SCA_TD_MARK(k) do { } while (0)
#endif
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
    const double *nd_per_agent;   // [n] agent.neighborDist where the agents differ (scaPolicy.py:299 reads the agent's own), else null
    // agent.turning_radius / agent.pitchlims where the tracked agents differ (scaPolicy.py:95,272,302 read the agent's own), else null.
    // The HOST tracker plans every agent with its own values.  On the DEVICE the re-plan kernels keep Rmin and the pitch limits in scalar
    // registers throughout the search (they sit at 256 VGPRs): tracked agents are grouped into CLASSES of equal (R, pitch_lo, pitch_hi), the
    // re-plan kernels are launched once per class with the class's values in turning_radius / pitch_lo / pitch_hi, and an agent of another
    // class leaves at once (cls[agent] != class_id).  The decision (track_decide: k = 3 R, the 2 R test) reads R_pa per agent.  With more
    // classes than launches are worth (sca_device_tracker_set_agent_params), plo_pa / phi_pa are DEVICE arrays too and cls is null: every
    // re-plan then has a wavefront of its own, which loads its agent's three values into scalar registers (own_plan_params).
    const double *R_pa, *plo_pa, *phi_pa;
    const uint8_t *cls;           // [n] the agent's class (device), null: one class
    int class_id;
};
SCA_DHD static inline double trk_R(const TrackView &T, int i) { return T.R_pa ? T.R_pa[i] : T.turning_radius; }

struct Pool;
struct Tracker {
    int n = 0;
    Pool *pool = nullptr;
    std::vector<double> goal, goal_heading, pref_speed;
    std::vector<uint8_t> zaxis;
    double turning_radius = 1.5, pitchlims[2] = {-PI / 4, PI / 4}, neighbor_dist = 10.0;
    std::vector<double> nd_per_agent;                              // empty: neighbor_dist for everybody (sca_tracker_set_neighbor_dist)
    std::vector<double> R_pa, plo_pa, phi_pa;                       // empty: turning_radius / pitchlims for everybody (sca_tracker_set_agent_params)
    std::vector<AgentTrack> st;
    TrackView view() const {
        return TrackView{goal.data(), goal_heading.data(), pref_speed.data(), zaxis.data(), turning_radius, pitchlims[0], pitchlims[1],
                         neighbor_dist, nd_per_agent.empty() ? nullptr : nd_per_agent.data(),
                         R_pa.empty() ? nullptr : R_pa.data(), plo_pa.empty() ? nullptr : plo_pa.data(), phi_pa.empty() ? nullptr : phi_pa.data(),
                         nullptr, 0};
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
#if defined(__HIP_DEVICE_COMPILE__)
    const double pl[2] = {T.pitch_lo, T.pitch_hi};                   // (device: the launch's class, see TrackView)
    adopt_plan(a, SCA_PLAN3D_LANE(qi, qf, T.turning_radius, pl));
#else
    const double pl[2] = {T.plo_pa ? T.plo_pa[i] : T.pitch_lo, T.phi_pa ? T.phi_pa[i] : T.pitch_hi};
    adopt_plan(a, SCA_PLAN3D_LANE(qi, qf, trk_R(T, i), pl));
#endif
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
    SCA_TD_MARK(0);
    const double dis_goal = l3norm(pos, goal);
    const double k = 3.0 * trk_R(T, i);
    if (!a.is_use_dubins) {
        a.is_use_dubins = true;
        return true;
    }
    SCA_TD_MARK(1);
    update_dubins(T, a, i, pos);
    SCA_TD_MARK(2);
    const double dis = l3norm(pos, a.now_goal);
    const double max_size = round5_py(6 * a.sampling_size);
    const double pApG[3] = {goal[0] - pos[0], goal[1] - pos[1], goal[2] - pos[2]};
    const double vA64[3] = {(double)vel[0], (double)vel[1], (double)vel[2]};
    const float nvA = std::sqrt((float)((double)(float)(vel[0] * vel[0]) + (double)(float)(vel[1] * vel[1]) + (double)(float)(vel[2] * vel[2])));
    double cs = fma3(vA64, pApG) / ((double)nvA * std::sqrt(fma3(pApG, pApG)));
    if (!(cs < 1.0)) cs = 1.0;                                       // min(x, 1.0); nan -> 1.0 as Python's min does here
    if (cs < -1.0) cs = -1.0;                                        // the reference would raise; clamp
    SCA_TD_MARK(3);
    const double theta = round5_py(m_acos(cs));
    SCA_TD_MARK(4);
    const double deg100 = round5_np(100.0 * (PI / 180.0));
    const double min_dist_ob = nbr0_dsq >= 0 ? round5_py(std::sqrt(nbr0_dsq)) : std::rint(T.nd_per_agent ? T.nd_per_agent[i] : T.neighbor_dist);
    const bool condition_dist = T.zaxis[i] ? (min_dist_ob >= 2.0 * trk_R(T, i)) : false;
    const bool par_ = is_parallel(vel, a.v_pref);
    SCA_TD_MARK(5);
    if (((par_ || dis_goal <= k) && dis < max_size) || (theta >= deg100) || condition_dist) {
        update_dubins(T, a, i, pos);
        SCA_TD_MARK(6);
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
