// sca_core.h -- scalar building blocks of the SCA / RVO3D / S-RVO3D / ORCA3D velocity solver.
//
// Every function here is pure arithmetic on plain values and is compiled twice:
//   * by hipcc for gfx950 as __device__ code used by the kernels in sca_kernels.hip (the product);
//   * by g++ in tests/core_harness.cpp (test-only) so the arithmetic can be checked against the
//     oracle and the golden vectors on a machine without a GPU.
// No FMA contraction is allowed (-ffp-contract=off); fma() is written where the reference's numpy
// dot product fuses (see oracle/sca_oracle.c header for the measured numpy arithmetic).
//
// Reference citations are file:line into wuuya1/SCA (mamp/...).
#pragma once
#include <math.h>
#include <stdint.h>
#include "sca_glibc_math.h"

#if defined(__HIPCC__)
#define SCA_HD __host__ __device__ __forceinline__
#else
#define SCA_HD inline
#endif

namespace sca {

// math.atan2 / math.sin / math.cos of the reference (util.py:48-49,150-152, mampenv.py:93-95) are the host libm's; the device
// library's are within an ulp of them, not equal.  Both builds of this header go through the restated glibc (sca_glibc_math.h,
// constant tables): equal bits on gfx950 and on the host, hence free-running episodes whose positions and headings ARE the reference's.
SCA_HD double m_atan2(double y, double x) { return sca_gm::g_atan2_glob(y, x); }
SCA_HD double m_pow2(double x) { return sca_gm::g_pow2_glob(x); }                    // np.float64 ** 2 = pow(x, 2.0) (mampenv.py:94)
// the same value INLINE, for call sites inside the neighbour queries: a call would cost those kernels the callee's register window
// (k_neighbors_kd4 62 -> 80 VGPRs, measured) for a branch only scenes with obstacles take.  Its argument there is l3norm(pA, pO) - r_O:
// zero or between 1e-17 and 1e4 in magnitude, always inside the branch-free form's domain (2^-360 <= |x| < 2^361); x * x stands in
// formally for the rest of the double range.
SCA_HD double m_pow2_inline(double x) {
    bool dom = true;
    const double r = sca_gm::pow2_core<sca_gm::TabGlobal>(x, dom);
    return dom ? r : x * x;
}
// ... and INLINE forms for the policy epilogue (k_action: one lane per agent, nine libm evaluations in a row): as calls they are nine
// dependent chains of table gathers one after the other (5.6 -> 9.8 us at c3); inlined, the independent ones -- atan2(vy, vx) beside the two
// pow under the square root, the two sincos, the three pow of the travelled distance -- overlap their gathers.
SCA_HD double m_atan2_i(double y, double x) { return sca_gm::g_atan2<sca_gm::TabGlobal>(y, x); }
SCA_HD double m_pow2_i(double x) { bool dom = true; const double r = sca_gm::pow2_core<sca_gm::TabGlobal>(x, dom); return dom ? r : sca_gm::g_pow2_ref(x); }
SCA_HD void m_sincos_i(double x, double &s, double &c) { sca_gm::g_sincos<sca_gm::TabGlobal>(x, s, c); }
SCA_HD void m_sincos(double x, double &s, double &c) { const sca_gm::SinCos r = sca_gm::g_sincos_glob(x); s = r.s; c = r.c; }

constexpr int K_MAX = 16;              // agent.py:32 maxNeighbors
constexpr int MAX_LEAF = 10;           // kdTree.py:53
constexpr double EPS5 = 100000.0;      // config.py:1
constexpr double RVO_EPS = 1e-5;       // config.py:4
constexpr double PI = 3.141592653589793;
constexpr double TWO_PI = 6.283185307179586;   // 2 * pi as Python evaluates it

enum Policy : int { POL_SCA = 0, POL_RVO = 1, POL_SRVO = 2, POL_ORCA = 3, POL_ORCA_LP = 4, POL_RVO_DUBINS = 5 };
enum Flags : uint32_t { FLAG_AT_GOAL = 1, FLAG_COLLISION = 2, FLAG_TIMEOUT = 4 };
enum Status : int32_t {
    ST_SQRT_DOMAIN = 2,        // reference would raise ValueError in math.sqrt (scaPolicy.py:159)
    ST_BAD_PREF_SPEED = 4,     // np.arange(0.5, ps+0.03, ps-0.5) not of length 2 (scaPolicy.py:195)
    ST_KD_STACK = 16,          // kd traversal stack overflow
    ST_NBR_OVERFLOW = 32,      // grid mode: more than K objects in range (reference list is order dependent)
    ST_TRACKER_EDGE = 64,      // never set since round 3 (the device tracker computes the reference's bits: sca_glibc_math.h)
    ST_VPREF_EDGE = 128,       // never set since round 6 (update_velocitie and cartesian2spherical compute the reference's bits)
};
constexpr int NBR_OBSTACLE_BIT = 1 << 30;

struct Params {                 // agent.py:27-36, config.py
    double neighbor_dist;       // 10.0
    double time_step;           // 0.1
    double time_horizon;        // 10.0
    double max_speed;           // 1.0
    double max_heading_change;  // pi/4
    double near_goal_threshold; // 0.5
    double cos_heading_thr;     // smallest c with acos(c) <= max_heading_change (host libm bisection)
    int max_neighbors;          // 16
    int pad;
    double dt_nominal;          // 0.1: the integrator's step (agent.py:41, mampenv.py:90-92)
    double range_sq;            // neighborDist ** 2 as the reference computes it (scaPolicy.py:112: libm's pow, on the host: sca_gm::g_pow2)
};

// The reference keeps the solver attributes on every Agent object (agent.py:24-41) and every policy reads ITS agent's.  A context holds one value
// of each (Params); a swarm whose agents differ hands over one AgentPar per agent (sca_set_agent_params) and every per-agent code path then
// works on agent_params(d, P, agent) instead of P.  64 bytes, read once per agent and pass.
struct AgentPar {
    double neighbor_dist, time_step, time_horizon, max_speed, cos_heading_thr, dt_nominal;
    int max_neighbors, pad;
    double range_sq;            // neighbor_dist ** 2 (libm's pow, computed on the host)
};

// 48-byte public record: everything another agent (or another GPU) needs to know about an agent.
struct alignas(16) PubRec {
    double px, py, pz;
    float vx, vy, vz;
    uint32_t flags;
    double radius;
};
static_assert(sizeof(PubRec) == 48, "PubRec must be 48 bytes");

struct V3 { double x, y, z; };
SCA_HD V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
SCA_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
SCA_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
SCA_HD V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
SCA_HD V3 operator/(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
// np.dot(float64[3], float64[3]) as OpenBLAS evaluates it: fma chain
SCA_HD double dot(V3 a, V3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
SCA_HD double norm(V3 a) { return sqrt(dot(a, a)); }
// np.cross: products rounded separately
SCA_HD V3 cross(V3 a, V3 b) {
    V3 c;
    double t;
    c.x = a.y * b.z; t = a.z * b.y; c.x = c.x - t;
    c.y = a.z * b.x; t = a.x * b.z; c.y = c.y - t;
    c.z = a.x * b.y; t = a.y * b.x; c.z = c.z - t;
    return c;
}
struct F3 { float x, y, z; };
SCA_HD V3 to_v3(F3 a) { return v3((double)a.x, (double)a.y, (double)a.z); }
// np.dot(float32[3], float32[3]): float products, double accumulation, float result
SCA_HD float dotf(F3 a, F3 b) {
    double s = (double)(a.x * b.x);
    s += (double)(a.y * b.y);
    s += (double)(a.z * b.z);
    return (float)s;
}
SCA_HD float normf(F3 a) { return sqrtf(dotf(a, a)); }

// ---- rounding idioms --------------------------------------------------------------------------
// Python round(x, 5) for a float: nearest multiple of 1e-5 to the EXACT binary value (ties to even),
// returned as the nearest double.  y + e is the exact product x * 1e5 (one fma), so the rounding
// direction is decided exactly; *k_out receives the integer numerator.
SCA_HD double round5_py(double x, double *k_out = nullptr) {
    double y = x * EPS5;
    double e = fma(x, EPS5, -y);
    double r = rint(y);
    double d = y - r;
    if (d == 0.5) { if (e > 0.0) r += 1.0; }
    else if (d == -0.5) { if (e < 0.0) r -= 1.0; }
    if (k_out) *k_out = r;
    return r / EPS5;
}
// numpy float64 round(x, 5): rint(x * 1e5) / 1e5
SCA_HD double round5_np(double x) { return rint(x * EPS5) / EPS5; }
// int(x * 1e5) / 1e5 ; Python ints have no signed zero
SCA_HD double trunc5(double x) {
    double t = trunc(x * EPS5);
    if (t == 0.0) t = 0.0;
    return t / EPS5;
}
SCA_HD V3 trunc5(V3 a) { return v3(trunc5(a.x), trunc5(a.y), trunc5(a.z)); }

// util.py:104 l3norm.  (x ** 2 is pow(x, 2) in the reference; x * x differs from it by 1 ulp for
// ~0.1 % of inputs, which can only matter when the sum sits within 1e-10 of a rounding boundary.)
SCA_HD double l3norm(V3 a, V3 b, double *k_out = nullptr) {
    double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    double s = dx * dx + dy * dy;
    s = s + dz * dz;
    return round5_py(sqrt(s), k_out);
}
// util.py:140 distance
SCA_HD double distance5(V3 a, V3 b) {
    double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    double s = dx * dx + dy * dy;
    s = s + dz * dz;
    return round5_py(sqrt(s) + 1e-5);
}
// util.py:100 l3normsq
SCA_HD double l3normsq(V3 a, V3 b) {
    double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    double s = dx * dx + dy * dy;
    s = s + dz * dz;
    return round5_np(s);
}
// l3norm(vA_f32, [0,0,0]) : squares and sum stay float32 (scaPolicy.py:34); distance(): orca3dPolicy.py:53
SCA_HD double l3norm_f32zero(F3 v, bool plus_eps) {
    float s = v.x * v.x + v.y * v.y;
    s = s + v.z * v.z;
    double r = sqrt((double)s);
    return round5_py(plus_eps ? r + 1e-5 : r);
}
// util.py:145 get_phi -> integer numerator P (phi = P / 1e5)
// LIBM 1: for device code inside a kernel that has the libm tables in LDS (the tracker's kernels: sca_gm::lds_tables_load) -- the
// arctangent inline on those tables instead of a call into the constant-table copy (same operations, same bits)
// (LIBM 2: inline on the constant tables -- the lane-per-agent prologue kernels, where the call's latency is the kernel's)
template <int LIBM = 0>
SCA_HD double get_phi_num(double vx, double vy) {
    double phi;
    const double at = LIBM == 1 ? sca_gm::g_atan2<sca_gm::TabDefault>(vy, vx) : LIBM == 2 ? m_atan2_i(vy, vx) : m_atan2(vy, vx);
    if (vy >= 0) phi = at;
    else phi = TWO_PI + at;
    double t = trunc(phi * EPS5);
    if (t == 0.0) t = 0.0;
    return t;
}
// util.py:109 pi_2_pi with Python float modulo
SCA_HD double py_mod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) { if ((b < 0) != (m < 0)) m += b; }
    else m = copysign(0.0, b);
    return m;
}
SCA_HD double pi_2_pi(double angle) { return py_mod(angle + PI, TWO_PI) - PI; }

// util.py:44-55 cartesian2spherical (official = orca3dPolicyOfficial.py:331-342)
template <bool INLINE_LIBM = false>
SCA_HD void cartesian2spherical(double yaw, double pitch, V3 v, bool official, double act[7]) {
    V3 zero = v3(0, 0, 0);
    double speed = official ? distance5(v, zero) : l3norm(v, zero);
    double alpha = 0.0, beta = 0.0;
    if (!(speed < 0.001)) {
        if (INLINE_LIBM) {
            alpha = m_atan2_i(v.y, v.x) - yaw;
            beta = m_atan2_i(v.z, sqrt(m_pow2_i(v.x) + m_pow2_i(v.y))) - pitch;
        } else {
            alpha = m_atan2(v.y, v.x) - yaw;
            beta = m_atan2(v.z, sqrt(m_pow2(v.x) + m_pow2(v.y))) - pitch;    // sqrt(pow(v[0], 2) + pow(v[1], 2)): libm's pow, not x * x (util.py:49)
        }
    }
    act[0] = v.x; act[1] = v.y; act[2] = v.z; act[3] = speed; act[4] = alpha; act[5] = beta; act[6] = 0.0;
}

// straight-line compute_v_pref: rvo3dPolicy.py:182-196 (l3norm) / orca3dPolicy.py:348-362 (distance).  The reference's bit for bit on
// identical inputs -- and since round 6 the inputs of a free-running episode ARE identical (update_velocitie integrates on the restated
// glibc, m_sincos above), so the round-5 "a 5-decimal rounding sits near a flip" mark (SCA_ST_VPREF_EDGE) is gone: reserved, never set.
SCA_HD V3 straight_v_pref(V3 goal, V3 pos, double pref_speed, bool use_distance) {
    V3 zero = v3(0, 0, 0);
    V3 dif = goal - pos;
    double nrm = use_distance ? distance5(dif, zero) : l3norm(dif, zero);
    nrm = trunc5(nrm);
    V3 v = v3(dif.x * pref_speed / nrm, dif.y * pref_speed / nrm, dif.z * pref_speed / nrm);
    if (l3norm(goal, pos) < 0.2) v = zero;                 // util.reached :23
    return trunc5(v);
}

// ---- posture constraint (util.py:6-20) -----------------------------------------------------------
// Returns the clamped cosine; the caller compares it with Params::cos_heading_thr, which is the exact
// image of `acos(c) <= max_heading_change` under the host libm (monotone acos).
SCA_HD double posture_cos(F3 vA, double nvA_f32, V3 cand) {
    double c = dot(to_v3(vA), cand) / (nvA_f32 * norm(cand));
    if (c > 1.0) c = 1.0;
    else if (c < -1.0) c = -1.0;
    return c;
}
SCA_HD bool posture_ok(const Params &P, F3 vA, double nvA_f32, double pos_z, V3 cand) {
    double next_z = pos_z + P.time_step * cand.z;
    double c = posture_cos(vA, nvA_f32, cand);
    // nan (zero velocity or zero candidate): acos(nan) <= x is False in the reference
    return (c >= P.cos_heading_thr) && (next_z >= 0.0);
}

// ---- RVO cone (scaPolicy.py:47-60, util.py:30-41) ---------------------------------------------------
struct Cone {
    V3 apex;       // transl_vB_vA
    V3 pAB;        // pB - pA
    double R;      // combined radius (+0.05 each)
    double g;      // d^2 - R^2 with d = max(|pAB|, R)   (>= 0)
    double absSq;  // dot(pAB, pAB)
};
SCA_HD Cone make_cone(V3 pA, F3 vA, double rA, V3 pB, F3 vB, double rB, bool other_static) {
    Cone c;
    if (other_static) c.apex = pA;
    else {
        F3 h;                                            // 0.5 * (vB + vA) evaluated in float32
        h.x = 0.5f * (vB.x + vA.x); h.y = 0.5f * (vB.y + vA.y); h.z = 0.5f * (vB.z + vA.z);
        c.apex = pA + to_v3(h);
    }
    c.pAB = pB - pA;
    double agent_rad = rA + 0.05, obj_rad = rB + 0.05;
    c.R = obj_rad + agent_rad;
    c.absSq = dot(c.pAB, c.pAB);
    double d = sqrt(c.absSq);
    // asin(R/d) <= acos(c)  <=>  not (dot > 0 and dot^2 > (d^2 - R^2) |v|^2)
    c.g = (d <= c.R) ? 0.0 : c.absSq - c.R * c.R;
    if (c.g < 0.0) c.g = 0.0;
    return c;
}
// is_intersect for v_dif = (cand + pA) - apex, given s = cand + pA (hoisted, same value)
SCA_HD bool cone_hit(const Cone &c, V3 s) {
    V3 vd = s - c.apex;
    double dt = dot(c.pAB, vd);
    double n2 = dot(vd, vd);
    return (dt * fabs(dt) > c.g * n2) || (n2 == 0.0);
}
SCA_HD bool cone_hit_vdif(V3 pAB, double g, V3 vd) {
    double dt = dot(pAB, vd);
    double n2 = dot(vd, vd);
    return (dt * fabs(dt) > g * n2) || (n2 == 0.0);
}
// time to collision inside compute_without_suitV (scaPolicy.py:158-161)
SCA_HD double cone_tc(V3 pAB, double absSq_pAB, double R, V3 vd, int *status) {
    double dv = dot(vd, pAB);
    double a = dot(vd, vd);
    double discr = dv * dv - a * (absSq_pAB - R * R);
    if (discr < 0.0) { *status |= ST_SQRT_DOMAIN; discr = 0.0; }
    double tc = (dv - sqrt(discr)) / a;
    if (tc < 0) tc = 0.0;
    return tc;
}

// ---- ORCA plane (orca3dPolicyOfficial.py:56-106 == orca3dPolicy.py:57-107) --------------------------
struct Plane { V3 p, n; };
struct OrcaOb {
    Plane pl;
    V3 relPos;
    double R, g, absSq;
    F3 vB;            // float32 velocity of the other agent (zero for obstacles)
    int vB_f32;       // 1: agent (float32 velocity), 0: obstacle (float64 zeros)
};
SCA_HD OrcaOb make_orca(const Params &P, V3 pA, F3 vA, double rA, V3 pB, F3 vB, double rB, bool is_obstacle) {
    OrcaOb o;
    double invTimeHorizon = 1.0 / P.time_horizon;
    V3 relPos = pB - pA;
    V3 relVel;
    if (!is_obstacle) {                                   // float32 - float32 stays float32
        F3 d; d.x = vA.x - vB.x; d.y = vA.y - vB.y; d.z = vA.z - vB.z;
        relVel = to_v3(d);
    } else relVel = to_v3(vA);
    double distSq = dot(relPos, relPos);
    double agent_rad = rA + 0.05, obj_rad = rB + 0.05;
    double R = agent_rad + obj_rad;
    double RSq = R * R;
    V3 u, nrm;
    if (distSq > RSq) {
        V3 w = relVel - invTimeHorizon * relPos;
        double wLengthSq = dot(w, w);
        double dotProduct = dot(w, relPos);
        if (dotProduct < 0.0 && dotProduct * dotProduct > RSq * wLengthSq) {
            double wLength = sqrt(wLengthSq);
            nrm = w / wLength;
            u = (R * invTimeHorizon - wLength) * nrm;
        } else {
            double difSq = distSq - RSq;
            double dot_product = dot(relPos, relVel);
            V3 cr = cross(relPos, relVel);
            double wwSq = dot(cr, cr) / difSq;
            double pApBLength = sqrt(distSq);
            double pAp1Length = dot_product / pApBLength;
            double p1otLength = sqrt(wwSq) * (R / pApBLength);
            double pAotlength = pAp1Length + p1otLength;
            double t = pAotlength / pApBLength;
            V3 ww = relVel - t * relPos;
            double wwLength = norm(ww);
            nrm = ww / wwLength;
            u = (R * t - wwLength) * nrm;
        }
    } else {
        double invTimeStep = 1.0 / P.time_step;
        V3 w = relVel - invTimeStep * relPos;
        double wLength = norm(w);
        nrm = w / wLength;
        u = (R * invTimeStep - wLength) * nrm;
    }
    o.pl.p = to_v3(vA) + 0.5 * u;
    o.pl.n = nrm;
    o.relPos = relPos;
    o.R = R;
    o.absSq = distSq;
    double d = sqrt(distSq);
    o.g = (d <= R) ? 0.0 : distSq - RSq;
    if (o.g < 0.0) o.g = 0.0;
    o.vB = vB;
    o.vB_f32 = is_obstacle ? 0 : 1;
    return o;
}
// is_inORCA (orca3dPolicy.py:328-333)
SCA_HD bool in_orca(const Plane &pl, V3 cand) { return dot(cand - pl.p, pl.n) >= 0.0; }
// v_dif of the ORCA fallback (orca3dPolicy.py:388)
SCA_HD V3 orca_fallback_vdif(const OrcaOb &o, F3 vA, V3 cand) {
    bool moving = o.vB_f32 ? (normf(o.vB) > (float)1e-5) : false;
    if (!moving) return cand;
    F3 h; h.x = 0.5f * (vA.x + o.vB.x); h.y = 0.5f * (vA.y + o.vB.y); h.z = 0.5f * (vA.z + o.vB.z);
    return cand - to_v3(h);
}

// ---- linear programs (orca3dPolicyOfficial.py:126-300), scalar form --------------------------------
// PL is anything indexable to a Plane: a `const Plane *` (host harness, the wave-per-agent kernels) or an accessor over a
// lane-transposed LDS array (k_lp: one lane per agent).
template <class PL>
SCA_HD bool lp1(const PL &pl, int planeNo, V3 lpnt, V3 ldir, double maxSpeed, V3 vpref, bool dir_opt, V3 &nv) {
    double dotProduct = dot(lpnt, ldir);
    double disc = dotProduct * dotProduct + maxSpeed * maxSpeed - dot(lpnt, lpnt);
    if (disc < 0.0) return false;
    double sq = sqrt(disc);
    double tLeft = -dotProduct - sq, tRight = -dotProduct + sq;
    for (int i = 0; i < planeNo; i++) {
        const Plane q = pl[i];
        double numerator = dot(q.p - lpnt, q.n);
        double denominator = dot(ldir, q.n);
        if (denominator * denominator <= RVO_EPS) {
            if (numerator > 0.0) return false;
            continue;
        }
        double t = numerator / denominator;
        if (denominator >= 0.0) { if (t > tLeft) tLeft = t; }
        else { if (t < tRight) tRight = t; }
        if (tLeft > tRight) return false;
    }
    double tt;
    if (dir_opt) tt = (dot(vpref, ldir) > 0.0) ? tRight : tLeft;
    else {
        double t = dot(ldir, vpref - lpnt);
        tt = (t < tLeft) ? tLeft : (t > tRight ? tRight : t);
    }
    nv = lpnt + tt * ldir;
    return true;
}
template <class PL>
SCA_HD bool lp2(const PL &pl, int planeNo, double maxSpeed, V3 vpref, bool dir_opt, V3 &nv) {
    const Plane P = pl[planeNo];
    double planeDist = dot(P.p, P.n);
    double planeDistSq = planeDist * planeDist, radiusSq = maxSpeed * maxSpeed;
    if (planeDistSq > radiusSq) return false;
    double planeRadiusSq = radiusSq - planeDistSq;
    V3 center = planeDist * P.n;
    if (dir_opt) {
        V3 pov = vpref - dot(vpref, P.n) * P.n;
        double lsq = dot(pov, pov);
        if (lsq <= RVO_EPS) nv = center;
        else nv = center + sqrt(planeRadiusSq / lsq) * pov;
    } else {
        double dp = dot(P.p - vpref, P.n);
        nv = vpref + dp * P.n;
        if (dot(nv, nv) > radiusSq) {
            V3 res = nv - center;
            double rl = dot(res, res);
            nv = center + sqrt(planeRadiusSq / rl) * res;
        }
    }
    for (int i = 0; i < planeNo; i++) {
        const Plane Q = pl[i];
        if (dot(Q.n, Q.p - nv) > 0.0) {
            V3 cp = cross(Q.n, P.n);
            if (dot(cp, cp) <= RVO_EPS) return false;
            V3 ldir = cp / norm(cp);
            V3 lineNormal = cross(ldir, P.n);
            double dp1 = dot(Q.p - P.p, Q.n), dp2 = dot(lineNormal, Q.n);
            V3 lpnt = P.p + (dp1 / dp2) * lineNormal;
            if (!lp1(pl, i, lpnt, ldir, maxSpeed, vpref, dir_opt, nv)) return false;
        }
    }
    return true;
}
template <class PL>
SCA_HD int lp3(const PL &pl, int np, double maxSpeed, V3 vpref, bool dir_opt, V3 &nv) {
    if (dir_opt) nv = v3(vpref.x * maxSpeed, vpref.y * maxSpeed, vpref.z * maxSpeed);
    else if (dot(vpref, vpref) > maxSpeed * maxSpeed) {
        V3 t = vpref / norm(vpref);
        nv = v3(t.x * maxSpeed, t.y * maxSpeed, t.z * maxSpeed);
    } else nv = vpref;
    for (int i = 0; i < np; i++) {
        const Plane Q = pl[i];
        if (dot(Q.n, Q.p - nv) > 0.0) {
            V3 tmp = nv;
            if (!lp2(pl, i, maxSpeed, vpref, dir_opt, nv)) { nv = tmp; return i; }
        }
    }
    return np;
}
// proj must hold K_MAX planes
SCA_HD void lp4(const Plane *pl, int np, int beginPlane, double radius, V3 &nv, Plane *proj) {
    for (int i = beginPlane; i < np; i++) {
        // :264 np.dot(normal, (point - new_velocity) > 0.0): the comparison sits INSIDE the dot product,
        // the result is tested for truthiness (non-zero).  Replicated on purpose.
        V3 m = v3((pl[i].p.x - nv.x) > 0.0 ? 1.0 : 0.0, (pl[i].p.y - nv.y) > 0.0 ? 1.0 : 0.0,
                  (pl[i].p.z - nv.z) > 0.0 ? 1.0 : 0.0);
        double s = dot(pl[i].n, m);
        if (s != 0.0) {
            int cnt = 0;
            for (int j = 0; j < i; j++) {
                Plane q;
                V3 cp = cross(pl[j].n, pl[i].n);
                if (dot(cp, cp) <= RVO_EPS) {
                    if (dot(pl[i].n, pl[j].n) > 0.0) continue;
                    q.p = 0.5 * (pl[i].p + pl[j].p);
                } else {
                    V3 lineNormal = cross(cp, pl[i].n);
                    double dp1 = dot(pl[j].p - pl[i].p, pl[j].n), dp2 = dot(lineNormal, pl[j].n);
                    q.p = pl[i].p + (dp1 / dp2) * lineNormal;
                }
                V3 dn = pl[j].n - pl[i].n;
                q.n = dn / norm(dn);
                proj[cnt++] = q;
            }
            V3 tmp = nv;
            if (lp3(proj, cnt, radius, pl[i].n, true, nv) < cnt) nv = tmp;
        }
    }
}

// ---- candidate helpers ------------------------------------------------------------------------------
// np.arange(0.5, ps + 0.03, ps - 0.5) must have exactly two elements [0.5, 0.5 + (ps - 0.5)]
SCA_HD bool candidate_speeds(double pref_speed, double &rad1) {
    double step = pref_speed - 0.5;
    if (!(step > 0.0)) return false;
    double len = ceil(((pref_speed + 0.03) - 0.5) / step);
    rad1 = 0.5 + step;
    return len == 2.0;
}

// sort key of the main path packed into 32 bits: (round5 numerator of |v - v_pref|) << 10 | generation index
SCA_HD uint32_t pack_key(double k_num, int idx) {
    double k = k_num > 4194303.0 ? 4194303.0 : k_num;
    return ((uint32_t)k << 10) | (uint32_t)idx;
}

}  // namespace sca
