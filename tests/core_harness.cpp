// core_harness.cpp -- TEST-ONLY host build of sca_amd/csrc/sca_core.h.
//
// Runs the per-agent solve serially on the CPU with exactly the arithmetic building blocks the HIP kernel
// uses (cone test in algebraic form, exact round5, posture threshold, ORCA planes, LP1-4), so that this
// arithmetic can be compared with the oracle and the golden vectors on a machine without a GPU.  It is not
// part of the product and is never loaded by sca_amd.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "sca_core.h"

using namespace sca;

namespace {
struct Tab { const double *unit; const double *phi; int num_N; double rad1; int vp_idx; };
V3 cand_from_idx(const Tab &T, int idx, V3 vpref) {
    if (idx >= T.vp_idx) return vpref;
    const int n0 = (idx >= T.num_N) ? idx - T.num_N : idx;
    const double rad = (idx >= T.num_N) ? T.rad1 : 0.5;
    return v3(rad * T.unit[n0], rad * T.unit[T.num_N + n0], rad * T.unit[2 * T.num_N + n0]);
}
double phi_from_idx(const Tab &T, int idx, V3 vpref) {
    if (idx >= T.vp_idx) return get_phi_num(vpref.x, vpref.y);
    const int n0 = (idx >= T.num_N) ? idx - T.num_N : idx;
    return T.phi[n0];
}
struct Key { double a, b; int idx; };
bool less(const Key &x, const Key &y) {
    if (x.a < y.a) return true;
    if (x.a > y.a) return false;
    if (x.b < y.b) return true;
    if (x.b > y.b) return false;
    return x.idx < y.idx;
}
Key invalid() { return Key{std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity(), INT32_MAX}; }

int select_list(bool shunted, double thr, int total, const bool *inl, const double *key, const Tab &T, V3 vpref, V3 vA64) {
    int count = 0;
    Key best = invalid();
    for (int i = 0; i < total; i++) if (inl[i]) { count++; Key k{key[i], 0.0, i}; if (less(k, best)) best = k; }
    if (!shunted || count <= 1) return best.idx;
    const double s0 = l3norm(cand_from_idx(T, best.idx, vpref), vA64);
    Key fail = invalid();
    for (int i = 0; i < total; i++) if (inl[i]) {
        const double s = l3norm(cand_from_idx(T, i, vpref), vA64);
        if (!(std::fabs(s0 - s) < thr)) { Key k{key[i], 0.0, i}; if (less(k, fail)) fail = k; }
    }
    Key kmin = invalid(), kmax = invalid();
    for (int i = 0; i < total; i++) if (inl[i]) {
        Key me{key[i], 0.0, i};
        if (less(me, fail)) {
            const double ph = phi_from_idx(T, i, vpref);
            Key a{ph, key[i], i}, b{-ph, key[i], i};
            if (less(a, kmin)) kmin = a;
            if (less(b, kmax)) kmax = b;
        }
    }
    const double phi_min = kmin.a / EPS5, phi_max = (-kmax.a) / EPS5;
    return (std::fabs(phi_max - phi_min) <= PI) ? kmin.idx : kmax.idx;
}
}  // namespace

extern "C" {

double core_round5_py(double x) { return round5_py(x); }
double core_trunc5(double x) { return trunc5(x); }
double core_l3norm(const double *a, const double *b) { return l3norm(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2])); }
double core_l3normsq(const double *a, const double *b) { return l3normsq(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2])); }
double core_distance(const double *a, const double *b) { return distance5(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2])); }
double core_get_phi(const double *v) { return get_phi_num(v[0], v[1]) / EPS5; }
double core_pi_2_pi(double a) { return pi_2_pi(a); }
int core_posture_ok(double cos_thr, double time_step, const float *vA, double pos_z, const double *c) {
    Params P{}; P.cos_heading_thr = cos_thr; P.time_step = time_step;
    F3 v{vA[0], vA[1], vA[2]};
    return posture_ok(P, v, (double)normf(v), pos_z, v3(c[0], c[1], c[2])) ? 1 : 0;
}
int core_is_intersect(const double *pA, const double *pB, double R, const double *vd) {
    V3 pAB = v3(pB[0] - pA[0], pB[1] - pA[1], pB[2] - pA[2]);
    double absSq = dot(pAB, pAB), d = std::sqrt(absSq);
    double g = (d <= R) ? 0.0 : absSq - R * R;
    if (g < 0) g = 0;
    return cone_hit_vdif(pAB, g, v3(vd[0], vd[1], vd[2])) ? 1 : 0;
}
void core_c2s(const double *heading, const double *v, int official, double *act) {
    cartesian2spherical(heading[0], heading[1], v3(v[0], v[1], v[2]), official != 0, act);
}

// One agent, same structure as k_solve.  par = [neighbor_dist, time_step, time_horizon, max_speed,
// max_heading_change, near_goal, cos_thr].  Returns status bits.
int core_solve_agent(const double *par, int pol, int zaxis, double pref_speed, const double *pos, const float *vel,
                     double radius, const double *heading, const double *goal, int vpref_given, const double *vpref_in,
                     int K, const double *nb_pos, const float *nb_vel, const double *nb_rad, const uint8_t *nb_obst,
                     const uint8_t *nb_goal, const double *unit256, const double *unit128, const double *phi256,
                     const double *phi128, float *action, double *vpref_out, int32_t *diag) {
    Params P{};
    P.neighbor_dist = par[0]; P.time_step = par[1]; P.time_horizon = par[2]; P.max_speed = par[3];
    P.max_heading_change = par[4]; P.near_goal_threshold = par[5]; P.cos_heading_thr = par[6]; P.max_neighbors = 16;
    int st = 0;
    const bool orca = (pol == POL_ORCA || pol == POL_ORCA_LP);
    const V3 pA = v3(pos[0], pos[1], pos[2]);
    F3 vA{vel[0], vel[1], vel[2]};
    const V3 vA64 = to_v3(vA);
    V3 vpref = vpref_given ? v3(vpref_in[0], vpref_in[1], vpref_in[2])
                           : straight_v_pref(v3(goal[0], goal[1], goal[2]), pA, pref_speed, orca);
    vpref_out[0] = vpref.x; vpref_out[1] = vpref.y; vpref_out[2] = vpref.z;
    for (int k = 0; k < 5; k++) diag[k] = -1;
    const bool first_step = l3norm_f32zero(vA, orca) <= 1e-5;
    V3 vpost;
    if (first_step) vpost = v3(0.3 * vpref.x, 0.3 * vpref.y, 0.3 * vpref.z);
    else {
        Cone cones[K_MAX]; OrcaOb obs[K_MAX]; Plane planes[K_MAX], proj[K_MAX];
        for (int j = 0; j < K; j++) {
            V3 pB = v3(nb_pos[3 * j], nb_pos[3 * j + 1], nb_pos[3 * j + 2]);
            F3 vB{nb_vel[3 * j], nb_vel[3 * j + 1], nb_vel[3 * j + 2]};
            if (nb_obst[j]) vB = F3{0, 0, 0};
            if (!orca) cones[j] = make_cone(pA, vA, radius, pB, vB, nb_rad[j], nb_obst[j] || nb_goal[j]);
            else { obs[j] = make_orca(P, pA, vA, radius, pB, vB, nb_rad[j], nb_obst[j] != 0); planes[j] = obs[j].pl; }
        }
        if (pol == POL_ORCA_LP) {
            V3 nv = v3(0, 0, 0);
            int pf = lp3(planes, K, P.max_speed, vpref, false, nv);
            diag[3] = pf; diag[4] = 0;
            if (pf < K) { lp4(planes, K, pf, P.max_speed, nv, proj); diag[4] = 1; }
            vpost = nv;
        } else {
            Tab T;
            T.num_N = (pol == POL_SCA && zaxis) ? 128 : 256;
            T.unit = T.num_N == 256 ? unit256 : unit128;
            T.phi = T.num_N == 256 ? phi256 : phi128;
            T.vp_idx = 2 * T.num_N;
            if (!candidate_speeds(pref_speed, T.rad1)) { st |= ST_BAD_PREF_SPEED; T.rad1 = pref_speed; }
            const int total = T.vp_idx + 1;
            const double nvA = (double)normf(vA);
            static thread_local bool okp[520], suit[520], inl[520];
            static thread_local double key[520];
            int n_suit = 0;
            for (int i = 0; i < total; i++) {
                const V3 c = cand_from_idx(T, i, vpref);
                okp[i] = posture_ok(P, vA, nvA, pA.z, c);
                bool ok = okp[i];
                const V3 sh = c + pA;
                for (int j = 0; j < K && ok; j++) {
                    if (!orca) { if (cone_hit(cones[j], sh)) ok = false; }
                    else { if (!in_orca(planes[j], c)) ok = false; }
                }
                suit[i] = ok; n_suit += ok;
            }
            diag[0] = n_suit;
            const bool shunted = (pol == POL_SCA || pol == POL_SRVO);
            int chosen;
            if (n_suit > 0) {
                diag[1] = 0;
                for (int i = 0; i < total; i++) { inl[i] = suit[i]; key[i] = suit[i] ? l3norm(cand_from_idx(T, i, vpref), vpref) : 0.0; }
                chosen = select_list(shunted, pol == POL_SCA ? 3e-2 : 1e-1, total, inl, key, T, vpref, vA64);
            } else {
                diag[1] = 1;
                for (int i = 0; i < total; i++) {
                    const V3 c = cand_from_idx(T, i, vpref);
                    double tcm = 0.0; bool have = false;
                    if (okp[i])
                        for (int j = 0; j < K; j++) {
                            V3 pAB, vd; double g, R, absSq;
                            if (!orca) { pAB = cones[j].pAB; g = cones[j].g; R = cones[j].R; absSq = cones[j].absSq; vd = (c + pA) - cones[j].apex; }
                            else { pAB = obs[j].relPos; g = obs[j].g; R = obs[j].R; absSq = obs[j].absSq; vd = orca_fallback_vdif(obs[j], vA, c); }
                            if (cone_hit_vdif(pAB, g, vd)) {
                                const double tc = cone_tc(pAB, absSq, R, vd, &st);
                                if (!have || tc < tcm) { tcm = tc; have = true; }
                            }
                        }
                    inl[i] = true;
                    key[i] = (0.2 / (tcm + 1e-5)) + l3norm(c, vpref);
                }
                chosen = select_list(shunted, pol == POL_SCA ? 5e-2 : 1e-1, total, inl, key, T, vpref, vA64);
            }
            diag[2] = chosen;
            vpost = trunc5(cand_from_idx(T, chosen, vpref));
        }
    }
    double act[7];
    cartesian2spherical(heading[0], heading[1], vpost, pol == POL_ORCA_LP, act);
    for (int k = 0; k < 7; k++) action[k] = (float)act[k];
    return st;
}

}  // extern "C"
