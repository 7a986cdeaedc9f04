/*
 * sca_oracle.c -- CPU restatement of the wuuya1/SCA per-agent hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker* for the HIP path in sca_amd/csrc.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  It is never a fallback for the product.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against golden
 * vectors recorded from the reference itself (tools/gen_golden.py imports /root/reference in the
 * build container and commits inputs/outputs under tests/golden/).
 *
 * It follows the reference's Python literally, statement by statement, including the dtype flow
 * (velocities are float32 after the first step, mampenv.py:31,104), the three rounding idioms
 * (Python round(), numpy round(), int(x*1e5)/1e5) and the arithmetic that numpy performs for 3-vectors
 * on the build host:
 *    np.dot(float64[3], float64[3])  == fma(a2,b2, fma(a1,b1, a0*b0))        (OpenBLAS ddot tail loop)
 *    np.dot(float32[3], float32[3])  == (float)((double)(float)(a0*b0) + (double)(float)(a1*b1) + ...)
 *    scalar x ** 2                   == pow(x, 2.0)   (differs from x*x in ~0.1 % of inputs, glibc 2.35)
 * (measured in the build container; see DESIGN.md "Oracle").
 *
 * Each function cites the reference file:line it restates (paths relative to /root/reference).
 * Build:  make -C oracle      ->  oracle/liboracle.so
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define K_MAX 16
#define MAX_LEAF 10          /* kdTree.py:53 */
#define EPS5 100000.0        /* config.py:1 eps = 10 ** 5 */
#define RVO_EPS 1e-5         /* config.py:4 rvo3d_epsilon */

enum { POL_SCA = 0, POL_RVO = 1, POL_SRVO = 2, POL_ORCA = 3, POL_ORCA_LP = 4, POL_RVO_DUBINS = 5 };
enum { FLAG_AT_GOAL = 1, FLAG_COLLISION = 2, FLAG_TIMEOUT = 4 };
enum { ST_ACOS_DOMAIN = 1, ST_SQRT_DOMAIN = 2, ST_BAD_PREF_SPEED = 4, ST_DIV_ZERO = 8 };

/* solver parameters: agent.py:27-36 */
typedef struct {
    double neighbor_dist;       /* agent.py:33  10.0 */
    int max_neighbors;          /* agent.py:32  16   */
    double time_step;           /* agent.py:34  DT = 0.1 */
    double time_horizon;        /* agent.py:35  10.0 */
    double max_speed;           /* agent.py:36  1.0 */
    double max_heading_change;  /* agent.py:29  pi/4 */
    double near_goal_threshold; /* config.py:3  0.5 */
    double dt_nominal;          /* agent.py:41  DT = 0.1: the integrator's step (mampenv.py:90-92), not timeStep */
} OrcParams;

/* g_ctx: what orc_set_params set (one value per scene).  g_par: what the functions below read -- the agent's own attributes while
 * orc_policy_step / orc_env_update work on that agent (the reference reads them off the Agent object per call), thread-local because
 * orc_policy_step runs agents in parallel. */
static OrcParams g_ctx = {10.0, 16, 0.1, 10.0, 1.0, 0.78539816339744830962, 0.5, 0.1};
static _Thread_local OrcParams g_par = {10.0, 16, 0.1, 10.0, 1.0, 0.78539816339744830962, 0.5, 0.1};
/* per-agent attributes (agent.py:24-41 are per-object): NULL = the scene's value.  Set by orc_set_agent_params, n entries each. */
static int g_pa_n = 0;
static double *g_pa_neighbor_dist, *g_pa_time_step, *g_pa_time_horizon, *g_pa_max_speed, *g_pa_max_heading_change, *g_pa_dt_nominal;
static int32_t *g_pa_max_neighbors;
static void par_for_agent(int i) {
    g_par = g_ctx;
    if (i < 0 || i >= g_pa_n) return;
    if (g_pa_neighbor_dist) g_par.neighbor_dist = g_pa_neighbor_dist[i];
    if (g_pa_max_neighbors) g_par.max_neighbors = g_pa_max_neighbors[i];
    if (g_pa_time_step) g_par.time_step = g_pa_time_step[i];
    if (g_pa_time_horizon) g_par.time_horizon = g_pa_time_horizon[i];
    if (g_pa_max_speed) g_par.max_speed = g_pa_max_speed[i];
    if (g_pa_max_heading_change) g_par.max_heading_change = g_pa_max_heading_change[i];
    if (g_pa_dt_nominal) g_par.dt_nominal = g_pa_dt_nominal[i];
}
static double *dup_d(const double *a, int n) { if (!a) return NULL; double *r = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1)); memcpy(r, a, sizeof(double) * (size_t)n); return r; }
/* n = 0 (or all arrays NULL): back to one value per scene */
void orc_set_agent_params(int n, const double *neighbor_dist, const int32_t *max_neighbors, const double *time_step, const double *time_horizon,
                          const double *max_speed, const double *max_heading_change, const double *dt_nominal) {
    free(g_pa_neighbor_dist); free(g_pa_time_step); free(g_pa_time_horizon); free(g_pa_max_speed); free(g_pa_max_heading_change); free(g_pa_dt_nominal);
    free(g_pa_max_neighbors);
    g_pa_n = n;
    g_pa_neighbor_dist = dup_d(neighbor_dist, n); g_pa_time_step = dup_d(time_step, n); g_pa_time_horizon = dup_d(time_horizon, n);
    g_pa_max_speed = dup_d(max_speed, n); g_pa_max_heading_change = dup_d(max_heading_change, n); g_pa_dt_nominal = dup_d(dt_nominal, n);
    g_pa_max_neighbors = NULL;
    if (max_neighbors && n > 0) { g_pa_max_neighbors = (int32_t *)malloc(sizeof(int32_t) * (size_t)n); memcpy(g_pa_max_neighbors, max_neighbors, sizeof(int32_t) * (size_t)n); }
}

void orc_set_params(double neighbor_dist, int max_neighbors, double time_step, double time_horizon,
                    double max_speed, double max_heading_change, double near_goal_threshold) {
    g_ctx.neighbor_dist = neighbor_dist; g_ctx.max_neighbors = max_neighbors; g_ctx.time_step = time_step;
    g_ctx.time_horizon = time_horizon; g_ctx.max_speed = max_speed; g_ctx.max_heading_change = max_heading_change;
    g_ctx.near_goal_threshold = near_goal_threshold;
    g_par = g_ctx;                                            /* (the scalar helpers called directly by the KAT tests read g_par on this thread) */
}
void orc_set_dt_nominal(double dt_nominal) { g_ctx.dt_nominal = dt_nominal; g_par = g_ctx; }

/* ------------------------------------------------------------------ numpy / Python arithmetic idioms */

static inline double dot3(const double *a, const double *b) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
static inline float dot3f(const float *a, const float *b) {
    double s = (double)(float)(a[0] * b[0]);
    s += (double)(float)(a[1] * b[1]);
    s += (double)(float)(a[2] * b[2]);
    return (float)s;
}
static inline double norm3(const double *a) { return sqrt(dot3(a, a)); }       /* np.linalg.norm, float64 */
static inline float norm3f(const float *a) { return sqrtf(dot3f(a, a)); }      /* np.linalg.norm, float32 */
static inline double sqr(double x) { return pow(x, 2.0); }                     /* util.py:84 sqr / x ** 2 */
static inline float sqrf(float x) { return powf(x, 2.0f); }
static inline void cross3(const double *a, const double *b, double *c) {       /* np.cross: products rounded separately */
    double t;
    c[0] = a[1] * b[2]; t = a[2] * b[1]; c[0] -= t;
    c[1] = a[2] * b[0]; t = a[0] * b[2]; c[1] -= t;
    c[2] = a[0] * b[1]; t = a[1] * b[0]; c[2] -= t;
}

/* Python float round(x, 5): correctly rounded decimal rounding of the exact binary value. */
double orc_round5_py(double x) {
    if (!isfinite(x)) return x;
    double y = x * EPS5;
    double f = y - floor(y);
    if (fabs(f - 0.5) > 1e-6 && fabs(y) < 1e15) return rint(y) / EPS5;
    char buf[64];
    snprintf(buf, sizeof buf, "%.5f", x);
    return strtod(buf, NULL);
}
/* numpy float64 round(x, 5): rint(x * 1e5) / 1e5 */
double orc_round5_np(double x) { return rint(x * EPS5) / EPS5; }
/* int(x * eps) / eps  (scaPolicy.py:239) */
/* Python int() has no signed zero: int(-1e-12) == 0 -> 0 / 1e5 == +0.0 */
static inline double py_int(double y) { double t = trunc(y); return t == 0.0 ? 0.0 : t; }
double orc_trunc5(double x) { return py_int(x * EPS5) / EPS5; }

/* util.py:104 l3norm -- math.sqrt then Python round */
double orc_l3norm(const double *p1, const double *p2) {
    return orc_round5_py(sqrt(sqr(p1[0] - p2[0]) + sqr(p1[1] - p2[1]) + sqr(p1[2] - p2[2])));
}
/* l3norm(vf32, [0,0,0]) : the squares and the sum are float32 (scaPolicy.py:34) */
static double l3norm_f32_zero(const float *v) {
    float s = sqrf(v[0]) + sqrf(v[1]);
    s = s + sqrf(v[2]);
    return orc_round5_py(sqrt((double)s));
}
double orc_l3norm_f32zero(const float *v) { return l3norm_f32_zero(v); }
/* l3norm(v64, vf32) (scaPolicy.py:128): difference promotes to float64 */
static double l3norm_mixed(const double *p1, const float *p2) {
    double q[3] = {(double)p2[0], (double)p2[1], (double)p2[2]};
    return orc_l3norm(p1, q);
}
double orc_l3norm_mixed(const double *p1, const float *p2) { return l3norm_mixed(p1, p2); }
/* util.py:100 l3normsq -- numpy round */
double orc_l3normsq(const double *x, const double *y) {
    return orc_round5_np(sqr(x[0] - y[0]) + sqr(x[1] - y[1]) + sqr(x[2] - y[2]));
}
/* util.py:140 distance */
double orc_distance(const double *p1, const double *p2) {
    return orc_round5_py(sqrt(sqr(p1[0] - p2[0]) + sqr(p1[1] - p2[1]) + sqr(p1[2] - p2[2])) + 1e-5);
}
static double distance_f32_zero(const float *v) {
    float s = sqrf(v[0]) + sqrf(v[1]);
    s = s + sqrf(v[2]);
    return orc_round5_py(sqrt((double)s) + 1e-5);
}
/* util.py:145 get_phi */
double orc_get_phi(const double *vec) {
    double phi;
    if (vec[1] >= 0) phi = atan2(vec[1], vec[0]);
    else phi = 2 * M_PI + atan2(vec[1], vec[0]);
    return py_int(phi * EPS5) / EPS5;
}
static double py_mod(double a, double b) {           /* Python / numpy float % */
    double m = fmod(a, b);
    if (m != 0.0) { if ((b < 0) != (m < 0)) m += b; }
    else m = copysign(0.0, b);
    return m;
}
/* util.py:109 pi_2_pi */
double orc_pi_2_pi(double angle) { return py_mod(angle + M_PI, 2 * M_PI) - M_PI; }
/* util.py:113 mod2pi */
double orc_mod2pi(double theta) { return theta - 2.0 * M_PI * floor(theta / 2.0 / M_PI); }

/* util.py:30-41 is_intersect (also orca3dPolicy.py:315-325 which receives pAB directly).
 * The reference raises ValueError when rounding pushes the acos argument outside [-1,1]; here the
 * argument is clamped and ST_ACOS_DOMAIN is reported. */
static int is_intersect_pAB(const double *pAB, double R, const double *v_dif, int *status) {
    double dist = norm3(pAB);
    if (dist <= R) dist = R;
    double bound = asin(R / dist);
    double c = dot3(pAB, v_dif) / (dist * norm3(v_dif));
    if (c > 1.0 || c < -1.0) { if (status) *status |= ST_ACOS_DOMAIN; c = c > 0 ? 1.0 : -1.0; }
    double th = acos(c);
    if (bound <= th) return 0;
    return 1;                                         /* also the nan case (v_dif == 0) */
}
int orc_is_intersect(const double *pA, const double *pB, double R, const double *v_dif) {
    double pAB[3] = {pB[0] - pA[0], pB[1] - pA[1], pB[2] - pA[2]};
    int st = 0;
    int r = is_intersect_pAB(pAB, R, v_dif, &st);
    return st ? 2 : r;
}
/* util.py:6-20 satisfied_constraint; vA is float32, its norm is taken in float32 */
static int satisfied_constraint(const float *vA, double pos_z, const double *vCand) {
    double next_z = pos_z + g_par.time_step * vCand[2];
    double vA64[3] = {(double)vA[0], (double)vA[1], (double)vA[2]};
    double c = dot3(vA64, vCand) / ((double)norm3f(vA) * norm3(vCand));
    if (c > 1.0) c = 1.0;
    else if (c < -1.0) c = -1.0;
    double th = acos(c);
    if (th <= g_par.max_heading_change && next_z >= 0.0) return 1;
    return 0;
}
int orc_satisfied_constraint(const float *vA, double pos_z, const double *vCand) {
    return satisfied_constraint(vA, pos_z, vCand);
}
/* util.py:44-55 cartesian2spherical; official=1: orca3dPolicyOfficial.py:331-342 (speed via distance()) */
void orc_cartesian2spherical(const double *heading, const double *v, int official, double *action) {
    double zero[3] = {0, 0, 0};
    double speed = official ? orc_distance(v, zero) : orc_l3norm(v, zero);
    double alpha, beta;
    if (speed < 0.001) { alpha = 0.0; beta = 0.0; }
    else {
        alpha = atan2(v[1], v[0]) - heading[0];
        beta = atan2(v[2], sqrt(pow(v[0], 2) + pow(v[1], 2))) - heading[1];
    }
    action[0] = v[0]; action[1] = v[1]; action[2] = v[2]; action[3] = speed; action[4] = alpha; action[5] = beta;
    action[6] = 0.0;
}

/* ------------------------------------------------------------------ candidate table (scaPolicy.py:195-200) */
static void unit_candidate(int n, int num_N, double *u) {
    double param_phi = (sqrt(5.0) - 1.0) / 2.0;
    double z_n = (double)(2 * n - 1) / num_N - 1;
    double x_n = sqrt(1 - pow(z_n, 2.0)) * cos(2 * M_PI * n * param_phi);
    double y_n = sqrt(1 - pow(z_n, 2.0)) * sin(2 * M_PI * n * param_phi);
    u[0] = x_n; u[1] = y_n; u[2] = z_n;
}
/* fills cand[][3]; returns count without v_pref; np.arange(0.5, ps + 0.03, ps - 0.5) */
static int build_candidates(double pref_speed, int num_N, double (*cand)[3], int cap, int *status) {
    double start = 0.5, stop = pref_speed + 0.03, step = pref_speed - 0.5;
    if (step == 0.0) { *status |= ST_BAD_PREF_SPEED; return 0; }      /* reference: ZeroDivisionError */
    double len_d = ceil((stop - start) / step);
    int len = len_d > 0 ? (int)len_d : 0;
    double delta = (start + step) - start;                              /* numpy arange fill rule */
    int c = 0;
    for (int i = 0; i < len; i++) {
        double rad = (i == 0) ? start : (i == 1 ? start + step : start + i * delta);
        for (int n = 1; n <= num_N; n++) {
            if (c >= cap) { *status |= ST_BAD_PREF_SPEED; return c; }
            double u[3];
            unit_candidate(n, num_N, u);
            cand[c][0] = rad * u[0]; cand[c][1] = rad * u[1]; cand[c][2] = rad * u[2];
            c++;
        }
    }
    return c;
}
/* exported for the F0 fixture check */
int orc_candidate_table(double pref_speed, int num_N, double *out, int cap) {
    int st = 0;
    return build_candidates(pref_speed, num_N, (double(*)[3])out, cap, &st);
}

/* ------------------------------------------------------------------ kd-tree (kdTree.py) */
typedef struct { int begin, end, left, right; double mn[3], mx[3]; } Node;

/* kdTree.py:60-122 buildAgentTreeRecursive == :162-227 buildObstacleTreeRecursive */
static void kd_build(Node *tree, int *ids, const double *pos, int begin, int end, int node) {
    Node *nd = &tree[node];
    nd->begin = begin; nd->end = end;
    for (int k = 0; k < 3; k++) nd->mn[k] = nd->mx[k] = pos[3 * ids[begin] + k];
    for (int i = begin + 1; i < end; i++)
        for (int k = 0; k < 3; k++) {
            double v = pos[3 * ids[i] + k];
            if (v > nd->mx[k]) nd->mx[k] = v;            /* max(a, b): b if b > a */
            if (v < nd->mn[k]) nd->mn[k] = v;
        }
    if (end - begin > MAX_LEAF) {
        double d0 = nd->mx[0] - nd->mn[0], d1 = nd->mx[1] - nd->mn[1], d2 = nd->mx[2] - nd->mn[2];
        int coord = (d0 > d1 && d0 > d2) ? 0 : (d1 > d2 ? 1 : 2);
        double split = 0.5 * (nd->mx[coord] + nd->mn[coord]);
        int left = begin, right = end;
        while (left < right) {
            while (left < right && pos[3 * ids[left] + coord] < split) left++;
            while (right > left && pos[3 * ids[right - 1] + coord] >= split) right--;
            if (left < right) {
                int t = ids[left]; ids[left] = ids[right - 1]; ids[right - 1] = t;
                left++; right--;
            }
        }
        int leftSize = left - begin;
        if (leftSize == 0) { leftSize++; left++; right++; }
        nd->left = node + 1;
        nd->right = node + 2 * leftSize;
        kd_build(tree, ids, pos, begin, left, nd->left);
        kd_build(tree, ids, pos, left, end, nd->right);
    }
}

typedef struct {
    int n; int id[K_MAX]; uint8_t kind[K_MAX]; double dsq[K_MAX];
} NbrList;

/* the list.sort(key=takeSecond) after append: stable, so the new element goes after equal keys */
static void nbr_insert_sorted(NbrList *L, int id, int kind, double dsq) {
    if (L->n == g_par.max_neighbors) L->n--;                /* neighbors.pop() : drop the last */
    int j = L->n;
    while (j > 0 && L->dsq[j - 1] > dsq) { L->id[j] = L->id[j - 1]; L->kind[j] = L->kind[j - 1]; L->dsq[j] = L->dsq[j - 1]; j--; }
    L->id[j] = id; L->kind[j] = (uint8_t)kind; L->dsq[j] = dsq;
    L->n++;
}

typedef struct {
    int self; const double *pos; const double *radius; const double *opos; const double *oradius;
    int is_collision; NbrList *L;
} Query;

/* agent.py:79-99 insertAgentNeighbor.  rangeSq is passed by value in Python, so the callee's
 * `rangeSq = self.neighbors[-1][1]` never reaches the tree query: the range stays constant. */
static void insert_agent_neighbor(Query *q, int other, double rangeSq) {
    if (q->self == other) return;
    double distSq = orc_l3normsq(&q->pos[3 * q->self], &q->pos[3 * other]);
    if (distSq < sqr(q->radius[q->self] + q->radius[other]) && distSq < rangeSq) {
        if (!q->is_collision) { q->is_collision = 1; q->L->n = 0; }
        nbr_insert_sorted(q->L, other, 0, distSq);
    } else if (!q->is_collision && distSq < rangeSq) {
        nbr_insert_sorted(q->L, other, 0, distSq);
    }
}
/* agent.py:101-124 insertObstacleNeighbor */
static void insert_obstacle_neighbor(Query *q, int ob, double rangeSq) {
    double distSq1 = orc_l3normsq(&q->pos[3 * q->self], &q->opos[3 * ob]);
    double distSq = pow(orc_l3norm(&q->pos[3 * q->self], &q->opos[3 * ob]) - q->oradius[ob], 2.0);
    if (distSq1 < sqr(q->radius[q->self] + q->oradius[ob]) && distSq < rangeSq) {
        if (!q->is_collision) { q->is_collision = 1; q->L->n = 0; }
        nbr_insert_sorted(q->L, ob, 1, distSq);
    } else if (!q->is_collision && distSq < rangeSq) {
        nbr_insert_sorted(q->L, ob, 1, distSq);
    }
}
static double box_dist_sq(const Node *c, const double *p) {
    /* kdTree.py:132-145 ; sqr(max(0.0, a - b)) summed left to right */
    double s = sqr(fmax(0.0, c->mn[0] - p[0]));
    s = s + sqr(fmax(0.0, p[0] - c->mx[0]));
    s = s + sqr(fmax(0.0, c->mn[1] - p[1]));
    s = s + sqr(fmax(0.0, p[1] - c->mx[1]));
    s = s + sqr(fmax(0.0, c->mn[2] - p[2]));
    s = s + sqr(fmax(0.0, p[2] - c->mx[2]));
    return s;
}
/* kdTree.py:127-156 queryAgentTreeRecursive / :232-262 queryObstacleTreeRecursive */
static void kd_query(const Node *tree, const int *ids, Query *q, double rangeSq, int node, int obstacle) {
    const Node *nd = &tree[node];
    if (nd->end - nd->begin <= MAX_LEAF) {
        for (int i = nd->begin; i < nd->end; i++) {
            if (obstacle) insert_obstacle_neighbor(q, ids[i], rangeSq);
            else insert_agent_neighbor(q, ids[i], rangeSq);
        }
    } else {
        const double *p = &q->pos[3 * q->self];
        double dl = box_dist_sq(&tree[nd->left], p), dr = box_dist_sq(&tree[nd->right], p);
        if (dl < dr) {
            if (dl < rangeSq) {
                kd_query(tree, ids, q, rangeSq, nd->left, obstacle);
                if (dr < rangeSq) kd_query(tree, ids, q, rangeSq, nd->right, obstacle);
            }
        } else {
            if (dr < rangeSq) {
                kd_query(tree, ids, q, rangeSq, nd->right, obstacle);
                if (dl < rangeSq) kd_query(tree, ids, q, rangeSq, nd->left, obstacle);
            }
        }
    }
}

/* standalone kd build for tests: returns node count used is implicit; tree_out = (2n-1) * 10 doubles */
void orc_kd_build(int n, const double *pos, int32_t *perm, double *tree_out) {
    if (n <= 0) return;
    Node *tree = (Node *)calloc((size_t)(2 * n), sizeof(Node));
    int *ids = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) ids[i] = perm[i];
    kd_build(tree, ids, pos, 0, n, 0);
    for (int i = 0; i < n; i++) perm[i] = ids[i];
    if (tree_out)
        for (int i = 0; i < 2 * n - 1; i++) {
            double *t = &tree_out[10 * i];
            t[0] = tree[i].begin; t[1] = tree[i].end; t[2] = tree[i].left; t[3] = tree[i].right;
            for (int k = 0; k < 3; k++) { t[4 + k] = tree[i].mn[k]; t[7 + k] = tree[i].mx[k]; }
        }
    free(tree); free(ids);
}

/* ------------------------------------------------------------------ velocity selection */
typedef struct { double apex[3], pA[3], pB[3], R; } Cone;           /* RVO_BA = [transl, pA, pB, R] scaPolicy.py:59 */
typedef struct { double p[3], n[3]; } Plane;                         /* orca3dPolicy.py:21-24 */
typedef struct { Plane pl; double R; double relPos[3]; float vBf[3]; double vBd[3]; int vB_is_f32; } OrcaOb; /* :107 */

typedef struct {
    const float *vA; double pos[3]; double vpref[3]; int policy; int zaxis; double pref_speed;
} AgentCtx;

typedef struct { double key; int idx; } KeyIdx;
static int cmp_keyidx(const void *a, const void *b) {
    const KeyIdx *x = (const KeyIdx *)a, *y = (const KeyIdx *)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* scaPolicy.py:119-145 / srvo3dPolicy.py:100-126 ; v_list given as ordered index list into cand */
static int shunted_strategy(const float *vA, double (*cand)[3], const int *order, int len, double thr) {
    int nopt = 1;
    double s0 = l3norm_mixed(cand[order[0]], vA);
    int i = 1;
    while (1) {
        if (fabs(s0 - l3norm_mixed(cand[order[i]], vA)) < thr) {
            nopt++; i++;
            if (i == len) break;
        } else break;
    }
    int imin = 0, imax = 0;
    double pmin = orc_get_phi(cand[order[0]]), pmax = pmin;
    for (int k = 1; k < nopt; k++) {
        double p = orc_get_phi(cand[order[k]]);
        if (p < pmin) { pmin = p; imin = k; }      /* min(): first minimal */
        if (p > pmax) { pmax = p; imax = k; }      /* max(): first maximal */
    }
    if (fabs(pmax - pmin) <= M_PI) return order[imin];   /* min(opt_v_best, key=get_phi) -> vmin */
    return order[imax];
}

/* scaPolicy.py:148-165 compute_without_suitV (RVO family) */
static double min_tc_rvo(const AgentCtx *a, const Cone *cones, int nc, const double *v, int *status) {
    double best = 0.0; int have = 0;
    for (int j = 0; j < nc; j++) {
        const Cone *c = &cones[j];
        double v_dif[3], pApB[3];
        for (int k = 0; k < 3; k++) { v_dif[k] = (v[k] + c->pA[k]) - c->apex[k]; pApB[k] = c->pB[k] - c->pA[k]; }
        if (is_intersect_pAB(pApB, c->R, v_dif, status) && satisfied_constraint(a->vA, a->pos[2], v)) {
            double discr = sqr(dot3(v_dif, pApB)) - dot3(v_dif, v_dif) * (dot3(pApB, pApB) - sqr(c->R));
            if (discr < 0) { *status |= ST_SQRT_DOMAIN; discr = 0; }   /* reference: ValueError */
            double tc = (dot3(v_dif, pApB) - sqrt(discr)) / dot3(v_dif, v_dif);
            if (tc < 0) tc = 0.0;
            if (!have || tc < best) { best = tc; have = 1; }
        }
    }
    return have ? best : 0.0;
}
/* orca3dPolicy.py:381-397 compute_without_suitV (ORCA sampled) */
static double min_tc_orca(const AgentCtx *a, const OrcaOb *obs, int nc, const double *v, int *status) {
    double best = 0.0; int have = 0;
    for (int j = 0; j < nc; j++) {
        const OrcaOb *o = &obs[j];
        double v_dif[3];
        int moving;
        if (o->vB_is_f32) moving = norm3f(o->vBf) > (float)1e-5;
        else moving = norm3(o->vBd) > 1e-5;
        if (moving) {
            if (o->vB_is_f32)
                for (int k = 0; k < 3; k++) { float h = 0.5f * (a->vA[k] + o->vBf[k]); v_dif[k] = v[k] - (double)h; }
            else
                for (int k = 0; k < 3; k++) { double h = 0.5 * ((double)a->vA[k] + o->vBd[k]); v_dif[k] = v[k] - h; }
        } else
            for (int k = 0; k < 3; k++) v_dif[k] = v[k];
        if (is_intersect_pAB(o->relPos, o->R, v_dif, status) && satisfied_constraint(a->vA, a->pos[2], v)) {
            double discr = sqr(dot3(v_dif, o->relPos)) - dot3(v_dif, v_dif) * (dot3(o->relPos, o->relPos) - sqr(o->R));
            if (discr < 0) { *status |= ST_SQRT_DOMAIN; discr = 0; }
            double tc = (dot3(v_dif, o->relPos) - sqrt(discr)) / dot3(v_dif, v_dif);
            if (tc < 0) tc = 0.0;
            if (!have || tc < best) { best = tc; have = 1; }
        }
    }
    return have ? best : 0.0;
}

/* scaPolicy.py:187-240, rvo3dPolicy.py:141-179, srvo3dPolicy.py:185-231, orca3dPolicy.py:400-439,
 * rvo3dDubinsPolicy.py:156-194 : candidate sweep + selection.  diag: [n_suit, fallback] */
static void intersect_select(const AgentCtx *a, const Cone *cones, const OrcaOb *orca, int nc, double *vpost,
                             int *diag, int *status) {
    static __thread double cand[1100][3];
    static __thread uint8_t suit[1100];
    static __thread KeyIdx keys[1100];
    static __thread int order[1100];
    int pol = a->policy;
    int num_N = (pol == POL_SCA && a->zaxis) ? 128 : 256;               /* scaPolicy.py:188-190 */
    int nc_tab = build_candidates(a->pref_speed, num_N, cand, 1024, status);
    int ncand = nc_tab + 1;
    for (int k = 0; k < 3; k++) cand[nc_tab][k] = a->vpref[k];           /* new_v = v_pref[:] */
    int n_suit = 0;
    for (int i = 0; i < ncand; i++) {
        /* compute_newV_is_suit: scaPolicy.py:168-184 / orca3dPolicy.py:365-378 */
        int ok = satisfied_constraint(a->vA, a->pos[2], cand[i]);
        if (ok) {
            for (int j = 0; j < nc; j++) {
                if (pol == POL_ORCA) {
                    double rel[3] = {cand[i][0] - orca[j].pl.p[0], cand[i][1] - orca[j].pl.p[1], cand[i][2] - orca[j].pl.p[2]};
                    if (!(dot3(rel, orca[j].pl.n) >= 0.0)) { ok = 0; break; }   /* is_inORCA :328-333 */
                } else {
                    const Cone *c = &cones[j];
                    double v_dif[3], pAB[3];
                    for (int k = 0; k < 3; k++) { v_dif[k] = (cand[i][k] + c->pA[k]) - c->apex[k]; pAB[k] = c->pB[k] - c->pA[k]; }
                    if (is_intersect_pAB(pAB, c->R, v_dif, status)) { ok = 0; break; }
                }
            }
        }
        suit[i] = (uint8_t)ok;
        n_suit += ok;
    }
    diag[0] = n_suit;
    int chosen;
    if (n_suit > 0) {
        diag[1] = 0;
        int m = 0;
        for (int i = 0; i < ncand; i++)
            if (suit[i]) { keys[m].key = orc_l3norm(cand[i], a->vpref); keys[m].idx = i; m++; }
        qsort(keys, (size_t)m, sizeof(KeyIdx), cmp_keyidx);              /* stable sort by key */
        for (int i = 0; i < m; i++) order[i] = keys[i].idx;
        if ((pol == POL_SCA || pol == POL_SRVO) && m > 1)
            chosen = shunted_strategy(a->vA, cand, order, m, pol == POL_SCA ? 3e-2 : 1e-1);
        else
            chosen = order[0];
    } else {
        diag[1] = 1;
        for (int i = 0; i < ncand; i++) {
            double tc = (pol == POL_ORCA) ? min_tc_orca(a, orca, nc, cand[i], status)
                                          : min_tc_rvo(a, cones, nc, cand[i], status);
            double tcv = tc + 1e-5;
            keys[i].key = (0.2 / tcv) + orc_l3norm(cand[i], a->vpref);
            keys[i].idx = i;
        }
        if (pol == POL_SCA || pol == POL_SRVO) {
            qsort(keys, (size_t)ncand, sizeof(KeyIdx), cmp_keyidx);
            for (int i = 0; i < ncand; i++) order[i] = keys[i].idx;
            if (ncand > 1) chosen = shunted_strategy(a->vA, cand, order, ncand, pol == POL_SCA ? 5e-2 : 1e-1);
            else chosen = order[0];
        } else {
            int best = 0;                                                /* min(): first minimal */
            for (int i = 1; i < ncand; i++) if (keys[i].key < keys[best].key) best = i;
            chosen = best;
        }
    }
    for (int k = 0; k < 3; k++) vpost[k] = orc_trunc5(cand[chosen][k]);
    diag[2] = chosen;
}

/* ------------------------------------------------------------------ ORCA planes (orca3dPolicyOfficial.py:56-106) */
static void build_orca_plane(const float *vA, const double *pA, const double *pB, const float *vBf, const double *vBd,
                             int vB_is_f32, double rA, double rB, OrcaOb *out) {
    double invTimeHorizon = 1.0 / g_par.time_horizon;
    double relPos[3], relVel[3];
    for (int k = 0; k < 3; k++) relPos[k] = pB[k] - pA[k];
    if (vB_is_f32) for (int k = 0; k < 3; k++) relVel[k] = (double)(float)(vA[k] - vBf[k]);    /* f32 - f32 */
    else for (int k = 0; k < 3; k++) relVel[k] = (double)vA[k] - vBd[k];
    double distSq = dot3(relPos, relPos);
    double agent_rad = rA + 0.05, obj_rad = rB + 0.05;
    double R = agent_rad + obj_rad;
    double RSq = sqr(R);
    double u[3], nrm[3];
    if (distSq > RSq) {
        double w[3];
        for (int k = 0; k < 3; k++) w[k] = relVel[k] - invTimeHorizon * relPos[k];
        double wLengthSq = dot3(w, w);
        double dotProduct = dot3(w, relPos);
        if (dotProduct < 0.0 && sqr(dotProduct) > RSq * wLengthSq) {
            double wLength = sqrt(wLengthSq);
            for (int k = 0; k < 3; k++) nrm[k] = w[k] / wLength;
            double s = R * invTimeHorizon - wLength;
            for (int k = 0; k < 3; k++) u[k] = s * nrm[k];
        } else {
            double difSq = distSq - RSq;
            double dot_product = dot3(relPos, relVel);
            double cr[3];
            cross3(relPos, relVel, cr);
            double wwSq = dot3(cr, cr) / difSq;
            double pApBLength = norm3(relPos);
            double pAp1Length = dot_product / pApBLength;
            double p1otLength = sqrt(wwSq) * (R / pApBLength);
            double pAotlength = pAp1Length + p1otLength;
            double t = pAotlength / pApBLength;
            double ww[3];
            for (int k = 0; k < 3; k++) ww[k] = relVel[k] - t * relPos[k];
            double wwLength = norm3(ww);
            for (int k = 0; k < 3; k++) nrm[k] = ww[k] / wwLength;
            double s = R * t - wwLength;
            for (int k = 0; k < 3; k++) u[k] = s * nrm[k];
        }
    } else {
        double invTimeStep = 1.0 / g_par.time_step;
        double w[3];
        for (int k = 0; k < 3; k++) w[k] = relVel[k] - invTimeStep * relPos[k];
        double wLength = norm3(w);
        for (int k = 0; k < 3; k++) nrm[k] = w[k] / wLength;
        double s = R * invTimeStep - wLength;
        for (int k = 0; k < 3; k++) u[k] = s * nrm[k];
    }
    for (int k = 0; k < 3; k++) { out->pl.p[k] = (double)vA[k] + 0.5 * u[k]; out->pl.n[k] = nrm[k]; out->relPos[k] = relPos[k]; }
    out->R = R;
    out->vB_is_f32 = vB_is_f32;
    for (int k = 0; k < 3; k++) { out->vBf[k] = vB_is_f32 ? vBf[k] : 0.0f; out->vBd[k] = vB_is_f32 ? 0.0 : vBd[k]; }
}

/* ------------------------------------------------------------------ LP (orca3dPolicyOfficial.py:126-300) */
static int lp1(const Plane *pl, int planeNo, const double *lpnt, const double *ldir, double maxSpeed,
               const double *vpref, int dir_opt, double *nv) {
    double dotProduct = dot3(lpnt, ldir);
    double disc = sqr(dotProduct) + sqr(maxSpeed) - dot3(lpnt, lpnt);
    if (disc < 0.0) return 0;
    double sq = sqrt(disc);
    double tLeft = -dotProduct - sq, tRight = -dotProduct + sq;
    for (int i = 0; i < planeNo; i++) {
        double d[3] = {pl[i].p[0] - lpnt[0], pl[i].p[1] - lpnt[1], pl[i].p[2] - lpnt[2]};
        double numerator = dot3(d, pl[i].n);
        double denominator = dot3(ldir, pl[i].n);
        if (sqr(denominator) <= RVO_EPS) {
            if (numerator > 0.0) return 0;
            continue;
        }
        double t = numerator / denominator;
        if (denominator >= 0.0) { if (t > tLeft) tLeft = t; }
        else { if (t < tRight) tRight = t; }
        if (tLeft > tRight) return 0;
    }
    double tt;
    if (dir_opt) tt = (dot3(vpref, ldir) > 0.0) ? tRight : tLeft;
    else {
        double d[3] = {vpref[0] - lpnt[0], vpref[1] - lpnt[1], vpref[2] - lpnt[2]};
        double t = dot3(ldir, d);
        tt = (t < tLeft) ? tLeft : (t > tRight ? tRight : t);
    }
    for (int k = 0; k < 3; k++) nv[k] = lpnt[k] + tt * ldir[k];
    return 1;
}
static int lp2(const Plane *pl, int planeNo, double maxSpeed, const double *vpref, int dir_opt, double *nv) {
    const Plane *P = &pl[planeNo];
    double planeDist = dot3(P->p, P->n);
    double planeDistSq = sqr(planeDist), radiusSq = sqr(maxSpeed);
    if (planeDistSq > radiusSq) return 0;
    double planeRadiusSq = radiusSq - planeDistSq;
    double center[3] = {planeDist * P->n[0], planeDist * P->n[1], planeDist * P->n[2]};
    if (dir_opt) {
        double dp = dot3(vpref, P->n);
        double pov[3] = {vpref[0] - dp * P->n[0], vpref[1] - dp * P->n[1], vpref[2] - dp * P->n[2]};
        double lsq = dot3(pov, pov);
        if (lsq <= RVO_EPS) for (int k = 0; k < 3; k++) nv[k] = center[k];
        else { double s = sqrt(planeRadiusSq / lsq); for (int k = 0; k < 3; k++) nv[k] = center[k] + s * pov[k]; }
    } else {
        double d[3] = {P->p[0] - vpref[0], P->p[1] - vpref[1], P->p[2] - vpref[2]};
        double dp = dot3(d, P->n);
        for (int k = 0; k < 3; k++) nv[k] = vpref[k] + dp * P->n[k];
        if (dot3(nv, nv) > radiusSq) {
            double res[3] = {nv[0] - center[0], nv[1] - center[1], nv[2] - center[2]};
            double rl = dot3(res, res);
            double s = sqrt(planeRadiusSq / rl);
            for (int k = 0; k < 3; k++) nv[k] = center[k] + s * res[k];
        }
    }
    for (int i = 0; i < planeNo; i++) {
        double d[3] = {pl[i].p[0] - nv[0], pl[i].p[1] - nv[1], pl[i].p[2] - nv[2]};
        if (dot3(pl[i].n, d) > 0.0) {
            double cp[3];
            cross3(pl[i].n, P->n, cp);
            if (dot3(cp, cp) <= RVO_EPS) return 0;
            double nn = norm3(cp);
            double ldir[3] = {cp[0] / nn, cp[1] / nn, cp[2] / nn};
            double lineNormal[3];
            cross3(ldir, P->n, lineNormal);
            double e[3] = {pl[i].p[0] - P->p[0], pl[i].p[1] - P->p[1], pl[i].p[2] - P->p[2]};
            double dp1 = dot3(e, pl[i].n), dp2 = dot3(lineNormal, pl[i].n);
            double q = dp1 / dp2;
            double lpnt[3] = {P->p[0] + q * lineNormal[0], P->p[1] + q * lineNormal[1], P->p[2] + q * lineNormal[2]};
            if (!lp1(pl, i, lpnt, ldir, maxSpeed, vpref, dir_opt, nv)) return 0;
        }
    }
    return 1;
}
static int lp3(const Plane *pl, int np, double maxSpeed, const double *vpref, int dir_opt, double *nv) {
    if (dir_opt) for (int k = 0; k < 3; k++) nv[k] = vpref[k] * maxSpeed;
    else if (dot3(vpref, vpref) > sqr(maxSpeed)) {
        double nn = norm3(vpref);
        for (int k = 0; k < 3; k++) nv[k] = (vpref[k] / nn) * maxSpeed;
    } else for (int k = 0; k < 3; k++) nv[k] = vpref[k];
    for (int i = 0; i < np; i++) {
        double d[3] = {pl[i].p[0] - nv[0], pl[i].p[1] - nv[1], pl[i].p[2] - nv[2]};
        if (dot3(pl[i].n, d) > 0.0) {
            double tmp[3] = {nv[0], nv[1], nv[2]};
            if (!lp2(pl, i, maxSpeed, vpref, dir_opt, nv)) {
                nv[0] = tmp[0]; nv[1] = tmp[1]; nv[2] = tmp[2];
                return i;
            }
        }
    }
    return np;
}
static void lp4(const Plane *pl, int np, int beginPlane, double radius, double *nv) {
    Plane proj[K_MAX];
    for (int i = beginPlane; i < np; i++) {
        /* :264  np.dot(normal, (point - new_velocity) > 0.0): the comparison is INSIDE the dot */
        double mask[3];
        for (int k = 0; k < 3; k++) mask[k] = ((pl[i].p[k] - nv[k]) > 0.0) ? 1.0 : 0.0;
        double s = dot3(pl[i].n, mask);
        if (s != 0.0) {
            int m = 0;
            for (int j = 0; j < i; j++) {
                Plane q;
                double cp[3];
                cross3(pl[j].n, pl[i].n, cp);
                if (dot3(cp, cp) <= RVO_EPS) {
                    if (dot3(pl[i].n, pl[j].n) > 0.0) continue;
                    for (int k = 0; k < 3; k++) q.p[k] = 0.5 * (pl[i].p[k] + pl[j].p[k]);
                } else {
                    double lineNormal[3];
                    cross3(cp, pl[i].n, lineNormal);
                    double e[3] = {pl[j].p[0] - pl[i].p[0], pl[j].p[1] - pl[i].p[1], pl[j].p[2] - pl[i].p[2]};
                    double dp1 = dot3(e, pl[j].n), dp2 = dot3(lineNormal, pl[j].n);
                    double qq = dp1 / dp2;
                    for (int k = 0; k < 3; k++) q.p[k] = pl[i].p[k] + qq * lineNormal[k];
                }
                double dn[3] = {pl[j].n[0] - pl[i].n[0], pl[j].n[1] - pl[i].n[1], pl[j].n[2] - pl[i].n[2]};
                double nn = norm3(dn);
                for (int k = 0; k < 3; k++) q.n[k] = dn[k] / nn;
                proj[m++] = q;
            }
            double tmp[3] = {nv[0], nv[1], nv[2]};
            if (lp3(proj, m, radius, pl[i].n, 1, nv) < m) { nv[0] = tmp[0]; nv[1] = tmp[1]; nv[2] = tmp[2]; }
        }
    }
}

/* straight-line compute_v_pref: rvo3dPolicy.py:182-196 (l3norm) / orca3dPolicy.py:348-362 (distance) */
static void straight_v_pref(const double *goal, const double *pos, double pref_speed, int use_distance, double *V_des,
                            double *v_pref_raw) {
    double zero[3] = {0, 0, 0};
    double dif[3] = {goal[0] - pos[0], goal[1] - pos[1], goal[2] - pos[2]};
    double nrm = use_distance ? orc_distance(dif, zero) : orc_l3norm(dif, zero);
    nrm = py_int(nrm * EPS5) / EPS5;
    double v[3];
    for (int k = 0; k < 3; k++) v[k] = dif[k] * pref_speed / nrm;
    if (orc_l3norm(goal, pos) < 0.2) v[0] = v[1] = v[2] = 0.0;          /* util.reached :23, bound 0.2 */
    for (int k = 0; k < 3; k++) { v_pref_raw[k] = v[k]; V_des[k] = orc_trunc5(v[k]); }
}
void orc_straight_v_pref(const double *goal, const double *pos, double pref_speed, int use_distance, double *V_des) {
    double raw[3];
    straight_v_pref(goal, pos, pref_speed, use_distance, V_des, raw);
}

/* ------------------------------------------------------------------ one policy pass over all agents
 * = the first loop of MACAEnv._take_action (mampenv.py:28-40) + every find_next_action.
 * vpref_mode[i]: 0 straight line computed here, 1 = vpref_ext[i] supplied (Dubins tracker output, scaPolicy.py:264).
 * diag: n*5 = [n_suit, fallback, chosen, plane_fail, lp4_ran]   (-1 when not applicable) */
int orc_policy_step(int n, int m, const double *pos, const float *vel, const double *heading, const double *radius,
                    const double *pref_speed, uint8_t *flags, const double *goal, const uint8_t *policy,
                    const uint8_t *zaxis, const double *vpref_ext, const uint8_t *vpref_mode, int32_t *perm,
                    const double *obs_pos, const double *obs_radius, double *action64, float *action32,
                    int32_t *nbr_n, int32_t *nbr_id, uint8_t *nbr_kind, double *nbr_dsq, uint8_t *nbr_valid,
                    double *vpref_used, int32_t *diag, int32_t *status, int nthreads) {
    Node *atree = (Node *)calloc((size_t)(2 * n + 1), sizeof(Node));
    int *aids = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; i++) aids[i] = perm[i];
    if (n > 0) kd_build(atree, aids, pos, 0, n, 0);                     /* mampenv.py:28 */
    for (int i = 0; i < n; i++) perm[i] = aids[i];
    Node *otree = NULL; int *oids = NULL;
    if (m > 0) {                                                         /* mampenv.py:20, built once from identity */
        otree = (Node *)calloc((size_t)(2 * m + 1), sizeof(Node));
        oids = (int *)malloc(sizeof(int) * (size_t)m);
        for (int i = 0; i < m; i++) oids[i] = i;
        kd_build(otree, oids, obs_pos, 0, m, 0);
    }
    uint8_t *flags_in = (uint8_t *)malloc((size_t)n + 1);
    memcpy(flags_in, flags, (size_t)n);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 8)
#endif
    for (int i = 0; i < n; i++) {
        par_for_agent(i);                                                /* the attributes find_next_action reads off THIS agent */
        double rangeSq = pow(g_par.neighbor_dist, 2.0);                  /* scaPolicy.py:112 */
        int st = 0;
        for (int k = 0; k < 7; k++) { action64[7 * i + k] = 0.0; action32[7 * i + k] = 0.0f; }
        nbr_valid[i] = 0; nbr_n[i] = 0;
        for (int k = 0; k < 5; k++) diag[5 * i + k] = -1;
        for (int k = 0; k < 3; k++) vpref_used[3 * i + k] = NAN;
        status[i] = 0;
        if (flags_in[i] & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT)) continue;   /* mampenv.py:35 */
        int pol = policy[i];
        const float *vA = &vel[3 * i];
        const double *pA = &pos[3 * i];
        double vpref[3], raw[3];
        if (vpref_mode && vpref_mode[i]) for (int k = 0; k < 3; k++) vpref[k] = vpref_ext[3 * i + k];
        else straight_v_pref(&goal[3 * i], pA, pref_speed[i], pol == POL_ORCA || pol == POL_ORCA_LP, vpref, raw);
        for (int k = 0; k < 3; k++) vpref_used[3 * i + k] = vpref[k];
        NbrList L; L.n = 0;
        Query q = {i, pos, radius, obs_pos, obs_radius, 0, &L};
        double vpost[3];
        int is_orca = (pol == POL_ORCA || pol == POL_ORCA_LP);
        int first_step;
        if (is_orca) {
            /* orca3dPolicy.py:51-53: neighbours are computed before the first-step test */
            if (otree) kd_query(otree, oids, &q, rangeSq, 0, 1);
            kd_query(atree, aids, &q, rangeSq, 0, 0);
            nbr_valid[i] = 1;
            first_step = distance_f32_zero(vA) <= 1e-5;
        } else {
            first_step = l3norm_f32_zero(vA) <= 1e-5;                   /* scaPolicy.py:34 */
            if (!first_step) {
                if (otree) kd_query(otree, oids, &q, rangeSq, 0, 1);     /* scaPolicy.py:114 obstacles first */
                kd_query(atree, aids, &q, rangeSq, 0, 0);
                nbr_valid[i] = 1;
            }
        }
        if (q.is_collision) flags[i] |= FLAG_COLLISION;
        nbr_n[i] = L.n;
        for (int k = 0; k < L.n; k++) {
            nbr_id[K_MAX * i + k] = L.id[k]; nbr_kind[K_MAX * i + k] = L.kind[k]; nbr_dsq[K_MAX * i + k] = L.dsq[k];
        }
        for (int k = L.n; k < K_MAX; k++) { nbr_id[K_MAX * i + k] = -1; nbr_kind[K_MAX * i + k] = 0; nbr_dsq[K_MAX * i + k] = 0; }
        if (first_step) {
            for (int k = 0; k < 3; k++) vpost[k] = 0.3 * vpref[k];      /* scaPolicy.py:38 */
        } else {
            AgentCtx a;
            a.vA = vA; a.policy = pol; a.zaxis = zaxis ? zaxis[i] : 0; a.pref_speed = pref_speed[i];
            for (int k = 0; k < 3; k++) { a.pos[k] = pA[k]; a.vpref[k] = vpref[k]; }
            Cone cones[K_MAX]; OrcaOb orca[K_MAX];
            double agent_rad = radius[i] + 0.05;
            for (int j = 0; j < L.n; j++) {
                int o = L.id[j];
                int isob = L.kind[j];
                const double *pB = isob ? &obs_pos[3 * o] : &pos[3 * o];
                double rB = isob ? obs_radius[o] : radius[o];
                if (is_orca) {
                    double zerod[3] = {0, 0, 0};
                    build_orca_plane(vA, pA, pB, isob ? NULL : &vel[3 * o], zerod, !isob, radius[i], rB, &orca[j]);
                } else {
                    /* scaPolicy.py:47-60 ; obstacles are always is_at_goal (obstacle.py:22) */
                    int at_goal = isob ? 1 : ((flags_in[o] & FLAG_AT_GOAL) != 0);
                    Cone *c = &cones[j];
                    for (int k = 0; k < 3; k++) {
                        c->pA[k] = pA[k]; c->pB[k] = pB[k];
                        if (at_goal) c->apex[k] = pA[k];
                        else { float h = 0.5f * (vel[3 * o + k] + vA[k]); c->apex[k] = pA[k] + (double)h; }
                    }
                    double obj_rad = rB + 0.05;
                    c->R = obj_rad + agent_rad;
                }
            }
            if (pol == POL_ORCA_LP) {
                Plane pls[K_MAX];
                for (int j = 0; j < L.n; j++) pls[j] = orca[j].pl;
                double nv[3] = {0, 0, 0};
                int planeFail = lp3(pls, L.n, g_par.max_speed, vpref, 0, nv);
                diag[5 * i + 3] = planeFail;
                diag[5 * i + 4] = 0;
                if (planeFail < L.n) { lp4(pls, L.n, planeFail, g_par.max_speed, nv); diag[5 * i + 4] = 1; }
                for (int k = 0; k < 3; k++) vpost[k] = nv[k];
            } else {
                int dg[3] = {-1, -1, -1};
                intersect_select(&a, cones, orca, L.n, vpost, dg, &st);
                diag[5 * i + 0] = dg[0]; diag[5 * i + 1] = dg[1]; diag[5 * i + 2] = dg[2];
            }
        }
        orc_cartesian2spherical(&heading[3 * i], vpost, pol == POL_ORCA_LP, &action64[7 * i]);
        for (int k = 0; k < 7; k++) action32[7 * i + k] = (float)action64[7 * i + k];   /* mampenv.py:31,40 */
        status[i] = st;
    }
    free(atree); free(aids); free(otree); free(oids); free(flags_in);
    g_par = g_ctx;                  /* (the calling thread ran some agents too: the scalar KAT helpers must see the scene's values again) */
    return 0;
}

/* ------------------------------------------------------------------ env update
 * second loop of _take_action (mampenv.py:42-46): update_velocitie :83-105, check_agent_state :61-80,
 * then is_done :51-59.  Sequential on purpose: agent i is checked against already-moved j<i and
 * not-yet-moved j>i, exactly as the reference does. */
int orc_env_update(int n, int m, double *pos, float *vel, double *heading, const double *radius, uint8_t *flags,
                   const double *goal, const float *action32, double *total_dist, const double *max_run_dist,
                   int32_t *step_num, const double *obs_pos, const double *obs_radius) {
    for (int i = 0; i < n; i++) {
        par_for_agent(i);
        const double dt = g_par.dt_nominal;                              /* agent.dt_nominal (agent.py:41) */
        const float *act = &action32[7 * i];
        double speed = (double)act[3];
        double a = orc_pi_2_pi(heading[3 * i + 0] + (double)act[4]);
        double b = orc_pi_2_pi(heading[3 * i + 1] + (double)act[5]);
        double g = orc_pi_2_pi(heading[3 * i + 2] + (double)act[6]);
        double dx = speed * cos(b) * cos(a) * dt;
        double dy = speed * cos(b) * sin(a) * dt;
        double dz = speed * sin(b) * dt;
        double length = sqrt(pow(dx, 2.0) + pow(dy, 2.0) + pow(dz, 2.0));
        total_dist[i] += length;
        pos[3 * i] += dx; pos[3 * i + 1] += dy; pos[3 * i + 2] += dz;
        heading[3 * i] = a; heading[3 * i + 1] = b; heading[3 * i + 2] = g;
        vel[3 * i] = act[0]; vel[3 * i + 1] = act[1]; vel[3 * i + 2] = act[2];
        if (!(flags[i] & FLAG_AT_GOAL)) step_num[i] += 1;
        for (int o = 0; o < m; o++)
            if (orc_l3norm(&pos[3 * i], &obs_pos[3 * o]) <= radius[i] + obs_radius[o]) flags[i] |= FLAG_COLLISION;
        for (int j = 0; j < n; j++) {
            if (j == i) continue;
            if (orc_l3norm(&pos[3 * i], &pos[3 * j]) <= radius[i] + radius[j]) {
                if (!(flags[j] & FLAG_AT_GOAL)) flags[j] |= FLAG_COLLISION;
                if (!(flags[i] & FLAG_AT_GOAL)) flags[i] |= FLAG_COLLISION;
            }
        }
        if (total_dist[i] > max_run_dist[i]) flags[i] |= FLAG_TIMEOUT;
    }
    int all_done = 1;
    for (int i = 0; i < n; i++) {
        if (orc_l3norm(&pos[3 * i], &goal[3 * i]) <= g_ctx.near_goal_threshold) flags[i] |= FLAG_AT_GOAL;
        if (!(flags[i] & (FLAG_AT_GOAL | FLAG_COLLISION | FLAG_TIMEOUT))) all_done = 0;
    }
    g_par = g_ctx;                  /* (the last agent's attributes must not outlive the call on this thread) */
    return all_done;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
