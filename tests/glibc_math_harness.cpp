// Host check of sca_amd/csrc/sca_glibc_math.h against the running glibc: bit equality on random and boundary arguments.
// usage: glibc_math_harness <million arguments per case>;  prints one line per case: name count mismatches [first bad argument]
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "sca_glibc_math.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static inline double uni(double lo, double hi) { return lo + (hi - lo) * ((rnd() >> 11) * (1.0 / 9007199254740992.0)); }
static inline double logmag(double lo_exp, double hi_exp) { const double m = std::exp2(uni(lo_exp, hi_exp)); return (rnd() & 1) ? m : -m; }
static inline bool same(double a, double b) { return sca_gm::bits(a) == sca_gm::bits(b) || (a != a && b != b); }
static double (*volatile L_sin)(double) = sin;
static double (*volatile L_cos)(double) = cos;
static double (*volatile L_acos)(double) = acos;
static double (*volatile L_atan2)(double, double) = atan2;
static double (*volatile L_pow)(double, double) = pow;

static long total_bad = 0;
static bool use_ref = false;
static inline double my_atan2(double y, double x) { return use_ref ? sca_gm::g_atan2_ref(y, x) : sca_gm::g_atan2(y, x); }
template <class F, class G, class A> static void run1(const char *name, long n, F f, G g, A arg) {
    long bad = 0; double first = 0;
    for (long i = 0; i < n; i++) { const double x = arg(); if (!same(f(x), g(x))) { if (!bad) first = x; bad++; } }
    printf("%-34s %10ld %8ld", name, n, bad);
    if (bad) printf("   first %.17g: mine %.17g glibc %.17g", first, f(first), g(first));
    printf("\n");
    total_bad += bad;
}
template <class A> static void run2(const char *name, long n, A arg) {
    long bad = 0; double fy = 0, fx = 0;
    for (long i = 0; i < n; i++) { double y, x; arg(y, x); if (!same(my_atan2(y, x), L_atan2(y, x))) { if (!bad) { fy = y; fx = x; } bad++; } }
    printf("%-34s %10ld %8ld", name, n, bad);
    if (bad) printf("   first atan2(%.17g, %.17g): mine %.17g glibc %.17g", fy, fx, my_atan2(fy, fx), L_atan2(fy, fx));
    printf("\n");
    total_bad += bad;
}

int main(int argc, char **argv) {
    const long M = (long)(1e6 * (argc > 1 ? atof(argv[1]) : 1.0));
    // argv[2] == "ref": the literal restatements (glibc's control flow); default: the branch-free forms the kernels use
    const bool ref = argc > 2 && !strcmp(argv[2], "ref");
    use_ref = ref;
    auto gs = [ref](double x) { return ref ? sca_gm::g_sin_ref(x) : sca_gm::g_sin(x); };
    auto gc = [ref](double x) { if (ref) return sca_gm::g_cos_ref(x); double s, c; sca_gm::g_sincos(x, s, c); const double c1 = sca_gm::g_cos(x);
                                return sca_gm::bits(c) == sca_gm::bits(c1) && sca_gm::bits(s) == sca_gm::bits(sca_gm::g_sin(x)) ? c : NAN; };
    auto ga = [](double x) { return sca_gm::g_acos(x); };
    auto gp = [ref](double x) { return ref ? sca_gm::g_pow2_ref(x) : sca_gm::g_pow2(x); };
    auto ls = [](double x) { return L_sin(x); };
    auto lc = [](double x) { return L_cos(x); };
    auto la = [](double x) { return L_acos(x); };
    auto lp = [](double x) { return L_pow(x, 2.0); };
    // sin / cos: every range of s_sin.c and its boundaries
    const double edges[] = {0.126, 0.855469, 2.426265, 105414350.0, 1.4901161193847656e-08, 7.450580596923828e-09, 0.0078125, 1.5707963267948966,
                            3.141592653589793, 6.283185307179586, 0.7853981633974483};
    for (int which = 0; which < 2; which++) {
        const char *nm = which ? "cos" : "sin";
        char buf[64];
        auto F = [&](const char *tag, long n, auto arg) { snprintf(buf, sizeof buf, "%s %s", nm, tag); if (which) run1(buf, n, gc, lc, arg); else run1(buf, n, gs, ls, arg); };
        F("[-0.126, 0.126]", M, [] { return uni(-0.126, 0.126); });
        F("[-0.86, 0.86]", 2 * M, [] { return uni(-0.86, 0.86); });
        F("[-2.43, 2.43]", 2 * M, [] { return uni(-2.43, 2.43); });
        F("[-2pi, 2pi]", 4 * M, [] { return uni(-6.283185307179586, 6.283185307179586); });
        F("[-1000, 1000]", 2 * M, [] { return uni(-1000, 1000); });
        F("[-1e8, 1e8]", 2 * M, [] { return uni(-1.05e8, 1.05e8); });
        F("2^[-60, 27]", 2 * M, [] { return logmag(-60, 26.6); });
        F("near range edges", M, [&] { const double e = edges[rnd() % 11]; const double x = e * (1.0 + uni(-1e-12, 1e-12)); return (rnd() & 1) ? x : -x; });
        F("near k*pi/2", M, [] { const double k = (double)(rnd() % 4000); const double x = k * 1.5707963267948966 + uni(-1e-9, 1e-9); return (rnd() & 1) ? x : -x; });
        F("specials", 16, [] { static int i = 0; const double v[] = {0.0, -0.0, 5e-324, -5e-324, 1e-300, INFINITY, -INFINITY, NAN, 1.0, -1.0, 2.2250738585072014e-308, 1e-10, -1e-10, 0.5, 3.0, 100.0}; return v[i++ % 16]; });
    }
    // acos
    run1("acos [-1, 1]", 4 * M, ga, la, [] { return uni(-1, 1); });
    run1("acos [-0.13, 0.13]", M, ga, la, [] { return uni(-0.13, 0.13); });
    run1("acos 1 - 2^[-52, -3]", 2 * M, ga, la, [] { const double x = 1.0 - std::exp2(uni(-52, -3)); return (rnd() & 1) ? x : -x; });
    run1("acos 2^[-60, 0]", M, ga, la, [] { return logmag(-60, 0); });
    run1("acos near interval edges", M, ga, la, [] { const double e[] = {0.125, 0.25, 0.5, 0.75, 0.921875, 0.953125, 0.96875, 0.9687957763671875};
                                                    const double x = e[rnd() % 8] * (1.0 + uni(-1e-13, 1e-13)); return (rnd() & 1) ? x : -x; });
    run1("acos specials", 12, ga, la, [] { static int i = 0; const double v[] = {0.0, -0.0, 1.0, -1.0, 1.0000000000000002, -1.5, NAN, INFINITY, 0.9999999999999999, -0.9999999999999999, 1e-20, 0.125}; return v[i++ % 12]; });
    // pow(x, 2)
    run1("pow(x, 2) [-100, 100]", 4 * M, gp, lp, [] { return uni(-100, 100); });
    run1("pow(x, 2) [-2, 2]", 2 * M, gp, lp, [] { return uni(-2, 2); });
    run1("pow(x, 2) 2^[-60, 60]", 2 * M, gp, lp, [] { return logmag(-60, 60); });
    run1("pow(x, 2) 2^[-359, 359]", M, gp, lp, [] { return logmag(-359, 359); });
    run1("pow(x, 2) near 1", M, gp, lp, [] { return 1.0 + uni(-1e-6, 1e-6) * std::exp2(uni(-40, 0)); });
    run1("pow(x, 2) specials", 10, gp, lp, [] { static int i = 0; const double v[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 1e-200, 1e200, 3.0}; return v[i++ % 10]; });
    // atan2
    run2("atan2 [-3, 3]^2", 8 * M, [](double &y, double &x) { y = uni(-3, 3); x = uni(-3, 3); });
    run2("atan2 [-1e5, 1e5]^2", 4 * M, [](double &y, double &x) { y = uni(-1e5, 1e5); x = uni(-1e5, 1e5); });
    run2("atan2 2^[-40, 40] both", 4 * M, [](double &y, double &x) { y = logmag(-40, 40); x = logmag(-40, 40); });
    run2("atan2 ratio 2^[-70, 70]", 2 * M, [](double &y, double &x) { x = logmag(-20, 20); y = x * std::exp2(uni(-70, 70)) * ((rnd() & 1) ? 1 : -1); });
    run2("atan2 (+-2, p)", 2 * M, [](double &y, double &x) { y = (rnd() & 1) ? 2.0 : -2.0; x = uni(0, 5e4); });
    run2("atan2 |y| ~ |x|", 2 * M, [](double &y, double &x) { x = logmag(-10, 10); y = x * (1.0 + uni(-1e-12, 1e-12)) * ((rnd() & 1) ? 1 : -1); });
    run2("atan2 ratio near 1/16, k/256", 2 * M, [](double &y, double &x) { x = logmag(-3, 3); const double r = (rnd() % 256 + 1) / 256.0 * (1.0 + uni(-1e-13, 1e-13));
                                                                            if (rnd() & 1) { y = x * r; } else { y = x; x = y * r; } if (rnd() & 1) y = -y; });
    run2("atan2 2^[-1000, 1000] both", 2 * M, [](double &y, double &x) { y = logmag(-1000, 1000); x = logmag(-1000, 1000); });
    run2("atan2 specials", 400, [](double &y, double &x) { static int i = 0; const double v[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 5e-324, -5e-324, 1e-310, 1e308, -1e308,
                                                                                                  2.2250738585072014e-308, 1e-160, 1e160, 2.0, 0.0625, 16.0, 1e-20, -1e20}; y = v[i % 20]; x = v[(i / 20) % 20]; i++; });
    printf("TOTAL mismatches %ld\n", total_bad);
    return total_bad ? 1 : 0;
}
