#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
// Host check of the scorer's log(1 + a) (rt_kernels.hpp log1p_pos): the same float64 operations in the same order, with the
// reciprocal seeds taken from a float division (the device takes v_rcp_f32: 1 ulp of float either way, refined to double
// by the Newton steps below), against long-double log1pl over 2e7 arguments and the corner cases.
//   gcc -O2 -ffp-contract=off -o /tmp/log1p_check scripts/log1p_check.c -lm && /tmp/log1p_check
// -> worst 2.0 ulp, 3.4e-16 relative
static inline double rcp_seed(double d) { return (double)(1.0f / (float)d); }   // stands for v_rcp_f32 of the rounded value
static double log1p_pos(double a) {
    const double w = 1.0 + a;
    double c = (a >= 1.0) ? 1.0 - (w - a) : a - (w - 1.0);        // 1 + a = w + c exactly
    int e; double m = frexp(w, &e);                                // m in [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }          // m in [sqrt(1/2), sqrt(2))
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r = rcp_seed(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    double s = f * r;
    s = fma(r, fma(-d, s, f), s);                                  // s = f / (2 + f), correctly rounded up to an ulp
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = fma(p, z, 1.0 / 19.0); p = fma(p, z, 1.0 / 17.0); p = fma(p, z, 1.0 / 15.0); p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0); p = fma(p, z, 1.0 / 9.0);  p = fma(p, z, 1.0 / 7.0);  p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    // log(m) = 2 s + 2 s z p;  c / w: first order of the rounding error of 1 + a
    double rw = rcp_seed(w);
    rw = fma(rw, fma(-w, rw, 1.0), rw);
    const double corr = c * rw;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double ed = (double)e;
    const double s2 = s + s;
    double res = fma(ed, ln2_hi, s2) + (fma(s2 * z, p, ed * ln2_lo) + corr);
    if (!(a <= 1.7976931348623157e308)) res = a;                   // inf, NaN
    return res;
}
static double ulp_err(double got, double want) {
    if (got == want) return 0; if (isnan(want)) return isnan(got) ? 0 : 1e9; if (isinf(want)) return 1e9;
    double u = nextafter(fabs(want), INFINITY) - fabs(want);
    return fabs(got - want) / u;
}
int main(int argc, char** argv) {
    const int count = argc > 1 ? atoi(argv[1]) : 20000000;
    double worst = 0, worst_a = 0, worst_rel = 0; srand48(1);
    long n = 0;
    for (int i = 0; i < count; ++i) {
        double a;
        int k = i % 4;
        if (k == 0) a = exp((drand48() * 2 - 1) * 700);
        else if (k == 1) a = drand48() * 3;
        else if (k == 2) a = exp((drand48() * 2 - 1) * 40);
        else a = ldexp(1.0 + drand48() * 1e-6 * (drand48() < .5 ? -1 : 1), (int)(drand48() * 200 - 100));
        long double wl = log1pl((long double)a);
        double got = log1p_pos(a);
        double u = nextafter(fabs((double)wl), INFINITY) - fabs((double)wl);
        double err = fabsl((long double)got - wl) / u;
        if (err > worst) { worst = err; worst_a = a; }
        double rel = fabsl(((long double)got - wl) / wl); if (rel > worst_rel) worst_rel = rel;
        ++n;
    }
    printf("n %ld worst %.3f ulp at a = %.17g, worst rel %.3g\n", n, worst, worst_a, worst_rel);
    double sp[] = {0.0, 5e-324, 1e-310, 2.2250738585072014e-308, 1e-17, 1.1102230246251565e-16, 2.220446049250313e-16, 0.41421356237309503, 0.41421356237309515, 1.0, 0.99999999999999989, 3.0, 1e154, 1.3407807929942596e154, 1e300, 1.7976931348623157e308, INFINITY, NAN};
    for (unsigned i = 0; i < sizeof sp / sizeof *sp; ++i) printf("a = %-24.17g mine %-24.17g libm %-24.17g  %.2f ulp\n", sp[i], log1p_pos(sp[i]), log1p(sp[i]), ulp_err(log1p_pos(sp[i]), log1p(sp[i])));
    return worst <= 2.5 ? 0 : 1;
}
