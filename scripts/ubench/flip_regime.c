// CPU study (round 6; not product, not oracle): where do two implementations of the reference's float32-rounded state recursion
// that agree to ~1e-16 per operation -- the reference's per-cell sum  s1 = sum_j (z_j - x) / R_j  and the sufficient-statistics
// form  s1 = S0 (zbar - x)  -- stop producing the same float32 levels?  Same covariance recursion for both, plain float64, no GPU.
//   gcc -O2 -ffp-contract=off -o flip_regime flip_regime.c -lm;  ./flip_regime N M WALK_SD LEVEL_OFFSET
// Measured (N = 1e6, M = 64, walk 0.03): level offset 60 / 100 / 130 / 200: 0 level values differ (a handful of trend values, one
// ulp); offset 264 (levels 264..298): 0.73 % of the level values differ, in 4060 episodes of 1-2 bins, 35 % of the trend values,
// by up to 2.6e-5; offset 500: 0.27 %; offset 1000: 0 again.  M = 8 at offset 264: 0.02 %.  The window is where one float32 ulp
// of the level (3e-5 in [256, 512)) is the size of the trend's own increments: a level that rounds the other way once kicks the
// trend by p10 gs ulp ~ 3e-6 -- 1e4 trend-ulps --, which makes the next level rounding differ with probability ~10 % per bin: the
// two trajectories keep each other apart at the ulp scale instead of merging.  Below the window a flipped trend merges back
// within ~1e2..1e4 bins (merge_time.c); above it the trend no longer reaches the level's last bit.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define R32(x) ((double)(float)(x))
static uint64_t s = 88172645463325252ull;
static double u01(void){ s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; }
static double gauss(void){ double a = u01(), b = u01(); if (a < 1e-300) a = 1e-300; return sqrt(-2.0 * log(a)) * cos(6.283185307179586 * b); }
int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 5000000; const int m = argc > 2 ? atoi(argv[2]) : 64;
    const double walk = argc > 3 ? atof(argv[3]) : 0.03, offset = argc > 4 ? atof(argv[4]) : 0.0;
    const double F01 = 1.0, Q00 = (float)1e-3, Q11 = (float)1e-4, pad = (float)1e-4;
    float *z = malloc(sizeof(float) * m), *v = malloc(sizeof(float) * m);
    double xr0 = 0, xr1 = 0, xs0 = 0, xs1 = 0, p00 = 1000, p01 = 0, p11 = 1000, lat = offset;
    int64_t differ = 0, flips = 0, trend_only = 0; int prev = 0; double maxlev = 0, maxdt = 0, sumdt = 0; int64_t big = 0;
    for (int64_t k = 0; k < n; ++k) {
        lat += walk * gauss();
        for (int j = 0; j < m; ++j) { z[j] = (float)(lat + 0.5 * gauss()); v[j] = (float)(0.25 * exp(0.2 * gauss())); }
        // covariance prediction (reference rounding points)
        const double t00 = p00 + F01 * p01, t01 = p01 + F01 * p11;
        const double a00 = R32(t00 + t01 * F01 + Q00), a01 = R32(t01), a10 = R32(p01 + p11 * F01), a11 = R32(p11 + Q11);
        // reference form
        double s0 = 0, s1 = 0;
        const double xrp0 = R32(xr0 + F01 * xr1), xrp1 = R32(xr1);
        for (int j = 0; j < m; ++j) { double r = (double)v[j] + pad; if (r < 1e-12) r = 1e-12; const double w = 1.0 / r; s1 += w * ((double)z[j] - xrp0); s0 += w; }
        const double is = 1.0 + a00 * s0, delta = s1 / is;
        xr0 = R32(xrp0 + a00 * delta); xr1 = R32(xrp1 + a10 * delta);
        // statistics form (pivot, fma like bin_stats)
        const double piv = (double)z[0];
        double S0 = 0, A = 0;
        for (int j = 0; j < m; ++j) { double r = (double)v[j] + pad; if (r < 1e-12) r = 1e-12; const double w = 1.0 / r; S0 += w; A = fma(w, (double)z[j] - piv, A); }
        const double zbar = piv + A / S0;
        const double gs = S0 / (1.0 + a00 * S0);
        const double xsp0 = R32(xs0 + F01 * xs1), xsp1 = R32(xs1);
        const double dl = gs * (zbar - xsp0);
        xs0 = R32(xsp0 + a00 * dl); xs1 = R32(xsp1 + a10 * dl);
        // covariance update
        const double g = s0 / is, gH = s0 / (is * is), i00 = 1.0 - a00 * g, i10 = -(a10 * g);
        p00 = R32(i00 * i00 * a00 + gH * a00 * a00);
        p01 = R32(i00 * (i10 * a00 + a01) + gH * a00 * a10);
        p11 = R32((i10 * i10 * a00 + 2.0 * i10 * a10 + a11) + gH * a10 * a10);
        const int d = (xr0 != xs0), dt = (xr1 != xs1);
        if (d) differ++;
        if (!d && dt) trend_only++;
        if (d && !prev) flips++;
        { const double e = fabs(xr1 - xs1); if (e > maxdt) maxdt = e; sumdt += e; if (e > 1e-7) big++; }
        prev = d;
        if (fabs(xr0) > maxlev) maxlev = fabs(xr0);
    }
    printf("max |trend diff| %.3g mean %.3g, bins with > 1e-7: %lld\n", maxdt, sumdt / n, (long long)big);
    printf("n %lld m %d walk %.3g offset %g: max |level| %.1f; level differs on %lld bins (%.3g), %lld episodes, trend-only differences on %lld bins\n",
           (long long)n, m, walk, offset, maxlev, (long long)differ, (double)differ / n, (long long)flips, (long long)trend_only);
    return 0;
}
