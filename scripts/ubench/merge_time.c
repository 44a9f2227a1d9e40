// How long do two float32-rounded state trajectories (levelTrend filter, same gains, different start) take to coincide
// bit for bit?  Synthetic chain like the bench's: latent random walk, m samples, R ~ 0.25 exp(N(0,0.2^2)).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static double urand(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; }
static double nrand(void) { double a = 0; for (int i = 0; i < 12; ++i) a += urand(); return a - 6.0; }
static double r32(double x) { return (double)(float)x; }
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1244783, m = 32;
    const double Q00 = argc > 2 ? atof(argv[2]) : 1e-3, Q11 = argc > 3 ? atof(argv[3]) : 1e-4, F01 = 1.0, pad = 1e-4;
    double *S0 = malloc(sizeof(double) * n), *ZB = malloc(sizeof(double) * n);
    double x = 0;
    for (int k = 0; k < n; ++k) {
        x += 0.03 * nrand();
        double s0 = 0, s1 = 0;
        for (int j = 0; j < m; ++j) {
            const float z = (float)(x + 0.5 * nrand()), v = (float)(0.25 * exp(0.2 * nrand()));
            const double w = 1.0 / ((double)v + pad);
            s0 += w; s1 += w * (double)z;
        }
        S0[k] = s0; ZB[k] = s1 / s0;
    }
    // gains from the covariance recursion (float32-rounded carries)
    double *GS = malloc(sizeof(double) * n); float *P00 = malloc(sizeof(float) * n), *P10 = malloc(sizeof(float) * n);
    float c00 = 1000.f, c01 = 0.f, c11 = 1000.f;
    for (int k = 0; k < n; ++k) {
        const double t00 = fma(F01, c01, c00), t01 = fma(F01, c11, c01);
        const double a00 = r32(fma(t01, F01, t00 + Q00)), a01 = r32(t01), a10 = r32(fma(c11, F01, (double)c01)), a11 = r32(c11 + Q11);
        const double is = fma(a00, S0[k], 1.0), r = 1.0 / is, gG = S0[k] * r, gH = gG * r;
        const double i00 = fma(-a00, gG, 1.0), i10 = -(a10 * gG);
        c00 = (float)fma(gH, a00 * a00, i00 * i00 * a00);
        c01 = (float)fma(gH, a00 * a10, i00 * fma(i10, a00, a01));
        c11 = (float)fma(gH, a10 * a10, fma(i10 * i10, a00, fma(2.0 * i10, a10, a11)));
        GS[k] = gG; P00[k] = (float)a00; P10[k] = (float)a10;
    }
    // true state trajectory
    float *X0 = malloc(sizeof(float) * n), *X1 = malloc(sizeof(float) * n);
    float x0 = 0.f, x1 = 0.f;
    for (int k = 0; k < n; ++k) {
        const double xp0 = r32(fma(F01, (double)x1, (double)x0)), xp1 = x1;
        const double dl = GS[k] * (ZB[k] - xp0);
        x0 = (float)fma((double)P00[k], dl, xp0); x1 = (float)fma((double)P10[k], dl, xp1);
        X0[k] = x0; X1[k] = x1;
    }
    // cold starts every STRIDE bins: when does the trajectory coincide with the true one (both components, bitwise)?
    const int STRIDE = 4096, MAXW = 65536;
    int hist[20]; memset(hist, 0, sizeof hist); int never = 0, total = 0;
    long long sum = 0;
    for (int st = STRIDE; st + MAXW < n; st += STRIDE) {
        float y0 = (float)ZB[st], y1 = 0.f;       // a reasonable cold start: the local mean, no trend
        int merged = -1;
        for (int k = st; k < st + MAXW; ++k) {
            const double xp0 = r32(fma(F01, (double)y1, (double)y0)), xp1 = y1;
            const double dl = GS[k] * (ZB[k] - xp0);
            y0 = (float)fma((double)P00[k], dl, xp0); y1 = (float)fma((double)P10[k], dl, xp1);
            if (y0 == X0[k] && y1 == X1[k]) { merged = k - st + 1; break; }
        }
        ++total;
        if (merged < 0) { ++never; continue; }
        sum += merged;
        int b = 0; while ((1 << b) < merged) ++b;
        ++hist[b];
    }
    printf("starts %d, never merged within %d: %d, mean merge %.0f\n", total, MAXW, never, (double)sum / (total - never));
    int cum = 0;
    for (int b = 0; b < 18; ++b) { cum += hist[b]; if (hist[b]) printf("  <= %6d steps: %5d  (cum %.3f)\n", 1 << b, hist[b], (double)cum / total); }
    return 0;
}
