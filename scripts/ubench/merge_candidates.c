// Are several speculative candidates per block (different cold starts) independent draws?  For every block start s: K
// candidates started at s - W - c*STEP from (zbar, 0); candidate c is "true at s" if it equals the sequential trajectory
// bit for bit there.  Prints P(none of the first k is true) next to p^k.
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
    const int n = 1244783, m = 32;
    const double Q00 = argc > 1 ? atof(argv[1]) : 1e-3, Q11 = argc > 2 ? atof(argv[2]) : 1e-4, F01 = 1.0, pad = 1e-4;
    const int W = argc > 3 ? atoi(argv[3]) : 16384, STEP = argc > 4 ? atoi(argv[4]) : 512, MODE = argc > 5 ? atoi(argv[5]) : 0;
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
    float *X0 = malloc(sizeof(float) * n), *X1 = malloc(sizeof(float) * n);
    float x0 = 0.f, x1 = 0.f;
    for (int k = 0; k < n; ++k) {
        const double xp0 = r32(fma(F01, (double)x1, (double)x0)), xp1 = x1;
        const double dl = GS[k] * (ZB[k] - xp0);
        x0 = (float)fma((double)P00[k], dl, xp0); x1 = (float)fma((double)P10[k], dl, xp1);
        X0[k] = x0; X1[k] = x1;
    }
    enum { K = 8 };
    const int B = 8192;
    int none[K + 1]; memset(none, 0, sizeof none); int total = 0, single[K]; memset(single, 0, sizeof single);
    for (int st = 4 * B + K * STEP + W; st < n; st += B) {
        int ok[K];
        for (int c = 0; c < K; ++c) {
            const int from = MODE == 0 ? st - W - c * STEP : st - W;
            float y0 = (float)ZB[from], y1 = MODE == 0 ? 0.f : (float)(1e-3 * (c - K / 2));    // MODE 1: same start, different trend guess
            for (int k = from; k < st; ++k) {
                const double xp0 = r32(fma(F01, (double)y1, (double)y0)), xp1 = y1;
                const double dl = GS[k] * (ZB[k] - xp0);
                y0 = (float)fma((double)P00[k], dl, xp0); y1 = (float)fma((double)P10[k], dl, xp1);
            }
            ok[c] = (y0 == X0[st - 1] && y1 == X1[st - 1]);
            single[c] += !ok[c];
        }
        ++total;
        int any = 0;
        for (int k = 1; k <= K; ++k) { any |= ok[k - 1]; if (!any) ++none[k]; }
    }
    double p = 0; for (int c = 0; c < K; ++c) p += (double)single[c] / total; p /= K;
    printf("blocks %d, W %d, step %d, mode %d: single-candidate failure p = %.3f\n", total, W, STEP, MODE, p);
    for (int k = 1; k <= K; k *= 2) printf("  k = %d: P(no true candidate) = %.4f   p^k = %.4f\n", k, (double)none[k] / total, pow(p, k));
    return 0;
}
