/* CPU study (not product, not oracle): could the FRONT of the exact state chain be speculated?  Inside a stretch where the true trajectory T does not
 * meet the cold-start trajectory S of a superblock, T = S + delta.  For every superblock the true trajectory leaves without having met S: what is delta
 * at its end (the carry offset the NEXT superblock would have to guess), how many distinct values cover most cases, and is delta at 1/2 and 3/4 of the
 * superblock a predictor of it?  Recipe = bench workload, one chr1-sized chain (as episodes.c).  Usage: front_spec B N SEED */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define R32(x) ((double)(float)(x))
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float gauss(uint64_t key) {
    const uint64_t a = mix64(key), b = mix64(key ^ 0xD1B54A32D192ED03ull);
    const float u1 = ((float)(a >> 40) + 1.0f) * (1.0f / 16777217.0f), u2 = (float)(b >> 40) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
}
typedef struct { double gs, zb; float p0, p1; } rec;
static rec *RC; static float *S, *T;
static inline void step(float *x0, float *x1, int64_t k) {
    const float xpf = *x0 + *x1; const double xp0 = xpf, x1d = *x1;
    const double dl = RC[k].gs * (RC[k].zb - xp0);
    *x0 = (float)(xp0 + (double)RC[k].p0 * dl); *x1 = (float)(x1d + (double)RC[k].p1 * dl);
}
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 24576;
    const int64_t N = argc > 2 ? atoll(argv[2]) : 1244783;
    const uint64_t SEED = argc > 3 ? (uint64_t)atoll(argv[3]) : 1234ull;
    const int m = 32;
    RC = malloc(sizeof(rec) * N); S = malloc(8 * N); T = malloc(8 * N);
    {
        uint64_t s = SEED * 0x9E3779B97F4A7C15ull + 12345; int64_t g = 0;
        double x = 0.0, p00 = 1000.0, p01 = 0.0, p11 = 1000.0;
        const double F01 = 1.0, Q00 = (double)1e-3f, Q11 = (double)1e-4f;
        for (int64_t k = 0; k < N; ++k, ++g) {
            double acc = 0.0;
            s ^= s << 13; s ^= s >> 7; s ^= s << 17; const uint64_t a = s;
            s ^= s << 13; s ^= s >> 7; s ^= s << 17; const uint64_t b = s;
            for (int q = 0; q < 6; ++q) acc += (double)((a >> (q * 10)) & 1023) / 1024.0;
            for (int q = 0; q < 6; ++q) acc += (double)((b >> (q * 10)) & 1023) / 1024.0;
            x += 0.03 * (acc - 6.0);
            const float lat = (float)x;
            double s0 = 0, s1z = 0;
            for (int j = 0; j < m; ++j) {
                const uint64_t key = SEED * 0x100000001B3ull + ((uint64_t)j << 40) + (uint64_t)g;
                const float z = lat + 0.5f * gauss(key * 2), v = 0.25f * expf(0.2f * gauss(key * 2 + 1));
                double r = (double)v; if (r < 1e-12) r = 1e-12;
                s0 += 1.0 / r; s1z += (double)z / r;
            }
            const double t00 = p00 + F01 * p01, t01 = p01 + F01 * p11;
            const double a00 = R32(t00 + t01 * F01 + Q00), a01 = R32(t01), a10 = R32(p01 + p11 * F01), a11 = R32(p11 + Q11);
            const double is = 1.0 + a00 * s0, gg = s0 / is, gH = s0 / (is * is);
            RC[g].gs = gg; RC[g].zb = s1z / s0; RC[g].p0 = (float)a00; RC[g].p1 = (float)a10;
            const double i00 = 1.0 - a00 * gg, i10 = -(a10 * gg);
            p00 = R32(i00 * i00 * a00 + gH * a00 * a00);
            p01 = R32(i00 * (i10 * a00 + a01) + gH * a00 * a10);
            p11 = R32((i10 * i10 * a00 + 2.0 * i10 * a10 + a11) + gH * a10 * a10);
        }
    }
    { float x0 = 0, x1 = 0; for (int64_t k = 0; k < N; ++k) { step(&x0, &x1, k); T[2*k] = x0; T[2*k+1] = x1; } }
    for (int64_t o = 0; o < N; o += B) { float x0 = 0, x1 = 0; for (int64_t k = o; k < N && k < o + B; ++k) { step(&x0, &x1, k); S[2*k] = x0; S[2*k+1] = x1; } }
    /* superblocks the true trajectory runs through without meeting S */
    typedef struct { int a, b; long n; } cell;
    cell cells[4096]; int ncell = 0;
    long through = 0, total = 0, sameHalf = 0, sameQ3 = 0, stretchNow = 0, stretchMax = 0;
    long stretchHist[8] = {0};
    for (int64_t o = B; o < N; o += B) {
        const int64_t end = o + B < N ? o + B : N;
        ++total;
        int merged = 0;
        for (int64_t k = o; k < end; ++k) if (T[2*k] == S[2*k] && T[2*k+1] == S[2*k+1]) { merged = 1; break; }
        if (merged) { if (stretchNow) stretchHist[stretchNow < 7 ? stretchNow : 7]++; stretchNow = 0; continue; }
        ++through; ++stretchNow; if (stretchNow > stretchMax) stretchMax = stretchNow;
        const int64_t kl = end - 1, kh = o + (end - o) / 2, kq = o + 3 * (end - o) / 4;
        /* delta in units of the ulp of S at that bin (level: ulp of the level; trend: ulp of the trend) */
        const float u0 = nextafterf(fabsf(S[2*kl]), 1e30f) - fabsf(S[2*kl]), u1 = nextafterf(fabsf(S[2*kl+1]), 1e30f) - fabsf(S[2*kl+1]);
        const int a = (int)lround((T[2*kl] - S[2*kl]) / u0), b = (int)lround((T[2*kl+1] - S[2*kl+1]) / u1);
        int f = -1;
        for (int i = 0; i < ncell; ++i) if (cells[i].a == a && cells[i].b == b) { f = i; break; }
        if (f < 0 && ncell < 4096) { f = ncell++; cells[f].a = a; cells[f].b = b; cells[f].n = 0; }
        if (f >= 0) cells[f].n++;
        /* is the offset (as floats) at 1/2 resp. 3/4 of the superblock the same as at its end? */
        if (T[2*kh] - S[2*kh] == T[2*kl] - S[2*kl] && T[2*kh+1] - S[2*kh+1] == T[2*kl+1] - S[2*kl+1]) ++sameHalf;
        if (T[2*kq] - S[2*kq] == T[2*kl] - S[2*kl] && T[2*kq+1] - S[2*kq+1] == T[2*kl+1] - S[2*kl+1]) ++sameQ3;
    }
    if (stretchNow) stretchHist[stretchNow < 7 ? stretchNow : 7]++;
    for (int i = 0; i < ncell; ++i) for (int j = i + 1; j < ncell; ++j) if (cells[j].n > cells[i].n) { cell t = cells[i]; cells[i] = cells[j]; cells[j] = t; }
    printf("B=%d N=%lld seed=%llu: %ld of %ld superblocks are run through without a meeting; longest stretch of consecutive ones %ld\n", B, (long long)N, (unsigned long long)SEED, through, total, stretchMax);
    printf("  stretches by length:"); for (int i = 1; i < 8; ++i) printf(" %d:%ld", i, stretchHist[i]); printf("\n");
    printf("  distinct (level, trend) offsets at the end, in ulps: %d; the likeliest:", ncell);
    { long cum = 0; for (int i = 0; i < ncell && i < 8; ++i) { cum += cells[i].n; printf("  (%d,%d) %.2f", cells[i].a, cells[i].b, (double)cells[i].n / through); } printf("   top-3 cover %.2f\n", through ? (double)(cells[0].n + (ncell > 1 ? cells[1].n : 0) + (ncell > 2 ? cells[2].n : 0)) / through : 0.0); (void)cum; }
    printf("  offset at 1/2 of the superblock == offset at its end: %.2f; at 3/4: %.2f\n", through ? (double)sameHalf / through : 0.0, through ? (double)sameQ3 / through : 0.0);
    return 0;
}
