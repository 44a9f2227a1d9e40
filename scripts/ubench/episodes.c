/* CPU study (not product, not oracle): structure of the bins where (true - old) trajectory offset changes inside re-run superblocks, and the cost of
 * round schemes.  Recipe = bench workload (as sb_async_sim.c). */
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
    const int J = argc > 3 ? atoi(argv[3]) : 4;
    const int m = 32;
    RC = malloc(sizeof(rec) * N); S = malloc(8 * N); T = malloc(8 * N);
    {
        uint64_t s = 1234ull * 0x9E3779B97F4A7C15ull + 12345; int64_t g = 0;
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
                const uint64_t key = 1234ull * 0x100000001B3ull + ((uint64_t)j << 40) + (uint64_t)g;
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
    /* per superblock: first repair run with the true carry against the cold-start trajectory */
    long batches = 0, events = 0, episodes = 0, roundsA = 0, roundsB = 0, walkB = 0, fbA = 0;
    long hist[16] = {0}; long d0hist[8] = {0};
    double maxLevel = 0;
    for (int64_t o = B; o < N; o += B) {
        const int64_t end = o + B < N ? o + B : N;
        int merged = 0;
        for (int64_t s = o; s < end && !merged; s += 64) {
            const int left = (int)(end - s < 64 ? end - s : 64);
            ++batches;
            /* event structure */
            int ev = 0, run = 0;
            for (int i = 0; i < left; ++i) {
                const int64_t k = s + i;
                const float d0 = T[2*k] - S[2*k], d1 = T[2*k+1] - S[2*k+1];
                const float q0 = T[2*k-2] - S[2*k-2], q1 = T[2*k-1] - S[2*k-1];
                if (fabs(T[2*k]) > maxLevel) maxLevel = fabs(T[2*k]);
                const int ch = d0 != q0 || d1 != q1;
                if (ch) { ++ev; ++run; } else { if (run) { ++episodes; hist[run < 15 ? run : 15]++; } run = 0; }
                float u = nextafterf(fabsf(S[2*k]), 1e30f) - fabsf(S[2*k]);
                int a = (int)lround(fabs(d0) / u); d0hist[a < 7 ? a : 7]++;
            }
            if (run) { ++episodes; hist[run < 15 ? run : 15]++; }
            events += ev;
            if (ev + 1 > 20) { roundsA += 20; ++fbA; } else roundsA += ev + 1;
            /* scheme B: faithful rounds with mini-walk J */
            {
                int pos = 0; float t0 = T[2*s-2], t1 = T[2*s-1];
                float d0 = t0 - S[2*s-2], d1 = t1 - S[2*s-1];
                while (pos < left) {
                    ++roundsB;
                    int f = left;
                    for (int i = pos; i < left; ++i) {
                        const int64_t k = s + i;
                        float q0 = i == pos ? t0 : S[2*k-2] + d0, q1 = i == pos ? t1 : S[2*k-1] + d1;
                        step(&q0, &q1, k);
                        const float c0 = S[2*k] + d0, c1 = S[2*k+1] + d1;
                        if (q0 != c0 || q1 != c1) { f = i; break; }
                    }
                    int hi = f < left ? f : left - 1;
                    if (f < left) { int w = J; while (w-- > 0 && hi < left - 1) { ++hi; ++walkB; } }
                    const int64_t k = s + hi;
                    t0 = T[2*k]; t1 = T[2*k+1]; d0 = t0 - S[2*k]; d1 = t1 - S[2*k+1];
                    pos = hi + 1;
                }
            }
            const int64_t kl = s + left - 1;
            if (T[2*kl] == S[2*kl] && T[2*kl+1] == S[2*kl+1]) merged = 1;
        }
    }
    printf("B=%d N=%lld J=%d max|level|=%.1f\n", B, (long long)N, J, maxLevel);
    printf("batches %ld, events/batch %.2f, episodes/batch %.2f, mean episode len %.2f\n", batches, (double)events / batches, (double)episodes / batches, (double)events / episodes);
    printf("episode length hist:"); for (int i = 1; i < 16; ++i) printf(" %d:%.3f", i, (double)hist[i] / episodes); printf("\n");
    printf("|d0| in ulps hist:"); { long t = 0; for (int i = 0; i < 8; ++i) t += d0hist[i]; for (int i = 0; i < 8; ++i) printf(" %d:%.4f", i, (double)d0hist[i] / t); } printf("\n");
    printf("scheme A (shipped): rounds/batch %.2f (fallback batches %.3f)\n", (double)roundsA / batches, (double)fbA / batches);
    printf("scheme B (mini-walk J=%d): rounds/batch %.2f, walk steps/batch %.2f\n", J, (double)roundsB / batches, (double)walkB / batches);
    for (int R = 120; R <= 280; R += 80) printf("  cost/batch cycles with ROUND=%d STEP=70: A %.0f  B %.0f\n", R, (double)roundsA / batches * R + (double)fbA / batches * 64 * 70, (double)roundsB / batches * R + (double)walkB / batches * 70);
    return 0;
}
