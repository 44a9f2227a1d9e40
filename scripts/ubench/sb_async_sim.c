/* CPU study (not product, not oracle): would the repair passes of the bit-exact state chain finish sooner WITHOUT the
 * barrier between passes?  Synchronous form (shipped): pass p re-runs every superblock whose carry-in differs from its
 * neighbour's carry-out; a pass costs its slowest superblock.  Asynchronous form: one resident wavefront per superblock
 * re-runs whenever its predecessor has published a new carry-out (optionally abandoning a run in flight), and the job is
 * over when the "final" mark has travelled down every chain.  Both are simulated on the bench workload's recipe with the same
 * cost model for a 64-bin batch in delta form: OVH + rounds * ROUND cycles, rounds = 1 + (bins where new - old trajectory
 * changes), or after 20 rounds a 64-step walk at STEP cycles per bin; a run stops where the new trajectory meets the old one.
 *   gcc -O2 -ffp-contract=off -o /tmp/sb_async_sim scripts/ubench/sb_async_sim.c -lm && /tmp/sb_async_sim [B] [scale] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define R32(x) ((double)(float)(x))
static const double OVH = 150.0, ROUND = 272.0, STEP = 46.0, GHZ = 2.0;

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
static int64_t N, NB; static int B;
static rec *RC; static float *cur, *truth;
static int64_t *chainOf, *chainStart;      /* per superblock: chain id, first bin */
static int64_t *sbOff; static int *sbLen; static char *sbFirst;

static inline void step(float *x0, float *x1, int64_t k) {
    const float xpf = *x0 + *x1; const double xp0 = xpf, x1d = *x1;
    const double dl = RC[k].gs * (RC[k].zb - xp0);
    *x0 = (float)(xp0 + (double)RC[k].p0 * dl); *x1 = (float)(x1d + (double)RC[k].p1 * dl);
}
/* one 64-bin batch of superblock b from true state (x0, x1): writes the new trajectory, returns its cost in cycles */
static double run_batch(int64_t s, int len, float *x0, float *x1, float *pd0, float *pd1) {
    int ev = 0;
    for (int i = 0; i < len; ++i) {
        step(x0, x1, s + i);
        const float d0 = *x0 - cur[2 * (s + i)], d1 = *x1 - cur[2 * (s + i) + 1];
        if (d0 != *pd0 || d1 != *pd1) ++ev;
        *pd0 = d0; *pd1 = d1;
        cur[2 * (s + i)] = *x0; cur[2 * (s + i) + 1] = *x1;
    }
    const int rounds = ev + 1;
    return rounds > 20 ? OVH + 20 * ROUND + 64 * STEP : OVH + rounds * ROUND;
}

typedef struct {
    float cin0, cin1, out0, out1; int ver, final, seen;      /* published carry-out + version */
    int running, t, runFinal, brk; float x0, x1, pd0, pd1; double clk; int runs, aborts;
} wave;


/* The same schedule WITH the speculative walks on the clock (a batch of a walk costs WOVH + 64 * STEP cycles), optionally with
 * the first superblock of every chain split into `split` pieces and with walks that are ABANDONED when the predecessor
 * publishes a final carry (the part walked so far is then re-run in delta form, the rest is walked from the true state).
 * Returns the time of the last final publication in microseconds. */
typedef struct {
    float cin0, cin1, out0, out1; int ver, final, seen;
    int walking, wt, running, t, runFinal, brk, sEnd, nb; float x0, x1, pd0, pd1; double clk; int runs, aborts;
} wave2;
/* readyUs (may be NULL): per chain, the time at which its statistics and covariance chain are done -- no superblock of the chain
 * starts its first walk earlier ("head per chain group").  chainEndUs (may be NULL): out, when each chain was final. */
static const double *g_readyUs = NULL; static double *g_chainEndUs = NULL;
static double timed(const int64_t *len, int split, int walkAbort, long *runsOut, long *abortsOut, int *okOut) {
    const double WOVH = 50.0, INF = 1e300, WAKE = 2.0 * GHZ * 1e3;
    int64_t nbk = 0;
    for (int c = 0; c < 22; ++c) { const int64_t first = len[c] < B ? len[c] : B; nbk += (len[c] > B ? (len[c] - B + B - 1) / B : 0) + ((first + B / split - 1) / (B / split)); }
    int64_t *off = malloc(sizeof(int64_t) * nbk); int *ln = malloc(sizeof(int) * nbk); char *fst = malloc(nbk);
    { int64_t b = 0, o = 0; const int Bs = B / split;
      for (int c = 0; c < 22; ++c) {
          const int64_t first = len[c] < B ? len[c] : B;
          for (int64_t k = 0; k < first; k += Bs, ++b) { off[b] = o + k; ln[b] = (int)(first - k < Bs ? first - k : Bs); fst[b] = k == 0; }
          for (int64_t k = first; k < len[c]; k += B, ++b) { off[b] = o + k; ln[b] = (int)(len[c] - k < B ? len[c] - k : B); fst[b] = 0; }
          o += len[c];
      }
      nbk = b; }
    wave2 *W = calloc(nbk, sizeof(wave2));
    long live = 0, runsAll = 0, abortsAll = 0;
    int *chainOfB = malloc(sizeof(int) * nbk);
    { int64_t b = 0; const int Bs = B / split;
      for (int c = 0; c < 22; ++c) {
          const int64_t first = len[c] < B ? len[c] : B;
          for (int64_t k = 0; k < first; k += Bs, ++b) chainOfB[b] = c;
          for (int64_t k = first; k < len[c]; k += B, ++b) chainOfB[b] = c;
      } }
    for (int64_t b = 0; b < nbk; ++b) {
        W[b].walking = 1; W[b].nb = (ln[b] + 63) / 64; ++live;
        W[b].clk = g_readyUs ? g_readyUs[chainOfB[b]] * GHZ * 1e3 : 0.0;
    }
    double total = 0;
    while (live) {
        int64_t b = -1; double best = INF;
        for (int64_t i = 0; i < nbk; ++i) if (W[i].clk < best) { best = W[i].clk; b = i; }
        if (b < 0) { printf("  deadlock\n"); break; }
        wave2 *w = &W[b], *p = fst[b] ? NULL : &W[b - 1];
        if (w->walking) {
            if (walkAbort && p && p->final && p->ver != w->seen) {      /* the true carry is here: stop walking from the cold prior */
                w->seen = p->ver; w->walking = 0; w->sEnd = w->wt; w->brk = 1 << 30;
                w->pd0 = p->out0 - 0.f; w->pd1 = p->out1 - 0.f;
                w->cin0 = p->out0; w->cin1 = p->out1; w->x0 = p->out0; w->x1 = p->out1; w->t = 0; w->running = 1; w->runFinal = 1; w->runs++; ++runsAll;
                continue;
            }
            const int l = ln[b] - w->wt * 64 < 64 ? ln[b] - w->wt * 64 : 64;
            const int64_t s = off[b] + (int64_t)w->wt * 64;
            for (int i = 0; i < l; ++i) { step(&w->x0, &w->x1, s + i); cur[2 * (s + i)] = w->x0; cur[2 * (s + i) + 1] = w->x1; }
            w->clk += WOVH + 64 * STEP; w->wt++;
            if (w->wt == w->nb) {
                w->walking = 0; w->out0 = w->x0; w->out1 = w->x1; w->ver = 1; w->final = fst[b]; w->sEnd = w->nb; w->brk = 0;
                if (w->final) { --live; if (w->clk > total) total = w->clk; if (g_chainEndUs && w->clk / GHZ / 1e3 > g_chainEndUs[chainOfB[b]]) g_chainEndUs[chainOfB[b]] = w->clk / GHZ / 1e3; w->clk = INF; }
                if (b + 1 < nbk && !fst[b + 1] && !W[b + 1].final && W[b + 1].clk == INF) W[b + 1].clk = best + WAKE;
            }
            continue;
        }
        const int fresh = p->ver != w->seen;
        if (!w->running || fresh) {
            if (!fresh) { w->clk = INF; continue; }
            w->seen = p->ver;
            const int same = p->out0 == w->cin0 && p->out1 == w->cin1;
            if (same) {
                if (w->running) w->runFinal |= p->final;
                else if (p->final) { w->final = 1; w->ver++; --live; if (w->clk > total) total = w->clk; if (g_chainEndUs && w->clk / GHZ / 1e3 > g_chainEndUs[chainOfB[b]]) g_chainEndUs[chainOfB[b]] = w->clk / GHZ / 1e3; w->clk = INF; if (b + 1 < nbk && !fst[b + 1] && W[b + 1].clk == INF && !W[b + 1].final) W[b + 1].clk = best + WAKE; continue; }
                else { w->clk = INF; continue; }
            } else {
                if (w->running) { if (w->t > w->sEnd) w->sEnd = w->t; if (w->brk != (1 << 30) && w->t * 64 > w->brk) w->brk = w->t * 64; w->aborts++; ++abortsAll; }
                w->pd0 = p->out0 - w->cin0; w->pd1 = p->out1 - w->cin1;
                w->cin0 = p->out0; w->cin1 = p->out1; w->x0 = p->out0; w->x1 = p->out1; w->t = 0; w->running = 1; w->runFinal = p->final; w->runs++; ++runsAll;
            }
        }
        const int l = ln[b] - w->t * 64 < 64 ? ln[b] - w->t * 64 : 64;
        const int64_t s = off[b] + (int64_t)w->t * 64;
        const float o0 = cur[2 * (s + l - 1)], o1 = cur[2 * (s + l - 1) + 1];
        if (w->t < w->sEnd) w->clk += run_batch(s, l, &w->x0, &w->x1, &w->pd0, &w->pd1);
        else { for (int i = 0; i < l; ++i) { step(&w->x0, &w->x1, s + i); cur[2 * (s + i)] = w->x0; cur[2 * (s + i) + 1] = w->x1; } w->clk += WOVH + 64 * STEP; }
        w->t++;
        const int end = w->t >= w->nb;
        const int merged = w->t <= w->sEnd && w->x0 == o0 && w->x1 == o1 && w->brk != (1 << 30) && w->t * 64 > w->brk;
        if (end || merged) {
            if (end) { w->out0 = w->x0; w->out1 = w->x1; w->brk = 0; w->sEnd = w->nb; }
            w->running = 0;
            if (p->ver != w->seen && p->out0 == w->cin0 && p->out1 == w->cin1) { w->seen = p->ver; w->runFinal |= p->final; }
            w->final = w->runFinal; w->ver++;
            if (w->final) { --live; if (w->clk > total) total = w->clk; if (g_chainEndUs && w->clk / GHZ / 1e3 > g_chainEndUs[chainOfB[b]]) g_chainEndUs[chainOfB[b]] = w->clk / GHZ / 1e3; }
            const double now = w->clk;
            if (w->final || p->ver == w->seen) w->clk = INF;
            if (b + 1 < nbk && !fst[b + 1] && !W[b + 1].final && W[b + 1].clk == INF) W[b + 1].clk = now + WAKE;
        }
    }
    int bad = 0; for (int64_t k = 0; k < 2 * N; ++k) if (cur[k] != truth[k]) { bad = 1; break; }
    *runsOut = runsAll; *abortsOut = abortsAll; *okOut = !bad;
    free(off); free(ln); free(fst); free(W); free(chainOfB);
    return total / GHZ / 1e3;
}

int main(int argc, char **argv) {
    B = argc > 1 ? atoi(argv[1]) : 24576;
    const double scale = argc > 2 ? atof(argv[2]) : 1.0;
    const int m = 32;
    static const int64_t hg38[22] = {1244783, 1210968, 991478, 951073, 907692, 854030, 796730, 725694, 691974, 668988, 675433,
                                     666377, 571822, 535219, 509957, 451692, 416288, 401887, 293088, 322221, 233550, 254093};
    int64_t len[22]; N = 0;
    for (int c = 0; c < 22; ++c) { len[c] = (int64_t)(hg38[c] * scale); N += len[c]; }
    RC = malloc(sizeof(rec) * N); cur = malloc(sizeof(float) * 2 * N); truth = malloc(sizeof(float) * 2 * N);
    /* records: the bench recipe (csr_batch_synthesize / k_synth), sufficient statistics, sequential covariance recursion */
    {
        uint64_t s = 1234ull * 0x9E3779B97F4A7C15ull + 12345; int64_t g = 0;
        for (int c = 0; c < 22; ++c) {
            double x = 0.0, p00 = 1000.0, p01 = 0.0, p11 = 1000.0;
            const double F01 = 1.0, Q00 = (double)1e-3f, Q11 = (double)1e-4f;
            for (int64_t k = 0; k < len[c]; ++k, ++g) {
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
    }
    NB = 0; for (int c = 0; c < 22; ++c) NB += (len[c] + B - 1) / B;
    sbOff = malloc(sizeof(int64_t) * NB); sbLen = malloc(sizeof(int) * NB); sbFirst = malloc(NB);
    { int64_t b = 0, off = 0; for (int c = 0; c < 22; ++c) { for (int64_t k = 0; k < len[c]; k += B, ++b) { sbOff[b] = off + k; sbLen[b] = (int)(len[c] - k < B ? len[c] - k : B); sbFirst[b] = k == 0; } off += len[c]; } }
    /* truth */
    { for (int64_t b = 0; b < NB; ++b) { static float x0, x1; if (sbFirst[b]) { x0 = 0.f; x1 = 0.f; } for (int i = 0; i < sbLen[b]; ++i) { step(&x0, &x1, sbOff[b] + i); truth[2 * (sbOff[b] + i)] = x0; truth[2 * (sbOff[b] + i) + 1] = x1; } } }
    wave *W = calloc(NB, sizeof(wave));
    for (int mode = 0; mode < 3; ++mode) {           /* 0 synchronous passes, 1 asynchronous, 2 asynchronous with abort */
        /* speculative pass: every superblock from the cold prior */
        for (int64_t b = 0; b < NB; ++b) {
            float x0 = 0.f, x1 = 0.f;
            for (int i = 0; i < sbLen[b]; ++i) { step(&x0, &x1, sbOff[b] + i); cur[2 * (sbOff[b] + i)] = x0; cur[2 * (sbOff[b] + i) + 1] = x1; }
            memset(&W[b], 0, sizeof(wave)); W[b].out0 = x0; W[b].out1 = x1; W[b].ver = 1; W[b].final = sbFirst[b];
        }
        double total = 0; long runsAll = 0, abortsAll = 0;
        if (mode == 0) {
            float *no0 = malloc(sizeof(float) * NB), *no1 = malloc(sizeof(float) * NB);
            for (int pass = 1;; ++pass) {
                double worst = 0, sum = 0; long reruns = 0;
                for (int64_t b = 0; b < NB; ++b) { no0[b] = W[b].out0; no1[b] = W[b].out1; }
                for (int64_t b = 0; b < NB; ++b) {
                    if (sbFirst[b]) continue;
                    if (W[b - 1].out0 == W[b].cin0 && W[b - 1].out1 == W[b].cin1) continue;
                    float x0 = W[b - 1].out0, x1 = W[b - 1].out1;
                    float pd0 = x0 - W[b].cin0, pd1 = x1 - W[b].cin1;      /* (old carry-in: the state before the block's old trajectory) */
                    W[b].cin0 = x0; W[b].cin1 = x1;
                    double cost = 0; int merged = 0;
                    for (int t = 0; t * 64 < sbLen[b]; ++t) {
                        const int l = sbLen[b] - t * 64 < 64 ? sbLen[b] - t * 64 : 64;
                        const int64_t s = sbOff[b] + (int64_t)t * 64;
                        const float o0 = cur[2 * (s + l - 1)], o1 = cur[2 * (s + l - 1) + 1];
                        cost += run_batch(s, l, &x0, &x1, &pd0, &pd1);
                        if (x0 == o0 && x1 == o1) { merged = 1; break; }
                    }
                    if (!merged) { no0[b] = x0; no1[b] = x1; }
                    if (cost > worst) worst = cost;
                    sum += cost; ++reruns;
                }
                for (int64_t b = 0; b < NB; ++b) { W[b].out0 = no0[b]; W[b].out1 = no1[b]; }
                if (!reruns) break;
                printf("  sync pass %d: %ld superblocks, slowest %.0f us, mean %.0f us\n", pass, reruns, worst / GHZ / 1e3, sum / reruns / GHZ / 1e3);
                total += worst / GHZ / 1e3 + 5.0; runsAll += reruns;
            }
            free(no0); free(no1);
        } else {
            /* event loop: always advance the wave with the smallest clock (idle waves sleep until their predecessor publishes) */
            const double INF = 1e300, WAKE = 2.0 * GHZ * 1e3;  /* 2 us from a publication to the neighbour seeing it */
            long live = 0;
            for (int64_t b = 0; b < NB; ++b) { W[b].clk = sbFirst[b] ? INF : 0.0; W[b].seen = 0; if (!sbFirst[b]) ++live; }
            /* old carry-in = cold prior */
            while (live) {
                int64_t b = -1; double best = INF;
                for (int64_t i = 0; i < NB; ++i) if (W[i].clk < best) { best = W[i].clk; b = i; }
                if (b < 0) { printf("  deadlock\n"); break; }
                wave *w = &W[b], *p = &W[b - 1];
                const int fresh = p->ver != w->seen;
                if (!w->running || (mode == 2 && fresh)) {
                    if (!fresh) { w->clk = INF; continue; }
                    w->seen = p->ver;
                    const int same = p->out0 == w->cin0 && p->out1 == w->cin1;
                    if (same) {
                        if (w->running) { w->runFinal |= p->final; }
                        else if (p->final) { w->final = 1; w->ver++; --live; w->clk = INF; if (b + 1 < NB && !sbFirst[b + 1] && W[b + 1].clk == INF && !W[b + 1].final) W[b + 1].clk = best + WAKE; continue; }
                        else { w->clk = INF; continue; }
                    } else {
                        if (w->running) { if (w->t * 64 > w->brk) w->brk = w->t * 64; w->aborts++; ++abortsAll; }
                        w->pd0 = p->out0 - w->cin0; w->pd1 = p->out1 - w->cin1;
                        w->cin0 = p->out0; w->cin1 = p->out1; w->x0 = p->out0; w->x1 = p->out1; w->t = 0; w->running = 1; w->runFinal = p->final; w->runs++; ++runsAll;
                    }
                }
                /* one batch */
                const int l = sbLen[b] - w->t * 64 < 64 ? sbLen[b] - w->t * 64 : 64;
                const int64_t s = sbOff[b] + (int64_t)w->t * 64;
                const float o0 = cur[2 * (s + l - 1)], o1 = cur[2 * (s + l - 1) + 1];
                w->clk += run_batch(s, l, &w->x0, &w->x1, &w->pd0, &w->pd1);
                w->t++;
                const int end = w->t * 64 >= sbLen[b];
                const int merged = w->x0 == o0 && w->x1 == o1 && w->t * 64 > w->brk;   /* the compared bin lies in the last continuous piece */
                if (end || merged) {
                    if (end) { w->out0 = w->x0; w->out1 = w->x1; w->brk = 0; }
                    w->running = 0;
                    /* finality is re-read at the end: the predecessor may have been marked final (same carry) meanwhile */
                    if (p->ver != w->seen && p->out0 == w->cin0 && p->out1 == w->cin1) { w->seen = p->ver; w->runFinal |= p->final; }
                    w->final = w->runFinal; w->ver++;
                    if (w->final) { --live; if (w->clk > total) total = w->clk; }
                    const double now = w->clk;
                    if (w->final || p->ver == w->seen) w->clk = INF;
                    if (b + 1 < NB && !sbFirst[b + 1] && !W[b + 1].final && W[b + 1].clk == INF) W[b + 1].clk = now + WAKE;
                }
            }
            total = total / GHZ / 1e3;
        }
        int bad = 0; for (int64_t k = 0; k < 2 * N; ++k) if (cur[k] != truth[k]) { bad = 1; break; }
        long ab = 0; for (int64_t b = 0; b < NB; ++b) ab += W[b].aborts;
        printf("%s: %.0f us, %ld superblock runs, %ld aborted; result %s\n", mode == 0 ? "synchronous passes" : mode == 1 ? "asynchronous" : "asynchronous + abort",
               total, runsAll, ab, bad ? "WRONG" : "== sequential");
    }
    {
        static const int cfg[][2] = {{1, 0}, {1, 1}, {6, 0}, {6, 1}, {12, 1}};
        for (unsigned k = 0; k < sizeof cfg / sizeof cfg[0]; ++k) {
            long r, a; int ok;
            const double us = timed(len, cfg[k][0], cfg[k][1], &r, &a, &ok);
            printf("walks on the clock, first superblock of a chain in %2d piece(s), walks %s: %.0f us, %ld runs, %ld abandoned; result %s\n",
                   cfg[k][0], cfg[k][1] ? "abandoned for a final carry" : "always finished          ", us, r, a, ok ? "== sequential" : "WRONG");
        }
    }
    /* ---- round 5: the head of a step PER GROUP OF CHAINS.  Today every chain waits for the statistics and the covariance chain of
     * ALL chains (HEAD us, measured) before the single launch of the state chain starts.  With a ready word per chain that gates a
     * superblock's first walk, the head kernels could be launched per group of chains on a feeder stream; chain c is then ready when
     * the heads of the groups up to its own are done: head(g) = FIX + HEAD * bins(g) / bins (the two kernels are bandwidth-bound).
     * Simulated: G groups of about equal bins, chains assigned in the given ORDER; the state chain's cost model is scaled so that
     * the ungrouped launch lasts what the GPU measures (2.0 ms); SLOW = how much a walking / repairing wavefront slows down while
     * head kernels of later groups share the chip with it (the statistics kernel fills every CU). */
    {
        const double HEAD = 1090.0, FIX = 25.0, TAIL = 790.0, MEASURED = 2000.0;
        double fin0[22]; for (int c = 0; c < 22; ++c) fin0[c] = 0.0;
        long r, a; int ok;
        g_readyUs = NULL; g_chainEndUs = fin0;
        const double base = timed(len, 1, 1, &r, &a, &ok);
        const double k = MEASURED / base;        /* cost-model scale */
        printf("\nhead per chain group (cost model scaled by %.2f so that the single launch lasts %.0f us like on the GPU)\n", k, MEASURED);
        printf("chain finish times of the ungrouped launch (us, scaled):"); for (int c = 0; c < 22; ++c) printf(" %.0f", fin0[c] * k); printf("\n");
        printf("today: head %.0f + state chain %.0f + tail %.0f = %.0f us per step\n", HEAD, MEASURED, TAIL, HEAD + MEASURED + TAIL);
        /* per-chain mean |zbar| (what a cheap pre-pass over the data could know before anything else has run) */
        double lvl[22]; { int64_t g = 0; for (int c = 0; c < 22; ++c) { double sa = 0; for (int64_t q = 0; q < len[c]; ++q, ++g) sa += fabs(RC[g].zb); lvl[c] = sa / (double)len[c]; } }
        for (int order = 0; order < 4; ++order) {
            int idx[22]; for (int c = 0; c < 22; ++c) idx[c] = c;
            double key[22];
            for (int c = 0; c < 22; ++c) key[c] = order == 0 ? (double)len[c] : order == 1 ? fin0[c] : order == 2 ? lvl[c] * sqrt((double)len[c]) : -(double)c;
            for (int i = 0; i < 22; ++i) for (int j = i + 1; j < 22; ++j) if (key[idx[j]] > key[idx[i]]) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
            const char *name = order == 0 ? "longest chain first" : order == 1 ? "slowest chain of the PREVIOUS identical step first (oracle)" : order == 2 ? "mean |level| x sqrt(length) first (predictor)" : "genome order";
            for (int G = 2; G <= 6; G += (G == 2 ? 1 : (G == 3 ? 1 : 2))) {
                for (int sl = 0; sl < 3; ++sl) {
                    const double SLOW = sl == 0 ? 1.0 : sl == 1 ? 1.15 : 1.3;
                    double ready[22], endc[22]; int grp[22];
                    { double acc = 0; int g = 0; for (int i = 0; i < 22; ++i) { grp[idx[i]] = g; acc += (double)len[idx[i]]; if (acc >= (double)N * (g + 1) / G && g < G - 1) ++g; } }
                    double tg[8], t = 0; for (int g = 0; g < G; ++g) { double bins = 0; for (int c = 0; c < 22; ++c) if (grp[c] == g) bins += (double)len[c]; t += FIX + HEAD * bins / (double)N; tg[g] = t; }
                    /* the simulator runs in unscaled cycles: hand it ready times divided by k * SLOW-on-average.  A chain's work that
                     * overlaps later heads runs SLOW times slower: approximated by stretching the whole state chain by the share of
                     * its run that lies before the last head ends (iterated once) */
                    for (int c = 0; c < 22; ++c) { ready[c] = tg[grp[c]] / k; endc[c] = 0; }
                    g_readyUs = ready; g_chainEndUs = endc;
                    (void)timed(len, 1, 1, &r, &a, &ok);
                    double stepEnd = 0;
                    for (int c = 0; c < 22; ++c) {
                        const double start = tg[grp[c]], dur = endc[c] * k - start;              /* this chain's state chain, undisturbed */
                        const double overlap = tg[G - 1] > start ? (tg[G - 1] - start < dur * SLOW ? tg[G - 1] - start : dur * SLOW) : 0.0;
                        const double fin = start + dur + overlap * (SLOW - 1.0) / SLOW;
                        if (fin > stepEnd) stepEnd = fin;
                    }
                    printf("  %-62s G=%d slow x%.2f: state chains end at %.0f us -> step %.0f us (%+.1f %%)%s\n", name, G, SLOW, stepEnd, stepEnd + TAIL,
                           100.0 * (stepEnd + TAIL - (HEAD + MEASURED + TAIL)) / (HEAD + MEASURED + TAIL), ok ? "" : "  WRONG RESULT");
                }
            }
        }
        g_readyUs = NULL; g_chainEndUs = NULL;
    }
    return 0;
}
