/* tolerant_emul.c -- CPU study tool (not product, not oracle): the speculative blocked recurrences of the tolerant
 * validation mode restated on the host, next to the sequential recursion, to measure what an acceptance rule lets
 * through.  Arithmetic = the reference's (fp64 on float32-rounded carries, pyx:388-527, 6758-6848, 8244-8298).
 * One chain, levelTrend, kappa re-weighting only (the reference's default ECM).  Build: scripts/tolerant_emul.py.
 *
 * Blocks of B bins; a block's forward walk starts Wf bins early from the cold prior, its smoother walk Wb bins late
 * from the filtered moments; rule selects the acceptance test of a block's carry-in against its neighbour's carry-out
 * (a failing block is re-run from that carry-out, in chain order -- the fixed point of the device's repair passes).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define R32(x) ((double)(float)(x))

typedef struct {
    int64_t m, n;
    const float *data, *munc;
    double F01, Q00, Q11, pad, kmin, kmax, nu, init, cinit;
} prob;

typedef struct { float x0, x1, p00, p01, p11; } fcarry;
typedef struct { float x0, x1, p00, p01, p10, p11; int fresh; } bcarry;

static inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline float ulp_of(float mag) {
    union { float f; uint32_t u; } a; a.f = mag; a.u &= 0x7f800000u; return a.f * 1.1920929e-07f;
}

/* per-bin data-only statistics (computed once) */
typedef struct { double s0, s1z, s2z; } bstat;   /* sum w, sum w z, sum w z^2 about 0 is cancellation-prone: keep cells */

static void fwd_step(const prob *P, const float *kappa, int64_t k, fcarry *c, float *xf, float *Pf, float *pn, double *sens) {
    const double kap = kappa ? clampd((double)kappa[k], P->kmin, P->kmax) : 1.0;
    double x0 = c->x0, x1 = c->x1, p00 = c->p00, p01 = c->p01, p10 = c->p01, p11 = c->p11;
    const double xp0 = x0 + P->F01 * x1, xp1 = x1;
    x0 = R32(xp0); x1 = R32(xp1);
    const double Q00 = (1.0 / kap) * P->Q00, Q11 = (1.0 / kap) * P->Q11;
    const double t00 = p00 + P->F01 * p10, t01 = p01 + P->F01 * p11, t10 = p10, t11 = p11;
    p00 = R32(t00 + t01 * P->F01 + Q00);
    p01 = R32(t01 + 0.0);
    p10 = R32(t10 + t11 * P->F01 + 0.0);
    p11 = R32(t11 + Q11);
    double s0 = 0, s1 = 0;
    for (int64_t j = 0; j < P->m; ++j) {
        const int64_t idx = j * P->n + k;
        const double innov = (double)P->data[idx] - x0;
        double r = (double)P->munc[idx] + P->pad;
        if (r < 1e-12) r = 1e-12;
        const double w = 1.0 / r;
        s1 += w * innov; s0 += w;
    }
    const double iscale = 1.0 + p00 * s0;
    const double delta = s1 / iscale;
    const double gG = s0 / iscale, gH = s0 / (iscale * iscale);
    /* sensitivity of the (real-arithmetic) state recursion to its carry: e' = (I - K H) F e, K = [p00 gG, p10 gG] */
    if (sens) {
        /* sens = 2x2 matrix M (row-major) with e_k = M e_in ; update M <- A M, A = (I-KH)F, F=[[1,f],[0,1]], H=[1,0] */
        const double k0 = p00 * gG, k1 = p10 * gG;
        const double a00 = 1.0 - k0, a01 = (1.0 - k0) * P->F01, a10 = -k1, a11 = 1.0 - k1 * P->F01;
        const double m00 = a00 * sens[0] + a01 * sens[2], m01 = a00 * sens[1] + a01 * sens[3];
        const double m10 = a10 * sens[0] + a11 * sens[2], m11 = a10 * sens[1] + a11 * sens[3];
        sens[0] = m00; sens[1] = m01; sens[2] = m10; sens[3] = m11;
    }
    x0 = R32(x0 + p00 * delta);
    x1 = R32(x1 + p10 * delta);
    const double i00 = 1.0 - (p00 * gG), i10 = -(p10 * gG);
    const double n00 = (i00 * i00 * p00) + (gH * (p00 * p00));
    const double n01 = (i00 * (i10 * p00 + p01)) + (gH * (p00 * p10));
    const double n11 = ((i10 * i10 * p00) + 2.0 * i10 * p10 + p11) + (gH * (p10 * p10));
    c->x0 = (float)x0; c->x1 = (float)x1; c->p00 = (float)n00; c->p01 = (float)n01; c->p11 = (float)n11;
    if (xf) {
        xf[k * 2] = c->x0; xf[k * 2 + 1] = c->x1;
        Pf[k * 4] = c->p00; Pf[k * 4 + 1] = c->p01; Pf[k * 4 + 2] = c->p01; Pf[k * 4 + 3] = c->p11;
        if (k > 0) { pn[(k - 1) * 4] = (float)Q00; pn[(k - 1) * 4 + 1] = 0.f; pn[(k - 1) * 4 + 2] = 0.f; pn[(k - 1) * 4 + 3] = (float)Q11; }
    }
}

/* smoother step at bin k (k < n-1 uses carry = moments of k+1); sens: 2x2 map of the state carry (e_k = J e_{k+1}) */
static void bwd_step(const prob *P, int64_t k, const float *xf, const float *Pf, const float *pn, bcarry *c,
                     float *xs, float *Ps, float *lag, double *sens) {
    if (c->fresh) {
        c->x0 = xf[k * 2]; c->x1 = xf[k * 2 + 1];
        c->p00 = Pf[k * 4]; c->p01 = Pf[k * 4 + 1]; c->p10 = Pf[k * 4 + 2]; c->p11 = Pf[k * 4 + 3];
        c->fresh = 0;
    } else {
        const double F01 = P->F01;
        const double f00 = Pf[k * 4], f01 = Pf[k * 4 + 1], f10 = Pf[k * 4 + 2], f11 = Pf[k * 4 + 3];
        const double a0 = xf[k * 2], a1 = xf[k * 2 + 1];
        const double xp0 = a0 + F01 * a1, xp1 = a1;
        const double Q00 = pn[k * 4], Q01 = pn[k * 4 + 1], Q10 = pn[k * 4 + 2], Q11 = pn[k * 4 + 3];
        double c00 = f00 + F01 * f10, c01 = f01 + F01 * f11, c10 = f10, c11 = f11;
        const double pp00 = c00 + c01 * F01 + Q00, pp01 = c01 + Q01, pp10 = c10 + c11 * F01 + Q10, pp11 = c11 + Q11;
        const double det = (pp00 * pp11) - (pp01 * pp10);
        const double v00 = pp11 / det, v01 = -pp01 / det, v10 = -pp10 / det, v11 = pp00 / det;
        c00 = f00 + f01 * F01; c01 = f01; c10 = f10 + f11 * F01; c11 = f11;
        const double J00 = c00 * v00 + c01 * v10, J01 = c00 * v01 + c01 * v11;
        const double J10 = c10 * v00 + c11 * v10, J11 = c10 * v01 + c11 * v11;
        if (sens) {
            const double m00 = J00 * sens[0] + J01 * sens[2], m01 = J00 * sens[1] + J01 * sens[3];
            const double m10 = J10 * sens[0] + J11 * sens[2], m11 = J10 * sens[1] + J11 * sens[3];
            sens[0] = m00; sens[1] = m01; sens[2] = m10; sens[3] = m11;
        }
        const double dx0 = (double)c->x0 - xp0, dx1 = (double)c->x1 - xp1;
        const double d00 = (double)c->p00 - pp00, d01 = (double)c->p01 - pp01, d10 = (double)c->p10 - pp10, d11 = (double)c->p11 - pp11;
        const double r00 = d00 * J00 + d01 * J01, r01 = d00 * J10 + d01 * J11;
        const double r10 = d10 * J00 + d11 * J01, r11 = d10 * J10 + d11 * J11;
        if (lag) {
            lag[k * 4] = (float)(c00 + (J00 * d00 + J01 * d10));
            lag[k * 4 + 1] = (float)(c01 + (J00 * d01 + J01 * d11));
            lag[k * 4 + 2] = (float)(c10 + (J10 * d00 + J11 * d10));
            lag[k * 4 + 3] = (float)(c11 + (J10 * d01 + J11 * d11));
        }
        c->x0 = (float)(a0 + (J00 * dx0 + J01 * dx1));
        c->x1 = (float)(a1 + (J10 * dx0 + J11 * dx1));
        c->p00 = (float)(f00 + (J00 * r00 + J01 * r10));
        c->p01 = (float)(f01 + (J00 * r01 + J01 * r11));
        c->p10 = c->p01;
        c->p11 = (float)(f11 + (J10 * r01 + J11 * r11));
    }
    if (xs) {
        xs[k * 2] = c->x0; xs[k * 2 + 1] = c->x1;
        Ps[k * 4] = c->p00; Ps[k * 4 + 1] = c->p01; Ps[k * 4 + 2] = c->p10; Ps[k * 4 + 3] = c->p11;
    }
}

static void estep_kappa(const prob *P, const float *xs, const float *Ps, const float *lag, float *kappa) {
    const double f00 = 1, f01 = P->F01, f10 = 0, f11 = 1;
    const double det = P->Q00 * P->Q11;
    const double qi00 = P->Q11 / det, qi11 = P->Q00 / det;
    kappa[0] = 1.0f;
    for (int64_t k = 0; k < P->n - 1; ++k) {
        const double x0 = xs[k * 2], x1 = xs[k * 2 + 1], y0 = xs[(k + 1) * 2], y1 = xs[(k + 1) * 2 + 1];
        const double xx00 = (double)Ps[k * 4] + x0 * x0, xx01 = (double)Ps[k * 4 + 1] + x0 * x1;
        const double xx10 = (double)Ps[k * 4 + 2] + x1 * x0, xx11 = (double)Ps[k * 4 + 3] + x1 * x1;
        const double yy00 = (double)Ps[(k + 1) * 4] + y0 * y0, yy01 = (double)Ps[(k + 1) * 4 + 1] + y0 * y1;
        const double yy10 = (double)Ps[(k + 1) * 4 + 2] + y1 * y0, yy11 = (double)Ps[(k + 1) * 4 + 3] + y1 * y1;
        const double xy00 = (double)lag[k * 4] + x0 * y0, xy01 = (double)lag[k * 4 + 1] + x0 * y1;
        const double xy10 = (double)lag[k * 4 + 2] + x1 * y0, xy11 = (double)lag[k * 4 + 3] + x1 * y1;
        const double yx00 = xy00, yx01 = xy10, yx10 = xy01, yx11 = xy11;
        const double t00 = f00, t01 = f10, t10 = f01, t11 = f11;
        double w00 = yy00 - (yx00 * t00 + yx01 * t10), w01 = yy01 - (yx00 * t01 + yx01 * t11);
        double w10 = yy10 - (yx10 * t00 + yx11 * t10), w11 = yy11 - (yx10 * t01 + yx11 * t11);
        w00 -= (f00 * xy00 + f01 * xy10); w01 -= (f00 * xy01 + f01 * xy11);
        w10 -= (f10 * xy00 + f11 * xy10); w11 -= (f10 * xy01 + f11 * xy11);
        const double g00 = f00 * xx00 + f01 * xx10, g01 = f00 * xx01 + f01 * xx11;
        const double g10 = f10 * xx00 + f11 * xx10, g11 = f10 * xx01 + f11 * xx11;
        w00 += (g00 * t00 + g01 * t10); w01 += (g00 * t01 + g01 * t11);
        w10 += (g10 * t00 + g11 * t10); w11 += (g10 * t01 + g11 * t11);
        if (w00 < 0) w00 = 0;
        if (w11 < 0) w11 = 0;
        (void)w01; (void)w10;
        double delta = qi00 * w00 + qi11 * w11;
        if (delta < 0) delta = 0;
        double kap = (P->nu + 2.0) / (P->nu + delta);
        kap = clampd(kap, P->kmin, P->kmax);
        kappa[k + 1] = (float)kap;
    }
}

/* acceptance rules */
typedef struct { int rule; int k; double budget; } rulecfg;
static int near_ulps(float a, float b, float scale, int k) {
    return fabsf(a - b) <= (float)k * ulp_of(fmaxf(fmaxf(fabsf(a), fabsf(b)), scale));
}
static int same_P(const rulecfg *R, float a00, float a01, float a11, float b00, float b01, float b11) {
    if (a00 == b00 && a01 == b01 && a11 == b11) return 1;
    if (R->k <= 0 || R->rule == 3) return 0;        /* rule 3: the covariance carry must be bit-equal */
    return near_ulps(a00, b00, 0.f, R->k) && near_ulps(a11, b11, 0.f, R->k) && near_ulps(a01, b01, sqrtf(fabsf(a00 * a11)), R->k);
}
/* amp0, amp1: max over the block of |d level_j / d x0_in|, |d level_j / d x1_in| (rule 2) */
static int same_X(const rulecfg *R, const prob *P, float ax0, float ax1, float bx0, float bx1, double amp0, double amp1) {
    if (ax0 == bx0 && ax1 == bx1) return 1;
    if (R->rule == 0) return 1;
    if (R->k <= 0) return 0;
    if (R->rule == 3) {
        /* rule 3 (round 6): the LEVEL bit-equal, the trend within k of ITS OWN ulps (the trend is 3-4 orders below the level:
           k trend-ulps are ~1e-3 level-ulps) */
        if (ax0 != bx0) return 0;
        return fabsf(ax1 - bx1) <= (float)R->k * ulp_of(fmaxf(fabsf(ax1), fabsf(bx1)));
    }
    const float mag = fmaxf(fmaxf(fabsf(ax0), fabsf(bx0)), 1.0f);
    const float ulp = ulp_of(mag);
    if (R->rule == 1) {
        const float lim = (float)R->k * ulp;
        return fabsf(ax0 - bx0) <= lim && (float)fabs(P->F01) * fabsf(ax1 - bx1) <= lim;
    }
    /* rule 2: bound the integrated effect on the level over the block */
    const double eff = amp0 * fabs((double)ax0 - bx0) + amp1 * fabs((double)ax1 - bx1);
    return eff <= R->budget * (double)ulp;
}

/* first_f / first_b: blocks whose carry-in fails against the neighbour's SPECULATIVE carry-out (what a first validation pass
   sees); chain_f / chain_b: longest run of consecutive re-run blocks of the in-order repair (the chain reaction) */
typedef struct { int64_t rerun_f, rerun_b, blocks; double max_amp_f, max_amp_b; int64_t first_f, first_b, chain_f, chain_b; } estats;

static void forward_blocked(const prob *P, const float *kappa, int B, int W, const rulecfg *R, float *xf, float *Pf,
                            float *pn, estats *st) {
    const int64_t n = P->n, NB = (n + B - 1) / B;
    fcarry *cin = malloc(sizeof(fcarry) * NB), *cout = malloc(sizeof(fcarry) * NB);
    double *amp = calloc(2 * NB, sizeof(double));
    for (int64_t b = 0; b < NB; ++b) {
        const int64_t s = b * B, e = s + B < n ? s + B : n;
        int64_t w0 = s - W; if (w0 < 0) w0 = 0;
        fcarry c = {(float)P->init, 0.f, (float)P->cinit, 0.f, (float)P->cinit};
        for (int64_t k = w0; k < s; ++k) fwd_step(P, kappa, k, &c, NULL, NULL, NULL, NULL);
        cin[b] = c;
        double M[4] = {1, 0, 0, 1}, a0 = 1, a1 = 0;
        for (int64_t k = s; k < e; ++k) {
            fwd_step(P, kappa, k, &c, xf, Pf, pn, M);
            if (fabs(M[0]) > a0) a0 = fabs(M[0]);
            if (fabs(M[1]) > a1) a1 = fabs(M[1]);
        }
        amp[2 * b] = a0; amp[2 * b + 1] = a1;
        if (a1 > st->max_amp_f) st->max_amp_f = a1;
        cout[b] = c;
    }
    if (R->rule >= 0) {
        for (int64_t b = 1; b < NB; ++b) {
            const fcarry a = cout[b - 1], q = cin[b];
            if (!(same_P(R, a.p00, a.p01, a.p11, q.p00, q.p01, q.p11) &&
                  same_X(R, P, a.x0, a.x1, q.x0, q.x1, amp[2 * b], amp[2 * b + 1]))) st->first_f++;
        }
        int64_t run = 0;
        for (int64_t b = 1; b < NB; ++b) {
            const fcarry a = cout[b - 1], q = cin[b];
            const int ok = same_P(R, a.p00, a.p01, a.p11, q.p00, q.p01, q.p11) &&
                           same_X(R, P, a.x0, a.x1, q.x0, q.x1, amp[2 * b], amp[2 * b + 1]);
            if (ok) run = 0; else { run++; if (run > st->chain_f) st->chain_f = run; }
            if (!ok) {
                fcarry c = a;
                const int64_t s = b * B, e = s + B < n ? s + B : n;
                for (int64_t k = s; k < e; ++k) fwd_step(P, kappa, k, &c, xf, Pf, pn, NULL);
                cout[b] = c;
                st->rerun_f++;
            }
        }
    }
    st->blocks += NB;
    free(cin); free(cout); free(amp);
}

static void backward_blocked(const prob *P, int B, int W, const rulecfg *R, const float *xf, const float *Pf, const float *pn,
                             float *xs, float *Ps, float *lag, estats *st) {
    const int64_t n = P->n, NB = (n + B - 1) / B;
    bcarry *cin = malloc(sizeof(bcarry) * NB), *cout = malloc(sizeof(bcarry) * NB);
    double *amp = calloc(2 * NB, sizeof(double));
    for (int64_t b = NB - 1; b >= 0; --b) {
        const int64_t s = b * B, e = s + B < n ? s + B : n;
        int64_t w1 = e + W; if (w1 > n) w1 = n;
        bcarry c; memset(&c, 0, sizeof c); c.fresh = 1;
        for (int64_t k = w1 - 1; k >= e; --k) bwd_step(P, k, xf, Pf, pn, &c, NULL, NULL, NULL, NULL);
        cin[b] = c;
        double M[4] = {1, 0, 0, 1}, a0 = 1, a1 = 0;
        for (int64_t k = e - 1; k >= s; --k) {
            bwd_step(P, k, xf, Pf, pn, &c, xs, Ps, lag, M);
            if (fabs(M[0]) > a0) a0 = fabs(M[0]);
            if (fabs(M[1]) > a1) a1 = fabs(M[1]);
        }
        amp[2 * b] = a0; amp[2 * b + 1] = a1;
        if (a1 > st->max_amp_b) st->max_amp_b = a1;
        cout[b] = c;
    }
    if (R->rule >= 0) {
        for (int64_t b = NB - 2; b >= 0; --b) {
            const bcarry a = cout[b + 1], q = cin[b];
            if (!(same_P(R, a.p00, a.p01, a.p11, q.p00, q.p01, q.p11) && (a.p10 == q.p10 || (R->k > 0 && R->rule != 3)) &&
                  same_X(R, P, a.x0, a.x1, q.x0, q.x1, amp[2 * b], amp[2 * b + 1]))) st->first_b++;
        }
        int64_t run = 0;
        for (int64_t b = NB - 2; b >= 0; --b) {
            const bcarry a = cout[b + 1], q = cin[b];
            const int ok = same_P(R, a.p00, a.p01, a.p11, q.p00, q.p01, q.p11) && (a.p10 == q.p10 || (R->k > 0 && R->rule != 3)) &&
                           same_X(R, P, a.x0, a.x1, q.x0, q.x1, amp[2 * b], amp[2 * b + 1]);
            if (ok) run = 0; else { run++; if (run > st->chain_b) st->chain_b = run; }
            if (!ok) {
                bcarry c = a;
                const int64_t s = b * B, e = s + B < n ? s + B : n;
                for (int64_t k = e - 1; k >= s; --k) bwd_step(P, k, xf, Pf, pn, &c, xs, Ps, lag, NULL);
                cout[b] = c;
                st->rerun_b++;
            }
        }
    }
    free(cin); free(cout); free(amp);
}

/* rule < 0 with B >= n: sequential.  Outputs after `iters` ECM iterations of `inner` sweeps: xs (n,2), kappa (n).
 * xs_iter: optional (iters, n, 2) smoothed states after each iteration's last sweep. */
int emul_ecm(int64_t m, int64_t n, const float *data, const float *munc, double F01, double Q00, double Q11, double pad,
             double kmin, double kmax, double nu, int iters, int inner, int B, int Wf, int Wb, int rule, int k, double budget,
             const float *kappa_init, float *xs_out, float *kappa_out, float *xs_iter, double *stats_out) {
    prob P = {m, n, data, munc, F01, Q00, Q11, pad, kmin, kmax, nu, 0.0, 1000.0};
    rulecfg R = {rule, k, budget};
    float *xf = malloc(sizeof(float) * n * 2), *Pf = malloc(sizeof(float) * n * 4), *pn = calloc(n * 4, sizeof(float));
    float *xs = malloc(sizeof(float) * n * 2), *Ps = malloc(sizeof(float) * n * 4), *lag = calloc(n * 4, sizeof(float));
    float *kap = malloc(sizeof(float) * n);
    for (int64_t i = 0; i < n; ++i) kap[i] = kappa_init ? kappa_init[i] : 1.0f;
    estats st; memset(&st, 0, sizeof st);
    if (B <= 0 || B > n) B = (int)n;
    for (int it = 0; it < iters; ++it) {
        for (int in = 0; in < inner; ++in) {
            forward_blocked(&P, kap, B, Wf, &R, xf, Pf, pn, &st);
            backward_blocked(&P, B, Wb, &R, xf, Pf, pn, xs, Ps, lag, &st);
            estep_kappa(&P, xs, Ps, lag, kap);
        }
        if (xs_iter) memcpy(xs_iter + (size_t)it * n * 2, xs, sizeof(float) * n * 2);
    }
    memcpy(xs_out, xs, sizeof(float) * n * 2);
    memcpy(kappa_out, kap, sizeof(float) * n);
    if (stats_out) { stats_out[0] = (double)st.rerun_f; stats_out[1] = (double)st.rerun_b; stats_out[2] = (double)st.blocks;
                     stats_out[3] = st.max_amp_f; stats_out[4] = st.max_amp_b; }
    free(xf); free(Pf); free(pn); free(xs); free(Ps); free(lag); free(kap);
    return 0;
}

/* emul_ecm + the filtered state of the iteration's closing forward pass (pyx:8300: the NLL pass with the final kappa) and the
 * per-bin NIS of that pass; stats_out gets 9 entries (the five of emul_ecm, then first_f, first_b, chain_f, chain_b).
 * sweeps_fwd_only > 0: no ECM, just that many plain forward+backward passes with kappa_init (the bench recipe's step). */
int emul_ecm2(int64_t m, int64_t n, const float *data, const float *munc, double F01, double Q00, double Q11, double pad,
              double kmin, double kmax, double nu, int iters, int inner, int B, int Wf, int Wb, int rule, int k, double budget,
              const float *kappa_init, float *xs_out, float *kappa_out, float *xf_out, float *nis_out, double *stats_out) {
    prob P = {m, n, data, munc, F01, Q00, Q11, pad, kmin, kmax, nu, 0.0, 1000.0};
    rulecfg R = {rule, k, budget};
    float *xf = malloc(sizeof(float) * n * 2), *Pf = malloc(sizeof(float) * n * 4), *pn = calloc(n * 4, sizeof(float));
    float *xs = malloc(sizeof(float) * n * 2), *Ps = malloc(sizeof(float) * n * 4), *lag = calloc(n * 4, sizeof(float));
    float *kap = malloc(sizeof(float) * n);
    for (int64_t i = 0; i < n; ++i) kap[i] = kappa_init ? kappa_init[i] : 1.0f;
    estats st; memset(&st, 0, sizeof st);
    if (B <= 0 || B > n) B = (int)n;
    for (int it = 0; it < iters; ++it) {
        for (int in = 0; in < inner; ++in) {
            forward_blocked(&P, iters > 0 && inner > 0 ? kap : NULL, B, Wf, &R, xf, Pf, pn, &st);
            backward_blocked(&P, B, Wb, &R, xf, Pf, pn, xs, Ps, lag, &st);
            estep_kappa(&P, xs, Ps, lag, kap);
        }
    }
    forward_blocked(&P, kap, B, Wf, &R, xf, Pf, pn, &st);
    if (iters == 0) backward_blocked(&P, B, Wb, &R, xf, Pf, pn, xs, Ps, lag, &st);
    memcpy(xs_out, xs, sizeof(float) * n * 2);
    memcpy(kappa_out, kap, sizeof(float) * n);
    if (xf_out) memcpy(xf_out, xf, sizeof(float) * n * 2);
    if (nis_out) {
        /* NIS of bin k from the filtered moments of bin k-1 (pyx:443-470): quadForm / m with the stored float32 carries */
        double x0 = 0.0, x1 = 0.0, p00 = 1000.0, p01 = 0.0, p11 = 1000.0;
        for (int64_t kk = 0; kk < n; ++kk) {
            const double kp = clampd((double)kap[kk], kmin, kmax);
            const double xp0 = R32(x0 + F01 * x1);
            const double t00 = p00 + F01 * p01, t01 = p01 + F01 * p11;
            const double a00 = R32(t00 + t01 * F01 + (1.0 / kp) * Q00);
            double s0 = 0, s1 = 0, s2 = 0;
            for (int64_t j = 0; j < m; ++j) {
                double r = (double)munc[j * n + kk] + pad; if (r < 1e-12) r = 1e-12;
                const double w = 1.0 / r, in = (double)data[j * n + kk] - xp0;
                s0 += w; s1 += w * in; s2 += w * in * in;
            }
            const double is = 1.0 + a00 * s0;
            nis_out[kk] = (float)((s2 - (a00 / is) * s1 * s1) / (double)m);
            x0 = xf[kk * 2]; x1 = xf[kk * 2 + 1]; p00 = Pf[kk * 4]; p01 = Pf[kk * 4 + 1]; p11 = Pf[kk * 4 + 3];
        }
    }
    if (stats_out) { stats_out[0] = (double)st.rerun_f; stats_out[1] = (double)st.rerun_b; stats_out[2] = (double)st.blocks;
                     stats_out[3] = st.max_amp_f; stats_out[4] = st.max_amp_b; stats_out[5] = (double)st.first_f;
                     stats_out[6] = (double)st.first_b; stats_out[7] = (double)st.chain_f; stats_out[8] = (double)st.chain_b; }
    free(xf); free(Pf); free(pn); free(xs); free(Ps); free(lag); free(kap);
    return 0;
}

/* Event statistics for the "inductive verification" idea: S = speculative blocked forward state (block B, window W, no
 * validation), T = sequential.  delta_k = T_k - S_k (both components, exact float differences).  An EVENT is a bin where
 * delta_k != delta_{k-1}.  out[0] = events inside blocks, out[1] = block boundaries with an event, out[2] = bins with
 * delta != 0, out[3] = number of maximal runs of consecutive events (episodes), out[4] = longest episode,
 * out[5] = number of 64-bin chunks containing at least one event; gaps between events are returned in hist (log2 bins). */
int emul_events(int64_t m, int64_t n, const float *data, const float *munc, double F01, double Q00, double Q11, double pad,
                const float *kappa, int B, int W, double *out, int64_t *hist) {
    prob P = {m, n, data, munc, F01, Q00, Q11, pad, 5e-3, 5e3, 8.0, 0.0, 1000.0};
    float *xfS = malloc(sizeof(float) * n * 2), *xfT = malloc(sizeof(float) * n * 2);
    float *Pf = malloc(sizeof(float) * n * 4), *pn = calloc(n * 4, sizeof(float));
    rulecfg R = {-1, 0, 0.0};
    estats st; memset(&st, 0, sizeof st);
    forward_blocked(&P, kappa, B, W, &R, xfS, Pf, pn, &st);
    forward_blocked(&P, kappa, (int)n, 0, &R, xfT, Pf, pn, &st);
    double ev = 0, bev = 0, nz = 0, epi = 0, longest = 0, chunks = 0;
    int64_t run = 0, lastev = -1;
    int chunkHas = 0;
    for (int i = 0; i < 32; ++i) hist[i] = 0;
    float pd0 = 0.f, pd1 = 0.f;
    for (int64_t k = 0; k < n; ++k) {
        const float d0 = xfT[2 * k] - xfS[2 * k], d1 = xfT[2 * k + 1] - xfS[2 * k + 1];
        if (d0 != 0.f || d1 != 0.f) nz += 1;
        const int isev = (k > 0) && (d0 != pd0 || d1 != pd1);
        if ((k & 63) == 0) { chunks += chunkHas; chunkHas = 0; }
        if (isev) {
            chunkHas = 1;
            if (k % B == 0) bev += 1; else ev += 1;
            if (lastev >= 0) { int64_t g = k - lastev; int b = 0; while (g > 1) { g >>= 1; ++b; } hist[b]++; }
            if (lastev == k - 1) run++; else { if (run > longest) longest = (double)run; run = 1; epi += 1; }
            lastev = k;
        }
        pd0 = d0; pd1 = d1;
    }
    chunks += chunkHas;
    if (run > longest) longest = (double)run;
    out[0] = ev; out[1] = bev; out[2] = nz; out[3] = epi; out[4] = longest; out[5] = chunks;
    free(xfS); free(xfT); free(Pf); free(pn);
    return 0;
}

/* Superblock repair passes of the bit-exact state chain on the host: pass 0 = every superblock from the cold prior, pass p =
 * every superblock whose carry-in differs from its neighbour's carry-out re-run from it.  For the re-run blocks of each pass:
 * events = bins where (new - old trajectory) changes, batches (64 bins) by number of events.  out[pass*6 + ..] = {blocks re-run,
 * bins, events, batches, batches with > 20 events, bins until new == old (sum over merged blocks)} */
int emul_sb_passes(int64_t m, int64_t n, const float *data, const float *munc, double F01, double Q00, double Q11, double pad,
                   int B, int maxpass, double *out) {
    prob P = {m, n, data, munc, F01, Q00, Q11, pad, 5e-3, 5e3, 8.0, 0.0, 1000.0};
    const int64_t NB = (n + B - 1) / B;
    /* exact gains: one sequential pass of the fused recursion gives P and the state; the state chain alone is re-walked below
       with the SAME covariance trajectory (the covariance chain is validated separately on the device) */
    float *xfT = malloc(sizeof(float) * n * 2), *Pf = malloc(sizeof(float) * n * 4), *pn = calloc(n * 4, sizeof(float));
    rulecfg R = {-1, 0, 0.0};
    estats st; memset(&st, 0, sizeof st);
    forward_blocked(&P, NULL, (int)n, 0, &R, xfT, Pf, pn, &st);
    /* per-bin gain record from the sequential covariance: re-derive gs, p00pred, p10pred, zbar by re-running the P recursion */
    double *gs = malloc(sizeof(double) * n), *zb = malloc(sizeof(double) * n);
    float *pp0 = malloc(sizeof(float) * n), *pp1 = malloc(sizeof(float) * n);
    {
        double p00 = 1000.0, p01 = 0.0, p11 = 1000.0;
        for (int64_t k = 0; k < n; ++k) {
            const double t00 = p00 + F01 * p01, t01 = p01 + F01 * p11;
            const double a00 = R32(t00 + t01 * F01 + Q00), a01 = R32(t01), a10 = R32(p01 + p11 * F01), a11 = R32(p11 + Q11);
            double s0 = 0, s1z = 0;
            for (int64_t j = 0; j < m; ++j) {
                double r = (double)munc[j * n + k] + pad; if (r < 1e-12) r = 1e-12;
                s0 += 1.0 / r; s1z += (double)data[j * n + k] / r;
            }
            const double is = 1.0 + a00 * s0, g = s0 / is, gH = s0 / (is * is);
            gs[k] = g; zb[k] = s1z / s0; pp0[k] = (float)a00; pp1[k] = (float)a10;
            const double i00 = 1.0 - a00 * g, i10 = -(a10 * g);
            p00 = R32(i00 * i00 * a00 + gH * a00 * a00);
            p01 = R32(i00 * (i10 * a00 + a01) + gH * a00 * a10);
            p11 = R32((i10 * i10 * a00 + 2.0 * i10 * a10 + a11) + gH * a10 * a10);
        }
    }
    float *cur = malloc(sizeof(float) * n * 2), *nw = malloc(sizeof(float) * n * 2);
    float *cin = malloc(sizeof(float) * NB * 2), *cout = malloc(sizeof(float) * NB * 2), *cout2 = malloc(sizeof(float) * NB * 2);
#define STATE_STEP(x0, x1, k) do { const float xpf = (x0) + (float)F01 * (x1); const double xp0 = xpf, x1d = (x1); \
        const double dl = gs[k] * (zb[k] - xp0); (x0) = (float)(xp0 + (double)pp0[k] * dl); (x1) = (float)(x1d + (double)pp1[k] * dl); } while (0)
    for (int64_t b = 0; b < NB; ++b) {
        float x0 = 0.f, x1 = 0.f;
        cin[2 * b] = x0; cin[2 * b + 1] = x1;
        const int64_t s = b * B, e = s + B < n ? s + B : n;
        for (int64_t k = s; k < e; ++k) { STATE_STEP(x0, x1, k); cur[2 * k] = x0; cur[2 * k + 1] = x1; }
        cout[2 * b] = x0; cout[2 * b + 1] = x1;
    }
    int pass = 0;
    for (pass = 1; pass <= maxpass; ++pass) {
        double *o = out + (pass - 1) * 6;
        for (int i = 0; i < 6; ++i) o[i] = 0;
        memcpy(cout2, cout, sizeof(float) * NB * 2);
        for (int64_t b = 1; b < NB; ++b) {
            if (cout[2 * (b - 1)] == cin[2 * b] && cout[2 * (b - 1) + 1] == cin[2 * b + 1]) continue;
            float x0 = cout[2 * (b - 1)], x1 = cout[2 * (b - 1) + 1];
            cin[2 * b] = x0; cin[2 * b + 1] = x1;
            const int64_t s = b * B, e = s + B < n ? s + B : n;
            float pd0 = x0 - cur[2 * s - 2], pd1 = x1 - cur[2 * s - 1];
            int evb = 0; int64_t mergedAt = -1;
            o[0] += 1;
            for (int64_t k = s; k < e; ++k) {
                STATE_STEP(x0, x1, k);
                const float d0 = x0 - cur[2 * k], d1 = x1 - cur[2 * k + 1];
                if (d0 != pd0 || d1 != pd1) { o[2] += 1; evb += 1; }
                pd0 = d0; pd1 = d1;
                nw[2 * k] = x0; nw[2 * k + 1] = x1;
                if (mergedAt < 0 && d0 == 0.f && d1 == 0.f) mergedAt = k - s;
                if (((k - s) & 63) == 63 || k == e - 1) { o[3] += 1; if (evb > 20) o[4] += 1; evb = 0; }
                o[1] += 1;
            }
            if (mergedAt >= 0) o[5] += (double)mergedAt; else o[5] += (double)(e - s);
            for (int64_t k = s; k < e; ++k) { cur[2 * k] = nw[2 * k]; cur[2 * k + 1] = nw[2 * k + 1]; }
            cout2[2 * b] = x0; cout2[2 * b + 1] = x1;
        }
        memcpy(cout, cout2, sizeof(float) * NB * 2);
        if (o[0] == 0) break;
    }
    int bad = 0;
    for (int64_t k = 0; k < n * 2; ++k) if (cur[k] != xfT[k]) { bad = 1; break; }
    free(xfT); free(Pf); free(pn); free(gs); free(zb); free(pp0); free(pp1); free(cur); free(nw); free(cin); free(cout); free(cout2);
    return bad ? -pass : pass;
}
