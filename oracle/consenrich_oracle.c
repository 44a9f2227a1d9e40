/*
 * consenrich_oracle.c -- TEST INFRASTRUCTURE ONLY (see consenrich_oracle.h).
 *
 * Straight-line CPU restatement of the reference hot path.  Arithmetic is IEEE fp64 with the reference's
 * float32 rounding points; build with -ffp-contract=off (the reference x86-64 build has no FMA, setup.py:274-279).
 * "pyx" = /root/reference/src/consenrich/cconsenrich.pyx.
 */
#include "consenrich_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>

#define R32(x) ((double)(float)(x)) /* round-trip through float32: the reference's <double><float32_t> casts */

static inline double clampd(double v, double lo, double hi) { /* pyx:135-140 */
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

/* Per-bin collapsed multi-sample sums, pyx:259-282 driven by the loops at pyx:443-456 / 640-653. */
typedef struct obs_sums {
    double s0, s1, s2, sl;
} obs_sums;

static inline obs_sums observe_bin(const cor_fwd_io *io, int64_t k, double level, double pad,
                                   double obs_prec, int want_nll) {
    obs_sums a = {0.0, 0.0, 0.0, 0.0};
    for (int64_t j = 0; j < io->m; ++j) {
        const int64_t idx = j * io->n + k;
        const double innov = (double)io->data[idx] - level;
        double r = (double)io->munc[idx] + pad;
        if (r < 1.0e-12) r = 1.0e-12;
        const double w = obs_prec / r;
        if (want_nll) a.sl += (log(r) - log(obs_prec));
        a.s2 += w * (innov * innov);
        a.s1 += w * innov;
        a.s0 += w;
    }
    return a;
}

/* Adaptive process noise feedback, pyx:510-527 (trend) and pyx:688-705 (level). */
static inline double apn_update(const cor_model *mdl, double apn, float d_stat, double noise_now,
                                double q_diag) {
    if ((double)d_stat > mdl->apn_thresh && noise_now < mdl->apn_max_q) {
        apn *= sqrt(mdl->apn_scale * ((double)d_stat - mdl->apn_thresh) + mdl->apn_pc);
    } else if ((double)d_stat <= mdl->apn_thresh && noise_now > mdl->apn_min_q) {
        apn *= 1.0 / sqrt(mdl->apn_scale * (mdl->apn_thresh - (double)d_stat) + mdl->apn_pc);
    }
    const double after = apn * q_diag;
    if (after < mdl->apn_min_q) apn = mdl->apn_min_q / q_diag;
    else if (after > mdl->apn_max_q) apn = mdl->apn_max_q / q_diag;
    return apn;
}

static void forward_trend(const cor_model *mdl, const cor_fwd_io *io, cor_fwd_out *out) {
    const double F00 = mdl->F[0], F01 = mdl->F[1], F10 = mdl->F[2], F11 = mdl->F[3];
    const double qd = 0.5 * (mdl->Q0[0] + mdl->Q0[3]); /* pyx:6574 */
    const double log2pi = log(6.2831853071795864769);  /* pyx:6483 */
    const double mD = (double)io->m;
    const int store = (io->xf != NULL);
    double x0 = R32(mdl->state_init), x1 = 0.0; /* pyx:340-345 */
    double p00 = R32(mdl->state_covar_init), p01 = 0.0, p10 = 0.0, p11 = R32(mdl->state_covar_init);
    double apn = 1.0;

    out->sum_d = 0.0;
    out->sum_nll = 0.0;
    out->invalid_block_index = -1;

    for (int64_t k = 0; k < io->n; ++k) {
        const int64_t blk = (int64_t)io->block_map[k];
        if (blk < 0 || blk >= io->block_count) { /* pyx:389-392 */
            out->invalid_block_index = k;
            return;
        }
        const double kap = io->kappa ? clampd((double)io->kappa[k], mdl->k_min, mdl->k_max) : 1.0;

        /* state prediction, rounded (pyx:403-406) */
        const double xp0 = F00 * x0 + F01 * x1;
        const double xp1 = F10 * x0 + F11 * x1;
        x0 = R32(xp0);
        x1 = R32(xp1);

        /* process noise (pyx:408-415) */
        const double qs = io->qscale ? (double)io->qscale[k] : apn;
        const double Q00 = (qs / kap) * mdl->Q0[0];
        const double Q01 = (qs / kap) * mdl->Q0[1];
        const double Q10 = (qs / kap) * mdl->Q0[2];
        const double Q11 = (qs / kap) * mdl->Q0[3];

        /* covariance prediction, rounded (pyx:417-430) */
        const double t00 = F00 * p00 + F01 * p10;
        const double t01 = F00 * p01 + F01 * p11;
        const double t10 = F10 * p00 + F11 * p10;
        const double t11 = F10 * p01 + F11 * p11;
        p00 = R32(t00 * F00 + t01 * F01 + Q00);
        p01 = R32(t00 * F10 + t01 * F11 + Q01);
        p10 = R32(t10 * F00 + t11 * F01 + Q10);
        p11 = R32(t10 * F10 + t11 * F11 + Q11);

        const double lam = io->lambda ? clampd((double)io->lambda[k], mdl->w_min, mdl->w_max) : 1.0;
        const obs_sums a = observe_bin(io, k, x0, mdl->pad, lam, io->return_nll);

        /* collapsed innovation statistics (pyx:458-475) */
        const double iscale = 1.0 + p00 * a.s0;
        const double glike = p00 / iscale;
        double quad = a.s2 - glike * (a.s1 * a.s1);
        if (quad < 0.0) quad = 0.0;
        double bin_nll = 0.0;
        if (io->return_nll) {
            bin_nll = 0.5 * (a.sl + log(iscale) + quad + mD * log2pi);
            out->sum_nll += bin_nll;
        }
        const double stat = (io->return_nll && io->store_nll_in_d) ? bin_nll : quad / mD;
        io->D[k] = (float)stat;
        out->sum_d += (double)io->D[k];

        /* state update (pyx:477-479) */
        const double delta = a.s1 / iscale;
        x0 = R32(x0 + p00 * delta);
        x1 = R32(x1 + p10 * delta);

        /* Joseph-form covariance update (pyx:481-495) */
        const double gG = a.s0 / iscale;
        const double gH = a.s0 / (iscale * iscale);
        const double i00 = 1.0 - (p00 * gG);
        const double i10 = -(p10 * gG);
        const double n00 = (i00 * i00 * p00) + (gH * (p00 * p00));
        const double n01 = (i00 * (i10 * p00 + p01)) + (gH * (p00 * p10));
        const double n11 = ((i10 * i10 * p00) + 2.0 * i10 * p10 + p11) + (gH * (p10 * p10));
        p00 = R32(n00);
        p01 = R32(n01);
        p10 = p01;
        p11 = R32(n11);

        if (store) { /* pyx:497-508 */
            io->xf[k * 2] = (float)x0;
            io->xf[k * 2 + 1] = (float)x1;
            io->Pf[k * 4] = (float)p00;
            io->Pf[k * 4 + 1] = (float)p01;
            io->Pf[k * 4 + 2] = (float)p10;
            io->Pf[k * 4 + 3] = (float)p11;
            if (k > 0) {
                io->pnoise[(k - 1) * 4] = (float)Q00;
                io->pnoise[(k - 1) * 4 + 1] = (float)Q01;
                io->pnoise[(k - 1) * 4 + 2] = (float)Q10;
                io->pnoise[(k - 1) * 4 + 3] = (float)Q11;
            }
        }

        if (io->use_apn && !io->qscale) apn = apn_update(mdl, apn, io->D[k], 0.5 * (Q00 + Q11), qd);
    }
}

static void forward_level(const cor_model *mdl, const cor_fwd_io *io, cor_fwd_out *out) {
    const double q0 = mdl->Q0[0];
    const double log2pi = log(6.2831853071795864769);
    const double mD = (double)io->m;
    const int store = (io->xf != NULL);
    double x = mdl->state_init; /* carries are NOT rounded in the level model (pyx:579-580) */
    double p = mdl->state_covar_init;
    double apn = 1.0;

    out->sum_d = 0.0;
    out->sum_nll = 0.0;
    out->invalid_block_index = -1;

    for (int64_t k = 0; k < io->n; ++k) {
        const int64_t blk = (int64_t)io->block_map[k];
        if (blk < 0 || blk >= io->block_count) {
            out->invalid_block_index = k;
            return;
        }
        const double kap = io->kappa ? clampd((double)io->kappa[k], mdl->k_min, mdl->k_max) : 1.0;
        const double qs = io->qscale ? (double)io->qscale[k] : apn;
        const double Q = (qs / kap) * q0; /* pyx:626 */
        p += Q;

        const double lam = io->lambda ? clampd((double)io->lambda[k], mdl->w_min, mdl->w_max) : 1.0;
        const obs_sums a = observe_bin(io, k, x, mdl->pad, lam, io->return_nll);

        const double iscale = 1.0 + p * a.s0; /* pyx:655-671 */
        const double glike = p / iscale;
        double quad = a.s2 - glike * (a.s1 * a.s1);
        if (quad < 0.0) quad = 0.0;
        double bin_nll = 0.0;
        if (io->return_nll) {
            bin_nll = 0.5 * (a.sl + log(iscale) + quad + mD * log2pi);
            out->sum_nll += bin_nll;
        }
        const double stat = (io->return_nll && io->store_nll_in_d) ? bin_nll : quad / mD;
        io->D[k] = (float)stat;
        out->sum_d += (double)io->D[k];

        const double delta = a.s1 / iscale; /* pyx:673-680 */
        x += p * delta;
        const double gG = a.s0 / iscale;
        const double gH = a.s0 / (iscale * iscale);
        const double ikh = 1.0 - p * gG;
        p = (ikh * ikh * p) + (gH * (p * p));

        if (store) { /* pyx:682-686 */
            io->xf[k] = (float)x;
            io->Pf[k] = (float)p;
            if (k > 0) io->pnoise[k - 1] = (float)Q;
        }

        if (io->use_apn && !io->qscale) apn = apn_update(mdl, apn, io->D[k], apn * q0, q0);
    }
}

void cor_forward(const cor_model *mdl, const cor_fwd_io *io, cor_fwd_out *out) {
    if (mdl->state_dim == 1) forward_level(mdl, io, out);
    else forward_trend(mdl, io, out);
}

static void backward_trend(const cor_model *mdl, int64_t m, int64_t n, const float *data,
                           const float *xf, const float *Pf, const float *pn, float *xs, float *Ps,
                           float *lag, int64_t lag_rows, float *resid) {
    const double F00 = mdl->F[0], F01 = mdl->F[1], F10 = mdl->F[2], F11 = mdl->F[3];
    if (n <= 0) return;
    const int64_t e = n - 1; /* pyx:6744-6755 */
    xs[e * 2] = xf[e * 2];
    xs[e * 2 + 1] = xf[e * 2 + 1];
    for (int c = 0; c < 4; ++c) Ps[e * 4 + c] = Pf[e * 4 + c];
    for (int64_t j = 0; j < m; ++j) resid[e * m + j] = (float)((double)data[j * n + e] - (double)xs[e * 2]);

    for (int64_t k = n - 2; k >= 0; --k) { /* pyx:6758-6848 */
        const double f00 = (double)Pf[k * 4], f01 = (double)Pf[k * 4 + 1];
        const double f10 = (double)Pf[k * 4 + 2], f11 = (double)Pf[k * 4 + 3];
        const double a0 = (double)xf[k * 2], a1 = (double)xf[k * 2 + 1];
        const double xp0 = F00 * a0 + F01 * a1;
        const double xp1 = F10 * a0 + F11 * a1;
        const double Q00 = (double)pn[k * 4], Q01 = (double)pn[k * 4 + 1];
        const double Q10 = (double)pn[k * 4 + 2], Q11 = (double)pn[k * 4 + 3];

        double c00 = F00 * f00 + F01 * f10;
        double c01 = F00 * f01 + F01 * f11;
        double c10 = F10 * f00 + F11 * f10;
        double c11 = F10 * f01 + F11 * f11;
        const double pp00 = c00 * F00 + c01 * F01 + Q00;
        const double pp01 = c00 * F10 + c01 * F11 + Q01;
        const double pp10 = c10 * F00 + c11 * F01 + Q10;
        const double pp11 = c10 * F10 + c11 * F11 + Q11;

        const double det = (pp00 * pp11) - (pp01 * pp10); /* unguarded (pyx:6780) */
        const double v00 = pp11 / det, v01 = -pp01 / det, v10 = -pp10 / det, v11 = pp00 / det;

        c00 = f00 * F00 + f01 * F01; /* Pf F^T */
        c01 = f00 * F10 + f01 * F11;
        c10 = f10 * F00 + f11 * F01;
        c11 = f10 * F10 + f11 * F11;
        const double J00 = c00 * v00 + c01 * v10;
        const double J01 = c00 * v01 + c01 * v11;
        const double J10 = c10 * v00 + c11 * v10;
        const double J11 = c10 * v01 + c11 * v11;

        const double dx0 = (double)xs[(k + 1) * 2] - xp0;
        const double dx1 = (double)xs[(k + 1) * 2 + 1] - xp1;
        xs[k * 2] = (float)(a0 + (J00 * dx0 + J01 * dx1));
        xs[k * 2 + 1] = (float)(a1 + (J10 * dx0 + J11 * dx1));

        const double d00 = (double)Ps[(k + 1) * 4] - pp00;
        const double d01 = (double)Ps[(k + 1) * 4 + 1] - pp01;
        const double d10 = (double)Ps[(k + 1) * 4 + 2] - pp10;
        const double d11 = (double)Ps[(k + 1) * 4 + 3] - pp11;
        const double r00 = d00 * J00 + d01 * J01;
        const double r01 = d00 * J10 + d01 * J11;
        const double r10 = d10 * J00 + d11 * J01;
        const double r11 = d10 * J10 + d11 * J11;
        const double s00 = f00 + (J00 * r00 + J01 * r10);
        const double s01 = f01 + (J00 * r01 + J01 * r11);
        const double s11 = f11 + (J10 * r01 + J11 * r11);
        Ps[k * 4] = (float)s00;
        Ps[k * 4 + 1] = (float)s01;
        Ps[k * 4 + 2] = (float)s01;
        Ps[k * 4 + 3] = (float)s11;

        if (k < lag_rows) { /* pyx:6825-6844 */
            lag[k * 4] = (float)(c00 + (J00 * d00 + J01 * d10));
            lag[k * 4 + 1] = (float)(c01 + (J00 * d01 + J01 * d11));
            lag[k * 4 + 2] = (float)(c10 + (J10 * d00 + J11 * d10));
            lag[k * 4 + 3] = (float)(c11 + (J10 * d01 + J11 * d11));
        }
        for (int64_t j = 0; j < m; ++j)
            resid[k * m + j] = (float)((double)data[j * n + k] - (double)xs[k * 2]);
    }
}

static void backward_level(int64_t m, int64_t n, const float *data, const float *xf, const float *Pf,
                           const float *pn, float *xs, float *Ps, float *lag, int64_t lag_rows,
                           float *resid) {
    if (n <= 0) return;
    const int64_t e = n - 1; /* pyx:7117-7123 */
    xs[e] = xf[e];
    Ps[e] = Pf[e];
    for (int64_t j = 0; j < m; ++j) resid[e * m + j] = (float)((double)data[j * n + e] - (double)xs[e]);
    for (int64_t k = n - 2; k >= 0; --k) { /* pyx:7125-7148 */
        const double pf = (double)Pf[k];
        double pp = pf + (double)pn[k];
        if (pp < 1.0e-12) pp = 1.0e-12;
        const double J = pf / pp;
        const double dx = (double)xs[k + 1] - (double)xf[k];
        xs[k] = (float)((double)xf[k] + J * dx);
        const double dP = (double)Ps[k + 1] - pp;
        double ps = pf + (J * J * dP);
        if (ps < 0.0) ps = 0.0;
        Ps[k] = (float)ps;
        if (k < lag_rows) lag[k] = (float)(pf + (J * dP));
        for (int64_t j = 0; j < m; ++j)
            resid[k * m + j] = (float)((double)data[j * n + k] - (double)xs[k]);
    }
}

void cor_backward(const cor_model *mdl, int64_t m, int64_t n, const float *data, const float *xf,
                  const float *Pf, const float *pnoise, float *xs, float *Ps, float *lag,
                  int64_t lag_rows, float *resid) {
    if (mdl->state_dim == 1) backward_level(m, n, data, xf, Pf, pnoise, xs, Ps, lag, lag_rows, resid);
    else backward_trend(mdl, m, n, data, xf, Pf, pnoise, xs, Ps, lag, lag_rows, resid);
}

/* lambda E-step: pyx:8210-8239 (trend) == pyx:7470-7494 (level) */
static void estep_lambda(const cor_model *mdl, const cor_ecm_cfg *cfg, int64_t m, int64_t n,
                         const float *data, const float *munc, const int32_t *block_map,
                         int64_t block_count, const float *xs, const float *Ps, float *lambda) {
    const int d = mdl->state_dim;
    for (int64_t k = 0; k < n; ++k) {
        const int64_t b = (int64_t)block_map[k];
        if (b < 0 || b >= block_count) {
            lambda[k] = 1.0f;
            continue;
        }
        double p00 = (double)Ps[k * d * d];
        if (p00 < 0.0) p00 = 0.0;
        double u2 = 0.0;
        for (int64_t j = 0; j < m; ++j) {
            double r = (double)munc[j * n + k] + mdl->pad;
            if (r < 1.0e-12) r = 1.0e-12;
            const double res = (double)data[j * n + k] - (double)xs[k * d];
            const double t = (res * res + p00);
            u2 += t / r;
        }
        double w = (cfg->nu + (double)m) / (cfg->nu + u2);
        if (w < mdl->w_min) w = mdl->w_min;
        else if (w > mdl->w_max) w = mdl->w_max;
        lambda[k] = (float)w;
    }
}

/* kappa E-step, trend: pyx:8244-8298 with the MAT2 helpers pyx:4123-4175 */
static void estep_kappa_trend(const cor_model *mdl, const cor_ecm_cfg *cfg, int64_t n,
                              const int32_t *block_map, int64_t block_count, const float *qscale,
                              const float *xs, const float *Ps, const float *lag, float *kappa) {
    const double f00 = mdl->F[0], f01 = mdl->F[1], f10 = mdl->F[2], f11 = mdl->F[3];
    const double det = (mdl->Q0[0] * mdl->Q0[3] - mdl->Q0[1] * mdl->Q0[2]);
    const double qi00 = mdl->Q0[3] / det, qi01 = -mdl->Q0[1] / det;
    const double qi10 = -mdl->Q0[2] / det, qi11 = mdl->Q0[0] / det;
    kappa[0] = 1.0f;
    for (int64_t k = 0; k < n - 1; ++k) {
        const int64_t b = (int64_t)block_map[k];
        if (b < 0 || b >= block_count) {
            kappa[k + 1] = 1.0f;
            continue;
        }
        const double x0 = (double)xs[k * 2], x1 = (double)xs[k * 2 + 1];
        const double y0 = (double)xs[(k + 1) * 2], y1 = (double)xs[(k + 1) * 2 + 1];
        /* E[xx^T], E[yy^T], E[xy^T] */
        const double xx00 = (double)Ps[k * 4] + x0 * x0, xx01 = (double)Ps[k * 4 + 1] + x0 * x1;
        const double xx10 = (double)Ps[k * 4 + 2] + x1 * x0, xx11 = (double)Ps[k * 4 + 3] + x1 * x1;
        const double yy00 = (double)Ps[(k + 1) * 4] + y0 * y0, yy01 = (double)Ps[(k + 1) * 4 + 1] + y0 * y1;
        const double yy10 = (double)Ps[(k + 1) * 4 + 2] + y1 * y0, yy11 = (double)Ps[(k + 1) * 4 + 3] + y1 * y1;
        const double xy00 = (double)lag[k * 4] + x0 * y0, xy01 = (double)lag[k * 4 + 1] + x0 * y1;
        const double xy10 = (double)lag[k * 4 + 2] + x1 * y0, xy11 = (double)lag[k * 4 + 3] + x1 * y1;
        /* yx = xy^T ; Ft = F^T */
        const double yx00 = xy00, yx01 = xy10, yx10 = xy01, yx11 = xy11;
        const double t00 = f00, t01 = f10, t10 = f01, t11 = f11;
        /* ww = yy - yx*Ft - F*xy + (F*xx)*Ft */
        double w00 = yy00 - (yx00 * t00 + yx01 * t10);
        double w01 = yy01 - (yx00 * t01 + yx01 * t11);
        double w10 = yy10 - (yx10 * t00 + yx11 * t10);
        double w11 = yy11 - (yx10 * t01 + yx11 * t11);
        w00 = w00 - (f00 * xy00 + f01 * xy10);
        w01 = w01 - (f00 * xy01 + f01 * xy11);
        w10 = w10 - (f10 * xy00 + f11 * xy10);
        w11 = w11 - (f10 * xy01 + f11 * xy11);
        const double g00 = f00 * xx00 + f01 * xx10, g01 = f00 * xx01 + f01 * xx11;
        const double g10 = f10 * xx00 + f11 * xx10, g11 = f10 * xx01 + f11 * xx11;
        w00 = w00 + (g00 * t00 + g01 * t10);
        w01 = w01 + (g00 * t01 + g01 * t11);
        w10 = w10 + (g10 * t00 + g11 * t10);
        w11 = w11 + (g10 * t01 + g11 * t11);
        if (w00 < 0.0) w00 = 0.0;
        if (w11 < 0.0) w11 = 0.0;
        double delta = qi00 * w00 + qi01 * w10 + qi10 * w01 + qi11 * w11; /* traceProd pyx:4174 */
        if (qscale) delta = delta / (double)qscale[k + 1];
        if (delta < 0.0) delta = 0.0;
        double kap = (cfg->nu + 2.0) / (cfg->nu + delta);
        if (kap < mdl->k_min) kap = mdl->k_min;
        else if (kap > mdl->k_max) kap = mdl->k_max;
        kappa[k + 1] = (float)kap;
    }
}

/* kappa E-step, level: pyx:7496-7521 */
static void estep_kappa_level(const cor_model *mdl, const cor_ecm_cfg *cfg, int64_t n,
                              const int32_t *block_map, int64_t block_count, const float *qscale,
                              const float *xs, const float *Ps, const float *lag, float *kappa) {
    const double q0inv = 1.0 / mdl->Q0[0];
    kappa[0] = 1.0f;
    for (int64_t k = 0; k < n - 1; ++k) {
        const int64_t b = (int64_t)block_map[k];
        if (b < 0 || b >= block_count) {
            kappa[k + 1] = 1.0f;
            continue;
        }
        const double x0 = (double)xs[k], y0 = (double)xs[k + 1];
        const double pk = (double)Ps[k], pk1 = (double)Ps[k + 1], ck = (double)lag[k];
        double delta = ((pk1 + y0 * y0) - (2.0 * (ck + x0 * y0)) + (pk + x0 * x0)) * q0inv;
        if (qscale) delta = delta / (double)qscale[k + 1];
        if (delta < 0.0) delta = 0.0;
        double kap = (cfg->nu + 1.0) / (cfg->nu + delta);
        if (kap < mdl->k_min) kap = mdl->k_min;
        else if (kap > mdl->k_max) kap = mdl->k_max;
        kappa[k + 1] = (float)kap;
    }
}

void cor_ecm(const cor_model *mdl, const cor_ecm_cfg *cfg, int64_t m, int64_t n, const float *data,
             const float *munc, const int32_t *block_map, int64_t block_count, const float *qscale,
             float *lambda, float *kappa, float *xf, float *Pf, float *pnoise, float *xs, float *Ps,
             float *lag, float *resid, float *D_scratch, double *nll_path, cor_ecm_out *out) {
    const int64_t lag_rows = (n - 1 > 1) ? n - 1 : 1; /* pyx:7935 */
    cor_fwd_io io;
    cor_fwd_out fo;
    io.m = m;
    io.n = n;
    io.data = data;
    io.munc = munc;
    io.block_map = block_map;
    io.block_count = block_count;
    io.lambda = cfg->use_lambda ? lambda : NULL;
    io.kappa = cfg->use_kappa ? kappa : NULL;
    io.qscale = qscale;
    io.use_apn = cfg->use_apn;
    io.store_nll_in_d = 0;
    io.D = D_scratch;

    out->iters_done = 0;
    out->final_nll = 0.0;
    out->initial_nll = 0.0;
    out->abs_rel_change = 0.0;
    out->rel_improvement = 0.0;
    out->stable_iters = 0;
    out->nll_increase_count = 0;
    out->converged = 0;
    out->skipped = 0;
    out->has_initial_nll = 0;
    out->invalid_block_index = -1;

    if (n <= 5) { /* pyx:7998-8129: filter+smoother only */
        out->skipped = 1;
        if (n > 0 && m > 0) {
            io.return_nll = 0;
            io.xf = xf;
            io.Pf = Pf;
            io.pnoise = pnoise;
            cor_forward(mdl, &io, &fo);
            if (fo.invalid_block_index >= 0) {
                out->invalid_block_index = fo.invalid_block_index;
                return;
            }
            cor_backward(mdl, m, n, data, xf, Pf, pnoise, xs, Ps, lag, lag_rows, resid);
            io.return_nll = 1;
            io.xf = NULL;
            io.Pf = NULL;
            io.pnoise = NULL;
            cor_forward(mdl, &io, &fo);
            out->final_nll = fo.sum_nll;
        }
        out->initial_nll = out->final_nll;
        return;
    }

    double prev = 1.0e16, cur = 0.0; /* pyx:7958-7992 */
    int have_init = 0;
    const int64_t patience = 2;

    for (int64_t it = 0; it < cfg->max_iters; ++it) { /* pyx:8151 */
        out->iters_done = it + 1;
        for (int64_t inner = 0; inner < cfg->inner_iters; ++inner) {
            io.return_nll = 0;
            io.xf = xf;
            io.Pf = Pf;
            io.pnoise = pnoise;
            cor_forward(mdl, &io, &fo);
            if (fo.invalid_block_index >= 0) {
                out->invalid_block_index = fo.invalid_block_index;
                return;
            }
            cor_backward(mdl, m, n, data, xf, Pf, pnoise, xs, Ps, lag, lag_rows, resid);
            if (cfg->use_lambda)
                estep_lambda(mdl, cfg, m, n, data, munc, block_map, block_count, xs, Ps, lambda);
            if (cfg->use_kappa) {
                if (mdl->state_dim == 1)
                    estep_kappa_level(mdl, cfg, n, block_map, block_count, qscale, xs, Ps, lag, kappa);
                else
                    estep_kappa_trend(mdl, cfg, n, block_map, block_count, qscale, xs, Ps, lag, kappa);
            }
        }
        io.return_nll = 1; /* pyx:8300 */
        io.xf = NULL;
        io.Pf = NULL;
        io.pnoise = NULL;
        cor_forward(mdl, &io, &fo);
        if (fo.invalid_block_index >= 0) {
            out->invalid_block_index = fo.invalid_block_index;
            return;
        }
        cur = fo.sum_nll;
        if (nll_path) nll_path[it] = cur;

        /* convergence bookkeeping, pyx:8337-8407 */
        const int have_prev = have_init;
        if (!have_prev) {
            out->initial_nll = cur;
            have_init = 1;
        } else if (cur > prev + (1.0e-12 * fmax(fabs(prev), 1.0))) {
            out->nll_increase_count += 1;
        }
        double delta, scale;
        if (have_prev) {
            delta = fabs(cur - prev);
            scale = fabs(prev);
        } else {
            delta = 0.0;
            scale = fabs(cur);
        }
        if (fabs(cur) > scale) scale = fabs(cur);
        if (scale < 1.0) scale = 1.0;
        if (have_prev) {
            out->rel_improvement = (prev - cur) / scale;
            out->abs_rel_change = delta / scale;
        } else {
            out->rel_improvement = 0.0;
            out->abs_rel_change = 0.0;
        }
        const double tol = cfg->rtol * scale;
        prev = cur;
        if (have_prev && delta <= tol) out->stable_iters += 1;
        else out->stable_iters = 0;
        if (out->stable_iters >= patience) {
            out->converged = 1;
            break;
        }
    }
    out->has_initial_nll = have_init;
    out->final_nll = prev;
}

void cor_transition_sums(int32_t state_dim, int64_t n, const double *xs, const double *Ps,
                         const double *lag, const double *F, double *sum_level, double *sum_trend,
                         int64_t *count) {
    *sum_level = 0.0;
    *sum_trend = 0.0;
    *count = (n - 1 > 0) ? n - 1 : 0;
    if (n - 1 <= 0) return;
    if (state_dim == 1) { /* pyx:852-863 */
        double acc = 0.0;
        for (int64_t k = 0; k < n - 1; ++k) {
            const double x0 = xs[k], y0 = xs[k + 1];
            const double e0 = Ps[k] + (x0 * x0);
            const double e1 = Ps[k + 1] + (y0 * y0);
            const double ec = lag[k] + (x0 * y0);
            double mom = e1 - (2.0 * ec) + e0;
            if (mom < 0.0) mom = 0.0;
            acc += mom;
        }
        *sum_level = acc;
        return;
    }
    const double f00 = F[0], f01 = F[1], f10 = F[2], f11 = F[3];
    double accL = 0.0, accT = 0.0;
    for (int64_t k = 0; k < n - 1; ++k) { /* pyx:770-813 */
        const double x00 = xs[k * 2], x01 = xs[k * 2 + 1];
        const double y0 = xs[(k + 1) * 2], y1 = xs[(k + 1) * 2 + 1];
        const double a00 = Ps[k * 4] + (x00 * x00), a01 = Ps[k * 4 + 1] + (x00 * x01);
        const double a10 = Ps[k * 4 + 2] + (x01 * x00), a11 = Ps[k * 4 + 3] + (x01 * x01);
        const double b00 = Ps[(k + 1) * 4] + (y0 * y0), b11 = Ps[(k + 1) * 4 + 3] + (y1 * y1);
        const double c00 = lag[k * 4] + (x00 * y0), c01 = lag[k * 4 + 1] + (x00 * y1);
        const double c10 = lag[k * 4 + 2] + (x01 * y0), c11 = lag[k * 4 + 3] + (x01 * y1);
        double lm = (b00 - (2.0 * ((f00 * c00) + (f01 * c10))) + (f00 * f00 * a00) + (f00 * f01 * a01)
                     + (f01 * f00 * a10) + (f01 * f01 * a11));
        double tm = (b11 - (2.0 * ((f10 * c01) + (f11 * c11))) + (f10 * f10 * a00) + (f10 * f11 * a01)
                     + (f11 * f10 * a10) + (f11 * f11 * a11));
        if (lm < 0.0) lm = 0.0;
        if (tm < 0.0) tm = 0.0;
        accL += lm;
        accT += tm;
    }
    *sum_level = accL;
    *sum_trend = accT;
}


/* ================================================================================================================
 * background update natives
 * ================================================================================================================ */
int64_t cor_background_stats(int64_t m, int64_t n, const float *resid, const float *inv_var, double *weight,
                             double *rhs) { /* pyx:9712-9724 */
    int64_t support = 0;
    for (int64_t i = 0; i < n; ++i) {
        double ws = 0.0, rs = 0.0;
        for (int64_t j = 0; j < m; ++j) {
            const double w = (double)inv_var[j * n + i];
            ws += w;
            rs += w * (double)resid[j * n + i];
        }
        weight[i] = ws;
        rhs[i] = rs;
        if (ws > 0.0) ++support;
    }
    return support;
}

/* penalty stencils: pyx:905-941 */
static double pen2_diag(int64_t n, int64_t i, double lam) {
    if (n < 3 || lam <= 0.0) return 0.0;
    if (n == 3) return i == 1 ? 4.0 * lam : lam;
    if (i == 0 || i == n - 1) return lam;
    if (i == 1 || i == n - 2) return 5.0 * lam;
    return 6.0 * lam;
}
static double pen2_off1(int64_t n, int64_t i, double lam) {
    if (n < 3 || lam <= 0.0) return 0.0;
    if (n == 3) return -2.0 * lam;
    return (i == 0 || i == n - 2) ? -2.0 * lam : -4.0 * lam;
}
static double pen1_diag(int64_t n, int64_t i, double lam) {
    if (n < 2 || lam <= 0.0) return 0.0;
    return (i == 0 || i == n - 1) ? lam : 2.0 * lam;
}
static double pen1_off1(int64_t n, double lam) { return (n < 2 || lam <= 0.0) ? 0.0 : -lam; }

int64_t cor_solve_background(int64_t n, const double *weight, const double *rhs_in, double lam, int zero_center,
                             double lam_first, double *out, double *bad_value) {
    const double floor_ = 1.0e-12;
    int64_t bad = -1;
    double badv = 0.0;
    for (int64_t i = 0; i < n; ++i) out[i] = 0.0;
    if (n <= 0) return -1;
    if (n == 1) { /* pyx:1001-1011 */
        if (!zero_center) {
            const double den = weight[0];
            if (den < floor_) {
                if (bad_value) *bad_value = den;
                return 0;
            }
            out[0] = rhs_in[0] / den;
        }
        return -1;
    }
    double *d = (double *)malloc(sizeof(double) * (size_t)n * 4);
    double *r = d + n, *c = r + n, *l1 = c + n;
#define CSR_FLOOR(i)                     \
    if (d[i] < floor_) {                 \
        if (bad < 0) { bad = (i); badv = d[i]; } \
        d[i] = floor_;                   \
    }
    for (int64_t i = 0; i < n; ++i) { /* pyx:1026-1036 */
        d[i] = weight[i] + pen1_diag(n, i, lam_first) + pen2_diag(n, i, lam);
        r[i] = rhs_in[i];
        c[i] = 1.0;
        l1[i] = 0.0;
        CSR_FLOOR(i)
    }
    /* factorisation, pyx:1041-1063: the second sub-diagonal of L is lam / d[i-2] */
    double off = pen1_off1(n, lam_first) + pen2_off1(n, 0, lam);
    l1[1] = off / d[0];
    d[1] = d[1] - l1[1] * l1[1] * d[0];
    CSR_FLOOR(1)
    for (int64_t i = 2; i < n; ++i) {
        off = pen1_off1(n, lam_first) + pen2_off1(n, i - 1, lam);
        l1[i] = (off - lam * l1[i - 1]) / d[i - 1];
        d[i] = d[i] - l1[i] * l1[i] * d[i - 1] - (lam * lam) / d[i - 2];
        CSR_FLOOR(i)
    }
#undef CSR_FLOOR
    /* forward substitution for the data right-hand side and for A^-1 1, pyx:1067-1072 */
    r[1] = r[1] - l1[1] * r[0];
    c[1] = c[1] - l1[1] * c[0];
    for (int64_t i = 2; i < n; ++i) {
        const double l2 = lam / d[i - 2];
        r[i] = r[i] - l1[i] * r[i - 1] - l2 * r[i - 2];
        c[i] = c[i] - l1[i] * c[i - 1] - l2 * c[i - 2];
    }
    for (int64_t i = 0; i < n; ++i) { r[i] = r[i] / d[i]; c[i] = c[i] / d[i]; } /* pyx:1074-1076 */
    /* back substitution, pyx:1079-1084 */
    r[n - 2] = r[n - 2] - l1[n - 1] * r[n - 1];
    c[n - 2] = c[n - 2] - l1[n - 1] * c[n - 1];
    for (int64_t i = n - 3; i >= 0; --i) {
        const double l2 = lam / d[i];
        r[i] = r[i] - l1[i + 1] * r[i + 1] - l2 * r[i + 2];
        c[i] = c[i] - l1[i + 1] * c[i + 1] - l2 * c[i + 2];
    }
    if (zero_center) { /* pyx:1086-1096 */
        double sr = 0.0, sc = 0.0;
        for (int64_t i = 0; i < n; ++i) { sr += r[i]; sc += c[i]; }
        const double mu = fabs(sc) > floor_ ? sr / sc : sr / (double)n;
        for (int64_t i = 0; i < n; ++i) out[i] = r[i] - mu * c[i];
    } else {
        for (int64_t i = 0; i < n; ++i) out[i] = r[i];
    }
    free(d);
    if (bad >= 0 && bad_value) *bad_value = badv;
    return bad;
}


/* ================================================================================================================
 * delete-block calibration natives (cuncertainty.pyx)
 * ================================================================================================================ */
static double exchangeable_information(double sum_w, double sum_sqrt, int64_t count, double rho) { /* unc:37-57 */
    if (count <= 0 || sum_w <= 0.0) return 0.0;
    if (rho <= 0.0) return sum_w;
    const double omr = 1.0 - rho;
    const double denom = omr + rho * (double)count;
    const double adjusted = sum_w / omr - rho * sum_sqrt * sum_sqrt / (omr * denom);
    return adjusted > sum_w ? sum_w : adjusted;
}
static inline double munc_at(const void *munc, int f64, int64_t idx) {
    return f64 ? ((const double *)munc)[idx] : (double)((const float *)munc)[idx];
}

void cor_total_information(int64_t m, int64_t n, const void *munc, int munc_f64, const uint8_t *active,
                           const double *lambda, double pad, double rho, double *total) { /* unc:131-156 */
    for (int64_t i = 0; i < n; ++i) {
        double tot = 0.0, ssq = 0.0;
        int64_t count = 0;
        const double lam = lambda ? lambda[i] : 1.0;
        for (int64_t j = 0; j < m; ++j) {
            if (active[j * n + i] != 0) {
                const double v = lam / (munc_at(munc, munc_f64, j * n + i) + pad);
                tot += v;
                if (rho > 0.0) { ssq += sqrt(v); ++count; }
            }
        }
        total[i] = rho > 0.0 ? exchangeable_information(tot, ssq, count, rho) : tot;
    }
}

void cor_fold_mask_information(int64_t m, int64_t n, int64_t block_len, int64_t fold, const int32_t *block_fold,
                               const int64_t *reps_count, const int64_t *reps, int64_t slots, const void *munc,
                               int munc_f64, const uint8_t *active, const double *total, const double *lambda, double pad,
                               double rho, uint8_t *mask, double *kept, double *heldout, double *h, double *nominal) {
    const int64_t block_count = (n + block_len - 1) / block_len;
    for (int64_t k = 0; k < m * n; ++k) mask[k] = 1;
    for (int64_t i = 0; i < n; ++i) { heldout[i] = 0.0; if (nominal) nominal[i] = 0.0; }
    for (int64_t b = 0; b < block_count; ++b) { /* unc:253-272 */
        if (block_fold[b] != fold) continue;
        const int64_t start = b * block_len, end = start + block_len > n ? n : start + block_len;
        for (int64_t hh = 0; hh < reps_count[b]; ++hh) {
            const int64_t rep = reps[b * slots + hh];
            for (int64_t i = start; i < end; ++i) {
                mask[rep * n + i] = 0;
                if (active[rep * n + i] != 0) {
                    double v = 1.0 / (munc_at(munc, munc_f64, rep * n + i) + pad);
                    if (lambda) v *= lambda[i];
                    if (rho <= 0.0) heldout[i] += v;
                    if (nominal) nominal[i] += v;
                }
            }
        }
    }
    for (int64_t i = 0; i < n; ++i) { /* unc:273-302 */
        const double tot = total[i];
        if (rho > 0.0) {
            double kp = 0.0, ssq = 0.0;
            int64_t count = 0;
            const double lam = lambda ? lambda[i] : 1.0;
            for (int64_t j = 0; j < m; ++j)
                if (active[j * n + i] != 0 && mask[j * n + i] != 0) {
                    const double v = lam / (munc_at(munc, munc_f64, j * n + i) + pad);
                    kp += v;
                    ssq += sqrt(v);
                    ++count;
                }
            kp = exchangeable_information(kp, ssq, count, rho);
            kept[i] = kp;
            heldout[i] = tot - kp;
        } else {
            kept[i] = tot - heldout[i];
        }
        h[i] = tot > 0.0 ? heldout[i] / tot : NAN;
    }
}
