// csr_background.h -- SURVEY 8(f) rank 1: the native pieces of the background update between ECM phases
// (reference: cconsenrich.pyx:944-1096 `csolveZeroCenteredBackground`, pyx:9700-9724
// `cbackgroundWeightedStatsWithSupport`; callers core.py:5064-5137, 7531-7589, 8085-8378).
//
// The reference factorises the pentadiagonal SPD system  (diag(w) + lamF D1'D1 + lam D2'D2) x = r  of one chromosome
// with a sequential LDL' (1.2 M dependent steps).  Its Green's function decays over thousands of bins for the default
// penalties (span 750 bins: an overlap of 8000 bins still leaves 5e-4), so windowed / speculative variants do not
// pay.  Here the system is solved EXACTLY by a two-level partition (domain decomposition with 2-bin separators):
//
//   * every chain is cut into blocks of Bp bins; the last two bins of a block (all but the chain's last block) form a
//     separator S_k, the rest is the interior I_k.  Bandwidth 2 => interiors are mutually decoupled given the
//     separators.
//   * k_bg_local (one lane per interior, 64 interiors per wavefront, blocked-transposed storage => coalesced rows):
//     LDL' of A[I_k,I_k] with the reference's own recurrence and pivot floor, forward/back substitution of the data
//     column(s) and of the four coupling columns A[I_k, S_{k-1} u S_k]; emits the 4x4 Schur contribution
//     T_k = C_k' A_II^-1 C_k, t_k = C_k' A_II^-1 r and keeps the solution columns.
//   * k_bg_reduced (one lane per chain): block-tridiagonal (2x2 blocks) elimination over the separators.
//   * k_bg_combine (one thread per bin): x_I = A_II^-1 r - (A_II^-1 C) x_S, written in natural order.
//   * k_bg_center (zero-sum Lagrange correction, pyx:1086-1096) when requested.
//
// All arithmetic is fp64 with IEEE division.  A single-block chain (n < 2 Bp) degenerates to the reference's own
// recurrence on one lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csr {

struct BgPrm {
    int nchains, Bp, SB, NR;            // SB: rows allocated per 64-block group (>= longest block); NR: 1 or 2 columns
    int64_t NBk, NGk;                   // blocks, groups of 64 blocks
    double lam, lamF;
    const int64_t *chainOff, *chainLen, *chainFirstBlk, *chainNumBlk;
    const int4 *blk;                    // x: natural index of the block's first bin, y: length, z: chain, w: 1 = has separator
    const double *w, *rhs;              // natural, concatenated chains
    double *invd, *l1;                  // blocked (NGk*SB*64)
    double *X[6];                       // blocked solution columns: data [, ones], L0, L1, R0, R1 (slots NR..NR+3)
    double *T, *t;                      // per block: 16, 4*NR
    double *sepIn, *sepOut;             // reduced system, struct-of-arrays planes of NBk doubles (7+2NR in, 3+2NR out)
    double *sepG;                       // per block (its separator): x_S, 2*NR
    int64_t *badIdx;                    // per block: first modified pivot (chain-local index) or -1
    double *badVal;
    int64_t *chainBadIdx;               // per chain
    double *chainBadVal;
    double *out0, *out1;                // natural solutions (data, ones)
    double *chainMu;
    const unsigned char *active;        // per chain: 0 = leave this chain's solution untouched (IRLS lock-step); null = all
};

__device__ __forceinline__ int64_t bgidx(const BgPrm &p, int64_t kb, int s) {
    return (((kb >> 6) * (int64_t)p.SB + s) << 6) + (kb & 63);
}
// stencil of A for a chain of n bins (pyx:905-941)
__device__ __forceinline__ double bg_pen_diag(int64_t n, int64_t i, double lam, double lamF) {
    double v = 0.0;
    if (n >= 2 && lamF > 0.0) v += (i == 0 || i == n - 1) ? lamF : 2.0 * lamF;
    if (n >= 3 && lam > 0.0) {
        if (n == 3) v += (i == 1) ? 4.0 * lam : lam;
        else if (i == 0 || i == n - 1) v += lam;
        else if (i == 1 || i == n - 2) v += 5.0 * lam;
        else v += 6.0 * lam;
    }
    return v;
}
__device__ __forceinline__ double bg_off1(int64_t n, int64_t i, double lam, double lamF) {   // A[i, i+1]
    double v = (n >= 2 && lamF > 0.0) ? -lamF : 0.0;
    if (n >= 3 && lam > 0.0) v += (n == 3 || i == 0 || i == n - 2) ? -2.0 * lam : -4.0 * lam;
    return v;
}

template <int NR>
__global__ __launch_bounds__(64) void k_bg_local(BgPrm p) {
    const int64_t kb = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (kb >= p.NBk) return;
    const int4 bi = p.blk[kb];
    if (p.active && !p.active[bi.z]) return;
    const int64_t off = p.chainOff[bi.z], n = p.chainLen[bi.z];
    const int64_t a = (int64_t)bi.x - off;             // chain-local index of the first interior bin
    const int L = bi.y - (bi.w ? 2 : 0);               // interior length (>= 2 unless the chain itself is shorter)
    const bool hasL = a > 0, hasR = bi.w != 0;
    const double lam = p.lam, lamF = p.lamF, floor_ = 1.0e-12;
    const double *w = p.w + off, *r = p.rhs + off;
    int64_t bad = -1;
    double badv = 0.0;
    constexpr int NC = NR + 2;                         // dense columns of the forward pass: data[, ones], L0, L1
    double y1[NC], y2[NC];                             // y_{s-1}, y_{s-2} (before the division by d)
#pragma unroll
    for (int j = 0; j < NC; ++j) { y1[j] = 0.0; y2[j] = 0.0; }
    double d1 = 1.0, id1 = 1.0, id2 = 1.0, l1p = 0.0;  // d_{s-1}, 1/d_{s-1}, 1/d_{s-2}, l1_{s-1}
    const double cL0_0 = hasL ? lam : 0.0;                                   // A[a, a-2]
    const double cL1_0 = hasL ? bg_off1(n, a - 1, lam, lamF) : 0.0;          // A[a, a-1]
    const double cL1_1 = hasL ? lam : 0.0;                                   // A[a+1, a-1]
    // inputs are fetched PF steps ahead of the dependent recurrence (natural layout: a lane's reads are sequential)
    constexpr int PF = 8;
    double wq[PF], rq[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) { wq[u] = u < L ? w[a + u] : 0.0; rq[u] = u < L ? r[a + u] : 0.0; }
    for (int s0 = 0; s0 < L; s0 += PF) {
        double wc[PF], rc[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) { wc[u] = wq[u]; rc[u] = rq[u]; }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int sn = s0 + PF + u;
            wq[u] = sn < L ? w[a + sn] : 0.0;
            rq[u] = sn < L ? r[a + sn] : 0.0;
        }
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int s = s0 + u;
        if (s >= L) break;
        const int64_t i = a + s;
        double dg = wc[u] + bg_pen_diag(n, i, lam, lamF);
        if (dg < floor_) { if (bad < 0) { bad = i; badv = dg; } dg = floor_; }      // pyx:1031-1035
        double c[NC];
        c[0] = rc[u];
        if (NR == 2) c[1] = 1.0;
        c[NR] = s == 0 ? cL0_0 : 0.0;
        c[NR + 1] = s == 0 ? cL1_0 : (s == 1 ? cL1_1 : 0.0);
        // one IEEE division per step: the reciprocals of the two previous pivots are carried (the reference divides
        // four times per step; the difference is rounding-level and far below the system's own roundoff index)
        double l1 = 0.0, d = dg;
        if (s == 1) {
            l1 = bg_off1(n, i - 1, lam, lamF) * id1;                                // pyx:1042-1043
            d = dg - l1 * l1 * d1;
        } else if (s >= 2) {
            l1 = (bg_off1(n, i - 1, lam, lamF) - lam * l1p) * id1;                  // pyx:1052-1058
            d = dg - l1 * l1 * d1 - (lam * lam) * id2;
        }
        if (s >= 1 && d < floor_) { if (bad < 0) { bad = i; badv = d; } d = floor_; }
        const double l2 = s >= 2 ? lam * id2 : 0.0;
        const double id = 1.0 / d;
        const int64_t q = bgidx(p, kb, s);
        p.invd[q] = id;
        p.l1[q] = l1;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            double y = c[j];
            if (s >= 1) y -= l1 * y1[j];
            if (s >= 2) y -= l2 * y2[j];
            y2[j] = y1[j];
            y1[j] = y;
            p.X[j][q] = y * id;                                                      // pyx:1074-1076
        }
        d1 = d; id2 = id1; id1 = id; l1p = l1;
      }
    }
    // right coupling columns: nonzero only in the last two interior rows
    //   R0 (bin a+L):   A[a+L-2, a+L] = lam, A[a+L-1, a+L] = off1(a+L-1);   R1 (bin a+L+1): A[a+L-1, a+L+1] = lam
    const double offR = hasR ? bg_off1(n, a + L - 1, lam, lamF) : 0.0;
    const double lamR = hasR ? lam : 0.0;
    // ---- back substitution (pyx:1079-1084) of all columns; z = y/d is stored, x overwrites it
    constexpr int NX = NR + 4;
    double x1[NX], x2[NX];                             // x_{s+1}, x_{s+2}
    double xe[NX], xe1[NX];                            // x_{L-1}, x_{L-2}
#pragma unroll
    for (int j = 0; j < NX; ++j) { x1[j] = 0.0; x2[j] = 0.0; xe[j] = 0.0; xe1[j] = 0.0; }
    double l1n = 0.0;                                  // l1_{s+1}
    // row s-1 is fetched before row s is overwritten (the stores may alias the loads as far as the compiler knows)
    double nInvd = 0.0, nL1 = 0.0, nz[NC];
    if (L > 0) {
        const int64_t q0 = bgidx(p, kb, L - 1);
        nInvd = p.invd[q0]; nL1 = p.l1[q0];
#pragma unroll
        for (int j = 0; j < NC; ++j) nz[j] = p.X[j][q0];
    }
    for (int s = L - 1; s >= 0; --s) {
        const int64_t q = bgidx(p, kb, s);
        const double invd = nInvd, l1s = nL1;
        const double l2n = lam * invd;                 // l2_{s+2} = lam / d_s
        double z[NX];
#pragma unroll
        for (int j = 0; j < NC; ++j) z[j] = nz[j];
        if (s > 0) {
            const int64_t qm = q - 64;
            nInvd = p.invd[qm]; nL1 = p.l1[qm];
#pragma unroll
            for (int j = 0; j < NC; ++j) nz[j] = p.X[j][qm];
        }
        // forward-substituted right columns: y_{L-2} = c_{L-2}, y_{L-1} = c_{L-1} - l1_{L-1} y_{L-2}
        z[NR + 2] = 0.0; z[NR + 3] = 0.0;
        if (L >= 2) {
            if (s == L - 2) z[NR + 2] = lamR * invd;
            if (s == L - 1) { z[NR + 2] = (offR - l1s * lamR) * invd; z[NR + 3] = lamR * invd; }
        } else if (s == L - 1) {                       // one-bin interior (only in chains shorter than a block)
            z[NR + 2] = offR * invd; z[NR + 3] = lamR * invd;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            double x = z[j];
            if (s + 1 < L) x -= l1n * x1[j];
            if (s + 2 < L) x -= l2n * x2[j];
            x2[j] = x1[j];
            x1[j] = x;
            p.X[j][q] = x;
            if (s == L - 1) xe[j] = x;
            if (s == L - 2) xe1[j] = x;
        }
        l1n = l1s;
    }
    // x1 = x_0, x2 = x_1 (if L >= 2).  Schur contributions: rows of C' = [L0, L1, R0, R1]
    double *T = p.T + kb * 16, *t = p.t + kb * 4 * NR;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const double X0 = x1[j], X1 = L >= 2 ? x2[j] : 0.0, Xe = xe[j], Xe1 = L >= 2 ? xe1[j] : 0.0;
        const double rowL0 = cL0_0 * X0;
        const double rowL1 = cL1_0 * X0 + cL1_1 * X1;
        const double rowR0 = lamR * Xe1 + offR * Xe;
        const double rowR1 = lamR * Xe;
        if (j < NR) {
            t[j * 4 + 0] = rowL0; t[j * 4 + 1] = rowL1; t[j * 4 + 2] = rowR0; t[j * 4 + 3] = rowR1;
        } else {
            const int cidx = j - NR;
            T[0 * 4 + cidx] = rowL0; T[1 * 4 + cidx] = rowL1; T[2 * 4 + cidx] = rowR0; T[3 * 4 + cidx] = rowR1;
        }
    }
    p.badIdx[kb] = bad;
    p.badVal[kb] = badv;
}

// Reduced system, step 1 (parallel, one thread per separator): assemble the 2x2 diagonal block D_s, the coupling
// U_{s-1} to the previous separator (through interior s) and the right-hand side f_s into struct-of-array form.
//   D_s = A[S_s,S_s] - T_s[R,R] - T_{s+1}[L,L];  U_{s-1} = -T_s[L,R];  f_s = r[S_s] - t_s[R] - t_{s+1}[L]
template <int NR>
__global__ __launch_bounds__(256) void k_bg_sep_assemble(BgPrm p) {
    const int64_t kb = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (kb >= p.NBk) return;
    const int4 bi = p.blk[kb];
    if (!bi.w) return;                                  // the chain's last block has no separator
    if (p.active && !p.active[bi.z]) return;
    const int64_t off = p.chainOff[bi.z], n = p.chainLen[bi.z];
    const int64_t b = (int64_t)bi.x - off + bi.y - 2;   // chain-local index of the separator's first bin
    const double lam = p.lam, lamF = p.lamF, floor_ = 1.0e-12;
    const double *w = p.w + off, *r = p.rhs + off;
    const double *Tk = p.T + kb * 16, *Tn = p.T + (kb + 1) * 16;
    double dg0 = w[b] + bg_pen_diag(n, b, lam, lamF), dg1 = w[b + 1] + bg_pen_diag(n, b + 1, lam, lamF);
    int64_t bad = p.badIdx[kb];
    double badv = p.badVal[kb];
    if (dg0 < floor_) { if (bad < 0) { bad = b; badv = dg0; } dg0 = floor_; }
    if (dg1 < floor_) { if (bad < 0) { bad = b + 1; badv = dg1; } dg1 = floor_; }
    p.badIdx[kb] = bad;
    p.badVal[kb] = badv;
    const int64_t NB = p.NBk;
    double *S = p.sepIn;                                // SoA: 7 + 2 NR planes of NBk doubles
    S[0 * NB + kb] = dg0 - Tk[2 * 4 + 2] - Tn[0 * 4 + 0];
    S[1 * NB + kb] = bg_off1(n, b, lam, lamF) - Tk[2 * 4 + 3] - Tn[0 * 4 + 1];
    S[2 * NB + kb] = dg1 - Tk[3 * 4 + 3] - Tn[1 * 4 + 1];
    S[3 * NB + kb] = -Tk[0 * 4 + 2]; S[4 * NB + kb] = -Tk[0 * 4 + 3];
    S[5 * NB + kb] = -Tk[1 * 4 + 2]; S[6 * NB + kb] = -Tk[1 * 4 + 3];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double *tk = p.t + kb * 4 * NR + j * 4, *tn = p.t + (kb + 1) * 4 * NR + j * 4;
        S[(7 + 2 * j) * NB + kb] = (j == 0 ? r[b] : 1.0) - tk[2] - tn[0];
        S[(8 + 2 * j) * NB + kb] = (j == 0 ? r[b + 1] : 1.0) - tk[3] - tn[1];
    }
}

// Reduced system, step 2: block-tridiagonal (2x2 blocks) elimination over the separators of one chain.  One wavefront
// per chain: the lanes stage 64 separators at a time through LDS (coalesced loads / stores), every lane then runs the
// same dependent recurrence on broadcast LDS reads, so the serial chain never waits on global memory.
template <int NR>
__global__ __launch_bounds__(64) void k_bg_reduced(BgPrm p) {
    constexpr int NI = 7 + 2 * NR, NO = 3 + 2 * NR;
    __shared__ double sin_[NI][64];
    __shared__ double sout[NO][64];
    const int c = blockIdx.x, lane = threadIdx.x;
    if (p.active && !p.active[c]) return;
    const int64_t kb0 = p.chainFirstBlk[c], K = p.chainNumBlk[c], nsep = K - 1, NB = p.NBk;
    const double floor_ = 1.0e-12;
    // first modified pivot of the local factorisations / separator diagonals (chain order)
    int64_t bad = -1;
    double badv = 0.0;
    for (int64_t k0 = 0; k0 < K && bad < 0; k0 += 64) {
        const int64_t k = k0 + lane;
        const int64_t bi_ = k < K ? p.badIdx[kb0 + k] : -1;
        const unsigned long long m = __ballot(bi_ >= 0);
        if (m) {
            const int first = __ffsll((long long)m) - 1;
            bad = __shfl(bi_, first);
            badv = __shfl(k < K ? p.badVal[kb0 + k] : 0.0, first);
        }
    }
    double Mi00 = 0, Mi01 = 0, Mi11 = 0, gp[2 * NR];
#pragma unroll
    for (int j = 0; j < 2 * NR; ++j) gp[j] = 0.0;
    const double *S = p.sepIn;
    double *O = p.sepOut;                               // SoA: Minv (3) + g (2 NR) planes
    for (int64_t base = 0; base < nsep; base += 64) {
        const int cnt = (int)(nsep - base < 64 ? nsep - base : 64);
        if (lane < cnt) {
#pragma unroll
            for (int q = 0; q < NI; ++q) sin_[q][lane] = S[q * NB + kb0 + base + lane];
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            double M00 = sin_[0][j], M01 = sin_[1][j], M11 = sin_[2][j];
            double g[2 * NR];
#pragma unroll
            for (int q = 0; q < 2 * NR; ++q) g[q] = sin_[7 + q][j];
            if (base + j > 0) {
                const double U0 = sin_[3][j], U1 = sin_[4][j], U2 = sin_[5][j], U3 = sin_[6][j];
                const double a00 = Mi00 * U0 + Mi01 * U2, a01 = Mi00 * U1 + Mi01 * U3;
                const double a10 = Mi01 * U0 + Mi11 * U2, a11 = Mi01 * U1 + Mi11 * U3;
                M00 -= U0 * a00 + U2 * a10;
                M01 -= U0 * a01 + U2 * a11;
                M11 -= U1 * a01 + U3 * a11;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const double h0 = Mi00 * gp[2 * q] + Mi01 * gp[2 * q + 1], h1 = Mi01 * gp[2 * q] + Mi11 * gp[2 * q + 1];
                    g[2 * q] -= U0 * h0 + U2 * h1;
                    g[2 * q + 1] -= U1 * h0 + U3 * h1;
                }
            }
            double p0 = M00;                            // 2x2 LDL' pivots with the reference's floor
            if (p0 < floor_) { if (bad < 0) { bad = -2 - (base + j) * 2; badv = p0; } p0 = floor_; }
            const double l = M01 / p0;
            double p1 = M11 - l * l * p0;
            if (p1 < floor_) { if (bad < 0) { bad = -2 - ((base + j) * 2 + 1); badv = p1; } p1 = floor_; }
            Mi11 = 1.0 / p1;
            Mi01 = -l * Mi11;
            Mi00 = 1.0 / p0 + l * l * Mi11;
            if (lane == j) {
                sout[0][j] = Mi00; sout[1][j] = Mi01; sout[2][j] = Mi11;
#pragma unroll
                for (int q = 0; q < 2 * NR; ++q) sout[3 + q][j] = g[q];
            }
#pragma unroll
            for (int q = 0; q < 2 * NR; ++q) gp[q] = g[q];
        }
        __syncthreads();
        if (lane < cnt) {
#pragma unroll
            for (int q = 0; q < NO; ++q) O[q * NB + kb0 + base + lane] = sout[q][lane];
        }
        __syncthreads();
    }
    // back substitution: x_s = Minv_s (g_s - U_s x_{s+1}),  U_s = coupling stored with separator s+1
    double xn[2 * NR];
#pragma unroll
    for (int j = 0; j < 2 * NR; ++j) xn[j] = 0.0;
    for (int64_t hi = nsep; hi > 0; hi -= 64) {
        const int64_t base = hi >= 64 ? hi - 64 : 0;
        const int cnt = (int)(hi - base);
        if (lane < cnt) {
            const int64_t sI = kb0 + base + lane;
#pragma unroll
            for (int q = 0; q < NO; ++q) sout[q][lane] = O[q * NB + sI];
            const bool hasNext = base + lane + 1 < nsep;
#pragma unroll
            for (int q = 0; q < 4; ++q) sin_[q][lane] = hasNext ? S[(3 + q) * NB + sI + 1] : 0.0;
        }
        __syncthreads();
        for (int j = cnt - 1; j >= 0; --j) {
            const double m00 = sout[0][j], m01 = sout[1][j], m11 = sout[2][j];
            const double u0 = sin_[0][j], u1 = sin_[1][j], u2 = sin_[2][j], u3 = sin_[3][j];
            double xs[2 * NR];
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const double g0 = sout[3 + 2 * q][j] - (u0 * xn[2 * q] + u1 * xn[2 * q + 1]);
                const double g1 = sout[4 + 2 * q][j] - (u2 * xn[2 * q] + u3 * xn[2 * q + 1]);
                xs[2 * q] = m00 * g0 + m01 * g1;
                xs[2 * q + 1] = m01 * g0 + m11 * g1;
            }
#pragma unroll
            for (int q = 0; q < 2 * NR; ++q) xn[q] = xs[q];
            if (lane == j) {
#pragma unroll
                for (int q = 0; q < 2 * NR; ++q) sin_[4 + q][j] = xs[q];
            }
        }
        __syncthreads();
        if (lane < cnt) {
#pragma unroll
            for (int q = 0; q < 2 * NR; ++q) p.sepG[(kb0 + base + lane) * 2 * NR + q] = sin_[4 + q][lane];
        }
        __syncthreads();
    }
    if (lane == 0) {
        if (bad <= -2) {                                // a Schur pivot: report the separator bin it belongs to
            const int64_t e = -2 - bad, sidx = e >> 1;
            const int4 bi = p.blk[kb0 + sidx];
            bad = (int64_t)bi.x - p.chainOff[c] + bi.y - 2 + (e & 1);
        }
        p.chainBadIdx[c] = bad;
        p.chainBadVal[c] = badv;
    }
}

// x_I = A_II^-1 r - (A_II^-1 C) x_S ; separators copy their own solution.  One thread per blocked slot.
template <int NR>
__global__ __launch_bounds__(256) void k_bg_combine(BgPrm p) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(slot & 63);
    const int64_t row = slot >> 6;
    const int64_t G = row / p.SB;
    const int s = (int)(row - G * p.SB);
    const int64_t kb = G * 64 + lane;
    if (G >= p.NGk || kb >= p.NBk) return;
    const int4 bi = p.blk[kb];
    if (s >= bi.y) return;
    if (p.active && !p.active[bi.z]) return;
    const int L = bi.y - (bi.w ? 2 : 0);
    const int64_t g = (int64_t)bi.x + s;
    const bool hasL = (int64_t)bi.x > p.chainOff[bi.z];
    double *outs[2] = {p.out0, p.out1};
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        double x;
        if (s < L) {
            const int64_t q = bgidx(p, kb, s);
            x = p.X[j][q];
            if (hasL) {
                const double *xs = p.sepG + (kb - 1) * 2 * NR + 2 * j;
                x -= p.X[NR][q] * xs[0] + p.X[NR + 1][q] * xs[1];
            }
            if (bi.w) {
                const double *xs = p.sepG + kb * 2 * NR + 2 * j;
                x -= p.X[NR + 2][q] * xs[0] + p.X[NR + 3][q] * xs[1];
            }
        } else {
            x = p.sepG[kb * 2 * NR + 2 * j + (s - L)];
        }
        outs[j][g] = x;
    }
}

// zero-sum Lagrange correction (pyx:1086-1096): one workgroup per chain, fixed-shape reduction
__global__ __launch_bounds__(1024) void k_bg_center(BgPrm p) {
    __shared__ double sr[1024], sc[1024];
    const int c = blockIdx.x;
    if (p.active && !p.active[c]) return;
    const int64_t off = p.chainOff[c], n = p.chainLen[c];
    double ar = 0.0, ac = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) { ar += p.out0[off + i]; ac += p.out1[off + i]; }
    sr[threadIdx.x] = ar; sc[threadIdx.x] = ac;
    __syncthreads();
    for (int wd = 512; wd > 0; wd >>= 1) {
        if ((int)threadIdx.x < wd) { sr[threadIdx.x] += sr[threadIdx.x + wd]; sc[threadIdx.x] += sc[threadIdx.x + wd]; }
        __syncthreads();
    }
    const double mu = fabs(sc[0]) > 1.0e-12 ? sr[0] / sc[0] : sr[0] / (double)n;
    for (int64_t i = threadIdx.x; i < n; i += 1024) p.out0[off + i] -= mu * p.out1[off + i];
    if (threadIdx.x == 0) p.chainMu[c] = mu;
}

// pyx:9700-9724: weight = sum_j invVar, rhs = sum_j invVar * resid (fp64 over float32 matrices), natural (m, n)
__global__ __launch_bounds__(256) void k_bg_weighted_stats(int64_t m, int64_t n, const float *resid, const float *inv,
                                                           double *weight, double *rhs, unsigned long long *support) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool pos = false;
    if (i < n) {
        double ws = 0.0, rs = 0.0;
        for (int64_t j = 0; j < m; ++j) {
            const double w = (double)inv[j * n + i];
            ws += w;
            rs += w * (double)resid[j * n + i];
        }
        weight[i] = ws;
        rhs[i] = rs;
        pos = ws > 0.0;
    }
    const unsigned long long bal = __ballot(pos);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(support, (unsigned long long)__popcll(bal));
}


// ---------------------------------------------------------------------------------------------------------------
// device-resident background update (core.py:5064-5083, 8085-8378) on a csr batch: inputs never leave HBM
// ---------------------------------------------------------------------------------------------------------------
struct BgBatch {
    const int *groupChain;              // chain of every 64-bin group of the natural layout
    const int64_t *chainOff, *chainLen;
    int nchains, useLambda;
    float padf, wMinf, wMaxf;
    const float *lamNat;                // natural lambda (exported) or null
    const float *xsNat;                 // natural smoothed state (Npad, d), level = component 0; null = identically zero
    int xsStride;
    double *w, *rhs, *wAdj, *sol;
    unsigned long long *selAns;         // per chain x 2: order statistic being built bit by bit
    const long long *selRank;           // per chain x 2: 0-based rank among the positive weights, -1 = unused
    unsigned char *maskPrev, *maskNew;
    const float *bgCur;
    float *bgNext;
    const unsigned char *active;        // chains still iterating
    const double *pen;                  // per chain negative-penalty weight
    unsigned int *flags;                // per chain: bit0 any negative, bit1 mask changed, bit2 non-finite solution
    // per-chain reductions without atomics: every wavefront owns a run of <= BG_GPW 64-bin groups of ONE chain and writes
    // one partial record; a per-chain workgroup then folds its records in a fixed order (deterministic)
    const int *waveChain, *waveG0, *waveG1, *chainWave0, *chainWaveN;
    int NW, pad_;
    double *part;                       // NW x 4 partial records
    double *chainSum;                   // per chain: [sumW, support, sum w d^2, sum w next^2, sum w cur^2]
};

// weight / rhs per bin exactly as the reference forms them (core.py:5064-5083): float32 invVar = 1/max(munc+pad,1e-8)
// (* clip(lambda)), float32 residual = data - xs0 (ORIGINAL data: the residual contains the background), fp64 sums.
__global__ __launch_bounds__(256) void k_bg_batch_stats(Prm p, BgBatch a) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= p.Npad) return;
    const int c = a.groupChain[g >> 6];
    double ws = 0.0, rs = 0.0;
    if (c >= 0 && g - a.chainOff[c] < a.chainLen[c]) {
        float lam = 1.f;
        if (a.useLambda) lam = fminf(fmaxf(a.lamNat[g], a.wMinf), a.wMaxf);
        const float xs0 = a.xsNat ? a.xsNat[g * a.xsStride] : 0.0f;      // null: background warm start (core.py:2857)
        for (int j = 0; j < p.m; ++j) {
            float iv = __fdiv_rn(1.0f, fmaxf(p.munc[(int64_t)j * p.Npad + g] + a.padf, 1.0e-8f));
            if (a.useLambda) iv *= lam;
            const float res = p.data[(int64_t)j * p.Npad + g] - xs0;
            ws += (double)iv;
            rs += (double)iv * (double)res;
        }
    }
    a.w[g] = ws;
    a.rhs[g] = rs;
}

// the two per-phase diagnostics of `runConsenrich` that read the (m, n) matrices, per bin of ONE chain:
//   rel = level - inverse-variance weighted mean of the background-adjusted observations, the reference's row-by-row float64
//         accumulation (core.py:2663-2697: finite data, finite positive munc + pad, w = 1 / max(munc + pad, 1e-12)); NaN where no
//         observation counts;
//   fit = sum_j iv (r - g)^2, cnt = #cells with finite r and finite positive iv -- the matrices of the background update
//         (float32 iv = 1 / max(munc + pad, 1e-8) (* clip(lambda)), float32 r = data - level; core.py:5064-5076) against the
//         PROPOSAL g in float64 (core.py:4546-4552, 4587-4596).
struct PhaseArgs {
    int64_t off, len, Npad;
    int m, xsStride, useLambda, withFit;
    const float *data, *munc, *xsNat, *lamNat, *bgCur, *bgNext;
    float padf, wMinf, wMaxf;
    double pad;
    double *rel, *fit;
    int *cnt;
};
__global__ __launch_bounds__(256) void k_phase_tracks(PhaseArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.len) return;
    const int64_t g = a.off + i;
    const float lvl = a.xsNat[g * a.xsStride];
    const double x = (double)lvl, bg = a.bgCur ? (double)a.bgCur[g] : 0.0;
    const bool xok = isfinite(x);
    float lam = 1.f;
    if (a.useLambda) lam = fminf(fmaxf(a.lamNat[g], a.wMinf), a.wMaxf);
    const double gn = a.withFit ? (double)a.bgNext[g] : 0.0;
    double tot = 0.0, ws = 0.0, f = 0.0;
    int cn = 0;
    for (int j = 0; j < a.m; ++j) {
        const float d32 = a.data[(int64_t)j * a.Npad + g], v32 = a.munc[(int64_t)j * a.Npad + g];
        const double row = (double)d32, den = (double)v32 + a.pad;
        if (xok && isfinite(row) && isfinite(den) && den > 0.0) {
            const double w = 1.0 / fmax(den, 1.0e-12);
            tot += (row - bg) * w;
            ws += w;
        }
        if (a.withFit) {
            float iv = __fdiv_rn(1.0f, fmaxf(v32 + a.padf, 1.0e-8f));
            if (a.useLambda) iv *= lam;
            const float res = d32 - lvl;
            const double iv64 = (double)iv, fr = (double)res - gn;
            f += iv64 * fr * fr;
            cn += (isfinite(res) && isfinite(iv) && iv > 0.0f) ? 1 : 0;
        }
    }
    a.rel[i] = ws > 0.0 ? x - tot / ws : __longlong_as_double(0x7ff8000000000000ll);
    if (a.withFit) { a.fit[i] = f; a.cnt[i] = cn; }
}

constexpr int BG_GPW = 32;           // 64-bin groups per wavefront record

// mode 0: maskPrev = (current background < 0)  (initialBackground, core.py:8306-8316)
// mode 2: maskPrev = maskNew for the chains that iterate on
__global__ __launch_bounds__(256) void k_bg_mask(Prm p, BgBatch a, int mode) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= p.Npad) return;
    const int c = a.groupChain[g >> 6];
    if (c < 0 || g - a.chainOff[c] >= a.chainLen[c]) return;
    if (mode == 0) { a.maskPrev[g] = (a.bgCur != nullptr && (double)a.bgCur[g] < 0.0) ? 1 : 0; return; }
    if (a.active[c]) a.maskPrev[g] = a.maskNew[g];
}

// One pass over the bins with per-wavefront partial records (no atomics).  what:
//   0  sums:    part = [sum w, #(w > 0)]                                        (core.py:8148, 8157-8160)
//   1  select:  part = [#(key <= trial0), #(key <= trial1)] for bit `bit`      (median of the positive weights)
//   2  mask:    maskNew = (solution < 0); part[0] = OR of {any negative, mask changed, non-finite} (core.py:8331-8340)
//   3  finish:  bgNext = float32(solution) (0 without support); part[0] = sum w (next - current)^2 (core.py:5199-5215)
__global__ __launch_bounds__(256) void k_bg_wave_pass(Prm p, BgBatch a, int what, int bit, const unsigned char *hasSupport) {
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= a.NW) return;
    const int c = a.waveChain[wv];
    const int64_t off = a.chainOff[c], len = a.chainLen[c];
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    unsigned int fl = 0u;
    bool run = true;
    unsigned long long t0 = 0, t1 = 0;
    if (what == 1) {
        run = a.selRank[2 * c] >= 0;
        const unsigned long long low = (1ull << bit) - 1ull;
        t0 = a.selAns[2 * c] | low; t1 = a.selAns[2 * c + 1] | low;
    }
    if (what == 2) run = a.active[c] != 0;
    const bool sup = what == 3 ? hasSupport[c] != 0 : true;
    if (run) {
        for (int G = a.waveG0[wv]; G < a.waveG1[wv]; ++G) {
            const int64_t g = ((int64_t)G << 6) + lane;
            const bool in = g - off < len;
            if (what == 0) {
                const double w = in ? a.w[g] : 0.0;
                r0 += w;
                r1 += w > 0.0 ? 1.0 : 0.0;
            } else if (what == 1) {
                const double w = in ? a.w[g] : 0.0;
                const bool ok = w > 0.0 && isfinite(w);
                const unsigned long long key = (unsigned long long)__double_as_longlong(w);
                r0 += (double)__popcll(__ballot(ok && key <= t0));
                r1 += (double)__popcll(__ballot(ok && key <= t1));
            } else if (what == 2) {
                if (in) {
                    const double x = a.sol[g];
                    const unsigned char neg = x < 0.0 ? 1 : 0;
                    a.maskNew[g] = neg;
                    fl |= (neg ? 1u : 0u) | ((neg != a.maskPrev[g]) ? 2u : 0u) | (isfinite(x) ? 0u : 4u);
                }
            } else if (in) {
                const float nx = sup ? (float)a.sol[g] : 0.f;
                a.bgNext[g] = nx;
                const double cur = a.bgCur ? (double)a.bgCur[g] : 0.0, dlt = (double)nx - cur, w = a.w[g];
                r0 += w * dlt * dlt;                     // core.py:5209-5240: shift, proposal and reference RMS numerators
                r1 += w * (double)nx * (double)nx;
                r2 += w * cur * cur;
            }
        }
    }
    double *rec = a.part + 4 * (int64_t)wv;
    if (what == 1) {                                    // ballot counts are already wave totals
        if (lane == 0) { rec[0] = r0; rec[1] = r1; }
        return;
    }
    if (what == 2) {
        for (int o = 32; o > 0; o >>= 1) fl |= __shfl_xor(fl, o);
        if (lane == 0) rec[0] = (double)fl;
        return;
    }
    for (int o = 32; o > 0; o >>= 1) {                  // fixed butterfly order
        r0 += __shfl_xor(r0, o); r1 += __shfl_xor(r1, o); r2 += __shfl_xor(r2, o);
    }
    if (lane == 0) { rec[0] = r0; rec[1] = r1; rec[2] = r2; }
}

// fold the records of every chain (one wavefront per chain, fixed order) and act on the totals
__global__ __launch_bounds__(64) void k_bg_wave_fold(BgBatch a, int what, int bit) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int w0 = a.chainWave0[c], nw = a.chainWaveN[c];
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    unsigned int fl = 0u;
    for (int i = lane; i < nw; i += 64) {
        const double *rec = a.part + 4 * (int64_t)(w0 + i);
        if (what == 2) fl |= (unsigned int)rec[0];
        else { r0 += rec[0]; r1 += rec[1]; if (what == 3) r2 += rec[2]; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        r0 += __shfl_xor(r0, o); r1 += __shfl_xor(r1, o); r2 += __shfl_xor(r2, o); fl |= __shfl_xor(fl, o);
    }
    if (lane != 0) return;
    if (what == 0) { a.chainSum[c * 5 + 0] = r0; a.chainSum[c * 5 + 1] = r1; }
    else if (what == 1) {
        // bit b of the k-th smallest key is 0 iff at least k+1 keys are <= (prefix | all-ones below b)
        const long long k0 = a.selRank[2 * c], k1 = a.selRank[2 * c + 1];
        if (k0 >= 0 && r0 < (double)(k0 + 1)) a.selAns[2 * c] |= (1ull << bit);
        if (k1 >= 0 && r1 < (double)(k1 + 1)) a.selAns[2 * c + 1] |= (1ull << bit);
    } else if (what == 2) a.flags[c] = fl;
    else { a.chainSum[c * 5 + 2] = r0; a.chainSum[c * 5 + 3] = r1; a.chainSum[c * 5 + 4] = r2; }
}

// adjusted weights of the asymmetric IRLS (core.py:8312-8313, 8342-8343)
__global__ __launch_bounds__(256) void k_bg_adjust(Prm p, BgBatch a, int useMask) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= p.Npad) return;
    const int c = a.groupChain[g >> 6];
    if (c < 0 || !a.active[c]) { if (c < 0) a.wAdj[g] = 0.0; return; }
    double w = a.w[g];
    if (useMask && a.maskPrev[g]) w += a.pen[c];
    a.wAdj[g] = w;
}

}  // namespace csr
