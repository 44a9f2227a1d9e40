// csr_device.h -- gfx950 device code for the Consenrich estimator hot path (forward Kalman filter, RTS smoother,
// ECM precision re-weighting).  Hand-written HIP for CDNA4; compiled with -ffp-contract=off so that every fused
// multiply-add below is an explicit fma(): the arithmetic of a recurrence step is then identical in every kernel
// and template instance that uses it (speculative run, warm-up, fix-up, sequential fallback), which is what makes
// the bitwise carry validation meaningful.
//
// DATA LAYOUT
//   natural  : the reference's layouts -- data/munc (m, Npad) row-major over the concatenated chains,
//              per-bin outputs (Npad, ...) -- only at the API boundary.
//   blocked  : every per-bin intermediate lives in a block-transposed layout.  Each chain is cut into blocks of B
//              bins (blocks never span chains); block b = 64*G + l belongs to wave-group G, lane l; element (b, s)
//              is stored at ((G*B + s)*64 + l).  A wavefront that walks 64 consecutive blocks in lock-step (one
//              block per lane, step s) therefore touches one fully coalesced 64-element row per step -- the serial
//              recurrences read/write HBM at full line efficiency with no LDS staging.
//
// TIME-PARALLEL RECURRENCES (SURVEY.md section 7 "hard parts")
//   The reference recursions are strictly sequential in k and round their carries to float32 every step
//   (pyx:405-406, 427-430, 478-479, 492-495).  A re-associated scan cannot reproduce that.  Instead every block
//   runs the exact rounded recursion speculatively: it starts W bins early from a cold prior, discards the warm-up,
//   and records the carry it entered its own range with.  A validation kernel then compares, bit for bit, each
//   block's carry-in with its predecessor's carry-out and re-runs exactly the mismatching blocks from the true
//   carry (iterated to a fixed point on the host), so the result equals the sequential recursion regardless of W.
//   The forward pass is split into a covariance chain (independent of the data, short memory) and a state chain
//   (affine in the state given the gains, 16 flops/step), and the per-bin NIS/NLL terms are evaluated afterwards by
//   a fully parallel kernel from the stored predicted covariance and the previous filtered state.
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include <stdint.h>

namespace csr {

// ---------------------------------------------------------------------------------------------------------------
// parameters shared by all kernels (passed by value)
// ---------------------------------------------------------------------------------------------------------------
struct Prm {
    int d;              // state dimension (2 = levelTrend, 1 = level)
    int B;              // block length (bins), multiple of 32
    int m;              // samples
    int nchains;
    int64_t NB;         // total blocks
    int64_t NG;         // wave-groups = ceil(NB/64)
    int64_t Npad;       // natural row stride (bins), chains start at multiples of 64
    double F00, F01, F10, F11;
    double Q00, Q01, Q10, Q11;
    double init, cinit, pad;
    double wMin, wMax, kMin, kMax;
    double apnMinQ, apnMaxQ, apnThresh, apnScale, apnPC, qDiag;
    double nu;          // ECM robust-t degrees of freedom
    uint32_t flags;     // CSR_* bits
    int warm;           // warm-up length in blocks for the kernel being launched
    int debugForce;     // debugging aid: validation treats every carry as mismatching
    int predCompact;    // fused forward chain: only the NIS epilogue reads the gain record, and only P00pred of it --
                        // store that float (tPP, 4 B/bin) instead of the 16-byte record
    float *tPP;
    int estepKappa;     // ECM: the smoother's main phase also evaluates the kappa E-step of the transition it just
                        // smoothed (pyx:8244-8298): it holds the moments of bins k and k+1 and the lag covariance
    int storeMoments;   // 0: inner ECM sweeps whose smoothed moments nobody reads are not stored at all
    int natOut;         // smoother (levelTrend): 1 = write xs / Ps / lag straight into the reference-layout arrays below
    float *natXs, *natPs, *natLag;
    float *natD;        // NIS/NLL epilogue: non-null = write D in the reference layout (through an LDS tile) instead of tD
    const float *bg;    // natural (Npad) current background, subtracted from the data in float32 (core.py:3253); may be null
    int qFromMult;      // smoother: 1 = process noise is the constant float32(Q0) (internal forward pass without
                        //           kappa / qScale / APN), 0 = read the stored / imported pNoise array tQ
    int qFromKappa;     // ECM sweeps with a DIAGONAL base process noise (the reference's default, core.py:4198-4205): the
                        // forward pass stores only the two diagonal float32 entries of pNoise (tQ2, 8 B instead of 16; the
                        // off-diagonal ones are float32(qf * 0) = 0) and the smoother reads them through the *Q2 policies.
                        // (Rebuilding Q in the smoother from kappa, or from the stored scalar qScale / kappa, saves more bytes
                        // but was measured SLOWER: 0.28 -> 0.35 / 0.32 ms per sweep -- that chain is bound by instruction
                        // issue, an IEEE division or even four multiply + convert pairs per step cost more than the bytes.)
    float2 *tQ2;
    int storePP;        // fused forward chain: 0 = the predicted level variance is not stored (no NIS/NLL epilogue follows)
    int xTolUlps;       // forward state chain validation: 0 = bitwise, k = accept a carry-in within k float32 ulps

    // block table: x = natural index of first bin, y = length, z = first block of chain, w = last block of chain
    const int4 *blk;
    const int *blkChain;
    const double *chainQ;   // per chain: Q00 Q01 Q10 Q11 (row-major base process noise) or null = the model's Q0 for all
    const unsigned char *chainActive;   // nullptr = all chains active

    // natural inputs
    const float *data;
    const float *munc;

    // blocked per-bin statistics (a1)
    double2 *tSZ;       // .x = S0u = sum_j 1/R_j, .y = zbar = weighted mean of z: one 16-byte record per bin (one load /
                        // one LDS-DMA instruction per step of the serial chains instead of two / four)
    double *tS2c;       // sum_j (z_j - zbar)^2 / R_j
    double *tLogR;      // sum_j log R_j
    // 2-ulp throughput mode: the two statistics only the NIS / NLL terms read travel as ONE float32 pair {S2c, log R} (8 B per
    // bin written and read instead of 16 + 16; statsF32 says which form the resident statistics have -- load_s2l() reads either)
    float2 *tS2L;
    int statsF32;
    // fused forward chain with reference-layout outputs (2-ulp mode): the tile walker evaluates the NIS / NLL terms of the bins it
    // filters (it holds S0, zbar - xpred and 1 + P00pred S0 in registers), writes D in the reference layout through its LDS tile
    // and the block's partial sums -- no epilogue kernel, no predicted-variance track, no second read of the statistics
    int nisInChain;
    double rM;          // 1 / m
    // ... and when the process noise is one constant matrix its tile walker writes xf / Pf ONLY in the reference layout (natOnly: no
    // blocked copies, 24 B per bin less); the smoother then reads them THERE through its LDS tiles (natIn, k_smooth_natin)
    int natOnly;
    int natIn;
    const float2 *natXfIn;
    const float4 *natPfIn;
    double2 *natSZ;     // statistics kernel: non-null = the {S0u, zbar} records ALSO in the reference layout (where the superblock
                        // state chain reads them: no conversion launch in front of it)
    // blocked multipliers
    float *tLam, *tKap, *tQs;
    float *tKapOut;     // where the smoother's fused kappa E-step writes (the ECM loop ping-pongs scratch buffers so that a
                        // sweep never overwrites the kappa its own forward pass was run with; == tKap outside that loop)
    // forward covariance chain outputs
    // gain record of the forward covariance chain, one 16-byte element per bin (one load / one LDS-DMA per step):
    //   trend: { double gs = S0/innovScale ; float P00pred ; float P10pred }
    //   level: { double gs ; double Ppred }  (the level model keeps its carries in double)
    float4 *tXin;
    float4 *tPf;        // filtered covariance (trend: 00,01,10,11; level: .x)
    float4 *tQ;         // process noise used for the transition k -> k+1, stored at k (== pNoiseForward[k])
    // forward state chain outputs
    float2 *tXf;        // filtered state (level: .x)
    double *tXd;        // level: filtered state in double (needed to rebuild the unrounded prediction)
    float *tD;          // NIS or NLL per bin
    // backward
    float2 *tXs;
    float4 *tPs;
    float4 *tLag;
    // per-block partial sums (deterministic reductions)
    double *blkSumD, *blkSumNLL;
    double *chainSumD, *chainSumNLL;
    // speculation bookkeeping (raw bytes, sized for the largest carry)
    void *carryIn;      // carry each block actually started from
    void *carryOutA;    // ping
    void *carryOutB;    // pong
    unsigned int *rerunCount;
    unsigned int *rerunCountPass;   // counter of THIS validation pass (deferred validation launches 1-4 passes and checks the last)
    // Folded validation: a clean stage launches no validation kernel of its own -- the NEXT speculative kernel of the stream
    // compares, in its prologue, the previous stage's carry-ins with its neighbours' carry-outs (check only, no repair: a
    // mismatch is counted and the pipeline replayed).  The kernel boundary in between makes the carries visible; the two
    // stages use different carry sets.  prevKind = CK_* of the previous stage's policy (0: nothing to check).
    // Warm-started speculation (ECM sweeps of small batches): consecutive sweeps differ only by the kappa of one E-step, so
    // the carry the PREVIOUS sweep held at the first bin of a block's window is a far better start than the cold prior, and
    // the window can be a fraction of the cold one.  ckptIn[b]: carry of the previous sweep's walk of block b at the point
    // where a window of p.warm bins for a later (forward) / earlier (smoother) block begins; ckptOut / ckptSaveWarm: where,
    // and for which window length, this sweep records its own.  Validation is unchanged (carry-in against the neighbour's
    // carry-out), so a poor start costs a re-run, never a wrong result.  Null pointers = cold start / nothing recorded.
    const void *ckptIn;
    void *ckptOut;
    int ckptSaveWarm;
    unsigned int *localFixCount;    // blocks repaired inside the speculative kernel (wave_local_repair): statistics only
    // superblock view of the bit-exact state chain (k_sb_state_*): widened {gs, zbar, P00pred, P10pred} records in the view's
    // blocking, and the index of a padding block behind the last group that lanes without a block of their own walk
    unsigned long long *sbDbg;      // CONSENRICH_AMD_SB_DEBUG: counters of the delta-form repair passes (blocks, batches, rounds, fallback batches, merged exits), else null
    int prevKind;
    const void *prevCarryIn, *prevCarryOut;
    unsigned int *prevCount, *prevCountPass;
    const unsigned char *prevActive;
};

enum : uint32_t {
    F_LAMBDA = 1u << 0, F_KAPPA = 1u << 1, F_QSCALE = 1u << 2, F_APN = 1u << 3, F_NLL = 1u << 4,
    F_NLL_IN_D = 1u << 5
};

// ---------------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double r32(double x) { return (double)(float)x; }   // the reference's <double><float32_t>

__device__ __forceinline__ double clampd(double v, double lo, double hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}

// fast reciprocal for the serial chains: v_rcp_f64 + two Newton steps (<= ~1 ulp; the division it replaces is the
// only multi-instruction operation on the dependent path of the forward covariance chain)
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    return r;
}

// Natural logarithm for the per-bin likelihood terms (k_fwd_dstat is bound by fp64 instruction issue, and the library
// log is ~100 of them): x = 2^e m, m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716; the odd
// series to s^21 leaves a truncation error below 2e-18.  Relative error ~2e-16 (1-2 ulp), also for x -> 1 (s is formed
// from the exactly representable m - 1).  Non-positive, non-finite and subnormal arguments take the library routine.
__device__ __forceinline__ double log_pos(double x) {
    if (!(x >= 2.2250738585072014e-308) || !(x <= 1.7976931348623157e308)) return log(x);
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752440) {
        m *= 2.0;
        e -= 1;
    }
    const double s = (m - 1.0) * rcp_nr(m + 1.0);
    const double z = s * s;
    double q = 1.0 / 21.0;
    q = fma(q, z, 1.0 / 19.0);
    q = fma(q, z, 1.0 / 17.0);
    q = fma(q, z, 1.0 / 15.0);
    q = fma(q, z, 1.0 / 13.0);
    q = fma(q, z, 1.0 / 11.0);
    q = fma(q, z, 1.0 / 9.0);
    q = fma(q, z, 1.0 / 7.0);
    q = fma(q, z, 1.0 / 5.0);
    q = fma(q, z, 1.0 / 3.0);
    const double lm = fma(2.0 * s * z, q, 2.0 * s);
    return fma((double)e, 0.693147180559945309417232121458, lm);
}

__device__ __forceinline__ int64_t tbase(int64_t b, int B) { return ((b >> 6) * (int64_t)B) * 64 + (b & 63); }
__device__ __forceinline__ int64_t tidx(int64_t b, int s, int B) { return tbase(b, B) + (int64_t)s * 64; }

__device__ __forceinline__ bool chain_on(const Prm &p, int64_t b) {
    return p.chainActive == nullptr || p.chainActive[p.blkChain[b]] != 0;
}

// Per-chain base process noise (csr_batch_set_chain_q: every chromosome is seeded with its own Q0, core.py:5667): the
// kernels that use Q work on a per-lane copy of the parameter block whose Q fields come from the lane's chain.  A lane
// only ever walks blocks of ONE chain (warm-up included), so the copy is made once per kernel.
__device__ __forceinline__ Prm chain_model(const Prm &p, int chain) {
    Prm q = p;
    if (p.chainQ != nullptr) {
        const double *t = p.chainQ + 4 * (int64_t)chain;
        q.Q00 = t[0]; q.Q01 = t[1]; q.Q10 = t[2]; q.Q11 = t[3];
        q.qDiag = 0.5 * (t[0] + t[3]);
    }
    return q;
}
__device__ __forceinline__ Prm lane_model(const Prm &p, int64_t b) {
    return chain_model(p, p.chainQ != nullptr ? p.blkChain[b < p.NB ? b : p.NB - 1] : 0);
}
// The serial chain kernels get the per-lane copy only in their PCQ instantiation (launched when a per-chain table is
// set): with it the Q fields live in vector registers (12 more VGPRs in the fused forward chain) and the constant-Q
// arithmetic the compiler otherwise folds into scalar code runs per lane; the default instantiation is the same
// machine code as without the feature (same register counts).
template <bool PCQ>
__device__ __forceinline__ Prm lane_model_if(const Prm &p, int64_t b) {
    if constexpr (PCQ) return lane_model(p, b);
    else return p;
}

__device__ __forceinline__ unsigned f2u(float f) { return __float_as_uint(f); }

// gain record packing (Prm::tXin)
__device__ __forceinline__ float4 pack_gain_trend(double gs, float c00p, float c10p) {
    const long long b = __double_as_longlong(gs);
    return make_float4(__uint_as_float((unsigned)(b & 0xffffffffll)), __uint_as_float((unsigned)((unsigned long long)b >> 32)),
                       c00p, c10p);
}
__device__ __forceinline__ float4 pack_gain_level(double gs, double pp) {
    const long long a = __double_as_longlong(gs), b = __double_as_longlong(pp);
    return make_float4(__uint_as_float((unsigned)(a & 0xffffffffll)), __uint_as_float((unsigned)((unsigned long long)a >> 32)),
                       __uint_as_float((unsigned)(b & 0xffffffffll)), __uint_as_float((unsigned)((unsigned long long)b >> 32)));
}
__device__ __forceinline__ double unpack_d(float lo, float hi) {
    return __longlong_as_double((long long)(((unsigned long long)__float_as_uint(hi) << 32) | __float_as_uint(lo)));
}

// LDS-DMA helpers (see k_chain_spec_dma)
typedef __attribute__((address_space(3))) void *lds_vptr;
typedef const __attribute__((address_space(1))) void *gbl_cvptr;

__device__ __forceinline__ void dma4(const void *g, unsigned *ldsRow) {
    __builtin_amdgcn_global_load_lds((gbl_cvptr)g, (lds_vptr)ldsRow, 4, 0, 0);
}
__device__ __forceinline__ void dma16(const void *g, unsigned *ldsRow) {
    __builtin_amdgcn_global_load_lds((gbl_cvptr)g, (lds_vptr)ldsRow, 16, 0, 0);
}
// N consecutive 1-KiB rows (64 lanes x 16 B) from g (this lane's address of row 0) into the LDS rows from ldsRow on: ONE address
// pair and ONE M0, the row index in the instruction's immediate offset (it is applied to the global and the LDS address alike;
// 13 bits signed: rows 0..3)
template <int... U>
__device__ __forceinline__ void dma16_rows(const void *g, unsigned *ldsRow, std::integer_sequence<int, U...>) {
    (__builtin_amdgcn_global_load_lds((gbl_cvptr)g, (lds_vptr)ldsRow, 16, U * 1024, 0), ...);
}
__device__ __forceinline__ unsigned lds_off(const unsigned *q) {
    return (unsigned)(size_t)((__attribute__((address_space(3))) const unsigned *)q);
}
__device__ __forceinline__ unsigned lds_rd32(const unsigned *q) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(lds_off(q)) : "memory");
    return v;
}
__device__ __forceinline__ uint4 lds_rd128(const unsigned *q) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_off(q)) : "memory");
    return v;
}
__device__ __forceinline__ double words2double(unsigned lo, unsigned hi) {
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

constexpr int DMA_L = 8;      // steps in flight
constexpr int DMA_R = 9;      // ring slots (> DMA_L: the slot a step refills is the one the previous step finished reading)


// ---------------------------------------------------------------------------------------------------------------
// a1  per-bin sufficient statistics  (pyx:259-282 collapsed about the weighted mean)
//     S0u = sum 1/R, zbar = sum z/R / S0u, S2c = sum (z-zbar)^2/R, logR = sum log R,  R = max(v + pad, 1e-12)
//     The reference's S1, S2 about the predicted level x follow exactly:  S1 = S0 (zbar - x),
//     S2 = S2c + S0 (zbar - x)^2  (no cancellation: both terms are non-negative).
// ---------------------------------------------------------------------------------------------------------------
struct BinStats {
    double s0, zbar, s2c, logr;
};

// One pass over the m samples (register-light, so 8 waves/SIMD hide the HBM latency): weighted sums about the
// pivot z_0 (first sample), then zbar = z_0 + A/S0 and S2c = B - A^2/S0.  The shift keeps the cancellation in B - A^2/S0
// at (z_0 - zbar)^2 / var, i.e. a relative error of ~1e-16 * that ratio -- >= 8 digits of headroom to the 1e-5 budget
// even for a 10^4-sigma pivot.  1/R uses v_rcp_f64 + two Newton steps (<= 1 ulp).
// UN = sample rows loaded together (2 * UN loads in flight per thread)
template <int UN = 8>
__device__ __forceinline__ BinStats bin_stats(const float *__restrict__ data, const float *__restrict__ munc,
                                              int64_t stride, int64_t g, int m, double pad, float bgv) {
    // bgv: background of this bin; z = data - background is formed in float32 like the reference's dataAdjusted
    const double piv = (double)(data[g] - bgv);
    double s0 = 0.0, A = 0.0, Bq = 0.0, mant = 1.0;
    int ex = 0;
    const float *dp = data + g, *mp = munc + g;
    int j = 0;
    for (; j + UN <= m; j += UN) {
        float z[UN], v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            z[u] = dp[(int64_t)(j + u) * stride] - bgv;
            v[u] = mp[(int64_t)(j + u) * stride];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            double R = (double)v[u] + pad;
            if (R < 1.0e-12) R = 1.0e-12;
            const double w = rcp_nr(R);
            const double dz = (double)z[u] - piv;
            s0 += w;
            A = fma(w, dz, A);
            Bq = fma(w * dz, dz, Bq);
            int e;
            mant *= frexp(R, &e);
            ex += e;
        }
        int e2;
        mant = frexp(mant, &e2);
        ex += e2;
    }
    for (; j < m; ++j) {
        double R = (double)mp[(int64_t)j * stride] + pad;
        if (R < 1.0e-12) R = 1.0e-12;
        const double w = rcp_nr(R);
        const double dz = (double)(dp[(int64_t)j * stride] - bgv) - piv;
        s0 += w;
        A = fma(w, dz, A);
        Bq = fma(w * dz, dz, Bq);
        int e;
        mant *= frexp(R, &e);
        ex += e;
    }
    BinStats o;
    o.s0 = s0;
    const double shift = (s0 > 0.0) ? A / s0 : 0.0;
    o.zbar = (s0 > 0.0) ? piv + shift : 0.0;
    double s2 = Bq - A * shift;
    o.s2c = s2 > 0.0 ? s2 : 0.0;
    o.logr = log(mant) + (double)ex * 0.693147180559945309417232121458;
    return o;
}

// Four consecutive bins per thread, 16-byte loads (same sums in the same order as bin_stats: results are bit-identical).
// A wave-instruction then moves 1 KiB instead of 256 B -- the statistics pass is a pure HBM stream, and the wide form is
// what the memory system sustains its copy rate with (MI355X_MICROARCH.md: ~10 B/cycle/CU for global_load_dwordx4).
template <int UN>
__device__ __forceinline__ void bin_stats4(const float *__restrict__ data, const float *__restrict__ munc, int64_t stride,
                                           int64_t g, int m, double pad, float4 bg, BinStats o[4]) {
    const float4 z0 = *reinterpret_cast<const float4 *>(data + g);
    const double piv[4] = {(double)(z0.x - bg.x), (double)(z0.y - bg.y), (double)(z0.z - bg.z), (double)(z0.w - bg.w)};
    const float bgv[4] = {bg.x, bg.y, bg.z, bg.w};
    double s0[4] = {0, 0, 0, 0}, A[4] = {0, 0, 0, 0}, Bq[4] = {0, 0, 0, 0}, mant[4] = {1, 1, 1, 1};
    int ex[4] = {0, 0, 0, 0};
    const float *dp = data + g, *mp = munc + g;
    // the frexp renormalisation of the running product happens at the same sample indices as in bin_stats<8>
    int j = 0;
    for (; j + UN <= m; j += UN) {
        float4 z[UN], v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            z[u] = *reinterpret_cast<const float4 *>(dp + (int64_t)(j + u) * stride);
            v[u] = *reinterpret_cast<const float4 *>(mp + (int64_t)(j + u) * stride);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float zz[4] = {z[u].x, z[u].y, z[u].z, z[u].w}, vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double R = (double)vv[q] + pad;
                if (R < 1.0e-12) R = 1.0e-12;
                const double w = rcp_nr(R);
                const double dz = (double)(zz[q] - bgv[q]) - piv[q];
                s0[q] += w;
                A[q] = fma(w, dz, A[q]);
                Bq[q] = fma(w * dz, dz, Bq[q]);
                int e;
                mant[q] *= frexp(R, &e);
                ex[q] += e;
            }
            if (((j + u) & 7) == 7) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int e2;
                    mant[q] = frexp(mant[q], &e2);
                    ex[q] += e2;
                }
            }
        }
    }
    for (; j < m; ++j) {
        const float4 z = *reinterpret_cast<const float4 *>(dp + (int64_t)j * stride);
        const float4 v = *reinterpret_cast<const float4 *>(mp + (int64_t)j * stride);
        const float zz[4] = {z.x, z.y, z.z, z.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double R = (double)vv[q] + pad;
            if (R < 1.0e-12) R = 1.0e-12;
            const double w = rcp_nr(R);
            const double dz = (double)(zz[q] - bgv[q]) - piv[q];
            s0[q] += w;
            A[q] = fma(w, dz, A[q]);
            Bq[q] = fma(w * dz, dz, Bq[q]);
            int e;
            mant[q] *= frexp(R, &e);
            ex[q] += e;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        o[q].s0 = s0[q];
        const double shift = (s0[q] > 0.0) ? A[q] / s0[q] : 0.0;
        o[q].zbar = (s0[q] > 0.0) ? piv[q] + shift : 0.0;
        const double s2 = Bq[q] - A[q] * shift;
        o[q].s2c = s2 > 0.0 ? s2 : 0.0;
        o[q].logr = log(mant[q]) + (double)ex[q] * 0.693147180559945309417232121458;
    }
}

// TS steps x TL = 1024/TS lanes per workgroup, one pass: thread t owns steps 4*(t % (TS/4)) .. +3 of lane t / (TS/4).
template <int TS, int UN = 4>
__global__ __launch_bounds__(256) void k_stats_v4(Prm p) {
    constexpr int TPB = TS / 4;             // threads per block row
    constexpr int TL = 256 / TPB;           // lanes (blocks) per tile
    constexpr int LT = 64 / TL;
    __shared__ double tile[4][TS][TL + 1];
    const int tilesPerGroup = (p.B / TS) * LT;
    const int64_t G = blockIdx.x / tilesPerGroup;
    const int rem = (int)(blockIdx.x % tilesPerGroup);
    const int s0 = (rem / LT) * TS;
    const int l0 = (rem % LT) * TL;
    const int t = threadIdx.x;
    const int ll = t / TPB, si = (t % TPB) * 4;
    const int64_t b = G * 64 + l0 + ll;
    BinStats o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = BinStats{0.0, 0.0, 0.0, 0.0};
    int valid = 0;
    if (b < p.NB) {
        const int4 bi = p.blk[b];
        if (s0 + si < bi.y && chain_on(p, b)) {
            const int64_t g = (int64_t)bi.x + s0 + si;     // 16-byte aligned; g + 3 stays inside the chain's 64-bin padding
            const float4 bg = p.bg ? *reinterpret_cast<const float4 *>(p.bg + g) : make_float4(0.f, 0.f, 0.f, 0.f);
            bin_stats4<UN>(p.data, p.munc, p.Npad, g, p.m, p.pad, bg, o);
            valid = bi.y - (s0 + si);
            if (p.natSZ != nullptr) {           // (the thread's four bins are consecutive there: 64 contiguous bytes)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < valid) p.natSZ[g + q] = make_double2(o[q].s0, o[q].zbar);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool ok = q < valid;
        tile[0][si + q][ll] = ok ? o[q].s0 : 0.0;
        tile[1][si + q][ll] = ok ? o[q].zbar : 0.0;
        tile[2][si + q][ll] = ok ? o[q].s2c : 0.0;
        tile[3][si + q][ll] = ok ? o[q].logr : 0.0;
    }
    __syncthreads();
    const int64_t rowBase = (G * (int64_t)p.B + s0) * 64 + l0;
    constexpr int NP = TS * TL / 256;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const int idx = it * 256 + t;
        const int row = idx / TL, l2 = idx % TL;
        const int64_t oo = rowBase + (int64_t)row * 64 + l2;
        p.tSZ[oo] = make_double2(tile[0][row][l2], tile[1][row][l2]);
        if (p.statsF32) {
            p.tS2L[oo] = make_float2((float)tile[2][row][l2], (float)tile[3][row][l2]);
        } else {
            p.tS2c[oo] = tile[2][row][l2];
            p.tLogR[oo] = tile[3][row][l2];
        }
    }
}

// {S2c, log R} of slot i in whichever form the resident statistics have (Prm::statsF32)
__device__ __forceinline__ double load_s2c(const Prm &p, int64_t i) { return p.statsF32 ? (double)p.tS2L[i].x : p.tS2c[i]; }
__device__ __forceinline__ double load_logr(const Prm &p, int64_t i) { return p.statsF32 ? (double)p.tS2L[i].y : p.tLogR[i]; }
__device__ __forceinline__ float2 load_s2l(const Prm &p, int64_t i) {
    return p.statsF32 ? p.tS2L[i] : make_float2((float)p.tS2c[i], (float)p.tLogR[i]);
}

// ---------------------------------------------------------------------------------------------------------------
// recurrence steps.  Each chain policy CH provides
//   Carry, In, U (prefetch depth), FWD, load(), step<STORE>(), init_true(), init_cold(), same()
// ---------------------------------------------------------------------------------------------------------------

// k-ulp (float32) proximity tests used by the tolerant validation mode (Prm::xTolUlps = k > 0); k = 0 never matches.
__device__ __forceinline__ float ulp_of(float mag) { return __uint_as_float(f2u(mag) & 0x7f800000u) * 1.1920929e-07f; }
__device__ __forceinline__ bool near_ulps(float a, float b, float scale, int k) {
    return fabsf(a - b) <= (float)k * ulp_of(fmaxf(fmaxf(fabsf(a), fabsf(b)), scale));
}


// ---- forward covariance chain, levelTrend (pyx:394-401, 408-435, 458, 481-495) -------------------------------
#ifndef CSR_U_P
#define CSR_U_P 8
#endif
#ifndef CSR_U_X
#define CSR_U_X 8
#endif
#ifndef CSR_U_B
#define CSR_U_B 4
#endif
enum { FAM_OTHER = 0, FAM_FWD_FUSED = 1, FAM_BWD_TREND = 2 };
enum { CK_NONE = 0, CK_FWDP_TREND, CK_FWDX_TREND, CK_FWD_FUSED_TREND, CK_FWDP_LEVEL, CK_FWDX_LEVEL, CK_FWD_FUSED_LEVEL,
       CK_BWD_TREND, CK_BWD_LEVEL };      // carry types of the folded validation (Prm::prevKind)      // what run_chain dispatches its LDS-DMA variants on

// UF ("unit F"): F = [[1, f], [0, 1]] -- what the reference's constructMatrixF always builds (core.py:2164-2176).  1 * x and
// 0 * x + y are exact, so the UF instances drop those operations and produce THE SAME BITS with fewer dependent-issue
// instructions (8 of ~75 in the fused forward step, 17 of ~110 in the smoother step): the latency-bound chains of small
// batches (8-GPU shards, ECM sweeps) are a count of exactly those instructions.  The general instances stay for any other F.
template <bool UF>
struct FwdPTrendT {
    static constexpr bool UNITF = UF;
    static constexpr int FAMILY = FAM_OTHER;
    static constexpr int KIND = CK_FWDP_TREND;
    static constexpr bool USES_Q = true;      // reads the base process noise (per-chain Q0 needs the PCQ kernels)
    // default mode, superblock state chain: the main phase can write the gain records (and Pf) in the reference layout through
    // LDS tiles (walk_nat_gain) -- the state chain reads them there, a conversion launch less in front of it
    static constexpr bool NATOUT_FWD = true;
    static constexpr bool GAIN_NAT = true;
    static constexpr bool NATOUT = false;
    static constexpr bool DMA = false;
    static constexpr int NW = 1, ND = 1;
    static constexpr bool FWD = true;
    static constexpr bool PINGPONG = false;   // measured: pays only for the latency-bound state chain
    static constexpr int U = CSR_U_P;
    struct Carry {
        float c00, c01, c11;   // filtered covariance after the float32 rounding (c10 == c01, pyx:494)
        float pad_;
    };
    struct In {
        double s0u;
        float lam, kap, qs;
    };
    __device__ static __forceinline__ In load(const Prm &p, int64_t i, int64_t bq, int s, int len) {
        In in;
        in.s0u = p.tSZ[i].x;
        in.lam = (p.flags & F_LAMBDA) ? p.tLam[i] : 1.0f;
        in.kap = (p.flags & F_KAPPA) ? p.tKap[i] : 1.0f;
        in.qs = (p.flags & F_QSCALE) ? p.tQs[i] : 1.0f;
        return in;
    }
    __device__ static __forceinline__ Carry init_true(const Prm &p) {
        Carry c;
        c.c00 = (float)p.cinit;
        c.c01 = 0.0f;
        c.c11 = (float)p.cinit;
        c.pad_ = 0.0f;
        return c;
    }
    __device__ static __forceinline__ Carry init_cold(const Prm &p) { return init_true(p); }
    __device__ static __forceinline__ bool same(const Prm &p, const Carry &a, const Carry &b) {
        const bool bits = ((f2u(a.c00) ^ f2u(b.c00)) | (f2u(a.c01) ^ f2u(b.c01)) | (f2u(a.c11) ^ f2u(b.c11))) == 0u;
        // Tolerant mode: two float32-rounded Riccati trajectories contract to within an ulp quickly but may then sit one
        // ulp apart for a long time (the contraction of a 1-ulp difference rounds back to 0 or 1 ulp); accept k ulps
        // per entry, the cross term measured against sqrt(c00 c11).
        const int k = p.xTolUlps;
        const bool near = (k > 0) & near_ulps(a.c00, b.c00, 0.f, k) & near_ulps(a.c11, b.c11, 0.f, k) &
                          near_ulps(a.c01, b.c01, sqrtf(fabsf(a.c00 * a.c11)), k);
        return bits | near;
    }
    struct Gain {            // what the state update of the same bin needs (also the content of the gain record)
        double gs;
        float p00, p10;
        double is, lam;      // 1 + P00pred S0 and the clamped observation-precision multiplier (NIS / NLL terms inside the fused chain)
    };
    // b, s: block / step of this bin (for the shifted pNoise store)
    template <bool STORE>
    __device__ static __forceinline__ void step(const Prm &p, Carry &c, const In &in, int64_t b, int s, int64_t i,
                                                int64_t bfirst) {
        Gain g;
        advance<STORE>(p, c, in, b, s, i, bfirst, g);
    }
    template <bool STORE, class InT>
    __device__ static __forceinline__ void advance(const Prm &p, Carry &c, const InT &in, int64_t b, int s, int64_t i,
                                                   int64_t bfirst, Gain &gout) {
        const double kap = (p.flags & F_KAPPA) ? clampd((double)in.kap, p.kMin, p.kMax) : 1.0;
        const double lam = (p.flags & F_LAMBDA) ? clampd((double)in.lam, p.wMin, p.wMax) : 1.0;
        double qf = (double)in.qs;
        if (p.flags & F_KAPPA) qf = qf / kap;     // (qScale / procPrec), pyx:412
        const double Q00 = qf * p.Q00, Q01 = qf * p.Q01, Q10 = qf * p.Q10, Q11 = qf * p.Q11;
        const double c00 = (double)c.c00, c01 = (double)c.c01, c11 = (double)c.c11;
        // F P F^T + Q, rounded to float32 (pyx:417-430); the carry is symmetric (c10 == c01)
        double a00, a01, a10, a11;
        if constexpr (UF) {
            const double t00 = fma(p.F01, c01, c00);          // fma(F01, c01, 1 * c00)
            const double t01 = fma(p.F01, c11, c01);
            a00 = r32(fma(t01, p.F01, t00 + Q00));            // fma(t00, 1, Q00) == t00 + Q00 (one rounding)
            a01 = r32(t01 + Q01);                             // fma(t01, 1, fma(t00, 0, Q01))
            a10 = r32(fma(c11, p.F01, c01 + Q10));            // t11 == c11, t10 == c01
            a11 = r32(c11 + Q11);
        } else {
            const double t00 = fma(p.F01, c01, p.F00 * c00);
            const double t01 = fma(p.F01, c11, p.F00 * c01);
            const double t10 = fma(p.F11, c01, p.F10 * c00);
            const double t11 = fma(p.F11, c11, p.F10 * c01);
            a00 = r32(fma(t01, p.F01, fma(t00, p.F00, Q00)));
            a01 = r32(fma(t01, p.F11, fma(t00, p.F10, Q01)));
            a10 = r32(fma(t11, p.F01, fma(t10, p.F00, Q10)));
            a11 = r32(fma(t11, p.F11, fma(t10, p.F10, Q11)));
        }
        // collapsed measurement update (pyx:458, 481-495)
        const double S0 = lam * in.s0u;
        const double is = fma(a00, S0, 1.0);
        const double r = rcp_nr(is);
        const double gG = S0 * r;
        const double gH = gG * r;
        const double i00 = fma(-a00, gG, 1.0);
        const double i10 = -(a10 * gG);
        const double n00 = fma(gH, a00 * a00, i00 * i00 * a00);
        const double n01 = fma(gH, a00 * a10, i00 * fma(i10, a00, a01));
        const double n11 = fma(gH, a10 * a10, fma(i10 * i10, a00, fma(2.0 * i10, a10, a11)));
        c.c00 = (float)n00;
        c.c01 = (float)n01;
        c.c11 = (float)n11;
        gout.gs = gG;
        gout.p00 = (float)a00;
        gout.p10 = (float)a10;
        gout.is = is;
        gout.lam = lam;
        if constexpr (STORE) {
            if (p.predCompact) { if (p.storePP) p.tPP[i] = (float)a00; }
            else p.tXin[i] = pack_gain_trend(gG, (float)a00, (float)a10);
            p.tPf[i] = make_float4(c.c00, c.c01, c.c01, c.c11);
            // pNoiseForward[k-1] = Q used to reach k (pyx:504-508): shifted store, skipped at the chain's first bin
            if (p.qFromKappa) {
                if (s > 0) p.tQ2[i - 64] = make_float2((float)Q00, (float)Q11);
                else if (b > bfirst) p.tQ2[tidx(b - 1, p.B - 1, p.B)] = make_float2((float)Q00, (float)Q11);
            } else if (!p.qFromMult) {     // constant float32(Q0) otherwise: neither stored nor read back (smoother, export)
                if (s > 0) p.tQ[i - 64] = make_float4((float)Q00, (float)Q01, (float)Q10, (float)Q11);
                else if (b > bfirst) p.tQ[tidx(b - 1, p.B - 1, p.B)] = make_float4((float)Q00, (float)Q01, (float)Q10, (float)Q11);
            }
        }
    }
};

using FwdPTrend = FwdPTrendT<false>;

// ---- forward covariance chain, level (pyx:613-633, 655, 676-680); carries stay in double ----------------------
struct FwdPLevel {
    static constexpr bool UNITF = false;
    static constexpr int FAMILY = FAM_OTHER;
    static constexpr int KIND = CK_FWDP_LEVEL;
    static constexpr bool USES_Q = true;      // reads the base process noise (per-chain Q0 needs the PCQ kernels)
    static constexpr bool NATOUT_FWD = false;
    static constexpr bool NATOUT = false;
    static constexpr bool DMA = false;
    static constexpr int NW = 1, ND = 1;
    static constexpr bool FWD = true;
    static constexpr bool PINGPONG = false;   // measured: pays only for the latency-bound state chain
    static constexpr int U = CSR_U_P;
    struct Carry {
        double p;
    };
    using In = FwdPTrend::In;
    __device__ static __forceinline__ In load(const Prm &p, int64_t i, int64_t bq, int s, int len) {
        return FwdPTrend::load(p, i, bq, s, len);
    }
    __device__ static __forceinline__ Carry init_true(const Prm &p) { return Carry{p.cinit}; }
    __device__ static __forceinline__ Carry init_cold(const Prm &p) { return Carry{p.cinit}; }
    __device__ static __forceinline__ bool same(const Prm &p, const Carry &a, const Carry &b) {
        const bool bits = __double_as_longlong(a.p) == __double_as_longlong(b.p);
        // double carries (pyx:579-580) never coalesce bitwise: 1e-12 relative in exact mode, k float32 ulps otherwise
        const double rel = p.xTolUlps > 0 ? 5.9604644775390625e-8 * (double)p.xTolUlps : 1.0e-12;
        const bool tol = fabs(a.p - b.p) <= rel * fmax(fabs(a.p), fabs(b.p));
        return bits | tol;
    }
    struct Gain {
        double gs, pp;
    };
    template <bool STORE>
    __device__ static __forceinline__ void step(const Prm &p, Carry &c, const In &in, int64_t b, int s, int64_t i,
                                                int64_t bfirst) {
        Gain g;
        advance<STORE>(p, c, in, b, s, i, bfirst, g);
    }
    template <bool STORE, class InT>
    __device__ static __forceinline__ void advance(const Prm &p, Carry &c, const InT &in, int64_t b, int s, int64_t i,
                                                   int64_t bfirst, Gain &gout) {
        const double kap = (p.flags & F_KAPPA) ? clampd((double)in.kap, p.kMin, p.kMax) : 1.0;
        const double lam = (p.flags & F_LAMBDA) ? clampd((double)in.lam, p.wMin, p.wMax) : 1.0;
        double qf = (double)in.qs;
        if (p.flags & F_KAPPA) qf = qf / kap;
        const double Q = qf * p.Q00;
        const double pp = c.p + Q;
        const double S0 = lam * in.s0u;
        const double is = fma(pp, S0, 1.0);
        const double r = rcp_nr(is);
        const double gG = S0 * r;
        const double gH = gG * r;
        const double ikh = fma(-pp, gG, 1.0);
        c.p = fma(gH, pp * pp, ikh * ikh * pp);
        gout.gs = gG;
        gout.pp = pp;
        if constexpr (STORE) {
            p.tXin[i] = pack_gain_level(gG, pp);
            p.tPf[i] = make_float4((float)c.p, 0.f, 0.f, 0.f);
            if (p.qFromKappa) {
                if (s > 0) p.tQ2[i - 64] = make_float2((float)Q, 0.f);
                else if (b > bfirst) p.tQ2[tidx(b - 1, p.B - 1, p.B)] = make_float2((float)Q, 0.f);
            } else if (!p.qFromMult) {
                if (s > 0) p.tQ[i - 64] = make_float4((float)Q, 0.f, 0.f, 0.f);
                else if (b > bfirst) p.tQ[tidx(b - 1, p.B - 1, p.B)] = make_float4((float)Q, 0.f, 0.f, 0.f);
            }
        }
    }
};

// ---- forward state chain, levelTrend (pyx:403-406, 477-479) --------------------------------------------------
template <bool UF>
struct FwdXTrendT {
    static constexpr bool UNITF = UF;
    static constexpr int FAMILY = FAM_OTHER;
    static constexpr int KIND = CK_FWDX_TREND;
    static constexpr bool USES_Q = false;      // reads the base process noise (per-chain Q0 needs the PCQ kernels)
    static constexpr bool NATOUT_FWD = false;
    static constexpr bool NATOUT = false;
    static constexpr bool FWD = true;
    static constexpr bool PINGPONG = true;   // measured: pays only for the latency-bound state chain
    static constexpr int U = CSR_U_X;
    struct Carry {
        float x0, x1;
    };
    struct In {
        double zbar, gs;
        float2 cp;
    };
    __device__ static __forceinline__ In load(const Prm &p, int64_t i, int64_t bq, int s, int len) {
        In in;
        in.zbar = p.tSZ[i].y;
        const float4 r = p.tXin[i];
        in.gs = unpack_d(r.x, r.y);
        in.cp = make_float2(r.z, r.w);
        return in;
    }
    // LDS-DMA traits: the 16-byte gain record in one piece ([lane][4] words), zbar as two 4-byte pieces
    static constexpr bool DMA = true;
    static constexpr int NW = 6, ND = 3;
    __device__ static __forceinline__ void dma_issue(const Prm &p, int64_t i, unsigned *slot) {
        const char *z = reinterpret_cast<const char *>(p.tSZ + i) + 8;
        dma16(p.tXin + i, slot);
        dma4(z, slot + 256);
        dma4(z + 4, slot + 320);
    }
    __device__ static __forceinline__ In dma_read(const Prm &, const unsigned *slot, int lane) {
        const uint4 r = lds_rd128(slot + lane * 4);
        const unsigned w0 = lds_rd32(slot + 256 + lane), w1 = lds_rd32(slot + 320 + lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        In in;
        in.zbar = words2double(w0, w1);
        in.gs = words2double(r.x, r.y);
        in.cp = make_float2(__uint_as_float(r.z), __uint_as_float(r.w));
        return in;
    }
    __device__ static __forceinline__ Carry init_true(const Prm &p) { return Carry{(float)p.init, 0.0f}; }
    __device__ static __forceinline__ Carry init_cold(const Prm &p) { return Carry{(float)p.init, 0.0f}; }
    __device__ static __forceinline__ bool same(const Prm &p, const Carry &a, const Carry &b) {
        const bool bits = ((f2u(a.x0) ^ f2u(b.x0)) | (f2u(a.x1) ^ f2u(b.x1))) == 0u;
        // Tolerance mode (p.xTolUlps > 0): the rounded 2-D state recursion re-excites ulp-level differences (a level
        // ulp is >> a trend ulp), so exact coalescence of two trajectories can take 10^4 steps although they agree to
        // an ulp after ~10^2.  Accept a carry-in whose two contributions to the next predicted level, |dx0| and
        // |F01||dx1|, are each within k ulps of the level; both values finite.
        // ulp of max(|level|, 1): a level crossing zero has arbitrarily small ulps while the rounding noise it inherits
        // from its neighbours does not shrink; 2 ulp(1) = 2.4e-7 absolute is an order below the reference tests' atol
        const float mag = fmaxf(fmaxf(fabsf(a.x0), fabsf(b.x0)), 1.0f);
        const float ulp = __uint_as_float((f2u(mag) & 0x7f800000u)) * 1.1920929e-07f;   // 2^(e-23)
        const float lim = (float)p.xTolUlps * ulp;
        const bool near = (p.xTolUlps > 0) & (fabsf(a.x0 - b.x0) <= lim) & ((float)fabs(p.F01) * fabsf(a.x1 - b.x1) <= lim);
        return bits | near;
    }
    template <bool STORE>
    __device__ static __forceinline__ void step(const Prm &p, Carry &c, const In &in, int64_t b, int s, int64_t i, int64_t bf) {
        double dz, dl;
        step_d<STORE>(p, c, in, b, s, i, bf, dz, dl);
    }
    // the same step; also hands out the innovation dz = zbar - xpred and dl = gs dz (NIS = (lam S2c + dl dz) / m)
    template <bool STORE>
    __device__ static __forceinline__ void step_d(const Prm &p, Carry &c, const In &in, int64_t, int, int64_t i, int64_t,
                                                  double &dz, double &dl) {
        const double x0 = (double)c.x0, x1 = (double)c.x1;
        double xp0, xp1;
        if constexpr (UF) {
            xp0 = r32(fma(p.F01, x1, x0));
            xp1 = x1;                                         // r32(fma(1, x1, 0 * x0)) of a float32 value
        } else {
            xp0 = r32(fma(p.F01, x1, p.F00 * x0));
            xp1 = r32(fma(p.F11, x1, p.F10 * x0));
        }
        dz = in.zbar - xp0;
        dl = in.gs * dz;                                      // S1/innovScale with S1 = S0 (zbar - x)
        c.x0 = (float)fma((double)in.cp.x, dl, xp0);
        c.x1 = (float)fma((double)in.cp.y, dl, xp1);
        if constexpr (STORE) p.tXf[i] = make_float2(c.x0, c.x1);
    }
};

using FwdXTrend = FwdXTrendT<false>;

// ---- forward state chain, level (pyx:673-674); double carry ---------------------------------------------------
struct FwdXLevel {
    static constexpr bool UNITF = false;
    static constexpr int FAMILY = FAM_OTHER;
    static constexpr int KIND = CK_FWDX_LEVEL;
    static constexpr bool USES_Q = false;      // reads the base process noise (per-chain Q0 needs the PCQ kernels)
    static constexpr bool NATOUT_FWD = false;
    static constexpr bool NATOUT = false;
    static constexpr bool DMA = true;
    static constexpr int NW = 6, ND = 3;
    __device__ static __forceinline__ void dma_issue(const Prm &p, int64_t i, unsigned *slot) {
        const char *z = reinterpret_cast<const char *>(p.tSZ + i) + 8;
        dma16(p.tXin + i, slot);
        dma4(z, slot + 256);
        dma4(z + 4, slot + 320);
    }
    static constexpr bool FWD = true;
    static constexpr bool PINGPONG = true;   // measured: pays only for the latency-bound state chain
    static constexpr int U = CSR_U_X;
    struct Carry {
        double x;
    };
    struct In {
        double zbar, gs, pp;
    };
    __device__ static __forceinline__ In load(const Prm &p, int64_t i, int64_t bq, int s, int len) {
        In in;
        in.zbar = p.tSZ[i].y;
        const float4 r = p.tXin[i];
        in.gs = unpack_d(r.x, r.y);
        in.pp = unpack_d(r.z, r.w);
        return in;
    }
    __device__ static __forceinline__ In dma_read(const Prm &, const unsigned *slot, int lane) {
        const uint4 r = lds_rd128(slot + lane * 4);
        const unsigned w0 = lds_rd32(slot + 256 + lane), w1 = lds_rd32(slot + 320 + lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        In in;
        in.zbar = words2double(w0, w1);
        in.gs = words2double(r.x, r.y);
        in.pp = words2double(r.z, r.w);
        return in;
    }
    __device__ static __forceinline__ Carry init_true(const Prm &p) { return Carry{p.init}; }
    __device__ static __forceinline__ Carry init_cold(const Prm &p) { return Carry{p.init}; }
    __device__ static __forceinline__ bool same(const Prm &p, const Carry &a, const Carry &b) {
        const bool bits = __double_as_longlong(a.x) == __double_as_longlong(b.x);
        const double rel = p.xTolUlps > 0 ? 5.9604644775390625e-8 * (double)p.xTolUlps : 1.0e-12;
        const double mag = fmax(fabs(a.x), fabs(b.x));
        const bool tol = fabs(a.x - b.x) <= rel * (p.xTolUlps > 0 ? fmax(mag, 1.0) : mag) + 1.0e-300;
        return bits | tol;
    }
    template <bool STORE>
    __device__ static __forceinline__ void step(const Prm &p, Carry &c, const In &in, int64_t, int, int64_t i, int64_t) {
        const double dl = in.gs * (in.zbar - c.x);
        c.x = fma(in.pp, dl, c.x);
        if constexpr (STORE) {
            p.tXf[i] = make_float2((float)c.x, 0.f);
            p.tXd[i] = c.x;
        }
    }
};



// ---- fused forward chains (tolerant validation only): covariance and state of a bin advance in the same step -------
// With k-ulp validation both chains need the same ~80-bin window and a single validation pass, so one kernel replaces
// two (one fixed launch/drain cost, no gain-record round trip through HBM for the state update).  The arithmetic is the
// split chains' own (advance() / step() above are called as they are), so the results are the same numbers.
template <bool UF>
struct FwdTrendFusedT {
    static constexpr bool UNITF = UF;
    static constexpr int FAMILY = FAM_FWD_FUSED;
    static constexpr int KIND = CK_FWD_FUSED_TREND;
    using PT = FwdPTrendT<UF>;
    using XT = FwdXTrendT<UF>;
    static constexpr bool USES_Q = true;      // reads the base process noise (per-chain Q0 needs the PCQ kernels)
    static constexpr bool NATOUT = false;
    static constexpr bool NATOUT_FWD = true;   // main phase can also emit xf / Pf in the reference layout (walk_nat_fwd)
    static constexpr bool DMA = false;
    static constexpr int NW = 1, ND = 1;
    static constexpr bool FWD = true;
    static constexpr bool PINGPONG = false;
    static constexpr int U = CSR_U_P;
    struct Carry {
        typename PT::Carry P;
        typename XT::Carry X;
        float pad_[2];
    };
    struct In {
        double s0u, zbar;
        float lam, kap, qs;
    };
    __device__ static __forceinline__ In load(const Prm &p, int64_t i, int64_t bq, int s, int len) {
        const typename PT::In a = PT::load(p, i, bq, s, len);
        In in;
        in.s0u = a.s0u; in.lam = a.lam; in.kap = a.kap; in.qs = a.qs;
        in.zbar = p.tSZ[i].y;
        return in;
    }
    __device__ static __forceinline__ Carry init_true(const Prm &p) {
        Carry c;
        c.P = PT::init_true(p);
        c.X = XT::init_true(p);
        c.pad_[0] = c.pad_[1] = 0.f;
        return c;
    }
    __device__ static __forceinline__ Carry init_cold(const Prm &p) { return init_true(p); }
    __device__ static __forceinline__ bool same(const Prm &p, const Carry &a, const Carry &b) {
        return PT::same(p, a.P, b.P) & XT::same(p, a.X, b.X);
    }
    template <bool STORE>
    __device__ static __forceinline__ void step(const Prm &p, Carry &c, const In &in, int64_t b, int s, int64_t i,
                                                int64_t bfirst) {
        typename PT::Gain g;
        PT::template advance<STORE>(p, c.P, in, b, s, i, bfirst, g);
        typename XT::In xin;
        xin.zbar = in.zbar;
        xin.gs = g.gs;
        xin.cp = make_float2(g.p00, g.p10);
        XT::template step<STORE>(p, c.X, xin, b, s, i, bfirst);
    }
    // The step with the NIS / NLL terms of its bin (pyx:458-475) from what it holds in registers anyway.  With S1 = S0 dz,
    // S2 = lam S2c + S0 dz^2 and is = 1 + P00pred S0 the reference's quadForm = S2 - (P00pred / is) S1^2 is lam S2c + (S0 / is) dz^2
    // = lam S2c + gs dz^2 -- the cancellation-free form, one fma on top of the state update's own dl = gs dz.  The log terms of
    // the NLL never enter the serial path: sum_k log(is_k) and sum_k log(lam_k) are the logs of running PRODUCTS (renormalised
    // by the walker every 8 steps, NisAcc::renorm), evaluated once per block.
    struct NisAcc {
        double sumD = 0.0, sumQ = 0.0, sumSL = 0.0, prodIs = 1.0, prodLam = 1.0;
        int exIs = 0, exLam = 0, bins = 0;
        __device__ __forceinline__ void renorm(const Prm &p) {
            int e;
            prodIs = frexp(prodIs, &e); exIs += e;
            if (p.flags & F_LAMBDA) { prodLam = frexp(prodLam, &e); exLam += e; }
        }
        __device__ __forceinline__ double nll(const Prm &p) const {
            const double ln2 = 0.693147180559945309417232121458, log2pi = 1.8378770664093454835606594728112;
            const double mD = (double)p.m;
            double sl = sumSL;
            if (p.flags & F_LAMBDA) sl -= mD * (log_pos(prodLam) + (double)exLam * ln2);
            return 0.5 * (sl + (log_pos(prodIs) + (double)exIs * ln2) + sumQ + (double)bins * mD * log2pi);
        }
    };
    template <bool STORE>
    __device__ static __forceinline__ float step_nis(const Prm &p, Carry &c, const In &in, float2 s2l, int64_t b, int s, int64_t i,
                                                     int64_t bfirst, NisAcc &acc) {
        typename PT::Gain g;
        PT::template advance<STORE>(p, c.P, in, b, s, i, bfirst, g);
        typename XT::In xin;
        xin.zbar = in.zbar;
        xin.gs = g.gs;
        xin.cp = make_float2(g.p00, g.p10);
        double dz, dl;
        XT::template step_d<STORE>(p, c.X, xin, b, s, i, bfirst, dz, dl);
        double quad = fma(dl, dz, g.lam * (double)s2l.x);
        if (quad < 0.0) quad = 0.0;
        const float D = (float)(quad * p.rM);
        acc.sumD += (double)D;
        if (p.flags & F_NLL) {
            acc.sumQ += quad;
            acc.sumSL += (double)s2l.y;
            acc.prodIs *= g.is;
            if (p.flags & F_LAMBDA) acc.prodLam *= g.lam;
            acc.bins += 1;
        }
        return D;
    }
};
using FwdTrendFused = FwdTrendFusedT<false>;
// LDS-DMA variant of the fused forward chain (k_chain_spec_dma).  PMC on the plain kernel: the wavefront is parked on
// s_waitcnt 62-66 % of its cycles and issues only 27-30 % -- with loads AND stores in flight hipcc waits vmcnt(0) for
// every register-prefetched batch, i.e. also for the store acknowledgements of the previous batch.  Through the ring the
// inputs arrive in LDS DMA_L steps ahead under a counted wait that younger stores only make more conservative.
// Rows of a slot (64 lanes x 4 bytes each): s0u lo / hi, zbar lo / hi [, lambda, kappa, qScale when MULT].
template <int MULT, bool UF = false>      // MULT 0: no per-bin multipliers; 1: kappa only (the reference's default ECM); 2: all three
struct FwdTrendFusedDma : FwdTrendFusedT<UF> {
    using Plain = FwdTrendFusedT<UF>;         // the register-prefetch policy with the same arithmetic (tile walker)
    using In = typename Plain::In;
    static constexpr bool DMA = true;
    static constexpr bool NATOUT_FWD = false;
    // slot: [lane][4 words] = the (S0u, zbar) record in one 16-byte DMA, then one 64-word row per multiplier
    static constexpr int NW = MULT == 0 ? 4 : (MULT == 1 ? 5 : 7), ND = MULT == 0 ? 1 : (MULT == 1 ? 2 : 4);
    __device__ static __forceinline__ void dma_issue(const Prm &p, int64_t i, unsigned *slot) {
        dma16(p.tSZ + i, slot);
        if constexpr (MULT == 1) dma4(p.tKap + i, slot + 256);
        if constexpr (MULT == 2) {
            // a multiplier that is switched off is fetched from a valid dummy address (ND must not depend on flags)
            const float *d = reinterpret_cast<const float *>(p.tSZ + i);
            dma4((p.flags & F_LAMBDA) ? p.tLam + i : d, slot + 256);
            dma4((p.flags & F_KAPPA) ? p.tKap + i : d, slot + 320);
            dma4((p.flags & F_QSCALE) ? p.tQs + i : d, slot + 384);
        }
    }
    __device__ static __forceinline__ In dma_read(const Prm &p, const unsigned *slot, int lane) {
        const uint4 r = lds_rd128(slot + lane * 4);
        unsigned l = 0, k = 0, q = 0;
        if constexpr (MULT == 1) k = lds_rd32(slot + 256 + lane);
        if constexpr (MULT == 2) {
            l = lds_rd32(slot + 256 + lane);
            k = lds_rd32(slot + 320 + lane);
            q = lds_rd32(slot + 384 + lane);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        In in;
        in.s0u = words2double(r.x, r.y);
        in.zbar = words2double(r.z, r.w);
        in.lam = in.kap = in.qs = 1.0f;       // the values FwdPTrend::load delivers: 1 when a multiplier is switched off
        if constexpr (MULT == 1) in.kap = __uint_as_float(k);
        if constexpr (MULT == 2) {
            if (p.flags & F_LAMBDA) in.lam = __uint_as_float(l);
            if (p.flags & F_KAPPA) in.kap = __uint_as_float(k);
            if (p.flags & F_QSCALE) in.qs = __uint_as_float(q);
        }
        return in;
    }
};

struct FwdLevelFused {
    static constexpr bool UNITF = false;
    static constexpr int FAMILY = FAM_OTHER;
    static constexpr int KIND = CK_FWD_FUSED_LEVEL;
    static constexpr bool USES_Q = true;      // reads the base process noise (per-chain Q0 needs the PCQ kernels)
    static constexpr bool NATOUT_FWD = false;
    static constexpr bool NATOUT = false;
    static constexpr bool DMA = false;
    static constexpr int NW = 1, ND = 1;
    static constexpr bool FWD = true;
    static constexpr bool PINGPONG = false;
    static constexpr int U = CSR_U_P;
    struct Carry {
        FwdPLevel::Carry P;
        FwdXLevel::Carry X;
    };
    using In = FwdTrendFused::In;
    __device__ static __forceinline__ In load(const Prm &p, int64_t i, int64_t bq, int s, int len) {
        return FwdTrendFused::load(p, i, bq, s, len);
    }
    __device__ static __forceinline__ Carry init_true(const Prm &p) { return Carry{FwdPLevel::init_true(p), FwdXLevel::init_true(p)}; }
    __device__ static __forceinline__ Carry init_cold(const Prm &p) { return init_true(p); }
    __device__ static __forceinline__ bool same(const Prm &p, const Carry &a, const Carry &b) {
        return FwdPLevel::same(p, a.P, b.P) & FwdXLevel::same(p, a.X, b.X);
    }
    template <bool STORE>
    __device__ static __forceinline__ void step(const Prm &p, Carry &c, const In &in, int64_t b, int s, int64_t i,
                                                int64_t bfirst) {
        FwdPLevel::Gain g;
        FwdPLevel::advance<STORE>(p, c.P, in, b, s, i, bfirst, g);
        FwdXLevel::In xin;
        xin.zbar = in.zbar;
        xin.gs = g.gs;
        xin.pp = g.pp;
        FwdXLevel::step<STORE>(p, c.X, xin, b, s, i, bfirst);
    }
};

// kappa E-step of one transition k -> k+1 (pyx:8244-8298 with the MAT2 helpers pyx:4123-4175) from the float32 smoothed
// moments of both bins and the lag-one covariance; qsNext = qScale[k+1] (1 if unused)
template <bool UF = false>
__device__ __forceinline__ float estep_kappa_trend(const Prm &p, float2 xa, float4 pk, float2 ya, float4 pk1, float4 lg,
                                                   float qsNext) {
    const double x0 = xa.x, x1 = xa.y, y0 = ya.x, y1 = ya.y;
    const double f00 = p.F00, f01 = p.F01, f10 = p.F10, f11 = p.F11;
    const double xx00 = (double)pk.x + x0 * x0, xx01 = (double)pk.y + x0 * x1;
    const double xx10 = (double)pk.z + x1 * x0, xx11 = (double)pk.w + x1 * x1;
    const double yy00 = (double)pk1.x + y0 * y0, yy01 = (double)pk1.y + y0 * y1;
    const double yy10 = (double)pk1.z + y1 * y0, yy11 = (double)pk1.w + y1 * y1;
    const double xy00 = (double)lg.x + x0 * y0, xy01 = (double)lg.y + x0 * y1;
    const double xy10 = (double)lg.z + x1 * y0, xy11 = (double)lg.w + x1 * y1;
    // yx = xy^T, Ft = F^T : ww = yy - yx Ft - F xy + (F xx) Ft
    double w00, w01, w10, w11;
    if constexpr (UF) {
        // F = [[1, f], [0, 1]]: products with 1 are the operand, products with 0 vanish from the sums (x * 0 + y == y) --
        // every remaining operation is the general expression's, in its order
        w00 = yy00 - (xy00 + xy10 * f01);
        w01 = yy01 - xy10;
        w10 = yy10 - (xy01 + xy11 * f01);
        w11 = yy11 - xy11;
        w00 -= (xy00 + f01 * xy10);
        w01 -= (xy01 + f01 * xy11);
        w10 -= xy10;
        w11 -= xy11;
        const double g00 = xx00 + f01 * xx10, g01 = xx01 + f01 * xx11;
        const double g10 = xx10, g11 = xx11;
        w00 += (g00 + g01 * f01);
        w01 += g01;
        w10 += (g10 + g11 * f01);
        w11 += g11;
    } else {
        w00 = yy00 - (xy00 * f00 + xy10 * f01);
        w01 = yy01 - (xy00 * f10 + xy10 * f11);
        w10 = yy10 - (xy01 * f00 + xy11 * f01);
        w11 = yy11 - (xy01 * f10 + xy11 * f11);
        w00 -= (f00 * xy00 + f01 * xy10);
        w01 -= (f00 * xy01 + f01 * xy11);
        w10 -= (f10 * xy00 + f11 * xy10);
        w11 -= (f10 * xy01 + f11 * xy11);
        const double g00 = f00 * xx00 + f01 * xx10, g01 = f00 * xx01 + f01 * xx11;
        const double g10 = f10 * xx00 + f11 * xx10, g11 = f10 * xx01 + f11 * xx11;
        w00 += (g00 * f00 + g01 * f01);
        w01 += (g00 * f10 + g01 * f11);
        w10 += (g10 * f00 + g11 * f01);
        w11 += (g10 * f10 + g11 * f11);
    }
    if (w00 < 0.0) w00 = 0.0;
    if (w11 < 0.0) w11 = 0.0;
    const double det = p.Q00 * p.Q11 - p.Q01 * p.Q10;
    const double qi00 = p.Q11 / det, qi01 = -p.Q01 / det, qi10 = -p.Q10 / det, qi11 = p.Q00 / det;
    double delta = qi00 * w00 + qi01 * w10 + qi10 * w01 + qi11 * w11;
    if (p.flags & F_QSCALE) delta = delta / (double)qsNext;
    if (delta < 0.0) delta = 0.0;
    double kap = (p.nu + (double)p.d) / (p.nu + delta);
    if (kap < p.kMin) kap = p.kMin;
    else if (kap > p.kMax) kap = p.kMax;
    return (float)kap;
}

// ---- backward RTS chain, levelTrend (pyx:6758-6822) ------------------------------------------------------------
// J and PPred depend only on filtered quantities of bin k (off the dependent path); the carries are the float32
// smoothed state/covariance of bin k+1, exactly what the reference re-reads from its output arrays.
template <bool UF>
struct BwdTrendT {
    static constexpr bool UNITF = UF;
    static constexpr int FAMILY = FAM_BWD_TREND;
    static constexpr int KIND = CK_BWD_TREND;
    static constexpr bool USES_Q = true;      // reads the base process noise (per-chain Q0 needs the PCQ kernels)
    static constexpr bool NATOUT_FWD = false;
    static constexpr bool NATOUT = true;     // main phase can emit the reference layout through LDS tiles (walk_nat)
    static constexpr bool DMA = false;
    static constexpr int NW = 1, ND = 1;
    static constexpr bool FWD = false;
    static constexpr bool PINGPONG = false;   // measured: pays only for the latency-bound state chain
    static constexpr int U = CSR_U_B;
    struct Carry {
        float x0, x1, p00, p01, p10, p11;
        int fresh;      // 1: next visited bin seeds the chain with its filtered values (true chain end or cold start)
        int pad_;
    };
    struct In {
        float2 xf;
        float4 pf, q;
    };
    __device__ static __forceinline__ In load(const Prm &p, int64_t i, int64_t bq, int s, int len) {
        In in;
        if (p.natIn) {      // the forward pass left xf / Pf in the reference layout only (a lane's own scattered loads: the
                            // validation kernel's rare re-runs; the speculative pass reads them through LDS tiles, k_smooth_natin)
            const int64_t g = (int64_t)p.blk[bq].x + s;
            in.xf = p.natXfIn[g];
            in.pf = p.natPfIn[g];
        } else {
            in.xf = p.tXf[i];
            in.pf = p.tPf[i];
        }
        // without multipliers the stored process noise is the constant float32(Q0): nothing to read
        if (p.qFromMult) in.q = make_float4((float)p.Q00, (float)p.Q01, (float)p.Q10, (float)p.Q11);
        else in.q = p.tQ[i];
        return in;
    }
    __device__ static __forceinline__ Carry init_true(const Prm &) { return Carry{0, 0, 0, 0, 0, 0, 1, 0}; }
    __device__ static __forceinline__ Carry init_cold(const Prm &) { return Carry{0, 0, 0, 0, 0, 0, 1, 0}; }
    __device__ static __forceinline__ bool same(const Prm &p, const Carry &a, const Carry &b) {
        const bool bits = ((f2u(a.x0) ^ f2u(b.x0)) | (f2u(a.x1) ^ f2u(b.x1)) | (f2u(a.p00) ^ f2u(b.p00)) |
                           (f2u(a.p01) ^ f2u(b.p01)) | (f2u(a.p10) ^ f2u(b.p10)) | (f2u(a.p11) ^ f2u(b.p11))) == 0u;
        // tolerant mode: same criterion as the forward chains (level / trend against the level's ulp, covariance entries
        // against their own ulp, cross terms against sqrt(p00 p11))
        const int k = p.xTolUlps;
        const float lim = (float)k * ulp_of(fmaxf(fmaxf(fabsf(a.x0), fabsf(b.x0)), 1.0f));
        const float ps = sqrtf(fabsf(a.p00 * a.p11));
        const bool near = (k > 0) & (fabsf(a.x0 - b.x0) <= lim) & ((float)fabs(p.F01) * fabsf(a.x1 - b.x1) <= lim) &
                          near_ulps(a.p00, b.p00, 0.f, k) & near_ulps(a.p11, b.p11, 0.f, k) &
                          near_ulps(a.p01, b.p01, ps, k) & near_ulps(a.p10, b.p10, ps, k);
        return bits | near;
    }
    struct Gain {
        double J00, J01, J10, J11, pp00, pp01, pp10, pp11, c00, c01, c10, c11;
    };
    __device__ static __forceinline__ Gain gain(const Prm &p, const float4 &pf, const float4 &q) {
        Gain g;
        const double f00 = pf.x, f01 = pf.y, f10 = pf.z, f11 = pf.w;
        if constexpr (UF) {
            const double a00 = fma(p.F01, f10, f00), a01 = fma(p.F01, f11, f01);      // F Pf with F = [[1, f], [0, 1]]: a10 = f10, a11 = f11
            g.pp00 = fma(a01, p.F01, a00 + (double)q.x);
            g.pp01 = a01 + (double)q.y;
            g.pp10 = fma(f11, p.F01, f10 + (double)q.z);
            g.pp11 = f11 + (double)q.w;
        } else {
            const double a00 = fma(p.F01, f10, p.F00 * f00);      // F Pf
            const double a01 = fma(p.F01, f11, p.F00 * f01);
            const double a10 = fma(p.F11, f10, p.F10 * f00);
            const double a11 = fma(p.F11, f11, p.F10 * f01);
            g.pp00 = fma(a01, p.F01, fma(a00, p.F00, (double)q.x));
            g.pp01 = fma(a01, p.F11, fma(a00, p.F10, (double)q.y));
            g.pp10 = fma(a11, p.F01, fma(a10, p.F00, (double)q.z));
            g.pp11 = fma(a11, p.F11, fma(a10, p.F10, (double)q.w));
        }
        const double det = fma(g.pp00, g.pp11, -(g.pp01 * g.pp10));   // unguarded, pyx:6780
        const double rd = rcp_nr(det);
        const double v00 = g.pp11 * rd, v01 = -g.pp01 * rd, v10 = -g.pp10 * rd, v11 = g.pp00 * rd;
        if constexpr (UF) {
            g.c00 = fma(f01, p.F01, f00);                     // Pf F^T
            g.c01 = f01;
            g.c10 = fma(f11, p.F01, f10);
            g.c11 = f11;
        } else {
            g.c00 = fma(f01, p.F01, f00 * p.F00);             // Pf F^T
            g.c01 = fma(f01, p.F11, f00 * p.F10);
            g.c10 = fma(f11, p.F01, f10 * p.F00);
            g.c11 = fma(f11, p.F11, f10 * p.F10);
        }
        g.J00 = fma(g.c01, v10, g.c00 * v00);
        g.J01 = fma(g.c01, v11, g.c00 * v01);
        g.J10 = fma(g.c11, v10, g.c10 * v00);
        g.J11 = fma(g.c11, v11, g.c10 * v01);
        return g;
    }
    struct Out {            // what one bin contributes to the outputs
        float2 xs;
        float4 ps, lag;
        bool hasLag;
    };
    template <bool WANT>
    __device__ static __forceinline__ void advance(const Prm &p, Carry &c, const In &in, Out &o) {
        o.hasLag = false;
        if (c.fresh) {      // pyx:6744-6750
            c.x0 = in.xf.x; c.x1 = in.xf.y;
            c.p00 = in.pf.x; c.p01 = in.pf.y; c.p10 = in.pf.z; c.p11 = in.pf.w;
            c.fresh = 0;
        } else {
            const Gain g = gain(p, in.pf, in.q);
            const double xf0 = in.xf.x, xf1 = in.xf.y;
            const double dx0 = (double)c.x0 - (UF ? fma(p.F01, xf1, xf0) : fma(p.F01, xf1, p.F00 * xf0));
            const double dx1 = (double)c.x1 - (UF ? xf1 : fma(p.F11, xf1, p.F10 * xf0));
            const double d00 = (double)c.p00 - g.pp00, d01 = (double)c.p01 - g.pp01;
            const double d10 = (double)c.p10 - g.pp10, d11 = (double)c.p11 - g.pp11;
            const double r00 = fma(d01, g.J01, d00 * g.J00);
            const double r01 = fma(d01, g.J11, d00 * g.J10);
            const double r10 = fma(d11, g.J01, d10 * g.J00);
            const double r11 = fma(d11, g.J11, d10 * g.J10);
            c.x0 = (float)(xf0 + fma(g.J01, dx1, g.J00 * dx0));
            c.x1 = (float)(xf1 + fma(g.J11, dx1, g.J10 * dx0));
            c.p00 = (float)((double)in.pf.x + fma(g.J01, r10, g.J00 * r00));
            c.p01 = (float)((double)in.pf.y + fma(g.J01, r11, g.J00 * r01));
            c.p10 = c.p01;                                     // pyx:6821
            c.p11 = (float)((double)in.pf.w + fma(g.J11, r11, g.J10 * r01));
            if constexpr (WANT) {
                // lag-one covariance C[k] = Pf F^T + J (Ps[k+1] - PPred), pyx:6825-6844 (dP uses the incoming carry)
                o.lag = make_float4((float)(g.c00 + fma(g.J01, d10, g.J00 * d00)),
                                    (float)(g.c01 + fma(g.J01, d11, g.J00 * d01)),
                                    (float)(g.c10 + fma(g.J11, d10, g.J10 * d00)),
                                    (float)(g.c11 + fma(g.J11, d11, g.J10 * d01)));
                o.hasLag = true;
            }
        }
        if constexpr (WANT) {
            o.xs = make_float2(c.x0, c.x1);
            o.ps = make_float4(c.p00, c.p01, c.p10, c.p11);
        }
    }
    template <bool STORE>
    __device__ static __forceinline__ void step(const Prm &p, Carry &c, const In &in, int64_t b, int s, int64_t i,
                                                int64_t bfirst) {
        Out o;
        if constexpr (STORE) {
            const Carry nextBin = c;        // float32 smoothed moments of bin k+1 (exactly what is / would be stored)
            advance<true>(p, c, in, o);
            if (p.estepKappa) {
                if (o.hasLag) {
                    const int64_t nx = (s + 1 < p.B) ? i + 64 : tidx(b + 1, 0, p.B);
                    const float qs = (p.flags & F_QSCALE) ? p.tQs[nx] : 1.0f;
                    p.tKapOut[nx] = estep_kappa_trend<UF>(p, o.xs, o.ps, make_float2(nextBin.x0, nextBin.x1),
                                                      make_float4(nextBin.p00, nextBin.p01, nextBin.p10, nextBin.p11), o.lag, qs);
                }
                if (s == 0 && b == bfirst) p.tKapOut[i] = 1.0f;   // processPrecExp[0] = 1 (pyx:8245)
            }
            if (p.storeMoments) {
                if (o.hasLag) p.tLag[i] = o.lag;
                p.tXs[i] = o.xs;
                p.tPs[i] = o.ps;
            }
        } else {
            advance<false>(p, c, in, o);
        }
    }
};

using BwdTrend = BwdTrendT<false>;

// ECM sweeps, diagonal base process noise: pNoise arrives as its two diagonal entries (Prm::tQ2)
template <bool UF>
struct BwdTrendQ2T : BwdTrendT<UF> {
    using In = typename BwdTrendT<UF>::In;
    static constexpr int FAMILY = FAM_OTHER;  // no LDS-DMA variant
    static constexpr bool NATOUT = false;    // ECM sweeps never write the reference layout
    __device__ static __forceinline__ In load(const Prm &p, int64_t i, int64_t bq, int s, int len) {
        In in;
        in.xf = p.tXf[i];
        in.pf = p.tPf[i];
        const float2 q = p.tQ2[i];
        in.q = make_float4(q.x, 0.f, 0.f, q.y);
        return in;
    }
};

// LDS-DMA inputs of the smoother chain (used for its warm-up phase, k_chain_spec_dmawarm_natbwd): the filtered
// covariance in one 16-byte DMA ([lane][4] words), the filtered state as two 4-byte rows, the stored process noise in
// one more 16-byte DMA when it varies per bin (QARR).
using BwdTrendQ2 = BwdTrendQ2T<false>;

template <bool QARR, bool UF = false>
struct BwdTrendDma : BwdTrendT<UF> {
    using Plain = BwdTrendT<UF>;
    using In = typename Plain::In;
    static constexpr bool DMA = true;
    static constexpr bool NATOUT = false;
    static constexpr int NW = QARR ? 10 : 6, ND = QARR ? 4 : 3;
    __device__ static __forceinline__ void dma_issue(const Prm &p, int64_t i, unsigned *slot) {
        const char *x = reinterpret_cast<const char *>(p.tXf + i);
        dma16(p.tPf + i, slot);
        dma4(x, slot + 256);
        dma4(x + 4, slot + 320);
        if constexpr (QARR) dma16(p.tQ + i, slot + 384);
    }
    __device__ static __forceinline__ In dma_read(const Prm &p, const unsigned *slot, int lane) {
        const uint4 f = lds_rd128(slot + lane * 4);
        const unsigned x0 = lds_rd32(slot + 256 + lane), x1 = lds_rd32(slot + 320 + lane);
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (QARR) q = lds_rd128(slot + 384 + lane * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        In in;
        in.xf = make_float2(__uint_as_float(x0), __uint_as_float(x1));
        in.pf = make_float4(__uint_as_float(f.x), __uint_as_float(f.y), __uint_as_float(f.z), __uint_as_float(f.w));
        if constexpr (QARR) in.q = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
        else in.q = make_float4((float)p.Q00, (float)p.Q01, (float)p.Q10, (float)p.Q11);
        return in;
    }
};

// ---- backward RTS chain, level (pyx:7125-7140) -----------------------------------------------------------------
struct BwdLevel {
    static constexpr bool UNITF = false;
    static constexpr int FAMILY = FAM_OTHER;
    static constexpr int KIND = CK_BWD_LEVEL;
    static constexpr bool USES_Q = true;      // reads the base process noise (per-chain Q0 needs the PCQ kernels)
    static constexpr bool NATOUT_FWD = false;
    static constexpr bool NATOUT = false;
    static constexpr bool DMA = false;
    static constexpr int NW = 1, ND = 1;
    static constexpr bool FWD = false;
    static constexpr bool PINGPONG = false;   // measured: pays only for the latency-bound state chain
    static constexpr int U = CSR_U_B;
    struct Carry {
        float x, ps;
        int fresh, pad_;
    };
    struct In {
        float xf, pf, q;
    };
    __device__ static __forceinline__ In load(const Prm &p, int64_t i, int64_t bq, int s, int len) {
        In in;
        in.xf = p.tXf[i].x;
        in.pf = p.tPf[i].x;
        if (p.qFromMult) in.q = (float)p.Q00;
        else in.q = p.tQ[i].x;
        return in;
    }
    __device__ static __forceinline__ Carry init_true(const Prm &) { return Carry{0, 0, 1, 0}; }
    __device__ static __forceinline__ Carry init_cold(const Prm &) { return Carry{0, 0, 1, 0}; }
    __device__ static __forceinline__ bool same(const Prm &p, const Carry &a, const Carry &b) {
        const bool bits = ((f2u(a.x) ^ f2u(b.x)) | (f2u(a.ps) ^ f2u(b.ps))) == 0u;
        const int k = p.xTolUlps;
        const bool near = (k > 0) & near_ulps(a.x, b.x, 1.0f, k) & near_ulps(a.ps, b.ps, 0.f, k);
        return bits | near;
    }
    template <bool STORE>
    __device__ static __forceinline__ void step(const Prm &p, Carry &c, const In &in, int64_t, int, int64_t i, int64_t) {
        if (c.fresh) {
            c.x = in.xf;
            c.ps = in.pf;
            c.fresh = 0;
        } else {
            const double pf = in.pf;
            double pp = pf + (double)in.q;
            if (pp < 1.0e-12) pp = 1.0e-12;
            const double J = pf / pp;
            const double dx = (double)c.x - (double)in.xf;
            const double dP = (double)c.ps - pp;
            c.x = (float)fma(J, dx, (double)in.xf);
            double ps = fma(J * J, dP, pf);
            if (ps < 0.0) ps = 0.0;
            c.ps = (float)ps;
            if constexpr (STORE) p.tLag[i] = make_float4((float)fma(J, dP, pf), 0.f, 0.f, 0.f);   // pyx:7142
        }
        if constexpr (STORE) {
            p.tXs[i] = make_float2(c.x, 0.f);
            p.tPs[i] = make_float4(c.ps, 0.f, 0.f, 0.f);
        }
    }
};

struct BwdLevelQ2 : BwdLevel {
    __device__ static __forceinline__ In load(const Prm &p, int64_t i, int64_t bq, int s, int len) {
        In in;
        in.xf = p.tXf[i].x;
        in.pf = p.tPf[i].x;
        in.q = p.tQ2[i].x;
        return in;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// block walker: all valid steps of block `bq` in chain order (ascending for forward chains, descending for the
// smoother).  Inputs of the next U steps are requested before the current U steps are computed.
// (Measured, profiles/r01_notes.md: hipcc sinks these loads next to their first use, so a wavefront still pays an
// L2 round trip per batch; a hand-scheduled prefetch ring is the next optimisation of these kernels.)
template <class CH, bool STORE, bool FAST>
__device__ __forceinline__ void walk_impl(const Prm &p, typename CH::Carry &c, int64_t bq, int len, bool act,
                                          int64_t bfirst, int sLo, int sHi) {
    // steps [sLo, sHi) of block bq (both multiples of 2U), ascending for forward chains, descending for the smoother.
    // Two register buffers in ping-pong: while the recursion consumes one batch of U steps the loads of the next
    // batch are in flight; no buffer is ever copied, so no wait is needed until a value is really consumed.
    constexpr int U = CH::U;
    const int B = p.B;
    const int64_t base = tbase(bq, B);
    const int cnt = sHi - sLo;
    typename CH::In bufA[U], bufB[U];
#define CSR_LOAD(buf, off)                                                                       \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                              \
        const int s_ = CH::FWD ? (sLo + (off) + u) : (sHi - 1 - ((off) + u));                    \
        if (FAST || (act && s_ < len)) buf[u] = CH::load(p, base + (int64_t)s_ * 64, bq, s_, len);           \
    }
#define CSR_STEP(buf, off)                                                                       \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                              \
        const int s_ = CH::FWD ? (sLo + (off) + u) : (sHi - 1 - ((off) + u));                    \
        if (FAST || (act && s_ < len))                                                           \
            CH::template step<STORE>(p, c, buf[u], bq, s_, base + (int64_t)s_ * 64, bfirst);     \
    }
    CSR_LOAD(bufA, 0)
#pragma unroll 1
    for (int i0 = 0; i0 < cnt; i0 += 2 * U) {
        CSR_LOAD(bufB, i0 + U)
        asm volatile("" ::: "memory");
        CSR_STEP(bufA, i0)
        if (i0 + 2 * U < cnt) { CSR_LOAD(bufA, i0 + 2 * U) }
        asm volatile("" ::: "memory");
        CSR_STEP(bufB, i0 + U)
    }
#undef CSR_LOAD
#undef CSR_STEP
}

// FAST: every lane of the wavefront walks the whole range -> no per-step predicates, one straight-line loop body,
// so the compiler's s_waitcnt accounting stays exact (counted vmcnt) and the next batch really is in flight while the
// dependent recursion runs.  Wavefronts at chain ends / starts take the predicated variant.
template <class CH, bool STORE>
__device__ __forceinline__ void walk_simple(const Prm &p, typename CH::Carry &c, int64_t bq, int len, bool act,
                                            int64_t bfirst, int sLo, int sHi) {
    // one register buffer copied forward per batch: smallest code, best when the kernel is bandwidth-bound anyway
    constexpr int U = CH::U;
    const int64_t base = tbase(bq, p.B);
    const int cnt = sHi - sLo;
    typename CH::In cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int s = CH::FWD ? (sLo + u) : (sHi - 1 - u);
        if (act && s < len) cur[u] = CH::load(p, base + (int64_t)s * 64, bq, s, len);
    }
#pragma unroll 1
    for (int i0 = 0; i0 < cnt; i0 += U) {
        if (i0 + U < cnt) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = CH::FWD ? (sLo + i0 + U + u) : (sHi - 1 - (i0 + U + u));
                if (act && s < len) nxt[u] = CH::load(p, base + (int64_t)s * 64, bq, s, len);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = CH::FWD ? (sLo + i0 + u) : (sHi - 1 - (i0 + u));
            if (act && s < len) CH::template step<STORE>(p, c, cur[u], bq, s, base + (int64_t)s * 64, bfirst);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
}

template <class CH, bool STORE>
__device__ __forceinline__ void walk_block(const Prm &p, typename CH::Carry &c, int64_t bq, int len, bool act,
                                           int64_t bfirst, int sLo, int sHi) {
    if constexpr (CH::PINGPONG) {
        if (__all(act && len >= sHi)) walk_impl<CH, STORE, true>(p, c, bq, len, act, bfirst, sLo, sHi);
        else walk_impl<CH, STORE, false>(p, c, bq, len, act, bfirst, sLo, sHi);
    } else {
        walk_simple<CH, STORE>(p, c, bq, len, act, bfirst, sLo, sHi);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Main phase of the smoother with outputs in the REFERENCE layout (no export pass for xs / Ps / lag).  A lane's own
// stores would be 64 scattered 16-byte pieces per instruction (measured: the chain gets 3x slower).  Instead the
// wavefront stages 8 steps x 64 lanes per array in LDS; 8 consecutive bins of a lane are 128 contiguous bytes in the
// natural layout, so 8 threads write one full line and an instruction writes 8 full lines.
// ---------------------------------------------------------------------------------------------------------------
struct NatTiles {
    float4 ps[8][65], lag[8][65];       // [row = step within the batch][lane], padded against bank conflicts
    float2 xs[8][65];
    int gbase[64], len[64], last[64];   // natural index of the lane's block, valid steps (0 = inactive), chain's last block
};
template <class CH>
__device__ __forceinline__ void walk_nat(const Prm &p, typename CH::Carry &c, int64_t bq, int len, bool act, bool lastBlk,
                                         int gbase, NatTiles &T) {
    const int lane = threadIdx.x;
    const int B = p.B;
    const int64_t base = tbase(bq, B);
    T.gbase[lane] = gbase;
    T.len[lane] = act ? len : 0;
    T.last[lane] = lastBlk ? 1 : 0;
    float4 *natPs = reinterpret_cast<float4 *>(p.natPs), *natLag = reinterpret_cast<float4 *>(p.natLag);
    float2 *natXs = reinterpret_cast<float2 *>(p.natXs);
    typename CH::In cur[8], nxt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int s = B - 1 - u;
        if (act && s < len) cur[u] = CH::load(p, base + (int64_t)s * 64, bq, s, len);
    }
#pragma unroll 1
    for (int s8 = B - 8; s8 >= 0; s8 -= 8) {
        if (s8 >= 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s8 - 1 - u;
                if (act && s < len) nxt[u] = CH::load(p, base + (int64_t)s * 64, bq, s, len);
            }
        }
        const bool any = __any(act && s8 < len);
        if (any) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s8 + 7 - u;
                if (act && s < len) {
                    typename CH::Out o;
                    CH::template advance<true>(p, c, cur[u], o);
                    T.xs[7 - u][lane] = o.xs;
                    T.ps[7 - u][lane] = o.ps;
                    if (o.hasLag) T.lag[7 - u][lane] = o.lag;
                }
            }
            __syncthreads();
            // float4 arrays: thread -> (lane L = k*8 + t/8, row r = t%8): 8 threads cover one 128-byte line
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int L = k * 8 + (lane >> 3), r = lane & 7, s = s8 + r;
                const int ln = T.len[L];
                if (s < ln) {
                    const int64_t g = (int64_t)T.gbase[L] + s;
                    natPs[g] = T.ps[r][L];
                    if (!(T.last[L] && s == ln - 1)) natLag[g] = T.lag[r][L];     // the chain's last bin has no lag row
                }
            }
            // float2 array: thread -> (lane L = k*16 + t/4, rows 2*(t%4), +1): 16 bytes per thread, 64 bytes per lane
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int L = k * 16 + (lane >> 2), r = (lane & 3) * 2, s = s8 + r;
                const int ln = T.len[L];
                if (s + 1 < ln) {
                    const float2 a = T.xs[r][L], b2 = T.xs[r + 1][L];
                    *reinterpret_cast<float4 *>(natXs + (int64_t)T.gbase[L] + s) = make_float4(a.x, a.y, b2.x, b2.y);
                } else if (s < ln) {
                    natXs[(int64_t)T.gbase[L] + s] = T.xs[r][L];
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
    }
}

// Re-run path of the validation kernel with natural outputs: blocks re-run rarely, so every lane simply writes its own
// bins (scattered 16-byte stores) -- no LDS, which keeps the validation kernel's dispatch cheap (an LDS allocation alone
// made it 6x slower to launch).
template <class CH>
__device__ __forceinline__ void walk_nat_direct(const Prm &p, typename CH::Carry &c, int64_t bq, int len, bool act,
                                                bool lastBlk, int gbase) {
    const int64_t base = tbase(bq, p.B);
    float4 *natPs = reinterpret_cast<float4 *>(p.natPs), *natLag = reinterpret_cast<float4 *>(p.natLag);
    float2 *natXs = reinterpret_cast<float2 *>(p.natXs);
    (void)lastBlk;
#pragma unroll 1
    for (int s = p.B - 1; s >= 0; --s) {
        if (act && s < len) {
            const typename CH::In in = CH::load(p, base + (int64_t)s * 64, bq, s, len);
            typename CH::Out o;
            CH::template advance<true>(p, c, in, o);
            const int64_t g = (int64_t)gbase + s;
            natXs[g] = o.xs;
            natPs[g] = o.ps;
            if (o.hasLag) natLag[g] = o.lag;
        }
    }
}

// Forward counterpart of walk_nat for the fused forward chain: the blocked stores stay (the smoother reads them), and the
// filtered state / covariance are ALSO written in the reference layout through the same LDS tiles, so the export pass
// has nothing left to convert but D.  natXs / natPs point at the natural xf / Pf arrays here.
struct NatTilesFwd {                    // the forward walker's share of NatTiles (no lag tile) + the D tile: 15 KB
    float4 ps[8][65];
    float2 xs[8][65];
    int gbase[64], len[64];
    float d[8][65];                     // NIS of the batch (Prm::nisInChain)
};
// where a forward walker stages D: NatTilesFwd has a tile of its own, NatTiles lends its (unused) lag tile
__device__ __forceinline__ float (*d_tile(NatTilesFwd &T))[65] { return T.d; }
__device__ __forceinline__ float (*d_tile(NatTiles &T))[65] { return reinterpret_cast<float (*)[65]>(&T.lag[0][0]); }

template <class CH, class TT, bool NIS, bool NATONLY>
__device__ __forceinline__ void walk_nat_fwd_impl(const Prm &p, typename CH::Carry &c, int64_t bq, int len, bool act,
                                                  int64_t bfirst, int gbase, TT &T) {
    const int lane = threadIdx.x;
    const int B = p.B;
    const int64_t base = tbase(bq, B);
    T.gbase[lane] = gbase;
    T.len[lane] = act ? len : 0;
    float4 *natPf = reinterpret_cast<float4 *>(p.natPs);
    float2 *natXf = reinterpret_cast<float2 *>(p.natXs);
    float (*dT)[65] = d_tile(T);
    typename CH::NisAcc acc;
    typename CH::In cur[8], nxt[8];
    float2 sl[8], sln[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (act && u < len) {
            cur[u] = CH::load(p, base + (int64_t)u * 64, bq, u, len);
            if constexpr (NIS) sl[u] = load_s2l(p, base + (int64_t)u * 64);
        }
#pragma unroll 1
    for (int s8 = 0; s8 < B; s8 += 8) {
        if (s8 + 8 < B) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s8 + 8 + u;
                if (act && s < len) {
                    nxt[u] = CH::load(p, base + (int64_t)s * 64, bq, s, len);
                    if constexpr (NIS) sln[u] = load_s2l(p, base + (int64_t)s * 64);
                }
            }
        }
        if (__any(act && s8 < len)) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s8 + u;
                if (act && s < len) {
                    // NATONLY (constant process noise): the step stores nothing in the blocked layout
                    if constexpr (NIS) dT[u][lane] = CH::template step_nis<!NATONLY>(p, c, cur[u], sl[u], bq, s, base + (int64_t)s * 64, bfirst, acc);
                    else CH::template step<!NATONLY>(p, c, cur[u], bq, s, base + (int64_t)s * 64, bfirst);
                    T.xs[u][lane] = make_float2(c.X.x0, c.X.x1);
                    T.ps[u][lane] = make_float4(c.P.c00, c.P.c01, c.P.c01, c.P.c11);
                }
            }
            if constexpr (NIS) acc.renorm(p);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int L = k * 8 + (lane >> 3), r = lane & 7, s = s8 + r;
                if (s < T.len[L]) natPf[(int64_t)T.gbase[L] + s] = T.ps[r][L];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int L = k * 16 + (lane >> 2), r = (lane & 3) * 2, s = s8 + r;
                const int ln = T.len[L];
                if (s + 1 < ln) {
                    const float2 a = T.xs[r][L], b2 = T.xs[r + 1][L];
                    *reinterpret_cast<float4 *>(natXf + (int64_t)T.gbase[L] + s) = make_float4(a.x, a.y, b2.x, b2.y);
                } else if (s < ln) {
                    natXf[(int64_t)T.gbase[L] + s] = T.xs[r][L];
                }
            }
            if constexpr (NIS) {
                // float array: thread -> (lane L = k*32 + t/2, rows 4*(t%2) .. +3): 16 bytes per thread, 32 bytes per lane
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int L = k * 32 + (lane >> 1), r = (lane & 1) * 4, s = s8 + r;
                    const int ln = T.len[L];
                    float *dst = p.natD + (int64_t)T.gbase[L] + s;
                    if (s + 3 < ln) {
                        *reinterpret_cast<float4 *>(dst) = make_float4(dT[r][L], dT[r + 1][L], dT[r + 2][L], dT[r + 3][L]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                            if (s + e < ln) dst[e] = dT[r + e][L];
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { cur[u] = nxt[u]; if constexpr (NIS) sl[u] = sln[u]; }
    }
    if constexpr (NIS) {
        if (act) {
            p.blkSumD[bq] = acc.sumD;
            p.blkSumNLL[bq] = (p.flags & F_NLL) ? acc.nll(p) : 0.0;
        }
    }
}
template <class CH, class TT = NatTiles>
__device__ __forceinline__ void walk_nat_fwd(const Prm &p, typename CH::Carry &c, int64_t bq, int len, bool act,
                                             int64_t bfirst, int gbase, TT &T) {
    if (p.nisInChain && p.natOnly) walk_nat_fwd_impl<CH, TT, true, true>(p, c, bq, len, act, bfirst, gbase, T);
    else if (p.nisInChain) walk_nat_fwd_impl<CH, TT, true, false>(p, c, bq, len, act, bfirst, gbase, T);
    else if (p.natOnly) walk_nat_fwd_impl<CH, TT, false, true>(p, c, bq, len, act, bfirst, gbase, T);
    else walk_nat_fwd_impl<CH, TT, false, false>(p, c, bq, len, act, bfirst, gbase, T);
}
// re-run path (validation kernel): scattered natural stores, no LDS
template <class CH>
__device__ __forceinline__ void walk_nat_fwd_direct(const Prm &p, typename CH::Carry &c, int64_t bq, int len, bool act,
                                                    int64_t bfirst, int gbase) {
    const int64_t base = tbase(bq, p.B);
    float4 *natPf = reinterpret_cast<float4 *>(p.natPs);
    float2 *natXf = reinterpret_cast<float2 *>(p.natXs);
    typename CH::NisAcc acc;
#pragma unroll 1
    for (int s = 0; s < p.B; ++s) {
        if (act && s < len) {
            const typename CH::In in = CH::load(p, base + (int64_t)s * 64, bq, s, len);
            if (p.nisInChain) {
                const float2 s2l = load_s2l(p, base + (int64_t)s * 64);
                p.natD[(int64_t)gbase + s] = p.natOnly
                    ? CH::template step_nis<false>(p, c, in, s2l, bq, s, base + (int64_t)s * 64, bfirst, acc)
                    : CH::template step_nis<true>(p, c, in, s2l, bq, s, base + (int64_t)s * 64, bfirst, acc);
                if ((s & 7) == 7) acc.renorm(p);
            } else if (p.natOnly) {
                CH::template step<false>(p, c, in, bq, s, base + (int64_t)s * 64, bfirst);
            } else {
                CH::template step<true>(p, c, in, bq, s, base + (int64_t)s * 64, bfirst);
            }
            natXf[(int64_t)gbase + s] = make_float2(c.X.x0, c.X.x1);
            natPf[(int64_t)gbase + s] = make_float4(c.P.c00, c.P.c01, c.P.c01, c.P.c11);
        }
    }
    if (p.nisInChain && act) {
        p.blkSumD[bq] = acc.sumD;
        p.blkSumNLL[bq] = (p.flags & F_NLL) ? acc.nll(p) : 0.0;
    }
}

// Covariance chain of the default mode with the superblock state chain: the gain records go to the reference layout (that is
// where k_sb_async / k_sb_sys read them; Prm::natLag points at them HERE) and, when Prm::natPs is set, Pf as well, through the
// LDS tiles of walk_nat (ps: Pf, lag: gain record).  The blocked Pf / pNoise stores of the step stay (the smoother reads them);
// the blocked gain record is not written in this mode (Prm::predCompact: the NIS epilogue reads P00pred from tPP).
template <class CH, class = void>
struct GainNatOf : std::false_type {};
template <class CH>
struct GainNatOf<CH, std::void_t<decltype(CH::GAIN_NAT)>> : std::bool_constant<CH::GAIN_NAT> {};
template <class CH>
__device__ __forceinline__ void walk_nat_gain(const Prm &p, typename CH::Carry &c, int64_t bq, int len, bool act,
                                              int64_t bfirst, int gbase, NatTiles &T) {
    const int lane = threadIdx.x;
    const int B = p.B;
    const int64_t base = tbase(bq, B);
    T.gbase[lane] = gbase;
    T.len[lane] = act ? len : 0;
    float4 *natPf = reinterpret_cast<float4 *>(p.natPs);
    float4 *natGn = reinterpret_cast<float4 *>(p.natLag);
    typename CH::In cur[8], nxt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (act && u < len) cur[u] = CH::load(p, base + (int64_t)u * 64, bq, u, len);
#pragma unroll 1
    for (int s8 = 0; s8 < B; s8 += 8) {
        if (s8 + 8 < B) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s8 + 8 + u;
                if (act && s < len) nxt[u] = CH::load(p, base + (int64_t)s * 64, bq, s, len);
            }
        }
        if (__any(act && s8 < len)) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s8 + u;
                if (act && s < len) {
                    typename CH::Gain g;
                    if (p.natOnly) {    // Pf leaves in the reference layout only (the smoother reads it there); P00pred for the epilogue
                        CH::template advance<false>(p, c, cur[u], bq, s, base + (int64_t)s * 64, bfirst, g);
                        if (p.storePP) p.tPP[base + (int64_t)s * 64] = g.p00;
                    } else {
                        CH::template advance<true>(p, c, cur[u], bq, s, base + (int64_t)s * 64, bfirst, g);
                    }
                    T.lag[u][lane] = pack_gain_trend(g.gs, g.p00, g.p10);
                    T.ps[u][lane] = make_float4(c.c00, c.c01, c.c01, c.c11);
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int L = k * 8 + (lane >> 3), r = lane & 7, s = s8 + r;
                if (s < T.len[L]) {
                    natGn[(int64_t)T.gbase[L] + s] = T.lag[r][L];
                    if (natPf != nullptr) natPf[(int64_t)T.gbase[L] + s] = T.ps[r][L];
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
    }
}
// re-run path (validation kernel): scattered natural stores, no LDS
template <class CH>
__device__ __forceinline__ void walk_nat_gain_direct(const Prm &p, typename CH::Carry &c, int64_t bq, int len, bool act,
                                                     int64_t bfirst, int gbase) {
    const int64_t base = tbase(bq, p.B);
    float4 *natPf = reinterpret_cast<float4 *>(p.natPs);
    float4 *natGn = reinterpret_cast<float4 *>(p.natLag);
#pragma unroll 1
    for (int s = 0; s < p.B; ++s) {
        if (act && s < len) {
            const typename CH::In in = CH::load(p, base + (int64_t)s * 64, bq, s, len);
            typename CH::Gain g;
            if (p.natOnly) {
                CH::template advance<false>(p, c, in, bq, s, base + (int64_t)s * 64, bfirst, g);
                if (p.storePP) p.tPP[base + (int64_t)s * 64] = g.p00;
            } else {
                CH::template advance<true>(p, c, in, bq, s, base + (int64_t)s * 64, bfirst, g);
            }
            natGn[(int64_t)gbase + s] = pack_gain_trend(g.gs, g.p00, g.p10);
            if (natPf != nullptr) natPf[(int64_t)gbase + s] = make_float4(c.c00, c.c01, c.c01, c.c11);
        }
    }
}

// Folded validation of the PREVIOUS stage (Prm::prevKind), run by every lane for its own block in the prologue of a
// speculative kernel.
template <class PCH>
__device__ __forceinline__ bool prev_stage_bad(const Prm &p, int64_t b, const int4 &bi) {
    using Carry = typename PCH::Carry;
    const Carry *cin = reinterpret_cast<const Carry *>(p.prevCarryIn);
    const Carry *cout = reinterpret_cast<const Carry *>(p.prevCarryOut);
    const bool live = b < p.NB && (p.prevActive == nullptr || p.prevActive[p.blkChain[b]] != 0);
    const bool edge = PCH::FWD ? (b == (int64_t)bi.z) : (b == (int64_t)bi.w);
    const bool check = live && !edge;
    const int64_t nbr = check ? (PCH::FWD ? b - 1 : b + 1) : 0;
    const Carry prev = cout[nbr];
    const Carry mine = cin[live ? b : 0];
    return check & !PCH::same(p, prev, mine);
}
__device__ __forceinline__ void check_previous_stage(const Prm &p, int64_t b, const int4 &bi) {
    if (p.prevKind == CK_NONE) return;          // uniform
    bool bad = false;
    switch (p.prevKind) {
        case CK_FWDP_TREND: bad = prev_stage_bad<FwdPTrendT<false>>(p, b, bi); break;
        case CK_FWDX_TREND: bad = prev_stage_bad<FwdXTrendT<false>>(p, b, bi); break;
        case CK_FWD_FUSED_TREND: bad = prev_stage_bad<FwdTrendFusedT<false>>(p, b, bi); break;
        case CK_FWDP_LEVEL: bad = prev_stage_bad<FwdPLevel>(p, b, bi); break;
        case CK_FWDX_LEVEL: bad = prev_stage_bad<FwdXLevel>(p, b, bi); break;
        case CK_FWD_FUSED_LEVEL: bad = prev_stage_bad<FwdLevelFused>(p, b, bi); break;
        case CK_BWD_TREND: bad = prev_stage_bad<BwdTrendT<false>>(p, b, bi); break;
        case CK_BWD_LEVEL: bad = prev_stage_bad<BwdLevel>(p, b, bi); break;
        default: break;
    }
    if (bad) {
        atomicAdd(p.prevCount, 1u);
        atomicAdd(p.prevCountPass, 1u);
    }
}
// the same check as a kernel of its own (a stage that nothing follows before the next settle point)
__global__ __launch_bounds__(64) void k_chain_check(Prm p) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    int4 bi = make_int4(0, 0, 0, 0);
    if (b < p.NB) bi = p.blk[b];
    check_previous_stage(p, b, bi);
}

// Wave-local validation and repair (warm-started sweeps only).  63 of a wavefront's 64 blocks have their neighbour in the
// same wavefront: its carry-out is one lane away, so the comparison that the validation pass makes (carry-in against the
// neighbour's carry-out) and the repair (re-run from that carry-out) can happen before the kernel ends, round by round
// until no lane changes.  A warm-started window is short because nearly every block's start is good; the few blocks whose
// multipliers moved a lot since the previous sweep fail, and repaired here they cost one more block walk of one wavefront
// instead of a replay of the iteration.  The block at a wavefront's edge is still checked by the next kernel's prologue
// (check_previous_stage) against the neighbouring wavefront's final carry-out.
template <class CH>
__device__ __forceinline__ typename CH::Carry wave_neighbour_carry(const typename CH::Carry &c) {
    constexpr int NWORDS = (int)(sizeof(typename CH::Carry) / 4);
    static_assert(sizeof(typename CH::Carry) % 4 == 0, "carries are whole words");
    int w[NWORDS];
    __builtin_memcpy(w, &c, sizeof(c));
#pragma unroll
    for (int k = 0; k < NWORDS; ++k) w[k] = CH::FWD ? __shfl_up(w[k], 1) : __shfl_down(w[k], 1);
    typename CH::Carry r;
    __builtin_memcpy(&r, w, sizeof(r));
    return r;
}
template <class CH>
__device__ __forceinline__ void carry_select(typename CH::Carry &dst, const typename CH::Carry &src, bool take) {
    // word-wise (a divergent struct phi was mis-compiled once: see k_chain_fix)
    constexpr int NWORDS = (int)(sizeof(typename CH::Carry) / 4);
    int a[NWORDS], b[NWORDS];
    __builtin_memcpy(a, &dst, sizeof(dst));
    __builtin_memcpy(b, &src, sizeof(src));
#pragma unroll
    for (int k = 0; k < NWORDS; ++k) a[k] = take ? b[k] : a[k];
    __builtin_memcpy(&dst, a, sizeof(dst));
}
template <class CH, class WCH>
__device__ __forceinline__ void wave_local_repair(const Prm &p, typename CH::Carry &cinLane, typename CH::Carry &cOut,
                                                  int64_t b, const int4 &bi, bool live) {
    const int lane = threadIdx.x & 63;
    const int64_t bfirst = bi.z, blast = bi.w;
    const bool hasNb = live && (CH::FWD ? (lane > 0 && b > bfirst) : (lane < 63 && b < blast));
    for (int round = 0; round < 64; ++round) {
        const typename CH::Carry nb = wave_neighbour_carry<CH>(cOut);
        const bool bad = hasNb & !CH::same(p, nb, cinLane);
        if (!__any(bad)) break;
        typename CH::Carry c2 = nb;
        walk_block<WCH, true>(p, c2, b, bi.y, bad, bfirst, 0, p.B);
        carry_select<CH>(cinLane, nb, bad);
        carry_select<CH>(cOut, c2, bad);
        if (bad) atomicAdd(p.localFixCount, 1u);
    }
}
template <class CH, class = void>
struct PlainOf { using type = CH; };
template <class CH>
struct PlainOf<CH, std::void_t<typename CH::Plain>> { using type = typename CH::Plain; };

// Speculative pass: one lane per block, 64 consecutive blocks per wavefront.
// NAT = true: separate instantiation whose main phase writes the reference layout through LDS tiles (walk_nat*); the
// plain one stays as lean as before (the tile walkers cost ~50-100 VGPRs and slowed the ECM sweeps by 25 % when both
// paths lived in one kernel).
// WS = true: the warm-started instance (Prm::ckptIn / ckptOut, wave_local_repair) -- a separate one, because the extra
// code cost the plain ECM sweep kernels 8 % when it lived in them (register allocation of the hot loop).
template <class CH, bool NAT = false, bool PCQ = false, bool WS = false>
__global__ __launch_bounds__(64) void k_chain_spec(Prm p_) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const Prm p = lane_model_if<PCQ>(p_, b);
    const bool live = b < p.NB && chain_on(p, b);
    int4 bi = make_int4(0, 0, 0, 0);
    if (b < p.NB) bi = p.blk[b];
    check_previous_stage(p_, b, bi);
    const int64_t bfirst = bi.z, blast = bi.w;
    typename CH::Carry c = CH::init_cold(p);
    typename CH::Carry *cin = reinterpret_cast<typename CH::Carry *>(p.carryIn);
    typename CH::Carry *cout = reinterpret_cast<typename CH::Carry *>(p.carryOutA);
    // p.warm = warm-up length in bins (multiple of 8): qmax blocks are visited, the farthest one only partially
    const int B = p.B;
    const int qmax = (p.warm + B - 1) / B;
    const int rem = p.warm - (qmax - 1) * B;           // bins taken from the farthest block, in (0, B]
    // recording point of this block's own checkpoint (for a window of ckptSaveWarm bins of the NEXT sweep):
    // FWD: the window begins at step svSplit of this block; BWD: steps [0, svSplit) of this block are inside the window
    const int svq = (WS && p.ckptSaveWarm > 0) ? (p.ckptSaveWarm + B - 1) / B : 0;
    const int svSplit = CH::FWD ? svq * B - p.ckptSaveWarm : p.ckptSaveWarm - (svq - 1) * B;
    if constexpr (CH::FWD) {
        const int avail = live ? (int)(b - bfirst) : 0;        // preceding blocks of this chain (all full)
        const int qstart = avail < qmax ? avail : qmax;
        if (live && qstart == avail) c = CH::init_true(p);
        else if (WS && live && p.ckptIn != nullptr && qmax > 0)      // window strictly inside the chain: start from the previous sweep's carry there
            c = reinterpret_cast<const typename CH::Carry *>(p.ckptIn)[b - qmax];
        for (int q = qmax; q >= 1; --q) {
            const bool act = live && q <= qstart;
            if (!__any(act)) continue;
            // a lane whose chain starts inside the window walks its first block in full (from the true prior)
            const int lo = (q == qmax) ? (B - rem) : 0;
            if (q == qmax && __any(act && avail == qmax && rem < B)) {
                // mixed wave: lanes at their chain start need the whole block, the others only the tail
                walk_block<CH, false>(p, c, b - q, B, act && avail == qmax, bfirst, 0, lo);
            }
            walk_block<CH, false>(p, c, b - q, B, act, bfirst, lo, B);
        }
        typename CH::Carry cinLane = c;
        if constexpr (!WS) { if (live) cin[b] = c; }
        if constexpr (NAT && CH::NATOUT_FWD) {
            extern __shared__ __attribute__((aligned(16))) unsigned char natTileMemF[];
            if constexpr (GainNatOf<CH>::value) walk_nat_gain<CH>(p, c, b, bi.y, live, bfirst, bi.x, *reinterpret_cast<NatTiles *>(natTileMemF));
            else walk_nat_fwd<CH>(p, c, b, bi.y, live, bfirst, bi.x, *reinterpret_cast<NatTiles *>(natTileMemF));
        } else if (WS && svq > 0 && p.ckptOut != nullptr) {
            if (svSplit > 0) walk_block<CH, true>(p, c, b, bi.y, live, bfirst, 0, svSplit);
            if (live && b != blast) reinterpret_cast<typename CH::Carry *>(p.ckptOut)[b] = c;
            walk_block<CH, true>(p, c, b, bi.y, live, bfirst, svSplit, B);
            if constexpr (WS) {
                if (p.ckptIn != nullptr) wave_local_repair<CH, CH>(p, cinLane, c, b, bi, live);
            }
        } else {
            walk_block<CH, true>(p, c, b, bi.y, live, bfirst, 0, B);
        }
        if constexpr (WS) { if (live) cin[b] = cinLane; }
        if (live) cout[b] = c;
    } else {
        const int avail = live ? (int)(blast - b) : 0;         // following blocks of this chain
        const int qstart = avail < qmax ? avail : qmax;
        int lastLen = B;
        if (live) lastLen = p.blk[blast].y;
        if (WS && live && p.ckptIn != nullptr && qmax > 0 && avail > qmax)      // window strictly inside the chain
            c = reinterpret_cast<const typename CH::Carry *>(p.ckptIn)[b + qmax];
        for (int q = qmax; q >= 1; --q) {
            const bool act = live && q <= qstart;
            if (!__any(act)) continue;
            const int64_t bq = b + q;
            const int len = (bq == blast) ? lastLen : B;
            const int hi = (q == qmax) ? rem : B;
            if (q == qmax && __any(act && avail == qmax && rem < B)) {
                // lanes whose chain ends in this block start from the true end (seeded by the first visited bin)
                walk_block<CH, false>(p, c, bq, len, act && avail == qmax, bfirst, hi, B);
            }
            walk_block<CH, false>(p, c, bq, len, act, bfirst, 0, hi);
        }
        typename CH::Carry cinLane = c;
        if constexpr (!WS) { if (live) cin[b] = c; }
        if constexpr (NAT && CH::NATOUT) {
            // dynamic LDS (sizeof(NatTiles) bytes) of this instantiation only
            extern __shared__ __attribute__((aligned(16))) unsigned char natTileMem[];
            walk_nat<CH>(p, c, b, bi.y, live, b == blast, bi.x, *reinterpret_cast<NatTiles *>(natTileMem));
        } else if (WS && svq > 0 && p.ckptOut != nullptr) {
            if (svSplit < B) walk_block<CH, true>(p, c, b, bi.y, live, bfirst, svSplit, B);
            if (live && b != blast) reinterpret_cast<typename CH::Carry *>(p.ckptOut)[b] = c;
            if (svSplit > 0) walk_block<CH, true>(p, c, b, bi.y, live, bfirst, 0, svSplit);
            if constexpr (WS) {
                if (p.ckptIn != nullptr) wave_local_repair<CH, CH>(p, cinLane, c, b, bi, live);
            }
        } else {
            walk_block<CH, true>(p, c, b, bi.y, live, bfirst, 0, B);
        }
        if constexpr (WS) { if (live) cin[b] = cinLane; }
        if (live) cout[b] = c;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant of the speculative pass (latency-bound chains).  The register prefetch of walk_* is at the mercy of
// hipcc's scheduling (it sinks loads next to their use and drains them at loop back-edges), so a wavefront pays an
// L2/HBM round trip per batch.  Here every input row of a step is fetched with global_load_lds (no VGPR destination)
// into a per-wave LDS ring DMA_L steps ahead of the recursion; completion is tracked with a COUNTED s_waitcnt
// vmcnt((DMA_L-1)*ND) -- valid because every step issues exactly ND DMA instructions with all lanes enabled (inactive
// lanes fetch a clamped, in-bounds address), and younger main-phase stores only make the wait more conservative.
// The ring is read back with inline-asm ds_read (an LDS read the compiler can see would get a vmcnt(0) in front).
// The lane walks its whole trajectory (warm-up blocks + own block) as one stream, so the pipeline is filled once.
// Chain policies opt in with DMA = true and provide dma_issue() / dma_read().
// ---------------------------------------------------------------------------------------------------------------
// Position of a wavefront on its lanes' trajectories.  (k, s) is wave-uniform: block offset relative to the lane's own
// block and step inside it; FWD visits k = -Q .. 0 with s ascending, BWD visits k = Q .. 0 with s descending.  The
// per-lane state (blocked slot, valid length, membership of block b+k in the lane's chain) is refreshed only when the
// position wraps into the next block, so a step costs one 64-bit add.
template <bool FWD>
struct LaneCursor {
    int k, s;
    int64_t idx;
    int len;
    bool ok;
    __device__ __forceinline__ void locate(const Prm &p, int64_t b, bool live, const int4 &bi, int lastLen) {
        const int64_t bq = b + k;
        ok = live && (FWD ? bq >= (int64_t)bi.z : bq <= (int64_t)bi.w);
        len = !ok ? 0 : (k == 0 ? bi.y : ((!FWD && bq == (int64_t)bi.w) ? lastLen : p.B));
        const int64_t bs = ok ? bq : (live ? b : 0);           // inactive lanes fetch a harmless in-bounds slot
        idx = tbase(bs, p.B) + (int64_t)(ok ? s : 0) * 64;
    }
    __device__ __forceinline__ void init(const Prm &p, int W, int64_t b, bool live, const int4 &bi, int lastLen) {
        const int B = p.B, qmax = (W + B - 1) / B, rem = W - (qmax - 1) * B;
        if (W == 0) { k = 0; s = FWD ? 0 : B - 1; }
        else if (FWD) { k = -qmax; s = B - rem; }
        else { k = qmax; s = rem - 1; }
        locate(p, b, live, bi, lastLen);
    }
    __device__ __forceinline__ void next(const Prm &p, int64_t b, bool live, const int4 &bi, int lastLen) {
        if (FWD) {
            ++s;
            idx += ok ? 64 : 0;
            if (s == p.B) { s = 0; ++k; locate(p, b, live, bi, lastLen); }
        } else {
            --s;
            idx -= ok ? 64 : 0;
            if (s < 0) { s = p.B - 1; --k; locate(p, b, live, bi, lastLen); }
        }
    }
};

template <class CH, bool STORE>
__device__ __forceinline__ void dma_phase(const Prm &p, typename CH::Carry &c, unsigned *ring, int64_t b, bool live,
                                          const int4 &bi, int lastLen, LaneCursor<CH::FWD> &cons,
                                          LaneCursor<CH::FWD> &iss, int &t, int tEnd, int T) {
    constexpr int ND = CH::ND;
    const int lane = threadIdx.x;
    const int64_t bfirst = bi.z;
#pragma unroll 1
    for (; t < tEnd; ++t) {
        if (t + DMA_L <= T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DMA_L - 1) * ND) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        typename CH::In in = CH::dma_read(p, ring + (t % DMA_R) * (CH::NW * 64), lane);
        __builtin_amdgcn_sched_barrier(0);
        if (t + DMA_L < T) {
            CH::dma_issue(p, iss.idx, ring + ((t + DMA_L) % DMA_R) * (CH::NW * 64));
            iss.next(p, b, live, bi, lastLen);
        }
        if (cons.ok && cons.s < cons.len)
            CH::template step<STORE>(p, c, in, b + cons.k, cons.s, cons.idx, bfirst);
        cons.next(p, b, live, bi, lastLen);
    }
}

template <class CH, bool WS = false>      // WS: warm-started instance (see k_chain_spec)
__global__ __launch_bounds__(64) void k_chain_spec_dma(Prm p) {     // state chains only: they never read Q
    extern __shared__ unsigned ringMem[];
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b < p.NB && chain_on(p, b);
    int4 bi = make_int4(0, 0, 0, 0);
    if (b < p.NB) bi = p.blk[b];
    int lastLen = p.B;
    if (!CH::FWD && live) lastLen = p.blk[bi.w].y;
    check_previous_stage(p, b, bi);
    typename CH::Carry c = CH::init_cold(p);
    typename CH::Carry *cin = reinterpret_cast<typename CH::Carry *>(p.carryIn);
    typename CH::Carry *cout = reinterpret_cast<typename CH::Carry *>(p.carryOutA);
    const int W = p.warm, T = W + p.B;
    if (WS && p.ckptIn != nullptr && W > 0 && live) {
        // warm-started window (see Prm::ckptIn): strictly inside the chain, begin from the previous sweep's carry there
        const int qw = (W + p.B - 1) / p.B;
        const int64_t avail = CH::FWD ? b - (int64_t)bi.z : (int64_t)bi.w - b;
        if (avail > qw) c = reinterpret_cast<const typename CH::Carry *>(p.ckptIn)[CH::FWD ? b - qw : b + qw];
    }
    LaneCursor<CH::FWD> cons, iss;
    cons.init(p, W, b, live, bi, lastLen);
    iss.init(p, W, b, live, bi, lastLen);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // block-table loads retired before the ring starts
    for (int t0 = 0; t0 < DMA_L && t0 < T; ++t0) {
        CH::dma_issue(p, iss.idx, ringMem + (t0 % DMA_R) * (CH::NW * 64));
        iss.next(p, b, live, bi, lastLen);
    }
    int t = 0;
    dma_phase<CH, false>(p, c, ringMem, b, live, bi, lastLen, cons, iss, t, W, T);
    typename CH::Carry cinLane = c;
    if constexpr (!WS) { if (live) cin[b] = c; }
    if (WS && p.ckptSaveWarm > 0 && p.ckptOut != nullptr) {
        // main-phase steps before this block's own checkpoint: FWD the bins ahead of the next sweep's window, BWD the bins
        // behind it (the same count in both directions)
        const int sq = (p.ckptSaveWarm + p.B - 1) / p.B;
        const int done = sq * p.B - p.ckptSaveWarm;
        dma_phase<CH, true>(p, c, ringMem, b, live, bi, lastLen, cons, iss, t, W + done, T);
        if (live && b != (int64_t)bi.w) reinterpret_cast<typename CH::Carry *>(p.ckptOut)[b] = c;    // (a chain's last block is nobody's window start)
    }
    dma_phase<CH, true>(p, c, ringMem, b, live, bi, lastLen, cons, iss, t, T, T);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (WS) {
        if (p.ckptIn != nullptr) wave_local_repair<CH, typename PlainOf<CH>::type>(p, cinLane, c, b, bi, live);
        if (live) cin[b] = cinLane;
    }
    if (live) cout[b] = c;
}

// Hybrid for the fused forward chain WITH reference-layout outputs: the warm-up (no stores: 38 % of a lane's steps at genome
// scale, 71 % on a 1/8-genome shard) runs through the LDS-DMA ring, the main phase is the tile walker of k_chain_spec
// (plain loads: its ~5 stores per step could not be told apart from the DMA by a counted wait).  The ring is drained
// before the main phase, so the tile walker's barriers and LDS reads see no DMA in flight.
template <class DCH>
__global__ __launch_bounds__(64) void k_chain_spec_dmawarm_natfwd(Prm p) {
    extern __shared__ __attribute__((aligned(16))) unsigned dynMemW[];
    unsigned *ringMem = dynMemW;
    // the tiles OVERLAY the ring: it is drained before the main phase starts (same LDS footprint as the plain kernel)
    NatTilesFwd &tiles = *reinterpret_cast<NatTilesFwd *>(dynMemW);
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b < p.NB && chain_on(p, b);
    int4 bi = make_int4(0, 0, 0, 0);
    if (b < p.NB) bi = p.blk[b];
    const int lastLen = p.B;
    check_previous_stage(p, b, bi);
    typename DCH::Carry c = DCH::init_cold(p);
    typename DCH::Carry *cin = reinterpret_cast<typename DCH::Carry *>(p.carryIn);
    typename DCH::Carry *cout = reinterpret_cast<typename DCH::Carry *>(p.carryOutA);
    const int W = p.warm;
    LaneCursor<true> cons, iss;
    cons.init(p, W, b, live, bi, lastLen);
    iss.init(p, W, b, live, bi, lastLen);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // block-table loads retired before the ring starts
    for (int t0 = 0; t0 < DMA_L && t0 < W; ++t0) {
        DCH::dma_issue(p, iss.idx, ringMem + (t0 % DMA_R) * (DCH::NW * 64));
        iss.next(p, b, live, bi, lastLen);
    }
    int t = 0;
    dma_phase<DCH, false>(p, c, ringMem, b, live, bi, lastLen, cons, iss, t, W, W);     // T = W: nothing fetched beyond
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (live) cin[b] = c;
    walk_nat_fwd<typename DCH::Plain, NatTilesFwd>(p, c, b, bi.y, live, bi.z, bi.x, tiles);
    if (live) cout[b] = c;
}

// Same hybrid for the smoother chain (warm-up blocks b+q .. b+1 walked downwards through the ring, tile walker for the
// lane's own block).  Ring and tiles share the same LDS (the ring is drained first), so the footprint is the plain
// kernel's.
template <class DCH>
__global__ __launch_bounds__(64) void k_chain_spec_dmawarm_natbwd(Prm p) {
    extern __shared__ __attribute__((aligned(16))) unsigned dynMemB[];
    unsigned *ringMem = dynMemB;
    NatTiles &tiles = *reinterpret_cast<NatTiles *>(dynMemB);      // overlays the (drained) ring
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b < p.NB && chain_on(p, b);
    int4 bi = make_int4(0, 0, 0, 0);
    if (b < p.NB) bi = p.blk[b];
    int lastLen = p.B;
    if (live) lastLen = p.blk[bi.w].y;
    check_previous_stage(p, b, bi);
    typename DCH::Carry c = DCH::init_cold(p);
    typename DCH::Carry *cin = reinterpret_cast<typename DCH::Carry *>(p.carryIn);
    typename DCH::Carry *cout = reinterpret_cast<typename DCH::Carry *>(p.carryOutA);
    const int W = p.warm;
    LaneCursor<false> cons, iss;
    cons.init(p, W, b, live, bi, lastLen);
    iss.init(p, W, b, live, bi, lastLen);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int t0 = 0; t0 < DMA_L && t0 < W; ++t0) {
        DCH::dma_issue(p, iss.idx, ringMem + (t0 % DMA_R) * (DCH::NW * 64));
        iss.next(p, b, live, bi, lastLen);
    }
    int t = 0;
    dma_phase<DCH, false>(p, c, ringMem, b, live, bi, lastLen, cons, iss, t, W, W);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (live) cin[b] = c;
    walk_nat<typename DCH::Plain>(p, c, b, bi.y, live, b == bi.w, bi.x, tiles);
    if (live) cout[b] = c;
}

// Smoother chain whose INPUTS are the reference-layout xf / Pf (Prm::natIn: the forward pass wrote no blocked copies) and whose
// outputs go there too.  One lane per block as everywhere; a lane's trajectory is its warm-up window (the W bins behind its block,
// cut at the chain's end) and then its own block, walked downwards 8 steps at a time.  Both directions pass through the SAME LDS
// tiles: 8 threads fetch one full 128-byte line (8 consecutive bins of one lane's range), deposit it as [row][lane], every lane
// reads its own column, computes its 8 steps, writes xs / Ps back into the column it has just consumed, and the same 8 threads
// store full lines -- a thread reads back exactly the tile entries it deposits next, so no barrier separates store and deposit.
// The next batch's lines are requested before the current batch is computed.  CH = BwdTrendT<UF>.
// QARR: the process noise varies per bin and is read from the blocked pNoise rows inside the steps (a separate instance: a load in
// the step code makes hipcc wait for ALL outstanding loads there, i.e. for the next batch's lines it should leave in flight).
template <class CH, bool QARR>
__global__ __launch_bounds__(64) void k_smooth_natin(Prm p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char natInMem[];
    NatTiles &T = *reinterpret_cast<NatTiles *>(natInMem);
    const int lane = threadIdx.x;
    const int64_t b = (int64_t)blockIdx.x * 64 + lane;
    const bool live = b < p.NB && chain_on(p, b);
    int4 bi = make_int4(0, 0, 0, 0);
    if (b < p.NB) bi = p.blk[b];
    check_previous_stage(p, b, bi);
    typename CH::Carry c = CH::init_cold(p);
    typename CH::Carry *cin = reinterpret_cast<typename CH::Carry *>(p.carryIn);
    typename CH::Carry *cout = reinterpret_cast<typename CH::Carry *>(p.carryOutA);
    const int B = p.B, W = p.warm;          // W: multiple of 8 (host)
    // lim: bins of the trajectory this lane really has -- its block, plus the window unless the chain ends before
    int lim = 0;
    if (live) {
        if (b == (int64_t)bi.w) lim = bi.y;
        else {
            const int4 bl = p.blk[bi.w];
            const int64_t avail = ((int64_t)bl.x + bl.y) - ((int64_t)bi.x + B);
            lim = B + (int)(avail < (int64_t)W ? avail : (int64_t)W);
        }
    }
    T.gbase[lane] = bi.x;
    T.len[lane] = lim;
    T.last[lane] = (live && b == (int64_t)bi.w) ? 1 : 0;
    __syncthreads();
    float4 *natPs = reinterpret_cast<float4 *>(p.natPs), *natLag = reinterpret_cast<float4 *>(p.natLag);
    float2 *natXs = reinterpret_cast<float2 *>(p.natXs);
    const float4 qc = make_float4((float)p.Q00, (float)p.Q01, (float)p.Q10, (float)p.Q11);
    // what this thread moves, fixed for the life of the kernel: rows r of the lanes L_k it serves (kept in registers -- the tile's
    // index arrays are read once).  Every fetch is UNCONDITIONAL (a row past a lane's range is clamped to its last bin: the value
    // is deposited and never used), so the 12 loads of a batch leave back to back.
    const int rP = lane & 7, rX = (lane & 3) * 2;
    int gP[8], gX[4];
    int lP[8], lX[4], lastP[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int L = k * 8 + (lane >> 3);
        gP[k] = T.gbase[L]; lP[k] = T.len[L]; lastP[k] = T.last[L];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int L = k * 16 + (lane >> 2);
        gX[k] = T.gbase[L]; lX[k] = T.len[L];
    }
    // (named registers, not arrays behind a lambda: hipcc left such arrays in scratch memory and waited for every line to land
    // before "spilling" it there -- the prefetch was gone)
    float4 rp0, rp1, rp2, rp3, rp4, rp5, rp6, rp7, rx0, rx1, rx2, rx3;
#define CSR_FETCH_P(K, S8)                                                                          \
    {                                                                                               \
        const int top_ = lP[K] > 0 ? lP[K] - 1 : 0, s_ = (S8) + rP;                                 \
        rp##K = p.natPfIn[(int64_t)gP[K] + (s_ < top_ ? s_ : top_)];                                \
    }
    // (a pair may reach one bin past the lane's range: inside the array -- chains are padded to 64 bins -- and unused)
#define CSR_FETCH_X(K, S8)                                                                          \
    {                                                                                               \
        const int top_ = (lX[K] > 0 ? lX[K] - 1 : 0) & ~1, s_ = (S8) + rX;                          \
        rx##K = *reinterpret_cast<const float4 *>(p.natXfIn + (int64_t)gX[K] + (s_ < top_ ? s_ : top_)); \
    }
#define CSR_FETCH(S8)                                                                               \
    CSR_FETCH_P(0, S8) CSR_FETCH_P(1, S8) CSR_FETCH_P(2, S8) CSR_FETCH_P(3, S8) CSR_FETCH_P(4, S8) CSR_FETCH_P(5, S8) \
    CSR_FETCH_P(6, S8) CSR_FETCH_P(7, S8) CSR_FETCH_X(0, S8) CSR_FETCH_X(1, S8) CSR_FETCH_X(2, S8) CSR_FETCH_X(3, S8)
    const int sTop = B + W - 8;
    if (W == 0 && live) cin[b] = c;
    CSR_FETCH(sTop)
#pragma unroll 1
    for (int s8 = sTop; s8 >= 0; s8 -= 8) {
        // deposit the batch's inputs (thread -> the tile entries it fetched)
        {
            const int L0 = lane >> 3;
            T.ps[rP][L0] = rp0; T.ps[rP][8 + L0] = rp1; T.ps[rP][16 + L0] = rp2; T.ps[rP][24 + L0] = rp3;
            T.ps[rP][32 + L0] = rp4; T.ps[rP][40 + L0] = rp5; T.ps[rP][48 + L0] = rp6; T.ps[rP][56 + L0] = rp7;
            const int X0 = lane >> 2;
            T.xs[rX][X0] = make_float2(rx0.x, rx0.y); T.xs[rX + 1][X0] = make_float2(rx0.z, rx0.w);
            T.xs[rX][16 + X0] = make_float2(rx1.x, rx1.y); T.xs[rX + 1][16 + X0] = make_float2(rx1.z, rx1.w);
            T.xs[rX][32 + X0] = make_float2(rx2.x, rx2.y); T.xs[rX + 1][32 + X0] = make_float2(rx2.z, rx2.w);
            T.xs[rX][48 + X0] = make_float2(rx3.x, rx3.y); T.xs[rX + 1][48 + X0] = make_float2(rx3.z, rx3.w);
        }
        __syncthreads();
        if (s8 >= 8) { CSR_FETCH(s8 - 8) }
        __builtin_amdgcn_sched_barrier(0);          // (the lines of the next batch are requested HERE, not next to their use)
        const bool main = s8 < B;
        if (__any(live && s8 < lim)) {
#pragma unroll
            for (int u = 7; u >= 0; --u) {
                const int s = s8 + u;
                // (the lane's own column, read step by step: all eight held at once cost 48 registers the prefetch needs)
                typename CH::In in;
                in.xf = T.xs[u][lane];
                in.pf = T.ps[u][lane];
                in.q = qc;
                if (live && s < lim) {
                    if constexpr (QARR) {       // stored process noise (per-bin multipliers): blocked rows of the block that owns bin s
                        const int kq = s / B;
                        in.q = p.tQ[tidx(b + kq, s - kq * B, B)];
                    }
                    typename CH::Out o;
                    if (main) {
                        CH::template advance<true>(p, c, in, o);
                        T.xs[u][lane] = o.xs;
                        T.ps[u][lane] = o.ps;
                        if (o.hasLag) T.lag[u][lane] = o.lag;
                    } else {
                        CH::template advance<false>(p, c, in, o);
                    }
                }
            }
        }
        if (s8 == B && live) cin[b] = c;          // the carry this lane enters its own block with
        if (main) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int L = k * 8 + (lane >> 3), s = s8 + rP;
                if (s < lP[k]) {
                    natPs[(int64_t)gP[k] + s] = T.ps[rP][L];
                    if (!(lastP[k] && s == lP[k] - 1)) natLag[(int64_t)gP[k] + s] = T.lag[rP][L];     // the chain's last bin has no lag row
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int L = k * 16 + (lane >> 2), s = s8 + rX;
                if (s + 1 < lX[k]) {
                    const float2 a = T.xs[rX][L], b2 = T.xs[rX + 1][L];
                    *reinterpret_cast<float4 *>(natXs + (int64_t)gX[k] + s) = make_float4(a.x, a.y, b2.x, b2.y);
                } else if (s < lX[k]) {
                    natXs[(int64_t)gX[k] + s] = T.xs[rX][L];
                }
            }
        }
        __syncthreads();
    }
#undef CSR_FETCH
#undef CSR_FETCH_X
#undef CSR_FETCH_P
    if (live) cout[b] = c;
}

// ---------------------------------------------------------------------------------------------------------------
// Bit-exact state chain, levelTrend (validation mode k = 0).  Two float32-rounded trajectories that start one ulp apart
// stay one ulp apart for 10^2..10^5 bins, so speculation never coalesces BITWISE: the fix-up iteration of k_chain_fix
// degenerates into a front that moves one block per pass (hg38 x 32: ~1600 passes, 18 M block re-runs, 100 ms per forward
// pass).  The exact recursion is inherently sequential per chain -- so run it that way, but as cheaply as the hardware
// allows: ONE wavefront per chain; its 64 lanes fetch the gain / statistic records of the next 64 bins (one record per
// lane), the recursion itself runs on wave-uniform values (v_readlane of the record of bin j, the same FwdXTrend::step as
// everywhere else), results are collected with v_writelane and stored 64 bins at a time.  The dependent path per bin is
// ~9 fp64 instructions (~35 ns): a chromosome takes (bins x 35 ns) -- 45 ms for chr1 at 200 bp -- instead of 100 ms, and
// all chains of the batch run concurrently.  The covariance chain (short memory, validates bitwise with a 256-bin window)
// and the smoother stay speculative.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rl64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), lane);
    return words2double(lo, hi);
}
__device__ __forceinline__ float rl32(float v, int lane) {
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
}
// UNITF: F = [[1, f], [0, 1]] (what the reference's constructMatrixF always builds, core.py:2164-2176): 1 * x and
// 0 * x + x are exact, so the predicted trend is the filtered trend itself and the predicted level one fma -- the same bits
// as the general expression with five instructions less on an issue-bound wave (one wave per SIMD: every VALU instruction
// costs >= 4 cycles whether or not it is on the dependent path).
template <bool UNITF>
__global__ __launch_bounds__(64) void k_state_seq_trend(Prm p, const int64_t *chainFirstBlock, const int64_t *chainNumBlocks) {
    const int c = blockIdx.x;
    if (p.chainActive != nullptr && !p.chainActive[c]) return;
    const int lane = threadIdx.x;
    const int64_t b0 = chainFirstBlock[c], nb = chainNumBlocks[c];
    const int B = p.B;
    const int64_t n = (nb - 1) * (int64_t)B + p.blk[b0 + nb - 1].y;       // bins of this chain
    FwdXTrend::Carry cx = FwdXTrend::init_true(p);
    // records of bins [k0, k0 + 64): lane j holds bin k0 + j
    auto slot = [&](int64_t k) -> int64_t {         // k < 2^31 (batch size limit): 32-bit division
        const int ki = (int)k, q = ki / B;
        return tidx(b0 + q, ki - q * B, B);
    };
    // The records of a batch go through LDS: the recursion reads them back with two wave-uniform 16-byte LDS loads per bin
    // (broadcast, issued bins ahead by the scheduler) instead of eight v_readlane -- on a wave that is bound by instruction
    // issue that is 6 of 23 instructions per bin.
    __shared__ double4 rec[2][64];               // {gs, zbar, (double)P00pred, (double)P10pred} of the bins of a batch
    double4 cur = make_double4(0.0, 0.0, 0.0, 0.0);
    auto fetch = [&](int64_t k0, double4 &r4) {
        const int64_t k = k0 + lane;
        if (k < n) {
            const int64_t i = slot(k);
            const float4 r = p.tXin[i];
            r4 = make_double4(unpack_d(r.x, r.y), p.tSZ[i].y, (double)r.z, (double)r.w);    // widened here, 64 bins at a time
        }
    };
    auto one = [&](const double4 &r4) {
        const double g = r4.x, z = r4.y, a = r4.z, b = r4.w;
        if constexpr (UNITF) {
            const double x0 = (double)cx.x0, x1 = (double)cx.x1;
            const double xp0 = r32(fma(p.F01, x1, x0));     // == r32(fma(F01, x1, 1 * x0)); xp1 == r32(fma(1, x1, 0 * x0)) == x1
            const double dl = g * (z - xp0);
            cx.x0 = (float)fma(a, dl, xp0);
            cx.x1 = (float)fma(b, dl, x1);
        } else {
            FwdXTrend::In in;
            in.gs = g; in.zbar = z;
            in.cp = make_float2((float)a, (float)b);        // exact round trip of float32 values
            FwdXTrend::step<false>(p, cx, in, 0, 0, 0, 0);
        }
    };
    // results of a batch: every bin's (x0, x1) is written to LDS by the whole wavefront (same address, same value: one
    // instruction) and picked up lane-wise after the batch -- collecting them in registers costs four instructions per bin
    __shared__ float2 outb[64];
    fetch(0, cur);
    int buf = 0;
    for (int64_t k0 = 0; k0 < n; k0 += 64, buf ^= 1) {
        rec[buf][lane] = cur;
        __syncthreads();                                    // one wavefront: a waitcnt, no cross-wave barrier cost
        double4 nxt = make_double4(0.0, 0.0, 0.0, 0.0);
        fetch(k0 + 64, nxt);                                // in flight while this batch's recursion runs
        const int cnt = (int)((n - k0 < 64) ? (n - k0) : 64);
        const double4 *rb = rec[buf];
        if (cnt == 64) {
            // groups of four bins: the records of the next group are requested before this group's recursion starts
            double4 ra[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ra[u] = rb[u];
#pragma unroll
            for (int j0 = 0; j0 < 64; j0 += 4) {
                double4 rn[4];
                if (j0 + 4 < 64) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) rn[u] = rb[j0 + 4 + u];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    one(ra[u]);
                    outb[j0 + u] = make_float2(cx.x0, cx.x1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) ra[u] = rn[u];
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < cnt; ++j) {
                one(rb[j]);
                outb[j] = make_float2(cx.x0, cx.x1);
            }
        }
        __syncthreads();
        if (k0 + lane < n) p.tXf[slot(k0 + lane)] = outb[lane];
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// SYSTOLIC walker of the bit-exact state chain (round 3; replaces the lane-per-superblock walker above by default).
// One wavefront per superblock; its 64 lanes are 64 CONSECUTIVE BINS of the chain, so the records are read and the filtered
// state is written in the natural (reference) layout, fully coalesced, once per 64 steps -- the lane-per-superblock walker
// read one 2-KB record row per step and was bound by what a single wavefront can stream with 16 rows of look-ahead
// (scripts/ubench/stream_lat.hip: 30-40 GB/s per wavefront, 60 ns per step whatever its instruction count).
// The recursion itself runs as a shift register: in every step ALL lanes evaluate  x <- step(x of the lane below, own record);
// lane j holds the true filtered state of its bin after step j and recomputes the same bits afterwards (its input, lane j-1,
// no longer changes), lanes above j hold values that are overwritten when their turn comes.  The move between neighbouring
// lanes is a DPP wave shift (v_mov_b32_dpp wave_shr:1) whose destination keeps the CARRY in lane 0; there is no LDS, no
// v_readlane and no memory instruction on the step path: 2 DPP moves + 9 arithmetic instructions, ~20 ns per bin.
// MODE 2: F = [[1, 1], [0, 1]] (predicted level = one float32 add, see sb_step); 1: F = [[1, f], [0, 1]]; 0: any F.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_ror1(float src) {          // lane k <- lane k - 1, lane 0 <- lane 63
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, src), __builtin_bit_cast(int, src), 0x13C /* wave_ror:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shr1_keep0(float keepLane0, float src) {
    // lanes 1..63 <- src of the lane below; lane 0 keeps keepLane0 (bound_ctrl = 0: an out-of-range source leaves the destination)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keepLane0), __builtin_bit_cast(int, src),
                                                                 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
// Four superblocks per workgroup: the four wavefronts of a workgroup go to the four SIMDs of a CU, so a launch of <= 1024
// superblocks runs ONE wavefront per SIMD (single-wavefront workgroups were placed two and three to a SIMD while others
// stayed empty: 47-70 instead of 25 ns per step).
template <int MODE>
__global__ __launch_bounds__(256) void k_sb_sys(Prm p, const float4 *__restrict__ natGain, const float4 *__restrict__ natSZ,
                                                float2 *__restrict__ natXf) {
    // (the speculative pass of the PASS form, CONSENRICH_AMD_SB_ASYNC=0 and the fallback of a bailed-out single launch:
    // every superblock from the prior -- true for a chain's first one, cold otherwise; repairs: k_sb_delta)
    const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (b >= p.NB || !chain_on(p, b)) return;                 // wave-uniform
    const int lane = threadIdx.x & 63;
    const int4 bi = p.blk[b];
    using Carry = FwdXTrend::Carry;
    Carry *cin = reinterpret_cast<Carry *>(p.carryIn);
    Carry *onxt = reinterpret_cast<Carry *>(p.carryOutA);
    float c0 = (float)p.init, c1 = 0.0f;                      // FwdXTrend::init_true == init_cold
    if (lane == 0) cin[b] = Carry{c0, c1};
    const int n = bi.y;
    const int64_t g0 = (int64_t)bi.x + lane;
    const int nb = (n + 63) >> 6;
    // records of batches t and t + 1 are resident, t + 2 in flight (3 x 8 registers)
    float4 ga = natGain[g0], sa = natSZ[g0];
    float4 gb = ga, sb_ = sa;
    if (nb > 1) { gb = natGain[g0 + 64]; sb_ = natSZ[g0 + 64]; }
    float x0v = 0.0f, x1v = 0.0f;
#pragma unroll 1
    for (int t = 0; t < nb; ++t) {
        float4 gc = gb, sc = sb_;
        if (t + 2 < nb) { gc = natGain[g0 + (int64_t)(t + 2) * 64]; sc = natSZ[g0 + (int64_t)(t + 2) * 64]; }
        const double gs = unpack_d(ga.x, ga.y), zbar = unpack_d(sa.z, sa.w);
        const double p00 = (double)ga.z, p10 = (double)ga.w;
        float s0 = c0, s1 = c1;                                // shifted vectors: lane 0 = the carry into this batch
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                s0 = dpp_shr1_keep0(s0, x0v);
                s1 = dpp_shr1_keep0(s1, x1v);
                if constexpr (MODE == 2) {
                    const float xpf = s0 + s1;
                    const double xp0 = (double)xpf, x1d = (double)s1;
                    const double dl = gs * (zbar - xp0);
                    x0v = (float)fma(p00, dl, xp0);
                    x1v = (float)fma(p10, dl, x1d);
                } else if constexpr (MODE == 1) {
                    const double x1d = (double)s1;
                    const double xp0 = r32(fma(p.F01, x1d, (double)s0));
                    const double dl = gs * (zbar - xp0);
                    x0v = (float)fma(p00, dl, xp0);
                    x1v = (float)fma(p10, dl, x1d);
                } else {
                    FwdXTrend::Carry c{s0, s1};
                    FwdXTrend::In in;
                    in.gs = gs; in.zbar = zbar;
                    in.cp = make_float2(ga.z, ga.w);
                    FwdXTrend::step<false>(p, c, in, 0, 0, 0, 0);
                    x0v = c.x0; x1v = c.x1;
                }
            }
        }
        const int left = n - (t << 6);                         // bins of this batch (64 except at a chain's end)
        if (lane < left) natXf[g0 + (int64_t)t * 64] = make_float2(x0v, x1v);
        const int last = left >= 64 ? 63 : left - 1;
        c0 = rl32(x0v, last);
        c1 = rl32(x1v, last);
        ga = gb; sa = sb_; gb = gc; sb_ = sc;
    }
    if (lane == 0) onxt[b] = Carry{c0, c1};
}

// One state step of a lane from a given predecessor state with the lane's own record (the systolic walker's step body)
template <int MODE>
__device__ __forceinline__ void sys_step(const Prm &p, float s0, float s1, double gs, double zbar, double p00, double p10,
                                         float gz, float gw, float &o0, float &o1) {
    if constexpr (MODE == 2) {
        const float xpf = s0 + s1;
        const double xp0 = (double)xpf, x1d = (double)s1;
        const double dl = gs * (zbar - xp0);
        o0 = (float)fma(p00, dl, xp0);
        o1 = (float)fma(p10, dl, x1d);
    } else if constexpr (MODE == 1) {
        const double x1d = (double)s1;
        const double xp0 = r32(fma(p.F01, x1d, (double)s0));
        const double dl = gs * (zbar - xp0);
        o0 = (float)fma(p00, dl, xp0);
        o1 = (float)fma(p10, dl, x1d);
    } else {
        FwdXTrend::Carry c{s0, s1};
        FwdXTrend::In in;
        in.gs = gs; in.zbar = zbar;
        in.cp = make_float2(gz, gw);
        FwdXTrend::step<false>(p, c, in, 0, 0, 0, 0);
        o0 = c.x0; o1 = c.x1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// REPAIR passes of the bit-exact state chain in DELTA form (round 3).  A superblock that is re-run holds its previous
// trajectory S (the filtered state an earlier pass wrote, a valid trajectory of the SAME records from a slightly different
// carry) and is given the true carry T.  While the levels agree, T - S stays EXACTLY constant between the bins where a level
// rounds the other way, so instead of walking the 64 bins of a batch one after the other every lane
// k tests in parallel the hypothesis  T_{k-1} = S_{k-1} + delta  =>  step(S_{k-1} + delta, record_k) == S_k + delta .
// All lanes up to the first failing one f-1 are then PROVEN (induction from the true state at the batch's first unresolved
// bin, whose lane uses that true state itself as its predecessor), and lane f's own result is the true T_f as well -- its
// predecessor was proven -- although it differs from the hypothesis: it becomes the new base, delta is re-derived there and
// the next round starts behind it.  A round costs one parallel step + ballot + four v_readlane and resolves the bins up to
// and including the next change of delta (~12 % of the bins change it: ~9 rounds per 64 bins instead of 64 dependent steps).
// The floats S + delta are hypotheses only (an inexact sum just fails the test); what is stored is always a step() result
// of a proven predecessor, so the pass is exact by construction.  Where delta changes at every bin (the first ~100 bins
// behind a carry that was far off, and stretches where the levels flip densely) the rest of a batch is walked as a shift
// register as soon as the rounds so far have settled fewer bins each than a round is worth (the rule travels in the launch
// argument: at least `advMin` bins per round from round `advFrom` on; a round costs ~6 steps).  When T meets S bit
// for bit the rest of the superblock is already right: the wavefront stops and keeps the old carry-out.
// ---------------------------------------------------------------------------------------------------------------
#ifndef SB_DELTA_DEPTH
#define SB_DELTA_DEPTH 8
#endif
// (This is the PASS form's repair kernel -- CONSENRICH_AMD_SB_ASYNC=0 and the fallback of a bailed-out single launch; the
// barrier-free k_sb_async carries the round-4 form of the same rounds: LDS ring, h-vector, shadow step.)
template <int MODE>
__global__ __launch_bounds__(256) void k_sb_delta(Prm p, const float4 *__restrict__ natGain, const float4 *__restrict__ natSZ,
                                                  float2 *__restrict__ natXf, int which, int rule) {
    // (readfirstlane: the wavefront's index is uniform, and telling the compiler so keeps the superblock's table entry, the
    // carries and the whole round control -- pos, f, delta -- in scalar registers)
    const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (b >= p.NB || !chain_on(p, b)) return;
    const int lane = threadIdx.x & 63;
    const int advMin = (rule >> 8) & 0xff, advFrom = (rule >> 16) & 0xff;     // the fallback rule (see batch())
    const int4 bi = p.blk[b];
    using Carry = FwdXTrend::Carry;
    Carry *cin = reinterpret_cast<Carry *>(p.carryIn);
    const Carry *ocur = reinterpret_cast<const Carry *>(which ? p.carryOutB : p.carryOutA);
    Carry *onxt = reinterpret_cast<Carry *>(which ? p.carryOutA : p.carryOutB);
    if (b == (int64_t)bi.z) {                             // a chain's first superblock started from the true prior
        if (lane == 0) onxt[b] = ocur[b];
        return;
    }
    const Carry prev = ocur[b - 1], mine = cin[b], oldOut = ocur[b];
    if ((((f2u(prev.x0) ^ f2u(mine.x0)) | (f2u(prev.x1) ^ f2u(mine.x1))) == 0u)) {
        if (lane == 0) onxt[b] = oldOut;
        return;
    }
    if (lane == 0) cin[b] = prev;
    if (p.sbDbg != nullptr && lane == 0) atomicAdd(p.sbDbg, 1ull);
    const long long dbgT0 = p.sbDbg != nullptr ? wall_clock64() : 0;
    unsigned dbgFb = 0, dbgRounds = 0, dbgBatches = 0;
    float t0 = prev.x0, t1 = prev.x1;                          // TRUE state at the bin before the next unresolved one
    float sc0 = mine.x0, sc1 = mine.x1;                        // the old trajectory's state at the bin before the batch
    const int n = bi.y;
    const int64_t g0 = (int64_t)bi.x + lane;
    const int nb = (n + 63) >> 6;
    // A batch in delta form takes a few hundred cycles -- far less than a trip to HBM -- so the records and the old trajectory
    // of SB_DELTA_DEPTH batches ahead are kept in flight in a ring of registers (statically indexed: the batch loop is unrolled
    // over the ring's slots).
    constexpr int DEPTH = SB_DELTA_DEPTH;
    float4 rg[DEPTH], rs[DEPTH];
    float2 ro[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
        const int tt = u < nb ? u : nb - 1;
        rg[u] = natGain[g0 + (int64_t)tt * 64];
        rs[u] = natSZ[g0 + (int64_t)tt * 64];
        ro[u] = natXf[g0 + (int64_t)tt * 64];
    }
    bool merged = false;
    auto batch = [&](int t, const float4 &ga, const float4 &sa, const float2 &so) {
        const double gs = unpack_d(ga.x, ga.y), zbar = unpack_d(sa.z, sa.w);
        const double p00 = (double)ga.z, p10 = (double)ga.w;
        const int left = min(64, n - (t << 6));
        const unsigned long long leftMask = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
        // old trajectory at the previous bin: lane 0 <- sc (the bin before the batch)
        const float sp0 = dpp_shr1_keep0(sc0, so.x), sp1 = dpp_shr1_keep0(sc1, so.y);
        const float tb0 = t0, tb1 = t1;                        // true carry into the batch (for the fallback)
        float to0 = so.x, to1 = so.y;
        float d0 = t0 - sc0, d1 = t1 - sc1;
        int pos = 0, rounds = 0;
        bool fallback = false;
#pragma unroll 1
        while (pos < left) {
            // a round costs about six sequential steps: where the rounds so far have settled fewer than advMin bins each, the
            // rest of the batch is walked (default: give up after 20 rounds; see csr_ctx::sbAdvMin)
            if (rounds >= advFrom && pos < advMin * rounds) { fallback = true; break; }
            ++rounds;
            const bool base = lane == pos;
            const float q0 = base ? t0 : sp0 + d0, q1 = base ? t1 : sp1 + d1;
            float n0, n1;
            sys_step<MODE>(p, q0, q1, gs, zbar, p00, p10, ga.z, ga.w, n0, n1);
            const float c0 = so.x + d0, c1 = so.y + d1;
            // lane masks straight from the comparisons (no boolean is materialised)
            const unsigned long long okm = __builtin_amdgcn_uicmp(f2u(n0), f2u(c0), 32 /* ICMP_EQ */) &
                                           __builtin_amdgcn_uicmp(f2u(n1), f2u(c1), 32);
            const unsigned long long fail = ~okm & (~0ull << pos) & leftMask;
            const int f = fail ? (int)__ffsll((long long)fail) - 1 : left;
            const int hi = f < left ? f : left - 1;            // bins pos .. hi are settled by this round: their step() results stand
            if (lane >= pos && lane <= hi) { to0 = n0; to1 = n1; }
            t0 = rl32(n0, hi);
            t1 = rl32(n1, hi);
            d0 = t0 - rl32(so.x, hi);
            d1 = t1 - rl32(so.y, hi);
            pos = hi + 1;
        }
        if (fallback) {
            // delta changes at (nearly) every bin here: walk the unsettled bins pos .. left-1 as a shift register.  Lane 0 keeps
            // the batch's true carry, the settled lanes hold their true states and recompute the same bits from true
            // predecessors; lane pos + j is right after step j + 1.
            float s0 = tb0, s1 = tb1, x0v = to0, x1v = to1;
#pragma unroll 1
            for (int q = pos; q < left; ++q) {
                s0 = dpp_shr1_keep0(s0, x0v);
                s1 = dpp_shr1_keep0(s1, x1v);
                sys_step<MODE>(p, s0, s1, gs, zbar, p00, p10, ga.z, ga.w, x0v, x1v);
            }
            to0 = x0v; to1 = x1v;
            t0 = rl32(x0v, left - 1);
            t1 = rl32(x1v, left - 1);
        }
        if (lane < left) natXf[g0 + (int64_t)t * 64] = make_float2(to0, to1);
        sc0 = rl32(so.x, left - 1);
        sc1 = rl32(so.y, left - 1);
        // the true trajectory has met the old one bit for bit: everything behind this bin is already right
        merged = ((f2u(t0) ^ f2u(sc0)) | (f2u(t1) ^ f2u(sc1))) == 0u;
        dbgFb += fallback ? 1u : 0u; dbgRounds += (unsigned)rounds; dbgBatches += 1u;
    };
#pragma unroll 1
    for (int tg = 0; tg < nb && !merged; tg += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int t = tg + u;
            if (t < nb && !merged) {
                batch(t, rg[u], rs[u], ro[u]);
                const int tn = t + DEPTH;
                if (tn < nb) {
                    rg[u] = natGain[g0 + (int64_t)tn * 64];
                    rs[u] = natSZ[g0 + (int64_t)tn * 64];
                    ro[u] = natXf[g0 + (int64_t)tn * 64];
                }
            }
        }
    }
    if (lane == 0) {
        onxt[b] = merged ? oldOut : Carry{t0, t1};
        atomicAdd(p.rerunCount, 1u);
        atomicAdd(p.rerunCountPass, 1u);
        if (p.sbDbg != nullptr) {
            const unsigned long long dt = (unsigned long long)(wall_clock64() - dbgT0);
            // slowest superblock of the launch sequence: ticks (10 ns), packed with its batches / rounds / fallbacks
            const unsigned long long packed = (dt << 40) | ((unsigned long long)(dbgBatches & 0xfff) << 28) | ((unsigned long long)(dbgRounds & 0xffff) << 12) | (dbgFb & 0xfff);
            atomicMax(p.sbDbg + 5, packed);
            atomicAdd(p.sbDbg + 6, dt);
            atomicAdd(p.sbDbg + 1, (unsigned long long)dbgBatches);
            atomicAdd(p.sbDbg + 2, (unsigned long long)dbgRounds);
            atomicAdd(p.sbDbg + 3, (unsigned long long)dbgFb);
            if (merged) atomicAdd(p.sbDbg + 4, 1ull);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The bit-exact state chain WITHOUT a barrier between passes (round 3).  In the pass form above every repair pass costs its
// slowest superblock (a stretch where the levels flip densely takes three times the mean) while the front that matters --
// the true trajectory working its way down a chain until it meets what lies ahead -- only waits for its own superblock.  Here
// one launch does the whole chain: a wavefront owns a superblock for the life of the kernel; it walks it once from the cold
// prior (the speculative pass), publishes its carry-out, and then re-runs it in delta form (same rounds as k_sb_delta)
// whenever its PREDECESSOR publishes a carry-out that differs from the carry it last started from.  A run in flight is
// abandoned for a newer carry at a batch boundary; the stored trajectory is then a sequence of valid pieces and the delta
// rounds need no more than that (a hypothesis across a seam just fails) -- only the early exit "the new trajectory has met the
// old one" has to wait until the run is past the last seam (`brk`), because what lies behind the meeting point must be one
// piece ending in the published carry-out.  "Final" travels down a chain with the carries: a chain's first superblock is
// final after its walk, a superblock is final when a run that started from a final carry ends.  The result is the same fixed
// point as the pass form's: the sequential recursion, bit for bit.
// Progress: superblocks are handed out by a ticket taken when a workgroup STARTS, so the predecessor a wavefront waits for
// belongs to a workgroup that is already running (or to its own); nothing depends on all workgroups being resident.  Every
// wait is bounded: a wavefront that polls `spinLimit` times without news raises the bail-out flag and leaves, every waiting
// wavefront leaves when it sees the flag, and the host then runs the pass form instead.
// Publication: the carry (two floats in one 64-bit word) is stored first, then {version, final} with release order; a reader
// that sees a version with acquire order reads a carry at least that new.
// ---------------------------------------------------------------------------------------------------------------
struct SbAsync {
    unsigned long long *carry;      // [NB] published carry-out
    unsigned long long *vf;         // [NB] (version << 1) | final; 0 = nothing published yet
    unsigned int *ctl;              // [0] ticket, [1] bail-out flag, [2] delta runs, [3] abandoned runs
    unsigned int *hostDone;         // [chains], host-visible, or nullptr: set to 1 when a chain's last superblock is final
    int spinLimit;
};
// a chain is final (its whole filtered state stands in the reference layout): tell the host, which may start that chain's
// smoother / residuals on another stream while other chains are still being repaired
__device__ __forceinline__ void sb_chain_done(const Prm &p, const SbAsync &a, int64_t b, const int4 &bi, int lane) {
    if (a.hostDone != nullptr && b == (int64_t)bi.w) {          // (wave-uniform)
        // every lane stored a part of the chain's last track: the WHOLE wavefront releases its stores at system scope, then
        // lane 0 publishes (the scoped memory model orders a release only behind the releasing thread's own writes)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (lane == 0) __hip_atomic_store(a.hostDone + p.blkChain[b], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void sb_publish(const SbAsync &a, int64_t b, int lane, float o0, float o1, unsigned ver, bool fin) {
    // the trajectory this carry-out belongs to was stored by all 64 lanes: the whole wavefront releases at agent scope before
    // lane 0 publishes {carry, version} (round-3 review: lane 0's release alone covers only lane 0's writes in the scoped model;
    // it worked because s_waitcnt is per wavefront)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) {
        __hip_atomic_store(a.carry + b, ((unsigned long long)f2u(o1) << 32) | f2u(o0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.vf + b, ((unsigned long long)ver << 1) | (fin ? 1ull : 0ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// LDS ring of the repair runs: per wavefront two halves + one row for the poll word.  A half holds a GROUP of SBG batches
// as three regions in the natural layout, each filled by 1-KiB DMAs (64 lanes x 16 B): the gain records (SBG KiB), the
// statistics records {S0, zbar} (SBG KiB; zbar is read back at +8) and the stored trajectory (SBG/2 KiB: one DMA brings
// the float2 states of TWO batches): 2.5 DMAs per batch.
#ifndef SB_GROUP
#define SB_GROUP 6
#endif
constexpr int SBG = SB_GROUP;
static_assert(SBG % 2 == 0, "the stored trajectory travels two batches per DMA");
constexpr int SB_HALF_W = SBG * 256 + SBG * 256 + SBG * 128;     // words per half
constexpr int SB_WAVE_W = 2 * SB_HALF_W + 64;
constexpr size_t SB_ASYNC_LDS = 4 * (size_t)SB_WAVE_W * sizeof(unsigned);
static_assert(SBG * 1024 + 8 < 65536 && 2 * SBG * 1024 < 65536, "immediate offsets of the ring's reads");
static_assert(2 * SBG + SBG / 2 + 1 + SBG <= 63, "the group's DMAs and stores must fit the 6-bit vmcnt");
__device__ __forceinline__ unsigned lds_rd32_wait(const unsigned *q) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_off(q)) : "memory");
    return v;
}
// DBG: the instance with the section timers and counters of CONSENRICH_AMD_SB_DEBUG (they cost a dozen instructions and five
// branches per batch: not in the production instance)
template <int MODE, bool DBG = false>
__global__ __launch_bounds__(256) void k_sb_async(Prm p, const float4 *__restrict__ natGain, const float4 *__restrict__ natSZ,
                                                  float2 *__restrict__ natXf, SbAsync a) {
    extern __shared__ __attribute__((aligned(16))) unsigned sbRing[];
    __shared__ unsigned int sTicket;
    if (threadIdx.x == 0) sTicket = atomicAdd(a.ctl, 1u);
    __syncthreads();
    const int64_t b = (int64_t)sTicket * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (b >= p.NB || !chain_on(p, b)) return;
    const int lane = threadIdx.x & 63;
    const int4 bi = p.blk[b];
    const bool first = b == (int64_t)bi.z;
    const int n = bi.y;
    const int64_t g0 = (int64_t)bi.x + lane;
    const int nb = (n + 63) >> 6;
    // ---- the speculative walk: from the prior (true for a chain's first superblock, cold otherwise)
    float out0 = (float)p.init, out1 = 0.0f;
    if (DBG && b == 0 && lane == 0) p.sbDbg[0] = (unsigned long long)wall_clock64();
    {
        float4 ga = natGain[g0], sa = natSZ[g0];
        float4 gb = ga, sb_ = sa;
        if (nb > 1) { gb = natGain[g0 + 64]; sb_ = natSZ[g0 + 64]; }
        float x0v = 0.0f, x1v = 0.0f;
#pragma unroll 1
        for (int t = 0; t < nb; ++t) {
            float4 gc = gb, sc = sb_;
            if (t + 2 < nb) { gc = natGain[g0 + (int64_t)(t + 2) * 64]; sc = natSZ[g0 + (int64_t)(t + 2) * 64]; }
            const double gs = unpack_d(ga.x, ga.y), zbar = unpack_d(sa.z, sa.w);
            const double p00 = (double)ga.z, p10 = (double)ga.w;
            float s0 = out0, s1 = out1;
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    s0 = dpp_shr1_keep0(s0, x0v);
                    s1 = dpp_shr1_keep0(s1, x1v);
                    sys_step<MODE>(p, s0, s1, gs, zbar, p00, p10, ga.z, ga.w, x0v, x1v);
                }
            }
            const int left = n - (t << 6);
            if (lane < left) natXf[g0 + (int64_t)t * 64] = make_float2(x0v, x1v);
            const int last = left >= 64 ? 63 : left - 1;
            out0 = rl32(x0v, last);
            out1 = rl32(x1v, last);
            ga = gb; sa = sb_; gb = gc; sb_ = sc;
        }
    }
    unsigned ver = 1;
    sb_publish(a, b, lane, out0, out1, ver, first);
    if (DBG && lane == 0) atomicMax(p.sbDbg + 1, (unsigned long long)wall_clock64());
    if (first) {
        sb_chain_done(p, a, b, bi, lane);
        if (DBG && lane == 0) {
            atomicMax(p.sbDbg + 2, (unsigned long long)wall_clock64());
            if (b == (int64_t)bi.w) p.sbDbg[8 + p.blkChain[b]] = (unsigned long long)wall_clock64();     // a one-superblock chain is final
        }
        return;
    }
    // ---- repairs
    // Round 4: the records and the stored trajectory of a run reach the rounds through a per-wavefront LDS ring filled by
    // LDS-DMA, a GROUP of SBG batches at a time into one of two halves, behind hand-counted waits.  (Round 3 kept a ring of
    // registers; hipcc could not count its loads across the conditional refills and waited for ALL of them at every batch --
    // s_waitcnt vmcnt(1) -- so a batch cost max(rounds, one trip to HBM) ~ 1 us.)  In program order a full group issues
    // [its successor's 5 x SBG record DMAs + the poll DMA] and then exactly SBG stores (one per batch), so vmcnt(SBG) at the top
    // of the next group says that group's records and its poll word have landed.  The predecessor's version word travels the
    // same way (one sc1 DMA per group): nothing the compiler would wait for sits between two groups.
    float cin0 = (float)p.init, cin1 = 0.0f;        // the carry the latest run started from
    float trj0 = cin0, trj1 = cin1;                 // the carry the stored batch 0 was computed from (hypotheses only)
    unsigned seen = 0, runs = 0, aborts = 0;
    unsigned dbgBatches = 0, dbgRounds = 0, dbgFb = 0;
    unsigned long long dbgTicks = 0;
    unsigned long long dbgSec[4] = {0, 0, 0, 0};      // s_memtime ticks: group top, batch prologue, rounds, batch epilogue (debug only)
    constexpr bool dbgOn = DBG;
    int brk = 0;                                    // first bin of the stored trajectory's last piece
    unsigned *const ring = sbRing + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * SB_WAVE_W;
    unsigned *const pollRow = ring + 2 * SB_HALF_W;
    const unsigned ldsLaneG = lds_off(ring) + (unsigned)lane * 16u, ldsLaneX = lds_off(ring) + (unsigned)lane * 8u + 2u * SBG * 1024u;
    const unsigned long long *pvf = a.vf + (b - 1), *pcarry = a.carry + (b - 1);
    const int ng = (nb + SBG - 1) / SBG;
    const int64_t bx = (int64_t)__builtin_amdgcn_readfirstlane(bi.x);
    auto issue_group = [&](int g) {
        unsigned *half = ring + (size_t)(g & 1) * SB_HALF_W;
#pragma unroll
        for (int u = 0; u < SBG; ++u) {
            const int t = g * SBG + u;
            const int64_t i = bx + (int64_t)(t < nb ? t : nb - 1) * 64 + lane;   // (a batch beyond the end re-reads the last one: the count of DMAs per group is fixed)
            dma16(natGain + i, half + u * 256);
            dma16(natSZ + i, half + SBG * 256 + u * 256);
            if ((u & 1) == 0)       // lane k brings bins 2k, 2k + 1 of the 128 bins from batch t on (the array ends in 64 spare bins)
                dma16(reinterpret_cast<const float4 *>(natXf + (bx + (int64_t)(t < nb ? t : nb - 1) * 64)) + lane, half + 2 * SBG * 256 + (u >> 1) * 256);
        }
        // the predecessor's {version, final} word (low half: versions stay far below 2^31), agent scope
        __builtin_amdgcn_global_load_lds((gbl_cvptr)pvf, (lds_vptr)pollRow, 4, 0, 16 /* sc1 */);
    };
    // a group whose SBG batches all exist: per region one or two base addresses, the batch in the immediate offset (the generic
    // form above spends ~100 instructions on its 16 addresses; this one ~35)
    const float4 *const laneGain = natGain + bx + lane, *const laneSZ = natSZ + bx + lane;
    const float4 *const laneXf = reinterpret_cast<const float4 *>(natXf + bx) + lane;
    auto issue_group_full = [&](int g) {
        unsigned *half = ring + (size_t)(g & 1) * SB_HALF_W;
        const int64_t r0 = (int64_t)g * (SBG * 64);               // records (bins) before the group
        constexpr int LO = SBG < 4 ? SBG : 4, HI = SBG - LO;      // rows reachable from the first base, from the second (+ 4 KiB)
        dma16_rows(laneGain + r0, half, std::make_integer_sequence<int, LO>{});
        if constexpr (HI > 0) dma16_rows(laneGain + r0 + 256, half + 1024, std::make_integer_sequence<int, HI>{});
        dma16_rows(laneSZ + r0, half + SBG * 256, std::make_integer_sequence<int, LO>{});
        if constexpr (HI > 0) dma16_rows(laneSZ + r0 + 256, half + SBG * 256 + 1024, std::make_integer_sequence<int, HI>{});
        // (the stored trajectory: 8 B per bin, a row = two batches)
        dma16_rows(laneXf + (r0 >> 1), half + 2 * SBG * 256, std::make_integer_sequence<int, SBG / 2>{});
        __builtin_amdgcn_global_load_lds((gbl_cvptr)pvf, (lds_vptr)pollRow, 4, 0, 16 /* sc1 */);
    };
    auto issue = [&](int g) {
        if ((g + 1) * SBG <= nb) issue_group_full(g);
        else issue_group(g);
    };
    for (;;) {
        // wait for news from the predecessor
        unsigned long long vf;
        for (unsigned spins = 0;; ++spins) {
            vf = uni64(__hip_atomic_load(pvf, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
            if ((unsigned)(vf >> 1) != seen) break;
            if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return;
            if (spins > (unsigned)a.spinLimit) {
                if (lane == 0) __hip_atomic_store(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __builtin_amdgcn_s_sleep(64);
        }
        seen = (unsigned)(vf >> 1);
        bool runFinal = (vf & 1ull) != 0ull;
        unsigned long long cw = uni64(__hip_atomic_load(pcarry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (cw == (((unsigned long long)f2u(cin1) << 32) | f2u(cin0))) {
            if (runFinal) { sb_publish(a, b, lane, out0, out1, ++ver, true); break; }
            continue;
        }
        // a run (restarted from batch 0 whenever a newer carry arrives while it is in flight)
        bool completed = false, merged = false;
        for (;;) {
            cin0 = __uint_as_float((unsigned)cw); cin1 = __uint_as_float((unsigned)(cw >> 32));
            ++runs;
            float qi0 = cin0, qi1 = cin1;                  // lane 0: TRUE state at the bin before the next batch
            float d0 = cin0 - trj0, d1 = cin1 - trj1;      // the hypothesis T - S (trj: the stored trajectory's state at the bin before batch 0)
            bool newer = false;
            unsigned long long cwNew = cw;
            int done = 0;
            const long long dbgT0 = DBG ? wall_clock64() : 0;
            // one 64-bin batch of the run.  Carried from batch to batch: qi (lane 0: the TRUE state at the bin before the batch),
            // d (the current hypothesis T - S as two scalars), `done`, `merged`.
            auto batch = [&](auto fullTag, int g, int u) {
                constexpr bool FULL = decltype(fullTag)::value;
                const int t = g * SBG + u;
                const long long dbgB = dbgOn ? (long long)__builtin_readcyclecounter() : 0;
                // (two vector adds form the lane's addresses; zbar sits SBG KiB + 8 B behind the lane's gain record)
                const unsigned halfOff = (unsigned)((g & 1) * SB_HALF_W * 4);
                const unsigned aG = ldsLaneG + halfOff + (unsigned)(u * 1024), aX = ldsLaneX + halfOff + (unsigned)(u * 512);
                uint4 gr;
                uint2 zw, sx;
                asm volatile("ds_read_b128 %0, %3\n\tds_read_b64 %1, %3 offset:%5\n\tds_read_b64 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(gr), "=&v"(zw), "=&v"(sx) : "v"(aG), "v"(aX), "n"(SBG * 1024 + 8) : "memory");
                const double gs = words2double(gr.x, gr.y), zbar = words2double(zw.x, zw.y);
                const float gz = __uint_as_float(gr.z), gw = __uint_as_float(gr.w);
                const double p00 = (double)gz, p10 = (double)gw;
                const float so0 = __uint_as_float(sx.x), so1 = __uint_as_float(sx.y);
                const int left = FULL ? 64 : min(64, n - (t << 6));
                const unsigned long long leftMask = (FULL || left >= 64) ? ~0ull : ((1ull << (left & 63)) - 1ull);
                // h: lane k holds the TRUE state of its bin once settled, the hypothesis S_k + delta otherwise.  Every
                // lane's predecessor is a wave shift of h (lane 0: the true carry), so settled lanes recompute their own
                // bits and pass; the first failing lane f had a proven predecessor: its step result n_f is true.  The
                // second step m = step(shift(n)) is issued while the scalar unit digests the comparison: m_k = n_k up
                // to f, and m_{f+1} is true as well (from n_f), so the round settles through f + 1 and re-bases there.
                // What a round costs is its INSTRUCTION COUNT: one wavefront per SIMD issues one instruction every
                // ~5 ticks whatever its type or dependencies (scripts/ubench/issue_cost.hip).  Hence: ONE backward branch
                // per round, the re-base computed unconditionally (a sentinel bit at the batch's last bin makes a round
                // without a failing lane re-base there, where m = n: nothing changes), rare paths behind the loop, the
                // shifted vectors loop-carried so that their lane 0 keeps the true carry without a copy per round, one
                // 64-bit comparison of the packed pair.
                float h0 = so0 + d0, h1 = so1 + d1;
                float q0 = qi0, q1 = qi1, r0 = qi0, r1 = qi1;         // lane 0 of the shifted vectors: the true carry, for the whole batch
                const unsigned long long sentinel = 1ull << (left - 1);
                int rounds = 0, s;
                unsigned long long fail;
                const long long dbgC = dbgOn ? (long long)__builtin_readcyclecounter() : 0;
                if (dbgOn) dbgSec[1] += (unsigned long long)(dbgC - dbgB);
#pragma unroll 1
                do {
                    q0 = dpp_shr1_keep0(q0, h0); q1 = dpp_shr1_keep0(q1, h1);
                    float n0, n1;
                    sys_step<MODE>(p, q0, q1, gs, zbar, p00, p10, gz, gw, n0, n1);
                    const unsigned long long ne = __builtin_amdgcn_uicmpl(((unsigned long long)f2u(n1) << 32) | f2u(n0),
                                                                          ((unsigned long long)f2u(h1) << 32) | f2u(h0), 33 /* ICMP_NE */);
                    __builtin_amdgcn_sched_barrier(0);
                    r0 = dpp_shr1_keep0(r0, n0); r1 = dpp_shr1_keep0(r1, n1);
                    float m0, m1;
                    sys_step<MODE>(p, r0, r1, gs, zbar, p00, p10, gz, gw, m0, m1);
                    const float e0 = m0 - so0, e1 = m1 - so1;
                    __builtin_amdgcn_sched_barrier(0);
                    fail = FULL ? ne : (ne & leftMask);
                    // bins 0 .. s are settled: m holds their true states (s = f + 1 behind a failing lane f)
                    s = min((int)__builtin_ctzll(fail | sentinel) + 1, left - 1);
                    d0 = rl32(e0, s); d1 = rl32(e1, s);
                    const bool le = lane <= s;
                    h0 = le ? m0 : so0 + d0;
                    h1 = le ? m1 : so1 + d1;
                    if constexpr (DBG) ++rounds;
                    // (no give-up rule: a round settles at least two more bins -- the lane behind the settled ones has a
                    // proven predecessor, so the next failing lane lies beyond it -- i.e. at most 32 rounds per batch, which
                    // costs what the round-3 rule "20 rounds, then walk the rest" cost where the levels flip densely)
                } while (fail != 0ull);
                const long long dbgD = dbgOn ? (long long)__builtin_readcyclecounter() : 0;
                if (dbgOn) dbgSec[2] += (unsigned long long)(dbgD - dbgC);
                if (FULL || lane < left) natXf[g0 + (int64_t)t * 64] = make_float2(h0, h1);
                if constexpr (DBG) { dbgRounds += (unsigned)rounds; ++dbgBatches; if (rounds > 20) ++dbgFb; }
                // the batch's last bin: its true state becomes lane 0 of the next batch's shifted vectors (a full batch: one wave
                // rotation per component), T - S there the next hypothesis (after the loop d is the delta of lane s <= left - 1; the
                // settled lanes hold m, so it is read again at the last bin), and T == S there means the run has met the stored
                // trajectory: what lies behind is right and ends in `out`, provided the stored trajectory is one piece from here
                const unsigned long long eq = __builtin_amdgcn_uicmpl(((unsigned long long)f2u(h1) << 32) | f2u(h0),
                                                                      ((unsigned long long)f2u(so1) << 32) | f2u(so0), 32 /* ICMP_EQ */);
                const float dv0 = h0 - so0, dv1 = h1 - so1;
                d0 = rl32(dv0, left - 1); d1 = rl32(dv1, left - 1);
                if constexpr (FULL) {
                    qi0 = dpp_ror1(h0); qi1 = dpp_ror1(h1);
                } else {
                    qi0 = rl32(h0, left - 1); qi1 = rl32(h1, left - 1);
                }
                done = (t << 6) + left;                    // bins of the superblock settled by this run
                merged = done > brk && ((eq >> (left - 1)) & 1ull) != 0ull;
                if (dbgOn) dbgSec[3] += (unsigned long long)((long long)__builtin_readcyclecounter() - dbgD);
            };
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wavefront's earlier stores of the trajectory are out; the ring is free
            issue(0);
#pragma unroll 1
            for (int g = 0; g < ng && !merged && !newer; ++g) {
                long long dbgA = dbgOn ? (long long)__builtin_readcyclecounter() : 0;
                if (g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SBG) : "memory");
                // news from the predecessor (as of the moment this group's records were asked for)?
                const unsigned pvLo = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_rd32_wait(pollRow + lane));
                if ((pvLo >> 1) != (seen & 0x7fffffffu)) {
                    seen = pvLo >> 1;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    cwNew = uni64(__hip_atomic_load(pcarry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (cwNew == cw) runFinal = runFinal || (pvLo & 1u) != 0u;
                    else {                                              // abandon: the stored trajectory gets a seam here
                        newer = true;
                        runFinal = (pvLo & 1u) != 0u;
                        if (done > brk) brk = done;
                        break;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < ng) issue(g + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (g == 0) { trj0 = cin0; trj1 = cin1; }          // (this group rewrites batch 0 of the stored trajectory from cin)
                if (dbgOn) { const long long nowA = (long long)__builtin_readcyclecounter(); dbgSec[0] += (unsigned long long)(nowA - dbgA); }
                // (a group whose SBG batches are all full -- every group but a superblock's last -- runs the instance of the batch
                // with left = 64 folded in: no length mask in the round, an unpredicated store, constant lane indices)
                if ((g * SBG + SBG) * 64 <= n) {
#pragma unroll
                    for (int u = 0; u < SBG; ++u)
                        if (!merged) batch(std::true_type{}, g, u);
                } else {
#pragma unroll
                    for (int u = 0; u < SBG; ++u)
                        if (g * SBG + u < nb && !merged) batch(std::false_type{}, g, u);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no DMA of this run still lands in the ring
            if (!newer && done >= n) { completed = true; out0 = rl32(qi0, 0); out1 = rl32(qi1, 0); brk = 0; }      // the run reached the superblock's end
            if (DBG) dbgTicks += (unsigned long long)(wall_clock64() - dbgT0);
            if (!newer) break;
            ++aborts;
            cw = cwNew;
        }
        (void)completed;
        sb_publish(a, b, lane, out0, out1, ++ver, runFinal);
        if (runFinal) break;
    }
    sb_chain_done(p, a, b, bi, lane);       // (every path that reaches this point has published "final")
    if (lane == 0) {
        atomicAdd(a.ctl + 2, runs);
        atomicAdd(a.ctl + 3, aborts);
        if (DBG) {
            atomicMax(p.sbDbg + 2, (unsigned long long)wall_clock64());
            atomicAdd(p.sbDbg + 3, (unsigned long long)dbgBatches);
            atomicAdd(p.sbDbg + 4, (unsigned long long)dbgRounds);
            atomicAdd(p.sbDbg + 5, (unsigned long long)dbgFb);
            atomicAdd(p.sbDbg + 6, dbgTicks);
            atomicMax(p.sbDbg + 7, (dbgTicks << 24) | (unsigned long long)(dbgBatches & 0xffffff));
            for (int q = 0; q < 4; ++q) atomicAdd(p.sbDbg + 8 + p.nchains + q, dbgSec[q]);
            if (b == (int64_t)bi.w) p.sbDbg[8 + p.blkChain[b]] = (unsigned long long)wall_clock64();     // the chain is final
        }
    }
}

// natural float2 track -> the batch's blocked layout through LDS tiles (32 steps x 64 blocks per workgroup): coalesced on
// both sides (the per-slot gather of k_import_f32 reads one 128-byte line per 8 bytes it needs)
// (g0: first wavefront-group of 64 blocks the launch covers -- a launch for some chains only starts at their blocks)
template <class V>
__global__ __launch_bounds__(256) void k_import_tiled(Prm p, const V *__restrict__ nat, V *__restrict__ dst, int64_t g0) {
    __shared__ V tile[32][65];
    const int tilesPerGroup = p.B >> 5;
    const int64_t G = g0 + blockIdx.x / tilesPerGroup;
    const int s0 = (int)(blockIdx.x % tilesPerGroup) << 5;
    const int t = threadIdx.x;
    const int r = t >> 5, si = t & 31;
    // (the eight block records first, then the eight rows: all in flight together -- a workgroup used to walk them one dependent
    // pair of loads after the other, 60 us for a launch that moves 20 MB)
    int4 bis[8];
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int64_t b = G * 64 + pass * 8 + r;
        bis[pass] = make_int4(0, 0, 0, 0);
        if (b < p.NB && chain_on(p, b)) bis[pass] = p.blk[b];
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        V v{};
        if (s0 + si < bis[pass].y) v = nat[(int64_t)bis[pass].x + s0 + si];
        tile[si][pass * 8 + r] = v;
    }
    __syncthreads();
    const int lane = t & 63, r0 = t >> 6;
    const int64_t rowBase = (G * (int64_t)p.B + s0) * 64;
    const int64_t bw = G * 64 + lane;
    int len = 0;                                               // slots of inactive chains / past a block's end stay untouched
    if (bw < p.NB && chain_on(p, bw)) len = p.blk[bw].y;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + r0;
        if (s0 + row < len) dst[rowBase + (int64_t)row * 64 + lane] = tile[row][lane];
    }
}

// Validation / fix-up pass: a block whose recorded carry-in differs from its neighbour's current carry-out is re-run
// from that carry.  Iterated (ping-pong outCur/outNext) until no block re-runs: the fixed point is the sequential
// recursion.  which = 0: read A write B; 1: read B write A.
template <class CH, bool NAT = false, bool PCQ = false>
__global__ __launch_bounds__(64) void k_chain_fix(Prm p_, int which) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const Prm p = lane_model_if<PCQ>(p_, b);
    const bool live = b < p.NB && chain_on(p, b);
    using Carry = typename CH::Carry;
    Carry *cin = reinterpret_cast<Carry *>(p.carryIn);
    const Carry *ocur = reinterpret_cast<const Carry *>(which ? p.carryOutB : p.carryOutA);
    Carry *onxt = reinterpret_cast<Carry *>(which ? p.carryOutA : p.carryOutB);
    if (p.debugForce & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    int4 bi = make_int4(0, 0, 0, 0);
    if (b < p.NB) bi = p.blk[b];
    const int64_t bfirst = bi.z, blast = bi.w;
    // NOTE: written as straight-line code on purpose.  hipcc 7.2 mis-compiled the natural form
    //   `Carry c = cold; if (...) { if (!same(prev, cin[b])) { c = prev; ... } }`
    // (the divergent struct phi kept the cold c00/c01 on the rerun path); same() therefore uses non-short-circuit
    // bit operations and the carry is assigned unconditionally.  tests/test_gpu_parity.py::
    // test_speculative_blocks_equal_sequential_recursion pins this.
    const bool edge = CH::FWD ? (b == bfirst) : (b == blast);
    const bool check = live && !edge;
    const int64_t nbr = check ? (CH::FWD ? b - 1 : b + 1) : 0;
    const int64_t self = live ? b : 0;
    const Carry prev = ocur[nbr];
    const Carry mine = cin[self];
    const bool rerun = check && (((p.debugForce & 1) != 0) | !CH::same(p, prev, mine));
    Carry c = prev;
    if (rerun) cin[b] = prev;
    // (loaded here rather than held across the comparison: a held copy becomes an alloca that hipcc promotes to LDS,
    // and the LDS allocation alone made this kernel's dispatch ~6x slower)
    if (live && !rerun) onxt[b] = ocur[self];
    if (!__any(rerun)) return;
    if constexpr (NAT && CH::NATOUT) {
        walk_nat_direct<CH>(p, c, b, bi.y, rerun, b == blast, bi.x);
    } else if constexpr (NAT && CH::NATOUT_FWD) {
        if constexpr (GainNatOf<CH>::value) walk_nat_gain_direct<CH>(p, c, b, bi.y, rerun, bfirst, bi.x);
        else walk_nat_fwd_direct<CH>(p, c, b, bi.y, rerun, bfirst, bi.x);
    } else {
        walk_block<CH, true>(p, c, b, bi.y, rerun, bfirst, 0, p.B);
    }
    if (rerun) {
        onxt[b] = c;
        atomicAdd(p.rerunCount, 1u);
        atomicAdd(p.rerunCountPass, 1u);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fully parallel epilogues (blocked layout).  One lane per (block, quarter) walks its steps; rows stay coalesced.
// ---------------------------------------------------------------------------------------------------------------

// NIS / NLL per bin (pyx:458-475) from the stored predicted covariance, the previous filtered state and the bin
// statistics; per-block partial sums give deterministic sumD / sumNLL.
template <bool NATD>
__global__ __launch_bounds__(256) void k_fwd_dstat(Prm p) {
    __shared__ double redD[4][64], redN[4][64];
    // NATD: D leaves the kernel in the reference layout -- the workgroup's 64 blocks x B steps are staged in LDS
    // ([lane][B + 1] floats, dynamic) and written as one contiguous run of B floats per block; no tD, no export pass
    extern __shared__ float dTile[];
    __shared__ int dBase[64], dLen[64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * 64 + lane;
    const bool live = b < p.NB && chain_on(p, b);
    double sumD = 0.0, sumN = 0.0;
    if (NATD && part == 0) { dBase[lane] = 0; dLen[lane] = 0; }
    if (live) {
        const int4 bi = p.blk[b];
        if (NATD && part == 0) { dBase[lane] = bi.x; dLen[lane] = bi.y; }
        const int64_t base = tbase(b, p.B);
        const int q4 = p.B >> 2;
        const int sBeg = part * q4;
        const int sEnd = (sBeg + q4 < bi.y) ? sBeg + q4 : bi.y;
        const double mD = (double)p.m;
        const double log2pi = 1.8378770664093454835606594728112;
        const bool wantNLL = (p.flags & F_NLL) != 0;
        // DU steps at a time: their loads are issued together (the stores to tD may alias the inputs as far as the
        // compiler knows, so a step-by-step loop keeps only one step of loads in flight per wave)
        constexpr int DU = 4;
        for (int s0 = sBeg; s0 < sEnd; s0 += DU) {
            double lamv[DU], xp0v[DU], ppv[DU], s0u[DU], zb[DU], s2c[DU], slr[DU];
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                const int s = s0 + u;
                lamv[u] = 1.0; xp0v[u] = 0.0; ppv[u] = 0.0; s0u[u] = 0.0; zb[u] = 0.0; s2c[u] = 0.0; slr[u] = 0.0;
                if (s >= sEnd) continue;
                const int64_t i = base + (int64_t)s * 64;
                if (p.flags & F_LAMBDA) lamv[u] = clampd((double)p.tLam[i], p.wMin, p.wMax);
                if (p.d == 2) {
                    double x0, x1;
                    if (s > 0) { const float2 v = p.tXf[i - 64]; x0 = v.x; x1 = v.y; }
                    else if (b > bi.z) { const float2 v = p.tXf[tidx(b - 1, p.B - 1, p.B)]; x0 = v.x; x1 = v.y; }
                    else { x0 = (double)(float)p.init; x1 = 0.0; }
                    xp0v[u] = r32(fma(p.F01, x1, p.F00 * x0));
                    ppv[u] = p.predCompact ? (double)p.tPP[i] : (double)p.tXin[i].z;
                } else {
                    if (s > 0) xp0v[u] = p.tXd[i - 64];
                    else if (b > bi.z) xp0v[u] = p.tXd[tidx(b - 1, p.B - 1, p.B)];
                    else xp0v[u] = p.init;
                    { const float4 r = p.tXin[i]; ppv[u] = unpack_d(r.z, r.w); }
                }
                { const double2 sz = p.tSZ[i]; s0u[u] = sz.x; zb[u] = sz.y; }
                s2c[u] = load_s2c(p, i);
                if (wantNLL) slr[u] = load_logr(p, i);
            }
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                const int s = s0 + u;
                if (s >= sEnd) continue;
                const int64_t i = base + (int64_t)s * 64;
                const double lam = lamv[u], pp = ppv[u];
                const double S0 = lam * s0u[u];
                const double dz = zb[u] - xp0v[u];
                const double S1 = S0 * dz;
                const double S2 = fma(S0, dz * dz, lam * s2c[u]);
                const double is = 1.0 + pp * S0;
                const double gl = pp * rcp_nr(is);
                double quad = S2 - gl * (S1 * S1);
                if (quad < 0.0) quad = 0.0;
                double nll = 0.0;
                if (wantNLL) {
                    double SL = slr[u];
                    if (p.flags & F_LAMBDA) SL -= mD * log_pos(lam);
                    nll = 0.5 * (SL + log_pos(is) + quad + mD * log2pi);
                    sumN += nll;
                }
                const float D = (float)((wantNLL && (p.flags & F_NLL_IN_D)) ? nll : quad / mD);
                if constexpr (NATD) dTile[lane * (p.B + 1) + s] = D;
                else p.tD[i] = D;
                sumD += (double)D;
            }
        }
    }
    redD[part][lane] = sumD;
    redN[part][lane] = sumN;
    __syncthreads();
    if (part == 0 && live) {
        p.blkSumD[b] = ((redD[0][lane] + redD[1][lane]) + redD[2][lane]) + redD[3][lane];
        p.blkSumNLL[b] = ((redN[0][lane] + redN[1][lane]) + redN[2][lane]) + redN[3][lane];
    }
    if constexpr (NATD) {
        const int total = 64 * p.B;
        for (int e = threadIdx.x; e < total; e += 256) {
            const int L = e / p.B, st = e - L * p.B;
            if (st < dLen[L]) p.natD[(int64_t)dBase[L] + st] = dTile[L * (p.B + 1) + st];
        }
    }
}

// per-chain sums in block order (fixed partition and fixed-shape tree: deterministic).  The partition is that of 1024 threads
// (16 wavefronts); a workgroup of NT < 1024 threads plays them in turn -- thread t is the virtual threads t, t + NT, ... (same
// lane, wavefront t / 64 + v NT / 64) -- so the sums have the same bits whichever kernel computes them (round 6 tried them inside
// the residual kernel, profiles/r06_step_close_ab.txt).  sd, sn: 16 doubles of LDS each.
// Four independent accumulators per thread keep the loads of a long chain (10^4 blocks) in flight together.
template <int NT>
__device__ __forceinline__ void chain_sums_dev(const Prm &p, int c, const int64_t *chainFirstBlock, const int64_t *chainNumBlocks,
                                               double *sd, double *sn) {
    static_assert(NT % 64 == 0 && 1024 % NT == 0, "whole wavefronts, a divisor of 1024");
    constexpr int V = 1024 / NT;
    const int64_t b0 = chainFirstBlock[c], nb = chainNumBlocks[c];
    const bool on = p.chainActive == nullptr || p.chainActive[c];
#pragma unroll 1
    for (int v = 0; v < V; ++v) {
        const int vt = (int)threadIdx.x + v * NT;
        // (sixteen independent loads per array and thread in flight: a chain's few 10^4 partial sums are one or two trips to
        // memory, not five dependent ones -- the kernel is pure latency)
        double aD[4] = {0.0, 0.0, 0.0, 0.0}, aN[4] = {0.0, 0.0, 0.0, 0.0};
        if (on) {
            for (int64_t i = vt; i < nb; i += 16384) {
                double vD[16], vN[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int64_t j = i + (int64_t)u * 1024;
                    vD[u] = j < nb ? p.blkSumD[b0 + j] : 0.0;
                    vN[u] = j < nb ? p.blkSumNLL[b0 + j] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    aD[u & 3] += vD[u];
                    aN[u & 3] += vN[u];
                }
            }
        }
        // fixed-order reduction: inside a wavefront by shuffles, the sixteen wavefront sums through LDS (one barrier)
        double d = (aD[0] + aD[1]) + (aD[2] + aD[3]), n = (aN[0] + aN[1]) + (aN[2] + aN[3]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            d += __shfl_down(d, off);
            n += __shfl_down(n, off);
        }
        if ((threadIdx.x & 63) == 0) {
            sd[vt >> 6] = d;
            sn[vt >> 6] = n;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && on) {
        double td = 0.0, tn = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            td += sd[w];
            tn += sn[w];
        }
        p.chainSumD[c] = td;
        p.chainSumNLL[c] = tn;
    }
    __syncthreads();        // (sd / sn may be reused for the next chain)
}
__global__ __launch_bounds__(1024) void k_chain_sums(Prm p, const int64_t *chainFirstBlock, const int64_t *chainNumBlocks) {
    __shared__ double sd[16], sn[16];
    chain_sums_dev<1024>(p, blockIdx.x, chainFirstBlock, chainNumBlocks, sd, sn);
}

// neighbour k+1 of (b, s) in the blocked layout, or -1 at the chain end
__device__ __forceinline__ int64_t next_slot(const Prm &p, int64_t b, int s, const int4 &bi) {
    if (s + 1 < bi.y) return tidx(b, s + 1, p.B);
    if (b < bi.w) return tidx(b + 1, 0, p.B);
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------
// ECM E-steps (blocked layout, elementwise)
// ---------------------------------------------------------------------------------------------------------------
// lambda (pyx:8210-8239): u2 = sum_j ((z_j - xs0)^2 + P00)/R_j = S2c + S0u (zbar - xs0)^2 + P00 S0u
__global__ __launch_bounds__(256) void k_estep_lambda(Prm p) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = (int)(slot & 63);
    const int64_t row = slot >> 6;
    const int64_t G = row / p.B;
    const int s = (int)(row % p.B);
    const int64_t b = G * 64 + l;
    if (b >= p.NB || !chain_on(p, b)) return;
    const int4 bi = p.blk[b];
    if (s >= bi.y) return;
    double p00 = (double)p.tPs[slot].x;
    if (p00 < 0.0) p00 = 0.0;
    const double2 sz = p.tSZ[slot];
    const double s0u = sz.x;
    const double dz = sz.y - (double)p.tXs[slot].x;
    const double u2 = fma(p00, s0u, fma(s0u, dz * dz, load_s2c(p, slot)));
    double w = (p.nu + (double)p.m) / (p.nu + u2);
    if (w < p.wMin) w = p.wMin;
    else if (w > p.wMax) w = p.wMax;
    p.tLam[slot] = (float)w;
}

// kappa (pyx:8244-8298 with MAT2 helpers pyx:4123-4175; level pyx:7496-7521); also produces lag (needed anyway)
__global__ __launch_bounds__(256) void k_estep_kappa(Prm p_) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = (int)(slot & 63);
    const int64_t row = slot >> 6;
    const int64_t G = row / p_.B;
    const int s = (int)(row % p_.B);
    const int64_t b = G * 64 + l;
    if (b >= p_.NB || !chain_on(p_, b)) return;
    const Prm p = lane_model(p_, b);
    const int4 bi = p.blk[b];
    if (s >= bi.y) return;
    if (s == 0 && b == bi.z) p.tKap[slot] = 1.0f;     // processPrecExp[0] = 1 (pyx:8245)
    const int64_t nx = next_slot(p, b, s, bi);
    if (nx < 0) return;
    const float4 lg = p.tLag[slot];
    if (p.d == 2) {
        p.tKap[nx] = estep_kappa_trend(p, p.tXs[slot], p.tPs[slot], p.tXs[nx], p.tPs[nx], lg,
                                       (p.flags & F_QSCALE) ? p.tQs[nx] : 1.0f);
        return;
    }
    const double x0 = p.tXs[slot].x, y0 = p.tXs[nx].x;
    const double pk = p.tPs[slot].x, pk1 = p.tPs[nx].x, ck = lg.x;
    double delta = ((pk1 + y0 * y0) - (2.0 * (ck + x0 * y0)) + (pk + x0 * x0)) * (1.0 / p.Q00);
    if (p.flags & F_QSCALE) delta = delta / (double)p.tQs[nx];
    if (delta < 0.0) delta = 0.0;
    double kap = (p.nu + (double)p.d) / (p.nu + delta);
    if (kap < p.kMin) kap = p.kMin;
    else if (kap > p.kMax) kap = p.kMax;
    p.tKap[nx] = (float)kap;
}

// ---------------------------------------------------------------------------------------------------------------
// sequential fallback with adaptive process noise (pyx:510-527 / 688-705): D[k] feeds back into Q[k+1], so the
// chain cannot be cut; one lane per chain runs the fused step.  Off by default in the reference (constants.py:272).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_fwd_apn(Prm p_, const int64_t *chainFirstBlock, const int64_t *chainNumBlocks) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= p_.nchains) return;
    if (p_.chainActive != nullptr && !p_.chainActive[c]) return;
    const Prm p = chain_model(p_, c);
    const int64_t b0 = chainFirstBlock[c], nb = chainNumBlocks[c];
    const double mD = (double)p.m;
    const double log2pi = 1.8378770664093454835606594728112;
    const bool wantNLL = (p.flags & F_NLL) != 0;
    double apn = 1.0;
    double sumD = 0.0, sumN = 0.0;
    // trend carries
    double x0 = (double)(float)p.init, x1 = 0.0;
    double c00 = (double)(float)p.cinit, c01 = 0.0, c11 = (double)(float)p.cinit;
    // level carries
    double xl = p.init, pl = p.cinit;
    for (int64_t b = b0; b < b0 + nb; ++b) {
        const int4 bi = p.blk[b];
        for (int s = 0; s < bi.y; ++s) {
            const int64_t i = tidx(b, s, p.B);
            const double kap = (p.flags & F_KAPPA) ? clampd((double)p.tKap[i], p.kMin, p.kMax) : 1.0;
            const double lam = (p.flags & F_LAMBDA) ? clampd((double)p.tLam[i], p.wMin, p.wMax) : 1.0;
            const double qs = (p.flags & F_QSCALE) ? (double)p.tQs[i] : apn;
            const double qf = qs / kap;
            const double2 sz = p.tSZ[i];
            const double S0 = lam * sz.x;
            double pp, xp0, quad, is, Qd;
            if (p.d == 2) {
                const double Q00 = qf * p.Q00, Q01 = qf * p.Q01, Q10 = qf * p.Q10, Q11 = qf * p.Q11;
                Qd = 0.5 * (Q00 + Q11);
                const double xq0 = r32(fma(p.F01, x1, p.F00 * x0));
                const double xq1 = r32(fma(p.F11, x1, p.F10 * x0));
                const double t00 = fma(p.F01, c01, p.F00 * c00), t01 = fma(p.F01, c11, p.F00 * c01);
                const double t10 = fma(p.F11, c01, p.F10 * c00), t11 = fma(p.F11, c11, p.F10 * c01);
                const double a00 = r32(fma(t01, p.F01, fma(t00, p.F00, Q00)));
                const double a01 = r32(fma(t01, p.F11, fma(t00, p.F10, Q01)));
                const double a10 = r32(fma(t11, p.F01, fma(t10, p.F00, Q10)));
                const double a11 = r32(fma(t11, p.F11, fma(t10, p.F10, Q11)));
                pp = a00;
                xp0 = xq0;
                is = 1.0 + a00 * S0;
                const double dz = sz.y - xq0;
                const double S1 = S0 * dz;
                const double S2 = fma(S0, dz * dz, lam * load_s2c(p, i));
                quad = S2 - (a00 / is) * (S1 * S1);
                const double delta = S1 / is;
                x0 = r32(xq0 + a00 * delta);
                x1 = r32(xq1 + a10 * delta);
                const double gG = S0 / is, gH = S0 / (is * is);
                const double i00 = 1.0 - a00 * gG, i10 = -(a10 * gG);
                c00 = r32(i00 * i00 * a00 + gH * (a00 * a00));
                c01 = r32(i00 * (i10 * a00 + a01) + gH * (a00 * a10));
                c11 = r32((i10 * i10 * a00 + 2.0 * i10 * a10 + a11) + gH * (a10 * a10));
                p.tXin[i] = pack_gain_trend(S0 / is, (float)a00, (float)a10);
                p.tXf[i] = make_float2((float)x0, (float)x1);
                p.tPf[i] = make_float4((float)c00, (float)c01, (float)c01, (float)c11);
                const float4 qv = make_float4((float)Q00, (float)Q01, (float)Q10, (float)Q11);
                if (s > 0) p.tQ[i - 64] = qv;
                else if (b > b0) p.tQ[tidx(b - 1, p.B - 1, p.B)] = qv;
            } else {
                const double Q = qf * p.Q00;
                Qd = apn * p.Q00;
                pl += Q;
                pp = pl;
                xp0 = xl;
                is = 1.0 + pl * S0;
                const double dz = sz.y - xl;
                const double S1 = S0 * dz;
                const double S2 = fma(S0, dz * dz, lam * load_s2c(p, i));
                quad = S2 - (pl / is) * (S1 * S1);
                xl += pl * (S1 / is);
                const double gG = S0 / is, gH = S0 / (is * is);
                const double ikh = 1.0 - pl * gG;
                pl = ikh * ikh * pl + gH * (pl * pl);
                p.tXin[i] = pack_gain_level(S0 / is, pp);
                p.tXf[i] = make_float2((float)xl, 0.f);
                p.tXd[i] = xl;
                p.tPf[i] = make_float4((float)pl, 0.f, 0.f, 0.f);
                const float4 qv = make_float4((float)Q, 0.f, 0.f, 0.f);
                if (s > 0) p.tQ[i - 64] = qv;
                else if (b > b0) p.tQ[tidx(b - 1, p.B - 1, p.B)] = qv;
            }
            (void)pp; (void)xp0;
            if (quad < 0.0) quad = 0.0;
            double nll = 0.0;
            if (wantNLL) {
                double SL = load_logr(p, i);
                if (p.flags & F_LAMBDA) SL -= mD * log(lam);
                nll = 0.5 * (SL + log(is) + quad + mD * log2pi);
                sumN += nll;
            }
            const float D = (float)((wantNLL && (p.flags & F_NLL_IN_D)) ? nll : quad / mD);
            p.tD[i] = D;
            sumD += (double)D;
            if (!(p.flags & F_QSCALE)) {      // APN feedback
                const double qdiag = (p.d == 2) ? p.qDiag : p.Q00;
                if ((double)D > p.apnThresh && Qd < p.apnMaxQ)
                    apn *= sqrt(p.apnScale * ((double)D - p.apnThresh) + p.apnPC);
                else if ((double)D <= p.apnThresh && Qd > p.apnMinQ)
                    apn *= 1.0 / sqrt(p.apnScale * (p.apnThresh - (double)D) + p.apnPC);
                const double after = apn * qdiag;
                if (after < p.apnMinQ) apn = p.apnMinQ / qdiag;
                else if (after > p.apnMaxQ) apn = p.apnMaxQ / qdiag;
            }
        }
    }
    p.chainSumD[c] = sumD;
    p.chainSumNLL[c] = sumN;
}

// ---------------------------------------------------------------------------------------------------------------
// layout conversion at the API boundary
// ---------------------------------------------------------------------------------------------------------------
// natural float32 (nat[(g + shift)*ncomp + comp]) -> blocked float array element `comp4` of a float/float2/float4 slot
__global__ __launch_bounds__(256) void k_import_f32(Prm p, const float *nat, int ncomp, int comp, float *dst, int dstStride,
                                                   int dstComp) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = (int)(slot & 63);
    const int64_t row = slot >> 6;
    const int64_t G = row / p.B;
    const int s = (int)(row % p.B);
    const int64_t b = G * 64 + l;
    if (b >= p.NB || !chain_on(p, b)) return;
    const int4 bi = p.blk[b];
    if (s >= bi.y) return;
    const int64_t g = (int64_t)bi.x + s;
    dst[slot * dstStride + dstComp] = nat[g * ncomp + comp];
}

// blocked -> blocked copy of a per-bin float array, active chains only (ECM: the kappa of a validated iteration becomes
// the resident one; chains that are masked out keep theirs)
__global__ __launch_bounds__(256) void k_copy_active_f32(Prm p, const float *src, float *dst) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = (int)(slot & 63);
    const int64_t G = (slot >> 6) / p.B;
    const int64_t b = G * 64 + l;
    if (b >= p.NB || !chain_on(p, b)) return;
    dst[slot] = src[slot];
}

// blocked -> natural for a list of arrays in ONE launch: a workgroup owns a (wave-group, 32-step) tile, reads 32
// coalesced rows of every array into LDS and writes, for each of its 64 blocks, 32 consecutive bins (contiguous
// 128-512 bytes) of the reference layout.  E = floats per blocked slot, n <= E = leading components exported.
struct ExpDesc {
    const float *src;       // blocked source, or nullptr: fill with the constant row cval[0..n)
    float *dst;
    int E, n, skipLast, pad_;
    float cval[4];
};
struct ExpList {
    int count;
    int pad_;
    ExpDesc d[8];
};
template <int E, int N>
__device__ __forceinline__ void export_tile(const Prm &p, const ExpDesc &d, float *tile, int64_t G, int s0, int t) {
    typedef float vecE __attribute__((ext_vector_type(E)));
    typedef float vecN __attribute__((ext_vector_type(N)));
    constexpr int RS = 65 * E;                // padded row stride in floats
    const int lane = t & 63, r0 = t >> 6;
    const int64_t rowBase = (G * (int64_t)p.B + s0) * 64;
    const bool fill = d.src == nullptr;
    if (!fill) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + r0;
            const vecE v = *reinterpret_cast<const vecE *>(d.src + (rowBase + (int64_t)row * 64 + lane) * E);
            *reinterpret_cast<vecE *>(tile + row * RS + lane * E) = v;
        }
    }
    __syncthreads();
    const int r = t >> 5, si = t & 31;
#pragma unroll 1
    for (int pass = 0; pass < 8; ++pass) {
        const int l = pass * 8 + r;
        const int64_t b = G * 64 + l;
        if (b < p.NB && chain_on(p, b)) {
            const int4 bi = p.blk[b];
            const int s = s0 + si;
            if (s < bi.y && !(d.skipLast && b == bi.w && s == bi.y - 1)) {
                const int64_t g = (int64_t)bi.x + s;
                const float *q = tile + si * RS + l * E;
                vecN o;
                // constant fill = the base process noise; with per-chain Q0 it is the chain's
                const double *cq = (fill && p.chainQ != nullptr) ? p.chainQ + 4 * (int64_t)p.blkChain[b] : nullptr;
                if constexpr (N == 1) o = fill ? (cq ? (float)cq[0] : d.cval[0]) : q[0];
                else {
#pragma unroll
                    for (int k = 0; k < N; ++k) o[k] = fill ? (cq ? (float)cq[k] : d.cval[k]) : q[k];
                }
                *reinterpret_cast<vecN *>(d.dst + g * N) = o;
            }
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_export_tiled(Prm p, ExpList L) {
    __shared__ __attribute__((aligned(16))) float tile[32 * 65 * 4];
    const int tilesPerGroup = p.B >> 5;
    const int64_t G = blockIdx.x / tilesPerGroup;
    const int s0 = (int)(blockIdx.x % tilesPerGroup) << 5;
    const int t = threadIdx.x;
    for (int a = 0; a < L.count; ++a) {
        const ExpDesc &d = L.d[a];
        if (d.E == 1) export_tile<1, 1>(p, d, tile, G, s0, t);
        else if (d.E == 2 && d.n == 2) export_tile<2, 2>(p, d, tile, G, s0, t);
        else if (d.E == 2) export_tile<2, 1>(p, d, tile, G, s0, t);
        else if (d.n == 4) export_tile<4, 4>(p, d, tile, G, s0, t);
        else export_tile<4, 1>(p, d, tile, G, s0, t);
    }
}

// constant process-noise track (pNoiseForward with constant Q: every row is Q0): plain streaming fill of the natural
// array, 16-byte stores (N = 4) -- the tiled converter spends 90 us on what is a 230 MB fill
template <int N>
__global__ __launch_bounds__(256) void k_fill_rows(float *dst, int64_t rows, float c0, float c1, float c2, float c3) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < rows; g += stride) {
        if constexpr (N == 4) reinterpret_cast<float4 *>(dst)[g] = make_float4(c0, c1, c2, c3);
        else dst[g] = c0;
    }
}

// natural xs0 -> residuals (pyx:6846-6848): resid[g][j] = float(data[j][g] - xs0[g]); (m, Npad) -> (Npad, m) through LDS
// (Both residual kernels also carry a pending folded validation, Prm::prevKind: the smoother stage's check when the
// residuals are the next launch of the stream -- one thread per block, the grid always has more threads than blocks.)
__device__ __forceinline__ void resid_prologue_check(const Prm &p) {
    if (p.prevKind == CK_NONE) return;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b < p.NB) check_previous_stage(p, b, p.blk[b]);
}
__global__ __launch_bounds__(256) void k_resid(Prm p, const float *xsNat, int xsStride, float *resid, int64_t nBins) {
    extern __shared__ float tileR[];                 // [m][65]
    resid_prologue_check(p);
    const int64_t g0 = (int64_t)blockIdx.x * 64;
    const int t = threadIdx.x;
    const int gl = t & 63, r0 = t >> 6;
    const int64_t g = g0 + gl;
    float x = 0.f;
    if (g < nBins) x = xsNat[g * xsStride];
    for (int j = r0; j < p.m; j += 4) {
        float v = 0.f;
        if (g < nBins) v = (float)((double)(p.data[(int64_t)j * p.Npad + g] - (p.bg ? p.bg[g] : 0.f)) - (double)x);
        tileR[j * 65 + gl] = v;
    }
    __syncthreads();
    const int total = 64 * p.m;
    for (int e = t; e < total; e += 256) {
        const int bin = e / p.m, j = e - bin * p.m;
        if (g0 + bin < nBins) resid[(g0 + bin) * (int64_t)p.m + j] = tileR[j * 65 + bin];
    }
}

// vectorised variant for m % 4 == 0: 16-byte loads of four consecutive bins per sample row, 16-byte stores of four
// consecutive samples of one bin.  One workgroup = K * 64 bins x m samples: K independent 16-byte loads per thread and
// sweep are in flight together (K = 1 leaves the kernel latency-bound: 2 loads per thread at m = 32).
template <int K>
__global__ __launch_bounds__(256) void k_resid_v4(Prm p, const float *xsNat, int xsStride, float *resid, int64_t nBins) {
    extern __shared__ float tileR[];                 // [m][K*64+4]: the row stride keeps float4 rows 16-B aligned
    constexpr int RS = K * 64 + 4;
    resid_prologue_check(p);
    const int64_t g0 = (int64_t)blockIdx.x * (K * 64);
    const int t = threadIdx.x;
    const int q = t & 15, r0 = t >> 4;               // q: group of 4 bins, r0: sample row within a sweep of 16
    float x[K][4];
    float4 bg4[K];
    bool ok[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const int64_t g = g0 + u * 64 + 4 * q;
        ok[u] = g + 3 < nBins;
        bg4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        x[u][0] = x[u][1] = x[u][2] = x[u][3] = 0.f;
        if (ok[u]) {
            x[u][0] = xsNat[(g + 0) * xsStride]; x[u][1] = xsNat[(g + 1) * xsStride];
            x[u][2] = xsNat[(g + 2) * xsStride]; x[u][3] = xsNat[(g + 3) * xsStride];
            if (p.bg) bg4[u] = *reinterpret_cast<const float4 *>(p.bg + g);
        }
    }
    for (int j = r0; j < p.m; j += 16) {
        float4 z[K];
#pragma unroll
        for (int u = 0; u < K; ++u) {
            z[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[u]) z[u] = *reinterpret_cast<const float4 *>(p.data + (int64_t)j * p.Npad + g0 + u * 64 + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < K; ++u) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[u]) {
                v.x = (float)((double)(z[u].x - bg4[u].x) - (double)x[u][0]);
                v.y = (float)((double)(z[u].y - bg4[u].y) - (double)x[u][1]);
                v.z = (float)((double)(z[u].z - bg4[u].z) - (double)x[u][2]);
                v.w = (float)((double)(z[u].w - bg4[u].w) - (double)x[u][3]);
            }
            *reinterpret_cast<float4 *>(tileR + j * RS + u * 64 + 4 * q) = v;
        }
    }
    __syncthreads();
    const int m4 = p.m >> 2;
    const int total4 = K * 64 * m4;
    for (int e = t; e < total4; e += 256) {
        const int bin = e / m4, j = (e - bin * m4) << 2;
        if (g0 + bin < nBins) {
            const float4 o = make_float4(tileR[(j + 0) * RS + bin], tileR[(j + 1) * RS + bin],
                                         tileR[(j + 2) * RS + bin], tileR[(j + 3) * RS + bin]);
            *reinterpret_cast<float4 *>(resid + (g0 + bin) * (int64_t)p.m + j) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// a10  expected transition residual sums (pyx:710-863), float64 natural inputs, two-stage deterministic reduction
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tsums(int d, int64_t n, const double *xs, const double *Ps, const double *lag,
                                              double f00, double f01, double f10, double f11, double *partL,
                                              double *partT) {
    __shared__ double sL[256], sT[256];
    double aL = 0.0, aT = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n - 1; k += (int64_t)gridDim.x * 256) {
        if (d == 1) {
            const double x0 = xs[k], y0 = xs[k + 1];
            const double e0 = Ps[k] + (x0 * x0), e1 = Ps[k + 1] + (y0 * y0), ec = lag[k] + (x0 * y0);
            double mom = e1 - (2.0 * ec) + e0;
            if (mom < 0.0) mom = 0.0;
            aL += mom;
        } else {
            const double x00 = xs[k * 2], x01 = xs[k * 2 + 1], y0 = xs[(k + 1) * 2], y1 = xs[(k + 1) * 2 + 1];
            const double a00 = Ps[k * 4] + (x00 * x00), a01 = Ps[k * 4 + 1] + (x00 * x01);
            const double a10 = Ps[k * 4 + 2] + (x01 * x00), a11 = Ps[k * 4 + 3] + (x01 * x01);
            const double b00 = Ps[(k + 1) * 4] + (y0 * y0), b11 = Ps[(k + 1) * 4 + 3] + (y1 * y1);
            const double c00 = lag[k * 4] + (x00 * y0), c01 = lag[k * 4 + 1] + (x00 * y1);
            const double c10 = lag[k * 4 + 2] + (x01 * y0), c11 = lag[k * 4 + 3] + (x01 * y1);
            double lm = (b00 - (2.0 * ((f00 * c00) + (f01 * c10))) + (f00 * f00 * a00) + (f00 * f01 * a01) +
                         (f01 * f00 * a10) + (f01 * f01 * a11));
            double tm = (b11 - (2.0 * ((f10 * c01) + (f11 * c11))) + (f10 * f10 * a00) + (f10 * f11 * a01) +
                         (f11 * f10 * a10) + (f11 * f11 * a11));
            if (lm < 0.0) lm = 0.0;
            if (tm < 0.0) tm = 0.0;
            aL += lm;
            aT += tm;
        }
    }
    sL[threadIdx.x] = aL;
    sT[threadIdx.x] = aT;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sL[threadIdx.x] += sL[threadIdx.x + w];
            sT[threadIdx.x] += sT[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partL[blockIdx.x] = sL[0];
        partT[blockIdx.x] = sT[0];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// synthetic workload generator (SURVEY 8(d) distributions; counter-based RNG, not NumPy's stream)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float gauss(uint64_t key) {
    const uint64_t a = mix64(key), b = mix64(key ^ 0xD1B54A32D192ED03ull);
    const float u1 = ((float)(a >> 40) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)(b >> 40) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
}
// latent[g] (natural, length Npad) holds the random-walk level uploaded by the host
__global__ __launch_bounds__(256) void k_synth(Prm p, const float *latent, float *data, float *munc, uint64_t seed,
                                              int64_t nBins) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= nBins) return;
    const float x = latent[g];
    for (int j = 0; j < p.m; ++j) {
        const uint64_t key = seed * 0x100000001B3ull + ((uint64_t)j << 40) + (uint64_t)g;
        data[(int64_t)j * p.Npad + g] = x + 0.5f * gauss(key * 2);
        munc[(int64_t)j * p.Npad + g] = 0.25f * expf(0.2f * gauss(key * 2 + 1));
    }
}


// ---------------------------------------------------------------------------------------------------------------
// per-interval output diagnostics (core.py:7734-7878).  Natural layout, one thread per bin: the
// reference's per-bin Python loop carries no state (bin k uses the STORED float32 filtered covariance of bin k-1).
// ---------------------------------------------------------------------------------------------------------------
struct DiagArgs {
    const float *Pf;        // natural (Npad, d*d) filtered covariance
    const float *pn;        // natural (Npad, d*d) process noise, row k-1 entering bin k; nullptr = not used
    const float *lam, *kap, *qs;        // natural (Npad) multipliers or nullptr
    float *g0, *g1, *eql, *eqt, *trace;
    const int64_t *chainOff, *chainLen;
    int nchains, pad_;
};
__global__ __launch_bounds__(256) void k_diag_natural(Prm p_, DiagArgs a) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= p_.Npad) return;
    int lo = 0, hi = a.nchains - 1;          // chain whose [off, off + padded len) holds g
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.chainOff[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const int64_t k = g - a.chainOff[lo];
    if (k >= a.chainLen[lo]) return;
    if (p_.chainActive != nullptr && !p_.chainActive[lo]) return;
    const Prm p = chain_model(p_, lo);
    const double tiny = 2.2250738585072014e-308;
    const int d = p.d, dd = d * d;
    double lam = 1.0;
    if (a.lam) lam = fmax(clampd((double)a.lam[g], p.wMin, p.wMax), tiny);
    // R / lambda and lambda / R through one Newton-refined reciprocal per cell (<= 1 ulp from the IEEE quotient; the
    // two fp64 divisions per cell otherwise make this HBM-streaming kernel compute-bound)
    const double invLam = rcp_nr(lam);
    double tr = 0.0, sInv = 0.0;
#pragma unroll 4
    for (int j = 0; j < p.m; ++j) {
        const double R = fmax((double)p.munc[(int64_t)j * p.Npad + g] + p.pad, 1.0e-12);
        const double e = R * invLam, w = lam * rcp_nr(R);
        if (isfinite(e)) tr += e;
        if (isfinite(w)) sInv += w;
    }
    double qs = 1.0;
    if (a.qs && k > 0) qs = fmax((double)a.qs[g], tiny);
    // only the first column of the predicted covariance (pred00, pred10) is needed: Q01 never enters
    double Q00 = p.Q00 * qs, Q10 = p.Q10 * qs, Q11 = p.Q11 * qs;
    if (a.kap) {
        const double kp = fmax(clampd((double)a.kap[g], p.kMin, p.kMax), tiny);
        Q00 /= kp; Q10 /= kp; Q11 /= kp;
    } else if (a.pn && k > 0) {
        const float *q = a.pn + (g - 1) * dd;
        if (d == 2) {
            const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            if (isfinite(q0) && isfinite(q1) && isfinite(q2) && isfinite(q3)) { Q00 = q0; Q10 = q2; Q11 = q3; }
        } else {
            const double q0 = q[0];
            if (isfinite(q0)) Q00 = q0;
        }
    }
    double c00 = p.cinit, c01 = 0.0, c10 = 0.0, c11 = p.cinit;
    if (k > 0) {
        const float *c = a.Pf + (g - 1) * dd;
        c00 = c[0];
        if (d == 2) { c01 = c[1]; c10 = c[2]; c11 = c[3]; }
    }
    double pred00, pred10 = 0.0;
    if (d == 2) {
        // F P F^T + Q in float64 (core.py:7850)
        const double t00 = p.F00 * c00 + p.F01 * c10, t01 = p.F00 * c01 + p.F01 * c11;
        const double t10 = p.F10 * c00 + p.F11 * c10, t11 = p.F10 * c01 + p.F11 * c11;
        pred00 = (t00 * p.F00 + t01 * p.F01) + Q00;
        pred10 = (t10 * p.F00 + t11 * p.F01) + Q10;
    } else {
        pred00 = c00 + Q00;
    }
    pred00 = fmax(pred00, 0.0);
    const double den = 1.0 + pred00 * sInv;
    double g0 = 0.0, g1 = 0.0;
    if (isfinite(den) && den > 0.0) {
        const double sc = sInv / den;
        g0 = pred00 * sc;
        g1 = pred10 * sc;
    }
    a.g0[g] = (float)g0;
    a.g1[g] = (float)g1;
    a.eql[g] = (float)Q00;
    a.eqt[g] = d == 2 ? (float)Q11 : 0.f;
    a.trace[g] = (float)tr;
}

}  // namespace csr
