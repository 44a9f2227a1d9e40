// csr_folds.h -- SURVEY 8(f) rank 2b: natives of the delete-block uncertainty calibration
// (reference: src/consenrich/cuncertainty.pyx:97-157 `cobservationTotalInformation`, :160-305
// `cmakeFoldMaskAndInformation`; caller uncertainty.py:1370-1419 runs one full fit per fold with the masked cells'
// variance set to 1e30, constants.py:387).  One thread per bin: the m-loop runs in the reference's own order (ascending
// rows for totals / kept information, the fold spec's replicate order for the held-out sums), fp64 with IEEE division
// and correctly rounded sqrt, so the tracks equal the reference's bit for bit.  `csr_batch_make_fold` additionally writes
// the fold's masked variance matrix (and a copy of the data) straight into another chain of the batch: the folds of a
// chromosome become extra chains of the same device-resident fit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csr {

struct FoldArgs {
    int64_t m, n, blockLen, fold, slots;
    int64_t stride;                 // elements between consecutive rows of munc / active / mask (n, or Npad in a batch)
    const void *munc;
    int muncF64, useLambda, wantNominal, hasActive;
    const uint8_t *active;          // (m, stride) or null = every cell active
    const int32_t *blockFold;
    const int64_t *repsCount, *reps;
    const double *lambda, *totalIn;
    double pad, rho;
    uint8_t *mask;                  // (m, stride) or null
    double *total, *kept, *heldout, *h, *nominal;
    // batch fold creation: masked copy of the variances (and plain copy of the data) into another chain
    const float *srcData;
    float *dstData, *dstMunc;
    float maskedVariance;
};

__device__ __forceinline__ double fold_exchangeable(double sumW, double sumSqrt, int64_t count, double rho) {   // unc:37-57
    if (count <= 0 || sumW <= 0.0) return 0.0;
    if (rho <= 0.0) return sumW;
    const double omr = 1.0 - rho;
    const double denom = omr + rho * (double)count;
    const double adjusted = sumW / omr - rho * sumSqrt * sumSqrt / (omr * denom);
    return adjusted > sumW ? sumW : adjusted;
}
__device__ __forceinline__ double fold_munc(const FoldArgs &a, int64_t idx) {
    return a.muncF64 ? ((const double *)a.munc)[idx] : (double)((const float *)a.munc)[idx];
}
__device__ __forceinline__ bool fold_active(const FoldArgs &a, int64_t idx) { return !a.hasActive || a.active[idx] != 0; }

// total information per bin (unc:131-156)
__global__ __launch_bounds__(256) void k_fold_total(FoldArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    double tot = 0.0, ssq = 0.0;
    int64_t count = 0;
    const double lam = a.useLambda ? a.lambda[i] : 1.0;
    for (int64_t j = 0; j < a.m; ++j) {
        const int64_t idx = j * a.stride + i;
        if (fold_active(a, idx)) {
            const double v = lam / (fold_munc(a, idx) + a.pad);
            tot += v;
            if (a.rho > 0.0) { ssq += __dsqrt_rn(v); ++count; }
        }
    }
    a.total[i] = a.rho > 0.0 ? fold_exchangeable(tot, ssq, count, a.rho) : tot;
}

// mask + kept / held-out information of one fold (unc:253-302); optionally the fold's chain of a batch
__global__ __launch_bounds__(256) void k_fold_mask(FoldArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int64_t blk = i / a.blockLen;
    const bool inFold = (int64_t)a.blockFold[blk] == a.fold;
    const int64_t cnt = inFold ? a.repsCount[blk] : 0;
    const int64_t *reps = a.reps + blk * a.slots;
    double held = 0.0, nominal = 0.0;
    // deleted replicates in the fold spec's order (that order is the reference's summation order)
    for (int64_t hh = 0; hh < cnt; ++hh) {
        const int64_t idx = reps[hh] * a.stride + i;
        if (fold_active(a, idx)) {
            double v = 1.0 / (fold_munc(a, idx) + a.pad);
            if (a.useLambda) v *= a.lambda[i];
            if (a.rho <= 0.0) held += v;
            nominal += v;
        }
    }
    double kp = 0.0, ssq = 0.0;
    int64_t count = 0;
    const double lam = a.useLambda ? a.lambda[i] : 1.0;
    for (int64_t j = 0; j < a.m; ++j) {
        bool deleted = false;
        for (int64_t hh = 0; hh < cnt; ++hh) deleted |= reps[hh] == j;
        const int64_t idx = j * a.stride + i;
        if (a.mask) a.mask[idx] = deleted ? 0 : 1;
        if (a.dstMunc) {
            a.dstMunc[idx] = deleted ? a.maskedVariance : ((const float *)a.munc)[idx];
            a.dstData[idx] = a.srcData[idx];
        }
        if (a.rho > 0.0 && !deleted && fold_active(a, idx)) {
            const double v = lam / (fold_munc(a, idx) + a.pad);
            kp += v;
            ssq += __dsqrt_rn(v);
            ++count;
        }
    }
    const double tot = a.totalIn[i];
    if (a.rho > 0.0) {
        kp = fold_exchangeable(kp, ssq, count, a.rho);
        held = tot - kp;
    } else {
        kp = tot - held;
    }
    a.kept[i] = kp;
    a.heldout[i] = held;
    a.h[i] = tot > 0.0 ? held / tot : __longlong_as_double(0x7ff8000000000000LL);
    if (a.wantNominal) a.nominal[i] = nominal;
}

}  // namespace csr
