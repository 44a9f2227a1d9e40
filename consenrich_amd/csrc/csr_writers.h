// csr_writers.h -- SURVEY 8(f) rank 3: bedGraph text of a track, byte for byte what the reference's writer emits
// (consenrich.py:9797-9805: pandas `to_csv(sep="\t", header=False, index=False, float_format="%.4f",
// lineterminator="\n")` on [Chromosome, Start, End, value]; optional value transforms of the caller:
// `getPrimaryState` core.py:6145-6166 = np.round(x, 4) in float32, uncertainty = sqrt(P00) consenrich.py:9476-9477).
//
// Three passes over the rows: byte length of every row, exclusive scan (row offsets), formatting.  "%.4f" of a float32
// is exact integer work: v * 10^4 is EXACT in double (24-bit x 14-bit mantissas), rint() gives the round-half-even
// integer R, the text is R / 10^4 "." R % 10^4.  |v| >= 2^63 / 10^4 takes a 192-bit path (float32 reaches 3.4e38 and
// the reference prints all 39 integer digits).  NaN prints as the empty string, infinities as inf / -inf (pandas).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csr {

enum { BGW_NONE = 0, BGW_ROUND4 = 1, BGW_SQRT = 2 };

struct BgwArgs {
    int64_t n;
    const float *values;        // value of row k at values[k * stride + comp]
    int stride, comp, transform, chromLen;
    const int64_t *starts, *ends;   // per-row intervals, or null: start = start0 + k*step, end = min(start + step, endCap)
    int64_t start0, step, endCap;
    char chrom[64];
    int *rowLen;                // n
    int64_t *rowOff;            // n (exclusive scan of rowLen)
    int64_t *blockSum;          // ceil(n / 1024) + 1
    char *out;
};

__device__ __forceinline__ float bgw_value(const BgwArgs &a, int64_t k) {
    float v = a.values[k * a.stride + a.comp];
    if (a.transform == BGW_ROUND4) v = __fdiv_rn(rintf(v * 10000.0f), 10000.0f);       // np.round(x, 4), float32
    else if (a.transform == BGW_SQRT) v = (float)__dsqrt_rn((double)v);     // correctly rounded float32 sqrt (53 >= 2*24+2 bits)
    return v;
}
__device__ __forceinline__ int bgw_udigits(unsigned long long x) {
    int d = 1;
    while (x >= 10ull) { x /= 10ull; ++d; }
    return d;
}
// 192-bit magnitude of |v| * 10^4 for huge |v| (6 x 32-bit limbs, little endian)
__device__ __forceinline__ void bgw_big(float v, unsigned int limb[6]) {
    const unsigned int bits = __float_as_uint(v) & 0x7fffffffu;
    const int e = (int)(bits >> 23) - 150;                      // |v| = m * 2^e, m has 24 bits (normal numbers here)
    unsigned long long M = (unsigned long long)((bits & 0x7fffffu) | 0x800000u) * 625ull;      // |v| * 10^4 = M * 2^(e+4)
    const int sh = e + 4;                                        // > 0 on this path
#pragma unroll
    for (int i = 0; i < 6; ++i) limb[i] = 0u;
    const int w = sh >> 5, b = sh & 31;
    // M < 2^34: spread over up to three limbs after the shift
    const unsigned long long lo = M << b;                        // M << b fits in 64 bits only if b <= 30; handle carry
    const unsigned long long hi = b ? (M >> (64 - b)) : 0ull;
    if (w < 6) limb[w] = (unsigned int)lo;
    if (w + 1 < 6) limb[w + 1] = (unsigned int)(lo >> 32);
    if (w + 2 < 6) limb[w + 2] = (unsigned int)hi;
}
// decimal digits of a 192-bit number into buf (most significant first); returns the count
__device__ __forceinline__ int bgw_big_digits(unsigned int limb[6], char *buf) {
    char tmp[64];
    int nd = 0;
    bool nz = true;
    while (nz) {
        unsigned long long rem = 0ull;
        nz = false;
        for (int i = 5; i >= 0; --i) {
            const unsigned long long cur = (rem << 32) | limb[i];
            limb[i] = (unsigned int)(cur / 1000000000ull);
            rem = cur % 1000000000ull;
            nz |= limb[i] != 0u;
        }
        for (int k = 0; k < 9; ++k) {
            tmp[nd++] = (char)('0' + (int)(rem % 10ull));
            rem /= 10ull;
            if (!nz && rem == 0ull) break;
        }
    }
    for (int k = 0; k < nd; ++k) buf[k] = tmp[nd - 1 - k];
    return nd;
}
// text of one value; returns its length.  buf must hold 48 bytes.
__device__ __forceinline__ int bgw_format_value(float v, char *buf) {
    if (isnan(v)) return 0;
    int n = 0;
    if (__float_as_uint(v) >> 31) buf[n++] = '-';
    if (isinf(v)) { buf[n++] = 'i'; buf[n++] = 'n'; buf[n++] = 'f'; return n; }
    const double t = fabs((double)v) * 10000.0;          // exact
    if (t < 9.0e18) {
        const unsigned long long R = (unsigned long long)rint(t);       // round half to even on the exact product
        const unsigned long long ip = R / 10000ull;
        unsigned int fp = (unsigned int)(R % 10000ull);
        const int nd = bgw_udigits(ip);
        unsigned long long x = ip;
        for (int k = nd - 1; k >= 0; --k) { buf[n + k] = (char)('0' + (int)(x % 10ull)); x /= 10ull; }
        n += nd;
        buf[n++] = '.';
        buf[n + 3] = (char)('0' + fp % 10u); fp /= 10u;
        buf[n + 2] = (char)('0' + fp % 10u); fp /= 10u;
        buf[n + 1] = (char)('0' + fp % 10u); fp /= 10u;
        buf[n + 0] = (char)('0' + fp % 10u);
        return n + 4;
    }
    // huge: t is an integer already (|v| >= 9e14 has no fractional bits); all digits of t, the last four are decimals
    unsigned int limb[6];
    bgw_big(v, limb);
    char dg[64];
    const int nd = bgw_big_digits(limb, dg);
    for (int k = 0; k < nd - 4; ++k) buf[n++] = dg[k];
    buf[n++] = '.';
    for (int k = nd - 4; k < nd; ++k) buf[n++] = dg[k];
    return n;
}
__device__ __forceinline__ void bgw_interval(const BgwArgs &a, int64_t k, int64_t &s, int64_t &e) {
    if (a.starts) { s = a.starts[k]; e = a.ends[k]; }
    else {
        s = a.start0 + k * a.step;
        e = s + a.step;
        if (a.endCap > 0 && e > a.endCap) e = a.endCap;
    }
}
__device__ __forceinline__ int bgw_idigits(int64_t x) { return x < 0 ? 1 + bgw_udigits((unsigned long long)(-x)) : bgw_udigits((unsigned long long)x); }
__device__ __forceinline__ int bgw_put_int(int64_t x, char *buf) {
    int n = 0;
    unsigned long long u = (unsigned long long)x;
    if (x < 0) { buf[n++] = '-'; u = (unsigned long long)(-x); }
    const int nd = bgw_udigits(u);
    for (int k = nd - 1; k >= 0; --k) { buf[n + k] = (char)('0' + (int)(u % 10ull)); u /= 10ull; }
    return n + nd;
}

// pass 1: row lengths + per-block (1024 rows) sums
__global__ __launch_bounds__(1024) void k_bgw_len(BgwArgs a) {
    __shared__ int sh[1024];
    const int64_t k = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    int len = 0;
    if (k < a.n) {
        int64_t s, e;
        bgw_interval(a, k, s, e);
        char tmp[48];
        len = a.chromLen + 1 + bgw_idigits(s) + 1 + bgw_idigits(e) + 1 + bgw_format_value(bgw_value(a, k), tmp) + 1;
        a.rowLen[k] = len;
    }
    sh[threadIdx.x] = len;
    __syncthreads();
    for (int wd = 512; wd > 0; wd >>= 1) {
        if ((int)threadIdx.x < wd) sh[threadIdx.x] += sh[threadIdx.x + wd];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.blockSum[blockIdx.x] = sh[0];
}
// pass 2a: exclusive scan of the block sums (one workgroup; nb <= a few 10^4)
__global__ __launch_bounds__(1024) void k_bgw_scan_blocks(BgwArgs a, int64_t nb) {
    __shared__ int64_t sh[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < nb ? a.blockSum[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {            // Hillis-Steele inclusive scan
            const int64_t add = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        const int64_t incl = sh[threadIdx.x] + carry;
        if (i < nb) a.blockSum[i] = incl - v;             // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry = incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.blockSum[nb] = carry;         // total bytes
}
// pass 2b: row offsets inside each block
__global__ __launch_bounds__(1024) void k_bgw_scan_rows(BgwArgs a) {
    __shared__ int sh[1024];
    const int64_t k = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int v = k < a.n ? a.rowLen[k] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int add = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    if (k < a.n) a.rowOff[k] = a.blockSum[blockIdx.x] + (int64_t)(sh[threadIdx.x] - v);
}
// pass 3: the text
__global__ __launch_bounds__(256) void k_bgw_write(BgwArgs a) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.n) return;
    char *o = a.out + a.rowOff[k];
    int n = 0;
    for (int i = 0; i < a.chromLen; ++i) o[n++] = a.chrom[i];
    o[n++] = '\t';
    int64_t s, e;
    bgw_interval(a, k, s, e);
    n += bgw_put_int(s, o + n);
    o[n++] = '\t';
    n += bgw_put_int(e, o + n);
    o[n++] = '\t';
    char tmp[48];
    const int nv = bgw_format_value(bgw_value(a, k), tmp);
    for (int i = 0; i < nv; ++i) o[n++] = tmp[i];
    o[n++] = '\n';
}

}  // namespace csr
