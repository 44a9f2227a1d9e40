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


// ---------------------------------------------------------------------------------------------------------------
// bigWig (io.py:530-760: the reference converts its bedGraph files with pyBigWig).  The body of a bigWig file is byte
// work with fixed record sizes, produced here straight from the resident track: the data sections (24-byte header + 12-byte
// bedGraph items, Kent et al. 2010, file format supplement) and the reduction records a file needs (total summary, zoom
// levels).  The values are the ones the reference's file holds: it writes `"%.4f" % v` to the bedGraph and parses that text
// back (io.py:707), so an item is float32(R / 10^4) with R = the round-half-even integer of v * 10^4 (same exact integer
// arithmetic as bgw_format_value).  Assembly (chromosome tree, R-tree index, optional zlib of every block, headers) is
// O(sections) host work in consenrich_amd/writers.py.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bgw_text4_value(float v) {      // value of the decimal text "%.4f" % v, as float32
    if (!isfinite(v)) return v;
    const double t = fabs((double)v) * 10000.0;                   // exact
    const double r = rint(t) / 10000.0;                           // correctly rounded quotient = strtod of the text
    return (float)((__float_as_uint(v) >> 31) ? -r : r);
}
struct BwArgs {
    BgwArgs g;                  // track, intervals (fixed step or explicit), transform
    unsigned int chromId;
    int itemsPerSection;
    int64_t binsPerRecord;      // zoom: consecutive rows per reduction record
    unsigned char *out;         // sections / zoom records
    double *part;               // per-workgroup partial summaries: [bases, min, max, sum, sumSq, nonFinite] x gridDim
};
// one thread per item; the first thread of a section also writes its header
__global__ __launch_bounds__(256) void k_bw_sections(BwArgs a) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double bases = 0.0, mn = INFINITY, mx = -INFINITY, sm = 0.0, sq = 0.0, bad = 0.0;
    if (k < a.g.n) {
        int64_t s, e;
        bgw_interval(a.g, k, s, e);
        const float v = bgw_text4_value(bgw_value(a.g, k));
        const int64_t sec = k / a.itemsPerSection, j = k - sec * a.itemsPerSection;
        unsigned int *o = reinterpret_cast<unsigned int *>(a.out + sec * (24 + 12 * (int64_t)a.itemsPerSection) + 24 + 12 * j);
        o[0] = (unsigned int)s; o[1] = (unsigned int)e; o[2] = __float_as_uint(v);
        if (j == 0) {
            const int64_t last = (sec + 1) * a.itemsPerSection < a.g.n ? (sec + 1) * a.itemsPerSection - 1 : a.g.n - 1;
            int64_t s2, e2;
            bgw_interval(a.g, last, s2, e2);
            unsigned int *h = reinterpret_cast<unsigned int *>(a.out + sec * (24 + 12 * (int64_t)a.itemsPerSection));
            h[0] = a.chromId; h[1] = (unsigned int)s; h[2] = (unsigned int)e2; h[3] = 0u; h[4] = 0u;
            h[5] = 1u | ((unsigned int)(last - k + 1) << 16);      // type 1 (bedGraph), reserved 0, itemCount (u16)
        }
        if (isfinite(v)) {
            const double w = (double)(e - s), dv = (double)v;
            bases = w; mn = dv; mx = dv; sm = dv * w; sq = dv * dv * w;
        } else bad = 1.0;
    }
    __shared__ double sh[6][256];
    sh[0][threadIdx.x] = bases; sh[1][threadIdx.x] = mn; sh[2][threadIdx.x] = mx;
    sh[3][threadIdx.x] = sm; sh[4][threadIdx.x] = sq; sh[5][threadIdx.x] = bad;
    __syncthreads();
    for (int wd = 128; wd > 0; wd >>= 1) {
        if ((int)threadIdx.x < wd) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + wd];
            sh[1][threadIdx.x] = fmin(sh[1][threadIdx.x], sh[1][threadIdx.x + wd]);
            sh[2][threadIdx.x] = fmax(sh[2][threadIdx.x], sh[2][threadIdx.x + wd]);
            sh[3][threadIdx.x] += sh[3][threadIdx.x + wd];
            sh[4][threadIdx.x] += sh[4][threadIdx.x + wd];
            sh[5][threadIdx.x] += sh[5][threadIdx.x + wd];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int q = 0; q < 6; ++q) a.part[(int64_t)blockIdx.x * 6 + q] = sh[q][0];
}
// zoom level: one thread per reduction record of binsPerRecord consecutive rows -> (chromId, start, end, validCount, min,
// max, sum, sumSquares), 32 bytes (bbiSummaryOnDisk); sums over BASES in double, stored as float32
__global__ __launch_bounds__(256) void k_bw_zoom(BwArgs a, int64_t nrec) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    const int64_t k0 = r * a.binsPerRecord, k1 = (k0 + a.binsPerRecord < a.g.n) ? k0 + a.binsPerRecord : a.g.n;
    double mn = INFINITY, mx = -INFINITY, sm = 0.0, sq = 0.0;
    int64_t valid = 0, s0 = 0, e1 = 0;
    for (int64_t k = k0; k < k1; ++k) {
        int64_t s, e;
        bgw_interval(a.g, k, s, e);
        if (k == k0) s0 = s;
        e1 = e;
        const double dv = (double)bgw_text4_value(bgw_value(a.g, k)), w = (double)(e - s);
        valid += e - s;
        mn = fmin(mn, dv); mx = fmax(mx, dv); sm += dv * w; sq += dv * dv * w;
    }
    unsigned int *o = reinterpret_cast<unsigned int *>(a.out + r * 32);
    o[0] = a.chromId; o[1] = (unsigned int)s0; o[2] = (unsigned int)e1; o[3] = (unsigned int)valid;
    o[4] = __float_as_uint((float)mn); o[5] = __float_as_uint((float)mx);
    o[6] = __float_as_uint((float)sm); o[7] = __float_as_uint((float)sq);
}

}  // namespace csr
