// micro-benchmark (gfx950), round 4: cheaper ROUNDS for the delta-form repair runs of the bit-exact state chain.
// round_lat.hip measured the shipped round at 272 ticks (arithmetic 80, bookkeeping 156).  Variants here keep ONE vector h per
// state component -- lane k holds the true state of its bin once settled, the hypothesis S_k + delta otherwise -- so that
//   * the predecessor of every lane is a wave shift of h (the base lane needs no select: its predecessor is settled),
//   * settled lanes recompute their own bits and pass the comparison by construction (no `pos` mask, no 64-bit shifts),
//   * the bookkeeping is: compare -> s_ff1 -> two v_readlane of e = n - S -> h = lane <= f ? n : S + e_f.
// Synthetic failure pattern as in round_lat.hip: exactly one failing lane, `stride` lanes ahead of the previous one.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__device__ __forceinline__ float rl(float v, int lane) {
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
}
__device__ __forceinline__ float shr1(float keep0, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep0), __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ void step(float s0, float s1, double gs, double zbar, double p00, double p10, float &o0, float &o1) {
    const float xpf = s0 + s1;
    const double xp0 = (double)xpf, x1d = (double)s1;
    const double dl = gs * (zbar - xp0);
    o0 = (float)fma(p00, dl, xp0);
    o1 = (float)fma(p10, dl, x1d);
}
template <int V>
__global__ void k(float *out, long long *cyc, float a, int stride) {
    const int lane = threadIdx.x;
    const double gs = 0.26 + lane * 1e-4, zbar = 3.1, p00 = 0.8, p10 = 0.05;
    const float so0 = a + lane * 1e-3f, so1 = 1e-4f * lane;
    float h0 = so0, h1 = so1;
    const float tc0 = a, tc1 = 0.f;
    int pos = 0;
    float acc = 0.f;
    long long c0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N; ++it) {
        if (V == 5 || V == 7 || V == 8) {
            const float q0 = shr1(tc0, h0), q1 = shr1(tc1, h1);
            float n0, n1;
            step(q0, q1, gs, zbar, p00, p10, n0, n1);
            const float e0 = n0 - so0, e1 = n1 - so1;
            unsigned long long fail = __builtin_amdgcn_uicmp(__float_as_uint(n0), __float_as_uint(h0), 33 /* NE */) |
                                      __builtin_amdgcn_uicmp(__float_as_uint(n1), __float_as_uint(h1), 33);
            const int X = (pos + stride) & 63;
            fail &= 0x5ull << X;                                              // synthetic: the real comparisons of lanes X and X + 2 only
            const int f = fail ? (int)__builtin_ctzll(fail) : X;              // (the kernel leaves the loop on fail == 0: same scalar test)
            if (V == 7) {               // + two shift-register steps behind the failing lane before the re-base
                float w0 = lane <= f ? n0 : h0, w1 = lane <= f ? n1 : h1;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float p0 = shr1(tc0, w0), p1 = shr1(tc1, w1);
                    step(p0, p1, gs, zbar, p00, p10, w0, w1);
                }
                const int g = min(f + 2, 63);
                const float d0 = rl(w0 - so0, g), d1 = rl(w1 - so1, g);
                const bool le = lane <= g;
                h0 = le ? w0 : so0 + d0;
                h1 = le ? w1 : so1 + d1;
                pos = g;
            } else if (V == 8) {        // the re-base through EXEC: readfirstlane of e under the fail mask (no s_ff1, no lane index)
                float d0, d1;
                asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, %2\n\tv_readfirstlane_b32 %0, %3\n\tv_readfirstlane_b32 %1, %4\n\ts_mov_b64 exec, s[20:21]"
                             : "=s"(d0), "=s"(d1) : "s"(fail), "v"(e0), "v"(e1) : "s20", "s21");
                const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(fail >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fail, 0u));
                const bool le = below == 0u;        // no failing lane strictly below: lanes <= f
                h0 = le ? n0 : so0 + d0;
                h1 = le ? n1 : so1 + d1;
                pos = f;
            } else {
                const float d0 = rl(e0, f), d1 = rl(e1, f);
                const bool le = lane <= f;
                h0 = le ? n0 : so0 + d0;
                h1 = le ? n1 : so1 + d1;
                pos = f;
            }
        } else if (V == 6) {            // V5 with ONE 64-bit comparison of the packed pair
            const float q0 = shr1(tc0, h0), q1 = shr1(tc1, h1);
            float n0, n1;
            step(q0, q1, gs, zbar, p00, p10, n0, n1);
            const float e0 = n0 - so0, e1 = n1 - so1;
            const unsigned long long nn = ((unsigned long long)__float_as_uint(n1) << 32) | __float_as_uint(n0);
            const unsigned long long hh = ((unsigned long long)__float_as_uint(h1) << 32) | __float_as_uint(h0);
            unsigned long long fail = __builtin_amdgcn_uicmpl(nn, hh, 33);
            const int X = (pos + stride) & 63;
            fail &= 0x5ull << X;
            const int f = fail ? (int)__builtin_ctzll(fail) : X;
            const float d0 = rl(e0, f), d1 = rl(e1, f);
            const bool le = lane <= f;
            h0 = le ? n0 : so0 + d0;
            h1 = le ? n1 : so1 + d1;
            pos = f;
        } else if (V == 9) {            // lower bound of the scheme: shift + step + hypothesis refresh, no cross-lane decision at all
            const float q0 = shr1(tc0, h0), q1 = shr1(tc1, h1);
            float n0, n1;
            step(q0, q1, gs, zbar, p00, p10, n0, n1);
            h0 = so0 + (n0 - so0); h1 = so1 + (n1 - so1);
        } else if (V == 10) {           // one shift-register step (the walk's cost per bin, for the same constants)
            const float q0 = shr1(tc0, h0), q1 = shr1(tc1, h1);
            step(q0, q1, gs, zbar, p00, p10, h0, h1);
        }
        acc += h0;
    }
    long long c1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc + h0 + h1 + pos;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
template <int V> void run(const char *name, int stride) {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 4 * 64); (void)hipMalloc(&cyc, 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, 3.0f, stride); (void)hipDeviceSynchronize(); }
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-86s %.1f ticks per round\n", name, (double)c / N);
}
int main() {
    run<5>("h-vector round: 2 compares, s_ff1, 2 v_readlane, select", 7);
    run<6>("h-vector round: one 64-bit compare", 7);
    run<7>("h-vector round + 2 shift-register steps behind the failing lane", 7);
    run<8>("h-vector round: re-base by v_readfirstlane under EXEC = fail mask, lanes <= f by v_mbcnt", 7);
    run<9>("lower bound: shift + step + refresh (no cross-lane decision)", 7);
    run<10>("one shift-register step", 7);
    return 0;
}
