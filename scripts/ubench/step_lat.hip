// micro-benchmark (gfx950): dependent latency of candidate formulations of the levelTrend STATE step (pyx:403-406, 477-479)
// for the bit-exact walker (one wavefront per SIMD: the step is a chain of dependent instructions).  Records are register
// constants, so only the arithmetic path is timed.    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off step_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 8192
__device__ __forceinline__ double r32(double x) { return (double)(float)x; }

template <int V>
__global__ void k(float *out, long long *cyc, double gs, double zbar, double p00, double p10, double f01, float c0, float c1, double magic) {
    float x0f = c0 + threadIdx.x * 1e-3f, x1f = c1;
    double x0 = x0f, x1 = x1f;
    const double g0 = p00 * gs, g1 = p10 * gs;
    long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int it = 0; it < N / 16; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (V == 0) {           // shipped form: carries as doubles
                const double xp0 = r32(fma(f01, x1, x0));
                const double dl = gs * (zbar - xp0);
                const float a = (float)fma(p00, dl, xp0), b = (float)fma(p10, dl, x1);
                x0 = (double)a; x1 = (double)b;
            } else if (V == 1) {    // float32 carries, F01 == 1: the predicted level is ONE float32 add (innocuous double rounding)
                const float xpf = x0f + x1f;
                const double xp0 = (double)xpf, x1d = (double)x1f;
                const double dl = gs * (zbar - xp0);
                x0f = (float)fma(p00, dl, xp0); x1f = (float)fma(p10, dl, x1d);
            } else if (V == 2) {    // + gain products precomputed off the path
                const float xpf = x0f + x1f;
                const double xp0 = (double)xpf, x1d = (double)x1f;
                const double dz = zbar - xp0;
                x0f = (float)fma(g0, dz, xp0); x1f = (float)fma(g1, dz, x1d);
            } else if (V == 3) {    // doubles throughout, rounding by a magic constant valid for the level's binade (trend: cvt)
                const double xp0 = ((x0 + x1) + magic) - magic;
                const double dz = zbar - xp0;
                x0 = (fma(g0, dz, xp0) + magic) - magic;
                x1 = r32(fma(g1, dz, x1));
            } else if (V == 4) {    // only the conversions: float -> double -> float chain
                x0f = (float)((double)x0f * 1.0);
                x0f = (float)((double)x0f * 1.0);
            } else if (V == 5) {    // dependent float32 adds
                x0f = x0f + x1f; x0f = x0f + x1f; x0f = x0f + x1f; x0f = x0f + x1f;
            } else if (V == 6) {    // dependent fp64 fma
                x0 = fma(x0, gs, zbar); x0 = fma(x0, gs, zbar); x0 = fma(x0, gs, zbar); x0 = fma(x0, gs, zbar);
            } else if (V == 7) {    // dependent fp64 add
                x0 = x0 + zbar; x0 = x0 + gs; x0 = x0 + zbar; x0 = x0 + gs;
            } else if (V == 8) {    // v_cvt_f32_f64 + v_cvt_f64_f32 pair, dependent, x4
                x0 = r32(x0); x0 = r32(x0 + 0.0); x0 = r32(x0); x0 = r32(x0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0f + x1f + (float)x0 + (float)x1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V> void run(const char *name, double per) {
    float *out; long long *cyc;
    hipMalloc(&out, 4 * 64); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        k<V><<<1, 64>>>(out, cyc, 0.26, 3.1, 0.8, 0.05, 1.0, 3.0f, 1e-4f, 1.5 * 4503599627370496.0 / 8388608.0 * 2.0);
        hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %.1f counter ticks per step (%.1f per op)\n", name, (double)c / N, (double)c / N / per);
    hipFree(out); hipFree(cyc);
}
int main() {
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("device clock %d kHz, wall clock %d kHz (readcyclecounter = s_memtime ticks)\n", clk, wclk);
    run<0>("V0 shipped step (doubles, 2 x r32)", 1);
    run<1>("V1 float32 carries, v_add_f32 prediction", 1);
    run<2>("V2 V1 + precomputed gain products", 1);
    run<3>("V3 magic-constant rounding of the level", 1);
    run<4>("cvt f32->f64, mul, cvt f64->f32  (x2 per step)", 2);
    run<5>("v_add_f32 dependent (x4 per step)", 4);
    run<6>("v_fma_f64 dependent (x4 per step)", 4);
    run<7>("v_add_f64 dependent (x4 per step)", 4);
    run<8>("r32 = cvt pair dependent (x4 per step)", 4);
    return 0;
}
