// micro-benchmark: dependent-chain latency and independent issue rate of the fp64 ops used by the recurrences (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 4096
template <int OP, int ILP>
__global__ void k(double *out, long long *cyc, double a, double b) {
    double x[ILP];
    for (int i = 0; i < ILP; ++i) x[i] = a + threadIdx.x * 1e-3 + i;
    long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int it = 0; it < N / 16; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (OP == 0) x[i] = fma(x[i], b, a);
                if (OP == 1) x[i] = x[i] * b;
                if (OP == 2) x[i] = x[i] + a;
                if (OP == 3) x[i] = (double)(float)x[i];          // 2 cvts
                if (OP == 4) x[i] = __builtin_amdgcn_rcp(x[i]);
                if (OP == 5) x[i] = a / x[i];                      // IEEE div
                if (OP == 6) { float f = (float)x[i]; f = f * 1.0000001f + 1e-9f; x[i] = (double)f; } // cvt+f32 fma+cvt
                if (OP == 7) { float f = __double2float_rn(x[i]); x[i] = (double)f + a; }   // r32 + add
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP, int ILP> void run(const char *name, int waves) {
    double *out; long long *cyc;
    hipMalloc(&out, 8 * 64 * 4096); hipMalloc(&cyc, 8 * 4096);
    k<OP, ILP><<<1, 64 * waves>>>(out, cyc, 1.0000001, 0.9999999);
    hipDeviceSynchronize();
    k<OP, ILP><<<1, 64 * waves>>>(out, cyc, 1.0000001, 0.9999999);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s ILP=%d waves/WG=%d : %.1f cycles per op-group (%.1f per op)\n", name, ILP, waves, (double)c / N, (double)c / N / ILP);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 1>("v_fma_f64 dependent", 1); run<0, 4>("v_fma_f64 4 indep", 1); run<0, 8>("v_fma_f64 8 indep", 1);
    run<1, 1>("v_mul_f64 dependent", 1); run<1, 4>("v_mul_f64 4 indep", 1);
    run<2, 1>("v_add_f64 dependent", 1); run<2, 4>("v_add_f64 4 indep", 1);
    run<3, 1>("r32 (2 cvt) dependent", 1); run<3, 4>("r32 (2 cvt) 4 indep", 1);
    run<4, 1>("v_rcp_f64 dependent", 1); run<4, 4>("v_rcp_f64 4 indep", 1);
    run<5, 1>("IEEE div dependent", 1); run<5, 4>("IEEE div 4 indep", 1);
    run<6, 1>("cvt+f32fma+cvt dependent", 1);
    run<7, 1>("r32+add dependent", 1);
    run<0, 1>("v_fma_f64 dependent", 4); run<0, 4>("v_fma_f64 4 indep", 4);
    return 0;
}
