// micro-benchmark (gfx950), round 4: ISSUE cost of the instructions of the bit-exact state chain for ONE wavefront on its SIMD
// (independent instructions, 8 accumulators), alone on the CU and with one wavefront on each of the CU's four SIMDs.  The
// state chain runs one wavefront per SIMD, so what a round costs is the sum of its instructions' issue costs, not their latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
template <int V>
__global__ void k(float *out, long long *cyc, float a) {
    const int lane = threadIdx.x & 63;
    double d[8]; float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = a + lane * 1e-3 + i; f[i] = a + lane + i; }
    const double c = 1.0000001, e = 1e-9;
    unsigned long long sm = 0x10ull << (lane & 1);
    sm = (unsigned long long)__builtin_amdgcn_readfirstlane((int)sm) | 0x100ull;
    int sg[8]; unsigned long long sm8[8];
    for (int i = 0; i < 8; ++i) { sg[i] = __builtin_amdgcn_readfirstlane(i + (int)a); sm8[i] = sm + i; }
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // (inline asm: hipcc packs, hoists or deletes the plain C++ forms)
                if (V == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(c), "v"(e));
                else if (V == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
                else if (V == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
                else if (V == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(f[i]) : "v"(d[i]));
                else if (V == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[i]) : "v"(f[i]));
                else if (V == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(a));
                else if (V == 6) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                else if (V == 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(a));
                else if (V == 8) asm volatile("v_readlane_b32 %0, %1, 5" : "+s"(sg[i]) : "v"(f[i]));
                else if (V == 9) asm volatile("v_cmp_ne_u32 %0, %1, %2" : "+s"(sm8[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]));
                else if (V == 10) asm volatile("s_ff1_i32_b64 %0, %1" : "+s"(sg[i]) : "s"(sm));
                else if (V == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(e));
                else if (V == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(a) : );
                else if (V == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(f[i]) : "s"(a));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += (float)d[i] + f[i] + (float)sg[i] + (float)(sm8[i] & 0xff);
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V> void run(const char *name, int per) {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 4 * 256 * 256); (void)hipMalloc(&cyc, 8);
    double res[2];
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int threads = cfg == 0 ? 64 : 256, blocks = cfg == 0 ? 1 : 256;
        for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.25f); (void)hipDeviceSynchronize(); }
        long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        res[cfg] = (double)c / N / 32.0 / per;
    }
    printf("%-44s %.2f ticks per instruction alone, %.2f with a wavefront on every SIMD of every CU\n", name, res[0], res[1]);
}
int main() {
    run<0>("v_fma_f64", 1); run<1>("v_add_f64", 1); run<2>("v_mul_f64", 1); run<3>("v_cvt_f32_f64", 1); run<4>("v_cvt_f64_f32", 1);
    run<5>("v_add_f32", 1); run<6>("v_mov_b32_dpp wave_shr:1", 1); run<7>("v_fma_f32", 1);
    run<8>("v_readlane_b32 (const lane)", 1); run<9>("v_cmp_ne_u32 -> sgpr pair", 1); run<10>("s_ff1_i32_b64", 1);
    run<11>("v_pk_add_f32", 1); run<12>("v_cndmask_b32", 1); run<13>("v_mov_b32 (sgpr -> vgpr)", 1);
    return 0;
}
