// micro-benchmark (gfx950): what a trip between the vector and the scalar unit costs on a dependent chain -- the shape of a
// round of k_sb_delta: vector compare -> lane mask -> s_ff1 -> v_readlane (scalar lane select) -> vector use of the result.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int V>
__global__ void k(float *out, long long *cyc, float a) {
    const int lane = threadIdx.x;
    float x = a + lane * 1e-3f, y = a * 0.5f + lane;
    int pos = 0;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N; ++it) {
        if (V == 0) {                   // vector only: 4 dependent v_add_f32
            x = x + y; x = x + y; x = x + y; x = x + y;
        } else if (V == 1) {            // compare -> mask -> s_ff1 -> v_readlane -> vector add
            const unsigned long long m = __builtin_amdgcn_uicmp(__float_as_uint(x), __float_as_uint(y), 33 /* NE */) | 1ull;
            const int f = __ffsll((long long)(m >> (pos & 31))) - 1;
            const float b = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(y), f & 63));
            x = x + b;
            pos = f + 1;
        } else if (V == 2) {            // v_readlane with a CONSTANT lane -> vector add (vector -> scalar -> vector, no scalar ALU)
            const float b = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(x), 5));
            x = x + b;
        } else if (V == 3) {            // compare -> mask -> scalar popcount -> vector add of the scalar (no readlane)
            const unsigned long long m = __builtin_amdgcn_uicmp(__float_as_uint(x), __float_as_uint(y), 33);
            x = x + (float)__popcll(m);
        } else if (V == 4) {            // the same through DPP only: wave shift + add (stays in vector registers)
            const float s = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
            x = x + s;
        } else if (V == 5) {            // ds_bpermute broadcast of a lane chosen per lane (vector only, through LDS hardware)
            const int idx = ((int)x & 63) << 2;
            const float b = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)__float_as_uint(y)));
            x = x + b;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + pos;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V> void run(const char *name) {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 4 * 64); (void)hipMalloc(&cyc, 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, 1.25f); (void)hipDeviceSynchronize(); }
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-78s %.1f cycles per iteration\n", name, (double)c / N);
}
int main() {
    run<0>("4 dependent v_add_f32");
    run<1>("compare -> mask -> s_ff1 -> v_readlane(scalar lane) -> v_add");
    run<2>("v_readlane(constant lane) -> v_add");
    run<3>("compare -> mask -> s_bcnt1 -> v_cvt -> v_add");
    run<4>("v_mov_dpp wave_shr:1 -> v_add");
    run<5>("ds_bpermute -> v_add");
    return 0;
}
