// micro-benchmark (gfx950): a latency-bound "walker" wavefront (one 1.5-KB record row per step, 16 rows of look-ahead, a chain of
// dependent fp64 operations per step) with 0 .. 3 PREFETCH wavefronts in the same workgroup that stream the same rows into the
// L2 a bounded distance ahead (progress flag in global memory, no barrier).  28 workgroups, like the bit-exact superblock walker.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
#define ROWB 1536
template <int NPF, int DEP>
__global__ __launch_bounds__(64 * (1 + NPF)) void k(const char *base, int rows, int ahead, double *sink, long long *ticks, int *prog) {
    const size_t per = (size_t)rows * ROWB;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base) + (size_t)blockIdx.x * per, 0, (int)per, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int *pg = prog + blockIdx.x * 32;
    if (wave == 0) {
        constexpr int U = 16;
        u4 a[U], a2[U]; u2 b[U], b2[U];
        double x = 1.0 + lane * 1e-3;
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, u * ROWB, 0); b[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, 1024 + lane * 8, u * ROWB, 0); }
        const long long t0 = wall_clock64();
#pragma unroll 1
        for (int r = 0; r + 2 * U < rows; r += 2 * U) {
            if (NPF > 0 && lane == 0) __hip_atomic_store(pg, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < U; ++u) { a2[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (r + U + u) * ROWB, 0); b2[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, 1024 + lane * 8, (r + U + u) * ROWB, 0); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double z = __hiloint2double(a[u].y & 0xfffff | 0x3ff00000, a[u].x) + (double)(b[u].x & 1);
#pragma unroll
                for (int d = 0; d < DEP; ++d) x = fma(x, 0.999999, z * 1e-9);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (r + 2 * U + u) * ROWB, 0); b[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, 1024 + lane * 8, (r + 2 * U + u) * ROWB, 0); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double z = __hiloint2double(a2[u].y & 0xfffff | 0x3ff00000, a2[u].x) + (double)(b2[u].x & 1);
#pragma unroll
                for (int d = 0; d < DEP; ++d) x = fma(x, 0.999999, z * 1e-9);
            }
        }
        const long long t1 = wall_clock64();
        if (NPF > 0 && lane == 0) __hip_atomic_store(pg, 1 << 30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sink[blockIdx.x * 64 + lane] = x;
        if (lane == 0) ticks[blockIdx.x] = t1 - t0;
    } else {
        // prefetch: 1-KB pieces p = wave-1, wave-1+NPF, ... of the block's record region, at most `ahead` rows before the walker
        const int pieces = (int)(per / 1024);
        unsigned acc = 0;
        int seen = 0;
        for (int p0 = (wave - 1); p0 < pieces; p0 += NPF * 8) {
            const int row = (int)(((size_t)p0 * 1024) / ROWB);
            while (row > seen + ahead) { __builtin_amdgcn_s_sleep(8); seen = __hip_atomic_load(pg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int p = p0 + q * NPF;
                if (p < pieces) { const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, p * 1024, 0); acc ^= v.x; }
            }
        }
        if (acc == 0x12345678u) sink[0] = 1.0;
    }
}
template <int NPF, int DEP> void run(const char *d, int wgs, int rows, int ahead, double *sink, long long *ticks, int *prog) {
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(prog, 0, 4 * 32 * wgs);
        hipLaunchKernelGGL((k<NPF, DEP>), dim3(wgs), dim3(64 * (1 + NPF)), 0, 0, d, rows, ahead, sink, ticks, prog);
        (void)hipDeviceSynchronize();
    }
    std::vector<long long> t(wgs);
    (void)hipMemcpy(t.data(), ticks, 8 * wgs, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (long long v : t) mx = v > mx ? v : mx;
    printf("workgroups %3d  prefetch waves %d  dependent fma per row %2d  ahead %4d rows : %6.1f ns per row\n", wgs, NPF, DEP, ahead, (double)mx * 10.0 / (rows - 32));
}
int main() {
    const int rows = 8192, maxW = 64;
    char *d; double *sink; long long *ticks; int *prog;
    (void)hipMalloc(&d, (size_t)maxW * rows * ROWB);
    (void)hipMemset(d, 1, (size_t)maxW * rows * ROWB);
    (void)hipMalloc(&sink, 8 * 64 * maxW); (void)hipMalloc(&ticks, 8 * maxW); (void)hipMalloc(&prog, 4 * 32 * maxW);
    for (int wgs : {3, 28}) {
        run<0, 8>(d, wgs, rows, 0, sink, ticks, prog);
        run<1, 8>(d, wgs, rows, 128, sink, ticks, prog);
        run<2, 8>(d, wgs, rows, 128, sink, ticks, prog);
        run<3, 8>(d, wgs, rows, 128, sink, ticks, prog);
        run<2, 8>(d, wgs, rows, 512, sink, ticks, prog);
        run<3, 8>(d, wgs, rows, 32, sink, ticks, prog);
        run<0, 2>(d, wgs, rows, 0, sink, ticks, prog);
        run<2, 2>(d, wgs, rows, 128, sink, ticks, prog);
        run<3, 2>(d, wgs, rows, 128, sink, ticks, prog);
    }
    return 0;
}
