// micro-benchmark (gfx950): at what clock does a latency-bound wavefront run?  A chain of N dependent v_fma_f64 (4 cycles of
// issue each on an otherwise idle SIMD) is timed with the constant 100 MHz counter (wall_clock64) and with s_memtime
// (clock64), for a lone wavefront, for one wavefront per SIMD of 5/8 of the device (what the bit-exact state chain launches),
// and for the same while a bandwidth-bound kernel keeps the rest of the device busy on another stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N (1 << 20)
__global__ void chain(double *out, long long *t, double a) {
    double x = a + threadIdx.x;
    const long long w0 = wall_clock64(), c0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = fma(x, 1.0000001, 1e-9);
    const long long w1 = wall_clock64(), c1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t[0] = w1 - w0; t[1] = c1 - c0; }
}
__global__ void stream(const float4 *a, float4 *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    double *out; long long *t;
    (void)hipMalloc(&out, 8 * 256 * 1024); (void)hipMalloc(&t, 16);
    float4 *A, *B; const size_t n = (size_t)1 << 26;      // 1 GiB each
    (void)hipMalloc(&A, n * 16); (void)hipMalloc(&B, n * 16); (void)hipMemset(A, 0, n * 16);
    hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    auto run = [&](const char *name, int blocks, bool heater) {
        for (int rep = 0; rep < 2; ++rep) {
            if (heater) for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(stream, dim3(2048), dim3(256), 0, s2, A, B, n);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, s1, out, t, 1.0);
            (void)hipDeviceSynchronize();
        }
        long long h[2]; (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        const double ns = (double)h[0] * 10.0;
        printf("%-58s %8.1f us for %d dependent fma: %.2f ns each = %.2f GHz at 4 cycles each; s_memtime ticks per fma %.2f\n", name, ns * 1e-3, N,
               ns / N, 4.0 * N / ns, (double)h[1] / N);
    };
    run("one wavefront-quad (1 workgroup of 256)", 1, false);
    run("149 workgroups of 256 (596 wavefronts, one per SIMD)", 149, false);
    run("256 workgroups of 256 (every SIMD)", 256, false);
    run("149 workgroups + a streaming kernel on another stream", 149, true);
    return 0;
}
