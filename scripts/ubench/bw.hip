// micro-benchmark: what the memory system of THIS box sustains for the access patterns of the two streaming kernels of
// the step (k_stats: m row streams read once; k_resid: read m rows, write the (n, m) transpose), next to a plain copy.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/bw scripts/ubench/bw.hip && /tmp/bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <functional>

__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}
// read-only: every thread owns 4 consecutive bins and walks ROWS rows (stride = row length), UN rows in flight
template <int UN>
__global__ __launch_bounds__(256) void k_rows(const float *__restrict__ a, const float *__restrict__ b, int64_t n, int rows,
                                              float *out) {
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (g >= n) return;
    float acc = 0.f;
    for (int j = 0; j < rows; j += UN) {
        float4 x[UN], y[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            x[u] = *reinterpret_cast<const float4 *>(a + (int64_t)(j + u) * n + g);
            y[u] = *reinterpret_cast<const float4 *>(b + (int64_t)(j + u) * n + g);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += x[u].x + x[u].w + y[u].y + y[u].z;
    }
    if (acc == 123456.789f) out[g] = acc;
}
// write-only
__global__ __launch_bounds__(256) void k_fill(float4 *b, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <bool NT>
__global__ __launch_bounds__(256) void k_fill_nt(float4 *b, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        typedef float vf4 __attribute__((ext_vector_type(4)));
        const vf4 v = {1.f, 2.f, 3.f, 4.f};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<vf4 *>(b) + i); else reinterpret_cast<vf4 *>(b)[i] = v;
    }
}

static double run(const char *name, double bytes, std::function<void()> f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %6.2f TB/s\n", name, best, bytes / best / 1e9);
    return best;
}
#include <functional>
int main() {
    const int64_t n = 14375040, m = 32;        // bins (multiple of 64), samples
    float *a, *b, *c;
    hipMalloc(&a, 4 * n * m); hipMalloc(&b, 4 * n * m); hipMalloc(&c, 4 * n * m);
    hipMemset(a, 0, 4 * n * m); hipMemset(b, 0, 4 * n * m);
    const int64_t n4 = n * m / 4;
    run("float4 copy 1.84 GB -> 1.84 GB (grid-stride)", 8.0 * n * m, [&] { k_copy<<<8192, 256>>>((float4 *)a, (float4 *)c, n4); });
    run("hipMemcpyAsync D2D 1.84 GB", 8.0 * n * m, [&] { hipMemcpyAsync(c, a, 4 * n * m, hipMemcpyDeviceToDevice, 0); });
    run("read 2 x 32 row streams, float4, 4 rows in flight", 8.0 * n * m, [&] { k_rows<4><<<(unsigned)((n / 4 + 255) / 256), 256>>>(a, b, n, m, c); });
    run("read 2 x 32 row streams, float4, 8 rows in flight", 8.0 * n * m, [&] { k_rows<8><<<(unsigned)((n / 4 + 255) / 256), 256>>>(a, b, n, m, c); });
    run("fill 1.84 GB (plain stores)", 4.0 * n * m, [&] { k_fill_nt<false><<<8192, 256>>>((float4 *)c, n4); });
    run("fill 1.84 GB (nontemporal stores)", 4.0 * n * m, [&] { k_fill_nt<true><<<8192, 256>>>((float4 *)c, n4); });
    return 0;
}
