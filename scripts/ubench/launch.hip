// micro-benchmark: cost of a kernel boundary on one stream (dependent launches of a near-empty kernel with the grid of a
// 1/8-genome shard's chain kernels: 830 workgroups of 64 threads), and of the same with a chain of 1 .. 4 DEPENDENT global
// loads in front (block table -> chain id -> activity -> carry: what a chain kernel's prologue does).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/launch scripts/ubench/launch.hip && /tmp/launch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEPTH>
__global__ __launch_bounds__(64) void k_chain(const int *a, int *out) {
    int idx = blockIdx.x * 64 + threadIdx.x;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) idx = a[idx];          // a[i] == i: every hop is a fresh dependent load of the same line set
    if (idx == -1) out[0] = idx;
}
template <int WIDTH>
__global__ __launch_bounds__(64) void k_wide(const int *a, int *out) {      // WIDTH independent loads
    const int idx = blockIdx.x * 64 + threadIdx.x;
    int acc = 0;
#pragma unroll
    for (int d = 0; d < WIDTH; ++d) acc += a[idx + d * 830 * 64];
    if (acc == -1) out[0] = acc;
}
template <class F>
static void run(const char *name, F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-44s %6.2f us per dependent launch\n", name, ms * 1000.f / 200.f);
}
int main() {
    const int G = 830, N = G * 64;
    int *a, *out;
    hipMalloc(&a, sizeof(int) * N * 4); hipMalloc(&out, 64);
    int *h = new int[N * 4];
    for (int i = 0; i < N * 4; ++i) h[i] = i % N;
    hipMemcpy(a, h, sizeof(int) * N * 4, hipMemcpyHostToDevice);
    run("0 loads", [&] { k_chain<0><<<G, 64>>>(a, out); });
    run("1 load", [&] { k_chain<1><<<G, 64>>>(a, out); });
    run("2 dependent loads", [&] { k_chain<2><<<G, 64>>>(a, out); });
    run("3 dependent loads", [&] { k_chain<3><<<G, 64>>>(a, out); });
    run("4 dependent loads", [&] { k_chain<4><<<G, 64>>>(a, out); });
    run("4 independent loads", [&] { k_wide<4><<<G, 64>>>(a, out); });
    run("0 loads, grid 26", [&] { k_chain<0><<<26, 64>>>(a, out); });
    run("4 dependent loads, grid 26", [&] { k_chain<4><<<26, 64>>>(a, out); });
    return 0;
}
