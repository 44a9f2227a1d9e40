// micro-benchmark (gfx950): what one wavefront can stream from HBM with K 16-byte-per-lane loads in flight (rows of
// 64 lanes x 16 B = 1 KB, consecutive rows contiguous -- the access pattern of the bit-exact superblock walker), for 1 .. 28
// wavefronts on different CUs.  Little's law: time per row = latency / K until the per-wave op limit (vmcnt: 63).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(64) void k(const char *base, size_t bytesPerWave, int rows, unsigned *sink, long long *ticks) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base) + (size_t)blockIdx.x * bytesPerWave, 0,
                                                                        (int)bytesPerWave, 0x00020000);
    const int vo = threadIdx.x * 16;
    u4 buf[K];
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < K; ++u) buf[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, u * 1024, 0);
    const long long t0 = wall_clock64();
#pragma unroll 1
    for (int r = 0; r + K < rows; r += K) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            acc += buf[u].x ^ buf[u].w;
            buf[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (r + K + u) * 1024, 0);
        }
    }
    const long long t1 = wall_clock64();
#pragma unroll
    for (int u = 0; u < K; ++u) acc += buf[u].y;
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int K> void run(const char *d, int waves, int rows, unsigned *sink, long long *ticks) {
    const size_t per = (size_t)rows * 1024;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<K>, dim3(waves), dim3(64), 0, 0, d, per, rows, sink, ticks);
        (void)hipDeviceSynchronize();
    }
    std::vector<long long> t(waves);
    (void)hipMemcpy(t.data(), ticks, 8 * waves, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (long long v : t) mx = v > mx ? v : mx;
    const double ns = (double)mx * 10.0 / (rows - K);      // wall_clock64: 100 MHz
    printf("waves %3d  K %2d : %6.1f ns per 1-KB row  -> implied latency %6.0f ns, %5.1f GB/s per wave\n", waves, K, ns, ns * K, 1024.0 / ns);
}

int main() {
    const int rows = 16384;                 // 16 MB per wave
    const int maxWaves = 256;
    char *d; unsigned *sink; long long *ticks;
    (void)hipMalloc(&d, (size_t)maxWaves * rows * 1024);
    (void)hipMemset(d, 1, (size_t)maxWaves * rows * 1024);
    (void)hipMalloc(&sink, 4 * 64 * maxWaves); (void)hipMalloc(&ticks, 8 * maxWaves);
    for (int waves : {1, 3, 28, 112, 256}) {
        run<8>(d, waves, rows, sink, ticks);
        run<16>(d, waves, rows, sink, ticks);
        run<32>(d, waves, rows, sink, ticks);
        run<48>(d, waves, rows, sink, ticks);
        run<60>(d, waves, rows, sink, ticks);
    }
    return 0;
}
