// micro-benchmark: host <-> device copies of a 160 MB array (the residual matrix of a chr1 x 32 smoother call) on THIS box:
// pageable (what the drop-in callables get from NumPy), pinned, registered in place, and chunked through a pinned ring with
// the host-side memcpy on 1..8 threads.   hipcc -O3 --offload-arch=gfx950 -o /tmp/pcie scripts/ubench/pcie.hip -lpthread && /tmp/pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(char *dst, const char *src, size_t n, int T) {
    if (T <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n / T + 4095) / 4096 * 4096;
    for (int t = 0; t < T; ++t) {
        const size_t o = per * t;
        if (o >= n) break;
        th.emplace_back([=] { memcpy(dst + o, src + o, std::min(per, n - o)); });
    }
    for (auto &x : th) x.join();
}
int main() {
    const size_t N = 160u << 20;
    char *d; hipMalloc(&d, N); hipMemset(d, 1, N);
    char *pg = (char *)malloc(N); memset(pg, 2, N);
    char *pin; hipHostMalloc(&pin, N, hipHostMallocDefault); memset(pin, 3, N);
    hipStream_t s; hipStreamCreate(&s);
    auto rep = [&](const char *name, auto f) {
        f(); double best = 1e9;
        for (int r = 0; r < 4; ++r) { const double t = now(); f(); best = std::min(best, now() - t); }
        printf("%-64s %7.2f ms  %6.2f GB/s\n", name, best * 1e3, N / best / 1e9);
    };
    rep("H2D pageable hipMemcpy", [&] { hipMemcpy(d, pg, N, hipMemcpyHostToDevice); });
    rep("D2H pageable hipMemcpy (touched pages)", [&] { hipMemcpy(pg, d, N, hipMemcpyDeviceToHost); });
    rep("H2D pinned", [&] { hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); });
    rep("D2H pinned", [&] { hipMemcpyAsync(pin, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); });
    rep("D2H pageable into a FRESH malloc (first touch)", [&] { char *f = (char *)malloc(N); hipMemcpy(f, d, N, hipMemcpyDeviceToHost); free(f); });
    rep("hipHostRegister + H2D + unregister", [&] { hipHostRegister(pg, N, hipHostRegisterDefault); hipMemcpyAsync(d, pg, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); hipHostUnregister(pg); });
    rep("hipHostRegister + D2H + unregister", [&] { hipHostRegister(pg, N, hipHostRegisterDefault); hipMemcpyAsync(pg, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); hipHostUnregister(pg); });
    for (int T : {1, 2, 4, 8}) {
        char nm[96];
        snprintf(nm, sizeof nm, "host memcpy 160 MB, %d thread(s)", T);
        rep(nm, [&] { par_copy(pin, pg, N, T); });
    }
    // chunked ring: memcpy into pinned chunk k while chunk k-1 is on the wire
    for (int T : {1, 4}) for (size_t C : {(size_t)4 << 20, (size_t)16 << 20}) {
        char nm[96];
        hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
        snprintf(nm, sizeof nm, "H2D through a 2-slot pinned ring, %zu MB chunks, %d thread(s)", C >> 20, T);
        rep(nm, [&] {
            int k = 0;
            for (size_t o = 0; o < N; o += C, ++k) {
                const size_t len = std::min(C, N - o);
                char *slot = pin + (k & 1) * C;
                if (k >= 2) hipEventSynchronize(ev[k & 1]);
                par_copy(slot, pg + o, len, T);
                hipMemcpyAsync(d + o, slot, len, hipMemcpyHostToDevice, s);
                hipEventRecord(ev[k & 1], s);
            }
            hipStreamSynchronize(s);
        });
        snprintf(nm, sizeof nm, "D2H through a 2-slot pinned ring, %zu MB chunks, %d thread(s)", C >> 20, T);
        rep(nm, [&] {
            int k = 0; size_t prevO = 0, prevLen = 0;
            for (size_t o = 0; o < N; o += C, ++k) {
                const size_t len = std::min(C, N - o);
                char *slot = pin + (k & 1) * C;
                hipMemcpyAsync(slot, d + o, len, hipMemcpyDeviceToHost, s);
                hipEventRecord(ev[k & 1], s);
                if (k >= 1) { hipEventSynchronize(ev[(k - 1) & 1]); par_copy(pg + prevO, pin + ((k - 1) & 1) * C, prevLen, T); }
                prevO = o; prevLen = len;
            }
            hipEventSynchronize(ev[(k - 1) & 1]); par_copy(pg + prevO, pin + ((k - 1) & 1) * C, prevLen, T);
        });
    }
    return 0;
}
