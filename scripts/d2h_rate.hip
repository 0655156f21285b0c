// Device -> pinned host copy rate against chunk size, number of streams and per-chunk events.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/d2h_rate.hip -o build/d2h_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t total = (size_t)1 << 30, ring_bytes = (size_t)256 << 20;
    char *dev, *host;
    CK(hipMalloc(&dev, total));
    CK(hipMemset(dev, 1, total));
    CK(hipHostMalloc(&host, ring_bytes, hipHostMallocDefault));
    hipStream_t st[4];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> evs(4096);
    for (auto &e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int dir = 0; dir < 2; dir++)
    for (int ns : {1, 2, 4})
        for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20})
            for (int with_ev : {0, 1}) {
                double best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipDeviceSynchronize());
                    const double t0 = now();
                    size_t k = 0;
                    for (size_t off = 0; off < total; off += chunk, k++) {
                        hipStream_t s = st[k % ns];
                        char *h = host + (off % ring_bytes);
                        if (dir == 0) CK(hipMemcpyAsync(h, dev + off, chunk, hipMemcpyDeviceToHost, s));
                        else CK(hipMemcpyAsync(dev + off, h, chunk, hipMemcpyHostToDevice, s));
                        if (with_ev) CK(hipEventRecord(evs[k], s));
                    }
                    CK(hipDeviceSynchronize());
                    const double dt = now() - t0;
                    if (dt < best) best = dt;
                }
                printf("%s streams=%d chunk=%3zu MiB events=%d  %6.2f GB/s\n", dir ? "H2D" : "D2H", ns, chunk >> 20, with_ev, total / best / 1e9);
            }
    return 0;
}
