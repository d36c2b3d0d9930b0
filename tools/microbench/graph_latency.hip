// Launch + completion latency of k tiny dependent kernels: separate launches vs one captured hipGraph (run on the GPU box).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(float* p, volatile unsigned* flag, unsigned v) {
    p[threadIdx.x] += 1.0f;
    if (flag && threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* d; hipMalloc(&d, 4096);
    unsigned* flag; hipHostMalloc(&flag, 64, hipHostMallocMapped); *flag = 0;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int k = 1; k <= 3; ++k) {
        std::vector<double> a, b, c, e;
        for (int it = 0; it < 1200; ++it) {
            double t0 = now();
            for (int i = 0; i < k; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (volatile unsigned*)nullptr, 0u);
            hipStreamSynchronize(s);
            a.push_back(now() - t0);
        }
        // same, completion by polling a pinned flag written by the last kernel
        for (int it = 0; it < 1200; ++it) {
            const unsigned v = (unsigned)(it + 1 + 100000 * k);
            double t0 = now();
            for (int i = 0; i < k; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, i == k - 1 ? (volatile unsigned*)flag : (volatile unsigned*)nullptr, v);
            while (*(volatile unsigned*)flag != v) {}
            c.push_back(now() - t0);
        }
        hipStreamSynchronize(s);
        // completion by a stream write of the pinned flag (no kernel touches it)
        for (int it = 0; it < 1200; ++it) {
            const unsigned v = (unsigned)(it + 7 + 300000 * k);
            double t0 = now();
            for (int i = 0; i < k; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (volatile unsigned*)nullptr, 0u);
            hipStreamWriteValue32(s, flag, v, 0);
            while (*(volatile unsigned*)flag != v) {}
            e.push_back(now() - t0);
        }
        hipStreamSynchronize(s);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (volatile unsigned*)nullptr, 0u);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int it = 0; it < 1200; ++it) {
            double t0 = now();
            hipGraphLaunch(ge, s);
            hipStreamSynchronize(s);
            b.push_back(now() - t0);
        }
        auto med = [](std::vector<double>& v) { std::sort(v.begin() + 200, v.end()); return v[200 + (v.size() - 200) / 2]; };
        printf("%d kernels: launches + sync %.1f us, launches + in-kernel flag poll %.1f us, launches + stream-write flag poll %.1f us, graph + sync %.1f us\n", k, med(a), med(c), med(e), med(b));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
