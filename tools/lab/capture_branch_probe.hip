// Probe (round 3): parallel branches inside a stream capture WITHOUT a second stream -- hipStreamGetCaptureInfo_v2 +
// hipStreamUpdateCaptureDependencies(hipStreamSetCaptureDependencies) -- on a stream that is itself forked from the capture's origin
// (where an event fork to yet another stream crashes hipStreamEndCapture on ROCm 7.2).  Checks the result and whether the branches overlap.
//   hipcc --offload-arch=gfx950 -O2 -o capture_branch_probe capture_branch_probe.hip && ./capture_branch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 2; } } while (0)

// a slow, narrow kernel: 8 blocks spinning on dependent fmas (leaves most of the GPU free)
__global__ void spin(float* p, int iters, float add) {
    float v = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0000001f + add;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void addk(float* o, const float* a, const float* b) { int i = blockIdx.x * blockDim.x + threadIdx.x; o[i] = a[i] + b[i]; }

static int deps_of(hipStream_t s, std::vector<hipGraphNode_t>& v) {
    hipStreamCaptureStatus st; unsigned long long id; hipGraph_t g; const hipGraphNode_t* d; size_t n;
    CK(hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &d, &n));
    if (st != hipStreamCaptureStatusActive) { printf("not capturing\n"); return 2; }
    v.assign(d, d + n);
    return 0;
}

static int build(bool branches, bool on_forked, hipGraphExec_t* ex, float* a, float* b, float* o, int chain, int iters) {
    hipStream_t origin, side; hipEvent_t e0, e1;
    CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipStreamBeginCapture(origin, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, origin, a, 10, 0.f);
    hipStream_t s = origin;
    if (on_forked) { CK(hipEventRecord(e0, origin)); CK(hipStreamWaitEvent(side, e0, 0)); s = side; hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, origin, o, 10, 0.f); }
    std::vector<hipGraphNode_t> fork, t1, t2;
    if (branches && deps_of(s, fork)) return 2;
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, s, a, iters, 1.f);
    if (branches) {
        if (deps_of(s, t1)) return 2;
        CK(hipStreamUpdateCaptureDependencies(s, fork.data(), fork.size(), hipStreamSetCaptureDependencies));
    }
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, s, b, iters, 2.f);
    if (branches) CK(hipStreamUpdateCaptureDependencies(s, t1.data(), t1.size(), hipStreamAddCaptureDependencies));
    if (on_forked) { CK(hipEventRecord(e1, side)); CK(hipStreamWaitEvent(origin, e1, 0)); }
    hipLaunchKernelGGL(addk, dim3(8), dim3(256), 0, origin, o, a, b);
    hipGraph_t g;
    CK(hipStreamEndCapture(origin, &g));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    size_t ne = 0; CK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
    printf("  graph: %zu nodes, %zu edges\n", nn, ne);
    CK(hipGraphInstantiate(ex, g, nullptr, nullptr, 0));
    return 0;
}

int main() {
    const int n = 8 * 256, chain = 20, iters = 20000;
    float *a, *b, *o; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&o, n * 4));
    std::vector<float> h(n), ref;
    hipStream_t run; CK(hipStreamCreateWithFlags(&run, hipStreamNonBlocking));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int forked = 0; forked < 2; ++forked)
        for (int br = 0; br < 2; ++br) {
            printf("branches=%d on_forked_stream=%d\n", br, forked);
            hipGraphExec_t ex;
            if (build(br, forked, &ex, a, b, o, chain, iters)) return 2;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(a, 0, n * 4, run)); CK(hipMemsetAsync(b, 0, n * 4, run));
                CK(hipEventRecord(t0, run)); CK(hipGraphLaunch(ex, run)); CK(hipEventRecord(t1, run)); CK(hipStreamSynchronize(run));
                CK(hipEventElapsedTime(&ms, t0, t1));
            }
            CK(hipMemcpy(h.data(), o, n * 4, hipMemcpyDeviceToHost));
            if (ref.empty()) ref = h;
            bool same = true; for (int i = 0; i < n; ++i) same = same && h[i] == ref[i];
            printf("  replay %.3f ms, o[0]=%g, equal to the serial capture: %s\n", ms, h[0], same ? "yes" : "NO");
            CK(hipGraphExecDestroy(ex));
        }
    printf("probe done\n");
    return 0;
}
