// Probe (round 3, for the round-4 plan): what would ORDER-INDEPENDENT statistics cost?  Today every conv block writes its BatchNorm partial
// sums to a slot of its own and chap_bn_finalize (one more launch on the chain, 5-9 us) reduces the <= 1024 slots in fixed order.  Integer
// (fixed-point) atomics are associative, hence bitwise reproducible in any order: with them the totals would exist when the conv ends and
// every consumer could derive scale / shift from 2C numbers in its prologue -- no finalize launch.  This measures the tail such atomics add:
// `blocks` blocks of 256 threads "compute" for a fixed time, then add 2*C values (a) to private slots, (b) with one u64 atomic per value,
// (c) with two u64 atomics per value (96-bit fixed point: hi and lo words, exact for any magnitude).
//   hipcc --offload-arch=gfx950 -O2 -o atomic_stats_probe atomic_stats_probe.hip && ./atomic_stats_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)

template <int MODE>
__global__ void k(float* slots, unsigned long long* acc, int C2, int spin) {
    float v = threadIdx.x * 1e-3f + blockIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0000001f + 0.5f;           // stands for the tile's work
    const int t = threadIdx.x;
    if (t < C2) {
        if (MODE == 0) slots[(long)blockIdx.x * C2 + t] = v;
        else {
            const long long fx = (long long)(v * 16777216.f);
            if (MODE == 1) __hip_atomic_fetch_add(acc + t, (unsigned long long)fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else {
                __hip_atomic_fetch_add(acc + 2 * t, (unsigned long long)(fx & 0xffffffffll), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(acc + 2 * t + 1, (unsigned long long)(fx >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int main() {
    float* slots; unsigned long long* acc;
    CK(hipMalloc(&slots, 2048l * 512 * 4)); CK(hipMalloc(&acc, 1024 * 8)); CK(hipMemset(acc, 0, 1024 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int cfgs[][2] = {{512, 16}, {1024, 16}, {512, 32}, {384, 32}, {280, 32}, {96, 128}, {2048, 16}};
    for (auto& c : cfgs) {
        const int blocks = c[0], C2 = 2 * c[1];
        for (int spin : {0, 2000}) {
            float us[3];
            for (int mode = 0; mode < 3; ++mode) {
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, slots, acc, C2, spin);
                    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, slots, acc, C2, spin);
                    else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, slots, acc, C2, spin);
                };
                for (int i = 0; i < 20; ++i) launch();
                CK(hipEventRecord(e0));
                for (int i = 0; i < 200; ++i) launch();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                us[mode] = ms * 1e3f / 200;
            }
            printf("blocks=%4d C=%3d spin=%4d : slots %.2f us | 1 atomic/value %.2f us | 2 atomics/value %.2f us\n", blocks, c[1], spin, us[0], us[1], us[2]);
        }
    }
    return 0;
}
