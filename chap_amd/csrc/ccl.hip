// Largest connected component per (image, class > 0), 8-connectivity, entirely on the GPU:
// union-find with atomicMin roots (root = smallest linear index of the component, i.e. the
// component met first in raster order -- the tie-break of np.argmax(np.bincount(...)) over
// skimage.measure.label's raster-ordered labels, train_ours_2D.py:134-136).
#include "common.h"

__device__ __forceinline__ int uf_find(const int* L, int i) {
    int r = L[i];
    while (r != i) { i = r; r = L[i]; }
    return r;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    bool done = false;
    while (!done) {
        a = uf_find(L, a); b = uf_find(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }       // a < b: hang b under a
        const int old = atomicMin(&L[b], a);
        done = (old == b);
        b = old;
    }
}

__global__ void lcc_init_kernel(const int64_t* lab, int* L, unsigned* size, unsigned long long* best, long total, int nbest) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        L[i] = lab[i] > 0 ? (int)i : -1;
        size[i] = 0;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nbest; i += (long)gridDim.x * blockDim.x) best[i] = 0ull;
}
__global__ void lcc_merge_kernel(const int64_t* lab, int* L, int N, int D, int H, int W) {
    const long total = (long)N * D * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int64_t c = lab[i];
        if (c <= 0) continue;
        const unsigned ui = (unsigned)i;
        const int x = (int)(ui % (unsigned)W), y = (int)((ui / (unsigned)W) % (unsigned)H), z = (int)((ui / (unsigned)(W * H)) % (unsigned)D);
        // "forward" half of the 8- (2D) / 26- (3D) neighbourhood: each undirected edge is visited once
        for (int dz = 0; dz <= (D > 1 ? 1 : 0); ++dz)
            for (int dy = (dz ? -1 : 0); dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dz == 0 && (dy < 0 || (dy == 0 && dx <= 0))) continue;
                    const int zz = z + dz, yy = y + dy, xx = x + dx;
                    if (zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const long q = i + ((long)dz * H + dy) * W + dx;
                    if (lab[q] == c) uf_union(L, (int)i, (int)q);
                }
    }
}
__global__ void lcc_count_kernel(int* L, unsigned* size, long total) {
    // Lanes of a wave cover 64 consecutive pixels, usually of ONE component: elect a leader per distinct root
    // and add the whole group's count with one atomic (a 500k-voxel component was 500k same-address atomics).
    const long nwork = (total + 63) / 64 * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nwork; i += (long)gridDim.x * blockDim.x) {
        int r = -1;
        if (i < total && L[i] >= 0) {
            r = uf_find(L, (int)i);
            L[i] = r;                               // path compression; roots keep L[r] == r
        }
        unsigned long long todo = __ballot(r >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lr = __shfl(r, leader, 64);
            const unsigned long long same = __ballot(r == lr) & todo;
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&size[lr], (unsigned)__popcll(same));
            todo &= ~same;
        }
    }
}
__global__ void lcc_best_kernel(const int64_t* lab, const int* L, const unsigned* size, unsigned long long* best, int HW, int ncls, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (L[i] != (int)i) continue;               // roots only
        const int n = (int)(i / HW);
        const int c = (int)lab[i];
        if (c >= ncls) continue;
        const unsigned long long key = ((unsigned long long)size[i] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        atomicMax(&best[(long)n * ncls + c], key);
    }
}
__global__ void lcc_write_kernel(const int64_t* lab, const int* L, const unsigned long long* best, int64_t* out, int HW, int ncls, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int64_t c = lab[i];
        int64_t o = 0;
        if (c > 0 && c < ncls) {
            const int r = L[i];
            const unsigned long long b = best[(long)(i / HW) * ncls + c];
            if ((unsigned)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFull)) == (unsigned)r) o = c;
        }
        out[i] = o;
    }
}

extern "C" size_t chap_lcc_ws(const chap_lcc_params* p) {
    if (!p) return 0;
    const size_t total = (size_t)p->N * (p->D > 1 ? p->D : 1) * p->H * p->W;
    return total * (sizeof(int) + sizeof(unsigned)) + ((size_t)p->N * p->num_classes + 8) * sizeof(unsigned long long) + 64;
}
extern "C" int chap_largest_cc(const chap_lcc_params* p, void* stream) {
    CHAP_CHECK_ARG(p && p->labels && p->out && p->ws, "chap_largest_cc: null argument");
    const int D = p->D > 1 ? p->D : 1;
    const long total = (long)p->N * D * p->H * p->W;
    CHAP_CHECK_ARG(total < 0x7FFFFFFFL, "chap_largest_cc: too many pixels");
    hipStream_t s = (hipStream_t)stream;
    const int nbest = p->N * p->num_classes;
    unsigned long long* best = (unsigned long long*)p->ws;                  // 8-byte aligned first
    int* L = (int*)(best + ((nbest + 7) / 8) * 8);
    unsigned* size = (unsigned*)(L + total);
    const int nb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(lcc_init_kernel, dim3(nb), dim3(256), 0, s, p->labels, L, size, best, total, nbest);
    hipLaunchKernelGGL(lcc_merge_kernel, dim3(nb), dim3(256), 0, s, p->labels, L, p->N, D, p->H, p->W);
    hipLaunchKernelGGL(lcc_count_kernel, dim3(nb), dim3(256), 0, s, L, size, total);
    hipLaunchKernelGGL(lcc_best_kernel, dim3(nb), dim3(256), 0, s, p->labels, (const int*)L, (const unsigned*)size, best, D * p->H * p->W, p->num_classes, total);
    hipLaunchKernelGGL(lcc_write_kernel, dim3(nb), dim3(256), 0, s, p->labels, (const int*)L, (const unsigned long long*)best, p->out, D * p->H * p->W, p->num_classes, total);
    CHAP_LAUNCH_CHECK("chap_largest_cc");
    return CHAP_OK;
}
