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
// find with path halving: every visited node is re-pointed at its grandparent.  Plain stores are enough -- a parent pointer only
// ever moves to another (smaller) ancestor of the same tree, a lost update just leaves an older, still valid ancestor.
__device__ __forceinline__ int uf_find_halve(int* L, int i) {
    int r = L[i];
    while (r != i) {
        const int rr = L[r];
        if (rr != r) L[i] = rr;
        i = r; r = rr;
    }
    return r;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    bool done = false;
    while (!done) {
        a = uf_find_halve(L, a); b = uf_find_halve(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }       // a < b: hang b under a
        const int old = atomicMin(&L[b], a);
        done = (old == b);
        b = old;
    }
}

// Initial forest: the links ALONG A ROW cost no atomics.  A pixel whose left neighbour has the same label points at the
// start of its run inside the wave (ballot + bit scan), or -- when the run comes in from the previous wave -- at the pixel just
// left of the wave's first lane: parents always have a smaller index and the same component, chains along a row have at most
// W / 64 + 1 hops.  `rowlen` consecutive linear indices form a row (W).
__global__ void lcc_init_kernel(const int64_t* lab, int* L, unsigned* size, unsigned long long* best, long total, int nbest, int W) {
    const long nwork = (total + 63) / 64 * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nwork; i += (long)gridDim.x * blockDim.x) {
        const bool in = i < total;
        const int64_t c = in ? lab[i] : 0;
        const int x = in ? (int)((unsigned)i % (unsigned)W) : 0;
        const bool fg = c > 0;
        const bool joins_left = fg && x > 0 && lab[i - 1] == c;
        const int lane = threadIdx.x & 63;
        const unsigned long long starts = __ballot(!joins_left);                 // lanes that start a run (or are background / padding)
        int parent = -1;
        if (fg) {
            if (!joins_left) parent = (int)i;
            else {
                const unsigned long long below = starts & ((2ull << lane) - 1ull);   // run starts at or below this lane
                parent = below ? (int)(i - lane + (63 - __clzll((long long)below))) : (int)(i - lane - 1);
            }
        }
        if (in) { L[i] = parent; size[i] = 0; }
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nbest; i += (long)gridDim.x * blockDim.x) best[i] = 0ull;
}
// Links across rows (and planes).  The three (2D) / nine (3D) neighbours in the next row / plane that carry the pixel's label
// are already connected to EACH OTHER through that row's / plane's own links whenever the middle one carries it, so one union
// with the middle neighbour covers them; only when the middle differs are the outer ones united one by one.  Most unions
// find both pixels under one root already (no atomic): the atomicMin's that remain are about one per run, not four per pixel.
__device__ __forceinline__ void lcc_row3(const int64_t* lab, int* L, long i, long q, int x, int W, int64_t c) {
    // q = index of the pixel straight "below" (same x) in a neighbouring row; unite with the label-c pixels among q-1, q, q+1
    if (lab[q] == c) { uf_union(L, (int)i, (int)q); return; }
    if (x > 0 && lab[q - 1] == c) uf_union(L, (int)i, (int)(q - 1));
    if (x + 1 < W && lab[q + 1] == c) uf_union(L, (int)i, (int)(q + 1));
}
__global__ void lcc_merge_kernel(const int64_t* lab, int* L, int N, int D, int H, int W) {
    const long total = (long)N * D * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int64_t c = lab[i];
        if (c <= 0) continue;
        const unsigned ui = (unsigned)i;
        const int x = (int)(ui % (unsigned)W), y = (int)((ui / (unsigned)W) % (unsigned)H), z = (int)((ui / (unsigned)(W * H)) % (unsigned)D);
        if (y + 1 < H) lcc_row3(lab, L, i, i + W, x, W, c);                      // next row of this plane
        if (z + 1 < D) {                                                         // next plane (26-connectivity)
            const long p = i + (long)H * W;
            if (lab[p] == c) { uf_union(L, (int)i, (int)p); continue; }          // the middle of the 3x3: its label-c neighbours hang on it in-plane
            if (x > 0 && lab[p - 1] == c) uf_union(L, (int)i, (int)(p - 1));
            if (x + 1 < W && lab[p + 1] == c) uf_union(L, (int)i, (int)(p + 1));
            if (y > 0) lcc_row3(lab, L, i, p - W, x, W, c);
            if (y + 1 < H) lcc_row3(lab, L, i, p + W, x, W, c);
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
// Largest component per (image, class): key = size << 32 | ~root (ties: the smallest root = the component met first in raster
// order).  Roots are few but, on noisy label maps, thousands of them hit the same few (image, class) words: a block first
// takes its maximum per word in LDS and then issues one global atomicMax per word it touched.
constexpr int LCC_BEST_LDS = 256;
__global__ __launch_bounds__(256) void lcc_best_kernel(const int64_t* lab, const int* L, const unsigned* size, unsigned long long* best, int HW, int ncls, long total, int nbest) {
    __shared__ unsigned long long sbest[LCC_BEST_LDS];
    const bool local = nbest <= LCC_BEST_LDS;
    if (local) { for (int k = threadIdx.x; k < nbest; k += 256) sbest[k] = 0ull; __syncthreads(); }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        if (L[i] != (int)i) continue;               // roots only
        const int n = (int)(i / HW);
        const int c = (int)lab[i];
        if (c >= ncls) continue;
        const unsigned long long key = ((unsigned long long)size[i] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        if (local) atomicMax(&sbest[n * ncls + c], key); else atomicMax(&best[(long)n * ncls + c], key);
    }
    if (local) {
        __syncthreads();
        for (int k = threadIdx.x; k < nbest; k += 256) if (sbest[k]) atomicMax(&best[k], sbest[k]);
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
    hipLaunchKernelGGL(lcc_init_kernel, dim3(nb), dim3(256), 0, s, p->labels, L, size, best, total, nbest, p->W);
    hipLaunchKernelGGL(lcc_merge_kernel, dim3(nb), dim3(256), 0, s, p->labels, L, p->N, D, p->H, p->W);
    hipLaunchKernelGGL(lcc_count_kernel, dim3(nb), dim3(256), 0, s, L, size, total);
    const int nb_best = nb < 512 ? nb : 512;
    hipLaunchKernelGGL(lcc_best_kernel, dim3(nb_best), dim3(256), 0, s, p->labels, (const int*)L, (const unsigned*)size, best, D * p->H * p->W, p->num_classes, total, nbest);
    hipLaunchKernelGGL(lcc_write_kernel, dim3(nb), dim3(256), 0, s, p->labels, (const int*)L, (const unsigned long long*)best, p->out, D * p->H * p->W, p->num_classes, total);
    CHAP_LAUNCH_CHECK("chap_largest_cc");
    return CHAP_OK;
}
