// Launch layer of libchap_hip.so: every kernel on the networks' forward / backward path is a __device__ body behind ONE generic
// __global__ trampoline that takes up to CHAP_MAX_GROUP argument blocks and runs block (x, y, z) on block z's arguments.
//
// Why: a training iteration is a chain of ~600 dependent launches of kernels that do 2-5 us of work in 8-20 us (fixed cost: launch
// gap, block prologue, one cold memory round trip).  The two decoders of a DualDecoder run the SAME layer shapes on different
// tensors, so do two passes of one network (pass A and the first VAT forward): between chap_group_begin() and chap_group_end()
// the library records the launches per "lane" and then issues the j-th launch of every lane as ONE grid (gridDim.z = lanes) when
// they resolved to the same kernel instance and launch geometry -- same blocks, same per-block work, same reduction slots, hence
// bit-identical results to the one-by-one launches, with one fixed cost instead of 2-4.  Lanes that do not match are launched one
// after the other in lane order.  (include/chap_hip.h: chap_group_begin / chap_group_next_lane / chap_group_end.)
#pragma once
#include <cstring>
#include <type_traits>
#include "common.h"

#ifndef CHAP_MAX_GROUP_N
#define CHAP_MAX_GROUP_N 4
#endif
constexpr int CHAP_MAX_GROUP = CHAP_MAX_GROUP_N;
#ifdef CHAP_TIMELINE
constexpr int CHAP_MAX_LANES = CHAP_MAX_GROUP - 1;      // lanes merged into one grid (lab builds: the last argument block carries the timeline pointer)
#else
constexpr int CHAP_MAX_LANES = CHAP_MAX_GROUP;
#endif
// CHAP_TIMELINE (lab builds only, chap_amd/csrc/Makefile.lab -> tools/timeline_untraced.py): every launch behind the trampolines gets three
// (start, end) pairs: its first, middle and last block (linear block id 0, n / 2, n - 1) store the 100 MHz s_memrealtime counter at their entry and
// exit, plain stores to their own words, so that the CONCURRENT schedule of an untraced graph replay can be read back (a rocprofv3 kernel trace all
// but serialises the graph).  Stamping EVERY block perturbs the step too much: one atomicMax per block on a per-launch word +42 %, a plain store pair
// per block +26 % (the scalar clock read at a block's entry sits in front of its first kernel-argument load); three blocks per launch: see
// profiles/r04_timeline_untraced_*.json for the measured overhead.  A kernel's interval is [min start, max end] of the three (persistent grids end
// together; in a multi-round grid the last block id is dispatched last).  The pointer is baked into the launch (and thus into the captured graph
// node); the product build has neither the field nor the stamps.  The directly launched kernels (losses, largest-CC, VAT helpers, SGD) are followed
// by a one-thread marker kernel on their stream (CHAP_TL_MARK in CHAP_LAUNCH_CHECK, common.h): its time stamp bounds their end.
// The pointer travels in the LAST 8 BYTES of the argument block of lane 3 (lab builds merge at most 3 lanes): the kernel-argument size stays what
// the product build has (a first version with an extra field made every kernel of the lab build slower -- 8.4 vs 6.75 ms per 2D step with the
// stamps switched OFF: the conv's argument block crossed 1 KB).
#ifdef CHAP_TIMELINE
template <typename A> struct chap_group { A p[CHAP_MAX_GROUP]; };
#define CHAP_TL_PTR(G) (*(unsigned long long* const*)((const char*)&(G) + sizeof(G) - 8))
unsigned long long* chap_timeline_next_slot(const char* name, const void* fn, dim3 grid, hipStream_t s);
__device__ __forceinline__ int chap_tl_pair() {      // 0 / 1 / 2 for the first / middle / last block of the grid, -1 for every other block
    if (threadIdx.x != 0) return -1;
    const unsigned long nb = (unsigned long)gridDim.x * gridDim.y * gridDim.z;
    const unsigned long b = (unsigned long)blockIdx.x + (unsigned long)gridDim.x * ((unsigned long)blockIdx.y + (unsigned long)gridDim.y * blockIdx.z);
    return b == 0 ? 0 : (b == nb - 1 ? 2 : (b == nb / 2 ? 1 : -1));
}
#define CHAP_TL_ENTER(G) unsigned long long* const chap_tl_p = CHAP_TL_PTR(G); const int chap_tl_k = chap_tl_p ? chap_tl_pair() : -1; \
    do { if (chap_tl_k >= 0) chap_tl_p[2 * chap_tl_k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define CHAP_TL_EXIT(G) do { if (chap_tl_k >= 0) chap_tl_p[2 * chap_tl_k + 1] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
template <typename A> struct chap_group { A p[CHAP_MAX_GROUP]; };
#define CHAP_TL_ENTER(G) do {} while (0)
#define CHAP_TL_EXIT(G) do {} while (0)
#endif

// group index of this block: gridDim.z = groups x base_z (base_z = the kernel's own grid.z, 1 for all but the weight gradient)
__device__ __forceinline__ int chap_group_index(int base_z = 1) { return base_z == 1 ? (int)blockIdx.z : (int)blockIdx.z / base_z; }

template <typename A, void (*BODY)(const A&), int MAXT, int MINW>
__global__ __launch_bounds__(MAXT, MINW) void chap_grouped(const chap_group<A> G) { CHAP_TL_ENTER(G); BODY(G.p[blockIdx.z]); CHAP_TL_EXIT(G); }
// kernels whose own grid uses z (the weight gradient): the body finds its group from A::base_z
template <typename A, void (*BODY)(const A&, int), int MAXT, int MINW>
__global__ __launch_bounds__(MAXT, MINW) void chap_grouped_z(const chap_group<A> G) {
    const int bz = G.p[0].base_z;
    const int grp = (int)blockIdx.z / bz;
    CHAP_TL_ENTER(G);
    BODY(G.p[grp], (int)blockIdx.z - grp * bz);
    CHAP_TL_EXIT(G);
}

struct chap_pending {
    const void* fn;
    dim3 grid, block;
    unsigned lds;
    int (*merged)(const chap_pending* const* items, int n, hipStream_t s);
    const char* name;
    alignas(16) unsigned char arg[768];
};
bool chap_group_recording();
int chap_group_record(const chap_pending& p, hipStream_t s);

template <typename A>
static int chap_launch_merged(const chap_pending* const* items, int n, hipStream_t s) {
    chap_group<A> g;
    for (int i = 0; i < CHAP_MAX_GROUP; ++i) memcpy(&g.p[i], items[i < n ? i : 0]->arg, sizeof(A));
    dim3 grid = items[0]->grid;
    grid.z *= (unsigned)n;
#ifdef CHAP_TIMELINE
    {   // (lane 3 is never a real lane in a lab build: chap_group_end merges at most CHAP_MAX_LANES = 3)
        unsigned long long* tl = chap_timeline_next_slot(items[0]->name, items[0]->fn, grid, s);
        memcpy((char*)&g + sizeof(g) - 8, &tl, 8);
    }
#endif
    void* args[] = {&g};
    const hipError_t e = hipLaunchKernel(items[0]->fn, grid, items[0]->block, args, items[0]->lds, s);
    if (e != hipSuccess) { (void)hipGetLastError(); chap_set_error("%s: launch failed: %s", items[0]->name, hipGetErrorString(e)); return CHAP_ELAUNCH; }
    return CHAP_OK;
}

template <typename A>
static int chap_launch_ptr(const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t s, const A& a, const char* name) {
    static_assert(sizeof(A) <= sizeof(chap_pending::arg) && std::is_trivially_copyable<A>::value, "kernel argument block too large / not POD");
    chap_pending p;
    p.fn = fn; p.grid = grid; p.block = block; p.lds = (unsigned)lds; p.merged = &chap_launch_merged<A>; p.name = name;
    memcpy(p.arg, &a, sizeof(A));
    if (chap_group_recording()) return chap_group_record(p, s);
    const chap_pending* one = &p;
    return chap_launch_merged<A>(&one, 1, s);
}

template <typename A, void (*BODY)(const A&), int MAXT, int MINW = 1>
static const void* chap_kernel() { return (const void*)chap_grouped<A, BODY, MAXT, MINW>; }

template <typename A, void (*BODY)(const A&), int MAXT, int MINW = 1>
static int chap_launch(dim3 grid, dim3 block, size_t lds, hipStream_t s, const A& a, const char* name) {
    return chap_launch_ptr<A>(chap_kernel<A, BODY, MAXT, MINW>(), grid, block, lds, s, a, name);
}
template <typename A, void (*BODY)(const A&, int), int MAXT, int MINW = 1>
static int chap_launch_z(dim3 grid, dim3 block, size_t lds, hipStream_t s, const A& a, const char* name) {
    return chap_launch_ptr<A>((const void*)chap_grouped_z<A, BODY, MAXT, MINW>, grid, block, lds, s, a, name);
}

// entry points whose kernels are launched directly (not through chap_launch) must not be called inside a group region: their
// launch would overtake the recorded ones
#define CHAP_NOT_IN_GROUP(name) do { if (chap_group_recording()) { chap_set_error("%s: not allowed between chap_group_begin() and chap_group_end()", name); return CHAP_EUNSUPPORTED; } } while (0)
