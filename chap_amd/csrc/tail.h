// In-launch totals of per-block partial rows: "the block that arrives last finishes", in two levels (round 4).
//
// The BatchNorm statistics of a conv and the two sums of the BatchNorm backward are per-block partial rows that a second, tiny kernel totalled
// (bn_finalize_kernel, act_bwd_sum_kernel: ~150 of the ~740 launches of a 2D iteration, 4-5 us each plus the ~4 us gap in front of a dependent
// launch -- every one of them on a dependency chain).  Here the producing launch totals its own rows:
//   * every block stores its row with write-through (sc1) stores, drains them, and takes a ticket of its GROUP of CHAP_TAIL_GROUP rows;
//   * the block that draws a group's last ticket adds the group's rows (fp64, row order) into the group's fp64 row and takes a second-level ticket;
//   * the block that draws the last second-level ticket adds the group rows (group order) and finishes (scale / shift, parameter gradients).
// Two hops of one round of loads each, whatever the row count (<= 1024 rows = 32 groups); a launch with <= 32 rows has one hop.  The total is the same
// sum in the same order whichever blocks happen to be last.  Cross-XCD visibility follows MI355X_MICROARCH.md (inter-workgroup visibility): the rows
// are stored sc1 and drained (s_waitcnt vmcnt(0) in every storing wave, then the workgroup barrier) before ONE lane's relaxed agent-scope ticket add;
// the finisher runs an agent-scope acquire behind its ticket and reads the rows with sc1 loads.  The ticket words are zeroed by the host side before
// the launch (one memset per pass for all layers), never by the kernels: a launch that died cannot poison the next one.
#pragma once
#include "common.h"
#include "chap_hip.h"

// CHAP_TAIL_GROUP rows per first-level group, CHAP_TAIL_GROUPS groups, CHAP_TAIL_TICKETS ticket words: chap_hip.h

__device__ __forceinline__ void tail_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tail_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float tail_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double tail_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane: ticket of `word`; true when it was the last of `expected`.  The acquire sits behind the returned ticket (the wave has to wait for the
// value anyway), in front of the barrier that releases the other waves to their loads.
__device__ __forceinline__ bool tail_ticket(int* word, int expected) {
    const int t = __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == expected - 1;
    if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return last;
}

// Called by EVERY block of the launch, by all its threads, after the block stored its share of row `row` with tail_store().
//   rows: fp32 [nrows][W];  arrivals_per_row: blocks that share one row (gridDim.y of the convs: each stores its channels of the row)
//   tickets: int32 [1 + CHAP_TAIL_GROUPS], zero at launch;  grows: fp64 [CHAP_TAIL_GROUPS][W] scratch (nothing to zero)
//   tot: LDS, fp64 [W];  flag: LDS, one int -- LDS the block no longer uses
// Returns true in exactly ONE block of the launch, whose `tot` then holds the totals of all rows (visible to all its threads).
template <int THREADS>
__device__ __forceinline__ bool tail_total(const float* rows, int W, int nrows, int row, int arrivals_per_row, int* tickets, double* grows,
                                           double* tot, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's row stores have left
    __syncthreads();
    const int grp = row / CHAP_TAIL_GROUP, ngroups = (nrows + CHAP_TAIL_GROUP - 1) / CHAP_TAIL_GROUP;
    const int r0 = grp * CHAP_TAIL_GROUP, rn = min(CHAP_TAIL_GROUP, nrows - r0);
    if (threadIdx.x == 0) *flag = tail_ticket(tickets + 1 + grp, rn * arrivals_per_row) ? 1 : 0;
    __syncthreads();
    if (*flag == 0) return false;
    // ---- this block finishes its group: one round of loads per value (all CHAP_TAIL_GROUP in flight), added in row order
    for (int v = threadIdx.x; v < W; v += THREADS) {
        float x[CHAP_TAIL_GROUP];
#pragma unroll
        for (int i = 0; i < CHAP_TAIL_GROUP; ++i) {              // (uniform row pointer + one 32-bit lane offset: no 64-bit address per load)
            const float* rp = rows + (long)(r0 + (i < rn ? i : 0)) * W;
            x[i] = tail_load(rp + (unsigned)v);
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < CHAP_TAIL_GROUP; ++i) s += i < rn ? (double)x[i] : 0.0;
        if (ngroups == 1) tot[v] = s; else tail_store(grows + (long)grp * W + v, s);
    }
    if (ngroups == 1) { __syncthreads(); return true; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                            // (also: every thread has read *flag)
    if (threadIdx.x == 0) *flag = tail_ticket(tickets, ngroups) ? 1 : 0;
    __syncthreads();
    if (*flag == 0) return false;
    // ---- this block finishes the launch
    for (int v = threadIdx.x; v < W; v += THREADS) {
        double x[CHAP_TAIL_GROUPS];
#pragma unroll
        for (int g = 0; g < CHAP_TAIL_GROUPS; ++g) {
            const double* gp = grows + (long)(g < ngroups ? g : 0) * W;
            x[g] = tail_load(gp + (unsigned)v);
        }
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < CHAP_TAIL_GROUPS; ++g) s += g < ngroups ? x[g] : 0.0;
        tot[v] = s;
    }
    __syncthreads();
    return true;
}

// ---- BatchNorm finalize by the last block of a conv launch (chap_conv_params.fin): what bn_finalize_kernel does, from the fp64 totals.
// tot: [2][Clog] (sum(x - c), sum((x - c)^2) per LOGICAL channel; the sub-lattice rows of a transposed conv, Clog = nsub x C, are folded here)
template <int THREADS>
__device__ __forceinline__ void bn_finalize_from_totals(const chap_bn_fin_t& F, const double* tot, int Clog, const float* stats_shift) {
    const int C = F.C, nsub = Clog / C;
    const bool upd = F.momentum > 0.f && F.running_mean != nullptr;
    for (int c = threadIdx.x; c < C; c += THREADS) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < nsub; ++k) { s += tot[k * C + c]; q += tot[Clog + k * C + c]; }
        const double cnt = (double)F.count;
        const double ms = s / cnt;                               // mean of (x - shift)
        double var = q / cnt - ms * ms;
        var = var > 0.0 ? var : 0.0;
        const float mean = (float)((double)(stats_shift ? stats_shift[c] : 0.f) + ms);
        const float invstd = (float)(1.0 / sqrt(var + (double)F.eps));
        const float sc = F.gamma[c] * invstd;
        F.affine[c] = sc;
        F.affine[C + c] = F.beta[c] - mean * sc;
        F.affine[2 * C + c] = mean;
        F.affine[3 * C + c] = invstd;
        if (upd) {
            const float unb = (float)(cnt > 1.0 ? var * cnt / (cnt - 1.0) : var);
            F.running_mean[c] = (1.f - F.momentum) * F.running_mean[c] + F.momentum * mean;
            F.running_var[c] = (1.f - F.momentum) * F.running_var[c] + F.momentum * unb;
        }
    }
    if (threadIdx.x == 0 && F.num_batches_tracked && F.momentum > 0.f) *F.num_batches_tracked += 1;
}

// LDS the last block needs (fp64 totals + the flag); the launchers request at least this much
static inline size_t conv_fin_lds_bytes(const chap_conv_params* p) { return p->fin.tickets ? (size_t)2 * p->Cout * sizeof(double) + 16 : 0; }

// The end of every statistics-producing conv kernel: the block's partial slot from the four waves' LDS rows (bstat: [4 waves][2][16 NT], already
// behind a barrier), and with P.fin the in-launch finalize.  smem: the block's dynamic LDS (free by now: tail_total opens with a barrier).
template <int NT>
__device__ __forceinline__ void conv_stats_store(const chap_conv_params& P, const float* bstat, int nt0, char* smem) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *(int*)P.stats = (int)gridDim.x;      // header: slots in use
    // slot rows are indexed by the LOGICAL channel (a transposed conv's sub-lattices are folded by the finalize)
    float* st = P.stats + CHAP_STATS_HDR + (long)blockIdx.x * 2 * P.Cout;
    const bool tail = P.fin.tickets != nullptr;
    for (int i = threadIdx.x; i < 2 * 16 * NT; i += 256) {
        const int which = i / (16 * NT), k = i % (16 * NT);
        const int nl = nt0 * 16 + k;
        const float v = (bstat[(0 * 2 + which) * 16 * NT + k] + bstat[(1 * 2 + which) * 16 * NT + k]) + (bstat[(2 * 2 + which) * 16 * NT + k] + bstat[(3 * 2 + which) * 16 * NT + k]);
        if (nl < P.Cout) { if (tail) tail_store(st + which * P.Cout + nl, v); else st[which * P.Cout + nl] = v; }
    }
    if (tail) {
        double* tot = (double*)smem;
        int* flag = (int*)(tot + 2 * P.Cout);
        if (tail_total<256>(P.stats + CHAP_STATS_HDR, 2 * P.Cout, (int)gridDim.x, (int)blockIdx.x, (int)gridDim.y, P.fin.tickets, P.fin.rows, tot, flag))
            bn_finalize_from_totals<256>(P.fin, tot, P.Cout, P.stats_shift);
    }
}
