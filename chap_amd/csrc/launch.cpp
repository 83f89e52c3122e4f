// Recording / merging of launches between chap_group_begin() and chap_group_end() (launch.h).
#include <vector>
#include "launch.h"

namespace {
struct group_state {
    bool on = false;
    hipStream_t stream = nullptr;
    std::vector<std::vector<chap_pending>> lanes;
    int error = CHAP_OK;
};
thread_local group_state G;

bool same_launch(const chap_pending& a, const chap_pending& b) {
    return a.fn == b.fn && a.lds == b.lds && a.grid.x == b.grid.x && a.grid.y == b.grid.y && a.grid.z == b.grid.z &&
           a.block.x == b.block.x && a.block.y == b.block.y && a.block.z == b.block.z;
}
}  // namespace

bool chap_group_recording() { return G.on; }

int chap_group_record(const chap_pending& p, hipStream_t s) {
    if (s != G.stream) {
        chap_set_error("%s: launched on another stream than the one given to chap_group_begin()", p.name);
        G.error = CHAP_EINVAL;
        return CHAP_EINVAL;
    }
    G.lanes.back().push_back(p);
    return CHAP_OK;
}

extern "C" int chap_group_begin(void* stream) {
    if (G.on) { chap_set_error("chap_group_begin: already recording (regions do not nest)"); return CHAP_EINVAL; }
    G.on = true;
    G.stream = (hipStream_t)stream;
    G.error = CHAP_OK;
    G.lanes.clear();
    G.lanes.emplace_back();
    return CHAP_OK;
}

extern "C" int chap_group_next_lane(void) {
    if (!G.on) { chap_set_error("chap_group_next_lane: not recording"); return CHAP_EINVAL; }
    G.lanes.emplace_back();
    return CHAP_OK;
}

// Issues what was recorded: position j of every lane together when the launches match, at most CHAP_MAX_GROUP per grid;
// returns the number of grids launched (>= 0) or a negative error code.
extern "C" int chap_group_end(void) {
    if (!G.on) { chap_set_error("chap_group_end: not recording"); return CHAP_EINVAL; }
    G.on = false;
    int rc = G.error, launched = 0;
    size_t maxlen = 0;
    for (const auto& l : G.lanes) maxlen = l.size() > maxlen ? l.size() : maxlen;
    for (size_t j = 0; j < maxlen && rc == CHAP_OK; ++j) {
        std::vector<const chap_pending*> items;
        for (const auto& l : G.lanes) if (j < l.size()) items.push_back(&l[j]);
        std::vector<char> used(items.size(), 0);
        for (size_t i = 0; i < items.size() && rc == CHAP_OK; ++i) {
            if (used[i]) continue;
            const chap_pending* grp[CHAP_MAX_GROUP];
            int n = 0;
            grp[n++] = items[i];
            used[i] = 1;
            for (size_t k = i + 1; k < items.size() && n < CHAP_MAX_LANES; ++k)
                if (!used[k] && same_launch(*items[i], *items[k])) { grp[n++] = items[k]; used[k] = 1; }
            rc = items[i]->merged(grp, n, G.stream);
            ++launched;
        }
    }
    G.lanes.clear();
    return rc == CHAP_OK ? launched : rc;
}

// Leaves the recording state WITHOUT issuing anything (the caller failed half-way through a region: its buffers are about to be released).
extern "C" int chap_group_cancel(void) {
    if (!G.on) { chap_set_error("chap_group_cancel: not recording"); return CHAP_EINVAL; }
    G.on = false;
    G.lanes.clear();
    return CHAP_OK;
}

#ifdef CHAP_TIMELINE
// Lab only (see launch.h): bookkeeping of the untraced timeline.  Not declared in include/chap_hip.h -- the product library does not have it.
#include <mutex>
#include <string>
namespace {
struct tl_entry { std::string name; const void* fn; unsigned gx, gy, gz; void* stream; long first; };      // first: index of the launch's first (start, end) pair
std::mutex tl_mu;
std::vector<tl_entry> tl_entries;
unsigned long long* tl_buf = nullptr;
long tl_cap = 0, tl_used = 0;          // in (start, end) pairs
}
unsigned long long* chap_timeline_next_slot(const char* name, const void* fn, dim3 grid, hipStream_t s) {
    std::lock_guard<std::mutex> lk(tl_mu);
    const long nb = 3;                  // first / middle / last block
    if (!tl_buf || tl_used + nb > tl_cap) return nullptr;
    tl_entries.push_back({name ? name : "?", fn, grid.x, grid.y, grid.z, (void*)s, tl_used});
    tl_used += nb;
    return tl_buf + 2 * tl_entries.back().first;
}
__global__ void chap_tl_marker_kernel(unsigned long long* p) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); p[0] = t; p[1] = t; }
void chap_timeline_mark(const char* name, void* stream) {
    unsigned long long* p = nullptr;
    {
        std::lock_guard<std::mutex> lk(tl_mu);
        if (!tl_buf || tl_used + 3 > tl_cap) return;
        tl_entries.push_back({std::string(name ? name : "?") + " (end marker)", nullptr, 1, 1, 1, stream, tl_used});
        tl_used += 3;
        p = tl_buf + 2 * tl_entries.back().first;
    }
    hipLaunchKernelGGL(chap_tl_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p);
    (void)hipGetLastError();
}
extern "C" int chap_timeline_enable(void* buf, long npairs) {      // buf: device memory, 2 * npairs uint64; NULL switches the stamps off
    std::lock_guard<std::mutex> lk(tl_mu);
    tl_buf = (unsigned long long*)buf; tl_cap = npairs; tl_used = 0; tl_entries.clear();
    return CHAP_OK;
}
extern "C" long chap_timeline_count(void) { std::lock_guard<std::mutex> lk(tl_mu); return (long)tl_entries.size(); }
extern "C" int chap_timeline_entry(long i, char* name, int cap, unsigned* grid3, void** stream, const void** fn, long* first) {
    std::lock_guard<std::mutex> lk(tl_mu);
    if (i < 0 || i >= (long)tl_entries.size()) return CHAP_EINVAL;
    const tl_entry& e = tl_entries[i];
    snprintf(name, cap, "%s", e.name.c_str());
    grid3[0] = e.gx; grid3[1] = e.gy; grid3[2] = e.gz; *stream = e.stream; *fn = e.fn; *first = e.first;
    return CHAP_OK;
}
#endif
