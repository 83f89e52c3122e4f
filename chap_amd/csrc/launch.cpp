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
            for (size_t k = i + 1; k < items.size() && n < CHAP_MAX_GROUP; ++k)
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

namespace { thread_local const float* g_stats_ptr = nullptr; thread_local int g_stats_slots = 0; }
void chap_note_stats_slots(const float* stats, int slots) { g_stats_ptr = stats; g_stats_slots = slots; }
int chap_known_stats_slots(const float* stats) {      // one-shot: only the finalize call right behind the conv may use the note
    const int n = (stats != nullptr && stats == g_stats_ptr) ? g_stats_slots : 0;
    g_stats_ptr = nullptr;
    return n;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Capture points (chap_hip.h, chap_capture_*): parallel branches of a captured HIP graph on ONE stream.  A capturing stream has a
// "dependency set": the graph nodes the next captured launch will depend on.  Saving that set (mark), restoring it (goto) and adding a
// saved set back (join) forks and joins branches without a second stream and without events -- which matters on ROCm 7.2, where an
// event dependency between two streams that were both forked from the capture's origin crashes hipStreamEndCapture, so a pass that
// already runs on a forked stream could not fork again.  Outside a capture the three calls do nothing (stream order stands).
namespace {
constexpr int CAP_SLOTS = 16;
struct cap_slot { unsigned long long id = 0; std::vector<hipGraphNode_t> nodes; bool set = false; };
thread_local cap_slot g_cap[CAP_SLOTS];

// 1 = capturing (deps filled), 0 = not capturing, < 0 error
int capture_deps(hipStream_t s, const char* who, unsigned long long* id, std::vector<hipGraphNode_t>* deps) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    hipGraph_t g = nullptr;
    const hipGraphNode_t* d = nullptr;
    size_t n = 0;
    const hipError_t e = hipStreamGetCaptureInfo_v2(s, &st, id, &g, &d, &n);
    if (e != hipSuccess) { (void)hipGetLastError(); chap_set_error("%s: hipStreamGetCaptureInfo_v2 failed: %s", who, hipGetErrorString(e)); return CHAP_ELAUNCH; }
    if (st != hipStreamCaptureStatusActive) return 0;
    if (deps) deps->assign(d, d + n);
    return 1;
}
}  // namespace

extern "C" int chap_capture_mark(void* stream, int slot, int accumulate) {
    CHAP_CHECK_ARG(slot >= 0 && slot < CAP_SLOTS, "chap_capture_mark: slot %d out of range [0, %d)", slot, CAP_SLOTS);
    CHAP_NOT_IN_GROUP("chap_capture_mark");
    unsigned long long id = 0;
    std::vector<hipGraphNode_t> deps;
    const int rc = capture_deps((hipStream_t)stream, "chap_capture_mark", &id, &deps);
    if (rc <= 0) return rc;
    cap_slot& c = g_cap[slot];
    if (!accumulate || !c.set || c.id != id) { c.nodes.clear(); c.id = id; c.set = true; }
    for (hipGraphNode_t nd : deps) {
        bool have = false;
        for (hipGraphNode_t o : c.nodes) have = have || o == nd;
        if (!have) c.nodes.push_back(nd);
    }
    return 1;
}

static int capture_update(void* stream, int slot, unsigned flags, bool clear, const char* who) {
    CHAP_CHECK_ARG(slot >= 0 && slot < CAP_SLOTS, "%s: slot %d out of range [0, %d)", who, slot, CAP_SLOTS);
    CHAP_NOT_IN_GROUP(who);
    unsigned long long id = 0;
    const int rc = capture_deps((hipStream_t)stream, who, &id, nullptr);
    if (rc <= 0) return rc;
    cap_slot& c = g_cap[slot];
    if (!c.set || c.id != id) { chap_set_error("%s: slot %d was not marked in this capture", who, slot); return CHAP_EINVAL; }
    const hipError_t e = hipStreamUpdateCaptureDependencies((hipStream_t)stream, c.nodes.data(), c.nodes.size(), flags);
    if (e != hipSuccess) { (void)hipGetLastError(); chap_set_error("%s: hipStreamUpdateCaptureDependencies failed: %s", who, hipGetErrorString(e)); return CHAP_ELAUNCH; }
    if (clear) { c.nodes.clear(); c.set = false; }
    return 1;
}
extern "C" int chap_capture_goto(void* stream, int slot) { return capture_update(stream, slot, hipStreamSetCaptureDependencies, false, "chap_capture_goto"); }
extern "C" int chap_capture_join(void* stream, int slot) { return capture_update(stream, slot, hipStreamAddCaptureDependencies, true, "chap_capture_join"); }
