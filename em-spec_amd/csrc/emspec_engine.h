// emspec_engine.h — internal: the engine object behind the C ABI (include/emspec.h), shared by
// emspec_api.cpp (engine, batch, streaming) and emspec_comm.cpp (RCCL gather of finished columns).
#pragma once
// the library is built with -fvisibility=hidden: only the C ABI of include/emspec.h is exported
#pragma GCC visibility push(default)
#include "../../include/emspec.h"
#pragma GCC visibility pop
#include "emspec_launch.h"

#include <map>
#include <string>
#include <vector>

struct emspec_engine;
namespace emspec {
struct Plan {
    int n = 0;
    float2* d_tw = nullptr;
    float* d_ebin = nullptr;
    std::vector<float> h_tw, h_ebin;
    // EXACT mode (cfg.mode == EMSPEC_MODE_EXACT): the binary64 tables
    double2* d_tw64 = nullptr;
    double* d_ebin64 = nullptr;
    double h_e0 = 0.0, h_eR = 0.0;   // ends of the binary64 edge table
};
// Live multi-stream streaming session (emspec_live.cpp; include/emspec.h: emspec_columns / emspec_push_samples_multi):
// S streams, each with its own sample position, sample ring and pending-column ring, advanced together by ONE launch per call.
struct LiveState {
    int S = 0, n = 0, hop = 0, reassign = -1, D = 0;
    int form = 0;             // 0 none, 1 per-frame (emspec_columns), 2 per-sample-block (emspec_push_samples_multi)
    int slots = 0;            // column-ring slots per stream: 2 D + mmax
    int mmax = 0;             // frames per stream and launch, at most
    int64_t cap = 0;          // samples per stream the staging block holds (form 2: mmax * hop)
    int ring_mask = 0;        // device sample ring per stream: ring_mask + 1 >= n + cap samples (form 2)
    std::vector<int64_t> fed, emitted, seen, newbase;   // per stream: frames fed, columns emitted, samples received, samples in the device ring
    std::vector<int> pend;    // per stream: samples waiting in the staging block
    void* d_cells = nullptr; size_t cells_bytes = 0;    // [S][slots][rows] float32 (FAST) / u64 (EXACT)
    float* d_sring = nullptr; size_t sring_bytes = 0;   // [S][ring_mask + 1]
    unsigned* d_done = nullptr; size_t done_bytes = 0;  // [S] arrival counters
    float* d_raw = nullptr; size_t raw_bytes = 0;       // display post-process: raw dB [S][mmax][rows]
    float* d_pstate = nullptr; size_t pstate_bytes = 0; // display post-process: [S][rows + 4] (AGC level, initialised, -, -, previous column)
    // page-locked, device-visible host buffers: descriptors [S], staging samples [S][cap], staging outputs [S][mmax][rows]
    void* h_desc = nullptr; size_t desc_bytes = 0;
    float* h_fresh = nullptr; size_t fresh_bytes = 0;
    float* h_odb = nullptr; size_t odb_bytes = 0;
    uint8_t* h_orgba = nullptr; size_t orgba_bytes = 0;
    unsigned long long* stamps = nullptr;   // diagnostic build: [S][8] page-locked, set by emspec_debug_live_stamps
};

// emspec_api.cpp: the plan cache and the per-shape constants handed to the kernels
int latency(int n, int hop, int reassign);
int check_shape(const emspec_engine* e, int n, int hop);
int get_plan(emspec_engine* e, int n, Plan** out);
PlanDev plan_dev(const emspec_engine* e, const Plan& p, int hop, int reassign);
ExactPlanDev exact_plan_dev(const emspec_engine* e, const Plan& p, int hop, int reassign);
ExactDbMap exact_db_map(const emspec_engine* e, int n, const ExactPlanDev& pd);
DbMap db_map(const emspec_engine* e, int n);
bool host_pinned(const void* p);   // p is null or page-locked host memory the device can address
void live_destroy(emspec_engine* e);   // emspec_live.cpp: called by emspec_destroy
void live_reset(emspec_engine* e);     // drops the live session's stream state (emspec_reset); buffers are kept
bool live_pending(const emspec_engine* e);   // some stream of either session has fed frames whose columns were not emitted yet
}  // namespace emspec

struct emspec_engine {
    emspec_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    // host-buffer batch pipeline (emspec_batch / emspec_batch_packed): H2D and D2H copy streams beside the compute stream,
    // [in | computed | out] events per staging set, the packed images' headers in pinned host memory
    hipStream_t stream_in = nullptr, stream_out = nullptr;
    hipEvent_t pipe_ev[9] = {};
    uint8_t* h_hdr = nullptr; size_t h_hdr_bytes = 0;
    void* d_packscratch = nullptr; size_t packscratch_bytes = 0;
    std::string arch;
    mutable std::string err;
    std::map<int, emspec::Plan> plans;
    std::vector<float> custom_edges_hz;   // rows+1 entries when emspec_set_row_edges_hz was called
    uint8_t* d_lut = nullptr;
    // batch workspace (generic path per-bin records; host-API staging)
    float* d_hist = nullptr;
    size_t hist_bytes = 0;
    bool exact() const { return cfg.mode == EMSPEC_MODE_EXACT; }
    // EXACT fused kernel (exact_fused_lr.hip.inc): the ring's low rows, [workgroup][slots][rl] u64; launches that use it
    // are serialised across HIP streams through xlow_event
    unsigned long long* d_xlow = nullptr;
    size_t xlow_bytes = 0;
    hipEvent_t xlow_event = nullptr;
    bool xlow_used = false;
    char* d_stage = nullptr;
    size_t stage_bytes = 0;
    // display post-process (emspec_set_display)
    float smoothing = 0.0f, agc = 0.0f;
    float* d_raw = nullptr; size_t raw_bytes = 0;      // raw dB columns of a batch
    float* d_post = nullptr; size_t post_bytes = 0;    // post-processed dB when the caller wants none
    float* d_peak = nullptr; size_t peak_bytes = 0;    // column peaks + gains
    // streaming (emspec_live.cpp): the live multi-stream session, and the single-stream calls' own (emspec_column,
    // emspec_push_samples: the same machinery with one stream); independent of each other
    emspec::LiveState live, one;
    // multi-GPU gather of finished columns (emspec_comm.cpp); opaque here so this header needs no rccl.h
    struct emspec_comm_state* comm = nullptr;
};

namespace emspec {
// records the message on the engine (or, for e == nullptr, as the thread's emspec_create error) and returns code
int fail(const emspec_engine* e, int code, const std::string& msg);
// (re)allocates *ptr to at least `want` bytes of device memory
int grow(emspec_engine* e, void** ptr, size_t* have, size_t want);
void comm_destroy(emspec_engine* e);   // emspec_comm.cpp: called by emspec_destroy
bool comm_shares_device(const emspec_engine* e);   // the engine has a communicator with other ranks (world > 1)
}  // namespace emspec

#define HIPCHK(e, call)                                                                          \
    do {                                                                                         \
        hipError_t _r = (call);                                                                  \
        if (_r != hipSuccess)                                                                    \
            return emspec::fail((e), _r == hipErrorOutOfMemory ? EMSPEC_ERR_OUT_OF_MEMORY : EMSPEC_ERR_HIP, \
                                std::string(#call) + ": " + hipGetErrorString(_r));              \
    } while (0)
