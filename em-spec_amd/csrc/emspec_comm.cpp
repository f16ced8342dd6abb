// emspec_comm.cpp — multi-GPU side of the C ABI: RCCL communicator per engine and the gather of finished
// palette-index columns to one rank (BASELINE.json north_star: "many independent audio streams shard embarrassingly
// across the 8 GPUs of one node with a single RCCL gather over xGMI to collect finished columns").
//
// The reference has no GPU path and no collectives, so nothing here has a reference file:line;
// the RCCL entry points used are /opt/rocm/include/rccl/rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220,
// ncclCommDestroy :260, ncclAllGather :678, ncclSend :700, ncclRecv :722, ncclGroupStart/End.
//
// One process (or thread) per GPU, one engine per process; the streams are sharded by the host, every rank runs
// emspec_batch_device on its own shard, and the only exchange is this gather.  xGMI is point-to-point, so each
// rank reaches the root over one link: the columns travel in the lossless wire image of pack.hip.inc
// (per-column payload offset + bit mask + non-zero indices, ~186 B instead of 1024 B per column on the bench input) and are expanded on the
// root.  Sizes differ per rank and RCCL's send/recv counts must match on both sides, so a gather is
//     pack (GPU)  ->  ncclAllGather of the 8-byte image sizes  ->  one host sync to read them
//     ->  grouped ncclSend (ranks) / ncclRecv x (world-1) (root)  ->  expand on the root (GPU)
// all enqueued on the caller's stream.
// Failure semantics (round 4): what can fail on one rank alone is detected before the size exchange and travels through it
// as an error mark, so every rank returns from the same call; the one host wait is bounded (emspec_comm_set_timeout); an RCCL
// error or a missing peer aborts the communicator.  The diagnostic build can replace the three RCCL calls by an in-process
// stand-in (EMSPEC_COMM_MOCK=1, MockGroup below) so that the multi-rank logic runs with 2 .. 8 ranks on one GPU
// (tests/test_gather.py).
#include "emspec_engine.h"

#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#ifdef EMSPEC_DIAG
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#endif
#include <string>
#include <vector>

using namespace emspec;

namespace emspec {
uint64_t* wire_total_ptr(void* scratch, int64_t columns);
}  // namespace emspec

#ifdef EMSPEC_DIAG
// Diagnostic build only (libemspec_diag.so, EMSPEC_COMM_MOCK=1): an in-process stand-in for the RCCL communicator, so that
// everything AROUND the three RCCL calls of a gather - the per-rank packing, the (bytes, columns) exchange with its error
// marks, the root's layout of uneven shards, the directory of the packed form, the expand of several images, the timeout -
// runs with 2 .. 8 "ranks" (threads, one engine each) on ONE GPU.  One GPU per box is all the tests get, and two ranks on
// one device are refused by RCCL itself, so before round 4 none of that logic had ever run with world > 1.
// The stand-in: a generation barrier among the ranks' threads, the size pairs through host memory, and "send/recv" as
// device-to-device copies enqueued by the root (the ranks share the device's address space).
struct MockGroup {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, arrived = 0;
    uint64_t gen = 0;
    bool broken = false;
    std::vector<uint64_t> pairs;           // [world][2] (image bytes or the error mark, columns)
    std::vector<const uint8_t*> wire;      // [world] the ranks' packed images (device pointers)
    bool barrier(double timeout_s) {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const uint64_t g = gen;
        if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); return true; }
        const auto limit = std::chrono::duration<double>(timeout_s > 0 ? timeout_s : 1e9);
        const bool ok = cv.wait_for(lk, limit, [&] { return gen != g || broken; });
        if (!ok || broken) { broken = true; cv.notify_all(); return false; }   // a peer is missing: the group is dead (cf. ncclCommAbort)
        return true;
    }
};
static std::mutex g_mock_mu;
static std::map<std::string, std::shared_ptr<MockGroup>> g_mock_groups;
#endif

struct emspec_comm_state {
    ncclComm_t comm = nullptr;
#ifdef EMSPEC_DIAG
    std::shared_ptr<MockGroup> mock;                        // set instead of comm under EMSPEC_COMM_MOCK=1
#endif
    int rank = 0, world = 1;
    // device workspaces, grown on demand
    uint8_t* d_wire = nullptr; size_t wire_bytes = 0;        // this rank's packed image
    void* d_scratch = nullptr; size_t scratch_bytes = 0;     // pack / unpack scan workspace
    uint8_t* d_recv = nullptr; size_t recv_bytes = 0;        // root: the other ranks' images, back to back
    uint64_t* d_sizes = nullptr;                             // [world][2] (image bytes, columns) after the all-gather
    uint64_t* h_sizes = nullptr;                             // page-locked copy
    // layout of the last EMSPEC_GATHER_PACKED gather on the root: per rank (offset in gathered_dev, image bytes, columns)
    std::vector<uint64_t> packed_layout;
    uint64_t* h_dir = nullptr;                               // page-locked directories (kDirSlots of them, used in rotation: the
    unsigned dir_next = 0;                                   //  copy to the head of gathered_dev is still in flight at return)
    double timeout_s = 120.0;                                // bound on the one host wait of a gather (0: wait for ever)
};
static constexpr unsigned kDirSlots = 8;

namespace {

#define NCCLCHK(e, call)                                                                      \
    do {                                                                                      \
        ncclResult_t _r = (call);                                                             \
        if (_r != ncclSuccess)                                                                \
            return fail((e), EMSPEC_ERR_COMM, std::string(#call) + ": " + ncclGetErrorString(_r)); \
    } while (0)

// A communicator that returned an error, or whose peers did not show up, is in an undefined state: tear it down
// (ncclCommAbort also ends the kernels it still has in flight) so that later calls fail at once with EMSPEC_ERR_STATE
// instead of blocking.
void abort_comm(emspec_comm_state* c) {
    if (c && c->comm) {
        (void)ncclCommAbort(c->comm);
        c->comm = nullptr;
    }
#ifdef EMSPEC_DIAG
    if (c && c->mock) {
        { std::lock_guard<std::mutex> g(c->mock->mu); c->mock->broken = true; }
        c->mock->cv.notify_all();
        c->mock.reset();
    }
#endif
}
bool has_comm(const emspec_comm_state* c) {
#ifdef EMSPEC_DIAG
    if (c && c->mock) return true;
#endif
    return c && c->comm;
}
#define NCCLCHK_ABORT(e, c, call)                                                             \
    do {                                                                                      \
        ncclResult_t _r = (call);                                                             \
        if (_r != ncclSuccess) {                                                              \
            abort_comm(c);                                                                    \
            return fail((e), EMSPEC_ERR_COMM, std::string(#call) + ": " + ncclGetErrorString(_r) + " (communicator aborted)"); \
        }                                                                                     \
    } while (0)

// the rank-local part of a gather that can fail before the size exchange
constexpr uint64_t kRankFailed = ~0ull;   // in the (bytes, columns) pair of the size exchange: "this rank cannot take part"

int ensure_wire_buffers(emspec_engine* e, emspec_comm_state* c, int64_t columns) {
    int rc;
    if ((rc = grow(e, (void**)&c->d_wire, &c->wire_bytes, (size_t)wire_bound_bytes(columns, e->cfg.rows)))) return rc;
    if ((rc = grow(e, &c->d_scratch, &c->scratch_bytes, wire_scratch_bytes(columns)))) return rc;
    return EMSPEC_OK;
}

// engines without a communicator still need the pack workspaces (emspec_wire_pack / _unpack)
emspec_comm_state* state(emspec_engine* e) {
    if (!e->comm) e->comm = new (std::nothrow) emspec_comm_state();
    return e->comm;
}

}  // namespace

namespace emspec {
bool comm_shares_device(const emspec_engine* e) { return has_comm(e->comm) && e->comm->world > 1; }
void comm_destroy(emspec_engine* e) {
    emspec_comm_state* c = e->comm;
    if (!c) return;
    if (c->comm) (void)ncclCommDestroy(c->comm);
    (void)hipFree(c->d_wire); (void)hipFree(c->d_scratch); (void)hipFree(c->d_recv); (void)hipFree(c->d_sizes);
    if (c->h_sizes) (void)hipHostFree(c->h_sizes);
    if (c->h_dir) (void)hipHostFree(c->h_dir);
    delete c;
    e->comm = nullptr;
}
}  // namespace emspec

extern "C" {

int emspec_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return EMSPEC_ERR_INVALID_ARG;
    static_assert(EMSPEC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return fail(nullptr, EMSPEC_ERR_COMM, "ncclGetUniqueId failed");
    std::memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return EMSPEC_OK;
}

int emspec_comm_init(emspec_engine* e, const uint8_t* id_bytes, int32_t rank, int32_t world) {
    if (!e || !id_bytes) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(e, EMSPEC_ERR_INVALID_ARG, "rank must be in [0, world)");
    emspec_comm_state* c = state(e);
    if (!c) return fail(e, EMSPEC_ERR_OUT_OF_MEMORY, "out of host memory");
    if (has_comm(c)) return fail(e, EMSPEC_ERR_STATE, "this engine already has a communicator");
    HIPCHK(e, hipSetDevice(e->device));
    bool mocked = false;
#ifdef EMSPEC_DIAG
    if (const char* ev = getenv("EMSPEC_COMM_MOCK")) mocked = ev[0] == '1';
    if (mocked) {   // join (or create) the in-process group of this id
        std::lock_guard<std::mutex> g(g_mock_mu);
        auto& grp = g_mock_groups[std::string(reinterpret_cast<const char*>(id_bytes), NCCL_UNIQUE_ID_BYTES)];
        if (!grp) { grp = std::make_shared<MockGroup>(); grp->world = world; grp->pairs.assign(2 * (size_t)world, 0); grp->wire.assign((size_t)world, nullptr); }
        if (grp->world != world) return fail(e, EMSPEC_ERR_INVALID_ARG, "mock communicator: the ranks disagree about the world size");
        c->mock = grp;
    }
#endif
    if (!mocked) {
        ncclUniqueId id;
        std::memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
        NCCLCHK(e, ncclCommInitRank(&c->comm, world, id, rank));
    }
    c->rank = rank;
    c->world = world;
    HIPCHK(e, hipMalloc(&c->d_sizes, sizeof(uint64_t) * 2 * (size_t)(world + 1)));
    HIPCHK(e, hipHostMalloc((void**)&c->h_sizes, sizeof(uint64_t) * 2 * (size_t)(world + 1), hipHostMallocDefault));
    HIPCHK(e, hipHostMalloc((void**)&c->h_dir, sizeof(uint64_t) * 4 * (size_t)world * kDirSlots + 64, hipHostMallocDefault));
    return EMSPEC_OK;
}

int emspec_comm_destroy(emspec_engine* e) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    (void)hipSetDevice(e->device);
    comm_destroy(e);
    return EMSPEC_OK;
}

int emspec_comm_set_timeout(emspec_engine* e, double seconds) {
    if (!e || !(seconds >= 0.0)) return fail(e, EMSPEC_ERR_INVALID_ARG, "seconds must be >= 0");
    emspec_comm_state* c = state(e);
    if (!c) return fail(e, EMSPEC_ERR_OUT_OF_MEMORY, "out of host memory");
    c->timeout_s = seconds;
    return EMSPEC_OK;
}

int32_t emspec_comm_rank(const emspec_engine* e) { return e && has_comm(e->comm) ? e->comm->rank : -1; }
int32_t emspec_comm_world(const emspec_engine* e) { return e && has_comm(e->comm) ? e->comm->world : 0; }

int64_t emspec_wire_bound(int64_t columns, int32_t rows) {
    if (columns < 0 || rows < 4 || rows % 4) return -1;
    return wire_bound_bytes(columns, rows);
}

int emspec_wire_pack(emspec_engine* e, const uint8_t* index_dev, int64_t columns, uint8_t* wire_dev, int64_t* wire_bytes,
                     void* hip_stream) {
    if (!e || !index_dev || !wire_dev || columns < 1) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument / no columns");
    if ((uint64_t)columns * (uint64_t)e->cfg.rows >= (1ull << 32)) return fail(e, EMSPEC_ERR_INVALID_ARG, "at most 2^32 cells per call");
    emspec_comm_state* c = state(e);
    if (!c) return fail(e, EMSPEC_ERR_OUT_OF_MEMORY, "out of host memory");
    HIPCHK(e, hipSetDevice(e->device));
    int rc;
    if ((rc = grow(e, &c->d_scratch, &c->scratch_bytes, wire_scratch_bytes(columns)))) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    HIPCHK(e, launch_wire_pack(index_dev, columns, e->cfg.rows, wire_dev, c->d_scratch, st));
    if (wire_bytes) {   // the size is produced on the device: reading it is the one synchronisation of this call
        uint64_t total = 0;
        HIPCHK(e, hipMemcpyAsync(&total, wire_total_ptr(c->d_scratch, columns), sizeof(total), hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipStreamSynchronize(st));
        *wire_bytes = (int64_t)total;
    }
    return EMSPEC_OK;
}

int emspec_wire_unpack(emspec_engine* e, const uint8_t* wire_dev, int64_t wire_bytes, int64_t columns, uint8_t* index_dev,
                       void* hip_stream) {
    if (!e || !index_dev || !wire_dev || columns < 1) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument / no columns");
    if ((uint64_t)columns * (uint64_t)e->cfg.rows >= (1ull << 32)) return fail(e, EMSPEC_ERR_INVALID_ARG, "at most 2^32 cells per call");
    emspec_comm_state* c = state(e);
    if (!c) return fail(e, EMSPEC_ERR_OUT_OF_MEMORY, "out of host memory");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)hip_stream;
    // validate the header before trusting the image (an image from another configuration would index out of range)
    uint32_t h[8];
    if (wire_bytes < 32) return fail(e, EMSPEC_ERR_INVALID_ARG, "wire image shorter than its header");
    HIPCHK(e, hipMemcpyAsync(h, wire_dev, sizeof(h), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    const uint64_t hcols = (uint64_t)h[2] | ((uint64_t)h[3] << 32), hpay = (uint64_t)h[4] | ((uint64_t)h[5] << 32);
    const int64_t need = wire_fixed_bytes(columns, e->cfg.rows) + (int64_t)((hpay + 15) & ~(uint64_t)15);
    if (h[0] != 0x32574D45u /* "EMW2" */ || (int32_t)h[1] != e->cfg.rows || hcols != (uint64_t)columns || hpay > (uint64_t)columns * e->cfg.rows ||
        wire_bytes < need)
        return fail(e, EMSPEC_ERR_INVALID_ARG, "wire image does not match this engine's rows / the column count");
    HIPCHK(e, launch_wire_unpack(wire_dev, columns, e->cfg.rows, index_dev, st));
    return EMSPEC_OK;
}

int emspec_gather_columns(emspec_engine* e, const uint8_t* index_dev, int64_t columns, int32_t root, uint8_t* gathered_dev,
                          int64_t gathered_capacity, uint32_t flags, void* hip_stream, int64_t* wire_bytes_sent) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    emspec_comm_state* c = e->comm;
    if (!has_comm(c)) return fail(e, EMSPEC_ERR_STATE, "no communicator: call emspec_comm_init first (or it was aborted after an error)");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int R = e->cfg.rows, world = c->world, me = c->rank;
    const bool is_root = me == root;
    const bool loopback = (flags & EMSPEC_GATHER_LOOPBACK) != 0;   // the root's own columns take the wire too (tests)
    const bool packed = (flags & EMSPEC_GATHER_PACKED) != 0;       // the root keeps the images packed (no expand)
    const size_t col_bytes = (size_t)columns * R;
    const bool i_send = !is_root || loopback;
    const bool i_pack = i_send || packed;                          // packed: the root's own columns become an image too
    if (wire_bytes_sent) *wire_bytes_sent = 0;

    // ---- everything that can fail on THIS rank alone happens before the size exchange, and a failure does not return:
    // the rank still enters the all-gather, with kRankFailed in place of its image size, so that every rank learns of it
    // in the same collective and all of them return together (a rank that simply returned would leave its peers blocked
    // in the all-gather for ever).
    int local_rc = EMSPEC_OK;
    std::string local_msg;
    auto local_fail = [&](int code, const std::string& msg) { if (local_rc == EMSPEC_OK) { local_rc = code; local_msg = msg; } };
    if (!index_dev || columns < 1) local_fail(EMSPEC_ERR_INVALID_ARG, "null argument / no columns");
    else if (root < 0 || root >= world) local_fail(EMSPEC_ERR_INVALID_ARG, "root out of range");
    else if (is_root && !gathered_dev) local_fail(EMSPEC_ERR_INVALID_ARG, "the root needs the gathered buffer");
    else if ((uint64_t)columns * (uint64_t)R >= (1ull << 32)) local_fail(EMSPEC_ERR_INVALID_ARG, "at most 2^32 cells per call");
#ifdef EMSPEC_DIAG
    if (const char* ev = getenv("EMSPEC_GATHER_FAIL_RANK"))        // test hook: this rank "runs out of memory" before the exchange
        if (atoi(ev) == me) local_fail(EMSPEC_ERR_OUT_OF_MEMORY, "injected failure before the size exchange (EMSPEC_GATHER_FAIL_RANK)");
#endif
    // ---- this rank's wire image
    uint64_t* d_total = c->d_sizes + 2 * (size_t)world;            // spare pair behind the gathered sizes: used when nothing was packed
    if (local_rc == EMSPEC_OK) {
        int rc = i_pack ? ensure_wire_buffers(e, c, columns) : grow(e, &c->d_scratch, &c->scratch_bytes, wire_scratch_bytes(columns));
        if (rc != EMSPEC_OK) local_fail(rc, e->err);
    }
    if (local_rc == EMSPEC_OK && i_pack) {
        const hipError_t he = launch_wire_pack(index_dev, columns, R, c->d_wire, c->d_scratch, st);
        if (he != hipSuccess) local_fail(EMSPEC_ERR_HIP, std::string("wire pack launch: ") + hipGetErrorString(he));
    }
    if (local_rc == EMSPEC_OK) {
        d_total = wire_total_ptr(c->d_scratch, columns);
        if (!i_pack) HIPCHK(e, hipMemsetAsync(d_total, 0, sizeof(uint64_t), st));   // the root sends nothing
        c->h_sizes[2 * world] = (uint64_t)columns;                 // page-locked: read by the async copy below
        HIPCHK(e, hipMemcpyAsync(d_total + 1, &c->h_sizes[2 * world], sizeof(uint64_t), hipMemcpyHostToDevice, st));
    } else {
        c->h_sizes[2 * world] = kRankFailed;
        c->h_sizes[2 * world + 1] = 0;
        HIPCHK(e, hipMemcpyAsync(d_total, &c->h_sizes[2 * world], 2 * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    }
    // ---- everybody learns everybody's image size and column count (RCCL counts must match on both sides of a
    // send/recv; shards may differ in size: a host that gives the root fewer streams balances its extra expand work)
#ifdef EMSPEC_DIAG
    if (c->mock) {   // in-process stand-in: this rank's pair to the host, then through the group
        uint64_t mine[2];
        HIPCHK(e, hipMemcpyAsync(mine, d_total, sizeof(mine), hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipStreamSynchronize(st));
        auto grp = c->mock;
        { std::lock_guard<std::mutex> g(grp->mu); grp->pairs[2 * me] = mine[0]; grp->pairs[2 * me + 1] = mine[1]; grp->wire[me] = c->d_wire; }
        if (!grp->barrier(c->timeout_s)) {
            abort_comm(c);
            return fail(e, EMSPEC_ERR_COMM, "the size exchange of the gather did not complete within the communicator's timeout "
                                            "(a peer rank is missing); communicator aborted");
        }
        { std::lock_guard<std::mutex> g(grp->mu); for (int r = 0; r < 2 * world; ++r) c->h_sizes[r] = grp->pairs[r]; }
    } else
#endif
    {
    NCCLCHK_ABORT(e, c, ncclAllGather(d_total, c->d_sizes, 2, ncclUint64, c->comm, st));
    HIPCHK(e, hipMemcpyAsync(c->h_sizes, c->d_sizes, sizeof(uint64_t) * 2 * (size_t)world, hipMemcpyDeviceToHost, st));
    {   // the one host wait of the call, bounded: a peer that never enters the collective must not hang this rank for ever
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t q;
        while ((q = hipStreamQuery(st)) == hipErrorNotReady) {
            if (c->timeout_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
                abort_comm(c);
                return fail(e, EMSPEC_ERR_COMM, "the size exchange of the gather did not complete within the communicator's timeout "
                                                "(a peer rank is missing); communicator aborted");
            }
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        HIPCHK(e, q);
    }
    }
    if (local_rc != EMSPEC_OK) return fail(e, local_rc, local_msg);          // (the peers return EMSPEC_ERR_COMM below)
    for (int r = 0; r < world; ++r)
        if (c->h_sizes[2 * r] == kRankFailed)
            return fail(e, EMSPEC_ERR_COMM, "rank " + std::to_string(r) + " failed before the exchange: no columns were transferred");
    int rc;
    for (int r = 0; r < world; ++r) {
        const uint64_t bytes_r = c->h_sizes[2 * r], cols_r = c->h_sizes[2 * r + 1];
        if (cols_r < 1 || cols_r * (uint64_t)R >= (1ull << 32) || bytes_r > (uint64_t)wire_bound_bytes((int64_t)cols_r, R))
            return fail(e, EMSPEC_ERR_COMM, "a rank announced an impossible wire image (column count / size)");
    }
    if (wire_bytes_sent) *wire_bytes_sent = (int64_t)c->h_sizes[2 * me];

    // ---- the exchange: one grouped set of point-to-point transfers, every rank -> root
    std::vector<size_t> off((size_t)world + 1, 0), dst_off((size_t)world + 1, 0);
    const size_t dir_bytes = packed ? ((sizeof(uint64_t) * 4 * (size_t)world + 255) & ~(size_t)255) : 0;
    if (is_root) {
        for (int r = 0; r < world; ++r) {
            off[r + 1] = off[r] + (((size_t)c->h_sizes[2 * r] + 255) & ~(size_t)255);
            dst_off[r + 1] = dst_off[r] + (size_t)c->h_sizes[2 * r + 1] * R;
        }
        if (!packed && (rc = grow(e, (void**)&c->d_recv, &c->recv_bytes, off[world] + 256))) return rc;
    }
    // a gathered buffer that cannot hold the announced shards is the root's error alone: the transfers below still run
    // (into the root's own receive buffer), so that the other ranks' sends complete, and only the expand is skipped
    const size_t need_cap = packed ? dir_bytes + off[world] : dst_off[world];
    const bool fits = !is_root || need_cap <= (size_t)(gathered_capacity > 0 ? gathered_capacity : 0);
    uint8_t* recv_base = c->d_recv;
    if (is_root && packed) {
        if (fits) recv_base = gathered_dev + dir_bytes;            // the images land where they stay
        else if ((rc = grow(e, (void**)&c->d_recv, &c->recv_bytes, off[world] + 256))) return rc; else recv_base = c->d_recv;
    }
#ifdef EMSPEC_DIAG
    if (c->mock) {   // "send/recv": the root copies every sender's image device-to-device, then everybody meets again
        auto grp = c->mock;
        hipError_t he = hipSuccess;
        if (is_root) {
            for (int r = 0; r < world && he == hipSuccess; ++r)
                if (c->h_sizes[2 * r] > 0 && (r != me || loopback)) {
                    const uint8_t* src;
                    { std::lock_guard<std::mutex> g(grp->mu); src = grp->wire[r]; }
                    he = hipMemcpyAsync(recv_base + off[r], src, (size_t)c->h_sizes[2 * r], hipMemcpyDeviceToDevice, st);
                }
            if (he == hipSuccess) he = hipStreamSynchronize(st);      // the senders may reuse their images after the barrier below
        }
        const bool met = grp->barrier(c->timeout_s);
        if (he != hipSuccess || !met) {
            abort_comm(c);
            return fail(e, EMSPEC_ERR_COMM, he != hipSuccess ? std::string("mock transfer: ") + hipGetErrorString(he)
                                                             : std::string("a peer rank left the gather; communicator aborted"));
        }
    } else
#endif
    {
    NCCLCHK_ABORT(e, c, ncclGroupStart());
    ncclResult_t nr = ncclSuccess;
    if (i_send && c->h_sizes[2 * me] > 0) nr = ncclSend(c->d_wire, (size_t)c->h_sizes[2 * me], ncclUint8, root, c->comm, st);
    if (is_root)
        for (int r = 0; r < world && nr == ncclSuccess; ++r)
            if (c->h_sizes[2 * r] > 0 && (r != me || loopback))
                nr = ncclRecv(recv_base + off[r], (size_t)c->h_sizes[2 * r], ncclUint8, r, c->comm, st);
    const ncclResult_t ge = ncclGroupEnd();
    if (nr != ncclSuccess || ge != ncclSuccess) {
        abort_comm(c);
        return fail(e, EMSPEC_ERR_COMM, std::string("ncclSend/ncclRecv: ") + ncclGetErrorString(nr != ncclSuccess ? nr : ge) + " (communicator aborted)");
    }
    }

    if (!fits) return fail(e, EMSPEC_ERR_INVALID_ARG, "the gathered buffer is smaller than the shards the ranks announced");

    // ---- root: expand every image into its rank's block of the gathered buffer (blocks in rank order, each as long as
    // that rank's shard); its own columns are a device copy
    if (is_root && packed) {
        // directory (per rank: offset from the start of gathered_dev, image bytes, columns, 0) + the images, 256-byte
        // aligned, in rank order; the root's own image is a device copy of what it packed above.  Expand any of them
        // later with emspec_wire_unpack(gathered_dev + offset, bytes, columns, ...): emspec_gather_packed_layout.
        c->packed_layout.assign((size_t)world * 3, 0);
        uint64_t* dir = c->h_dir + (size_t)(c->dir_next++ % kDirSlots) * 4 * (size_t)world;
        for (int r = 0; r < world; ++r) {
            dir[4 * r] = c->packed_layout[3 * r] = (uint64_t)(dir_bytes + off[r]);
            dir[4 * r + 1] = c->packed_layout[3 * r + 1] = c->h_sizes[2 * r];
            dir[4 * r + 2] = c->packed_layout[3 * r + 2] = c->h_sizes[2 * r + 1];
            dir[4 * r + 3] = 0;
        }
        HIPCHK(e, hipMemcpyAsync(gathered_dev, dir, sizeof(uint64_t) * 4 * (size_t)world, hipMemcpyHostToDevice, st));
        if (!loopback)
            HIPCHK(e, hipMemcpyAsync(gathered_dev + dir_bytes + off[me], c->d_wire, (size_t)c->h_sizes[2 * me], hipMemcpyDeviceToDevice, st));
        return EMSPEC_OK;
    }
    if (is_root) {
        for (int r = 0; r < world; ++r) {
            uint8_t* dst = gathered_dev + dst_off[r];
            if (r == me && !loopback) {
                if (dst != index_dev) HIPCHK(e, hipMemcpyAsync(dst, index_dev, col_bytes, hipMemcpyDeviceToDevice, st));
                continue;
            }
            HIPCHK(e, launch_wire_unpack(c->d_recv + off[r], (int64_t)c->h_sizes[2 * r + 1], R, dst, st));
        }
    }
    return EMSPEC_OK;
}

int emspec_gather_packed_layout(const emspec_engine* e, int32_t rank, int64_t* offset, int64_t* bytes, int64_t* columns) {
    if (!e || !has_comm(e->comm)) return fail(e, EMSPEC_ERR_STATE, "no communicator");
    const emspec_comm_state* c = e->comm;
    if (rank < 0 || rank >= c->world || c->packed_layout.size() != (size_t)c->world * 3)
        return fail(e, EMSPEC_ERR_STATE, "no EMSPEC_GATHER_PACKED gather has completed on this engine as the root");
    if (offset) *offset = (int64_t)c->packed_layout[3 * rank];
    if (bytes) *bytes = (int64_t)c->packed_layout[3 * rank + 1];
    if (columns) *columns = (int64_t)c->packed_layout[3 * rank + 2];
    return EMSPEC_OK;
}

int emspec_batch_gather(emspec_engine* e, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop, int32_t reassign,
                        int32_t root, uint8_t* gathered_index, float* db_local, int64_t* wire_bytes_sent) {
    if (!e || !pcm) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    emspec_comm_state* c = e->comm;
    if (!has_comm(c)) return fail(e, EMSPEC_ERR_STATE, "no communicator: call emspec_comm_init first");
    if (root < 0 || root >= c->world) return fail(e, EMSPEC_ERR_INVALID_ARG, "root out of range");
    if (c->rank == root && !gathered_index) return fail(e, EMSPEC_ERR_INVALID_ARG, "the root needs the gathered buffer");
    const int64_t C = emspec_num_columns(L, n, hop);
    if (S < 1 || C < 1) return fail(e, EMSPEC_ERR_INVALID_ARG, "need at least one stream of at least fft-size samples");
    HIPCHK(e, hipSetDevice(e->device));
    const size_t cells = (size_t)S * (size_t)C * (size_t)e->cfg.rows;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const bool is_root = c->rank == root;
    const size_t need = al((size_t)S * L * 4) + al(cells) + (db_local ? al(cells * 4) : 0) + (is_root ? al(cells * c->world) : 0) + 256;
    int rc;
    if ((rc = grow(e, (void**)&e->d_stage, &e->stage_bytes, need))) return rc;
    char* base = e->d_stage;
    float* d_pcm = (float*)base; base += al((size_t)S * L * 4);
    uint8_t* d_idx = (uint8_t*)base; base += al(cells);
    float* d_db = nullptr;
    if (db_local) { d_db = (float*)base; base += al(cells * 4); }
    uint8_t* d_all = is_root ? (uint8_t*)base : nullptr;
    HIPCHK(e, hipMemcpyAsync(d_pcm, pcm, (size_t)S * L * 4, hipMemcpyHostToDevice, e->stream));
    if ((rc = emspec_batch_device(e, d_pcm, S, L, n, hop, reassign, d_db, nullptr, d_idx, e->stream))) return rc;
    if ((rc = emspec_gather_columns(e, d_idx, (int64_t)S * C, root, d_all, (int64_t)(cells * c->world), 0, e->stream, wire_bytes_sent))) return rc;
    if (db_local) HIPCHK(e, hipMemcpyAsync(db_local, d_db, cells * 4, hipMemcpyDeviceToHost, e->stream));
    if (is_root) HIPCHK(e, hipMemcpyAsync(gathered_index, d_all, cells * c->world, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return EMSPEC_OK;
}

}  // extern "C"
