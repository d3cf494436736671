// Multi-GPU exchange behind the C ABI: one process per GPU, RCCL over xGMI on the planet's own stream.
//
// The reference is a single-threaded browser worker (js/planet-worker.js): it has no exchange to replace.  These entry points
// are what a multi-GPU host — the JS worker pool of INTEGRATION.md or bench.py — needs so that the data path never leaves the
// devices: the communicator is created from a 128-byte id that rank 0 obtains and the HOST distributes (any channel: a
// message to the worker threads, torch.distributed, a file), and the two exchange shapes are
//   * wo_planet_exchange_allgather: every rank contributes the values of its send list (wo_planet_set_halo), every rank
//     receives all the others' — the merge of the landmass decomposition (decomposed.py), ncclAllGather;
//   * wo_planet_exchange_neighbors: the send list goes in two parts to the previous / next rank of a chain and the receive
//     list comes from them — the one-ring halo of a band decomposition (banded.py), ncclSend / ncclRecv in one group.
// Pack and unpack are the kernels of wo_planet_pack_halo / unpack_halo; everything is enqueued on the planet's stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the library itself is bound at the first multi-GPU call (below)

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/worogen.h"
#include "device.h"

struct wo_comm {
    ncclComm_t comm = nullptr;
    int32_t nranks = 0, rank = 0, device = 0;
    float* gather = nullptr; size_t gatherCap = 0;      // [nranks][maxPerRank] landing buffer of the all-gather
    float* send = nullptr; size_t sendCap = 0;
};

namespace {
__global__ __launch_bounds__(wo::WO_BLOCK) void k_comm_pack(const float* __restrict__ e, const int32_t* __restrict__ idx, int32_t n, float* __restrict__ out) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = e[idx[i]];
}
__global__ __launch_bounds__(wo::WO_BLOCK) void k_comm_unpack(float* __restrict__ e, const int32_t* __restrict__ idx, int32_t n, const float* __restrict__ in) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) e[idx[i]] = in[i];
}

// librccl is NOT a link-time dependency of libworogen.so: a single-GPU host never maps it (hundreds of MB of kernels), and the
// library loads on a machine without it.  The first entry point that needs a collective binds these by name.
struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    std::string error;
};
const Rccl& rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) { R.error = std::string("librccl could not be loaded: ") + dlerror(); return; }
        auto bind = [&](auto& fp, const char* sym) { fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(dlsym(h, sym)); if (!fp && R.error.empty()) R.error = std::string("librccl lacks ") + sym; };
        bind(R.GetUniqueId, "ncclGetUniqueId"); bind(R.CommInitRank, "ncclCommInitRank"); bind(R.CommDestroy, "ncclCommDestroy"); bind(R.CommAbort, "ncclCommAbort");
        bind(R.GetErrorString, "ncclGetErrorString"); bind(R.AllReduce, "ncclAllReduce"); bind(R.AllGather, "ncclAllGather"); bind(R.Broadcast, "ncclBroadcast");
        bind(R.Send, "ncclSend"); bind(R.Recv, "ncclRecv"); bind(R.GroupStart, "ncclGroupStart"); bind(R.GroupEnd, "ncclGroupEnd");
    });
    return R;
}
bool rccl_ready(const char* fn) {
    const Rccl& R = rccl();
    if (R.error.empty()) return true;
    wo::set_error(std::string(fn) + ": " + R.error);
    return false;
}
bool nccl_ok(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return true;
    wo::set_error(std::string(what) + ": " + rccl().GetErrorString(r));
    return false;
}
// every exchange entry point: the planet's device current, and the communicator living on that device
bool comm_ready(wo_planet* p, wo_comm* c, const char* fn) {
    if (!p || !c || !p->ctx) { wo::set_error(std::string(fn) + ": bad arguments"); return false; }
    if (!c->comm) { wo::set_error(std::string(fn) + ": the communicator is closed (an earlier collective failed and aborted it)"); return false; }
    if (c->device != p->ctx->device) { wo::set_error(std::string(fn) + ": the communicator was created on another device than the planet's"); return false; }
    const hipError_t e = hipSetDevice(p->ctx->device);
    if (e != hipSuccess) { wo::set_error(std::string(fn) + ": hipSetDevice failed: " + hipGetErrorString(e)); return false; }
    return true;
}
// ncclGroupStart is always answered by ncclGroupEnd, whatever happens in between
struct GroupGuard {
    bool open = false;
    bool start() { open = nccl_ok(rccl().GroupStart(), "ncclGroupStart"); return open; }
    bool end() { if (!open) return true; open = false; return nccl_ok(rccl().GroupEnd(), "ncclGroupEnd"); }
    ~GroupGuard() { if (open) (void)rccl().GroupEnd(); }
};

// RCCL form of the landmass decomposition's flood exchange (wo_planet_set_flood_exchange_comm)
struct FloodLink {
    wo_planet* planet = nullptr; wo_comm* comm = nullptr;
    std::vector<int32_t> counts, cells;              // cells: every rank's land cells, concatenated in rank order
    std::vector<int64_t> start;
    int32_t maxCount = 1;
    int32_t* d_flag = nullptr; int32_t* h_flag = nullptr;
    float *d_send = nullptr, *d_all = nullptr, *h_send = nullptr, *h_all = nullptr;
    float* d_land = nullptr; size_t landCap = 0;     // the flooded land heights on their way through ncclBroadcast
};
void flood_link_free(void* v) {
    FloodLink* k = (FloodLink*)v;
    if (!k) return;
    if (k->d_flag) (void)hipFree(k->d_flag);
    if (k->h_flag) (void)hipHostFree(k->h_flag);
    if (k->d_send) (void)hipFree(k->d_send);
    if (k->d_all) (void)hipFree(k->d_all);
    if (k->d_land) (void)hipFree(k->d_land);
    if (k->h_send) (void)hipHostFree(k->h_send);
    if (k->h_all) (void)hipHostFree(k->h_all);
    delete k;
}
int flood_link_exchange(void* user, int32_t phase, void* buf, int64_t n) {
    FloodLink* k = (FloodLink*)user;
    if (phase == -1) {                                   // handshake (include/worogen.h): this link implements protocol 2 (phases 0-3)
        int32_t* proto = (int32_t*)buf;
        if (*proto != WO_FLOOD_EXCHANGE_PROTOCOL) return 1;
        *proto = -*proto;
        return 0;
    }
    if (phase < 0 || phase > 3) { wo::set_error("flood exchange: unknown phase"); return 1; }
    wo_comm* c = k->comm;
    if (!c || !c->comm) { wo::set_error("flood exchange: the communicator is closed"); return 1; }
    hipStream_t s = k->planet->ctx->stream;
    try {
        if (hipSetDevice(k->planet->ctx->device) != hipSuccess) { wo::set_error("flood exchange: hipSetDevice failed"); return 1; }
        if (phase == 0) {
            *k->h_flag = *(int32_t*)buf;
            WO_HIP(hipMemcpyAsync(k->d_flag, k->h_flag, sizeof(int32_t), hipMemcpyHostToDevice, s));
            if (!nccl_ok(rccl().AllReduce(k->d_flag, k->d_flag, 1, ncclInt32, ncclMax, c->comm, s), "ncclAllReduce")) return 1;
            WO_HIP(hipMemcpyAsync(k->h_flag, k->d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            WO_HIP(hipStreamSynchronize(s));
            *(int32_t*)buf = *k->h_flag;
            return 0;
        }
        if (phase >= 2) {
            // the one rank that flooded the whole planet (phase 2) sends the planet's land heights: who it is, then the heights
            const bool sender = phase == 2;
            *k->h_flag = sender ? c->rank : -1;
            WO_HIP(hipMemcpyAsync(k->d_flag, k->h_flag, sizeof(int32_t), hipMemcpyHostToDevice, s));
            if (!nccl_ok(rccl().AllReduce(k->d_flag, k->d_flag, 1, ncclInt32, ncclMax, c->comm, s), "ncclAllReduce")) return 1;
            WO_HIP(hipMemcpyAsync(k->h_flag, k->d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            WO_HIP(hipStreamSynchronize(s));
            const int32_t root = *k->h_flag;
            if (root < 0 || root >= c->nranks) { wo::set_error("flood exchange: no rank sent the flooded heights"); return 1; }
            const size_t need = (size_t)std::max<int64_t>(n, 1);
            if (need > k->landCap) {
                if (k->d_land) (void)hipFree(k->d_land);
                k->d_land = nullptr; k->landCap = 0;
                WO_HIP(hipMalloc((void**)&k->d_land, need * sizeof(float)));
                k->landCap = need;
            }
            if (sender) WO_HIP(hipMemcpyAsync(k->d_land, buf, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s));
            if (!nccl_ok(rccl().Broadcast(k->d_land, k->d_land, (size_t)n, ncclFloat, root, c->comm, s), "ncclBroadcast")) return 1;
            if (!sender) WO_HIP(hipMemcpyAsync(buf, k->d_land, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
            WO_HIP(hipStreamSynchronize(s));
            return 0;
        }
        float* field = (float*)buf;
        const size_t M = (size_t)k->maxCount;
        if (!k->d_send) {
            WO_HIP(hipMalloc((void**)&k->d_send, M * sizeof(float)));
            WO_HIP(hipMalloc((void**)&k->d_all, M * (size_t)c->nranks * sizeof(float)));
            WO_HIP(hipHostMalloc((void**)&k->h_send, M * sizeof(float)));
            WO_HIP(hipHostMalloc((void**)&k->h_all, M * (size_t)c->nranks * sizeof(float)));
        }
        const int32_t* mine = k->cells.data() + k->start[c->rank];
        const int32_t nMine = k->counts[c->rank];
        for (int32_t i = 0; i < nMine; ++i) { if (mine[i] >= n) { wo::set_error("flood exchange: cell id out of range"); return 1; } k->h_send[i] = field[mine[i]]; }
        for (size_t i = (size_t)nMine; i < M; ++i) k->h_send[i] = 0.0f;
        WO_HIP(hipMemcpyAsync(k->d_send, k->h_send, M * sizeof(float), hipMemcpyHostToDevice, s));
        if (!nccl_ok(rccl().AllGather(k->d_send, k->d_all, M, ncclFloat, c->comm, s), "ncclAllGather")) return 1;
        WO_HIP(hipMemcpyAsync(k->h_all, k->d_all, M * (size_t)c->nranks * sizeof(float), hipMemcpyDeviceToHost, s));
        WO_HIP(hipStreamSynchronize(s));
        for (int32_t j = 0; j < c->nranks; ++j) {
            if (j == c->rank) continue;
            const int32_t* cj = k->cells.data() + k->start[j];
            const float* vj = k->h_all + (size_t)j * M;
            wo::parallel_ranges(k->counts[j], [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; ++i) field[cj[i]] = vj[i]; });
        }
        return 0;
    } catch (const wo::HipError& e) { wo::set_error(std::string("flood exchange: ") + e.msg); return 1; }
}

template <class T> void grow(T*& p, size_t& cap, size_t need) {
    if (need <= cap) return;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    if (hipMalloc((void**)&p, need * sizeof(T)) == hipSuccess) cap = need;
}
}  // namespace

extern "C" {

int wo_comm_unique_id(uint8_t* id) {
    if (!id) { wo::set_error("wo_comm_unique_id: null output"); return 1; }
    static_assert(sizeof(ncclUniqueId) == WO_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!rccl_ready("wo_comm_unique_id")) return 1;
    ncclUniqueId u;
    if (!nccl_ok(rccl().GetUniqueId(&u), "ncclGetUniqueId")) return 1;
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

int wo_comm_create(wo_ctx* ctx, const uint8_t* id, int32_t nranks, int32_t rank, wo_comm** out) {
    if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) { wo::set_error("wo_comm_create: bad arguments"); return 1; }
    if (!rccl_ready("wo_comm_create")) return 1;
    if (hipSetDevice(ctx->device) != hipSuccess) { wo::set_error("wo_comm_create: cannot select the context's device"); return 1; }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    wo_comm* c = new wo_comm;
    c->nranks = nranks; c->rank = rank; c->device = ctx->device;
    if (!nccl_ok(rccl().CommInitRank(&c->comm, nranks, u, rank), "ncclCommInitRank")) { delete c; return 1; }
    *out = c;
    return 0;
}

int wo_comm_destroy(wo_comm* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->gather) (void)hipFree(c->gather);
    if (c->send) (void)hipFree(c->send);
    delete c;
    return 0;
}

int wo_comm_rank(const wo_comm* c) { return c ? c->rank : -1; }
int wo_comm_size(const wo_comm* c) { return c ? c->nranks : 0; }

// counts[j] = length of rank j's send list (counts[rank] must equal this planet's); the receive list (set_halo) is the
// concatenation of the other ranks' contributions in rank order.
int wo_planet_exchange_allgather(wo_planet* p, wo_comm* c, const int32_t* counts) {
    if (!counts) { wo::set_error("wo_planet_exchange_allgather: bad arguments"); return 1; }
    if (!comm_ready(p, c, "wo_planet_exchange_allgather")) return 1;
    int64_t others = 0; int32_t maxCount = 1;
    for (int32_t j = 0; j < c->nranks; ++j) { if (counts[j] < 0) { wo::set_error("wo_planet_exchange_allgather: negative count"); return 1; } if (j != c->rank) others += counts[j]; maxCount = std::max(maxCount, counts[j]); }
    if (counts[c->rank] != p->nHaloSend || others != p->nHaloRecv) { wo::set_error("wo_planet_exchange_allgather: counts do not match the planet's halo lists (wo_planet_set_halo)"); return 1; }
    try {
        hipStream_t s = p->ctx->stream;
        grow(c->send, c->sendCap, (size_t)maxCount);
        grow(c->gather, c->gatherCap, (size_t)maxCount * (size_t)c->nranks);
        if (!c->send || !c->gather) { wo::set_error("wo_planet_exchange_allgather: out of device memory"); return 1; }
        if (p->nHaloSend > 0) wo::launch(p, wo::FAM_MISC, k_comm_pack, wo::blocks_for(p->nHaloSend), wo::WO_BLOCK, (const float*)p->d_e, (const int32_t*)p->d_haloSend, p->nHaloSend, c->send);
        if (p->nHaloSend < maxCount) WO_HIP(hipMemsetAsync(c->send + p->nHaloSend, 0, (size_t)(maxCount - p->nHaloSend) * sizeof(float), s));
        if (!nccl_ok(rccl().AllGather(c->send, c->gather, (size_t)maxCount, ncclFloat, c->comm, s), "ncclAllGather")) return 1;
        // the others' parts, compacted in rank order into the unpack buffer
        size_t at = 0;
        for (int32_t j = 0; j < c->nranks; ++j) {
            if (j == c->rank || counts[j] == 0) continue;
            WO_HIP(hipMemcpyAsync(p->d_haloBuf + at, c->gather + (size_t)j * maxCount, (size_t)counts[j] * sizeof(float), hipMemcpyDeviceToDevice, s));
            at += (size_t)counts[j];
        }
        if (p->nHaloRecv > 0) wo::launch(p, wo::FAM_MISC, k_comm_unpack, wo::blocks_for(p->nHaloRecv), wo::WO_BLOCK, p->d_e, (const int32_t*)p->d_haloRecv, p->nHaloRecv, (const float*)p->d_haloBuf);
        WO_HIP(hipStreamSynchronize(s));
        return 0;
    } catch (const wo::HipError& e) { wo::set_error(std::string("wo_planet_exchange_allgather: ") + e.msg); return 1; }
}

// Chain neighbours: the first nToPrev entries of the send list go to rank - 1, the rest to rank + 1; the first nFromPrev
// entries of the receive list come from rank - 1, the rest from rank + 1.  A missing neighbour (ends of the chain) has count 0.
int wo_planet_exchange_neighbors(wo_planet* p, wo_comm* c, int32_t nToPrev, int32_t nFromPrev) {
    if (!comm_ready(p, c, "wo_planet_exchange_neighbors")) return 1;
    const int32_t nToNext = p->nHaloSend - nToPrev, nFromNext = p->nHaloRecv - nFromPrev;
    const bool hasPrev = c->rank > 0, hasNext = c->rank + 1 < c->nranks;
    if (nToPrev < 0 || nFromPrev < 0 || nToNext < 0 || nFromNext < 0 || (!hasPrev && (nToPrev || nFromPrev)) || (!hasNext && (nToNext || nFromNext))) {
        wo::set_error("wo_planet_exchange_neighbors: counts do not match the planet's halo lists / the rank's place in the chain"); return 1;
    }
    try {
        hipStream_t s = p->ctx->stream;
        grow(c->send, c->sendCap, (size_t)std::max(p->nHaloSend, 1));
        if (!c->send) { wo::set_error("wo_planet_exchange_neighbors: out of device memory"); return 1; }
        if (p->nHaloSend > 0) wo::launch(p, wo::FAM_MISC, k_comm_pack, wo::blocks_for(p->nHaloSend), wo::WO_BLOCK, (const float*)p->d_e, (const int32_t*)p->d_haloSend, p->nHaloSend, c->send);
        GroupGuard group;
        if (!group.start()) return 1;
        bool ok = true;
        if (hasPrev && nToPrev) ok = ok && nccl_ok(rccl().Send(c->send, (size_t)nToPrev, ncclFloat, c->rank - 1, c->comm, s), "ncclSend");
        if (hasNext && nToNext) ok = ok && nccl_ok(rccl().Send(c->send + nToPrev, (size_t)nToNext, ncclFloat, c->rank + 1, c->comm, s), "ncclSend");
        if (hasPrev && nFromPrev) ok = ok && nccl_ok(rccl().Recv(p->d_haloBuf, (size_t)nFromPrev, ncclFloat, c->rank - 1, c->comm, s), "ncclRecv");
        if (hasNext && nFromNext) ok = ok && nccl_ok(rccl().Recv(p->d_haloBuf + nFromPrev, (size_t)nFromNext, ncclFloat, c->rank + 1, c->comm, s), "ncclRecv");
        if (!group.end() || !ok) { (void)rccl().CommAbort(c->comm); c->comm = nullptr; wo::set_error("wo_planet_exchange_neighbors: send / receive failed, the communicator was aborted"); return 1; }
        if (p->nHaloRecv > 0) wo::launch(p, wo::FAM_MISC, k_comm_unpack, wo::blocks_for(p->nHaloRecv), wo::WO_BLOCK, p->d_e, (const int32_t*)p->d_haloRecv, p->nHaloRecv, (const float*)p->d_haloBuf);
        WO_HIP(hipStreamSynchronize(s));
        return 0;
    } catch (const wo::HipError& e) { wo::set_error(std::string("wo_planet_exchange_neighbors: ") + e.msg); return 1; }
}

int wo_planet_set_flood_exchange_comm(wo_planet* p, const uint8_t* trueOcean, wo_comm* c, const int32_t* counts, const int32_t* cellsByRank) {
    if (!counts || !cellsByRank || !trueOcean) { wo::set_error("wo_planet_set_flood_exchange_comm: bad arguments"); return 1; }
    if (!comm_ready(p, c, "wo_planet_set_flood_exchange_comm")) return 1;
    FloodLink* k = new FloodLink;
    k->planet = p; k->comm = c;
    k->counts.assign(counts, counts + c->nranks);
    k->start.assign((size_t)c->nranks + 1, 0);
    for (int32_t j = 0; j < c->nranks; ++j) {
        if (counts[j] < 0) { delete k; wo::set_error("wo_planet_set_flood_exchange_comm: negative count"); return 1; }
        k->start[j + 1] = k->start[j] + counts[j];
        k->maxCount = std::max(k->maxCount, counts[j]);
    }
    k->cells.assign(cellsByRank, cellsByRank + k->start[c->nranks]);
    for (int32_t v : k->cells) if (v < 0 || v >= p->N) { delete k; wo::set_error("wo_planet_set_flood_exchange_comm: cell id out of range"); return 1; }
    if (hipMalloc((void**)&k->d_flag, sizeof(int32_t)) != hipSuccess || hipHostMalloc((void**)&k->h_flag, sizeof(int32_t)) != hipSuccess) {
        flood_link_free(k); wo::set_error("wo_planet_set_flood_exchange_comm: out of memory"); return 1;
    }
    if (wo_planet_set_flood_exchange(p, trueOcean, flood_link_exchange, k) != 0) { flood_link_free(k); return 1; }     // (frees a previous link)
    p->floodLink = k; p->floodLinkFree = flood_link_free;
    return 0;
}

}  // extern "C"
