// comm.cpp -- transports of the sharded build's exchange layer (see comm.h)
#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>

namespace katome {

// ---- ranks as threads of one process ------------------------------------------------------------------------------
bool LocalGroup::barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (poisoned) return false;
    const uint64_t gen = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(lk, [&] { return generation != gen || poisoned; });
    return !poisoned;
}
void LocalGroup::poison() {
    std::lock_guard<std::mutex> lk(mu);
    poisoned = true;
    cv.notify_all();
}
#define KBARRIER(g)                                                                                       \
    do {                                                                                                  \
        if (!(g)->barrier()) { set_error("another rank of this build failed"); return KATOME_E_DEVICE; } \
    } while (0)

namespace {

struct LocalTransport : Transport {
    std::shared_ptr<LocalGroup> g;
    int device;
    LocalTransport(std::shared_ptr<LocalGroup> grp, int r, int dev) : g(std::move(grp)), device(dev) { rank = r; world = g->world; }
    const char* kind() const override { return "local"; }
    int alltoallv(const void* send, const uint64_t* send_off, const uint64_t* send_cnt, void* recv, const uint64_t* recv_off,
                  const uint64_t* recv_cnt, size_t elem_bytes, int on_device, hipStream_t stream) override {
        // publish where my outgoing data lies (complete: my stream is drained first), then PULL my share from every peer
        if (on_device) KCHECK_HIP(hipStreamSynchronize(stream));
        g->send[rank] = send; g->send_off[rank] = send_off; g->send_cnt[rank] = send_cnt; g->dev[rank] = device;
        KBARRIER(g);
        int rc = KATOME_OK;
        for (int i = 0; i < world && rc == KATOME_OK; ++i) {
            const int p = (rank + i) % world;                      // start with myself, then round the ring: spreads the load
            const uint64_t n = g->send_cnt[p][rank];
            if (n != recv_cnt[p]) { set_error("alltoallv: rank %d sends %llu elements to rank %d which expects %llu", p, (unsigned long long)n, rank, (unsigned long long)recv_cnt[p]); rc = KATOME_E_ARG; break; }
            if (n == 0) continue;
            const char* src = static_cast<const char*>(g->send[p]) + g->send_off[p][rank] * elem_bytes;
            char* dst = static_cast<char*>(recv) + recv_off[p] * elem_bytes;
            if (!on_device) { memcpy(dst, src, n * elem_bytes); continue; }
            hipError_t e = g->dev[p] == device ? hipMemcpyAsync(dst, src, n * elem_bytes, hipMemcpyDeviceToDevice, stream)
                                               : hipMemcpyPeerAsync(dst, device, src, g->dev[p], n * elem_bytes, stream);
            if (e != hipSuccess) { set_error("alltoallv: peer copy %d -> %d failed: %s", p, rank, hipGetErrorString(e)); rc = KATOME_E_DEVICE; }
        }
        if (on_device && rc == KATOME_OK && hipStreamSynchronize(stream) != hipSuccess) { set_error("alltoallv: device failure"); rc = KATOME_E_DEVICE; }
        if (rc != KATOME_OK) { g->poison(); return rc; }
        KBARRIER(g);                                               // nobody reuses a send buffer that is still being read
        return rc;
    }
    int allreduce(uint64_t* vals, size_t n, int op) override {
        g->red[rank] = vals;
        KBARRIER(g);
        std::vector<uint64_t> acc(vals, vals + n);
        for (int p = 0; p < world; ++p) {
            if (p == rank) continue;
            const uint64_t* o = g->red[p];
            for (size_t i = 0; i < n; ++i) acc[i] = op == OP_SUM ? acc[i] + o[i] : op == OP_MAX ? std::max(acc[i], o[i]) : std::min(acc[i], o[i]);
        }
        KBARRIER(g);                                               // everybody has read everybody
        std::copy(acc.begin(), acc.end(), vals);
        return KATOME_OK;
    }
};

// ---- RCCL, opened at run time ---------------------------------------------------------------------------------------
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

int rccl_api(RcclApi** out) {
    static std::mutex mu;
    static RcclApi api;
    std::lock_guard<std::mutex> lk(mu);
    if (!api.h) {
        const char* names[] = {getenv("KATOME_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { if (n && *n && (api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break; }
        if (!api.h) { set_error("RCCL is not available (librccl.so.1 could not be opened: %s)", dlerror()); return KATOME_E_DEVICE; }
#define KATOME_NCCL_SYM(name) \
        if (!(api.name = reinterpret_cast<decltype(api.name)>(dlsym(api.h, "nccl" #name)))) { set_error("librccl lacks nccl" #name); api.h = nullptr; return KATOME_E_DEVICE; }
        KATOME_NCCL_SYM(GetUniqueId) KATOME_NCCL_SYM(CommInitRank) KATOME_NCCL_SYM(CommInitAll) KATOME_NCCL_SYM(CommDestroy)
        KATOME_NCCL_SYM(GroupStart) KATOME_NCCL_SYM(GroupEnd) KATOME_NCCL_SYM(Send) KATOME_NCCL_SYM(Recv) KATOME_NCCL_SYM(AllReduce)
        KATOME_NCCL_SYM(GetErrorString)
#undef KATOME_NCCL_SYM
    }
    *out = &api;
    return KATOME_OK;
}

#define KCHECK_NCCL(api, expr)                                                                                       \
    do {                                                                                                             \
        ncclResult_t _r = (expr);                                                                                    \
        if (_r != ncclSuccess) { set_error("RCCL error %d (%s) at %s:%d: %s", (int)_r, (api)->GetErrorString(_r), __FILE__, __LINE__, #expr); return KATOME_E_DEVICE; } \
    } while (0)

struct RcclTransport : Transport {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int device = 0;
    DevBuf red;                        // device staging for host reductions
    uint64_t* pinned = nullptr;        // ... and their pinned host side (a pageable H2D / D2H copy costs a millisecond, this one microseconds)
    size_t pinned_cap = 0;
    hipStream_t ctl = nullptr;         // ... on a stream of their own
    const char* kind() const override { return "rccl"; }
    ~RcclTransport() override {
        (void)hipSetDevice(device);
        red.release();
        if (pinned) (void)hipHostFree(pinned);
        if (ctl) { dev_retire_stream(ctl); (void)hipStreamDestroy(ctl); }
        if (comm) (void)api->CommDestroy(comm);
    }
    int alltoallv(const void* send, const uint64_t* send_off, const uint64_t* send_cnt, void* recv, const uint64_t* recv_off,
                  const uint64_t* recv_cnt, size_t elem_bytes, int on_device, hipStream_t stream) override {
        if (!on_device) { set_error("the RCCL transport moves device buffers only"); return KATOME_E_ARG; }
        KCHECK_NCCL(api, api->GroupStart());
        for (int i = 0; i < world; ++i) {
            const int p = (rank + i) % world, q = (rank - i + world) % world;        // send "forwards", receive "backwards"
            if (send_cnt[p]) KCHECK_NCCL(api, api->Send(static_cast<const char*>(send) + send_off[p] * elem_bytes, send_cnt[p] * elem_bytes, ncclChar, p, comm, stream));
            if (recv_cnt[q]) KCHECK_NCCL(api, api->Recv(static_cast<char*>(recv) + recv_off[q] * elem_bytes, recv_cnt[q] * elem_bytes, ncclChar, q, comm, stream));
        }
        KCHECK_NCCL(api, api->GroupEnd());
        return KATOME_OK;
    }
    int allreduce(uint64_t* vals, size_t n, int op) override {
        if (n == 0) return KATOME_OK;        // (a world of one goes through RCCL too: the one-GPU rehearsal of what N ranks do)
        KCHECK_HIP(hipSetDevice(device));
        hipStream_t s = have_work_stream ? work_stream : ctl;
        if (red.bytes < n * 8) KCHECK(red.alloc(std::max<size_t>(n * 8, 4096), ctl));
        if (pinned_cap < n) {
            if (pinned) (void)hipHostFree(pinned);
            pinned = nullptr; pinned_cap = 0;
            const size_t want = std::max<size_t>(n, 512);
            KCHECK_HIP(hipHostMalloc((void**)&pinned, want * 8, hipHostMallocDefault));
            pinned_cap = want;
        }
        memcpy(pinned, vals, n * 8);
        KCHECK_HIP(hipMemcpyAsync(red.p, pinned, n * 8, hipMemcpyHostToDevice, s));
        KCHECK_NCCL(api, api->AllReduce(red.p, red.p, n, ncclUint64, op == OP_SUM ? ncclSum : op == OP_MAX ? ncclMax : ncclMin, comm, s));
        KCHECK_HIP(hipMemcpyAsync(pinned, red.p, n * 8, hipMemcpyDeviceToHost, s));
        KCHECK_HIP(hipStreamSynchronize(s));
        memcpy(vals, pinned, n * 8);
        return KATOME_OK;
    }
};

struct CallbackTransport : Transport {
    katome_comm_callbacks cb;
    const char* kind() const override { return "callbacks"; }
    int alltoallv(const void* send, const uint64_t* send_off, const uint64_t* send_cnt, void* recv, const uint64_t* recv_off,
                  const uint64_t* recv_cnt, size_t elem_bytes, int on_device, hipStream_t stream) override {
        if (on_device) KCHECK_HIP(hipStreamSynchronize(stream));
        const int rc = cb.alltoallv(cb.user, send, send_off, send_cnt, recv, recv_off, recv_cnt, elem_bytes, on_device);
        if (rc != 0) { set_error("the caller's alltoallv failed with %d", rc); return KATOME_E_DEVICE; }
        return KATOME_OK;
    }
    int allreduce(uint64_t* vals, size_t n, int op) override {
        const int rc = cb.allreduce_u64(cb.user, vals, n, op);
        if (rc != 0) { set_error("the caller's allreduce failed with %d", rc); return KATOME_E_DEVICE; }
        return KATOME_OK;
    }
};

int finish_rccl(RcclApi* api, ncclComm_t c, int rank, int world, int device, katome_comm** out) {
    auto t = std::make_unique<RcclTransport>();
    t->api = api; t->comm = c; t->rank = rank; t->world = world; t->device = device;
    KCHECK_HIP(hipStreamCreateWithFlags(&t->ctl, hipStreamNonBlocking));
    katome_comm* kc = new katome_comm();
    kc->t = std::move(t); kc->device = device;
    *out = kc;
    return KATOME_OK;
}

}  // namespace

int make_local_comm(std::shared_ptr<LocalGroup> group, int rank, int device, katome_comm** out) {
    katome_comm* kc = new katome_comm();
    kc->t = std::make_unique<LocalTransport>(std::move(group), rank, device);
    kc->device = device;
    *out = kc;
    return KATOME_OK;
}

int rccl_unique_id(uint8_t* id128) {
    RcclApi* api = nullptr;
    KCHECK(rccl_api(&api));
    static_assert(sizeof(ncclUniqueId) == KATOME_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    KCHECK_NCCL(api, api->GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return KATOME_OK;
}

int make_rccl_comm(const uint8_t* id128, int rank, int world, int device, katome_comm** out) {
    RcclApi* api = nullptr;
    KCHECK(rccl_api(&api));
    KCHECK(use_device(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    KCHECK_NCCL(api, api->CommInitRank(&c, world, id, rank));
    return finish_rccl(api, c, rank, world, device, out);
}

int make_rccl_comms_all(const int* devices, int n, katome_comm** out) {
    RcclApi* api = nullptr;
    KCHECK(rccl_api(&api));
    std::vector<ncclComm_t> cs(n, nullptr);
    KCHECK_NCCL(api, api->CommInitAll(cs.data(), n, devices));
    for (int r = 0; r < n; ++r) {
        KCHECK(use_device(devices[r]));
        KCHECK(finish_rccl(api, cs[r], r, n, devices[r], &out[r]));
    }
    return KATOME_OK;
}

}  // namespace katome

using namespace katome;

int katome_comm::exchange_counts(const uint64_t* send_cnt, uint64_t* recv_cnt, uint64_t* global_max, uint64_t* global_total) {
    const int w = world(), r = rank();
    std::vector<uint64_t> m((size_t)w * w, 0);                     // m[src][dst]; everybody fills its own row
    for (int p = 0; p < w; ++p) m[(size_t)r * w + p] = send_cnt[p];
    KCHECK(t->allreduce(m.data(), m.size(), OP_SUM));
    for (int p = 0; p < w; ++p) recv_cnt[p] = m[(size_t)p * w + r];
    if (global_max) { uint64_t mx = 0; for (uint64_t v : m) mx = std::max(mx, v); *global_max = mx; }
    if (global_total) { uint64_t t = 0; for (uint64_t v : m) t += v; *global_total = t; }      // (records on the move anywhere)
    return KATOME_OK;
}

int katome_comm::allgather(uint64_t v, uint64_t* out) {
    const int w = world();
    for (int p = 0; p < w; ++p) out[p] = 0;
    out[rank()] = v;
    return t->allreduce(out, (size_t)w, OP_SUM);
}

int katome_comm::exchange(const void* send, const uint64_t* send_cnt, void* recv, const uint64_t* recv_cnt, size_t elem_bytes, int on_device,
                          hipStream_t stream, bool one_round, uint64_t known_max, const uint64_t* send_off) {
    const int w = world();
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> so(w, 0), ro(w, 0);
    uint64_t biggest = 0, out_b = 0, in_b = 0;
    for (int p = 0; p < w; ++p) {
        if (p) { so[p] = so[p - 1] + send_cnt[p - 1]; ro[p] = ro[p - 1] + recv_cnt[p - 1]; }
        if (send_off) so[p] = send_off[p];
        biggest = std::max(biggest, std::max(send_cnt[p], recv_cnt[p]));
        if (p != rank()) { out_b += send_cnt[p] * elem_bytes; in_b += recv_cnt[p] * elem_bytes; stats.max_pair_bytes = std::max<uint64_t>(stats.max_pair_bytes, send_cnt[p] * elem_bytes); }
    }
    const uint64_t chunk = std::max<uint64_t>(1, max_message_bytes / elem_bytes);
    uint64_t rounds = 1;
    if (one_round) rounds = 1;
    else if (w > 1) {                                              // every rank must run the same number of rounds
        if (known_max != MAX_UNKNOWN) biggest = known_max;         // (the count matrix was seen whole: no second agreement)
        else KCHECK(t->allreduce(&biggest, 1, OP_MAX));
        rounds = std::max<uint64_t>(1, (biggest + chunk - 1) / chunk);
    } else rounds = std::max<uint64_t>(1, (biggest + chunk - 1) / chunk);
    if (rounds == 1) {
        KCHECK(t->alltoallv(send, so.data(), send_cnt, recv, ro.data(), recv_cnt, elem_bytes, on_device, stream));
    } else {
        std::vector<uint64_t> s_off(w), s_n(w), r_off(w), r_n(w);
        for (uint64_t r = 0; r < rounds; ++r) {
            for (int p = 0; p < w; ++p) {
                const uint64_t sb = std::min(send_cnt[p], r * chunk), rb = std::min(recv_cnt[p], r * chunk);
                s_off[p] = so[p] + sb; s_n[p] = std::min(send_cnt[p] - sb, chunk);
                r_off[p] = ro[p] + rb; r_n[p] = std::min(recv_cnt[p] - rb, chunk);
            }
            KCHECK(t->alltoallv(send, s_off.data(), s_n.data(), recv, r_off.data(), r_n.data(), elem_bytes, on_device, stream));
        }
    }
    stats.calls += 1; stats.bytes_out += out_b; stats.bytes_in += in_b;
    stats.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return KATOME_OK;
}

extern "C" {

int katome_comm_unique_id(uint8_t* id) {
    if (!id) { set_error("null argument"); return KATOME_E_ARG; }
    return rccl_unique_id(id);
}
int katome_comm_create_rccl(const uint8_t* id, int rank, int world, int device, katome_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { set_error("bad communicator arguments"); return KATOME_E_ARG; }
    *out = nullptr;
    return make_rccl_comm(id, rank, world, device, out);
}
int katome_comm_create_callbacks(const katome_comm_callbacks* cb, int rank, int world, int device, katome_comm** out) {
    if (!cb || !cb->alltoallv || !cb->allreduce_u64 || !out || world < 1 || rank < 0 || rank >= world) { set_error("bad communicator arguments"); return KATOME_E_ARG; }
    auto t = std::make_unique<CallbackTransport>();
    t->cb = *cb; t->rank = rank; t->world = world;
    katome_comm* kc = new katome_comm();
    kc->t = std::move(t); kc->device = device;
    *out = kc;
    return KATOME_OK;
}
void katome_comm_destroy(katome_comm* c) { delete c; }
int katome_comm_rank(const katome_comm* c) { return c ? c->rank() : -1; }
int katome_comm_world(const katome_comm* c) { return c ? c->world() : 0; }
const char* katome_comm_kind(const katome_comm* c) { return c ? c->t->kind() : ""; }
int katome_comm_set_max_message_bytes(katome_comm* c, uint64_t bytes) {
    if (!c || bytes == 0) { set_error("bad argument"); return KATOME_E_ARG; }
    c->max_message_bytes = bytes;
    return KATOME_OK;
}
int katome_comm_allreduce_u64(katome_comm* c, uint64_t* vals, uint64_t n, int op) {
    if (!c || (!vals && n) || op < 0 || op > 2) { set_error("bad argument"); return KATOME_E_ARG; }
    return c->allreduce(vals, n, op);
}
int katome_comm_exchange(katome_comm* c, const void* send, const uint64_t* send_cnt, void* recv, uint64_t recv_capacity, uint64_t* recv_cnt,
                         uint64_t elem_bytes, int on_device, void* stream) {
    if (!c || !send_cnt || !recv_cnt || elem_bytes == 0) { set_error("bad argument"); return KATOME_E_ARG; }
    KCHECK(c->exchange_counts(send_cnt, recv_cnt));
    uint64_t total = 0;
    for (int p = 0; p < c->world(); ++p) total += recv_cnt[p];
    // (every rank must still take part: a rank whose buffer is too small fails AFTER the collective would deadlock the rest,
    // so the check is collective too)
    uint64_t bad = total > recv_capacity ? 1 : 0;
    KCHECK(c->allreduce(&bad, 1, OP_MAX));
    if (bad) { set_error("exchange: a receive buffer is too small (%llu elements arrive here, room for %llu)", (unsigned long long)total, (unsigned long long)recv_capacity); return KATOME_E_ARG; }
    return c->exchange(send, send_cnt, recv, recv_cnt, elem_bytes, on_device, (hipStream_t)stream);
}

}  // extern "C"
