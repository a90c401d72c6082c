// comm.h -- the exchange layer of the sharded (multi-GPU) build.
//
// The reference has no counterpart (katome never spawns a thread): this is the one real exchange
// step of the MI355X design -- records leave the rank that extracted them for the rank that owns their key.  A
// communicator is one rank's end of a group of `world` ranks, one rank per GPU, and offers exactly two operations:
//   * alltoallv: device buffers of fixed-size elements, a (offset, count) pair per peer on both sides;
//   * allreduce: a vector of u64 on the host (counts, histograms, agreement on rounds).
// Transports:
//   * RcclTransport -- RCCL over xGMI (grouped ncclSend/ncclRecv: RCCL has no alltoallv; the 8 GPUs of an MI355X node
//     are a full mesh, so the exchange runs on all 7 links of every GPU at once and is bound by the most loaded link).
//     One communicator per rank: ranks are threads of one process (katome_build_* with settings.n_devices > 1) or one
//     process per GPU (bench.py under a launcher: the unique id travels out of band).  librccl is opened at run time,
//     only when such a communicator is asked for.
//   * LocalTransport -- ranks are threads of one process and move the data themselves with peer copies
//     (hipMemcpyPeerAsync, pull side).  What lets several ranks SHARE one GPU (RCCL refuses that): the rehearsal of the
//     whole sharded route on a one-GPU box.
//   * CallbackTransport -- the caller moves the bytes (tests: torch.distributed/gloo on host buffers).
#pragma once
#include <condition_variable>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"

namespace katome {

enum ReduceOp { OP_SUM = 0, OP_MAX = 1, OP_MIN = 2 };

struct Transport {
    int rank = 0, world = 1;
    // the stream the caller's device work (and its alltoallv calls) is ordered on; RCCL reductions are enqueued there too, so
    // that every operation of a communicator is issued on ONE stream, in one order
    hipStream_t work_stream = nullptr;
    bool have_work_stream = false;
    virtual ~Transport() {}
    // element counts and offsets per peer; on return `recv` is complete for work enqueued on `stream` afterwards and
    // `send` may be overwritten by such work.  on_device = 0: host buffers (control-plane sized)
    virtual int alltoallv(const void* send, const uint64_t* send_off, const uint64_t* send_cnt, void* recv, const uint64_t* recv_off,
                          const uint64_t* recv_cnt, size_t elem_bytes, int on_device, hipStream_t stream) = 0;
    virtual int allreduce(uint64_t* vals, size_t n, int op) = 0;
    virtual const char* kind() const = 0;
};

// per-exchange accounting (bench.py: bytes that left this rank per phase, and the time the exchange took)
struct ExchangeStats { uint64_t calls = 0, bytes_out = 0, bytes_in = 0, max_pair_bytes = 0; double ms = 0; };

}  // namespace katome

// the opaque handle of the C ABI
struct katome_comm {
    std::unique_ptr<katome::Transport> t;
    int device = 0;
    // a single message larger than this goes in rounds (a 3.8 GB exchange came back corrupted from RCCL 2.26 on this stack)
    uint64_t max_message_bytes = 1ull << 30;
    katome::ExchangeStats stats;
    int rank() const { return t->rank; }
    int world() const { return t->world; }
    // all-to-all of one u64 per peer
    // (global_max, optional: the largest single (rank -> peer) count of the whole matrix -- every rank sees all of it --, which
    // exchange() would otherwise agree on with a reduction of its own: pass it on as `known_max`)
    int exchange_counts(const uint64_t* send_cnt, uint64_t* recv_cnt, uint64_t* global_max = nullptr, uint64_t* global_total = nullptr);
    int allgather(uint64_t v, uint64_t* out);
    int allreduce(uint64_t* vals, size_t n, int op) { return t->allreduce(vals, n, op); }
    void use_stream(hipStream_t s) { t->work_stream = s; t->have_work_stream = true; }
    // records grouped by destination, send_cnt[p] elements for peer p, contiguous in peer order; recv likewise by source
    // (recv_cnt from exchange_counts).  Splits into rounds when a pair's message exceeds max_message_bytes.
    // one_round: the caller knows that no pair's message exceeds max_message_bytes on any rank (no agreement needed)
    static constexpr uint64_t MAX_UNKNOWN = ~0ull;
    // send_off (optional): where each peer's records start in `send`, in elements (they need not be adjacent)
    int exchange(const void* send, const uint64_t* send_cnt, void* recv, const uint64_t* recv_cnt, size_t elem_bytes, int on_device,
                 hipStream_t stream, bool one_round = false, uint64_t known_max = MAX_UNKNOWN, const uint64_t* send_off = nullptr);
};

namespace katome {

// ranks that are threads of one process: shared rendezvous state (LocalTransport, and the thread group of an n_devices build)
struct LocalGroup {
    explicit LocalGroup(int n) : world(n), send(n), send_off(n), send_cnt(n), dev(n), red(n) {}
    const int world;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void*> send;
    std::vector<const uint64_t*> send_off, send_cnt;
    std::vector<int> dev;
    std::vector<uint64_t*> red;
    bool poisoned = false;          // a rank failed: every barrier, now and later, returns false instead of waiting for it
    bool barrier();
    void poison();
};

int make_local_comm(std::shared_ptr<LocalGroup> group, int rank, int device, katome_comm** out);
int rccl_unique_id(uint8_t* id128);
int make_rccl_comm(const uint8_t* id128, int rank, int world, int device, katome_comm** out);
// n communicators of one process at once (ncclCommInitAll), one per device of `devices`
int make_rccl_comms_all(const int* devices, int n, katome_comm** out);

}  // namespace katome
