// dist.hip -- the sharded (multi-GPU) build: one rank per GPU, one exchange step per phase.
//
// The reference's Build::create (builder.rs:42-54) is one sequential loop over the reads (builder.rs:152-160); reads are
// independent until their k-mers meet in the graph, so what shards is the read set:
//   1. every rank cuts its own reads (a contiguous run of the input) into records -- TILES, the (k+span-1)-mers covering
//      `span` consecutive windows, plus the windows left over (table.hip) -- and routes each record to its owner rank,
//      owner = mulhi(mix(record), world) (radix.hip partition; three-word tiles included);
//   2. one all-to-all per batch brings the records to their owners (comm.h), which count them in their tile table;
//   3. every rank turns its distinct tiles into (k-mer, count) records and a second, much smaller all-to-all brings those
//      to the K-MER's owner = mulhi(mix(canonical middle (k-2)-mer), world): shared by a k-mer and its reverse complement
//      and by all out-edges of a node, so a node and its <= 4 out-edges live on one rank (the HmGIR shape, hm_gir.rs:91-153);
//   4. the graph is numbered across ranks.  By packed key: a rank's nodes with out-edges are the run heads of its own sorted
//      edges; every edge asks the owner of its target for the target's id (one key out, one id back), which also registers
//      the nodes without out-edges (they count in node_count, stats/collections.rs:196).  In the reference's own numbering
//      (KATOME_FLAG_FIRST_SEEN_ORDER; petgraph indices, pt_graph.rs:149,194): every record carries where it sits in the
//      input, owners keep the earliest sequence number per k-mer and strand, and edges and nodes get their GLOBAL rank
//      among all ranks' sequence numbers (global_rank: range partition by splitters from a merged histogram, local sort,
//      ranks back over the mirrored exchange) -- the same indices whatever the number of ranks.
// The stages of assemble_with_graph that walk petgraph's adjacency and swap_remove by index (pruner.rs:36-82) need the whole
// graph in index order: katome_dist_gather brings it to one rank, where prune.hip runs unchanged.
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include <cstring>
#include "dist_builder.h"

const char* const XPHASE_NAMES[X_COUNT] = {"exchange_records", "exchange_kmers", "exchange_targets", "exchange_ids",
                                           "rank_nodes", "rank_edges", "gather", "exchange_mid_tiles", "prune"};

namespace {

// ---- small kernels ---------------------------------------------------------------------------------------------------
#define KLAUNCH(kernel, n, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid_for((n), BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, __VA_ARGS__)
#define KLOOP(i, n) for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < (n); i += (u64)gridDim.x * BLOCK)

// every record whose key was not found among the sources (rank == UINT64_MAX): its key and where it stands.  One cursor
// atomic per workgroup tile of 2048 records (the misses are a few per cent: one atomic each on a single address serialises --
// 13 ms for 3e6 of them)
constexpr int MISS_ITEMS = 8;
template <int NW>
__global__ __launch_bounds__(BLOCK) void compact_missing_kernel(const u64* __restrict__ rank, const u64* __restrict__ keys, u64 n,
                                                                 u64* __restrict__ mk, u32* __restrict__ mpos, u64* cursor) {
    __shared__ u32 wtot[BLOCK / 64];
    __shared__ u64 base;
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 tile = (u64)BLOCK * MISS_ITEMS;
    for (u64 t0 = (u64)blockIdx.x * tile; t0 < n; t0 += (u64)gridDim.x * tile) {
        bool miss[MISS_ITEMS]; u32 mine = 0;
#pragma unroll
        for (int j = 0; j < MISS_ITEMS; ++j) {
            const u64 i = t0 + (u64)j * BLOCK + threadIdx.x;
            miss[j] = i < n && rank[i] == ~0ull;
            mine += miss[j];
        }
        u32 incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += v; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        u32 woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) { if (w < (int)wave) woff += wtot[w]; total += wtot[w]; }
        if (threadIdx.x == 0 && total) base = atomicAdd((unsigned long long*)cursor, (unsigned long long)total);
        __syncthreads();
        if (total) {
            u64 pos = base + woff + (incl - mine);
#pragma unroll
            for (int j = 0; j < MISS_ITEMS; ++j) {
                if (!miss[j]) continue;
                const u64 i = t0 + (u64)j * BLOCK + threadIdx.x;
#pragma unroll
                for (int q = 0; q < NW; ++q) mk[pos * NW + q] = keys[i * NW + q];
                mpos[pos] = (u32)i;
                ++pos;
            }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(BLOCK) void fill_missing_kernel(u64* __restrict__ rank, const u32* __restrict__ mpos, const u64* __restrict__ mrank,
                                                              u64 m, u64 offset) {
    KLOOP(i, m) rank[mpos[i]] = offset + mrank[i];
}
// out[j] = map ? map[in[j]] : in[j] + base
__global__ __launch_bounds__(BLOCK) void map_ids_kernel(const u64* __restrict__ in, u64 n, const u64* __restrict__ map, u64 base, u64* __restrict__ out) {
    KLOOP(j, n) out[j] = map ? map[in[j]] : in[j] + base;
}
__global__ __launch_bounds__(BLOCK) void scatter_u64_kernel(const u64* __restrict__ vals, const u32* __restrict__ pos, u64 n, u64* __restrict__ out) {
    KLOOP(j, n) out[pos[j]] = vals[j];
}
// first-seen order: node_first[node] = earliest (2 * sequence number + role) over the edges that touch it
// (source role: a node's out-edges are one run of the sorted edges -- its head stores the run's minimum, no atomic)
__global__ __launch_bounds__(BLOCK) void src_first_kernel(const u64* __restrict__ lsrc, const u64* __restrict__ seq, u64 n, u64* __restrict__ node_first) {
    KLOOP(e, n) {
        const u64 s = lsrc[e];
        if (e > 0 && lsrc[e - 1] == s) continue;
        u64 m = seq[e];
        for (u64 j = e + 1; j < n && lsrc[j] == s; ++j) m = seq[j] < m ? seq[j] : m;
        node_first[s] = 2 * m;
    }
}
__global__ __launch_bounds__(BLOCK) void dst_first_kernel(const u64* __restrict__ local, const u64* __restrict__ val, u64 n, u64* node_first) {
    KLOOP(j, n) atomicMin((unsigned long long*)&node_first[local[j]], (unsigned long long)val[j]);
}
__global__ __launch_bounds__(BLOCK) void target_value_kernel(const u64* __restrict__ seq, const u32* __restrict__ origin, u64 n, u64* __restrict__ out) {
    KLOOP(j, n) out[j] = 2 * seq[origin[j]] + 1;
}
__global__ __launch_bounds__(BLOCK) void value_hist_kernel(const u64* __restrict__ v, u64 n, u64 width, unsigned long long* hist) {
    KLOOP(i, n) atomicAdd(&hist[v[i] / width], 1ull);
}
__global__ __launch_bounds__(BLOCK) void assign_rank_kernel(const u32* __restrict__ pos, u64 n, u64 base, u64* __restrict__ out) {
    KLOOP(i, n) out[pos[i]] = base + i;
}
// gather-to-root: records land at their global index
template <int NW>
__global__ __launch_bounds__(BLOCK) void place_keys_kernel(const u64* __restrict__ gid, const u64* __restrict__ in, u64 n, u64* __restrict__ out) {
    KLOOP(j, n) {
#pragma unroll
        for (int q = 0; q < NW; ++q) out[gid[j] * NW + q] = in[j * NW + q];
    }
}
template <class T>
__global__ __launch_bounds__(BLOCK) void place_kernel(const u64* __restrict__ gid, const T* __restrict__ in, u64 n, T* __restrict__ out) {
    KLOOP(j, n) out[gid[j]] = in[j];
}

// out[pos[j]] = v for the entries [a, b) of a partitioned list (the owner rank of a segment's look-ups)
__global__ __launch_bounds__(BLOCK) void scatter_const_kernel(const u32* __restrict__ pos, u64 a, u64 b, u64 v, u64* __restrict__ out) {
    KLOOP(j, b - a) out[pos[a + j]] = v;
}

uint64_t sum(const std::vector<uint64_t>& v) { uint64_t t = 0; for (uint64_t x : v) t += x; return t; }

// KATOME_DIST_TRACE=1: order-free checksum of a device array of u64 words at the checkpoints of a sharded build (stderr)
void trace_words(const char* what, int rank, const void* d_ptr, uint64_t n_words, hipStream_t stream) {
    static const bool on = getenv("KATOME_DIST_TRACE") != nullptr;
    if (!on) return;
    std::vector<uint64_t> h(n_words);
    (void)hipStreamSynchronize(stream);
    if (n_words) (void)hipMemcpy(h.data(), d_ptr, n_words * 8, hipMemcpyDeviceToHost);
    if (const char* dir = getenv("KATOME_DIST_DUMP")) {
        std::string name = std::string(dir) + "/" + what + "." + std::to_string(rank) + ".bin";
        for (char& c : name) if (c == ' ' || c == ':') c = '_';
        if (FILE* f = fopen(name.c_str(), "wb")) {
            if (strstr(what, "slots")) { for (uint64_t i = 0; i + 3 < n_words; i += 4) if (h[i] >> 63) fwrite(&h[i], 8, 4, f); }   // occupied 32-byte slots only
            else fwrite(h.data(), 8, n_words, f);
            fclose(f);
        }
    }
    uint64_t a = 0, x = 0, mx = 0;
    for (uint64_t v : h) { a += v * 0x9E3779B97F4A7C15ull; x ^= v; mx = std::max(mx, v); }
    fprintf(stderr, "[dist %d] %-28s n=%llu sum=%016llx xor=%016llx max=%llu\n", rank, what, (unsigned long long)n_words,
            (unsigned long long)a, (unsigned long long)x, (unsigned long long)mx);
}

}  // namespace

namespace {

// One batch's records, already grouped by owner (`part`, counts per owner; idx: each record's index in this rank's batch,
// first-seen order only): exchange, then count what arrived in the tile table (tiles) or the k-mer table.
int route_and_insert(katome_dist_builder* d, const u64* part, const u32* idx, const std::vector<uint64_t>& counts, uint32_t nwr, bool tiles,
                     uint64_t read0, uint32_t per_read, uint32_t win0, uint32_t span, hipStream_t stream) {
    katome_builder* b = d->b;
    const int world = d->world();
    std::vector<uint64_t> rcnt(world, 0);
    uint64_t pair_max = 0;
    KCHECK(d->comm->exchange_counts(counts.data(), rcnt.data(), &pair_max));
    const uint64_t nR = sum(rcnt);
    DevBuf recv(stream), ridx(stream);
    KCHECK(recv.alloc(std::max<uint64_t>(nR, 1) * 8 * nwr));
    KCHECK(d->xchg(X_RECORDS, part, counts.data(), recv.p, rcnt.data(), 8 * nwr, stream, false, pair_max));
    SeenOrigin origin;
    if (d->first_seen) {
        KCHECK(ridx.alloc(std::max<uint64_t>(nR, 1) * 4));
        KCHECK(d->xchg(X_RECORDS, idx, counts.data(), ridx.p, rcnt.data(), 4, stream, false, pair_max));
        std::vector<uint64_t> read0s(world, 0);
        KCHECK(d->comm->allgather(read0, read0s.data()));
        origin.idx = ridx.as<u32>(); origin.n_seg = (uint32_t)world;
        for (int p = 0; p < world; ++p) { origin.seg_off[p + 1] = origin.seg_off[p] + rcnt[p]; origin.seg_read0[p] = read0s[p]; }
        origin.per_read = per_read; origin.span = span; origin.windows = d->W; origin.win0 = win0; origin.rc = d->rc;
    }
    if (nR == 0) return KATOME_OK;
    if (tiles) {
        b->span = span;
        return builder_insert(b, b->tiles, b->tiles_ready, d->nwt, b->s.table_slots_hint / 4, recv.as<u64>(), nullptr, nR,
                              d->first_seen ? &origin : nullptr, PH_INSERT_TILES, stream);
    }
    return builder_insert(b, b->table, b->table_ready, d->nw, b->s.table_slots_hint, recv.as<u64>(), nullptr, nR,
                          d->first_seen ? &origin : nullptr, PH_INSERT, stream);
}

// Weighted records [with their two sequence numbers] to their owners, which add them to `table`: in slices of at most a quarter
// of one message's size, the same number of rounds on every rank.  core_bases == 0: the owner is a hash of the whole record.
// Software pipeline over the slices: slice j is partitioned on the build's stream, exchanged on the exchange stream (every
// operation of the communicator goes there for the duration), and added to the table on the build's stream again -- while
// slice j + 1 is already on the links.  Two sets of buffers; events order the two streams (a buffer is written again only
// behind the wait for the exchange or the insertion that read it, both of which the build's stream has passed by then).
// what arrived, kept as records instead of being counted in a table (the last level counted by sorting, table.hip)
struct Collected {
    DevBuf keys, weights;
    uint64_t n = 0, cap = 0;
    explicit Collected(hipStream_t s) : keys(s), weights(s) {}
    int append(const void* k, const void* w, uint64_t m, uint32_t nwr, uint64_t guess, hipStream_t stream) {
        if (n + m > cap) {                                   // (a rank receives about what it sends: rarely more than one growth)
            const uint64_t want = std::max<uint64_t>(std::max<uint64_t>(cap + cap / 2, n + m), guess);
            DevBuf nk(stream), nwt(stream);
            KCHECK(nk.alloc((want + 1) * 8 * nwr)); KCHECK(nwt.alloc((want + 1) * 4));
            if (n) {
                KCHECK_HIP(hipMemcpyAsync(nk.p, keys.p, n * 8 * nwr, hipMemcpyDeviceToDevice, stream));
                KCHECK_HIP(hipMemcpyAsync(nwt.p, weights.p, n * 4, hipMemcpyDeviceToDevice, stream));
            }
            keys.release(); weights.release();
            { const size_t bytes = nk.bytes; keys.adopt(nk.take(), bytes); }
            { const size_t bytes = nwt.bytes; weights.adopt(nwt.take(), bytes); }
            cap = want;
        }
        KCHECK_HIP(hipMemcpyAsync((char*)keys.p + n * 8 * nwr, k, m * 8 * nwr, hipMemcpyDeviceToDevice, stream));
        KCHECK_HIP(hipMemcpyAsync((char*)weights.p + n * 4, w, m * 4, hipMemcpyDeviceToDevice, stream));
        n += m;
        return KATOME_OK;
    }
};

// `collect`: the records that arrive are appended there instead of being added to `table`
int route_weighted(katome_dist_builder* d, int xphase, const DevBuf& keys, const DevBuf& weights, const DevBuf& seen, uint64_t n_rec, uint32_t nwr,
                   uint32_t core_shift, uint32_t core_bases, Table& table, bool& ready, uint64_t hint, int phase, hipStream_t stream,
                   Collected* collect = nullptr) {
    katome_builder* b = d->b;
    const int world = d->world();
    const uint64_t per_slice = std::max<uint64_t>(1, d->comm->max_message_bytes / 4 / (8 * nwr));
    uint64_t ns = (n_rec + per_slice - 1) / per_slice;
    KCHECK(d->comm->allreduce(&ns, 1, OP_MAX));
    if (ns == 0) return KATOME_OK;
    if (!d->xstream) KCHECK_HIP(hipStreamCreateWithFlags(&d->xstream, hipStreamNonBlocking));
    hipStream_t X = d->xstream;
    struct Slot {
        DevBuf pk, pw, pidx, idx, ppairs, rk, rw, rp;
        std::vector<uint64_t> counts, rcnt;
        uint64_t nR = 0;
        hipEvent_t parted = nullptr, arrived = nullptr;
        explicit Slot(hipStream_t s) : pk(s), pw(s), pidx(s), idx(s), ppairs(s), rk(s), rw(s), rp(s) {}
        ~Slot() { if (parted) (void)hipEventDestroy(parted); if (arrived) (void)hipEventDestroy(arrived); }
    };
    Slot slot[2] = {Slot(stream), Slot(stream)};
    for (auto& sl : slot) {
        KCHECK_HIP(hipEventCreateWithFlags(&sl.parted, hipEventDisableTiming));
        KCHECK_HIP(hipEventCreateWithFlags(&sl.arrived, hipEventDisableTiming));
    }
    d->comm->use_stream(X);
    auto insert_slice = [&](Slot& sl) -> int {
        KCHECK_HIP(hipStreamWaitEvent(stream, sl.arrived, 0));
        if (xphase == X_KMERS) {
            trace_words("kmers received: keys", d->rank(), sl.rk.p, sl.nR * nwr, stream);
            trace_words("kmers received: weights", d->rank(), sl.rw.p, sl.nR / 2, stream);
        }
        SeenOrigin origin;
        if (d->first_seen) { origin.pairs = sl.rp.as<u64>(); origin.rc = d->rc; }
        // (room for what this rank sends and an eighth more to begin with; KATOME_SORTED_COUNT=2 -- tests -- starts from nothing,
        // so that every slice makes the array grow)
        static const bool tight = getenv("KATOME_SORTED_COUNT") && atoi(getenv("KATOME_SORTED_COUNT")) == 2;
        if (sl.nR && collect) KCHECK(collect->append(sl.rk.p, sl.rw.p, sl.nR, nwr, tight ? 0 : n_rec + n_rec / 8 + (1u << 20), stream));
        else if (sl.nR) KCHECK(builder_insert(b, table, ready, nwr, hint, sl.rk.as<u64>(), sl.rw.as<u32>(), sl.nR, d->first_seen ? &origin : nullptr, phase, stream));
        return KATOME_OK;
    };
    int rc = KATOME_OK;
    for (uint64_t j = 0; j < ns && rc == KATOME_OK; ++j) {
        Slot& sl = slot[j & 1];
        const uint64_t a = std::min(n_rec, j * per_slice), e = std::min(n_rec, (j + 1) * per_slice), m = e - a;
        sl.counts.assign(world, 0); sl.rcnt.assign(world, 0);
        // (a) partition slice j by owner, on the build's stream
        if (m) {
            if ((rc = sl.pk.alloc(m * 8 * nwr)) || (rc = sl.pw.alloc(m * 4))) break;
            if (d->first_seen) {
                if ((rc = sl.idx.alloc(m * 4)) || (rc = sl.pidx.alloc(m * 4)) || (rc = sl.ppairs.alloc(m * 16))) break;
                if ((rc = dev_iota(sl.idx.as<u32>(), m, stream))) break;
                if ((rc = dev_partition(keys.as<u64>() + a * nwr, sl.idx.as<u32>(), m, nwr, world, sl.pk.as<u64>(), sl.pidx.as<u32>(), sl.counts.data(), stream, core_shift, core_bases))) break;
                if ((rc = dev_gather_u32(weights.as<u32>() + a, sl.pidx.as<u32>(), sum(sl.counts), sl.pw.as<u32>(), stream))) break;
                if ((rc = dev_gather_keys(seen.as<u64>() + 2 * a, sl.pidx.as<u32>(), sum(sl.counts), 2, sl.ppairs.as<u64>(), stream))) break;
            } else {
                if ((rc = dev_partition(keys.as<u64>() + a * nwr, weights.as<u32>() + a, m, nwr, world, sl.pk.as<u64>(), sl.pw.as<u32>(), sl.counts.data(), stream, core_shift, core_bases))) break;
            }
        }
        if (hipEventRecord(sl.parted, stream) != hipSuccess) { set_error("event record failed"); rc = KATOME_E_DEVICE; break; }
        // (b) exchange it, on the exchange stream
        if ((rc = d->comm->exchange_counts(sl.counts.data(), sl.rcnt.data()))) break;
        sl.nR = sum(sl.rcnt);
        if ((rc = sl.rk.alloc(std::max<uint64_t>(sl.nR, 1) * 8 * nwr)) || (rc = sl.rw.alloc(std::max<uint64_t>(sl.nR, 1) * 4))) break;
        if (d->first_seen && (rc = sl.rp.alloc(std::max<uint64_t>(sl.nR, 1) * 16))) break;
        if (hipStreamWaitEvent(X, sl.parted, 0) != hipSuccess) { set_error("event wait failed"); rc = KATOME_E_DEVICE; break; }
        // (a slice is a quarter of a message on every rank: one round, nothing to agree on)
        if ((rc = d->xchg(xphase, sl.pk.p, sl.counts.data(), sl.rk.p, sl.rcnt.data(), 8 * nwr, X, true))) break;
        if ((rc = d->xchg(xphase, sl.pw.p, sl.counts.data(), sl.rw.p, sl.rcnt.data(), 4, X, true))) break;
        if (d->first_seen && (rc = d->xchg(xphase, sl.ppairs.p, sl.counts.data(), sl.rp.p, sl.rcnt.data(), 16, X, true))) break;
        if (hipEventRecord(sl.arrived, X) != hipSuccess) { set_error("event record failed"); rc = KATOME_E_DEVICE; break; }
        // (c) meanwhile: the slice before goes into the table
        if (j > 0) rc = insert_slice(slot[(j - 1) & 1]);
    }
    if (rc == KATOME_OK) rc = insert_slice(slot[(ns - 1) & 1]);
    (void)hipStreamSynchronize(X);
    (void)hipStreamSynchronize(stream);                       // (the buffers go back to the cache behind everything that used them)
    d->comm->use_stream(stream);
    return rc;
}

// The same exchange for records that already lie grouped by owner (records_to_edges_sorted with an OwnerSplit: owner p's records
// are keys[base[p] .. base[p] + count[p])): no partition pass; a round sends every owner the next stretch of its records.  The
// collectives are route_weighted's -- one agreement on the number of rounds, then counts + keys + weights per round --, so ranks
// that took different routes to their records still meet.
int route_owned(katome_dist_builder* d, int xphase, const DevBuf& keys, const DevBuf& weights, const OwnerSplit& sp, uint32_t nwr, hipStream_t stream,
                Collected& collect, uint64_t n_rec) {
    const int world = d->world();
    const uint64_t chunk = std::max<uint64_t>(1, d->comm->max_message_bytes / 4 / (8 * nwr));
    uint64_t ns = 0;
    for (int p = 0; p < world; ++p) ns = std::max(ns, (sp.count[p] + chunk - 1) / chunk);
    KCHECK(d->comm->allreduce(&ns, 1, OP_MAX));
    if (ns == 0) return KATOME_OK;
    if (!d->xstream) KCHECK_HIP(hipStreamCreateWithFlags(&d->xstream, hipStreamNonBlocking));
    hipStream_t X = d->xstream;
    struct Slot {
        DevBuf rk, rw;
        std::vector<uint64_t> counts, offs, rcnt;
        uint64_t nR = 0;
        hipEvent_t arrived = nullptr;
        explicit Slot(hipStream_t s) : rk(s), rw(s) {}
        ~Slot() { if (arrived) (void)hipEventDestroy(arrived); }
    };
    Slot slot[2] = {Slot(stream), Slot(stream)};
    hipEvent_t ready = nullptr;
    KCHECK_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    for (auto& sl : slot) KCHECK_HIP(hipEventCreateWithFlags(&sl.arrived, hipEventDisableTiming));
    d->comm->use_stream(X);
    static const bool tight = getenv("KATOME_SORTED_COUNT") && atoi(getenv("KATOME_SORTED_COUNT")) == 2;
    auto take_slice = [&](Slot& sl) -> int {
        KCHECK_HIP(hipStreamWaitEvent(stream, sl.arrived, 0));
        trace_words("kmers received: keys", d->rank(), sl.rk.p, sl.nR * nwr, stream);
        trace_words("kmers received: weights", d->rank(), sl.rw.p, sl.nR / 2, stream);
        if (sl.nR) KCHECK(collect.append(sl.rk.p, sl.rw.p, sl.nR, nwr, tight ? 0 : n_rec + n_rec / 8 + (1u << 20), stream));
        return KATOME_OK;
    };
    int rc = KATOME_OK;
    if (hipEventRecord(ready, stream) != hipSuccess || hipStreamWaitEvent(X, ready, 0) != hipSuccess) { set_error("event failed"); rc = KATOME_E_DEVICE; }
    for (uint64_t j = 0; j < ns && rc == KATOME_OK; ++j) {
        Slot& sl = slot[j & 1];
        sl.counts.assign(world, 0); sl.offs.assign(world, 0); sl.rcnt.assign(world, 0);
        for (int p = 0; p < world; ++p) {
            const uint64_t done = std::min(sp.count[p], j * chunk);
            sl.counts[p] = std::min(sp.count[p] - done, chunk); sl.offs[p] = sp.base[p] + done;
        }
        if ((rc = d->comm->exchange_counts(sl.counts.data(), sl.rcnt.data()))) break;
        sl.nR = sum(sl.rcnt);
        if ((rc = sl.rk.alloc(std::max<uint64_t>(sl.nR, 1) * 8 * nwr)) || (rc = sl.rw.alloc(std::max<uint64_t>(sl.nR, 1) * 4))) break;
        if ((rc = d->xchg(xphase, keys.p, sl.counts.data(), sl.rk.p, sl.rcnt.data(), 8 * nwr, X, true, katome_comm::MAX_UNKNOWN, sl.offs.data()))) break;
        if ((rc = d->xchg(xphase, weights.p, sl.counts.data(), sl.rw.p, sl.rcnt.data(), 4, X, true, katome_comm::MAX_UNKNOWN, sl.offs.data()))) break;
        if (hipEventRecord(sl.arrived, X) != hipSuccess) { set_error("event record failed"); rc = KATOME_E_DEVICE; break; }
        if (j > 0) rc = take_slice(slot[(j - 1) & 1]);
    }
    if (rc == KATOME_OK) rc = take_slice(slot[(ns - 1) & 1]);
    (void)hipStreamSynchronize(X);
    (void)hipStreamSynchronize(stream);
    (void)hipEventDestroy(ready);
    d->comm->use_stream(stream);
    return rc;
}

// may the k-mer records that arrive be kept and counted by sorting (katome_dev_edges' rule: one-word k-mers by packed key,
// nothing in the k-mer table yet)?
bool may_collect(const katome_dist_builder* d) {
    static const int sorted_count = getenv("KATOME_SORTED_COUNT") ? atoi(getenv("KATOME_SORTED_COUNT")) : 1;
    return sorted_count && !d->first_seen && d->nw == 1 && !d->b->table_ready;
}
// the collected records of this rank's k-mers -> its sorted edges (lds_count_kernel), or, out of that kernel's range, into the
// k-mer table after all.  No collective in here: every rank decides for itself.
int count_collected(katome_dist_builder* d, Collected& c, hipStream_t stream) {
    static const int sorted_count = getenv("KATOME_SORTED_COUNT") ? atoi(getenv("KATOME_SORTED_COUNT")) : 1;
    katome_builder* b = d->b;
    if (c.n == 0) return KATOME_OK;
    if ((c.n >= (1ull << 22) || sorted_count == 2) && (c.n >> 21) <= 2900) {
        uint64_t distinct = 0;
        int rc;
        {
            PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
            rc = records_to_edges_sorted(c.keys, c.weights, c.n, d->s.k, d->rc, b->prune_weight, b->edge_key, b->edge_weight, &b->n_edges, &distinct, stream);
        }
        if (rc == KATOME_OK) {
            b->stat_kmers = distinct; b->stat_kmer_slots = 0;
            c.keys.release(); c.weights.release();
            PhaseScope ps(b->prof, PH_SORT_EDGES, stream);
            KCHECK(dev_sort_bufs(b->edge_key, &b->edge_weight, b->n_edges, b->nw, 2 * d->s.k, stream, true));
            b->edges_ready = true;
            return KATOME_OK;
        }
        if (rc != KATOME_E_UNSUPPORTED) return rc;
    }
    return builder_insert(b, b->table, b->table_ready, d->nw, b->s.table_slots_hint, c.keys.as<u64>(), c.weights.as<u32>(), c.n, nullptr, PH_INSERT, stream);
}

// every rank's distinct tiles -> (k-mer, count[, sequence numbers]) records -> the k-mers' owners.  `span`: the plan all ranks
// agreed on (a rank may hold no tiles at all and still takes part in every exchange).
int expand_and_route_kmers(katome_dist_builder* d, uint32_t span, hipStream_t stream) {
    katome_builder* b = d->b;
    const int world = d->world();
    const uint32_t nw = d->nw, k = d->s.k;
    // Big tiles meet on their owners, but the mid tiles they are cut into are shared between big tiles that overlap (the same
    // stretch of genome tiled from another read start) and those live on other ranks: expanded rank by rank the mid level held
    // 2.4 x the mid tiles -- and sent 2.4 x the k-mer records -- of a one-rank build at 8 ranks (C3-like reads).  So the mid
    // tiles are routed to owners of their own and counted there, like the big tiles before them.
    const uint32_t span2 = mid_span(span);
    const bool route_mid = span2 != 0 && (world > 1 || getenv("KATOME_ROUTE_MID_TILES"));
    DevBuf keys(stream), weights(stream), seen(stream);         // the k-mer records this rank sends on
    uint64_t n_rec = 0;
    if (route_mid) {
        const uint32_t kk2 = k + span2 - 1, n_sub = span / span2, nw2 = (uint32_t)key_words_for_k(kk2);
        DevBuf mk(stream), mw(stream), ms(stream);
        uint64_t n_mid = 0;
        if (b->tiles_ready) {
            KCHECK(table_occupied(b->tiles, &b->stat_tiles, stream));
            b->stat_tile_slots = b->tiles.cap;
            {
                PhaseScope ps(b->prof, PH_EXPAND_MID, stream);
                KCHECK(table_expand_tiles_to_subtiles(b->tiles, kk2, n_sub, span2, d->rc, mk, mw, &n_mid, stream, d->first_seen ? &ms : nullptr));
            }
            b->tiles.release();
        }
        b->span = span; b->span2 = span2;
        // (a rank receives about what it sends, and about half of that is distinct; the table grows if it is not.  Twice the
        // slots doubled the time of the scan that expands them)
        const uint64_t mid_hint = std::max<uint64_t>(n_mid, 1u << 16);
        // (Counting what arrives here by sorting, as one GPU counts its tile levels, was measured at an eighth of C3 per rank: 5.9 + 6.2 ms
        // against the tables' 5.3 + 6.0 -- the sorted levels' fixed costs, a few host round trips each, eat their gain at 10^8 records.)
        KCHECK(route_weighted(d, X_MID_TILES, mk, mw, ms, n_mid, nw2, 0, 0, b->tiles2, b->tiles2_ready, mid_hint, PH_EXPAND_MID, stream));
        b->tiles_ready = b->tiles2_ready;                    // (what is left to expand, if anything arrived)
        if (b->tiles2_ready) { KCHECK(table_occupied(b->tiles2, &b->stat_tiles2, stream)); b->stat_tile2_slots = b->tiles2.cap; }
    }
    if (b->tiles_ready) {
        Table* last = &b->tiles2; uint32_t last_span = span2;
        if (!route_mid) {
            trace_words("tiles: slots", d->rank(), b->tiles.slots.p, b->tiles.cap * b->tiles.slot_bytes() / 8, stream);
            KCHECK(expand_to_last_level(b, &last, &last_span, stream));
        }
        trace_words("last level: slots", d->rank(), last->slots.p, last->cap * last->slot_bytes() / 8, stream);
        {
            PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
            if (d->first_seen) KCHECK(table_expand_tiles_to_records(*last, k, last_span, d->rc, keys, weights, &n_rec, stream, &seen));
            else KCHECK(table_tiles_to_records_fast(*last, k, last_span, d->rc, keys, weights, &n_rec, stream));      // (same records, any order)
        }
        trace_words("last level after: slots", d->rank(), last->slots.p, last->cap * last->slot_bytes() / 8, stream);
        b->tiles.release(); b->tiles2.release();
        b->tiles_ready = false; b->tiles2_ready = false;
    }
    if (getenv("KATOME_DIST_STATS"))
        fprintf(stderr, "[dist] rank %d of %d: %llu distinct tiles, %llu on the last level -> %llu k-mer records to route\n", d->rank(), world,
                (unsigned long long)b->stat_tiles, (unsigned long long)(b->span2 ? b->stat_tiles2 : b->stat_tiles), (unsigned long long)n_rec);
    trace_words("expand: record keys", d->rank(), keys.p, n_rec * nw, stream);
    trace_words("expand: record weights", d->rank(), weights.p, n_rec / 2, stream);
    if (may_collect(d)) {
        Collected got(stream);
        KCHECK(route_weighted(d, X_KMERS, keys, weights, seen, n_rec, nw, 2, k - 2, b->table, b->table_ready, b->s.table_slots_hint, PH_INSERT, stream, &got));
        keys.release(); weights.release();
        return count_collected(d, got, stream);
    }
    return route_weighted(d, X_KMERS, keys, weights, seen, n_rec, nw, 2, k - 2, b->table, b->table_ready, b->s.table_slots_hint, PH_INSERT, stream);
}

// Global rank of every value among the DISTINCT u64 values held by all ranks (sequence numbers): values are spread over the
// ranks by range -- splitters from the merged 2^16-bucket histogram, so every rank gets about the same number -- sorted
// there, and each value's rank (ranks before + position) travels back over the mirrored exchange.
int global_rank(katome_dist_builder* d, int xphase, const u64* vals, uint64_t n, uint64_t vmax, u64* out_rank, hipStream_t stream) {
    const int world = d->world(), rank = d->rank();
    constexpr uint64_t NB = 1u << 16;
    if (n >= (1ull << 32)) { set_error("more than 2^32 values to rank on one GPU"); return KATOME_E_UNSUPPORTED; }
    const uint64_t width = vmax / NB + 1;
    DevBuf hist(stream);
    KCHECK(hist.alloc(NB * 8));
    KCHECK_HIP(hipMemsetAsync(hist.p, 0, NB * 8, stream));
    if (n) KLAUNCH(value_hist_kernel, n, stream, vals, n, width, hist.as<unsigned long long>());
    KCHECK_HIP(hipGetLastError());
    std::vector<uint64_t> h(NB);
    KCHECK_HIP(hipMemcpyAsync(h.data(), hist.p, NB * 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    KCHECK(d->comm->allreduce(h.data(), NB, OP_SUM));
    uint64_t total = 0;
    for (uint64_t x : h) total += x;
    // bounds[p] = first value of part p + 1: the bucket where the running count passes (p + 1) / world of the total
    std::vector<uint64_t> bounds(std::max(world - 1, 1), ~0ull);
    {
        uint64_t run = 0; int p = 0;
        for (uint64_t bkt = 0; bkt < NB && p < world - 1; ++bkt) {
            while (p < world - 1 && run >= (total * (uint64_t)(p + 1) + world - 1) / world) bounds[p++] = bkt * width;
            run += h[bkt];
        }
    }
    DevBuf d_bounds(stream), idx(stream), pv(stream), pidx(stream);
    KCHECK(d_bounds.alloc(bounds.size() * 8));
    KCHECK_HIP(hipMemcpyAsync(d_bounds.p, bounds.data(), bounds.size() * 8, hipMemcpyHostToDevice, stream));
    std::vector<uint64_t> counts(world, 0), rcnt(world, 0);
    KCHECK(idx.alloc((n + 1) * 4)); KCHECK(pv.alloc((n + 1) * 8)); KCHECK(pidx.alloc((n + 1) * 4));
    KCHECK(dev_iota(idx.as<u32>(), n, stream));
    KCHECK(dev_partition_range(vals, idx.as<u32>(), n, d_bounds.as<u64>(), (uint32_t)world, pv.as<u64>(), pidx.as<u32>(), counts.data(), stream));
    KCHECK_HIP(hipStreamSynchronize(stream));               // (bounds was read from a host vector)
    uint64_t pair_max = 0;
    KCHECK(d->comm->exchange_counts(counts.data(), rcnt.data(), &pair_max));
    const uint64_t nR = sum(rcnt);
    if (nR >= (1ull << 32)) { set_error("more than 2^32 values to rank on one GPU"); return KATOME_E_UNSUPPORTED; }
    DevBuf rv(stream), pos(stream), ans(stream), back(stream);
    KCHECK(rv.alloc((nR + 1) * 8)); KCHECK(pos.alloc((nR + 1) * 4)); KCHECK(ans.alloc((nR + 1) * 8)); KCHECK(back.alloc((n + 1) * 8));
    KCHECK(d->xchg(xphase, pv.p, counts.data(), rv.p, rcnt.data(), 8, stream, false, pair_max));
    uint32_t bits = 1;
    while (bits < 64 && (vmax >> bits)) ++bits;
    KCHECK(dev_iota(pos.as<u32>(), nR, stream));
    KCHECK(dev_sort(rv.as<u64>(), pos.as<u32>(), nR, 1, bits, stream));
    std::vector<uint64_t> all(world, 0);
    KCHECK(d->comm->allgather(nR, all.data()));
    uint64_t base = 0;
    for (int p = 0; p < rank; ++p) base += all[p];
    if (nR) KLAUNCH(assign_rank_kernel, nR, stream, pos.as<u32>(), nR, base, ans.as<u64>());
    KCHECK_HIP(hipGetLastError());
    KCHECK(d->xchg(xphase, ans.p, rcnt.data(), back.p, counts.data(), 8, stream, false, pair_max));       // the mirrored route
    if (n) KLAUNCH(scatter_u64_kernel, n, stream, back.as<u64>(), pidx.as<u32>(), n, out_rank);
    KCHECK_HIP(hipGetLastError());
    KCHECK_HIP(hipStreamSynchronize(stream));
    return KATOME_OK;
}

int fill_graph(katome_dist_builder* d, katome_dist_graph* out) {
    if (!out) return KATOME_OK;
    katome_builder* b = d->b;
    memset(out, 0, sizeof *out);
    out->n_edges = d->n_edges; out->n_nodes = d->n_nodes; out->total_edges = d->total_edges; out->total_nodes = d->total_nodes;
    out->node_base = d->node_base; out->key_words = d->nw; out->label_stride = label_stride_for_k(d->s.k);
    out->d_edge_key = b->edge_key.as<u64>(); out->d_edge_weight = b->edge_weight.as<u32>();
    out->d_edge_src = d->edge_src.as<u64>(); out->d_edge_dst = d->edge_dst.as<u64>(); out->d_edge_label = d->edge_label.as<uint8_t>();
    out->d_node_key = d->node_key.as<u64>();
    out->d_edge_id = d->first_seen ? d->edge_gid.as<u64>() : nullptr;
    out->d_node_id = d->first_seen ? d->node_gid.as<u64>() : nullptr;
    out->d_edge_age = d->edge_age.p ? d->edge_age.as<u64>() : nullptr;
    return KATOME_OK;
}

}  // namespace

extern "C" {

void katome_shard_range(uint64_t total_reads, uint32_t world, uint32_t rank, uint64_t* first, uint64_t* count) {
    if (world == 0) world = 1;
    const uint64_t per = ((total_reads + world - 1) / world + 63) / 64 * 64;
    const uint64_t r0 = std::min(total_reads, (uint64_t)rank * per), r1 = std::min(total_reads, r0 + per);
    if (first) *first = r0;
    if (count) *count = r1 - r0;
}

uint32_t katome_dist_exchange_count(void) { return X_COUNT; }
const char* katome_dist_exchange_name(uint32_t phase) { return phase < X_COUNT ? XPHASE_NAMES[phase] : ""; }
int katome_dist_exchange_read(katome_dist_builder* d, uint64_t* out) {
    if (!d || !out) { set_error("null argument"); return KATOME_E_ARG; }
    if (!d->xevents.empty()) {
        KCHECK_HIP(hipSetDevice(d->s.device));
        KCHECK_HIP(hipDeviceSynchronize());
        for (auto& e : d->xevents) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) d->xstats[e.phase].ms += ms;
            (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
        }
        d->xevents.clear();
    }
    for (int i = 0; i < X_COUNT; ++i) {
        out[4 * i] = d->xstats[i].calls; out[4 * i + 1] = d->xstats[i].bytes_out; out[4 * i + 2] = d->xstats[i].max_pair_bytes;
        out[4 * i + 3] = (uint64_t)(d->xstats[i].ms * 1000.0);
        d->xstats[i] = katome::ExchangeStats();
    }
    return KATOME_OK;
}

int katome_dist_create(const katome_settings* s, katome_comm* comm, katome_dist_builder** out) {
    if (!s || !comm || !out) { set_error("null argument"); return KATOME_E_ARG; }
    *out = nullptr;
    if (comm->world() > KATOME_MAX_RANKS) { set_error("at most %d ranks", KATOME_MAX_RANKS); return KATOME_E_UNSUPPORTED; }
    katome_settings mine = *s;
    mine.n_devices = 1;
    mine.flags = (uint16_t)(s->flags & KATOME_FLAG_FIRST_SEEN_ORDER);      // (stages run after katome_dist_gather, on the root)
    // the hint is for the whole build; every rank owns about 1/world of the keys
    if (mine.table_slots_hint) mine.table_slots_hint = (uint64_t)((double)mine.table_slots_hint / comm->world() * 1.1) + 1024;
    katome_builder* b = nullptr;
    KCHECK(katome_builder_create(&mine, &b));
    katome_dist_builder* d = new (std::nothrow) katome_dist_builder();
    if (!d) { katome_builder_destroy(b); set_error("out of host memory"); return KATOME_E_OOM; }
    d->s = *s; d->comm = comm; d->b = b; d->nw = b->nw; d->rc = b->rc; d->first_seen = b->first_seen;
    // routes (KATOME_DIST_ROUTE=local|tiles|supermers overrides): few ranks -- every rank counts its own reads and
    // routes its distinct k-mers ("local"); from three ranks on -- tiles, mid tiles and k-mer records routed level by level ("tiles").
    // "supermers": the reads travel as supermers, ONCE, before anything is counted (by packed key, k <= 31) -- SURVEY 8(e)'s single
    // exchange, built and measured in round 4: an eighth of C3 per rank costs 65.7 ms of GPU time that way against 52.2 level by level
    // (a read makes 12.9 supermer records against 4 tiles; profiles/r04_share.md), so it is offered, not the default.
    d->local_first = comm->world() <= 2;
    d->want_supermers = false;
    if (const char* e = getenv("KATOME_DIST_ROUTE")) {
        d->local_first = strcmp(e, "local") == 0 ? true : (strcmp(e, "tiles") == 0 || strcmp(e, "supermers") == 0) ? false : d->local_first;
        d->want_supermers = strcmp(e, "supermers") == 0 && !d->first_seen;
    }
    if (d->want_supermers) d->local_first = false;
    *out = d;
    return KATOME_OK;
}

void katome_dist_destroy(katome_dist_builder* d) {
    if (!d) return;
    (void)hipSetDevice(d->s.device);
    katome_builder_destroy(d->b);
    delete d;
}

katome_builder* katome_dist_inner(katome_dist_builder* d) { return d ? d->b : nullptr; }
const char* katome_dist_route(const katome_dist_builder* d) {
    return !d ? "" : d->supermers ? "supermers" : d->local_first ? "local" : (d->planned || d->finalized) && !d->want_supermers ? "tiles" : d->want_supermers ? "supermers (if the reads allow)" : "tiles";
}

int katome_dist_current_graph(katome_dist_builder* d, katome_dist_graph* out) {
    if (!d) { set_error("null argument"); return KATOME_E_ARG; }
    if (!d->finalized) { set_error("katome_dist_current_graph: not finalized"); return KATOME_E_ARG; }
    return fill_graph(d, out);
}

int katome_dist_remove_weak_edges(katome_dist_builder* d, uint32_t threshold) {
    if (!d) { set_error("null argument"); return KATOME_E_ARG; }
    if (d->first_seen) { set_error("first-seen order: remove_weak_edges runs on the gathered graph (katome_dist_gather), with petgraph's numbering"); return KATOME_E_ARG; }
    return katome_dev_remove_weak_edges(d->b, threshold, nullptr);   // (before the edges are read out: only records the threshold)
}

int katome_dist_add_reads(katome_dist_builder* d, const uint8_t* d_packed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                          const uint8_t* d_skip, uint64_t batch_reads, void* stream_) {
    if (!d || (!d_packed && n_reads)) { set_error("null argument"); return KATOME_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    katome_builder* b = d->b;
    KCHECK_HIP(hipSetDevice(d->s.device));
    d->comm->use_stream(stream);
    if (d->finalized) { set_error("builder already finalized"); return KATOME_E_ARG; }
    const uint32_t k = d->s.k, nw = d->nw;
    if (read_len < k) { set_error("Read is too short!"); return KATOME_E_SHORT_READ; }       // pt_graph.rs:278
    if (!d->planned) {
        d->read_len = read_len; d->W = read_len - k + 1;
        if (!katome_tile_plan_limited(k, read_len, 3, &d->span, &d->tiles_per_read, &d->rest)) { d->span = 1; d->tiles_per_read = 0; d->rest = d->W; }
        d->nwt = d->span > 1 ? katome_tile_words(k, d->span) : nw;
        d->supermers = d->want_supermers && nw == 1 && supermer_route_takes(k, read_len, SUPERMER_M);
        if (d->want_supermers && !d->supermers) d->want_supermers = false;          // (these reads take the level-by-level route)
        if (d->supermers) { d->owner_m = SUPERMER_M; d->sm_slots = supermer_slots(k, read_len, SUPERMER_M); }
        d->planned = true;
    } else if (read_len != d->read_len) {
        set_error("the sharded build takes reads of one length (%u, then %u)", d->read_len, read_len);
        return KATOME_E_UNSUPPORTED;
    }
    const int world = d->world();
    if (const char* f = getenv("KATOME_DIST_ADD_FAIL"))          // (tests: this rank's reads are refused, as a failed allocation would refuse them)
        if (atoi(f) == d->comm->rank()) { set_error("reads refused (KATOME_DIST_ADD_FAIL)"); return KATOME_E_OOM; }
    if (d->supermers) {
        // the reads' supermer records wait for the one exchange (katome_dist_finalize): slots per read, then this call's spill region.
        // How many records spill is a property of the reads (short minimizer windows at small k cut a read into more runs than it has
        // slots); the kernel counts them all, so a region that proved too small is made the size it takes and the call's reads are cut
        // again -- never an error on one rank alone, which would leave the others waiting in the exchange
        uint64_t spill_cap = std::max<uint64_t>(1024, n_reads / 8), add = 0;
        DevBuf cursor(stream);
        KCHECK(cursor.alloc(8));
        for (int attempt = 0;; ++attempt) {
            add = n_reads * d->sm_slots + spill_cap;
            if (d->sm_n + add > d->sm_cap) {
                const uint64_t want = d->sm_n ? std::max(d->sm_n + add, d->sm_cap * 2) : add;
                DevBuf grown(stream);
                KCHECK(grown.alloc(want * 16 + 64));
                if (d->sm_n) KCHECK_HIP(hipMemcpyAsync(grown.p, d->sm_recs.p, d->sm_n * 16, hipMemcpyDeviceToDevice, stream));
                const size_t bytes = grown.bytes;
                d->sm_recs.stream = stream; d->sm_recs.adopt(grown.take(), bytes);
                d->sm_cap = want;
            }
            u64* slots_at = d->sm_recs.as<u64>() + d->sm_n * 2;
            u64* spill_at = slots_at + n_reads * d->sm_slots * 2;
            KCHECK_HIP(hipMemsetAsync(cursor.p, 0, 8, stream));
            KCHECK_HIP(hipMemsetAsync(spill_at, 0xFF, spill_cap * 16, stream));
            {
                PhaseScope ps(b->prof, PH_EXTRACT, stream);
                KCHECK(dev_supermers_extract(d_packed, n_reads, read_len, d_skip, k, d->owner_m, d->rc, (uint32_t)world, d->sm_slots, slots_at, spill_at, spill_cap,
                                             cursor.as<u64>(), stream));
            }
            uint64_t spilled = 0;
            KCHECK_HIP(hipMemcpyAsync(&spilled, cursor.p, 8, hipMemcpyDeviceToHost, stream));
            KCHECK_HIP(hipStreamSynchronize(stream));
            if (spilled <= spill_cap) break;
            if (attempt) { set_error("supermers: %llu records beyond the reads' slots on the second cut, %llu on the first", (unsigned long long)spilled, (unsigned long long)spill_cap); return KATOME_E_UNSUPPORTED; }
            spill_cap = spilled;
        }
        d->sm_n += add;
        d->reads_end = std::max(d->reads_end, first_read + n_reads);
        return KATOME_OK;
    }
    const bool tiled = d->span > 1;
    const uint32_t per_read = tiled ? d->tiles_per_read : d->W, nwr = tiled ? d->nwt : nw, stride = (read_len + 3) / 4;
    if (d->local_first) {
        // this rank's reads are counted here, exactly as a one-GPU build counts them (tiles + the windows after them; the
        // sequence numbers are those of the GLOBAL read order: first_read is this rank's offset in it) -- no exchange yet
        uint64_t batch = batch_reads ? batch_reads : (16ull << 20);
        batch = std::max<uint64_t>(64, batch / 64 * 64);
        const uint64_t cap_reads = std::min(batch, std::max<uint64_t>(n_reads, 1));
        const uint32_t first_rest = tiled ? d->tiles_per_read * d->span : 0;
        DevBuf recbuf(stream);
        KCHECK(recbuf.alloc(cap_reads * std::max<uint32_t>(per_read, d->rest) * 8 * nwr + 64));
        for (uint64_t r0 = 0; r0 < n_reads; r0 += batch) {
            const uint64_t nr = std::min(batch, n_reads - r0);
            const uint8_t* p = d_packed + r0 * stride;
            const uint8_t* sk = d_skip ? d_skip + r0 : nullptr;
            SeenOrigin origin;
            origin.read0 = first_read + r0; origin.windows = d->W; origin.rc = d->rc;
            {
                PhaseScope ps(b->prof, PH_EXTRACT, stream);
                KCHECK(launch_extract_fixed(k, d->rc, p, nr, read_len, sk, recbuf.as<u64>(), stream, tiled ? d->span : 1, d->first_seen && d->rc));
            }
            origin.per_read = per_read; origin.span = tiled ? d->span : 1; origin.win0 = 0;
            if (tiled) {
                b->span = d->span;
                // (one GPU's rule: two-word tiles of one-word k-mers by packed key wait as records and are counted by sorting at the end)
                bool kept = false;
                if (!d->first_seen && tile_recs_shape(d->nwt, nw, false) && !b->tiles_ready && !b->tile_recs_closed && sorted_count_mode() && sorted_tiles_mode() == 2) {
                    PhaseScope ps(b->prof, PH_INSERT_TILES, stream);
                    KCHECK(keep_tile_recs(b, recbuf.as<u64>(), nr * per_read, d->nwt, &kept, stream));
                }
                if (!kept)
                KCHECK(builder_insert(b, b->tiles, b->tiles_ready, d->nwt, b->s.table_slots_hint / 4, recbuf.as<u64>(), nullptr, nr * per_read,
                                      d->first_seen ? &origin : nullptr, PH_INSERT_TILES, stream));
            } else {
                KCHECK(builder_insert(b, b->table, b->table_ready, nw, b->s.table_slots_hint, recbuf.as<u64>(), nullptr, nr * per_read,
                                      d->first_seen ? &origin : nullptr, PH_INSERT, stream));
            }
            if (tiled && d->rest) {                                // the windows after the last whole tile of every read
                {
                    PhaseScope ps(b->prof, PH_EXTRACT, stream);
                    KCHECK(launch_extract_fixed(k, d->rc, p, nr, read_len, sk, recbuf.as<u64>(), stream, 1, d->first_seen && d->rc, first_rest, d->rest));
                }
                origin.per_read = d->rest; origin.span = 1; origin.win0 = first_rest;
                KCHECK(builder_insert(b, b->table, b->table_ready, nw, b->s.table_slots_hint, recbuf.as<u64>(), nullptr, nr * d->rest,
                                      d->first_seen ? &origin : nullptr, PH_INSERT, stream));
            }
            KCHECK_HIP(hipStreamSynchronize(stream));
        }
        d->reads_end = std::max(d->reads_end, first_read + n_reads);
        return KATOME_OK;
    }
    // tile records are small (a few per read): large batches mean few exchange rounds
    uint64_t batch = batch_reads ? batch_reads : (tiled ? (16ull << 20) : (4ull << 20));
    batch = std::max<uint64_t>(64, batch / 64 * 64);
    uint64_t nb = (n_reads + batch - 1) / batch;
    const uint64_t cap_reads = std::min(batch, std::max<uint64_t>(n_reads, 1));
    const uint32_t first_rest = tiled ? d->tiles_per_read * d->span : 0;
    DevBuf recbuf(stream), part(stream), idx(stream), pidx(stream);
    const uint64_t cap_rec = cap_reads * std::max<uint32_t>(per_read, d->rest);
    // the batch buffers first, and whether every rank got them travels with the number of rounds: a rank that could not must not
    // leave the others waiting in the first exchange (every rank takes part in every exchange)
    int mine = recbuf.alloc(cap_rec * 8 * nwr + 64);
    if (!mine) mine = part.alloc(cap_rec * 8 * nwr + 64);
    if (!mine && d->first_seen) { mine = idx.alloc(cap_rec * 4 + 64); if (!mine) mine = pidx.alloc(cap_rec * 4 + 64); }
    uint64_t agree[2] = {nb, (uint64_t)(mine != KATOME_OK)};
    KCHECK(d->comm->allreduce(agree, 2, OP_MAX));
    nb = agree[0];
    if (mine) return mine;
    if (agree[1]) { set_error("another rank of this build could not allocate its batch buffers"); return KATOME_E_OOM; }
    for (uint64_t i = 0; i < nb; ++i) {
        const uint64_t r0 = std::min(n_reads, i * batch), nr = std::min(batch, n_reads - r0);
        const uint8_t* p = d_packed ? d_packed + r0 * stride : nullptr;
        const uint8_t* sk = d_skip ? d_skip + r0 : nullptr;
        std::vector<uint64_t> counts(world, 0);
        if (nr) {
            {
                PhaseScope ps(b->prof, PH_EXTRACT, stream);
                KCHECK(launch_extract_fixed(k, d->rc, p, nr, read_len, sk, recbuf.as<u64>(), stream, tiled ? d->span : 1, d->first_seen && d->rc));
            }
            const uint64_t n_rec = nr * per_read;
            if (d->first_seen) KCHECK(dev_iota(idx.as<u32>(), n_rec, stream));
            PhaseScope ps(b->prof, PH_REGION_ORDER, stream);       // ("region_order" doubles as the routing pass here)
            // a tile's owner may be any function of the tile -- identical tiles only have to meet on one rank; a k-mer's owner
            // is its canonical middle
            KCHECK(dev_partition(recbuf.as<u64>(), d->first_seen ? idx.as<u32>() : nullptr, n_rec, nwr, world, part.as<u64>(),
                                 d->first_seen ? pidx.as<u32>() : nullptr, counts.data(), stream, tiled ? 0 : 2, tiled ? 0 : k - 2));
        }
        KCHECK(route_and_insert(d, part.as<u64>(), pidx.as<u32>(), counts, nwr, tiled, first_read + r0, per_read, 0, tiled ? d->span : 1, stream));
        if (tiled && d->rest) {                                    // the windows after the last whole tile of every read
            std::vector<uint64_t> rcounts(world, 0);
            if (nr) {
                {
                    PhaseScope ps(b->prof, PH_EXTRACT, stream);
                    KCHECK(launch_extract_fixed(k, d->rc, p, nr, read_len, sk, recbuf.as<u64>(), stream, 1, d->first_seen && d->rc, first_rest, d->rest));
                }
                const uint64_t n_rec = nr * d->rest;
                if (d->first_seen) KCHECK(dev_iota(idx.as<u32>(), n_rec, stream));
                PhaseScope ps(b->prof, PH_REGION_ORDER, stream);
                KCHECK(dev_partition(recbuf.as<u64>(), d->first_seen ? idx.as<u32>() : nullptr, n_rec, nw, world, part.as<u64>(),
                                     d->first_seen ? pidx.as<u32>() : nullptr, rcounts.data(), stream, 2, k - 2));
            }
            KCHECK(route_and_insert(d, part.as<u64>(), pidx.as<u32>(), rcounts, nw, false, first_read + r0, d->rest, first_rest, 1, stream));
        }
        KCHECK_HIP(hipStreamSynchronize(stream));
    }
    d->reads_end = std::max(d->reads_end, first_read + n_reads);
    return KATOME_OK;
}

int katome_dist_finalize(katome_dist_builder* d, katome_dist_graph* out, void* stream_) {
    if (!d) { set_error("null argument"); return KATOME_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    katome_builder* b = d->b;
    KCHECK_HIP(hipSetDevice(d->s.device));
    d->comm->use_stream(stream);
    if (d->finalized) return fill_graph(d, out);
    const int world = d->world(), rank = d->rank();
    const uint32_t nw = d->nw, k = d->s.k, node_bits = 2 * (k - 1);
    // agree on the plan (a rank that was given no reads has none) and on the extent of the input
    uint64_t plan[2] = {d->planned ? d->span : 0, d->reads_end};
    KCHECK(d->comm->allreduce(plan, 2, OP_MAX));
    const bool tiled = plan[0] > 1;
    const uint64_t total_reads = plan[1];
    {   // (a rank that was given no reads made no plan: it takes the route of those that did)
        uint64_t route = d->planned ? (d->supermers ? 2 : 1) : 0;
        KCHECK(d->comm->allreduce(&route, 1, OP_MAX));
        if (!d->planned && route == 2) { d->supermers = true; d->owner_m = SUPERMER_M; }
        if (d->planned && d->supermers != (route == 2)) { set_error("the ranks of a sharded build took different routes"); return KATOME_E_ARG; }
    }
    if (d->supermers) {
        // ---- the one exchange: supermer records to their owners, then this rank counts what it received ------------------------------
        DevBuf part(stream), recv(stream);
        std::vector<uint64_t> counts(world, 0), rcnt(world, 0);
        KCHECK(part.alloc((d->sm_n + 1) * 16));
        {
            PhaseScope ps(b->prof, PH_REGION_ORDER, stream);          // ("region_order" doubles as the routing pass, as on the other routes)
            KCHECK(dev_partition_supermers(d->sm_recs.as<u64>(), d->sm_n, (uint32_t)world, part.as<u64>(), counts.data(), stream));
        }
        d->sm_recs.release(); d->sm_n = d->sm_cap = 0;
        uint64_t pair_max = 0;
        KCHECK(d->comm->exchange_counts(counts.data(), rcnt.data(), &pair_max));
        const uint64_t nR = sum(rcnt);
        KCHECK(recv.alloc((nR + 1) * 16));
        KCHECK(d->xchg(X_RECORDS, part.p, counts.data(), recv.p, rcnt.data(), 16, stream, false, pair_max));
        KCHECK_HIP(hipStreamSynchronize(stream));
        part.release();
        if (getenv("KATOME_DIST_STATS"))
            fprintf(stderr, "[dist] rank %d of %d: %llu supermer records sent, %llu received\n", rank, world, (unsigned long long)sum(counts), (unsigned long long)nR);
        // distinct supermers with their counts (as one GPU counts its tiles: two hash passes, counted in LDS)
        DevBuf lk(stream), lw(stream);
        uint64_t n1 = 0, d1 = 0;
        int rc = KATOME_E_UNSUPPORTED;
        if (nR) {
            PhaseScope ps(b->prof, PH_INSERT_TILES, stream);
            TileLevelScope tl;
            DevBuf ones(stream);                                      // (stays empty: every record counts once)
            rc = records_to_edges_sorted(recv, ones, nR, 40 /* any k of two words: the records are opaque 128-bit keys here */, false, 0, lk, lw, &n1, &d1, stream);
            if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
            if (rc == KATOME_E_UNSUPPORTED) {                         // (out of the LDS count's range: every record on its own, once)
                const size_t bytes = recv.bytes;
                lk.stream = stream; lk.adopt(recv.take(), bytes);
                KCHECK(lw.alloc((nR + 1) * 4));
                KCHECK(dev_fill_u32(lw.as<u32>(), nR, 1u, stream));
                n1 = nR;
            }
        }
        recv.release();
        b->stat_tiles = n1; b->stat_tile_slots = 0; b->span = 0; b->span2 = 0;
        Collected got(stream);
        {
            PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
            KCHECK(dev_supermers_expand(lk.as<u64>(), lw.as<u32>(), n1, k, d->rc, got.keys, got.weights, &got.n, stream));
        }
        got.cap = got.n;
        lk.release(); lw.release();
        KCHECK(count_collected(d, got, stream));
    } else
    if (d->local_first) {
        // every rank finishes its own counting; its DISTINCT k-mers (count, earliest sequence numbers) go to their owners, which
        // add them up in a fresh table
        DevBuf keys(stream), weights(stream), pairs(stream);
        uint64_t n_rec = 0;
        OwnerSplit split;
        bool split_used = false;
        static const int sorted_count = getenv("KATOME_SORTED_COUNT") ? atoi(getenv("KATOME_SORTED_COUNT")) : 1;
        // (owner split: grouped by the hash that names the owner, every group's keys written into its owner's stretch: no partition pass
        // before the exchange; KATOME_DIST_OWNER_SPLIT=0: by the whole k-mer's hash, then route_weighted's partition)
        static const bool owner_split_on = !getenv("KATOME_DIST_OWNER_SPLIT") || atoi(getenv("KATOME_DIST_OWNER_SPLIT")) != 0;
        if (b->tile_recs_n && (!may_collect(d) || (b->tile_recs_n * b->span < (1ull << 22) && sorted_count != 2))) KCHECK(flush_tile_recs(b, stream));
        if (b->tile_recs_n) {
            // the tiles kept as records: every level by sorting, down to this rank's distinct k-mers
            DevBuf rk(stream), rw(stream);
            uint64_t n_win = 0, distinct = 0;
            int rc = tile_recs_to_kmer_records(b, rk, rw, &n_win, 0, stream);
            if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
            if (rc == KATOME_OK) {
                PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
                if (owner_split_on) { split.n_parts = (uint32_t)world; split.core_shift = 2; split.core_bases = k - 2; split_used = true; }
                rc = sorted_fail("last") ? KATOME_E_UNSUPPORTED
                    : records_to_edges_sorted(rk, rw, n_win, k, false, 0, keys, weights, &n_rec, &distinct, stream, split_used ? &split : nullptr);
                if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
                if (rc == KATOME_E_UNSUPPORTED) {          // (beyond the LDS route: the records, counts and all, into the k-mer table)
                    split_used = false; n_rec = 0;
                    KCHECK(builder_insert(b, b->table, b->table_ready, nw, b->s.table_slots_hint, rk.as<u64>(), rw.as<u32>(), n_win, nullptr, PH_INSERT, stream));
                }
            }
        }
        if (b->tiles_ready && may_collect(d)) {
            // the rank's own distinct k-mers by sorting (as katome_dev_edges does, but one record per canonical k-mer: no strands yet)
            uint64_t n_tiles = 0;
            KCHECK(table_occupied(b->tiles, &n_tiles, stream));
            const uint64_t bound = n_tiles * b->span;
            if ((bound >= (1ull << 22) || (sorted_count == 2 && bound)) && (bound >> 21) <= 2900) {
                Table* last = nullptr; uint32_t last_span = 1;
                KCHECK(expand_to_last_level(b, &last, &last_span, stream));
                DevBuf rk(stream), rw(stream);
                uint64_t n_win = 0, distinct = 0;
                PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
                KCHECK(table_tiles_to_records_fast(*last, k, last_span, d->rc, rk, rw, &n_win, stream));
                b->tiles.release(); b->tiles2.release();
                b->tiles_ready = false; b->tiles2_ready = false;
                if (owner_split_on) { split.n_parts = (uint32_t)world; split.core_shift = 2; split.core_bases = k - 2; split_used = true; }
                KCHECK(records_to_edges_sorted(rk, rw, n_win, k, false, 0, keys, weights, &n_rec, &distinct, stream, split_used ? &split : nullptr));
            }
        }
        if (b->tiles_ready) KCHECK(expand_tiles(b, stream));
        if (b->table_ready) {
            PhaseScope ps(b->prof, PH_EMIT_EDGES, stream);
            KCHECK(table_to_records(b->table, keys, weights, &n_rec, stream, d->first_seen ? &pairs : nullptr));
        }
        b->table.release(); b->table_ready = false;
        if (getenv("KATOME_DIST_STATS"))
            fprintf(stderr, "[dist] rank %d of %d: %llu distinct k-mers of its own reads to route\n", rank, world, (unsigned long long)n_rec);
        if (may_collect(d)) {
            Collected got(stream);
            if (split_used) KCHECK(route_owned(d, X_KMERS, keys, weights, split, nw, stream, got, n_rec));
            else KCHECK(route_weighted(d, X_KMERS, keys, weights, pairs, n_rec, nw, 2, k - 2, b->table, b->table_ready, b->s.table_slots_hint, PH_INSERT, stream, &got));
            keys.release(); weights.release();
            KCHECK(count_collected(d, got, stream));
        } else KCHECK(route_weighted(d, X_KMERS, keys, weights, pairs, n_rec, nw, 2, k - 2, b->table, b->table_ready, b->s.table_slots_hint, PH_INSERT, stream));
    } else if (tiled) KCHECK(expand_and_route_kmers(d, (uint32_t)plan[0], stream));
    KCHECK(katome_dev_edges(b, nullptr, nullptr, nullptr, stream));       // this rank's distinct oriented edges, ascending (+ edge_seq)
    const uint64_t E = b->n_edges;
    if (E >= (1ull << 32)) { set_error("more than 2^32 edges on one rank"); return KATOME_E_UNSUPPORTED; }
    const u64* keys = b->edge_key.as<u64>();
    const u64* seq = d->first_seen ? b->edge_seq.as<u64>() : nullptr;
    trace_words("edges: keys", rank, keys, E * nw, stream);
    if (seq) trace_words("edges: seq", rank, seq, E, stream);
    // every edge asks the owner of its target (k-1)-mer for the target's id, remembering which edge asked
    DevBuf T(stream), origin(stream), P(stream), porigin(stream), tv(stream);
    std::vector<uint64_t> counts(world, 0), rcnt(world, 0);
    KCHECK(T.alloc((E + 1) * 8 * nw)); KCHECK(origin.alloc((E + 1) * 4)); KCHECK(P.alloc((E + 1) * 8 * nw)); KCHECK(porigin.alloc((E + 1) * 4));
    if (E) {
        KCHECK(dev_endpoints(keys, E, k, nullptr, T.as<u64>(), stream));
        KCHECK(dev_iota(origin.as<u32>(), E, stream));
        KCHECK(dev_partition(T.as<u64>(), origin.as<u32>(), E, nw, world, P.as<u64>(), porigin.as<u32>(), counts.data(), stream, 0, k - 2, d->owner_m));
    }
    T.release(); origin.release();
    uint64_t pair_max = 0;
    KCHECK(d->comm->exchange_counts(counts.data(), rcnt.data(), &pair_max));
    const uint64_t nR = sum(rcnt);
    if (nR >= (1ull << 32)) { set_error("more than 2^32 target look-ups on one rank"); return KATOME_E_UNSUPPORTED; }
    DevBuf R(stream), rv(stream);
    KCHECK(R.alloc((nR + 1) * 8 * nw));
    KCHECK(d->xchg(X_TARGETS, P.p, counts.data(), R.p, rcnt.data(), 8 * nw, stream, false, pair_max));
    if (d->first_seen) {                                     // ... and tells it when it first touched the target (2 * seq + 1)
        KCHECK(tv.alloc((E + 1) * 8)); KCHECK(rv.alloc((nR + 1) * 8));
        if (E) KLAUNCH(target_value_kernel, E, stream, seq, porigin.as<u32>(), E, tv.as<u64>());
        KCHECK_HIP(hipGetLastError());
        KCHECK(d->xchg(X_TARGETS, tv.p, counts.data(), rv.p, rcnt.data(), 8, stream, false, pair_max));
        tv.release();
    }
    P.release();
    // this rank's nodes with out-edges are the sources of its own sorted edges: no sort, no exchange
    DevBuf S(stream), lsrc(stream);
    uint64_t n_src = 0;
    KCHECK(lsrc.alloc((E + 1) * 8));
    if (E) KCHECK(dev_source_ids(keys, E, k, S, lsrc.as<u64>(), &n_src, stream));
    // answer: position among the sources or -- for a node without out-edges -- among the other nodes owned here
    DevBuf local(stream), sinks(stream);
    uint64_t n_sinks = 0;
    KCHECK(local.alloc((nR + 1) * 8));
    if (nR) {
        if (n_src) KCHECK(dev_rank(S.as<u64>(), n_src, nw, node_bits, R.as<u64>(), nR, local.as<u64>(), stream));
        else KCHECK_HIP(hipMemsetAsync(local.p, 0xFF, nR * 8, stream));
        DevBuf mk(stream), mpos(stream), cursor(stream), mrank(stream);
        KCHECK(cursor.alloc(8));
        KCHECK_HIP(hipMemsetAsync(cursor.p, 0, 8, stream));
        KCHECK(mk.alloc((nR + 1) * 8 * nw)); KCHECK(mpos.alloc((nR + 1) * 4));
        const dim3 mgrid(grid_for(nR, BLOCK * MISS_ITEMS, 256u * 16u));
        if (nw == 1) hipLaunchKernelGGL(compact_missing_kernel<1>, mgrid, dim3(BLOCK), 0, stream, local.as<u64>(), R.as<u64>(), nR, mk.as<u64>(), mpos.as<u32>(), cursor.as<u64>());
        else         hipLaunchKernelGGL(compact_missing_kernel<2>, mgrid, dim3(BLOCK), 0, stream, local.as<u64>(), R.as<u64>(), nR, mk.as<u64>(), mpos.as<u32>(), cursor.as<u64>());
        KCHECK_HIP(hipGetLastError());
        uint64_t m = 0;
        KCHECK_HIP(hipMemcpyAsync(&m, cursor.p, 8, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        if (m) {
            KCHECK(sinks.alloc(m * 8 * nw)); KCHECK(mrank.alloc(m * 8));
            KCHECK_HIP(hipMemcpyAsync(sinks.p, mk.p, m * 8 * nw, hipMemcpyDeviceToDevice, stream));
            KCHECK(dev_sort(sinks.as<u64>(), nullptr, m, nw, node_bits, stream));
            KCHECK(dev_unique(sinks.as<u64>(), m, nw, &n_sinks, stream));
            KCHECK(dev_rank(sinks.as<u64>(), n_sinks, nw, node_bits, mk.as<u64>(), m, mrank.as<u64>(), stream));
            KLAUNCH(fill_missing_kernel, m, stream, local.as<u64>(), mpos.as<u32>(), mrank.as<u64>(), m, n_src);
            KCHECK_HIP(hipGetLastError());
        }
    }
    R.release();
    const uint64_t n_owned = n_src + n_sinks;
    std::vector<uint64_t> all(world, 0);
    KCHECK(d->comm->allgather(n_owned, all.data()));
    uint64_t base = 0, total_nodes = 0;
    for (int p = 0; p < world; ++p) { if (p < rank) base += all[p]; total_nodes += all[p]; }
    // node keys: the sources ascending, then the others ascending
    KCHECK(d->node_key.alloc((n_owned + 1) * 8 * nw, stream));
    if (n_src) KCHECK_HIP(hipMemcpyAsync(d->node_key.p, S.p, n_src * 8 * nw, hipMemcpyDeviceToDevice, stream));
    if (n_sinks) KCHECK_HIP(hipMemcpyAsync(d->node_key.as<u64>() + n_src * nw, sinks.p, n_sinks * 8 * nw, hipMemcpyDeviceToDevice, stream));
    S.release(); sinks.release();
    const u64* id_map = nullptr;
    if (d->first_seen) {
        // the reference numbers a node when its first edge is added (add_fasta_node, pt_graph.rs:142-154): source of the
        // edge's first insertion at 2 * seq, target at 2 * seq + 1; the node's index is the rank of the earliest such number
        const uint64_t max_seq = 2 * (total_reads + 1) * 2 * (uint64_t)std::max<uint32_t>(d->W, 1) + 2;
        DevBuf node_first(stream);
        KCHECK(node_first.alloc((n_owned + 1) * 8));
        KCHECK_HIP(hipMemsetAsync(node_first.p, 0xFF, n_owned * 8, stream));
        if (E) KLAUNCH(src_first_kernel, E, stream, lsrc.as<u64>(), seq, E, node_first.as<u64>());
        if (nR) KLAUNCH(dst_first_kernel, nR, stream, local.as<u64>(), rv.as<u64>(), nR, node_first.as<u64>());
        KCHECK_HIP(hipGetLastError());
        KCHECK(d->node_gid.alloc((n_owned + 1) * 8, stream));
        KCHECK(global_rank(d, X_RANK_NODES, node_first.as<u64>(), n_owned, 2 * max_seq + 2, d->node_gid.as<u64>(), stream));
        id_map = d->node_gid.as<u64>();
        base = 0;
    }
    rv.release();
    // ids travel back over the mirrored exchange and are put where their edges are
    DevBuf ans(stream), back(stream);
    KCHECK(ans.alloc((nR + 1) * 8)); KCHECK(back.alloc((E + 1) * 8));
    if (nR) KLAUNCH(map_ids_kernel, nR, stream, local.as<u64>(), nR, id_map, base, ans.as<u64>());
    KCHECK_HIP(hipGetLastError());
    KCHECK(d->xchg(X_IDS, ans.p, rcnt.data(), back.p, counts.data(), 8, stream, false, pair_max));
    KCHECK(d->edge_src.alloc((E + 1) * 8, stream)); KCHECK(d->edge_dst.alloc((E + 1) * 8, stream));
    if (E) {
        KLAUNCH(scatter_u64_kernel, E, stream, back.as<u64>(), porigin.as<u32>(), E, d->edge_dst.as<u64>());
        KLAUNCH(map_ids_kernel, E, stream, lsrc.as<u64>(), E, id_map, base, d->edge_src.as<u64>());
    }
    KCHECK_HIP(hipGetLastError());
    if (d->first_seen) {
        // for the stages that run on the sharded graph (dist_prune.hip): where every edge's target lives -- owner rank (the
        // segment its look-up travelled in) and the node's local index there (the owner's answer, unmapped)
        KCHECK(d->xchg(X_IDS, local.p, rcnt.data(), back.p, counts.data(), 8, stream, false, pair_max));
        KCHECK(d->edge_dlocal.alloc((E + 1) * 8, stream)); KCHECK(d->edge_drank.alloc((E + 1) * 8, stream));
        if (E) KLAUNCH(scatter_u64_kernel, E, stream, back.as<u64>(), porigin.as<u32>(), E, d->edge_dlocal.as<u64>());
        uint64_t off = 0;
        for (int p = 0; p < world; ++p) {
            if (counts[p]) KLAUNCH(scatter_const_kernel, counts[p], stream, porigin.as<u32>(), off, off + counts[p], (u64)p, d->edge_drank.as<u64>());
            off += counts[p];
        }
        KCHECK_HIP(hipGetLastError());
        { const size_t bytes = lsrc.bytes; d->edge_lsrc.stream = stream; d->edge_lsrc.adopt(lsrc.take(), bytes); }
        d->n_src = n_src;
    }
    if (d->first_seen) {                                     // petgraph edge index = rank of the edge's first insertion (pt_graph.rs:194)
        const uint64_t max_seq = 2 * (total_reads + 1) * 2 * (uint64_t)std::max<uint32_t>(d->W, 1) + 2;
        KCHECK(d->edge_gid.alloc((E + 1) * 8, stream));
        KCHECK(global_rank(d, X_RANK_EDGES, seq, E, max_seq, d->edge_gid.as<u64>(), stream));
    }
    const uint32_t lstride = label_stride_for_k(k);
    KCHECK(d->edge_label.alloc((E + 1) * (size_t)lstride + 16, stream));
    {
        PhaseScope ps(b->prof, PH_LABELS, stream);
        KCHECK(dev_labels(keys, E, k, d->edge_label.as<uint8_t>(), stream));
    }
    trace_words("finalize end: keys", rank, keys, E * nw, stream);
    trace_words("finalize end: src", rank, d->edge_src.p, E, stream);
    trace_words("finalize end: dst", rank, d->edge_dst.p, E, stream);
    if (d->first_seen) { trace_words("finalize end: edge ids", rank, d->edge_gid.p, E, stream); trace_words("finalize end: node ids", rank, d->node_gid.p, n_owned, stream); }
    uint64_t tot = E;
    KCHECK(d->comm->allreduce(&tot, 1, OP_SUM));
    KCHECK_HIP(hipStreamSynchronize(stream));
    d->n_edges = E; d->n_nodes = n_owned; d->total_edges = tot; d->total_nodes = total_nodes; d->node_base = d->first_seen ? 0 : base;
    d->finalized = true;
    return fill_graph(d, out);
}

int katome_dist_gather(katome_dist_builder* d, int root, katome_builder** root_builder, void* stream_) {
    if (!d) { set_error("null argument"); return KATOME_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    katome_builder* b = d->b;
    KCHECK_HIP(hipSetDevice(d->s.device));
    d->comm->use_stream(stream);
    if (root_builder) *root_builder = nullptr;
    if (!d->finalized || !d->first_seen) { set_error("katome_dist_gather: a finalized FIRST_SEEN_ORDER build only (its indices place the edges)"); return KATOME_E_ARG; }
    if (d->gathered) { set_error("katome_dist_gather: the ranks' shares were gathered already"); return KATOME_E_ARG; }
    const int world = d->world(), rank = d->rank();
    if (root < 0 || root >= world) { set_error("root out of range"); return KATOME_E_ARG; }
    const uint32_t nw = d->nw;
    const uint64_t E = d->n_edges, N = d->n_nodes, TE = d->total_edges, TN = d->total_nodes;
    if (TE >= 0xFFFFFFFFull || TN >= 0xFFFFFFFFull) { set_error("katome_dist_gather: the whole graph (%llu edges) does not fit one GPU's 2^32 indices", (unsigned long long)TE); return KATOME_E_UNSUPPORTED; }
    std::vector<uint64_t> ecnt(world, 0), ncnt(world, 0), ercnt(world, 0), nrcnt(world, 0);
    ecnt[root] = E; ncnt[root] = N;
    KCHECK(d->comm->exchange_counts(ecnt.data(), ercnt.data()));
    KCHECK(d->comm->exchange_counts(ncnt.data(), nrcnt.data()));
    const bool me = rank == root;
    const uint64_t rE = me ? TE : 0, rN = me ? TN : 0;
    // this rank's share moves out of its builder; the root's builder then receives the whole graph
    DevBuf l_key(stream), l_w(stream);
    { const size_t n = b->edge_key.bytes; l_key.adopt(b->edge_key.take(), n); }
    { const size_t n = b->edge_weight.bytes; l_w.adopt(b->edge_weight.take(), n); }
    b->edge_seq.release();
    DevBuf g_gid(stream), tmp(stream);
    KCHECK(g_gid.alloc((rE + 1) * 8));
    KCHECK(d->xchg(X_GATHER, d->edge_gid.p, ecnt.data(), g_gid.p, ercnt.data(), 8, stream));
    auto bring = [&](const void* mine, size_t elem, void** landed) -> int {
        KCHECK(tmp.alloc((rE + 1) * elem));
        KCHECK(d->xchg(X_GATHER, mine, ecnt.data(), tmp.p, ercnt.data(), elem, stream));
        *landed = tmp.p;
        return KATOME_OK;
    };
    void* in = nullptr;
    trace_words("gather: my keys", rank, l_key.p, E * nw, stream);
    trace_words("gather: ids at root", rank, g_gid.p, rE, stream);
    KCHECK(bring(l_key.p, 8 * nw, &in));
    trace_words("gather: keys at root", rank, in, rE * nw, stream);
    if (me) {
        KCHECK(b->edge_key.alloc((TE + 1) * 8 * nw, stream));
        if (TE) { if (nw == 1) KLAUNCH(place_keys_kernel<1>, TE, stream, g_gid.as<u64>(), (const u64*)in, TE, b->edge_key.as<u64>());
                  else         KLAUNCH(place_keys_kernel<2>, TE, stream, g_gid.as<u64>(), (const u64*)in, TE, b->edge_key.as<u64>()); }
    }
    l_key.release();
    KCHECK(bring(l_w.p, 4, &in));
    if (me) {
        KCHECK(b->edge_weight.alloc((TE + 1) * 4, stream));
        if (TE) KLAUNCH(place_kernel<u32>, TE, stream, g_gid.as<u64>(), (const u32*)in, TE, b->edge_weight.as<u32>());
    }
    l_w.release();
    KCHECK(bring(d->edge_src.p, 8, &in));
    if (me) {
        KCHECK(b->edge_src.alloc((TE + 1) * 8, stream));
        if (TE) KLAUNCH(place_kernel<u64>, TE, stream, g_gid.as<u64>(), (const u64*)in, TE, b->edge_src.as<u64>());
    }
    KCHECK(bring(d->edge_dst.p, 8, &in));
    if (me) {
        KCHECK(b->edge_dst.alloc((TE + 1) * 8, stream));
        if (TE) KLAUNCH(place_kernel<u64>, TE, stream, g_gid.as<u64>(), (const u64*)in, TE, b->edge_dst.as<u64>());
    }
    KCHECK_HIP(hipGetLastError());
    // nodes
    DevBuf n_gid(stream), n_key(stream);
    KCHECK(n_gid.alloc((rN + 1) * 8)); KCHECK(n_key.alloc((rN + 1) * 8 * nw));
    KCHECK(d->xchg(X_GATHER, d->node_gid.p, ncnt.data(), n_gid.p, nrcnt.data(), 8, stream));
    KCHECK(d->xchg(X_GATHER, d->node_key.p, ncnt.data(), n_key.p, nrcnt.data(), 8 * nw, stream));
    if (me) {
        KCHECK(b->node_key.alloc((TN + 1) * 8 * nw, stream));
        if (TN) { if (nw == 1) KLAUNCH(place_keys_kernel<1>, TN, stream, n_gid.as<u64>(), n_key.as<u64>(), TN, b->node_key.as<u64>());
                  else         KLAUNCH(place_keys_kernel<2>, TN, stream, n_gid.as<u64>(), n_key.as<u64>(), TN, b->node_key.as<u64>()); }
        KCHECK_HIP(hipGetLastError());
        b->n_edges = TE; b->n_nodes = TN;
        trace_words("gather: placed keys", rank, b->edge_key.p, TE * nw, stream);
        trace_words("gather: placed nodes", rank, b->node_key.p, TN * nw, stream);
        const uint32_t lstride = label_stride_for_k(d->s.k);
        KCHECK(b->edge_label.alloc((TE + 1) * (size_t)lstride + 16, stream));
        KCHECK(dev_labels(b->edge_key.as<u64>(), TE, d->s.k, b->edge_label.as<uint8_t>(), stream));
        b->edge_age.release();
        b->edges_ready = true; b->finalized = true;
        if (root_builder) *root_builder = b;
    }
    KCHECK_HIP(hipStreamSynchronize(stream));
    // the ranks' shares have been consumed
    d->edge_src.release(); d->edge_dst.release(); d->edge_label.release(); d->node_key.release(); d->edge_gid.release(); d->node_gid.release();
    d->edge_lsrc.release(); d->edge_drank.release(); d->edge_dlocal.release(); d->edge_age.release();
    d->n_edges = d->n_nodes = d->n_src = 0;
    d->gathered = true;
    return KATOME_OK;
}

}  // extern "C"
