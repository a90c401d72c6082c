// radix.hip -- hand-written device primitives for gfx950 wave64: LSD radix sort (keys, optional
// u32 values, 64- or 128-bit keys), partition-by-owner (the same pass with a different digit),
// unique, rank-in-sorted-array, and the edge -> endpoints / label transforms.
//
// These build the graph out of the k-mer table: PtGraph node numbering (reference
// collections/graphs/pt_graph.rs:142-154), the endpoints of each edge (compress_kmer halves,
// compress.rs:23-26) and the compress_edge labels (compress.rs:250-271; post-pass
// pt_graph.rs:339-343).  All streaming, HBM-bound passes; no MFMA (integer keys).
#include <algorithm>

#include "common.h"

namespace katome {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
#ifndef KATOME_SORT_ITEMS
#define KATOME_SORT_ITEMS 16
#endif
#ifndef KATOME_SORT_ITEMS_WIDE
#define KATOME_SORT_ITEMS_WIDE 8
#endif
#ifndef KATOME_SORT_ITEMS_3
#define KATOME_SORT_ITEMS_3 5
#endif
#ifndef KATOME_SORT_WAVES
#define KATOME_SORT_WAVES 4      // workgroups per CU the scatter kernel is compiled for (register budget)
#endif
// keys per thread and per workgroup tile: 4096 one-word keys, 2048 wider ones -- the same 32 KiB of LDS and the same
// 128 bytes per digit run either way (a 64 KiB tile of 128-bit keys left two workgroups per CU: C5's passes 13.5 -> 10 ms)
template <int NW> struct SortTile {
    // (three-word records -- 95-base tiles, two-word keys with a tag: 1280 of them = 30 KiB, four workgroups per CU like the others;
    // at 2048 = 48 KiB there were two)
    static constexpr int ITEMS = NW == 1 ? KATOME_SORT_ITEMS : NW == 2 ? KATOME_SORT_ITEMS_WIDE : KATOME_SORT_ITEMS_3;
    static constexpr int KEYS = BLOCK * ITEMS;
};
// workgroups per offset chunk: the chunk kernel walks a chunk's workgroups serially, the offsets kernel walks the chunks
// serially -- about the square root of the workgroup count keeps both short (4 M keys per chunk at most: < 2^32)
static inline unsigned chunk_blocks_for(unsigned long long nblocks) {
    unsigned c = 64;
    while (c < 1024 && (unsigned long long)c * c < nblocks) c <<= 1;
    return c;
}
static_assert(BLOCK == RADIX, "one thread per digit in the offset kernels");

template <int NW> struct RadixDigit {
    u32 shift, bits;
    __device__ __forceinline__ u32 operator()(const Key<NW>& k) const { return key_digit(k, shift, bits); }
};
template <int NW> struct OwnerDigit {
    u64 n_parts;
    u32 core_shift, core_bases;          // core_bases == 0: owner by the whole key; else kmer_bits.h core_owner
    u32 minimizer = 0;                   // ... or, > 0, by the core's minimizer of that many bases (kmer_bits.h minimizer_owner)
    __device__ __forceinline__ u32 operator()(const Key<NW>& k0) const {
        if (!key_valid(k0)) return (u32)n_parts;
        Key<NW> k = k0;
        k.w[0] &= ~RC_MARK;              // first-seen-order records carry the orientation they dropped: not part of the key
        if (minimizer) return (u32)minimizer_owner(k, core_shift, core_bases, minimizer, n_parts);
        return core_bases ? (u32)core_owner(k, core_shift, core_bases, n_parts) : (u32)whole_key_owner(k, n_parts);
    }
};
// supermer records name their owner themselves (kmer_bits.h): a slot a read did not fill is invalid and goes last
struct SupermerOwnerDigit {
    u32 n_parts;
    __device__ __forceinline__ u32 operator()(const Key<2>& k) const { return key_valid(k) ? supermer_owner(k) : n_parts; }
};
// owner = the part of an ascending list of u64 values a value falls into: bounds[p] = first value of part p + 1
// (n_parts - 1 of them); used to spread sequence numbers over the ranks for a global ranking
struct RangeDigit {
    const u64* bounds; u32 n_parts;
    __device__ __forceinline__ u32 operator()(const Key<1>& k) const {
        u32 p = 0;
        for (u32 i = 0; i + 1 < n_parts; ++i) p += k.w[0] >= bounds[i];
        return p;
    }
};

// table-region digit: the table slot is mulhi(hash, capacity), monotone in the hash, so the top bits of
// the hash name the region of the table a record will land in
template <int NW> struct HashDigit {
    u32 shift;
    __device__ __forceinline__ u32 operator()(const Key<NW>& k) const { return (u32)(hash_key(k) >> shift) & (RADIX - 1); }
};

// the same digit for records that carry one more word behind the k-mer (first-seen builds: the packed sequence numbers travel as
// the record's last word; only the k-mer's NW - 1 words are hashed)
template <int NW> struct HashTaggedDigit {
    u32 shift;
    __device__ __forceinline__ u32 operator()(const Key<NW>& k) const {
        Key<NW - 1> c;
#pragma unroll
        for (int q = 0; q < NW - 1; ++q) c.w[q] = k.w[q];
        return (u32)(hash_key(c) >> shift) & (RADIX - 1);
    }
};

// ... and for a rank's own distinct k-mers on their way to their owners: the digit comes from the hash of the k-mer's CORE, whose
// top bits name the owner (kmer_bits.h core_owner), so that the groups of the LDS count are grouped by owner as well
template <int NW> struct CoreHashDigit {
    u32 shift, core_shift, core_bases;
    __device__ __forceinline__ u32 operator()(const Key<NW>& k) const { return (u32)(core_hash(k, core_shift, core_bases) >> shift) & (RADIX - 1); }
};

template <int NW> __device__ __forceinline__ Key<NW> load_key(const u64* p, u64 i) {
    Key<NW> k;
    if (NW == 1) { k.w[0] = p[i]; }
    else if (NW == 2) { ulonglong2 v = *reinterpret_cast<const ulonglong2*>(p + 2 * i); k.w[0] = v.x; k.w[NW - 1] = v.y; }
    else { k.w[0] = p[3 * i]; k.w[NW > 2 ? 1 : 0] = p[3 * i + 1]; k.w[NW - 1] = p[3 * i + 2]; }      // three-word tiles (64..95 bases)
    return k;
}
// the same load for data that is read once and not again by this kernel (a pass's input): non-temporal, so that the L2 lines it
// would take stay with the partial output lines that consecutive tiles complete (radix_scatter_kernel: 42.0 -> 40.5 ms per edge sort)
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
#ifndef KATOME_STREAM_LOADS
#define KATOME_STREAM_LOADS 1        // 0: plain loads; 1: the scatter pass; 2: + histogram; 3: + run sort
#endif
template <int NW> __device__ __forceinline__ Key<NW> load_key_stream(const u64* p, u64 i) {
    Key<NW> k;
    if (NW == 1) { k.w[0] = __builtin_nontemporal_load(p + i); }
    else if (NW == 2) { u64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(p + 2 * i)); k.w[0] = v.x; k.w[NW - 1] = v.y; }
    else { k.w[0] = __builtin_nontemporal_load(p + 3 * i); k.w[NW > 2 ? 1 : 0] = __builtin_nontemporal_load(p + 3 * i + 1); k.w[NW - 1] = __builtin_nontemporal_load(p + 3 * i + 2); }
    return k;
}
template <int NW> __device__ __forceinline__ void store_key(u64* p, u64 i, const Key<NW>& k) {
    if (NW == 1) p[i] = k.w[0];
    else if (NW == 2) *reinterpret_cast<ulonglong2*>(p + 2 * i) = make_ulonglong2(k.w[0], k.w[NW - 1]);
    else { p[3 * i] = k.w[0]; p[3 * i + 1] = k.w[NW > 2 ? 1 : 0]; p[3 * i + 2] = k.w[NW - 1]; }
}

// ---- pass 1: per-workgroup digit histogram -> counts[block][digit] ----------------------------
template <int NW, class Digit>
__global__ __launch_bounds__(BLOCK) void radix_hist_kernel(const u64* __restrict__ keys, u64 n, Digit dg, u32* __restrict__ counts) {
    __shared__ u32 h[RADIX];
    const u32 tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    constexpr int SORT_ITEMS = SortTile<NW>::ITEMS, SORT_TILE = SortTile<NW>::KEYS;
    const u64 base = (u64)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        u64 i = base + (u64)j * BLOCK + tid;
#if KATOME_STREAM_LOADS >= 2
        if (i < n) atomicAdd(&h[dg(load_key_stream<NW>(keys, i))], 1u);
#else
        if (i < n) atomicAdd(&h[dg(load_key<NW>(keys, i))], 1u);
#endif
    }
    __syncthreads();
    counts[(u64)blockIdx.x * RADIX + tid] = h[tid];
}

// ---- pass 2a: per chunk of workgroups, per digit: sum; counts become exclusive prefixes inside
// the chunk (u32), chunk_sum[chunk][digit] holds the chunk totals ---------------------------------
__global__ __launch_bounds__(BLOCK) void radix_chunk_kernel(u32* __restrict__ counts, u64 nblocks, u64* __restrict__ chunk_sum, u32 chunk_blocks) {
    const u32 d = threadIdx.x;
    const u64 b0 = (u64)blockIdx.x * chunk_blocks;
    const u64 b1 = b0 + chunk_blocks < nblocks ? b0 + chunk_blocks : nblocks;
    u32 run = 0;
    // (eight rows' loads in flight before the first store: a walk that waited for every load -- 1024 rows, one after the other, with
    // 1.5 workgroups per CU -- took 0.4-0.5 ms per pass)
    constexpr int U = 8;
    u64 b = b0;
    for (; b + U <= b1; b += U) {
        u32 c[U];
#pragma unroll
        for (int j = 0; j < U; ++j) c[j] = counts[(b + j) * RADIX + d];
#pragma unroll
        for (int j = 0; j < U; ++j) { counts[(b + j) * RADIX + d] = run; run += c[j]; }
    }
    for (; b < b1; ++b) {
        u32 c = counts[b * RADIX + d];
        counts[b * RADIX + d] = run;
        run += c;
    }
    chunk_sum[(u64)blockIdx.x * RADIX + d] = run;
}

// ---- pass 2b: one workgroup turns chunk sums into global exclusive offsets, digit-major --------
// chunk_sum[chunk][digit] -> chunk_off[chunk][digit]; digit_total[digit] gets each digit's count
__global__ __launch_bounds__(BLOCK) void radix_offsets_kernel(u64* __restrict__ chunk_sum, u64 nchunks, u64* __restrict__ digit_total) {
    __shared__ u64 tot[RADIX];
    const u32 d = threadIdx.x;
    u64 t = 0;
    for (u64 c = 0; c < nchunks; ++c) t += chunk_sum[c * RADIX + d];
    tot[d] = t;
    if (digit_total) digit_total[d] = t;
    __syncthreads();
    u64 start = 0;
    for (u32 e = 0; e < d; ++e) start += tot[e];
    constexpr int U = 8;                       // (as in radix_chunk_kernel: the loads of eight chunks before the first store)
    u64 c = 0;
    for (; c + U <= nchunks; c += U) {
        u64 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = chunk_sum[(c + j) * RADIX + d];
#pragma unroll
        for (int j = 0; j < U; ++j) { chunk_sum[(c + j) * RADIX + d] = start; start += v[j]; }
    }
    for (; c < nchunks; ++c) {
        u64 v = chunk_sum[c * RADIX + d];
        chunk_sum[c * RADIX + d] = start;
        start += v;
    }
}

// ---- pass 3: stable scatter ----------------------------------------------------------------------
// Each wave owns 1024 consecutive keys of the tile and ranks them 64 at a time with ballot-built
// match masks (rank = keys of the same digit earlier in the wave); a cross-wave prefix gives the
// key's place in the tile's digit-sorted order, the tile is reordered through LDS and written out
// so that consecutive lanes store consecutive addresses of each digit's run.
// STABLE = false (round 4): a pass that nothing depends on the order of -- the FIRST pass of a level's two hash passes and of a sort
// of distinct keys: whatever order the records arrive in is as good as any other -- ranks a record with one LDS atomic on its digit's
// counter instead of the eight ballots of the match step.  Built to close the gap to the pass's memory-pattern ceiling
// (profiles/r04_scatter_ceiling.txt: 0.80 of it) and measured at C3 on one box, A/B/A/B: 203.8 / 205.8 / 207.3 / 205.1 ms per build --
// nothing: the LDS atomics cost what the ballots cost.  Kept behind KATOME_UNSTABLE_FIRST=1, off by default.
template <int NW, bool HAS_VAL, class Digit, bool STABLE = true>
__global__ __launch_bounds__(BLOCK, KATOME_SORT_WAVES) void radix_scatter_kernel(const u64* __restrict__ keys_in, const u32* __restrict__ vals_in,
                                                               u64 n, Digit dg, const u32* __restrict__ rel,
                                                               const u64* __restrict__ chunk_off, u64* __restrict__ keys_out,
                                                               u32* __restrict__ vals_out, u32 chunk_blocks, u32 xcd_tiles) {
    constexpr int SORT_ITEMS = SortTile<NW>::ITEMS, SORT_TILE = SortTile<NW>::KEYS;
    extern __shared__ u64 smem[];
    u64* skeys = smem;                                            // [SORT_TILE * NW]; reused for the values afterwards
    __shared__ u32 whist[BLOCK / 64][RADIX];
    __shared__ u32 dstart[RADIX];
    __shared__ u64 gbase[RADIX];
    __shared__ u32 wsum[BLOCK / 64];

    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups are dealt out round-robin over the 8 XCDs, each with an L2 of its own; tiles that follow each other write
    // stretches that follow each other in every digit's run.  With xcd_tiles (tiles per XCD) > 0 workgroup i takes tile
    // (i % 8) * xcd_tiles + i / 8, so that an XCD works through consecutive tiles and the halves of a cache line two tiles share
    // meet in ONE L2 instead of being written back from two.
    const u64 bid = xcd_tiles ? (u64)(blockIdx.x & 7u) * xcd_tiles + (blockIdx.x >> 3) : (u64)blockIdx.x;
    const u64 base = bid * SORT_TILE;
    if (base >= n) return;                                  // (the grid is rounded up to 8 x xcd_tiles)
    const u32 cnt = (u32)((n - base) < (u64)SORT_TILE ? (n - base) : (u64)SORT_TILE);
    for (u32 i = tid; i < (BLOCK / 64) * RADIX; i += BLOCK) (&whist[0][0])[i] = 0;
    __syncthreads();

    Key<NW> key[SORT_ITEMS]; u32 val[SORT_ITEMS]; u32 dig[SORT_ITEMS]; u32 rnk[SORT_ITEMS];
    const u64 lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const u32 idx = wave * (64 * SORT_ITEMS) + j * 64 + lane;
        const bool valid = idx < cnt;
        u32 d = 0;
        if (valid) {
#if KATOME_STREAM_LOADS >= 1
            key[j] = load_key_stream<NW>(keys_in, base + idx);
            if (HAS_VAL) val[j] = __builtin_nontemporal_load(&vals_in[base + idx]);
#else
            key[j] = load_key<NW>(keys_in, base + idx);
            if (HAS_VAL) val[j] = vals_in[base + idx];
#endif
            d = dg(key[j]);
        }
        if (!STABLE) {
            dig[j] = d;
            rnk[j] = valid ? atomicAdd(&whist[0][d], 1u) : 0u;       // (place among the tile's records of this digit, in arrival order)
            continue;
        }
        // the lanes of this row that hold the same digit: a lane differs from another where, for some bit, the row's vote on that bit
        // and its own bit disagree.  `mine` is the lane's bit spread over a word (0 / ~0), so a bit costs one compare (the vote), two
        // xors and two ors; as `bit ? vote : ~vote` on 64-bit masks hipcc 7.2 spent eleven VALU operations per bit and so many
        // scalar pairs that the kernel's arguments lived in VGPR lanes (538 v_readlane per tile): 183 -> 117 VALU operations per key
        const u64 vmask = __ballot(valid);
        u32 diff_lo = 0, diff_hi = 0;
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const u32 mine = (u32)((int)(d << (31 - b)) >> 31);
            const u64 vote = __ballot(mine != 0);
            diff_lo |= (u32)vote ^ mine;
            diff_hi |= (u32)(vote >> 32) ^ mine;
        }
        const u64 m = ~((u64)diff_hi << 32 | diff_lo) & vmask;
        const u32 prior = __popcll(m & lt_mask);
        u32 old = 0;
        if (valid) old = whist[wave][d];
        if (valid && prior == 0) whist[wave][d] = old + __popcll(m);
        dig[j] = d;
        rnk[j] = old + prior;
    }
    __syncthreads();

    {   // thread = digit: wave-exclusive prefixes, tile-wide digit starts, global run bases
        const u32 d = tid;
        u32 run = 0;
        if (!STABLE) { run = whist[0][d]; whist[0][d] = 0; }       // (one counter per digit: the ranks are tile-wide already)
        else {
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) { u32 c = whist[w][d]; whist[w][d] = run; run += c; }
        }
        u32 incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { u32 v = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u32 woff = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) if (w < (int)wave) woff += wsum[w];
        dstart[d] = woff + incl - run;
        gbase[d] = chunk_off[(bid / chunk_blocks) * RADIX + d] + rel[bid * RADIX + d];
    }
    __syncthreads();

    // the tile goes through LDS twice, keys first and then the values through the same buffer: half the LDS per
    // workgroup means twice the workgroups (and bytes in flight) per CU, which is what bounds this kernel
    u32 pos[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const u32 idx = wave * (64 * SORT_ITEMS) + j * 64 + lane;
        pos[j] = idx < cnt ? dstart[dig[j]] + whist[wave][dig[j]] + rnk[j] : 0;
        if (idx < cnt) {
#pragma unroll
            for (int q = 0; q < NW; ++q) skeys[pos[j] * NW + q] = key[j].w[q];
        }
    }
    __syncthreads();
    u32 dout[SORT_ITEMS];            // digit of the key this thread writes out in row j (needed again for its value)
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const u32 i = j * BLOCK + tid;
        dout[j] = 0;
        if (i < cnt) {
            Key<NW> k;
#pragma unroll
            for (int q = 0; q < NW; ++q) k.w[q] = skeys[i * NW + q];
            const u32 d = dg(k);
            dout[j] = d;
            store_key<NW>(keys_out, gbase[d] + (i - dstart[d]), k);
        }
    }
    if (HAS_VAL) {
        u32* svals = reinterpret_cast<u32*>(smem);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const u32 idx = wave * (64 * SORT_ITEMS) + j * 64 + lane;
            if (idx < cnt) svals[pos[j]] = val[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const u32 i = j * BLOCK + tid;
            if (i < cnt) vals_out[gbase[dout[j]] + (i - dstart[dout[j]])] = svals[i];
        }
    }
}

struct PassBuffers {
    DevBuf counts, chunk, totals;
    u64 nblocks = 0, nchunks = 0;
    u32 chunk_blocks = 64;
    int init(u64 n, int nw, hipStream_t stream) {
        counts.stream = chunk.stream = totals.stream = stream;
        const u64 tile = nw == 1 ? SortTile<1>::KEYS : nw == 2 ? SortTile<2>::KEYS : SortTile<3>::KEYS;
        nblocks = (n + tile - 1) / tile;
        chunk_blocks = chunk_blocks_for(nblocks);
        nchunks = (nblocks + chunk_blocks - 1) / chunk_blocks;
        KCHECK(counts.alloc(nblocks * RADIX * sizeof(u32)));
        KCHECK(chunk.alloc(nchunks * RADIX * sizeof(u64)));
        KCHECK(totals.alloc(RADIX * sizeof(u64)));
        return KATOME_OK;
    }
};

// which per-kernel timer (common.h K_*) a pass with this digit reports to
template <class Digit> struct DigitTimers { static constexpr int HIST = K_SORT_HIST, SCATTER = K_SORT_SCATTER; };
template <int NW> struct DigitTimers<HashDigit<NW>> { static constexpr int HIST = K_HASH_HIST, SCATTER = K_HASH_SCATTER; };
template <int NW> struct DigitTimers<HashTaggedDigit<NW>> { static constexpr int HIST = K_HASH_HIST, SCATTER = K_HASH_SCATTER; };
template <int NW> struct DigitTimers<CoreHashDigit<NW>> { static constexpr int HIST = K_HASH_HIST, SCATTER = K_HASH_SCATTER; };
template <int NW> struct DigitTimers<OwnerDigit<NW>> { static constexpr int HIST = K_OWNER_HIST, SCATTER = K_OWNER_SCATTER; };
template <> struct DigitTimers<SupermerOwnerDigit> { static constexpr int HIST = K_OWNER_HIST, SCATTER = K_OWNER_SCATTER; };
template <> struct DigitTimers<RangeDigit> { static constexpr int HIST = K_OWNER_HIST, SCATTER = K_OWNER_SCATTER; };

static bool unstable_first() {
    static const bool on = getenv("KATOME_UNSTABLE_FIRST") && atoi(getenv("KATOME_UNSTABLE_FIRST")) != 0;      // (off: measured, no gain -- see the kernel)
    return on;
}
template <int NW, bool HAS_VAL, class Digit, bool STABLE = true>
static int radix_pass(const u64* kin, const u32* vin, u64 n, Digit dg, u64* kout, u32* vout, PassBuffers& pb, hipStream_t stream,
                      bool have_counts = false) {
    if (pb.nblocks > 0x7fffffffull) { set_error("radix pass: %llu keys exceed the grid limit", (unsigned long long)n); return KATOME_E_ARG; }
    dim3 block(BLOCK);
    if (!have_counts) {          // (have_counts: whoever wrote the records counted this pass's digits per tile as it went -- pb.counts holds them)
        KernelScope ks(DigitTimers<Digit>::HIST, stream, n);
        hipLaunchKernelGGL((radix_hist_kernel<NW, Digit>), dim3((unsigned)pb.nblocks), block, 0, stream, kin, n, dg, pb.counts.as<u32>());
    }
    {
        KernelScope ks(K_PASS_OFFSETS, stream, n);
        hipLaunchKernelGGL(radix_chunk_kernel, dim3((unsigned)pb.nchunks), block, 0, stream, pb.counts.as<u32>(), pb.nblocks, pb.chunk.as<u64>(), pb.chunk_blocks);
        hipLaunchKernelGGL(radix_offsets_kernel, dim3(1), block, 0, stream, pb.chunk.as<u64>(), pb.nchunks, pb.totals.as<u64>());
    }
    const size_t lds = (size_t)SortTile<NW>::KEYS * NW * 8;
    if (lds > (64u << 10)) {          // three-word records: 96 KiB of the CU's 160 KiB
        KCHECK_HIP(hipFuncSetAttribute((const void*)radix_scatter_kernel<NW, HAS_VAL, Digit, STABLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    {
        static const bool xcd_aware = !getenv("KATOME_XCD_TILES") || atoi(getenv("KATOME_XCD_TILES")) != 0;       // (0: workgroup i takes tile i)
        const u32 xcd_tiles = xcd_aware && pb.nblocks >= 64 ? (u32)((pb.nblocks + 7) / 8) : 0u;
        KernelScope ks((!HAS_VAL && DigitTimers<Digit>::SCATTER == K_SORT_SCATTER) ? (int)K_SORT_SCATTER_KEYS : (int)DigitTimers<Digit>::SCATTER, stream, n);
        hipLaunchKernelGGL((radix_scatter_kernel<NW, HAS_VAL, Digit, STABLE>), dim3(xcd_tiles ? xcd_tiles * 8u : (unsigned)pb.nblocks), block, lds, stream, kin, vin, n, dg,
                           pb.counts.as<u32>(), pb.chunk.as<u64>(), kout, vout, pb.chunk_blocks, xcd_tiles);
    }
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

// After a stable sort on the TOP bits only (bits [low, key_bits)) the keys are in order except inside the runs that share
// those bits.  With 8 * passes >= log2(n) top bits such runs are a handful of records wherever the keys are spread out
// (k-mers of a genome: the top 16 bases), so the remaining passes -- half of them for k = 31, three quarters for k = 63 --
// are replaced by ONE streaming pass: every record finds its run by scanning its neighbours in LDS (left while the top
// bits agree, then right), counts the records of the run that must precede it (smaller key, or equal key further left:
// stable) and is written to run start + count.  A workgroup stages its tile plus a halo on both sides; a run that leaves
// the staged window raises `overflow` and the caller falls back to the remaining passes.
#ifndef KATOME_RUN_TILE
#define KATOME_RUN_TILE 4096
#endif
#ifndef KATOME_RUN_HALO
#define KATOME_RUN_HALO 512
#endif
#ifndef KATOME_RUN_TILE_WIDE
#define KATOME_RUN_TILE_WIDE 2048
#endif
// records a workgroup places: 4096 one-word keys (40 KiB staged with the halo), 2048 wider ones (48 KiB; 4096 of them
// took 80 KiB and left one workgroup per CU: C5's run sort 28 -> 15 ms)
template <int NW> struct RunTile { static constexpr u32 KEYS = NW == 1 ? KATOME_RUN_TILE : KATOME_RUN_TILE_WIDE; };
constexpr u32 RUN_HALO = KATOME_RUN_HALO;               // longest run that can be followed on either side
template <int NW, bool HAS_VAL>
__global__ __launch_bounds__(BLOCK) void run_sort_kernel(const u64* __restrict__ keys_in, const u32* __restrict__ vals_in, u64 n, u32 low,
                                                          u64* __restrict__ keys_out, u32* __restrict__ vals_out, u32* __restrict__ overflow) {
    extern __shared__ u64 lk[];                        // [(RUN_TILE + 2 * RUN_HALO) * NW]
    constexpr u32 RUN_TILE = RunTile<NW>::KEYS;
    constexpr u32 SPAN = RUN_TILE + 2 * RUN_HALO, PER = SPAN / BLOCK, OWN = RUN_TILE / BLOCK;
    static_assert(SPAN % BLOCK == 0 && RUN_TILE % BLOCK == 0, "whole rows per thread");
    const u64 n_tiles = (n + RUN_TILE - 1) / RUN_TILE;
    for (u64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const u64 t0 = tile * RUN_TILE;
        const u64 g0 = t0 >= RUN_HALO ? t0 - RUN_HALO : 0;
        const u64 t1 = t0 + RUN_TILE < n ? t0 + RUN_TILE : n;
        const u64 g1 = t1 + RUN_HALO < n ? t1 + RUN_HALO : n;
        const u32 cnt = (u32)(g1 - g0), off = (u32)(t0 - g0), own = (u32)(t1 - t0);
        // all loads of the tile are issued before the first one is waited for (a load per loop trip would serialise ~20
        // memory latencies per workgroup and tile)
        Key<NW> stage[PER];
#pragma unroll
        for (u32 r = 0; r < PER; ++r) {
            const u32 j = r * BLOCK + threadIdx.x;
            if (j < cnt) stage[r] = load_key<NW>(keys_in, g0 + j);
        }
        u32 val[OWN];
        if (HAS_VAL) {
#pragma unroll
            for (u32 r = 0; r < OWN; ++r) {
                const u32 i = r * BLOCK + threadIdx.x;
                if (i < own) val[r] = vals_in[t0 + i];
            }
        }
#pragma unroll
        for (u32 r = 0; r < PER; ++r) {
            const u32 j = r * BLOCK + threadIdx.x;
            if (j < cnt) {
#pragma unroll
                for (int q = 0; q < NW; ++q) lk[j * NW + q] = stage[r].w[q];
            }
        }
        __syncthreads();
        // A scan step is one LDS round trip whose result decides whether there is a next one, and a wave scans for as long as
        // its longest run: U records per thread are scanned in lockstep so that U reads are in flight per step
        constexpr u32 U = 8;
        static_assert(OWN % U == 0, "whole groups");
#pragma unroll
        for (u32 r0 = 0; r0 < OWN; r0 += U) {
            Key<NW> key[U], top[U];
            u32 before[U], lo[U], hi[U], live = 0, lost = 0;
#pragma unroll
            for (u32 u = 0; u < U; ++u) {
                const u32 i = (r0 + u) * BLOCK + threadIdx.x, j = off + i;
                before[u] = 0; lo[u] = j; hi[u] = j + 1;
                if (i < own) {
                    live |= 1u << u;
#pragma unroll
                    for (int q = 0; q < NW; ++q) key[u].w[q] = lk[j * NW + q];
                    top[u] = key_shr(key[u], low);
                }
            }
            u32 go = live;
            while (go) {                                // to the left: records of the run with a key <= this one come first
#pragma unroll
                for (u32 u = 0; u < U; ++u) {
                    if (!(go & (1u << u))) continue;
                    if (lo[u] == 0) { if (g0 > 0) lost |= 1u << u; go &= ~(1u << u); continue; }
                    Key<NW> x;
#pragma unroll
                    for (int q = 0; q < NW; ++q) x.w[q] = lk[(lo[u] - 1) * NW + q];
                    if (!key_eq(key_shr(x, low), top[u])) { go &= ~(1u << u); continue; }
                    before[u] += key_lt(key[u], x) ? 0u : 1u;
                    --lo[u];
                }
            }
            go = live;
            while (go) {                                // to the right: only strictly smaller keys
#pragma unroll
                for (u32 u = 0; u < U; ++u) {
                    if (!(go & (1u << u))) continue;
                    if (hi[u] == cnt) { if (g1 < n) lost |= 1u << u; go &= ~(1u << u); continue; }
                    Key<NW> x;
#pragma unroll
                    for (int q = 0; q < NW; ++q) x.w[q] = lk[hi[u] * NW + q];
                    if (!key_eq(key_shr(x, low), top[u])) { go &= ~(1u << u); continue; }
                    before[u] += key_lt(x, key[u]) ? 1u : 0u;
                    ++hi[u];
                }
            }
#pragma unroll
            for (u32 u = 0; u < U; ++u) {
                if (!(live & (1u << u))) continue;
                const u32 j = off + (r0 + u) * BLOCK + threadIdx.x;
                u64 out = g0 + lo[u] + before[u];
                if (lost & (1u << u)) { *overflow = 1; out = g0 + j; }   // (the result is discarded; keep the store in bounds)
                store_key<NW>(keys_out, out, key[u]);
                if (HAS_VAL) vals_out[out] = val[r0 + u];
            }
        }
        __syncthreads();
    }
}

// The same placement without LDS: where the keys are spread out a run is a few records, so a wave takes a stretch of records with
// RW_REACH more on either side and every lane finds its run among its neighbours' keys by wave shuffles -- no staging, no
// round trips to LDS whose results decide whether there is another one.  A run that leaves the shuffled window (or the wave's
// stretch) is walked in global memory by the lanes it concerns (rare: a window of 2 * RW_REACH + 1 records of equal top bits);
// one longer than RW_LONGEST raises `overflow` like the staged kernel.
constexpr u32 RW_REACH = 7, RW_OWN = 64 - 2 * RW_REACH, RW_LONGEST = 1u << 12;
template <int NW> __device__ __forceinline__ Key<NW> shfl_key_up(const Key<NW>& k, u32 s) {
    Key<NW> r;
#pragma unroll
    for (int q = 0; q < NW; ++q) r.w[q] = __shfl_up(k.w[q], s, 64);
    return r;
}
template <int NW> __device__ __forceinline__ Key<NW> shfl_key_down(const Key<NW>& k, u32 s) {
    Key<NW> r;
#pragma unroll
    for (int q = 0; q < NW; ++q) r.w[q] = __shfl_down(k.w[q], s, 64);
    return r;
}
template <int NW, bool HAS_VAL>
__global__ __launch_bounds__(BLOCK) void run_sort_wave_kernel(const u64* __restrict__ keys_in, const u32* __restrict__ vals_in, u64 n, u32 low,
                                                               u64* __restrict__ keys_out, u32* __restrict__ vals_out, u32* __restrict__ overflow) {
    const u32 lane = threadIdx.x & 63;
    const u64 n_chunks = (n + RW_OWN - 1) / RW_OWN;
    // every XCD (workgroup index mod 8) works through its own eighth of the stretches: neighbouring stretches share cache lines,
    // which then meet in one L2 (see radix_scatter_kernel); the grid is a multiple of 8 workgroups
    const u64 per_xcd = (n_chunks + 7) / 8, xcd = blockIdx.x & 7u;
    const u64 wave = ((u64)(blockIdx.x >> 3) * BLOCK + threadIdx.x) >> 6, n_waves = ((u64)(gridDim.x >> 3) * BLOCK) >> 6;
    const u64 c_end = (xcd + 1) * per_xcd < n_chunks ? (xcd + 1) * per_xcd : n_chunks;
    // A wave works through a CONTIGUOUS block of stretches (round 4): the 2 * RW_REACH records two neighbouring stretches share are
    // then in this wave's own registers -- the last lanes of the stretch before -- and every key is fetched once (with the stretches
    // dealt out round-robin the halo was fetched again by another wave: 1.5 x the algorithmic read, profiles/r03_summary.md).
    const u64 per_wave = (per_xcd + n_waves - 1) / n_waves;
    const u64 c_first = xcd * per_xcd + wave * per_wave, c_last = c_first + per_wave < c_end ? c_first + per_wave : c_end;
    // (the next stretch's loads are issued before this one is worked on; `all`: every lane loads, else only the lanes whose record the
    // stretch before did not hold)
    auto fetch = [&](u64 c, bool all, Key<NW>& key, u32& val) {
        const long long j = (long long)(c * RW_OWN) - (long long)RW_REACH + (long long)lane;
        const bool there = c < n_chunks && j >= 0 && (u64)j < n, fresh = there && (all || lane >= 2 * RW_REACH);
#pragma unroll
        for (int q = 0; q < NW; ++q) key.w[q] = 0;
        val = 0;
        // (a value is only ever read by the lane that owns its record: every own lane loads its own, carried key or not)
#if KATOME_STREAM_LOADS >= 3
        if (fresh) key = load_key_stream<NW>(keys_in, (u64)j);
        if (HAS_VAL && there && lane >= RW_REACH && lane < RW_REACH + RW_OWN) val = __builtin_nontemporal_load(&vals_in[j]);
#else
        if (fresh) key = load_key<NW>(keys_in, (u64)j);
        if (HAS_VAL && there && lane >= RW_REACH && lane < RW_REACH + RW_OWN) val = vals_in[j];
#endif
    };
    Key<NW> key_next; u32 val_next;
    fetch(c_first < c_last ? c_first : n_chunks, true, key_next, val_next);
    for (u64 c = c_first; c < c_last; ++c) {
        const long long j = (long long)(c * RW_OWN) - (long long)RW_REACH + (long long)lane;         // the record this lane looks at
        const bool there = j >= 0 && (u64)j < n;
        const bool own = there && lane >= RW_REACH && lane < RW_REACH + RW_OWN;
        const Key<NW> key = key_next;
        const u32 val = val_next;
        fetch(c + 1 < c_last ? c + 1 : n_chunks, false, key_next, val_next);
        {   // the first 2 * RW_REACH records of the next stretch are this stretch's last ones: lane l takes lane l + RW_OWN's
            Key<NW> carried;
#pragma unroll
            for (int q = 0; q < NW; ++q) carried.w[q] = __shfl(key.w[q], (int)((lane + RW_OWN) & 63u), 64);
            if (lane < 2 * RW_REACH && c + 1 < c_last) key_next = carried;
        }
        const Key<NW> top = key_shr(key, low);
        // to the left: records of the run with a key <= this one come first; to the right: only strictly smaller keys
        u32 left = 0, before = 0;
        bool open = there, more_left = false, more_right = false;
#pragma unroll
        for (u32 s = 1; s <= RW_REACH; ++s) {
            const Key<NW> x = shfl_key_up<NW>(key, s);
            const bool has = lane >= s && j - (long long)s >= 0;
            const bool same = open && has && key_eq(key_shr(x, low), top);
            left += same ? 1u : 0u;
            before += (same && !key_lt(key, x)) ? 1u : 0u;
            open = same;
        }
        more_left = open && j - (long long)RW_REACH > 0;              // the window ended inside the run
        open = there;
#pragma unroll
        for (u32 s = 1; s <= RW_REACH; ++s) {
            const Key<NW> x = shfl_key_down<NW>(key, s);
            const bool has = lane + s < 64 && (u64)j + s < n;
            const bool same = open && has && key_eq(key_shr(x, low), top);
            before += (same && key_lt(x, key)) ? 1u : 0u;
            open = same;
        }
        more_right = open && (u64)j + RW_REACH + 1 < n;
        if (own && (more_left || more_right)) {       // a long run: the rest of it from global memory
            u64 a = (u64)j - left, b = (u64)j + 1;
            u32 steps = 0;
            if (more_left) {
                while (a > 0 && steps < RW_LONGEST) {
                    const Key<NW> x = load_key<NW>(keys_in, a - 1);
                    if (!key_eq(key_shr(x, low), top)) break;
                    before += key_lt(key, x) ? 0u : 1u; ++left; --a; ++steps;
                }
            }
            // (the right side is counted afresh: the window's share of it is in `before` already only for the first RW_REACH records)
            if (more_right) {
                b = (u64)j + RW_REACH + 1;
                while (b < n && steps < RW_LONGEST) {
                    const Key<NW> x = load_key<NW>(keys_in, b);
                    if (!key_eq(key_shr(x, low), top)) break;
                    before += key_lt(x, key) ? 1u : 0u; ++b; ++steps;
                }
            }
            if (steps >= RW_LONGEST) { *overflow = 1; before = left; }       // (the result is discarded; the store stays in bounds)
        }
        if (own) {
            const u64 out = (u64)j - left + before;
            store_key<NW>(keys_out, out, key);
            if (HAS_VAL) vals_out[out] = val;
        }
    }
}

// (A one-kernel pass -- digit offsets of all passes from one read of the keys, tile offsets by decoupled look-back over tiles
// numbered in the order they start -- was built and measured in round 3 and removed again: profiles/r03_lookback.md.  The
// look-back words must be read at agent scope, past the XCD's own L2, and every tile waits for that chain before it can write:
// C3's four sort passes 53 + 10 + 3 ms -> 67 ms (75 ms with eight look-back loads in flight per digit), the two 9-bit hash
// passes 23 + 5 + 1 -> 38 ms.)
// passes over the top bits before the run sort takes over: the fewest with 2^(8 * passes) >= n
static u32 top_passes_for(u64 n) {
    u32 t = 1;
    while (t < 8 && (n >> (8 * t)) != 0) ++t;
    return t;
}

// own_k / own_v (optional): the buffers that hold d_keys / d_vals.  An odd number of passes leaves the result in the
// temporaries; with the owners given they simply take those over (no copy back: 24 B per pair saved).
template <int NW, bool HAS_VAL>
static int sort_t(u64* d_keys, u32* d_vals, u64 n, u32 key_bits, hipStream_t stream, DevBuf* own_k = nullptr, DevBuf* own_v = nullptr,
                  bool distinct_keys = false) {
    if (n < 2) return KATOME_OK;
    PassBuffers pb;
    KCHECK(pb.init(n, NW, stream));
    DevBuf tk(stream), tv(stream);
    KCHECK(tk.alloc(own_k ? std::max<size_t>(n * 8 * NW, own_k->bytes) : n * 8 * NW));       // (a swapped-in buffer must be no smaller)
    if (HAS_VAL) KCHECK(tv.alloc(own_v ? std::max<size_t>(n * 4, own_v->bytes) : n * 4));
    u64* kin = d_keys; u64* kout = tk.as<u64>();
    u32* vin = d_vals; u32* vout = tv.as<u32>();
    auto flip = [&]() { u64* t = kin; kin = kout; kout = t; u32* tv2 = vin; vin = vout; vout = tv2; };
    bool first = true;
    auto pass = [&](u32 shift) -> int {
        RadixDigit<NW> dg{shift, (key_bits - shift) < (u32)RADIX_BITS ? (key_bits - shift) : (u32)RADIX_BITS};
        // (keys that are all different end up in key order whatever order equal DIGITS keep in the first pass; later passes must keep
        // what earlier ones established)
        if (first && distinct_keys && unstable_first()) KCHECK((radix_pass<NW, HAS_VAL, RadixDigit<NW>, false>(kin, vin, n, dg, kout, vout, pb, stream)));
        else KCHECK((radix_pass<NW, HAS_VAL>(kin, vin, n, dg, kout, vout, pb, stream)));
        first = false;
        flip();
        return KATOME_OK;
    };
    const u32 all_passes = (key_bits + RADIX_BITS - 1) / RADIX_BITS, top_passes = top_passes_for(n);
    u32 low = 0;                                            // bits below `low` are left to the run sort
    // the run sort costs about one pass: worth it from two saved passes on
    if (top_passes + 2 <= all_passes && n >= (1u << 16) && !getenv("KATOME_FULL_SORT")) low = key_bits - top_passes * RADIX_BITS;
    for (u32 shift = low; shift < key_bits; shift += RADIX_BITS) KCHECK(pass(shift));
    if (low) {
        DevBuf overflow(stream);
        KCHECK(overflow.alloc(16));
        KCHECK_HIP(hipMemsetAsync(overflow.p, 0, 4, stream));
        constexpr u32 RUN_TILE = RunTile<NW>::KEYS;
        const size_t lds = (size_t)(RUN_TILE + 2 * RUN_HALO) * NW * 8;
        if (lds > (64u << 10)) KCHECK_HIP(hipFuncSetAttribute((const void*)run_sort_kernel<NW, HAS_VAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        {
            static const int by_waves = getenv("KATOME_RUN_SORT") ? atoi(getenv("KATOME_RUN_SORT")) : 2;      // 1: the staged kernel (A/B)
            KernelScope ks(HAS_VAL ? K_RUN_SORT : K_RUN_SORT_KEYS, stream, n);
            if (by_waves == 2)
                hipLaunchKernelGGL((run_sort_wave_kernel<NW, HAS_VAL>), dim3((grid_for(n, (BLOCK / 64) * RW_OWN * 4, 256u * 32u) + 7u) & ~7u), dim3(BLOCK), 0, stream, kin, vin, n,
                                   low, kout, vout, overflow.as<u32>());
            else
                hipLaunchKernelGGL((run_sort_kernel<NW, HAS_VAL>), dim3(grid_for(n, RUN_TILE, 256u * 16u)), dim3(BLOCK), lds, stream, kin, vin, n, low,
                                   kout, vout, overflow.as<u32>());
        }
        KCHECK_HIP(hipGetLastError());
        u32 h = 0;
        KCHECK_HIP(hipMemcpyAsync(&h, overflow.p, 4, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        if (!h) flip();
        else {                                              // a long run of keys sharing their top bits: the passes that were left out, then
            for (u32 shift = 0; shift < low; shift += RADIX_BITS) {      // the top ones again (LSD order)
                RadixDigit<NW> dg{shift, (low - shift) < (u32)RADIX_BITS ? (low - shift) : (u32)RADIX_BITS};
                KCHECK((radix_pass<NW, HAS_VAL>(kin, vin, n, dg, kout, vout, pb, stream)));
                flip();
            }
            for (u32 shift = low; shift < key_bits; shift += RADIX_BITS) KCHECK(pass(shift));
        }
    }
    if (kin != d_keys) {
        if (own_k && (!HAS_VAL || own_v)) {                 // the owners take the temporaries over, the old buffers go back to the cache
            { void* q = tk.take(); const size_t b = own_k->bytes; void* old = own_k->take(); own_k->adopt(q, b); tk.adopt(old, b); }
            if (HAS_VAL) { void* q = tv.take(); const size_t b = own_v->bytes; void* old = own_v->take(); own_v->adopt(q, b); tv.adopt(old, b); }
        } else {
            KCHECK_HIP(hipMemcpyAsync(d_keys, kin, n * 8 * NW, hipMemcpyDeviceToDevice, stream));
            if (HAS_VAL) KCHECK_HIP(hipMemcpyAsync(d_vals, vin, n * 4, hipMemcpyDeviceToDevice, stream));
        }
    }
    return KATOME_OK;      // temporaries go back to the stream-ordered cache
}

// keys (and values) held in DevBufs: sorted "in place" from the caller's point of view, but the buffers may be exchanged for
// the sort's temporaries instead of copied back (pointers taken from them before the call are stale afterwards)
int dev_sort_bufs(DevBuf& keys, DevBuf* vals, uint64_t n, uint32_t nw, uint32_t key_bits, hipStream_t stream, bool distinct_keys) {
    if (nw != 1 && nw != 2) { set_error("key_words must be 1 or 2"); return KATOME_E_ARG; }
    if (key_bits == 0 || key_bits > 64 * nw) { set_error("key_bits out of range"); return KATOME_E_ARG; }
    if (keys.bytes < n * 8 * nw || (vals && vals->bytes < n * 4)) { set_error("sort: buffer too small"); return KATOME_E_ARG; }
    if (nw == 1) return vals ? sort_t<1, true>(keys.as<u64>(), vals->as<u32>(), n, key_bits, stream, &keys, vals, distinct_keys) : sort_t<1, false>(keys.as<u64>(), nullptr, n, key_bits, stream, &keys, nullptr, distinct_keys);
    return vals ? sort_t<2, true>(keys.as<u64>(), vals->as<u32>(), n, key_bits, stream, &keys, vals, distinct_keys) : sort_t<2, false>(keys.as<u64>(), nullptr, n, key_bits, stream, &keys, nullptr, distinct_keys);
}
int dev_sort(uint64_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t nw, uint32_t key_bits, hipStream_t stream) {
    if (nw != 1 && nw != 2) { set_error("key_words must be 1 or 2"); return KATOME_E_ARG; }
    if (key_bits == 0 || key_bits > 64 * nw) { set_error("key_bits out of range"); return KATOME_E_ARG; }
    if (nw == 1) return d_vals ? sort_t<1, true>(d_keys, d_vals, n, key_bits, stream) : sort_t<1, false>(d_keys, nullptr, n, key_bits, stream);
    return d_vals ? sort_t<2, true>(d_keys, d_vals, n, key_bits, stream) : sort_t<2, false>(d_keys, nullptr, n, key_bits, stream);
}

// group records by owner rank; invalid records go last (part n_parts) and are not counted.
// Optional u32 values travel with their records.
int dev_partition(const uint64_t* d_in, const uint32_t* v_in, uint64_t n, uint32_t nw, uint32_t n_parts, uint64_t* d_out,
                  uint32_t* v_out, uint64_t* h_counts, hipStream_t stream, uint32_t core_shift, uint32_t core_bases, uint32_t minimizer) {
    if (core_bases && (core_shift + 2 * core_bases > 64u * nw || 2 * core_bases > 190)) { set_error("partition: core outside the key"); return KATOME_E_ARG; }
    if (minimizer && (minimizer > 16 || minimizer > core_bases)) { set_error("partition: minimizer longer than the core (or than 16 bases)"); return KATOME_E_ARG; }
    if (n_parts == 0 || n_parts >= (u32)RADIX) { set_error("n_parts must be 1..255"); return KATOME_E_ARG; }
    if (nw < 1 || nw > 3) { set_error("key_words must be 1..3"); return KATOME_E_ARG; }
    if ((v_in == nullptr) != (v_out == nullptr)) { set_error("partition: values in and out must both be given"); return KATOME_E_ARG; }
    for (u32 p = 0; p < n_parts; ++p) h_counts[p] = 0;
    if (n == 0) return KATOME_OK;
    PassBuffers pb;
    KCHECK(pb.init(n, (int)nw, stream));
    if (nw == 1) {
        OwnerDigit<1> dg{n_parts, core_shift, core_bases, minimizer};
        if (v_in) KCHECK((radix_pass<1, true>(d_in, v_in, n, dg, d_out, v_out, pb, stream)));
        else      KCHECK((radix_pass<1, false>(d_in, nullptr, n, dg, d_out, nullptr, pb, stream)));
    } else if (nw == 2) {
        OwnerDigit<2> dg{n_parts, core_shift, core_bases, minimizer};
        if (v_in) KCHECK((radix_pass<2, true>(d_in, v_in, n, dg, d_out, v_out, pb, stream)));
        else      KCHECK((radix_pass<2, false>(d_in, nullptr, n, dg, d_out, nullptr, pb, stream)));
    } else {                          // three-word tiles (k > 32 with a useful span: C5's 90-mers)
        OwnerDigit<3> dg{n_parts, core_shift, core_bases, minimizer};
        if (v_in) KCHECK((radix_pass<3, true>(d_in, v_in, n, dg, d_out, v_out, pb, stream)));
        else      KCHECK((radix_pass<3, false>(d_in, nullptr, n, dg, d_out, nullptr, pb, stream)));
    }
    u64 totals[RADIX];
    KCHECK_HIP(hipMemcpyAsync(totals, pb.totals.p, sizeof totals, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    for (u32 p = 0; p < n_parts; ++p) h_counts[p] = totals[p];
    return KATOME_OK;
}

// supermer records (two words, owner inside: kmer_bits.h) grouped by owner; slots left invalid are dropped
int dev_partition_supermers(const uint64_t* d_in, uint64_t n, uint32_t n_parts, uint64_t* d_out, uint64_t* h_counts, hipStream_t stream) {
    if (n_parts == 0 || n_parts > 16) { set_error("supermers: 1..16 owners"); return KATOME_E_ARG; }
    for (u32 p = 0; p < n_parts; ++p) h_counts[p] = 0;
    if (n == 0) return KATOME_OK;
    PassBuffers pb;
    KCHECK(pb.init(n, 2, stream));
    SupermerOwnerDigit dg{n_parts};
    KCHECK((radix_pass<2, false>(d_in, nullptr, n, dg, d_out, nullptr, pb, stream)));
    u64 totals[RADIX];
    KCHECK_HIP(hipMemcpyAsync(totals, pb.totals.p, sizeof totals, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    for (u32 p = 0; p < n_parts; ++p) h_counts[p] = totals[p];
    return KATOME_OK;
}

// group u64 values (with their u32 companions) by the range they fall into: part p = [bounds[p-1], bounds[p]), d_bounds on the
// device (n_parts - 1 of them, ascending).  Stable.  Used to spread sequence numbers over the ranks (dist.hip global_rank).
int dev_partition_range(const uint64_t* d_vals, const uint32_t* idx_in, uint64_t n, const uint64_t* d_bounds, uint32_t n_parts,
                        uint64_t* d_out, uint32_t* idx_out, uint64_t* h_counts, hipStream_t stream) {
    if (n_parts == 0 || n_parts >= (u32)RADIX) { set_error("n_parts must be 1..255"); return KATOME_E_ARG; }
    for (u32 p = 0; p < n_parts; ++p) h_counts[p] = 0;
    if (n == 0) return KATOME_OK;
    PassBuffers pb;
    KCHECK(pb.init(n, 1, stream));
    RangeDigit dg{d_bounds, n_parts};
    KCHECK((radix_pass<1, true>(d_vals, idx_in, n, dg, d_out, idx_out, pb, stream)));
    u64 totals[RADIX];
    KCHECK_HIP(hipMemcpyAsync(totals, pb.totals.p, sizeof totals, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    for (u32 p = 0; p < n_parts; ++p) h_counts[p] = totals[p];
    return KATOME_OK;
}

// order records by the table region they hash to (1 or 2 stable 8-bit passes over the top hash bits), so
// that the insert kernel that follows works through the table one cache-sized region at a time.
// Result lands in `bufs[passes & 1]` where bufs = {scratch_a, scratch_b}; returns that pointer.
template <int NW>
static int region_order_t(const u64* d_in, const u32* w_in, u64 n, int passes, u64* ka, u64* kb, u32* wa, u32* wb,
                          const u64** k_out, const u32** w_out, hipStream_t stream, const u32* first_counts = nullptr) {
    PassBuffers pb;
    KCHECK(pb.init(n, NW, stream));
    // (first_counts: the first pass's digit counts per tile, [ceil(n / dev_sort_tile_keys)][256], made while the records were written)
    if (first_counts) KCHECK_HIP(hipMemcpyAsync(pb.counts.p, first_counts, pb.nblocks * RADIX * sizeof(u32), hipMemcpyDeviceToDevice, stream));
    const u64* kin = d_in; const u32* win = w_in;
    u64* kdst[2] = {ka, kb}; u32* wdst[2] = {wa, wb};
    for (int p = 0; p < passes; ++p) {
        HashDigit<NW> dg{(u32)(64 - 8 * (passes - p))};     // least significant region byte first
        const bool have = p == 0 && first_counts != nullptr;
        if (p == 0 && unstable_first()) {          // (nothing is ordered yet: the first pass need not be stable)
            if (w_in) KCHECK((radix_pass<NW, true, HashDigit<NW>, false>(kin, win, n, dg, kdst[p & 1], wdst[p & 1], pb, stream, have)));
            else      KCHECK((radix_pass<NW, false, HashDigit<NW>, false>(kin, nullptr, n, dg, kdst[p & 1], nullptr, pb, stream, have)));
        } else
        if (w_in) KCHECK((radix_pass<NW, true>(kin, win, n, dg, kdst[p & 1], wdst[p & 1], pb, stream, have)));
        else      KCHECK((radix_pass<NW, false>(kin, nullptr, n, dg, kdst[p & 1], nullptr, pb, stream, have)));
        kin = kdst[p & 1]; win = w_in ? wdst[p & 1] : nullptr;
    }
    *k_out = kin; *w_out = win;
    return KATOME_OK;
}
// (k-mer, count) records of one to three words ordered by the top 16 bits of the k-mer's hash, for the counting in LDS (table.hip): two
// stable 8-bit passes.  The result is where *k_out / *w_out point (one of the two buffer pairs); *group_bits = 16.
int dev_hash_order(const uint64_t* d_in, const uint32_t* w_in, uint64_t n, uint32_t nw, uint64_t* ka, uint64_t* kb, uint32_t* wa, uint32_t* wb,
                   const uint64_t** k_out, const uint32_t** w_out, uint32_t* group_bits, hipStream_t stream, const uint32_t* first_counts) {
    *group_bits = 16;
    if (nw == 1) return region_order_t<1>(d_in, w_in, n, 2, ka, kb, wa, wb, k_out, w_out, stream, first_counts);
    if (nw == 3) return region_order_t<3>(d_in, w_in, n, 2, ka, kb, wa, wb, k_out, w_out, stream, first_counts);      // (tiles of 64..95 bases)
    return region_order_t<2>(d_in, w_in, n, 2, ka, kb, wa, wb, k_out, w_out, stream, first_counts);
}
// records per tile of a partition pass over records of nw words, and the digit of dev_hash_order's first pass (for a kernel that
// writes such records and counts that pass's digits per tile as it goes: table.hip, list_to_records_kernel)
uint32_t dev_sort_tile_keys(uint32_t nw) { return nw == 1 ? SortTile<1>::KEYS : nw == 2 ? SortTile<2>::KEYS : SortTile<3>::KEYS; }
// records of nwk + 1 words (k-mer, tag) with their counts, ordered by the top 16 bits of the K-MER's hash (two stable passes)
template <int NW>
static int tagged_order_t(const u64* d_in, const u32* w_in, u64 n, u64* ka, u64* kb, u32* wa, u32* wb, const u64** k_out, const u32** w_out, hipStream_t stream,
                          const u32* first_counts = nullptr) {
    PassBuffers pb;
    KCHECK(pb.init(n, NW, stream));
    if (first_counts) KCHECK_HIP(hipMemcpyAsync(pb.counts.p, first_counts, pb.nblocks * RADIX * sizeof(u32), hipMemcpyDeviceToDevice, stream));
    const u64* kin = d_in; const u32* win = w_in;
    u64* kdst[2] = {ka, kb}; u32* wdst[2] = {wa, wb};
    for (int p = 0; p < 2; ++p) {
        HashTaggedDigit<NW> dg{(u32)(64 - 8 * (2 - p))};
        const bool have = p == 0 && first_counts != nullptr;       // (counted by whoever wrote the records: table.hip list_to_tagged_records_kernel)
        if (p == 0 && unstable_first()) {
            if (w_in) KCHECK((radix_pass<NW, true, HashTaggedDigit<NW>, false>(kin, win, n, dg, kdst[p & 1], wdst[p & 1], pb, stream, have)));
            else      KCHECK((radix_pass<NW, false, HashTaggedDigit<NW>, false>(kin, nullptr, n, dg, kdst[p & 1], nullptr, pb, stream, have)));
        } else
        if (w_in) KCHECK((radix_pass<NW, true>(kin, win, n, dg, kdst[p & 1], wdst[p & 1], pb, stream, have)));
        else      KCHECK((radix_pass<NW, false>(kin, nullptr, n, dg, kdst[p & 1], nullptr, pb, stream, have)));      // (records that count once each)
        kin = kdst[p & 1]; win = w_in ? wdst[p & 1] : nullptr;
    }
    *k_out = kin; *w_out = win;
    return KATOME_OK;
}
// one-word (k-mer, count) records ordered by the top 16 bits of the hash of their core (see CoreHashDigit)
int dev_hash_order_core(const uint64_t* d_in, const uint32_t* w_in, uint64_t n, uint32_t core_shift, uint32_t core_bases, uint64_t* ka, uint64_t* kb,
                        uint32_t* wa, uint32_t* wb, const uint64_t** k_out, const uint32_t** w_out, uint32_t* group_bits, hipStream_t stream) {
    *group_bits = CORE_GROUP_BITS;
    PassBuffers pb;
    KCHECK(pb.init(n, 1, stream));
    const u64* kin = d_in; const u32* win = w_in;
    u64* kdst[2] = {ka, kb}; u32* wdst[2] = {wa, wb};
    for (int p = 0; p < 2; ++p) {
        CoreHashDigit<1> dg{(u32)(64 - 8 * (2 - p)), core_shift, core_bases};
        KCHECK((radix_pass<1, true>(kin, win, n, dg, kdst[p & 1], wdst[p & 1], pb, stream)));
        kin = kdst[p & 1]; win = wdst[p & 1];
    }
    *k_out = kin; *w_out = win;
    return KATOME_OK;
}
int dev_hash_order_tagged(const uint64_t* d_in, const uint32_t* w_in, uint64_t n, uint32_t nwk, uint64_t* ka, uint64_t* kb, uint32_t* wa, uint32_t* wb,
                          const uint64_t** k_out, const uint32_t** w_out, uint32_t* group_bits, hipStream_t stream, const uint32_t* first_counts) {
    *group_bits = 16;
    if (nwk == 1) return tagged_order_t<2>(d_in, w_in, n, ka, kb, wa, wb, k_out, w_out, stream, first_counts);
    if (nwk == 2) return tagged_order_t<3>(d_in, w_in, n, ka, kb, wa, wb, k_out, w_out, stream, first_counts);
    set_error("tagged records: k-mers of one or two words");
    return KATOME_E_UNSUPPORTED;
}
int dev_region_order(const uint64_t* d_in, const uint32_t* w_in, uint64_t n, uint32_t nw, int passes, uint64_t* ka, uint64_t* kb,
                     uint32_t* wa, uint32_t* wb, const uint64_t** k_out, const uint32_t** w_out, hipStream_t stream) {
    if (nw == 1) return region_order_t<1>(d_in, w_in, n, passes, ka, kb, wa, wb, k_out, w_out, stream);
    return region_order_t<2>(d_in, w_in, n, passes, ka, kb, wa, wb, k_out, w_out, stream);
}

// ---- unique -------------------------------------------------------------------------------------
constexpr int UNIQ_ITEMS = 8;
constexpr int UNIQ_TILE = BLOCK * UNIQ_ITEMS;

template <int NW> __device__ __forceinline__ bool is_head(const u64* keys, u64 i) {
    return i == 0 || !key_eq(load_key<NW>(keys, i), load_key<NW>(keys, i - 1));
}

__device__ __forceinline__ u32 block_excl_scan(u32 mine, u32* wsum /*[BLOCK/64]*/, u32& total) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { u32 v = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 woff = 0; total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) { if (w < (int)wave) woff += wsum[w]; total += wsum[w]; }
    __syncthreads();
    return woff + incl - mine;
}

template <int NW>
__global__ __launch_bounds__(BLOCK) void uniq_count_kernel(const u64* __restrict__ keys, u64 n, u32* __restrict__ block_counts) {
    __shared__ u32 wsum[BLOCK / 64];
    const u64 base = (u64)blockIdx.x * UNIQ_TILE + (u64)threadIdx.x * UNIQ_ITEMS;
    u32 mine = 0;
#pragma unroll
    for (int j = 0; j < UNIQ_ITEMS; ++j) if (base + j < n && is_head<NW>(keys, base + j)) ++mine;
    u32 total;
    (void)block_excl_scan(mine, wsum, total);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

// exclusive scan of m u32 counts into u64 offsets, one workgroup; offs[m] = grand total
template <class T>
__global__ __launch_bounds__(1024) void scan_counts_kernel(const T* __restrict__ counts, u64 m, u64* __restrict__ offs) {
    __shared__ u64 wsum[16];
    __shared__ u64 carry_s;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (u64 b0 = 0; b0 < m; b0 += 1024) {
        u64 i = b0 + tid;
        u64 v = i < m ? (u64)counts[i] : 0;
        u64 incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { u64 t = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u64 woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < (int)wave) woff += wsum[w]; tot += wsum[w]; }
        const u64 carry = carry_s;
        if (i < m) offs[i] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (tid == 0) offs[m] = carry_s;
}
// long arrays: every workgroup sums SCAN_CHUNK counts, one workgroup scans those sums, every workgroup scans its chunk
// again from its sum's offset (one workgroup walking 600 k counts alone took 2 ms, as long as a pass over 10 GB)
constexpr u32 SCAN_CHUNK = 4096;
__global__ __launch_bounds__(1024) void scan_chunk_sums_kernel(const u32* __restrict__ counts, u64 m, u64* __restrict__ sums) {
    __shared__ u64 wsum[16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK;
    u64 v = 0;
    for (u32 j = tid; j < SCAN_CHUNK; j += 1024) if (base + j < m) v += counts[base + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) wsum[wave] = v;
    __syncthreads();
    if (tid == 0) { u64 t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; sums[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void scan_chunks_kernel(const u32* __restrict__ counts, u64 m, const u64* __restrict__ chunk_offs,
                                                           u64* __restrict__ offs) {
    __shared__ u64 wsum[16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK + (u64)tid * (SCAN_CHUNK / 1024);       // 4 consecutive counts per thread
    u64 c[SCAN_CHUNK / 1024], mine = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_CHUNK / 1024; ++j) { c[j] = base + j < m ? (u64)counts[base + j] : 0; mine += c[j]; }
    u64 incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { u64 t = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u64 run = chunk_offs[blockIdx.x] + incl - mine;
    for (int w = 0; w < 16; ++w) if (w < (int)wave) run += wsum[w];
#pragma unroll
    for (u32 j = 0; j < SCAN_CHUNK / 1024; ++j) { if (base + j < m) offs[base + j] = run; run += c[j]; }
    if (blockIdx.x == gridDim.x - 1 && tid == 1023) offs[m] = chunk_offs[gridDim.x];
}

template <int NW>
__global__ __launch_bounds__(BLOCK) void uniq_write_kernel(const u64* __restrict__ keys, u64 n, const u64* __restrict__ block_offs,
                                                            u64* __restrict__ out) {
    __shared__ u32 wsum[BLOCK / 64];
    const u64 base = (u64)blockIdx.x * UNIQ_TILE + (u64)threadIdx.x * UNIQ_ITEMS;
    bool head[UNIQ_ITEMS]; u32 mine = 0;
#pragma unroll
    for (int j = 0; j < UNIQ_ITEMS; ++j) { head[j] = base + j < n && is_head<NW>(keys, base + j); mine += head[j]; }
    u32 total;
    u64 pos = block_offs[blockIdx.x] + block_excl_scan(mine, wsum, total);
#pragma unroll
    for (int j = 0; j < UNIQ_ITEMS; ++j) if (head[j]) { store_key<NW>(out, pos, load_key<NW>(keys, base + j)); ++pos; }
}

int dev_scan_counts(const uint32_t* d_counts, uint64_t m, uint64_t* d_offs, hipStream_t stream) {
    if (m <= 16 * SCAN_CHUNK) {
        hipLaunchKernelGGL(scan_counts_kernel<u32>, dim3(1), dim3(1024), 0, stream, d_counts, m, d_offs);
    } else {
        const u64 chunks = (m + SCAN_CHUNK - 1) / SCAN_CHUNK;
        DevBuf sums(stream), chunk_offs(stream);
        KCHECK(sums.alloc(chunks * 8 + 16)); KCHECK(chunk_offs.alloc((chunks + 1) * 8 + 16));
        hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3((unsigned)chunks), dim3(1024), 0, stream, d_counts, m, sums.as<u64>());
        hipLaunchKernelGGL(scan_counts_kernel<u64>, dim3(1), dim3(1024), 0, stream, sums.as<u64>(), chunks, chunk_offs.as<u64>());
        hipLaunchKernelGGL(scan_chunks_kernel, dim3((unsigned)chunks), dim3(1024), 0, stream, d_counts, m, chunk_offs.as<u64>(), d_offs);
    }
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

int dev_unique(uint64_t* d_keys, uint64_t n, uint32_t nw, uint64_t* n_out, hipStream_t stream) {
    *n_out = n;
    if (n < 2) return KATOME_OK;
    const u64 nblocks = (n + UNIQ_TILE - 1) / UNIQ_TILE;
    if (nblocks > 0x7fffffffull) { set_error("unique: too many keys"); return KATOME_E_ARG; }
    DevBuf counts(stream), offs(stream), tmp(stream);
    KCHECK(counts.alloc(nblocks * 4));
    KCHECK(offs.alloc((nblocks + 1) * 8));
    KCHECK(tmp.alloc(n * 8 * nw));
    if (nw == 1) hipLaunchKernelGGL(uniq_count_kernel<1>, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, d_keys, n, counts.as<u32>());
    else         hipLaunchKernelGGL(uniq_count_kernel<2>, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, d_keys, n, counts.as<u32>());
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, counts.as<u32>(), nblocks, offs.as<u64>());
    if (nw == 1) hipLaunchKernelGGL(uniq_write_kernel<1>, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, d_keys, n, offs.as<u64>(), tmp.as<u64>());
    else         hipLaunchKernelGGL(uniq_write_kernel<2>, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, d_keys, n, offs.as<u64>(), tmp.as<u64>());
    KCHECK_HIP(hipGetLastError());
    KCHECK_HIP(hipMemcpyAsync(n_out, offs.as<u64>() + nblocks, 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    KCHECK_HIP(hipMemcpyAsync(d_keys, tmp.p, *n_out * 8 * nw, hipMemcpyDeviceToDevice, stream));
    return KATOME_OK;
}

// ---- rank of query keys in a sorted unique array ------------------------------------------------
// A bucket index over the top B bits (index[b] = first position whose top bits are >= b) narrows
// each lookup to a few consecutive keys; a binary search inside the bucket finishes it.
template <int NW> __device__ __forceinline__ u32 top_bits(const Key<NW>& k, u32 key_bits, u32 B) { return key_digit(k, key_bits - B, B); }

template <int NW>
__global__ __launch_bounds__(BLOCK) void bucket_index_kernel(const u64* __restrict__ sorted, u64 n, u32 key_bits, u32 B, u64* __restrict__ index) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i <= n; i += (u64)gridDim.x * BLOCK) {
        long long prev = i > 0 ? (long long)top_bits(load_key<NW>(sorted, i - 1), key_bits, B) : -1ll;
        long long cur = i < n ? (long long)top_bits(load_key<NW>(sorted, i), key_bits, B) : (1ll << B);
        for (long long b = prev + 1; b <= cur; ++b) index[b] = i;
    }
}

template <int NW>
__global__ __launch_bounds__(BLOCK) void rank_kernel(const u64* __restrict__ sorted, u64 n, u32 key_bits, u32 B,
                                                      const u64* __restrict__ index, const u64* __restrict__ q, u64 nq,
                                                      u64* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < nq; i += (u64)gridDim.x * BLOCK) {
        Key<NW> key = load_key<NW>(q, i);
        u32 b = top_bits(key, key_bits, B);
        u64 lo = index[b], hi = index[b + 1];
        while (lo < hi) {
            u64 mid = (lo + hi) >> 1;
            if (key_lt(load_key<NW>(sorted, mid), key)) lo = mid + 1; else hi = mid;
        }
        out[i] = (lo < n && key_eq(load_key<NW>(sorted, lo), key)) ? lo : ~0ull;
    }
}

int dev_rank(const uint64_t* d_sorted, uint64_t n_sorted, uint32_t nw, uint32_t key_bits, const uint64_t* d_q, uint64_t nq,
             uint64_t* d_out, hipStream_t stream) {
    if (nq == 0) return KATOME_OK;
    u32 B = 1;
    while ((2ull << B) <= n_sorted / 8 && B < 27) ++B;
    if (B > key_bits) B = key_bits;
    DevBuf index(stream);
    KCHECK(index.alloc(((1ull << B) + 2) * 8));
    dim3 block(BLOCK);
    if (nw == 1) {
        hipLaunchKernelGGL(bucket_index_kernel<1>, dim3(grid_for(n_sorted + 1, BLOCK)), block, 0, stream, d_sorted, n_sorted, key_bits, B, index.as<u64>());
        hipLaunchKernelGGL(rank_kernel<1>, dim3(grid_for(nq, BLOCK, 256u * 32u)), block, 0, stream, d_sorted, n_sorted, key_bits, B, index.as<u64>(), d_q, nq, d_out);
    } else {
        hipLaunchKernelGGL(bucket_index_kernel<2>, dim3(grid_for(n_sorted + 1, BLOCK)), block, 0, stream, d_sorted, n_sorted, key_bits, B, index.as<u64>());
        hipLaunchKernelGGL(rank_kernel<2>, dim3(grid_for(nq, BLOCK, 256u * 32u)), block, 0, stream, d_sorted, n_sorted, key_bits, B, index.as<u64>(), d_q, nq, d_out);
    }
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

// ---- node numbering straight from the sorted edge list --------------------------------------------
// Edges are sorted by packed k-mer, so their source (k-1)-mers (key >> 2) are sorted too: the nodes
// that have out-edges are the run heads of that sequence -- no sort needed.  Targets are looked up in
// that list; the few that are absent (nodes without out-edges: read ends nothing continues) are
// collected, sorted and appended.  Node ids: sources in ascending key order, then the out-edge-less
// nodes in ascending key order (add_fasta_node, pt_graph.rs:142-154, numbers in first-seen order; no
// order is pinned by the reference -- DESIGN.md section 1).
template <int NW> __device__ __forceinline__ bool is_src_head(const u64* keys, u64 i) {
    return i == 0 || !key_eq(key_shr(load_key<NW>(keys, i), 2), key_shr(load_key<NW>(keys, i - 1), 2));
}
template <int NW>
__global__ __launch_bounds__(BLOCK) void src_count_kernel(const u64* __restrict__ keys, u64 n, u32* __restrict__ block_counts) {
    __shared__ u32 wsum[BLOCK / 64];
    const u64 base = (u64)blockIdx.x * UNIQ_TILE + threadIdx.x;          // rows of BLOCK consecutive edges: coalesced loads
    u32 mine = 0;
#pragma unroll
    for (int j = 0; j < UNIQ_ITEMS; ++j) if (base + (u64)j * BLOCK < n && is_src_head<NW>(keys, base + (u64)j * BLOCK)) ++mine;
    u32 total;
    (void)block_excl_scan(mine, wsum, total);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}
// writes the distinct sources (= node keys) and every edge's source id.  Rows of BLOCK consecutive edges are
// taken one after the other (coalesced loads and stores); a ballot scan per row keeps the running head count.
template <int NW>
__global__ __launch_bounds__(BLOCK) void src_write_kernel(const u64* __restrict__ keys, u64 n, const u64* __restrict__ block_offs,
                                                           u64* __restrict__ nodes, u64* __restrict__ edge_src, u64* __restrict__ seg_edge, u32 seg_nodes) {
    __shared__ u32 wtot[UNIQ_ITEMS][BLOCK / 64];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 base = (u64)blockIdx.x * UNIQ_TILE;
    bool head[UNIQ_ITEMS]; u32 before[UNIQ_ITEMS];
#pragma unroll
    for (int j = 0; j < UNIQ_ITEMS; ++j) {
        const u64 e = base + (u64)j * BLOCK + threadIdx.x;
        head[j] = e < n && is_src_head<NW>(keys, e);
        const u64 m = __ballot(head[j]);
        before[j] = __popcll(m & (lane ? (~0ull >> (64 - lane)) : 0ull));
        if (lane == 0) wtot[j][wave] = __popcll(m);
    }
    __syncthreads();
    u64 carry = block_offs[blockIdx.x];
#pragma unroll
    for (int j = 0; j < UNIQ_ITEMS; ++j) {
        u32 woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) { if (w < (int)wave) woff += wtot[j][w]; total += wtot[j][w]; }
        const u64 e = base + (u64)j * BLOCK + threadIdx.x;
        if (e < n) {
            const u64 pos = carry + woff + before[j];           // heads strictly before this edge
            if (head[j]) {
                store_key<NW>(nodes, pos, key_shr(load_key<NW>(keys, e), 2));
                if (seg_edge && pos % seg_nodes == 0) seg_edge[pos / seg_nodes] = e;      // first out-edge of every seg_nodes-th source
            }
            edge_src[e] = head[j] ? pos : pos - 1;
        }
        carry += total;
    }
}
// edge_dst[e] = position of the edge's target in `nodes`, or ~0 when it is not a source of any edge
template <int NW>
__global__ __launch_bounds__(BLOCK) void dst_rank_kernel(const u64* __restrict__ nodes, u64 n_nodes, u32 key_bits, u32 B,
                                                          const u64* __restrict__ index, const u64* __restrict__ keys, u64 n,
                                                          u32 k, u64* __restrict__ edge_dst, u64* __restrict__ n_missing) {
    u32 miss = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Key<NW> key = target_node(load_key<NW>(keys, i), k);
        u32 b = top_bits(key, key_bits, B);
        u64 lo = index[b], hi = index[b + 1];
        while (lo < hi) {
            u64 mid = (lo + hi) >> 1;
            if (key_lt(load_key<NW>(nodes, mid), key)) lo = mid + 1; else hi = mid;
        }
        const bool found = lo < n_nodes && key_eq(load_key<NW>(nodes, lo), key);
        edge_dst[i] = found ? lo : ~0ull;
        miss += !found;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) miss += __shfl_down(miss, o, 64);
    if ((threadIdx.x & 63) == 0 && miss) atomicAdd((unsigned long long*)n_missing, (unsigned long long)miss);
}
// gather the targets that were not found (unordered; they are sorted afterwards).  One cursor atomic per
// workgroup tile: the misses are sparse, and one atomic per wave on a single address serialises.
template <int NW>
__global__ __launch_bounds__(BLOCK) void missing_gather_kernel(const u64* __restrict__ keys, u64 n, u32 k, const u64* __restrict__ edge_dst,
                                                                u64* __restrict__ out, u64* __restrict__ out_edge, u64* cursor) {
    __shared__ u32 wsum[BLOCK / 64];
    __shared__ u64 block_base;
    const u64 tile = (u64)BLOCK * UNIQ_ITEMS;
    for (u64 t0 = (u64)blockIdx.x * tile; t0 < n; t0 += (u64)gridDim.x * tile) {
        bool miss[UNIQ_ITEMS]; u32 mine = 0;
#pragma unroll
        for (int j = 0; j < UNIQ_ITEMS; ++j) {
            const u64 i = t0 + (u64)j * BLOCK + threadIdx.x;
            miss[j] = i < n && edge_dst[i] == ~0ull;
            mine += miss[j];
        }
        u32 total;
        const u32 excl = block_excl_scan(mine, wsum, total);
        if (threadIdx.x == 0 && total) block_base = atomicAdd((unsigned long long*)cursor, (unsigned long long)total);
        __syncthreads();
        if (total) {
            u64 pos = block_base + excl;
#pragma unroll
            for (int j = 0; j < UNIQ_ITEMS; ++j)
                if (miss[j]) {
                    const u64 e = t0 + (u64)j * BLOCK + threadIdx.x;
                    store_key<NW>(out, pos, target_node(load_key<NW>(keys, e), k));
                    out_edge[pos] = e;
                    ++pos;
                }
        }
        __syncthreads();
    }
}
// second lookup, only for the edges whose target was not a source: id = n_sources + rank among the extra nodes
template <int NW>
__global__ __launch_bounds__(BLOCK) void missing_rank_kernel(const u64* __restrict__ miss_key, const u64* __restrict__ miss_edge, u64 n_miss,
                                                              const u64* __restrict__ extra, u64 n_extra, u64 n_sources,
                                                              u64* __restrict__ edge_dst) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n_miss; i += (u64)gridDim.x * BLOCK) {
        Key<NW> key = load_key<NW>(miss_key, i);
        u64 lo = 0, hi = n_extra;
        while (lo < hi) {
            u64 mid = (lo + hi) >> 1;
            if (key_lt(load_key<NW>(extra, mid), key)) lo = mid + 1; else hi = mid;
        }
        edge_dst[miss_edge[i]] = n_sources + lo;
    }
}

// ---- targets looked up by merging ----------------------------------------------------------------------------------
// Sorted by packed k-mer, the edges fall into four quarters by their first base, and inside a quarter the TARGETS (the low
// 2(k-1) bits) ascend too.  The sources (ascending) are cut into segments of DST_SEG nodes; a workgroup stages its segment
// in LDS and walks, quarter by quarter, the one contiguous stretch of edges whose targets lie in the segment's key range:
// sources and edges are each read once, coalesced, and a target costs a binary search in LDS -- instead of a bucket look-up
// and a binary search in HBM per edge (dst_rank_kernel) and a second pass that collects the targets not found
// (missing_gather_kernel): those are staged in LDS and leave with one cursor atomic per MISS_CAP of them.
#ifndef KATOME_DST_SEG
#define KATOME_DST_SEG 2048
#endif
constexpr u64 DST_IN1 = 1ull << 40;                      // first-seen order: mark in edge_dst, "the target has this in-edge only"
constexpr u64 DST_FD = 1ull << 41;                       // ... and "this edge is the first to touch its target" (it introduces the node)
constexpr u64 DST_MARKS = DST_IN1 | DST_FD;
constexpr u32 DST_SEG = KATOME_DST_SEG;
// edges per thread and trip (loads and searches in flight): 2 for one-word k-mers, 4 for two-word ones -- measured both ways at C3
// (11.4 ms with 2, 15.4 with 4) and at k = 40 / 50 M reads (12.0 with 2, 9.2 with 4); -DKATOME_DST_ROWS=n sets both
template <int NW> struct DstRows {
#ifdef KATOME_DST_ROWS
    static constexpr u32 value = KATOME_DST_ROWS;
#else
    static constexpr u32 value = NW == 1 ? 2 : 4;
#endif
};
template <int NW> struct MissCap { static constexpr u32 value = (DstRows<NW>::value > 2 ? 2048 : 1024) / NW; };
static_assert(MissCap<1>::value >= DstRows<1>::value * BLOCK && MissCap<2>::value >= DstRows<2>::value * BLOCK, "a whole trip of misses fits the staging buffer");

template <int NW> __device__ __forceinline__ Key<NW> with_quarter(Key<NW> node, u32 q, u32 node_bits) {
    if (NW == 1) node.w[0] |= (u64)q << node_bits;
    else if (node_bits >= 64) node.w[0] |= (u64)q << (node_bits - 64);
    else { node.w[NW - 1] |= (u64)q << node_bits; node.w[0] |= (u64)q >> (64 - node_bits); }      // (k = 32: node_bits = 62)
    return node;
}
template <int NW> __device__ __forceinline__ u64 lower_bound_keys(const u64* __restrict__ keys, u64 n, const Key<NW>& x) {
    u64 lo = 0, hi = n;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (key_lt(load_key<NW>(keys, mid), x)) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// seg[q][s] = first edge of quarter q whose target is not below the first source of segment s (s = 0: the quarter's start;
// s = n_seg: its end)
template <int NW>
__global__ __launch_bounds__(BLOCK) void dst_seg_kernel(const u64* __restrict__ nodes, const u64* __restrict__ keys, u64 n, u32 node_bits,
                                                         u64 n_seg, u64* __restrict__ seg) {
    const u64 total = 4 * (n_seg + 1);
    for (u64 t = (u64)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (u64)gridDim.x * BLOCK) {
        const u32 q = (u32)(t / (n_seg + 1));
        const u64 s = t % (n_seg + 1);
        Key<NW> v;
#pragma unroll
        for (int j = 0; j < NW; ++j) v.w[j] = 0;
        u64 pos;
        if (s == n_seg) pos = q == 3 ? n : lower_bound_keys<NW>(keys, n, with_quarter(v, q + 1, node_bits));
        else {
            if (s) v = load_key<NW>(nodes, s * DST_SEG);
            pos = lower_bound_keys<NW>(keys, n, with_quarter(v, q, node_bits));
        }
        seg[t] = pos;
    }
}
// edge_dst[e] = position of the edge's target among the sources, or ~0; the targets not found go to miss_key / miss_edge
// (unordered, as many as fit miss_cap) and are counted in *cursor
// FIRST (first-seen order): seq[e] = sequence number of the edge's first insertion; node_first[v] (holding the source-role
// minimum, 2*seq, or all-ones) is lowered to the first touch as a target, 2*seq + 1, with atomics in LDS only
template <int NW, bool FIRST>
__global__ __launch_bounds__(BLOCK) void dst_merge_kernel(const u64* __restrict__ nodes, u64 n_src, const u64* __restrict__ keys, u32 k,
                                                           u64 n_seg, const u64* __restrict__ seg, u64* __restrict__ edge_dst,
                                                           u64* __restrict__ miss_key, u64* __restrict__ miss_edge, u64 miss_cap, u64* cursor,
                                                           const u64* __restrict__ seq, u64* __restrict__ node_first,
                                                           const u64* __restrict__ edge_src, const u64* __restrict__ seg_edge, u64 n_edges) {
    constexpr u32 MISS_CAP = MissCap<NW>::value, DST_ROWS = DstRows<NW>::value;
    extern __shared__ u64 lmem[];
    u64* ls = lmem;                                     // [DST_SEG * NW] the segment's sources
    u64* lmk = ls + DST_SEG * NW;                       // [MISS_CAP * NW] + [MISS_CAP]: targets not found, and their edges
    u64* lme = lmk + MISS_CAP * NW;
    u64* lfirst = lme + MISS_CAP;                       // FIRST: [DST_SEG] first touch as a target
    u32* lonce = reinterpret_cast<u32*>(lfirst + DST_SEG);   // FIRST: two bitmaps [DST_SEG / 32]: has an in-edge, has several
    u32* lmore = lonce + DST_SEG / 32;
    __shared__ u32 lmiss;
    __shared__ u64 lbase;
    const u32 tid = threadIdx.x;
    auto flush = [&]() {                                // (called by every thread, between barriers)
        const u32 m = lmiss;
        if (tid == 0 && m) lbase = atomicAdd((unsigned long long*)cursor, (unsigned long long)m);
        __syncthreads();
        if (m) {
            const u64 base = lbase;
            for (u32 j = tid; j < m; j += BLOCK)
                if (base + j < miss_cap) {
                    Key<NW> x;
#pragma unroll
                    for (int w = 0; w < NW; ++w) x.w[w] = lmk[j * NW + w];
                    store_key<NW>(miss_key, base + j, x);
                    miss_edge[base + j] = lme[j];
                }
        }
        __syncthreads();
        if (tid == 0) lmiss = 0;
        __syncthreads();
    };
    for (u64 sg = blockIdx.x; sg < n_seg; sg += gridDim.x) {
        const u64 a = sg * DST_SEG;
        const u32 cnt = (u32)((n_src - a) < (u64)DST_SEG ? (n_src - a) : (u64)DST_SEG);
        // the four stretches (one per quarter) are walked as one list of `total` edges, so that a trip is full whatever the
        // quarters' sizes, and the loads of the next trip are issued before this one's searches (a segment was a chain of
        // ~9 memory round trips: two per quarter, the second nearly empty)
        const u64 lo0 = seg[sg], lo1 = seg[(n_seg + 1) + sg], lo2 = seg[2 * (n_seg + 1) + sg], lo3 = seg[3 * (n_seg + 1) + sg];
        const u64 c1 = seg[sg + 1] - lo0, c2 = c1 + (seg[(n_seg + 1) + sg + 1] - lo1), c3 = c2 + (seg[2 * (n_seg + 1) + sg + 1] - lo2);
        const u64 total = c3 + (seg[3 * (n_seg + 1) + sg + 1] - lo3);
        auto edge_of = [&](u64 v) -> u64 { return v < c1 ? lo0 + v : v < c2 ? lo1 + (v - c1) : v < c3 ? lo2 + (v - c2) : lo3 + (v - c3); };
        Key<NW> e[DST_ROWS], en[DST_ROWS]; u64 sq[DST_ROWS], sqn[DST_ROWS], ei[DST_ROWS], ein[DST_ROWS];
        auto fetch = [&](u64 c, Key<NW>* ek, u64* es, u64* ex) {
#pragma unroll
            for (u32 r = 0; r < DST_ROWS; ++r) {
                const u64 v = c + (u64)r * BLOCK + tid;
#pragma unroll
                for (int w = 0; w < NW; ++w) ek[r].w[w] = 0;
                es[r] = 0; ex[r] = ~0ull;
                if (v < total) { const u64 i = edge_of(v); ex[r] = i; ek[r] = load_key<NW>(keys, i); if (FIRST) es[r] = seq[i]; }
            }
        };
        fetch(0, e, sq, ei);                            // (in flight together with the segment's sources below)
        Key<NW> stage[DST_SEG / BLOCK];
#pragma unroll
        for (u32 r = 0; r < DST_SEG / BLOCK; ++r) { const u32 j = r * BLOCK + tid; if (j < cnt) stage[r] = load_key<NW>(nodes, a + j); }
#pragma unroll
        for (u32 r = 0; r < DST_SEG / BLOCK; ++r) {
            const u32 j = r * BLOCK + tid;
            if (j < cnt) {
#pragma unroll
                for (int w = 0; w < NW; ++w) ls[j * NW + w] = stage[r].w[w];
                if (FIRST) lfirst[j] = ~0ull;
            }
        }
        if (FIRST && tid < 2 * (DST_SEG / 32)) lonce[tid] = 0;          // (lmore follows lonce)
        if (tid == 0) lmiss = 0;
        __syncthreads();
        if (FIRST) {    // source role: the segment's out-edges are one stretch of the edge list; first touch 2 * seq (pt_graph.rs:180-185)
            const u64 e0 = seg_edge[sg], e1 = sg + 1 < n_seg ? seg_edge[sg + 1] : n_edges;
            for (u64 e = e0 + tid; e < e1; e += BLOCK)
                atomicMin((unsigned long long*)&lfirst[(u32)(edge_src[e] - a)], (unsigned long long)(2 * seq[e]));
        }
        for (u64 c = 0; c < total; c += (u64)DST_ROWS * BLOCK) {
            fetch(c + (u64)DST_ROWS * BLOCK, en, sqn, ein);
            // the searches of a thread's edges advance in lockstep, a fixed number of halving steps each (branch-free lower
            // bound): DST_ROWS independent LDS reads are in flight per step
            Key<NW> d[DST_ROWS]; u32 l[DST_ROWS];
#pragma unroll
            for (u32 r = 0; r < DST_ROWS; ++r) { d[r] = target_node(e[r], k); l[r] = 0; }
#pragma unroll
            for (u32 step = DST_SEG; step >= 1; step >>= 1) {
#pragma unroll
                for (u32 r = 0; r < DST_ROWS; ++r) {
                    const u32 idx = l[r] + step;
                    const bool in = idx <= cnt;
                    Key<NW> x;
#pragma unroll
                    for (int w = 0; w < NW; ++w) x.w[w] = ls[(in ? idx - 1 : 0) * NW + w];
                    if (in && key_lt(x, d[r])) l[r] = idx;
                }
            }
#pragma unroll
            for (u32 r = 0; r < DST_ROWS; ++r) {
                const u64 i = ei[r];
                if (i == ~0ull) continue;
                bool found = false;
                if (l[r] < cnt) {
                    Key<NW> x;
#pragma unroll
                    for (int w = 0; w < NW; ++w) x.w[w] = ls[l[r] * NW + w];
                    found = key_eq(x, d[r]);
                }
                if (found) {
                    edge_dst[i] = a + l[r];
                    if (FIRST) {
                        atomicMin((unsigned long long*)&lfirst[l[r]], (unsigned long long)(2 * sq[r] + 1));
                        const u32 bit = 1u << (l[r] & 31);
                        if (atomicOr(&lonce[l[r] >> 5], bit) & bit) atomicOr(&lmore[l[r] >> 5], bit);
                    }
                } else {
                    edge_dst[i] = ~0ull;
                    const u32 p = atomicAdd(&lmiss, 1u);
#pragma unroll
                    for (int w = 0; w < NW; ++w) lmk[p * NW + w] = d[r].w[w];
                    lme[p] = i;
                }
            }
            // room for another trip's misses?  Every wave must decide the same, so the count is read between two barriers: a
            // wave that ran ahead into the next trip could otherwise add to it before a slower one had looked (with the loads
            // prefetched that did happen: waves parted ways at flush()'s barriers and the grid never finished)
            __syncthreads();
            const u32 staged = lmiss;
            __syncthreads();
            if (staged + DST_ROWS * BLOCK > MISS_CAP) flush();
#pragma unroll
            for (u32 r = 0; r < DST_ROWS; ++r) { e[r] = en[r]; sq[r] = sqn[r]; ei[r] = ein[r]; }
        }
        flush();
        if (FIRST) {                                    // (flush ends with a barrier: the segment's minima are complete)
            for (u32 j = tid; j < cnt; j += BLOCK) node_first[a + j] = lfirst[j];      // the node's first touch, either role
            // (no barrier needed before the sweep below reads lfirst: flush ended with one, and nobody has written since)
            // second sweep: the edge that is the first to touch its target is marked (DST_FD: what the renumbering asks of every
            // edge, here without a look-up), and so is an edge whose target has no other in-edge (DST_IN1) -- with the matching
            // mark on the source side (one out-edge) the renumbering can tell the nodes nobody will ever look up (dev_assign_nodes)
            for (u64 v = tid; v < total; v += BLOCK) {              // (the thread that wrote edge_dst[i])
                const u64 i = edge_of(v);
                const u64 d = edge_dst[i];
                if (d == ~0ull) continue;
                const u32 l = (u32)(d - a), bit = 1u << (l & 31);
                u64 marks = (lmore[l >> 5] & bit) ? 0 : DST_IN1;
                if (lfirst[l] == 2 * seq[i] + 1) marks |= DST_FD;
                if (marks) edge_dst[i] = d | marks;
            }
            __syncthreads();
        }
    }
}
// the targets that are no source (their ids were written by missing_rank_kernel): first touch of these nodes
__global__ __launch_bounds__(BLOCK) void missing_first_kernel(const u64* __restrict__ miss_edge, u64 n_miss, const u64* __restrict__ edge_dst,
                                                               const u64* __restrict__ seq, u64* node_first) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n_miss; i += (u64)gridDim.x * BLOCK) {
        const u64 e = miss_edge[i];
        atomicMin((unsigned long long*)&node_first[edge_dst[e]], (unsigned long long)(2 * seq[e] + 1));
    }
}

// the distinct source (k-1)-mers of sorted edges (the run heads of key >> 2), ascending, and every edge's position among them
// (`slack`: room kept behind them in node_key, in nodes, for the caller to append to)
template <int NW>
static int source_ids_t(const u64* d_edge_key, u64 E, DevBuf& node_key, u64* edge_src, u64* n_src_out, hipStream_t stream, bool with_slack = false,
                        DevBuf* seg_edge = nullptr) {
    *n_src_out = 0;
    if (E == 0) { KCHECK(node_key.alloc(16, stream)); return KATOME_OK; }
    const u64 nblocks = (E + UNIQ_TILE - 1) / UNIQ_TILE;
    if (nblocks > 0x7fffffffull) { set_error("node numbering: too many edges"); return KATOME_E_ARG; }
    DevBuf counts(stream), offs(stream);
    KCHECK(counts.alloc(nblocks * 4));
    KCHECK(offs.alloc((nblocks + 1) * 8));
    {
        KernelScope ks(K_SRC_IDS, stream, E);
        hipLaunchKernelGGL(src_count_kernel<NW>, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, d_edge_key, E, counts.as<u32>());
        KCHECK(dev_scan_counts(counts.as<u32>(), nblocks, offs.as<u64>(), stream));      // (C3: 8e5 counts -- one workgroup walking them alone took 1.2 ms)
    }
    u64 n_src = 0;
    KCHECK_HIP(hipMemcpyAsync(&n_src, offs.as<u64>() + nblocks, 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    // (with_slack: the caller appends the nodes without out-edges -- usually a few percent -- instead of copying the lot)
    KCHECK(node_key.alloc((n_src + (with_slack ? n_src / 8 + (1u << 16) : 0) + 1) * 8 * NW, stream));
    if (seg_edge) KCHECK(seg_edge->alloc(((n_src + DST_SEG - 1) / DST_SEG + 1) * 8));        // first out-edge of every DST_SEG-th source
    {
        KernelScope ks(K_SRC_IDS, stream, E);
        hipLaunchKernelGGL(src_write_kernel<NW>, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, d_edge_key, E, offs.as<u64>(), node_key.as<u64>(), edge_src,
                           seg_edge ? seg_edge->as<u64>() : nullptr, DST_SEG);
    }
    KCHECK_HIP(hipGetLastError());
    *n_src_out = n_src;
    return KATOME_OK;
}
int dev_source_ids(const uint64_t* d_edge_key, uint64_t n_edges, uint32_t k, DevBuf& node_key, uint64_t* d_edge_src, uint64_t* n_src,
                   hipStream_t stream) {
    if (key_words_for_k(k) == 1) return source_ids_t<1>(d_edge_key, n_edges, node_key, d_edge_src, n_src, stream);
    return source_ids_t<2>(d_edge_key, n_edges, node_key, d_edge_src, n_src, stream);
}


// seq + node_first (first-seen order, both or neither): node_first[v] = the first touch of node v, 2*seq as a source, 2*seq + 1
// as a target; left empty when the merging look-up is switched off (the caller then runs dev_node_first)
template <int NW>
static int node_ids_t(const u64* d_edge_key, u64 E, u32 k, DevBuf& node_key, u64* edge_src, u64* edge_dst, u64* n_nodes,
                      hipStream_t stream, const u64* seq = nullptr, DevBuf* node_first = nullptr, u64* n_marked = nullptr) {
    *n_nodes = 0;
    if (n_marked) *n_marked = 0;
    if (node_first) node_first->release();
    if (E == 0) { KCHECK(node_key.alloc(16, stream)); return KATOME_OK; }
    const u32 node_bits = 2 * (k - 1);
    DevBuf aux(stream);
    KCHECK(aux.alloc(16));
    KCHECK_HIP(hipMemsetAsync(aux.p, 0, 16, stream));
    u64 n_src = 0;
    static const bool old_lookup = getenv("KATOME_DST_RANK") != nullptr;
    const bool first = seq && node_first && !old_lookup;
    DevBuf seg_edge(stream);
    KCHECK((source_ids_t<NW>(d_edge_key, E, node_key, edge_src, &n_src, stream, true, first ? &seg_edge : nullptr)));
    u64* nodes = node_key.as<u64>();
    // targets -> positions among the sources
    if (first && n_marked) *n_marked = n_src;            // (edge_dst carries the merge's marks for the targets that are sources)
    DevBuf miss_key(stream), miss_edge(stream);
    u64 miss_cap = 0, n_missing = 0;
    if (!old_lookup) {
        // merged against the sources segment by segment; the targets that are no source are set aside on the way
        const u64 n_seg = (n_src + DST_SEG - 1) / DST_SEG;
        DevBuf seg(stream);
        KCHECK(seg.alloc(4 * (n_seg + 1) * 8));
        miss_cap = E / 8 + (1u << 16);
        if (miss_key.alloc(miss_cap * 8 * NW + 16) != KATOME_OK || miss_edge.alloc(miss_cap * 8 + 16) != KATOME_OK) {
            miss_key.release(); miss_edge.release();
            miss_cap = 1u << 16;
            KCHECK(miss_key.alloc(miss_cap * 8 * NW + 16));
            KCHECK(miss_edge.alloc(miss_cap * 8 + 16));
        }
        hipLaunchKernelGGL(dst_seg_kernel<NW>, dim3(grid_for(4 * (n_seg + 1), BLOCK)), dim3(BLOCK), 0, stream, nodes, d_edge_key, E, node_bits, n_seg, seg.as<u64>());
        const size_t lds = (size_t)(DST_SEG * NW + MissCap<NW>::value * (NW + 1) + (first ? DST_SEG : 0)) * 8 + (first ? 2 * (DST_SEG / 32) * 4 : 0);
        const dim3 grid((unsigned)std::min<u64>(n_seg, 256u * 32u));
        KernelScope ks(K_DST_MERGE, stream, E);
        if (first) {
            // (room for the nodes without out-edges, like node_key's)
            // (the merge writes the first touch of every source; the room behind them, for the nodes without out-edges, starts at all-ones)
            KCHECK(node_first->alloc(node_key.bytes / NW));
            KCHECK_HIP(hipMemsetAsync(node_first->as<u64>() + n_src, 0xFF, node_first->bytes - n_src * 8, stream));
            hipLaunchKernelGGL((dst_merge_kernel<NW, true>), grid, dim3(BLOCK), lds, stream, nodes, n_src, d_edge_key, k, n_seg, seg.as<u64>(), edge_dst,
                               miss_key.as<u64>(), miss_edge.as<u64>(), miss_cap, aux.as<u64>(), seq, node_first->as<u64>(), edge_src, seg_edge.as<u64>(), E);
        } else {
            hipLaunchKernelGGL((dst_merge_kernel<NW, false>), grid, dim3(BLOCK), lds, stream, nodes, n_src, d_edge_key, k, n_seg, seg.as<u64>(), edge_dst,
                               miss_key.as<u64>(), miss_edge.as<u64>(), miss_cap, aux.as<u64>(), nullptr, nullptr, nullptr, nullptr, E);
        }
        KCHECK_HIP(hipGetLastError());
    } else {
        u32 B = 1;
        while ((2ull << B) <= n_src / 8 && B < 27) ++B;
        if (B > node_bits) B = node_bits;
        DevBuf index(stream);
        KCHECK(index.alloc(((1ull << B) + 2) * 8));
        hipLaunchKernelGGL(bucket_index_kernel<NW>, dim3(grid_for(n_src + 1, BLOCK)), dim3(BLOCK), 0, stream, nodes, n_src, node_bits, B, index.as<u64>());
        hipLaunchKernelGGL(dst_rank_kernel<NW>, dim3(grid_for(E, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, nodes, n_src, node_bits, B,
                           index.as<u64>(), d_edge_key, E, k, edge_dst, aux.as<u64>());
        KCHECK_HIP(hipGetLastError());
    }
    KCHECK_HIP(hipMemcpyAsync(&n_missing, aux.p, 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    u64 n_extra = 0;
    if (n_missing) {
        DevBuf extra(stream);
        KCHECK(extra.alloc(n_missing * 8 * NW + 16));
        if (n_missing > miss_cap) {
            // (the old look-up, or more targets without out-edges than were given room: a second pass over edge_dst collects
            // them.  Setting them aside inside dst_rank_kernel was tried: 62 % of its waves hold one, and that many atomics
            // on one cursor cost more than this pass)
            miss_key.release(); miss_edge.release();
            KCHECK(miss_key.alloc(n_missing * 8 * NW + 16));
            KCHECK(miss_edge.alloc(n_missing * 8 + 16));
            hipLaunchKernelGGL(missing_gather_kernel<NW>, dim3(grid_for(E, BLOCK * UNIQ_ITEMS, 256u * 16u)), dim3(BLOCK), 0, stream, d_edge_key, E, k,
                               edge_dst, miss_key.as<u64>(), miss_edge.as<u64>(), aux.as<u64>() + 1);
            KCHECK_HIP(hipGetLastError());
        }
        KCHECK_HIP(hipMemcpyAsync(extra.p, miss_key.p, n_missing * 8 * NW, hipMemcpyDeviceToDevice, stream));
        KCHECK(dev_sort(extra.as<u64>(), nullptr, n_missing, NW, node_bits, stream));
        n_extra = n_missing;
        KCHECK(dev_unique(extra.as<u64>(), n_missing, NW, &n_extra, stream));
        hipLaunchKernelGGL(missing_rank_kernel<NW>, dim3(grid_for(n_missing, BLOCK)), dim3(BLOCK), 0, stream, miss_key.as<u64>(),
                           miss_edge.as<u64>(), n_missing, extra.as<u64>(), n_extra, n_src, edge_dst);
        KCHECK_HIP(hipGetLastError());
        if (first) {
            if ((n_src + n_extra + 1) * 8 > node_first->bytes) {
                DevBuf all(stream);
                KCHECK(all.alloc((n_src + n_extra + 1) * 8));
                KCHECK_HIP(hipMemcpyAsync(all.p, node_first->p, n_src * 8, hipMemcpyDeviceToDevice, stream));
                KCHECK_HIP(hipMemsetAsync(all.as<u64>() + n_src, 0xFF, (n_extra + 1) * 8, stream));
                const size_t bytes = all.bytes;
                node_first->adopt(all.take(), bytes);
            }
            hipLaunchKernelGGL(missing_first_kernel, dim3(grid_for(n_missing, BLOCK)), dim3(BLOCK), 0, stream, miss_edge.as<u64>(), n_missing, edge_dst, seq,
                               node_first->as<u64>());
            KCHECK_HIP(hipGetLastError());
        }
        // node_key = sources ++ extra
        if ((n_src + n_extra + 1) * 8 * NW <= node_key.bytes) {
            KCHECK_HIP(hipMemcpyAsync(nodes + n_src * NW, extra.p, n_extra * 8 * NW, hipMemcpyDeviceToDevice, stream));
        } else {
            DevBuf all(stream);
            KCHECK(all.alloc((n_src + n_extra + 1) * 8 * NW));
            KCHECK_HIP(hipMemcpyAsync(all.p, nodes, n_src * 8 * NW, hipMemcpyDeviceToDevice, stream));
            KCHECK_HIP(hipMemcpyAsync(all.as<u64>() + n_src * NW, extra.p, n_extra * 8 * NW, hipMemcpyDeviceToDevice, stream));
            const size_t bytes = all.bytes;
            node_key.adopt(all.take(), bytes);
        }
    }
    *n_nodes = n_src + n_extra;
    return KATOME_OK;
}

int dev_node_ids(const uint64_t* d_edge_key, uint64_t n_edges, uint32_t k, DevBuf& node_key, uint64_t* d_edge_src,
                 uint64_t* d_edge_dst, uint64_t* n_nodes, hipStream_t stream, const uint64_t* d_seq, DevBuf* node_first, uint64_t* n_marked) {
    if (key_words_for_k(k) == 1) return node_ids_t<1>(d_edge_key, n_edges, k, node_key, d_edge_src, d_edge_dst, n_nodes, stream, d_seq, node_first, n_marked);
    return node_ids_t<2>(d_edge_key, n_edges, k, node_key, d_edge_src, d_edge_dst, n_nodes, stream, d_seq, node_first, n_marked);
}

// ---- first-seen order: small permutation helpers ---------------------------------------------------
__global__ __launch_bounds__(BLOCK) void iota_kernel(u32* __restrict__ out, u64 n) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) out[i] = (u32)i;
}
template <class T>
__global__ __launch_bounds__(BLOCK) void gather_kernel(const T* __restrict__ src, const u32* __restrict__ idx, u64 n, T* __restrict__ dst) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) dst[i] = src[idx[i]];
}
template <int NW>
__global__ __launch_bounds__(BLOCK) void gather_keys_kernel(const u64* __restrict__ src, const u32* __restrict__ idx, u64 n, u64* __restrict__ dst) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) store_key<NW>(dst, i, load_key<NW>(src, idx[i]));
}
// dst[i] = map[src[idx[i]]]
__global__ __launch_bounds__(BLOCK) void gather_mapped_kernel(const u64* __restrict__ src, const u32* __restrict__ idx, const u64* __restrict__ map,
                                                               u64 n, u64* __restrict__ dst) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) dst[i] = map[src[idx[i]]];
}
// first-seen order: the edges leave key order for sequence order.  Four separate gathers cost six random reads per edge
// (two of them through the node map); packing each edge into one 32-byte record first (its end points already mapped,
// the source map read nearly in order because sources ascend with the keys) leaves two.
struct PackedEdge { u64 k0, k1; u32 src, dst, weight, pad; };
static_assert(sizeof(PackedEdge) == 32, "packed edge layout");
template <int NW>
__global__ __launch_bounds__(BLOCK) void pack_edges_kernel(const u64* __restrict__ key, const u32* __restrict__ weight, const u64* __restrict__ src,
                                                            const u64* __restrict__ dst, const u64* __restrict__ new_id, u64 n,
                                                            PackedEdge* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        PackedEdge e;
        e.k0 = key[i * NW]; e.k1 = NW == 2 ? key[i * NW + 1] : 0;
        e.src = (u32)new_id[src[i]]; e.dst = (u32)new_id[dst[i]]; e.weight = weight[i]; e.pad = 0;
        out[i] = e;
    }
}
template <int NW>
__global__ __launch_bounds__(BLOCK) void unpack_edges_kernel(const PackedEdge* __restrict__ in, const u32* __restrict__ idx, u64 n,
                                                              u64* __restrict__ key, u32* __restrict__ weight, u64* __restrict__ src,
                                                              u64* __restrict__ dst) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const PackedEdge e = in[idx[i]];
        key[i * NW] = e.k0;
        if (NW == 2) key[i * NW + 1] = e.k1;
        weight[i] = e.weight; src[i] = e.src; dst[i] = e.dst;
    }
}
// in place: edge arrays permuted by idx (new position i <- old position idx[i]) with end points mapped through new_id;
// `scratch` needs n * 32 bytes
int dev_permute_edges(uint64_t* key, uint32_t* weight, uint64_t* src, uint64_t* dst, const uint64_t* new_id, const uint32_t* idx,
                      uint64_t n, uint32_t nw, void* scratch, hipStream_t stream) {
    if (n == 0) return KATOME_OK;
    PackedEdge* aos = (PackedEdge*)scratch;
    const dim3 grid(grid_for(n, BLOCK, 256u * 32u)), blk(BLOCK);
    if (nw == 1) {
        hipLaunchKernelGGL(pack_edges_kernel<1>, grid, blk, 0, stream, key, weight, src, dst, new_id, n, aos);
        hipLaunchKernelGGL(unpack_edges_kernel<1>, grid, blk, 0, stream, aos, idx, n, key, weight, src, dst);
    } else {
        hipLaunchKernelGGL(pack_edges_kernel<2>, grid, blk, 0, stream, key, weight, src, dst, new_id, n, aos);
        hipLaunchKernelGGL(unpack_edges_kernel<2>, grid, blk, 0, stream, aos, idx, n, key, weight, src, dst);
    }
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

// ---- first-seen order without sorting the nodes -------------------------------------------------------------------------
// A node's index is the rank of its first touch (2 * seq as the source of an edge's first insertion, 2 * seq + 1 as its
// target), and every touch belongs to exactly one edge: once the edges are in sequence order, the node indices are a running
// count of "this edge introduces its source / its target" -- a scan over the edges instead of a sort of the nodes.
// pack: the 32-byte record of dev_permute_edges with the OLD end points and, in `pad`, bit 0 = introduces its source,
// bit 1 = introduces its target (node_first from dev_node_first)
template <int NW>
__global__ __launch_bounds__(BLOCK) void pack_edges_intro_kernel(const u64* __restrict__ key, const u32* __restrict__ weight, const u64* __restrict__ src,
                                                                  const u64* __restrict__ dst, const u64* __restrict__ seq,
                                                                  const u64* __restrict__ node_first, u64 n, u64 n_marked, PackedEdge* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        PackedEdge e;
        e.k0 = key[i * NW]; e.k1 = NW == 2 ? key[i * NW + 1] : 0;
        const u64 s = src[i], dm = dst[i], d = dm & ~DST_MARKS, q = seq[i];
        e.src = (u32)s; e.dst = (u32)d; e.weight = weight[i];
        const bool out1 = (i == 0 || src[i - 1] != s) && (i + 1 >= n || src[i + 1] != s);     // the source has this out-edge only
        // (targets below n_marked carry the answer as a mark from the merge; the others -- nodes without out-edges, or no merge --
        // are looked up: node_first[s] is read in order, node_first[d] is not)
        const bool fd = d < n_marked ? (dm & DST_FD) != 0 : node_first[d] == 2 * q + 1;
        e.pad = (node_first[s] == 2 * q ? 1u : 0u) | (fd ? 2u : 0u) | ((dm & DST_IN1) ? 4u : 0u) | (out1 ? 8u : 0u);
        out[i] = e;
    }
}
// unpack in sequence order (new position i <- old position idx[i]); the flags ride in bit 32 of the (old) end points;
// cnt[i] = nodes the edge introduces
template <int NW>
__global__ __launch_bounds__(BLOCK) void unpack_edges_intro_kernel(const PackedEdge* __restrict__ in, const u32* __restrict__ idx, u64 n,
                                                                    u64* __restrict__ key, u32* __restrict__ weight, u64* __restrict__ src,
                                                                    u64* __restrict__ dst, u32* __restrict__ cnt) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const PackedEdge e = in[idx[i]];
        key[i * NW] = e.k0;
        if (NW == 2) key[i * NW + 1] = e.k1;
        weight[i] = e.weight;
        src[i] = (u64)e.src | ((u64)(e.pad & 1u) << 32) | ((u64)((e.pad >> 3) & 1u) << 33);
        dst[i] = (u64)e.dst | ((u64)((e.pad >> 1) & 1u) << 32) | ((u64)((e.pad >> 2) & 1u) << 33);
        cnt[i] = (e.pad & 1u) + ((e.pad >> 1) & 1u);
    }
}
// offs = exclusive scan of cnt: the edge's nodes get indices offs[i] (source, if introduced) and the next one (target).
// The indices reach the OTHER edges of a node through new_id[old index] -- a scattered write and a scattered read per node,
// the two most expensive steps of the renumbering.  Most nodes never need either: in sequence order an edge is usually
// followed by the next window of the same read, so a node is introduced as the target of edge i and used as the source of
// edge i + 1 (remap_ends_kernel reads it off its neighbour); when it has no other in- or out-edge (bits 33: marks from the
// merge and from the runs of sources) nobody else will ask for it and the write is left out as well.
template <int NW>
__global__ __launch_bounds__(BLOCK) void assign_nodes_kernel(const u64* __restrict__ key, const u64* __restrict__ src, const u64* __restrict__ dst,
                                                              const u64* __restrict__ offs, u64 n, u32 k, u64* __restrict__ new_id,
                                                              u64* __restrict__ node_key) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u64 s = src[i], d = dst[i];
        const u32 fs = (u32)(s >> 32) & 1u, fd = (u32)(d >> 32) & 1u;
        if (!(fs | fd)) continue;
        const Key<NW> e = load_key<NW>(key, i);
        const u64 base = offs[i];
        if (fs) { new_id[(u32)s] = base; store_key<NW>(node_key, base, source_node(e)); }
        if (fd) {
            store_key<NW>(node_key, base + fs, target_node(e, k));
            bool alone = false;                         // one in-edge (this one), one out-edge, and that one comes next
            if (((d >> 33) & 1u) && i + 1 < n) { const u64 s1 = src[i + 1]; alone = ((s1 >> 33) & 1u) && (u32)s1 == (u32)d; }
            if (!alone) new_id[(u32)d] = base + fs;
        }
    }
}
__global__ __launch_bounds__(BLOCK) void remap_ends_kernel(const u64* __restrict__ src, const u64* __restrict__ dst, const u64* __restrict__ offs,
                                                            const u64* __restrict__ new_id, u64 n, u64* __restrict__ osrc, u64* __restrict__ odst) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u64 s = src[i], d = dst[i];
        const u32 fs = (u32)(s >> 32) & 1u, fd = (u32)(d >> 32) & 1u;
        const u64 base = (fs | fd) ? offs[i] : 0;
        u64 so;
        if (fs) so = base;
        else {
            const u64 dp = i ? dst[i - 1] : 0;
            if (i && ((dp >> 32) & 1u) && (u32)dp == (u32)s) so = offs[i - 1] + ((src[i - 1] >> 32) & 1u);      // introduced by the edge before
            else so = new_id[(u32)s];
        }
        osrc[i] = so;
        odst[i] = fd ? base + fs : new_id[(u32)d];
    }
}
__global__ __launch_bounds__(BLOCK) void clear_marks_kernel(u64* __restrict__ v, u64 n) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) v[i] &= ~DST_MARKS;
}
int dev_clear_dst_marks(uint64_t* dst, uint64_t n, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(clear_marks_kernel, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, dst, n);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
int dev_pack_edges_intro(const uint64_t* key, const uint32_t* weight, const uint64_t* src, const uint64_t* dst, const uint64_t* seq,
                         const uint64_t* node_first, uint64_t n, uint32_t nw, void* aos, hipStream_t stream, uint64_t n_marked) {
    if (n == 0) return KATOME_OK;
    const dim3 grid(grid_for(n, BLOCK, 256u * 32u)), blk(BLOCK);
    if (nw == 1) hipLaunchKernelGGL(pack_edges_intro_kernel<1>, grid, blk, 0, stream, key, weight, src, dst, seq, node_first, n, n_marked, (PackedEdge*)aos);
    else         hipLaunchKernelGGL(pack_edges_intro_kernel<2>, grid, blk, 0, stream, key, weight, src, dst, seq, node_first, n, n_marked, (PackedEdge*)aos);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
int dev_unpack_edges_intro(const void* aos, const uint32_t* idx, uint64_t n, uint32_t nw, uint64_t* key, uint32_t* weight, uint64_t* src,
                           uint64_t* dst, uint32_t* cnt, hipStream_t stream) {
    if (n == 0) return KATOME_OK;
    const dim3 grid(grid_for(n, BLOCK, 256u * 32u)), blk(BLOCK);
    if (nw == 1) hipLaunchKernelGGL(unpack_edges_intro_kernel<1>, grid, blk, 0, stream, (const PackedEdge*)aos, idx, n, key, weight, src, dst, cnt);
    else         hipLaunchKernelGGL(unpack_edges_intro_kernel<2>, grid, blk, 0, stream, (const PackedEdge*)aos, idx, n, key, weight, src, dst, cnt);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
int dev_assign_nodes(const uint64_t* key, const uint64_t* src, const uint64_t* dst, const uint64_t* offs, uint64_t n, uint32_t nw, uint32_t k,
                     uint64_t* new_id, uint64_t* node_key, uint64_t* out_src, uint64_t* out_dst, hipStream_t stream) {
    if (n == 0) return KATOME_OK;
    const dim3 grid(grid_for(n, BLOCK, 256u * 32u)), blk(BLOCK);
    if (nw == 1) hipLaunchKernelGGL(assign_nodes_kernel<1>, grid, blk, 0, stream, key, src, dst, offs, n, k, new_id, node_key);
    else         hipLaunchKernelGGL(assign_nodes_kernel<2>, grid, blk, 0, stream, key, src, dst, offs, n, k, new_id, node_key);
    hipLaunchKernelGGL(remap_ends_kernel, grid, blk, 0, stream, src, dst, offs, new_id, n, out_src, out_dst);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

// inverse of a permutation: inv[perm[i]] = i
__global__ __launch_bounds__(BLOCK) void invert_kernel(const u32* __restrict__ perm, u64 n, u64* __restrict__ inv) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) inv[perm[i]] = i;
}
// a node is created by the first edge insertion that touches it: as the source of the first window of a strand
// (2*seq) or as a target (2*seq + 1) -- add_single_edge_fastaq, pt_graph.rs:180-185
// Source role: the edges are in key order, so a node's out-edges are one run of equal src (and src ascending): the run's head
// takes the minimum over its run and stores it plainly -- one writer per node, no atomic.  Target role: atomicMin, afterwards.
__global__ __launch_bounds__(BLOCK) void node_first_src_kernel(const u64* __restrict__ src, const u64* __restrict__ seq, u64 n, u64* __restrict__ node_first) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u64 s = src[i];
        if (i > 0 && src[i - 1] == s) continue;
        u64 m = seq[i];
        for (u64 j = i + 1; j < n && src[j] == s; ++j) m = seq[j] < m ? seq[j] : m;      // (<= 4 out-edges per node; BFCounter lists may repeat)
        node_first[s] = 2 * m;
    }
}
__global__ __launch_bounds__(BLOCK) void node_first_dst_kernel(const u64* __restrict__ dst, const u64* __restrict__ seq, u64 n, u64* node_first) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK)
        atomicMin((unsigned long long*)&node_first[dst[i]], (unsigned long long)(2 * seq[i] + 1));
}

int dev_iota(uint32_t* d, uint64_t n, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, stream, d, n);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
__global__ __launch_bounds__(BLOCK) void fill_u32_kernel(u32* __restrict__ d, u64 n, u32 v) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) d[i] = v;
}
int dev_fill_u32(uint32_t* d, uint64_t n, uint32_t v, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(fill_u32_kernel, dim3(grid_for(n, BLOCK, 256u * 8u)), dim3(BLOCK), 0, stream, d, n, v);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
// {sequence number, weight} pairs (table.hip emit, first-seen order) brought into the order of idx[] and split
__global__ __launch_bounds__(BLOCK) void gather_seq_weight_kernel(const ulonglong2* __restrict__ pairs, const u32* __restrict__ idx, u64 n,
                                                                  u64* __restrict__ seq, u32* __restrict__ weight) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const ulonglong2 p = pairs[idx[i]];
        seq[i] = p.x; weight[i] = (u32)p.y;
    }
}
int dev_gather_seq_weight(const uint64_t* pairs, const uint32_t* idx, uint64_t n, uint64_t* seq, uint32_t* weight, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(gather_seq_weight_kernel, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream,
                              reinterpret_cast<const ulonglong2*>(pairs), idx, n, seq, weight);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
int dev_gather_u32(const uint32_t* src, const uint32_t* idx, uint64_t n, uint32_t* dst, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(gather_kernel<u32>, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, idx, n, dst);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
int dev_gather_u64(const uint64_t* src, const uint32_t* idx, uint64_t n, uint64_t* dst, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(gather_kernel<u64>, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, idx, n, dst);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
int dev_gather_keys(const uint64_t* src, const uint32_t* idx, uint64_t n, uint32_t nw, uint64_t* dst, hipStream_t stream) {
    if (n) {
        if (nw == 1) hipLaunchKernelGGL(gather_keys_kernel<1>, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, idx, n, dst);
        else         hipLaunchKernelGGL(gather_keys_kernel<2>, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, idx, n, dst);
    }
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
int dev_gather_mapped(const uint64_t* src, const uint32_t* idx, const uint64_t* map, uint64_t n, uint64_t* dst, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(gather_mapped_kernel, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, idx, map, n, dst);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
int dev_invert(const uint32_t* perm, uint64_t n, uint64_t* inv, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(invert_kernel, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, perm, n, inv);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}
int dev_node_first(const uint64_t* src, const uint64_t* dst, const uint64_t* seq, uint64_t n, uint64_t* node_first, hipStream_t stream) {
    // (edges in key order: src ascending in runs)
    if (n) {
        hipLaunchKernelGGL(node_first_src_kernel, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, seq, n, node_first);
        hipLaunchKernelGGL(node_first_dst_kernel, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, dst, seq, n, node_first);
    }
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

// ---- edge -> endpoints, labels ----------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(BLOCK) void endpoints_kernel(const u64* __restrict__ ek, u64 n, u32 k, u64* __restrict__ src, u64* __restrict__ dst) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Key<NW> key = load_key<NW>(ek, i);
        if (src) store_key<NW>(src, i, source_node(key));
        if (dst) store_key<NW>(dst, i, target_node(key, k));
    }
}
int dev_endpoints(const uint64_t* d_edge_key, uint64_t n, uint32_t k, uint64_t* d_src, uint64_t* d_dst, hipStream_t stream) {
    if (n == 0) return KATOME_OK;
    dim3 grid(grid_for(n, BLOCK)), block(BLOCK);
    if (key_words_for_k(k) == 1) hipLaunchKernelGGL(endpoints_kernel<1>, grid, block, 0, stream, d_edge_key, n, k, d_src, d_dst);
    else                         hipLaunchKernelGGL(endpoints_kernel<2>, grid, block, 0, stream, d_edge_key, n, k, d_src, d_dst);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

// BFCounter input (create_bfc builder.rs:79-115 -> add_read_bfc pt_graph.rs:317-330 -> add_single_edge_bfc 201-213): every
// kept line is ONE edge -- and with reverse_complement a second one for its reverse complement, right after it -- added
// with `add_edge` unconditionally: a k-mer listed twice, or a k-mer that is its own reverse complement, stays as parallel
// edges.  So there is no table here: line i becomes edge i (2i and 2i+1 with both strands), whose sequence number is its
// petgraph index.
template <int NW>
__global__ __launch_bounds__(BLOCK) void bfc_edges_kernel(const u64* __restrict__ fwd, const u32* __restrict__ w, u64 n, u32 k, bool rc,
                                                           u64* __restrict__ ek, u32* __restrict__ ew, u64* __restrict__ seq) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Key<NW> key = load_key<NW>(fwd, i);
        const u32 wi = w[i];
        if (rc) {
            store_key<NW>(ek, 2 * i, key); store_key<NW>(ek, 2 * i + 1, revcomp(key, k));
            ew[2 * i] = wi; ew[2 * i + 1] = wi;
            if (seq) { seq[2 * i] = 2 * i; seq[2 * i + 1] = 2 * i + 1; }
        } else {
            store_key<NW>(ek, i, key);
            ew[i] = wi;
            if (seq) seq[i] = i;
        }
    }
}
int dev_bfc_edges(const uint64_t* d_fwd, const uint32_t* d_w, uint64_t n, uint32_t k, bool rc, uint64_t* d_edge_key,
                  uint32_t* d_edge_weight, uint64_t* d_edge_seq, hipStream_t stream) {
    if (n == 0) return KATOME_OK;
    dim3 grid(grid_for(n, BLOCK)), block(BLOCK);
    if (key_words_for_k(k) == 1) hipLaunchKernelGGL(bfc_edges_kernel<1>, grid, block, 0, stream, d_fwd, d_w, n, k, rc, d_edge_key, d_edge_weight, d_edge_seq);
    else                         hipLaunchKernelGGL(bfc_edges_kernel<2>, grid, block, 0, stream, d_fwd, d_w, n, k, rc, d_edge_key, d_edge_weight, d_edge_seq);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

// compress_edge format: [pad][ceil(k/4) bytes].  A workgroup builds the labels of LABEL_ITEMS * 256 edges in LDS and streams
// them out as whole dwords (the byte stride is odd for most k); four keys per thread are loaded before the first is used (with
// 256 edges per trip a workgroup moved 4 KB and a CU had too few bytes in flight: 8.1 ms for C3's 27.6 GB).
constexpr u32 LABEL_ITEMS = 4;
template <int NW>
__global__ __launch_bounds__(BLOCK) void labels_kernel(const u64* __restrict__ ek, u64 n, u32 k, uint8_t* __restrict__ out) {
    extern __shared__ u32 lbuf[];
    uint8_t* lb = reinterpret_cast<uint8_t*>(lbuf);
    const u32 stride = label_stride_for_k(k), nb = stride - 1, pad = label_pad_for_k(k);
    constexpr u32 TILE = BLOCK * LABEL_ITEMS;
    const u64 ntiles = (n + TILE - 1) / TILE;
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const u64 e0 = t * TILE;
        const u32 cnt = (u32)((n - e0) < (u64)TILE ? (n - e0) : (u64)TILE);
        Key<NW> key[LABEL_ITEMS];
#pragma unroll
        for (u32 r = 0; r < LABEL_ITEMS; ++r) { const u32 j = r * BLOCK + threadIdx.x; if (j < cnt) key[r] = load_key<NW>(ek, e0 + j); }
#pragma unroll
        for (u32 r = 0; r < LABEL_ITEMS; ++r) {
            const u32 j = r * BLOCK + threadIdx.x;
            if (j < cnt) {
                uint8_t* p = lb + j * stride;
                p[0] = (uint8_t)pad;
                for (u32 i = 0; i < nb; ++i) p[1 + i] = label_byte(key[r], k, i);
            }
        }
        __syncthreads();
        const u64 byte0 = e0 * stride;                 // TILE*stride is a multiple of 4 -> dword aligned
        const u32 nbytes = cnt * stride;
        u32* o32 = reinterpret_cast<u32*>(out + byte0);
        for (u32 i = threadIdx.x; i < nbytes / 4; i += BLOCK) o32[i] = lbuf[i];
        for (u32 i = (nbytes / 4) * 4 + threadIdx.x; i < nbytes; i += BLOCK) out[byte0 + i] = lb[i];
        __syncthreads();
    }
}
int dev_labels(const uint64_t* d_edge_key, uint64_t n, uint32_t k, uint8_t* d_label, hipStream_t stream) {
    if (n == 0) return KATOME_OK;
    const size_t lds = (size_t)BLOCK * LABEL_ITEMS * label_stride_for_k(k) + 16;
    dim3 grid(grid_for(n, BLOCK * LABEL_ITEMS, 256u * 16u)), block(BLOCK);
    if ((uintptr_t)d_label % 4) { set_error("label buffer must be 4-byte aligned"); return KATOME_E_ARG; }
    if (key_words_for_k(k) == 1) hipLaunchKernelGGL(labels_kernel<1>, grid, block, lds, stream, d_edge_key, n, k, d_label);
    else                         hipLaunchKernelGGL(labels_kernel<2>, grid, block, lds, stream, d_edge_key, n, k, d_label);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

}  // namespace katome
