// supermer.hip -- the sharded build's single exchange: a read travels as its SUPERMERS.
//
// The reference's build is one sequential loop over windows (pt_graph.rs:277-315); what the sharded build has to get right is that
// every occurrence of a k-mer is counted on ONE rank and that a node lives with its out-edges (the HmGIR shape, hm_gir.rs:91-153).
// Both follow when a k-mer's owner is a function of its canonical middle (k-2)-mer, the core.  Here that function is the core's
// MINIMIZER (kmer_bits.h): consecutive windows of a read mostly share it, so a read is cut into a dozen runs of windows with one
// minimizer -- hence one owner -- and each run is shipped as ONE fixed-size record of nwin + k - 1 bases (16 bytes for k <= 31)
// instead of nwin k-mer records: ~0.2 KB per 150-bp read against 0.96 KB, once, before anything is counted.  What arrives on a
// rank is every occurrence of the k-mers it owns: it counts the distinct supermers by sorting (table.hip records_to_edges_sorted, as
// one GPU counts its tiles), cuts each distinct supermer into its k-mers with the supermer's count, counts those, and has its share
// of the edges -- the one-GPU pipeline on what it received, no second exchange of records, no partition pass per level.
//
// Runs are cut where the minimizer's POSITION changes (the leftmost lowest m-mer of the window's core): a position stays inside the
// core for at most core - m + 1 windows, which bounds the record without any reference to the read's own offsets -- two reads that
// cover the same stretch of genome cut it at the same places (except at their ends) and their records are equal.
#include "common.h"

namespace katome {
namespace {

typedef uint16_t u16;
constexpr u32 SM_WAVE = 64;

// One wave per read.  LDS per wave: the read's bases as byte-swapped dwords, the hash of the canonical m-mer at every position, the
// position of every window's minimizer.  Record j of read r goes to slot r * slots + j; the slots a read does not fill are written
// invalid (the partition pass drops them); a read with more than `slots` runs puts the rest behind a cursor in `spill`.
template <bool RC>
__global__ __launch_bounds__(SM_WAVE) void supermer_extract_kernel(const uint8_t* __restrict__ packed, u64 n_reads, u32 read_len, u32 stride,
                                                                    const uint8_t* __restrict__ skip, u32 k, u32 m, u32 n_owners, u32 slots,
                                                                    u64* __restrict__ out, u64* __restrict__ spill, u64 spill_cap,
                                                                    unsigned long long* spill_cursor) {
    extern __shared__ u32 sm_lds[];
    const u32 n_words = (read_len + 15) / 16 + 6;                 // (windows near the end read past it: zero words)
    const u32 P = read_len - m + 1, W = read_len - k + 1, w = k - 2 - m + 1;
    u32* words = sm_lds;                                          // [n_words]
    u32* hsh = sm_lds + n_words;                                  // [P]
    u16* mpos = reinterpret_cast<u16*>(hsh + P);                  // [W]
    const u32 lane = threadIdx.x;
    const u64 lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (u64 r = blockIdx.x; r < n_reads; r += gridDim.x) {
        u64* mine = out + r * slots * 2;
        if (skip && skip[r]) {                                    // a read with a base that is not ACGT: nothing (builder.rs:155-157)
            for (u32 j = lane; j < slots; j += SM_WAVE) { mine[2 * j] = INVALID_WORD; mine[2 * j + 1] = INVALID_WORD; }
            continue;
        }
        __syncthreads();                                          // (one wave: orders this read's LDS writes behind the last read's reads)
        const uint8_t* src = packed + r * stride;
        for (u32 j = lane; j < n_words; j += SM_WAVE) {
            u32 v = 0;
#pragma unroll
            for (u32 b = 0; b < 4; ++b) { const u32 at = 4 * j + b; v = (v << 8) | (at < stride ? (u32)src[at] : 0u); }
            words[j] = v;                                         // 16 bases, the first one on top
        }
        __syncthreads();
        for (u32 p = lane; p < P; p += SM_WAVE) {
            const u32 di = p >> 4, sh = (p & 15) * 2;
            u32 d[3] = {words[di], words[di + 1], words[di + 2]};
            const Key<1> f = extract_window(d, sh, m, (Key<1>*)nullptr);
            hsh[p] = mmer_hash(f.w[0], revcomp(f, m).w[0]);
        }
        __syncthreads();
        // window i: k-mer at bases [i, i + k); its core at [i + 1, i + k - 1); the core's m-mers start at i + 1 .. i + w
        for (u32 i = lane; i < W; i += SM_WAVE) {
            u32 best = hsh[i + 1], at = i + 1;
            for (u32 j = 2; j <= w; ++j) { const u32 h = hsh[i + j]; if (h < best) { best = h; at = i + j; } }
            mpos[i] = (u16)at;
        }
        __syncthreads();
        u32 made = 0;                                             // runs of this read so far (wave-uniform)
        for (u32 i0 = 0; i0 < W; i0 += SM_WAVE) {
            const u32 i = i0 + lane;
            const bool start = i < W && (i == 0 || mpos[i] != mpos[i - 1]);
            const u64 ballot = __ballot(start);
            if (start) {
                u32 nwin = 1;                                     // (at most w: the position leaves the core after that many windows)
                while (i + nwin < W && mpos[i + nwin] == mpos[i]) ++nwin;
                const u32 nb = nwin + k - 1, di = i >> 4, sh = (i & 15) * 2;
                u32 d[5] = {words[di], words[di + 1], words[di + 2], words[di + 3], words[di + 4]};
                Key<2> s = extract_window(d, sh, nb, (Key<2>*)nullptr);
                if (RC) s = canonical(s, nb);
                const u32 owner = (u32)minimizer_owner_of(hsh[mpos[i]], n_owners);
                s.w[0] |= ((u64)owner << SUPERMER_OWNER_SHIFT) | ((u64)nwin << SUPERMER_LEN_SHIFT);
                const u32 j = made + (u32)__popcll(ballot & lt_mask);
                if (j < slots) { mine[2 * j] = s.w[0]; mine[2 * j + 1] = s.w[1]; }
                else {
                    const unsigned long long q = atomicAdd(spill_cursor, 1ull);
                    if (q < spill_cap) { spill[2 * q] = s.w[0]; spill[2 * q + 1] = s.w[1]; }
                }
            }
            made += (u32)__popcll(ballot);
        }
        for (u32 j = made + lane; j < slots; j += SM_WAVE) { mine[2 * j] = INVALID_WORD; mine[2 * j + 1] = INVALID_WORD; }
    }
}

__global__ __launch_bounds__(BLOCK) void supermer_len_kernel(const u64* __restrict__ list, u64 n, u32* __restrict__ len) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Key<2> s; s.w[0] = list[2 * i]; s.w[1] = list[2 * i + 1];
        len[i] = supermer_windows(s);
    }
}
// every distinct supermer -> its k-mers (canonical when both strands are counted), each with the supermer's count
template <bool RC>
__global__ __launch_bounds__(BLOCK) void supermer_expand_kernel(const u64* __restrict__ list, const u32* __restrict__ counts, const u64* __restrict__ offs,
                                                                 u64 n, u32 k, u64* __restrict__ out_keys, u32* __restrict__ out_w) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Key<2> s; s.w[0] = list[2 * i]; s.w[1] = list[2 * i + 1];
        const u32 nwin = supermer_windows(s), c = counts[i];
        const Key<2> bases = supermer_bases(s);
        u64 at = offs[i];
        for (u32 j = 0; j < nwin; ++j, ++at) {
            Key<1> x = sub_window<2, 1>(bases, k, nwin, 1, j);
            if (RC) x = canonical(x, k);
            out_keys[at] = x.w[0];
            out_w[at] = c;
        }
    }
}

}  // namespace

// slots per read in the extraction's output: a 150-bp read at k = 31 makes ~13 runs; beyond `slots` a read's runs go to the spill list
uint32_t supermer_slots(uint32_t k, uint32_t read_len, uint32_t m) {
    const uint32_t W = read_len - k + 1, w = k - 2 - m + 1;
    const uint32_t expect = 2 * W / (w + 1) + 2;                  // (a minimizer changes about every (w + 1) / 2 windows)
    return std::min<uint32_t>(W, expect + expect / 2 + 2);
}
bool supermer_route_takes(uint32_t k, uint32_t read_len, uint32_t m) {
    return k >= m + 4 && k <= 31 && read_len >= k && read_len <= 2000 && (k - 2 - m + 1) <= SUPERMER_MAX_WINDOWS;
}

// a batch of reads -> their supermer records: out [n_reads * slots][2] (invalid where a read made fewer), the rest behind *spill_cursor
int dev_supermers_extract(const uint8_t* d_packed, uint64_t n_reads, uint32_t read_len, const uint8_t* d_skip, uint32_t k, uint32_t m, bool rc,
                          uint32_t n_owners, uint32_t slots, uint64_t* d_out, uint64_t* d_spill, uint64_t spill_cap, uint64_t* d_spill_cursor,
                          hipStream_t stream) {
    if (!supermer_route_takes(k, read_len, m) || n_owners == 0 || n_owners > 16) { set_error("supermers: k = %u, reads of %u bases, %u owners", k, read_len, n_owners); return KATOME_E_ARG; }
    if (n_reads == 0) return KATOME_OK;
    const uint32_t stride = (read_len + 3) / 4, n_words = (read_len + 15) / 16 + 6, P = read_len - m + 1, W = read_len - k + 1;
    const size_t lds = (size_t)(n_words + P) * 4 + (size_t)W * 2 + 16;
    const dim3 grid(grid_for(n_reads, 1, 256u * 64u)), block(SM_WAVE);
    unsigned long long* cur = reinterpret_cast<unsigned long long*>(d_spill_cursor);
    if (rc) hipLaunchKernelGGL(supermer_extract_kernel<true>, grid, block, lds, stream, d_packed, n_reads, read_len, stride, d_skip, k, m, n_owners, slots, d_out,
                               d_spill, spill_cap, cur);
    else    hipLaunchKernelGGL(supermer_extract_kernel<false>, grid, block, lds, stream, d_packed, n_reads, read_len, stride, d_skip, k, m, n_owners, slots, d_out,
                               d_spill, spill_cap, cur);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

// a list of distinct supermers with their counts -> the (k-mer, count) records of their windows
int dev_supermers_expand(const uint64_t* d_list, const uint32_t* d_counts, uint64_t n, uint32_t k, bool rc, DevBuf& keys, DevBuf& weights, uint64_t* n_records,
                         hipStream_t stream) {
    *n_records = 0;
    DevBuf len(stream), offs(stream);
    KCHECK(len.alloc((n + 1) * 4)); KCHECK(offs.alloc((n + 2) * 8));
    if (n) {
        hipLaunchKernelGGL(supermer_len_kernel, dim3(grid_for(n, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, d_list, n, len.as<u32>());
        KCHECK_HIP(hipGetLastError());
        KCHECK(dev_scan_counts(len.as<u32>(), n, offs.as<u64>(), stream));
        KCHECK_HIP(hipMemcpyAsync(n_records, offs.as<u64>() + n, 8, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
    }
    KCHECK(keys.alloc((*n_records + 1) * 8, stream)); KCHECK(weights.alloc((*n_records + 1) * 4, stream));
    if (n) {
        KernelScope ks(K_RECORDS, stream, n);
        const dim3 grid(grid_for(n, BLOCK, 256u * 32u)), block(BLOCK);
        if (rc) hipLaunchKernelGGL(supermer_expand_kernel<true>, grid, block, 0, stream, d_list, d_counts, offs.as<u64>(), n, k, keys.as<u64>(), weights.as<u32>());
        else    hipLaunchKernelGGL(supermer_expand_kernel<false>, grid, block, 0, stream, d_list, d_counts, offs.as<u64>(), n, k, keys.as<u64>(), weights.as<u32>());
        KCHECK_HIP(hipGetLastError());
    }
    return KATOME_OK;
}

}  // namespace katome
