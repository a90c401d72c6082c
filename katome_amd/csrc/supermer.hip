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

// One THREAD per read, one wave per workgroup: every lane walks its own read from left to right with the whole state in registers,
// so all 64 lanes do the same useful thing at every step and nothing waits on a neighbour.  (Three earlier forms that spread one
// read's positions and windows over the lanes were all bound by the vector unit -- ~460 wave instructions per read, 20 ms per
// 25 M reads: index arithmetic and half-empty waves; this one issues ~130.)  The tile's packed bytes are staged in LDS as
// byte-swapped dwords (coalesced 16-byte loads); a lane then
//   A. rolls the m-mer and its reverse complement along the read (two shifts and the entering base), hashes the canonical one, and
//      keeps per block of w positions the running minimum from the block's left end and to its right end (van Herk / Gil-Werman):
//      the minimizer of the window whose w positions end at position e is min(suffix of the block before, prefix of this one) --
//      w is a template parameter, so the three w-long arrays are registers.  A run starts where the minimizer's hash differs from
//      the window's before and, so that a record's length is bounded whatever the read holds (poly-A: every hash equal), every w
//      windows inside a longer run; the run's first window and its hash go into the lane's list in LDS;
//   B. run by run (all lanes their j-th run together) cuts nwin + k - 1 bases out of the staged read, takes the canonical form, adds
//      owner and length and writes the record into slot r * slots + j; slots a read does not fill are written invalid (the
//      partition pass drops them); a read with more than `slots` runs puts the rest behind a cursor in `spill`.
// Reads of at most k + 127 bases (128 windows).
constexpr u32 SM_READS = 64, SM_MAXR = 40;             // reads per tile (one per lane); runs a lane can list (beyond: one window per record, spilled)
template <bool RC, int WC>
__global__ __launch_bounds__(SM_READS) void supermer_extract_kernel(const uint8_t* __restrict__ packed, u64 n_reads, u32 read_len, u32 stride,
                                                                     const uint8_t* __restrict__ skip, u32 k, u32 m, u32 n_owners, u32 slots,
                                                                     u64* __restrict__ out, u64* __restrict__ spill, u64 spill_cap,
                                                                     unsigned long long* spill_cursor) {
    extern __shared__ u32 sm_lds[];
    const u32 P = read_len - m + 1, W = read_len - k + 1;
    const u32 NB = (P - 1 + WC - 1) / WC;                         // blocks of w positions, the first at position 1 (position 0 is in no core)
    const u32 tile_dwords = ((SM_READS * stride + 15) / 16) * 4 + 8;
    u32* words = sm_lds;                                          // [tile_dwords]: the tile's bases, 16 per dword, the first one on top
    u64* runs = reinterpret_cast<u64*>(sm_lds + tile_dwords + (tile_dwords & 1));      // [SM_MAXR][SM_READS]: hash << 8 | first window
    const u32 lane = threadIdx.x;
    const u64 total_bytes = n_reads * stride, n_tiles = (n_reads + SM_READS - 1) / SM_READS;
    const u32 mmask = m >= 16 ? 0xFFFFFFFFu : (1u << (2 * m)) - 1, rshift = 2 * (m - 1);
    for (u64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const u64 r0 = tile * SM_READS;
        const u32 nr = (u32)((n_reads - r0) < (u64)SM_READS ? (n_reads - r0) : (u64)SM_READS);
        const u64 byte0 = r0 * stride;
        const u32 nchunks = (nr * stride + 15) / 16;
        __syncthreads();                                          // (one wave: orders the LDS traffic of consecutive tiles)
        for (u32 c = lane; c < nchunks; c += SM_READS) {
            const u64 off = byte0 + (u64)c * 16;
            u32 t[4] = {0, 0, 0, 0};
            const uintptr_t addr = reinterpret_cast<uintptr_t>(packed) + off;
            if (off + 16 <= total_bytes && (addr & 15) == 0) {
                const uint4 v = *reinterpret_cast<const uint4*>(packed + off);
                t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
            } else if (off + 16 <= total_bytes && (addr & 3) == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const u32*>(packed + off + 4 * q);
            } else {                                              // the end of the buffer, or a buffer at an odd address: byte loads
                for (u32 b = 0; b < 16 && off + b < total_bytes; ++b) t[b >> 2] |= (u32)packed[off + b] << (8 * (b & 3));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) words[c * 4 + q] = __builtin_bswap32(t[q]);
        }
        if (lane < 8) words[nchunks * 4 + lane] = 0;              // (windows near the tile's end read past it)
        __syncthreads();
        const bool live = lane < nr && !(skip && skip[r0 + lane]);        // (a read with a base that is not ACGT: nothing, builder.rs:155-157)
        const u32 rbit = lane * stride * 8;
        auto base_at = [&](u32 q) -> u32 { const u32 bit = rbit + 2 * q; return (words[bit >> 5] >> (30 - (bit & 31))) & 3u; };
        // ---- A: the runs of this lane's read -------------------------------------------------------------------------------------------
        u32 count = 0, single_from = W;                           // runs listed; first window of the one-window records past the list's end
        if (live) {
            u32 f = 0, rv = 0;
            for (u32 q = 0; q + 1 < m; ++q) { const u32 b = base_at(q); f = (f << 2) | b; rv = (rv >> 2) | ((3u - b) << rshift); }
            { const u32 b = base_at(m - 1); f = ((f << 2) | b) & mmask; rv = (rv >> 2) | ((3u - b) << rshift); }          // position 0: in no core
            u32 prev_suf[WC];
#pragma unroll
            for (int t = 0; t < WC; ++t) prev_suf[t] = 0xFFFFFFFFu;
            u32 last = 0, runlen = 0;
            bool any = false;
            for (u32 b = 0; b < NB; ++b) {
                const u32 p0 = 1 + b * WC;
                u32 h[WC], pre[WC];
                u32 run = 0xFFFFFFFFu;
#pragma unroll
                for (int t = 0; t < WC; ++t) {
                    h[t] = 0xFFFFFFFFu;
                    if (p0 + t < P) {
                        const u32 bs = base_at(p0 + t + m - 1);
                        f = ((f << 2) | bs) & mmask; rv = (rv >> 2) | ((3u - bs) << rshift);
                        h[t] = mmer_hash(f, rv);
                    }
                    run = h[t] < run ? h[t] : run;
                    pre[t] = run;
                }
                // the windows whose last position lies in this block: window i = (b - 1) * w + t + 1 ends at position p0 + t
#pragma unroll
                for (int t = 0; t < WC; ++t) {
                    const int i = (int)(b * WC) - WC + t + 1;
                    if (i < 0 || (u32)i >= W) continue;
                    const u32 mvv = t + 1 < WC ? (prev_suf[t + 1 < WC ? t + 1 : 0] < pre[t] ? prev_suf[t + 1 < WC ? t + 1 : 0] : pre[t]) : pre[t];
                    if (!any || mvv != last || runlen == WC) {
                        if (count < SM_MAXR - 1) { runs[count * SM_READS + lane] = ((u64)mvv << 8) | (u32)i; ++count; }
                        else if (single_from == W) single_from = (u32)i;
                        runlen = 0; any = true;
                    }
                    last = mvv; ++runlen;
                }
                run = 0xFFFFFFFFu;
#pragma unroll
                for (int t = WC - 1; t >= 0; --t) { run = h[t] < run ? h[t] : run; prev_suf[t] = run; }
            }
        }
        // ---- B: the records, every lane its j-th run ------------------------------------------------------------------------------------
        u64* mine = out + (r0 + lane) * slots * 2;
        u32 maxc = count;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const u32 v = __shfl_xor(maxc, o, 64); maxc = v > maxc ? v : maxc; }
        auto emit = [&](u32 i, u32 nwin, u32 hash, u32 j) {
            const u32 nb = nwin + k - 1, bit = rbit + 2 * i, di = bit >> 5, sh = bit & 31;
            u32 d[5] = {words[di], words[di + 1], words[di + 2], words[di + 3], words[di + 4]};
            Key<2> s = extract_window(d, sh, nb, (Key<2>*)nullptr);
            if (RC) s = canonical(s, nb);
            s.w[0] |= ((u64)minimizer_owner_of(hash, n_owners) << SUPERMER_OWNER_SHIFT) | ((u64)nwin << SUPERMER_LEN_SHIFT);
            if (j < slots) *reinterpret_cast<ulonglong2*>(mine + 2 * j) = make_ulonglong2(s.w[0], s.w[1]);
            else {
                const unsigned long long q = atomicAdd(spill_cursor, 1ull);
                if (q < spill_cap) { spill[2 * q] = s.w[0]; spill[2 * q + 1] = s.w[1]; }
            }
        };
        for (u32 j = 0; j < maxc; ++j) {
            if (j >= count) continue;
            const u64 e = runs[j * SM_READS + lane];
            const u32 i = (u32)e & 255u;
            const u32 next = j + 1 < count ? (u32)runs[(j + 1) * SM_READS + lane] & 255u : single_from;
            emit(i, next - i, (u32)(e >> 8), j);
        }
        if (single_from < W) {                                    // (a read of more runs than the list holds: the rest window by window -- rare, slow, exact)
            for (u32 i = single_from; i < W; ++i) {
                const Key<2> kk = [&] { const u32 bit = rbit + 2 * i, di = bit >> 5, sh = bit & 31;
                                        u32 d[5] = {words[di], words[di + 1], words[di + 2], words[di + 3], words[di + 4]};
                                        return extract_window(d, sh, k, (Key<2>*)nullptr); }();
                Key<1> x; x.w[0] = kk.w[1];
                emit(i, 1, core_minimizer(x, 2, k - 2, m), slots);            // (slots: straight to the spill list)
            }
        }
        if (lane < nr) for (u32 j = count; j < slots; ++j) *reinterpret_cast<ulonglong2*>(mine + 2 * j) = make_ulonglong2(INVALID_WORD, INVALID_WORD);
    }
}

__global__ __launch_bounds__(BLOCK) void supermer_len_kernel(const u64* __restrict__ list, u64 n, u32* __restrict__ len) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Key<2> s; s.w[0] = list[2 * i]; s.w[1] = list[2 * i + 1];
        len[i] = supermer_windows(s);
    }
}
// every distinct supermer -> its k-mers (canonical when both strands are counted), each with the supermer's count.  A workgroup takes
// SX supermers; their records are consecutive in the output (offs), so thread t makes record base + t -- finding its supermer by a
// binary search over the chunk's offsets in LDS -- and a wave's stores are 64 consecutive keys
constexpr u32 SX = 256;
template <bool RC>
__global__ __launch_bounds__(BLOCK) void supermer_expand_kernel(const u64* __restrict__ list, const u32* __restrict__ counts, const u64* __restrict__ offs,
                                                                 u64 n, u32 k, u64* __restrict__ out_keys, u32* __restrict__ out_w) {
    __shared__ u64 sk[SX][2];
    __shared__ u32 sc[SX], so[SX + 1];
    const u32 tid = threadIdx.x;
    const u64 n_chunks = (n + SX - 1) / SX;
    for (u64 c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const u64 s0 = c * SX;
        const u32 ns = (u32)((n - s0) < (u64)SX ? (n - s0) : (u64)SX);
        const u64 base = offs[s0];
        __syncthreads();
        for (u32 t = tid; t < ns; t += BLOCK) {
            sk[t][0] = list[2 * (s0 + t)]; sk[t][1] = list[2 * (s0 + t) + 1];
            sc[t] = counts[s0 + t];
            so[t] = (u32)(offs[s0 + t] - base);
        }
        if (tid == 0) so[ns] = (u32)(offs[s0 + ns] - base);
        __syncthreads();
        const u32 total = so[ns];
        for (u32 t = tid; t < total; t += BLOCK) {
            u32 lo = 0, hi = ns;                                  // the supermer x with so[x] <= t < so[x + 1]
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (so[mid] <= t) lo = mid; else hi = mid; }
            Key<2> sm; sm.w[0] = sk[lo][0]; sm.w[1] = sk[lo][1];
            const u32 nwin = supermer_windows(sm), j = t - so[lo];
            Key<1> x = sub_window<2, 1>(supermer_bases(sm), k, nwin, 1, j);
            if (RC) x = canonical(x, k);
            out_keys[base + t] = x.w[0];
            out_w[base + t] = sc[lo];
        }
    }
}

}  // namespace

// slots per read in the extraction's output: a 150-bp read at k = 31 makes ~13 runs; beyond `slots` a read's runs go to the spill list
uint32_t supermer_slots(uint32_t k, uint32_t read_len, uint32_t m) {
    const uint32_t W = read_len - k + 1, w = k - 2 - m + 1;
    const uint32_t expect = 2 * W / (w + 1) + 2;                  // (a minimizer changes about every (w + 1) / 2 windows)
    return std::min<uint32_t>(W, expect + expect / 3 + 2);        // (C3's reads: 12.9 runs on average, sd 1.8, 20 slots)
}
bool supermer_route_takes(uint32_t k, uint32_t read_len, uint32_t m) {
    return k >= m + 4 && k <= 31 && k - 1 - m <= 19 && read_len >= k && read_len - k + 1 <= 128 && (k - 2 - m + 1) <= SUPERMER_MAX_WINDOWS;
}

// a batch of reads -> their supermer records: out [n_reads * slots][2] (invalid where a read made fewer), the rest behind *spill_cursor
int dev_supermers_extract(const uint8_t* d_packed, uint64_t n_reads, uint32_t read_len, const uint8_t* d_skip, uint32_t k, uint32_t m, bool rc,
                          uint32_t n_owners, uint32_t slots, uint64_t* d_out, uint64_t* d_spill, uint64_t spill_cap, uint64_t* d_spill_cursor,
                          hipStream_t stream) {
    if (!supermer_route_takes(k, read_len, m) || n_owners == 0 || n_owners > 16) { set_error("supermers: k = %u, reads of %u bases, %u owners", k, read_len, n_owners); return KATOME_E_ARG; }
    if (n_reads == 0) return KATOME_OK;
    const uint32_t stride = (read_len + 3) / 4, w = k - 2 - m + 1;
    const uint32_t tile_dwords = ((SM_READS * stride + 15) / 16) * 4 + 8;
    const size_t lds = (size_t)(tile_dwords + (tile_dwords & 1)) * 4 + (size_t)SM_MAXR * SM_READS * 8 + 16;
    const uint64_t n_tiles = (n_reads + SM_READS - 1) / SM_READS;
    const dim3 grid(grid_for(n_tiles, 1, 256u * 64u)), block(SM_READS);
    unsigned long long* cur = reinterpret_cast<unsigned long long*>(d_spill_cursor);
#define KATOME_SM_LAUNCH(WCV)                                                                                                         \
    case WCV:                                                                                                                         \
        if (rc) hipLaunchKernelGGL((supermer_extract_kernel<true, WCV>), grid, block, lds, stream, d_packed, n_reads, read_len, stride, d_skip, k, m, n_owners,   \
                                   slots, d_out, d_spill, spill_cap, cur);                                                            \
        else    hipLaunchKernelGGL((supermer_extract_kernel<false, WCV>), grid, block, lds, stream, d_packed, n_reads, read_len, stride, d_skip, k, m, n_owners,  \
                                   slots, d_out, d_spill, spill_cap, cur);                                                            \
        break;
    switch (w) {          // (w = k - 1 - m: 3 for k = 15 ... 19 for k = 31)
        KATOME_SM_LAUNCH(3) KATOME_SM_LAUNCH(4) KATOME_SM_LAUNCH(5) KATOME_SM_LAUNCH(6) KATOME_SM_LAUNCH(7) KATOME_SM_LAUNCH(8) KATOME_SM_LAUNCH(9)
        KATOME_SM_LAUNCH(10) KATOME_SM_LAUNCH(11) KATOME_SM_LAUNCH(12) KATOME_SM_LAUNCH(13) KATOME_SM_LAUNCH(14) KATOME_SM_LAUNCH(15) KATOME_SM_LAUNCH(16)
        KATOME_SM_LAUNCH(17) KATOME_SM_LAUNCH(18) KATOME_SM_LAUNCH(19)
        default: set_error("supermers: no extraction kernel for k = %u (minimizers of %u bases)", k, m); return KATOME_E_UNSUPPORTED;
    }
#undef KATOME_SM_LAUNCH
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
        const dim3 grid(grid_for(n, SX, 256u * 32u)), block(BLOCK);
        if (rc) hipLaunchKernelGGL(supermer_expand_kernel<true>, grid, block, 0, stream, d_list, d_counts, offs.as<u64>(), n, k, keys.as<u64>(), weights.as<u32>());
        else    hipLaunchKernelGGL(supermer_expand_kernel<false>, grid, block, 0, stream, d_list, d_counts, offs.as<u64>(), n, k, keys.as<u64>(), weights.as<u32>());
        KCHECK_HIP(hipGetLastError());
    }
    return KATOME_OK;
}

}  // namespace katome
