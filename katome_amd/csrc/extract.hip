// extract.hip -- k-mer extraction kernels (gfx950).
//
// Restates, per window, compress_kmer / compress_kmer_with_rev_compl (reference
// src/katome/compress.rs:18-48) over `read.windows(K)` (collections/graphs/pt_graph.rs:294,310):
// one record per forward window.  With reverse_complement the record is min(k-mer, rc(k-mer));
// the reverse strand the reference inserts separately (pt_graph.rs:282-308) is re-created when
// the table is turned into edges (table.hip), so the edge multiset is the same.
//
// HBM-bound streaming kernel: algorithmic traffic per read = ceil(L/4) bytes in,
// (L-k+1) * 8*NW bytes out (998 B for L=150, k=31).  No MFMA: integer shift/compare work.
#include "common.h"

namespace katome {

// ---------------------------------------------------------------------------------------------
// Fixed-length reads, LDS-staged.  A workgroup (256 threads = 4 waves) owns tiles of TR reads (256 of them when a read is
// at most 64 bytes, else 64: with four 16-byte records per 150-base read a tile of 64 reads moves 6.5 KB per trip, and eight
// such workgroups per CU keep half of what 8 TB/s need in flight):
//   1. the tile's packed bytes are pulled in with 16-byte coalesced loads, each dword is
//      byte-swapped once and parked in LDS (so every later use sees "16 bases, first base on top");
//   2. every lane produces output records 2p, 2p+1 (NW=1) or record p (NW=2) of the tile's
//      contiguous output range, so each wave store is 64 x 16 contiguous bytes;
//      a record = 2*NW+1 LDS dwords -> funnel shift -> (canonical) key.
// ---------------------------------------------------------------------------------------------
constexpr u32 MARK_FLAG = 1u << 31;     // or-ed into `step`: tag records with RC_MARK (first-seen-order mode)

template <int NW, bool RC>
__device__ __forceinline__ Key<NW> record_from_lds(const u32* lds, const uint8_t* lds_skip, u32 i, u32 W, u32 magicW,
                                                    u32 stride_bytes, u32 k, u32 step, u32 win0) {
    u32 r = W == 1 ? i : __umulhi(i, magicW);   // i / W (magic multiply, exact for i, W < 2^16; W = 1 has no 32-bit magic)
    u32 w = win0 + (i - r * W) * (step & ~MARK_FLAG);
    if (lds_skip[r]) return key_invalid<NW>();
    u32 bit = r * stride_bytes * 8 + 2 * w;
    u32 di = bit >> 5, sh = bit & 31;
    u32 d[2 * NW + 1];
#pragma unroll
    for (int j = 0; j < 2 * NW + 1; ++j) d[j] = lds[di + j];
    Key<NW> key = extract_window(d, sh, k, (Key<NW>*)nullptr);
    if (RC) {
        bool flipped;
        key = canonical_flip(key, k, flipped);
        if (step & MARK_FLAG) key.w[0] |= flipped ? RC_MARK : 0;     // first-seen-order mode: remember the orientation
    }
    return key;
}

template <int NW, bool RC, int TILE_READS>
__global__ __launch_bounds__(BLOCK) void extract_fixed_kernel(const uint8_t* __restrict__ packed, u64 n_reads,
                                                               u32 stride_bytes, u32 k, u32 W, u32 magicW, u32 step, u32 win0,
                                                               const uint8_t* __restrict__ skip, u64* __restrict__ out) {
    extern __shared__ u32 lds[];
    const u32 tile_bytes_max = TILE_READS * stride_bytes;
    const u32 tile_dwords = ((tile_bytes_max + 15) / 16) * 4;
    uint8_t* lds_skip = (uint8_t*)(lds + tile_dwords + 8);
    const u64 total_bytes = n_reads * stride_bytes;
    const u64 n_tiles = (n_reads + TILE_READS - 1) / TILE_READS;
    const u32 tid = threadIdx.x;

    for (u64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const u64 r0 = tile * TILE_READS;
        const u32 nr = (u32)((n_reads - r0) < (u64)TILE_READS ? (n_reads - r0) : (u64)TILE_READS);
        const u64 byte0 = r0 * stride_bytes;
        const u32 nchunks = (nr * stride_bytes + 15) / 16;
        for (u32 c = tid; c < nchunks; c += BLOCK) {
            u64 off = byte0 + (u64)c * 16;
            uint4 v;
            if (off + 16 <= total_bytes) {
                v = *reinterpret_cast<const uint4*>(packed + off);
            } else {   // last chunk of the whole buffer: byte loads, zero fill
                u32 t[4] = {0, 0, 0, 0};
                for (u32 b = 0; b < 16 && off + b < total_bytes; ++b) t[b >> 2] |= (u32)packed[off + b] << (8 * (b & 3));
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            lds[c * 4 + 0] = __builtin_bswap32(v.x);
            lds[c * 4 + 1] = __builtin_bswap32(v.y);
            lds[c * 4 + 2] = __builtin_bswap32(v.z);
            lds[c * 4 + 3] = __builtin_bswap32(v.w);
        }
        if (tid < 8) lds[nchunks * 4 + tid] = 0;      // windows near the tile end read past it
        for (u32 t = tid; t < (u32)TILE_READS; t += BLOCK) lds_skip[t] = (skip && t < nr) ? skip[r0 + t] : 0;
        __syncthreads();

        const u32 nrec = nr * W;
        const u64 out0 = r0 * (u64)W;
        if (NW == 1) {
            for (u32 p = tid; 2 * p < nrec; p += BLOCK) {
                u32 i0 = 2 * p;
                Key<NW> a = record_from_lds<NW, RC>(lds, lds_skip, i0, W, magicW, stride_bytes, k, step, win0);
                if (i0 + 1 < nrec) {
                    Key<NW> b = record_from_lds<NW, RC>(lds, lds_skip, i0 + 1, W, magicW, stride_bytes, k, step, win0);
                    *reinterpret_cast<ulonglong2*>(out + out0 + i0) = make_ulonglong2(a.w[0], b.w[0]);
                } else {
                    out[out0 + i0] = a.w[0];
                }
            }
        } else {
            for (u32 i = tid; i < nrec; i += BLOCK) {
                Key<NW> a = record_from_lds<NW, RC>(lds, lds_skip, i, W, magicW, stride_bytes, k, step, win0);
                if (NW == 2) {
                    *reinterpret_cast<ulonglong2*>(out + (out0 + i) * 2) = make_ulonglong2(a.w[0], a.w[NW - 1]);
                } else {
#pragma unroll
                    for (int j = 0; j < NW; ++j) out[(out0 + i) * NW + j] = a.w[j];
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// General reads (variable length, or fixed length too long for the LDS tile): one record per
// thread, read located by binary search over the window prefix sums, bytes fetched through L1/L2.
// ---------------------------------------------------------------------------------------------
struct ArrayAddr {
    const u64* byte_off; const u32* len; const u64* rec_prefix; u64 n_reads; bool mark_;
    u32 k0, span, mode;          // mode 0: every window; 1: whole tiles of `span` windows; 2: the windows after the last whole tile
    __device__ __forceinline__ void locate(u64 i, u64& boff, u32& w) const {
        u64 lo = 0, hi = n_reads;            // largest r with rec_prefix[r] <= i
        while (hi - lo > 1) {
            u64 mid = (lo + hi) >> 1;
            if (rec_prefix[mid] <= i) lo = mid; else hi = mid;
        }
        boff = byte_off[lo];
        const u32 j = (u32)(i - rec_prefix[lo]);
        w = mode == 1 ? j * span : mode == 2 ? ((len[lo] - k0 + 1) / span) * span + j : j;
    }
    __device__ __forceinline__ bool skipped(u64) const { return false; }
    __device__ __forceinline__ bool mark() const { return mark_; }
};
struct FixedAddr {
    u64 stride_bytes; u64 W; const uint8_t* skip; u64 r; u32 step; u32 win0;
    __device__ __forceinline__ void locate(u64 i, u64& boff, u32& w) {
        r = i / W;
        boff = r * stride_bytes;
        w = win0 + (u32)(i - r * W) * (step & ~MARK_FLAG);
    }
    __device__ __forceinline__ bool mark() const { return (step & MARK_FLAG) != 0; }
    __device__ __forceinline__ bool skipped(u64) const { return skip && skip[r]; }
};

template <int NW, bool RC, class Addr>
__global__ __launch_bounds__(BLOCK) void extract_general_kernel(const uint8_t* __restrict__ packed, u64 packed_bytes,
                                                                 Addr addr, u64 total_windows, u32 k,
                                                                 u64* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < total_windows; i += (u64)gridDim.x * BLOCK) {
        u64 boff; u32 w;
        addr.locate(i, boff, w);
        Key<NW> key;
        if (addr.skipped(i)) {
            key = key_invalid<NW>();
        } else {
            u64 b0 = boff + (w >> 2);
            u32 sh = 2 * (w & 3);
            u32 d[2 * NW + 1];
#pragma unroll
            for (int j = 0; j < 2 * NW + 1; ++j) {
                u32 v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    u64 idx = b0 + 4 * j + b;
                    u32 byte = idx < packed_bytes ? packed[idx] : 0u;
                    v = (v << 8) | byte;
                }
                d[j] = v;
            }
            key = extract_window(d, sh, k, (Key<NW>*)nullptr);
            if (RC) {
                bool flipped;
                key = canonical_flip(key, k, flipped);
                if (addr.mark() && flipped) key.w[0] |= RC_MARK;
            }
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) out[i * NW + j] = key.w[j];
    }
}

// `k` is the length of the window that becomes a record; `step` the distance between consecutive window
// starts and W the records per read (step 1, W = L-k+1: every k-mer; step = span, W = (L-k0+1)/span with
// k = k0+span-1: the tiles of `span` consecutive k0-mers that the tiled counting path stores).
template <int NW, bool RC>
static int extract_fixed_t(const uint8_t* d_packed, u64 n_reads, u32 read_len, u32 k, u32 step, u32 W, u32 win0, const uint8_t* d_skip,
                           u64* d_records, hipStream_t stream) {
    const u32 stride = (read_len + 3) / 4;
    const u64 total = n_reads * (u64)W;
    if (total == 0) return KATOME_OK;
    const u32 tile_reads = stride <= 64 && 256ull * W < 65536 ? 256 : 64;
    const bool lds_ok = stride <= 256 && (u64)tile_reads * W < 65536 && ((uintptr_t)d_packed % 16 == 0) &&
                        ((uintptr_t)d_records % 16 == 0);
    if (lds_ok) {
        const u32 tile_dwords = ((tile_reads * stride + 15) / 16) * 4;
        const size_t lds_bytes = (tile_dwords + 8) * 4 + tile_reads;
        const u32 magicW = (u32)((1ull << 32) / W) + 1;
        const u64 n_tiles = (n_reads + tile_reads - 1) / tile_reads;
        unsigned grid = (unsigned)(n_tiles < 256u * 8u ? n_tiles : 256u * 8u);
        if (tile_reads == 256)
            hipLaunchKernelGGL((extract_fixed_kernel<NW, RC, 256>), dim3(grid), dim3(BLOCK), lds_bytes, stream, d_packed, n_reads,
                               stride, k, W, magicW, step, win0, d_skip, d_records);
        else
            hipLaunchKernelGGL((extract_fixed_kernel<NW, RC, 64>), dim3(grid), dim3(BLOCK), lds_bytes, stream, d_packed, n_reads,
                               stride, k, W, magicW, step, win0, d_skip, d_records);
    } else {
        FixedAddr a{stride, W, d_skip, 0, step, win0};
        hipLaunchKernelGGL((extract_general_kernel<NW, RC, FixedAddr>), dim3(grid_for(total, BLOCK)), dim3(BLOCK), 0, stream,
                           d_packed, n_reads * (u64)stride, a, total, k, d_records);
    }
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

// Records of fixed-length reads.  Record j of a read covers `span` consecutive windows starting at window
// first_window + j*span (as one (k+span-1)-mer); records_per_read = 0 means "as many as fit".
int launch_extract_fixed(uint32_t k, bool rc, const uint8_t* d_packed, uint64_t n_reads, uint32_t read_len,
                         const uint8_t* d_skip, uint64_t* d_records, hipStream_t stream, uint32_t span, bool mark,
                         uint32_t first_window, uint32_t records_per_read) {
    if (read_len < k) { set_error("Read is too short!"); return KATOME_E_SHORT_READ; }   // pt_graph.rs:278
    const u32 windows = read_len - k + 1;
    if (span == 0 || first_window > windows) { set_error("bad tile span %u / first window %u", span, first_window); return KATOME_E_ARG; }
    const u32 kk = k + span - 1, W = records_per_read ? records_per_read : (windows - first_window) / span;
    if (first_window + (u64)W * span > windows) { set_error("records reach past the last window of a read"); return KATOME_E_ARG; }
    if (W == 0) return KATOME_OK;
    const int nw = key_words_for_k(kk);
    const u32 step = span | (mark ? MARK_FLAG : 0u);
    if (nw == 1) return rc ? extract_fixed_t<1, true>(d_packed, n_reads, read_len, kk, step, W, first_window, d_skip, d_records, stream)
                           : extract_fixed_t<1, false>(d_packed, n_reads, read_len, kk, step, W, first_window, d_skip, d_records, stream);
    if (nw == 2) return rc ? extract_fixed_t<2, true>(d_packed, n_reads, read_len, kk, step, W, first_window, d_skip, d_records, stream)
                           : extract_fixed_t<2, false>(d_packed, n_reads, read_len, kk, step, W, first_window, d_skip, d_records, stream);
    return rc ? extract_fixed_t<3, true>(d_packed, n_reads, read_len, kk, step, W, first_window, d_skip, d_records, stream)
              : extract_fixed_t<3, false>(d_packed, n_reads, read_len, kk, step, W, first_window, d_skip, d_records, stream);
}

// d_rec_prefix: records before each read ([n_reads + 1]); mode / span as in ArrayAddr (mode 1 records are (k+span-1)-mers)
int launch_extract_var(uint32_t k0, bool rc, const uint8_t* d_packed, uint64_t packed_bytes, const uint64_t* d_byte_off,
                       const uint32_t* d_len, const uint64_t* d_rec_prefix, uint64_t n_reads, uint64_t total_windows,
                       uint64_t* d_records, hipStream_t stream, bool mark, uint32_t span, uint32_t mode) {
    if (total_windows == 0 || n_reads == 0) return KATOME_OK;
    if (span == 0 || mode > 2) { set_error("bad span / mode"); return KATOME_E_ARG; }
    ArrayAddr a{d_byte_off, d_len, d_rec_prefix, n_reads, mark, k0, span, mode};
    const uint32_t k = mode == 1 ? k0 + span - 1 : k0;
    const int nw = key_words_for_k(k);
    dim3 grid(grid_for(total_windows, BLOCK)), block(BLOCK);
    if (nw == 1) {
        if (rc) hipLaunchKernelGGL((extract_general_kernel<1, true, ArrayAddr>), grid, block, 0, stream, d_packed, packed_bytes, a, total_windows, k, d_records);
        else    hipLaunchKernelGGL((extract_general_kernel<1, false, ArrayAddr>), grid, block, 0, stream, d_packed, packed_bytes, a, total_windows, k, d_records);
    } else if (nw == 2) {
        if (rc) hipLaunchKernelGGL((extract_general_kernel<2, true, ArrayAddr>), grid, block, 0, stream, d_packed, packed_bytes, a, total_windows, k, d_records);
        else    hipLaunchKernelGGL((extract_general_kernel<2, false, ArrayAddr>), grid, block, 0, stream, d_packed, packed_bytes, a, total_windows, k, d_records);
    } else {                      // tiles of 64..95 bases
        if (rc) hipLaunchKernelGGL((extract_general_kernel<3, true, ArrayAddr>), grid, block, 0, stream, d_packed, packed_bytes, a, total_windows, k, d_records);
        else    hipLaunchKernelGGL((extract_general_kernel<3, false, ArrayAddr>), grid, block, 0, stream, d_packed, packed_bytes, a, total_windows, k, d_records);
    }
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

}  // namespace katome
