// probe_pair_store.hip -- isolates the "one tile in 5000 lost" finding of round 2 (table.hip, expand_tiles_kernel<.., TO_TABLE =
// false> with sequence numbers).  The loop body of the library kernel, word for word, in four variants of its last two stores:
//   V0  pair read from LDS BEFORE the key/weight stores, written as two 8-byte agent-scope stores   (what the library ships)
//   V1  pair read AFTER the key/weight stores, written as ONE 16-byte plain store                  (the variant that lost tiles)
//   V2  pair read AFTER the stores, written as two 8-byte plain stores
//   V3  pair read BEFORE the stores, written as one 16-byte plain store
// Every variant expands the same synthetic two-word tile table into poisoned output buffers; an order-free checksum of the
// records (key, weight, pair) is compared with one computed straight from the table (one thread per slot, no LDS, no cursor),
// and the output is scanned for poison left behind.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I katome_amd/csrc tools/probe_pair_store.hip -o tools/probe_pair_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kmer_bits.h"

using namespace katome;
constexpr int BLOCKP = 256;
constexpr u64 OCCB = 1ull << 63, LOCKB = 1ull << 62, KEYB = ~(OCCB | LOCKB);
struct Slot2P { u64 hi; u64 lo; u32 count; u32 pad[3]; };
constexpr u64 POISON = 0xEEEEEEEEEEEEEEEEull;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ u64 rec_hash(u64 key, u32 w, u64 a, u64 b) {
    return mix64(key ^ mix64(a + 0x1234567ull * w) ^ mix64(b ^ 0x9E3779B97F4A7C15ull));
}

__global__ void fill_kernel(Slot2P* slots, u64* seen, u64 cap, u32 k_tile, u64 seed) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (u64)gridDim.x * blockDim.x) {
        const u64 r = mix64(seed + i);
        Slot2P s{};
        if ((r & 1023) < 150) {                                  // ~15 % occupied, like C3's mid-tile table
            const u32 hi_bits = 2 * k_tile - 64;
            s.hi = (mix64(r + 1) & ((1ull << hi_bits) - 1)) | OCCB;
            s.lo = mix64(r + 2);
            s.count = (u32)(mix64(r + 3) % 100) + 1;
        }
        slots[i] = s;
        seen[2 * i] = mix64(r + 4) >> 28; seen[2 * i + 1] = mix64(r + 5) >> 28;
    }
}

__global__ void reference_kernel(const Slot2P* __restrict__ slots, const u64* __restrict__ seen, u64 cap, u32 k, u32 span, u32 stride,
                                 unsigned long long* sum, unsigned long long* cnt) {
    u64 acc = 0, n = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (u64)gridDim.x * blockDim.x) {
        const Slot2P s = slots[i];
        if (!(s.hi & OCCB)) continue;
        Key<2> tk; tk.w[0] = s.hi & KEYB; tk.w[1] = s.lo;
        for (u32 o = 0; o < span; ++o) {
            Key<1> x = sub_window<2, 1>(tk, k, span, stride, o);
            bool flipped = false;
            x = canonical_flip(x, k, flipped);
            const u64 f = seen[2 * i] + (u64)o * stride, r = seen[2 * i + 1] + (u64)(span - 1 - o) * stride;
            acc += rec_hash(x.w[0], s.count, flipped ? r : f, flipped ? f : r);
            ++n;
        }
    }
    atomicAdd(sum, (unsigned long long)acc);
    atomicAdd(cnt, (unsigned long long)n);
}

template <int V>
__global__ __launch_bounds__(BLOCKP) void expand_variant(const Slot2P* __restrict__ tiles, u64 tile_cap, u32 k, u32 span, u32 stride,
                                                          u64* __restrict__ out_keys, u32* __restrict__ out_w, u64* cursor,
                                                          const u64* __restrict__ tile_seen, u64* kmer_seen) {
    __shared__ u64 lkey[BLOCKP * 2];
    __shared__ u64 lseen[BLOCKP * 2];
    __shared__ u32 lcnt[BLOCKP];
    __shared__ u32 wtot[BLOCKP / 64];
    __shared__ u64 bbase;
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u64 i0 = (u64)blockIdx.x * BLOCKP; i0 < tile_cap; i0 += (u64)gridDim.x * BLOCKP) {
        const u64 i = i0 + threadIdx.x;
        Key<2> tile; u32 n = 0; bool have = false;
        if (i < tile_cap) {
            Slot2P s = tiles[i];
            tile.w[0] = s.hi & KEYB; tile.w[1] = s.lo; have = (s.hi & OCCB) != 0;
            n = s.count;
        }
        const u64 m = __ballot(have);
        const u32 before = __popcll(m & (lane ? (~0ull >> (64 - lane)) : 0ull));
        if (lane == 0) wtot[wave] = __popcll(m);
        __syncthreads();
        u32 woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < BLOCKP / 64; ++w) { if (w < (int)wave) woff += wtot[w]; total += wtot[w]; }
        if (have) {
            lkey[(woff + before) * 2] = tile.w[0]; lkey[(woff + before) * 2 + 1] = tile.w[1];
            lcnt[woff + before] = n;
            lseen[2 * (woff + before)] = tile_seen[2 * i]; lseen[2 * (woff + before) + 1] = tile_seen[2 * i + 1];
        }
        if (threadIdx.x == 0 && total) bbase = atomicAdd((unsigned long long*)cursor, (unsigned long long)total * span);
        __syncthreads();
        const u32 pairs = total * span;
        for (u32 p = threadIdx.x; p < pairs; p += BLOCKP) {
            const u32 t = p / span, o = p - t * span;
            Key<2> tk; tk.w[0] = lkey[t * 2]; tk.w[1] = lkey[t * 2 + 1];
            Key<1> x = sub_window<2, 1>(tk, k, span, stride, o);
            bool flipped = false;
            x = canonical_flip(x, k, flipped);
            u64 seq_fwd = 0, seq_rev = 0;
            if (V == 0 || V == 3) { seq_fwd = lseen[2 * t] + (u64)o * stride; seq_rev = lseen[2 * t + 1] + (u64)(span - 1 - o) * stride; }
            out_keys[bbase + p] = x.w[0];
            out_w[bbase + p] = lcnt[t];
            if (V == 1 || V == 2) { seq_fwd = lseen[2 * t] + (u64)o * stride; seq_rev = lseen[2 * t + 1] + (u64)(span - 1 - o) * stride; }
            if (V == 0) {
                __hip_atomic_store(&kmer_seen[2 * (bbase + p)], flipped ? seq_rev : seq_fwd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&kmer_seen[2 * (bbase + p) + 1], flipped ? seq_fwd : seq_rev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (V == 2) {
                kmer_seen[2 * (bbase + p)] = flipped ? seq_rev : seq_fwd;
                kmer_seen[2 * (bbase + p) + 1] = flipped ? seq_fwd : seq_rev;
            } else {
                ulonglong2 v; v.x = flipped ? seq_rev : seq_fwd; v.y = flipped ? seq_fwd : seq_rev;
                *reinterpret_cast<ulonglong2*>(&kmer_seen[2 * (bbase + p)]) = v;
            }
        }
        __syncthreads();
    }
}

__global__ void check_kernel(const u64* __restrict__ keys, const u32* __restrict__ w, const u64* __restrict__ pairs, u64 n,
                             unsigned long long* sum, unsigned long long* poison_keys, unsigned long long* poison_w,
                             unsigned long long* poison_pairs) {
    u64 acc = 0; u32 pk = 0, pw = 0, pp = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 key = keys[i]; const u32 wt = w[i]; const u64 a = pairs[2 * i], b = pairs[2 * i + 1];
        pk += key == POISON; pw += wt == (u32)POISON; pp += (a == POISON) + (b == POISON);
        acc += rec_hash(key, wt, a, b);
    }
    atomicAdd(sum, (unsigned long long)acc);
    if (pk) atomicAdd(poison_keys, (unsigned long long)pk);
    if (pw) atomicAdd(poison_w, (unsigned long long)pw);
    if (pp) atomicAdd(poison_pairs, (unsigned long long)pp);
}

int main(int argc, char** argv) {
    const u64 cap = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 25);
    const int rounds = argc > 2 ? atoi(argv[2]) : 6;
    const u32 k = 31, span = 6, stride = 1, k_tile = k + span - 1;
    Slot2P* slots; u64 *seen, *keys, *pairs, *cursor; u32* w; unsigned long long* acc;
    CK(hipMalloc(&slots, cap * sizeof(Slot2P))); CK(hipMalloc(&seen, cap * 16));
    const u64 max_rec = cap * span / 5;                         // 15 % occupancy: < 20 %
    CK(hipMalloc(&keys, max_rec * 8)); CK(hipMalloc(&w, max_rec * 4)); CK(hipMalloc(&pairs, max_rec * 16));
    CK(hipMalloc(&cursor, 8)); CK(hipMalloc(&acc, 64));
    int bad_total = 0;
    for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, slots, seen, cap, k_tile, 0xABCDEF12345ull * (r + 1));
        CK(hipMemset(acc, 0, 64));
        hipLaunchKernelGGL(reference_kernel, dim3(4096), dim3(256), 0, 0, slots, seen, cap, k, span, stride, acc, acc + 1);
        unsigned long long ref[2];
        CK(hipMemcpy(ref, acc, 16, hipMemcpyDeviceToHost));
        if (ref[1] > max_rec) { fprintf(stderr, "table too full\n"); return 2; }
        for (int v = 0; v < 4; ++v) {
            CK(hipMemset(keys, 0xEE, max_rec * 8)); CK(hipMemset(w, 0xEE, max_rec * 4)); CK(hipMemset(pairs, 0xEE, max_rec * 16));
            CK(hipMemset(cursor, 0, 8)); CK(hipMemset(acc, 0, 64));
            const dim3 grid(256u * 32u), block(BLOCKP);
            switch (v) {
                case 0: hipLaunchKernelGGL(expand_variant<0>, grid, block, 0, 0, slots, cap, k, span, stride, keys, w, cursor, seen, pairs); break;
                case 1: hipLaunchKernelGGL(expand_variant<1>, grid, block, 0, 0, slots, cap, k, span, stride, keys, w, cursor, seen, pairs); break;
                case 2: hipLaunchKernelGGL(expand_variant<2>, grid, block, 0, 0, slots, cap, k, span, stride, keys, w, cursor, seen, pairs); break;
                default: hipLaunchKernelGGL(expand_variant<3>, grid, block, 0, 0, slots, cap, k, span, stride, keys, w, cursor, seen, pairs); break;
            }
            CK(hipGetLastError());
            u64 n_rec = 0;
            CK(hipMemcpy(&n_rec, cursor, 8, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, keys, w, pairs, n_rec, acc, acc + 1, acc + 2, acc + 3);
            unsigned long long got[4];
            CK(hipMemcpy(got, acc, 32, hipMemcpyDeviceToHost));
            const bool ok = n_rec == ref[1] && got[0] == ref[0] && !got[1] && !got[2] && !got[3];
            printf("round %d V%d: records %llu (want %llu) checksum %s, poison left: keys %llu weights %llu pair words %llu -> %s\n", r, v,
                   (unsigned long long)n_rec, ref[1], got[0] == ref[0] ? "equal" : "DIFFERENT", got[1], got[2], got[3], ok ? "ok" : "BAD");
            bad_total += !ok;
        }
    }
    printf("probe_pair_store: %d bad runs of %d\n", bad_total, rounds * 4);
    return 0;
}
