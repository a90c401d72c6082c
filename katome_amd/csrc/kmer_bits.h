// kmer_bits.h -- bit-level k-mer arithmetic shared by every kernel (and compiled for the host
// by tests/hostshim so the same code is checked against the oracle without a GPU).
//
// Formats (all restating src/katome/compress.rs of the reference):
//   * a base is 2 bits, A=0 C=1 G=2 T=3                      (encode_fasta_symbol, compress.rs:347-378)
//   * packed reads: 4 bases per byte, first base in the two MOST significant bits, last byte
//     left-aligned                                            (compress_node, compress.rs:55-73)
//   * a k-mer KEY is the 2k bits of the window, right-aligned in NW 64-bit words, w[0] the most
//     significant word; NW = 1 for k <= 31, 2 for k <= 63.  Key order == lexicographic order of
//     the ACGT string, and the key's top 2(k-1) bits / low 2(k-1) bits are the source / target
//     node of the edge, i.e. the two halves of compress_kmer  (compress.rs:18-28).
//   * the reverse complement of a key is what compress_kmer_with_rev_compl builds out of
//     reverse_compressed_node (compress.rs:34-48,153-169): reverse the 2-bit groups, complement.
//   * an edge LABEL is compress_edge format: [pad][ceil(k/4) bytes, left-aligned] (compress.rs:250-271)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KD __host__ __device__ __forceinline__
#else
#define KD inline
#endif

namespace katome {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 INVALID_WORD = ~0ull;   // w[0] of a record that carries no k-mer
// first-seen-order mode only: set in w[0] of a record whose stored (canonical) key is the reverse complement of
// what the read holds; keys use at most 62 bits of w[0] (k <= 31: 2k bits; k <= 63: 2k-64 bits)
constexpr u64 RC_MARK = 1ull << 62;

template <int NW> struct Key { u64 w[NW]; };

// k-mers are at most 63 bases (two words); TILES of k-mers may take three (up to 95 bases: 190 bits, w[0] keeps its two
// top bits free for the table's flags, as it does for one and two words)
KD int key_words_for_k(u32 k) { return 2 * k <= 62 ? 1 : 2 * k <= 126 ? 2 : 3; }

// ---- hashing ------------------------------------------------------------------------------
KD u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
// mix64 is a bijection of the 64-bit words (murmur3's finalizer): a xor-shift by 33 undoes itself, the multipliers have inverses mod 2^64.
// table.hip's lds_count_packed_kernel stores a one-word k-mer as the low bits of its hash and gets it back with this
// (tests/test_kmer_bits_host.py: unmix64(mix64(x)) == x)
KD u64 unmix64(u64 x) {
    x ^= x >> 33; x *= 0x9cb4b2f8129337dbull;
    x ^= x >> 33; x *= 0x4f74430c22a54005ull;
    x ^= x >> 33;
    return x;
}
KD u64 hash_key(const Key<1>& k) { return mix64(k.w[0]); }
KD u64 hash_key(const Key<2>& k) { return mix64(k.w[1] ^ mix64(k.w[0] + 0x9E3779B97F4A7C15ull)); }
KD u64 hash_key(const Key<3>& k) { return mix64(k.w[2] ^ mix64(k.w[1] ^ mix64(k.w[0] + 0x9E3779B97F4A7C15ull))); }

KD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}
// owner / slot of a hash in [0, n)
KD u64 hash_to_range(u64 h, u64 n) { return mulhi64(h, n); }

// ---- comparisons --------------------------------------------------------------------------
KD bool key_eq(const Key<1>& a, const Key<1>& b) { return a.w[0] == b.w[0]; }
KD bool key_eq(const Key<2>& a, const Key<2>& b) { return a.w[0] == b.w[0] && a.w[1] == b.w[1]; }
KD bool key_eq(const Key<3>& a, const Key<3>& b) { return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2]; }
KD bool key_lt(const Key<1>& a, const Key<1>& b) { return a.w[0] < b.w[0]; }
KD bool key_lt(const Key<2>& a, const Key<2>& b) { return a.w[0] < b.w[0] || (a.w[0] == b.w[0] && a.w[1] < b.w[1]); }
KD bool key_lt(const Key<3>& a, const Key<3>& b) {
    if (a.w[0] != b.w[0]) return a.w[0] < b.w[0];
    if (a.w[1] != b.w[1]) return a.w[1] < b.w[1];
    return a.w[2] < b.w[2];
}
template <int NW> KD bool key_valid(const Key<NW>& a) { return a.w[0] != INVALID_WORD; }
template <int NW> KD Key<NW> key_invalid() { Key<NW> r; for (int i = 0; i < NW; ++i) r.w[i] = INVALID_WORD; return r; }

// ---- shifts -------------------------------------------------------------------------------
// (hi:lo) << s, top 64 bits; s in [0,63]
KD u64 shl_fill(u64 hi, u64 lo, u32 s) { return s ? (hi << s) | (lo >> (64 - s)) : hi; }

KD Key<1> key_shr(const Key<1>& a, u32 s) { Key<1> r; r.w[0] = s >= 64 ? 0 : a.w[0] >> s; return r; }
KD Key<2> key_shr(const Key<2>& a, u32 s) {
    Key<2> r;
    if (s == 0) return a;
    if (s >= 128) { r.w[0] = 0; r.w[1] = 0; }
    else if (s >= 64) { r.w[0] = 0; r.w[1] = a.w[0] >> (s - 64); }
    else { r.w[0] = a.w[0] >> s; r.w[1] = (a.w[1] >> s) | (a.w[0] << (64 - s)); }
    return r;
}
// (the words are picked by selects, not by a run-time index: an index into w[] made the compiler keep the key in scratch memory --
// 32-56 bytes per thread in the record kernels of three-word tiles, which ran at 0.2 of the HBM rate where their one- and two-word
// forms run at 0.5)
KD Key<3> key_shr(const Key<3>& a, u32 s) {
    Key<3> r;
    if (s >= 192) { r.w[0] = r.w[1] = r.w[2] = 0; return r; }
    const u32 ws = s >> 6, bs = s & 63;               // whole words, then bits
    const u64 x0 = ws == 0 ? a.w[0] : 0;
    const u64 x1 = ws == 0 ? a.w[1] : ws == 1 ? a.w[0] : 0;
    const u64 x2 = ws == 0 ? a.w[2] : ws == 1 ? a.w[1] : a.w[0];
    if (bs == 0) { r.w[0] = x0; r.w[1] = x1; r.w[2] = x2; return r; }
    r.w[0] = x0 >> bs;
    r.w[1] = (x1 >> bs) | (x0 << (64 - bs));
    r.w[2] = (x2 >> bs) | (x1 << (64 - bs));
    return r;
}
// keep the low `bits` bits
KD Key<1> key_low_bits(const Key<1>& a, u32 bits) { Key<1> r; r.w[0] = bits >= 64 ? a.w[0] : a.w[0] & ((1ull << bits) - 1); return r; }
KD Key<2> key_low_bits(const Key<2>& a, u32 bits) {
    Key<2> r = a;
    if (bits >= 128) return r;
    if (bits >= 64) r.w[0] = bits == 64 ? 0 : a.w[0] & ((1ull << (bits - 64)) - 1);
    else { r.w[0] = 0; r.w[1] = a.w[1] & ((1ull << bits) - 1); }
    return r;
}
KD Key<3> key_low_bits(const Key<3>& a, u32 bits) {
    Key<3> r = a;
    if (bits >= 192) return r;
    if (bits >= 128) { r.w[0] = bits == 128 ? 0 : a.w[0] & ((1ull << (bits - 128)) - 1); return r; }
    r.w[0] = 0;
    if (bits >= 64) { r.w[1] = bits == 64 ? 0 : a.w[1] & ((1ull << (bits - 64)) - 1); return r; }
    r.w[1] = 0; r.w[2] = a.w[2] & ((1ull << bits) - 1);
    return r;
}
// bits [shift, shift+nbits) of the key, nbits <= 32
KD u32 key_digit(const Key<1>& a, u32 shift, u32 nbits) { return (u32)(shift >= 64 ? 0 : (a.w[0] >> shift)) & ((1u << nbits) - 1); }
KD u32 key_digit(const Key<2>& a, u32 shift, u32 nbits) { return (u32)key_shr(a, shift).w[1] & ((1u << nbits) - 1); }
KD u32 key_digit(const Key<3>& a, u32 shift, u32 nbits) { return (u32)key_shr(a, shift).w[2] & ((1u << nbits) - 1); }

// ---- window extraction --------------------------------------------------------------------
// `d` holds 2*NW+1 consecutive 32-bit words of the packed read, each already byte-swapped so
// that its numeric value reads the 16 bases most-significant-first; the window starts `sh`
// bits (even, < 32) into d[0].  Returns the 2k-bit key, right-aligned.
KD Key<1> extract_window(const u32* d, u32 sh, u32 k, Key<1>*) {
    u64 w0 = ((u64)d[0] << 32) | d[1], w1 = (u64)d[2] << 32;
    Key<1> r; r.w[0] = shl_fill(w0, w1, sh) >> (64 - 2 * k);
    return r;
}
KD Key<2> extract_window(const u32* d, u32 sh, u32 k, Key<2>*) {
    u64 w0 = ((u64)d[0] << 32) | d[1], w1 = ((u64)d[2] << 32) | d[3], w2 = (u64)d[4] << 32;
    Key<2> x; x.w[0] = shl_fill(w0, w1, sh); x.w[1] = shl_fill(w1, w2, sh);
    return key_shr(x, 128 - 2 * k);
}

KD Key<3> extract_window(const u32* d, u32 sh, u32 k, Key<3>*) {
    u64 w0 = ((u64)d[0] << 32) | d[1], w1 = ((u64)d[2] << 32) | d[3], w2 = ((u64)d[4] << 32) | d[5], w3 = (u64)d[6] << 32;
    Key<3> x; x.w[0] = shl_fill(w0, w1, sh); x.w[1] = shl_fill(w1, w2, sh); x.w[2] = shl_fill(w2, w3, sh);
    return key_shr(x, 192 - 2 * k);
}

// ---- reverse complement -------------------------------------------------------------------
KD u64 brev64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
#endif
}
// reverse the order of the 32 two-bit groups of a word
KD u64 rev_groups64(u64 x) {
    u64 y = brev64(x);
    return ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
}
KD Key<1> revcomp(const Key<1>& a, u32 k) {
    Key<1> r; r.w[0] = (~rev_groups64(a.w[0])) >> (64 - 2 * k);
    return r;
}
KD Key<2> revcomp(const Key<2>& a, u32 k) {
    Key<2> x; x.w[0] = ~rev_groups64(a.w[1]); x.w[1] = ~rev_groups64(a.w[0]);
    return key_shr(x, 128 - 2 * k);
}
KD Key<3> revcomp(const Key<3>& a, u32 k) {
    Key<3> x; x.w[0] = ~rev_groups64(a.w[2]); x.w[1] = ~rev_groups64(a.w[1]); x.w[2] = ~rev_groups64(a.w[0]);
    return key_shr(x, 192 - 2 * k);
}
template <int NW> KD Key<NW> canonical(const Key<NW>& a, u32 k) {
    Key<NW> rc = revcomp(a, k);
    return key_lt(rc, a) ? rc : a;
}
// same, telling whether the reverse complement was taken
template <int NW> KD Key<NW> canonical_flip(const Key<NW>& a, u32 k, bool& flipped) {
    Key<NW> rc = revcomp(a, k);
    flipped = key_lt(rc, a);
    return flipped ? rc : a;
}

// ---- edge endpoints (the two halves of compress_kmer, compress.rs:23-26) --------------------
// Owner of a key among n ranks by its CORE: the `core` bases that end `shift/2` bases before the key's right end, taken
// canonically (the smaller of the core and its reverse complement).  For a k-mer, shift = 2 and core = k-2 name its
// middle (k-2)-mer: shared by the k-mer and its reverse complement (so a stored canonical k-mer and both oriented edges
// it stands for agree) and equal to the last k-2 bases of its source node.  For a (k-1)-mer node, shift = 0 and
// core = k-2 name that same tail: every out-edge of a node lives on the rank that owns the node.
// Owner of a record among n ranks by the whole record (tiles, mid tiles).  NOT hash_to_range(hash_key(.)): the owner's table
// puts a key at slot mulhi(hash_key, capacity), so a rank that owned one hash RANGE would fill one n-th of its table with all
// of its keys (eight ranks: the mid-tile table, sized for what arrives, ran 8 x over its load limit in that stretch and an
// insert of 1.6 M records took 2 s).  A second mix makes the owner independent of the slot.
template <int NW> KD u64 whole_key_owner(const Key<NW>& a, u64 n) { return hash_to_range(mix64(hash_key(a) ^ 0x5851F42D4C957F2Dull), n); }
// hash of a record's core (the `core` bases `shift` bits above its low end, either strand); the record's owner is a function of
// the top CORE_GROUP_BITS of it only, so that records ordered by those bits are ordered by owner as well (a rank's own distinct
// k-mers leave its LDS count already grouped by owner: table.hip, records_to_edges_sorted with an OwnerSplit)
constexpr u32 CORE_GROUP_BITS = 16;
template <int NW> KD u64 core_hash(const Key<NW>& a, u32 shift, u32 core) {
    const Key<NW> m = key_low_bits(key_shr(a, shift), 2 * core);
    return hash_key(canonical(m, core));
}
KD u64 core_group_owner(u64 group, u64 n) { return (group * n) >> CORE_GROUP_BITS; }          // group = top CORE_GROUP_BITS of core_hash
template <int NW> KD u64 core_owner(const Key<NW>& a, u32 shift, u32 core, u64 n) {
    return core_group_owner(core_hash(a, shift, core) >> (64 - CORE_GROUP_BITS), n);
}

// ---- owner by minimizer (the supermer route of the sharded build) -------------------------------------------
// A core's MINIMIZER: among its core - m + 1 m-mers, each taken canonically (the smaller of it and its reverse complement), the one
// whose hash is lowest.  A core and its reverse complement hold the same canonical m-mers, so the minimizer -- like core_hash -- is
// the same for a k-mer and its reverse complement; and consecutive windows of a read mostly share it (it changes about every
// (core - m + 2) / 2 windows), which is what lets a read travel as a dozen SUPERMERS -- runs of windows with one minimizer, hence one
// owner -- instead of as 120 k-mer records.
constexpr u64 MINIMIZER_SALT = 0xD6E8FEB86659FD93ull;
// (m <= 16: a canonical m-mer is a 32-bit value and its hash a 32-bit finalizer -- two multiplies -- because the hash runs once per
// position in the extraction kernel and core - m + 1 times per key wherever an owner is asked for pointwise)
KD u32 fmix32(u32 h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
KD u32 mmer_hash(u32 fwd, u32 rev) { return fmix32(fwd < rev ? fwd : rev); }
// lowest mmer_hash over the m-mers of the `core` bases that end `shift` bits above the key's low end (m <= 16, m <= core)
template <int NW> KD u32 core_minimizer(const Key<NW>& a, u32 shift, u32 core, u32 m) {
    const Key<NW> c = key_low_bits(key_shr(a, shift), 2 * core);
    const u32 mask = m >= 16 ? 0xFFFFFFFFu : (1u << (2 * m)) - 1;
    u32 f = 0, r = 0, best = 0xFFFFFFFFu;
    for (u32 j = 0; j < core; ++j) {
        const u32 base = key_digit(c, 2 * (core - 1 - j), 2);
        f = ((f << 2) | base) & mask;
        r = (r >> 2) | ((3u - base) << (2 * (m - 1)));
        if (j + 1 >= m) { const u32 h = mmer_hash(f, r); best = h < best ? h : best; }
    }
    return best;
}
KD u64 minimizer_owner_of(u32 min_hash, u64 n) { return hash_to_range(mix64((u64)min_hash ^ MINIMIZER_SALT), n); }
template <int NW> KD u64 minimizer_owner(const Key<NW>& a, u32 shift, u32 core, u32 m, u64 n) {
    return minimizer_owner_of(core_minimizer(a, shift, core, m), n);
}
// Supermer record (k <= 31: two words): the run's nwin + k - 1 bases right-aligned in the low bits (canonical when both strands are
// counted: a run and its reverse complement are one record), the owner in bits 116..119, nwin in bits 120..124 of the 128
constexpr u32 SUPERMER_OWNER_SHIFT = 52, SUPERMER_LEN_SHIFT = 56;        // (bit positions inside w[0])
constexpr u32 SUPERMER_MAX_WINDOWS = 24;                                  // nwin + k - 1 <= 54 bases = 108 bits
KD u32 supermer_windows(const Key<2>& s) { return (u32)(s.w[0] >> SUPERMER_LEN_SHIFT) & 31u; }
KD u32 supermer_owner(const Key<2>& s) { return (u32)(s.w[0] >> SUPERMER_OWNER_SHIFT) & 15u; }
KD Key<2> supermer_bases(const Key<2>& s) { Key<2> r = s; r.w[0] &= (1ull << SUPERMER_OWNER_SHIFT) - 1; return r; }

template <int NW> KD Key<NW> source_node(const Key<NW>& kmer) { return key_shr(kmer, 2); }
template <int NW> KD Key<NW> target_node(const Key<NW>& kmer, u32 k) { return key_low_bits(kmer, 2 * (k - 1)); }

// ---- tiles: a tile is the (k+span-1)-mer covering `span` consecutive k-mers of a read ------------
KD Key<1> narrow_key(const Key<1>& a, Key<1>*) { return a; }
KD Key<2> narrow_key(const Key<2>& a, Key<2>*) { return a; }
KD Key<1> narrow_key(const Key<2>& a, Key<1>*) { Key<1> r; r.w[0] = a.w[1]; return r; }
KD Key<3> narrow_key(const Key<3>& a, Key<3>*) { return a; }
KD Key<2> narrow_key(const Key<3>& a, Key<2>*) { Key<2> r; r.w[0] = a.w[1]; r.w[1] = a.w[2]; return r; }
KD Key<1> narrow_key(const Key<3>& a, Key<1>*) { Key<1> r; r.w[0] = a.w[2]; return r; }
// the o-th sub-window (o = 0 is the leftmost) of a tile made of n_sub windows of sub_len bases whose starts are
// `stride` bases apart (k-mers of a tile: sub_len = k, stride = 1; the smaller tiles of a big tile: stride = their span)
template <int NWT, int NWK> KD Key<NWK> sub_window(const Key<NWT>& tile, u32 sub_len, u32 n_sub, u32 stride, u32 o) {
    return narrow_key(key_low_bits(key_shr(tile, 2 * stride * (n_sub - 1 - o)), 2 * sub_len), (Key<NWK>*)nullptr);
}

// ---- compress_edge label (compress.rs:250-271) ----------------------------------------------
KD u32 label_stride_for_k(u32 k) { return 1 + (k + 3) / 4; }
KD u32 label_pad_for_k(u32 k) { return (4 - k % 4) % 4; }
// byte i (0-based, i < ceil(k/4)) of the left-aligned packed k-mer
KD uint8_t label_byte(const Key<1>& a, u32 k, u32 i) {
    u32 nb = (k + 3) / 4;
    u64 v = a.w[0] << (2 * label_pad_for_k(k));      // now exactly 8*nb bits, right-aligned
    return (uint8_t)(v >> (8 * (nb - 1 - i)));
}
KD uint8_t label_byte(const Key<2>& a, u32 k, u32 i) {
    u32 nb = (k + 3) / 4, pad2 = 2 * label_pad_for_k(k);
    Key<2> v; v.w[0] = shl_fill(a.w[0], a.w[1], pad2); v.w[1] = a.w[1] << pad2;
    return (uint8_t)key_shr(v, 8 * (nb - 1 - i)).w[1];
}

// ---- synthetic workload (DESIGN.md "Synthetic workload"; same definition as the oracle's) ----
KD u64 splitmix64(u64 x) {
    u64 z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace katome
