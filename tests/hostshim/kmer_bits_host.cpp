// Host build of katome_amd/csrc/kmer_bits.h so that the exact bit arithmetic the kernels use can
// be checked against the oracle on a machine without a GPU (test infrastructure).
#include <stdint.h>
#include <string.h>

#include "../../katome_amd/csrc/kmer_bits.h"

using namespace katome;

template <int NW> static void extract_t(const uint8_t* packed, uint64_t nbytes, uint64_t byte_off, uint32_t w, uint32_t k, uint64_t* out) {
    // same dword assembly as extract_general_kernel (extract.hip)
    uint64_t b0 = byte_off + (w >> 2);
    uint32_t sh = 2 * (w & 3);
    uint32_t d[2 * NW + 1];
    for (int j = 0; j < 2 * NW + 1; ++j) {
        uint32_t v = 0;
        for (int b = 0; b < 4; ++b) {
            uint64_t idx = b0 + 4 * j + b;
            v = (v << 8) | (idx < nbytes ? packed[idx] : 0u);
        }
        d[j] = v;
    }
    Key<NW> key = extract_window(d, sh, k, (Key<NW>*)nullptr);
    for (int i = 0; i < NW; ++i) out[i] = key.w[i];
}
template <int NW> static Key<NW> ld(const uint64_t* p) { Key<NW> k; for (int i = 0; i < NW; ++i) k.w[i] = p[i]; return k; }
template <int NW> static void st(uint64_t* p, const Key<NW>& k) { for (int i = 0; i < NW; ++i) p[i] = k.w[i]; }

extern "C" {
int hs_key_words(uint32_t k) { return key_words_for_k(k); }
void hs_extract(const uint8_t* packed, uint64_t nbytes, uint64_t byte_off, uint32_t w, uint32_t k, uint64_t* out) {
    const int nw = key_words_for_k(k);
    if (nw == 1) extract_t<1>(packed, nbytes, byte_off, w, k, out); else if (nw == 2) extract_t<2>(packed, nbytes, byte_off, w, k, out);
    else extract_t<3>(packed, nbytes, byte_off, w, k, out);
}
// dword-aligned variant: what extract_fixed_kernel does with its LDS image
void hs_extract_aligned(const uint8_t* packed, uint64_t nbytes, uint64_t bit_off, uint32_t k, uint64_t* out) {
    uint32_t di = (uint32_t)(bit_off >> 5), sh = (uint32_t)(bit_off & 31);
    uint32_t d[7];
    int nw = key_words_for_k(k);
    for (int j = 0; j < 2 * nw + 1; ++j) {
        uint32_t v = 0;
        for (int b = 0; b < 4; ++b) { uint64_t idx = (uint64_t)(di + j) * 4 + b; v = (v << 8) | (idx < nbytes ? packed[idx] : 0u); }
        d[j] = v;
    }
    if (nw == 1) { Key<1> key = extract_window(d, sh, k, (Key<1>*)nullptr); out[0] = key.w[0]; }
    else if (nw == 2) { Key<2> key = extract_window(d, sh, k, (Key<2>*)nullptr); out[0] = key.w[0]; out[1] = key.w[1]; }
    else { Key<3> key = extract_window(d, sh, k, (Key<3>*)nullptr); out[0] = key.w[0]; out[1] = key.w[1]; out[2] = key.w[2]; }
}
void hs_revcomp(const uint64_t* in, uint32_t k, uint64_t* out) {
    const int nw = key_words_for_k(k);
    if (nw == 1) st<1>(out, revcomp(ld<1>(in), k)); else if (nw == 2) st<2>(out, revcomp(ld<2>(in), k)); else st<3>(out, revcomp(ld<3>(in), k));
}
void hs_canonical(const uint64_t* in, uint32_t k, uint64_t* out) {
    const int nw = key_words_for_k(k);
    if (nw == 1) st<1>(out, canonical(ld<1>(in), k)); else if (nw == 2) st<2>(out, canonical(ld<2>(in), k)); else st<3>(out, canonical(ld<3>(in), k));
}
// sub-window o of a tile of n_sub windows of sub_len bases, `stride` bases apart (kmer_bits.h sub_window); nwt / nwk words
void hs_sub_window(const uint64_t* tile, int nwt, int nwk, uint32_t sub_len, uint32_t n_sub, uint32_t stride, uint32_t o, uint64_t* out) {
    if (nwt == 3 && nwk == 3) st<3>(out, sub_window<3, 3>(ld<3>(tile), sub_len, n_sub, stride, o));
    else if (nwt == 3 && nwk == 2) st<2>(out, sub_window<3, 2>(ld<3>(tile), sub_len, n_sub, stride, o));
    else if (nwt == 3 && nwk == 1) st<1>(out, sub_window<3, 1>(ld<3>(tile), sub_len, n_sub, stride, o));
    else if (nwt == 2 && nwk == 2) st<2>(out, sub_window<2, 2>(ld<2>(tile), sub_len, n_sub, stride, o));
    else if (nwt == 2 && nwk == 1) st<1>(out, sub_window<2, 1>(ld<2>(tile), sub_len, n_sub, stride, o));
    else st<1>(out, sub_window<1, 1>(ld<1>(tile), sub_len, n_sub, stride, o));
}
void hs_endpoints(const uint64_t* in, uint32_t k, uint64_t* src, uint64_t* dst) {
    if (key_words_for_k(k) == 1) { st<1>(src, source_node(ld<1>(in))); st<1>(dst, target_node(ld<1>(in), k)); }
    else { st<2>(src, source_node(ld<2>(in))); st<2>(dst, target_node(ld<2>(in), k)); }
}
void hs_label(const uint64_t* in, uint32_t k, uint8_t* out) {
    uint32_t nb = (k + 3) / 4;
    out[0] = (uint8_t)label_pad_for_k(k);
    for (uint32_t i = 0; i < nb; ++i) out[1 + i] = key_words_for_k(k) == 1 ? label_byte(ld<1>(in), k, i) : label_byte(ld<2>(in), k, i);
}
uint64_t hs_hash(const uint64_t* in, int nw) { return nw == 1 ? hash_key(ld<1>(in)) : nw == 2 ? hash_key(ld<2>(in)) : hash_key(ld<3>(in)); }
uint64_t hs_owner(const uint64_t* in, int nw, uint64_t n) { return nw == 1 ? whole_key_owner(ld<1>(in), n) : nw == 2 ? whole_key_owner(ld<2>(in), n) : whole_key_owner(ld<3>(in), n); }
uint64_t hs_core_owner(const uint64_t* in, int nw, uint32_t shift, uint32_t core, uint64_t n) {
    return nw == 1 ? core_owner(ld<1>(in), shift, core, n) : core_owner(ld<2>(in), shift, core, n);
}
uint32_t hs_digit(const uint64_t* in, int nw, uint32_t shift, uint32_t bits) { return nw == 1 ? key_digit(ld<1>(in), shift, bits) : key_digit(ld<2>(in), shift, bits); }
uint64_t hs_splitmix64(uint64_t x) { return splitmix64(x); }
uint64_t hs_mix64(uint64_t x) { return mix64(x); }
uint64_t hs_unmix64(uint64_t x) { return unmix64(x); }
}
