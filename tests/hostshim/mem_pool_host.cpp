// Host build of katome_amd/csrc/mem_pool.h over malloc, for tests/test_mem_pool_host.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

#include "../../katome_amd/csrc/mem_pool.h"

namespace {
struct FakeBackend {
    typedef int Stream;
    static size_t& budget() { static size_t b = ~(size_t)0; return b; }
    static size_t& used() { static size_t u = 0; return u; }
    static long& allocs() { static long n = 0; return n; }
    static long& syncs() { static long n = 0; return n; }
    static std::map<void*, size_t>& sizes() { static std::map<void*, size_t> m; return m; }
    // address space only: the test never touches the memory, so "segments" are reservations of fake addresses
    void* alloc(size_t bytes, int) {
        if (used() + bytes > budget()) return nullptr;
        static uintptr_t next = 1ull << 40;
        void* p = reinterpret_cast<void*>(next);
        next += bytes + (1ull << 30);
        used() += bytes; allocs() += 1; sizes()[p] = bytes;
        return p;
    }
    void release(void* p) { used() -= sizes()[p]; sizes().erase(p); }
    void sync(int) { syncs() += 1; }
};
katome::SegmentPool<FakeBackend>* pool() { static katome::SegmentPool<FakeBackend> p; return &p; }
}  // namespace

extern "C" {
void hs_pool_budget(uint64_t bytes) { FakeBackend::budget() = bytes; }
uint64_t hs_pool_alloc(uint64_t bytes, int stream) { return (uint64_t)(uintptr_t)pool()->allocate(bytes, 0, stream); }
int hs_pool_free(uint64_t p, int stream) { return pool()->deallocate((void*)(uintptr_t)p, stream, true) ? 1 : 0; }
void hs_pool_release() { pool()->release_free_segments(-1); }
uint64_t hs_pool_round(uint64_t bytes) { return katome::SegmentPool<FakeBackend>::round_size(bytes); }
void hs_pool_stats(uint64_t* out) {      // backend bytes, backend allocations, free bytes, segment bytes, live blocks, free blocks, syncs
    out[0] = FakeBackend::used(); out[1] = (uint64_t)FakeBackend::allocs(); out[2] = pool()->free_bytes();
    out[3] = pool()->segment_bytes(); out[4] = pool()->live_blocks(); out[5] = pool()->free_blocks(); out[6] = (uint64_t)FakeBackend::syncs();
}
}
