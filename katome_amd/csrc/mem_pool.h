// mem_pool.h -- the bookkeeping of the caching device allocator (mem.cpp), free of HIP so that it can be exercised on the
// host (tests/hostshim/mem_pool_host.cpp).
//
// Memory comes from the backend in SEGMENTS (one hipMalloc each).  A segment is a chain of blocks in address order; a
// freed block merges with free neighbours of its segment, a request takes the smallest free block that holds it and
// leaves the rest of that block free (if the rest is worth keeping).  So the buffers a build allocates after it has
// given its tables back are cut out of the tables' segments instead of being new segments each: the first build in a
// process asks the driver for about its peak working set, not for the sum of every distinct buffer size (hipMalloc
// runs at ~35 GB/s on MI355X, hipFree of a cache of 100+ GiB takes seconds).  Small requests have their own exact-size
// cache so that a few live bytes do not pin the middle of a large segment.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <unordered_map>
#include <vector>

namespace katome {

template <class Backend>            // Backend: void* alloc(size_t, int device); void release(void*); void sync(Stream)
class SegmentPool {
public:
    typedef typename Backend::Stream Stream;
    static constexpr size_t SMALL = 8u << 20;           // requests below this: exact-size cache
    static constexpr size_t KEEP = 32u << 20;           // a remainder smaller than this stays with the block handed out

    explicit SegmentPool(Backend b = Backend()) : backend_(b) {}
    ~SegmentPool() {}                                   // (device memory is left to process teardown, as before)

    static size_t round_size(size_t n) {
        if (n < 512) n = 512;
        const size_t g = n >= SMALL ? (2u << 20) : 512;
        return (n + g - 1) / g * g;
    }

    // nullptr: the backend is out of memory even after every wholly free segment was given back
    void* allocate(size_t bytes, int device, Stream stream) {
        const size_t want = round_size(bytes);
        Block* b = take_free(want, device);
        if (!b) {
            void* p = backend_.alloc(want, device);
            if (!p) {
                release_free_segments(device);
                p = backend_.alloc(want, device);
                if (!p) return nullptr;
            }
            segment_bytes_ += want;
            b = new Block();
            b->p = static_cast<char*>(p); b->bytes = want; b->seg_base = b->p; b->seg_bytes = want; b->device = device;
            b->stream = stream; b->prev = b->next = nullptr;
        } else {
            if (b->stream != stream) backend_.sync(b->stream);
            b->stream = stream;
        }
        b->is_free = false;
        live_[b->p] = b;
        return b->p;
    }

    // false: not one of ours
    bool deallocate(void* p, Stream stream, bool have_stream) {
        auto it = live_.find(p);
        if (it == live_.end()) return false;
        Block* b = it->second;
        live_.erase(it);
        if (have_stream) b->stream = stream;
        b->is_free = true;
        if (b->next && b->next->is_free) { remove_free(b->next); absorb_next(b); }
        if (b->prev && b->prev->is_free) { Block* before = b->prev; remove_free(before); absorb_next(before); b = before; }
        insert_free(b);
        return true;
    }

    void release_free_segments(int device) {                            // device < 0: all
        for (auto it = free_.begin(); it != free_.end();) {
            Block* b = it->second;
            if ((device < 0 || b->device == device) && b->p == b->seg_base && b->bytes == b->seg_bytes) {
                backend_.release(b->seg_base);
                segment_bytes_ -= b->seg_bytes;
                free_bytes_ -= b->bytes;
                it = free_.erase(it);
                delete b;
            } else ++it;
        }
    }

    // a stream is about to be destroyed (its work is complete): blocks last used on it no longer need a wait -- and must
    // not name it any more
    void retire_stream(Stream s, Stream idle) {
        for (auto& kv : free_) if (kv.second->stream == s) kv.second->stream = idle;
        for (auto& kv : live_) if (kv.second->stream == s) kv.second->stream = idle;
    }

    size_t free_bytes() const { return free_bytes_; }                  // cached: free blocks, whole segments or parts
    size_t free_bytes_on(int device) const {                           // ... of one device
        size_t t = 0;
        for (const auto& kv : free_) if (kv.second->device == device) t += kv.second->bytes;
        return t;
    }
    size_t segment_bytes() const { return segment_bytes_; }            // everything taken from the backend
    size_t segment_bytes_on(int device) const {                        // ... on one device: its free + its live blocks
        size_t t = free_bytes_on(device);
        for (const auto& kv : live_) if (kv.second->device == device) t += kv.second->bytes;
        return t;
    }
    size_t live_blocks_on(int device) const {
        size_t t = 0;
        for (const auto& kv : live_) if (kv.second->device == device) ++t;
        return t;
    }
    size_t live_blocks() const { return live_.size(); }
    size_t free_blocks() const { return free_.size(); }

private:
    struct Block {
        char* p; size_t bytes;
        char* seg_base; size_t seg_bytes;
        int device; Stream stream; bool is_free;
        Block *prev, *next;                                             // address order inside the segment
    };

    Block* take_free(size_t want, int device) {
        for (auto it = free_.lower_bound(want); it != free_.end(); ++it) {
            Block* b = it->second;
            if (b->device != device) continue;
            const bool small_req = want < SMALL, small_blk = b->seg_bytes < SMALL;
            if (small_req != small_blk) { if (small_req) return nullptr; continue; }   // the two caches do not mix
            if (small_req && b->bytes != want) return nullptr;                         // exact sizes only
            free_.erase(it);
            free_bytes_ -= b->bytes;
            if (!small_req && b->bytes - want >= KEEP) {                                // cut the request off the front
                Block* rest = new Block(*b);
                rest->p = b->p + want; rest->bytes = b->bytes - want; rest->is_free = true;
                rest->prev = b; rest->next = b->next;
                if (b->next) b->next->prev = rest;
                b->next = rest; b->bytes = want;
                insert_free(rest);
            }
            return b;
        }
        return nullptr;
    }
    void absorb_next(Block* b) {                                        // neither b nor b->next is listed as free
        Block* n = b->next;
        if (n->stream != b->stream) backend_.sync(n->stream);
        b->bytes += n->bytes;
        b->next = n->next;
        if (n->next) n->next->prev = b;
        delete n;
    }
    void insert_free(Block* b) { free_.emplace(b->bytes, b); free_bytes_ += b->bytes; }
    void remove_free(Block* b) {
        auto range = free_.equal_range(b->bytes);
        for (auto it = range.first; it != range.second; ++it)
            if (it->second == b) { free_.erase(it); free_bytes_ -= b->bytes; return; }
    }

    Backend backend_;
    std::multimap<size_t, Block*> free_;
    std::unordered_map<void*, Block*> live_;
    size_t free_bytes_ = 0, segment_bytes_ = 0;
};

}  // namespace katome
