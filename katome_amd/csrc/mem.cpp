// mem.cpp -- caching device-memory allocator behind every temporary of the build.
//
// hipMalloc / hipFree of multi-GiB buffers cost tens to hundreds of milliseconds each and hipFree
// synchronises the device; a build allocates the table, the batch scratch, sort ping-pong buffers
// and the result arrays.  Freed blocks are therefore kept per device and handed out again
// (best fit, <= 25 % slack).  When a fresh hipMalloc runs out of memory the smallest cached block that is big
// enough is handed out whatever its slack (giving cached blocks back to the driver costs seconds once tens of
// GiB are cached); only if there is none is the cache released (also on katome_dev_release_cache()).  A block
// remembers the stream it was last used on: handing it to a different stream first waits for that stream.
#include <map>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace katome {

namespace {
struct Block { void* p; size_t bytes; hipStream_t stream; int device; };
struct Cache {
    std::mutex mu;
    std::multimap<size_t, Block> free_blocks;              // by size
    std::unordered_map<void*, Block> live;
    size_t cached_bytes = 0;
    bool trace = getenv("KATOME_TRACE_ALLOC") != nullptr;
};
Cache& cache() { static Cache c; return c; }

size_t round_size(size_t n) {
    if (n < 512) n = 512;
    const size_t g = n >= (8u << 20) ? (2u << 20) : 512;    // 2 MiB granules for large blocks
    return (n + g - 1) / g * g;
}

void release_all_locked(Cache& c, int device) {
    for (auto it = c.free_blocks.begin(); it != c.free_blocks.end();) {
        if (device < 0 || it->second.device == device) {
            (void)hipFree(it->second.p);
            c.cached_bytes -= it->second.bytes;
            it = c.free_blocks.erase(it);
        } else ++it;
    }
}
}  // namespace

int dev_malloc(void** out, size_t bytes, hipStream_t stream) {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    int device = 0;
    (void)hipGetDevice(&device);
    const size_t want = round_size(bytes);
    for (auto it = c.free_blocks.lower_bound(want); it != c.free_blocks.end() && it->first <= want + want / 4; ++it) {
        if (it->second.device != device) continue;
        Block b = it->second;
        c.free_blocks.erase(it);
        c.cached_bytes -= b.bytes;
        if (b.stream != stream) (void)hipStreamSynchronize(b.stream);
        b.stream = stream;
        c.live[b.p] = b;
        *out = b.p;
        return KATOME_OK;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {                  // out of memory: any cached block that is big enough will do
        (void)hipGetLastError();
        for (auto it = c.free_blocks.lower_bound(want); it != c.free_blocks.end(); ++it) {
            if (it->second.device != device) continue;
            Block b = it->second;
            c.free_blocks.erase(it);
            c.cached_bytes -= b.bytes;
            if (b.stream != stream) (void)hipStreamSynchronize(b.stream);
            b.stream = stream;
            c.live[b.p] = b;
            if (c.trace) fprintf(stderr, "[katome alloc] %.1f MiB served from a cached block of %.1f MiB\n", want / 1048576.0, b.bytes / 1048576.0);
            *out = b.p;
            return KATOME_OK;
        }
        (void)hipDeviceSynchronize();       // none: give the cached blocks back and try once more
        release_all_locked(c, device);
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        *out = nullptr;
        return KATOME_E_OOM;
    }
    if (c.trace) fprintf(stderr, "[katome alloc] hipMalloc %.1f MiB\n", want / 1048576.0);
    c.live[p] = Block{p, want, stream, device};
    *out = p;
    return KATOME_OK;
}

void dev_free(void* p, hipStream_t stream) {
    if (!p) return;
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.live.find(p);
    if (it == c.live.end()) { (void)hipFree(p); return; }
    Block b = it->second;
    c.live.erase(it);
    if (stream) b.stream = stream;
    c.free_blocks.emplace(b.bytes, b);
    c.cached_bytes += b.bytes;
}

size_t dev_cached_bytes() {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    return c.cached_bytes;
}

void dev_release_cache(int device) {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    (void)hipDeviceSynchronize();
    release_all_locked(c, device);
}

}  // namespace katome
