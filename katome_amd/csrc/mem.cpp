// mem.cpp -- caching device-memory allocator behind every temporary of the build.
//
// hipMalloc / hipFree of multi-GiB buffers cost tens to hundreds of milliseconds each (~35 GB/s) and hipFree
// synchronises the device; a build allocates the tables, the batch scratch, sort ping-pong buffers and the result
// arrays.  Freed memory is therefore kept and handed out again: mem_pool.h holds the policy (segments, blocks cut from
// and merged back into them).  When a fresh hipMalloc runs out of memory the wholly free segments are given back to the
// driver and the request is tried once more (also on katome_dev_release_cache()).  A block remembers the stream it was
// last used on: handing it to a different stream first waits for that stream.
#include <chrono>
#include <mutex>

#include "common.h"
#include "mem_pool.h"

namespace katome {

namespace {
struct HipBackend {
    typedef hipStream_t Stream;
    bool trace = getenv("KATOME_TRACE_ALLOC") != nullptr;
    void* alloc(size_t bytes, int) {
        const auto t0 = std::chrono::steady_clock::now();
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        if (trace) fprintf(stderr, "[katome alloc] hipMalloc %.1f MiB %s in %.1f ms\n", bytes / 1048576.0, p ? "ok" : "FAILED",
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        return p;
    }
    void release(void* p) { (void)hipFree(p); }
    void sync(hipStream_t s) { (void)hipStreamSynchronize(s); }
};
struct Cache {
    std::mutex mu;
    SegmentPool<HipBackend> pool;
};
Cache& cache() { static Cache c; return c; }
}  // namespace

int dev_malloc(void** out, size_t bytes, hipStream_t stream) {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    int device = 0;
    (void)hipGetDevice(&device);
    void* p = c.pool.allocate(bytes, device, stream);
    static const bool log_all = getenv("KATOME_TRACE_BLOCKS") != nullptr;     // every block handed out / taken back (overlap hunting)
    if (log_all) fprintf(stderr, "[katome block] + %p %zu dev %d stream %p\n", p, SegmentPool<HipBackend>::round_size(bytes), device, (void*)stream);
    if (!p) {
        set_error("hipMalloc(%zu bytes) failed: out of device memory (%zu bytes held in segments, %zu of them free)", bytes,
                  c.pool.segment_bytes(), c.pool.free_bytes());
        *out = nullptr;
        return KATOME_E_OOM;
    }
    *out = p;
    return KATOME_OK;
}

void dev_free(void* p, hipStream_t stream) {
    if (!p) return;
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    static const bool log_all = getenv("KATOME_TRACE_BLOCKS") != nullptr;
    if (log_all) fprintf(stderr, "[katome block] - %p stream %p\n", p, (void*)stream);
    if (!c.pool.deallocate(p, stream, stream != nullptr)) (void)hipFree(p);
}

void dev_retire_stream(hipStream_t stream) {
    if (!stream) return;
    (void)hipStreamSynchronize(stream);
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    c.pool.retire_stream(stream, nullptr);
}

size_t dev_cached_bytes() {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    int device = 0;
    (void)hipGetDevice(&device);
    return c.pool.free_bytes_on(device);            // what the CURRENT device could get back (ranks of one process own one device each)
}

void dev_release_cache(int device) {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    (void)hipDeviceSynchronize();
    c.pool.release_free_segments(device);
}

void dev_cache_stats(int device, uint64_t out[3]) {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    out[0] = c.pool.segment_bytes_on(device);
    out[1] = c.pool.free_bytes_on(device);
    out[2] = c.pool.live_blocks_on(device);
}

}  // namespace katome
