// Second probe for the lost-tile finding (see probe_top_vgpr.hip, profiles/r03_pair_store.md): the top register of a 32-VGPR
// allocation written by v_mul_lo_u32 (a multi-pass VALU operation, as in the library kernel) with an LDS read in flight, right
// after a workgroup barrier behind which one wave arrives late (it waits for a returning global atomic), with global stores of
// the previous trip still in flight -- the shape of expand_tiles_kernel's inner loop.  The control uses the same code with the
// allocation raised to 40 registers.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32;
typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr u32 MAXLOG = 4096, LOGW = 8;

#define CANARY(NAME, TOP)                                                                                                          \
    __global__ __launch_bounds__(256) void NAME(u32* log, u32* count, u64* cursor, u64* sink, u32 outer, u32 two) {               \
        __shared__ uint4 tile[256];                                                                                                \
        __shared__ u64 base;                                                                                                       \
        const u32 tid = threadIdx.x;                                                                                               \
        asm volatile("v_mov_b32 " TOP ", 0" ::: TOP);                                                                              \
        u32 wrong = 0, got_first = 0, want_first = 0, it_first = 0;                                                                \
        for (u32 it = 0; it < outer; ++it) {                                                                                       \
            tile[tid] = make_uint4(tid * 7u + it, it, tid, 0x1234u);                                                               \
            __syncthreads();                                                                                                       \
            if (tid == 0) base = atomicAdd(cursor, 256ull);                                                                        \
            __syncthreads();                                                                                                       \
            for (u32 p = tid; p < 512; p += 256) {                                                                                 \
                const u32 x = (p * 2654435761u) >> 26;                       /* 0..63 */                                           \
                u32 got; uint4 t;                                                                                                  \
                asm volatile("ds_read_b128 %1, %2\n\t"                                                                             \
                             "v_mul_lo_u32 v31, %3, %4\n\t"                                                                        \
                             "v_cmp_ne_u32_e32 vcc, 0, v31\n\t"                                                                    \
                             "s_waitcnt lgkmcnt(0)\n\t"                                                                            \
                             "v_mov_b32 %0, v31"                                                                                   \
                             : "=v"(got), "=v"(t) : "v"((p & 255u) * 16u + (u32)(size_t)0), "v"(x), "s"(two) : "v31", "vcc", "memory"); \
                if (got != x * two) { if (!wrong) { got_first = got; want_first = x * two; it_first = it * 2 + (p >> 8); } ++wrong; } \
                const u64 at = (base + p) % (1ull << 22);                                                                          \
                sink[at * 2] = t.x + got; sink[at * 2 + 1] = t.y;                /* stores in flight over the next trip */         \
            }                                                                                                                      \
            __syncthreads();                                                                                                       \
        }                                                                                                                          \
        if (wrong) {                                                                                                               \
            const u32 i = atomicAdd(count, 1u);                                                                                    \
            if (i < MAXLOG) {                                                                                                      \
                u32* l = log + i * LOGW;                                                                                           \
                l[0] = blockIdx.x; l[1] = tid; l[2] = wrong; l[3] = got_first; l[4] = it_first; l[5] = want_first;                 \
                l[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); l[7] = 0;                                                        \
            }                                                                                                                      \
        }                                                                                                                          \
    }

CANARY(canary_top_of_32, "v31")
CANARY(canary_in_40, "v39")

typedef void (*kernel_t)(u32*, u32*, u64*, u64*, u32, u32);

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20;
    const u32 outer = argc > 2 ? (u32)atoi(argv[2]) : 200;
    u32 *log = nullptr, *count = nullptr; u64 *cursor = nullptr, *sink = nullptr;
    CHECK(hipMalloc(&log, MAXLOG * LOGW * 4)); CHECK(hipMalloc(&count, 4)); CHECK(hipMalloc(&cursor, 8)); CHECK(hipMalloc(&sink, (2ull << 22) * 8));
    struct { const char* name; kernel_t k; } kernels[] = {{"v31 = top of a 32-register allocation", canary_top_of_32}, {"v31 inside a 40-register allocation", canary_in_40}};
    for (auto& kn : kernels) {
        hipFuncAttributes fa;
        CHECK(hipFuncGetAttributes(&fa, (const void*)kn.k));
        unsigned long long lanes = 0, lanes_lane = 0, lanes_tid = 0;
        std::vector<u32> first;
        for (int l = 0; l < launches; ++l) {
            CHECK(hipMemset(count, 0, 4)); CHECK(hipMemset(cursor, 0, 8));
            hipLaunchKernelGGL(kn.k, dim3(2048), dim3(256), 0, 0, log, count, cursor, sink, outer, 2u);
            CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
            u32 n = 0;
            CHECK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost));
            if (!n) continue;
            std::vector<u32> h(std::min(n, MAXLOG) * LOGW);
            CHECK(hipMemcpy(h.data(), log, h.size() * 4, hipMemcpyDeviceToHost));
            lanes += n;
            for (u32 i = 0; i < std::min(n, MAXLOG); ++i) {
                const u32* r = &h[i * LOGW];
                if (r[3] == (r[1] & 63)) ++lanes_lane;
                if (r[3] == r[1]) ++lanes_tid;
                if (first.size() < 16 * LOGW) first.insert(first.end(), r, r + LOGW);
            }
        }
        printf("%-40s numRegs %3d: %d launches x 2048 blocks x %u barriers x 2 trips: lanes that read v31 wrong %llu (read == lane id %llu, == thread id %llu)\n", kn.name,
               fa.numRegs, launches, outer, lanes, lanes_lane, lanes_tid);
        for (size_t i = 0; i < first.size(); i += LOGW)
            printf("      block %u thread %u: %u wrong; first in trip %u: read %u, v_mul_lo_u32 should have given %u; HW_ID %08x\n", first[i], first[i + 1], first[i + 2],
                   first[i + 4], first[i + 3], first[i + 5], first[i + 6]);
        fflush(stdout);
    }
    return 0;
}
