// Minimal reproducer for the lost-tile finding (profiles/r03_pair_store.md): a 64-bit shift (v_lshrrev_b64 / v_lshlrev_b64 /
// v_ashrrev_i64) whose shift AMOUNT sits in the last VGPR of the wave's allocation (v31 of 32, v39 of 40, ...: the next register is
// not allocated).  LLVM knows this as Shift64HighRegBug and works around it for gfx90a only (GCNHazardRecognizer::
// fixShift64HighRegBug swaps the amount into another register); for gfx950 hipcc 7.2 emits the shift as it is.  On MI355X the shift
// then sometimes uses v0 (the thread id; its low six bits are the lane id) as the amount -- the out-of-range substitute, as if the
// amount were fetched as the pair v[31:32].
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_shift64_top_vgpr tools/probe_shift64_top_vgpr.hip
//   ./tools/probe_shift64_top_vgpr [launches] [barriers per launch] [workgroups]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32;
typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr u32 MAXLOG = 4096, LOGW = 8;

// AMT: the register the amount is put into; TOP: named only to fix the size of the allocation; OP: the shift
template <int KIND> __device__ __forceinline__ u64 shifted(u64 x, u32 amt) {       // 0 lshr, 1 lshl, 2 ashr
    return KIND == 0 ? x >> amt : KIND == 1 ? x << amt : (u64)((long long)x >> amt);
}

#define PROBE(NAME, KIND, OP, AMT, TOP)                                                                                                  \
    __global__ __launch_bounds__(256) void NAME(u32* log, u32* count, u64* cursor, u32 outer) {                                     \
        __shared__ u64 lds[256];                                                                                                     \
        __shared__ u64 base;                                                                                                         \
        const u32 tid = threadIdx.x;                                                                                                 \
        asm volatile("v_mov_b32 " TOP ", 0" ::: TOP);                                                                                \
        u32 wrong = 0, as_v0 = 0, first_trip = 0, first_amt = 0;                                                                     \
        for (u32 it = 0; it < outer; ++it) {                                                                                         \
            lds[tid] = 0x9E3779B97F4A7C15ull * (tid + 1 + it);                                                                       \
            __syncthreads();                                                                                                         \
            if (tid == 0) base = atomicAdd(cursor, 256ull);          /* one wave reaches the barrier late */                         \
            __syncthreads();                                                                                                         \
            for (u32 trip = 0; trip < 2; ++trip) {                                                                                   \
                const u64 x = lds[(tid * 5 + trip) & 255] | 1ull << 63;                                                              \
                const u32 amt = 1 + ((tid * 7 + it + trip) % 10);         /* small amounts, as in the library kernel */              \
                u64 got;                                                                                                             \
                asm volatile("v_mov_b32 " AMT ", %2\n\ts_nop 1\n\t" OP " %0, " AMT ", %1" : "=v"(got) : "v"(x), "v"(amt) : AMT);     \
                const u64 want = shifted<KIND>(x, amt);                                                                              \
                if (got != want) {                                                                                                   \
                    const u32 a0 = tid & 63;                                                                                         \
                    const u64 v0 = shifted<KIND>(x, a0);                                                                             \
                    if (!wrong) { first_trip = it * 2 + trip; first_amt = amt; }                                                     \
                    ++wrong; if (got == v0) ++as_v0;                                                                                 \
                }                                                                                                                    \
            }                                                                                                                        \
            __syncthreads();                                                                                                         \
        }                                                                                                                            \
        if (wrong) {                                                                                                                 \
            const u32 i = atomicAdd(count, 1u);                                                                                      \
            if (i < MAXLOG) { u32* l = log + i * LOGW; l[0] = blockIdx.x; l[1] = tid; l[2] = wrong; l[3] = as_v0; l[4] = first_trip; l[5] = first_amt;  \
                              l[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); l[7] = (u32)base; }                                  \
        }                                                                                                                            \
    }

PROBE(lshr_v31_of_32, 0, "v_lshrrev_b64", "v31", "v31")
PROBE(lshl_v31_of_32, 1, "v_lshlrev_b64", "v31", "v31")
PROBE(ashr_v31_of_32, 2, "v_ashrrev_i64", "v31", "v31")
PROBE(lshr_v30_of_32, 0, "v_lshrrev_b64", "v30", "v31")
PROBE(lshr_v31_of_40, 0, "v_lshrrev_b64", "v31", "v39")
PROBE(lshr_v39_of_40, 0, "v_lshrrev_b64", "v39", "v39")
PROBE(lshr_v23_of_24, 0, "v_lshrrev_b64", "v23", "v23")
PROBE(lshr_v63_of_64, 0, "v_lshrrev_b64", "v63", "v63")

typedef void (*kernel_t)(u32*, u32*, u64*, u32);

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20;
    const u32 outer = argc > 2 ? (u32)atoi(argv[2]) : 200;
    const u32 blocks = argc > 3 ? (u32)atoi(argv[3]) : 2048;        // 2048 x 4 waves fill every wave slot of the chip (8 per SIMD)
    u32 *log = nullptr, *count = nullptr; u64* cursor = nullptr;
    CHECK(hipMalloc(&log, MAXLOG * LOGW * 4)); CHECK(hipMalloc(&count, 4)); CHECK(hipMalloc(&cursor, 8));
    struct { const char* name; kernel_t k; } kernels[] = {
        {"v_lshrrev_b64, amount in v31 = LAST of 32", lshr_v31_of_32}, {"v_lshlrev_b64, amount in v31 = LAST of 32", lshl_v31_of_32},
        {"v_ashrrev_i64, amount in v31 = LAST of 32", ashr_v31_of_32}, {"v_lshrrev_b64, amount in v30 of 32", lshr_v30_of_32},
        {"v_lshrrev_b64, amount in v31 of 40", lshr_v31_of_40},        {"v_lshrrev_b64, amount in v39 = LAST of 40", lshr_v39_of_40},
        {"v_lshrrev_b64, amount in v23 = LAST of 24", lshr_v23_of_24}, {"v_lshrrev_b64, amount in v63 = LAST of 64", lshr_v63_of_64}};
    for (auto& kn : kernels) {
        hipFuncAttributes fa;
        CHECK(hipFuncGetAttributes(&fa, (const void*)kn.k));
        unsigned long long lanes = 0, shifts = 0, as_v0 = 0;
        std::vector<u32> first;
        for (int l = 0; l < launches; ++l) {
            CHECK(hipMemset(count, 0, 4)); CHECK(hipMemset(cursor, 0, 8));
            hipLaunchKernelGGL(kn.k, dim3(blocks), dim3(256), 0, 0, log, count, cursor, outer);
            CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
            u32 n = 0;
            CHECK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost));
            if (!n) continue;
            std::vector<u32> h(std::min(n, MAXLOG) * LOGW);
            CHECK(hipMemcpy(h.data(), log, h.size() * 4, hipMemcpyDeviceToHost));
            lanes += n;
            for (u32 i = 0; i < std::min(n, MAXLOG); ++i) {
                shifts += h[i * LOGW + 2]; as_v0 += h[i * LOGW + 3];
                if (first.size() < 8 * LOGW) first.insert(first.end(), &h[i * LOGW], &h[i * LOGW] + LOGW);
            }
        }
        const double total = (double)launches * blocks * 256 * outer * 2;
        printf("%-44s numRegs %3d: %.3g shifts; wrong results %llu in %llu lanes (%llu of them = the value shifted by thread id & 63)\n", kn.name, fa.numRegs, total,
               shifts, lanes, as_v0);
        for (size_t i = 0; i < first.size(); i += LOGW)
            printf("      block %u thread %u: %u wrong (%u as if shifted by v0); first in trip %u (amount %u); HW_ID %08x\n", first[i], first[i + 1], first[i + 2], first[i + 3],
                   first[i + 4], first[i + 5], first[i + 6]);
        fflush(stdout);
    }
    return 0;
}
