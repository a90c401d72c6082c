// Probe for the lost-tile finding (profiles/r03_pair_store.md): does the TOP register of a wave's VGPR allocation read back what
// was written to it?  The library kernel that lost records used exactly 32 VGPRs with its shift amount in v31, and the wrong
// records are the right tile shifted by the lane's THREAD ID -- what a read of an out-of-range VGPR returns (v0).
//
// Each kernel below names one VGPR explicitly (the top one of its allocation, or one below the top as a control), writes a
// sentinel to it, reads it back and reports every mismatch with the value read, the thread id and where the wave ran (HW_ID).
//   hipcc --offload-arch=gfx950 -O3 -o probe_top_vgpr tools/probe_top_vgpr.hip && ./probe_top_vgpr [launches] [iters]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr u32 MAXLOG = 4096, LOGW = 8;

// REG: the register under test; TOP: a register named only to set the size of the allocation (TOP >= REG)
#define CANARY(NAME, REG, TOP)                                                                                                     \
    __global__ __launch_bounds__(256) void NAME(u32* log, u32* count, u32 iters, u32 sleep) {                                     \
        __shared__ u32 pad[2310];                                  /* 9240 B of LDS like the library kernel */                     \
        const u32 tid = threadIdx.x;                                                                                               \
        pad[tid] = tid;                                                                                                            \
        __syncthreads();                                                                                                           \
        asm volatile("v_mov_b32 " TOP ", 0" ::: TOP);                                                                              \
        u32 wrong = 0, got_first = 0, it_first = 0;                                                                                \
        for (u32 it = 0; it < iters; ++it) {                                                                                       \
            const u32 sentinel = 0x5A000000u + it * 4099u + pad[(tid + it) & 255];                                                 \
            u32 got;                                                                                                               \
            if (sleep) asm volatile("v_mov_b32 " REG ", %1\n\ts_sleep 4\n\tv_mov_b32 %0, " REG : "=v"(got) : "v"(sentinel) : REG); \
            else asm volatile("v_mov_b32 " REG ", %1\n\ts_nop 7\n\tv_mov_b32 %0, " REG : "=v"(got) : "v"(sentinel) : REG);         \
            if (got != sentinel) { if (!wrong) { got_first = got; it_first = it; } ++wrong; }                                      \
        }                                                                                                                          \
        if (wrong) {                                                                                                               \
            const u32 i = atomicAdd(count, 1u);                                                                                    \
            if (i < MAXLOG) {                                                                                                      \
                u32* l = log + i * LOGW;                                                                                           \
                l[0] = blockIdx.x; l[1] = tid; l[2] = wrong; l[3] = got_first; l[4] = it_first;                                    \
                l[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      /* HW_ID */                                                 \
                l[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     /* XCC_ID */                                                \
                l[7] = 0x5A000000u + it_first * 4099u + pad[(tid + it_first) & 255];                                               \
            }                                                                                                                      \
        }                                                                                                                          \
    }

CANARY(canary_v31_of_32, "v31", "v31")
CANARY(canary_v30_of_32, "v30", "v31")
CANARY(canary_v31_of_40, "v31", "v39")
CANARY(canary_v39_of_40, "v39", "v39")
CANARY(canary_v23_of_24, "v23", "v23")
CANARY(canary_v63_of_64, "v63", "v63")
CANARY(canary_v127_of_128, "v127", "v127")

typedef void (*kernel_t)(u32*, u32*, u32, u32);

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 50;
    const u32 iters = argc > 2 ? (u32)atoi(argv[2]) : 2000;
    u32 *log = nullptr, *count = nullptr;
    CHECK(hipMalloc(&log, MAXLOG * LOGW * 4));
    CHECK(hipMalloc(&count, 4));
    struct { const char* name; kernel_t k; } kernels[] = {
        {"v31 of a 32-register allocation (top)", canary_v31_of_32}, {"v30 of a 32-register allocation", canary_v30_of_32},
        {"v31 of a 40-register allocation", canary_v31_of_40},       {"v39 of a 40-register allocation (top)", canary_v39_of_40},
        {"v23 of a 24-register allocation (top)", canary_v23_of_24}, {"v63 of a 64-register allocation (top)", canary_v63_of_64},
        {"v127 of a 128-register allocation (top)", canary_v127_of_128}};
    for (u32 sleep = 0; sleep < 2; ++sleep)
        for (auto& kn : kernels) {
            hipFuncAttributes fa;
            CHECK(hipFuncGetAttributes(&fa, (const void*)kn.k));
            unsigned long long waves_wrong = 0, lanes_tid = 0, lanes_other = 0;
            std::vector<u32> first;
            for (int l = 0; l < launches; ++l) {
                CHECK(hipMemset(count, 0, 4));
                hipLaunchKernelGGL(kn.k, dim3(2048), dim3(256), 0, 0, log, count, iters, sleep);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
                u32 n = 0;
                CHECK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost));
                if (!n) continue;
                std::vector<u32> h(std::min(n, MAXLOG) * LOGW);
                CHECK(hipMemcpy(h.data(), log, h.size() * 4, hipMemcpyDeviceToHost));
                waves_wrong += n;
                for (u32 i = 0; i < std::min(n, MAXLOG); ++i) {
                    const u32* r = &h[i * LOGW];
                    if (r[3] == r[1]) ++lanes_tid; else ++lanes_other;
                    if (first.size() < 12 * LOGW) first.insert(first.end(), r, r + LOGW);
                }
            }
            printf("%-45s numRegs %3d  %s  launches %d x 2048 x 256 threads x %u round trips: lanes with a wrong read-back %llu (read == thread id: %llu, other: %llu)\n",
                   kn.name, fa.numRegs, sleep ? "s_sleep between write and read" : "s_nop between write and read  ", launches, iters, waves_wrong, lanes_tid, lanes_other);
            for (size_t i = 0; i < first.size(); i += LOGW)
                printf("      block %u thread %u: %u wrong, first at iteration %u: read %08x wrote %08x  HW_ID %08x (wave %u simd %u cu %u sh %u se %u) XCC %u\n", first[i], first[i + 1],
                       first[i + 2], first[i + 4], first[i + 3], first[i + 7], first[i + 5], first[i + 5] & 15, (first[i + 5] >> 4) & 3, (first[i + 5] >> 8) & 15,
                       (first[i + 5] >> 12) & 1, (first[i + 5] >> 13) & 7, first[i + 6] & 15);
            fflush(stdout);
        }
    return 0;
}
