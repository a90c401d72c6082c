// probe_vmem_lds_war.hip -- does an LDS load that returns into the ADDRESS registers of an earlier global store, with no
// s_waitcnt vmcnt between them, redirect or lose that store on gfx950?
//
// The instruction sequence hipcc 7.2 emitted for table.hip's expand_tiles_kernel (records with sequence numbers, pair read from
// LDS after the key and weight stores; build_variants/table_v1.hip):
//     global_store_dwordx2 v[8:9], v[6:7], off        ; key        (address v[8:9])
//     global_store_dword   v[6:7], v22, off           ; weight     (address v[6:7])
//     s_cbranch_vccnz ...                              ; not taken
//     ds_read_b128 v[6:9], v30 offset:4096            ; LDS data returns into both address pairs
// lost about one tile in 5000 (tools/check_pair_store.py: v1/v2 wrong, v3/v4 -- other destination registers -- and v5 -- the
// same registers behind `s_waitcnt vmcnt(0)` -- right).  Here the sequence stands alone, in inline assembly: every thread
// stores to its slot of `intended` and then loads, from LDS, the address of its slot of `trap` into the very register pair
// that held the store's address.  A store that leaves after the LDS data has landed goes to `trap` (a valid address, so
// nothing faults) and leaves poison in `intended`.  Variants: with / without a never-taken branch between store and LDS
// load; with `s_waitcnt vmcnt(0)` in between (control).  Many waves and a stride of one cache line per thread keep the
// memory pipeline backed up, as it is in the real kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_vmem_lds_war.hip -o tools/probe_vmem_lds_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
constexpr u64 POISON = 0xEEEEEEEEEEEEEEEEull;

template <int VARIANT>        // 0: store, LDS load   1: store, branch, LDS load   2: store, s_waitcnt vmcnt(0), LDS load
__global__ __launch_bounds__(256) void war_kernel(u64* intended, u64* trap, int iters, u64 stride_words) {
    __shared__ u64 lds[256];
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 threads = (u64)gridDim.x * 256;
    for (int it = 0; it < iters; ++it) {
        const u64 slot = ((u64)it * threads + gid) * stride_words;
        lds[threadIdx.x] = (u64)(trap + slot);              // where a late store would land instead
        __syncthreads();
        u64 addr = (u64)(intended + slot);
        const u64 val = slot + 1;
        const unsigned lds_addr = (unsigned)(threadIdx.x * 8);
        if (VARIANT == 0)
            asm volatile("global_store_dwordx2 %0, %1, off\n\tds_read_b64 %0, %2\n\ts_waitcnt lgkmcnt(0)"
                         : "+v"(addr) : "v"(val), "v"(lds_addr) : "memory");
        else if (VARIANT == 1)
            asm volatile("global_store_dwordx2 %0, %1, off\n\ts_cbranch_execz 1f\n1:\n\tds_read_b64 %0, %2\n\ts_waitcnt lgkmcnt(0)"
                         : "+v"(addr) : "v"(val), "v"(lds_addr) : "memory");
        else
            asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)\n\tds_read_b64 %0, %2\n\ts_waitcnt lgkmcnt(0)"
                         : "+v"(addr) : "v"(val), "v"(lds_addr) : "memory");
        if (addr != (u64)(trap + slot)) intended[0] = 0;    // (keeps `addr` live: the LDS value is what the asm must produce)
        __syncthreads();
    }
}

// The library's sequence in full, on fixed registers: two stores (A: 8 bytes, address v[42:43], data v[40:41]; B: 4 bytes, its
// address written by the VALU into A's data registers right after A was issued), a never-taken branch, then ONE ds_read_b128
// into v[40:43] -- both addresses at once --, lanes in groups of 6 reading the same LDS address (as the 6 records of a tile do)
// and the last wave of a workgroup only partly active.  The LDS supplies {address of the lane's B trap slot, address of its A
// trap slot}.  WAIT: the same with s_waitcnt vmcnt(0) in front of the LDS load.
template <bool WAIT>
__global__ __launch_bounds__(256) void war2_kernel(u64* a_out, unsigned* b_out, u64* a_trap, unsigned* b_trap, int iters, u64 stride_words, unsigned active) {
    __shared__ u64 lds[2 * 64];
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 threads = (u64)gridDim.x * 256;
    for (int it = 0; it < iters; ++it) {
        const u64 slot = ((u64)it * threads + gid) * stride_words;
        const u64 gslot = ((u64)it * threads + (u64)blockIdx.x * 256 + (threadIdx.x / 6) * 6) * stride_words;   // the group's first lane
        if (threadIdx.x % 6 == 0) { lds[2 * (threadIdx.x / 6)] = (u64)(b_trap + 2 * gslot); lds[2 * (threadIdx.x / 6) + 1] = (u64)(a_trap + gslot); }
        __syncthreads();
        if (threadIdx.x < active) {
            const u64 a_addr = (u64)(a_out + slot), b_addr = (u64)(b_out + 2 * slot), val = slot + 1;
            const unsigned w = (unsigned)(slot + 7), lds_addr = (threadIdx.x / 6) * 16;
            unsigned o0, o1, o2, o3;
            if (WAIT)
                asm volatile("v_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\tv_mov_b32 v40, %6\n\tv_mov_b32 v41, %7\n\t"
                             "global_store_dwordx2 v[42:43], v[40:41], off\n\t"
                             "v_mov_b32 v40, %8\n\tv_mov_b32 v41, %9\n\t"
                             "global_store_dword v[40:41], %10, off\n\t"
                             "s_cbranch_execz 1f\n1:\n\t"
                             "s_waitcnt vmcnt(0)\n\t"
                             "ds_read_b128 v[40:43], %11\n\ts_waitcnt lgkmcnt(0)\n\t"
                             "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43"
                             : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)
                             : "v"((unsigned)a_addr), "v"((unsigned)(a_addr >> 32)), "v"((unsigned)val), "v"((unsigned)(val >> 32)),
                               "v"((unsigned)b_addr), "v"((unsigned)(b_addr >> 32)), "v"(w), "v"(lds_addr)
                             : "v40", "v41", "v42", "v43", "memory");
            else
                asm volatile("v_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\tv_mov_b32 v40, %6\n\tv_mov_b32 v41, %7\n\t"
                             "global_store_dwordx2 v[42:43], v[40:41], off\n\t"
                             "v_mov_b32 v40, %8\n\tv_mov_b32 v41, %9\n\t"
                             "global_store_dword v[40:41], %10, off\n\t"
                             "s_cbranch_execz 1f\n1:\n\t"
                             "ds_read_b128 v[40:43], %11\n\ts_waitcnt lgkmcnt(0)\n\t"
                             "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43"
                             : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)
                             : "v"((unsigned)a_addr), "v"((unsigned)(a_addr >> 32)), "v"((unsigned)val), "v"((unsigned)(val >> 32)),
                               "v"((unsigned)b_addr), "v"((unsigned)(b_addr >> 32)), "v"(w), "v"(lds_addr)
                             : "v40", "v41", "v42", "v43", "memory");
            if (o0 == 1 && o1 == 2 && o2 == 3 && o3 == 4) a_out[0] = 0;        // (keeps the loaded values live)
        }
        __syncthreads();
    }
}
__global__ void count2_kernel(const u64* a_out, const unsigned* b_out, const u64* a_trap, const unsigned* b_trap, u64 slots, u64 stride_words,
                              unsigned active, u64* out) {
    u64 a_missing = 0, a_wrong = 0, b_missing = 0, b_wrong = 0, trapped = 0;
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += (u64)gridDim.x * blockDim.x) {
        const u64 i = s * stride_words;
        if ((s % 256) < active) {
            if (a_out[i] == POISON) ++a_missing; else if (a_out[i] != i + 1) ++a_wrong;
            if (b_out[2 * i] == (unsigned)POISON) ++b_missing; else if (b_out[2 * i] != (unsigned)(i + 7)) ++b_wrong;
        }
        if (a_trap[i] != POISON || b_trap[2 * i] != (unsigned)POISON) ++trapped;
    }
    if (a_missing) atomicAdd(out, a_missing);
    if (a_wrong) atomicAdd(out + 1, a_wrong);
    if (b_missing) atomicAdd(out + 2, b_missing);
    if (b_wrong) atomicAdd(out + 3, b_wrong);
    if (trapped) atomicAdd(out + 4, trapped);
}

__global__ void count_kernel(const u64* intended, const u64* trap, u64 slots, u64 stride_words, u64* out) {
    u64 missing = 0, trapped = 0, wrong = 0;
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += (u64)gridDim.x * blockDim.x) {
        const u64 a = intended[s * stride_words], b = trap[s * stride_words];
        if (a == POISON) ++missing; else if (a != s * stride_words + 1) ++wrong;
        if (b != POISON) ++trapped;
    }
    if (missing) atomicAdd(out, missing);
    if (trapped) atomicAdd(out + 1, trapped);
    if (wrong) atomicAdd(out + 2, wrong);
}

int main(int argc, char** argv) {
    const int blocks = 8192, iters = argc > 1 ? atoi(argv[1]) : 8, rounds = argc > 2 ? atoi(argv[2]) : 3;
    const u64 stride_words = 16;                                   // one 128-byte line per thread and iteration
    const u64 slots = (u64)blocks * 256 * iters, words = slots * stride_words;
    u64 *intended, *trap, *out;
    CK(hipMalloc(&intended, words * 8)); CK(hipMalloc(&trap, words * 8)); CK(hipMalloc(&out, 32));
    const char* names[3] = {"store, LDS load into the store's address registers", "store, never-taken branch, LDS load", "store, s_waitcnt vmcnt(0), LDS load (control)"};
    for (int r = 0; r < rounds; ++r)
        for (int v = 0; v < 3; ++v) {
            CK(hipMemset(intended, 0xEE, words * 8)); CK(hipMemset(trap, 0xEE, words * 8)); CK(hipMemset(out, 0, 32));
            if (v == 0) hipLaunchKernelGGL(war_kernel<0>, dim3(blocks), dim3(256), 0, 0, intended, trap, iters, stride_words);
            else if (v == 1) hipLaunchKernelGGL(war_kernel<1>, dim3(blocks), dim3(256), 0, 0, intended, trap, iters, stride_words);
            else hipLaunchKernelGGL(war_kernel<2>, dim3(blocks), dim3(256), 0, 0, intended, trap, iters, stride_words);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(count_kernel, dim3(2048), dim3(256), 0, 0, intended, trap, slots, stride_words, out);
            u64 h[4];
            CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
            printf("round %d variant %d (%s): %llu stores; missing from their slot %llu, landed at the LDS-supplied address %llu, wrong value %llu\n",
                   r, v, names[v], slots, h[0], h[1], h[2]);
        }
    // the library's sequence in full
    {
        u64 *a_out, *a_trap, *out2; unsigned *b_out, *b_trap;
        CK(hipMalloc(&a_out, words * 8)); CK(hipMalloc(&a_trap, words * 8)); CK(hipMalloc(&b_out, words * 8)); CK(hipMalloc(&b_trap, words * 8));
        CK(hipMalloc(&out2, 64));
        const unsigned active = 200;
        for (int r = 0; r < rounds; ++r)
            for (int v = 0; v < 2; ++v) {
                CK(hipMemset(a_out, 0xEE, words * 8)); CK(hipMemset(a_trap, 0xEE, words * 8)); CK(hipMemset(b_out, 0xEE, words * 8)); CK(hipMemset(b_trap, 0xEE, words * 8));
                CK(hipMemset(out2, 0, 64));
                if (v == 0) hipLaunchKernelGGL(war2_kernel<false>, dim3(blocks), dim3(256), 0, 0, a_out, b_out, a_trap, b_trap, iters, stride_words, active);
                else hipLaunchKernelGGL(war2_kernel<true>, dim3(blocks), dim3(256), 0, 0, a_out, b_out, a_trap, b_trap, iters, stride_words, active);
                CK(hipGetLastError());
                CK(hipDeviceSynchronize());
                hipLaunchKernelGGL(count2_kernel, dim3(2048), dim3(256), 0, 0, a_out, b_out, a_trap, b_trap, slots, stride_words, active, out2);
                u64 h[8];
                CK(hipMemcpy(h, out2, 64, hipMemcpyDeviceToHost));
                printf("round %d full sequence%s: key store missing %llu wrong %llu; weight store missing %llu wrong %llu; trap slots written %llu\n", r,
                       v ? " + s_waitcnt vmcnt(0)" : "", h[0], h[1], h[2], h[3], h[4]);
            }
    }
    return 0;
}
