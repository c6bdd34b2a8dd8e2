// Diagnostic (never shipped): what does a barrier among the 32 workgroups of ONE XCD cost, compared with a kernel
// boundary (4.6 us in the decode graph) and with device-wide (memory-side) signalling?
// 256 workgroups x 256 threads, one per CU; each reads its XCC id, takes a rank inside its XCD, then runs ROUNDS of
//     store my slot (plain store -> L2) ; wait for the store ; arrive (atomic add on the XCD's round counter) ;
//     spin until the counter reaches the XCD's population ; read a neighbour's slot and check it is this round's.
// Variants: scope bits on the poll / data loads (sc0 = bypass L1 only, sc1 = agent scope) and on the atomic.
// Spins are bounded (SPIN_LIMIT): a wrong assumption ends in an error count, never in a hang.
// hipcc --offload-arch=gfx950 -O3 tools/probe_xcd_barrier.hip -o tools/probe_xcd_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NXCD = 8, MAXR = 64, SPIN_LIMIT = 100000;

struct Args {
    unsigned* pop;        // [NXCD] population (first pass) / rank counter
    unsigned* cnt;        // [NXCD][MAXR] arrival counters
    unsigned* slots;      // [NXCD][64][16] one 64-byte line per (xcd, rank)
    long long* cycles;    // [256] wall-clock ticks per workgroup for the whole loop
    unsigned* err;        // [4]: 0 timeouts, 1 stale reads, 2 max spins
    unsigned* xcc_of;     // [256]
    int rounds, expect, variant;
};

template <int V>
__device__ __forceinline__ unsigned ld_poll(const unsigned* p) {
    unsigned v;
    if (V == 0) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (V == 1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int V>
__device__ __forceinline__ void atomic_arrive(unsigned* p) {
    unsigned one = 1;
    if (V == 0) asm volatile("global_atomic_add %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(one) : "memory");
    else asm volatile("global_atomic_add %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(one) : "memory");
}
template <int V>
__device__ __forceinline__ void st_slot(unsigned* p, unsigned v) {
    if (V == 0) asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}

// LV: scope of loads (0 = sc0, 1 = sc1, 2 = sc0 sc1); AV: atomic (0 = plain, 1 = sc1); SV: store (0 plain, 1 sc1)
template <int LV, int AV, int SV>
__global__ __launch_bounds__(256) void barrier_kernel(Args a) {
    __shared__ unsigned s_rank, s_ok;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (threadIdx.x == 0) {
        s_rank = atomicAdd(a.pop + NXCD + xcc, 1u);       // rank inside the XCD (second half of pop[])
        a.xcc_of[blockIdx.x] = xcc;
    }
    __syncthreads();
    const unsigned rank = s_rank;
    unsigned* my = a.slots + ((size_t)xcc * 64 + rank) * 32;          // two 64-byte lines: one per round parity
    const unsigned* nb = a.slots + ((size_t)xcc * 64 + (rank + 1) % a.expect) * 32;
    const long long t0 = wall_clock64();
    unsigned maxspin = 0;
    for (int r = 0; r < a.rounds; ++r) {
        if (threadIdx.x < 16) st_slot<SV>(my + (r & 1) * 16 + threadIdx.x, (unsigned)(r + 1));   // this round's payload
        __syncthreads();
        if (threadIdx.x == 0) {
            atomic_arrive<AV>(a.cnt + xcc * MAXR + r);
            unsigned spins = 0, ok = 1;
            while (ld_poll<LV>(a.cnt + xcc * MAXR + r) < (unsigned)a.expect) {
                if (++spins > SPIN_LIMIT) { atomicAdd(a.err, 1u); ok = 0; break; }
            }
            maxspin = spins > maxspin ? spins : maxspin;
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) break;
        if (threadIdx.x < 16) {
            const unsigned v = ld_poll<LV>(nb + (r & 1) * 16 + threadIdx.x);
            if (v != (unsigned)(r + 1)) atomicAdd(a.err + 1, 1u);                // stale neighbour data
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        a.cycles[blockIdx.x] = t1 - t0;
        atomicMax(a.err + 2, maxspin);
    }
}

// Chip-wide barrier among all 256 workgroups, agent scope everywhere.  HIER = 0: one counter, 256 arrivals;
// HIER = 1: 8 group counters (blockIdx % 8), the last arriver of a group bumps the top counter, everyone polls the top.
template <int HIER>
__global__ __launch_bounds__(256) void grid_barrier_kernel(Args a) {
    __shared__ unsigned s_ok;
    const unsigned grp = blockIdx.x & 7;
    unsigned* my = a.slots + (size_t)blockIdx.x * 32;
    const unsigned* nb = a.slots + (size_t)((blockIdx.x + 37) % gridDim.x) * 32;     // a workgroup on another XCD
    const long long t0 = wall_clock64();
    unsigned maxspin = 0;
    for (int r = 0; r < a.rounds; ++r) {
        if (threadIdx.x < 16) st_slot<1>(my + (r & 1) * 16 + threadIdx.x, (unsigned)(r + 1));
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* top = a.cnt + NXCD * MAXR / 2 + r;          // second half of cnt[]: top-level counters
            if (HIER) {
                unsigned old;
                unsigned one = 1;
                asm volatile("global_atomic_add %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(a.cnt + grp * 32 + (r % 32)), "v"(one) : "memory");
                if ((old & 31u) == 31u) atomic_arrive<1>(top);  // last of the 32 of this group (counters only grow)
            } else {
                atomic_arrive<1>(top);
            }
            const unsigned target = HIER ? 8u : 256u;
            unsigned spins = 0, ok = 1;
            while (ld_poll<1>(top) < target) {
                if (++spins > SPIN_LIMIT) { atomicAdd(a.err, 1u); ok = 0; break; }
            }
            maxspin = spins > maxspin ? spins : maxspin;
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) break;
        if (threadIdx.x < 16) {
            const unsigned v = ld_poll<1>(nb + (r & 1) * 16 + threadIdx.x);
            if (v != (unsigned)(r + 1)) atomicAdd(a.err + 1, 1u);
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        a.cycles[blockIdx.x] = t1 - t0;
        atomicMax(a.err + 2, maxspin);
    }
}

__global__ void count_kernel(Args a) {           // population per XCD with this grid
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicAdd(a.pop + (xcc & 0xf), 1u);
}

int main() {
    Args a{};
    const int rounds = 32;
    CK(hipMalloc(&a.pop, 2 * NXCD * 4));
    CK(hipMalloc(&a.cnt, NXCD * MAXR * 4));
    CK(hipMalloc(&a.slots, 256 * 32 * 4 > NXCD * 64 * 32 * 4 ? 256 * 32 * 4 : NXCD * 64 * 32 * 4));
    CK(hipMalloc(&a.cycles, 256 * 8));
    CK(hipMalloc(&a.err, 16));
    CK(hipMalloc(&a.xcc_of, 256 * 4));
    a.rounds = rounds;
    CK(hipMemset(a.pop, 0, 2 * NXCD * 4));
    hipLaunchKernelGGL(count_kernel, dim3(256), dim3(256), 0, 0, a);
    CK(hipDeviceSynchronize());
    unsigned pop[2 * NXCD];
    CK(hipMemcpy(pop, a.pop, sizeof(pop), hipMemcpyDeviceToHost));
    printf("population per XCC (256 workgroups of 256 threads):");
    bool uniform = true;
    for (int i = 0; i < NXCD; ++i) { printf(" %u", pop[i]); uniform = uniform && pop[i] == 32; }
    printf("\n");
    if (!uniform) { printf("not 32 per XCC: barrier probe skipped\n"); return 0; }
    a.expect = 32;
    const char* names[] = {"loads sc0, atomic plain, store plain (XCD-local hope)", "loads sc1, atomic plain, store plain",
                           "loads sc1, atomic sc1, store sc1 (agent scope everywhere)", "loads sc0 sc1, atomic sc1, store sc1"};
    for (int v = 0; v < 4; ++v) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(a.pop, 0, 2 * NXCD * 4));
            CK(hipMemset(a.cnt, 0, NXCD * MAXR * 4));
            CK(hipMemset(a.slots, 0, NXCD * 64 * 32 * 4));
            CK(hipMemset(a.err, 0, 16));
            if (v == 0) hipLaunchKernelGGL((barrier_kernel<0, 0, 0>), dim3(256), dim3(256), 0, 0, a);
            if (v == 1) hipLaunchKernelGGL((barrier_kernel<1, 0, 0>), dim3(256), dim3(256), 0, 0, a);
            if (v == 2) hipLaunchKernelGGL((barrier_kernel<1, 1, 1>), dim3(256), dim3(256), 0, 0, a);
            if (v == 3) hipLaunchKernelGGL((barrier_kernel<2, 1, 1>), dim3(256), dim3(256), 0, 0, a);
            CK(hipDeviceSynchronize());
            long long cyc[256];
            unsigned err[4];
            CK(hipMemcpy(cyc, a.cycles, sizeof(cyc), hipMemcpyDeviceToHost));
            CK(hipMemcpy(err, a.err, sizeof(err), hipMemcpyDeviceToHost));
            long long mx = 0;
            for (int i = 0; i < 256; ++i) mx = cyc[i] > mx ? cyc[i] : mx;
            // wall_clock64 ticks at 100 MHz
            printf("%-62s rep %d: %6.2f us per round (slowest workgroup), timeouts %u, stale reads %u, max spins %u\n",
                   names[v], rep, (double)mx / 100.0 / rounds, err[0], err[1], err[2]);
        }
    }
    const char* gnames[] = {"chip-wide barrier, one counter (256 arrivals)", "chip-wide barrier, 8 group counters + top"};
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(a.cnt, 0, NXCD * MAXR * 4));
            CK(hipMemset(a.slots, 0, NXCD * 64 * 32 * 4));
            CK(hipMemset(a.err, 0, 16));
            if (v == 0) hipLaunchKernelGGL((grid_barrier_kernel<0>), dim3(256), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL((grid_barrier_kernel<1>), dim3(256), dim3(256), 0, 0, a);
            CK(hipDeviceSynchronize());
            long long cyc[256];
            unsigned err[4];
            CK(hipMemcpy(cyc, a.cycles, sizeof(cyc), hipMemcpyDeviceToHost));
            CK(hipMemcpy(err, a.err, sizeof(err), hipMemcpyDeviceToHost));
            long long mx = 0;
            for (int i = 0; i < 256; ++i) mx = cyc[i] > mx ? cyc[i] : mx;
            printf("%-62s rep %d: %6.2f us per round (slowest workgroup), timeouts %u, stale reads %u, max spins %u\n",
                   gnames[v], rep, (double)mx / 100.0 / rounds, err[0], err[1], err[2]);
        }
    }
    return 0;
}
