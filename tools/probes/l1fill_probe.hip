// Development probe (round 6): what does a 4-byte gather that misses the L1 cost under each cache-policy combination (sc0 / sc1 / nt)?
// The Louvain scan's label gathers miss the L1 0.8 times per entry and every miss fills a 128-byte line; if a policy moves less
// than a line per miss, the scan's bound moves with it.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l1fill_probe.hip -o tools/probes/l1fill_probe.bin && tools/probes/l1fill_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__device__ __forceinline__ uint32_t ld(const uint32_t *p)
{
    uint32_t v;
    if (MODE == 0) asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 1) asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 4) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 5) asm volatile("global_load_dword %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 6) asm volatile("global_load_dword %0, %1, off sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 7) asm volatile("global_load_dword %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// U independent gathers in flight per lane and iteration (the scan holds four)
template <int MODE, int U>
__global__ void __launch_bounds__(256) k(const uint32_t *__restrict__ lab, uint32_t n, int iters, uint32_t *out)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t x = gid * 2654435761u + 12345u, acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t idx[U], v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { x = x * 1664525u + 1013904223u; idx[u] = (uint32_t)(((uint64_t)x * n) >> 32); }
        if (MODE < 8) {
            // issue all, then wait once
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t *p = lab + idx[u];
                if (MODE == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(v[u]) : "v"(p) : "memory");
                if (MODE == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v[u]) : "v"(p) : "memory");
                if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v[u]) : "v"(p) : "memory");
                if (MODE == 3) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[u]) : "v"(p) : "memory");
                if (MODE == 4) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v[u]) : "v"(p) : "memory");
                if (MODE == 5) asm volatile("global_load_dword %0, %1, off sc0 nt" : "=v"(v[u]) : "v"(p) : "memory");
                if (MODE == 6) asm volatile("global_load_dword %0, %1, off sc1 nt" : "=v"(v[u]) : "v"(p) : "memory");
                if (MODE == 7) asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "=v"(v[u]) : "v"(p) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; u++) { asm volatile("" : "+v"(v[u])); acc += v[u]; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const char *names[] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    uint32_t *lab, *out;
    const uint32_t sizes[] = {981184u, 8u << 20, 64u << 20};       // 3.9 MB of labels (an XCD's L2 holds them), 32 MB, 256 MB
    hipMalloc(&lab, sizeof(uint32_t) * (size_t)sizes[2]);
    hipMalloc(&out, 8);
    hipMemset(lab, 0, sizeof(uint32_t) * (size_t)sizes[2]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400, U = 4;
    for (int sz = 0; sz < 3; sz++) {
        printf("labels: %u x 4 bytes\n", sizes[sz]);
#define RUN(M) { k<M, U><<<256 * 8, 256>>>(lab, sizes[sz], 10, out); hipDeviceSynchronize(); hipEventRecord(e0); k<M, U><<<256 * 8, 256>>>(lab, sizes[sz], iters, out); hipEventRecord(e1); \
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); const double winst = 8.0 * 4 * iters * U; /* wave-instructions per CU */ \
    printf("  %-12s %8.3f ms  %6.1f cycles per 64-lane gather per CU  (%.1f G lanes/s chip-wide)\n", names[M], ms, ms * 1e-3 * 2.4e9 / winst, 256.0 * winst * 64 / (ms * 1e-3) / 1e9); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    }
    return 0;
}
