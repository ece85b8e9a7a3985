// Development probe (round 5): what does the texture addresser charge a gather for -- active lanes, quads with an active lane, or
// distinct lines?  Decides whether kr_kb_tails gains from fetching each DISTINCT column of a 64-word group once (compact lanes) instead
// of once per word.   hipcc --offload-arch=gfx950 -O3 tools/probes/ta_probe.hip -o tools/probes/ta_probe.bin && ./ta_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(1024) k(const double *__restrict__ u, const uint32_t *__restrict__ w, int iters, int stride_cols, double *out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 1024 + threadIdx.x) >> 6;
    double acc = 0.0;
    uint32_t base = (uint32_t)wave * 977u;
    for (int it = 0; it < iters; it++) {
        base = (base + 17u * 64u) % (uint32_t)stride_cols;
        if (MODE == 0) {            // 64 lanes, runs of 4 lanes per column (sorted words, 3.8 entries per column)
            acc += u[base + (lane >> 2)];
        } else if (MODE == 1) {     // 16 compact lanes, one column each
            if (lane < 16) acc += u[base + lane];
        } else if (MODE == 2) {     // 16 scattered lanes (every 4th), one column each
            if ((lane & 3) == 0) acc += u[base + (lane >> 2)];
        } else if (MODE == 3) {     // 64 lanes, 64 consecutive columns
            acc += u[base + lane];
        } else if (MODE == 4) {     // 64 lanes, 64 random lines
            acc += u[(base + (uint32_t)lane * 4099u) % (uint32_t)stride_cols];
        } else if (MODE == 5) {     // coalesced 4-byte word load
            acc += (double)w[(size_t)base * 4 + lane];
        } else if (MODE == 6) {     // coalesced 16-byte word load (four words per lane)
            const uint4 v = ((const uint4 *)w)[(size_t)base + lane];
            acc += (double)(v.x ^ v.y ^ v.z ^ v.w);
        }
    }
    if (acc == 12345.678) out[0] = acc;
}
int main()
{
    const int cols = 383000;               // 3 MB of operands: one XCD's L2 holds them
    double *u, *out; uint32_t *w;
    hipMalloc(&u, sizeof(double) * (cols + 64 * 4100));
    hipMalloc(&w, sizeof(uint32_t) * ((size_t)cols * 4 + 4096) * 4);
    hipMalloc(&out, 8);
    hipMemset(u, 0, sizeof(double) * (cols + 64 * 4100));
    hipMemset(w, 0, sizeof(uint32_t) * ((size_t)cols * 4 + 4096) * 4);
    const int iters = 2000;
    const char *names[] = {"64 lanes, runs of 4 per column", "16 compact lanes", "16 scattered lanes", "64 lanes consecutive columns", "64 lanes random lines",
                           "coalesced b32 words", "coalesced b128 words"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(M) { k<M><<<256, 1024>>>(u, w, 10, cols, out); hipDeviceSynchronize(); hipEventRecord(e0); k<M><<<256, 1024>>>(u, w, iters, cols, out); hipEventRecord(e1); hipEventSynchronize(e1); \
    float ms; hipEventElapsedTime(&ms, e0, e1); const double inst = 256.0 * 16 * iters; \
    printf("%-34s %8.3f ms  %6.1f ns per wave-instruction per CU  (%.1f cycles at 2.4 GHz)\n", names[M], ms, ms * 1e6 / (16.0 * iters), ms * 1e6 / (16.0 * iters) * 2.4); (void)inst; }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    return 0;
}
