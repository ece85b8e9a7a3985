// Stand-alone probe (GPU box): how fast can the ~26 M single-occurrence ("tail") operand gathers of a C3-sized KR SpMV go
// when the tails of a SUPER-TILE of rows are processed in column order?  Synthetic tails (uniform random columns over n =
// 2^20, rows uniform inside their super-tile) reproduce the access pattern of the inter-genome noise exactly.
//   A  row-major order (what the row kernels do): every gather its own 128-byte line
//   B  column order inside a super-tile of S rows, products accumulated per row in LDS (64-bit integer atomics, i.e. an
//      order-free fixed-point sum): gathers of neighbouring lanes share lines
// build: hipcc --offload-arch=gfx950 -O3 -o tails_probe tools/probes/tails_probe.hip ; run: ./tails_probe
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// key = super-tile << 40 | column << 16 | row inside the super-tile (S <= 65536)
__global__ void gen(int64_t m, int64_t n, int S, int64_t n_super, uint64_t *keys, uint16_t *cnt)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    uint64_t h = mix(k * 0x9E3779B97F4A7C15ull + 12345);
    uint64_t st = k / (m / n_super + 1);
    if (st >= (uint64_t)n_super) st = n_super - 1;
    uint64_t col = h % (uint64_t)n;
    uint64_t row = (h >> 32) % (uint64_t)S;
    keys[k] = (st << 40) | (col << 16) | row;
    cnt[k] = (uint16_t)(1 + (h >> 50) % 7);
}

// A: thread per entry in ROW order (key2 = super-tile, row, column): gather, sum per thread (kept alive by a rare store)
__global__ void __launch_bounds__(256) gather_rowmajor(int64_t m, const uint64_t *keys, const uint16_t *cnt, const double *u, double *sink)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t k0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k0 < m; k0 += 4 * stride) {
        uint64_t key[4];
        double g[4];
#pragma unroll
        for (int q = 0; q < 4; q++) key[q] = k0 + q * stride < m ? keys[k0 + q * stride] : 0;
#pragma unroll
        for (int q = 0; q < 4; q++) g[q] = u[(key[q] >> 16) & 0xFFFFFF];
#pragma unroll
        for (int q = 0; q < 4; q++) acc += g[q] * (double)(k0 + q * stride < m ? cnt[k0 + q * stride] : 0);
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}

// B: one workgroup per super-tile chunk; entries in column order; fixed-point products added to LDS rows.
// TWO: a second (32-bit) accumulator takes the part of every product below the 64-bit grid
template <int S, bool TWO, int NT>
__global__ void __launch_bounds__(NT) gather_colsorted(const int64_t *off, int chunks, const uint64_t *keys, const uint16_t *cnt,
                                                        const double *u, double scale, long long *ytail, int *ylo)
{
    __shared__ unsigned long long acc[S];
    __shared__ int lo[TWO ? S : 1];
    const int st = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    for (int i = threadIdx.x; i < S; i += NT) { acc[i] = 0; if (TWO) lo[i] = 0; }
    __syncthreads();
    const int64_t a = off[st], b = off[st + 1];
    const int64_t len = (b - a + chunks - 1) / chunks;
    const int64_t s = a + ch * len, e = s + len < b ? s + len : b;
    for (int64_t k0 = s + threadIdx.x; k0 < e; k0 += 4 * NT) {
        uint64_t key[4];
        double g[4];
#pragma unroll
        for (int q = 0; q < 4; q++) key[q] = k0 + q * NT < e ? keys[k0 + q * NT] : 0;
#pragma unroll
        for (int q = 0; q < 4; q++) g[q] = u[(key[q] >> 16) & 0xFFFFFF];
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (k0 + q * NT < e) {
                const double p = g[q] * (double)cnt[k0 + q * NT] * scale;
                const long long f = __double2ll_rn(p);
                atomicAdd(&acc[key[q] & (S - 1)], (unsigned long long)f);
                if (TWO) atomicAdd(&lo[key[q] & (S - 1)], (int)__double2int_rn((p - (double)f) * 1048576.0));
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S; i += NT) {
        if (acc[i]) atomicAdd((unsigned long long *)&ytail[(int64_t)st * S + i], acc[i]);
        if (TWO && lo[i]) atomicAdd(&ylo[(int64_t)st * S + i], lo[i]);
    }
}

template <int S>
void run_b(int64_t n, int64_t m, uint64_t *keys, uint64_t *keys2, uint16_t *cnt, uint16_t *cnt2, void *tmp, size_t tb, double *u,
           long long *ytail, int *ylo, hipEvent_t e0, hipEvent_t e1)
{
    const int64_t n_super = n / S;
    gen<<<(unsigned)((m + 255) / 256), 256>>>(m, n, S, n_super, keys, cnt);
    CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys, keys2, cnt, cnt2, (int)m, 0, 52));
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> hk(m);
    CK(hipMemcpy(hk.data(), keys2, m * 8, hipMemcpyDeviceToHost));
    std::vector<int64_t> hoff(n_super + 1, 0);
    { int64_t p = 0; for (int64_t st = 0; st <= n_super; st++) { while (p < m && (int64_t)(hk[p] >> 40) < st) p++; hoff[st] = p; } }
    int64_t *off; CK(hipMalloc(&off, (n_super + 1) * 8));
    CK(hipMemcpy(off, hoff.data(), (n_super + 1) * 8, hipMemcpyHostToDevice));
    auto timeit = [&](const char *name, auto launch) {
        for (int w = 0; w < 3; w++) launch();
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; r++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-64s %8.1f us per pass\n", name, ms / 10 * 1e3);
    };
    for (int chunks : {1, 2, 3, 4, 6, 8}) {
        char nm[128];
        snprintf(nm, sizeof nm, "B  S=%d one accumulator, 512 threads, %d chunks/tile", S, chunks);
        timeit(nm, [&] { gather_colsorted<S, false, 512><<<(unsigned)(n_super * chunks), 512>>>(off, chunks, keys2, cnt2, u, 1048576.0, ytail, ylo); });
        snprintf(nm, sizeof nm, "B  S=%d hi+lo accumulators, 512 threads, %d chunks/tile", S, chunks);
        timeit(nm, [&] { gather_colsorted<S, true, 512><<<(unsigned)(n_super * chunks), 512>>>(off, chunks, keys2, cnt2, u, 1048576.0, ytail, ylo); });
        snprintf(nm, sizeof nm, "B  S=%d hi+lo accumulators, 1024 threads, %d chunks/tile", S, chunks);
        timeit(nm, [&] { gather_colsorted<S, true, 1024><<<(unsigned)(n_super * chunks), 1024>>>(off, chunks, keys2, cnt2, u, 1048576.0, ytail, ylo); });
    }
    CK(hipFree(off));
}

int main(int argc, char **argv)
{
    const int64_t n = 1 << 20, m = 26000000;
    constexpr int S = 8192;
    const int64_t n_super = n / S;
    uint64_t *keys, *keys2;
    uint16_t *cnt, *cnt2;
    double *u, *sink;
    long long *ytail;
    int64_t *off;
    CK(hipMalloc(&keys, m * 8)); CK(hipMalloc(&keys2, m * 8)); CK(hipMalloc(&cnt, m * 2)); CK(hipMalloc(&cnt2, m * 2));
    CK(hipMalloc(&u, n * 8)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&ytail, n * 8)); CK(hipMalloc(&off, (n_super + 1) * 8));
    std::vector<double> hu(n);
    for (int64_t i = 0; i < n; i++) hu[i] = 1.0 + (double)(i % 97) * 0.01;
    CK(hipMemcpy(u, hu.data(), n * 8, hipMemcpyHostToDevice));
    gen<<<(unsigned)((m + 255) / 256), 256>>>(m, n, S, n_super, keys, cnt);
    // column order inside each super-tile = plain sort of the keys
    size_t tb = 0;
    CK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys, keys2, cnt, cnt2, (int)m, 0, 52));
    void *tmp; CK(hipMalloc(&tmp, tb));
    CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys, keys2, cnt, cnt2, (int)m, 0, 52));
    CK(hipDeviceSynchronize());
    // super-tile offsets (host: keys are sorted by super-tile)
    std::vector<uint64_t> hk(m);
    CK(hipMemcpy(hk.data(), keys2, m * 8, hipMemcpyDeviceToHost));
    std::vector<int64_t> hoff(n_super + 1, 0);
    { int64_t p = 0; for (int64_t st = 0; st <= n_super; st++) { while (p < m && (int64_t)(hk[p] >> 40) < st) p++; hoff[st] = p; } }
    CK(hipMemcpy(off, hoff.data(), (n_super + 1) * 8, hipMemcpyHostToDevice));
    // row order for variant A: key = st | row | col  -> build by swapping fields, sort
    std::vector<uint64_t> hr(m);
    for (int64_t k = 0; k < m; k++) { uint64_t st = hk[k] >> 40, col = (hk[k] >> 16) & 0xFFFFFF, row = hk[k] & 0xFFFF; hr[k] = (st << 40) | (row << 24) | col; }
    uint64_t *keysr, *keysr2; CK(hipMalloc(&keysr, m * 8)); CK(hipMalloc(&keysr2, m * 8));
    CK(hipMemcpy(keysr, hr.data(), m * 8, hipMemcpyHostToDevice));
    CK(hipcub::DeviceRadixSort::SortKeys(tmp, tb, keysr, keysr2, (int)m, 0, 48));
    // back to the (col << 16 | row) field layout so that both kernels decode the same way
    CK(hipMemcpy(hr.data(), keysr2, m * 8, hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < m; k++) { uint64_t st = hr[k] >> 40, row = (hr[k] >> 24) & 0xFFFF, col = hr[k] & 0xFFFFFF; hr[k] = (st << 40) | (col << 16) | row; }
    CK(hipMemcpy(keysr, hr.data(), m * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        for (int w = 0; w < 3; w++) launch();
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; r++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-46s %8.1f us per pass  (%.1f G gathers/s)\n", name, ms / 10 * 1e3, m / (ms / 10 * 1e-3) / 1e9);
    };
    timeit("A  row order, 26 M gathers", [&] { gather_rowmajor<<<256 * 16, 256>>>(m, keysr, cnt, u, sink); });
    timeit("A' column order, no accumulation", [&] { gather_rowmajor<<<256 * 16, 256>>>(m, keys2, cnt2, u, sink); });
    int *ylo; CK(hipMalloc(&ylo, n * 4));
    CK(hipFuncSetAttribute((const void *)gather_colsorted<8192, true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 0));
    run_b<8192>(n, m, keys, keys2, cnt, cnt2, tmp, tb, u, ytail, ylo, e0, e1);
    run_b<4096>(n, m, keys, keys2, cnt, cnt2, tmp, tb, u, ytail, ylo, e0, e1);
    run_b<2048>(n, m, keys, keys2, cnt, cnt2, tmp, tb, u, ytail, ylo, e0, e1);
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
