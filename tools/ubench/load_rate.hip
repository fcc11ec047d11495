// Vector-memory return rate of one CU for the load shapes of K4 / K5: every wave reads 1 KiB rows (dwordx4 per lane),
// 512 B rows (dwordx2) or 256 B rows (dword) from a table of `rows` rows (L1-resident when small, L2-resident when
// large), `inflight` independent loads per iteration.  Prints bytes / clock / CU for a full-chip launch.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/load_rate.hip -o /tmp/load_rate && /tmp/load_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <typename T, int INF>
__global__ __launch_bounds__(256) void k(const T* __restrict__ tab, int rows, int iters, long long* out, float* sink) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    unsigned r = wave * 2654435761u;
    T acc[INF];
    for (int i = 0; i < INF; ++i) acc[i] = T{};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < INF; ++i) {
            r = r * 1664525u + 1013904223u;
            const int row = (r >> 8) % rows;                       // wave-uniform pseudo-random row
            const T v = tab[(long long)row * 64 + lane];
            if constexpr (sizeof(T) == 16) { acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w; }
            else if constexpr (sizeof(T) == 8) { acc[i].x += v.x; acc[i].y += v.y; }
            else acc[i] += v;
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < INF; ++i) {
        if constexpr (sizeof(T) == 16) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        else if constexpr (sizeof(T) == 8) s += acc[i].x + acc[i].y;
        else s += acc[i];
    }
    if (s == 1234.5f) sink[0] = s;
    if (threadIdx.x == 0) atomicMax((unsigned long long*)out, (unsigned long long)(t1 - t0));
}

template <typename T, int INF>
void run(const char* name, int rows, int wg_per_cu) {
    T* tab; hipMalloc(&tab, (size_t)rows * 64 * sizeof(T)); hipMemset(tab, 0, (size_t)rows * 64 * sizeof(T));
    long long* out; hipMalloc(&out, 8); float* sink; hipMalloc(&sink, 4);
    const int iters = 2000, grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(out, 0, 8);
        hipLaunchKernelGGL((k<T, INF>), dim3(grid), dim3(256), 0, 0, tab, rows, iters, out, sink);
        hipDeviceSynchronize();
    }
    long long cyc; hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost);
    const double bytes_per_cu = (double)wg_per_cu * 4 * iters * INF * 64 * sizeof(T);
    printf("%-8s rows %6d (%7.1f KB) %2d waves/CU, %2d in flight: %6.1f B/clk/CU\n", name, rows, rows * 64.0 * sizeof(T) / 1024,
           wg_per_cu * 4, INF, bytes_per_cu / (double)cyc);
    hipFree(tab); hipFree(out); hipFree(sink);
}

int main() {
    for (int rows : {16, 1500, 200000}) {
        for (int wg : {1, 2, 4}) {
            run<float4, 8>("dwordx4", rows, wg);
            run<float2, 8>("dwordx2", rows * 2, wg);
            run<float, 8>("dword", rows * 4, wg);
        }
    }
    return 0;
}
