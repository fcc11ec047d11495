// ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4i16): what does lane i of a 16-lane group receive?  LDS holds its own
// half-index; lane i of group g supplies the address of row (4 g' + i / 4), columns 4 (i % 4) .. + 3 of a [row][16] block -- the
// "[4-key][16-col] block, 4 contiguous halves per lane" of cdna_hip_programming.md T10.  Prints (row, col) of the 4 values per lane.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/tr16_semantics.hip -o /tmp/tr16 && /tmp/tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
#define STRIDE 520          // halves per row (an arbitrary row stride)
__global__ void k(int* out) {
    __shared__ __attribute__((aligned(16))) _Float16 l[16 * STRIDE];
    for (int i = threadIdx.x; i < 16 * STRIDE; i += 64) l[i] = (_Float16)0.f;
    __syncthreads();
    short* ls = reinterpret_cast<short*>(l);
    for (int r = threadIdx.x; r < 16 * 16; r += 64) ls[(r / 16) * STRIDE + (r % 16)] = (short)((r / 16) * 100 + (r % 16));   // value = 100 row + col
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    const int row = 4 * g + (i >> 2), col = 4 * (i & 3);
    auto p = (__attribute__((address_space(3))) s4*)(ls + row * STRIDE + col);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    int* d; hipMalloc((void**)&d, 64 * 4 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; ++lane) {
        printf("lane %2d (supplied row %2d cols %2d..):", lane, 4 * (lane >> 4) + ((lane & 15) >> 2), 4 * (lane & 3));
        for (int j = 0; j < 4; ++j) printf("  (r%d,c%d)", h[lane * 4 + j] / 100, h[lane * 4 + j] % 100);
        printf("\n");
    }
    return 0;
}
