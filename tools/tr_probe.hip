#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(unsigned short* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    int lane = threadIdx.x;
    int addr_elems;
    if (mode == 0) addr_elems = lane * 4;                                   // contiguous 8 B per lane
    else if (mode == 1) addr_elems = (lane & 15) * 64 + (lane >> 4) * 4;    // lane%16 -> row (stride 64 elems), lane/16 -> col block
    else addr_elems = (lane & 3) * 64 + ((lane >> 2) & 3) * 4 + (lane >> 4) * 16;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr_elems));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" %4d", h[l * 4 + e]); printf("%s", (l % 4 == 3) ? "\n" : "   "); }
    }
    return 0;
}
