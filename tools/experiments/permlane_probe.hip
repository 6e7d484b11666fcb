// Prints what v_permlane16_swap / v_permlane32_swap return on this GPU for (v, v) with v = lane id (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    const unsigned v = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane16_swap[0]", "permlane16_swap[1]", "permlane32_swap[0]", "permlane32_swap[1]"};
    for (int q = 0; q < 4; ++q) { printf("%s:", names[q]); for (int l = 0; l < 64; l += 4) printf(" %u", h[64 * q + l]); printf("\n"); }
    return 0;
}
