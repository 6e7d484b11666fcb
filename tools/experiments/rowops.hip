// Experiment (not product code): cost of scattered 512-byte row operations, one wave per row op,
// lane l touching floats l and 64+l -- the access shape of k_round at k = 128.  Which flavour does the
// memory side serve fastest?  build: hipcc --offload-arch=gfx950 -O3 -o rowops rowops.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE, int PER>
__global__ void __launch_bounds__(256) k(float *buf, unsigned rows, unsigned salt, float *sink) {
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const unsigned r = mix(wave * PER + t + salt) % rows;
        float *p = buf + (size_t)r * 128 + lane;
        if (MODE == 0) { acc += p[0] + p[64]; }
        else if (MODE == 1) { acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(p + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else if (MODE == 2) { p[0] = (float)t; p[64] = (float)t; }
        else if (MODE == 3) { __hip_atomic_store(p, (float)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + 64, (float)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else if (MODE == 4) { unsafeAtomicAdd(p, 1.0f); unsafeAtomicAdd(p + 64, 1.0f); }
        else if (MODE == 5) { acc += atomicExch(p, 0.f) + atomicExch(p + 64, 0.f); }
        else if (MODE == 6) { acc += atomicAdd(p, 1.0f) + atomicAdd(p + 64, 1.0f); }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE, int PER>
void run(const char *name, float *buf, unsigned rows, float *sink, unsigned waves) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 40;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, PER>), dim3(waves / 4), dim3(256), 0, 0, buf, rows, (unsigned)w * 7919u, sink);
    hipEventRecord(a);
    for (int t = 0; t < reps; ++t) hipLaunchKernelGGL((k<MODE, PER>), dim3(waves / 4), dim3(256), 0, 0, buf, rows, (unsigned)(t + 9) * 104729u, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, ops = (double)waves * PER;
    printf("%-34s waves=%u per=%d  %7.2f us/launch  %6.2f G rows/s  %6.1f G 64B-req/s  %5.2f TB/s\n", name, waves, PER, us, ops / us * 1e-3, ops * 8 / us * 1e-3, ops * 512 / us * 1e-6);
}

int main() {
    const unsigned rows = 200000;
    float *buf, *sink;
    hipMalloc((void **)&buf, (size_t)rows * 512); hipMemset(buf, 0, (size_t)rows * 512);
    hipMalloc((void **)&sink, 4);
    for (unsigned waves : {4096u, 16384u}) {
        run<0, 16>("plain loads", buf, rows, sink, waves);
        run<1, 16>("sc1 loads", buf, rows, sink, waves);
        run<2, 16>("plain stores", buf, rows, sink, waves);
        run<3, 16>("sc1 stores", buf, rows, sink, waves);
        run<4, 16>("float atomic add, no return", buf, rows, sink, waves);
        run<5, 16>("atomic exchange, returning", buf, rows, sink, waves);
        run<6, 16>("float atomic add, returning", buf, rows, sink, waves);
    }
    return 0;
}
