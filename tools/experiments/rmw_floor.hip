// Experiment (not product code): the memory floor of one S-round's access pattern without any of the
// round kernel's control flow: 4096 waves x 8 events; per event gather 2 random item rows (of 200,000)
// and the wave's user row, then write the 2 item rows back (in place).  No counters, no atomics, no
// staging, no arithmetic beyond one add.  How long does the chip need for that?
// build: hipcc --offload-arch=gfx950 -O3 -o rmw_floor rmw_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// quarter-wave shape: group g of 16 lanes owns event 4s+g, lane h holds floats [4h, 4h+4) and [64+4h, ..)
template <int WRITE_FRACTION_256>
__global__ void __launch_bounds__(256) k(float *Q, const float *P, unsigned n_items, unsigned first_user, unsigned salt) {
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, g = lane >> 4, h = lane & 15;
    f4 qi[2][2], qj[2][2], p[2][2];
    unsigned ri[2], rj[2];
    const unsigned user = first_user + wave / 6;                      // ~50 events per user, 8 per wave
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned e = wave * 8 + 4 * s + g;
        ri[s] = mix(e * 2 + salt) % n_items; rj[s] = mix(e * 2 + 1 + salt) % n_items;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            qi[s][v] = *(const f4 *)(Q + (size_t)ri[s] * 128 + 64 * v + 4 * h);
            qj[s][v] = *(const f4 *)(Q + (size_t)rj[s] * 128 + 64 * v + 4 * h);
            p[s][v] = *(const f4 *)(P + (size_t)user * 128 + 64 * v + 4 * h);
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const bool w = (mix(wave * 8 + 4 * s + g + 77u) & 255u) < (unsigned)WRITE_FRACTION_256;
            if (w) {
                *(f4 *)(Q + (size_t)ri[s] * 128 + 64 * v + 4 * h) = qi[s][v] + p[s][v];
                *(f4 *)(Q + (size_t)rj[s] * 128 + 64 * v + 4 * h) = qj[s][v] - p[s][v];
            }
        }
}

template <int WF>
void run(const char *name, float *Q, float *P, unsigned waves) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 200;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<WF>), dim3(waves / 4), dim3(256), 0, 0, Q, P, 200000u, (unsigned)w * 700u, (unsigned)w);
    hipEventRecord(a);
    for (int t = 0; t < reps; ++t) hipLaunchKernelGGL((k<WF>), dim3(waves / 4), dim3(256), 0, 0, Q, P, 200000u, (unsigned)(t * 683) % 900000u, (unsigned)t * 7919u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, ev = waves * 8.0;
    printf("%-44s waves=%u  %6.2f us/launch  %5.2f ns/event  algorithmic %.2f TB/s (3072 B/event)\n", name, waves, us, us * 1e3 / ev, ev * 3072 / us * 1e-6);
}

int main() {
    float *Q, *P;
    hipMalloc((void **)&Q, (size_t)200000 * 512); hipMemset(Q, 0, (size_t)200000 * 512);
    hipMalloc((void **)&P, (size_t)1000000 * 512); hipMemset(P, 0, (size_t)1000000 * 512);
    for (unsigned waves : {4096u, 8192u, 16384u}) {
        run<0>("gathers only", Q, P, waves);
        run<128>("gathers + half of the item rows written", Q, P, waves);
        run<256>("gathers + every item row written back", Q, P, waves);
    }
    return 0;
}
