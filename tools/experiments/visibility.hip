// Experiment (not product code): can the LAST toucher of a contended row read differences that the
// other touchers staged with ordinary stores, inside one launch, across XCDs, when the staging buffer
// is reused launch after launch?  Producer: store a 512-byte row (flavour S), s_waitcnt vmcnt(0),
// returning atomicAdd on the group's counter.  The producer whose add returns c-1 is the consumer:
// optional acquire fence, load the c rows (flavour L), compare with what this launch must have written.
//   S: 0 plain, 1 write-through (__hip_atomic_store relaxed, agent scope -> sc1)
//   L: 0 plain, 1 plain after fence(acquire, agent), 2 agent-scope atomic loads (sc1), 3 returning atomics
// build: hipcc --offload-arch=gfx950 -O3 -o visibility visibility.hip ; run: ./visibility
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float expect(unsigned launch, unsigned row, unsigned e) {
    return (float)((launch * 2654435761u + row * 40503u + e * 97u) & 0xFFFFF);
}

template <int S, int L>
__global__ void __launch_bounds__(64) k(float *stage, unsigned *cnt, unsigned *err, unsigned launch, int c) {
    const unsigned wave = blockIdx.x, lane = threadIdx.x;
    // spread the producers of one group over the grid: group g has rows g + s * G
    const unsigned G = gridDim.x / c;
    const unsigned g = wave % G, row = wave;
    if (wave >= G * c) return;
    float *dst = stage + (size_t)row * 128;
    for (int r = 0; r < 2; ++r) {
        const float v = expect(launch, row, lane + 64 * r);
        if (S == 0) dst[lane + 64 * r] = v;
        else __hip_atomic_store(dst + lane + 64 * r, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0;
    if (lane == 0) old = atomicAdd(cnt + g, 1u);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)(c - 1)) return;
    if (lane == 0) cnt[g] = 0u;
    if (L == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    unsigned bad = 0;
    for (int s = 0; s < c; ++s) {
        const unsigned rr = g + s * G;
        float *src = stage + (size_t)rr * 128;
        for (int r = 0; r < 2; ++r) {
            float v;
            if (L == 0 || L == 1) v = src[lane + 64 * r];
            else if (L == 2) v = __hip_atomic_load(src + lane + 64 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = atomicAdd(src + lane + 64 * r, 0.0f);
            if (v != expect(launch, rr, lane + 64 * r)) ++bad;
        }
    }
    if (bad) atomicAdd(err, bad);
}

template <int S, int L>
void run(const char *name, float *stage, unsigned *cnt, unsigned *err, int c, int launches, unsigned blocks) {
    hipMemset(err, 0, 4);
    hipMemset(cnt, 0, 4 * blocks);
    for (int t = 1; t <= launches; ++t) hipLaunchKernelGGL((k<S, L>), dim3(blocks), dim3(64), 0, 0, stage, cnt, err, (unsigned)t, c);
    unsigned h = 0;
    hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
    printf("%-52s c=%d launches=%d  stale elements: %u of %llu\n", name, c, launches, h, (unsigned long long)launches * blocks * 128ull);
}

int main() {
    const unsigned blocks = 8192;
    float *stage; unsigned *cnt, *err;
    hipMalloc((void **)&stage, (size_t)blocks * 128 * 4);
    hipMalloc((void **)&cnt, 4 * blocks);
    hipMalloc((void **)&err, 4);
    for (int c : {2, 8}) {
        run<0, 0>("plain stores, plain loads", stage, cnt, err, c, 400, blocks);
        run<0, 1>("plain stores, acquire fence + plain loads", stage, cnt, err, c, 400, blocks);
        run<1, 0>("write-through stores, plain loads", stage, cnt, err, c, 400, blocks);
        run<1, 1>("write-through stores, acquire fence + plain loads", stage, cnt, err, c, 400, blocks);
        run<1, 2>("write-through stores, agent-scope (sc1) loads", stage, cnt, err, c, 400, blocks);
        run<0, 3>("plain stores, returning atomics as loads", stage, cnt, err, c, 400, blocks);
        run<1, 3>("write-through stores, returning atomics as loads", stage, cnt, err, c, 400, blocks);
    }
    return 0;
}
