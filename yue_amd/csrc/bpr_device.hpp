// Device-side building blocks shared by the BPR training kernels (train_kernels.hpp, round_kernels.hpp, chain_kernels.hpp):
// argument blocks, the counter-based sampler, the 64-lane butterfly sum, the reference triplet update on one element and
// the raw-buffer access macros.  No __global__ functions here: the header may be included by several translation units.
// BPR training kernels for gfx950 (CDNA4).  Wave = 64 lanes.
//
// Layout of one triplet (u, i, j) inside a wave: lane l holds elements 64*r + l of P[u], Q[i]
// and Q[j], so every load / store / atomic wave-instruction covers 256 contiguous bytes of one
// row -- the shape that runs at the full float-atomic and plain-store rates on MI355X.  The
// k-length dots are reduced with a 64-lane butterfly (DPP-fused adds), which is the summation
// order oracle/bpr_oracle.c:dot64 restates.
//
// Arithmetic follows recommender/cf/BPR.py:50-57 of the reference: margin in fp32, sigmoid in
// double on it (tool/qmath.py:115-116), coefficient rounded to fp32 once, every multiply and
// add rounded separately (the file is compiled with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace yue {

constexpr int kNllSlots = 1024;
constexpr int kMaxAttempts = 64;

struct TrainArgs {
    float *P, *Q, *dP, *dQ;
    float *stage;                // staged item-row differences of the running round: one k-float row per touch
#ifdef YUE_STAMPS
    unsigned long long *stamps;  // diagnostic build only (make stamps): 8 phase time stamps per update wave
#endif
    const int32_t *ev_u, *ev_i;
    int32_t *ev_j;
    const int64_t *indptr;
    const int32_t *indices;
    double *nll_slots;
    int64_t m, n;
    int k;
    float ru, ri;
    double lr;
    uint64_t seed;
    uint32_t epoch;
    int32_t neg_lo, neg_range;
};

// One round of the S-round schedule.
struct RoundArgs {
    int64_t e_begin, e_end;      // events updated by this launch
    int64_t n_begin, n_end;      // events of the NEXT round: negatives drawn and row touches counted here
    // Item-row touch counters, one 64-bit word per row: high half = touches in the round (fixed while
    // the round runs: decides in-place store vs atomic path), low half = touches not yet retired
    // (decremented by the last-arriver protocol).  A late wave must never see a decremented total.
    unsigned long long *cnt_cur;     // round [e_begin, e_end)   (filled by the previous launch)
    unsigned long long *cnt_next;    // round [n_begin, n_end)
    uint32_t *cntp_cur;          // user-row flushes (runs of equal users inside a wave's batch) in this round
    uint32_t *cntp_next;         // ... in the next round
    // Slot table, kStageMax words per item row: the staging slots (2*(event - first event of the round)
    // + 0 for the positive / 1 for the negative) of the row's first kStageMax touches, in ticket order.
    uint32_t *tab_cur, *tab_next;
    int staged;                  // 1: rows with 2..kStageMax touches go through the staging rows instead of float atomics
    int apply_p;                 // 1: user rows are finished in this launch; 0: dP is left for the all-reduce
};

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31; return z;
}

// Counter-based draw for (seed, epoch, event, attempt): same function as the oracle's ctr_draw.
__device__ __forceinline__ int32_t ctr_draw(uint64_t seed, uint32_t epoch, uint64_t e, uint32_t a, int32_t lo, int32_t n_range) {
    uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ull * (e + 1));
    z = mix64(z ^ (0xD1B54A32D192ED03ull * (uint64_t)(epoch + 1) + 0x8CB92BA72F3D8DD7ull * (uint64_t)a));
    return lo + (int32_t)(((z >> 32) * (uint64_t)(uint32_t)n_range) >> 32);
}

__device__ __forceinline__ bool csr_contains(const int64_t *indptr, const int32_t *indices, int64_t row, int32_t x) {
    int64_t lo = indptr[row], hi = indptr[row + 1];
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        int32_t v = indices[mid];
        if (v < x) lo = mid + 1; else if (v > x) hi = mid; else return true;
    }
    return false;
}

// BPR.py:46-48 with a counter-based stream: first attempt outside the user's listened row.
__device__ __forceinline__ int32_t sample_negative(const TrainArgs &a, int32_t u, int64_t e) {
    for (uint32_t t = 0; t < (uint32_t)kMaxAttempts; ++t) {
        int32_t c = ctr_draw(a.seed, a.epoch, (uint64_t)e, t, a.neg_lo, a.neg_range);
        if (!csr_contains(a.indptr, a.indices, u, c)) return c;
    }
    return -1;
}

__device__ __forceinline__ float rdlane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes in the canonical order of oracle/bpr_oracle.c:dot64 -- butterfly with
// partner lane^1, ^2, ^4, ^8, ^16, then (lanes 0-31) + (lanes 32-63).  The first four steps are
// DPP-fused adds (quad_perm, quad_perm, row_half_mirror, row_mirror: same operands as the xor
// partners because the value is already constant over the smaller groups), xor 16 is a v_permlane16_swap of the value with itself.
// Returns the total as a wave-uniform value.
__device__ __forceinline__ float wave_sum(float v) {
    v = v + dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v = v + dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v = v + dpp_mov<0x141>(v);      // row_half_mirror
    v = v + dpp_mov<0x140>(v);      // row_mirror
    // partner lane ^ 16: v_permlane16_swap (gfx950) of the value with itself leaves rows 0, 0, 2, 2 in one result and rows
    // 1, 1, 3, 3 in the other -- their sum is v + v(lane ^ 16) on every lane (the add commutes), without the LDS round
    // trip of a ds_swizzle (16 of them sat in the latency chain of every round-kernel wave)
    // (written as inline assembly: through __builtin_amdgcn_permlane16_swap this compiler adds the FIRST result to itself;
    // the s_nops cover the VALU-write -> lane-crossing-read wait states the compiler would otherwise count for us)
    float w = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v), "+v"(w));
    v = v + w;
    return rdlane(v, 0) + rdlane(v, 32);
}

// exp(y) in double for |y| <= 700, as a SHORT dependency chain: this value sits on the critical path of every triplet (the
// epoch's time is its longest chain of dependent triplets times the latency of one).  Cody-Waite reduction y = n ln2 + r,
// |r| <= ln2 / 2, the degree-13 Taylor polynomial of exp(r) by Estrin's scheme (4 levels of independent fused multiply-adds
// instead of 13 dependent ones), ldexp.  Error about 1 ulp -- like the libm value the reference's math.exp returns, it is the
// fp32 rounding of lr * (1 - s) that enters the factors (a 1-ulp difference in exp moves that rounding once in ~1e8 triplets).
__device__ __forceinline__ double chain_exp(double y) {
    const double n = __builtin_rint(y * 1.4426950408889634074);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, y);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    const double r2 = r * r;
    const double a0 = __builtin_fma(r, 1.0, 1.0), a1 = __builtin_fma(r, 1.0 / 6.0, 0.5), a2 = __builtin_fma(r, 1.0 / 120.0, 1.0 / 24.0),
                 a3 = __builtin_fma(r, 1.0 / 5040.0, 1.0 / 720.0), a4 = __builtin_fma(r, 1.0 / 362880.0, 1.0 / 40320.0),
                 a5 = __builtin_fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0), a6 = __builtin_fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
    const double r4 = r2 * r2;
    const double b0 = __builtin_fma(a1, r2, a0), b1 = __builtin_fma(a3, r2, a2), b2 = __builtin_fma(a5, r2, a4);
    const double r8 = r4 * r4;
    const double c0 = __builtin_fma(b1, r4, b0), c1 = __builtin_fma(a6, r4, b2);
    return __builtin_ldexp(__builtin_fma(c1, r8, c0), (int)n);
}
// 1 / d for a finite d >= 1: hardware reciprocal, two Newton steps, one residual correction (faithful; the division the
// reference performs is correctly rounded: the two agree except in rare last-bit cases, see chain_exp).
__device__ __forceinline__ double chain_rcp(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(e, y, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(e, y, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(e, y, y);
}

// sigmoid of the fp32 margin in double (qmath.py:115-116), as every exact kernel and the loss use it
__device__ __forceinline__ double chain_sigmoid(float x) {
    const double xd = (double)x;
    if (__builtin_fabs(xd) <= 700.0) return chain_rcp(1.0 + chain_exp(-xd));
    return 1.0 / (1.0 + exp(-xd));
}
// fp32(lr (1 - sigmoid(x))) = lr / (1 + e^x) in SINGLE precision -- the short form of the step's coefficient (option
// chain_fast): five instructions on the dependency chain (v_mul, v_exp_f32, v_add, v_rcp_f32, v_mul) instead of ~35
// double-precision ones.  Not bit-equal to the reference's value: v_exp_f32 and v_rcp_f32 are good to 1 ulp, the rounding of
// x log2 e adds |x| * 6e-8 relative -- a few ulp of c for the margins that occur, against north_star's 1e-5 on the factors
// (measured: tests/test_gpu_exact.py, tests/test_gpu_baseline_configs.py).
__device__ __forceinline__ float chain_coef_fast(float x, float lr) {
    // (an infinite e^x gives 0, a vanishing one gives lr, a NaN margin stays a NaN -- the reference aborts on a NaN loss)
    return lr * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f));
}

// Sum over the 64 lanes in NO particular order (option chain_fast only; wave_sum keeps the oracle's order): four DPP-fused
// adds give every lane its row's sum, row_bcast15 / row_bcast31 carry the row sums up, lane 63 holds the total.
__device__ __forceinline__ float wave_sum_any(float v) {
    v = v + dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v = v + dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v = v + dpp_mov<0x141>(v);      // row_half_mirror
    v = v + dpp_mov<0x140>(v);      // row_mirror
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));   // row_bcast15 -> rows 1, 3
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));   // row_bcast31 -> rows 2, 3
    return rdlane(v, 63);
}

// Reference triplet update on one element (BPR.py:51-57), every product and sum rounded separately.
struct Elem { float p2, qi2, qj2; };
__device__ __forceinline__ Elem bpr_elem(float p, float qi, float qj, float c, float ru, float ri) {
    const float d = qi - qj;
    const float td = c * d;
    const float p1 = p + td;            // :51
    const float tq = c * p1;
    const float qi1 = qi + tq;          // :52 (updated P[u])
    const float qj1 = qj - tq;          // :53
    const float rp = ru * p1;
    const float ra = ri * qi1;
    const float rb = ri * qj1;
    Elem o;
    o.p2 = p1 - rp;                     // :55
    o.qi2 = qi1 - ra;                   // :56
    o.qj2 = qj1 - rb;                   // :57
    return o;
}

// Raw buffer access: one 128-bit descriptor per matrix in SGPRs, the row's byte offset in an SGPR
// (soffset), the lane's element offset in one VGPR (voffset) -- no per-access address arithmetic in
// the vector ALU.  A lane whose element index is >= k gets a voffset beyond num_records: the
// hardware returns 0 for its loads and drops its stores and atomics.
#define YUE_BLOAD(rs, vo, so) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((rs), (vo), (so), 0))
#define YUE_BSTORE(val, rs, vo, so) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (val)), (rs), (vo), (so), 0)
#define YUE_BLOAD_SC1(rs, vo, so) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((rs), (vo), (so), 16))
#define YUE_BSTORE_SC1(val, rs, vo, so) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (val)), (rs), (vo), (so), 16)
#define YUE_BATOMIC(val, rs, vo, so) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((val), (rs), (vo), (so), 0)
#ifdef YUE_STAMPS
#define YUE_STAMP(ix, waits) do { asm volatile(waits ::: "memory"); if (a.stamps && lane == 0) a.stamps[(size_t)wave * 8 + (ix)] = wall_clock64(); } while (0)
#else
#define YUE_STAMP(ix, waits) do {} while (0)
#endif
constexpr unsigned kOobOffset = 0x80000000u;
constexpr unsigned long long kTouch = 0x100000001ull;      // +1 touch in both halves of a counter word
constexpr int kRsrcFlags = 0x00020000;
constexpr unsigned kStageMax = 4;                          // touches per row served by the staging rows

// TPW consecutive words of an event array, loaded as one block (4-byte alignment: scalar block loads need no more)
constexpr int kHeaderSlack = 16;           // words of slack the event and metadata arrays carry behind their last event
template <int TPW>
struct __attribute__((aligned(4))) HeaderBlock { int32_t w[TPW]; };

}  // namespace yue
