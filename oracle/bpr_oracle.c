/*
 * oracle/bpr_oracle.c -- CPU restatement of the reference's BPR hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / the timed CPU baseline.  The product (yue_amd/) never
 * imports it and fails loudly when its HIP library is missing.
 *
 * Pinned against golden vectors captured from the reference itself
 * (tools/make_goldens.py -> tests/golden/, checked by tests/test_oracle_golden.py).
 *
 * Reference lines restated (paths relative to 0411tony/Yue):
 *   recommender/cf/BPR.py:40-62      epoch loop, sampler, triplet update, loss
 *   tool/qmath.py:115-116            sigmoid = 1/(1+exp(-x)) in double
 *   base/IterativeRecommender.py:58-60    predict = Q.dot(P[u])
 *   base/IterativeRecommender.py:98-145   mask + seed + overwrite-scan selection
 *   base/IterativeRecommender.py:186-228  ranking_performance's variant (same scan)
 * Third-party arithmetic restated from its published algorithm:
 *   CPython 3.10 `random` (MT19937, init_by_array seeding, getrandbits/_randbelow, choice).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  Float
 * contraction must stay off: the reference rounds every multiply and add separately.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* MT19937 exactly as CPython's _randommodule.c uses it                       */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } mt_t;

static void mt_init_genrand(mt_t *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

static void mt_init_by_array(mt_t *s, const uint32_t *key, int klen) {
    mt_init_genrand(s, 19650218u);
    int i = 1, j = 0;
    int kk = 624 > klen ? 624 : klen;
    for (; kk; kk--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
        if (j >= klen) j = 0;
    }
    for (kk = 623; kk; kk--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
    }
    s->mt[0] = 0x80000000u;
    s->idx = 624;
}

/* random.seed(int): key = 32-bit little-endian words of abs(seed), at least one word */
static void mt_seed_python(mt_t *s, uint64_t seed) {
    uint32_t key[2] = { (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32) };
    mt_init_by_array(s, key, key[1] ? 2 : 1);
}

static uint32_t mt_u32(mt_t *s) {
    static const uint32_t mag01[2] = { 0u, 0x9908b0dfu };
    if (s->idx >= 624) {
        int kk; uint32_t y;
        for (kk = 0; kk < 624 - 397; kk++) {
            y = (s->mt[kk] & 0x80000000u) | (s->mt[kk + 1] & 0x7fffffffu);
            s->mt[kk] = s->mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < 623; kk++) {
            y = (s->mt[kk] & 0x80000000u) | (s->mt[kk + 1] & 0x7fffffffu);
            s->mt[kk] = s->mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        y = (s->mt[623] & 0x80000000u) | (s->mt[0] & 0x7fffffffu);
        s->mt[623] = s->mt[396] ^ (y >> 1) ^ mag01[y & 1u];
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* Random._randbelow_with_getrandbits(n), n < 2^32: k = n.bit_length(); r = getrandbits(k) until r < n */
static uint32_t mt_randbelow(mt_t *s, uint32_t n) {
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) k++;
    uint32_t r = mt_u32(s) >> (32 - k);
    while (r >= n) r = mt_u32(s) >> (32 - k);
    return r;
}

static int csr_contains(const int64_t *indptr, const int32_t *indices, int64_t row, int32_t x) {
    int64_t lo = indptr[row], hi = indptr[row + 1];
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        int32_t v = indices[mid];
        if (v < x) lo = mid + 1; else if (v > x) hi = mid; else return 1;
    }
    return 0;
}

/*
 * BPR.py:46-48 -- for every event (in userRecord order): j = choice(itemList), redraw
 * while j is in the user's listened set.  itemList is in id order, so the drawn index
 * is the item id.  Continues the MT stream over `epochs` passes, as the reference does.
 * draws_out (optional, size draws_cap) logs every raw draw; returns the draw count.
 */
int64_t orc_sample_python(uint64_t seed, int epochs, const int32_t *ev_u, int64_t E, int32_t n,
                          const int64_t *indptr, const int32_t *indices,
                          int32_t *j_out, int32_t *draws_out, int64_t draws_cap) {
    mt_t s; mt_seed_python(&s, seed);
    int64_t nd = 0;
    for (int ep = 0; ep < epochs; ep++)
        for (int64_t e = 0; e < E; e++) {
            int32_t j;
            for (;;) {
                j = (int32_t)mt_randbelow(&s, (uint32_t)n);
                if (draws_out && nd < draws_cap) draws_out[nd] = j;
                nd++;
                if (!csr_contains(indptr, indices, ev_u[e], j)) break;
            }
            j_out[(int64_t)ep * E + e] = j;
        }
    return nd;
}

/* ------------------------------------------------------------------------- */
/* Counter-based sampler (ours; shared definition with the HIP kernel)         */
/* ------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31; return z;
}
#define ORC_MAX_ATTEMPTS 64
/* draw for (seed, epoch, event e, attempt a) over n_range items starting at lo */
static inline int32_t ctr_draw(uint64_t seed, uint32_t epoch, uint64_t e, uint32_t a, int32_t lo, int32_t n_range) {
    uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ull * (e + 1));
    z = mix64(z ^ (0xD1B54A32D192ED03ull * (uint64_t)(epoch + 1) + 0x8CB92BA72F3D8DD7ull * (uint64_t)a));
    return lo + (int32_t)(((z >> 32) * (uint64_t)(uint32_t)n_range) >> 32);
}
/* j = first attempt not in the user's listened row; -1 when all ORC_MAX_ATTEMPTS are rejected */
void orc_sample_counter(uint64_t seed, uint32_t epoch, const int32_t *ev_u, int64_t e0, int64_t E, int32_t lo, int32_t n_range,
                        const int64_t *indptr, const int32_t *indices, int32_t *j_out) {
    for (int64_t e = 0; e < E; e++) {
        int32_t j = -1;
        for (uint32_t a = 0; a < ORC_MAX_ATTEMPTS; a++) {
            int32_t c = ctr_draw(seed, epoch, (uint64_t)(e0 + e), a, lo, n_range);
            if (!csr_contains(indptr, indices, ev_u[e], c)) { j = c; break; }
        }
        j_out[e] = j;
    }
}

/* ------------------------------------------------------------------------- */
/* Triplet arithmetic                                                          */
/* ------------------------------------------------------------------------- */
/*
 * Summation order of the k-length dot.  The reference calls BLAS sdot, whose order is
 * not pinned; we fix one: 64 strided partials (element 64r+l goes to
 * partial l, r ascending, product and sum rounded separately) followed by a butterfly
 * with partner l^1, l^2, l^4, l^8, l^16, l^32.  The HIP kernels use the same order, so
 * replay is comparable bit for bit; against the reference's BLAS result it agrees to
 * fp32 rounding.
 */
static float dot64(const float *a, const float *b, int k) {
    float part[64];
    for (int l = 0; l < 64; l++) {
        float acc = 0.0f;
        for (int e = l; e < k; e += 64) { float pr = a[e] * b[e]; acc = acc + pr; }
        part[l] = acc;
    }
    for (int off = 1; off <= 32; off <<= 1) {
        float nxt[64];
        for (int l = 0; l < 64; l++) nxt[l] = part[l] + part[l ^ off];
        memcpy(part, nxt, sizeof part);
    }
    return part[0];
}

typedef struct { float c, ru, ri; double s; } coef_t;

/* BPR.py:50 + qmath.py:115-116: s in double on the fp32 margin; coefficient rounded to fp32 once */
static coef_t coef(const float *p, const float *qi, const float *qj, int k, double lr, double regU, double regI) {
    coef_t r;
    float x = dot64(p, qi, k) - dot64(p, qj, k);
    r.s = 1.0 / (1.0 + exp(-(double)x));
    r.c = (float)(lr * (1.0 - r.s));
    r.ru = (float)(lr * regU);
    r.ri = (float)(lr * regI);
    return r;
}

/* BPR.py:51-57 on one element; every product and sum rounded to fp32 separately (F10) */
static inline void upd1(float p, float qi, float qj, coef_t cf, float *p2, float *qi2, float *qj2) {
    float d = qi - qj;
    float t = cf.c * d;
    float p1 = p + t;               /* :51 */
    float tq = cf.c * p1;
    float qi1 = qi + tq;            /* :52 (uses updated P[u]) */
    float qj1 = qj - tq;            /* :53 */
    float rp = cf.ru * p1;  *p2 = p1 - rp;     /* :55 */
    float ra = cf.ri * qi1; *qi2 = qi1 - ra;   /* :56 */
    float rb = cf.ri * qj1; *qj2 = qj1 - rb;   /* :57 */
}

/*
 * BPR.py:42-58 with an explicit triplet stream: strictly sequential, in place.
 * Returns sum of -log(s) (double, :58).  Triplets with j < 0 are skipped.
 */
double orc_bpr_sequential(float *P, float *Q, int k, const int32_t *u, const int32_t *i, const int32_t *j, int64_t T,
                          double lr, double regU, double regI) {
    double nll = 0.0;
    for (int64_t t = 0; t < T; t++) {
        if (j[t] < 0) continue;
        float *p = P + (int64_t)u[t] * k, *qi = Q + (int64_t)i[t] * k, *qj = Q + (int64_t)j[t] * k;
        coef_t cf = coef(p, qi, qj, k, lr, regU, regI);
        for (int e = 0; e < k; e++) upd1(p[e], qi[e], qj[e], cf, &p[e], &qi[e], &qj[e]);
        nll += -log(cf.s);
    }
    return nll;
}

/* Dependency depth of a triplet stream under the sequential semantics of BPR.py:42-58: a triplet depends on the latest
 * earlier triplet that touches P[u], Q[i] or Q[j]; the depth is the longest chain of dependent triplets -- the number of
 * steps no schedule of the exact loop can go below (measurement aid of bench.py / the tests, not part of the path).
 * row_max_out (may be NULL): the largest number of touches of one item row. */
int64_t orc_dependency_depth(const int32_t *u, const int32_t *i, const int32_t *j, int64_t T, int64_t m, int64_t n, int64_t *row_max_out) {
    int32_t *lp = (int32_t *)calloc((size_t)m, sizeof(int32_t)), *lq = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int32_t *cq = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int32_t depth = 0, row_max = 0;
    if (!lp || !lq || !cq) { free(lp); free(lq); free(cq); return -1; }
    for (int64_t t = 0; t < T; ++t) {
        if (j[t] < 0) continue;
        int32_t l = lp[u[t]];
        if (lq[i[t]] > l) l = lq[i[t]];
        if (lq[j[t]] > l) l = lq[j[t]];
        ++l;
        lp[u[t]] = lq[i[t]] = lq[j[t]] = l;
        if (l > depth) depth = l;
        if (++cq[i[t]] > row_max) row_max = cq[i[t]];
        if (++cq[j[t]] > row_max) row_max = cq[j[t]];
    }
    if (row_max_out) *row_max_out = row_max;
    free(lp); free(lq); free(cq);
    return depth;
}


/*
 * A TIMING MODEL of the exact path's dataflow launch (yue_amd/csrc/chain_kernels.hpp), not a checker: `workers` wave groups
 * claim the runs of equal consecutive users in stream order; a run costs `startup` before its first triplet; a triplet starts
 * when its predecessor in the run is done and both item rows have arrived -- `hop` after the previous touch of the row finished
 * when that touch belongs to another run, `hop_same` when it belongs to this run -- and takes `step`.  Returns the makespan in
 * the unit of the cost arguments (tools/chain_model.py: which of step / hop / window bounds an epoch of a given stream).
 */
double orc_dataflow_model(const int32_t *u, const int32_t *i, const int32_t *j, int64_t T, int64_t n, int workers,
                          double startup, double step, double hop, double hop_same, int64_t *hops_on_path_out) {
    double *rq = (double *)calloc((size_t)n, sizeof(double)), *heap = (double *)calloc((size_t)workers, sizeof(double));
    int32_t *hq = (int32_t *)calloc((size_t)n, sizeof(int32_t));      /* cross-run hops on the longest path into the row's last touch */
    int64_t *owner = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    double makespan = 0.0; int64_t hops_best = 0;
    if (!rq || !heap || !hq || !owner) { free(rq); free(heap); free(hq); free(owner); return -1.0; }
    for (int64_t r = 0; r < n; ++r) owner[r] = -1;
    int64_t t = 0, run = 0;
    while (t < T) {
        int64_t e = t; while (e < T && u[e] == u[t]) ++e;
        /* the worker that frees first takes the run (heap[0] = minimum) */
        double f = heap[0] + startup; int32_t h = 0;
        for (int64_t x = t; x < e; ++x) {
            if (j[x] < 0) continue;
            const int32_t rows[2] = { i[x], j[x] };
            for (int q = 0; q < 2; ++q) {
                const int32_t r = rows[q];
                if (owner[r] < 0) continue;
                const double ready = rq[r] + (owner[r] == run ? hop_same : hop);
                if (ready > f) { f = ready; h = hq[r] + (owner[r] == run ? 0 : 1); }
            }
            f += step;
            rq[i[x]] = rq[j[x]] = f; owner[i[x]] = owner[j[x]] = run; hq[i[x]] = hq[j[x]] = h;
        }
        if (f > makespan) { makespan = f; hops_best = h; }
        /* replace the heap's minimum by this worker's new free time, sift down */
        int k = 0; heap[0] = f;
        for (;;) {
            int a = 2 * k + 1, b = a + 1, s = k;
            if (a < workers && heap[a] < heap[s]) s = a;
            if (b < workers && heap[b] < heap[s]) s = b;
            if (s == k) break;
            double tmp = heap[s]; heap[s] = heap[k]; heap[k] = tmp; k = s;
        }
        t = e; ++run;
    }
    if (hops_on_path_out) *hops_on_path_out = hops_best;
    free(rq); free(heap); free(hq); free(owner);
    return makespan;
}

/*
 * Timing baseline only (bench.py cpu_baseline): the same loop run Hogwild-style by
 * `threads` threads, each over a contiguous slice of the triplet stream, racing on shared rows.
 * Not a checker: its result depends on the interleaving.
 */
#include <pthread.h>
typedef struct { float *P, *Q; int k; const int32_t *u, *i, *j; int64_t t0, t1; double lr, regU, regI, nll; } hog_t;
static void *hog_run(void *arg) {
    hog_t *h = (hog_t *)arg;
    h->nll = orc_bpr_sequential(h->P, h->Q, h->k, h->u + h->t0, h->i + h->t0, h->j + h->t0, h->t1 - h->t0, h->lr, h->regU, h->regI);
    return NULL;
}
double orc_bpr_hogwild(float *P, float *Q, int k, const int32_t *u, const int32_t *i, const int32_t *j, int64_t T,
                       double lr, double regU, double regI, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    hog_t job[256];
    for (int t = 0; t < threads; t++) {
        job[t] = (hog_t){ P, Q, k, u, i, j, T * t / threads, T * (t + 1) / threads, lr, regU, regI, 0.0 };
        pthread_create(&tid[t], NULL, hog_run, &job[t]);
    }
    double nll = 0.0;
    for (int t = 0; t < threads; t++) { pthread_join(tid[t], NULL); nll += job[t].nll; }
    return nll;
}

/*
 * Round semantics S-round (ours): events are cut into consecutive rounds
 * round_ptr[r]..round_ptr[r+1].  Inside a round every triplet's update is the reference
 * update (above) evaluated on the factors as they were when the round started; the
 * per-row differences (new - old) are summed and added once.  Rounds are applied in
 * order.  A round of one triplet reproduces the sequential loop up to x+(x''-x) rounding.
 */
double orc_bpr_rounds(float *P, float *Q, int64_t m, int64_t n, int k,
                      const int32_t *u, const int32_t *i, const int32_t *j,
                      const int64_t *round_ptr, int64_t n_rounds, double lr, double regU, double regI) {
    double nll = 0.0;
    float *dP = (float *)calloc((size_t)m * k, sizeof(float));
    float *dQ = (float *)calloc((size_t)n * k, sizeof(float));
    for (int64_t r = 0; r < n_rounds; r++) {
        for (int64_t t = round_ptr[r]; t < round_ptr[r + 1]; t++) {
            if (j[t] < 0) continue;
            const float *p = P + (int64_t)u[t] * k, *qi = Q + (int64_t)i[t] * k, *qj = Q + (int64_t)j[t] * k;
            coef_t cf = coef(p, qi, qj, k, lr, regU, regI);
            float *dp = dP + (int64_t)u[t] * k, *dqi = dQ + (int64_t)i[t] * k, *dqj = dQ + (int64_t)j[t] * k;
            for (int e = 0; e < k; e++) {
                float p2, qi2, qj2;
                upd1(p[e], qi[e], qj[e], cf, &p2, &qi2, &qj2);
                dp[e] += p2 - p[e]; dqi[e] += qi2 - qi[e]; dqj[e] += qj2 - qj[e];
            }
            nll += -log(cf.s);
        }
        for (int64_t t = round_ptr[r]; t < round_ptr[r + 1]; t++) {
            if (j[t] < 0) continue;
            int64_t rows[3] = { (int64_t)u[t], (int64_t)i[t], (int64_t)j[t] };
            float *X[3] = { P, Q, Q }, *D[3] = { dP, dQ, dQ };
            for (int w = 0; w < 3; w++) {
                float *x = X[w] + rows[w] * k, *d = D[w] + rows[w] * k;
                for (int e = 0; e < k; e++) { x[e] += d[e]; d[e] = 0.0f; }
            }
        }
    }
    free(dP); free(dQ);
    return nll;
}

/*
 * S-round with SEQUENTIAL USER ROWS (ours): as orc_bpr_rounds, but every triplet sees the
 * user row as its predecessors of the same user left it -- P[u] is updated in place, event by event, exactly as the
 * reference does (BPR.py:51, :55) -- while the ITEM rows keep round semantics: read as they were when the round started,
 * per-row differences summed and added once.  A round of one triplet is the sequential loop (up to x + (x' - x) rounding
 * of the two item rows).
 */
double orc_bpr_rounds_seq_user(float *P, float *Q, int64_t m, int64_t n, int k,
                               const int32_t *u, const int32_t *i, const int32_t *j,
                               const int64_t *round_ptr, int64_t n_rounds, double lr, double regU, double regI) {
    double nll = 0.0;
    float *dQ = (float *)calloc((size_t)n * k, sizeof(float));
    (void)m;
    for (int64_t r = 0; r < n_rounds; r++) {
        for (int64_t t = round_ptr[r]; t < round_ptr[r + 1]; t++) {
            if (j[t] < 0) continue;
            float *p = P + (int64_t)u[t] * k;
            const float *qi = Q + (int64_t)i[t] * k, *qj = Q + (int64_t)j[t] * k;
            coef_t cf = coef(p, qi, qj, k, lr, regU, regI);
            float *dqi = dQ + (int64_t)i[t] * k, *dqj = dQ + (int64_t)j[t] * k;
            for (int e = 0; e < k; e++) {
                float p2, qi2, qj2;
                upd1(p[e], qi[e], qj[e], cf, &p2, &qi2, &qj2);
                p[e] = p2; dqi[e] += qi2 - qi[e]; dqj[e] += qj2 - qj[e];
            }
            nll += -log(cf.s);
        }
        for (int64_t t = round_ptr[r]; t < round_ptr[r + 1]; t++) {
            if (j[t] < 0) continue;
            int64_t rows[2] = { (int64_t)i[t], (int64_t)j[t] };
            for (int w = 0; w < 2; w++) {
                float *x = Q + rows[w] * k, *d = dQ + rows[w] * k;
                for (int e = 0; e < k; e++) { x[e] += d[e]; d[e] = 0.0f; }
            }
        }
    }
    free(dQ);
    return nll;
}

/*
 * One round's summed differences WITHOUT applying them (factors untouched): what one rank
 * contributes for a user block when items are sharded over several GPUs (multi-GPU).
 * dP[m*k] and dQ[n*k] must be zero on entry.  Returns the round's sum of -log(s).
 */
double orc_bpr_round_deltas(const float *P, const float *Q, int k,
                            const int32_t *u, const int32_t *i, const int32_t *j, int64_t T,
                            double lr, double regU, double regI, float *dP, float *dQ) {
    double nll = 0.0;
    for (int64_t t = 0; t < T; t++) {
        if (j[t] < 0) continue;
        const float *p = P + (int64_t)u[t] * k, *qi = Q + (int64_t)i[t] * k, *qj = Q + (int64_t)j[t] * k;
        coef_t cf = coef(p, qi, qj, k, lr, regU, regI);
        float *dp = dP + (int64_t)u[t] * k, *dqi = dQ + (int64_t)i[t] * k, *dqj = dQ + (int64_t)j[t] * k;
        for (int e = 0; e < k; e++) {
            float p2, qi2, qj2;
            upd1(p[e], qi[e], qj[e], cf, &p2, &qi2, &qj2);
            dp[e] += p2 - p[e]; dqi[e] += qi2 - qi[e]; dqj[e] += qj2 - qj[e];
        }
        nll += -log(cf.s);
    }
    return nll;
}

/* BPR.py:59 -- sums of squares; products in fp32 as NumPy forms P*P, accumulated in double */
double orc_sumsq(const float *X, int64_t count) {
    double s = 0.0;
    for (int64_t t = 0; t < count; t++) { float sq = X[t] * X[t]; s += (double)sq; }
    return s;
}

/* ------------------------------------------------------------------------- */
/* Scoring + overwrite-scan selection                                          */
/* ------------------------------------------------------------------------- */
/*
 * predict (IterativeRecommender.py:58-60) = Q.dot(P[u]); BLAS sgemv order is not pinned,
 * we fix the k-ascending fused-multiply-add chain (what v_mfma_f32_32x32x2_f32 computes).
 */
static float score_chain(const float *p, const float *q, int k) {
    float acc = 0.0f;
    for (int e = 0; e < k; e++) acc = fmaf(p[e], q[e], acc);
    return acc;
}

void orc_scores(const float *P, const float *Q, int64_t n, int k, int32_t user, float *out) {
    const float *p = P + (int64_t)user * k;
    for (int64_t it = 0; it < n; it++) out[it] = score_chain(p, Q + it * k, k);
}

/*
 * IterativeRecommender.py:98-145.  Candidates = item ids 0..n-1 in ascending order minus the
 * user's masked items.  Seed a[] with the first N candidates, stable-sort descending (:107-116).
 * Then for EVERY candidate in id order (the first N included): if a[N-1] < s, p = first slot
 * with a[p] < s (binary search :127-139; ties go after equals) and slot p is OVERWRITTEN (:142-144).
 * Returns 0, or -1 for a user with fewer than N candidates (the reference raises IndexError :126).
 * mask rows are indexed by position in users[] (mask_indptr has nu+1 entries).
 */
int orc_topn_scan(const float *P, const float *Q, int64_t n, int k,
                  const int32_t *users, int64_t nu, int N,
                  const int64_t *mask_indptr, const int32_t *mask_indices,
                  int32_t *out_ids, float *out_scores) {
    float *sc = (float *)malloc((size_t)n * sizeof(float));
    int rc = 0;
    for (int64_t t = 0; t < nu; t++) {
        orc_scores(P, Q, n, k, users[t], sc);
        float *a = out_scores + t * N; int32_t *id = out_ids + t * N;
        int cnt = 0;
        for (int64_t it = 0; it < n && cnt < N; it++) {
            if (csr_contains(mask_indptr, mask_indices, t, (int32_t)it)) continue;
            /* stable insertion keeps earlier ids first among equal scores == list.sort(reverse=True) */
            int pos = cnt;
            while (pos > 0 && a[pos - 1] < sc[it]) { a[pos] = a[pos - 1]; id[pos] = id[pos - 1]; pos--; }
            a[pos] = sc[it]; id[pos] = (int32_t)it; cnt++;
        }
        if (cnt < N) { rc = -1; for (int q = cnt; q < N; q++) { a[q] = -INFINITY; id[q] = -1; } continue; }
        for (int64_t it = 0; it < n; it++) {
            if (csr_contains(mask_indptr, mask_indices, t, (int32_t)it)) continue;
            float s = sc[it];
            if (a[N - 1] < s) {
                int p = 0;
                while (a[p] >= s) p++;      /* first slot strictly below s (a[] stays sorted) */
                a[p] = s; id[p] = (int32_t)it;
            }
        }
    }
    free(sc);
    return rc;
}

/* A true top-N (ties -> lower id first) for the SURVEY 8(f) "next" row and sanity checks. */
void orc_topn_true(const float *P, const float *Q, int64_t n, int k, const int32_t *users, int64_t nu, int N,
                   const int64_t *mask_indptr, const int32_t *mask_indices, int32_t *out_ids, float *out_scores) {
    float *sc = (float *)malloc((size_t)n * sizeof(float));
    for (int64_t t = 0; t < nu; t++) {
        orc_scores(P, Q, n, k, users[t], sc);
        float *a = out_scores + t * N; int32_t *id = out_ids + t * N; int cnt = 0;
        for (int64_t it = 0; it < n; it++) {
            if (csr_contains(mask_indptr, mask_indices, t, (int32_t)it)) continue;
            if (cnt == N && !(a[N - 1] < sc[it])) continue;
            int pos = cnt < N ? cnt : N - 1;
            while (pos > 0 && a[pos - 1] < sc[it]) { a[pos] = a[pos - 1]; id[pos] = id[pos - 1]; pos--; }
            a[pos] = sc[it]; id[pos] = (int32_t)it; if (cnt < N) cnt++;
        }
        for (int q = cnt; q < N; q++) { a[q] = -INFINITY; id[q] = -1; }
    }
    free(sc);
}
