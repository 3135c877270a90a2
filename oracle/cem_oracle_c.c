/* cem_oracle_c.c — TEST INFRASTRUCTURE / CPU BASELINE: a plain C + OpenMP restatement (fp32) of one CEM-MPC plan, the same algorithm as
 * oracle/cem_oracle.py (which cites the reference line by line; "parity unpinned": the reference ships no fixtures and TensorFlow is not
 * importable — this file is pinned to cem_oracle.py by tests/test_oracle_c.py).  Only tests/ and bench.py's cpu_baseline leg may load it.
 *
 *   cem_mpc.py:35-68        the optimiser loop: sample, tile over particles, unfold, objective, top-k, best-so-far, moments, smoothing, stop
 *   transition_model.py:64-87, mlp_ensemble.py:18-34,122-132,189-193   scale, Dense+relu layers, Gaussian heads (softplus + 1e-4), members by row chunk
 *   mpc_policy.py:26-39 / safe_cem_mpc.py:76-96,110-120                objectives (the reward mask order differs), Beta safety filter
 *   safety_gym.py:110-119,140-176,188-192                              'goal' reward, cost, closest_distance
 *
 * Rows r = p*N + n use member r / (P*N/E).  Noise: explicit tensors (eps_act [I][N][H][A], eps_model [I][H][P*N][O], eps_out [A]) for the
 * pinning tests, or (null pointers) a per-thread xoshiro128+ / Box-Muller generator for timing, as the reference draws its noise inside the plan.
 * Build: gcc -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile; a fixed ISA level because the built file travels
 * to another host).  No -ffast-math. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t O, A, U, L, E, P, N, H, k, I;
    float smoothing, one_minus_smoothing, stddev_threshold, noise_stddev;
    int32_t variant;                  /* 0 cem, 1 safe */
    float posterior_mean_threashold;
    int32_t scale_features, sampling_propagation;
    /* scorer ('goal' task) */
    int32_t observe_goal_lidar, goal_lo, goal_hi;
    float lidar_max_dist, goal_thresh /* fl32(goal_size * 0.8) */, reward_distance, reward_goal, reward_clip;
    int32_t constrain_indicator, n_cost;
    int32_t cost_lo[4], cost_hi[4];
    float cost_size[4];
} cem_c_config;

#define RB 16                         /* rows per block: a member's weights are re-used from cache across the block */

static inline float softplus_tf(float x)
{   /* tf.math.softplus as Eigen evaluates it (cem_oracle.py softplus_tf) */
    const float thr = logf(1.1920929e-07f) + 2.0f;
    if (x > -thr) return x;
    const float ex = expf(x);
    if (x < thr) return ex;
    return log1pf(ex);
}

static inline float closest_distance(const float *o, int lo, int hi, float D)
{
    float m = INFINITY;
    for (int f = lo; f < hi; ++f) {
        float v = D - D * (1.0f - o[f]);
        v = v < 0.f ? 0.f : (v > D ? D : v);
        if (v < m) m = v;
    }
    return m;
}
static inline float goal_dist(const cem_c_config *c, const float *o)
{
    if (c->observe_goal_lidar) return closest_distance(o, c->goal_lo, c->goal_hi, c->lidar_max_dist);
    return o[c->goal_lo] > 0.f ? o[c->goal_lo] : 0.f;
}
static inline float cost_of(const cem_c_config *c, const float *o)
{
    float s = 0.f;
    for (int k = 0; k < c->n_cost; ++k) s = s + (closest_distance(o, c->cost_lo[k], c->cost_hi[k], c->lidar_max_dist) <= c->cost_size[k] ? 1.0f : 0.0f);
    return c->constrain_indicator ? (s > 0.f ? 1.0f : 0.0f) : s;
}

/* xoshiro128+ and Box-Muller: the timing mode's noise (statistical quality is irrelevant to the timing; deterministic per thread and seed) */
typedef struct { uint32_t s[4]; int have; float spare; } rng_t;
static inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
static inline uint32_t rng_next(rng_t *r)
{
    const uint32_t res = r->s[0] + r->s[3], t = r->s[1] << 9;
    r->s[2] ^= r->s[0]; r->s[3] ^= r->s[1]; r->s[1] ^= r->s[2]; r->s[0] ^= r->s[3]; r->s[2] ^= t; r->s[3] = rotl(r->s[3], 11);
    return res;
}
static inline float rng_normal(rng_t *r)
{
    if (r->have) { r->have = 0; return r->spare; }
    const float u1 = ((float)(rng_next(r) >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = ((float)(rng_next(r) >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1)), ang = 6.2831853f * u2;
    r->spare = rad * sinf(ang); r->have = 1;
    return rad * cosf(ang);
}
static void rng_seed(rng_t *r, uint64_t seed, uint64_t stream)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + stream * 0xBF58476D1CE4E5B9ull + 1;
    for (int i = 0; i < 4; ++i) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; r->s[i] = (uint32_t)z | 1u; }
    r->have = 0; r->spare = 0.f;
}

/* natural weight blob of one member (include/cem_mpc.h): W_0 [D][U], b_0 [U], W_l [U][U], b_l ... , W_mu [U][O], b_mu, W_var [U][O], b_var */
static size_t member_floats(const cem_c_config *c)
{
    const size_t D = c->O + c->A, U = c->U, O = c->O;
    return D * U + U + (size_t)(c->L - 1) * (U * U + U) + 2 * (U * O + O);
}

/* out[r][j] = bias[j] + sum_k in[r][k] W[k][j], k ascending for every element (register tile of 4 rows x 16 columns: the weights of a k are
 * loaded once for four rows and the partial sums never leave the registers; the order of additions per element is unchanged) */
#define TR 4
#define TJ 16
static void dense(const float *in, int ldi, int K, const float *W, const float *b, int J, float *out, int ldo, int rows)
{
    for (int r0 = 0; r0 < rows; r0 += TR) {
        const int nr = rows - r0 < TR ? rows - r0 : TR;
        for (int j0 = 0; j0 < J; j0 += TJ) {
            const int nj = J - j0 < TJ ? J - j0 : TJ;
            float acc[TR][TJ];
            for (int r = 0; r < TR; ++r)
                for (int j = 0; j < TJ; ++j) acc[r][j] = j < nj ? b[j0 + j] : 0.f;
            if (nr == TR && nj == TJ) {
                for (int k = 0; k < K; ++k) {
                    const float *w = W + (size_t)k * J + j0;
                    const float a0 = in[(size_t)(r0 + 0) * ldi + k], a1 = in[(size_t)(r0 + 1) * ldi + k], a2 = in[(size_t)(r0 + 2) * ldi + k], a3 = in[(size_t)(r0 + 3) * ldi + k];
#pragma omp simd
                    for (int j = 0; j < TJ; ++j) { const float wv = w[j]; acc[0][j] += a0 * wv; acc[1][j] += a1 * wv; acc[2][j] += a2 * wv; acc[3][j] += a3 * wv; }
                }
            } else {
                for (int k = 0; k < K; ++k) {
                    const float *w = W + (size_t)k * J + j0;
                    for (int r = 0; r < nr; ++r) { const float a = in[(size_t)(r0 + r) * ldi + k]; for (int j = 0; j < nj; ++j) acc[r][j] += a * w[j]; }
                }
            }
            for (int r = 0; r < nr; ++r)
                for (int j = 0; j < nj; ++j) out[(size_t)(r0 + r) * ldo + j0 + j] = acc[r][j];
        }
    }
}

/* One CEM iteration's rollout + per-row objective pieces: ret[B] (done-masked return), costs[H][B] (safe: masked cost per step) */
static void rollout(const cem_c_config *c, const float *blob, const float *imin, const float *idelta, const float *state, const float *actions /*[N][H][A]*/,
                    const float *eps_model_it /*[H][B][O] or null*/, uint64_t seed, int it, float *ret, uint8_t *costs)
{
    const int O = c->O, A = c->A, U = c->U, L = c->L, N = c->N, P = c->P, H = c->H, D = O + A;
    const long B = (long)P * N, chunk = B / c->E;
    const size_t mf = member_floats(c);
    const long n_blocks = (B + RB - 1) / RB;
#pragma omp parallel
    {
        float *s = (float *)malloc(sizeof(float) * RB * O), *x = (float *)malloc(sizeof(float) * RB * D);
        float *h0 = (float *)malloc(sizeof(float) * RB * U), *h1 = (float *)malloc(sizeof(float) * RB * U);
        float *mu = (float *)malloc(sizeof(float) * RB * O), *var = (float *)malloc(sizeof(float) * RB * O);
        rng_t rng;
#pragma omp for schedule(dynamic, 4)
        for (long blk = 0; blk < n_blocks; ++blk) {
            /* a block never straddles a member boundary: cut it at the next multiple of `chunk` */
            const long r0 = blk * RB;
            long r1 = r0 + RB < B ? r0 + RB : B;
            rng_seed(&rng, seed, (uint64_t)it * 1000003ull + (uint64_t)blk);
            for (long ra = r0; ra < r1;) {
                const long member = ra / chunk;
                long rb = (member + 1) * chunk < r1 ? (member + 1) * chunk : r1;
                const int rows = (int)(rb - ra);
                const float *Wm = blob + (size_t)member * mf;
                float cum[RB], dprev[RB]; int done[RB];
                for (int r = 0; r < rows; ++r) { memcpy(s + (size_t)r * O, state, sizeof(float) * O); cum[r] = 0.f; done[r] = 0; dprev[r] = goal_dist(c, state); }
                for (int t = 0; t < H; ++t) {
                    for (int r = 0; r < rows; ++r) {
                        const long n = (ra + r) % N;
                        float *xr = x + (size_t)r * D;
                        for (int f = 0; f < O; ++f) xr[f] = s[(size_t)r * O + f];
                        for (int a = 0; a < A; ++a) xr[O + a] = actions[((size_t)n * H + t) * A + a];
                        if (c->scale_features) for (int f = 0; f < D; ++f) xr[f] = (xr[f] - imin[f]) / idelta[f];
                    }
                    const float *w = Wm; float *hin = x; int K = D, ldi = D; float *hout = h0;
                    for (int l = 0; l < L; ++l) {
                        dense(hin, ldi, K, w, w + (size_t)K * U, U, hout, U, rows);
                        for (int i = 0; i < rows * U; ++i) hout[i] = hout[i] > 0.f ? hout[i] : 0.f;
                        w += (size_t)K * U + U; hin = hout; hout = (hout == h0) ? h1 : h0; K = U; ldi = U;
                    }
                    dense(hin, U, U, w, w + (size_t)U * O, O, mu, O, rows); w += (size_t)U * O + O;
                    dense(hin, U, U, w, w + (size_t)U * O, O, var, O, rows);
                    for (int r = 0; r < rows; ++r) {
                        float *sr = s + (size_t)r * O;
                        /* cost of s_t (before the update), safety_gym.py:62-66 */
                        const float c_t = c->variant == 1 ? cost_of(c, sr) : 0.f;
                        for (int f = 0; f < O; ++f) {
                            const float sd = sqrtf(softplus_tf(var[(size_t)r * O + f]) + 1e-4f);
                            const float e = !c->sampling_propagation ? 0.f : (eps_model_it ? eps_model_it[((size_t)t * B + (ra + r)) * O + f] : rng_normal(&rng));
                            sr[f] = sr[f] + (mu[(size_t)r * O + f] + sd * e);
                        }
                        const float dn = goal_dist(c, sr);
                        const int ga = dprev[r] <= c->goal_thresh;
                        float rew = (dprev[r] - dn) * c->reward_distance + (ga ? 1.0f : 0.0f) * c->reward_goal;
                        if (c->reward_clip > 0.f) rew = rew < -c->reward_clip ? -c->reward_clip : (rew > c->reward_clip ? c->reward_clip : rew);
                        if (c->variant == 1) {                              /* safe_cem_mpc.py:86-93: done OR-ed first */
                            done[r] = done[r] || ga;
                            const float nd = done[r] ? 0.0f : 1.0f;
                            costs[(size_t)t * B + (ra + r)] = (uint8_t)(c_t * nd);
                            cum[r] = cum[r] + rew * nd;
                        } else {                                            /* mpc_policy.py:34-37 */
                            cum[r] = cum[r] + rew * (done[r] ? 0.0f : 1.0f);
                            done[r] = done[r] || ga;
                        }
                        dprev[r] = dn;
                    }
                }
                for (int r = 0; r < rows; ++r) ret[ra + r] = cum[r];
                ra = rb;
            }
        }
        free(s); free(x); free(h0); free(h1); free(mu); free(var);
    }
}

typedef struct { float score; int32_t idx; } si_t;
static int cmp_desc(const void *a, const void *b)
{   /* larger score first; ties -> lower index first (tf.nn.top_k) */
    const si_t *x = (const si_t *)a, *y = (const si_t *)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static int cmp_idx(const void *a, const void *b) { const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return x < y ? -1 : (x > y ? 1 : 0); }

/* returns 0; scores_dbg (optional) receives [I][N] */
int cem_c_plan(const cem_c_config *c, const float *blob, const float *inputs_min, const float *inputs_max, const float *lb, const float *ub,
               const float *mu0, const float *sigma0, const float *state, const float *eps_act, const float *eps_model, const float *eps_out,
               uint64_t seed, float *action_out, float *best_score_out, int32_t *iters_out, float *scores_dbg)
{
    const int O = c->O, A = c->A, N = c->N, P = c->P, H = c->H, k = c->k, D = O + A, HA = H * A;
    const long B = (long)P * N;
    float *imin = (float *)malloc(sizeof(float) * D), *idelta = (float *)malloc(sizeof(float) * D);
    for (int f = 0; f < D; ++f) { float d = inputs_max[f] - inputs_min[f]; if (d < 1e-5f) d = 1.01f; imin[f] = inputs_min[f]; idelta[f] = d; }
    float *mu = (float *)malloc(sizeof(float) * HA), *sg = (float *)malloc(sizeof(float) * HA);
    for (int t = 0; t < H; ++t) for (int a = 0; a < A; ++a) { mu[t * A + a] = mu0[a]; sg[t * A + a] = sigma0[a]; }
    float *actions = (float *)malloc(sizeof(float) * (size_t)N * HA), *ret = (float *)malloc(sizeof(float) * B), *scores = (float *)malloc(sizeof(float) * N);
    uint8_t *costs = c->variant == 1 ? (uint8_t *)malloc((size_t)H * B) : NULL;
    si_t *order = (si_t *)malloc(sizeof(si_t) * N);
    int32_t *elite = (int32_t *)malloc(sizeof(int32_t) * k);
    float *best = (float *)calloc(A, sizeof(float)), best_score = -INFINITY;
    float *mean = (float *)malloc(sizeof(float) * HA), *var = (float *)malloc(sizeof(float) * HA);
    /* Beta prior of safe_cem_mpc.py:113-115 at mu 0.5, sigma 0.27 (fp32) */
    const float bmu = 0.5f, bsg = 0.27f;
    const float alpha = (((1.0f - bmu) / (bsg * bsg)) - 1.0f / bmu) * (bmu * bmu), beta = alpha * (1.0f / bmu - 1.0f);
    int iters = 0;
    for (int it = 0; it < c->I; ++it) {
        /* cem_mpc.py:44-48 */
#pragma omp parallel
        {
            rng_t rng;
#pragma omp for schedule(static)
            for (int n = 0; n < N; ++n) {
                rng_seed(&rng, seed ^ 0xA5A5A5A5ull, (uint64_t)it * 1000003ull + (uint64_t)n);
                for (int j = 0; j < HA; ++j) {
                    const float e = eps_act ? eps_act[((size_t)it * N + n) * HA + j] : rng_normal(&rng);
                    float v = e * sg[j] + mu[j];
                    const int a = j % A;
                    v = v < lb[a] ? lb[a] : (v > ub[a] ? ub[a] : v);
                    actions[(size_t)n * HA + j] = v;
                }
            }
        }
        rollout(c, blob, imin, idelta, state, actions, eps_model ? eps_model + (size_t)it * H * B * O : NULL, seed, it, ret, costs);
        /* particle mean, Beta safety filter */
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; ++n) {
            float sum = 0.f;
            for (int p = 0; p < P; ++p) sum = sum + ret[(size_t)p * N + n];
            float sc = sum / (float)P;
            if (c->variant == 1) {
                int unsafe = 0;
                for (int t = 0; t < H; ++t) {
                    float cnt = 0.f;
                    for (int p = 0; p < P; ++p) cnt = cnt + (float)costs[(size_t)t * B + (size_t)p * N + n];
                    const float post = (alpha + cnt) / (alpha + beta + (float)P);
                    if (!(post <= c->posterior_mean_threashold)) unsafe = 1;
                }
                sc = sc - (unsafe ? 1.0f : 0.0f) * 100.0f;
            }
            scores[n] = sc;
        }
        if (scores_dbg) memcpy(scores_dbg + (size_t)it * N, scores, sizeof(float) * N);
        /* cem_mpc.py:56-67 */
        for (int n = 0; n < N; ++n) { order[n].score = scores[n]; order[n].idx = n; }
        qsort(order, N, sizeof(si_t), cmp_desc);
        for (int e = 0; e < k; ++e) elite[e] = order[e].idx;
        qsort(elite, k, sizeof(int32_t), cmp_idx);
        int bj = elite[0];
        for (int e = 1; e < k; ++e) if (scores[elite[e]] > scores[bj]) bj = elite[e];      /* first maximum in ascending index order */
        if (scores[bj] > best_score) { best_score = scores[bj]; for (int a = 0; a < A; ++a) best[a] = actions[(size_t)bj * HA + a]; }
        for (int j = 0; j < HA; ++j) {
            float s1 = 0.f;
            for (int e = 0; e < k; ++e) s1 = s1 + actions[(size_t)elite[e] * HA + j];
            mean[j] = s1 / (float)k;
            float s2 = 0.f;
            for (int e = 0; e < k; ++e) { const float d = actions[(size_t)elite[e] * HA + j] - mean[j]; s2 = s2 + d * d; }
            var[j] = s2 / (float)k;
        }
        float ssum = 0.f;
        for (int j = 0; j < HA; ++j) {
            mu[j] = c->smoothing * mu[j] + c->one_minus_smoothing * mean[j];
            sg[j] = c->smoothing * sg[j] + c->one_minus_smoothing * sqrtf(var[j]);
            ssum = ssum + sg[j];
        }
        ++iters;
        if (ssum / (float)HA <= c->stddev_threshold) break;
    }
    for (int a = 0; a < A; ++a) action_out[a] = best[a] + (eps_out ? eps_out[a] : 0.f) * c->noise_stddev;
    *best_score_out = best_score; *iters_out = iters;
    free(imin); free(idelta); free(mu); free(sg); free(actions); free(ret); free(scores); free(costs); free(order); free(elite); free(best); free(mean); free(var);
    return 0;
}

void cem_c_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int cem_c_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
