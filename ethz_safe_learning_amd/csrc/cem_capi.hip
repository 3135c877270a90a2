// cem_capi.hip — host side of the C ABI declared in include/cem_mpc.h.
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC (see csrc/build.sh).
#include "cem_device.h"
#include "cem_train.h"
#include "cem_train_tile.h"
#include "cem_rollout_split.h"
#include "cem_rollout_wide.h"
#include "../../include/cem_mpc.h"

#include <dlfcn.h>
#include <link.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

static thread_local int g_last_hip = 0;
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_last_hip = (int)e_; return CEM_ERR_HIP; } } while (0)

namespace {

struct Dims {
    int O, A, Din, U, L, E, P, N, H, k, I, W, R;
    int Nloc, n_off, Bloc, Btot;
    int KB_in, KB_obs, NFW, KF0;     // KF0 = 4*NFW: layer-0 groups per wave, zero padded so every stage is a multiple of 4
    int act_q0, act_nq;              // feature quads of the network input that hold action features: [act_q0, act_q0 + act_nq)
    bool wide;                       // units > 128 or an activation other than relu: the generic rollout kernel (cem_rollout_wide.h)
    bool split;                      // precision CEM_PRECISION_SPLIT_BF16X3: cem_rollout_split.h (weight stream in 6 KB chunk groups)
    int wave_groups[4]; uint32_t wave_off_f4[4]; uint32_t member_stride_f4;
    size_t nat_member_floats;
};

// Which select kernel an iteration gets (cem_mpc.h select_mode; used by validate() and enqueue_select() alike, so that what one accepts
// the other can run).  requested 0 = automatic: the one-workgroup kernel while the population is below 24 000 candidates AND that kernel
// can hold its elite list + two per-(step, action) arrays + the staged keys in dynamic LDS; otherwise the multi-workgroup forms, which
// have no such limit — fused into one launch (3) when all its workgroups can be resident at once, else the eight-launch chain (2).
// Returns 0 if the REQUESTED form cannot serve this shape (only an explicit request for the one-workgroup kernel can fail).
int resolve_select_mode(int requested, long long N, long long k, long long HA, size_t dyn_limit, bool can_fuse, bool *cache_out)
{
    const size_t base = (size_t)((k + 3) & ~3ll) * 4 + (size_t)2 * HA * 4;
    const bool one_wg_ok = k <= 24576 && base <= dyn_limit;
    const bool cache = one_wg_ok && base + (size_t)CEM_SEL_KWORDS(N) * 4 <= dyn_limit;
    if (cache_out) *cache_out = cache;
    int mode = requested;
    if (mode == 0) mode = (N >= 24000 || !cache) ? 3 : 1;
    if (mode == 1 && !one_wg_ok) return 0;
    if (mode == 3 && !can_fuse) mode = 2;
    return mode;
}

int validate(const cem_config_t *c)
{
    if (!c) return CEM_ERR_INVALID_ARG;
    if (c->abi_version != CEM_ABI_VERSION) return CEM_ERR_INVALID_ARG;
    if (c->obs_dim < 1 || c->act_dim < 1 || c->act_dim > CEM_MAX_ACT || c->n_layers < 1 || c->ensemble_size < 1 ||
        c->particles < 1 || c->n_samples < 1 || c->horizon < 1 || c->horizon > 65535 || c->iterations < 1 || c->iterations > 65535 ||
        c->n_elite < 1 || c->n_elite > c->n_samples || c->world_size < 1 || c->rank < 0 || c->rank >= c->world_size)
        return CEM_ERR_INVALID_ARG;
    if (c->units < 1) return CEM_ERR_INVALID_ARG;
    if (c->activation < CEM_ACT_RELU || c->activation > CEM_ACT_GELU) return CEM_ERR_INVALID_ARG;
    if (!(fabs((double)c->one_minus_smoothing - (1.0 - (double)c->smoothing)) <= 2e-7)) return CEM_ERR_INVALID_ARG;   // see cem_mpc.h
    if (c->units > CEM_WIDE_U) return CEM_ERR_UNSUPPORTED;  // <= 128: the fast kernel (narrower layers run zero-padded, exactly); 129..256: cem_rollout_wide.h
    if (c->obs_dim + c->act_dim > CEM_U) return CEM_ERR_UNSUPPORTED;
    if (c->n_samples % c->world_size != 0) return CEM_ERR_INVALID_ARG;
    if (((long long)c->particles * c->n_samples) % c->ensemble_size != 0) return CEM_ERR_SPLIT;
    // the ONE-workgroup select kernel keeps the elite list and two per-(step, action) arrays in dynamic LDS (140 KB available, at most
    // 24576 elites): only an explicit request for it can be refused — the automatic choice routes such shapes to the multi-workgroup forms
    if (c->select_mode >= 0 && c->select_mode <= 3 &&
        resolve_select_mode(c->select_mode, c->n_samples, c->n_elite, (long long)c->horizon * c->act_dim, 140 * 1024, true, nullptr) == 0)
        return CEM_ERR_UNSUPPORTED;
    if (c->scorer.n_cost_kinds < 0 || c->scorer.n_cost_kinds > CEM_MAX_COST_KINDS) return CEM_ERR_INVALID_ARG;
    {   // scorer slices: inside the observation and non-empty (an empty lidar slice would make closest_distance +inf and rewards NaN)
        const cem_scorer_t &s = c->scorer;
        if (s.goal_mode != 0 && s.goal_mode != 1) return CEM_ERR_INVALID_ARG;
        if (s.goal_lo < 0 || s.goal_lo >= c->obs_dim) return CEM_ERR_INVALID_ARG;
        if (s.goal_mode == 0 && (s.goal_hi <= s.goal_lo || s.goal_hi > c->obs_dim)) return CEM_ERR_INVALID_ARG;
        for (int k = 0; k < s.n_cost_kinds; ++k)
            if (s.cost_lo[k] < 0 || s.cost_hi[k] <= s.cost_lo[k] || s.cost_hi[k] > c->obs_dim) return CEM_ERR_INVALID_ARG;
        if (!(s.lidar_max_dist >= 0.f) || !(s.goal_reached_dist == s.goal_reached_dist)) return CEM_ERR_INVALID_ARG;
    }
    // flat int indices of the sample / rollout / reduce kernels: N*H*A and H*P*N/world must fit an int32
    if ((long long)c->n_samples * c->horizon * ((c->act_dim + 3) & ~3) > 0x7fffffffll) return CEM_ERR_UNSUPPORTED;
    if ((long long)c->particles * (c->n_samples / c->world_size) * c->horizon > 0x7fffffffll) return CEM_ERR_UNSUPPORTED;
    {   // the hot kernel addresses the padded action quads with 32-bit byte offsets (one buffer resource)
        const long long nq = (c->obs_dim + c->act_dim + 3) / 4 - c->obs_dim / 4;
        if ((long long)c->n_samples * c->horizon * nq * 16 > 0x7fffffffll) return CEM_ERR_UNSUPPORTED;
    }
    if (c->variant != CEM_VARIANT_CEM && c->variant != CEM_VARIANT_SAFE) return CEM_ERR_INVALID_ARG;
    if (c->chunks_per_tile < 0 || c->chunks_per_tile > 4) return CEM_ERR_INVALID_ARG;
    if (c->rollout_segments < 0 || c->rollout_segments > 64) return CEM_ERR_INVALID_ARG;
    if (c->select_mode < 0 || c->select_mode > 3) return CEM_ERR_INVALID_ARG;
    if (c->precision != CEM_PRECISION_FP32 && c->precision != CEM_PRECISION_SPLIT_BF16X3) return CEM_ERR_INVALID_ARG;
    if (c->precision == CEM_PRECISION_SPLIT_BF16X3 && (c->units > CEM_U || c->activation != CEM_ACT_RELU)) return CEM_ERR_UNSUPPORTED;
    if ((long long)c->particles * c->n_samples > (1ll << 30)) return CEM_ERR_UNSUPPORTED;
    return CEM_OK;
}

Dims make_dims(const cem_config_t *c)
{
    Dims d{};
    d.O = c->obs_dim; d.A = c->act_dim; d.Din = d.O + d.A; d.U = c->units; d.L = c->n_layers; d.E = c->ensemble_size;
    d.P = c->particles; d.N = c->n_samples; d.H = c->horizon; d.k = c->n_elite; d.I = c->iterations;
    d.W = c->world_size; d.R = c->rank;
    d.Nloc = d.N / d.W; d.n_off = d.R * d.Nloc; d.Bloc = d.P * d.Nloc; d.Btot = d.P * d.N;
    d.KB_in = (d.Din + 15) / 16; d.KB_obs = (d.O + 15) / 16; d.NFW = (d.KB_in + 3) / 4; d.KF0 = 4 * d.NFW;
    d.act_q0 = d.O / 4; d.act_nq = (d.Din + 3) / 4 - d.act_q0;
    d.wide = d.U > CEM_U || c->activation != CEM_ACT_RELU || std::getenv("CEM_FORCE_GENERIC_ROLLOUT") != nullptr;   // (the variable: a diagnostic, scripts/sweep_configs.py)   // the tuned kernels: units <= 128 and relu; everything else takes the generic rollout kernel
    d.split = !d.wide && c->precision == CEM_PRECISION_SPLIT_BF16X3;
    uint32_t off = 0;
    for (int w = 0; w < 4; ++w) {
        // fp32 stream: 2 KB groups, one per 16-feature input block; split stream: 6 KB groups, one per K = 32 chunk (two blocks)
        const int per_stage = d.split ? CEM_SPLIT_CHUNKS : CEM_NG;
        int g = (d.split ? cem_split_l0_chunks(d.NFW) : d.KF0) + per_stage * (d.L - 1);
        for (int i = 0; i < d.NFW; ++i) if (w + 4 * i < d.KB_obs) g += per_stage;
        d.wave_groups[w] = g; d.wave_off_f4[w] = off; off += (uint32_t)g * (d.split ? 384u : 128u);
    }
    d.member_stride_f4 = off + 256u;    // +2 groups of slack: the prefetch queue may run ahead of a short stream
    d.nat_member_floats = (size_t)d.Din * d.U + d.U + (size_t)(d.L - 1) * ((size_t)d.U * d.U + d.U) + 2 * ((size_t)d.U * d.O + d.O);
    return d;
}

// natural blob offsets of one member
struct NatOff { std::vector<size_t> W, b; size_t Wmu, bmu, Wvar, bvar; };
NatOff nat_offsets(const Dims &d)
{
    NatOff n; size_t o = 0; int fi = d.Din;
    for (int l = 0; l < d.L; ++l) { n.W.push_back(o); o += (size_t)fi * d.U; n.b.push_back(o); o += d.U; fi = d.U; }
    n.Wmu = o; o += (size_t)d.U * d.O; n.bmu = o; o += d.O; n.Wvar = o; o += (size_t)d.U * d.O; n.bvar = o;
    return n;
}

// Weight stream of (member, wave): A-operand order of v_mfma_f32_16x16x4_f32 for out^T = W^T h^T.
// group = [g(2)][lane(64)][r(4)]; lane = 16*kq + i holds W[k = 16F + 4kq + r][out = 16G + i]: the k-quad of
// MFMA step (F, r) is {16F + r, 16F+4 + r, 16F+8 + r, 16F+12 + r}, i.e. exactly what accumulator register r
// of the producing layer holds across the four lane groups.
void pack_member(const Dims &d, const float *nat, float *out)
{
    const NatOff no = nat_offsets(d);
    std::memset(out, 0, (size_t)d.member_stride_f4 * 4 * sizeof(float));
    for (int w = 0; w < 4; ++w) {
        float *dst = out + (size_t)d.wave_off_f4[w] * 4;
        auto emit = [&](const float *W, int in_dim, int out_dim, int ld, int F, int g, int Gout) {
            for (int lane = 0; lane < 64; ++lane) {
                const int kq = lane >> 4, i = lane & 15;
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * F + 4 * kq + r, o = 16 * Gout + i;
                    dst[((size_t)g * 64 + lane) * 4 + r] = (k < in_dim && o < out_dim) ? W[(size_t)k * ld + o] : 0.f;
                }
            }
        };
        // groups are stored in the order the wave visits them: its own input blocks first (cem_perm_l0 / cem_perm_hidden)
        for (int P = 0; P < d.KF0; ++P) {                         // layer 0 (blocks >= KB_in are all zero)
            const int F = cem_perm_l0(w, d.NFW, P);
            emit(nat + no.W[0], d.Din, d.U, d.U, F, 0, 2 * w); emit(nat + no.W[0], d.Din, d.U, d.U, F, 1, 2 * w + 1);
            dst += 512;
        }
        for (int l = 1; l < d.L; ++l)
            for (int P = 0; P < CEM_NG; ++P) {
                const int F = cem_perm_hidden(w, P), Fb = cem_perm_hidden(w, P < 2 ? (P ^ 1) : P);   // second accumulator: own blocks swapped
                emit(nat + no.W[l], d.U, d.U, d.U, F, 0, 2 * w); emit(nat + no.W[l], d.U, d.U, d.U, Fb, 1, 2 * w + 1);
                dst += 512;
            }
        for (int i = 0; i < d.NFW; ++i) {
            const int Fo = w + 4 * i;
            if (Fo >= d.KB_obs) continue;
            for (int P = 0; P < CEM_NG; ++P) {                    // heads: g=0 mu, g=1 var of obs block Fo
                const int F = cem_perm_hidden(w, P), Fb = cem_perm_hidden(w, P < 2 ? (P ^ 1) : P);
                emit(nat + no.Wmu, d.U, d.O, d.O, F, 0, Fo); emit(nat + no.Wvar, d.U, d.O, d.O, Fb, 1, Fo);
                dst += 512;
            }
        }
    }
}

// Weight stream of (member, wave) for cem_rollout_split.h: groups of 6 KB in visiting order, one per K = 32 chunk (input blocks
// 2F, 2F + 1): [a planes 0..2][b planes 0..2], a plane = [64 lanes][8 bf16].  Lane 16 q + i holds, for output feature 16 G + i, the
// chunk's k slots s = 0..7 = input features 16 (2F + s / 4) + 4 q + s % 4 — what a lane's two accumulator quads of the producing
// stage hold — as piece `plane` of the exact three-way bf16 split of the weight (cem_split3_bits).
void pack_member_split(const Dims &d, const float *nat, uint16_t *out)
{
    const NatOff no = nat_offsets(d);
    std::memset(out, 0, (size_t)d.member_stride_f4 * 16);
    for (int w = 0; w < 4; ++w) {
        uint16_t *dst = out + (size_t)d.wave_off_f4[w] * 8;
        auto emit = [&](const float *W, int in_dim, int out_dim, int ld, int F, int ab, int Gout) {
            for (int lane = 0; lane < 64; ++lane) {
                const int kq = lane >> 4, i = lane & 15;
                for (int s = 0; s < 8; ++s) {
                    const int k = 16 * (2 * F + (s >> 2)) + 4 * kq + (s & 3), o = 16 * Gout + i;
                    const float v = (k < in_dim && o < out_dim) ? W[(size_t)k * ld + o] : 0.f;
                    unsigned a[3]; cem_split3_bits(v, a[0], a[1], a[2]);
                    for (int pl = 0; pl < 3; ++pl) dst[((size_t)(ab * 3 + pl) * 64 + lane) * 8 + s] = (uint16_t)(a[pl] >> 16);
                }
            }
        };
        for (int P = 0; P < cem_split_l0_chunks(d.NFW); ++P) {    // layer 0: chunks in ascending order (no own chunk); chunks past the input are zero
            emit(nat + no.W[0], d.Din, d.U, d.U, P, 0, 2 * w); emit(nat + no.W[0], d.Din, d.U, d.U, P, 1, 2 * w + 1);
            dst += 3072;
        }
        for (int l = 1; l < d.L; ++l)
            for (int P = 0; P < CEM_SPLIT_CHUNKS; ++P) {
                const int F = cem_split_perm(w, P);
                emit(nat + no.W[l], d.U, d.U, d.U, F, 0, 2 * w); emit(nat + no.W[l], d.U, d.U, d.U, F, 1, 2 * w + 1);
                dst += 3072;
            }
        for (int i = 0; i < d.NFW; ++i) {
            const int Fo = w + 4 * i;
            if (Fo >= d.KB_obs) continue;
            for (int P = 0; P < CEM_SPLIT_CHUNKS; ++P) {          // heads: a = mean, b = variance of observation block Fo
                const int F = cem_split_perm(w, P);
                emit(nat + no.Wmu, d.U, d.O, d.O, F, 0, Fo); emit(nat + no.Wvar, d.U, d.O, d.O, F, 1, Fo);
                dst += 3072;
            }
        }
    }
}

// Weight image of one member for cem_rollout_wide_kernel: 1 KB groups [64 lanes][4] in cem_wide_base / cem_wide_groups order;
// lane (q, j), word r of group (k block kb, output block ob) = W[16 kb + 4 q + r][16 ob + j], zero past the matrix.
// rows of the per-member feature table (RolloutParams::etab): the wide kernel's hidden layers are up to 256 features = two rows each
size_t etab_rows(const Dims &d) { return CEM_ET_ROWS + (d.wide ? 2 : 1) * (size_t)d.L; }
size_t wide_image_floats(const Dims &d) { return (size_t)cem_wide_groups(d.L, d.KB_in, (d.U + 15) / 16, d.KB_obs) * 256; }
void pack_member_wide(const Dims &d, const float *nat, float *out)
{
    const NatOff no = nat_offsets(d);
    const int nbU = (d.U + 15) / 16, nbIn = d.KB_in, nbO = d.KB_obs;
    auto emit = [&](const float *W, int in_dim, int out_dim, int g, int kb, int ob) {
        float *dst = out + (size_t)g * 256;
        for (int lane = 0; lane < 64; ++lane) {
            const int q = lane >> 4, j = lane & 15;
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * kb + 4 * q + r, o = 16 * ob + j;
                dst[lane * 4 + r] = (k < in_dim && o < out_dim) ? W[(size_t)k * out_dim + o] : 0.f;
            }
        }
    };
    for (int l = 0; l < d.L; ++l) {
        const int in = l == 0 ? d.Din : d.U, nbK = l == 0 ? nbIn : nbU, gl = cem_wide_base(l, nbIn, nbU);
        for (int kb = 0; kb < nbK; ++kb)
            for (int ob = 0; ob < nbU; ++ob) emit(nat + no.W[l], in, d.U, gl + kb * nbU + ob, kb, ob);
    }
    const int gh = cem_wide_base(d.L, nbIn, nbU);
    for (int kb = 0; kb < nbU; ++kb)
        for (int ob = 0; ob < nbO; ++ob) {
            emit(nat + no.Wmu, d.U, d.O, gh + kb * nbO + ob, kb, ob);
            emit(nat + no.Wvar, d.U, d.O, gh + nbU * nbO + kb * nbO + ob, kb, ob);
        }
}

struct Tile6 { int32_t v[6]; };

// tiles of one particle-major row space.  Rows r = p*nstride + n (n in [n_lo, n_hi)) use member
// (p*N + n) / chunk; a tile never straddles a member boundary (mlp_ensemble.py:123-126).
void build_plan_tiles(const Dims &d, int rc, std::vector<Tile6> &out)
{
    out.clear();
    const long long chunk = (long long)d.Btot / d.E;
    const int rows_per_tile = 16 * rc;
    for (int p = 0; p < d.P; ++p) {
        int n = d.n_off; const int n_end = d.n_off + d.Nloc;
        while (n < n_end) {
            const long long rg = (long long)p * d.N + n;
            const int member = (int)(rg / chunk);
            const long long seg_end_g = std::min<long long>((long long)(member + 1) * chunk, (long long)p * d.N + n_end);
            const int seg = (int)(seg_end_g - rg);
            const int ntile = (seg + rows_per_tile - 1) / rows_per_tile;
            int done = 0;
            for (int t = 0; t < ntile; ++t) {
                const int cnt = (seg - done + (ntile - t) - 1) / (ntile - t);     // even split
                Tile6 td; td.v[0] = p * d.Nloc + (n - d.n_off) + done; td.v[1] = cnt; td.v[2] = member;
                td.v[3] = n + done; td.v[4] = (int)(rg + done); td.v[5] = -1;
                out.push_back(td); done += cnt;
            }
            n += seg;
        }
    }
    // XCD-aware order: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch), so give each XCD a
    // contiguous range of tiles = as few members (weight sets) as possible per private L2.
    const int T = (int)out.size();
    std::vector<Tile6> perm(T);
    const int qd = T / 8, rm = T % 8;
    for (int b = 0; b < T; ++b) {
        const int xcd = b % 8, slot = b / 8;
        const int base = xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd;
        perm[b] = out[base + slot];
    }
    out.swap(perm);
}

// ---- tile-size choice ------------------------------------------------------------------------------------------------
// A tile is 16*rc rows of one member; a CU keeps as many workgroups resident as their registers allow.  fp32 MFMA and
// VALU work add up on a SIMD whichever wave they come from, so a co-resident workgroup cannot hide arithmetic — but it
// does hide the stalls (barrier skew, LDS and L2 latency at the stage boundaries).  Cost of one 16-row chunk for the whole
// horizon in ms at H = 30 on MI355X, measured with exactly m tiles per CU (scripts/sweep_chunk_costs.py, round 3 with the
// rotating issue priority: profiles/r03_chunk_costs.jsonl): kChunkStart[nfw][rc][k-1] when k tiles start together on a CU and all
// stay resident (k up to the residency), kChunkNext[nfw][rc] for every further tile the dispatcher starts as an earlier one
// retires (fitted to m = 6 and to the BASELINE-config sweeps: B3 32, B5 10, B4 8 chunks per CU).  Co-resident tiles now advance
// together and retire together, so ONE tile more than the residency runs its whole horizon alone: it costs a round of two.
// Larger tiles re-use each streamed weight group for more rows; smaller ones pack the CUs more evenly and co-reside more easily.
// The constants are GENERATED from a sweep by fixed rules (scripts/sweep_chunk_costs.py --emit-table): regenerate them for another device
// or after a kernel change; tests/test_gpu_tileplan.py bounds how far the resulting automatic choice may be from the best forced one.
#include "cem_tile_costs.inc"
#define CEM_MAX_DEVICES 64
// workgroups of a <rc, nfw> tile one CU keeps resident, from the kernels' VGPR counts (512 registers per SIMD lane; round 3:
// plain kernel 139/159/186/218, 165/217/253/288; segment kernel 144/163/190/223, 171/221/255/292; round 4, with the sampler as the tiles'
// prologue: 153/161/189/222, 167/219/243/293 and 157/163/190/223, 172/222/255/294), [form][nfw - 1][rc - 1]
static const int kResidentStatic[2][2][4] = {{{4, 3, 2, 2}, {3, 2, 2, 1}}, {{3, 3, 2, 2}, {3, 2, 2, 1}}};

template <int RC, int NFW>
int query_resident(bool seg)
{
    int n = 0;
    const size_t lds = CEM_ROLLOUT_LDS_BYTES(RC);
    const hipError_t e = seg ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cem_rollout_seg_kernel<RC, NFW>, 256, lds)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cem_rollout_kernel<RC, NFW, 0>, 256, lds);
    if (e != hipSuccess || n < 1) { (void)hipGetLastError(); return 0; }
    return n;
}

// Per-device facts the tile plan is priced with, asked from the runtime once per device (thread-safe: one std::call_once per
// device slot; the last slot serves a process without a device and holds the static tables).
struct DeviceFacts { std::once_flag once; int resident[2][2][4]; int cus; int cus_real; };
static DeviceFacts g_facts[CEM_MAX_DEVICES + 1];

const DeviceFacts &device_facts()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = CEM_MAX_DEVICES; }     // no device: the static table's slot
    if (dev < 0 || dev > CEM_MAX_DEVICES) dev = CEM_MAX_DEVICES;
    DeviceFacts &f = g_facts[dev];
    std::call_once(f.once, [&] {
        const bool have = dev < CEM_MAX_DEVICES;
        for (int seg = 0; seg < 2; ++seg)
            for (int nfw = 1; nfw <= 2; ++nfw)
                for (int rc = 1; rc <= 4; ++rc) {
                    int n = 0;
                    if (have) {
#define CEM_CASE(R, F) if (rc == R && nfw == F) n = query_resident<R, F>(seg != 0);
                        CEM_CASE(1, 1) CEM_CASE(2, 1) CEM_CASE(3, 1) CEM_CASE(4, 1)
                        CEM_CASE(1, 2) CEM_CASE(2, 2) CEM_CASE(3, 2) CEM_CASE(4, 2)
#undef CEM_CASE
                    }
                    f.resident[seg][nfw - 1][rc - 1] = n > 0 ? n : kResidentStatic[seg][nfw - 1][rc - 1];
                }
        f.cus = 256;                                // MI355X; a partitioned or different device reports its own count
        hipDeviceProp_t pr;
        if (have && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) f.cus = pr.multiProcessorCount;
        else (void)hipGetLastError();
        // The cost model is per CU (tiles on the busiest CU x the cost of a chunk among k co-resident ones): another CU count only
        // changes how many tiles a CU gets, not the table.  CEM_ASSUME_CUS=n prices plans for n CUs (a diagnostic: the GPU-less host
        // helpers and tests/test_capi_cpu.py use it to see the choice move with the CU count).  It moves the tile PLAN only: whatever
        // has to hold on the device that runs the plan — how many workgroups are resident at once (the fused select's grid barriers, where
        // the sampler runs) — is computed from the real count, cus_real.
        f.cus_real = f.cus;
        if (const char *e = std::getenv("CEM_ASSUME_CUS")) { const int n = std::atoi(e); if (n >= 1 && n <= 4096) f.cus = n; }
    });
    return f;
}

// seg: the pinned + floating-segment launch (cem_rollout_seg_kernel) instead of one workgroup per tile (cem_rollout_kernel)
int resident_workgroups(int nfw, int rc, bool seg = false) { return device_facts().resident[seg ? 1 : 0][nfw - 1][rc - 1]; }
int num_cus() { return device_facts().cus; }            // what tile plans are priced for (CEM_ASSUME_CUS moves it)
int real_cus() { return device_facts().cus_real; }      // what the device has

// ---- pinned tiles + floating horizon segments ----------------------------------------------------------------------------
// One workgroup per tile for the whole horizon makes the busiest CU carry ceil(tiles / CUs) tiles while the mean is
// tiles / CUs (B2: 3 vs 2.44).  When every tile is resident at once that ratio is lost outright.  cem_rollout_seg_kernel gets
// most of it back: floor(tiles / CUs) tiles per CU stay whole ("pinned"), the remainder "float" — cut into S horizon segments
// that run, at raised priority, in whatever slot is free, so every CU carries about the same share of them.
// Measured on MI355X (scripts/sweep_seg.py, profiles/r02_sweep_floating_segments.jsonl): B2 0.475 -> 0.405 ms with S = 4..15;
// no gain with ONE pinned tile per CU (375 / 470 tiles: 0.335 -> 0.335, 0.353 -> 0.373 ms: the pinned tile runs without a
// partner and a floater's 30-step chain plus its hand-overs is as long as two whole tiles) and none when the remainder nearly
// fills the CUs anyway (750 tiles: 0.498 -> 0.492).  Hence: at least two pinned tiles per CU, and a predicted gain of > 4 %.
static const int kSegMaxSegments = 6, kSegMinSteps = 5;
// kFloatFactor (cem_tile_costs.inc): pinned + floating launch vs (mean tiles per CU) x the all-resident chunk cost, measured at B2

// cost of `per_cu` tiles of rc chunks queued on the BUSIEST CU, which keeps `occ` resident; `fill` = mean tiles per CU / per_cu <= 1.
// Co-resident tiles are priced by a sweep in which EVERY CU carries them; when only some do (fill < 1), the chip as a whole streams
// fewer weight groups through L2 and the doubly / triply loaded CUs finish earlier than that sweep says (375 one-chunk tiles: 0.305 ms
// against 2 x kChunkStart[0][0][1] = 0.335): kPartialFill, measured by the same sweep at 1.5 tiles per CU, scales that back.
double cu_cost(int nfw, int rc, long per_cu, int occ, double fill)
{
    const long k = std::min<long>(per_cu, occ);                             // tiles that start together (the table has three columns: a fourth co-resident tile is priced like the third)
    if (k < 1) return 0.0;
    const long later = per_cu - k + (per_cu == k + 1 && k > 1 ? 1 : 0);    // one tile beyond the residency: as dear as two (see above)
    const double relief = k >= 2 ? 1.0 - kPartialFill * (1.0 - std::min(std::max(fill, 0.0), 1.0)) : 1.0;
    return (double)rc * ((double)k * kChunkStart[nfw - 1][rc - 1][std::min<long>(k, 3) - 1] * relief + (double)later * kChunkNext[nfw - 1][rc - 1]);
}

int segments_for(const Dims &d, int rc, size_t n_tiles, int requested)
{
    if (requested == 1 || n_tiles >= (size_t)1 << 23) return 1;       // items are packed as (tile << 8 | segment)
    auto clamp_to_horizon = [&](int S) { S = std::min(S, d.H); while (S > 1 && (d.H + S - 1) / S * (S - 1) >= d.H) --S; return std::max(S, 1); };
    if (requested > 1) return clamp_to_horizon(requested);
    const int occ = resident_workgroups(d.NFW, rc, true), kNumCUs = num_cus();
    if (n_tiles < (size_t)2 * kNumCUs || n_tiles > (size_t)kNumCUs * occ) return 1;  // < 2 pinned tiles per CU: no gain (measured);
                                                                                  // more tiles than slots: the dispatcher already refills
    const int S = clamp_to_horizon(std::min(kSegMaxSegments, d.H / kSegMinSteps));
    if (S < 2) return 1;
    const double L = (double)n_tiles / kNumCUs;
    const long per_cu = (long)std::ceil(L);
    const double plain = cu_cost(d.NFW, rc, per_cu, occ, L / (double)per_cu), floating = (double)rc * L * kChunkStart[d.NFW - 1][rc - 1][std::min<long>(per_cu, 3) - 1] * kFloatFactor;
    return plain / floating > 1.10 ? S : 1;      // (a remainder that nearly fills the CUs gains nothing: 750 tiles measured 0.498 -> 0.492 ms in round 2)
}

double tile_plan_cost(const Dims &d, int rc, size_t n_tiles, int requested_segments)
{
    const int kNumCUs = num_cus();
    const int S = segments_for(d, rc, n_tiles, requested_segments);
    const double L = (double)n_tiles / kNumCUs;
    const long per_cu = (long)((n_tiles + kNumCUs - 1) / kNumCUs);
    if (S > 1 && n_tiles > (size_t)kNumCUs) {     // pinned tiles + floating segments: every CU carries the mean load
        if (n_tiles < (size_t)2 * kNumCUs)        // ONE pinned tile per CU runs without a partner and the floaters' chains set the time
            return (double)rc * L * kChunkStart[d.NFW - 1][rc - 1][0] * 1.35;       // (B2 at rc 2: 0.53 ms for 1.23 x 2 chunks per CU, round 3)
        return (double)rc * L * kChunkStart[d.NFW - 1][rc - 1][std::min<long>(per_cu, 3) - 1] * kFloatFactor;
    }
    return cu_cost(d.NFW, rc, per_cu, resident_workgroups(d.NFW, rc, false), per_cu > 0 ? L / (double)per_cu : 1.0);
}

int auto_chunks(const Dims &d, int requested_segments)
{
    int best = 1; double bestc = 1e30;
    for (int rc = 1; rc <= 4; ++rc) {
        std::vector<Tile6> t; build_plan_tiles(d, rc, t);
        const double cost = tile_plan_cost(d, rc, t.size(), requested_segments);
        // rc ascends: a cost within 0.5 % of the best so far goes to the larger tile (fewer workgroups, less weight traffic)
        if (cost <= bestc * 1.005) { best = rc; bestc = std::min(bestc, cost); }
    }
    return best;
}

// Tile size of the split-product rollout (cem_rollout_split.h).  It streams 1.5x the weight bytes at 2.7x the matrix rate, so one-chunk
// tiles are bound by the L2 -> CU path and larger tiles pay; what decides between the sizes is how many tiles the busiest CU gets
// (ceil(tiles / CUs)) and how many of them run side by side.  K cycles per step of ONE CU running m co-resident tiles of rc chunks,
// measured over population sizes (scripts/sweep_split_tiles.py, profiles/r03_split_tile_sizes.txt); residency: registers / LDS.
static const double kSplitStep[2][4][3] = {{{11.2, 18.2, 28.5}, {15.8, 25.0, 0}, {21.6, 34.2, 0}, {27.2, 0, 0}},
                                           {{14.2, 26.0, 0}, {21.0, 39.0, 0}, {28.5, 0, 0}, {36.4, 0, 0}}};
static const int kSplitResident[2][4] = {{3, 2, 2, 1}, {2, 2, 1, 1}};
int auto_chunks_split(const Dims &d)
{
    int best = 1; double bestc = 1e30;
    for (int rc = 1; rc <= 4; ++rc) {
        std::vector<Tile6> t; build_plan_tiles(d, rc, t);
        const long per_cu = (long)((t.size() + num_cus() - 1) / num_cus());
        const int R = kSplitResident[d.NFW - 1][rc - 1];
        const double *ts = kSplitStep[d.NFW - 1][rc - 1];
        const double cost = (double)(per_cu / R) * ts[R - 1] + (per_cu % R ? ts[per_cu % R - 1] : 0.0);
        // rc ascends: within 5 % the larger tile wins (fewer workgroups, a third or half of the weight traffic)
        if (cost <= bestc * 1.05) { best = rc; bestc = std::min(bestc, cost); }
    }
    return best;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
    size_t ctrl, musig, act_bounds, scores_local, scores_global, actions, act_pad, elite, returns, costs, result, wpack, bias_h, bias_mu, bias_var,
        nmin, ndelta, omask, kind_sel, etab, tiles, eps_out, stamps, seg_queue, seg_flags, seg_state,
        ms_hist, ms_sel, ms_counts, ms_best_sc, ms_best_ix, ms_part, ms_colmean, total;
};

struct Plan { int rc, n_tiles, n_seg, seg_len, n_pinned; };

// tile size, tile count and horizon segments of a configuration: one function, so workspace_bytes / create / the host helpers agree
Plan make_plan(const cem_config_t *c, const Dims &d)
{
    Plan pl{};
    pl.rc = d.wide ? 1 : (c->chunks_per_tile ? c->chunks_per_tile : auto_chunks(d, c->rollout_segments));   // the wide kernel: 16-row tiles
    if (d.split && !c->chunks_per_tile) pl.rc = auto_chunks_split(d);
    std::vector<Tile6> t; build_plan_tiles(d, pl.rc, t);
    pl.n_tiles = (int)t.size();
    pl.n_seg = (d.wide || d.split) ? 1 : segments_for(d, pl.rc, t.size(), c->rollout_segments);
    pl.seg_len = (d.H + pl.n_seg - 1) / pl.n_seg;
    pl.n_seg = (d.H + pl.seg_len - 1) / pl.seg_len;
    // tiles every CU gets the same number of stay whole ("pinned"); only the remainder floats in segments
    pl.n_pinned = pl.n_seg > 1 ? (pl.n_tiles / num_cus()) * num_cus() : pl.n_tiles;
    if (c->rollout_segments > 1 && pl.n_pinned == pl.n_tiles && pl.n_seg > 1) pl.n_pinned = 0;   // an explicit request floats everything
    return pl;
}

Layout make_layout(const cem_config_t *c, const Dims &d, size_t max_tiles)
{
    Layout l{}; size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    l.ctrl = take(sizeof(CtrlBlock));
    l.musig = take((size_t)2 * d.H * d.A * 4);
    l.act_bounds = take(64 * 4);
    l.scores_local = take((size_t)d.Nloc * 4);
    l.scores_global = d.W > 1 ? take((size_t)d.N * 4) : l.scores_local;
    l.actions = take((size_t)d.N * d.H * d.A * 4);
    l.act_pad = take((size_t)d.N * d.H * d.act_nq * 16);
    l.elite = take((size_t)d.k * 4);
    l.returns = take((size_t)d.Bloc * 4);
    l.costs = take((size_t)d.H * d.Bloc);
    l.result = take(64 * 4);
    l.wpack = take(d.wide ? (size_t)d.E * (align256(d.nat_member_floats * 4) + wide_image_floats(d) * 4) : (size_t)d.E * d.member_stride_f4 * 16);   // wide: natural blobs, then the packed images
    l.bias_h = take((size_t)d.E * d.L * CEM_U * 4);
    l.bias_mu = take((size_t)d.E * CEM_U * 4);
    l.bias_var = take((size_t)d.E * CEM_U * 4);
    l.nmin = take(CEM_U * 4);
    l.ndelta = take(CEM_U * 4);
    l.omask = take(2 * CEM_U * 4);
    l.kind_sel = take(CEM_NKIND * CEM_U * 4);
    l.etab = take((size_t)d.E * etab_rows(d) * CEM_U * 4);
    l.tiles = take(max_tiles * sizeof(TileDesc));
    l.eps_out = take(CEM_MAX_ACT * 4);
    l.stamps = take(std::max<size_t>(max_tiles * 4 * 8, 128) * sizeof(long long));      // [tiles][4][8] rollout stamps; [64..71] select stamps
    const Plan pl = make_plan(c, d);
    l.seg_queue = take(256);
    const size_t n_float = (size_t)(pl.n_tiles - pl.n_pinned);
    l.seg_flags = take(std::max<size_t>(n_float * std::max(pl.n_seg - 1, 1), 1) * 4);
    l.seg_state = take(pl.n_seg > 1 ? std::max<size_t>(n_float, 1) * (2 * d.NFW * pl.rc * 256 + 64) * 16 : 16);
    // multi-workgroup select (large populations): digit histograms, per-slice counts and best elites, moment partial sums
    const size_t msG = ((size_t)d.N + CEM_MS_KEYS - 1) / CEM_MS_KEYS, msG2 = ((size_t)d.k + CEM_MS_EPG - 1) / CEM_MS_EPG;
    l.ms_hist = take(3 * CEM_MS_BINS * 4); l.ms_sel = take(256);
    l.ms_counts = take(msG * 2 * 4); l.ms_best_sc = take(msG * 4); l.ms_best_ix = take(msG * 4);
    l.ms_part = take(2 * msG2 * (size_t)d.H * d.A * 4); l.ms_colmean = take((size_t)d.H * d.A * 4);
    l.total = o;
    return l;
}

size_t max_tiles_of(const Dims &d) { std::vector<Tile6> t; build_plan_tiles(d, 1, t); return t.size(); }

}  // namespace

// ---- RCCL, opened at run time: the library has no link-time dependency on it (single-GPU users never load it), and in a
// process that already holds an RCCL (torch bundles one under the same SONAME) dlopen returns THAT copy, not a second one.
namespace {
struct CemNcclId { char internal[CEM_COMM_ID_BYTES]; };
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(CemNcclId *) = nullptr;
    int (*CommInitRank)(void **, int, CemNcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
};
Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;                      // two handles may ask at once (one per env thread)
    std::call_once(once, [] {
        // CEM_RCCL_LIBRARY: load THIS file instead (a site's own RCCL build; tests/fakes/libfake_rccl.so, the shared-memory
        // stand-in with which several ranks share a one-GPU box).  Set and not loadable: no fallback, the error says so.
        if (const char *over = std::getenv("CEM_RCCL_LIBRARY")) {
            r.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) std::fprintf(stderr, "cem_mpc: CEM_RCCL_LIBRARY=%s: %s\n", over, dlerror());
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.lib) break;
            }
        }
        if (r.lib) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
            r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
            r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");     // optional: only cem_planner_comm_ranks needs it
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) r.lib = nullptr;
        }
        // an environment variable has just decided which code the product calls for its collectives: say so, once, either way
        if (const char *over = std::getenv("CEM_RCCL_LIBRARY"))
            std::fprintf(stderr, r.lib ? "cem_mpc: RCCL entry points bound from CEM_RCCL_LIBRARY=%s%s\n" : "cem_mpc: CEM_RCCL_LIBRARY=%s lacks a required entry point%s\n",
                         over, (r.lib && !r.CommCount) ? " (no ncclCommCount: cem_planner_comm_ranks unavailable)" : "");
    });
    return r.lib ? &r : nullptr;
}
const int kNcclFloat32 = 7;        // ncclDataType_t ncclFloat32 (rccl.h)
}  // namespace
#define NCCLCHK(x) do { int e_ = (x); if (e_ != 0) { g_last_hip = e_; return CEM_ERR_COMM; } } while (0)

// A host that loads this library BEFORE the HIP runtime that owns its device memory (torch's wheel carries its own libamdhip64) ends up
// with two runtimes in the process: pointers from one are foreign to the other and the first call fails with hipErrorNoDevice or worse
// (INTEGRATION.md section 2).  The Python binding avoids it by importing torch first; any other host gets told, once, at its first create.
namespace {
int count_hip_runtimes(struct dl_phdr_info *info, size_t, void *data)
{
    if (info->dlpi_name && std::strstr(info->dlpi_name, "libamdhip64")) {
        auto *v = static_cast<std::vector<std::string> *>(data);
        if (std::find(v->begin(), v->end(), std::string(info->dlpi_name)) == v->end()) v->push_back(info->dlpi_name);
    }
    return 0;
}
void warn_if_two_hip_runtimes()
{
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> libs;
        dl_iterate_phdr(count_hip_runtimes, &libs);
        if (libs.size() > 1) {
            std::fprintf(stderr, "cem_mpc: %zu HIP runtimes are loaded in this process:", libs.size());
            for (const auto &l : libs) std::fprintf(stderr, " %s", l.c_str());
            std::fprintf(stderr, "\ncem_mpc: device memory and streams of one are foreign to the other — load libcem_mpc_gfx950.so AFTER the runtime that owns them (INTEGRATION.md)\n");
        }
    });
}
}  // namespace

struct cem_planner {
    cem_config_t cfg;
    Dims d;
    Layout lay;
    char *ws;
    hipStream_t stream;
    bool own_stream;
    int rc;
    int n_tiles;
    int n_seg, seg_len, n_pinned;           // horizon segments of the floating tiles (1 = one workgroup per tile), tiles that stay whole
    bool have_weights;
    bool in_plan;
    bool sample_in_rollout;                  // the sampler runs as the rollout tiles' prologue (else: cem_sample_kernel in front of the rollout launch)
    const float *eps_act, *eps_model;       // current plan's explicit noise (device) or null
    // pinned host staging
    CtrlBlock *h_ctrl;
    float *h_result;
    uint32_t plan_seq;                       // plans staged on this handle (CtrlBlock::seq)
    const CtrlBlock *d_h_ctrl; float *d_h_result;     // the same two blocks as the DEVICE addresses them (hipHostGetDevicePointer)
    // timing
    bool timing; std::vector<hipEvent_t> ev; float roll_ms, sel_ms, red_ms, samp_ms; int roll_n;
    std::vector<std::pair<int, int>> ev_kind;   // (event index of start, kind 0 rollout / 1 select / 2 reduce / 3 sampler launch)
    // graph
    hipGraph_t graph; hipGraphExec_t gexec; bool graph_ready;
    ScorerDev sc;
    float alpha, beta;
    void *comm;                              // ncclComm_t of cem_planner_comm_init, or null (the host exchanges scores_local -> scores_global)
    int plans_since_comm;                    // the first plan after comm_init runs eagerly (RCCL sets itself up lazily), then the graph is captured
    bool graph_failed;                       // capture with the collective did not work on this stack: stay eager
    size_t sel_dyn_limit;                    // dynamic-LDS allowance of the select kernels on this handle's device
    int fused_resident;                      // workgroups of cem_msel_fused_kernel the device keeps resident at once (occupancy x CUs)
    bool sel_zeroed;                         // the multi-workgroup select's histograms / barrier counter were cleared by this iteration's reduce kernel
    bool fuse_banned;                        // a grid barrier of the fused select expired on this handle once (recovered in-stream): select_mode 2 from then on
    uint32_t inject_next;                    // cem_planner_inject_fault: CtrlBlock::inject of the next plan
    // grow-only device scratch of the standalone ops (unfold_sequences tiles + returns, compute_objective returns + costs)
    char *scratch; size_t scratch_bytes;
    std::vector<float> h_etab;               // host copy of RolloutParams::etab ([E][CEM_ET_ROWS + L][128]); re-uploaded whole by create / set_weights / set_normaliser
};

extern "C" {

int cem_abi_version(void) { return CEM_ABI_VERSION; }
int cem_last_hip_error(void) { return g_last_hip; }

const char *cem_status_string(int s)
{
    switch (s) {
    case CEM_OK: return "ok";
    case CEM_ERR_INVALID_ARG: return "invalid argument";
    case CEM_ERR_UNSUPPORTED: return "unsupported configuration (units <= 256, obs+act <= 128, task 'goal')";
    case CEM_ERR_SPLIT: return "particles*n_samples is not divisible by ensemble_size (tf.split would raise)";
    case CEM_ERR_WORKSPACE: return "workspace too small or misaligned";
    case CEM_ERR_HIP: return "HIP runtime error (see cem_last_hip_error)";
    case CEM_ERR_NO_WEIGHTS: return "set_weights has not been called";
    case CEM_ERR_STATE: return "stepwise plan calls out of order, or a standalone call while a plan is in flight";
    case CEM_ERR_COMM: return "RCCL: library not found or a collective call failed (see cem_last_hip_error for the ncclResult_t)";
    case CEM_ERR_DEVICE: return "a kernel could not finish its work (a floating rollout segment starved of its work-queue entry, or an expired fused-select barrier whose recovery did not run): result not valid";
    default: return "unknown status";
    }
}

size_t cem_weight_blob_floats(const cem_config_t *cfg)
{
    if (validate(cfg) != CEM_OK) return 0;
    const Dims d = make_dims(cfg);
    return d.nat_member_floats * d.E;
}

size_t cem_packed_weight_floats(const cem_config_t *cfg)
{
    if (validate(cfg) != CEM_OK) return 0;
    const Dims d = make_dims(cfg);
    if (d.wide) return 0;                               // the wide kernel reads the natural blob: nothing is packed
    return (size_t)d.member_stride_f4 * 4 * d.E;
}

size_t cem_workspace_bytes(const cem_config_t *cfg)
{
    if (validate(cfg) != CEM_OK) return 0;
    const Dims d = make_dims(cfg);
    return make_layout(cfg, d, max_tiles_of(d)).total;
}

int cem_pack_weights_host(const cem_config_t *cfg, const float *blob, float *packed)
{
    int st = validate(cfg); if (st) return st;
    if (!blob || !packed) return CEM_ERR_INVALID_ARG;
    const Dims d = make_dims(cfg);
    if (d.wide) return CEM_ERR_UNSUPPORTED;
    for (int m = 0; m < d.E; ++m) {
        if (d.split) pack_member_split(d, blob + (size_t)m * d.nat_member_floats, reinterpret_cast<uint16_t *>(packed + (size_t)m * d.member_stride_f4 * 4));
        else pack_member(d, blob + (size_t)m * d.nat_member_floats, packed + (size_t)m * d.member_stride_f4 * 4);
    }
    return CEM_OK;
}

int cem_plan_tiles_host(const cem_config_t *cfg, int32_t *rc_out, int32_t *n_tiles_out, int32_t *tiles_out, int32_t max_tiles)
{
    int st = validate(cfg); if (st) return st;
    const Dims d = make_dims(cfg);
    const int rc = make_plan(cfg, d).rc;
    std::vector<Tile6> t; build_plan_tiles(d, rc, t);
    if (rc_out) *rc_out = rc;
    if (n_tiles_out) *n_tiles_out = (int32_t)t.size();
    if (tiles_out) {
        if ((int)t.size() > max_tiles) return CEM_ERR_INVALID_ARG;
        std::memcpy(tiles_out, t.data(), t.size() * sizeof(Tile6));
    }
    return CEM_OK;
}

int cem_plan_segments_host(const cem_config_t *cfg, int32_t *segments_out, int32_t *steps_per_segment_out)
{
    int st = validate(cfg); if (st) return st;
    const Dims d = make_dims(cfg);
    const Plan pl = make_plan(cfg, d);
    if (segments_out) *segments_out = pl.n_seg;
    if (steps_per_segment_out) *steps_per_segment_out = pl.seg_len;
    return CEM_OK;
}

int cem_rollout_residency(int32_t chunks_per_tile, int32_t input_blocks_per_wave, int32_t *table_out, int32_t *runtime_out)
{
    // [0]: one workgroup per tile (cem_rollout_kernel), [1]: the pinned + floating-segment form (cem_rollout_seg_kernel)
    if (chunks_per_tile < 1 || chunks_per_tile > 4 || input_blocks_per_wave < 1 || input_blocks_per_wave > 2) return CEM_ERR_INVALID_ARG;
    for (int seg = 0; seg < 2; ++seg) {
        if (table_out) table_out[seg] = kResidentStatic[seg][input_blocks_per_wave - 1][chunks_per_tile - 1];
        if (runtime_out) {
            int n = 0;
#define CEM_CASE(R, F) if (chunks_per_tile == R && input_blocks_per_wave == F) n = query_resident<R, F>(seg != 0);
            CEM_CASE(1, 1) CEM_CASE(2, 1) CEM_CASE(3, 1) CEM_CASE(4, 1)
            CEM_CASE(1, 2) CEM_CASE(2, 2) CEM_CASE(3, 2) CEM_CASE(4, 2)
#undef CEM_CASE
            runtime_out[seg] = n;
        }
    }
    return CEM_OK;
}

int cem_planner_create(const cem_config_t *cfg, void *workspace, size_t workspace_bytes, void *hip_stream, cem_planner_t **out)
{
    int st = validate(cfg); if (st) return st;
    if (!workspace || !out) return CEM_ERR_INVALID_ARG;
    warn_if_two_hip_runtimes();
    cem_planner *h = new (std::nothrow) cem_planner();
    if (!h) return CEM_ERR_INVALID_ARG;
    h->cfg = *cfg; h->d = make_dims(cfg);
    h->lay = make_layout(cfg, h->d, max_tiles_of(h->d));
    if (workspace_bytes < h->lay.total || ((uintptr_t)workspace & 255)) { delete h; return CEM_ERR_WORKSPACE; }
    h->ws = (char *)workspace; h->stream = (hipStream_t)hip_stream; h->own_stream = false;
    if (!h->stream) {      // the legacy default stream cannot be captured into a hipGraph: use a stream of our own
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { g_last_hip = (int)hipGetLastError(); delete h; return CEM_ERR_HIP; }
        h->own_stream = true;
    }
    { const Plan pl = make_plan(cfg, h->d); h->rc = pl.rc; h->n_seg = pl.n_seg; h->seg_len = pl.seg_len; h->n_pinned = pl.n_pinned;
      // Where the sampler runs (cem_device.h: cem_tile_sample_actions vs cem_sample_kernel).  Inside the rollout launch when ALL its
      // tiles are resident at once (one round of prologues per launch: B1, B2) — one launch and one graph node fewer per iteration for
      // about what the launch cost; as a launch of its own when tiles queue for slots (B3: 8 tiles per CU; every round of tiles would
      // pay the prologue on its critical path, and each candidate is sampled once per particle: K = 16 measured +1.5 % on the launch).
      const int slots = real_cus() * ((h->d.wide || h->d.split) ? 1 : resident_workgroups(h->d.NFW, pl.rc, pl.n_seg > 1));
      h->sample_in_rollout = pl.n_tiles <= slots;
      if (const char *e = std::getenv("CEM_FORCE_SAMPLER"))      // diagnostic / tests: "tile" or "kernel" — the results do not depend on it
          h->sample_in_rollout = std::strcmp(e, "kernel") != 0; }
    h->have_weights = false; h->in_plan = false; h->eps_act = h->eps_model = nullptr;
    h->timing = false; h->roll_ms = h->sel_ms = h->red_ms = h->samp_ms = 0.f; h->roll_n = 0;
    h->graph = nullptr; h->gexec = nullptr; h->graph_ready = false;
    h->comm = nullptr; h->plans_since_comm = 0; h->graph_failed = false;
    h->h_ctrl = nullptr; h->h_result = nullptr;
    h->scratch = nullptr; h->scratch_bytes = 0; h->sel_zeroed = false; h->plan_seq = 0; h->fuse_banned = false; h->inject_next = 0;
    // every failure from here on frees what was acquired and reports the HIP code
    auto fail = [&](int status) {
        g_last_hip = (int)hipGetLastError();
        if (h->h_ctrl) hipHostFree(h->h_ctrl);
        if (h->h_result) hipHostFree(h->h_result);
        if (h->own_stream) hipStreamDestroy(h->stream);
        delete h;
        return status;
    };
    // mapped + coherent (fine-grained) host memory, asked for explicitly: the device reads the staged block and writes the result in place,
    // and the host polls that result while the stream is still running
    if (hipHostMalloc((void **)&h->h_ctrl, sizeof(CtrlBlock), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostMalloc((void **)&h->h_result, 64 * 4, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return fail(CEM_ERR_HIP);
    std::memset(h->h_ctrl, 0, sizeof(CtrlBlock));
    std::memset(h->h_result, 0, 64 * 4);
    {   // kernels read the staged control block and write the plan's result in place (no copy nodes around a plan)
        void *dc = nullptr, *dr = nullptr;
        if (hipHostGetDevicePointer(&dc, h->h_ctrl, 0) != hipSuccess || hipHostGetDevicePointer(&dr, h->h_result, 0) != hipSuccess) return fail(CEM_ERR_HIP);
        h->d_h_ctrl = (const CtrlBlock *)dc; h->d_h_result = (float *)dr;
    }
    auto upload = [&](size_t off, const void *src, size_t bytes) {
        return hipMemcpyAsync(h->ws + off, src, bytes, hipMemcpyHostToDevice, h->stream) == hipSuccess;
    };

    // scorer constants, rounded the way the reference's Python-float -> fp32 tensor conversion rounds them
    const cem_scorer_t &s = cfg->scorer;
    ScorerDev &sc = h->sc;
    sc.goal_mode = s.goal_mode; sc.goal_lo = s.goal_lo; sc.goal_hi = s.goal_hi; sc.D = s.lidar_max_dist;
    sc.goal_thresh = s.goal_reached_dist;        // fl32(0.8 * goal_size) evaluated in double by the caller (safety_gym.py:116)
    sc.reward_distance = s.reward_distance; sc.reward_goal = s.reward_goal; sc.reward_clip = s.reward_clip;
    sc.indicator = s.constrain_indicator; sc.n_cost = s.n_cost_kinds;
    for (int i = 0; i < 4; ++i) { sc.cost_lo[i] = s.cost_lo[i]; sc.cost_hi[i] = s.cost_hi[i]; sc.cost_size[i] = s.cost_size[i]; }
    {   // Beta prior of safe_cem_mpc.py:113-115 in fp32 tensor arithmetic (mu = 0.5, sigma = 0.27 from :81)
        const float mu = 0.5f, sg = 0.27f;
        const float alpha = (((1.0f - mu) / (sg * sg)) - 1.0f / mu) * (mu * mu);
        h->alpha = alpha; h->beta = alpha * (1.0f / mu - 1.0f);
    }

    // tiles, per-feature predicate tables of the rollout epilogue, identity normaliser (until set_normaliser) -> device
    std::vector<Tile6> tiles; build_plan_tiles(h->d, h->rc, tiles);
    h->n_tiles = (int)tiles.size();
    const Dims &d = h->d;
    std::vector<float> om(2 * CEM_U, 0.f), ks(CEM_NKIND * CEM_U, std::numeric_limits<float>::infinity());
    const float ninf = -std::numeric_limits<float>::infinity();
    for (int f = 0; f < CEM_U; ++f) {
        om[f] = f < d.O ? 1.f : 0.f;
        om[CEM_U + f] = (f >= d.O && f < d.O + d.A) ? 1.f : 0.f;
        const int ghi = s.goal_mode ? s.goal_lo + 1 : s.goal_hi;
        if (f >= s.goal_lo && f < ghi && f < d.O) ks[f] = ninf;
        for (int k = 0; k < s.n_cost_kinds; ++k)
            if (f >= s.cost_lo[k] && f < s.cost_hi[k] && f < d.O) ks[(k + 1) * CEM_U + f] = ninf;
    }
    std::vector<float> mn(CEM_U, 0.f), dl(CEM_U, 1.f);
    // the hot kernel's per-member table: identity normaliser and zero biases until set_normaliser / set_weights fill them in
    const size_t et_rows = etab_rows(d);
    h->h_etab.assign((size_t)d.E * et_rows * CEM_U, 0.f);
    for (int m = 0; m < d.E; ++m) {
        float *et = &h->h_etab[(size_t)m * et_rows * CEM_U];
        for (int f = 0; f < CEM_U; ++f) {
            et[CEM_ET_RDELTA * CEM_U + f] = 1.f;
            et[CEM_ET_OBS * CEM_U + f] = om[f]; et[CEM_ET_ACT * CEM_U + f] = om[CEM_U + f];
            et[CEM_ET_SEL0 * CEM_U + f] = ks[f]; et[CEM_ET_SEL1 * CEM_U + f] = ks[CEM_U + f];
        }
    }
    float bounds[64] = {0.f};                              // tf.clip_by_value's lb / ub (cem_mpc.py:48), read by the rollout tiles' sampling prologue
    for (int a = 0; a < d.A; ++a) { bounds[a] = cfg->act_lb[a]; bounds[32 + a] = cfg->act_ub[a]; }
    if (!upload(h->lay.act_bounds, bounds, sizeof(bounds)) ||
        !upload(h->lay.tiles, tiles.data(), tiles.size() * sizeof(Tile6)) || !upload(h->lay.omask, om.data(), om.size() * 4) ||
        !upload(h->lay.kind_sel, ks.data(), ks.size() * 4) || !upload(h->lay.nmin, mn.data(), CEM_U * 4) ||
        !upload(h->lay.ndelta, dl.data(), CEM_U * 4) || !upload(h->lay.etab, h->h_etab.data(), h->h_etab.size() * 4) ||
        hipMemsetAsync(h->ws + h->lay.act_pad, 0, (size_t)d.N * d.H * d.act_nq * 16, h->stream) != hipSuccess ||   // the padding words of the action quads are never written again
        hipStreamSynchronize(h->stream) != hipSuccess)
        return fail(CEM_ERR_HIP);

    // Select kernel: scores staged in LDS when they fit.  gfx950 has 160 KB per CU and this kernel is the CU's only workgroup;
    // beyond the default 64 KB per workgroup the runtime has to be asked, per device (19 KB of the budget are the kernel's
    // static arrays).  Both variants: the uncached one still keeps the elite list (up to 24576 indices = 96 KB) in dynamic LDS.
    h->sel_dyn_limit = 48 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&cem_select_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            140 * 1024) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&cem_select_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            140 * 1024) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&cem_select_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            140 * 1024) == hipSuccess) h->sel_dyn_limit = 140 * 1024;
    else (void)hipGetLastError();
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cem_msel_fused_kernel, 1024, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 0; }
        h->fused_resident = per_cu * real_cus();
    }
    *out = h;
    return CEM_OK;
}

int cem_planner_destroy(cem_planner_t *h)
{
    if (!h) return CEM_ERR_INVALID_ARG;
    // a polled plan returns when its result block lands, which may be before the stream has drained (the graph's tail — e.g. the
    // early-exit kernels behind an early stop — still reads the workspace and the pinned blocks freed below)
    if (hipStreamSynchronize(h->stream) != hipSuccess) (void)hipGetLastError();
    if (h->scratch) hipFree(h->scratch);
    if (h->comm) { if (Rccl *r = rccl()) r->CommDestroy(h->comm); h->comm = nullptr; }
    if (h->gexec) hipGraphExecDestroy(h->gexec);
    if (h->graph) hipGraphDestroy(h->graph);
    for (auto e : h->ev) hipEventDestroy(e);
    if (h->h_ctrl) hipHostFree(h->h_ctrl);
    if (h->h_result) hipHostFree(h->h_result);
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
    return CEM_OK;
}

int cem_planner_layout(const cem_planner_t *h, cem_layout_t *o)
{
    if (!h || !o) return CEM_ERR_INVALID_ARG;
    o->scores_local = h->lay.scores_local; o->scores_global = h->lay.scores_global; o->actions = h->lay.actions;
    o->mu_sigma = h->lay.musig; o->elite_idx = h->lay.elite; o->returns = h->lay.returns; o->costs = h->lay.costs;
    o->result = h->lay.result; o->stamps = h->lay.stamps; o->total = h->lay.total;
    return CEM_OK;
}

int cem_planner_set_weights(cem_planner_t *h, const float *blob, size_t n_floats)
{
    if (!h || !blob) return CEM_ERR_INVALID_ARG;
    const Dims &d = h->d;
    if (n_floats != d.nat_member_floats * d.E) return CEM_ERR_INVALID_ARG;
    if (d.wide) {                                       // the wide kernel: natural blobs (biases) + per-member operand-order images
        const size_t img = wide_image_floats(d);
        std::vector<float> images(img * d.E);
        for (int m = 0; m < d.E; ++m) pack_member_wide(d, blob + (size_t)m * d.nat_member_floats, images.data() + (size_t)m * img);
        HIPCHK(hipMemcpyAsync(h->ws + h->lay.wpack, blob, n_floats * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->ws + h->lay.wpack + align256(n_floats * 4), images.data(), images.size() * 4, hipMemcpyHostToDevice, h->stream));
        const NatOff no = nat_offsets(d);                // the biases as rows of the per-member table (hidden layers: 256 features = 2 rows)
        const size_t et_rows = etab_rows(d);
        for (int m = 0; m < d.E; ++m) {
            const float *nat = blob + (size_t)m * d.nat_member_floats;
            float *et = &h->h_etab[(size_t)m * et_rows * CEM_U];
            std::memset(et + CEM_ET_BMU * CEM_U, 0, 2 * CEM_U * 4);
            std::memcpy(et + CEM_ET_BMU * CEM_U, nat + no.bmu, d.O * 4);
            std::memcpy(et + CEM_ET_BVAR * CEM_U, nat + no.bvar, d.O * 4);
            std::memset(et + CEM_ET_ROWS * CEM_U, 0, (size_t)2 * d.L * CEM_U * 4);
            for (int l = 0; l < d.L; ++l) std::memcpy(et + (CEM_ET_ROWS + 2 * l) * CEM_U, nat + no.b[l], d.U * 4);
        }
        HIPCHK(hipMemcpyAsync(h->ws + h->lay.etab, h->h_etab.data(), h->h_etab.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->have_weights = true;
        return CEM_OK;
    }
    std::vector<float> packed((size_t)d.member_stride_f4 * 4 * d.E);
    std::vector<float> bh((size_t)d.E * d.L * CEM_U, 0.f), bmu((size_t)d.E * CEM_U, 0.f), bvar((size_t)d.E * CEM_U, 0.f);
    const NatOff no = nat_offsets(d);
    for (int m = 0; m < d.E; ++m) {
        const float *nat = blob + (size_t)m * d.nat_member_floats;
        if (d.split) pack_member_split(d, nat, reinterpret_cast<uint16_t *>(packed.data() + (size_t)m * d.member_stride_f4 * 4));
        else pack_member(d, nat, packed.data() + (size_t)m * d.member_stride_f4 * 4);
        for (int l = 0; l < d.L; ++l) std::memcpy(&bh[((size_t)m * d.L + l) * CEM_U], nat + no.b[l], d.U * 4);
        std::memcpy(&bmu[(size_t)m * CEM_U], nat + no.bmu, d.O * 4);
        std::memcpy(&bvar[(size_t)m * CEM_U], nat + no.bvar, d.O * 4);
    }
    HIPCHK(hipMemcpyAsync(h->ws + h->lay.wpack, packed.data(), packed.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->ws + h->lay.bias_h, bh.data(), bh.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->ws + h->lay.bias_mu, bmu.data(), bmu.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->ws + h->lay.bias_var, bvar.data(), bvar.size() * 4, hipMemcpyHostToDevice, h->stream));
    {   // the same biases as rows of the hot kernel's per-member table
        const size_t et_rows = etab_rows(d);
        for (int m = 0; m < d.E; ++m) {
            float *et = &h->h_etab[(size_t)m * et_rows * CEM_U];
            std::memcpy(et + CEM_ET_BMU * CEM_U, &bmu[(size_t)m * CEM_U], CEM_U * 4);
            std::memcpy(et + CEM_ET_BVAR * CEM_U, &bvar[(size_t)m * CEM_U], CEM_U * 4);
            for (int l = 0; l < d.L; ++l) std::memcpy(et + (CEM_ET_ROWS + l) * CEM_U, &bh[((size_t)m * d.L + l) * CEM_U], CEM_U * 4);
        }
        HIPCHK(hipMemcpyAsync(h->ws + h->lay.etab, h->h_etab.data(), h->h_etab.size() * 4, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_weights = true;
    return CEM_OK;
}

int cem_planner_set_normaliser(cem_planner_t *h, const float *imin, const float *imax)
{
    if (!h || !imin || !imax) return CEM_ERR_INVALID_ARG;
    const Dims &d = h->d;
    std::vector<float> mn(CEM_U, 0.f), dl(CEM_U, 1.f);
    if (h->cfg.scale_features) {                      // transition_model.py:79-87
        for (int f = 0; f < d.Din; ++f) {
            float delta = imax[f] - imin[f];
            if (delta < 1e-5f) delta = 1.01f;
            mn[f] = imin[f]; dl[f] = 1.0f / delta;      // the kernel multiplies by 1/delta (<= 1.5 ulp from the reference's division)
        }
    }
    HIPCHK(hipMemcpyAsync(h->ws + h->lay.nmin, mn.data(), CEM_U * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->ws + h->lay.ndelta, dl.data(), CEM_U * 4, hipMemcpyHostToDevice, h->stream));
    {
        const size_t et_rows = etab_rows(d);
        for (int m = 0; m < d.E; ++m) {
            float *et = &h->h_etab[(size_t)m * et_rows * CEM_U];
            std::memcpy(et + CEM_ET_NMIN * CEM_U, mn.data(), CEM_U * 4);
            std::memcpy(et + CEM_ET_RDELTA * CEM_U, dl.data(), CEM_U * 4);
        }
        HIPCHK(hipMemcpyAsync(h->ws + h->lay.etab, h->h_etab.data(), h->h_etab.size() * 4, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return CEM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------------------
namespace {

template <int RC, int NFW, int MODE>
hipError_t launch_rollout_t(const RolloutParams &p, int n_tiles, hipStream_t st)
{
    const size_t lds = CEM_ROLLOUT_LDS_BYTES(RC);
    hipLaunchKernelGGL((cem_rollout_kernel<RC, NFW, MODE>), dim3(n_tiles), dim3(256), lds, st, p);
    return hipGetLastError();
}

template <int RC, int NFW>
hipError_t launch_rollout_seg_t(const RolloutParams &p, int grid, hipStream_t st)
{
    const size_t lds = CEM_ROLLOUT_LDS_BYTES(RC);
    hipLaunchKernelGGL((cem_rollout_seg_kernel<RC, NFW>), dim3(grid), dim3(256), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_rollout_seg(int rc, int nfw, const RolloutParams &p, int grid, hipStream_t st)
{
#define CEM_CASE(R, F) if (rc == R && nfw == F) return launch_rollout_seg_t<R, F>(p, grid, st);
    CEM_CASE(1, 1) CEM_CASE(2, 1) CEM_CASE(3, 1) CEM_CASE(4, 1)
    CEM_CASE(1, 2) CEM_CASE(2, 2) CEM_CASE(3, 2) CEM_CASE(4, 2)
#undef CEM_CASE
    return hipErrorInvalidValue;
}

template <int MODE>
hipError_t launch_rollout(int rc, int nfw, const RolloutParams &p, int n_tiles, hipStream_t st)
{
#define CEM_CASE(R, F) if (rc == R && nfw == F) return launch_rollout_t<R, F, MODE>(p, n_tiles, st);
    CEM_CASE(1, 1) CEM_CASE(2, 1) CEM_CASE(3, 1) CEM_CASE(4, 1)
    CEM_CASE(1, 2) CEM_CASE(2, 2) CEM_CASE(3, 2) CEM_CASE(4, 2)
#undef CEM_CASE
    return hipErrorInvalidValue;
}

// the split-product rollout (cem_rollout_split.h): rc 1 / 2, whole-horizon tiles; mode 1 = caller-supplied noise tensors
template <int RC, int NFW, int MODE>
hipError_t launch_rollout_split_t(const RolloutParams &p, int n_tiles, hipStream_t st)
{
    const size_t lds = (size_t)2 * CEM_SPLIT_XB(RC) + CEM_PART_FLOATS * 4;
    hipLaunchKernelGGL((cem_rollout_split_kernel<RC, NFW, MODE>), dim3(n_tiles), dim3(256), lds, st, p);
    return hipGetLastError();
}
hipError_t launch_rollout_split(int rc, int nfw, int mode, const RolloutParams &p, int n_tiles, hipStream_t st)
{
#define CEM_CASE(R, F, M) if (rc == R && nfw == F && mode == M) return launch_rollout_split_t<R, F, M>(p, n_tiles, st);
    CEM_CASE(1, 1, 0) CEM_CASE(2, 1, 0) CEM_CASE(1, 2, 0) CEM_CASE(2, 2, 0) CEM_CASE(3, 1, 0) CEM_CASE(4, 1, 0) CEM_CASE(3, 2, 0) CEM_CASE(4, 2, 0)
    CEM_CASE(1, 1, 1) CEM_CASE(2, 1, 1) CEM_CASE(1, 2, 1) CEM_CASE(2, 2, 1) CEM_CASE(3, 1, 1) CEM_CASE(4, 1, 1) CEM_CASE(3, 2, 1) CEM_CASE(4, 2, 1)
#undef CEM_CASE
    return hipErrorInvalidValue;
}

// mode 0: planning; mode 1: caller-supplied action / noise tensors, trajectory and head-moment outputs
hipError_t launch_rollout_wide(const cem_planner *h, const RolloutParams &rp, int n_tiles, int mode)
{
    WideParams wp; wp.r = rp;
    wp.U = h->d.U; wp.act = h->cfg.activation;
    wp.wimg = (const f4 *)(h->ws + h->lay.wpack + align256((size_t)h->d.E * h->d.nat_member_floats * 4)); wp.img_f4 = (uint32_t)(wide_image_floats(h->d) / 4);
    if (mode == 0) hipLaunchKernelGGL(cem_rollout_wide_kernel<0>, dim3(n_tiles), dim3(256), CEM_WIDE_SMEM, h->stream, wp);
    else hipLaunchKernelGGL(cem_rollout_wide_kernel<1>, dim3(n_tiles), dim3(256), CEM_WIDE_SMEM, h->stream, wp);
    return hipGetLastError();
}

void fill_rollout_common(const cem_planner *h, RolloutParams &p)
{
    const Dims &d = h->d; const Layout &l = h->lay; char *ws = h->ws;
    std::memset(&p, 0, sizeof(p));
    p.wpack = (const f4 *)(ws + l.wpack); p.bias_h = (const float *)(ws + l.bias_h);
    p.bias_mu = (const float *)(ws + l.bias_mu); p.bias_var = (const float *)(ws + l.bias_var);
    p.nmin = (const float *)(ws + l.nmin); p.nrdelta = (const float *)(ws + l.ndelta);
    p.omask = (const float *)(ws + l.omask); p.kind_sel = (const float *)(ws + l.kind_sel);
    p.etab = (const float *)(ws + l.etab);
    p.act_pad = (const f4 *)(ws + l.act_pad); p.act_pad_bytes = (uint32_t)((size_t)d.N * d.H * d.act_nq * 16); p.act_q0 = d.act_q0; p.act_nq = d.act_nq;
    p.ctrl = (const CtrlBlock *)(ws + l.ctrl);
    p.member_stride_f4 = d.member_stride_f4;
    for (int w = 0; w < 4; ++w) { p.wave_off_f4[w] = d.wave_off_f4[w]; p.wave_groups[w] = (uint32_t)d.wave_groups[w]; }
    p.O = d.O; p.A = d.A; p.L = d.L; p.KB_in = d.KB_in; p.KB_obs = d.KB_obs;
    p.sampling = h->cfg.sampling_propagation; p.sc = h->sc;
}

hipEvent_t get_event(cem_planner *h, size_t i)
{
    while (h->ev.size() <= i) { hipEvent_t e; hipEventCreate(&e); h->ev.push_back(e); }
    return h->ev[i];
}

int enqueue_begin(cem_planner *h)
{
    const Dims &d = h->d; const Layout &l = h->lay;
    InitParams ip{}; ip.ctrl = (CtrlBlock *)(h->ws + l.ctrl); ip.musig = (float *)(h->ws + l.musig); ip.HA = d.H * d.A; ip.A = d.A;
    ip.host_ctrl = h->d_h_ctrl;                          // the block stage_ctrl filled, read from pinned host memory by the kernel itself
    for (int a = 0; a < d.A; ++a) { ip.mu0[a] = h->cfg.act_mu0[a]; ip.sigma0[a] = h->cfg.act_sigma0[a]; }
    if (h->n_seg > 1) { ip.seg_queue = (uint32_t *)(h->ws + l.seg_queue); ip.seg_flags = (uint32_t *)(h->ws + l.seg_flags); ip.n_ready = (h->n_tiles - h->n_pinned) * (h->n_seg - 1); }
    const int n = std::max<int>(ip.HA, (int)(sizeof(CtrlBlock) / 4));
    hipLaunchKernelGGL(cem_init_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, ip);
    HIPCHK(hipGetLastError());
    return CEM_OK;
}

// One iteration up to the scores: the rollout launch (its tiles sample their own action sequences first: cem_tile_sample_actions), then
// the particle mean / Beta filter — unless `fold_reduce`: a single-rank whole plan on the CemMpc objective lets the select kernel form
// the particle mean while it stages its keys (same sum, same order, one launch and one graph node fewer per iteration).
int enqueue_rollout(cem_planner *h, int it, bool fold_reduce)
{
    const Dims &d = h->d; const Layout &l = h->lay; char *ws = h->ws;
    const bool queued = h->n_seg > 1 && !h->eps_model;       // explicit eps_model tensors take the general (MODE 1) kernel
    RolloutParams rp; fill_rollout_common(h, rp);
    rp.tiles = (const TileDesc *)(ws + l.tiles); rp.s0 = nullptr; rp.actions = (const float *)(ws + l.actions);
    rp.eps_model = h->eps_model ? h->eps_model + (size_t)it * d.H * d.Btot * d.O : nullptr;
    rp.ret = (float *)(ws + l.returns); rp.costs = h->cfg.variant == CEM_VARIANT_SAFE ? (uint8_t *)(ws + l.costs) : nullptr;
    rp.H = d.H; rp.Bloc = d.Bloc; rp.Btot = d.Btot; rp.it = it; rp.variant = h->cfg.variant; rp.check_done = 1;
    rp.stamps = (long long *)(ws + l.stamps);
    // the sampler's inputs and outputs (cem_mpc.py:44-48)
    rp.musig = (const float *)(ws + l.musig); rp.eps_act = h->eps_act ? h->eps_act + (size_t)it * d.N * d.H * d.A : nullptr;
    rp.act_bounds = (const float *)(ws + l.act_bounds); rp.actions_w = (float *)(ws + l.actions); rp.act_pad_w = (float *)(ws + l.act_pad);
    rp.pad_shift = d.O - 4 * d.act_q0; rp.pad_floats = 4 * d.act_nq; rp.N = d.N; rp.Nloc = d.Nloc; rp.n_off = d.n_off; rp.n_tiles = h->n_tiles;
    if (!h->sample_in_rollout) {                          // all N candidates once, in front of the rollout launch (which then samples nothing)
        const int total = d.N * d.H * ((d.A + 3) / 4);
        size_t es = 0;
        if (h->timing) { es = h->ev_kind.size() * 2; h->ev_kind.push_back({(int)es, 3}); hipEventRecord(get_event(h, es), h->stream); }
        hipLaunchKernelGGL(cem_sample_kernel, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0, h->stream, rp);
        HIPCHK(hipGetLastError());
        if (h->timing) hipEventRecord(get_event(h, es + 1), h->stream);
        rp.musig = nullptr;
    }
    size_t e0 = 0;
    if (h->timing) { e0 = h->ev_kind.size() * 2; h->ev_kind.push_back({(int)e0, 0}); hipEventRecord(get_event(h, e0), h->stream); }
    if (d.wide) HIPCHK(launch_rollout_wide(h, rp, h->n_tiles, rp.eps_model ? 1 : 0));
    else if (d.split) HIPCHK(launch_rollout_split(h->rc, d.NFW, rp.eps_model ? 1 : 0, rp, h->n_tiles, h->stream));
    else if (rp.eps_model) HIPCHK(launch_rollout<1>(h->rc, d.NFW, rp, h->n_tiles, h->stream));
    else if (queued) {
        rp.seg_queue = (uint32_t *)(ws + l.seg_queue); rp.seg_flags = (uint32_t *)(ws + l.seg_flags); rp.seg_state = (f4 *)(ws + l.seg_state);
        rp.seg_len = h->seg_len; rp.n_seg = h->n_seg; rp.n_pinned = h->n_pinned;
        // one workgroup per pinned tile, then one per (floating tile, segment) item
        HIPCHK(launch_rollout_seg(h->rc, d.NFW, rp, h->n_pinned + h->n_seg * (h->n_tiles - h->n_pinned), h->stream));
    } else HIPCHK(launch_rollout<0>(h->rc, d.NFW, rp, h->n_tiles, h->stream));
    if (h->timing) hipEventRecord(get_event(h, e0 + 1), h->stream);
    if (fold_reduce) return CEM_OK;

    ReduceParams qp{}; qp.ret = rp.ret; qp.costs = rp.costs; qp.scores = (float *)(ws + l.scores_local); qp.ctrl = rp.ctrl;
    qp.Nloc = d.Nloc; qp.P = d.P; qp.H = d.H; qp.variant = h->cfg.variant; qp.check_done = 1;
    qp.alpha = h->alpha; qp.beta = h->beta; qp.thr = h->cfg.posterior_mean_threashold;
    qp.zero = (uint32_t *)(ws + l.ms_hist); qp.zero_n = (3 * CEM_MS_BINS * 4 + 256) / 4; h->sel_zeroed = true;     // for this iteration's multi-workgroup select
    size_t er = 0;
    if (h->timing) { er = h->ev_kind.size() * 2; h->ev_kind.push_back({(int)er, 2}); hipEventRecord(get_event(h, er), h->stream); }
    hipLaunchKernelGGL(cem_reduce_kernel, dim3((d.Nloc + 63) / 64), dim3(CEM_REDUCE_THREADS), 0, h->stream, qp);
    HIPCHK(hipGetLastError());
    if (h->timing) hipEventRecord(get_event(h, er + 1), h->stream);
    return CEM_OK;
}

// whether a whole plan of this handle folds the particle mean into the select kernel: one rank (the scores need no exchange), the
// CemMpc objective (no per-step Beta counts), and a population the one-workgroup select serves with its keys staged in LDS
bool folds_reduce(const cem_planner *h)
{
    const Dims &d = h->d;
    if (d.W != 1 || h->comm || h->cfg.variant != CEM_VARIANT_CEM) return false;
    bool cache = false;
    return resolve_select_mode(h->cfg.select_mode, d.N, d.k, (long long)d.H * d.A, h->sel_dyn_limit, false, &cache) == 1 && cache;
}

// fold_final: the select writes the plan's result itself (whole plans only: eps_out is known before the loop) — returns through
// *folded whether it did (only the one-workgroup kernel does), so the caller knows whether a final kernel is still needed
int enqueue_select(cem_planner *h, int it, bool fold_reduce, bool fold_final = false, bool have_eps_out = false, bool *folded = nullptr)
{
    (void)it;
    if (folded) *folded = false;
    const Dims &d = h->d; const Layout &l = h->lay; char *ws = h->ws;
    SelectParams p{}; p.scores = (const float *)(ws + l.scores_global); p.actions = (const float *)(ws + l.actions);
    p.musig = (float *)(ws + l.musig); p.ctrl = (CtrlBlock *)(ws + l.ctrl); p.elite_idx = (int32_t *)(ws + l.elite);
    p.N = d.N; p.k = d.k; p.HA = d.H * d.A; p.A = d.A; p.check_done = 1;
    p.smoothing = h->cfg.smoothing; p.one_minus_smoothing = h->cfg.one_minus_smoothing; p.threshold = h->cfg.stddev_threshold;
    p.stamps = (long long *)(ws + l.stamps) + 64;          // past tile 0's rollout stamps; written by -DCEM_STAMPS builds only
    if (fold_reduce) { p.ret = (const float *)(ws + l.returns); p.P = d.P; p.scores_w = (float *)(ws + l.scores_local); }   // (folds_reduce(): world 1, so local == global)
    size_t lds = (size_t)((d.k + 3) & ~3) * 4 + (size_t)2 * d.H * d.A * 4;
    // Large populations (the replicated select of a many-GPU plan) go through multi-workgroup kernels (cem_mpc.h select_mode):
    // the fused form (one launch, grid barriers) whenever all its ceil(N / 4096) workgroups are resident at once, the eight-launch
    // chain beyond that (more workgroups than CUs), the one-workgroup kernel for populations it still serves faster.  Measured
    // (profiles/r03_select_forms.txt, us per select at N = 8000 / 16000 / 40000 / 65536): one workgroup 27 / 44 / 111 / 171, chain
    // 68 / 71 / 78 / 83, fused 60 / 61 / 69 / 76 — a grid barrier is an atomic and a poll at the device coherence point (~4 us with
    // the XCDs' L2s not coherent with each other), hardly cheaper than a kernel boundary inside a graph: the cross-over with the
    // one-workgroup kernel stays near 24 000 keys.
    // The fused form's grid barriers need all G workgroups resident at once: G is held against what the RUNTIME says the device keeps
    // resident of this kernel (asked once at create: workgroups per CU x CUs), not against the CU count alone.  What it cannot see is
    // other work on the device — another stream, handle or process, a CU-masked queue: the form assumes a GPU that is otherwise idle
    // for the few microseconds of the launch; under contention a barrier times out (bounded polls) and the plan fails with
    // CEM_ERR_DEVICE rather than hanging.  select_mode 2 has no such assumption.
    const int G = (d.N + CEM_MS_KEYS - 1) / CEM_MS_KEYS;
    const bool can_fuse = G <= h->fused_resident && !h->fuse_banned;
    bool cache = false;
    const int mode = resolve_select_mode(h->cfg.select_mode, d.N, d.k, (long long)d.H * d.A, h->sel_dyn_limit, can_fuse, &cache);
    if (mode == 0) return CEM_ERR_UNSUPPORTED;          // (an explicit select_mode 1 on a device that grants less dynamic LDS than validate() assumed)
    size_t e0 = 0;
    if (h->timing) { e0 = h->ev_kind.size() * 2; h->ev_kind.push_back({(int)e0, 1}); hipEventRecord(get_event(h, e0), h->stream); }
    if (mode >= 2) {
        MSelParams m{}; m.scores = p.scores; m.actions = p.actions; m.musig = p.musig; m.ctrl = p.ctrl; m.elite_idx = p.elite_idx;
        m.hist = (uint32_t *)(ws + l.ms_hist); m.sel = (uint32_t *)(ws + l.ms_sel); m.wg_counts = (uint32_t *)(ws + l.ms_counts);
        m.bar = m.sel + 8;
        m.best_sc = (float *)(ws + l.ms_best_sc); m.best_ix = (int32_t *)(ws + l.ms_best_ix);
        m.part = (float *)(ws + l.ms_part); m.colmean = (float *)(ws + l.ms_colmean);
        m.N = d.N; m.k = d.k; m.HA = p.HA; m.A = d.A; m.check_done = 1; m.smoothing = p.smoothing; m.one_minus_smoothing = p.one_minus_smoothing; m.threshold = p.threshold;
        m.G = G; m.G2 = (d.k + CEM_MS_EPG - 1) / CEM_MS_EPG;
        // the histograms and the barrier counter (adjacent in the workspace) start at zero; within a plan the reduce kernel of the
        // same iteration has already cleared them (ReduceParams::zero), this memset covers a select called on its own
        if (!h->sel_zeroed) HIPCHK(hipMemsetAsync(m.hist, 0, 3 * CEM_MS_BINS * 4 + 256, h->stream));
        h->sel_zeroed = false;
        if (mode == 3) {
            hipLaunchKernelGGL(cem_msel_fused_kernel, dim3(m.G), dim3(1024), 0, h->stream, m);
            // returns at once unless a grid barrier of the launch above expired; then it redoes this iteration's select alone (cem_device.h)
            hipLaunchKernelGGL(cem_msel_solo_kernel, dim3(1), dim3(1024), 0, h->stream, m);
        } else {
            hipLaunchKernelGGL(cem_msel_hist_kernel<0>, dim3(m.G), dim3(1024), 0, h->stream, m);
            hipLaunchKernelGGL(cem_msel_hist_kernel<1>, dim3(m.G), dim3(1024), 0, h->stream, m);
            hipLaunchKernelGGL(cem_msel_hist_kernel<2>, dim3(m.G), dim3(1024), 0, h->stream, m);
            hipLaunchKernelGGL(cem_msel_count_kernel, dim3(m.G), dim3(1024), 0, h->stream, m);
            hipLaunchKernelGGL(cem_msel_compact_kernel, dim3(m.G), dim3(1024), 0, h->stream, m);
            hipLaunchKernelGGL(cem_msel_moments_kernel<0>, dim3(m.G2), dim3(256), 0, h->stream, m);
            hipLaunchKernelGGL(cem_msel_moments_kernel<1>, dim3(m.G2), dim3(256), 0, h->stream, m);
            hipLaunchKernelGGL(cem_msel_final_kernel, dim3(1), dim3(256), 0, h->stream, m);
        }
    } else {
        if (fold_final) {
            p.is_last = it == d.I - 1;
            p.result = h->d_h_result; p.result_dev = (uint32_t *)(ws + l.result); p.eps_out = have_eps_out ? (const float *)(ws + l.eps_out) : nullptr; p.noise_stddev = h->cfg.noise_stddev;
            if (folded) *folded = true;
        }
        if (cache) lds += (size_t)CEM_SEL_KWORDS(d.N) * 4;
        // (SafeCemMpc's scores have a crowd near -100: the instantiation that counts and ranks wave by wave; same results either way)
        if (cache && h->cfg.variant == CEM_VARIANT_SAFE) hipLaunchKernelGGL((cem_select_kernel<true, true>), dim3(1), dim3(1024), lds, h->stream, p);
        else if (cache) hipLaunchKernelGGL((cem_select_kernel<true, false>), dim3(1), dim3(1024), lds, h->stream, p);
        else hipLaunchKernelGGL((cem_select_kernel<false, false>), dim3(1), dim3(1024), lds, h->stream, p);
    }
    HIPCHK(hipGetLastError());
    if (h->timing) hipEventRecord(get_event(h, e0 + 1), h->stream);
    return CEM_OK;
}

// the one exchange step of an iteration: every rank's N/world local scores -> all N scores on every rank
int enqueue_exchange(cem_planner *h)
{
    if (!h->comm) return CEM_OK;
    Rccl *r = rccl(); if (!r) return CEM_ERR_COMM;
    const Dims &d = h->d; const Layout &l = h->lay;
    // world 1: scores_global aliases scores_local and the all-gather is the in-place form (sendbuff == recvbuff + rank * count)
    NCCLCHK(r->AllGather(h->ws + l.scores_local, h->ws + l.scores_global, (size_t)d.Nloc, kNcclFloat32, h->comm, h->stream));
    return CEM_OK;
}

int enqueue_end(cem_planner *h, bool have_eps_out)
{
    const Dims &d = h->d; const Layout &l = h->lay; char *ws = h->ws;
    FinalParams fp{}; fp.ctrl = (const CtrlBlock *)(ws + l.ctrl); fp.eps_out = have_eps_out ? (const float *)(ws + l.eps_out) : nullptr;
    fp.result = h->d_h_result; fp.result_dev = (uint32_t *)(ws + l.result); fp.A = d.A; fp.noise_stddev = h->cfg.noise_stddev;       // pinned host memory: no copy node behind the kernel
    hipLaunchKernelGGL(cem_final_kernel, dim3(1), dim3(64), 0, h->stream, fp);
    HIPCHK(hipGetLastError());
    return CEM_OK;
}

void collect_timing(cem_planner *h)
{
    h->roll_ms = 0.f; h->sel_ms = 0.f; h->red_ms = 0.f; h->samp_ms = 0.f; h->roll_n = 0;
    for (auto &ek : h->ev_kind) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev[ek.first], h->ev[ek.first + 1]) == hipSuccess) {
            if (ek.second == 0) { h->roll_ms += ms; h->roll_n++; }
            else if (ek.second == 1) h->sel_ms += ms;
            else if (ek.second == 2) h->red_ms += ms;
            else h->samp_ms += ms;
        }
    }
    h->ev_kind.clear();
}

void stage_ctrl(cem_planner *h, const float *state, uint64_t seed, uint64_t call)
{
    CtrlBlock *c = h->h_ctrl;
    c->seed_lo = (uint32_t)seed; c->seed_hi = (uint32_t)(seed >> 32); c->call_lo = (uint32_t)call; c->call_hi = (uint32_t)(call >> 32);
    c->done = 0; c->iters = 0; c->fault = 0; c->best_score = -std::numeric_limits<float>::infinity();
    for (int f = 0; f < CEM_U; ++f) c->state[f] = f < h->d.O ? state[f] : 0.f;
    for (int a = 0; a < 32; ++a) c->best[a] = 0.f;
    c->seq = ++h->plan_seq;                             // echoed into result[36] by the kernel that completes the plan
    c->inject = h->inject_next; h->inject_next = 0;     // (test hook, cem_planner_inject_fault)
}

// The plan is queued: wait for its result.  The kernel that completes it stores the plan counter into pinned host memory after
// everything else of the result, so the host watches that word — a few hundred nanoseconds after the store — instead of going
// through hipStreamSynchronize (an interrupt / yield path that took ~10 us of a 1.9-ms plan).  Bounded: after 100 ms of polling the
// ordinary synchronisation takes over.
// the block is complete when it carries this plan's counter AND its checksum holds (device -> host writes arrive in no particular order)
bool result_landed(const cem_planner *h)
{
    const volatile uint32_t *r = reinterpret_cast<const volatile uint32_t *>(h->h_result);
    if (r[36] != h->plan_seq) return false;
    return cem_result_checksum(r, h->plan_seq) == r[37];      // position dependent: stale words cannot cancel (cem_device.h)
}

int wait_result(cem_planner *h)
{
    // CEM_NO_POLL: a host that would rather sleep than spin a core for the ~2 ms of a plan goes straight to the stream synchronisation
    static const bool no_poll = std::getenv("CEM_NO_POLL") != nullptr;
    if (!no_poll) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 1;; ++spin) {
            if (result_landed(h)) { std::atomic_thread_fence(std::memory_order_acquire); return CEM_OK; }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if ((spin & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(100)) break;   // a plan that long is not latency-critical; a faulted device never writes the block
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return CEM_OK;
}

int read_result(cem_planner *h, float *action_out, float *best_score_out, int32_t *iters_out)
{
    // (after a stream synchronisation the block has landed; the check costs nothing and a short wait covers a write still in flight)
    for (int spin = 0; spin < 2000000 && !result_landed(h); ++spin) { }
    if (!result_landed(h)) return CEM_ERR_DEVICE;
    if (action_out) std::memcpy(action_out, h->h_result, h->d.A * 4);
    if (best_score_out) *best_score_out = h->h_result[32];
    if (iters_out) *iters_out = reinterpret_cast<int32_t *>(h->h_result)[33];
    const int32_t fault = reinterpret_cast<int32_t *>(h->h_result)[35];                 // CtrlBlock::fault
    if (fault & CEM_FAULT_RECOVERED) {
        // A grid barrier of the fused select expired (its workgroups were not all resident: another stream, handle or process held
        // CUs) and cem_msel_solo_kernel redid that iteration's select in stream order: the plan is valid, with select_mode 2's bits.
        // This handle stops fusing: the next plan re-captures its graph on the eight-launch chain, which has no residency assumption.
        if (!h->fuse_banned) {
            std::fprintf(stderr, "cem_mpc: a grid barrier of the fused select timed out (GPU shared with other work?); that iteration's select was redone "
                                 "by the recovery kernel and this handle uses the multi-launch select (select_mode 2) from now on\n");
            h->fuse_banned = true;
            if (hipStreamSynchronize(h->stream) != hipSuccess) (void)hipGetLastError();      // the graph may still be draining behind the polled result
            if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
            if (h->graph) { hipGraphDestroy(h->graph); h->graph = nullptr; }
            h->graph_ready = false;
        }
    }
    return (fault & (CEM_FAULT_SEGMENT | CEM_FAULT_BARRIER)) ? CEM_ERR_DEVICE : CEM_OK;     // a kernel gave up and nothing made up for it
}

}  // namespace

extern "C" {

int cem_plan_begin(cem_planner_t *h, const float *state, uint64_t seed, uint64_t call, const float *eps_act_dev, const float *eps_model_dev)
{
    if (!h || !state) return CEM_ERR_INVALID_ARG;
    if (!h->have_weights) return CEM_ERR_NO_WEIGHTS;
    if ((eps_act_dev == nullptr) != (eps_model_dev == nullptr) && h->cfg.sampling_propagation) return CEM_ERR_INVALID_ARG;
    stage_ctrl(h, state, seed, call);
    h->eps_act = eps_act_dev; h->eps_model = eps_model_dev;
    h->ev_kind.clear();
    int st = enqueue_begin(h); if (st) return st;
    h->in_plan = true;
    return CEM_OK;
}

int cem_plan_rollout(cem_planner_t *h, int32_t it)
{
    if (!h) return CEM_ERR_INVALID_ARG;
    if (!h->in_plan) return CEM_ERR_STATE;
    if (it < 0 || it >= h->d.I) return CEM_ERR_INVALID_ARG;
    return enqueue_rollout(h, it, false);      // the stepwise form always leaves the scores in scores_local (the caller may exchange them)
}

int cem_plan_select(cem_planner_t *h, int32_t it)
{
    if (!h) return CEM_ERR_INVALID_ARG;
    if (!h->in_plan) return CEM_ERR_STATE;
    return enqueue_select(h, it, false);
}

int cem_plan_end(cem_planner_t *h, const float *eps_out_host, float *action_out, float *best_score_out, int32_t *iters_out)
{
    if (!h) return CEM_ERR_INVALID_ARG;
    if (!h->in_plan) return CEM_ERR_STATE;
    h->in_plan = false;
    if (eps_out_host) HIPCHK(hipMemcpyAsync(h->ws + h->lay.eps_out, eps_out_host, h->d.A * 4, hipMemcpyHostToDevice, h->stream));
    int st = enqueue_end(h, eps_out_host != nullptr); if (st) return st;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->timing) collect_timing(h);
    return read_result(h, action_out, best_score_out, iters_out);
}

int cem_planner_plan(cem_planner_t *h, const float *state, uint64_t seed, uint64_t call, const float *eps_act_dev,
                     const float *eps_model_dev, const float *eps_out_host, float *action_out, float *best_score_out, int32_t *iters_out)
{
    if (!h || !state) return CEM_ERR_INVALID_ARG;
    if (!h->have_weights) return CEM_ERR_NO_WEIGHTS;
    if (h->d.W != 1 && !h->comm) return CEM_ERR_STATE;   // sharded ranks without cem_planner_comm_init use the stepwise calls around their own collective
    // With a communicator the first plan runs eagerly: RCCL finishes its lazy set-up (buffers, kernels) outside any capture.
    const bool graphable = h->cfg.use_graph && !eps_act_dev && !eps_model_dev && !eps_out_host && !h->timing && !h->graph_failed &&
                           (!h->comm || h->plans_since_comm > 0);
    if (h->comm) h->plans_since_comm++;
    if (graphable) {
        stage_ctrl(h, state, seed, call);
        if (!h->graph_ready) {
            h->eps_act = h->eps_model = nullptr;
            // relaxed mode: RCCL may touch the runtime from its proxy thread while this thread captures
            HIPCHK(hipStreamBeginCapture(h->stream, h->comm ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal));
            int st = enqueue_begin(h);
            const bool fold = folds_reduce(h);
            bool final_folded = false;
            for (int it = 0; it < h->d.I && !st; ++it) {
                st = enqueue_rollout(h, it, fold);
                if (!st) st = enqueue_exchange(h);
                if (!st) st = enqueue_select(h, it, fold, true, false, &final_folded);
            }
            if (!st && !final_folded) st = enqueue_end(h, false);
            hipError_t ce = hipStreamEndCapture(h->stream, &h->graph);
            if (!st && ce == hipSuccess) ce = hipGraphInstantiate(&h->gexec, h->graph, nullptr, nullptr, 0);
            if (st || ce != hipSuccess) {
                if (!h->comm) { if (st) return st; HIPCHK(ce); }
                // a captured collective is not something every RCCL / runtime pair supports: fall back to eager launches for good
                (void)hipGetLastError();
                if (h->graph) { hipGraphDestroy(h->graph); h->graph = nullptr; }
                h->gexec = nullptr; h->graph_failed = true;
            } else {
                h->graph_ready = true;
            }
        }
        if (h->graph_ready) {
            HIPCHK(hipGraphLaunch(h->gexec, h->stream));
            { const int ws_ = wait_result(h); if (ws_) return ws_; }
            return read_result(h, action_out, best_score_out, iters_out);
        }
    }
    int st = cem_plan_begin(h, state, seed, call, eps_act_dev, eps_model_dev); if (st) return st;
    const bool fold = folds_reduce(h);                  // the same launches as the captured form
    if (eps_out_host) HIPCHK(hipMemcpyAsync(h->ws + h->lay.eps_out, eps_out_host, h->d.A * 4, hipMemcpyHostToDevice, h->stream));
    bool final_folded = false;
    for (int it = 0; it < h->d.I; ++it) {
        st = enqueue_rollout(h, it, fold); if (st) { h->in_plan = false; return st; }
        st = enqueue_exchange(h); if (st) { h->in_plan = false; return st; }
        st = enqueue_select(h, it, fold, true, eps_out_host != nullptr, &final_folded); if (st) { h->in_plan = false; return st; }
    }
    h->in_plan = false;
    if (!final_folded) { st = enqueue_end(h, eps_out_host != nullptr); if (st) return st; }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->timing) collect_timing(h);
    return read_result(h, action_out, best_score_out, iters_out);
}

int cem_comm_unique_id(void *id_out)
{
    if (!id_out) return CEM_ERR_INVALID_ARG;
    Rccl *r = rccl(); if (!r) return CEM_ERR_COMM;
    CemNcclId id;
    NCCLCHK(r->GetUniqueId(&id));
    std::memcpy(id_out, id.internal, CEM_COMM_ID_BYTES);
    return CEM_OK;
}

int cem_planner_comm_init(cem_planner_t *h, const void *id, int32_t n_ranks, int32_t rank)
{
    if (!h || !id) return CEM_ERR_INVALID_ARG;
    if (n_ranks != h->d.W || rank != h->d.R) return CEM_ERR_INVALID_ARG;       // the communicator IS the candidate sharding of this handle
    if (h->in_plan) return CEM_ERR_STATE;
    Rccl *r = rccl(); if (!r) return CEM_ERR_COMM;
    if (h->comm) { r->CommDestroy(h->comm); h->comm = nullptr; }
    CemNcclId nid; std::memcpy(nid.internal, id, CEM_COMM_ID_BYTES);
    void *comm = nullptr;
    NCCLCHK(r->CommInitRank(&comm, n_ranks, nid, rank));
    h->comm = comm; h->plans_since_comm = 0;
    // a graph captured without the collective (world 1) no longer describes the plan
    if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    if (h->graph) { hipGraphDestroy(h->graph); h->graph = nullptr; }
    h->graph_ready = false; h->graph_failed = false;
    return CEM_OK;
}

int cem_planner_comm_ranks(const cem_planner_t *h, int32_t *n_ranks_out)
{
    if (!h || !n_ranks_out) return CEM_ERR_INVALID_ARG;
    *n_ranks_out = 0;
    if (!h->comm) return CEM_OK;                     // no communicator: 0
    Rccl *r = rccl(); if (!r || !r->CommCount) return CEM_ERR_COMM;
    int n = 0;
    NCCLCHK(r->CommCount(h->comm, &n));
    *n_ranks_out = n;
    return CEM_OK;
}

int cem_planner_comm_destroy(cem_planner_t *h)
{
    if (!h) return CEM_ERR_INVALID_ARG;
    if (h->in_plan) return CEM_ERR_STATE;
    if (h->comm) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (Rccl *r = rccl()) r->CommDestroy(h->comm);
        h->comm = nullptr;
        if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
        if (h->graph) { hipGraphDestroy(h->graph); h->graph = nullptr; }
        h->graph_ready = false;
    }
    return CEM_OK;
}

int cem_planner_graph_status(const cem_planner_t *h, int32_t *status_out)
{
    if (!h || !status_out) return CEM_ERR_INVALID_ARG;
    *status_out = h->graph_ready ? 1 : (h->graph_failed ? 2 : 0);
    return CEM_OK;
}

int cem_planner_launches_per_iteration(const cem_planner_t *h, int32_t *launches_out)
{
    if (!h || !launches_out) return CEM_ERR_INVALID_ARG;
    const Dims &d = h->d;
    const int G = (d.N + CEM_MS_KEYS - 1) / CEM_MS_KEYS;
    const int mode = resolve_select_mode(h->cfg.select_mode, d.N, d.k, (long long)d.H * d.A, h->sel_dyn_limit, G <= h->fused_resident && !h->fuse_banned, nullptr);
    *launches_out = 1 + (h->sample_in_rollout ? 0 : 1) + (folds_reduce(h) ? 0 : 1) + (mode == 2 ? 8 : (mode == 3 ? 2 : 1));
    return CEM_OK;
}

int cem_plan_exchange(cem_planner_t *h)
{
    if (!h) return CEM_ERR_INVALID_ARG;
    if (!h->in_plan || !h->comm) return CEM_ERR_STATE;
    return enqueue_exchange(h);
}

// grow-only device scratch of the standalone ops, cached on the handle (a hipMalloc / hipFree pair per call is a
// device-wide synchronisation each)
static int ensure_scratch(cem_planner *h, size_t bytes)
{
    if (bytes <= h->scratch_bytes) return CEM_OK;
    if (h->scratch) { HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(hipFree(h->scratch)); h->scratch = nullptr; h->scratch_bytes = 0; }
    const size_t want = align256(bytes + bytes / 2);
    HIPCHK(hipMalloc((void **)&h->scratch, want));
    h->scratch_bytes = want;
    return CEM_OK;
}

// Philox key of a standalone call: only the 16-byte (seed, call) prefix of the device control block is written, and never
// while a stepwise plan is in flight (its best-so-far / early-stop state and its own key live in the same block)
static int upload_key(cem_planner *h, uint64_t seed, uint64_t call)
{
    CtrlBlock *c = h->h_ctrl;
    c->seed_lo = (uint32_t)seed; c->seed_hi = (uint32_t)(seed >> 32); c->call_lo = (uint32_t)call; c->call_hi = (uint32_t)(call >> 32);
    HIPCHK(hipMemcpyAsync(h->ws + h->lay.ctrl, c, 16, hipMemcpyHostToDevice, h->stream));
    return CEM_OK;
}

int cem_unfold_sequences(cem_planner_t *h, const float *s0_dev, const float *actions_dev, int32_t n_rows, int32_t horizon,
                         const float *eps_model_dev, uint64_t seed, uint64_t call, float *traj_out_dev, float *mu_out_dev, float *sd_out_dev)
{
    if (!h || !s0_dev || !actions_dev || n_rows < 1 || horizon < 1 || horizon > 65535) return CEM_ERR_INVALID_ARG;
    if (!h->have_weights) return CEM_ERR_NO_WEIGHTS;
    if (h->in_plan) return CEM_ERR_STATE;
    const Dims &d = h->d;
    if (n_rows % d.E != 0) return CEM_ERR_SPLIT;
    if ((long long)n_rows * (horizon + 1) * d.O > 0x7fffffff00ll) return CEM_ERR_UNSUPPORTED;
    const int chunk = n_rows / d.E;
    const int rc = d.wide ? 1 : (d.split ? (n_rows >= 256 * 32 ? 2 : 1) : (n_rows >= 256 * 64 ? 4 : (n_rows >= 256 * 32 ? 2 : 1)));
    std::vector<Tile6> tiles;
    for (int m = 0; m < d.E; ++m)
        for (int r = m * chunk; r < (m + 1) * chunk; r += 16 * rc) {
            Tile6 t; t.v[0] = r; t.v[1] = std::min(16 * rc, (m + 1) * chunk - r); t.v[2] = m; t.v[3] = r; t.v[4] = r; t.v[5] = r;
            tiles.push_back(t);
        }
    const size_t tile_bytes = align256(tiles.size() * sizeof(Tile6));
    int st = ensure_scratch(h, tile_bytes + (size_t)n_rows * 4); if (st) return st;
    TileDesc *dt = (TileDesc *)h->scratch; float *ret = (float *)(h->scratch + tile_bytes);
    // the tile list is pageable host memory: the copy has returned from it when hipMemcpyAsync returns only if it was staged;
    // synchronise before `tiles` goes out of scope (below, with the kernel)
    HIPCHK(hipMemcpyAsync(dt, tiles.data(), tiles.size() * sizeof(Tile6), hipMemcpyHostToDevice, h->stream));
    st = upload_key(h, seed, call); if (st) return st;
    RolloutParams rp; fill_rollout_common(h, rp);
    rp.tiles = dt; rp.s0 = s0_dev; rp.actions = actions_dev; rp.eps_model = eps_model_dev; rp.ret = ret; rp.costs = nullptr;
    rp.traj = traj_out_dev; rp.mu_out = mu_out_dev; rp.sd_out = sd_out_dev;
    rp.H = horizon; rp.Bloc = n_rows; rp.Btot = n_rows; rp.it = 0; rp.variant = 0; rp.check_done = 0;
    hipError_t e = d.wide ? launch_rollout_wide(h, rp, (int)tiles.size(), 1)
                 : d.split ? launch_rollout_split(rc, d.NFW, 1, rp, (int)tiles.size(), h->stream)
                           : launch_rollout<1>(rc, d.NFW, rp, (int)tiles.size(), h->stream);
    hipError_t e2 = hipStreamSynchronize(h->stream);
    HIPCHK(e); HIPCHK(e2);
    return CEM_OK;
}

int cem_compute_objective(cem_planner_t *h, const float *traj_dev, int32_t n_rows, int32_t horizon, float *scores_out_dev)
{
    if (!h || !traj_dev || !scores_out_dev || n_rows < 1 || horizon < 1 || horizon > 65535) return CEM_ERR_INVALID_ARG;
    const Dims &d = h->d;
    if (n_rows % d.P != 0) return CEM_ERR_INVALID_ARG;                 // reshape(cum, (particles, -1)) would raise (mpc_policy.py:38)
    if ((long long)n_rows * horizon > 0x7fffffffll) return CEM_ERR_UNSUPPORTED;
    const bool safe = h->cfg.variant == CEM_VARIANT_SAFE;
    const size_t ret_bytes = align256((size_t)n_rows * 4);
    int st = ensure_scratch(h, ret_bytes + (safe ? (size_t)n_rows * horizon : 0)); if (st) return st;
    ObjectiveParams op{}; op.traj = traj_dev; op.ret = (float *)h->scratch; op.costs = safe ? (uint8_t *)(h->scratch + ret_bytes) : nullptr;
    op.B = n_rows; op.H = horizon; op.O = d.O; op.variant = h->cfg.variant; op.sc = h->sc;
    hipLaunchKernelGGL(cem_objective_kernel, dim3((unsigned)(((size_t)n_rows * 16 + 255) / 256)), dim3(256), 0, h->stream, op);
    HIPCHK(hipGetLastError());
    ReduceParams qp{}; qp.ret = op.ret; qp.costs = op.costs; qp.scores = scores_out_dev; qp.ctrl = (const CtrlBlock *)(h->ws + h->lay.ctrl);
    qp.Nloc = n_rows / d.P; qp.P = d.P; qp.H = horizon; qp.variant = h->cfg.variant; qp.check_done = 0;
    qp.alpha = h->alpha; qp.beta = h->beta; qp.thr = h->cfg.posterior_mean_threashold;
    hipLaunchKernelGGL(cem_reduce_kernel, dim3((qp.Nloc + 63) / 64), dim3(CEM_REDUCE_THREADS), 0, h->stream, qp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return CEM_OK;
}

static int scorer_op(cem_planner *h, const float *obs, const float *next_obs, int32_t n, float *out, uint8_t *flag, int what)
{
    ScorerOpParams sp{}; sp.obs = obs; sp.next_obs = next_obs; sp.out = out; sp.flag = flag; sp.n = n; sp.O = h->d.O; sp.what = what; sp.sc = h->sc;
    hipLaunchKernelGGL(cem_scorer_kernel, dim3((unsigned)(((size_t)n * 16 + 255) / 256)), dim3(256), 0, h->stream, sp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return CEM_OK;
}

int cem_scorer_reward(cem_planner_t *h, const float *obs_dev, const float *next_obs_dev, int32_t n, float *reward_out_dev,
                      uint8_t *goal_achieved_out_dev)
{
    if (!h || !obs_dev || !next_obs_dev || !reward_out_dev || n < 1) return CEM_ERR_INVALID_ARG;
    return scorer_op(h, obs_dev, next_obs_dev, n, reward_out_dev, goal_achieved_out_dev, 0);
}

int cem_scorer_cost(cem_planner_t *h, const float *obs_dev, int32_t n, float *cost_out_dev)
{
    if (!h || !obs_dev || !cost_out_dev || n < 1) return CEM_ERR_INVALID_ARG;
    return scorer_op(h, obs_dev, nullptr, n, cost_out_dev, nullptr, 1);
}

int cem_fill_noise(cem_planner_t *h, uint64_t seed, uint64_t call, float *eps_act_dev, float *eps_model_dev, float *eps_out_dev)
{
    if (!h) return CEM_ERR_INVALID_ARG;
    if (h->in_plan) return CEM_ERR_STATE;
    const Dims &d = h->d;
    int st = upload_key(h, seed, call); if (st) return st;
    FillParams fp{}; fp.eps_act = eps_act_dev; fp.eps_model = eps_model_dev; fp.eps_out = eps_out_dev;
    fp.ctrl = (const CtrlBlock *)(h->ws + h->lay.ctrl); fp.I = d.I; fp.N = d.N; fp.H = d.H; fp.A = d.A; fp.B = d.Btot; fp.O = d.O;
    hipLaunchKernelGGL(cem_fill_noise_kernel, dim3(2048), dim3(256), 0, h->stream, fp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return CEM_OK;
}

int cem_philox_words(cem_planner_t *h, uint64_t seed, uint64_t call, uint32_t stream, uint32_t iteration, uint32_t t, uint32_t sub,
                     uint32_t idx0, uint32_t n, uint32_t *words_out_dev)
{
    if (!h || !words_out_dev || stream > 2 || iteration > 65535 || t > 65535 || sub > 65535) return CEM_ERR_INVALID_ARG;
    if (h->in_plan) return CEM_ERR_STATE;
    if (n == 0) return CEM_OK;
    int st = upload_key(h, seed, call); if (st) return st;
    WordsParams wp{}; wp.out = words_out_dev; wp.ctrl = (const CtrlBlock *)(h->ws + h->lay.ctrl);
    wp.stream = stream; wp.it = iteration; wp.t = t; wp.sub = sub; wp.idx0 = idx0; wp.n = n;
    hipLaunchKernelGGL(cem_philox_words_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, wp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return CEM_OK;
}

int cem_planner_select_mode(const cem_planner_t *h, int32_t *mode_out)
{
    if (!h || !mode_out) return CEM_ERR_INVALID_ARG;
    const Dims &d = h->d;
    const int G = (d.N + CEM_MS_KEYS - 1) / CEM_MS_KEYS;
    *mode_out = resolve_select_mode(h->cfg.select_mode, d.N, d.k, (long long)d.H * d.A, h->sel_dyn_limit, G <= h->fused_resident && !h->fuse_banned, nullptr);
    return CEM_OK;
}

int cem_planner_inject_fault(cem_planner_t *h, int32_t kind)
{
    if (!h || kind != 1) return CEM_ERR_INVALID_ARG;
    if (h->in_plan) return CEM_ERR_STATE;
    h->inject_next = 1u;            // staged with the next plan's control block (stage_ctrl) and consumed by it
    return CEM_OK;
}

int cem_planner_set_timing(cem_planner_t *h, int32_t enable)
{
    if (!h) return CEM_ERR_INVALID_ARG;
    h->timing = enable != 0;
    return CEM_OK;
}

int cem_planner_last_timing(cem_planner_t *h, float *rollout_ms_total, int32_t *rollout_launches, float *select_ms_total)
{
    if (!h) return CEM_ERR_INVALID_ARG;
    if (rollout_ms_total) *rollout_ms_total = h->roll_ms;
    if (rollout_launches) *rollout_launches = h->roll_n;
    if (select_ms_total) *select_ms_total = h->sel_ms;
    return CEM_OK;
}

int cem_planner_last_timing_detail(cem_planner_t *h, float *reduce_ms_total, float *sampler_ms_total)
{
    if (!h) return CEM_ERR_INVALID_ARG;
    if (reduce_ms_total) *reduce_ms_total = h->red_ms;
    if (sampler_ms_total) *sampler_ms_total = h->samp_ms;
    return CEM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// training
// ---------------------------------------------------------------------------------------------------------
struct cem_trainer {
    cem_train_config_t cfg;
    char *ws; hipStream_t stream; bool own_stream;
    size_t nat, scratch_pm;
    size_t oW, oM, oV, oG, oS, oL, oP, oT, total;
    bool tile_kernel;
    uint32_t steps_done;                     // training steps since create: the step index of the Dropout masks
    float *eval_part; size_t eval_part_floats;     // per-chunk loss partials of a one-launch validation pass (grown on demand, kept)
};

namespace {
int validate_train(const cem_train_config_t *c)
{
    if (!c || c->abi_version != CEM_ABI_VERSION) return CEM_ERR_INVALID_ARG;
    if (c->inputs_dim < 1 || c->outputs_dim < 1 || c->n_layers < 1 || c->ensemble_size < 1 || c->batch_size < 1) return CEM_ERR_INVALID_ARG;
    if (c->units < 1 || c->activation < CEM_ACT_RELU || c->activation > CEM_ACT_GELU) return CEM_ERR_INVALID_ARG;
    if (!(c->dropout_rate >= 0.f && c->dropout_rate < 1.f)) return CEM_ERR_INVALID_ARG;
    if (c->units > CEM_TWIDE || c->inputs_dim > CEM_U || c->outputs_dim > CEM_U || c->batch_size > CEM_TB) return CEM_ERR_UNSUPPORTED;
    return CEM_OK;
}
size_t train_nat(const cem_train_config_t *c)
{
    return (size_t)c->inputs_dim * c->units + c->units + (size_t)(c->n_layers - 1) * ((size_t)c->units * c->units + c->units) +
           2 * ((size_t)c->units * c->outputs_dim + c->outputs_dim);
}
void train_layout(cem_trainer *t)
{
    const cem_train_config_t &c = t->cfg;
    t->nat = train_nat(&c);
    // per (member, row part) workgroup: the input, L hidden outputs, seven head / loss / gradient matrices — and, for swish / gelu, the L kept
    // pre-activations the backward gate of a non-monotone activation needs (cem_train.h: GemmEpi::outz)
    t->scratch_pm = (size_t)(c.n_layers + 8 + (CEM_ACT_NEEDS_Z(c.activation) ? c.n_layers : 0)) * CEM_TROWS * (c.units > CEM_TS ? CEM_TWIDE : CEM_TS);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    t->oW = take(t->nat * c.ensemble_size * 4); t->oM = take(t->nat * c.ensemble_size * 4); t->oV = take(t->nat * c.ensemble_size * 4);
    t->oG = take(((t->nat * c.ensemble_size + 3) & ~(size_t)3) * CEM_TPARTS * 4); t->oS = take(t->scratch_pm * c.ensemble_size * CEM_TPARTS * 4);
    t->oL = take((size_t)c.ensemble_size * 4); t->oP = take((size_t)c.ensemble_size * CEM_TPARTS * 2 * 4);
    t->oT = take(32 * sizeof(long long));            // phase stamps of -DCEM_STAMPS diagnostic builds: the LAST 256 B of the workspace
    t->total = o;
}
// the training step: the tile kernel (cem_train_tile.h; one instantiation per layer count up to CEM_TT_MAXL), otherwise the
// GEMM-by-GEMM kernel (cem_train.h), which takes any depth
template <int L>
hipError_t tile_kernel_lds(size_t lds)
{
    return lds > 48 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(&cem_train_tile_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : hipSuccess;
}

void launch_train_step(const cem_trainer *t, const TrainParams &p)
{
    const size_t lds = (size_t)(t->cfg.n_layers + 5) * CEM_TT_NB * CEM_TT_BLK;
    const dim3 grid(t->cfg.ensemble_size * CEM_TPARTS, t->tile_kernel ? (p.Bt + p.chunk - 1) / p.chunk : 1);
    if (!t->tile_kernel) { hipLaunchKernelGGL(cem_train_step_kernel, grid, dim3(CEM_TNT), 0, t->stream, p); return; }
    switch (t->cfg.n_layers) {
#define CEM_CASE(LL) case LL: hipLaunchKernelGGL(cem_train_tile_kernel<LL>, grid, dim3(64 * CEM_TT_WAVES), lds, t->stream, p); break;
    CEM_CASE(1) CEM_CASE(2) CEM_CASE(3) CEM_CASE(4) CEM_CASE(5) CEM_CASE(6)
#undef CEM_CASE
    }
}

void fill_train_params(const cem_trainer *t, TrainParams &p)
{
    const cem_train_config_t &c = t->cfg;
    std::memset(&p, 0, sizeof(p));
    p.W = (float *)(t->ws + t->oW); p.Mo = (float *)(t->ws + t->oM); p.Vo = (float *)(t->ws + t->oV);
    p.grad = (float *)(t->ws + t->oG); p.scratch = (float *)(t->ws + t->oS); p.loss_part = (float *)(t->ws + t->oP);
    p.D = c.inputs_dim; p.O = c.outputs_dim; p.U = c.units; p.L = c.n_layers; p.E = c.ensemble_size;
    p.nat = (uint32_t)t->nat; p.scratch_per_member = (uint32_t)t->scratch_pm;
    p.gpart = (uint32_t)((t->nat * c.ensemble_size + 3) & ~(size_t)3);
    p.ts = c.units > CEM_TS ? CEM_TWIDE : CEM_TS;
    p.beta1 = c.beta1; p.beta2 = c.beta2; p.eps = c.epsilon; p.clip = c.clipvalue; p.act = c.activation;
    if (c.dropout_rate > 0.f) {
        p.drop_thresh = (uint32_t)std::min(4294967295.0, std::floor((double)c.dropout_rate * 4294967296.0));
        p.drop_keep = 1.0f - c.dropout_rate; p.drop_scale = 1.0f / p.drop_keep;
        p.drop_k0 = c.dropout_seed_lo; p.drop_k1 = c.dropout_seed_hi; p.drop_step = t->steps_done;
    }
    p.stamps = (long long *)(t->ws + t->oT);
}
}  // namespace

extern "C" {

size_t cem_trainer_workspace_bytes(const cem_train_config_t *cfg)
{
    if (validate_train(cfg)) return 0;
    cem_trainer t; t.cfg = *cfg; train_layout(&t); return t.total;
}
size_t cem_trainer_blob_floats(const cem_train_config_t *cfg) { return validate_train(cfg) ? 0 : train_nat(cfg) * cfg->ensemble_size; }

int cem_trainer_create(const cem_train_config_t *cfg, void *workspace, size_t workspace_bytes, void *hip_stream, cem_trainer_t **out)
{
    int st = validate_train(cfg); if (st) return st;
    if (!workspace || !out) return CEM_ERR_INVALID_ARG;
    cem_trainer *t = new (std::nothrow) cem_trainer();
    if (!t) return CEM_ERR_INVALID_ARG;
    t->cfg = *cfg; train_layout(t);
    if (workspace_bytes < t->total || ((uintptr_t)workspace & 255)) { delete t; return CEM_ERR_WORKSPACE; }
    t->ws = (char *)workspace; t->stream = (hipStream_t)hip_stream; t->own_stream = false;
    if (!t->stream) {
        if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) { g_last_hip = (int)hipGetLastError(); delete t; return CEM_ERR_HIP; }
        t->own_stream = true;
    }
    if (hipMemsetAsync(t->ws, 0, t->total, t->stream) != hipSuccess || hipStreamSynchronize(t->stream) != hipSuccess) {
        g_last_hip = (int)hipGetLastError(); delete t; return CEM_ERR_HIP;
    }
    {   // the tile kernel keeps every layer's activations in LDS: (n_layers + 5) x 8 KB, beyond 48 KB only with the runtime's leave
        const size_t lds = (size_t)(cfg->n_layers + 5) * CEM_TT_NB * CEM_TT_BLK;
        t->tile_kernel = cfg->n_layers <= CEM_TT_MAXL && cfg->units <= CEM_U && cfg->activation == CEM_ACT_RELU && cfg->dropout_rate == 0.f &&
                         std::getenv("CEM_TRAIN_GEMM_KERNEL") == nullptr;   // the tile kernel is 8 blocks wide and relu only
        hipError_t e = hipSuccess;
        if (t->tile_kernel) switch (cfg->n_layers) {
#define CEM_CASE(LL) case LL: e = tile_kernel_lds<LL>(lds); break;
            CEM_CASE(1) CEM_CASE(2) CEM_CASE(3) CEM_CASE(4) CEM_CASE(5) CEM_CASE(6)
#undef CEM_CASE
        }
        if (e != hipSuccess) { (void)hipGetLastError(); t->tile_kernel = false; }
    }
    *out = t;
    return CEM_OK;
}

int cem_trainer_destroy(cem_trainer_t *t)
{
    if (!t) return CEM_ERR_INVALID_ARG;
    if (t->own_stream) hipStreamDestroy(t->stream);
    if (t->eval_part) hipFree(t->eval_part);
    delete t;
    return CEM_OK;
}

int cem_trainer_set_state(cem_trainer_t *t, const float *weights, const float *m, const float *v)
{
    if (!t || !weights) return CEM_ERR_INVALID_ARG;
    const size_t bytes = t->nat * t->cfg.ensemble_size * 4;
    HIPCHK(hipMemcpyAsync(t->ws + t->oW, weights, bytes, hipMemcpyHostToDevice, t->stream));
    if (m) HIPCHK(hipMemcpyAsync(t->ws + t->oM, m, bytes, hipMemcpyHostToDevice, t->stream)); else HIPCHK(hipMemsetAsync(t->ws + t->oM, 0, bytes, t->stream));
    if (v) HIPCHK(hipMemcpyAsync(t->ws + t->oV, v, bytes, hipMemcpyHostToDevice, t->stream)); else HIPCHK(hipMemsetAsync(t->ws + t->oV, 0, bytes, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    // (steps_done is NOT reset: the Dropout masks are a function of the step index since create, so a trainer whose weights are
    // re-staged from outside — every fit of an agent loop — goes on to fresh masks instead of replaying the first ones)
    return CEM_OK;
}

int cem_trainer_get_state(cem_trainer_t *t, float *weights, float *m, float *v)
{
    if (!t) return CEM_ERR_INVALID_ARG;
    const size_t bytes = t->nat * t->cfg.ensemble_size * 4;
    if (weights) HIPCHK(hipMemcpyAsync(weights, t->ws + t->oW, bytes, hipMemcpyDeviceToHost, t->stream));
    if (m) HIPCHK(hipMemcpyAsync(m, t->ws + t->oM, bytes, hipMemcpyDeviceToHost, t->stream));
    if (v) HIPCHK(hipMemcpyAsync(v, t->ws + t->oV, bytes, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    return CEM_OK;
}

int cem_trainer_step(cem_trainer_t *t, const float *x_dev, const float *y_dev, const int32_t *perm_dev, int32_t nperm, int32_t offset,
                     int32_t bt, float lr_t, float *loss_dev)
{
    if (!t || !x_dev || !y_dev || !loss_dev || bt < 1 || bt > t->cfg.batch_size || offset < 0) return CEM_ERR_INVALID_ARG;
    if (perm_dev && offset + bt > nperm) return CEM_ERR_INVALID_ARG;
    TrainParams p; fill_train_params(t, p);
    p.x = x_dev; p.y = y_dev; p.perm = perm_dev; p.nperm = nperm; p.offset = offset; p.Bt = bt; p.chunk = bt; p.lr_t = lr_t; p.loss_out = loss_dev; p.train = 1;
    launch_train_step(t, p);
    t->steps_done += 1;
    const size_t n4 = (size_t)p.E * p.nat / 4;
    const unsigned adam_grid = (unsigned)std::min<size_t>(std::max<size_t>((n4 + 255) / 256, 1), 2048);
    hipLaunchKernelGGL(cem_adam_kernel, dim3(adam_grid), dim3(256), 0, t->stream, p);
    HIPCHK(hipGetLastError());
    return CEM_OK;
}

int cem_trainer_steps(cem_trainer_t *t, const float *x_dev, const float *y_dev, const int32_t *perm_dev, int32_t nperm, int32_t n_steps,
                      const int32_t *offsets, const int32_t *bts, const float *lr_ts, float *loss_dev)
{
    if (!t || !offsets || !bts || !lr_ts || n_steps < 0) return CEM_ERR_INVALID_ARG;
    for (int s = 0; s < n_steps; ++s) {
        const int st = cem_trainer_step(t, x_dev, y_dev, perm_dev, nperm, offsets[s], bts[s], lr_ts[s], loss_dev + (size_t)s * t->cfg.ensemble_size);
        if (st) return st;
    }
    return CEM_OK;
}

int cem_trainer_eval(cem_trainer_t *t, const float *x_dev, const float *y_dev, int32_t n, float *loss_out)
{
    if (!t || !x_dev || !y_dev || !loss_out || n < 1) return CEM_ERR_INVALID_ARG;
    const int E = t->cfg.ensemble_size;
    std::vector<float> sums((size_t)E * 2);
    std::fill(sums.begin(), sums.end(), 0.f);
    TrainParams p; fill_train_params(t, p);
    p.x = x_dev; p.y = y_dev; p.perm = nullptr; p.loss_out = (float *)(t->ws + t->oL); p.train = 0;
    const int B = t->cfg.batch_size, nchunks = (n + B - 1) / B;
    const size_t per_chunk = (size_t)E * CEM_TPARTS * 2;
    std::vector<float> part(per_chunk * (t->tile_kernel ? nchunks : 1));
    // the sums are added on the host in the same order either way: chunk by chunk, member by member, part by part
    auto add_chunk = [&](const float *pc, int rows) {
        const int nparts = (rows + CEM_TROWS - 1) / CEM_TROWS;
        for (int m = 0; m < E; ++m)
            for (int q = 0; q < nparts; ++q) { sums[2 * m] += pc[((size_t)m * CEM_TPARTS + q) * 2]; sums[2 * m + 1] += pc[((size_t)m * CEM_TPARTS + q) * 2 + 1]; }
    };
    if (t->tile_kernel) {
        // ONE launch for the whole set: grid.y walks the 64-row chunks, each writing its own loss partials
        if (t->eval_part_floats < part.size()) {
            if (t->eval_part) { HIPCHK(hipStreamSynchronize(t->stream)); HIPCHK(hipFree(t->eval_part)); t->eval_part = nullptr; t->eval_part_floats = 0; }
            HIPCHK(hipMalloc((void **)&t->eval_part, part.size() * 4)); t->eval_part_floats = part.size();
        }
        p.chunk = B;
        const int kMaxChunks = 32768;                   // grid.y is limited to 65535: very large sets go in several launches
        for (int c0 = 0; c0 < nchunks; c0 += kMaxChunks) {
            const int nc = std::min(kMaxChunks, nchunks - c0);
            p.offset = c0 * B; p.Bt = std::min(n - c0 * B, nc * B); p.loss_part = t->eval_part + (size_t)c0 * per_chunk;
            launch_train_step(t, p);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(part.data(), t->eval_part, part.size() * 4, hipMemcpyDeviceToHost, t->stream));
        HIPCHK(hipStreamSynchronize(t->stream));
        for (int ch = 0; ch < nchunks; ++ch) add_chunk(part.data() + (size_t)ch * per_chunk, std::min(B, n - ch * B));
    } else {
        for (int off = 0; off < n; off += B) {
            p.offset = off; p.Bt = std::min(B, n - off); p.chunk = p.Bt;
            launch_train_step(t, p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(part.data(), t->ws + t->oP, part.size() * 4, hipMemcpyDeviceToHost, t->stream));
            HIPCHK(hipStreamSynchronize(t->stream));
            add_chunk(part.data(), p.Bt);
        }
    }
    const double cnt = (double)n * t->cfg.outputs_dim;
    double total = 0;
    for (int m = 0; m < E; ++m) total += (0.5 * sums[2 * m] / cnt + 0.5 * sums[2 * m + 1] / cnt) / E;
    *loss_out = (float)total;
    return CEM_OK;
}

}  // extern "C"
