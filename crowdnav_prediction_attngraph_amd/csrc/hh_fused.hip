// hh_fused.hip -- the whole human-human block of the policy forward as ONE persistent kernel on gfx950:
//   embedding_layer (D -> 128 -> 512, ReLU) -> folded q|k|v projection -> 8-head attention over the humans of each env ->
//   folded out_proj∘spatial_linear (+ReLU)                       rl/networks/selfAttn_srnn_temp_node.py:63-91, :408
// on the compacted live rows.  Nothing between the observation and out_sp [rows,256] touches HBM: the activations live in
// LDS / registers, the weights stream from L2 straight into MFMA operand registers.
//
// Work split.  One workgroup per CU (all 160 KB of its LDS) owns a contiguous chunk of ~rows/256 live rows (whole envs) and walks it
// in tiles of <= 63 rows = <= 4 row blocks of 16.  Per tile:
//   e0   = relu(x W0^T + b0)                     VALU, written as bf16 hi/lo MFMA fragments into LDS
//   X    = relu(e0 W2^T + b2)        [rows,512]  MFMA; stays in LDS for the whole tile as hi/lo fragments (2 KB per row)
//   for each head h:  q|k|v = X Wh^T + bh        [rows,192]  MFMA (16 k-steps, the dominant loop)
//                     S^T = K Q^T, P = softmax over the keys of the query's env     MFMA + VALU (env block mask)
//                     O^T = V^T P^T              MFMA, V straight from the accumulators
//                     out += O Wos[:, h]^T       MFMA, accumulators persist over the heads
//   out_sp = relu(out + b)                       stored to HBM
// Arithmetic: every product is bf16x3 split precision (hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate),
// the same arithmetic as gemm3.h; softmax in fp32.
//
// Layout tricks that keep the chain on chip:
//   * a wavefront owns OUTPUT FEATURES (three 16-feature blocks of the head's q|k|v -- a PAIR of q blocks in the even wavefronts,
//     a pair of k blocks in the odd ones, and v block w; out_sp feature blocks 4w..4w+3), never rows: a weight
//     fragment is needed by exactly one wavefront, so it goes L2 -> VGPR with one coalesced 1 KB load (the fragments are stored in
//     streaming order by hh_fused_bake) and is reused by every row block; only activations use LDS.
//   * products are issued "transposed" (A operand = weight fragment, B operand = activation fragment): the C layout then has
//     4 consecutive output features of ONE row in a lane, i.e. half of the next product's activation fragment -- a pair of
//     feature blocks is one 16-byte fragment entry.  The contraction index of every consumer is permuted accordingly
//     (perm32: element u of lane group g <-> offset 16*(u>>2) + 4*g + (u&3)); weights are baked with the same permutation.
//   * V is produced in normal form (rows in the C layout) so that it is the A operand of O^T = V^T P^T without leaving the
//     registers, and S^T (keys in the C layout rows) is exactly the P fragment of the same product.
//   * LDS holds fragments in fragment-major order [plane][k-step][row block][lane][16 B]: every ds_read_b128 / ds_write_b128 is
//     lane-linear.  Q and K are stored as whole entries (that is what the q / k pairs above are for); only the 8-byte
//     half-fragment stores of O are 2-way (its 16-byte entries are shared by two wavefronts: a wavefront has ONE v block).
//
// Two kernels share this file:
//   hh_fused_kernel       (crowds of <= 63 humans, the default): 8 wavefronts = TWO TEAMS of four, two wavefronts per SIMD, 256
//       registers each.  After the embedding phases the teams walk ALTERNATE HEADS on their own (team barriers on LDS counters,
//       s_barrier only between tile phases), so that on every SIMD the barrier / LDS / softmax chain of one head sits beside the
//       q.k.v MFMA loop of another head.  Tiles of <= 48 rows give each team its own scratch region; tiles of 49..63 rows fill the
//       LDS with X and share one scratch region in turns (namespace team, struct Layout).  The teams' partial out_sp accumulators
//       are exchanged through the (dead) X region at the end of the tile.  DESIGN.md section 4 has the measurements behind each choice.
//   hh_fused_wide_kernel  (crowds of 64 humans -- an env must fit one tile -- and CN_HH_WIDE=1): the round-2 schedule, 4 wavefronts (one per SIMD),
//       64-row tiles.
#include "hh_fused.h"
#include "row_plan.h"

#include <climits>
#include <type_traits>
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// weight fragments are read once per tile by exactly one wavefront: stream them past the vector L1
__device__ __forceinline__ bf16x8 ldw(const char *p)
{
#ifndef HH_NT_LOADS
    return *reinterpret_cast<const bf16x8 *>(p);
#else
    return __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(p));
#endif
}
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): an unrolled loop whose index is a constant expression
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct Split8 { bf16x8 hi, lo; };
__device__ __forceinline__ void split1(float x, __bf16 &hi, __bf16 &lo)
{
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}
__device__ __forceinline__ Split8 split8(f32x4 a, f32x4 b)
{
    Split8 s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        __bf16 h, l;
        split1(a[q], h, l); s.hi[q] = h; s.lo[q] = l;
        split1(b[q], h, l); s.hi[4 + q] = h; s.lo[4 + q] = l;
    }
    return s;
}

// first index e in [0, n] with a[e] >= target (a ascending, a[n] readable); wave-uniform result, 64-ary search
// row offsets: the workgroup may have written them itself a moment ago (fused prefix sum below).  Writer and readers are wavefronts of
// ONE workgroup, i.e. one CU and one write-through L1: a workgroup-scope release / barrier / acquire orders them (waitcnt only), no
// cache maintenance -- a device-scope fence here costs an L2 write-back per workgroup and device-scope loads miss the L2 (the
// first version of this did both and lost 50 us per launch).
__device__ __forceinline__ int ld_ro(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ int lower_bound_wave(const int *a, int n, int target, int lane)
{
    int lo = 0, hi = n;
    while (hi > lo) {
        const int span = hi - lo, step = (span + 63) >> 6;
        const int idx = lo + lane * step;
        const int v = idx < hi ? ld_ro(a + idx) : INT_MAX;
        const unsigned long long m = __ballot(v >= target);
        const int f = m ? __ffsll(m) - 1 : 64;
        if (f == 0) { hi = lo; }
        else {
            const int nlo = lo + (f - 1) * step + 1;
            int nhi = lo + f * step;
            nhi = nhi < hi ? nhi : hi;
            lo = nlo < nhi ? nlo : nhi; hi = nhi;
        }
    }
    return __builtin_amdgcn_readfirstlane(lo);
}

// Chunk boundary c of n: the env start NEAREST to row c * total / n (both neighbours of a boundary compute the same env, so the
// chunks tile the batch exactly).  Nearest instead of "first at or behind" halves the spread of the chunk sizes: a chunk of more than
// ~104 rows may not split into a 3-block and a 4-block tile, and one such workgroup sets the duration of the launch.
__device__ __forceinline__ int chunk_boundary(const int *row_off, int E, int total, int c, int n, int lane)
{
    if (c <= 0) return 0;
    if (c >= n) return E;
    const int target = (int)((long long)c * total / n);
    const int e1 = lower_bound_wave(row_off, E, target, lane); // first env with start >= target (E if none)
    if (e1 == 0) return 0;
    const int above = ld_ro(row_off + e1) - target, below = target - ld_ro(row_off + e1 - 1);
    return below < above ? e1 - 1 : e1;
}

#ifdef HH_DEBUG
__device__ int *g_hh_dbg = nullptr;
#endif
#ifdef HH_TIMING
__device__ long long *g_hh_tim = nullptr; // [block][16] phase cycle sums of wavefront 0
#define HH_T(k) do { const long long now_ = clock64(); tacc[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define HH_T(k) do {} while (0)
#endif

struct TileCtx {
    int e_lo, n_env, r0, nrows; // envs [e_lo, e_lo + n_env), first compacted row, rows (e_lo / r0: contiguous tiles only)
    int my_env, my_start;       // lane l <-> row l of the tile: env index inside the tile (-1 beyond nrows), its first row
    int my_src, my_out;         // ... its row of the [E*H, D] input and its compacted output row (two-team kernel: tiles need not be contiguous)
    int tile_ord;
};

// ===================================================== wide kernel: 4 wavefronts, 64-row tiles (49..64 humans) ====================
namespace wide {
constexpr int FR = 64;                  // rows per tile (4 row blocks of 16)
constexpr int LDS_X = 0;                // [plane 2][kx 16][rb 4][lane 64][16 B]  = 128 KB
constexpr int LDS_S = 131072;           // scratch: two halves of 16 KB, [plane 2][ks 2][rb 4][lane 64][16 B] each
constexpr int LDS_H0 = LDS_S, LDS_H1 = LDS_S + 16384;
constexpr int LDS_BYTES = 163840;

template <int NRB>
__device__ __forceinline__ void tile_body(const TileCtx &t, int H, int D, const float *__restrict__ se, const HhFusedWeights &W,
                                          float *__restrict__ out_sp, char *lds, int lane, int wave)
{
    const int i = lane & 15, g = lane >> 4;
    const int loff = lane * 16;
    constexpr int NKS = (NRB + 1) / 2; // key k-steps of 32 rows
#ifdef HH_TIMING
    long long tacc[16] = {0}, tlast = clock64();
#endif

    // ---------------- e0: relu(x W0^T + b0) for feature k-step `wave` (natural k order), all row blocks ----------------
    {
        const int c0 = 32 * wave + 8 * g;
        const float *xp[NRB];
        float v[NRB][8];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            int row = rb * 16 + i;
            row = row < t.nrows ? row : t.nrows - 1; // padded rows repeat the last live row: finite values, masked later
            const int env = __shfl(t.my_env, row, 64), st = __shfl(t.my_start, row, 64);
            xp[rb] = se + ((size_t)(t.e_lo + env) * H + (row - st)) * D;
#pragma unroll
            for (int u = 0; u < 8; ++u) v[rb][u] = W.emb0_b[c0 + u];
        }
        for (int d = 0; d < D; ++d) { // input feature outermost: its 8 weights are loaded once and reused by every row block
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = W.emb0_w[(c0 + u) * D + d];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const float xd = xp[rb][d];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[rb][u] += xd * w[u];
            }
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            bf16x8 hi, lo;
#pragma unroll
            for (int u = 0; u < 8; ++u) { __bf16 h, l; split1(fmaxf(v[rb][u], 0.0f), h, l); hi[u] = h; lo[u] = l; }
            *reinterpret_cast<bf16x8 *>(lds + LDS_S + ((0 * 4 + wave) * 4 + rb) * 1024 + loff) = hi;
            *reinterpret_cast<bf16x8 *>(lds + LDS_S + ((1 * 4 + wave) * 4 + rb) * 1024 + loff) = lo;
        }
    }
    HH_T(0);
    __syncthreads();
    HH_T(1);
    // ---------------- X = relu(e0 W2^T + b2): wavefront w produces X k-steps 4w..4w+3 (feature blocks 8w..8w+7) ----------------
    // in two halves of 4 feature blocks (2 X k-steps) so that the accumulators + a prefetched weight k-step stay in registers
    {
        const char *wp = (const char *)W.emb2_frag + (size_t)wave * 4 * 16 * 1024 + loff;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            f32x4 acc[4][NRB];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[j][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 wq[2][8]; // [slot][j*2 + plane]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wq[0][2 * j] = ldw(wp + ((0 * 8 + half * 4 + j) * 2 + 0) * 1024);
                wq[0][2 * j + 1] = ldw(wp + ((0 * 8 + half * 4 + j) * 2 + 1) * 1024);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        wq[(ks + 1) & 1][2 * j] = ldw(wp + (((ks + 1) * 8 + half * 4 + j) * 2 + 0) * 1024);
                        wq[(ks + 1) & 1][2 * j + 1] = ldw(wp + (((ks + 1) * 8 + half * 4 + j) * 2 + 1) * 1024);
                    }
                }
                bf16x8 xh[NRB], xl[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    xh[rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_S + ((0 * 4 + ks) * 4 + rb) * 1024 + loff);
                    xl[rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_S + ((1 * 4 + ks) * 4 + rb) * 1024 + loff);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 *w = wq[ks & 1];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) acc[j][rb] = mfma(w[2 * j + 1], xh[rb], acc[j][rb]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) acc[j][rb] = mfma(w[2 * j], xl[rb], acc[j][rb]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) acc[j][rb] = mfma(w[2 * j], xh[rb], acc[j][rb]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int fb0 = 8 * wave + 4 * half + 2 * p;
                const f32x4 ba = *reinterpret_cast<const f32x4 *>(W.emb2_b + fb0 * 16 + 4 * g);
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(W.emb2_b + (fb0 + 1) * 16 + 4 * g);
                const int kx = 4 * wave + 2 * half + p;
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    f32x4 a = acc[2 * p][rb] + ba, b = acc[2 * p + 1][rb] + bb;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a[q] = fmaxf(a[q], 0.0f); b[q] = fmaxf(b[q], 0.0f); }
                    const Split8 s = split8(a, b);
                    *reinterpret_cast<bf16x8 *>(lds + LDS_X + ((0 * 16 + kx) * 4 + rb) * 1024 + loff) = s.hi;
                    *reinterpret_cast<bf16x8 *>(lds + LDS_X + ((1 * 16 + kx) * 4 + rb) * 1024 + loff) = s.lo;
                }
            }
        }
    }
    HH_T(2);
    __syncthreads();
    HH_T(3);

    // env block mask of this lane's S^T entries: query = 16*wave + i, keys 16*jb + 4*g + r
    unsigned vmask = 0;
    {
        const int q = 16 * wave + i;
        const int eq = __shfl(t.my_env, q, 64);
#pragma unroll
        for (int jb = 0; jb < NRB; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ek = __shfl(t.my_env, 16 * jb + 4 * g + r, 64);
                if (ek == eq && eq >= 0) vmask |= 1u << (jb * 4 + r);
            }
    }

#ifdef HH_DEBUG
    if (g_hh_dbg && blockIdx.x == 0 && t.e_lo == 0) {
        int *d = g_hh_dbg + wave * 64 * 8 + lane * 8;
        d[0] = t.my_env; d[1] = t.my_start; d[2] = (int)vmask; d[3] = t.nrows; d[4] = t.n_env; d[5] = t.r0; d[6] = NRB; d[7] = t.e_lo;
    }
#endif
    f32x4 acc_os[4][NRB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc_os[j][rb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Head order: every CU of an XCD would otherwise walk the SAME weight lines at the same moment (32 simultaneous readers of a
    // line, then the next line ...), which serialises on the L2 channel that owns the line.  Staggering the starting head by the
    // workgroup's index inside its XCD spreads the readers over 8 different streams (the sum over heads is order independent).
    const int h0 = ((int)blockIdx.x >> 3) & 7;
    for (int hh = 0; hh < 8; ++hh) {
        const int h = (h0 + hh) & 7;
        // ---------------- head h, every row block, 16 k-steps over X: feature blocks w & ~1, (w & ~1) + 1 of q (even wavefronts) or k
        // (odd) in aq, ak, and v feature block w in av ----------------
        f32x4 aq[NRB], ak[NRB], av[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) { aq[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; ak[rb] = aq[rb]; av[rb] = aq[rb]; }
        const char *wp = (const char *)W.qkv_frag + ((size_t)(h * 4 + wave) * 16) * 6 * 1024 + loff;
        constexpr int PF = 4; // weight prefetch depth (k-steps): the ring holds steps ks .. ks+PF-1
        bf16x8 wf[PF][6];
#pragma unroll
        for (int p = 0; p < PF - 1; ++p)
#pragma unroll
            for (int c = 0; c < 6; ++c) wf[p][c] = ldw(wp + (p * 6 + c) * 1024);
        // X fragments are double buffered: step ks+1 is read from LDS while the MFMAs of step ks run
        bf16x8 xh[2][NRB], xl[2][NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            xh[0][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + ((0 * 16 + 0) * 4 + rb) * 1024 + loff);
            xl[0][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + ((1 * 16 + 0) * 4 + rb) * 1024 + loff);
        }
#pragma unroll 1
        for (int k4 = 0; k4 < 16; k4 += PF) {
#pragma unroll
        for (int ku = 0; ku < PF; ++ku) {
            const int ks = k4 + ku;
            {
                // prefetch k-step ks + PF - 1 into the ring slot consumed last iteration (clamped: the tail re-reads step 15).
                // The scheduling barrier pins the issue point: left alone, the scheduler sinks these loads next to their use
                // three k-steps later and the prefetch distance collapses to one L2 round trip per k-step.
                const int kp = ks + PF - 1 < 16 ? ks + PF - 1 : 15;
#ifndef HH_EXP_NO_WLOAD
#pragma unroll
                for (int c = 0; c < 6; ++c) wf[(ku + PF - 1) % PF][c] = ldw(wp + (kp * 6 + c) * 1024);
#else
                asm volatile("" : "+v"(wf[(ku + PF - 1) % PF][0]) : "s"(kp));
#endif
                const int kn = ks + 1 < 16 ? ks + 1 : 15;
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    xh[(ku + 1) & 1][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + ((0 * 16 + kn) * 4 + rb) * 1024 + loff);
                    xl[(ku + 1) & 1][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + ((1 * 16 + kn) * 4 + rb) * 1024 + loff);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const bf16x8 *w6 = wf[ku]; // pair block A hi, lo, pair block B hi, lo (q in even wavefronts, k in odd), v hi, v lo
            const bf16x8 *xhc = xh[ku & 1], *xlc = xl[ku & 1];
#ifdef HH_EXP_NO_MFMA
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                aq[rb][0] += (float)w6[0][0] + (float)w6[1][0] + (float)xhc[rb][0]; ak[rb][0] += (float)w6[2][0] + (float)w6[3][0] + (float)xlc[rb][0];
                av[rb][0] += (float)w6[4][0] + (float)w6[5][0];
            }
#else
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                aq[rb] = mfma(w6[1], xhc[rb], aq[rb]);
                ak[rb] = mfma(w6[3], xhc[rb], ak[rb]);
                av[rb] = mfma(xhc[rb], w6[5], av[rb]);
            }
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                aq[rb] = mfma(w6[0], xlc[rb], aq[rb]);
                ak[rb] = mfma(w6[2], xlc[rb], ak[rb]);
                av[rb] = mfma(xlc[rb], w6[4], av[rb]);
            }
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                aq[rb] = mfma(w6[0], xhc[rb], aq[rb]);
                ak[rb] = mfma(w6[2], xhc[rb], ak[rb]);
                av[rb] = mfma(xhc[rb], w6[4], av[rb]);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        HH_T(4);
        {
            // aq / ak: the two feature blocks of this wavefront's pair, of q (even wavefronts) or k (odd) -- see bake_qkv_kernel
            const float *bp = W.qkv_b + (wave & 1) * 512 + h * 64 + 16 * (wave & ~1) + 4 * g;
            const f32x4 bq = *reinterpret_cast<const f32x4 *>(bp);
            const f32x4 bk = *reinterpret_cast<const f32x4 *>(bp + 16);
            const float bv = W.qkv_b[1024 + h * 64 + 16 * wave + i];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                aq[rb] += bq; ak[rb] += bk;
#pragma unroll
                for (int q = 0; q < 4; ++q) av[rb][q] += bv;
            }
        }
        HH_T(5);
        __syncthreads(); // A: the previous head's O fragments (H1) have been consumed by every wavefront
        HH_T(6);
        // Q -> H0 (even wavefronts), K -> H1 (odd): lane (i, g) holds head features 32 s + 4g + r (aq) and 32 s + 16 + 4g + r (ak), s = w >> 1,
        // of its rows = all 8 contraction entries of lane group g in k-step s: whole fragment entries, 16-byte stores
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const Split8 s = split8(aq[rb], ak[rb]);
            char *const dst = lds + ((wave & 1) ? LDS_H1 : LDS_H0) + ((wave >> 1) * 4 + rb) * 1024 + loff;
            *reinterpret_cast<bf16x8 *>(dst) = s.hi;
            *reinterpret_cast<bf16x8 *>(dst + 8192) = s.lo;
        }
        HH_T(7);
        __syncthreads(); // B
        HH_T(8);
        // this head's out_proj∘spatial_linear fragments: in flight during the attention phases, consumed after barrier E
        bf16x8 wos[2][4][2];
        {
            const char *op = (const char *)W.os_frag + ((size_t)(h * 4 + wave) * 2) * 8 * 1024 + loff;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wos[ks][j][0] = ldw(op + ((ks * 4 + j) * 2 + 0) * 1024);
                    wos[ks][j][1] = ldw(op + ((ks * 4 + j) * 2 + 1) * 1024);
                }
        }
        // ---------------- S^T = K Q^T for query block `wave`, masked softmax over the keys of the query's env ----------------
        float p[4][4]; // [jb][r]
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[jb][r] = 0.0f;
        if (wave < NRB) {
            f32x4 s[NRB];
#pragma unroll
            for (int jb = 0; jb < NRB; ++jb) s[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 qh = *reinterpret_cast<const bf16x8 *>(lds + LDS_H0 + (ks * 4 + wave) * 1024 + loff);
                const bf16x8 ql = *reinterpret_cast<const bf16x8 *>(lds + LDS_H0 + 8192 + (ks * 4 + wave) * 1024 + loff);
#pragma unroll
                for (int jb = 0; jb < NRB; ++jb) {
                    const bf16x8 kh = *reinterpret_cast<const bf16x8 *>(lds + LDS_H1 + (ks * 4 + jb) * 1024 + loff);
                    const bf16x8 kl = *reinterpret_cast<const bf16x8 *>(lds + LDS_H1 + 8192 + (ks * 4 + jb) * 1024 + loff);
                    s[jb] = mfma(kl, qh, s[jb]);
                    s[jb] = mfma(kh, ql, s[jb]);
                    s[jb] = mfma(kh, qh, s[jb]);
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int jb = 0; jb < NRB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((vmask >> (jb * 4 + r)) & 1u) mx = fmaxf(mx, s[jb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.0f;
#pragma unroll
            for (int jb = 0; jb < NRB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = ((vmask >> (jb * 4 + r)) & 1u) ? __builtin_amdgcn_exp2f((s[jb][r] - mx) * 1.44269504088896340736f) : 0.0f;
                    p[jb][r] = e; sum += e;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
#pragma unroll
            for (int jb = 0; jb < NRB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[jb][r] *= inv;
        }
        HH_T(9);
        __syncthreads(); // C: Q (H0) is dead, P may overwrite it
        HH_T(10);
        if (wave < NRB) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const Split8 sp = split8(f32x4{p[2 * ks][0], p[2 * ks][1], p[2 * ks][2], p[2 * ks][3]},
                                         f32x4{p[2 * ks + 1][0], p[2 * ks + 1][1], p[2 * ks + 1][2], p[2 * ks + 1][3]});
                *reinterpret_cast<bf16x8 *>(lds + LDS_H0 + (ks * 4 + wave) * 1024 + loff) = sp.hi;
                *reinterpret_cast<bf16x8 *>(lds + LDS_H0 + 8192 + (ks * 4 + wave) * 1024 + loff) = sp.lo;
            }
        }
        __syncthreads(); // D
        HH_T(11);
        // ---------------- O^T = V^T P^T: value features 16w..16w+15 of the head for every query block ----------------
        f32x4 o[NRB];
#pragma unroll
        for (int ib = 0; ib < NRB; ++ib) o[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const Split8 vf = split8(av[2 * ks], 2 * ks + 1 < NRB ? av[2 * ks + 1 < NRB ? 2 * ks + 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int ib = 0; ib < NRB; ++ib) {
                const bf16x8 ph = *reinterpret_cast<const bf16x8 *>(lds + LDS_H0 + (ks * 4 + ib) * 1024 + loff);
                const bf16x8 pl = *reinterpret_cast<const bf16x8 *>(lds + LDS_H0 + 8192 + (ks * 4 + ib) * 1024 + loff);
                o[ib] = mfma(vf.lo, ph, o[ib]);
                o[ib] = mfma(vf.hi, pl, o[ib]);
                o[ib] = mfma(vf.hi, ph, o[ib]);
            }
        }
#pragma unroll
        for (int ib = 0; ib < NRB; ++ib) {
            bf16x4 oh, ol;
#pragma unroll
            for (int q = 0; q < 4; ++q) { __bf16 a, b; split1(o[ib][q], a, b); oh[q] = a; ol[q] = b; }
            const int off = ((wave >> 1) * 4 + ib) * 1024 + loff + 8 * (wave & 1);
            *reinterpret_cast<bf16x4 *>(lds + LDS_H1 + off) = oh;
            *reinterpret_cast<bf16x4 *>(lds + LDS_H1 + 8192 + off) = ol;
        }
        HH_T(12);
        __syncthreads(); // E
        HH_T(13);
        // ---------------- out += O Wos[:, head h]^T: feature blocks 4w..4w+3 ----------------
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 oh[NRB], ol[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                oh[rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_H1 + (ks * 4 + rb) * 1024 + loff);
                ol[rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_H1 + 8192 + (ks * 4 + rb) * 1024 + loff);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc_os[j][rb] = mfma(wos[ks][j][1], oh[rb], acc_os[j][rb]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc_os[j][rb] = mfma(wos[ks][j][0], ol[rb], acc_os[j][rb]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc_os[j][rb] = mfma(wos[ks][j][0], oh[rb], acc_os[j][rb]);
        }
        HH_T(14);
    }
    // ---------------- out_sp = relu(out + b) ----------------
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f0 = (4 * wave + j) * 16 + 4 * g;
        const f32x4 b = *reinterpret_cast<const f32x4 *>(W.os_b + f0);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const int row = rb * 16 + i;
            if (row < t.nrows) {
                f32x4 v = acc_os[j][rb] + b;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.0f);
                // streamed past this XCD's L2: the reader (rn_fused) runs on other XCDs anyway, and the 24 MB of rows would push
                // the 3.9 MB weight image, which every tile of every WG on the XCD re-reads, out of the 4 MB L2
#ifndef HH_NO_NT_STORE
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(out_sp + (size_t)(t.r0 + row) * 256 + f0));
#else
                *reinterpret_cast<f32x4 *>(out_sp + (size_t)(t.r0 + row) * 256 + f0) = v;
#endif
            }
        }
    }
    __syncthreads(); // the next tile's e0 fragments overwrite the scratch halves
    HH_T(15);
#ifdef HH_TIMING
    if (g_hh_tim && wave == 0 && lane == 0) {
        long long *d = g_hh_tim + (size_t)blockIdx.x * 20;
        for (int k = 0; k < 16; ++k) d[k] += tacc[k];
        d[16] += 1; d[17] += t.nrows; d[18] += NRB;
        long long *d2 = g_hh_tim + (size_t)256 * 20 + ((size_t)blockIdx.x * 4 + (t.tile_ord < 3 ? t.tile_ord : 3)) * 4;
        d2[0] += tacc[4]; d2[1] += NRB; d2[2] += 1;
    }
#endif
}
} // namespace wide

// Row compaction fused into the launch: row_off[e] = sum_{e' < e} clamp(detected_human_num[e'], 1, H).  EVERY workgroup
// computes the whole prefix sum (E values: a few microseconds next to ~150) and writes the whole array -- all workgroups
// write identical values, so nobody has to wait for anybody, and the separate 16 us single-wavefront launch (plus its
// launch gap) is gone from the critical path.  The robot-node kernel behind this one reads the same array.
template <int NT> // threads of the workgroup
__device__ __forceinline__ void row_offsets_prologue(int E, int H, const float *__restrict__ det, int *row_off, unsigned long long *live_total, char *lds)
{
    constexpr int NW = NT / 64;
    int *part = reinterpret_cast<int *>(lds);
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    const int chunk = (E + NT - 1) / NT;
    const int lo = tid * chunk < E ? tid * chunk : E, hi = lo + chunk < E ? lo + chunk : E;
    int sum = 0;
    for (int e = lo; e < hi; ++e) { int nd = (int)det[e]; nd = nd < 1 ? 1 : (nd > H ? H : nd); sum += nd; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (ln >= o) incl += v;
    }
    if (ln == 63) part[wv] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wv; ++w) run += part[w];
    for (int e = lo; e < hi; ++e) {
        row_off[e] = run;
        int nd = (int)det[e]; nd = nd < 1 ? 1 : (nd > H ? H : nd);
        run += nd;
    }
    if (tid == NT - 1) {
        row_off[E] = run;
        if (live_total && blockIdx.x == 0) *live_total += (unsigned long long)run; // measurement aid: live rows over the profiled launches
    }
    (void)NW;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The next tile of a workgroup's chunk [e, e_end): whole envs, at most FR rows (every env has 1..H <= FR rows), in the fewest tiles
// (a tile costs a pass over the 3.9 MB weight stream whatever its size).  Among the env boundaries that keep that tile count, the one
// that minimises the 16-row blocks of this tile plus those of the rest is taken: the MFMA time of a tile is proportional to its row
// blocks, so 100 rows are cut 48 + 52 (3 + 4 blocks), not 50 + 50 (4 + 4) -- measured per workgroup: 27 k cycles per row block on top
// of 42 k per tile, and the launch lasts as long as its slowest workgroup.  Wave-uniform; every lane of the calling wavefront active.
template <int FR>
__device__ __forceinline__ TileCtx next_tile(const int *row_off, int e, int e_end, int chunk_end_row, int tile_ord, int lane)
{
    TileCtx t;
    t.tile_ord = tile_ord;
    t.e_lo = e;
    t.r0 = ld_ro(row_off + e);
    const int probe = e + 1 + lane;
    const int v = probe <= e_end ? ld_ro(row_off + probe) : INT_MAX; // end row of env e + lane
    const int left = chunk_end_row - t.r0;
    const int ntile = (left + FR - 1) / FR;
    // candidate: the tile ends behind env e + lane
    const int rows = v == INT_MAX ? INT_MAX : v - t.r0;
    const bool ok = rows <= FR && left - rows <= FR * (ntile - 1);
    const int rest = left - rows;
    // cost: row blocks (this tile + lower bound for the rest); ties: the larger tile (the rest may then round down a block later)
    int key = ok ? ((((rows + 15) >> 4) + ((rest + 15) >> 4)) << 8) + (FR - rows) : INT_MAX;
    int best = key;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(best, o, 64); best = x < best ? x : best; }
    int n_env;
    if (best == INT_MAX) {
        // no boundary keeps the minimal tile count (env sizes make FR-row tiles impossible to fill): take as many envs as fit
        n_env = __popcll(__ballot(rows <= FR));
        n_env = n_env < 1 ? 1 : n_env; // one env always fits (H <= FR)
    } else {
        n_env = __ffsll(__ballot(key == best)); // first lane holding the minimum: lane index + 1 = number of envs
    }
    t.n_env = n_env;
    t.nrows = __builtin_amdgcn_readfirstlane(__shfl(v, n_env - 1, 64)) - t.r0;
    // row -> env map: lane k < n_env knows the start of env k, lane l then counts the starts <= l
    // (the shuffle must run with every lane active: as part of the conditional below the compiler executes it under the
    // narrowed EXEC mask and a ds_bpermute from an inactive lane returns 0)
    const int prev_end = __shfl(v, lane >= 1 ? lane - 1 : 0, 64);
    const int st = lane >= n_env ? INT_MAX : (lane == 0 ? 0 : prev_end - t.r0);
    int cnt = 0;
    for (int k = 0; k < n_env; ++k) cnt += lane >= __builtin_amdgcn_readlane(st, k) ? 1 : 0;
    t.my_env = lane < t.nrows ? cnt - 1 : -1;
    t.my_start = __shfl(st, cnt - 1 >= 0 ? cnt - 1 : 0, 64);
    t.my_src = 0; t.my_out = t.r0 + lane; // my_src: set by the caller that needs it (it knows H)
    return t;
}

// A tile of the row plan (row_plan.h): the envs listed for it, in the order listed.  The two loads (item k of the list on lane k, then the
// first output row of that env) are issued a tile ahead (PlanPre) -- two memory round trips that would otherwise open every tile.
struct PlanPre { int it, out0; };
__device__ __forceinline__ PlanPre plan_prefetch(const int32_t *plan, int E, int tile, int lane)
{
    PlanPre p;
    p.it = (plan + rp_off_items(E))[(size_t)tile * 64 + lane];
    p.out0 = (plan + rp_off_rowoff())[p.it & 0xffff]; // slots behind the end of the list hold stale items: any 16-bit index stays inside the plan
    return p;
}
__device__ __forceinline__ TileCtx plan_tile(const PlanPre &pre, int H, int tile_ord, int lane)
{
    TileCtx t;
    const unsigned long long zero = __ballot((pre.it >> 16) == 0); // the list ends at its first zero item
    const int cnt = zero ? __ffsll((long long)zero) - 1 : 64;
    const int it = lane < cnt ? pre.it : 0;
    const int rows = it >> 16, id = it & 0xffff;
    const int out0 = pre.out0;
    int incl = rows;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
    const int st = lane < cnt ? incl - rows : INT_MAX;
    t.tile_ord = tile_ord; t.e_lo = 0; t.r0 = 0; t.n_env = cnt;
    t.nrows = __builtin_amdgcn_readfirstlane(__shfl(incl, 63, 64));
    int c = 0;
    for (int k = 0; k < cnt; ++k) c += lane >= __builtin_amdgcn_readlane(st, k) ? 1 : 0;
    const int me = c - 1 >= 0 ? c - 1 : 0;
    t.my_env = lane < t.nrows ? c - 1 : -1;
    t.my_start = __shfl(st, me, 64);
    const int off = lane - t.my_start;
    t.my_src = __shfl(id, me, 64) * H + off;
    t.my_out = __shfl(out0, me, 64) + off;
    return t;
}

__global__ __launch_bounds__(256, 1) void hh_fused_wide_kernel(int E, int H, int D, const float *__restrict__ se, const float *__restrict__ det,
                                                               int *row_off, unsigned long long *live_total, HhFusedWeights W, float *__restrict__ out_sp,
                                                               unsigned long long *stamp)
{
    const CnStampScope stamp_scope(stamp);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (det) row_offsets_prologue<256>(E, H, det, row_off, live_total, lds);
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) stamp[1] = (unsigned long long)row_off[E];
    // this kernel is the critical path of the step; the simulator's ORCA wavefronts of the side stream share the SIMDs with it and
    // are latency tolerant: win the issue arbitration against them
    if (W.prio) __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total = ld_ro(row_off + E);
    // chunk of this workgroup: envs [boundary(c), boundary(c + 1))
    int e = chunk_boundary(row_off, E, total, (int)blockIdx.x, (int)gridDim.x, lane);
    const int e_end = chunk_boundary(row_off, E, total, (int)blockIdx.x + 1, (int)gridDim.x, lane);
    if (e >= e_end) return;
    int tile_ord = 0;
    const int chunk_end_row = ld_ro(row_off + e_end);
    while (e < e_end) {
        const TileCtx t = next_tile<wide::FR>(row_off, e, e_end, chunk_end_row, tile_ord++, lane);
        const int nrb = (t.nrows + 15) >> 4;
        switch (nrb) {
        case 1: wide::tile_body<1>(t, H, D, se, W, out_sp, lds, lane, wave); break;
        case 2: wide::tile_body<2>(t, H, D, se, W, out_sp, lds, lane, wave); break;
        case 3: wide::tile_body<3>(t, H, D, se, W, out_sp, lds, lane, wave); break;
        default: wide::tile_body<4>(t, H, D, se, W, out_sp, lds, lane, wave); break;
        }
        e += t.n_env;
    }
}

// ===================================================== two-team kernel: 8 wavefronts, 63-row tiles (<= 63 humans) =====================
namespace team {

#ifndef HH_TEAM_PFC
#define HH_TEAM_PFC 8   // cap on the k-steps of weight fragments requested across the attention chain (default: no cap below PF - 1)
#endif

constexpr int FR = 63;                       // rows per tile: at most 4 row blocks of 16; the last row of a fourth block is never live (see CTR)
constexpr int LDS_BYTES = 163840;            // all of the CU's LDS
// The LDS layout depends on the row blocks of the tile:
//   <= 3 row blocks (<= 48 rows, the common case at 4096 envs): X takes 96 KB and each team has its OWN 24 KB scratch region -- the
//       attention chains of the two teams run independently of each other;
//   4 row blocks (49..63 rows): X takes 128 KB and the teams share ONE 32 KB scratch region in turns.
template <int RB_>
struct Layout {
    static constexpr int RB = RB_;                     // row-block stride of every LDS image
    static constexpr bool TURNS = RB_ == 4;            // one shared scratch region, used in turns
    static constexpr int X_PLANE = 16 * RB * 1024;     // X: [plane 2][kx 16][rb RB][lane 64][16 B]
    static constexpr int LDS_X = 0;
    static constexpr int LDS_S = 2 * X_PLANE;          // scratch: H0 (Q, then P) and H1 (K, then O), [plane 2][ks 2][rb RB][lane 64][16 B] each
    static constexpr int H_PLANE = 2 * RB * 1024;
    static constexpr int H_BYTES = 2 * H_PLANE;
    static constexpr int TEAM_S = TURNS ? 0 : 2 * H_BYTES; // offset of team 1's scratch region
    static constexpr int E0_PLANE = 4 * RB * 1024;     // e0: [plane 2][ks 4][rb RB][lane 64][16 B] = one scratch region
    static constexpr int XCH_TEAM = 4 * 2 * RB * 1024; // accumulator exchange at the end of a tile: [wave 4][jj 2][rb RB][lane 64][16 B] per team, over X
    // Synchronisation counters (+0 / +4: team barriers, +8: finished scratch turns x 4; zeroed at the start of every tile).  With 3 row
    // blocks there is room behind the scratch regions.  With 4 the images fill the LDS to the last byte and the counters live in the
    // fragment slot of the one (row, k) position that can never matter: plane lo, X k-step 15, row block 3, lane 63 = row 63 of the
    // tile, k 24..31 -- a tile holds at most 63 rows, so row 63 is always padding.  Nobody else writes the slot (the X epilogue skips
    // it); whoever reads it as an X fragment sees small integers = tiny finite bf16 values, which only ever reach the (discarded)
    // outputs of the padded row.
    static constexpr int CTR = TURNS ? LDS_X + X_PLANE + (15 * RB + 3) * 1024 + 63 * 16 : LDS_S + 4 * H_BYTES;
    static_assert(LDS_S + (TURNS ? 2 : 4) * H_BYTES + (TURNS ? 0 : 16) <= LDS_BYTES, "layout exceeds the LDS");
};

typedef __attribute__((address_space(3))) unsigned lds_u32;

// Weight fragments come in through buffer loads: (resource in SGPRs) + (wave-uniform byte offset in an SGPR) + (lane offset, ONE VGPR
// for every stream of the kernel).  With flat addressing the compiler hoists a 64-bit VGPR address pair per fragment out of the tile
// loop and spills them at 256 registers.  One resource spans the three fragment images (they are carved from one allocation).
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct WeightBuf {
    __amdgpu_buffer_rsrc_t rs;
    unsigned emb2, qkv, os; // byte offsets of the three images inside the resource
};
__device__ __forceinline__ WeightBuf make_weight_buf(const HhFusedWeights &W)
{
    const char *a = (const char *)W.emb2_frag, *b = (const char *)W.qkv_frag, *c = (const char *)W.os_frag;
    const char *lo = a < b ? a : b; lo = lo < c ? lo : c;
    WeightBuf wb;
    wb.rs = __builtin_amdgcn_make_buffer_rsrc((void *)lo, 0, 0x7fffffff, 0x00020000);
    wb.emb2 = (unsigned)(a - lo); wb.qkv = (unsigned)(b - lo); wb.os = (unsigned)(c - lo);
    return wb;
}
__device__ __forceinline__ bf16x8 ldb(const WeightBuf &wb, unsigned soff, unsigned voff)
{
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wb.rs, voff, soff, 0));
}

// Training outputs go out through buffer stores whose resource covers exactly the live rows of the tile: a padded row's store is out of
// range and the hardware drops it -- no per-row predication (EXEC juggling, one SGPR pair per condition) in the hot loops.
struct RowBuf { __amdgpu_buffer_rsrc_t rs; };
__device__ __forceinline__ RowBuf make_row_buf(float *base, int r0, int nrows, int width)
{
    RowBuf rb;
    rb.rs = __builtin_amdgcn_make_buffer_rsrc((void *)(base + (size_t)r0 * width), 0, nrows * width * 4, 0x00020000);
    return rb;
}
// Stores of the TRAINING forward's saved activations (e0 / x / qkv / attn: 11.8 KB per row, 4.8 GB per launch at 405 k rows).  With the
// default cache policy they allocate in the XCD's 4 MB L2 and evict the 3.9 MB weight image that every tile streams from there: round 6's
// counters show 2.5 GB of fabric reads per training launch against 19 MB in the rollout (weights resident).  nt (aux bit 1) = streaming
// stores that do not displace the weights.  HH_TRAIN_STORE_AUX=0 restores the default policy (A/B).
#ifndef HH_TRAIN_STORE_AUX
#define HH_TRAIN_STORE_AUX 2
#endif
__device__ __forceinline__ void st4(const RowBuf &b, int float_off, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), b.rs, float_off * 4, 0, HH_TRAIN_STORE_AUX);
}
__device__ __forceinline__ void st1(const RowBuf &b, int float_off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), b.rs, float_off * 4, 0, HH_TRAIN_STORE_AUX);
}

// Barrier among the four wavefronts of ONE team (gfx950 has a single s_barrier per workgroup, which would couple the teams): a
// monotonic LDS counter, every wavefront adds 1 and spins until the count reaches 4 x (barriers so far).  LDS operations of one
// wavefront complete in order and lgkmcnt(0) has been waited for before the add, so whoever sees the count sees the data (and the
// adder's own earlier LDS reads have returned: the write-after-read side).  The asm statements are compiler barriers as well.
__device__ __forceinline__ void team_barrier(char *lds, int bar_off, unsigned &target, int lane)
{
    lds_u32 *ctr = (lds_u32 *)(lds + bar_off);
    target += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while ((int)(__builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - target) < 0)
        __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// The scratch region is used in TURNS: turn n belongs to team n & 1 (heads alternate between the teams).  A wavefront may touch the scratch
// in turn n once all 4 wavefronts of turn n - 1 have finished with it (which also covers its own team's turn n - 2: it replaces the
// barrier before the Q / K exchange).  `done` counts finished (turn, wavefront) pairs.
__device__ __forceinline__ void wait_turn(char *lds, int ctr_off, unsigned turn)
{
    lds_u32 *done = (lds_u32 *)(lds + ctr_off + 8);
    const unsigned need = 4u * turn;
    while ((int)(__builtin_amdgcn_readfirstlane(__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - need) < 0)
        __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void finish_turn(char *lds, int ctr_off, int lane)
{
    lds_u32 *done = (lds_u32 *)(lds + ctr_off + 8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wavefront's last reads of the scratch have returned
    if (lane == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// out_sp rows of feature blocks 4*wave + J0, 4*wave + J0 + 1: own partial sum + the other team's (from LDS) + bias, ReLU, streamed to HBM
template <int NRB, int J0, int RB>
__device__ __forceinline__ void finish_rows(const TileCtx &t, const f32x4 (&acc)[4][NRB], const char *xch, const HhFusedWeights &W,
                                            float *__restrict__ out_sp, int lane, int wave, const f32x4 *bpre = nullptr)
{
    const int i = lane & 15, g = lane >> 4, loff = lane * 16;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int f0 = (4 * wave + J0 + jj) * 16 + 4 * g;
        const f32x4 b = bpre ? bpre[jj] : *reinterpret_cast<const f32x4 *>(W.os_b + f0);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const f32x4 other = *reinterpret_cast<const f32x4 *>(xch + ((wave * 2 + jj) * RB + rb) * 1024 + loff);
            const int row = rb * 16 + i;
            const int orow = __shfl(t.my_out, row, 64); // every lane takes part in the shuffle: rows 0..63 = lanes 0..63
            if (row < t.nrows) {
                f32x4 v = acc[J0 + jj][rb] + other + b;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.0f);
                // streamed past this XCD's L2: the reader (rn_fused) runs on other XCDs anyway, and the 24 MB of rows would push
                // the 3.9 MB weight image, which every tile of every workgroup on the XCD re-reads, out of the 4 MB L2
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(out_sp + (size_t)orow * 256 + f0));
            }
        }
    }
}

// XM: how the q.k.v loop holds its X fragments -- 1: two sets, the next k-step's read while this one is multiplied; 2: ONE set refilled in
// place (see the loop); 0: one set, read at the top of the step
template <int NRB, int PF, int XM, bool TRAIN>
__device__ __forceinline__ void tile_body(const TileCtx &t, int H, int D, const float *__restrict__ se, const HhFusedWeights &W, const WeightBuf &WB,
                                          float *__restrict__ out_sp, char *lds, int lane_in, int wave, int tm)
{
    // Everything derived from the lane index is recomputed per tile: visible as loop invariant, the compiler hoists a few dozen
    // per-lane addresses out of the tile loop and keeps them alive (spilled) across the whole kernel.
    using L = Layout<(NRB <= 3 ? 3 : 4)>;
    constexpr int RB = L::RB, X_PLANE = L::X_PLANE, LDS_X = L::LDS_X, LDS_S = L::LDS_S, H_PLANE = L::H_PLANE, H_BYTES = L::H_BYTES,
                  E0_PLANE = L::E0_PLANE, XCH_TEAM = L::XCH_TEAM, LDS_CTR = L::CTR;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int i = lane & 15, g = lane >> 4;
    const int loff = lane * 16;
    const unsigned uoff = (unsigned)loff;
    RowBuf o_e0{}, o_x{}, o_qkv{}, o_attn{};
    if (TRAIN) {
        o_e0 = make_row_buf(W.e0_out, t.r0, t.nrows, 128); o_x = make_row_buf(W.x_out, t.r0, t.nrows, 512);
        o_qkv = make_row_buf(W.qkv_out, t.r0, t.nrows, 1536); o_attn = make_row_buf(W.attn_out, t.r0, t.nrows, 512);
    }
    constexpr int NKS = (NRB + 1) / 2; // key k-steps of 32 rows
    const int w8 = 4 * tm + wave;
    RowBuf BB;
    BB.rs = __builtin_amdgcn_make_buffer_rsrc((void *)W.qkv_b, 0, 1536 * 4, 0x00020000);
#ifdef HH_TIMING
    long long tacc[16] = {0}, tlast = clock64();
#endif

    // the synchronisation counters of this tile's layout start at zero (every wavefront is past the previous tile's last team barrier
    // when it gets here, and nobody touches them before the second __syncthreads below)
    if (w8 == 0 && lane < 4) *reinterpret_cast<unsigned *>(lds + LDS_CTR + 4 * lane) = 0u;
    unsigned bar_target = 0;
    // The embedding_layer.2 weights of this wavefront (4 k-steps x 8 fragments, 32 KB) are requested before anything else: in the
    // X phase below all eight wavefronts are in the same phase, nothing hides its 256 KB weight stream (6 k cycles at the L2 rate,
    // the longest item of the tile prologue) -- here it runs under the latency-bound e0 phase.
    bf16x8 wq[4][8]; // [k-step][j*2 + plane]
    {
        const int wv = w8 >> 1, half = w8 & 1; // position in the baked image: [wave 4][ks 4][j 8 = half 2 x 4][plane 2]
        const unsigned wp = WB.emb2 + (unsigned)wv * 4 * 16 * 1024;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wq[ks][2 * j] = ldb(WB, wp + ((ks * 8 + half * 4 + j) * 2 + 0) * 1024, uoff);
                wq[ks][2 * j + 1] = ldb(WB, wp + ((ks * 8 + half * 4 + j) * 2 + 1) * 1024, uoff);
            }
    }
    // ---------------- e0: relu(x W0^T + b0) for feature k-step `wave` (natural k order); team tm takes the row blocks of its parity ----------------
    {
        const int c0 = 32 * wave + 8 * g;
#pragma unroll
        for (int rr = 0; rr < (NRB + 1) / 2; ++rr) {
            const int rb = 2 * rr + tm;
            if (rb < NRB) {
                int row = rb * 16 + i;
                row = row < t.nrows ? row : t.nrows - 1; // padded rows repeat the last live row: finite values, masked later
                const float *xp = se + (size_t)__shfl(t.my_src, row, 64) * D;
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = W.emb0_b[c0 + u];
                for (int d = 0; d < D; ++d) {
                    const float xd = xp[d];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] += xd * W.emb0_w[(c0 + u) * D + d];
                }
                bf16x8 hi, lo;
#pragma unroll
                for (int u = 0; u < 8; ++u) { v[u] = fmaxf(v[u], 0.0f); __bf16 h, l; split1(v[u], h, l); hi[u] = h; lo[u] = l; }
                if (TRAIN) {
                    st4(o_e0, (rb * 16 + i) * 128 + c0, f32x4{v[0], v[1], v[2], v[3]});
                    st4(o_e0, (rb * 16 + i) * 128 + c0 + 4, f32x4{v[4], v[5], v[6], v[7]});
                }
                *reinterpret_cast<bf16x8 *>(lds + LDS_S + (wave * RB + rb) * 1024 + loff) = hi;
                *reinterpret_cast<bf16x8 *>(lds + LDS_S + E0_PLANE + (wave * RB + rb) * 1024 + loff) = lo;
            }
        }
    }
    __syncthreads();
    HH_T(0);
    // ---------------- X = relu(e0 W2^T + b2): wavefront w8 of the 8 produces feature blocks 4*w8..4*w8+3 (X k-steps 2*w8, 2*w8+1) ----------------
    {
        f32x4 acc[4][NRB];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[j][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 xh[NRB], xl[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                xh[rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_S + (ks * RB + rb) * 1024 + loff);
                xl[rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_S + E0_PLANE + (ks * RB + rb) * 1024 + loff);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 *w = wq[ks];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[j][rb] = mfma(w[2 * j + 1], xh[rb], acc[j][rb]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[j][rb] = mfma(w[2 * j], xl[rb], acc[j][rb]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[j][rb] = mfma(w[2 * j], xh[rb], acc[j][rb]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int fb0 = 4 * w8 + 2 * p;
            const f32x4 ba = *reinterpret_cast<const f32x4 *>(W.emb2_b + fb0 * 16 + 4 * g);
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(W.emb2_b + (fb0 + 1) * 16 + 4 * g);
            const int kx = 2 * w8 + p;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                f32x4 a = acc[2 * p][rb] + ba, b = acc[2 * p + 1][rb] + bb;
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[q] = fmaxf(a[q], 0.0f); b[q] = fmaxf(b[q], 0.0f); }
                if (TRAIN) {
                    st4(o_x, (rb * 16 + i) * 512 + fb0 * 16 + 4 * g, a);
                    st4(o_x, (rb * 16 + i) * 512 + fb0 * 16 + 4 * g + 16, b);
                }
                const Split8 s = split8(a, b);
                *reinterpret_cast<bf16x8 *>(lds + LDS_X + (kx * RB + rb) * 1024 + loff) = s.hi;
                if (!(L::TURNS && rb == 3 && kx == 15 && lane == 63)) // the synchronisation counters live in this slot (Layout::CTR)
                    *reinterpret_cast<bf16x8 *>(lds + LDS_X + X_PLANE + (kx * RB + rb) * 1024 + loff) = s.lo;
            }
        }
    }
    __syncthreads(); // X complete; the e0 fragments (team 0's scratch) are dead
    HH_T(1);

    // env block mask of this lane's S^T entries: query = 16*wave + i, keys 16*jb + 4*g + r
    unsigned vmask = 0;
    {
        const int q = 16 * wave + i;
        const int eq = __shfl(t.my_env, q, 64);
#pragma unroll
        for (int jb = 0; jb < NRB; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ek = __shfl(t.my_env, 16 * jb + 4 * g + r, 64);
                if (ek == eq && eq >= 0) vmask |= 1u << (jb * 4 + r);
            }
    }

    f32x4 acc_os[4][NRB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc_os[j][rb] = f32x4{0.f, 0.f, 0.f, 0.f};

    char *const H0 = lds + LDS_S + tm * L::TEAM_S, *const H1 = H0 + H_BYTES;
    const int bar = LDS_CTR + 4 * tm;
    // Head order: team tm takes every second head; the starting head is staggered over the workgroups of an XCD (32 simultaneous
    // readers of one weight line serialise on its L2 channel).  The sum over the heads is order independent up to fp32 rounding.
    const int h0 = ((int)blockIdx.x >> 3) & 7;
    // weight prefetch ring of the q.k.v loop: steps ks .. ks+PF-1.  The first PF-1 steps of a head are requested BEFORE the previous
    // head's attention chain (and, for the first head, before the X epilogue above), so that every loop starts on a warm ring instead
    // of a cold L2 round trip.
    bf16x8 wf[PF][6];
    constexpr bool XDB = XM == 1;
    constexpr int PFC = PF - 1 < HH_TEAM_PFC ? PF - 1 : HH_TEAM_PFC; // k-steps requested across the attention chain (they hold registers there)
    auto ring_prologue = [&](const int hh_) __attribute__((always_inline)) {
        const unsigned wp_ = WB.qkv + (unsigned)((((h0 + 2 * hh_ + tm) & 7) * 4 + wave) * 16) * 6 * 1024;
#pragma unroll
        for (int p = 0; p < PFC; ++p)
#pragma unroll
            for (int c = 0; c < 6; ++c) wf[p][c] = ldb(WB, wp_ + (p * 6 + c) * 1024, uoff);
    };
    ring_prologue(0);
#pragma unroll 1
    for (int hh = 0; hh < 4; ++hh) {
        const int h = (h0 + 2 * hh + tm) & 7;
        // ---------------- head h, every row block, 16 k-steps over X: feature blocks w & ~1, (w & ~1) + 1 of q (even wavefronts) or k
        // (odd) in aq, ak, and v feature block w in av ----------------
        f32x4 aq[NRB], ak[NRB], av[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) { aq[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; ak[rb] = aq[rb]; av[rb] = aq[rb]; }
        const unsigned wp = WB.qkv + (unsigned)((h * 4 + wave) * 16) * 6 * 1024;
        f32x4 bq_pf, bk_pf; float bv_pf; // this head's biases: requested under the last k-step instead of behind the loop
        constexpr int NXB = XDB ? 2 : 1;
        bf16x8 xh[NXB][NRB], xl[NXB][NRB];
        if (XDB) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                xh[0][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + rb * 1024 + loff);
                xl[0][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + X_PLANE + rb * 1024 + loff);
            }
        }
        if (XM == 2) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) xl[0][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + X_PLANE + rb * 1024 + loff);
        }
        // the rest of the ring's head start (steps PFC .. PF-2), requested here rather than across the chain
#pragma unroll
        for (int p = PFC; p < PF - 1; ++p)
#pragma unroll
            for (int c = 0; c < 6; ++c) wf[p][c] = ldb(WB, wp + (p * 6 + c) * 1024, uoff);
        static_assert(16 % PF == 0, "the ring position must be compile-time inside the unrolled group");
        // a group of PF k-steps; the last group is a separate instance whose prefetches end at compile time (a run-time test would
        // cut the k-step into basic blocks and the loads could no longer be interleaved with the MFMAs)
        auto group = [&](const int k4, auto tail_c) __attribute__((always_inline)) {
        constexpr bool TAIL = decltype(tail_c)::value;
        static_for<PF>([&](auto ku_c) __attribute__((always_inline)) {
            constexpr int ku = decltype(ku_c)::value;
            const int ks = k4 + ku;
            {
                // prefetch k-step ks + PF - 1 into the ring slot consumed last iteration.  The scheduling barrier pins the issue point:
                // left alone, the scheduler sinks these loads next to their use.
                if (!TAIL || ku == 0) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) wf[(ku + PF - 1) % PF][c] = ldb(WB, wp + ((ks + PF - 1) * 6 + c) * 1024, uoff);
                }
                if (XM == 2) {
                    // ONE set of X fragments, refilled in place: the lo plane is only used by the first third of the step (w_hi . x_lo),
                    // the hi plane by the other two.  x_hi of THIS step is read during the first third (its registers died with the previous
                    // step), x_lo of the NEXT step during the second third (its registers died with the first).  Same latency hiding as two
                    // sets, 8 * NRB registers less -- they pay for a deeper weight ring.
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) xh[0][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + (ks * RB + rb) * 1024 + loff);
                } else if (XDB) {
                    if (!TAIL || ku + 1 < PF) {
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb) {
                            xh[(ku + 1) & (NXB - 1)][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + ((ks + 1) * RB + rb) * 1024 + loff);
                            xl[(ku + 1) & (NXB - 1)][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + X_PLANE + ((ks + 1) * RB + rb) * 1024 + loff);
                        }
                    }
                } else {
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) {
                        xh[0][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + (ks * RB + rb) * 1024 + loff);
                        xl[0][rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + X_PLANE + (ks * RB + rb) * 1024 + loff);
                    }
                }
                if (XM == 0) __builtin_amdgcn_sched_barrier(0); // this step's own X fragments: nothing to interleave them with
            }
            if (TAIL && ku == PF - 1) {
                // aq / ak: the two feature blocks of this wavefront's pair, of q (even wavefronts) or k (odd) -- see bake_qkv_kernel
                const unsigned bo = (unsigned)(h * 64 + 16 * wave) * 4, bp = (unsigned)((wave & 1) * 512 + h * 64 + 16 * (wave & ~1)) * 4;
                bq_pf = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(BB.rs, 16u * (unsigned)g, bp, 0));
                bk_pf = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(BB.rs, 16u * (unsigned)g, bp + 64u, 0));
                bv_pf = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(BB.rs, 4u * (unsigned)i, bo + 4096u, 0));
            }
            const bf16x8 *w6 = wf[ku]; // pair block A hi, lo, pair block B hi, lo (q in even wavefronts, k in odd), v hi, v lo
            const bf16x8 *xhc = xh[ku & (NXB - 1)], *xlc = xl[ku & (NXB - 1)];
            if (XM == 2) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    aq[rb] = mfma(w6[0], xl[0][rb], aq[rb]);
                    ak[rb] = mfma(w6[2], xl[0][rb], ak[rb]);
                    av[rb] = mfma(xl[0][rb], w6[4], av[rb]);
                }
                bf16x8 xn[NRB];
                if (!TAIL || ku + 1 < PF) {
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) xn[rb] = *reinterpret_cast<const bf16x8 *>(lds + LDS_X + X_PLANE + ((ks + 1) * RB + rb) * 1024 + loff);
                }
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    aq[rb] = mfma(w6[1], xh[0][rb], aq[rb]);
                    ak[rb] = mfma(w6[3], xh[0][rb], ak[rb]);
                    av[rb] = mfma(xh[0][rb], w6[5], av[rb]);
                }
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    aq[rb] = mfma(w6[0], xh[0][rb], aq[rb]);
                    ak[rb] = mfma(w6[2], xh[0][rb], ak[rb]);
                    av[rb] = mfma(xh[0][rb], w6[4], av[rb]);
                }
                if (!TAIL || ku + 1 < PF) {
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) xl[0][rb] = xn[rb];
                }
            } else {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                aq[rb] = mfma(w6[1], xhc[rb], aq[rb]);
                ak[rb] = mfma(w6[3], xhc[rb], ak[rb]);
                av[rb] = mfma(xhc[rb], w6[5], av[rb]);
            }
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                aq[rb] = mfma(w6[0], xlc[rb], aq[rb]);
                ak[rb] = mfma(w6[2], xlc[rb], ak[rb]);
                av[rb] = mfma(xlc[rb], w6[4], av[rb]);
            }
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                aq[rb] = mfma(w6[0], xhc[rb], aq[rb]);
                ak[rb] = mfma(w6[2], xhc[rb], ak[rb]);
                av[rb] = mfma(xhc[rb], w6[4], av[rb]);
            }
            }
            // Issue order inside the k-step: one memory instruction after every few MFMAs, so that it issues in the shadow of a running
            // MFMA.  Issued as one burst at the top of the step (the plain sched_barrier version) the 6 + 2*NRB memory instructions
            // cost ~170 cycles per step during which this wavefront keeps the matrix pipe empty.
            if (XM == 2) {
                // x_hi reads under the first MFMAs, then the weight loads, then (once the first third is through) next step's x_lo reads
                constexpr int NM = 9 * NRB, NV = !TAIL || ku == 0 ? 6 : (ku == PF - 1 ? 3 : 0), NL = (!TAIL || ku + 1 < PF) ? NRB : 0;
#pragma unroll
                for (int d = 0; d < NRB; ++d) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                constexpr int GAP = 3 * NRB > NRB + NV ? 3 * NRB - NRB - NV : 0; // the rest of the first third
                __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
#pragma unroll
                for (int d = 0; d < NL; ++d) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NM - NRB - NV - GAP - NL, 0);
            } else {
                constexpr int NM = 9 * NRB, ND = XDB && (!TAIL || ku + 1 < PF) ? 2 * NRB : 0, NV = !TAIL || ku == 0 ? 6 : (ku == PF - 1 ? 3 : 0);
                constexpr int A = NM >= 12 + ND ? 2 : 1;                 // MFMAs in front of each weight load
                constexpr int B = (NM - NV * A) / (ND ? ND : 1);         // MFMAs in front of each X fragment read (unused when ND == 0)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    __builtin_amdgcn_sched_group_barrier(0x008, A, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    __builtin_amdgcn_sched_group_barrier(0x008, B, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NM - NV * A - ND * B, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        };
#pragma unroll 1
        for (int k4 = 0; k4 < 16 - PF; k4 += PF) group(k4, std::false_type{});
        group(16 - PF, std::true_type{});
        if (hh + 1 < 4) ring_prologue(hh + 1); // in flight during this head's attention chain
        HH_T(2);
        // the attention chain is short, latency bound and holds the shared scratch: let it win the issue arbitration against the
        // partner wavefront's q.k.v loop on this SIMD (which fills whatever is left)
        __builtin_amdgcn_s_setprio(3);
        {
            const f32x4 bq = bq_pf, bk = bk_pf;
            const float bv = bv_pf;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                aq[rb] += bq; ak[rb] += bk;
#pragma unroll
                for (int q = 0; q < 4; ++q) av[rb][q] += bv;
            }
            if (TRAIN) {
                // q (even wavefronts) / k (odd): lane (i, g) holds features 16 (w & ~1) + 4g .. +3 (aq) and 16 more (ak) of row 16 rb + i;
                // v (normal form): rows 16 rb + 4g + r of feature 16w + i
                const int c = h * 64 + 16 * wave, cp = (wave & 1) * 512 + h * 64 + 16 * (wave & ~1);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    st4(o_qkv, (rb * 16 + i) * 1536 + cp + 4 * g, aq[rb]);
                    st4(o_qkv, (rb * 16 + i) * 1536 + cp + 16 + 4 * g, ak[rb]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) st1(o_qkv, (rb * 16 + 4 * g + r) * 1536 + 1024 + c + i, av[rb][r]);
                }
            }
        }
        // A: the scratch may be rewritten -- shared region: once every wavefront of the previous turn has finished with it; own region:
        // once the team's previous head's P (H0) and O (H1) fragments have been consumed
        if (L::TURNS) wait_turn(lds, LDS_CTR, 2 * hh + tm);
        else team_barrier(lds, bar, bar_target, lane);
        HH_T(3);
        // Q -> H0 (even wavefronts), K -> H1 (odd): lane (i, g) holds head features 32 s + 4g + r (aq) and 32 s + 16 + 4g + r (ak), s = w >> 1,
        // of its rows = all 8 contraction entries of lane group g in k-step s: whole fragment entries, 16-byte stores with no two lanes
        // of a store group on one bank (8-byte halves from two wavefronts at a 16-byte lane stride were 2-way conflicts)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const Split8 s = split8(aq[rb], ak[rb]);
            char *const dst = ((wave & 1) ? H1 : H0) + ((wave >> 1) * RB + rb) * 1024 + loff;
            *reinterpret_cast<bf16x8 *>(dst) = s.hi;
            *reinterpret_cast<bf16x8 *>(dst + H_PLANE) = s.lo;
        }
        team_barrier(lds, bar, bar_target, lane); // B
        HH_T(4);
        // this head's out_proj∘spatial_linear fragments: in flight during the attention phases, consumed after barrier E
        bf16x8 wos[2][4][2];
        {
            const unsigned op = WB.os + (unsigned)((h * 4 + wave) * 2) * 8 * 1024;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wos[ks][j][0] = ldb(WB, op + ((ks * 4 + j) * 2 + 0) * 1024, uoff);
                    wos[ks][j][1] = ldb(WB, op + ((ks * 4 + j) * 2 + 1) * 1024, uoff);
                }
        }
        // ---------------- S^T = K Q^T for query block `wave`, masked softmax over the keys of the query's env ----------------
        float p[2 * NKS][4]; // [jb][r]
#pragma unroll
        for (int jb = 0; jb < 2 * NKS; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[jb][r] = 0.0f;
        if (wave < NRB) {
            f32x4 s[NRB];
#pragma unroll
            for (int jb = 0; jb < NRB; ++jb) s[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 qh = *reinterpret_cast<const bf16x8 *>(H0 + (ks * RB + wave) * 1024 + loff);
                const bf16x8 ql = *reinterpret_cast<const bf16x8 *>(H0 + H_PLANE + (ks * RB + wave) * 1024 + loff);
#pragma unroll
                for (int jb = 0; jb < NRB; ++jb) {
                    const bf16x8 kh = *reinterpret_cast<const bf16x8 *>(H1 + (ks * RB + jb) * 1024 + loff);
                    const bf16x8 kl = *reinterpret_cast<const bf16x8 *>(H1 + H_PLANE + (ks * RB + jb) * 1024 + loff);
                    s[jb] = mfma(kl, qh, s[jb]);
                    s[jb] = mfma(kh, ql, s[jb]);
                    s[jb] = mfma(kh, qh, s[jb]);
                }
            }
            if (TRAIN) { // the rollout folds 1/sqrt(64) into the q weights
#pragma unroll
                for (int jb = 0; jb < NRB; ++jb) s[jb] *= W.qscale;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int jb = 0; jb < NRB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((vmask >> (jb * 4 + r)) & 1u) mx = fmaxf(mx, s[jb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.0f;
#pragma unroll
            for (int jb = 0; jb < NRB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = ((vmask >> (jb * 4 + r)) & 1u) ? __builtin_amdgcn_exp2f((s[jb][r] - mx) * 1.44269504088896340736f) : 0.0f;
                    p[jb][r] = e; sum += e;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
#pragma unroll
            for (int jb = 0; jb < NRB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[jb][r] *= inv;
        }
        HH_T(5);
        // (no barrier here: P block `wave` goes to exactly the slots of Q block `wave`, which only this wavefront has read)
        HH_T(6);
        if (wave < NRB) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const Split8 sp = split8(f32x4{p[2 * ks][0], p[2 * ks][1], p[2 * ks][2], p[2 * ks][3]},
                                         f32x4{p[2 * ks + 1][0], p[2 * ks + 1][1], p[2 * ks + 1][2], p[2 * ks + 1][3]});
                *reinterpret_cast<bf16x8 *>(H0 + (ks * RB + wave) * 1024 + loff) = sp.hi;
                *reinterpret_cast<bf16x8 *>(H0 + H_PLANE + (ks * RB + wave) * 1024 + loff) = sp.lo;
            }
        }
        team_barrier(lds, bar, bar_target, lane); // D
        HH_T(7);
        // ---------------- O^T = V^T P^T: value features 16w..16w+15 of the head for every query block ----------------
        f32x4 o[NRB];
#pragma unroll
        for (int ib = 0; ib < NRB; ++ib) o[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const Split8 vf = split8(av[2 * ks], 2 * ks + 1 < NRB ? av[2 * ks + 1 < NRB ? 2 * ks + 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int ib = 0; ib < NRB; ++ib) {
                const bf16x8 ph = *reinterpret_cast<const bf16x8 *>(H0 + (ks * RB + ib) * 1024 + loff);
                const bf16x8 pl = *reinterpret_cast<const bf16x8 *>(H0 + H_PLANE + (ks * RB + ib) * 1024 + loff);
                o[ib] = mfma(vf.lo, ph, o[ib]);
                o[ib] = mfma(vf.hi, pl, o[ib]);
                o[ib] = mfma(vf.hi, ph, o[ib]);
            }
        }
#pragma unroll
        for (int ib = 0; ib < NRB; ++ib) {
            if (TRAIN) st4(o_attn, (ib * 16 + i) * 512 + h * 64 + 16 * wave + 4 * g, o[ib]);
            bf16x4 oh, ol;
#pragma unroll
            for (int q = 0; q < 4; ++q) { __bf16 a, b; split1(o[ib][q], a, b); oh[q] = a; ol[q] = b; }
            const int off = ((wave >> 1) * RB + ib) * 1024 + loff + 8 * (wave & 1);
            *reinterpret_cast<bf16x4 *>(H1 + off) = oh;
            *reinterpret_cast<bf16x4 *>(H1 + H_PLANE + off) = ol;
        }
        HH_T(8);
        team_barrier(lds, bar, bar_target, lane); // E
        HH_T(9);
        // ---------------- out += O Wos[:, head h]^T: feature blocks 4w..4w+3 ----------------
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 oh[NRB], ol[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                oh[rb] = *reinterpret_cast<const bf16x8 *>(H1 + (ks * RB + rb) * 1024 + loff);
                ol[rb] = *reinterpret_cast<const bf16x8 *>(H1 + H_PLANE + (ks * RB + rb) * 1024 + loff);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc_os[j][rb] = mfma(wos[ks][j][1], oh[rb], acc_os[j][rb]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc_os[j][rb] = mfma(wos[ks][j][0], ol[rb], acc_os[j][rb]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc_os[j][rb] = mfma(wos[ks][j][0], oh[rb], acc_os[j][rb]);
            if (L::TURNS && ks == 1) finish_turn(lds, LDS_CTR, lane); // the O fragments are in registers: the scratch may go to the other team
        }
        __builtin_amdgcn_s_setprio(1); // back to the loop priority (0 / 2 and alternating the preference between the teams measured the same)
        HH_T(10);
    }
    // ---------------- out_sp = relu(out_team0 + out_team1 + b): each team finishes two of its four feature blocks ----------------
    f32x4 bfin[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) bfin[jj] = *reinterpret_cast<const f32x4 *>(W.os_b + (4 * wave + 2 * tm + jj) * 16 + 4 * g);
    __syncthreads(); // both teams are through their heads: X and the scratch regions are dead
    HH_T(11);
    {
        char *const xch = lds + LDS_X;
        if (tm == 0) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) *reinterpret_cast<f32x4 *>(xch + ((wave * 2 + jj) * RB + rb) * 1024 + loff) = acc_os[2 + jj][rb];
        } else {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) *reinterpret_cast<f32x4 *>(xch + XCH_TEAM + ((wave * 2 + jj) * RB + rb) * 1024 + loff) = acc_os[jj][rb];
        }
        __syncthreads();
        if (tm == 0) finish_rows<NRB, 0, RB>(t, acc_os, xch + XCH_TEAM, W, out_sp, lane, wave, bfin);
        else finish_rows<NRB, 2, RB>(t, acc_os, xch, W, out_sp, lane, wave, bfin);
    }
    HH_T(12);
#ifdef HH_TIMING
    if (g_hh_tim && wave == 0 && lane == 0) {
        long long *d = g_hh_tim + ((size_t)blockIdx.x * 2 + tm) * 20;
        for (int k = 0; k < 16; ++k) d[k] += tacc[k];
        d[16] += 1; d[17] += t.nrows; d[18] += NRB;
    }
#endif
    // no barrier here: the next tile's e0 fragments go to team 0's scratch (dead since the barrier above), and X is rewritten
    // only behind the next tile's first __syncthreads()
}

} // namespace team

#ifndef HH_TEAM_PF3
#define HH_TEAM_PF3 2   // prefetch depth (k-steps) of the bodies of <= 3 row blocks (4 measured the same; 2 leaves registers for the cross-head prefetch)
#endif
#ifndef HH_TEAM_PF4
#define HH_TEAM_PF4 2   // prefetch depth of the 4-row-block body (register budget: 256)
#endif
#ifndef HH_TEAM_XM3
#define HH_TEAM_XM3 1   // X fragment mode of the q.k.v loop (tile_body) for <= 3 row blocks
#endif
#ifndef HH_TEAM_XM4
#define HH_TEAM_XM4 1
#endif

// TRAIN: also write e0 / x / qkv / attn (HhFusedWeights::*_out) and scale the scores by qscale -- the training forward (cn_hh_block_fwd)
template <bool TRAIN>
__global__ __launch_bounds__(512, 2) void hh_fused_kernel(int E, int H, int D, const float *__restrict__ se, const float *__restrict__ det,
                                                          int *row_off, unsigned long long *live_total, HhFusedWeights W, float *__restrict__ out_sp,
                                                          const int32_t *__restrict__ plan, unsigned long long *stamp)
{
    const CnStampScope stamp_scope(TRAIN ? nullptr : stamp); // (the training variant has no registers to spare: rocprof times it)
    extern __shared__ __attribute__((aligned(16))) char lds[];
#ifdef HH_TIMING
    const long long k_c0 = clock64(), k_r0 = wall_clock64(); // shader-clock cycles and the constant 100 MHz counter: their ratio is the clock
#endif
    // a row plan made with the observation (row_plan.h) replaces the row-offset scan and the contiguous tile splitter below
    bool planned = !TRAIN && det && rp_usable(plan, E, H) && plan[1] == (int)gridDim.x;
    if (planned) {
        // ... and it must be the plan of THIS observation: a caller that stepped / reset the batch in between, or changed
        // detected_human_num (the worst-case leg of bench.py), still holds a complete plan of the right shape.  Every workgroup
        // compares every env's row count in the plan with clamp(det) -- 3 coalesced loads per env, 8 envs per thread at 4096 envs --
        // and all of them come to the same verdict; a stale plan means the scan path below, exactly as without a plan.
        const int32_t *pro = plan + rp_off_rowoff();
        int bad = 0;
        for (int i = (int)threadIdx.x; i < E; i += (int)blockDim.x) {
            int nd = (int)det[i]; nd = nd < 1 ? 1 : (nd > H ? H : nd);
            bad |= (pro[i + 1] - pro[i]) ^ nd;
        }
        // (the verdict is combined through the dynamic LDS: __syncthreads_or would add a static __shared__ word, and this kernel's
        // dynamic allocation is the whole 160 KB -- the launch attribute is refused when static + dynamic exceed it)
        volatile int *flag = reinterpret_cast<volatile int *>(lds);
        if (threadIdx.x == 0) *flag = 0;
        __syncthreads();
        if (__ballot(bad != 0) != 0ull && (threadIdx.x & 63) == 0) *flag = 1;
        __syncthreads();
        planned = __builtin_amdgcn_readfirstlane(*flag) == 0;
        __syncthreads(); // everybody has read the verdict before the LDS is reused
    }
    if (det && !planned) row_offsets_prologue<512>(E, H, det, row_off, live_total, lds);
    if (!TRAIN && stamp && blockIdx.x == 0 && threadIdx.x == 0) stamp[1] = (unsigned long long)(planned ? plan[3] : row_off[E]);
#ifdef HH_TIMING
    const long long k_c1 = clock64();
#endif
    __builtin_amdgcn_s_setprio(1); // above the simulator's side-stream wavefronts; the attention chains go to 3 (tile_body)
    const int lane = threadIdx.x & 63, w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = w8 & 3, tm = w8 >> 2;
    // tiles of this workgroup: the plan's tiles c, c + NW, ... -- or, without a plan, the chunk of consecutive envs [boundary(c), boundary(c + 1))
    // cut by next_tile
    int e = 0, e_end = 0, chunk_end_row = 0, n_plan = 0;
    PlanPre pre{0, 0};
    if (planned) {
        if (live_total && blockIdx.x == 0 && threadIdx.x == 0) *live_total += (unsigned long long)plan[3];
        n_plan = plan[2];
        pre = plan_prefetch(plan, E, (int)blockIdx.x, lane);
        // the caller's row_off array gets the plan's offsets (it is what the kernels behind this one and the debug taps index by)
        for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i <= E; i += (int)(gridDim.x * blockDim.x)) row_off[i] = plan[rp_off_rowoff() + i];
    } else {
        const int total = ld_ro(row_off + E);
        e = chunk_boundary(row_off, E, total, (int)blockIdx.x, (int)gridDim.x, lane);
        e_end = chunk_boundary(row_off, E, total, (int)blockIdx.x + 1, (int)gridDim.x, lane);
        if (e >= e_end) return;
        chunk_end_row = ld_ro(row_off + e_end);
    }
    int tile_ord = 0;
    const team::WeightBuf WB = team::make_weight_buf(W);
    for (;;) {
        TileCtx t;
        if (planned) {
            if (tile_ord >= n_plan) break;
            const PlanPre cur = pre;
            ++tile_ord;
            if (tile_ord < n_plan) pre = plan_prefetch(plan, E, (int)blockIdx.x + tile_ord * (int)gridDim.x, lane);
            t = plan_tile(cur, H, tile_ord - 1, lane);
            if (t.nrows == 0) continue; // fewer envs than tiles
        } else {
            if (e >= e_end) break;
            t = next_tile<team::FR>(row_off, e, e_end, chunk_end_row, tile_ord++, lane);
            t.my_src = (t.e_lo + t.my_env) * H + (lane - t.my_start);
            e += t.n_env;
        }
        const int nrb = (t.nrows + 15) >> 4;
        switch (nrb) {
        case 1: team::tile_body<1, HH_TEAM_PF3, HH_TEAM_XM3, TRAIN>(t, H, D, se, W, WB, out_sp, lds, lane, wave, tm); break;
        case 2: team::tile_body<2, HH_TEAM_PF3, HH_TEAM_XM3, TRAIN>(t, H, D, se, W, WB, out_sp, lds, lane, wave, tm); break;
        case 3: team::tile_body<3, HH_TEAM_PF3, HH_TEAM_XM3, TRAIN>(t, H, D, se, W, WB, out_sp, lds, lane, wave, tm); break;
        default: team::tile_body<4, HH_TEAM_PF4, HH_TEAM_XM4, TRAIN>(t, H, D, se, W, WB, out_sp, lds, lane, wave, tm); break;
        }
    }
#ifdef HH_TIMING
    if (g_hh_tim && (threadIdx.x & 255) == 0) { // whole-kernel span of wavefront 0 of each team: [13] cycles, [14] 10 ns ticks, [15] prologue cycles
        long long *d = g_hh_tim + ((size_t)blockIdx.x * 2 + (threadIdx.x >> 8)) * 20;
        d[13] += clock64() - k_c0; d[14] += wall_clock64() - k_r0; d[15] += k_c1 - k_c0;
    }
#endif
}

// ---- weight baking: fp32 row-major [N,K] -> bf16 hi/lo MFMA fragments in the order the kernel streams them ----
// fragment entry (feature block fb, k-step ks, lane, u) = W[fb*16 + (lane & 15)][32*ks + koff(lane >> 4, u)]
__device__ __forceinline__ int koff(int g, int u, bool perm) { return perm ? 16 * (u >> 2) + 4 * g + (u & 3) : 8 * g + u; }

__device__ __forceinline__ void bake_emb2(int idx, const float *__restrict__ w, __bf16 *__restrict__ out) // idx: one fragment entry element
{
    if (idx >= 512 * 128) return;
    const int u = idx & 7, lane = (idx >> 3) & 63, j = (idx >> 9) & 7, ks = (idx >> 12) & 3, wv = idx >> 14;
    const int fb = 8 * wv + j;
    const float x = w[(size_t)(fb * 16 + (lane & 15)) * 128 + 32 * ks + koff(lane >> 4, u, false)];
    const __bf16 hi = (__bf16)x;
    const size_t base = ((((size_t)wv * 4 + ks) * 8 + j) * 2) * 512 + lane * 8 + u;
    out[base] = hi;
    out[base + 512] = (__bf16)(x - (float)hi);
}
__device__ __forceinline__ void bake_qkv(int idx, const float *__restrict__ w, __bf16 *__restrict__ out)
{
    if (idx >= 1536 * 512) return;
    const int u = idx & 7, lane = (idx >> 3) & 63;
    int rest = idx >> 9;                 // ((h*4 + wv)*16 + ks)*3 + j
    const int j = rest % 3; rest /= 3;
    const int ks = rest & 15, wv = (rest >> 4) & 3, h = rest >> 6;
    // wavefront wv streams TWO 16-feature blocks of ONE of q / k (even wv: q blocks wv, wv + 1; odd wv: k blocks wv - 1, wv) and its own
    // v block: a lane then holds all 8 entries of its lane group in k-step wv >> 1 of a Q or K fragment and stores them in one piece
    const int sec = j == 2 ? 2 : (wv & 1), blk = j == 2 ? wv : (wv & ~1) + j;
    const float x = w[(size_t)(sec * 512 + h * 64 + 16 * blk + (lane & 15)) * 512 + 32 * ks + koff(lane >> 4, u, true)];
    const __bf16 hi = (__bf16)x;
    const size_t base = (((((size_t)h * 4 + wv) * 16 + ks) * 3 + j) * 2) * 512 + lane * 8 + u;
    out[base] = hi;
    out[base + 512] = (__bf16)(x - (float)hi);
}
__device__ __forceinline__ void bake_os(int idx, const float *__restrict__ w, __bf16 *__restrict__ out)
{
    if (idx >= 256 * 512) return;
    const int u = idx & 7, lane = (idx >> 3) & 63, j = (idx >> 9) & 3, ks = (idx >> 11) & 1, wv = (idx >> 12) & 3, h = idx >> 14;
    const float x = w[(size_t)((4 * wv + j) * 16 + (lane & 15)) * 512 + h * 64 + 32 * ks + koff(lane >> 4, u, true)];
    const __bf16 hi = (__bf16)x;
    const size_t base = (((((size_t)h * 4 + wv) * 2 + ks) * 4 + j) * 2) * 512 + lane * 8 + u;
    out[base] = hi;
    out[base + 512] = (__bf16)(x - (float)hi);
}

__global__ void bake_all_kernel(const float *__restrict__ emb2_w, const float *__restrict__ qkv_w, const float *__restrict__ os_w, __bf16 *__restrict__ emb2_frag,
                                __bf16 *__restrict__ qkv_frag, __bf16 *__restrict__ os_frag)
{
    constexpr int B1 = 512 * 128 / 256, B2 = B1 + 1536 * 512 / 256;
    const int b = blockIdx.x;
    if (b < B1) bake_emb2(b * 256 + (int)threadIdx.x, emb2_w, emb2_frag);
    else if (b < B2) bake_qkv((b - B1) * 256 + (int)threadIdx.x, qkv_w, qkv_frag);
    else bake_os((b - B2) * 256 + (int)threadIdx.x, os_w, os_frag);
}

} // namespace

int hh_fused_bake(const float *emb2_w, const float *qkv_w, const float *os_w, void *emb2_frag, void *qkv_frag, void *os_frag, hipStream_t st)
{
    // one launch for the three images (the training forward bakes them every optimiser step): blocks 0..255 emb2, 256..3327 q.k.v, the rest os
    hipLaunchKernelGGL(bake_all_kernel, dim3((512 * 128 + 1536 * 512 + 256 * 512) / 256), dim3(256), 0, st, emb2_w, qkv_w, os_w, (__bf16 *)emb2_frag, (__bf16 *)qkv_frag,
                       (__bf16 *)os_frag);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

#ifdef HH_TIMING
extern "C" int cn_hh_fused_set_timing(long long *buf)
{
    CN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_hh_tim), &buf, sizeof(buf)));
    return CN_OK;
}
#endif
#ifdef HH_DEBUG
extern "C" int cn_hh_fused_set_debug(int *buf)
{
    CN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_hh_dbg), &buf, sizeof(buf)));
    return CN_OK;
}
#endif

int hh_fused_forward(int E, int H, int D, const float *spatial_edges, const float *det, int *row_off, unsigned long long *live_total,
                     const HhFusedWeights &w, float *out_sp, hipStream_t st, const int32_t *row_plan)
{
    unsigned long long *stamp = cn_stamp_slot(CN_K_HH_FUSED);
    static thread_local int attr_dev = -1; // the opt-in above 64 KB of dynamic LDS is per device
    int dev = 0;
    CN_HIP(hipGetDevice(&dev));
    if (dev != attr_dev) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hh_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, team::LDS_BYTES));
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hh_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, team::LDS_BYTES));
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hh_fused_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, wide::LDS_BYTES));
        attr_dev = dev;
    }
    // one workgroup per CU (the LDS footprint admits exactly one); small batches get fewer so that a chunk is >= one row block
    const int grid = rp_workgroups(E, H);
    // an env must fit one tile: the two-team kernel holds 63 rows, the wide one 64 (CN_HH_WIDE=1 forces the latter: A/B measurements)
    static int force_wide = -1;
    if (force_wide < 0) { const char *v = getenv("CN_HH_WIDE"); force_wide = v ? atoi(v) : 0; }
    // (round 6: crowds of 49..63 humans also go through the two-team kernel -- an env of <= 63 rows fits its 4-row-block layout; measured at 50
    // randomised humans x 8192 envs: 1.39 -> 1.14 ms beside the simulator's side-stream kernels.  CN_HH_TEAM_MAX=48 restores the round-5 split.)
    static int team_max = -1;
    if (team_max < 0) { const char *v = getenv("CN_HH_TEAM_MAX"); team_max = v ? atoi(v) : team::FR; }
    const bool train = w.e0_out != nullptr;
    if (train) {
        CN_REQUIRE(H <= 48 && w.x_out && w.qkv_out && w.attn_out, "hh_fused_forward: the training outputs need H <= 48 and all four buffers");
        hipLaunchKernelGGL(hh_fused_kernel<true>, dim3(grid), dim3(512), team::LDS_BYTES, st, E, H, D, spatial_edges, det, row_off, live_total, w, out_sp,
                           (const int32_t *)nullptr, stamp);
    } else if (H > team_max || force_wide > 0)
        hipLaunchKernelGGL(hh_fused_wide_kernel, dim3(grid), dim3(256), wide::LDS_BYTES, st, E, H, D, spatial_edges, det, row_off, live_total, w, out_sp, stamp);
    else
        hipLaunchKernelGGL(hh_fused_kernel<false>, dim3(grid), dim3(512), team::LDS_BYTES, st, E, H, D, spatial_edges, det, row_off, live_total, w, out_sp,
                           row_plan, stamp);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
