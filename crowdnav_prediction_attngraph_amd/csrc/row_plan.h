// Row plan: which envs share a tile of the fused human-human kernel (hh_fused.hip), built once per step on the device.
//
// The kernel walks the live (env, human) rows in tiles of whole envs, at most 63 rows each, and its cost per workgroup is
// (tiles) x (a pass over the 3.9 MB weight stream) + (16-row blocks) x (the MFMA work of a block); a launch lasts as long as its slowest
// workgroup.  Cutting the env sequence into CONTIGUOUS runs (the kernel's own splitter, still there as the fallback) fills the 16-row
// blocks to ~92 %: at 4096 envs x 5.8 detected humans a third of the workgroups end up with 7 blocks and the rest, with 6, wait for them.
// Packing envs into tiles regardless of their order fills them to 97-99 %: every workgroup gets 2 tiles of 45..48 rows = 6 blocks.
//
// The packing is longest-processing-time-first by size class, evaluated with a water-filling search instead of a priority queue so that
// ONE wavefront does it in ~50 us (4096 envs): for the envs with v rows (v = H .. 1) find the highest level L that the bins can be filled
// up to with at most m_v envs, give bin b floor((L - load_b) / v) of them, hand the remainder to the first bins that would gain one at
// level L + 1 -- with the 64 lanes as the bins, and each lane then placing its envs on the emptiest of its own tiles (build / fill below).
// The wavefront also writes the row offsets (prefix sum of the row counts by env index), so the consuming kernel needs no scan of its
// own.  It runs as workgroup 0 of the simulator's ORCA lane kernel (env_sim.hip), i.e. beside work that is on the step's critical path
// anyway (that kernel: 49 -> 51 us).  Everything it touches more than once lives in registers or LDS (a global round trip costs a lone
// wavefront 1-2 us, a ds_bpermute 100+ cycles: the scans and reductions are DPP).
//
// Layout (int32 words): header | row_off[E + 1] | tile_cnt[RP_TMAX] | items[RP_TMAX][64], item = env | rows << 16, a tile's list ends with
// a zero item (or at 64).
// Tile t belongs to workgroup t % NW; a workgroup walks tiles t = c, c + NW, ...
#pragma once
#include <stdint.h>

#define RP_MAGIC 0x52504c4e
#define RP_TMAX 1024   // tiles the builder can hold: 16 per lane, in registers
#define RP_EMAX 4096   // envs the builder can hold: 64 per lane, in registers
#define RP_HMAX 32     // rows per env (the lane kernel that hosts the builder stops at 32 agents per env)
#define RP_HDR 8       // [0] RP_MAGIC when valid [1] NW [2] tiles per workgroup [3] live rows [4] E [5] H [6] tiles [7] -

__host__ __device__ inline int rp_off_rowoff() { return RP_HDR; }
__host__ __device__ inline int rp_off_tcnt(int E) { return RP_HDR + ((E + 1 + 3) & ~3); }
__host__ __device__ inline int rp_off_items(int E) { return rp_off_tcnt(E) + RP_TMAX; }
__host__ __device__ inline int rp_words(int E) { return rp_off_items(E) + RP_TMAX * 64; }
// workgroups of the consuming kernel for a batch of E envs x H rows (hh_fused_forward uses the same rule)
__host__ __device__ inline int rp_workgroups(int E, int H)
{
    long long g = ((long long)E * H + 15) / 16;
    return g > 256 ? 256 : (g < 1 ? 1 : (int)g);
}

#ifdef __HIPCC__
// the consumers' test: a plan that was completed (the builder writes the magic last) for a batch of exactly this shape
__device__ __forceinline__ bool rp_usable(const int32_t *plan, int E, int H)
{
    return plan && plan[0] == RP_MAGIC && plan[4] == E && plan[5] == H;
}

namespace rowplan {

// LDS of the builder: per-lane size histogram / running positions [RP_HMAX + 1][64] and the env ids in hand-out order
struct Lds {
    int tbl[(RP_HMAX + 1) * 64];
    unsigned short ids[RP_EMAX];
};

// DPP: lane i reads lane i - n of its row of 16 (row_shr) / the last lane of the previous row(s) (row_bcast); a lane without a source
// keeps `old`
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_incl_scan(int x)
{
    x += dpp<0x111>(0, x); x += dpp<0x112>(0, x); x += dpp<0x114>(0, x); x += dpp<0x118>(0, x); // inclusive within each row of 16
    x += dpp<0x142, 0xa>(0, x);                                                                 // row_bcast:15 into rows 1 and 3
    x += dpp<0x143, 0xc>(0, x);                                                                 // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ int wave_sum(int x) { return __builtin_amdgcn_readlane(wave_incl_scan(x), 63); }
__device__ __forceinline__ int wave_min(int x)
{
    int y;
    y = dpp<0x111>(x, x); x = y < x ? y : x; y = dpp<0x112>(x, x); x = y < x ? y : x;
    y = dpp<0x114>(x, x); x = y < x ? y : x; y = dpp<0x118>(x, x); x = y < x ? y : x;
    y = dpp<0x142, 0xa>(x, x); x = y < x ? y : x; y = dpp<0x143, 0xc>(x, x); x = y < x ? y : x;
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ int wave_max(int x) { return -wave_min(-x); }

#ifdef RP_TIMING
#define RP_T(k) do { if (tim && (threadIdx.x & 63) == 0) tim[k] = wall_clock64(); } while (0)
#define RP_A(k) do { const long long now_ = wall_clock64(); if (tim && (threadIdx.x & 63) == 0) tim[k] += now_ - tl_; tl_ = now_; } while (0)
#define RP_A0() long long tl_ = wall_clock64()
#else
#define RP_T(k) do {} while (0)
#define RP_A(k) do {} while (0)
#define RP_A0() do {} while (0)
#endif

// The tiles of the plan: lane l owns tiles l * TB .. l * TB + TB - 1 (TBP = TB rounded up to a power of two: the per-lane loops over the
// tiles are unrolled without guards, surplus slots hold an all-ones key and never win).  Returns false when the envs do not fit.
template <int TBP>
__device__ __forceinline__ bool fill(int H, int T, int TB, int hist, int cstart, int32_t *__restrict__ items, int32_t *__restrict__ tcnt, Lds &lds, long long *tim)
{
    const int ln = threadIdx.x & 63;
    // key of tile j of this lane: rows << 11 | envs << 4 | j -- the smallest key is the emptiest tile (ties: fewer envs, then lower j)
    unsigned key[TBP];
#pragma unroll
    for (int j = 0; j < TBP; ++j) key[j] = (j < TB && ln * TB + j < T) ? (unsigned)j : 0xffffffffu;
    int lt = 0; // rows on this lane's tiles
    const int lcap = (ln * TB < T ? ((ln + 1) * TB <= T ? TB : T - ln * TB) : 0) * 63;
    const int nact = __popcll(__ballot(lcap > 0)); // lanes that own tiles (all 64 unless the batch is tiny)
    bool bad = false;
    for (int v = H; v >= 1; --v) {
        const int m = __builtin_amdgcn_readlane(hist, v);
        if (m == 0) continue;
        const int pos0 = __builtin_amdgcn_readlane(cstart, v);
        RP_A0();
        // level 1: envs of this class per lane = water filling on the lanes' row totals.  floor(r / v) = r * ceil(2^20 / v) >> 20 for r (v - 1) < 2^20
        const unsigned inv = ((1u << 20) + (unsigned)v - 1u) / (unsigned)v;
        auto take = [&](int L) __attribute__((always_inline)) -> int {
            const int room = (L < lcap ? L : lcap) - lt;
            return room > 0 ? (int)(((unsigned)room * inv) >> 20) : 0;
        };
        // Largest level whose demand is at most m.  With every lane taking part it lies in [mean, mean + v], mean = the lanes' average
        // total AFTER this class; lanes that are already above the level (or full) move it down: then bisect from the lowest total.
        int lo = (wave_sum(lt) + m * v) / nact, hi = lo + v;
        if (wave_sum(take(lo)) > m) { hi = lo - 1; lo = wave_min(lcap ? lt : (1 << 30)); }
        for (;;) {
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (wave_sum(take(mid)) <= m) lo = mid; else hi = mid - 1;
            }
            // ended on the upper end of the bracket: make sure the level really stops there (lanes at their cap shift the mean argument)
            const int top = wave_max(lcap ? lt : 0) + v * ((m + nact - 1) / nact + 1);
            if (lo >= top || wave_sum(take(lo + 1)) > m) break;
            lo = lo + 1; hi = top;
        }
        int k = take(lo);
        const int rem = m - wave_sum(k);
        if (rem > 0) { // the remainder goes to the first lanes that gain one more at the next level
            const int gain = take(lo + 1) > k ? 1 : 0;
            const int incl = wave_incl_scan(gain);
            if (gain && incl <= rem) k += 1;
            if (__builtin_amdgcn_readlane(incl, 63) < rem) bad = true; // they do not fit: no plan, the consumer falls back
        }
        RP_A(8);
        const int inclk = wave_incl_scan(k);
        const int at = pos0 + inclk - k;
        const int kmax = wave_max(k);
        RP_A(9);
        // level 2: each env on the lane's emptiest tile.  The ids are fetched eight at a time ahead of the placement chain.
        for (int q0 = 0; q0 < kmax; q0 += 8) {
            int id8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) id8[u] = q0 + u < k ? (int)lds.ids[at + q0 + u] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (q0 + u >= kmax) break;
                const bool on = q0 + u < k;
                unsigned mn = key[0];
#pragma unroll
                for (int j = 1; j < TBP; ++j) mn = key[j] < mn ? key[j] : mn;
                const int jm = (int)(mn & 15u), slot = (int)((mn >> 4) & 127u);
                if (on) {
                    if (mn == 0xffffffffu || slot >= 64) bad = true;
                    else items[(size_t)(ln * TB + jm) * 64 + slot] = id8[u] | (v << 16);
                    const unsigned delta = ((unsigned)v << 11) + (1u << 4);
#pragma unroll
                    for (int j = 0; j < TBP; ++j) key[j] += key[j] == mn ? delta : 0u;
                }
            }
        }
        lt += k * v;
        RP_A(10);
    }
#pragma unroll
    for (int j = 0; j < TBP; ++j)
        if (j < TB && ln * TB + j < T) {
            const int cnt = (int)((key[j] >> 4) & 127u);
            tcnt[ln * TB + j] = cnt;
            if (cnt < 64) items[(size_t)(ln * TB + j) * 64 + cnt] = 0; // terminator: a consumer needs the item list only (rows == 0 ends it)
            if ((key[j] >> 11) > 63u) bad = true;
        }
    return __ballot(bad) == 0ull;
}

// One wavefront (all 64 lanes active).  det: detected_human_num of the observation the plan is for.
//
// Two levels, so that almost nothing needs the other lanes: (1) the envs of a size class are dealt to the 64 LANES by water filling on
// the lanes' row totals (one division per lane and probe); (2) each lane puts the envs it was dealt, one by one, on the emptiest of ITS
// OWN tiles (tile = lane * TB + j: at most 16 per lane, loads and counts packed into one register key each).
__device__ __forceinline__ void build(int E, int H, int NW, const float *__restrict__ det, int32_t *__restrict__ plan, Lds &lds, long long *tim = nullptr)
{
    const int ln = threadIdx.x & 63;
    int32_t *hdr = plan, *row_off = plan + rp_off_rowoff(), *tcnt = plan + rp_off_tcnt(E), *items = plan + rp_off_items(E);
    RP_T(0);
    if (ln == 0) hdr[0] = 0;
    if (H > RP_HMAX || E > RP_EMAX || (E & 3)) return;
    // ---- rows per env: lane l owns envs [CH * l, CH * l + CH) (CH = 4 * ceil(E / 256) <= 64), all in registers ----
    const int CH = ((E + 255) >> 8) << 2;
    int c[64];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float4 d = float4{0.f, 0.f, 0.f, 0.f};
        const int e0 = ln * CH + 4 * q;
        if (4 * q < CH && e0 < E) d = *reinterpret_cast<const float4 *>(det + e0); // E % 4 == 0: a float4 never straddles the end
        const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int r = (int)dd[u];
            r = r < 1 ? 1 : (r > H ? H : r); // no detected human still occupies one (dummy) row: crowd_sim_var_num.py:290-292
            c[4 * q + u] = (4 * q < CH && e0 < E) ? r : 0;
        }
    }
    // ---- row offsets (serial inside the lane, one scan across); per-lane size histogram in LDS (column `lane` is private to the lane) ----
    for (int k = ln; k < (H + 1) * 64; k += 64) lds.tbl[k] = 0;
    int mysum = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) mysum += c[i];
    const int incl0 = wave_incl_scan(mysum);
    const int total = __builtin_amdgcn_readlane(incl0, 63);
    {
        int run = incl0 - mysum;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int4 o;
            o.x = run; run += c[4 * q]; o.y = run; run += c[4 * q + 1]; o.z = run; run += c[4 * q + 2]; o.w = run; run += c[4 * q + 3];
            if (4 * q < CH && ln * CH + 4 * q < E) *reinterpret_cast<int4 *>(row_off + ln * CH + 4 * q) = o; // row_off is 16-byte aligned (RP_HDR = 8)
        }
    }
    if (ln == 0) row_off[E] = total;
#pragma unroll
    for (int i = 0; i < 64; ++i)
        if (c[i]) __hip_atomic_fetch_add(&lds.tbl[c[i] * 64 + ln], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    RP_T(1);
    // ---- hand-out order: by rows descending; inside a size class by (lane, i).  tbl[v][lane] becomes the lane's next position ----
    int hist = 0, cstart = 0; // lane v: number of envs with v rows, first position of the class
    {
        int at = 0;
        for (int v = H; v >= 1; --v) {
            const int mine = lds.tbl[v * 64 + ln];
            const int incl = wave_incl_scan(mine);
            lds.tbl[v * 64 + ln] = at + incl - mine;
            const int m = __builtin_amdgcn_readlane(incl, 63);
            if (ln == v) { hist = m; cstart = at; }
            at += m;
        }
    }
    {
        int pos[64];
#pragma unroll
        for (int i = 0; i < 64; ++i)
            pos[i] = c[i] ? __hip_atomic_fetch_add(&lds.tbl[c[i] * 64 + ln], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
#pragma unroll
        for (int i = 0; i < 64; ++i)
            if (c[i]) lds.ids[pos[i]] = (unsigned short)(ln * CH + i);
    }
    RP_T(2);
    // ---- tiles ----
    int n = (total + 62 * NW - 1) / (62 * NW);
    n = n < 1 ? 1 : n;
    const int T = n * NW;
    if (T > RP_TMAX) return;
    const int TB = (T + 63) >> 6;
    bool ok;
    if (TB <= 1) ok = fill<1>(H, T, TB, hist, cstart, items, tcnt, lds, tim);
    else if (TB <= 2) ok = fill<2>(H, T, TB, hist, cstart, items, tcnt, lds, tim);
    else if (TB <= 4) ok = fill<4>(H, T, TB, hist, cstart, items, tcnt, lds, tim);
    else if (TB <= 8) ok = fill<8>(H, T, TB, hist, cstart, items, tcnt, lds, tim);
    else ok = fill<16>(H, T, TB, hist, cstart, items, tcnt, lds, tim);
    RP_T(3);
    if (!ok) return;
    __threadfence();
    if (ln == 0) {
        hdr[1] = NW; hdr[2] = n; hdr[3] = total; hdr[4] = E; hdr[5] = H; hdr[6] = T; hdr[7] = 0;
        __threadfence();
        hdr[0] = RP_MAGIC;
    }
    RP_T(4);
}

} // namespace rowplan
#endif
