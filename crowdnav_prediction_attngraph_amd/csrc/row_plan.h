// Row plan: which envs share a tile of the fused human-human kernel (hh_fused.hip), built once per step on the device.
//
// The kernel walks the live (env, human) rows in tiles of whole envs, at most 63 rows each, and its cost per workgroup is
// (tiles) x (a pass over the 3.9 MB weight stream) + (16-row blocks) x (the MFMA work of a block); a launch lasts as long as its slowest
// workgroup.  Cutting the env sequence into CONTIGUOUS runs (the kernel's own splitter, still there as the fallback) fills the 16-row
// blocks to ~92 %: at 4096 envs x 5.8 detected humans a third of the workgroups end up with 7 blocks and the rest, with 6, wait for them.
// Packing envs into tiles regardless of their order fills them to 97-99 %: every workgroup gets 2 tiles of 45..48 rows = 6 blocks.
//
// The packing (round 4): the batch is cut into groups of E / G envs (rp_groups), group g owns a contiguous range of tiles proportional to
// its rows and is packed by ONE wavefront of its own (fill / build below): by size class, largest first, each class dealt evenly to the
// group's tiles up to every tile's exact share of the group's rows, with one prefix sum per class telling every tile which envs are its
// own.  ~28 us per wavefront stand-alone, all groups in parallel (round 3: one wavefront, longest-processing-time-first through a
// water-filling level search of ~8 DPP reductions per class and a per-env placement chain, ~50 us).  The wavefronts also write the row
// offsets (prefix sum of the row counts by env index), so the consuming kernel needs no scan of its own.  They run as the first workgroups
// of the simulator's ORCA lane kernel (env_sim.hip), i.e. beside work that is on the step's critical path anyway.  Everything they touch
// more than once lives in registers or LDS (a global round trip costs a lone wavefront 1-2 us, a ds_bpermute 100+ cycles: the scans are DPP).
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
// Wavefronts that build the plan: the batch is cut into G index ranges of E / G envs; nothing crosses groups (every wavefront reads all
// counts: it needs the other groups' totals for its row offsets and its tile range).  Groups need E to be a multiple of 256 (a group is
// then a whole number of lanes of the count scan) and at least 256 envs each; with 8 groups of 512 envs a lane holds 8 envs and one tile.
__host__ __device__ inline int rp_groups(int E)
{
    if (E <= 0 || (E & 255)) return 1;
    return E >= 2048 ? 8 : (E >= 1024 ? 4 : (E >= 512 ? 2 : 1));
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

// The tiles of a group: lane l owns tiles l * TB .. l * TB + TB - 1 (TBP = TB rounded up to a power of two: the per-lane loops over the
// tiles are unrolled without guards).  First fit, decreasing, one PREFIX SUM per size class: with the tiles in (lane, j) order, tile i has
// room for c_i = floor((cap_i - load_i) / v) envs of v rows; the class's m envs go to the tiles in order, tile i taking the envs
// [C_i, C_i + c_i) of the class (C = exclusive scan of c, clipped at m).  cap_i = the tile's share of the group's rows (see below): with
// sizes 1 .. H and plenty of small envs first fit closes every tile on its share -- in the 30 sampled steps of the bench configuration
// (tools/row_plan_groups_study.py) no env ever needed the second pass.  One scan per class instead of the ~8 water-filling probes of the
// round-3 builder (each a 6-step DPP reduction: 16-20 us of its 50), and no per-env placement chain.  Returns false when the envs do not fit.
template <int TBP>
__device__ __forceinline__ bool fill(int H, int T, int TB, int rows, int hist, int cstart, int id_base, int32_t *__restrict__ items, int32_t *__restrict__ tcnt,
                                     Lds &lds, long long *tim)
{
    const int ln = threadIdx.x & 63;
    int load[TBP], cnt[TBP];
    bool own[TBP];
#pragma unroll
    for (int j = 0; j < TBP; ++j) { load[j] = 0; cnt[j] = 0; own[j] = j < TB && ln * TB + j < T; }
    // a tile's share of the group's rows: base or base + 1 -- the shares add up to exactly `rows`, so the loads come out equal to within
    // the few rows the fit leaves open (45..48 at 46.7 rows per tile)
    const int base = rows / T, extra = rows - base * T;
    int cap0[TBP];
#pragma unroll
    for (int j = 0; j < TBP; ++j) { const int c0 = base + (ln * TB + j < extra ? 1 : 0); cap0[j] = c0 > 63 ? 63 : c0; }
    bool bad = false;
    int rot = 0; // (wave-uniform) lane the next hand-out starts at
    for (int v = H; v >= 1; --v) {
        const int m = __builtin_amdgcn_readlane(hist, v);
        if (m == 0) continue;
        const int pos0 = __builtin_amdgcn_readlane(cstart, v);
        RP_A0();
        const unsigned inv = ((1u << 20) + (unsigned)v - 1u) / (unsigned)v; // floor(r / v) = r * ceil(2^20 / v) >> 20 for r (v - 1) < 2^20
        int placed = 0; // (wave-uniform) envs of the class that have a tile
        // pass 0: dealt EVENLY -- at most ceil(m / T) envs of the class per tile, the lanes in circular order from where the previous
        // class stopped -- so that every tile gets its share of every size (filled in plain first-fit order the first tiles of a group
        // hold three large envs and its last ones forty small ones; the consumer's per-tile bookkeeping is per env); pass 1: the rest
        // wherever a tile's share has room; passes 2..: what still has no tile (mid-sized envs when the shares are nearly full) with the
        // shares raised by 1, 2, 3, ... rows -- the emptiest tiles take them first, a tile ends at most a few rows above its share
        constexpr int NPASS = 13;
        const int raise[NPASS] = {0, 0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 63};
        for (int pass = 0; pass < NPASS && placed < m; ++pass) {
            const int left = m - placed;
            const int quota = pass == 0 ? (left + T - 1) / T : 64;
            int c[TBP], lc = 0;
#pragma unroll
            for (int j = 0; j < TBP; ++j) {
                int cap = cap0[j] + raise[pass];
                cap = cap > 63 ? 63 : cap;
                int r = own[j] && load[j] < cap ? (int)(((unsigned)(cap - load[j]) * inv) >> 20) : 0;
                r = r < 63 - cnt[j] ? r : 63 - cnt[j]; // an item list holds 63 envs + its terminator
                r = r < quota ? r : quota;
                c[j] = r; lc += r;
            }
            const int incl = wave_incl_scan(lc);
            const int fit = __builtin_amdgcn_readlane(incl, 63);
            // exclusive prefix in circular lane order starting at lane `rot`
            const int before = (incl - lc) - __builtin_amdgcn_readlane(incl - lc, rot) + (ln < rot ? fit : 0);
            int take = left - before;
            take = take < 0 ? 0 : (take > lc ? lc : take);
            int at = pos0 + placed + before;
            RP_A(8);
            const unsigned long long got = __ballot(take > 0);
            if (got) { // next class / pass starts behind the last lane served
                const unsigned long long low = got & ((1ull << rot) - 1ull);
                rot = (64 - __builtin_clzll(low ? low : got)) & 63;
            }
#pragma unroll
            for (int j = 0; j < TBP; ++j) {
                const int t = c[j] < take ? c[j] : take;
                int32_t *dst = items + (size_t)(ln * TB + j) * 64 + cnt[j];
                // a tile rarely takes more than four envs of one class: those ids are fetched together, ahead of the stores (as a loop
                // every env paid an LDS round trip and a wave-wide vote: 13 of the builder's 31 us)
                int idq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) idq[q] = q < t ? (int)lds.ids[at + q] : 0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < t) dst[q] = (idq[q] + id_base) | (v << 16);
                if (__ballot(t > 4) != 0ull)
                    for (int q = 4; __ballot(q < t) != 0ull; ++q)
                        if (q < t) dst[q] = ((int)lds.ids[at + q] + id_base) | (v << 16);
                cnt[j] += t; load[j] += t * v; at += t; take -= t;
            }
            placed += fit < left ? fit : left;
            RP_A(10);
        }
        if (placed < m) bad = true;
    }
#pragma unroll
    for (int j = 0; j < TBP; ++j)
        if (own[j]) {
            tcnt[ln * TB + j] = cnt[j];
            items[(size_t)(ln * TB + j) * 64 + cnt[j]] = 0; // terminator: a consumer needs the item list only (rows == 0 ends it); cnt <= 63
            if (load[j] > 63) bad = true;
        }
    return __ballot(bad) == 0ull;
}

// Wavefront g of G (all 64 lanes active; G = rp_groups(E), every wavefront its own workgroup / Lds).  det: detected_human_num of the
// observation the plan is for.
// The wavefronts meet once, at the end: each adds itself to the arrival counter (a failed group adds a flag), the last one
// publishes the header (magic last) and clears the counter for the next build.
// arrive: the groups' arrival counter -- ONE int32 that is zero before the first build and that only this function touches (the last group
// through resets it).  The simulator passes library-owned memory (cn_env_batch), so that a plan buffer of arbitrary content (a caller's
// hipMalloc, 0xFF-filled) works: header word 7 is no longer read.  NULL (tools/row_plan_probe): header word 7 of a zeroed buffer, as before.
__device__ __forceinline__ void build(int g, int G, int E, int H, int NW, const float *__restrict__ det, int32_t *__restrict__ plan, Lds &lds, long long *tim = nullptr,
                                      int32_t *arrive = nullptr)
{
    const int ln = threadIdx.x & 63;
    int32_t *hdr = plan, *row_off = plan + rp_off_rowoff(), *tcnt = plan + rp_off_tcnt(E), *items = plan + rp_off_items(E);
    RP_T(0);
    if (ln == 0) hdr[0] = 0;   // (every group: the magic only comes back once ALL of them are through)
    if (H > RP_HMAX || E > RP_EMAX || (E & 3)) return;
    // ---- rows per env of the WHOLE batch: lane l owns envs [CH * l, CH * l + CH) (CH = 4 * ceil(E / 256) <= 64): totals of every group ----
    const int CH = ((E + 255) >> 8) << 2;
    int mysum = 0;
    {
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
#pragma unroll
        for (int i = 0; i < 64; ++i) mysum += c[i];
        // row offsets of the lanes of THIS group (serial inside the lane, one scan across)
        const int incl0 = wave_incl_scan(mysum);
        const int LG = 64 / G;  // lanes of the count scan per group (G > 1 only when E % 256 == 0: CH = E / 64, a group = LG whole lanes)
        if (ln / LG == g || G == 1) {
            int run = incl0 - mysum;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int4 o;
                o.x = run; run += c[4 * q]; o.y = run; run += c[4 * q + 1]; o.z = run; run += c[4 * q + 2]; o.w = run; run += c[4 * q + 3];
                if (4 * q < CH && ln * CH + 4 * q < E) *reinterpret_cast<int4 *>(row_off + ln * CH + 4 * q) = o; // row_off is 16-byte aligned (RP_HDR = 8)
            }
        }
        mysum = incl0; // from here on: the inclusive scan
    }
    const int total = __builtin_amdgcn_readlane(mysum, 63);
    if (g == 0 && ln == 0) row_off[E] = total;
    // ---- tiles of the batch, and the contiguous share of this group: proportional to its rows ----
    int n = (total + 62 * NW - 1) / (62 * NW);
    n = n < 1 ? 1 : n;
    const int Tall = n * NW;
    int e_base = 0, Eg = E, t0 = 0, T = Tall, grows = total;
    if (G > 1) {
        const int LG = 64 / G;
        const int cum0 = g ? __builtin_amdgcn_readlane(mysum, g * LG - 1) : 0, cum1 = __builtin_amdgcn_readlane(mysum, (g + 1) * LG - 1);
        grows = cum1 - cum0;
        t0 = (int)(((long long)cum0 * Tall + total / 2) / total);
        const int t1 = g == G - 1 ? Tall : (int)(((long long)cum1 * Tall + total / 2) / total);
        T = t1 - t0;
        Eg = E / G; e_base = g * Eg;
    }
    bool ok = T >= 1 && Tall <= RP_TMAX; // (too many tiles: no plan, but every group still goes to the meeting point below)
    int c[64];
    int hist = 0, cstart = 0; // lane v: number of envs of this group with v rows, first position of the class
    if (ok) {
        // ---- rows per env of THIS group: lane l owns local envs [CHg * l, CHg * l + CHg), CHg = 4 * ceil(Eg / 256) ----
        const int CHg = ((Eg + 255) >> 8) << 2;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float4 d = float4{0.f, 0.f, 0.f, 0.f};
            const int e0 = ln * CHg + 4 * q;
            const bool in = 4 * q < CHg && e0 < Eg;
            if (in) d = *reinterpret_cast<const float4 *>(det + e_base + e0);
            const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int r = (int)dd[u];
                r = r < 1 ? 1 : (r > H ? H : r);
                c[4 * q + u] = in ? r : 0;
            }
        }
        // ---- per-lane size histogram in LDS (column `lane` is private to the lane) ----
        for (int k = ln; k < (H + 1) * 64; k += 64) lds.tbl[k] = 0;
#pragma unroll
        for (int i = 0; i < 64; ++i)
            if (c[i]) __hip_atomic_fetch_add(&lds.tbl[c[i] * 64 + ln], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        RP_T(1);
        // ---- hand-out order: by rows descending; inside a size class by (lane, i).  tbl[v][lane] becomes the lane's next position ----
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
                if (c[i]) lds.ids[pos[i]] = (unsigned short)(ln * CHg + i);   // local index: the group's base is added at the hand-out
        }
        RP_T(2);
        const int TB = (T + 63) >> 6;
        int32_t *it = items + (size_t)t0 * 64, *tc = tcnt + t0;
        if (TB <= 1) ok = fill<1>(H, T, TB, grows, hist, cstart, e_base, it, tc, lds, tim);
        else if (TB <= 2) ok = fill<2>(H, T, TB, grows, hist, cstart, e_base, it, tc, lds, tim);
        else if (TB <= 4) ok = fill<4>(H, T, TB, grows, hist, cstart, e_base, it, tc, lds, tim);
        else if (TB <= 8) ok = fill<8>(H, T, TB, grows, hist, cstart, e_base, it, tc, lds, tim);
        else ok = fill<16>(H, T, TB, grows, hist, cstart, e_base, it, tc, lds, tim);
    }
    RP_T(3);
    __threadfence();
    if (ln == 0) {
        // the groups meet here: count in the low half of word 7, a failure flag in the high half; the last one through decides
        int32_t *ctr = arrive ? arrive : &hdr[7];
        const int old = G > 1 ? atomicAdd(ctr, ok ? 1 : 0x10001) : (ok ? 0 : 0x10000);
        if ((old & 0xffff) == G - 1) {
            const bool all_ok = ok && (old >> 16) == 0;
            if (G > 1) *ctr = 0;
            if (all_ok) {
                hdr[1] = NW; hdr[2] = n; hdr[3] = total; hdr[4] = E; hdr[5] = H; hdr[6] = Tall;
                __threadfence();
                hdr[0] = RP_MAGIC;
            }
        }
    }
    RP_T(4);
}

} // namespace rowplan
#endif
