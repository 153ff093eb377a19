#!/usr/bin/env python3
"""Offline study (numpy, no GPU): can the row plan (csrc/row_plan.h) be built by G independent wavefronts?

The builder is one wavefront and ~50 us, as long as the ORCA agents it rides along with ("what comes next").  Its
water-filling probes are per size class whatever the batch size, but everything else scales with the envs per lane.  Split the batch into
G index ranges; group g gets a CONTIGUOUS range of tiles whose length is proportional to its rows (largest remainders), and is packed on
its own by the same two-level scheme -- lanes as bins, the level of a lane weighted by the number of tiles it owns (a group's tile count is
not a multiple of 64; a lane owns a contiguous block of TB = ceil(T_g / 64) tiles, so the last lanes own fewer or none -- dealing the
tiles round-robin instead, 1 or 2 per lane, makes the level of the single-tile lanes too coarse: 17 % of the steps then have a
49-row tile at G = 8).  No data crosses groups: every wavefront reads all detected-human counts (it needs the other groups' totals for
its row offsets and its tile range) and packs only its own envs.

    python tools/row_plan_groups_study.py [counts.npy | counts.npz]    # default: tools/det_counts_sample.npz

The sample holds the detected-human counts of 30 rollout steps of the bench configuration (4096 envs x 20 humans, every 10th step of a
300-step window behind the de-phasing steps, policy at its initial weights).  A tile may hold 48 rows for its workgroup to stay at six
16-row blocks; printed per G: steps in which some tile exceeds that, and the histogram of the fullest tile.
"""
import os
import sys

import numpy as np

T_ALL, LANES = 512, 64


def pack(d, T, H=20):
    """row totals of the T tiles of one group (None when the envs do not fit)"""
    TB = -(-T // LANES)
    ntl = np.array([max(0, min(TB, T - l * TB)) for l in range(LANES)])
    lcap, lt = ntl * 63, np.zeros(LANES, np.int64)
    tiles = [np.zeros(n, np.int64) for n in ntl]
    for v in range(H, 0, -1):
        m = int((d == v).sum())
        if m == 0:
            continue

        def take(L):      # L: level of a lane with TB tiles
            room = np.minimum(L * ntl // TB, lcap) - lt
            return np.where(room > 0, room // v, 0)
        lo, hi = 0, int(lt.max()) * TB + TB * v * (m // max(1, int((ntl > 0).sum())) + 2) + 64 * TB
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if take(mid).sum() <= m:
                lo = mid
            else:
                hi = mid - 1
        k = take(lo)
        rem = m - int(k.sum())
        if rem > 0:
            idx = np.nonzero(take(lo + 1) > k)[0][:rem]
            if len(idx) < rem:
                return None
            k[idx] += 1
        for l in range(LANES):
            for _ in range(int(k[l])):
                tiles[l][int(np.argmin(tiles[l]))] += v
        lt += k * v
    return np.concatenate(tiles)


def grouped(d, G):
    per = len(d) // G
    rows = np.array([d[g * per:(g + 1) * per].sum() for g in range(G)])
    share = T_ALL * rows / rows.sum()
    Tg = np.floor(share).astype(int)
    Tg[np.argsort(-(share - Tg))[:T_ALL - Tg.sum()]] += 1
    out = []
    for g in range(G):
        r = pack(d[g * per:(g + 1) * per], int(Tg[g]))
        if r is None:
            return None
        out.append(r)
    return np.concatenate(out)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "det_counts_sample.npz")
    z = np.load(path)
    det = np.clip((z["det"] if hasattr(z, "files") else z).astype(np.int64), 1, 20)
    for G in (1, 2, 4, 8):
        mx, fail = [], 0
        for d in det:
            L = grouped(d, G)
            if L is None:
                fail += 1
            else:
                mx.append(int(L.max()))
        mx = np.array(mx)
        print("G=%d: %d steps, %d without a plan, %d with a tile above 48 rows; fullest tile: %s" % (
            G, len(det), fail, int((mx > 48).sum()), {int(v): int(c) for v, c in zip(*np.unique(mx, return_counts=True))}))


if __name__ == "__main__":
    main()
