"""Speculative human placement for the episode generator -- CPU study of the algorithm (no GPU, no product code).

generate_circle_crossing_human (crowd_sim_var_num.py:116-146) places the humans of an episode one after another; every attempt draws
three doubles from the legacy MT19937 stream and is rejected when it is too close to the robot's start / goal or to an earlier human's
position / goal.  The device generator (env_sim.hip: gen_human) walks that chain attempt by attempt: ~26 attempts per episode at 20
humans, each a dependent chain of tempering, fp64 sin / cos and distance tests.

Speculation: the stream position of human i's FIRST attempt is known if no earlier attempt is rejected.  One pass computes the first
attempts of all unresolved humans at those positions (lane = human), tests every candidate against the robot and against all
candidates / accepted humans before it, and accepts everything in front of the first rejection f; the stream position after f's failed
attempt is known, the next pass starts there.  Passes = 1 + rejections; the arithmetic per candidate and per test is the sequential one.

This script restates both forms on the raw MT19937 word stream and checks that they produce identical episodes (positions, attributes,
stream position) over many seeds; it prints the attempts / passes statistics.  Run: python tools/spec_placement_study.py
"""
import math
import sys

import numpy as np

CIRCLE_R, ARENA, DD, H_RADIUS, R_RADIUS = 6.0 * math.sqrt(2.0), 6.0, 0.25, 0.3, 0.3


def words(seed, n=8192):
    rs = np.random.RandomState(seed)
    w = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32).astype(np.uint64)
    chk = np.random.RandomState(seed).random_sample(4)          # the word view is the stream random_sample() consumes
    for k in range(4):
        a, b = int(w[2 * k]) >> 5, int(w[2 * k + 1]) >> 6
        assert (a * 67108864.0 + b) / 9007199254740992.0 == chk[k]
    return w


def dbl(w, pos):
    a, b = int(w[pos]) >> 5, int(w[pos + 1]) >> 6
    return (a * 67108864.0 + b) / 9007199254740992.0


def uni(w, pos, lo, hi):
    return lo + (hi - lo) * dbl(w, pos)


def robot(w):
    pos = 0
    while True:
        px, py, gx, gy = (uni(w, pos + 2 * k, -ARENA, ARENA) for k in range(4))
        pos += 8
        if math.sqrt((px - gx) ** 2 + (py - gy) ** 2) >= 8.0:
            return pos, (px, py, gx, gy)


def candidate(w, pos):
    angle = dbl(w, pos) * math.pi * 2.0
    nx = uni(w, pos + 2, 0.0, 1.0) * 2.0
    ny = uni(w, pos + 4, 0.0, 1.0) * 2.0
    return CIRCLE_R * math.cos(angle) + nx, CIRCLE_R * math.sin(angle) + ny


def near(x, y, d):
    return math.sqrt(x * x + y * y) < d


def hits(px, py, rad, rb, others):
    md_r = rad + R_RADIUS + DD
    if near(px - rb[0], py - rb[1], md_r) or near(px - rb[2], py - rb[3], md_r):
        return True
    for (ox, oy, orad) in others:
        md = rad + orad + DD
        if near(px - ox, py - oy, md) or near(px + ox, py + oy, md):          # the goal of a human is the mirror image of its start
            return True
    return False


def sequential(w, n, rand):
    pos, rb = robot(w)
    hs, attempts = [], 0
    for _ in range(n):
        rad, vp, nd = H_RADIUS, 1.0, None
        if rand:
            nd, vp, rad = uni(w, pos, 5.0, 10.0), uni(w, pos + 2, 0.5, 1.5), uni(w, pos + 4, 0.3, 0.5)
            pos += 6
        while True:
            px, py = candidate(w, pos)
            pos += 6
            attempts += 1
            if not hits(px, py, rad, rb, [(h[0], h[1], h[2]) for h in hs]):
                break
        hs.append((px, py, rad, vp, nd))
    return hs, pos, attempts


def speculative(w, n, rand):
    pos, rb = robot(w)
    A = 6 if rand else 0
    done = []                        # accepted humans
    i0, attrs0 = 0, None             # first unresolved human; its attributes once drawn (a rejected first attempt keeps them)
    passes = 0
    while i0 < n:
        passes += 1
        cand, p = [], pos
        for i in range(i0, n):       # "lane i": everything below is independent across i given the assumed positions
            if i == i0 and attrs0 is not None:
                rad, vp, nd = attrs0
            elif rand:
                nd, vp, rad = uni(w, p, 5.0, 10.0), uni(w, p + 2, 0.5, 1.5), uni(w, p + 4, 0.3, 0.5)
                p += A
            else:
                rad, vp, nd = H_RADIUS, 1.0, None
            px, py = candidate(w, p)
            p += 6
            cand.append((px, py, rad, vp, nd, p))          # p = stream position after this attempt
        base = [(h[0], h[1], h[2]) for h in done]
        bad = [hits(c[0], c[1], c[2], rb, base + [(d[0], d[1], d[2]) for d in cand[:k]]) for k, c in enumerate(cand)]
        f = bad.index(True) if True in bad else len(cand)
        done += [c[:5] for c in cand[:f]]
        if f < len(cand):
            pos = cand[f][5]
            attrs0 = (cand[f][2], cand[f][3], cand[f][4])
        else:
            pos = cand[-1][5]
        i0 += f
    return done, pos, passes


def main():
    for n, rand in ((20, False), (20, True), (5, True), (50, True)):
        att, pas = [], []
        for seed in range(2000, 2000 + (300 if n <= 20 else 60)):
            w = words(seed, 8192 if n <= 20 else 65536)
            a, pa, na = sequential(w, n, rand)
            b, pb, np_ = speculative(w, n, rand)
            assert a == b and pa == pb, (n, rand, seed)
            att.append(na); pas.append(np_)
        print("H = %2d randomised = %-5s attempts %.1f (max %d)   passes %.1f (max %d)   identical episodes: %d"
              % (n, rand, np.mean(att), max(att), np.mean(pas), max(pas), len(att)))


if __name__ == "__main__":
    sys.exit(main())
