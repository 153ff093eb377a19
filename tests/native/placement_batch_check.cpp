// host emulation of place_by_rejection's control flow against the one-at-a-time loop on the same MT19937 stream
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
static const int MT_N = 624;
struct Rng { uint32_t k[MT_N]; int pos; };
static uint32_t mix(uint32_t a, uint32_t b, uint32_t c) { uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu); return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
static void twist(Rng &R) { for (int i = 0; i < 227; ++i) R.k[i] = mix(R.k[i], R.k[i + 1], R.k[i + 397]); for (int i = 227; i < 623; ++i) R.k[i] = mix(R.k[i], R.k[i + 1], R.k[i - 227]); R.k[623] = mix(R.k[623], R.k[0], R.k[396]); R.pos = 0; }
static uint32_t temper(uint32_t y) { y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18); return y; }
static uint32_t u32(Rng &R) { if (R.pos == MT_N) twist(R); return temper(R.k[R.pos++]); }
static double dbl(Rng &R) { uint32_t a = u32(R) >> 5, b = u32(R) >> 6; return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0; }
static void seed(Rng &R, uint32_t s) { for (int i = 0; i < MT_N; ++i) { R.k[i] = s; s = 1812433253u * (s ^ (s >> 30)) + i + 1; } R.pos = MT_N; }
static double thr;
static bool collides(double x, double y) { return fmod(fabs(x * 7.3 + y * 3.1), 1.0) < thr; }
static void make(double u0, double u1, double u2, double &x, double &y) { x = cos(u0 * 6.28) * 6 + u1 * 2; y = sin(u0 * 6.28) * 6 + u2 * 2; }
static void serial(Rng &R, int max_att, double &ox, double &oy) {
    for (int attempt = 0;; ++attempt) { double u0 = dbl(R), u1 = dbl(R), u2 = dbl(R); double x, y; make(u0, u1, u2, x, y); if (!collides(x, y) || attempt >= max_att) { ox = x; oy = y; return; } }
}
static void batch(Rng &R, int max_att, double &ox, double &oy) {
    // the device loop of place_by_rejection (csrc/env_sim.hip): passes of up to 64 candidates inside the current block; a candidate whose
    // six words straddle the end of the block is lane 0 of the first pass over the regenerated block
    int attempt = 0;
    for (;;) {
        const int left = MT_N - R.pos;
        int nb, first, nt = 0;
        uint32_t tl[5] = {0, 0, 0, 0, 0};
        if (left >= 6) { nb = left / 6 < 64 ? left / 6 : 64; first = R.pos; }
        else { nt = left; for (int k = 0; k < nt; ++k) tl[k] = R.k[R.pos + k]; twist(R); nb = 64; first = -1; }
        const int need = 6 - nt;
        int f = -1; double fx = 0, fy = 0;
        for (int lane = 0; lane < 64; ++lane) {
            const bool live = lane < nb;
            uint32_t wd[6];
            if (first >= 0) { const uint32_t *w = R.k + first + 6 * (live ? lane : 0); for (int k = 0; k < 6; ++k) wd[k] = w[k]; }
            else for (int k = 0; k < 6; ++k) { const int idx = lane == 0 ? (k < nt ? 0 : k - nt) : need + 6 * (lane - 1) + k; const uint32_t v = R.k[idx]; wd[k] = (lane == 0 && k < nt) ? tl[k < 5 ? k : 4] : v; }
            const uint32_t a0 = temper(wd[0]) >> 5, b0 = temper(wd[1]) >> 6, a1 = temper(wd[2]) >> 5, b1 = temper(wd[3]) >> 6, a2 = temper(wd[4]) >> 5, b2 = temper(wd[5]) >> 6;
            const double u0 = ((double)a0 * 67108864.0 + (double)b0) / 9007199254740992.0, u1 = ((double)a1 * 67108864.0 + (double)b1) / 9007199254740992.0, u2 = ((double)a2 * 67108864.0 + (double)b2) / 9007199254740992.0;
            double x, y; make(u0, u1, u2, x, y);
            if (live && (!collides(x, y) || attempt + lane >= max_att) && f < 0) { f = lane; fx = x; fy = y; }
        }
        if (f >= 0) { ox = fx; oy = fy; R.pos = first >= 0 ? first + 6 * (f + 1) : need + 6 * f; return; }
        R.pos = first >= 0 ? first + 6 * nb : need + 6 * (nb - 1);
        attempt += nb;
    }
}
int main() {
    long bad = 0, n = 0;
    for (int s = 0; s < 400; ++s) {
        Rng A, B; seed(A, 1000 + s); B = A;
        thr = (s % 5 == 0) ? 1.1 : (s % 5 == 1 ? 0.999 : (s % 5 == 2 ? 0.98 : (s % 5 == 3 ? 0.7 : 0.2)));
        const int caps[4] = {0, 5, 64, 1000};
        for (int it = 0; it < 300; ++it) {
            if (it % 7 == 3) { (void)u32(A); (void)u32(B); }       // odd offsets
            if (it % 11 == 5) { (void)dbl(A); (void)dbl(B); }
            const int cap = caps[(s + it) % 4];
            double ax, ay, bx, by; serial(A, cap, ax, ay); batch(B, cap, bx, by); ++n;
            if (ax != bx || ay != by || A.pos != B.pos || memcmp(A.k, B.k, sizeof A.k)) { if (bad < 5) printf("MISMATCH s=%d it=%d cap=%d pos %d %d\n", s, it, cap, A.pos, B.pos); ++bad; }
        }
    }
    printf("%ld placements, %ld mismatches\n", n, bad);
    return bad != 0;
}
