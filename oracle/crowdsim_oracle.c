/*
 * crowdsim_oracle.c -- see crowdsim_oracle.h.  TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline).
 * Scalar, one env at a time, written for clarity: each function cites the reference lines it restates.
 */
#include "crowdsim_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * numpy legacy RandomState == MT19937 (np.random.seed(int) -> init_genrand; random_sample -> 53-bit double)
 * reference use: crowd_sim_var_num.py:338, :98, :122-126; agent.py:22,49-50
 * ---------------------------------------------------------------------------------------------- */
void orc_mt_seed(OrcMT *mt, uint32_t seed)
{
    for (int pos = 0; pos < 624; ++pos) {
        mt->key[pos] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
    }
    mt->pos = 624;
}

static void mt_twist(OrcMT *mt)
{
    uint32_t *k = mt->key;
    uint32_t y;
    int i;
    for (i = 0; i < 624 - 397; ++i) {
        y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
        k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; i < 623; ++i) {
        y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
        k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (k[623] & 0x80000000u) | (k[0] & 0x7fffffffu);
    k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    mt->pos = 0;
}

uint32_t orc_mt_next(OrcMT *mt)
{
    if (mt->pos == 624) mt_twist(mt);
    uint32_t y = mt->key[mt->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

double orc_mt_double(OrcMT *mt)
{
    uint32_t a = orc_mt_next(mt) >> 5, b = orc_mt_next(mt) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

static double env_random(OrcEnv *e)
{
    e->rng_draws += 2;
    return orc_mt_double(&e->rng);
}
static double env_uniform(OrcEnv *e, double lo, double hi) { return lo + (hi - lo) * env_random(e); }
/* np.random.normal(loc, scale) of the legacy RandomState: loc + scale * legacy_gauss (polar Box-Muller; the second deviate of a pair
 * is cached and returned by the next call; np.random.seed clears the cache) */
double orc_mt_normal(OrcMT *mt, int32_t *has_gauss, double *gauss, double loc, double scale, uint64_t *words)
{
    double g;
    if (*has_gauss) { g = *gauss; *has_gauss = 0; *gauss = 0.0; }
    else {
        double x1, x2, r2;
        do {
            x1 = 2.0 * orc_mt_double(mt) - 1.0;
            x2 = 2.0 * orc_mt_double(mt) - 1.0;
            if (words) *words += 4;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        const double f = sqrt(-2.0 * orc_log(r2) / r2);
        *gauss = f * x1; *has_gauss = 1;
        g = f * x2;
    }
    return loc + scale * g;
}
static double env_normal(OrcEnv *e, double loc, double scale) { return orc_mt_normal(&e->rng, &e->has_gauss, &e->gauss, loc, scale, &e->rng_draws); }

/* legacy RandomState.randint(low, high) for the default int64 dtype: numpy/random/_bounded_integers (_rand_int64 ->
 * random_bounded_uint64_fill with use_masked = 1): no draw when the range is a single value, else 32-bit words masked to the
 * next power of two minus one and rejected while above the range (ranges here are far below 2^32). */
int64_t orc_mt_randint(OrcMT *mt, int64_t low, int64_t high, uint64_t *words)
{
    const uint64_t rng = (uint64_t)(high - 1 - low);
    if (rng == 0) return low;
    uint64_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    uint32_t v;
    do { v = orc_mt_next(mt) & (uint32_t)mask; if (words) *words += 1; } while (v > rng);
    return low + (int64_t)v;
}
static int env_randint(OrcEnv *e, int low, int high) { return (int)orc_mt_randint(&e->rng, low, high, &e->rng_draws); }

/* ------------------------------------------------------------------------------------------------
 * deterministic sin/cos for x in [0, 2*pi]: Cody-Waite reduction by pi/2 + the classic minimax
 * kernels.  Plain +,-,* only (no FMA) so the HIP twin is bit-identical.  <= ~1 ulp.
 * Stands in for np.cos/np.sin at crowd_sim_var_num.py:127-128 and crowd_sim.py:427-428.
 * ---------------------------------------------------------------------------------------------- */
static double k_sin(double x)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x, w = z * z;
    double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    double v = z * x;
    return x + v * (S1 + z * r);
}
static double k_cos(double x)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x, w = z * z;
    double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * r);
}
void orc_sincos(double x, double *s, double *c)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double PIO2_1T = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
    int k = (int)(x * INV_PIO2 + 0.5);
    double fk = (double)k;
    double r = (x - fk * PIO2_1) - fk * PIO2_1T;
    double sr = k_sin(r), cr = k_cos(r);
    switch (k & 3) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
    }
}

/* deterministic exp: Cody-Waite reduction by ln 2 + the classic degree-5 minimax kernel, plain +,-,*,/ only (no FMA) so the HIP
 * twin is bit-identical.  < 1 ulp.  Stands in for np.exp at crowd_nav/policy/social_force.py:37 (arguments there are
 * (r_i + r_j - d) / B, a few units at most). */
double orc_exp(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10, INV_LN2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x > 700.0) x = 700.0;
    if (x < -700.0) return 0.0;
    const int k = (int)(INV_LN2 * x + (x < 0.0 ? -0.5 : 0.5));
    const double fk = (double)k;
    const double hi = x - fk * LN2_HI, lo = fk * LN2_LO;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return ldexp(y, k);
}

/* deterministic natural logarithm for normal positive arguments: the classic reduction x = 2^k (1 + f), sqrt(2)/2 < 1 + f < sqrt(2),
 * log(1 + f) = 2 s + s R(s^2), s = f / (2 + f), with the degree-14 minimax R and the usual hi/lo split of k ln 2; plain +,-,*,/
 * (no FMA) so the HIP twin is bit-identical.  < 1 ulp.  Stands in for log() inside RandomState.normal's polar method
 * (numpy/random/src/legacy/legacy-distributions.c legacy_gauss), arguments in (0, 1). */
double orc_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int32_t hx = (int32_t)(bits >> 32);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i0 = (hx + 0x95f64) & 0x100000;
    bits = ((uint64_t)(uint32_t)(hx | (i0 ^ 0x3ff00000)) << 32) | (bits & 0xffffffffull); /* normalise x or x / 2 */
    memcpy(&x, &bits, 8);
    k += i0 >> 20;
    const double f = x - 1.0;
    const double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2^-20 */
        if (f == 0.0) return k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        const double R0 = f * f * (0.5 - 0.33333333333333333 * f);
        return k == 0 ? f - R0 : dk * ln2_hi - ((R0 - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const int32_t i = (hx - 0x6147a) | (0x6b851 - hx);
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* ------------------------------------------------------------------------------------------------
 * ORCA -- restatement of RVO2 Library v2.0.2 (Agent.cpp), fp32, RVO_EPSILON = 1e-5.
 * Reference call site: crowd_nav/policy/orca.py:80-114 (only agent 0's velocity is read, :114).
 * ---------------------------------------------------------------------------------------------- */
#define RVO_EPSILON 0.00001f
typedef struct { float x, y; } V2;
typedef struct { V2 point, direction; } Line;

static inline V2 v2(float x, float y) { V2 r = {x, y}; return r; }
static inline V2 vadd(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
static inline V2 vsub(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
static inline V2 vscale(float s, V2 a) { return v2(s * a.x, s * a.y); }
static inline float vdot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
static inline float vdet(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
static inline float vabssq(V2 a) { return vdot(a, a); }
static inline float vabs(V2 a) { return sqrtf(vdot(a, a)); }
static inline V2 vdiv(V2 a, float s) { const float inv = 1.0f / s; return v2(a.x * inv, a.y * inv); } /* Vector2::operator/ */
static inline V2 vnormalize(V2 a) { return vdiv(a, vabs(a)); }
static inline float sqrf(float a) { return a * a; }

static int lp1(const Line *lines, int lineNo, float radius, V2 optVelocity, int directionOpt, V2 *result)
{
    const float dotProduct = vdot(lines[lineNo].point, lines[lineNo].direction);
    const float discriminant = sqrf(dotProduct) + sqrf(radius) - vabssq(lines[lineNo].point);
    if (discriminant < 0.0f) return 0;
    const float sqrtDiscriminant = sqrtf(discriminant);
    float tLeft = -dotProduct - sqrtDiscriminant;
    float tRight = -dotProduct + sqrtDiscriminant;
    for (int i = 0; i < lineNo; ++i) {
        const float denominator = vdet(lines[lineNo].direction, lines[i].direction);
        const float numerator = vdet(lines[i].direction, vsub(lines[lineNo].point, lines[i].point));
        if (fabsf(denominator) <= RVO_EPSILON) {
            if (numerator < 0.0f) return 0;
            continue;
        }
        const float t = numerator / denominator;
        if (denominator >= 0.0f) tRight = fminf(tRight, t);
        else tLeft = fmaxf(tLeft, t);
        if (tLeft > tRight) return 0;
    }
    if (directionOpt) {
        if (vdot(optVelocity, lines[lineNo].direction) > 0.0f)
            *result = vadd(lines[lineNo].point, vscale(tRight, lines[lineNo].direction));
        else
            *result = vadd(lines[lineNo].point, vscale(tLeft, lines[lineNo].direction));
    } else {
        const float t = vdot(lines[lineNo].direction, vsub(optVelocity, lines[lineNo].point));
        if (t < tLeft) *result = vadd(lines[lineNo].point, vscale(tLeft, lines[lineNo].direction));
        else if (t > tRight) *result = vadd(lines[lineNo].point, vscale(tRight, lines[lineNo].direction));
        else *result = vadd(lines[lineNo].point, vscale(t, lines[lineNo].direction));
    }
    return 1;
}

static int lp2(const Line *lines, int n, float radius, V2 optVelocity, int directionOpt, V2 *result)
{
    if (directionOpt) *result = vscale(radius, optVelocity);
    else if (vabssq(optVelocity) > sqrf(radius)) *result = vscale(radius, vnormalize(optVelocity));
    else *result = optVelocity;
    for (int i = 0; i < n; ++i) {
        if (vdet(lines[i].direction, vsub(lines[i].point, *result)) > 0.0f) {
            const V2 tempResult = *result;
            if (!lp1(lines, i, radius, optVelocity, directionOpt, result)) {
                *result = tempResult;
                return i;
            }
        }
    }
    return n;
}

static void lp3(const Line *lines, int n, int beginLine, float radius, V2 *result)
{
    float distance = 0.0f;
    Line proj[ORC_MAX_HUMANS];
    for (int i = beginLine; i < n; ++i) {
        if (vdet(lines[i].direction, vsub(lines[i].point, *result)) > distance) {
            int np = 0;
            for (int j = 0; j < i; ++j) {
                Line line;
                const float determinant = vdet(lines[i].direction, lines[j].direction);
                if (fabsf(determinant) <= RVO_EPSILON) {
                    if (vdot(lines[i].direction, lines[j].direction) > 0.0f) continue;
                    line.point = vscale(0.5f, vadd(lines[i].point, lines[j].point));
                } else {
                    line.point = vadd(lines[i].point,
                                      vscale(vdet(lines[j].direction, vsub(lines[i].point, lines[j].point)) / determinant,
                                             lines[i].direction));
                }
                line.direction = vnormalize(vsub(lines[j].direction, lines[i].direction));
                proj[np++] = line;
            }
            const V2 tempResult = *result;
            if (lp2(proj, np, radius, v2(-lines[i].direction.y, lines[i].direction.x), 1, result) < np)
                *result = tempResult;
            distance = vdet(lines[i].direction, vsub(lines[i].point, *result));
        }
    }
}

int orc_orca_velocity(float self_px, float self_py, float self_vx, float self_vy, float self_radius,
                      float max_speed, float pref_vx, float pref_vy, float neighbor_dist,
                      int max_neighbors, float time_horizon, float time_step, int n_other,
                      const float *opx, const float *opy, const float *ovx, const float *ovy,
                      const float *oradius, float *out_vx, float *out_vy, float *lines_out, int *line_fail_out)
{
    /* Agent::computeNeighbors + insertAgentNeighbor: others with distSq < neighborDist^2, ascending distSq,
     * at most maxNeighbors.  Ties keep index order (RVO2's kd-tree visit order is the only other candidate;
     * exact fp32 ties between distinct agents do not occur in generated scenes). */
    int idx[ORC_MAX_HUMANS];
    float dsq[ORC_MAX_HUMANS];
    int nn = 0;
    float rangeSq = sqrf(neighbor_dist);
    const V2 pos = v2(self_px, self_py), vel = v2(self_vx, self_vy);
    if (max_neighbors > 0) {
        for (int j = 0; j < n_other; ++j) {
            const float d = vabssq(vsub(pos, v2(opx[j], opy[j])));
            if (d < rangeSq) {
                if (nn < max_neighbors) { idx[nn] = j; dsq[nn] = d; ++nn; }
                int i = nn - 1;
                /* when the list was already full the last element is overwritten (RVO2 behaviour) */
                while (i != 0 && d < dsq[i - 1]) { dsq[i] = dsq[i - 1]; idx[i] = idx[i - 1]; --i; }
                dsq[i] = d; idx[i] = j;
                if (nn == max_neighbors) rangeSq = dsq[nn - 1];
            }
        }
    }
    Line lines[ORC_MAX_HUMANS];
    const float invTimeHorizon = 1.0f / time_horizon;
    for (int k = 0; k < nn; ++k) {
        const int j = idx[k];
        const V2 relativePosition = vsub(v2(opx[j], opy[j]), pos);
        const V2 relativeVelocity = vsub(vel, v2(ovx[j], ovy[j]));
        const float distSq = vabssq(relativePosition);
        const float combinedRadius = self_radius + oradius[j];
        const float combinedRadiusSq = sqrf(combinedRadius);
        Line line;
        V2 u;
        if (distSq > combinedRadiusSq) {
            const V2 w = vsub(relativeVelocity, vscale(invTimeHorizon, relativePosition));
            const float wLengthSq = vabssq(w);
            const float dotProduct1 = vdot(w, relativePosition);
            if (dotProduct1 < 0.0f && sqrf(dotProduct1) > combinedRadiusSq * wLengthSq) {
                const float wLength = sqrtf(wLengthSq);
                const V2 unitW = vdiv(w, wLength);
                line.direction = v2(unitW.y, -unitW.x);
                u = vscale(combinedRadius * invTimeHorizon - wLength, unitW);
            } else {
                const float leg = sqrtf(distSq - combinedRadiusSq);
                if (vdet(relativePosition, w) > 0.0f) {
                    line.direction = vdiv(v2(relativePosition.x * leg - relativePosition.y * combinedRadius,
                                             relativePosition.x * combinedRadius + relativePosition.y * leg), distSq);
                } else {
                    const V2 t = vdiv(v2(relativePosition.x * leg + relativePosition.y * combinedRadius,
                                         -relativePosition.x * combinedRadius + relativePosition.y * leg), distSq);
                    line.direction = v2(-t.x, -t.y);
                }
                const float dotProduct2 = vdot(relativeVelocity, line.direction);
                u = vsub(vscale(dotProduct2, line.direction), relativeVelocity);
            }
        } else {
            const float invTimeStep = 1.0f / time_step;
            const V2 w = vsub(relativeVelocity, vscale(invTimeStep, relativePosition));
            const float wLength = vabs(w);
            const V2 unitW = vdiv(w, wLength);
            line.direction = v2(unitW.y, -unitW.x);
            u = vscale(combinedRadius * invTimeStep - wLength, unitW);
        }
        line.point = vadd(vel, vscale(0.5f, u));
        lines[k] = line;
        if (lines_out) {
            lines_out[4 * k + 0] = line.point.x; lines_out[4 * k + 1] = line.point.y;
            lines_out[4 * k + 2] = line.direction.x; lines_out[4 * k + 3] = line.direction.y;
        }
    }
    V2 result;
    const int lineFail = lp2(lines, nn, max_speed, v2(pref_vx, pref_vy), 0, &result);
    if (lineFail < nn) lp3(lines, nn, lineFail, max_speed, &result);
    if (line_fail_out) *line_fail_out = lineFail;
    *out_vx = result.x;
    *out_vy = result.y;
    return nn;
}

/* ------------------------------------------------------------------------------------------------
 * Environment
 * ---------------------------------------------------------------------------------------------- */
void orc_config_default(OrcConfig *c)
{
    /* crowd_nav/configs/config.py:16-120 with the non-randomised training preset of
     * trained_models/GST_predictor_non_rand/configs/config.py (BASELINE configs[1]) */
    memset(c, 0, sizeof(*c));
    c->human_num = 20; c->predict_steps = 5; c->env_kind = ORC_ENV_VARNUM;
    c->randomize_attributes = 0; c->random_goal_changing = 0; c->end_goal_changing = 1;
    c->sort_humans = 1; c->phase = ORC_PHASE_TRAIN; c->nenv = 1;
    c->val_size = 100; c->test_size = 500;
    c->time_step = 0.25; c->time_limit = 50.0;
    c->success_reward = 10.0; c->collision_penalty = -20.0; c->discomfort_dist = 0.25;
    c->discomfort_penalty_factor = 10.0;
    c->circle_radius = 6.0 * sqrt(2.0); c->arena_size = 6.0;
    c->human_radius = 0.3; c->human_v_pref = 1.0; c->robot_radius = 0.3; c->robot_v_pref = 1.0;
    c->sensor_range = 5.0; c->goal_change_chance = 0.5; c->end_goal_change_chance = 1.0;
    c->sf_A = 2.0; c->sf_B = 1.0; c->sf_KI = 1.0;
    c->orca_neighbor_dist = 10.0; c->orca_safety_space = 0.15; c->orca_time_horizon = 5.0;
    c->orca_time_horizon_obst = 5.0;
    c->robot_fov = 2.0; c->human_fov = 2.0;
    c->pred_interval = 1;
}

int orc_obs_width(const OrcConfig *cfg)
{
    if (cfg->env_kind == ORC_ENV_COLLECT) return 4; /* pred_info: frame id, prediction id, px, py (crowd_sim_var_num_collect.py:36) */
    return cfg->env_kind == ORC_ENV_VARNUM ? 2 : 2 * (cfg->predict_steps + 1);
}

void orc_env_init(OrcEnv *e, const OrcConfig *cfg, int64_t this_seed)
{
    memset(e, 0, sizeof(*e));
    e->cfg = *cfg;
    e->this_seed = this_seed;
    e->shared_neighbor_dist = cfg->orca_neighbor_dist;
    orc_mt_seed(&e->rng, 0);
}

static double norm2(double x, double y) { return sqrt(x * x + y * y); }

/* crowd_sim_var_num.py:116-146 generate_circle_crossing_human; `slot` is the index the new human will occupy,
 * `n_existing` how many entries of e->humans are currently in self.humans (on respawn that includes the
 * human being replaced: the list still holds it while the right-hand side is evaluated, :455). */
static void gen_circle_crossing_human(OrcEnv *e, int slot, int n_existing)
{
    const OrcConfig *c = &e->cfg;
    OrcHuman h;
    h.radius = c->human_radius;
    h.v_pref = c->human_v_pref;
    if (c->randomize_attributes) {
        e->shared_neighbor_dist = env_uniform(e, 5.0, 10.0); /* Agent.__init__, agent.py:21-22 */
        h.v_pref = env_uniform(e, 0.5, 1.5);                  /* agent.py:49 */
        h.radius = env_uniform(e, 0.3, 0.5);                  /* agent.py:50 */
    }
    double px, py;
    for (int attempt = 0;; ++attempt) { /* unbounded in the reference; see ORC_MAX_PLACEMENT_ATTEMPTS */
        const double angle = env_random(e) * M_PI * 2.0;
        const double px_noise = env_uniform(e, 0.0, 1.0) * 2.0;
        const double py_noise = env_uniform(e, 0.0, 1.0) * 2.0;
        double s, co;
        orc_sincos(angle, &s, &co);
        px = c->circle_radius * co + px_noise;
        py = c->circle_radius * s + py_noise;
        int collide = 0;
        /* [self.robot] + self.humans */
        {
            /* :133-136: a unicycle robot keeps new humans half a circle radius away from its start and goal */
            const double min_dist = c->kinematics == ORC_KIN_UNICYCLE ? c->circle_radius / 2.0 : h.radius + c->robot_radius + c->discomfort_dist;
            if (norm2(px - e->rpx, py - e->rpy) < min_dist || norm2(px - e->rgx, py - e->rgy) < min_dist) collide = 1;
        }
        for (int j = 0; j < n_existing && !collide; ++j) {
            const OrcHuman *a = &e->humans[j];
            const double min_dist = h.radius + a->radius + c->discomfort_dist;
            if (norm2(px - a->px, py - a->py) < min_dist || norm2(px - a->gx, py - a->gy) < min_dist) collide = 1;
        }
        if (!collide || attempt >= (c->max_placement_attempts > 0 ? c->max_placement_attempts : ORC_MAX_PLACEMENT_ATTEMPTS)) break;
    }
    h.px = px; h.py = py; h.gx = -px; h.gy = -py; h.vx = 0.0; h.vy = 0.0;
    e->humans[slot] = h;
    e->sim_valid[slot] = 0; /* new Human -> new ORCA policy object with sim None */
}

/* The field-of-view half of detect_visible (crowd_sim.py:513-537): agent 2 is inside agent 1's cone of `fov` * pi radians around
 * agent 1's heading -- the direction of its velocity when the ROBOT is holonomic (np.arctan2(vy, vx); at rest that is +x, or -x for
 * vx = -0.0), its theta otherwise (the robot's heading; 0 for every human, human.set(..., theta = 0)).
 * Restated in a decision-equivalent form: arccos(clip(v_fov . v_12)) <= fov / 2  <=>  clip(v_fov . v_12) >= cos(fov / 2), and
 * (cos, sin)(arctan2(vy, vx)) = v / |v|; the reference's own evaluation goes through BLAS dot / nrm2, whose last-bit rounding is
 * build dependent, so its offset angle is not reproducible to the bit either -- only the decision is.  Coincident agents give
 * 0 / 0 = NaN and are not visible, as in the reference (np.abs(nan) <= x is False). */
static int in_fov(const OrcConfig *c, double fov, double px1, double py1, double vx1, double vy1, double theta1, double px2, double py2)
{
    double fx, fy;
    if (c->kinematics == ORC_KIN_UNICYCLE) orc_sincos(theta1, &fy, &fx);
    else if (vx1 == 0.0 && vy1 == 0.0) { fx = signbit(vx1) ? -1.0 : 1.0; fy = 0.0; }
    else { const double nv = sqrt(vx1 * vx1 + vy1 * vy1); fx = vx1 / nv; fy = vy1 / nv; }
    const double dx = px2 - px1, dy = py2 - py1;
    const double n12 = sqrt(dx * dx + dy * dy);
    double d = fx * (dx / n12) + fy * (dy / n12);
    d = d < -1.0 ? -1.0 : (d > 1.0 ? 1.0 : d); /* np.clip keeps NaN */
    const double half = M_PI * fov / 2.0;
    double thr = -1.0;
    if (half < M_PI) { double sn; orc_sincos(half, &sn, &thr); }
    return d >= thr;
}

/* test hook: the cone test alone */
int orc_in_fov(int unicycle, double fov, double px1, double py1, double vx1, double vy1, double theta1, double px2, double py2)
{
    OrcConfig c;
    memset(&c, 0, sizeof(c));
    c.kinematics = unicycle ? ORC_KIN_UNICYCLE : ORC_KIN_HOLONOMIC;
    return in_fov(&c, fov, px1, py1, vx1, vy1, theta1, px2, py2);
}

/* crowd_sim.py:513-552 detect_visible(robot, human, robot1=True).  With robot FOV = 2*pi the arccos test is always true unless the two
 * agents coincide (0/0 -> NaN -> False). */
static int robot_sees(const OrcEnv *e, const OrcHuman *h)
{
    const double dx = e->rpx - h->px, dy = e->rpy - h->py;
    if (dx == 0.0 && dy == 0.0) return 0;
    if (e->cfg.robot_fov < 2.0 && !in_fov(&e->cfg, e->cfg.robot_fov, e->rpx, e->rpy, e->rvx, e->rvy, e->rtheta, h->px, h->py)) return 0;
    const double dist = norm2(dx, dy) - e->cfg.robot_radius - h->radius;
    return dist <= e->cfg.sensor_range;
}

static void truth_future_traj(OrcEnv *e);

static void write_obs(OrcEnv *e, OrcObs *obs, int reset)
{
    const OrcConfig *c = &e->cfg;
    /* H humans exist right now; the observation always has HM = human_num + human_num_range rows (:249, crowd_sim_pred.py:78) */
    const int H = e->n_humans, HM = c->human_num + c->human_num_range, D = orc_obs_width(c), P = c->predict_steps;
    /* get_num_human_in_fov, crowd_sim.py:558-572 */
    int num_visible = 0;
    for (int i = 0; i < HM; ++i) e->human_visibility[i] = 0;
    for (int i = 0; i < H; ++i) {
        e->human_visibility[i] = robot_sees(e, &e->humans[i]);
        num_visible += e->human_visibility[i];
    }
    if (c->env_kind != ORC_ENV_PRED) { /* :275 -- CrowdSimPred's own generate_ob never refreshes the list (stays [] from reset) */
        e->observed_count = num_visible; e->observed_max = -1;
        for (int i = 0; i < H; ++i) if (e->human_visibility[i]) e->observed_max = i;
    }
    if (c->env_kind == ORC_ENV_COLLECT) {
        /* crowd_sim_var_num_collect.py:100-133: humans that were visible at the last observation and are not now get fresh prediction
         * ids (ascending, in list order); pred_info row i = (frame, id, ABSOLUTE position of the robot's belief) for visible humans,
         * (frame, id, inf, inf) for the others; the belief update is the usual one */
        for (int i = 0; i < H; ++i)
            if (e->last_observability[i] && !e->human_visibility[i]) e->human_pred_id[i] = e->max_human_id++;
        for (int i = 0; i < H; ++i) {
            double *s = e->last_human_states[i];
            if (e->human_visibility[i]) {
                const OrcHuman *h = &e->humans[i];
                s[0] = h->px; s[1] = h->py; s[2] = h->vx; s[3] = h->vy; s[4] = h->radius;
            } else if (reset) {
                s[0] = 15.0; s[1] = 15.0; s[2] = 0.0; s[3] = 0.0; s[4] = 0.3;
            } else {
                s[0] = s[0] + s[2] * c->time_step;
                s[1] = s[1] + s[3] * c->time_step;
            }
        }
        const double frame = ((double)e->step_counter * c->time_step) / c->time_step; /* global_time / data.pred_timestep (== env.time_step) */
        for (int i = 0; i < HM; ++i) {
            float *row = obs->spatial_edges + 4 * i;
            row[0] = (float)frame; row[1] = (float)e->human_pred_id[i];
            row[2] = (i < H && e->human_visibility[i]) ? (float)e->last_human_states[i][0] : INFINITY;
            row[3] = (i < H && e->human_visibility[i]) ? (float)e->last_human_states[i][1] : INFINITY;
        }
        for (int i = 0; i < HM; ++i) e->last_observability[i] = i < H ? e->human_visibility[i] : 0;
        obs->robot_node[0] = (float)e->rpx; obs->robot_node[1] = (float)e->rpy; obs->robot_node[2] = (float)c->robot_radius;
        obs->robot_node[3] = (float)e->rgx; obs->robot_node[4] = (float)e->rgy; obs->robot_node[5] = (float)c->robot_v_pref;
        obs->robot_node[6] = (float)e->rtheta;
        obs->temporal_edges[0] = (float)e->rvx; obs->temporal_edges[1] = (float)e->rvy;
        memset(obs->visible_masks, 0, sizeof(obs->visible_masks));
        for (int i = 0; i < H; ++i) obs->visible_masks[i] = (uint8_t)e->human_visibility[i];
        obs->detected_human_num = (float)(num_visible == 0 ? 1 : num_visible);
        return;
    }
    /* robot_node = get_full_state_list_noV (agent.py:105): px, py, r, gx, gy, v_pref, theta */
    obs->robot_node[0] = (float)e->rpx; obs->robot_node[1] = (float)e->rpy; obs->robot_node[2] = (float)c->robot_radius;
    obs->robot_node[3] = (float)e->rgx; obs->robot_node[4] = (float)e->rgy; obs->robot_node[5] = (float)c->robot_v_pref;
    obs->robot_node[6] = (float)e->rtheta;
    double prev_vel[ORC_MAX_HUMANS][2];
    for (int i = 0; i < H; ++i) { prev_vel[i][0] = e->last_human_states[i][2]; prev_vel[i][1] = e->last_human_states[i][3]; }
    /* update_last_human_states, crowd_sim.py:243-273 */
    for (int i = 0; i < H; ++i) {
        double *s = e->last_human_states[i];
        if (e->human_visibility[i]) {
            const OrcHuman *h = &e->humans[i];
            s[0] = h->px; s[1] = h->py; s[2] = h->vx; s[3] = h->vy; s[4] = h->radius;
        } else if (reset) {
            s[0] = 15.0; s[1] = 15.0; s[2] = 0.0; s[3] = 0.0; s[4] = 0.3;
        } else {
            s[0] = s[0] + s[2] * c->time_step;
            s[1] = s[1] + s[3] * c->time_step;
        }
    }
    obs->temporal_edges[0] = (float)e->rvx; obs->temporal_edges[1] = (float)e->rvy;

    double edges[ORC_MAX_HUMANS][2 * (ORC_MAX_PRED + 1)];
    for (int i = 0; i < HM; ++i)
        for (int d = 0; d < D; ++d) edges[i][d] = INFINITY;
    if (c->env_kind == ORC_ENV_VARNUM) {
        /* crowd_sim_var_num.py:249-256 */
        for (int i = 0; i < H; ++i)
            if (e->human_visibility[i]) {
                edges[i][0] = e->last_human_states[i][0] - e->rpx;
                edges[i][1] = e->last_human_states[i][1] - e->rpy;
            }
    } else {
        /* calc_human_future_traj('const_vel'), crowd_sim_var_num.py:166-226 + crowd_sim_pred.py:80-86.
         * PredRealGST (crowd_sim_pred_real_gst.py:78-93) fills the future slots with the current
         * relative position (np.tile), the wrapper overwrites them later. */
        /* predict_method 'truth' (crowd_sim_pred.py:81 with config.sim.predict_method): the humans' own ORCA rolled forward */
        if (c->predict_truth) truth_future_traj(e);
        for (int k = 0; k <= P && !c->predict_truth; ++k)
            for (int i = 0; i < H; ++i) {
                if (e->human_visibility[i]) {
                    const double t = (double)k * c->time_step * (double)(c->pred_interval > 1 ? c->pred_interval : 1); /* :212 */
                    e->future_traj[k][i][0] = e->humans[i].px + t * prev_vel[i][0];
                    e->future_traj[k][i][1] = e->humans[i].py + t * prev_vel[i][1];
                } else {
                    e->future_traj[k][i][0] = 15.0; e->future_traj[k][i][1] = 15.0;
                }
            }
        for (int i = 0; i < H; ++i)
            if (e->human_visibility[i])
                for (int k = 0; k <= P; ++k) {
                    if (c->env_kind == ORC_ENV_PRED) {
                        edges[i][2 * k] = e->future_traj[k][i][0] - e->rpx;
                        edges[i][2 * k + 1] = e->future_traj[k][i][1] - e->rpy;
                    } else {
                        edges[i][2 * k] = e->last_human_states[i][0] - e->rpx;
                        edges[i][2 * k + 1] = e->last_human_states[i][1] - e->rpy;
                    }
                }
    }
    /* sorted(..., key=norm of first two) is stable; all-inf rows keep index order and end last */
    int order[ORC_MAX_HUMANS];
    for (int i = 0; i < HM; ++i) order[i] = i;
    /* PredRealGST never sorts in the env (crowd_sim_pred_real_gst.py:80: sort=False; the wrapper sorts later) */
    const int do_sort = c->sort_humans && c->env_kind != ORC_ENV_PRED_GST;
    if (do_sort) {
        double key[ORC_MAX_HUMANS];
        for (int i = 0; i < HM; ++i) key[i] = sqrt(edges[i][0] * edges[i][0] + edges[i][1] * edges[i][1]);
        for (int i = 1; i < HM; ++i) { /* stable insertion sort */
            int oi = order[i]; double ki = key[oi]; int j = i - 1;
            while (j >= 0 && key[order[j]] > ki) { order[j + 1] = order[j]; --j; }
            order[j + 1] = oi;
        }
    }
    for (int r = 0; r < HM; ++r)
        for (int d = 0; d < D; ++d) {
            double v = edges[order[r]][d];
            if (isinf(v)) v = 15.0;
            obs->spatial_edges[r * D + d] = (float)v;
        }
    memset(obs->visible_masks, 0, sizeof(obs->visible_masks));
    if (do_sort) {
        for (int i = 0; i < num_visible; ++i) obs->visible_masks[i] = 1;
    } else {
        for (int i = 0; i < H; ++i) obs->visible_masks[i] = (uint8_t)e->human_visibility[i];
    }
    obs->detected_human_num = (float)(num_visible == 0 ? 1 : num_visible);
}

void orc_env_reset(OrcEnv *e, OrcObs *obs)
{
    const OrcConfig *c = &e->cfg;
    /* crowd_sim_var_num.py:333-338; configure(): case_capacity val/test = 1000 */
    const uint64_t offset[3] = {2000u, 0u, 1000u};
    const int ph = c->phase;
    const uint64_t seed = offset[ph] + e->case_counter[ph] + (uint64_t)e->this_seed;
    orc_mt_seed(&e->rng, (uint32_t)seed);
    e->rng_draws = 0;
    e->has_gauss = 0; e->gauss = 0.0; /* _legacy_seeding */
    e->step_counter = 0;
    double px, py, gx, gy;
    e->observed_count = 0; e->observed_max = -1; /* :327 */
    if (c->kinematics == ORC_KIN_UNICYCLE) {
        /* generate_robot_humans, sim2real branch :78-91: start on the arena circle, goal >= 4 m away, random heading,
         * 1 .. human_num + human_num_range humans */
        const double angle = env_uniform(e, 0.0, M_PI * 2.0);
        double s, co;
        orc_sincos(angle, &s, &co);
        px = c->arena_size * co; py = c->arena_size * s;
        for (;;) {
            gx = env_uniform(e, -c->arena_size, c->arena_size);
            gy = env_uniform(e, -c->arena_size, c->arena_size);
            if (norm2(px - gx, py - gy) >= 4.0) break;
        }
        e->rtheta = env_uniform(e, 0.0, 2.0 * M_PI);
        e->n_humans = env_randint(e, 1, c->human_num + c->human_num_range + 1);
    } else {
        /* holonomic branch :97-104 */
        for (;;) {
            px = env_uniform(e, -c->arena_size, c->arena_size);
            py = env_uniform(e, -c->arena_size, c->arena_size);
            gx = env_uniform(e, -c->arena_size, c->arena_size);
            gy = env_uniform(e, -c->arena_size, c->arena_size);
            if (norm2(px - gx, py - gy) >= 8.0) break;
        }
        e->rtheta = M_PI / 2.0;
        /* randint(H - range, H + range + 1): consumes no draw when human_num_range == 0 */
        e->n_humans = env_randint(e, c->human_num - c->human_num_range, c->human_num + c->human_num_range + 1);
    }
    e->rpx = px; e->rpy = py; e->rgx = gx; e->rgy = gy; e->rvx = 0.0; e->rvy = 0.0;
    for (int i = 0; i < e->n_humans; ++i) gen_circle_crossing_human(e, i, i);
    memset(e->last_human_states, 0, sizeof(e->last_human_states)); /* :108 */
    const uint64_t case_size[3] = {4294967295ull - 2000ull, c->val_size, c->test_size};
    e->case_counter[ph] = (e->case_counter[ph] + (uint64_t)c->nenv) % case_size[ph];
    e->potential = -fabs(norm2(e->rgx - e->rpx, e->rgy - e->rpy));
    e->ep_return = 0.0; e->ep_len = 0;
    if (c->env_kind == ORC_ENV_COLLECT) { /* crowd_sim_var_num_collect.py:79-81 */
        for (int i = 0; i < ORC_MAX_HUMANS; ++i) { e->last_observability[i] = 0; e->human_pred_id[i] = i; }
        e->max_human_id = e->n_humans;
    }
    write_obs(e, obs, 1);
}

/* human i's private rvo2 simulator, orca.py:80-89: rebuilt when the agent count differs, parameters frozen at creation */
static void ensure_human_sim(OrcEnv *e, int i, int n_agents, const int *sees /* NULL: every other human as it is */)
{
    const OrcConfig *c = &e->cfg;
    const int H = e->n_humans;
    const OrcHuman *me = &e->humans[i];
    if (e->sim_valid[i] && e->sim_n[i] != n_agents) e->sim_valid[i] = 0; /* :80-82 crowd size changed -> new simulator */
    if (!e->sim_valid[i]) { /* :83-89 */
        e->sim_n[i] = n_agents;
        e->sim_nd[i] = (float)e->shared_neighbor_dist;
        e->sim_self_radius[i] = (float)(me->radius + 0.01 + c->orca_safety_space);
        e->sim_self_maxspeed[i] = (float)me->v_pref;
        /* addAgent takes the radius of the state it is handed: a human outside i's field of view at this moment is the dummy human with
         * the config radius (crowd_sim.py:688-693) -- and stays that size in this simulator */
        for (int j = 0; j < H; ++j)
            if (j != i) e->sim_seen_radius[i][j] = (float)((sees && !sees[j] ? c->human_radius : e->humans[j].radius) + 0.01 + c->orca_safety_space);
        e->sim_valid[i] = 1;
    }
}

/* detect_visible(human i, other agent) of get_human_actions (crowd_sim.py:686-699): no sensor range for humans, only the field of view */
static int human_sees(const OrcEnv *e, const OrcHuman *me, double ox, double oy)
{
    if (e->cfg.human_fov >= 2.0) return !(ox == me->px && oy == me->py);
    return in_fov(&e->cfg, e->cfg.human_fov, me->px, me->py, me->vx, me->vy, 0.0, ox, oy);
}

/* ORCA.predict for human i, orca.py:64-117 */
static void human_orca_action(OrcEnv *e, int i, float *avx, float *avy)
{
    const OrcConfig *c = &e->cfg;
    const int H = e->n_humans;
    const OrcHuman *me = &e->humans[i];
    int sees[ORC_MAX_HUMANS];
    for (int j = 0; j < H; ++j) sees[j] = j != i && human_sees(e, me, e->humans[j].px, e->humans[j].py);
    ensure_human_sim(e, i, H + (c->robot_visible ? 1 : 0), sees); /* self + the others (+ the robot) */
    float opx[ORC_MAX_HUMANS], opy[ORC_MAX_HUMANS], ovx[ORC_MAX_HUMANS], ovy[ORC_MAX_HUMANS], orad[ORC_MAX_HUMANS];
    int n = 0;
    for (int j = 0; j < H; ++j) {
        if (j == i) continue;
        const OrcHuman *o = &e->humans[j];
        /* get_human_actions, crowd_sim.py:686-693: human FOV = 2*pi -> visible unless coincident, else dummy (7,7,0,0) */
        if (!sees[j]) {
            opx[n] = 7.0f; opy[n] = 7.0f; ovx[n] = 0.0f; ovy[n] = 0.0f;
        } else {
            opx[n] = (float)o->px; opy[n] = (float)o->py; ovx[n] = (float)o->vx; ovy[n] = (float)o->vy;
        }
        orad[n] = e->sim_seen_radius[i][j];
        ++n;
    }
    if (c->robot_visible) {
        /* crowd_sim.py:695-699: the robot is appended as the last neighbour; FOV = 2*pi -> visible unless coincident, else the
         * dummy robot parked at (7,7).  Its rvo2 radius was fixed when human i's simulator was created. */
        if (!human_sees(e, me, e->rpx, e->rpy)) { opx[n] = 7.0f; opy[n] = 7.0f; ovx[n] = 0.0f; ovy[n] = 0.0f; }
        else { opx[n] = (float)e->rpx; opy[n] = (float)e->rpy; ovx[n] = (float)e->rvx; ovy[n] = (float)e->rvy; }
        orad[n] = (float)(c->robot_radius + 0.01 + c->orca_safety_space);
        ++n;
    }
    /* :97-100 pref velocity: goal vector, normalised only when longer than 1 */
    double vx = me->gx - me->px, vy = me->gy - me->py;
    const double speed = norm2(vx, vy);
    if (speed > 1.0) { vx = vx / speed; vy = vy / speed; }
    orc_orca_velocity((float)me->px, (float)me->py, (float)me->vx, (float)me->vy, e->sim_self_radius[i],
                      e->sim_self_maxspeed[i], (float)vx, (float)vy, e->sim_nd[i], n /* max_neighbors = len(human_states) */,
                      (float)c->orca_time_horizon, (float)c->time_step, n, opx, opy, ovx, ovy, orad, avx, avy, 0, 0);
}

/* SOCIAL_FORCE.predict for human i (crowd_nav/policy/social_force.py:11-52; humans.policy = 'social_force'), float64 throughout.
 * The other agents are the ones get_human_actions passes: every other human (true state unless coincident -> the dummy at (7,7)),
 * then the robot when robot.visible.  This path never touches rvo2: traces of it are the reference's own arithmetic end to end. */
static void human_sf_action(const OrcEnv *e, int i, double *avx, double *avy)
{
    const OrcConfig *c = &e->cfg;
    const int H = e->n_humans;
    const OrcHuman *me = &e->humans[i];
    const double dxg = me->gx - me->px, dyg = me->gy - me->py;
    const double dist_to_goal = sqrt(dxg * dxg + dyg * dyg);
    const double desired_vx = (dxg / dist_to_goal) * me->v_pref, desired_vy = (dyg / dist_to_goal) * me->v_pref;
    const double curr_dvx = c->sf_KI * (desired_vx - me->vx), curr_dvy = c->sf_KI * (desired_vy - me->vy);
    double ivx = 0.0, ivy = 0.0;
    for (int j = 0; j <= H; ++j) {
        double ox, oy, orad;
        if (j == i) continue;
        if (j < H) {
            const OrcHuman *o = &e->humans[j];
            if (!human_sees(e, me, o->px, o->py)) { ox = 7.0; oy = 7.0; orad = c->human_radius; } /* dummy_human: config radius */
            else { ox = o->px; oy = o->py; orad = o->radius; }
        } else {
            if (!c->robot_visible) continue;
            if (!human_sees(e, me, e->rpx, e->rpy)) { ox = 7.0; oy = 7.0; } else { ox = e->rpx; oy = e->rpy; }
            orad = c->robot_radius;
        }
        const double dx = me->px - ox, dy = me->py - oy;
        const double d = sqrt(dx * dx + dy * dy);
        const double f = c->sf_A * orc_exp((me->radius + orad - d) / c->sf_B);
        ivx += f * (dx / d);
        ivy += f * (dy / d);
    }
    const double nvx = me->vx + (curr_dvx + ivx) * c->time_step, nvy = me->vy + (curr_dvy + ivy) * c->time_step;
    const double act_norm = sqrt(nvx * nvx + nvy * nvy);
    if (act_norm > me->v_pref) { *avx = nvx / act_norm * me->v_pref; *avy = nvy / act_norm * me->v_pref; }
    else { *avx = nvx; *avy = nvy; }
}

/* calc_human_future_traj(method='truth'), crowd_sim_var_num.py:152-227 (test phase, :386-388): roll every human forward
 * predict_steps times with its own ORCA policy on the states predicted by the previous roll (act_joint_state ->
 * ORCA.predict on the human's private simulator, so the frozen radii / neighbour distance of human_orca_action apply;
 * the other humans' states are passed as they are, without the FOV / dummy substitution of get_human_actions), then
 * blank the humans the robot does not currently see (:222-224).  Only positions are kept (future_traj). */
static void truth_future_traj(OrcEnv *e)
{
    const OrcConfig *c = &e->cfg;
    const int H = e->n_humans, I = c->pred_interval > 1 ? c->pred_interval : 1, P = c->predict_steps * I; /* buffer_len rolls (crowd_sim.py:181) */
    double cur[ORC_MAX_HUMANS][4], nxt[ORC_MAX_HUMANS][4];
    for (int i = 0; i < H; ++i) {
        cur[i][0] = e->humans[i].px; cur[i][1] = e->humans[i].py; cur[i][2] = e->humans[i].vx; cur[i][3] = e->humans[i].vy;
        e->future_traj[0][i][0] = cur[i][0]; e->future_traj[0][i][1] = cur[i][1];
    }
    for (int k = 1; k <= P; ++k) {
        for (int i = 0; i < H; ++i) {
            const OrcHuman *me = &e->humans[i];
            if (c->humans_policy == ORC_HUMANS_SOCIAL_FORCE) {
                /* humans.policy = 'social_force': act_joint_state -> SOCIAL_FORCE.predict (social_force.py:11-52) on the rolled states;
                 * the others are the H - 1 fellow humans with their true radii (no dummy substitution, no robot: :183-190), float64 */
                const double dxg = me->gx - cur[i][0], dyg = me->gy - cur[i][1];
                const double dist_to_goal = sqrt(dxg * dxg + dyg * dyg);
                const double desired_vx = (dxg / dist_to_goal) * me->v_pref, desired_vy = (dyg / dist_to_goal) * me->v_pref;
                const double curr_dvx = c->sf_KI * (desired_vx - cur[i][2]), curr_dvy = c->sf_KI * (desired_vy - cur[i][3]);
                double ivx = 0.0, ivy = 0.0;
                for (int j = 0; j < H; ++j) {
                    if (j == i) continue;
                    const double dx = cur[i][0] - cur[j][0], dy = cur[i][1] - cur[j][1];
                    const double d = sqrt(dx * dx + dy * dy);
                    const double f = c->sf_A * orc_exp((me->radius + e->humans[j].radius - d) / c->sf_B);
                    ivx += f * (dx / d);
                    ivy += f * (dy / d);
                }
                const double nvx = cur[i][2] + (curr_dvx + ivx) * c->time_step, nvy = cur[i][3] + (curr_dvy + ivy) * c->time_step;
                const double act_norm = sqrt(nvx * nvx + nvy * nvy);
                double ax = nvx, ay = nvy;
                if (act_norm > me->v_pref) { ax = nvx / act_norm * me->v_pref; ay = nvy / act_norm * me->v_pref; }
                nxt[i][0] = cur[i][0] + ax * c->time_step; nxt[i][1] = cur[i][1] + ay * c->time_step;
                nxt[i][2] = ax; nxt[i][3] = ay;
                continue;
            }
            float opx[ORC_MAX_HUMANS], opy[ORC_MAX_HUMANS], ovx[ORC_MAX_HUMANS], ovy[ORC_MAX_HUMANS], orad[ORC_MAX_HUMANS];
            int n = 0;
            ensure_human_sim(e, i, H, NULL); /* only new at reset (predict_method 'truth'): act_joint_state builds it like ORCA.predict */
            for (int j = 0; j < H; ++j) {
                if (j == i) continue;
                opx[n] = (float)cur[j][0]; opy[n] = (float)cur[j][1]; ovx[n] = (float)cur[j][2]; ovy[n] = (float)cur[j][3];
                orad[n] = e->sim_seen_radius[i][j];
                ++n;
            }
            double vx = me->gx - cur[i][0], vy = me->gy - cur[i][1];
            const double speed = norm2(vx, vy);
            if (speed > 1.0) { vx = vx / speed; vy = vy / speed; }
            float ax, ay;
            orc_orca_velocity((float)cur[i][0], (float)cur[i][1], (float)cur[i][2], (float)cur[i][3], e->sim_self_radius[i],
                              e->sim_self_maxspeed[i], (float)vx, (float)vy, e->sim_nd[i], H - 1, (float)c->orca_time_horizon,
                              (float)c->time_step, n, opx, opy, ovx, ovy, orad, &ax, &ay, 0, 0);
            /* one_step_lookahead, agent.py:185-192 */
            nxt[i][0] = cur[i][0] + (double)ax * c->time_step; nxt[i][1] = cur[i][1] + (double)ay * c->time_step;
            nxt[i][2] = (double)ax; nxt[i][3] = (double)ay;
        }
        for (int i = 0; i < H; ++i) {
            for (int q = 0; q < 4; ++q) cur[i][q] = nxt[i][q];
            if (k % I == 0) { e->future_traj[k / I][i][0] = cur[i][0]; e->future_traj[k / I][i][1] = cur[i][1]; } /* [::pred_interval], :206 */
        }
    }
    for (int i = 0; i < H; ++i)
        if (!e->human_visibility[i])
            for (int k = 0; k <= c->predict_steps; ++k) { e->future_traj[k][i][0] = 15.0; e->future_traj[k][i][1] = 15.0; }
}

/* crowd_sim.py:415-450 (every human, goal_change_chance) and :453-485 (update_human_goal: one human, end_goal_change_chance;
 * `only` >= 0 selects it) */
static void update_human_goals_randomly(OrcEnv *e, int only)
{
    const OrcConfig *c = &e->cfg;
    const int H = e->n_humans;
    for (int i = 0; i < H; ++i) {
        OrcHuman *h = &e->humans[i];
        if (only >= 0 && i != only) continue;
        if (only < 0 && h->v_pref == 0.0) continue;
        if (env_random(e) <= (only >= 0 ? c->end_goal_change_chance : c->goal_change_chance)) {
            double gx, gy;
            for (int attempt = 0;; ++attempt) {
                const double angle = env_random(e) * M_PI * 2.0;
                const double v_pref = h->v_pref == 0.0 ? 1.0 : h->v_pref;
                const double gx_noise = (env_random(e) - 0.5) * v_pref;
                const double gy_noise = (env_random(e) - 0.5) * v_pref;
                double s, co;
                orc_sincos(angle, &s, &co);
                gx = c->circle_radius * co + gx_noise;
                gy = c->circle_radius * s + gy_noise;
                int collide = 0;
                {
                    const double md = h->radius + c->robot_radius + c->discomfort_dist;
                    if (norm2(gx - e->rpx, gy - e->rpy) < md || norm2(gx - e->rgx, gy - e->rgy) < md) collide = 1;
                }
                for (int j = 0; j < H && !collide; ++j) {
                    if (j == i) continue;
                    const OrcHuman *a = &e->humans[j];
                    const double md = h->radius + a->radius + c->discomfort_dist;
                    if (norm2(gx - a->px, gy - a->py) < md || norm2(gx - a->gx, gy - a->gy) < md) collide = 1;
                }
                if (!collide || attempt >= (c->max_placement_attempts > 0 ? c->max_placement_attempts : ORC_MAX_PLACEMENT_ATTEMPTS)) break;
            }
            h->gx = gx; h->gy = gy;
        }
    }
}

int orc_env_step(OrcEnv *e, const float action_in[2], OrcObs *obs, double *reward_out, int *info_out, double *danger_min_dist)
{
    const OrcConfig *c = &e->cfg;
    const int H = e->n_humans;
    /* srnn.clip_action (srnn.py:17-34): float32 arithmetic on the raw action */
    float ax = action_in[0], ay = action_in[1];
    double axd = 0.0, ayd = 0.0; /* float64 action of the social-force robot */
    double uni_v = 0.0, uni_r = 0.0; /* ActionRot(v, r) of the unicycle robot */
    if (c->robot_policy == ORC_ROBOT_SOCIAL_FORCE) {
        /* SOCIAL_FORCE.predict (crowd_nav/policy/social_force.py:11-52) on the robot's beliefs, all in float64 */
        const double dxg = e->rgx - e->rpx, dyg = e->rgy - e->rpy;
        const double dist_to_goal = sqrt(dxg * dxg + dyg * dyg);
        const double desired_vx = (dxg / dist_to_goal) * c->robot_v_pref, desired_vy = (dyg / dist_to_goal) * c->robot_v_pref;
        const double curr_dvx = c->sf_KI * (desired_vx - e->rvx), curr_dvy = c->sf_KI * (desired_vy - e->rvy);
        double ivx = 0.0, ivy = 0.0;
        for (int j = 0; j < H; ++j) {
            const double dx = e->rpx - e->last_human_states[j][0], dy = e->rpy - e->last_human_states[j][1];
            const double d = sqrt(dx * dx + dy * dy);
            const double f = c->sf_A * orc_exp((c->robot_radius + e->last_human_states[j][4] - d) / c->sf_B);
            ivx += f * (dx / d);
            ivy += f * (dy / d);
        }
        const double nvx = e->rvx + (curr_dvx + ivx) * c->time_step, nvy = e->rvy + (curr_dvy + ivy) * c->time_step;
        const double act_norm = sqrt(nvx * nvx + nvy * nvy);
        if (act_norm > c->robot_v_pref) { axd = nvx / act_norm * c->robot_v_pref; ayd = nvy / act_norm * c->robot_v_pref; }
        else { axd = nvx; ayd = nvy; }
    } else if (c->robot_policy == ORC_ROBOT_ORCA) {
        /* crowd_sim_var_num.py:371-375: robot.act(copy of last_human_states) -> ORCA.predict (orca.py:64-117) on the robot's
         * BELIEFS about all H humans (never-seen ones sit at the (15,15) dummy); no clip_action on this path */
        if (e->rob_sim_valid && e->rob_sim_n != H + 1) e->rob_sim_valid = 0; /* orca.py:80-82 */
        if (!e->rob_sim_valid) {
            e->rob_sim_n = H + 1;
            e->rob_sim_nd = (float)e->shared_neighbor_dist;
            e->rob_sim_self_radius = (float)(c->robot_radius + 0.01 + c->orca_safety_space);
            e->rob_sim_self_maxspeed = (float)c->robot_v_pref;
            for (int j = 0; j < H; ++j) e->rob_sim_seen_radius[j] = (float)(e->last_human_states[j][4] + 0.01 + c->orca_safety_space);
            e->rob_sim_valid = 1;
        }
        float opx[ORC_MAX_HUMANS], opy[ORC_MAX_HUMANS], ovx[ORC_MAX_HUMANS], ovy[ORC_MAX_HUMANS];
        for (int j = 0; j < H; ++j) {
            opx[j] = (float)e->last_human_states[j][0]; opy[j] = (float)e->last_human_states[j][1];
            ovx[j] = (float)e->last_human_states[j][2]; ovy[j] = (float)e->last_human_states[j][3];
        }
        double vx = e->rgx - e->rpx, vy = e->rgy - e->rpy;
        const double speed = norm2(vx, vy);
        if (speed > 1.0) { vx = vx / speed; vy = vy / speed; }
        orc_orca_velocity((float)e->rpx, (float)e->rpy, (float)e->rvx, (float)e->rvy, e->rob_sim_self_radius, e->rob_sim_self_maxspeed,
                          (float)vx, (float)vy, e->rob_sim_nd, H, (float)c->orca_time_horizon, (float)c->time_step, H, opx, opy, ovx, ovy,
                          e->rob_sim_seen_radius, &ax, &ay, 0, 0);
    } else if (c->kinematics == ORC_KIN_UNICYCLE) {
        /* srnn.py:36-43: (change of v, change of theta) clipped in float32; crowd_sim_var_num.py:379-381: the commanded speed is
         * the running sum self.desiredVelocity[0], clipped to +-v_pref.  With the numpy the reference pins (1.20.3) a float32
         * scalar combined with a Python float gives float64, so everything after the clip runs in float64. */
        const float dv = fminf(fmaxf(ax, (float)-0.1), (float)0.087);
        ay = fminf(fmaxf(ay, (float)-0.06), (float)0.06);
        e->desired_v = fmin(fmax(e->desired_v + (double)dv, -c->robot_v_pref), c->robot_v_pref);
        uni_v = e->desired_v; uni_r = (double)ay;
        if (c->env_kind != ORC_ENV_VARNUM) {
            /* CrowdSimPred.step (crowd_sim_pred.py:120-131; CrowdSimPredRealGST steps through it) sends the command through
             * smooth_action (crowd_sim.py:315-358): the wheel speeds of a Turtlebot2i (wheel radius 0.035 m, track 0.23 m), clipped to
             * +-17.5 rad/s, low-pass filtered in the test phase, then reduced towards zero by a noisy dead band N(1.8, 0.15) per wheel */
            const double w = uni_r / c->time_step;
            double left = (2.0 * uni_v - 0.23 * w) / (2.0 * 0.035), right = (2.0 * uni_v + 0.23 * w) / (2.0 * 0.035);
            left = fmin(fmax(left, -17.5), 17.5); right = fmin(fmax(right, -17.5), 17.5);
            if (c->phase == ORC_PHASE_TEST) {
                left = (1. - 0.1) * e->last_left + 0.1 * left;
                right = (1. - 0.1) * e->last_right + 0.1 * right;
            }
            e->last_left = left; e->last_right = right;
            if (left > 0) left = fmax(0., left - env_normal(e, 1.8, 0.15)); else left = fmin(0., left + env_normal(e, 1.8, 0.15));
            if (right > 0) right = fmax(0., right - env_normal(e, 1.8, 0.15)); else right = fmin(0., right + env_normal(e, 1.8, 0.15));
            uni_v = 0.035 / 2 * (left + right);
            uni_r = 0.035 / 0.23 * (right - left) * c->time_step;
        }
    } else {
        const float act_norm = sqrtf(ax * ax + ay * ay);
        const float vp = (float)c->robot_v_pref;
        if (act_norm > vp) { ax = ax / act_norm * vp; ay = ay / act_norm * vp; }
    }
    /* get_human_actions, crowd_sim.py:680-703 */
    float hax[ORC_MAX_HUMANS], hay[ORC_MAX_HUMANS];
    double haxd[ORC_MAX_HUMANS], hayd[ORC_MAX_HUMANS]; /* the actions as Python floats (ORCA: widened float32) */
    for (int i = 0; i < H; ++i) {
        if (c->humans_policy == ORC_HUMANS_SOCIAL_FORCE) {
            human_sf_action(e, i, &haxd[i], &hayd[i]);
            hax[i] = (float)haxd[i]; hay[i] = (float)hayd[i];
        } else {
            human_orca_action(e, i, &hax[i], &hay[i]);
            haxd[i] = (double)hax[i]; hayd[i] = (double)hay[i];
        }
        e->last_human_actions[i][0] = hax[i]; e->last_human_actions[i][1] = hay[i];
    }
    /* test phase (:386-388): the true future positions decide the Danger flag (and, for CrowdSimPred, the social reward) */
    if (c->phase == ORC_PHASE_TEST) truth_future_traj(e);
    /* calc_reward, crowd_sim_var_num.py:465-561 (train / val phase: danger zone = circle; test: future trajectories) */
    double dmin = INFINITY;
    int collision = 0;
    for (int i = 0; i < H; ++i) {
        const double dx = e->humans[i].px - e->rpx, dy = e->humans[i].py - e->rpy;
        const double closest = sqrt(dx * dx + dy * dy) - e->humans[i].radius - c->robot_radius;
        if (closest < 0.0) { collision = 1; break; }
        else if (closest < dmin) dmin = closest;
    }
    /* :487-492 */
    const int reaching_goal = norm2(e->rpx - e->rgx, e->rpy - e->rgy) < (c->kinematics == ORC_KIN_UNICYCLE ? 0.6 : c->robot_radius);
    const double global_time = (double)e->step_counter * c->time_step;
    double reward; int done, info; double mind = 0.0;
    int danger_cond = dmin < c->discomfort_dist; /* :496-498 */
    /* :499-511 intrusion into the humans' future positions (np.amin over the hits).  Test phase: the 'truth' roll-out just made.  Phase 'val'
     * (CrowdSimPred-v0 only -- the other two env classes never assign self.human_future_traj outside the test phase and fail at :501):
     * whatever the previous observation left there, i.e. its const_vel / truth predictions with the unseen humans blanked. */
    if (c->phase == ORC_PHASE_TEST || c->phase == ORC_PHASE_VAL) {
        danger_cond = 0;
        for (int k = 1; k <= c->predict_steps; ++k)
            for (int i = 0; i < H; ++i) {
                const double dx = e->future_traj[k][i][0] - e->rpx, dy = e->future_traj[k][i][1] - e->rpy;
                const double d = sqrt(dx * dx + dy * dy);
                if (d < c->robot_radius + c->human_radius) { if (!danger_cond || d < mind) mind = d; danger_cond = 1; }
            }
    }
    if (c->env_kind == ORC_ENV_COLLECT) {
        /* crowd_sim_var_num_collect.py:139-188: the data-collection env never ends an episode (global_time >= 40000 aside) and pays
         * no reward; a robot that reaches its goal gets a new one -- the median of the humans' positions or a uniform point of
         * the arena, each with probability 1/2 (np.random draws in this order: uniform(0, 1), then uniform(-a, a, size = 2)) */
        reward = 0.0; done = 0; info = ORC_INFO_NOTHING;
        if (global_time >= 40000.0) { done = 1; info = ORC_INFO_TIMEOUT; }
        else if (collision) info = ORC_INFO_COLLISION;
        else if (norm2(e->rpx - e->rgx, e->rpy - e->rgy) < c->robot_radius) {
            info = ORC_INFO_REACHGOAL;
            if (env_uniform(e, 0.0, 1.0) < 0.5) {
                double med[2];
                for (int d = 0; d < 2; ++d) { /* np.median(axis = 0): middle element, or the mean of the two middle ones */
                    double v[ORC_MAX_HUMANS];
                    for (int i = 0; i < H; ++i) v[i] = d == 0 ? e->humans[i].px : e->humans[i].py;
                    for (int i = 1; i < H; ++i) { const double x = v[i]; int j = i - 1; while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; } v[j + 1] = x; }
                    med[d] = (H & 1) ? v[H / 2] : (v[H / 2 - 1] + v[H / 2]) / 2.0;
                }
                e->rgx = med[0]; e->rgy = med[1];
            } else {
                e->rgx = env_uniform(e, -c->arena_size, c->arena_size);
                e->rgy = env_uniform(e, -c->arena_size, c->arena_size);
            }
        }
    }
    else if (global_time >= c->time_limit - 1.0) { reward = 0.0; done = 1; info = ORC_INFO_TIMEOUT; }
    else if (collision) { reward = c->collision_penalty; done = 1; info = ORC_INFO_COLLISION; }
    else if (reaching_goal) { reward = c->success_reward; done = 1; info = ORC_INFO_REACHGOAL; }
    else if (danger_cond) {
        reward = (dmin - c->discomfort_dist) * c->discomfort_penalty_factor * c->time_step;
        done = 0; info = ORC_INFO_DANGER;
    } else {
        const double potential_cur = norm2(e->rpx - e->rgx, e->rpy - e->rgy);
        reward = (c->kinematics == ORC_KIN_UNICYCLE ? 3.0 : 2.0) * (-fabs(potential_cur) - e->potential); /* :536-542 pot_factor */
        e->potential = -fabs(potential_cur);
        done = 0; info = ORC_INFO_NOTHING;
    }
    if (c->env_kind == ORC_ENV_PRED) {
        /* social reward, crowd_sim_pred.py:216-233; future_traj is the one stored by the previous generate_ob (in the
         * test phase it was just overwritten by the 'truth' roll-out above, as in the reference) */
        double rf = 0.0; /* np.min over products collision_idx * penalty (0 where no collision) */
        for (int k = 1; k <= c->predict_steps; ++k) {
            const double pen = c->collision_penalty / ldexp(1.0, k + 1);
            for (int i = 0; i < H; ++i) {
                const double dx = e->future_traj[k][i][0] - e->rpx, dy = e->future_traj[k][i][1] - e->rpy;
                if (sqrt(dx * dx + dy * dy) < c->robot_radius + c->human_radius) { if (pen < rf) rf = pen; }
            }
        }
        reward = reward + rf;
    }
    if (c->kinematics == ORC_KIN_UNICYCLE) {
        /* :548-559 rotation penalty and reversing penalty, added to every outcome */
        const double r_spin = -4.5 * (uni_r * uni_r);
        const double r_back = uni_v < 0.0 ? -2.0 * fabs(uni_v) : 0.0;
        reward = reward + r_spin + r_back;
    }
    /* apply actions, agent.py:143-183 */
    if (c->kinematics == ORC_KIN_UNICYCLE) {
        /* differential drive :148-165.  A rotation below 1e-4 sets R = 0, i.e. the robot does not translate on that step
         * (the reference's formula, restated as it is). */
        double R = 0.0;
        if (!(fabs(uni_r) < 0.0001)) { const double w = uni_r / c->time_step; R = uni_v / w; }
        double s0, c0, s1, c1;
        orc_sincos(e->rtheta, &s0, &c0);
        orc_sincos(e->rtheta + uni_r, &s1, &c1);
        e->rpx = e->rpx - R * s0 + R * s1;
        e->rpy = e->rpy + R * c0 - R * c1;
        double th = fmod(e->rtheta + uni_r, 2.0 * M_PI); /* Python / numpy %: the result takes the divisor's sign */
        if (th != 0.0 && th < 0.0) th += 2.0 * M_PI;
        e->rtheta = th;
        orc_sincos(th, &s0, &c0);
        e->rvx = uni_v * c0; e->rvy = uni_v * s0;
    } else if (c->robot_policy == ORC_ROBOT_SOCIAL_FORCE) {
        e->rpx = e->rpx + axd * c->time_step; e->rpy = e->rpy + ayd * c->time_step;
        e->rvx = axd; e->rvy = ayd;
    } else {
        e->rpx = e->rpx + (double)(ax * (float)c->time_step); /* float32 * python float stays float32 (NEP 50), exact for 0.25 */
        e->rpy = e->rpy + (double)(ay * (float)c->time_step);
        e->rvx = (double)ax; e->rvy = (double)ay;
    }
    for (int i = 0; i < H; ++i) {
        OrcHuman *h = &e->humans[i];
        h->px = h->px + haxd[i] * c->time_step;
        h->py = h->py + hayd[i] * c->time_step;
        h->vx = haxd[i]; h->vy = hayd[i];
    }
    e->step_counter += 1;
    const int every_5s = (e->step_counter % (int)llround(5.0 / c->time_step)) == 0; /* self.global_time % 5 == 0 */
    if (c->human_num_range > 0 && every_5s) {
        /* crowd_sim_var_num.py:404-437 / crowd_sim_pred.py:165-190 (CrowdSimPredRealGST steps through CrowdSimPred.step):
         * humans leave from the END of the list, and only ones the robot was not looking at; new ones are appended */
        const int HM = c->human_num + c->human_num_range;
        if (env_random(e) < 0.5) {
            int max_remove;
            if (c->env_kind == ORC_ENV_VARNUM) {
                max_remove = e->n_humans - (c->human_num - c->human_num_range);
                if (e->observed_count > 0 && (e->n_humans - 1) - e->observed_max < max_remove) max_remove = (e->n_humans - 1) - e->observed_max;
            } else {
                max_remove = e->observed_count == 0 ? e->n_humans - 1 : (e->n_humans - 1) - e->observed_max;
                if (c->human_num_range < max_remove) max_remove = c->human_num_range;
            }
            e->n_humans -= env_randint(e, 0, max_remove + 1);
        } else {
            const int add_num = env_randint(e, 0, c->human_num_range + 1);
            const int first = e->n_humans;
            for (int i = first; i < first + add_num && i < HM; ++i) {
                gen_circle_crossing_human(e, i, i);
                double *ls = e->last_human_states[i];
                ls[0] = 15.0; ls[1] = 15.0; ls[2] = 0.0; ls[3] = 0.0; ls[4] = 0.3;
                e->n_humans = i + 1;
            }
        }
    }
    write_obs(e, obs, 0);
    /* :446-448 random goal changing every 5 s of sim time */
    if (c->random_goal_changing && every_5s) update_human_goals_randomly(e, -1);
    /* :451-456 humans that reached their goal: respawned (holonomic robot) or given a new goal (unicycle robot) */
    if (c->end_goal_changing) {
        const int n = e->n_humans;
        for (int i = 0; i < n; ++i) {
            const OrcHuman *h = &e->humans[i];
            if (norm2(h->gx - h->px, h->gy - h->py) < h->radius) {
                /* crowd_sim_var_num.py:453-458 gives the humans of a unicycle robot a new goal; crowd_sim_pred.py:208-212 always respawns */
                if (c->kinematics == ORC_KIN_UNICYCLE && c->env_kind == ORC_ENV_VARNUM) update_human_goals_randomly(e, i);
                else gen_circle_crossing_human(e, i, n);
            }
        }
    }
    e->ep_return += reward; e->ep_len += 1;
    *reward_out = reward; *info_out = info;
    if (danger_min_dist) *danger_min_dist = info == ORC_INFO_DANGER ? mind : 0.0; /* Danger(min_dist) is the only info that carries it */
    return done;
}

int orc_env_step_autoreset(OrcEnv *e, const float action[2], OrcObs *obs, double *reward, int *info,
                           double *ep_return, int *ep_len, double *danger_min_dist)
{
    const int done = orc_env_step(e, action, obs, reward, info, danger_min_dist);
    if (done) {
        /* bench.Monitor: info['episode'] = {'r': round(sum, 6), 'l': steps}; shmem_vec_env.py:139-142 auto-reset */
        if (ep_return) *ep_return = e->ep_return;
        if (ep_len) *ep_len = e->ep_len;
        orc_env_reset(e, obs);
    }
    return done;
}

OrcEnv *orc_env_new(const OrcConfig *cfg, int64_t this_seed)
{
    OrcEnv *e = (OrcEnv *)malloc(sizeof(OrcEnv));
    if (e) orc_env_init(e, cfg, this_seed);
    return e;
}
void orc_env_free(OrcEnv *e) { free(e); }
/* crowd_sim_var_num.py:316-318: `case_counter[phase] = test_case` */
void orc_env_set_case_counter(OrcEnv *e, uint64_t value) { e->case_counter[e->cfg.phase] = value; }
int orc_env_human_count(const OrcEnv *e) { return e->n_humans; }
int orc_sizeof_env(void) { return (int)sizeof(OrcEnv); }
int orc_sizeof_obs(void) { return (int)sizeof(OrcObs); }
int orc_sizeof_config(void) { return (int)sizeof(OrcConfig); }

void orc_env_batch_step(OrcEnv **envs, int n, const float *actions, float *robot_node, float *temporal_edges,
                        float *spatial_edges, float *detected, uint8_t *visible, float *rewards, uint8_t *dones,
                        uint8_t *infos)
{
    for (int i = 0; i < n; ++i) {
        OrcObs obs;
        double r; int info;
        const int H = envs[i]->cfg.human_num + envs[i]->cfg.human_num_range, D = orc_obs_width(&envs[i]->cfg);
        const int done = orc_env_step_autoreset(envs[i], actions + 2 * i, &obs, &r, &info, 0, 0, 0);
        memcpy(robot_node + 7 * i, obs.robot_node, 7 * sizeof(float));
        memcpy(temporal_edges + 2 * i, obs.temporal_edges, 2 * sizeof(float));
        memcpy(spatial_edges + (size_t)i * H * D, obs.spatial_edges, (size_t)H * D * sizeof(float));
        detected[i] = obs.detected_human_num;
        memcpy(visible + (size_t)i * H, obs.visible_masks, (size_t)H);
        rewards[i] = (float)r; dones[i] = (uint8_t)done; infos[i] = (uint8_t)info;
    }
}

/* storage.py:123-132 (use_gae=True, use_proper_time_limits=False), torch fp32 op order:
 * delta = r + gamma * V[t+1] * m[t+1] - V[t];  gae = delta + gamma * lam * m[t+1] * gae;  R = gae + V[t] */
void orc_gae(int T, int N, const float *rewards, const float *values, const float *masks, double gamma_d, double lam_d, float *returns)
{
    /* python scalars: gamma -> float32 when multiplied into a tensor; gamma*gae_lambda is a python (double) product first */
    const float gamma = (float)gamma_d, gl = (float)(gamma_d * lam_d);
    for (int n = 0; n < N; ++n) {
        float gae = 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            const float delta = rewards[t * N + n] + gamma * values[(t + 1) * N + n] * masks[(t + 1) * N + n] - values[t * N + n];
            gae = delta + gl * masks[(t + 1) * N + n] * gae;
            returns[t * N + n] = gae + values[t * N + n];
        }
    }
}
