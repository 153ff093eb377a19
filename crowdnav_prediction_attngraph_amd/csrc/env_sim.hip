// env_sim.hip -- batched CrowdNav++ simulator on gfx950: ORCA humans, reward / termination, observation assembly,
// numpy-compatible scenario generation and auto-reset, all device resident.
//
// Compile with -ffp-contract=off: ORCA follows RVO2's fp32 operation order (no FMA), positions are fp64 like the
// Python reference, so results are reproducible bit for bit against the scalar CPU restatement used by the tests.
//
// Mapping (one wavefront = 64 lanes everywhere in this file; a lane is a human):
//   orca_lane_kernel one LANE per (env, human i) for crowds of <= 32 agents: neighbour keys ordered by a sorting network, ORCA lines
//                    and linearProgram2 in per-lane register vectors; the agents
//                    whose program is infeasible (about a third in a dense crossing) hand their lines to
//   orca_lp3_kernel  one wavefront per such agent: lane k = line k, RVO2's linearProgram3 wave-cooperatively (the outer loop over
//                    lines is its serial dependence; the inner clip of a line against all earlier lines is one lane-parallel
//                    min/max reduction)
//   orca_kernel      the whole solve one wavefront per agent (crowds of more than 32 agents; also the 'truth' roll-outs)
//   env_step_kernel  one wavefront per env: robot clip + reward/collision (lane-parallel distances, ballot/any),
//                    kinematics, visibility, belief update, distance rank sort + observation scatter, goal changes,
//                    respawns and the in-launch auto-reset.  The MT19937 stream of the env lives in HBM ([E][624]) and
//                    is staged into LDS only by the (rare) wavefronts that draw from it; the 624-word twist is
//                    lane-parallel.
// Reference semantics (file:line under the reference repo) are cited at each block.
#include "common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

constexpr int MT_N = 624;
constexpr float RVO_EPS = 0.00001f;

struct Lp3Hdr { int32_t agent, nn, line_fail; float rx, ry, radius; };

struct EnvDev {
    cn_env_config cfg;
    int E, H, D, P;
    int I, R;          // pred_interval (crowd_sim.py:180) and the 'truth' roll count R = P * I (buffer_len, :181); slice k * I of `tr` is prediction k
    int64_t seed_base; // thisSeed of env 0 of this batch
    // humans [E][8][H] double: px,py,vx,vy,gx,gy,radius,v_pref
    double *hum;
    // robot [E][8] double: px,py,vx,vy,gx,gy,theta,potential
    double *rob;
    double *lhs;   // last_human_states [E][5][H]
    double *ftraj; // [E][P][2][H] predicted positions k=1..P (const_vel), only for CN_ENV_PRED
    int32_t *step_counter; // [E]
    uint64_t *case_counter; // [E]
    double *ep_ret;         // [E] running episode return
    int32_t *ep_cnt;        // [E] running episode length
    double *shared_nd;      // [E] config.orca.neighbor_dist
    uint8_t *sim_valid;     // [E][H]
    float *sim_nd, *sim_self_radius, *sim_self_maxspeed; // [E][H]
    float *sim_seen; // [E][H][H] or nullptr (non-randomised: radii never change)
    uint32_t *mt;    // [E][624]
    int32_t *mt_pos; // [E]
    float *hact;     // [E][2][H] ORCA velocities of this step
    // next-episode staging: episode k+1 of env e is a pure function of (seed, e, k), so it is generated ahead of time on
    // the side stream (env_pregen_kernel) and a finishing env only copies it in (the serial MT19937 seeding + rejection
    // sampling of 20 humans would otherwise be the tail of env_step_kernel)
    double *nx_hum;     // [E][8][H]
    double *nx_rob;     // [E][8]
    double *nx_shared_nd; // [E]
    uint32_t *nx_mt;    // [E][624]
    int32_t *nx_mt_pos; // [E]
    int32_t *post_cnt, *post_list; // [1], [E] envs whose post-observation updates (goal changes, respawns) this step deferred to env_post_kernel
    int32_t *plan_arrive; // [1] row-plan builders' arrival counter (library-owned: the caller's plan buffer may hold anything)
    int coop_after;       // candidates a placement loop evaluates on one wavefront before the env's helper wavefronts join (env_step_kernel<false, 4>)
    uint8_t *nx_ready;  // [E]
    int32_t *nx_prog;   // [E] pre-generation in progress: 0 = not started, k + 1 = seed, robot and the first k humans are staged
    uint64_t *nx_case;  // [E] the case counter that staging was started for (a reset in between makes it stale)
    // test phase only (crowd_sim_var_num.py:386-388, :499-511): the humans' true future states rolled out with their own
    // ORCA policies, the robot's visibility flags of the last observation, and Danger's min_dist of the last step
    // robot.policy == 'orca': the robot's own rvo2 simulator, created at its first use and kept across episodes (orca.py:80-89)
    uint8_t *rob_sim_valid; // [E]
    float *rob_nd;          // [E]   neighbour distance frozen at creation
    float *rob_seen;        // [E][H] believed radii (+0.01 + safety space) frozen at creation
    double *tr;       // [E][R+1][4][H] px,py,vx,vy of roll k = 0..R; slice 0 unused (k = 1 reads the live state)
    uint8_t *vis;     // [E][H]
    double *min_dist; // [E]
    uint8_t *pend;    // [E] predict_truth only: 1 = the env was reset by the first half of the step (observation still to be written)
    // sim.human_num_range > 0 only (all null otherwise): H is then human_num + human_num_range = the lane stride and the number of
    // observation rows, and the crowd of env e is its first nh[e] slots (crowd_sim_var_num.py:103-104, :404-437)
    int32_t *nh;        // [E] len(self.humans)
    int32_t *nx_nh;     // [E] ... of the staged next episode
    int32_t *obs_cnt;   // [E] len(self.observed_human_ids)
    int32_t *obs_max;   // [E] max(self.observed_human_ids), -1 when empty
    uint8_t *sim_n;     // [E][H] agent count human i's private simulator was built for (orca.py:80-82 rebuilds on a change)
    uint8_t *rob_sim_n; // [E] ... the robot's (robot.policy == 'orca')
    // CrowdSimVarNumCollect-v0 only (crowd_sim_var_num_collect.py): prediction ids for the GST dataset
    int32_t *pred_id;   // [E][H] self.human_pred_id
    int32_t *max_pid;   // [E]    self.max_human_id
    uint8_t *last_obs;  // [E][H] self.last_human_observability
    int32_t *lp3_cnt;   // [1] agents of this step's ORCA pass whose linear program was infeasible (orca_lane_kernel -> orca_lp3_kernel)
    struct Lp3Hdr *lp3_hdr; // [E*H] where linearProgram2 stopped
    float4 *lp3_lines;  // [E*H][32] their ORCA lines (point, direction) in neighbour order
    double *desired_v;  // [E] unicycle robot only: self.desiredVelocity[0] (crowd_sim.py:82: set at construction, never reset)
    double *wheel;      // [E][4] unicycle robot in CrowdSimPred / PredRealGST: smooth_action's last_left, last_right (crowd_sim.py:84-85, never
                        // reset) and RandomState's cached normal deviate: value, has_gauss as 0 / 1 (cleared by every np.random.seed)
    unsigned long long *stamp; // launch stamps of THIS launch (common.h: cn_stamp_slot), set on the by-value copy a launch passes; NULL = none
};
// the by-value kernel argument of one launch, with the stamp slot of `kernel_id` for the current step (measurement aid)
static EnvDev stamped(const EnvDev &d, int kernel_id) { EnvDev c = d; c.stamp = cn_stamp_slot(kernel_id); return c; }

__device__ __forceinline__ int crowd_size(const EnvDev &s, int e) { return s.nh ? s.nh[e] : s.H; }

// The reference's rejection sampling of human positions / goals is unbounded; after this many attempts the last candidate is
// accepted (same constant and rule in the oracle: oracle/crowdsim_oracle.h ORC_MAX_PLACEMENT_ATTEMPTS).
constexpr int CN_MAX_PLACEMENT_ATTEMPTS = 1 << 16;

enum { F_PX = 0, F_PY, F_VX, F_VY, F_GX, F_GY, F_RAD, F_VPREF };
enum { R_PX = 0, R_PY, R_VX, R_VY, R_GX, R_GY, R_THETA, R_POT };

// ------------------------------------------------------------------------------------------------------------------
// deterministic sin/cos on [0, 2*pi]: Cody-Waite reduction by pi/2 + minimax kernels, +,-,* only.  Stands in for
// np.cos/np.sin (crowd_sim_var_num.py:127-128); documented in DESIGN.md (<= 1 ulp from libm).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double poly_sin(double x)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, w = z * z;
    const double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    const double v = z * x;
    return x + v * (S1 + z * r);
}
__device__ __forceinline__ double poly_cos(double x)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    double w = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * r);
}
__device__ __forceinline__ void det_sincos(double x, double &s, double &c)
{
    const double INV_PIO2 = 6.36619772367581382433e-01, PIO2_1 = 1.57079632673412561417e+00,
                 PIO2_1T = 6.07710050650619224932e-11;
    const int k = (int)(x * INV_PIO2 + 0.5);
    const double fk = (double)k;
    const double r = (x - fk * PIO2_1) - fk * PIO2_1T;
    const double sr = poly_sin(r), cr = poly_cos(r);
    switch (k & 3) {
    case 0: s = sr; c = cr; break;
    case 1: s = cr; c = -sr; break;
    case 2: s = -sr; c = -cr; break;
    default: s = -cr; c = sr; break;
    }
}

// deterministic exp: Cody-Waite reduction by ln 2 + degree-5 minimax kernel, +,-,*,/ only; the twin of the oracle's orc_exp.
// Stands in for np.exp in the social-force policy (crowd_nav/policy/social_force.py:37).
__device__ __forceinline__ double det_exp(double x)
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

// deterministic natural logarithm for normal positive arguments (classic reduction to sqrt(2)/2 < 1 + f < sqrt(2), degree-14 minimax in
// s = f / (2 + f)); the twin of the oracle's orc_log.  Stands in for log() in RandomState.normal's polar method (arguments in (0, 1)).
__device__ __forceinline__ double det_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int hx = (int)(bits >> 32);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i0 = (hx + 0x95f64) & 0x100000;
    bits = ((unsigned long long)(unsigned)(hx | (i0 ^ 0x3ff00000)) << 32) | (bits & 0xffffffffull);
    x = __longlong_as_double((long long)bits);
    k += i0 >> 20;
    const double f = x - 1.0;
    const double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {
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
    const int i = (hx - 0x6147a) | (0x6b851 - hx);
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// The field-of-view half of detect_visible (crowd_sim.py:513-537): agent 2 inside agent 1's cone of fov * pi radians around agent 1's
// heading -- the direction of its velocity when the robot is holonomic (at rest: +x, or -x for vx = -0.0, as np.arctan2 has it), its theta
// otherwise.  Decision-equivalent form of arccos(clip(v_fov . v_12)) <= fov / 2 (see the oracle's in_fov for why); coincident agents
// give NaN and are not visible.
__device__ __forceinline__ bool in_fov(const cn_env_config &c, double fov, double px1, double py1, double vx1, double vy1, double theta1,
                                       double px2, double py2)
{
    double fx, fy;
    if (c.kinematics == CN_KIN_UNICYCLE) det_sincos(theta1, fy, fx);
    else if (vx1 == 0.0 && vy1 == 0.0) { fx = __double_as_longlong(vx1) < 0 ? -1.0 : 1.0; fy = 0.0; }
    else { const double nv = sqrt(vx1 * vx1 + vy1 * vy1); fx = vx1 / nv; fy = vy1 / nv; }
    const double dx = px2 - px1, dy = py2 - py1;
    const double n12 = sqrt(dx * dx + dy * dy);
    double d = fx * (dx / n12) + fy * (dy / n12);
    d = d < -1.0 ? -1.0 : (d > 1.0 ? 1.0 : d); // keeps NaN, like np.clip
    const double half = M_PI * fov / 2.0;
    double thr = -1.0;
    if (half < M_PI) { double sn; det_sincos(half, sn, thr); }
    return d >= thr;
}

// ------------------------------------------------------------------------------------------------------------------
// Wave-cooperative RVO2 linear programs.  Lane k holds line k = (point, direction); `valid` marks live lines
// (bit k).  All scalars (result, t bounds, ...) are wave-uniform: every lane computes them identically.
// RVO2 v2.0.2 Agent.cpp linearProgram1/2/3; call site crowd_nav/policy/orca.py:113 (doStep).
// ------------------------------------------------------------------------------------------------------------------
struct LpLine { float px, py, dx, dy; };

__device__ __forceinline__ bool wv_any(bool p) { return __ballot(p) != 0ull; }

// Optimise along line i subject to the disc and to every earlier valid line (lane-parallel clip).  Returns success.
__device__ __forceinline__ bool lp1_wave(const LpLine &L, uint64_t valid, int i, float ipx, float ipy, float idx, float idy,
                                         float radius, float optx, float opty, bool dirOpt, int lane, float &rx, float &ry)
{
    const float dotProduct = ipx * idx + ipy * idy;
    const float discriminant = dotProduct * dotProduct + radius * radius - (ipx * ipx + ipy * ipy);
    if (discriminant < 0.0f) return false;
    const float sq = sqrtf(discriminant);
    float tLeft = -dotProduct - sq;
    float tRight = -dotProduct + sq;
    const bool mine = lane < i && ((valid >> lane) & 1ull);
    const float denominator = idx * L.dy - idy * L.dx;
    const float numerator = L.dx * (ipy - L.py) - L.dy * (ipx - L.px);
    const bool parallel = fabsf(denominator) <= RVO_EPS;
    const bool pfail = mine && parallel && numerator < 0.0f;
    const float t = numerator / denominator;
    const float candR = (mine && !parallel && denominator >= 0.0f) ? t : INFINITY;
    const float candL = (mine && !parallel && denominator < 0.0f) ? t : -INFINITY;
    tRight = fminf(tRight, wv_min(candR));
    tLeft = fmaxf(tLeft, wv_max(candL));
    // sequential RVO2 fails at the first prefix with tLeft > tRight or a parallel infeasible line; bounds are monotone,
    // so "any prefix fails" == "final bounds cross or any parallel line fails".
    if (wv_any(pfail) || tLeft > tRight) return false;
    float t_opt;
    if (dirOpt) {
        t_opt = (optx * idx + opty * idy > 0.0f) ? tRight : tLeft;
    } else {
        const float tt = idx * (optx - ipx) + idy * (opty - ipy);
        t_opt = tt < tLeft ? tLeft : (tt > tRight ? tRight : tt);
    }
    rx = ipx + t_opt * idx;
    ry = ipy + t_opt * idy;
    return true;
}

// Returns n on success, else the index of the line that failed.  RVO2 walks the lines in order and re-optimises at every
// line the current result violates; lines it does not violate are no-ops, so the walk jumps from violated line to
// violated line: every lane tests its own line against the current result, a ballot + ffs finds the next one.
__device__ __forceinline__ int lp2_wave(const LpLine &L, uint64_t valid, int n, float radius, float optx, float opty,
                                        bool dirOpt, int lane, float &rx, float &ry)
{
    if (dirOpt) {
        rx = radius * optx; ry = radius * opty;
    } else if (optx * optx + opty * opty > radius * radius) {
        const float inv = 1.0f / sqrtf(optx * optx + opty * opty);
        rx = radius * (optx * inv); ry = radius * (opty * inv);
    } else {
        rx = optx; ry = opty;
    }
    uint64_t todo = valid & (n >= 64 ? ~0ull : ((1ull << n) - 1ull));
    for (;;) {
        const uint64_t vm = __ballot(L.dx * (L.py - ry) - L.dy * (L.px - rx) > 0.0f) & todo;
        if (!vm) return n;
        const int i = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)vm) - 1);
        todo &= ~((2ull << i) - 1ull); // lines 0..i are behind us
        const float ipx = wv_readlane(L.px, i), ipy = wv_readlane(L.py, i);
        const float idx = wv_readlane(L.dx, i), idy = wv_readlane(L.dy, i);
        const float tx = rx, ty = ry;
        if (!lp1_wave(L, valid, i, ipx, ipy, idx, idy, radius, optx, opty, dirOpt, lane, rx, ry)) {
            rx = tx; ry = ty;
            return i;
        }
    }
}

__device__ __forceinline__ void lp3_wave(const LpLine &L, int n, int beginLine, float radius, int lane, float &rx, float &ry)
{
    float distance = 0.0f;
    uint64_t todo = (n >= 64 ? ~0ull : ((1ull << n) - 1ull)) & ~((1ull << beginLine) - 1ull);
    for (;;) {
        const uint64_t vm = __ballot(L.dx * (L.py - ry) - L.dy * (L.px - rx) > distance) & todo;
        if (!vm) return;
        const int i = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)vm) - 1);
        todo &= ~((2ull << i) - 1ull);
        const float ipx = wv_readlane(L.px, i), ipy = wv_readlane(L.py, i);
        const float idx = wv_readlane(L.dx, i), idy = wv_readlane(L.dy, i);
        // every lane j < i projects its line onto line i (RVO2 builds projLines sequentially; same set, same order)
        LpLine Pj;
        const float determinant = idx * L.dy - idy * L.dx;
        const bool par = fabsf(determinant) <= RVO_EPS;
        const bool skip = par && (idx * L.dx + idy * L.dy > 0.0f);
        if (par) {
            Pj.px = 0.5f * (ipx + L.px); Pj.py = 0.5f * (ipy + L.py);
        } else {
            const float s = (L.dx * (ipy - L.py) - L.dy * (ipx - L.px)) / determinant;
            Pj.px = ipx + s * idx; Pj.py = ipy + s * idy;
        }
        const float ddx = L.dx - idx, ddy = L.dy - idy;
        const float inv = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
        Pj.dx = ddx * inv; Pj.dy = ddy * inv;
        const uint64_t pvalid = __ballot(lane < i && !skip);
        const float tx = rx, ty = ry;
        if (lp2_wave(Pj, pvalid, i, radius, -idy, idx, true, lane, rx, ry) < i) { rx = tx; ry = ty; }
        distance = idx * (ipy - ry) - idy * (ipx - rx);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// linearProgram3 for TWO programs per wavefront: lanes 0..31 hold the lines of one, lanes 32..63 those of another (at most 32 lines
// each: the lane kernel's limit).  Per program the arithmetic is lp3_wave's; what is wave-uniform there (the result, the bounds, the
// index of the line being processed) is uniform per HALF here and lives in vector registers, each half's walk is predicated on its
// own state, and a loop ends when both halves are through.  hl = lane & 31.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hw_ballot(bool p, int lane)
{
    const uint64_t b = __ballot(p);
    return (lane & 32) ? (uint32_t)(b >> 32) : (uint32_t)b;
}
__device__ __forceinline__ float hw_read(float v, int lane, int i) // v of lane i of this lane's half (i uniform per half)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & 32) + i) << 2, __float_as_int(v)));
}
__device__ __forceinline__ float hw_last(float v, int lane) // lanes 31 / 63 -> every lane of their half
{
    const float a = wv_readlane(v, 31), b = wv_readlane(v, 63);
    return (lane & 32) ? b : a;
}
__device__ __forceinline__ float hw_min(float v, int lane)
{
    const float I = __builtin_inff();
    v = fminf(v, wv_dpp<0x111, 0xf>(v, I));
    v = fminf(v, wv_dpp<0x112, 0xf>(v, I));
    v = fminf(v, wv_dpp<0x114, 0xf>(v, I));
    v = fminf(v, wv_dpp<0x118, 0xf>(v, I));
    v = fminf(v, wv_dpp<0x142, 0xa>(v, I)); // row_bcast:15 into rows 1 and 3: lane 31 = lanes 0..31, lane 63 = lanes 32..63
    return hw_last(v, lane);
}
__device__ __forceinline__ float hw_max(float v, int lane)
{
    const float I = -__builtin_inff();
    v = fmaxf(v, wv_dpp<0x111, 0xf>(v, I));
    v = fmaxf(v, wv_dpp<0x112, 0xf>(v, I));
    v = fmaxf(v, wv_dpp<0x114, 0xf>(v, I));
    v = fmaxf(v, wv_dpp<0x118, 0xf>(v, I));
    v = fmaxf(v, wv_dpp<0x142, 0xa>(v, I));
    return hw_last(v, lane);
}

// lp1_wave with dirOpt = true for both halves at once; commits the new result only where `act` and the program is feasible
__device__ __forceinline__ bool lp1_pair(const LpLine &L, uint32_t valid, int i, float ipx, float ipy, float idx, float idy, float radius,
                                         float optx, float opty, bool act, int lane, float &rx, float &ry)
{
    const int hl = lane & 31;
    const float dotProduct = ipx * idx + ipy * idy;
    const float discriminant = dotProduct * dotProduct + radius * radius - (ipx * ipx + ipy * ipy);
    bool ok = !(discriminant < 0.0f);
    const float sq = sqrtf(discriminant);
    float tLeft = -dotProduct - sq;
    float tRight = -dotProduct + sq;
    const bool mine = hl < i && ((valid >> hl) & 1u);
    const float denominator = idx * L.dy - idy * L.dx;
    const float numerator = L.dx * (ipy - L.py) - L.dy * (ipx - L.px);
    const bool parallel = fabsf(denominator) <= RVO_EPS;
    const bool pfail = mine && parallel && numerator < 0.0f;
    const float t = numerator / denominator;
    const float candR = (mine && !parallel && denominator >= 0.0f) ? t : INFINITY;
    const float candL = (mine && !parallel && denominator < 0.0f) ? t : -INFINITY;
    tRight = fminf(tRight, hw_min(candR, lane));
    tLeft = fmaxf(tLeft, hw_max(candL, lane));
    const uint32_t pf = hw_ballot(pfail, lane); // (every cross-lane operation of these routines sits outside their predicated parts)
    ok = ok & (pf == 0u) & !(tLeft > tRight);
    const float t_opt = (optx * idx + opty * idy > 0.0f) ? tRight : tLeft;
    if (act && ok) {
        rx = ipx + t_opt * idx;
        ry = ipy + t_opt * idy;
    }
    return ok;
}

// lp2_wave with dirOpt = true over the lines 0 .. n-1 of each half (n, radius, opt uniform per half).  Returns n or the failing line.
__device__ __forceinline__ int lp2_pair(const LpLine &L, uint32_t valid, int n, float radius, float optx, float opty, bool act, int lane,
                                        float &rx, float &ry)
{
    if (act) { rx = radius * optx; ry = radius * opty; }
    uint32_t todo = valid & ((1u << n) - 1u); // n <= 31: line n itself is the one being projected on
    int res = n;
    bool running = act;
    for (;;) {
        const uint32_t vm = hw_ballot(L.dx * (L.py - ry) - L.dy * (L.px - rx) > 0.0f, lane) & todo;
        const bool go = running && vm != 0u;
        if (__ballot(go) == 0ull) return res;
        running = go; // a half without a violated line left is through
        const int i = go ? __ffs((int)vm) - 1 : 0;
        if (go) todo &= ~((2u << i) - 1u);
        const float ipx = hw_read(L.px, lane, i), ipy = hw_read(L.py, lane, i);
        const float idx = hw_read(L.dx, lane, i), idy = hw_read(L.dy, lane, i);
        const bool ok = lp1_pair(L, valid, i, ipx, ipy, idx, idy, radius, optx, opty, go, lane, rx, ry);
        if (go && !ok) { res = i; running = false; } // (lp1_pair left the result alone)
    }
}

__device__ __forceinline__ void lp3_pair(const LpLine &L, int n, int beginLine, float radius, bool act, int lane, float &rx, float &ry)
{
    const int hl = lane & 31;
    float distance = 0.0f;
    uint32_t todo = (n >= 32 ? ~0u : ((1u << n) - 1u)) & ~((1u << beginLine) - 1u);
    bool running = act;
    for (;;) {
        const uint32_t vm = hw_ballot(L.dx * (L.py - ry) - L.dy * (L.px - rx) > distance, lane) & todo;
        const bool go = running && vm != 0u;
        if (__ballot(go) == 0ull) return;
        running = go;
        const int i = go ? __ffs((int)vm) - 1 : 0;
        if (go) todo &= ~((2u << i) - 1u);
        const float ipx = hw_read(L.px, lane, i), ipy = hw_read(L.py, lane, i);
        const float idx = hw_read(L.dx, lane, i), idy = hw_read(L.dy, lane, i);
        LpLine Pj;
        const float determinant = idx * L.dy - idy * L.dx;
        const bool par = fabsf(determinant) <= RVO_EPS;
        const bool skip = par && (idx * L.dx + idy * L.dy > 0.0f);
        if (par) {
            Pj.px = 0.5f * (ipx + L.px); Pj.py = 0.5f * (ipy + L.py);
        } else {
            const float s = (L.dx * (ipy - L.py) - L.dy * (ipx - L.px)) / determinant;
            Pj.px = ipx + s * idx; Pj.py = ipy + s * idy;
        }
        const float ddx = L.dx - idx, ddy = L.dy - idy;
        const float inv = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
        Pj.dx = ddx * inv; Pj.dy = ddy * inv;
        const uint32_t pvalid = hw_ballot(hl < i && !skip, lane);
        const float tx = rx, ty = ry;
        const int f = lp2_pair(Pj, pvalid, i, radius, -idy, idx, go, lane, rx, ry);
        if (go && f < i) { rx = tx; ry = ty; }
        if (go) distance = idx * (ipy - ry) - idy * (ipx - rx);
    }
}

// One agent's new velocity.  Lane j < nl holds candidate neighbour j (cand == true) in index order.
// RVO2 Agent::computeNeighbors (range filter, ascending distSq, at most maxNeighbors) + computeNewVelocity.
__device__ __forceinline__ void orca_wave(int lane, int nl, bool cand, float opx, float opy, float ovx, float ovy, float orad,
                                          float spx, float spy, float svx, float svy, float srad, float maxspeed, float prefx,
                                          float prefy, float nd, int max_nb, float th, float dt, float &outx, float &outy)
{
    // neighbour selection: key = distSq if within range else +inf; rank by (key, index) -> stable ascending order
    const float ddx0 = spx - opx, ddy0 = spy - opy;
    const float dq = ddx0 * ddx0 + ddy0 * ddy0;
    const bool inrange = cand && dq < nd * nd;
    const float key = inrange ? dq : INFINITY;
    int rank = 0;
    for (int m = 0; m < nl; ++m) {
        const float km = wv_readlane(key, m);
        rank += (km < key || (km == key && m < lane)) ? 1 : 0;
    }
    if (lane >= nl) rank = lane;
    int nn = __popcll(__ballot(inrange));
    if (nn > max_nb) nn = max_nb;
    // ORCA half-plane of this lane's neighbour
    const float rpx = opx - spx, rpy = opy - spy;   // relativePosition
    const float rvx = svx - ovx, rvy = svy - ovy;   // relativeVelocity
    const float distSq = rpx * rpx + rpy * rpy;
    const float cr = srad + orad;
    const float crSq = cr * cr;
    float ldx, ldy, ux, uy;
    if (distSq > crSq) {
        const float invTH = 1.0f / th;
        const float wx = rvx - invTH * rpx, wy = rvy - invTH * rpy;
        const float wLenSq = wx * wx + wy * wy;
        const float dot1 = wx * rpx + wy * rpy;
        if (dot1 < 0.0f && dot1 * dot1 > crSq * wLenSq) {
            const float wLen = sqrtf(wLenSq);
            const float inv = 1.0f / wLen;
            const float uwx = wx * inv, uwy = wy * inv;
            ldx = uwy; ldy = -uwx;
            const float s = cr * invTH - wLen;
            ux = s * uwx; uy = s * uwy;
        } else {
            const float leg = sqrtf(distSq - crSq);
            const float invD = 1.0f / distSq;
            if (rpx * wy - rpy * wx > 0.0f) {
                ldx = (rpx * leg - rpy * cr) * invD;
                ldy = (rpx * cr + rpy * leg) * invD;
            } else {
                ldx = -((rpx * leg + rpy * cr) * invD);
                ldy = -((-rpx * cr + rpy * leg) * invD);
            }
            const float dot2 = rvx * ldx + rvy * ldy;
            ux = dot2 * ldx - rvx; uy = dot2 * ldy - rvy;
        }
    } else {
        const float invDT = 1.0f / dt;
        const float wx = rvx - invDT * rpx, wy = rvy - invDT * rpy;
        const float wLen = sqrtf(wx * wx + wy * wy);
        const float inv = 1.0f / wLen;
        const float uwx = wx * inv, uwy = wy * inv;
        ldx = uwy; ldy = -uwx;
        const float s = cr * invDT - wLen;
        ux = s * uwx; uy = s * uwy;
    }
    const float lpx = svx + 0.5f * ux, lpy = svy + 0.5f * uy;
    // scatter lines into sorted order: lane `rank` receives this lane's line
    LpLine L;
    L.px = __int_as_float(__builtin_amdgcn_ds_permute(rank << 2, __float_as_int(lpx)));
    L.py = __int_as_float(__builtin_amdgcn_ds_permute(rank << 2, __float_as_int(lpy)));
    L.dx = __int_as_float(__builtin_amdgcn_ds_permute(rank << 2, __float_as_int(ldx)));
    L.dy = __int_as_float(__builtin_amdgcn_ds_permute(rank << 2, __float_as_int(ldy)));
    const uint64_t valid = nn >= 64 ? ~0ull : ((1ull << nn) - 1ull);
    float rx, ry;
    const int lineFail = lp2_wave(L, valid, nn, maxspeed, prefx, prefy, false, lane, rx, ry);
    if (lineFail < nn) lp3_wave(L, nn, lineFail, maxspeed, lane, rx, ry);
    outx = rx; outy = ry;
}

// ------------------------------------------------------------------------------------------------------------------
// ORCA for every human of every env.  crowd_sim.py:680-703 get_human_actions + crowd_nav/policy/orca.py:64-117.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void orca_agent(const EnvDev &s, int agent, int lane)
{
    const int H = s.H;
    const int e = agent / H, i = agent - e * H;
    const int n = crowd_size(s, e); // humans present (== H unless sim.human_num_range > 0)
    if (i >= n) return;
    const double *hum = s.hum + (size_t)e * 8 * H;
    const bool isH = lane < n;
    const int lj = isH ? lane : 0;
    const double px = hum[F_PX * H + lj], py = hum[F_PY * H + lj], vx = hum[F_VX * H + lj], vy = hum[F_VY * H + lj];
    const double rad = hum[F_RAD * H + lj];
    // self (lane i) values, wave-uniform
    const double spx = __shfl(px, i, 64), spy = __shfl(py, i, 64), svx = __shfl(vx, i, 64), svy = __shfl(vy, i, 64);
    const double sgx = hum[F_GX * H + i], sgy = hum[F_GY * H + i], srad = hum[F_RAD * H + i], svpref = hum[F_VPREF * H + i];
    const double safety = s.cfg.orca_safety_space;
    // lazily (re)build human i's private simulator: orca.py:83-89
    const size_t ei = (size_t)e * H + i;
    float nd, self_r, self_ms, seen_r;
    const bool rv = s.cfg.robot_visible != 0;
    const int n_agents = n + (rv ? 1 : 0);
    // other humans as seen by i (human FOV = 2*pi: always the true state unless coincident, otherwise the ones inside i's cone; the rest
    // are the dummy (7,7,0,0)); with robot.visible the robot is appended as the last neighbour on lane n (crowd_sim.py:695-699), same
    // visibility rule
    const bool isR = rv && lane == n;
    const double *rob = s.rob + (size_t)e * 8;
    const double qx = isR ? rob[R_PX] : px, qy = isR ? rob[R_PY] : py, qvx = isR ? rob[R_VX] : vx, qvy = isR ? rob[R_VY] : vy;
    const bool coincident = s.cfg.human_fov < 2.0 ? !in_fov(s.cfg, s.cfg.human_fov, spx, spy, svx, svy, 0.0, qx, qy) : (qx == spx) && (qy == spy);
    if (!s.sim_valid[ei] || (s.sim_n && s.sim_n[ei] != n_agents)) {
        nd = (float)s.shared_nd[e];
        self_r = (float)(srad + 0.01 + safety);
        self_ms = (float)svpref;
        // addAgent takes the radius of the state it is handed: a human outside i's field of view right now is the dummy human with the
        // config radius, and keeps that size in this simulator
        seen_r = (float)((coincident ? s.cfg.human_radius : rad) + 0.01 + safety);
        if (s.sim_seen && isH) s.sim_seen[ei * H + lane] = seen_r;
        if (lane == 0) {
            s.sim_nd[ei] = nd; s.sim_self_radius[ei] = self_r; s.sim_self_maxspeed[ei] = self_ms; s.sim_valid[ei] = 1;
            if (s.sim_n) s.sim_n[ei] = (uint8_t)n_agents;
        }
    } else {
        nd = s.sim_nd[ei]; self_r = s.sim_self_radius[ei]; self_ms = s.sim_self_maxspeed[ei];
        seen_r = s.sim_seen ? s.sim_seen[ei * H + lj] : (float)(rad + 0.01 + safety);
    }
    if (isR) seen_r = (float)(s.cfg.robot_radius + 0.01 + safety); // fixed for the whole run
    const bool cand = (isH && lane != i) || isR;
    const float opx = coincident ? 7.0f : (float)qx, opy = coincident ? 7.0f : (float)qy;
    const float ovx = coincident ? 0.0f : (float)qvx, ovy = coincident ? 0.0f : (float)qvy;
    // preferred velocity: orca.py:97-100
    double gvx = sgx - spx, gvy = sgy - spy;
    const double speed = sqrt(gvx * gvx + gvy * gvy);
    if (speed > 1.0) { gvx = gvx / speed; gvy = gvy / speed; }
    float ox, oy;
    orca_wave(lane, n_agents, cand, opx, opy, ovx, ovy, seen_r, (float)spx, (float)spy, (float)svx, (float)svy, self_r, self_ms,
              (float)gvx, (float)gvy, nd, n_agents - 1, (float)s.cfg.orca_time_horizon, (float)s.cfg.time_step, ox, oy);
    if (lane == 0) {
        s.hact[(size_t)e * 2 * H + i] = ox;
        s.hact[(size_t)e * 2 * H + H + i] = oy;
    }
}

// The grid is capped (prefetch_orca): this kernel shares the chip with the policy forward on the caller's stream, and a resident-sized
// grid of wavefronts that walk the agents keeps its share of the issue slots bounded instead of flooding every SIMD.
__global__ __launch_bounds__(256) void orca_kernel(EnvDev s)
{
    const CnStampScope stamp_scope(s.stamp);
    const int lane = threadIdx.x & 63;
    const int total = s.E * s.H;
    for (int agent = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); agent < total; agent += gridDim.x * 4)
        orca_agent(s, agent, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// One LANE per agent (the common case of <= 32 agents in a crowd): the scalar RVO2 algorithm exactly as a CPU would run it,
// 64 agents per wavefront.  The kernel runs BEFORE the policy forward of the same step (prefetch_orca puts it on the caller's
// stream): next to the human-human kernel both slow down several-fold (that kernel saturates the L2 -> CU path this one's
// dependent loads queue behind), alone it takes ~1/10 of the step.
// Everything a lane indexes at run time lives in global memory (the env's agent records, L1-resident); everything it keeps in
// registers is indexed statically: the neighbour keys are ordered by a sorting network, the ORCA lines are built in that order,
// and the linear programs are fully unrolled over (line i, earlier line j).  linearProgram3 (the infeasible case, a few agents
// per thousand) would unroll to O(NB^3) code: those agents are put on a list and redone by the wave-cooperative routine above.
// Same arithmetic, same operation order as orca_wave / the oracle: results are bit-identical.
// ------------------------------------------------------------------------------------------------------------------
#include "orca_sortnet.inc"
#include "row_plan.h"

template <int W> struct LaneVec;
template <> struct LaneVec<8> { typedef float f __attribute__((ext_vector_type(8))); };
template <> struct LaneVec<32> { typedef float f __attribute__((ext_vector_type(32))); };

// NB = slots the sorting network orders (>= candidate neighbours incl. self), VW = width of the register vectors that hold the
// per-lane arrays.  The loops over neighbours / lines are ROLLED with wave-uniform counters: a vector element is then selected by
// a uniform register index (s_set_gpr_idx), not by 20-32 unrolled copies -- fully unrolled the kernel was 85 KB of straight-line
// code that every wavefront fetched exactly once (instruction-fetch bound, slower than the cooperative kernel).
template <int NB, int VW>
__global__ __launch_bounds__(64) void orca_lane_kernel(EnvDev s, const float *plan_det, int32_t *plan, int plan_groups, unsigned long long *plan_stamp)
{
    const CnStampScope stamp_scope((plan && (int)blockIdx.x < plan_groups) ? plan_stamp : s.stamp); // the plan builders' wavefronts have their own slot
    // the first workgroups (one wavefront each, rp_groups(E) of them) build the row plan of the policy's human-human kernel for the observation
    // that was just written (row_plan.h): they only need the detected-human counts, and this kernel is on the step's critical path anyway
    // (a builder's tables and the agents' line table below share one buffer: a workgroup is one or the other)
    constexpr int RAW = (int)sizeof(rowplan::Lds) > NB * 64 * 16 ? (int)sizeof(rowplan::Lds) : NB * 64 * 16;
    __shared__ __attribute__((aligned(16))) char s_raw[RAW];
    // (dispatched first; a builder is the longest chain of the launch, so it also takes the issue priority)
    if (plan && (int)blockIdx.x < plan_groups) {
        __builtin_amdgcn_s_setprio(3);
        rowplan::build((int)blockIdx.x, plan_groups, s.E, s.H, rp_workgroups(s.E, s.H), plan_det, plan, *reinterpret_cast<rowplan::Lds *>(s_raw), nullptr, s.plan_arrive);
        return;
    }
    const int blk = (int)blockIdx.x - (plan ? plan_groups : 0);
    typedef typename LaneVec<VW>::f vec;
    const int agent = blk * 64 + threadIdx.x;
    const int H = s.H;
    const bool live_lane = agent < s.E * H;
    const int e = live_lane ? agent / H : (blk * 64) / H, i = live_lane ? agent - e * H : 0;
    const int n = crowd_size(s, e);
    const bool active = live_lane && i < n; // (inactive lanes run along with nn = 0: the loop counters below must stay wave-uniform)
    const cn_env_config &c = s.cfg;
    // the agent records of the 1 + 63/H (+1) envs this wavefront's lanes belong to, staged once: every later access -- uniform in
    // pass 1, a per-lane gather in pass 2 -- is an LDS read instead of an L2 round trip (the kernel is a chain of dependent loads)
    // (the kernel for NB slots serves crowds of more than NB' agents, NB' the next smaller network: at most 63 / (NB' - 1) + 2 envs per wavefront)
    constexpr int NENV = NB == 8 ? 65 : (NB == 20 ? 10 : 5);
    __shared__ double s_px[128], s_py[128], s_vx[128], s_vy[128], s_rad[128], s_rob[NENV][4];
    {
        const int a0 = blk * 64;
        const int e0 = a0 / H, e1 = (min(a0 + 63, s.E * H - 1)) / H;
        const int nrows = (e1 - e0 + 1) * H; // <= 63 + 2 H <= 127 (H <= 32)
        for (int r = threadIdx.x; r < nrows; r += 64) {
            const int ee = e0 + r / H, j = r - (r / H) * H;
            const double *hm = s.hum + (size_t)ee * 8 * H;
            s_px[r] = hm[F_PX * H + j]; s_py[r] = hm[F_PY * H + j]; s_vx[r] = hm[F_VX * H + j]; s_vy[r] = hm[F_VY * H + j];
            s_rad[r] = hm[F_RAD * H + j];
        }
        if (c.robot_visible)
            for (int q = threadIdx.x; q <= e1 - e0; q += 64) {
                const double *rb = s.rob + (size_t)(e0 + q) * 8;
                s_rob[q][0] = rb[R_PX]; s_rob[q][1] = rb[R_PY]; s_rob[q][2] = rb[R_VX]; s_rob[q][3] = rb[R_VY];
            }
        __syncthreads();
    }
    const int eq = e - (blk * 64) / H, eb = eq * H; // this lane's env inside the staged block
    const double *hum = s.hum + (size_t)e * 8 * H;
    const double spx = s_px[eb + i], spy = s_py[eb + i], svx = s_vx[eb + i], svy = s_vy[eb + i], srad = s_rad[eb + i];
    const double sgx = hum[F_GX * H + i], sgy = hum[F_GY * H + i], svpref = hum[F_VPREF * H + i];
    const double safety = c.orca_safety_space;
    const bool rv = c.robot_visible != 0;
    const int n_agents = n + (rv ? 1 : 0);
    const size_t ei = (size_t)e * H + i;
    // lazily (re)build human i's private simulator: orca.py:80-89
    float nd = 0.0f, self_r = 0.0f, self_ms = 0.0f;
    if (active) {
        const bool rebuild = !s.sim_valid[ei] || (s.sim_n && s.sim_n[ei] != n_agents);
        if (rebuild) {
            nd = (float)s.shared_nd[e];
            self_r = (float)(srad + 0.01 + safety);
            self_ms = (float)svpref;
            if (s.sim_seen)
                for (int j = 0; j < n; ++j) s.sim_seen[ei * H + j] = (float)(s_rad[eb + j] + 0.01 + safety);
            s.sim_nd[ei] = nd; s.sim_self_radius[ei] = self_r; s.sim_self_maxspeed[ei] = self_ms; s.sim_valid[ei] = 1;
            if (s.sim_n) s.sim_n[ei] = (uint8_t)n_agents;
        } else {
            nd = s.sim_nd[ei]; self_r = s.sim_self_radius[ei]; self_ms = s.sim_self_maxspeed[ei];
        }
    }
    const float fpx = (float)spx, fpy = (float)spy, fvx = (float)svx, fvy = (float)svy;
    // pass 1: distance keys of the candidates in index order (slot j = agent j; self and empty slots get +inf)
    vec key, idx; // idx holds small integers as floats (exact)
    int nn = 0;
#pragma unroll 1
    for (int j = 0; j < NB; ++j) {
        const bool isR = rv && j == n;
        const bool cand = active && ((j < n && j != i) || isR);
        const int lj = j < n ? j : 0;
        const double qx = isR ? s_rob[eq][0] : s_px[eb + lj], qy = isR ? s_rob[eq][1] : s_py[eb + lj];
        const bool coincident = (qx == spx) && (qy == spy);
        const float opx = coincident ? 7.0f : (float)qx, opy = coincident ? 7.0f : (float)qy;
        const float ddx0 = fpx - opx, ddy0 = fpy - opy;
        const float dq = ddx0 * ddx0 + ddy0 * ddy0;
        const bool inrange = cand && dq < nd * nd;
        key[j] = inrange ? dq : INFINITY;
        idx[j] = (float)j;
        nn += inrange ? 1 : 0;
    }
    // ascending (distSq, index): RVO2's insertion order; the +inf slots end up behind the nn real neighbours
#define ORCA_CE(a, b)                                                                                  \
    {                                                                                                  \
        const float ka0 = key[a], kb0 = key[b], ia0 = idx[a], ib0 = idx[b];                             \
        const bool sw = kb0 < ka0 || (kb0 == ka0 && ib0 < ia0);                                         \
        key[a] = sw ? kb0 : ka0; key[b] = sw ? ka0 : kb0; idx[a] = sw ? ib0 : ia0; idx[b] = sw ? ia0 : ib0; \
    }
    if constexpr (NB == 8) { ORCA_SORTNET_8(ORCA_CE) }
    else if constexpr (NB == 20) { ORCA_SORTNET_20(ORCA_CE) }
    else { static_assert(NB == 32, "sorting networks exist for 8, 20 and 32 slots"); ORCA_SORTNET_32(ORCA_CE) }
#undef ORCA_CE
    int nmax = nn; // wave-uniform loop bound
    for (int off = 32; off >= 1; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    // pass 2: the ORCA half-plane of the k-th nearest neighbour (Agent::computeNewVelocity), kept in LDS as s_line[k][lane]: the
    // linear program below reads lines of OTHER lanes' agents at per-lane line numbers, which registers cannot do.
    // The three cases of RVO2 (cut-off circle, legs, collision) go through ONE sqrtf and ONE division whose operands are selected per
    // case -- the same operations on the same operands as the branchy form (no contraction in this file), so the same bits, but no
    // divergence: with 64 agents in a wavefront every branch was taken by somebody.
    const float invTH = 1.0f / (float)c.orca_time_horizon, invDT = 1.0f / (float)c.time_step;
    float4 *const s_line = reinterpret_cast<float4 *>(s_raw);
    const int tid = threadIdx.x;
#pragma unroll 1
    for (int k = 0; k < nmax; ++k) {
        float o_px = 0.0f, o_py = 0.0f, o_dx = 1.0f, o_dy = 0.0f;
        const int j = (int)idx[k];
        if (k < nn) {
            const bool isR = j == n; // only reachable when rv
            const int lj = isR ? 0 : j;
            const double qx = isR ? s_rob[eq][0] : s_px[eb + lj], qy = isR ? s_rob[eq][1] : s_py[eb + lj];
            const double qvx = isR ? s_rob[eq][2] : s_vx[eb + lj], qvy = isR ? s_rob[eq][3] : s_vy[eb + lj];
            float orad;
            if (isR) orad = (float)(c.robot_radius + 0.01 + safety); // fixed for the whole run
            else if (s.sim_seen) orad = s.sim_seen[ei * H + lj];
            else orad = (float)(s_rad[eb + lj] + 0.01 + safety);
            const bool coincident = (qx == spx) && (qy == spy);
            const float opx = coincident ? 7.0f : (float)qx, opy = coincident ? 7.0f : (float)qy;
            const float ovx = coincident ? 0.0f : (float)qvx, ovy = coincident ? 0.0f : (float)qvy;
            const float rpx = opx - fpx, rpy = opy - fpy;   // relativePosition
            const float rvx = fvx - ovx, rvy = fvy - ovy;   // relativeVelocity
            const float distSq = rpx * rpx + rpy * rpy;
            const float cr = self_r + orad;
            const float crSq = cr * cr;
            const bool collide = !(distSq > crSq);
            const float invT = collide ? invDT : invTH;
            const float wx = rvx - invT * rpx, wy = rvy - invT * rpy;
            const float wLenSq = wx * wx + wy * wy;
            const float dot1 = wx * rpx + wy * rpy;
            const bool circle = collide || (dot1 < 0.0f && dot1 * dot1 > crSq * wLenSq); // project on the cut-off circle
            const float sq = sqrtf(circle ? wLenSq : distSq - crSq);                        // wLen, or the leg length
            const float inv = 1.0f / (circle ? sq : distSq);                                // 1 / wLen, or 1 / distSq
            // cut-off circle (time horizon, or the time step on collision)
            const float uwx = wx * inv, uwy = wy * inv;
            const float sc = cr * invT - sq;
            // legs
            const bool left = rpx * wy - rpy * wx > 0.0f;
            const float lgx = left ? (rpx * sq - rpy * cr) * inv : -((rpx * sq + rpy * cr) * inv);
            const float lgy = left ? (rpx * cr + rpy * sq) * inv : -((-rpx * cr + rpy * sq) * inv);
            const float dot2 = rvx * lgx + rvy * lgy;
            const float ldx = circle ? uwy : lgx, ldy = circle ? -uwx : lgy;
            const float ux = circle ? sc * uwx : dot2 * lgx - rvx, uy = circle ? sc * uwy : dot2 * lgy - rvy;
            o_px = fvx + 0.5f * ux; o_py = fvy + 0.5f * uy; o_dx = ldx; o_dy = ldy;
        }
        s_line[k * 64 + tid] = make_float4(o_px, o_py, o_dx, o_dy);
    }
    // preferred velocity: orca.py:97-100
    double gvx = sgx - spx, gvy = sgy - spy;
    const double speed = sqrt(gvx * gvx + gvy * gvy);
    if (speed > 1.0) { gvx = gvx / speed; gvy = gvy / speed; }
    const float optx = (float)gvx, opty = (float)gvy, radius = self_ms;
    // linearProgram2 (optimise the preferred velocity, directionOpt = false)
    float rx, ry;
    if (optx * optx + opty * opty > radius * radius) {
        const float inv = 1.0f / sqrtf(optx * optx + opty * opty);
        rx = radius * (optx * inv); ry = radius * (opty * inv);
    } else {
        rx = optx; ry = opty;
    }
    // linearProgram1 of a violated line li cuts it against every earlier line lj < li of the same agent.  An agent violates ~1.2 of
    // its lines, but some agent of the 64 violates almost every line: a loop over lj run by the whole wavefront did 80 iterations per
    // wavefront for a handful of agents each time.  Instead the (violating agent, earlier line) pairs of a line are dealt to the 64
    // lanes, and the bounds of an agent are combined in LDS with integer min / max on order-preserving keys (min and max do not
    // depend on the order of their operands: the same tLeft / tRight as the sequential loop).
    // (their four 256-byte tables sit in row NB - 1 of the line table: an agent has at most NB - 1 neighbours.  The static LDS of a workgroup
    // stays below 1/6 of the CU's: five per CU would leave the 1281st workgroup of a 4096 x 20 batch waiting for a whole generation)
    unsigned *const s_tl = reinterpret_cast<unsigned *>(s_line + (NB - 1) * 64), *const s_tr = s_tl + 64;
    int *const s_pf = reinterpret_cast<int *>(s_tr + 64), *const s_vl = s_pf + 64;
    auto okey = [](float f) { const unsigned u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u); };
    auto okey_inv = [](unsigned o) { return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu)); };
    bool failed = false;
    int line_fail = 0;
#pragma unroll 1
    for (int li = 0; li < nmax; ++li) {
        const float4 Li = s_line[li * 64 + tid];
        const float ipx = Li.x, ipy = Li.y, idx_ = Li.z, idy = Li.w;
        const bool viol = li < nn && !failed && idx_ * (ipy - ry) - idy * (ipx - rx) > 0.0f;
        const unsigned long long vm = __ballot(viol);
        if (vm == 0ull) continue; // wave-uniform: nobody has to re-optimise on this line
        // linearProgram1 on line li against the disc ...
        const float dotProduct = ipx * idx_ + ipy * idy;
        const float discriminant = dotProduct * dotProduct + radius * radius - (ipx * ipx + ipy * ipy);
        bool ok = !(discriminant < 0.0f);
        const float sq = sqrtf(discriminant);
        float tLeft = -dotProduct - sq, tRight = -dotProduct + sq;
        bool pfail = false;
        // ... and against the earlier lines
        const int npairs = __popcll(vm) * li;
        if (npairs > 0) {
            if (viol) {
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(vm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vm, 0u));
                s_vl[rank] = tid; s_tl[tid] = okey(tLeft); s_tr[tid] = okey(tRight); s_pf[tid] = 0;
            }
            __syncthreads(); // (one wavefront: orders the LDS traffic, costs a waitcnt)
            const unsigned inv20 = ((1u << 20) + (unsigned)li - 1u) / (unsigned)li; // p / li = p * inv20 >> 20 for p (li - 1) < 2^20
#pragma unroll 1
            for (int p0 = 0; p0 < npairs; p0 += 64) {
                const int p = p0 + tid;
                const bool on = p < npairs;
                const int r = on ? (int)(((unsigned)p * inv20) >> 20) : 0;
                const int lj = on ? p - r * li : 0;
                const int a = s_vl[r];
                const float4 A = s_line[li * 64 + a], B = s_line[lj * 64 + a];
                const float denominator = A.z * B.w - A.w * B.z;
                const float numerator = B.z * (A.y - B.y) - B.w * (A.x - B.x);
                const bool parallel = fabsf(denominator) <= RVO_EPS;
                const float t = numerator / denominator;
                if (on) {
                    if (parallel) { if (numerator < 0.0f) s_pf[a] = 1; }
                    else if (denominator >= 0.0f) atomicMin(&s_tr[a], okey(t));
                    else atomicMax(&s_tl[a], okey(t));
                }
            }
            __syncthreads();
            if (viol) { tLeft = okey_inv(s_tl[tid]); tRight = okey_inv(s_tr[tid]); pfail = s_pf[tid] != 0; }
            __syncthreads(); // the slots are rewritten by the next violated line
        }
        // (sequential RVO2 fails at the first prefix that crosses; the bounds are monotone, so this is the same decision)
        ok = ok && !pfail && !(tLeft > tRight);
        if (viol) {
            if (ok) {
                const float tt = idx_ * (optx - ipx) + idy * (opty - ipy);
                const float t_opt = tt < tLeft ? tLeft : (tt > tRight ? tRight : tt);
                rx = ipx + t_opt * idx_;
                ry = ipy + t_opt * idy;
            } else {
                failed = true; // linearProgram3 needed
                line_fail = li;
            }
        }
    }
    // infeasible program: hand the lines and the state linearProgram2 stopped in to the wave-cooperative linearProgram3
    int slot = -1;
    if (active && failed) {
        slot = atomicAdd(s.lp3_cnt, 1);
        Lp3Hdr hd;
        hd.agent = agent; hd.nn = nn; hd.line_fail = line_fail; hd.rx = rx; hd.ry = ry; hd.radius = radius;
        s.lp3_hdr[slot] = hd;
    } else if (active) {
        s.hact[(size_t)e * 2 * H + i] = rx;
        s.hact[(size_t)e * 2 * H + H + i] = ry;
    }
    if (__ballot(slot >= 0) != 0ull) {
#pragma unroll 1
        for (int k = 0; k < nmax; ++k) {
            const float4 ln = s_line[k * 64 + tid];
            if (slot >= 0 && k < nn) s.lp3_lines[(size_t)slot * 32 + k] = ln;
        }
    }
}

// the agents orca_lane_kernel could not finish (infeasible program -> linearProgram3): two per wavefront, lane k of a half = line k
// calc_human_future_traj(method='truth') (crowd_sim_var_num.py:152-206), one roll per launch: every human acts with its own
// ORCA policy (act_joint_state -> ORCA.predict on its private simulator: frozen radii / neighbour distance) on the states
// predicted by roll k-1 and is stepped by one_step_lookahead (agent.py:185-192).  The other humans' states are passed as
// they are (no FOV / dummy substitution here).  Roll k needs all of roll k-1 of the same env -> one launch per roll.
__global__ __launch_bounds__(256) void orca_truth_kernel(EnvDev s, int k)
{
    const int lane = threadIdx.x & 63;
    const int agent = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (agent >= s.E * s.H) return;
    const int H = s.H;
    const int e = agent / H, i = agent - e * H;
    const int n = crowd_size(s, e);
    if (i >= n) return;
    const double *hum = s.hum + (size_t)e * 8 * H;
    double *trk = s.tr + ((size_t)e * (s.R + 1) + k) * 4 * H;
    const double *src = k == 1 ? hum : trk - 4 * H; // F_PX..F_VY are fields 0..3: the live state has the same [4][H] layout
    const bool isH = lane < n;
    const int lj = isH ? lane : 0;
    const double px = src[0 * H + lj], py = src[1 * H + lj], vx = src[2 * H + lj], vy = src[3 * H + lj];
    const double rad = hum[F_RAD * H + lj];
    const double spx = __shfl(px, i, 64), spy = __shfl(py, i, 64), svx = __shfl(vx, i, 64), svy = __shfl(vy, i, 64);
    const double sgx = hum[F_GX * H + i], sgy = hum[F_GY * H + i];
    const size_t ei = (size_t)e * H + i;
    float nd, self_r, self_ms, seen_r;
    if (!s.sim_valid[ei] || (s.sim_n && s.sim_n[ei] != n)) {
        // predict_method 'truth' as the observation predictor: the roll-out of a freshly reset env runs before any ORCA step, and
        // act_joint_state builds the private simulator exactly like ORCA.predict would (orca.py:83-89)
        const double safety = s.cfg.orca_safety_space;
        nd = (float)s.shared_nd[e];
        self_r = (float)(hum[F_RAD * H + i] + 0.01 + safety);
        self_ms = (float)hum[F_VPREF * H + i];
        seen_r = (float)(rad + 0.01 + safety);
        if (s.sim_seen && isH) s.sim_seen[ei * H + lane] = seen_r;
        if (lane == 0) {
            s.sim_nd[ei] = nd; s.sim_self_radius[ei] = self_r; s.sim_self_maxspeed[ei] = self_ms; s.sim_valid[ei] = 1;
            if (s.sim_n) s.sim_n[ei] = (uint8_t)n;
        }
    } else {
        nd = s.sim_nd[ei]; self_r = s.sim_self_radius[ei]; self_ms = s.sim_self_maxspeed[ei];
        seen_r = s.sim_seen ? s.sim_seen[ei * H + lj] : (float)(rad + 0.01 + s.cfg.orca_safety_space);
    }
    const bool cand = isH && lane != i;
    double gvx = sgx - spx, gvy = sgy - spy;
    const double speed = sqrt(gvx * gvx + gvy * gvy);
    if (speed > 1.0) { gvx = gvx / speed; gvy = gvy / speed; }
    float ox, oy;
    orca_wave(lane, n, cand, (float)px, (float)py, (float)vx, (float)vy, seen_r, (float)spx, (float)spy, (float)svx, (float)svy, self_r, self_ms,
              (float)gvx, (float)gvy, nd, n - 1, (float)s.cfg.orca_time_horizon, (float)s.cfg.time_step, ox, oy);
    if (lane == 0) {
        trk[0 * H + i] = spx + (double)ox * s.cfg.time_step;
        trk[1 * H + i] = spy + (double)oy * s.cfg.time_step;
        trk[2 * H + i] = (double)ox;
        trk[3 * H + i] = (double)oy;
    }
}

// The same roll-outs for humans.policy = 'social_force': act_joint_state -> SOCIAL_FORCE.predict (social_force.py:11-52) on the rolled
// states, the others being the H - 1 fellow humans with their true radii (no dummy substitution, no robot: crowd_sim_var_num.py:183-190).
// No solver and no private simulator: one wavefront per env (lane i = human i) walks all P rolls in one launch, the rolled states
// travel between the lanes by shuffles.  float64, same operation order as the step's own social-force block (env_step_kernel).
__global__ __launch_bounds__(64) void sf_truth_kernel(EnvDev s)
{
    const int e = blockIdx.x, lane = threadIdx.x;
    const int H = s.H, n = crowd_size(s, e);
    const cn_env_config &c = s.cfg;
    const double *hum = s.hum + (size_t)e * 8 * H;
    const bool isH = lane < n;
    const int lj = isH ? lane : 0;
    double px = hum[F_PX * H + lj], py = hum[F_PY * H + lj], vx = hum[F_VX * H + lj], vy = hum[F_VY * H + lj];
    const double rad = hum[F_RAD * H + lj], gx = hum[F_GX * H + lj], gy = hum[F_GY * H + lj], vpref = hum[F_VPREF * H + lj];
    for (int k = 1; k <= s.R; ++k) {
        const double dxg = gx - px, dyg = gy - py;
        const double dist_to_goal = sqrt(dxg * dxg + dyg * dyg);
        const double desired_vx = (dxg / dist_to_goal) * vpref, desired_vy = (dyg / dist_to_goal) * vpref;
        const double curr_dvx = c.sf_KI * (desired_vx - vx), curr_dvy = c.sf_KI * (desired_vy - vy);
        double ivx = 0.0, ivy = 0.0;
        for (int j = 0; j < n; ++j) {
            const double ox = __shfl(px, j, 64), oy = __shfl(py, j, 64), orad = __shfl(rad, j, 64);
            const double dx = px - ox, dy = py - oy;
            const double d = sqrt(dx * dx + dy * dy);
            const double f = c.sf_A * det_exp((rad + orad - d) / c.sf_B);
            if (j != lane) { ivx += f * (dx / d); ivy += f * (dy / d); }
        }
        const double nvx = vx + (curr_dvx + ivx) * c.time_step, nvy = vy + (curr_dvy + ivy) * c.time_step;
        const double act_norm = sqrt(nvx * nvx + nvy * nvy);
        double ax = nvx, ay = nvy;
        if (act_norm > vpref) { ax = nvx / act_norm * vpref; ay = nvy / act_norm * vpref; }
        // one_step_lookahead, agent.py:185-192 (every lane has read the old states: the shuffles above precede these writes)
        px = px + ax * c.time_step; py = py + ay * c.time_step; vx = ax; vy = ay;
        if (isH) {
            double *trk = s.tr + ((size_t)e * (s.R + 1) + k) * 4 * H;
            trk[0 * H + lane] = px; trk[1 * H + lane] = py; trk[2 * H + lane] = vx; trk[3 * H + lane] = vy;
        }
    }
}

// stand-alone batched solve (cn_orca_solve)
__global__ __launch_bounds__(256) void orca_solve_kernel(int B, int n_other, const float *self, const float *others, float nd,
                                                         int max_nb, float th, float dt, float *out)
{
    const int lane = threadIdx.x & 63;
    const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (b >= B) return;
    const float *sp = self + (size_t)b * 8;
    const bool cand = lane < n_other;
    const float *o = others + ((size_t)b * n_other + (cand ? lane : 0)) * 5;
    float ox, oy;
    orca_wave(lane, n_other, cand, o[0], o[1], o[2], o[3], o[4], sp[0], sp[1], sp[2], sp[3], sp[4], sp[5], sp[6], sp[7], nd,
              max_nb, th, dt, ox, oy);
    if (lane == 0) { out[2 * b] = ox; out[2 * b + 1] = oy; }
}

// ------------------------------------------------------------------------------------------------------------------
// MT19937 (numpy legacy RandomState) staged in LDS, wave-uniform draws.  The state belongs to ONE wavefront (R.mt: the block's array in the
// one-wavefront kernels, a per-wavefront slice in the ORCA tail kernel that also hosts the episode generator), so everything that orders its
// LDS traffic is wave-level: LDS operations of a wavefront are executed in issue order, the fence only keeps the compiler from moving them.
// ------------------------------------------------------------------------------------------------------------------
__shared__ uint32_t g_mt_lds[MT_N]; // the staged MT19937 state of a one-wavefront block
struct Rng {
    int pos;
    bool loaded;
    uint32_t *mt = g_mt_lds;
};
__device__ __forceinline__ void rng_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

__device__ __forceinline__ void rng_load(Rng &R, const EnvDev &s, int e, int lane)
{
    if (R.loaded) return;
    for (int k = lane; k < MT_N; k += 64) R.mt[k] = s.mt[(size_t)e * MT_N + k];
    R.pos = s.mt_pos[e];
    R.loaded = true;
    rng_sync();
}
__device__ __forceinline__ void rng_store(Rng &R, const EnvDev &s, int e, int lane)
{
    if (!R.loaded) return;
    rng_sync();
    for (int k = lane; k < MT_N; k += 64) s.mt[(size_t)e * MT_N + k] = R.mt[k];
    if (lane == 0) s.mt_pos[e] = R.pos;
}
// np.random.seed(int) == init_genrand: serial recurrence, computed redundantly by all lanes (wave-uniform)
__device__ __forceinline__ void rng_seed(Rng &R, uint32_t seed, int lane)
{
    rng_sync();
    // (the seed comes out of vector loads: without this the 624-step chain runs on the vector ALU -- shift, xor, a quarter-rate 32-bit
    // multiply and an add per step, ~13 us -- instead of four scalar instructions)
    uint32_t sd = (uint32_t)__builtin_amdgcn_readfirstlane((int)seed);
    for (int base = 0; base < MT_N; base += 64) {
        uint32_t mine = 0;
        for (int t = 0; t < 64; ++t) {
            const int pos = base + t;
            if (pos < MT_N) {
                if (t == lane) mine = sd;
                sd = 1812433253u * (sd ^ (sd >> 30)) + (uint32_t)pos + 1u;
            }
        }
        if (base + lane < MT_N) R.mt[base + lane] = mine;
    }
    R.pos = MT_N;
    R.loaded = true;
    rng_sync();
}
__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// lane-parallel regeneration of the 624-word block; dependencies are at distance 227 (>= 64), so 64-wide chunks
// processed in order reproduce the sequential recurrence exactly.
__device__ __forceinline__ void mt_twist_buf(uint32_t *k, int lane)
{
    rng_sync();
    for (int base = 0; base < 227; base += 64) {
        const int i = base + lane;
        const bool act = i < 227;
        uint32_t a = 0, b = 0, c = 0;
        if (act) { a = k[i]; b = k[i + 1]; c = k[i + 397]; }
        rng_sync();
        if (act) k[i] = mt_mix(a, b, c);
        rng_sync();
    }
    for (int base = 227; base < 623; base += 64) {
        const int i = base + lane;
        const bool act = i < 623;
        uint32_t a = 0, b = 0, c = 0;
        if (act) { a = k[i]; b = k[i + 1]; c = k[i - 227]; }
        rng_sync();
        if (act) k[i] = mt_mix(a, b, c);
        rng_sync();
    }
    if (lane == 0) k[623] = mt_mix(k[623], k[0], k[396]);
    rng_sync();
}
__device__ __forceinline__ void rng_twist(Rng &R, int lane)
{
    mt_twist_buf(R.mt, lane);
    R.pos = 0;
}
__device__ __forceinline__ uint32_t rng_u32(Rng &R, int lane)
{
    if (R.pos == MT_N) rng_twist(R, lane);
    uint32_t y = R.mt[R.pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
// random_sample(): 53-bit double from two words
__device__ __forceinline__ double rng_double(Rng &R, int lane)
{
    const uint32_t a = rng_u32(R, lane) >> 5, b = rng_u32(R, lane) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}
__device__ __forceinline__ double rng_uniform(Rng &R, int lane, double lo, double hi) { return lo + (hi - lo) * rng_double(R, lane); }
// np.random.normal(loc, scale) of the legacy RandomState: loc + scale * legacy_gauss (polar Box-Muller, the second deviate of a pair is
// cached for the next call).  gauss / has_gauss are the caller's copies of the cache (wave-uniform).
__device__ __forceinline__ double rng_normal(Rng &R, int lane, double loc, double scale, double &gauss, bool &has_gauss)
{
    double g;
    if (has_gauss) { g = gauss; has_gauss = false; gauss = 0.0; }
    else {
        double x1, x2, r2;
        do {
            x1 = 2.0 * rng_double(R, lane) - 1.0;
            x2 = 2.0 * rng_double(R, lane) - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        const double f = sqrt(-2.0 * det_log(r2) / r2);
        gauss = f * x1; has_gauss = true;
        g = f * x2;
    }
    return loc + scale * g;
}
// legacy RandomState.randint(low, high), default int64 dtype (numpy/random/_bounded_integers: _rand_int64 -> masked rejection on 32-bit
// words): no draw when the range is a single value
__device__ __forceinline__ int rng_randint(Rng &R, int lane, int low, int high)
{
    const uint32_t rng = (uint32_t)(high - 1 - low);
    if (rng == 0) return low;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = rng_u32(R, lane) & mask; } while (v > rng);
    return low + (int)v;
}

// ------------------------------------------------------------------------------------------------------------------
// Per-env wavefront state: lane j owns human j.
// ------------------------------------------------------------------------------------------------------------------
struct Lane {
    double px, py, vx, vy, gx, gy, rad, vpref; // human j
    double l0, l1, l2, l3, l4;                  // last_human_states[j]
    uint8_t simv;
};
struct Robot { double px, py, vx, vy, gx, gy, theta, pot; };

__device__ __forceinline__ double norm2(double x, double y) { return sqrt(x * x + y * y); }
// norm2(x, y) < d, decided without the square root whenever the squared distance is not within a few ulps of d * d: sqrt is correctly
// rounded and monotone, so outside that band the comparison of the squares gives the same answer; inside it (practically never) the
// reference expression itself is evaluated.  d >= 0.
__device__ __forceinline__ bool closer_than(double x, double y, double d)
{
    const double q = x * x + y * y, dd = d * d;
    if (q < dd * (1.0 - 0x1p-48)) return true;
    if (q > dd * (1.0 + 0x1p-48)) return false;
    return sqrt(q) < d;
}

__device__ __forceinline__ double wv_readlane_d(double v, int lane_uniform)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), lane_uniform);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), lane_uniform);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// ---- a long rejection loop over the W wavefronts of a workgroup (dense crowds, BASELINE configs[4]) ----
// In a crowd of ~50 randomised humans a placement takes 4 candidates in the median, one in 40 more than 64, and one in 10^4 runs to the bound
// of 65 536: ~1000 passes of one wavefront, 3 ms, while the other 8191 envs of the step are long done -- and with ~400 envs of a batch changing
// 25 goals each in a step, nearly every step has one.  The candidates are a pure function of the MT19937 stream (candidate j of a loop that
// starts at stream word g reads the words g + 6 j .. g + 6 j + 5, whatever its fate), and the stream is a recurrence with a lag of 227 words:
//     s[m] = s[m - 227] ^ f(s[m - 624], s[m - 623]),
// so one wavefront can run it 192 words at a time without ever waiting for anybody else.  Once a loop has run COOP_AFTER candidates on its
// own, the workgroup's other wavefronts (parked at a barrier until then) join in, in rounds of 64 (W - 1) candidates: the last wavefront is
// the PRODUCER -- while the others evaluate round r it extends the stream, in a ring of LDS blocks, as far as round r + 1 reads -- the master
// and the W - 2 helpers put 64 candidates each through a COARSE fp32 screen (coop_screen_pass: "collides for certain", with 1e-3 of slack on
// the squared thresholds; a bound-hitting loop is 65 537 candidates x ~100 points, and one CU evaluates ~400 candidates per microsecond this
// way whatever W is), and the master re-evaluates with the exact walk, in stream order, the passes that reported a candidate the screen
// could not reject: the first accepted candidate IN STREAM ORDER wins (or the first one past the bound) -- the same candidate, and the
// same staged block and position afterwards, as the serial loop, whose own cutting of the stream into passes has no influence on either.
// One workgroup barrier per round; every thread tracks the round's stream position itself, and the master only speaks up (a second
// barrier) in rounds where some wavefront reported something.
// groups of 64 candidates one evaluating wavefront screens per round, C per lane: a pair of points is read from LDS once (a broadcast read of
// 24 bytes per lane: 12 clocks of the CU's LDS pipe) and tested against C candidates (6 C packed instructions), so with C = 1 four busy SIMDs
// ask for twice what the LDS delivers
#ifndef CN_COOP_C4
#define CN_COOP_C4 4
#endif
#ifndef CN_COOP_C8
#define CN_COOP_C8 2
#endif
#ifndef CN_COOP_C16
#define CN_COOP_C16 1
#endif
constexpr int coop_c(int W) { return W <= 4 ? CN_COOP_C4 : (W <= 8 ? CN_COOP_C8 : CN_COOP_C16); }
template <int W>
struct CoopLds {
    static constexpr int NE = W - 1;                          // evaluating wavefronts (master + helpers)
    static constexpr int C = coop_c(W);
    static constexpr int NG = NE * C;                         // groups of 64 candidates per round
    static constexpr int NB1 = (384 * NG + 623) / MT_N;       // new blocks a round can need
    static constexpr int BW = MT_N * (NB1 + 1) + 227;         // one buffer: the last block of the round before, the new ones, the producer's overshoot
    int cmd;                        // 1 = a placement is published, 2 = the kernel is over
    int verdict;                    // the master's answer in a round with reports: 1 = the placement is over, 0 = next round
    int kind, n_pairs, max_att;
    int pos0, attempt0;             // position in the staged block / candidate number of the first cooperative candidate
    float circle_radius, vp;
    // the blocking points two by two, as the coarse screen reads them (one broadcast read per pair): {x0, x1, y0, y1} and the squared
    // thresholds minus the slack; pair 0 = the robot's goal and position, then the master's packed lists (goals, positions), the last
    // point twice when the count is odd
    float4 pxy[66];
    float2 plo[66];
    // round r reads buf[r & 1]: the stream LINEARLY from block b1(r - 1) (the last block round r - 1 touched; block 0 = the staged one, for
    // round 0) to block b1(r), whole blocks; the producer fills buf[(r + 1) & 1] meanwhile
    uint32_t buf[2][BW];
    unsigned long long take[2][NG]; // by round parity, per group
};
#ifdef CN_POST_DEBUG
__device__ long long g_post_dbg[8192 * 8]; // per block: ticks total, ticks in coop, coop placements, coop rounds, placements, serial passes, start tick, -
__shared__ long long g_dbg_blk[8];
#define DBG_ADD(i, v) do { if (lane == 0) g_dbg_blk[i] += (v); } while (0)
#else
#define DBG_ADD(i, v) do { } while (0)
#endif
constexpr int COOP_AFTER = 128;     // candidates a loop evaluates alone before the helpers join (98.5 % of the loops end earlier)
template <int W>
__device__ __forceinline__ CoopLds<W> &coop_lds()
{
    __shared__ CoopLds<W> q; // (only kernels instantiated with W > 1 reference it)
    return q;
}
// ONE wavefront: dst[0 .. 623] = hist[0 .. 623] (a complete block), then n_new more words of the stream behind it, 227 per iteration with a
// fixed word -> (step, lane) mapping: the lag-227 operand of a word is then the word the same lane made in the same step of the iteration
// before -- it never leaves its register -- and the other two operands (624 and 623 words back) were written at least one whole iteration
// earlier by this same wavefront (the LDS executes a wavefront's accesses in order), so they are loaded one iteration ahead and nothing in
// the loop waits for a store.  May overshoot n_new by up to 226 words (correct stream words; the buffer has the room).
__device__ __forceinline__ void coop_produce(uint32_t *dst, const uint32_t *hist, int lane, int n_new)
{
    if (hist) {
        uint32_t t[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) t[k] = hist[64 * k + (k < 9 || lane < MT_N - 576 ? lane : 0)];
#pragma unroll
        for (int k = 0; k < 9; ++k) dst[64 * k + lane] = t[k];
        if (lane < MT_N - 576) dst[576 + lane] = t[9];
        rng_sync();
    }
    // word m (relative to dst + 624) of an iteration that starts at m0: step u, lane l <-> m = m0 + 64 u + l, 64 u + l < 227
    const bool last = lane < 227 - 192;
    uint32_t *p = dst + lane; // &dst[m0 + lane], m0 = 0: operands at p[64 u], p[64 u + 1]; lag-227 operand at p[64 u + 397]; result to p[64 u + 624]
    uint32_t far[4], a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { far[u] = p[64 * u + 397]; a[u] = p[64 * u]; b[u] = p[64 * u + 1]; } // (u = 3, lanes >= 35: read but never used)
    for (int m = 0; m < n_new; m += 227, p += 227) {
        uint32_t na[4], nb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { na[u] = p[227 + 64 * u]; nb[u] = p[227 + 64 * u + 1]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t y = (a[u] & 0x80000000u) | (b[u] & 0x7fffffffu);
            far[u] = far[u] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            if (u < 3 || last) p[64 * u + MT_N] = far[u];
            a[u] = na[u]; b[u] = nb[u];
        }
        __builtin_amdgcn_wave_barrier();
    }
    rng_sync();
}

// groups p C .. p C + C - 1 (64 candidates each, C per lane) of the round whose first candidate starts at word `off` of `rb`: which candidates can
// the coarse screen NOT reject (or lie past the bound)?  -> Q.take[round & 1][group]  Candidate and squares in fp32 from the 27 high bits of each double's first word (the angle's sine and cosine from
// V_SIN_F32 / V_COS_F32, whose argument is in revolutions): the candidate is within ~1e-5 of the fp64 one, a square near md^2 ~ 1 within 3e-5
// of the true one, and "square < md^2 (1 - 1e-3)" therefore means closer than md for certain.  The other direction is not needed: whatever
// is not rejected here is evaluated by the master, exactly.
template <int W>
__device__ __forceinline__ void coop_screen_pass(CoopLds<W> &Q, const uint32_t *rb, int off, int att0, int p, int lane, int round,
                                                 int kind, int n_pairs, int max_att, float radius, float vp)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int C = CoopLds<W>::C;
    f2 xx[C], yy[C];
    float m[C]; // min over the points of (square - lowered threshold)
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const uint32_t *w = rb + off + 6 * (64 * (p * C + cc) + lane);
        const float u0 = (float)(mt_temper(w[0]) >> 5) * 0x1p-27f, u1 = (float)(mt_temper(w[2]) >> 5) * 0x1p-27f, u2 = (float)(mt_temper(w[4]) >> 5) * 0x1p-27f;
        const float cs = __builtin_amdgcn_cosf(u0), sn = __builtin_amdgcn_sinf(u0);
        const float nx = kind == 0 ? u1 * 2.0f : (u1 - 0.5f) * vp, ny = kind == 0 ? u2 * 2.0f : (u2 - 0.5f) * vp;
        const float xf = radius * cs + nx, yf = radius * sn + ny;
        xx[cc] = f2{xf, xf}; yy[cc] = f2{yf, yf};
        m[cc] = 1.0f;
    }
    for (int k = 0; k < n_pairs; ++k) {
        const float4 pq = Q.pxy[k];
        const float2 lo = Q.plo[k];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const f2 ax = xx[cc] - f2{pq.x, pq.y}, ay = yy[cc] - f2{pq.z, pq.w};
            const f2 d = (ax * ax + ay * ay) - f2{lo.x, lo.y};
            m[cc] = fminf(m[cc], fminf(d.x, d.y));
        }
    }
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const uint64_t take = __ballot(!(m[cc] < 0.0f) || att0 + 64 * (p * C + cc) + lane >= max_att);
        if (lane == 0) Q.take[round & 1][p * C + cc] = take;
    }
}

// The reference's placement loops (crowd_sim_var_num.py:116-146 positions, crowd_sim.py:415-485 goals) are rejection sampling: candidate k
// is made of the stream's next three doubles (angle, x noise, y noise), and the first candidate that keeps its distance from the robot
// and from every human of the list is taken.  One candidate costs six words of the MT19937 stream whatever its fate, so candidate k of a
// loop that starts at stream position p reads the words p + 6 k .. p + 6 k + 5: the candidates inside the current 624-word block are
// evaluated 64 AT A TIME, one per lane (each lane walks the human list itself: human j's state comes out of lane j by v_readlane), and
// the first accepted one in stream order wins -- the same candidate, the same stream position afterwards, as the one-at-a-time loop.
// A candidate whose six words straddle the end of the block rides as lane 0 of the first pass over the regenerated block.  In crowds of ~50 randomised humans these loops run for 10^2 .. 10^5 candidates (BASELINE configs[4]).
//   kind 0: position of a new human (noise = u * 2),  kind 1: new goal (noise = (u - 0.5) * vp)
//   humans 0 .. n_list - 1 except `skip` are tested with md = radius + rad_j + discomfort_dist against their position and their goal
template <int W = 1>
__device__ __forceinline__ void place_by_rejection(const EnvDev &s, Rng &R, int lane, int kind, double radius, double vp, double md_r, int n_list, int skip,
                                                   const Robot &rb, const Lane &h, double &out_x, double &out_y)
{
    const cn_env_config &c = s.cfg;
    const int max_att = c.max_placement_attempts > 0 ? c.max_placement_attempts : CN_MAX_PLACEMENT_ATTEMPTS;
    auto make = [&](double u0, double u1, double u2, double &x, double &y) {
        const double angle = u0 * M_PI * 2.0;
        const double nx = kind == 0 ? (0.0 + (1.0 - 0.0) * u1) * 2.0 : (u1 - 0.5) * vp;
        const double ny = kind == 0 ? (0.0 + (1.0 - 0.0) * u2) * 2.0 : (u2 - 0.5) * vp;
        double sn, cs;
        det_sincos(angle, sn, cs);
        x = c.circle_radius * cs + nx;
        y = c.circle_radius * sn + ny;
    };
    // does candidate (x, y) of this lane collide?  `live`: lanes whose answer matters (the walk ends once all of them have collided).
    // The walk decides `norm2(d) < md` on the SQUARES: q < md^2 (1 - 2^-48) means closer, q > md^2 (1 + 2^-48) means not (closer_than's
    // argument); a square inside that band (practically never) only marks the lane, and marked lanes that found no collision are walked
    // again with the reference expression itself.  (With the square root inside the walk -- the compiler evaluates it for every lane that
    // is not clearly closer, i.e. nearly always -- a (candidate, human) pair cost ~90 fp64 instructions instead of ~20.)
    // lane j keeps human j's thresholds: md_j = radius + rad_j + discomfort_dist
    const double md_l = radius + h.rad + c.discomfort_dist, dd_l = md_l * md_l;
    const double lo_l = dd_l * (1.0 - 0x1p-48), hi_l = dd_l * (1.0 + 0x1p-48);
    const double ddr = md_r * md_r, lo_r = ddr * (1.0 - 0x1p-48), hi_r = ddr * (1.0 + 0x1p-48);
    auto collides_exact = [&](double x, double y) {
        bool coll = norm2(x - rb.px, y - rb.py) < md_r || norm2(x - rb.gx, y - rb.gy) < md_r;
        for (int j = 0; j < n_list; ++j) {
            if (j == skip) continue;
            const double jx = wv_readlane_d(h.px, j), jy = wv_readlane_d(h.py, j), jgx = wv_readlane_d(h.gx, j), jgy = wv_readlane_d(h.gy, j);
            const double md = radius + wv_readlane_d(h.rad, j) + c.discomfort_dist;
            coll = coll || norm2(x - jx, y - jy) < md || norm2(x - jgx, y - jgy) < md;
        }
        return coll;
    };
    // Which (human, point) pairs can block a candidate at all?  Every candidate lies within n_max of the circle of radius R (its noise), so a
    // point whose distance from the origin is not inside (R - n_max - md, R + n_max + md) cannot come closer than md to any of them: mid-episode
    // most humans' POSITIONS are far inside the circle and drop out; the goals sit on it.  (1e-3 of slack for the rounding of cos / sin.)
    const double n_max = kind == 0 ? 2.0 * 1.4142135623730951 : 0.70710678118654757 * vp;
    const double w_l = n_max + md_l + 1e-3, r_in = c.circle_radius - w_l, r_out = c.circle_radius + w_l;
    const double in2 = r_in > 0.0 ? r_in * r_in : -1.0, out2 = r_out * r_out;
    const bool listed = lane < n_list && lane != skip;
    const double hp2 = h.px * h.px + h.py * h.py, hg2 = h.gx * h.gx + h.gy * h.gy;
    const uint64_t pos_mask = __ballot(listed && hp2 > in2 && hp2 < out2), goal_mask = __ballot(listed && hg2 > in2 && hg2 < out2);
    // The verdicts are kept as two running minima instead of lane masks (a mask update per test is a dozen scalar instructions; a
    // v_min_f64 is one): with d = q - lo,  closer  <=>  d < 0  (an IEEE difference has the sign of the comparison), and
    // inside the band  <=>  lo <= q <= hi  <=>  max(-d, q - hi) <= 0.
    auto collides64 = [&](double x, double y, bool live) {
        double ax = x - rb.px, ay = y - rb.py, bx = x - rb.gx, by = y - rb.gy;
        double q1 = ax * ax + ay * ay, q2 = bx * bx + by * by;
        double d1 = q1 - lo_r, d2 = q2 - lo_r;
        double cmin = fmin(d1, d2);
        double bmin = fmin(fmax(-d1, q1 - hi_r), fmax(-d2, q2 - hi_r));
        for (uint64_t m = goal_mask; m; m &= m - 1) {
            const int j = __ffsll((unsigned long long)m) - 1;
            const double jx = wv_readlane_d(h.gx, j), jy = wv_readlane_d(h.gy, j), lo = wv_readlane_d(lo_l, j), hi = wv_readlane_d(hi_l, j);
            ax = x - jx; ay = y - jy;
            q1 = ax * ax + ay * ay;
            d1 = q1 - lo;
            cmin = fmin(cmin, d1);
            bmin = fmin(bmin, fmax(-d1, q1 - hi));
        }
        for (uint64_t m = pos_mask; m; m &= m - 1) {
            const int j = __ffsll((unsigned long long)m) - 1;
            const double jx = wv_readlane_d(h.px, j), jy = wv_readlane_d(h.py, j), lo = wv_readlane_d(lo_l, j), hi = wv_readlane_d(hi_l, j);
            ax = x - jx; ay = y - jy;
            q1 = ax * ax + ay * ay;
            d1 = q1 - lo;
            cmin = fmin(cmin, d1);
            bmin = fmin(bmin, fmax(-d1, q1 - hi));
        }
        bool coll = cmin < 0.0;
        const bool unsure = bmin <= 0.0;
        if (__ballot(live && unsure && !coll) != 0ull) { // some square sat inside the band: the reference expression decides (all lanes walk again)
            const bool exact = collides_exact(x, y);
            if (unsure && !coll) coll = exact;
        }
        return coll;
    };
    // fp32 SCREEN in front of that walk.  A capped loop of a dense crowd is 65 536 candidates x ~100 points, and the walk above costs ~24
    // instructions per (candidate, point) of the one wavefront an env has.  In fp32, with two points as the two halves of packed
    // instructions, a pair of tests costs ~20: candidate and points rounded to float (|coordinate| < 32: 2^-20 absolute), the
    // square from a packed multiply + fma, compared with thresholds moved apart by 2e-5 relative -- several times what the roundings can
    // move a square near md^2 (|q32 - q| <= 2 |d| 3e-6 + 3e-7 q: 4e-6 relative at |d| ~ 1).  A candidate with some square below the lower
    // threshold collides, one with every square above the upper ones does not; anything else (a few candidates per million) sends the
    // batch through the fp64 walk.  Rounding of the thresholds themselves: 6e-8 relative, inside the 2e-5.
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float lor32 = (float)(ddr * (1.0 - 2e-5)), hir32 = (float)(ddr * (1.0 + 2e-5));
    const float rpx32 = (float)rb.px, rpy32 = (float)rb.py, rgx32 = (float)rb.gx, rgy32 = (float)rb.gy;
    // the points that can block (goals first, then positions) are packed into consecutive lanes once per placement -- lane k keeps point k
    // and its thresholds -- so that the walk takes them two at a time without caring which human they belong to (~65 points in a dense
    // crowd mid-episode: 33 packed steps instead of 50 human-by-human ones)
    const int n_g = __popcll(goal_mask), n_p = __popcll(pos_mask), n_pts = n_g + n_p; // <= 128: two lists of <= 64
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    auto pack = [&](uint64_t mask, int cnt, float v) {
        // lane j with its bit set sends v to lane rank(j); the others to the lanes behind the list (a full permutation: no two senders share a lane)
        const bool on = (mask >> lane) & 1ull;
        const int dst = on ? __popcll(mask & below) : cnt + __popcll(~mask & below);
        return __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(v)));
    };
    const float lo32 = (float)(dd_l * (1.0 - 2e-5)), hi32 = (float)(dd_l * (1.0 + 2e-5));
    const float Gx = pack(goal_mask, n_g, (float)h.gx), Gy = pack(goal_mask, n_g, (float)h.gy), Gl = pack(goal_mask, n_g, lo32), Gh = pack(goal_mask, n_g, hi32);
    const float Px = pack(pos_mask, n_p, (float)h.px), Py = pack(pos_mask, n_p, (float)h.py), Pl = pack(pos_mask, n_p, lo32), Ph = pack(pos_mask, n_p, hi32);
    (void)n_pts;
    auto collides = [&](double x, double y, bool live) {
        const float xf = (float)x, yf = (float)y;
        const f2 xx = f2{xf, xf}, yy = f2{yf, yf};
        f2 ax = xx - f2{rgx32, rpx32}, ay = yy - f2{rgy32, rpy32};
        f2 q = ax * ax + ay * ay;
        float m1 = fminf(q.x, q.y) - lor32;          // min over the tests of (square - lower threshold): < 0 -> collides for certain
        float m2 = fminf(q.x, q.y) - hir32;          // min over the tests of (square - upper threshold): > 0 -> free for certain
        for (int k = 0; k < n_g; k += 2) {
            const int k1 = k + 1 < n_g ? k + 1 : k;  // (an odd list: the last point twice)
            const f2 jx = f2{wv_readlane(Gx, k), wv_readlane(Gx, k1)}, jy = f2{wv_readlane(Gy, k), wv_readlane(Gy, k1)};
            const f2 lo = f2{wv_readlane(Gl, k), wv_readlane(Gl, k1)}, hi = f2{wv_readlane(Gh, k), wv_readlane(Gh, k1)};
            ax = xx - jx; ay = yy - jy;
            q = ax * ax + ay * ay;
            const f2 dl = q - lo, dh = q - hi;
            m1 = fminf(m1, fminf(dl.x, dl.y));
            m2 = fminf(m2, fminf(dh.x, dh.y));
        }
        for (int k = 0; k < n_p; k += 2) {
            const int k1 = k + 1 < n_p ? k + 1 : k;
            const f2 jx = f2{wv_readlane(Px, k), wv_readlane(Px, k1)}, jy = f2{wv_readlane(Py, k), wv_readlane(Py, k1)};
            const f2 lo = f2{wv_readlane(Pl, k), wv_readlane(Pl, k1)}, hi = f2{wv_readlane(Ph, k), wv_readlane(Ph, k1)};
            ax = xx - jx; ay = yy - jy;
            q = ax * ax + ay * ay;
            const f2 dl = q - lo, dh = q - hi;
            m1 = fminf(m1, fminf(dl.x, dl.y));
            m2 = fminf(m2, fminf(dh.x, dh.y));
        }
        const bool hit = m1 < 0.0f, open = m2 > 0.0f;
        if (__ballot(live && !hit && !open) != 0ull) { // a square between the moved thresholds: fp64 decides (rare; all lanes walk)
            const bool c64 = collides64(x, y, live);
            return (hit || open) ? hit : c64;
        }
        return hit;
    };
    // one pass: candidates of lanes 0 .. nb-1 read from block `blk` -- whole candidates from word `first` on, or (first < 0) the candidate that
    // straddles the block boundary as lane 0 (its nt words of the previous block in tl, the rest from the start of blk) and whole candidates
    // behind it; returns the lanes whose candidate is taken (free, or past the bound)
    auto eval_pass = [&](const uint32_t *blk, int first, int nt, const uint32_t *tl, int nb, int attempt0, double &x, double &y) -> uint64_t {
        const bool live = lane < nb;
        const int need = 6 - nt; // words of the new block that complete the straddling candidate
        uint32_t wd[6];
        if (first >= 0) {
            const uint32_t *w = blk + first + 6 * (live ? lane : 0);
#pragma unroll
            for (int k = 0; k < 6; ++k) wd[k] = w[k];
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                // lane 0: tail words, then words 0 .. need - 1 of the new block; lane a >= 1: words need + 6 (a - 1) + k
                const int idx = lane == 0 ? (k < nt ? 0 : k - nt) : need + 6 * (lane - 1) + k;
                const uint32_t v = blk[idx];
                wd[k] = (lane == 0 && k < nt) ? tl[k < 5 ? k : 4] : v;
            }
        }
        const uint32_t a0 = mt_temper(wd[0]) >> 5, b0 = mt_temper(wd[1]) >> 6, a1 = mt_temper(wd[2]) >> 5, b1 = mt_temper(wd[3]) >> 6,
                       a2 = mt_temper(wd[4]) >> 5, b2 = mt_temper(wd[5]) >> 6;
        const double u0 = ((double)a0 * 67108864.0 + (double)b0) / 9007199254740992.0;
        const double u1 = ((double)a1 * 67108864.0 + (double)b1) / 9007199254740992.0;
        const double u2 = ((double)a2 * 67108864.0 + (double)b2) / 9007199254740992.0;
        make(u0, u1, u2, x, y);
        const bool coll = collides(x, y, live);
        return __ballot(live && (!coll || attempt0 + lane >= max_att));
    };
    int attempt = 0; // number of the next candidate
    DBG_ADD(4, 1);
    for (;;) {
        if constexpr (W > 1) {
            if (attempt >= s.coop_after) {
                CoopLds<W> &Q = coop_lds<W>();
#ifdef CN_POST_DEBUG
                const long long dbg_t0 = wall_clock64();
                DBG_ADD(2, 1);
#endif
                // candidates 64 p .. 64 p + 63 of the round whose first candidate starts at word `off` of `rb`, the exact way
                auto eval_ring = [&](const uint32_t *rb, int off, int att0, int p, double &x, double &y) -> uint64_t {
                    const uint32_t *w = rb + off + 6 * (64 * p + lane);
                    uint32_t wd[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) wd[k] = w[k];
                    const uint32_t a0 = mt_temper(wd[0]) >> 5, b0 = mt_temper(wd[1]) >> 6, a1 = mt_temper(wd[2]) >> 5, b1 = mt_temper(wd[3]) >> 6,
                                   a2 = mt_temper(wd[4]) >> 5, b2 = mt_temper(wd[5]) >> 6;
                    const double u0 = ((double)a0 * 67108864.0 + (double)b0) / 9007199254740992.0;
                    const double u1 = ((double)a1 * 67108864.0 + (double)b1) / 9007199254740992.0;
                    const double u2 = ((double)a2 * 67108864.0 + (double)b2) / 9007199254740992.0;
                    make(u0, u1, u2, x, y);
                    const bool coll = collides(x, y, true);
                    return __ballot(!coll || att0 + 64 * p + lane >= max_att);
                };
                // the points for the coarse screen, two by two
                {
                    const float lo32c = (float)(dd_l * (1.0 - 1e-3));
                    const float Gc = pack(goal_mask, n_g, lo32c), Pc = pack(pos_mask, n_p, lo32c);
                    float *xy = reinterpret_cast<float *>(Q.pxy), *lo = reinterpret_cast<float *>(Q.plo);
                    auto put = [&](int slot, float x, float y, float l) {
                        xy[(slot >> 1) * 4 + (slot & 1)] = x; xy[(slot >> 1) * 4 + 2 + (slot & 1)] = y; lo[slot] = l;
                    };
                    const int n_s = 2 + n_g + n_p;
                    const float lorc = (float)(ddr * (1.0 - 1e-3));
                    if (lane == 0) { put(0, rgx32, rgy32, lorc); put(1, rpx32, rpy32, lorc); }
                    if (lane < n_g) put(2 + lane, Gx, Gy, Gc);
                    if (lane < n_p) put(2 + n_g + lane, Px, Py, Pc);
                    if (n_s & 1) { // (n_g + n_p is odd: the last point twice)
                        if (n_p > 0 ? lane == n_p - 1 : lane == n_g - 1) put(n_s, n_p > 0 ? Px : Gx, n_p > 0 ? Py : Gy, n_p > 0 ? Pc : Gc);
                    }
                    if (lane == 0) {
                        Q.kind = kind; Q.n_pairs = (n_s + 1) >> 1; Q.max_att = max_att; Q.circle_radius = (float)c.circle_radius; Q.vp = (float)vp;
                        Q.pos0 = R.pos; Q.attempt0 = attempt; Q.cmd = 1;
                    }
                }
                rng_sync();
                __syncthreads(); // the helpers wake up
                for (int k = threadIdx.x; k < MT_N; k += 64 * W) Q.buf[0][k] = R.mt[k]; // (all threads: block 0 = the staged block)
                __syncthreads();
                __syncthreads(); // the producer has made the first round's words
                constexpr int NG = CoopLds<W>::NG;
                int g0 = R.pos, bprev = 0;
                for (int round = 0;; ++round) {
                    const uint32_t *rb = Q.buf[round & 1];
                    const int off = g0 - MT_N * bprev;
                    double x, y;
                    uint64_t take = 0ull;
                    coop_screen_pass<W>(Q, rb, off, attempt, 0, lane, round, kind, (2 + n_g + n_p + 1) >> 1, max_att, (float)c.circle_radius, (float)vp);
                    __syncthreads(); // every wavefront's report is in (and the next round's words are made)
                    unsigned long long any = 0ull;
                    for (int p = 0; p < NG; ++p) any |= Q.take[round & 1][p];
                    if (any) {
                        int win = -1;
                        for (int p = 0; p < NG && win < 0; ++p) {
                            if (Q.take[round & 1][p] == 0ull) continue;
                            take = eval_ring(rb, off, attempt, p, x, y); // what the screen could not reject: exactly
                            if (take) win = p;
                        }
                        if (lane == 0) Q.verdict = win >= 0;
                        __syncthreads(); // the others learn whether the placement goes on
                        if (win >= 0) {
                            const int f = __ffsll((unsigned long long)take) - 1;
                            out_x = wv_readlane_d(x, f); out_y = wv_readlane_d(y, f);
                            // the staged state afterwards: the block that holds the last word read and the position behind that word
                            // (624 = "twist before the next draw", as the serial loop leaves it when a candidate ends a block)
                            const int g = g0 + 6 * (64 * win + f + 1);
                            const int bi = (g - 1) / MT_N;
                            rng_sync();
                            for (int k = lane; k < MT_N; k += 64) R.mt[k] = rb[(bi - bprev) * MT_N + k];
                            rng_sync();
                            R.pos = g - bi * MT_N;
#ifdef CN_POST_DEBUG
                            DBG_ADD(1, wall_clock64() - dbg_t0);
                            DBG_ADD(3, round + 1);
#endif
                            return;
                        }
                    }
                    bprev = (g0 + 384 * NG - 1) / MT_N;
                    g0 += 384 * NG;
                    attempt += 64 * NG;
                }
            }
        }
        const int left = MT_N - R.pos; // unread words of the current block
        int nb, first;                 // candidates of this pass; word index of lane 0's first word, or -1: lane 0 is the straddling candidate
        uint32_t tl[5] = {0u, 0u, 0u, 0u, 0u}; // the straddling candidate's words of the OLD block (raw, wave-uniform)
        int nt = 0;                                        // ... and how many there are
        if (left >= 6) {
            nb = left / 6 < 64 ? left / 6 : 64;
            first = R.pos;
        } else {
            // fewer than six words left: the next candidate straddles the end of the block (or starts the next one).  Its words of this
            // block are kept in registers, the block is regenerated, and the candidate is lane 0 of a pass whose other lanes take whole
            // candidates of the new block (as a pass of its own it cost a full walk for ONE candidate, every 104 candidates).
            nt = left;
            if (nt > 0) tl[0] = R.mt[R.pos];
            if (nt > 1) tl[1] = R.mt[R.pos + 1];
            if (nt > 2) tl[2] = R.mt[R.pos + 2];
            if (nt > 3) tl[3] = R.mt[R.pos + 3];
            if (nt > 4) tl[4] = R.mt[R.pos + 4];
            rng_twist(R, lane); // (R.pos = 0)
            nb = 64;            // 1 + (624 - 6) / 6 >= 64
            first = -1;
        }
        const int need = 6 - nt;
        double x, y;
        DBG_ADD(5, 1);
        const uint64_t take = eval_pass(R.mt, first, nt, tl, nb, attempt, x, y);
        // stream position behind candidate f of this pass
        if (take) {
            const int f = __ffsll((unsigned long long)take) - 1;
            out_x = wv_readlane_d(x, f); out_y = wv_readlane_d(y, f);
            R.pos = first >= 0 ? first + 6 * (f + 1) : need + 6 * f;
            return;
        }
        R.pos = first >= 0 ? first + 6 * nb : need + 6 * (nb - 1);
        attempt += nb;
    }
}

// the other wavefronts of an env: parked at the barrier until the master publishes a placement (place_by_rejection<W>), then round by
// round.  Waves 1 .. W - 2 (helpers) put candidates 64 wave .. 64 wave + 63 through the coarse screen and report what it cannot reject (the
// master decides those); wave W - 1 (producer) makes the next round's words meanwhile.
template <int W>
__device__ __forceinline__ void coop_helper_loop(int lane, int wave)
{
    constexpr int NG = CoopLds<W>::NG;
    CoopLds<W> &Q = coop_lds<W>();
    for (;;) {
        __syncthreads(); // a placement, or the end
        if (Q.cmd == 2) return;
        const int kind = Q.kind, n_pairs = Q.n_pairs, max_att = Q.max_att;
        const float circle_radius = Q.circle_radius, vp = Q.vp;
        int g0 = Q.pos0, attempt = Q.attempt0;
        for (int k = threadIdx.x; k < MT_N; k += 64 * W) Q.buf[0][k] = g_mt_lds[k];
        __syncthreads();
        if (wave == W - 1) { // ---- producer
            int bprev = 0, b1 = (g0 + 384 * NG - 1) / MT_N; // round 0 reads blocks 0 .. b1
            coop_produce(Q.buf[0], nullptr, lane, MT_N * b1);
            __syncthreads(); // the first round's words are made
            for (int round = 0;; ++round) {
                const int b2 = (g0 + 768 * NG - 1) / MT_N; // round + 1 reads blocks b1 .. b2
#ifdef CN_POST_DEBUG
                const long long dbg_p0 = wall_clock64();
#endif
                coop_produce(Q.buf[(round + 1) & 1], Q.buf[round & 1] + MT_N * (b1 - bprev), lane, MT_N * (b2 - b1));
#ifdef CN_POST_DEBUG
                DBG_ADD(7, wall_clock64() - dbg_p0);
#endif
                __syncthreads();
                unsigned long long any = 0ull;
                for (int p = 0; p < NG; ++p) any |= Q.take[round & 1][p];
                if (any) {
                    __syncthreads();
                    if (Q.verdict) break;
                }
                g0 += 384 * NG;
                bprev = b1; b1 = b2;
            }
            continue;
        }
        __syncthreads(); // the first round's words are made
        int bprev = 0;
        for (int round = 0;; ++round) {
#ifdef CN_POST_DEBUG
            const long long dbg_e0 = wall_clock64();
#endif
            coop_screen_pass<W>(Q, Q.buf[round & 1], g0 - MT_N * bprev, attempt, wave, lane, round, kind, n_pairs, max_att, circle_radius, vp);
#ifdef CN_POST_DEBUG
            if (wave == 1) DBG_ADD(6, wall_clock64() - dbg_e0);
#endif
            __syncthreads(); // every wavefront's report is in (and the next round's words are made)
            unsigned long long any = 0ull;
            for (int p = 0; p < NG; ++p) any |= Q.take[round & 1][p];
            if (any) {
                __syncthreads(); // the master has looked at the reports
                if (Q.verdict) break;
            }
            bprev = (g0 + 384 * NG - 1) / MT_N;
            g0 += 384 * NG;
            attempt += 64 * NG;
        }
    }
}

// crowd_sim_var_num.py:116-146 generate_circle_crossing_human (+ Agent.__init__/sample_random_attributes draws).
// All lanes compute the candidate position identically; the min-distance test against the existing agents is
// lane-parallel.  n_existing = number of humans currently in self.humans (slot itself included on respawn, :455).
template <int W = 1>
__device__ __forceinline__ void gen_human(const EnvDev &s, Rng &R, int lane, int slot, int n_existing, const Robot &rb, Lane &h, double &shared_nd)
{
    const cn_env_config &c = s.cfg;
    double radius = c.human_radius, vpref = c.human_v_pref;
    if (c.randomize_attributes) {
        shared_nd = rng_uniform(R, lane, 5.0, 10.0); // agent.py:21-22
        vpref = rng_uniform(R, lane, 0.5, 1.5);      // agent.py:49
        radius = rng_uniform(R, lane, 0.3, 0.5);     // agent.py:50
    }
    double px, py;
    // (unbounded in the reference: see CN_MAX_PLACEMENT_ATTEMPTS)
    // :133-136: a unicycle robot keeps new humans half a circle radius away from its start and goal
    const double md_r = c.kinematics == CN_KIN_UNICYCLE ? c.circle_radius / 2.0 : radius + c.robot_radius + c.discomfort_dist;
    place_by_rejection<W>(s, R, lane, 0, radius, 0.0, md_r, n_existing, -1, rb, h, px, py);
    if (lane == slot) {
        h.px = px; h.py = py; h.gx = -px; h.gy = -py; h.vx = 0.0; h.vy = 0.0; h.rad = radius; h.vpref = vpref;
        h.simv = 0; // new Human -> new ORCA object, sim rebuilt on next use
    }
}

// crowd_sim.py:415-450 update_human_goals_randomly (every human, goal_change_chance) and :453-485 update_human_goal (one human,
// end_goal_change_chance: `only` >= 0 selects it)
template <int W = 1>
__device__ __forceinline__ void change_goals(const EnvDev &s, Rng &R, int lane, int n, const Robot &rb, Lane &h, int only = -1)
{
    const cn_env_config &c = s.cfg;
    const int H = n; // the humans present
    for (int i = only >= 0 ? only : 0; i < (only >= 0 ? only + 1 : H); ++i) {
        double vp_i = __shfl(h.vpref, i, 64);
        const double rad_i = __shfl(h.rad, i, 64);
        if (only < 0 && vp_i == 0.0) continue;
        if (vp_i == 0.0) vp_i = 1.0;
        if (rng_double(R, lane) <= (only >= 0 ? c.end_goal_change_chance : c.goal_change_chance)) {
            double gx, gy;
            place_by_rejection<W>(s, R, lane, 1, rad_i, vp_i, rad_i + c.robot_radius + c.discomfort_dist, H, i, rb, h, gx, gy);
            if (lane == i) { h.gx = gx; h.gy = gy; }
        }
    }
}

// crowd_sim_var_num.py:233-279 generate_ob / crowd_sim_pred.py:62-97 / crowd_sim_pred_real_gst.py:76-93,
// crowd_sim.py:558-572 get_num_human_in_fov, :243-273 update_last_human_states.
__device__ __forceinline__ void write_obs(const EnvDev &s, int e, int lane, int n, bool reset, const Robot &rb, Lane &h, const cn_obs &ob, int step_counter)
{
    const cn_env_config &c = s.cfg;
    const int H = s.H, D = s.D, P = s.P; // H observation rows (crowd_sim_var_num.py:249, crowd_sim_pred.py:78), n humans present
    const bool isH = lane < n, isRow = lane < H;
    // detect_visible(robot, human, robot1=True), crowd_sim.py:513-552: inside the robot's field of view (FOV = 2*pi: iff not coincident) and
    // within sensor range
    const double dx = rb.px - h.px, dy = rb.py - h.py;
    bool vis = isH && !(dx == 0.0 && dy == 0.0) && (norm2(dx, dy) - c.robot_radius - h.rad <= c.sensor_range);
    if (c.robot_fov < 2.0) vis = vis && in_fov(c, c.robot_fov, rb.px, rb.py, rb.vx, rb.vy, rb.theta, h.px, h.py);
    const uint64_t vmask = __ballot(vis);
    const int num_visible = __popcll(vmask);
    if (s.vis && isRow) s.vis[(size_t)e * H + lane] = vis ? 1 : 0; // human_visibility, read by the next step's 'truth' blanking
    if (s.nh && c.env_kind != CN_ENV_PRED && lane == 0) {
        // observed_human_ids (crowd_sim_var_num.py:275): who may not leave at the next crowd-size change.  CrowdSimPred's own
        // generate_ob never refreshes the list (it stays [] from reset)
        s.obs_cnt[e] = num_visible;
        s.obs_max[e] = vmask ? 63 - __clzll((long long)vmask) : -1;
    }
    const double prev_vx = h.l2, prev_vy = h.l3;
    if (vis) { h.l0 = h.px; h.l1 = h.py; h.l2 = h.vx; h.l3 = h.vy; h.l4 = h.rad; }
    else if (isH && reset) { h.l0 = 15.0; h.l1 = 15.0; h.l2 = 0.0; h.l3 = 0.0; h.l4 = 0.3; }
    else if (isH) { h.l0 = h.l0 + h.l2 * c.time_step; h.l1 = h.l1 + h.l3 * c.time_step; }
    if (lane == 0) {
        float *rn = ob.robot_node + (size_t)e * 7;
        rn[0] = (float)rb.px; rn[1] = (float)rb.py; rn[2] = (float)c.robot_radius; rn[3] = (float)rb.gx; rn[4] = (float)rb.gy;
        rn[5] = (float)c.robot_v_pref; rn[6] = (float)rb.theta;
        ob.temporal_edges[(size_t)e * 2] = (float)rb.vx; ob.temporal_edges[(size_t)e * 2 + 1] = (float)rb.vy;
        ob.detected_human_num[e] = (float)(num_visible == 0 ? 1 : num_visible);
    }
    if (c.env_kind == CN_ENV_COLLECT) {
        // crowd_sim_var_num_collect.py:100-133: humans that were visible at the last observation and are not now get fresh prediction
        // ids (ascending, in list order); row i = (frame, id, ABSOLUTE believed position) if visible, (frame, id, inf, inf) otherwise
        const bool was = isRow && s.last_obs[(size_t)e * H + lane] != 0;
        const bool out = isH && was && !vis;
        const uint64_t omask = __ballot(out);
        const int base = s.max_pid[e];
        int pid = isRow ? s.pred_id[(size_t)e * H + lane] : 0;
        if (out) pid = base + __popcll(omask & ((1ull << lane) - 1ull));
        if (isRow) {
            s.pred_id[(size_t)e * H + lane] = pid;
            s.last_obs[(size_t)e * H + lane] = vis ? 1 : 0;
            float *se = ob.spatial_edges + ((size_t)e * H + lane) * 4;
            se[0] = (float)(((double)step_counter * c.time_step) / c.time_step); // global_time / data.pred_timestep (== env.time_step)
            se[1] = (float)pid;
            se[2] = vis ? (float)h.l0 : INFINITY;
            se[3] = vis ? (float)h.l1 : INFINITY;
            if (ob.visible_masks) ob.visible_masks[(size_t)e * H + lane] = vis ? 1 : 0;
        }
        if (lane == 0 && omask) s.max_pid[e] = base + __popcll(omask);
        return;
    }
    const double ex = h.l0 - rb.px, ey = h.l1 - rb.py; // == true relative position for visible humans
    const bool do_sort = c.sort_humans && c.env_kind != CN_ENV_PRED_GST;
    int row = lane;
    if (do_sort) {
        // sorted(key = norm(first two)) is stable, invisible rows (inf) keep index order and go last
        const double key = vis ? sqrt(ex * ex + ey * ey) : INFINITY;
        int rank = 0;
        for (int m = 0; m < H; ++m) {
            const double km = __shfl(key, m, 64);
            rank += (km < key || (km == key && m < lane)) ? 1 : 0;
        }
        row = rank;
    }
    if (isRow) {
        float *se = ob.spatial_edges + ((size_t)e * H + row) * D;
        if (c.env_kind == CN_ENV_VARNUM) {
            se[0] = vis ? (float)ex : 15.0f;
            se[1] = vis ? (float)ey : 15.0f;
        } else {
            double *ft = s.ftraj ? s.ftraj + (size_t)e * P * 2 * H : nullptr;
            const double *tre = c.predict_truth ? s.tr + (size_t)e * (s.R + 1) * 4 * H : nullptr;
            for (int k = 0; k <= P; ++k) {
                double fx = 15.0, fy = 15.0;
                if (vis && tre && k >= 1) {
                    // sim.predict_method = 'truth' (crowd_sim_pred.py:81 -> crowd_sim_var_num.py:180-206): the humans' own ORCA rolled
                    // forward from the state just reached, computed by orca_truth_kernel between the two halves of the step
                    fx = tre[(k * s.I * 4 + 0) * H + lane]; // human_future_traj[::pred_interval] (crowd_sim_var_num.py:206)
                    fy = tre[(k * s.I * 4 + 1) * H + lane];
                } else if (vis) {
                    const double t = (double)k * c.time_step * (double)s.I; // arange(P + 1) * time_step * pred_interval (crowd_sim_var_num.py:212)
                    fx = h.px + t * prev_vx;
                    fy = h.py + t * prev_vy;
                }
                if (ft && k >= 1) { ft[((k - 1) * 2 + 0) * H + lane] = fx; ft[((k - 1) * 2 + 1) * H + lane] = fy; }
                if (c.env_kind == CN_ENV_PRED) {
                    se[2 * k] = vis ? (float)(fx - rb.px) : 15.0f;
                    se[2 * k + 1] = vis ? (float)(fy - rb.py) : 15.0f;
                } else {
                    se[2 * k] = vis ? (float)ex : 15.0f;
                    se[2 * k + 1] = vis ? (float)ey : 15.0f;
                }
            }
        }
        if (ob.visible_masks) {
            uint8_t *vm = ob.visible_masks + (size_t)e * H;
            if (do_sort) vm[lane] = lane < num_visible ? 1 : 0;
            else vm[lane] = vis ? 1 : 0;
        }
    }
}

// crowd_sim_var_num.py:303-363 reset (seed, robot, humans, potential, first observation)
// the RNG-consuming part of reset(): seed, robot, humans (crowd_sim_var_num.py:333-340, :64-146)
__device__ __forceinline__ void gen_episode_head(const EnvDev &s, Rng &R, int e, int lane, Robot &rb, int &n)
{
    const cn_env_config &c = s.cfg;
    const uint64_t offset = c.phase == CN_PHASE_TRAIN ? 2000ull : (c.phase == CN_PHASE_VAL ? 0ull : 1000ull);
    const uint64_t seed = offset + s.case_counter[e] + (uint64_t)(s.seed_base + e);
    rng_seed(R, (uint32_t)seed, lane);
    double px, py, gx, gy;
    if (c.kinematics == CN_KIN_UNICYCLE) {
        // generate_robot_humans, sim2real branch :78-91: start on the arena circle, goal >= 4 m away, random heading,
        // 1 .. human_num + human_num_range humans
        const double angle = rng_uniform(R, lane, 0.0, M_PI * 2.0);
        double sn, cs;
        det_sincos(angle, sn, cs);
        px = c.arena_size * cs; py = c.arena_size * sn;
        for (;;) {
            gx = rng_uniform(R, lane, -c.arena_size, c.arena_size);
            gy = rng_uniform(R, lane, -c.arena_size, c.arena_size);
            if (norm2(px - gx, py - gy) >= 4.0) break;
        }
        rb.theta = rng_uniform(R, lane, 0.0, 2.0 * M_PI);
        n = rng_randint(R, lane, 1, c.human_num + c.human_num_range + 1);
    } else {
        for (;;) { // :97-100
            px = rng_uniform(R, lane, -c.arena_size, c.arena_size);
            py = rng_uniform(R, lane, -c.arena_size, c.arena_size);
            gx = rng_uniform(R, lane, -c.arena_size, c.arena_size);
            gy = rng_uniform(R, lane, -c.arena_size, c.arena_size);
            if (norm2(px - gx, py - gy) >= 8.0) break;
        }
        rb.theta = M_PI / 2.0;
        // :103-104 randint(human_num - range, human_num + range + 1): consumes no draw when human_num_range == 0
        n = rng_randint(R, lane, c.human_num - c.human_num_range, c.human_num + c.human_num_range + 1);
    }
    rb.px = px; rb.py = py; rb.gx = gx; rb.gy = gy; rb.vx = 0.0; rb.vy = 0.0;
}
__device__ __forceinline__ void gen_episode(const EnvDev &s, Rng &R, int e, int lane, Robot &rb, Lane &h, double &shared_nd, int &n)
{
    gen_episode_head(s, R, e, lane, rb, n);
    for (int i = 0; i < n; ++i) gen_human(s, R, lane, i, i, rb, h, shared_nd);
    rb.pot = -fabs(norm2(rb.gx - rb.px, rb.gy - rb.py));
}

// the rest of reset(): belief cleared (:108), case counter advanced (:348), episode statistics, first observation
__device__ __forceinline__ void finish_reset(const EnvDev &s, int e, int lane, int n, Robot &rb, Lane &h, const cn_obs &ob, bool with_obs = true)
{
    const cn_env_config &c = s.cfg;
    h.l0 = h.l1 = h.l2 = h.l3 = h.l4 = 0.0;
    const uint64_t case_size = c.phase == CN_PHASE_TRAIN ? (4294967295ull - 2000ull) : (c.phase == CN_PHASE_VAL ? c.val_size : c.test_size);
    if (lane == 0) {
        s.case_counter[e] = (s.case_counter[e] + (uint64_t)c.nenv) % case_size;
        s.step_counter[e] = 0; s.ep_ret[e] = 0.0; s.ep_cnt[e] = 0;
        if (s.nh) { s.obs_cnt[e] = 0; s.obs_max[e] = -1; } // :327 observed_human_ids = []
        if (s.max_pid) s.max_pid[e] = n; // crowd_sim_var_num_collect.py:79-81
        if (s.wheel) { s.wheel[(size_t)e * 4 + 2] = 0.0; s.wheel[(size_t)e * 4 + 3] = 0.0; } // np.random.seed -> _legacy_seeding: has_gauss = 0
    }
    if (s.pred_id && lane < s.H) { s.pred_id[(size_t)e * s.H + lane] = lane; s.last_obs[(size_t)e * s.H + lane] = 0; }
    if (with_obs) write_obs(s, e, lane, n, true, rb, h, ob, 0);
}

// crowd_sim_var_num.py:303-363 reset.  Uses the pre-generated episode when the side stream has one ready.
__device__ __forceinline__ void do_reset(const EnvDev &s, Rng &R, int e, int lane, Robot &rb, Lane &h, double &shared_nd, int &n, const cn_obs &ob,
                                         bool with_obs = true)
{
    if (s.nx_ready[e]) {
        n = s.nx_nh ? s.nx_nh[e] : s.H;
        const int H = s.H;
        const int lj = lane < H ? lane : 0;
        const double *hum = s.nx_hum + (size_t)e * 8 * H;
        h.px = hum[F_PX * H + lj]; h.py = hum[F_PY * H + lj]; h.vx = 0.0; h.vy = 0.0;
        h.gx = hum[F_GX * H + lj]; h.gy = hum[F_GY * H + lj]; h.rad = hum[F_RAD * H + lj]; h.vpref = hum[F_VPREF * H + lj];
        h.simv = 0;
        const double *r = s.nx_rob + (size_t)e * 8;
        rb.px = r[R_PX]; rb.py = r[R_PY]; rb.vx = 0.0; rb.vy = 0.0; rb.gx = r[R_GX]; rb.gy = r[R_GY]; rb.theta = r[R_THETA]; rb.pot = r[R_POT];
        shared_nd = s.nx_shared_nd[e];
        rng_sync();
        for (int k = lane; k < MT_N; k += 64) R.mt[k] = s.nx_mt[(size_t)e * MT_N + k];
        R.pos = s.nx_mt_pos[e];
        R.loaded = true;
        rng_sync();
        if (lane == 0) s.nx_ready[e] = 0;
    } else {
        gen_episode(s, R, e, lane, rb, h, shared_nd, n);
    }
    finish_reset(s, e, lane, n, rb, h, ob, with_obs);
}

__device__ __forceinline__ void load_env(const EnvDev &s, int e, int lane, Robot &rb, Lane &h)
{
    const int H = s.H;
    const int lj = lane < H ? lane : 0;
    const double *hum = s.hum + (size_t)e * 8 * H;
    h.px = hum[F_PX * H + lj]; h.py = hum[F_PY * H + lj]; h.vx = hum[F_VX * H + lj]; h.vy = hum[F_VY * H + lj];
    h.gx = hum[F_GX * H + lj]; h.gy = hum[F_GY * H + lj]; h.rad = hum[F_RAD * H + lj]; h.vpref = hum[F_VPREF * H + lj];
    const double *l = s.lhs + (size_t)e * 5 * H;
    h.l0 = l[lj]; h.l1 = l[H + lj]; h.l2 = l[2 * H + lj]; h.l3 = l[3 * H + lj]; h.l4 = l[4 * H + lj];
    h.simv = s.sim_valid[(size_t)e * H + lj];
    const double *r = s.rob + (size_t)e * 8;
    rb.px = r[R_PX]; rb.py = r[R_PY]; rb.vx = r[R_VX]; rb.vy = r[R_VY]; rb.gx = r[R_GX]; rb.gy = r[R_GY]; rb.theta = r[R_THETA]; rb.pot = r[R_POT];
}
__device__ __forceinline__ void store_env(const EnvDev &s, int e, int lane, const Robot &rb, const Lane &h)
{
    const int H = s.H;
    if (lane < H) {
        double *hum = s.hum + (size_t)e * 8 * H;
        hum[F_PX * H + lane] = h.px; hum[F_PY * H + lane] = h.py; hum[F_VX * H + lane] = h.vx; hum[F_VY * H + lane] = h.vy;
        hum[F_GX * H + lane] = h.gx; hum[F_GY * H + lane] = h.gy; hum[F_RAD * H + lane] = h.rad; hum[F_VPREF * H + lane] = h.vpref;
        double *l = s.lhs + (size_t)e * 5 * H;
        l[lane] = h.l0; l[H + lane] = h.l1; l[2 * H + lane] = h.l2; l[3 * H + lane] = h.l3; l[4 * H + lane] = h.l4;
        s.sim_valid[(size_t)e * H + lane] = h.simv;
    }
    if (lane == 0) {
        double *r = s.rob + (size_t)e * 8;
        r[R_PX] = rb.px; r[R_PY] = rb.py; r[R_VX] = rb.vx; r[R_VY] = rb.vy; r[R_GX] = rb.gx; r[R_GY] = rb.gy; r[R_THETA] = rb.theta; r[R_POT] = rb.pot;
    }
}

__global__ __launch_bounds__(64) void env_reset_kernel(EnvDev s, cn_obs ob, int with_obs)
{
    const int lane = threadIdx.x;
    const int e = blockIdx.x;
    Rng R{MT_N, false};
    Robot rb{};
    Lane h{};
    h.rad = s.cfg.human_radius;
    double shared_nd = s.cfg.orca_neighbor_dist;
    int n = s.H;
    if (e == 0 && lane == 0) *s.lp3_cnt = 0; // the ORCA pass that follows starts with an empty linearProgram3 list
    do_reset(s, R, e, lane, rb, h, shared_nd, n, ob, with_obs != 0);
    if (!with_obs && lane == 0) s.pend[e] = 1;
    store_env(s, e, lane, rb, h);
    if (lane == 0) { s.shared_nd[e] = shared_nd; if (s.nh) s.nh[e] = n; }
    rng_store(R, s, e, lane);
}

// Generates the NEXT episode of every env whose staging slot is empty (side stream, overlapped with the policy forward).
// The wavefronts of this kernel keep registers on their CUs, and the policy's human-human kernel (next on the caller's stream) needs
// every register of a CU to place a workgroup there: an env whose rejection sampling runs long (the tail reaches 150 us) used to hold
// one CU back for that long and with it the whole launch.  So the work is BUDGETED: a wavefront that has not finished after
// `budget` ticks of the 100 MHz clock saves where it is -- the staging arrays hold exactly the state between two humans -- and the
// next launch resumes there.  The episode is the same whichever way it is cut; an env that resets before its staging is complete
// generates in place, as it always could, and the stale staging is restarted (nx_case).
// (the body: one wavefront, one env; R.mt = that wavefront's 624-word LDS slice)
template <int W = 1>
__device__ __forceinline__ void pregen_env(const EnvDev &s, int e, int lane, long long budget, Rng &R)
{
    if (s.nx_ready[e]) return;
    const long long t0 = wall_clock64();
    const int H = s.H;
    int prog = s.nx_prog[e];
    if (prog > 0 && s.nx_case[e] != s.case_counter[e]) prog = 0;
    Robot rb{};
    Lane h{};
    h.rad = s.cfg.human_radius;
    double shared_nd = s.shared_nd[e]; // overwritten by the first Human() when randomised, unused otherwise
    int n = H;
    if (prog == 0) {
        gen_episode_head(s, R, e, lane, rb, n);
        if (lane == 0) s.nx_case[e] = s.case_counter[e];
        prog = 1;
    } else {
        const int lj = lane < H ? lane : 0;
        const double *hum = s.nx_hum + (size_t)e * 8 * H;
        h.px = hum[F_PX * H + lj]; h.py = hum[F_PY * H + lj]; h.gx = hum[F_GX * H + lj]; h.gy = hum[F_GY * H + lj];
        h.rad = hum[F_RAD * H + lj]; h.vpref = hum[F_VPREF * H + lj];
        const double *r = s.nx_rob + (size_t)e * 8;
        rb.px = r[R_PX]; rb.py = r[R_PY]; rb.gx = r[R_GX]; rb.gy = r[R_GY]; rb.theta = r[R_THETA];
        shared_nd = s.nx_shared_nd[e];
        n = s.nx_nh ? s.nx_nh[e] : H;
        rng_sync();
        for (int k = lane; k < MT_N; k += 64) R.mt[k] = s.nx_mt[(size_t)e * MT_N + k];
        R.pos = s.nx_mt_pos[e];
        R.loaded = true;
        rng_sync();
    }
    bool complete = true;
    for (int i = prog - 1; i < n; ++i) {
        gen_human<W>(s, R, lane, i, i, rb, h, shared_nd);
        if (i + 1 < n && __builtin_amdgcn_readfirstlane((int)(wall_clock64() - t0 > budget))) { prog = i + 2; complete = false; break; }
    }
    if (complete) rb.pot = -fabs(norm2(rb.gx - rb.px, rb.gy - rb.py));
    if (lane < H) {
        double *hum = s.nx_hum + (size_t)e * 8 * H;
        hum[F_PX * H + lane] = h.px; hum[F_PY * H + lane] = h.py; hum[F_GX * H + lane] = h.gx; hum[F_GY * H + lane] = h.gy;
        hum[F_RAD * H + lane] = h.rad; hum[F_VPREF * H + lane] = h.vpref;
    }
    if (lane == 0) {
        double *r = s.nx_rob + (size_t)e * 8;
        r[R_PX] = rb.px; r[R_PY] = rb.py; r[R_GX] = rb.gx; r[R_GY] = rb.gy; r[R_THETA] = rb.theta; r[R_POT] = rb.pot;
        s.nx_shared_nd[e] = shared_nd;
        s.nx_mt_pos[e] = R.pos;
        if (s.nx_nh) s.nx_nh[e] = n;
        s.nx_prog[e] = complete ? 0 : prog;
    }
    rng_sync();
    for (int k = lane; k < MT_N; k += 64) s.nx_mt[(size_t)e * MT_N + k] = R.mt[k];
    __threadfence(); // the staging is complete before the flag says so (the flag's readers run in later launches; belt and braces)
    if (lane == 0 && complete) s.nx_ready[e] = 1;
}

// (W > 1: with the helper wavefronts of place_by_rejection<W>, as in env_step_kernel; not launched -- see launch_pregen)
template <int W = 1>
__global__ __launch_bounds__(64 * W) void env_pregen_kernel(EnvDev s, long long budget)
{
    const CnStampScope stamp_scope(s.stamp);
    const int lane = threadIdx.x & 63;
    if constexpr (W > 1) {
        if (threadIdx.x >= 64) { coop_helper_loop<W>(lane, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))); return; }
    }
    Rng R{MT_N, false};
    pregen_env<W>(s, blockIdx.x, lane, budget, R);
    if constexpr (W > 1) { // release the helpers
        if (lane == 0) coop_lds<W>().cmd = 2;
        __syncthreads();
    }
}

// (Round 5 measured this generator INSIDE the ORCA tail's launch, i.e. behind the human-human kernel instead of beside the lane kernel, so
// that it no longer holds ~60 CUs when that kernel starts: the kernel got 17 us shorter (all its workgroups start within 8 us) and the
// step 4 % LONGER -- its workgroups then end together, and the 20 us in which the tail used to run on the CUs of the early finishers are
// gone; profiles/HISTORY.md section 10.)
__global__ __launch_bounds__(256) void orca_lp3_kernel(EnvDev s)
{
    const CnStampScope stamp_scope(s.stamp);
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int total = *s.lp3_cnt;
    const int pairs = (total + 1) >> 1;
    const int H = s.H;
    for (int p = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); p < pairs; p += gridDim.x * 4) {
        const bool act = 2 * p + half < total;
        const int k = act ? 2 * p + half : 2 * p; // (an odd list: the upper half of the last wavefront idles on a copy of the lower one's data)
        const Lp3Hdr hd = s.lp3_hdr[k];
        const float4 ln = hl < hd.nn ? s.lp3_lines[(size_t)k * 32 + hl] : make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        LpLine L;
        L.px = ln.x; L.py = ln.y; L.dx = ln.z; L.dy = ln.w;
        float rx = hd.rx, ry = hd.ry;
        lp3_pair(L, hd.nn, hd.line_fail, hd.radius, act, lane, rx, ry);
        if (act && hl == 0) {
            const int e = hd.agent / H, i = hd.agent - e * H;
            s.hact[(size_t)e * 2 * H + i] = rx;
            s.hact[(size_t)e * 2 * H + H + i] = ry;
        }
    }
}

// crowd_sim_var_num.py:366-460 step (+ crowd_sim_pred.py:216-233 social reward) and the vec-env auto-reset
// (rl/networks/shmem_vec_env.py:139-142).  ORCA velocities for this step were produced by orca_kernel.
// goal changes every 5 s and respawns of the humans that reached their goal (crowd_sim_var_num.py:446-456): after the observation
template <int W = 1>
__device__ __forceinline__ void post_obs_updates(const EnvDev &s, Rng &R, int e, int lane, int n, int step_counter, const Robot &rb, Lane &h, double &shared_nd)
{
    const cn_env_config &c = s.cfg;
    const int H = n; // the humans present
    const bool isH = lane < H;
    const int period = (int)(5.0 / c.time_step + 0.5);
    if (c.random_goal_changing && (step_counter % period) == 0) {
        rng_load(R, s, e, lane);
        change_goals<W>(s, R, lane, n, rb, h);
    }
    if (c.end_goal_changing) {
        uint64_t reached = __ballot(isH && norm2(h.gx - h.px, h.gy - h.py) < h.rad);
        if (reached) rng_load(R, s, e, lane);
        while (reached) {
            const int i = __ffsll((unsigned long long)reached) - 1;
            reached &= reached - 1;
            // :451-456 respawned (holonomic robot) or given a new goal (unicycle robot)
            // (crowd_sim_pred.py:208-212 always respawns)
            if (c.kinematics == CN_KIN_UNICYCLE && c.env_kind == CN_ENV_VARNUM) change_goals<W>(s, R, lane, n, rb, h, i);
            else gen_human<W>(s, R, lane, i, H, rb, h, shared_nd);
        }
    }
}

// The deferred post-observation updates of env_step_kernel<false, 1, true>: one workgroup of W wavefronts per listed env (the list is short:
// an env changes goals every 5 s, so ~1/20 of a dephased batch, plus the envs where a human reached its goal), the placement loops on all
// W of them (place_by_rejection<W>).  Same state in, same state out as the in-kernel call: load_env / store_env are exact.
template <int W>
__global__ __launch_bounds__(64 * W) void env_post_kernel(EnvDev s)
{
    if ((int)blockIdx.x >= *s.post_cnt) return;
    const CnStampScope stamp_scope(s.stamp);
    const int lane = threadIdx.x & 63;
    if constexpr (W > 1) {
        if (threadIdx.x >= 64) { coop_helper_loop<W>(lane, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))); return; }
    }
    const int e = s.post_list[blockIdx.x];
#ifdef CN_POST_DEBUG
    const long long dbg_start = wall_clock64();
    if (lane < 8) g_dbg_blk[lane] = 0;
#endif
    Rng R{MT_N, false};
    Robot rb;
    Lane h;
    load_env(s, e, lane, rb, h);
    double shared_nd = s.shared_nd[e];
    const int n = crowd_size(s, e);
    post_obs_updates<W>(s, R, e, lane, n, s.step_counter[e], rb, h, shared_nd);
#ifdef CN_POST_DEBUG
    if (lane == 0) g_dbg_blk[0] = wall_clock64() - dbg_start;
    if (lane < 8) g_post_dbg[blockIdx.x * 8 + lane] = g_dbg_blk[lane];
#endif
    store_env(s, e, lane, rb, h);
    if (lane == 0) s.shared_nd[e] = shared_nd;
    rng_store(R, s, e, lane);
    if constexpr (W > 1) { // release the helpers
        if (lane == 0) coop_lds<W>().cmd = 2;
        __syncthreads();
    }
}

// Second half of a step when the observation needs the 'truth' roll-outs of the state just reached (sim.predict_method = 'truth'):
// observation (reset or step form), then the post-observation updates of the envs that were not reset.
__global__ __launch_bounds__(64) void env_obs_kernel(EnvDev s, cn_obs ob)
{
    const int lane = threadIdx.x;
    const int e = blockIdx.x;
    Rng R{MT_N, false};
    Robot rb;
    Lane h;
    load_env(s, e, lane, rb, h);
    double shared_nd = s.shared_nd[e];
    const bool was_reset = s.pend[e] != 0;
    const int n = crowd_size(s, e);
    write_obs(s, e, lane, n, was_reset, rb, h, ob, was_reset ? 0 : s.step_counter[e]);
    if (!was_reset) post_obs_updates(s, R, e, lane, n, s.step_counter[e], rb, h, shared_nd);
    if (lane == 0) s.pend[e] = 0;
    store_env(s, e, lane, rb, h);
    if (lane == 0) s.shared_nd[e] = shared_nd;
    rng_store(R, s, e, lane);
}

// SPLIT = true: first half only (everything up to the kinematics and the reset bookkeeping); env_obs_kernel finishes the step after
// the roll-out kernels.
// W = 4: three helper wavefronts per env for the long placement loops of dense crowds (see CoopLds); W = 1: one wavefront per env
// DEFER = true (dense crowds without a lane kernel): the observation is written, the post-observation updates -- which nothing in the
// observation depends on -- are left to env_post_kernel on the side stream, in front of the ORCA pass that needs the new goals: the
// long placement loops of the few envs that change goals then run beside the policy forward instead of in front of it.
template <bool SPLIT, int W = 1, bool DEFER = false>
__global__ __launch_bounds__(64 * W, W > 1 ? 2 : 4) void env_step_kernel(EnvDev s, const float *actions, cn_obs ob, float *reward_out,
                                                      uint8_t *done_out, uint8_t *info_out, double *ep_ret_out, int32_t *ep_len_out, float *not_done_out)
{
    const CnStampScope stamp_scope(s.stamp);
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x;
    if constexpr (W > 1) {
        if (threadIdx.x >= 64) { coop_helper_loop<W>(lane, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))); return; }
    }
    const cn_env_config &c = s.cfg;
    const int H = s.H;
    int n = crowd_size(s, e);  // humans present during this step's reward / kinematics
    const bool isH = lane < n;
    Rng R{MT_N, false};
    Robot rb;
    Lane h;
    load_env(s, e, lane, rb, h);
    double shared_nd = s.shared_nd[e];
    int step_counter = s.step_counter[e];
    if (e == 0 && lane == 0) *s.lp3_cnt = 0; // the ORCA pass that follows starts with an empty linearProgram3 list

    // srnn.clip_action (crowd_nav/policy/srnn.py:17-34), float32 like the numpy action array
    float ax = actions[2 * e], ay = actions[2 * e + 1];
    double uni_v = 0.0, uni_r = 0.0; // ActionRot(v, r) of the unicycle robot
    double axd = 0.0, ayd = 0.0;     // float64 action of the social-force robot
    if (c.robot_policy == CN_ROBOT_SOCIAL_FORCE) {
        // SOCIAL_FORCE.predict (crowd_nav/policy/social_force.py:11-52) on the robot's beliefs, all in float64; lane j evaluates the
        // push of human j, the sum runs in list order
        const double dxg = rb.gx - rb.px, dyg = rb.gy - rb.py;
        const double dist_to_goal = sqrt(dxg * dxg + dyg * dyg);
        const double desired_vx = (dxg / dist_to_goal) * c.robot_v_pref, desired_vy = (dyg / dist_to_goal) * c.robot_v_pref;
        const double curr_dvx = c.sf_KI * (desired_vx - rb.vx), curr_dvy = c.sf_KI * (desired_vy - rb.vy);
        const double dx = rb.px - h.l0, dy = rb.py - h.l1;
        const double d = sqrt(dx * dx + dy * dy);
        const double f = c.sf_A * det_exp((c.robot_radius + h.l4 - d) / c.sf_B);
        const double fx = f * (dx / d), fy = f * (dy / d);
        double ivx = 0.0, ivy = 0.0;
        for (int j = 0; j < n; ++j) { ivx += __shfl(fx, j, 64); ivy += __shfl(fy, j, 64); }
        const double nvx = rb.vx + (curr_dvx + ivx) * c.time_step, nvy = rb.vy + (curr_dvy + ivy) * c.time_step;
        const double act_norm = sqrt(nvx * nvx + nvy * nvy);
        if (act_norm > c.robot_v_pref) { axd = nvx / act_norm * c.robot_v_pref; ayd = nvy / act_norm * c.robot_v_pref; }
        else { axd = nvx; ayd = nvy; }
    } else if (c.robot_policy == CN_ROBOT_ORCA) {
        // crowd_sim_var_num.py:371-375: action = robot.act(copy of last_human_states) -> ORCA.predict (orca.py:64-117) on the
        // robot's BELIEFS about all H humans (never-seen ones sit at the (15,15) dummy); no clip_action on this path
        float nd, seen_r;
        if (!s.rob_sim_valid[e] || (s.rob_sim_n && s.rob_sim_n[e] != n + 1)) { // orca.py:80-82: new simulator when the crowd size changed
            nd = (float)shared_nd;
            seen_r = (float)(h.l4 + 0.01 + c.orca_safety_space);
            if (isH) s.rob_seen[(size_t)e * H + lane] = seen_r;
            if (lane == 0) { s.rob_nd[e] = nd; s.rob_sim_valid[e] = 1; if (s.rob_sim_n) s.rob_sim_n[e] = (uint8_t)(n + 1); }
        } else {
            nd = s.rob_nd[e];
            seen_r = s.rob_seen[(size_t)e * H + (isH ? lane : 0)];
        }
        double gvx = rb.gx - rb.px, gvy = rb.gy - rb.py;
        const double speed = sqrt(gvx * gvx + gvy * gvy);
        if (speed > 1.0) { gvx = gvx / speed; gvy = gvy / speed; }
        orca_wave(lane, n, isH, (float)h.l0, (float)h.l1, (float)h.l2, (float)h.l3, seen_r, (float)rb.px, (float)rb.py, (float)rb.vx, (float)rb.vy,
                  (float)(c.robot_radius + 0.01 + c.orca_safety_space), (float)c.robot_v_pref, (float)gvx, (float)gvy, nd, n,
                  (float)c.orca_time_horizon, (float)c.time_step, ax, ay);
    } else if (c.kinematics == CN_KIN_UNICYCLE) {
        // srnn.py:36-43: (change of v, change of theta) clipped in float32; crowd_sim_var_num.py:379-381: the commanded speed is the
        // running sum self.desiredVelocity[0], clipped to +-v_pref (float64 from there on, as with the numpy the reference pins)
        const float dv = fminf(fmaxf(ax, (float)-0.1), (float)0.087);
        ay = fminf(fmaxf(ay, (float)-0.06), (float)0.06);
        uni_v = fmin(fmax(s.desired_v[e] + (double)dv, -c.robot_v_pref), c.robot_v_pref);
        uni_r = (double)ay;
        if (lane == 0) s.desired_v[e] = uni_v;
        if (s.wheel) {
            // CrowdSimPred.step (crowd_sim_pred.py:120-131) sends the command through smooth_action (crowd_sim.py:315-358): wheel speeds of
            // a Turtlebot2i (wheel radius 0.035 m, track 0.23 m) clipped to +-17.5 rad/s, low-pass filtered in the test phase, then
            // reduced towards zero by a noisy dead band N(1.8, 0.15) per wheel.  Wave-uniform; these are the first draws of the step.
            double *wh = s.wheel + (size_t)e * 4;
            const double last_left = wh[0], last_right = wh[1];
            double gauss = wh[2];
            bool has_gauss = wh[3] != 0.0;
            rng_load(R, s, e, lane);
            const double w = uni_r / c.time_step;
            double left = (2.0 * uni_v - 0.23 * w) / (2.0 * 0.035), right = (2.0 * uni_v + 0.23 * w) / (2.0 * 0.035);
            left = fmin(fmax(left, -17.5), 17.5); right = fmin(fmax(right, -17.5), 17.5);
            if (c.phase == CN_PHASE_TEST) {
                left = (1. - 0.1) * last_left + 0.1 * left;
                right = (1. - 0.1) * last_right + 0.1 * right;
            }
            const double keep_left = left, keep_right = right;
            if (left > 0) left = fmax(0., left - rng_normal(R, lane, 1.8, 0.15, gauss, has_gauss));
            else left = fmin(0., left + rng_normal(R, lane, 1.8, 0.15, gauss, has_gauss));
            if (right > 0) right = fmax(0., right - rng_normal(R, lane, 1.8, 0.15, gauss, has_gauss));
            else right = fmin(0., right + rng_normal(R, lane, 1.8, 0.15, gauss, has_gauss));
            uni_v = 0.035 / 2 * (left + right);
            uni_r = 0.035 / 0.23 * (right - left) * c.time_step;
            if (lane == 0) { wh[0] = keep_left; wh[1] = keep_right; wh[2] = gauss; wh[3] = has_gauss ? 1.0 : 0.0; }
        }
    } else {
        const float act_norm = sqrtf(ax * ax + ay * ay);
        const float vp = (float)c.robot_v_pref;
        if (act_norm > vp) { ax = ax / act_norm * vp; ay = ay / act_norm * vp; }
    }
    // humans.policy = 'social_force' (SOCIAL_FORCE.predict for every human, float64): lane i is human i and walks the list of the
    // other agents get_human_actions passes -- every other human (true state unless coincident -> the dummy at (7,7) with the
    // config radius), then the robot when robot.visible
    double sfx = 0.0, sfy = 0.0;
    if (c.humans_policy == CN_HUMANS_SOCIAL_FORCE) {
        const double dxg = h.gx - h.px, dyg = h.gy - h.py;
        const double dist_to_goal = sqrt(dxg * dxg + dyg * dyg);
        const double desired_vx = (dxg / dist_to_goal) * h.vpref, desired_vy = (dyg / dist_to_goal) * h.vpref;
        const double curr_dvx = c.sf_KI * (desired_vx - h.vx), curr_dvy = c.sf_KI * (desired_vy - h.vy);
        double ivx = 0.0, ivy = 0.0;
        for (int j = 0; j <= n; ++j) {
            if (j == n && !c.robot_visible) break;
            const int src = j < n ? j : 0; // the shuffles stay outside any conditional (they read inactive lanes as 0 otherwise)
            const double jx = __shfl(h.px, src, 64), jy = __shfl(h.py, src, 64), jr = __shfl(h.rad, src, 64);
            double ox = j < n ? jx : rb.px, oy = j < n ? jy : rb.py, orad = j < n ? jr : c.robot_radius;
            const bool hidden = c.human_fov < 2.0 ? !in_fov(c, c.human_fov, h.px, h.py, h.vx, h.vy, 0.0, ox, oy) : (ox == h.px && oy == h.py);
            if (hidden) { ox = 7.0; oy = 7.0; if (j < n) orad = c.human_radius; }
            const double dx = h.px - ox, dy = h.py - oy;
            const double d = sqrt(dx * dx + dy * dy);
            const double f = c.sf_A * det_exp((h.rad + orad - d) / c.sf_B);
            if (j != lane) { ivx += f * (dx / d); ivy += f * (dy / d); }
        }
        const double nvx = h.vx + (curr_dvx + ivx) * c.time_step, nvy = h.vy + (curr_dvy + ivy) * c.time_step;
        const double act_norm = sqrt(nvx * nvx + nvy * nvy);
        if (act_norm > h.vpref) { sfx = nvx / act_norm * h.vpref; sfy = nvy / act_norm * h.vpref; }
        else { sfx = nvx; sfy = nvy; }
    }
    // calc_reward (crowd_sim_var_num.py:465-561), pre-move positions.  "first collision in list order, break":
    // dmin is only consumed when there is no collision at all, so the lane-parallel min is equivalent.
    const double cdx = h.px - rb.px, cdy = h.py - rb.py;
    const double closest = isH ? sqrt(cdx * cdx + cdy * cdy) - h.rad - c.robot_radius : INFINITY;
    const bool collision = wv_any(closest < 0.0);
    const double dmin = wv_min(closest);
    const double goal_dist = norm2(rb.px - rb.gx, rb.py - rb.gy);
    const bool reaching_goal = goal_dist < (c.kinematics == CN_KIN_UNICYCLE ? 0.6 : c.robot_radius); // :487-492
    const double global_time = (double)step_counter * c.time_step;
    double reward;
    int done, info;
    // Danger condition: the discomfort circle (train, :496-498) or, in the test phase, an intrusion into the humans' TRUE
    // future positions k = 1..P (:499-511; humans the robot did not see in its last observation are blanked to (15,15))
    const bool test_phase = c.phase == CN_PHASE_TEST;
    bool danger_cond = dmin < c.discomfort_dist;
    double min_danger = 0.0, rf_truth = 0.0;
    if (test_phase) {
        const double *tre = s.tr + (size_t)e * (s.R + 1) * 4 * H;
        const bool seen = isH && s.vis[(size_t)e * H + lane];
        double best = INFINITY;
        for (int k = 1; k <= s.P; ++k) {
            if (isH) {
                const double fx = (seen ? tre[(k * s.I * 4 + 0) * H + lane] : 15.0) - rb.px, fy = (seen ? tre[(k * s.I * 4 + 1) * H + lane] : 15.0) - rb.py;
                const double d = sqrt(fx * fx + fy * fy);
                if (d < c.robot_radius + c.human_radius) {
                    best = fmin(best, d);
                    const double pen = c.collision_penalty / (double)(1 << (k + 1));
                    if (pen < rf_truth) rf_truth = pen;
                }
            }
        }
        best = wv_min(best);
        danger_cond = best < INFINITY;
        min_danger = danger_cond ? best : 0.0;
    } else if (c.phase == CN_PHASE_VAL) {
        // phase 'val' (CrowdSimPred-v0): the same test on what the previous observation left in self.human_future_traj -- its const_vel /
        // truth predictions, unseen humans already blanked (ftraj)
        const double *ft = s.ftraj + (size_t)e * s.P * 2 * H;
        double best = INFINITY;
        for (int k = 1; k <= s.P; ++k)
            if (isH) {
                const double fx = ft[((k - 1) * 2 + 0) * H + lane] - rb.px, fy = ft[((k - 1) * 2 + 1) * H + lane] - rb.py;
                const double d = sqrt(fx * fx + fy * fy);
                if (d < c.robot_radius + c.human_radius) best = fmin(best, d);
            }
        best = wv_min(best);
        danger_cond = best < INFINITY;
        min_danger = danger_cond ? best : 0.0;
    }
    if (c.env_kind == CN_ENV_COLLECT) {
        // crowd_sim_var_num_collect.py:139-188: the data-collection env never ends an episode (global_time >= 40000 aside) and pays no
        // reward; a robot that reaches its goal gets a new one -- the median of the humans' positions or a uniform point of the
        // arena, each with probability 1/2 (np.random draws in this order: uniform(0, 1), then uniform(-a, a, size = 2))
        reward = 0.0; done = 0; info = CN_INFO_NOTHING;
        if (global_time >= 40000.0) { done = 1; info = CN_INFO_TIMEOUT; }
        else if (collision) info = CN_INFO_COLLISION;
        else if (goal_dist < c.robot_radius) {
            info = CN_INFO_REACHGOAL;
            rng_load(R, s, e, lane);
            if (rng_uniform(R, lane, 0.0, 1.0) < 0.5) {
                // np.median(axis = 0): the middle element of the sorted column, or the mean of the two middle ones (lane-parallel rank)
                double med[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const double v = isH ? (d == 0 ? h.px : h.py) : INFINITY;
                    int rank = 0;
                    for (int m = 0; m < n; ++m) {
                        const double vm = __shfl(v, m, 64);
                        rank += (vm < v || (vm == v && m < lane)) ? 1 : 0;
                    }
                    const uint64_t hi_m = __ballot(isH && rank == n / 2), lo_m = __ballot(isH && rank == (n - 1) / 2);
                    const double vhi = __shfl(v, __ffsll((unsigned long long)hi_m) - 1, 64), vlo = __shfl(v, __ffsll((unsigned long long)lo_m) - 1, 64);
                    med[d] = (n & 1) ? vhi : (vlo + vhi) / 2.0;
                }
                rb.gx = med[0]; rb.gy = med[1];
            } else {
                rb.gx = rng_uniform(R, lane, -c.arena_size, c.arena_size);
                rb.gy = rng_uniform(R, lane, -c.arena_size, c.arena_size);
            }
        }
    }
    else if (global_time >= c.time_limit - 1.0) { reward = 0.0; done = 1; info = CN_INFO_TIMEOUT; }
    else if (collision) { reward = c.collision_penalty; done = 1; info = CN_INFO_COLLISION; }
    else if (reaching_goal) { reward = c.success_reward; done = 1; info = CN_INFO_REACHGOAL; }
    else if (danger_cond) {
        reward = (dmin - c.discomfort_dist) * c.discomfort_penalty_factor * c.time_step;
        done = 0; info = CN_INFO_DANGER;
    } else {
        reward = (c.kinematics == CN_KIN_UNICYCLE ? 3.0 : 2.0) * (-fabs(goal_dist) - rb.pot); // :536-542 pot_factor
        rb.pot = -fabs(goal_dist);
        done = 0; info = CN_INFO_NOTHING;
    }
    if (s.min_dist && lane == 0) s.min_dist[e] = info == CN_INFO_DANGER ? min_danger : 0.0; // Danger(min_dist)
    if (c.env_kind == CN_ENV_PRED && test_phase) {
        // test phase: self.human_future_traj was just overwritten by the 'truth' roll-out, so the social reward sees it too
        reward = reward + wv_min(rf_truth);
    } else if (c.env_kind == CN_ENV_PRED) {
        // social reward from the predictions stored by the previous observation (crowd_sim_pred.py:216-233)
        const double *ft = s.ftraj + (size_t)e * s.P * 2 * H;
        double rf = 0.0;
        for (int k = 1; k <= s.P; ++k) {
            const double pen = c.collision_penalty / (double)(1 << (k + 1));
            if (isH) {
                const double fx = ft[((k - 1) * 2 + 0) * H + lane] - rb.px, fy = ft[((k - 1) * 2 + 1) * H + lane] - rb.py;
                if (sqrt(fx * fx + fy * fy) < c.robot_radius + c.human_radius && pen < rf) rf = pen;
            }
        }
        reward = reward + wv_min(rf);
    }
    if (c.kinematics == CN_KIN_UNICYCLE) {
        // :548-559 rotation penalty and reversing penalty, added to every outcome
        const double r_spin = -4.5 * (uni_r * uni_r);
        const double r_back = uni_v < 0.0 ? -2.0 * fabs(uni_v) : 0.0;
        reward = reward + r_spin + r_back;
        // differential drive, agent.py:148-165.  A rotation below 1e-4 sets R = 0, i.e. the robot does not translate on that step
        // (the reference's formula, restated as it is)
        double Rr = 0.0;
        if (!(fabs(uni_r) < 0.0001)) { const double w = uni_r / c.time_step; Rr = uni_v / w; }
        double s0, c0, s1, c1;
        det_sincos(rb.theta, s0, c0);
        det_sincos(rb.theta + uni_r, s1, c1);
        rb.px = rb.px - Rr * s0 + Rr * s1;
        rb.py = rb.py + Rr * c0 - Rr * c1;
        double th = fmod(rb.theta + uni_r, 2.0 * M_PI); // Python %: the result takes the divisor's sign
        if (th != 0.0 && th < 0.0) th += 2.0 * M_PI;
        rb.theta = th;
        det_sincos(th, s0, c0);
        rb.vx = uni_v * c0; rb.vy = uni_v * s0;
    } else if (c.robot_policy == CN_ROBOT_SOCIAL_FORCE) {
        rb.px = rb.px + axd * c.time_step; rb.py = rb.py + ayd * c.time_step;
        rb.vx = axd; rb.vy = ayd;
    } else {
        // kinematics (crowd_sim/envs/utils/agent.py:170-183, holonomic)
        rb.px = rb.px + (double)(ax * (float)c.time_step);
        rb.py = rb.py + (double)(ay * (float)c.time_step);
        rb.vx = (double)ax; rb.vy = (double)ay;
    }
    if (isH && c.humans_policy == CN_HUMANS_SOCIAL_FORCE) {
        h.px = h.px + sfx * c.time_step;
        h.py = h.py + sfy * c.time_step;
        h.vx = sfx; h.vy = sfy;
        s.hact[(size_t)e * 2 * H + lane] = (float)sfx; s.hact[(size_t)e * 2 * H + H + lane] = (float)sfy; // for cn_env_get_human_actions
    } else if (isH) {
        const float hax = s.hact[(size_t)e * 2 * H + lane], hay = s.hact[(size_t)e * 2 * H + H + lane];
        h.px = h.px + (double)hax * c.time_step;
        h.py = h.py + (double)hay * c.time_step;
        h.vx = (double)hax; h.vy = (double)hay;
    }
    step_counter += 1;
    const double ep_ret = s.ep_ret[e] + reward;
    const int ep_cnt = s.ep_cnt[e] + 1;
    if (lane == 0) {
        reward_out[e] = (float)reward; done_out[e] = (uint8_t)done; info_out[e] = (uint8_t)info;
        ep_ret_out[e] = ep_ret; ep_len_out[e] = ep_cnt;
        if (not_done_out) not_done_out[e] = done ? 0.0f : 1.0f; // the `masks` tensor of train.py:185-186
    }
    if (done && c.auto_reset) {
        // vec-env auto-reset: the terminal observation is replaced by the first observation of the next episode.
        // (The terminal step's own crowd-size / goal-change / respawn draws happen before np.random.seed and cannot be observed.)
        do_reset(s, R, e, lane, rb, h, shared_nd, n, ob, !SPLIT);
        if (SPLIT && lane == 0) s.pend[e] = 1;
    } else {
        // (auto_reset == 0, the single-env gym object: a terminal step is an ordinary step -- terminal observation, goal
        // changes and respawns included, crowd_sim_var_num.py:430-458 -- and the caller resets explicitly)
        if (c.human_num_range > 0 && (step_counter % (int)(5.0 / c.time_step + 0.5)) == 0) {
            // crowd_sim_var_num.py:404-437 / crowd_sim_pred.py:165-190: every 5 s humans leave from the END of the list (only ones the
            // robot was not looking at) or new ones are appended, before the observation is generated
            rng_load(R, s, e, lane);
            if (rng_double(R, lane) < 0.5) {
                const int oc = s.obs_cnt[e], om = s.obs_max[e];
                int max_remove;
                if (c.env_kind == CN_ENV_VARNUM) {
                    max_remove = n - (c.human_num - c.human_num_range);
                    if (oc > 0 && (n - 1) - om < max_remove) max_remove = (n - 1) - om;
                } else {
                    max_remove = oc == 0 ? n - 1 : (n - 1) - om;
                    if (c.human_num_range < max_remove) max_remove = c.human_num_range;
                }
                n -= rng_randint(R, lane, 0, max_remove + 1);
            } else {
                const int add_num = rng_randint(R, lane, 0, c.human_num_range + 1);
                const int first = n;
                for (int i = first; i < first + add_num && i < H; ++i) {
                    gen_human(s, R, lane, i, i, rb, h, shared_nd);
                    if (lane == i) { h.l0 = 15.0; h.l1 = 15.0; h.l2 = 0.0; h.l3 = 0.0; h.l4 = 0.3; }
                    n = i + 1;
                }
            }
        }
        if (!SPLIT) {
            write_obs(s, e, lane, n, false, rb, h, ob, step_counter);
            if constexpr (DEFER) {
                // (a superset of the envs post_obs_updates does anything for: it evaluates `reached` after the periodic goal changes)
                bool need = c.random_goal_changing && (step_counter % (int)(5.0 / c.time_step + 0.5)) == 0;
                if (c.end_goal_changing) need = need || __ballot(lane < n && norm2(h.gx - h.px, h.gy - h.py) < h.rad) != 0;
                if (need && lane == 0) s.post_list[atomicAdd(s.post_cnt, 1)] = e;
            } else {
                post_obs_updates<W>(s, R, e, lane, n, step_counter, rb, h, shared_nd);
            }
        }
        if (lane == 0) { s.step_counter[e] = step_counter; s.ep_ret[e] = ep_ret; s.ep_cnt[e] = ep_cnt; }
    }
    store_env(s, e, lane, rb, h);
    if (lane == 0 && s.nh) s.nh[e] = n;
    if (lane == 0) s.shared_nd[e] = shared_nd;
    rng_store(R, s, e, lane);
    if constexpr (W > 1) { // release the helpers
        if (lane == 0) coop_lds<W>().cmd = 2;
        __syncthreads();
    }
}

__global__ void export_state_kernel(EnvDev s, double *humans, double *robot)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int H = s.H;
    if (humans && idx < s.E * H * 8) {
        const int e = idx / (H * 8), r = idx % (H * 8), j = r / 8, f = r % 8;
        humans[idx] = s.hum[((size_t)e * 8 + f) * H + j];
    }
    if (robot && idx < s.E * 8) robot[idx] = s.rob[idx];
}
__global__ void export_hact_kernel(EnvDev s, float *out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int H = s.H;
    if (idx < s.E * H * 2) {
        const int e = idx / (H * 2), r = idx % (H * 2), j = r / 2, f = r % 2;
        out[idx] = s.hact[((size_t)e * 2 + f) * H + j];
    }
}

__global__ void fill_i32_kernel(int n, int v, int32_t *out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = v;
}

} // namespace

struct cn_env_batch {
    EnvDev d;
    bool reset_done;
    void *blob;
    size_t blob_bytes; // the persistent state (what a snapshot holds); per-step scratch is carved behind it
    // ORCA of step t+1 only needs the simulator state left by step t, not the robot's next action: it is launched on a
    // side stream as soon as step t (or a reset) is enqueued and overlaps the caller's policy forward.
    hipStream_t side;
    hipEvent_t ev_state, ev_orca, ev_pre;
    bool orca_ready; // hact for the current state has been enqueued on `side`
    long long pregen_ticks; // time budget of one env_pregen_kernel launch (prefetch_orca)
    bool plan_ok;      // this configuration's step builds the row plan (lane kernel, crowds of <= 48: what the consumer takes)
    // the side-stream "tail" of a step (episode pre-generation + the infeasible third of the ORCA programs): launched by the step itself,
    // or -- cn_env_set_tail_deferral -- held back until the caller says that its big kernel is enqueued (cn_env_launch_tail)
    bool defer_tail, tail_pending;
    hipStream_t side2;   // deferred mode: the pre-generation runs beside the ORCA tail, not in front of it
    hipEvent_t ev_pg;
    bool pg_pending;     // ev_pg was recorded for work the next reader of the staging arrays has to wait for
    bool post_deferred;  // the step just enqueued left its post-observation updates to env_post_kernel (launch_tail runs it before ORCA)
};

// calc_human_future_traj(method='truth'): P rolls of every human with its own policy
static int truth_rollout(cn_env_batch *env, hipStream_t st)
{
    if (env->d.cfg.humans_policy == CN_HUMANS_SOCIAL_FORCE) {
        hipLaunchKernelGGL(sf_truth_kernel, dim3(env->d.E), dim3(64), 0, st, env->d);
        CN_CHECK_LAUNCH();
        return CN_OK;
    }
    const int agents = env->d.E * env->d.H;
    for (int k = 1; k <= env->d.R; ++k) { // roll k needs all of roll k - 1 of the same env: one launch per roll (R = predict_steps * pred_interval)
        hipLaunchKernelGGL(orca_truth_kernel, dim3((agents + 3) / 4), dim3(256), 0, st, env->d, k);
        CN_CHECK_LAUNCH();
    }
    return CN_OK;
}

// sim.predict_method = 'truth': roll the humans forward P times from the state the first half of the step (or the reset) left, then
// write the observation and run the post-observation updates
static int truth_rollout_and_obs(cn_env_batch *env, const cn_obs *obs, hipStream_t st)
{
    if (int rc = truth_rollout(env, st)) return rc;
    hipLaunchKernelGGL(env_obs_kernel, dim3(env->d.E), dim3(64), 0, st, env->d, *obs);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

static bool lane_path_of(const cn_env_batch *env)
{
    const int slots = env->d.H + (env->d.cfg.robot_visible ? 1 : 0); // candidate neighbours per agent (self included)
    static int coop = -1; // CN_ORCA_COOP=1 forces the one-wavefront-per-agent kernel (A/B measurements)
    if (coop < 0) { const char *v = getenv("CN_ORCA_COOP"); coop = v ? atoi(v) : 0; }
    // (a narrowed human field of view goes through the cooperative kernel: the lane kernel has no visibility test in its inner loops)
    return env->d.cfg.humans_policy == CN_HUMANS_ORCA && slots <= 32 && !coop && env->d.cfg.human_fov >= 2.0;
}

// refill the next-episode staging of the envs that just consumed theirs (rare: ~1.5 % of envs per step; a 60 us chain of serial fp64
// work per such env).  It only depends on the step that just ran; nothing needs it before those envs finish their NEXT episode.
// Budget (ticks of 10 ns; cn_env_set_pregen_budget): see cn_env_set_pregen_budget in the header.
// dense crowds (the goals' exclusion zones cover the circle: BASELINE configs[4]) run their long placement loops on four wavefronts per env
static bool dense_crowd(cn_env_batch *env)
{
    const cn_env_config &cf = env->d.cfg;
    const double zone = 2.0 * (2.0 * (cf.randomize_attributes ? 0.5 : cf.human_radius) + cf.discomfort_dist) * env->d.H;
    static const int coop_env = getenv("CN_ENV_COOP") ? atoi(getenv("CN_ENV_COOP")) : -1; // 0 / 1 force (A/B), default: by density
    static const int coop_after = getenv("CN_COOP_AFTER") ? atoi(getenv("CN_COOP_AFTER")) : COOP_AFTER;
    env->d.coop_after = coop_after;
    return coop_env >= 0 ? coop_env != 0 : zone > 0.9 * 2.0 * M_PI * cf.circle_radius;
}

static int launch_post(cn_env_batch *env, hipStream_t on)
{
    static const int post_waves = getenv("CN_POST_WAVES") ? atoi(getenv("CN_POST_WAVES")) : 4; // (A/B: 4, 8, 16)
    if (post_waves == 4) hipLaunchKernelGGL(env_post_kernel<4>, dim3(env->d.E), dim3(256), 0, on, stamped(env->d, CN_K_OTHER));
    else if (post_waves == 8) hipLaunchKernelGGL(env_post_kernel<8>, dim3(env->d.E), dim3(512), 0, on, stamped(env->d, CN_K_OTHER));
    else hipLaunchKernelGGL(env_post_kernel<16>, dim3(env->d.E), dim3(1024), 0, on, stamped(env->d, CN_K_OTHER));
    CN_CHECK_LAUNCH();
#ifdef CN_POST_DEBUG
    {
        static int calls = 0;
        (void)hipStreamSynchronize(on);
        int cnt = 0;
        (void)hipMemcpy(&cnt, env->d.post_cnt, 4, hipMemcpyDeviceToHost);
        static long long host[8192 * 8];
        (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_post_dbg), sizeof(long long) * 8 * (size_t)cnt);
        int worst = 0;
        for (int i = 0; i < cnt; ++i) if (host[i * 8] > host[worst * 8]) worst = i;
        if (++calls % 4 == 0 && cnt > 0) {
            const long long *w = host + worst * 8;
            fprintf(stderr, "[post] envs %d | worst block: %.1f us (coop %.1f us, %lld coop placements, %lld rounds: producer busy %.1f us, helper 1 screening %.1f us; %lld placements, %lld serial passes)\n",
                    cnt, w[0] * 0.01, w[1] * 0.01, w[2], w[3], w[7] * 0.01, w[6] * 0.01, w[4], w[5]);
        }
    }
#endif
    return CN_OK;
}

static int launch_pregen(cn_env_batch *env, hipStream_t on)
{
    // (one wavefront per env also for dense crowds: a NEW episode's humans are placed in 4 candidates on average -- positions, with noise of up to
    // 2 m, against goals on the far side -- it is the goal changes mid-episode that run long)
    hipLaunchKernelGGL(env_pregen_kernel<1>, dim3(env->d.E), dim3(64), 0, on, stamped(env->d, CN_K_PREGEN), env->pregen_ticks);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// The side-stream tail of the ORCA pass for the state the caller's stream has reached at this point: whatever the lane kernel could not
// finish (the infeasible programs -> linearProgram3), or the whole pass for configurations without a lane kernel; the 'truth' roll-outs of
// the test phase; and, in deferred mode, the episode pre-generation.  Ends with ev_orca, which the next step / reader waits for.
static int launch_tail(cn_env_batch *env, hipStream_t main)
{
    const int agents = env->d.E * env->d.H;
    const bool lane_path = lane_path_of(env);
    CN_HIP(hipEventRecord(env->ev_state, main));
    CN_HIP(hipStreamWaitEvent(env->side, env->ev_state, 0));
    if (env->defer_tail && lane_path) {
        // beside the ORCA tail and the caller's robot-node kernel, on a stream of its own (behind the tail on ONE stream the two chains add
        // up to ~105 us and the next step waits for them: measured 0.36 ms per step instead of 0.28)
        CN_HIP(hipStreamWaitEvent(env->side2, env->ev_state, 0));
        if (int rc = launch_pregen(env, env->side2)) return rc;
        CN_HIP(hipEventRecord(env->ev_pg, env->side2));
        env->pg_pending = true;
    }
    if (env->post_deferred) { // the goal changes / respawns of the step just enqueued: the ORCA pass below reads the new goals
        if (int rc = launch_post(env, env->side)) return rc;
        env->post_deferred = false;
    }
    if (env->d.cfg.humans_policy == CN_HUMANS_ORCA) { // social-force humans act inside env_step_kernel (one lane per human, no solver)
        if (lane_path) {
            // the infeasible programs are finished by the cooperative routine on the side stream, next to the policy forward.
            // the list length is only known on the device: a grid for a quarter of the agents (one per wavefront; the rest of the
            // wavefronts exit at once, longer lists are walked with a stride) keeps enough wavefronts in flight to hide the latency
            // of the cooperative routine
            const int blocks = (agents + 15) / 16;
            hipLaunchKernelGGL(orca_lp3_kernel, dim3(blocks), dim3(256), 0, env->side, stamped(env->d, CN_K_ORCA_LP3));
            CN_CHECK_LAUNCH();
        } else {
            hipLaunchKernelGGL(orca_kernel, dim3((agents + 3) / 4), dim3(256), 0, env->side, stamped(env->d, CN_K_ORCA_LP3));
            CN_CHECK_LAUNCH();
        }
    }
    if (env->d.cfg.phase == CN_PHASE_TEST) // 'truth' roll-out for the next step's Danger decision
        if (int rc = truth_rollout(env, env->side)) return rc;
    CN_HIP(hipEventRecord(env->ev_orca, env->side));
    env->orca_ready = true;
    env->tail_pending = false;
    return CN_OK;
}

// everything the library has in flight (or holds back) for the current state is ordered before what `st` gets next
static int sync_side(cn_env_batch *env, hipStream_t st)
{
    if (env->tail_pending) { if (int rc = launch_tail(env, st)) return rc; }
    if (env->orca_ready) CN_HIP(hipStreamWaitEvent(st, env->ev_orca, 0));
    if (env->pg_pending) { CN_HIP(hipStreamWaitEvent(st, env->ev_pg, 0)); env->pg_pending = false; }
    return CN_OK;
}

static int prefetch_orca(cn_env_batch *env, hipStream_t main, const cn_obs *obs)
{
    const float *plan_det = obs ? obs->detected_human_num : nullptr;
    int32_t *row_plan = obs ? obs->row_plan : nullptr;
    const int agents = env->d.E * env->d.H;
    const int slots = env->d.H + (env->d.cfg.robot_visible ? 1 : 0);
    const bool lane_path = lane_path_of(env);
    const bool defer = env->defer_tail && lane_path;
    if (!defer) {
        // Round 5: 40 us.  With the placement loops 64 candidates at a time an episode is generated in ~30 us, and the lane kernel in front of
        // the policy got shorter (42 us): same box, 200 steps each -- 55 us 0.2866 / 0.2860 ms per step (human-human kernel 153 us on its own
        // clock: a quarter of its workgroups wait for this kernel's CUs), 45 us 0.2825, 40 us 0.2799 (139 us), 35 us 0.2903, 30 us 0.3001 (the
        // ORCA tail then reaches the CUs early); env_step stays at 25 us down to 30 us: no env runs out of staged episodes any more.
        // Rounds 3-4 (notes kept):
        // beside the lane kernel, before the policy kernels take the whole LDS of every CU.  Budget: the lane kernel below takes ~50 us at
        // 4096 envs x 20 humans and the policy comes right behind it.  55 us cuts the long tail of the rejection sampling (up to 150 us)
        // and still lets the usual 60-odd new episodes of a step finish in one go.  Measured inside one box, human-human kernel of the
        // policy: unbounded 0.138-0.139 ms, 65 us 0.139, 55 us 0.133, 45 us 0.135, 30 us 0.161 -- shorter is NOT better in this mode: the
        // ORCA tail kernel is queued behind this one, and when it starts before the policy's kernel has its workgroups on the CUs, that
        // kernel waits for them (the deferred mode removes exactly this coupling)
        CN_HIP(hipEventRecord(env->ev_pre, main));
        if (env->post_deferred) {
            // the side stream starts with the deferred goal changes, which the ORCA pass and with it the next step wait for: the
            // pre-generation (whose workgroups mostly wait for the CUs the policy's kernel holds) goes beside them
            CN_HIP(hipStreamWaitEvent(env->side2, env->ev_pre, 0));
            if (int rc = launch_pregen(env, env->side2)) return rc;
            CN_HIP(hipEventRecord(env->ev_pg, env->side2));
            env->pg_pending = true;
        } else {
            CN_HIP(hipStreamWaitEvent(env->side, env->ev_pre, 0));
            if (int rc = launch_pregen(env, env->side)) return rc;
        }
    }
    if (lane_path) {
        // one lane per agent, on the CALLER's stream: the policy forward the caller enqueues next starts behind this kernel, not
        // beside it (see orca_lane_kernel), and a same-stream hand-over costs ~3 us where an event across streams costs 10-20
        int32_t *plan = (plan_det && env->plan_ok && ((uintptr_t)plan_det & 15u) == 0) ? row_plan : nullptr;
        if (plan) row_plan = nullptr; // built below
        const int pg = plan ? rp_groups(env->d.E) : 0;
        const dim3 grid((agents + 63) / 64 + pg), blk(64);
        const EnvDev dl = stamped(env->d, CN_K_ORCA_LANE);
        unsigned long long *pst = cn_stamp_slot(CN_K_ROW_PLAN);
        if (slots <= 8) hipLaunchKernelGGL((orca_lane_kernel<8, 8>), grid, blk, 0, main, dl, plan_det, plan, pg, pst);
        else if (slots <= 20) hipLaunchKernelGGL((orca_lane_kernel<20, 32>), grid, blk, 0, main, dl, plan_det, plan, pg, pst);
        else hipLaunchKernelGGL((orca_lane_kernel<32, 32>), grid, blk, 0, main, dl, plan_det, plan, pg, pst);
        CN_CHECK_LAUNCH();
    }
    // a caller's plan buffer that this step does not fill must not keep the previous observation's plan
    if (row_plan) CN_HIP(hipMemsetAsync(row_plan, 0, 4, main));
    if (defer) { env->tail_pending = true; env->orca_ready = false; return CN_OK; } // cn_env_launch_tail, or the next call into this batch
    return launch_tail(env, main);
}

extern "C" void cn_env_config_default(cn_env_config *c)
{
    // crowd_nav/configs/config.py:16-120 with the non-randomised training preset (BASELINE configs[1])
    *c = cn_env_config{};
    c->human_num = 20; c->predict_steps = 5; c->env_kind = CN_ENV_VARNUM;
    c->randomize_attributes = 0; c->random_goal_changing = 0; c->end_goal_changing = 1; c->sort_humans = 1;
    c->phase = CN_PHASE_TRAIN; c->nenv = 1; c->val_size = 100; c->test_size = 500; c->auto_reset = 1;
    c->time_step = 0.25; c->time_limit = 50.0;
    c->success_reward = 10.0; c->collision_penalty = -20.0; c->discomfort_dist = 0.25; c->discomfort_penalty_factor = 10.0;
    c->circle_radius = 6.0 * std::sqrt(2.0); c->arena_size = 6.0;
    c->human_radius = 0.3; c->human_v_pref = 1.0; c->robot_radius = 0.3; c->robot_v_pref = 1.0; c->sensor_range = 5.0;
    c->robot_fov = 2.0; c->human_fov = 2.0;
    c->goal_change_chance = 0.5; c->end_goal_change_chance = 1.0;
    c->orca_neighbor_dist = 10.0; c->orca_safety_space = 0.15; c->orca_time_horizon = 5.0; c->orca_time_horizon_obst = 5.0;
    c->sf_A = 2.0; c->sf_B = 1.0; c->sf_KI = 1.0; // config.py:126-128
}

extern "C" int cn_env_obs_width(const cn_env_config *cfg)
{
    if (cfg->env_kind == CN_ENV_COLLECT) return 4; // pred_info: frame id, prediction id, px, py
    return cfg->env_kind == CN_ENV_VARNUM ? 2 : 2 * (cfg->predict_steps + 1);
}

extern "C" int cn_env_create(const cn_env_config *cfg, int num_envs, int64_t seed, int64_t first_env_index, cn_env_batch **out)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(cfg && out, "cn_env_create: null argument");
    CN_REQUIRE(num_envs > 0, "cn_env_create: num_envs must be positive");
    CN_REQUIRE(cfg->human_num_range >= 0 && cfg->human_num_range < cfg->human_num, "cn_env_create: human_num_range must be in [0, human_num)");
    const int HM = cfg->human_num + cfg->human_num_range; // observation rows / lanes per env
    CN_REQUIRE(cfg->human_num >= 1 && HM <= CN_MAX_HUMANS, "cn_env_create: human_num + human_num_range must be in [1,%d]", CN_MAX_HUMANS);
    CN_REQUIRE(cfg->kinematics == CN_KIN_HOLONOMIC || (cfg->kinematics == CN_KIN_UNICYCLE && cfg->env_kind != CN_ENV_COLLECT && cfg->robot_policy == CN_ROBOT_NETWORK),
               "cn_env_create: kinematics must be holonomic, or unicycle with a network-driven robot outside CrowdSimVarNumCollect-v0 (the ORCA / "
               "social-force robot policies return ActionXY: crowd_sim_var_num.py:78-91, :379-381)");
    CN_REQUIRE(cfg->humans_policy == CN_HUMANS_ORCA || cfg->humans_policy == CN_HUMANS_SOCIAL_FORCE, "cn_env_create: unknown humans_policy %d", cfg->humans_policy);
    CN_REQUIRE(cfg->robot_fov > 0.0 && cfg->human_fov > 0.0, "cn_env_create: robot_fov / human_fov are in units of pi and must be positive (2 = all round)");
    CN_REQUIRE(cfg->predict_steps >= 1 && cfg->predict_steps <= CN_MAX_PRED, "cn_env_create: predict_steps must be in [1,%d]", CN_MAX_PRED);
    CN_REQUIRE(cfg->pred_interval >= 0 && cfg->pred_interval <= 16, "cn_env_create: pred_interval must be in [0,16] (0 = 1)");
    CN_REQUIRE(cfg->env_kind >= CN_ENV_VARNUM && cfg->env_kind <= CN_ENV_COLLECT, "cn_env_create: unknown env_kind %d", cfg->env_kind);
    CN_REQUIRE(cfg->env_kind != CN_ENV_COLLECT || (cfg->human_num_range == 0 && cfg->kinematics == CN_KIN_HOLONOMIC && cfg->phase == CN_PHASE_TRAIN &&
                                                   cfg->robot_policy == CN_ROBOT_ORCA && !cfg->predict_truth),
               "cn_env_create: CrowdSimVarNumCollect-v0 runs with a fixed crowd size, a holonomic ORCA-driven robot and phase train "
               "(what collect_data.py sets up; the reference's pred_info needs human_num_range == 0)");
    CN_REQUIRE(cfg->phase == CN_PHASE_TRAIN || cfg->phase == CN_PHASE_TEST || (cfg->phase == CN_PHASE_VAL && cfg->env_kind == CN_ENV_PRED),
               "cn_env_create: phase must be train or test, or val with CrowdSimPred-v0 (the other env classes fail in phase 'val': "
               "crowd_sim_var_num.py:501 reads self.human_future_traj, which they only assign in the test phase)");
    CN_REQUIRE(cfg->nenv >= 1, "cn_env_create: nenv (total env count) must be >= 1");
    CN_REQUIRE(cfg->robot_policy >= CN_ROBOT_NETWORK && cfg->robot_policy <= CN_ROBOT_SOCIAL_FORCE, "cn_env_create: unknown robot_policy %d", cfg->robot_policy);
    CN_REQUIRE(!cfg->robot_visible || HM <= CN_MAX_HUMANS - 1,
               "cn_env_create: robot_visible needs human_num + human_num_range <= %d (the robot is one more ORCA neighbour)", CN_MAX_HUMANS - 1);
    CN_REQUIRE(!cfg->robot_visible || cfg->env_kind != CN_ENV_PRED || cfg->predict_truth,
               "cn_env_create: robot_visible in CrowdSimPred-v0 needs sim.predict_method = 'truth' (with 'const_vel' the reference itself "
               "fails: crowd_sim_var_num.py:174 assigns H previous human states to H + 1 rows)");
    CN_REQUIRE(!cfg->predict_truth || cfg->env_kind == CN_ENV_PRED,
               "cn_env_create: predict_truth (sim.predict_method = 'truth') is CrowdSimPred-v0");
    CN_REQUIRE(cfg->time_step > 0 && std::fabs(5.0 / cfg->time_step - std::round(5.0 / cfg->time_step)) < 1e-9,
               "cn_env_create: time_step must divide 5 s");
    cn_env_batch *b = new (std::nothrow) cn_env_batch{};
    CN_REQUIRE(b, "cn_env_create: out of host memory");
    EnvDev &d = b->d;
    d.cfg = *cfg;
    d.E = num_envs; d.H = HM; d.D = cn_env_obs_width(cfg); d.P = cfg->predict_steps;
    d.I = cfg->pred_interval > 1 ? cfg->pred_interval : 1; d.R = d.P * d.I;
    d.seed_base = seed + first_env_index;
    const size_t E = num_envs, H = HM;
    // one allocation, carved (all sub-buffers 256-byte aligned)
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return o; };
    const size_t o_hum = carve(E * 8 * H * 8), o_rob = carve(E * 8 * 8), o_lhs = carve(E * 5 * H * 8);
    const size_t o_ft = cfg->env_kind == CN_ENV_PRED ? carve(E * d.P * 2 * H * 8) : 0;
    const size_t o_sc = carve(E * 4), o_cc = carve(E * 8), o_er = carve(E * 8), o_ec = carve(E * 4), o_nd = carve(E * 8);
    const size_t o_sv = carve(E * H), o_snd = carve(E * H * 4), o_ssr = carve(E * H * 4), o_ssm = carve(E * H * 4);
    const size_t o_seen = cfg->randomize_attributes ? carve(E * H * H * 4) : 0;
    const size_t o_mt = carve(E * MT_N * 4), o_mp = carve(E * 4), o_ha = carve(E * 2 * H * 4);
    const size_t o_nxh = carve(E * 8 * H * 8), o_nr = carve(E * 8 * 8), o_nn = carve(E * 8), o_nm = carve(E * MT_N * 4), o_np = carve(E * 4), o_ny = carve(E),
                 o_npg = carve(E * 4), o_ncs = carve(E * 8);
    const bool test_phase = cfg->phase == CN_PHASE_TEST;
    const bool truth_obs = cfg->predict_truth != 0;
    const bool rob_orca = cfg->robot_policy == CN_ROBOT_ORCA;
    const size_t o_rsv = rob_orca ? carve(E) : 0, o_rnd = rob_orca ? carve(E * 4) : 0, o_rsn = rob_orca ? carve(E * H * 4) : 0;
    const size_t o_tr = (test_phase || truth_obs) ? carve(E * (d.R + 1) * 4 * H * 8) : 0, o_vis = (test_phase || truth_obs) ? carve(E * H) : 0, o_md = carve(E * 8);
    const size_t o_pend = carve(E);
    const bool unicycle = cfg->kinematics == CN_KIN_UNICYCLE;
    const bool var_n = cfg->human_num_range > 0 || unicycle; // a unicycle episode holds randint(1, H + 1) humans
    const size_t o_nh = var_n ? carve(E * 4) : 0, o_nxnh = var_n ? carve(E * 4) : 0, o_oc = var_n ? carve(E * 4) : 0, o_om = var_n ? carve(E * 4) : 0;
    // agent count each private simulator was built for: the crowd size varies, or (robot.visible with 'truth' roll-outs) the real step
    // passes H others + the robot while the roll-outs pass the H - 1 fellow humans only -> two rebuilds per step (orca.py:80-82)
    const bool need_simn = var_n || (cfg->robot_visible && (test_phase || truth_obs));
    const size_t o_simn = need_simn ? carve(E * H) : 0, o_rsimn = (var_n && rob_orca) ? carve(E) : 0;
    const size_t o_dv = unicycle ? carve(E * 8) : 0;
    const bool wheel_model = unicycle && cfg->env_kind != CN_ENV_VARNUM; // CrowdSimPred.step's smooth_action
    const size_t o_wh = wheel_model ? carve(E * 4 * 8) : 0;
    const bool lane_orca = HM + (cfg->robot_visible ? 1 : 0) <= 32 && cfg->humans_policy == CN_HUMANS_ORCA && cfg->human_fov >= 2.0;
    const bool collect = cfg->env_kind == CN_ENV_COLLECT;
    const size_t o_pid = collect ? carve(E * H * 4) : 0, o_mpid = collect ? carve(E * 4) : 0, o_lobs = collect ? carve(E * H) : 0;
    const size_t state_bytes = off; // everything below is per-step scratch of the ORCA pass: not part of a snapshot
    const size_t o_pc = carve(4), o_pl = carve(E * 4); // envs with deferred post-observation updates (env_post_kernel)
    const size_t o_pa = carve(4); // arrival counter of the row-plan builders (row_plan.h): zero here, reset by the last builder of every build
    const size_t o_l3c = carve(4), o_l3h = lane_orca ? carve(E * H * sizeof(Lp3Hdr)) : 0, o_l3l = lane_orca ? carve(E * H * 32 * sizeof(float4)) : 0;
    char *base = nullptr;
    hipError_t herr = hipMalloc((void **)&base, off);
    if (herr != hipSuccess) { delete b; cn_set_error("cn_env_create: hipMalloc(%zu) failed: %s", off, hipGetErrorString(herr)); return CN_ERR_HIP; }
    herr = hipMemset(base, 0, off);
    if (herr != hipSuccess) { (void)hipFree(base); delete b; cn_set_error("cn_env_create: hipMemset failed: %s", hipGetErrorString(herr)); return CN_ERR_HIP; }
    b->blob = base;
    b->blob_bytes = state_bytes;
    d.hum = (double *)(base + o_hum); d.rob = (double *)(base + o_rob); d.lhs = (double *)(base + o_lhs);
    d.ftraj = cfg->env_kind == CN_ENV_PRED ? (double *)(base + o_ft) : nullptr;
    d.step_counter = (int32_t *)(base + o_sc); d.case_counter = (uint64_t *)(base + o_cc);
    d.ep_ret = (double *)(base + o_er); d.ep_cnt = (int32_t *)(base + o_ec); d.shared_nd = (double *)(base + o_nd);
    d.sim_valid = (uint8_t *)(base + o_sv); d.sim_nd = (float *)(base + o_snd); d.sim_self_radius = (float *)(base + o_ssr);
    d.sim_self_maxspeed = (float *)(base + o_ssm);
    d.sim_seen = cfg->randomize_attributes ? (float *)(base + o_seen) : nullptr;
    d.mt = (uint32_t *)(base + o_mt); d.mt_pos = (int32_t *)(base + o_mp); d.hact = (float *)(base + o_ha);
    d.nx_hum = (double *)(base + o_nxh); d.nx_rob = (double *)(base + o_nr); d.nx_shared_nd = (double *)(base + o_nn);
    d.nx_mt = (uint32_t *)(base + o_nm); d.nx_mt_pos = (int32_t *)(base + o_np); d.nx_ready = (uint8_t *)(base + o_ny);
    d.plan_arrive = (int32_t *)(base + o_pa);
    d.post_cnt = (int32_t *)(base + o_pc); d.post_list = (int32_t *)(base + o_pl);
    d.nx_prog = (int32_t *)(base + o_npg); d.nx_case = (uint64_t *)(base + o_ncs);
    d.tr = (test_phase || truth_obs) ? (double *)(base + o_tr) : nullptr; d.vis = (test_phase || truth_obs) ? (uint8_t *)(base + o_vis) : nullptr;
    d.pend = (uint8_t *)(base + o_pend);
    d.nh = var_n ? (int32_t *)(base + o_nh) : nullptr; d.nx_nh = var_n ? (int32_t *)(base + o_nxnh) : nullptr;
    d.obs_cnt = var_n ? (int32_t *)(base + o_oc) : nullptr; d.obs_max = var_n ? (int32_t *)(base + o_om) : nullptr;
    d.desired_v = unicycle ? (double *)(base + o_dv) : nullptr;
    d.wheel = wheel_model ? (double *)(base + o_wh) : nullptr;
    d.pred_id = collect ? (int32_t *)(base + o_pid) : nullptr; d.max_pid = collect ? (int32_t *)(base + o_mpid) : nullptr;
    d.last_obs = collect ? (uint8_t *)(base + o_lobs) : nullptr;
    d.lp3_cnt = (int32_t *)(base + o_l3c);
    d.lp3_hdr = lane_orca ? (Lp3Hdr *)(base + o_l3h) : nullptr; d.lp3_lines = lane_orca ? (float4 *)(base + o_l3l) : nullptr;
    d.sim_n = need_simn ? (uint8_t *)(base + o_simn) : nullptr; d.rob_sim_n = (var_n && rob_orca) ? (uint8_t *)(base + o_rsimn) : nullptr;
    d.min_dist = (double *)(base + o_md);
    d.rob_sim_valid = rob_orca ? (uint8_t *)(base + o_rsv) : nullptr; d.rob_nd = rob_orca ? (float *)(base + o_rnd) : nullptr;
    d.rob_seen = rob_orca ? (float *)(base + o_rsn) : nullptr;
    b->reset_done = false;
    b->orca_ready = false;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest); // side work yields to the caller's stream (critical path)
    if (hipStreamCreateWithPriority(&b->side, hipStreamNonBlocking, prio_least) != hipSuccess || hipEventCreateWithFlags(&b->ev_state, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_orca, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b->ev_pre, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithPriority(&b->side2, hipStreamNonBlocking, prio_least) != hipSuccess || hipEventCreateWithFlags(&b->ev_pg, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(base); delete b; cn_set_error("cn_env_create: stream/event creation failed"); return CN_ERR_HIP;
    }
    // the row plan is built by the lane kernel's extra workgroup: only configs that run that kernel have one (and the consumer, the
    // two-team human-human kernel, takes crowds of <= 48 humans)
    b->plan_ok = lane_orca && HM <= RP_HMAX && num_envs <= RP_EMAX;
    b->pregen_ticks = 4000; // 40 us: see prefetch_orca
    *out = b;
    return CN_OK;
}

extern "C" int cn_env_set_pregen_budget(cn_env_batch *env, int64_t ticks_10ns)
{
    CN_REQUIRE(env && ticks_10ns >= 0, "cn_env_set_pregen_budget: null handle or negative budget");
    env->pregen_ticks = ticks_10ns;
    return CN_OK;
}

extern "C" int64_t cn_row_plan_words(int num_envs) { return num_envs > 0 ? (int64_t)rp_words(num_envs) : 0; }

extern "C" int cn_env_destroy(cn_env_batch *env)
{
    if (!env) return CN_OK;
    (void)hipStreamSynchronize(env->side);
    if (env->side2) { (void)hipStreamSynchronize(env->side2); (void)hipStreamDestroy(env->side2); }
    if (env->ev_pg) (void)hipEventDestroy(env->ev_pg);
    (void)hipEventDestroy(env->ev_state); (void)hipEventDestroy(env->ev_orca); (void)hipEventDestroy(env->ev_pre); (void)hipStreamDestroy(env->side);
    if (env->blob) CN_HIP(hipFree(env->blob));
    delete env;
    return CN_OK;
}

static int check_obs(const cn_obs *obs)
{
    CN_REQUIRE(obs && obs->robot_node && obs->temporal_edges && obs->spatial_edges && obs->detected_human_num,
               "observation pointers must be non-null (visible_masks may be null)");
    // the plan builder (row_plan.h) writes the plan as int4 (and reads detected_human_num as float4: a count view at an odd offset, e.g. a
    // storage row of a batch whose size is not a multiple of four, simply gets no plan -- prefetch_orca)
    CN_REQUIRE(!obs->row_plan || ((uintptr_t)obs->row_plan & 15u) == 0,
               "cn_obs.row_plan must be 16-byte aligned and hold cn_row_plan_words(E) int32 words");
    return CN_OK;
}

extern "C" int cn_env_reset(cn_env_batch *env, const cn_obs *obs, void *stream)
{
    CN_REQUIRE(env, "cn_env_reset: null handle");
    if (int rc = check_obs(obs)) return rc;
    hipStream_t st = (hipStream_t)stream;
    // VecEnv.reset() resets every env; case counters keep running (crowd_sim_var_num.py:348)
    if (int rc = sync_side(env, st)) return rc; // an in-flight prefetch reads the old state
    const bool split = env->d.cfg.predict_truth != 0;
    hipLaunchKernelGGL(env_reset_kernel, dim3(env->d.E), dim3(64), 0, st, env->d, *obs, split ? 0 : 1);
    CN_CHECK_LAUNCH();
    if (split) { if (int rc = truth_rollout_and_obs(env, obs, st)) return rc; }
    env->reset_done = true;
    return prefetch_orca(env, st, obs);
}

extern "C" int cn_env_step(cn_env_batch *env, const float *actions, const cn_obs *obs, float *reward, uint8_t *done,
                           uint8_t *info, double *ep_return, int32_t *ep_len, float *not_done, void *stream)
{
    CN_REQUIRE(env, "cn_env_step: null handle");
    if (!env->reset_done) { cn_set_error("cn_env_step: call cn_env_reset first"); return CN_ERR_STATE; }
    if (int rc = check_obs(obs)) return rc;
    CN_REQUIRE(actions && reward && done && info && ep_return && ep_len, "cn_env_step: null output/input pointer");
    hipStream_t st = (hipStream_t)stream;
    if (!env->orca_ready && !env->tail_pending) { if (int rc = prefetch_orca(env, st, nullptr)) return rc; }
    if (int rc = sync_side(env, st)) return rc; // human velocities for the current state (computed on the side stream; a held-back tail goes out now)
    if (env->d.cfg.predict_truth) {
        hipLaunchKernelGGL(env_step_kernel<true>, dim3(env->d.E), dim3(64), 0, st, stamped(env->d, CN_K_ENV_STEP), actions, *obs, reward, done, info, ep_return, ep_len, not_done);
        CN_CHECK_LAUNCH();
        if (int rc = truth_rollout_and_obs(env, obs, st)) return rc;
    } else {
        const bool coop = dense_crowd(env);
        static const int defer_env = getenv("CN_ENV_DEFER") ? atoi(getenv("CN_ENV_DEFER")) : 1; // 0: placement loops inside the step kernel (A/B)
        // without a lane kernel the next consumer of the goals is the ORCA pass on the side stream: the updates go there, in front of it
        // (with one, that kernel follows on the caller's stream and the loops stay in the step kernel, on four wavefronts)
        const bool defer = coop && defer_env && !lane_path_of(env) && env->d.cfg.humans_policy == CN_HUMANS_ORCA;
        if (defer) {
            CN_HIP(hipMemsetAsync(env->d.post_cnt, 0, 4, st));
            hipLaunchKernelGGL((env_step_kernel<false, 1, true>), dim3(env->d.E), dim3(64), 0, st, stamped(env->d, CN_K_ENV_STEP), actions, *obs, reward, done, info, ep_return, ep_len, not_done);
            env->post_deferred = true;
            static const int post_main = getenv("CN_POST_MAIN") ? atoi(getenv("CN_POST_MAIN")) : 0; // (measurement: the kernel on its own, behind the step)
            if (post_main) { if (int rc = launch_post(env, st)) return rc; env->post_deferred = false; }
        }
        else if (coop) hipLaunchKernelGGL((env_step_kernel<false, 4>), dim3(env->d.E), dim3(256), 0, st, stamped(env->d, CN_K_ENV_STEP), actions, *obs, reward, done, info, ep_return, ep_len, not_done);
        else hipLaunchKernelGGL(env_step_kernel<false>, dim3(env->d.E), dim3(64), 0, st, stamped(env->d, CN_K_ENV_STEP), actions, *obs, reward, done, info, ep_return, ep_len, not_done);
        CN_CHECK_LAUNCH();
    }
    return prefetch_orca(env, st, obs); // next step's ORCA overlaps whatever the caller enqueues next (the policy forward)
}

extern "C" int cn_env_join(cn_env_batch *env, void *stream)
{
    CN_REQUIRE(env, "cn_env_join: null handle");
    return sync_side(env, (hipStream_t)stream);
}

extern "C" int cn_env_get_state(cn_env_batch *env, double *humans, double *robot, void *stream)
{
    CN_REQUIRE(env, "cn_env_get_state: null handle");
    if (int rc = sync_side(env, (hipStream_t)stream)) return rc; // ORCA also (re)builds sim_* lazily
    const int n = env->d.E * env->d.H * 8;
    hipLaunchKernelGGL(export_state_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, env->d, humans, robot);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_env_get_human_actions(cn_env_batch *env, float *out, void *stream)
{
    CN_REQUIRE(env && out, "cn_env_get_human_actions: null argument");
    // the velocities applied by the LAST step were overwritten by the prefetch for the next one: report the prefetched
    // ones (= the velocities the next step will apply), ordered behind the side stream
    if (int rc = sync_side(env, (hipStream_t)stream)) return rc;
    const int n = env->d.E * env->d.H * 2;
    hipLaunchKernelGGL(export_hact_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, env->d, out);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_env_get_human_counts(cn_env_batch *env, int32_t *out, void *stream)
{
    CN_REQUIRE(env && out, "cn_env_get_human_counts: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (env->d.nh) {
        CN_HIP(hipMemcpyAsync(out, env->d.nh, (size_t)env->d.E * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(fill_i32_kernel, dim3((env->d.E + 255) / 256), dim3(256), 0, st, env->d.E, env->d.H, out);
        CN_CHECK_LAUNCH();
    }
    return CN_OK;
}

extern "C" int cn_env_set_case_counters(cn_env_batch *env, const uint64_t *counters, void *stream)
{
    CN_REQUIRE(env && counters, "cn_env_set_case_counters: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = sync_side(env, st)) return rc; // the side stream may be pre-generating episodes
    CN_HIP(hipMemcpyAsync(env->d.case_counter, counters, (size_t)env->d.E * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    CN_HIP(hipMemsetAsync(env->d.nx_ready, 0, (size_t)env->d.E, st)); // staged episodes were generated for the old counters
    return CN_OK;
}

// ---- checkpointing (train.py:213-219 only saves the policy; a bit-exact --resume also needs the simulator) ----------
// The whole persistent state of a batch is ONE device blob (agent records, beliefs, counters, private ORCA simulators, the
// numpy MT19937 streams, the staged next episodes, the prefetched ORCA velocities): a snapshot is that blob behind a small
// header that pins the layout it was taken from.
struct SnapHeader {
    uint64_t magic, blob_bytes;
    int32_t E, H, D, P;
    int64_t seed_base;
    cn_env_config cfg;
    int32_t reset_done, pad;
};
constexpr uint64_t SNAP_MAGIC = 0x434e454e56303034ull; // "CNENV004" (round 4: cn_env_config gained fields since 002/003, the blob nx_prog / nx_case / wheel)
constexpr uint64_t SNAP_MAGIC_MASK = 0xffffffffff000000ull; // "CNENV" + three digits

extern "C" int64_t cn_env_snapshot_bytes(const cn_env_batch *env)
{
    return env ? (int64_t)(sizeof(SnapHeader) + env->blob_bytes) : 0;
}

extern "C" int cn_env_save(cn_env_batch *env, void *dst, void *stream)
{
    CN_REQUIRE(env && dst, "cn_env_save: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = sync_side(env, st)) return rc; // the side stream owns hact / sim_* / nx_* until then
    SnapHeader h{};
    h.magic = SNAP_MAGIC; h.blob_bytes = env->blob_bytes; h.E = env->d.E; h.H = env->d.H; h.D = env->d.D; h.P = env->d.P;
    h.seed_base = env->d.seed_base; h.cfg = env->d.cfg; h.reset_done = env->reset_done ? 1 : 0;
    CN_HIP(hipMemcpyAsync(dst, &h, sizeof(h), hipMemcpyHostToDevice, st));
    CN_HIP(hipMemcpyAsync((char *)dst + sizeof(h), env->blob, env->blob_bytes, hipMemcpyDeviceToDevice, st));
    CN_HIP(hipStreamSynchronize(st)); // `h` lives on this stack frame
    return CN_OK;
}

extern "C" int cn_env_load(cn_env_batch *env, const void *src, void *stream)
{
    CN_REQUIRE(env && src, "cn_env_load: null argument");
    hipStream_t st = (hipStream_t)stream;
    SnapHeader h{};
    CN_HIP(hipMemcpyAsync(&h, src, sizeof(h), hipMemcpyDeviceToHost, st));
    CN_HIP(hipStreamSynchronize(st));
    CN_REQUIRE((h.magic & SNAP_MAGIC_MASK) == (SNAP_MAGIC & SNAP_MAGIC_MASK), "cn_env_load: not a cn_env snapshot (bad magic)");
    CN_REQUIRE(h.magic == SNAP_MAGIC, "cn_env_load: snapshot from another layout version (CNENV%c%c%c; this library reads and writes CNENV004): "
               "snapshots do not carry over between library versions", (char)(h.magic >> 16), (char)(h.magic >> 8), (char)h.magic);
    CN_REQUIRE(h.blob_bytes == env->blob_bytes && h.E == env->d.E && h.H == env->d.H && h.D == env->d.D && h.P == env->d.P &&
                   h.seed_base == env->d.seed_base && std::memcmp(&h.cfg, &env->d.cfg, sizeof(cn_env_config)) == 0,
               "cn_env_load: the snapshot was taken from a batch with a different shape, seed, shard or configuration "
               "(E=%d H=%d seed_base=%lld vs E=%d H=%d seed_base=%lld)", h.E, h.H, (long long)h.seed_base, env->d.E, env->d.H, (long long)env->d.seed_base);
    CN_HIP(hipStreamSynchronize(env->side)); // nothing of ours may still be writing the blob
    CN_HIP(hipStreamSynchronize(env->side2));
    env->pg_pending = false;
    CN_HIP(hipMemcpyAsync(env->blob, (const char *)src + sizeof(h), env->blob_bytes, hipMemcpyDeviceToDevice, st));
    CN_HIP(hipMemsetAsync(env->d.lp3_cnt, 0, sizeof(int32_t), st)); // scratch of the ORCA pass (normally cleared by the step / reset kernels)
    env->reset_done = h.reset_done != 0;
    // the snapshot holds the prefetched velocities of its state, but the event that orders them is gone: recompute on demand
    // (orca_kernel / env_pregen_kernel are pure functions of the restored state, so the continuation is bit-identical)
    env->orca_ready = false;
    env->tail_pending = false;
    return CN_OK;
}

extern "C" int cn_env_set_tail_deferral(cn_env_batch *env, int enabled)
{
    CN_REQUIRE(env, "cn_env_set_tail_deferral: null handle");
    env->defer_tail = enabled != 0;
    return CN_OK;
}

extern "C" int cn_env_launch_tail(cn_env_batch *env, void *stream)
{
    CN_REQUIRE(env, "cn_env_launch_tail: null handle");
    if (!env->tail_pending) return CN_OK; // nothing held back (mode off, no step since, or already out)
    return launch_tail(env, (hipStream_t)stream);
}

extern "C" int cn_env_get_danger_min_dist(cn_env_batch *env, double *out, void *stream)
{
    CN_REQUIRE(env && out, "cn_env_get_danger_min_dist: null argument");
    CN_HIP(hipMemcpyAsync(out, env->d.min_dist, (size_t)env->d.E * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CN_OK;
}

extern "C" int cn_orca_solve(int B, int n_other, const float *self, const float *others, float neighbor_dist, int max_neighbors,
                             float time_horizon, float time_step, float *out_vel, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(B >= 0 && n_other >= 0 && n_other <= 63, "cn_orca_solve: n_other must be in [0,63]");
    CN_REQUIRE(self && out_vel && (others || n_other == 0), "cn_orca_solve: null pointer");
    if (B == 0) return CN_OK;
    hipLaunchKernelGGL(orca_solve_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, n_other, self, others,
                       neighbor_dist, max_neighbors, time_horizon, time_step, out_vel);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
