// Shared host/device helpers for libcrowdnav_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "crowdnav_hip.h"

void cn_set_error(const char *fmt, ...);

#define CN_HIP(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t cn_e_ = (expr);                                                                           \
        if (cn_e_ != hipSuccess) {                                                                           \
            cn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(cn_e_), __FILE__, __LINE__);      \
            return CN_ERR_HIP;                                                                               \
        }                                                                                                    \
    } while (0)

#define CN_CHECK_LAUNCH() CN_HIP(hipGetLastError())

#define CN_REQUIRE(cond, ...)               \
    do {                                    \
        if (!(cond)) {                      \
            cn_set_error(__VA_ARGS__);      \
            return CN_ERR_INVALID;          \
        }                                   \
    } while (0)

// The opt-in above 64 KB of dynamic LDS (hipFuncSetAttribute) is per DEVICE: a call site keeps one of these as a function-local static and
// repeats the opt-in whenever the calling thread's current device is one it has not served yet (a process-wide "done" flag left every
// device but the first without the attribute: the launch then fails there).  Races between threads only repeat an idempotent call.
struct CnLdsOptIn {
    unsigned long long served = 0; // bit d: done on device d
    bool needed(int *dev_out)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        *dev_out = dev;
        return !(served & (1ull << (dev & 63)));
    }
    void done(int dev) { served |= 1ull << (dev & 63); }
};

int cn_require_device();

// ---- device-side launch stamps (measurement aid: cn_prof_set_stamps / cn_prof_next_step, include/crowdnav_hip.h) ----
// One slot of CN_PROF_SLOT_WORDS uint64 per (step, kernel) = 2 x 64 sub-slots of 16 words (128 bytes: a cache line of their own each).
// Sub-slot b of the first half receives atomicMin of the 100 MHz wall clock (s_memrealtime) when workgroup b < 64 starts, sub-slot
// (workgroup & 63) of the second half atomicMax when a wavefront ends; word 1 of the slot is free for a kernel-specific count (the
// human-human kernel: live rows).  Same-address atomics serialise in the L2 at ~6 ns each -- 20 000 wavefronts stamping ONE word
// stretched a 47 us kernel to 135 us -- hence one line per sub-slot.  A kernel's duration on the device = max(ends) - min(starts) in
// 10 ns ticks, and since the clock is global the slots of a step also give its timeline (gaps, overlaps).  NULL = no stamping.
enum { CN_K_ENV_STEP = 0, CN_K_ORCA_LANE = 1, CN_K_HH_FUSED = 2, CN_K_RN_FUSED = 3, CN_K_ORCA_LP3 = 4, CN_K_PREGEN = 5, CN_K_ROW_PLAN = 6, CN_K_OTHER = 7 };
unsigned long long *cn_stamp_slot(int kernel_id); // host: the current step's slot of `kernel_id`, or NULL (off, masked out, ring exhausted)

#ifdef __HIPCC__
struct CnStampScope {
    unsigned long long *s;
    unsigned blk;   // the block's index within ITS part of the launch (a launch that hosts two kernels' work stamps each part from 0)
    __device__ __forceinline__ explicit CnStampScope(unsigned long long *slot, int block = -1) : s(slot), blk(block < 0 ? blockIdx.x : (unsigned)block)
    {
        if (s && threadIdx.x == 0) {
            const unsigned long long now = (unsigned long long)wall_clock64();
            if (blk < 64) atomicMin(s + blk * 16, now);
            atomicMax(s + (blk & 63) * 16 + 8, now);   // latest first instruction among the workgroups b with b & 63 == slot
        }
    }
    __device__ __forceinline__ ~CnStampScope()
    {
        if (s && (threadIdx.x & 63) == 0) atomicMax(s + (64 + (blk & 63)) * 16, (unsigned long long)wall_clock64());
    }
};
#endif

#define CN_WAVE 64

// ---- wave-level primitives (64 lanes) ----
__device__ __forceinline__ float wv_readlane(float x, int lane_uniform)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane_uniform));
}
// DPP reductions (gfx9 family): row_shr 1/2/4/8 leave each 16-lane row's result in its last lane, row_bcast:15 and
// row_bcast:31 fold the rows, lane 63 holds the wave result, one v_readlane broadcasts it.  7 VALU ops instead of six
// ds_bpermute round trips.  Lanes without a DPP source keep `identity` (bound_ctrl = 0 -> `old` operand).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float wv_dpp(float v, float identity)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wv_min(float v)
{
    const float I = __builtin_inff();
    v = fminf(v, wv_dpp<0x111, 0xf>(v, I));
    v = fminf(v, wv_dpp<0x112, 0xf>(v, I));
    v = fminf(v, wv_dpp<0x114, 0xf>(v, I));
    v = fminf(v, wv_dpp<0x118, 0xf>(v, I));
    v = fminf(v, wv_dpp<0x142, 0xa>(v, I));
    v = fminf(v, wv_dpp<0x143, 0xc>(v, I));
    return wv_readlane(v, 63);
}
__device__ __forceinline__ float wv_max(float v)
{
    const float I = -__builtin_inff();
    v = fmaxf(v, wv_dpp<0x111, 0xf>(v, I));
    v = fmaxf(v, wv_dpp<0x112, 0xf>(v, I));
    v = fmaxf(v, wv_dpp<0x114, 0xf>(v, I));
    v = fmaxf(v, wv_dpp<0x118, 0xf>(v, I));
    v = fmaxf(v, wv_dpp<0x142, 0xa>(v, I));
    v = fmaxf(v, wv_dpp<0x143, 0xc>(v, I));
    return wv_readlane(v, 63);
}
__device__ __forceinline__ float wv_sum(float v)
{
    // Hillis-Steele inside each 16-lane row (row_shr 1,2,4,8), then fold the rows; total lands in lane 63
    v += wv_dpp<0x111, 0xf>(v, 0.0f);
    v += wv_dpp<0x112, 0xf>(v, 0.0f);
    v += wv_dpp<0x114, 0xf>(v, 0.0f);
    v += wv_dpp<0x118, 0xf>(v, 0.0f);
    v += wv_dpp<0x142, 0xa>(v, 0.0f);
    v += wv_dpp<0x143, 0xc>(v, 0.0f);
    return wv_readlane(v, 63);
}
__device__ __forceinline__ double wv_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wv_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
