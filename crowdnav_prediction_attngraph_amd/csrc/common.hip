// Error reporting + device probing shared by all translation units of libcrowdnav_hip.so.
#include "common.h"

#include <cstring>

static thread_local char g_err[512] = "";

void cn_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *cn_last_error(void) { return g_err; }
extern "C" int cn_version(void) { return CN_ABI_VERSION; }

// ---- launch stamps: a caller-owned device ring, one row of CN_PROF_KERNELS slots per step (see common.h) ----
static unsigned long long *g_stamp_ring = nullptr;
static int g_stamp_steps = 0, g_stamp_cur = -1;
static unsigned g_stamp_mask = 0;

unsigned long long *cn_stamp_slot(int kernel_id)
{
    if (!g_stamp_ring || g_stamp_cur < 0 || g_stamp_cur >= g_stamp_steps || !((g_stamp_mask >> kernel_id) & 1u)) return nullptr;
    return g_stamp_ring + ((size_t)g_stamp_cur * CN_PROF_KERNELS + kernel_id) * CN_PROF_SLOT_WORDS;
}

extern "C" int cn_prof_set_stamps(uint64_t *ring, int steps, unsigned kernel_mask)
{
    CN_REQUIRE((ring == nullptr) == (steps == 0) && steps >= 0, "cn_prof_set_stamps: ring and steps must both be given (or both be zero)");
    g_stamp_ring = reinterpret_cast<unsigned long long *>(ring);
    g_stamp_steps = steps; g_stamp_cur = -1; g_stamp_mask = kernel_mask;
    return CN_OK;
}

extern "C" int cn_prof_next_step(void) { return g_stamp_ring ? ++g_stamp_cur : -1; }

extern "C" int cn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// There is no CPU fallback: every entry point that touches the device fails loudly without a gfx950 GPU.
int cn_require_device()
{
    static thread_local int cached_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || cn_device_count() == 0) {
        cn_set_error("no HIP device visible (libcrowdnav_hip has no CPU fallback)");
        return CN_ERR_NO_DEVICE;
    }
    if (dev == cached_dev) return CN_OK;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        cn_set_error("hipGetDeviceProperties failed");
        return CN_ERR_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        cn_set_error("device %d is %s; libcrowdnav_hip is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return CN_ERR_NO_DEVICE;
    }
    cached_dev = dev;
    return CN_OK;
}
