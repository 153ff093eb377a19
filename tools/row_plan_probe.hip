// Stand-alone harness of the row-plan builder (csrc/row_plan.h): rp_groups(E) workgroups of 64 threads, phase timestamps of group 0.  Developer probe, not part
// of the library:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o tools/row_plan_probe.so tools/row_plan_probe.hip ;
// python tools/row_plan_probe.py [det_counts.npy]
#include <hip/hip_runtime.h>
#define RP_TIMING
#include "../crowdnav_prediction_attngraph_amd/csrc/row_plan.h"
__global__ __launch_bounds__(64) void rp_test_kernel(int E, int H, int NW, const float *det, int32_t *plan, long long *tim)
{
    __shared__ rowplan::Lds l;
    rowplan::build((int)blockIdx.x, (int)gridDim.x, E, H, NW, det, plan, l, blockIdx.x == 0 ? tim : nullptr);
}
extern "C" int rp_test(int E, int H, const float *det, int32_t *plan, long long *tim, void *stream)
{
    hipLaunchKernelGGL(rp_test_kernel, dim3(rp_groups(E)), dim3(64), 0, (hipStream_t)stream, E, H, rp_workgroups(E, H), det, plan, tim);
    return (int)hipGetLastError();
}
extern "C" int rp_test_words(int E) { return rp_words(E); }
