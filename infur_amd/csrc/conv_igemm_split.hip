// conv_igemm_split.hip -- the instantiations of conv_igemm_kernel.h for one arithmetic mode (its own translation unit: the
// modes compile in parallel).
#include "conv_igemm_kernel.h"

namespace infur {

hipError_t conv_igemm_launch_split(const ConvArgs& a, int fp8_cross, int cfg, hipStream_t s) {
    return fp8_cross ? launch_t<float, float, true, true>(a, cfg, s) : launch_t<float, float, true>(a, cfg, s);
}
#ifdef KTRACE
hipError_t ktrace_read_split(unsigned long long* out) { return ktrace_read_tu(out); }
#endif

}  // namespace infur
