// conv_igemm_f32.hip -- the instantiations of conv_igemm_kernel.h for one arithmetic mode (its own translation unit: the
// modes compile in parallel).
#include "conv_igemm_kernel.h"

namespace infur {

hipError_t conv_igemm_launch_f32(const ConvArgs& a, int cfg, hipStream_t s) { return launch_t<float, float>(a, cfg, s); }
#ifdef KTRACE
hipError_t ktrace_read_f32(unsigned long long* out) { return ktrace_read_tu(out); }
#endif

}  // namespace infur
