// conv_igemm_f16.hip -- the instantiations of conv_igemm_kernel.h for one arithmetic mode (its own translation unit: the
// modes compile in parallel).
#include "conv_igemm_kernel.h"

namespace infur {

hipError_t conv_igemm_launch_f16(const ConvArgs& a, int out_f32, int cfg, hipStream_t s) {
    return out_f32 ? launch_t<_Float16, float>(a, cfg, s) : launch_t<_Float16, _Float16>(a, cfg, s);
}
#ifdef KTRACE
hipError_t ktrace_read_f16(unsigned long long* out) { return ktrace_read_tu(out); }
#endif

}  // namespace infur
