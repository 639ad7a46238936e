// conv_igemm_i8.hip -- the instantiations of conv_igemm_kernel.h for one arithmetic mode (its own translation unit: the
// modes compile in parallel).
#include "conv_igemm_kernel.h"

namespace infur {

hipError_t conv_igemm_launch_i8(const ConvArgs& a, int out_f32, int cfg, hipStream_t s) {
    return out_f32 ? launch_t<signed char, float>(a, cfg, s) : launch_t<signed char, unsigned char>(a, cfg, s);
}
#ifdef KTRACE
hipError_t ktrace_read_i8(unsigned long long* out) { return ktrace_read_tu(out); }
#endif

}  // namespace infur
