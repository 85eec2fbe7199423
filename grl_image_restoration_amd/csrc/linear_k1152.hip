// grl_linear_fwd instantiations for K = 768 and 1152 (see linear.hip / linear_impl.h)
#include "linear_impl.h"

int grl_linear_launch_k1152(const GrlLinearArgs& p, hipStream_t st) {
    return p.Kpad / 32 == 24 ? launch_split<24>(p, st) : launch_split<36>(p, st);
}
