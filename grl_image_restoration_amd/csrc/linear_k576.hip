// grl_linear_fwd instantiations for K = 384 and 576 (see linear.hip / linear_impl.h)
#include "linear_impl.h"

int grl_linear_launch_k576(const GrlLinearArgs& p, hipStream_t st) {
    return p.Kpad / 32 == 12 ? launch_split<12>(p, st) : launch_split<18>(p, st);
}
