// Instances of the fused acting kernel for 8 envs per wave (see act_fused_kernel.hpp).
#include "act_fused_kernel.hpp"
#ifndef MAGPO_ACT_PROF
template void launch_act<8>(const magpo::ActArgs&, hipStream_t);
#endif
