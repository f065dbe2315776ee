// lbm_internal.h — shared between the host (lbm_host.cpp) and device (lbm_kernels.hip) halves of
// liblbm_d2q9.so.  Not part of the ABI.
#pragma once
#include <string>

namespace lbm_internal {
// Sets the calling thread's lbm_last_error() message.
void set_error(const std::string& msg);
}  // namespace lbm_internal

// Steps of the next launch of the K-step kernels when `left` steps remain (lbm_host.cpp; lbm_plan_steps is its public
// form): shared by lbm_run, the split-phase macro-steps and the peer-to-peer loop, so that they cannot disagree.
extern "C" int lbm_plan_next(int K, int four_rows, int tail4, int left);
// The launches of the next group (one halo exchange) of a partitioned run; public form in lbm_d2q9.h.
extern "C" int lbm_plan_group(int K, int ghost, int group_max, int left, int* steps, int cap);
