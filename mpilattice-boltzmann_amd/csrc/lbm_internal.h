// lbm_internal.h — shared between the host (lbm_host.cpp) and device (lbm_kernels.hip) halves of
// liblbm_d2q9.so.  Not part of the ABI.
#pragma once
#include <string>

namespace lbm_internal {
// Sets the calling thread's lbm_last_error() message.
void set_error(const std::string& msg);
}  // namespace lbm_internal
