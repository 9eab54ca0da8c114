// Error plumbing shared by every C-ABI entry point (include/magpo.h).
#include "common.hpp"
#include <string>

namespace magpo {
static thread_local std::string g_last_error;
void set_error(const char* msg) { g_last_error = msg ? msg : ""; }
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return MAGPO_ELAUNCH;
  }
  return MAGPO_OK;
}
}  // namespace magpo

extern "C" const char* magpo_last_error() { return magpo::g_last_error.c_str(); }
extern "C" int magpo_abi_version() { return 3; }   // 3: magpo_sable_act takes dims_host[16] (precand / defer) and fragment-major weights (magpo_act_weight_layout); 2: per-call tuning arguments (no setters), discount output of the env steps
