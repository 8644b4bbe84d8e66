// Error state + library identity for the C ABI (include/lhrs_hip.h).
#include <string.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" void lhrs_set_error(const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* lhrs_last_error(void) { return g_err; }
extern "C" int lhrs_abi_version(void) { return 1; }
extern "C" const char* lhrs_target_arch(void) { return "gfx950"; }
