// Error reporting + version for the advchain C ABI.
#include <string.h>

static thread_local char g_last_error[512] = "";

extern "C" void advchain_set_error_(const char* msg) {
  strncpy(g_last_error, msg ? msg : "", sizeof(g_last_error) - 1);
  g_last_error[sizeof(g_last_error) - 1] = 0;
}

extern "C" const char* advchain_last_error(void) { return g_last_error; }

extern "C" int advchain_version(void) { return 120; }  // 0.1.2: kl term; slot_rows_max reset; gauss_small_pair, sign_axpy, nonzero_mask, consistency_finish
