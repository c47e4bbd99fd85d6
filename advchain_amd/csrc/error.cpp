// Error reporting + version + the process-wide deterministic switch of the advchain C ABI.
#include <string.h>

#include <atomic>

static thread_local char g_last_error[512] = "";

extern "C" void advchain_set_error_(const char* msg) {
  strncpy(g_last_error, msg ? msg : "", sizeof(g_last_error) - 1);
  g_last_error[sizeof(g_last_error) - 1] = 0;
}

extern "C" const char* advchain_last_error(void) { return g_last_error; }

// Deterministic mode (include/advchain_hip.h): read by the scatter launchers and by advchain_scatter_workspace.
static std::atomic<int> g_deterministic{0};
extern "C" void advchain_set_deterministic(int on) { g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed); }
extern "C" int advchain_get_deterministic(void) { return g_deterministic.load(std::memory_order_relaxed); }

extern "C" int advchain_version(void) { return 130; }  // 0.1.3: deterministic mode; 0.1.2: kl term; slot_rows_max reset; gauss_small_pair, sign_axpy, nonzero_mask, consistency_finish
