#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "common.h"
#include "options.h"

namespace sputnik_hip {
namespace {

Options read_options() {
  Options o;
  auto num = [](const char* name, int fallback) {
    const char* e = getenv(name);
    return e ? atoi(e) : fallback;
  };
  if (const char* e = getenv("SPUTNIK_HIP_SPMM_KERNEL")) {
    if (e[0] == 'w') o.spmm_kernel = (e[1] && e[2] && e[3] && e[4] == '5') ? -2 : -1;
    else o.spmm_kernel = e[0] == 'n' ? 1 : e[0] == 'g' ? 2 : e[0] == 'p' ? 3 : e[0] == 'f' ? -3 : e[0] == 'm' ? 4 : 0;
  }
  if (const char* e = getenv("SPUTNIK_HIP_SDDMM_KERNEL"))
    o.sddmm_kernel = e[0] == 't' ? 1 : e[0] == 'w' ? 2 : e[0] == 'm' ? 3 : 0;
  o.spmm_sparse = num("SPUTNIK_HIP_SPMM_SPARSE", -1);
  o.spmm_debug = num("SPUTNIK_HIP_SPMM_DEBUG", 0);
  o.spmm_tile = num("SPUTNIK_HIP_SPMM_MEDIUM", 0);
  o.sddmm_debug = num("SPUTNIK_HIP_SDDMM_DEBUG", 0);
  o.sddmm_flat = num("SPUTNIK_HIP_SDDMM_FLAT", 1);
  o.sddmm_slab = num("SPUTNIK_HIP_SDDMM_SLAB", 0);
  o.sddmm_panel = num("SPUTNIK_HIP_SDDMM_PANEL", 0);
  o.mfma_tile = num("SPUTNIK_HIP_MFMA_TILE", 0);
  o.mfma_debug = num("SPUTNIK_HIP_MFMA_DEBUG", 0);
  o.softmax_rpg = num("SPUTNIK_HIP_SOFTMAX_RPG", 0);
  o.softmax_depth = num("SPUTNIK_HIP_SOFTMAX_DEPTH", 1);
  o.softmax_nt = num("SPUTNIK_HIP_SOFTMAX_NT", -1);
  return o;
}

std::mutex g_mutex;
Options g_options;
std::atomic<bool> g_ready{false};

}  // namespace

const Options& options() {
  if (!g_ready.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_ready.load(std::memory_order_relaxed)) {
      g_options = read_options();
      g_ready.store(true, std::memory_order_release);
    }
  }
  return g_options;
}

}  // namespace sputnik_hip

extern "C" const char* sputnik_hip_version(void) { return "sputnik_hip 0.3.0 gfx950"; }

#ifndef SPUTNIK_HIP_BUILD_ID
#define SPUTNIK_HIP_BUILD_ID "unknown"
#endif
extern "C" const char* sputnik_hip_build_id(void) { return SPUTNIK_HIP_BUILD_ID; }

// Not for use while launches are being issued from other threads.
extern "C" void sputnik_hip_reload_options(void) {
  std::lock_guard<std::mutex> lock(sputnik_hip::g_mutex);
  sputnik_hip::g_options = sputnik_hip::read_options();
  sputnik_hip::g_ready.store(true, std::memory_order_release);
}
