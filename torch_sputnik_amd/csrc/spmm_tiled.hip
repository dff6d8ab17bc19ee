// LDS-tiled SpMM for gfx950 (placeholder until the tiled kernel lands: the
// dispatcher in spmm.hip then always takes the row-gather kernel).
#include "common.h"

namespace sputnik_hip {

size_t spmm_tiled_workspace_bytes(int, int, int, int) { return 0; }

int spmm_tiled_launch(int, int, int, int, int, const int*, const float*, int64_t, const int*,
                      const int*, const float*, int64_t, float*, int64_t, void*, size_t,
                      hipStream_t, bool* handled) {
  *handled = false;
  return 0;
}

}  // namespace sputnik_hip
