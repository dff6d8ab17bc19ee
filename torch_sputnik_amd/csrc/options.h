// Developer / test knobs (SPUTNIK_HIP_* environment variables), read ONCE at
// first use -- never on the launch path.  sputnik_hip_reload_options() re-reads
// them (the parity tests steer small inputs onto each kernel in turn).
#pragma once

namespace sputnik_hip {

struct Options {
  int spmm_kernel = 0;    // SPUTNIK_HIP_SPMM_KERNEL: 0 auto, -1 "wide", -2 "wide512", -3 "flat", 1 "narrow", 2 "gather", 3 "panel", 4 "mfma" (half dense operand, shared values: every shape the matrix-core kernel serves)
  int spmm_sparse = -1;   // SPUTNIK_HIP_SPMM_SPARSE: 0 / 1 forces the long- / short-segment variant (spmm_tiled); 2 / 3 / 4 the entry / group-straight / group-diagonal loop (spmm_flat)
  int spmm_debug = 0;     // SPUTNIK_HIP_SPMM_DEBUG: timing experiments only (wrong results)
  int spmm_tile = 0;      // SPUTNIK_HIP_SPMM_MEDIUM: 1 = medium, 2 = small tile
  int sddmm_kernel = 0;   // SPUTNIK_HIP_SDDMM_KERNEL: 0 auto, 1 "tiled", 2 "wave", 3 "mfma" (summed product on half operands: every shape the matrix-core kernel serves)
  int sddmm_panel = 0;    // SPUTNIK_HIP_SDDMM_PANEL (developer): forces the k-panel width of the tiled SDDMM
  int sddmm_debug = 0;    // SPUTNIK_HIP_SDDMM_DEBUG: timing experiments only (bits 8.. : the pair-flat kernel's)
  int sddmm_slab = 0;     // SPUTNIK_HIP_SDDMM_SLAB: 80 / 128 forces the slab rows of the summed product's 256-wide panels
  int sddmm_flat = 1;     // SPUTNIK_HIP_SDDMM_FLAT: 0 = planned products keep the rhs-stationary kernels
  int mfma_tile = 0;      // SPUTNIK_HIP_MFMA_TILE: 128 / 256 forces the rows of the matrix-core kernels' tile (0 = by shape)
  int mfma_debug = 0;     // SPUTNIK_HIP_MFMA_DEBUG: timing experiments only (wrong results)
  int softmax_rpg = 0;    // SPUTNIK_HIP_SOFTMAX_RPG: rows per group (0 = automatic)
  int softmax_nt = -1;    // SPUTNIK_HIP_SOFTMAX_NT (developer): nontemporal 0 none, 1 loads, 2 stores, 3 both; -1 default
  int softmax_depth = 1;  // SPUTNIK_HIP_SOFTMAX_DEPTH: rows in flight ahead (1..3)
};

const Options& options();

}  // namespace sputnik_hip
