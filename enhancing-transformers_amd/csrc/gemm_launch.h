// gemm_launch.h — what the planner (gemm.hip) hands to the per-operand-type kernel translation units (gemm_bf16.hip, gemm_f16.hip)
#pragma once
#include "gemm_tiles.h"

#define G4_BM 256
#define G4_BN 256

struct GemmLaunch {
  int family;            // 0 = register-staged fallback, 3 = pipe2 (128 x 128), 7 = w256 (256 x 256)
  int trans_a, trans_b;
  int mode;              // EPI_* of the call (w256 family: a template parameter of the kernel)
  int form;              // w256 family: 0 = one tile per workgroup, 1 = persistent (w256p), 2 = persistent with A staged through registers (w256r)
  int dyn;               // persistent forms: 1 = tiles claimed from the per-XCD queues, 0 = static partition
  unsigned grid;         // workgroups
  int lab;               // enh_debug_gemm_lab variant of the split-K weight-gradient loop (bf16 only; 0 = off)
};
// enqueue the kernel the plan names on `s` (no error check: the caller ends with enh_check_launch)
template <typename OT>
void gemm_launch(const GemmArgs& g, const GemmLaunch& L, hipStream_t s);
// the x3 split epilogues (bf16 only): [A-in-registers form][schedule][plain | bias + tanh]
void gemm_split_launch_bf16(const GemmArgs& g, int regstaged, int dyn, int tanh_mode, unsigned wgs, hipStream_t s);
