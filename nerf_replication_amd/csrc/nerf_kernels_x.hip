// Second translation unit of libnerf_mi355x.so: the split-fp16 ("f32x") MLP kernels, compiled with
// `-mllvm -amdgpu-mfma-vgpr-form=1` (csrc/Makefile: XFLAGS).
//
// Why a unit of their own.  hipcc's default puts every MFMA accumulator in AGPRs.  These kernels post-process each finished
// out-tile with ~150 VALU instructions (combine, ReLU, hi/lo split), which can only read VGPRs: 32 v_accvgpr_read per out-tile --
// 2000 per 128-point tile, 14 % of the issue slots of a kernel that is bound by exactly those (one wave per SIMD).  With the
// accumulators in VGPRs they disappear (the activation fragments move to AGPRs instead, which the MFMA reads directly), and
// the training instances, which spilled 64-112 bytes per lane, fit: fine inference launch -3.9 %, SAVE forward -4 % (interleaved
// A/B, LAB_NOTEBOOK.md A10).  The switch is a compiler-wide option, not a function attribute, and the fp32 kernels must NOT get
// it: it changes their register allocation so that the compiler copies registers an inline-asm global load is still writing
// (tools/check_asm_stream.py: 8-81 hazards per instance; run unchecked, the fp32 training kernels fault).  So: two units, each
// hazard-checked on its own flags; nerf_kernels.hip declares the instances below `extern template` and launches them.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "../../include/nerf_mi355x.h"
#include "nerf_layout.h"
#include "nerf_mlp_f32.hip.inc"        // MlpArgs, TrainSave, shared device helpers
#include "nerf_mlp_f16.hip.inc"        // ring constants
#include "nerf_mlp_f32x.hip.inc"
#include "nerf_mlp_bwd_f32.hip.inc"    // BwdArgs, TrainGrad
#include "nerf_mlp_bwd_f32x.hip.inc"
#include "nerf_kernels_x.inst.inc"
