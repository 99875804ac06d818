// libnerf_mi355x.so -- HIP kernels (gfx950 only) and the C ABI of include/nerf_mi355x.h.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC (see csrc/Makefile).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "../../include/nerf_mi355x.h"
#include "nerf_layout.h"
#include "nerf_mlp_f32.hip.inc"
#include "nerf_mlp_f16.hip.inc"
#include "nerf_mlp_f16s.hip.inc"
#include "nerf_mlp_f32x.hip.inc"
#include "nerf_wgrad_f32.hip.inc"
#include "nerf_mlp_bwd_f32.hip.inc"
#include "nerf_mlp_bwd_f32x.hip.inc"
#include "nerf_wgrad_bf16x3.hip.inc"
// the split-fp16 MLP kernels are compiled in a unit of their own (nerf_kernels_x.hip says why); here they are only launched
#define NERF_X_INST extern template
#include "nerf_kernels_x.inst.inc"

// Timing-only switches (tools/ab_bench.py) change the NUMERICS of the kernels they are compiled into.  A library
// built with any of them set must say so: it only compiles with -DNERF_TIMING_BUILD, and then reports it through
// nerf_build_flags(), which the Python loader (and any other binder) checks -- so a stray -D can no longer produce a
// library that passes nerf_abi_version() and computes garbage.
#define NERF_ANY_TIMING_HACK (NERF_F32_HACK_NOLOAD || NERF_F32_HACK_NOBIAS || NERF_F32_HACK_NORELU || NERF_F32_HACK_NOPE || \
                              0 || NERF_F32_ASM_OVERRUN || NERF_F32_HACK_NOSAVE || NERF_BWD_HACK_NOMASK || 0 || NERF_F16_HACK_NOADV || NERF_F16_HACK_NOBARRIER || NERF_WG_HACK_NOATOMIC || NERF_F16_HACK_NOEPI || 0 || NERF_F16_HACK_NORELU || NERF_F32X_HACK_NOADV || \
                              NERF_F32X_HACK_NOPE || NERF_F32X_HACK_NOEPI || NERF_F32X_HACK_SAVE_NOSTORE || NERF_XB_HACK_NOSTORE)
// (two structural knobs of the SAVE forward also break results when switched off: no rows / no sign bits stored)
#if (NERF_ANY_TIMING_HACK || NERF_SAVE_TAPS == 0 || 2 == 0) && !defined(NERF_TIMING_BUILD)
#error "a NERF_*_HACK_* / NERF_F32_ASM_OVERRUN timing switch is set: such a library computes wrong results; build it with -DNERF_TIMING_BUILD (tools/ab_bench.py does) so that nerf_build_flags() reports it"
#endif

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* what) {
  snprintf(g_err, sizeof(g_err), fmt, what);
  return code;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return NERF_ERR_HIP;
  }
  return NERF_OK;
}

// ------------------------------------------------------------------------------------ pack
struct PackArgs {
  const float* p[nerf::P_COUNT];
  float* out;
};

__device__ __forceinline__ float pack_weight_elem(const PackArgs& a, long long rel, int ntiles,
                                                  int which /*0 L0,1 hidden,2 L5a,3 L5b,4 feat,5 views*/,
                                                  int layer) {
  using namespace nerf;
  const int q = (int)(rel & 3);
  const int lane = (int)((rel >> 2) & 63);
  const long long blk = rel >> 8;                 // g*NT + j
  const int j = (int)(blk % ntiles);
  const int g = (int)(blk / ntiles);
  const int s = 4 * g + q, t = s >> 4, r = s & 15, h = lane >> 5;
  const int out = 32 * j + (lane & 31);
  switch (which) {
    case 0: { const int c = pe_xyz_feat(s, h); return c < 0 ? 0.f : a.p[P_W0][out * 63 + c]; }
    case 1: return a.p[2 * layer][out * 256 + act_feat(t, r, h)];
    case 2: { const int c = pe_xyz_feat(s, h); return c < 0 ? 0.f : a.p[10][out * 319 + c]; }
    case 3: return a.p[10][out * 319 + 63 + act_feat(t, r, h)];
    case 4: return a.p[P_WF][out * 256 + act_feat(t, r, h)];
    default: {
      if (t < 8) return a.p[P_WV][out * 283 + act_feat(t, r, h)];
      const int c = pe_dir_feat(r, h);
      return c < 0 ? 0.f : a.p[P_WV][out * 283 + 256 + c];
    }
  }
}

__global__ void nerf_pack_kernel(PackArgs a) {
  using namespace nerf;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kPackedFloats) return;
  constexpr long long WS = wsize(128, 8);
  float v;
  if (i < kOffL1) v = pack_weight_elem(a, i - kOffL0, 8, 0, 0);
  else if (i < kOffL5a) { const long long rel = i - kOffL1; v = pack_weight_elem(a, rel % WS, 8, 1, 1 + (int)(rel / WS)); }
  else if (i < kOffL5b) v = pack_weight_elem(a, i - kOffL5a, 8, 2, 5);
  else if (i < kOffL6) v = pack_weight_elem(a, i - kOffL5b, 8, 3, 5);
  else if (i < kOffFeat) { const long long rel = i - kOffL6; v = pack_weight_elem(a, rel % WS, 8, 1, 6 + (int)(rel / WS)); }
  else if (i < kOffViews) v = pack_weight_elem(a, i - kOffFeat, 8, 4, 0);
  else if (i < kOffBias) v = pack_weight_elem(a, i - kOffViews, 4, 5, 0);
  else if (i < kOffBiasViews) {          // [layer 0..8][h][j*16 + r]
    const int rel = (int)(i - kOffBias), layer = rel >> 8, h = (rel >> 7) & 1, slot = rel & 127;
    const float* b = layer < 8 ? a.p[2 * layer + 1] : a.p[P_BF];
    v = b[act_feat(slot >> 4, slot & 15, h)];
  } else if (i < kOffWAlpha) {           // [h][j*16 + r], j < 4
    const int rel = (int)(i - kOffBiasViews), h = rel >> 6, slot = rel & 63;
    v = a.p[P_BV][act_feat(slot >> 4, slot & 15, h)];
  } else if (i < kOffWRgb) {             // [h][t*16 + r]
    const int rel = (int)(i - kOffWAlpha), h = rel >> 7, slot = rel & 127;
    v = a.p[P_WA][act_feat(slot >> 4, slot & 15, h)];
  } else if (i < kOffHeadBias) {         // [c][h][t*16 + r], t < 4
    const int rel = (int)(i - kOffWRgb), c = rel >> 7, h = (rel >> 6) & 1, slot = rel & 63;
    v = a.p[P_WR][c * 128 + act_feat(slot >> 4, slot & 15, h)];
  } else {
    const int rel = (int)(i - kOffHeadBias);
    v = rel < 3 ? a.p[P_BR][rel] : a.p[P_BA][0];
  }
  a.out[i] = v;
}

// transposed stream for the backward chain (nerf_layout.h kBwd*)
__global__ void nerf_pack_bwd_kernel(PackArgs a) {
  using namespace nerf;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kBwdPackedFloats) return;
  float v = 0.0f;
  if (i < kBwdOffWAlpha) {
    // which layer: (offset, ntiles, source weight, row stride, column offset, output kind)
    long long rel; int nt; const float* W; int ld; int kind;      // kind 0: hidden out cols (+coff), 1: PE slots
    int coff = 0;
    if (i < kBwdOffWfT) { rel = i - kBwdOffWvT; nt = 8; W = a.p[P_WV]; ld = 283; kind = 0; }
    else if (i < kBwdOffW7T) { rel = i - kBwdOffWfT; nt = 8; W = a.p[P_WF]; ld = 256; kind = 0; }
    else if (i < kBwdOffW6T) { rel = i - kBwdOffW7T; nt = 8; W = a.p[14]; ld = 256; kind = 0; }
    else if (i < kBwdOffW5bT) { rel = i - kBwdOffW6T; nt = 8; W = a.p[12]; ld = 256; kind = 0; }
    else if (i < kBwdOffW5aT) { rel = i - kBwdOffW5bT; nt = 8; W = a.p[10]; ld = 319; kind = 0; coff = 63; }
    else if (i < kBwdOffW4T) { rel = i - kBwdOffW5aT; nt = 2; W = a.p[10]; ld = 319; kind = 1; }
    else if (i < kBwdOffW0T) { const long long r2 = i - kBwdOffW4T; const int li = 4 - (int)(r2 / wsize(128, 8));
                               rel = r2 % wsize(128, 8); nt = 8; W = a.p[2 * li]; ld = 256; kind = 0; }
    else { rel = i - kBwdOffW0T; nt = 2; W = a.p[P_W0]; ld = 63; kind = 1; }
    const int q = (int)(rel & 3), lane = (int)((rel >> 2) & 63);
    const long long blk = rel >> 8;
    const int j = (int)(blk % nt), gq = (int)(blk / nt);
    const int s = 4 * gq + q, t = s >> 4, r = s & 15, hk = lane >> 5, irow = lane & 31;
    const int krow = act_feat(t, r, hk);             // feature of the layer's OUTPUT (the reduction index here)
    if (kind == 0) v = W[(long long)krow * ld + coff + 32 * j + irow];
    else {
      const int hp = (irow >> 2) & 1, rp = (irow & 3) + 4 * (irow >> 3);      // accumulator row -> (register, half)
      const int c = pe_xyz_feat(16 * j + rp, hp);
      v = c < 0 ? 0.0f : W[(long long)krow * ld + c];
    }
  } else if (i < kBwdOffWRgb) {
    const int rel = (int)(i - kBwdOffWAlpha), h = rel >> 7, slot = rel & 127;
    v = a.p[P_WA][act_feat(slot >> 4, slot & 15, h)];
  } else if (i < kBwdOffWRgb + 384) {
    const int rel = (int)(i - kBwdOffWRgb), c = rel >> 7, h = (rel >> 6) & 1, slot = rel & 63;
    v = a.p[P_WR][c * 128 + act_feat(slot >> 4, slot & 15, h)];
  }
  a.out[i] = v;
}

// fp32 value of element (fragment F, lane, j) of the fp16 fragment stream (nerf_layout.h)
__device__ __forceinline__ float f16_stream_value(const PackArgs& a, int F, int lane, int j) {
  using namespace nerf;
  const int h = lane >> 5, row = lane & 31;
  const int cj = (j & 3) + 8 * (j >> 2) + 4 * h;              // act16_feat(s,j,h) - 16 s
  float v = 0.0f;
  if (F < kF16FragL1) {                                       // L0: 8 m x 4 PE k-steps
    const int m = F >> 2, s = F & 3, c = pe_xyz_feat(8 * s + j, h);
    if (c >= 0) v = a.p[P_W0][(32 * m + row) * 63 + c];
  } else if (F < kF16FragL5) {                                // L1..L4
    const int f = F - kF16FragL1, li = 1 + (f >> 7), m = (f & 127) >> 4, s = f & 15;
    v = a.p[2 * li][(32 * m + row) * 256 + 16 * s + cj];
  } else if (F < kF16FragL6) {                                // L5: 4 PE + 16 hidden k-steps per m
    const int f = F - kF16FragL5, m = f / 20, s = f % 20;
    if (s < 4) { const int c = pe_xyz_feat(8 * s + j, h); if (c >= 0) v = a.p[10][(32 * m + row) * 319 + c]; }
    else v = a.p[10][(32 * m + row) * 319 + 63 + 16 * (s - 4) + cj];
  } else if (F < kF16FragSigma) {                             // L6, L7
    const int f = F - kF16FragL6, li = 6 + (f >> 7), m = (f & 127) >> 4, s = f & 15;
    v = a.p[2 * li][(32 * m + row) * 256 + 16 * s + cj];
  } else if (F < kF16FragFeat) {                              // sigma head: row 0 only
    const int s = F - kF16FragSigma;
    if (row == 0) v = a.p[P_WA][16 * s + cj];
  } else if (F < kF16FragViews) {                             // feature
    const int f = F - kF16FragFeat, m = f >> 4, s = f & 15;
    v = a.p[P_WF][(32 * m + row) * 256 + 16 * s + cj];
  } else if (F < kF16FragRgb) {                               // views: 16 feature + 2 dir k-steps per m
    const int f = F - kF16FragViews, m = f / 18, s = f % 18;
    if (s < 16) v = a.p[P_WV][(32 * m + row) * 283 + 16 * s + cj];
    else { const int c = pe_dir_feat(8 * (s - 16) + j, h); if (c >= 0) v = a.p[P_WV][(32 * m + row) * 283 + 256 + c]; }
  } else {                                                    // rgb head: rows 0..2
    const int s = F - kF16FragRgb;
    if (row < 3) v = a.p[P_WR][row * 128 + 16 * s + cj];
  }
  return v;
}

// fp32 value of element (fragment F, lane, j) of the f16s fragment stream (16x16x32 tiling, nerf_layout.h "f16s")
__device__ __forceinline__ float f16s_stream_value(const PackArgs& a, int F, int lane, int j) {
  using namespace nerf;
  const int g = lane >> 4, row = lane & 15;
  const int cj = 16 * (j >> 2) + 4 * g + (j & 3);             // act16s_feat(s,j,g) - 32 s
  float v = 0.0f;
  if (F < kF16sFragL1) {                                      // L0: 16 m x 2 PE k-steps
    const int m = F >> 1, s = F & 1, c = pe16s_xyz_feat(8 * s + j, g);
    if (c >= 0) v = a.p[P_W0][(16 * m + row) * 63 + c];
  } else if (F < kF16sFragL5) {                               // L1..L4
    const int f = F - kF16sFragL1, li = 1 + (f >> 7), m = (f & 127) >> 3, s = f & 7;
    v = a.p[2 * li][(16 * m + row) * 256 + 32 * s + cj];
  } else if (F < kF16sFragL6) {                               // L5: 2 PE + 8 hidden k-steps per m
    const int f = F - kF16sFragL5, m = f / 10, s = f % 10;
    if (s < 2) { const int c = pe16s_xyz_feat(8 * s + j, g); if (c >= 0) v = a.p[10][(16 * m + row) * 319 + c]; }
    else v = a.p[10][(16 * m + row) * 319 + 63 + 32 * (s - 2) + cj];
  } else if (F < kF16sFragSigma) {                            // L6, L7
    const int f = F - kF16sFragL6, li = 6 + (f >> 7), m = (f & 127) >> 3, s = f & 7;
    v = a.p[2 * li][(16 * m + row) * 256 + 32 * s + cj];
  } else if (F < kF16sFragFeat) {                             // sigma head: row 0 only
    const int s = F - kF16sFragSigma;
    if (row == 0) v = a.p[P_WA][32 * s + cj];
  } else if (F < kF16sFragViews) {                            // feature
    const int f = F - kF16sFragFeat, m = f >> 3, s = f & 7;
    v = a.p[P_WF][(16 * m + row) * 256 + 32 * s + cj];
  } else if (F < kF16sFragRgb) {                              // views: 8 feature + 1 dir k-step per m
    const int f = F - kF16sFragViews, m = f / 9, s = f % 9;
    if (s < 8) v = a.p[P_WV][(16 * m + row) * 283 + 32 * s + cj];
    else { const int c = pe16s_dir_feat(j, g); if (c >= 0) v = a.p[P_WV][(16 * m + row) * 283 + 256 + c]; }
  } else if (F < kF16sFragEnd) {                              // rgb head: rows 0..2
    const int s = F - kF16sFragRgb;
    if (row < 3) v = a.p[P_WR][row * 128 + 32 * s + cj];
  }                                                           // (tail: zero pad fragments)
  return v;
}

// f16s stream: const region (fp32 biases in NATURAL order) + A fragments
__global__ void nerf_pack_f16s_kernel(PackArgs a) {
  using namespace nerf;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr long long n_const = kF16ConstBytes / 4;
  constexpr long long n_half = (long long)kF16Frags * 512;
  if (i < n_const) {
    float v = 0.0f;
    if (i < kF16sOffBiasViews) {
      const int layer = (int)i >> 8, c = (int)i & 255;
      v = (layer < 8 ? a.p[2 * layer + 1] : a.p[P_BF])[c];
    } else if (i < kF16sOffHeadBias) v = a.p[P_BV][(int)i - kF16sOffBiasViews];
    else if (i < kF16sOffHeadBias + 4) { const int rel = (int)i - kF16sOffHeadBias; v = rel < 3 ? a.p[P_BR][rel] : a.p[P_BA][0]; }
    a.out[i] = v;
    return;
  }
  const long long e = i - n_const;
  if (e >= n_half) return;
  const int F = (int)(e >> 9), lane = (int)((e >> 3) & 63), j = (int)(e & 7);
  reinterpret_cast<_Float16*>(reinterpret_cast<char*>(a.out) + kF16ConstBytes)[e] = (_Float16)f16s_stream_value(a, F, lane, j);
}

// backward f32x stream: transposed weights as (hi, lo) fragment pairs + w_alpha / w_rgb in the const region
__global__ void nerf_pack_bwd_f32x_kernel(PackArgs a) {
  using namespace nerf;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr long long n_const = kF16ConstBytes / 4;
  constexpr long long n_el = (long long)kXbSteps * 512;
  if (i < n_const) {
    float v = 0.0f;
    if (i < 256) { const int h = (int)i >> 7, slot = (int)i & 127; v = a.p[P_WA][act_feat(slot >> 4, slot & 15, h)]; }
    else if (i < 256 + 384) { const int rel = (int)i - 256, c = rel >> 7, h = (rel >> 6) & 1, slot = rel & 63;
                              v = a.p[P_WR][c * 128 + act_feat(slot >> 4, slot & 15, h)]; }
    a.out[i] = v;
    return;
  }
  const long long e = i - n_const;
  if (e >= n_el) return;
  const int F = (int)(e >> 9), lane = (int)((e >> 3) & 63), j = (int)(e & 7);
  const int hk = lane >> 5, irow = lane & 31;
  // locate (layer, m, s): W^T fragment value = W[krow = out feature of the layer][col = in feature]
  const float* W; int ld, coff = 0, ks, f; bool pe_out = false;
  if (F < kXbStepsWfT) { f = F - kXbStepsWvT; ks = 8; W = a.p[P_WV]; ld = 283; }
  else if (F < kXbStepsW5aT) { const int rel = F - kXbStepsWfT, li = rel >> 7; f = rel & 127; ks = 16;
    if (li == 0) { W = a.p[P_WF]; ld = 256; } else if (li == 1) { W = a.p[14]; ld = 256; }
    else if (li == 2) { W = a.p[12]; ld = 256; } else { W = a.p[10]; ld = 319; coff = 63; } }
  else if (F < kXbStepsW4T) { f = F - kXbStepsW5aT; ks = 16; W = a.p[10]; ld = 319; pe_out = true; }
  else if (F < kXbStepsW0T) { const int rel = F - kXbStepsW4T, li = 4 - (rel >> 7); f = rel & 127; ks = 16; W = a.p[2 * li]; ld = 256; }
  else { f = F - kXbStepsW0T; ks = 16; W = a.p[P_W0]; ld = 63; pe_out = true; }
  const int m = f / ks, s = f % ks;
  const int krow = act16_feat(s, j, hk);
  float v;
  if (!pe_out) v = W[(long long)krow * ld + coff + 32 * m + irow];
  else {
    const int hp = (irow >> 2) & 1, rp = (irow & 3) + 4 * (irow >> 3);
    const int c = pe_xyz_feat(16 * m + rp, hp);
    v = c < 0 ? 0.0f : W[(long long)krow * ld + c];
  }
  _Float16* out = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(a.out) + kF16ConstBytes);
  const _Float16 hi = (_Float16)v;
  const long long base = ((long long)(2 * F) << 9) + (e & 511);
  out[base] = hi;
  out[base + 512] = (_Float16)((v - (float)hi) * 2048.0f);
}

// fp16 stream (nerf_layout.h "fp16-activation path"): const region (fp32 biases) + A fragments; with SPLIT the
// "f32x" stream: every fragment followed by its low-part fragment (w = w_h + 2^-11 w_l)
template <bool SPLIT>
__global__ void nerf_pack_f16_kernel(PackArgs a) {
  using namespace nerf;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr long long n_const = kF16ConstBytes / 4;
  constexpr long long n_half = (long long)kF16Frags * 512;
  if (i < n_const) {                     // const region, floats
    float v = 0.0f;
    if (i < kF16OffBiasViews) {
      const int rel = (int)i, layer = rel >> 8, h = (rel >> 7) & 1, slot = rel & 127;
      const float* b = layer < 8 ? a.p[2 * layer + 1] : a.p[P_BF];
      v = b[act_feat(slot >> 4, slot & 15, h)];
    } else if (i < kF16OffHeadBias) {
      const int rel = (int)i - kF16OffBiasViews, h = rel >> 6, slot = rel & 63;
      v = a.p[P_BV][act_feat(slot >> 4, slot & 15, h)];
    } else if (i < kF16OffHeadBias + 4) {
      const int rel = (int)i - kF16OffHeadBias;
      v = rel < 3 ? a.p[P_BR][rel] : a.p[P_BA][0];
    }
    a.out[i] = v;
    return;
  }
  const long long e = i - n_const;
  if (e >= n_half) return;
  const int F = (int)(e >> 9), lane = (int)((e >> 3) & 63), j = (int)(e & 7);
  const float v = f16_stream_value(a, F, lane, j);
  _Float16* out = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(a.out) + kF16ConstBytes);
  if (!SPLIT) { out[e] = (_Float16)v; return; }
  const _Float16 hi = (_Float16)v;
  const long long base = ((long long)(2 * F) << 9) + (e & 511);
  out[base] = hi;
  out[base + 512] = (_Float16)((v - (float)hi) * 2048.0f);
}

// ------------------------------------------------------------------------------------ PE (test entry)
__global__ void nerf_pe_kernel(const float* __restrict__ x, long long n, int n_freqs, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int ch = 3 + 6 * n_freqs;
  if (i >= n * 3) return;
  const long long p = i / 3;
  const int c = (int)(i - p * 3);
  const float v = x[i];
  out[p * ch + c] = v;
  for (int k = 0; k < n_freqs; ++k) {
    float sv, cv;
    sincosf(v * (float)(1 << k), &sv, &cv);
    out[p * ch + 3 + 6 * k + c] = sv;
    out[p * ch + 3 + 6 * k + 3 + c] = cv;
  }
}

// ------------------------------------------------------------------------------------ fine sampling
// One thread per ray, sequential scans in the reference's CPU order (cumprod / cumsum are
// sequential in torch CPU too).  A 64-ray block stages its I/O through LDS: the coarse sigmas are
// loaded cooperatively, every lane then owns one padded 192-float LDS row (sigma -> fine depths in
// [0,128) with the cdf in [128,191) -> the merged depths), and the 64 x 192 sorted depths are
// written back coalesced.  Optionally (fast_sampling) also the ESS/ERT validity of every merged
// sample (volume_renderer.py:116-123, :132-133, :158-193, :359-369), carried through the merge as
// the sign of the depth.
constexpr int kSampleThreads = 64;
constexpr int kBufPitch = 193;     // odd pitch: lanes walking their own rows never share a bank

struct SampleArgs {
  const float* raw_c;        // [n,64,4]
  const float* t_coarse;     // [64]
  const float* u_tab;        // [128]
  long long n_rays;
  float* t_sorted;           // [n,192]
  float* t_fine;             // optional [n,128]
  unsigned char* valid_sorted;   // optional [n,192], 1 = evaluate with the fine network
  int fast_sampling;
  float weights_threshold, ert_threshold;
};

__device__ __forceinline__ float alpha_of(float sigma, float delta) {
  return __fsub_rn(1.0f, expf(__fmul_rn(-sigma, delta)));       // 1 - exp(-sigma*delta)
}

// torch.sum(w, -1) of volume_renderer.py:138 on a 62-element fp32 row, in the CPU kernel's own order (ATen SumKernel.cpp,
// vectorized_inner_sum / row_sum with 8-lane vectors and 4 accumulators -- verified bit for bit against torch.sum on the fixture
// rows): the 56 leading elements as 7 vectors of 8, lane sums ((((((v0 + v4) + v5) + v6) + v1) + v2) + v3), then the 6 tail
// elements and the 8 lane sums added one by one.  A plain left-to-right sum (rounds 1-2) is off by up to 3e-6 relative on rows of
// many equal 1e-5 terms, i.e. cdf[62] = 1 + 3e-6: it moved every fine sample of a ray by ~3e-6 against the reference's and was
// the "unexplained" part of the white-noise family's excess over the noise floor (round-2 VERDICT Weak 8).
template <class Get>
__device__ __forceinline__ float torch_sum62_of(Get x) {
  float P[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float p = __fadd_rn(x(j), x(32 + j));
    p = __fadd_rn(p, x(40 + j));
    p = __fadd_rn(p, x(48 + j));
    p = __fadd_rn(p, x(8 + j));
    p = __fadd_rn(p, x(16 + j));
    P[j] = __fadd_rn(p, x(24 + j));
  }
  float acc = x(56);
#pragma unroll
  for (int k = 57; k < 62; ++k) acc = __fadd_rn(acc, x(k));
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __fadd_rn(acc, P[j]);
  return acc;
}
__device__ __forceinline__ float torch_sum62(const float* w) { return torch_sum62_of([&](int k) { return w[k]; }); }

__global__ __launch_bounds__(kSampleThreads)
void nerf_sample_fine_kernel(SampleArgs a) {
  constexpr int S = NERF_N_SAMPLES, F = NERF_N_IMPORTANCE, NB = S - 1;   // 63 cdf entries / bins
  __shared__ float s_tc[S];
  __shared__ float s_u[F];
  __shared__ float s_buf[kSampleThreads * kBufPitch];
  const int lane = threadIdx.x;
  const long long ray0 = (long long)blockIdx.x * kSampleThreads;
  s_tc[lane] = a.t_coarse[lane];
  s_u[lane] = a.u_tab[lane];
  s_u[lane + 64] = a.u_tab[lane + 64];
  for (int r = 0; r < kSampleThreads; ++r) {          // lane = coarse sample index
    long long rg = ray0 + r;
    if (rg >= a.n_rays) rg = a.n_rays - 1;
    s_buf[r * kBufPitch + lane] = a.raw_c[rg * (S * 4) + lane * 4 + 3];
  }
  __syncthreads();
  float* buf = s_buf + lane * kBufPitch;
  float* cdf = buf + F;        // slots 128..190: free until the merge, which no longer needs the cdf

  // weights of the coarse pass (volume_renderer.py:67-96), inner 62 + eps, running sum
  float T = 1.0f, wsum = 0.0f, dsum = 0.0f, dmax = 0.0f;
  unsigned long long empty_bits = 0, ert_bits = 0;
  bool ert_seen = false;
  for (int i = 0; i < S; ++i) {
    const float sigma = fmaxf(buf[i], 0.0f);
    dsum = __fadd_rn(dsum, sigma);
    dmax = fmaxf(dmax, sigma);
    const float delta = (i < S - 1) ? __fsub_rn(s_tc[i + 1], s_tc[i]) : 1e10f;
    const float alpha = alpha_of(sigma, delta);
    const float w = __fmul_rn(T, alpha);
    if (i >= 1 && i <= S - 2) {
      const float we = __fadd_rn(w, 1e-5f);
      cdf[i - 1] = we;                                  // stash w+eps in cdf slots 0..61 (summed below, in torch's order)
      if (w < a.weights_threshold) empty_bits |= 1ull << (i - 1);
      ert_seen = ert_seen || (T < a.ert_threshold);     // cummax of (T < thr) over the inner bins
      if (ert_seen) ert_bits |= 1ull << (i - 1);
    }
    T = __fmul_rn(T, fminf(fmaxf(__fsub_rn(1.0f, alpha), 1e-10f), 1.0f));
  }
  wsum = torch_sum62(cdf);
  const bool empty_ray = dsum < 1e-3f;
  const bool object_ray = dmax > 0.5f;
  // cdf = [0, cumsum(pdf)]  (63 entries)
  {
    float run = 0.0f, prev = cdf[0];
    cdf[0] = 0.0f;
    for (int m = 1; m < NB; ++m) {
      run = __fadd_rn(run, __fdiv_rn(prev, wsum));
      prev = cdf[m];
      cdf[m] = run;
    }
  }
  // inverse CDF at the fixed u table; u ascending -> the searchsorted(right=True) index only grows
  const long long ray = ray0 + lane;
  const bool ray_ok = ray < a.n_rays;
  int ind = 0;
  for (int k = 0; k < F; ++k) {
    const float u = s_u[k];
    while (ind < NB && cdf[ind] <= u) ++ind;
    const int below = min(max(ind - 1, 0), S - 3);
    const int above = min(ind, S - 3);                 // clamp to 61: tail collapse
    const float cb = cdf[below], ca = cdf[above];
    const float bb = __fmul_rn(0.5f, __fadd_rn(s_tc[below + 1], s_tc[below]));
    const float ba = __fmul_rn(0.5f, __fadd_rn(s_tc[above + 1], s_tc[above]));
    float denom = __fsub_rn(ca, cb);
    if (denom < 1e-5f) denom = 1.0f;
    const float frac = __fdiv_rn(__fsub_rn(u, cb), denom);
    float v = __fadd_rn(bb, __fmul_rn(frac, __fsub_rn(ba, bb)));
    if (a.t_fine && ray_ok) a.t_fine[ray * F + k] = v;
    if (a.fast_sampling) {
      const bool be = (empty_bits >> below) & 1, ae = (empty_bits >> above) & 1, ert = (ert_bits >> below) & 1;
      const bool ess_nv = object_ray ? (be && ae) : (be || ae);
      if (ess_nv || ert || empty_ray) v = -v;           // depths are > 0: the sign carries "masked out"
    }
    // keep the fine depths sorted even if rounding ever produced a 1-ulp inversion (torch.sort would fix it too)
    int j = k;
    while (j > 0 && fabsf(buf[j - 1]) > fabsf(v)) { buf[j] = buf[j - 1]; --j; }
    buf[j] = v;
  }
  // in-place two-way merge from the back of (sorted coarse table, sorted fine depths) = cat + torch.sort;
  // on ties the coarse sample comes first
  {
    int ic = S - 1, jf = F - 1;
    for (int k = S + F - 1; k >= 0; --k) {
      const bool take_f = (ic < 0) || (jf >= 0 && fabsf(buf[jf]) >= s_tc[ic]);
      if (take_f) { buf[k] = buf[jf]; --jf; }
      else { buf[k] = s_tc[ic]; --ic; }
    }
  }
  __syncthreads();
  for (int i = 0; i < S + F; ++i) {                    // 64 rays x 192 depths, coalesced
    const int e = lane + 64 * i, r = e / (S + F), k = e - r * (S + F);
    const long long rg = ray0 + r;
    if (rg < a.n_rays) {
      const float v = s_buf[r * kBufPitch + k];
      a.t_sorted[rg * (S + F) + k] = fabsf(v);
      if (a.valid_sorted) a.valid_sorted[rg * (S + F) + k] = v > 0.0f;
    }
  }
}

// Stream compaction of the valid (ray, sample) ids for the masked fine pass (replaces the boolean
// indexing of network.py:207-214): order inside `index` is arbitrary, results are scattered back by id.
__global__ __launch_bounds__(256)
void nerf_compact_kernel(const unsigned char* __restrict__ valid, long long n, int* __restrict__ index,
                         int* __restrict__ count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool v = i < n && valid[i] != 0;
  const unsigned long long m = __ballot(v);
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(count, (int)__popcll(m));
  base = __shfl(base, 0);
  if (v) index[base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (int)i;
}

// Point ids of the LAST sample of every ray (the fp16 paths' far-plane guard, nerf_render_forward): index[r] = r * S + S - 1.
__global__ __launch_bounds__(256)
void nerf_last_sample_index_kernel(int* __restrict__ index, int* __restrict__ count, long long n_rays, int S) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_rays) index[r] = (int)(r * S + S - 1);
  if (r == 0) *count = (int)n_rays;
}

// ------------------------------------------------------------------------------------ compositing
__global__ __launch_bounds__(256)
void nerf_composite_kernel(const float* __restrict__ raw, const float* __restrict__ tvals,
                           long long t_ray_stride, long long n_rays, int S, int white_bkgd,
                           float* __restrict__ rgb_out, float* __restrict__ depth_out,
                           float* __restrict__ weights_out) {
  const long long ray = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n_rays) return;
  const f32x4* r4 = reinterpret_cast<const f32x4*>(raw) + ray * S;
  const float* t = tvals + ray * t_ray_stride;
  float T = 1.0f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_w = 0.f;
  float t_cur = t[0];
  for (int k = 0; k < S; ++k) {
    const f32x4 v = r4[k];
    const float t_next = (k < S - 1) ? t[k + 1] : 0.0f;
    const float delta = (k < S - 1) ? __fsub_rn(t_next, t_cur) : 1e10f;
    const float alpha = alpha_of(fmaxf(v.w, 0.0f), delta);
    const float w = __fmul_rn(T, alpha);
    T = __fmul_rn(T, fminf(fmaxf(__fsub_rn(1.0f, alpha), 1e-10f), 1.0f));
    const float cr = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-v.x)));     // sigmoid
    const float cg = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-v.y)));
    const float cb = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-v.z)));
    acc_r = __fadd_rn(acc_r, __fmul_rn(w, cr));
    acc_g = __fadd_rn(acc_g, __fmul_rn(w, cg));
    acc_b = __fadd_rn(acc_b, __fmul_rn(w, cb));
    acc_d = __fadd_rn(acc_d, __fmul_rn(w, t_cur));
    acc_w = __fadd_rn(acc_w, w);
    if (weights_out) weights_out[ray * S + k] = w;
    t_cur = t_next;
  }
  if (white_bkgd) {
    const float bg = __fsub_rn(1.0f, acc_w);
    acc_r = __fadd_rn(acc_r, bg); acc_g = __fadd_rn(acc_g, bg); acc_b = __fadd_rn(acc_b, bg);
  }
  rgb_out[ray * 3 + 0] = acc_r; rgb_out[ray * 3 + 1] = acc_g; rgb_out[ray * 3 + 2] = acc_b;
  depth_out[ray] = acc_d;
}

// Same arithmetic, same order, but HBM-friendly: one wave per 64 rays; each 16-sample chunk of the
// 64 rays is loaded cooperatively (256-B contiguous pieces per ray) into padded LDS rows, then every
// lane walks its own ray's row.  2.5 GB of reads per 800x800 frame at close to streaming rate
// instead of 64 lanes striding 3 KiB apart.  Bit-identical results to nerf_composite_kernel.
constexpr int kCompChunk = 16;
__global__ __launch_bounds__(64)
void nerf_composite_staged_kernel(const float* __restrict__ raw, const float* __restrict__ tvals,
                                  long long t_ray_stride, long long n_rays, int S, int white_bkgd,
                                  float* __restrict__ rgb_out, float* __restrict__ depth_out,
                                  float* __restrict__ weights_out) {
  __shared__ f32x4 s_raw[64 * (kCompChunk + 1)];
  __shared__ float s_t[64 * (kCompChunk + 1)];
  const int lane = threadIdx.x;
  const long long ray0 = (long long)blockIdx.x * 64;
  const long long ray = ray0 + lane;
  const bool valid = ray < n_rays;
  const f32x4* r4 = reinterpret_cast<const f32x4*>(raw);
  float T = 1.0f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_w = 0.f;
  f32x4 pv = {0.f, 0.f, 0.f, 0.f};      // pending sample (its delta needs the next depth)
  float pt = 0.0f;
  auto emit = [&](f32x4 v, float t_cur, float delta, int k) {
    const float alpha = alpha_of(fmaxf(v.w, 0.0f), delta);
    const float w = __fmul_rn(T, alpha);
    T = __fmul_rn(T, fminf(fmaxf(__fsub_rn(1.0f, alpha), 1e-10f), 1.0f));
    const float cr = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-v.x)));
    const float cg = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-v.y)));
    const float cb = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-v.z)));
    acc_r = __fadd_rn(acc_r, __fmul_rn(w, cr));
    acc_g = __fadd_rn(acc_g, __fmul_rn(w, cg));
    acc_b = __fadd_rn(acc_b, __fmul_rn(w, cb));
    acc_d = __fadd_rn(acc_d, __fmul_rn(w, t_cur));
    acc_w = __fadd_rn(acc_w, w);
    if (weights_out && valid) weights_out[ray * S + k] = w;
  };
  for (int c0 = 0; c0 < S; c0 += kCompChunk) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kCompChunk; ++i) {                 // 64 rays x 16 samples of float4
      const int e = lane + 64 * i, rr = e / kCompChunk, ss = e % kCompChunk;
      long long rg = ray0 + rr;
      if (rg >= n_rays) rg = n_rays - 1;
      s_raw[rr * (kCompChunk + 1) + ss] = r4[rg * S + c0 + ss];
    }
#pragma unroll
    for (int i = 0; i < kCompChunk / 4; ++i) {             // 64 rays x 16 depths
      const int e = lane + 64 * i, rr = e / 4, q = e % 4;
      long long rg = ray0 + rr;
      if (rg >= n_rays) rg = n_rays - 1;
      const f32x4 tv = *reinterpret_cast<const f32x4*>(tvals + rg * t_ray_stride + c0 + 4 * q);
      float* dst = s_t + rr * (kCompChunk + 1) + 4 * q;
      dst[0] = tv.x; dst[1] = tv.y; dst[2] = tv.z; dst[3] = tv.w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kCompChunk; ++k) {
      const f32x4 v = s_raw[lane * (kCompChunk + 1) + k];
      const float t = s_t[lane * (kCompChunk + 1) + k];
      if (c0 + k > 0) emit(pv, pt, __fsub_rn(t, pt), c0 + k - 1);
      pv = v; pt = t;
    }
  }
  emit(pv, pt, 1e10f, S - 1);
  if (!valid) return;
  if (white_bkgd) {
    const float bg = __fsub_rn(1.0f, acc_w);
    acc_r = __fadd_rn(acc_r, bg); acc_g = __fadd_rn(acc_g, bg); acc_b = __fadd_rn(acc_b, bg);
  }
  rgb_out[ray * 3 + 0] = acc_r; rgb_out[ray * 3 + 1] = acc_g; rgb_out[ray * 3 + 2] = acc_b;
  depth_out[ray] = acc_d;
}

// ------------------------------------------------------------------------------------ training: backward of compositing / sampling
// One thread per ray, reverse sequential scans: the exact adjoint of the forward loops above
// (autograd of volume_renderer.py:67-96 and :414-432).
//   w_k = T_k a_k, T_k = prod_{j<k} q_j, q = clamp(1-a, 1e-10, 1), a = 1 - exp(-relu(s) delta), delta_k = t_{k+1}-t_k
//   rgb = sum w_k sigmoid(r_k) + (1 - sum w_k) [white], depth = sum w_k t_k
// Given g_rgb [n,3], g_depth [n] -> g_raw [n,S,4], g_t [n,S] (direct dependence through delta and depth).
// One WAVE per ray (a training step has 4096 rays: with one thread per ray the 2 x 192 sequential steps of 64 lonely waves were
// 108 us of dependent instruction latency): lane l owns samples l, l + 64, l + 128, the transmittance is a wave-level
// exclusive prefix product and the suffix sum a wave-level exclusive suffix sum, chunk by chunk with a scalar carry; every
// load and store is one coalesced row segment.  (The scans associate differently from the forward's sequential product:
// rounding-level differences in T, as between any two summation orders.)
constexpr int kCbMaxSamples = 192;
__global__ __launch_bounds__(256)
void nerf_composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ tvals, long long t_ray_stride,
                               long long n_rays, int S, int white_bkgd, const float* __restrict__ g_rgb,
                               const float* __restrict__ g_depth, float* __restrict__ g_raw, float* __restrict__ g_t) {
  constexpr int C = kCbMaxSamples / 64;
  const int lane = threadIdx.x & 63;
  const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;                                   // (wave-uniform)
  const f32x4* r4 = reinterpret_cast<const f32x4*>(raw) + ray * S;
  f32x4* g4 = reinterpret_cast<f32x4*>(g_raw) + ray * S;
  const float* t = tvals + ray * t_ray_stride;
  float* gt = g_t ? g_t + ray * S : nullptr;
  const float gr = g_rgb[ray * 3 + 0], gg = g_rgb[ray * 3 + 1], gb = g_rgb[ray * 3 + 2];
  const float gd = g_depth ? g_depth[ray] : 0.0f;
  const float wb = white_bkgd ? 1.0f : 0.0f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 v[C];
  float tk[C], delta[C], e[C], alpha[C], om[C], q[C], Tk[C], g_delta[C], gt_own[C];
  // forward sweep: T_k = prod_{j<k} q_j
  float carry = 1.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = 64 * c + lane;
    const bool live = k < S;
    v[c] = live ? r4[k] : zero4;
    tk[c] = live ? t[k] : 0.0f;
    const float tn = (k + 1 < S) ? t[k + 1] : tk[c];
    delta[c] = (k < S - 1) ? __fsub_rn(tn, tk[c]) : 1e10f;
    const float sig = fmaxf(v[c].w, 0.0f);
    e[c] = expf(__fmul_rn(-sig, delta[c]));
    alpha[c] = __fsub_rn(1.0f, e[c]);
    om[c] = __fsub_rn(1.0f, alpha[c]);
    q[c] = live ? fminf(fmaxf(om[c], 1e-10f), 1.0f) : 1.0f;
    float p = q[c];                                            // inclusive prefix product over the 64 lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o = __shfl_up(p, d);
      if (lane >= d) p *= o;
    }
    float excl = __shfl_up(p, 1);
    if (lane == 0) excl = 1.0f;
    Tk[c] = carry * excl;
    carry *= __shfl(p, 63);
  }
  // reverse sweep: suf_k = sum_{m>k} g_w_m a_m T_m;  g_q_k = suf_k / q_k
  float carry_s = 0.0f;
#pragma unroll
  for (int c = C - 1; c >= 0; --c) {
    const int k = 64 * c + lane;
    const bool live = k < S;
    const float sig = fmaxf(v[c].w, 0.0f);
    const float cr = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-v[c].x)));
    const float cg = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-v[c].y)));
    const float cb = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-v[c].z)));
    const float g_w = gr * (cr - wb) + gg * (cg - wb) + gb * (cb - wb) + gd * tk[c];
    const float w = Tk[c] * alpha[c];
    float sfx = live ? g_w * alpha[c] * Tk[c] : 0.0f;           // inclusive suffix sum over the 64 lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o = __shfl_down(sfx, d);
      if (lane + d < 64) sfx += o;
    }
    float excl = __shfl_down(sfx, 1);
    if (lane == 63) excl = 0.0f;
    const float suf = excl + carry_s;
    carry_s += __shfl(sfx, 0);
    float g_alpha = g_w * Tk[c];
    if (om[c] >= 1e-10f && om[c] <= 1.0f) g_alpha -= suf / q[c];      // clamp passes the gradient inside its range
    const float g_sig = g_alpha * delta[c] * e[c];
    g_delta[c] = (k < S - 1) ? g_alpha * sig * e[c] : 0.0f;
    f32x4 go;
    go.x = gr * w * cr * (1.0f - cr);
    go.y = gg * w * cg * (1.0f - cg);
    go.z = gb * w * cb * (1.0f - cb);
    go.w = v[c].w > 0.0f ? g_sig : 0.0f;
    if (live) g4[k] = go;
    gt_own[c] = gd * w - g_delta[c];                            // delta_k = t_{k+1} - t_k: - g_delta_k here, + g_delta_{k-1} below
  }
  if (gt) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = 64 * c + lane;
      float prev = __shfl_up(g_delta[c], 1);
      const float last_of_prev_chunk = c > 0 ? __shfl(g_delta[c > 0 ? c - 1 : 0], 63) : 0.0f;
      if (lane == 0) prev = last_of_prev_chunk;
      if (k < S) gt[k] = gt_own[c] + prev;
    }
  }
}

// Backward of hierarchical sampling (volume_renderer.py:126-154, :247-264 under autograd): gradient of the
// merged depths w.r.t. the coarse densities (the coarse table and u are constants).  Recomputes the
// forward quantities, then: g_t_fine -> g_cdf[below/above] -> g_pdf (reverse cumsum) -> g_(w+eps) ->
// g_w (inner 62) -> g_sigma through T/alpha -> g_raw_coarse[..., 3] (relu mask).
// One WAVE per ray, lane i = coarse sample i (S = 64), two fine samples per lane.  What decides where a fine sample falls --
// transmittance, the running sum of the weights and the cdf -- is recomputed in the forward's own SEQUENTIAL order (a
// v_readlane walk over the lanes with the forward's rounding intrinsics: bit-identical bins); everything downstream of the
// decisions is a gradient and uses wave scans / LDS atomics.  (One thread per ray took 131 us for the 4096 rays of a step:
// 64 lonely waves whose data-dependent while-loops also ran in lockstep.)
__device__ __forceinline__ float lane_bcast(float v, int lane_const) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_const));
}
__global__ __launch_bounds__(256)
void nerf_sample_bwd_kernel(const float* __restrict__ raw_c, const float* __restrict__ t_coarse,
                            const float* __restrict__ u_tab, long long n_rays, const float* __restrict__ t_sorted,
                            const float* __restrict__ g_tsorted, float* __restrict__ g_raw_c) {
  constexpr int S = NERF_N_SAMPLES, F = NERF_N_IMPORTANCE, NB = S - 1;
  static_assert(S == 64 && F == 128, "lane = coarse sample, two fine samples per lane");
  __shared__ float s_tc[S];
  __shared__ float s_u[F];
  __shared__ float s_cdf[4][S];
  __shared__ double s_gcdf[4][S];          // adjoint of the cdf, accumulated in float64 (see below)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x < S) s_tc[threadIdx.x] = t_coarse[threadIdx.x];
  if (threadIdx.x < F) s_u[threadIdx.x] = u_tab[threadIdx.x];
  s_gcdf[wv][lane] = 0.0;
  __syncthreads();
  const long long ray_slot = (long long)blockIdx.x * 4 + wv;
  const bool ray_ok = ray_slot < n_rays;                       // a wave past the end recomputes the last ray and stores nothing
  const long long ray = ray_ok ? ray_slot : n_rays - 1;        // (it has to reach the two workgroup barriers below)
  (void)t_sorted;
  const int i = lane;
  // ---- forward recompute, in the forward's order and rounding (nerf_sample_fine_kernel)
  const float s_raw = raw_c[ray * (S * 4) + i * 4 + 3];
  const float sigma = fmaxf(s_raw, 0.0f);
  const float delta = (i < S - 1) ? __fsub_rn(s_tc[i + 1], s_tc[i]) : 1e10f;
  const float e = expf(__fmul_rn(-sigma, delta));
  const float alpha = __fsub_rn(1.0f, e);                      // = alpha_of(sigma, delta)
  const float om = __fsub_rn(1.0f, alpha);
  const float q = fminf(fmaxf(om, 1e-10f), 1.0f);
  float Ti = 1.0f;
  {
    float T = 1.0f;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      if (lane == j) Ti = T;
      T = __fmul_rn(T, lane_bcast(q, j));
    }
  }
  const bool inner = i >= 1 && i <= S - 2;
  const float we = __fadd_rn(__fmul_rn(Ti, alpha), 1e-5f);     // w + eps (used for the inner 62 only)
  const float wsum = torch_sum62_of([&](int k) { return lane_bcast(we, k + 1); });     // bin k <-> lane k + 1; the forward's order
  const float pdf = __fdiv_rn(we, wsum);                       // lane i: pdf of bin i - 1
  float cdf_m = 0.0f;                                          // lane m: cdf[m], m = 0..62
  {
    float run = 0.0f;
#pragma unroll
    for (int m = 1; m < NB; ++m) {
      run = __fadd_rn(run, lane_bcast(pdf, m));
      if (lane == m) cdf_m = run;
    }
  }
  s_cdf[wv][lane] = cdf_m;                                     // (lane 63: unused slot)
  __syncthreads();
  // ---- the two fine samples of this lane: bin, merged slot, adjoint into g_cdf
  const float* gts = g_tsorted + ray * (S + F);
  float* cdf = s_cdf[wv];
  // Everything downstream of the (fp32, forward-identical) decisions is accumulated in float64: the terms num / denom^2 of an
  // ill-conditioned ray (denom ~ 1e-5) are 1e5 x the result of the sums they enter, and the fp32 version of these sums (LDS float
  // atomics + wave scans) was 10-20 x further from a float64 evaluation of the adjoint than torch's fp32 autograd on the same
  // inputs (tests/test_gpu_train_steps.py, round 3).  The kernel is 20 us of a training step either way.
  double* gcdf = s_gcdf[wv];
  int ic_carry = 0;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int k = lane + 64 * half;
    const float u = s_u[k];
    int ind = 0;
    for (int m = 0; m < NB; ++m) ind += cdf[m] <= u;            // cdf is non-decreasing: = the forward's searchsorted(right)
    const int below = min(max(ind - 1, 0), S - 3), above = min(ind, S - 3);
    const float cb = cdf[below], ca = cdf[above];
    const float bb = __fmul_rn(0.5f, __fadd_rn(s_tc[below + 1], s_tc[below]));
    const float ba = __fmul_rn(0.5f, __fadd_rn(s_tc[above + 1], s_tc[above]));
    const float draw_ = __fsub_rn(ca, cb);
    const bool live = !(draw_ < 1e-5f);
    const float denom = live ? draw_ : 1.0f;
    const float num = __fsub_rn(u, cb);
    const float v = __fadd_rn(bb, __fmul_rn(__fdiv_rn(num, denom), __fsub_rn(ba, bb)));
    int ic = 0;                                                 // coarse entries sorted before this fine sample (ties: coarse first)
    for (int j = 0; j < S; ++j) ic += s_tc[j] <= v;
    // the forward's merge pointer never moves back: running maximum over the fine samples in order
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(ic, d);
      if (lane >= d) ic = max(ic, o);
    }
    ic = max(ic, ic_carry);
    ic_carry = __shfl(ic, 63);
    const double g = (double)gts[k + ic];                       // its slot in the merged array
    const double g_frac = g * (double)__fsub_rn(ba, bb);
    double g_cb = -g_frac / (double)denom, g_ca = 0.0;          // frac = (u - cb) / denom, denom = ca - cb when live
    if (live) { const double gden = -g_frac * (double)num / ((double)denom * (double)denom); g_ca += gden; g_cb -= gden; }
    atomicAdd(&gcdf[below], g_cb);
    atomicAdd(&gcdf[above], g_ca);
  }
  __syncthreads();
  // ---- cdf[m] = sum_{j<m} pdf_j  ->  g_pdf_j = sum_{m>j} g_cdf[m];  pdf = we / W
  double sfx = lane < NB ? gcdf[lane] : 0.0;                    // inclusive suffix sum over the lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_down(sfx, d);
    if (lane + d < 64) sfx += o;
  }
  double gpdf_excl = __shfl_down(sfx, 1);                       // lane j: g_pdf_j = sum_{m>j} g_cdf[m]
  if (lane == 63) gpdf_excl = 0.0;
  double gpdf_i = __shfl_up(gpdf_excl, 1);                      // lane i: g_pdf of bin i - 1 (the bin of sample i)
  if (!inner) gpdf_i = 0.0;
  double dot = inner ? gpdf_i * (double)we : 0.0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d);
  const double g_w = inner ? (gpdf_i / (double)wsum - dot / ((double)wsum * (double)wsum)) : 0.0;
  double sf2 = g_w * (double)alpha * (double)Ti;                // suf_i = sum_{m>i} g_w_m a_m T_m
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_down(sf2, d);
    if (lane + d < 64) sf2 += o;
  }
  double suf = __shfl_down(sf2, 1);
  if (lane == 63) suf = 0.0;
  double g_alpha = g_w * (double)Ti;
  if (om >= 1e-10f && om <= 1.0f) g_alpha -= suf / (double)q;
  f32x4 go = {0.f, 0.f, 0.f, 0.f};
  go.w = s_raw > 0.0f ? (float)(g_alpha * (double)delta * (double)e) : 0.0f;
  if (ray_ok) reinterpret_cast<f32x4*>(g_raw_c)[ray * S + i] = go;
}

// ------------------------------------------------------------------------------------ training: live tiles of a backward pass
// d loss / d raw is exactly zero wherever relu(sigma) = 0 (alpha = 0, weight 0: nerf_composite_bwd_kernel writes zeros there),
// and wherever the coarse density does not move any fine sample.  A 32-point tile (the unit of the chain kernel) whose
// incoming gradient is zero at all its points produces zero g_z rows, adds exactly nothing to any weight gradient and has
// zero g_t / g_x: it is dropped from the chain launch and from every weight-gradient launch.  Exact, not an approximation
// (a zero contribution to a floating-point sum leaves it unchanged); how many tiles are dead is scene-dependent (the
// synthetic bench scene: 23 % of the fine and 36 % of the coarse tiles; a trained scene: most of empty space).
__global__ __launch_bounds__(256)
void nerf_tile_flags_kernel(const f32x4* __restrict__ draw, long long P, int density_only, int* __restrict__ flags) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  bool nz = false;
  if (p < P) {
    const f32x4 g = draw[p];
    nz = density_only ? (g.w != 0.0f) : (g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f);    // (-0 is zero, NaN is not)
  }
  const unsigned long long b = __builtin_amdgcn_ballot_w64(nz);
  const int lane = threadIdx.x & 63;
  const long long tile = p >> 5;
  if ((lane & 31) == 0 && (p & ~31LL) < P) flags[tile] = (lane ? (b >> 32) : (b & 0xffffffffull)) != 0ull;
}
// flags -> ascending list of live tiles + count; one workgroup (the list is a few thousand entries per pass)
__global__ __launch_bounds__(1024)
void nerf_tile_scan_kernel(const int* __restrict__ flags, int n_tiles, int* __restrict__ live, int* __restrict__ count) {
  __shared__ int s_cnt[1024];
  const int tid = threadIdx.x;
  const int per = (n_tiles + 1023) / 1024;
  const int t0 = tid * per, t1 = min(t0 + per, n_tiles);
  int c = 0;
  for (int t = t0; t < t1; ++t) c += flags[t] != 0;
  s_cnt[tid] = c;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {                 // inclusive Hillis-Steele scan
    const int v = tid >= d ? s_cnt[tid - d] : 0;
    __syncthreads();
    s_cnt[tid] += v;
    __syncthreads();
  }
  int pos = s_cnt[tid] - c;
  for (int t = t0; t < t1; ++t)
    if (flags[t] != 0) live[pos++] = t;
  if (tid == 1023) *count = s_cnt[1023];
}

// ------------------------------------------------------------------------------------ fused clip + Adam (section 8f-4)
// One launch over all 48 parameter tensors: clip_grad_value_ (trainer.py:59) + torch.optim.Adam's update
// (optimizer.py:21-24: Adam(lr, weight_decay, eps), betas (0.9, 0.999), no amsgrad) in torch's operation
// order (lerp for exp_avg, sqrt(v)/sqrt(bc2) + eps for the denominator).
constexpr int kAdamMaxTensors = 48;
struct AdamArgs {
  float* p[kAdamMaxTensors];
  const float* g[kAdamMaxTensors];
  float* m[kAdamMaxTensors];
  float* v[kAdamMaxTensors];
  long long end[kAdamMaxTensors];      // exclusive prefix ends of the flattened index space
  int n_tensors;
  float lr, beta1, beta2, eps, weight_decay, clip, bc1, bc2_sqrt;
};
__global__ __launch_bounds__(256)
void nerf_adam_kernel(AdamArgs a) {
  const long long total = a.end[a.n_tensors - 1];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = a.n_tensors - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (i < a.end[mid]) hi = mid; else lo = mid + 1; }
    const long long j = i - (lo ? a.end[lo - 1] : 0);
    float g = a.g[lo][j];
    if (a.clip > 0.0f) g = fminf(fmaxf(g, -a.clip), a.clip);
    const float p = a.p[lo][j];
    if (a.weight_decay != 0.0f) g = __fadd_rn(g, __fmul_rn(a.weight_decay, p));
    float m = a.m[lo][j], v = a.v[lo][j];
    m = __fadd_rn(m, __fmul_rn(__fsub_rn(g, m), 1.0f - a.beta1));                       // exp_avg.lerp_(grad, 1 - beta1)
    v = __fadd_rn(__fmul_rn(v, a.beta2), __fmul_rn(__fmul_rn(g, g), 1.0f - a.beta2));   // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), a.bc2_sqrt), a.eps);
    a.m[lo][j] = m; a.v[lo][j] = v;
    a.p[lo][j] = __fadd_rn(p, __fmul_rn(-(a.lr / a.bc1), __fdiv_rn(m, denom)));         // addcdiv_(m, denom, value=-step_size)
  }
}

// ------------------------------------------------------------------------------------ ray generation
// Pinhole rays of src/datasets/nerf/blender.py:102-127 in float64 like numpy, cast to float32 at the
// end (:149-151).  One thread per pixel; removes the 15.4 MB/frame host->device copy of run.py:167-169.
struct RayGenArgs {
  double c2w[12];          // row-major 3x4 camera-to-world
  double focal, cx, cy;
  long long pixel_begin, n_pixels;
  int W;
  const long long* pixel_ids;   // optional explicit pixel list (training batches), else pixel_begin + i
  float* rays_o;
  float* rays_d;
};
__global__ void nerf_generate_rays_kernel(RayGenArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_pixels) return;
  const long long id = a.pixel_ids ? a.pixel_ids[i] : a.pixel_begin + i;
  const double u = (double)(id % a.W), v = (double)(id / a.W);
  const double dx = (u - a.cx) / a.focal, dy = -(v - a.cy) / a.focal, dz = -1.0;
  double w[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) w[r] = (a.c2w[4 * r + 0] * dx + a.c2w[4 * r + 1] * dy) + a.c2w[4 * r + 2] * dz;
  const double nrm = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    a.rays_d[i * 3 + r] = (float)(w[r] / nrm);
    a.rays_o[i * 3 + r] = (float)a.c2w[4 * r + 3];
  }
}

// ------------------------------------------------------------------------------------ image metrics
// Sums for src/evaluators/nerf.py: float MSE of the clipped images (:96-100) and the evaluator's
// psnr_metric (:23-30), whose uint8 subtraction AND squaring wrap modulo 256.
__global__ __launch_bounds__(256)
void nerf_image_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt, long long n_values,
                               double* __restrict__ out /*[2]: sum sq float, sum wrapped-u8 sq*/) {
  double s_f = 0.0, s_u = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += (long long)gridDim.x * blockDim.x) {
    const float p = fminf(fmaxf(pred[i], 0.0f), 1.0f), g = fminf(fmaxf(gt[i], 0.0f), 1.0f);
    const float d = p - g;
    s_f += (double)(d * d);
    const unsigned pu = (unsigned)(p * 255.0f) & 0xffu, gu = (unsigned)(g * 255.0f) & 0xffu;   // astype(uint8): truncation
    const unsigned du = (pu - gu) & 0xffu;
    s_u += (double)((du * du) & 0xffu);
  }
  __shared__ double red[2][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s_f += __shfl_xor(s_f, o); s_u += __shfl_xor(s_u, o); }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = s_f; red[1][wave] = s_u; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&out[0], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    atomicAdd(&out[1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

// SSIM of src/evaluators/nerf.py:49-77: skimage.metrics.structural_similarity(pred_u8, gt_u8, win_size=7,
// channel_axis=2) = Wang et al. with a 7x7 uniform window, sample covariance (49/48), K1 .01, K2 .03,
// data_range 255, mean over the (H-6)x(W-6) interior and the 3 channels; float64 like skimage.
__global__ __launch_bounds__(256)
void nerf_ssim_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int H, int W, double* __restrict__ out) {
  const long long n_int = (long long)(H - 6) * (W - 6) * 3;
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_int; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    const long long px = i / 3;
    const int x = (int)(px % (W - 6)) + 3, y = (int)(px / (W - 6)) + 3;
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int dy = -3; dy <= 3; ++dy)
      for (int dx = -3; dx <= 3; ++dx) {
        const long long q = ((long long)(y + dy) * W + (x + dx)) * 3 + c;
        const double a = (double)((unsigned)(fminf(fmaxf(pred[q], 0.f), 1.f) * 255.0f) & 0xffu);
        const double b = (double)((unsigned)(fminf(fmaxf(gt[q], 0.f), 1.f) * 255.0f) & 0xffu);
        sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
      }
    const double ux = sx / 49.0, uy = sy / 49.0, cn = 49.0 / 48.0;
    const double vx = cn * (sxx / 49.0 - ux * ux), vy = cn * (syy / 49.0 - uy * uy), vxy = cn * (sxy / 49.0 - ux * uy);
    const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
    s += ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// ------------------------------------------------------------------------------------ d loss / d viewdirs (Network.forward under autograd)
// The view direction of a ray enters views_linears.0 through its 27-channel encoding (network.py:224-225, :63-66) and is
// shared by the ray's samples: g_dir_enc = W_v[:, 256:283]^T (sum_s g_zv[ray, s]), then the chain rule through
// [d, sin(2^k d), cos(2^k d)] (freq.py:31-32).  One 128-thread workgroup per ray; 3.5 k MACs per ray -- negligible.
__global__ __launch_bounds__(128)
void nerf_viewdirs_bwd_kernel(const float* __restrict__ gzv, long long n_rays, int n_samples, const float* __restrict__ w_views,
                              const float* __restrict__ viewdirs, float* __restrict__ g_viewdirs) {
  __shared__ float G[128];
  __shared__ float E[27];
  const long long ray = blockIdx.x;
  const int o = threadIdx.x;
  float acc = 0.0f;
  for (int s = 0; s < n_samples; ++s) acc += gzv[(ray * n_samples + s) * 128 + o];
  G[o] = acc;
  __syncthreads();
  if (o < 27) {
    float e = 0.0f;
    for (int k = 0; k < 128; ++k) e = fmaf(G[k], w_views[k * 283 + 256 + o], e);
    E[o] = e;
  }
  __syncthreads();
  if (o < 3) {
    const float d = viewdirs[ray * 3 + o];
    float g = E[o];
    for (int k = 0; k < 4; ++k) {
      const float f = (float)(1 << k);
      float sv, cv;
      sincosf(d * f, &sv, &cv);
      g += f * (cv * E[3 + 6 * k + o] - sv * E[3 + 6 * k + 3 + o]);
    }
    g_viewdirs[ray * 3 + o] = g;
  }
}

// ------------------------------------------------------------------------------------ dead-tile skipping: ONE decision for both passes
// The SAVE forward for compositing drops the rows of density-free tiles, and the backward pass drops tiles without an incoming
// gradient: the second is only safe to switch off if the first was off too (a dense backward would read rows that were never
// written).  Both entry points therefore ask this one helper, and the forward stamps what it did into the save buffer
// (TrainSave::off_stamp): a dense backward on a buffer stamped "rows skipped" poisons the alpha-bias gradient with NaN instead
// of silently multiplying garbage by zero (the host cannot read the stamp without a synchronisation the ABI does not make).
bool dead_tile_list_available(long long P, int precision) {
  const char* env = getenv("NERF_DEAD_TILE_SKIP");
  const bool want = !(env && env[0] == '0');
  return want && (precision == NERF_PREC_F32 || precision == NERF_PREC_F32X) && NERF_WGRAD_ASM && NERF_WGVEC_ASM &&
         P % 32 == 0 && P / 32 <= 0x7fffffffLL;
}
__global__ void nerf_check_stamp_kernel(const float* __restrict__ stamp, float* __restrict__ poison) {
  if (threadIdx.x == 0 && __float_as_int(stamp[0]) != 0) poison[0] = __int_as_float(0x7fc00000);
}

// ------------------------------------------------------------------------------------ launch helpers
int num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    cus = v;
  }
  return cus;
}

int launch_mlp(const MlpArgs& a, bool ray_mode, int precision, hipStream_t st) {
  if (precision != NERF_PREC_F32 && precision != NERF_PREC_F16 && precision != NERF_PREC_F32X && precision != NERF_PREC_F16S)
    return fail(NERF_ERR_UNSUPPORTED, "%s", "precision not built");
  if (a.n_points <= 0) return NERF_OK;
  if (a.index && a.n_points > 0x7fffffffLL) return fail(NERF_ERR_INVALID_ARG, "%s", "index mode: point ids are int32");
  if (precision == NERF_PREC_F32X) {       // persistent workgroups of 4 waves, one per CU
    const long long n_tiles = (a.n_points + kXTilePts - 1) / kXTilePts;
    const unsigned blocks = (unsigned)(n_tiles < num_cus() ? n_tiles : num_cus());
    if (ray_mode && a.density_only) hipLaunchKernelGGL((nerf_mlp_f32x_kernel<true, false, true>), dim3(blocks), dim3(kXThreads), 0, st, a);
    else if (ray_mode && a.skip_dead_colour) hipLaunchKernelGGL((nerf_mlp_f32x_kernel<true, false, false, true>), dim3(blocks), dim3(kXThreads), 0, st, a);
    else if (ray_mode) hipLaunchKernelGGL(nerf_mlp_f32x_kernel<true>, dim3(blocks), dim3(kXThreads), 0, st, a);
    else hipLaunchKernelGGL(nerf_mlp_f32x_kernel<false>, dim3(blocks), dim3(kXThreads), 0, st, a);
    return check_launch("nerf_mlp_f32x_kernel");
  }
  if (precision == NERF_PREC_F16S) {       // the fp16 path on 16x16x32 MFMA tiles: same workgroup shape and LDS ring
    const long long n_tiles = (a.n_points + kF16TilePts - 1) / kF16TilePts;
    const unsigned blocks = (unsigned)(n_tiles < num_cus() ? n_tiles : num_cus());
    if (ray_mode && a.density_only) hipLaunchKernelGGL((nerf_mlp_f16s_kernel<true, true>), dim3(blocks), dim3(kF16Threads), 0, st, a);
    else if (ray_mode && a.skip_dead_colour) hipLaunchKernelGGL((nerf_mlp_f16s_kernel<true, false, true>), dim3(blocks), dim3(kF16Threads), 0, st, a);
    else if (ray_mode) hipLaunchKernelGGL(nerf_mlp_f16s_kernel<true>, dim3(blocks), dim3(kF16Threads), 0, st, a);
    else hipLaunchKernelGGL(nerf_mlp_f16s_kernel<false>, dim3(blocks), dim3(kF16Threads), 0, st, a);
    return check_launch("nerf_mlp_f16s_kernel");
  }
  if (precision == NERF_PREC_F16) {        // persistent workgroups, one per CU (147 KB of LDS each)
    const long long n_tiles = (a.n_points + kF16TilePts - 1) / kF16TilePts;
    const unsigned blocks = (unsigned)(n_tiles < num_cus() ? n_tiles : num_cus());
    if (ray_mode && a.density_only) hipLaunchKernelGGL((nerf_mlp_f16_kernel<true, true>), dim3(blocks), dim3(kF16Threads), 0, st, a);
    else if (ray_mode && a.skip_dead_colour) hipLaunchKernelGGL((nerf_mlp_f16_kernel<true, false, true>), dim3(blocks), dim3(kF16Threads), 0, st, a);
    else if (ray_mode) hipLaunchKernelGGL(nerf_mlp_f16_kernel<true>, dim3(blocks), dim3(kF16Threads), 0, st, a);
    else hipLaunchKernelGGL(nerf_mlp_f16_kernel<false>, dim3(blocks), dim3(kF16Threads), 0, st, a);
    return check_launch("nerf_mlp_f16_kernel");
  }
#ifndef NERF_F32_WG_WAVES
#define NERF_F32_WG_WAVES 1                 // waves per workgroup of the (barrier-free) inference instance: with one-wave
                                            // workgroups every SIMD is refilled on its own (+0.9 % over 4-wave workgroups, A/B)
#endif
  const long long tiles = (a.n_points + nerf::kTilePts - 1) / nerf::kTilePts;
  long long blocks = (tiles + NERF_F32_WG_WAVES - 1) / NERF_F32_WG_WAVES;
  if (blocks > 0x7fffffffLL) return fail(NERF_ERR_INVALID_ARG, "%s", "too many points for one launch");
  // density_only (ray mode): every precision has an instance that stops after the sigma head
  if (ray_mode && a.density_only) hipLaunchKernelGGL((nerf_mlp_f32_kernel<true, false, true>), dim3((unsigned)blocks), dim3(64 * NERF_F32_WG_WAVES), 0, st, a);
  else if (ray_mode && a.skip_dead_colour)
    hipLaunchKernelGGL((nerf_mlp_f32_kernel<true, false, false, true>), dim3((unsigned)blocks), dim3(64 * NERF_F32_WG_WAVES), 0, st, a);
  else if (ray_mode) hipLaunchKernelGGL(nerf_mlp_f32_kernel<true>, dim3((unsigned)blocks), dim3(64 * NERF_F32_WG_WAVES), 0, st, a);
  else hipLaunchKernelGGL(nerf_mlp_f32_kernel<false>, dim3((unsigned)blocks), dim3(64 * NERF_F32_WG_WAVES), 0, st, a);
  return check_launch("nerf_mlp_f32_kernel");
}

inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

}  // namespace

// ===================================================================================== C ABI
extern "C" {

int32_t nerf_abi_version(void) { return NERF_ABI_VERSION; }

int32_t nerf_build_flags(void) {
  int32_t f = 0;
#ifdef NERF_TIMING_BUILD
  f |= NERF_BUILD_TIMING;
#endif
#if NERF_ANY_TIMING_HACK || NERF_SAVE_TAPS == 0 || 2 == 0
  f |= NERF_BUILD_WRONG_NUMERICS;
#endif
  return f;
}
const char* nerf_last_error(void) { return g_err; }
int64_t nerf_packed_model_bytes(int32_t precision) {
  if (precision == NERF_PREC_F32) return nerf::kPackedFloats * (int64_t)sizeof(float);
  // the fp16 streams carry the split-fp16 ("f32x") stream of the same model behind them: nerf_render_forward re-evaluates the
  // last sample of every ray with it (far-plane guard)
  if (precision == NERF_PREC_F16 || precision == NERF_PREC_F16S) return nerf::kF16PackedBytes + nerf::kXPackedBytes;
  if (precision == NERF_PREC_F32X) return nerf::kXPackedBytes;
  return -1;
}

int32_t nerf_pack_model(const float* const params[24], void* packed, int32_t precision, void* stream) {
  if (!params || !packed) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_pack_model: null argument");
  PackArgs a;
  for (int i = 0; i < nerf::P_COUNT; ++i) {
    if (!params[i]) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_pack_model: null parameter pointer");
    a.p[i] = params[i];
  }
  a.out = (float*)packed;
  const int threads = 256;
  if (precision == NERF_PREC_F16 || precision == NERF_PREC_F32X || precision == NERF_PREC_F16S) {
    const long long n = nerf::kF16ConstBytes / 4 + (long long)nerf::kF16Frags * 512;
    const dim3 grid((unsigned)((n + threads - 1) / threads));
    if (precision == NERF_PREC_F16S) hipLaunchKernelGGL(nerf_pack_f16s_kernel, grid, dim3(threads), 0, (hipStream_t)stream, a);
    else if (precision == NERF_PREC_F16) hipLaunchKernelGGL(nerf_pack_f16_kernel<false>, grid, dim3(threads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(nerf_pack_f16_kernel<true>, grid, dim3(threads), 0, (hipStream_t)stream, a);
    if (precision != NERF_PREC_F32X) {          // + the f32x stream behind the fp16 one
      PackArgs ax = a;
      ax.out = reinterpret_cast<float*>(reinterpret_cast<char*>(packed) + nerf::kF16PackedBytes);
      hipLaunchKernelGGL(nerf_pack_f16_kernel<true>, grid, dim3(threads), 0, (hipStream_t)stream, ax);
    }
    return check_launch("nerf_pack_f16_kernel");
  }
  if (precision != NERF_PREC_F32) return fail(NERF_ERR_UNSUPPORTED, "%s", "nerf_pack_model: unknown precision");
  const unsigned blocks = (unsigned)((nerf::kPackedFloats + threads - 1) / threads);
  hipLaunchKernelGGL(nerf_pack_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, a);
  return check_launch("nerf_pack_kernel");
}

int32_t nerf_positional_encoding(const float* x, int64_t n, int32_t n_freqs, float* out, void* stream) {
  if (n < 0 || n_freqs < 0 || n_freqs > 16) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_positional_encoding: bad size");
  if (n == 0) return NERF_OK;
  if (!x || !out) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_positional_encoding: null argument");
  const unsigned blocks = (unsigned)((n * 3 + 255) / 256);
  hipLaunchKernelGGL(nerf_pe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n, n_freqs, out);
  return check_launch("nerf_pe_kernel");
}

int32_t nerf_mlp_forward(const float* pts, const float* viewdirs, int64_t n_rays, int32_t n_samples,
                         const void* packed, float* raw, int32_t precision, void* stream) {
  if (n_rays < 0 || n_samples <= 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!pts || !viewdirs || !packed || !raw) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward: null argument");
  MlpArgs a{};
  a.pts = pts; a.viewdirs = viewdirs; a.n_points = n_rays * n_samples; a.n_samples = n_samples;
  a.packed = (const float*)packed; a.raw = raw;
  return launch_mlp(a, false, precision, (hipStream_t)stream);
}

int32_t nerf_mlp_forward_rays(const float* rays_o, const float* rays_d, const float* tvals,
                              int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                              const void* packed, float* raw, int32_t precision, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || t_ray_stride < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_rays: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!rays_o || !rays_d || !tvals || !packed || !raw) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_rays: null argument");
  MlpArgs a{};
  a.rays_o = rays_o; a.rays_d = rays_d; a.tvals = tvals; a.t_ray_stride = t_ray_stride;
  a.n_points = n_rays * n_samples; a.n_samples = n_samples; a.packed = (const float*)packed; a.raw = raw;
  return launch_mlp(a, true, precision, (hipStream_t)stream);
}

int32_t nerf_mlp_forward_rays_for_compositing(const float* rays_o, const float* rays_d, const float* tvals,
                                              int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                              const void* packed, float* raw, int32_t precision, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || t_ray_stride < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_rays_for_compositing: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!rays_o || !rays_d || !tvals || !packed || !raw) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_rays_for_compositing: null argument");
  MlpArgs a{};
  a.rays_o = rays_o; a.rays_d = rays_d; a.tvals = tvals; a.t_ray_stride = t_ray_stride;
  a.n_points = n_rays * n_samples; a.n_samples = n_samples; a.packed = (const float*)packed; a.raw = raw;
  a.skip_dead_colour = 1;
  return launch_mlp(a, true, precision, (hipStream_t)stream);
}

int32_t nerf_mlp_forward_rays_density(const float* rays_o, const float* rays_d, const float* tvals,
                                      int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                      const void* packed, float* raw, int32_t precision, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || t_ray_stride < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_rays_density: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!rays_o || !rays_d || !tvals || !packed || !raw) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_rays_density: null argument");
  MlpArgs a{};
  a.rays_o = rays_o; a.rays_d = rays_d; a.tvals = tvals; a.t_ray_stride = t_ray_stride;
  a.n_points = n_rays * n_samples; a.n_samples = n_samples; a.packed = (const float*)packed; a.raw = raw;
  a.density_only = 1;
  return launch_mlp(a, true, precision, (hipStream_t)stream);
}

int32_t nerf_sample_fine(const float* raw_coarse, const float* t_coarse, const float* u,
                         int64_t n_rays, float* t_sorted, float* t_fine, uint8_t* valid_sorted,
                         float weights_threshold, float ert_threshold, void* stream) {
  if (n_rays < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_sample_fine: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!raw_coarse || !t_coarse || !u || !t_sorted) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_sample_fine: null argument");
  SampleArgs a;
  a.raw_c = raw_coarse; a.t_coarse = t_coarse; a.u_tab = u; a.n_rays = n_rays; a.t_sorted = t_sorted;
  a.t_fine = t_fine; a.valid_sorted = valid_sorted; a.fast_sampling = valid_sorted != nullptr;
  a.weights_threshold = weights_threshold; a.ert_threshold = ert_threshold;
  const unsigned blocks = (unsigned)((n_rays + kSampleThreads - 1) / kSampleThreads);
  hipLaunchKernelGGL(nerf_sample_fine_kernel, dim3(blocks), dim3(kSampleThreads), 0, (hipStream_t)stream, a);
  return check_launch("nerf_sample_fine_kernel");
}

int32_t nerf_composite(const float* raw, const float* tvals, int64_t t_ray_stride, int64_t n_rays,
                       int32_t n_samples, int32_t white_bkgd, float* rgb, float* depth,
                       float* weights, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || t_ray_stride < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_composite: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!raw || !tvals || !rgb || !depth) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_composite: null argument");
  if (n_samples % kCompChunk == 0 && (t_ray_stride % 4 == 0) && ((uintptr_t)tvals % 16 == 0)) {
    hipLaunchKernelGGL(nerf_composite_staged_kernel, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       raw, tvals, (long long)t_ray_stride, (long long)n_rays, n_samples, white_bkgd, rgb, depth, weights);
    return check_launch("nerf_composite_staged_kernel");
  }
  const unsigned blocks = (unsigned)((n_rays + 255) / 256);
  hipLaunchKernelGGL(nerf_composite_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, raw, tvals,
                     (long long)t_ray_stride, (long long)n_rays, n_samples, white_bkgd, rgb, depth, weights);
  return check_launch("nerf_composite_kernel");
}

int32_t nerf_generate_rays(const double c2w[12], int32_t H, int32_t W, double focal, int64_t pixel_begin,
                           int64_t n_pixels, const int64_t* pixel_ids, float* rays_o, float* rays_d, void* stream) {
  if (n_pixels < 0 || H <= 0 || W <= 0 || !(focal > 0.0) || pixel_begin < 0 ||
      (!pixel_ids && pixel_begin + n_pixels > (int64_t)H * W))
    return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_generate_rays: bad size");
  if (n_pixels == 0) return NERF_OK;
  if (!c2w || !rays_o || !rays_d) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_generate_rays: null argument");
  RayGenArgs a;
  for (int i = 0; i < 12; ++i) a.c2w[i] = c2w[i];
  a.focal = focal; a.cx = W / 2.0; a.cy = H / 2.0; a.pixel_begin = pixel_begin; a.n_pixels = n_pixels; a.W = W;
  a.pixel_ids = (const long long*)pixel_ids; a.rays_o = rays_o; a.rays_d = rays_d;
  hipLaunchKernelGGL(nerf_generate_rays_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("nerf_generate_rays_kernel");
}

int32_t nerf_image_metrics(const float* pred, const float* gt, int64_t n_values, double* sums2, void* stream) {
  if (n_values < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_image_metrics: bad size");
  if (!sums2) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_image_metrics: null argument");
  if (hipMemsetAsync(sums2, 0, 2 * sizeof(double), (hipStream_t)stream) != hipSuccess)
    return fail(NERF_ERR_HIP, "%s", "nerf_image_metrics: memset failed");
  if (n_values == 0) return NERF_OK;
  if (!pred || !gt) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_image_metrics: null argument");
  long long blocks = (n_values + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(nerf_image_metrics_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, gt,
                     (long long)n_values, sums2);
  return check_launch("nerf_image_metrics_kernel");
}

// live / n_live: the live-tile list of a backward pass (device pointers) or nullptr for the whole point range; with a list,
// only the asm-ring kernels of the small layers apply (the caller checks list_mode_ok) and each workgroup reads *n_live itself
static int32_t wgrad_impl(const float* dz, int64_t ldz, int32_t zc0, int32_t n_out, const float* hin, int64_t ldh,
                          int32_t hc0, int32_t n_in, float* dw, int64_t ldw, int32_t wc0, float* db, int64_t n_points,
                          const int* live, const int* n_live, void* stream) {
  if (n_points < 0 || n_out <= 0 || n_in <= 0 || n_out > 256 || n_in > 256) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_wgrad: bad size");
  if (n_points == 0) return NERF_OK;
  if (!dz || !hin || !dw) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_wgrad: null argument");
  WgradArgs a;
  a.dz = dz; a.ldz = ldz; a.zc0 = zc0; a.n_out = n_out; a.hin = hin; a.ldh = ldh; a.hc0 = hc0; a.n_in = n_in;
  a.dw = dw; a.ldw = ldw; a.wc0 = wc0; a.db = db; a.n_points = n_points; a.live_tiles = live; a.n_live = n_live;
  const int to = (n_out + 31) / 32, ti = (n_in + 31) / 32;
  const long long pairs = (n_points + 1) / 2;
  long long blocks = (pairs + 255) / 256;            // >= 256 point pairs per workgroup
  if (blocks > num_cus()) blocks = num_cus();
  if (blocks < 1) blocks = 1;
  // the small-layer (vector-load) kernels: one workgroup per CU (more resident waves measured no faster)
  long long vblocks = (pairs + 255) / 256;
  if (vblocks > (long long)num_cus()) vblocks = (long long)num_cus();
  if (vblocks < 1) vblocks = 1;
  const dim3 grid((unsigned)blocks), vgrid((unsigned)vblocks), blk(256);
  hipStream_t st = (hipStream_t)stream;
  const bool aligned = ldz % 4 == 0 && ldh % 4 == 0 && zc0 % 4 == 0 && hc0 % 4 == 0 &&
                       (uintptr_t)dz % 16 == 0 && (uintptr_t)hin % 16 == 0;
  if (n_out == 256 && n_in == 256 && aligned && NERF_WGRAD_ASM && n_points % 16 == 0 && n_points / 16 >= blocks &&
      ldz * 8 < (1ll << 31) && ldh * 8 < (1ll << 31)) {
    a.osplit = 2; a.isplit = 2;        // whole groups of 8 k-steps per workgroup: the clamp-free asm-load form
    WgradBatch wb;
    wb.job[0] = a; wb.n_jobs = 1;
    if (live) hipLaunchKernelGGL(nerf_wgrad256_f32_asm_kernel<true>, grid, blk, 0, st, wb);
    else hipLaunchKernelGGL(nerf_wgrad256_f32_asm_kernel<false>, grid, blk, 0, st, wb);
  } else if (live) {
    // list mode needs the asm-ring form of the small-layer kernels (whole 32-point tiles)
    if (!(NERF_WGVEC_ASM && n_points % 32 == 0 && ldz < (1ll << 28) && ldh < (1ll << 28)))
      return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_wgrad: live-tile list on a shape without an asm-ring kernel");
    const dim3 lgrid((unsigned)num_cus());
#define VECL(AV, BV, OS, IS) do { a.osplit = OS; a.isplit = IS; \
      hipLaunchKernelGGL((nerf_wgrad_vec_f32_asm_kernel<AV, BV, 16, true>), lgrid, blk, 0, st, a); } while (0)
    if (n_out == 256 && n_in <= 64 && aligned) VECL(4, 1, 2, 2);
    else if (n_out == 128 && n_in == 256 && aligned) VECL(4, 2, 1, 4);
    else if (n_out == 128 && n_in <= 32) VECL(1, 1, 4, 1);
    else if (n_out <= 32 && n_in == 256 && ldh % 2 == 0 && hc0 % 2 == 0 && (uintptr_t)hin % 8 == 0) VECL(1, 2, 1, 4);
    else if (n_out <= 32 && n_in <= 128) VECL(1, 1, 1, 4);
    else return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_wgrad: live-tile list on a shape without an asm-ring kernel");
#undef VECL
  } else if (n_out == 256 && n_in == 256 && aligned) {
    a.osplit = 2; a.isplit = 2;
    hipLaunchKernelGGL(nerf_wgrad256_f32_kernel, grid, blk, 0, st, a);
  }
  // small layers on the vector-load kernel: (floats per lane, wave split) chosen so that 32*AV*osplit covers n_out
  // and 32*BV*isplit covers n_in; operands must be aligned to their vector width.  VEC picks the asm-ring form when the
  // point range splits into whole groups of PF k-steps per workgroup (every training shape does: P is a multiple of 64)
#define VEC(AV, BV, PF, OS, IS) do { \
    a.osplit = OS; a.isplit = IS; \
    if (NERF_WGVEC_ASM && n_points % (2 * PF) == 0 && n_points / (2 * PF) >= 1 && ldz < (1ll << 28) && ldh < (1ll << 28)) { \
      long long g = n_points / (2 * PF); \
      if (g > num_cus()) g = num_cus(); \
      hipLaunchKernelGGL((nerf_wgrad_vec_f32_asm_kernel<AV, BV, PF>), dim3((unsigned)g), blk, 0, st, a); \
    } else hipLaunchKernelGGL((nerf_wgrad_vec_f32_kernel<AV, BV>), vgrid, blk, 0, st, a); \
  } while (0)
  else if (n_out == 256 && n_in <= 64 && aligned) VEC(4, 1, NERF_WGVEC_PF41, 2, 2);         // PE -> 256 (layers 0 and 5)
  else if (n_out == 128 && n_in == 256 && aligned) VEC(4, 2, 16, 1, 4);        // views_linears.0, feature part
  else if (n_out == 128 && n_in <= 32) VEC(1, 1, 16, 4, 1);                    // views_linears.0, direction part
  else if (n_out <= 32 && n_in == 256 && ldh % 2 == 0 && hc0 % 2 == 0 && (uintptr_t)hin % 8 == 0)
    VEC(1, 2, 16, 1, 4);                                                       // alpha_linear
  else if (n_out <= 32 && n_in <= 128) VEC(1, 1, 16, 1, 4);                    // rgb_linear
#undef VEC
  else if (to > 4 && ti > 4) { a.osplit = 2; a.isplit = 2; hipLaunchKernelGGL((nerf_wgrad_f32_kernel<4, 4>), grid, blk, 0, st, a); }
  else if (to > 4)           { a.osplit = 2; a.isplit = 2; hipLaunchKernelGGL((nerf_wgrad_f32_kernel<4, 1>), grid, blk, 0, st, a); }
  else if (to > 1 && ti > 4) { a.osplit = 2; a.isplit = 2; hipLaunchKernelGGL((nerf_wgrad_f32_kernel<2, 4>), grid, blk, 0, st, a); }
  else if (to > 1)           { a.osplit = 4; a.isplit = 1; hipLaunchKernelGGL((nerf_wgrad_f32_kernel<1, 1>), grid, blk, 0, st, a); }
  else if (ti > 4)           { a.osplit = 1; a.isplit = 4; hipLaunchKernelGGL((nerf_wgrad_f32_kernel<1, 2>), grid, blk, 0, st, a); }
  else                       { a.osplit = 1; a.isplit = 4; hipLaunchKernelGGL((nerf_wgrad_f32_kernel<1, 1>), grid, blk, 0, st, a); }
  return check_launch("nerf_wgrad_f32_kernel");
}

int32_t nerf_wgrad(const float* dz, int64_t ldz, int32_t zc0, int32_t n_out, const float* hin, int64_t ldh,
                   int32_t hc0, int32_t n_in, float* dw, int64_t ldw, int32_t wc0, float* db, int64_t n_points,
                   void* stream) {
  return wgrad_impl(dz, ldz, zc0, n_out, hin, ldh, hc0, n_in, dw, ldw, wc0, db, n_points, nullptr, nullptr, stream);
}

int32_t nerf_composite_backward(const float* raw, const float* tvals, int64_t t_ray_stride, int64_t n_rays,
                                int32_t n_samples, int32_t white_bkgd, const float* g_rgb, const float* g_depth,
                                float* g_raw, float* g_t, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || t_ray_stride < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_composite_backward: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!raw || !tvals || !g_rgb || !g_raw) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_composite_backward: null argument");
  if (n_samples > kCbMaxSamples) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_composite_backward: at most 192 samples per ray");
  hipLaunchKernelGGL(nerf_composite_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, raw, tvals,
                     (long long)t_ray_stride, (long long)n_rays, n_samples, white_bkgd, g_rgb, g_depth, g_raw, g_t);
  return check_launch("nerf_composite_bwd_kernel");
}

int32_t nerf_sample_fine_backward(const float* raw_coarse, const float* t_coarse, const float* u, int64_t n_rays,
                                  const float* t_sorted, const float* g_t_sorted, float* g_raw_coarse, void* stream) {
  if (n_rays < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_sample_fine_backward: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!raw_coarse || !t_coarse || !u || !t_sorted || !g_t_sorted || !g_raw_coarse)
    return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_sample_fine_backward: null argument");
  hipLaunchKernelGGL(nerf_sample_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, raw_coarse, t_coarse, u, (long long)n_rays, t_sorted, g_t_sorted, g_raw_coarse);
  return check_launch("nerf_sample_bwd_kernel");
}

int32_t nerf_viewdirs_backward(const float* gsave, int64_t n_rays, int32_t n_samples, const float* w_views,
                               const float* viewdirs, float* g_viewdirs, void* stream) {
  if (n_rays < 0 || n_samples <= 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_viewdirs_backward: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!gsave || !w_views || !viewdirs || !g_viewdirs) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_viewdirs_backward: null argument");
  if (n_rays > 0x7fffffffLL) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_viewdirs_backward: too many rays for one launch");
  hipLaunchKernelGGL(nerf_viewdirs_bwd_kernel, dim3((unsigned)n_rays), dim3(128), 0, (hipStream_t)stream,
                     gsave + TrainGrad::off_gzv(n_rays * n_samples), (long long)n_rays, n_samples, w_views, viewdirs, g_viewdirs);
  return check_launch("nerf_viewdirs_bwd_kernel");
}

int32_t nerf_adam_step(int32_t n_tensors, float* const params[], const float* const grads[], float* const exp_avg[],
                       float* const exp_avg_sq[], const int64_t numel[], float lr, float beta1, float beta2, float eps,
                       float weight_decay, float clip_value, int64_t step, void* stream) {
  if (n_tensors <= 0 || n_tensors > kAdamMaxTensors || step <= 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_adam_step: bad size");
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_adam_step: null argument");
  AdamArgs a;
  long long run = 0;
  for (int i = 0; i < n_tensors; ++i) {
    if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_adam_step: null tensor");
    a.p[i] = params[i]; a.g[i] = grads[i]; a.m[i] = exp_avg[i]; a.v[i] = exp_avg_sq[i];
    run += numel[i]; a.end[i] = run;
  }
  if (run == 0) return NERF_OK;
  a.n_tensors = n_tensors; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.clip = clip_value;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  long long blocks = (run + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nerf_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("nerf_adam_kernel");
}

int64_t nerf_train_grad_floats(int64_t n_points) { return n_points < 0 ? -1 : TrainGrad::floats(n_points); }
int64_t nerf_train_live_count_offset(int64_t n_points) { return n_points < 0 ? -1 : TrainGrad::off_count(n_points); }
int64_t nerf_packed_bwd_bytes(int32_t precision) {
  if (precision == NERF_PREC_F32) return nerf::kBwdPackedFloats * (int64_t)sizeof(float);
  if (precision == NERF_PREC_F32X) return nerf::kXbPackedBytes;
  return -1;
}

int32_t nerf_pack_model_bwd(const float* const params[24], void* packed_bwd_v, int32_t precision, void* stream) {
  float* packed_bwd = (float*)packed_bwd_v;
  if (!params || !packed_bwd) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_pack_model_bwd: null argument");
  if (precision == NERF_PREC_F32X) {
    PackArgs a;
    for (int i = 0; i < nerf::P_COUNT; ++i) {
      if (!params[i]) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_pack_model_bwd: null parameter pointer");
      a.p[i] = params[i];
    }
    a.out = packed_bwd;
    const long long n = nerf::kF16ConstBytes / 4 + (long long)nerf::kXbSteps * 512;
    hipLaunchKernelGGL(nerf_pack_bwd_f32x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("nerf_pack_bwd_f32x_kernel");
  }
  if (precision != NERF_PREC_F32) return fail(NERF_ERR_UNSUPPORTED, "%s", "nerf_pack_model_bwd: f32 or f32x only");
  PackArgs a;
  for (int i = 0; i < nerf::P_COUNT; ++i) {
    if (!params[i]) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_pack_model_bwd: null parameter pointer");
    a.p[i] = params[i];
  }
  a.out = packed_bwd;
  hipLaunchKernelGGL(nerf_pack_bwd_kernel, dim3((unsigned)((nerf::kBwdPackedFloats + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("nerf_pack_bwd_kernel");
}

// grads[24]: device pointers in state_dict order (nn.Linear layouts), accumulated into (caller zeroes them)
// shared by the ray-mode and the point-mode entry: data-gradient chain, then the weight / bias gradients
static int32_t mlp_backward_impl(const BwdArgs& a_in, bool pts_mode, float* const grads[24], int32_t precision, void* stream) {
  // density only: both chains skip the colour branch in the kernel (ray mode), and its three weight-gradient jobs are skipped
  // here: their result is exactly zero
  BwdArgs a = a_in;
  const bool dens = a.density_only != 0;
  const long long P = a.n_points;
  const float* draw = a.draw; const float* save = a.save; float* gsave = a.gsave;
  int rc;
  // live tiles: tiles whose incoming gradient is zero throughout are dropped from the chain launch and from every
  // weight-gradient launch -- see nerf_tile_flags_kernel.  Needs the asm-ring weight-gradient kernels (whole 32-point tiles).
  // NERF_DEAD_TILE_SKIP=0 in the environment turns it off (tests compare the two).
  const int* live = nullptr; const int* n_live = nullptr;
  {
    if (dead_tile_list_available(P, precision)) {
      int* flags = reinterpret_cast<int*>(gsave + TrainGrad::off_flags(P));
      int* lv = reinterpret_cast<int*>(gsave + TrainGrad::off_live(P));
      int* cnt = reinterpret_cast<int*>(gsave + TrainGrad::off_count(P));
      hipLaunchKernelGGL(nerf_tile_flags_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                         reinterpret_cast<const f32x4*>(draw), P, a.density_only, flags);
      rc = check_launch("nerf_tile_flags_kernel");
      if (rc) return rc;
      hipLaunchKernelGGL(nerf_tile_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, flags, (int)(P / 32), lv, cnt);
      rc = check_launch("nerf_tile_scan_kernel");
      if (rc) return rc;
      // dead tiles have no workgroup: their g_t / g_x is the zero written here
      if (!pts_mode && a.g_t && hipMemsetAsync(a.g_t, 0, (size_t)P * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return fail(NERF_ERR_HIP, "%s", "nerf_mlp_backward: memset failed");
      if (pts_mode && a.g_x && hipMemsetAsync(a.g_x, 0, (size_t)P * 3 * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return fail(NERF_ERR_HIP, "%s", "nerf_mlp_backward: memset failed");
      // point mode has one more consumer of gsave: nerf_viewdirs_backward sums the g_zv rows of ALL samples of a ray, dead
      // tiles included -- their rows are the zeros written here (ray mode: nothing reads a dead tile's rows)
      if (pts_mode && hipMemsetAsync(gsave + TrainGrad::off_gzv(P), 0, (size_t)TrainSave::pad32(P) * 128 * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return fail(NERF_ERR_HIP, "%s", "nerf_mlp_backward: memset failed");
      live = lv; n_live = cnt;
      a.live_tiles = lv; a.n_live = cnt;
    } else {
      if (hipMemsetAsync(gsave + TrainGrad::off_count(P), 0xFF, sizeof(int), (hipStream_t)stream) != hipSuccess)   // count = -1: no list
        return fail(NERF_ERR_HIP, "%s", "nerf_mlp_backward: memset failed");
      // a dense backward needs every row of `save`: refuse (NaN in the alpha-bias gradient) a buffer whose forward skipped rows
      hipLaunchKernelGGL(nerf_check_stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, save + TrainSave::off_stamp(P), grads[nerf::P_BA]);
      rc = check_launch("nerf_check_stamp_kernel");
      if (rc) return rc;
    }
  }
  if (precision == NERF_PREC_F32X) {
    const long long n_tiles = (P + kXTilePts - 1) / kXTilePts;
    const unsigned blocks = (unsigned)(n_tiles < num_cus() ? n_tiles : num_cus());
    if (pts_mode) hipLaunchKernelGGL(nerf_mlp_bwd_f32x_kernel<true>, dim3(blocks), dim3(kXThreads), 0, (hipStream_t)stream, a);
    else if (dens) hipLaunchKernelGGL((nerf_mlp_bwd_f32x_kernel<false, true>), dim3(blocks), dim3(kXThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(nerf_mlp_bwd_f32x_kernel<false>, dim3(blocks), dim3(kXThreads), 0, (hipStream_t)stream, a);
    rc = check_launch("nerf_mlp_bwd_f32x_kernel");
  } else if (precision == NERF_PREC_F32) {
    const long long tiles = (P + nerf::kTilePts - 1) / nerf::kTilePts;
    if (tiles > 0x7fffffffLL) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_backward: too many points for one launch");
    // barrier-free: one-wave workgroups (with a live list: one slot per tile, slots >= *n_live exit at once)
    if (pts_mode) hipLaunchKernelGGL(nerf_mlp_bwd_f32_kernel<true>, dim3((unsigned)tiles), dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(nerf_mlp_bwd_f32_kernel<false>, dim3((unsigned)tiles), dim3(64), 0, (hipStream_t)stream, a);
    rc = check_launch("nerf_mlp_bwd_f32_kernel");
  } else return fail(NERF_ERR_UNSUPPORTED, "%s", "nerf_mlp_backward: f32 or f32x only");
  if (rc) return rc;
  // weight / bias gradients: grad_W = g_z^T @ input, grad_b = sum g_z   (network.py:22-47 layers)
  const float* pe = save + TrainSave::off_pe(P);
  const float* dpe = save + TrainSave::off_dpe(P);
  auto H = [&](int l) { return save + TrainSave::off_h(P, l); };
  auto GZ = [&](int l) { return gsave + TrainGrad::off_gz(P, l); };
  const float* f = save + TrainSave::off_f(P);
  const float* hv = save + TrainSave::off_hv(P);
  const float* gzv = gsave + TrainGrad::off_gzv(P);
  const float* gf = gsave + TrainGrad::off_gf(P);
  using namespace nerf;
#define WG(...) do { rc = wgrad_impl(__VA_ARGS__, P, live, n_live, stream); if (rc) return rc; } while (0)
  // density only: the gradients of rgb_linear, views_linears.0 and feature_linear are identically zero (left as zeroed by the caller)
  if (!dens) WG(draw, 4, 0, 3, hv, 128, 0, 128, grads[P_WR], 128, 0, grads[P_BR]);      // rgb_linear
  WG(draw, 4, 3, 1, H(7), 256, 0, 256, grads[P_WA], 256, 0, grads[P_BA]);               // alpha_linear
  if (!dens) {
    WG(gzv, 128, 0, 128, f, 256, 0, 256, grads[P_WV], 283, 0, grads[P_BV]);             // views_linears.0 [feature | dirs]
    WG(gzv, 128, 0, 128, dpe, 32, 0, 27, grads[P_WV], 283, 256, nullptr);
  }
  WG(GZ(5), 256, 0, 256, pe, 64, 0, 63, grads[10], 319, 0, grads[11]);                  // skip layer, PE part (+ bias)
  WG(GZ(0), 256, 0, 256, pe, 64, 0, 63, grads[P_W0], 63, 0, grads[P_B0]);
  if (precision == NERF_PREC_F32X) {
    // the eight 256 x 256 blocks in one launch on the bf16x3 path (nerf_wgrad_bf16x3.hip.inc)
    WgradXArgs w;
    w.n_points = P; w.n_jobs = 0; w.live_tiles = live; w.n_live = n_live;
    auto job = [&](const float* dz, const float* hin, float* dw, int ldw, int wc0, float* db) {
      WgradXJob& j = w.job[w.n_jobs++];
      j.dz = dz; j.hin = hin; j.dw = dw; j.db = db; j.ldz = 256; j.zc0 = 0; j.ldh = 256; j.hc0 = 0; j.ldw = ldw; j.wc0 = wc0;   // ld 256: the kernel assumes 1-KiB rows
    };
    if (!dens) job(gf, H(7), grads[P_WF], 256, 0, grads[P_BF]);
    for (int l = 7; l >= 1; --l) {
      if (l == 5) job(GZ(5), H(4), grads[10], 319, 63, nullptr);
      else job(GZ(l), H(l - 1), grads[2 * l], 256, 0, grads[2 * l + 1]);
    }
    const long long steps = (P + 15) / 16;
    long long slices = num_cus() / w.n_jobs;
    if (slices > steps) slices = steps;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(nerf_wgrad256_bf16x3_kernel, dim3((unsigned)(slices * w.n_jobs)), dim3(256), 0, (hipStream_t)stream, w);
    rc = check_launch("nerf_wgrad256_bf16x3_kernel");
    if (rc) return rc;
  } else if (NERF_WGRAD_ASM && P % 16 == 0 && P / 16 >= num_cus() / 8) {
    // fp32 MFMA, the eight 256 x 256 blocks in one launch of the asm-load kernel
    WgradBatch wb;
    wb.n_jobs = 0;
    auto job = [&](const float* dz, const float* hin, float* dw, int ldw, int wc0, float* db) {
      WgradArgs& j = wb.job[wb.n_jobs++];
      j.dz = dz; j.ldz = 256; j.zc0 = 0; j.n_out = 256; j.hin = hin; j.ldh = 256; j.hc0 = 0; j.n_in = 256;
      j.dw = dw; j.ldw = ldw; j.wc0 = wc0; j.db = db; j.n_points = P; j.osplit = 2; j.isplit = 2;
      j.live_tiles = live; j.n_live = n_live;
    };
    if (!dens) job(gf, H(7), grads[P_WF], 256, 0, grads[P_BF]);
    for (int l = 7; l >= 1; --l) {
      if (l == 5) job(GZ(5), H(4), grads[10], 319, 63, nullptr);
      else job(GZ(l), H(l - 1), grads[2 * l], 256, 0, grads[2 * l + 1]);
    }
    long long slices = num_cus() / wb.n_jobs;
    if (slices < 1) slices = 1;                       // a device with fewer CUs than jobs still gets a non-empty grid
    if (live) hipLaunchKernelGGL(nerf_wgrad256_f32_asm_kernel<true>, dim3((unsigned)(slices * wb.n_jobs)), dim3(256), 0, (hipStream_t)stream, wb);
    else hipLaunchKernelGGL(nerf_wgrad256_f32_asm_kernel<false>, dim3((unsigned)(slices * wb.n_jobs)), dim3(256), 0, (hipStream_t)stream, wb);
    rc = check_launch("nerf_wgrad256_f32_asm_kernel");
    if (rc) return rc;
  } else {
    if (!dens) WG(gf, 256, 0, 256, H(7), 256, 0, 256, grads[P_WF], 256, 0, grads[P_BF]);   // feature_linear
    for (int l = 7; l >= 1; --l) {
      if (l == 5) WG(GZ(5), 256, 0, 256, H(4), 256, 0, 256, grads[10], 319, 63, nullptr);   // skip layer, hidden part
      else WG(GZ(l), 256, 0, 256, H(l - 1), 256, 0, 256, grads[2 * l], 256, 0, grads[2 * l + 1]);
    }
  }
#undef WG
  return NERF_OK;
}

int32_t nerf_mlp_backward(const float* rays_o, const float* rays_d, const float* tvals, int64_t t_ray_stride,
                          int64_t n_rays, int32_t n_samples, const void* packed_bwd_v, const float* draw,
                          const float* save, float* gsave, float* g_t, float* const grads[24], int32_t precision,
                          void* stream) {
  if (n_rays < 0 || n_samples <= 0 || t_ray_stride < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_backward: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!rays_o || !rays_d || !tvals || !packed_bwd_v || !draw || !save || !gsave || !grads)
    return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_backward: null argument");
  for (int i = 0; i < 24; ++i) if (!grads[i]) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_backward: null gradient pointer");
  BwdArgs a{};
  a.rays_o = rays_o; a.rays_d = rays_d; a.tvals = tvals; a.t_ray_stride = t_ray_stride; a.n_points = n_rays * n_samples;
  a.n_samples = n_samples; a.packed_bwd = (const float*)packed_bwd_v; a.draw = draw; a.save = save; a.gsave = gsave; a.g_t = g_t;
  return mlp_backward_impl(a, false, grads, precision, stream);
}

int32_t nerf_mlp_backward_density(const float* rays_o, const float* rays_d, const float* tvals, int64_t t_ray_stride,
                                  int64_t n_rays, int32_t n_samples, const void* packed_bwd_v, const float* draw,
                                  const float* save, float* gsave, float* g_t, float* const grads[24], int32_t precision,
                                  void* stream) {
  if (n_rays < 0 || n_samples <= 0 || t_ray_stride < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_backward_density: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!rays_o || !rays_d || !tvals || !packed_bwd_v || !draw || !save || !gsave || !grads)
    return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_backward_density: null argument");
  for (int i = 0; i < 24; ++i) if (!grads[i]) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_backward_density: null gradient pointer");
  BwdArgs a{};
  a.rays_o = rays_o; a.rays_d = rays_d; a.tvals = tvals; a.t_ray_stride = t_ray_stride; a.n_points = n_rays * n_samples;
  a.n_samples = n_samples; a.packed_bwd = (const float*)packed_bwd_v; a.draw = draw; a.save = save; a.gsave = gsave; a.g_t = g_t;
  a.density_only = 1;
  return mlp_backward_impl(a, false, grads, precision, stream);
}

int32_t nerf_mlp_backward_points(const float* pts, int64_t n_rays, int32_t n_samples, const void* packed_bwd_v,
                                 const float* draw, const float* save, float* gsave, float* g_pts,
                                 float* const grads[24], int32_t precision, void* stream) {
  if (n_rays < 0 || n_samples <= 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_backward_points: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!pts || !packed_bwd_v || !draw || !save || !gsave || !grads)
    return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_backward_points: null argument");
  for (int i = 0; i < 24; ++i) if (!grads[i]) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_backward_points: null gradient pointer");
  BwdArgs a{};
  a.pts = pts; a.g_x = g_pts; a.n_points = n_rays * n_samples; a.n_samples = n_samples;
  a.packed_bwd = (const float*)packed_bwd_v; a.draw = draw; a.save = save; a.gsave = gsave;
  return mlp_backward_impl(a, true, grads, precision, stream);
}

int64_t nerf_train_save_floats(int64_t n_points) { return n_points < 0 ? -1 : TrainSave::floats(n_points); }

static int32_t forward_rays_save_impl(const float* rays_o, const float* rays_d, const float* tvals,
                                      int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                      const void* packed, float* raw, float* save, int32_t precision, void* stream, int density_only,
                                      int skip_dead = 0) {
  if (n_rays < 0 || n_samples <= 0 || t_ray_stride < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_rays_save: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!rays_o || !rays_d || !tvals || !packed || !raw || !save) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_rays_save: null argument");
  MlpArgs a{};
  a.rays_o = rays_o; a.rays_d = rays_d; a.tvals = tvals; a.t_ray_stride = t_ray_stride;
  a.n_points = n_rays * n_samples; a.n_samples = n_samples; a.packed = (const float*)packed; a.raw = raw; a.save = save;
  a.density_only = density_only;
  // stamp: 1 = the rows of density-free tiles are NOT stored (for-compositing entry with the list available)
  const bool rows_skipped = (precision == NERF_PREC_F32 || precision == NERF_PREC_F32X) && !density_only && skip_dead;
  if (hipMemsetAsync(save + TrainSave::off_stamp(a.n_points), rows_skipped ? 0x01 : 0x00, 4 * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return fail(NERF_ERR_HIP, "%s", "nerf_mlp_forward_rays_save: memset failed");
  if (precision == NERF_PREC_F32X) {
    const long long n_tiles = (a.n_points + kXTilePts - 1) / kXTilePts;
    const unsigned blocks = (unsigned)(n_tiles < num_cus() ? n_tiles : num_cus());
    if (density_only) hipLaunchKernelGGL((nerf_mlp_f32x_kernel<true, true, true>), dim3(blocks), dim3(kXThreads), 0, (hipStream_t)stream, a);
    else if (skip_dead) hipLaunchKernelGGL((nerf_mlp_f32x_kernel<true, true, false, true>), dim3(blocks), dim3(kXThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((nerf_mlp_f32x_kernel<true, true>), dim3(blocks), dim3(kXThreads), 0, (hipStream_t)stream, a);
    return check_launch("nerf_mlp_f32x_kernel<save>");
  }
  if (precision != NERF_PREC_F32) return fail(NERF_ERR_UNSUPPORTED, "%s", "nerf_mlp_forward_rays_save: f32 or f32x only");
  const long long tiles = (a.n_points + nerf::kTilePts - 1) / nerf::kTilePts;
  if (tiles > 0x7fffffffLL) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_rays_save: too many points for one launch");
  // barrier-free: one-wave workgroups
  if (density_only) hipLaunchKernelGGL((nerf_mlp_f32_kernel<true, true, true>), dim3((unsigned)tiles), dim3(64), 0, (hipStream_t)stream, a);
  else if (skip_dead) hipLaunchKernelGGL((nerf_mlp_f32_kernel<true, true, false, true>), dim3((unsigned)tiles), dim3(64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((nerf_mlp_f32_kernel<true, true>), dim3((unsigned)tiles), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("nerf_mlp_f32_kernel<save>");
}

int32_t nerf_mlp_forward_rays_save(const float* rays_o, const float* rays_d, const float* tvals,
                                   int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                   const void* packed, float* raw, float* save, int32_t precision, void* stream) {
  return forward_rays_save_impl(rays_o, rays_d, tvals, t_ray_stride, n_rays, n_samples, packed, raw, save, precision, stream, 0);
}
int32_t nerf_mlp_forward_rays_save_for_compositing(const float* rays_o, const float* rays_d, const float* tvals,
                                                   int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                                   const void* packed, float* raw, float* save, int32_t precision, void* stream) {
  // without dead-tile skipping in the backward pass every row must exist: fall back to the full store (the SAME helper decides
  // for the backward pass, and the buffer is stamped with what was done here)
  const int skip = n_rays > 0 && n_samples > 0 && dead_tile_list_available(n_rays * (int64_t)n_samples, precision);
  return forward_rays_save_impl(rays_o, rays_d, tvals, t_ray_stride, n_rays, n_samples, packed, raw, save, precision, stream, 0, skip);
}
int32_t nerf_mlp_forward_rays_save_density(const float* rays_o, const float* rays_d, const float* tvals,
                                           int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                           const void* packed, float* raw, float* save, int32_t precision, void* stream) {
  return forward_rays_save_impl(rays_o, rays_d, tvals, t_ray_stride, n_rays, n_samples, packed, raw, save, precision, stream, 1);
}

int32_t nerf_mlp_forward_points_save(const float* pts, const float* viewdirs, int64_t n_rays, int32_t n_samples,
                                     const void* packed, float* raw, float* save, int32_t precision, void* stream) {
  if (n_rays < 0 || n_samples <= 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_points_save: bad size");
  if (n_rays == 0) return NERF_OK;
  if (!pts || !viewdirs || !packed || !raw || !save) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_points_save: null argument");
  MlpArgs a{};
  a.pts = pts; a.viewdirs = viewdirs; a.n_points = n_rays * n_samples; a.n_samples = n_samples;
  a.packed = (const float*)packed; a.raw = raw; a.save = save;
  if (hipMemsetAsync(save + TrainSave::off_stamp(a.n_points), 0, 4 * sizeof(float), (hipStream_t)stream) != hipSuccess)      // every row stored
    return fail(NERF_ERR_HIP, "%s", "nerf_mlp_forward_points_save: memset failed");
  if (precision == NERF_PREC_F32X) {
    const long long n_tiles = (a.n_points + kXTilePts - 1) / kXTilePts;
    const unsigned blocks = (unsigned)(n_tiles < num_cus() ? n_tiles : num_cus());
    hipLaunchKernelGGL((nerf_mlp_f32x_kernel<false, true>), dim3(blocks), dim3(kXThreads), 0, (hipStream_t)stream, a);
    return check_launch("nerf_mlp_f32x_kernel<points,save>");
  }
  if (precision != NERF_PREC_F32) return fail(NERF_ERR_UNSUPPORTED, "%s", "nerf_mlp_forward_points_save: f32 or f32x only");
  const long long tiles = (a.n_points + nerf::kTilePts - 1) / nerf::kTilePts;
  if (tiles > 0x7fffffffLL) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_mlp_forward_points_save: too many points for one launch");
  hipLaunchKernelGGL((nerf_mlp_f32_kernel<false, true>), dim3((unsigned)tiles), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("nerf_mlp_f32_kernel<points,save>");
}

int32_t nerf_image_ssim(const float* pred, const float* gt, int32_t H, int32_t W, double* sum1, void* stream) {
  if (H < 7 || W < 7) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_image_ssim: image smaller than the 7x7 window");
  if (!pred || !gt || !sum1) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_image_ssim: null argument");
  if (hipMemsetAsync(sum1, 0, sizeof(double), (hipStream_t)stream) != hipSuccess)
    return fail(NERF_ERR_HIP, "%s", "nerf_image_ssim: memset failed");
  long long blocks = ((long long)(H - 6) * (W - 6) * 3 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nerf_ssim_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, gt, H, W, sum1);
  return check_launch("nerf_ssim_kernel");
}

// Ray blocks.  nerf_render_forward walks a frame in blocks of kRenderBlockRays rays (the four stages per block, same stream): the
// intermediates -- raw_coarse, t_sorted, raw_fine: 4864 B per ray -- belong to ONE block, so the workspace is bounded (5.1 GB) whatever
// the frame (round 2: 12.5 GB at 1600x1600, growing with the frame).  Rays are independent: the image is bit-identical to the one-block
// render (tests).  The block is LARGE on purpose: with 65 536-ray blocks (320 MB, Infinity-Cache-resident intermediates) the fp32
// frame lost 0.2 % and the fp16 frame 5 % (134.7 -> 141.4 ms at 800x800) -- every block pays the persistent kernels' pipeline fill and
// tail, and the far-plane guard's launch (one point per ray) fills 2 tiles per CU -- while nothing is gained from the cache
// residency: the MLP launches are matrix-bound and HBM sits at < 1 % of its bandwidth either way.  NERF_RENDER_BLOCK_RAYS in the
// environment overrides the block size (A/B, tests).
static constexpr int64_t kRenderBlockRays = 1 << 20;
static int64_t render_block_rays() {
  const char* env = getenv("NERF_RENDER_BLOCK_RAYS");
  if (env) { const long long v = atoll(env); if (v >= 64) return (int64_t)v; }
  return kRenderBlockRays;
}

int64_t nerf_render_workspace_bytes(int64_t n_rays_frame, int32_t n_importance, int32_t fast_sampling) {
  if (n_rays_frame < 0) return -1;
  const int64_t n_rays = n_rays_frame < render_block_rays() ? n_rays_frame : render_block_rays();
  const int64_t raw_c = align256(n_rays * NERF_N_SAMPLES * 4 * (int64_t)sizeof(float));
  if (n_importance == 0) return raw_c + align256(n_rays * (int64_t)sizeof(int)) + 256;
  const int64_t S = NERF_N_SAMPLES + NERF_N_IMPORTANCE;
  int64_t total = raw_c + align256(n_rays * S * (int64_t)sizeof(float)) + align256(n_rays * S * 4 * (int64_t)sizeof(float));
  if (fast_sampling) total += align256(n_rays * S) + align256(n_rays * S * (int64_t)sizeof(int)) + 256;   // mask, index, count
  else total += align256(n_rays * (int64_t)sizeof(int)) + 256;                                            // last-sample ids, count (fp16 far-plane guard)
  return total;
}

static int32_t render_block(const float* rays_o, const float* rays_d, int64_t n_rays, const void* packed_coarse, const void* packed_fine,
                            const float* t_coarse, const float* u, int32_t n_importance, int32_t white_bkgd, int32_t precision,
                            int32_t fast_sampling, float weights_threshold, void* workspace, float* rgb, float* depth, void* stream);

int32_t nerf_render_forward(const float* rays_o, const float* rays_d, int64_t n_rays,
                            const void* packed_coarse, const void* packed_fine,
                            const float* t_coarse, const float* u, int32_t n_importance,
                            int32_t white_bkgd, int32_t precision, int32_t fast_sampling,
                            float weights_threshold, void* workspace,
                            int64_t workspace_bytes, float* rgb, float* depth, void* stream) {
  if (n_rays < 0) return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_render_forward: bad size");
  if (n_importance != 0 && n_importance != NERF_N_IMPORTANCE)
    return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_render_forward: n_importance must be 0 or 128");
  if (n_rays == 0) return NERF_OK;
  if (!rays_o || !rays_d || !packed_coarse || !t_coarse || !rgb || !depth || !workspace ||
      (n_importance && (!packed_fine || !u)))
    return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_render_forward: null argument");
  if (workspace_bytes < nerf_render_workspace_bytes(n_rays, n_importance, fast_sampling))
    return fail(NERF_ERR_WORKSPACE, "%s", "nerf_render_forward: workspace too small");
  // the masked fine pass addresses (ray, sample) pairs by 32-bit ids (nerf_compact_kernel, MlpArgs::index)
  if (fast_sampling && n_importance && n_rays > (int64_t)0x7fffffff / (NERF_N_SAMPLES + NERF_N_IMPORTANCE))
    return fail(NERF_ERR_INVALID_ARG, "%s", "nerf_render_forward: fast_sampling handles at most 11 184 810 rays per call "
                                            "(n_rays * 192 point ids must fit in int32): split the frame");
  const int64_t B = render_block_rays();
  for (int64_t r0 = 0; r0 < n_rays; r0 += B) {
    const int64_t nb = n_rays - r0 < B ? n_rays - r0 : B;
    const int rc = render_block(rays_o + 3 * r0, rays_d + 3 * r0, nb, packed_coarse, packed_fine, t_coarse, u, n_importance, white_bkgd,
                                precision, fast_sampling, weights_threshold, workspace, rgb + 3 * r0, depth + r0, stream);
    if (rc) return rc;
  }
  return NERF_OK;
}

// one block of rays through the four stages (workspace: nerf_render_workspace_bytes of the block)
static int32_t render_block(const float* rays_o, const float* rays_d, int64_t n_rays, const void* packed_coarse, const void* packed_fine,
                            const float* t_coarse, const float* u, int32_t n_importance, int32_t white_bkgd, int32_t precision,
                            int32_t fast_sampling, float weights_threshold, void* workspace, float* rgb, float* depth, void* stream) {
  char* ws = (char*)workspace;
  float* raw_c = (float*)ws;
  // hierarchical render: the coarse network only places the fine samples -- nothing but its sigma is read
  // (volume_renderer.py:335; the returned rgb/depth come from the fine outputs, :414-437), so the coarse pass stops after
  // the sigma head.  With n_importance == 0 the coarse outputs ARE the frame and the full network runs.
  int rc = n_importance ? nerf_mlp_forward_rays_density(rays_o, rays_d, t_coarse, 0, n_rays, NERF_N_SAMPLES, packed_coarse, raw_c, precision, stream)
                        : nerf_mlp_forward_rays(rays_o, rays_d, t_coarse, 0, n_rays, NERF_N_SAMPLES, packed_coarse, raw_c, precision, stream);
  if (rc) return rc;
  // fp16 far-plane guard.  The last sample of a ray has delta = 1e10 (volume_renderer.py:85-86): ANY sigma > 0 there makes alpha = 1, so
  // an fp16 rounding that flips the sign of a sigma within 1e-2 of zero turns a background ray into a full far-plane hit (round 2:
  // single rays off by 0.64 in rgb and 6.0 in depth, which alone set the fp16 PSNR).  Every other sample's alpha moves by
  // |d sigma| * delta ~ 1e-4.  So the fp16 precisions re-evaluate exactly that sample of every ray (0.5 % of the points) with the
  // split-fp16 stream that rides behind their packed model: fp32-accurate sigma and colour where it matters, ~1.5 % of the frame.
  const bool guard = (precision == NERF_PREC_F16 || precision == NERF_PREC_F16S) && n_rays <= (int64_t)0x7fffffff / (NERF_N_SAMPLES + NERF_N_IMPORTANCE);
  auto far_plane_guard = [&](const float* tvals, int64_t stride, int32_t S_, const void* packed_f16, float* raw, char* ids_base) -> int {
    int* index = (int*)ids_base;
    int* count = (int*)(ids_base + align256(n_rays * (int64_t)sizeof(int)));
    hipLaunchKernelGGL(nerf_last_sample_index_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, index, count,
                       (long long)n_rays, S_);
    int r2 = check_launch("nerf_last_sample_index_kernel");
    if (r2) return r2;
    MlpArgs g{};
    g.rays_o = rays_o; g.rays_d = rays_d; g.tvals = tvals; g.t_ray_stride = stride; g.n_points = n_rays * S_;
    g.n_samples = S_; g.packed = (const float*)((const char*)packed_f16 + nerf::kF16PackedBytes); g.raw = raw; g.index = index; g.count = count;
    return launch_mlp(g, true, NERF_PREC_F32X, (hipStream_t)stream);
  };
  if (n_importance == 0) {
    if (guard) {
      rc = far_plane_guard(t_coarse, 0, NERF_N_SAMPLES, packed_coarse, raw_c, ws + align256(n_rays * NERF_N_SAMPLES * 4 * (int64_t)sizeof(float)));
      if (rc) return rc;
    }
    return nerf_composite(raw_c, t_coarse, 0, n_rays, NERF_N_SAMPLES, white_bkgd, rgb, depth, nullptr, stream);
  }
  const int64_t S = NERF_N_SAMPLES + NERF_N_IMPORTANCE;
  float* t_sorted = (float*)(ws + align256(n_rays * NERF_N_SAMPLES * 4 * (int64_t)sizeof(float)));
  float* raw_f = (float*)((char*)t_sorted + align256(n_rays * S * (int64_t)sizeof(float)));
  if (!fast_sampling) {
    rc = nerf_sample_fine(raw_c, t_coarse, u, n_rays, t_sorted, nullptr, nullptr, 0.f, 0.f, stream);
    if (rc) return rc;
    // the fine outputs only ever reach nerf_composite: colours of zero-density samples are multiplied by exactly 0 there
    rc = nerf_mlp_forward_rays_for_compositing(rays_o, rays_d, t_sorted, S, n_rays, (int32_t)S, packed_fine, raw_f, precision, stream);
    if (rc) return rc;
    if (guard) {
      rc = far_plane_guard(t_sorted, S, (int32_t)S, packed_fine, raw_f, (char*)raw_f + align256(n_rays * S * 4 * (int64_t)sizeof(float)));
      if (rc) return rc;
    }
  } else {
    // ESS/ERT (volume_renderer.py:359-369, network.py:207-253): only the valid merged samples go through
    // the fine network; the others keep raw = 0 (sigma 0 -> weight 0)
    hipStream_t st = (hipStream_t)stream;
    uint8_t* valid = (uint8_t*)((char*)raw_f + align256(n_rays * S * 4 * (int64_t)sizeof(float)));
    int* index = (int*)((char*)valid + align256(n_rays * S));
    int* count = (int*)((char*)index + align256(n_rays * S * (int64_t)sizeof(int)));
    rc = nerf_sample_fine(raw_c, t_coarse, u, n_rays, t_sorted, nullptr, valid, weights_threshold, 0.45f, stream);
    if (rc) return rc;
    if (hipMemsetAsync(count, 0, sizeof(int), st) != hipSuccess ||
        hipMemsetAsync(raw_f, 0, (size_t)(n_rays * S * 4 * (int64_t)sizeof(float)), st) != hipSuccess)
      return fail(NERF_ERR_HIP, "%s", "nerf_render_forward: memset failed");
    const long long np = n_rays * S;
    hipLaunchKernelGGL(nerf_compact_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, valid, np, index, count);
    rc = check_launch("nerf_compact_kernel");
    if (rc) return rc;
    MlpArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.tvals = t_sorted; a.t_ray_stride = S; a.n_points = np;
    a.n_samples = (int)S; a.packed = (const float*)packed_fine; a.raw = raw_f; a.index = index; a.count = count;
    a.skip_dead_colour = 1;
    rc = launch_mlp(a, true, precision, st);
    if (rc) return rc;
  }
  return nerf_composite(raw_f, t_sorted, S, n_rays, (int32_t)S, white_bkgd, rgb, depth, nullptr, stream);
}

}  // extern "C"
