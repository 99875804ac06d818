// Packed-model layout shared by the pack kernel, the MLP kernels and the host code.
//
// The fused MLP keeps activations in MFMA accumulator layout from layer to layer:
// v_mfma_f32_32x32x2_f32 computes D[32 out-features][32 points]; lane l = (p = l&31, h = l>>5)
// holds, in register r of tile t, feature  act_feat(t,r,h) = 32t + (r&3) + 8(r>>2) + 4h  of
// point p.  Used as the B operand of the next layer, register r of tile t is exactly the K-pair
// {act_feat(t,r,0), act_feat(t,r,1)} of k-step s = 16t + r, so no data movement is needed between
// layers: only the WEIGHTS are permuted, once, on the device (nerf_pack_model), so that the A
// fragment of (k-step s, out-tile j) is  W[32j + (l&31)][act_feat(t,r,l>>5)].
//
// Weight stream: for a layer with KS k-steps and NT out-tiles, float4 index
//     ((g*NT + j)*64 + lane),  g = s/4, component q = s%4
// i.e. one 1-KiB wave-coalesced block per (4 k-steps, out-tile), in exactly the order consumed.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NERF_HD __host__ __device__
#else
#define NERF_HD
#endif

namespace nerf {

constexpr int kHidden = 256;       // network.py:22-32  (W)
constexpr int kXyzFreqs = 10;      // lego.yaml:39-41
constexpr int kDirFreqs = 4;       // lego.yaml:42-44
constexpr int kXyzCh = 63;         // 3 + 6*10
constexpr int kDirCh = 27;         // 3 + 6*4
constexpr int kViewsOut = 128;     // network.py:34-36  (W/2)
constexpr int kTilePts = 32;       // points per wave tile (MFMA N)

NERF_HD constexpr int act_feat(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

// Positional-encoding slot -> reference feature index (freq.py:31-32 ordering:
// [x(3), sin(2^0 x)(3), cos(2^0 x)(3), sin(2^1 x)(3), ...]).  -1 = zero pad.
// xyz: 2 tiles = 32 slots per lane-half.  Lane-half h evaluates sincos of args a = 15h + n/2
// (octave k = a/3, coordinate c = a%3); slots 30,31 carry the raw coordinates.
NERF_HD constexpr int pe_xyz_feat(int n, int h) {
  if (n < 30) {
    int a = 15 * h + n / 2, k = a / 3, c = a % 3;
    return 3 + 6 * k + 3 * (n & 1) + c;
  }
  if (n == 30) return h == 0 ? 0 : 2;
  return h == 0 ? 1 : -1;
}
// view dir: 1 tile = 16 slots per lane-half; args a = 6h + n/2 for n < 12; slots 12,13 raw.
NERF_HD constexpr int pe_dir_feat(int n, int h) {
  if (n < 12) {
    int a = 6 * h + n / 2, k = a / 3, c = a % 3;
    return 3 + 6 * k + 3 * (n & 1) + c;
  }
  if (n == 12) return h == 0 ? 0 : 2;
  if (n == 13) return h == 0 ? 1 : -1;
  return -1;
}

// ---- packed model: offsets in floats -------------------------------------------------------
constexpr int64_t wsize(int ksteps, int ntiles) { return (int64_t)(ksteps / 4) * ntiles * 64 * 4; }

constexpr int64_t kOffL0 = 0;                                  // pts_linears.0   K=63(pad 64) -> 256
constexpr int64_t kOffL1 = kOffL0 + wsize(32, 8);              // pts_linears.1..4
constexpr int64_t kOffL5a = kOffL1 + 4 * wsize(128, 8);        // pts_linears.5, skip (PE) columns 0..62
constexpr int64_t kOffL5b = kOffL5a + wsize(32, 8);            // pts_linears.5, hidden columns 63..318
constexpr int64_t kOffL6 = kOffL5b + wsize(128, 8);            // pts_linears.6, .7
constexpr int64_t kOffFeat = kOffL6 + 2 * wsize(128, 8);       // feature_linear
constexpr int64_t kOffViews = kOffFeat + wsize(128, 8);        // views_linears.0: 256 feature + 27(pad 32) dir
constexpr int64_t kOffBias = kOffViews + wsize(144, 4);        // biases: 9 x [2][128] (L0..L7, feature)
constexpr int64_t kOffBiasViews = kOffBias + 9 * 256;          // [2][64]
constexpr int64_t kOffWAlpha = kOffBiasViews + 128;            // [2][128]
constexpr int64_t kOffWRgb = kOffWAlpha + 256;                 // [3][2][64]
constexpr int64_t kOffHeadBias = kOffWRgb + 384;               // b_rgb[3], b_alpha
constexpr int64_t kPackedFloats = kOffHeadBias + 4;

// ---- fp16-activation path (nerf_mlp_f16.hip.inc): v_mfma_f32_32x32x16_f16 ---------------------
// Same idea, K-step = 16 features: after bias+ReLU the 16 accumulator registers of out-tile m are
// converted pairwise to fp16 and registers 8*s2 .. 8*s2+7 become, unchanged, the 8-element B
// fragment of k-step 2m+s2; lane-half h, element j of k-step s is feature
//     act16_feat(s,j,h) = 16s + (j&3) + 8(j>>2) + 4h.
// A fragment of (out-tile m, k-step s): lane l=(i,h), element j = W[32m+i][act16_feat(s,j,h)] as fp16,
// 1 KiB per fragment.  Fragments are streamed m-outer / k-inner (the order consumed) in 32-KiB
// chunks through a 4-slot LDS ring shared by the 8 waves of a workgroup.
NERF_HD constexpr int act16_feat(int s, int j, int h) { return 16 * s + (j & 3) + 8 * (j >> 2) + 4 * h; }

constexpr int kF16ConstBytes = 16384;          // fp32 biases (same [layer][h][..] layout as the f32 stream tail)
constexpr int kF16FragBytes = 1024;
constexpr int kF16ChunkFrags = 32;
constexpr int kF16ChunkBytes = kF16ChunkFrags * kF16FragBytes;
// fragment ranges of the stream
constexpr int kF16FragL0 = 0;                    // 8 m x 4 PE k-steps
constexpr int kF16FragL1 = 32;                   // L1..L4: 8 m x 16
constexpr int kF16FragL5 = kF16FragL1 + 4 * 128; // 8 m x (4 PE + 16 hidden) = 160
constexpr int kF16FragL6 = kF16FragL5 + 160;     // L6, L7
// the sigma head comes BEFORE the feature layer (both read relu(h7)): a density-only pass ends its stream right behind
// layer 7, and a wave whose 32 points carry no density knows it before the colour branch starts
constexpr int kF16FragSigma = kF16FragL6 + 2 * 128; // 1 m (row 0) x 16
constexpr int kF16FragFeat = kF16FragSigma + 16;    // 8 m x 16
constexpr int kF16FragViews = kF16FragFeat + 128;   // 4 m x (16 feature + 2 dir)
constexpr int kF16FragRgb = kF16FragViews + 72;    // 1 m (rows 0..2) x 8
constexpr int kF16Frags = kF16FragRgb + 8;         // 1184 = 37 chunks exactly
constexpr int kF16Chunks = kF16Frags / kF16ChunkFrags;
static_assert(kF16Frags % kF16ChunkFrags == 0, "stream must be whole chunks");
constexpr int64_t kF16PackedBytes = kF16ConstBytes + (int64_t)kF16Frags * kF16FragBytes;
// "f32x" (nerf_mlp_f32x.hip.inc): the same stream with every fragment followed by its low-part fragment
constexpr int kXFrags = 2 * kF16Frags;                       // (hi, lo) pairs
constexpr int kXChunks = kXFrags / kF16ChunkFrags;           // 74
constexpr int64_t kXPackedBytes = kF16ConstBytes + (int64_t)kXFrags * kF16FragBytes;
// backward f32x stream (nerf_mlp_bwd_f32x.hip.inc): (hi, lo) pairs of transposed-weight fragments
constexpr int kXbStepsWvT = 0;                          // 8 m x 8 s
constexpr int kXbStepsWfT = kXbStepsWvT + 64;           // Wf^T, W7^T, W6^T, W5[:,63:]^T : 8 m x 16 s each
constexpr int kXbStepsW5aT = kXbStepsWfT + 4 * 128;     // 2 m x 16 s (PE slots)
constexpr int kXbStepsW4T = kXbStepsW5aT + 32;          // W4^T .. W1^T
constexpr int kXbStepsW0T = kXbStepsW4T + 4 * 128;      // 2 m x 16 s (PE slots)
constexpr int kXbSteps = kXbStepsW0T + 32;              // 1152
constexpr int kXbChunks = 2 * kXbSteps / kF16ChunkFrags;   // 72
static_assert((2 * kXbSteps) % kF16ChunkFrags == 0, "backward stream must be whole chunks");
constexpr int64_t kXbPackedBytes = kF16ConstBytes + (int64_t)2 * kXbSteps * kF16FragBytes;
// offsets (floats) inside the const region
constexpr int kF16OffBias = 0;                   // 9 x [2][128]
constexpr int kF16OffBiasViews = 9 * 256;        // [2][64]
constexpr int kF16OffHeadBias = kF16OffBiasViews + 128;   // b_rgb[3], b_alpha

// ---- fp16-activation path on v_mfma_f32_16x16x32_f16 ("f16s", nerf_mlp_f16s.hip.inc) -------------
// Same arithmetic class and the same idea as the 32x32x16 path, on the other MFMA shape (the chip holds a higher clock on it,
// MI355X_MICROARCH.md "DVFS give-back" item 7): D[16 out-features][16 points] += A[16][32 k] B[32 k][16 points], lane l = (c = l&15,
// g = l>>4) holds rows 4g..4g+3 of its column c.  A wave still owns 32 points = two column groups (points 16n + c, n = 0, 1) that
// share every A fragment.  The accumulators of out-tiles 2s and 2s+1 (16 features each), converted pairwise to fp16, are unchanged
// the B fragment of k-step s of the next layer: element j of lane group g is feature
//     act16s_feat(s, j, g) = 32 s + 16 (j >> 2) + 4 g + (j & 3).
// A fragment of (out-tile m, k-step s): lane l = (i, g), element j = W[16 m + i][act16s_feat(s, j, g)]: 1 KiB as before, streamed
// m-outer / k-inner through the same LDS ring.  Biases are read in NATURAL order (lane g, tile m: floats 16 m + 4 g ..+3).
NERF_HD constexpr int act16s_feat(int s, int j, int g) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }
// xyz encoding: 64 slots per point = 16 per lane group = 8 (sin, cos) pairs: pairs 0..5 = octaves 2g, 2g+1 x (x, y, z); pairs
// 6, 7 = "extras" x = 2g + e: x < 6: octave 8 + x/3 of coordinate x % 3; x = 6: the raw (x, y); x = 7: (z, pad).
// Slot n of group g = element (n & 7) of PE k-step (n >> 3).
NERF_HD constexpr int pe16s_xyz_feat(int n, int g) {
  const int i = n >> 1, sc = n & 1;
  if (i < 6) return 3 + 6 * (2 * g + i / 3) + 3 * sc + i % 3;
  const int x = 2 * g + (i - 6);
  if (x < 6) return 3 + 6 * (8 + x / 3) + 3 * sc + x % 3;
  if (x == 6) return sc;                 // raw x, y
  return sc == 0 ? 2 : -1;               // raw z, pad
}
// view direction: 32 slots = 8 per lane group: pairs 0..2 = octave g of (x, y, z); pair 3: g = 0 raw (x, y), g = 1 (z, pad), else pad
NERF_HD constexpr int pe16s_dir_feat(int n, int g) {
  const int i = n >> 1, sc = n & 1;
  if (i < 3) return 3 + 6 * g + 3 * sc + i;
  if (g == 0) return sc;
  if (g == 1) return sc == 0 ? 2 : -1;
  return -1;
}
// fragment ranges of the f16s stream (the sigma head again starts a chunk; 12 zero fragments pad the tail to whole chunks)
constexpr int kF16sFragL0 = 0;                       // 16 m x 2 PE k-steps
constexpr int kF16sFragL1 = 32;                      // L1..L4: 16 m x 8
constexpr int kF16sFragL5 = kF16sFragL1 + 4 * 128;   // 16 m x (2 PE + 8 hidden) = 160
constexpr int kF16sFragL6 = kF16sFragL5 + 160;       // L6, L7
constexpr int kF16sFragSigma = kF16sFragL6 + 2 * 128; // 1 m (row 0) x 8
constexpr int kF16sFragFeat = kF16sFragSigma + 8;    // 16 m x 8
constexpr int kF16sFragViews = kF16sFragFeat + 128;  // 8 m x (8 feature + 1 dir)
constexpr int kF16sFragRgb = kF16sFragViews + 72;    // 1 m (rows 0..2) x 4
constexpr int kF16sFragEnd = kF16sFragRgb + 4;       // 1172
static_assert(kF16sFragSigma == kF16FragSigma && kF16sFragEnd <= kF16Frags, "both fp16 streams: 37 chunks, sigma head at chunk 30");
// const region (floats): biases in natural order
constexpr int kF16sOffBias = 0;                      // 9 x [256]: pts_linears.0..7, feature_linear
constexpr int kF16sOffBiasViews = 9 * 256;           // [128]
constexpr int kF16sOffHeadBias = kF16sOffBiasViews + 128;   // b_rgb[3], b_alpha

// ---- transposed stream for the backward (data-gradient) chain, nerf_mlp_bwd_f32.hip.inc -------
// Same block format; A fragment of (k-step over the layer's OUTPUT features, out-tile over its INPUT
// features) = W[act_feat(k-step, lane>>5)][input feature of row (lane&31)]; consumption order:
constexpr int64_t kBwdOffWvT = 0;                                   // Wv[:, :256]^T : 128 -> 256
constexpr int64_t kBwdOffWfT = kBwdOffWvT + wsize(64, 8);           // Wf^T
constexpr int64_t kBwdOffW7T = kBwdOffWfT + wsize(128, 8);
constexpr int64_t kBwdOffW6T = kBwdOffW7T + wsize(128, 8);
constexpr int64_t kBwdOffW5bT = kBwdOffW6T + wsize(128, 8);         // W5[:, 63:]^T
constexpr int64_t kBwdOffW5aT = kBwdOffW5bT + wsize(128, 8);        // W5[:, :63]^T : 256 -> 64 PE slots
constexpr int64_t kBwdOffW4T = kBwdOffW5aT + wsize(128, 2);         // W4^T .. W1^T
constexpr int64_t kBwdOffW0T = kBwdOffW4T + 4 * wsize(128, 8);      // W0^T : 256 -> 64 PE slots
constexpr int64_t kBwdOffWAlpha = kBwdOffW0T + wsize(128, 2);       // [2][128] as in the forward stream
constexpr int64_t kBwdOffWRgb = kBwdOffWAlpha + 256;                // [3][2][64]
constexpr int64_t kBwdPackedFloats = kBwdOffWRgb + 384 + 16 * 256;  // + slack the prefetch ring may read past the end

// Order of the 24 parameter tensors of one sub-model (reference state_dict order, network.py:22-47)
enum ParamIdx {
  P_W0 = 0, P_B0 = 1,           // pts_linears.i -> 2i, 2i+1
  P_WV = 16, P_BV = 17,         // views_linears.0
  P_WF = 18, P_BF = 19,         // feature_linear
  P_WA = 20, P_BA = 21,         // alpha_linear
  P_WR = 22, P_BR = 23,         // rgb_linear
  P_COUNT = 24
};

}  // namespace nerf
