// Internal convolution problem descriptors shared by msk_conv.hip (dispatch, packing,
// VALU reference kernels) and msk_conv_mfma.hip (MFMA kernels).
#pragma once
#include "msk_common.h"
struct WbfBnBwd;

// "Gather convolution": every dst voxel gathers from src voxels
//   transposed == 0:  spos = dpos*s - p + k                      (Conv3D forward, ConvT dgrad)
//   transposed == 1:  spos = (dpos + p - k)/s when divisible     (Conv3D dgrad, ConvT forward)
// dst[m][n] (+)= bias[n] + sum_{tap,k} src[spos(m,tap)][k] * W[tap][k][n]
struct GConv {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, SD, SH, SW, DD, DH, DW;  // src / dst spatial dims
  int CK, CN;                     // reduction channels (src), output channels (dst)
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int transposed;
  const float* bias;  // [CN] or null
  int accumulate;
  int flip;           // MFMA halo path: taps are enumerated flipped (conv dgrad as a 'same' conv)
  float* stats;       // msk_conv3d_fwd_ex: BatchNorm statistics (mean[CN], M2[CN]) of dst wanted; a kernel that produced them in
                      // its epilogue sets ctx->stats_fused, otherwise the caller runs msk_bn_stats
  void* xform;        // msk_conv3d_fwd_ex: caller-owned buffer that receives the transformed input (msk_conv3d_xform_bytes)
  const float* w_amax;   // NP = 2 pipelines: amax array of the weights if the caller has it (else taken here)
  const float* in_amax;  // NP = 2 pipelines: device scalar bounding max |src| (scales the source transform, undone in the output stage); NULL = unscaled
  const struct WbfBnBwd* fuse;  // msk_conv3d_bwd_bnact: src is not read; the input transform evaluates dy from (y, dout) on the fly
  const msk_bn_fin* fin;  // msk_conv3d_fwd_ex3: BatchNorm finalisation to run in the statistics merge (with stats), or null
  int ck_real;        // > 0: src holds only ck_real (< CK, % 4 == 0) channels per voxel, the others count as zeros (gconv_wbf_padded)
  int stats_ps;       // msk_conv3d_fwd_in: statistics PER SAMPLE -- stats is [N][2 CN], fin's save_mean / save_invstd / scale / shift
  int fin_stride;     // advance by fin_stride floats per sample (InstanceNorm); only kernels whose records are per tile serve it
  bool w_persistent;  // the weights are the caller's tensor (covered by msk_weights_changed): derived forms may be cached
  float* dst_lo;      // msk_conv3d_bwd_bnact_split: when the one-kernel matrix stage runs this (accumulating) problem it stores channels
  float* dst_hi;      // [0, dst_csplit) / [dst_csplit, CN) to these dense tensors instead of dst (sets ctx->dst_split_done); else ignored
  int dst_csplit;
  const float* acc_src;  // msk_conv3d_bwd_bnact_acc: an ACCUMULATING problem reads its old values from this tensor (geometry and voxel
                         // stride of dst) instead of dst -- the pass that would have written them to dst first is skipped (round 6)
  const float* prelu; // inference (msk_conv3d_fwd_act): per-channel PReLU slope applied after the bias, or null.  The
                      // Winograd kernels apply it in their epilogue; for every other kernel run_gconv_one adds a pass.
};

// Weight gradient: dW[cb][ca][tap] (+)= sum_opos A[opos*s - p + k][ca] * B[opos][cb]
struct WGrad {
  const float* A;  // "input-side" tensor  [N][AD][AH][AW][CA]
  int ald;
  const float* B;  // "output-side" tensor [N][BD][BH][BW][CB]
  int bld;
  int N, AD, AH, AW, BD, BH, BW;
  int CA, CB;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  const void* xform;  // msk_conv3d_wgrad_ex: transformed A written by msk_conv3d_fwd_ex for the same tensor, or null
  const void* yform;  // msk_conv3d_bwd_bnact: transformed B (A dy) already written by the dual transform, or null
  const float* y_amax;  // NP = 2: device scalar bounding max |B| when yform is given
  const float* y_cmax;  // NP = 2: per-channel max |B| [CB] written with yform (wbf_chan_shift), or null
  const float* b_amax;  // NP = 2, small-channel kernels: max |B| when the caller already has it (amax array), or null
  int cb_real;          // > 0: B holds only cb_real (< CB, % 4 == 0) channels per voxel, the others count as zeros (wgrad_wbf_padded)
  const float* a_amax;  // NP = 2: max |A| when the caller already has it and no kept transform brings it (msk_conv3d_wgrad_ex3), or null
  const struct WbfBnBwd* yfuse;  // msk_conv3d_bwd_bnact (split form): B is not read; its transform evaluates dy from (y, dout)
  float* dw;  // canonical [CB][CA][taps]
  int accumulate;
  float* db;   // bias gradient a kernel that reads the tensor anyway may produce (sets ctx->wgrad_db_done), or null
  int db_src;  // 1: column sums of B (convolution), 2: sums of A over every tap (transposed convolution)
  // Winograd kernels only (filled by msk_wgrad_wino): BD/BH/BW are then LOGICAL dims, a permutation of the tensor's
  // axes, and a voxel's index is n*vsn + d*vsd + h*vsh + w*vsw
  long vsn;
  int vsd, vsh, vsw;
};

// Packed-weight layouts
//   direct: Wp[tap][k][n]                       (n fastest)
//   mfma:   Wm[tap][kc][h][npad][4], k = kc*8 + h*4 + q, zero padded (K -> 8*KC, N -> npad)
// Source canonical weight w[a][b][tap]; swap != 0 -> (k, n) = (b, a) else (k, n) = (a, b).
int msk_pack_weights(msk_ctx* ctx, const float* w, int A, int B, int taps, int swap, int flip_taps,
                     int kd, int kh, int kw, int mfma, int K, int N, int KC, int npad, float* out);

// Cached packed images (round 5, msk_conv.hip SmallPackCache): same layouts, a persistent image per (weights, layout) that the
// optimizer kernels rebuild in one launch; nullptr on failure (msk_pack_scatter_get: also with option "small_pack_cache" 0)
const float* msk_pack_weights_get(msk_ctx* ctx, const float* w, int A, int B, int taps, int swap, int flip_taps, int kd, int kh,
                                  int kw, int mfma, int K, int N, int KC, int npad);
const float* msk_pack_scatter_get(msk_ctx* ctx, const float* w, int A, int B, int taps, int swap, int CK, int CN, int KC, int jpad);

// MFMA kernels (msk_conv_mfma.hip).  Return 1 if the problem was handled, 0 if not eligible,
// <0 on error.
int msk_gconv_halo_mfma(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
int msk_gconv_gather_mfma(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
// 'same' 5^3 conv with <= 4 reduction channels: (tap, channel) pairs enumerated tightly along K (msk_conv_tightk.hip)
int msk_gconv_halo_tightk(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
// the same class (CK <= 4 -> 32 channels) with fp16 two-piece operands, kd folded into K, marching along D (msk_conv_tightk.hip)
int msk_gconv_tk_h2(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
// 'same' 5^3 conv with ONE input channel (in_tr.conv1) on the fp32 matrix pipe, weights and tap offsets in registers (msk_conv_c1.hip)
int msk_gconv_c1_mfma(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
// the same class (and 3^3 with <= 32 output channels) on the 16-bit matrix pipe with fp16 operand pieces (msk_conv_c1.hip)
int msk_gconv_c1_h2(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
// 'same' 5^3 conv 32 -> (<= 4) channels, two voxels per thread on the VALU (msk_conv_valu2.hip)
int msk_gconv_halo_valu2(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
// 'same' 5^3 conv 32 -> (<= 3) channels on the fp32 matrix pipe, kd taps folded into the MFMA columns, marching along D
// (msk_conv_foldn.hip); applies g.prelu in its epilogue (*act_fused)
int msk_gconv_halo_foldn(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap, bool* act_fused);
// 'same' 5^3 conv with a 1-D Winograd F(2,5) transform along W (msk_conv_wino.hip)
int msk_gconv_halo_wino(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
// 'same' 5^3 conv as a three-stage Winograd F(4,5) pipeline with bf16x3 operands on the bf16 matrix pipe (msk_conv_wbf.hip)
int msk_gconv_wino_bf3(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
bool msk_gconv_wino_bf3_accepts(msk_ctx* ctx, const GConv& g);  // the same eligibility tests, nothing launched
// weight gradient of the same layers on the bf16 matrix pipe (msk_wgrad_wbf.hip)
int msk_wgrad_wbf(msk_ctx* ctx, const WGrad& g);
bool msk_wgrad_wbf_accepts(msk_ctx* ctx, const WGrad& g);  // its shape / alignment tests, nothing launched
bool msk_wgrad_wbf_fusable(msk_ctx* ctx, const WGrad& g, size_t* y_bytes);  // see msk_conv3d_bwd_bnact
// kernel == stride transposed gather (up-convs, down-conv data gradients): taps folded into N (msk_conv_scatter.hip)
int msk_gconv_scatter_mfma(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
// kernel == stride forward gather (down-convs, up-conv data gradients): flattened K, two operand batches in flight (msk_conv_ksfwd.hip)
int msk_gconv_ks_fwd(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
// msk_gconv_ks_fwd with the source evaluated from (y, dout) behind a BatchNorm + PReLU (msk_convT3d_bwd_bnact); dry: eligibility only
int msk_gconv_ks_fwd_bnbwd(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap, const float* y, int yld,
                           const float* dout, int dld, const float* scale, const float* shift, const float* alpha, const float* mean,
                           const float* invstd, const float* sums, double M_total, bool dry);
// transposed gather with kernel == stride along D, H and stride 1 along W: the anisotropic MRI levels (msk_conv_ksfwd.hip)
int msk_gconv_kst(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap);
int msk_wgrad_mfma(msk_ctx* ctx, const WGrad& g);
bool msk_gconv_foldn_h2_accepts(const msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, int cout);
int msk_wgrad_cbs(msk_ctx* ctx, const WGrad& g); // <= 4 output channels, 5^3 same (msk_wgrad_cbs.hip)
int msk_wgrad_c1(msk_ctx* ctx, const WGrad& g);  // one input channel, 5^3 same (msk_wgrad_c1.hip)
int msk_wgrad_ks(msk_ctx* ctx, const WGrad& g);  // kernel == stride, no padding (msk_wgrad_ks.hip)
// weight gradient of 'same' 5^3 convs with the Winograd F(2,5) transform along W (msk_wgrad_wino.hip)
int msk_wgrad_wino(msk_ctx* ctx, const WGrad& g);

// split-K reducer shared by both wgrad implementations:
// dw[cb][ca][tap] (+)= sum_s P[s][tap][ca][cb]
// max |x| of a tensor into dst (device float) or, dst == NULL, into a fresh device scalar of the context's ring (NP = 2
// pipelines); returns the scalar's address, NULL on error
const float* msk_absmax(msk_ctx* ctx, const float* x, int ld, int C, long voxels, float* dst = nullptr);
// n zeroed device scalars from the same ring (valid until ~1000 later requests)
float* msk_scalar_slots(msk_ctx* ctx, int n);
const float* msk_bn_bwd_bound(msk_ctx* ctx, int C, const float* scale, const float* sums, double M_total, const float* maxes);
// slabs of `pitch` floats: taps*CA*CB sums in (tap, ca, cb) order followed by nbias bias sums (-> db[0..nbias))
int msk_wgrad_reduce_ex(msk_ctx* ctx, const float* partial, int splits, long pitch, int taps, int CA, int CB, float* dw,
                        int accumulate, int nbias, float* db, int db_accumulate);
int msk_wgrad_reduce(msk_ctx* ctx, const float* partial, int splits, int taps, int CA, int CB, float* dw,
                     int accumulate);
