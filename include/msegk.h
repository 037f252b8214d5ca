/*
 * msegk.h -- C ABI of libmsegk.so: the MI355X (gfx950) kernels behind the
 * medicalseg VNet hot path.
 *
 * The reference (PaddleCV-SIG/MedicalSeg) has no FFI of its own: its "operator
 * API" is the set of paddle.* calls made by medicalseg/models/vnet.py,
 * the medicalseg/models/losses modules, medicalseg/core/train.py and the
 * tools/preprocess_utils modules.  Each entry point below names the reference call
 * site(s) it replaces (file:line relative to the reference root).  The Python
 * package medicalseg_amd binds these with ctypes (medicalseg_amd/_lib.py); see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = error (msk_last_error() has text);
 *   - tensors are caller-owned DEVICE pointers described by msk_tensor:
 *     fp32, NDHWC, `ld` floats between consecutive voxels (ld >= c lets a tensor
 *     be a channel slice of a wider buffer -> zero-copy concat);
 *   - weights cross the boundary in the reference's layouts
 *     (Conv3D [Cout,Cin,kD,kH,kW]; Conv3DTranspose [Cin,Cout,kD,kH,kW]);
 *   - all launches are asynchronous on the context's stream; msk_sync blocks;
 *   - one context per device per process; a context is not thread-safe.
 */
#ifndef MSEGK_H
#define MSEGK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msk_ctx msk_ctx;

typedef struct {
  void* p;                 /* device pointer to element (n=0,d=0,h=0,w=0,c=0) */
  int32_t n, d, h, w, c;   /* logical dims */
  int32_t ld;              /* floats per voxel in memory (>= c) */
} msk_tensor;

typedef struct {
  int32_t kd, kh, kw;      /* kernel */
  int32_t sd, sh, sw;      /* stride */
  int32_t pd, ph, pw;      /* zero padding (Conv3D only; 0 for Conv3DTranspose) */
} msk_conv_desc;

/* ---- context / memory ----------------------------------------------------- */
int msk_version(void);
int msk_device_count(int* count);
int msk_ctx_create(int device, msk_ctx** out);
int msk_ctx_destroy(msk_ctx* ctx);
const char* msk_last_error(msk_ctx* ctx);        /* ctx may be NULL (global error) */
int msk_sync(msk_ctx* ctx);
/* Weight-gradient kernels run on an internal side stream (option "wgrad_async"); msk_join_side makes
 * the main stream wait for them -- call it before consuming weight gradients (optimizer, all-reduce).
 * msk_sync and msk_d2h join implicitly. */
int msk_join_side(msk_ctx* ctx);
int msk_device_name(msk_ctx* ctx, char* buf, int buflen);
/* PCI bus id of the context's device ("0000:05:00.0"): the rank -> GPU binding a multi-rank job logs (parallel.py) */
int msk_device_pci_bus_id(msk_ctx* ctx, char* buf, int buflen);
int msk_malloc(msk_ctx* ctx, size_t bytes, void** out);
int msk_free(msk_ctx* ctx, void* p);
int msk_memset(msk_ctx* ctx, void* p, int value, size_t bytes);
/* Packed-weight cache contract.  The 5x5x5 / 3x3x3 convolution pipeline keeps a transformed, packed copy of every weight
 * tensor it has seen (keyed by its device address).  Every entry point of this library that WRITES caller memory
 * (msk_h2d*, msk_d2d, msk_memset, msk_sgd_momentum, msk_adam, msk_conv_fold_bn, msk_dp_allreduce_*, msk_dp_broadcast)
 * invalidates the copies of the bytes it writes, and msk_free drops those inside the freed allocation.  A caller that
 * changes weights ANY OTHER WAY (its own kernel or hipMemcpy on the buffer, or a library pass such as msk_copy_scale /
 * msk_interp_* / msk_ndhwc_to_ncdhw aimed at a weight tensor) must report the range here before the next convolution that
 * reads it; without the call that convolution would run with the stale packed weights (forward, data gradient and the
 * fused inference path -- the weight gradient never reads packed weights).  Cheap: a host-side table walk, no launch. */
int msk_weights_changed(msk_ctx* ctx, const void* p, size_t bytes);
/* host (pageable) -> device, ordered on the compute stream; src may be reused as soon as the call returns.
 * 64 KB .. 64 MB (the per-iteration batch upload of core/train.py:122-124) go through two internal pinned staging
 * buffers and do NOT synchronise the stream; other sizes copy and synchronise. */
int msk_h2d(msk_ctx* ctx, void* dst, const void* src, size_t bytes);
int msk_d2h(msk_ctx* ctx, void* dst, const void* src, size_t bytes);  /* synchronises the stream */
/* asynchronous host->device copy from PINNED memory (msk_pinned_alloc): returns immediately, the
 * source must stay untouched until the stream passes the copy (tools/prepare.py:200-259 loader path) */
int msk_h2d_async(msk_ctx* ctx, void* dst, const void* pinned_src, size_t bytes);
int msk_d2d(msk_ctx* ctx, void* dst, const void* src, size_t bytes);
int msk_pinned_alloc(msk_ctx* ctx, size_t bytes, void** out);
int msk_pinned_free(msk_ctx* ctx, void* p);
int msk_mem_info(msk_ctx* ctx, size_t* free_bytes, size_t* total_bytes);

/* ---- timing / profiling (HIP events on the context stream) ---------------- */
int msk_timer_start(msk_ctx* ctx);               /* records an event */
int msk_timer_stop(msk_ctx* ctx, float* ms);     /* records + synchronises; elapsed ms */
/* numbered marks on the compute stream: per-step times of a
 * run without a host synchronisation inside it (the reference's per-iteration batch_cost, core/train.py:172-173)        */
int msk_mark(msk_ctx* ctx, int idx /* 0..1023 */);
int msk_mark_elapsed(msk_ctx* ctx, int a, int b, float* ms);   /* waits for mark b */
/* ctx's compute stream waits (on the device, no host synchronisation) for everything enqueued so far on `other`'s compute
 * stream.  A second context = a second stream: the in-loop preprocessing of the NEXT sample (pinned H2D -> normalise ->
 * resample, tools/prepare_mri_spine_seg.py:71-80) runs there beside the training step and is handed over with two of these
 * (tools/bench_workloads.py --inloop-preprocess).  Both contexts must be on the same device. */
int msk_ctx_wait(msk_ctx* ctx, msk_ctx* other);
/* per-kernel profile: when enabled every launch is bracketed by events and its
 * duration accumulated under the kernel's tag.  The launches of the Winograd matrix stage (wbf_gemm_*) carry their events ON
 * the dispatch (hipExtLaunchKernelGGL start / stop events) instead of between two marker packets, which idle the queue for
 * ~6 us each; options: "prof_only_halo" 1 = only those launches (bench.py's roofline kernel), "prof_paused" 1 = take no events
 * until it is set back (no drain, no host synchronisation: sampling every Nth step), "prof_attach" 0 = marker brackets. */
int msk_prof_enable(msk_ctx* ctx, int on);
int msk_prof_reset(msk_ctx* ctx);
/* writes "tag\tcalls\ttotal_ms\n" lines into buf (NUL terminated); returns needed size via *len */
int msk_prof_report(msk_ctx* ctx, char* buf, int buflen, int* len);
/* knobs (debugging / A-B measurements; 0 = the product dispatch):
 *   "conv_impl": 1=VALU reference kernels everywhere, 3=reference wgrad only, 4=reference gather-conv only,
 *     5=no LDS wgrad, 6=no k==s scatter / forward-gather kernels, 7=scatter kernel at any size and the general gather
 *     kernel for the k==s forward convs, 8=no tight-K kernel,
 *     9=one-voxel VALU kernel for 32->ncls, 10=Winograd forward kernel even for tiny grids, 11=direct (non-Winograd)
 *     forward/data-gradient kernels, 12=fp32 Winograd weight gradient forced, 13=direct weight-gradient kernels,
 *     14=fp32 Winograd F(2,5) instead of F(4,5), 16=tap-row weight-gradient kernel for the kernel == stride convolutions,
 *     20=fp32-MFMA Winograd kernels instead of the bf16x3 pipeline, 21=the same for the weight gradient only,
 *     22=VALU kernel instead of the folded-column MFMA kernel for 32->ncls (conv_foldn_k), 23=VALU kernel instead of
 *     conv_c1_mfma_k for 1->16, 24=fp32-MFMA form of conv_foldn_k instead of the fp16 two-piece form,
 *     25=fp32-MFMA tight-K kernel instead of conv_tk_h2_k for ncls->32,
 *     26=fp32-MFMA (tap, class)-row weight gradient instead of wgrad_cbs_h2_k for 32->ncls;
 *   "wino_bf3" 0|1 (0 = fp32-MFMA Winograd kernels everywhere; also env MSEGK_WBF=0), "wbf_variant" (-1 auto | tile variant
 *     of wbf_gemm_k), "wbf_tin_map" 0|1 (lane mapping of the transform kernel);
 *   "wgrad_async" 0|1 (weight gradients on the side stream), "wgrad_async_max_m" (voxel limit for it, 0 = all);
 *   "prof_shapes" 0|1, "prof_only_halo" 0|1 (profile only the 5^3 halo-conv kernels), "poison_scratch" byte|-1;
 *   "direct_conv" 0|1 (1 = no Winograd kernels; also env MSEGK_DIRECT_CONV=1);
 *   "conv_split" 3|2: operand split of the Winograd pipelines: 3 = three bf16 pieces (exact fp32 operands, six MFMAs per
 *     product), 2 = two fp16 pieces with a scaled residual (22 significand bits, three MFMAs per product; gradients are
 *     scaled by a power of two derived on the device, forward activations must stay below 3000 in magnitude);
 *   "bwd_fuse" -1|0|1|2 (msk_conv3d_bwd_bnact: auto | three calls | one dual transform | one transform per stream),
 *     "foldn_wgs" (workgroups per CU targeted by the D segmentation of conv_foldn_k, 0 = 2);
 *     "wbf_pad_min_voxels" (smallest 5^3 problem whose channel counts are not multiples of 32 that is run through the
 *         16-bit pipeline on a zero-padded copy, default 2^18);
 *     "wbf_tin_groups" (workgroups below which the pipeline's transform kernels cut their W tiles into chunks, -1 = 8 per CU, 0 = never);
 *     "tile_staging" 0|1 (dense 5..32-channel voxel records -- the 20-class 1x1x1 head and its loss kernels -- through an LDS tile with
 *         whole-line accesses; 0 = one thread per voxel straight from HBM, bitwise the same results), "kst_pair" 0|1 (gconv_kst_k:
 *         both h-parity classes of <= 16 output channels in one matrix instruction);
 *     "reduce_vpl" (voxels per lane the per-channel reduction kernels aim for before they add workgroups, default 8),
 *         "reduce_vpl_site" (site * 1000 + voxels per lane for ONE family of reductions -- 0 forward statistics, 1 BatchNorm backward
 *         sums, 2 joins, 3 channel sums; 0 = follow "reduce_vpl" -- another summation order of the same sums: the full-size parity
 *         test measures the distance between two fp32 evaluations of a step with it);
 *   tuning: "halo_tile" / "wgrad_chunk" (-1 auto or table index), "wgrad_rounds" / "wgrad_wino_rounds" (workgroups
 *   per CU targeted by the split-K of the direct / Winograd weight-gradient kernels; "wgrad_wino_rounds" 0 = pick the
 *   split count that fills whole waves of resident workgroups, the default) */
int msk_set_option(msk_ctx* ctx, const char* key, int value);
/* effective value of "dp_mode" (after MSEGK_DP_MODE and msk_dp_init's fall-backs), "conv_split", "wgrad_async", "world", "rank" */
int msk_get_option(msk_ctx* ctx, const char* key, int* value);

/* ---- layout at the boundary ------------------------------------------------ */
/* NCDHW (reference layout, core/train.py:123) <-> NDHWC (device layout) */
int msk_ncdhw_to_ndhwc(msk_ctx* ctx, const float* src, msk_tensor dst);
int msk_ndhwc_to_ncdhw(msk_ctx* ctx, msk_tensor src, float* dst);

/* ---- convolutions ---------------------------------------------------------- */
/* paddle.nn.Conv3D forward   (models/vnet.py:36,67-68,98-99,165-166,169)
 *   y[N,OD,OH,OW,Cout] = conv(x[N,ID,IH,IW,Cin], w[Cout,Cin,k]) + bias        */
int msk_conv3d_fwd(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w,
                   const float* bias /*nullable*/, msk_tensor y);
/* Inference path (core/infer.py:62-94, core/val.py:89-110 run the net in eval mode): the chain
 * Conv3D -> BatchNorm3D(running statistics) -> PReLU of vnet.py:36-41 as ONE convolution.
 *   msk_conv_fold_bn:    w'[co][...] = w[co][...] * scale[co];  b'[co] = b[co] * scale[co] + shift[co]
 *                        (scale/shift from msk_bn_eval_coeffs; w canonical [Cout][inner = Cin * taps]; bias nullable)
 *   msk_conv3d_fwd_act:  y = PReLU_slope(conv(x, w) + bias); the Winograd kernels apply the slope in their epilogue,
 *                        every other kernel is followed by one in-place pass (same result).  slope nullable.        */
int msk_conv_fold_bn(msk_ctx* ctx, const float* w, const float* bias, const float* scale, const float* shift,
                     int cout, long inner, float* w_folded, float* b_folded);
int msk_conv3d_fwd_act(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias,
                       const float* prelu_slope, msk_tensor y);
/* Training-path forms of Conv3D forward / weight gradient for the conv -> BatchNorm units of vnet.py:36-41:
 *   msk_conv3d_fwd_ex:   as msk_conv3d_fwd; additionally
 *       stats_local (nullable): BatchNorm statistics of y (the msk_bn_stats record, mean[Cout] then M2[Cout]) -- taken in
 *                    the convolution's output stage when the kernel supports it (no second read of y), else by msk_bn_stats;
 *       xform (nullable): caller-owned device buffer of msk_conv3d_xform_bytes() bytes that receives the transformed input
 *                    (Winograd B^T x, split into bf16 pieces) the convolution computes anyway; it stays valid while x is
 *                    unchanged.  Pass it only when msk_conv3d_xform_bytes() > 0.  For the 32 -> ncls <= 3 class
 *                    (out_tr.conv1, vnet.py:165) the buffer is 512 bytes: only max |x|, which the fp16 two-piece forward
 *                    kernel measures anyway, travels to the weight gradient.
 *   msk_conv3d_wgrad_ex: as msk_conv3d_wgrad; xform (nullable) = the buffer msk_conv3d_fwd_ex filled for the SAME x,
 *                    which saves recomputing the transform of x (the layer input kept for backward, vnet.py:41).
 *   msk_conv3d_xform_bytes: size of that buffer for input x and cout output channels; 0 = the convolution does not
 *                    produce one (pass NULL).                                                              */
size_t msk_conv3d_xform_bytes(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, int cout);
int msk_conv3d_fwd_ex(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias /*nullable*/,
                      msk_tensor y, float* stats_local /*nullable*/, void* xform /*nullable*/);
int msk_conv3d_wgrad_ex(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy, float* dw, float* db /*nullable*/,
                        int accumulate, const void* xform /*nullable*/);
/* "amax arrays" (round 3): the fp16 two-piece convolution pipeline scales every operand tensor by a power of two taken from
 * its max |value|.  For a layer input that maximum can ride along in the pass that PRODUCES the tensor (bn1/relu1 of the
 * previous LUConv, vnet.py:41; the residual joins :110-111,154; the Dropout3D copies :102,147-148) instead of costing a
 * read of its own:
 *   msk_amax_new            n consecutive zeroed device arrays of 64 floats whose maximum counts (from a ring: valid for
 *                           the next ~500 requests, i.e. well beyond the training step that uses them);
 *   msk_*_amax              as the entry point without the suffix, and max |written values| is folded into out_amax
 *                           (nullable; several calls may fold into the same array: the two halves of a concat buffer);
 *   msk_conv3d_fwd_ex2      as msk_conv3d_fwd_ex with x_amax (nullable) = such an array covering all of x.        */
float* msk_amax_new(msk_ctx* ctx, int n /* consecutive arrays */);
int msk_conv3d_fwd_ex2(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias /*nullable*/,
                       msk_tensor y, float* stats_local /*nullable*/, void* xform /*nullable*/, const float* x_amax /*nullable*/);
int msk_affine_act_fwd_amax(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                            const float* alpha, msk_tensor out, float* out_amax);
int msk_affine_act_join_fwd_amax(msk_ctx* ctx, msk_tensor y, const float* scale, const float* shift, const float* alpha_inner,
                                 msk_tensor res, const float* alpha_outer, msk_tensor out, float* out_amax);
int msk_copy_scale_amax(msk_ctx* ctx, msk_tensor src, const float* mask, msk_tensor dst, int accumulate, float* dst_amax);
/* the gradient side of the same idea: dy_amax (nullable) = the amax array the pass that wrote dy folded max |dy| into
 * (msk_affine_act_bwd_apply_amax); msk_conv3d_wgrad_ex2 otherwise as msk_conv3d_wgrad_ex.                        */
int msk_conv3d_dgrad_ex(msk_ctx* ctx, msk_conv_desc cd, msk_tensor dy, const float* w, msk_tensor dx, int accumulate,
                        const float* dy_amax /*nullable*/);
int msk_conv3d_wgrad_ex2(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy, float* dw, float* db /*nullable*/,
                         int accumulate, const void* xform /*nullable*/, const float* dy_amax /*nullable*/);
/* ... and x_amax (nullable): the amax array of x from the pass that produced it, used when no kept transform brings the scale */
int msk_conv3d_wgrad_ex3(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy, float* dw, float* db /*nullable*/,
                         int accumulate, const void* xform /*nullable*/, const float* dy_amax /*nullable*/, const float* x_amax /*nullable*/);
/* Launch diet (round 3): the per-channel epilogues of a BatchNorm layer ride in the merge kernels that precede them.
 *   msk_bn_fin              the arguments of msk_bn_finalize(world = 1, count) as a struct;
 *   msk_conv3d_fwd_ex3      msk_conv3d_fwd_ex2, and with fin != NULL (stats_local required) the finalisation runs in the
 *                           launch that merges the statistics: bitwise the results of msk_conv3d_fwd_ex2 + msk_bn_finalize.
 *                           (SyncBatchNorm over several ranks keeps the separate calls: an all-gather sits between them.)
 *   msk_bn_stats_fin        msk_bn_stats + msk_bn_finalize(world = 1) the same way (the up-convolutions, vnet.py:150);
 *   msk_affine_act_bwd_reduce_pg / msk_add_act_join_bwd_pg
 *                           as the _ex forms, and the parameter gradients msk_affine_act_param_grads would take from the
 *                           sums are ADDED to dgamma / dbeta / dalpha (nullable) in the launch that merges the sums;
 *                           clear_maxes = 0: `maxes` arrives zeroed (msk_amax_new(ctx, 2)), no memset is enqueued.        */
typedef struct msk_bn_fin {
  const float* gamma; /* nullable */
  const float* beta;  /* nullable */
  float eps, momentum;
  double count;       /* values per channel (N*D*H*W) */
  float* running_mean; /* nullable */
  float* running_var;  /* nullable */
  float* save_mean;
  float* save_invstd;
  float* scale;        /* NULL = no finalisation */
  float* shift;
} msk_bn_fin;
int msk_conv3d_fwd_ex3(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias /*nullable*/,
                       msk_tensor y, float* stats_local /*nullable*/, void* xform /*nullable*/, const float* x_amax /*nullable*/,
                       const msk_bn_fin* fin /*nullable*/);
int msk_bn_stats_fin(msk_ctx* ctx, msk_tensor x, float* stats_local, const msk_bn_fin* fin /*nullable*/);
/* Convolution followed by INSTANCE statistics (paddle.nn.InstanceNorm3D: per (sample, channel) over D*H*W; the builder-defined
 * UNet3D of BASELINE configs[3]): stats is [N][2 Cout]; fin describes sample 0 (count = D*H*W, running_* ignored when NULL) and
 * sample n's save_mean / save_invstd / scale / shift lie fin_stride floats further per sample.  When the convolution kernel
 * keeps per-tile records (the one-kernel matrix stage, <= 64 channels) the N merges read those; otherwise one statistics pass
 * per sample runs on y -- the results are the same to fp32 rounding either way.                                              */
int msk_conv3d_fwd_in(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias /*nullable*/, msk_tensor y,
                      float* stats, void* xform /*nullable*/, const float* x_amax /*nullable*/, const msk_bn_fin* fin,
                      int fin_stride);
int msk_affine_act_bwd_reduce_pg(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                                 const float* alpha, const float* mean, const float* invstd, msk_tensor dout, float* sums,
                                 float* maxes /*nullable*/, int clear_maxes, float* dgamma, float* dbeta, float* dalpha);
int msk_add_act_join_bwd_pg(msk_ctx* ctx, msk_tensor y, const float* scale, const float* shift, const float* alpha_inner,
                            msk_tensor res, const float* alpha_outer, const float* mean, const float* invstd, msk_tensor dout,
                            msk_tensor da, msk_tensor dres, int dres_accumulate, float* dalpha_outer, float* unit_sums,
                            float* maxes /*nullable*/, int clear_maxes, float* unit_dgamma, float* unit_dbeta, float* unit_dalpha);
/* Backward of one conv -> BatchNorm(batch statistics) -> PReLU unit (LUConv, vnet.py:36-41; autograd of core/train.py:139)
 * in one call:   dy = msk_affine_act_bwd_apply(y, ..., dout, sums_total, M_total, bn_mode 1, no residual),
 *                dx (+)= conv^T(dy, w)   (dx.p NULL -> skipped),   dw (+)= sum dy * x.
 * When both gradients run on the transform pipeline (square 5^3 / 3^3 'same' layers; msk_conv3d_bwd_bnact_bytes() > 0) dy is
 * evaluated inside the kernel that writes its two transforms and never reaches HBM: then `ybuf` (caller-owned,
 * msk_conv3d_bwd_bnact_bytes() bytes, must stay untouched until the weight gradient -- possibly on the side stream -- has
 * run, i.e. until the next msk_sync / optimizer step) receives the second transform and `dy_scratch` is not written.
 * Otherwise (ybuf NULL, or the shape is not eligible) the three operations run one after the other with dy in `dy_scratch`
 * (a tensor of y's shape, required either way).  xform as in msk_conv3d_wgrad_ex.  Results agree with the separate calls
 * to fp32 rounding.  A convolution with ONE input channel and dx.p NULL (in_tr.conv1, vnet.py:67) evaluates dy inside its
 * weight-gradient kernel as well (no ybuf / xform / maxes needed; dy_scratch is not written), on the calling stream.  */
size_t msk_conv3d_bwd_bnact_bytes(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor y);
int msk_conv3d_bwd_bnact(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                         const float* shift, const float* alpha /*nullable*/, const float* mean, const float* invstd,
                         const float* gamma, msk_tensor dout, const float* sums_total, double M_total,
                         msk_tensor dy_scratch, msk_tensor dx, int dx_accumulate, float* dw, int dw_accumulate,
                         const void* xform /*nullable*/, void* ybuf /*nullable*/,
                         const float* maxes /*nullable: msk_affine_act_bwd_reduce_ex's, needed by the fused forms under "conv_split" 2*/);
/* msk_conv3d_bwd_bnact for the layer behind a zero-copy concat (UpTransition.ops[0]: x = the concat buffer, vnet.py:152-154).
 * dx accumulates into the interleaved gradient buffer as usual UNLESS the one-kernel matrix stage runs the data gradient: then
 * the sums (old dx + new) are STORED to the two dense half tensors dx_lo / dx_hi (channels [0, c/2) / [c/2, c), voxel stride
 * c/2) and *split_done = 1 -- the consumers of the halves (the up-convolution's and the skip's backward) read dense voxels
 * instead of one half of every 128-byte line.  *split_done = 0: dx holds the result, dx_lo / dx_hi are untouched. */
int msk_conv3d_bwd_bnact_split(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                               const float* shift, const float* alpha /*nullable*/, const float* mean, const float* invstd,
                               const float* gamma, msk_tensor dout, const float* sums_total, double M_total,
                               msk_tensor dy_scratch, msk_tensor dx, int dx_accumulate, float* dw, int dw_accumulate,
                               const void* xform /*nullable*/, void* ybuf /*nullable*/, const float* maxes /*nullable*/,
                               msk_tensor dx_lo, msk_tensor dx_hi, int* split_done);
/* msk_conv3d_bwd_bnact(_split) whose accumulating data gradient takes its OLD values from another tensor, dx_old (geometry and voxel
 * stride of dx), and writes the sums to dx (or to the dense halves dx_lo / dx_hi when the one-kernel matrix stage runs it:
 * *split_done = 1).  The residual joins of vnet.py:110-111,154 hand one gradient to both operands: msk_add_act_bwd /
 * msk_add_act_join_bwd_* accept a null second destination, and the layer behind the join reads the first one here -- one full
 * write of the tensor less per join.  dx_old.p == NULL: msk_conv3d_bwd_bnact(_split).  dx_lo / dx_hi / split_done may be null. */
int msk_conv3d_bwd_bnact_acc(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                             const float* shift, const float* alpha, const float* mean, const float* invstd, const float* gamma,
                             msk_tensor dout, const float* sums_total, double M_total, msk_tensor dy_scratch, msk_tensor dx,
                             int dx_accumulate, float* dw, int dw_accumulate, const void* xform, void* ybuf, const float* maxes,
                             msk_tensor dx_old, msk_tensor dx_lo, msk_tensor dx_hi, int* split_done);
/* Backward of an up-convolution unit  convT -> BatchNorm(batch statistics) -> PReLU  (UpTransition.up_conv / bn1 / relu1,
 * vnet.py:133-150; autograd of core/train.py:139) behind its reduce pass, in one call:
 *     dx (+)= convT^T(dy, w),  dw (+)= sum x * dy,   dy = msk_affine_act_bwd_apply(y, ..., dout, sums_total, M_total, bn_mode 1)
 * where the data gradient evaluates dy from (y, dout) in its own loads (<= 16 output channels of the convT, kernel == stride)
 * and the pass that writes dy_scratch runs on the weight-gradient stream in front of the weight gradient, its only reader.
 * Returns 0 = done, 1 = not eligible (NOTHING was launched: use msk_affine_act_bwd_apply + msk_convT3d_wgrad / _dgrad). */
int msk_convT3d_bwd_bnact(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                          const float* shift, const float* alpha /*nullable*/, const float* mean, const float* invstd,
                          msk_tensor dout, const float* sums_total, double M_total, msk_tensor dy_scratch, msk_tensor dx,
                          int dx_accumulate, float* dw, int dw_accumulate);
/* The in_tr.conv1 unit (vnet.py:57-79: out = PReLU(BN(conv5^3(x: ONE channel)) + tile(x)), no data gradient): weight gradient
 * with dy evaluated inside the kernel from (y, dout); `res` is either a null tensor or x itself (the tiled residual, read from the
 * kernel's own halo of x).  Returns 0 = done, 1 = not this class / declined (nothing was launched: use msk_affine_act_bwd_apply +
 * msk_conv3d_wgrad), < 0 = error.  Runs on the calling stream.                                                     */
/* The same unit with INSTANCE statistics (conv -> InstanceNorm -> PReLU; builder-defined UNet3D): sample n's scale / shift / mean /
 * invstd lie n * coef_stride floats after the given pointers, its sums (msk_affine_act_bwd_reduce* of that sample) n * sums_stride
 * floats, M_sample = D*H*W.  Only the fused form exists: returns 1 with NOTHING launched when it is not eligible (the caller then
 * runs msk_affine_act_bwd_apply per sample and the separate gradient calls), 0 when done.  maxes: the two amax arrays the
 * reduce passes of ALL samples folded into (required for the fp16 operand formats).                                         */
int msk_conv3d_bwd_inact(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                         const float* shift, const float* alpha /*nullable*/, const float* mean, const float* invstd, int coef_stride,
                         msk_tensor dout, const float* sums, int sums_stride, double M_sample, msk_tensor dx, int dx_accumulate,
                         float* dw, int dw_accumulate, const void* xform, void* ybuf, const float* maxes /*nullable*/);
int msk_conv3d_bwd_bnact_c1(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor y, const float* scale, const float* shift,
                            const float* alpha /*nullable*/, const float* mean, const float* invstd, msk_tensor res,
                            msk_tensor dout, const float* sums_total, double M_total, float* dw, int dw_accumulate);
/* autograd of the above (core/train.py:139 loss.backward()):
 *   dx (+)= conv^T(dy, w);  accumulate != 0 adds into dx                       */
int msk_conv3d_dgrad(msk_ctx* ctx, msk_conv_desc cd, msk_tensor dy, const float* w,
                     msk_tensor dx, int accumulate);
/*   dw[Cout,Cin,k] (+)= sum_voxels dy * x ;  db[Cout] (+)= sum dy  (db nullable) */
int msk_conv3d_wgrad(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy,
                     float* dw, float* db /*nullable*/, int accumulate);
/* paddle.nn.Conv3DTranspose forward (models/vnet.py:133-137), w[Cin,Cout,k],
 * out = (in-1)*s + k                                                           */
int msk_convT3d_fwd(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w,
                    const float* bias /*nullable*/, msk_tensor y);
/* the same + the BatchNorm statistics of y for the unit UpTransition.up_conv -> bn1 (vnet.py:133-150): stats_local [2 C] = (mean, M2)
 * of this rank's values, fin (nullable) = msk_bn_finalize(world 1) in the merge launch -- taken in the convolution's store pass where
 * the kernel can (no separate read of y), by msk_bn_stats_fin otherwise.  Same results to fp32 rounding. */
int msk_convT3d_fwd_ex(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias, msk_tensor y,
                       float* stats_local, const msk_bn_fin* fin /*nullable*/);
int msk_convT3d_dgrad(msk_ctx* ctx, msk_conv_desc cd, msk_tensor dy, const float* w,
                      msk_tensor dx, int accumulate);
int msk_convT3d_wgrad(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy,
                      float* dw, float* db /*nullable*/, int accumulate);

/* ---- BatchNorm3D / SyncBatchNorm + PReLU + residual ------------------------- */
/* paddle.nn.BatchNorm3D -> SyncBatchNorm statistics (models/vnet.py:38,70,100,139,167;
 * cvlibs/config.py:322).  Welford/Chan merge, biased variance over N*D*H*W.
 * stats_local[2*C] = {mean[C], M2[C]} of THIS rank's x.                        */
int msk_bn_stats(msk_ctx* ctx, msk_tensor x, float* stats_local);
/* Merge `world` rank-local stats (gathered[world][2*C], equal counts `count_per_rank`)
 * into mean/var, write scale = gamma*invstd, shift = beta - mean*scale, save
 * mean[C], invstd[C]; running <- momentum*running + (1-momentum)*batch when
 * running_mean/var are non-NULL.                                               */
int msk_bn_finalize(msk_ctx* ctx, const float* gathered, int world, double count_per_rank,
                    int C, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, float* save_mean,
                    float* save_invstd, float* scale, float* shift);
/* eval mode: scale/shift from running statistics (model.eval(), core/val.py:57) */
int msk_bn_eval_coeffs(msk_ctx* ctx, int C, const float* gamma, const float* beta,
                       const float* running_mean, const float* running_var, float eps,
                       float* save_mean, float* save_invstd, float* scale, float* shift);
/* out = prelu(scale[c]*x + shift[c] + res, alpha[c])
 *   scale/shift NULL -> identity affine;  res.p NULL -> no residual;
 *   res.c < out.c -> residual channel = c % res.c (x.tile, vnet.py:78);
 *   alpha NULL -> no activation.  Replaces bn1/relu1/relu2/paddle.add chains
 *   (vnet.py:41,77-79,107-111,150-154,173).                                    */
int msk_affine_act_fwd(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift,
                       msk_tensor res, const float* alpha, msk_tensor out);
/* the same with the maximum of |out| folded into an amax array (msk_affine_act_fwd_amax) AND a second copy of the result in out2
 * (nullable; its own voxel stride): InputTransition writes its output into the skip half of the up-transition's concat buffer
 * (vnet.py:152) and, for the readers of that half alone, as a dense tensor -- a 16-channel half of a 32-channel voxel is 64 of
 * every 128-byte line.  Channel-stationary float4 kernel only (C / 4 a power of two). */
int msk_affine_act_fwd_amax2(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                             const float* alpha, msk_tensor out, float* out_amax /*nullable*/, msk_tensor out2);
/* backward pass 1: per-channel sums over this rank:
 *   sums[0..C)   = sum du            (du = dout * (u>0 ? 1 : alpha))
 *   sums[C..2C)  = sum du * xhat     (xhat = (x-mean)*invstd; 0 if no BN)
 *   sums[2C..3C) = sum dout * u * [u<=0]   (d alpha)                           */
int msk_affine_act_bwd_reduce(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift,
                              msk_tensor res, const float* alpha, const float* mean,
                              const float* invstd, msk_tensor dout, float* sums);
/* the same; maxes (nullable, device float[2][64]: two "amax arrays" -- the maximum of each row counts, the kernel spreads
 * its atomics over the entries) additionally receives max |du| and max |xhat| over the tensor -- the bound
 * msk_conv3d_bwd_bnact needs to scale dy into fp16 range when option "conv_split" is 2.  Requires C % 4 == 0, voxel
 * strides % 4 == 0 and 16-byte aligned tensors.                                                              */
int msk_affine_act_bwd_reduce_ex(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift,
                                 msk_tensor res, const float* alpha, const float* mean,
                                 const float* invstd, msk_tensor dout, float* sums, float* maxes /*nullable*/);
/* backward pass 2: dx = BN-backward(du) and dres (+)= du.
 *   bn_mode 0: no BN (dx = du); 1: training BN (uses sums, total count M over all
 *   ranks); 2: eval BN (dx = scale*du).  dres.p NULL -> skipped; dres_acc adds.   */
int msk_affine_act_bwd_apply(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift,
                             msk_tensor res, const float* alpha, const float* mean,
                             const float* invstd, const float* gamma, msk_tensor dout,
                             const float* sums_total, double M_total, int bn_mode,
                             msk_tensor dx, msk_tensor dres, int dres_acc);
/* the same; dx_amax (nullable, an amax array of msk_amax_new) additionally receives max |dx| -- the gradient kernels of the
 * convolution in front (msk_conv3d_dgrad_ex / msk_conv3d_wgrad_ex2) then skip their own pass over dx.  One array may collect
 * several calls (per-sample InstanceNorm units): the maximum is associative.                                    */
int msk_affine_act_bwd_apply_amax(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift,
                                  msk_tensor res, const float* alpha, const float* mean,
                                  const float* invstd, const float* gamma, msk_tensor dout,
                                  const float* sums_total, double M_total, int bn_mode,
                                  msk_tensor dx, msk_tensor dres, int dres_acc, float* dx_amax /*nullable*/);
/* parameter gradients of the above from the (all-reduced) sums:
 *   dgamma (+)= sums[C..2C), dbeta (+)= sums[0..C), dalpha (+)= sums[2C..3C)     */
int msk_affine_act_param_grads(msk_ctx* ctx, int C, const float* sums, float* dgamma,
                               float* dbeta, float* dalpha, int accumulate);

/* ---- concat / dropout / small elementwise ---------------------------------- */
/* dst = src * mask[n*C+c] (mask NULL -> 1): paddle.concat slices (vnet.py:152) and
 * nn.Dropout3D with a given mask (vnet.py:103,144-145);  accumulate adds.       */
int msk_copy_scale(msk_ctx* ctx, msk_tensor src, const float* mask, msk_tensor dst, int accumulate);
/* counter-based Dropout3D mask: mask[n*C+c] = keep ? 1/(1-p) : 0, keyed by
 * (seed, step, site).                                                          */
int msk_dropout_mask(msk_ctx* ctx, uint64_t seed, uint64_t step, uint32_t site, int count,
                     float p, float* mask);
/* out[c] (+)= sum over voxels of x[..,c]  (bias gradients)                      */
int msk_channel_sum(msk_ctx* ctx, msk_tensor x, float* out, int accumulate);
/* paddle.argmax(logit, axis=1) (core/infer.py:92) */
int msk_argmax_c(msk_ctx* ctx, msk_tensor x, int32_t* out);
/* out[v][c] = softmax over c of x[v][:]  -- F.softmax(logits, axis=1) of the AUC path of evaluate (core/val.py:121-123)  */
int msk_softmax_c(msk_ctx* ctx, msk_tensor x, msk_tensor out);

/* ---- loss ------------------------------------------------------------------ */
/* losses/loss_utils.py:31-40 class_weights: w_c = sum(1-softmax_c)/sum(softmax_c) */
int msk_class_weights(msk_ctx* ctx, msk_tensor logits, float* weights);
/* Fused CrossEntropyLoss + DiceLoss forward (losses/cross_entropy_loss.py:47-87,
 * losses/dice_loss.py:76-102).  out[0]=CE, out[1]=dice loss, out[2..2+C)=per-channel
 * dice; stats (device, 3*C+2 doubles) keeps {I_c, S_c, T_c, ce_num, ce_den} for bwd. */
int msk_loss_fwd(msk_ctx* ctx, msk_tensor logits, const int32_t* labels, const float* weights,
                 int ignore_index, float* out, double* stats);
/* dlogits = coef_ce * dCE/dz + coef_dice * dDice/dz */
int msk_loss_bwd(msk_ctx* ctx, msk_tensor logits, const int32_t* labels, const float* weights,
                 int ignore_index, const double* stats, float coef_ce, float coef_dice,
                 msk_tensor dlogits);

/* DiceLoss(sigmoid_norm=False) and DiceLoss(weight=...) (losses/dice_loss.py:36-43,68-69): dice_softmax != 0
 * normalises the dice term's probabilities with softmax over the classes instead of the sigmoid; dice_weight
 * (device, [C], or NULL) multiplies each class's intersection.  The CE term is unchanged. */
int msk_loss_fwd_ex(msk_ctx* ctx, msk_tensor logits, const int32_t* labels, const float* weights,
                    int ignore_index, int dice_softmax, const float* dice_weight, float* out, double* stats);
int msk_loss_bwd_ex(msk_ctx* ctx, msk_tensor logits, const int32_t* labels, const float* weights,
                    int ignore_index, int dice_softmax, const float* dice_weight, const double* stats,
                    float coef_ce, float coef_dice, msk_tensor dlogits);

/* ---- optimizer --------------------------------------------------------------- */
/* paddle.optimizer.Momentum(momentum, weight_decay=L2) over one flat arena
 * (cvlibs/config.py:212-214): g += wd*p; v = mu*v + g; p -= lr*v.
 * grad_scale multiplies g first (1/nranks after a sum all-reduce).              */
int msk_sgd_momentum(msk_ctx* ctx, float* param, const float* grad, float* velocity,
                     size_t count, float lr, float momentum, float weight_decay,
                     float grad_scale);

/* The same update for ONE slice of the arena as soon as everything that produces its gradients has been enqueued
 * (core/train.py:139-140: loss.backward(); optimizer.step() -- nothing reads a block's weights between its data gradient and the
 * next forward).  The update and the re-pack of the slice's convolution weights run at the end of the internal weight-gradient
 * stream, behind an event on the calling stream's current tail; the calling stream does not wait.  Opt-in
 * (optimizer.Momentum.enable_eager): parameters change during backward, and the gradients must be final when the call is made
 * (one rank, or after the slice's all-reduce).  msk_sgd_momentum_finish joins: call it where optimizer.step() stands. */
int msk_sgd_momentum_eager(msk_ctx* ctx, float* param, const float* grad, float* velocity,
                           size_t count, float lr, float momentum, float weight_decay,
                           float grad_scale);
int msk_sgd_momentum_finish(msk_ctx* ctx);

/* paddle.optimizer.Adam(beta1, beta2, epsilon, weight_decay=L2) over one flat arena (cvlibs/config.py:214-216):
 * g = grad_scale*grad + wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= lr sqrt(1-beta2_pow)/(1-beta1_pow) * m / (sqrt(v) + epsilon sqrt(1-beta2_pow));
 * beta*_pow = beta*^t of THIS step (t = 1 on the first call); the caller keeps the powers. */
int msk_adam(msk_ctx* ctx, float* param, const float* grad, float* moment1, float* moment2,
             size_t count, float lr, float beta1, float beta2, float epsilon, double beta1_pow,
             double beta2_pow, float weight_decay, float grad_scale);

/* ---- preprocessing (tools/preprocess_utils) ----------------------------------- */
/* geometry.py:31-69 resample == scipy.ndimage.zoom(order 0|1, grid_mode=False):
 * align-corner coordinate map; order 0 = floor(c+0.5), order 1 = trilinear.
 * dtype: 0 = float32, 1 = int32 (labels).                                       */
int msk_resample3d(msk_ctx* ctx, const void* src, int sd, int sh, int sw, void* dst,
                   int dd, int dh, int dw, int order, int dtype);
/* ---- loader augmentations on the device (medicalseg/transforms) ------- */
/* functional.py:103-110 resized_crop_3d: crop box (i,j,k)+(cd,ch,cw) of a (sd,sh,sw) volume,
 * zoomed to (dd,dh,dw) with msk_resample3d's coordinate map (order 1 image / 0 label,
 * transform.py:334-339).                                                              */
int msk_crop_resample3d(msk_ctx* ctx, const void* src, int sd, int sh, int sw, int i, int j,
                        int k, int cd, int ch, int cw, void* dst, int dd, int dh, int dw,
                        int order, int dtype);
/* functional.py:80-88 flip_3d (np.flip along axis 0|1|2); out of place               */
int msk_flip3d(msk_ctx* ctx, const void* src, void* dst, int d, int h, int w, int axis,
               int dtype);
/* functional.py:91-100 rotate_3d == scipy.ndimage.rotate(angle, axes=(axis_a, axis_b),
 * order, mode='constant', cval, reshape=False): affine map about the plane centre in
 * double, no interpolation beyond the edges, int32 rounds half away from zero.       */
int msk_rotate3d(msk_ctx* ctx, const void* src, void* dst, int d, int h, int w, int axis_a,
                 int axis_b, double angle_deg, int order, double cval, int dtype);
/* values.py:67-87 HUnorm */
int msk_hu_norm(msk_ctx* ctx, const float* src, float* dst, size_t count, float hu_min,
                float hu_max, float hu_nan);
/* values.py:54-64 normalize; use_bounds==0 -> min/max of the volume */
int msk_minmax_norm(msk_ctx* ctx, const float* src, float* dst, size_t count, int use_bounds,
                    float min_val, float max_val);
/* transforms/transform.py:67-69: im / im.max() when max > 0 */
int msk_max_norm(msk_ctx* ctx, const float* src, float* dst, size_t count);
/* values.py:37-51 label_remap (sequential key->value passes) */
int msk_label_remap(msk_ctx* ctx, int32_t* label, size_t count, const int32_t* keys,
                    const int32_t* vals, int npairs);

/* Bias gradient of a convolution that feeds a BatchNorm, from the sums msk_affine_act_bwd_reduce
 * already produced (no extra pass over dy): with batch statistics sum_v dy[v][c] is identically 0
 * (the BN backward removes the mean) -- callers skip db there; with running statistics (eval-mode
 * BN, vnet.py:351-397 alignment runs) it is scale[c] * sums[c].                          */
int msk_bn_bias_grad(msk_ctx* ctx, int C, const float* sums, const float* scale, float* dbias,
                     int accumulate);
/* Adjoint of the residual join out = prelu(a + b, alpha) (vnet.py:110-111,154) in ONE pass:
 * da = dout * prelu'(a+b); db (+)= the same; dalpha[c] += sum dout*(a+b) over a+b <= 0.
 * (A join has no BatchNorm, so its data gradient needs no reduction first.)  float4-aligned
 * tensors with C % 4 == 0.                                                              */
int msk_add_act_bwd(msk_ctx* ctx, msk_tensor a, msk_tensor b, const float* alpha, msk_tensor dout,
                    msk_tensor da, msk_tensor db, int db_accumulate, float* dalpha);
/* The residual join right behind a conv -> BatchNorm -> PReLU unit (vnet.py:107-111, 150-154: the last LUConv of a stage and
 * relu2(add(out, down))) in ONE pass over the convolution output y, without materialising the unit's activation:
 *   out = prelu(prelu(scale*y + shift, alpha_inner) + res, alpha_outer)            (saves 8 bytes per element of traffic)
 * and its backward: da = gradient w.r.t. the unit's (never stored) activation, dres (+)= the same, dalpha_outer += ...;
 * the unit's own backward (msk_affine_act_bwd_reduce / msk_conv3d_bwd_bnact with dout = da) follows unchanged.        */
int msk_affine_act_join_fwd(msk_ctx* ctx, msk_tensor y, const float* scale, const float* shift, const float* alpha_inner,
                            msk_tensor res, const float* alpha_outer, msk_tensor out);
int msk_add_act_join_bwd(msk_ctx* ctx, msk_tensor y, const float* scale, const float* shift, const float* alpha_inner,
                         msk_tensor res, const float* alpha_outer, msk_tensor dout, msk_tensor da, msk_tensor dres,
                         int dres_accumulate, float* dalpha_outer);
/* the same pass additionally leaves the UNIT's backward sums (what msk_affine_act_bwd_reduce_ex(y, ..., dout = da) would
 * compute: unit_sums[0..3C) = sum du, sum du*xhat, d alpha_inner; unit_sums needs 4*C floats) and maxes (nullable, as in
 * msk_affine_act_bwd_reduce_ex): the unit's reduce pass is not needed afterwards.                                      */
int msk_add_act_join_bwd_ex(msk_ctx* ctx, msk_tensor y, const float* scale, const float* shift, const float* alpha_inner,
                            msk_tensor res, const float* alpha_outer, const float* mean, const float* invstd, msk_tensor dout,
                            msk_tensor da, msk_tensor dres, int dres_accumulate, float* dalpha_outer, float* unit_sums,
                            float* maxes /*nullable*/);

/* ELUCons(elu=True) (vnet.py:25-29): paddle.nn.ELU(alpha) as its own pass (the shipped configs use PReLU, which is fused
 * into the BatchNorm kernels above).  out = x > 0 ? x : alpha*(exp(x)-1), in place allowed;
 * dx (+)= dout * (out > 0 ? 1 : out + alpha) -- the derivative from the OUTPUT. */
int msk_elu_fwd(msk_ctx* ctx, msk_tensor x, float alpha, msk_tensor out);
int msk_elu_bwd(msk_ctx* ctx, msk_tensor out, msk_tensor dout, float alpha, msk_tensor dx, int accumulate);

/* ---- deep supervision (models/vnet_deepsup.py:266-277) -------------- */
/* F.interpolate(d, size=x.shape[2:], mode='trilinear') of a conv3^3 head: align_corners=
 * False, align_mode=0 [PADDLE]: per axis ratio = in/out, src = max(ratio*(o+0.5)-0.5, 0),
 * i0 = floor(src), i1 = min(i0+1, in-1).  src/dst: NDHWC with equal n and c.         */
int msk_interp_trilinear_fwd(msk_ctx* ctx, msk_tensor src, msk_tensor dst);
/* Adjoint: ddst (gradient at the resized size) -> dsrc (head resolution); separable
 * gather passes, deterministic.  scratch: device buffer of >= msk_interp_scratch_bytes
 * (may be NULL when only the depth axis is resized).                                  */
int msk_interp_scratch_bytes(msk_ctx* ctx, msk_tensor src, msk_tensor dst, size_t* bytes);
int msk_interp_trilinear_bwd(msk_ctx* ctx, msk_tensor ddst, msk_tensor dsrc, int accumulate,
                             void* scratch, size_t scratch_bytes);

/* ---- data parallel (RCCL over xGMI; core/train.py:81-85 fleet DataParallel) ---- */
#define MSK_UNIQUE_ID_BYTES 128
int msk_dp_unique_id(char* id128);                       /* rank 0 */
int msk_dp_rccl_version(int* version);                   /* ncclGetVersion: e.g. 22703 for 2.27.3; no context, no GPU needed */
int msk_dp_init(msk_ctx* ctx, const char* id128, int rank, int world);
int msk_dp_allreduce_sum(msk_ctx* ctx, float* buf, size_t count);   /* joins the weight-gradient side stream first */
/* SyncBatchNorm backward sums (2*C floats, produced on the main stream): same reduction WITHOUT the join,
 * so the 24 per-step statistics exchanges do not serialise the side stream */
int msk_dp_allreduce_stats(msk_ctx* ctx, float* buf, size_t count);
/* Gradient buckets (the Paddle reducer's role behind core/train.py:82-85).  Where the sum runs depends on the arrangement
 * chosen with msk_set_option(ctx, "dp_mode", m) / env MSEGK_DP_MODE before msk_dp_init (measurements: msk_dp.hip):
 *   0 (default) on the compute stream, like msk_dp_allreduce_sum -- callers then send the whole buffer once after backward;
 *   2           on the context's COMMUNICATION stream with its own communicator (ncclCommSplit of the first): it starts after
 *               everything enqueued so far on the compute and weight-gradient streams, blocks neither, and never queues
 *               behind the SyncBatchNorm exchanges of the compute stream;
 *   1 / 3       on the communication stream with the one communicator (1: the statistics exchanges run there too).
 * msk_dp_wait makes the compute stream wait for all outstanding buckets (call it before the optimizer); msk_sync implies it. */
int msk_dp_allreduce_async(msk_ctx* ctx, float* buf, size_t count);
int msk_dp_wait(msk_ctx* ctx);
int msk_dp_allgather(msk_ctx* ctx, const float* send, float* recv, size_t count_per_rank);
int msk_dp_broadcast(msk_ctx* ctx, float* buf, size_t count, int root);
int msk_dp_barrier(msk_ctx* ctx);
int msk_dp_destroy(msk_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MSEGK_H */
