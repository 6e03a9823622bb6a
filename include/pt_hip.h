/*
 * pt_hip.h - C ABI of libpt_hip.so: the MI355X (gfx950) implementation of the
 * Point-Teacher teacher->student training hot path.
 *
 * The reference (ZhuHaoranEIS/Point-Teacher) is 100 % Python; what its hot path
 * calls natively are `mmcv.ops` (un-vendored) and long chains of small torch kernels.
 * Every entry point below replaces ONE such call site; the comment above each names
 * it as file:line relative to /root/reference/HBB_TOD/mmdet/.  INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless marked [host]; fp32 / int32,
 *     densely packed, row-major; nothing is allocated or freed inside a call;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls
 *     only enqueue work - no host synchronisation, safe under hipGraph capture;
 *   - return value: 0 on success, a negative PT_E* code on a rejected argument,
 *     a positive hipError_t if a launch failed; pt_last_error() gives the text;
 *   - images of a batch are concatenated; `off[B+1]` (int32, device) holds the
 *     prefix offsets of per-image ground truths (off[0] = 0);
 *   - index outputs use 0 = background / i+1 = gt i of that image
 *     (AssignResult.gt_inds convention, core/bbox/assigners/assign_result.py).
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_EINVAL (-1)  /* bad size / null pointer / unsupported parameter */
#define PT_ELIMIT (-2)  /* size above a compiled-in limit (message says which) */

const char* pt_last_error(void);
/* ABI version, bumped when a signature changes. */
int pt_abi_version(void);

/* ------------------------------------------------------------------ assigners --
 * TopkAssigner.assign with num_pre <= topk (every shipped config), i.e.
 * core/bbox/assigners/topk_assigner.py:120-144: per gt the `num_pre` L1-nearest
 * points (PointCost, match_costs/match_cost.py:188-214), later gt wins.
 * points[P,2]; gt_xy[sumG,2]; off[B+1]; out gt_inds[B*P] (int32).
 * If cand != NULL it receives the candidate rows [sumG,num_pre] (int32).
 * gt_valid (uint8[sumG], may be NULL): gts with 0 are skipped but keep their index, which
 * equals assigning the order-preserving filtered list (strong_augmentation drops gts that
 * leave the crop, detectors/syn_images_generator_v2.py:77-90) without a host round trip.
 * Ties in distance are broken towards the LOWEST point index. num_pre <= 8. */
int pt_topk_assign(const float* points, int P, const float* gt_xy, const uint8_t* gt_valid,
                   const int32_t* off, int B, int sumG, int num_pre, int32_t* gt_inds,
                   int32_t* cand, void* stream);

/* FUSETopkAssigner.assign (core/bbox/assigners/fuse_topk_assigner.py:91-118) fed the
 * way _gnerate_pseudo_single does (dense_heads/fcos_head_p2b_ts.py:747-752):
 * reg[B*P,4] are (l,t,r,b) distances decoded with `points`; cls[B*P,C] logits;
 * cost = FocalLossCost*cls_w (match_cost.py:54-99) + InsiderCost*loc_w (:217-252).
 * num_pre <= 8, topk < num_pre or >= num_pre (then every candidate is taken).
 * cand[sumG,num_pre] (int32, required) receives the stage-1 candidates. */
int pt_fuse_assign(const float* points, int P, const float* reg, const float* cls, int C,
                   const float* gt_xy, const int32_t* gt_labels, const int32_t* off, int B,
                   int sumG, int num_pre, int topk, float cls_w, float reg_w, float loc_w,
                   int32_t* gt_inds, int32_t* cand, void* stream);

/* Score-weighted pseudo-box fusion, fcos_head_p2b_ts.py:763-789.
 * Uses gt_inds/cand from pt_fuse_assign.  Outputs per gt: pseudo_bboxes[sumG,4]
 * (8x8 box around the point when no grid point was assigned), pseudo_points[sumG,2],
 * pseudo_scores[sumG], nassigned[sumG] (int32) and iou_with_gt[sumG] (IoU with
 * gt_bboxes, 0 where nassigned == 0; its mean over assigned gts is `mean_ious_pred`). */
int pt_pseudo_boxes(const float* points, int P, const float* reg, const float* cls, int C,
                    const float* gt_xy, const int32_t* gt_labels, const float* gt_bboxes,
                    const int32_t* off, int B, int sumG, int num_pre, const int32_t* gt_inds,
                    const int32_t* cand, float* pseudo_bboxes, float* pseudo_points,
                    float* pseudo_scores, int32_t* nassigned, float* iou_with_gt, void* stream);

/* FCOS target build, fcos_head_p2b_ts.py:586-603 and :669-706: labels[B*P] (int32,
 * background = num_classes) and bbox_targets[B*P,4] = (l,t,r,b) w.r.t. the assigned
 * box (box 0 of the image for unassigned points, zeros when the image has no box).
 * If ctr_target != NULL it receives centerness_target (:1019-1038) for assigned
 * points and 0 elsewhere. */
int pt_fcos_targets(const float* points, int P, const int32_t* gt_inds, const float* boxes,
                    const int32_t* box_labels, const int32_t* off, int B, int num_classes,
                    int32_t* labels, float* bbox_targets, float* ctr_target, void* stream);

/* ---------------------------------------------------------------------- losses --
 * mmcv.ops.sigmoid_focal_loss, call site models/losses/focal_loss.py:85 (the
 * importable twin py_sigmoid_focal_loss :11-56 is the oracle): element-wise
 * loss[N,C]; labels[N] int32 in [0,C] (C = background); weight[N] may be NULL.
 * fwd writes loss[N,C] (if non-NULL) and block partial sums partial[pt_focal_nblocks(N,C)];
 * bwd writes grad[N,C] = scale[0] * weight[n] * dloss/dlogit (scale: device scalar). */
int pt_focal_nblocks(int N, int C);
int pt_sigmoid_focal_loss_fwd(const float* logits, const int32_t* labels, const float* weight,
                              int N, int C, float gamma, float alpha, float* loss,
                              float* partial, void* stream);
int pt_sigmoid_focal_loss_bwd(const float* logits, const int32_t* labels, const float* weight,
                              const float* scale, int N, int C, float gamma, float alpha,
                              float* grad, void* stream);

/* diou_loss (models/losses/iou_loss.py:139-189) and the min-over-9-shifted-targets
 * part of DN_diou_loss (:414-463) in one pass over pred/target[N,4] (xyxy):
 * diou[N] and dnmin[N] (dnmin may be NULL).  bwd: grad_pred[N,4] =
 * g_diou[n]*d diou/d pred + g_dn[n]*d dnmin/d pred (either g may be NULL).
 * torch autograd tie conventions are kept (max/min ties split 0.5, clamp passes at 0). */
int pt_diou_fwd(const float* pred, const float* target, int N, float eps, float hyper,
                float* diou, float* dnmin, void* stream);
int pt_diou_bwd(const float* pred, const float* target, const float* g_diou, const float* g_dn,
                int N, float eps, float hyper, float* grad_pred, void* stream);

/* bbox_overlaps (core/bbox/iou_calculators/iou2d_calculator.py:74-260).
 * mode 0 iou, 1 iof, 2 giou.  aligned: out[M]; pairwise: out[M,N]. */
int pt_bbox_overlaps_aligned(const float* a, const float* b, int M, int mode, float eps,
                             float* out, void* stream);
int pt_bbox_overlaps_pairwise(const float* a, const float* b, int M, int N, int mode, float eps,
                              float* out, void* stream);

/* DeltaXYWHBBoxCoder.decode with means 0 / stds 1 (coder/delta_xywh_bbox_coder.py:144-270,
 * call site fcos_head_p2b_ts.py:1210) fwd and bwd (w.r.t. deltas).  max_h/max_w <= 0: no clip. */
int pt_delta2bbox_fwd(const float* rois, const float* deltas, int N, float max_h, float max_w,
                      float wh_ratio_clip, float* out, void* stream);
int pt_delta2bbox_bwd(const float* rois, const float* deltas, const float* grad_out, int N,
                      float max_h, float max_w, float wh_ratio_clip, float* grad_deltas,
                      void* stream);

/* -------------------------------------------------------------------- RoIAlign --
 * mmcv.ops.RoIAlign(output_size, spatial_scale, sampling_ratio=0, pool_mode='avg',
 * aligned=True), built models/roi_heads/roi_extractors/base_roi_extractor.py:53-58,
 * called fcos_head_p2b_ts.py:1202,1243,1268.  rois[K,5] = (batch, x1,y1,x2,y2).
 * feat is [B,C,H,W] when channels_last == 0 and [B,H,W,C] when 1 (same for grad_feat);
 * out / grad_out are always [K,C,out,out] (the layout the FC stack flattens).
 * bwd ACCUMULATES into grad_feat (zero it first).  `group` (>= 1) says how many CONSECUTIVE RoIs belong together
 * (the U2 shaken boxes of one MIL bag overlap): the channels_last out-7 kernels give a workgroup a run of such RoIs
 * (the largest divisor of `group` up to 16); a run whose taps fall on <= 5x5 feature pixels is reduced in registers,
 * any other RoI takes the separable path inside the same launch.  Any group value gives the same result. */
int pt_roi_align_fwd(const float* feat, const float* rois, int B, int C, int H, int W, int K,
                     int out_size, float spatial_scale, int sampling_ratio, int aligned,
                     int channels_last, int group, float* out, void* stream);
/* The out-7 channels_last forward writing the RoI blocks as three bf16 planes (x = x0 + x1 + x2 exactly; row-major
 * [3][(K + 1) * C * 49], row K zeros) - the operand format of the FC stack's matrix kernel (pt_conv_bf16x6 with the RoIs as pixels
 * of a 1 x 1 convolution, fcos_head_p2b_ts.py:1202-1236): neither the fp32 block [K, C, 7, 7] nor a split pass over it touches HBM.
 * Same values as pt_roi_align_fwd (the planes sum to them bit for bit).  feat [B,H,W,C]; C * 49 a multiple of 8. */
int pt_roi_align_fwd_planes(const float* feat, const float* rois, int B, int C, int H, int W, int K, float spatial_scale,
                            int sampling_ratio, int aligned, int group, uint16_t* planes, int64_t plane_stride,
                            void* stream);
/* The same block as TWO fp16 planes (value = h0 + h1, 22 significant bits; [2][(K + 1) * C * 49]): the operand of the MIL head's
 * first FC layer on three fp16 MFMA products (pt_conv_desc.operand_f16).  RoI features are O(1) .. O(100): inside fp16's range. */
int pt_roi_align_fwd_planes_f16(const float* feat, const float* rois, int B, int C, int H, int W, int K, float spatial_scale,
                                int sampling_ratio, int aligned, int group, uint16_t* planes, int64_t plane_stride,
                                void* stream);
int pt_roi_align_bwd(const float* grad_out, const float* rois, int B, int C, int H, int W, int K,
                     int out_size, float spatial_scale, int sampling_ratio, int aligned,
                     int channels_last, int group, float* grad_feat, void* stream);

/* ------------------------------------------------------------------ MIL bags --
 * fine_proposals_from_cfg (detectors/syn_images_generator_v2.py:262-324) for a batch:
 * boxes[sumG,4] -> props[sumG*U,4], valid[sumG*U] (uint8, IoF with the image > 0.7),
 * U = R*R*(1+4*S) with R = n_ratios, S = n_shake.  ratios/shake are [host] arrays. */
int pt_fine_proposals(const float* boxes, int sumG, const float* ratios, int n_ratios,
                      const float* shake, int n_shake, float min_scale, float img_h, float img_w,
                      float* props, uint8_t* valid, void* stream);

/* gen_negative_proposals (syn_images_generator_v2.py:234-259): u[B,4,n] uniforms
 * (the four torch.rand draws), pos[sum_pos,4] with pos_off[B+1];
 * neg[B*n,4], neg_ok[B*n] (uint8: IoU < thr with every positive of its image). */
int pt_negative_proposals(const float* u, int B, int n, const float* pos, const int32_t* pos_off,
                          float img_h, float img_w, float iou_thr, float* neg, uint8_t* neg_ok,
                          void* stream);

/* mil_bag_training + gfocal_loss (fcos_head_p2b_ts.py:1147-1180, :1074-1078).
 * cls/ins[NB,U2,C] (NB = num_gt*U1 bags), valid[NB*U2] uint8, labels[NB] int32.
 * fwd: bag_loss[NB] (weighted gfocal summed over classes, 0 for empty bags) and
 * bag_valid[NB] (uint8).  bwd: grads = scale[0] * d(sum bag_loss)/d(cls, ins). U2 <= 1024. */
int pt_mil_bag_loss_fwd(const float* cls, const float* ins, const uint8_t* valid,
                        const int32_t* labels, int NB, int U2, int C, float* bag_loss,
                        uint8_t* bag_valid, void* stream);
int pt_mil_bag_loss_bwd(const float* cls, const float* ins, const uint8_t* valid,
                        const int32_t* labels, const float* scale, int NB, int U2, int C,
                        float* grad_cls, float* grad_ins, void* stream);
/* negative bags, same file :1169-1179: loss[M] = gfocal(sigmoid(x), 0, w) summed over C. */
int pt_mil_neg_loss_fwd(const float* neg_cls, const uint8_t* neg_w, int M, int C, float* loss,
                        void* stream);
int pt_mil_neg_loss_bwd(const float* neg_cls, const uint8_t* neg_w, const float* scale, int M,
                        int C, float* grad, void* stream);

/* mil_bag_selection (+_single) fcos_head_p2b_ts.py:1112-1145, :1092-1110: per gt the
 * top-k of sigmoid(cls)[label]*norm-softmax(ins)[label] over the whole U1*U2 bag,
 * weighted box mean, clamp to the image, (1-beta)*box + beta*pseudo. topk <= 8. */
int pt_mil_bag_select(const float* cls, const float* ins, const uint8_t* valid,
                      const int32_t* labels, const float* bags, const float* pseudo, int NG,
                      int U1, int U2, int C, int topk, float beta, float img_h, float img_w,
                      float* merged, void* stream);

/* ------------------------------------------------------- EMA / optimizer step --
 * update_teacher_model (detectors/fcos_p2b_teacher_student.py:254-257) over ONE flat
 * parameter buffer: teacher = alpha*teacher + one_minus_alpha*student (the caller rounds
 * 1-alpha from double exactly like `t.mul_(a).add_(1 - a, s)` does). */
int pt_ema_update(float* teacher, const float* student, int64_t n, float alpha,
                  float one_minus_alpha, void* stream);
/* squared L2 norm of a flat gradient buffer -> partial[pt_sqnorm_nblocks(n)] */
int pt_sqnorm_nblocks(int64_t n);
int pt_sqnorm_partial(const float* g, int64_t n, float* partial, void* stream);
/* mmcv OptimizerHook grad-clip (max_norm, L2) + torch.optim.SGD(momentum, weight_decay)
 * with the paramwise_cfg of aitodv2_point_teacher_0%.py:212-215 (bias lr x2, bias decay 0):
 * elements [0,split) are weights, [split,n) biases.  sqnorm[0] = total squared grad
 * norm (device scalar, already reduced over ranks if any); lr is a device scalar.
 * Since ABI 3 the kernel moves 16 bytes per lane: param / grad / momentum_buf must be 16-byte aligned and `split` a multiple of
 * 4 (pad the weight segment), else PT_EINVAL - the two-group form of pt_sgd_step_groups below, same preconditions. */
int pt_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, int64_t split,
                const float* lr, float momentum, float weight_decay, float bias_lr_mult,
                float bias_decay_mult, const float* sqnorm, float max_norm, int first_step,
                void* stream);

/* The same step under the full paramwise_cfg of mmcv's DefaultOptimizerConstructor (bias_lr_mult, bias_decay_mult,
 * norm_decay_mult, dwconv_decay_mult, dcn_offset_lr_mult, custom_keys - e.g. configs/baselines/aitodv2_yolof_r50_1x.py:70-71:
 * norm_decay_mult=0, custom_keys={'backbone': lr_mult 1/3}): the buffer is n_groups <= PT_MAX_PARAM_GROUPS contiguous
 * segments, group g = [group_end[g-1], group_end[g]) with lr = lr*lr_mult[g], decay = weight_decay*decay_mult[g].
 * group_end / lr_mult / decay_mult are [host] arrays of n_groups entries (copied into the launch); ends are multiples of 4,
 * the last equals n.  Parameters that never receive a gradient (torch.optim.SGD skips `grad is None`: no decay, no momentum)
 * are simply not part of the buffer. */
#define PT_MAX_PARAM_GROUPS 8
int pt_sgd_step_groups(float* param, const float* grad, float* momentum_buf, int64_t n,
                       const int64_t* group_end, const float* lr_mult, const float* decay_mult,
                       int n_groups, const float* lr, float momentum, float weight_decay,
                       const float* sqnorm, float max_norm, int first_step, void* stream);

/* ------------------------------------------------ fp32 GEMM on the bf16 matrix cores --
 * The MIL FC stacks (dense_heads/fcos_head_p2b_ts.py:1202-1236, :1240-1256: `self.shared_fcs_reg/_bag[stage]` = Linear
 * 12544 -> 1024 -> 1024 + ReLU over K RoIs, and their backward) are fp32 by the config's definition.  gfx950's fp32 MFMA runs at
 * 1/16 of its bf16 rate; an fp32 value is exactly x0 + x1 + x2 with three bf16 terms, and the six leading cross products with
 * fp32 accumulation reproduce the fp32 product to ~2^-26 (csrc/gemm_split.hip).
 *
 * pt_split_bf16x3: src[R, C] fp32 (row stride ld) -> three bf16 planes (planes + p * plane_stride; bf16 stored as uint16_t) of
 * the GEMM operand  transpose == 0: rows = R, k = C      transpose != 0: rows = C, k = R
 * in the BLOCKED layout [ceil(rows / 16)][ceil(k / 32)][16 rows][4 k-slots][8]: each 16 x 32 block is one contiguous KiB in
 * the order the GEMM's LDS image wants it (k-slot XOR (row >> 2) & 3), zero padded; pt_split_bf16x3_plane_elems(rows, k) =
 * elements of one plane.  The reduce dimension of an operand thus becomes block-contiguous whatever the layout of the fp32 tensor. */
int64_t pt_split_bf16x3_plane_elems(int rows, int k);
int pt_split_bf16x3(const float* src, int64_t ld, int R, int C, int transpose, uint16_t* planes,
                    int64_t plane_stride, void* stream);
/* c[M, N] (row stride ldc) = A[M, K] * B[N, K]^T (+ bias[N]) (ReLU if relu != 0) from the planes of pt_split_bf16x3 (A: rows M,
 * k K; B: rows N, k K).  tile_rows = rows of the output tile (96, 128, ..., 256; 0 = pt_gemm_bf16x6_tile_rows(M, N), the height
 * that fills 256 CUs best).  A torch.nn.Linear maps to it as
 *   forward  y  = x W^T:   A = split(x),      B = split(W)          dgrad  dx = dy W:  A = split(dy),  B = split(W, transpose)
 *   wgrad    dW = dy^T x:  A = split(dy, transpose), B = split(x, transpose). */
int pt_gemm_bf16x6_tile_rows(int M, int N);
int pt_gemm_bf16x6_nt(const uint16_t* a_planes, int64_t a_plane_stride, const uint16_t* b_planes,
                      int64_t b_plane_stride, float* c, int64_t ldc, const float* bias, int M, int N,
                      int K, int relu, int tile_rows, void* stream);
/* EXPERIMENT, not on the training path: the same product from fp16 x 2 operands - x = h0 + h1 (two fp16
 * planes in the blocked layout of pt_split_bf16x3; 22 significant bits, <= 3e-8 absolute below 0.125) and THREE fp16 MFMA products
 * (a0 b0 + a0 b1 + a1 b0) instead of six bf16 ones.  Operands must be scaled into fp16's range by the caller.  Replaces nothing in
 * the reference; measured against pt_gemm_bf16x6_nt by tools/gemm_bench.py. */
int pt_split_f16x2(const float* src, int64_t ld, int R, int C, uint16_t* planes, int64_t plane_stride,
                   void* stream);
int pt_gemm_f16x3_nt(const uint16_t* a_planes, int64_t a_plane_stride, const uint16_t* b_planes,
                     int64_t b_plane_stride, float* c, int64_t ldc, const float* bias, int M, int N, int K,
                     int relu, int tile_rows, void* stream);

/* The 3 x 3, stride 1, pad 1 convolutions of the dense head's towers (anchor_free_head.py:198-219; fp32 by the config) as an
 * implicit GEMM on the same kernel: out[B*H*W, Cout] (NHWC, row stride ldo) = conv(x, w) (* scale[Cout]) (+ bias[Cout]) (ReLU);
 * scale / bias = the (scale, shift) of a frozen eval-mode BatchNorm behind the convolution (backbones/resnet.py:262-303) or NULL.
 *   x_planes: pt_split_bf16x3_rows of the [B*H*W, Cin] NHWC activations - ROW-MAJOR planes [3][(P + 1) * Cin], row P = zeros
 *             (what a tap outside the image reads); nothing like im2col is ever stored;
 *   w_planes: pt_split_bf16x3 (transpose == 0) of the [Cout, 9 * Cin] weight matrix, k = (ky, kx, cin) - a channels_last weight
 *             [Cout, Cin, 3, 3] is that matrix.  The input gradient is the same call on the output gradient with the weights
 *             w'[cin, (2 - ky, 2 - kx), cout]; the weight gradient is pt_conv3x3_wgrad_bf16x6_nhwc.  Cin % 32 == 0. */
/* pt_split_bf16x3_rows: optional backward preparation in the same pass - relu_of (the forward output [P, C]; the gradient is
 * zeroed where it is <= 0), col_scale [C] (the frozen BatchNorm's scale), masked_out (fp32 [P, C]: the effective gradient the
 * library's weight-gradient kernel reads); any of the three may be NULL. */
int pt_split_bf16x3_rows(const float* src, int64_t ld, int P, int C, const float* relu_of, const float* col_scale,
                         float* masked_out, uint16_t* planes, int64_t plane_stride, void* stream);
int pt_conv3x3_bf16x6_nhwc(const uint16_t* x_planes, int64_t x_plane_stride, const uint16_t* w_planes, int64_t w_plane_stride,
                           float* out, int64_t ldo, const float* bias, const float* scale, int B, int H, int W, int Cin,
                           int Cout, int relu, int tile_rows, void* stream);

/* Weight gradient of the same convolution (replaces the convolution_backward the towers' autograd reaches from
 * anchor_free_head.py:198-219): dw[Cout][3][3][Cin] (the memory of a channels_last [Cout, Cin, 3, 3] weight)
 *   = sum over pixels p of gy[p][cout] * x[p + (ky - 1) * W + (kx - 1)][cin]   (taps outside the image contribute nothing).
 * gy_planes / x_planes: pt_split_bf16x3_rows planes of the [B*H*W, Cout] output gradient (the ones the input gradient used) and of
 * the [B*H*W, Cin] activations (the ones the forward used) - the pixel index is their row index; the kernel reads both
 * column-wise out of LDS (ds_read_b64_tr_b16), nothing is transposed in memory.  The pixels are cut into `splits` chunks
 * (0 = pt_conv3x3_wgrad_bf16x6_splits); partial sums go to `workspace` (>= splits * Cout * 9 * Cin floats) and are added in a
 * fixed order: deterministic, no atomics.  Cin % 128 == 0, Cout % 128 == 0. */
int pt_conv3x3_wgrad_bf16x6_splits(int B, int H, int W, int Cin, int Cout);
int pt_conv3x3_wgrad_bf16x6_nhwc(const uint16_t* gy_planes, int64_t gy_plane_stride, const uint16_t* x_planes,
                                 int64_t x_plane_stride, float* dw, float* workspace, int64_t workspace_elems, int B,
                                 int H, int W, int Cin, int Cout, int splits, void* stream);

/* ---------------------------------------------- plane-native convolutions (ABI 4) --
 * The same implicit GEMM for every convolution of the trunk: the Bottlenecks' 1 x 1 / 3 x 3 convolutions with their frozen BatchNorm,
 * identity add and ReLU (backbones/resnet.py:262-303; `caffe` style puts the stride on conv1, :153-158), FPN's laterals and output
 * convolutions (necks/fpn.py:151-202), PSAGG's 1 x 1 convolutions (necks/ps_fpn.py:56-75) and the dense head's towers
 * (anchor_free_head.py:198-219) - and for their input gradients (the same call on the output gradient with mode-1 weight planes).
 * Activations travel between layers as ROW-MAJOR split planes [3][(pixels + 1) * C] (x = x0 + x1 + x2 exactly, last row zeros) so
 * that no layer re-splits its input and no fp32 copy of an intermediate activation is written:
 *   acc[p][o]  = sum over taps, cin of x[(y * stride - pad + ky, x * stride - pad + kx)][cin] * w[o][ky][kx][cin]     (zero outside)
 *   v          = acc * scale[o] + shift[o]  (+ res_planes[p][o] summed)  (+ res_f32[p][o])     -> ReLU when relu != 0
 *   v          = 0 where mask_planes' plane 0 is <= 0   (the ReLU mask of the tensor whose gradient is being formed)
 *   out_f32[p][o] = v   and / or   out_planes = split(v) (row M = zeros).
 * scatter_stride == 2: output row (b, y, x) is written to row (b * scatter_H + 2 y) * scatter_W + 2 x of mask / out_f32 / out_planes -
 * the input gradient of a stride-2 1 x 1 convolution computed on the coarse grid and placed into a buffer the caller zeroed
 * (res_* stay indexed by the coarse row).  Every pointer of the epilogue may be NULL; at least one output is required.
 * w_planes: blocked planes of [Cout][KH * KW * Cin] (pt_conv_weight_planes_batch / pt_split_bf16x3).  Cin % 32 == 0, Cout % 8 == 0.
 * Split-k: a launch with few output tiles and a long reduce dimension (the teacher's batch of 2; layer4's 25 x 25 maps) cuts the k-steps
 * into `splits` chunks (0 = pt_conv_bf16x6_splits(...) when a workspace is given, else 1), every (tile, chunk) workgroup stores its
 * raw fp32 tile to workspace[splits][M][Cout] and a second launch adds them in a fixed order and runs the epilogue (deterministic).
 * dstride == 2: the INPUT GRADIENT of a 3 x 3 stride-2 convolution (a `pytorch`-style Bottleneck's conv2, resnet.py:153-158) as a
 * transposed convolution: x_planes = the output gradient on the coarse grid [B, Hs, Ws], the result lives on the convolution's
 * input grid [B, out_H, out_W]; stride = 1, pad = KH - 1 - (forward padding), w_planes in the mode-1 form; taps whose coordinate is
 * odd read the zero row.  dstride == 0 / 1: off.
 * np == 1: every operand and result is ONE bf16 plane - a bf16 NHWC tensor with a zero row behind it - and one MFMA product
 * replaces the six: the same kernels as the bf16 trunk of BASELINE configs[2] (bf16 operands, fp32 accumulation, one rounding per
 * fused epilogue).  np == 0 / 3: three planes.
 * tile_rows: 64, 96, ..., 256 or 0 = pt_gemm_bf16x6_tile_rows.  [host] struct; device pointers inside. */
typedef struct {
  int32_t B, Hs, Ws, Cin, Cout, KH, KW, stride, pad;
  int32_t relu;
  const uint16_t* x_planes;
  int64_t x_plane_stride;
  const uint16_t* w_planes;
  int64_t w_plane_stride;
  const float* scale;
  const float* shift;
  const uint16_t* res_planes;
  int64_t res_plane_stride;
  const float* res_f32;
  const uint16_t* mask_planes;
  float* out_f32;
  uint16_t* out_planes;
  int64_t out_plane_stride;
  int32_t scatter_stride, scatter_H, scatter_W;
  int32_t tile_rows;
  float* workspace;
  int64_t workspace_elems;
  int32_t splits;
  int32_t dstride, out_H, out_W;
  int32_t np;
  int32_t operand_f16;   /* != 0: x_planes / w_planes are TWO fp16 planes each (value = h0 + h1), three fp16 MFMA products per fp32
                          * product instead of six bf16 ones; the epilogue's planes (res / mask / out) stay bf16 x np.  The caller
                          * keeps the operands inside fp16's range: power-of-two scales, undone by `alpha` (see pt_planes_to_f16) */
  float alpha;           /* the accumulator is multiplied by alpha first (0 = 1): the operands' power-of-two scales */
  int32_t reserved;
  const float* alpha_dev; /* optional DEVICE scalar multiplied into alpha (1 / the scale pt_planes_to_f16 chose for a gradient) */
  int32_t out_f16;        /* out_planes are two fp16 planes IN THE SCALED FORMAT ("H2", ABI 6): [2][out_plane_stride] fp16 with
                          * out_plane_stride >= (rows + 1) * Cout + 8; the fp32 word at element (rows + 1) * Cout of plane 0 (the
                          * "tail") holds 1 / s, the represented value is (h0 + h1) / s.  The launch writes the tail: a copy of
                          * *out_inv_scale_src, or 1 when that pointer is NULL */
  int32_t res_f16;        /* res_planes are two fp16 planes (value = (h0 + h1) * *res_alpha_dev, or h0 + h1 when that is NULL) */
  const float* res_alpha_dev;
  const float* out_inv_scale_src; /* a GRADIENT CHAIN keeps the scale of its first link: the caller leaves the operand's 1 / s out of
                          * alpha (the result then IS s * gradient) and passes the operand's tail here, so the output carries it on */
  int32_t* census;        /* optional DEVICE int32[4] of the fp16 output (out_f16): [0] += elements that saturated (|v| > 60 000),
                          * [1] = max(bits of the largest stored magnitude), and with census_mode 2 also [2] += non-zero elements
                          * below 0.125 (second term subnormal) and [3] += elements written.  Mode 1 costs nothing measurable: the
                          * atomics are issued per wavefront and only when they would change the word */
  int32_t census_mode;    /* 0 / 1: saturation + maximum; 2: all four words (a debugging census) */
  int32_t reserved2;
} pt_conv_desc;
int pt_conv_bf16x6_splits(int B, int Hs, int Ws, int Cin, int Cout, int KH, int KW, int stride, int pad, int tile_rows);
/* ABI 7: the chunk count the launch takes for THIS descriptor when it is given a workspace and splits <= 0 (the tile height and the
 * k-step costs depend on np / operand_f16, which the shape-only query above cannot see): size the workspace with it. */
int pt_conv_bf16x6_plan(const pt_conv_desc* desc);
int pt_conv_bf16x6(const pt_conv_desc* desc, void* stream);

/* Weight (and bias) gradient of the same convolutions: dw[Cout][KH][KW][Cin] (+)= row_scale[o] * sum over output pixels p of
 * gy[p][o] * x[src(p, tap)][cin]; dbias[o] (+)= sum_p gy[p][o] (formed by the same launch: one more MFMA operand of ones, no extra
 * pass).  gy_planes: row-major planes of the [B*Ho*Wo, Cout] output gradient (what the input-gradient call read), x_planes: of the
 * [B*Hs*Ws, Cin] activations (what the forward read).  accumulate != 0: added to dw / dbias (a parameter's gradient buffer) instead
 * of stored.  workspace >= splits * (Cout * KH * KW * Cin [+ Cout]) floats; fixed-order reduction: deterministic, no atomics.
 * Cin % 128 == 0, Cout % 128 == 0.  [host] struct; device pointers inside. */
typedef struct {
  int32_t B, Hs, Ws, Cin, Cout, KH, KW, stride, pad;
  int32_t accumulate;
  const uint16_t* gy_planes;
  int64_t gy_plane_stride;
  const uint16_t* x_planes;
  int64_t x_plane_stride;
  float* dw;
  float* dbias;
  const float* row_scale;
  float* workspace;
  int64_t workspace_elems;
  int32_t splits;
  int32_t np;
  int32_t operand_f16;   /* gy_planes / x_planes are two fp16 planes each (as in pt_conv_desc) */
  float alpha;           /* dw and dbias are multiplied by alpha (0 = 1) */
  const float* alpha_dev; /* optional DEVICE scalar multiplied into alpha (the tail of a scaled fp16 gradient: 1 / its scale) */
  const float* alpha_dev2; /* a second one (the tail of scaled fp16 activations); ABI 6 */
} pt_conv_wgrad_desc;
int pt_conv_wgrad_bf16x6_splits(int B, int Ho, int Wo, int KH, int KW, int Cin, int Cout);
/* Trainable BatchNorm (eval-mode statistics) behind a convolution - OBB config 5, `norm_cfg=dict(type='BN', requires_grad=True)`,
 * `norm_eval=True` (OBB_TOD/configs/point teacher/sodaa_fcos_pointteacher_1x.py:36-38): after pt_conv_wgrad_bf16x6 with
 * row_scale == NULL and dbias = sum_e, this turns the raw weight gradient dw[Cout][rowlen] into the three parameter gradients
 *   dgamma[o] = rstd[o] * (<w[o], dw[o]> - mean[o] * sum_e[o])      dbeta[o] = sum_e[o]      dw[o] *= scale[o]  (in place)
 * (sum_p e * conv = <w[o], dw[o]> because conv = w x: no pass over the activations).  rowlen = KH * KW * Cin, a multiple of 4. */
int pt_bn_wgrad_finish(float* dw, const float* w, int Cout, int rowlen, const float* scale, const float* rstd, const float* mean,
                       const float* sum_e, float* dgamma, void* stream);
int pt_conv_wgrad_bf16x6(const pt_conv_wgrad_desc* desc, void* stream);

/* NHWC [B, Hs, Ws, C] (pixel stride ld; fp32, or bf16 when src_bf16 != 0) -> row-major planes of the pixels (y * stride, x * stride):
 * [np][(B * Ho * Wo + 1) * C], Ho = (Hs - 1) / stride + 1, last row zeros.  np = 3: fp32 -> x0 + x1 + x2; np = 1: the bf16 rounding
 * (fp32 source) or a copy (bf16 source: the autocast stem's output entering the bf16 trunk).  stride 2 = the pixels layer2's
 * first Bottleneck reads from the frozen layer1 (resnet.py:153-158). */
int pt_split_bf16x3_gather(const void* src, int src_bf16, int64_t ld, int B, int Hs, int Ws, int C, int stride, int np,
                           uint16_t* planes, int64_t plane_stride, void* stream);
/* out = split(m * (a + b + c)) element-wise on row-major planes (n elements per plane, n % 8 == 0; np = 3 or 1 planes each): a, b
 * planes (b may be NULL), c fp32 (may be NULL), m = (mask's plane 0 > 0) or 1 when mask == NULL; out (planes) and / or out_f32.  The
 * gradient of an activation with several consumers (a stage output feeding the next stage and an FPN lateral), or planes -> fp32. */
int pt_planes_combine(const uint16_t* a, int64_t a_plane_stride, const uint16_t* b, int64_t b_plane_stride, const float* c,
                      const uint16_t* mask, int64_t n, int np, uint16_t* out, int64_t out_plane_stride, float* out_f32, void* stream);
/* fp16 operands for the largest products of the path (the MIL head's first FC layer, fcos_head_p2b_ts.py:1202-1236: 12 544 -> 1 024
 * over K RoIs): three bf16 planes (value = x0 + x1 + x2) -> TWO fp16 planes of scale * value (h0 + h1: 22
 * significant bits, <= 3e-8 absolute below 0.125; saturated at +-60 000), same row-major layout, n elements per plane.  scale > 0: the
 * caller's power of two.  scale == 0: chosen on the device - the power of two that brings the tensor's largest magnitude into
 * [512, 1024) (gradient magnitudes depend on the loss normalisation: 1e-3 ... 1e-9 per element) - and written with its reciprocal to
 * auto_scale[0], auto_scale[1] (device; the consumers' `alpha_dev` = auto_scale + 1); workspace: 1024 device floats.  Two launches. */
int pt_planes_to_f16(const uint16_t* planes, int64_t plane_stride, int64_t n, float scale, uint16_t* out,
                     int64_t out_stride, float* auto_scale, float* workspace, void* stream);

/* ABI 6 - the scaled fp16 x 2 plane format ("H2") as THE activation / gradient format of the trainable trunk, necks and heads
 * (backbones/resnet.py:262-303, necks/fpn.py:151-202, necks/ps_fpn.py:56-75, dense_heads/anchor_free_head.py:198-219,
 * fcos_head_p2b_ts.py:1202-1256): 4 bytes per element instead of the 6 of three bf16 planes, three MFMA products instead of six.
 *
 * pt_planes_mix: out = fmt_out(so * m * (ia * a + ib * b + c)) element-wise over n elements (n % 8 == 0) - format conversion, the
 * exact addition of the gradients of an activation with two consumers, the entry of a gradient into an fp16 chain.
 *   a, b: row-major planes; *_fmt: PT_FMT_F32 (a plain fp32 array; the pointer is reinterpreted), PT_FMT_BF16 (one bf16 plane),
 *         PT_FMT_H2 (two fp16 planes, ia / ib = *a_inv / *b_inv - their tails - or 1 when NULL), PT_FMT_BF16X3; b may be NULL.
 *   c: fp32 or NULL.  mask: plane 0 of any 16-bit plane set (m = plane 0 > 0) or NULL; relu_of: fp32 (m = relu_of > 0) or NULL.
 *   out (format out_fmt, may be NULL) and / or out_f32.  For out_fmt == PT_FMT_H2 the output scale `so` follows scale_mode:
 *     PT_SCALE_ONE      so = 1;
 *     PT_SCALE_AUTO     the power of two that brings the largest magnitude of the result into [128, 256) (a first launch reduces
 *                       the maximum per workgroup into `workspace`, >= 1024 floats; a zero / non-finite tensor keeps 1): ~250 x
 *                       of headroom for a gradient that grows along its chain, 500 x above the census' floor of 0.25;
 *     PT_SCALE_MERGE    1 / max(ia, ib): two chains meet, the result carries the smaller of their scales (no overflow);
 *   and the tail out_plane_stride - 8 ... is written with 1 / so: out_plane_stride >= n + 8, the tail sits at element n of plane 0.
 *   n_valid <= n (a multiple of 8): elements [n_valid, n) are written as zeros without reading any source - the zero row behind
 *   an fp32 source that has none.  census: as in pt_conv_desc (mode 1), or NULL. */
#define PT_FMT_F32 0
#define PT_FMT_BF16 1
#define PT_FMT_H2 2
#define PT_FMT_BF16X3 3
#define PT_SCALE_ONE 0
#define PT_SCALE_AUTO 1
#define PT_SCALE_MERGE 2
int pt_planes_mix(const void* a, int a_fmt, int64_t a_plane_stride, const float* a_inv, const void* b, int b_fmt, int64_t b_plane_stride,
                  const float* b_inv, const float* c, const uint16_t* mask, const float* relu_of, int64_t n, int64_t n_valid, void* out, int out_fmt,
                  int64_t out_plane_stride, int scale_mode, float* out_f32, float* workspace, int32_t* census, void* stream);
/* fp32 NHWC [B, Hs, Ws, C] (pixel stride ld) -> H2 planes of the pixels (y * stride, x * stride) with a zero last row and tail 1:
 * the frozen stem's output entering the fp16 trunk (resnet.py:153-158 caffe style: layer2's first Bottleneck reads through its
 * stride).  plane_stride >= (B * Ho * Wo + 1) * C + 8. */
int pt_split_gather_h2(const float* src, int64_t ld, int B, int Hs, int Ws, int C, int stride, uint16_t* planes, int64_t plane_stride,
                       int32_t* census, void* stream);

/* GroupNorm (+ ReLU) on channels_last activations x[N, HW, C] (replaces torch.nn.GroupNorm behind the tower convolutions of the
 * oriented head: mmcv ConvModule with norm_cfg=dict(type='GN', num_groups=32) - the default of
 * OBB_TOD/mmrotate/models/dense_heads/rotated_fcos_head_p2rb_ts.py:135, built by mmdet's AnchorFreeHead._init_cls_convs /
 * _init_reg_convs, HBB_TOD/mmdet/models/dense_heads/anchor_free_head.py:86-135):
 *   y = (x - mean[n, g]) * rstd[n, g] * gamma[c] + beta[c]   (ReLU if relu != 0),  statistics over the H*W * C/G elements of a
 * group (biased variance, float64 accumulation), mean / rstd [N, G] saved for the backward.
 * bwd: dy = grad_y * (y > 0) when y != NULL (the fused ReLU) else grad_y;  grad_x, grad_gamma[C], grad_beta[C] are written.
 * workspace: pt_group_norm_cl_workspace_bytes(N, HW, C, G) bytes, 16-byte aligned.  Fixed-order reductions: deterministic.
 * C % 4 == 0, (C / G) % 4 == 0, C / 4 divides 256, C / G divides 256, G <= 64. */
int64_t pt_group_norm_cl_workspace_bytes(int N, int HW, int C, int G);
int pt_group_norm_cl_fwd(const float* x, const float* gamma, const float* beta, int N, int HW, int C, int G, float eps,
                         int relu, float* y, float* mean, float* rstd, void* workspace, void* stream);
int pt_group_norm_cl_bwd(const float* grad_y, const float* x, const float* y, const float* gamma, const float* mean,
                         const float* rstd, int N, int HW, int C, int G, float* grad_x, float* grad_gamma,
                         float* grad_beta, void* workspace, void* stream);

/* The weight planes of many convolutions (1 x 1 and 3 x 3) in ONE launch (the weights change once per iteration: one launch instead
 * of a split - and, for the input-gradient form, a flip and a copy - per weight and form).  `items`: DEVICE array; every item = a
 * channels_last fp32 weight [Cout][KH][KW][Cin] (taps = KH * KW = 1 or 9; Cin % 32 == 0, Cout % 32 == 0), 16-byte aligned planes of
 * pt_split_bf16x3_plane_elems(rows, k) elements each (plane stride `plane`), and the form: mode 0 = rows Cout, k = (ky, kx, cin) - what
 * pt_conv_bf16x6's forward takes; mode 1 = rows cin, k = (KH - 1 - ky, KW - 1 - kx, cout) - its input gradient; mode 2 (ABI 6) =
 * rows (ky, kx, cin), k = cout - the transpose of the forward matrix: d col = g W of a convolution evaluated as a GEMM over gathered
 * columns (the deformable convolutions of `dcn_on_last_conv`, dense_heads/anchor_free_head.py:101-102).  scale (NULL or
 * [Cout], device): w[o] * scale[o] is what gets split - the scale of the frozen BatchNorm behind the convolution folded into the
 * input-gradient weights (backbones/resnet.py:262-303: dx = (g * scale) W = g (diag(scale) W)).  first_block = number of 16 x 32 blocks
 * of the items before this one (blocks of an item = ceil(rows / 16) * (taps * cols / 32)); total_blocks = their sum over all items.
 * np: planes written - 3 (fp32 = x0 + x1 + x2), or 1 (the bf16 rounding of the weight alone: the operand of the bf16 trunk).
 * ABI 4: the item grew `taps`, `np` and `scale`. */
typedef struct {
  const float* w;
  uint16_t* dst;
  int64_t plane;
  int32_t O, I;
  int32_t mode;
  int32_t first_block;
  int32_t taps;
  int32_t np;
  const float* scale;
} pt_conv_weight_item;
#define PT_F16_WEIGHT_SCALE 16.0f   /* item.np == 2: the planes are two fp16 planes of PT_F16_WEIGHT_SCALE * w (operand_f16 consumers) */
int pt_conv_weight_planes_batch(const pt_conv_weight_item* items, int n_items, int total_blocks, void* stream);

/* Frozen BatchNorm (+ residual add) (+ ReLU) in one pass each way.  Every BatchNorm on the path
 * is in eval mode with a frozen affine (models/backbones/resnet.py:647-658, config
 * norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True), i.e. y = x*scale[c] + shift[c]
 * with scale = gamma/sqrt(var+eps), shift = beta - mean*scale; the Bottleneck tail
 * (resnet.py:262-303) adds the identity and applies ReLU.  inner = H*W for NCHW tensors, 1 for
 * channels_last; y may alias x.  bwd: m = relu ? (y > 0) : 1; grad_res = g*m (may be NULL);
 * grad_x = g*m*scale[c] (may be NULL).  n % 4 == 0. */
int pt_affine_relu_fwd(const float* x, const float* scale, const float* shift,
                       const float* residual, int64_t n, int C, int64_t inner, int relu, float* y,
                       void* stream);
int pt_affine_relu_bwd(const float* grad_y, const float* y, const float* scale, int64_t n, int C,
                       int64_t inner, int relu, float* grad_x, float* grad_res, void* stream);
/* The frozen stem's tail in one pass (models/backbones/resnet.py:633-640: x = conv1(x); x = norm1(x); x = relu(x); x = maxpool(x),
 * with frozen_stages >= 0 no gradient reaches it): y[B, Ho, Wo, C] = max_pool2d(relu(x * scale[c] + shift[c]), kernel 3, stride 2,
 * pad 1) of a channels_last map x[B, H, W, C]; Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1; C % 4 == 0; forward only. */
int pt_affine_relu_maxpool_fwd(const float* x, const float* scale, const float* shift, int B, int H, int W,
                               int C, float* y, void* stream);

/* Top-down step of the necks on channels_last maps ([N,H,W,C], fp32 or bf16 = uint16 bit patterns when `bf16`):
 * out = a + nearest_upsample(b) with a, out [N,Ha,Wa,C] and b [N,Hb,Wb,C] - `laterals[i-1] += F.interpolate(laterals[i],
 * size=..., mode='nearest')` of HBB_TOD/mmdet/models/necks/fpn.py:165-173 and `inputs[index-1] + F.interpolate(...)` of
 * necks/ps_fpn.py:64-72; source index min(floor(dst * (float)in / out), in - 1) as torch's 'nearest'.
 * Backward: grad_a is grad_out itself; pt_upsample_add_bwd writes grad_b (sum over the pixels that read each source
 * pixel, fp32 accumulation, no atomics).  C % 4 == 0 (fp32) / C % 8 == 0 (bf16); N*Ha <= 65535. */
int pt_upsample_add_fwd(const void* a, const void* b, int N, int Ha, int Wa, int Hb, int Wb, int C, int bf16,
                        void* out, void* stream);
int pt_upsample_add_bwd(const void* grad_out, int N, int Ha, int Wa, int Hb, int Wb, int C, int bf16,
                        void* grad_b, void* stream);

/* The same epilogue for an eval-mode BatchNorm whose affine TRAINS (OBB config 5:
 * norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, OBB_TOD/configs/point teacher/
 * sodaa_fcos_pointteacher_1x.py:36-38): one pass produces grad_x / grad_res as pt_affine_relu_bwd and
 * accumulates sums[c] += g*m (dL/dshift) and sums[C+c] += g*m*x (dL/dscale), x = the convolution
 * output the forward read.  channels_last only ([N,H,W,C] flattened, n = N*H*W*C); C/4 must divide 256
 * or be a multiple of it.  No atomics: every workgroup writes a private row of
 * partial_ws[pt_affine_train_rows(n, C)][2*C] and a second launch adds the rows into sums[2*C]
 * (deterministic).  x == NULL: the epilogue of a convolution with a trainable bias followed by ReLU (the FCOS towers,
 * anchor_free_head.py:86-135 via mmcv ConvModule: conv -> +bias -> ReLU, three element-wise passes in the reference) with
 * scale == 1: only sums[0..C) (the bias gradient) is meaningful, sums[C..2C) are zero. */
int pt_affine_train_rows(int64_t n, int C);
int pt_affine_relu_bwd_train(const float* grad_y, const float* y, const float* x, const float* scale,
                             int64_t n, int C, int relu, float* grad_x, float* grad_res, float* sums,
                             float* partial_ws, void* stream);

/* bf16 activations (BASELINE configs[2]: bf16 autocast backbone): the frozen-BN epilogue on channels_last bf16
 * tensors (stored as uint16_t), arithmetic in fp32, fp32 scale/shift, C % 8 == 0. */
int pt_affine_relu_fwd_bf16(const uint16_t* x, const float* scale, const float* shift,
                            const uint16_t* residual, int64_t n, int C, int relu, uint16_t* y,
                            void* stream);
int pt_affine_relu_bwd_bf16(const uint16_t* grad_y, const uint16_t* y, const float* scale, int64_t n,
                            int C, int relu, uint16_t* grad_x, uint16_t* grad_res, void* stream);

/* ------------------------------------------------------------------------ NMS --
 * mmcv.ops.nms (offset 0), call site core/post_processing/bbox_nms.py:76 through
 * batched_nms: boxes[N,4] must be sorted by descending score; class-aware when
 * class_id != NULL.  keep[N] uint8.  N <= 32768. mask_ws: N*ceil(N/64) uint64 workspace. */
int pt_nms_sorted(const float* boxes, const int32_t* class_id, int N, float iou_thr,
                  uint64_t* mask_ws, uint8_t* keep, void* stream);
/* mmcv.ops.box_iou_rotated (OBB rotate_iou2d_calculator; syn_images_generator_v2.py:667 via
 * nms_rotated): boxes (cx,cy,w,h,angle[rad]). aligned: out[M]; else out[M,N]. */
int pt_box_iou_rotated(const float* a, const float* b, int M, int N, int aligned, float* out,
                       void* stream);
/* mmcv.ops.nms_rotated: dets[N,5] sorted by descending score; keep[N] uint8. N <= 32768. */
int pt_nms_rotated_sorted(const float* dets, int N, float iou_thr, uint64_t* mask_ws,
                          uint8_t* keep, void* stream);
/* The same for the images of a batch in ONE pair of launches (generate_black_paper runs one NMS per image,
 * syn_images_generator_v2.py:667; each is a latency-bound serial scan): candidates of image i are rows
 * [seg_off[i], seg_off[i+1]) of dets (each segment sorted by descending score), seg_off is a [host] array of n_seg + 1
 * entries, n_seg <= 16, <= 8192 candidates per segment; mask_ws: sum_i n_i * ceil(n_i / 64) uint64. */
int pt_nms_rotated_sorted_segments(const float* dets, const int32_t* seg_off, int n_seg, float iou_thr,
                                   uint64_t* mask_ws, uint8_t* keep, void* stream);
/* the 255-mask of generate_black_paper (syn_images_generator_v2.py:678-688): every pixel
 * inside or on the boundary of one of the quads[Q,8] (vertices truncated to int32 like
 * polygon.astype(np.int32)) with alive[q] != 0 is set to `value` in img[C,H,W]. */
int pt_fill_quads(float* img, int C, int H, int W, const float* quads, const uint8_t* alive,
                  int Q, float value, void* stream);
/* ... and for a whole batch: img [B, C, H, W]; quad q paints image img_of[q] (int32, device). */
int pt_fill_quads_batch(float* img, int B, int C, int H, int W, const float* quads, const uint8_t* alive,
                        const int32_t* img_of, int Q, float value, void* stream);

/* ------------------------------------------------- oriented boxes (OBB variant) --
 * mmcv.ops.diff_iou_rotated_2d, call sites OBB_TOD/mmrotate/models/losses/rotated_iou_loss.py:47,90
 * (RotatedIoULoss / DN_IoULoss): aligned IoU of boxes (cx,cy,w,h,angle[rad]) [N,5], differentiable
 * w.r.t. boxes1 (the prediction; the reference detaches the target).  bwd: grad_boxes1[N,5] =
 * grad_iou[n] * d iou / d boxes1[n]. */
int pt_diff_iou_rotated_fwd(const float* boxes1, const float* boxes2, int N, float* iou,
                            void* stream);
int pt_diff_iou_rotated_bwd(const float* boxes1, const float* boxes2, const float* grad_iou, int N,
                            float* grad_boxes1, void* stream);
/* mmcv.ops.RoIAlignRotated(out_size, spatial_scale, sample_num, aligned=True, clockwise), cfg
 * OBB_TOD/configs/point teacher/sodaa_fcos_pointteacher_1x.py:71-80, called
 * OBB_TOD/mmrotate/models/roi_heads/roi_extractors/rotate_single_level_roi_extractor.py:126.
 * rois[K,6] = (batch, cx, cy, w, h, theta); layouts as pt_roi_align_*; bwd accumulates. */
int pt_roi_align_rotated_fwd(const float* feat, const float* rois, int B, int C, int H, int W,
                             int K, int out_size, float spatial_scale, int sample_num, int aligned,
                             int clockwise, int channels_last, float* out, void* stream);
int pt_roi_align_rotated_bwd(const float* grad_out, const float* rois, int B, int C, int H, int W,
                             int K, int out_size, float spatial_scale, int sample_num, int aligned,
                             int clockwise, int channels_last, float* grad_feat, void* stream);

/* ---- teacher->student glue of the OBB head (TS_P2RBRotatedFCOSHead) ----------------
 * FUSETopkAssigner as OBB_TOD/mmrotate/models/dense_heads/rotated_fcos_head_p2rb_ts.py:883-885
 * calls it: `dec`[B*P,5] are the DistanceAnglePointCoder-decoded (cx,cy,w,h,a) boxes; InsiderCost
 * (HBB_TOD/mmdet/core/bbox/match_costs/match_cost.py:235-241) reads columns 0-3 as an
 * axis-aligned cxcywh box.  Everything else as pt_fuse_assign. */
int pt_fuse_assign_obb(const float* points, int P, const float* dec, const float* cls, int C,
                       const float* gt_xy, const int32_t* gt_labels, const int32_t* off, int B,
                       int sumG, int num_pre, int topk, float cls_w, float reg_w, float loc_w,
                       int32_t* gt_inds, int32_t* cand, void* stream);
/* _gnerate_pseudo_single, rotated_fcos_head_p2rb_ts.py:899-917: score-weighted mean of the
 * decoded 5-vectors (angle included) -> pseudo_bboxes[sumG,5]; (gx,gy,8,8,0) when nothing was
 * assigned.  pseudo_points[sumG,2], pseudo_scores[sumG], nassigned[sumG]. */
int pt_pseudo_boxes_obb(const float* dec, int P, const float* cls, int C, const float* gt_xy,
                        const int32_t* gt_labels, const int32_t* off, int B, int sumG, int num_pre,
                        const int32_t* gt_inds, const int32_t* cand, float* pseudo_bboxes,
                        float* pseudo_points, float* pseudo_scores, int32_t* nassigned,
                        void* stream);
/* _get_target_single / _get_target_pseudo_single, rotated_fcos_head_p2rb_ts.py:671-716, :781-843
 * (+ centerness_target :1118-1138): boxes[sumG,5]; out labels[B*P], bbox_targets[B*P,4] =
 * (l,t,r,b) in the frame of the assigned box (box 0 of the image when unassigned),
 * angle_targets[B*P], ctr_target[B*P] (0 when unassigned). */
int pt_fcos_targets_obb(const float* points, int P, const int32_t* gt_inds, const float* boxes,
                        const int32_t* box_labels, const int32_t* off, int B, int num_classes,
                        int32_t* labels, float* bbox_targets, float* angle_targets,
                        float* ctr_target, void* stream);
/* mil_bag_selection(_single), rotated_fcos_head_p2rb_ts.py:1198-1250: as pt_mil_bag_select on
 * (cx,cy,w,h,a) bags[NG*U1*U2,5] / pseudo[NG,5]; columns 0 and 1 are clamped to [0,w] and then
 * to [0,h] (:1211-1212), w/h/a are left alone. */
int pt_mil_bag_select_obb(const float* cls, const float* ins, const uint8_t* valid,
                          const int32_t* labels, const float* bags, const float* pseudo, int NG,
                          int U1, int U2, int C, int topk, float beta, float img_h, float img_w,
                          float* merged, void* stream);

/* ------------------------------------------------ (modulated) deformable convolution --
 * mmcv.ops.DeformConv2d / ModulatedDeformConv2d(Pack) (`conv_cfg=dict(type='DCN'|'DCNv2')`), reachable on the path
 * through `dcn_on_last_conv=True` (HBB_TOD/mmdet/models/dense_heads/anchor_free_head.py:101-102,121-122,
 * fcos_head_p2b_ts.py:197-198; no shipped Point-Teacher config turns it on).  The contraction is a plain GEMM on
 * col (caller: hipBLASLt); these entries are the data-dependent halves.  x[B,C,H,W] (NCHW),
 * offset[B, 2*dg*kh*kw, Ho, Wo] = (dy, dx) per tap per deformable group, mask[B, dg*kh*kw, Ho, Wo] or NULL
 * (DCNv1), col / grad_col[B, C*kh*kw, Ho*Wo].  Samples outside (-1,H)x(-1,W) are 0, neighbours outside the
 * map contribute 0 (mmcv modulated_deform_conv_cuda_kernel.cuh).  col2im ACCUMULATES into grad_x;
 * col2im_coord writes grad_offset (and grad_mask when mask != NULL). */
int pt_deform_im2col(const float* x, const float* offset, const float* mask, int B, int C, int H, int W,
                     int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                     int dil_w, int deform_groups, float* col, void* stream);
int pt_deform_col2im(const float* grad_col, const float* offset, const float* mask, int B, int C, int H,
                     int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                     int dil_w, int deform_groups, float* grad_x, void* stream);
int pt_deform_col2im_coord(const float* grad_col, const float* x, const float* offset, const float* mask,
                           int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                           int stride_h, int stride_w, int dil_h, int dil_w, int deform_groups,
                           float* grad_offset, float* grad_mask, void* stream);

/* The same operator on the layout the training path keeps its maps in (channels_last): x[B,H,W,C],
 * offset[B,Ho,Wo,2*dg*kh*kw], mask[B,Ho,Wo,dg*kh*kw] or NULL, col / grad_col[B*Ho*Wo, kh*kw, C] - so that the GEMM
 * col[B*L, K*C] x weight[O, kh, kw, C]^T reads and writes NHWC without a transpose.  One wavefront per (pixel, tap,
 * deformable group), 4 channels per lane (C / dg must be a multiple of 4): contiguous 1-KiB gathers / stores / atomics.
 * pt_deform_col2im_cl does the whole backward of the gather in one pass: it ACCUMULATES into grad_x (NULL: skipped) and
 * writes grad_offset and (mask != NULL) grad_mask. */
int pt_deform_im2col_cl(const float* x, const float* offset, const float* mask, int B, int C, int H, int W,
                        int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                        int dil_w, int deform_groups, float* col, void* stream);
int pt_deform_col2im_cl(const float* grad_col, const float* x, const float* offset, const float* mask, int B,
                        int C, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h,
                        int stride_w, int dil_h, int dil_w, int deform_groups, float* grad_x,
                        float* grad_offset, float* grad_mask, void* stream);

/* -------------------------------------------------------- evaluator (next row N1) --
 * evaluateImg of the COCO / AI-TOD protocol (aitodpycocotools.cocoeval.COCOeval, called at
 * HBB_TOD/mmdet/datasets/aitod.py:109-146): greedy matching of the detections of every (image,
 * category) segment, in descending score order, against that segment's ground truths, for A area
 * ranges x T IoU thresholds at once (A*T <= 64).  det_box[Nd,4] xyxy sorted by (segment, -score);
 * det_off[S+1], gt_off[S+1] segment offsets; gt_flags bit 0 = ignore, bit 1 = iscrowd; only the first
 * max_det detections of a segment are matched.  gt_matched[Ng*64] is zeroed scratch.  Outputs, per
 * detection and lane (a*T + t): dtm = matched gt row or -1, dt_ig = ignored flag (matched to an ignored
 * gt, or unmatched with an area outside the range). */
int pt_coco_match(const float* det_box, const int32_t* det_off, const float* gt_box,
                  const float* gt_area, const uint8_t* gt_flags, const int32_t* gt_off, int S,
                  const float* area_lo, const float* area_hi, int A, const float* iou_thr, int T,
                  int max_det, uint8_t* gt_matched, int32_t* dtm, uint8_t* dt_ig, void* stream);

/* SODA-A protocol (OBB_TOD/mmrotate/datasets/sodaa_eval/sodaa_eval.py:158-179 computeIoU over mmcv's
 * box_iou_rotated, :181-262 evaluateImg): the same greedy matching with the IoU between ORIENTED boxes.
 * pt_segment_iou_rotated fills, per (image, category) segment, the row-major [min(D_s, max_det), G_s] IoU matrix
 * at iou[iou_off[s]] (boxes (cx, cy, w, h, a); max_pairs = the largest segment's pair count, sizes the grid;
 * at most 65535 segments per call).  pt_coco_match_iou is pt_coco_match reading those matrices; det_area[Nd] is
 * w*h of each detection (the area test of unmatched detections).  gt_zero / det_zero (row index or -1): SODAAeval
 * numbers ground truths and detections from 0 (:108, :121) and reads 0 as "no match" (:392, :415, :509) - a
 * detection matched to row gt_zero is reported unmatched and detection det_zero does not block its ground truth. */
int pt_segment_iou_rotated(const float* det_box, const int32_t* det_off, const float* gt_box,
                           const int32_t* gt_off, int S, int max_det, const int64_t* iou_off,
                           int64_t max_pairs, float* iou, void* stream);
int pt_coco_match_iou(const float* det_area, const int32_t* det_off, const float* iou,
                      const int64_t* iou_off, const float* gt_area, const uint8_t* gt_flags,
                      const int32_t* gt_off, int S, const float* area_lo, const float* area_hi, int A,
                      const float* iou_thr, int T, int max_det, int gt_zero, int det_zero,
                      uint8_t* gt_matched, int32_t* dtm, uint8_t* dt_ig, void* stream);

/* ----------------------------------------------------- data pipeline (next row N2) --
 * Fused Resize -> RandomFlip -> Normalize -> Pad -> DefaultFormatBundle (+ collate zero padding) of the
 * reference's pipelines (HBB_TOD/mmdet/datasets/pipelines/transforms.py:212-237, :437-440, :652-655,
 * :587-599; formating.py:196-203; configs `train_pipeline`, e.g. aitodv2_point_teacher_0%.py:180-189),
 * whose pixel work is mmcv 1.x over OpenCV (cv2.resize INTER_LINEAR 8-bit fixed point, cv2.flip,
 * cv2.subtract / cv2.multiply, cv2.copyMakeBorder).
 * src: DEVICE uint8 [src_h, src_w, 3] (BGR as decoded), rows src_row_stride bytes apart.
 * (rs_h, rs_w): size after Resize; flip: 0 none, 1 horizontal, 2 vertical, 3 diagonal (applied after the resize);
 * mean_host / stdinv_host: HOST arrays of 3 (indexed by OUTPUT channel) or both NULL = no Normalize;
 * to_rgb swaps channels 0 and 2 before the mean; [rs, pad) is filled with pad_val, [pad, out) with 0.
 * dst: DEVICE float, element strides (c, h, w) - NCHW planes or the channels-last slice of a batch. */
int pt_image_prep(const uint8_t* src, int src_h, int src_w, int64_t src_row_stride, int channels,
                  int rs_h, int rs_w, int flip, const float* mean_host, const double* stdinv_host,
                  int to_rgb, int pad_h, int pad_w, float pad_val, int out_h, int out_w, float* dst,
                  int64_t dst_stride_c, int64_t dst_stride_h, int64_t dst_stride_w, void* stream);

/* ------------------------------------------- supervised FCOS baseline (next row N4) --
 * FCOSHead.get_targets / _get_target_single (HBB_TOD/mmdet/models/dense_heads/fcos_head.py:806-1007) of
 * configs/baselines/aitodv2_fcos_r50_1x.py: per image and point, the smallest-area ground truth whose box
 * (centre-sampled with sample_radius[p] = stride * center_sample_radius when center_sampling != 0) contains the
 * point and whose largest side distance lies in regress_ranges[p] = (lo, hi); none -> label num_classes.
 * points[P,2], regress_ranges[P,2], sample_radius[P], target_norm[P] (stride when norm_on_bbox, else 1);
 * boxes[sumG,4] xyxy, box_labels[sumG], off[B+1].  Outputs image-major: labels[B*P], bbox_targets[B*P,4]
 * = (l,t,r,b) / target_norm (of box 0 for background points, as the reference leaves them; zeros when the
 * image has no box), ctr_target[B*P] = centerness of the positive points, 0 elsewhere (either may be NULL). */
int pt_fcos_dense_targets(const float* points, const float* regress_ranges, const float* sample_radius,
                          const float* target_norm, int P, const float* boxes, const int32_t* box_labels,
                          const int32_t* off, int B, int num_classes, int center_sampling,
                          int32_t* labels, float* bbox_targets, float* ctr_target, void* stream);

/* MaxIoUAssigner.assign / assign_wrt_overlaps (HBB_TOD/mmdet/core/bbox/assigners/max_iou_assigner.py:98-212) of the
 * anchor-based baselines (configs/baselines/aitodv2_retinanet_r50_1x.py), for a whole batch without the [G, A] overlap
 * matrix.  anchors[A,4] xyxy shared by the images; gt_boxes[sumG,4], off[B+1].  Per image: max_overlaps[B*A] and its
 * argmax, assigned_gt_inds[B*A] = -1 ignore / 0 background / i+1 (box i of the image): background where
 * neg_iou_lo <= max < neg_iou_hi (a float neg_iou_thr t is (0, t)), positive where max >= pos_iou_thr; with
 * match_low_quality every box whose best IoU >= min_pos_iou claims all anchors that tie that best (gt_max_assign_all)
 * or its first best anchor, later boxes overriding earlier ones.  An image without boxes is all background.
 * argmax_ws[B*A] int32 and gt_best_ws[sumG] uint64 (ZEROED by the caller) are scratch. */
int pt_max_iou_assign(const float* anchors, int A, const float* gt_boxes, const int32_t* off, int B,
                      float pos_iou_thr, float neg_iou_lo, float neg_iou_hi, float min_pos_iou,
                      int match_low_quality, int gt_max_assign_all, float* max_overlaps,
                      int32_t* argmax_ws, uint64_t* gt_best_ws, int32_t* assigned_gt_inds, void* stream);

/* ------------------------------------------------------------------ glue (csrc/glue.hip) --
 * Fused replacements of launch-bound torch chains of the detector logic; no arithmetic beyond the cited lines.
 * pt_box_convert: core/bbox/transforms.py:250-262 (mode 0, xyxy -> cxcywh) / :236-247 (mode 1); in / out [n,4]. */
int pt_box_convert(const float* in, float* out, int n, int mode, void* stream);

/* Geometry half of strong_augmentation (detectors/syn_images_generator_v2.py:41-92, :114-120) for all rows of a batch:
 * in / out [N, ncoord] (ncoord 2 = points, 4 = boxes: corners re-ordered after the flips), images delimited by off[B+1];
 * params [B,6] = (flip_x, flip_y, scale, margin_w, margin_h, scale >= 1) per image.  valid (uint8 [N], points only, may be
 * NULL): the point stays inside the centre crop (:78-79, :84-85). */
int pt_aug_geometry(const float* in, float* out, uint8_t* valid, const int32_t* off, int B, int N, int ncoord,
                    const float* params, float H, float W, void* stream);

/* Burn-in step 1, the candidate table of generate_black_paper (syn_images_generator_v2.py:597-663) for the whole batch.
 * gt [sumG, gt_cols] xyxy boxes of the real objects (images delimited by goff[B+1]); prior [L,4] = shape_list;
 * draws [10, sumG] = the reference's per-object draws (scale, x, y, wn, rn, a, boost, itv, itv2, dev); cls [sumG] the
 * prior index of every object (:473).  Per image the table holds [G real objects | G rectangles | 2 x 5 adjacency copies]
 * = 2 G + 10 rows (x, y, w, h, a, score), image b starting at row 2 goff[b] + 10 b.  exist: row is present; key: int64
 * whose ascending stable sort orders the batch image-major by descending score (rows that do not exist last). */
int pt_black_paper_rects(const float* gt, int gt_cols, const int32_t* goff, int B, const float* prior, int L, int dense_n,
                         const float* draws, const int32_t* cls, int sumG, float imgsize, float* table, int64_t* key,
                         uint8_t* exist, void* stream);

/* Rows in sorted order (order [T] from the sort of `key`): sorted [T,6]; nms_in [T,5] (absent rows -> a far-away
 * speck; input of pt_nms_rotated_sorted per image, :667); polys [T,8] (data_augument_bank.py:516-541); hull [T,4]
 * (fcos_p2b_teacher_student.py:486-492); pre [T] = exists & score < 1 & inside the image (:669-675). */
int pt_black_paper_sorted(const float* table, const int64_t* order, const uint8_t* exist, int T, float imgsize,
                          float* sorted, float* nms_in, float* polys, float* hull, uint8_t* pre, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PT_HIP_H */
