/*
 * mrcnn_hip.h -- C-ABI of the MI355X (gfx950) Mask R-CNN hot path.
 *
 * The reference (SKA-INAF/caesar-mrcnn) has no FFI: its hot path is the Keras/TF1 graph wired by
 * MaskRCNN.build (mrcnn/model.py:1935-2166).  Each entry point below replaces one group of TF/Keras ops
 * of that graph; the reference call site it stands in for is cited next to it.  All pointers are DEVICE
 * pointers (allocated by the caller, e.g. PyTorch-ROCm), all tensors are float32 NHWC unless stated,
 * `stream` is a hipStream_t passed as void*, and every function returns 0 on success or a negative
 * status (mirroring the reference's 0 / -1 convention, scripts/run.py:1747-1757).  No function
 * allocates, frees or synchronises; all are re-entrant per stream.
 */
#ifndef MRCNN_HIP_H
#define MRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRCNN_OK 0
#define MRCNN_ERR_ARG (-1)
#define MRCNN_ERR_LAUNCH (-2)
#define MRCNN_ERR_WORKSPACE (-3)
#define MRCNN_ERR_UNSUPPORTED (-4)   /* valid request this entry point has no kernel for: use the unfused calls */

#define MRCNN_ACT_NONE 0
#define MRCNN_ACT_RELU 1
#define MRCNN_ACT_SIGMOID 2

#define MRCNN_RES_NONE 0
#define MRCNN_RES_SAME 1     /* res has the shape/strides of out                            */
#define MRCNN_RES_UP2 2      /* res is [N, OH/2, OW/2, Cout]; nearest 2x upsample then add   */

#define MRCNN_OUT_NHWC 0     /* out[n*out_n_stride + oh*out_h_stride + ow*out_w_stride + co] */
#define MRCNN_OUT_DECONV2 1  /* GEMM column = (a*2+b)*cmod + co -> pixel (2oh+a, 2ow+b), channel co */

/* Convolution as implicit GEMM: M = N*OH*OW pixels, N = Cout, K = KH*KW*Cin.
 * Replaces KL.Conv2D / TimeDistributed(Conv2D) / Dense / Conv2DTranspose(2x2,s2) with the following
 * BatchNorm(training=False) + Add + Activation folded into the epilogue
 * (mrcnn/model.py:99-210, 916-957, 986-1091, 2005-2022).
 * Weights are HWIO ([KH,KW,Cin,Cout] row-major == the GEMM B matrix [K, Cout]), i.e. the Keras layout.
 * Epilogue:  z = acc + bias[c];  (z_out gets z);  y = scale ? scale[c]*z + shift[c] : z;
 *            y += res (res_mode);  y = act(y);  out gets y.        c = column % cmod.          */
typedef struct mrcnn_conv_desc {
    int32_t N, H, W, Cin;
    int32_t Cout;            /* GEMM N (for DECONV2: 4*cmod) */
    int32_t KH, KW, stride, pad_t, pad_l;
    int32_t OH, OW;
    int32_t act, res_mode, out_mode;
    int32_t cmod;            /* channel modulus for bias/scale/shift (== Cout except DECONV2) */
    int64_t out_n_stride, out_h_stride, out_w_stride;   /* element strides of out / z_out / res(SAME) */
} mrcnn_conv_desc;

int mrcnn_conv2d_fwd(const mrcnn_conv_desc* d, const float* x, const float* w, const float* bias,
                     const float* scale, const float* shift, const float* res, float* out,
                     float* z_out, void* stream);

/* Same, with scratch for split-K: layers with few output tiles (small feature maps) cut the K loop
 * into slices run by separate workgroups; partial slabs are summed in a fixed order by a second
 * kernel that also applies the epilogue.  mrcnn_conv2d_fwd_workspace() gives the bytes needed
 * (0 = no split); a NULL/short workspace silently runs unsplit.                                     */
size_t mrcnn_conv2d_fwd_workspace(const mrcnn_conv_desc* d);
int mrcnn_conv2d_fwd_ws(const mrcnn_conv_desc* d, const float* x, const float* w, const float* bias,
                        const float* scale, const float* shift, const float* res, float* out,
                        float* z_out, void* workspace, size_t workspace_bytes, void* stream);

/* dW[K, Cout] = sum over pixels of im2col(x)^T . dy   (gradient of KL.Conv2D kernels, taken by TF
 * autodiff in the reference: keras fit_generator, mrcnn/model.py:2487).  dy is dense [N,OH,OW,Cout].
 * Partial sums over `splits` pixel ranges go to `workspace` (splits*K*Cout floats) and are reduced in
 * a fixed order (bitwise reproducible); beta_acc != 0 accumulates into dw instead of overwriting.   */
int mrcnn_conv2d_wgrad(const mrcnn_conv_desc* d, const float* x, const float* dy, float* dw,
                       float* workspace, size_t workspace_bytes, int beta_acc, void* stream);
size_t mrcnn_conv2d_wgrad_workspace(const mrcnn_conv_desc* d);

/* w_t[(KH-1-kh, KW-1-kw, co), ci] = w[(kh,kw,ci), co]: the weights of the data-gradient convolution. */
int mrcnn_weight_flip_transpose(const float* w, float* w_t, int KH, int KW, int Cin, int Cout,
                                void* stream);
/* The same for every layer of a flat parameter buffer in one launch.  table (device): n_layers records of
 * { int64 offset (floats; same in params and params_t), int32 KH, KW, Cin, Cout, first_tile, pad } sorted by
 * first_tile, a tile being 32 x 32 (ci, co) of one tap; total_tiles = sum over layers of
 * KH*KW*ceil(Cin/32)*ceil(Cout/32).                                                                     */
int mrcnn_weight_flip_transpose_batched(const float* params, float* params_t, const void* table, int n_layers,
                                        int total_tiles, void* stream);

/* n <= 5 independent convolutions in ONE launch (plus one for their split-K reductions).  Replaces the Python loops
 * of the reference that apply one layer to several small feature maps in turn: the shared RPN model over the pyramid
 * levels (mrcnn/model.py:2040-2055, rpn_graph :916-957) and the FPN smoothing convolutions (:2018-2026).  Every problem
 * is an ordinary mrcnn_conv2d_fwd (own descriptor, pointers and output strides); all must fall into the same output
 * tile class (Cout <= 32, <= 64 or larger), else MRCNN_ERR_UNSUPPORTED and nothing is launched.  Results are those of
 * n mrcnn_conv2d_fwd_ws calls (same kernels; the K-slice count may differ, i.e. fp32 summation order only).      */
typedef struct mrcnn_conv_problem {
    mrcnn_conv_desc d;
    const float* x; const float* w; const float* bias; const float* scale; const float* shift; const float* res;
    float* out; float* z_out;
} mrcnn_conv_problem;
size_t mrcnn_conv2d_fwd_multi_workspace(const mrcnn_conv_problem* problems, int n);
int mrcnn_conv2d_fwd_multi(const mrcnn_conv_problem* problems, int n, void* workspace, size_t workspace_bytes,
                           void* stream);

/* The weight gradients of up to 4 layers in one launch (+ one for their slab reductions): TF autodiff of the three
 * Conv2D layers of a bottleneck block (mrcnn/model.py:99-172) on the small feature maps, where each is a few dozen
 * output tiles with a few hundred pixels to contract.  Every problem is an ordinary mrcnn_conv2d_wgrad (accumulate !=
 * 0 adds to dw); all must fit the LDS-DMA kernel (Cin, Cout multiples of 128, fewer than 65 536 output pixels,
 * 16-byte aligned buffers), else MRCNN_ERR_UNSUPPORTED (workspace query: 0) and nothing is launched.             */
typedef struct mrcnn_wgrad_problem {
    mrcnn_conv_desc d;
    const float* x; const float* dy; float* dw;
    int32_t accumulate;
} mrcnn_wgrad_problem;
size_t mrcnn_conv2d_wgrad_multi_workspace(const mrcnn_wgrad_problem* problems, int n);
int mrcnn_conv2d_wgrad_multi(const mrcnn_wgrad_problem* problems, int n, float* workspace, size_t workspace_bytes,
                             void* stream);

/* Data-gradient convolution fused with the epilogue backward of the layer below (TF autodiff through Conv2D, then
 * through the lower layer's Activation / BatchNorm / bias): y = conv(dz, w_t) (+ res) is d(loss)/d(out_below); stored is
 *   dz_below = y * act'(out_below) * scale_below,   and   dbeta += sum y*act',  dgamma += sum y*act'*(z-mean)*rstd,
 *   dbias += sum dz_below   (atomics, like mrcnn_epilogue_bwd).
 * Same result as mrcnn_conv2d_fwd followed by mrcnn_epilogue_bwd, without writing / re-reading y.  Supported: the
 * large layers (LDS-DMA kernel: >= 640 tiles of 128x128, Cin % 32 == 0, Cout % 128 == 0) and the split-K layers (small
 * feature maps; Cout a power of two, workspace as for mrcnn_conv2d_fwd_ws: the slab reduction then carries the
 * backward epilogue), dense output; otherwise MRCNN_ERR_UNSUPPORTED and nothing is launched.  res_mode SAME adds
 * `res` to y first.                                                                                            */
typedef struct mrcnn_bwd_epilogue {
    const float* out; const float* z; const float* scale; const float* mean; const float* rstd;
    float* dgamma; float* dbeta; float* dbias;
    int32_t act;                 /* MRCNN_ACT_NONE or MRCNN_ACT_RELU of the layer below */
    float* dy;                   /* optional second output y*act' (what a residual branch of the layer below receives: the
                                    block-output gradient of a bottleneck block); split-K layers only, else
                                    MRCNN_ERR_UNSUPPORTED */
} mrcnn_bwd_epilogue;
int mrcnn_conv2d_dgrad_ep(const mrcnn_conv_desc* d, const float* dz, const float* w_t, const float* res,
                          float* dz_below, const mrcnn_bwd_epilogue* ep, void* workspace, size_t workspace_bytes,
                          void* stream);

/* Frozen BatchNorm (KL.BatchNormalization with training=False, mrcnn/model.py:57-72; eps = Keras
 * default 1e-3):  scale = gamma*rsqrt(var+eps), shift = beta - mean*scale, for n channels.          */
int mrcnn_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                  float eps, float* scale, float* shift, float* rstd, int64_t n, void* stream);

/* Backward of the conv epilogue over a dense [M, C] tensor:
 *   dy = dout * act'(out)  (relu: out > 0; sigmoid: out*(1-out));  dy_out (optional) gets dy;  dz_out gets dy*scale (or dy);
 *   dbeta[c] += sum dy;  dgamma[c] += sum dy*(z-mean)*rstd;  dbias[c] += sum dz.
 * (reference: TF autodiff through BatchNorm/Add/Activation).  dbeta/dgamma/dbias accumulate (atomics);
 * any of scale/z/mean/rstd/dgamma/dbeta may be NULL for layers without BatchNorm.                     */
int mrcnn_epilogue_bwd(const float* dout, const float* out, const float* z, const float* scale,
                       const float* mean, const float* rstd, float* dy_out, float* dz_out,
                       float* dgamma, float* dbeta, float* dbias, int64_t M, int C, int act,
                       void* stream);

/* KL.MaxPooling2D((3,3), strides=(2,2), padding="same") (mrcnn/model.py:187); TF SAME padding:
 * pad_before = total/2 (0 at even input sizes), window clipped at the borders.  argmax index kept
 * for the backward pass (first maximum in window scan order, as TF's MaxPoolGrad... see DESIGN.md). */
int mrcnn_maxpool3x3s2_fwd(const float* x, float* out, int32_t* argmax, int N, int H, int W, int C,
                           int OH, int OW, int pad_t, int pad_l, void* stream);
int mrcnn_maxpool3x3s2_bwd(const float* dout, const int32_t* argmax, float* dx, int N, int H, int W,
                           int C, int OH, int OW, void* stream);

/* out[n, oh, ow, c] = x[n, 2oh, 2ow, c]  (KL.MaxPooling2D(pool_size=(1,1), strides=2) "fpn_p6",
 * mrcnn/model.py:2022) and its adjoint (accumulating scatter).                                       */
int mrcnn_subsample2_fwd(const float* x, float* out, int N, int H, int W, int C, void* stream);
int mrcnn_subsample2_bwd_acc(const float* dout, float* dx, int N, int H, int W, int C, void* stream);

/* dsrc[n, h, w, c] (+)= sum of the 2x2 block of dout: adjoint of KL.UpSampling2D(2,2)
 * (mrcnn/model.py:2007-2013).                                                                       */
int mrcnn_upsample2_bwd(const float* dout, float* dsrc, int N, int H, int W, int C, int accumulate,
                        void* stream);

/* dst[n,h,w,(a*2+b)*C+c] = src[n,2h+a,2w+b,c] (gradient regrouping for Conv2DTranspose 2x2/s2). */
int mrcnn_pixel_unshuffle2(const float* src, float* dst, int N, int H, int W, int C, void* stream);
/* Fused backward of the mask-head output stage (mrcnn/model.py:1087-1090): given dL/dmask [M,H,W,C] (after
 * the sigmoid), mask [M,H,W,C], up = relu(deconv) [M,H,W,Cd] and the 1x1 kernel w_mask [Cd,C]:
 * dzg [M,H/2,W/2,4*Cd] = ReLU-masked gradient of the deconv GEMM output (columns (a*2+b)*Cd+co);
 * dw_mask [Cd,C], db_mask [C], db_deconv [Cd] accumulate (float atomics).  C <= 16, Cd % 64 == 0.     */
int mrcnn_mask_out_bwd(const float* d_mask_out, const float* mask_out, const float* up, const float* w_mask,
                       float* dzg, float* dw_mask, float* db_mask, float* db_deconv, int64_t M, int H, int W,
                       int Cd, int C, void* stream);
/* DMA helpers (hipMemcpy2DAsync / hipMemsetAsync on `stream`), device to device. */
int mrcnn_copy2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t row_bytes,
                 size_t rows, void* stream);
int mrcnn_fill_zero(void* dst, size_t bytes, void* stream);
/* GT instance masks of a batch (mrcnn/model.py:1721-1904: [B, H, W, MAX_GT_INSTANCES] bool from data_generator) cross PCIe
 * bit-packed: byte b of a pixel = instances 8b .. 8b+7, least significant bit first.  out [npix, G] uint8 (0 / 1); instances
 * >= n_used are written as zeros.                                                                                           */
int mrcnn_unpack_mask_bits(const void* packed, void* out, int64_t npix, int nbytes_per_pixel, int n_used, int G, void* stream);

/* detect() post-processing (MaskRCNN.unmold_detections, mrcnn/model.py:2607-2619; utils.unmold_mask, mrcnn/utils.py:629-645;
 * utils.resize -> skimage.transform.resize(order=1, mode='constant', cval=0, clip=True), mrcnn/utils.py:957-978): for each of
 * the n detections of ONE image, resize the class channel of its MH x MW mask to its integer pixel box (float64, half-pixel
 * centres, zero outside the mask, clipped to the mask's [min, max]), threshold `>= 0.5` and paste it into the image plane.
 *   mrcnn_mask [n_rows, MH, MW, C] float32 (the graph's output for this image);
 *   dets [n, 6] int32 = (y1, x1, y2, x2, class_id, row): box in pixels of the ORIGINAL image (y2 / x2 exclusive, as
 *        utils.denorm_boxes returns them; the box arithmetic and the zero-area filter stay with the caller), the class
 *        channel and the row of mrcnn_mask the detection came from;
 *   out  packed == 0: [H, W, n] uint8 0 / 1 (np.stack(full_masks, axis=-1) of the reference, viewable as bool);
 *        packed != 0: [H, W, ceil(n / 8)] uint8, bit (d & 7) of byte d >> 3 = detection d (little bit order).
 * Pixels of a box outside the image are dropped.  workspace: mrcnn_unmold_masks_workspace(n, MH, MW) bytes.          */
size_t mrcnn_unmold_masks_workspace(int n, int MH, int MW);
int mrcnn_unmold_masks(const float* mrcnn_mask, int n_rows, int MH, int MW, int C, const int32_t* dets, int n, int H, int W,
                       int packed, void* out, void* workspace, size_t workspace_bytes, void* stream);

/* FITS tile -> network input image (utils.read_fits, mrcnn/utils.py:1033-1163, on the run.py path: stretch, normalize,
 * convertToRGB, to_uint8; per-channel astropy ZScaleInterval(contrast) [3P] -> / max -> round(255 x), :1101-1111, :1166-1208):
 *   raw   [H, W] float32 as stored in the file (big_endian != 0: FITS byte order, swapped on the device) -- the caller has
 *         parsed the header and cut the tile; NaN pixels are replaced by the minimum of the others (:1090-1091);
 *   zscale_contrasts  3 doubles (HOST pointer, read at call time);
 *   rgb   [H, W, 3] uint8, byte-identical to the host statement caesar-mrcnn_amd/fits.py:read_fits.
 * Tiles up to 4096 x 1024 pixels (else MRCNN_ERR_UNSUPPORTED: use the host path).  workspace: mrcnn_fits_workspace(H, W).  */
size_t mrcnn_fits_workspace(int H, int W);
int mrcnn_fits_to_rgb(const void* raw, int big_endian, int H, int W, const double* zscale_contrasts, void* rgb, void* workspace,
                      size_t workspace_bytes, void* stream);

/* detect() pre-processing (MaskRCNN.mold_inputs, mrcnn/model.py:2519-2556, for uint8 images): utils.resize_image's pixel work
 * (mrcnn/utils.py:456-561: bilinear up-scaling h x w -> oh x ow through skimage.transform.resize(order=1, mode='constant',
 * clip=True, preserve_range=True) [3P], zero padding into the OH x OW canvas at (top, left), cast to uint8) and mold_image
 * (mrcnn/model.py:2964-2969: float32 minus MEAN_PIXEL).  src [h, w, C] uint8 (C <= 4), out [OH, OW, C] float32 -- identical to
 * the host path (caesar-mrcnn_amd/utils.py:resize_image + mold_image); the caller computes scale, (oh, ow) = (round(h * scale),
 * round(w * scale)) and the padding as the reference does.  mean_pixel: C doubles on the HOST.  workspace: >= 8 bytes.   */
int mrcnn_mold_image_u8(const void* src, int h, int w, int C, int oh, int ow, int top, int left, int OH, int OW,
                        const double* mean_pixel, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* Elementwise helpers on flat buffers. */
int mrcnn_add_inplace(float* dst, const float* src, int64_t n, void* stream);
int mrcnn_softmax_rows(const float* logits, float* probs, int64_t rows, int C, void* stream);

/* PyramidROIAlign (mrcnn/model.py:428-534): level = clamp(4 + round(log2(sqrt(h*w)/(224/sqrt(area)))),
 * 2, 5); tf.image.crop_and_resize(bilinear, extrapolation 0), one sample per bin, output in the
 * original ROI order.  boxes [B, R, 4] normalised (y1,x1,y2,x2); fm[l] = P(2+l) as [B, Hl, Wl, C].
 * out [B, R, P, P, C].  C must be a multiple of 64*4 = 256 lanes-of-float4 ... any multiple of 4.   */
typedef struct mrcnn_roialign_desc {
    int32_t B, R, P, C;
    int32_t H[4], W[4];
    float image_area;        /* IMAGE_SHAPE[0]*IMAGE_SHAPE[1] as float32 */
} mrcnn_roialign_desc;
int mrcnn_roialign_fwd(const mrcnn_roialign_desc* d, const float* boxes, const float* fm2,
                       const float* fm3, const float* fm4, const float* fm5, float* out,
                       int32_t* level_out, void* stream);
/* Adjoint: scatter-add (float atomics) of dout into the four pre-zeroed/accumulating maps. */
int mrcnn_roialign_bwd(const mrcnn_roialign_desc* d, const float* boxes, const float* dout,
                       float* dfm2, float* dfm3, float* dfm4, float* dfm5, void* stream);

/* Same adjoint in gather form for the case where every ROI carries gradient (the class head in training: TF autodiff
 * of tf.image.crop_and_resize at mrcnn/model.py:505).  Bilinear sampling is separable, so each destination pixel reads
 * off its contributions directly: one wave per pyramid pixel walks the ROIs of its image and level, finds the sample
 * rows / columns whose floor or ceil is that pixel, and accumulates  wy*wx*dout[roi,py,px,:]  in registers -- one row of
 * atomics per pixel instead of four per bin, no sorting, fixed summation order.  C must be 256 and P <= 32 (else
 * MRCNN_ERR_UNSUPPORTED, nothing launched); same values as mrcnn_roialign_bwd up to fp32 summation order.          */
int mrcnn_roialign_bwd_gather(const mrcnn_roialign_desc* d, const float* boxes, const float* dout, float* dfm2,
                              float* dfm3, float* dfm4, float* dfm5, void* stream);
/* The same with 16-bit pyramid levels and a 16-bit pooled output (BASELINE configs[4]; dtype MRCNN_DTYPE_F16 / _BF16):
 * interpolation in float32, one rounding.  The adjoint reads the 16-bit gradient (times `multiplier`, e.g. 1 / loss scale)
 * and adds float32 atomics into the float32 pyramid gradients.                                                        */
int mrcnn_roialign_fwd_h16(const mrcnn_roialign_desc* d, int dtype, const float* boxes, const void* fm2, const void* fm3,
                           const void* fm4, const void* fm5, void* out, void* stream);
int mrcnn_roialign_bwd_h16(const mrcnn_roialign_desc* d, int dtype, const float* boxes, const void* dout, float multiplier,
                           float* dfm2, float* dfm3, float* dfm4, float* dfm5, void* stream);

/* ProposalLayer (mrcnn/model.py:329-406): per image, scores = rpn_probs[:, 1]; top-k(min(pre_nms, A),
 * sorted, ties -> lower index); decode with deltas*std; clip to [0,1]; greedy NMS (IoU > thr
 * suppresses, TF non_max_suppression semantics); gather; zero-pad to proposal_count.
 * Outputs: rois [B, proposal_count, 4]; optional debug outputs top_idx [B, K] (int32 anchor ids in
 * sorted order), keep_idx [B, proposal_count] (positions into the sorted list, -1 padded),
 * num_keep [B].  workspace from mrcnn_proposal_workspace().                                        */
typedef struct mrcnn_proposal_desc {
    int32_t B, A, pre_nms_limit, proposal_count;
    float nms_threshold;
    float std_dev[4];
} mrcnn_proposal_desc;
size_t mrcnn_proposal_workspace(const mrcnn_proposal_desc* d);
int mrcnn_proposal_fwd(const mrcnn_proposal_desc* d, const float* rpn_probs, const float* rpn_bbox,
                       const float* anchors, float* rois, int32_t* top_idx, int32_t* keep_idx,
                       int32_t* num_keep, void* workspace, size_t workspace_bytes, void* stream);

/* Health words of the multi-workgroup top-k selection (A >= 32 768), for a host that has synchronised anyway: returns
 * the byte offset from `workspace` of image 0's three uint32 {keys collected, keys announced, stores refused by the
 * bounds guard}; *stride_bytes = distance between images.  A healthy call leaves collected == announced == K and
 * refused == 0; anything else makes the sort kernel discard the pre-selection and select by itself (same result).  */
size_t mrcnn_proposal_status_offset(const mrcnn_proposal_desc* d, const void* workspace, size_t* stride_bytes);

/* DetectionTargetLayer (mrcnn/model.py:570-763) for one batch.  rand_keys [B, R] uniform floats
 * replace tf.random.shuffle: candidates are taken in increasing key order (ties -> lower index).
 * gt_masks is the reference layout [B, MH, MW, G] (uint8 0/1).  Outputs are zero padded to T.       */
typedef struct mrcnn_dettarget_desc {
    int32_t B, R, G, T, MH, MW, mask_h, mask_w;
    int32_t positive_count;      /* int(TRAIN_ROIS_PER_IMAGE * ROI_POSITIVE_RATIO), model.py:635 */
    float negative_ratio_r;      /* float32(1.0 / ROI_POSITIVE_RATIO), model.py:641-642 */
    float bbox_std_dev[4];
    int32_t use_mini_mask;
} mrcnn_dettarget_desc;
int mrcnn_detection_targets(const mrcnn_dettarget_desc* d, const float* proposals,
                            const int32_t* gt_class_ids, const float* gt_boxes,
                            const uint8_t* gt_masks, const float* rand_keys, float* rois,
                            int32_t* target_class_ids, float* target_bbox, float* target_mask,
                            int32_t* roi_gt_assignment, int32_t* counts, void* stream);

/* ---- 16-bit matrix-core path (BASELINE.json configs[4]; stage 1: the convolutions of the ROI heads) ------------
 * Operands float16 / bfloat16, accumulation + bias + frozen-BN affine + activation float32, one rounding of the
 * result.  Shapes: Cin % 32 == 0, Cout % 128 == 0, dense NHWC output, no residual (the mask / class head layers).
 * w_t is the transposed weight image [Cout][KH*KW*Cin] written by mrcnn_weights_to_h16 (wt_fwd); its second
 * output wt_dgrad [Cin][KH*KW*Cout] (taps rotated by 180 degrees) is the operand of the data gradient, which is
 * the same convolution applied to dz with Cin/Cout swapped and padding (k-1)/2.  Either output may be NULL.   */
#define MRCNN_DTYPE_F16 0
#define MRCNN_DTYPE_BF16 1
int mrcnn_conv2d_fwd_h16(const mrcnn_conv_desc* d, int dtype, const void* x, const void* w_t, const float* bias,
                         const float* scale, const float* shift, void* out, void* z_out, void* stream);
/* The same with a 16-bit residual `res` (shape and strides of out; d->res_mode must be MRCNN_RES_SAME exactly when res is
 * given) added before the activation -- the shortcut of a bottleneck block (mrcnn/model.py:99-131), or the accumulating
 * input of a data gradient -- and with arbitrary output strides (a stride-2 1x1 data gradient scatters into a zeroed
 * tensor).  Shapes: as above, or the small-tile kernel for the trunk's layers: Cin % 64 == 0, Cout % 64 == 0, plain NHWC
 * (strided allowed), the only one with the residual port.  mrcnn_conv2d_fwd_h16_supported() tells whether a descriptor
 * (with / without residual) has a 16-bit kernel.                                                                     */
int mrcnn_conv2d_fwd_h16_res(const mrcnn_conv_desc* d, int dtype, const void* x, const void* w_t, const float* bias,
                             const float* scale, const float* shift, const void* res, void* out, void* z_out, void* stream);
int mrcnn_conv2d_fwd_h16_supported(const mrcnn_conv_desc* d, int has_res);
/* mrcnn_conv2d_dgrad_ep in 16 bits (small-tile kernel: Cin % 64 == 0, Cout % 64 == 0, stride 1, dense output): the data
 * gradient y = conv(dz, w_t) (+ res), fused with the epilogue backward of the layer below -- stored is
 * dz_below = y * act'(out_below) * scale_below (and dy = y * act' when asked); dbeta / dgamma / dbias accumulate in
 * float32, multiplied by grad_multiplier (1 / loss scale).  MRCNN_ERR_UNSUPPORTED for other shapes.                */
typedef struct mrcnn_bwd_epilogue_h16 {
    const void* out; const void* z;              /* 16-bit activated output / pre-BN value of the layer below */
    const float* scale; const float* mean; const float* rstd;
    float* dgamma; float* dbeta; float* dbias;
    int32_t act;                                 /* MRCNN_ACT_NONE or MRCNN_ACT_RELU */
    void* dy;                                    /* optional second output (16 bit) */
    float grad_multiplier;
} mrcnn_bwd_epilogue_h16;
int mrcnn_conv2d_dgrad_ep_h16(const mrcnn_conv_desc* d, int dtype, const void* dz, const void* w_t, const void* res,
                              void* dz_below, const mrcnn_bwd_epilogue_h16* ep, void* stream);
int mrcnn_weights_to_h16(const float* w_hwio, void* wt_fwd, void* wt_dgrad, int KH, int KW, int Cin, int Cout,
                         int dtype, void* stream);
/* All 16-bit weight images of a model in one launch.  table: n_layers records {int64 offset of the float32 HWIO kernel in
 * `params` (floats); uint64 W^T image pointer; uint64 data-gradient image pointer or 0; int32 KH, KW, Cin, Cout, first_tile,
 * pad} (48 bytes each), first_tile = running sum of KH*KW*ceil(Cin/32)*ceil(Cout/32); total_tiles = the sum over all layers. */
int mrcnn_weights_to_h16_batched(const float* params, const void* table, int n_layers, int total_tiles, int dtype, void* stream);
/* Weight gradient with 16-bit operands x [N,H,W,Cin], dy [N,OH,OW,Cout]: dw (float32, HWIO) = multiplier * sum
 * (or dw += ... with beta_acc); float32 accumulation, slabs per pixel split summed in a fixed order.
 * Cin % 256 == 0, Cout % 128 == 0.  The workspace also holds a per-pixel offset / tap-mask table.           */
size_t mrcnn_conv2d_wgrad_h16_workspace(const mrcnn_conv_desc* d);
int mrcnn_conv2d_wgrad_h16(const mrcnn_conv_desc* d, int dtype, const void* x, const void* dy, float* dw,
                           void* workspace, size_t workspace_bytes, int beta_acc, float multiplier, void* stream);
/* mrcnn_epilogue_bwd on 16-bit tensors (act NONE / RELU, C a power of two >= 16): dz_out 16 bit; the float32 channel
 * sums dgamma / dbeta / dbias are multiplied by grad_multiplier (1 / loss scale) before they are added.          */
int mrcnn_epilogue_bwd_h16(int dtype, const void* dout, const void* out, const void* z, const float* scale,
                           const float* mean, const float* rstd, void* dz_out, float* dgamma, float* dbeta,
                           float* dbias, int64_t M, int C, int act, float grad_multiplier, void* stream);
/* The same with a second output dy_out = dout * act'(out) (before the BatchNorm scale): what the shortcut of a bottleneck
 * block receives when the block's last convolution is differentiated (mrcnn/model.py:128-130, Add + Activation).      */
int mrcnn_epilogue_bwd_h16_dy(int dtype, const void* dout, const void* out, const void* z, const float* scale,
                              const float* mean, const float* rstd, void* dz_out, void* dy_out, float* dgamma, float* dbeta,
                              float* dbias, int64_t M, int C, int act, float grad_multiplier, void* stream);
/* Output stage of the mask head on a 16-bit deconvolution output `up` [M,H,W,Cd] (mrcnn_conv2d_fwd_h16 with out_mode
 * MRCNN_OUT_DECONV2 writes it): mask_out [M,H,W,C] float32 = sigmoid(up . w_mask + b_mask); and the one-pass backward
 * of mrcnn_mask_out_bwd with dzg written in 16 bits times loss_scale (the sums dw_mask / db_mask / db_deconv stay
 * float32 and unscaled).  Cd % 256 == 0 (forward), C <= 16.                                                        */
int mrcnn_mask_out_fwd_h16(int dtype, const void* up, const float* w_mask, const float* b_mask, float* mask_out,
                           int64_t npix, int Cd, int C, void* stream);
int mrcnn_mask_out_bwd_h16(int dtype, const float* d_mask_out, const float* mask_out, const void* up, const float* w_mask,
                           void* dzg, float* dw_mask, float* db_mask, float* db_deconv, int64_t M, int H, int W, int Cd,
                           int C, float loss_scale, void* stream);
int mrcnn_cast_to_h16(const float* src, void* dst, int64_t n, int dtype, float multiplier, void* stream);
int mrcnn_cast_from_h16(const void* src, float* dst, int64_t n, int dtype, float multiplier, void* stream);
/* dst[i] += multiplier * src[i]: a 16-bit data gradient (scaled by the loss scale) added to a float32 accumulator, e.g.
 * the RPN's contribution to the pyramid gradients the ROI heads have already written (fan-out of P2..P5, model.py:2040). */
int mrcnn_axpy_from_h16(const void* src, float* dst, int64_t n, int dtype, float multiplier, void* stream);

/* build_rpn_targets (mrcnn/model.py:1536-1644), the per-image RPN target builder of the CPU input
 * pipeline, for one batch on the device.  anchors [A,4] float64 pixels (utils.generate_pyramid_anchors);
 * gt_class_ids [B,G] (>0 instance, <0 crowd, 0 padding); gt_boxes [B,G,4] int32 pixels; rand_keys [B,A]
 * uniform floats replace the two np.random.choice draws: when more than n_train/2 positives (or more than
 * n_train - positives negatives) exist, the surplus with the SMALLEST keys is reset to neutral (ties ->
 * lower anchor index).  Outputs: rpn_match [B,A] in {-1,0,1}; rpn_bbox [B,n_train,4] = deltas of the
 * positives in ascending anchor order / bbox_std_dev, zero padded.  IoUs are float64 in the reference's
 * operation order, so rpn_match is bit-exact.  G <= 512.                                              */
typedef struct mrcnn_rpntarget_desc {
    int32_t B, A, G, n_train;    /* n_train = RPN_TRAIN_ANCHORS_PER_IMAGE */
    double bbox_std_dev[4];      /* RPN_BBOX_STD_DEV (float64 in the reference) */
} mrcnn_rpntarget_desc;
size_t mrcnn_rpn_targets_workspace(const mrcnn_rpntarget_desc* d);
int mrcnn_rpn_targets(const mrcnn_rpntarget_desc* d, const double* anchors, const int32_t* gt_class_ids,
                      const int32_t* gt_boxes, const float* rand_keys, int32_t* rpn_match, float* rpn_bbox,
                      void* workspace, size_t workspace_bytes, void* stream);

/* DetectionLayer / refine_detections_graph (mrcnn/model.py:770-909): detections [B, max_inst, 6]. */
typedef struct mrcnn_detection_desc {
    int32_t B, R, C, max_instances;
    float min_confidence, nms_threshold;
    float bbox_std_dev[4];
} mrcnn_detection_desc;
size_t mrcnn_detection_workspace(const mrcnn_detection_desc* d);
int mrcnn_detection_fwd(const mrcnn_detection_desc* d, const float* rois, const float* probs,
                        const float* deltas, const float* windows, float* detections,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The five losses and their gradients w.r.t. the network outputs (mrcnn/model.py:1098-1270).
 * losses[5] = rpn_class, rpn_bbox, mrcnn_class, mrcnn_bbox, mrcnn_mask (each already the batch mean).
 * loss_weights scale the gradients (LOSS_WEIGHTS * USE_LOSSES).  Gradients are written dense.        */
typedef struct mrcnn_loss_desc {
    int32_t B, A, T, C, mask_h, mask_w, max_rpn_pos;   /* max_rpn_pos = rows of input_rpn_bbox */
    int32_t mask_loss_dice;
    float w[5];
} mrcnn_loss_desc;
size_t mrcnn_losses_workspace(const mrcnn_loss_desc* d);
int mrcnn_losses_fwd_bwd(const mrcnn_loss_desc* d, const int32_t* rpn_match, const float* rpn_bbox_t,
                         const float* rpn_class_logits, const float* rpn_bbox,
                         const int32_t* target_class_ids, const float* target_bbox,
                         const float* target_mask, const int32_t* active_class_ids,
                         const float* mrcnn_class_logits, const float* mrcnn_bbox,
                         const float* mrcnn_mask, float* losses, float* d_rpn_class_logits,
                         float* d_rpn_bbox, float* d_mrcnn_class_logits, float* d_mrcnn_bbox,
                         float* d_mrcnn_mask, void* workspace, size_t workspace_bytes, void* stream);

/* keras.optimizers.SGD(lr, momentum, clipnorm) + L2/numel regulariser (mrcnn/model.py:2255-2291) on
 * flat buffers of n floats (n % 64 == 0).  gran_coef[n/64] describes each 64-float granule (every
 * tensor starts on a granule): >= 0 trainable with that L2 gradient coefficient (2*WEIGHT_DECAY/numel,
 * 0 for gamma/beta); < 0 frozen tensor or alignment padding (gradient forced to 0, parameter kept).
 *   mrcnn_grad_prepare: grads = grads*grad_scale + coef*params (grad_scale = 1/world averages the
 *     data-parallel sum) and *sumsq_out = sum(grads^2), reduced in a fixed order so that every rank
 *     derives the same clip factor;
 *   mrcnn_sgd_momentum: g *= clipnorm/norm if norm >= clipnorm; v = momentum*v - lr*g; w += v.       */
int mrcnn_grad_prepare(float* grads, const float* params, float grad_scale, const float* gran_coef,
                       int64_t n, float* sumsq_out, void* stream);
int mrcnn_sumsq(const float* g, int64_t n, float* out_scalar, void* stream);
int mrcnn_sgd_momentum(float* params, float* momentum_buf, const float* grads, const float* sumsq,
                       float clipnorm, float lr, float momentum, const float* gran_coef, int64_t n,
                       void* stream);
/* Mixed-precision form (BASELINE configs[4]; no counterpart in the float32 reference): when *sumsq is not finite -- a
 * float16 gradient overflowed under the loss scale -- the update is skipped on the device (weights and momentum untouched,
 * no host synchronisation) and *skipped_steps (device, 32-bit) is incremented.                                              */
int mrcnn_sgd_momentum_guarded(float* params, float* momentum_buf, const float* grads, const float* sumsq,
                               float clipnorm, float lr, float momentum, const float* gran_coef, int64_t n,
                               unsigned* skipped_steps, void* stream);

/* ---- Winograd F(m x m, 3x3), m = `tile` = 2 or 4, for the stride-1 "same" 3x3 convolutions of the mask head
 * (KL.Conv2D(256, (3, 3), padding="same"), mrcnn/model.py:1058-1082), float32: (m + 2)^2 multiplications per m x m outputs and
 * channel pair instead of 9 m^2 -- 16 per 4 (tile 2) or 36 per 16 (tile 4).  Three launches per layer -- input transform, the
 * (m + 2)^2 transform-domain GEMMs as ONE launch, output transform with the layer's epilogue (or, for a data gradient, the epilogue
 * backward of the layer below) -- and one weight transform per weight update.  C % 4 == 0 (GEMM: K % 16 == 0, Cout % 128 == 0);
 * tile 2 wants H, W even, tile 4 takes any extent (edge tiles hang over the map: their inputs read as zero, their outputs are
 * dropped).  V / Mt hold mrcnn_winograd_buffer_floats floats each: [(m + 2)^2][rows][C], rows = tiles rounded up to whole 128-row
 * tiles.  Tile 2 uses +-1 and 1/2 only: the result differs from the direct kernels by float32 summation order (measured 2e-6 of
 * the output maximum); tile 4 has constants up to 8 and costs about one more decimal digit (measured 1e-5).                       */
size_t mrcnn_winograd_buffer_floats(int N, int H, int W, int C, int tile);
int mrcnn_winograd_weights(const float* w_hwio, float* U, int Cin, int Cout, int tile, void* stream);   /* U [(m + 2)^2][Cin][Cout] */
int mrcnn_winograd_input(const float* x, float* V, int N, int H, int W, int C, int tile, void* stream);
int mrcnn_gemm_batched_f32(const float* V, const float* U, float* Mt, int nb, int rows, int K, int Cout, void* stream);
/* the same product by persistent workgroups (next tile's first stage in flight under the current tile's stores): the default */
int mrcnn_winograd_gemm(const float* V, const float* U, float* Mt, int nb, int rows, int K, int Cout, void* stream);
int mrcnn_winograd_output(const float* Mt, float* out, float* z_out, const float* bias, const float* scale, const float* shift,
                          int N, int H, int W, int C, int act, int tile, void* stream);
/* weight gradient: dM = A dy A^T (adjoint of the output transform) [(m + 2)^2][rows][C]; dU[xi] = V[xi]^T . dM[xi] are 1 x 1 weight
 * gradients (mrcnn_conv2d_wgrad on the first `tiles` rows of each matrix); dW = G^T dU G (accumulate != 0: added to dW).       */
int mrcnn_winograd_dy(const float* dy, float* dM, int N, int H, int W, int C, int tile, void* stream);
int mrcnn_winograd_dw(const float* dU, float* dw_hwio, int Cin, int Cout, int accumulate, int tile, void* stream);
int mrcnn_winograd_output_bwd(const float* Mt, float* dz_below, const float* below_out, const float* below_z, const float* scale,
                              const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dbias, int N, int H, int W,
                              int C, int act, int tile, void* stream);

/* The same passes for one GROUP of tiles -- th_n x tw_n tiles per map of oth x otw outputs (2 or 4 each), the first at output
 * (oh0, ow0) -- so that a map whose extent is not a multiple of 4 is covered without overhang: 14 x 14 = 3 x 3 tiles of 4 x 4,
 * 3 x 1 of 4 x 2 (columns 12..13), 1 x 3 of 2 x 4 (rows 12..13), one of 2 x 2: 484 multiplications per map and channel pair instead of
 * 576 (uniform 4 x 4 with overhang) or 784 (uniform 2 x 2).  Every group has its own V / Mt ((oth + 2)(otw + 2) matrices of
 * mrcnn_winograd_group_floats / C rows) and its own weight transform U = G_oth g G_otw^T; the groups of a map write disjoint
 * outputs, their channel sums (output_bwd) and weight gradients (dw with accumulate) add up.  The `tile` entries above are the
 * one-group case.                                                                                                               */
typedef struct mrcnn_wino_group {
    int oth, otw;      /* outputs per tile, rows / columns: 2 or 4 */
    int th_n, tw_n;    /* tiles per map */
    int oh0, ow0;      /* output row / column of the first tile */
} mrcnn_wino_group;
size_t mrcnn_winograd_group_floats(const mrcnn_wino_group* g, int N, int C);
int mrcnn_winograd_input_g(const float* x, float* V, int N, int H, int W, int C, const mrcnn_wino_group* g, void* stream);
int mrcnn_winograd_weights_g(const float* w_hwio, float* U, int Cin, int Cout, int oth, int otw, void* stream);
int mrcnn_winograd_output_g(const float* Mt, float* out, float* z_out, const float* bias, const float* scale, const float* shift,
                            int N, int H, int W, int C, int act, const mrcnn_wino_group* g, void* stream);
int mrcnn_winograd_output_bwd_g(const float* Mt, float* dz_below, const float* below_out, const float* below_z, const float* scale,
                                const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dbias, int N, int H, int W,
                                int C, int act, const mrcnn_wino_group* g, void* stream);
/* The same when the layer below is a ReLU layer with a frozen-BN affine (forward epilogue out = max(scale * z + shift, 0)): the
 * ReLU mask is recomputed from the stored z with the forward's own expression, so the activated output is not read. */
int mrcnn_winograd_output_bwd_zmask_g(const float* Mt, float* dz_below, const float* below_z, const float* scale, const float* shift,
                                      const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dbias, int N, int H,
                                      int W, int C, const mrcnn_wino_group* g, void* stream);

/* The transform-domain GEMMs of a 4 x 4 tile group WITH the group's output transform in the same launch (round 3): tiles are
 * walked row-block-major, every workgroup counts its finished tile into its row block's counter and the workgroup that completes
 * a row block transforms its 128 tiles on the spot -- no waiting, no second launch; the transform's memory pass runs beside the
 * other workgroups' MFMAs.  mode 1 = mrcnn_winograd_gemm + mrcnn_winograd_output_g, mode 2 = mrcnn_winograd_gemm +
 * mrcnn_winograd_output_bwd_g / _zmask_g (shift != NULL), same results up to the order of the channel-sum atomics.  Cout = 256,
 * a 4 x 4 group (36 matrices); else MRCNN_ERR_UNSUPPORTED.  counters: rows / 128 ints, zero on entry (left zero).           */
typedef struct mrcnn_wino_fuse {
    int32_t mode, N, H, W, act;
    mrcnn_wino_group g;
    float* out;              /* mode 1: out [N,H,W,256]; mode 2: dz of the layer below */
    float* z;                /* mode 1: pre-BN output or NULL */
    const float* bias; const float* scale; const float* shift;
    const float* below_out; const float* below_z; const float* mean; const float* rstd;
    float* dgamma; float* dbeta; float* dbias;
    int32_t* counters;
} mrcnn_wino_fuse;
int mrcnn_winograd_gemm_fused(const float* V, const float* U, float* Mt, int nb, int rows, int K, int N, const mrcnn_wino_fuse* f,
                              void* stream);
int mrcnn_winograd_dy_g(const float* dy, float* dM, int N, int H, int W, int C, const mrcnn_wino_group* g, void* stream);
int mrcnn_winograd_dw_g(const float* dU, float* dw_hwio, int Cin, int Cout, int accumulate, int oth, int otw, void* stream);

/* KL.Conv2DTranspose(256, (2, 2), strides=2, activation="relu") of build_fpn_mask_graph (mrcnn/model.py:1084-1086) as one product
 * [N H W x Cin] . [Cin x 4 Cd] on the persistent GEMM (mrcnn_winograd_gemm's kernel) with bias, activation and the pixel-shuffle
 * store in its epilogue: K is a single filter tap deep, where the persistent form beats the one-tile-per-workgroup kernel
 * (2.15 -> 1.75 ms at 2048 ROIs).  w_gemm [Cin][(a, b, co)] (the layout mrcnn_conv2d_fwd takes for MRCNN_OUT_DECONV2);
 * out [N][2 H][2 W][Cd].  Cin % 16 == 0, Cd % 32 == 0, 4 Cd % 128 == 0, act NONE or RELU; else MRCNN_ERR_UNSUPPORTED.          */
int mrcnn_deconv2x2_gemm(const float* x, const float* w_gemm, const float* bias, float* out, int N, int H, int W, int Cin, int Cd,
                         int act, void* stream);

/* ---- data-parallel gradient exchange over RCCL / xGMI ---------------------------------------------------------------
 * Replaces the in-graph tower aggregation of mrcnn/parallel_model.py:54-104 (weights shared between towers, gradients
 * summed implicitly by TF, scalar losses averaged :97-99): one process per GPU, each rank sums contiguous ranges
 * [start, end) of its flat float32 gradient buffer with all peers, in place, on `stream`; the division by the world size
 * happens in mrcnn_grad_prepare (grad_scale = 1 / world).  RCCL is bound at run time:
 *   mrcnn_allreduce_load(path)   dlopen + dlsym; path = the librccl the process already uses (PyTorch's), or NULL for
 *                                the default search.  MRCNN_ERR_UNSUPPORTED when no RCCL can be loaded.
 *   mrcnn_allreduce_unique_id    rank 0 creates the 128-byte rendezvous id; the caller hands it to every rank (any
 *                                side channel: a torch.distributed broadcast, a file, MPI).
 *   mrcnn_allreduce_init         ncclCommInitRank on the calling thread's current HIP device; collective over all ranks.
 *   mrcnn_allreduce_grad         algo MRCNN_ALLREDUCE_RCCL: ncclAllReduce(sum);  MRCNN_ALLREDUCE_DIRECT: reduce-scatter +
 *                                all-gather as grouped point-to-point transfers, one per xGMI link, chunks summed in rank
 *                                order by the owner (bitwise identical on every rank); needs mrcnn_allreduce_scratch()
 *                                bytes of device scratch.  Enqueues only; never synchronises.
 * Calls on one communicator must be issued in the same order on every rank (RCCL's rule).                          */
#define MRCNN_UNIQUE_ID_BYTES 128
#define MRCNN_ALLREDUCE_RCCL 0
#define MRCNN_ALLREDUCE_DIRECT 1
int mrcnn_allreduce_load(const char* librccl_path);
int mrcnn_allreduce_unique_id(void* id_128_bytes);
int mrcnn_allreduce_init(void** comm, const void* id_128_bytes, int rank, int world);
size_t mrcnn_allreduce_scratch(int world, int64_t max_range_floats, int algo);
int mrcnn_allreduce_grad(void* comm, float* grads, int64_t start, int64_t end, int algo, float* scratch,
                         size_t scratch_bytes, void* stream);
/* The direct form as data (no GPU, no RCCL needed): plan[p*4 .. p*4+3] = (send_off, send_len, recv_off, recv_len) of `rank`'s
 * transfer with peer p in phase 0 (reduce-scatter: sends index the gradient buffer, receives the scratch buffer) or phase 1
 * (all-gather: both index the gradient buffer); zeros for p == rank.  own_off / own_len: the chunk `rank` sums, slot_stride:
 * floats between scratch slots.  mrcnn_allreduce_grad walks the same function.                                            */
int mrcnn_allreduce_direct_plan(int world, int rank, int64_t start, int64_t end, int phase, int64_t* plan, int64_t* own_off,
                                int64_t* own_len, int64_t* slot_stride);
/* Test entry: the direct exchange among `world` fabricated ranks that all live on the calling GPU (grads[r], scratch[r]:
 * host arrays of device pointers); transfers become device-to-device copies after the send / receive lengths of every pair
 * have been checked against each other, the sums run the real kernel.  2 <= world <= 64.                                 */
int mrcnn_allreduce_direct_simulate(float* const* grads, float* const* scratch, size_t scratch_bytes, int world, int64_t start,
                                    int64_t end, void* stream);
int mrcnn_allreduce_destroy(void* comm);
const char* mrcnn_allreduce_last_error(void);

/* Process-wide tuning values, read by the host side of later launches.  Keys: "wgrad_lds_pad" (0..32768 bytes of extra
 * LDS per workgroup of the large weight-gradient kernel: 8192 caps it at four workgroups per CU so that kernels of another
 * stream find a free slot on every CU); "h16_phase" (0 / 1); "h16_slab" (-1 = MRCNN_H16_SLAB decides, default off; 0 / 1: the slab form of the phased 16-bit kernel off /
 * on -- tests and A/B timing); "sk16" (0 / 1: the 16 x 16-tile single-launch kernel for layers of a few hundred pixels may be
 * picked -- the engine sets it while it issues the inference graph); "proposal_skip_zero" (tests only, fault injection: 1 suppresses
 * the counter reset of mrcnn_proposal_fwd's multi-workgroup selection).  MRCNN_ERR_UNSUPPORTED for an unknown key.   */
int mrcnn_tuning_set(const char* key, long long value);

const char* mrcnn_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MRCNN_HIP_H */
