/* relnet_hip.h -- C-ABI of librelnet_hip.so: the MI355X (gfx950) hot path of
 * msracver/Relation-Networks-for-Object-Detection.
 *
 * Plain pointers and sizes only (no torch / MXNet types).  Unless a function says "host", every
 * pointer is DEVICE memory and the work is enqueued on `stream` (a hipStream_t passed as void*;
 * NULL = the default stream) without any host synchronisation, so every entry point can be
 * captured in a hipGraph.  Return value: 0 on success, negative on error; the message of the
 * last error on the calling thread is relnet_last_error().  dtype codes: RELNET_F32 = 0,
 * RELNET_BF16 = 1 (bf16 operands, fp32 accumulation).
 *
 * Citations are file:line in the reference repository (the interface or code each entry point
 * replaces).  SYM_REL = relation_rcnn/symbols/
 * resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py.
 */
#ifndef RELNET_HIP_H
#define RELNET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum { RELNET_F32 = 0, RELNET_BF16 = 1 };

int relnet_version(void);                 /* 100 = 0.1.0 */
const char* relnet_last_error(void);
/* 0 when `stream` is not being captured into a hipGraph, else the id of the capture sequence (hipStreamGetCaptureInfo) */
unsigned long long relnet_stream_capture_id(void* stream);

/* ---- lib/nms/gpu_nms.hpp:1-2 (the reference's own C prototype, bound by gpu_nms.pyx:15-16) -----
 * HOST pointers.  boxes_host: boxes_num rows [x1,y1,x2,y2,score] pre-sorted by score (descending);
 * keep_out[boxes_num] receives the kept row indices in ascending order, *num_out their count.
 * Suppression rule of nms_kernel.cu:24-32,71: IoU with +1 pixel extents, strictly greater than
 * nms_overlap_thresh.  On a HIP error *num_out = -1 (the reference only prints CUDA errors).     */
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
          float nms_overlap_thresh, int device_id);

/* ---- operator_py/proposal.py:51-168 (ProposalOperator.forward), split in four device stages ----
 * decode: anchors in (y, x, a) order over the h x w cropped grid (:85,113-123), float64 decode of
 * lib/bbox/bbox_transform.py:103-140, clip (:45-60), min-size filter (:133-135, score := -inf).
 * strides4 = element strides (b, c, y, x) of the [B,2A,H,W] / [B,4A,H,W] maps (host arrays of 4 longs).
 * softmax_pairs != 0: cls_prob holds raw rpn_cls_score and the 2-way softmax of SYM_REL:218-223 is
 * applied on the fly.                                                                            */
int relnet_proposal_decode(const float* cls_prob, const long* cls_strides4, const float* deltas,
                           const long* delta_strides4, const float* im_info, const double* base_anchors,
                           float* boxes /*[B,n,4]*/, float* scores /*[B,n]*/, int B, int A, int h, int w,
                           int feat_stride, int min_size, int softmax_pairs, void* stream);
/* descending top-K (proposal.py:140-144): out_boxes5 [B,K,5] = the float32 `det` of :149,
 * out_index [B,K] anchor indices, out_count [B] entries with finite score.  n < 65536, K <= 8192.
 * Equal scores are ordered by descending index (the reference's argsort is unstable).            */
int relnet_topk_sort(const float* scores, const float* boxes, float* out_boxes5, int* out_index,
                     int* out_count, int B, int n, int K, void* stream);
/* nms_kernel.cu:34-78 (bitmask, upper triangle only); mask [B, n, ceil(n/64)] uint64              */
int relnet_nms_mask(const float* boxes5, const int* counts /*[B] or NULL*/, unsigned long long* mask,
                    int B, int n, int n_stride, float thresh, void* stream);
/* nms_kernel.cu:118-140 (greedy scan) on device + proposal.py:151-168 (first `post` kept boxes, pad,
 * batch index column).  rois [B,post,5], roi_scores [B,post], keep [B,max_keep] (any may be NULL). */
int relnet_nms_scan(const unsigned long long* mask, const float* boxes5, const int* counts, float* rois,
                    float* roi_scores, int* keep, int* num_keep, int B, int n, int n_stride, int post,
                    int max_keep, int batch_index_base, void* stream);

/* nms_kernel.cu:24-32,118-140 + proposal.py:151-153 fused: greedy NMS that keeps only the first
 * `post` (<= 2048) boxes, testing each 64-box block against the kept list instead of building the
 * n x n bitmask; identical keep list to relnet_nms_mask + relnet_nms_scan(post).                    */
int relnet_nms_greedy(const float* boxes5, const int* counts, float* rois, float* roi_scores, int* keep,
                      int* num_keep, int B, int n, int n_stride, int post, float thresh,
                      int batch_index_base, void* stream);

/* ---- ROIAlign (named by BASELINE.json's north_star; the reference graphs pool with ROIPooling, SYM_REL:252-253, and ship no ROIAlign:
 * this follows the published algorithm -- He et al., Mask R-CNN (2017), section 3, as in Detectron's RoIAlign / mx.contrib.sym.ROIAlign(data,
 * rois, pooled_size, spatial_scale, sample_ratio) of MXNet >= 1.3): no coordinate rounding, sampling_ratio^2 bilinear samples per bin
 * (<= 0: ceil(roi extent / pooled size)), averaged; `aligned` subtracts the half-pixel offset.  Strides as relnet_roi_pool_fwd.  Backward:
 * grad_in fp32, pre-zeroed, strides (b, c, y, x) in elements; every sample scatters its four bilinear weights (atomics).              */
int relnet_roi_align_fwd(const void* data, const long* data_strides4, const float* rois /*[R,5]*/, void* out, const long* out_strides4,
                         int R, int C, int H, int W, int PH, int PW, float spatial_scale, int sampling_ratio, int aligned,
                         int batch_index_base, int dtype, void* stream);
int relnet_roi_align_bwd(const void* grad_out, const long* out_strides4, const float* rois, float* grad_in,
                         const long* grad_in_strides4, int R, int C, int H, int W, int PH, int PW, float spatial_scale,
                         int sampling_ratio, int aligned, int batch_index_base, int dtype, void* stream);

/* ---- mx.symbol.ROIPooling(pooled_size=(7,7), spatial_scale=1/16), SYM_REL:252-253 ---------------
 * data/out described by element strides (b|r, c, y, x) so NCHW and channels-last both work.       */
int relnet_roi_pool_fwd(const void* data, const long* data_strides4, const float* rois /*[R,5]*/, void* out,
                        const long* out_strides4, int* argmax /*or NULL*/, int R, int C, int H, int W, int PH,
                        int PW, float spatial_scale, int batch_index_base, int dtype, void* stream);
int relnet_roi_pool_bwd(const void* grad_out, const int* argmax, const long* out_strides4, const float* rois,
                        float* grad_in /*fp32, pre-zeroed*/, long gs_b, long gs_c, int R, int C, int W, int PH,
                        int PW, int batch_index_base, int dtype, void* stream);

/* ---- mx.symbol.FullyConnected (fc_new_1/2, cls_score, bbox_pred, query_i, key_i: SYM_REL:254-280,
 * 120-129) and the grouped linear_out_i product (:146-150): C = A W^T (+bias)(+resid)(ReLU).
 * A [batch][M,K] (lda, strideA), W [batch][N,K], C [batch][M,N]; bias_mode 0 none / 1 per column /
 * 2 per row; resid has C's layout and dtype.  bf16: K % 64 == 0; f32 (exact fp32 MFMA): K % 16 == 0. */
int relnet_gemm_nt(const void* A, long lda, long strideA, const void* W, long ldw, long strideW, void* C,
                   long ldc, long strideC, const float* bias, int bias_mode, const void* resid, int relu,
                   int M, int N, int K, int batch, int in_dtype, int out_dtype, void* stream);
void relnet_gemm_force_tile(int cfg);     /* tuning knob: 0 = auto, 1..relnet_gemm_tile_count() = fixed tile configuration */
int relnet_gemm_get_forced_tile(void);    /* the value last given to relnet_gemm_force_tile (0 = auto): lets a caller restore it */
void relnet_gemm_force_nloop(int n);      /* tuning knob: 0 = auto, n = column tiles per workgroup     */
void relnet_gemm_set_swizzle(int on);     /* tuning knob: XCD-aware tile order (default 1)               */
void relnet_gemm_debug_korder(int on);    /* tuning knob: (channel chunk, tap) k order of the spatial ring convolutions (default 1) */
void relnet_gemm_debug_asm(int on);       /* tuning knob: 0 = the automatic tile choice never takes tiles 18 / 19 (hand-scheduled k-loops); default 1 */
void relnet_gemm_debug_phase_ts(void* buf); /* measurement knob: the ring kernels (tiles 6 - 12, 16 - 19) write, per workgroup, 8 int64 words into buf --
                                             * wall clock (100 MHz) at entry / k-loop start / k-loop end / exit, [4] = shader cycles entry -> k-loop end;
                                             * NULL (default) = off.  tools/tile_phase_probe.py */
void relnet_gemm_debug_ablate(int a);     /* measurement knob for tile 8: 1 = fill path only, 2 = LDS + MFMA only (garbage results) */
void relnet_chain_debug(int flags);        /* measurement knob for chain256_roles_kernel (garbage results while non-zero): 1 = half of the weight loads, 2 = none, 4 = no shortcut-slice loads, 8 = no global stores */
int relnet_gemm_tile_count(void);         /* number of tile configurations (valid relnet_gemm_force_tile values 1..count) */
/* Work area of the split-K tile (configuration 23, round 6: launches of at most one 64 x 64 workgroup per CU with a long k-loop -- fc_new_1 and
 * rpn_conv_3x3 of the one-image step of core/tester.py:219-295 / train_end2end.py with BATCH_IMAGES: 1 -- split their k-loop over 3..4 workgroups per
 * tile; fp32 partial tiles and per-tile arrival counters live here).  `ws`: device memory owned by the caller, 256-byte aligned, its first 16 KB
 * ZERO when first handed over (the kernels leave them zero); NULL = none.  The pointer is remembered per HOST THREAD and used by the GEMM /
 * convolution entry points called after it; launches that share an area must be ordered with respect to each other (same stream, or the
 * dependencies of one captured graph): a caller that launches on several streams keeps one area per stream and names it before each call.
 * Without an area no launch is split.  Nothing is allocated by the library (capture-safe). */
int relnet_gemm_set_workspace(void* ws, long bytes);
void relnet_gemm_debug_splitk(int k);     /* tuning knob: 0 = auto (launches of <= 128 tiles with >= 128 k-slabs), 1 = never split, k >= 2 = k ways wherever configuration 23 runs, -2 = auto incl. 129..320-tile launches */
int relnet_gemm_pick_tile(int M, int N, int K, int batch, int out_dtype);   /* the configuration `auto` selects */

/* ---- mx.symbol.Convolution + BatchNorm(use_global_stats) + Activation of
 * relation_rcnn/symbols/resnet_v1_101_rcnn_base.py:29-693 as an NHWC implicit GEMM (BN folded by the
 * caller); `resid` fuses the bottleneck's broadcast_add + ReLU.  in [B,H,W,>=Cin] bf16 with pixel /
 * image strides in elements, w [Cout, R*S*Cin] (k = (r*S+s)*Cin + ic), out rows of ldc elements.    */
int relnet_conv2d_nhwc(const void* in, long in_pix, long in_img, const void* w, const float* bias,
                       const void* resid, int relu, void* out, long ldc, int B, int H, int W, int Cin, int Cout,
                       int R, int S, int stride, int dil, int pad, int out_dtype, void* stream);

/* The same convolution with float32 operands on the exact-fp32 MFMA kernel (v_mfma_f32_32x32x2f32: an fmaf chain per output
 * element): the float32 PARITY path of mx.symbol.Convolution (+ folded BatchNorm + Activation) for every convolution of the
 * graph -- resnet_v1_101_rcnn_base.py:29-693, the FPN neck symbols/..._fpn_...:804-840 -- in place of a library call.
 * in [B,H,W,>=Cin] fp32 (Cin % 16 == 0: the 3-channel stem image is zero-padded to 16 channels), w [Cout][R*S*Cin] fp32,
 * out / resid [B*Hout*Wout][ldc] fp32; relu: 0 none, 1 ReLU, 2 = resid is a ReLU mask (float32 data gradients).             */
int relnet_conv2d_nhwc_f32(const float* in, long in_pix, long in_img, const float* w, const float* bias, const float* resid,
                           int relu, float* out, long ldc, int B, int H, int W, int Cin, int Cout, int R, int S, int stride,
                           int dil, int pad, void* stream);
/* pool1 of that path: mx.symbol.Pooling(kernel 3x3, stride 2, max, pooling_convention='full') (resnet_v1_101_rcnn_base.py:35-36),
 * ceil-mode windows clipped at the border.  in [B,H,W,C] fp32 NHWC dense, C % 4 == 0 -> out [B,Ho,Wo,C].                    */
int relnet_maxpool_nhwc_f32(const float* in, float* out, int B, int H, int W, int C, int ksize, int stride, void* stream);

/* ---- conv1 7x7/2 pad 3, Cin = 3 (resnet_v1_101_rcnn_base.py:30-31) on the MFMA kernel: pack the
 * [B,3,H,W] image (fp32 or bf16) into zero-padded NHWC4 bf16 [B,Hp,Wp,4], then the implicit GEMM with
 * w256 [Cout][256], k = ty*32 + tx*4 + c (ops.pack_stem_weight), bias + ReLU fused.                  */
int relnet_stem_pack_input(const void* in, void* out, int B, int H, int W, int Hp, int Wp, int pad, int in_dtype,
                           void* stream);
int relnet_stem_conv7(const void* packed, const void* w256, const float* bias, int relu, void* out, long ldc,
                      int B, int Hp, int Wp, int Hout, int Wout, int Cout, int out_dtype, void* stream);

/* ---- bn_conv1 (folded) + conv1_relu + pool1 (3x3/2, pooling_convention='full'), resnet_v1_101_rcnn_base.py:
 * 30-36, fused into one pass over the NHWC stem output: out = relu(maxpool_ceil(in) + bias).        */
int relnet_stem_bias_relu_pool(const void* in, const float* bias, void* out, int B, int H, int W, int C,
                               int ksize, int stride, void* stream);
/* conv1 7x7/2 (pad 3) + folded-BN bias + ReLU + pool1 3x3/2 (pooling_convention='full') in ONE kernel, raw NCHW image ->
 * pooled NHWC bf16 map (resnet_v1_101_rcnn_base.py:30-36): bit-identical to relnet_stem_pack_input + relnet_stem_conv7 +
 * relnet_stem_bias_relu_pool without the conv map's HBM round trip.  data [B,3,H,W] (in_dtype 0 fp32 / 1 bf16),
 * w256 [64][256] bf16 with k = ty*32 + tx*4 + c (zero for ty = 7, tx = 7, c = 3), bias [64] fp32, out [B,Hp,Wp,64].    */
int relnet_stem_fused(const void* data, int in_dtype, const void* w256, const float* bias, void* out, int B, int H, int W,
                      void* stream);

/* ---- SYM_REL:46-83 extract_position_matrix + :29-44 extract_position_embedding + :109-116
 * pair_pos_fc1 + ReLU + the log(max(.,1e-6)) of :139, fused (the [N,M,64] embedding is never stored).
 * boxes [B,N,box_stride] with x1 at +box_off; wp_t [64, nmod*16] (embedding-index major), bp [nmod*16];
 * divisors8: HOST array, wave_length^(k/8) in fp32; bias [nmod,B,16,N,Mpad] fp32.
 * pos_mat [B,N,M,4] / pos_emb [B,N,M,64]: optional debug outputs (NULL to skip).                   */
int relnet_geometry_bias(const float* boxes, int box_stride, int box_off, const float* wp_t, const float* bp,
                         const float* divisors8, void* bias, int bias_half /* 0: float32 log G in the oracle's arithmetic (sin / cos / log correctly rounded, float64 accumulation: the parity path); 1: fp16 log2 G (bf16 throughput path, matrix cores); 2: float32 ln G from the same matrix-core kernel (training backward: the G the forward saw); -1: float32 log G with float32 libm arithmetic */, float* pos_mat,
                         float* pos_emb, int B, int N, int M, int Mpad, int fc_dim, int nmod, void* stream);

/* ---- SYM_REL:132-150: logits = bias + scale * Q K^T (`weighted_aff`), softmax over keys, value sum
 * and grouped linear_out (re-associated: vwt = (F_K Wout^T)^T, [B][H*64][Mpad], zero padded).
 * q/k rows of 64-wide heads at column h*64; out / out_act [B][N][H*64]; out_act = relu(resid + out)
 * (SYM_REL:267-268); logits [B][N][H][M] fp32 optional.  Any of out/out_act/logits may be NULL.
 * bf16 + fp16 bias (bias_half = 1) selects the LDS-shared throughput kernel (no logits output).     */
int relnet_relation_attention(const void* q, long q_ld, long q_bs, const void* k, long k_ld, long k_bs,
                              const void* vwt, long vwt_ld, long vwt_bs, const void* bias, int bias_half,
                              long bias_bs, const float* bout, const void* resid, long resid_ld, long resid_bs, void* out,
                              long out_ld, long out_bs, void* out_act, long act_ld, long act_bs, float* logits,
                              int B, int H, int N, int M, int Mpad, float scale, int in_dtype, int out_dtype,
                              void* stream);

/* Same, for a FIXED-SIZE roi buffer that holds a different number of real rows per image (the reference runs one image
 * per executor, so its graphs simply see a different roi count each time: the FPN loader's dummy roi of an empty level,
 * core/rcnn.py:61-71, TRAIN.TOP_ROIS truncation :128-146): key_count [B] (device, may be NULL) = keys of image b that exist;
 * keys in [key_count[b], M) are masked exactly like the columns past M.                                              */
int relnet_relation_attention_kc(const void* q, long q_ld, long q_bs, const void* k, long k_ld, long k_bs,
                                 const void* vwt, long vwt_ld, long vwt_bs, const void* bias, int bias_half,
                                 long bias_bs, const float* bout, const void* resid, long resid_ld, long resid_bs, void* out,
                                 long out_ld, long out_bs, void* out_act, long act_ld, long act_bs, float* logits,
                                 int B, int H, int N, int M, int Mpad, float scale, int in_dtype, int out_dtype,
                                 const int* key_count, void* stream);
/* tuning / test knob: 1 (default, round 6) = a bf16 launch with a FLOAT32 bias (ln G: the training forward) runs on the LDS-resident kernel like the fp16-bias
 * inference launches (no logits output); 0 = on the streaming kernel of rounds 1 - 5 */
void relnet_relation_attention_debug_lds_f32(int on);

/* Geometry + attention of ONE relation module in a single kernel (bf16 throughput path; csrc/relation.hip:
 * relation_fused_kernel): the position embedding (SYM_REL:29-83), pair_pos_fc1 + ReLU + log (:109-116, :139) and the
 * attention (:132-150) run per 32-query x 16-head workgroup; the [B][16][N][Mpad] bias tensor never reaches HBM.
 * boxes [B][N][box_stride] fp32 (xyxy at +box_off); wp [16][64] / bp [16] fp32 = pair_pos_fc1_<i>_{weight,bias};
 * divisors8 = wave_length^(t/8) (HOST pointer); other operands as relnet_relation_attention.  H = 16, M <= 640.  */
int relnet_relation_attention_fused(const void* q, long q_ld, long q_bs, const void* k, long k_ld, long k_bs,
                                    const void* vwt, long vwt_ld, long vwt_bs, const float* boxes, int box_stride,
                                    int box_off, const float* wp, const float* bp, const float* divisors8,
                                    const float* bout, const void* resid, long resid_ld, long resid_bs, void* out,
                                    long out_ld, long out_bs, void* out_act, long act_ld, long act_bs, int B, int H,
                                    int N, int M, int Mpad, float scale, void* stream);

/* Residual-block boundary of the ResNet trunk as ONE pixel-wise kernel (csrc/bottleneck.hip; reference graph
 * symbols/resnet_v1_101_rcnn_base.py: res<s><u>_branch2c + bn + shortcut + relu, then res<s><u+1>_branch2a + bn + relu):
 *   x_next = relu(W3 . mid2 + b3 + x),  mid1_next = relu(W1n . x_next + b1n)        (BN folded into W / b)
 * mid2 [P][mid], x / x_next [P][4 mid], mid1_next [P][mid] bf16, dense pixel rows.  w3f = relnet_pack_w_frag(W3 [4 mid][mid]);
 * w1f = W1n [mid][4 mid] in the accumulator-permuted fragment order: block (rt, ks) = 64 lanes x 8 values, lane (l31, half)
 * slot t <- W1n[32 rt + l31][16 ks + 8 (t >> 2) + 4 half + (t & 3)].  mid = 64 (res2), 128 (res3) or 256 (res4).  w1f = b1 =
 * mid1_next = NULL: only x_next (last unit of a stage; also mid = 512, res5).  x_next may alias x (in place).  The 4 mid-channel
 * map must stay below 4 GiB (32-bit byte offsets).                                                                   */
int relnet_bottleneck_chain(const void* mid2, const void* x, const void* w3f, const void* w1f, const float* b3,
                            const float* b1, void* x_next, void* mid1_next, long P, int mid, void* stream);
/* First unit of res2 (1x1 projection shortcut, stride 1; resnet_v1_101_rcnn_base.py: res2a_branch1 / bn2a_branch1 + res2a_branch2c
 * + shortcut + ReLU [+ res2b_branch2a + ReLU]): x_next = relu(conv1x1(mid2; W3) + conv1x1(x_in; Wp) + b3p) with b3p = b3 + bp, the
 * projection being four more k-steps of the expand product (its 256-channel output is never written); mid1_next as in
 * relnet_bottleneck_chain (or NULL).  mid = 64; w3f / wpf = relnet_pack_w_frag images of W3 / Wp [256][64]; x_in [P][64] dense bf16. */
int relnet_bottleneck_chain_proj(const void* mid2, const void* x_in, const void* w3f, const void* wpf, const void* w1f,
                                 const float* b3p, const float* b1, void* x_next, void* mid1_next, long P, int mid, void* stream);

/* 3x3 / stride 1 / pad 1 convolution + bias (+ ReLU) of a dense 64-channel NHWC bf16 tensor with the input tile and its halo
 * resident in LDS (res2*_branch2b + BN + ReLU, resnet_v1_101_rcnn_base.py:52-56): the input is fetched 1.33 x instead of once
 * per tap.  w_frag = relnet_pack_w_frag of the packed weight [64][576] (k = (r * 3 + s) * 64 + c).                        */
int relnet_conv3x3_c64(const void* in, const void* w_frag, const float* bias, int relu, void* out, int B, int H, int W,
                       void* stream);


/* Row-panel form of the 1x1 convolutions (csrc/gemm.hip:gemm_panelw_kernel): `w_frag` is the weight matrix re-ordered once
 * at model load by relnet_pack_w_frag ([Cout][K] bf16 -> MFMA fragment order, same byte count; Cout % 32 == 0, K % 16 == 0).
 * relnet_conv2d_nhwc_wf == relnet_conv2d_nhwc when w_frag is NULL or the layer is not a stride-1 1x1 convolution with
 * K in {64,128,256,512} and Cout % 256 == 0.                                                                        */
int relnet_pack_w_frag(const void* w, long ldw, void* out, int N, int K, void* stream);
int relnet_conv2d_nhwc_wf(const void* in, long in_pix, long in_img, const void* w, const void* w_frag, const float* bias,
                          const void* resid, int relu, void* out, long ldc, int B, int H, int W, int Cin, int Cout, int R,
                          int S, int stride, int dil, int pad, int out_dtype, void* stream);

/* ---- relation_rcnn/core/tester.py:148-156 (im_detect) + :244-277 (per-class NMS, max_per_image) ----
 * detect_head: SoftmaxActivation over classes + class-agnostic decode (bbox_transform.py:103-140,
 * float64) + clip + 1/scale; boxes [R,4] float64.                                                   */
int relnet_detect_head(const float* cls_score, long cs_ld, const float* bbox_pred, long bp_ld, const float* rois,
                       const float* im_info, float* cls_prob, double* boxes, int R, int C, int rois_per_image,
                       int delta_off, void* stream);
/* ..._ex: n_valid [B] (device, may be NULL): rows past n_valid[b] of image b are padding of a fixed-size roi buffer
 * (see relnet_fpn_roi_dispatch_ex) and get all-zero probabilities / boxes, i.e. they can never become detections.  */
int relnet_detect_head_ex(const float* cls_score, long cs_ld, const float* bbox_pred, long bp_ld, const float* rois,
                          const float* im_info, float* cls_prob, double* boxes, int R, int C, int rois_per_image,
                          int delta_off, const int* n_valid, void* stream);
/* lib/nms/nms.py:85-141 soft_nms (soft != 0, nms_param = sigma) or :45-82 nms (nms_param = IoU
 * threshold), float64 like numpy; dets [B,C-1,N,5] in pick order, counts [B,C-1].                    */
int relnet_class_nms(const float* cls_prob, const double* boxes, double* dets, int* counts, int B, int N, int C,
                     float score_thresh, double nms_param, int soft, int max_picks, void* stream);
/* same, for the `py_nms_wrapper` / `py_softnms_wrapper` call form of lib/nms/nms.py:21-31 (one class, float64
 * `dets`): scores64 [B,N] replaces cls_prob when non-null (then C == 2); pick_index [B,C-1,N] (nullable) receives
 * the roi index of every pick = the `keep` list of nms.py:45-82.                                           */
int relnet_class_nms_ex(const float* cls_prob, const double* scores64, const double* boxes, double* dets, int* counts,
                        int* pick_index, int B, int N, int C, float score_thresh, double nms_param, int soft,
                        int max_picks, void* stream);
/* relnet_class_nms with image-level pruning (tester.py:270-277): a class list stops as soon as its next pick cannot be among the
 * top_k scores of its image.  Every pick of the image is counted in `hist` (B x relnet_class_nms_hist_bins() unsigned ints, ZEROED
 * by the caller before every call); a class stops when its latest pick falls below the bin in which the count from the top reaches
 * top_k.  The lists are prefixes of relnet_class_nms's and contain every pick >= the final image threshold, so relnet_image_topk
 * returns the same detections; counts = picks produced.  N <= 1024. */
int relnet_class_nms_hist_bins(void);
int relnet_class_nms_topk(const float* cls_prob, const double* boxes, double* dets, int* counts, void* hist, int B, int N, int C,
                          float score_thresh, double nms_param, int soft, int max_picks, int top_k, void* stream);
/* tester.py:270-277: image threshold = max_per_image-th largest score; out [B,max_out,6] =
 * (class, score, x1, y1, x2, y2), class-major in pick order.                                        */
int relnet_image_topk(const double* dets, const int* counts, double* thresh, int* total, float* out,
                      int* out_count, int B, int NC, int N, int max_per_image, int max_out, void* stream);

/* ---- relation_rcnn/operator_py/learn_nms.py:238-401 (LearnNmsOperator.forward), device stages ------
 * prepare: softmax over classes (background dropped) + refine_bbox_nd (:175-217, float32) + clip.
 * means4 / stds4: HOST arrays of 4 floats or NULL (the test graph passes None, :420-421).           */
int relnet_lnms_prepare(const float* cls_score, long cs_ld, const float* bbox_pred, long bp_ld, const float* rois,
                        const float* im_info, float* prob /*[B,N,C-1]*/, float* boxes /*[B,N,4]*/, int B, int N,
                        int C, int delta_off, const float* means4, const float* stds4, void* stream);
/* ..._ex: n_valid [B] (device, may be NULL): padding rows get probability 0 (they sort behind every real roi).       */
int relnet_lnms_prepare_ex(const float* cls_score, long cs_ld, const float* bbox_pred, long bp_ld, const float* rois,
                           const float* im_info, float* prob, float* boxes, int B, int N, int C, int delta_off,
                           const float* means4, const float* stds4, const int* n_valid, void* stream);
/* :289-308: per (image, class) descending sort, first_n ranks; rank_idx [B,NC,F], sorted_score [B,F,NC],
 * sorted_bbox [B,F,NC,4], class_boxes [B,NC,F,4], class_max [B,NC].  Ties: smaller roi index first.  */
int relnet_lnms_sort(const float* prob, const float* boxes, int* rank_idx, float* sorted_score, float* sorted_bbox,
                     float* class_boxes, float* class_max, int B, int N, int NC, int first_n, void* stream);
/* :335-344: x[b,c,r,:] = roi_feat_embedding[b, rank_idx[b,c,r], :] + rank_feat[r, :]                    */
int relnet_lnms_embed(const void* roi_emb, const float* rank_feat, const int* rank_idx, void* x, int B, int N,
                      int NC, int first_n, int D, int dtype, void* stream);
/* :349-381 + symbols/..._learn_nms.py:553-560 + core/tester.py:231-242: relu(x + attention), logit FC,
 * sigmoid, x sorted_score (0 for classes failing the valid-class rule :293-302), merge over the T
 * thresholds (merge = -1 mean, -2 max, k >= 0 slice), compaction of score > score_thresh into
 * dets [B,NC,F,5] float64 (boxes / im scale) + counts [B,NC] (NULL to skip).                          */
int relnet_lnms_score(const void* x, const void* att, const float* w_logit, const float* b_logit,
                      const float* sorted_score, const float* sorted_bbox, const float* class_max,
                      const float* im_info, float* multi, float* final_score, double* dets, int* counts, int B,
                      int NC, int first_n, int D, int T, int H, int dv, int att_hstride, int merge,
                      float class_thresh, float score_thresh, int dtype, void* stream);

/* ---- training-target operators (A10), no host round trip ---------------------------------------
 * operator_py/proposal_target.py:44-93 with BATCH_ROIS = -1 -> core/rcnn.py:288-325 (sample_rois_v2),
 * lib/bbox/bbox.pyx:33-55 (float64 IoU), bbox_transform.py:74-100, bbox_regression.py:120-140.
 * rois [B,N,5], gt [B,Gmax,5], num_gt [B]; outputs have N+Gmax rows (rows past N+num_gt[b]: label -1).
 * means4/stds4/weights4: HOST double[4].                                                           */
int relnet_proposal_target(const float* rois, const float* gt, const int* num_gt, float* rois_out, float* label,
                           float* bbox_target, float* bbox_weight, int B, int N, int Gmax, int num_reg,
                           int class_agnostic, float bg_thresh_hi, const double* means4, const double* stds4,
                           const double* weights4, void* stream);
/* ..._ex: num_rois [B] (device, may be NULL): only the first num_rois[b] of the N input rows of image b are proposals
 * (core/rcnn.py:128-146 hands the reference a different roi count per image; a batched step pads to a common N): the
 * padded rows come out like the rows past num_gt -- zero box, label -1, zero weights -- so no loss ever sees them. */
int relnet_proposal_target_ex(const float* rois, const float* gt, const int* num_gt, float* rois_out, float* label,
                              float* bbox_target, float* bbox_weight, int B, int N, int Gmax, int num_reg,
                              int class_agnostic, float bg_thresh_hi, const double* means4, const double* stds4,
                              const double* weights4, const int* num_rois, void* stream);
/* lib/rpn/rpn.py:80-244 `assign_anchor(feat_shape, gt_boxes, im_info, cfg, feat_stride, scales, ratios, allowed_border)`
 * -- RPN labels / regression targets, host numpy inside the reference's data loader -- for B images on the device.
 * gt [B,Gmax,5] float32, num_gt [B], im_info [B,3]; base_anchors: HOST double [A,4] (generate_anchors); outputs in the
 * reference's layouts: label [B, A*fh*fw] ((a, y, x) order, -1 / 0 / 1), bbox_target and bbox_weight [B, 4A, fh, fw];
 * label_all (may be NULL) = the labels before sub-sampling; workspace: B*Gmax 64-bit words.  float64 overlaps and
 * bbox_transform as numpy.  The random sub-sampling (npr.choice, :189-204) keeps the anchors with the LARGEST keys,
 * key = 32-bit hash of (seed, image, anchor index in (y, x, a) order): a uniformly random subset for every seed,
 * reproducible and independent of the launch geometry (oracle/anchors.py restates the hash).  seed_dev (may be NULL):
 * a device word added to `seed` -- a step counter that a captured hipGraph advances, so replays draw fresh subsets.   */
int relnet_assign_anchor(const float* gt, const int* num_gt, const float* im_info, const double* base_anchors,
                         float* label, float* bbox_target, float* bbox_weight, float* label_all,
                         unsigned long long* workspace, int B, int A, int feat_h, int feat_w, int Gmax, int feat_stride,
                         int rpn_batch_size, int num_fg, double negative_overlap, double positive_overlap,
                         int clobber_positives, int allowed_border, unsigned long long seed,
                         const unsigned long long* seed_dev, void* stream);
/* operator_py/box_annotator_ohem.py:26-53: per-roi loss (-log softmax[label] + sum w*smooth_l1), keep the
 * roi_per_img largest; others get label -1 / zero weights.  R <= 2048.  loss [B,R] optional.           */
int relnet_box_annotator_ohem(const float* cls_score, const float* bbox_pred, const float* labels,
                              const float* bbox_targets, const float* bbox_weights, float* labels_ohem,
                              float* weights_ohem, float* loss, int B, int R, int C, int D, int roi_per_img,
                              void* stream);
/* lib/bbox/bbox.pyx:15-55 bbox_overlaps_cython: float64 IoU matrix with +1 extents, 0 when disjoint;
 * boxes [N,4], query_boxes [K,4], overlaps [N,K] -- DEVICE pointers (the Python twin relnet_amd/bbox/bbox.py
 * does the host copies when handed numpy arrays, as the Cython module's callers do).                      */
int relnet_bbox_overlaps(const double* boxes, const double* query_boxes, double* overlaps, int N, int K, void* stream);
/* operator_py/nms_multi_target.py:24-74: per class and IoU threshold, the highest-scoring box among those
 * whose arg-max gt is g and IoU > t gets target 1.  thresh: HOST double[T], T <= 8; first_n <= 256.      */
int relnet_nms_multi_target(const float* bbox, const float* gt, const int* num_gt, const float* score, float* out,
                            int B, int F, int C, int Gmax, const double* thresh, int T, void* stream);

/* ---- DCN configuration -------------------------------------------------
 * relation_rcnn/operator_cxx/deformable_convolution-inl.h:91-143 (DeformableConvolutionOp::Forward) =
 * relnet_deformable_im2col (nn/deformable_im2col.cuh:215-262, bilinear :76-113) followed by
 * relnet_gemm_nt on the column matrix.  data: logical [B,C,H,W] with element strides (NCHW fp32 or
 * channels-last bf16); offset: fp32 logical [B, 2*KH*KW*num_deformable_group, Ho, Wo] with element
 * strides; col: [B*Ho*Wo][col_ld] with column index (i*KW + j)*C + c  -- multiply by weights packed
 * [Cout][KH][KW][Cin].  Samples outside [0,H)x[0,W) contribute 0 (:247).                              */
int relnet_deformable_im2col(const void* data, const long* data_strides4, const float* offset,
                             const long* offset_strides4, void* col, long col_ld, int B, int C, int H, int W,
                             int KH, int KW, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                             int dil_w, int num_deformable_group, int data_dtype, int col_dtype, void* stream);
/* relation_rcnn/operator_cxx/deformable_psroi_pooling.cu:51-138 (DeformablePSROIPoolForwardKernel, called
 * from deformable_psroi_pooling-inl.h:64-95).  data logical [B, output_dim*group_size^2, H, W]; rois [R,5];
 * trans fp32 contiguous [R, 2*num_classes, part, part] or NULL (= no_trans); out / top_count logical
 * [R, output_dim, P, P] with element strides (top_count fp32, may be NULL).  part_size 0 = pooled_size. */
int relnet_deformable_psroi_pool_fwd(const void* data, const long* data_strides4, const float* rois,
                                     const float* trans, void* out, const long* out_strides4, float* top_count,
                                     int R, int C, int H, int W, int output_dim, int group_size, int pooled_size,
                                     int part_size, int sample_per_part, float spatial_scale, float trans_std,
                                     int num_classes, int batch_index_base, int dtype, void* stream);

/* ---- FPN configuration -------------------------------------------------
 * relation_rcnn/core/rcnn.py:53-74 (cfg.network.ROIDispatch, host numpy in the reference): pyramid level
 * feat_id = clip(floor(2 + log2(sqrt(w*h)/224)), 0, 3) per roi and the stable regrouping by level in which
 * symbols/resnet_v1_101_rcnn_fpn_..._learn_nms.py:1108-1121 concatenates rois_0..rois_3.  rois: [B,N,box_stride]
 * fp32 with x1,y1,x2,y2 at box_off.  Outputs: rois_out [B,N,5] (image index + base, box) level-sorted,
 * level_out [B,N], perm [B,N] (original row of each sorted row), counts [B,4].  This form does not append the
 * reference's all-zero dummy roi of an empty level (rcnn.py:61-71): counts tells the caller; ..._ex does.            */
int relnet_fpn_roi_dispatch(const float* rois, int box_stride, int box_off, float* rois_out, int* level_out,
                            int* perm, int* counts, int B, int N, int batch_index_base, void* stream);
/* The loader's full behaviour on a fixed-size row buffer: outputs have n_out rows per image (n_out >= N + 4 when
 * pad_empty).  Row order of an image: level 0 | level 1 | level 2 | level 3 | padding, where with pad_empty != 0 a level
 * that received no roi contributes ONE all-zero roi (perm = -1) exactly as rcnn.py:61-71 builds it (test AND train
 * loaders, :53-74 / :153-212).  n_valid [B] (may be NULL): only the first n_valid[b] input rows are rois; the others are
 * moved behind the real rows as zero boxes (level 0, perm = their input row).  n_rows [B] (may be NULL) = real rows of
 * the image (rois + dummies): feed it to relnet_relation_attention_kc (key_count) and relnet_detect_head_ex (n_valid). */
int relnet_fpn_roi_dispatch_ex(const float* rois, int box_stride, int box_off, float* rois_out, int* level_out,
                               int* perm, int* counts, int B, int N, int batch_index_base, const int* n_valid,
                               int pad_empty, int n_out, int* n_rows, void* stream);

/* The four mx.symbol.ROIPooling calls at 1/4, 1/8, 1/16, 1/32 + Concat(dim=0) (symbols/...fpn...:1108-1119)
 * as ONE launch: roi r pools from level roi_level[r].  data_levels / heights / widths / spatial_scales are
 * HOST arrays of num_levels entries (device pointers inside data_levels); data_strides4_levels is
 * [num_levels][4] element strides (b, c, y, x); all levels share C.                  */
int relnet_roi_pool_fpn_fwd(const void* const* data_levels, const long* data_strides4_levels, const int* heights,
                            const int* widths, const float* spatial_scales, int num_levels, const float* rois,
                            const int* roi_level, void* out, const long* out_strides4, int* argmax /*or NULL*/,
                            int R, int C, int PH, int PW, int batch_index_base, int dtype, void* stream);

/* mx.symbol.UpSampling(scale=2, sample_type='nearest') + mx.sym.ElementWiseSum of the FPN top-down pathway
 * (symbols/...fpn...:817-829): lateral[b,y,x,c] += top[b,y/2,x/2,c], NHWC contiguous, in place.             */
int relnet_upsample2x_add(const void* top, void* lateral, int B, int H, int W, int C, int dtype, void* stream);

/* ---- Training losses ---------------------------------------------------
 * mx.sym.SoftmaxOutput(normalization='valid', use_ignore, ignore_label=-1[, multi_output], grad_scale):
 * rpn_cls_prob / cls_prob (symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16_learn_nms.py
 * :272-273, :372-373, :379).  data / prob / grad: logical [outer, C, inner] (softmax over C; inner = 1 for the
 * [rois, classes] head, inner = A*H*W for the RPN's Reshape(0,2,-1,0)); label [outer, inner] float class ids.
 * grad (may be NULL) = (prob - onehot) * grad_scale / max(#labels != ignore, 1), 0 where ignored.
 * valid_count_scratch: one device int.                                                                    */
int relnet_softmax_output(const float* data, const float* label, float* prob, float* grad, int* valid_count_scratch,
                          long outer, int C, long inner, int use_ignore, float ignore_label, float grad_scale,
                          void* stream);
/* ..._ex: 'valid' normalisation per group of `group_positions` consecutive (outer, inner) positions (0 = all of them): the
 * reference normalises per executor = per image; a batched call passes one image's positions (valid_count_scratch then
 * holds one int per group).                                                                                          */
int relnet_softmax_output_ex(const float* data, const float* label, float* prob, float* grad, int* valid_count_scratch,
                             long outer, int C, long inner, int use_ignore, float ignore_label, float grad_scale,
                             long group_positions, void* stream);

/* weight * mx.sym.smooth_l1(scalar=sigma, data=pred - target) inside mx.sym.MakeLoss(grad_scale) (:276-278,
 * :374-377): loss (may be NULL) = w * f(pred - target), grad (may be NULL) = grad_scale * w * f'(pred - target);
 * f(x) = 0.5 (sigma x)^2 if |x| < 1/sigma^2 else |x| - 0.5/sigma^2.  weight NULL = 1.                        */
int relnet_smooth_l1_loss(const float* pred, const float* target, const float* weight, float* loss, float* grad,
                          long n, float sigma, float grad_scale, void* stream);

/* nms_pos_loss / nms_neg_loss (:536-551): pos = -k t log(s + eps), neg = -k (1 - t) log(1 - s + eps),
 * k = nms_loss_scale / (first_n * num_thresh); grad = d(pos_scale * pos + neg) / ds.                         */
int relnet_nms_loss(const float* score, const float* target, float* pos_loss, float* neg_loss, float* grad, long n,
                    float eps, float loss_scale_over_normalizer, float pos_scale, void* stream);

/* ---- Backward of the relation module (training; adjoint of relnet_geometry_bias / relnet_relation_attention,
 * i.e. of symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py:85-151 -- MXNet derives it
 * by autograd, the reference has no backward source).  All gradients fp32.
 * relnet_transpose_2d: out[c][r] = in[r][c] per batch item (operand layouts below).                          */
int relnet_transpose_2d(const void* in, long in_ld, long in_bs, void* out, long out_ld, long out_bs, int rows,
                        int cols, int batch, int dtype, void* stream);

/* The data-gradient layouts of ALL weights of a training step in ONE launch (MXNet's Convolution / FullyConnected backward
 * transposes its filter inside the cuDNN / GEMM call; here the copies are refreshed once per step, after the SGD update):
 *   dst[ci][(taps - 1 - tap) * dst_co + co] = src[co][tap * cin + ci]     bf16; taps = 1: W^T, taps = 9: tap-flipped 3x3 filter.
 * table: DEVICE array of n descriptors; tile_start = exclusive prefix sum of taps * tiles_co * tiles_ci (64 x 64 tiles);
 * total_tiles = the sum.  Pad columns co in [cout, dst_co) are never written (zero them once).                            */
typedef struct relnet_relayout_desc {
  const void* src; void* dst;
  int cout, cin, taps, dst_ld;          /* dst_ld = taps * dst_co                                                        */
  int dst_co, tiles_co, tiles_ci, tile_start;
} relnet_relayout_desc;
int relnet_weight_relayout(const void* table, int n, int total_tiles, void* stream);

/* The MFMA-fragment-order copies of the TRAINED weights that relnet_bottleneck_chain reads, all layers in one launch per step
 * (the training-time twin of relnet_pack_w_frag, which inference runs once at load time; SYM_BASE res3..res5 branch2c / branch2a).
 * mode 0: relnet_pack_w_frag order (W3 of the expand product); mode 1: the accumulator-permuted order of the next unit's reduce weights
 * (lane l, slot t <- W[32 nb + (l & 31)][16 kb + 8 (t >> 2) + 4 (l >> 5) + (t & 3)]).  N % 32 == 0, K % 16 == 0, ldw % 8 == 0.
 * table: DEVICE array; block_start = exclusive prefix of ceil((N / 32) (K / 16) 64 / 256); total_blocks = the sum.                 */
typedef struct relnet_fragpack_desc {
  const void* src; void* dst;
  long ldw;
  int N, K, mode, block_start;
} relnet_fragpack_desc;
int relnet_weight_fragpack(const void* table, int n, int total_blocks, void* stream);

/* C = (A W^T + resid) where mask > 0, else 0 (bf16 in / out, batch 1): the data gradient through `relu(conv1x1(.) + shortcut)` of a
 * residual unit (autograd of resnet_v1_101_rcnn_base.py res*_relu after broadcast_add) in one launch.  resid (may be NULL) and mask
 * have C's layout (row stride ldc); K % 64 == 0, N % 8 == 0, 16-byte aligned rows.                                                  */
int relnet_gemm_nt_mask(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const void* resid, const void* mask,
                        int M, int N, int K, void* stream);

/* C (fp32) = A W^T with IEEE-half (fp16) operands, v_mfma_f32_32x32x16_f16 on the LDS-tiled kernel (tile 2 = 256 x 128, 3 = 128 x 128): the fp16
 * twin of relnet_gemm_nt(bf16 in, fp32 out).  BASELINE configs[4] words its FPN run as "fp16 MFMA stress"; the reference itself is fp32
 * (experiments/relation_rcnn/cfgs/..._fpn_relation_learn_nms_8epoch.yaml) -- this entry is what measures that fp16 and bf16 operands run
 * at the same rate on gfx950 (tools/fp16_rate.py, profiles/r05_notes/fp16_vs_bf16.txt).  K % 64 == 0, lda / ldw % 8 == 0.            */
int relnet_gemm_nt_f16(const void* A, long lda, const void* W, long ldw, float* C, long ldc, int M, int N, int K, int tile, void* stream);

/* Backward of the relation module's projections (autograd of SYM_REL:120-129,146-150): the fp32 gradients of the attention backward --
 * dq [B][N][d], dk and dvw [B][M][d], M <= N -- rounded to bf16 into ONE operand out [B][N][3 d] = (dQ | dK | dVW), key blocks zero for
 * rows >= M: dF = out . [Wq; Wk; Wout] and d[Wq; Wk; Wout] = out^T F are then one GEMM and one weight-gradient product.  d % 8 == 0.  */
int relnet_relation_bwd_pack(const float* dq, const float* dk, const float* dvw, void* out, int B, int N, int M, int d, void* stream);

/* Adjoint of the learn-NMS head's per-class sort + take (symbols/..._learn_nms.py:438-446):
 * d_prob[b][rank_idx[b][c][f]][c] += d_sorted[b][f][c]; d_prob [B][N][C] fp32 zeroed by the caller, rank_idx [B][C][F] int32
 * (negative = padding, skipped), d_sorted [B][F][C].                                                                               */
int relnet_lnms_scatter_bwd(const float* d_sorted, const int* rank_idx, float* d_prob, int B, int N, int C, int F, void* stream);

/* ---- element-wise chains of the learn-NMS head's train branch and of its adjoint as single kernels (csrc/lnms_train.hip, round 6) --------
 * The reference spells them as chains of small operators (symbols/resnet_v1_101_rcnn_learn_nms_1024_attention_1024_pairwise_position_multi_head_16.py):
 *   pad_params     nms_linear_out_1 [128,128] bf16 / bias [128] and nms_logit [T,128] bf16 / bias [T] into the zero-padded operands of the 64-wide
 *                  tiles (wout_pad [1024,128]: row 64 h + j <- row 8 h + j; wl_pad [64,128]); the pad rows are the caller's zeros and stay untouched
 *   residual_relu  :489-491  out [rows,128] = relu(x [rows,128] + att [rows,1024][:, 64 h + j], j < 8)   (bf16; the sum is rounded before the ReLU)
 *   cond_multi     :497-505  cond [B,F,C,T] = sigmoid(logit [(b C + c) F + f][t]) (row stride ld), multi = sorted_score [B,F,C] x cond
 *   cond_bwd       adjoint of cond_multi: d_sorted [B,F,C] = sum_t d_multi cond, d_logit bf16 [(b C + c) F + f][64] = d_multi score cond (1 - cond), 0 beyond T (T <= 8)
 *   take_bwd       :447-452  d_emb bf16 [B N,128] = for every roi the fp32 sum of the rows of d_x bf16 [B,C,F,128] whose rank_idx [B,C,F] names it (every row written;
 *                  negative ranks skipped; C <= 128)
 *   softmax_bwd    :430-433  prob = softmax(cls_score)[:, 1:]: d_cls[b][n][0] += -(1 - sum prob) inner, d_cls[b][n][1 + c] += prob_c (d_prob_c - inner),
 *                  inner = sum_c prob_c d_prob_c; d_cls rows ld_row apart, images ld_img apart (ACCUMULATES into the detector's own cls_score gradient) */
/* out[0] = scale * sum(x[0..n))  (mode 0)  or  the count of entries >= 0 (mode 1): the scalar metrics of a training step (the MakeLoss outputs the
 * reference's metric classes sum per image, core/metric.py; the OHEM keep count) -- a 4-byte memset + one launch of <= 64 workgroups each */
int relnet_reduce_scalar(const float* x, long n, float scale, int mode, float* out, void* stream);
int relnet_lnms_pad_params(const void* wo, const float* bo, const void* wl, const float* bl, void* wout_pad, float* bout_pad, void* wl_pad,
                           float* bl_pad, int T, void* stream);
int relnet_lnms_residual_relu(const void* att, const void* x, void* out, long rows, void* stream);
int relnet_lnms_cond_multi(const float* logit, long ld, const float* sorted_score, float* cond, float* multi, int B, int C, int F, int T, void* stream);
int relnet_lnms_cond_bwd(const float* d_multi, const float* cond, const float* sorted_score, float* d_sorted, void* d_logit, int B, int C, int F, int T,
                         void* stream);
int relnet_lnms_take_bwd(const void* d_x, const int* rank_idx, void* d_emb, int B, int N, int C, int F, void* stream);
int relnet_lnms_softmax_bwd(const float* prob, const float* d_prob, float* d_cls, long ld_row, long ld_img, int B, int N, int C, void* stream);
/* the geometry bias ln G of the learn-NMS head's class-batched relation module from ONE per-image table: out [B C][16][F][Fpad] fp32 with
 * out[b C + c][h][f1][f2] = img [B][16][N][Npad] at (rank_idx[b][c][f1], rank_idx[b][c][f2]) for f2 < F (pad columns unwritten); rank_idx [B][C][F] >= 0.
 * The class's boxes are the image's boxes re-ordered by its ranks (operator_py/learn_nms.py:291-308), so this equals relnet_geometry_bias on the
 * gathered boxes bit for bit at B N^2 instead of B C F^2 pair evaluations */
int relnet_lnms_gather_bias(const float* img, const int* rank_idx, float* out, int B, int C, int N, int Npad, int F, int Fpad, void* stream);

/* q [B][N][..], k [B][M][..] as in the forward; kt = K^T [B][H*64][>=Mpad] and qt = Q^T, dyt = dY^T
 * [B][H*64][>=Npad] zero padded; vw = F_K Wout^T [B][M][H*64] (not transposed); bias = fp32 log G of the forward;
 * dy / y = gradient / value of the module output [B][N][H*64] (y includes bout).  Writes prob (softmax) and dlog
 * (d loss / d logits) [B][H][N][Mpad], dq [B][N][H*64], dk and dvw [B][M][H*64].
 * prob == NULL (bf16 operands, N and Mpad <= 128, 16-byte aligned rows): the one-workgroup-per-(image, head) form -- the rows of K, VW, Q, dY
 * staged in LDS (their transposes are read from there: kt / qt / dyt are not used and may be NULL), S and dL handed from the query tiles to the
 * key tiles through LDS, no S map in HBM; same results bit for bit.  With
 * dk == dvw == NULL as well, dq is a bf16 [B][N][3 H 64] buffer that receives (dQ | dK | dVW) -- the operand relnet_relation_bwd_pack
 * would build -- and whose key blocks must already be zero for rows >= M.                                                       */
int relnet_relation_attention_bwd(const void* q, long q_ld, long q_bs, const void* k, long k_ld, long k_bs,
                                  const void* kt, long kt_ld, long kt_bs, const void* vw, long vw_ld, long vw_bs,
                                  const float* bias, long bias_bs, const void* dy, long dy_ld, long dy_bs,
                                  const void* y, long y_ld, long y_bs, const float* bout, const void* qt, long qt_ld,
                                  long qt_bs, const void* dyt, long dyt_ld, long dyt_bs, float* prob, float* dlog,
                                  float* dq, float* dk, float* dvw, int B, int H, int N, int M, int Mpad, int Npad,
                                  float scale, int dtype, void* stream);
/* ..._kc: key_count as in relnet_relation_attention_kc (masked keys get zero probability and zero gradients).       */
int relnet_relation_attention_bwd_kc(const void* q, long q_ld, long q_bs, const void* k, long k_ld, long k_bs,
                                     const void* kt, long kt_ld, long kt_bs, const void* vw, long vw_ld, long vw_bs,
                                     const float* bias, long bias_bs, const void* dy, long dy_ld, long dy_bs,
                                     const void* y, long y_ld, long y_bs, const float* bout, const void* qt, long qt_ld,
                                     long qt_bs, const void* dyt, long dyt_ld, long dyt_bs, float* prob, float* dlog,
                                     float* dq, float* dk, float* dvw, int B, int H, int N, int M, int Mpad, int Npad,
                                     float scale, int dtype, const int* key_count, void* stream);

/* Weight gradients of convolution / FullyConnected layers (the adjoint MXNet's autograd derives; no reference source), n
 * layers per launch:          dw_i [Cout][Ktot] fp32 (row pitch dw_ld)  +=  row_scale_i[m]^2 * sum_p dy_i[p][m] * X_i[p][k]
 * dy [P][dy_cols] bf16 (pitch dy_ld; columns >= Cout are zero padding) and the activation x (bf16, channels contiguous,
 * element stride x_pix between pixels) are read as they lie in memory: LDS-transposed MFMA fragments (ds_read_b64_tr_b16),
 * no transposed copies; ks = 1 / stride = 1: X[p] = pixel (row) p, Ktot = Cin; otherwise x is [B][Hin][Win] pixels and
 * column k = tap (k / Cin) of channel k % Cin in pack_conv_weight order, gathered on the fly (zero outside the image).
 * The (layer, 256 x 256 tile, 64-pixel slab) units of ALL n layers are dealt in equal contiguous shares to one persistent
 * workgroup per CU (stream-K): tiles cut by a share boundary are combined with hardware float atomics, so dw must be
 * initialised (zero or a running sum) and the summation order is not fixed.  row_scale (may be NULL): the folded BatchNorm
 * factor per output row.  descs: HOST array; table_workspace: relnet_wgrad_workspace_bytes(n) bytes of device memory the
 * launch owns until it has run (filled by small kernels on the stream: no host copy, capture safe).                     */
typedef struct relnet_wgrad_desc {
  const void* dy; long dy_ld; int dy_cols;
  const void* x; long x_pix;
  float* dw; long dw_ld;
  const float* row_scale;
  int P, Cout, Cin, ks, stride, dil, pad, B, Hout, Wout, Hin, Win;
} relnet_wgrad_desc;
long relnet_wgrad_workspace_bytes(int n);
int relnet_wgrad_grouped(const relnet_wgrad_desc* descs, int n, void* table_workspace, void* stream);
/* Test / measurement aids: relnet_wgrad_debug_plain(1) assembles the MFMA fragments with scalar LDS reads;
 * relnet_wgrad_tune(workgroups (0 = one per CU), ablation bits (1 no flush, 4 no loads), wave rows (0 = by shape));
 * relnet_debug_tr_probe dumps what ds_read_b64_tr_b16 returns for lane-linear addresses (256 values).                 */
void relnet_wgrad_debug_plain(int on);
void relnet_wgrad_tune(int workgroups, int mode, int wave_rows);
/* share rule of relnet_wgrad_grouped: 0 / 1 = stream-K shares + float atomics (default), 2 = every workgroup owns WHOLE output tiles and flushes them by plain
 * read-modify-write (bit-identical reruns; needs the layers of the group to accumulate into disjoint memory, which the entry point checks -- otherwise the atomics
 * stay).  Round 6, measured slower at every batch size (one image: 7.16 -> 8.14 ms per training step): opt-in */
void relnet_wgrad_debug_tiles(int mode);
int relnet_debug_tr_probe(unsigned short* out256, void* stream);

/* d pair_pos_fc1_{weight [16][64], bias [16]} += from dlog and the forward's fp32 bias (= log max(G,1e-6)):
 * dpre = dlog / G where G > 1e-6; the 64-d embedding is recomputed from the boxes (SYM_REL:29-83).            */
int relnet_geometry_bias_bwd(const float* boxes, int box_stride, int box_off, const float* bias, const float* dlog,
                             const float* divisors8, float* dwp, float* dbp, int B, int N, int M, int Mpad,
                             int fast_math /* 1: hardware sin / cos / exp (bf16 training path) */, void* stream);

/* ---- Elementwise pieces of the training step ------------------------------
 * Gradient through Activation(relu) and the fused conv(+residual)+ReLU epilogues: dx = dy * (y > 0) (+ add).  */
int relnet_relu_bwd(const void* dy, const void* y, const void* add /*or NULL*/, void* dx, long n, int dtype,
                    void* stream);

/* Adjoint of the input sampling of a stride-s 1x1 convolution (symbols/resnet_v1_101_rcnn_base.py: res3a / res4a branch1 + branch2a, stride (2,2)):
 * out [B,H,W,C] = low [B,Ho,Wo,C] at the pixels (s i, s j), zero elsewhere; mask (or NULL) [B,H,W,C] = the saved ReLU output that IS that map:
 * the result is multiplied by (mask > 0) in the same pass.  Ho = (H - 1) / s + 1, Wo likewise; C a multiple of 8 (bf16) / 4 (fp32).            */
int relnet_strided_scatter(const void* low, const void* mask /*or NULL*/, void* out, int B, int H, int W, int C, int Ho, int Wo, int stride,
                           int dtype, void* stream);

/* out[c] += sum_r x[r, c]: the gradient of a bias (`FullyConnected` / `Convolution` bias, mx autograd) from the upstream gradient
 * [rows, cols] (dtype: 0 fp32, 1 bf16; row stride ld elements), accumulated in fp32 (atomic per column and row chunk).               */
int relnet_colsum_add(const void* x, long ld, long rows, int cols, int dtype, float* out, void* stream);
/* the same for up to 16 bf16 matrices in ONE launch (round 6: the 12 - 13 bias gradients of a training step's heads bucket): xs[i] [rows[i]][cols[i]]
 * with row pitch lds[i] elements (8 | cols, 8 | ld, 16-byte aligned rows), outs[i] fp32 [cols[i]] += column sums.  The arrays are HOST arrays. */
int relnet_colsum_add_grouped(const void* const* xs, const long* lds, const long* rows, const int* cols, float* const* outs, int n, void* stream);
/* mx.optimizer.SGD as set up in relation_rcnn/train_end2end.py:163-168 (momentum, wd, rescale_grad 1.0, no
 * clipping): mom = momentum*mom - lr*(rescale*grad + wd*w); w += mom.  fp32 master weights; w_bf16 (may be
 * NULL) receives the rounded copy the MFMA kernels read.                                                    */
int relnet_sgd_update(float* w, float* mom, const float* grad, void* w_bf16, long n, float lr, float momentum,
                      float wd, float rescale_grad, void* stream);

/* ---- DCN backward (DeformableConvolutionOp::Backward, deformable_convolution-inl.h:145-237) ---------
 * The column gradient dcol = dY W comes from relnet_gemm_nt; this entry is deformable_col2im
 * (nn/deformable_im2col.cuh:313-351) + deformable_col2im_coord (:420-470) in one pass: grad_data (fp32, logical
 * [B,C,H,W], element strides) and grad_offset (fp32, logical [B,2*KH*KW*DG,Ho,Wo], may be NULL) are ACCUMULATED
 * into (atomics) -- zero them first.  dcol: [B*Ho*Wo][dcol_ld], column (i*KW + j)*C + c, fp32 or bf16.            */
/* tuning / test knob of relnet_deformable_col2im: 0 = auto (round 6: on stride-1 channels-last bf16 layers the data gradient is GATHERED per feature cell
 * from the (pixel, tap) pairs whose bilinear corners lie within 3 cells of their undeformed tap position, float atomics only for the others), 1 = the
 * one-kernel atomic scatter everywhere, 2 = every pair treated as far (offset kernel + far-only scatter pass), 10 + D = window radius D */
void relnet_deformable_col2im_debug(int mode);
int relnet_deformable_col2im(const void* dcol, long dcol_ld, int dcol_dtype, const void* data,
                             const long* data_strides4, int data_dtype, const float* offset,
                             const long* offset_strides4, float* grad_data, const long* grad_data_strides4,
                             float* grad_offset, const long* grad_offset_strides4, int B, int C, int H, int W, int KH,
                             int KW, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                             int num_deformable_group, void* stream);

/* DeformablePSROIPoolBackwardAccKernel (deformable_psroi_pooling.cu:178-285): grad_data (fp32 logical [B,C,H,W]) and
 * grad_trans (fp32 [R,2*num_classes,part,part], NULL iff trans is NULL) are accumulated into; top_count is
 * recomputed from the geometry.                                                                              */
int relnet_deformable_psroi_pool_bwd(const void* grad_out, const long* grad_out_strides4, const void* data,
                                     const long* data_strides4, const float* rois, const float* trans,
                                     float* grad_data, const long* grad_data_strides4, float* grad_trans, int R, int C,
                                     int H, int W, int output_dim, int group_size, int pooled_size, int part_size,
                                     int sample_per_part, float spatial_scale, float trans_std, int num_classes,
                                     int batch_index_base, int dtype, void* stream);
/* test / measurement knob of the entry above: 0 = auto (round 6: four channels per thread where the roi geometry / offsets / per-axis cell sums of a bin do not
 * depend on the channel: group_size 1, one offset class or none, sample_per_part <= 4, 256 | output_dim), 1 = one channel per thread everywhere */
void relnet_deformable_psroi_pool_bwd_debug(int mode);

/* Adjoint of relnet_roi_pool_fpn_fwd (argmax from the forward, same strides as grad_out): roi r scatters into
 * grad_in_levels[roi_level[r]] (fp32 [B,C,H_l*W_l] with batch / channel strides gs_b / gs_c; host arrays).     */
int relnet_roi_pool_fpn_bwd(const void* grad_out, const int* argmax, const long* out_strides4, const float* rois,
                            const int* roi_level, float* const* grad_in_levels, const long* gs_b_levels,
                            const long* gs_c_levels, int num_levels, int R, int C, int PH, int PW,
                            int batch_index_base, int dtype, void* stream);
/* ..._ex of both backward entries: gs_p = element stride between PIXELS of the gradient buffer (the plain entries use 1 =
 * [B,C,H*W]).  With gs_c = 1, gs_p = C the buffer is [B,H,W,C]: a wavefront's 64 consecutive channels of one bin hit 64
 * consecutive words, i.e. coalesced hardware float atomics, and the result is already in the layout the NHWC convolution
 * backward consumes.                                                                                                  */
int relnet_roi_pool_bwd_ex(const void* grad_out, const int* argmax, const long* out_strides4, const float* rois,
                           float* grad_in, long gs_b, long gs_c, long gs_p, int R, int C, int W, int PH, int PW,
                           int batch_index_base, int dtype, void* stream);
/* The same adjoint for a channels-last gradient: grad_in fp32 [B][H][W][C] (dense, accumulated into).  Round 6: where the operands allow (channels contiguous
 * in grad_out / argmax, 8 | C, H W <= 4608 cells) and the step is large enough to give every CU a workgroup (B C / 8 >= 256) ONE workgroup owns (image, 8
 * channels): the H x W slab is accumulated in LDS over every (roi, bin) of the image and flushed once -- no global atomics (31 M contended float atomics =
 * 0.44 ms per 8-image training step before); otherwise the scatter kernel.  relnet_roi_pool_bwd_debug(1) forces the scatter kernel;
 * 4 + bits = timing ablations of the owner kernel (results WRONG): 1 = no LDS adds, 2 = no scattered loads. */
int relnet_roi_pool_bwd_cl(const void* grad_out, const int* argmax, const long* out_strides4, const float* rois, float* grad_in,
                           int B, int H, int W, int R, int C, int PH, int PW, int batch_index_base, int dtype, void* stream);
void relnet_roi_pool_bwd_debug(int mode);
int relnet_roi_pool_fpn_bwd_ex(const void* grad_out, const int* argmax, const long* out_strides4, const float* rois,
                               const int* roi_level, float* const* grad_in_levels, const long* gs_b_levels,
                               const long* gs_c_levels, const long* gs_p_levels /* NULL = 1 */, int num_levels, int R, int C,
                               int PH, int PW, int batch_index_base, int dtype, void* stream);

/* grad[r,c] += row_scale[r]^2 * sum_s parts[s,r,c]: split-K partial sums of a weight gradient, the folded frozen-BN
 * factor (NULL = 1) and the accumulation into the flat gradient buffer in one pass.  cols % 4 == 0.             */
int relnet_wgrad_accumulate(const float* parts, int splits, long rows, int cols, const float* row_scale, float* grad,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RELNET_HIP_H */
