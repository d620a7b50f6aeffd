/*
 * ssdhip.h -- C ABI of libssdhip.so: the per-anchor hot path of an SSD detector
 * (target encoding, multibox loss, prediction decoding + NMS) as hand-written
 * gfx950 (MI355X / CDNA4) HIP kernels.
 *
 * The reference (pierluigiferrari/ssd_keras) is pure Python and has no FFI of its own;
 * each entry point below replaces the arithmetic of one of its Python callables, cited
 * as file:line relative to the reference root.  The Python package `ssd_keras_amd`
 * re-creates the reference's classes/functions on top of these calls (INTEGRATION.md
 * shows the ctypes binding a maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed or a torch CUDA tensor's data_ptr);
 *   - the caller allocates all outputs and the scratch workspace (`*_workspace_bytes`);
 *   - calls only ENQUEUE work on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and return at once: no allocation, no host sync, no global mutable state,
 *     so they are re-entrant for distinct workspaces/streams and graph-capturable;
 *   - tensors are dense row-major with the reference's layout: last axis of y_pred /
 *     y_true = [C class scores (index 0 = background) | 4 offsets | 4 anchor coords |
 *     4 variances];
 *   - return value: SSDHIP_OK or a negative SSDHIP_E_* code (`ssdhip_strerror`).
 */
#ifndef SSDHIP_H
#define SSDHIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDHIP_ABI_VERSION 1

enum {
    SSDHIP_OK = 0,
    SSDHIP_E_BADARG = -1,    /* inconsistent sizes / unsupported option combination */
    SSDHIP_E_WORKSPACE = -2, /* ws == NULL or ws_bytes too small                      */
    SSDHIP_E_LAUNCH = -3     /* hipGetLastError() after a launch was not hipSuccess   */
};

/* element types of tensors crossing the boundary */
enum { SSDHIP_F32 = 0, SSDHIP_F64 = 1 };
/* box coordinate formats (reference: 'centroids' (cx,cy,w,h), 'corners' (xmin,ymin,xmax,ymax),
 * 'minmax' (xmin,xmax,ymin,ymax)) */
enum { SSDHIP_CENTROIDS = 0, SSDHIP_CORNERS = 1, SSDHIP_MINMAX = 2 };
/* border_pixels: 'half' d=0, 'include' d=+1, 'exclude' d=-1.  As in the reference's iou()
 * (bounding_box_utils/bounding_box_utils.py:345) only the two box AREAS see d; the
 * intersection is always computed with d = 0. */
enum { SSDHIP_BORDER_HALF = 0, SSDHIP_BORDER_INCLUDE = 1, SSDHIP_BORDER_EXCLUDE = 2 };
/* which reference callable's arithmetic the decoder follows */
enum {
    SSDHIP_SEM_NUMPY = 0,  /* ssd_encoder_decoder/ssd_output_decoder.py decode_detections(_fast):
                              d*(var*a)+c decode, float64 IoU/compare for float32 input routed through
                              convert_coordinates (centroids, minmax), input-dtype arithmetic for 'corners';
                              unsorted top-k                                                              */
    SSDHIP_SEM_KERAS = 1,  /* keras_layers/keras_layer_DecodeDetections(Fast).py: (d*var)*a+c decode,
                              float32 strict '>' threshold, rows sorted by confidence, zero padded       */
    SSDHIP_SEM_DEBUG = 2   /* decode_detections_debug (:342-467): (d*a)*var+c decode, otherwise NUMPY   */
};

int ssdhip_abi_version(void);
const char* ssdhip_strerror(int rc);

/* ------------------------------------------------------------------------------------------
 * Decoder: box decode + confidence threshold + per-class (or class-agnostic) greedy NMS + top-k.
 * Replaces ssd_output_decoder.py: decode_detections :111-226, decode_detections_fast :228-333,
 * decode_detections_debug :342-467, _greedy_nms* :77-109, and the in-graph
 * keras_layer_DecodeDetections.py:109-265 / keras_layer_DecodeDetectionsFast.py:111-248.
 *
 *   y_pred        [B, N, C+12] of `in_dtype`.  SSDHIP_F32 (what the model emits): float32 decode, then the reference's
 *                 float64 flow from convert_coordinates on.  SSDHIP_F64: the reference's all-float64 flow
 *                 (ssd_output_decoder.py:172-198 computes in the input's dtype); not with SEM_KERAS (a float32 graph)
 *   C             number of classes INCLUDING background (class 0)
 *   conf_thresh   candidates need score > conf_thresh (class_agnostic + SEM_NUMPY: >=, and class != 0)
 *   iou_thresh    a box is dropped when IoU with an already kept box is NOT <= iou_thresh
 *   top_k         rows kept per image by confidence; <= 0: keep all NMS survivors
 *   nms_cap       max NMS survivors per class (tf.image.non_max_suppression max_output_size);
 *                 <= 0: the reference-NumPy behaviour (uncapped, internally stops at top_k survivors
 *                 per class when top_k > 0, which cannot change the top-k set)
 *   class_agnostic 0: one NMS per (image, class 1..C-1); 1: class = first argmax, one NMS per image
 *   out           [B, out_rows, 6] of `out_dtype`: class, conf, xmin, ymin, xmax, ymax; rows beyond
 *                 out_count[b] are zero.  Row order: SEM_KERAS -> confidence descending (ties: lower
 *                 class, then higher NMS rank first); otherwise class ascending / confidence descending
 *                 when nothing had to be cut, else unspecified (the reference uses np.argpartition).
 *   out_count     [B] valid rows per image
 *   out_anchor_idx [B, out_rows] anchor index of each row (-1 padding) or NULL
 */
size_t ssdhip_decode_workspace_bytes(int B, int N, int C, int top_k, int nms_cap, int class_agnostic, int in_dtype);
int ssdhip_decode_detections(const void* y_pred, int in_dtype, int B, int N, int C,
                             double conf_thresh, double iou_thresh, int top_k, int nms_cap,
                             int class_agnostic, int semantics,
                             int coords, int normalize_coords, double img_height, double img_width,
                             int border_pixels,
                             void* out, int out_dtype, int out_rows, int* out_count, int* out_anchor_idx,
                             void* ws, size_t ws_bytes, void* stream);
/* Profiling aid: run only the selected kernels of the decoder on the state left in `ws` by the previous ones.
 * stages: bit 0 = scan (decode + threshold + candidate lists), bit 1 = per-class NMS, bit 2 = top-k/output.
 * `ssdhip_decode_detections` == stages 7.  Used by bench.py to time each kernel with events on the stream. */
int ssdhip_decode_stages(int stages, const void* y_pred, int in_dtype, int B, int N, int C,
                         double conf_thresh, double iou_thresh, int top_k, int nms_cap,
                         int class_agnostic, int semantics,
                         int coords, int normalize_coords, double img_height, double img_width,
                         int border_pixels,
                         void* out, int out_dtype, int out_rows, int* out_count, int* out_anchor_idx,
                         void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Encoder: ground truth -> training targets.  Replaces SSDInputEncoder.__call__
 * (ssd_encoder_decoder/ssd_input_encoder.py:277-418) including iou() (bounding_box_utils.py:283-383),
 * match_bipartite_greedy / match_multi (matching_utils.py:22-116) and the template tiling
 * (generate_encoding_template :550-611).  All arithmetic in float64 with the reference's operation order.
 *
 *   anchors       [N,4] float64, in `coords` format, exactly the values of SSDInputEncoder.boxes_list
 *   variances     [4] float64
 *   gt            [G_total,5] float64 rows class,xmin,ymin,xmax,ymax in absolute pixels ('corners'),
 *                 images concatenated; gt_offsets [B+1] int32 CSR offsets (an image may have 0 rows)
 *   max_gt_per_image  largest row count of any image (the host built the CSR, so it knows); <= 1024
 *   matching_type 0 'bipartite', 1 'multi'
 *   y_encoded_f32 / y_encoded_f64  [B,N,C+12] outputs, either may be NULL
 *   match_gt      [B,N] int32: >=0 index (within the image) of the matched GT, -1 background,
 *                 -2 neutral (class vector all zero); may be NULL
 * Degenerate GT boxes (xmax<=xmin or ymax<=ymin) must be rejected by the caller (DegenerateBoxError).
 */
size_t ssdhip_encode_workspace_bytes(int B, int N, int C, int G_total);
int ssdhip_encode(const double* anchors, const double* variances, const double* gt, const int* gt_offsets,
                  int G_total, int max_gt_per_image, int B, int N, int C, double img_height, double img_width,
                  int matching_type, double pos_iou_threshold, double neg_iou_limit,
                  int coords, int normalize_coords, int border_pixels, int background_id,
                  float* y_encoded_f32, double* y_encoded_f64, int* match_gt,
                  void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss: SSDLoss.compute_loss (keras_loss_function/keras_ssd_loss.py:98-211; smooth_L1_loss :53-75,
 * log_loss :77-96), float32, hard-negative mining global over the batch it is given.
 *
 *   loss_per_item [B] float32
 *   stats         [4] float32: n_positive, n_neg_losses, k (negatives kept), k-th largest negative loss
 *   keep_mask     [B,N] uint8: 1 where the anchor is a kept hard negative (its classification loss
 *                 enters the sum besides the positives') -- saved for backward
 *   backward: grad_y_pred [B,N,C+12] = d(sum_b grad_out[b]*loss[b]) / d y_pred (last 8 columns zero)
 */
size_t ssdhip_loss_workspace_bytes(int B, int N, int C);
int ssdhip_loss_forward(const float* y_true, const float* y_pred, int B, int N, int C,
                        int neg_pos_ratio, int n_neg_min, float alpha,
                        float* loss_per_item, float* stats, unsigned char* keep_mask,
                        void* ws, size_t ws_bytes, void* stream);
int ssdhip_loss_backward(const float* y_true, const float* y_pred, const unsigned char* keep_mask,
                         const float* stats, const float* grad_out, int B, int N, int C, float alpha,
                         float* grad_y_pred, void* stream);

/* ------------------------------------------------------------------------------------------
 * Public box utilities (stand-alone; the encoder / decoder above fuse the same arithmetic into their kernels).
 * dtypes are SSDHIP_F32 / SSDHIP_F64 and follow NumPy's rules, because those decide where the reference rounds.
 */
/* conversion codes of convert_coordinates (bounding_box_utils/bounding_box_utils.py:24-87) */
enum {
    SSDHIP_MINMAX2CENTROIDS = 0, SSDHIP_CENTROIDS2MINMAX = 1, SSDHIP_CORNERS2CENTROIDS = 2, SSDHIP_CENTROIDS2CORNERS = 3,
    SSDHIP_SWAP_MINMAX_CORNERS = 4    /* 'minmax2corners' and 'corners2minmax' are the same permutation */
};
/* convert_coordinates(tensor, start_index, conversion, border_pixels) (:24-87; convert_coordinates2 :89-117 is the same
 * arithmetic for a float64 input): `in` is the tensor flattened to [n_rows, row_len] of in_dtype; `out` is its float64 copy
 * with columns start_index..start_index+3 converted.  Right-hand sides are evaluated in in_dtype (as NumPy does), then widened. */
int ssdhip_convert_coordinates(const void* in, int in_dtype, double* out, long long n_rows, int row_len,
                               int start_index, int conversion, int border_pixels, void* stream);
/* iou() :283-383 (op 0) and intersection_area() / intersection_area_() :119-280 (op 1).
 *   boxes1 [m,4] of dtype1, boxes2 [n,4] of dtype2, both in `coords` format
 *   mode 0 'outer_product' -> out [m,n];  mode 1 'element-wise' -> out [max(m,n)], m == n or one of them 1 (broadcast)
 *   out dtype = ssdhip_iou_result_dtype(dtype1, dtype2, coords): F32 only for two F32 inputs in 'corners'/'minmax'
 *   (the 'centroids' conversion yields float64), else F64.
 * op 0 keeps the reference's quirk (:345): the intersection ignores border_pixels, the two areas do not. */
int ssdhip_iou_result_dtype(int dtype1, int dtype2, int coords);
int ssdhip_box_overlap(int op, const void* boxes1, int dtype1, int m, const void* boxes2, int dtype2, int n,
                       int coords, int mode, int border_pixels, void* out, void* stream);
/* match_bipartite_greedy(weight_matrix) (ssd_encoder_decoder/matching_utils.py:22-79): weight_matrix [m,n] float64 (not
 * modified), matches [m] int32 = the column matched to each row.  m <= 4096.  Ties -> lowest row, then lowest column;
 * removed rows/columns count as zeros exactly like the reference's in-place zeroing. */
int ssdhip_match_bipartite_greedy(const double* weight_matrix, int m, int n, int* matches, void* stream);
/* match_multi(weight_matrix, threshold) (:81-116): per column the first arg-max row, kept where the value >= threshold.
 * gt_idx / anchor_idx [n] int32 receive the `*count` kept (row, column) pairs in ascending column order. */
size_t ssdhip_match_multi_workspace_bytes(int m, int n);
int ssdhip_match_multi(const double* weight_matrix, int m, int n, double threshold, int* gt_idx, int* anchor_idx,
                       int* count, void* ws, size_t ws_bytes, void* stream);
/* greedy_nms / _greedy_nms / _greedy_nms2 / _greedy_nms_debug (ssd_encoder_decoder/ssd_output_decoder.py:27-109, 469-486)
 * on a float64 table rows [n_rows_total, row_len]: segments (batch items) given by seg_offsets [n_segments+1]; the score
 * is column score_col, the box columns box_col..box_col+3 in `coords` format.  Per segment: repeatedly keep the first
 * maximum-score row and drop every row whose IoU with it is not <= iou_threshold.
 * kept_idx [n_rows_total]: for segment s, kept_idx[seg_offsets[s] + k] = row (within the segment) of the k-th kept box,
 * k < kept_count[s] (selection order = score descending). */
size_t ssdhip_greedy_nms_workspace_bytes(int n_rows_total);
int ssdhip_greedy_nms(const double* rows, int n_rows_total, int row_len, int score_col, int box_col,
                      const int* seg_offsets, int n_segments, double iou_threshold, int coords, int border_pixels,
                      int* kept_idx, int* kept_count, void* ws, size_t ws_bytes, void* stream);

/* BoxFilter.__call__ (data_generator/object_detection_2d_image_boxes_validation_utils.py:147-232) for a whole batch: boxes [G,4]
 * float64 'corners' of all images concatenated, box_image [G] int32 image index of each box, image_hw [n_images,2] float64
 * (height, width) of the image / patch each box is validated against.  overlap_criterion 0 'center_point', 1 'iou', 2 'area';
 * keep [G] uint8 = 1 where the box passes every enabled check (degenerate, min_area, overlap within (lower, upper]). */
int ssdhip_box_filter(const double* boxes, const int* box_image, const double* image_hw, int G, int n_images,
                      int check_overlap, int check_min_area, int check_degenerate, int overlap_criterion,
                      double lower, double upper, double min_area, int border_pixels, unsigned char* keep, void* stream);

/* ------------------------------------------------------------------------------------------
 * Evaluator.match_predictions, one class (eval_utils/average_precision_evaluator.py:604-725).
 *   pred        [P,5] float32 rows confidence, xmin, ymin, xmax, ymax (the reference keeps predictions as 'f4')
 *   pred_image  [P] int32 index of each prediction's image
 *   gt_boxes    [G,4] float64 'corners' boxes of THIS class, images concatenated; gt_offsets [n_images+1] int32 CSR
 *   gt_neutral  [G] uint8, non-zero = evaluation-neutral ('difficult') box, or NULL
 * Outputs, all [P] int32 and in the reference's order (confidence descending, equal confidences in input order, i.e.
 * np.argsort(-confidence, kind='mergesort')): `order` (index into pred), true_pos, false_pos and their cumulative sums.
 * A prediction whose best-IoU box (first maximum) is below the threshold, or whose image has no box of the class, is a false
 * positive; a neutral best match is neither; otherwise it is a true positive iff no earlier prediction claimed that box. */
size_t ssdhip_match_predictions_workspace_bytes(int P, int G);
int ssdhip_match_predictions(const float* pred, const int* pred_image, int P, const double* gt_boxes, const int* gt_offsets,
                             const unsigned char* gt_neutral, int n_images, int G, double matching_iou_threshold,
                             int border_pixels, int* order, int* true_pos, int* false_pos, int* cum_true_pos,
                             int* cum_false_pos, void* ws, size_t ws_bytes, void* stream);
/* The whole class loop of Evaluator.match_predictions (:601-725) in ONE call (round 6).  The predictions of all classes concatenated
 * slot by slot (slot = position in the caller's class list): pred_segment [P] = slot * n_images + image index, pred_class [P] = slot,
 * class_start [n_slots+1] (device) the slots' stretches; the ground truth as CSR over the n_slots * n_images segments (gt_offsets
 * [n_segments+1]).  Outputs [P] in the concatenated order, every slot's stretch sorted as above; the cumulative sums restart per slot.
 * n_slots <= 128, P < 2^25; workspace from ssdhip_match_predictions_workspace_bytes(P, G). */
int ssdhip_match_predictions_multi(const float* pred, const int* pred_segment, const int* pred_class, int P, const double* gt_boxes,
                                   const int* gt_offsets, const unsigned char* gt_neutral, int n_segments, int G, const int* class_start,
                                   int n_slots, double matching_iou_threshold, int border_pixels, int* order, int* true_pos,
                                   int* false_pos, int* cum_true_pos, int* cum_false_pos, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Graph glue between the convolutions (bf16 activations, NHWC = torch channels_last; C % 8 == 0; pointers
 * 16-byte aligned).  Each call is ONE pass over the tensor where the framework path runs 2-7 elementwise kernels.
 *
 * ssdhip_bias_act_nhwc_bf16        y = act(x + bias[c]) (x, y may alias): Conv2D(..., activation='relu')'s epilogue,
 *                                  models/keras_ssd300.py:274-313.  bias may be NULL; relu 0/1.
 * ssdhip_bias_act_maxpool_nhwc_bf16  the same followed by MaxPooling2D (kernel, stride, pad; window clipped to the map --
 *                                  Keras 'same' pooling == pad 0 + Ho = ceil(H/stride)); y [B,Ho,Wo,C].
 * ssdhip_l2_normalize_nhwc_bf16    keras_layers/keras_layer_L2Normalization.py:61-63: x * rsqrt(max(sum_c x^2, 1e-12)) * gamma[c],
 *                                  gamma float32 [C].
 * ssdhip_preprocess_nhwc_f32_to_bf16  the Lambda input layers, models/keras_ssd300.py:247-272: subtract mean, divide by
 *                                  stddev, reorder channels; mean_h / divide_h / swap_h are HOST arrays of `channels`
 *                                  entries (NULL = identity); out[...,c] = bf16((img[...,swap[c]] - mean[swap[c]]) / divide[swap[c]]).
 * ssdhip_assemble_predictions_bf16  Reshape + Concatenate + softmax + AnchorBoxes + Concatenate, models/keras_ssd300.py:363-419
 *                                  and keras_layers/keras_layer_AnchorBoxes.py:245-255: per predictor layer l the NHWC conv outputs
 *                                  conf_h[l] [B, n_anchors_h[l], C] and loc_h[l] [B, n_anchors_h[l], 4] (bf16 DEVICE pointers held
 *                                  in HOST arrays, as are the optional per-layer biases [n_boxes*C], [n_boxes*4]) become
 *                                  y_pred [B, N, C+12] float32 = [softmax(conf) | loc | anchors_var[N,8]]; N = sum n_anchors_h.
 */
int ssdhip_bias_act_nhwc_bf16(const void* x, const void* bias, void* y, long long n_pixels, int C, int relu, void* stream);
int ssdhip_bias_act_maxpool_nhwc_bf16(const void* x, const void* bias, void* y, int B, int H, int W, int C,
                                      int kernel, int stride, int pad, int Ho, int Wo, int relu, void* stream);
int ssdhip_l2_normalize_nhwc_bf16(const void* x, const float* gamma, void* y, long long n_pixels, int C, void* stream);
/* pool4 + conv4_3_norm in one pass (round 6): MaxPooling2D((2, 2), strides (2, 2), 'same') and L2Normalization of the same 512-channel map
 * (models/keras_ssd300.py:287, 316): x [B, H, W, 512] bf16 -> y_pool [B, ceil(H/2), ceil(W/2), 512], y_norm [B, H, W, 512]; bit-identical
 * to the two separate calls. */
int ssdhip_pool2_l2_normalize_nhwc_bf16(const void* x, const float* gamma, void* y_pool, void* y_norm, int B, int H, int W, int C, void* stream);
int ssdhip_preprocess_nhwc_f32_to_bf16(const float* images, void* out, long long n_pixels, int channels,
                                       const float* mean_h, const float* divide_h, const int* swap_h, void* stream);
int ssdhip_assemble_predictions_bf16(int n_layers, const void* const* conf_h, const void* const* loc_h,
                                     const void* const* conf_bias_h, const void* const* loc_bias_h,
                                     const int* n_anchors_h, const int* n_boxes_h, const float* anchors_var,
                                     int B, int N, int C, float* y_pred, void* stream);
/* ------------------------------------------------------------------------------------------
 * Training-step glue (csrc/ssdhip_train.hip): the backward of Conv2D(activation='relu') and MaxPooling2D
 * (models/keras_ssd300.py:274-313; TF graph ops in the reference, ssd300_training.ipynb), bf16 NHWC.
 *
 * ssdhip_relu_bwd_bias_nhwc_bf16   out = gy where y > 0 else 0 (threshold_backward) and per-workgroup float32 partial sums of
 *                                  `out` per channel: partial [n_blocks, C], n_blocks = ssdhip_relu_bwd_bias_blocks(n_pixels, C)
 *                                  (0: shape not supported -- C / 8 must divide 256); the bias gradient is their sum over axis 0.
 * ssdhip_maxpool_bwd_nhwc_bf16     gx [B,H,W,C] = gradient of max-pooling (kernel, stride, pad; windows clipped to the map) given its
 *                                  input x and the gradient gy [B,Ho,Wo,C] of its output: every window's gradient goes to its first
 *                                  maximum in row-major order (NaN wins), as max_pool2d's backward. */
/* L2Normalization (keras_layers/keras_layer_L2Normalization.py:61-63) for the float32 model and for the training step: x, y, dy, dx
 * [n_pixels, C] NHWC float32 (is_bf16 = 0, C % 4 == 0) or bf16 (is_bf16 = 1, C % 8 == 0), gamma float32 [C]; float32 math.
 * ssdhip_l2_normalize_fwd   y = x * rsqrt(max(sum_c x^2, 1e-12)) * gamma[c]; inv_norm [n_pixels] float32 (may be NULL) receives the
 *                           pixel's rsqrt(...) for the backward.
 * ssdhip_l2_normalize_bwd   dx = gamma * dy * inv - x * inv^3 * sum_c(dy gamma x) (the second term only where the norm was not clamped)
 *                           and dgamma_partial [n_waves, C]: per-wave sums of dy * x * inv over the wave's pixels, n_waves =
 *                           ssdhip_l2_normalize_bwd_waves(n_pixels, C, is_bf16) (0: shape not supported); dgamma is their sum over axis 0. */
int ssdhip_l2_normalize_bwd_waves(long long n_pixels, int C, int is_bf16);
int ssdhip_l2_normalize_fwd(const void* x, const float* gamma, void* y, float* inv_norm, long long n_pixels, int C, int is_bf16, void* stream);
int ssdhip_l2_normalize_bwd(const void* x, const void* dy, const float* gamma, const float* inv_norm, void* dx, float* dgamma_partial,
                            int n_waves, long long n_pixels, int C, int is_bf16, void* stream);
int ssdhip_relu_bwd_bias_blocks(long long n_pixels, int C);
/* Channel sums of a bf16 map alone (the bias gradient of a layer without activation: the predictor heads), as per-workgroup partial
 * sums [n_blocks][C] like the two passes around it. */
int ssdhip_channel_sums_nhwc_bf16(const void* gy, float* partial, long long n_pixels, int C, int n_blocks, void* stream);

/* Weight gradient of the 1x1 stride-1 layers (fc7, conv6_1 ... conv9_1: models/keras_ssd300.py:296-313 under model.fit_generator; round 5) as
 * a GEMM over the pixels: dw [Cout][Cin] float32 from x [n_pixels][Cin] and dy [n_pixels][Cout] bf16 (NHWC maps read as matrices).
 * Cin % 128 == 0 and Cout % 128 == 0 (ssdhip_conv1x1_wgrad_workspace_bytes returns 0 otherwise: the caller falls back to the
 * framework).  Fixed summation order (bit-reproducible); bias_partial / bias_rows / db as ssdhip_conv3x3_wgrad_bias_nhwc_bf16. */
size_t ssdhip_conv1x1_wgrad_workspace_bytes(long long n_pixels, int Cin, int Cout);
int ssdhip_conv1x1_wgrad_bias_nhwc_bf16(const void* x, const void* dy, float* dw, const float* bias_partial, int bias_rows, float* db,
                                        long long n_pixels, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* Weight gradient of the 3x3 layers of ANY stride / padding / dilation (round 6: fc6 with dilation 6, the stride-2 conv6_2 / conv7_2
 * behind ZeroPadding2D, the 'valid' conv8_2 / conv9_2: models/keras_ssd300.py:294, 299-313 under model.fit_generator; replaces the
 * framework's aten.convolution_backward -- MIOpen -- for them): dw [Cout][3][3][Cin] float32 from x [B,H,W,Cin] and dy [B,Ho,Wo,Cout] bf16,
 * Ho = (H + 2 padding - 2 dilation - 1) / stride + 1.  Cin % 128 == 0, Cout % 128 == 0 (the workspace query returns 0 otherwise).
 * Fixed summation order (bit-reproducible); bias_partial / bias_rows / db as ssdhip_conv3x3_wgrad_bias_nhwc_bf16. */
size_t ssdhip_conv3x3_taps_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int stride, int padding, int dilation);
int ssdhip_conv3x3_taps_wgrad_bias_nhwc_bf16(const void* x, const void* dy, float* dw, const float* bias_partial, int bias_rows, float* db,
                                             int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int stride, int padding, int dilation,
                                             void* ws, size_t ws_bytes, void* stream);
/* out[c] = sum over the rows of partial [rows][C] float32 in a fixed order (C % 4 == 0): finishes the per-workgroup partial sums the
 * backward passes of this header leave behind where no weight-gradient reduction launch is at hand to take them along (conv1_1, L2Normalization's
 * gamma).  A kernel, not a library reduction: no memset node in a captured training step. */
int ssdhip_row_sums_f32(const float* partial, int rows, int C, float* out, void* stream);
/* First half of the DATA gradient of a strided or 'valid' 3x3 convolution (same layers): z [B,H,W,C] = zeros with gy [B,Ho,Wo,C] at
 * (offset + stride i, offset + stride j), offset = 1 - padding; the 3x3 'same' convolution of z with the transposed, tap-flipped
 * filters (any forward entry of this header) is d loss / d input. */
int ssdhip_embed_strided_nhwc_bf16(const void* gy, void* z, int B, int Ho, int Wo, int C, int H, int W, int stride, int offset, void* stream);

/* Backward of ssdhip_assemble_predictions_strided_bf16 for PACKED heads (the training step, round 5): the backward of the graph's
 * Reshape + Concatenate + softmax + Concatenate (models/keras_ssd300.py:363-419) in one launch.  grad_pred, y_pred [B, N, C+12] float32
 * (d loss / d predictions, and the predictions: the softmax probabilities are read from them); grad_heads[l]: the gradient of source map
 * l's packed head output, [B, n_anchors[l] / n_boxes[l], stride[l]] bf16 with channels [conf n_boxes C | loc n_boxes 4 | padding], written
 * whole (padding channels zero); host arrays of n_layers entries; stride[l] % 8 == 0, stride[l] >= n_boxes[l] (C + 4). */
int ssdhip_assemble_predictions_backward_bf16(int n_layers, void* const* grad_heads, const int* n_anchors, const int* n_boxes,
                                              const int* stride, const float* y_pred, const float* grad_pred, int B, int N, int C,
                                              void* stream);
/* LDS bytes that launch needs for these source maps (two 128-anchor row tiles + the widest packed stage); 0: refused arguments.  It
 * runs up to 160 KB - 64 (SSD300 / SSD512 heads: 21 classes 36 KB, 81 classes 117 KB); callers gate on this, not on a class count. */
size_t ssdhip_assemble_backward_lds_bytes(int n_layers, const int* n_boxes, const int* stride, int C);
#define SSDHIP_ASSEMBLE_BACKWARD_MAX_LDS (160 * 1024 - 64)

/* The FIRST layer's backward in one pass (round 5): conv1_1 of the training graph (Conv2D(64, (3, 3), activation='relu', padding='same') on the
 * 3-channel image, models/keras_ssd300.py:274) has no data gradient, so ReLU mask, bias gradient and weight gradient are one read of
 * gy, y [B,H,W,64] bf16 (gradient of the post-ReLU output, that output) and x [B,H,W,3] bf16.  wpart [n_blocks][64][27] float32 with
 * k = (kh*3 + kw)*3 + ci and bpart [n_blocks][64] float32 are per-workgroup partial sums: the caller adds the rows in order.
 * n_blocks = ssdhip_conv1_1_bwd_blocks(B, H, W) (0: bad arguments). */
int ssdhip_conv1_1_bwd_blocks(int B, int H, int W);
int ssdhip_conv1_1_bwd_nhwc_bf16(const void* gy, const void* y, const void* x, float* wpart, float* bpart, int B, int H, int W, int n_blocks,
                                 void* stream);
int ssdhip_relu_bwd_bias_nhwc_bf16(const void* gy, const void* y, void* out, float* partial, long long n_pixels, int C,
                                   int n_blocks, void* stream);

/* The same for a Conv2D(relu) -> MaxPooling2D(2, 2, 'same') pair (models/keras_ssd300.py:275-283: pool1 .. pool3) in ONE pass: y
 * [B,H,W,C] the post-ReLU activation, gp [B,ceil(H/2),ceil(W/2),C] the pooled map's gradient; out [B,H,W,C] = max-pool gradient (first
 * maximum of a window in row-major order, NaN wins) masked by y > 0; partial [n_blocks][C] as above with n_pixels = B H W.  Bit-identical
 * to ssdhip_maxpool_bwd_nhwc_bf16 followed by ssdhip_relu_bwd_bias_nhwc_bf16. */
int ssdhip_maxpool2_relu_bwd_bias_nhwc_bf16(const void* y, const void* gp, void* out, float* partial, int B, int H, int W, int C,
                                            int n_blocks, void* stream);
int ssdhip_maxpool_bwd_nhwc_bf16(const void* x, const void* gy, void* gx, int B, int H, int W, int C, int kernel, int stride,
                                 int pad, int Ho, int Wo, void* stream);

/* The same when a layer's two heads were computed by ONE wider convolution (conf and loc filters concatenated along Cout,
 * padded to the MFMA kernel's 64-channel granularity): conf_h[l] / loc_h[l] point at the first conf / loc channel of pixel 0 and
 * conf_stride_h[l] / loc_stride_h[l] give the number of bf16 elements between consecutive pixels (>= n_boxes*C / n_boxes*4;
 * NULL arrays = dense heads as above). */
int ssdhip_assemble_predictions_strided_bf16(int n_layers, const void* const* conf_h, const void* const* loc_h,
                                             const void* const* conf_bias_h, const void* const* loc_bias_h,
                                             const int* n_anchors_h, const int* n_boxes_h,
                                             const int* conf_stride_h, const int* loc_stride_h,
                                             const float* anchors_var, int B, int N, int C, float* y_pred, void* stream);

/* 'same' convolution (kernel 1 or 3, stride 1, any dilation, zero padding) + bias + ReLU as one implicit-GEMM MFMA kernel:
 * Conv2D(filters, (k,k), padding='same', activation='relu'[, dilation_rate]) of models/keras_ssd300.py:274-300.
 *   x [B,H,W,Cin] bf16 NHWC, weight [Cout,k,k,Cin] bf16 (torch OIHW weight in channels_last memory), bias [Cout] bf16 or NULL,
 *   y [B,H,W,Cout] bf16.  Cin % 64 == 0, Cout % 64 == 0; float32 accumulation, one rounding to bf16 after bias + activation. */
int ssdhip_conv2d_same_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                 int Cin, int Cout, int kernel, int dilation, int relu, void* stream);
/* General form of the above (same kernel, only the tile prologue's pixel -> address map differs): stride 1..4 and zero padding
 * 0 <= pad <= (kernel/2)*dilation on every side, torch.nn.Conv2d semantics -- the SSD extra layers conv6_2 / conv7_2
 * (ZeroPadding2D(1) + 3x3 stride 2, models/keras_ssd300.py:302-307) and conv8_2 / conv9_2 (3x3 'valid', :310-313).
 *   y [B,Ho,Wo,Cout], Ho = (H + 2*pad - dilation*(kernel-1) - 1) / stride + 1.  Tensors below 2 GiB (31-bit buffer offsets). */
int ssdhip_conv2d_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W, int Cin, int Cout,
                            int kernel, int stride, int pad, int dilation, int relu, void* stream);
/* The same with an explicit kernel variant (4: the default, two LDS stages; 5 / 6: the multi-stage ring of 32-channel slices with
 * loads three / two steps ahead of the MFMAs): the 10x10 ... 1x1 maps of the extra layers leave at most one workgroup per CU, where
 * only the prefetch depth hides the L2 latency.  Same numerics bar (the ring accumulates 32-channel slices: a different float32
 * summation order). */
int ssdhip_conv2d_nhwc_bf16_variant(int variant, const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                    int Cin, int Cout, int kernel, int stride, int pad, int dilation, int relu, void* stream);
/* Split-K form of ssdhip_conv2d_nhwc_bf16 for the layers whose K loop is one workgroup deep -- the SSD extra layers conv6_1 ...
 * conv9_2 (models/keras_ssd300.py:301-313: 10x10 ... 1x1 maps, a handful of 128-pixel tiles, each walking 4-36 K-steps at one L2
 * round trip per step): every tile's K loop is cut into `ksplit` ranges (0: chosen so that the launch has about one workgroup per
 * CU) that run side by side and leave float32 partial tiles in the caller's workspace; a second launch adds the ranges IN ORDER
 * (one fixed float32 summation order whatever the grid), then bias, activation and one rounding to bf16.  Same arguments and
 * numerics bar as ssdhip_conv2d_nhwc_bf16 (a different float32 summation order: not bit-identical to it).
 *   ws: ssdhip_conv2d_splitk_workspace_bytes(same geometry, same ksplit) bytes, 16-byte aligned. */
/* Reference-precision convolution: the reference's Conv2D layers are float32 (models/keras_ssd300.py:274-335).  A float32 value is the
 * sum of two float16 numbers to 2^-22 (hi = fl16(v), lo = fl16(v - hi)) and x.w = xhi.whi + xhi.wlo + xlo.whi to the same order, so
 * the convolution runs as ONE float16 MFMA K loop over 3 C channels with float32 accumulation -- float32-grade results at up to a
 * third of the 16-bit MFMA rate instead of the 1/16-rate float32 MFMA.
 *   x      [B,H,W,2C] float16 NHWC: channels [0,C) = hi, [C,2C) = lo of the float32 activation
 *   weight [Cout,k,k,3C] float16: [w hi | w lo | w hi] of (float32 filter / oscale), oscale a power of two chosen by the caller so that
 *          the scaled filters sit well inside float16's normal range
 *   bias   [Cout] float32 or NULL;  y = act(oscale * sum + bias) as float32 [B,Ho,Wo,Cout] (out_f32 != 0) or split again as float16
 *          [B,Ho,Wo,2 Cout]
 *   kernel 1 | 3, stride 1..4, 0 <= pad <= (kernel/2) dilation (torch.nn.Conv2d semantics); pool != 0: MaxPooling2D(2, 2, 'same')
 *   fused (stride 1, pad = (kernel/2) dilation only), y is then [B, ceil(H/2), ceil(W/2), .].  C % 64 == 0, Cout % 64 == 0. */
int ssdhip_conv2d_x3_nhwc_f16(const void* x, const void* weight, const float* bias, void* y, int B, int H, int W, int C, int Cout,
                              int kernel, int stride, int pad, int dilation, int relu, int pool, int out_f32, float oscale,
                              void* stream);

/* The same arithmetic on the slab kernel (csrc/ssdhip_convh.hip) for the 3x3 'same' layers with C % 128 == 0 and Cout % 128 == 0
 * (conv2_2 ... conv5_3 and the packed predictor heads): x [B,H,W,2C], weight [Cout,3,3,3C], bias float32 or NULL, y [B,Ho,Wo,2 Cout]
 * float16 = [hi | lo] of act(oscale * sum + bias); pool != 0 fuses MaxPooling2D(2, 2, 'same').
 * C == 64 (conv2_1) is accepted with the filters padded to four 64-channel slices: weight [Cout,3,3,256] = [w hi | w hi | w lo | 0]
 * (the K loop walks slices in pairs; the activation slices it meets are hi, lo, hi, lo). */
int ssdhip_conv3x3_halo_x3_nhwc_f16(const void* x, const void* weight, const float* bias, void* y, int B, int H, int W, int C, int Cout,
                                    int relu, int pool, float oscale, void* stream);

/* The layers of the reference-precision path that are not 64-channel GEMMs (csrc/ssdhip_layers.hip):
 * ssdhip_x3_split_nhwc    float32 [n_pixels, C] -> float16 [n_pixels, 2 C] = [hi | lo] (hi = fl16(v), lo = fl16(v - hi)); C % 8 == 0.
 * ssdhip_x3_merge_nhwc    the inverse: float32 hi + lo.
 * ssdhip_conv1_1_x3_nhwc  conv1_1 (models/keras_ssd300.py:274: Conv2D(64, (3, 3), padding='same', activation='relu') on the 3-channel
 *                         image) in float32, x [B,H,W,3] float32, weight [64,3,3,3] float32 (co, kh, kw, ci), bias float32 [64] or
 *                         NULL, y [B,H,W,128] float16 = [hi | lo]. */
int ssdhip_x3_split_nhwc(const float* x, void* y, long long n_pixels, int C, void* stream);
int ssdhip_x3_merge_nhwc(const void* x, float* y, long long n_pixels, int C, void* stream);
/* MaxPooling2D on a pair map without leaving the pair representation (round 6; pool4 / pool5 of models/keras_ssd300.py:287, 296 in the
 * reference-precision path): x [B, H, W, 2 C] float16 -> y [B, Ho, Wo, 2 C], the pair of the window's largest hi + lo; windows clipped
 * to the map (Keras 'same' / ceil mode: the caller passes Ho, Wo). */
int ssdhip_x3_maxpool_nhwc(const void* x, void* y, int B, int H, int W, int C, int kernel, int stride, int pad, int Ho, int Wo, void* stream);
/* L2Normalization (keras_layers/keras_layer_L2Normalization.py:62-70, conv4_3_norm of models/keras_ssd300.py:316) on a pair map: x, y
 * [n_pixels][2 C] float16 = [hi | lo], true input (hi + lo) * scale, output stored with divisor 1; C % 8 == 0.  The float32 result of
 * ssdhip_l2_normalize_fwd on the merged map, re-split -- one pass instead of merge + normalise + split (round 6). */
int ssdhip_x3_l2_normalize_nhwc(const void* x, const float* gamma, void* y, long long n_pixels, int C, float scale, void* stream);
int ssdhip_conv1_1_x3_nhwc(const float* x, const float* weight, const float* bias, void* y, int B, int H, int W, int relu, void* stream);
/* ... with the graph's input Lambdas in front (models/keras_ssd300.py:254-264: mean subtraction, stddev division, channel swap): images
 * [B,H,W,3] float32 as the generator hands them over, mean_h / divide_h / swap_h HOST arrays of three entries or NULL; conv1_1 reads
 * (images[swap[c]] - mean[swap[c]]) / divide[swap[c]], the framework's float32 expression bit for bit (round 6: three passes over the
 * batch less in the reference-precision step). */
int ssdhip_conv1_1_x3_pre_nhwc(const float* images, const float* weight, const float* bias, void* y, int B, int H, int W, int relu,
                               const float* mean_h, const float* divide_h, const int* swap_h, void* stream);

size_t ssdhip_conv2d_splitk_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kernel, int stride, int pad, int dilation,
                                            int ksplit);
int ssdhip_conv2d_splitk_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W, int Cin, int Cout,
                                   int kernel, int stride, int pad, int dilation, int relu, int ksplit, void* ws, size_t ws_bytes,
                                   void* stream);

/* Profiling aid: the same with an explicit kernel variant (4: the shipped kernel -- 128-pixel tile, two LDS stages, buffer-addressed
 * LDS-DMA loads, batched fragment reads; 1: its predecessor with per-lane pointers; 3: 256-pixel tile, three-stage weight pipeline,
 * kw-reuse of the activation strip; 5 / 6: four / three-stage LDS ring of 32-channel slices; 9: eight waves per workgroup with the
 * step's MFMAs split over two wave groups -- all kept for A/B timing, all covered by the parity tests). */
int ssdhip_conv2d_same_nhwc_bf16_variant(int variant, const void* x, const void* weight, const void* bias, void* y,
                                         int B, int H, int W, int Cin, int Cout, int kernel, int dilation, int relu, void* stream);

/* The two entry points below from FLOAT32 head outputs whose bias the convolution has already added -- the packed conf + loc maps of the
 * reference-precision path (ssdhip_conv2d_x3_nhwc_f16 with out_f32; models/precise.py): strides count float32 elements, no bias
 * arrays.  The softmax is the float32 expression of the bf16 form. */
int ssdhip_assemble_predictions_strided_f32(int n_layers, const void* const* conf_h, const void* const* loc_h, const int* n_anchors_h,
                                            const int* n_boxes_h, const int* conf_stride_h, const int* loc_stride_h,
                                            const float* anchors_var, int B, int N, int C, float* y_pred, void* stream);
int ssdhip_decode_from_heads_f32(int n_layers, const void* const* conf_h, const void* const* loc_h, const int* n_anchors_h,
                                 const int* n_boxes_h, const int* conf_stride_h, const int* loc_stride_h, const float* anchors_var,
                                 int B, int N, int C, double conf_thresh, double iou_thresh, int top_k, int nms_cap, int class_agnostic,
                                 int semantics, int coords, int normalize_coords, double img_height, double img_width, int border_pixels,
                                 void* out, int out_dtype, int out_rows, int* out_count, int* out_anchor_idx, void* ws, size_t ws_bytes,
                                 void* stream);

/* DecodeDetections straight from the predictor heads: the arguments of ssdhip_assemble_predictions_strided_bf16
 * followed by those of ssdhip_decode_detections.  The [C+12]-float prediction rows are built in LDS (bias, softmax, anchors) and
 * decoded / thresholded at once; y_pred is never written.  Results are identical to assembling and then decoding. */
int ssdhip_decode_from_heads(int n_layers, const void* const* conf_h, const void* const* loc_h,
                             const void* const* conf_bias_h, const void* const* loc_bias_h,
                             const int* n_anchors_h, const int* n_boxes_h, const int* conf_stride_h,
                             const int* loc_stride_h, const float* anchors_var, int B, int N, int C,
                             double conf_thresh, double iou_thresh, int top_k, int nms_cap, int class_agnostic,
                             int semantics, int coords, int normalize_coords, double img_height, double img_width,
                             int border_pixels, void* out, int out_dtype, int out_rows, int* out_count,
                             int* out_anchor_idx, void* ws, size_t ws_bytes, void* stream);

/* The same convolution followed by MaxPooling2D(pool_size 2, strides 2, padding 'same') in ONE kernel (conv1_2 -> pool1,
 * conv2_2 -> pool2, conv3_3 -> pool3 of models/keras_ssd300.py:274-290): the 2x2 maximum is taken on the float32 accumulators,
 * then bias, ReLU and one bf16 rounding -- equal to pooling the rounded activations because all three are monotonic.
 *   y [B, ceil(H/2), ceil(W/2), Cout] bf16; windows are clipped to the map (odd H or W). */
int ssdhip_conv2d_same_pool2_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                       int Cin, int Cout, int kernel, int dilation, int relu, void* stream);

/* n_problems (<= 8) independent 'same' convolutions of the kind above in ONE launch -- the predictor heads of all source layers
 * (models/keras_ssd300.py:322-361), whose small members are latency-bound when launched alone.  Every array is a HOST array with one
 * entry per problem (device pointers inside x_h / weight_h / bias_h / y_h; bias_h or its entries may be NULL). */
int ssdhip_conv2d_same_group_nhwc_bf16(int n_problems, const void* const* x_h, const void* const* weight_h,
                                       const void* const* bias_h, void* const* y_h, const int* B_h, const int* H_h,
                                       const int* W_h, const int* Cin_h, const int* Cout_h, const int* kernel_h,
                                       const int* dilation_h, int relu, void* stream);

/* 3x3 'same' convolution for the Cin = 64 layers (conv1_2, conv2_1: models/keras_ssd300.py:275-279) + bias + ReLU, optionally with
 * MaxPooling2D(2, 2, 'same') fused (pool != 0; y then [B, ceil(H/2), ceil(W/2), Cout]).  Persistent workgroups keep the whole
 * 72 KB filter bank of a 64-channel output slice resident in LDS and stream one activation halo per 128-pixel tile
 * (csrc/ssdhip_conv64.hip).  Cin must be 64, Cout % 64 == 0; n_workgroups = persistent workgroups to launch (the CU count; 0 = 256).
 * Same numerics as ssdhip_conv2d_same[_pool2]_nhwc_bf16 (bit-identical results). */
int ssdhip_conv3x3_c64_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                 int Cin, int Cout, int relu, int pool, int n_workgroups, void* stream);
/* The same layer followed by MaxPooling2D(2, 2, 'same') in the TRAINING step (round 6): one launch writes y_pooled [B, ceil(H/2), ceil(W/2),
 * Cout] (as pool != 0 above) AND y_full [B, H, W, Cout], the activation the backward pass reads -- bit-identical to the un-pooled launch
 * followed by ssdhip_bias_act_maxpool, without reading the full-resolution map back (conv1_2 -> pool1: 368 MB at batch 32). */
int ssdhip_conv3x3_c64_pool_keep_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y_full, void* y_pooled, int B, int H,
                                           int W, int Cin, int Cout, int relu, int n_workgroups, void* stream);

/* 3x3 'same' convolution (stride 1, dilation 1) + bias + ReLU for the VGG blocks with Cin % 128 == 0 and Cout % 128 == 0 (conv2_2,
 * conv3_x, conv4_x, conv5_x: models/keras_ssd300.py:279-296, keras_ssd512.py twins), optionally with MaxPooling2D(2, 2, 'same')
 * fused (pool != 0: y is [B, ceil(H/2), ceil(W/2), Cout]): the activations of a 256-pixel tile come into LDS once per 64-channel
 * slice and the nine taps read them at nine displacements (csrc/ssdhip_convh.hip).  SSDHIP_E_BADARG for other channel counts.
 * Same numerics as ssdhip_conv2d_same[_pool2]_nhwc_bf16 (bit-identical results). */
int ssdhip_conv3x3_halo_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                  int Cin, int Cout, int relu, int pool, void* stream);

/* The tiling the two slab entries pick for a batch of H x W maps (host arithmetic, no launch): plan[0] = 0 (padded position grid:
 * unpooled maps up to 94 wide) | 4 (16 x 16 pixel tiles) | 5 (8 x 32), plan[1] = position tiles (x Cout / 128 = tile units; one
 * persistent workgroup per CU walks them), plan[2] = row pitch of the STACKED batch -- pooled calls lay the images on top of each
 * other with a gap of one or two zero rows and tile the stack when that needs fewer tiles than tiling every image (SSD300's conv3_3 +
 * pool3 at batch 32: 760 instead of 800, six rounds of 256 CUs instead of seven) -- or 0, plan[3] = rows of tiles.  An un-pooled map
 * up to 94 wide leaves the position grid for 2-D tiles when those finish in fewer rounds of one workgroup per CU (Cout / 128 tile
 * units per position tile: SSD512's conv4_x / conv5_x at batch 16 -- four rounds instead of five, one instead of two).  The results
 * do not depend on the tiling (same accumulation order per output). */
int ssdhip_conv3x3_halo_plan(int B, int H, int W, int Cout, int pool, int* plan);

/* Training step: the data gradient of a 3x3 'same' layer whose input is the ReLU output of the layer below, with that layer's
 * threshold_backward (keras Conv2D(activation='relu') under autodiff, models/keras_ssd300.py:279-291) in the epilogue: y = the 3x3
 * 'same' convolution of x [B,H,W,Cin] with weight [Cout,3,3,Cin] (the transposed, tap-flipped filters; no bias, no activation) where
 * mask [B,H,W,Cout] is > 0 or NaN, zero elsewhere.  Cin % 128 == 0, Cout % 128 == 0.  Bit-identical to ssdhip_conv3x3_halo_nhwc_bf16
 * followed by the mask of ssdhip_relu_bwd_bias_nhwc_bf16; saves that pass's read of both maps and its write. */
int ssdhip_conv3x3_halo_masked_nhwc_bf16(const void* x, const void* weight, const void* mask, void* y, float* bias_partial,
                                         int bias_rows, int B, int H, int W, int Cin, int Cout, void* stream);
/* bias_partial (or NULL): [bias_rows][Cout] float32 whose column sums are the channel sums of y -- the bias gradient of the layer below,
 * which then needs no pass over the map at all; every entry is written, in a fixed summation order.  bias_rows must be what this
 * returns for the call's geometry (0: not supported). */
int ssdhip_conv3x3_halo_masked_bias_rows(int B, int H, int W, int Cout);

/* Training step: Conv2D(relu) -> MaxPooling2D(2, 2, 'same') (models/keras_ssd300.py:279-287: conv2_2 -> pool2, conv3_3 -> pool3) in ONE
 * launch that writes the activation the backward pass needs, y_full [B,H,W,Cout], AND the pooled map y_pooled
 * [B,ceil(H/2),ceil(W/2),Cout] -- the accumulators survive the pooled epilogue -- instead of an un-pooled launch and a pooling pass
 * that reads the map back.  Cin % 128 == 0, Cout % 128 == 0.  Both maps bit-identical to ssdhip_conv3x3_halo_nhwc_bf16 (pool 0 / 1). */
int ssdhip_conv3x3_halo_pool_keep_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y_full, void* y_pooled, int B, int H,
                                            int W, int Cin, int Cout, int relu, void* stream);

/* A chain of small convolutions (+ bias + ReLU) in ONE launch, one workgroup per image, the intermediate maps in LDS: the tail of the SSD
 * extra layers conv7_1 ... conv9_2 (models/keras_ssd300.py:304-313).  x [B, H, W, C0] bf16 NHWC; layer i: k_i x k_i, stride_i, zero padding
 * pad_i, Cout_i, bias_i (bf16 or NULL), ReLU if relu_i != 0; y_h[i] != NULL: that layer's map is also written to y_h[i]
 * [B, H_i, W_i, Cout_i].  All arrays are HOST arrays of n_layers (<= 8) entries; packed_h[i] = the layer's filters [Cout, k, k, Cin] bf16
 * re-ordered by ssdhip_conv_chain_pack_weight (MFMA fragment order, same byte count: ssdhip_conv_chain_packed_bytes, 0 = unsupported).
 * Cin_i % 128 == 0, Cout_i % 32 == 0 and one image's maps must fit the CU's LDS; SSDHIP_E_BADARG otherwise (csrc/ssdhip_chain.hip). */
size_t ssdhip_conv_chain_packed_bytes(int k, int Cin, int Cout);
int ssdhip_conv_chain_pack_weight(const void* weight, void* packed, int k, int Cin, int Cout, void* stream);
int ssdhip_conv_chain_nhwc_bf16(const void* x, int B, int H, int W, int C0, int n_layers, const void* const* packed_h,
                                const void* const* bias_h, void* const* y_h, const int* k_h, const int* stride_h, const int* pad_h,
                                const int* cout_h, const int* relu_h, void* stream);
/* The same chain at the reference's precision (round 6; the float32 Keras graph of models/keras_ssd300.py:304-313 on the float16 x 3 MFMA
 * path): x [B, H, W, 2 C0] float16 = [hi | lo] pairs; layer i computes act(mul_i * conv(x, w_i) + bias_i) in float32 -- three float16
 * products per K-step, mul_i = oscale_i * (divisor of its input) / (divisor of its output), bias_i float32 already divided by the output
 * divisor -- and re-splits; y_h[i] != NULL: [B, H_i, W_i, 2 Cout_i] pairs.  packed_h[i]: the layer's [Cout, k, k, 3 Cin] float16 filters
 * ([w hi | w lo | w hi], as for ssdhip_conv2d_x3_nhwc_f16) re-ordered by ssdhip_conv_chain_x3_pack_weight.  The first layer must be
 * 1 x 1 / stride 1 / no padding (it reads x from global memory); Cin_i % 64 == 0, Cout_i % 32 == 0; the later maps must fit the CU's LDS. */
size_t ssdhip_conv_chain_x3_packed_bytes(int k, int Cin, int Cout);
int ssdhip_conv_chain_x3_pack_weight(const void* weight_x3, void* packed, int k, int Cin, int Cout, void* stream);
int ssdhip_conv_chain_x3_nhwc_f16(const void* x, int B, int H, int W, int C0, int n_layers, const void* const* packed_h,
                                  const float* const* bias_h, void* const* y_h, const int* k_h, const int* stride_h, const int* pad_h,
                                  const int* cout_h, const int* relu_h, const float* mul_h, void* stream);

/* Weight gradient of a 3x3 'same' stride-1 dilation-1 convolution of the training graph -- what the TensorFlow graph behind
 * model.fit_generator computes for every Conv2D of models/keras_ssd300.py:274-296 (ssd300_training.ipynb:171-173):
 *     dw[co][kh][kw][ci] = sum_{b,h,w} dy[b,h,w,co] * x[b, h + kh - 1, w + kw - 1, ci]      (zero padding)
 * x [B, H, W, Cin] bf16, dy [B, H, W, Cout] bf16 (gradient w.r.t. the convolution's output, activation mask already applied),
 * dw [Cout, 3, 3, Cin] float32 (the channels_last layout of a [Cout, Cin, 3, 3] tensor).  MFMA kernel on transposed LDS fragments
 * (ds_read_b64_tr_b16), split over positions with an ordered float32 reduction (csrc/ssdhip_wgrad.hip): bit-reproducible.
 * Cin % 64 == 0 and (Cout % 128 == 0, W <= 190) or (Cout % 64 == 0, W <= 318); SSDHIP_E_BADARG otherwise.
 * ssdhip_conv3x3_wgrad_workspace_bytes returns 0 for an unsupported geometry. */
size_t ssdhip_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
/* ssdhip_conv3x3_wgrad_bias_nhwc_bf16: the same, plus the layer's bias gradient db [Cout] float32 from bias_partial [bias_rows][Cout]
 * (per-workgroup channel sums of dy as the backward passes of csrc/ssdhip_train.hip write them), its rows added in a fixed order by
 * extra workgroups of the reduction launch; bias_partial == NULL: weight gradient only. */
int ssdhip_conv3x3_wgrad_bias_nhwc_bf16(const void* x, const void* dy, float* dw, const float* bias_partial, int bias_rows, float* db,
                                        int B, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int ssdhip_conv3x3_wgrad_nhwc_bf16(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout, void* ws,
                                   size_t ws_bytes, void* stream);

/* 3x3 convolution, stride 1 | 2, zero padding 0 | 1 (torch.nn.Conv2d semantics), Cin % 128 == 0, Cout % 128 == 0, map up to 94 wide,
 * through the same kernel: the SSD extra layers conv6_2 / conv7_2 (ZeroPadding2D(1) + stride 2, models/keras_ssd300.py:302-307) and
 * conv8_2 / conv9_2 ('valid', :310-313).  y is [B, (H + 2 pad - 3) / stride + 1, (W + 2 pad - 3) / stride + 1, Cout]; bit-identical to
 * ssdhip_conv2d_nhwc_bf16. */
int ssdhip_conv3x3_halo_strided_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                          int Cin, int Cout, int stride, int pad, int relu, void* stream);

/* n_problems (<= 8) independent convolutions of the kind above (no pooling, maps up to 62 wide) in ONE launch of persistent
 * workgroups, deepest problem first -- the packed predictor heads of all source maps (models/keras_ssd300.py:322-335); arrays are HOST
 * arrays of per-problem arguments; max_workgroups > 0 caps the workgroups (one per CU) so a concurrent stream finds free CUs. */
int ssdhip_conv3x3_halo_group_nhwc_bf16(int n_problems, const void* const* x_h, const void* const* weight_h, const void* const* bias_h,
                                        void* const* y_h, const int* B_h, const int* H_h, const int* W_h, const int* Cin_h,
                                        const int* Cout_h, int relu, int max_workgroups, void* stream);

/* conv1_1 -> conv1_2 [-> pool1] as ONE kernel (models/keras_ssd300.py:274-276): x3 [B, H, W, 3] bf16 image, w1 [64, 3, 3, 3] + b1 the
 * first layer (ReLU), weight [Cout, 3, 3, 64] + bias the second one, pool != 0 fuses MaxPooling2D(2, 2, 'same').  The 64-channel map
 * between the two layers is never written: each tile's halo of it is recomputed from the image inside the kernel
 * (csrc/ssdhip_conv64.hip).  Bit-identical to ssdhip_conv3x3_cin3_nhwc_bf16 followed by ssdhip_conv3x3_c64_nhwc_bf16. */
int ssdhip_conv1_block_nhwc_bf16(const void* x3, const void* w1, const void* b1, const void* weight, const void* bias, void* y,
                                 int B, int H, int W, int Cout, int relu, int pool, int n_workgroups, void* stream);

/* First layer (conv1_1, models/keras_ssd300.py:274): 3x3 'same' convolution of a 3-channel NHWC bf16 image into 64 channels
 * + bias + ReLU, one thread per pixel (the op is bound by writing the 64-channel map).  Cin must be 3, Cout 64. */
int ssdhip_conv3x3_cin3_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                  int Cin, int Cout, int relu, void* stream);

/* The geometric half of SSDDataAugmentation (data_generator/data_augmentation_chain_original_ssd.py:208-280: expansion -> random crop ->
 * random flip -> resize with a random interpolation mode) for a whole batch in ONE launch: x [B, H, W, C] uint8 (device), y [B, Ho, Wo, C];
 * per image its own tap tables ix / wx [B][Wo][nx], iy / wy [B][Ho][ny]: the taps of cv2.resize on that image's patch, composed on the
 * host with its expansion / crop / flip index maps -- an index addresses a column / row of the ORIGINAL image, -1 = background
 * (background [B][C] uint8).  Arithmetic of ssdhip_image_resize_u8. */
int ssdhip_image_resize_gather_u8(const void* x, void* y, int B, int H, int W, int Ho, int Wo, int C, const int* ix_dev,
                                  const double* wx_dev, int nx, const int* iy_dev, const double* wy_dev, int ny,
                                  const void* background_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image half of the training-time augmentation (csrc/ssdhip_image.hip): what the reference does per image on the host
 * through OpenCV (data_generator/object_detection_2d_photometric_ops.py:23-480, object_detection_2d_geometric_ops.py:27-148), for a
 * batch of images resident on the device.  dtype codes: 0 uint8, 1 float32, 2 float64.
 *
 * ssdhip_image_program   x [n_images][pixels][3] (uint8 | float32 | float64) -> y (dtype out_dtype): image i runs ops[i][0..15] (0 ends):
 *     1 astype(float32)   2 np.round(.).astype(uint8)   3 brightness: clip(v + arg, 0, 255)   4 contrast: clip(127.5 + arg (v - 127.5), 0, 255)
 *     5 saturation: channel 1 = clip(ch1 * arg, 0, 255)   6 hue: channel 0 = (ch0 + arg) % 180   7 RGB -> HSV   8 HSV -> RGB
 *     9 RGB -> grey on all three channels   10 channel permutation arg = o0 + 4 o1 + 16 o2
 *   in NumPy's arithmetic for the array's current dtype (:128-130, :185, :242, :300: float32 operations on float32 images; float64
 *   results for uint8 images under 3 / 4; truncating in-place stores for 5 / 6 on uint8 images); 7 / 8 are cv2.cvtColor's 8-bit
 *   (H in [0, 180)) or float32 (H in [0, 360)) conversions.  out_dtype must be the dtype the program ends in.  ops / args: device arrays
 *   [n_images][16] (int32 / float64).
 * ssdhip_image_resize_u8 cv2.resize as separable resampling (object_detection_2d_geometric_ops.py:70-72): x [B,H,W,C] -> y [B,Ho,Wo,C],
 *   out = rint(sum_j wy[yo][j] * (sum_t wx[xo][t] * x[iy[yo][j]][ix[xo][t]])) clipped to [0, 255]; the caller builds the tap tables
 *   (device arrays [Wo][nx], [Ho][ny]; int32 indices, float64 weights) for the interpolation mode it wants.
 * ssdhip_image_hist_u8   256-bin histogram of one channel of an interleaved uint8 image (cv2.equalizeHist's first half, :407).
 * ssdhip_image_lut_u8    y = table[x] on the channels of channel_mask, x elsewhere (cv2.LUT :359 / the equalisation table). */
int ssdhip_image_program(const void* x, int in_dtype, void* y, int out_dtype, int n_images, long long pixels_per_image,
                         const int* ops_dev, const double* args_dev, void* stream);
int ssdhip_image_resize_u8(const void* x, void* y, int B, int H, int W, int Ho, int Wo, int C, const int* ix_dev, const double* wx_dev,
                           int nx, const int* iy_dev, const double* wy_dev, int ny, void* stream);
/* cv2.resize on 8-bit images with OpenCV's own arithmetic (round 6; data_generator/object_detection_2d_geometric_ops.py:70-72 calls
 * cv2.resize -> imgproc/resize.cpp): `kind` 0 nearest, 1 linear (11-bit fixed-point coefficients, the two-stage vertical rounding),
 * 2 cubic / Lanczos-4 (fixed point, (sum + 2^21) >> 22), 3 area (float32 tables, cvRound), 4 fast area (block sum * (1.f / area)),
 * 5 fast area 2 x 2 ((sum + 2) >> 2), 6 copy; the tables hold the shorts / float32 weights / ones as float64 values.
 * ssdhip_image_resize_cv_u8: one plan for the batch, ix / wx [Wo][nx], iy / wy [Ho][ny].  ssdhip_image_resize_gather_cv_u8: a plan per
 * image, plan [B][4] = kind, area, taps per column, taps per row; tables [B][Wo][nx] / [B][Ho][ny], nx, ny >= 2; index -1 reads
 * background [B][C] (the augmentation chain's expansion canvas). */
int ssdhip_image_resize_cv_u8(const void* x, void* y, int B, int H, int W, int Ho, int Wo, int C, int kind, int area, const int* ix_dev,
                              const double* wx_dev, int nx, const int* iy_dev, const double* wy_dev, int ny, void* stream);
int ssdhip_image_resize_gather_cv_u8(const void* x, void* y, int B, int H, int W, int Ho, int Wo, int C, const int* plan_dev,
                                     const int* ix_dev, const double* wx_dev, int nx, const int* iy_dev, const double* wy_dev, int ny,
                                     const void* background_dev, void* stream);
int ssdhip_image_hist_u8(const void* x, long long n_pixels, int C, int channel, unsigned int* hist_dev, void* stream);
int ssdhip_image_lut_u8(const void* x, void* y, long long n_values, int C, int channel_mask, const void* table_dev, void* stream);

/* 3x3 'same' convolution with any dilation on SMALL maps (H * W <= 384 pixels), one image per tile with its 64-channel slices resident in
 * LDS and the dilated taps as per-lane addresses (csrc/ssdhip_convimg.hip).  Replaces Conv2D(1024, (3, 3), dilation_rate=(6, 6),
 * activation='relu', padding='same') -- fc6, models/keras_ssd300.py:298 -- on the 19 x 19 map; same K order, hence the same bits, as
 * ssdhip_conv2d_same_nhwc_bf16.  x [B, H, W, Cin] bf16, weight [Cout, 3, 3, Cin] bf16, bias [Cout] bf16 or NULL, y [B, H, W, Cout] bf16;
 * Cin % 64 == 0, Cout % 64 == 0, 1 <= dilation <= 16.  SSDHIP_E_BADARG for any other geometry.  A tile is one image x 128 output channels
 * (or x 64 where that is needed to fill the chip: conv5_x at batch 32). */
int ssdhip_conv3x3_image_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W, int Cin, int Cout,
                                   int dilation, int relu, void* stream);
/* The same kernel family for k x k filters, k in {1, 3}, with stride and zero padding (round 6): fc7 = Conv2D(1024, (1, 1)) and
 * conv6_1 = Conv2D(256, (1, 1)) (models/keras_ssd300.py:299, 301: a 1 x 1 layer is ONE step per 64-channel slice, the image's slice
 * resident in LDS), conv6_2 = ZeroPadding2D(1) + Conv2D(512, (3, 3), strides=(2, 2), padding='valid') (:302-303: 19 x 19 -> 10 x 10,
 * a 128-pixel tile form).  y [B, Ho, Wo, Cout] with Ho = (H + 2 padding - dilation (k - 1) - 1) / stride + 1; H * W <= 384,
 * Ho * Wo <= 384, 1 <= stride <= 4, 0 <= padding <= dilation (k / 2); otherwise as above.  Bit-identical to ssdhip_conv2d_nhwc_bf16. */
int ssdhip_conv2d_image_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W, int Cin, int Cout,
                                  int ksize, int stride, int padding, int dilation, int relu, void* stream);
/* ... and its reference-precision form (models/precise.py; the float32 graph of models/keras_ssd300.py:274-335 on the float16 MFMA rate):
 * x [B, H, W, 2 C] float16 (hi | lo), weight [Cout, k, k, 3 C] float16 (w hi | w lo | w hi), bias float32 or NULL, y the next layer's
 * pairs [B, Ho, Wo, 2 Cout] float16 or (out_f32) [B, Ho, Wo, Cout] float32; acc * oscale + bias, activation on float32.  Bit-identical
 * to ssdhip_conv2d_x3_nhwc_f16 (same K order). */
int ssdhip_conv2d_image_x3_nhwc_f16(const void* x, const void* weight, const float* bias, void* y, int B, int H, int W, int C, int Cout,
                                    int ksize, int stride, int padding, int dilation, int relu, int out_f32, float oscale, void* stream);

/* ------------------------------------------------------------------------------------------
 * The parameter side of the training step (csrc/ssdhip_optim.hip): ONE launch over all parameters.  The reference trains float32
 * weights with keras.optimizers.SGD(lr=0.001, momentum=0.9) (ssd300_training.ipynb:169-173); the MFMA kernels read bf16 copies.
 *
 * A table of descriptors in DEVICE memory names the tensors (the caller builds it once: the pointers are those of persistent buffers):
 *   weights   src = float32 master filters [O][I][kh][kw] (contiguous; or [O][kh][kw][I] with src_channels_last = 1), KK = kh * kw <= 16;
 *             cl  = bf16 [O][kh][kw][I] (channels_last: what the forward and the weight-gradient kernels read) or NULL;
 *             tr  = bf16 [I][kh][kw][tr_ostride] with this tensor's O channels at offset tr_ooff and the taps FLIPPED -- the filters of
 *                   the data gradient (a stride-1 'same' convolution of dL/dy), or NULL; tr_ostride / tr_ooff let the conf and loc
 *                   heads of a source map share one packed tensor (as `cl` does through its pointer);
 *             tile0 = index of the tensor's first 32 x 32 (O x I) tile in the launch (tiles are numbered tensor by tensor,
 *                   ceil(O / 32) * ceil(I / 32) each); n_tiles = their total.
 *   vectors   (descriptors n_weights .. n_weights + n_vectors - 1) src = float32 [O], cl = bf16 [O]; tile0 = index of its first
 *             256-element block among the n_vector_blocks vector blocks.
 * ssdhip_shadow_refresh      bf16 (round to nearest even) copies of every tensor of the table in the layouts above.
 * ssdhip_sgd_momentum_step   torch.optim.SGD's update with momentum (dampening 0, no Nesterov) = Keras SGD's for a constant learning
 *                            rate, over n_tensors float32 tensors named by HOST arrays of device pointers (parameter, gradient,
 *                            momentum buffer -- zeros before the first step -- and element count; 16-byte aligned): buf = momentum *
 *                            buf + g, p -= lr * buf, g += weight_decay * p first when weight_decay != 0.  The table travels in the
 *                            kernel arguments (80 tensors per launch): no upload, no host synchronisation. */
typedef struct ssdhip_shadow_desc {
    const void* src;
    void* cl;
    void* tr;
    int O, I, KK, tr_ostride, tr_ooff, tile0;
    int src_channels_last;   /* weights: 0 = the master is [O][I][kh][kw], 1 = [O][kh][kw][I] (a channels_last nn.Conv2d weight) */
    int reserved;
} ssdhip_shadow_desc;
int ssdhip_shadow_refresh(const ssdhip_shadow_desc* table_dev, int n_weights, int n_tiles, int n_vectors, int n_vector_blocks, void* stream);
int ssdhip_sgd_momentum_step(int n_tensors, void* const* params_h, const void* const* grads_h, void* const* bufs_h,
                             const long long* numel_h, double lr, double momentum, double weight_decay, void* stream);

/* ------------------------------------------------------------------------------------------
 * The decisions of the original-SSD augmentation chain for a whole batch in ONE launch (csrc/ssdhip_augment.hip): SSDExpand ->
 * SSDRandomCrop -> RandomFlip -> ResizeRandomInterp of data_generator/data_augmentation_chain_original_ssd.py:208-280 (with
 * object_detection_2d_patch_sampling_ops.py:24-339, object_detection_2d_geometric_ops.py:86-262 and the BoxFilter / ImageValidator of
 * object_detection_2d_image_boxes_validation_utils.py:79-322), everything but the pixels.  One wave per image consumes the image's own
 * NumPy MT19937 stream exactly as numpy.random does (uniform, randint, choice), so decisions, labels and the stream position afterwards
 * are those of the reference chain called under the same generator state.
 *   mt_state      [B][625] uint32: np.random.RandomState.get_state() -- 624 key words + the position -- of every image;
 *   labels        [B][64][5] float64 rows (class_id, xmin, ymin, xmax, ymax), n_labels [B] <= 64 (int64 and float64 label arrays are
 *                 both exact in float64);
 *   geometry      [B][12] int32: expanded?, canvas top, left, height, width | cropped?, patch top, left, height, width | flipped?,
 *                 interpolation mode -- what the gather launch (ssdhip_image_resize_gather_u8) needs;
 *   labels_out    [B][64][5] float64 + n_labels_out [B]: the surviving boxes in the output image's coordinates;
 *   mt_state_out  [B][625]: the generator states behind the chain. */
typedef struct ssdhip_augment_params {
    int img_height, img_width;                       /* size of the batch's images */
    double expand_prob, expand_min_scale, expand_max_scale;
    double crop_prob, crop_min_scale, crop_max_scale, crop_min_aspect_ratio, crop_max_aspect_ratio;
    int n_trials, n_bounds;
    double bound_cdf[8], bound_lower[8], bound_upper[8];   /* BoundGenerator: cumsum(weights) / its last element, (lower, upper] pairs */
    double flip_prob;
    int n_modes, interpolation_modes[8], out_height, out_width;
    int max_rounds;                                  /* 0: 100 000 sampling rounds at most (the reference loops without a limit) */
} ssdhip_augment_params;
/* ssdhip_augment_taps: the tap tables of ssdhip_image_resize_gather_u8 for the batch, built on the device from `geometry` (as
 * ssdhip_ssd_augment_decide leaves it): cv2.resize's source indices / float64 weights of each image's interpolation mode (0 nearest, 1
 * linear, 2 cubic, 3 area, 4 Lanczos-4) composed with its flip, crop window and expansion canvas (-1: a canvas pixel).  ix / wx
 * [B][out_w][n_taps], iy / wy [B][out_h][n_taps]; n_taps >= 8 and >= ceil(largest source / output extent ratio) + 1 (the area filter). */
int ssdhip_augment_taps(const int* geometry_dev, int B, int H, int W, int out_h, int out_w, int n_taps, int* ix_dev, double* wx_dev,
                        int* iy_dev, double* wy_dev, void* stream);
/* ssdhip_augment_plans (round 6): as ssdhip_augment_taps, with cv2.resize's own 8-bit arithmetic: plan [B][4] = kind, area, taps per
 * column, taps per row and the tables of ssdhip_image_resize_gather_cv_u8 (fixed-point shorts / float32 area weights / ones as float64
 * values); n_taps >= 8 and >= ceil(largest source / output extent ratio) + 2. */
int ssdhip_augment_plans(const int* geometry_dev, int B, int H, int W, int out_h, int out_w, int n_taps, int* plan_dev, int* ix_dev,
                         double* wx_dev, int* iy_dev, double* wy_dev, void* stream);
int ssdhip_ssd_augment_decide(const ssdhip_augment_params* params, int B, const unsigned int* mt_state, const double* labels,
                              const int* n_labels, int* geometry, double* labels_out, int* n_labels_out, unsigned int* mt_state_out,
                              void* stream);
/* Round 6: the same decisions on ONE generator for the whole batch -- the reference's own semantics: its generator loop
 * (data_generator/object_detection_2d_data_generator.py:1050-1089) calls SSDDataAugmentation (data_augmentation_chain_original_ssd.py:
 * 208-280) image after image on the global np.random stream.  One wave walks the images in order and ALSO takes the photometric decisions
 * (SSDPhotometricDistortions :146-208), written as the per-image programs of ssdhip_image_program: programs_ops [B][16] int32,
 * programs_args [B][16] float64.  mt_state / mt_state_out [625]: np.random.get_state() before the batch / the state to put back after it.
 * photo: probability and uniform range of RandomBrightness, RandomContrast, RandomSaturation, RandomHue (in this order);
 * RandomChannelSwap's probability must be 0 (the original-SSD configuration). */
typedef struct ssdhip_augment_photo {
    double prob[4], lower[4], upper[4];
    double swap_prob;
} ssdhip_augment_photo;
int ssdhip_ssd_augment_decide_stream(const ssdhip_augment_params* params, const ssdhip_augment_photo* photo, int B,
                                     const unsigned int* mt_state, const double* labels, const int* n_labels, int* programs_ops,
                                     double* programs_args, int* geometry, double* labels_out, int* n_labels_out,
                                     unsigned int* mt_state_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSDHIP_H */
