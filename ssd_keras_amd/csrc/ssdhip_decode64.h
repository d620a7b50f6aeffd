// Internal interface of csrc/ssdhip_decode64.hip (the decoders' float64 flow), called by decode_run in ssdhip_decode.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ssdhip {

size_t decode64_workspace_bytes(int B, int N, int C, int top_k, int nms_cap, int class_agnostic);

// Same arguments as ssdhip_decode_stages with in_dtype == SSDHIP_F64 (already validated by the caller).
int decode64_run(int stages, const double* y_pred, int B, int N, int C, double conf_thresh, double iou_thresh, int top_k,
                 int nms_cap, int class_agnostic, int semantics, int coords, int normalize_coords, double img_height,
                 double img_width, int border_pixels, void* out, int out_dtype, int out_rows, int* out_count,
                 int* out_anchor_idx, void* ws, size_t ws_bytes, hipStream_t stream);

}  // namespace ssdhip
