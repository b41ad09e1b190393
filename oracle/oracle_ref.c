/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of the two native ops on the HVRNet
 * forward path.  Nothing under hvrnet_amd/ may link, import or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 *   oracle_roi_align_fwd / _bwd : line-by-line scalar restatement of
 *       mmdet/ops/roi_align/src/roi_align_kernel.cu:16-61  (bilinear_interpolate)
 *       mmdet/ops/roi_align/src/roi_align_kernel.cu:63-118 (ROIAlignForward, one output element)
 *       mmdet/ops/roi_align/src/roi_align_kernel.cu:143-258 (gradient weights + scatter)
 *     The reference has no CPU implementation (mmdet/ops/roi_align/roi_align.py:27-28), and
 *     the .cu cannot be built here (needs CUDA + THC), so this restatement IS the oracle for
 *     RoIAlign; it is pinned by the geometry cases in tests/golden.
 *   oracle_nms : restatement of mmdet/ops/nms/src/nms_cpu.cpp:5-59 (greedy, `ovr >= thr`,
 *     "+1" areas, result = ascending original indices).  Pinned against the reference's own
 *     nms_cpu.cpp compiled unmodified into oracle/_ref (oracle/build_ref.py) and against the
 *     docstring known-answer mmdet/ops/nms/nms_wrapper.py:26-36.
 *
 * Build: make -C oracle   (gcc, -ffp-contract=off so float arithmetic is evaluated as written)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* roi_align_kernel.cu:16-61 */
static float bilinear_interpolate(const float* bottom_data, int height, int width, float y, float x) {
  if (y < -1.0 || y > height || x < -1.0 || x > width) return 0;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
  float lt = bottom_data[y_low * width + x_low], rt = bottom_data[y_low * width + x_high];
  float lb = bottom_data[y_high * width + x_low], rb = bottom_data[y_high * width + x_high];
  float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return (w1 * lt + w2 * rt + w3 * lb + w4 * rb);
}

/* roi_align_kernel.cu:63-118; features [B][C][H][W], rois [K][5], out [K][C][PH][PW] */
void oracle_roi_align_fwd(const float* bottom_data, const float* bottom_rois, float* top_data, int channels, int height,
                          int width, int num_rois, int pooled_height, int pooled_width, float spatial_scale,
                          int sample_num) {
  long nthreads = (long)num_rois * channels * pooled_height * pooled_width;
  for (long index = 0; index < nthreads; ++index) {
    int pw = index % pooled_width;
    int ph = (index / pooled_width) % pooled_height;
    int c = (index / pooled_width / pooled_height) % channels;
    int n = index / pooled_width / pooled_height / channels;
    const float* offset_bottom_rois = bottom_rois + n * 5;
    int roi_batch_ind = offset_bottom_rois[0];
    float roi_start_w = offset_bottom_rois[1] * spatial_scale;
    float roi_start_h = offset_bottom_rois[2] * spatial_scale;
    float roi_end_w = (offset_bottom_rois[3] + 1) * spatial_scale;
    float roi_end_h = (offset_bottom_rois[4] + 1) * spatial_scale;
    float roi_width = fmaxf(roi_end_w - roi_start_w, 0.);
    float roi_height = fmaxf(roi_end_h - roi_start_h, 0.);
    float bin_size_h = roi_height / pooled_height;
    float bin_size_w = roi_width / pooled_width;
    const float* offset_bottom_data = bottom_data + ((long)roi_batch_ind * channels + c) * height * width;
    int sample_num_h = (sample_num > 0) ? sample_num : (int)ceilf(roi_height / pooled_height);
    int sample_num_w = (sample_num > 0) ? sample_num : (int)ceilf(roi_width / pooled_width);
    float output_val = 0;
    for (int iy = 0; iy < sample_num_h; iy++) {
      const float y = roi_start_h + ph * bin_size_h + (float)(iy + .5f) * bin_size_h / (float)(sample_num_h);
      for (int ix = 0; ix < sample_num_w; ix++) {
        const float x = roi_start_w + pw * bin_size_w + (float)(ix + .5f) * bin_size_w / (float)(sample_num_w);
        output_val += bilinear_interpolate(offset_bottom_data, height, width, y, x);
      }
    }
    output_val /= (sample_num_h * sample_num_w);
    top_data[index] = output_val;
  }
}

/* roi_align_kernel.cu:143-183 + 187-258; bottom_diff [B][C][H][W] is accumulated into */
void oracle_roi_align_bwd(const float* top_diff, const float* bottom_rois, float* bottom_diff, int channels, int height,
                          int width, int num_rois, int pooled_height, int pooled_width, float spatial_scale,
                          int sample_num) {
  long nthreads = (long)num_rois * channels * pooled_height * pooled_width;
  for (long index = 0; index < nthreads; ++index) {
    int pw = index % pooled_width;
    int ph = (index / pooled_width) % pooled_height;
    int c = (index / pooled_width / pooled_height) % channels;
    int n = index / pooled_width / pooled_height / channels;
    const float* r = bottom_rois + n * 5;
    int roi_batch_ind = r[0];
    float roi_start_w = r[1] * spatial_scale, roi_start_h = r[2] * spatial_scale;
    float roi_end_w = (r[3] + 1) * spatial_scale, roi_end_h = (r[4] + 1) * spatial_scale;
    float roi_width = fmaxf(roi_end_w - roi_start_w, 0.), roi_height = fmaxf(roi_end_h - roi_start_h, 0.);
    float bin_size_h = roi_height / pooled_height, bin_size_w = roi_width / pooled_width;
    float* offset_bottom_diff = bottom_diff + ((long)roi_batch_ind * channels + c) * height * width;
    float offset_top_diff = top_diff[index];
    int sample_num_h = (sample_num > 0) ? sample_num : (int)ceilf(roi_height / pooled_height);
    int sample_num_w = (sample_num > 0) ? sample_num : (int)ceilf(roi_width / pooled_width);
    const float count = (float)(sample_num_h * sample_num_w);
    for (int iy = 0; iy < sample_num_h; iy++) {
      float y = roi_start_h + ph * bin_size_h + (float)(iy + .5f) * bin_size_h / (float)(sample_num_h);
      for (int ix = 0; ix < sample_num_w; ix++) {
        float x = roi_start_w + pw * bin_size_w + (float)(ix + .5f) * bin_size_w / (float)(sample_num_w);
        float yy = y, xx = x;
        if (yy < -1.0 || yy > height || xx < -1.0 || xx > width) continue;
        if (yy <= 0) yy = 0;
        if (xx <= 0) xx = 0;
        int y_low = (int)yy, x_low = (int)xx, y_high, x_high;
        if (y_low >= height - 1) { y_high = y_low = height - 1; yy = (float)y_low; } else { y_high = y_low + 1; }
        if (x_low >= width - 1) { x_high = x_low = width - 1; xx = (float)x_low; } else { x_high = x_low + 1; }
        float ly = yy - y_low, lx = xx - x_low, hy = 1.f - ly, hx = 1.f - lx;
        float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        offset_bottom_diff[y_low * width + x_low] += offset_top_diff * w1 / count;
        offset_bottom_diff[y_low * width + x_high] += offset_top_diff * w2 / count;
        offset_bottom_diff[y_high * width + x_low] += offset_top_diff * w3 / count;
        offset_bottom_diff[y_high * width + x_high] += offset_top_diff * w4 / count;
      }
    }
  }
}

/* nms_cpu.cpp:5-59.  order = indices sorted by score descending (supplied by the caller, who
 * uses the same stable sort in every oracle path); keep receives ascending original indices. */
int oracle_nms(const float* dets, const int64_t* order, int64_t ndets, float threshold, int64_t* keep) {
  if (ndets == 0) return 0;
  uint8_t* suppressed = (uint8_t*)calloc((size_t)ndets, 1);
  float* areas = (float*)malloc(sizeof(float) * (size_t)ndets);
  for (int64_t i = 0; i < ndets; ++i)
    areas[i] = (dets[i * 5 + 2] - dets[i * 5 + 0] + 1) * (dets[i * 5 + 3] - dets[i * 5 + 1] + 1);
  for (int64_t _i = 0; _i < ndets; _i++) {
    int64_t i = order[_i];
    if (suppressed[i] == 1) continue;
    float ix1 = dets[i * 5 + 0], iy1 = dets[i * 5 + 1], ix2 = dets[i * 5 + 2], iy2 = dets[i * 5 + 3];
    float iarea = areas[i];
    for (int64_t _j = _i + 1; _j < ndets; _j++) {
      int64_t j = order[_j];
      if (suppressed[j] == 1) continue;
      float xx1 = fmaxf(ix1, dets[j * 5 + 0]), yy1 = fmaxf(iy1, dets[j * 5 + 1]);
      float xx2 = fminf(ix2, dets[j * 5 + 2]), yy2 = fminf(iy2, dets[j * 5 + 3]);
      float w = fmaxf(0.f, xx2 - xx1 + 1), h = fmaxf(0.f, yy2 - yy1 + 1);
      float inter = w * h;
      float ovr = inter / (iarea + areas[j] - inter);
      if (ovr >= threshold) suppressed[j] = 1;
    }
  }
  int n = 0;
  for (int64_t i = 0; i < ndets; ++i)
    if (!suppressed[i]) keep[n++] = i;
  free(suppressed);
  free(areas);
  return n;
}
