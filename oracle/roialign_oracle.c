/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see roipool_oracle.c for the rules).
 *
 * CPU restatement (plain C, float32, serial) of ROIAlign with aligned=True as the Stage-3 detector uses it
 * (detectron2/detectron2/modeling/poolers.py:204-213 "ROIAlignV2" -> detectron2/detectron2/layers/roi_align.py:7-74, which
 * calls torchvision.ops.roi_align — third-party, absent from /root/reference).  The same arithmetic is stated in-tree in the
 * first detectron2 fork:
 *   forward : uwsod/detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:20-218
 *   backward: uwsod/detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:220-400
 * and that is what is followed here line by line (offset 0.5, no clamp of the ROI size to 1, sampling grid
 * ceil(roi / pooled) when sampling_ratio == 0, samples outside [-1, size] contribute 0, average over the grid).
 *
 * PARITY PINNING: restatement only (the in-tree file is a torch extension source: ATen headers + its own build).
 *
 * Layout: input NCHW float32, rois (R,5) = (batch_idx, x1, y1, x2, y2), output (R,C,PH,PW) float32.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ROIAlign_cpu.cpp:47-106 (forward) == :236-276 (backward): the four corners and weights of one sample; returns 0 if the
 * sample lies outside and contributes nothing */
static int corners(int height, int width, float y, float x, int* yl, int* xl, int* yh, int* xh, float* w) {
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return 0;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else x_high = x_low + 1;
  const float ly = y - (float)y_low, lx = x - (float)x_low;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  w[0] = hy * hx; w[1] = hy * lx; w[2] = ly * hx; w[3] = ly * lx;
  *yl = y_low; *xl = x_low; *yh = y_high; *xh = x_high;
  return 1;
}

typedef struct { float start_h, start_w, bin_h, bin_w; int grid_h, grid_w; float count; int batch; } Geom;

/* ROIAlign_cpu.cpp:137-169 */
static Geom geom(const float* roi, float scale, int ph, int pw, int sampling_ratio) {
  Geom g;
  g.batch = (int)roi[0];
  const float offset = 0.5f;
  g.start_w = roi[1] * scale - offset;
  g.start_h = roi[2] * scale - offset;
  const float end_w = roi[3] * scale - offset, end_h = roi[4] * scale - offset;
  const float rw = end_w - g.start_w, rh = end_h - g.start_h;
  g.bin_h = rh / (float)ph;
  g.bin_w = rw / (float)pw;
  g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
  g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
  const int c = g.grid_h * g.grid_w;
  g.count = (float)(c > 1 ? c : 1);
  return g;
}

void oracle_roi_align_fwd(const float* input, float spatial_scale, int channels, int height, int width, int ph_n, int pw_n,
                          int sampling_ratio, const float* rois, int n_rois, float* output) {
  for (int n = 0; n < n_rois; ++n) {
    const Geom g = geom(rois + 5 * n, spatial_scale, ph_n, pw_n, sampling_ratio);
    for (int c = 0; c < channels; ++c) {
      const float* in = input + ((long)g.batch * channels + c) * height * width;
      for (int ph = 0; ph < ph_n; ++ph)
        for (int pw = 0; pw < pw_n; ++pw) {
          float v = 0.f;
          for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
              const float x = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
              int yl, xl, yh, xh; float w[4];
              if (!corners(height, width, y, x, &yl, &xl, &yh, &xh, w)) continue;
              v += w[0] * in[yl * width + xl] + w[1] * in[yl * width + xh] + w[2] * in[yh * width + xl] + w[3] * in[yh * width + xh];
            }
          }
          output[(((long)n * channels + c) * ph_n + ph) * pw_n + pw] = v / g.count;
        }
    }
  }
}

/* ROIAlign_cpu.cpp:286-400: grad_input (N,C,H,W), zero-initialised by the caller, += w * grad_output / count */
void oracle_roi_align_bwd(const float* grad_output, float spatial_scale, int channels, int height, int width, int ph_n, int pw_n,
                          int sampling_ratio, const float* rois, int n_rois, float* grad_input) {
  for (int n = 0; n < n_rois; ++n) {
    const Geom g = geom(rois + 5 * n, spatial_scale, ph_n, pw_n, sampling_ratio);
    for (int c = 0; c < channels; ++c) {
      float* gi = grad_input + ((long)g.batch * channels + c) * height * width;
      for (int ph = 0; ph < ph_n; ++ph)
        for (int pw = 0; pw < pw_n; ++pw) {
          const float go = grad_output[(((long)n * channels + c) * ph_n + ph) * pw_n + pw];
          for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
              const float x = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
              int yl, xl, yh, xh; float w[4];
              if (!corners(height, width, y, x, &yl, &xl, &yh, &xh, w)) continue;
              gi[yl * width + xl] += go * w[0] / g.count;
              gi[yl * width + xh] += go * w[1] / g.count;
              gi[yh * width + xl] += go * w[2] / g.count;
              gi[yh * width + xh] += go * w[3] / g.count;
            }
          }
        }
    }
  }
}
