/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the
 * product path (sos-wsod_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * CPU restatement (plain C, float32, serial) of max ROI pooling as the
 * reference states it in-tree:
 *   forward : uwsod/projects/WSL/wsl/layers/csrc/ROILoopPool/ROILoopPool_cpu.cpp:13-79
 *   backward: uwsod/projects/WSL/wsl/layers/csrc/ROILoopPool/ROILoopPool_cpu.cpp:81-123
 * which is the arithmetic of torchvision==0.7.0 ops.RoIPool (the op the hot
 * path actually calls, uwsod/projects/WSL/wsl/modeling/poolers.py:183-186,267-270;
 * torchvision is a third-party dependency absent from /root/reference).
 *
 * PARITY PINNING: the reference file above cannot be compiled in this image
 * (it includes <TH/TH.h>, which torch 2.10 no longer ships; providing a
 * stand-in header is not allowed), so this stage is pinned by restatement
 * only ("parity unpinned" for the ROIPool stage).
 *
 * Layout: input NCHW float32, rois (R,5) = (batch_idx, x1, y1, x2, y2),
 * output (R,C,PH,PW) float32, argmax (R,C,PH,PW) int32 (h*W+w, or -1).
 */
#include <float.h>
#include <math.h>
#include <string.h>

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ROILoopPool_cpu.cpp:13-79 */
void oracle_roi_pool_fwd(const float* input, float spatial_scale, int channels,
                         int height, int width, int pooled_height,
                         int pooled_width, const float* rois, int num_rois,
                         float* output, int* argmax_data) {
  for (int n = 0; n < num_rois; ++n) {
    const float* r = rois + n * 5;
    int roi_batch_ind = (int)r[0];
    /* :29-32  round() of the float product, half away from zero */
    int roi_start_w = (int)roundf(r[1] * spatial_scale);
    int roi_start_h = (int)roundf(r[2] * spatial_scale);
    int roi_end_w = (int)roundf(r[3] * spatial_scale);
    int roi_end_h = (int)roundf(r[4] * spatial_scale);
    /* :35-38 malformed ROIs forced to 1x1; float bin size */
    int roi_width = imax(roi_end_w - roi_start_w + 1, 1);
    int roi_height = imax(roi_end_h - roi_start_h + 1, 1);
    float bin_size_h = (float)roi_height / (float)pooled_height;
    float bin_size_w = (float)roi_width / (float)pooled_width;
    for (int ph = 0; ph < pooled_height; ++ph) {
      for (int pw = 0; pw < pooled_width; ++pw) {
        /* :42-45 */
        int hstart = (int)floorf((float)ph * bin_size_h);
        int wstart = (int)floorf((float)pw * bin_size_w);
        int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
        int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
        /* :48-52 */
        hstart = imin(imax(hstart + roi_start_h, 0), height);
        hend = imin(imax(hend + roi_start_h, 0), height);
        wstart = imin(imax(wstart + roi_start_w, 0), width);
        wend = imin(imax(wend + roi_start_w, 0), width);
        int is_empty = (hend <= hstart) || (wend <= wstart);
        for (int c = 0; c < channels; ++c) {
          /* :56-58 empty bin -> 0 / -1 */
          float maxval = is_empty ? 0.0f : -FLT_MAX;
          int maxidx = -1;
          const float* in =
              input + ((long)roi_batch_ind * channels + c) * height * width;
          /* :63-71 strict '>' : first max in row-major scan order wins */
          for (int h = hstart; h < hend; ++h)
            for (int w = wstart; w < wend; ++w) {
              int idx = h * width + w;
              if (in[idx] > maxval) { maxval = in[idx]; maxidx = idx; }
            }
          long o = (((long)n * channels + c) * pooled_height + ph) * pooled_width + pw;
          output[o] = maxval;
          argmax_data[o] = maxidx;
        }
      }
    }
  }
}

/* ROILoopPool_cpu.cpp:81-123 ; grad_input must be zero-filled by the caller
 * (the reference does at::zeros, ROILoopPool_cpu.cpp:176-177). */
void oracle_roi_pool_bwd(const float* grad_output, const int* argmax_data,
                         int num_rois, int channels, int height, int width,
                         int pooled_height, int pooled_width,
                         float* grad_input, const float* rois) {
  for (int n = 0; n < num_rois; ++n) {
    int roi_batch_ind = (int)rois[n * 5];
    for (int c = 0; c < channels; ++c) {
      float* gi = grad_input + ((long)roi_batch_ind * channels + c) * height * width;
      long base = ((long)n * channels + c) * pooled_height * pooled_width;
      for (int i = 0; i < pooled_height * pooled_width; ++i) {
        int a = argmax_data[base + i];
        if (a != -1) gi[a] += grad_output[base + i];
      }
    }
  }
}
