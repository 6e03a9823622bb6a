// Fused image preparation for gfx950 (next row N2): one launch turns the decoded uint8 HWC image into the
// float tensor the detector eats.  It replaces the per-image CPU chain of the reference's train/test pipeline
//   Resize -> RandomFlip -> Normalize -> Pad -> DefaultFormatBundle (+ the zero padding of mmcv's collate)
// (HBB_TOD/mmdet/datasets/pipelines/transforms.py:212-237 `_resize_img`, :437-440 `imflip`, :652-655
// `imnormalize`, :587-599 `_pad_img`; formating.py:196-203 transpose + to_tensor), whose pixel work is done by
// mmcv/OpenCV (third-party, absent from /root/reference; mmcv 1.x image/geometric.py + photometric.py over
// cv2.resize / cv2.flip / cv2.subtract / cv2.multiply / cv2.copyMakeBorder).  Published algorithm restated:
//   * cv2.resize(INTER_LINEAR) on 8-bit data is fixed point: 11-bit tap weights (cvRound of the float weight x
//     2048), a horizontal pass in int32 and the vertical pass  (((b0*(h0>>4))>>16) + ((b1*(h1>>4))>>16) + 2)>>2;
//     (an exact 2x shrink is routed to INTER_AREA: the rounded mean of each 2x2 block);
//   * the source coordinate is (float)((d + 0.5) * scale - 0.5) with scale = 1 / (dst / src) in double;
//     columns are clamped with a zero fraction, rows are clamped by index only;
//   * imnormalize: float32 pixel, BGR->RGB swap, float32 subtract of the mean, multiply by 1/std in double
//     rounded once to float32;
//   * Pad writes pad_val to the right/bottom up to (pad_h, pad_w); the batch collate pads further with 0.
// HBM-bound byte work: 3 B read + 12 B written per output pixel; a thread owns four pixels of a row = 48 contiguous
// bytes of the channels-last layout the backbone consumes, written as three 16-byte stores.
#include "pt_common.h"

namespace pt {

struct PrepGeom {
  int src_h, src_w, rs_h, rs_w, pad_h, pad_w, out_h, out_w, flip, to_rgb, normalize;
  long src_stride, sc, sh, sw;
  double scale_x, scale_y;
  float mean[3];
  double stdinv[3];
  float pad_val;
};

// tap of cv::resize's linear table for one axis: index of the first tap and its 11-bit weights
__device__ __forceinline__ void linear_tap(int d, double scale, int n, bool clamp_frac, int& s, int& w0, int& w1) {
  float f = (float)(__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5));
  s = (int)floorf(f);
  f -= (float)s;
  if (clamp_frac) {                 // columns: resize.cpp clamps the index AND zeroes the fraction
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
  }
  w0 = __float2int_rn((1.f - f) * 2048.f);
  w1 = __float2int_rn(f * 2048.f);
}

// one prepared pixel (3 channels) of the padded output at (x, y)
__device__ __forceinline__ void prep_pixel(const uint8_t* __restrict__ src, const PrepGeom& g, int x, int y, float v[3]) {
  if (y < g.rs_h && x < g.rs_w) {
    const int xr = (g.flip & 1) ? g.rs_w - 1 - x : x, yr = (g.flip & 2) ? g.rs_h - 1 - y : y;
    int p[3];
    if (g.rs_h == g.src_h && g.rs_w == g.src_w) {
      const uint8_t* s = src + (size_t)yr * g.src_stride + (size_t)xr * 3;
      p[0] = s[0]; p[1] = s[1]; p[2] = s[2];
    } else if (g.src_h == 2 * g.rs_h && g.src_w == 2 * g.rs_w) {
      // resize.cpp: an exact 2x shrink with INTER_LINEAR is routed to INTER_AREA (rounded mean of the 2x2 block)
      const uint8_t* r0 = src + (size_t)(2 * yr) * g.src_stride + (size_t)(2 * xr) * 3;
      const uint8_t* r1 = r0 + g.src_stride;
#pragma unroll
      for (int c = 0; c < 3; ++c) p[c] = (r0[c] + r0[3 + c] + r1[c] + r1[3 + c] + 2) >> 2;
    } else {
      int sx, a0, a1, sy, b0, b1;
      linear_tap(xr, g.scale_x, g.src_w, true, sx, a0, a1);
      linear_tap(yr, g.scale_y, g.src_h, false, sy, b0, b1);
      const int sx1 = min(sx + 1, g.src_w - 1);
      const int y0 = min(max(sy, 0), g.src_h - 1), y1 = min(max(sy + 1, 0), g.src_h - 1);
      const uint8_t* r0 = src + (size_t)y0 * g.src_stride;
      const uint8_t* r1 = src + (size_t)y1 * g.src_stride;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
        const int h1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
        p[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float px = (float)p[g.to_rgb ? 2 - c : c];
      v[c] = g.normalize ? (float)((double)(px - g.mean[c]) * g.stdinv[c]) : px;
    }
  } else {
    const float f = (y < g.pad_h && x < g.pad_w) ? g.pad_val : 0.f;
    v[0] = v[1] = v[2] = f;
  }
}

// A thread owns FOUR consecutive output pixels of a row.  In the channels-last layout the backbone consumes (element
// strides c = 1, w = 3) these are 12 consecutive floats: three 16-byte stores per lane, full lines per wavefront, instead
// of twelve 4-byte stores 12 bytes apart (VEC = true; the host checks width % 4 and 16-byte alignment of every row).
template <bool VEC>
__global__ void __launch_bounds__(256) image_prep_kernel(const uint8_t* __restrict__ src, PrepGeom g, float* __restrict__ dst) {
  const int nq = (g.out_w + 3) >> 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nq * g.out_h) return;
  const int y = (int)(i / nq), x0 = (int)(i - (long)y * nq) * 4;
  float v[4][3];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (VEC || x0 + k < g.out_w) prep_pixel(src, g, x0 + k, y, v[k]);
  float* o = dst + (size_t)y * g.sh + (size_t)x0 * g.sw;
  if (VEC) {
    float4* o4 = reinterpret_cast<float4*>(o);
    o4[0] = make_float4(v[0][0], v[0][1], v[0][2], v[1][0]);
    o4[1] = make_float4(v[1][1], v[1][2], v[2][0], v[2][1]);
    o4[2] = make_float4(v[2][2], v[3][0], v[3][1], v[3][2]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (x0 + k < g.out_w) {
        float* q = o + (size_t)k * g.sw;
        q[0] = v[k][0];
        q[g.sc] = v[k][1];
        q[2 * g.sc] = v[k][2];
      }
  }
}

}  // namespace pt

using namespace pt;

extern "C" int pt_image_prep(const uint8_t* src, int src_h, int src_w, int64_t src_row_stride, int channels, int rs_h,
                             int rs_w, int flip, const float* mean_host, const double* stdinv_host, int to_rgb,
                             int pad_h, int pad_w, float pad_val, int out_h, int out_w, float* dst, int64_t dst_stride_c,
                             int64_t dst_stride_h, int64_t dst_stride_w, void* stream) {
  PT_REQUIRE(src && dst, PT_EINVAL, "pt_image_prep: null buffer");
  PT_REQUIRE(channels == 3, PT_EINVAL, "pt_image_prep: %d channels (the 'color' decode flag always yields 3)", channels);
  PT_REQUIRE(src_h >= 1 && src_w >= 1 && rs_h >= 1 && rs_w >= 1, PT_EINVAL, "pt_image_prep: empty image");
  PT_REQUIRE(src_row_stride >= (int64_t)src_w * 3, PT_EINVAL, "pt_image_prep: row stride %ld below %d*3",
             (long)src_row_stride, src_w);
  PT_REQUIRE(pad_h >= rs_h && pad_w >= rs_w && out_h >= pad_h && out_w >= pad_w, PT_EINVAL,
             "pt_image_prep: need resized (%d,%d) <= padded (%d,%d) <= output (%d,%d)", rs_h, rs_w, pad_h, pad_w, out_h,
             out_w);
  PT_REQUIRE(flip >= 0 && flip <= 3, PT_EINVAL, "pt_image_prep: flip=%d (0 none, 1 horizontal, 2 vertical, 3 diagonal)", flip);
  PT_REQUIRE((mean_host == nullptr) == (stdinv_host == nullptr), PT_EINVAL, "pt_image_prep: mean and 1/std come together");
  PrepGeom g;
  g.src_h = src_h; g.src_w = src_w; g.rs_h = rs_h; g.rs_w = rs_w; g.pad_h = pad_h; g.pad_w = pad_w;
  g.out_h = out_h; g.out_w = out_w; g.flip = flip; g.to_rgb = to_rgb != 0; g.normalize = mean_host != nullptr;
  g.src_stride = (long)src_row_stride; g.sc = (long)dst_stride_c; g.sh = (long)dst_stride_h; g.sw = (long)dst_stride_w;
  g.scale_x = 1.0 / ((double)rs_w / (double)src_w);          // resize.cpp: inv_scale = dsize / ssize; scale = 1. / inv_scale
  g.scale_y = 1.0 / ((double)rs_h / (double)src_h);
  for (int c = 0; c < 3; ++c) {
    g.mean[c] = mean_host ? mean_host[c] : 0.f;
    g.stdinv[c] = stdinv_host ? stdinv_host[c] : 1.0;
  }
  g.pad_val = pad_val;
  const bool vec = dst_stride_c == 1 && dst_stride_w == 3 && out_w % 4 == 0 && dst_stride_h % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
  const long threads = (long)((out_w + 3) / 4) * out_h;
  if (vec)
    hipLaunchKernelGGL(image_prep_kernel<true>, dim3(cdiv(threads, 256)), dim3(256), 0, as_stream(stream), src, g, dst);
  else
    hipLaunchKernelGGL(image_prep_kernel<false>, dim3(cdiv(threads, 256)), dim3(256), 0, as_stream(stream), src, g, dst);
  PT_LAUNCH_CHECK("pt_image_prep");
  return PT_OK;
}
