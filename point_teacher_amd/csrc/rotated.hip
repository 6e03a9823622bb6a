// Oriented-box kernels for the OBB variant of the path (OBB_TOD/mmrotate, config 5):
//   * differentiable rotated IoU  (replaces mmcv.ops.diff_iou_rotated_2d, call sites
//     OBB_TOD/mmrotate/models/losses/rotated_iou_loss.py:47,90)
//   * RoIAlignRotated fwd/bwd      (replaces mmcv.ops.RoIAlignRotated, call site
//     OBB_TOD/mmrotate/models/roi_heads/roi_extractors/rotate_single_level_roi_extractor.py:126)
// The IoU gradient is obtained by running the same convex-clip code on forward-mode dual numbers
// (value + 5 partials w.r.t. the predicted box), one thread per box pair: no sort kernel, no
// intermediate vertex tensors (mmcv materialises [B,N,24,2] vertices and a sort index).
#include "pt_common.h"

namespace pt {

// ------------------------------------------------------------------ dual numbers --
template <int ND>
struct Dual {
  float v;
  float d[ND];
  __device__ Dual() {}
  __device__ Dual(float x) : v(x) {
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = 0.f;
  }
};
template <int ND>
__device__ __forceinline__ Dual<ND> operator+(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
template <int ND>
__device__ __forceinline__ Dual<ND> operator-(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
template <int ND>
__device__ __forceinline__ Dual<ND> operator*(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
template <int ND>
__device__ __forceinline__ Dual<ND> operator/(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r; const float inv = 1.f / b.v; r.v = a.v * inv;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ __forceinline__ float val(float x) { return x; }
template <int ND>
__device__ __forceinline__ float val(const Dual<ND>& x) { return x.v; }

template <typename T>
struct Pt {
  T x, y;
};

// Intersection area of the convex quad `p1` (type T: float or dual) with the float quad `p2`
// (counter-clockwise in image axes), Sutherland-Hodgman + shoelace.
template <typename T>
__device__ T clip_area(const Pt<T>* p1, const Pt<float>* p2) {
  Pt<T> A[10], Bf[10];
  int n = 4;
  for (int i = 0; i < 4; ++i) A[i] = p1[i];
  Pt<T>* cur = A;
  Pt<T>* nxt = Bf;
  for (int e = 0; e < 4 && n > 0; ++e) {
    const Pt<float> a = p2[e], b = p2[(e + 1) & 3];
    const float ex = b.x - a.x, ey = b.y - a.y;
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const Pt<T> p = cur[i], q = cur[(i + 1 == n) ? 0 : i + 1];
      const T sp = (p.y - T(a.y)) * T(ex) - (p.x - T(a.x)) * T(ey);
      const T sq = (q.y - T(a.y)) * T(ex) - (q.x - T(a.x)) * T(ey);
      if (val(sp) >= 0.f) nxt[m++] = p;
      if ((val(sp) > 0.f && val(sq) < 0.f) || (val(sp) < 0.f && val(sq) > 0.f)) {
        const T t = sp / (sp - sq);
        nxt[m].x = p.x + t * (q.x - p.x);
        nxt[m].y = p.y + t * (q.y - p.y);
        ++m;
      }
    }
    n = m;
    Pt<T>* tmp = cur; cur = nxt; nxt = tmp;
  }
  T area(0.f);
  for (int i = 0; i < n; ++i) {
    const Pt<T> p = cur[i], q = cur[(i + 1 == n) ? 0 : i + 1];
    area = area + (p.x * q.y - q.x * p.y);
  }
  return area * T(0.5f);
}

template <typename T>
__device__ __forceinline__ void corners(const T cx, const T cy, const T w, const T h, const T c, const T s, Pt<T>* p) {
  const T hw = w * T(0.5f), hh = h * T(0.5f);
  const float sx[4] = {-1.f, 1.f, 1.f, -1.f}, sy[4] = {-1.f, -1.f, 1.f, 1.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const T dx = hw * T(sx[i]), dy = hh * T(sy[i]);
    p[i].x = cx + dx * c - dy * s;
    p[i].y = cy + dx * s + dy * c;
  }
}

__global__ void diff_iou_rotated_kernel(const float* __restrict__ b1, const float* __restrict__ b2, int N,
                                        const float* __restrict__ gout, float* __restrict__ iou,
                                        float* __restrict__ grad1) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* a = b1 + (size_t)n * 5;
  const float* b = b2 + (size_t)n * 5;
  // box 2 relative to box 1's centre (fp32 precision), float corners
  Pt<float> q[4];
  corners<float>(b[0] - a[0], b[1] - a[1], b[2], b[3], cosf(b[4]), sinf(b[4]), q);
  const float a2 = b[2] * b[3];
  if (!grad1) {
    Pt<float> p[4];
    corners<float>(0.f, 0.f, a[2], a[3], cosf(a[4]), sinf(a[4]), p);
    const float inter = fabsf(clip_area<float>(p, q));
    const float a1 = a[2] * a[3];
    iou[n] = (a1 < 1e-14f || a2 < 1e-14f) ? 0.f : inter / (a1 + a2 - inter);
    return;
  }
  typedef Dual<5> D;
  D cx(0.f), cy(0.f), w(a[2]), h(a[3]), c(cosf(a[4])), s(sinf(a[4]));
  cx.d[0] = 1.f; cy.d[1] = 1.f; w.d[2] = 1.f; h.d[3] = 1.f;
  c.d[4] = -sinf(a[4]); s.d[4] = cosf(a[4]);
  Pt<D> p[4];
  corners<D>(cx, cy, w, h, c, s, p);
  D inter = clip_area<D>(p, q);
  if (inter.v < 0.f) inter = D(0.f) - inter;
  const D a1 = w * h;
  const D io = inter / (a1 + D(a2) - inter);
  const float g = gout[n];
  const bool dead = (a1.v < 1e-14f || a2 < 1e-14f);
#pragma unroll
  for (int k = 0; k < 5; ++k) grad1[(size_t)n * 5 + k] = dead ? 0.f : g * io.d[k];
}

// ------------------------------------------------------------ RoIAlignRotated ----
struct RRoi {
  int b;
  float cw, ch, start_w, start_h, bin_w, bin_h, cosv, sinv, count;
  int grid_w, grid_h;
};

__device__ __forceinline__ RRoi rroi_geom(const float* __restrict__ r, int out_size, float scale, int sample_num,
                                          int aligned, int clockwise, int B) {
  RRoi g;
  g.b = min(max((int)r[0], 0), B - 1);
  const float off = aligned ? 0.5f : 0.f;
  g.cw = r[1] * scale - off;
  g.ch = r[2] * scale - off;
  float rw = r[3] * scale, rh = r[4] * scale;
  float theta = r[5];
  if (clockwise) theta = -theta;
  if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  g.bin_h = rh / (float)out_size;
  g.bin_w = rw / (float)out_size;
  g.grid_h = sample_num > 0 ? sample_num : (int)ceilf(rh / (float)out_size);
  g.grid_w = sample_num > 0 ? sample_num : (int)ceilf(rw / (float)out_size);
  g.start_h = -rh / 2.f;
  g.start_w = -rw / 2.f;
  g.cosv = cosf(theta);
  g.sinv = sinf(theta);
  g.count = fmaxf((float)(g.grid_h * g.grid_w), 1.f);
  return g;
}

struct Tap4 {
  int y0, y1, x0, x1;
  float w1, w2, w3, w4;
  bool valid;
};
__device__ __forceinline__ Tap4 tap4(float y, float x, int H, int W) {
  Tap4 r;
  r.valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
  const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
  r.y0 = yl; r.y1 = yh; r.x0 = xl; r.x1 = xh;
  r.w1 = hy * hx; r.w2 = hy * lx; r.w3 = ly * hx; r.w4 = ly * lx;
  return r;
}

// Generic over the two layouts through element strides (sc, sy, sx) and the thread->element map:
// CL = 1: consecutive threads are consecutive channels (coalesced [B,H,W,C] reads / atomics).
template <bool BWD, bool CL>
__global__ void __launch_bounds__(256)
    roi_align_rotated_kernel(const float* __restrict__ src, const float* __restrict__ rois, int B, long total, int C,
                             int H, int W, int out_size, float scale, int sample_num, int aligned, int clockwise,
                             float* __restrict__ dst) {
  const long sb = (long)C * H * W;
  const long sc = CL ? 1 : (long)H * W, sy = CL ? (long)W * C : W, sx = CL ? C : 1;
  const int bins = out_size * out_size;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c, bin, k;
    if (CL) { c = (int)(i % C); bin = (int)((i / C) % bins); k = (int)(i / ((long)C * bins)); }
    else { bin = (int)(i % bins); c = (int)((i / bins) % C); k = (int)(i / ((long)C * bins)); }
    const int ph = bin / out_size, pw = bin - ph * out_size;
    const RRoi g = rroi_geom(rois + (size_t)k * 6, out_size, scale, sample_num, aligned, clockwise, B);
    const long oidx = ((long)k * C + c) * bins + bin;            // [K,C,out,out]
    const long base = (long)g.b * sb + (long)c * sc;
    float acc = 0.f;
    const float gv = BWD ? src[oidx] / g.count : 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float yy = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const float xx = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
        const float y = yy * g.cosv - xx * g.sinv + g.ch;
        const float x = yy * g.sinv + xx * g.cosv + g.cw;
        const Tap4 t = tap4(y, x, H, W);
        if (!t.valid) continue;
        const long o1 = base + t.y0 * sy + t.x0 * sx, o2 = base + t.y0 * sy + t.x1 * sx;
        const long o3 = base + t.y1 * sy + t.x0 * sx, o4 = base + t.y1 * sy + t.x1 * sx;
        if (BWD) {
          atomicAdd(&dst[o1], gv * t.w1); atomicAdd(&dst[o2], gv * t.w2);
          atomicAdd(&dst[o3], gv * t.w3); atomicAdd(&dst[o4], gv * t.w4);
        } else {
          acc += t.w1 * src[o1] + t.w2 * src[o2] + t.w3 * src[o3] + t.w4 * src[o4];
        }
      }
    }
    if (!BWD) dst[oidx] = acc / g.count;
  }
}

}  // namespace pt

using namespace pt;

extern "C" int pt_diff_iou_rotated_fwd(const float* boxes1, const float* boxes2, int N, float* iou, void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(boxes1 && boxes2 && iou && N > 0, PT_EINVAL, "pt_diff_iou_rotated_fwd: bad argument");
  hipLaunchKernelGGL(diff_iou_rotated_kernel, dim3(cdiv(N, 64)), dim3(64), 0, as_stream(stream), boxes1, boxes2, N,
                     (const float*)nullptr, iou, (float*)nullptr);
  PT_LAUNCH_CHECK("pt_diff_iou_rotated_fwd");
  return PT_OK;
}

extern "C" int pt_diff_iou_rotated_bwd(const float* boxes1, const float* boxes2, const float* grad_iou, int N,
                                       float* grad_boxes1, void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(boxes1 && boxes2 && grad_iou && grad_boxes1 && N > 0, PT_EINVAL, "pt_diff_iou_rotated_bwd: bad argument");
  hipLaunchKernelGGL(diff_iou_rotated_kernel, dim3(cdiv(N, 64)), dim3(64), 0, as_stream(stream), boxes1, boxes2, N,
                     grad_iou, (float*)nullptr, grad_boxes1);
  PT_LAUNCH_CHECK("pt_diff_iou_rotated_bwd");
  return PT_OK;
}

template <bool BWD>
static int rroi_launch(const char* fn, const float* src, const float* rois, int B, int C, int H, int W, int K,
                       int out_size, float scale, int sample_num, int aligned, int clockwise, int channels_last,
                       float* dst, void* stream) {
  if (K == 0) return PT_OK;
  PT_REQUIRE(src && rois && dst && B > 0 && C > 0 && H > 0 && W > 0 && K > 0 && out_size >= 1, PT_EINVAL,
             "%s: bad argument", fn);
  const long total = (long)K * C * out_size * out_size;
  int nb = cdiv(total, 256);
  if (nb > 65536) nb = 65536;
  hipStream_t s = as_stream(stream);
  if (channels_last)
    hipLaunchKernelGGL((roi_align_rotated_kernel<BWD, true>), dim3(nb), dim3(256), 0, s, src, rois, B, total, C, H, W,
                       out_size, scale, sample_num, aligned, clockwise, dst);
  else
    hipLaunchKernelGGL((roi_align_rotated_kernel<BWD, false>), dim3(nb), dim3(256), 0, s, src, rois, B, total, C, H, W,
                       out_size, scale, sample_num, aligned, clockwise, dst);
  PT_LAUNCH_CHECK(fn);
  return PT_OK;
}

extern "C" int pt_roi_align_rotated_fwd(const float* feat, const float* rois, int B, int C, int H, int W, int K,
                                        int out_size, float spatial_scale, int sample_num, int aligned, int clockwise,
                                        int channels_last, float* out, void* stream) {
  return rroi_launch<false>("pt_roi_align_rotated_fwd", feat, rois, B, C, H, W, K, out_size, spatial_scale, sample_num,
                            aligned, clockwise, channels_last, out, stream);
}

extern "C" int pt_roi_align_rotated_bwd(const float* grad_out, const float* rois, int B, int C, int H, int W, int K,
                                        int out_size, float spatial_scale, int sample_num, int aligned, int clockwise,
                                        int channels_last, float* grad_feat, void* stream) {
  return rroi_launch<true>("pt_roi_align_rotated_bwd", grad_out, rois, B, C, H, W, K, out_size, spatial_scale,
                           sample_num, aligned, clockwise, channels_last, grad_feat, stream);
}
