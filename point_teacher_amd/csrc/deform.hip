// (Modulated) deformable convolution for gfx950 - replaces mmcv.ops.DeformConv2d / ModulatedDeformConv2d
// (`conv_cfg=dict(type='DCNv2')`, reachable through `dcn_on_last_conv=True`:
// HBB_TOD/mmdet/models/dense_heads/anchor_free_head.py:101-102,121-122, fcos_head_p2b_ts.py:197-198).
//
// The contraction stays on the matrix cores through the vendor GEMM (north_star: "MFMA used only for the conv
// contractions"); what is hand-written is the data-dependent part around it:
//   deform_im2col        x, offset, mask      -> col[B, C*K, L]     (bilinear gather, one thread per (c, l): K taps)
//   deform_col2im        dcol, offset, mask   -> dx                 (scatter with f32 atomics)
//   deform_col2im_coord  dcol, x, offset, mask-> doffset, dmask     (one thread per offset element)
// Sampling rule of the published algorithm (mmcv modulated_deform_conv_cuda_kernel.cuh, dmcn_im2col_bilinear):
// a sample outside (-1, H) x (-1, W) is 0; inside, the four neighbours are weighted bilinearly and neighbours
// outside the map contribute 0.  Offsets are (dy, dx) interleaved per kernel tap, per deformable group.
#include "pt_common.h"

namespace pt {

struct DeformGeom {
  int B, C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, dg, Ho, Wo;
};

__device__ __forceinline__ float dcn_bilinear(const float* __restrict__ im, int H, int W, float h, float w) {
  if (!(h > -1.f && w > -1.f && h < (float)H && w < (float)W)) return 0.f;
  const int hl = (int)floorf(h), wl = (int)floorf(w), hh = hl + 1, wh = wl + 1;
  const float lh = h - (float)hl, lw = w - (float)wl, uh = 1.f - lh, uw = 1.f - lw;
  const float v1 = (hl >= 0 && wl >= 0) ? im[hl * W + wl] : 0.f;
  const float v2 = (hl >= 0 && wh <= W - 1) ? im[hl * W + wh] : 0.f;
  const float v3 = (hh <= H - 1 && wl >= 0) ? im[hh * W + wl] : 0.f;
  const float v4 = (hh <= H - 1 && wh <= W - 1) ? im[hh * W + wh] : 0.f;
  return uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4;
}

__global__ void __launch_bounds__(256)
    deform_im2col_kernel(const float* __restrict__ x, const float* __restrict__ offset, const float* __restrict__ mask,
                         DeformGeom g, float* __restrict__ col) {
  const int L = g.Ho * g.Wo, K = g.kh * g.kw;
  const long total = (long)g.B * g.C * L;
  const int cpg = g.C / g.dg;                                   // channels per deformable group
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int l = (int)(i % L), c = (int)((i / L) % g.C), b = (int)(i / ((long)L * g.C));
    const int ho = l / g.Wo, wo = l - ho * g.Wo, d = c / cpg;
    const float* im = x + ((size_t)b * g.C + c) * g.H * g.W;
    const float* off = offset + ((size_t)b * g.dg + d) * 2 * K * L;
    const float* mk = mask ? mask + ((size_t)b * g.dg + d) * K * L : nullptr;
    float* cp = col + ((size_t)b * g.C + c) * K * L + l;
    for (int t = 0; t < K; ++t) {
      const int ki = t / g.kw, kj = t - ki * g.kw;
      const float h = (float)(ho * g.sh - g.ph + ki * g.dh) + off[(size_t)(2 * t) * L + l];
      const float w = (float)(wo * g.sw - g.pw + kj * g.dw) + off[(size_t)(2 * t + 1) * L + l];
      float v = dcn_bilinear(im, g.H, g.W, h, w);
      if (mk) v *= mk[(size_t)t * L + l];
      cp[(size_t)t * L] = v;
    }
  }
}

__global__ void __launch_bounds__(256)
    deform_col2im_kernel(const float* __restrict__ dcol, const float* __restrict__ offset, const float* __restrict__ mask,
                         DeformGeom g, float* __restrict__ dx) {
  const int L = g.Ho * g.Wo, K = g.kh * g.kw;
  const long total = (long)g.B * g.C * K * L;
  const int cpg = g.C / g.dg;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int l = (int)(i % L), t = (int)((i / L) % K), c = (int)((i / ((long)L * K)) % g.C);
    const int b = (int)(i / ((long)L * K * g.C));
    const int ho = l / g.Wo, wo = l - ho * g.Wo, d = c / cpg, ki = t / g.kw, kj = t - ki * g.kw;
    const float* off = offset + ((size_t)b * g.dg + d) * 2 * K * L;
    const float h = (float)(ho * g.sh - g.ph + ki * g.dh) + off[(size_t)(2 * t) * L + l];
    const float w = (float)(wo * g.sw - g.pw + kj * g.dw) + off[(size_t)(2 * t + 1) * L + l];
    if (!(h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W)) continue;
    float gv = dcol[i];
    if (mask) gv *= mask[(((size_t)b * g.dg + d) * K + t) * L + l];
    if (gv == 0.f) continue;
    const int hl = (int)floorf(h), wl = (int)floorf(w), hh = hl + 1, wh = wl + 1;
    const float lh = h - (float)hl, lw = w - (float)wl, uh = 1.f - lh, uw = 1.f - lw;
    float* im = dx + ((size_t)b * g.C + c) * g.H * g.W;
    if (hl >= 0 && wl >= 0) atomicAdd(&im[hl * g.W + wl], gv * uh * uw);
    if (hl >= 0 && wh <= g.W - 1) atomicAdd(&im[hl * g.W + wh], gv * uh * lw);
    if (hh <= g.H - 1 && wl >= 0) atomicAdd(&im[hh * g.W + wl], gv * lh * uw);
    if (hh <= g.H - 1 && wh <= g.W - 1) atomicAdd(&im[hh * g.W + wh], gv * lh * lw);
  }
}

// One thread per (b, deformable group, tap, l): sums over the channels of the group.
__global__ void __launch_bounds__(256)
    deform_col2im_coord_kernel(const float* __restrict__ dcol, const float* __restrict__ x,
                               const float* __restrict__ offset, const float* __restrict__ mask, DeformGeom g,
                               float* __restrict__ doffset, float* __restrict__ dmask) {
  const int L = g.Ho * g.Wo, K = g.kh * g.kw;
  const long total = (long)g.B * g.dg * K * L;
  const int cpg = g.C / g.dg;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int l = (int)(i % L), t = (int)((i / L) % K), d = (int)((i / ((long)L * K)) % g.dg);
    const int b = (int)(i / ((long)L * K * g.dg));
    const int ho = l / g.Wo, wo = l - ho * g.Wo, ki = t / g.kw, kj = t - ki * g.kw;
    const float* off = offset + ((size_t)b * g.dg + d) * 2 * K * L;
    const float h = (float)(ho * g.sh - g.ph + ki * g.dh) + off[(size_t)(2 * t) * L + l];
    const float w = (float)(wo * g.sw - g.pw + kj * g.dw) + off[(size_t)(2 * t + 1) * L + l];
    const float m = mask ? mask[(((size_t)b * g.dg + d) * K + t) * L + l] : 1.f;
    float gh = 0.f, gw = 0.f, gm = 0.f;
    const bool inside = h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
    if (inside) {
      const int hl = (int)floorf(h), wl = (int)floorf(w), hh = hl + 1, wh = wl + 1;
      const float lh = h - (float)hl, lw = w - (float)wl, uh = 1.f - lh, uw = 1.f - lw;
      const bool a1 = hl >= 0 && wl >= 0, a2 = hl >= 0 && wh <= g.W - 1, a3 = hh <= g.H - 1 && wl >= 0,
                 a4 = hh <= g.H - 1 && wh <= g.W - 1;
      for (int cc = 0; cc < cpg; ++cc) {
        const int c = d * cpg + cc;
        const float* im = x + ((size_t)b * g.C + c) * g.H * g.W;
        const float v1 = a1 ? im[hl * g.W + wl] : 0.f, v2 = a2 ? im[hl * g.W + wh] : 0.f;
        const float v3 = a3 ? im[hh * g.W + wl] : 0.f, v4 = a4 ? im[hh * g.W + wh] : 0.f;
        const float gv = dcol[(((size_t)b * g.C + c) * K + t) * L + l];
        gh += gv * m * (uw * (v3 - v1) + lw * (v4 - v2));
        gw += gv * m * (uh * (v2 - v1) + lh * (v4 - v3));
        gm += gv * (uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4);
      }
    }
    doffset[(((size_t)b * g.dg + d) * 2 * K + 2 * t) * L + l] = gh;
    doffset[(((size_t)b * g.dg + d) * 2 * K + 2 * t + 1) * L + l] = gw;
    if (dmask) dmask[(((size_t)b * g.dg + d) * K + t) * L + l] = gm;
  }
}

}  // namespace pt

using namespace pt;

static int deform_geom(const char* fn, int B, int C, int H, int W, int kh, int kw, int ph, int pw, int sh, int sw, int dh,
                       int dw, int dg, DeformGeom* g) {
  PT_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && dg > 0 &&
                 ph >= 0 && pw >= 0,
             PT_EINVAL, "%s: bad size", fn);
  PT_REQUIRE(C % dg == 0, PT_EINVAL, "%s: C=%d not divisible by deform_groups=%d", fn, C, dg);
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1, Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  PT_REQUIRE(Ho > 0 && Wo > 0, PT_EINVAL, "%s: empty output", fn);
  *g = DeformGeom{B, C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, dg, Ho, Wo};
  return PT_OK;
}

static int blocks_for(long total) {
  long nb = (total + 255) / 256;
  return (int)(nb > 65536 ? 65536 : (nb < 1 ? 1 : nb));
}

extern "C" int pt_deform_im2col(const float* x, const float* offset, const float* mask, int B, int C, int H, int W,
                                int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                                int deform_groups, float* col, void* stream) {
  PT_REQUIRE(x && offset && col, PT_EINVAL, "pt_deform_im2col: NULL pointer");
  DeformGeom g;
  int rc = deform_geom("pt_deform_im2col", B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deform_groups, &g);
  if (rc) return rc;
  hipLaunchKernelGGL(deform_im2col_kernel, dim3(blocks_for((long)B * C * g.Ho * g.Wo)), dim3(256), 0, as_stream(stream), x,
                     offset, mask, g, col);
  PT_LAUNCH_CHECK("pt_deform_im2col");
  return PT_OK;
}

extern "C" int pt_deform_col2im(const float* grad_col, const float* offset, const float* mask, int B, int C, int H, int W,
                                int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                                int deform_groups, float* grad_x, void* stream) {
  PT_REQUIRE(grad_col && offset && grad_x, PT_EINVAL, "pt_deform_col2im: NULL pointer");
  DeformGeom g;
  int rc = deform_geom("pt_deform_col2im", B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deform_groups, &g);
  if (rc) return rc;
  hipLaunchKernelGGL(deform_col2im_kernel, dim3(blocks_for((long)B * C * kh * kw * g.Ho * g.Wo)), dim3(256), 0,
                     as_stream(stream), grad_col, offset, mask, g, grad_x);
  PT_LAUNCH_CHECK("pt_deform_col2im");
  return PT_OK;
}

extern "C" int pt_deform_col2im_coord(const float* grad_col, const float* x, const float* offset, const float* mask, int B,
                                      int C, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                                      int dil_h, int dil_w, int deform_groups, float* grad_offset, float* grad_mask,
                                      void* stream) {
  PT_REQUIRE(grad_col && x && offset && grad_offset && (!grad_mask || mask), PT_EINVAL, "pt_deform_col2im_coord: NULL pointer");
  DeformGeom g;
  int rc = deform_geom("pt_deform_col2im_coord", B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                       deform_groups, &g);
  if (rc) return rc;
  hipLaunchKernelGGL(deform_col2im_coord_kernel, dim3(blocks_for((long)B * deform_groups * kh * kw * g.Ho * g.Wo)), dim3(256),
                     0, as_stream(stream), grad_col, x, offset, mask, g, grad_offset, grad_mask);
  PT_LAUNCH_CHECK("pt_deform_col2im_coord");
  return PT_OK;
}
