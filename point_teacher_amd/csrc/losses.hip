// Element-wise loss kernels (focal, DIoU / DN-DIoU, box overlaps, delta decode) for gfx950.
// All are HBM/latency-bound streaming kernels: one thread per element (or per box),
// coalesced float4 box loads, no intermediate tensors (the reference materialises ~10
// temporaries per DIoU evaluation and evaluates it 10 times for DN-DIoU).
#include "pt_common.h"

namespace pt {

// ---------------------------------------------------------------- focal loss --
__device__ __forceinline__ float focal_elem(float x, float t, float gamma, float alpha, float* dldx) {
  // py_sigmoid_focal_loss, models/losses/focal_loss.py:33-39
  const float p = sigmoidf_(x);
  const float pt = (1.f - p) * t + p * (1.f - t);
  const float at = alpha * t + (1.f - alpha) * (1.f - t);
  const float ptg = (gamma == 2.f) ? pt * pt : powf(pt, gamma);
  const float fw = at * ptg;
  // binary_cross_entropy_with_logits: max(x,0) - x*t + log1p(exp(-|x|))
  const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
  if (dldx) {
    const float dpt = (1.f - 2.f * t) * p * (1.f - p);
    const float dptg = (gamma == 2.f) ? 2.f * pt : gamma * powf(pt, gamma - 1.f);
    *dldx = (p - t) * fw + bce * at * dptg * dpt;
  }
  return bce * fw;
}

constexpr int FOCAL_EPB = 1024;  // elements per block (256 threads x 4)

__global__ void __launch_bounds__(256)
    focal_fwd_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                     const float* __restrict__ weight, long total, int C, float gamma, float alpha,
                     float* __restrict__ loss, float* __restrict__ partial) {
  __shared__ float sm[17];
  float acc = 0.f;
  for (long e = (long)blockIdx.x * FOCAL_EPB + threadIdx.x; e < total; e += (long)gridDim.x * FOCAL_EPB) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = e + u * 256;
      if (i < total) {
        const long n = i / C;
        const int c = (int)(i - n * C);
        const float t = (labels[n] == c) ? 1.f : 0.f;
        float l = focal_elem(logits[i], t, gamma, alpha, nullptr);
        if (loss) loss[i] = l;
        if (weight) l *= weight[n];
        acc += l;
      }
    }
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256)
    focal_bwd_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                     const float* __restrict__ weight, const float* __restrict__ scale, long total, int C,
                     float gamma, float alpha, float* __restrict__ grad) {
  const float s = scale[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / C;
    const int c = (int)(i - n * C);
    const float t = (labels[n] == c) ? 1.f : 0.f;
    float d;
    focal_elem(logits[i], t, gamma, alpha, &d);
    if (weight) d *= weight[n];
    grad[i] = d * s;
  }
}

// ------------------------------------------------------------------ DIoU -----
__device__ __forceinline__ float sel_max(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float sel_min(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

// diou_loss for one box pair (models/losses/iou_loss.py:156-189). If g != nullptr adds
// coef * d loss / d pred to g[0..3] with torch autograd's tie conventions.
__device__ __forceinline__ float diou_one(const float4 p, const float4 t, float eps, float coef, float* g) {
  const float ltx = fmaxf(p.x, t.x), lty = fmaxf(p.y, t.y);
  const float rbx = fminf(p.z, t.z), rby = fminf(p.w, t.w);
  const float dx = rbx - ltx, dy = rby - lty;
  const float w = fmaxf(dx, 0.f), h = fmaxf(dy, 0.f);
  const float ov = w * h;
  const float pw = p.z - p.x, ph = p.w - p.y;
  const float ap = pw * ph;
  const float ag = (t.z - t.x) * (t.w - t.y);
  const float uni = ap + ag - ov + eps;
  const float iou = ov / uni;
  const float ex1 = fminf(p.x, t.x), ey1 = fminf(p.y, t.y);
  const float ex2 = fmaxf(p.z, t.z), ey2 = fmaxf(p.w, t.w);
  const float ewx = ex2 - ex1, ewy = ey2 - ey1;
  const float cw = fmaxf(ewx, 0.f), ch = fmaxf(ewy, 0.f);
  const float c2 = cw * cw + ch * ch + eps;
  const float sx = (t.x + t.z) - (p.x + p.z), sy = (t.y + t.w) - (p.y + p.w);
  const float rho2 = sx * sx / 4.f + sy * sy / 4.f;
  const float loss = 1.f - (iou - rho2 / c2);
  if (g) {
    const float g_iou = -coef;
    const float g_rho2 = coef / c2;
    const float g_c2 = -coef * rho2 / (c2 * c2);
    const float g_uni = -g_iou * ov / (uni * uni);
    const float g_ap = g_uni;
    const float g_ov = g_iou / uni - g_uni;
    const float g_dx = g_ov * h * (dx >= 0.f ? 1.f : 0.f);
    const float g_dy = g_ov * w * (dy >= 0.f ? 1.f : 0.f);
    const float g_ex = g_c2 * 2.f * cw * (ewx >= 0.f ? 1.f : 0.f);
    const float g_ey = g_c2 * 2.f * ch * (ewy >= 0.f ? 1.f : 0.f);
    // x1: ltx=max(px1,tx1) (-g_dx), ap (-ph), ex1=min(px1,tx1) (-g_ex), rho2
    g[0] += -g_dx * sel_max(p.x, t.x) - g_ap * ph - g_ex * sel_min(p.x, t.x) - g_rho2 * sx / 2.f;
    g[1] += -g_dy * sel_max(p.y, t.y) - g_ap * pw - g_ey * sel_min(p.y, t.y) - g_rho2 * sy / 2.f;
    g[2] += g_dx * sel_min(p.z, t.z) + g_ap * ph + g_ex * sel_max(p.z, t.z) - g_rho2 * sx / 2.f;
    g[3] += g_dy * sel_min(p.w, t.w) + g_ap * pw + g_ey * sel_max(p.w, t.w) - g_rho2 * sy / 2.f;
  }
  return loss;
}

// target shifted as DN_diou_loss does (iou_loss.py:419-426)
__device__ __forceinline__ float4 dn_shift(const float4 t, float anx, int i, int j) {
  const float w = t.z - t.x, h = t.w - t.y;
  float4 r;
  r.x = t.x - anx * w * (float)i;
  r.z = t.z + anx * w * (float)j;
  r.y = t.y - anx * h * (float)i;
  r.w = t.w + anx * h * (float)j;
  return r;
}

__global__ void __launch_bounds__(256)
    diou_fwd_kernel(const float4* __restrict__ pred, const float4* __restrict__ target, int N, float eps,
                    float hyper, float* __restrict__ diou, float* __restrict__ dnmin) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float4 p = pred[n], t = target[n];
  if (diou) diou[n] = diou_one(p, t, eps, 0.f, nullptr);
  if (dnmin) {
    const float anx = hyper / 2.f;
    float m = INFINITY;
#pragma unroll
    for (int i = -1; i <= 1; ++i)
#pragma unroll
      for (int j = -1; j <= 1; ++j) {
        const float l = diou_one(p, dn_shift(t, anx, i, j), eps, 0.f, nullptr);
        m = (l < m) ? l : m;   // NaN never replaces (torch.min would propagate; weights mask those rows)
      }
    dnmin[n] = m;
  }
}

__global__ void __launch_bounds__(256)
    diou_bwd_kernel(const float4* __restrict__ pred, const float4* __restrict__ target,
                    const float* __restrict__ g_diou, const float* __restrict__ g_dn, int N, float eps,
                    float hyper, float4* __restrict__ grad) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float4 p = pred[n], t = target[n];
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  if (g_diou) {
    const float c = g_diou[n];
    if (c != 0.f) diou_one(p, t, eps, c, g);
  }
  if (g_dn) {
    const float c = g_dn[n];
    if (c != 0.f) {
      const float anx = hyper / 2.f;
      float m = INFINITY;
      int bi = 0, bj = 0;
#pragma unroll
      for (int i = -1; i <= 1; ++i)
#pragma unroll
        for (int j = -1; j <= 1; ++j) {
          const float l = diou_one(p, dn_shift(t, anx, i, j), eps, 0.f, nullptr);
          if (l < m) { m = l; bi = i; bj = j; }
        }
      diou_one(p, dn_shift(t, anx, bi, bj), eps, c, g);
    }
  }
  grad[n] = make_float4(g[0], g[1], g[2], g[3]);
}

// ------------------------------------------------------------ bbox_overlaps --
__device__ __forceinline__ float overlap_one(const float4 a, const float4 b, int mode, float eps) {
  const float a1 = (a.z - a.x) * (a.w - a.y), a2 = (b.z - b.x) * (b.w - b.y);
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
  const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float ov = w * h;
  float uni = (mode == 1) ? a1 : (a1 + a2 - ov);
  uni = fmaxf(uni, eps);
  const float iou = ov / uni;
  if (mode != 2) return iou;
  const float ew = fmaxf(fmaxf(a.z, b.z) - fminf(a.x, b.x), 0.f);
  const float eh = fmaxf(fmaxf(a.w, b.w) - fminf(a.y, b.y), 0.f);
  const float ea = fmaxf(ew * eh, eps);
  return iou - (ea - uni) / ea;
}

__global__ void overlaps_aligned_kernel(const float4* __restrict__ a, const float4* __restrict__ b, int M,
                                        int mode, float eps, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) out[i] = overlap_one(a[i], b[i], mode, eps);
}

__global__ void overlaps_pairwise_kernel(const float4* __restrict__ a, const float4* __restrict__ b, int M, int N,
                                         int mode, float eps, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * N) return;
  const int m = (int)(i / N), n = (int)(i - (long)m * N);
  out[i] = overlap_one(a[m], b[n], mode, eps);
}

// ------------------------------------------------------------ MaxIoUAssigner (row N4) --
// core/bbox/assigners/max_iou_assigner.py:98-212 (assign + assign_wrt_overlaps) for the anchor-based baselines
// (configs/baselines/aitodv2_retinanet_r50_1x.py).  The reference materialises overlaps[G, A] (A ~ 120 000 anchors, G ~ 300:
// 144 MB per image) and walks it four times; here the matrix is never stored: pass 1 computes each anchor's best box and,
// with one packed 64-bit atomicMax per (box, workgroup), each box's best anchor; pass 2 re-evaluates the same IoU (same
// instruction sequence -> same bits) for the "every anchor that ties a box's maximum" rule.  Boxes go through LDS in tiles.
constexpr int MI_TILE = 256;

// overlap_one(g, a, iou) with the division skipped for disjoint boxes: 0 / union is exactly 0, so the value is the same
__device__ __forceinline__ float iou_or_zero(const float4 g, const float4 a, float eps) {
  const float w = fminf(g.z, a.z) - fmaxf(g.x, a.x), h = fminf(g.w, a.w) - fmaxf(g.y, a.y);
  if (!(w > 0.f && h > 0.f)) return 0.f;
  const float ov = w * h;
  const float uni = fmaxf((g.z - g.x) * (g.w - g.y) + (a.z - a.x) * (a.w - a.y) - ov, eps);
  return ov / uni;
}

__global__ void __launch_bounds__(256)
    max_iou_pass1_kernel(const float4* __restrict__ anchors, int A, const float4* __restrict__ gts, const int32_t* __restrict__ off,
                         float eps, float* __restrict__ max_ov, int32_t* __restrict__ argmax,
                         unsigned long long* __restrict__ gt_best) {
  __shared__ float4 tile[MI_TILE];
  __shared__ unsigned long long tile_best[MI_TILE];
  const int b = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
  const int g0 = off[b], G = off[b + 1] - g0;
  const bool live = a < A;
  const float4 an = live ? anchors[a] : make_float4(0.f, 0.f, 0.f, 0.f);
  float best = -1.f;
  int bi = 0;
  for (int base = 0; base < G; base += MI_TILE) {
    const int n = min(MI_TILE, G - base);
    __syncthreads();
    if ((int)threadIdx.x < n) {
      tile[threadIdx.x] = gts[g0 + base + threadIdx.x];
      tile_best[threadIdx.x] = 0ull;
    }
    __syncthreads();
    if (live)
      for (int k = 0; k < n; ++k) {
        const float iou = iou_or_zero(tile[k], an, eps);
        if (iou > best) { best = iou; bi = base + k; }               // first box on ties (overlaps.max(dim=0))
        if (iou > 0.f) {                                             // disjoint pairs (almost all) never touch the per-box maximum:
          // key: IoU bits (> 0, monotonic as an integer) above the complemented anchor index -> max = highest IoU, lowest index.
          // A box no anchor overlaps keeps key 0 = "maximum 0 at anchor 0", which is what overlaps.max(dim=1) returns for it.
          const unsigned long long key = ((unsigned long long)__float_as_uint(iou) << 32) | (0xFFFFFFFFu - (unsigned)a);
          if (key > tile_best[k]) atomicMax(&tile_best[k], key);
        }
      }
    __syncthreads();
    if ((int)threadIdx.x < n && tile_best[threadIdx.x]) atomicMax(&gt_best[g0 + base + threadIdx.x], tile_best[threadIdx.x]);
  }
  if (live) {
    max_ov[(size_t)b * A + a] = G ? best : 0.f;
    argmax[(size_t)b * A + a] = bi;
  }
}

__global__ void __launch_bounds__(256)
    max_iou_pass2_kernel(const float4* __restrict__ anchors, int A, const float4* __restrict__ gts, const int32_t* __restrict__ off,
                         float eps, const float* __restrict__ max_ov, const int32_t* __restrict__ argmax,
                         const unsigned long long* __restrict__ gt_best, float pos_thr, float neg_lo, float neg_hi, float min_pos,
                         int match_low_quality, int assign_all, int32_t* __restrict__ assigned) {
  __shared__ float4 tile[MI_TILE];
  __shared__ unsigned long long tile_best[MI_TILE];
  const int b = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
  const int g0 = off[b], G = off[b + 1] - g0;
  const bool live = a < A;
  const float4 an = live ? anchors[a] : make_float4(0.f, 0.f, 0.f, 0.f);
  int res = G ? -1 : 0;                                               // no box in the image: everything is background
  if (live && G) {
    const float mo = max_ov[(size_t)b * A + a];
    if (mo >= neg_lo && mo < neg_hi) res = 0;
    if (mo >= pos_thr) res = argmax[(size_t)b * A + a] + 1;
  }
  if (match_low_quality)
    for (int base = 0; base < G; base += MI_TILE) {
      const int n = min(MI_TILE, G - base);
      __syncthreads();
      if ((int)threadIdx.x < n) {
        tile[threadIdx.x] = gts[g0 + base + threadIdx.x];
        tile_best[threadIdx.x] = gt_best[g0 + base + threadIdx.x];
      }
      __syncthreads();
      if (live)
        for (int k = 0; k < n; ++k) {                                 // ascending box index: a later box overrides (the python loop)
          const unsigned long long kb = tile_best[k];
          const float gmax = __uint_as_float((unsigned)(kb >> 32));
          if (!(gmax >= min_pos)) continue;
          if (assign_all) {
            if (iou_or_zero(tile[k], an, eps) == gmax) res = base + k + 1;
          } else if ((unsigned)a == (kb ? 0xFFFFFFFFu - (unsigned)(kb & 0xFFFFFFFFull) : 0u)) {
            res = base + k + 1;
          }
        }
    }
  if (live) assigned[(size_t)b * A + a] = res;
}

// --------------------------------------------------------------- delta2bbox --
__global__ void delta2bbox_kernel(const float4* __restrict__ rois, const float4* __restrict__ deltas,
                                  const float4* __restrict__ gout, int N, float max_h, float max_w,
                                  float max_ratio, float4* __restrict__ out, float4* __restrict__ gdelta) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float4 r = rois[n], d = deltas[n];
  const float px = (r.x + r.z) * 0.5f, py = (r.y + r.w) * 0.5f, pw = r.z - r.x, ph = r.w - r.y;
  const float dw = fminf(fmaxf(d.z, -max_ratio), max_ratio), dh = fminf(fmaxf(d.w, -max_ratio), max_ratio);
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  const float gx = px + pw * d.x, gy = py + ph * d.y;
  float b[4] = {gx - gw * 0.5f, gy - gh * 0.5f, gx + gw * 0.5f, gy + gh * 0.5f};
  float pass[4] = {1.f, 1.f, 1.f, 1.f};
  if (max_h > 0.f && max_w > 0.f) {
    const float mx[4] = {max_w, max_h, max_w, max_h};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (b[k] < 0.f) { b[k] = 0.f; pass[k] = 0.f; }
      if (b[k] > mx[k]) { b[k] = mx[k]; pass[k] = 0.f; }
    }
  }
  if (out) out[n] = make_float4(b[0], b[1], b[2], b[3]);
  if (gdelta) {
    const float4 go = gout[n];
    const float g0 = go.x * pass[0], g1 = go.y * pass[1], g2 = go.z * pass[2], g3 = go.w * pass[3];
    const float ggx = g0 + g2, ggy = g1 + g3, ggw = 0.5f * (g2 - g0), ggh = 0.5f * (g3 - g1);
    const float inw = (d.z >= -max_ratio && d.z <= max_ratio) ? 1.f : 0.f;
    const float inh = (d.w >= -max_ratio && d.w <= max_ratio) ? 1.f : 0.f;
    gdelta[n] = make_float4(ggx * pw, ggy * ph, ggw * gw * inw, ggh * gh * inh);
  }
}

}  // namespace pt

using namespace pt;

extern "C" int pt_focal_nblocks(int N, int C) {
  long total = (long)N * C;
  int nb = cdiv(total, FOCAL_EPB);
  return nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
}

extern "C" int pt_sigmoid_focal_loss_fwd(const float* logits, const int32_t* labels, const float* weight, int N, int C,
                                         float gamma, float alpha, float* loss, float* partial, void* stream) {
  PT_REQUIRE(N >= 0 && C > 0 && partial, PT_EINVAL, "pt_sigmoid_focal_loss_fwd: bad argument");
  const int nb = pt_focal_nblocks(N, C);
  PT_REQUIRE(N == 0 || (logits && labels), PT_EINVAL, "pt_sigmoid_focal_loss_fwd: NULL input");
  hipLaunchKernelGGL(focal_fwd_kernel, dim3(nb), dim3(256), 0, as_stream(stream), logits, labels, weight, (long)N * C,
                     C, gamma, alpha, loss, partial);
  PT_LAUNCH_CHECK("pt_sigmoid_focal_loss_fwd");
  return PT_OK;
}

extern "C" int pt_sigmoid_focal_loss_bwd(const float* logits, const int32_t* labels, const float* weight,
                                         const float* scale, int N, int C, float gamma, float alpha, float* grad,
                                         void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(logits && labels && scale && grad && N > 0 && C > 0, PT_EINVAL, "pt_sigmoid_focal_loss_bwd: bad argument");
  const long total = (long)N * C;
  int nb = cdiv(total, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(nb), dim3(256), 0, as_stream(stream), logits, labels, weight, scale, total,
                     C, gamma, alpha, grad);
  PT_LAUNCH_CHECK("pt_sigmoid_focal_loss_bwd");
  return PT_OK;
}

extern "C" int pt_diou_fwd(const float* pred, const float* target, int N, float eps, float hyper, float* diou,
                           float* dnmin, void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(pred && target && N > 0 && (diou || dnmin), PT_EINVAL, "pt_diou_fwd: bad argument");
  hipLaunchKernelGGL(diou_fwd_kernel, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(pred), reinterpret_cast<const float4*>(target), N, eps, hyper,
                     diou, dnmin);
  PT_LAUNCH_CHECK("pt_diou_fwd");
  return PT_OK;
}

extern "C" int pt_diou_bwd(const float* pred, const float* target, const float* g_diou, const float* g_dn, int N,
                           float eps, float hyper, float* grad_pred, void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(pred && target && grad_pred && N > 0, PT_EINVAL, "pt_diou_bwd: bad argument");
  hipLaunchKernelGGL(diou_bwd_kernel, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(pred), reinterpret_cast<const float4*>(target), g_diou, g_dn, N,
                     eps, hyper, reinterpret_cast<float4*>(grad_pred));
  PT_LAUNCH_CHECK("pt_diou_bwd");
  return PT_OK;
}

extern "C" int pt_bbox_overlaps_aligned(const float* a, const float* b, int M, int mode, float eps, float* out,
                                        void* stream) {
  if (M == 0) return PT_OK;
  PT_REQUIRE(a && b && out && M > 0 && mode >= 0 && mode <= 2, PT_EINVAL, "pt_bbox_overlaps_aligned: bad argument");
  hipLaunchKernelGGL(overlaps_aligned_kernel, dim3(cdiv(M, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), M, mode, eps, out);
  PT_LAUNCH_CHECK("pt_bbox_overlaps_aligned");
  return PT_OK;
}

extern "C" int pt_bbox_overlaps_pairwise(const float* a, const float* b, int M, int N, int mode, float eps, float* out,
                                         void* stream) {
  if (M == 0 || N == 0) return PT_OK;
  PT_REQUIRE(a && b && out && M > 0 && N > 0 && mode >= 0 && mode <= 2, PT_EINVAL,
             "pt_bbox_overlaps_pairwise: bad argument");
  hipLaunchKernelGGL(overlaps_pairwise_kernel, dim3(cdiv((long)M * N, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), M, N, mode, eps, out);
  PT_LAUNCH_CHECK("pt_bbox_overlaps_pairwise");
  return PT_OK;
}

extern "C" int pt_delta2bbox_fwd(const float* rois, const float* deltas, int N, float max_h, float max_w,
                                 float wh_ratio_clip, float* out, void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(rois && deltas && out && N > 0 && wh_ratio_clip > 0.f, PT_EINVAL, "pt_delta2bbox_fwd: bad argument");
  hipLaunchKernelGGL(delta2bbox_kernel, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(rois), reinterpret_cast<const float4*>(deltas),
                     (const float4*)nullptr, N, max_h, max_w, fabsf(logf(wh_ratio_clip)),
                     reinterpret_cast<float4*>(out), (float4*)nullptr);
  PT_LAUNCH_CHECK("pt_delta2bbox_fwd");
  return PT_OK;
}

extern "C" int pt_delta2bbox_bwd(const float* rois, const float* deltas, const float* grad_out, int N, float max_h,
                                 float max_w, float wh_ratio_clip, float* grad_deltas, void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(rois && deltas && grad_out && grad_deltas && N > 0 && wh_ratio_clip > 0.f, PT_EINVAL,
             "pt_delta2bbox_bwd: bad argument");
  hipLaunchKernelGGL(delta2bbox_kernel, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(rois), reinterpret_cast<const float4*>(deltas),
                     reinterpret_cast<const float4*>(grad_out), N, max_h, max_w, fabsf(logf(wh_ratio_clip)),
                     (float4*)nullptr, reinterpret_cast<float4*>(grad_deltas));
  PT_LAUNCH_CHECK("pt_delta2bbox_bwd");
  return PT_OK;
}

extern "C" int pt_max_iou_assign(const float* anchors, int A, const float* gt_boxes, const int32_t* off, int B, float pos_iou_thr,
                                 float neg_iou_lo, float neg_iou_hi, float min_pos_iou, int match_low_quality,
                                 int gt_max_assign_all, float* max_overlaps, int32_t* argmax_ws, uint64_t* gt_best_ws,
                                 int32_t* assigned_gt_inds, void* stream) {
  PT_REQUIRE(anchors && off && max_overlaps && argmax_ws && assigned_gt_inds && A > 0 && B > 0, PT_EINVAL,
             "pt_max_iou_assign: bad argument");
  PT_REQUIRE(B <= 65535, PT_ELIMIT, "pt_max_iou_assign: B=%d above 65535", B);
  PT_REQUIRE((reinterpret_cast<uintptr_t>(anchors) & 15) == 0 && (!gt_boxes || (reinterpret_cast<uintptr_t>(gt_boxes) & 15) == 0),
             PT_EINVAL, "pt_max_iou_assign: boxes must be 16-byte aligned");
  const dim3 grid(cdiv(A, 256), B);
  hipLaunchKernelGGL(max_iou_pass1_kernel, grid, dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(anchors), A,
                     reinterpret_cast<const float4*>(gt_boxes), off, 1e-6f, max_overlaps, argmax_ws,
                     reinterpret_cast<unsigned long long*>(gt_best_ws));
  PT_LAUNCH_CHECK("pt_max_iou_assign (pass 1)");
  hipLaunchKernelGGL(max_iou_pass2_kernel, grid, dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(anchors), A,
                     reinterpret_cast<const float4*>(gt_boxes), off, 1e-6f, max_overlaps, argmax_ws,
                     reinterpret_cast<const unsigned long long*>(gt_best_ws), pos_iou_thr, neg_iou_lo, neg_iou_hi, min_pos_iou,
                     match_low_quality, gt_max_assign_all, assigned_gt_inds);
  PT_LAUNCH_CHECK("pt_max_iou_assign (pass 2)");
  return PT_OK;
}
