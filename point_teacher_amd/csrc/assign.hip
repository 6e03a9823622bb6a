// Point -> ground-truth assignment, FCOS target build and pseudo-box fusion for gfx950.
//
// One 256-thread workgroup per ground truth: the [P,G] cost matrices the reference
// materialises (3 x 12 MB per image) never exist; each workgroup streams the P points
// (80 KB, L2-resident), keeps a per-thread sorted top-k of (distance, index) keys in
// registers, merges them with k workgroup-wide 64-bit min reductions and publishes the
// result with integer atomicMax ("later gt wins" == max gt index), which is
// deterministic.  HBM traffic is 8*(P+G) bytes read + 4*P written per image.
#include <stdarg.h>
#include <string.h>

#include "pt_common.h"

namespace pt {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

constexpr int KMAX = 8;
constexpr unsigned long long KEY_INF = ~0ull;

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
  unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), o, 64);
  unsigned hi = __shfl_xor((unsigned)(v >> 32), o, 64);
  return ((unsigned long long)hi << 32) | lo;
}

// Sorted insertion into a register-resident ascending list (static indices only).
__device__ __forceinline__ void insert_key(unsigned long long (&best)[KMAX], unsigned long long key) {
  if (key >= best[KMAX - 1]) return;
#pragma unroll
  for (int i = KMAX - 1; i > 0; --i) {
    const unsigned long long prev = best[i - 1];
    best[i] = (key < prev) ? prev : ((key < best[i]) ? key : best[i]);
  }
  best[0] = (key < best[0]) ? key : best[0];
}

// The k L1-nearest points of one gt (ties -> lowest point index); result in rows_sm[0..k).
// dist = (|px-gx| + |py-gy|) * w, bit-identical to PointCost (match_cost.py:206-210,214).
__device__ void nearest_k(const float* __restrict__ points, int P, float gx, float gy, float w, int k,
                          unsigned long long* red_sm, int* rows_sm) {
  unsigned long long best[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) best[i] = KEY_INF;
  const float2* pts = reinterpret_cast<const float2*>(points);
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    const float2 q = pts[p];
    float d = fabsf(q.x - gx) + fabsf(q.y - gy);
    d = d * w;
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)p;
    insert_key(best, key);
  }
  // lists longer than k are irrelevant: cap by pushing INF beyond k
#pragma unroll
  for (int i = 0; i < KMAX; ++i)
    if (i >= k) best[i] = KEY_INF;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = 0; r < k; ++r) {
    unsigned long long m = best[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long t = shfl_xor_u64(m, o);
      m = t < m ? t : m;
    }
    __syncthreads();
    if (lane == 0) red_sm[wv] = m;
    __syncthreads();
    unsigned long long g = red_sm[0];
    for (int i = 1; i < nw; ++i) g = red_sm[i] < g ? red_sm[i] : g;
    if (best[0] == g && g != KEY_INF) {  // unique owner (keys are unique): pop
#pragma unroll
      for (int i = 0; i < KMAX - 1; ++i) best[i] = best[i + 1];
      best[KMAX - 1] = KEY_INF;
    }
    if (threadIdx.x == 0) rows_sm[r] = (g == KEY_INF) ? -1 : (int)(g & 0xffffffffu);
  }
  __syncthreads();
}

__device__ __forceinline__ int image_of(const int32_t* off, int B, int g) {
  int b = 0;
  while (b + 1 < B && g >= off[b + 1]) ++b;
  return b;
}

__global__ void __launch_bounds__(256) topk_assign_kernel(const float* __restrict__ points, int P,
                                                          const float* __restrict__ gt_xy,
                                                          const uint8_t* __restrict__ gt_valid,
                                                          const int32_t* __restrict__ off, int B, int k,
                                                          int32_t* __restrict__ gt_inds,
                                                          int32_t* __restrict__ cand) {
  __shared__ unsigned long long red_sm[4];
  __shared__ int rows_sm[KMAX];
  const int g = blockIdx.x;
  if (gt_valid && !gt_valid[g]) {  // workgroup-uniform exit
    if (cand && threadIdx.x < k) cand[(size_t)g * k + threadIdx.x] = -1;
    return;
  }
  const int b = image_of(off, B, g);
  const int local = g - off[b];
  nearest_k(points, P, gt_xy[2 * g], gt_xy[2 * g + 1], 1.0f, k, red_sm, rows_sm);
  if (threadIdx.x < k) {
    const int row = rows_sm[threadIdx.x];
    if (cand) cand[(size_t)g * k + threadIdx.x] = row;
    if (row >= 0) atomicMax(&gt_inds[(size_t)b * P + row], local + 1);
  }
}

// FocalLossCost for one logit (match_cost.py:92-98)
__device__ __forceinline__ float focal_cost(float x, float alpha, float gamma_is2, float eps) {
  const float p = sigmoidf_(x);
  const float q = 1.0f - p;
  const float neg = -logf(q + eps) * (1.0f - alpha) * (p * p);
  const float pos = -logf(p + eps) * alpha * (q * q);
  return pos - neg;
}

// DEC = false: `reg` holds (l,t,r,b) distances [B*P,4] that are decoded with `points` (HBB head).
// DEC = true : `reg` holds DECODED oriented boxes (cx,cy,w,h,a) [B*P,5]; InsiderCost reads the first
//              four columns as an axis-aligned cxcywh box and ignores the angle (match_cost.py:235-241),
//              which is what the OBB head feeds it (rotated_fcos_head_p2rb_ts.py:883-885).
template <bool DEC>
__global__ void __launch_bounds__(256)
    fuse_assign_kernel(const float* __restrict__ points, int P, const float* __restrict__ reg,
                       const float* __restrict__ cls, int C, const float* __restrict__ gt_xy,
                       const int32_t* __restrict__ gt_labels, const int32_t* __restrict__ off, int B, int k,
                       int topk, float cls_w, float reg_w, float loc_w, int32_t* __restrict__ gt_inds,
                       int32_t* __restrict__ cand) {
  __shared__ unsigned long long red_sm[4];
  __shared__ int rows_sm[KMAX];
  __shared__ float box_sm[KMAX][4];
  __shared__ float fc_sm[KMAX][64];
  __shared__ unsigned mask_sm;
  const int g = blockIdx.x;
  const int b = image_of(off, B, g);
  const int g0 = off[b], G = off[b + 1] - g0;
  const int local = g - g0;
  nearest_k(points, P, gt_xy[2 * g], gt_xy[2 * g + 1], reg_w, k, red_sm, rows_sm);
  if (threadIdx.x == 0) mask_sm = 0u;
  unsigned mask = 0u;
  if (k <= topk) {
    mask = (1u << k) - 1u;
  } else {
    // per candidate: decoded box exactly as distance2bbox -> xyxy_to_cxcywh -> InsiderCost do
    if (threadIdx.x < k && rows_sm[threadIdx.x] >= 0) {
      const int row = rows_sm[threadIdx.x];
      float cx, cy, w, h;
      if (DEC) {
        const float* d = reg + ((size_t)b * P + row) * 5;
        cx = d[0]; cy = d[1]; w = d[2]; h = d[3];
      } else {
        const float px = points[2 * row], py = points[2 * row + 1];
        const float* d = reg + ((size_t)b * P + row) * 4;
        const float x1 = px - d[0], y1 = py - d[1], x2 = px + d[2], y2 = py + d[3];
        cx = (x1 + x2) / 2; cy = (y1 + y2) / 2; w = x2 - x1; h = y2 - y1;
      }
      box_sm[threadIdx.x][0] = cx - w / 2;
      box_sm[threadIdx.x][1] = cy - h / 2;
      box_sm[threadIdx.x][2] = cx + w / 2;
      box_sm[threadIdx.x][3] = cy + h / 2;
    }
    for (int t = threadIdx.x; t < k * C; t += blockDim.x) {
      const int r = t / C, c = t % C;
      const int row = rows_sm[r];
      fc_sm[r][c] = row >= 0 ? focal_cost(cls[((size_t)b * P + row) * C + c], 0.25f, 2.f, 1e-12f) * cls_w : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < G; j += blockDim.x) {
      const float gx = gt_xy[2 * (g0 + j)], gy = gt_xy[2 * (g0 + j) + 1];
      const int lab = gt_labels[g0 + j];
      float cost[KMAX];
#pragma unroll
      for (int r = 0; r < KMAX; ++r) {
        if (r < k) {
          const bool in = (gx >= box_sm[r][0]) & (gx <= box_sm[r][2]) & (gy >= box_sm[r][1]) & (gy <= box_sm[r][3]);
          cost[r] = fc_sm[r][lab] + (in ? 0.f : 1.f) * loc_w;
        } else {
          cost[r] = 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < KMAX; ++r) {
        if (r < k) {
          int rank = 0;
#pragma unroll
          for (int s = 0; s < KMAX; ++s)
            if (s < k) rank += (cost[s] < cost[r]) || (cost[s] == cost[r] && s < r);
          if (rank < topk) mask |= 1u << r;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mask |= __shfl_xor(mask, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicOr(&mask_sm, mask);
    __syncthreads();
    mask = mask_sm;
  }
  if (threadIdx.x < k) {
    const int row = rows_sm[threadIdx.x];
    cand[(size_t)g * k + threadIdx.x] = row;
    if (row >= 0 && ((mask >> threadIdx.x) & 1u)) atomicMax(&gt_inds[(size_t)b * P + row], local + 1);
  }
}

__global__ void pseudo_boxes_kernel(const float* __restrict__ points, int P, const float* __restrict__ reg,
                                    const float* __restrict__ cls, int C, const float* __restrict__ gt_xy,
                                    const int32_t* __restrict__ gt_labels, const float* __restrict__ gt_bboxes,
                                    const int32_t* __restrict__ off, int B, int sumG, int k,
                                    const int32_t* __restrict__ gt_inds, const int32_t* __restrict__ cand,
                                    float* __restrict__ pb, float* __restrict__ pp, float* __restrict__ ps,
                                    int32_t* __restrict__ nass, float* __restrict__ iou) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= sumG) return;
  const int b = image_of(off, B, g);
  const int local = g - off[b];
  const int lab = gt_labels[g];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, ss = 0.f;
  int n = 0;
  for (int r = 0; r < k; ++r) {
    const int row = cand[(size_t)g * k + r];
    if (row < 0) continue;
    const size_t q = (size_t)b * P + row;
    if (gt_inds[q] != local + 1) continue;
    const float sc = sigmoidf_(cls[q * C + lab]);
    const float px = points[2 * row], py = points[2 * row + 1];
    const float* d = reg + q * 4;
    s0 += (px - d[0]) * sc;
    s1 += (py - d[1]) * sc;
    s2 += (px + d[2]) * sc;
    s3 += (py + d[3]) * sc;
    ss += sc;
    ++n;
  }
  const float gx = gt_xy[2 * g], gy = gt_xy[2 * g + 1];
  float x1, y1, x2, y2, cx, cy, score, io = 0.f;
  if (n > 0) {
    x1 = s0 / ss; y1 = s1 / ss; x2 = s2 / ss; y2 = s3 / ss;
    score = ss / (float)n;
    cx = (x1 + x2) / 2; cy = (y1 + y2) / 2;
    if (gt_bboxes) {
      const float* t = gt_bboxes + (size_t)g * 4;
      const float a1 = (x2 - x1) * (y2 - y1), a2 = (t[2] - t[0]) * (t[3] - t[1]);
      const float w = fmaxf(fminf(x2, t[2]) - fmaxf(x1, t[0]), 0.f);
      const float h = fmaxf(fminf(y2, t[3]) - fmaxf(y1, t[1]), 0.f);
      const float ov = w * h;
      io = ov / fmaxf(a1 + a2 - ov, 1e-6f);
    }
  } else {
    x1 = gx - 0.5f * 8.f; y1 = gy - 0.5f * 8.f; x2 = gx + 0.5f * 8.f; y2 = gy + 0.5f * 8.f;
    score = 0.f; cx = gx; cy = gy;
  }
  pb[4 * g] = x1; pb[4 * g + 1] = y1; pb[4 * g + 2] = x2; pb[4 * g + 3] = y2;
  pp[2 * g] = cx; pp[2 * g + 1] = cy;
  ps[g] = score;
  nass[g] = n;
  if (iou) iou[g] = io;
}

__global__ void fcos_targets_kernel(const float* __restrict__ points, int P, const int32_t* __restrict__ gt_inds,
                                    const float* __restrict__ boxes, const int32_t* __restrict__ box_labels,
                                    const int32_t* __restrict__ off, int B, int num_classes,
                                    int32_t* __restrict__ labels, float* __restrict__ tg, float* __restrict__ ctr) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * P) return;
  const int b = (int)(i / P), p = (int)(i % P);
  const int g0 = off[b], G = off[b + 1] - g0;
  const int gi = gt_inds[i];
  float l = 0.f, t = 0.f, r = 0.f, bt = 0.f, c = 0.f;
  int lab = num_classes;
  if (G > 0) {
    if (gi > 0) lab = box_labels ? box_labels[g0 + gi - 1] : 0;
    if (boxes) {
      const int idx = gi > 0 ? gi - 1 : 0;
      const float* bx = boxes + (size_t)(g0 + idx) * 4;
      const float px = points[2 * p], py = points[2 * p + 1];
      l = px - bx[0]; t = py - bx[1]; r = bx[2] - px; bt = bx[3] - py;
      if (gi > 0) {
        const float a = fmaxf(fminf(l, r), 0.01f) / fmaxf(l, r);
        const float d = fmaxf(fminf(t, bt), 0.01f) / fmaxf(t, bt);
        c = sqrtf(a * d);
      }
    }
  }
  labels[i] = lab;
  if (tg) { tg[4 * i] = l; tg[4 * i + 1] = t; tg[4 * i + 2] = r; tg[4 * i + 3] = bt; }
  if (ctr) ctr[i] = c;
}

// _gnerate_pseudo_single of the OBB head (rotated_fcos_head_p2rb_ts.py:899-917): score-weighted mean of
// the DECODED (cx,cy,w,h,a) boxes of the points assigned to each gt - the angle is averaged like any
// other column - or (gx, gy, 8, 8, 0) when no point was assigned.
__global__ void pseudo_boxes_obb_kernel(const float* __restrict__ dec, int P, const float* __restrict__ cls, int C,
                                        const float* __restrict__ gt_xy, const int32_t* __restrict__ gt_labels,
                                        const int32_t* __restrict__ off, int B, int sumG, int k,
                                        const int32_t* __restrict__ gt_inds, const int32_t* __restrict__ cand,
                                        float* __restrict__ pb, float* __restrict__ pp, float* __restrict__ ps,
                                        int32_t* __restrict__ nass) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= sumG) return;
  const int b = image_of(off, B, g);
  const int local = g - off[b];
  const int lab = gt_labels[g];
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, ss = 0.f;
  int n = 0;
  for (int r = 0; r < k; ++r) {
    const int row = cand[(size_t)g * k + r];
    if (row < 0) continue;
    const size_t q = (size_t)b * P + row;
    if (gt_inds[q] != local + 1) continue;
    const float sc = sigmoidf_(cls[q * C + lab]);
    const float* d = dec + q * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) s[c] += d[c] * sc;
    ss += sc;
    ++n;
  }
  float o[5];
  float score = 0.f;
  if (n > 0) {
#pragma unroll
    for (int c = 0; c < 5; ++c) o[c] = s[c] / ss;
    score = ss / (float)n;
  } else {
    o[0] = gt_xy[2 * g]; o[1] = gt_xy[2 * g + 1]; o[2] = 8.f; o[3] = 8.f; o[4] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < 5; ++c) pb[5 * (size_t)g + c] = o[c];
  pp[2 * g] = o[0]; pp[2 * g + 1] = o[1];
  ps[g] = score;
  nass[g] = n;
}

// _get_target_single / _get_target_pseudo_single of the OBB head (:671-716, :781-843): the (l,t,r,b)
// distances of every point in the frame of its ASSIGNED oriented box (box 0 of the image when unassigned,
// as `inds * 0` does), that box's angle, the label and the centerness target (:1118-1138).
__global__ void fcos_targets_obb_kernel(const float* __restrict__ points, int P, const int32_t* __restrict__ gt_inds,
                                        const float* __restrict__ boxes, const int32_t* __restrict__ box_labels,
                                        const int32_t* __restrict__ off, int B, int num_classes,
                                        int32_t* __restrict__ labels, float* __restrict__ tg,
                                        float* __restrict__ ang, float* __restrict__ ctr) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * P) return;
  const int b = (int)(i / P), p = (int)(i % P);
  const int g0 = off[b], G = off[b + 1] - g0;
  const int gi = gt_inds[i];
  float l = 0.f, t = 0.f, r = 0.f, bt = 0.f, a = 0.f, c = 0.f;
  int lab = num_classes;
  if (G > 0) {
    if (gi > 0) lab = box_labels ? box_labels[g0 + gi - 1] : 0;
    const int idx = gi > 0 ? gi - 1 : 0;
    const float* bx = boxes + (size_t)(g0 + idx) * 5;
    a = bx[4];
    const float ca = cosf(a), sa = sinf(a);
    const float dx = points[2 * p] - bx[0], dy = points[2 * p + 1] - bx[1];
    const float ox = ca * dx + sa * dy, oy = -sa * dx + ca * dy;
    l = bx[2] / 2 + ox; r = bx[2] / 2 - ox; t = bx[3] / 2 + oy; bt = bx[3] / 2 - oy;
    if (gi > 0) {
      const float u = fmaxf(fminf(l, r), 0.01f) / fmaxf(l, r);
      const float v = fmaxf(fminf(t, bt), 0.01f) / fmaxf(t, bt);
      c = sqrtf(u * v);
    }
  }
  labels[i] = lab;
  tg[4 * i] = l; tg[4 * i + 1] = t; tg[4 * i + 2] = r; tg[4 * i + 3] = bt;
  ang[i] = a;
  ctr[i] = c;
}


// ------------------------------------------------------------ dense FCOS targets (row N4) --
// FCOSHead._get_target_single (HBB_TOD/mmdet/models/dense_heads/fcos_head.py:877-1007) for the supervised FCOS baseline
// (configs/baselines/aitodv2_fcos_r50_1x.py): every point takes the SMALLEST-AREA ground truth whose (centre-sampled)
// box contains it and whose largest side distance lies in the level's regress range; none -> background.  The reference
// materialises five [P, G] float tensors per image; here one thread owns a point and streams the image's boxes through
// LDS (256 per tile), keeping the running minimum in registers.  Ties keep the first box, an all-INF row keeps box 0
// (what `areas.min(dim=1)` returns), so the unused targets of background points match as well.
struct DenseGt {
  float x1, y1, x2, y2, area;
  int label;
};

__global__ void __launch_bounds__(256)
    fcos_dense_targets_kernel(const float* __restrict__ points, const float* __restrict__ ranges,
                              const float* __restrict__ radius, const float* __restrict__ norm, int P,
                              const float* __restrict__ boxes, const int32_t* __restrict__ box_labels,
                              const int32_t* __restrict__ off, int num_classes, int center_sampling,
                              int32_t* __restrict__ labels, float* __restrict__ bbox_targets, float* __restrict__ ctr_target) {
  __shared__ DenseGt tile[256];
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  const int g0 = off[b], G = off[b + 1] - g0;
  const bool live = p < P;
  const float x = live ? points[2 * p] : 0.f, y = live ? points[2 * p + 1] : 0.f;
  const float lo = live ? ranges[2 * p] : 0.f, hi = live ? ranges[2 * p + 1] : 0.f;
  const float rad = live ? radius[p] : 0.f;
  constexpr float INF = 1e8f;
  float best = INF, bl = 0.f, bt = 0.f, br = 0.f, bb = 0.f;
  int bidx = -1;
  for (int base = 0; base < G; base += 256) {
    const int n = min(256, G - base);
    __syncthreads();
    if ((int)threadIdx.x < n) {
      const float* q = boxes + (size_t)(g0 + base + threadIdx.x) * 4;
      DenseGt t;
      t.x1 = q[0]; t.y1 = q[1]; t.x2 = q[2]; t.y2 = q[3];
      t.area = (q[2] - q[0]) * (q[3] - q[1]);
      t.label = box_labels[g0 + base + threadIdx.x];
      tile[threadIdx.x] = t;
    }
    __syncthreads();
    if (!live) continue;
    for (int k = 0; k < n; ++k) {
      const DenseGt t = tile[k];
      const float l = x - t.x1, r = t.x2 - x, tp = y - t.y1, bo = t.y2 - y;
      bool inside;
      if (center_sampling) {
        const float cx = (t.x1 + t.x2) / 2.f, cy = (t.y1 + t.y2) / 2.f;
        const float xmin = cx - rad, ymin = cy - rad, xmax = cx + rad, ymax = cy + rad;
        const float c0 = xmin > t.x1 ? xmin : t.x1, c1 = ymin > t.y1 ? ymin : t.y1;
        const float c2 = xmax > t.x2 ? t.x2 : xmax, c3 = ymax > t.y2 ? t.y2 : ymax;
        inside = fminf(fminf(x - c0, y - c1), fminf(c2 - x, c3 - y)) > 0.f;
      } else {
        inside = fminf(fminf(l, tp), fminf(r, bo)) > 0.f;
      }
      const float maxd = fmaxf(fmaxf(l, tp), fmaxf(r, bo));
      const float area = (inside && maxd >= lo && maxd <= hi) ? t.area : INF;
      if (bidx < 0 || area < best) {               // the first box initialises (all-INF rows keep index 0)
        bl = l; bt = tp; br = r; bb = bo;
        best = area;
        bidx = base + k;
      }
    }
  }
  if (!live) return;
  const size_t o = (size_t)b * P + p;
  const bool pos = bidx >= 0 && best < INF;
  labels[o] = pos ? box_labels[g0 + bidx] : num_classes;
  const float nm = norm[p];
  const float tl = bl / nm, tt = bt / nm, tr = br / nm, tb = bb / nm;      // norm_on_bbox: / stride (1 otherwise)
  if (bbox_targets) {
    float* q = bbox_targets + o * 4;
    q[0] = tl; q[1] = tt; q[2] = tr; q[3] = tb;
  }
  if (ctr_target)   // fcos_head.py:1009-1031 on the (normalised) targets of positive points; 0 elsewhere
    ctr_target[o] = pos ? sqrtf((fminf(tl, tr) / fmaxf(tl, tr)) * (fminf(tt, tb) / fmaxf(tt, tb))) : 0.f;
}

}  // namespace pt

using namespace pt;

extern "C" const char* pt_last_error(void) { return pt::g_err; }
extern "C" int pt_abi_version(void) { return 7; }   // 2: pt_roi_align_* lost their workspace argument; 3: r03 additions (pt_sgd_step_groups, ...); 4: pt_conv_weight_item grew taps / scale, plane-native convolutions; 5: fp16 operands (operand_f16 / alpha in the descriptors); 6: the scaled fp16 plane format (tails, pt_planes_mix, census); 7: pt_conv_bf16x6_plan

extern "C" int pt_topk_assign(const float* points, int P, const float* gt_xy, const uint8_t* gt_valid,
                              const int32_t* off, int B, int sumG, int num_pre, int32_t* gt_inds, int32_t* cand,
                              void* stream) {
  PT_REQUIRE(points && off && gt_inds && P > 0 && B > 0 && sumG >= 0, PT_EINVAL, "pt_topk_assign: bad argument");
  PT_REQUIRE(num_pre >= 1 && num_pre <= KMAX, PT_ELIMIT, "pt_topk_assign: num_pre=%d outside [1,%d]", num_pre, KMAX);
  PT_REQUIRE(num_pre <= P, PT_EINVAL, "pt_topk_assign: num_pre > P");
  hipStream_t s = as_stream(stream);
  hipError_t e = hipMemsetAsync(gt_inds, 0, sizeof(int32_t) * (size_t)B * P, s);
  if (e != hipSuccess) { set_error("pt_topk_assign: memset: %s", hipGetErrorString(e)); return (int)e; }
  if (sumG == 0) return PT_OK;
  PT_REQUIRE(gt_xy, PT_EINVAL, "pt_topk_assign: gt_xy is NULL");
  hipLaunchKernelGGL(topk_assign_kernel, dim3(sumG), dim3(256), 0, s, points, P, gt_xy, gt_valid, off, B, num_pre, gt_inds,
                     cand);
  PT_LAUNCH_CHECK("pt_topk_assign");
  return PT_OK;
}

extern "C" int pt_fuse_assign(const float* points, int P, const float* reg, const float* cls, int C,
                              const float* gt_xy, const int32_t* gt_labels, const int32_t* off, int B, int sumG,
                              int num_pre, int topk, float cls_w, float reg_w, float loc_w, int32_t* gt_inds,
                              int32_t* cand, void* stream) {
  PT_REQUIRE(points && reg && cls && off && gt_inds && P > 0 && B > 0 && sumG >= 0, PT_EINVAL,
             "pt_fuse_assign: bad argument");
  PT_REQUIRE(num_pre >= 1 && num_pre <= KMAX, PT_ELIMIT, "pt_fuse_assign: num_pre=%d outside [1,%d]", num_pre, KMAX);
  PT_REQUIRE(C >= 1 && C <= 64, PT_ELIMIT, "pt_fuse_assign: C=%d outside [1,64]", C);
  PT_REQUIRE(topk >= 1 && num_pre <= P, PT_EINVAL, "pt_fuse_assign: bad topk/num_pre");
  hipStream_t s = as_stream(stream);
  hipError_t e = hipMemsetAsync(gt_inds, 0, sizeof(int32_t) * (size_t)B * P, s);
  if (e != hipSuccess) { set_error("pt_fuse_assign: memset: %s", hipGetErrorString(e)); return (int)e; }
  if (sumG == 0) return PT_OK;
  PT_REQUIRE(gt_xy && gt_labels && cand, PT_EINVAL, "pt_fuse_assign: NULL gt array");
  hipLaunchKernelGGL(fuse_assign_kernel<false>, dim3(sumG), dim3(256), 0, s, points, P, reg, cls, C, gt_xy, gt_labels,
                     off, B, num_pre, topk, cls_w, reg_w, loc_w, gt_inds, cand);
  PT_LAUNCH_CHECK("pt_fuse_assign");
  return PT_OK;
}

extern "C" int pt_fuse_assign_obb(const float* points, int P, const float* dec, const float* cls, int C,
                                  const float* gt_xy, const int32_t* gt_labels, const int32_t* off, int B, int sumG,
                                  int num_pre, int topk, float cls_w, float reg_w, float loc_w, int32_t* gt_inds,
                                  int32_t* cand, void* stream) {
  PT_REQUIRE(points && dec && cls && off && gt_inds && P > 0 && B > 0 && sumG >= 0, PT_EINVAL,
             "pt_fuse_assign_obb: bad argument");
  PT_REQUIRE(num_pre >= 1 && num_pre <= KMAX, PT_ELIMIT, "pt_fuse_assign_obb: num_pre=%d outside [1,%d]", num_pre, KMAX);
  PT_REQUIRE(C >= 1 && C <= 64, PT_ELIMIT, "pt_fuse_assign_obb: C=%d outside [1,64]", C);
  PT_REQUIRE(topk >= 1 && num_pre <= P, PT_EINVAL, "pt_fuse_assign_obb: bad topk/num_pre");
  hipStream_t s = as_stream(stream);
  hipError_t e = hipMemsetAsync(gt_inds, 0, sizeof(int32_t) * (size_t)B * P, s);
  if (e != hipSuccess) { set_error("pt_fuse_assign_obb: memset: %s", hipGetErrorString(e)); return (int)e; }
  if (sumG == 0) return PT_OK;
  PT_REQUIRE(gt_xy && gt_labels && cand, PT_EINVAL, "pt_fuse_assign_obb: NULL gt array");
  hipLaunchKernelGGL(fuse_assign_kernel<true>, dim3(sumG), dim3(256), 0, s, points, P, dec, cls, C, gt_xy, gt_labels,
                     off, B, num_pre, topk, cls_w, reg_w, loc_w, gt_inds, cand);
  PT_LAUNCH_CHECK("pt_fuse_assign_obb");
  return PT_OK;
}

extern "C" int pt_pseudo_boxes_obb(const float* dec, int P, const float* cls, int C, const float* gt_xy,
                                   const int32_t* gt_labels, const int32_t* off, int B, int sumG, int num_pre,
                                   const int32_t* gt_inds, const int32_t* cand, float* pseudo_bboxes,
                                   float* pseudo_points, float* pseudo_scores, int32_t* nassigned, void* stream) {
  if (sumG == 0) return PT_OK;
  PT_REQUIRE(dec && cls && gt_xy && gt_labels && off && gt_inds && cand && pseudo_bboxes && pseudo_points &&
                 pseudo_scores && nassigned,
             PT_EINVAL, "pt_pseudo_boxes_obb: NULL argument");
  PT_REQUIRE(P > 0 && B > 0 && C > 0 && num_pre >= 1 && num_pre <= KMAX, PT_EINVAL, "pt_pseudo_boxes_obb: bad size");
  hipLaunchKernelGGL(pseudo_boxes_obb_kernel, dim3(cdiv(sumG, 64)), dim3(64), 0, as_stream(stream), dec, P, cls, C,
                     gt_xy, gt_labels, off, B, sumG, num_pre, gt_inds, cand, pseudo_bboxes, pseudo_points,
                     pseudo_scores, nassigned);
  PT_LAUNCH_CHECK("pt_pseudo_boxes_obb");
  return PT_OK;
}

extern "C" int pt_fcos_targets_obb(const float* points, int P, const int32_t* gt_inds, const float* boxes,
                                   const int32_t* box_labels, const int32_t* off, int B, int num_classes,
                                   int32_t* labels, float* bbox_targets, float* angle_targets, float* ctr_target,
                                   void* stream) {
  PT_REQUIRE(points && gt_inds && off && boxes && labels && bbox_targets && angle_targets && ctr_target && P > 0 &&
                 B > 0,
             PT_EINVAL, "pt_fcos_targets_obb: bad argument");
  const size_t n = (size_t)B * P;
  hipLaunchKernelGGL(fcos_targets_obb_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), points, P, gt_inds,
                     boxes, box_labels, off, B, num_classes, labels, bbox_targets, angle_targets, ctr_target);
  PT_LAUNCH_CHECK("pt_fcos_targets_obb");
  return PT_OK;
}

extern "C" int pt_pseudo_boxes(const float* points, int P, const float* reg, const float* cls, int C,
                               const float* gt_xy, const int32_t* gt_labels, const float* gt_bboxes,
                               const int32_t* off, int B, int sumG, int num_pre, const int32_t* gt_inds,
                               const int32_t* cand, float* pseudo_bboxes, float* pseudo_points,
                               float* pseudo_scores, int32_t* nassigned, float* iou_with_gt, void* stream) {
  if (sumG == 0) return PT_OK;
  PT_REQUIRE(points && reg && cls && gt_xy && gt_labels && off && gt_inds && cand && pseudo_bboxes &&
                 pseudo_points && pseudo_scores && nassigned,
             PT_EINVAL, "pt_pseudo_boxes: NULL argument");
  PT_REQUIRE(P > 0 && B > 0 && C > 0 && num_pre >= 1 && num_pre <= KMAX, PT_EINVAL, "pt_pseudo_boxes: bad size");
  hipLaunchKernelGGL(pseudo_boxes_kernel, dim3(cdiv(sumG, 64)), dim3(64), 0, as_stream(stream), points, P, reg, cls, C,
                     gt_xy, gt_labels, gt_bboxes, off, B, sumG, num_pre, gt_inds, cand, pseudo_bboxes, pseudo_points,
                     pseudo_scores, nassigned, iou_with_gt);
  PT_LAUNCH_CHECK("pt_pseudo_boxes");
  return PT_OK;
}

extern "C" int pt_fcos_targets(const float* points, int P, const int32_t* gt_inds, const float* boxes,
                               const int32_t* box_labels, const int32_t* off, int B, int num_classes,
                               int32_t* labels, float* bbox_targets, float* ctr_target, void* stream) {
  PT_REQUIRE(points && gt_inds && off && labels && P > 0 && B > 0, PT_EINVAL, "pt_fcos_targets: bad argument");
  PT_REQUIRE(boxes || (!bbox_targets && !ctr_target), PT_EINVAL,
             "pt_fcos_targets: bbox_targets/ctr_target requested without boxes");
  const size_t n = (size_t)B * P;
  hipLaunchKernelGGL(fcos_targets_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), points, P, gt_inds,
                     boxes, box_labels, off, B, num_classes, labels, bbox_targets, ctr_target);
  PT_LAUNCH_CHECK("pt_fcos_targets");
  return PT_OK;
}

extern "C" int pt_fcos_dense_targets(const float* points, const float* regress_ranges, const float* sample_radius,
                                     const float* target_norm, int P, const float* boxes, const int32_t* box_labels,
                                     const int32_t* off, int B, int num_classes, int center_sampling, int32_t* labels,
                                     float* bbox_targets, float* ctr_target, void* stream) {
  PT_REQUIRE(points && regress_ranges && sample_radius && target_norm && off && labels && P > 0 && B > 0, PT_EINVAL,
             "pt_fcos_dense_targets: bad argument");
  PT_REQUIRE(B <= 65535, PT_ELIMIT, "pt_fcos_dense_targets: B=%d above 65535", B);
  hipLaunchKernelGGL(fcos_dense_targets_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, as_stream(stream), points, regress_ranges,
                     sample_radius, target_norm, P, boxes, box_labels, off, num_classes, center_sampling, labels, bbox_targets,
                     ctr_target);
  PT_LAUNCH_CHECK("pt_fcos_dense_targets");
  return PT_OK;
}
