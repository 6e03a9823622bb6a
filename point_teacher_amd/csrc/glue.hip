// Small fused kernels for the per-iteration "glue" of the detector logic: each replaces a chain of 7 ... 170 launch-bound
// torch element-wise launches (box format changes, the geometry half of strong_augmentation, the candidate table of the
// burn-in step-1 rectangle generator) with ONE launch over the whole batch.  All of them are a few kilobytes of traffic:
// latency bound, reported in microseconds.  fp32 arithmetic in the reference's operation order (-ffp-contract=off), so
// index-valued decisions downstream (assignment, NMS keep set, inside-the-image filter) see the same roundings.
#include <math.h>

#include "pt_common.h"

namespace pt {

// ---------------------------------------------------------------------------------------------------- box formats --
// mode 0: core/bbox/transforms.py:250-262 bbox_xyxy_to_cxcywh; mode 1: :236-247 bbox_cxcywh_to_xyxy
__global__ void box_convert_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 b = reinterpret_cast<const float4*>(in)[i];
  float4 o;
  if (mode == 0) {
    o.x = (b.x + b.z) / 2.f; o.y = (b.y + b.w) / 2.f; o.z = b.z - b.x; o.w = b.w - b.y;
  } else {
    o.x = b.x - 0.5f * b.z; o.y = b.y - 0.5f * b.w; o.z = b.x + 0.5f * b.z; o.w = b.y + 0.5f * b.w;
  }
  reinterpret_cast<float4*>(out)[i] = o;
}

// --------------------------------------------------------------------------- strong augmentation: geometry half --
// detectors/syn_images_generator_v2.py:41-62 (flip), :64-92 (rescale + centre crop / pad), :114-120 (corner re-order).
// params[b] = {flip_x, flip_y, scale, bw, bh, grows} (host floats; bw / bh are the integer margins of :66-71).
// rows [N, nc] with nc = 2 (points) or 4 (boxes), images delimited by off[B+1].  `valid` (points only, may be NULL):
// the scaled, un-shifted point lies inside the centre crop (:78-79, :84-85); all ones when the image shrinks.
__global__ void aug_geometry_kernel(const float* __restrict__ in, float* __restrict__ out, uint8_t* __restrict__ valid,
                                    const int32_t* __restrict__ off, int B, int nc, const float* __restrict__ params,
                                    float H, float W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = off[B];
  if (i >= N) return;
  int b = 0;
  while (b + 1 < B && i >= off[b + 1]) ++b;
  const float* p = params + b * 6;
  const bool fx = p[0] != 0.f, fy = p[1] != 0.f, grows = p[5] != 0.f;
  const float scale = p[2], bw = p[3], bh = p[4];
  float v[4], q[4];
  for (int k = 0; k < nc; ++k) {
    float t = in[(size_t)i * nc + k];
    if (k & 1) { if (fy) t = H - t; } else { if (fx) t = W - t; }
    const float s = t * scale;
    q[k] = s;
    const float m = (k & 1) ? bh : bw;
    v[k] = grows ? s - m : s + m;
  }
  if (nc == 2) {
    out[(size_t)i * 2] = v[0]; out[(size_t)i * 2 + 1] = v[1];
    if (valid) valid[i] = grows ? (q[0] >= bw && q[0] < W + bw && q[1] >= bh && q[1] < H + bh) : 1;
  } else {
    const float w = fabsf(v[0] - v[2]), h = fabsf(v[1] - v[3]);
    const float x = fminf(v[0], v[2]), y = fminf(v[1], v[3]);
    const float cx = x + w / 2.f, cy = y + h / 2.f;
    out[(size_t)i * 4] = cx - 0.5f * w; out[(size_t)i * 4 + 1] = cy - 0.5f * h;
    out[(size_t)i * 4 + 2] = cx + 0.5f * w; out[(size_t)i * 4 + 3] = cy + 0.5f * h;
  }
}

// --------------------------------------------------- burn-in step 1: candidate rectangles of generate_black_paper --
// detectors/syn_images_generator_v2.py:597-663 for the whole batch: one workgroup per image.  Per image the table is
// [real objects (G) | one rectangle per object (G) | 2 x 5 adjacency copies], rows (x, y, w, h, a, score).
// draws [11, sumG]: scale, x, y, wn, rn, a, boost, itv, itv2, dev (uniform / normal as the reference draws them) + row 10
// unused; cls [sumG] int32 prior indices.  key = (image << 32) | order-preserving bits of -score (rows that do not exist
// sort last): ONE stable int64 sort of the batch then yields every image's descending-score order.
__device__ __forceinline__ unsigned int ordered_bits(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void __launch_bounds__(256)
    black_paper_rects_kernel(const float* __restrict__ gt, int gt_cols, const int32_t* __restrict__ goff, int B,
                             const float* __restrict__ prior, int L, int dense_n, const float* __restrict__ draws,
                             const int32_t* __restrict__ cls, int sumG, float imgsize, float* __restrict__ table,
                             long long* __restrict__ key, uint8_t* __restrict__ exist) {
  __shared__ int hit[2];
  const int b = blockIdx.x;
  const int g0 = goff[b], G = goff[b + 1] - g0;
  const int t0 = 2 * g0 + 10 * b;                 // first table row of this image
  const float PI = 3.14159265358979323846f, HALF_PI = (float)(3.14159265358979323846 / 2);
  const float inv = 1.0f / imgsize;               // torch divides by a host scalar through its reciprocal
  const float cen_lo = 50.f, cen_hi = imgsize - 50.f;
  const float* D = draws;
  if (threadIdx.x == 0) {                          // the first two objects whose np.random.random() < 0.2 (:640, adjboost = 2)
    int n = 0;
    hit[0] = hit[1] = -1;
    for (int j = 0; j < G && n < 2; ++j)
      if (D[6 * (size_t)sumG + g0 + j] < 0.2f) hit[n++] = j;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < G + 10; j += blockDim.x) {
    const bool extra = j >= G;
    const int src = extra ? hit[(j - G) / 5] : j;                 // the object an adjacency copy derives from
    const int k = extra ? (j - G) % 5 + 1 : 0;
    float row[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool ok = !extra;
    if (src >= 0) {
      const size_t o = (size_t)g0 + src;
      const int c = min(max(cls[o], 0), L - 1);
      const float* pr = prior + c * 4;
      const float base = D[o] * 2.0f + 0.5f;                                                   // :597
      float x = D[1 * (size_t)sumG + o] * (cen_hi - cen_lo) + cen_lo;                            // :613-614
      float y = D[2 * (size_t)sumG + o] * (cen_hi - cen_lo) + cen_lo;
      float w = base * expf(fminf(fmaxf(D[3 * (size_t)sumG + o] * 0.4f, -1.f), 1.f) * pr[2]);   // :615-617
      float h = w * expf(fminf(fmaxf(D[4 * (size_t)sumG + o] * 0.4f, -1.f), 1.f) * pr[3]);       // :618-621
      w = w * pr[0];
      h = h * pr[1];
      const float a = D[5 * (size_t)sumG + o] * PI - HALF_PI;                                    // :625
      x = fminf(fmaxf(x, 0.71f * w), (imgsize - 1.f) - 0.71f * w);                              // Tensor.clip: the upper bound wins
      y = fminf(fmaxf(y, 0.71f * h), (imgsize - 1.f) - 0.71f * h);
      const float score = ((w * h) * inv) * inv + 0.1f;
      if (!extra) {
        row[0] = x; row[1] = y; row[2] = w; row[3] = h; row[4] = a; row[5] = score;
      } else {                                                                                  // :640-663
        const bool dense = c < dense_n;
        const float itv = dense ? D[7 * (size_t)sumG + o] * 4.f + 2.f : D[8 * (size_t)sumG + o] * 40.f + 10.f;
        const float dv = dense ? D[9 * (size_t)sumG + o] * 8.f - 4.f : 0.f;
        const float ofx = (h + itv) * sinf(-a) + dv * cosf(a);
        const float ofy = (h + itv) * cosf(a) + dv * sinf(a);
        const float kk = (float)k;
        row[0] = x + kk * ofx; row[1] = y + kk * ofy; row[2] = w; row[3] = h; row[4] = a;
        row[5] = score - 0.001f * kk;
        ok = k <= (dense ? 5 : 3);
      }
    }
    const int r = t0 + G + j;                                        // rectangles follow the G real objects
    for (int q = 0; q < 6; ++q) table[(size_t)r * 6 + q] = row[q];
    exist[r] = ok ? 1 : 0;
    const float sc = ok ? row[5] : -1.f;
    key[r] = ((long long)b << 32) | (long long)ordered_bits(-sc);
  }
  for (int j = threadIdx.x; j < G; j += blockDim.x) {                // the real objects: 0.7 * prior squares, angle 0, score 1 (:599-602)
    const size_t o = (size_t)g0 + j;
    const int c = min(max(cls[o], 0), L - 1);
    const float* bx = gt + o * gt_cols;
    const float cx = (bx[0] + bx[2]) / 2.f, cy = (bx[1] + bx[3]) / 2.f;
    const float s = prior[c * 4] * 0.7f;
    const int r = t0 + j;
    table[(size_t)r * 6 + 0] = cx; table[(size_t)r * 6 + 1] = cy; table[(size_t)r * 6 + 2] = s;
    table[(size_t)r * 6 + 3] = s; table[(size_t)r * 6 + 4] = 0.f; table[(size_t)r * 6 + 5] = 1.f;
    exist[r] = 1;
    key[r] = ((long long)b << 32) | (long long)ordered_bits(-1.f);
  }
}

// After the sort: row i takes table[order[i]].  Writes the sorted rows, the NMS input (rows that do not exist become a
// far-away speck), the polygon (data_augument_bank.py:516-541), the axis-aligned hull (:486-492) and
// pre = exists & score < 1 & inside the image (:669-675); the caller ANDs the NMS keep mask onto it.
__global__ void black_paper_sorted_kernel(const float* __restrict__ table, const long long* __restrict__ order,
                                          const uint8_t* __restrict__ exist, int T, float imgsize,
                                          float* __restrict__ sorted, float* __restrict__ nms_in,
                                          float* __restrict__ polys, float* __restrict__ hull, uint8_t* __restrict__ pre) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T) return;
  const long long src = order[i];
  float r[6];
  for (int q = 0; q < 6; ++q) { r[q] = table[(size_t)src * 6 + q]; sorted[(size_t)i * 6 + q] = r[q]; }
  const bool ex = exist[src] != 0;
  const float far[5] = {-1e4f, -1e4f, 1e-3f, 1e-3f, 0.f};
  for (int q = 0; q < 5; ++q) nms_in[(size_t)i * 5 + q] = ex ? r[q] : far[q];
  const float cx = r[0], cy = r[1], w = r[2], h = r[3], a = r[4];
  const float ca = fabsf(cosf(a)), sa = fabsf(sinf(a));                         // obb2xyxy, syn_images_generator_v2.py:382-396
  const float dw = ca * w + sa * h, dh = sa * w + ca * h;
  const float x1 = cx - dw / 2.f, y1 = cy - dh / 2.f, x2 = cx + dw / 2.f, y2 = cy + dh / 2.f;
  const float mn = fminf(fminf(x1, y1), fminf(x2, y2)), mx = fmaxf(fmaxf(x1, y1), fmaxf(x2, y2));
  pre[i] = (ex && r[5] < 1.f && mn >= 0.f && mx <= imgsize - 1.f) ? 1 : 0;
  const float s = sinf(a), c = cosf(a);
  const float xs[4] = {-w * .5f, w * .5f, w * .5f, -w * .5f}, ys[4] = {-h * .5f, -h * .5f, h * .5f, h * .5f};
  float hx0 = 3.4e38f, hy0 = 3.4e38f, hx1 = -3.4e38f, hy1 = -3.4e38f;
  for (int q = 0; q < 4; ++q) {
    const float px = c * xs[q] - s * ys[q] + cx, py = s * xs[q] + c * ys[q] + cy;
    polys[(size_t)i * 8 + 2 * q] = px; polys[(size_t)i * 8 + 2 * q + 1] = py;
    hx0 = fminf(hx0, px); hy0 = fminf(hy0, py); hx1 = fmaxf(hx1, px); hy1 = fmaxf(hy1, py);
  }
  hull[(size_t)i * 4] = hx0; hull[(size_t)i * 4 + 1] = hy0; hull[(size_t)i * 4 + 2] = hx1; hull[(size_t)i * 4 + 3] = hy1;
}

}  // namespace pt

using namespace pt;

extern "C" int pt_box_convert(const float* in, float* out, int n, int mode, void* stream) {
  if (n == 0) return PT_OK;
  PT_REQUIRE(in && out && n > 0 && (mode == 0 || mode == 1), PT_EINVAL, "pt_box_convert: bad argument");
  hipLaunchKernelGGL(box_convert_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), in, out, n, mode);
  PT_LAUNCH_CHECK("pt_box_convert");
  return PT_OK;
}

extern "C" int pt_aug_geometry(const float* in, float* out, uint8_t* valid, const int32_t* off, int B, int N, int ncoord,
                               const float* params, float H, float W, void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(in && out && off && params && B > 0 && N > 0 && (ncoord == 2 || ncoord == 4), PT_EINVAL,
             "pt_aug_geometry: bad argument");
  hipLaunchKernelGGL(aug_geometry_kernel, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), in, out, valid, off, B,
                     ncoord, params, H, W);
  PT_LAUNCH_CHECK("pt_aug_geometry");
  return PT_OK;
}

extern "C" int pt_black_paper_rects(const float* gt, int gt_cols, const int32_t* goff, int B, const float* prior, int L,
                                    int dense_n, const float* draws, const int32_t* cls, int sumG, float imgsize,
                                    float* table, int64_t* key, uint8_t* exist, void* stream) {
  PT_REQUIRE(goff && prior && table && key && exist && B > 0 && L > 0 && gt_cols >= 4, PT_EINVAL,
             "pt_black_paper_rects: bad argument");
  PT_REQUIRE(sumG == 0 || (gt && draws && cls), PT_EINVAL, "pt_black_paper_rects: NULL input");
  hipLaunchKernelGGL(black_paper_rects_kernel, dim3(B), dim3(256), 0, as_stream(stream), gt, gt_cols, goff, B, prior, L,
                     dense_n, draws, cls, sumG, imgsize, table, reinterpret_cast<long long*>(key), exist);
  PT_LAUNCH_CHECK("pt_black_paper_rects");
  return PT_OK;
}

extern "C" int pt_black_paper_sorted(const float* table, const int64_t* order, const uint8_t* exist, int T, float imgsize,
                                     float* sorted, float* nms_in, float* polys, float* hull, uint8_t* pre, void* stream) {
  if (T == 0) return PT_OK;
  PT_REQUIRE(table && order && exist && sorted && nms_in && polys && hull && pre && T > 0, PT_EINVAL,
             "pt_black_paper_sorted: bad argument");
  hipLaunchKernelGGL(black_paper_sorted_kernel, dim3(cdiv(T, 256)), dim3(256), 0, as_stream(stream), table,
                     reinterpret_cast<const long long*>(order), exist, T, imgsize, sorted, nms_in, polys, hull, pre);
  PT_LAUNCH_CHECK("pt_black_paper_sorted");
  return PT_OK;
}
