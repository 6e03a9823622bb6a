// Rotated-box IoU shared by nms.hip (pt_box_iou_rotated, rotated NMS) and cocoeval.hip (SODA-A evaluator).
#pragma once
#include "pt_common.h"

namespace pt {

// --------------------------------------------------------------- rotated IoU --
struct P2 {
  float x, y;
};

__device__ __forceinline__ void rbox_corners(const float* b, P2* p) {
  const float c = cosf(b[4]), s = sinf(b[4]);
  const float hw = b[2] * 0.5f, hh = b[3] * 0.5f;
  const float dx[4] = {-hw, hw, hw, -hw}, dy[4] = {-hh, -hh, hh, hh};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    p[i].x = b[0] + dx[i] * c - dy[i] * s;
    p[i].y = b[1] + dx[i] * s + dy[i] * c;
  }
}

// Convex clip of `poly` (n vertices, CCW or CW consistent with the clip box) by the
// half-plane on the left of a->b.  Sutherland-Hodgman; at most n+1 output vertices.
__device__ __forceinline__ int clip_edge(const P2* in, int n, P2 a, P2 b, P2* out) {
  int m = 0;
  const float ex = b.x - a.x, ey = b.y - a.y;
  for (int i = 0; i < n; ++i) {
    const P2 p = in[i], q = in[(i + 1 == n) ? 0 : i + 1];
    const float sp = ex * (p.y - a.y) - ey * (p.x - a.x);
    const float sq = ex * (q.y - a.y) - ey * (q.x - a.x);
    if (sp >= 0.f) out[m++] = p;
    if ((sp > 0.f && sq < 0.f) || (sp < 0.f && sq > 0.f)) {
      const float t = sp / (sp - sq);
      out[m].x = p.x + t * (q.x - p.x);
      out[m].y = p.y + t * (q.y - p.y);
      ++m;
    }
  }
  return m;
}

__device__ inline float rotated_iou(const float* b1, const float* b2) {
  const float a1 = b1[2] * b1[3], a2 = b2[2] * b2[3];
  if (a1 < 1e-14f || a2 < 1e-14f) return 0.f;
  {  // bounding circles apart -> the polygons cannot intersect (exact 0, skips the clip for most pairs)
    const float dx = b2[0] - b1[0], dy = b2[1] - b1[1];
    const float r = 0.5f * (sqrtf(b1[2] * b1[2] + b1[3] * b1[3]) + sqrtf(b2[2] * b2[2] + b2[3] * b2[3]));
    if (dx * dx + dy * dy > r * r * 1.0001f) return 0.f;
  }
  // translate to the first centre to keep fp32 precision (as mmcv's kernel does)
  float c1[5] = {0.f, 0.f, b1[2], b1[3], b1[4]};
  float c2[5] = {b2[0] - b1[0], b2[1] - b1[1], b2[2], b2[3], b2[4]};
  P2 p1[4], p2[4], bufA[12], bufB[12];
  rbox_corners(c1, p1);
  rbox_corners(c2, p2);
  int n = 4;
  for (int i = 0; i < 4; ++i) bufA[i] = p1[i];
  P2* cur = bufA;
  P2* nxt = bufB;
  for (int e = 0; e < 4 && n > 0; ++e) {
    n = clip_edge(cur, n, p2[e], p2[(e + 1) & 3], nxt);
    P2* t = cur; cur = nxt; nxt = t;
  }
  float inter = 0.f;
  for (int i = 0; i < n; ++i) {
    const P2 p = cur[i], q = cur[(i + 1 == n) ? 0 : i + 1];
    inter += p.x * q.y - q.x * p.y;
  }
  inter = fabsf(inter) * 0.5f;
  return inter / (a1 + a2 - inter);
}

}  // namespace pt
