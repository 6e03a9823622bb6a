// Detection <-> ground-truth matching of the COCO / AI-TOD evaluator (the reference runs it in the
// C extension of aitodpycocotools, HBB_TOD/mmdet/datasets/aitod.py:109-146).
//
// One wavefront per (image, category) segment; lane = (area range a, IoU threshold t), A*T <= 64.  Every lane walks
// the segment's detections in descending score order and, for each, the segment's ground truths in the order
// pycocotools visits them (non-ignored first, ignored after - "ignored" depends on the lane's area range), keeping
// its own matched flags in a byte plane gt_matched[g][lane].  All lanes look at the same (detection, gt) pair at the
// same time, so box loads are wave-uniform broadcasts and the IoU is computed once per pair and lane.
#include "pt_common.h"
#include "pt_rotated_iou.h"

namespace pt {

__device__ __forceinline__ float coco_iou(const float* d, const float* g, bool crowd) {
  const float w = fminf(d[2], g[2]) - fmaxf(d[0], g[0]);
  const float h = fminf(d[3], g[3]) - fmaxf(d[1], g[1]);
  if (w <= 0.f || h <= 0.f) return 0.f;
  const float inter = w * h;
  const float da = (d[2] - d[0]) * (d[3] - d[1]), ga = (g[2] - g[0]) * (g[3] - g[1]);
  return inter / (crowd ? da : da + ga - inter);
}

// PRE = false: axis-aligned IoU computed on the fly from det_box / gt_box (xyxy).
// PRE = true : IoU read from the ragged per-segment matrices of segment_iou_rotated_kernel (`det_box` then holds one
//              AREA per detection and gt_box is unused): the SODA-A protocol, whose IoU is between oriented boxes.
template <bool PRE>
__global__ void __launch_bounds__(64)
    coco_match_kernel(const float* __restrict__ det_box, const int32_t* __restrict__ det_off,
                      const float* __restrict__ iou_pre, const int64_t* __restrict__ iou_off,
                      const float* __restrict__ gt_box, const float* __restrict__ gt_area,
                      const uint8_t* __restrict__ gt_flags, const int32_t* __restrict__ gt_off, int S,
                      const float* __restrict__ area_lo, const float* __restrict__ area_hi, int A,
                      const float* __restrict__ iou_thr, int T, int max_det, int gt_zero, int det_zero,
                      uint8_t* __restrict__ gt_matched, int32_t* __restrict__ dtm, uint8_t* __restrict__ dt_ig) {
  // gt_zero / det_zero (-1 = none): SODAAeval numbers instances from 0 and reads 0 as "no match" - a detection matched to
  // ground truth `gt_zero` is reported unmatched, and detection `det_zero` does not block the ground truth it takes.
  const int s = blockIdx.x, lane = threadIdx.x;
  const bool live = lane < A * T;
  const int a = live ? lane / T : 0, t = live ? lane - a * T : 0;
  const float lo = area_lo[a], hi = area_hi[a];
  const float thr0 = fminf(iou_thr[t], 1.f - 1e-10f);
  const int d0 = det_off[s], d1 = min(det_off[s + 1], det_off[s] + max_det);
  const int g0 = gt_off[s], g1 = gt_off[s + 1];
  const int L = A * T;
  const int ng = g1 - g0;
  const float* iou_s = PRE ? iou_pre + iou_off[s] : nullptr;
  for (int d = d0; d < d1; ++d) {
    const float* db = det_box + (size_t)d * (PRE ? 1 : 4);
    float best = thr0;
    int m = -1;
    bool m_ig = false;
    // pass 0: ground truths this lane does not ignore; pass 1: the ignored ones (the stable sort of evaluateImg)
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1 && m >= 0) break;                      // `break` rule: a regular match is never traded for an ignored gt
      for (int g = g0; g < g1; ++g) {
        const uint8_t fl = gt_flags[g];                    // bit 0: ignore / crowd flag, bit 1: iscrowd
        const float ar = gt_area[g];
        const bool ig = (fl & 1) || ar < lo || ar > hi;
        if (ig != (pass == 1)) continue;
        const bool crowd = fl & 2;
        if (gt_matched[(size_t)g * 64 + lane] && !crowd) continue;
        const float iou = PRE ? iou_s[(size_t)(d - d0) * ng + (g - g0)] : coco_iou(db, gt_box + (size_t)g * 4, crowd);
        if (iou < best) continue;
        best = iou;
        m = g;
        m_ig = ig;
      }
    }
    if (live) {
      const float dar = PRE ? db[0] : (db[2] - db[0]) * (db[3] - db[1]);
      const bool out_rng = dar < lo || dar > hi;
      uint8_t ign = 0;
      if (m >= 0) {
        if (d != det_zero) gt_matched[(size_t)m * 64 + lane] = 1;
        ign = m_ig;
        if (m == gt_zero) {
          m = -1;
          ign = m_ig || out_rng;
        }
      } else {
        ign = out_rng;
      }
      dtm[(size_t)d * L + lane] = m;
      dt_ig[(size_t)d * L + lane] = ign;
    }
  }
}

// IoU of every (detection, ground truth) pair inside each (image, category) segment, oriented boxes (cx, cy, w, h, a):
// row-major [min(D_s, max_det), G_s] at iou[iou_off[s]].  grid.y = segment, grid.x strides over the segment's pairs.
__global__ void __launch_bounds__(256)
    segment_iou_rotated_kernel(const float* __restrict__ det_box, const int32_t* __restrict__ det_off,
                               const float* __restrict__ gt_box, const int32_t* __restrict__ gt_off, int max_det,
                               const int64_t* __restrict__ iou_off, float* __restrict__ iou) {
  const int s = blockIdx.y;
  const int d0 = det_off[s], nd = min(det_off[s + 1] - d0, max_det), g0 = gt_off[s], ng = gt_off[s + 1] - g0;
  const long n = (long)nd * ng;
  float* out = iou + iou_off[s];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i / ng), g = (int)(i - (long)d * ng);
    float x[5], y[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { x[k] = det_box[(size_t)(d0 + d) * 5 + k]; y[k] = gt_box[(size_t)(g0 + g) * 5 + k]; }
    out[i] = rotated_iou(x, y);
  }
}

}  // namespace pt

using namespace pt;

extern "C" int pt_coco_match(const float* det_box, const int32_t* det_off, const float* gt_box, const float* gt_area,
                             const uint8_t* gt_flags, const int32_t* gt_off, int S, const float* area_lo,
                             const float* area_hi, int A, const float* iou_thr, int T, int max_det, uint8_t* gt_matched,
                             int32_t* dtm, uint8_t* dt_ig, void* stream) {
  if (S == 0) return PT_OK;
  PT_REQUIRE(det_off && gt_off && area_lo && area_hi && iou_thr && dtm && dt_ig && S > 0, PT_EINVAL,
             "pt_coco_match: bad argument");
  PT_REQUIRE(A >= 1 && T >= 1 && A * T <= 64, PT_ELIMIT, "pt_coco_match: A*T=%d above 64", A * T);
  PT_REQUIRE(max_det >= 1, PT_EINVAL, "pt_coco_match: max_det < 1");
  hipLaunchKernelGGL(coco_match_kernel<false>, dim3(S), dim3(64), 0, as_stream(stream), det_box, det_off, nullptr, nullptr,
                     gt_box, gt_area, gt_flags, gt_off, S, area_lo, area_hi, A, iou_thr, T, max_det, -1, -1, gt_matched, dtm, dt_ig);
  PT_LAUNCH_CHECK("pt_coco_match");
  return PT_OK;
}

extern "C" int pt_segment_iou_rotated(const float* det_box, const int32_t* det_off, const float* gt_box,
                                      const int32_t* gt_off, int S, int max_det, const int64_t* iou_off, int64_t max_pairs,
                                      float* iou, void* stream) {
  if (S == 0 || max_pairs == 0) return PT_OK;
  PT_REQUIRE(det_box && det_off && gt_box && gt_off && iou_off && iou && S > 0 && max_det >= 1 && max_pairs > 0, PT_EINVAL,
             "pt_segment_iou_rotated: bad argument");
  PT_REQUIRE(S <= 65535, PT_ELIMIT, "pt_segment_iou_rotated: %d segments in one call (at most 65535; split the call)", S);
  const int gx = (int)((max_pairs + 255) / 256 < 2048 ? (max_pairs + 255) / 256 : 2048);
  hipLaunchKernelGGL(segment_iou_rotated_kernel, dim3(gx, S), dim3(256), 0, as_stream(stream), det_box, det_off, gt_box, gt_off,
                     max_det, iou_off, iou);
  PT_LAUNCH_CHECK("pt_segment_iou_rotated");
  return PT_OK;
}

extern "C" int pt_coco_match_iou(const float* det_area, const int32_t* det_off, const float* iou, const int64_t* iou_off,
                                 const float* gt_area, const uint8_t* gt_flags, const int32_t* gt_off, int S,
                                 const float* area_lo, const float* area_hi, int A, const float* iou_thr, int T, int max_det,
                                 int gt_zero, int det_zero, uint8_t* gt_matched, int32_t* dtm, uint8_t* dt_ig, void* stream) {
  if (S == 0) return PT_OK;
  PT_REQUIRE(det_off && gt_off && iou_off && area_lo && area_hi && iou_thr && dtm && dt_ig && S > 0, PT_EINVAL,
             "pt_coco_match_iou: bad argument");
  PT_REQUIRE(A >= 1 && T >= 1 && A * T <= 64, PT_ELIMIT, "pt_coco_match_iou: A*T=%d above 64", A * T);
  PT_REQUIRE(max_det >= 1, PT_EINVAL, "pt_coco_match_iou: max_det < 1");
  hipLaunchKernelGGL(coco_match_kernel<true>, dim3(S), dim3(64), 0, as_stream(stream), det_area, det_off, iou, iou_off, nullptr,
                     gt_area, gt_flags, gt_off, S, area_lo, area_hi, A, iou_thr, T, max_det, gt_zero, det_zero, gt_matched, dtm, dt_ig);
  PT_LAUNCH_CHECK("pt_coco_match_iou");
  return PT_OK;
}
