/* Oracle (TEST INFRASTRUCTURE ONLY): plain-C restatement of the reference's rotated-box geometry, fp32, one pair at a time.
 *   box_overlap / iou_bev   detector3d/pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:28-234 (same arithmetic as the CPU twin
 *                           iou3d_cpu.cpp, which needs CUDA headers + torch to build and is therefore not compiled here)
 *   nms (mask + greedy)     iou3d_nms_kernel.cu:267-302 + iou3d_nms/src/iou3d_nms.cpp:90-135
 *   nms_normal              iou3d_nms_kernel.cu:305-355
 *   points_in_boxes         roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-36,313-337  (MARGIN 1e-5, first containing box)
 * Build: make -C oracle  ->  oracle/liboracle_geometry.so (gcc -O2 -ffp-contract=off). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EPSF 1e-8f

typedef struct { float x, y; } pt_t;

static float cross2(pt_t a, pt_t b) { return a.x * b.y - a.y * b.x; }                       /* :28-30 */
static float cross3(pt_t p1, pt_t p2, pt_t p0) {                                            /* :32-34 */
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static int rect_cross(pt_t p1, pt_t p2, pt_t q1, pt_t q2) {                                 /* :36-42 */
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}
static int in_box2d(const float* box, pt_t p) {                                             /* :44-52, MARGIN 1e-2 */
  const float MARGIN = 1e-2f;
  float cx = box[0], cy = box[1];
  float ac = cosf(-box[6]), as = sinf(-box[6]);
  float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
  float ry = (p.x - cx) * as + (p.y - cy) * ac;
  return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}
static int intersection(pt_t p1, pt_t p0, pt_t q1, pt_t q0, pt_t* ans) {                    /* :54-83 */
  if (!rect_cross(p0, p1, q0, q1)) return 0;
  float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > EPSF) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}
static pt_t rot_about(pt_t c, float ac, float as, pt_t p) {                                 /* :85-89 */
  pt_t r;
  r.x = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
  r.y = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
  return r;
}

float orc_box_overlap(const float* a, const float* b) {                                      /* :95-223 */
  float aa = a[6], ba = b[6];
  float adx = a[3] / 2, bdx = b[3] / 2, ady = a[4] / 2, bdy = b[4] / 2;
  pt_t ca = {a[0], a[1]}, cb = {b[0], b[1]};
  pt_t A[5] = {{a[0] - adx, a[1] - ady}, {a[0] + adx, a[1] - ady}, {a[0] + adx, a[1] + ady}, {a[0] - adx, a[1] + ady}};
  pt_t B[5] = {{b[0] - bdx, b[1] - bdy}, {b[0] + bdx, b[1] - bdy}, {b[0] + bdx, b[1] + bdy}, {b[0] - bdx, b[1] + bdy}};
  float aac = cosf(aa), aas = sinf(aa), bac = cosf(ba), bas = sinf(ba);
  for (int k = 0; k < 4; k++) { A[k] = rot_about(ca, aac, aas, A[k]); B[k] = rot_about(cb, bac, bas, B[k]); }
  A[4] = A[0]; B[4] = B[0];
  pt_t cp[16 + 8], centre = {0, 0};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (intersection(A[i + 1], A[i], B[j + 1], B[j], &cp[cnt])) { centre.x += cp[cnt].x; centre.y += cp[cnt].y; cnt++; }
  for (int k = 0; k < 4; k++) {
    if (in_box2d(a, B[k])) { centre.x += B[k].x; centre.y += B[k].y; cp[cnt++] = B[k]; }
    if (in_box2d(b, A[k])) { centre.x += A[k].x; centre.y += A[k].y; cp[cnt++] = A[k]; }
  }
  centre.x /= cnt; centre.y /= cnt;
  for (int j = 0; j < cnt - 1; j++)                                                          /* bubble sort by angle, :198-207 */
    for (int i = 0; i < cnt - j - 1; i++)
      if (atan2f(cp[i].y - centre.y, cp[i].x - centre.x) > atan2f(cp[i + 1].y - centre.y, cp[i + 1].x - centre.x)) {
        pt_t t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; k++) {
    pt_t u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y}, v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
    area += cross2(u, v);
  }
  return fabsf(area) / 2.0f;
}

float orc_iou_bev(const float* a, const float* b) {                                          /* :225-231 */
  float sa = a[3] * a[4], sb = b[3] * b[4], so = orc_box_overlap(a, b);
  return so / fmaxf(sa + sb - so, EPSF);
}

static float iou_normal(const float* a, const float* b) {                                    /* :304-314 */
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f), inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, EPSF);
}

void orc_boxes_overlap_bev(const float* A, int na, const float* B, int nb, float* out, int iou) {
  for (int i = 0; i < na; i++)
    for (int j = 0; j < nb; j++) out[(int64_t)i * nb + j] = iou ? orc_iou_bev(A + i * 7, B + j * 7) : orc_box_overlap(A + i * 7, B + j * 7);
}

/* greedy NMS over boxes already sorted by score: keep[] <- kept indices, returns their number */
int orc_nms(const float* boxes, int n, float thresh, int64_t* keep, int normal) {
  unsigned char* removed = (unsigned char*)calloc(n > 0 ? n : 1, 1);
  int num = 0;
  for (int i = 0; i < n; i++) {
    if (removed[i]) continue;
    keep[num++] = i;
    for (int j = i + 1; j < n; j++) {
      if (removed[j]) continue;
      float v = normal ? iou_normal(boxes + i * 7, boxes + j * 7) : orc_iou_bev(boxes + i * 7, boxes + j * 7);
      if (v > thresh) removed[j] = 1;
    }
  }
  free(removed);
  return num;
}

/* points (B,M,3), boxes (B,T,7) -> idx (B,M) first box containing the point or -1 */
void orc_points_in_boxes(const float* boxes, const float* pts, int B, int T, int M, int32_t* out) {
  const float MARGIN = 1e-5f;
  for (int b = 0; b < B; b++)
    for (int m = 0; m < M; m++) {
      const float* p = pts + ((int64_t)b * M + m) * 3;
      int32_t found = -1;
      for (int k = 0; k < T && found < 0; k++) {
        const float* bx = boxes + ((int64_t)b * T + k) * 7;
        if (fabsf(p[2] - bx[2]) > bx[5] / 2.0) continue;                                   /* double compare as in the kernel (:28) */
        float sx = p[0] - bx[0], sy = p[1] - bx[1];
        float ca = cosf(-bx[6]), sa = sinf(-bx[6]);
        float lx = sx * ca + sy * (-sa), ly = sx * sa + sy * ca;
        if ((fabs(lx) < bx[3] / 2.0 + MARGIN) & (fabs(ly) < bx[4] / 2.0 + MARGIN)) found = k;   /* double arithmetic, :30-31 */
      }
      out[(int64_t)b * M + m] = found;
    }
}
