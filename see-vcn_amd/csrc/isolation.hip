// Point isolation on the GPU: the step in front of VCN.  The reference does all of it on the CPU
// with numpy / open3d (see/surface_completion/SEE_VCN.py:61-82,144-181, datasets/shared_utils.py:36-106,
// datasets/kitti/kitti_objects.py:153-176, datasets/kitti/kitti_utils.py:58-114):
//   k_crop_boxes       : open3d PointCloud.crop(OrientedBoundingBox) for every ground-truth box of a scene (isolate_gt_pts)
//   k_project_kitti    : lidar -> reference camera -> rectified -> image plane, image-FOV test (map_pointcloud_to_image)
//   k_mask_select      : mask[v,u] lookup per instance (get_pts_in_mask)
//   k_isolate_cluster  : per instance range-adaptive DBSCAN (min_points >= 1, border points included), largest cluster
// Index lists come out in ascending point order (numpy boolean indexing / np.argwhere order); all geometry in float64 like
// numpy / Eigen.  One workgroup per box / instance.
#include "common.h"

#define ISO_THREADS 1024
#define ISO_WAVES (ISO_THREADS / SV_WAVE)
#define ISO_LDS_N 4096    // instances up to this many points are clustered entirely in LDS, larger ones in caller scratch

// Ascending list of the i in [0,n) with pred(i): out[r] = r-th such i (written while r < cap); returns their number.
// Whole workgroup must call; uses 17 ints of LDS.
template <class Pred>
__device__ __forceinline__ int iso_ordered_select(int64_t n, Pred pred, int32_t* out, int64_t cap) {
  __shared__ int s_wcnt[ISO_WAVES];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __syncthreads();
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int64_t start = 0; start < n; start += ISO_THREADS) {
    const int64_t i = start + tid;
    const bool p = i < n && pred(i);
    const unsigned long long vote = __ballot(p);
    if (lane == 0) s_wcnt[wave] = __popcll(vote);
    __syncthreads();
    int pos = s_base + __popcll(vote & ((1ull << lane) - 1));
    for (int w = 0; w < wave; ++w) pos += s_wcnt[w];
    if (p && pos < cap) out[pos] = (int32_t)i;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < ISO_WAVES; ++w) t += s_wcnt[w];
      s_base += t;
    }
    __syncthreads();
  }
  return s_base;
}

// ---------------------------------------------------------------------------------------------------------------------
// open3d OrientedBoundingBox::GetPointIndicesWithinBoundingBox: d = p - center; inside iff |d . R[:,a]| <= extent[a]/2 for
// the three box axes a (float64).  box row = [cx,cy,cz, R row-major (9), ex,ey,ez].
__global__ __launch_bounds__(ISO_THREADS) void k_crop_boxes(const float* __restrict__ points, int64_t n, int row_stride,
                                                            const double* __restrict__ boxes, int64_t cap,
                                                            int32_t* __restrict__ out_index, int32_t* __restrict__ out_count) {
  const int g = blockIdx.x;
  const double* B = boxes + (size_t)g * 15;
  const double cx = B[0], cy = B[1], cz = B[2];
  const double ax0 = B[3], ax1 = B[6], ax2 = B[9];      // R (1,0,0)
  const double ay0 = B[4], ay1 = B[7], ay2 = B[10];     // R (0,1,0)
  const double az0 = B[5], az1 = B[8], az2 = B[11];     // R (0,0,1)
  const double hx = B[12] / 2, hy = B[13] / 2, hz = B[14] / 2;
  const int total = iso_ordered_select(
      n,
      [&](int64_t i) {
        const float* p = points + i * row_stride;
        const double d0 = (double)p[0] - cx, d1 = (double)p[1] - cy, d2 = (double)p[2] - cz;
        return fabs(d0 * ax0 + d1 * ax1 + d2 * ax2) <= hx && fabs(d0 * ay0 + d1 * ay1 + d2 * ay2) <= hy &&
               fabs(d0 * az0 + d1 * az1 + d2 * az2) <= hz;
      },
      out_index + (size_t)g * cap, cap);
  if (threadIdx.x == 0) out_count[g] = total;
}

// ---------------------------------------------------------------------------------------------------------------------
// KITTI projection chain (kitti_utils.py:69-114) and FOV test (kitti_objects.py:160-173), float64:
//   ref = [x y z 1] . V2C^T ; rect = R0 . ref ; img = [rect 1] . P^T ; (u,v) = img[:2] / img[2]
//   fov = 0 <= u < W and 0 <= v < H and x > min_dist ; pts_img = floor(u,v)
struct ProjectArgs {
  const float* points;
  int64_t n;
  int row_stride;
  double v2c[12], r0[9], p[12];
  int img_w, img_h;
  double min_dist;
  int32_t* uv;        // (n,2), written for every point (undefined outside the FOV)
  uint8_t* fov;       // (n)
  float* rect;        // (n,3) or null: project_velo_to_rect of every point
};

__global__ __launch_bounds__(256) void k_project_kitti(ProjectArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const float* q = a.points + i * a.row_stride;
  const double x = q[0], y = q[1], z = q[2];
  double ref[3], rc[3], im[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) ref[c] = x * a.v2c[c * 4] + y * a.v2c[c * 4 + 1] + z * a.v2c[c * 4 + 2] + a.v2c[c * 4 + 3];
#pragma unroll
  for (int c = 0; c < 3; ++c) rc[c] = a.r0[c * 3] * ref[0] + a.r0[c * 3 + 1] * ref[1] + a.r0[c * 3 + 2] * ref[2];
#pragma unroll
  for (int c = 0; c < 3; ++c) im[c] = rc[0] * a.p[c * 4] + rc[1] * a.p[c * 4 + 1] + rc[2] * a.p[c * 4 + 2] + a.p[c * 4 + 3];
  const double u = im[0] / im[2], v = im[1] / im[2];
  const bool in = u < (double)a.img_w && u >= 0.0 && v < (double)a.img_h && v >= 0.0 && x > a.min_dist;
  a.fov[i] = in ? 1 : 0;
  a.uv[i * 2] = in ? (int32_t)floor(u) : -1;
  a.uv[i * 2 + 1] = in ? (int32_t)floor(v) : -1;
  if (a.rect) a.rect[i * 3] = (float)rc[0], a.rect[i * 3 + 1] = (float)rc[1], a.rect[i * 3 + 2] = (float)rc[2];
}

// CustomDatasetObjects.map_pointcloud_to_image (datasets/custom_dataset/custom_dataset_objects.py:141-192): one 3x4 extrinsic, a
// pinhole (radial k1,k2,k3 + tangential p1,p2) or equidistant (fisheye, 4 coefficients) distortion model, intrinsics, float64.
//   cam = E . [x y z 1]; (xn, yn) = cam.xy / cam.z; pre = cam.z > 0 and |xn| < atan(W / H)
//   fov = pre and 0 < u < W-1 and 0 < v < H-1; pts_img = round-half-even(u, v, depth)
struct CameraArgs {
  const float* points;
  int64_t n;
  int row_stride;
  double e[12], kmat[9], dist[5];
  int model, img_w, img_h;     // model 0 pinhole, 1 equidistant
  int32_t* uvd_int;            // (n,3) rounded [u, v, depth] (-1 outside the FOV)
  double* uvd;                 // (n,3) or null
  uint8_t* fov;
};

__global__ __launch_bounds__(256) void k_project_camera(CameraArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const float* q = a.points + i * a.row_stride;
  const double x = q[0], y = q[1], z = q[2];
  double cam[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) cam[c] = a.e[c * 4] * x + a.e[c * 4 + 1] * y + a.e[c * 4 + 2] * z + a.e[c * 4 + 3];
  const double xn = cam[0] / cam[2], yn = cam[1] / cam[2];
  const bool pre = cam[2] > 0.0 && fabs(xn) < atan((double)a.img_w / (double)a.img_h);
  const double r2 = xn * xn + yn * yn;
  double u, v;
  if (a.model == 1) {
    const double r1 = sqrt(r2), a0 = atan(r1);
    const double a2 = a0 * a0, a4 = a2 * a2;
    const double a1 = a0 * (1 + a.dist[0] * a2 + a.dist[1] * a4 + a.dist[2] * (a4 * a2) + a.dist[3] * (a4 * a4));
    u = (a1 / r1) * xn, v = (a1 / r1) * yn;
  } else {
    const double td = 1 + a.dist[0] * r2 + a.dist[1] * (r2 * r2) + a.dist[4] * (r2 * r2 * r2);
    u = xn * td + 2 * a.dist[2] * xn * yn + a.dist[3] * (r2 + 2 * (xn * xn));
    v = yn * td + a.dist[2] * (r2 + 2 * (yn * yn)) + 2 * a.dist[3] * xn * yn;
  }
  u = a.kmat[0] * u + a.kmat[2];
  v = a.kmat[4] * v + a.kmat[5];
  const bool in = pre && u > 0.0 && u < (double)(a.img_w - 1) && v > 0.0 && v < (double)(a.img_h - 1);
  a.fov[i] = in ? 1 : 0;
  a.uvd_int[i * 3] = in ? (int32_t)rint(u) : -1;
  a.uvd_int[i * 3 + 1] = in ? (int32_t)rint(v) : -1;
  a.uvd_int[i * 3 + 2] = in ? (int32_t)rint(cam[2]) : -1;
  if (a.uvd) a.uvd[i * 3] = u, a.uvd[i * 3 + 1] = v, a.uvd[i * 3 + 2] = cam[2];
}

// NuScenesObjects.map_pointcloud_to_image (datasets/nuscenes/nuscenes_objects.py:237-295, the nuscenes-devkit transform chain it restates):
// lidar frame -> ego (sweep time) -> global -> ego (image time) -> camera, then the pinhole projection.  The devkit's LidarPointCloud keeps
// its points in FLOAT32 and every rotate() / translate() stores back into that array, so each of the 8 steps rounds to float32: the chain
// is kept step by step (a pre-multiplied 4x4 would differ in the last bits and flip pixels / FOV tests at the borders).
// step s: p = float32(R_s . p) (rotate: float64 dot, stored float32)  or  p = float32(p + t_s) (translate).
struct NuscArgs {
  const float* points;
  int64_t n;
  int row_stride;
  double rot[4][9];       // lidar->ego, ego->global, global->ego_cam (already transposed), ego_cam->camera (already transposed)
  double trans[4][3];     // +t for steps 0, 1; -t for steps 2, 3 (already negated)
  double kmat[9];         // camera intrinsic
  int img_w, img_h;
  double min_dist;
  float* pc_cam;          // (n, 3) float32 camera-frame points
  int32_t* pts_img;       // (n, 2) floor(u, v), -1 outside the FOV
  uint8_t* fov;
};
__device__ __forceinline__ void nusc_rotate(float (&p)[3], const double* R) {
  const double x = p[0], y = p[1], z = p[2];
  p[0] = (float)(R[0] * x + R[1] * y + R[2] * z);
  p[1] = (float)(R[3] * x + R[4] * y + R[5] * z);
  p[2] = (float)(R[6] * x + R[7] * y + R[8] * z);
}
__global__ __launch_bounds__(256) void k_project_nuscenes(NuscArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const float* q = a.points + i * a.row_stride;
  float p[3] = {q[0], q[1], q[2]};
#pragma unroll
  for (int s = 0; s < 2; ++s) {                  // sensor -> ego -> global: rotate, then translate
    nusc_rotate(p, a.rot[s]);
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = (float)((double)p[c] + a.trans[s][c]);
  }
#pragma unroll
  for (int s = 2; s < 4; ++s) {                  // global -> ego (image time) -> camera: translate by -t, then rotate by R^T
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = (float)((double)p[c] + a.trans[s][c]);
    nusc_rotate(p, a.rot[s]);
  }
  // view_points(pc_cam, K, normalize=True): float64 K . p, divided by the third row
  const double x = p[0], y = p[1], z = p[2];
  const double u3 = a.kmat[0] * x + a.kmat[1] * y + a.kmat[2] * z, v3 = a.kmat[3] * x + a.kmat[4] * y + a.kmat[5] * z,
               w3 = a.kmat[6] * x + a.kmat[7] * y + a.kmat[8] * z;
  const double u = u3 / w3, v = v3 / w3;
  const bool in = (double)p[2] > a.min_dist && u > 0.0 && u < (double)a.img_w && v > 0.0 && v < (double)a.img_h;
  a.fov[i] = in ? 1 : 0;
  a.pc_cam[i * 3] = p[0], a.pc_cam[i * 3 + 1] = p[1], a.pc_cam[i * 3 + 2] = p[2];
  a.pts_img[i * 2] = in ? (int32_t)floor(u) : -1;
  a.pts_img[i * 2 + 1] = in ? (int32_t)floor(v) : -1;
}

// get_pts_in_mask (shared_utils.py:36-106): for instance g, the FOV points whose pixel is set in mask g (masks (I,H,W) uint8)
// or, with rects (I,4) = [x0,y0,x1,y1] already truncated to int, inside the box (use_bbox, :56-60).
__global__ __launch_bounds__(ISO_THREADS) void k_mask_select(const int32_t* __restrict__ uv, const uint8_t* __restrict__ fov, int64_t n,
                                                             const uint8_t* __restrict__ masks, const int32_t* __restrict__ rects,
                                                             int img_w, int img_h, int64_t cap, int32_t* __restrict__ out_index,
                                                             int32_t* __restrict__ out_count) {
  const int g = blockIdx.x;
  const uint8_t* M = masks ? masks + (size_t)g * img_w * img_h : nullptr;
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  if (rects) x0 = rects[g * 4], y0 = rects[g * 4 + 1], x1 = rects[g * 4 + 2], y1 = rects[g * 4 + 3];
  const int total = iso_ordered_select(
      n,
      [&](int64_t i) {
        if (!fov[i]) return false;
        const int u = uv[i * 2], v = uv[i * 2 + 1];
        if (M) return M[(size_t)v * img_w + u] != 0;
        return v >= y0 && v < y1 && u >= x0 && u < x1;
      },
      out_index + (size_t)g * cap, cap);
  if (threadIdx.x == 0) out_count[g] = total;
}

// ---------------------------------------------------------------------------------------------------------------------
// isolate_det_pts (SEE_VCN.py:144-181) / db_scan(..., return_largest_cluster) (shared_utils.py:395-409) for one instance:
//   eps = clip(eps_scaling * (|mean(xyz)| * tan(vres)), min_eps, max_eps)      (or a fixed eps)
//   labels = open3d cluster_dbscan(eps, min_points); members of the first largest cluster, ascending.
// open3d's sequential DBSCAN reduces to: core = at least min_points points with d^2 < eps^2 (itself included); clusters =
// connected components of the cores, numbered by their smallest core index; a non-core point within eps of a core takes the
// smallest cluster number among those cores (the first cluster that reaches it); the rest is noise.
struct IsoArgs {
  const float* points;
  int row_stride;
  const int32_t* point_index;   // rows of `points`, or null: the instance's points are rows starts[g] .. starts[g]+counts[g]-1
  const int64_t* starts;        // (I) first slot of instance g in point_index / out_local
  const int32_t* counts;        // (I)
  int32_t* out_local;           // slot starts[g]+r <- position (within the instance) of the r-th member of the selected cluster
  int32_t* out_count;           // (I)
  double* out_eps;              // (I)
  int32_t* scratch;             // I * scratch_stride ints, used by instances with more than ISO_LDS_N points
  int64_t scratch_stride;
  double tan_vres, eps_scaling, min_eps, max_eps, fixed_eps;
  int min_points, min_cluster;
  int64_t max_points;
};

__device__ __forceinline__ int iso_find(volatile int* parent, int x) {
  int p = parent[x];
  while (p != x) {
    const int g = parent[p];
    if (g != p) parent[x] = g;       // path halving (parents only ever decrease)
    x = p, p = g;
  }
  return x;
}

__global__ __launch_bounds__(ISO_THREADS) void k_isolate_cluster(IsoArgs a) {
  __shared__ float s_xyz[ISO_LDS_N * 3];
  __shared__ int s_parent[ISO_LDS_N], s_aux[ISO_LDS_N], s_lab[ISO_LDS_N];
  __shared__ double s_eps2;
  __shared__ unsigned long long s_best;
  const int g = blockIdx.x, tid = threadIdx.x;
  const int n = a.counts[g];
  const int64_t start = a.starts[g];
  if (n <= a.min_cluster || n <= 0) {          // `if xyz.shape[0] > min_cluster` (SEE_VCN.py:158)
    if (tid == 0) a.out_count[g] = 0, a.out_eps[g] = 0.0;
    return;
  }
  if (n > a.max_points) {                      // caller's bound violated: report, touch nothing
    if (tid == 0) a.out_count[g] = -1, a.out_eps[g] = 0.0;
    return;
  }
  float* xyz = s_xyz;
  int *parent = s_parent, *aux = s_aux, *lab = s_lab;
  if (n > ISO_LDS_N) {
    int32_t* sc = a.scratch + (size_t)g * a.scratch_stride;
    xyz = reinterpret_cast<float*>(sc), parent = sc + (size_t)3 * n, aux = sc + (size_t)4 * n, lab = sc + (size_t)5 * n;
  }
  for (int i = tid; i < n; i += ISO_THREADS) {
    const int64_t row = a.point_index ? (int64_t)a.point_index[start + i] : start + i;
    const float* p = a.points + row * a.row_stride;
    xyz[i * 3] = p[0], xyz[i * 3 + 1] = p[1], xyz[i * 3 + 2] = p[2];
    parent[i] = i;
  }
  if (tid == 0) s_best = 0ull;
  __syncthreads();
  if (tid == 0) {
    double eps = a.fixed_eps;
    if (!(eps >= 0.0)) {
      double sx = 0.0, sy = 0.0, sz = 0.0;     // open3d get_center(): sequential float64 accumulate, then / n
      for (int i = 0; i < n; ++i) sx += (double)xyz[i * 3], sy += (double)xyz[i * 3 + 1], sz += (double)xyz[i * 3 + 2];
      const double cx = sx / n, cy = sy / n, cz = sz / n;
      const double dist = sqrt(cx * cx + cy * cy + cz * cz);
      const double ring_height = dist * a.tan_vres;
      eps = a.eps_scaling * ring_height;
      eps = eps < a.min_eps ? a.min_eps : (eps > a.max_eps ? a.max_eps : eps);   // np.clip
    }
    a.out_eps[g] = eps;
    s_eps2 = eps * eps;
  }
  __syncthreads();
  const double eps2 = s_eps2;

  // neighbour counts (the point itself included) -> core flags
  for (int i = tid; i < n; i += ISO_THREADS) {
    const double x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    int c = 0;
#pragma unroll 4
    for (int j = 0; j < n; ++j) {
      const double dx = (double)xyz[j * 3] - x, dy = (double)xyz[j * 3 + 1] - y, dz = (double)xyz[j * 3 + 2] - z;
      c += (dx * dx + dy * dy + dz * dz < eps2) ? 1 : 0;
    }
    aux[i] = c;
  }
  __syncthreads();
  // union the cores that are within eps of each other; roots are the smallest index of a component
  for (int i = tid; i < n; i += ISO_THREADS) {
    if (aux[i] < a.min_points) continue;
    const double x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    int my_parent = i;
    for (int j = i + 1; j < n; ++j) {
      const double dx = (double)xyz[j * 3] - x, dy = (double)xyz[j * 3 + 1] - y, dz = (double)xyz[j * 3 + 2] - z;
      if (dx * dx + dy * dy + dz * dz < eps2 && aux[j] >= a.min_points && parent[j] != my_parent) {
        int ra = i, rb = j;
        while (true) {
          ra = iso_find(parent, ra), rb = iso_find(parent, rb);
          if (ra == rb) break;
          if (ra > rb) { const int t = ra; ra = rb; rb = t; }
          if (atomicCAS(&parent[rb], rb, ra) == rb) break;
        }
        my_parent = iso_find(parent, i);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += ISO_THREADS) {
    int l = -1;
    if (aux[i] >= a.min_points) {
      l = iso_find(parent, i);
    } else if (aux[i] >= 2) {                 // border candidate: the smallest cluster among the cores within eps
      const double x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
      int best = 0x7fffffff;
      for (int j = 0; j < n; ++j) {
        if (aux[j] < a.min_points) continue;
        const double dx = (double)xyz[j * 3] - x, dy = (double)xyz[j * 3 + 1] - y, dz = (double)xyz[j * 3 + 2] - z;
        if (dx * dx + dy * dy + dz * dz < eps2) {
          const int r = iso_find(parent, j);
          best = r < best ? r : best;
        }
      }
      l = best == 0x7fffffff ? -1 : best;
    }
    lab[i] = l;
  }
  __syncthreads();
  for (int i = tid; i < n; i += ISO_THREADS) aux[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += ISO_THREADS)
    if (lab[i] >= 0) atomicAdd(&aux[lab[i]], 1);
  __syncthreads();
  for (int i = tid; i < n; i += ISO_THREADS)
    if (aux[i] > 0) atomicMax(&s_best, ((unsigned long long)aux[i] << 32) | (unsigned)(0x7fffffff - i));   // ties: first label
  __syncthreads();
  const unsigned long long best = s_best;
  if (best == 0ull) {                          // every point is noise (`if len(y) > 0`, SEE_VCN.py:173)
    if (tid == 0) a.out_count[g] = 0;
    return;
  }
  const int best_root = 0x7fffffff - (int)(best & 0xffffffffu);
  const int total = iso_ordered_select(n, [&](int64_t i) { return lab[i] == best_root; }, a.out_local + start, n);
  if (tid == 0) a.out_count[g] = total;
}

// ---------------------------------------------------------------------------------------------------------------------
extern "C" int sv_crop_points_in_boxes(const float* points, int64_t n_points, int row_stride, const double* boxes, int n_boxes,
                                       int64_t cap, int32_t* out_index, int32_t* out_count, void* stream) {
  SV_CHECK_ARG(n_points >= 0 && n_boxes >= 0 && cap >= 0 && row_stride >= 3, "sv_crop_points_in_boxes: bad sizes");
  if (n_boxes == 0) return SV_OK;
  SV_CHECK_ARG((points || n_points == 0) && boxes && out_count && (out_index || cap == 0), "sv_crop_points_in_boxes: null pointer");
  hipLaunchKernelGGL(k_crop_boxes, dim3(n_boxes), dim3(ISO_THREADS), 0, sv_stream(stream), points, n_points, row_stride, boxes, cap,
                     out_index, out_count);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_project_lidar_to_image_kitti(const float* points, int64_t n_points, int row_stride, const double* v2c, const double* r0,
                                               const double* p, int img_w, int img_h, double min_dist, int32_t* uv, uint8_t* fov,
                                               float* rect, void* stream) {
  SV_CHECK_ARG(n_points >= 0 && row_stride >= 3 && img_w > 0 && img_h > 0, "sv_project_lidar_to_image_kitti: bad sizes");
  if (n_points == 0) return SV_OK;
  SV_CHECK_ARG(points && v2c && r0 && p && uv && fov, "sv_project_lidar_to_image_kitti: null pointer");
  ProjectArgs a;
  a.points = points, a.n = n_points, a.row_stride = row_stride;
  for (int i = 0; i < 12; ++i) a.v2c[i] = v2c[i], a.p[i] = p[i];      // host pointers: 33 doubles of calibration
  for (int i = 0; i < 9; ++i) a.r0[i] = r0[i];
  a.img_w = img_w, a.img_h = img_h, a.min_dist = min_dist, a.uv = uv, a.fov = fov, a.rect = rect;
  hipLaunchKernelGGL(k_project_kitti, dim3(sv_div_up(n_points, 256)), dim3(256), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_project_lidar_to_image_camera(const float* points, int64_t n_points, int row_stride, const double* extrinsic,
                                                const double* intrinsic, const double* distcoeff, int camera_model, int img_w, int img_h,
                                                int32_t* uvd_int, double* uvd, uint8_t* fov, void* stream) {
  SV_CHECK_ARG(n_points >= 0 && row_stride >= 3 && img_w > 0 && img_h > 0, "sv_project_lidar_to_image_camera: bad sizes");
  SV_CHECK_ARG(camera_model == 0 || camera_model == 1, "sv_project_lidar_to_image_camera: camera_model 0 (pinhole) or 1 (equidistant)");
  if (n_points == 0) return SV_OK;
  SV_CHECK_ARG(points && extrinsic && intrinsic && distcoeff && uvd_int && fov, "sv_project_lidar_to_image_camera: null pointer");
  CameraArgs a;
  a.points = points, a.n = n_points, a.row_stride = row_stride;
  for (int i = 0; i < 12; ++i) a.e[i] = extrinsic[i];                 // host pointers: 26 doubles of calibration
  for (int i = 0; i < 9; ++i) a.kmat[i] = intrinsic[i];
  for (int i = 0; i < 5; ++i) a.dist[i] = distcoeff[i];
  a.model = camera_model, a.img_w = img_w, a.img_h = img_h, a.uvd_int = uvd_int, a.uvd = uvd, a.fov = fov;
  hipLaunchKernelGGL(k_project_camera, dim3(sv_div_up(n_points, 256)), dim3(256), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_project_lidar_to_image_nuscenes(const float* points, int64_t n_points, int row_stride, const double* rotations, const double* translations,
                                                  const double* intrinsic, int img_w, int img_h, double min_dist, float* pc_cam, int32_t* pts_img,
                                                  uint8_t* fov, void* stream) {
  SV_CHECK_ARG(n_points >= 0 && row_stride >= 3 && img_w > 0 && img_h > 0, "sv_project_lidar_to_image_nuscenes: bad sizes");
  if (n_points == 0) return SV_OK;
  SV_CHECK_ARG(points && rotations && translations && intrinsic && pc_cam && pts_img && fov, "sv_project_lidar_to_image_nuscenes: null pointer");
  NuscArgs a;
  a.points = points, a.n = n_points, a.row_stride = row_stride;
  for (int s = 0; s < 4; ++s) {
    for (int e = 0; e < 9; ++e) a.rot[s][e] = rotations[s * 9 + e];
    for (int e = 0; e < 3; ++e) a.trans[s][e] = translations[s * 3 + e];
  }
  for (int e = 0; e < 9; ++e) a.kmat[e] = intrinsic[e];
  a.img_w = img_w, a.img_h = img_h, a.min_dist = min_dist, a.pc_cam = pc_cam, a.pts_img = pts_img, a.fov = fov;
  hipLaunchKernelGGL(k_project_nuscenes, dim3(sv_div_up(n_points, 256)), dim3(256), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_points_in_masks(const int32_t* uv, const uint8_t* fov, int64_t n_points, const uint8_t* masks, const int32_t* rects,
                                  int n_instances, int img_w, int img_h, int64_t cap, int32_t* out_index, int32_t* out_count,
                                  void* stream) {
  SV_CHECK_ARG(n_points >= 0 && n_instances >= 0 && cap >= 0 && img_w > 0 && img_h > 0, "sv_points_in_masks: bad sizes");
  SV_CHECK_ARG((masks != nullptr) != (rects != nullptr) || n_instances == 0, "sv_points_in_masks: give masks or rects, not both");
  if (n_instances == 0) return SV_OK;
  SV_CHECK_ARG((uv && fov) || n_points == 0, "sv_points_in_masks: null pointer");
  SV_CHECK_ARG(out_count && (out_index || cap == 0), "sv_points_in_masks: null output");
  hipLaunchKernelGGL(k_mask_select, dim3(n_instances), dim3(ISO_THREADS), 0, sv_stream(stream), uv, fov, n_points, masks, rects, img_w,
                     img_h, cap, out_index, out_count);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------
// COCO polygons -> binary instance masks: the `dataset.annToMask(instance)` of get_pts_in_mask (see/surface_completion/datasets/shared_utils.py:66),
// i.e. pycocotools' rleFrPoly + rleMerge(union) + rleDecode (cocoapi common/maskApi.c), without the run-length detour:
//   * every polygon edge is walked at 5x the pixel scale, one step per unit of its longer axis (same integer / double arithmetic as rleFrPoly: the
//     vertices (int)(5 v + .5), the minor coordinate (int)(start + slope * t + .5));
//   * wherever the walk's x changes, the down-sampled (column, row) boundary point -- when its column lands on a pixel centre -- is a run boundary of
//     the column-major run-length code, key = column * h + row.  rleFrPoly sorts the keys and turns differences into runs, dropping zero-length
//     runs: a pixel is inside iff an ODD number of keys is <= its column-major index.  Here every key TOGGLES one bit of a per-polygon bit array
//     (atomic xor: equal keys cancel like the zero-length runs), and the mask is the running parity down the columns, carried from column to column;
//   * the polygons of an instance OR into its mask (rleMerge, intersect = 0).
// One workgroup per polygon: polygons of a few hundred vertices, a few thousand boundary points, h x w parity steps.
// ------------------------------------------------------------------------------------------------
constexpr int PM_THREADS = 256, PM_MAXV = 4096, PM_MAXW = 8192;
__device__ __forceinline__ void pm_point(const int* sx, const int* sy, const int* start, int k, int p, int& u, int& v) {
  int lo = 0, hi = k - 1;                                   // last edge whose first point is <= p
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (start[mid] <= p) lo = mid;
    else hi = mid - 1;
  }
  const int j = lo, d = p - start[j];
  int xs = sx[j], xe = sx[j + 1], ys = sy[j], ye = sy[j + 1];
  const int dx = abs(xe - xs), dy = abs(ys - ye);
  const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
  if (flip) {
    int t = xs; xs = xe; xe = t;
    t = ys; ys = ye; ye = t;
  }
  if (dx >= dy) {
    const int t = flip ? dx - d : d;
    u = t + xs;
    if (dx == 0) v = ys;                                      // a repeated vertex: one point (its slope is 0 / 0 in the C code and never used by a key)
    else {
      const double s = (double)(ye - ys) / (double)dx;
      v = (int)((double)ys + s * (double)t + .5);
    }
  } else {
    const int t = flip ? dy - d : d;
    const double s = (double)(xe - xs) / (double)dy;
    v = t + ys;
    u = (int)((double)xs + s * (double)t + .5);
  }
}

// shrink (nullable): per polygon, the distance its boundary moves inwards (shared_utils.py:295-330 shrinks by Polygon.buffer(-d)): a pixel of the
// polygon is then kept only if it lies at least d from every edge.  kept (nullable): pixels written per polygon (0 = the shrunken part is empty).
__global__ __launch_bounds__(PM_THREADS) void k_polygon_masks(const double* __restrict__ xy, const int32_t* __restrict__ poly_off, const int32_t* __restrict__ poly_inst,
                                                            int h, int w, int64_t words, uint32_t* __restrict__ scratch, uint8_t* __restrict__ masks,
                                                            const double* __restrict__ shrink, int32_t* __restrict__ kept) {
  __shared__ int s_x[PM_MAXV + 1], s_y[PM_MAXV + 1], s_start[PM_MAXV + 1];
  __shared__ uint8_t s_col[PM_MAXW];
  __shared__ int s_scan[PM_THREADS];
  const int poly = blockIdx.x, tid = threadIdx.x;
  const int v0 = poly_off[poly], k = poly_off[poly + 1] - v0;
  uint32_t* T = scratch + (int64_t)poly * words;
  for (int64_t i = tid; i < words; i += PM_THREADS) T[i] = 0u;
  if (k < 1) return;                                        // uniform
  for (int j = tid; j < k; j += PM_THREADS) {
    s_x[j] = (int)(5.0 * xy[2 * (int64_t)(v0 + j)] + .5);
    s_y[j] = (int)(5.0 * xy[2 * (int64_t)(v0 + j) + 1] + .5);
  }
  __syncthreads();
  if (tid == 0) s_x[k] = s_x[0], s_y[k] = s_y[0];
  __syncthreads();
  // points per edge -> exclusive prefix (block scan, PM_MAXV / PM_THREADS edges per thread)
  constexpr int PER = PM_MAXV / PM_THREADS;
  int cnt[PER], mine = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int j = tid * PER + q;
    cnt[q] = j < k ? max(abs(s_x[j] - s_x[j + 1]), abs(s_y[j] - s_y[j + 1])) + 1 : 0;
    mine += cnt[q];
  }
  s_scan[tid] = mine;
  __syncthreads();
  for (int off = 1; off < PM_THREADS; off <<= 1) {
    const int t = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += t;
    __syncthreads();
  }
  int run = s_scan[tid] - mine;
  const int m = s_scan[PM_THREADS - 1];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int j = tid * PER + q;
    if (j < k) s_start[j] = run;
    run += cnt[q];
  }
  __threadfence();                                          // the zeroed words are in place before any toggle of another wave
  __syncthreads();
  for (int p = 1 + tid; p < m; p += PM_THREADS) {
    int u, v, up, vp;
    pm_point(s_x, s_y, s_start, k, p, u, v);
    pm_point(s_x, s_y, s_start, k, p - 1, up, vp);
    if (u == up) continue;
    double xd = (double)(u < up ? u : u - 1);
    xd = (xd + .5) / 5.0 - .5;
    if (floor(xd) != xd || xd < 0 || xd > (double)(w - 1)) continue;
    double yd = (double)(v < vp ? v : vp);
    yd = (yd + .5) / 5.0 - .5;
    if (yd < 0) yd = 0;
    else if (yd > (double)h) yd = (double)h;
    yd = ceil(yd);
    const int64_t key = (int64_t)(int)xd * h + (int)yd;
    if (key < (int64_t)h * w) atomicXor(&T[key >> 5], 1u << (key & 31));
  }
  __threadfence();
  __syncthreads();
  // parity of every column's toggles, then the parity carried into each column
  auto bit = [&](int64_t idx) { return (T[idx >> 5] >> (idx & 31)) & 1u; };
  for (int x = tid; x < w; x += PM_THREADS) {
    const int64_t a = (int64_t)x * h, b = a + h;            // bits [a, b)
    unsigned par = 0;
    for (int64_t wd = a >> 5; wd <= (b - 1) >> 5; ++wd) {
      uint32_t bits = T[wd];
      const int64_t lo = wd << 5;
      if (lo < a) bits &= ~0u << (a - lo);
      if (lo + 32 > b) bits &= ~0u >> (lo + 32 - b);
      par ^= __popc(bits) & 1u;
    }
    s_col[x] = (uint8_t)par;
  }
  __syncthreads();
  if (tid == 0) {                                           // exclusive running parity over the columns (w <= PM_MAXW)
    unsigned c = 0;
    for (int x = 0; x < w; ++x) {
      const unsigned t = s_col[x];
      s_col[x] = (uint8_t)c;
      c ^= t;
    }
  }
  __syncthreads();
  uint8_t* M = masks + (int64_t)poly_inst[poly] * h * w;
  const double d = shrink ? shrink[poly] : 0.0, d2 = d * d;
  const double* V = xy + 2 * (int64_t)v0;
  int written = 0;
  for (int x = tid; x < w; x += PM_THREADS) {
    unsigned state = s_col[x];
    int64_t idx = (int64_t)x * h;
    for (int y = 0; y < h; ++y, ++idx) {
      state ^= bit(idx);
      if (!state) continue;
      if (d > 0.0) {
        // squared distance of the pixel (its centre is the integer point: rleFrPoly samples column x at x) to the nearest edge, original vertices
        const double px = (double)x, py = (double)y;
        bool far = true;
        for (int j = 0; j < k && far; ++j) {
          const int jn = j + 1 < k ? j + 1 : 0;
          const double ax = V[2 * j], ay = V[2 * j + 1], ex = V[2 * jn] - ax, ey = V[2 * jn + 1] - ay;
          const double len2 = ex * ex + ey * ey;
          double t = len2 > 0.0 ? ((px - ax) * ex + (py - ay) * ey) / len2 : 0.0;
          t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
          const double qx = ax + t * ex - px, qy = ay + t * ey - py;
          far = qx * qx + qy * qy >= d2;
        }
        if (!far) continue;
      }
      M[(int64_t)y * w + x] = 1;                            // the parts of an instance only ever write ones: their union, whatever the order
      ++written;
    }
  }
  if (kept) {
    __syncthreads();
    s_scan[tid] = written;
    __syncthreads();
    for (int off = PM_THREADS / 2; off > 0; off >>= 1) {
      if (tid < off) s_scan[tid] += s_scan[tid + off];
      __syncthreads();
    }
    if (tid == 0) kept[poly] = s_scan[0];
  }
}

extern "C" size_t sv_polygon_masks_scratch_bytes(int n_polygons, int img_h, int img_w) {
  return (size_t)n_polygons * (((size_t)img_h * img_w + 31) / 32 + 1) * sizeof(uint32_t);
}
// xy: the polygons' vertices, flat doubles (x, y pairs); poly_off (n_polygons + 1): first vertex of each polygon; poly_inst (n_polygons): the instance a
// polygon belongs to; masks (n_instances, img_h, img_w) uint8, written whole (zeros outside the polygons).  All pointers on the device.
// shrink (n_polygons doubles or NULL): inward distance per polygon; kept (n_polygons int32 or NULL): pixels each polygon wrote
extern "C" int sv_polygons_to_masks_shrunk(const double* xy, const int32_t* poly_off, const int32_t* poly_inst, const double* shrink, int n_polygons,
                                           int max_vertices, int n_instances, int img_h, int img_w, void* scratch, uint8_t* masks, int32_t* kept, void* stream) {
  SV_CHECK_ARG(n_polygons >= 0 && n_instances >= 0 && img_h > 0 && img_w > 0, "sv_polygons_to_masks: bad sizes");
  SV_CHECK_ARG(img_w <= PM_MAXW && max_vertices <= PM_MAXV, "sv_polygons_to_masks: images up to %d columns, polygons up to %d vertices", PM_MAXW, PM_MAXV);
  SV_CHECK_ARG((int64_t)img_h * img_w < ((int64_t)1 << 31), "sv_polygons_to_masks: image too large");
  if (n_instances == 0) return SV_OK;
  SV_CHECK_ARG(masks, "sv_polygons_to_masks: null masks");
  hipStream_t st = sv_stream(stream);
  SV_HIP(hipMemsetAsync(masks, 0, (size_t)n_instances * img_h * img_w, st));
  if (n_polygons == 0) return SV_OK;
  SV_CHECK_ARG(xy && poly_off && poly_inst && scratch, "sv_polygons_to_masks: null pointer");
  const int64_t words = ((int64_t)img_h * img_w + 31) / 32 + 1;
  hipLaunchKernelGGL(k_polygon_masks, dim3(n_polygons), dim3(PM_THREADS), 0, st, xy, poly_off, poly_inst, img_h, img_w, words,
                     static_cast<uint32_t*>(scratch), masks, shrink, kept);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_polygons_to_masks(const double* xy, const int32_t* poly_off, const int32_t* poly_inst, int n_polygons, int max_vertices, int n_instances,
                                    int img_h, int img_w, void* scratch, uint8_t* masks, void* stream) {
  return sv_polygons_to_masks_shrunk(xy, poly_off, poly_inst, nullptr, n_polygons, max_vertices, n_instances, img_h, img_w, scratch, masks, nullptr, stream);
}

extern "C" int64_t sv_isolate_cluster_scratch_bytes(int n_instances, int64_t max_points) {
  if (max_points <= ISO_LDS_N) return 0;
  return (int64_t)n_instances * 6 * max_points * 4;
}

extern "C" int sv_isolate_largest_cluster(const float* points, int row_stride, const int32_t* point_index, const int64_t* starts,
                                          const int32_t* counts, int n_instances, int64_t max_points, double tan_vres,
                                          double eps_scaling, double min_eps, double max_eps, double fixed_eps, int min_points,
                                          int min_cluster, void* scratch, int32_t* out_local, int32_t* out_count, double* out_eps,
                                          void* stream) {
  SV_CHECK_ARG(n_instances >= 0 && row_stride >= 3 && min_points >= 1 && max_points >= 0, "sv_isolate_largest_cluster: bad sizes");
  if (n_instances == 0) return SV_OK;
  SV_CHECK_ARG(points && starts && counts && out_local && out_count && out_eps, "sv_isolate_largest_cluster: null pointer");
  SV_CHECK_ARG(max_points <= ISO_LDS_N || scratch, "sv_isolate_largest_cluster: instances above %d points need scratch", ISO_LDS_N);
  IsoArgs a;
  a.points = points, a.row_stride = row_stride, a.point_index = point_index, a.starts = starts, a.counts = counts;
  a.out_local = out_local, a.out_count = out_count, a.out_eps = out_eps;
  a.scratch = static_cast<int32_t*>(scratch), a.scratch_stride = 6 * max_points;
  a.tan_vres = tan_vres, a.eps_scaling = eps_scaling, a.min_eps = min_eps, a.max_eps = max_eps, a.fixed_eps = fixed_eps;
  a.min_points = min_points, a.min_cluster = min_cluster, a.max_points = max_points;
  hipLaunchKernelGGL(k_isolate_cluster, dim3(n_instances), dim3(ISO_THREADS), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
