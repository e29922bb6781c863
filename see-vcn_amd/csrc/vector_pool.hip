// The remaining pointnet2_stack_cuda entry points (PV-RCNN++ only; detector3d/pcdet/ops/pointnet2/pointnet2_stack/src/
// pointnet2_api.cpp:12-31): voxel_query_wrapper (voxel_query_gpu.cu:11-87), vector_pool_wrapper / vector_pool_grad_wrapper
// (vector_pool_gpu.cu:217-455), query_stacked_local_neighbor_idxs_wrapper_stack (:113-190),
// query_three_nn_by_stacked_local_idxs_wrapper_stack (:18-83).
// The reference runs one thread per query over ALL support points of its scene (and keeps a 1000-int array per thread in
// scratch memory).  Here a 64-lane wave owns a query: 64 support points are tested per step, matches are consumed in ascending
// point order (the order the reference's sequential loop sees them), feature rows are added with one lane per output channel.
#include "common.h"

struct StackBatch {
  int bs, start, n;   // scene of the query, first support row of that scene, support rows in that scene
};

__device__ __forceinline__ StackBatch stack_batch_of(int q, int batch, const int32_t* new_cnt, const int32_t* xyz_cnt) {
  StackBatch r;
  r.bs = 0;
  int acc = new_cnt[0];
  for (int k = 1; k < batch; ++k) {
    if (q < acc) break;
    acc += new_cnt[k];
    r.bs = k;
  }
  r.start = 0;
  for (int k = 0; k < r.bs; ++k) r.start += xyz_cnt[k];
  r.n = xyz_cnt[r.bs];
  return r;
}

__device__ __forceinline__ bool vp_in_range(float lx, float ly, float lz, float dist, float radius2, int neighbor_type) {
  if (neighbor_type == 1) return !(lx * lx + ly * ly + lz * lz > radius2);
  return !(fabs(lx) > dist || fabs(ly) > dist || fabs(lz) > dist);
}

// ---------------------------------------------------------------------------------------------------------------------
// voxel_query: one thread per query walks the (2r+1)^3 voxel window around its voxel in z, y, x order
__global__ __launch_bounds__(256) void k_voxel_query(int M, int R1, int R2, int R3, int nsample, float radius, int z_range, int y_range,
                                                     int x_range, const float* __restrict__ new_xyz, const float* __restrict__ xyz,
                                                     const int32_t* __restrict__ new_coords, const int32_t* __restrict__ point_indices,
                                                     int32_t* __restrict__ idx) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= M) return;
  const float radius2 = radius * radius;
  const float qx = new_xyz[q * 3], qy = new_xyz[q * 3 + 1], qz = new_xyz[q * 3 + 2];
  const int32_t* c = new_coords + (size_t)q * 4;
  const int b = c[0], cz = c[1], cy = c[2], cx = c[3];
  int32_t* out = idx + (size_t)q * nsample;
  int cnt = 0;
  for (int dz = -z_range; dz <= z_range; ++dz) {
    const int z = cz + dz;
    if (z < 0 || z >= R1) continue;
    for (int dy = -y_range; dy <= y_range; ++dy) {
      const int y = cy + dy;
      if (y < 0 || y >= R2) continue;
      for (int dx = -x_range; dx <= x_range; ++dx) {
        const int x = cx + dx;
        if (x < 0 || x >= R3) continue;
        const int nb = point_indices[(((size_t)b * R1 + z) * R2 + y) * R3 + x];
        if (nb < 0) continue;
        const float px = xyz[(size_t)nb * 3], py = xyz[(size_t)nb * 3 + 1], pz = xyz[(size_t)nb * 3 + 2];
        const float d2 = (px - qx) * (px - qx) + (py - qy) * (py - qy) + (pz - qz) * (pz - qz);
        if (d2 > radius2) continue;
        if (cnt < nsample) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) out[l] = nb;
          out[cnt] = nb;
          ++cnt;
        }
      }
    }
  }
  if (cnt == 0) out[0] = -1;
}

// ---------------------------------------------------------------------------------------------------------------------
struct VectorPoolArgs {
  const float* support_xyz;
  const float* support_features;
  const int32_t* xyz_batch_cnt;
  const float* new_xyz;
  const int32_t* new_xyz_batch_cnt;
  float* new_features;        // (M, num_c_out) zero-filled by the caller; sums (pooling 0) or the chosen point's features (1)
  float* new_local_xyz;       // (M, 3*G)
  int32_t* point_cnt_of_grid; // (M, G)
  int32_t* grouped_idxs;      // (num_max_sum_points, 3): [support row, query, grid]
  int32_t* cum_sum;           // (1) zero before the launch
  int gx, gy, gz, batch, M, c_in, c_out, c_each, G, use_xyz, num_max_sum_points, nsample, neighbor_type, pooling_type;
  float dist, sx, sy, sz;
};

__global__ __launch_bounds__(256) void k_vector_pool(VectorPoolArgs a) {
  __shared__ int s_k[4][64], s_g[4][64];
  __shared__ float s_l[4][64][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  if (q >= a.M) return;
  const StackBatch sb = stack_batch_of(q, a.batch, a.new_xyz_batch_cnt, a.xyz_batch_cnt);
  const float* sxyz = a.support_xyz + (size_t)sb.start * 3;
  const float* sfeat = a.support_features + (size_t)sb.start * a.c_in;
  const float qx = a.new_xyz[q * 3], qy = a.new_xyz[q * 3 + 1], qz = a.new_xyz[q * 3 + 2];
  float* out = a.new_features + (size_t)q * a.c_out;
  float* oxyz = a.new_local_xyz + (size_t)q * 3 * a.G;
  int32_t* cnt = a.point_cnt_of_grid + (size_t)q * a.G;
  const float radius2 = a.dist * a.dist;
  int sample_cnt = 0;
  bool done = false;
  for (int base = 0; base < sb.n && !done; base += 64) {
    const int k = base + lane;
    bool hit = false;
    float lx = 0.f, ly = 0.f, lz = 0.f;
    int g = 0;
    if (k < sb.n) {
      lx = sxyz[(size_t)k * 3] - qx, ly = sxyz[(size_t)k * 3 + 1] - qy, lz = sxyz[(size_t)k * 3 + 2] - qz;
      hit = vp_in_range(lx, ly, lz, a.dist, radius2, a.neighbor_type);
      const int ix = (int)floorf((lx + a.dist) / a.sx), iy = (int)floorf((ly + a.dist) / a.sy), iz = (int)floorf((lz + a.dist) / a.sz);
      g = min(max(ix * a.gy * a.gz + iy * a.gz + iz, 0), a.G - 1);
    }
    unsigned long long vote = __ballot(hit);
    int taken = 0;
    while (vote && !done) {                       // wave-uniform: matches of this step in ascending point order
      const int l = __ffsll((long long)vote) - 1;
      vote &= vote - 1;
      const int mg = __shfl(g, l);
      int seen = 0;
      if (a.pooling_type == 1) {                  // lane 0 owns the counters: its own program order keeps read-after-write exact
        if (lane == 0) seen = cnt[mg];
        seen = __shfl(seen, 0);
      }
      if (seen != 0) continue;
      if (lane == l) s_k[wave][taken] = k, s_g[wave][taken] = g, s_l[wave][taken][0] = lx, s_l[wave][taken][1] = ly, s_l[wave][taken][2] = lz;
      if (lane == 0) cnt[mg] += 1;
      __threadfence_block();
      ++taken, ++sample_cnt;
      if (a.pooling_type == 0) done = a.nsample > 0 && sample_cnt >= a.nsample;
      else done = (a.nsample > 0 && sample_cnt >= a.nsample) || sample_cnt >= a.G;
    }
    __threadfence_block();
    if (taken == 0) continue;
    int slot0 = 0;
    if (lane == 0) slot0 = atomicAdd(a.cum_sum, taken);
    slot0 = __shfl(slot0, 0);
    if (lane < taken && slot0 + lane < a.num_max_sum_points) {
      int32_t* row = a.grouped_idxs + (size_t)(slot0 + lane) * 3;
      row[0] = sb.start + s_k[wave][lane], row[1] = q, row[2] = s_g[wave][lane];
    }
    for (int t = 0; t < taken; ++t) {             // accumulate in match order; lane j owns output channel j of every grid
      const int mk = s_k[wave][t], mg = s_g[wave][t];
      const float* f = sfeat + (size_t)mk * a.c_in;
      for (int j = lane; j < a.c_each; j += 64) {
        float acc = a.pooling_type == 0 ? out[mg * a.c_each + j] : 0.f;
        for (int i = j; i < a.c_in; i += a.c_each) acc = a.pooling_type == 0 ? acc + f[i] : f[i];
        out[mg * a.c_each + j] = acc;
      }
      if (a.use_xyz && lane < 3) {
        const float v = s_l[wave][t][lane];
        oxyz[mg * 3 + lane] = a.pooling_type == 0 ? oxyz[mg * 3 + lane] + v : v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_vector_pool_grad(int64_t total, int n_rows, int c_in, int c_out, int c_each, int G,
                                                          const float* __restrict__ grad_new, const int32_t* __restrict__ cnt_of_grid,
                                                          const int32_t* __restrict__ grouped, float* __restrict__ grad_support) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (grouped row, input channel), channel fastest
  if (e >= total) return;
  const int64_t r = e / c_in;
  const int c = (int)(e - r * c_in);
  const int sup = grouped[r * 3], q = grouped[r * 3 + 1], g = grouped[r * 3 + 2];
  const int npts = cnt_of_grid[(size_t)q * G + g];
  const float w = 1 / fmaxf((float)npts, 1.0f);
  atomicAdd(&grad_support[(size_t)sup * c_in + c], grad_new[(size_t)q * c_out + g * c_each + c % c_each] * w);
}

// ---------------------------------------------------------------------------------------------------------------------
// query_stacked_local_neighbor_idxs: per query the first <= min(1000, nsample) support rows within range (ascending), packed
// behind each other; start_len[q] = [first slot, count]; cumsum = total.
#define SLN_CAP 1000
__global__ __launch_bounds__(256) void k_stacked_local_neighbors(const float* __restrict__ support_xyz, const int32_t* __restrict__ xyz_cnt,
                                                                 const float* __restrict__ new_xyz, const int32_t* __restrict__ new_cnt,
                                                                 int32_t* __restrict__ stack_idxs, int32_t* __restrict__ start_len,
                                                                 int32_t* __restrict__ cumsum, int avg_len, float dist, int batch, int M,
                                                                 int nsample, int neighbor_type) {
  __shared__ int s_tmp[4][SLN_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  if (q >= M) return;
  const StackBatch sb = stack_batch_of(q, batch, new_cnt, xyz_cnt);
  const float* sxyz = support_xyz + (size_t)sb.start * 3;
  const float qx = new_xyz[q * 3], qy = new_xyz[q * 3 + 1], qz = new_xyz[q * 3 + 2];
  const float radius2 = dist * dist;
  const int limit = nsample > 0 && nsample < SLN_CAP ? nsample : SLN_CAP;
  int count = 0;
  for (int base = 0; base < sb.n && count < limit; base += 64) {
    const int k = base + lane;
    bool hit = false;
    if (k < sb.n) hit = vp_in_range(sxyz[(size_t)k * 3] - qx, sxyz[(size_t)k * 3 + 1] - qy, sxyz[(size_t)k * 3 + 2] - qz, dist, radius2, neighbor_type);
    const unsigned long long vote = __ballot(hit);
    const int pos = count + __popcll(vote & ((1ull << lane) - 1));
    if (hit && pos < limit) s_tmp[wave][pos] = k;
    count = min(count + __popcll(vote), limit);
  }
  __threadfence_block();
  int first = 0;
  if (lane == 0) {
    first = atomicAdd(cumsum, count);
    start_len[q * 2] = first, start_len[q * 2 + 1] = count;
  }
  first = __shfl(first, 0);
  const int max_thresh = avg_len * M;
  if (first >= max_thresh) return;
  if (first + count >= max_thresh) count = max_thresh - first;
  for (int t = lane; t < count; t += 64) stack_idxs[first + t] = s_tmp[wave][t] + sb.start;
}

// query_three_nn_by_stacked_local_idxs: one thread per (query, grid centre), three nearest of the query's neighbour list
__global__ __launch_bounds__(256) void k_three_nn_local(const float* __restrict__ support_xyz, const float* __restrict__ centers,
                                                        int32_t* __restrict__ out_idx, float* __restrict__ out_d2,
                                                        const int32_t* __restrict__ stack_idxs, const int32_t* __restrict__ start_len,
                                                        int64_t total, int G) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (query, grid)
  if (e >= total) return;
  const int64_t q = e / G;
  const float cx = centers[e * 3], cy = centers[e * 3 + 1], cz = centers[e * 3 + 2];
  const int32_t* nb = stack_idxs + start_len[q * 2];
  const int len = start_len[q * 2 + 1];
  double b1 = 1e40, b2 = 1e40, b3 = 1e40;
  int i1 = -1, i2 = -1, i3 = -1;
  for (int k = 0; k < len; ++k) {
    const int p = nb[k];
    const float x = support_xyz[(size_t)p * 3], y = support_xyz[(size_t)p * 3 + 1], z = support_xyz[(size_t)p * 3 + 2];
    const float d = (cx - x) * (cx - x) + (cy - y) * (cy - y) + (cz - z) * (cz - z);
    if (d < b1) b3 = b2, i3 = i2, b2 = b1, i2 = i1, b1 = d, i1 = p;
    else if (d < b2) b3 = b2, i3 = i2, b2 = d, i2 = p;
    else if (d < b3) b3 = d, i3 = p;
  }
  if (i2 == -1) i2 = i1, b2 = b1;
  if (i3 == -1) i3 = i1, b3 = b1;
  out_d2[e * 3] = (float)b1, out_d2[e * 3 + 1] = (float)b2, out_d2[e * 3 + 2] = (float)b3;
  out_idx[e * 3] = i1, out_idx[e * 3 + 1] = i2, out_idx[e * 3 + 2] = i3;
}

// ---------------------------------------------------------------------------------------------------------------------
extern "C" int sv_voxel_query(int M, int R1, int R2, int R3, int nsample, float radius, int z_range, int y_range, int x_range,
                              const float* new_xyz, const float* xyz, const int32_t* new_coords, const int32_t* point_indices, int32_t* idx,
                              void* stream) {
  SV_CHECK_ARG(M >= 0 && nsample >= 1 && R1 >= 1 && R2 >= 1 && R3 >= 1 && z_range >= 0 && y_range >= 0 && x_range >= 0, "sv_voxel_query: bad sizes");
  if (M == 0) return SV_OK;
  SV_CHECK_ARG(new_xyz && xyz && new_coords && point_indices && idx, "sv_voxel_query: null pointer");
  hipLaunchKernelGGL(k_voxel_query, dim3(sv_div_up(M, 256)), dim3(256), 0, sv_stream(stream), M, R1, R2, R3, nsample, radius, z_range, y_range,
                     x_range, new_xyz, xyz, new_coords, point_indices, idx);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_vector_pool(const float* support_xyz, const int32_t* xyz_batch_cnt, const float* support_features, const float* new_xyz,
                              const int32_t* new_xyz_batch_cnt, float* new_features, float* new_local_xyz, int32_t* point_cnt_of_grid,
                              int32_t* grouped_idxs, int32_t* cum_sum, int batch, int M, int num_c_in, int num_c_out, int num_grid_x,
                              int num_grid_y, int num_grid_z, float max_neighbour_distance, int use_xyz, int num_max_sum_points, int nsample,
                              int neighbor_type, int pooling_type, void* stream) {
  const int G = num_grid_x * num_grid_y * num_grid_z;
  SV_CHECK_ARG(batch >= 1 && M >= 0 && G >= 1 && num_c_in >= 1 && num_c_out >= G && num_c_out % G == 0, "sv_vector_pool: bad sizes");
  SV_CHECK_ARG(pooling_type == 0 || pooling_type == 1, "sv_vector_pool: pooling_type 0 (avg) or 1 (first point)");
  SV_CHECK_ARG(cum_sum, "sv_vector_pool: null counter");
  SV_HIP(hipMemsetAsync(cum_sum, 0, sizeof(int32_t), sv_stream(stream)));
  if (M == 0) return SV_OK;
  SV_CHECK_ARG(support_xyz && xyz_batch_cnt && support_features && new_xyz && new_xyz_batch_cnt && new_features && new_local_xyz &&
                   point_cnt_of_grid && (grouped_idxs || num_max_sum_points == 0),
               "sv_vector_pool: null pointer");
  VectorPoolArgs a;
  a.support_xyz = support_xyz, a.support_features = support_features, a.xyz_batch_cnt = xyz_batch_cnt, a.new_xyz = new_xyz;
  a.new_xyz_batch_cnt = new_xyz_batch_cnt, a.new_features = new_features, a.new_local_xyz = new_local_xyz;
  a.point_cnt_of_grid = point_cnt_of_grid, a.grouped_idxs = grouped_idxs, a.cum_sum = cum_sum;
  a.gx = num_grid_x, a.gy = num_grid_y, a.gz = num_grid_z, a.batch = batch, a.M = M, a.c_in = num_c_in, a.c_out = num_c_out;
  a.c_each = num_c_out / G, a.G = G, a.use_xyz = use_xyz, a.num_max_sum_points = num_max_sum_points, a.nsample = nsample;
  a.neighbor_type = neighbor_type, a.pooling_type = pooling_type, a.dist = max_neighbour_distance;
  a.sx = max_neighbour_distance * 2 / num_grid_x, a.sy = max_neighbour_distance * 2 / num_grid_y, a.sz = max_neighbour_distance * 2 / num_grid_z;
  hipLaunchKernelGGL(k_vector_pool, dim3(sv_div_up(M, 4)), dim3(256), 0, sv_stream(stream), a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_vector_pool_grad(const float* grad_new_features, const int32_t* point_cnt_of_grid, const int32_t* grouped_idxs,
                                   float* grad_support_features, int num_grouped, int num_c_in, int num_c_out, int num_total_grids,
                                   void* stream) {
  SV_CHECK_ARG(num_grouped >= 0 && num_c_in >= 1 && num_total_grids >= 1 && num_c_out % num_total_grids == 0, "sv_vector_pool_grad: bad sizes");
  const int64_t total = (int64_t)num_grouped * num_c_in;
  if (total == 0) return SV_OK;
  SV_CHECK_ARG(grad_new_features && point_cnt_of_grid && grouped_idxs && grad_support_features, "sv_vector_pool_grad: null pointer");
  hipLaunchKernelGGL(k_vector_pool_grad, dim3(sv_div_up(total, 256)), dim3(256), 0, sv_stream(stream), total, num_grouped, num_c_in, num_c_out,
                     num_c_out / num_total_grids, num_total_grids, grad_new_features, point_cnt_of_grid, grouped_idxs, grad_support_features);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_query_stacked_local_neighbor_idxs(const float* support_xyz, const int32_t* xyz_batch_cnt, const float* new_xyz,
                                                    const int32_t* new_xyz_batch_cnt, int32_t* stack_neighbor_idxs, int32_t* start_len,
                                                    int32_t* cumsum, int avg_length_of_neighbor_idxs, float max_neighbour_distance, int batch,
                                                    int M, int nsample, int neighbor_type, void* stream) {
  SV_CHECK_ARG(batch >= 1 && M >= 0 && avg_length_of_neighbor_idxs >= 0, "sv_query_stacked_local_neighbor_idxs: bad sizes");
  if (M == 0) return SV_OK;
  SV_CHECK_ARG(support_xyz && xyz_batch_cnt && new_xyz && new_xyz_batch_cnt && start_len && cumsum &&
                   (stack_neighbor_idxs || avg_length_of_neighbor_idxs == 0),
               "sv_query_stacked_local_neighbor_idxs: null pointer");
  hipLaunchKernelGGL(k_stacked_local_neighbors, dim3(sv_div_up(M, 4)), dim3(256), 0, sv_stream(stream), support_xyz, xyz_batch_cnt, new_xyz,
                     new_xyz_batch_cnt, stack_neighbor_idxs, start_len, cumsum, avg_length_of_neighbor_idxs, max_neighbour_distance, batch, M,
                     nsample, neighbor_type);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_query_three_nn_by_stacked_local_idxs(const float* support_xyz, const float* new_xyz_grid_centers, int32_t* new_xyz_grid_idxs,
                                                       float* new_xyz_grid_dist2, const int32_t* stack_neighbor_idxs, const int32_t* start_len,
                                                       int M, int num_total_grids, void* stream) {
  SV_CHECK_ARG(M >= 0 && num_total_grids >= 1, "sv_query_three_nn_by_stacked_local_idxs: bad sizes");
  const int64_t total = (int64_t)M * num_total_grids;
  if (total == 0) return SV_OK;
  SV_CHECK_ARG(support_xyz && new_xyz_grid_centers && new_xyz_grid_idxs && new_xyz_grid_dist2 && start_len,
               "sv_query_three_nn_by_stacked_local_idxs: null pointer");
  hipLaunchKernelGGL(k_three_nn_local, dim3(sv_div_up(total, 256)), dim3(256), 0, sv_stream(stream), support_xyz, new_xyz_grid_centers,
                     new_xyz_grid_idxs, new_xyz_grid_dist2, stack_neighbor_idxs, start_len, total, num_total_grids);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
