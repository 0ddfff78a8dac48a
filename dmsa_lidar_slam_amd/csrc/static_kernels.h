// static_kernels.h — launchers of the kernels behind include/dmsa_static_points.h:
// "is there a cloud point within a radius" on a uniform cell grid in HBM, the selection / compaction of
// DmsaSlam::addStaticPoints (DmsaSlam.h:300-344) and the leaf pick of randomGridDownsampling (helpers.h:67-182).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dmsa {

// bounds of the finite points of a cloud, as order-preserving unsigned images of the floats (atomicMin / atomicMax)
struct CloudBounds {
    uint32_t lo[3], hi[3];
    uint32_t num_finite;
    uint32_t pad;
};
__host__ __device__ inline uint32_t float_to_ordered(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    return (v.u & 0x80000000u) ? ~v.u : (v.u | 0x80000000u);
}
__host__ __device__ inline float ordered_to_float(uint32_t o) {
    union { float f; uint32_t u; } v;
    v.u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return v.f;
}

// cell grid: cell (ix,iy,iz) = floor((p - lo) * inv) per axis in double, linear code ix + nx*(iy + ny*iz)
struct CellGrid {
    double lo[3];
    double inv;          // 1 / (1.001 * radius): every point within the radius lies in the 27 cells around the query's cell
    int64_t nx, ny, nz;  // cells per axis
};
struct CellHashEntry {
    uint64_t key;        // linear cell code, ~0 = empty
    uint32_t start;      // first sorted position of the cell
    uint32_t pad;
};

void launch_cloud_bounds_init(CloudBounds* b, hipStream_t s);
void launch_cloud_bounds(const float4* pts, int64_t n, CloudBounds* b, hipStream_t s);
// codes (invalid marker = ~0 for non-finite points) + identity index
void launch_cell_codes(const float4* pts, int64_t n, CellGrid g, uint64_t* code, uint32_t* idx, hipStream_t s);
void launch_cell_codes32(const float4* pts, int64_t n, CellGrid g, uint32_t* code, uint32_t* idx, hipStream_t s);
// sorted point copies + hash of the cell heads; `code_sorted` is u32 or u64 (key32)
void launch_cell_table(const float4* pts, const uint32_t* idx_sorted, const void* code_sorted, bool key32, int64_t n, float4* pts_sorted,
                       CellHashEntry* table, uint32_t table_mask, hipStream_t s);
// flag[i] = 1 iff some cloud point has ((dx*dx + dy*dy) + dz*dz) <= r2 (float) to query i
void launch_radius_exists(const float4* query, int64_t nq, CellGrid g, const float4* pts_sorted, const void* code_sorted, bool key32, int64_t n,
                          const CellHashEntry* table, uint32_t table_mask, float r2, uint8_t* flag, hipStream_t s);
// DmsaSlam.h:318-336: selected = within && isVisible(currPos, point) -> int flags for the scan
void launch_static_flags(const float4* key_xyz, const float4* key_normal, const uint8_t* within, int64_t n, float px, float py, float pz, int32_t* sel,
                         hipStream_t s);
// stable compaction: out[scan[i]] = point i (w = 1) / ring id
void launch_static_scatter(const float4* key_xyz, const int32_t* key_ring, const int32_t* sel, const int32_t* scan_excl, int64_t n, float4* out_xyz,
                           int32_t* out_id, hipStream_t s);
// out[k] = scan_excl[offset[k]] for k <= K (offset[K] == n reads the total = scan_excl[n-1] + sel[n-1])
void launch_pick_offsets(const int32_t* scan_excl, const int32_t* sel, const int64_t* offsets, int K, int64_t n, int32_t* out, hipStream_t s);
void launch_count_flags(const uint8_t* flag, int64_t n, unsigned long long* count, hipStream_t s);
// helpers.h:93-101: out[l] = idx_sorted[start_l + int((double)rnd[l] / RAND_MAX * (double)(cnt_l - 1))]
void launch_leaf_pick(const int32_t* leaf_start, const uint32_t* idx_sorted, const int32_t* rnd, int num_leaves, int32_t* out, hipStream_t s);
// DmsaSlam::preProcess after the grid filter (DmsaSlam.h:594-630): ranges of the picked points (as sortable bits) + identity index,
// the range gate against the threshold at sorted position `thres_pos`, and the stable compaction + lidar->IMU transform (w = 1)
void launch_scan_ranges(const float4* raw, const int32_t* pick, int m, uint32_t* range_bits, uint32_t* iota, hipStream_t s);
void launch_scan_range_gate(const uint32_t* range_bits, const uint32_t* sorted_bits, int m, int thres_pos, float min_dist_ds, float min_dist, int32_t* sel,
                            hipStream_t s);
void launch_scan_emit(const float4* raw, const int32_t* pick, const int32_t* sel, const int32_t* scan_excl, int m, const float* tform_colmajor, float4* out_xyz,
                      int32_t* out_src, int32_t* total, hipStream_t s);
// ContinuousTrajectory::registerPcBuffer (:240-260): out[k] = min(lower_bound(traj_time, stamps[k] - t0), n_total - 1)
// include/dmsa_window_ring.h: one resident scan -> the window's local points (x, y, z, pose-table row) and ring ids
void launch_ring_assemble(const float4* xyz, const double* stamps, const int32_t* ring, int64_t n, double t0, const double* traj_time, int n_total,
                          float4* local_out, int32_t* ring_out, hipStream_t s);
void launch_tform_indices(const double* stamps, int64_t n, double t0, const double* traj_time, int n_total, int32_t* out, hipStream_t s);
// dmsa_slam_ros::callbackPointCloud (:399-486): byte offsets inside a point of the fields the sensor type reads
struct PointCloud2Fields {
    uint32_t x, y, z, stamp, ring;
};
void launch_decode_pointcloud2(const uint8_t* data, uint32_t n, uint32_t point_step, PointCloud2Fields f, int sensor, double stamp_msg, double delta_t, float4* xyz,
                               double* stamp, int32_t* id, hipStream_t s);
// DmsaSlam::updateNormals (DmsaSlam.h:553-567): exact k-NN (k <= 8) on the cell grid built over the same cloud + PCL's normal
// estimation (single-pass float covariance, eigen33, viewpoint flip).  normal = (nx, ny, nz, curvature); nn_index (n x k, optional)
// = neighbour indices in search order, -1 padded.
void launch_knn_normals(const float4* cloud, int64_t n, int k, CellGrid g, double cell_size, const float4* pts_sorted, const uint32_t* idx_sorted,
                        const void* code_sorted, bool key32, const CellHashEntry* table, uint32_t table_mask, uint32_t num_finite, float vpx, float vpy, float vpz,
                        float4* normal, int32_t* nn_index, hipStream_t s);
// addNewKeyframeToMap (:518-523): local[k] = Rinv * (global[pick[k]] - t), ring[k] = ids[pick[k]]
void launch_to_keyframe_frame(const float4* global, const int32_t* ids, const int32_t* pick, int m, const float* rinv_rowmajor, float tx, float ty, float tz,
                              float4* local, int32_t* ring, hipStream_t s);

}  // namespace dmsa

namespace dmsa {
// include/dmsa_aos.h: strided PCL containers packed on the device.  raw = `count` points of `stride` bytes as they lie in the caller's
// cloud; out[i] = (x, y, z, row as int bits); window clouds: row = index[i] (tformIdPerPoint), must lie in [0, row_limit) or *bad is set;
// static points: row = fixed_row; the int32 at aux_offset is the ring id.
void launch_pack_aos_window(const uint8_t* raw, int64_t count, int stride, int xyz_offset, int aux_offset, const int32_t* index, int row_limit, int fixed_row,
                            float4* local_out, int32_t* ring_out, int32_t* bad, hipStream_t s);
// a scan for the resident ring (include/dmsa_window_ring.h): coordinates (w = 1), the double at stamp_offset, the int32 at id_offset
void launch_unpack_ring_scan(const uint8_t* raw, int64_t count, int stride, int xyz_offset, int stamp_offset, int id_offset, float4* xyz_out, double* stamp_out,
                             int32_t* id_out, hipStream_t s);
// keyframe clouds: row = the frame, normal = the four floats at aux_offset
void launch_pack_aos_keyframe(const uint8_t* raw, int64_t count, int stride, int xyz_offset, int aux_offset, int row, float4* local_out, float4* normal_out,
                              hipStream_t s);
}  // namespace dmsa
