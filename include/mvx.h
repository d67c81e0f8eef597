/*
 * mvx: C ABI of the MI355X (gfx950) multi-view triangulation kernels.
 *
 * Every entry point takes raw device pointers (float32, contiguous, row-major),
 * sizes, and the HIP stream to launch on (a hipStream_t passed as void*; NULL is
 * the default stream).  Nothing here allocates, frees or synchronises: outputs
 * and workspaces are caller-owned device memory, launches are asynchronous on
 * the given stream.  Return value: MVX_OK or an MVX_ERR_* code; the message of
 * the calling thread's last error is available from mvx_last_error().
 *
 * Each declaration names the function of the reference implementation
 * (huskyth/PoseEstimation, package `mvn`) that it replaces.
 */
#ifndef MVX_H
#define MVX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVX_VERSION 200 /* major*10000 + minor*100 + patch; 0.2.0 adds the backward entry points */

enum mvx_status {
    MVX_OK = 0,
    MVX_ERR_INVALID_ARGUMENT = 1, /* null pointer, non-positive size, misaligned buffer */
    MVX_ERR_HIP = 2,              /* a HIP runtime call or launch failed */
    MVX_ERR_WORKSPACE = 3,        /* workspace smaller than the *_workspace_bytes() answer */
    MVX_ERR_UNSUPPORTED = 4       /* shape outside what the kernels are built for */
};

/* View aggregation of mvx_unproject; mvn/utils/op.py:150-164 ("conf" and "conf_norm" are both MVX_AGG_CONF:
 * the caller normalises the confidences, as mvn/models/triangulation.py:305-306 does). */
enum mvx_aggregation {
    MVX_AGG_SUM = 0,
    MVX_AGG_CONF = 1,
    MVX_AGG_MAX = 2,
    MVX_AGG_SOFTMAX = 3
};

int mvx_version(void);
const char* mvx_last_error(void);

/*
 * Testing aid with no counterpart in the reference: several entry points choose between kernels that compute the
 * same result (by shape and alignment).  A bit set here (process-wide, default 0) takes one kernel out of the
 * choice, so that tests and A/B timings can run the same inputs through the others.  The mask is one plain
 * process-wide variable read by every launch: it is NOT thread-safe and not meant for production callers (set it
 * while no other thread is inside the library).
 */
#define MVX_SELECT_NO_UNPROJECT_BRICKS 1u /* mvx_unproject*: 8x8x8-voxel brick kernel (4 views x 32 channels) off */
#define MVX_SELECT_NO_DLT_ROWS 2u         /* DLT entry points: 16-lanes-per-point kernel for small batches off */
#define MVX_SELECT_NO_SOFT_ARGMAX_RESIDENT 4u /* mvx_soft_argmax_3d*: one-read resident kernel off (three passes) */
#define MVX_SELECT_SOFT_ARGMAX_HELP_ALWAYS 8u /* one-read kernel: every workgroup recomputes its team-mates' partial
                                                 sums itself instead of reading them (the path it otherwise takes
                                                 only for a team-mate that is not resident) */
#define MVX_SELECT_NO_BACKWARD_IN_LDS 16u /* mvx_unproject_backward: gradient slices accumulated in LDS off (global
                                             atomics, the any-shape kernel) */
void mvx_set_kernel_selection(unsigned mask);

/*
 * Soft-argmax of n_maps heatmaps of h x w.           Replaces mvn/utils/op.py:11-47 (integrate_tensor_2d),
 * with the caller's `heatmaps * multiplier` (mvn/models/triangulation.py:190) folded in: each element is
 * first rounded to float32(x * multiplier), exactly as the separate torch multiply does (pass 1.0f for none).
 *   softmax != 0: p = softmax over h*w; coords = (sum p*col, sum p*row); heatmaps_out = p.
 *   softmax == 0: p = max(x, 0);       coords = (sum p*col, sum p*row) / sum p; heatmaps_out = p (not normalised).
 * heatmaps      [n_maps][h][w]
 * coords        [n_maps][2]   (x, y) in heatmap pixels
 * heatmaps_out  [n_maps][h][w] or NULL to skip the write (must not overlap `heatmaps`)
 */
int mvx_soft_argmax_2d(const float* heatmaps, int64_t n_maps, int h, int w, float multiplier, int softmax,
                       float* coords, float* heatmaps_out, void* stream);

/*
 * Confidence-weighted DLT triangulation, one 3-D point per (frame, joint).
 * Replaces mvn/utils/multiview.py:175-198 (triangulate_batch_of_points) and, per point,
 * mvn/utils/multiview.py:143-172 (triangulate_point_from_multiple_views_linear_torch; its host LAPACK SVD
 * becomes a 4x4 symmetric eigen-solve in float64 registers).
 * proj    [n_frames][n_views][3][4]
 * points  [n_frames][n_views][n_joints][2]
 * conf    [n_frames][n_views][n_joints] or NULL (all ones)
 * out     [n_frames][n_joints][3]
 */
int mvx_triangulate_dlt(const float* proj, const float* points, const float* conf, int64_t n_frames, int n_views,
                        int n_joints, float* out, void* stream);

/*
 * The tensor glue of AlgebraicTriangulationNet.forward fused with the DLT.
 * Replaces mvn/models/triangulation.py:199-200 (conf = conf / sum_views(conf) + 1e-5),
 * :208-211 (keypoints scaled from heatmap to image pixels) and :217-220 (the triangulation call).
 * points_hm      [n_frames][n_views][n_joints][2]  heatmap-pixel keypoints (output of mvx_soft_argmax_2d)
 * conf_raw       [n_frames][n_views][n_joints]     confidences as the backbone emits them
 * points_img_out [n_frames][n_views][n_joints][2]  x * scale_x, y * scale_y
 * conf_norm_out  [n_frames][n_views][n_joints]
 * out            [n_frames][n_joints][3]
 */
int mvx_algebraic_glue_dlt(const float* proj, const float* points_hm, const float* conf_raw, int64_t n_frames,
                           int n_views, int n_joints, float scale_x, float scale_y, float* points_img_out,
                           float* conf_norm_out, float* out, void* stream);

/*
 * Everything AlgebraicTriangulationNet.forward does after the backbone (mvn/models/triangulation.py:189-220)
 * behind one call: mvx_soft_argmax_2d on the [n_frames*n_views*n_joints] heatmaps, then mvx_algebraic_glue_dlt.
 * heatmaps             [n_frames][n_views][n_joints][h][w]   raw backbone output
 * heatmaps_out         same shape, or NULL to skip the write of the normalised maps
 * points_hm_workspace  [n_frames][n_views][n_joints][2]      receives the heatmap-pixel keypoints
 * other arguments as in the two calls it stands for.
 */
int mvx_algebraic_forward(const float* heatmaps, const float* conf_raw, const float* proj, int64_t n_frames,
                          int n_views, int n_joints, int h, int w, float multiplier, int softmax, float scale_x,
                          float scale_y, float* heatmaps_out, float* points_img_out, float* conf_norm_out,
                          float* out, float* points_hm_workspace, void* stream);

/*
 * Projects n_points 3-D points with one 3x4 matrix.  Replaces mvn/utils/multiview.py:91-112
 * (project_3d_points_to_image_plane_without_distortion) for device tensors.
 * out [n_points][2] if to_euclidean != 0 (divided by the third component), else [n_points][3].
 */
int mvx_project_points(const float* proj, const float* points, int64_t n_points, int to_euclidean, float* out,
                       void* stream);

/*
 * Unprojects per-view feature maps into a voxel grid and aggregates the views.
 * Replaces mvn/utils/op.py:99-166 (unproject_heatmaps).
 * features   [n_frames][n_views][n_channels][h][w]
 * proj       [n_frames][n_views][3][4]              projection to feature-map pixels
 * coords     [n_frames][vx][vy][vz][3]              world position of every voxel
 * vol_conf   [n_frames][n_views][n_channels]        MVX_AGG_CONF only, else may be NULL
 * volumes    [n_frames][n_channels][vx][vy][vz]
 * workspace  mvx_unproject_workspace_bytes() bytes of device memory, 256-byte aligned (channel-last copy of
 *            the feature maps)
 */
size_t mvx_unproject_workspace_bytes(int64_t n_frames, int n_views, int n_channels, int h, int w);
int mvx_unproject(const float* features, const float* proj, const float* coords, const float* vol_conf,
                  int64_t n_frames, int n_views, int n_channels, int h, int w, int vx, int vy, int vz,
                  int aggregation, float* volumes, void* workspace, size_t workspace_bytes, void* stream);

/*
 * mvx_unproject for feature maps that are already channel-last in memory (a torch tensor of shape
 * [n_frames*n_views][n_channels][h][w] in torch.channels_last format, as a backbone run in that format
 * produces): no layout copy and no workspace.  Needs n_channels % 4 == 0 and 16-byte aligned features.
 * Same operation, same results as mvx_unproject (mvn/utils/op.py:99-166).
 * features   [n_frames][n_views][h][w][n_channels]
 */
int mvx_unproject_channels_last(const float* features, const float* proj, const float* coords, const float* vol_conf,
                                int64_t n_frames, int n_views, int n_channels, int h, int w, int vx, int vy, int vz,
                                int aggregation, float* volumes, void* stream);

/*
 * mvx_unproject / mvx_unproject_channels_last with the voxel positions given by the parameters of a regular grid
 * instead of a coordinate volume (see mvx_soft_argmax_3d_grid): replaces mvn/utils/op.py:99-166 for the grid that
 * mvn/models/triangulation.py:341-385 builds, without reading its 12 bytes per voxel; bit-identical to mvx_unproject
 * fed the volume mvx_build_coord_volumes writes for the same parameters.  channels_last != 0: features are
 * [n_frames][n_views][h][w][n_channels] and no workspace is needed; otherwise as mvx_unproject.
 */
int mvx_unproject_grid(const float* features, int channels_last, const float* proj, const float* position,
                       const float* center, const float* rotation, float step0, float step1, float step2,
                       int swap_yz_flip, const float* vol_conf, int64_t n_frames, int n_views, int n_channels, int h,
                       int w, int vx, int vy, int vz, int aggregation, float* volumes, void* workspace,
                       size_t workspace_bytes, void* stream);

/*
 * Soft-argmax over voxel grids.
 * coords_volume != NULL replaces mvn/utils/op.py:84-96 (integrate_tensor_3d_with_coordinates):
 *   softmax != 0: p = softmax over the grid; out = sum p * coord;  softmax == 0: p = max(x,0), out = sum p * coord.
 * coords_volume == NULL replaces mvn/utils/op.py:50-81 (integrate_tensor_3d): expectation of the voxel index
 *   (x, y, z) instead; with softmax == 0 the result is divided by sum p.
 * As in mvx_soft_argmax_2d every element is first rounded to float32(x * multiplier)
 * (mvn/models/triangulation.py:399).
 * volumes        [n_frames][n_volumes][sx][sy][sz]
 * coords_volume  [n_frames][sx][sy][sz][3] or NULL
 * out            [n_frames][n_volumes][3]
 * volumes_out    [n_frames][n_volumes][sx][sy][sz] or NULL to skip the write.  Either the same pointer as `volumes`
 *                (in place: served by the three-pass kernels) or a buffer that does not overlap `volumes` at all;
 *                a partial overlap is refused with MVX_ERR_INVALID_ARGUMENT.  The same holds for
 *                mvx_soft_argmax_3d_grid and mvx_soft_argmax_3d_static.
 * workspace      mvx_soft_argmax_3d_workspace_bytes() bytes of device memory, 256-byte aligned
 */
size_t mvx_soft_argmax_3d_workspace_bytes(int64_t n_frames, int n_volumes, int sx, int sy, int sz);
int mvx_soft_argmax_3d(const float* volumes, const float* coords_volume, int64_t n_frames, int n_volumes, int sx,
                       int sy, int sz, float multiplier, int softmax, float* out, float* volumes_out,
                       void* workspace, size_t workspace_bytes, void* stream);

/*
 * mvx_soft_argmax_3d with the voxel positions given by the parameters of a regular grid instead of a coordinate
 * volume: the grid VolumetricTriangulationNet.forward builds (mvn/models/triangulation.py:341-385) is evaluated inside
 * the kernel, and the 12 bytes per voxel of that volume are never read.  Without volumes_out, or without softmax,
 * the positions are evaluated per voxel in the float32 operation order of mvx_build_coord_volumes and the result is
 * bit-identical to mvx_soft_argmax_3d fed the volume that mvx_build_coord_volumes writes for the same parameters.
 * With softmax and volumes_out (the one-read kernel) the expectation of the voxel index is taken in float64 and mapped
 * through the grid once per volume: the positions it stands for are the exact ones, not their float32 roundings, so
 * the coordinates differ from the tensor-fed result by at most the rounding of one position: one float32 unit in the
 * last place of the grid's largest position along the axis (1.2e-4 mm for grids reaching 1-2 m).  Grid arguments as in mvx_build_coord_volumes (s0, s1, s2 = sx, sy, sz).
 */
int mvx_soft_argmax_3d_grid(const float* volumes, const float* position, const float* center, const float* rotation,
                            float step0, float step1, float step2, int swap_yz_flip, int64_t n_frames, int n_volumes,
                            int sx, int sy, int sz, float multiplier, int softmax, float* out, float* volumes_out,
                            void* workspace, size_t workspace_bytes, void* stream);

/*
 * mvx_soft_argmax_3d for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference: same operation and
 * same results as mvx_soft_argmax_3d (mvn/utils/op.py:84-96, 50-81), but the partial records live in a fixed
 * MVX_SOFT_ARGMAX_3D_STATIC_BYTES array inside the library (one copy per GPU, held until the process ends)
 * instead of a caller-owned workspace; the call allocates and frees nothing.  Shapes whose
 * mvx_soft_argmax_3d_workspace_bytes() exceed that size are rejected with MVX_ERR_WORKSPACE (17 volumes of
 * 64^3: up to 12 frames per call).  All calls on one GPU share the array: calls that overlap in time on one
 * GPU (different streams or threads) race; calls on one stream, or on different GPUs, do not.
 * The first call on a device looks up the array's address and the kernel's occupancy (which may load code objects):
 * make one call per device outside any stream capture before capturing it into a graph.
 */
#define MVX_SOFT_ARGMAX_3D_STATIC_BYTES (8u << 20)
int mvx_soft_argmax_3d_static(const float* volumes, const float* coords_volume, int64_t n_frames, int n_volumes,
                              int sx, int sy, int sz, float multiplier, int softmax, float* out, float* volumes_out,
                              void* stream);

/*
 * Backward of mvx_unproject: what torch autograd computes when the reference trains through
 * mvn/utils/op.py:99-166 (train.py:281-290) -- the transpose of the bilinear sampling of every view (no gradient
 * through taps outside a map, nor for voxels behind a camera, op.py:121,141), times the derivative of the view
 * aggregation: 1 (sum), vol_conf (confidences), the first view that holds the maximum (max), or
 * s_v (1 + x_v - out) with s the softmax over the views' samples x (softmax).  No gradient is produced for proj or
 * coords.
 * features_cl       [n_frames][n_views][h][w][Cp]     the forward call's maps, channel-last, Cp = n_channels rounded
 *                                                     up to a multiple of 4 (padding = 0), 16-byte aligned; may be
 *                                                     NULL for MVX_AGG_SUM, and for MVX_AGG_CONF without grad_vol_conf
 * proj, coords, vol_conf                              as in mvx_unproject
 * grad_volumes      [n_frames][n_channels][vx][vy][vz]
 * grad_features_cl  [n_frames][n_views][h][w][Cp]     out, channel-last like features_cl; every element written
 * grad_vol_conf     [n_frames][n_views][n_channels]   out, MVX_AGG_CONF only (sum over the voxels of gradient x
 *                                                     sample); written by the call; may be NULL
 * workspace         mvx_unproject_backward_workspace_bytes() bytes of device memory, 16-byte aligned: for
 *                   MVX_AGG_SOFTMAX, the derivative of the aggregate with respect to each view's sample, one float
 *                   per (view, channel, voxel) -- the size of the per-view volumes of the reference, 134 MB per frame
 *                   for 4 views x 32 channels x 64^3; for MVX_AGG_MAX the first maximal view per (channel, voxel);
 *                   for MVX_AGG_CONF the partial sums behind grad_vol_conf (about 1 KB per frame, view and 2048
 *                   voxels); 0 bytes for MVX_AGG_SUM (NULL allowed)
 * Slices of the gradient (two channels of one view's map; larger maps in bands of rows) are accumulated in LDS in
 * float64 and written once: no atomics in global memory, and two runs agree except for float64 rounding before the
 * final conversion.  Shapes that path does not take (rows of more than about 9 700 pixels; with grad_vol_conf more than
 * 64 channels or 16 views) use float32 atomic additions in global memory in no fixed order.
 */
size_t mvx_unproject_backward_workspace_bytes(int64_t n_frames, int n_views, int n_channels, int h, int w, int vx, int vy,
                                              int vz, int aggregation);
int mvx_unproject_backward(const float* features_cl, const float* proj, const float* coords, const float* vol_conf,
                           const float* grad_volumes, int64_t n_frames, int n_views, int n_channels, int h, int w, int vx,
                           int vy, int vz, int aggregation, float* grad_features_cl, float* grad_vol_conf,
                           void* workspace, size_t workspace_bytes, void* stream);

/*
 * Backward of mvx_soft_argmax_3d: what torch autograd computes through mvn/utils/op.py:84-96 (coords_volume != NULL)
 * or mvn/utils/op.py:50-81 (coords_volume == NULL: voxel indices, and for softmax == 0 the division by the mass).
 * With p = volumes_out of the forward call and G_i = grad_volumes_out_i + grad_coordinates . X_i:
 *   softmax != 0: grad_volumes_i = multiplier * p_i * (G_i - sum_j p_j G_j / sum_j p_j) (the float32 p add up to 1 only
 *   to rounding, and G carries the large common offset grad_coordinates . X);  softmax == 0: multiplier * G_i where
 *   p_i > 0 (index space: with X_i - out in place of X_i, divided by sum p).  Sums and G are float64.
 * volumes_out        [n_frames][n_volumes][sx][sy][sz]  what the forward call wrote
 * coords_volume      [n_frames][sx][sy][sz][3] or NULL
 * grad_coordinates   [n_frames][n_volumes][3] or NULL (= zero)
 * grad_volumes_out   [n_frames][n_volumes][sx][sy][sz] or NULL (= zero)
 * grad_volumes       [n_frames][n_volumes][sx][sy][sz]  out: gradient of the forward call's `volumes`
 * workspace          mvx_soft_argmax_3d_backward_workspace_bytes() bytes of device memory, 8-byte aligned (partial sums
 *                    of the segments a volume is cut into; not used for softmax == 0 with coords_volume)
 */
size_t mvx_soft_argmax_3d_backward_workspace_bytes(int64_t n_frames, int n_volumes, int sx, int sy, int sz);
int mvx_soft_argmax_3d_backward(const float* volumes_out, const float* coords_volume, int64_t n_frames, int n_volumes,
                                int sx, int sy, int sz, float multiplier, int softmax, const float* grad_coordinates,
                                const float* grad_volumes_out, float* grad_volumes, void* workspace,
                                size_t workspace_bytes, void* stream);

/*
 * Coordinate volumes of VolumetricTriangulationNet.forward: replaces the per-frame meshgrid / scale / rotate
 * loop of mvn/models/triangulation.py:341-385 (and mvn/utils/volumetric.py:102-114) with one launch.
 * For frame b and voxel (i0, i1, i2):  g = position[b] + step * (i0, i1, i2);  out = rotation[b] . (g - center[b])
 * + center[b], every step in float32 in that order (bit-identical to the reference when rotation is the
 * identity).  swap_yz_flip != 0 applies the reference's CMU-to-Human3.6M transfer (axes 1 and 2 swapped, new
 * axis 1 reversed; needs s1 == s2).
 * position  [n_frames][3]     minimum corner of the cuboid
 * center    [n_frames][3]     rotation centre (the base point)
 * rotation  [n_frames][3][3]  row-major
 * out       [n_frames][s0][s1][s2][3]
 */
int mvx_build_coord_volumes(const float* position, const float* center, const float* rotation, int64_t n_frames,
                            int s0, int s1, int s2, float step0, float step1, float step2, int swap_yz_flip,
                            float* out, void* stream);

/*
 * Integer keypoints at the heatmap maxima, scaled to image pixels: replaces the arg-max and the upscale of
 * RANSACTriangulationNet.forward, mvn/models/triangulation.py:48-56.  Per map the flat index of the maximum
 * (torch.max ordering: NaN first, then the largest value; the lowest index among equals) is split into
 * x = index % w, y = index / w, and each is multiplied in float32 by its scale and truncated toward zero.
 * heatmaps   [n_maps][h][w]
 * keypoints  [n_maps][2] int64 (x, y)
 */
int mvx_heatmap_argmax(const float* heatmaps, int64_t n_maps, int h, int w, float scale_x, float scale_y,
                       int64_t* keypoints, void* stream);

/*
 * RANSAC triangulation of every (frame, joint): replaces RANSACTriangulationNet.triangulate_ransac and the
 * loop that calls it, mvn/models/triangulation.py:67-73, 83-148, with mvn/utils/multiview.py:115-140 (DLT) and
 * :201-208 (reprojection error, HALF the pixel distance) inside.  For each of the n_iters view pairs: DLT from
 * the two views, inliers = the pair plus every view whose reprojection error is below epsilon; the first largest
 * inlier set wins; DLT over it; with direct_optimization != 0 the point is then moved to the minimiser of
 * 1/2 sum huber(error^2) over the inlier views (what scipy's least_squares(loss='huber') approaches from the
 * same start).  All arithmetic in float64.  The pairs come from the caller: the reference draws them with
 * Python's random.sample, and results match it only for the same draws.
 * proj          [n_frames][n_views][3][4]
 * points        [n_frames][n_views][n_joints][2]  float64 keypoints
 * view_pairs    [n_frames][n_joints][n_iters][2]  view indices; a pair with an index >= n_views is skipped
 * out           [n_frames][n_joints][3]           float64
 * inlier_masks  [n_frames][n_joints]              bit v set = view v is an inlier (n_views <= 32)
 */
int mvx_triangulate_ransac(const float* proj, const double* points, const uint8_t* view_pairs, int64_t n_frames,
                           int n_views, int n_joints, int n_iters, double epsilon, int direct_optimization,
                           double* out, uint32_t* inlier_masks, void* stream);

/*
 * Savitzky-Golay smoothing of time series: replaces SAVGOLFilter.__call__, savgol_filer.py:32-70
 * (scipy.signal.savgol_filter(x, window, polyorder, axis=0), default mode 'interp', applied per coordinate).
 * hat is the window x window hat matrix A pinv(A) of the polynomial fit (A = Vandermonde matrix of the window
 * positions, polyorder + 1 columns), float64, device memory: interior samples use its middle row on the window
 * centred on them, the first and last window/2 samples use rows 0.. / ..window-1 on the first / last window.
 * window must be odd, or equal to n_steps (the reference cuts the window to a short sequence).
 * x, out  [n_steps][n_series]   (a [T][K][C] pose tensor is n_series = K*C)
 */
int mvx_savgol_filter(const float* x, int64_t n_steps, int64_t n_series, const double* hat, int window, float* out,
                      void* stream);

/*
 * Joint order and axis convention of the real-time pipelines' wire format: replaces the loop of
 * real_pipeline_trt.py:371-396.  out[f][j] = scale * (-x, z, -y) of points[f][joint_order[j]], float32.
 * points, out  [n_frames][n_joints][3];  joint_order  [n_joints] int32, device memory
 */
int mvx_unity_points(const float* points, const int32_t* joint_order, int64_t n_frames, int n_joints, float scale,
                     float* out, void* stream);

/*
 * The per-frame tail of the real-time pipelines in one launch: replaces real_pipeline_trt.py:349-396 for one
 * frame.  `ring` keeps the last n_steps poses on the device as a ring of slots; the call stores new_pose in slot
 * ring_pos (the caller advances ring_pos = (ring_pos + 1) % n_steps per frame), so time runs over slots
 * ring_pos + 1, ..., ring_pos (mod n_steps).
 *   filtered == 0 (the pipeline's first slide_window_size frames): out[0] = new_pose mapped as by mvx_unity_points.
 *   filtered != 0: the n_steps poses are smoothed as by mvx_savgol_filter; out[0] is the mean of the last two
 *   smoothed poses, out[1] the last smoothed pose, both mapped as by mvx_unity_points.
 * ring  [n_steps][n_joints][3] (read and written);  new_pose  [n_joints][3];  out  [2][n_joints][3]
 */
int mvx_realtime_tail(float* ring, const float* new_pose, int ring_pos, int n_steps, int n_joints, const double* hat,
                      int window, const int32_t* joint_order, float scale, int filtered, float* out, void* stream);

/*
 * HOST function (no device work): fills view_pairs with the sorted pairs that n_draws successive calls of
 * Python's random.sample(range(n_views), 2) return, reading and advancing the Mersenne Twister state of Python's
 * `random` module (random.getstate()[1]: 624 words and the position).  This is how RANSACTriangulationNet draws its
 * hypotheses, mvn/models/triangulation.py:96-97; doing it here avoids one interpreter call per draw.
 * mt_state    [624] in/out      mt_pos  in/out (0..624)      view_pairs  [n_draws][2], first < second
 */
int mvx_draw_view_pairs(uint32_t* mt_state, int32_t* mt_pos, int n_views, int64_t n_draws, uint8_t* view_pairs);

#ifdef __cplusplus
}
#endif
#endif /* MVX_H */
