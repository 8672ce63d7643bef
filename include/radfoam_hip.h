/*
 * radfoam_hip.h -- C-ABI of the MI355X-native Voronoi ray tracer (libradfoam_hip.so).
 *
 * This is the drop-in boundary for the reference's native tracing library: every entry
 * point replaces one method of radfoam::Pipeline / one free function of
 * /root/reference/src/tracing/pipeline.h (cited per function).  Plain pointers and sizes
 * only: all pointers are DEVICE pointers (tensor.data_ptr()) unless marked HOST, `stream`
 * is a hipStream_t passed as void* (NULL = default stream).  No torch types.
 *
 * Where the reference selects a template instance <attr_scalar, sh_degree> through
 * create_pipeline() (pipeline.cu:776-805), these functions take (sh_degree, attr_type).
 *
 * Every function returns RF_OK (0) or a negative rf_status and records a message that
 * rf_last_error() returns (thread-local).  Launch errors (hipGetLastError) are reported
 * the same way, like cuda_check() after launch_kernel_1d (src/utils/common_kernels.cuh:40).
 * Nothing synchronises the device.
 */
#ifndef RADFOAM_HIP_H
#define RADFOAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rf_status {
    RF_OK = 0,
    RF_ERR_INVALID_ARGUMENT = -1, /* unsupported sh_degree / attr_type / null pointer / bad size */
    RF_ERR_WORKSPACE = -2,        /* workspace missing or too small                               */
    RF_ERR_LAUNCH = -3            /* HIP runtime error while launching                            */
} rf_status;

/* attribute scalar type: radfoam::ScalarType subset used by create_pipeline (typing.h:23-30) */
typedef enum rf_attr_type { RF_ATTR_FLOAT32 = 0, RF_ATTR_FLOAT16 = 1 } rf_attr_type;

/* radfoam::TraceSettings, src/tracing/pipeline.h:10-13 (defaults :15-20: 1e-3, 1024) */
typedef struct rf_trace_settings {
    float weight_threshold;
    uint32_t max_intersections;
} rf_trace_settings;

/* radfoam::Camera by value, src/tracing/camera.h:17-26 (HOST struct) */
typedef struct rf_camera {
    float position[3];
    float forward[3];
    float right[3];
    float up[3];
    float fov;       /* vertical, radians */
    uint32_t width;
    uint32_t height;
    uint32_t model;  /* 0 = Pinhole, 1 = Fisheye (camera.h:12-15) */
} rf_camera;

/* Launch options that have no counterpart in the reference (all optional; zero-init = defaults). */
typedef struct rf_launch_opts {
    void *workspace;          /* device scratch, >= rf_workspace_bytes(); holds the packed foam      */
    size_t workspace_bytes;
    uint32_t foam_prepared;   /* 1: workspace already holds rf_prepare_foam() output for these inputs */
                              /*   (for rf_trace_benchmark: prepared WITH the same adjacent_diff)     */
                              /* 2: it holds the output for the same ADJACENCY, but points and/or     */
                              /*   attributes changed: only cells and face offsets are repacked       */
                              /*   (rf_trace_forward / rf_trace_backward; what an optimiser step      */
                              /*   between two triangulation rebuilds needs)                          */
    uint32_t image_width;     /* rays form a row-major [image_height, image_width] grid: lets a wave  */
    uint32_t image_height;    /*   own an 8x8 pixel tile.  0,0 = treat rays as a flat list            */
    uint32_t backward_mode;   /* rf_trace_backward only: 0 = auto, 1 = per-lane atomics, 2 = wave     */
                              /*   pre-reduced atomics, 3 = block-level LDS write-combining, 4 =      */
                              /*   direct row-coalesced atomics (3, 4 need the trail and fall back to */
                              /*   2 without it; auto = 3 for image-shaped batches, 4 for flat ones)  */
    uint64_t *stats;          /* optional device uint64[8]: walk counters for the roofline figure     */
                              /*   [0] cells scanned [1] faces scanned [2] hops [3] segments          */
                              /*   [4] lit segments; accumulated with atomics, caller zeroes          */
    /* Hop trail (optional): rf_trace_forward records, for every ray, the cell each hop entered;     */
    /* rf_trace_backward given the SAME buffers (and the same foam, rays, start cells, quantiles and  */
    /* settings) replays it instead of re-scanning every cell's faces.  Rays with more than trail_cap */
    /* hops are walked again by scanning, so any trail_cap >= 1 is correct.                           */
    uint32_t *trail;          /* device uint32[trail_cap][trail_slots]                                */
    uint32_t *trail_hops;     /* device uint32[trail_slots]                                           */
    uint32_t trail_cap;
    uint32_t trail_slots;     /* >= rf_trail_slots(num_rays, image_width, image_height)               */
    /* Flat ray lists only (image_width = 0), optional: device uint32[num_rays], a permutation of the  */
    /* ray indices (rf_build_ray_order).  Thread slot s traces ray ray_order[s], so that the rays of a */
    /* wave / block are neighbours in space even when the caller's batch is shuffled (train.py:61);    */
    /* every input and output stays indexed by the caller's ray index.  The same order must be given   */
    /* to the rf_trace_backward call that replays a trail.                                             */
    const uint32_t *ray_order;
    /* rf_trace_forward with `stats` only, optional: device uint8[num_points]; byte i is set to 1 when  */
    /* any ray scans cell i (the caller zeroes it).  Feeds the compulsory-traffic floor of bench.py's   */
    /* roofline: bytes of the distinct cells, face lists and colour rows a frame touches at least once. */
    uint8_t *visit_marks;
    uint32_t forward_mode;    /* HOW the scans of a launch are scheduled; every mode returns the same results, bit for   */
                              /*   bit: the reference's scan (tracing_utils.cuh:43-67 -- the face with the smallest ROUNDED */
                              /*   quotient t = ((P + o/2) - O).o / o.d among those with o.d > 0, the lowest index among    */
                              /*   equal quotients).  0 = auto, 1 = the face scan requests a cell's blocks one at a time    */
                              /*   (six waves per SIMD hide the latency: large image launches), 2 = the first six blocks    */
                              /*   are requested together at the hop that enters the cell (four waves per SIMD: flat        */
                              /*   batches, launches of at most 1024 blocks).  0..2, 4, 5 find the exit by a tournament on   */
                              /*   cross-multiplied products with a certificate and divide only the winner (cells whose      */
                              /*   certificate fails -- two exits within 3 floats -- are scanned again by the dividing scan); */
                              /*   the certificate's proof needs the compared quotients to be NORMAL floats (it is exact     */
                              /*   for any inputs whose exit distances are not subnormal: nonzero coordinates of magnitude   */
                              /*   >= 2^-40 suffice), and the library does not rely on the caller for that: a cell whose      */
                              /*   winning quotient is zero or subnormal goes to the dividing scan as well (fail-safe),      */
                              /*   so the result is the reference's for EVERY input the reference itself defines;            */
                              /*   3 = every face of every cell divided, as the reference writes it: the independent         */
                              /*   instance the others are tested against, 40 % slower.  rf_trace_benchmark honours 3;       */
                              /*   rf_trace_backward accepts and ignores the field (a trail replays under any mode).         */
                              /*   4 (rf_trace_forward, experiment) = persistent waves that refill their dead lanes from  */
                              /*   a queue by ballot + prefix count -- slower on every workload measured.  The queue head   */
                              /*   is a word of the workspace: two mode-4 launches that share a workspace must be on ONE    */
                              /*   stream.  With depth quantiles or `stats` the launch falls back to mode 1.                */
                              /*   5 = mode 2 behind a block-level LDS table of cell records + face blocks (what auto     */
                              /*   picks for a flat batch given with a ray_order: its 256-slot groups re-visit cells).    */
    /* Optional: device uint32[rf_launch_blocks(...)], the tile each block of the launch walks -- a 16x16-pixel tile of   */
    /* an image-shaped batch, a group of 256 consecutive thread slots of a flat one (values >= the number of tiles: the   */
    /* block owns no rays).  It MUST name every tile exactly once; the library does not check it (the table lives on the */
    /* device).  A tile that is left out is not traced: its rays' outputs (rgba, depths, num_intersections, trail hop     */
    /* counts) keep whatever the buffers held; a tile named twice is walked twice, which is harmless in the forward and   */
    /* DOUBLES its rays' gradients in rf_trace_backward.  Any permutation gives the same results; the order decides which */
    /* blocks are still running when the launch drains.  Default: tiles dealt to the XCDs in strips (dealt_tile).         */
    const uint32_t *tile_order;
    /* rf_trace_forward / rf_trace_benchmark, optional: device uint32[number of tiles] (image-shaped batches: 16x16-pixel  */
    /* tiles, row-major; flat batches: groups of 256 thread slots); entry t is raised (atomic max) to the step count of   */
    /* tile t's longest ray.  The caller zeroes it.  What a tile_order for the next launches over the same rays is built  */
    /* from.                                                                                                            */
    uint32_t *tile_cost;
    /* rf_trace_backward, optional: floats between two rows of attribute_grad (0 = the attribute dimension A, i.e. the     */
    /* reference's dense [N][A] layout).  A gradient row leaves the kernels as ONE atomic instruction whose lanes are    */
    /* the row's columns, and the memory side works per (instruction, 64-byte line): rows of A = 13 / 28 / 49 floats      */
    /* start at every 4-byte offset and span 1.75 / 2.6 / 3.9 lines on average; with a pitch of 16 / 32 / 64 floats and a */
    /* 64-byte aligned buffer they span 1 / 2 / 3 -- up to a quarter fewer requests for a backward that is bound by them  */
    /* (flat batches whose every segment is lit).  Columns A .. pitch-1 are never written.                                */
    uint32_t attr_grad_pitch;
} rf_launch_opts;

/* Last error message of the calling thread ("" if none). */
const char *rf_last_error(void);

/* Pipeline::attribute_dim(), pipeline.cu:768-770: 1 + 3*(d+1)^2, or 0 if d is unsupported. */
uint32_t rf_attribute_dim(int sh_degree);

/* Thread slots a launch over these rays uses (trail buffers are indexed by slot). */
uint32_t rf_trail_slots(uint32_t num_rays, uint32_t image_width, uint32_t image_height);

/* Blocks a launch over these rays has (the length of rf_launch_opts.tile_order) and, with `tiles` non-null (host
 * memory, that many entries), the default assignment: tiles[b] = the tile block b walks (>= the number of tiles:
 * none); block b runs on XCD b % 8. */
uint32_t rf_launch_blocks(uint32_t num_rays, uint32_t image_width, uint32_t image_height, uint32_t *tiles);

/* Bytes of device scratch the tracer needs for a foam of this size: 16-byte cell records,
 * the 16-byte-per-face table (fp16 neighbour offsets as in the reference's half4 table, plus
 * the neighbour index and its face range) with the reference's +32 entries of slack, and
 * repacked SH rows when the attribute row is not 16-byte aligned.  Replaces the per-call
 * CUDAArray<Vec4h>(E+32) of pipeline.cu:613,667.  Always size the workspace by calling this function with the
 * library that will use it: the figure is not part of the ABI (round 5 added 256 bytes for forward_mode 4's queue head). */
size_t rf_workspace_bytes(uint32_t num_points, uint32_t point_adjacency_size, int sh_degree,
                          int attr_type);

/* prefetch_adjacent_diff(), src/tracing/pipeline.h:46-53 / pipeline.cu:546-586:
 * adjacent_diff[e] = half4(points[adj[e]] - points[owner(e)], 0), RNE.  adjacent_diff holds
 * point_adjacency_size entries of 8 bytes. */
int rf_build_adjacent_diff(const float *points, uint32_t num_points,
                           uint32_t point_adjacency_size, const uint32_t *point_adjacency,
                           const uint32_t *point_adjacency_offsets, void *adjacent_diff,
                           void *stream);

/* Packs the foam into `workspace` for the walk kernels: cell records {x,y,z,density}, the face
 * table (fp16 offsets with the same values as rf_build_adjacent_diff, or taken verbatim from
 * `adjacent_diff` when that is not NULL) and aligned SH rows.
 * The reference does the equivalent (prefetch_adjacent_diff) inside every trace_forward /
 * trace_backward call (pipeline.cu:613-620,667-674); here the result may be reused while
 * points/attributes/adjacency are unchanged (opts->foam_prepared).  rf_trace_backward's trail
 * replay derives the fp16 offsets from the points, so a workspace packed from a caller-supplied
 * `adjacent_diff` must only be reused by rf_trace_benchmark. */
int rf_prepare_foam(int sh_degree, int attr_type, uint32_t num_points, const float *points,
                    const void *attributes, uint32_t point_adjacency_size,
                    const uint32_t *point_adjacency, const uint32_t *point_adjacency_offsets,
                    const void *adjacent_diff, void *workspace, size_t workspace_bytes,
                    void *stream);

/* Pipeline::trace_forward, src/tracing/pipeline.h:62-78 (kernel: pipeline.cu:14-130).
 * rays: [num_rays][6] (origin, direction; direction is normalised in the kernel).
 * ray_rgba: [num_rays][4] attr type.  quantile_depths / quantile_point_indices:
 * [num_rays][num_depth_quantiles] (NULL iff depth_quantiles NULL).  num_intersections may be
 * NULL.  point_contribution: [num_points] FLOAT32 for either attr type (accumulated with fp32
 * atomics; the host mirror rounds to the attr type once), zero-filled by the caller, or NULL. */
int rf_trace_forward(int sh_degree, int attr_type, const rf_trace_settings *settings,
                     uint32_t num_points, const float *points, const void *attributes,
                     uint32_t point_adjacency_size, const uint32_t *point_adjacency,
                     const uint32_t *point_adjacency_offsets, uint32_t num_rays,
                     const float *rays, const uint32_t *start_point_index,
                     uint32_t num_depth_quantiles, const float *depth_quantiles, void *ray_rgba,
                     float *quantile_depths, uint32_t *quantile_point_indices,
                     uint32_t *num_intersections, void *point_contribution,
                     const rf_launch_opts *opts, void *stream);

/* Pipeline::trace_backward, src/tracing/pipeline.h:80-100 (kernel: pipeline.cu:132-343).
 * points_grad [num_points][3], attribute_grad [num_points][A] and point_error [num_points]
 * (optional) are FLOAT32 accumulators for either attr type (the host mirror rounds to the attr
 * type once); they must be zero-filled by the caller (the reference binding does so,
 * pipeline_bindings.cpp:441-452) and are accumulated into.  ray_grad is accepted for signature
 * parity and, as in the reference, never written. */
int rf_trace_backward(int sh_degree, int attr_type, const rf_trace_settings *settings,
                      uint32_t num_points, const float *points, const void *attributes,
                      uint32_t point_adjacency_size, const uint32_t *point_adjacency,
                      const uint32_t *point_adjacency_offsets, uint32_t num_rays,
                      const float *rays, const uint32_t *start_point_index,
                      uint32_t num_depth_quantiles, const float *depth_quantiles,
                      const uint32_t *quantile_point_indices, const void *ray_rgba,
                      const void *ray_rgba_grad, const float *depth_grad, const void *ray_error,
                      float *ray_grad, float *points_grad, void *attribute_grad,
                      void *point_error, const rf_launch_opts *opts, void *stream);

/* Pipeline::trace_benchmark, src/tracing/pipeline.h:117-126 (kernel: pipeline.cu:472-544).
 * adjacent_diff is the CALLER's half4 table of point_adjacency_size entries (benchmark.py:44-54);
 * camera is a HOST struct; start_point_index points at ONE device uint32; ray_rgba is
 * uint32[height*width], RGBA8 packed as make_rgba8 (tracing_utils.cuh:105-115).
 * The reference reads up to 3 table entries past the end of the last cell
 * (tracing_utils.cuh:43-50); this implementation never reads past point_adjacency_size: the
 * table is copied entry by entry into the packed face table (cached via opts->foam_prepared). */
int rf_trace_benchmark(int sh_degree, int attr_type, const rf_trace_settings *settings,
                       uint32_t num_points, const float *points, const void *attributes,
                       uint32_t point_adjacency_size, const uint32_t *point_adjacency,
                       const uint32_t *point_adjacency_offsets, const void *adjacent_diff,
                       const rf_camera *camera, const uint32_t *start_point_index,
                       uint32_t *ray_rgba, const rf_launch_opts *opts, void *stream);

/* A cost prior for the 16x16-pixel tiles of a frame that has never been traced (no counterpart in the reference, whose
 * launches take pixels in index order; benchmark.py:95-139 renders another camera every frame, so the step counts a
 * previous trace measured -- rf_launch_opts.tile_cost -- do not exist for it).  rf_build_cost_grid fills `grid`
 * (rf_cost_grid_bytes(res) bytes of device memory) with a res^3 voxel grid over the points' bounding box: per voxel the
 * cells a line crosses per unit length (1.455 n^(1/3), n = points per unit volume) and the mean density (last attribute).
 * rf_estimate_tile_cost marches five rays of every tile through it -- rays: float[height][width][6] on the device, or
 * rays == NULL and `camera` (host struct, as for rf_trace_benchmark) -- until the transmittance exp(-optical depth) falls
 * below weight_threshold, the ray leaves the box or max_intersections steps are estimated, and writes the estimated
 * steps of each tile's longest ray to tile_cost[number of tiles] (row-major tiles, the layout of rf_launch_opts.tile_cost).
 * The host turns that into a rf_launch_opts.tile_order; any order gives the same results. */
size_t rf_cost_grid_bytes(uint32_t res);

int rf_build_cost_grid(const float *points, const void *attributes, int attr_type, uint32_t attr_dim, uint32_t num_points,
                       uint32_t res, void *grid, size_t grid_bytes, void *stream);

int rf_estimate_tile_cost(const void *grid, uint32_t res, const float *rays, const rf_camera *camera, uint32_t width,
                          uint32_t height, float weight_threshold, uint32_t max_intersections, uint32_t *tile_cost,
                          void *stream);

/* Block -> tile tables (rf_launch_opts.tile_order) from a cost map (rf_launch_opts.tile_cost of a forward, or
 * rf_estimate_tile_cost), built on the device in one launch; no counterpart in the reference.  Up to two tables per call
 * (order_b may be NULL), each uint32[rf_launch_blocks(num_rays, image_width, image_height)]; image_width = 0: a flat batch
 * of num_rays rays (its 256-slot groups).  rule 1, param q: every XCD keeps the tiles the static dealing gives it and takes
 * them longest first in classes of q steps (static order within a class); rule 2, param n: the static order, except that
 * the n cheapest tiles of the frame come last, the longest of them first.  At most 65536 blocks. */
int rf_build_tile_orders(const uint32_t *tile_cost, uint32_t num_rays, uint32_t image_width, uint32_t image_height,
                         uint32_t rule_a, uint32_t param_a, uint32_t *order_a, uint32_t rule_b, uint32_t param_b,
                         uint32_t *order_b, void *stream);

/* A frame may take the order ANOTHER frame's trace measured only when it shows nearly the same picture (a camera path).
 * rf_tile_order_reference records five rays of the frame an order was learnt on into reference[40] (device floats; the
 * frame is given as for rf_estimate_tile_cost -- a ray tensor with one start_point_index per ray, or `camera` with ONE);
 * rf_gate_tile_order compares the same five rays of a new frame with it -- angle between the directions plus the shift of
 * the origin relative to its distance from the entry cell's point, at most max_angle_radians for every sample -- and
 * writes `order` = learnt_order if so, the static dealing if not, plus (optional) *verdict = 1 / 0; all on the device. */
int rf_tile_order_reference(const float *rays, const rf_camera *camera, const uint32_t *start_point_index, const float *points,
                            uint32_t image_width, uint32_t image_height, float *reference, void *stream);

int rf_gate_tile_order(const float *rays, const rf_camera *camera, const float *reference, uint32_t image_width,
                       uint32_t image_height, float max_angle_radians, const uint32_t *learnt_order, uint32_t *order,
                       uint32_t *verdict, void *stream);

/* Number of adjacency entries of a foam = point_adjacency_offsets[num_points], read back from the device (one
 * 4-byte copy on `stream`, which is synchronised).  Pipeline::trace_benchmark (src/tracing/pipeline.h:117-126)
 * receives no adjacency size, while rf_workspace_bytes() / rf_trace_benchmark() need it on the host: a
 * reference-side binding asks here (once per scene: the value only changes with the triangulation). */
int rf_adjacency_size(uint32_t num_points, const uint32_t *point_adjacency_offsets,
                      uint32_t *point_adjacency_size /* HOST */, void *stream);

/* Rounds an fp32 accumulator into a buffer of the pipeline's attribute type: dst[i] = (attr_type) src[i] for
 * i < count (fp16: round to nearest even; fp32: a copy).  The reference's `point_contribution`,
 * `attribute_grad` and `point_error` buffers have the attribute type (pipeline.h:62-100) while this library
 * accumulates them in fp32 for both types; an fp16 binding accumulates into fp32 scratch and finishes with this. */
int rf_cast_accumulator(const float *src, size_t count, int attr_type, void *dst, void *stream);

/* The part of rf_prepare_foam that depends on points / attributes: rewrites cell records, fp16 face
 * offsets and SH rows of a workspace whose adjacency-derived part (padded offsets, links) was
 * packed by rf_prepare_foam for the same point_adjacency / offsets (without adjacent_diff). */
int rf_prepare_foam_geometry(int sh_degree, int attr_type, uint32_t num_points, const float *points,
                             const void *attributes, uint32_t point_adjacency_size, void *workspace,
                             size_t workspace_bytes, void *stream);

/* ---- the callers on either side of the tracer -------------------------------- */

/* RadFoamScene.get_trace_data, radfoam_model/scene.py:202-217, in one pass:
 *   attributes[i] = [ att_dc[i][0..3) | att_sh[i][0..A-4) | activation_scale * softplus(density[i], beta=10) ]
 * cast to the pipeline's attribute type (fp32, or fp16 RNE).  att_dc [N][3], att_sh [N][A-4]
 * (may be NULL when A == 4), density [N] are fp32 device arrays; attributes is [N][A]. */
int rf_pack_attributes(int sh_degree, int attr_type, uint32_t num_points, const float *att_dc,
                       const float *att_sh, const float *density, float activation_scale,
                       void *attributes, void *stream);

/* Its backward: attr_grad is the fp32 [N][A] gradient trace_backward produced; writes
 * att_dc_grad [N][3], att_sh_grad [N][A-4] and density_grad[i] = attr_grad[i][A-1] *
 * activation_scale * d softplus(density[i]) (evaluated as torch's softplus_backward does). */
int rf_pack_attributes_backward(int sh_degree, uint32_t num_points, const float *density,
                                float activation_scale, const float *attr_grad, float *att_dc_grad,
                                float *att_sh_grad, float *density_grad, void *stream);

/* radfoam.nn for a few queries (the entry cell of each camera: scene.py:224-234, benchmark.py:89;
 * reference: triangulation_bindings.cpp:142-181 over the AABB tree of aabb_tree.cu:343-415).
 * Exact nearest point by squared fp32 distance, the lowest index among exact ties; cost
 * O(num_points * num_queries).  scratch: num_queries * 8 bytes of device memory. */
int rf_nearest_point(const float *points, uint32_t num_points, const float *queries,
                     uint32_t num_queries, uint32_t *indices, void *scratch, void *stream);

/* radfoam.nn through the tree (torch_bindings/triangulation_bindings.cpp:142-181 over nn_kernel, src/aabb_tree/
 * aabb_tree.cu:343-415): `points` in kd-order, `aabb_tree` the float[pow2_round_up(num_points)][2][3] boxes of
 * rf_build_aabb_tree (or the reference's build_aabb_tree) over those points.  Same result as rf_nearest_point -- exact,
 * lowest index among exact ties -- in O(log num_points) boxes per query: the choice for many queries. */
int rf_nearest_point_tree(const float *points, uint32_t num_points, const float *aabb_tree, const float *queries,
                          uint32_t num_queries, uint32_t *indices, void *stream);

/* radfoam.farthest_neighbor, src/delaunay/triangulation_ops.cu:9-44: per point the first farthest
 * Delaunay neighbour (0xFFFFFFFF if none) and the mean half-distance to its neighbours. */
int rf_farthest_neighbor(const float *points, uint32_t num_points, const uint32_t *point_adjacency,
                         const uint32_t *point_adjacency_offsets, uint32_t *indices,
                         float *cell_radius, void *stream);

/* radfoam.BatchFetcher(data, batch_size, shuffle).next() for a DEVICE-resident array (torch_bindings.cpp:77-83 over
 * src/utils/batch_fetcher.cpp:60-77): out[j] = data[index(batch_index * batch_size + j)], rows of stride_bytes (a multiple
 * of 4); index = randint(make_rng(seq), 0, num_elements) of src/utils/random.h:13-57 when shuffle, seq % num_elements
 * otherwise -- the reference's sequence, so fetchers over rays / colours / alphas stay aligned.  num_elements < 2^32
 * (the reference's "Too many elements").  The reference gathers on a host thread and uploads; here the set stays in HBM. */
int rf_fetch_batch(const void *data, uint64_t num_elements, uint32_t stride_bytes, uint32_t batch_index,
                   uint32_t batch_size, int shuffle, void *out, void *stream);

/* One rank's share of a batch (data-parallel training, BASELINE config 4: "rays row-sharded across 8 GPUs"; the reference is
 * single-GPU, so this has no counterpart beyond batch_fetcher.cpp:60-70's sequence): out[j] = data[index(first_sequence + j)]
 * for j < count, with the index rule of rf_fetch_batch.  Rank r of W takes first_sequence = batch_index * batch_size +
 * r * batch_size / W and count = batch_size / W: the ranks' shares concatenated in rank order are the batch the reference's
 * single fetcher returns, so W ranks see exactly the rays one process would. */
int rf_fetch_batch_range(const void *data, uint64_t num_elements, uint32_t stride_bytes, uint64_t first_sequence,
                         uint32_t count, int shuffle, void *out, void *stream);

/* Coherent processing order for a flat batch of rays: sorts the ray indices by (entry cell, Morton
 * code of the direction on a 2^16 x 2^16 octahedral grid), so that 64 / 256 consecutive entries are
 * a compact patch of directions from one origin.  rays is float[num_rays][6] (direction need not be
 * normalised), start_point_index uint32[num_rays]; writes the permutation to ray_order.
 * No counterpart in the reference, whose kernels take the batch as it comes. */
size_t rf_ray_order_workspace_bytes(uint32_t num_rays);

int rf_build_ray_order(const float *rays, const uint32_t *start_point_index, uint32_t num_rays,
                       uint32_t *ray_order, void *workspace, size_t workspace_bytes, void *stream);

/* CSR point adjacency from tetrahedra (find_adjacency, src/delaunay/delaunay.cu:140-229): tets is
 * uint32[num_tets][4]; writes point_adjacency_offsets[num_points + 1], the neighbours of every
 * point in ascending order into point_adjacency (capacity 12 * num_tets entries) and their number
 * into the DEVICE word *point_adjacency_size.  Tets with an index >= num_points or a repeated
 * vertex contribute nothing. */
size_t rf_adjacency_workspace_bytes(uint32_t num_tets);
int rf_build_adjacency(const uint32_t *tets, uint32_t num_tets, uint32_t num_points,
                       uint32_t *point_adjacency, uint32_t *point_adjacency_offsets,
                       uint32_t *point_adjacency_size, void *workspace, size_t workspace_bytes,
                       void *stream);

/* ---- triangulation (the Delaunay build itself) ------------------------------------------------ */

/* radfoam.build_aabb_tree (torch_bindings/triangulation_bindings.cpp:117-140; build_aabb_tree,
 * src/aabb_tree/aabb_tree.cu:192-283): the implicit balanced tree over points that are already in kd-order.
 * aabb_tree: [pow2_round_up(num_points)][2][3] floats {min, max}; level d (2^d nodes, node k bounding the points
 * [k << (depth - d), (k + 1) << (depth - d))) starts at node 2^depth - 2^(d+1); indices past num_points repeat the
 * last point; the last entry is not written (as in the reference). */
int rf_build_aabb_tree(const float *points, uint32_t num_points, float *aabb_tree, void *stream);

/* sort_points (src/aabb_tree/aabb_tree.cu:62-190), the order Triangulation::rebuild puts the points in before it
 * builds the tree (delaunay.cu:316): with P = pow2_round_up(N), everything sorted by x, every consecutive P/2
 * segment by y, every P/4 segment by z, ... down to segments of 2, stable.  permutation[i] = index in `points` of the
 * i-th point of the order (Triangulation::permutation()); sorted_points[i] = points[permutation[i]]. */
size_t rf_kd_order_workspace_bytes(uint32_t num_points);
int rf_kd_order(const float *points, uint32_t num_points, uint32_t *permutation, float *sorted_points,
                void *workspace, size_t workspace_bytes, void *stream);

/* Triangulation::rebuild + point_adjacency()/point_adjacency_offsets() (src/delaunay/delaunay.h:17-45,
 * delaunay.cu:273-370): the Delaunay neighbour lists of `points` (kd-ordered, num_points >= 32 like the reference),
 * every list ascending, as find_adjacency (delaunay.cu:140-229) emits them.  Every point builds its own star
 * against `aabb_tree` (rf_build_aabb_tree of the same points) with exact predicates; see csrc/rf_star.hpp.
 *   seed_adjacency / seed_offsets  optional (both or neither): neighbour lists of a previous triangulation of the
 *       same point count -- the incremental = true case.  They only make the search cheaper; the result is the
 *       same triangulation either way.
 *   point_adjacency [adjacency_capacity], point_adjacency_offsets [num_points + 1]: device outputs.
 *   info: HOST array of 12 words written before returning (the call synchronises the stream, as the reference's
 *       rebuild does): [0] adjacency size E (the lists are complete only if E <= adjacency_capacity),
 *       [1] stars that failed (degenerate or cospherical neighbourhood), [2] stars that needed the large instance,
 *       [3] points that coincide with another point, [4] directed edges whose reverse is missing,
 *       [5..6] tree nodes visited (low, high word), [7] link insertions, [8..10] the failed stars of [1] by cause
 *       (no non-coplanar start, link no longer a sphere, beyond the limits below), [11] hull candidates.
 *       [1], [3] or [4] non-zero = what the
 *       reference reports by throwing TriangulationFailedError ("ambiguous triangulation", "duplicate points
 *       found"): the caller perturbs the points and retries (radfoam_model/scene.py:160-186).
 *   workspace: rf_delaunay_workspace_bytes(num_points) device bytes.  That holds result rows for one star in 32 in
 *       the second pass (the stars on the rim of the cloud, a few per thousand).  If a cloud needs more, the call
 *       returns RF_ERR_WORKSPACE with info[2] = how many; rf_delaunay_workspace_bytes_for(num_points, info[2]) is
 *       the size to come back with.
 *   Limits: 4095 neighbours per point, and at most 64 points with more than 249 (hubs inside empty shells).
 *   Environment (tuning / experiments, read at the call): RF_DELAUNAY_OWNER=1|2 selects the experimental two-pass
 *   build that certifies every tetrahedron once (it also enlarges the workspace by 760 B per point). */
size_t rf_delaunay_workspace_bytes(uint32_t num_points);
size_t rf_delaunay_workspace_bytes_for(uint32_t num_points, uint32_t second_pass_stars);
int rf_delaunay_adjacency(const float *points, uint32_t num_points, const float *aabb_tree,
                          const uint32_t *seed_adjacency, const uint32_t *seed_offsets,
                          uint32_t *point_adjacency, uint32_t adjacency_capacity,
                          uint32_t *point_adjacency_offsets, uint32_t *info, void *workspace,
                          size_t workspace_bytes, void *stream);

/* ---- multi-GPU gradient exchange (radfoam_amd/dist.py; no counterpart: the reference is single-GPU) ---- */

/* Floats per packed gradient row: 1 (cell index, as bits) + 3 (points_grad) + A (attr_grad), rounded up to
 * a multiple of 4 (16-byte rows). */
uint32_t rf_grad_row_pitch(uint32_t attr_dim);

/* Compacts the rows of the gradient buffers that hold anything: for every cell i with a non-zero value among
 * points_grad[i][0..3) and attr_grad[i][0..A), appends {i, points_grad[i], attr_grad[i]} to `packed`
 * ([capacity][rf_grad_row_pitch(A)] floats, row order unspecified).  *count (DEVICE uint32, zeroed by the
 * caller) receives the number of such rows even when it exceeds `capacity` (rows past the capacity are not
 * written: the caller re-runs with a larger buffer).  A row-sharded frame touches only the cells its own
 * rays cross, so ranks exchange these rows instead of all-reducing the dense N*(3+A) buffer. */
int rf_compact_grad_rows(const float *points_grad, const float *attr_grad, uint32_t num_points,
                         uint32_t attr_dim, uint32_t capacity, uint32_t *count, float *packed, void *stream);

/* Applies `num_rows` packed rows to the dense buffers: mode 0 adds every row to points_grad / attr_grad at its
 * cell index; mode 1 zeroes those rows instead.  The cell indices of one call must be distinct (they are: one
 * rank's compaction lists a cell once), so calls issued in rank order give the same sums on every rank. */
int rf_scatter_grad_rows(const float *packed, uint32_t num_rows, uint32_t num_points, uint32_t attr_dim,
                         int mode, float *points_grad, float *attr_grad, void *stream);

/* The same two with attr_grad rows `attr_pitch` floats apart (>= attr_dim): what rf_trace_backward writes when
 * rf_launch_opts.attr_grad_pitch pads the rows to 64-byte lines.  The packed rows are the same either way. */
int rf_compact_grad_rows_pitched(const float *points_grad, const float *attr_grad, uint32_t num_points,
                                 uint32_t attr_dim, uint32_t attr_pitch, uint32_t capacity, uint32_t *count, float *packed,
                                 void *stream);
int rf_scatter_grad_rows_pitched(const float *packed, uint32_t num_rows, uint32_t num_points, uint32_t attr_dim,
                                 uint32_t attr_pitch, int mode, float *points_grad, float *attr_grad, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RADFOAM_HIP_H */
