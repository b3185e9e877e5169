/*
 * tsd_oracle.h -- CPU restatement (plain C99 + OpenMP) of the ohm_tsd_slam per-scan hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  The shipped path (ohm_tsd_slam_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * Pinning status (see oracle/README.md and DESIGN.md):
 *   - ICP post-assignment chain (DistanceFilter, ReciprocalFilter, PairAssignment chain logic;
 *     SURVEY row I4) is PINNED against the compiled reference (oracle/_ref, built from the three
 *     GSL/FLANN-free reference translation units where they lie under /root/reference).
 *   - every other row (S1-S3, P1-P6, R1-R2, I1-I3, I5-I6, L1) is "PARITY UNPINNED": the reference
 *     ships no tests / golden vectors and the remaining translation units need GSL + FLANN, which
 *     are absent from this image (and stand-ins are not allowed), so those rows are a line-by-line
 *     restatement of the cited reference source only.
 *
 * All `file:line` citations are relative to /root/reference/src/.
 */
#ifndef TSD_ORACLE_H
#define TSD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_TILE_DIM   32            /* SlamNode.cpp:77 hard-codes LAYOUT_32x32 */
#define ORA_TILE_PITCH 33            /* 32 + 1 halo, TsdGridPartition.cpp:97 */
#define ORA_TILE_CELLS (33 * 33)

typedef struct ora_grid ora_grid;

typedef struct {
  int64_t cells_updated;       /* addTsd calls that passed sd >= -maxTrunc (N_upd) */
  int64_t cells_visited;       /* cells of UPDATE tiles that were back-projected */
  int32_t tiles_total;
  int32_t tiles_range_pass;    /* passed the two range tests of isInRange */
  int32_t tiles_update;        /* isInRange returned true */
  int32_t tiles_new;           /* lazily initialised this push */
  int32_t tiles_new_from_empty;/* ... of which with initWeight > 0 (N_init_from_empty) */
  int32_t tiles_emptied_init;  /* increaseEmptiness on an initialised tile (N_emptied_init) */
  int32_t tiles_emptied_uninit;/* increaseEmptiness on an uninitialised tile */
} ora_push_stats;

typedef struct {
  int    iterations;           /* icp_iterations: maxIterations == convCnt (ThreadLocalize.cpp:223-225) */
  double dist_filter_max;      /* DistanceFilter maxdist */
  double dist_filter_min;      /* DistanceFilter mindist */
  double min_x, max_x, min_y, max_y; /* OutOfBoundsFilter2D bounds = TsdGrid::getMin/Max* */
  int    nn_mode;              /* 0 = brute force (first minimum), 1 = kd-tree (exact) */
} ora_icp_params;

typedef struct {
  double T[9];                 /* Icp::getFinalTransformation, 3x3 row-major */
  double rms;                  /* mean SQUARED pair distance of the last estimating step */
  int    pairs;                /* pairs of the last step */
  int    iterations;           /* steps executed */
  int    state;                /* EnumIcpState (Icp.h:25-32) */
} ora_icp_result;

/* ---- linear algebra helpers (restating the GSL call pattern) ---- */
void ora_mat3_mul(const double A[9], const double B[9], double C[9]);
void ora_mat3_inv(const double A[9], double Ainv[9]);

/* ---- S1: scan ingest ---- */
/* Sensor::setRealMeasurementData(vector<float>, 1.0f) + SensorPolar2D::setStandardMask */
void ora_sensor_ingest_f32(const float* ranges, int n, double max_range, double ang_res,
                           double* data_out, uint8_t* mask_out);
/* Sensor::setRealMeasurementData(double*) + setStandardMask (ThreadMapping::queuePush path) */
void ora_sensor_ingest_f64(const double* ranges, int n, double max_range, double ang_res,
                           double* data_out, uint8_t* mask_out);
/* ThreadLocalize::laserCallBack range clamp (ThreadLocalize.cpp:250-256) */
void ora_laser_min_range_clamp(float* ranges, int n, double laser_min_range);

/* ---- S2: back projection ---- */
int  ora_backproject(const double pose_inv[9], double x, double y, double phi_min, double ang_res,
                     int beams);

/* ---- S3: ray maps ---- */
void ora_rays_local(int beams, double phi_min, double ang_res, double* rays_2xB);
void ora_rays_transform(const double T[9], double* rays_2xB, int beams);
void ora_rays_rescale(double* rays_2xB, int beams, double norm_new, double norm_old);
int  ora_scene_from_scan(const double* rays_local_2xB, const double* data, const uint8_t* mask,
                         int beams, double* scene_2B, uint8_t* mask_s);

/* ---- T0 / P3: grid ---- */
ora_grid* ora_grid_create(int map_size_log2, double cell_size, double max_trunc_request);
void      ora_grid_destroy(ora_grid* g);
int       ora_grid_cells(const ora_grid* g);
int       ora_grid_tiles(const ora_grid* g);
double    ora_grid_max_trunc(const ora_grid* g);
double    ora_grid_max_x(const ora_grid* g);
int       ora_free_footprint(ora_grid* g, const double center[2], double width, double height);
void      ora_grid_tile_state(const ora_grid* g, uint8_t* initialized, double* init_weight);
/* copies 33*33 tsd and weight of one tile; returns 0 if the tile is not initialised */
int       ora_grid_tile_cells(const ora_grid* g, int tile, double* tsd, double* weight);
/* canonical dump of all tiles into caller arrays [tiles][1089] (uninitialised tiles -> NaN / 0) */
void      ora_grid_dump(const ora_grid* g, uint8_t* initialized, double* init_weight,
                        double* tsd, double* weight);
void      ora_grid_load(ora_grid* g, const uint8_t* initialized, const double* init_weight,
                        const double* tsd, const double* weight);
/* getDoubleLine / getIntLine of the grid text format (obcore/base/tools.cpp:190-215) applied to the lines of a file */
int       ora_text_lines(const char* path, const int* kinds, int n, double* out);
/* digest of that dump by the rule of include/tsd_hip.h (tsd_grid_digest) */
void      ora_grid_digest(const ora_grid* g, uint64_t* hash, int64_t* cells_valid, int32_t* tiles_initialized,
                          double* sum_tsd, double* sum_weight);
/* TsdGrid::storeGrid (TsdGrid.cpp:548-607) / TsdGrid(file) (:25-110): the reference's text format */
int       ora_grid_store_text(const ora_grid* g, const char* path);
ora_grid* ora_grid_load_text(const char* path);

/* ---- P1-P6: push ---- */
void ora_push(ora_grid* g, const double pose[9], const double* data, const uint8_t* mask,
              int beams, double ang_res, double phi_min, double max_range, double min_range,
              double low_refl_range, int threads, ora_push_stats* stats);

/* ---- R1-R2: ray cast ---- */
int  ora_interpolate_bilinear(const ora_grid* g, double x, double y, double* tsd);
int  ora_raycast(const ora_grid* g, const double pose[9], const double* rays_world_2xB, int beams,
                 double min_range, double max_range, int threads, double* coords_2B,
                 double* normals_2B, uint8_t* mask_B);

/* ---- I1-I6: registration ---- */
void ora_icp(const double* model_xy, int n_model, const double* scene_xy, int n_scene,
             const double pose[9], const ora_icp_params* p, ora_icp_result* out,
             double* trace_per_iter /* optional [iterations][4] = pairs, rms, thr_before, state */);
/* ... with PointToLine2DEstimator (PointToLineEstimator2D.cpp:52-157) on the model normals */
void ora_icp_point_to_line(const double* model_xy, const double* model_normals_xy, int n_model, const double* scene_xy,
                           int n_scene, const double pose33[9], const ora_icp_params* p, ora_icp_result* out,
                           double* trace);
/* pair chain of ONE ICP step for unit tests: returns number of pairs; thr is updated in place */
int  ora_icp_pairs(const double* model_xy, int n_model, const double* scene_xy, int n_scene,
                   const double pose[9], const ora_icp_params* p, double* thr_sqr,
                   int* pair_model, int* pair_scene);
double ora_distance_filter_multiplier(double maxdist, double mindist, int icp_iterations);

/* ---- L1: gates ---- */
double ora_calc_angle(const double T[9]);
int    ora_is_registration_error(const double T[9], double trs_max, double sin_rot_max);
int    ora_is_pose_change_significant(const double last_pose[9], const double cur_pose[9]);

/* ---- N1: occupancy extraction (RayCastAxisAligned2D::calcCoords + ThreadGrid marking) ---- */
/* content = persistent _occGridContent (caller initialises to -1 once, ThreadGrid.cpp:27-28);
 * out = the published OccupancyGrid data (copy of content + 100 marks).  Returns #sign changes. */
int  ora_occupancy(const ora_grid* g, int8_t* content, int8_t* out, int inflate, int inflate_factor);
/* TsdGrid::grid2ColorImage (TsdGrid.cpp:429-488): rgb[3 * width * height] */
void ora_grid_color_image(const ora_grid* g, unsigned char* image, unsigned int width, unsigned int height);

/* ---- whole-loop driver: ThreadLocalize::init + eventLoop body + ThreadMapping pushes ---- */
typedef struct ora_slam ora_slam;
typedef struct {
  int    map_size_log2; double cell_size; int truncation_radius;
  int    beams; double angle_min; double angle_increment;
  double max_range, min_range, low_refl_range;
  double x_offset, y_offset, local_offset_x, local_offset_y, local_offset_yaw;
  double footprint_width, footprint_height, footprint_x_offset;
  double laser_min_range;
  int    icp_iterations; double dist_filter_max, dist_filter_min;
  double reg_trs_max, reg_sin_rot_max;
  int    nn_mode; int threads;
  /* registration_mode 3 (TSD_PDF pre-registration): ThreadLocalize.cpp:105-128 */
  int    registration_mode; int trials; int size_control_set; double zrand; double ransac_phi_max /* degrees */;
} ora_slam_config;
typedef struct {
  double pose[9]; double T[9];
  double rms; int pairs; int iterations; int icp_state;
  int valid_model; int valid_scene;
  int reg_error; int pushed; int no_model;
  double t_raycast, t_icp, t_push;  /* seconds */
} ora_scan_result;
ora_slam* ora_slam_create(const ora_slam_config* cfg);
/* a further localiser on the grid of `first` (the reference's multi-robot mode, SlamNode.cpp:101-122) */
ora_slam* ora_slam_create_shared(const ora_slam_config* cfg, ora_slam* first);
void      ora_slam_destroy(ora_slam* s);
ora_grid* ora_slam_grid(ora_slam* s);
/* first call = ThreadLocalize::init (freeFootprint + initPush), later calls = eventLoop body with a
 * synchronous push where the reference calls queuePush */
void      ora_slam_process_scan(ora_slam* s, const float* ranges, ora_scan_result* out);
void      ora_slam_last_push_stats(const ora_slam* s, ora_push_stats* out);
/* registration_mode 3: the raw rand() values the next scan's TSD_PDFMatching::match consumes (subsampleMask: beams
 * values; pickControlSet: size_control_set; trial picks: trials), and what the last pre-registration found */
void      ora_slam_set_draws(ora_slam* s, const int* sub, const int* ctrl, const int* trials);
void      ora_slam_last_prereg(const ora_slam* s, double T[9], double* prob, int* idx, int* i);

/* ---- N3: TSD_PDFMatching::match (TSD_PDFMatching.cpp:31-294) with the three rand() streams as inputs.  M / S are
 * beam-indexed (n x 2) with their masks, like ThreadLocalize hands them over.  Returns 0, or 1 when the reference
 * returns the identity early (too few points). */
int ora_tsdpdf_match(const ora_grid* g, const double pose[9], const double* M, const uint8_t* maskM, const double* S,
                     const uint8_t* maskS, int n, int trials, int size_control_set, double zrand, double phi_max,
                     double resolution, const int* draws_subsample, const int* draws_control, const int* draws_trials,
                     double T_out[9], double* best_prob, int* best_idx, int* best_i, int* candidates);
void ora_icp_init(const double* model, int n_model, const double* scene_in, int n_scene, const double pose[9],
                  const ora_icp_params* p, const double Tinit33[9], ora_icp_result* out, double* trace);

#ifdef __cplusplus
}
#endif
#endif
