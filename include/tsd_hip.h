/*
 * tsd_hip.h -- C ABI of the MI355X (gfx950) implementation of ohm_tsd_slam's per-scan hot path.
 *
 * This is the drop-in boundary: three device entry points (push / raycast / icp, plus the fused
 * localize) that replace the three places where the reference's worker threads enter the vendored
 * "obviously" library, with the TSD grid resident in HBM between calls.  Plain pointers and sizes
 * only; no C++/torch types.  Host pointers unless the name ends in _dev.  Every call returns
 * TSD_OK (0), a negative TSD_E* code (argument / HIP failure; text via tsd_last_error), or -- for the
 * registration calls -- fills `state` with the reference's EnumIcpState.  Nothing throws.
 *
 * Citations `file:line` are relative to the reference tree (autonohm/ohm_tsd_slam, src/).
 * A tsd_ctx serialises its work on one HIP stream; distinct contexts are independent (one grid per
 * GPU for the multi-robot case).  Calls on one ctx must not be issued concurrently from two threads
 * (the C++ facade in ohm_tsd_slam_amd/csrc/host serialises them with a mutex).
 */
#ifndef TSD_HIP_H
#define TSD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSD_OK            0
#define TSD_E_ARG        -1   /* bad argument (null pointer, size out of range) */
#define TSD_E_HIP        -2   /* a HIP runtime call failed; see tsd_last_error */
#define TSD_E_NODEVICE   -3   /* no usable gfx950 device */
#define TSD_E_BOUNDS     -4   /* freeFootprint rectangle outside the grid (TsdGrid.cpp:617-622) */
#define TSD_E_CAPACITY   -5   /* more beams / points than the kernels are built for */

#define TSD_TILE_DIM     32   /* SlamNode.cpp:77 hard-codes LAYOUT_32x32 */
#define TSD_TILE_PITCH   33   /* 32 cells + 1 halo (TsdGridPartition.cpp:97) */
#define TSD_TILE_CELLS   1089
#define TSD_MAX_BEAMS    4096 /* scan staged in LDS by the push kernel */
#define TSD_MAX_ICP_POINTS 2048 /* model/scene resident in LDS in the ICP kernel */
#define TSD_ICP_TRACE_MAX 256  /* iterations recorded by tsd_icp_trace */
#define TSD_ICP_TRACE_STRIDE 8 /* doubles per recorded iteration */

/* EnumIcpState (obvision/registration/icp/Icp.h:25-32) */
#define TSD_ICP_PROCESSING     1
#define TSD_ICP_NOTMATCHABLE   2
#define TSD_ICP_MAXITERATIONS  3
#define TSD_ICP_SUCCESS        5

typedef struct tsd_ctx tsd_ctx;

/* Work counters of one TsdGrid::push (TsdGrid.cpp:217-284); the terms of the algorithmic-bytes
 * formula of DESIGN.md are computed from these. */
typedef struct {
  int64_t cells_updated;        /* addTsd calls that passed sd >= -maxTruncation */
  int64_t cells_visited;        /* back-projected cells (1024 per UPDATE tile) */
  int32_t tiles_total;
  int32_t tiles_range_pass;     /* passed both range tests of isInRange */
  int32_t tiles_update;         /* isInRange() == true */
  int32_t tiles_new;            /* lazily initialised by this push */
  int32_t tiles_new_from_empty; /* ... of which _initWeight > 0 */
  int32_t tiles_emptied_init;   /* increaseEmptiness() on an initialised tile */
  int32_t tiles_emptied_uninit; /* increaseEmptiness() on an uninitialised tile */
} tsd_push_stats;

/* Registration set-up as ThreadLocalize builds it (ThreadLocalize.cpp:211-225, 571-581) */
/* estimator of Icp::step: the one the node constructs (ThreadLocalize.cpp:214) or the reference's other one */
#define TSD_ESTIMATOR_CLOSED_FORM   0  /* ClosedFormEstimator2D (ClosedFormEstimator2D.cpp:36-109), point to point */
#define TSD_ESTIMATOR_POINT_TO_LINE 1  /* PointToLine2DEstimator (PointToLineEstimator2D.cpp:52-157), needs model normals */

typedef struct {
  int    iterations;            /* icp_iterations: Icp max iterations == convergence counter */
  int    estimator;             /* TSD_ESTIMATOR_*; occupies what used to be padding: set it (0 = the node's choice) */
  double dist_filter_max;       /* DistanceFilter(maxdist, mindist, icp_iterations - 10) */
  double dist_filter_min;
  double min_x, max_x, min_y, max_y; /* OutOfBoundsFilter2D = TsdGrid::getMin/MaxX/Y */
  /* Tinit of Icp::iterate(rms, pairs, iterations, Tinit) (Icp.cpp:464-486): the pre-registration result in
   * registration_mode 1-3 (ThreadLocalize.cpp:531-569, :580), 3x3 row-major.  Used by tsd_icp / tsd_icp_normals when
   * use_t_init != 0 (0 = the identity of registration_mode 0); the fused calls always start from the identity. */
  double t_init[9];
  int    use_t_init;
  int    reserved;
} tsd_icp_params;

typedef struct {
  double T[9];                  /* Icp::getFinalTransformation() 3x3 row-major */
  double rms;                   /* estimator's "rms": mean squared pair distance (closed form) or mean |n.(s - m)| (point to line) */
  int32_t pairs;                /* pairs of the last step */
  int32_t iterations;           /* steps executed */
  int32_t state;                /* EnumIcpState */
  int32_t n_model;              /* valid model points (ray-cast hits)  -- tsd_localize only */
  int32_t n_scene;              /* valid scene points                  -- tsd_localize only */
  int32_t reserved;             /* < 0: a TSD_E_* code (TSD_E_CAPACITY: more points than the registration holds).  >= 0 (diagnostic): the
                                 * scene points whose first neighbour search was delivered by the helper workgroups of the launch
                                 * (0 with tsd_debug_set_icp_helpers(ctx, 0)); the results do not depend on it */
} tsd_icp_result;

/* ---- life cycle ------------------------------------------------------------------------------ */
int      tsd_device_count(void);
/* Free and total memory of `device` in bytes as the HIP runtime this library is linked against reports it (hipMemGetInfo): what a
 * caller sizes map_size against -- the node accepts up to 2^15 x 2^15 cells (SlamNode.cpp:71-75), 18.8 GB of fp64 cells here.
 * TSD_E_NODEVICE without such a device. */
int      tsd_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);
/* new TsdGrid(cellSize, LAYOUT_32x32, map_size) + setMaxTruncation(max_trunc)
 * (SlamNode.cpp:77-78, TsdGrid.cpp:112-169, :206-215).  NULL on failure. */
tsd_ctx* tsd_create(int device, int map_size_log2, double cell_size, double max_trunc);
void     tsd_destroy(tsd_ctx* ctx);
int      tsd_reset(tsd_ctx* ctx);                          /* TsdGrid::reset (TsdGrid.cpp:194-198) */
int      tsd_set_max_truncation(tsd_ctx* ctx, double val); /* TsdGrid::setMaxTruncation (:206-215) */
int      tsd_sync(tsd_ctx* ctx);
int      tsd_device(const tsd_ctx* ctx);                   /* HIP device ordinal of the context */
void*    tsd_stream(tsd_ctx* ctx);                         /* its hipStream_t: work a host enqueues there is ordered with the grid's */
const char* tsd_last_error(const tsd_ctx* ctx);

/* sizeof() of a public struct of this header by name ("tsd_push_stats", "tsd_icp_params", ...; 0 if unknown): lets a
 * binding (ctypes, cgo, JNI) check its mirror of the layout at load time */
int      tsd_abi_sizeof(const char* struct_name);

/* ---- geometry getters (TsdGrid.h getCellsX, getCellSize, getMaxTruncation, getMinX.., getMaxX..) -- */
int    tsd_cells(const tsd_ctx* ctx);
int    tsd_tiles(const tsd_ctx* ctx);
double tsd_cell_size(const tsd_ctx* ctx);
double tsd_max_truncation(const tsd_ctx* ctx);
double tsd_min_x(const tsd_ctx* ctx);
double tsd_max_x(const tsd_ctx* ctx);
double tsd_min_y(const tsd_ctx* ctx);
double tsd_max_y(const tsd_ctx* ctx);

/* ---- map update ------------------------------------------------------------------------------ */
/* TsdGrid::freeFootprint (TsdGrid.cpp:609-638) */
int tsd_free_footprint(tsd_ctx* ctx, const double center[2], double width, double height);

/* TsdGrid::push(SensorPolar2D*) (TsdGrid.cpp:217-284): isInRange classification incl. the
 * increaseEmptiness side effect (TsdGridComponent.cpp:43-124), lazy tile init
 * (TsdGridPartition.cpp:88-134), addTsd (TsdGridPartition.h:170-212), propagateBorders
 * (TsdGrid.cpp:372-427).  `ranges`/`mask` are the sensor's PROCESSED data (after setStandardMask:
 * > max_range -> +inf, no NaN).  Asynchronous unless `stats` is non-NULL (then it waits and fills). */
int tsd_push(tsd_ctx* ctx, const double pose33[9], const double* ranges, const uint8_t* mask,
             int beams, double ang_res, double phi_min, double max_range, double min_range,
             double low_refl_range, tsd_push_stats* stats);

/* ---- localisation ---------------------------------------------------------------------------- */
/* RayCastPolar2D::calcCoordsFromCurrentViewMask (RayCastPolar2D.cpp:113-192).  rays_world_2xB is
 * Sensor::getNormalizedRayMap(cellSize): row 0 = x, row 1 = y of every beam's world ray of length
 * cellSize.  Outputs are in the sensor frame; only hit slots are written. */
int tsd_raycast(tsd_ctx* ctx, const double pose33[9], const double* rays_world_2xB, int beams,
                double min_range, double max_range, double* coords_2B, double* normals_2B,
                uint8_t* mask_B, int* n_valid);

/* Icp::reset + setModel + setScene + iterate + getFinalTransformation with the assigner/filter/
 * estimator chain of ThreadLocalize, registration_mode 0 (ThreadLocalize.cpp:571-581;
 * Icp.cpp:410-512; PairAssignment.cpp:38-84; ClosedFormEstimator2D.cpp:36-109).  `pose33` is the
 * sensor pose handed to OutOfBoundsFilter2D::setPose. */
int tsd_icp(tsd_ctx* ctx, const double* model_xy, int n_model, const double* scene_xy, int n_scene,
            const double pose33[9], const tsd_icp_params* params, tsd_icp_result* result);
/* The same with the model normals Icp::setModel(coords, normals) takes (Icp.cpp:150-203); required by
 * TSD_ESTIMATOR_POINT_TO_LINE, ignored by the closed form.  tsd_localize / tsd_scan use the ray cast's normals. */
int tsd_icp_normals(tsd_ctx* ctx, const double* model_xy, const double* model_normals_xy, int n_model,
                    const double* scene_xy, int n_scene, const double pose33[9], const tsd_icp_params* params,
                    tsd_icp_result* result);

/* Fused body of ThreadLocalize::eventLoop between setStandardMask and isRegistrationError
 * (ThreadLocalize.cpp:353-377): ray cast -> dataToCartesianVectorMask -> maskMatrix compaction ->
 * doRegistration, without leaving the device.  rays_local_2xB = Sensor::_raysLocal.  params->t_init (use_t_init != 0) is the
 * Tinit of Icp::iterate (Icp.cpp:481-486), applied while the device stages the scene: registration_mode 3 hands its
 * pre-registration result over this way (csrc/host/ThreadLocalize.cpp: processScanPreRegistered). */
int tsd_localize(tsd_ctx* ctx, const double pose33[9], const double* rays_world_2xB,
                 const double* rays_local_2xB, const double* ranges, const uint8_t* mask, int beams,
                 double min_range, double max_range, const tsd_icp_params* params,
                 tsd_icp_result* result);

/* ---- pre-registration (registration_mode 3) ------------------------------------------------------------- */
/* obvious::TSD_PDFMatching(grid, trials, epsThresh, sizeControlSet, zrand) (TSD_PDFMatching.cpp:6-26; ThreadLocalize.cpp:193)
 * and the arguments of its match() that are not point sets (ThreadLocalize.cpp:559) */
typedef struct {
  int    trials;                /* "trials" (ThreadLocalize.cpp:105), default 100 */
  int    size_control_set;      /* "sizeControlSet" (:106), default 140 */
  double eps_thresh;            /* "epsThresh" (:107); only sets _scaleDistance, which match() never reads */
  double zrand;                 /* "zrand" (:112): clipped probability of a control point without a valid look-up */
  double phi_max;               /* deg2rad("ransac_phi_max"), capped at pi/2 inside (TSD_PDFMatching.cpp:163) */
  double ang_res;               /* sensor->getAngularResolution() */
} tsd_tsdpdf_params;
typedef struct {
  double T[9];                  /* TBest, 3x3 row-major (identity when nothing scored above 0) */
  double probability;           /* bestProb */
  int32_t idx_model, idx_scene; /* the winning pair (beam indices), -1 if none */
  int32_t candidates;           /* (trial, i) pairs scored */
  int32_t valid_model, valid_scene, control_points;   /* idxMValid.size(), idxSValid.size(), Control->getCols() */
  int32_t reserved;
} tsd_tsdpdf_result;
/* obvious::TSD_PDFMatching::match(TSensor, M, maskM, NULL, S, maskS, phiMax, transMax, resolution)
 * (TSD_PDFMatching.cpp:31-294).  model / scene are beam-indexed (beams x 2, row-major) with their masks, exactly what
 * ThreadLocalize builds from _modelCoords / _scene (ThreadLocalize.cpp:367-369).  The reference draws from rand() in
 * three places; here the raw rand() values are inputs: draws_subsample[beams] (RandomMatching::subsampleMask,
 * RandomMatching.cpp:176-189: one per beam, consumed only when 180 / validScenePoints < 0.99), draws_control
 * [size_control_set] (RandomMatching::pickControlSet, :65), draws_trials[trials] (TSD_PDFMatching.cpp:190-194).  Trials
 * are evaluated in trial order (the reference's OpenMP loop leaves the order to the scheduler): the first candidate
 * that reaches the best probability wins.  Scoring runs on the device against the grid in HBM. */
int tsd_tsdpdf_match(tsd_ctx* ctx, const double pose33[9], const double* model_xy_2B, const uint8_t* mask_m,
                     const double* scene_xy_2B, const uint8_t* mask_s, int beams, const tsd_tsdpdf_params* params,
                     const int* draws_subsample, const int* draws_control, const int* draws_trials,
                     tsd_tsdpdf_result* result);

/* ---- fused scan: ThreadLocalize::eventLoop + ThreadMapping push without a host round trip ------------ */
/* Device-resident mirror of one robot's obvious::SensorPolar2D (pose, world / local ray maps) and of
 * ThreadLocalize's pose bookkeeping (_lastPose).  Several sensors may share one grid context
 * (multi-robot mode, SlamNode.cpp:101-122). */
typedef struct tsd_sensor tsd_sensor;

/* gates of ThreadLocalize: isRegistrationError(T, reg_trs_max, reg_sin_rot_max) (ThreadLocalize.cpp:593-600)
 * and isPoseChangeSignificant (:728-736) with TRNS_MIN / ROT_MIN (ThreadLocalize.h:63-64) */
typedef struct {
  double reg_trs_max, reg_sin_rot_max;
  double trs_min, rot_min;
} tsd_gate_params;

typedef struct {
  tsd_icp_result icp;           /* as tsd_localize */
  double pose[9];               /* sensor pose after this scan (unchanged on reg_error / no_model) */
  int32_t reg_error;            /* isRegistrationError: pose kept, caller publishes the NaN pose */
  int32_t pushed;               /* isPoseChangeSignificant: the scan was integrated into the grid */
  int32_t no_model;             /* ray cast found no model points: scan skipped (ThreadLocalize.cpp:354-358) */
  int32_t reserved;
} tsd_scan_result;

/* new SensorPolar2D(beams, ang_res, phi_min, max_range, min_range, low_refl_range) (ThreadLocalize.cpp:498) */
tsd_sensor* tsd_sensor_create(tsd_ctx* ctx, int beams, double ang_res, double phi_min, double max_range,
                              double min_range, double low_refl_range);
void tsd_sensor_destroy(tsd_sensor* s);
/* Upload the sensor state after ThreadLocalize::init (pose, Sensor::getNormalizedRayMap(cellSize), local
 * rays); forgets _lastPose. */
int tsd_sensor_set_pose(tsd_sensor* s, const double pose33[9], const double* rays_world_2xB,
                        const double* rays_local_2xB);
/* One scan: ray cast -> registration -> isRegistrationError -> Sensor::transform ->
 * isPoseChangeSignificant -> TsdGrid::push, all in stream order on the device (the body of
 * ThreadLocalize::eventLoop, ThreadLocalize.cpp:353-406, plus ThreadMapping::eventLoop's push,
 * ThreadMapping.cpp:51-56).  `ranges` / `mask` are the processed scan (setStandardMask), `mask_push` is
 * the mask of the copy ThreadMapping::queuePush makes (ThreadMapping.cpp:65-76; NULL = same mask).
 * Returns as soon as the result record is there (the device writes it to pinned host memory right
 * after the gates); the push of this scan may still be running and is ordered before anything enqueued
 * later on this context.  tsd_sync() waits for it. */
int tsd_scan(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push,
             const tsd_icp_params* params, const tsd_gate_params* gates, tsd_scan_result* result);

/* tsd_scan in two halves, and the NEXT scan staged ahead.  tsd_scan = tsd_scan_submit + tsd_scan_collect.  Between the two a
 * caller that already holds the next scan (a queued LaserScan; a replayed log) hands it to tsd_scan_stage: its copy and the
 * range-query tables of its push run on the side stream while the current registration is busy, and the next
 * tsd_scan_submit(s, NULL, NULL, NULL, ...) starts from the staged data -- the host's per-scan work no longer sits between
 * the ray cast and the registration.  A staged scan that is not the one that comes next is dropped by passing the real one. */
int tsd_scan_submit(tsd_sensor* s, const double* ranges /* NULL: the staged scan */, const uint8_t* mask, const uint8_t* mask_push,
                    const tsd_icp_params* params, const tsd_gate_params* gates);
int tsd_scan_stage(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push /* may be NULL: = mask */);
int tsd_scan_collect(tsd_sensor* s, tsd_scan_result* result);

/* The same scan in two halves for several robots on ONE grid (the reference's multi-robot mode, SlamNode.cpp:101-122),
 * one calling thread per SENSOR.  tsd_scan_begin enqueues copy, tables, ray cast and registration (+ gates,
 * Sensor::transform) on the sensor's own stream and buffers; tsd_scan_wait blocks the calling thread until the result
 * record is there; tsd_scan_finish (which waits itself if need be) enqueues the push on the grid's stream and returns
 * the result.  Registrations of different robots overlap (each occupies one compute unit and does not touch the grid); a
 * ray cast waits for the grid writes enqueued before it and a push for the ray casts enqueued before it, in the order
 * the calls reach their short, internally locked ordered sections -- pushes are serialised, like the reference's single
 * ThreadMapping does, and no ray cast ever sees a half-written tile.  These calls may be issued concurrently for
 * different sensors of one grid; one scan per sensor in flight.  For calls issued in turn the results are those of
 * tsd_scan. */
int tsd_scan_begin(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push,
                   const tsd_icp_params* params, const tsd_gate_params* gates);
int tsd_scan_wait(tsd_sensor* s);
int tsd_scan_finish(tsd_sensor* s, tsd_scan_result* result);

/* Batched form for several robots on ONE grid: the robots that have a scan pending are registered by ONE launch of each kernel
 * (tables: workgroup = scan; ray cast: block row = sensor; registration: workgroup = robot, one compute unit each) on the
 * batch's own stream, their pushes follow one after the other on the grid's stream.  All ray casts of a batch read the same
 * grid state, the pushes are applied in the order of the batch -- one of the interleavings the reference's N ThreadLocalize
 * workers and its single ThreadMapping can produce (SlamNode.cpp:101-122, ThreadMapping.cpp:47-75).  A caller keeps two or
 * three batches and uses them in turn: while one registers, the other's pushes run.
 *   tsd_batch_begin    copy, tables, ray casts, registrations (+ gates, Sensor::transform) of n sensors; returns at once
 *   tsd_batch_push     enqueue the n pushes (gated on the device); may be called BEFORE the registrations have finished: ray
 *                      casts enqueued later wait for these pushes, ray casts enqueued earlier do not
 *   tsd_batch_poll     1 when every result record of the batch has arrived, 0 otherwise (never blocks)
 *   tsd_batch_results  waits for the records, enqueues the pushes unless tsd_batch_push did, fills results[0..n), frees the slot
 * The sensors of a batch must be distinct, attached to the batch's grid and without a scan in flight; one estimator per batch.
 * Hand-offs inside a batch (ray casts -> registrations, a robot's registration -> its push) are waits on the device (a flag set by
 * a one-wave kernel behind the ray casts; a one-wave gate kernel ahead of each push) ONLY where tsd_batch_create's start-up probe has
 * shown that kernels of the two streams involved run side by side (not a given: HIP maps streams onto a few in-order hardware queues,
 * profilers and blocking launches serialise dispatches); otherwise stream events.  TSD_BATCH_EVENT_WAIT=1 forces events.
 * Failure contract (the reference's: "failures are logged and the scan is skipped", ThreadLocalize.cpp:354-358, :381-387): a
 * device-side wait is bounded, and one that runs out does NOT register or push on stale data -- the robots concerned get
 * tsd_scan_result.reserved = 1 (timeout) / 2 (batch abandoned), pose and grid untouched, and tsd_batch_results (or the next
 * tsd_batch_* call on the slot, for a push gate) returns TSD_E_HIP with tsd_last_error() saying so; the slot then uses events, and
 * slot and sensors stay usable. */
#define TSD_BATCH_MAX_SCANS 64
typedef struct tsd_batch tsd_batch;
tsd_batch* tsd_batch_create(tsd_ctx* ctx, int max_scans);
void tsd_batch_destroy(tsd_batch* b);
int tsd_batch_capacity(const tsd_batch* b);
int tsd_batch_inflight(const tsd_batch* b);      /* scans begun and not yet collected */
int tsd_batch_begin(tsd_batch* b, int n, tsd_sensor* const* sensors, const double* const* ranges, const uint8_t* const* mask,
                    const uint8_t* const* mask_push /* NULL or per-scan NULL: = mask */, const tsd_icp_params* params /* [n] */,
                    const tsd_gate_params* gates /* [n] */);
int tsd_batch_push(tsd_batch* b);
int tsd_batch_poll(tsd_batch* b);
int tsd_batch_results(tsd_batch* b, tsd_scan_result* results /* [n] */);

/* Per-iteration record of the most recent tsd_icp / tsd_localize on this ctx (the role of
 * Icp::activateTrace, Icp.cpp:60-70): out[TSD_ICP_TRACE_STRIDE * i + {0..7}] = pairs, rms, DistanceFilter threshold
 * before the step, state after loop control, and the step's Tlast = [[c, -s, tx], [s, c, ty]] as c, s, tx, ty (NaN
 * when the step had too few pairs to estimate), for i < min(iterations, max_iters). */
int tsd_icp_trace(tsd_ctx* ctx, double* out, int max_iters);

/* Parity / debug entry: the pair lists of the first `calls` PairAssignment::determinePairs calls (PairAssignment.cpp:38-84: exact
 * 1-NN, then DistanceFilter.cpp:32-64 with its threshold schedule for params->iterations, then ReciprocalFilter.cpp:32-78) on a
 * STATIC scene -- the scene is not moved between the calls -- as the registration kernel itself forms them (the same code path as
 * tsd_icp: tiers of the exact NN search, LDS atomic-min reciprocal filter; a dedicated instantiation writes the winners out).
 * n_pairs[k] pairs of call k at model_idx / scene_idx[k * n_scene + i], in ascending model index like the reference's output.
 * This is what tests/golden/ref_chain_pairs.npz holds from the COMPILED reference, so the HIP filter chain is pinned to it directly. */
int tsd_icp_pairs(tsd_ctx* ctx, const double* model_xy, int n_model, const double* scene_xy, int n_scene, const double pose33[9],
                  const tsd_icp_params* params, int calls, int* n_pairs /* [calls] */, int* model_idx /* [calls * n_scene] */,
                  int* scene_idx /* [calls * n_scene] */);

/* ---- map I/O --------------------------------------------------------------------------------- */
/* Canonical dump / restore of the tile state: initialized[tiles], init_weight[tiles],
 * tsd[tiles][1089], weight[tiles][1089] (uninitialised tiles read back NaN / 0).  Logical content of
 * TsdGrid::storeGrid / file ctor (TsdGrid.cpp:25-110, 548-607) plus the halo. */
int tsd_download_tiles(tsd_ctx* ctx, uint8_t* initialized, double* init_weight, double* tsd,
                       double* weight);
int tsd_upload_tiles(tsd_ctx* ctx, const uint8_t* initialized, const double* init_weight,
                     const double* tsd, const double* weight);
/* Digest of that canonical dump without moving it: an order-free 64-bit hash over (tile flag, initWeight, the bit
 * patterns of every cell's tsd and weight of the initialised tiles, halo included; NaN and -0.0 canonicalised), the
 * number of non-NaN cells and the sums of tsd / weight over them (per tile, then in tile order).  The hash of the
 * oracle's dump is computed by the same rule (oracle/tsd_oracle.c: ora_grid_digest); tests/golden pins it for
 * BASELINE configs 1-3. */
typedef struct {
  uint64_t hash;
  int64_t  cells_valid;
  int32_t  tiles_initialized;
  int32_t  reserved;
  double   sum_tsd, sum_weight;
} tsd_grid_digest_t;
int tsd_grid_digest(tsd_ctx* ctx, tsd_grid_digest_t* out);
/* bits per stored cell value of this build: 64 (fp64 like the reference, cells bit-identical to the oracle's) or 32
 * (libtsd_hip_q32.so: fixed point, 8 bytes per cell; within n * 2^-27 of the fp64 result after n pushes) */
int tsd_storage_bits(void);
/* tile flags only (cheap) */
int tsd_download_tile_state(tsd_ctx* ctx, uint8_t* initialized, double* init_weight);
/* TsdGrid::storeGrid(path) (TsdGrid.cpp:548-607) and the file constructor TsdGrid(path, FILE_SOURCE) (:25-110): the
 * reference's text format (one value per line, 6 significant digits; per tile its identifier and, for content
 * tiles, tsd and weight of the 32 x 32 interior cells).  Loading replaces the grid's content; the file's layout and
 * cell size must be the context's (TSD_E_ARG otherwise), its maxTruncation is taken over. */
int tsd_store_grid_text(tsd_ctx* ctx, const char* path);
int tsd_load_grid_text(tsd_ctx* ctx, const char* path);

/* Occupancy map of ThreadGrid::eventLoop (ThreadGrid.cpp:72-118) built from
 * RayCastAxisAligned2D::calcCoords (RayCastAxisAligned2D.cpp:13-105): int8 cells*cells,
 * -1 unknown / 0 free / 100 occupied.  The map persists inside the ctx between calls like
 * ThreadGrid::_occGridContent.  The _dev form writes to a device pointer (e.g. a torch tensor that is
 * then max-all-reduced over RCCL). */
int tsd_occupancy(tsd_ctx* ctx, int8_t* occ_host, int inflate, int inflate_factor, int* n_surface);
int tsd_occupancy_dev(tsd_ctx* ctx, void* occ_dev, int inflate, int inflate_factor);
/* the same without the final wait: the extraction kernels are only enqueued on the context's stream (what the RCCL
 * merge of include/tsd_comm.h puts its all-reduce behind) */
int tsd_occupancy_dev_async(tsd_ctx* ctx, void* occ_dev, int inflate, int inflate_factor);
/* TsdGrid::grid2ColorImage(image, width, height) (TsdGrid.cpp:429-488): RGB8, rgb[3 * (h * width + w)], the debug
 * image ThreadGrid publishes with every occupancy map (ThreadGrid.cpp:119-131).  Host buffer of 3*width*height. */
int tsd_color_image(tsd_ctx* ctx, uint8_t* rgb_host, unsigned int width, unsigned int height);

/* ---- measurement ----------------------------------------------------------------------------- */
/* Per-kernel HIP-event timing on the ctx stream.  Kernel names: "push_classify", "push_update",
 * "push_halo", "raycast", "icp", "occupancy", "tsdpdf". */
int tsd_profile_enable(tsd_ctx* ctx, int on);
/* restrict timing to a comma separated list of kernel names, or "all"; a "/n" suffix times every n-th
 * launch only (two event records cost ~13 us of stream time per timed launch); a name may carry its own
 * period, "push_update:1,all/8" = every dispatch of k_push_update, every 8th of the others */
int tsd_profile_select(tsd_ctx* ctx, const char* kernels_csv);
int tsd_profile_reset(tsd_ctx* ctx);
int tsd_profile_get(tsd_ctx* ctx, const char* kernel, double* total_ms, int* launches);
/* spread of the timed dispatches of one kernel since the last reset: shortest, longest, standard deviation (ms) */
int tsd_profile_get_spread(tsd_ctx* ctx, const char* kernel, double* min_ms, double* max_ms, double* std_ms);
/* the timed dispatches themselves, in launch order (ms; at most 65 536 are kept): copies up to `cap` of them, returns how many there are */
int tsd_profile_get_samples(tsd_ctx* ctx, const char* kernel, float* ms_out, int cap);

/* Counter calibration for profiles/: `reps` launches of k_calib_rmw, a read-modify-write of two arrays
 * of n_doubles fp64 values with the push kernel's 8-byte-per-lane access shape (known traffic: 16 B read
 * + 16 B written per element and launch).  Allocates and frees its own scratch. */
int tsd_calibrate_rmw(tsd_ctx* ctx, int64_t n_doubles, int reps);

/* The box's own stream bandwidth with the push kernel's access shape: `reps` event-timed launches of k_calib_rmw over two
 * arrays of n_doubles (each read and written once per launch: 32 B per element).  Choose n_doubles so that the footprint
 * (16 B per element) exceeds the 256 MiB Infinity Cache, or the figure is a cache bandwidth.  GB/s, best and mean launch. */
int tsd_measure_stream(tsd_ctx* ctx, int64_t n_doubles, int reps, double* gbs_best, double* gbs_mean);

/* Sum of the work counters of every push completed on this ctx since the last reset (the numerator of
 * the algorithmic-bytes formula without a host sync per push).  Waits for pushes still in flight. */
int tsd_push_stats_total(tsd_ctx* ctx, tsd_push_stats* total, int64_t* pushes, int reset);

/* registration_mode 3 inside the fused scan (ThreadLocalize.cpp:557-567 with the call structure of tsd_scan): arms the
 * pre-registration -- obvious::TSD_PDFMatching::match, as tsd_tsdpdf_match -- for the NEXT tsd_scan_submit / tsd_scan of this sensor.
 * It then runs on the device between that scan's ray cast (whose hits are its model) and its registration (whose Tinit it becomes,
 * Icp.cpp:481-486): PCA normals of both point sets, extractSamples, pickControlSet, the trial picks and the candidate list in the
 * reference's serial order, scoring, arg-max -- nothing returns to the host in between.  scene_xy_2B / mask_s: what
 * Sensor::dataToCartesianVectorMask gives for the scan (beam-indexed); the draws as in tsd_tsdpdf_match.  One-shot.  The inputs are
 * copied at the call (the caller's buffers are free on return) and travel to the device on the side stream right away.  It may be
 * called while a scan of this sensor is in flight (submitted, not yet collected): it then arms the scan AFTER that one, and the copy is
 * ordered behind the in-flight scan's own pre-registration kernels (refused with TSD_E_ARG when that needs a larger layout than the
 * in-flight scan's: arm it after the collect).
 * Several robots on one grid: a sensor armed this way may be part of a tsd_batch_begin -- its pre-registration then runs inside the
 * batch, on the grid's stream behind the batch's ray casts (all armed robots score against the grid as it is before any push of the
 * batch; the reference runs `case TSD` in every robot's own thread, ThreadLocalize.cpp:557-567), and its result is that robot's Tinit.
 * Armed and unarmed sensors may share a batch.  Not while the sensor has a batched / split scan in flight. */
int tsd_scan_preregister(tsd_sensor* s, const tsd_tsdpdf_params* params, const double* scene_xy_2B, const uint8_t* mask_s,
                         const int* draws_subsample, const int* draws_control, const int* draws_trials);
/* Asynchronous mapping for the fused scan of this sensor.  The reference's ThreadMapping is a thread of its own: queuePush returns at
 * once and the push lands when the mapping thread gets to it (ThreadMapping.cpp:51-76), so the localiser's next ray cast may or may not
 * see it.  on = 0 (default): strict order -- the next scan's ray cast sees this scan's push.  on = 1: the next scan's ray cast is taken
 * right behind this scan's registration, on a grid WITHOUT this scan's push (exactly one push behind, every scan: one of the
 * reference's interleavings, and a deterministic one), and the push runs beside the next registration on a stream of its own.  Every
 * other entry point of the context is ordered behind a push still in flight there.  Not while a scan is in flight. */
int tsd_sensor_set_async_mapping(tsd_sensor* s, int on);
/* TEST HOOK: every asynchronous push of this context is held back by `microseconds` on the push stream (a one-wave kernel ahead of
 * it), 0 = off.  Lets a test make the push stream lag behind the registrations the way a busy device can
 * (tests/test_gpu_async_mapping.py: the scan / table buffers of a lagging push must not be re-staged under it). */
int tsd_debug_stall_push_stream(tsd_ctx* ctx, unsigned int microseconds);
/* TEST HOOK: on = 0 makes every registration of this context search for itself in its first step instead of taking that step's
 * nearest neighbours from the helper workgroups the launch brings by default (icp_kernels.hip: IcpSeed).  The results are the same
 * either way, bit for bit (tests/test_gpu_parity.py::test_icp_helpers_change_nothing); only the time differs. */
int tsd_debug_set_icp_helpers(tsd_ctx* ctx, int on);
/* TEST HOOK: on = 0 makes tsd_batch_push enqueue one push per robot, in the batch's order (the reference's mapper: one sensor after the
 * other, ThreadMapping.cpp:43-76), instead of the one pass per tile that applies the robots' updates in that order inside every tile
 * (csrc/push_multi.hip).  The grids are the same cell for cell, halo included (tests/test_gpu_batch.py). */
int tsd_debug_set_push_multi(tsd_ctx* ctx, int on);
/* TEST HOOK: how a scan of this sensor reaches the device -- 1: the host stores it into device memory through the PCIe BAR (the default
 * wherever the device's memory is mapped into the host's address space and the start-up probe's kernel read back what the host wrote),
 * 0: through a pinned host buffer (no such mapping, or TSD_SCAN_PINNED=1), 2: as 1 with TSD_SCAN_BAR_VERIFY's device-side cross-check. */
int tsd_debug_sensor_scan_path(const tsd_sensor* s);

/* the pre-registration's outcome for the scan collected last (TBest, probability, winning pair, counts) */
int tsd_scan_preregistration_result(tsd_sensor* s, tsd_tsdpdf_result* result);

#ifdef __cplusplus
}
#endif
#endif /* TSD_HIP_H */
