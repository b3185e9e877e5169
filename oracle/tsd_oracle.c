/*
 * tsd_oracle.c -- CPU restatement of the ohm_tsd_slam per-scan hot path (see tsd_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or called from the product path.
 * Pinning: row I4 pinned against oracle/_ref (compiled reference); all other rows PARITY UNPINNED
 * (reference has no tests; the remaining reference TUs need GSL/FLANN which this image lacks).
 *
 * Every function cites the reference file:line (relative to /root/reference/src/) it restates.
 * Arithmetic is fp64 with the reference's operation order; build with -ffp-contract=off.
 */
#include "tsd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define D   ORA_TILE_DIM
#define PT  ORA_TILE_PITCH
#define TC  ORA_TILE_CELLS
#define TSDGRIDMAXWEIGHT 32.0   /* obvision/reconstruct/reconstruct_defs.h:4 */
#define TSDINC 1.0              /* reconstruct_defs.h:6 */

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------- */
/* Linear algebra: the GSL call pattern of obcore/math/linalg/gsl/Matrix.cpp.                    */
/* GSL/gslcblas (libgsl-dev, unpinned in docker/Dockerfile:21-22; Ubuntu 22.04 = 2.7.1) is a    */
/* third-party dependency absent from /root/reference.  Published algorithm restated:           */
/*  - cblas_dgemm reference kernel: C[i][j] = sum_k A[i][k]*B[k][j], k ascending from 0.0       */
/*  - gsl_linalg_LU_decomp/LU_invert: Gaussian elimination with partial pivoting, inverse by     */
/*    solving against the identity columns.                                                      */
/* ------------------------------------------------------------------------------------------- */

/* Matrix::operator* / operator*= -> gsl_blas_dgemm(NoTrans,NoTrans) (gsl/Matrix.cpp:45-52,88-95) */
void ora_mat3_mul(const double A[9], const double B[9], double C[9])
{
  double R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double t = 0.0;
      for (int k = 0; k < 3; k++) t += A[3 * i + k] * B[3 * k + j];
      R[3 * i + j] = t;
    }
  memcpy(C, R, sizeof(R));
}

/* Matrix::invert -> gsl_linalg_LU_decomp + gsl_linalg_LU_invert (gsl/Matrix.cpp:168-179) */
void ora_mat3_inv(const double A[9], double Ainv[9])
{
  double lu[9];
  int perm[3] = {0, 1, 2};
  memcpy(lu, A, sizeof(lu));
  for (int j = 0; j < 3; j++) {
    int piv = j;
    double best = fabs(lu[3 * j + j]);
    for (int i = j + 1; i < 3; i++)
      if (fabs(lu[3 * i + j]) > best) { best = fabs(lu[3 * i + j]); piv = i; }
    if (piv != j) {
      for (int k = 0; k < 3; k++) { double t = lu[3 * j + k]; lu[3 * j + k] = lu[3 * piv + k]; lu[3 * piv + k] = t; }
      int t = perm[j]; perm[j] = perm[piv]; perm[piv] = t;
    }
    for (int i = j + 1; i < 3; i++) {
      lu[3 * i + j] = lu[3 * i + j] / lu[3 * j + j];
      for (int k = j + 1; k < 3; k++) lu[3 * i + k] -= lu[3 * i + j] * lu[3 * j + k];
    }
  }
  for (int c = 0; c < 3; c++) {
    double x[3];
    for (int i = 0; i < 3; i++) x[i] = (perm[i] == c) ? 1.0 : 0.0;
    for (int i = 1; i < 3; i++)           /* forward: L y = P e_c */
      for (int k = 0; k < i; k++) x[i] -= lu[3 * i + k] * x[k];
    for (int i = 2; i >= 0; i--) {        /* backward: U x = y */
      for (int k = i + 1; k < 3; k++) x[i] -= lu[3 * i + k] * x[k];
      x[i] = x[i] / lu[3 * i + i];
    }
    for (int i = 0; i < 3; i++) Ainv[3 * i + c] = x[i];
  }
}

/* ------------------------------------------------------------------------------------------- */
/* S1: scan ingest                                                                               */
/* ------------------------------------------------------------------------------------------- */

/* SensorPolar2D::setStandardMask (SensorPolar2D.cpp:59-65) = resetMask (Sensor.cpp:246-250),
 * maskZeroDepth (:252-256), maskInvalidDepth (:258-272), maskDepthDiscontinuity(3 deg)
 * (SensorPolar2D.cpp:67-98) */
static void standard_mask(double* data, uint8_t* mask, int n, double max_range, double ang_res)
{
  for (int i = 0; i < n; i++) mask[i] = 1;
  for (int i = 0; i < n; i++) mask[i] = mask[i] && (data[i] != 0.0);
  for (int i = 0; i < n; i++) {
    if (data[i] > max_range) data[i] = INFINITY;
    if (isnan(data[i])) { mask[i] = 0; data[i] = INFINITY; }
  }
  const double thresh = 3.0 * M_PI / 180.0;   /* deg2rad(3.0), mathbase.h:175-178 */
  const int radius = 1;
  const double cosphi = cos(ang_res), sinphi = sin(ang_res);   /* sincos() */
  for (int i = radius; i < n - radius; i++) {
    double betamin = M_PI;
    const double a = data[i];
    if (isinf(a)) continue;
    for (int j = -radius; j <= radius; j++) {
      const double b = data[i + j];
      if (isinf(b)) continue;
      const double c = sqrt(a * a + b * b - 2 * a * b * cosphi);
      if (a > b) {
        const double beta = asin(b / c * sinphi);
        if (beta < betamin) betamin = beta;
      }
    }
    if (betamin < thresh) mask[i] = 0;
  }
}

/* Sensor::setRealMeasurementData(vector<float>, float scale=1) (Sensor.cpp:136-145): the multiply
 * happens in float, then widens */
void ora_sensor_ingest_f32(const float* ranges, int n, double max_range, double ang_res,
                           double* data_out, uint8_t* mask_out)
{
  const float scale = 1.0f;
  for (int i = 0; i < n; i++) data_out[i] = (double)(ranges[i] * scale);
  standard_mask(data_out, mask_out, n, max_range, ang_res);
}

/* Sensor::setRealMeasurementData(double*, scale=1) = memcpy (Sensor.cpp:125-134), then
 * setStandardMask: what ThreadMapping::queuePush does to its deep copy (ThreadMapping.cpp:65-76) */
void ora_sensor_ingest_f64(const double* ranges, int n, double max_range, double ang_res,
                           double* data_out, uint8_t* mask_out)
{
  if (data_out != ranges) memcpy(data_out, ranges, (size_t)n * sizeof(double));
  standard_mask(data_out, mask_out, n, max_range, ang_res);
}

/* ThreadLocalize::laserCallBack (ThreadLocalize.cpp:250-256): float < double comparison */
void ora_laser_min_range_clamp(float* ranges, int n, double laser_min_range)
{
  for (int i = 0; i < n; i++)
    if (ranges[i] < laser_min_range) ranges[i] = 0.0f;
}

/* ------------------------------------------------------------------------------------------- */
/* S2: back projection, batched form (SensorPolar2D.cpp:117-135)                                */
/* coords2D = PoseInv * M^T via dgemm(NoTrans,Trans): ((0 + a*x) + b*y) + c*1                   */
/* ------------------------------------------------------------------------------------------- */
static inline int backproject(const double Pi[9], double x, double y, double phi_min,
                              double ang_res_inv, double phi_lower, double phi_upper)
{
  double lx = 0.0, ly = 0.0;
  lx += Pi[0] * x; lx += Pi[1] * y; lx += Pi[2] * 1.0;
  ly += Pi[3] * x; ly += Pi[4] * y; ly += Pi[5] * 1.0;
  const double phi = atan2(ly, lx);
  if (phi <= phi_lower) return -2;
  if (phi >= phi_upper) return -1;
  return (int)round((phi - phi_min) * ang_res_inv);
}

/* sensor bounds: SensorPolar2D ctor (SensorPolar2D.cpp:26-30) */
static inline void sensor_bounds(double phi_min, double ang_res, int beams, double* lo, double* up)
{
  *lo = -0.5 * ang_res + phi_min;
  *up = phi_min + (((double)beams) - 0.5) * ang_res;
}

int ora_backproject(const double pose_inv[9], double x, double y, double phi_min, double ang_res,
                    int beams)
{
  double lo, up;
  sensor_bounds(phi_min, ang_res, beams, &lo, &up);
  return backproject(pose_inv, x, y, phi_min, 1.0 / ang_res, lo, up);
}

/* ------------------------------------------------------------------------------------------- */
/* S3: ray maps                                                                                  */
/* ------------------------------------------------------------------------------------------- */

/* SensorPolar2D ctor (SensorPolar2D.cpp:37-47): rays(0,i)=cos(phi), rays(1,i)=sin(phi) */
void ora_rays_local(int beams, double phi_min, double ang_res, double* rays)
{
  for (int i = 0; i < beams; i++) {
    const double phi = phi_min + ((double)i) * ang_res;
    rays[i] = cos(phi);
    rays[beams + i] = sin(phi);
  }
}

/* Sensor::transform (Sensor.cpp:50-55): rays = R * rays with R = T[0:2,0:2] (dgemm NoTrans,NoTrans) */
void ora_rays_transform(const double T[9], double* rays, int beams)
{
  for (int i = 0; i < beams; i++) {
    const double x = rays[i], y = rays[beams + i];
    double nx = 0.0, ny = 0.0;
    nx += T[0] * x; nx += T[1] * y;
    ny += T[3] * x; ny += T[4] * y;
    rays[i] = nx; rays[beams + i] = ny;
  }
}

/* Sensor::getNormalizedRayMap (Sensor.cpp:36-48): in-place rescale by (norm/_rayNorm) */
void ora_rays_rescale(double* rays, int beams, double norm_new, double norm_old)
{
  if (norm_new != norm_old)
    for (int i = 0; i < beams; i++) {
      rays[i] *= (norm_new / norm_old);
      rays[beams + i] *= (norm_new / norm_old);
    }
}

/* Sensor::dataToCartesianVectorMask (Sensor.cpp:168-190) */
int ora_scene_from_scan(const double* rays_local, const double* data, const uint8_t* mask,
                        int beams, double* scene, uint8_t* mask_s)
{
  int valid = 0;
  for (int i = 0; i < beams; i++) {
    if (!isinf(data[i]) && mask[i]) {
      scene[2 * i] = rays_local[i] * data[i];
      scene[2 * i + 1] = rays_local[beams + i] * data[i];
      valid++;
      mask_s[i] = 1;
    } else {
      mask_s[i] = 0;
    }
  }
  return valid;
}

/* ------------------------------------------------------------------------------------------- */
/* T0 / P3: grid                                                                                 */
/* ------------------------------------------------------------------------------------------- */
struct ora_grid {
  int map_log2, N, PX, tiles;
  double cs, inv_cs, max_trunc;
  double min_x, max_x, min_y, max_y;
  uint8_t* init;        /* TsdGridPartition::_initialized */
  double* init_weight;  /* TsdGridPartition::_initWeight */
  double** tsd;         /* lazily allocated 33x33 (TsdGridPartition.cpp:97) */
  double** weight;
};

/* TsdGrid::init (TsdGrid.cpp:112-169) + setMaxTruncation (:206-215) */
ora_grid* ora_grid_create(int map_size_log2, double cell_size, double max_trunc_request)
{
  ora_grid* g = (ora_grid*)calloc(1, sizeof(ora_grid));
  g->map_log2 = map_size_log2;
  g->N = 1 << map_size_log2;
  g->PX = g->N / D;
  g->tiles = g->PX * g->PX;
  g->cs = cell_size;
  g->inv_cs = 1.0 / cell_size;
  g->max_trunc = 2.0 * cell_size;
  g->min_x = 0.0; g->max_x = ((double)g->N + 0.5) * cell_size;
  g->min_y = 0.0; g->max_y = ((double)g->N + 0.5) * cell_size;
  double val = max_trunc_request;
  if (val < 2 * cell_size) val = 2 * cell_size;
  g->max_trunc = val;
  g->init = (uint8_t*)calloc((size_t)g->tiles, 1);
  g->init_weight = (double*)calloc((size_t)g->tiles, sizeof(double));
  g->tsd = (double**)calloc((size_t)g->tiles, sizeof(double*));
  g->weight = (double**)calloc((size_t)g->tiles, sizeof(double*));
  return g;
}

void ora_grid_destroy(ora_grid* g)
{
  if (!g) return;
  for (int i = 0; i < g->tiles; i++) { free(g->tsd[i]); free(g->weight[i]); }
  free(g->tsd); free(g->weight); free(g->init); free(g->init_weight); free(g);
}

int ora_grid_cells(const ora_grid* g) { return g->N; }
int ora_grid_tiles(const ora_grid* g) { return g->tiles; }
double ora_grid_max_trunc(const ora_grid* g) { return g->max_trunc; }
double ora_grid_max_x(const ora_grid* g) { return g->max_x; }

/* TsdGridPartition::init (TsdGridPartition.cpp:88-134) */
static void tile_init(ora_grid* g, int p)
{
  if (g->init[p]) return;
  g->tsd[p] = (double*)malloc(TC * sizeof(double));
  g->weight[p] = (double*)malloc(TC * sizeof(double));
  const double iw = g->init_weight[p];
  if (iw > 0.0) {
    for (int i = 0; i < TC; i++) { g->tsd[p][i] = 1.0; g->weight[p][i] = iw; }
  } else {
    for (int i = 0; i < TC; i++) { g->tsd[p][i] = NAN; g->weight[p][i] = iw; }
  }
  g->init[p] = 1;
}

/* TsdGrid::freeFootprint (TsdGrid.cpp:609-638) */
int ora_free_footprint(ora_grid* g, const double c[2], double width, double height)
{
  unsigned int minX = (unsigned int)((c[0] - width * 0.5) / g->cs + 0.5);
  unsigned int maxX = (unsigned int)((c[0] + width * 0.5) / g->cs + 0.5);
  unsigned int minY = (unsigned int)((c[1] - height * 0.5) / g->cs + 0.5);
  unsigned int maxY = (unsigned int)((c[1] + height * 0.5) / g->cs + 0.5);
  if (minX > (unsigned)g->N || maxX > (unsigned)g->N || minY > (unsigned)g->N || maxY > (unsigned)g->N)
    return 0;
  for (unsigned int rows = minY; rows < maxY; rows++)
    for (unsigned int cols = minX; cols < maxX; cols++) {
      unsigned int py = rows / D, px = cols / D;
      int p = (int)(py * (unsigned)g->PX + px);
      if (!g->init[p]) tile_init(g, p);
      g->tsd[p][(rows % D) * PT + (cols % D)] = TSDINC;
    }
  return 1;
}

void ora_grid_tile_state(const ora_grid* g, uint8_t* initialized, double* init_weight)
{
  memcpy(initialized, g->init, (size_t)g->tiles);
  memcpy(init_weight, g->init_weight, (size_t)g->tiles * sizeof(double));
}

int ora_grid_tile_cells(const ora_grid* g, int tile, double* tsd, double* weight)
{
  if (!g->init[tile]) return 0;
  memcpy(tsd, g->tsd[tile], TC * sizeof(double));
  memcpy(weight, g->weight[tile], TC * sizeof(double));
  return 1;
}

void ora_grid_dump(const ora_grid* g, uint8_t* initialized, double* init_weight, double* tsd,
                   double* weight)
{
  ora_grid_tile_state(g, initialized, init_weight);
  for (int p = 0; p < g->tiles; p++) {
    if (g->init[p]) {
      memcpy(tsd + (size_t)p * TC, g->tsd[p], TC * sizeof(double));
      memcpy(weight + (size_t)p * TC, g->weight[p], TC * sizeof(double));
    } else {
      for (int i = 0; i < TC; i++) { tsd[(size_t)p * TC + i] = NAN; weight[(size_t)p * TC + i] = 0.0; }
    }
  }
}

/* Digest of the canonical dump (the rule of include/tsd_hip.h: tsd_grid_digest): order-free 64-bit hash of
 * (tile flag, initWeight, bit patterns of tsd / weight of every cell of the initialised tiles; NaN and -0.0
 * canonicalised), count of non-NaN cells, sums over them per tile then in tile order.  Test infrastructure: lets the
 * cfg 1-3 fixtures pin whole grids without shipping the dumps. */
static uint64_t digest_mix(uint64_t k, uint64_t a, uint64_t b)
{
  uint64_t x = k * 0x9E3779B97F4A7C15ull + a;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  x += b;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return x;
}
static uint64_t digest_bits(double v)
{
  if (isnan(v)) return 0x7FF8000000000000ull;
  v = v + 0.0;
  uint64_t u; memcpy(&u, &v, 8);
  return u;
}
void ora_grid_digest(const ora_grid* g, uint64_t* hash, int64_t* cells_valid, int32_t* tiles_initialized,
                     double* sum_tsd, double* sum_weight)
{
  uint64_t h = 0; int64_t nv = 0; int32_t ni = 0; double st = 0.0, sw = 0.0;
  for (int p = 0; p < g->tiles; p++) {
    h += digest_mix((uint64_t)p * 2048ull, g->init[p] ? 1ull : 0ull, digest_bits(g->init_weight[p]));
    if (!g->init[p]) continue;
    ni++;
    double tt = 0.0, tw = 0.0;
    for (int i = 0; i < TC; i++) {
      const double t = g->tsd[p][i], w = g->weight[p][i];
      h += digest_mix((uint64_t)p * 2048ull + 1ull + (uint64_t)i, digest_bits(t), digest_bits(w));
      if (!isnan(t)) { nv++; tt += t; tw += w; }
    }
    st += tt; sw += tw;
  }
  *hash = h; *cells_valid = nv; *tiles_initialized = ni; *sum_tsd = st; *sum_weight = sw;
}

void ora_grid_load(ora_grid* g, const uint8_t* initialized, const double* init_weight,
                   const double* tsd, const double* weight)
{
  for (int p = 0; p < g->tiles; p++) {
    g->init_weight[p] = init_weight[p];
    if (initialized[p]) {
      if (!g->init[p]) tile_init(g, p);
      memcpy(g->tsd[p], tsd + (size_t)p * TC, TC * sizeof(double));
      memcpy(g->weight[p], weight + (size_t)p * TC, TC * sizeof(double));
    } else if (g->init[p]) {
      free(g->tsd[p]); free(g->weight[p]); g->tsd[p] = NULL; g->weight[p] = NULL; g->init[p] = 0;
    }
  }
}

/* TsdGrid::storeGrid (TsdGrid.cpp:548-607): text, one value per line in the default ostream format (6 significant
 * digits == "%g"): cellSize, partition layout, grid layout, maxTruncation, then per tile (y outer, x inner) its
 * identifier 0 uninitialised / 1 empty (+ initWeight) / 2 content (+ tsd, weight of the 32 x 32 interior cells,
 * row-major; the halo is not stored). */
int ora_grid_store_text(const ora_grid* g, const char* path)
{
  if (!path || !path[0]) return 0;
  FILE* f = fopen(path, "w");
  if (!f) return 0;
  fprintf(f, "%g\n%d\n%d\n%g\n", g->cs, 5 /* LAYOUT_32x32 */, g->map_log2, g->max_trunc);
  for (int p = 0; p < g->tiles; p++) {
    if (g->init[p]) {
      fprintf(f, "2\n");
      for (int py = 0; py < D; py++)
        for (int px = 0; px < D; px++) fprintf(f, "%g\n%g\n", g->tsd[p][py * PT + px], g->weight[p][py * PT + px]);
    } else if (g->init_weight[p] > 0.0) {        /* isEmpty() */
      fprintf(f, "1\n%g\n", g->init_weight[p]);
    } else {
      fprintf(f, "0\n");
    }
  }
  fclose(f);
  return 1;
}

/* getDoubleLine / getIntLine (obcore/base/tools.cpp:190-215) */
static double text_double_line(FILE* f)
{
  char line[1024];
  if (!fgets(line, sizeof(line), f) || line[0] == '\n' || line[0] == 0) return NAN;
  return strtod(line, NULL);
}
static int text_int_line(FILE* f)
{
  char line[1024];
  if (!fgets(line, sizeof(line), f) || line[0] == '\n' || line[0] == 0) return 0;
  return atoi(line);
}

/* the two line readers on a file, line i as a double (kinds[i] == 0) or an int (1): pinned against the compiled
 * obcore/base/tools.cpp in oracle/_ref (tests/test_cpu_oracle_ref.py) */
int ora_text_lines(const char* path, const int* kinds, int n, double* out)
{
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  for (int i = 0; i < n; i++) out[i] = kinds[i] ? (double)text_int_line(f) : text_double_line(f);
  fclose(f);
  return 1;
}

/* TsdGrid(const std::string&, FILE_SOURCE) (TsdGrid.cpp:25-110): a new grid from such a file.  A content tile is
 * init()-ed (halo included) and its interior overwritten; the halo keeps the init value until the next push. */
ora_grid* ora_grid_load_text(const char* path)
{
  FILE* f = fopen(path, "r");
  if (!f) return NULL;
  const double cell_size = text_double_line(f);
  const int layout_part = text_int_line(f);
  const int layout_grid = text_int_line(f);
  if (layout_grid < 0 || layout_part != 5 || layout_grid > 15) { fclose(f); return NULL; }
  const double max_trunc = text_double_line(f);
  ora_grid* g = ora_grid_create(layout_grid, cell_size, max_trunc);
  for (int p = 0; p < g->tiles; p++) {
    const int id = text_int_line(f);
    if (id == 0) continue;
    if (id == 1) {
      g->init_weight[p] = fmin(text_double_line(f), TSDGRIDMAXWEIGHT);
    } else if (id == 2) {
      tile_init(g, p);
      for (int py = 0; py < D; py++)
        for (int px = 0; px < D; px++) {
          g->tsd[p][py * PT + px] = text_double_line(f);
          g->weight[p][py * PT + px] = text_double_line(f);
        }
    } else { fclose(f); ora_grid_destroy(g); return NULL; }
  }
  fclose(f);
  return g;
}

/* ------------------------------------------------------------------------------------------- */
/* P1-P6: push                                                                                   */
/* ------------------------------------------------------------------------------------------- */

/* TsdGridPartition::increaseEmptiness (TsdGridPartition.cpp:136-164) */
static void increase_emptiness(ora_grid* g, int p)
{
  if (g->init[p]) {
    double* t = g->tsd[p];
    double* w = g->weight[p];
    for (int i = 0; i < TC; i++) {         /* y<=cellsY, x<=cellsX: halo included */
      if (isnan(t[i])) {
        w[i] += 1.0;
        t[i] = 1.0;
      } else {
        w[i] = fmin(w[i] + 1, TSDGRIDMAXWEIGHT);
        t[i] = (t[i] * (w[i] - 1.0) + 1.0) / w[i];
      }
    }
  } else {
    g->init_weight[p] += 1.0;
    g->init_weight[p] = fmin(g->init_weight[p], TSDGRIDMAXWEIGHT);
  }
}

/* TsdGridPartition::addTsd (TsdGridPartition.h:170-212); returns 1 if the cell was updated */
static inline int add_tsd(double* tsd, double* weight, double sd, double part_weight,
                          double max_trunc, double inv_max_trunc, double eps)
{
  if (sd >= -max_trunc) {
    double v = fmin(sd * inv_max_trunc, TSDINC);
    double w = 0.01;
    if (fabs(sd) < eps) w = 1.0;            /* dead: eps = -cellSize/2 < 0 (TsdGridPartition.cpp:95) */
    w *= part_weight;
    if (isnan(*tsd)) {
      *tsd = v;
      *weight += w;
    } else {
      *tsd = (*tsd * *weight + v * w) / (*weight + w);
      *weight = fmin(*weight + w, TSDGRIDMAXWEIGHT);
    }
    return 1;
  }
  return 0;
}

typedef struct {
  const double* data; const uint8_t* mask; int beams;
  double phi_min, ang_res_inv, phi_lower, phi_upper;
  double max_range, min_range, low_refl;
  double Pi[9]; double tr[2];
} scan_ctx;

/* tile geometry: TsdGridPartition ctor (TsdGridPartition.cpp:48-70) */
static inline void tile_geometry(const ora_grid* g, int p, double e[4][2], double c[2], double* rad)
{
  const unsigned int x = (unsigned)(p % g->PX) * D, y = (unsigned)(p / g->PX) * D;
  e[0][0] = ((double)x + 0.5) * g->cs;       e[0][1] = ((double)y + 0.5) * g->cs;
  e[1][0] = ((double)(x + D) + 0.5) * g->cs; e[1][1] = ((double)y + 0.5) * g->cs;
  e[2][0] = ((double)x + 0.5) * g->cs;       e[2][1] = ((double)(y + D) + 0.5) * g->cs;
  e[3][0] = ((double)(x + D) + 0.5) * g->cs; e[3][1] = ((double)(y + D) + 0.5) * g->cs;
  c[0] = (e[0][0] + e[1][0] + e[2][0] + e[3][0]) / 4.0;
  c[1] = (e[0][1] + e[1][1] + e[2][1] + e[3][1]) / 4.0;
  const double dx = e[3][0] - e[0][0], dy = e[3][1] - e[0][1];
  *rad = sqrt(dx * dx + dy * dy) * 0.5;
}

/* TsdGridComponent::isInRange (TsdGridComponent.cpp:43-124), leaf branch.
 * returns 1 = UPDATE, 0 = skip; *range_pass / *emptied report the path taken */
static int is_in_range(ora_grid* g, int p, const scan_ctx* s, int* range_pass, int* emptied)
{
  double e[4][2], c[2], rad;
  tile_geometry(g, p, e, c, &rad);
  /* euklideanDistance<obfloat>(pos, _centroid, 2) (mathbase.h:369-378) */
  double sqr = 0.0;
  { double t0 = s->tr[0] - c[0]; sqr += t0 * t0; double t1 = s->tr[1] - c[1]; sqr += t1 * t1; }
  const double distance = sqrt(sqr);
  const double closest = distance - rad - g->max_trunc;
  if (closest > s->max_range) return 0;
  const double farthest = distance + rad + g->max_trunc;
  if (farthest < s->min_range) return 0;
  *range_pass = 1;

  int idx[4];
  for (int k = 0; k < 4; k++)
    idx[k] = backproject(s->Pi, e[k][0], e[k][1], s->phi_min, s->ang_res_inv, s->phi_lower, s->phi_upper);
  int any_vis = 0, all_vis = 1;
  for (int k = 0; k < 4; k++) {
    if (idx[k] == -1) { idx[k] = s->beams - 1; all_vis = 0; }
    else if (idx[k] == -2) { idx[k] = 0; all_vis = 0; }
    else any_vis = 1;
  }
  if (!any_vis) return 0;
  /* minmaxArray<int> (mathbase.h:55-64) */
  int lo = idx[0], hi = idx[0];
  for (int k = 1; k < 4; k++) { if (lo > idx[k]) lo = idx[k]; else if (hi < idx[k]) hi = idx[k]; }

  int visible = 0;
  for (int j = lo; j <= hi; j++) visible = visible || ((s->data[j] > closest) && s->mask[j]);
  if (!visible) return 0;

  if (all_vis) {
    int empty = 1;
    for (int j = lo; j <= hi; j++) {
      if (isinf(s->data[j])) empty = empty && (distance < s->low_refl);
      else empty = empty && (s->data[j] > farthest) && s->mask[j];
    }
    if (empty) { increase_emptiness(g, p); *emptied = 1; return 0; }
  }
  return 1;
}

/* TsdGrid::propagateBorders (TsdGrid.cpp:372-427) */
static void propagate_borders(ora_grid* g)
{
  const int PX = g->PX;
  for (int py = 0; py < PX; py++)
    for (int px = 0; px < PX; px++) {
      const int p = py * PX + px;
      if (!g->init[p]) continue;
      if (px < PX - 1 && g->init[p + 1])
        for (int i = 0; i < D; i++) {
          g->tsd[p][i * PT + D] = g->tsd[p + 1][i * PT];
          g->weight[p][i * PT + D] = g->weight[p + 1][i * PT];
        }
      if (py < PX - 1 && g->init[p + PX])
        for (int i = 0; i < D; i++) {
          g->tsd[p][D * PT + i] = g->tsd[p + PX][i];
          g->weight[p][D * PT + i] = g->weight[p + PX][i];
        }
      if (px < PX - 1 && py < PX - 1 && g->init[p + PX + 1]) {
        g->tsd[p][D * PT + D] = g->tsd[p + PX + 1][0];
        g->weight[p][D * PT + D] = g->weight[p + PX + 1][0];
      }
    }
}

/* TsdGrid::push (TsdGrid.cpp:217-284) */
void ora_push(ora_grid* g, const double pose[9], const double* data, const uint8_t* mask,
              int beams, double ang_res, double phi_min, double max_range, double min_range,
              double low_refl_range, int threads, ora_push_stats* stats)
{
  scan_ctx s;
  s.data = data; s.mask = mask; s.beams = beams;
  s.phi_min = phi_min; s.ang_res_inv = 1.0 / ang_res;
  sensor_bounds(phi_min, ang_res, beams, &s.phi_lower, &s.phi_upper);
  s.max_range = max_range; s.min_range = min_range; s.low_refl = low_refl_range;
  ora_mat3_inv(pose, s.Pi);
  s.tr[0] = pose[2]; s.tr[1] = pose[5];   /* Sensor::getPosition (Sensor.cpp:114-118) */

  const double max_trunc = g->max_trunc;
  const double inv_max_trunc = 1.0 / max_trunc;
  const double eps = -g->cs / 2.0;
  long long n_upd = 0, n_vis = 0;
  int n_range = 0, n_update = 0, n_new = 0, n_new_e = 0, n_emp_i = 0, n_emp_u = 0;
  (void)threads;
#ifdef _OPENMP
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(threads) \
    reduction(+ : n_upd, n_vis, n_range, n_update, n_new, n_new_e, n_emp_i, n_emp_u)
#endif
  for (int p = 0; p < g->tiles; p++) {
    int range_pass = 0, emptied = 0;
    const int was_init = g->init[p];
    const int in = is_in_range(g, p, &s, &range_pass, &emptied);
    n_range += range_pass;
    if (emptied) { if (was_init) n_emp_i++; else n_emp_u++; }
    if (!in) continue;
    n_update++;
    if (!was_init) { n_new++; if (g->init_weight[p] > 0.0) n_new_e++; }
    tile_init(g, p);

    double e[4][2], c[2], rad;
    tile_geometry(g, p, e, c, &rad);
    double dist_c = sqrt((c[0] - s.tr[0]) * (c[0] - s.tr[0]) + (c[1] - s.tr[1]) * (c[1] - s.tr[1]));
    if (dist_c > max_range) dist_c = max_range;
    double pw = (max_range - dist_c) / max_range;
    pw *= pw;

    const unsigned int x0 = (unsigned)(p % g->PX) * D, y0 = (unsigned)(p / g->PX) * D;
    double* T = g->tsd[p];
    double* W = g->weight[p];
    for (unsigned int iy = 0; iy < D; iy++)
      for (unsigned int ix = 0; ix < D; ix++) {
        const double cx = ((double)(x0 + ix) + 0.5) * g->cs;   /* TsdGridPartition.cpp:127-128 */
        const double cy = ((double)(y0 + iy) + 0.5) * g->cs;
        const int index = backproject(s.Pi, cx, cy, s.phi_min, s.ang_res_inv, s.phi_lower, s.phi_upper);
        n_vis++;
        if (index >= 0 && mask[index]) {
          const int ci = (int)(iy * PT + ix);
          if (!isinf(data[index])) {
            const double sd = data[index] - sqrt((cx - s.tr[0]) * (cx - s.tr[0]) + (cy - s.tr[1]) * (cy - s.tr[1]));
            n_upd += add_tsd(&T[ci], &W[ci], sd, pw, max_trunc, inv_max_trunc, eps);
          } else {
            const double dist = sqrt((cx - s.tr[0]) * (cx - s.tr[0]) + (cy - s.tr[1]) * (cy - s.tr[1]));
            if (dist < low_refl_range)
              n_upd += add_tsd(&T[ci], &W[ci], max_trunc, pw, max_trunc, inv_max_trunc, eps);
          }
        }
      }
  }
  propagate_borders(g);
  if (stats) {
    stats->cells_updated = n_upd; stats->cells_visited = n_vis;
    stats->tiles_total = g->tiles; stats->tiles_range_pass = n_range; stats->tiles_update = n_update;
    stats->tiles_new = n_new; stats->tiles_new_from_empty = n_new_e;
    stats->tiles_emptied_init = n_emp_i; stats->tiles_emptied_uninit = n_emp_u;
  }
}

/* ------------------------------------------------------------------------------------------- */
/* R1-R2: ray cast                                                                               */
/* ------------------------------------------------------------------------------------------- */
enum { INTERP_SUCCESS = 0, INTERP_INVALIDINDEX = 1, INTERP_EMPTYPARTITION = 2, INTERP_ISNAN = 3 };

/* TsdGrid::coord2Cell (TsdGrid.h:306-340) */
static inline int coord2cell(const ora_grid* g, double x, double y, int* p, int* lx, int* ly,
                             double* dx, double* dy)
{
  const double dcx = x * g->inv_cs, dcy = y * g->inv_cs;
  int xi = (int)floor(dcx), yi = (int)floor(dcy);
  *dx = ((double)xi + 0.5) * g->cs;
  *dy = ((double)yi + 0.5) * g->cs;
  if (x < *dx) { xi--; (*dx) -= g->cs; }
  if (y < *dy) { yi--; (*dy) -= g->cs; }
  if (xi >= g->N || xi < 0 || yi >= g->N || yi < 0) return 0;
  *p = yi / D * g->PX + xi / D;
  *lx = xi % D;
  *ly = yi % D;
  return 1;
}

/* TsdGrid::interpolateBilinear (TsdGrid.h:284-304) + TsdGridPartition::interpolateBilinear
 * (TsdGridPartition.h:214-221) */
int ora_interpolate_bilinear(const ora_grid* g, double x, double y, double* tsd)
{
  int p, lx, ly; double dx, dy;
  if (!coord2cell(g, x, y, &p, &lx, &ly, &dx, &dy)) return INTERP_INVALIDINDEX;
  if (!g->init[p]) return INTERP_EMPTYPARTITION;
  const double wx = fabs((x - dx) * g->inv_cs);
  const double wy = fabs((y - dy) * g->inv_cs);
  const double* t = g->tsd[p];
  *tsd = t[ly * PT + lx] * (1. - wy) * (1. - wx)
       + t[(ly + 1) * PT + lx] * wy * (1. - wx)
       + t[ly * PT + lx + 1] * (1. - wy) * wx
       + t[(ly + 1) * PT + lx + 1] * wy * wx;
  if (isnan(*tsd)) return INTERP_ISNAN;
  return INTERP_SUCCESS;
}

/* TsdGrid::interpolateNormal (TsdGrid.cpp:517-546) + norm2 (mathbase.h:211-218) */
static int interpolate_normal(const ora_grid* g, const double c[2], double n[2])
{
  double inc = 0, dec = 0;
  if (ora_interpolate_bilinear(g, c[0] + g->cs, c[1], &inc) != INTERP_SUCCESS) return 0;
  if (ora_interpolate_bilinear(g, c[0] - g->cs, c[1], &dec) != INTERP_SUCCESS) return 0;
  n[0] = inc - dec;
  if (ora_interpolate_bilinear(g, c[0], c[1] + g->cs, &inc) != INTERP_SUCCESS) return 0;
  if (ora_interpolate_bilinear(g, c[0], c[1] - g->cs, &dec) != INTERP_SUCCESS) return 0;
  n[1] = inc - dec;
  const double len = sqrt(n[0] * n[0] + n[1] * n[1]);
  if (fabs(len) <= 10e-6) return 1;
  n[0] /= len;
  n[1] /= len;
  return 1;
}

/* RayCastPolar2D::rayCastFromCurrentView (RayCastPolar2D.cpp:194-281) */
static int raycast_beam(const ora_grid* g, const double tr[2], const double ray[2], double gxmin,
                        double gymin, double gxmax, double gymax, double idx_min_s, double idx_max_s,
                        double c[2], double n[2])
{
  const int xDim = g->N, yDim = g->N;
  const double cs = g->cs;
  double pos[2];
  double interp = 0.0;

  double xmin = gxmin, ymin = gymin;
  if (fabs(ray[0]) > 10e-6) xmin = ((double)(ray[0] > 0.0 ? 0 : (xDim - 1) * cs) - tr[0]) / ray[0];
  if (fabs(ray[1]) > 10e-6) ymin = ((double)(ray[1] > 0.0 ? 0 : (yDim - 1) * cs) - tr[1]) / ray[1];
  double idxMin = fmax(xmin, ymin);
  idxMin = fmax(idxMin, 0.0);

  double xmax = gxmax, ymax = gymax;
  if (fabs(ray[0]) > 10e-6) xmax = ((double)(ray[0] > 0.0 ? (xDim - 1) * cs : 0) - tr[0]) / ray[0];
  if (fabs(ray[1]) > 10e-6) ymax = ((double)(ray[1] > 0.0 ? (yDim - 1) * cs : 0) - tr[1]) / ray[1];
  double idxMax = fmin(xmax, ymax);

  idxMin = fmax(idxMin, idx_min_s);
  idxMax = fmin(idxMax, idx_max_s);
  if (idxMin >= idxMax) return 0;

  const double partitionSize = (double)D;
  for (double i = idxMin; i < idxMax; i += partitionSize) {
    double tmp;
    pos[0] = tr[0] + i * ray[0];
    pos[1] = tr[1] + i * ray[1];
    int rv = ora_interpolate_bilinear(g, pos[0], pos[1], &tmp);
    if (rv != INTERP_EMPTYPARTITION && rv != INTERP_INVALIDINDEX) break;
    else idxMin = i;
  }

  double tsd_prev;
  pos[0] = tr[0] + idxMin * ray[0];
  pos[1] = tr[1] + idxMin * ray[1];
  if (ora_interpolate_bilinear(g, pos[0], pos[1], &tsd_prev) != INTERP_SUCCESS) tsd_prev = NAN;

  int found = 0;
  for (double i = idxMin; i <= idxMax; i += 1.0) {
    pos[0] += ray[0];
    pos[1] += ray[1];
    double tsd = NAN;
    if (ora_interpolate_bilinear(g, pos[0], pos[1], &tsd) != INTERP_SUCCESS) {
      tsd_prev = NAN;       /* reference assigns tsd, which is NaN on every failing path */
      continue;
    }
    if (tsd_prev > 0 && tsd < 0) { interp = tsd_prev / (tsd_prev - tsd); found = 1; break; }
    else if (tsd_prev < 0 && tsd > 0) { found = 0; break; }
    tsd_prev = tsd;
  }
  if (!found) return 0;
  c[0] = pos[0] + ray[0] * (interp - 1.0);
  c[1] = pos[1] + ray[1] * (interp - 1.0);
  return interpolate_normal(g, c, n);
}

/* RayCastPolar2D::calcCoordsFromCurrentViewMask (RayCastPolar2D.cpp:113-192) */
int ora_raycast(const ora_grid* g, const double pose[9], const double* rays, int beams,
                double min_range, double max_range, int threads, double* coords, double* normals,
                uint8_t* mask)
{
  double T[9];
  ora_mat3_inv(pose, T);
  const double tr[2] = {pose[2], pose[5]};
  double gxmin, gymin, gxmax, gymax;
  /* TsdGrid::isInsideGrid (TsdGrid.h:342-347) */
  if (tr[0] > g->min_x && tr[0] < g->max_x && tr[1] > g->min_y && tr[1] < g->max_y) {
    gxmin = -10e9; gymin = -10e9; gxmax = 10e9; gymax = 10e9;
  } else {
    gxmin = 10e9; gymin = 10e9; gxmax = -10e9; gymax = -10e9;
  }
  const double idx_min = min_range / g->cs;
  const double idx_max = max_range / g->cs;
  int cnt = 0;
  (void)threads;
#ifdef _OPENMP
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(threads) reduction(+ : cnt)
#endif
  for (int beam = 0; beam < beams; beam++) {
    const double ray[2] = {rays[beam], rays[beams + beam]};
    double c[2], n[2];
    if (raycast_beam(g, tr, ray, gxmin, gymin, gxmax, gymax, idx_min, idx_max, c, n)) {
      /* M = T * M, N = T * N (dgemm NoTrans,NoTrans; zero entries of T are skipped by gslcblas,
       * which only matters for non-finite operands) */
      double m0 = 0.0, m1 = 0.0, n0 = 0.0, n1 = 0.0;
      m0 += T[0] * c[0]; m0 += T[1] * c[1]; m0 += T[2] * 1.0;
      m1 += T[3] * c[0]; m1 += T[4] * c[1]; m1 += T[5] * 1.0;
      n0 += T[0] * n[0]; n0 += T[1] * n[1]; n0 += T[2] * 0.0;
      n1 += T[3] * n[0]; n1 += T[4] * n[1]; n1 += T[5] * 0.0;
      coords[2 * beam] = m0; coords[2 * beam + 1] = m1;
      normals[2 * beam] = n0; normals[2 * beam + 1] = n1;
      mask[beam] = 1;
      cnt++;
    } else {
      mask[beam] = 0;
    }
  }
  return cnt;
}

/* ------------------------------------------------------------------------------------------- */
/* I1-I6: registration                                                                           */
/* ------------------------------------------------------------------------------------------- */

/* DistanceFilter ctor (DistanceFilter.cpp:11-20) as called from ThreadLocalize.cpp:212 with
 * `icpIterations - 10` (int) converted to `unsigned int iterations` */
double ora_distance_filter_multiplier(double maxdist, double mindist, int icp_iterations)
{
  unsigned int iterations = (unsigned int)(icp_iterations - 10);
  double it = (double)(iterations - 1);
  if (iterations < 1) it = 1.0;
  return pow((mindist / maxdist), 1.0 / it);
}

/* exact 1-NN, FLANN kd-tree single index stand-in (FlannPairAssignment.cpp:31-54,64-92).
 * FLANN (libflann-dev, unpinned, docker/Dockerfile:24) is absent from /root/reference; published
 * behaviour restated: exact nearest neighbour under L2<double> (eps = 0), returning the SQUARED
 * distance sum_d (a_d - b_d)^2 accumulated x then y.  Tie-breaking is implementation-defined. */
typedef struct { int* idx; int n; const double* pts; int* node_axis; } kdtree;

static const double* kd_sort_pts; static int kd_sort_axis;
static int kd_cmp(const void* a, const void* b)
{
  const double va = kd_sort_pts[2 * (*(const int*)a) + kd_sort_axis];
  const double vb = kd_sort_pts[2 * (*(const int*)b) + kd_sort_axis];
  return (va > vb) - (va < vb);
}
static void kd_build_rec(kdtree* t, int lo, int hi)
{
  if (hi - lo <= 1) { if (hi > lo) t->node_axis[lo] = 0; return; }
  double mn[2] = {INFINITY, INFINITY}, mx[2] = {-INFINITY, -INFINITY};
  for (int i = lo; i < hi; i++)
    for (int a = 0; a < 2; a++) {
      double v = t->pts[2 * t->idx[i] + a];
      if (v < mn[a]) mn[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  int axis = (mx[1] - mn[1] > mx[0] - mn[0]) ? 1 : 0;
  kd_sort_pts = t->pts; kd_sort_axis = axis;
  qsort(t->idx + lo, (size_t)(hi - lo), sizeof(int), kd_cmp);
  int mid = (lo + hi) / 2;
  t->node_axis[mid] = axis;
  kd_build_rec(t, lo, mid);
  kd_build_rec(t, mid + 1, hi);
}
static void kd_search_rec(const kdtree* t, int lo, int hi, const double q[2], int* best, double* bd)
{
  if (hi <= lo) return;
  int mid = (lo + hi) / 2;
  int pi = t->idx[mid];
  double dx = q[0] - t->pts[2 * pi], dy = q[1] - t->pts[2 * pi + 1];
  double d = dx * dx + dy * dy;
  if (d < *bd || (d == *bd && pi < *best)) { *bd = d; *best = pi; }
  if (hi - lo == 1) return;
  int axis = t->node_axis[mid];
  double diff = q[axis] - t->pts[2 * pi + axis];
  if (diff <= 0) {
    kd_search_rec(t, lo, mid, q, best, bd);
    if (diff * diff <= *bd) kd_search_rec(t, mid + 1, hi, q, best, bd);
  } else {
    kd_search_rec(t, mid + 1, hi, q, best, bd);
    if (diff * diff <= *bd) kd_search_rec(t, lo, mid, q, best, bd);
  }
}

typedef struct { unsigned int un_idx, model; double d2; } recip_pair;
/* operator< of StrReciprocalPair (ReciprocalFilter.cpp:16-21) */
static int recip_cmp(const void* a, const void* b)
{
  const recip_pair* x = (const recip_pair*)a; const recip_pair* y = (const recip_pair*)b;
  if (x->model != y->model) return x->model < y->model ? -1 : 1;
  if (x->d2 != y->d2) return x->d2 < y->d2 ? -1 : 1;
  /* std::sort leaves exact ties in unspecified order; choose the earlier pair deterministically */
  return (x->un_idx > y->un_idx) - (x->un_idx < y->un_idx);
}

typedef struct {
  const double* model; int n_model; kdtree kd; int use_kd;
  double P[9]; double min_x, max_x, min_y, max_y;
  double thr, multiplier, min_sqr;
  int* pm; int* ps; double* pd; recip_pair* rp; uint8_t* premask;
} icp_ctx;

/* PairAssignment::determinePairs (PairAssignment.cpp:38-84) with the chain ThreadLocalize builds
 * (ThreadLocalize.cpp:211-221): OutOfBoundsFilter2D (OutOfBoundsFilter2D.cpp:27-37) ->
 * FlannPairAssignment::determinePairsSequential -> DistanceFilter::filter (DistanceFilter.cpp:32-64)
 * -> ReciprocalFilter::filter (ReciprocalFilter.cpp:32-78).  Returns number of pairs, written to
 * pm/ps in the order the reference hands them to the estimator (sorted by model index). */
static int determine_pairs(icp_ctx* c, const double* scene, int n_scene)
{
  /* pre-filter: S.transform(P): (0 + x*R00) + y*R01, then + t (gsl/Matrix.cpp:403-432) */
  for (int i = 0; i < n_scene; i++) {
    double wx = 0.0, wy = 0.0;
    wx += scene[2 * i] * c->P[0]; wx += scene[2 * i + 1] * c->P[1];
    wy += scene[2 * i] * c->P[3]; wy += scene[2 * i + 1] * c->P[4];
    wx += c->P[2]; wy += c->P[5];
    c->premask[i] = !(wx < c->min_x || wx > c->max_x || wy < c->min_y || wy > c->max_y);
  }
  /* NN + distance filter */
  int np = 0;
  for (int i = 0; i < n_scene; i++) {
    if (!c->premask[i]) continue;
    int best = -1; double bd = INFINITY;
    const double* q = scene + 2 * i;
    if (c->use_kd) {
      kd_search_rec(&c->kd, 0, c->n_model, q, &best, &bd);
    } else {
      for (int k = 0; k < c->n_model; k++) {
        double dx = q[0] - c->model[2 * k], dy = q[1] - c->model[2 * k + 1];
        double d = dx * dx + dy * dy;
        if (d < bd) { bd = d; best = k; }
      }
    }
    if (bd <= c->thr) { c->pm[np] = best; c->ps[np] = i; c->pd[np] = bd; np++; }
  }
  c->thr *= c->multiplier;
  if (c->thr < c->min_sqr) c->thr = c->min_sqr;
  /* reciprocal filter */
  if (np == 0) return 0;
  for (int i = 0; i < np; i++) { c->rp[i].un_idx = (unsigned)i; c->rp[i].model = (unsigned)c->pm[i]; c->rp[i].d2 = c->pd[i]; }
  qsort(c->rp, (size_t)np, sizeof(recip_pair), recip_cmp);
  int nf = 0;
  unsigned int last = c->rp[0].model;
  int* fm = (int*)malloc(sizeof(int) * (size_t)np);
  int* fs = (int*)malloc(sizeof(int) * (size_t)np);
  fm[0] = c->pm[c->rp[0].un_idx]; fs[0] = c->ps[c->rp[0].un_idx]; nf = 1;
  for (int i = 1; i < np; i++) {
    if (c->rp[i].model == last) continue;
    last = c->rp[i].model;
    fm[nf] = c->pm[c->rp[i].un_idx]; fs[nf] = c->ps[c->rp[i].un_idx]; nf++;
  }
  memcpy(c->pm, fm, sizeof(int) * (size_t)nf);
  memcpy(c->ps, fs, sizeof(int) * (size_t)nf);
  free(fm); free(fs);
  return nf;
}

static void icp_ctx_init(icp_ctx* c, const double* model, int n_model, int n_scene,
                         const double pose[9], const ora_icp_params* p)
{
  memset(c, 0, sizeof(*c));
  c->model = model; c->n_model = n_model;
  memcpy(c->P, pose, sizeof(c->P));
  c->min_x = p->min_x; c->max_x = p->max_x; c->min_y = p->min_y; c->max_y = p->max_y;
  c->thr = p->dist_filter_max * p->dist_filter_max;        /* DistanceFilter::reset */
  c->min_sqr = p->dist_filter_min * p->dist_filter_min;
  c->multiplier = ora_distance_filter_multiplier(p->dist_filter_max, p->dist_filter_min, p->iterations);
  c->pm = (int*)malloc(sizeof(int) * (size_t)(n_scene + 1));
  c->ps = (int*)malloc(sizeof(int) * (size_t)(n_scene + 1));
  c->pd = (double*)malloc(sizeof(double) * (size_t)(n_scene + 1));
  c->rp = (recip_pair*)malloc(sizeof(recip_pair) * (size_t)(n_scene + 1));
  c->premask = (uint8_t*)malloc((size_t)(n_scene + 1));
  c->use_kd = (p->nn_mode == 1) && n_model > 0;
  if (c->use_kd) {                                         /* FlannPairAssignment::setModel */
    c->kd.n = n_model; c->kd.pts = model;
    c->kd.idx = (int*)malloc(sizeof(int) * (size_t)n_model);
    c->kd.node_axis = (int*)calloc((size_t)n_model, sizeof(int));
    for (int i = 0; i < n_model; i++) c->kd.idx[i] = i;
    kd_build_rec(&c->kd, 0, n_model);
  }
}
static void icp_ctx_free(icp_ctx* c)
{
  free(c->pm); free(c->ps); free(c->pd); free(c->rp); free(c->premask);
  if (c->use_kd) { free(c->kd.idx); free(c->kd.node_axis); }
}

int ora_icp_pairs(const double* model, int n_model, const double* scene, int n_scene,
                  const double pose[9], const ora_icp_params* p, double* thr_sqr, int* pair_model,
                  int* pair_scene)
{
  icp_ctx c;
  icp_ctx_init(&c, model, n_model, n_scene, pose, p);
  c.thr = *thr_sqr;
  int n = determine_pairs(&c, scene, n_scene);
  memcpy(pair_model, c.pm, sizeof(int) * (size_t)n);
  memcpy(pair_scene, c.ps, sizeof(int) * (size_t)n);
  *thr_sqr = c.thr;
  icp_ctx_free(&c);
  return n;
}

/* obvious::Matrix::solve (gsl/Matrix.cpp:343-355): gsl_linalg_LU_decomp (partial pivoting, first maximum) +
 * gsl_linalg_LU_solve (x = P b, unit-lower forward substitution, upper back substitution); 3 x 3 */
static void lu3_solve(const double A[9], const double b[3], double x[3])
{
  double lu[9];
  int perm[3] = {0, 1, 2};
  memcpy(lu, A, sizeof(lu));
  for (int j = 0; j < 3; j++) {
    int piv = j;
    double best = fabs(lu[3 * j + j]);
    for (int i = j + 1; i < 3; i++)
      if (fabs(lu[3 * i + j]) > best) { best = fabs(lu[3 * i + j]); piv = i; }
    if (piv != j) {
      for (int k = 0; k < 3; k++) { double t = lu[3 * j + k]; lu[3 * j + k] = lu[3 * piv + k]; lu[3 * piv + k] = t; }
      int t = perm[j]; perm[j] = perm[piv]; perm[piv] = t;
    }
    for (int i = j + 1; i < 3; i++) {
      lu[3 * i + j] = lu[3 * i + j] / lu[3 * j + j];
      for (int k = j + 1; k < 3; k++) lu[3 * i + k] -= lu[3 * i + j] * lu[3 * j + k];
    }
  }
  for (int i = 0; i < 3; i++) x[i] = b[perm[i]];
  for (int i = 1; i < 3; i++)
    for (int k = 0; k < i; k++) x[i] -= lu[3 * i + k] * x[k];
  for (int i = 2; i >= 0; i--) {
    for (int k = i + 1; k < 3; k++) x[i] -= lu[3 * i + k] * x[k];
    x[i] = x[i] / lu[3 * i + i];
  }
}

/* Icp::iterate (Icp.cpp:464-512) driving Icp::step (:410-462), Tinit = identity (registration_mode 0), with
 * ClosedFormEstimator2D (ClosedFormEstimator2D.cpp:36-109) when `normals` is NULL -- what the node constructs
 * (ThreadLocalize.cpp:214) -- or PointToLine2DEstimator (PointToLineEstimator2D.cpp:52-157) on the model normals. */
static void icp_impl(const double* model, const double* normals, int n_model, const double* scene_in, int n_scene,
                     const double pose[9], const ora_icp_params* p, ora_icp_result* out, double* trace, const double* Tinit33)
{
  enum { PROCESSING = 1, NOTMATCHABLE = 2, MAXITERATIONS = 3, SUCCESS = 5 };
  double Tf[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   /* _Tfinal4x4->setIdentity() */
  out->rms = 0.0; out->pairs = 0; out->iterations = 0;                 /* ThreadLocalize.cpp:577-579 */
  if (n_model == 0 || n_scene == 0) {                                  /* Icp.cpp:467-471 */
    out->state = NOTMATCHABLE;
    for (int i = 0; i < 9; i++) out->T[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  double* sc = (double*)malloc(sizeof(double) * 2 * (size_t)n_scene);
  /* applyTransformation(sceneTmp, Tinit) (Icp.cpp:481-486, :371-408): (0 + x*R00) + y*R01, then + t.  Tinit is the
   * identity in registration_mode 0 and the pre-registration's result otherwise (ThreadLocalize.cpp:531-569) */
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double* Ti = Tinit33 ? Tinit33 : I3;
  for (int i = 0; i < n_scene; i++) {
    double x = scene_in[2 * i], y = scene_in[2 * i + 1];
    double nx = 0.0, ny = 0.0;
    nx += x * Ti[0]; nx += y * Ti[1]; ny += x * Ti[3]; ny += y * Ti[4];
    sc[2 * i] = nx + Ti[2]; sc[2 * i + 1] = ny + Ti[5];
  }
  /* (*_Tfinal4x4) = (*Tinit) * (*_Tfinal4x4) with _Tfinal4x4 == I (Icp.cpp:485) */
  Tf[0] = Ti[0]; Tf[1] = Ti[1]; Tf[3] = Ti[2]; Tf[4] = Ti[3]; Tf[5] = Ti[4]; Tf[7] = Ti[5];
  icp_ctx c;
  icp_ctx_init(&c, model, n_model, n_scene, pose, p);

  int state = PROCESSING;
  unsigned int iter = 0, conv_cnt = 0;
  double rms = 0.0, rms_prev = 10e12;
  int pairs = 0;
  const unsigned int max_it = (unsigned)p->iterations, conv_need = (unsigned)p->iterations;
  while (state == PROCESSING) {
    const double thr_before = c.thr;
    /* ---- step ---- */
    pairs = determine_pairs(&c, sc, n_scene);
    double tl_co = NAN, tl_si = NAN, tl_dx = NAN, tl_dy = NAN;        /* Tlast of this step (trace) */
    if (pairs > 2) {
      double co, si, dX, dY;
      if (normals) {
        /* PointToLine2DEstimator::setPairs: rms = mean |(scene - model) . normal| (:52-77) */
        rms = 0.0;
        for (int i = 0; i < pairs; i++) {
          const double* pm = model + 2 * c.pm[i];
          const double* ps = sc + 2 * c.ps[i];
          const double* pn = normals + 2 * c.pm[i];
          double v0 = ps[0] - pm[0], v1 = ps[1] - pm[1];
          rms += fabs(v0 * pn[0] + v1 * pn[1]);
        }
        rms /= (double)pairs;
        /* estimateTransformation (:89-157): 3x3 normal equations over (a_z, n_x, n_y), LU solve */
        double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
        for (int i = 0; i < pairs; i++) {
          const double* q = model + 2 * c.pm[i];
          const double* pp = sc + 2 * c.ps[i];
          const double* n = normals + 2 * c.pm[i];
          const double az = pp[0] * n[1] - pp[1] * n[0];
          A[0] += az * az;   A[1] += az * n[0];   A[2] += az * n[1];
          A[3] += az * n[0]; A[4] += n[0] * n[0]; A[5] += n[0] * n[1];
          A[6] += az * n[1]; A[7] += n[1] * n[0]; A[8] += n[1] * n[1];
          const double pq0 = pp[0] - q[0], pq1 = pp[1] - q[1];
          const double tmp = pq0 * n[0] + pq1 * n[1];
          b[0] -= az * tmp; b[1] -= n[0] * tmp; b[2] -= n[1] * tmp;
        }
        double x[3];
        lu3_solve(A, b, x);
        co = cos(x[0]); si = sin(x[0]); dX = x[1]; dY = x[2];
      } else {
      /* ClosedFormEstimator2D::setPairs */
      double cm[2] = {0, 0}, cs_[2] = {0, 0};
      rms = 0.0;
      for (int i = 0; i < pairs; i++) {
        const double* pm = model + 2 * c.pm[i];
        const double* ps = sc + 2 * c.ps[i];
        cm[0] += pm[0]; cm[1] += pm[1];
        cs_[0] += ps[0]; cs_[1] += ps[1];
        { double dx = ps[0] - pm[0], dy = ps[1] - pm[1]; rms += dx * dx + dy * dy; }  /* distSqr2D(model, scene) */
      }
      const double size_inv = 1.0 / (double)pairs;
      rms *= size_inv; cm[0] *= size_inv; cm[1] *= size_inv; cs_[0] *= size_inv; cs_[1] *= size_inv;
      /* estimateTransformation */
      double nom = 0.0, den = 0.0;
      for (int i = 0; i < pairs; i++) {
        double xF = model[2 * c.pm[i]] - cm[0], yF = model[2 * c.pm[i] + 1] - cm[1];
        double xS = sc[2 * c.ps[i]] - cs_[0], yS = sc[2 * c.ps[i] + 1] - cs_[1];
        nom += yF * xS - xF * yS;
        den += xF * xS + yF * yS;
      }
      const double th = atan2(nom, den);
      co = cos(th); si = sin(th);
      dX = (cm[0] - (co * cs_[0] - si * cs_[1]));
      dY = (cm[1] - (co * cs_[1] + si * cs_[0]));
      }
      double Tl[16] = {co, -si, 0, dX, si, co, 0, dY, 0, 0, 1, 0, 0, 0, 0, 1};
      tl_co = co; tl_si = si; tl_dx = dX; tl_dy = dY;
      /* applyTransformation: Matrix::multiply(R, data) = data * R^T via dgemm(NoTrans,Trans)
       * (gsl/Matrix.cpp:489-497), then translation */
      for (int i = 0; i < n_scene; i++) {
        double x = sc[2 * i], y = sc[2 * i + 1];
        double nx = 0.0, ny = 0.0;
        nx += x * co; nx += y * (-si);
        ny += x * si; ny += y * co;
        sc[2 * i] = nx + dX; sc[2 * i + 1] = ny + dY;
      }
      /* Tfinal = Tlast * Tfinal (4x4 dgemm) */
      double R[16];
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
          double t = 0.0;
          for (int k = 0; k < 4; k++) t += Tl[4 * i + k] * Tf[4 * k + j];
          R[4 * i + j] = t;
        }
      memcpy(Tf, R, sizeof(R));
      state = PROCESSING;
    } else {
      state = NOTMATCHABLE;
    }
    /* ---- loop control ---- */
    iter++;
    if (fabs(rms - rms_prev) < 10e-10) conv_cnt++; else conv_cnt = 0;
    if (rms <= 0.0 /* _maxRMS */ || conv_cnt >= conv_need) state = SUCCESS;
    else if (iter >= max_it) state = MAXITERATIONS;
    rms_prev = rms;
    if (trace) {
      double* tr = trace + 8 * (iter - 1);       /* layout of include/tsd_hip.h: tsd_icp_trace */
      tr[0] = (double)pairs; tr[1] = rms; tr[2] = thr_before; tr[3] = (double)state;
      tr[4] = tl_co; tr[5] = tl_si; tr[6] = tl_dx; tr[7] = tl_dy;
    }
  }
  /* getFinalTransformation (Icp.cpp:528-546) */
  out->T[0] = Tf[0]; out->T[1] = Tf[1]; out->T[2] = Tf[3];
  out->T[3] = Tf[4]; out->T[4] = Tf[5]; out->T[5] = Tf[7];
  out->T[6] = 0; out->T[7] = 0; out->T[8] = 1;
  out->rms = rms; out->pairs = pairs; out->iterations = (int)iter; out->state = state;
  icp_ctx_free(&c);
  free(sc);
}

void ora_icp(const double* model, int n_model, const double* scene_in, int n_scene,
             const double pose[9], const ora_icp_params* p, ora_icp_result* out, double* trace)
{
  icp_impl(model, NULL, n_model, scene_in, n_scene, pose, p, out, trace, NULL);
}

/* Icp::iterate with a pre-registration result as Tinit (registration_mode 1-3, ThreadLocalize.cpp:580) */
void ora_icp_init(const double* model, int n_model, const double* scene_in, int n_scene, const double pose[9],
                  const ora_icp_params* p, const double Tinit33[9], ora_icp_result* out, double* trace)
{
  icp_impl(model, NULL, n_model, scene_in, n_scene, pose, p, out, trace, Tinit33);
}

/* the same registration with PointToLine2DEstimator on the model normals Icp::setModel(coords, normals) takes */
void ora_icp_point_to_line(const double* model, const double* normals, int n_model, const double* scene_in, int n_scene,
                           const double pose[9], const ora_icp_params* p, ora_icp_result* out, double* trace)
{
  icp_impl(model, normals, n_model, scene_in, n_scene, pose, p, out, trace, NULL);
}

/* ------------------------------------------------------------------------------------------------
 * Row N3: TSD_PDFMatching::match (registration/ransacMatching/TSD_PDFMatching.cpp:31-294), the pre-registration
 * ThreadLocalize runs in registration_mode 3 (ThreadLocalize.cpp:557-567) -- what config/single-laser.yaml ships.
 * The reference draws from rand() in three places: RandomMatching::subsampleMask (RandomMatching.cpp:176-189),
 * RandomMatching::pickControlSet (:65) and the trial pick (TSD_PDFMatching.cpp:190-194), the last one after
 * srand(time(NULL)) and inside an OpenMP loop.  Here the draws are INPUTS (the raw rand() values, in the order
 * of a serial run), and the trials run in trial order, so the result is a function of its arguments:
 * "first candidate that reaches the best probability wins", which is what the strict `>` of :264 gives a serial
 * run.  PARITY UNPINNED: the translation unit needs GSL (Matrix, pcaAnalysis) and cannot be compiled here.
 *
 * Matrix::pcaAnalysis (obcore/math/linalg/gsl/Matrix.cpp:227-327) for n x 2 input: centroid by gsl_stats_mean
 * (running mean in long double, GSL 2.7.1 statistics/mean_source.c), M^T M by dgemm, gsl_linalg_SV_decomp_jacobi of
 * the symmetric 2 x 2 product (restated as its closed-form eigen-decomposition: the same V up to the sign of a
 * column and rounding, and every use below is invariant to the sign), extents along the two axes. */
static void pca2_axes(const double* pts, int n, double axes[2][4])
{
  double cent[2];
  for (int j = 0; j < 2; j++) {
    long double mean = 0.0L;
    for (int i = 0; i < n; i++) mean += ((long double)pts[2 * i + j] - mean) / (long double)(i + 1);
    cent[j] = (double)mean;
  }
  double* Mc = (double*)malloc(sizeof(double) * 2 * (size_t)n);
  for (int i = 0; i < n; i++) { Mc[2 * i] = pts[2 * i] + (-cent[0]); Mc[2 * i + 1] = pts[2 * i + 1] + (-cent[1]); }
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = 0; i < n; i++) { a += Mc[2 * i] * Mc[2 * i]; b += Mc[2 * i] * Mc[2 * i + 1]; c += Mc[2 * i + 1] * Mc[2 * i + 1]; }
  /* eigenvectors of [[a b][b c]], major axis first */
  const double th = 0.5 * atan2(2.0 * b, a - c);
  const double V[2][2] = {{cos(th), -sin(th)}, {sin(th), cos(th)}};          /* V[j][i]: component j of eigenvector i */
  double mx[2], mn[2];
  for (int i = 0; i < 2; i++) {
    mx[i] = -INFINITY; mn[i] = INFINITY;
    for (int r = 0; r < n; r++) {
      double pr = 0.0;
      pr += V[0][i] * Mc[2 * r]; pr += V[1][i] * Mc[2 * r + 1];               /* P = V^T M^T */
      if (pr > mx[i]) mx[i] = pr;
      if (pr < mn[i]) mn[i] = pr;
    }
  }
  for (int i = 0; i < 2; i++) {
    const double ext = mx[i] - mn[i];
    double align = 0.0;
    if (ext > 1e-6) align = (mx[i] + mn[i]) / 2.0;
    for (int j = 0; j < 2; j++) cent[j] += V[j][i] * align;
  }
  for (int i = 0; i < 2; i++) {
    const double ext = mx[i] - mn[i];
    for (int j = 0; j < 2; j++) {
      const double e = V[j][i] * ext / 2.0;
      axes[i][2 * j] = cent[j] - e; axes[i][2 * j + 1] = cent[j] + e;
    }
  }
  free(Mc);
}

/* RandomMatching::calcNormals (RandomMatching.cpp:82-153) */
static void rm_calc_normals(const double* M, int points, double* N, const uint8_t* mask_in, uint8_t* mask_out, int sr)
{
  for (int i = 0; i < sr && i < points; i++) mask_out[i] = 0;
  for (int i = points - sr; i < points; i++) if (i >= 0) mask_out[i] = 0;
  double* A = (double*)malloc(sizeof(double) * 2 * (size_t)(2 * sr + 1));
  for (int i = sr; i < points - sr; i++) {
    if (!mask_in[i]) continue;
    unsigned cnt = 0;
    for (int j = -sr; j < sr; j++) if (mask_in[i + j]) cnt++;
    if (cnt > 3) {
      cnt = 0;
      for (int j = -sr; j < sr; j++) if (mask_in[i + j]) { A[2 * cnt] = M[2 * (i + j)]; A[2 * cnt + 1] = M[2 * (i + j) + 1]; cnt++; }
      double ax[2][4];
      pca2_axes(A, (int)cnt, ax);
      const double xLong = ax[0][1] - ax[0][0], yLong = ax[0][3] - ax[0][2];
      const double xShort = ax[1][1] - ax[1][0], yShort = ax[1][3] - ax[1][2];
      const double lenLongSqr = xLong * xLong + yLong * yLong, lenShortSqr = xShort * xShort + yShort * yShort;
      if (lenShortSqr > 1e-6 && (lenLongSqr / lenShortSqr) < 4.0) { mask_out[i] = 0; continue; }
      const double len = sqrt(lenShortSqr);
      if ((M[2 * i] * xShort + M[2 * i + 1] * yShort) < 0.0) { N[2 * i] = xShort / len; N[2 * i + 1] = yShort / len; }
      else { N[2 * i] = -xShort / len; N[2 * i + 1] = -yShort / len; }
    } else mask_out[i] = 0;
  }
  free(A);
}

int ora_tsdpdf_match(const ora_grid* g, const double pose[9], const double* M, const uint8_t* maskM, const double* S,
                     const uint8_t* maskS, int n, int trials_cfg, int size_control_set, double zrand, double phi_max,
                     double resolution, const int* draws_subsample, const int* draws_control, const int* draws_trials,
                     double T_out[9], double* best_prob_out, int* best_idx_out, int* best_i_out, int* candidates_out)
{
  const int SR = 10 / 2;                                               /* _pcaSearchRange / 2 */
  for (int i = 0; i < 9; i++) T_out[i] = (i % 4 == 0) ? 1.0 : 0.0;     /* TBest.setIdentity() */
  if (best_prob_out) *best_prob_out = 0.0;
  if (best_idx_out) *best_idx_out = -1;
  if (best_i_out) *best_i_out = -1;
  if (candidates_out) *candidates_out = 0;
  if (n < 3) return 1;
  double* NM = (double*)calloc(2 * (size_t)n, sizeof(double));
  double* NS = (double*)calloc(2 * (size_t)n, sizeof(double));
  double* phiM = (double*)malloc(sizeof(double) * (size_t)n);
  double* phiS = (double*)malloc(sizeof(double) * (size_t)n);
  uint8_t* mMp = (uint8_t*)malloc((size_t)n);
  uint8_t* mSp = (uint8_t*)malloc((size_t)n);
  int* idxM = (int*)malloc(sizeof(int) * (size_t)n);
  int* idxS = (int*)malloc(sizeof(int) * (size_t)n);
  int* tmp = (int*)malloc(sizeof(int) * (size_t)n);
  int nMv = 0, nSv = 0, rc = 0;
  /* model (:63-77) */
  memcpy(mMp, maskM, (size_t)n);
  rm_calc_normals(M, n, NM, maskM, mMp, SR);
  for (int i = 0; i < n; i++) phiM[i] = mMp[i] ? atan2(NM[2 * i + 1], NM[2 * i]) : -1e6;      /* calcPhi */
  for (int i = SR; i < n - SR; i++) if (mMp[i]) idxM[nMv++] = i;                              /* extractSamples */
  /* scene (:81-102) */
  memcpy(mSp, maskS, (size_t)n);
  unsigned valid = 0;
  for (int i = 0; i < n; i++) if (mSp[i]) valid++;
  double probability = 180.0 / (double)valid;
  if (probability < 0.99) {                                            /* subsampleMask (RandomMatching.cpp:176-189) */
    if (probability > 1.0) probability = 1.0;
    if (probability < 0.0) probability = 0.0;
    const int thresh = (int)(1000.0 - probability * 1000.0 + 0.5);
    for (int i = 0; i < n; i++) if ((draws_subsample[i] % 1000) < thresh) mSp[i] = 0;
  }
  rm_calc_normals(S, n, NS, maskS, mSp, SR);
  for (int i = 0; i < n; i++) phiS[i] = mSp[i] ? atan2(NS[2 * i + 1], NS[2 * i]) : -1e6;
  for (int i = SR; i < n - SR; i++) if (mSp[i]) idxS[nSv++] = i;
  /* control set (:106-118, RandomMatching::pickControlSet) */
  int nC = size_control_set;
  if (nSv < nC) nC = nSv;
  double* C = (double*)malloc(sizeof(double) * 2 * (size_t)(nC > 0 ? nC : 1));
  {
    int nt = nSv;
    memcpy(tmp, idxS, sizeof(int) * (size_t)nSv);
    for (int k = 0; k < nC; k++) {
      const unsigned r = (unsigned)draws_control[k] % (unsigned)nt;
      const int idx = tmp[r];
      memmove(tmp + r, tmp + r + 1, sizeof(int) * (size_t)(nt - (int)r - 1)); nt--;
      C[2 * k] = S[2 * idx]; C[2 * k + 1] = S[2 * idx + 1];
    }
  }
  if (nSv < 3 || nMv < 3) { rc = 1; goto done; }                         /* :129-139 */
  {
    int trials = trials_cfg;
    if (nMv < trials) trials = nMv;
    if (phi_max > M_PI * 0.5) phi_max = M_PI * 0.5;                      /* min(phiMax, pi/2) */
    int span;
    if (resolution > 1e-6) { span = (int)floor(phi_max / resolution); if (span > n) span = n; }
    else { rc = 2; goto done; }
    double best = 0.0;
    int nt = nMv, cand = 0;
    memcpy(tmp, idxM, sizeof(int) * (size_t)nMv);
    for (int trial = 0; trial < trials; trial++) {
      const int r = (int)((unsigned)draws_trials[trial] % (unsigned)nt);
      const int idx = tmp[r];
      memmove(tmp + r, tmp + r + 1, sizeof(int) * (size_t)(nt - r - 1)); nt--;
      const int iMin = (idx - span > SR) ? idx - span : SR;
      const int iMax = (idx + span < n - SR) ? idx + span : n - SR;
      for (int i = iMin; i < iMax; i++) {
        if (!mSp[i]) continue;
        double phi = phiM[idx] - phiS[i];
        if (phi > M_PI) phi -= 2.0 * M_PI; else if (phi < -M_PI) phi += 2.0 * M_PI;
        if (!(fabs(phi) < phi_max)) continue;
        cand++;
        /* T = TransformationMatrix33(phi, 0, 0) + translation (:217-223) */
        double T[9] = {cos(phi), -sin(phi), 0, sin(phi), cos(phi), 0, 0, 0, 1};
        const double sx = S[2 * i], sy = S[2 * i + 1];
        T[2] = M[2 * idx] - (T[0] * sx + T[1] * sy);
        T[5] = M[2 * idx + 1] - (T[3] * sx + T[4] * sy);
        double TMap[9];
        ora_mat3_mul(pose, T, TMap);                                     /* TSensor * T (dgemm) */
        double prob = 1.0;
        for (int s = 0; s < nC; s++) {
          /* STemp = TMap * Control, Control = [x; y; 1] (dgemm: k ascending from 0.0) */
          double cx = 0.0, cy = 0.0;
          cx += TMap[0] * C[2 * s]; cx += TMap[1] * C[2 * s + 1]; cx += TMap[2] * 1.0;
          cy += TMap[3] * C[2 * s]; cy += TMap[4] * C[2 * s + 1]; cy += TMap[5] * 1.0;
          double tsd;
          if (ora_interpolate_bilinear(g, cx, cy, &tsd) == 0) prob *= (1.0 - (1.0 - zrand) * fabs(tsd));
          else prob *= zrand;
        }
        if (prob > best) {
          memcpy(T_out, T, sizeof(T));
          best = prob;
          if (best_idx_out) *best_idx_out = idx;
          if (best_i_out) *best_i_out = i;
        }
      }
    }
    if (best_prob_out) *best_prob_out = best;
    if (candidates_out) *candidates_out = cand;
  }
done:
  free(NM); free(NS); free(phiM); free(phiS); free(mMp); free(mSp); free(idxM); free(idxS); free(tmp); free(C);
  return rc;
}

/* TsdGrid::grid2ColorImage (TsdGrid.cpp:429-488), what ThreadGrid publishes next to the occupancy map
 * (ThreadGrid.cpp:125): rgb[3 * (h * width + w)].  px / py by REPEATED addition of the step like the reference. */
void ora_grid_color_image(const ora_grid* g, unsigned char* image, unsigned int width, unsigned int height)
{
  const double stepW = g->max_x / (double)width;
  const double stepH = g->max_y / (double)height;
  double py = 0.0;
  size_t i = 0;
  for (unsigned int h = 0; h < height; h++) {
    double px = 0.0;
    for (unsigned int w = 0; w < width; w++, i++) {
      int p, x, y; double dx, dy;
      double tsd = NAN;
      int is_empty = 0;
      if (coord2cell(g, px, py, &p, &x, &y, &dx, &dy)) {
        if (g->init[p]) tsd = g->tsd[p][y * PT + x];
        is_empty = !g->init[p] && g->init_weight[p] > 0.0;     /* isEmpty(), TsdGridPartition.h:72 */
      }
      unsigned char rgb[3];
      if (tsd > 0.0) { rgb[0] = (unsigned char)(tsd * 255.0); rgb[1] = 255; rgb[2] = (unsigned char)(tsd * 255.0); }
      else if (tsd < 0.0) { rgb[0] = (unsigned char)((1.0 + tsd) * 255.0); rgb[1] = 0; rgb[2] = 0; }
      else if (is_empty) { rgb[0] = 255; rgb[1] = 255; rgb[2] = 255; }
      else { rgb[0] = 0; rgb[1] = 0; rgb[2] = 0; }
      memcpy(image + 3 * i, rgb, 3);
      px += stepW;
    }
    py += stepH;
  }
}

/* ------------------------------------------------------------------------------------------- */
/* L1: gates                                                                                     */
/* ------------------------------------------------------------------------------------------- */

/* ThreadLocalize::calcAngle (ThreadLocalize.cpp:715-726) */
double ora_calc_angle(const double T[9])
{
  double angle = 0.0;
  const double ARCSIN = asin(T[3]);
  const double ARCSINEG = asin(T[1]);
  const double ARCOS = acos(T[0]);
  if ((ARCSIN > 0.0) && (ARCSINEG < 0.0)) angle = ARCOS;
  else if ((ARCSIN < 0.0) && (ARCSINEG > 0.0)) angle = 2.0 * M_PI - ARCOS;
  return angle;
}

/* ThreadLocalize::isRegistrationError (ThreadLocalize.cpp:593-600) */
int ora_is_registration_error(const double T[9], double trs_max, double sin_rot_max)
{
  const double dx = T[2], dy = T[5];
  const double trs = sqrt(dx * dx + dy * dy);
  const double dphi = ora_calc_angle(T);
  return (trs > trs_max) || (fabs(sin(dphi)) > sin_rot_max);
}

/* ThreadLocalize::isPoseChangeSignificant (ThreadLocalize.cpp:728-736), ROT_MIN 0.03, TRNS_MIN 0.05
 * (ThreadLocalize.h:63-64) */
int ora_is_pose_change_significant(const double last[9], const double cur[9])
{
  const double dx = cur[2] - last[2], dy = cur[5] - last[5];
  double dphi = ora_calc_angle(cur) - ora_calc_angle(last);
  dphi = fabs(sin(dphi));
  const double trs = sqrt(dx * dx + dy * dy);
  return (dphi > 0.03 || trs > 0.05);
}

/* ------------------------------------------------------------------------------------------- */
/* N1: occupancy extraction                                                                      */
/* RayCastAxisAligned2D::calcCoords (RayCastAxisAligned2D.cpp:13-105) with occupiedGrid, then the */
/* marking loop of ThreadGrid::eventLoop (ThreadGrid.cpp:88-118).  `occ` persists between calls   */
/* in the reference (_occGridContent, initialised to -1 in ThreadGrid.cpp:27-28).                 */
/* ------------------------------------------------------------------------------------------- */
int ora_occupancy(const ora_grid* g, int8_t* occ, int8_t* out, int inflate, int inflate_factor)
{
  const unsigned int PXn = (unsigned)g->PX, Nn = (unsigned)g->N;
  const double cs = g->cs;
  const unsigned int cellsPPart = D * D, cellsPPX = D;
  size_t cap = 1024, cnt = 0;
  double* coords = (double*)malloc(cap * sizeof(double));
  for (unsigned int y = 1; y + 1 < PXn; y++)
    for (unsigned int x = 1; x + 1 < PXn; x++) {
      const int p = (int)(y * PXn + x);
      const unsigned int off = y * cellsPPart * PXn + x * cellsPPX;
      if (g->init[p]) {
        /* isEmpty() is false for initialised tiles (TsdGridPartition.h:72) */
        const double* t = g->tsd[p];
        for (unsigned int py = 0; py < D + 1; py++) {
          double prev = t[py * PT + 0];
          occ[off + py * Nn] = ((prev > 0.0) ? 0 : -1);
          for (unsigned int px = 1; px < D + 1; px++) {
            double v = t[py * PT + px];
            occ[off + py * Nn + px] = ((v > 0.0) ? 0 : -1);
            if ((prev > 0 && v < 0) || (prev < 0 && v > 0)) {
              double interp = prev / (prev - v);
              if (cnt + 2 > cap) { cap *= 2; coords = (double*)realloc(coords, cap * sizeof(double)); }
              coords[cnt] = px * cs + cs * (interp - 1.0) + (x * D) * cs;
              coords[cnt + 1] = py * cs + (y * D) * cs;
              cnt += 2;
            }
            prev = v;
          }
        }
        for (unsigned int px = 0; px < D + 1; px++) {
          double prev = t[0 * PT + px];
          for (unsigned int py = 1; py < D + 1; py++) {
            double v = t[py * PT + px];
            if ((prev > 0 && v < 0) || (prev < 0 && v > 0)) {
              double interp = prev / (prev - v);
              if (cnt + 2 > cap) { cap *= 2; coords = (double*)realloc(coords, cap * sizeof(double)); }
              coords[cnt] = px * cs + (x * D) * cs;
              coords[cnt + 1] = py * cs + cs * (interp - 1.0) + (y * D) * cs;
              cnt += 2;
            }
            prev = v;
          }
        }
      } else if (g->init_weight[p] > 0.0) {   /* isEmpty() */
        for (unsigned int py = 0; py < D; py++) {
          occ[off + py * Nn] = 0;
          for (unsigned int px = 1; px < D; px++) occ[off + py * Nn + px] = 0;
        }
      }
    }
  /* ThreadGrid: _occGrid->data = _occGridContent, then the marks go into the COPY (:91-118) */
  memcpy(out, occ, (size_t)Nn * Nn);
  occ = out;
  int marked = 0;
  for (size_t i = 0; i < cnt / 2; i++) {
    double x = coords[2 * i], y = coords[2 * i + 1];
    unsigned int u = (unsigned int)round(x / cs);
    unsigned int v = (unsigned int)round(y / cs);
    if (u > 0 && u < Nn && v > 0 && v < Nn) {
      occ[v * Nn + u] = 100;
      marked++;
      if (inflate)
        for (unsigned int ii = v - (unsigned)inflate_factor; ii < v + (unsigned)inflate_factor; ii++)
          for (unsigned int jj = u - (unsigned)inflate_factor; jj < u + (unsigned)inflate_factor; jj++) {
            if (u >= Nn || v >= Nn) continue;
            if ((size_t)ii * Nn + jj < (size_t)Nn * Nn) occ[ii * Nn + jj] = 100;   /* reference: unchecked */
          }
    }
  }
  free(coords);
  (void)marked;
  return (int)(cnt / 2);
}

/* ------------------------------------------------------------------------------------------- */
/* Whole-loop driver                                                                             */
/* ------------------------------------------------------------------------------------------- */
struct ora_slam {
  ora_slam_config cfg;
  ora_grid* grid;
  int initialized, reverse;
  int beams; double ang_res, phi_min;
  double pose[9], last_pose[9];
  double* rays; double* rays_local; double ray_norm;
  double* data; uint8_t* mask;
  double* model; double* normals; uint8_t* mask_m; double* scene; uint8_t* mask_s;
  int have_last_pose;
  ora_push_stats last_stats;
  /* registration_mode 3: the rand() draws of the next scan's TSD_PDFMatching::match (caller-supplied) */
  int* draws_sub; int* draws_ctrl; int* draws_trials;
  double pre_T[9]; double pre_prob; int pre_idx, pre_i;
  int owns_grid, skip_init_push;      /* multi-robot mode: several localisers on ONE grid (SlamNode.cpp:101-122) */
};

ora_slam* ora_slam_create(const ora_slam_config* cfg)
{
  ora_slam* s = (ora_slam*)calloc(1, sizeof(ora_slam));
  s->cfg = *cfg;
  /* SlamNode::initialize (SlamNode.cpp:77-78) */
  s->grid = ora_grid_create(cfg->map_size_log2, cfg->cell_size, (double)cfg->truncation_radius * cfg->cell_size);
  s->owns_grid = 1;
  s->beams = cfg->beams;
  size_t B = (size_t)cfg->beams;
  s->rays = (double*)malloc(2 * B * sizeof(double));
  s->rays_local = (double*)malloc(2 * B * sizeof(double));
  s->data = (double*)malloc(B * sizeof(double));
  s->mask = (uint8_t*)malloc(B);
  s->model = (double*)calloc(2 * B, sizeof(double));
  s->normals = (double*)calloc(2 * B, sizeof(double));
  s->mask_m = (uint8_t*)malloc(B);
  s->scene = (double*)calloc(2 * B, sizeof(double));
  s->mask_s = (uint8_t*)malloc(B);
  return s;
}

void ora_slam_destroy(ora_slam* s)
{
  if (!s) return;
  if (s->owns_grid) ora_grid_destroy(s->grid);
  free(s->draws_sub); free(s->draws_ctrl); free(s->draws_trials);
  free(s->rays); free(s->rays_local); free(s->data); free(s->mask);
  free(s->model); free(s->normals); free(s->mask_m); free(s->scene); free(s->mask_s);
  free(s);
}

ora_grid* ora_slam_grid(ora_slam* s) { return s->grid; }
/* a further localiser on the grid of `first` (SlamNode.cpp:101-122: N ThreadLocalize, one TsdGrid, one ThreadMapping):
 * its init frees its footprint but does not push (the mapper is initialised by then, ThreadLocalize.cpp:506-507) */
ora_slam* ora_slam_create_shared(const ora_slam_config* cfg, ora_slam* first)
{
  ora_slam* s = ora_slam_create(cfg);
  ora_grid_destroy(s->grid);
  s->grid = first->grid; s->owns_grid = 0; s->skip_init_push = 1;
  return s;
}
void ora_slam_set_draws(ora_slam* s, const int* sub, const int* ctrl, const int* trials)
{
  const size_t B = (size_t)s->beams;
  if (!s->draws_sub) {
    s->draws_sub = (int*)malloc(B * sizeof(int));
    s->draws_ctrl = (int*)malloc((size_t)(s->cfg.size_control_set > 0 ? s->cfg.size_control_set : 1) * sizeof(int));
    s->draws_trials = (int*)malloc((size_t)(s->cfg.trials > 0 ? s->cfg.trials : 1) * sizeof(int));
  }
  memcpy(s->draws_sub, sub, B * sizeof(int));
  memcpy(s->draws_ctrl, ctrl, (size_t)s->cfg.size_control_set * sizeof(int));
  memcpy(s->draws_trials, trials, (size_t)s->cfg.trials * sizeof(int));
}
void ora_slam_last_prereg(const ora_slam* s, double T[9], double* prob, int* idx, int* i)
{
  memcpy(T, s->pre_T, sizeof(s->pre_T)); *prob = s->pre_prob; *idx = s->pre_idx; *i = s->pre_i;
}
void ora_slam_last_push_stats(const ora_slam* s, ora_push_stats* out) { *out = s->last_stats; }

static void reverse_f32(float* a, int n)
{
  for (int i = 0; i < n / 2; i++) { float t = a[i]; a[i] = a[n - 1 - i]; a[n - 1 - i] = t; }
}

void ora_slam_process_scan(ora_slam* s, const float* ranges_in, ora_scan_result* out)
{
  const ora_slam_config* cfg = &s->cfg;
  const int B = s->beams;
  memset(out, 0, sizeof(*out));
  float* ranges = (float*)malloc((size_t)B * sizeof(float));
  memcpy(ranges, ranges_in, (size_t)B * sizeof(float));
  ora_laser_min_range_clamp(ranges, B, cfg->laser_min_range);          /* ThreadLocalize.cpp:250-256 */

  if (!s->initialized) {
    /* ThreadLocalize::init (ThreadLocalize.cpp:411-511) */
    const double grid_w = (double)s->grid->N * s->grid->cs;               /* :40-41 */
    const double phi = cfg->local_offset_yaw;
    const double startX = grid_w * 0.5 + cfg->x_offset + cfg->local_offset_x;
    const double startY = grid_w * 0.5 + cfg->y_offset + cfg->local_offset_y;
    double Tinit[9] = {cos(phi), -sin(phi), startX, sin(phi), cos(phi), startY, 0, 0, 1};
    double inc = cfg->angle_increment, angle_min = cfg->angle_min;
    if (cfg->angle_increment < 0.0 && cfg->angle_min > 0) {
      s->reverse = 1; inc = -inc; angle_min = -angle_min;
      reverse_f32(ranges, B);
    }
    s->ang_res = inc; s->phi_min = angle_min;
    ora_rays_local(B, angle_min, inc, s->rays);
    memcpy(s->rays_local, s->rays, 2 * (size_t)B * sizeof(double));
    s->ray_norm = 1.0;
    double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    ora_sensor_ingest_f32(ranges, B, cfg->max_range, inc, s->data, s->mask);
    ora_rays_transform(Tinit, s->rays, B);                                /* Sensor::transform */
    ora_mat3_mul(I, Tinit, s->pose);
    double t[2] = {startX + cfg->footprint_x_offset, startY};
    ora_free_footprint(s->grid, t, cfg->footprint_width, cfg->footprint_height);
    double t0 = now_s();
    if (!s->skip_init_push) {            /* if(!_mapper.initialized()) _mapper.initPush(_sensor) (ThreadLocalize.cpp:506-507) */
      ora_push(s->grid, s->pose, s->data, s->mask, B, s->ang_res, s->phi_min, cfg->max_range,
               cfg->min_range, cfg->low_refl_range, cfg->threads, &s->last_stats);   /* initPush */
      out->pushed = 1;
    }
    out->t_push = now_s() - t0;
    s->initialized = 1;
    memcpy(out->pose, s->pose, sizeof(s->pose));
    double Id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(out->T, Id, sizeof(Id));
    free(ranges);
    return;
  }

  /* ThreadLocalize::eventLoop body (ThreadLocalize.cpp:319-406) */
  if (s->reverse) reverse_f32(ranges, B);
  ora_sensor_ingest_f32(ranges, B, cfg->max_range, s->ang_res, s->data, s->mask);
  if (!s->have_last_pose) { memcpy(s->last_pose, s->pose, sizeof(s->pose)); s->have_last_pose = 1; }

  double t0 = now_s();
  ora_rays_rescale(s->rays, B, s->grid->cs, s->ray_norm);               /* getNormalizedRayMap */
  s->ray_norm = s->grid->cs;
  out->valid_model = ora_raycast(s->grid, s->pose, s->rays, B, cfg->min_range, cfg->max_range,
                                 cfg->threads, s->model, s->normals, s->mask_m);
  out->t_raycast = now_s() - t0;
  memcpy(out->pose, s->pose, sizeof(s->pose));
  if (out->valid_model == 0) { out->no_model = 1; free(ranges); return; }
  out->valid_scene = ora_scene_from_scan(s->rays_local, s->data, s->mask, B, s->scene, s->mask_s);

  /* maskMatrix (ThreadLocalize.cpp:738-755) */
  double* Mv = (double*)malloc(2 * (size_t)B * sizeof(double));
  double* Sv = (double*)malloc(2 * (size_t)B * sizeof(double));
  int nm = 0, ns = 0;
  for (int i = 0; i < B; i++) if (s->mask_m[i]) { Mv[2 * nm] = s->model[2 * i]; Mv[2 * nm + 1] = s->model[2 * i + 1]; nm++; }
  for (int i = 0; i < B; i++) if (s->mask_s[i]) { Sv[2 * ns] = s->scene[2 * i]; Sv[2 * ns + 1] = s->scene[2 * i + 1]; ns++; }

  /* doRegistration, mode 0 (ThreadLocalize.cpp:571-581) */
  ora_icp_params ip;
  ip.iterations = cfg->icp_iterations; ip.dist_filter_max = cfg->dist_filter_max; ip.dist_filter_min = cfg->dist_filter_min;
  ip.min_x = s->grid->min_x; ip.max_x = s->grid->max_x; ip.min_y = s->grid->min_y; ip.max_y = s->grid->max_y;
  ip.nn_mode = cfg->nn_mode;
  ora_icp_result r;
  t0 = now_s();
  if (cfg->registration_mode == 3 && s->draws_sub) {
    /* doRegistration case TSD (ThreadLocalize.cpp:557-567): M, S keep the ray model (beam-indexed, masks) */
    ora_tsdpdf_match(s->grid, s->pose, s->model, s->mask_m, s->scene, s->mask_s, B, cfg->trials, cfg->size_control_set,
                     cfg->zrand, cfg->ransac_phi_max * M_PI / 180.0 /* deg2rad(_ranPhiMax) */, s->ang_res,
                     s->draws_sub, s->draws_ctrl, s->draws_trials, s->pre_T, &s->pre_prob, &s->pre_idx, &s->pre_i, NULL);
    ora_icp_init(Mv, nm, Sv, ns, s->pose, &ip, s->pre_T, &r, NULL);
  } else
  ora_icp(Mv, nm, Sv, ns, s->pose, &ip, &r, NULL);
  out->t_icp = now_s() - t0;
  free(Mv); free(Sv);
  memcpy(out->T, r.T, sizeof(r.T));
  out->rms = r.rms; out->pairs = r.pairs; out->iterations = r.iterations; out->icp_state = r.state;

  if (ora_is_registration_error(r.T, cfg->reg_trs_max, cfg->reg_sin_rot_max)) {
    out->reg_error = 1;
    free(ranges);
    return;
  }
  ora_rays_transform(r.T, s->rays, B);                                   /* _sensor->transform(&T) */
  ora_mat3_mul(s->pose, r.T, s->pose);
  memcpy(out->pose, s->pose, sizeof(s->pose));
  if (ora_is_pose_change_significant(s->last_pose, s->pose)) {
    memcpy(s->last_pose, s->pose, sizeof(s->pose));
    /* ThreadMapping::queuePush deep copy + re-mask (ThreadMapping.cpp:65-76), then TsdGrid::push */
    double* d2 = (double*)malloc((size_t)B * sizeof(double));
    uint8_t* m2 = (uint8_t*)malloc((size_t)B);
    ora_sensor_ingest_f64(s->data, B, cfg->max_range, s->ang_res, d2, m2);
    t0 = now_s();
    ora_push(s->grid, s->pose, d2, m2, B, s->ang_res, s->phi_min, cfg->max_range, cfg->min_range,
             cfg->low_refl_range, cfg->threads, &s->last_stats);
    out->t_push = now_s() - t0;
    out->pushed = 1;
    free(d2); free(m2);
  }
  free(ranges);
}
