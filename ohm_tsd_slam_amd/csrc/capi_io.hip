// capi_io.hip -- the entry points of include/tsd_hip.h that move the grid in and out of the device and measure it: canonical tile
// I/O (tsd_download_tiles / tsd_upload_tiles / tsd_grid_digest), the reference's text grid files, the occupancy map and the colour image
// (row N1), the calibration / stream measurements of bench.py's roofline, the per-kernel HIP-event profile.
#include "capi_internal.hpp"

using namespace tsd;

extern "C" {
int tsd_download_tile_state(tsd_ctx* ctx, uint8_t* initialized, double* init_weight)
{
  if (!ctx || !initialized || !init_weight) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  const size_t T = (size_t)ctx->grid.tiles;
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(initialized, ctx->grid.flags, T, hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(init_weight, ctx->grid.init_weight, T * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return TSD_OK;
}

// tiles per chunk of the canonical tile I/O: 2 x 4096 x 1089 doubles = 71 MB of device staging
static constexpr int kIoChunk = 4096;

int tsd_download_tiles(tsd_ctx* ctx, uint8_t* initialized, double* init_weight, double* tsd_out,
                       double* weight_out)
{
  if (!ctx || !initialized || !init_weight || !tsd_out || !weight_out) return TSD_E_ARG;
  int rc = tsd_download_tile_state(ctx, initialized, init_weight);
  if (rc != TSD_OK) return rc;
  const GridDev& g = ctx->grid;
  const double qnan = std::nan("");
  const int chunk = g.tiles < kIoChunk ? g.tiles : kIoChunk;
  double* d_t = nullptr; double* d_w = nullptr;
  const size_t cb = (size_t)chunk * TSD_TILE_CELLS * sizeof(double);
  TSD_HIP_CHECK(ctx, hipMalloc(&d_t, cb));
  hipError_t e = hipMalloc(&d_w, cb);
  if (e != hipSuccess) { hipFree(d_t); return set_error(ctx, TSD_E_HIP, "tsd_download_tiles staging", e); }
  for (int t0 = 0; t0 < g.tiles && rc == TSD_OK; t0 += chunk) {
    const int n = g.tiles - t0 < chunk ? g.tiles - t0 : chunk;
    bool any = false;
    for (int p = t0; p < t0 + n; p++) any |= initialized[p] != 0;
    double* t = tsd_out + (size_t)t0 * TSD_TILE_CELLS;
    double* w = weight_out + (size_t)t0 * TSD_TILE_CELLS;
    if (!any) {                                    // most of a big grid: nothing to fetch
      for (size_t i = 0; i < (size_t)n * TSD_TILE_CELLS; i++) { t[i] = qnan; w[i] = 0.0; }
      continue;
    }
    rc = launch_export_tiles(ctx, t0, n, d_t, d_w);
    if (rc != TSD_OK) break;
    e = hipMemcpyAsync(t, d_t, (size_t)n * TSD_TILE_CELLS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(w, d_w, (size_t)n * TSD_TILE_CELLS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = set_error(ctx, TSD_E_HIP, "tsd_download_tiles copy", e);
  }
  hipFree(d_t); hipFree(d_w);
  return rc;
}

int tsd_upload_tiles(tsd_ctx* ctx, const uint8_t* initialized, const double* init_weight,
                     const double* tsd_in, const double* weight_in)
{
  if (ctx) ctx->epoch++;                  // (invalidates ray casts enqueued ahead of their scan)
  if (!ctx || !initialized || !init_weight || !tsd_in || !weight_in) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  std::lock_guard<std::mutex> lk_order(ctx->order_mutex);
  if (int rcw = wait_for_readers(ctx)) return rcw;
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const GridDev& g = ctx->grid;
  const size_t T = (size_t)g.tiles;
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(g.flags, initialized, T, hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(g.init_weight, init_weight, T * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const int chunk = g.tiles < kIoChunk ? g.tiles : kIoChunk;
  double* d_t = nullptr; double* d_w = nullptr;
  const size_t cb = (size_t)chunk * TSD_TILE_CELLS * sizeof(double);
  TSD_HIP_CHECK(ctx, hipMalloc(&d_t, cb));
  hipError_t e = hipMalloc(&d_w, cb);
  if (e != hipSuccess) { hipFree(d_t); return set_error(ctx, TSD_E_HIP, "tsd_upload_tiles staging", e); }
  int rc = TSD_OK;
  for (int t0 = 0; t0 < g.tiles && rc == TSD_OK; t0 += chunk) {
    const int n = g.tiles - t0 < chunk ? g.tiles - t0 : chunk;
    bool any = false;
    for (int p = t0; p < t0 + n; p++) any |= initialized[p] != 0;
    if (!any) continue;
    e = hipMemcpyAsync(d_t, tsd_in + (size_t)t0 * TSD_TILE_CELLS, (size_t)n * TSD_TILE_CELLS * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_w, weight_in + (size_t)t0 * TSD_TILE_CELLS, (size_t)n * TSD_TILE_CELLS * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { rc = set_error(ctx, TSD_E_HIP, "tsd_upload_tiles copy", e); break; }
    rc = launch_import_tiles(ctx, t0, n, d_t, d_w);
    if (rc == TSD_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_error(ctx, TSD_E_HIP, "tsd_upload_tiles sync", hipGetLastError());
  }
  hipFree(d_t); hipFree(d_w);
  if (rc != TSD_OK) return rc;
  rc = launch_neg_scan(ctx);              // which tiles can show a sign change to the ray cast
  if (rc != TSD_OK) return rc;
  // The halos came as they were given (the text format does not store them: NaN): nothing says they agree with the neighbours' edge
  // cells, which the incremental propagateBorders of the push relies on for the tiles it does not touch.  Every tile that holds data is
  // marked like a freeFootprint write: the next push refreshes the halos around all of them -- the reference's full sweep after its
  // first push (TsdGrid.cpp:372-427).
  TSD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_dirty, initialized, T, hipMemcpyHostToDevice, ctx->stream));
  {
    TileBox all; all.x0 = 0; all.y0 = 0; all.x1 = g.PX - 1; all.y1 = g.PX - 1;
    ctx->box_dirty.add(all);
  }
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return TSD_OK;
}

int tsd_grid_digest(tsd_ctx* ctx, tsd_grid_digest_t* out)
{
  if (!ctx || !out) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  const size_t T = (size_t)ctx->grid.tiles;
  unsigned long long* d_o = nullptr; double* d_s = nullptr;
  TSD_HIP_CHECK(ctx, hipMalloc(&d_o, T * 2 * sizeof(unsigned long long)));
  hipError_t e = hipMalloc(&d_s, T * 2 * sizeof(double));
  if (e != hipSuccess) { hipFree(d_o); return set_error(ctx, TSD_E_HIP, "tsd_grid_digest", e); }
  int rc = launch_grid_digest(ctx, d_o, d_s);
  std::vector<unsigned long long> ho(T * 2); std::vector<double> hs(T * 2); std::vector<uint8_t> fl(T);
  if (rc == TSD_OK) {
    e = hipMemcpyAsync(ho.data(), d_o, T * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hs.data(), d_s, T * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(fl.data(), ctx->grid.flags, T, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = set_error(ctx, TSD_E_HIP, "tsd_grid_digest copy", e);
  }
  hipFree(d_o); hipFree(d_s);
  if (rc != TSD_OK) return rc;
  out->hash = 0; out->cells_valid = 0; out->tiles_initialized = 0; out->sum_tsd = 0.0; out->sum_weight = 0.0;
  for (size_t p = 0; p < T; p++) {          // tile order: the sums are reproducible
    out->hash += ho[2 * p]; out->cells_valid += (int64_t)ho[2 * p + 1];
    out->sum_tsd += hs[2 * p]; out->sum_weight += hs[2 * p + 1];
    out->tiles_initialized += fl[p] ? 1 : 0;
  }
  return TSD_OK;
}

int tsd_storage_bits(void) { return (int)(8 * sizeof(tsd_cell_t)); }

int tsd_abi_sizeof(const char* n)
{
  if (!n) return 0;
#define TSD_SZ(T) if (std::strcmp(n, #T) == 0) return (int)sizeof(T)
  TSD_SZ(tsd_push_stats); TSD_SZ(tsd_icp_params); TSD_SZ(tsd_icp_result); TSD_SZ(tsd_gate_params); TSD_SZ(tsd_scan_result);
  TSD_SZ(tsd_grid_digest_t); TSD_SZ(tsd_tsdpdf_params); TSD_SZ(tsd_tsdpdf_result);
#undef TSD_SZ
  return 0;
}

int tsd_store_grid_text(tsd_ctx* ctx, const char* path)
{
  if (!ctx || !path || !path[0]) return TSD_E_ARG;
  const GridDev& g = ctx->grid;
  const size_t T = (size_t)g.tiles;
  std::vector<uint8_t> init(T);
  std::vector<double> iw(T), tsd(T * TSD_TILE_CELLS), w(T * TSD_TILE_CELLS);
  int rc = tsd_download_tiles(ctx, init.data(), iw.data(), tsd.data(), w.data());
  if (rc != TSD_OK) return rc;
  std::FILE* f = std::fopen(path, "w");
  if (!f) return set_error(ctx, TSD_E_ARG, "tsd_store_grid_text: cannot open the file", hipSuccess);
  int map_log2 = 0;
  while ((1 << map_log2) < g.N) map_log2++;
  // "%g" is the default ostream format of the reference's `outFile << value`
  std::fprintf(f, "%g\n%d\n%d\n%g\n", g.cs, 5 /* LAYOUT_32x32 */, map_log2, g.max_trunc);
  for (size_t p = 0; p < T; p++) {
    if (init[p]) {
      std::fprintf(f, "2\n");
      for (int py = 0; py < TILE_DIM; py++)
        for (int px = 0; px < TILE_DIM; px++) {
          const size_t i = p * TSD_TILE_CELLS + (size_t)(py * TILE_PITCH + px);
          std::fprintf(f, "%g\n%g\n", tsd[i], w[i]);
        }
    } else if (iw[p] > 0.0) {            // isEmpty()
      std::fprintf(f, "1\n%g\n", iw[p]);
    } else {
      std::fprintf(f, "0\n");
    }
  }
  std::fclose(f);
  return TSD_OK;
}

// getDoubleLine / getIntLine (obcore/base/tools.cpp:190-215)
static double text_double_line(std::FILE* f)
{
  char line[1024];
  if (!std::fgets(line, sizeof(line), f) || line[0] == '\n' || line[0] == 0) return std::nan("");
  return std::strtod(line, nullptr);
}
static int text_int_line(std::FILE* f)
{
  char line[1024];
  if (!std::fgets(line, sizeof(line), f) || line[0] == '\n' || line[0] == 0) return 0;
  return std::atoi(line);
}

int tsd_load_grid_text(tsd_ctx* ctx, const char* path)
{
  if (!ctx || !path || !path[0]) return TSD_E_ARG;
  std::FILE* f = std::fopen(path, "r");
  if (!f) return set_error(ctx, TSD_E_ARG, "tsd_load_grid_text: cannot open the file", hipSuccess);
  const GridDev& g = ctx->grid;
  const double cell_size = text_double_line(f);
  const int layout_part = text_int_line(f), layout_grid = text_int_line(f);
  const double max_trunc = text_double_line(f);
  int map_log2 = 0;
  while ((1 << map_log2) < g.N) map_log2++;
  if (layout_part != 5 || layout_grid != map_log2 || !(std::fabs(cell_size - g.cs) <= 1e-5 * g.cs)) {
    std::fclose(f);
    return set_error(ctx, TSD_E_ARG, "tsd_load_grid_text: the file's layout / cell size is not this grid's", hipSuccess);
  }
  const size_t T = (size_t)g.tiles;
  std::vector<uint8_t> init(T, 0);
  std::vector<double> iw(T, 0.0), tsd(T * TSD_TILE_CELLS), w(T * TSD_TILE_CELLS, 0.0);
  for (size_t p = 0; p < T; p++) {
    const int id = text_int_line(f);
    if (id == 0) continue;
    if (id == 1) { iw[p] = std::fmin(text_double_line(f), 32.0 /* TSDGRIDMAXWEIGHT */); continue; }
    if (id != 2) { std::fclose(f); return set_error(ctx, TSD_E_ARG, "tsd_load_grid_text: unknown tile identifier", hipSuccess); }
    // curPart->init(maxTruncation) on a fresh partition (_initWeight 0): every cell NaN / 0, halo included; then the
    // interior cells from the file
    init[p] = 1;
    for (int i = 0; i < TSD_TILE_CELLS; i++) tsd[p * TSD_TILE_CELLS + (size_t)i] = std::nan("");
    for (int py = 0; py < TILE_DIM; py++)
      for (int px = 0; px < TILE_DIM; px++) {
        const size_t i = p * TSD_TILE_CELLS + (size_t)(py * TILE_PITCH + px);
        tsd[i] = text_double_line(f);
        w[i] = text_double_line(f);
      }
  }
  std::fclose(f);
  int rc = tsd_reset(ctx);
  if (rc != TSD_OK) return rc;
  rc = tsd_set_max_truncation(ctx, max_trunc);
  if (rc != TSD_OK) return rc;
  return tsd_upload_tiles(ctx, init.data(), iw.data(), tsd.data(), w.data());
}

int tsd_occupancy_dev_async(tsd_ctx* ctx, void* occ_dev, int inflate, int inflate_factor)
{
  if (!ctx || !occ_dev) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  return launch_occupancy(ctx, static_cast<int8_t*>(occ_dev), inflate, inflate_factor);
}

int tsd_occupancy_dev(tsd_ctx* ctx, void* occ_dev, int inflate, int inflate_factor)
{
  int rc = tsd_occupancy_dev_async(ctx, occ_dev, inflate, inflate_factor);
  if (rc != TSD_OK) return rc;
  TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return TSD_OK;
}

int tsd_color_image(tsd_ctx* ctx, uint8_t* rgb_host, unsigned int width, unsigned int height)
{
  if (!ctx || !rgb_host || width == 0 || height == 0) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  // px / py exactly as the reference accumulates them (TsdGrid.cpp:433-486): start at 0, += step per pixel
  std::vector<double> pq((size_t)width + height);
  const double stepW = ctx->grid.max_x / (double)width, stepH = ctx->grid.max_y / (double)height;
  { double v = 0.0; for (unsigned w = 0; w < width; w++) { pq[w] = v; v += stepW; } }
  { double v = 0.0; for (unsigned h = 0; h < height; h++) { pq[(size_t)width + h] = v; v += stepH; } }
  // (device staging kept by the context and grown on demand: ThreadGrid publishes map and image every occ_grid_time_interval
  // beside the localisers, ThreadGrid.cpp:72-133 -- a hipMalloc / hipFree pair per call would stall every stream of the device)
  const size_t img_bytes = (size_t)3 * width * height, pq_bytes = pq.size() * sizeof(double);
  if (ctx->img_bytes < img_bytes + pq_bytes + 64) {
    TSD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_img) hipFree(ctx->d_img);
    ctx->d_img = nullptr; ctx->img_bytes = 0;
    TSD_HIP_CHECK(ctx, hipMalloc(&ctx->d_img, img_bytes + pq_bytes + 64));
    ctx->img_bytes = img_bytes + pq_bytes + 64;
  }
  double* d_pq = reinterpret_cast<double*>(ctx->d_img);
  uint8_t* d_img = ctx->d_img + ((pq_bytes + 63) & ~(size_t)63);
  int rc = TSD_OK;
  hipError_t e = hipMemcpyAsync(d_pq, pq.data(), pq_bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) rc = launch_color_image(ctx, d_pq, d_pq + width, width, height, d_img);
  if (e == hipSuccess && rc == TSD_OK) e = hipMemcpyAsync(rgb_host, d_img, img_bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);       // (also when the launch failed: `pq` must outlive its copy)
  if (e != hipSuccess) return set_error(ctx, TSD_E_HIP, "tsd_color_image", e);
  return rc;
}

int tsd_occupancy(tsd_ctx* ctx, int8_t* occ_host, int inflate, int inflate_factor, int* n_surface)
{
  if (!ctx || !occ_host) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  const size_t cells = (size_t)ctx->grid.N * ctx->grid.N;
  if (!ctx->d_occ_out) TSD_HIP_CHECK(ctx, hipMalloc(&ctx->d_occ_out, cells));      // once per context (see tsd_color_image)
  // The map leaves on a stream of its own behind the extraction kernels' event: 16 MiB at 4096^2 are ~0.7 ms of PCIe, and on the grid's
  // stream every push and ray cast enqueued meanwhile (the localisers keep running: ThreadGrid.cpp:72-133 extracts beside them) would
  // sit behind the copy; the caller waits for the copy stream only.
  if (!ctx->stream_io) {
    TSD_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->stream_io, hipStreamNonBlocking));
    TSD_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_io, hipEventDisableTiming));
  }
  int rc = launch_occupancy(ctx, ctx->d_occ_out, inflate, inflate_factor);
  if (rc == TSD_OK) {
    int n = 0;
    hipError_t e = hipEventRecord(ctx->ev_io, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream_io, ctx->ev_io, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(occ_host, ctx->d_occ_out, cells, hipMemcpyDeviceToHost, ctx->stream_io);
    if (e == hipSuccess) e = hipMemcpyAsync(&n, ctx->d_occ_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream_io);
    const hipError_t es = hipStreamSynchronize(ctx->stream_io);
    if (e == hipSuccess) e = es;
    if (n_surface) *n_surface = n;
    if (e != hipSuccess) rc = set_error(ctx, TSD_E_HIP, "tsd_occupancy copy", e);
  }
  return rc;
}

int tsd_calibrate_rmw(tsd_ctx* ctx, int64_t n_doubles, int reps)
{
  if (!ctx || n_doubles <= 0 || reps <= 0) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  double *t = nullptr, *w = nullptr;
  TSD_HIP_CHECK(ctx, hipMalloc(&t, (size_t)n_doubles * sizeof(double)));
  hipError_t e = hipMalloc(&w, (size_t)n_doubles * sizeof(double));
  if (e != hipSuccess) { hipFree(t); return set_error(ctx, TSD_E_HIP, "tsd_calibrate_rmw", e); }
  hipMemsetAsync(t, 0, (size_t)n_doubles * sizeof(double), ctx->stream);
  hipMemsetAsync(w, 0, (size_t)n_doubles * sizeof(double), ctx->stream);
  int rc = TSD_OK;
  for (int r = 0; r < reps && rc == TSD_OK; r++) rc = launch_calibrate(ctx, t, w, (size_t)n_doubles);
  hipStreamSynchronize(ctx->stream);
  hipFree(t); hipFree(w);
  return rc;
}

int tsd_measure_stream(tsd_ctx* ctx, int64_t n_doubles, int reps, double* gbs_best, double* gbs_mean)
{
  if (!ctx || n_doubles <= 0 || reps <= 0 || reps > 64) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  double *t = nullptr, *w = nullptr;
  const size_t bytes = (size_t)n_doubles * sizeof(double);
  TSD_HIP_CHECK(ctx, hipMalloc(&t, bytes));
  hipError_t e = hipMalloc(&w, bytes);
  if (e != hipSuccess) { hipFree(t); return set_error(ctx, TSD_E_HIP, "tsd_measure_stream", e); }
  hipMemsetAsync(t, 0, bytes, ctx->stream);
  hipMemsetAsync(w, 0, bytes, ctx->stream);
  int rc = launch_calibrate(ctx, t, w, (size_t)n_doubles);      // untimed: first touch
  std::vector<hipEvent_t> ev((size_t)reps + 1, nullptr);
  for (auto& x : ev) if (hipEventCreate(&x) != hipSuccess) rc = set_error(ctx, TSD_E_HIP, "tsd_measure_stream: events", hipGetLastError());
  if (rc == TSD_OK) hipEventRecord(ev[0], ctx->stream);
  for (int r = 0; r < reps && rc == TSD_OK; r++) { rc = launch_calibrate(ctx, t, w, (size_t)n_doubles); hipEventRecord(ev[(size_t)r + 1], ctx->stream); }
  hipStreamSynchronize(ctx->stream);
  double best = 0.0, sum = 0.0; int n = 0;
  for (int r = 0; r < reps && rc == TSD_OK; r++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev[(size_t)r], ev[(size_t)r + 1]) != hipSuccess || !(ms > 0.f)) continue;
    const double gbs = 4.0 * (double)bytes / ((double)ms * 1e-3) / 1e9;     // two arrays, each read and written once
    if (gbs > best) best = gbs;
    sum += gbs; n++;
  }
  for (auto x : ev) if (x) hipEventDestroy(x);
  hipFree(t); hipFree(w);
  if (rc != TSD_OK) return rc;
  if (!n) return set_error(ctx, TSD_E_HIP, "tsd_measure_stream: no timed launch", hipSuccess);
  if (gbs_best) *gbs_best = best;
  if (gbs_mean) *gbs_mean = sum / n;
  return TSD_OK;
}

int tsd_profile_enable(tsd_ctx* ctx, int on)
{
  if (!ctx) return TSD_E_ARG;
  ctx->profile = on != 0;
  if (ctx->profile && ctx->profile_mask == 0) ctx->profile_mask = ~0u;
  return TSD_OK;
}

int tsd_profile_select(tsd_ctx* ctx, const char* kernels_csv)
{
  if (!ctx || !kernels_csv) return TSD_E_ARG;
  // "a,b,c/n": the listed kernels (or "all"), every n-th launch of each; a name may carry its own period, "a:m", which
  // wins over the list's (bench.py times EVERY dispatch of the roofline kernel in a short run and every n-th of the others)
  unsigned mask = 0;
  std::string csv(kernels_csv);
  ctx->profile_every = 1;
  const size_t slash = csv.rfind('/');
  if (slash != std::string::npos) {
    const int n = std::atoi(csv.c_str() + slash + 1);
    ctx->profile_every = n > 1 ? (unsigned)n : 1u;
    csv = csv.substr(0, slash);
  }
  constexpr unsigned NK = sizeof(kKernelNames) / sizeof(kKernelNames[0]);
  for (unsigned i = 0; i < NK; i++) ctx->profile_every_k[i] = 0;
  size_t pos = 0;
  while (pos <= csv.size()) {
    size_t end = csv.find(',', pos);
    if (end == std::string::npos) end = csv.size();
    std::string tok = csv.substr(pos, end - pos);
    pos = end + 1;
    if (tok.empty()) continue;
    unsigned own = 0;
    const size_t colon = tok.find(':');
    if (colon != std::string::npos) { const int m = std::atoi(tok.c_str() + colon + 1); own = m >= 1 ? (unsigned)m : 1u; tok = tok.substr(0, colon); }
    for (unsigned i = 0; i < NK; i++)
      if (tok == "all" || tok == kKernelNames[i]) { mask |= 1u << i; if (own && tok != "all") ctx->profile_every_k[i] = own; }
  }
  ctx->profile_mask = mask;
  return TSD_OK;
}

int tsd_profile_reset(tsd_ctx* ctx)
{
  if (!ctx) return TSD_E_ARG;
  hipStreamSynchronize(ctx->stream);
  drain_timers(ctx);
  ctx->timers.clear();
  return TSD_OK;
}

int tsd_profile_get(tsd_ctx* ctx, const char* kernel, double* total_ms, int* launches)
{
  if (!ctx || !kernel) return TSD_E_ARG;
  hipStreamSynchronize(ctx->stream);
  drain_timers(ctx);
  auto it = ctx->timers.find(kernel);
  if (total_ms) *total_ms = (it == ctx->timers.end()) ? 0.0 : it->second.total_ms;
  if (launches) *launches = (it == ctx->timers.end()) ? 0 : it->second.launches;
  return TSD_OK;
}

int tsd_profile_get_spread(tsd_ctx* ctx, const char* kernel, double* min_ms, double* max_ms, double* std_ms)
{
  if (!ctx || !kernel) return TSD_E_ARG;
  hipStreamSynchronize(ctx->stream);
  drain_timers(ctx);
  std::lock_guard<std::mutex> lk(ctx->misc_mutex);
  auto it = ctx->timers.find(kernel);
  const bool have = it != ctx->timers.end() && it->second.launches > 0;
  const double n = have ? (double)it->second.launches : 1.0;
  const double mean = have ? it->second.total_ms / n : 0.0;
  double var = have ? it->second.sum_sq / n - mean * mean : 0.0;
  if (var < 0.0) var = 0.0;
  if (min_ms) *min_ms = have ? it->second.min_ms : 0.0;
  if (max_ms) *max_ms = have ? it->second.max_ms : 0.0;
  if (std_ms) *std_ms = std::sqrt(var);
  return TSD_OK;
}

int tsd_profile_get_samples(tsd_ctx* ctx, const char* kernel, float* ms_out, int cap)
{
  if (!ctx || !kernel || cap < 0 || (cap > 0 && !ms_out)) return TSD_E_ARG;
  hipStreamSynchronize(ctx->stream);
  drain_timers(ctx);
  std::lock_guard<std::mutex> lk(ctx->misc_mutex);
  auto it = ctx->timers.find(kernel);
  if (it == ctx->timers.end()) return 0;
  const int n = (int)std::min(it->second.samples.size(), (size_t)cap);
  for (int i = 0; i < n; i++) ms_out[i] = it->second.samples[(size_t)i];
  return (int)it->second.samples.size();
}

int tsd_push_stats_total(tsd_ctx* ctx, tsd_push_stats* total, int64_t* pushes, int reset)
{
  if (!ctx) return TSD_E_ARG;
  TSD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rcd_ = drain_async_push(ctx)) return rcd_;
  return read_total_stats(ctx, total, pushes, reset != 0);
}

}  // extern "C"

