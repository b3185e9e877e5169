/* stub_tsd_hip.c -- a RECORDING STAND-IN for the device side of include/tsd_hip.h (the symbols libohm_tsd_slam.so imports),
 * so that the C++ facade's THREAD CONTRACT can be exercised on a box without a GPU and under -fsanitize=thread / address:
 *   newest-scan-wins  (/root/reference/src/ThreadLocalize.cpp:319-332),
 *   first scan synchronous + initPush (ThreadLocalize.cpp:248-276, ThreadMapping.cpp:23-41),
 *   LIFO mapper (ThreadMapping.cpp:43-76), announceNext accept / drop, shutdown (ThreadSLAM.cpp:19-33).
 * It is NOT the oracle and computes nothing: every "registration" returns the canned motion T = translation by (0.06, 0) --
 * above the 0.05 m push gate -- and every call is appended to a log (operation, scan tag = the scan's beam-0 range, calling
 * thread) that tests/thread_contract.cpp reads back.  Per-operation delays stand for device time.  Test infrastructure only:
 * nothing in the product links it. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/tsd_hip.h"
#include "stub_tsd_hip.h"

struct tsd_ctx { int cells; double cs, max_trunc; };
struct tsd_sensor {
  tsd_ctx* ctx; int beams; double pose[9];
  int staged; double staged_tag; int submitted; double cur_tag; int async_mapping;
};
struct tsd_batch { tsd_ctx* ctx; int cap; };

static pthread_mutex_t g_mx = PTHREAD_MUTEX_INITIALIZER;
static stub_entry g_log[STUB_LOG_MAX];
static int g_n = 0;
static int g_delay_us[STUB_OP_COUNT];
static int g_live_ctx = 0, g_live_sensors = 0;

static void stub_sleep_us(int us)
{
  if (us <= 0) return;
  struct timespec ts = {us / 1000000, (long)(us % 1000000) * 1000L};
  nanosleep(&ts, NULL);
}

static void rec(int op, double tag, int flag)
{
  pthread_mutex_lock(&g_mx);
  if (g_n < STUB_LOG_MAX) {
    g_log[g_n].op = op; g_log[g_n].tag = tag; g_log[g_n].flag = flag;
    g_log[g_n].thread = (unsigned long long)(uintptr_t)pthread_self();
    g_n++;
  }
  const int d = g_delay_us[op];
  pthread_mutex_unlock(&g_mx);
  stub_sleep_us(d);
}

void stub_reset(void) { pthread_mutex_lock(&g_mx); g_n = 0; memset(g_delay_us, 0, sizeof(g_delay_us)); pthread_mutex_unlock(&g_mx); }
void stub_set_delay_us(int op, int us) { pthread_mutex_lock(&g_mx); if (op >= 0 && op < STUB_OP_COUNT) g_delay_us[op] = us; pthread_mutex_unlock(&g_mx); }
int stub_log_count(void) { pthread_mutex_lock(&g_mx); const int n = g_n; pthread_mutex_unlock(&g_mx); return n; }
stub_entry stub_log_get(int i) { stub_entry e; memset(&e, 0, sizeof(e)); pthread_mutex_lock(&g_mx); if (i >= 0 && i < g_n) e = g_log[i]; pthread_mutex_unlock(&g_mx); return e; }
int stub_live_objects(void) { pthread_mutex_lock(&g_mx); const int n = g_live_ctx + g_live_sensors; pthread_mutex_unlock(&g_mx); return n; }

/* pose <- pose * T, T = [[1,0,0.06],[0,1,0],[0,0,1]] */
static void canned_motion(double pose[9], tsd_icp_result* icp)
{
  const double T[9] = {1, 0, 0.06, 0, 1, 0, 0, 0, 1};
  double out[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) out[3 * r + c] = pose[3 * r] * T[c] + pose[3 * r + 1] * T[3 + c] + pose[3 * r + 2] * T[6 + c];
  memcpy(pose, out, sizeof(out));
  if (icp) {
    memset(icp, 0, sizeof(*icp));
    memcpy(icp->T, T, sizeof(T));
    icp->rms = 1e-4; icp->pairs = 300; icp->iterations = 30; icp->state = TSD_ICP_MAXITERATIONS; icp->n_model = 350; icp->n_scene = 340;
  }
}

tsd_ctx* tsd_create(int device, int map_size_log2, double cell_size, double max_trunc)
{
  (void)device;
  tsd_ctx* c = (tsd_ctx*)calloc(1, sizeof(*c));
  c->cells = 1 << map_size_log2; c->cs = cell_size; c->max_trunc = max_trunc;
  pthread_mutex_lock(&g_mx); g_live_ctx++; pthread_mutex_unlock(&g_mx);
  rec(STUB_CREATE, 0, 0);
  return c;
}
void tsd_destroy(tsd_ctx* ctx) { if (!ctx) return; rec(STUB_DESTROY, 0, 0); pthread_mutex_lock(&g_mx); g_live_ctx--; pthread_mutex_unlock(&g_mx); free(ctx); }
int tsd_reset(tsd_ctx* ctx) { (void)ctx; return TSD_OK; }
int tsd_set_max_truncation(tsd_ctx* ctx, double v) { ctx->max_trunc = v; return TSD_OK; }
int tsd_sync(tsd_ctx* ctx) { (void)ctx; rec(STUB_SYNC, 0, 0); return TSD_OK; }
const char* tsd_last_error(const tsd_ctx* ctx) { (void)ctx; return "stub"; }
int tsd_cells(const tsd_ctx* ctx) { return ctx->cells; }
double tsd_cell_size(const tsd_ctx* ctx) { return ctx->cs; }
double tsd_min_x(const tsd_ctx* ctx) { (void)ctx; return 0.0; }
double tsd_min_y(const tsd_ctx* ctx) { (void)ctx; return 0.0; }
double tsd_max_x(const tsd_ctx* ctx) { return (ctx->cells + 0.5) * ctx->cs; }
double tsd_max_y(const tsd_ctx* ctx) { return (ctx->cells + 0.5) * ctx->cs; }

int tsd_free_footprint(tsd_ctx* ctx, const double center[2], double width, double height)
{
  (void)ctx; (void)center; (void)width; (void)height;
  rec(STUB_FREE_FOOTPRINT, 0, 0);
  return TSD_OK;
}
int tsd_push(tsd_ctx* ctx, const double pose33[9], const double* ranges, const uint8_t* mask, int beams, double ang_res, double phi_min,
             double max_range, double min_range, double low_refl_range, tsd_push_stats* stats)
{
  (void)ctx; (void)pose33; (void)mask; (void)ang_res; (void)phi_min; (void)max_range; (void)min_range; (void)low_refl_range;
  if (stats) memset(stats, 0, sizeof(*stats));
  rec(STUB_PUSH, beams > 0 ? ranges[0] : 0.0, 0);
  return TSD_OK;
}
int tsd_raycast(tsd_ctx* ctx, const double pose33[9], const double* rays, int beams, double min_range, double max_range, double* coords,
                double* normals, uint8_t* mask, int* valid)
{
  (void)ctx; (void)pose33; (void)rays; (void)min_range; (void)max_range;
  for (int i = 0; i < beams; i++) { coords[2 * i] = 1.0; coords[2 * i + 1] = 0.0; if (normals) { normals[2 * i] = -1.0; normals[2 * i + 1] = 0.0; } mask[i] = 1; }
  if (valid) *valid = beams;
  rec(STUB_RAYCAST, 0, 0);
  return TSD_OK;
}
int tsd_icp(tsd_ctx* ctx, const double* model, int n_model, const double* scene, int n_scene, const double pose33[9], const tsd_icp_params* p,
            tsd_icp_result* result)
{
  (void)ctx; (void)model; (void)scene; (void)p;
  double pose[9]; memcpy(pose, pose33, sizeof(pose));
  canned_motion(pose, result);
  result->n_model = n_model; result->n_scene = n_scene;
  rec(STUB_ICP, 0, 0);
  return TSD_OK;
}
int tsd_localize(tsd_ctx* ctx, const double pose33[9], const double* rays_world, const double* rays_local, const double* ranges,
                 const uint8_t* mask, int beams, double min_range, double max_range, const tsd_icp_params* p, tsd_icp_result* result)
{
  (void)ctx; (void)rays_world; (void)rays_local; (void)mask; (void)min_range; (void)max_range; (void)p;
  double pose[9]; memcpy(pose, pose33, sizeof(pose));
  canned_motion(pose, result);
  rec(STUB_LOCALIZE, beams > 0 ? ranges[0] : 0.0, 0);
  return TSD_OK;
}
int tsd_tsdpdf_match(tsd_ctx* ctx, const double pose33[9], const double* m, const uint8_t* mm, const double* s, const uint8_t* ms, int beams,
                     const tsd_tsdpdf_params* prm, const int* d0, const int* d1, const int* d2, tsd_tsdpdf_result* result)
{
  (void)ctx; (void)pose33; (void)m; (void)mm; (void)s; (void)ms; (void)beams; (void)prm; (void)d0; (void)d1; (void)d2;
  memset(result, 0, sizeof(*result));
  result->T[0] = result->T[4] = result->T[8] = 1.0;
  rec(STUB_TSDPDF, 0, 0);
  return TSD_OK;
}

tsd_sensor* tsd_sensor_create(tsd_ctx* ctx, int beams, double ang_res, double phi_min, double max_range, double min_range, double low_refl)
{
  (void)ang_res; (void)phi_min; (void)max_range; (void)min_range; (void)low_refl;
  tsd_sensor* s = (tsd_sensor*)calloc(1, sizeof(*s));
  s->ctx = ctx; s->beams = beams;
  s->pose[0] = s->pose[4] = s->pose[8] = 1.0;
  pthread_mutex_lock(&g_mx); g_live_sensors++; pthread_mutex_unlock(&g_mx);
  return s;
}
void tsd_sensor_destroy(tsd_sensor* s) { if (!s) return; pthread_mutex_lock(&g_mx); g_live_sensors--; pthread_mutex_unlock(&g_mx); free(s); }
int tsd_sensor_set_pose(tsd_sensor* s, const double pose33[9], const double* rw, const double* rl)
{
  (void)rw; (void)rl;
  memcpy(s->pose, pose33, sizeof(s->pose));
  rec(STUB_SET_POSE, 0, 0);
  return TSD_OK;
}
int tsd_sensor_set_async_mapping(tsd_sensor* s, int on) { s->async_mapping = on; return TSD_OK; }

static void fill_scan_result(tsd_sensor* s, tsd_scan_result* r)
{
  memset(r, 0, sizeof(*r));
  canned_motion(s->pose, &r->icp);
  memcpy(r->pose, s->pose, sizeof(s->pose));
  r->pushed = 1;
}
int tsd_scan_stage(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push)
{
  (void)mask; (void)mask_push;
  if (s->staged) return TSD_E_ARG;
  s->staged = 1; s->staged_tag = ranges[0];
  rec(STUB_SCAN_STAGE, ranges[0], 0);
  return TSD_OK;
}
int tsd_scan_submit(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push, const tsd_icp_params* p,
                    const tsd_gate_params* g)
{
  (void)mask; (void)mask_push; (void)p; (void)g;
  if (s->submitted) return TSD_E_ARG;
  if (!ranges && !s->staged) return TSD_E_ARG;
  /* flag: 1 = started from the staged scan, 2 = a staged scan was dropped for the one that came, 0 = plain */
  const int flag = !ranges ? 1 : (s->staged ? 2 : 0);
  s->cur_tag = ranges ? ranges[0] : s->staged_tag;
  s->staged = 0; s->submitted = 1;
  rec(STUB_SCAN_SUBMIT, s->cur_tag, flag);
  return TSD_OK;
}
int tsd_scan_collect(tsd_sensor* s, tsd_scan_result* r)
{
  if (!s->submitted) return TSD_E_ARG;
  s->submitted = 0;
  rec(STUB_SCAN_COLLECT, s->cur_tag, 0);      /* (the delay of this operation stands for the registration's device time) */
  fill_scan_result(s, r);
  return TSD_OK;
}
int tsd_scan(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push, const tsd_icp_params* p,
             const tsd_gate_params* g, tsd_scan_result* r)
{
  const int rc = tsd_scan_submit(s, ranges, mask, mask_push, p, g);
  return rc != TSD_OK ? rc : tsd_scan_collect(s, r);
}
int tsd_scan_preregister(tsd_sensor* s, const tsd_tsdpdf_params* prm, const double* scene, const uint8_t* ms, const int* d0, const int* d1,
                         const int* d2)
{
  (void)s; (void)prm; (void)scene; (void)ms; (void)d0; (void)d1; (void)d2;
  rec(STUB_PREREGISTER, 0, 0);
  return TSD_OK;
}
int tsd_scan_begin(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* mask_push, const tsd_icp_params* p,
                   const tsd_gate_params* g)
{
  (void)mask; (void)mask_push; (void)p; (void)g;
  s->cur_tag = ranges[0]; s->submitted = 1;
  rec(STUB_SCAN_BEGIN, ranges[0], 0);
  return TSD_OK;
}
int tsd_scan_finish(tsd_sensor* s, tsd_scan_result* r)
{
  s->submitted = 0;
  rec(STUB_SCAN_FINISH, s->cur_tag, 0);
  fill_scan_result(s, r);
  return TSD_OK;
}

/* (the batched multi-robot dispatcher is not part of the single-robot thread contract: present so that the library links) */
tsd_batch* tsd_batch_create(tsd_ctx* ctx, int max_scans) { tsd_batch* b = (tsd_batch*)calloc(1, sizeof(*b)); b->ctx = ctx; b->cap = max_scans; return b; }
void tsd_batch_destroy(tsd_batch* b) { free(b); }
int tsd_batch_begin(tsd_batch* b, int n, tsd_sensor* const* sensors, const double* const* ranges, const uint8_t* const* mask,
                    const uint8_t* const* mask_push, const tsd_icp_params* params, const tsd_gate_params* gates)
{
  (void)b; (void)n; (void)sensors; (void)ranges; (void)mask; (void)mask_push; (void)params; (void)gates;
  return TSD_E_ARG;
}
int tsd_batch_push(tsd_batch* b) { (void)b; return TSD_E_ARG; }
int tsd_batch_poll(tsd_batch* b) { (void)b; return 1; }
int tsd_batch_results(tsd_batch* b, tsd_scan_result* results) { (void)b; (void)results; return TSD_E_ARG; }
