// obvious.cpp -- host-side sensor model and the device-grid handle (see obvious.h).
#include "obvious.h"
#include <cstdlib>
#include <ctime>

#include <algorithm>
#include <chrono>
#include <cstdio>

namespace obvious {

// ---------------------------------------------------------------------------------------- Matrix
Matrix::Matrix(unsigned rows, unsigned cols) : _rows(rows), _cols(cols)
{
  std::memset(_d, 0, sizeof(_d));
}
Matrix::Matrix(unsigned rows, unsigned cols, const double* data) : _rows(rows), _cols(cols)
{
  std::memset(_d, 0, sizeof(_d));
  std::memcpy(_d, data, sizeof(double) * rows * cols);
}
void Matrix::setIdentity()
{
  for (unsigned r = 0; r < _rows; r++)
    for (unsigned c = 0; c < _cols; c++) (*this)(r, c) = (r == c) ? 1.0 : 0.0;
}
void Matrix::setData(const double* array) { std::memcpy(_d, array, sizeof(double) * _rows * _cols); }
void Matrix::getData(double* array) const { std::memcpy(array, _d, sizeof(double) * _rows * _cols); }

// gsl_blas_dgemm(NoTrans, NoTrans, 1.0, A, B, 0.0, C): C[i][j] = sum_k A[i][k] B[k][j], k ascending
Matrix Matrix::operator*(const Matrix& M) const
{
  Matrix R(_rows, M._cols);
  for (unsigned i = 0; i < _rows; i++)
    for (unsigned j = 0; j < M._cols; j++) {
      double t = 0.0;
      for (unsigned k = 0; k < _cols; k++) t += (*this)(i, k) * M(k, j);
      R(i, j) = t;
    }
  return R;
}

// gsl_linalg_LU_decomp + gsl_linalg_LU_invert (gsl/Matrix.cpp:168-179): partial pivoting, inverse by
// solving against the identity columns.  3x3 only (poses).
void Matrix::invert()
{
  const unsigned n = _rows;
  double lu[16];
  int perm[4] = {0, 1, 2, 3};
  for (unsigned i = 0; i < n; i++)
    for (unsigned j = 0; j < n; j++) lu[n * i + j] = (*this)(i, j);
  for (unsigned j = 0; j < n; j++) {
    unsigned piv = j;
    double best = std::fabs(lu[n * j + j]);
    for (unsigned i = j + 1; i < n; i++)
      if (std::fabs(lu[n * i + j]) > best) { best = std::fabs(lu[n * i + j]); piv = i; }
    if (piv != j) {
      for (unsigned k = 0; k < n; k++) std::swap(lu[n * j + k], lu[n * piv + k]);
      std::swap(perm[j], perm[piv]);
    }
    for (unsigned i = j + 1; i < n; i++) {
      lu[n * i + j] = lu[n * i + j] / lu[n * j + j];
      for (unsigned k = j + 1; k < n; k++) lu[n * i + k] -= lu[n * i + j] * lu[n * j + k];
    }
  }
  for (unsigned c = 0; c < n; c++) {
    double x[4];
    for (unsigned i = 0; i < n; i++) x[i] = (perm[i] == (int)c) ? 1.0 : 0.0;
    for (unsigned i = 1; i < n; i++)
      for (unsigned k = 0; k < i; k++) x[i] -= lu[n * i + k] * x[k];
    for (int i = (int)n - 1; i >= 0; i--) {
      for (unsigned k = (unsigned)i + 1; k < n; k++) x[i] -= lu[n * i + k] * x[k];
      x[i] = x[i] / lu[n * i + i];
    }
    for (unsigned i = 0; i < n; i++) (*this)(i, c) = x[i];
  }
}

// --------------------------------------------------------------------------------- SensorPolar2D
SensorPolar2D::SensorPolar2D(unsigned int size, double angularRes, double phiMin, double maxRange,
                             double minRange, double lowReflectivityRange)
    : _size(size), _angularRes(angularRes), _phiMin(phiMin), _maxRange(maxRange), _minRange(minRange),
      _lowReflectivityRange(lowReflectivityRange), _rayNorm(1.0), _T(3, 3), _data(size, 0.0),
      _mask(size, 1), _rays(2 * size), _raysLocal(2 * size)
{
  _T.setIdentity();
  _phiLowerBound = -0.5 * _angularRes + _phiMin;                       // SensorPolar2D.cpp:26
  _phiUpperBound = _phiMin + (((double)size) - 0.5) * _angularRes;    // :30
  for (unsigned int i = 0; i < _size; i++) {                          // :39-44
    const double phi = _phiMin + ((double)i) * _angularRes;
    _rays[i] = std::cos(phi);
    _rays[_size + i] = std::sin(phi);
  }
  _raysLocal = _rays;
}

void SensorPolar2D::setRealMeasurementData(const std::vector<float>& data, float scale)
{
  const size_t n = std::min<size_t>(data.size(), _size);
  for (size_t i = 0; i < n; i++) _data[i] = (double)(data[i] * scale);   // float multiply, then widen
}

void SensorPolar2D::setRealMeasurementData(const double* data, double scale)
{
  if (scale == 1.0) std::memcpy(_data.data(), data, _size * sizeof(*data));
  else
    for (unsigned int i = 0; i < _size; i++) _data[i] = data[i] * scale;
}

void SensorPolar2D::setStandardMask()
{
  resetMask();
  maskZeroDepth();
  maskInvalidDepth();
  maskDepthDiscontinuity(3.0 * M_PI / 180.0);   // deg2rad(3.0), mathbase.h:175-178
}

void SensorPolar2D::resetMask() { std::fill(_mask.begin(), _mask.end(), 1); }

void SensorPolar2D::maskZeroDepth()
{
  for (unsigned int i = 0; i < _size; i++) _mask[i] = _mask[i] && (_data[i] != 0.0);
}

void SensorPolar2D::maskInvalidDepth()
{
  _nanBeams.clear();
  for (unsigned int i = 0; i < _size; i++) {
    if (_data[i] > _maxRange) _data[i] = INFINITY;
    if (std::isnan(_data[i])) { _mask[i] = 0; _data[i] = INFINITY; _nanBeams.push_back(i); }
  }
}

// The mask ThreadMapping::queuePush's copy ends up with (ThreadMapping.cpp:65-76): setStandardMask() is
// run again on the PROCESSED ranges.  Zero and over-range readings and every depth discontinuity come
// out as before (same data); only a former NaN reading, +inf by now, is no longer masked (Appendix B #20).
// copyForMapping() does the two passes literally; this is its result without the second pass.
void SensorPolar2D::maskForMapping(std::vector<uint8_t>& out) const
{
  out = _mask;
  for (unsigned int i : _nanBeams) out[i] = 1;
}

void SensorPolar2D::maskDepthDiscontinuity(double thresh)
{
  // Reference loop (SensorPolar2D.cpp:67-98): betamin = min over the neighbours j in {-1,0,+1} with a > b
  // of asin(b / c * sin(res)), c = sqrt(a^2 + b^2 - 2ab cos(res)); mask if betamin < thresh.  Only j = -1
  // and j = +1 can have a > b, and min(asin) < thresh <=> some asin(v) < thresh.  asin is monotone, so v is
  // first compared with sin(thresh) widened by 1e-12: outside that band the libm call cannot change the
  // decision and is skipped (this loop is the whole per-scan host cost of the ingest).
  const int radius = 1;
  const double cosphi = std::cos(_angularRes), sinphi = std::sin(_angularRes);
  const double sv = std::sin(thresh), svLo = sv * (1.0 - 1e-12), svHi = sv * (1.0 + 1e-12);
  const bool fastOk = thresh > 0.0 && thresh < 1.5;
  for (int i = radius; i < ((int)_size) - radius; i++) {
    const double a = _data[i];
    if (std::isinf(a)) continue;
    bool hit = false;
    for (int j = -radius; j <= radius && !hit; j += 2) {
      const double b = _data[i + j];
      if (std::isinf(b) || !(a > b)) continue;
      const double c = std::sqrt(a * a + b * b - 2 * a * b * cosphi);   // law of cosines
      const double v = b / c * sinphi;                                   // law of sines
      if (fastOk && v < svLo && v >= 0.0) hit = true;
      else if (fastOk && v > svHi) hit = false;
      else hit = std::asin(v) < thresh;
    }
    if (hit) _mask[i] = 0;
  }
}

void SensorPolar2D::transform(Matrix* T)
{
  // (*_rays) = R * (*_rays), R = T[0:2,0:2]
  for (unsigned int i = 0; i < _size; i++) {
    const double x = _rays[i], y = _rays[_size + i];
    double nx = 0.0, ny = 0.0;
    nx += (*T)(0, 0) * x; nx += (*T)(0, 1) * y;
    ny += (*T)(1, 0) * x; ny += (*T)(1, 1) * y;
    _rays[i] = nx; _rays[_size + i] = ny;
  }
  _T = _T * (*T);   // P' = P T
}

const double* SensorPolar2D::getNormalizedRayMap(double norm)
{
  if (norm != _rayNorm) {
    for (unsigned int i = 0; i < _size; i++) {
      _rays[i] *= (norm / _rayNorm);
      _rays[_size + i] *= (norm / _rayNorm);
    }
    _rayNorm = norm;
  }
  return _rays.data();
}

unsigned int SensorPolar2D::dataToCartesianVectorMask(double* coords, bool* validityMask)
{
  unsigned int validPoints = 0;
  for (unsigned int i = 0; i < _size; i++) {
    if (!std::isinf(_data[i]) && _mask[i]) {
      coords[2 * i] = _raysLocal[i] * _data[i];
      coords[2 * i + 1] = _raysLocal[_size + i] * _data[i];
      validPoints++;
      validityMask[i] = true;
    } else {
      validityMask[i] = false;
    }
  }
  return validPoints;
}

int SensorPolar2D::backProject(double data[2])
{
  Matrix PoseInv = _T;
  PoseInv.invert();
  double lx = 0.0, ly = 0.0;
  lx += PoseInv(0, 0) * data[0]; lx += PoseInv(0, 1) * data[1]; lx += PoseInv(0, 2) * 1.0;
  ly += PoseInv(1, 0) * data[0]; ly += PoseInv(1, 1) * data[1]; ly += PoseInv(1, 2) * 1.0;
  const double phi = std::atan2(ly, lx);
  if (phi <= _phiLowerBound) return -2;
  if (phi >= _phiUpperBound) return -1;
  return (int)std::round((phi - _phiMin) / _angularRes);
}

SensorPolar2D* SensorPolar2D::copyForMapping() const
{
  // same geometry, pose and PROCESSED ranges (the reference constructs a fresh sensor and copies them;
  // cloning skips recomputing the 2 x size ray table, which costs more than the masking itself)
  SensorPolar2D* c = new SensorPolar2D(*this);
  c->_dev = nullptr;        // the device twin stays with the original
  c->_rays = c->_raysLocal; // a fresh sensor's rays: untransformed, norm 1 (unused by push)
  c->_rayNorm = 1.0;
  c->setStandardMask();     // a former NaN reading is +inf by now and becomes valid-infinite (Appendix B #20)
  return c;
}

// --------------------------------------------------------------------------------------- TsdGrid
TsdGrid::TsdGrid(double cellSize, EnumTsdGridLayout layoutPartition, EnumTsdGridLayout layoutGrid, int device)
    : _ctx(nullptr), _initialPushAccomplished(false)
{
  if (layoutPartition != LAYOUT_32x32) {
    std::fprintf(stderr, "TsdGrid: only LAYOUT_32x32 partitions are implemented (SlamNode.cpp:77)\n");
    return;
  }
  // TsdGrid::init leaves _maxTruncation = 2 * cellSize until setMaxTruncation (TsdGrid.cpp:136)
  _ctx = tsd_create(device, (int)layoutGrid, cellSize, 2.0 * cellSize);
}

TsdGrid::~TsdGrid()
{
  _batcher.reset();                    // (stops the dispatcher thread and frees its batch slots while the context still exists)
  std::lock_guard<std::mutex> lk(_mutex);
  tsd_destroy(_ctx);
  _ctx = nullptr;
}

void TsdGrid::reset()
{
  std::lock_guard<std::mutex> lk(_mutex);
  tsd_reset(_ctx);
  _initialPushAccomplished = false;
}

void TsdGrid::setMaxTruncation(double val)
{
  std::lock_guard<std::mutex> lk(_mutex);
  tsd_set_max_truncation(_ctx, val);
}

void TsdGrid::push(SensorPolar2D* sensor)
{
  double pose[9];
  sensor->getTransformation().getData(pose);
  std::lock_guard<std::mutex> lk(_mutex);
  const int rc = tsd_push(_ctx, pose, sensor->getRealMeasurementData(), sensor->maskBytes(),
                          (int)sensor->getRealMeasurementSize(), sensor->getAngularResolution(),
                          sensor->getPhiMin(), sensor->getMaximumRange(), sensor->getMinimumRange(),
                          sensor->getLowReflectivityRange(), nullptr);
  if (rc != TSD_OK) std::fprintf(stderr, "TsdGrid::push failed (%d): %s\n", rc, tsd_last_error(_ctx));
  _initialPushAccomplished = true;
}

bool TsdGrid::freeFootprint(const double centerCoords[2], double width, double height)
{
  std::lock_guard<std::mutex> lk(_mutex);
  return tsd_free_footprint(_ctx, centerCoords, width, height) == TSD_OK;
}

int TsdGrid::localize(SensorPolar2D* sensor, const tsd_icp_params& params, tsd_icp_result* result)
{
  double pose[9];
  sensor->getTransformation().getData(pose);
  const double* rays = sensor->getNormalizedRayMap(getCellSize());   // RayCastPolar2D.cpp:122
  std::lock_guard<std::mutex> lk(_mutex);
  return tsd_localize(_ctx, pose, rays, sensor->getLocalRayMap(), sensor->getRealMeasurementData(),
                      sensor->maskBytes(), (int)sensor->getRealMeasurementSize(),
                      sensor->getMinimumRange(), sensor->getMaximumRange(), &params, result);
}

int TsdGrid::raycast(SensorPolar2D* sensor, double* coords, double* normals, bool* mask, unsigned int* validPoints)
{
  double pose[9];
  sensor->getTransformation().getData(pose);
  const double* rays = sensor->getNormalizedRayMap(getCellSize());   // RayCastPolar2D.cpp:122
  int n = 0;
  std::lock_guard<std::mutex> lk(_mutex);
  const int rc = tsd_raycast(_ctx, pose, rays, (int)sensor->getRealMeasurementSize(), sensor->getMinimumRange(),
                             sensor->getMaximumRange(), coords, normals, reinterpret_cast<uint8_t*>(mask), &n);
  if (validPoints) *validPoints = (unsigned)n;
  return rc;
}

int TsdGrid::icp(const double* modelValid, unsigned nModel, const double* sceneValid, unsigned nScene, const Matrix& sensorPose,
                 const tsd_icp_params& params, tsd_icp_result* result)
{
  double pose[9];
  sensorPose.getData(pose);
  std::lock_guard<std::mutex> lk(_mutex);
  return tsd_icp(_ctx, modelValid, (int)nModel, sceneValid, (int)nScene, pose, &params, result);
}

int TsdGrid::attachSensor(SensorPolar2D* sensor)
{
  double pose[9];
  sensor->getTransformation().getData(pose);
  const double* rays = sensor->getNormalizedRayMap(getCellSize());   // RayCastPolar2D.cpp:122
  std::lock_guard<std::mutex> lk(_mutex);
  if (!sensor->deviceHandle()) {
    tsd_sensor* h = tsd_sensor_create(_ctx, (int)sensor->getRealMeasurementSize(), sensor->getAngularResolution(),
                                      sensor->getPhiMin(), sensor->getMaximumRange(), sensor->getMinimumRange(),
                                      sensor->getLowReflectivityRange());
    if (!h) return TSD_E_HIP;
    sensor->setDeviceHandle(h);
  }
  return tsd_sensor_set_pose(sensor->deviceHandle(), pose, rays, sensor->getLocalRayMap());
}

int TsdGrid::scan(SensorPolar2D* sensor, const uint8_t* maskPush, const tsd_icp_params& params,
                  const tsd_gate_params& gates, tsd_scan_result* result)
{
  std::lock_guard<std::mutex> lk(_mutex);
  const int rc = tsd_scan(sensor->deviceHandle(), sensor->getRealMeasurementData(), sensor->maskBytes(), maskPush,
                          &params, &gates, result);
  if (rc == TSD_OK && result->pushed) _initialPushAccomplished = true;
  return rc;
}

int TsdGrid::scanSubmit(SensorPolar2D* sensor, bool useStaged, const uint8_t* maskPush, const tsd_icp_params& params,
                        const tsd_gate_params& gates)
{
  std::lock_guard<std::mutex> lk(_mutex);
  if (useStaged) return tsd_scan_submit(sensor->deviceHandle(), nullptr, nullptr, nullptr, &params, &gates);
  return tsd_scan_submit(sensor->deviceHandle(), sensor->getRealMeasurementData(), sensor->maskBytes(), maskPush, &params, &gates);
}

int TsdGrid::scanStage(SensorPolar2D* sensor, const uint8_t* maskPush)
{
  std::lock_guard<std::mutex> lk(_mutex);
  return tsd_scan_stage(sensor->deviceHandle(), sensor->getRealMeasurementData(), sensor->maskBytes(), maskPush);
}

int TsdGrid::scanPreregister(SensorPolar2D* sensor, const tsd_tsdpdf_params& prm, const double* scene, const bool* maskS,
                             const int* dSub, const int* dCtrl, const int* dTrials)
{
  std::lock_guard<std::mutex> lk(_mutex);
  return tsd_scan_preregister(sensor->deviceHandle(), &prm, scene, reinterpret_cast<const uint8_t*>(maskS), dSub, dCtrl, dTrials);
}

int TsdGrid::scanCollect(SensorPolar2D* sensor, tsd_scan_result* result)
{
  // (the wait itself needs no lock: it polls the sensor's own result record)
  const int rc = tsd_scan_collect(sensor->deviceHandle(), result);
  if (rc == TSD_OK && result->pushed) _initialPushAccomplished = true;
  return rc;
}

void TsdGrid::enableBatchedScans(int robots, int slots)
{
  if (!_batcher && _ctx && robots > 1) _batcher.reset(new ScanBatcher(_ctx, robots, slots));
}

int TsdGrid::scanConcurrent(SensorPolar2D* sensor, const uint8_t* maskPush, const tsd_icp_params& params,
                            const tsd_gate_params& gates, tsd_scan_result* result)
{
  if (_batcher) {
    const int rc = _batcher->scan(sensor->deviceHandle(), sensor->getRealMeasurementData(), sensor->maskBytes(), maskPush, params, gates, result);
    if (rc == TSD_OK && result->pushed) _initialPushAccomplished = true;
    return rc;
  }
  // both halves run on this robot's thread without the facade's grid mutex: the C ABI orders the robots' ray casts and
  // pushes internally (ctx->order_mutex) and the sensor's private stream is this thread's alone
  int rc = tsd_scan_begin(sensor->deviceHandle(), sensor->getRealMeasurementData(), sensor->maskBytes(), maskPush, &params, &gates);
  if (rc != TSD_OK) return rc;
  rc = tsd_scan_finish(sensor->deviceHandle(), result);     // waits for the registration (other robots run meanwhile), then the push
  if (rc == TSD_OK && result->pushed) _initialPushAccomplished = true;
  return rc;
}

// --------------------------------------------------------------------------------------- ScanBatcher
static inline long long steady_ns()
{
  return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline void cpu_relax()
{
#if defined(__x86_64__)
  __builtin_ia32_pause();
#endif
}

ScanBatcher::ScanBatcher(tsd_ctx* ctx, int robots, int slots) : _ctx(ctx), _stop(false)
{
  if (slots < 1) slots = 1;
  if (slots > robots) slots = robots;
  _cap = (robots + slots - 1) / slots;
  // how long a batch waits for the other robots that are about to hand their next scan in (those whose previous scan came in
  // less than 2 ms ago); robots scanning at sensor rate never make anybody wait
  const char* e = std::getenv("TSD_BATCH_LINGER_US");
  _lingerNs = 1000ll * (e ? std::atol(e) : 120);
  _slots.resize((size_t)slots);
  for (auto& sl : _slots) sl.b = tsd_batch_create(ctx, _cap);
  _thread = std::thread([this] { run(); });
}

ScanBatcher::~ScanBatcher()
{
  {
    std::lock_guard<std::mutex> lk(_m);
    _stop = true;
  }
  _cvWork.notify_all();
  if (_thread.joinable()) _thread.join();
  for (auto& sl : _slots) tsd_batch_destroy(sl.b);
}

ScanBatcher::Stats ScanBatcher::stats()
{
  std::lock_guard<std::mutex> lk(_m);
  return _stats;
}

int ScanBatcher::scan(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* maskPush, const tsd_icp_params& params,
                      const tsd_gate_params& gates, tsd_scan_result* result)
{
  Request r;
  r.s = s; r.ranges = ranges; r.mask = mask; r.maskPush = maskPush; r.params = params; r.gates = gates; r.result = result;
  r.t_submit = steady_ns();
  {
    std::lock_guard<std::mutex> lk(_m);
    if (_stop) return TSD_E_ARG;
    _pending.push_back(&r);
    bool known = false;
    for (auto& ls : _lastSubmit) if (ls.first == s) { ls.second = r.t_submit; known = true; }
    if (!known) _lastSubmit.emplace_back(s, r.t_submit);
  }
  _cvWork.notify_one();
  // a round of batched registrations takes ~0.4 ms: spin for a few of those before sleeping (the wake-up of a sleeping
  // thread costs 50 us and more, and a robot that comes back late misses its batch)
  const long long t_spin_end = r.t_submit + 2000000;
  while (!r.done.load(std::memory_order_acquire)) {
    if (steady_ns() > t_spin_end) {
      std::unique_lock<std::mutex> lk(_m);
      _cvDone.wait(lk, [&] { return r.done.load(std::memory_order_acquire) != 0; });
      break;
    }
    cpu_relax();
  }
  return r.rc;
}

void ScanBatcher::complete(Request* r, int rc)
{
  r->rc = rc;
  r->done.store(1, std::memory_order_release);      // (the request lives on its thread's stack: not touched after this)
}

void ScanBatcher::run()
{
  std::vector<tsd_sensor*> sensors; std::vector<const double*> ranges; std::vector<const uint8_t*> mask, maskPush;
  std::vector<tsd_icp_params> params; std::vector<tsd_gate_params> gates; std::vector<tsd_scan_result> results;
  long long t_last_free = 0;                           // when the latest batch was handed out
  unsigned long long begun = 0;                        // batches begun so far; the oldest batch in flight is collected first
  std::vector<size_t> by_age(_slots.size());
  for (;;) {
    bool progress = false;
    // 1. batches whose result records have arrived: enqueue their pushes (unless that happened already), hand the results out
    for (size_t k = 0; k < _slots.size(); k++) by_age[k] = k;
    std::sort(by_age.begin(), by_age.end(), [&](size_t x, size_t y) { return _slots[x].order < _slots[y].order; });
    for (size_t k = 0; k < _slots.size(); k++) {
      Slot& sl = _slots[by_age[k]];
      if (sl.reqs.empty() || tsd_batch_poll(sl.b) != 1) continue;
      results.resize(sl.reqs.size());
      const int rc = tsd_batch_results(sl.b, results.data());
      {
        std::lock_guard<std::mutex> lk(_m);           // (pairs with the sleepers' predicate check)
        for (size_t i = 0; i < sl.reqs.size(); i++) {
          if (rc == TSD_OK) *sl.reqs[i]->result = results[i];
          complete(sl.reqs[i], rc);
        }
        sl.reqs.clear();
      }
      _cvDone.notify_all();
      t_last_free = steady_ns();
      progress = true;
    }
    // 2. a free slot and scans pending: begin a batch
    Slot* free_slot = nullptr;
    size_t free_idx = 0;
    for (size_t k = 0; k < _slots.size() && !free_slot; k++) if (_slots[k].reqs.empty()) { free_slot = &_slots[k]; free_idx = k; }
    bool any_inflight = false;
    for (auto& sl : _slots) any_inflight = any_inflight || !sl.reqs.empty();
    if (free_slot) {
      std::unique_lock<std::mutex> lk(_m);
      if (_stop && _pending.empty() && !any_inflight) return;
      if (_pending.empty() && !any_inflight) {         // idle: sleep until a scan comes in
        _cvWork.wait(lk, [&] { return _stop || !_pending.empty(); });
        if (_stop && _pending.empty()) return;
      }
      if (!_pending.empty()) {
        const long long now = steady_ns();
        // (a batch that has just been handed out brings its robots back within the linger time: count it from then, or the
        // scans that were waiting for the slot leave without them and the batches stay small for good)
        const long long since = std::max(_pending.front()->t_submit, t_last_free);
        bool go = (int)_pending.size() >= _cap || now - since > _lingerNs;
        if (!go) {
          // robots that are about to come back: their previous scan came in less than 2 ms ago and they are neither pending
          // nor in flight.  Nobody like that: no reason to wait.
          int expected = 0;
          for (auto& ls : _lastSubmit) {
            if (now - ls.second > 2000000) continue;
            bool busy = false;
            for (Request* r : _pending) busy = busy || r->s == ls.first;
            for (auto& sl : _slots) for (Request* r : sl.reqs) busy = busy || r->s == ls.first;
            if (!busy) expected++;
          }
          go = expected == 0;
        }
        if (go) {
          const size_t n = std::min(_pending.size(), (size_t)_cap);
          free_slot->reqs.assign(_pending.begin(), _pending.begin() + (long)n);
          _pending.erase(_pending.begin(), _pending.begin() + (long)n);
          _stats.batches++; _stats.scans += n;
        }
      }
      lk.unlock();
      if (!free_slot->reqs.empty()) {
        const size_t n = free_slot->reqs.size();
        sensors.resize(n); ranges.resize(n); mask.resize(n); maskPush.resize(n); params.resize(n); gates.resize(n);
        for (size_t i = 0; i < n; i++) {
          Request* r = free_slot->reqs[i];
          sensors[i] = r->s; ranges[i] = r->ranges; mask[i] = r->mask; maskPush[i] = r->maskPush; params[i] = r->params; gates[i] = r->gates;
        }
        const int rc = tsd_batch_begin(free_slot->b, (int)n, sensors.data(), ranges.data(), mask.data(), maskPush.data(), params.data(), gates.data());
        if (rc != TSD_OK) {
          {
            std::lock_guard<std::mutex> lk2(_m);
            for (Request* r : free_slot->reqs) complete(r, rc);
            free_slot->reqs.clear();
          }
          _cvDone.notify_all();
        } else {
          // the other slots' pushes go right behind this begin: this batch's ray casts do not wait for them
          free_slot->order = ++begun;
          for (size_t k = 0; k < _slots.size(); k++) {
            const size_t j = by_age[k];
            if (j != free_idx && !_slots[j].reqs.empty()) tsd_batch_push(_slots[j].b);
          }
        }
        progress = true;
      }
    }
    if (!progress) cpu_relax();
  }
}

// --------------------------------------------------------------------------------------- TSD_PDFMatching
TSD_PDFMatching::TSD_PDFMatching(TsdGrid& grid, unsigned int trials, double epsThresh, unsigned int sizeControlSet, double zrand)
    : _grid(grid), _trials(trials), _sizeControlSet(sizeControlSet), _epsThresh(epsThresh), _zrand(zrand), _seed(-1), _calls(0)
{
  std::memset(&_last, 0, sizeof(_last));
}

void TSD_PDFMatching::drawStreams(unsigned int points, std::vector<int>& dSub, std::vector<int>& dCtrl, std::vector<int>& dTrials)
{
  // the three rand() streams, in the reference's call order.  (Seeded -- tests only --: srand + the draws as ONE step under a process-
  // wide lock, so that several robots' localiser threads each get their own reproducible sequence; unseeded it is the reference's
  // plain rand(), whose interleaving between threads is as unspecified as in the reference.)
  static std::mutex seededDraws;
  std::unique_lock<std::mutex> lk(seededDraws, std::defer_lock);
  if (_seed >= 0) { lk.lock(); std::srand((unsigned)(_seed + (long)_calls)); }
  _calls++;
  dSub.assign(points, 0); dCtrl.assign(_sizeControlSet > 0 ? _sizeControlSet : 1, 0); dTrials.assign(_trials > 0 ? _trials : 1, 0);
  for (auto& v : dSub) v = std::rand();                      // RandomMatching::subsampleMask (RandomMatching.cpp:183)
  for (auto& v : dCtrl) v = std::rand();                     // RandomMatching::pickControlSet (:65)
  if (_seed < 0) std::srand((unsigned)time(NULL));           // TSD_PDFMatching.cpp:184
  for (auto& v : dTrials) v = std::rand();                   // TSD_PDFMatching.cpp:190
}

tsd_tsdpdf_params TSD_PDFMatching::params(double phiMax, double resolution) const
{
  tsd_tsdpdf_params prm;
  prm.trials = (int)_trials; prm.size_control_set = (int)_sizeControlSet; prm.eps_thresh = _epsThresh; prm.zrand = _zrand;
  prm.phi_max = phiMax; prm.ang_res = resolution;
  return prm;
}

Matrix TSD_PDFMatching::match(Matrix TSensor, const double* M, const bool* maskM, const double* /*NM*/, const double* S,
                              const bool* maskS, unsigned int points, double phiMax, const double /*transMax*/,
                              const double resolution)
{
  Matrix TBest(3, 3);
  TBest.setIdentity();
  std::vector<int> dSub, dCtrl, dTrials;
  drawStreams(points, dSub, dCtrl, dTrials);
  const tsd_tsdpdf_params prm = params(phiMax, resolution);
  double pose[9];
  TSensor.getData(pose);
  int rc;
  {
    std::lock_guard<std::mutex> lk(_grid.mutex());
    rc = tsd_tsdpdf_match(_grid.context(), pose, M, reinterpret_cast<const uint8_t*>(maskM), S,
                          reinterpret_cast<const uint8_t*>(maskS), (int)points, &prm, dSub.data(), dCtrl.data(), dTrials.data(), &_last);
  }
  if (rc != TSD_OK) {
    std::fprintf(stderr, "TSD_PDFMatching::match failed (%d): %s\n", rc, tsd_last_error(_grid.context()));
    return TBest;
  }
  TBest.setData(_last.T);
  return TBest;
}

}  // namespace obvious
