// obvious.h -- host-side mirror of the vendored `obvious` types that appear in the public signatures
// of ThreadSLAM / ThreadMapping / ThreadLocalize: Matrix (3x3 poses), SensorPolar2D (scan container
// + sensor model, kept on the host as SURVEY 8(a) rows S1/S3 prescribe) and TsdGrid, whose storage and
// arithmetic live on the GPU behind the C ABI of include/tsd_hip.h.
//
// Same names, argument meaning and error behaviour as the reference classes; citations are
// file:line of autonohm/ohm_tsd_slam (src/).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "tsd_hip.h"

namespace obvious {

// obvious::Matrix restricted to what the path needs (<= 4x4).  Products follow
// gsl_blas_dgemm's k-ascending accumulation (gsl/Matrix.cpp:45-52,88-95), the inverse follows
// gsl_linalg_LU_decomp + LU_invert (:168-179).
class Matrix
{
public:
  Matrix(unsigned rows = 3, unsigned cols = 3);
  Matrix(unsigned rows, unsigned cols, const double* data);
  double& operator()(unsigned r, unsigned c) { return _d[r * _cols + c]; }
  double operator()(unsigned r, unsigned c) const { return _d[r * _cols + c]; }
  unsigned getRows() const { return _rows; }
  unsigned getCols() const { return _cols; }
  void setIdentity();
  void setData(const double* array);
  void getData(double* array) const;
  void invert();
  Matrix operator*(const Matrix& M) const;
  const double* data() const { return _d; }
private:
  unsigned _rows, _cols;
  double _d[16];
};

// obvious::Sensor + obvious::SensorPolar2D (reconstruct/Sensor.{h,cpp}, grid/SensorPolar2D.{h,cpp})
class SensorPolar2D
{
public:
  SensorPolar2D(unsigned int size, double angularRes, double phiMin, double maxRange = INFINITY,
                double minRange = 0.0, double lowReflectivityRange = INFINITY);
  void setRealMeasurementData(const std::vector<float>& data, float scale = 1.f);   // Sensor.cpp:136-145
  void setRealMeasurementData(const double* data, double scale = 1.0);              // Sensor.cpp:125-134
  double* getRealMeasurementData() { return _data.data(); }
  bool* getRealMeasurementMask() { return reinterpret_cast<bool*>(_mask.data()); }
  const uint8_t* maskBytes() const { return _mask.data(); }
  unsigned int getRealMeasurementSize() const { return _size; }

  void setStandardMask();                       // SensorPolar2D.cpp:59-65
  void resetMask();                             // Sensor.cpp:246-250
  void maskZeroDepth();                         // Sensor.cpp:252-256
  void maskInvalidDepth();                      // Sensor.cpp:258-272
  void maskDepthDiscontinuity(double thresh);   // SensorPolar2D.cpp:67-98

  void transform(Matrix* T);                    // Sensor.cpp:50-60
  Matrix getTransformation() const { return _T; }
  void setTransformation(const Matrix& T) { _T = T; }
  void getPosition(double* tr) const { tr[0] = _T(0, 2); tr[1] = _T(1, 2); }   // Sensor.cpp:114-118

  double getMaximumRange() const { return _maxRange; }
  double getMinimumRange() const { return _minRange; }
  double getLowReflectivityRange() const { return _lowReflectivityRange; }
  double getAngularResolution() const { return _angularRes; }
  double getPhiMin() const { return _phiMin; }

  const double* getNormalizedRayMap(double norm);   // Sensor.cpp:36-48; 2 x size, row-major
  const double* getLocalRayMap() const { return _raysLocal.data(); }
  unsigned int dataToCartesianVectorMask(double* coords, bool* validityMask);   // Sensor.cpp:168-190
  int backProject(double data[2]);              // SensorPolar2D.cpp:100-115
  /** the deep copy ThreadMapping::queuePush hands to the mapping thread (ThreadMapping.cpp:65-76):
   *  same geometry and pose, the PROCESSED ranges, mask rebuilt by setStandardMask() */
  SensorPolar2D* copyForMapping() const;
  /** mask of that copy without building it */
  void maskForMapping(std::vector<uint8_t>& out) const;

  // device-resident twin of this sensor on a grid context (fused scan path); owned by the sensor
  tsd_sensor* deviceHandle() const { return _dev; }
  void setDeviceHandle(tsd_sensor* h) { _dev = h; }
  ~SensorPolar2D() { if (_dev) tsd_sensor_destroy(_dev); }
  SensorPolar2D(const SensorPolar2D&) = default;
  SensorPolar2D& operator=(const SensorPolar2D&) = delete;

private:
  tsd_sensor* _dev = nullptr;
  unsigned int _size;
  double _angularRes, _phiMin, _phiLowerBound, _phiUpperBound;
  double _maxRange, _minRange, _lowReflectivityRange;
  double _rayNorm;
  Matrix _T;
  std::vector<double> _data;
  std::vector<uint8_t> _mask;
  std::vector<unsigned int> _nanBeams;   // readings that were NaN when the mask was built
  std::vector<double> _rays, _raysLocal;
};

enum EnumTsdGridLayout { LAYOUT_1x1 = 0, LAYOUT_2x2 = 1, LAYOUT_4x4 = 2, LAYOUT_8x8 = 3, LAYOUT_16x16 = 4,
  LAYOUT_32x32 = 5, LAYOUT_64x64 = 6, LAYOUT_128x128 = 7, LAYOUT_256x256 = 8, LAYOUT_512x512 = 9,
  LAYOUT_1024x1024 = 10, LAYOUT_2048x2048 = 11, LAYOUT_4096x4096 = 12, LAYOUT_8192x8192 = 13,
  LAYOUT_16384x16384 = 14, LAYOUT_36768x36768 = 15 };   // TsdGrid.h:11-26

// Dispatcher of the multi-robot mode (addition; the reference runs N ThreadLocalize workers against one TsdGrid and lets the
// scheduler interleave them, SlamNode.cpp:101-122).  The robots' threads hand their scans in and block; ONE dispatcher thread
// groups the scans that are pending into batches (tsd_batch_*: one launch of each kernel for all robots of the batch) and
// uses two batch slots in turn, so that one batch registers while the other one's pushes run.  The pushes of a slot are
// enqueued right behind the NEXT begin of the other slot (or when the slot's results have been seen, whichever comes
// first): the other slot's ray casts then do not wait for them, and the slot's own next ray casts do.
class ScanBatcher
{
public:
  ScanBatcher(tsd_ctx* ctx, int robots, int slots);
  ~ScanBatcher();
  /** blocking: returns when the registration of this scan has finished (its push is enqueued in stream order) */
  int scan(tsd_sensor* s, const double* ranges, const uint8_t* mask, const uint8_t* maskPush, const tsd_icp_params& params,
           const tsd_gate_params& gates, tsd_scan_result* result);
  struct Stats { unsigned long long batches = 0, scans = 0; };
  Stats stats();
private:
  struct Request {
    tsd_sensor* s; const double* ranges; const uint8_t* mask; const uint8_t* maskPush;
    tsd_icp_params params; tsd_gate_params gates; tsd_scan_result* result;
    int rc = 0; std::atomic<int> done{0};
    long long t_submit = 0;
  };
  struct Slot { tsd_batch* b = nullptr; std::vector<Request*> reqs; unsigned long long order = 0; };
  void run();
  void complete(Request* r, int rc);
  tsd_ctx* _ctx;
  int _cap;
  long long _lingerNs;
  std::vector<Slot> _slots;
  std::deque<Request*> _pending;
  std::vector<std::pair<tsd_sensor*, long long>> _lastSubmit;    // per sensor: when its latest scan came in
  std::mutex _m;
  std::condition_variable _cvWork, _cvDone;
  bool _stop;
  Stats _stats;
  std::thread _thread;
};

// obvious::TsdGrid: the handle of a grid that lives in HBM.  Only LAYOUT_32x32 partitions exist
// (SlamNode.cpp:77 hard-codes it).  Every device call of this grid is serialised by mutex().
class TsdGrid
{
public:
  TsdGrid(double cellSize, EnumTsdGridLayout layoutPartition, EnumTsdGridLayout layoutGrid, int device = 0);
  virtual ~TsdGrid();
  TsdGrid(const TsdGrid&) = delete;
  TsdGrid& operator=(const TsdGrid&) = delete;

  bool valid() const { return _ctx != nullptr; }
  void reset();                                         // TsdGrid.cpp:194-198
  unsigned int getCellsX() const { return (unsigned)tsd_cells(_ctx); }
  unsigned int getCellsY() const { return (unsigned)tsd_cells(_ctx); }
  double getCellSize() const { return tsd_cell_size(_ctx); }
  double getMinX() const { return tsd_min_x(_ctx); }
  double getMaxX() const { return tsd_max_x(_ctx); }
  double getMinY() const { return tsd_min_y(_ctx); }
  double getMaxY() const { return tsd_max_y(_ctx); }
  unsigned int getPartitionSize() const { return TSD_TILE_DIM; }
  void setMaxTruncation(double val);                    // TsdGrid.cpp:206-215
  double getMaxTruncation() const { return tsd_max_truncation(_ctx); }
  virtual void push(SensorPolar2D* sensor);             // TsdGrid.cpp:217-284 (asynchronous on the stream)
  bool containsData() const { return _initialPushAccomplished; }
  virtual bool freeFootprint(const double centerCoords[2], double width, double height);   // TsdGrid.cpp:609-638

  // ThreadLocalize's ray cast + registration, fused on the device (tsd_localize)
  virtual int localize(SensorPolar2D* sensor, const tsd_icp_params& params, tsd_icp_result* result);

  // the two halves of localize() as separate calls (registration_mode 1-3 runs a pre-registration between them,
  // ThreadLocalize.cpp:353-377, :531-581): RayCastPolar2D::calcCoordsFromCurrentViewMask and Icp::iterate with Tinit
  virtual int raycast(SensorPolar2D* sensor, double* coords, double* normals, bool* mask, unsigned int* validPoints);
  virtual int icp(const double* modelValid, unsigned nModel, const double* sceneValid, unsigned nScene, const Matrix& sensorPose,
                  const tsd_icp_params& params, tsd_icp_result* result);

  // fused scan path: the sensor's pose / ray maps live on the device next to the grid, one call runs
  // ray cast -> registration -> gates -> Sensor::transform -> push in stream order (tsd_scan)
  virtual int attachSensor(SensorPolar2D* sensor);
  virtual int scan(SensorPolar2D* sensor, const uint8_t* maskPush, const tsd_icp_params& params,
                   const tsd_gate_params& gates, tsd_scan_result* result);

  // tsd_scan in two halves with the next scan staged in between (tsd_scan_submit / _stage / _collect): a caller that
  // already holds the next LaserScan has its copy and tables done while the current registration runs.
  // scanSubmit(sensor == data source or nullptr for the scan staged by scanStage)
  virtual int scanSubmit(SensorPolar2D* sensor, bool useStaged, const uint8_t* maskPush, const tsd_icp_params& params,
                         const tsd_gate_params& gates);
  virtual int scanStage(SensorPolar2D* sensor, const uint8_t* maskPush);
  // registration_mode 3 inside the fused scan: arms the pre-registration of the sensor's next scanSubmit (tsd_scan_preregister)
  virtual int scanPreregister(SensorPolar2D* sensor, const tsd_tsdpdf_params& prm, const double* scene, const bool* maskS,
                              const int* dSub, const int* dCtrl, const int* dTrials);
  virtual int scanCollect(SensorPolar2D* sensor, tsd_scan_result* result);

  // the same scan in two halves for several robots on this one grid (tsd_scan_begin / _wait / _finish): the mutex is
  // held for the two enqueue steps only, the wait in between is lock-free, so the robots' registrations overlap
  virtual int scanConcurrent(SensorPolar2D* sensor, const uint8_t* maskPush, const tsd_icp_params& params,
                             const tsd_gate_params& gates, tsd_scan_result* result);

  /** multi-robot mode: route scanConcurrent through a ScanBatcher (batched launches, `slots` batch slots used in turn) */
  void enableBatchedScans(int robots, int slots = 2);
  ScanBatcher* batcher() { return _batcher.get(); }

  tsd_ctx* context() { return _ctx; }
  std::mutex& mutex() { return _mutex; }
protected:
  tsd_ctx* _ctx;
  std::unique_ptr<ScanBatcher> _batcher;
  std::mutex _mutex;
  bool _initialPushAccomplished;
};

// obvious::TSD_PDFMatching (registration/ransacMatching/TSD_PDFMatching.{h,cpp}) on obvious::RandomMatching: the
// pre-registration of registration_mode 3.  Same constructor and match() signature as the reference; the scoring of
// the candidates runs on the device (tsd_tsdpdf_match), the three rand() streams are drawn here, where the reference
// draws them (RandomMatching.cpp:65, :183; TSD_PDFMatching.cpp:190).
class TSD_PDFMatching
{
public:
  TSD_PDFMatching(TsdGrid& grid, unsigned int trials = 50, double epsThresh = 0.15, unsigned int sizeControlSet = 140,
                  double zrand = 0.05);
  virtual ~TSD_PDFMatching() {}
  /** NM (model normals) is accepted for signature compatibility; ThreadLocalize passes NULL and the normals come from
   *  the PCA of RandomMatching::calcNormals, as in the reference's call (ThreadLocalize.cpp:559) */
  Matrix match(Matrix TSensor, const double* M, const bool* maskM, const double* NM, const double* S, const bool* maskS,
               unsigned int points, double phiMax = M_PI / 4.0, const double transMax = 1.5, const double resolution = 0.0);
  /** addition (tests): >= 0 makes the draws reproducible -- srand(seed + number of calls so far) before drawing; the
   *  default -1 keeps the reference's behaviour (srand(time(NULL)) before the trial picks, TSD_PDFMatching.cpp:184) */
  void setSeed(long seed) { _seed = seed; }
  const tsd_tsdpdf_result& lastResult() const { return _last; }
  /** the three rand() streams of one match() call, drawn in the reference's order (RandomMatching.cpp:183, :65;
   *  TSD_PDFMatching.cpp:184-190) -- what match() itself uses and what the fused scan hands to tsd_scan_preregister */
  void drawStreams(unsigned int points, std::vector<int>& dSub, std::vector<int>& dCtrl, std::vector<int>& dTrials);
  /** the parameters of match() as the C ABI takes them */
  tsd_tsdpdf_params params(double phiMax, double resolution) const;
private:
  TsdGrid& _grid;
  unsigned int _trials, _sizeControlSet;
  double _epsThresh, _zrand;
  long _seed;
  unsigned long _calls;
  tsd_tsdpdf_result _last;
};

}  // namespace obvious
