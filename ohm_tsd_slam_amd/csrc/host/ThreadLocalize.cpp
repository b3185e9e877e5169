#include "ThreadLocalize.h"
#include "ThreadMapping.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <limits>

namespace ohm_tsd_slam
{

namespace
{
// declares the parameter with the reference's default if SlamNode / a YAML did not provide it
template <class T>
rclcpp::Parameter param(const std::shared_ptr<rclcpp::Node>& node, const std::string& name, const T& def)
{
#if OHM_TSD_SLAM_HAVE_ROS
  if(!node->has_parameter(name))
    node->declare_parameter<T>(name, def);
#else
  node->declare_parameter(name, def);
#endif
  return node->get_parameter(name);
}

// tf2::Quaternion::setEuler(yaw = 0, pitch = 0, roll = theta): rotation about z (ThreadLocalize.cpp:608-609)
void setYaw(geometry_msgs::msg::Quaternion& q, double theta)
{
  q.x = 0.0;
  q.y = 0.0;
  q.z = std::sin(theta * 0.5);
  q.w = std::cos(theta * 0.5);
}
}

ThreadLocalize::ThreadLocalize(obvious::TsdGrid* grid, ThreadMapping* mapper, const std::shared_ptr<rclcpp::Node>& node,
                               const std::string& robot_name, const double xOffset, const double yOffset):
    ThreadSLAM(*grid),
    _node(node),
    _mapper(*mapper),
    _sensor(nullptr),
    _initialized(false),
    _synchronous(false),
    _fused(true),
    _concurrent(false),
    _gridWidth(grid->getCellsX() * grid->getCellSize()),
    _gridHeight(grid->getCellsY() * grid->getCellSize()),
    _gridOffSetX(-(grid->getCellsX() * grid->getCellSize() * 0.5 + xOffset)),
    _gridOffSetY(-(grid->getCellsY() * grid->getCellSize() * 0.5 + yOffset)),
    _xOffset(xOffset),
    _yOffset(yOffset),
    _robotName(robot_name),
    _reverseScan(false),
    _lastPose(new obvious::Matrix(3, 3)),
    _haveLastPose(false),
    _busy(false),
    _processed(0)
{
  // namespaces (ThreadLocalize.cpp:76-83)
  if(_robotName.size() > 0 && _robotName.back() != '/')
    _robotName += "/";
  const std::string node_name = std::string(node->get_name()) + "/";
  _nameSpace = node_name + _robotName;
  const std::string poseTopic = _nameSpace + "estimated_pose";

  // parameters: names, types and defaults of ThreadLocalize.cpp:86-129 (Appendix D)
  declareParameters(node, _robotName);
  const double distFilterMax = node->get_parameter(_robotName + "dist_filter_max").as_double();
  const double distFilterMin = node->get_parameter(_robotName + "dist_filter_min").as_double();
  const int icpIterations    = (int)node->get_parameter(_robotName + "icp_iterations").as_int();
  _tfLaserFrameId     = node->get_parameter(_robotName + "tf_laser_frame").as_string();
  _tfMapFrameId       = param(node, "tf_map_frame", std::string("map")).as_string();     // declared by SlamNode (SlamNode.cpp:58)
  _tfOdomFrameId      = node->get_parameter(_robotName + "tf_odom_frame").as_string();
  _tfFootprintFrameId = node->get_parameter(_robotName + "tf_footprint_frame").as_string();
  _trnsMax     = node->get_parameter("reg_trs_max").as_double();
  _rotMax      = node->get_parameter("reg_sin_rot_max").as_double();
  _lasMinRange = node->get_parameter("laser_min_range").as_double();
  _regMode     = (int)node->get_parameter(_robotName + "registration_mode").as_int();
  _ranPhiMax   = node->get_parameter(_robotName + "ransac_phi_max").as_double();
  _preMatcher.reset();
  switch(_regMode)
  {
  case 0:   // ICP: no instance needed
    break;
  case 3:   // TSD (ThreadLocalize.cpp:193): TSD_PDFMatching(_grid, trials, epsThresh, sizeControlSet, zrand)
    _preMatcher.reset(new obvious::TSD_PDFMatching(*grid, (unsigned)node->get_parameter("trials").as_int(),
                                                   node->get_parameter("epsThresh").as_double(),
                                                   (unsigned)node->get_parameter("sizeControlSet").as_int(),
                                                   node->get_parameter("zrand").as_double()));
    _preMatcher->setSeed((long)param(node, _robotName + "tsdpdf_seed", -1).as_int());   // addition: reproducible draws for tests
    break;
  case 1:   // EXP: RandomNormalMatching
  case 2:   // PDF: PDFMatching
    std::fprintf(stderr, "Localizer(%s): registration mode %d (RandomNormalMatching / PDFMatching pre-registration) is not part of "
                         "the GPU hot path (SURVEY 2: out of scope); using default = ICP.\n", _nameSpace.c_str(), _regMode);
    _regMode = 0;
    break;
  default:  // ThreadLocalize.cpp:188-190
    std::fprintf(stderr, "Localizer(%s): Unknown registration mode %d use default = ICP.\n", _nameSpace.c_str(), _regMode);
    _regMode = 0;
  }

  // ICP set-up (ThreadLocalize.cpp:211-225): DistanceFilter(max, min, icpIterations - 10), bounds filter
  // over the grid extent, maxRMS 0, max iterations == convergence counter == icpIterations
  std::memset(&_icpParams, 0, sizeof(_icpParams));
  _icpParams.iterations      = icpIterations;
  // "async_mapping" (addition, default 0): the reference's mapper is a thread of its own -- queuePush returns at once and the push lands
  // when the mapping thread gets to it (ThreadMapping.cpp:51-76).  0 keeps the strict order this facade has by default (the next ray
  // cast sees this scan's push); 1 lets the push run beside the next registration, the next ray cast exactly one push behind.
  _asyncMapping = param(node, _robotName + "async_mapping", 0).as_int() != 0;
  // The node constructs ClosedFormEstimator2D (ThreadLocalize.cpp:214); "icp_estimator" = 1 selects the reference's
  // other estimator, PointToLine2DEstimator, on the ray cast's normals (an addition: the reference has no such key)
  _icpParams.estimator       = (int)param(node, _robotName + "icp_estimator", 0).as_int() == 1
                                   ? TSD_ESTIMATOR_POINT_TO_LINE : TSD_ESTIMATOR_CLOSED_FORM;
  _icpParams.dist_filter_max = distFilterMax;
  _icpParams.dist_filter_min = distFilterMin;
  _icpParams.min_x = grid->getMinX();
  _icpParams.max_x = grid->getMaxX();
  _icpParams.min_y = grid->getMinY();
  _icpParams.max_y = grid->getMaxY();

  _posePub = _node->create_publisher<geometry_msgs::msg::PoseStamped>(poseTopic, 1);
  _tfBroadcaster = std::make_unique<tf2_ros::TransformBroadcaster>(*_node);
  _tf_buffer = std::make_unique<tf2_ros::Buffer>(_node->get_clock());                     // ThreadLocalize.cpp:198-199
  _tf_transform_listener = std::make_unique<tf2_ros::TransformListener>(*_tf_buffer);
  _poseStamped.header.frame_id = _tfMapFrameId;
  _tf.header.frame_id = _tfMapFrameId;
  _tf.child_frame_id = _robotName + _tfOdomFrameId;
  std::memset(&_report, 0, sizeof(_report));
  startThread();
}

ThreadLocalize::~ThreadLocalize()
{
  terminateThread();
  joinThread();
  delete _sensor;
  delete _lastPose;
  _laserData.clear();
}

// ThreadLocalize.cpp:86-129: every parameter the reference's constructor declares, with its type and default
// (tests/test_cpu_abi_and_host.py diffs this set against a table transcribed from those lines)
void ThreadLocalize::declareParameters(const std::shared_ptr<rclcpp::Node>& node, const std::string& robotName)
{
  (void)param(node, robotName + "dist_filter_max", 1.0);          // DIST_FILT_MAX (ThreadLocalize.h:66)
  (void)param(node, robotName + "dist_filter_min", 0.1);          // DIST_FILT_MIN (:65)
  (void)param(node, robotName + "icp_iterations", 25);            // ICP_ITERATIONS (:58)
  (void)param(node, robotName + "tf_laser_frame", robotName + "laser");
  (void)param(node, robotName + "tf_odom_frame", robotName + "odom");
  (void)param(node, robotName + "tf_footprint_frame", robotName + "base_footprint");
  (void)param(node, "reg_trs_max", 0.25);                         // TRNS_THRESH
  (void)param(node, "reg_sin_rot_max", 0.17);                     // ROT_THRESH
  (void)param(node, "max_velocity_lin", 1.5);                     // TRNS_VEL_MAX
  (void)param(node, "max_velocity_rot", 2.0 * M_PI);              // ROT_VEL_MAX
  (void)param(node, "ude_odom_rescue", false);
  (void)param(node, "wait_for_odom_tf", 1.0);
  (void)param(node, "laser_min_range", 0.0);
  (void)param(node, "trials", 100);
  (void)param(node, "sizeControlSet", 140);
  (void)param(node, "epsThresh", 0.15);
  (void)param(node, "zhit", 0.45);
  (void)param(node, "zphi", 0.0);
  (void)param(node, "zshort", 0.25);
  (void)param(node, "zmax", 0.05);
  (void)param(node, "zrand", 0.25);
  (void)param(node, "percentagePointsInC", 0.9);
  (void)param(node, "rangemax", 20.0);
  (void)param(node, "sigphi", M_PI / 180.0 * 3);
  (void)param(node, "sighit", 0.2);
  (void)param(node, "lamshort", 0.08);
  (void)param(node, "maxAngleDiff", 3.0);
  (void)param(node, "maxAnglePenalty", 0.5);
  (void)param(node, robotName + "ransac_trials", 50);             // RANSAC_TRIALS
  (void)param(node, robotName + "ransac_eps_thresh", 0.15);       // RANSAC_EPS_THRESH
  (void)param(node, robotName + "ransac_ctrlset_size", 180);      // RANSAC_CTRL_SET_SIZE
  (void)param(node, robotName + "ransac_phi_max", 30.0);
  (void)param(node, robotName + "registration_mode", 0);          // ICP
}

// ThreadLocalize::init (ThreadLocalize.cpp:424-432): the per-robot parameters declared with the first scan
void ThreadLocalize::declareInitParameters(const std::shared_ptr<rclcpp::Node>& node, const std::string& nameSpace)
{
  (void)param(node, nameSpace + "local_offset_x", 0.0);
  (void)param(node, nameSpace + "local_offset_y", 0.0);
  (void)param(node, nameSpace + "local_offset_yaw", 0.0);
  (void)param(node, nameSpace + "max_range", 30.0);
  (void)param(node, nameSpace + "min_range", 0.001);
  (void)param(node, nameSpace + "low_reflectivity_range", 2.0);
  (void)param(node, nameSpace + "footprint_width", 1.0);
  (void)param(node, nameSpace + "footprint_height", 1.0);
  (void)param(node, nameSpace + "footprint_x_offset", 0.28);
}

void ThreadLocalize::laserCallBack(const std::shared_ptr<sensor_msgs::msg::LaserScan> scan)
{
  // the reference clamps in place through an aliased shared_ptr (Appendix B #15); copy instead
  auto scanCopy = std::make_shared<sensor_msgs::msg::LaserScan>(*scan);
  for(auto& iter : scanCopy->ranges)
  {
    if(iter < _lasMinRange)
      iter = 0.0;
  }
  if(!_initialized)
  {
    this->init(*scanCopy);
    _stampLaserOld = scan->header.stamp;
  }
  else if(_synchronous)
  {
    std::vector<float> ranges = scanCopy->ranges;
    if(_reverseScan)
      std::reverse(ranges.begin(), ranges.end());
    processScan(ranges, scanCopy->header.stamp);
  }
  else
  {
    {
      std::lock_guard<std::mutex> lk(_dataMutex);
      _laserData.push_front(scanCopy);
    }
    this->unblock();
  }
}

void ThreadLocalize::announceNext(const std::shared_ptr<sensor_msgs::msg::LaserScan> scan)
{
  auto copy = std::make_shared<sensor_msgs::msg::LaserScan>(*scan);
  for(auto& iter : copy->ranges)
  {
    if(iter < _lasMinRange)
      iter = 0.0;
  }
  if(_reverseScan)
    std::reverse(copy->ranges.begin(), copy->ranges.end());
  std::lock_guard<std::mutex> lk(_dataMutex);
  _ahead = copy;
}

void ThreadLocalize::eventLoop(void)
{
  while(_stayActive)
  {
    waitForWork();
    for(;;)
    {
      std::vector<float> ranges;
      builtin_interfaces::msg::Time stamp;
      {
        std::lock_guard<std::mutex> lk(_dataMutex);
        if(!_stayActive || _laserData.empty())
          break;
        // newest scan wins, older ones are dropped (ThreadLocalize.cpp:319-332)
        ranges = _laserData.front()->ranges;
        stamp = _laserData.front()->header.stamp;
        _laserData.clear();
        _busy = true;
      }
      if(_reverseScan)
        std::reverse(ranges.begin(), ranges.end());
      processScan(ranges, stamp);
      {
        std::lock_guard<std::mutex> lk(_dataMutex);
        _busy = false;
      }
    }
  }
}

void ThreadLocalize::processScan(const std::vector<float>& ranges, const builtin_interfaces::msg::Time& stamp)
{
  ScanReport rep;
  std::memset(&rep, 0, sizeof(rep));
  rep.initialised = true;
  rep.stampNs = (long long)stamp.sec * 1000000000LL + (long long)stamp.nanosec;
  _stampLaserOld = _stampLaser;
  _stampLaser = stamp;

  // (a scan that was announced, ingested and staged on the device during the previous registration is in _sensor already)
  // Accepted only if it IS this scan: same stamp AND the same readings (drivers that leave the stamp at 0 or repeat it must not
  // get the previously staged ranges registered in place of the scan they delivered); anything else is a drop.
  const bool staged = _stagedValid && _stagedStampNs == rep.stampNs && _stagedRanges.size() == ranges.size() &&
                      (ranges.empty() || std::memcmp(_stagedRanges.data(), ranges.data(), ranges.size() * sizeof(float)) == 0);
  if(_stagedValid && !staged) _stagedValid = false;     // something else came: tsd_scan_submit drops the staged scan
  if(!staged)
  {
    _sensor->setRealMeasurementData(ranges);
    _sensor->setStandardMask();
  }

  if(!_haveLastPose)   // first call (ThreadLocalize.cpp:342-350)
  {
    *_lastPose = _sensor->getTransformation();
    _haveLastPose = true;
  }

  // registration_mode 3: inside the fused scan (the pre-registration on the device between the ray cast and the registration) when
  // this robot has the fused path to itself; the reference's own call structure otherwise (and with TSD_MODE3_UNFUSED set: A/B)
  static const bool mode3Unfused = std::getenv("TSD_MODE3_UNFUSED") != nullptr;
  // (several robots on one grid: the batched dispatcher runs the pre-registration behind its batch's ray casts; the split scan does not)
  if(_regMode == 3 && _preMatcher && (mode3Unfused || !_preFusedOk || !_fused || !_sensor->deviceHandle() || (_concurrent && !_grid.batcher())))
  {
    processScanPreRegistered(rep);
    return;
  }
  if(_fused && _sensor->deviceHandle())
  {
    processScanFused(rep);
    return;
  }

  // reconstruction + registration on the device (ThreadLocalize.cpp:353-377)
  tsd_icp_result res;
  std::memset(&res, 0, sizeof(res));
  const int rc = _grid.localize(_sensor, _icpParams, &res);
  _sensor->getTransformation().getData(rep.pose);
  if(rc != TSD_OK)
  {
    std::fprintf(stderr, "Localizer(%s) device error %d\n", _nameSpace.c_str(), rc);
    std::lock_guard<std::mutex> lk(_reportMutex);
    _report = rep; _processed++;
    return;
  }
  rep.validModel = res.n_model; rep.validScene = res.n_scene;
  if(res.n_model == 0)
  {
    // "Raycasting found no coordinates" -> skip the scan (ThreadLocalize.cpp:354-358)
    rep.noModel = true;
    std::lock_guard<std::mutex> lk(_reportMutex);
    _report = rep; _processed++;
    return;
  }
  finishScan(rep, res);
}

// registration_mode 3 (ThreadLocalize.cpp:353-406 with doRegistration's case TSD, :557-567): ray cast, the TSD_PDF
// pre-registration on the beam-indexed model / scene, then Icp::iterate with its result as Tinit.  The unfused call
// structure of the reference: the pre-registration's host part sits between the ray cast and the registration.
namespace {
// TSD_MODE3_TIMING=1: where a registration_mode-3 scan's host time goes (printed every 100 scans)
struct Mode3Timing {
  bool on = std::getenv("TSD_MODE3_TIMING") != nullptr;
  double acc[6] = {0, 0, 0, 0, 0, 0}; int scans = 0;
  std::chrono::steady_clock::time_point t;
  void start() { if(on) t = std::chrono::steady_clock::now(); }
  void lap(int i) { if(!on) return; const auto now = std::chrono::steady_clock::now(); acc[i] += std::chrono::duration<double, std::micro>(now - t).count(); t = now; }
  void scan() {
    if(!on || ++scans % 100) return;
    std::fprintf(stderr, "mode 3, us per scan: ray cast %.1f | scene points %.1f | TSD_PDF match %.1f | registration (fused) %.1f | finish + push %.1f\n",
                 acc[0] / scans, acc[1] / scans, acc[2] / scans, acc[3] / scans, acc[4] / scans);
  }
};
// (one per localiser thread: every robot of a multi-robot node takes this path from its own thread)
thread_local Mode3Timing g_m3;
}

void ThreadLocalize::processScanPreRegistered(ScanReport& rep)
{
  g_m3.start();
  const unsigned int n = _sensor->getRealMeasurementSize();
  if(_modelCoords.size() != 2 * (size_t)n)      // first call: buffers (ThreadLocalize.cpp:342-350)
  {
    _modelCoords.assign(2 * (size_t)n, 0.0); _modelNormals.assign(2 * (size_t)n, 0.0); _scene.assign(2 * (size_t)n, 0.0);
    _maskM.assign(n, 0); _maskS.assign(n, 0);
  }
  bool* maskM = reinterpret_cast<bool*>(_maskM.data());
  bool* maskS = reinterpret_cast<bool*>(_maskS.data());
  unsigned int validModelPoints = 0;
  const int rcR = _grid.raycast(_sensor, _modelCoords.data(), _modelNormals.data(), maskM, &validModelPoints);
  g_m3.lap(0);
  _sensor->getTransformation().getData(rep.pose);
  rep.validModel = (int)validModelPoints;
  if(rcR != TSD_OK || validModelPoints == 0)
  {
    if(rcR != TSD_OK) std::fprintf(stderr, "Localizer(%s) device error %d\n", _nameSpace.c_str(), rcR);
    rep.noModel = rcR == TSD_OK;
    std::lock_guard<std::mutex> lk(_reportMutex);
    _report = rep; _processed++;
    return;
  }
  const unsigned int validScenePoints = _sensor->dataToCartesianVectorMask(_scene.data(), maskS);
  rep.validScene = (int)validScenePoints;
  g_m3.lap(1);
  // doRegistration, case TSD: T = _TSD_PDFMatcher->match(sensor->getTransformation(), M, _maskM, NULL, S, _maskS,
  //                                                      deg2rad(_ranPhiMax), _trnsMax, sensor->getAngularResolution())
  obvious::Matrix Tpre = _preMatcher->match(_sensor->getTransformation(), _modelCoords.data(), maskM, nullptr, _scene.data(), maskS,
                                            n, _ranPhiMax * M_PI / 180.0, _trnsMax, _sensor->getAngularResolution());
  g_m3.lap(2);
  tsd_icp_params p = _icpParams;
  Tpre.getData(p.t_init);
  p.use_t_init = 1;
  tsd_icp_result res;
  std::memset(&res, 0, sizeof(res));
  // Icp::iterate with Tinit on the maskMatrix-compacted point sets (ThreadLocalize.cpp:367-377, :738-755): the fused device path --
  // ray cast again (12 us), compaction of both sets by their masks and the registration without leaving the device -- instead of
  // compacting on the host and sending both sets back: tsd_icp on arbitrary point sets sorts the model by angle on the host and
  // costs ~90 us of host work and copies per call, which the ray cast's own beam order makes unnecessary
  const int rc = _grid.localize(_sensor, p, &res);
  if(rc != TSD_OK)
  {
    std::fprintf(stderr, "Localizer(%s) device error %d\n", _nameSpace.c_str(), rc);
    std::lock_guard<std::mutex> lk(_reportMutex);
    _report = rep; _processed++;
    return;
  }
  g_m3.lap(3);
  finishScan(rep, res);
  g_m3.lap(4); g_m3.scan();
}

// what follows the registration (ThreadLocalize.cpp:381-406): gate, Sensor::transform, pose / tf, push decision
void ThreadLocalize::finishScan(ScanReport& rep, const tsd_icp_result& res)
{
  obvious::Matrix T(3, 3, res.T);
  std::memcpy(rep.T, res.T, sizeof(rep.T));
  rep.rms = res.rms; rep.pairs = res.pairs; rep.iterations = res.iterations; rep.icpState = res.state;
  if(isRegistrationError(&T, _trnsMax, _rotMax))
  {
    rep.regError = true;
    sendNanTransform();
  }
  else
  {
    _sensor->transform(&T);
    obvious::Matrix curPose = _sensor->getTransformation();
    curPose.getData(rep.pose);
    sendTransform(&curPose);
    if(this->isPoseChangeSignificant(_lastPose, &curPose))
    {
      *_lastPose = curPose;
      if(_synchronous)
      {
        // what queuePush + the mapping thread do, in stream order on this thread
        std::unique_ptr<obvious::SensorPolar2D> local(_sensor->copyForMapping());
        _grid.push(local.get());
      }
      else
        _mapper.queuePush(_sensor);
      rep.pushed = true;
    }
  }
  std::lock_guard<std::mutex> lk(_reportMutex);
  _report = rep; _processed++;
}

// The event-loop body with the device doing everything between scan ingest and pose publication
// (tsd_scan): ray cast, registration, isRegistrationError, Sensor::transform, isPoseChangeSignificant and
// the push that the mapping thread would do, in stream order.  The host sensor is kept as a mirror.
void ThreadLocalize::processScanFused(ScanReport& rep)
{
  // mask of the copy ThreadMapping::queuePush would make (ThreadMapping.cpp:65-76, Appendix B #20)
  std::vector<uint8_t> maskPush;
  _sensor->maskForMapping(maskPush);
  tsd_gate_params gates = {_trnsMax, _rotMax, TRNS_MIN, ROT_MIN};
  tsd_scan_result sr;
  std::memset(&sr, 0, sizeof(sr));
  int rc;
  if(_concurrent)
  {
    // several robots on one grid: the grid's dispatcher batches the robots' scans (or the split scan)
    rc = TSD_OK;
    if(_regMode == 3 && _preMatcher && _preFusedOk && _grid.batcher())
    {
      // doRegistration, case TSD (ThreadLocalize.cpp:557-567), armed for this robot's scan of the batch: scene points + the three
      // rand() streams from the host, everything else on the device behind the batch's ray casts (tsd_batch_begin)
      const unsigned int n = _sensor->getRealMeasurementSize();
      if(_scene.size() != 2 * (size_t)n) { _scene.assign(2 * (size_t)n, 0.0); _maskS.assign(n, 0); }
      bool* maskS = reinterpret_cast<bool*>(_maskS.data());
      _sensor->dataToCartesianVectorMask(_scene.data(), maskS);
      _preMatcher->drawStreams(n, _dSub, _dCtrl, _dTrials);
      rc = _grid.scanPreregister(_sensor, _preMatcher->params(_ranPhiMax * M_PI / 180.0, _sensor->getAngularResolution()), _scene.data(), maskS,
                                 _dSub.data(), _dCtrl.data(), _dTrials.data());
      if(rc == TSD_E_CAPACITY)
      {
        std::fprintf(stderr, "Localizer(%s): registration_mode 3 runs unfused (%s)\n", _nameSpace.c_str(), tsd_last_error(_grid.context()));
        _preFusedOk = false;
        processScanPreRegistered(rep);
        return;
      }
    }
    if(rc == TSD_OK) rc = _grid.scanConcurrent(_sensor, maskPush.data(), _icpParams, gates, &sr);
  }
  else
  {
    const bool useStaged = _stagedValid;
    _stagedValid = false;
    const bool preStaged = _preStagedValid && useStaged;      // (armed while the previous scan registered; a dropped staged scan takes it along)
    // ... but not its draws: they do not depend on the scan, and the scan that came instead uses the very same ones, so that a seeded
    // sequence (tsdpdf_seed) stays one drawStreams() call per REGISTERED scan, as in the unfused path
    if(_preStagedValid && !useStaged) _drawsReady = true;
    _preStagedValid = false;
    rc = TSD_OK;
    if(_regMode == 3 && _preMatcher && !preStaged)
    {
      // doRegistration, case TSD (ThreadLocalize.cpp:557-567): the scene points and the match's three rand() streams are all the
      // pre-registration needs from the host; its model is the ray cast's output, on the device
      const unsigned int n = _sensor->getRealMeasurementSize();
      if(_scene.size() != 2 * (size_t)n) { _scene.assign(2 * (size_t)n, 0.0); _maskS.assign(n, 0); }
      bool* maskS = reinterpret_cast<bool*>(_maskS.data());
      _sensor->dataToCartesianVectorMask(_scene.data(), maskS);
      // (the draws do not depend on the scan: they were taken from rand() while the previous scan registered -- same values, same
      // order, as long as this localiser is the process's only rand() user, which it is in the reference's single-robot node too)
      if(!_drawsReady || _dSub.size() != n) _preMatcher->drawStreams(n, _dSub, _dCtrl, _dTrials);
      _drawsReady = false;
      rc = _grid.scanPreregister(_sensor, _preMatcher->params(_ranPhiMax * M_PI / 180.0, _sensor->getAngularResolution()), _scene.data(), maskS,
                                 _dSub.data(), _dCtrl.data(), _dTrials.data());
      if(rc == TSD_E_CAPACITY)
      {
        // more trials / control points than the device-side list building holds in LDS: the reference's call structure instead,
        // from this scan on (nothing was enqueued; the draws are consumed, like a match() that ran)
        std::fprintf(stderr, "Localizer(%s): registration_mode 3 runs unfused (%s)\n", _nameSpace.c_str(), tsd_last_error(_grid.context()));
        _preFusedOk = false;
        processScanPreRegistered(rep);
        return;
      }
    }
    if(rc == TSD_OK) rc = _grid.scanSubmit(_sensor, useStaged, maskPush.data(), _icpParams, gates);
    // the next scan, if it is known already: ingest + copy + tables while the device registers this one
    std::shared_ptr<sensor_msgs::msg::LaserScan> next;
    {
      std::lock_guard<std::mutex> lk(_dataMutex);
      next.swap(_ahead);
      if(!next && !_synchronous && !_laserData.empty())
      {
        // threaded mode: the newest scan that queued up meanwhile is what the event loop takes next -- unless a newer one
        // still arrives, in which case the staged one is dropped like the reference drops it (ThreadLocalize.cpp:319-332)
        next = std::make_shared<sensor_msgs::msg::LaserScan>(*_laserData.front());
        if(_reverseScan)
          std::reverse(next->ranges.begin(), next->ranges.end());
      }
    }
    if(rc == TSD_OK && next && next->ranges.size() == _sensor->getRealMeasurementSize())
    {
      _sensor->setRealMeasurementData(next->ranges);
      _sensor->setStandardMask();
      std::vector<uint8_t> maskNext;
      _sensor->maskForMapping(maskNext);
      if(_grid.scanStage(_sensor, maskNext.data()) == TSD_OK)
      {
        _stagedValid = true;
        _stagedStampNs = (long long)next->header.stamp.sec * 1000000000LL + (long long)next->header.stamp.nanosec;
        _stagedRanges = next->ranges;
        if(_regMode == 3 && _preMatcher && _preFusedOk)
        {
          // ... and its pre-registration: scene points, draws and the inputs' copy NOW, while the device registers this scan -- between
          // two scans the host then only submits (the device was waiting for the host there: 40 us of host work per scan in mode 3)
          const unsigned int n = _sensor->getRealMeasurementSize();
          if(_scene.size() != 2 * (size_t)n) { _scene.assign(2 * (size_t)n, 0.0); _maskS.assign(n, 0); }
          bool* maskS = reinterpret_cast<bool*>(_maskS.data());
          _sensor->dataToCartesianVectorMask(_scene.data(), maskS);
          if(!_drawsReady || _dSub.size() != n) _preMatcher->drawStreams(n, _dSub, _dCtrl, _dTrials);
          _drawsReady = false;
          if(_grid.scanPreregister(_sensor, _preMatcher->params(_ranPhiMax * M_PI / 180.0, _sensor->getAngularResolution()), _scene.data(), maskS,
                                   _dSub.data(), _dCtrl.data(), _dTrials.data()) == TSD_OK)
            _preStagedValid = true;
        }
      }
    }
    if(rc == TSD_OK && _regMode == 3 && _preMatcher && _preFusedOk && !_drawsReady && !_preStagedValid)
    {
      _preMatcher->drawStreams(_sensor->getRealMeasurementSize(), _dSub, _dCtrl, _dTrials);      // the next scan's, while the device is busy
      _drawsReady = true;
    }
    if(rc == TSD_OK) rc = _grid.scanCollect(_sensor, &sr);
  }
  _sensor->getTransformation().getData(rep.pose);
  if(rc != TSD_OK)
  {
    std::fprintf(stderr, "Localizer(%s) device error %d\n", _nameSpace.c_str(), rc);
    std::lock_guard<std::mutex> lk(_reportMutex);
    _report = rep; _processed++;
    return;
  }
  rep.validModel = sr.icp.n_model; rep.validScene = sr.icp.n_scene;
  if(sr.no_model)
  {
    rep.noModel = true;
    std::lock_guard<std::mutex> lk(_reportMutex);
    _report = rep; _processed++;
    return;
  }
  obvious::Matrix T(3, 3, sr.icp.T);
  std::memcpy(rep.T, sr.icp.T, sizeof(rep.T));
  rep.rms = sr.icp.rms; rep.pairs = sr.icp.pairs; rep.iterations = sr.icp.iterations; rep.icpState = sr.icp.state;
  if(sr.reg_error)
  {
    rep.regError = true;
    sendNanTransform();
  }
  else
  {
    _sensor->transform(&T);                          // host mirror: same arithmetic as the device
    obvious::Matrix devPose(3, 3, sr.pose);
    _sensor->setTransformation(devPose);             // the device pose is the authoritative one
    obvious::Matrix curPose = _sensor->getTransformation();
    curPose.getData(rep.pose);
    sendTransform(&curPose);
    if(sr.pushed)
    {
      *_lastPose = curPose;
      rep.pushed = true;
    }
  }
  std::lock_guard<std::mutex> lk(_reportMutex);
  _report = rep; _processed++;
}

void ThreadLocalize::init(const sensor_msgs::msg::LaserScan& scan)
{
  // per-robot parameters (ThreadLocalize.cpp:424-442)
  declareInitParameters(_node, _nameSpace);
  const double localXoffset   = _node->get_parameter(_nameSpace + "local_offset_x").as_double();
  const double localYoffset   = _node->get_parameter(_nameSpace + "local_offset_y").as_double();
  const double localYawOffset = _node->get_parameter(_nameSpace + "local_offset_yaw").as_double();
  const double maxRange       = _node->get_parameter(_nameSpace + "max_range").as_double();
  const double minRange       = _node->get_parameter(_nameSpace + "min_range").as_double();
  const double lowReflectivityRange = _node->get_parameter(_nameSpace + "low_reflectivity_range").as_double();
  const double footPrintWidth   = _node->get_parameter(_nameSpace + "footprint_width").as_double();
  const double footPrintHeight  = _node->get_parameter(_nameSpace + "footprint_height").as_double();
  const double footPrintXoffset = _node->get_parameter(_nameSpace + "footprint_x_offset").as_double();

  const double phi    = localYawOffset;
  const double startX = _gridWidth * 0.5 + _xOffset + localXoffset;
  const double startY = _gridHeight * 0.5 + _yOffset + localYoffset;
  double tf[9] = {std::cos(phi), -std::sin(phi), startX,
                  std::sin(phi),  std::cos(phi), startY,
                  0,              0,             1};
  obvious::Matrix Tinit(3, 3);
  Tinit.setData(tf);

  double inc       = scan.angle_increment;
  double angle_min = scan.angle_min;
  std::vector<float> ranges = scan.ranges;
  if(scan.angle_increment < 0.0 && scan.angle_min > 0)   // clockwise scanner (:491-497)
  {
    _reverseScan = true;
    inc       = -inc;
    angle_min = -angle_min;
    std::reverse(ranges.begin(), ranges.end());
  }
  _sensor = new obvious::SensorPolar2D(ranges.size(), inc, angle_min, maxRange, minRange, lowReflectivityRange);
  _sensor->setRealMeasurementData(ranges, 1.0);
  _sensor->setStandardMask();
  _sensor->transform(&Tinit);
  double t[2] = {startX + footPrintXoffset, startY};
  if(!_grid.freeFootprint(t, footPrintWidth, footPrintHeight))
    std::fprintf(stderr, "Localizer (%s) warning! Footprint could not be freed!\n", _nameSpace.c_str());
  bool pushed = false;
  if(!_mapper.initialized())
  {
    _mapper.initPush(_sensor);
    pushed = true;
  }
  if(_fused && _grid.attachSensor(_sensor) != TSD_OK)
    std::fprintf(stderr, "Localizer (%s): no device sensor, using the unfused path\n", _nameSpace.c_str());
  else if(_fused && _asyncMapping && !_concurrent && _sensor->deviceHandle())
  {
    std::lock_guard<std::mutex> lk(_grid.mutex());
    if(tsd_sensor_set_async_mapping(_sensor->deviceHandle(), 1) != TSD_OK)
      std::fprintf(stderr, "Localizer (%s): asynchronous mapping not available (%s)\n", _nameSpace.c_str(), tsd_last_error(_grid.context()));
  }
  _initialized = true;
  {
    std::lock_guard<std::mutex> lk(_reportMutex);
    std::memset(&_report, 0, sizeof(_report));
    _sensor->getTransformation().getData(_report.pose);
    _report.T[0] = _report.T[4] = _report.T[8] = 1.0;
    _report.pushed = pushed; _report.initialised = true;
    _report.stampNs = (long long)scan.header.stamp.sec * 1000000000LL + (long long)scan.header.stamp.nanosec;
    _processed++;
  }
  this->unblock();
}

bool ThreadLocalize::isRegistrationError(obvious::Matrix* T, const double trnsMax, const double rotMax)
{
  const double deltaX   = (*T)(0, 2);
  const double deltaY   = (*T)(1, 2);
  const double trnsAbs  = std::sqrt(deltaX * deltaX + deltaY * deltaY);
  const double deltaPhi = calcAngle(T);
  return (trnsAbs > trnsMax) || (std::abs(std::sin(deltaPhi)) > rotMax);
}

// ThreadLocalize.cpp:603-689.  What leaves on tf is map -> odom: the laser pose in the map, taken to base_footprint with the
// laser -> base_footprint look-up and on to odom with base_footprint -> odom, i.e. pose * T_laser_footprint * T_footprint_odom.
// A look-up that fails (no such frames yet: the reference catches tf2::TransformException) is skipped like the reference skips it, and -- also like the reference -- the message's
// transform is written ONLY where the odom look-up succeeded (:657): without an odom tree the broadcaster repeats whatever _tf.transform
// held (the identity at start, the last good correction later, NaN after sendNanTransform).  The PoseStamped is always the laser pose.
void ThreadLocalize::sendTransform(obvious::Matrix* T)
{
  const double curTheta = calcAngle(T);
  const double posX = (*T)(0, 2) + _gridOffSetX;
  const double posY = (*T)(1, 2) + _gridOffSetY;

  tf2::Quaternion orientation;
  orientation.setEuler(0.0, 0.0, curTheta);
  tf2::Transform pose;
  pose.setOrigin(tf2::Vector3(posX, posY, 0.0));
  pose.setRotation(orientation);
  _tf.child_frame_id = _tfOdomFrameId;
  _tf.header.frame_id = _tfMapFrameId;

  // pose <- pose * lookup(target, source); false when the buffer has no such transform
  auto compose = [&](const std::string& target, const std::string& source, int which) -> bool
  {
    // (asked first with tf2::BufferCore::canTransform, which answers "no" without throwing: a robot without a tf tree -- the bench's
    // case -- paid two C++ exceptions per scan on this path for the answer the reference's catch blocks turn into "skip"; the
    // look-up itself stays guarded, the tree may lose the edge between the two calls)
    int state = 1;
    std::string why;
    if(!_tf_buffer->canTransform(target, source, tf2::TimePointZero, &why))
      state = 0;
    else
    {
      try
      {
        const geometry_msgs::msg::TransformStamped st = _tf_buffer->lookupTransform(target, source, tf2::TimePointZero);
        tf2::Transform step, product;
        tf2::fromMsg(st.transform, step);
        product.mult(pose, step);
        pose = product;
      }
      catch(const tf2::TransformException& ex)
      {
        state = 0;
        why = ex.what();
      }
    }
    if(state != _tfLookUpState[which])      // (the reference logs this at INFO level on every scan; here: when it changes)
    {
      _tfLookUpState[which] = state;
      if(!state)
        std::fprintf(stderr, "Localizer(%s): no transform from %s to %s available (%s)\n", _nameSpace.c_str(), source.c_str(), target.c_str(), why.c_str());
    }
    return state != 0;
  };
  compose(_tfLaserFrameId, _tfFootprintFrameId, 0);                 // correction of laser to base_footprint (:618-641)
  if(compose(_tfFootprintFrameId, _tfOdomFrameId, 1))               // correction of odom (:643-666)
  {
    _tf.child_frame_id = _tfOdomFrameId;
    _tf.header.frame_id = _tfMapFrameId;
    _tf.transform = tf2::toMsg(pose);
  }

  _poseStamped.header.stamp    = _stampLaser;
  _poseStamped.pose.position.x = posX;
  _poseStamped.pose.position.y = posY;
  _poseStamped.pose.position.z = 0.0;
  setYaw(_poseStamped.pose.orientation, curTheta);
  _tf.header.stamp = _stampLaser;

  _posePub->publish(_poseStamped);
  _tfBroadcaster->sendTransform(_tf);
}

void ThreadLocalize::sendNanTransform()
{
  const double nan = std::numeric_limits<double>::quiet_NaN();
  _poseStamped.header.stamp = _node->get_clock()->now();
  _poseStamped.pose.position.x = nan;
  _poseStamped.pose.position.y = nan;
  _poseStamped.pose.position.z = nan;
  _poseStamped.pose.orientation.w = nan;
  _poseStamped.pose.orientation.x = nan;
  _poseStamped.pose.orientation.y = nan;
  _poseStamped.pose.orientation.z = nan;
  _tf.header.stamp = _node->get_clock()->now();
  _tf.transform.translation.x = nan;
  _tf.transform.translation.y = nan;
  _tf.transform.translation.z = nan;
  _tf.transform.rotation = _poseStamped.pose.orientation;
  _posePub->publish(_poseStamped);
  _tfBroadcaster->sendTransform(_tf);
}

double ThreadLocalize::calcAngle(obvious::Matrix* T)
{
  double angle          = 0.0;
  const double ARCSIN   = std::asin((*T)(1, 0));
  const double ARCSINEG = std::asin((*T)(0, 1));
  const double ARCOS    = std::acos((*T)(0, 0));
  if((ARCSIN > 0.0) && (ARCSINEG < 0.0))
    angle = ARCOS;
  else if((ARCSIN < 0.0) && (ARCSINEG > 0.0))
    angle = 2.0 * M_PI - ARCOS;
  return(angle);
}

bool ThreadLocalize::isPoseChangeSignificant(obvious::Matrix* lastPose, obvious::Matrix* curPose)
{
  const double deltaX = (*curPose)(0, 2) - (*lastPose)(0, 2);
  const double deltaY = (*curPose)(1, 2) - (*lastPose)(1, 2);
  double deltaPhi     = calcAngle(curPose) - calcAngle(lastPose);
  deltaPhi            = std::fabs(std::sin(deltaPhi));
  const double trnsAbs = std::sqrt(deltaX * deltaX + deltaY * deltaY);
  return(deltaPhi > ROT_MIN || trnsAbs > TRNS_MIN);
}

ThreadLocalize::ScanReport ThreadLocalize::lastReport()
{
  std::lock_guard<std::mutex> lk(_reportMutex);
  return _report;
}

bool ThreadLocalize::idle()
{
  std::lock_guard<std::mutex> lk(_dataMutex);
  return _laserData.empty() && !_busy;
}

uint64_t ThreadLocalize::processedScans()
{
  std::lock_guard<std::mutex> lk(_reportMutex);
  return _processed;
}

} /* namespace ohm_tsd_slam */
