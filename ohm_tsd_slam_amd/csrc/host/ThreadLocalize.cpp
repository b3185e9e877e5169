#include "ThreadLocalize.h"
#include "ThreadMapping.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
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

  // parameters: names and defaults of SlamNode.cpp:40-58 / ThreadLocalize.cpp:86-129 (Appendix D)
  const double distFilterMax = param(node, _robotName + "dist_filter_max", 1.0).as_double();
  const double distFilterMin = param(node, _robotName + "dist_filter_min", 0.1).as_double();
  const int icpIterations    = (int)param(node, _robotName + "icp_iterations", 25).as_int();
  _tfLaserFrameId     = param(node, _robotName + "tf_laser_frame", std::string("laser")).as_string();
  _tfMapFrameId       = param(node, "tf_map_frame", std::string("map")).as_string();
  _tfOdomFrameId      = param(node, _robotName + "tf_odom_frame", std::string("odom")).as_string();
  _tfFootprintFrameId = param(node, _robotName + "tf_footprint_frame", std::string("base_footprint")).as_string();
  _trnsMax     = param(node, "reg_trs_max", 0.25).as_double();
  _rotMax      = param(node, "reg_sin_rot_max", 0.17).as_double();
  (void)param(node, "max_velocity_lin", 1.5);
  (void)param(node, "max_velocity_rot", 2.0 * M_PI);
  (void)param(node, "ude_odom_rescue", false);
  (void)param(node, "wait_for_odom_tf", 1.0);
  _lasMinRange = param(node, "laser_min_range", 0.0).as_double();
  _regMode     = (int)param(node, _robotName + "registration_mode", 0).as_int();
  if(_regMode != 0)
  {
    std::fprintf(stderr, "Localizer(%s): registration mode %d (wall-clock seeded RANSAC pre-registration) is not part of "
                         "the GPU hot path; using default = ICP.\n", _nameSpace.c_str(), _regMode);
    _regMode = 0;
  }

  // ICP set-up (ThreadLocalize.cpp:211-225): DistanceFilter(max, min, icpIterations - 10), bounds filter
  // over the grid extent, maxRMS 0, max iterations == convergence counter == icpIterations
  _icpParams.iterations      = icpIterations;
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

  _sensor->setRealMeasurementData(ranges);
  _sensor->setStandardMask();

  if(!_haveLastPose)   // first call (ThreadLocalize.cpp:342-350)
  {
    *_lastPose = _sensor->getTransformation();
    _haveLastPose = true;
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
  obvious::Matrix T(3, 3, res.T);
  std::memcpy(rep.T, res.T, sizeof(rep.T));
  rep.rms = res.rms; rep.pairs = res.pairs; rep.iterations = res.iterations; rep.icpState = res.state;

  const bool regErrorT = isRegistrationError(&T, _trnsMax, _rotMax);
  if(regErrorT)
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
  const int rc = _grid.scan(_sensor, maskPush.data(), _icpParams, gates, &sr);
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
  const double localXoffset   = param(_node, _nameSpace + "local_offset_x", 0.0).as_double();
  const double localYoffset   = param(_node, _nameSpace + "local_offset_y", 0.0).as_double();
  const double localYawOffset = param(_node, _nameSpace + "local_offset_yaw", 0.0).as_double();
  const double maxRange       = param(_node, _nameSpace + "max_range", 30.0).as_double();
  const double minRange       = param(_node, _nameSpace + "min_range", 0.001).as_double();
  const double lowReflectivityRange = param(_node, _nameSpace + "low_reflectivity_range", 2.0).as_double();
  const double footPrintWidth   = param(_node, _nameSpace + "footprint_width", 1.0).as_double();
  const double footPrintHeight  = param(_node, _nameSpace + "footprint_height", 1.0).as_double();
  const double footPrintXoffset = param(_node, _nameSpace + "footprint_x_offset", 0.28).as_double();

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

void ThreadLocalize::sendTransform(obvious::Matrix* T)
{
  const double curTheta = calcAngle(T);
  const double posX = (*T)(0, 2) + _gridOffSetX;
  const double posY = (*T)(1, 2) + _gridOffSetY;

  // without a tf tree (laser->base_footprint, base_footprint->odom look-ups throw in the reference and
  // are skipped, ThreadLocalize.cpp:617-661) map->odom carries the laser pose itself
  _tf.child_frame_id = _tfOdomFrameId;
  _tf.header.frame_id = _tfMapFrameId;
  _tf.transform.translation.x = posX;
  _tf.transform.translation.y = posY;
  _tf.transform.translation.z = 0.0;
  setYaw(_tf.transform.rotation, curTheta);

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
