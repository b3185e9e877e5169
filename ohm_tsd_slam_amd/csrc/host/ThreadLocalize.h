// ThreadLocalize.h -- per-robot localisation worker; public surface of the reference's
// ThreadLocalize (src/ThreadLocalize.h:73-104): same constructor, same laserCallBack entry point, same
// parameters, topic and message content.  The arithmetic of eventLoop (ray cast, ICP) runs on the GPU
// through obvious::TsdGrid::localize; scan ingest, gates and pose bookkeeping stay on the host.
// registration_mode 0 (ICP only) and 3 (TSD_PDF pre-registration + ICP, what config/single-laser.yaml ships) are
// implemented; modes 1 / 2 (RandomNormalMatching / PDFMatching, SURVEY 2: out of scope) fall back to 0 with a warning,
// as the reference does for unknown modes (ThreadLocalize.cpp:188-190).
#pragma once
#include <deque>
#include <memory>
#include <string>

#include "ThreadSLAM.h"
#include "ros_shim.h"

namespace ohm_tsd_slam
{

class ThreadMapping;

namespace
{
  const double TRNS_MIN = 0.05;   // minimal pose change that triggers a push (ThreadLocalize.h:63-64)
  const double ROT_MIN  = 0.03;
}

class ThreadLocalize: public ThreadSLAM
{
public:
  ThreadLocalize(obvious::TsdGrid* grid, ThreadMapping* mapper, const std::shared_ptr<rclcpp::Node>& node,
                 const std::string& robot_name, const double xOffset, const double yOffset);
  virtual ~ThreadLocalize();

  /** laser subscription callback (ThreadLocalize.cpp:248-276): first scan initialises synchronously,
   *  later scans are queued for the localisation thread (newest wins) */
  void laserCallBack(const std::shared_ptr<sensor_msgs::msg::LaserScan> scan);

  // ---- additions for tests / benchmarks (not in the reference) ----
  /** run the event-loop body on the caller's thread and push in stream order instead of handing
   *  over to the two worker threads: the strict R -> I -> P order the parity tests need */
  void setSynchronous(bool on) { _synchronous = on; }
  /** fused scan path (default): sensor state on the device, gates + push decided there in stream
   *  order (tsd_scan).  Off: tsd_localize, host gates, ThreadMapping::queuePush as in the reference. */
  void setFused(bool on) { _fused = on; }
  /** this grid is shared with other localisers (SlamNode's multi-robot mode): use the split scan of the fused path
   *  (tsd_scan_begin / _wait / _finish) so that the robots' registrations overlap on the device */
  void setConcurrent(bool on) { _concurrent = on; }
  /** the scan that will come NEXT is known already (queued behind the one being processed; a replayed log): the fused path
   *  ingests it and stages it on the device while the current registration runs (tsd_scan_stage).  The following
   *  laserCallBack with the same stamp then starts from the staged data. */
  void announceNext(const std::shared_ptr<sensor_msgs::msg::LaserScan> scan);
  struct ScanReport {
    double pose[9]; double T[9]; double rms; int pairs; int iterations; int icpState;
    int validModel; int validScene; bool regError; bool pushed; bool noModel; bool initialised;
    long long stampNs;
  };
  ScanReport lastReport();
  uint64_t processedScans();
  /** nothing queued and nothing being processed */
  bool idle();
  std::shared_ptr<rclcpp::Publisher<geometry_msgs::msg::PoseStamped>> posePublisher() { return _posePub; }
  /** the tf buffer sendTransform looks laser -> base_footprint and base_footprint -> odom up in (a TransformListener fills it under
   *  ROS; tests and ROS-free hosts feed it with Buffer::setTransform) and the broadcaster map -> odom leaves through */
  tf2_ros::Buffer* tfBuffer() { return _tf_buffer.get(); }
  tf2_ros::TransformBroadcaster* tfBroadcaster() { return _tfBroadcaster.get(); }
  obvious::SensorPolar2D* sensor() { return _sensor; }

  // every parameter the reference declares in the constructor (ThreadLocalize.cpp:86-129) / in init (:424-432)
  static void declareParameters(const std::shared_ptr<rclcpp::Node>& node, const std::string& robotName);
  static void declareInitParameters(const std::shared_ptr<rclcpp::Node>& node, const std::string& nameSpace);

  // gates (static: also unit-tested against the oracle)
  static double calcAngle(obvious::Matrix* T);                                               // :715-726
  static bool isRegistrationError(obvious::Matrix* T, const double trnsMax, const double rotMax);   // :593-600
  static bool isPoseChangeSignificant(obvious::Matrix* lastPose, obvious::Matrix* curPose);   // :728-736

protected:
  virtual void eventLoop(void);

private:
  void init(const sensor_msgs::msg::LaserScan& scan);
  void processScan(const std::vector<float>& rangesIn, const builtin_interfaces::msg::Time& stamp);
  void processScanFused(ScanReport& rep);
  void processScanPreRegistered(ScanReport& rep);
  void finishScan(ScanReport& rep, const tsd_icp_result& res);
  void sendTransform(obvious::Matrix* T);
  void sendNanTransform();

  std::shared_ptr<rclcpp::Node> _node;
  ThreadMapping& _mapper;
  obvious::SensorPolar2D* _sensor;
  bool _initialized;
  bool _synchronous;
  bool _fused;
  bool _concurrent;
  const double _gridWidth, _gridHeight, _gridOffSetX, _gridOffSetY, _xOffset, _yOffset;
  std::string _robotName, _nameSpace;
  std::string _tfMapFrameId, _tfOdomFrameId, _tfLaserFrameId, _tfFootprintFrameId;
  double _trnsMax, _rotMax, _lasMinRange;
  int _regMode;
  double _ranPhiMax;
  std::unique_ptr<obvious::TSD_PDFMatching> _preMatcher;      // _TSD_PDFMatcher (ThreadLocalize.h)
  std::vector<double> _modelCoords, _modelNormals, _scene;     // beam-indexed buffers of the event loop (:342-350)
  std::vector<uint8_t> _maskM, _maskS;
  bool _reverseScan;
  tsd_icp_params _icpParams;
  obvious::Matrix* _lastPose;
  bool _haveLastPose;
  builtin_interfaces::msg::Time _stampLaser, _stampLaserOld;

  std::shared_ptr<sensor_msgs::msg::LaserScan> _ahead;      // announced next scan (clamped like laserCallBack does)
  bool _stagedValid = false;                                // _sensor holds, and the device has staged, the scan with ...
  bool _asyncMapping = false;       // "async_mapping" (addition): the fused scan's push beside the next registration (tsd_sensor_set_async_mapping)
  bool _preFusedOk = true;          // registration_mode 3: the device-side pre-registration takes these parameters (else: the unfused calls)
  // registration_mode 3, fused: the NEXT scan's three rand() streams, drawn while the device registers this one (1 311 rand() calls
  // are 23 us of host time -- more than the host has between two scans before the device runs dry)
  std::vector<int> _dSub, _dCtrl, _dTrials;
  bool _drawsReady = false;
  bool _preStagedValid = false;     // the staged scan's pre-registration is armed on the device already (tsd_scan_preregister ahead of the collect)
  long long _stagedStampNs = 0;                             // ... this stamp ...
  std::vector<float> _stagedRanges;                         // ... and these readings (in the sensor's beam order)
  std::deque<std::shared_ptr<sensor_msgs::msg::LaserScan>> _laserData;
  std::mutex _dataMutex;
  bool _busy;

  std::shared_ptr<rclcpp::Publisher<geometry_msgs::msg::PoseStamped>> _posePub;
  std::unique_ptr<tf2_ros::TransformBroadcaster> _tfBroadcaster;
  std::unique_ptr<tf2_ros::Buffer> _tf_buffer;                          // ThreadLocalize.h:403-404
  std::unique_ptr<tf2_ros::TransformListener> _tf_transform_listener;
  int _tfLookUpState[2] = {-1, -1};                                     // last outcome of the two look-ups (logged on change only)
  geometry_msgs::msg::PoseStamped _poseStamped;
  geometry_msgs::msg::TransformStamped _tf;

  std::mutex _reportMutex;
  ScanReport _report;
  uint64_t _processed;
};

} /* namespace ohm_tsd_slam */
