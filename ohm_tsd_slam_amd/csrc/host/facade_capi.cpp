// facade_capi.cpp -- a flat C surface over the C++ facade so that the Python test / bench drivers can
// exercise ThreadLocalize / ThreadMapping exactly as SlamNode wires them (SlamNode.cpp:27-129): one
// TsdGrid, one ThreadMapping, N ThreadLocalize sharing them, laser callbacks per robot.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "ThreadLocalize.h"
#include "ThreadMapping.h"

using namespace ohm_tsd_slam;

struct tsd_node
{
  std::shared_ptr<rclcpp::Node> node;
  obvious::TsdGrid* grid = nullptr;
  ThreadMapping* mapping = nullptr;
  std::vector<ThreadLocalize*> localizers;
  std::vector<uint64_t> submitted;
  bool synchronous = false;
  bool fused = true;
};

extern "C" {

tsd_node* tsd_node_create(const char* node_name)
{
  tsd_node* n = new tsd_node();
  n->node = std::make_shared<rclcpp::Node>(node_name ? node_name : "tsd_slam");
  return n;
}

void tsd_node_set_double(tsd_node* n, const char* name, double v) { n->node->set_parameter(name, v); }
void tsd_node_set_int(tsd_node* n, const char* name, int v) { n->node->set_parameter(name, v); }
void tsd_node_set_bool(tsd_node* n, const char* name, int v) { n->node->set_parameter(name, v != 0); }
void tsd_node_set_string(tsd_node* n, const char* name, const char* v) { n->node->set_parameter(name, std::string(v)); }

// the parameters SlamNode::initialize declares (SlamNode.cpp:40-58)
static void declare_node_parameters(const std::shared_ptr<rclcpp::Node>& node)
{
  node->declare_parameter("robot_nbr", 1);
  node->declare_parameter("x_off_factor", 0.5);
  node->declare_parameter("y_off_factor", 0.5);
  node->declare_parameter("x_offset", 0.0);
  node->declare_parameter("y_offset", 0.0);
  node->declare_parameter("map_size", 10);
  node->declare_parameter("cellsize", 0.025);
  node->declare_parameter("truncation_radius", 3);
  node->declare_parameter("occ_grid_time_interval", 2.0);
  node->declare_parameter("tf_map_frame", std::string("map"));
  // ThreadGrid's (ThreadGrid.cpp:42-47): the occupancy thread itself is out of scope, its extraction kernels (row N1) take
  // these as arguments; declared so that the shipped YAMLs (config/single-laser.yaml) load as they are
  node->declare_parameter("pub_tsd_color_map", true);
  node->declare_parameter("object_inflation_factor", 2);
  node->declare_parameter("use_object_inflation", false);
}

#if !OHM_TSD_SLAM_HAVE_ROS
// Parameter surface without a device (CPU test): what SlamNode::initialize, every ThreadLocalize constructor and
// every ThreadLocalize::init would declare, as "name|type|default" lines.  Returns the bytes needed.
int tsd_node_declared_parameters(tsd_node* n, char* buf, int cap)
{
  auto& node = n->node;
  declare_node_parameters(node);
  const unsigned robotNbr = (unsigned)node->get_parameter("robot_nbr").as_int();
  const std::string node_name = std::string(node->get_name()) + "/";
  for(unsigned i = 0; i < robotNbr; i++)
  {
    std::string robot = "";
    if(robotNbr > 1)
    {
      const std::string key = "robot_" + std::to_string(i) + "/name";
      node->declare_parameter(key, "robot_" + std::to_string(i));
      robot = node->get_parameter(key).as_string();
      if(robot.size() > 0 && robot.back() != '/') robot += "/";
    }
    ThreadLocalize::declareParameters(node, robot);
    ThreadLocalize::declareInitParameters(node, node_name + robot);
  }
  std::string out;
  for(const auto& kv : node->declared_parameters())
  {
    const auto& v = kv.second.value();
    char num[64];
    if(std::holds_alternative<bool>(v)) out += kv.first + "|bool|" + (std::get<bool>(v) ? "true" : "false") + "\n";
    else if(std::holds_alternative<int64_t>(v)) out += kv.first + "|int|" + std::to_string(std::get<int64_t>(v)) + "\n";
    else if(std::holds_alternative<double>(v)) { std::snprintf(num, sizeof(num), "%.17g", std::get<double>(v)); out += kv.first + "|double|" + num + "\n"; }
    else out += kv.first + "|string|" + std::get<std::string>(v) + "\n";
  }
  if(buf && cap > 0)
  {
    std::strncpy(buf, out.c_str(), (size_t)cap - 1);
    buf[cap - 1] = 0;
  }
  return (int)out.size() + 1;
}
#endif

// SlamNode::initialize (SlamNode.cpp:40-122): grid parameters, grid, mapping thread, localisers
int tsd_node_initialize(tsd_node* n, int device)
{
  auto& node = n->node;
  declare_node_parameters(node);
  const unsigned robotNbr = (unsigned)node->get_parameter("robot_nbr").as_int();
  const double xOffset = node->get_parameter("x_offset").as_double();
  const double yOffset = node->get_parameter("y_offset").as_double();
  unsigned octaveFactor = (unsigned)node->get_parameter("map_size").as_int();
  const double cellSize = node->get_parameter("cellsize").as_double();
  const double truncationRadius = (double)node->get_parameter("truncation_radius").as_int();
  if(octaveFactor > 15)
    octaveFactor = 10;   // "Unknown map size -> set to default" (SlamNode.cpp:71-75)

  n->grid = new obvious::TsdGrid(cellSize, obvious::LAYOUT_32x32, static_cast<obvious::EnumTsdGridLayout>(octaveFactor), device);
  if(!n->grid->valid())
    return TSD_E_NODEVICE;
  n->grid->setMaxTruncation(truncationRadius * cellSize);
  n->mapping = new ThreadMapping(n->grid);
  if(robotNbr == 1)
  {
    n->localizers.push_back(new ThreadLocalize(n->grid, n->mapping, node, "", xOffset, yOffset));
  }
  else
  {
    for(unsigned i = 0; i < robotNbr; i++)
    {
      const std::string key = "robot_" + std::to_string(i) + "/name";
      node->declare_parameter(key, "robot_" + std::to_string(i));
      const std::string robot_name = node->get_parameter(key).as_string();
      n->localizers.push_back(new ThreadLocalize(n->grid, n->mapping, node, robot_name, xOffset, yOffset));
    }
  }
  n->submitted.assign(n->localizers.size(), 0);
  // several robots on one grid: their scans go through the grid's dispatcher as batched launches (TSD_NO_BATCH: the split
  // scan on one stream per robot instead; TSD_BATCH_SLOTS: batch slots used in turn, default 2 -- A/B switches for measurements)
  if(n->localizers.size() > 1 && !std::getenv("TSD_NO_CONCURRENT") && !std::getenv("TSD_NO_BATCH"))
  {
    const char* e = std::getenv("TSD_BATCH_SLOTS");
    n->grid->enableBatchedScans((int)n->localizers.size(), e ? std::atoi(e) : 2);
  }
  for(auto* l : n->localizers)
  {
    l->setSynchronous(n->synchronous);
    l->setFused(n->fused);
    l->setConcurrent(n->localizers.size() > 1 && !std::getenv("TSD_NO_CONCURRENT"));   // (env: A/B switch for tests / measurements)
  }
  return TSD_OK;
}

// before the first scan: choose the fused device scan (default) or the reference's three-call flow
void tsd_node_set_fused(tsd_node* n, int on)
{
  n->fused = on != 0;
  for(auto* l : n->localizers)
    l->setFused(n->fused);
}

void tsd_node_set_synchronous(tsd_node* n, int on)
{
  n->synchronous = on != 0;
  for(auto* l : n->localizers)
    l->setSynchronous(n->synchronous);
}

int tsd_node_laser(tsd_node* n, int robot, const float* ranges, int count, double angle_min,
                   double angle_increment, long long stamp_ns)
{
  if(!n || robot < 0 || robot >= (int)n->localizers.size())
    return TSD_E_ARG;
  auto scan = std::make_shared<sensor_msgs::msg::LaserScan>();
  scan->ranges.assign(ranges, ranges + count);
  scan->angle_min = (float)angle_min;
  scan->angle_increment = (float)angle_increment;
  scan->header.stamp.sec = (int32_t)(stamp_ns / 1000000000LL);
  scan->header.stamp.nanosec = (uint32_t)(stamp_ns % 1000000000LL);
  n->localizers[robot]->laserCallBack(scan);
  return TSD_OK;
}

// the scan that tsd_node_laser will deliver NEXT (ThreadLocalize::announceNext): known ahead in a replay, or queued
int tsd_node_laser_ahead(tsd_node* n, int robot, const float* ranges, int count, double angle_min, double angle_increment,
                         long long stamp_ns)
{
  if(!n || robot < 0 || robot >= (int)n->localizers.size())
    return TSD_E_ARG;
  auto scan = std::make_shared<sensor_msgs::msg::LaserScan>();
  scan->ranges.assign(ranges, ranges + count);
  scan->angle_min = (float)angle_min;
  scan->angle_increment = (float)angle_increment;
  scan->header.stamp.sec = (int32_t)(stamp_ns / 1000000000LL);
  scan->header.stamp.nanosec = (uint32_t)(stamp_ns % 1000000000LL);
  n->localizers[robot]->announceNext(scan);
  return TSD_OK;
}

// Replay of recorded scans, the way `rosbag play` feeds the reference node: one publisher thread per robot, each handing
// its robot's scans to laserCallBack in order (synchronous facade: the callback returns when the scan has been processed, so
// every robot runs as fast as the node lets it and the robots' scans overlap).  scans[r] = n_scans x count floats.
int tsd_node_play(tsd_node* n, int robots, const float* const* scans, int first, int n_scans, int count, double angle_min,
                  double angle_increment, long long stamp0_ns, long long stamp_step_ns)
{
  if(!n || robots < 1 || robots > (int)n->localizers.size() || !scans || n_scans < 0 || first < 0)
    return TSD_E_ARG;
  std::vector<std::thread> pubs;
  std::atomic<int> failed{0};
  for(int r = 0; r < robots; r++)
    pubs.emplace_back([&, r] {
      for(int k = first; k < first + n_scans; k++)
      {
        if(k + 1 < first + n_scans)       // (a replay knows the next scan: the localiser stages it during this registration)
          tsd_node_laser_ahead(n, r, scans[r] + (size_t)(k + 1) * (size_t)count, count, angle_min, angle_increment,
                               stamp0_ns + (long long)(k + 1) * stamp_step_ns);
        if(tsd_node_laser(n, r, scans[r] + (size_t)k * (size_t)count, count, angle_min, angle_increment,
                          stamp0_ns + (long long)k * stamp_step_ns) != TSD_OK)
          failed++;
      }
    });
  for(auto& t : pubs)
    t.join();
  return failed ? TSD_E_ARG : TSD_OK;
}

// multi-robot dispatcher: batches begun and scans they carried (0, 0 without a dispatcher)
void tsd_node_batch_stats(tsd_node* n, unsigned long long* batches, unsigned long long* scans)
{
  obvious::ScanBatcher::Stats st;
  if(n && n->grid && n->grid->batcher())
    st = n->grid->batcher()->stats();
  if(batches) *batches = st.batches;
  if(scans) *scans = st.scans;
}

// asynchronous mode: wait until the localiser has nothing queued, the mapping queue is drained and
// the device is idle.  Returns 0 when idle, 1 on timeout.
int tsd_node_wait_idle(tsd_node* n, int timeout_ms)
{
  const auto t0 = std::chrono::steady_clock::now();
  for(;;)
  {
    bool idle = true;
    for(auto* l : n->localizers)
      idle = idle && l->idle();
    idle = idle && n->mapping->pending() == 0;
    if(idle)
    {
      std::lock_guard<std::mutex> lk(n->grid->mutex());
      tsd_sync(n->grid->context());
      return 0;
    }
    if(std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms))
      return 1;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

unsigned long long tsd_node_processed(tsd_node* n, int robot) { return n->localizers[robot]->processedScans(); }

// report layout: pose[9], T[9], rms, pairs, iterations, icpState, validModel, validScene, regError,
// pushed, noModel, initialised, stamp [ns]  (29 doubles)
void tsd_node_report(tsd_node* n, int robot, double* out28)
{
  const ThreadLocalize::ScanReport r = n->localizers[robot]->lastReport();
  std::memcpy(out28, r.pose, sizeof(r.pose));
  std::memcpy(out28 + 9, r.T, sizeof(r.T));
  out28[18] = r.rms; out28[19] = r.pairs; out28[20] = r.iterations; out28[21] = r.icpState;
  out28[22] = r.validModel; out28[23] = r.validScene; out28[24] = r.regError; out28[25] = r.pushed;
  out28[26] = r.noModel; out28[27] = r.initialised; out28[28] = (double)r.stampNs;
}

// last PoseStamped on <node>/<robot/>estimated_pose: x, y, z, qx, qy, qz, qw, publish count
void tsd_node_pose_msg(tsd_node* n, int robot, double* out8)
{
  auto pub = n->localizers[robot]->posePublisher();
  const auto m = pub->last();
  out8[0] = m.pose.position.x; out8[1] = m.pose.position.y; out8[2] = m.pose.position.z;
  out8[3] = m.pose.orientation.x; out8[4] = m.pose.orientation.y; out8[5] = m.pose.orientation.z;
  out8[6] = m.pose.orientation.w; out8[7] = (double)pub->count();
}

const char* tsd_node_pose_topic(tsd_node* n, int robot)
{
  return n->localizers[robot]->posePublisher()->topic().c_str();
}

// last TransformStamped the robot's broadcaster sent (map -> odom): tx, ty, tz, qx, qy, qz, qw, send count; frames as "frame_id|child_frame_id"
void tsd_node_tf_msg(tsd_node* n, int robot, double* out8, char* frames, int cap)
{
  auto* b = n->localizers[robot]->tfBroadcaster();
  const auto m = b->last();
  out8[0] = m.transform.translation.x; out8[1] = m.transform.translation.y; out8[2] = m.transform.translation.z;
  out8[3] = m.transform.rotation.x; out8[4] = m.transform.rotation.y; out8[5] = m.transform.rotation.z;
  out8[6] = m.transform.rotation.w; out8[7] = (double)b->count();
  if(frames && cap > 0)
    std::snprintf(frames, (size_t)cap, "%s|%s", m.header.frame_id.c_str(), m.child_frame_id.c_str());
}

// feed the robot's tf buffer (what a TransformListener does under ROS): `child` expressed in `parent`
int tsd_node_set_transform(tsd_node* n, int robot, const char* parent, const char* child, const double xyz[3], const double qxyzw[4])
{
  geometry_msgs::msg::TransformStamped t;
  t.header.frame_id = parent; t.child_frame_id = child;
  t.transform.translation.x = xyz[0]; t.transform.translation.y = xyz[1]; t.transform.translation.z = xyz[2];
  t.transform.rotation.x = qxyzw[0]; t.transform.rotation.y = qxyzw[1]; t.transform.rotation.z = qxyzw[2]; t.transform.rotation.w = qxyzw[3];
  return n->localizers[robot]->tfBuffer()->setTransform(t, "tsd_node_set_transform", true) ? 0 : -1;
}

tsd_ctx* tsd_node_grid_ctx(tsd_node* n) { return n->grid ? n->grid->context() : nullptr; }
// the facade's grid mutex (obvious::TsdGrid::mutex()): whoever talks to the context through the raw tsd_* ABI while the
// node's worker threads may be running takes it around each call, like the facade's own classes do
void tsd_node_grid_lock(tsd_node* n) { if(n && n->grid) n->grid->mutex().lock(); }
void tsd_node_grid_unlock(tsd_node* n) { if(n && n->grid) n->grid->mutex().unlock(); }

// SlamNode::~SlamNode shutdown order (SlamNode.cpp:131-152): terminate + join every thread
void tsd_node_destroy(tsd_node* n)
{
  if(!n)
    return;
  for(auto* l : n->localizers)
  {
    l->terminateThread();
    while(!l->alive(1)) {}
    delete l;
  }
  if(n->mapping)
  {
    n->mapping->terminateThread();
    while(!n->mapping->alive(1)) {}
    delete n->mapping;
  }
  delete n->grid;
  delete n;
}

// host-only helpers for the CPU unit tests of the sensor model and gates ------------------------
void tsd_host_sensor_ingest_f32(const float* ranges, int n, double ang_res, double phi_min, double max_range,
                                double* data_out, unsigned char* mask_out, int remask)
{
  obvious::SensorPolar2D s((unsigned)n, ang_res, phi_min, max_range, 0.001, 2.0);
  s.setRealMeasurementData(std::vector<float>(ranges, ranges + n));
  s.setStandardMask();
  if(remask == 2)   // the shortcut the fused scan path uses
  {
    std::vector<uint8_t> m;
    s.maskForMapping(m);
    std::memcpy(data_out, s.getRealMeasurementData(), sizeof(double) * (size_t)n);
    std::memcpy(mask_out, m.data(), (size_t)n);
    return;
  }
  if(remask)
  {
    std::unique_ptr<obvious::SensorPolar2D> c(s.copyForMapping());
    std::memcpy(data_out, c->getRealMeasurementData(), sizeof(double) * (size_t)n);
    std::memcpy(mask_out, c->maskBytes(), (size_t)n);
    return;
  }
  std::memcpy(data_out, s.getRealMeasurementData(), sizeof(double) * (size_t)n);
  std::memcpy(mask_out, s.maskBytes(), (size_t)n);
}

// pose after `transform(T1)` then `transform(T2)`, world rays scaled to `norm`, scene points
void tsd_host_sensor_chain(int n, double ang_res, double phi_min, const double* T1, const double* T2, double norm,
                           const float* ranges, double* pose_out, double* rays_out, double* rays_local_out,
                           double* scene_out, unsigned char* scene_mask_out, int* valid_out)
{
  obvious::SensorPolar2D s((unsigned)n, ang_res, phi_min, 30.0, 0.001, 2.0);
  obvious::Matrix A(3, 3, T1), B(3, 3, T2);
  s.setRealMeasurementData(std::vector<float>(ranges, ranges + n));
  s.setStandardMask();
  s.transform(&A);
  const double* r = s.getNormalizedRayMap(norm);
  (void)r;
  s.transform(&B);
  r = s.getNormalizedRayMap(norm);
  s.getTransformation().getData(pose_out);
  std::memcpy(rays_out, r, sizeof(double) * 2 * (size_t)n);
  std::memcpy(rays_local_out, s.getLocalRayMap(), sizeof(double) * 2 * (size_t)n);
  std::vector<char> m((size_t)n);
  *valid_out = (int)s.dataToCartesianVectorMask(scene_out, reinterpret_cast<bool*>(m.data()));
  for(int i = 0; i < n; i++) scene_mask_out[i] = m[(size_t)i] ? 1 : 0;
}

double tsd_host_calc_angle(const double* T) { obvious::Matrix M(3, 3, T); return ThreadLocalize::calcAngle(&M); }
int tsd_host_is_registration_error(const double* T, double trs, double rot)
{
  obvious::Matrix M(3, 3, T);
  return ThreadLocalize::isRegistrationError(&M, trs, rot) ? 1 : 0;
}
int tsd_host_is_pose_change_significant(const double* last, const double* cur)
{
  obvious::Matrix A(3, 3, last), B(3, 3, cur);
  return ThreadLocalize::isPoseChangeSignificant(&A, &B) ? 1 : 0;
}
void tsd_host_mat3_inv(const double* A, double* out) { obvious::Matrix M(3, 3, A); M.invert(); M.getData(out); }
int tsd_host_backproject(const double* pose, double x, double y, int n, double ang_res, double phi_min)
{
  obvious::SensorPolar2D s((unsigned)n, ang_res, phi_min, 30.0, 0.001, 2.0);
  s.setTransformation(obvious::Matrix(3, 3, pose));
  double p[2] = {x, y};
  return s.backProject(p);
}

}  // extern "C"
