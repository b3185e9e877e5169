// thread_contract.cpp -- the facade's worker-thread contract on a box without a GPU, built with -fsanitize=thread and with
// -fsanitize=address,undefined by tests/test_cpu_thread_contract.py.  Links the facade's own sources (csrc/host/*.cpp) against
// tests/stub_tsd_hip.c, a recording stand-in for the device ABI (canned results; NOT the oracle).  What the reference promises
// and SURVEY 4 / 5 ask to be tested:
//   first scan      ThreadLocalize::laserCallBack runs init on the CALLER's thread: freeFootprint, then ThreadMapping::initPush,
//                   done when the callback returns (ThreadLocalize.cpp:248-276, :411-511; ThreadMapping.cpp:32-41)
//   newest wins     scans that arrive while one is being registered are dropped but the newest (ThreadLocalize.cpp:319-332)
//   LIFO mapper     ThreadMapping pushes the most recently queued sensor first (ThreadMapping.cpp:43-76)
//   announceNext    a staged scan is used only if it is the scan that comes; anything else drops it
//   shutdown        terminateThread() + alive(ms) end the loops; destructors join with work still queued (ThreadSLAM.cpp:19-33)
// Prints "ok <case>" per case; exit code = number of failed checks.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <set>
#include <thread>
#include <vector>

#include "ThreadLocalize.h"
#include "ThreadMapping.h"
#include "stub_tsd_hip.h"

using namespace ohm_tsd_slam;

static int g_failed = 0;
#define CHECK(cond, ...) do { if(!(cond)) { g_failed++; std::fprintf(stderr, "FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); \
                                           std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } } while(0)

static const int BEAMS = 360;
// (a reading inside the sensor's range, exact in float and double: 10.0, 10.25, 10.5, ...)
static inline float TAG(int number) { return 10.0f + 0.25f * (float)number; }

static std::shared_ptr<sensor_msgs::msg::LaserScan> make_scan(int number)
{
  auto s = std::make_shared<sensor_msgs::msg::LaserScan>();
  s->ranges.assign(BEAMS, 4.0f);
  s->ranges[0] = TAG(number);                    // the scan's number travels in beam 0 (the stub logs it)
  s->angle_min = -3.14159265f;
  s->angle_increment = 6.2831853f / BEAMS;
  s->header.stamp.sec = number;
  s->header.stamp.nanosec = 0;
  return s;
}

static std::vector<stub_entry> log_of(int op)
{
  std::vector<stub_entry> v;
  const int n = stub_log_count();
  for(int i = 0; i < n; i++) { const stub_entry e = stub_log_get(i); if(e.op == op) v.push_back(e); }
  return v;
}

static bool wait_until(const std::function<bool()>& f, int ms)
{
  const auto t0 = std::chrono::steady_clock::now();
  while(!f())
  {
    if(std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(ms)) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  return true;
}

struct Rig
{
  std::shared_ptr<rclcpp::Node> node;
  obvious::TsdGrid* grid;
  ThreadMapping* mapping;
  ThreadLocalize* loc;
  Rig(bool fused, bool synchronous)
  {
    stub_reset();
    node = std::make_shared<rclcpp::Node>("tsd_slam");
    node->set_parameter("laser_min_range", 0.26);
    grid = new obvious::TsdGrid(0.05, obvious::LAYOUT_32x32, static_cast<obvious::EnumTsdGridLayout>(9), 0);
    grid->setMaxTruncation(0.15);
    mapping = new ThreadMapping(grid);
    loc = new ThreadLocalize(grid, mapping, node, "", 0.0, 0.0);
    loc->setFused(fused);
    loc->setSynchronous(synchronous);
  }
  ~Rig()
  {
    // SlamNode::~SlamNode (SlamNode.cpp:131-152): terminate + join every thread, then the grid
    loc->terminateThread();
    CHECK(loc->alive(2000), "the localiser did not end within 2 s of terminateThread()");
    delete loc;
    mapping->terminateThread();
    CHECK(mapping->alive(2000), "the mapper did not end within 2 s of terminateThread()");
    delete mapping;
    delete grid;
    CHECK(stub_live_objects() == 0, "%d device objects left alive", stub_live_objects());
  }
};

// ------------------------------------------------------------------------------------------------------------------------------
static void case_first_scan_is_synchronous()
{
  Rig r(/*fused*/ false, /*synchronous*/ false);
  stub_set_delay_us(STUB_PUSH, 20000);                         // a slow initial push: the callback must still wait for it
  CHECK(!r.mapping->initialized(), "initialised before any scan");
  const unsigned long long me = (unsigned long long)(uintptr_t)pthread_self();
  r.loc->laserCallBack(make_scan(0));
  // when the callback returns: footprint freed, ONE push done, both on the caller's thread, mapper initialised
  const auto ff = log_of(STUB_FREE_FOOTPRINT), ps = log_of(STUB_PUSH);
  CHECK(ff.size() == 1 && ps.size() == 1, "free_footprint %zu, push %zu", ff.size(), ps.size());
  if(ff.size() == 1 && ps.size() == 1)
  {
    CHECK(ff[0].thread == me && ps[0].thread == me, "init ran on another thread");
    CHECK(ps[0].tag == TAG(0), "the initial push carried scan %g", ps[0].tag);
  }
  CHECK(r.mapping->initialized(), "mapper not initialised after the first scan");
  CHECK(r.loc->processedScans() == 1 && r.loc->lastReport().initialised && r.loc->lastReport().pushed, "first report");
  CHECK(log_of(STUB_LOCALIZE).empty(), "the first scan was registered");
  std::printf("ok first_scan_is_synchronous\n");
}

static void case_newest_scan_wins()
{
  Rig r(false, false);
  r.loc->laserCallBack(make_scan(0));
  stub_set_delay_us(STUB_LOCALIZE, 30000);                     // one registration = 30 ms of "device time"
  r.loc->laserCallBack(make_scan(1));                          // taken at once by the (idle) event loop
  CHECK(wait_until([&] { return log_of(STUB_LOCALIZE).size() == 1; }, 2000), "scan 1 never reached the device");
  for(int k = 2; k <= 6; k++) r.loc->laserCallBack(make_scan(k));     // arrive while scan 1 registers: 2..5 must be dropped
  CHECK(wait_until([&] { return r.loc->idle() && r.mapping->pending() == 0; }, 5000), "not idle");
  const auto regs = log_of(STUB_LOCALIZE);
  CHECK(regs.size() == 2, "%zu registrations for 6 scans, expected 2 (scan 1 and the newest)", regs.size());
  if(regs.size() == 2)
    CHECK(regs[0].tag == TAG(1) && regs[1].tag == TAG(6), "registered scans %g, %g", regs[0].tag, regs[1].tag);
  CHECK(r.loc->lastReport().stampNs == 6LL * 1000000000LL, "last report is of stamp %lld", r.loc->lastReport().stampNs);
  // the registrations ran on the localiser's thread, not on the publisher's
  const unsigned long long me = (unsigned long long)(uintptr_t)pthread_self();
  for(const auto& e : regs) CHECK(e.thread != me, "a registration ran on the publisher's thread in threaded mode");
  std::printf("ok newest_scan_wins\n");
}

static void case_mapper_is_lifo()
{
  Rig r(false, false);
  r.loc->laserCallBack(make_scan(0));
  obvious::SensorPolar2D* s = r.loc->sensor();
  stub_set_delay_us(STUB_PUSH, 30000);
  auto queue = [&](int number) {
    std::vector<float> ranges(BEAMS, 4.0f);
    ranges[0] = TAG(number);
    s->setRealMeasurementData(ranges);
    s->setStandardMask();
    r.mapping->queuePush(s);                                   // deep copy (ThreadMapping.cpp:65-76): s may change afterwards
  };
  queue(1);
  CHECK(wait_until([&] { return log_of(STUB_PUSH).size() == 2; }, 2000), "push 1 never started");    // (init push + scan 1, now busy)
  queue(2); queue(3); queue(4);                                // queued while push 1 is busy
  CHECK(r.mapping->pending() >= 3, "pending %zu", r.mapping->pending());
  CHECK(wait_until([&] { return r.mapping->pending() == 0; }, 5000), "mapper never drained");
  const auto ps = log_of(STUB_PUSH);
  CHECK(ps.size() == 5, "%zu pushes", ps.size());
  if(ps.size() == 5)
    CHECK(ps[1].tag == TAG(1) && ps[2].tag == TAG(4) && ps[3].tag == TAG(3) && ps[4].tag == TAG(2),
          "push order %g %g %g %g, expected scans 1 4 3 2 (LIFO)", ps[1].tag, ps[2].tag, ps[3].tag, ps[4].tag);
  std::printf("ok mapper_is_lifo\n");
}

static void case_threaded_unfused_scan_goes_through_the_mapper()
{
  Rig r(false, false);
  r.loc->laserCallBack(make_scan(0));
  r.loc->laserCallBack(make_scan(1));
  CHECK(wait_until([&] { return r.loc->processedScans() == 2 && r.mapping->pending() == 0 && log_of(STUB_PUSH).size() == 2; }, 3000), "scan 1 not mapped");
  const auto ps = log_of(STUB_PUSH);
  const auto regs = log_of(STUB_LOCALIZE);
  if(ps.size() == 2 && regs.size() == 1)
  {
    CHECK(ps[1].tag == TAG(1), "pushed scan %g", ps[1].tag);
    CHECK(ps[1].thread != regs[0].thread && ps[1].thread != ps[0].thread, "the push of a registered scan must run on the MAPPING thread");
  }
  else CHECK(false, "pushes %zu registrations %zu", ps.size(), regs.size());
  CHECK(r.loc->lastReport().pushed && !r.loc->lastReport().regError, "report");
  CHECK(r.loc->posePublisher()->count() >= 1, "no pose published");
  std::printf("ok threaded_unfused_scan_goes_through_the_mapper\n");
}

static void case_announce_next_accept_and_drop()
{
  Rig r(/*fused*/ true, /*synchronous*/ true);
  r.loc->laserCallBack(make_scan(0));
  // scan 1 with scan 2 announced: 2 is staged during 1's registration ...
  r.loc->announceNext(make_scan(2));
  r.loc->laserCallBack(make_scan(1));
  // ... and accepted when it comes
  r.loc->announceNext(make_scan(3));
  r.loc->laserCallBack(make_scan(2));
  // 3 was staged, but 7 comes (a different stamp and different readings): the staged scan is dropped
  r.loc->laserCallBack(make_scan(7));
  // same stamp as an announced scan but other readings must not be taken for it either
  auto fake = make_scan(9);
  r.loc->announceNext(fake);
  r.loc->laserCallBack(make_scan(8));                          // stages 9 meanwhile
  auto impostor = make_scan(9);
  impostor->ranges[5] = 2.5f;
  r.loc->laserCallBack(impostor);
  const auto sub = log_of(STUB_SCAN_SUBMIT), stg = log_of(STUB_SCAN_STAGE);
  CHECK(sub.size() == 5, "%zu submits", sub.size());
  if(sub.size() == 5)
  {
    CHECK(sub[0].tag == TAG(1) && sub[0].flag == 0, "scan 1: tag %g flag %d", sub[0].tag, sub[0].flag);
    CHECK(sub[1].tag == TAG(2) && sub[1].flag == 1, "scan 2 should start from the staged data: tag %g flag %d", sub[1].tag, sub[1].flag);
    CHECK(sub[2].tag == TAG(7) && sub[2].flag == 2, "scan 7 should drop the staged scan 3: tag %g flag %d", sub[2].tag, sub[2].flag);
    CHECK(sub[3].tag == TAG(8) && sub[3].flag == 0, "scan 8: tag %g flag %d", sub[3].tag, sub[3].flag);
    CHECK(sub[4].tag == TAG(9) && sub[4].flag == 2, "the impostor of scan 9 must not use the staged readings: tag %g flag %d", sub[4].tag, sub[4].flag);
  }
  CHECK(stg.size() == 3 && stg[0].tag == TAG(2) && stg[1].tag == TAG(3) && stg[2].tag == TAG(9), "%zu stagings", stg.size());
  CHECK(r.loc->processedScans() == 6, "processed %llu", (unsigned long long)r.loc->processedScans());
  std::printf("ok announce_next_accept_and_drop\n");
}

static void case_threaded_fused_stages_the_queued_scan()
{
  // threaded + fused: a scan that queued up during a registration is staged ahead and, if a NEWER one arrives, dropped for it
  Rig r(true, false);
  r.loc->laserCallBack(make_scan(0));
  stub_set_delay_us(STUB_SCAN_COLLECT, 30000);
  r.loc->laserCallBack(make_scan(1));
  CHECK(wait_until([&] { return log_of(STUB_SCAN_SUBMIT).size() == 1; }, 2000), "scan 1 never submitted");
  std::this_thread::sleep_for(std::chrono::milliseconds(50));     // (the event loop is in scan 1's collect by now)
  for(int k = 2; k <= 5; k++) r.loc->laserCallBack(make_scan(k));
  CHECK(wait_until([&] { return r.loc->idle(); }, 5000), "not idle");
  const auto sub = log_of(STUB_SCAN_SUBMIT);
  CHECK(sub.size() >= 2 && sub.size() <= 3, "%zu submits", sub.size());
  CHECK(!sub.empty() && sub.back().tag == TAG(5), "the last registered scan is %g, expected the newest (scan 5)", sub.empty() ? 0.0 : sub.back().tag);
  CHECK(r.loc->lastReport().stampNs == 5LL * 1000000000LL, "last stamp %lld", r.loc->lastReport().stampNs);
  std::printf("ok threaded_fused_stages_the_queued_scan\n");
}

static void case_shutdown_with_work_queued()
{
  {
    Rig r(false, false);
    r.loc->laserCallBack(make_scan(0));
    stub_set_delay_us(STUB_LOCALIZE, 20000);
    stub_set_delay_us(STUB_PUSH, 20000);
    for(int k = 1; k <= 4; k++) r.loc->laserCallBack(make_scan(k));
    obvious::SensorPolar2D* s = r.loc->sensor();
    (void)s;
    // ~Rig: terminate while a registration is in flight and scans are queued -- must end promptly, join, free everything
  }
  {
    // a mapper that is torn down with sensors still queued frees them (ASan: no leak), and unblock() after terminate is harmless
    stub_reset();
    obvious::TsdGrid* grid = new obvious::TsdGrid(0.05, obvious::LAYOUT_32x32, static_cast<obvious::EnumTsdGridLayout>(9), 0);
    auto* mapping = new ThreadMapping(grid);
    obvious::SensorPolar2D sensor(BEAMS, 6.2831853 / BEAMS, -3.14159265, 30.0, 0.001, 2.0);
    std::vector<float> ranges(BEAMS, 4.0f);
    sensor.setRealMeasurementData(ranges);
    sensor.setStandardMask();
    stub_set_delay_us(STUB_PUSH, 30000);
    for(int k = 0; k < 6; k++) mapping->queuePush(&sensor);
    mapping->terminateThread();
    mapping->unblock();
    CHECK(mapping->alive(2000), "mapper still running 2 s after terminateThread()");
    delete mapping;
    delete grid;
    CHECK(log_of(STUB_PUSH).size() < 6, "every queued push ran although the thread was told to stop");
  }
  std::printf("ok shutdown_with_work_queued\n");
}

// ------------------------------------------------------------------------------------------------------------------------------
// The tf half of sendTransform (ThreadLocalize.cpp:603-689).  (i) no tree: both look-ups throw, are skipped, and the message's
// transform is NOT written (the reference assigns it only where the odom look-up succeeded, :657) -- the broadcaster repeats
// the identity the message started with; the PoseStamped carries the laser pose.  (ii) odom -> base_footprint -> laser heard:
// map -> odom = pose * T(laser <- footprint) * T(footprint <- odom); the numbers are printed ("tf_result ...") and
// tests/test_cpu_thread_contract.py checks them against a numpy product of 4 x 4 matrices.  (iii) the tree goes away again: the
// transform keeps its last good value.
static void print_tf(const char* what, const geometry_msgs::msg::Transform& t)
{
  std::printf("tf_result %s %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", what, t.translation.x, t.translation.y, t.translation.z,
              t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w);
}

static void case_tf_map_to_odom()
{
  Rig r(/*fused*/ false, /*synchronous*/ true);
  r.loc->laserCallBack(make_scan(0));
  r.loc->laserCallBack(make_scan(1));
  auto* bc = r.loc->tfBroadcaster();
  CHECK(bc->count() >= 1, "nothing was broadcast");
  {
    const auto m = bc->last();
    CHECK(m.header.frame_id == "map" && m.child_frame_id == "odom", "frames %s -> %s", m.header.frame_id.c_str(), m.child_frame_id.c_str());
    CHECK(m.transform.translation.x == 0.0 && m.transform.translation.y == 0.0 && m.transform.translation.z == 0.0 &&
          m.transform.rotation.x == 0.0 && m.transform.rotation.y == 0.0 && m.transform.rotation.z == 0.0 && m.transform.rotation.w == 1.0,
          "without a tf tree the transform was written (%g %g)", m.transform.translation.x, m.transform.translation.y);
    CHECK(m.header.stamp.sec == 1, "stamp %d", m.header.stamp.sec);
    const auto p = r.loc->posePublisher()->last();
    const auto rep = r.loc->lastReport();
    const double W = 512 * 0.05;
    CHECK(std::fabs(p.pose.position.x - (rep.pose[2] - 0.5 * W)) < 1e-12 && std::fabs(p.pose.position.y - (rep.pose[5] - 0.5 * W)) < 1e-12,
          "PoseStamped %g %g against pose %g %g", p.pose.position.x, p.pose.position.y, rep.pose[2], rep.pose[5]);
  }
  // odom -> base_footprint -> laser, both with a rotation that is not about z only
  geometry_msgs::msg::TransformStamped ob, bl;
  ob.header.frame_id = "odom"; ob.child_frame_id = "base_footprint";
  ob.transform.translation.x = 1.25; ob.transform.translation.y = -0.5; ob.transform.translation.z = 0.0;
  { tf2::Quaternion q; q.setEuler(0.02, -0.01, 0.7); ob.transform.rotation.x = q.x(); ob.transform.rotation.y = q.y(); ob.transform.rotation.z = q.z(); ob.transform.rotation.w = q.w(); }
  bl.header.frame_id = "base_footprint"; bl.child_frame_id = "laser";
  bl.transform.translation.x = 0.3; bl.transform.translation.y = 0.05; bl.transform.translation.z = 0.2;
  { tf2::Quaternion q; q.setEuler(0.0, 0.0, -0.1); bl.transform.rotation.x = q.x(); bl.transform.rotation.y = q.y(); bl.transform.rotation.z = q.z(); bl.transform.rotation.w = q.w(); }
  CHECK(r.loc->tfBuffer()->setTransform(ob, "test", false) && r.loc->tfBuffer()->setTransform(bl, "test", true), "setTransform");
  const uint64_t before = bc->count();
  r.loc->laserCallBack(make_scan(2));
  CHECK(bc->count() == before + 1, "one scan, %llu broadcasts", (unsigned long long)(bc->count() - before));
  const auto good = bc->last();
  {
    const auto p = r.loc->posePublisher()->last();
    geometry_msgs::msg::Transform laser;
    laser.translation.x = p.pose.position.x; laser.translation.y = p.pose.position.y; laser.translation.z = p.pose.position.z;
    laser.rotation = p.pose.orientation;
    print_tf("laser_pose", laser);
    print_tf("odom_base", ob.transform);
    print_tf("base_laser", bl.transform);
    print_tf("map_odom", good.transform);
    CHECK(good.header.frame_id == "map" && good.child_frame_id == "odom" && good.header.stamp.sec == 2, "frames / stamp of the corrected transform");
  }
  // the tree is lost again: the look-ups throw, the transform is left as it was
  r.loc->tfBuffer()->clear();
  r.loc->laserCallBack(make_scan(3));
  {
    const auto m = bc->last();
    CHECK(m.header.stamp.sec == 3, "stamp %d", m.header.stamp.sec);
    CHECK(std::memcmp(&m.transform, &good.transform, sizeof(m.transform)) == 0, "the transform changed although the odom look-up threw");
  }
  // only laser -> base_footprint known: still no write (the reference composes it into `pose` but publishes nothing new, :618-666)
  r.loc->tfBuffer()->setTransform(bl, "test", true);
  r.loc->laserCallBack(make_scan(4));
  {
    const auto m = bc->last();
    CHECK(std::memcmp(&m.transform, &good.transform, sizeof(good.transform)) == 0, "the transform changed without an odom look-up");
  }
  std::printf("ok tf_map_to_odom\n");
}

int main()
{
  case_first_scan_is_synchronous();
  case_newest_scan_wins();
  case_mapper_is_lifo();
  case_threaded_unfused_scan_goes_through_the_mapper();
  case_announce_next_accept_and_drop();
  case_threaded_fused_stages_the_queued_scan();
  case_shutdown_with_work_queued();
  case_tf_map_to_odom();
  if(g_failed) std::fprintf(stderr, "%d check(s) failed\n", g_failed);
  else std::printf("thread_contract: all cases ok\n");
  return g_failed;
}
