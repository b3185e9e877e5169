// ThreadMapping.h -- the map-update worker; public surface of the reference's ThreadMapping
// (src/ThreadMapping.h:25-87, src/ThreadMapping.cpp).
#pragma once
#include <deque>
#include <mutex>

#include "ThreadSLAM.h"

namespace ohm_tsd_slam
{

class ThreadMapping : public ThreadSLAM
{
public:
  ThreadMapping(obvious::TsdGrid* grid);
  virtual ~ThreadMapping();

  /** deep-copies the sensor and queues it; the mapping thread pushes LIFO (ThreadMapping.cpp:65-76) */
  void queuePush(obvious::SensorPolar2D* sensor);
  bool initialized(void);
  /** one synchronous TsdGrid::push on the caller's thread (ThreadMapping.cpp:32-41) */
  void initPush(obvious::SensorPolar2D* sensor);
  /** number of queued sensors (test / drain helper; not in the reference) */
  size_t pending(void);

protected:
  virtual void eventLoop(void);

private:
  std::deque<obvious::SensorPolar2D*> _sensors;
  std::mutex _pushMutex;
  bool _initialized;
  bool _busy;
};

} /* namespace */
