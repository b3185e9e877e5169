#include "ThreadMapping.h"

namespace ohm_tsd_slam
{

namespace { const unsigned int INIT_PSHS = 1; }   // SlamNode.h:30

ThreadMapping::ThreadMapping(obvious::TsdGrid* grid):
    ThreadSLAM(*grid),
    _initialized(false),
    _busy(false)
{
  startThread();
}

ThreadMapping::~ThreadMapping()
{
  terminateThread();
  joinThread();
  for(auto* s : _sensors)
    delete s;
  _sensors.clear();
}

bool ThreadMapping::initialized(void)
{
  std::lock_guard<std::mutex> lk(_pushMutex);
  return _initialized;
}

size_t ThreadMapping::pending(void)
{
  std::lock_guard<std::mutex> lk(_pushMutex);
  return _sensors.size() + (_busy ? 1 : 0);
}

void ThreadMapping::initPush(obvious::SensorPolar2D* sensor)
{
  if(this->initialized())
    return;
  std::lock_guard<std::mutex> lk(_pushMutex);
  for(unsigned int i = 0; i < INIT_PSHS; i++)
    _grid.push(sensor);
  _initialized = true;
}

void ThreadMapping::eventLoop(void)
{
  while(_stayActive)
  {
    waitForWork();
    for(;;)
    {
      obvious::SensorPolar2D* sensor = nullptr;
      {
        std::lock_guard<std::mutex> lk(_pushMutex);
        if(!_stayActive || _sensors.empty())
          break;
        sensor = _sensors.back();     // LIFO (ThreadMapping.cpp:51-52)
        _sensors.pop_back();
        _busy = true;
      }
      _grid.push(sensor);
      {
        std::lock_guard<std::mutex> lk(_pushMutex);
        delete sensor;
        _initialized = true;
        _busy = false;
      }
    }
  }
}

void ThreadMapping::queuePush(obvious::SensorPolar2D* sensor)
{
  obvious::SensorPolar2D* copy = sensor->copyForMapping();
  {
    std::lock_guard<std::mutex> lk(_pushMutex);
    _sensors.push_back(copy);
  }
  this->unblock();
}

} /* namespace */
