#include "ThreadSLAM.h"

namespace ohm_tsd_slam
{

ThreadSLAM::ThreadSLAM(obvious::TsdGrid& grid) :
    _thread(nullptr),
    _wake(false),
    _stayActive(true),
    _grid(grid)
{
  _doneFuture = _done.get_future().share();
}

ThreadSLAM::~ThreadSLAM()
{
  joinThread();
}

void ThreadSLAM::startThread(void)
{
  _thread = new std::thread([this]() {
    this->eventLoop();
    _done.set_value();
  });
}

void ThreadSLAM::joinThread(void)
{
  if(_thread)
  {
    if(_thread->joinable())
      _thread->join();
    delete _thread;
    _thread = nullptr;
  }
}

void ThreadSLAM::unblock(void)
{
  {
    std::lock_guard<std::mutex> lk(_sleepMutex);
    _wake = true;
  }
  _sleepCond.notify_all();
}

void ThreadSLAM::waitForWork(void)
{
  std::unique_lock<std::mutex> lk(_sleepMutex);
  _sleepCond.wait(lk, [this]() { return _wake || !_stayActive; });
  _wake = false;
}

bool ThreadSLAM::alive(unsigned int ms)
{
  // boost::thread::timed_join: true when the thread has finished within the timeout
  if(!_thread)
    return true;
  return _doneFuture.wait_for(std::chrono::milliseconds(ms)) == std::future_status::ready;
}

void ThreadSLAM::terminateThread(void)
{
  _stayActive = false;
  this->unblock();
}

} /* namespace ohm_tsd_slam */
