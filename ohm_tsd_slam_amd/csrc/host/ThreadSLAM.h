// ThreadSLAM.h -- base class of the SLAM worker threads; public surface of the reference's
// ThreadSLAM (src/ThreadSLAM.h:20-85, src/ThreadSLAM.cpp).  std::thread / std::condition_variable
// replace boost; the wait is done on a locked mutex with a wake-up flag, which fixes the reference's
// condition_variable_any::wait on an unlocked mutex without changing the
// observable behaviour of unblock() / alive() / terminateThread().
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <future>
#include <mutex>
#include <thread>

#include "obvision/obvious.h"

namespace ohm_tsd_slam
{

class ThreadSLAM
{
public:
  ThreadSLAM(obvious::TsdGrid& grid);
  virtual ~ThreadSLAM();

  /** wake the thread up (ThreadSLAM.cpp:19-22) */
  void unblock(void);
  /** true if the thread terminated within ms milliseconds (ThreadSLAM.cpp:24-27, timed_join) */
  bool alive(unsigned int ms);
  /** ask the event loop to leave (ThreadSLAM.cpp:29-33) */
  void terminateThread(void);

protected:
  virtual void eventLoop(void) = 0;
  /** derived constructors call this once they are fully built (the reference starts the thread in
   *  the base constructor and relies on the first condvar wait) */
  void startThread(void);
  /** block until unblock() was called since the last wait, or the thread is told to stop */
  void waitForWork(void);
  void joinThread(void);

  std::thread* _thread;
  std::mutex _sleepMutex;
  std::condition_variable _sleepCond;
  bool _wake;
  std::atomic<bool> _stayActive;
  obvious::TsdGrid& _grid;
  std::promise<void> _done;
  std::shared_future<void> _doneFuture;
};

} /* namespace ohm_tsd_slam */
