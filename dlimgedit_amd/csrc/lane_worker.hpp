// LaneWorker: one host thread that runs posted tasks in order (pure C++, no HIP: tests/sanitize/lane_worker_tsan.cpp builds
// it under ThreadSanitizer).
#pragma once

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace dlimg {

// A host thread that enqueues the passes of ONE execution lane (the device-step queue below): a pass is ~95 kernel
// launches = 0.3-0.6 ms of host time, and a caller that feeds four lanes from one thread gives the fourth lane its first
// kernel 1.3-1.7 ms after the first (measured, tools/enqueue_time.py) -- every burst starts with most of the chip idle.
// With a worker per lane the caller only plans and hands over; the lanes' launch streams are written in parallel.
// Tasks run in the order they were posted; the destructor finishes what is queued and joins.
class LaneWorker {
  public:
    LaneWorker() : thread_([this] { run(); }) {}
    ~LaneWorker() {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        wake_.notify_all();
        if (thread_.joinable()) thread_.join();
    }
    LaneWorker(LaneWorker const&) = delete;
    LaneWorker& operator=(LaneWorker const&) = delete;

    void post(std::function<void()> task) {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            tasks_.push_back(std::move(task));
        }
        wake_.notify_one();
    }
    // returns when nothing is queued and nothing is running
    void drain() {
        std::unique_lock<std::mutex> lock(mutex_);
        idle_.wait(lock, [&] { return tasks_.empty() && !running_; });
    }

  private:
    void run() {
        std::unique_lock<std::mutex> lock(mutex_);
        for (;;) {
            wake_.wait(lock, [&] { return stop_ || !tasks_.empty(); });
            if (tasks_.empty()) return;                    // stop_ and nothing left
            std::function<void()> task = std::move(tasks_.front());
            tasks_.pop_front();
            running_ = true;
            lock.unlock();
            task();                                        // tasks report their own failures (they must not throw)
            task = nullptr;                                // what the task captured goes before the worker counts as idle
            lock.lock();
            running_ = false;
            if (tasks_.empty()) idle_.notify_all();
        }
    }
    std::mutex mutex_;
    std::condition_variable wake_, idle_;
    std::deque<std::function<void()>> tasks_;
    bool running_ = false, stop_ = false;
    std::thread thread_;                                   // last: the members above exist before run() starts
};

}  // namespace dlimg
