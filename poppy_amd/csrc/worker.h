// worker.h — one persistent helper thread that runs one job at a time.  The pair set-up hands the second image's chain and half of the
// detector's host work to a helper; creating a std::thread for each (three per set-up) costs 50-100 us apiece plus the HIP runtime's
// per-thread initialisation at its first call, which a 4.6 ms set-up notices.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace poppy_hip {

class Worker {
public:
    Worker() = default;
    Worker(const Worker&) = delete;
    Worker& operator=(const Worker&) = delete;
    ~Worker() { stop(); }
    // starts `job` on the helper thread (created at the first call); the previous job must have been waited for
    void run(std::function<void()> job) {
        std::unique_lock<std::mutex> lk(m_);
        if (!thread_.joinable()) thread_ = std::thread([this]() { loop(); });
        job_ = std::move(job);
        busy_ = true;
        lk.unlock();
        cv_.notify_all();
    }
    // returns when the job started last has finished (at once when there is none)
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this]() { return !busy_; });
    }
    void stop() {
        {
            std::unique_lock<std::mutex> lk(m_);
            if (!thread_.joinable()) return;
            cv_.wait(lk, [this]() { return !busy_; });
            quit_ = true;
        }
        cv_.notify_all();
        thread_.join();
        quit_ = false;
    }

private:
    void loop() {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [this]() { return busy_ || quit_; });
            if (quit_) return;
            std::function<void()> job = std::move(job_);
            job_ = nullptr;
            lk.unlock();
            job();
            lk.lock();
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::thread thread_;
    std::function<void()> job_;
    bool busy_ = false, quit_ = false;
};

}  // namespace poppy_hip
