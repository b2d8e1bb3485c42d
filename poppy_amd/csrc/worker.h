// worker.h — one persistent helper thread that runs one job at a time.  The pair set-up hands the second image's chain and half of the
// detector's host work to a helper; creating a std::thread for each (three per set-up) costs 50-100 us apiece plus the HIP runtime's
// per-thread initialisation at its first call, which a 4.6 ms set-up notices.
#pragma once
#include <condition_variable>
#include <functional>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

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
    // returns when the job started last has finished (at once when there is none); false = that job threw (what() in error()): a
    // std::bad_alloc on a helper thread reaches the caller as a status instead of std::terminate
    bool wait() {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this]() { return !busy_; });
        const bool ok = !failed_;
        failed_ = false;
        return ok;
    }
    std::string error() { std::lock_guard<std::mutex> lk(m_); return what_; }
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
            bool threw = false;
            std::string what;
            try { job(); } catch (const std::exception& e) { threw = true; what = e.what(); } catch (...) { threw = true; what = "unknown exception"; }
            lk.lock();
            if (threw) { failed_ = true; what_ = what; }
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::thread thread_;
    std::function<void()> job_;
    std::string what_;
    bool busy_ = false, quit_ = false, failed_ = false;
};

// A team of persistent threads that all run the same job (the frame planners of a multi-frame call: sixteen std::thread per call were
// 0.6-0.8 ms of thread creation on the calling thread before the first frame could be submitted).  run() wakes them, wait() returns when
// every one of them has finished the job; threads are created at the first run() and grow to the largest count asked for.
class Team {
public:
    Team() = default;
    Team(const Team&) = delete;
    Team& operator=(const Team&) = delete;
    ~Team() { stop(); }
    void run(int n, std::function<void()> job) {
        std::unique_lock<std::mutex> lk(m_);
        job_ = std::move(job);
        active_ = n;
        pending_ = n;
        ++generation_;
        while ((int)threads_.size() < n) {
            const int id = (int)threads_.size();
            threads_.emplace_back([this, id]() { loop(id, generation_ - 1); });
        }
        lk.unlock();
        cv_.notify_all();
    }
    // no member is inside a job
    bool idle() { std::lock_guard<std::mutex> lk(m_); return pending_ == 0; }
    // false = a member's job threw (error()): see Worker::wait
    bool wait() {
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
        const bool ok = !failed_;
        failed_ = false;
        return ok;
    }
    std::string error() { std::lock_guard<std::mutex> lk(m_); return what_; }
    void stop() {
        {
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [this]() { return pending_ == 0; });
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
        threads_.clear();
        quit_ = false;
    }

private:
    void loop(int id, unsigned long seen) {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&]() { return quit_ || generation_ != seen; });
            if (quit_) return;
            seen = generation_;
            if (id >= active_) continue;                    // this round uses fewer threads
            std::function<void()> job = job_;
            lk.unlock();
            bool threw = false;
            std::string what;
            try { job(); } catch (const std::exception& e) { threw = true; what = e.what(); } catch (...) { threw = true; what = "unknown exception"; }
            lk.lock();
            if (threw) { failed_ = true; what_ = what; }
            if (--pending_ == 0) done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    std::function<void()> job_;
    std::string what_;
    unsigned long generation_ = 0;
    int active_ = 0, pending_ = 0;
    bool quit_ = false, failed_ = false;
};

}  // namespace poppy_hip
