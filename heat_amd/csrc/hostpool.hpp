// hostpool.hpp — a small pool of host threads for the gathers / scatters between the caller's SimulationState and
// the pinned staging buffers of heat_batch_march (the device copies run meanwhile). Host-only.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace heat {

class HostPool {
  public:
    explicit HostPool(int n_threads) {
        n_ = n_threads < 1 ? 1 : n_threads;
        for (int t = 1; t < n_; t++) workers_.emplace_back([this, t] { loop(t); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    int size() const { return n_; }
    // fn(begin, end) over [0, n) cut into size() contiguous pieces; returns when all of them are done.
    void run(int64_t n, const std::function<void(int64_t, int64_t)> &fn) {
        if (n <= 0) return;
        if (n_ == 1 || n < 4096) {
            fn(0, n);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            total_ = n;
            pending_ = n_ - 1;
            gen_++;
        }
        cv_.notify_all();
        piece(0);
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

  private:
    void piece(int t) {
        const int64_t b = total_ * t / n_, e = total_ * (t + 1) / n_;
        if (e > b) (*fn_)(b, e);
    }
    void loop(int t) {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            piece(t);
            {
                std::lock_guard<std::mutex> g(m_);
                pending_--;
            }
            done_.notify_one();
        }
    }
    int n_ = 1;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int64_t, int64_t)> *fn_ = nullptr;
    int64_t total_ = 0;
    int pending_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};

}  // namespace heat
