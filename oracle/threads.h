// oracle/threads.h -- threading knobs of the CPU restatement (TEST INFRASTRUCTURE ONLY, see mml_oracle.h).
// The defaults (1, 1) are the bit-reference configuration every parity test uses.  The "reference-shaped" CPU baseline
// of bench_cpu.cpp sets them to the reference's own layout:
//   livox_line_threads = 6   one thread per Livox line around detectFeaturePoints (unionFeatureExtract.cpp:1008-1015);
//                            the Velodyne rings stay serial (:1228-1230)
//   solve_threads      = 6   ceres::Solver::Options::num_threads (Estimator.cpp:1430): residual blocks evaluated in
//                            parallel.  Partial sums are combined in chunk order, so the result differs from the serial
//                            sum in the last bits, exactly as a multi-threaded Ceres run differs from a serial one.
#ifndef MMLO_THREADS_H
#define MMLO_THREADS_H

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace mmlo {

// thread-local so that scan-parallel drivers (one scan per host thread) can keep their workers serial
struct Threading {
    int livox_line_threads = 1;
    int solve_threads = 1;
};
inline Threading& threading() {
    static thread_local Threading t;
    return t;
}

// a persistent pool (Ceres keeps one too): run fn(0..n-1) on up to n workers, the caller takes task 0
class Pool {
   public:
    explicit Pool(int workers) {
        for (int i = 0; i < workers; ++i) th_.emplace_back([this, i] { loop(i + 1); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int workers() const { return (int)th_.size(); }
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 1 || th_.empty()) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> l(m_);
            fn_ = &fn;
            n_ = n;
            pending_ = (int)th_.size();
            ++gen_;
        }
        cv_.notify_all();
        for (int i = 0; i < n; i += (int)th_.size() + 1) fn(i);  // the caller is worker 0
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }

   private:
    void loop(int id) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)>* fn;
            int n;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
                n = n_;
            }
            for (int i = id; i < n; i += (int)th_.size() + 1) (*fn)(i);
            {
                std::lock_guard<std::mutex> l(m_);
                --pending_;
            }
            done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0, pending_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

// one pool per calling thread, sized on first use
inline Pool& pool(int threads) {
    static thread_local Pool* p = nullptr;
    static thread_local int size = 0;
    if (!p || size != threads) {
        delete p;
        p = new Pool(threads - 1);
        size = threads;
    }
    return *p;
}

}  // namespace mmlo
#endif
