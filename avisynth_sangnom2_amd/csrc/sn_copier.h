// sn_copier.h -- row-band copies spread over a few worker threads (plain C++, no HIP): used by the host ring
// of sn_api.hip; tests/test_capi_cpu.py stress-tests it under ThreadSanitizer (host/copier_test.cpp).
#pragma once

#include <stdint.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace sn {

// The host ring's staging copies (frame planes <-> pinned memory), spread over a few worker threads: one thread's
// memcpy tops out near 14 GB/s, which capped the pipelined host path at 870 2160p frames/s.  Bands of rows are
// handed out through an atomic counter; the calling thread works too and returns when every band is done.
class Copier {
public:
    struct Job {
        uint8_t* dst;
        const uint8_t* src;
        int dpitch, spitch, row_bytes, rows;
    };
    explicit Copier(int workers)
    {
        for (int i = 0; i < workers; ++i) threads_.emplace_back([this] { loop(); });
    }
    ~Copier()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void run(const Job* jobs, int njobs)
    {
        bands_.clear();
        for (int j = 0; j < njobs; ++j) {
            const Job& b = jobs[j];
            const int step = b.row_bytes > 0 ? (kBandBytes + b.row_bytes - 1) / b.row_bytes : b.rows;
            for (int y = 0; y < b.rows; y += step) {
                Job band = b;
                band.dst += (size_t)y * b.dpitch;
                band.src += (size_t)y * b.spitch;
                band.rows = b.rows - y < step ? b.rows - y : step;
                bands_.push_back(band);
            }
        }
        if (threads_.empty() || bands_.size() < 2) {
            for (const Job& b : bands_) copy(b);
            return;
        }
        {   // bands_ is complete before the counters are reset: a worker that sees the reset sees the bands
            std::lock_guard<std::mutex> lk(m_);
            nbands_.store((int)bands_.size());
            left_.store((int)bands_.size());
            next_.store(0);
            ++epoch_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);  // every band copied and no worker still inside work()
        done_.wait(lk, [this] { return left_.load() == 0 && busy_ == 0; });
    }

private:
    static constexpr int kBandBytes = 512 * 1024;
    static void copy(const Job& b)
    {
        if (b.dpitch == b.spitch && b.dpitch == b.row_bytes) {
            memcpy(b.dst, b.src, (size_t)b.row_bytes * b.rows);
            return;
        }
        for (int y = 0; y < b.rows; ++y) memcpy(b.dst + (size_t)y * b.dpitch, b.src + (size_t)y * b.spitch, b.row_bytes);
    }
    void work()
    {
        for (int i = next_.fetch_add(1); i < nbands_.load(); i = next_.fetch_add(1)) {
            copy(bands_[i]);
            left_.fetch_sub(1);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return quit_ || epoch_ != seen; });
            if (quit_) return;
            seen = epoch_;
            ++busy_;
            lk.unlock();
            work();
            lk.lock();
            --busy_;
            done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::vector<Job> bands_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::atomic<int> next_{0}, left_{0}, nbands_{0};
    uint64_t epoch_ = 0;
    int busy_ = 0;  // workers inside work(), guarded by m_
    bool quit_ = false;
};

}  // namespace sn
