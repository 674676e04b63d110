// sn_copier.h -- row-band copies spread over a few worker threads (plain C++, no HIP): used by the host ring
// of sn_api.hip; tests/test_capi_cpu.py stress-tests it under ThreadSanitizer (host/copier_test.cpp).
#pragma once

#include <stdint.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <memory>
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
        // Every call hands out its OWN immutable batch (bands + counters).  A worker that wakes late still holds
        // the batch of the call it was woken for: its counters are exhausted, so it leaves without touching the
        // bands of a later call (one shared set of counters let such a worker take a band index of the previous
        // call and apply it to the next one: a band copied twice, another one not yet copied when run() returned).
        auto batch = std::make_shared<Batch>();
        for (int j = 0; j < njobs; ++j) {
            const Job& b = jobs[j];
            const int step = b.row_bytes > 0 ? (kBandBytes + b.row_bytes - 1) / b.row_bytes : b.rows;
            for (int y = 0; y < b.rows; y += step) {
                Job band = b;
                band.dst += (size_t)y * b.dpitch;
                band.src += (size_t)y * b.spitch;
                band.rows = b.rows - y < step ? b.rows - y : step;
                batch->bands.push_back(band);
            }
        }
        if (threads_.empty() || batch->bands.size() < 2) {
            for (const Job& b : batch->bands) copy(b);
            return;
        }
        batch->left.store((int)batch->bands.size());
        {
            std::lock_guard<std::mutex> lk(m_);
            current_ = batch;
            ++epoch_;
        }
        cv_.notify_all();
        work(*batch);
        std::unique_lock<std::mutex> lk(m_);  // every band of THIS batch copied
        done_.wait(lk, [&] { return batch->left.load() == 0; });
    }

private:
    static constexpr int kBandBytes = 512 * 1024;
    struct Batch {
        std::vector<Job> bands;
        std::atomic<int> next{0}, left{0};
    };
    static void copy(const Job& b)
    {
        if (b.dpitch == b.spitch && b.dpitch == b.row_bytes) {
            memcpy(b.dst, b.src, (size_t)b.row_bytes * b.rows);
            return;
        }
        for (int y = 0; y < b.rows; ++y) memcpy(b.dst + (size_t)y * b.dpitch, b.src + (size_t)y * b.spitch, b.row_bytes);
    }
    // returns true if this thread copied the batch's last band
    static bool work(Batch& b)
    {
        bool last = false;
        const int n = (int)b.bands.size();
        for (int i = b.next.fetch_add(1); i < n; i = b.next.fetch_add(1)) {
            copy(b.bands[i]);
            last = b.left.fetch_sub(1) == 1;
        }
        return last;
    }
    void loop()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return quit_ || epoch_ != seen; });
            if (quit_) return;
            seen = epoch_;
            std::shared_ptr<Batch> batch = current_;
            lk.unlock();
            const bool last = work(*batch);
            lk.lock();
            if (last) done_.notify_all();  // under m_: run() is either before its predicate check or waiting
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::shared_ptr<Batch> current_;  // guarded by m_
    uint64_t epoch_ = 0;
    bool quit_ = false;
};

}  // namespace sn
