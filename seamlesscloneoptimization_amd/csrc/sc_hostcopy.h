// sc_hostcopy.h -- persistent helper threads for the host side of the drop-in call (gfx950 library, host code).
//
// Pageable caller images are packed row by row into pinned staging before they cross PCIe, and the result is
// spliced back row by row (sc_api.cpp).  One core copies ~14 GB/s, so a 2048^2 call (29 MB in, 12 MB out) needs
// several; creating std::threads per copy cost more than the copy itself.  RowCopier keeps a few helpers parked on
// a condition variable and hands them row ranges.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sc {

class RowCopier {
public:
    explicit RowCopier(int helpers);
    ~RowCopier();
    RowCopier(const RowCopier &) = delete;
    RowCopier &operator=(const RowCopier &) = delete;
    int width() const { return (int)th_.size() + 1; }      // threads that take part, the caller included
    // fn(i) for i in [0, nparts): shared between the helpers and the calling thread; returns when all are done
    void parallel(int nparts, const std::function<void(int)> &fn);

private:
    void worker();
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(int)> *fn_ = nullptr;
    int nparts_ = 0, busy_ = 0;
    unsigned long gen_ = 0;
    std::atomic<int> next_{ 0 };
    bool stop_ = false;
};

} // namespace sc
