// sc_hostcopy.cpp -- see sc_hostcopy.h
#include "sc_hostcopy.h"

namespace sc {

RowCopier::RowCopier(int helpers)
{
    for (int i = 0; i < helpers; ++i) th_.emplace_back([this] { worker(); });
}

RowCopier::~RowCopier()
{
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    cv_work_.notify_all();
    for (std::thread &t : th_) t.join();
}

void RowCopier::worker()
{
    unsigned long seen = 0;
    for (;;) {
        const std::function<void(int)> *fn;
        int n;
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_work_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            fn = fn_; n = nparts_;
        }
        for (int i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < n;) (*fn)(i);
        {
            std::lock_guard<std::mutex> lk(m_);
            if (--busy_ == 0) cv_done_.notify_one();
        }
    }
}

void RowCopier::parallel(int nparts, const std::function<void(int)> &fn)
{
    if (nparts <= 1 || th_.empty()) {
        for (int i = 0; i < nparts; ++i) fn(i);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(m_);
        fn_ = &fn; nparts_ = nparts;
        next_.store(0, std::memory_order_relaxed);
        busy_ = (int)th_.size();
        ++gen_;
    }
    cv_work_.notify_all();
    for (int i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < nparts;) fn(i);
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return busy_ == 0; });
}

} // namespace sc
