// Host transfers of the runtime behind the C ABI (ebm_runtime.hip): a process-wide pool of host threads and, per handle, a
// pinned staging ring with a DMA stream and a worker thread for the asynchronous outputs of ebm_integrate.  Header-only,
// included by ebm_runtime.hip alone; not part of the public interface (include/ebm_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ebm_host {

// ---- host transfers: pinned staging ring + a few host threads ---------------------------------------
// The caller's buffers are pageable.  A device -> pageable copy through the runtime alone runs at 7-8 GB/s
// (one thread faults the destination's pages in and copies); here the DMA engine fills a pinned slot while
// the previous slot is copied on to the caller's buffer by the host threads of a small process-wide pool (up to 16).

// Process-wide pool: parallel_for(n, fn) runs fn(i) for i in [0, n) on the pool's threads and returns when
// all are done.  One caller at a time (callers serialise on `gate`).
class HostPool {
public:
    static HostPool &get() {
        static HostPool *pool = new HostPool();      // never destroyed: its threads sleep until the process ends
        return *pool;
    }
    int size() const { return (int)threads_.size(); }
    void parallel_for(int n, const std::function<void(int)> &fn) {
        if (n <= 0) return;
        std::lock_guard<std::mutex> one(gate_);
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            next_ = 0;
            total_ = n;
            left_ = n;
            ++generation_;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return left_ == 0; });
        fn_ = nullptr;
    }

private:
    HostPool() {
        unsigned hw = std::thread::hardware_concurrency();
        int n = (int)std::min(16u, std::max(1u, hw));
        for (int i = 0; i < n; ++i) threads_.emplace_back([this] { run(); }), threads_.back().detach();
    }
    void run() {
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return generation_ != seen; });
            seen = generation_;
            while (next_ < total_) {
                const int i = next_++;
                const std::function<void(int)> *fn = fn_;
                lk.unlock();
                (*fn)(i);
                lk.lock();
                if (--left_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex gate_, m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    int next_ = 0, total_ = 0, left_ = 0;
    unsigned long long generation_ = 0;
};

// contiguous copy split over the pool (pieces of >= 1 MiB)
inline void parallel_memcpy(void *dst, const void *src, size_t bytes) {
    HostPool &pool = HostPool::get();
    const size_t piece = std::max<size_t>((size_t)1 << 20, (bytes + pool.size() * 4 - 1) / (pool.size() * 4));
    const int n = (int)((bytes + piece - 1) / piece);
    if (n <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    pool.parallel_for(n, [&](int i) {
        const size_t off = (size_t)i * piece;
        std::memcpy((char *)dst + off, (const char *)src + off, std::min(piece, bytes - off));
    });
}

constexpr int kRingSlots = 4;
constexpr size_t kSlotBytes = (size_t)16 << 20;

// One device -> host copy: nrows rows of row_elems doubles, src_pitch elements apart on the device, packed on the host.
struct CopyJob {
    const double *src = nullptr;
    size_t src_pitch = 0, row_elems = 0, nrows = 0;
    double *dst = nullptr;
};

// The handle's copier: a pinned ring, a copy stream and (for the asynchronous jobs of ebm_integrate) a worker
// thread that runs the jobs in order while the caller keeps launching steps.
struct HostCopier {
    int device = 0;
    char *ring = nullptr;                           // kRingSlots x kSlotBytes, pinned
    hipStream_t stream = nullptr;                   // DMA stream (ordered after the compute stream by ev_ready)
    hipEvent_t slot_ev[kRingSlots] = {nullptr};
    hipEvent_t ev_ready = nullptr;
    std::thread worker;
    std::mutex m;
    std::condition_variable cv, idle;
    std::deque<CopyJob> q;
    bool stop = false, busy = false;
    hipError_t err = hipSuccess;

    hipError_t init(int dev) {
        device = dev;
        hipError_t e = hipHostMalloc((void **)&ring, kRingSlots * kSlotBytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        for (int i = 0; i < kRingSlots && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&slot_ev[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming);
        return e;
    }
    void shutdown() {
        if (worker.joinable()) {
            {
                std::lock_guard<std::mutex> lk(m);
                stop = true;
            }
            cv.notify_all();
            worker.join();
        }
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto &ev : slot_ev)
            if (ev) (void)hipEventDestroy(ev);
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (stream) (void)hipStreamDestroy(stream);
        if (ring) (void)hipHostFree(ring);
        ring = nullptr;
        stream = nullptr;
        ev_ready = nullptr;
        for (auto &ev : slot_ev) ev = nullptr;
    }
    // everything the compute stream has been given so far happens before the copies submitted from now on
    hipError_t order_after(hipStream_t compute) {
        hipError_t e = hipEventRecord(ev_ready, compute);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, ev_ready, 0);
        return e;
    }
    // device -> pinned slot -> caller's buffer, the DMA of piece i+1.. in flight while piece i is copied on
    hipError_t run(const CopyJob &j) {
        if (j.nrows == 0 || j.row_elems == 0) return hipSuccess;
        const size_t row_bytes = sizeof(double) * j.row_elems;
        const bool packed = j.src_pitch == j.row_elems;
        // pieces: whole rows when a row fits a slot, else slices of one row
        const size_t rows_per = std::max<size_t>(1, kSlotBytes / row_bytes);
        const size_t slices = row_bytes > kSlotBytes ? (row_bytes + kSlotBytes - 1) / kSlotBytes : 1;
        const size_t npieces = slices > 1 ? j.nrows * slices : (j.nrows + rows_per - 1) / rows_per;
        auto piece = [&](size_t i, const char *&src, char *&dst, size_t &nr, size_t &bytes_per_row) {
            if (slices > 1) {
                const size_t r = i / slices, sl = i % slices, off = sl * kSlotBytes;
                src = (const char *)(j.src + r * j.src_pitch) + off;
                dst = (char *)(j.dst + r * j.row_elems) + off;
                nr = 1;
                bytes_per_row = std::min(kSlotBytes, row_bytes - off);
            } else {
                const size_t r0 = i * rows_per;
                src = (const char *)(j.src + r0 * j.src_pitch);
                dst = (char *)(j.dst + r0 * j.row_elems);
                nr = std::min(rows_per, j.nrows - r0);
                bytes_per_row = row_bytes;
            }
        };
        auto issue = [&](size_t i) -> hipError_t {
            const char *src; char *dst; size_t nr, bpr;
            piece(i, src, dst, nr, bpr);
            char *slot = ring + (i % kRingSlots) * kSlotBytes;
            hipError_t e = (packed || nr == 1)
                ? hipMemcpyAsync(slot, src, nr * bpr, hipMemcpyDeviceToHost, stream)
                : hipMemcpy2DAsync(slot, bpr, src, sizeof(double) * j.src_pitch, bpr, nr, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipEventRecord(slot_ev[i % kRingSlots], stream);
            return e;
        };
        hipError_t e = hipSuccess;
        for (size_t i = 0; i < std::min<size_t>(kRingSlots - 1, npieces) && e == hipSuccess; ++i) e = issue(i);
        for (size_t i = 0; i < npieces && e == hipSuccess; ++i) {
            if (i + kRingSlots - 1 < npieces) e = issue(i + kRingSlots - 1);      // its slot was drained at piece i-1
            if (e == hipSuccess) e = hipEventSynchronize(slot_ev[i % kRingSlots]);
            if (e != hipSuccess) break;
            const char *src; char *dst; size_t nr, bpr;
            piece(i, src, dst, nr, bpr);
            parallel_memcpy(dst, ring + (i % kRingSlots) * kSlotBytes, nr * bpr);
        }
        if (e != hipSuccess) (void)hipStreamSynchronize(stream);
        return e;
    }
    // caller's buffer -> pinned slot -> device
    hipError_t upload(double *dst_dev, size_t dst_pitch, const double *src, size_t row_elems, size_t nrows) {
        const size_t row_bytes = sizeof(double) * row_elems;
        if (row_bytes > kSlotBytes)      // (a single row beyond a slot: not a shape this library has — keep it simple)
            return hipMemcpy2D(dst_dev, sizeof(double) * dst_pitch, src, row_bytes, row_bytes, nrows, hipMemcpyHostToDevice);
        const size_t rows_per = std::max<size_t>(1, kSlotBytes / row_bytes);
        const size_t npieces = (nrows + rows_per - 1) / rows_per;
        hipError_t e = hipSuccess;
        for (size_t i = 0; i < npieces && e == hipSuccess; ++i) {
            const size_t r0 = i * rows_per, nr = std::min(rows_per, nrows - r0);
            char *slot = ring + (i % kRingSlots) * kSlotBytes;
            if (i >= kRingSlots) e = hipEventSynchronize(slot_ev[i % kRingSlots]);      // the slot's previous DMA has read it
            if (e != hipSuccess) break;
            parallel_memcpy(slot, src + r0 * row_elems, nr * row_bytes);
            e = (dst_pitch == row_elems)
                ? hipMemcpyAsync(dst_dev + r0 * dst_pitch, slot, nr * row_bytes, hipMemcpyHostToDevice, stream)
                : hipMemcpy2DAsync(dst_dev + r0 * dst_pitch, sizeof(double) * dst_pitch, slot, row_bytes, row_bytes, nr,
                                   hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipEventRecord(slot_ev[i % kRingSlots], stream);
        }
        hipError_t e2 = hipStreamSynchronize(stream);
        return e != hipSuccess ? e : e2;
    }
    // asynchronous jobs (ebm_integrate): queued, run in order by the worker thread
    void submit(const CopyJob &j) {
        {
            std::lock_guard<std::mutex> lk(m);
            if (!worker.joinable()) worker = std::thread([this] { loop(); });
            q.push_back(j);
        }
        cv.notify_all();
    }
    hipError_t wait_all() {
        std::unique_lock<std::mutex> lk(m);
        idle.wait(lk, [&] { return q.empty() && !busy; });
        hipError_t e = err;
        err = hipSuccess;
        return e;
    }
    void loop() {
        (void)hipSetDevice(device);
        for (;;) {
            CopyJob j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                j = q.front();
                q.pop_front();
                busy = true;
            }
            hipError_t e = run(j);
            {
                std::lock_guard<std::mutex> lk(m);
                if (e != hipSuccess && err == hipSuccess) err = e;
                busy = false;
            }
            idle.notify_all();
        }
    }
};


}  // namespace ebm_host
