// jpeg_pool.h — encoder threads of the node's left-eye JPEG (stereonet_infer/src/stereonet_node.cpp:749-786 does the encode
// inline in FeedImg).
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "bin_data.h"

namespace hobot {
namespace stereonet {

// The reference encodes the left eye on the executor thread inside FeedImg (stereonet_node.cpp:749-786), which is fine at
// a 30 fps camera and a 15-40 ms encoder but serialises a backend that finishes a pair in 0.5 ms.  Here FeedImg only queues
// the encode: a fixed set of worker threads works through the queue, several frames at a time, and PostProcess — which
// the completion thread calls in request order — waits for the request's own JPEG.  The queue is bounded (back-pressure
// on the executor thread) so that frames cannot pile up behind a slow encoder.
class JpegPool {
 public:
  explicit JpegPool(int threads) : cap_(4 * (size_t)threads + 16) {
    for (int i = 0; i < threads; ++i) th_.emplace_back([this] { Loop(); });
  }
  ~JpegPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void Post(std::function<void()> job) {
    {
      std::unique_lock<std::mutex> lk(mu_);
      room_.wait(lk, [&] { return q_.size() < cap_ || stop_; });
      q_.push_back(std::move(job));
    }
    cv_.notify_one();
  }
  int threads() const { return (int)th_.size(); }

 private:
  void Loop() {
    for (;;) {
      std::function<void()> task;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        task = std::move(q_.front());
        q_.pop_front();
      }
      room_.notify_one();
      task();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, room_;
  std::deque<std::function<void()>> q_;
  std::vector<std::thread> th_;
  const size_t cap_;
  bool stop_ = false;
};


// One frame = several slices of MCU rows behind restart markers, one pool task each: with task_num = 4 requests in
// flight, whole-frame tasks would keep at most ~5 encoder threads busy and PostProcess would wait for the JPEG (measured:
// 13.6 ms per frame and thread -> 400 frames/s whatever the pool size).  The task that finishes last assembles the
// stream (header, slice, RSTm, slice, ..., EOI) into out->jpeg and completes the returned future; nobody waits inside
// the pool.  `keep_alive` owns the memory `nv12` points into (the subscription's message) until the last slice is done.
// The stream is byte for byte EncodeNv12ToJpegSliced(nv12, ..., rows_per_slice = ceil(MCU rows / slices)).
std::shared_future<bool> SubmitSlicedJpeg(JpegPool& pool, std::shared_ptr<const void> keep_alive, const uint8_t* nv12, int w, int h,
                                          int pitch, int quality, int slices, std::shared_ptr<BinDataType> out);

}  // namespace stereonet
}  // namespace hobot
