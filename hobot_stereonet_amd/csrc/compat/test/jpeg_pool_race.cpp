// jpeg_pool_race.cpp — the node's encoder threads under ThreadSanitizer (make tsan): frames x slices in flight on a pool,
// several submitting threads (the executor thread is single in the node; more here to stress the queue), every assembled
// stream compared with the sequential EncodeNv12ToJpegSliced.  Exit code 0 = all streams equal and TSan silent.
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "jpeg_nv12.h"
#include "jpeg_pool.h"

using namespace hobot::stereonet;

int main() {
  const int w = 320, h = 176, pitch = 2 * w;
  std::vector<uint8_t> sbs((size_t)pitch * h * 3 / 2);
  unsigned x = 12345u;
  for (auto& b : sbs) {
    x = x * 1664525u + 1013904223u;
    b = (uint8_t)(128 + (int)((x >> 24) % 64) - 32);
  }
  int bad = 0;
  for (int slices : {1, 4, 11}) {
    const int rows = JpegMcuRows(h);
    const int nsl = slices > rows ? rows : slices, per = (rows + nsl - 1) / nsl;
    std::vector<uint8_t> want;
    if (!EncodeNv12ToJpegSliced(sbs.data(), w, h, pitch, 90, per, want)) return 2;
    JpegPool pool(6);
    std::vector<std::thread> feeders;
    std::atomic<int> mismatches{0};
    for (int t = 0; t < 3; ++t)
      feeders.emplace_back([&] {
        std::vector<std::shared_ptr<BinDataType>> outs;
        std::vector<std::shared_future<bool>> futs;
        for (int f = 0; f < 20; ++f) {
          outs.push_back(std::make_shared<BinDataType>());
          futs.push_back(SubmitSlicedJpeg(pool, nullptr, sbs.data(), w, h, pitch, 90, slices, outs.back()));
        }
        for (size_t i = 0; i < futs.size(); ++i)
          if (!futs[i].get() || outs[i]->jpeg != want) ++mismatches;
      });
    for (auto& t : feeders) t.join();
    printf("slices %2d: %d mismatching streams of 60\n", slices, (int)mismatches);
    bad += mismatches;
  }
  return bad ? 1 : 0;
}
