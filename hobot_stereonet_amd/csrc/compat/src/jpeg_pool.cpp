// jpeg_pool.cpp — see include/jpeg_pool.h: the slice tasks of one frame and the assembly of its stream.  Depends on the
// encoder only (no node, no HIP), so that test/jpeg_pool_race.cpp can run it under ThreadSanitizer.
#include "jpeg_pool.h"

#include <atomic>

#include "jpeg_nv12.h"

namespace hobot {
namespace stereonet {

std::shared_future<bool> SubmitSlicedJpeg(JpegPool& pool, std::shared_ptr<const void> keep_alive, const uint8_t* nv12, int w, int h,
                                          int pitch, int quality, int slices, std::shared_ptr<BinDataType> out) {
  struct Sliced {
    std::vector<std::vector<uint8_t>> part;
    std::atomic<int> left_to_do{0};
    std::atomic<bool> ok{true};
    std::promise<bool> done;
  };
  const int rows = JpegMcuRows(h);
  int nsl = slices < 1 ? 1 : slices;
  if (nsl > rows) nsl = rows;
  const int per = (rows + nsl - 1) / nsl;
  nsl = (rows + per - 1) / per;
  auto st = std::make_shared<Sliced>();
  st->part.resize((size_t)nsl);
  st->left_to_do = nsl;
  std::shared_future<bool> fut = st->done.get_future().share();
  for (int k = 0; k < nsl; ++k)
    pool.Post([keep_alive, nv12, out, st, w, h, pitch, quality, per, rows, nsl, k] {
      const int r0 = k * per, r1 = r0 + per < rows ? r0 + per : rows;
      if (!JpegAppendMcuRows(nv12, w, h, pitch, quality, r0, r1, st->part[(size_t)k])) st->ok = false;
      if (--st->left_to_do != 0) return;
      out->jpeg.clear();
      bool ok = st->ok && JpegAppendHeader(w, h, quality, nsl > 1 ? per * ((w + 15) / 16) : 0, out->jpeg);
      if (ok) {
        size_t total = out->jpeg.size();
        for (const auto& p : st->part) total += p.size() + 2;
        out->jpeg.reserve(total);
        for (int i = 0; i < nsl; ++i) {
          out->jpeg.insert(out->jpeg.end(), st->part[(size_t)i].begin(), st->part[(size_t)i].end());
          out->jpeg.push_back(0xFF);
          out->jpeg.push_back(i + 1 < nsl ? (uint8_t)(0xD0 + (i & 7)) : (uint8_t)0xD9);
        }
      }
      st->done.set_value(ok);
    });
  return fut;
}

}  // namespace stereonet
}  // namespace hobot
