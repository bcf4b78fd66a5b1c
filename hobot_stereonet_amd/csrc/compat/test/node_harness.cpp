// node_harness.cpp — drives StereonetNode through the in-process rclcpp stand-in (no ROS 2 in this image):
// publishes side-by-side NV12 frames on the hbmem topic and records what the node publishes.
//   node_harness <model.snw> <sbs_nv12.bin> <w> <h> <nframes> <out_prefix> [parse]
// writes <out_prefix>.<i>.msg (payload bytes) and prints one line per frame.  Exit code 3 = Init failed
// (e.g. no GPU / missing model), the reference's "Node init fail!" path.
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <vector>

#include "parser.h"
#include "stereonet_node.h"

using hobot::stereonet::StereonetNode;

// node_harness --bench <model.snw> <sbs_nv12.bin> <w> <h> <nframes>
// Node-level throughput of the FeedImg -> Run -> PostProcess -> publish path (stereonet_node.cpp:657-818, 980-1089): the
// harness plays the camera (one executor thread publishing side-by-side NV12 frames back to back; publish() returns when
// FeedImg has queued the request, and blocks while all task slots are busy) and a subscriber of the output topic.  Prints
// one JSON line.  STEREONET_PUB_OUTPUT=0 measures the node without the wire message (no JPEG, nothing published).
static int bench_main(int argc, char** argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s --bench model sbs.bin w h nframes\n", argv[0]);
    return 2;
  }
  const std::string model = argv[2], sbs_path = argv[3];
  const int w = atoi(argv[4]), h = atoi(argv[5]), nframes = atoi(argv[6]);
  rclcpp::init(argc, argv);
  rclcpp::NodeOptions opt;
  opt.append_parameter_override("model_file", model);
  auto node = std::make_shared<StereonetNode>("stereonet_node", opt);
  if (!rclcpp::ok() || !node->IsReady()) {
    fprintf(stderr, "node init failed\n");
    return 3;
  }
  std::vector<uint8_t> sbs((size_t)2 * w * h * 3 / 2);
  {
    std::ifstream f(sbs_path, std::ios::binary);
    f.read(reinterpret_cast<char*>(sbs.data()), sbs.size());
    if ((size_t)f.gcount() != sbs.size()) return 2;
  }
  const bool pub_on = !(getenv("STEREONET_PUB_OUTPUT") && atoi(getenv("STEREONET_PUB_OUTPUT")) == 0);
  std::mutex mu;
  std::condition_variable cv;
  long received = 0, bytes = 0;
  rclcpp::Node listener("listener");
  auto sub = listener.create_subscription<sensor_msgs::msg::Image>(
      "stereonet_node_output", 10, [&](sensor_msgs::msg::Image::ConstSharedPtr m) {
        std::lock_guard<std::mutex> lk(mu);
        ++received;
        bytes += (long)m->data.size();
        cv.notify_all();
      });
  auto pub = listener.create_publisher<hbm_img_msgs::msg::HbmMsg1080P>("hbmem_stereo_img", 10);
  // the camera's messages are built once (a camera node fills shared memory, it does not copy 2.76 MB per publish here);
  // a few distinct ones so that consecutive requests do not share a payload
  std::vector<std::shared_ptr<hbm_img_msgs::msg::HbmMsg1080P>> msgs(4);
  for (size_t k = 0; k < msgs.size(); ++k) {
    msgs[k] = std::make_shared<hbm_img_msgs::msg::HbmMsg1080P>();
    auto& m = *msgs[k];
    m.index = 100 + (uint32_t)k;
    m.time_stamp.sec = 7;
    m.height = h;
    m.width = 2 * w;
    m.data_size = (uint32_t)sbs.size();
    memcpy(m.encoding.data(), "nv12", 5);
    m.data = sbs;
    for (size_t i = k; i < (size_t)w; i += 97) m.data[i] ^= (uint8_t)(k + 1);
  }
  auto run = [&](int n) {
    const long base = received;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
      pub->publish_shared(msgs[i % msgs.size()]);      // the hbmem transport hands the node a shared buffer, not a copy
    }
    if (pub_on) {
      std::unique_lock<std::mutex> lk(mu);
      if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return received - base >= n; })) return -1.0;
    } else {
      node->WaitIdle();
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  if (run(8) < 0) return 4;                      // warm-up: first use of every task slot, graph capture
  const double dt = run(nframes);
  if (dt < 0) {
    fprintf(stderr, "timeout\n");
    return 4;
  }
  printf("{\"node_bench\": true, \"frames\": %d, \"seconds\": %.4f, \"frames_per_s\": %.1f, \"ms_per_frame\": %.4f, "
         "\"publish\": %s, \"width\": %d, \"height\": %d, \"payload_bytes_per_frame\": %ld, \"jpeg_threads\": \"%s\"}\n",
         nframes, dt, nframes / dt, dt / nframes * 1e3, pub_on ? "true" : "false", w, h, pub_on && received ? bytes / received : 0,
         getenv("STEREONET_JPEG_THREADS") ? getenv("STEREONET_JPEG_THREADS") : "auto");
  node.reset();
  rclcpp::shutdown();
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "--bench")) return bench_main(argc, argv);
  if (argc < 7) {
    fprintf(stderr, "usage: %s model sbs.bin w h nframes out_prefix\n", argv[0]);
    return 2;
  }
  const std::string model = argv[1], sbs_path = argv[2], prefix = argv[6];
  const int w = atoi(argv[3]), h = atoi(argv[4]), nframes = atoi(argv[5]);
  rclcpp::init(argc, argv);
  rclcpp::NodeOptions opt;
  opt.append_parameter_override("model_file", model);
  auto node = std::make_shared<StereonetNode>("stereonet_node", opt);
  if (!rclcpp::ok() || !node->IsReady()) {
    fprintf(stderr, "node init failed\n");
    return 3;
  }
  std::vector<uint8_t> sbs((size_t)2 * w * h * 3 / 2);
  {
    std::ifstream f(sbs_path, std::ios::binary);
    f.read(reinterpret_cast<char*>(sbs.data()), sbs.size());
    if ((size_t)f.gcount() != sbs.size()) return 2;
  }
  std::mutex mu;
  std::condition_variable cv;
  int received = 0;
  rclcpp::Node listener("listener");
  auto sub = listener.create_subscription<sensor_msgs::msg::Image>(
      "stereonet_node_output", 10, [&](sensor_msgs::msg::Image::ConstSharedPtr m) {
        std::lock_guard<std::mutex> lk(mu);
        std::ofstream o(prefix + "." + std::to_string(received) + ".msg", std::ios::binary);
        o.write(reinterpret_cast<const char*>(m->data.data()), m->data.size());
        printf("frame_id=%s height=%u width=%u encoding=%s step=%u len=%zu stamp=%d.%u\n", m->header.frame_id.c_str(),
               m->height, m->width, m->encoding.c_str(), m->step, m->data.size(), m->header.stamp.sec,
               m->header.stamp.nanosec);
        ++received;
        cv.notify_all();
      });
  auto pub = listener.create_publisher<hbm_img_msgs::msg::HbmMsg1080P>("hbmem_stereo_img", 10);
  // negative cases the reference rejects (stereonet_node.cpp:672-690): wrong encoding, wrong geometry
  {
    hbm_img_msgs::msg::HbmMsg1080P bad;
    bad.height = h;
    bad.width = 2 * w;
    memcpy(bad.encoding.data(), "bgr8", 5);
    bad.data = sbs;
    pub->publish(bad);
    memcpy(bad.encoding.data(), "nv12", 5);
    bad.width = w;
    pub->publish(bad);
  }
  for (int i = 0; i < nframes; ++i) {
    hbm_img_msgs::msg::HbmMsg1080P m;
    m.index = 100 + i;
    m.time_stamp.sec = 7;
    m.time_stamp.nanosec = 1000 + i;
    m.height = h;
    m.width = 2 * w;
    m.data_size = (uint32_t)sbs.size();
    memcpy(m.encoding.data(), "nv12", 5);
    m.data = sbs;
    pub->publish(m);   // FeedImg runs here (executor thread); Run() is asynchronous
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return received >= nframes; })) {
      fprintf(stderr, "timeout: %d of %d frames\n", received, nframes);
      return 4;
    }
  }
  printf("received=%d\n", received);
  node.reset();
  rclcpp::shutdown();
  return 0;
}
