// Minimal in-process stand-in for rclcpp, used ONLY when ROS 2 is not installed (this image, the GPU
// box) so that StereonetNode and the dnn_node compat layer can be compiled and exercised by tests.
// With a real ROS 2 workspace the build uses the real <rclcpp/rclcpp.hpp> instead (compat/Makefile,
// INTEGRATION.md).  Semantics: publish() delivers synchronously to every subscription of the topic in
// this process (the caller plays the executor thread); parameters come from NodeOptions overrides.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

namespace rclcpp {

inline std::atomic<bool>& ok_flag() {
  static std::atomic<bool> f{true};
  return f;
}
inline void init(int, char**) { ok_flag() = true; }
inline bool ok() { return ok_flag(); }
inline void shutdown() { ok_flag() = false; }

struct Logger {
  std::string name;
};
inline Logger get_logger(const std::string& name) { return Logger{name}; }
inline int log_threshold() {   // 0 debug, 1 info, 2 warn, 3 error
  static int t = [] {
    const char* e = getenv("SN_LOG_LEVEL");
    return e ? atoi(e) : 2;
  }();
  return t;
}
inline void log_line(int level, const Logger& lg, const std::string& msg) {
  if (level < log_threshold()) return;
  static const char* tag[] = {"DEBUG", "INFO", "WARN", "ERROR"};
  fprintf(stderr, "[%s] [%s]: %s\n", tag[level], lg.name.c_str(), msg.c_str());
}
inline std::string log_fmt(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return buf;
}

class NodeOptions {
 public:
  NodeOptions& append_parameter_override(const std::string& name, const std::string& value) {
    overrides_[name] = value;
    return *this;
  }
  const std::map<std::string, std::string>& overrides() const { return overrides_; }

 private:
  std::map<std::string, std::string> overrides_;
};

// ---- in-process topic bus ---------------------------------------------------------------------------
template <class Msg>
struct Bus {
  using Callback = std::function<void(std::shared_ptr<const Msg>)>;
  static std::mutex& mu() {
    static std::mutex m;
    return m;
  }
  static std::map<std::string, std::vector<std::pair<int, Callback>>>& subs() {
    static std::map<std::string, std::vector<std::pair<int, Callback>>> s;
    return s;
  }
};
inline std::string norm_topic(const std::string& t) { return (!t.empty() && t[0] == '/') ? t.substr(1) : t; }

template <class Msg>
class Subscription {
 public:
  using SharedPtr = std::shared_ptr<Subscription<Msg>>;
  using ConstSharedPtr = std::shared_ptr<const Subscription<Msg>>;
  Subscription(const std::string& topic, typename Bus<Msg>::Callback cb) : topic_(norm_topic(topic)) {
    static std::atomic<int> next{1};
    id_ = next++;
    std::lock_guard<std::mutex> lk(Bus<Msg>::mu());
    Bus<Msg>::subs()[topic_].emplace_back(id_, std::move(cb));
  }
  ~Subscription() {
    std::lock_guard<std::mutex> lk(Bus<Msg>::mu());
    auto& v = Bus<Msg>::subs()[topic_];
    for (size_t i = 0; i < v.size(); ++i)
      if (v[i].first == id_) {
        v.erase(v.begin() + i);
        break;
      }
  }

 private:
  std::string topic_;
  int id_;
};

template <class Msg>
class Publisher {
 public:
  using SharedPtr = std::shared_ptr<Publisher<Msg>>;
  explicit Publisher(const std::string& topic) : topic_(norm_topic(topic)) {}
  void publish(const Msg& m) const { deliver(std::make_shared<const Msg>(m)); }
  void publish(Msg&& m) const { deliver(std::make_shared<const Msg>(std::move(m))); }
  void publish(std::unique_ptr<Msg> m) const { deliver(std::shared_ptr<const Msg>(std::move(m))); }
  // stand-in for a zero-copy transport (hbmem shared memory, loaned messages): hands an existing message over as is
  void publish_shared(std::shared_ptr<const Msg> m) const { deliver(std::move(m)); }
  const std::string& get_topic_name() const { return topic_; }

 private:
  void deliver(std::shared_ptr<const Msg> m) const {
    std::vector<typename Bus<Msg>::Callback> cbs;
    {
      std::lock_guard<std::mutex> lk(Bus<Msg>::mu());
      for (auto& s : Bus<Msg>::subs()[topic_]) cbs.push_back(s.second);
    }
    for (auto& cb : cbs) cb(m);
  }
  std::string topic_;
};

class TimerBase {
 public:
  using SharedPtr = std::shared_ptr<TimerBase>;
};

class Node {
 public:
  using SharedPtr = std::shared_ptr<Node>;
  explicit Node(const std::string& name, const NodeOptions& options = NodeOptions())
      : name_(name), overrides_(options.overrides()) {}
  virtual ~Node() = default;
  const char* get_name() const { return name_.c_str(); }
  Logger get_logger() const { return Logger{name_}; }

  template <class T>
  void declare_parameter(const std::string& name, const T& default_value) {
    if (params_.count(name)) return;
    auto it = overrides_.find(name);
    if (it != overrides_.end()) {
      params_[name] = it->second;
    } else {
      std::ostringstream ss;
      ss << default_value;
      params_[name] = ss.str();
    }
  }
  template <class T>
  bool get_parameter(const std::string& name, T& out) const {
    auto it = params_.find(name);
    if (it == params_.end()) return false;
    assign(out, it->second);
    return true;
  }

  template <class Msg, class Cb>
  typename Subscription<Msg>::SharedPtr create_subscription(const std::string& topic, int /*qos_depth*/, Cb&& cb) {
    return std::make_shared<Subscription<Msg>>(topic, typename Bus<Msg>::Callback(std::forward<Cb>(cb)));
  }
  template <class Msg>
  typename Publisher<Msg>::SharedPtr create_publisher(const std::string& topic, int /*qos_depth*/) {
    return std::make_shared<Publisher<Msg>>(topic);
  }

 private:
  template <class T>
  static void assign(T& o, const std::string& v) {
    std::istringstream ss(v);
    ss >> o;
  }
  static void assign(std::string& o, const std::string& v) { o = v; }
  std::string name_;
  std::map<std::string, std::string> overrides_;
  std::map<std::string, std::string> params_;
};

template <class T>
inline void spin(std::shared_ptr<T>) {
  while (ok()) std::this_thread::sleep_for(std::chrono::milliseconds(10));
}

}  // namespace rclcpp

#define RCLCPP_LOG_(level, logger, ...) ::rclcpp::log_line(level, logger, ::rclcpp::log_fmt(__VA_ARGS__))
#define RCLCPP_DEBUG(logger, ...) RCLCPP_LOG_(0, logger, __VA_ARGS__)
#define RCLCPP_INFO(logger, ...) RCLCPP_LOG_(1, logger, __VA_ARGS__)
#define RCLCPP_WARN(logger, ...) RCLCPP_LOG_(2, logger, __VA_ARGS__)
#define RCLCPP_ERROR(logger, ...) RCLCPP_LOG_(3, logger, __VA_ARGS__)
#define RCLCPP_STREAM_(level, logger, expr)               \
  do {                                                    \
    std::ostringstream rclcpp_ss_;                        \
    rclcpp_ss_ << expr;                                   \
    ::rclcpp::log_line(level, logger, rclcpp_ss_.str());  \
  } while (0)
#define RCLCPP_DEBUG_STREAM(logger, expr) RCLCPP_STREAM_(0, logger, expr)
#define RCLCPP_INFO_STREAM(logger, expr) RCLCPP_STREAM_(1, logger, expr)
#define RCLCPP_WARN_STREAM(logger, expr) RCLCPP_STREAM_(2, logger, expr)
#define RCLCPP_ERROR_STREAM(logger, expr) RCLCPP_STREAM_(3, logger, expr)
