// sn_mgpu.hip — multi-GPU form of the C ABI (include/stereonet_hip.h, sn_mgpu_*): independent stereo pairs of one
// batch are sharded contiguously over the GPUs of one node, one host thread + one engine (sn_handle) per GPU, no
// data-path collective; the single exchange is the gather of the int32 / float maps to the root.
//
// Reference semantics: frames are independent units of work — dnn_node keeps task_num = 4 of them in flight
// (stereonet_infer/src/stereonet_node.cpp:144) behind the one Run() call site (:812); this spreads such units over
// devices instead of over BPU task slots.  Built only on the single-GPU C ABI plus HIP peer copies; RCCL (dlopen'ed,
// opt-in with SN_MGPU_GATHER=rccl) replaces the peer copies with one grouped ncclSend/ncclRecv exchange.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/stereonet_hip.h"

namespace {

// ---- RCCL through dlopen: only the six entry points of the grouped send / recv exchange ----------------------
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t st) = nullptr;
  int (*Recv)(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t st) = nullptr;
  bool load() {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return false;
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
  }
};
constexpr int kNcclInt8 = 0;      // ncclInt8 / ncclChar (rccl.h): the maps travel as bytes

struct Worker {
  int dev = 0;
  sn_handle* h = nullptr;
  hipStream_t st = nullptr;          // exchange stream of this device
  int32_t* raw = nullptr;            // local int32 maps of this device's shard (device mode, shards 1..)
  float* disp = nullptr;
  void* comm = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, quit = false;
  int rc = 0;
};

}  // namespace

struct sn_mgpu {
  int ndev = 0, max_batch = 0, per_dev = 0, W = 0, H = 0;
  int gather = 1;                    // 1 = hipMemcpyPeerAsync over xGMI, 2 = RCCL grouped send/recv
  std::vector<Worker*> w;
  Rccl rccl;
  std::string err;
};

namespace {

void worker_main(Worker* w) {
  hipSetDevice(w->dev);
  std::unique_lock<std::mutex> lk(w->mu);
  for (;;) {
    w->cv.wait(lk, [&] { return w->has_job || w->quit; });
    if (w->quit) return;
    std::function<int()> job = std::move(w->job);
    w->has_job = false;
    lk.unlock();
    const int rc = job();
    lk.lock();
    w->rc = rc;
    w->done = true;
    w->cv.notify_all();
  }
}

void post(Worker* w, std::function<int()> job) {
  std::lock_guard<std::mutex> lk(w->mu);
  w->job = std::move(job);
  w->has_job = true;
  w->done = false;
  w->cv.notify_all();
}

int wait_done(Worker* w) {
  std::unique_lock<std::mutex> lk(w->mu);
  w->cv.wait(lk, [&] { return w->done; });
  return w->rc;
}

// runs job(k) on every worker thread, returns the first failure
int run_all(sn_mgpu* m, const std::function<int(int)>& job) {
  for (int k = 0; k < m->ndev; ++k) post(m->w[k], [&job, k] { return job(k); });
  int rc = SN_OK;
  for (int k = 0; k < m->ndev; ++k) {
    const int r = wait_done(m->w[k]);
    if (r != SN_OK && rc == SN_OK) {
      rc = r;
      const char* d = m->w[k]->h ? sn_last_error(m->w[k]->h) : "";
      m->err = "device " + std::to_string(m->w[k]->dev) + ": " + sn_strerror(r) + (d && *d ? std::string(": ") + d : std::string());
    }
  }
  return rc;
}

}  // namespace

extern "C" {

// Contiguous shard of n pairs for device k of ndev: the first n % ndev shards get one extra pair (the same rule as
// hobot_stereonet_amd/dist.py::shard_range, so both multi-GPU forms cut a batch identically).
int sn_mgpu_shard(int n, int ndev, int k, int* first, int* count) {
  if (n < 0 || ndev <= 0 || k < 0 || k >= ndev || !first || !count) return SN_ERR_ARG;
  const int q = n / ndev, r = n % ndev;
  *first = k * q + (k < r ? k : r);
  *count = q + (k < r ? 1 : 0);
  return SN_OK;
}

const char* sn_mgpu_last_error(const sn_mgpu* m) { return m ? m->err.c_str() : ""; }

int sn_mgpu_destroy(sn_mgpu* m) {
  if (!m) return SN_ERR_ARG;
  for (Worker* w : m->w) {
    if (w->th.joinable()) {
      {
        std::lock_guard<std::mutex> lk(w->mu);
        w->quit = true;
        w->cv.notify_all();
      }
      w->th.join();
    }
    hipSetDevice(w->dev);
    if (w->comm && m->rccl.CommDestroy) m->rccl.CommDestroy(w->comm);
    if (w->raw) hipFree(w->raw);
    if (w->disp) hipFree(w->disp);
    if (w->st) hipStreamDestroy(w->st);
    if (w->h) sn_destroy(w->h);
    delete w;
  }
  delete m;
  return SN_OK;
}

int sn_mgpu_create(const char* model_file, const sn_config* cfg, const int* devices, int ndev, sn_mgpu** out) {
  if (!model_file || !out || ndev <= 0 || ndev > 64) return SN_ERR_ARG;
  *out = nullptr;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return SN_ERR_DEVICE;
  for (int k = 0; k < ndev; ++k) {
    const int d = devices ? devices[k] : k;
    if (d < 0 || d >= have) return SN_ERR_ARG;
    for (int j = 0; j < k; ++j)
      if ((devices ? devices[j] : j) == d) return SN_ERR_ARG;      // a device may hold one shard only
  }
  sn_mgpu* m = new sn_mgpu();
  m->ndev = ndev;
  sn_config c{};
  if (cfg) c = *cfg;
  m->max_batch = c.max_batch > 0 ? c.max_batch : ndev;
  m->per_dev = (m->max_batch + ndev - 1) / ndev;
  if (const char* e = getenv("SN_MGPU_GATHER")) m->gather = !strcmp(e, "rccl") ? 2 : 1;
  if (m->gather == 2 && (ndev == 1 || !m->rccl.load())) m->gather = 1;
  int rc = SN_OK;
  for (int k = 0; k < ndev && rc == SN_OK; ++k) {
    Worker* w = new Worker();
    m->w.push_back(w);
    w->dev = devices ? devices[k] : k;
    sn_config ck = c;
    ck.device = w->dev;
    ck.max_batch = m->per_dev;
    rc = sn_create(model_file, &ck, &w->h);
    if (rc != SN_OK) {
      const char* d = sn_last_error(nullptr);
      m->err = "sn_create on device " + std::to_string(w->dev) + ": " + sn_strerror(rc) + (d && *d ? std::string(": ") + d : std::string());
      break;
    }
    sn_io_info info;
    sn_get_io_info(w->h, &info);
    m->W = info.width;
    m->H = info.height;
    if (hipSetDevice(w->dev) != hipSuccess || hipStreamCreateWithFlags(&w->st, hipStreamNonBlocking) != hipSuccess) rc = SN_ERR_DEVICE;
    if (rc == SN_OK && k > 0) {          // staging for the maps that travel to the root in device mode
      const size_t bytes = (size_t)m->per_dev * m->W * m->H * 4;
      if (hipMalloc(reinterpret_cast<void**>(&w->raw), bytes) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&w->disp), bytes) != hipSuccess)
        rc = SN_ERR_NOMEM;
      // direct xGMI copies in both directions between this device and the root (an "already enabled" error is fine)
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, w->dev, m->w[0]->dev) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(m->w[0]->dev, 0);
      hipSetDevice(m->w[0]->dev);
      if (hipDeviceCanAccessPeer(&can, m->w[0]->dev, w->dev) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(w->dev, 0);
      (void)hipGetLastError();
    }
  }
  if (rc == SN_OK && m->gather == 2) {
    std::vector<void*> comms(ndev, nullptr);
    std::vector<int> devs(ndev);
    for (int k = 0; k < ndev; ++k) devs[k] = m->w[k]->dev;
    if (m->rccl.CommInitAll(comms.data(), ndev, devs.data()) != 0) {
      m->gather = 1;                     // RCCL would not initialise: peer copies carry the gather
    } else {
      for (int k = 0; k < ndev; ++k) m->w[k]->comm = comms[k];
    }
  }
  if (rc != SN_OK) {
    std::string keep = m->err;
    fprintf(stderr, "sn_mgpu_create: %s\n", keep.c_str());
    sn_mgpu_destroy(m);
    return rc;
  }
  for (Worker* w : m->w) w->th = std::thread(worker_main, w);
  *out = m;
  return SN_OK;
}

int sn_mgpu_get_info(const sn_mgpu* m, int* ndev, int* per_device_batch, int* gather_kind) {
  if (!m) return SN_ERR_ARG;
  if (ndev) *ndev = m->ndev;
  if (per_device_batch) *per_device_batch = m->per_dev;
  if (gather_kind) *gather_kind = m->gather;
  return SN_OK;
}

int sn_mgpu_get_handle(sn_mgpu* m, int k, sn_handle** h) {
  if (!m || !h || k < 0 || k >= m->ndev) return SN_ERR_ARG;
  *h = m->w[k]->h;
  return SN_OK;
}

// Host buffers: every device copies its shard in, runs it, and copies its maps straight into the caller's arrays —
// the host is the gather root, nothing crosses between devices.
int sn_mgpu_infer_batch(sn_mgpu* m, int n, const int8_t* in, int32_t* out_i32, float* out_disp) {
  if (!m) return SN_ERR_ARG;
  if (!in || (!out_i32 && !out_disp) || n <= 0 || n > m->max_batch) {
    m->err = "sn_mgpu_infer_batch: bad arguments";
    return SN_ERR_ARG;
  }
  const size_t HW = (size_t)m->W * m->H;
  return run_all(m, [&](int k) -> int {
    int first = 0, cnt = 0;
    sn_mgpu_shard(n, m->ndev, k, &first, &cnt);
    if (cnt == 0) return SN_OK;
    return sn_infer_batch(m->w[k]->h, cnt, in + (size_t)first * 6 * HW, out_i32 ? out_i32 + (size_t)first * HW : nullptr,
                          out_disp ? out_disp + (size_t)first * HW : nullptr, SN_MEM_HOST, nullptr);
  });
}

// Device buffers: in_per_device[k] holds shard k's pairs in the memory of device k; the int32 / float maps of all n
// pairs are gathered, in batch order, into out_* in the memory of device 0 (the root) over xGMI.
int sn_mgpu_infer_batch_device(sn_mgpu* m, int n, const int8_t* const* in_per_device, int32_t* out_i32_root,
                               float* out_disp_root) {
  if (!m) return SN_ERR_ARG;
  if (!in_per_device || (!out_i32_root && !out_disp_root) || n <= 0 || n > m->max_batch) {
    m->err = "sn_mgpu_infer_batch_device: bad arguments";
    return SN_ERR_ARG;
  }
  const size_t HW = (size_t)m->W * m->H;
  const int root = m->w[0]->dev;
  return run_all(m, [&](int k) -> int {
    Worker* w = m->w[k];
    int first = 0, cnt = 0;
    sn_mgpu_shard(n, m->ndev, k, &first, &cnt);
    int32_t* raw = !out_i32_root ? nullptr : (k == 0 ? out_i32_root : w->raw);
    float* disp = !out_disp_root ? nullptr : (k == 0 ? out_disp_root : w->disp);
    if (cnt > 0) {
      if (!in_per_device[k]) return SN_ERR_ARG;
      const int rc = sn_infer_batch(w->h, cnt, in_per_device[k], raw, disp, SN_MEM_DEVICE, nullptr);   // synchronous
      if (rc != SN_OK) return rc;
    }
    if (m->ndev == 1) return SN_OK;
    if (m->gather == 2) {              // one grouped exchange: root posts a recv per peer, every peer one send per map kind
      bool ok = m->rccl.GroupStart() == 0;
      for (int kind = 0; kind < 2 && ok; ++kind) {
        char* root_buf = reinterpret_cast<char*>(kind == 0 ? (void*)out_i32_root : (void*)out_disp_root);
        const void* mine = kind == 0 ? (const void*)raw : (const void*)disp;
        if (!root_buf) continue;
        if (k == 0) {
          for (int r = 1; r < m->ndev && ok; ++r) {
            int f = 0, c = 0;
            sn_mgpu_shard(n, m->ndev, r, &f, &c);
            if (c > 0) ok = m->rccl.Recv(root_buf + (size_t)f * HW * 4, (size_t)c * HW * 4, kNcclInt8, r, w->comm, w->st) == 0;
          }
        } else if (cnt > 0) {
          ok = m->rccl.Send(mine, (size_t)cnt * HW * 4, kNcclInt8, 0, w->comm, w->st) == 0;
        }
      }
      ok = (m->rccl.GroupEnd() == 0) && ok;
      if (!ok || hipStreamSynchronize(w->st) != hipSuccess) return SN_ERR_DEVICE;
      return SN_OK;
    }
    if (k > 0 && cnt > 0) {            // peer copies: every non-root device pushes its maps over its own link
      if (raw && hipMemcpyPeerAsync(out_i32_root + (size_t)first * HW, root, raw, w->dev, (size_t)cnt * HW * 4, w->st) != hipSuccess)
        return SN_ERR_DEVICE;
      if (disp && hipMemcpyPeerAsync(out_disp_root + (size_t)first * HW, root, disp, w->dev, (size_t)cnt * HW * 4, w->st) != hipSuccess)
        return SN_ERR_DEVICE;
      if (hipStreamSynchronize(w->st) != hipSuccess) return SN_ERR_DEVICE;
    }
    return SN_OK;
  });
}

}  // extern "C"
