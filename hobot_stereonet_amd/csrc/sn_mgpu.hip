// sn_mgpu.hip — multi-GPU form of the C ABI (include/stereonet_hip.h, sn_mgpu_*): independent stereo pairs of one
// batch are sharded contiguously over the GPUs of one node, one host thread + one engine (sn_handle) per GPU, no
// data-path collective; the single exchange is the gather of the int32 / float maps to the root.
//
// Reference semantics: frames are independent units of work — dnn_node keeps task_num = 4 of them in flight
// (stereonet_infer/src/stereonet_node.cpp:144) behind the one Run() call site (:812); this spreads such units over
// devices instead of over BPU task slots.  Built only on the single-GPU C ABI; the gather is one grouped RCCL
// ncclSend/ncclRecv exchange per batch (RCCL is dlopen'ed; the default whenever more than one distinct device takes
// part and the library loads), with HIP peer copies as the fallback (SN_MGPU_GATHER=peer forces them).
// Device-resident batches are ASYNCHRONOUS and double buffered: sn_mgpu_submit_device enqueues shard compute on each
// device's compute stream and the exchange on its exchange stream behind an event, and returns a ticket; two tickets
// may be in flight, so the gather of batch k crosses xGMI while batch k+1 computes (the reference's async Run with
// task slots, stereonet_node.cpp:144,812).  sn_mgpu_infer_batch_device = submit + wait.
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
#include "sn_internal.h"

namespace {

// ---- RCCL through dlopen: only the entry points of the grouped send / recv exchange (+ ncclCommAbort) ----------------------
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*CommAbort)(void* comm) = nullptr;      // frees a communicator whose exchange can no longer complete
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
    CommAbort = reinterpret_cast<decltype(CommAbort)>(sym("ncclCommAbort"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    // ncclCommAbort is REQUIRED: it is the only way to release an exchange whose other half never arrives (the error path of
    // sn_mgpu_submit_device); ncclCommDestroy on such a communicator may block on the outstanding operation.  A library
    // without it does not carry the gather (peer copies do).
    return CommInitAll && CommDestroy && CommAbort && GroupStart && GroupEnd && Send && Recv;
  }
};
constexpr int kNcclInt8 = 0;      // ncclInt8 / ncclChar (rccl.h): the maps travel as bytes

constexpr int kSlots = 2;            // batches in flight per sn_mgpu (double buffering)

struct Worker {
  int dev = 0;
  sn_handle* h = nullptr;
  hipStream_t st = nullptr;          // exchange stream of this device
  hipStream_t cs = nullptr;          // compute stream of this device (device-resident batches)
  int32_t* raw[kSlots] = {nullptr, nullptr};     // local int32 maps of this device's shard (device mode, shards 1..)
  float* disp[kSlots] = {nullptr, nullptr};
  hipEvent_t ev_compute[kSlots] = {nullptr, nullptr}, ev_done[kSlots] = {nullptr, nullptr};
  void* comm = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, quit = false;
  int rc = 0;
};

}  // namespace

// All workers agree on one status between "my shard's compute is enqueued" and "I join the exchange": a failed or
// missing shard must keep EVERY rank out of the grouped RCCL exchange, otherwise the root's posted recv never
// completes and the call hangs instead of returning an error.
struct StatusBarrier {
  std::mutex mu;
  std::condition_variable cv;
  int n = 0, arrived = 0, rc = 0, result = 0;
  unsigned long long gen = 0;
  int arrive_and_wait(int my_rc) {
    std::unique_lock<std::mutex> lk(mu);
    if (my_rc != 0 && rc == 0) rc = my_rc;
    const unsigned long long g = gen;
    if (++arrived == n) {
      result = rc;
      rc = 0;
      arrived = 0;
      ++gen;
      cv.notify_all();
      return result;
    }
    cv.wait(lk, [&] { return gen != g; });
    return result;
  }
};

struct sn_mgpu {
  int ndev = 0, max_batch = 0, per_dev = 0, W = 0, H = 0;
  int gather = 1;                    // 1 = hipMemcpyPeerAsync over xGMI, 2 = RCCL grouped send/recv
  std::vector<Worker*> w;
  Rccl rccl;
  StatusBarrier agree;
  sn_mgpu_ring ring{};
  std::mutex api_mu;                 // submit / wait bookkeeping
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

// ---- ticket ring of the asynchronous form (pure bookkeeping; needs no GPU, tests/test_mgpu.py) -------------------
// Tickets count 1, 2, 3, ..; ticket t uses buffer slot t % SN_MGPU_SLOTS; a slot is busy from submit until the wait of
// its ticket, so at most SN_MGPU_SLOTS tickets are in flight and they may be waited for in any order.
int sn_mgpu_ring_init(sn_mgpu_ring* r) {
  if (!r) return SN_ERR_ARG;
  r->next = 1;
  for (int i = 0; i < SN_MGPU_SLOTS; ++i) r->slot_ticket[i] = 0;
  return SN_OK;
}
int sn_mgpu_ring_submit(sn_mgpu_ring* r, uint64_t* ticket, int* slot) {
  if (!r || !ticket || !slot || r->next == 0) return SN_ERR_ARG;
  const int s = (int)(r->next % SN_MGPU_SLOTS);
  if (r->slot_ticket[s] != 0) return SN_ERR_BUSY;          // its previous batch has not been waited for
  r->slot_ticket[s] = r->next;
  *ticket = r->next++;
  *slot = s;
  return SN_OK;
}
int sn_mgpu_ring_wait(sn_mgpu_ring* r, uint64_t ticket, int* slot) {
  if (!r || !slot || ticket == 0) return SN_ERR_ARG;
  const int s = (int)(ticket % SN_MGPU_SLOTS);
  if (r->slot_ticket[s] != ticket) return SN_ERR_TICKET;   // unknown, or consumed already
  r->slot_ticket[s] = 0;
  *slot = s;
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
    if (w->st) hipStreamSynchronize(w->st);
    if (w->cs) hipStreamSynchronize(w->cs);
    if (w->comm && m->rccl.CommDestroy) m->rccl.CommDestroy(w->comm);
    for (int i = 0; i < kSlots; ++i) {
      if (w->raw[i]) hipFree(w->raw[i]);
      if (w->disp[i]) hipFree(w->disp[i]);
      if (w->ev_compute[i]) hipEventDestroy(w->ev_compute[i]);
      if (w->ev_done[i]) hipEventDestroy(w->ev_done[i]);
    }
    if (w->st) hipStreamDestroy(w->st);
    if (w->cs) hipStreamDestroy(w->cs);
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
  // SN_MGPU_ALLOW_DUP=1 (tests): several shards may name the same device, so the worker threads, the shard
  // arithmetic and the gather run for ndev > 1 on a one-GPU box.  RCCL needs distinct devices: peer copies then.
  const bool allow_dup = getenv("SN_MGPU_ALLOW_DUP") != nullptr && atoi(getenv("SN_MGPU_ALLOW_DUP")) == 1;
  bool distinct = true;
  for (int k = 0; k < ndev; ++k) {
    const int d = devices ? devices[k] : k;
    if (d < 0 || d >= have) return SN_ERR_ARG;
    for (int j = 0; j < k; ++j)
      if ((devices ? devices[j] : j) == d) {
        if (!allow_dup) return SN_ERR_ARG;                           // a device may hold one shard only
        distinct = false;
      }
  }
  sn_mgpu* m = new sn_mgpu();
  m->ndev = ndev;
  m->agree.n = ndev;
  sn_mgpu_ring_init(&m->ring);
  sn_config c{};
  if (cfg) c = *cfg;
  m->max_batch = c.max_batch > 0 ? c.max_batch : ndev;
  m->per_dev = (m->max_batch + ndev - 1) / ndev;
  // the north-star's exchange is RCCL over xGMI: the default whenever it can run; SN_MGPU_GATHER=peer / rccl force one
  m->gather = (ndev > 1 && distinct) ? 2 : 1;
  if (const char* e = getenv("SN_MGPU_GATHER")) {
    if (!strcmp(e, "peer")) m->gather = 1;
    else if (!strcmp(e, "rccl") && distinct) m->gather = 2;      // (also with ndev = 1: loads RCCL, builds the communicator, runs an empty group)
  }
  if (m->gather == 2 && !m->rccl.load()) m->gather = 1;
  int rc = SN_OK;
  for (int k = 0; k < ndev && rc == SN_OK; ++k) {
    Worker* w = new Worker();
    m->w.push_back(w);
    w->dev = devices ? devices[k] : k;
    sn_config ck = c;
    ck.device = w->dev;
    ck.max_batch = m->per_dev;
    // more than one device: a gather runs beside the engines on this library's own exchange streams — the engines'
    // pipeline streams go to their own (high-priority) hardware queues so that it cannot serialise with the towers
    // (-30 % for the root otherwise); the caller does not have to know about SN_STREAM_PRIORITY
    rc = sn_create_prio(model_file, &ck, ndev > 1 ? 1 : -1, &w->h);
    if (rc != SN_OK) {
      const char* d = sn_last_error(nullptr);
      m->err = "sn_create on device " + std::to_string(w->dev) + ": " + sn_strerror(rc) + (d && *d ? std::string(": ") + d : std::string());
      break;
    }
    sn_io_info info;
    sn_get_io_info(w->h, &info);
    m->W = info.width;
    m->H = info.height;
    if (hipSetDevice(w->dev) != hipSuccess || hipStreamCreateWithFlags(&w->st, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&w->cs, hipStreamNonBlocking) != hipSuccess)
      rc = SN_ERR_DEVICE;
    for (int i = 0; i < kSlots && rc == SN_OK; ++i)
      if (hipEventCreateWithFlags(&w->ev_compute[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&w->ev_done[i], hipEventDisableTiming) != hipSuccess)
        rc = SN_ERR_DEVICE;
    if (rc == SN_OK && k > 0) {          // staging for the maps that travel to the root in device mode, one set per slot
      const size_t bytes = (size_t)m->per_dev * m->W * m->H * 4;
      for (int i = 0; i < kSlots && rc == SN_OK; ++i)
        if (hipMalloc(reinterpret_cast<void**>(&w->raw[i]), bytes) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&w->disp[i]), bytes) != hipSuccess)
          rc = SN_ERR_NOMEM;
      // direct xGMI copies in both directions between this device and the root (an "already enabled" error is fine)
      if (w->dev != m->w[0]->dev) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, w->dev, m->w[0]->dev) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(m->w[0]->dev, 0);
        hipSetDevice(m->w[0]->dev);
        if (hipDeviceCanAccessPeer(&can, m->w[0]->dev, w->dev) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(w->dev, 0);
      }
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
  // the host form runs on the engines' own streams and shares their workspaces with device-resident batches that are
  // still in flight on the compute streams: refuse to overlap them (wait for the tickets first)
  std::lock_guard<std::mutex> api(m->api_mu);
  for (int i = 0; i < SN_MGPU_SLOTS; ++i)
    if (m->ring.slot_ticket[i] != 0) {
      m->err = "sn_mgpu_infer_batch: device-resident batches are in flight (sn_mgpu_wait for their tickets first)";
      return SN_ERR_BUSY;
    }
  return run_all(m, [&](int k) -> int {
    int first = 0, cnt = 0;
    sn_mgpu_shard(n, m->ndev, k, &first, &cnt);
    if (cnt == 0) return SN_OK;
    return sn_infer_batch(m->w[k]->h, cnt, in + (size_t)first * 6 * HW, out_i32 ? out_i32 + (size_t)first * HW : nullptr,
                          out_disp ? out_disp + (size_t)first * HW : nullptr, SN_MEM_HOST, nullptr);
  });
}

// Device buffers, asynchronous: in_per_device[k] holds shard k's pairs in the memory of device k; the int32 / float maps
// of all n pairs are gathered, in batch order, into out_* in the memory of device 0 (the root) over xGMI.  Returns as soon
// as every device has its work enqueued; the inputs and the root buffers belong to the call until sn_mgpu_wait(ticket).
int sn_mgpu_submit_device(sn_mgpu* m, int n, const int8_t* const* in_per_device, int32_t* out_i32_root, float* out_disp_root,
                          uint64_t* ticket) {
  if (!m) return SN_ERR_ARG;
  if (!in_per_device || !ticket || (!out_i32_root && !out_disp_root) || n <= 0 || n > m->max_batch) {
    m->err = "sn_mgpu_submit_device: bad arguments";
    return SN_ERR_ARG;
  }
  std::lock_guard<std::mutex> api(m->api_mu);
  int slot = 0;
  uint64_t t = 0;
  int rc = sn_mgpu_ring_submit(&m->ring, &t, &slot);
  if (rc != SN_OK) {
    m->err = "sn_mgpu_submit_device: two batches are in flight already (wait for a ticket first)";
    return rc;
  }
  const size_t HW = (size_t)m->W * m->H;
  const int root = m->w[0]->dev;
  rc = run_all(m, [&](int k) -> int {
    Worker* w = m->w[k];
    int first = 0, cnt = 0;
    sn_mgpu_shard(n, m->ndev, k, &first, &cnt);
    int32_t* raw = !out_i32_root ? nullptr : (k == 0 ? out_i32_root : w->raw[slot]);
    float* disp = !out_disp_root ? nullptr : (k == 0 ? out_disp_root : w->disp[slot]);
    int my = SN_OK;
    if (cnt > 0) {
      if (!in_per_device[k]) my = SN_ERR_ARG;
      else my = sn_infer_batch(w->h, cnt, in_per_device[k], raw, disp, SN_MEM_DEVICE, w->cs);      // enqueued on the compute stream
    }
    if (my == SN_OK && hipEventRecord(w->ev_compute[slot], w->cs) != hipSuccess) my = SN_ERR_DEVICE;
    // every rank learns whether ALL shards are on their way before any of them enters the exchange
    const int all = m->ndev > 1 ? m->agree.arrive_and_wait(my) : my;
    if (all != SN_OK) {
      (void)hipEventRecord(w->ev_done[slot], w->st);
      return my != SN_OK ? my : all;
    }
    // (a failure here is reported through the second agreement below when RCCL carries the gather: this rank still
    // enters the group so that nobody is left with an unmatched half)
    const bool waited = hipStreamWaitEvent(w->st, w->ev_compute[slot], 0) == hipSuccess;
    if (!waited && m->gather != 2) {
      (void)hipEventRecord(w->ev_done[slot], w->st);
      return SN_ERR_DEVICE;
    }
    if (m->gather == 2) {     // one grouped exchange: root posts a recv per peer, every peer one send per map kind
      const bool started = waited && m->rccl.GroupStart() == 0;
      bool ok = started;
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
      if (started) ok = (m->rccl.GroupEnd() == 0) && ok;        // never an unpaired GroupEnd
      // second agreement, AFTER the exchange is enqueued: a rank that failed inside the group leaves its peers with an
      // unmatched send / recv on their exchange streams, which a later hipStreamSynchronize would wait on forever.  If any
      // rank failed, every rank aborts its communicator (that releases the enqueued half) and peer copies carry the
      // gather from then on.
      const int mine2 = ok ? SN_OK : SN_ERR_DEVICE;
      const int all2 = m->ndev > 1 ? m->agree.arrive_and_wait(mine2) : mine2;
      if (all2 != SN_OK) {
        if (w->comm) {
          m->rccl.CommAbort(w->comm);        // (Rccl::load refuses a library without it)
          w->comm = nullptr;
        }
        if (k == 0) m->gather = 1;
        (void)hipEventRecord(w->ev_done[slot], w->st);        // the slot's event is always recorded
        return mine2 != SN_OK ? mine2 : all2;
      }
    } else if (k > 0 && cnt > 0) {       // peer copies: every non-root device pushes its maps over its own link
      bool ok = true;
      if (raw && hipMemcpyPeerAsync(out_i32_root + (size_t)first * HW, root, raw, w->dev, (size_t)cnt * HW * 4, w->st) != hipSuccess)
        ok = false;
      if (ok && disp && hipMemcpyPeerAsync(out_disp_root + (size_t)first * HW, root, disp, w->dev, (size_t)cnt * HW * 4, w->st) != hipSuccess)
        ok = false;
      if (!ok) {
        (void)hipEventRecord(w->ev_done[slot], w->st);
        return SN_ERR_DEVICE;
      }
    }
    return hipEventRecord(w->ev_done[slot], w->st) == hipSuccess ? SN_OK : SN_ERR_DEVICE;
  });
  if (rc != SN_OK) {                     // nothing usable is in flight: drain what was enqueued and free the slot
    run_all(m, [&](int k) -> int {
      hipStreamSynchronize(m->w[k]->cs);
      hipStreamSynchronize(m->w[k]->st);
      return SN_OK;
    });
    int s2 = 0;
    sn_mgpu_ring_wait(&m->ring, t, &s2);
    return rc;
  }
  *ticket = t;
  return SN_OK;
}

int sn_mgpu_wait(sn_mgpu* m, uint64_t ticket) {
  if (!m) return SN_ERR_ARG;
  // Blocks on the events WITHOUT the API lock (a submit from another thread must not stall for a whole batch); the slot
  // is released only afterwards, so a concurrent submit cannot reuse its staging buffers while the gather still runs.
  // The events are waited for from this thread: the worker threads stay free for that submit.
  int slot = 0;
  {
    std::lock_guard<std::mutex> api(m->api_mu);
    slot = (int)(ticket % SN_MGPU_SLOTS);
    if (ticket == 0 || m->ring.slot_ticket[slot] != ticket) {
      m->err = "sn_mgpu_wait: unknown or already-consumed ticket";
      return SN_ERR_TICKET;
    }
  }
  int rc = SN_OK;
  for (int k = 0; k < m->ndev; ++k)
    if (hipEventSynchronize(m->w[k]->ev_done[slot]) != hipSuccess && rc == SN_OK) rc = SN_ERR_DEVICE;
  std::lock_guard<std::mutex> api(m->api_mu);
  int s2 = 0;
  const int r2 = sn_mgpu_ring_wait(&m->ring, ticket, &s2);       // fails if another thread consumed the same ticket meanwhile
  if (r2 != SN_OK) {
    m->err = "sn_mgpu_wait: unknown or already-consumed ticket";
    return r2;
  }
  return rc;
}

int sn_mgpu_infer_batch_device(sn_mgpu* m, int n, const int8_t* const* in_per_device, int32_t* out_i32_root,
                               float* out_disp_root) {
  uint64_t t = 0;
  const int rc = sn_mgpu_submit_device(m, n, in_per_device, out_i32_root, out_disp_root, &t);
  if (rc != SN_OK) return rc;
  return sn_mgpu_wait(m, t);
}

}  // extern "C"
