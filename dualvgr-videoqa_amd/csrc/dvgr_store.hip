// dvgr_store.hip -- feature store: mapped flat file, id index, threaded row gather, pinned staging + one H2D copy per
// batch (include/dvgr_store.h).  Host code only; built by hipcc for the HIP runtime calls.
#include "../../include/dvgr_store.h"

#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {
struct FileHeader {
  char magic[8];
  uint32_t version, elem_type, ndim, reserved;
  uint64_t shape[4];
  uint64_t items, ids_off, data_off;
};
static_assert(sizeof(FileHeader) == 80, "header layout");
}  // namespace

struct dvgr_store {
  int fd = -1;
  void* map = nullptr;
  size_t map_bytes = 0;
  FileHeader h{};
  const int64_t* ids = nullptr;
  const char* data = nullptr;
  size_t row_bytes = 0;
  std::unordered_map<int64_t, int64_t> index;
  // device path: two pinned staging slots used alternately
  void* stage[2] = {nullptr, nullptr};
  size_t stage_cap[2] = {0, 0};
  hipEvent_t done[2] = {nullptr, nullptr};
  bool pending[2] = {false, false};
  int next = 0;
};

namespace {
void copy_rows(const dvgr_store* s, const int64_t* rows, int64_t n, char* dst, int threads) {
  const size_t rb = s->row_bytes;
  if (threads <= 0) {
    const size_t total = rb * (size_t)n;
    threads = total < (4u << 20) ? 1 : 8;
  }
  if (threads > n) threads = (int)(n > 0 ? n : 1);
  auto work = [&](int t) {
    // contiguous share of the bytes of the batch: a row may be split between two threads
    const size_t total = rb * (size_t)n;
    const size_t lo = total * (size_t)t / (size_t)threads, hi = total * (size_t)(t + 1) / (size_t)threads;
    size_t pos = lo;
    while (pos < hi) {
      const size_t r = pos / rb, off = pos - r * rb;
      size_t len = rb - off;
      if (pos + len > hi) len = hi - pos;
      std::memcpy(dst + pos, s->data + (size_t)rows[r] * rb + off, len);
      pos += len;
    }
  };
  if (threads == 1) { work(0); return; }
  std::vector<std::thread> pool;
  pool.reserve(threads - 1);
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
}
int check_rows(const dvgr_store* s, const int64_t* rows, int64_t n) {
  for (int64_t i = 0; i < n; ++i)
    if (rows[i] < 0 || (uint64_t)rows[i] >= s->h.items) return DVGR_E_DIMS;
  return 0;
}
}  // namespace

struct dvgr_prefetch {
  enum State { FREE = 0, STAGED = 1, COPYING = 2 };
  int n_stores = 0, depth = 0, threads = 0, device = 0;
  const dvgr_store* stores[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<int64_t> rows[4];
  int64_t total = 0, batch = 0, n_batches = 0;
  std::vector<char*> slot[4];          // [store][slot] pinned
  std::vector<hipEvent_t> done;        // per slot
  std::vector<int> state;              // per slot
  std::vector<int64_t> count;          // samples staged in the slot
  int64_t consumed = 0;                // batches handed to the consumer
  int error = 0;
  bool stop = false;
  std::mutex mu;
  std::condition_variable cv;
  std::thread worker;

  void run() {
    (void)hipSetDevice(device);
    for (int64_t b = 0; b < n_batches; ++b) {
      const int sl = (int)(b % depth);
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || state[sl] != STAGED; });
        if (stop) return;
      }
      if (state[sl] == COPYING) {      // the copy that reads this slot must have finished
        const hipError_t e = hipEventSynchronize(done[sl]);
        if (e != hipSuccess) { std::lock_guard<std::mutex> lk(mu); error = (int)e; cv.notify_all(); return; }
      }
      const int64_t lo = b * batch, n = (lo + batch <= total ? batch : total - lo);
      for (int i = 0; i < n_stores; ++i) copy_rows(stores[i], rows[i].data() + lo, n, slot[i][sl], threads);
      {
        std::lock_guard<std::mutex> lk(mu);
        count[sl] = n;
        state[sl] = STAGED;
      }
      cv.notify_all();
    }
  }
};

extern "C" {

int dvgr_prefetch_start(dvgr_store* const* stores, int n_stores, const int64_t* const* rows, int64_t total, int64_t batch, int depth,
                        int threads, dvgr_prefetch** out) {
  if (!stores || !rows || !out) return DVGR_E_NULL;
  *out = nullptr;
  if (n_stores < 1 || n_stores > 4 || total < 0 || batch < 1 || depth < 2 || depth > 8) return DVGR_E_DIMS;
  for (int i = 0; i < n_stores; ++i) {
    if (!stores[i] || (total > 0 && !rows[i])) return DVGR_E_NULL;
    const int rc = check_rows(stores[i], rows[i], total);
    if (rc) return rc;
  }
  dvgr_prefetch* p = new dvgr_prefetch();
  p->n_stores = n_stores; p->depth = depth; p->threads = threads; p->total = total; p->batch = batch;
  p->n_batches = (total + batch - 1) / batch;
  if (hipGetDevice(&p->device) != hipSuccess) { delete p; return (int)hipGetLastError(); }
  p->state.assign(depth, dvgr_prefetch::FREE);
  p->count.assign(depth, 0);
  p->done.assign(depth, nullptr);
  for (int i = 0; i < n_stores; ++i) {
    p->stores[i] = stores[i];
    p->rows[i].assign(rows[i], rows[i] + total);
    p->slot[i].assign(depth, nullptr);
  }
  hipError_t e = hipSuccess;
  for (int sl = 0; sl < depth && e == hipSuccess; ++sl) {
    e = hipEventCreateWithFlags(&p->done[sl], hipEventDisableTiming);
    for (int i = 0; i < n_stores && e == hipSuccess; ++i)
      e = hipHostMalloc((void**)&p->slot[i][sl], stores[i]->row_bytes * (size_t)batch, hipHostMallocDefault);
  }
  if (e != hipSuccess) { dvgr_prefetch_stop(p); return (int)e; }
  p->worker = std::thread([p] { p->run(); });
  *out = p;
  return 0;
}

int dvgr_prefetch_next(dvgr_prefetch* p, void* const* dst_device, int64_t* n_out, void* stream) {
  if (!p || !dst_device || !n_out) return DVGR_E_NULL;
  *n_out = 0;
  if (p->consumed >= p->n_batches) return 1;
  const int sl = (int)(p->consumed % p->depth);
  {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv.wait(lk, [&] { return p->error || p->state[sl] == dvgr_prefetch::STAGED; });
    if (p->error) return p->error;
  }
  const int64_t n = p->count[sl];
  for (int i = 0; i < p->n_stores; ++i) {
    if (!dst_device[i]) return DVGR_E_NULL;
    const hipError_t e = hipMemcpyAsync(dst_device[i], p->slot[i][sl], p->stores[i]->row_bytes * (size_t)n, hipMemcpyHostToDevice,
                                        (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  const hipError_t e = hipEventRecord(p->done[sl], (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->state[sl] = dvgr_prefetch::COPYING;
  }
  p->cv.notify_all();
  p->consumed += 1;
  *n_out = n;
  return 0;
}

void dvgr_prefetch_stop(dvgr_prefetch* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->stop = true;
  }
  p->cv.notify_all();
  if (p->worker.joinable()) p->worker.join();
  for (int sl = 0; sl < (int)p->done.size(); ++sl) {
    if (p->done[sl]) {
      if (p->state[sl] == dvgr_prefetch::COPYING) (void)hipEventSynchronize(p->done[sl]);
      (void)hipEventDestroy(p->done[sl]);
    }
    for (int i = 0; i < p->n_stores; ++i)
      if (sl < (int)p->slot[i].size() && p->slot[i][sl]) (void)hipHostFree(p->slot[i][sl]);
  }
  delete p;
}

int dvgr_store_open(const char* path, dvgr_store** out) {
  if (!path || !out) return DVGR_E_NULL;
  *out = nullptr;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return DVGR_E_IO;
  struct stat stt;
  if (fstat(fd, &stt) != 0) { close(fd); return DVGR_E_IO; }
  if ((size_t)stt.st_size < sizeof(FileHeader)) { close(fd); return DVGR_E_FORMAT; }
  void* map = mmap(nullptr, (size_t)stt.st_size, PROT_READ, MAP_SHARED, fd, 0);
  if (map == MAP_FAILED) { close(fd); return DVGR_E_IO; }
  dvgr_store* s = new dvgr_store();
  s->fd = fd; s->map = map; s->map_bytes = (size_t)stt.st_size;
  std::memcpy(&s->h, map, sizeof(FileHeader));
  const FileHeader& h = s->h;
  bool ok = std::memcmp(h.magic, "DVGRFEAT", 8) == 0 && h.version == 1 && h.elem_type == 0 && h.ndim >= 1 && h.ndim <= 4;
  size_t elems = 1;
  if (ok) for (int i = 0; i < 4; ++i) { if (h.shape[i] == 0 || (i >= (int)h.ndim && h.shape[i] != 1)) ok = false; else elems *= h.shape[i]; }
  s->row_bytes = elems * 4;
  ok = ok && (h.ids_off % 8 == 0) && (h.data_off % 4096 == 0) && h.ids_off >= sizeof(FileHeader) &&
       h.ids_off + h.items * 8 <= h.data_off && h.data_off + h.items * s->row_bytes <= s->map_bytes;
  if (!ok) { dvgr_store_close(s); return DVGR_E_FORMAT; }
  s->ids = reinterpret_cast<const int64_t*>((const char*)map + h.ids_off);
  s->data = (const char*)map + h.data_off;
  s->index.reserve(h.items * 2);
  for (uint64_t i = 0; i < h.items; ++i) s->index[s->ids[i]] = (int64_t)i;     // later duplicates win, as the dict comprehension does
  *out = s;
  return 0;
}

void dvgr_store_close(dvgr_store* s) {
  if (!s) return;
  for (int i = 0; i < 2; ++i) {
    if (s->pending[i]) (void)hipEventSynchronize(s->done[i]);
    if (s->done[i]) (void)hipEventDestroy(s->done[i]);
    if (s->stage[i]) (void)hipHostFree(s->stage[i]);
  }
  if (s->map) munmap(s->map, s->map_bytes);
  if (s->fd >= 0) close(s->fd);
  delete s;
}

int dvgr_store_get_info(const dvgr_store* s, dvgr_store_info* info) {
  if (!s || !info) return DVGR_E_NULL;
  info->items = (int64_t)s->h.items;
  info->row_elems = (int64_t)(s->row_bytes / 4);
  info->ndim = (int32_t)s->h.ndim;
  info->elem_type = (int32_t)s->h.elem_type;
  for (int i = 0; i < 4; ++i) info->shape[i] = (int64_t)s->h.shape[i];
  return 0;
}

const int64_t* dvgr_store_ids(const dvgr_store* s) { return s ? s->ids : nullptr; }

int dvgr_store_lookup(const dvgr_store* s, const int64_t* ids, int64_t n, int64_t* rows) {
  if (!s || (n > 0 && (!ids || !rows))) return DVGR_E_NULL;
  int rc = 0;
  for (int64_t i = 0; i < n; ++i) {
    const auto it = s->index.find(ids[i]);
    if (it == s->index.end()) { rows[i] = -1; rc = DVGR_E_MISSING; } else rows[i] = it->second;
  }
  return rc;
}

int dvgr_store_gather_host(const dvgr_store* s, const int64_t* rows, int64_t n, void* dst_host, int threads) {
  if (!s || (n > 0 && (!rows || !dst_host))) return DVGR_E_NULL;
  if (n <= 0) return 0;
  const int rc = check_rows(s, rows, n);
  if (rc) return rc;
  copy_rows(s, rows, n, (char*)dst_host, threads);
  return 0;
}

int dvgr_store_gather_device(dvgr_store* s, const int64_t* rows, int64_t n, void* dst_device, int threads, void* stream) {
  if (!s || (n > 0 && (!rows || !dst_device))) return DVGR_E_NULL;
  if (n <= 0) return 0;
  const int rc = check_rows(s, rows, n);
  if (rc) return rc;
  const size_t bytes = s->row_bytes * (size_t)n;
  const int slot = s->next;
  s->next ^= 1;
  if (s->pending[slot]) {                         // the copy that last read this slot must have left it
    const hipError_t e = hipEventSynchronize(s->done[slot]);
    if (e != hipSuccess) return (int)e;
    s->pending[slot] = false;
  }
  if (s->stage_cap[slot] < bytes) {
    if (s->stage[slot]) (void)hipHostFree(s->stage[slot]);
    s->stage[slot] = nullptr; s->stage_cap[slot] = 0;
    const hipError_t e = hipHostMalloc(&s->stage[slot], bytes, hipHostMallocDefault);
    if (e != hipSuccess) return (int)e;
    s->stage_cap[slot] = bytes;
  }
  if (!s->done[slot]) {
    const hipError_t e = hipEventCreateWithFlags(&s->done[slot], hipEventDisableTiming);
    if (e != hipSuccess) return (int)e;
  }
  copy_rows(s, rows, n, (char*)s->stage[slot], threads);
  hipError_t e = hipMemcpyAsync(dst_device, s->stage[slot], bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  e = hipEventRecord(s->done[slot], (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  s->pending[slot] = true;
  return 0;
}

}  // extern "C"
