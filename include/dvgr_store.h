/* dvgr_store.h -- C ABI of the feature store in front of the model.
 *
 * Replaces, in the reference, the per-item HDF5 reads of `VideoQADataset.__getitem__` (DataLoader.py:61-84: two
 * `h5py.File` opens per sample, `f['resnet_features'][i]`, `f['resnext_features'][i]`) and the id -> row dictionaries
 * `VideoQADataLoader.__init__` builds from the files' `ids` datasets (DataLoader.py:137-144).  The `.h5` files written by
 * preprocess/preprocess_features.py:143-203 are converted once (tools/h5_to_flat.py) into one flat little-endian file per
 * feature kind:
 *
 *   offset 0    char[8]  "DVGRFEAT"
 *          8    u32      version (1)
 *         12    u32      element type (0 = float32)
 *         16    u32      ndim of one item (1..4)
 *         20    u32      reserved (0)
 *         24    u64[4]   shape of one item, unused trailing dims = 1   (appearance: N,16,2048; motion: N,2048)
 *         56    u64      item count V
 *         64    u64      byte offset of the ids (int64[V], the `ids` dataset of the .h5)
 *         72    u64      byte offset of the data (multiple of 4096), V rows of prod(shape) elements, row-major
 *
 * which is mapped read-only; a batch is assembled by copying its rows into a pinned staging buffer with a few host threads
 * and ONE host-to-device copy per feature kind (the per-item opens and per-item H2D copies of the reference's default
 * collate + `.to(device)` are gone).  0 / DVGR_E_* / -errno style negative / hipError_t return codes.
 * Threading: open / lookup / gather_host are safe from any number of threads on one store; gather_device and a
 * prefetcher's next() keep per-object staging state and belong to ONE consumer thread each (several prefetchers may
 * share a store).
 */
#ifndef DVGR_STORE_H
#define DVGR_STORE_H
#include "dvgr.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dvgr_store dvgr_store;

#define DVGR_E_IO (-20)      /* open / stat / mmap failed                    */
#define DVGR_E_FORMAT (-21)  /* not a DVGRFEAT file, or a truncated one      */
#define DVGR_E_MISSING (-22) /* an id is not in the store                    */

typedef struct dvgr_store_info {
  int64_t items;       /* V                                  */
  int64_t row_elems;   /* prod(shape)                        */
  int32_t ndim;
  int32_t elem_type;   /* 0 = float32                        */
  int64_t shape[4];
} dvgr_store_info;

int dvgr_store_open(const char *path, dvgr_store **out);
void dvgr_store_close(dvgr_store *s);
int dvgr_store_get_info(const dvgr_store *s, dvgr_store_info *info);
/* the `ids` array of the file (V entries), valid until close */
const int64_t *dvgr_store_ids(const dvgr_store *s);
/* rows[i] = row of ids[i] (what `app_feat_id_to_index[str(video_idx)]` is in DataLoader.py:70); DVGR_E_MISSING if absent */
int dvgr_store_lookup(const dvgr_store *s, const int64_t *ids, int64_t n, int64_t *rows);
/* dst_host[i] = row rows[i]; `threads` <= 0: pick.  No GPU involved (this is what the CPU tests call). */
int dvgr_store_gather_host(const dvgr_store *s, const int64_t *rows, int64_t n, void *dst_host, int threads);
/* The same rows into device memory: staged through one of the store's two pinned buffers (allocated on first use, grown
 * on demand), then one hipMemcpyAsync on `stream`.  Returns once the copy is enqueued; the next call using the same
 * staging slot waits for it. */
int dvgr_store_gather_device(dvgr_store *s, const int64_t *rows, int64_t n, void *dst_device, int threads, void *stream);

/* ---- epoch prefetcher: a native worker thread stages batch after batch into a ring of pinned slots (no interpreter
 * lock involved), the consumer turns each staged slot into one host-to-device copy per store.  Replaces the worker
 * processes + default collate + `.to(device)` of `torch.utils.data.DataLoader` as VideoQADataLoader uses it
 * (DataLoader.py:165, train.py:128-137). */
typedef struct dvgr_prefetch dvgr_prefetch;

/* stores[i] / rows[i]: up to 4 stores and, for each, the row of every sample of the epoch IN EPOCH ORDER (`total` entries,
 * copied); batches are consecutive runs of `batch` samples (the last one may be short).  `depth` slots (2..8) of
 * batch * row_bytes pinned bytes per store are allocated here.  The calling thread's current device is the target. */
int dvgr_prefetch_start(dvgr_store *const *stores, int n_stores, const int64_t *const *rows, int64_t total, int64_t batch,
                        int depth, int threads, dvgr_prefetch **out);
/* Next batch: waits until it is staged, enqueues dst_device[i] <- slot on `stream` for every store, sets *n_out to its
 * sample count.  Returns 0, or 1 with *n_out = 0 when the epoch is exhausted. */
int dvgr_prefetch_next(dvgr_prefetch *p, void *const *dst_device, int64_t *n_out, void *stream);
void dvgr_prefetch_stop(dvgr_prefetch *p);

#ifdef __cplusplus
}
#endif
#endif
