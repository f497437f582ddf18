// Host side of the streaming input path: the minibatch's decoded frames gathered from memory-mapped shards straight into a pinned
// staging buffer by a few threads.  Replaces, for decoded-frame shards, what the reference's DataLoader workers do per sample
// (dataloaders/q_learning_real.py:55-73: open, decode, resize, stack; torch's collate then copies every sample once more into the
// batch, and the pinned-memory thread a third time): here every frame is copied ONCE, from the page cache to the buffer the
// host-to-device copy reads.  No device code in this file.
#include <string.h>

#include <thread>
#include <vector>

#include "common.h"

// dst + i * bytes_each <- src[i][0 .. bytes_each) for i < n, on `threads` host threads (records are dealt in contiguous runs so
// that a thread writes one contiguous piece of dst).  The call returns when every record is in place.
extern "C" int vdqn_host_gather(void* dst, const void* const* src, int64_t n, int64_t bytes_each, int32_t threads) {
  VDQN_CHECK(dst && src && n >= 0 && bytes_each > 0, "vdqn_host_gather: bad arguments");
  if (n == 0) return VDQN_OK;
  int t = threads < 1 ? 1 : threads;
  if ((int64_t)t > n) t = (int)n;
  auto run = [=](int64_t lo, int64_t hi) {
    unsigned char* d = static_cast<unsigned char*>(dst) + lo * bytes_each;
    for (int64_t i = lo; i < hi; ++i, d += bytes_each) memcpy(d, src[i], (size_t)bytes_each);
  };
  if (t == 1) {
    run(0, n);
    return VDQN_OK;
  }
  std::vector<std::thread> pool;
  pool.reserve((size_t)t - 1);
  const int64_t per = (n + t - 1) / t;
  // A thread that cannot be created (EAGAIN under a pid limit, 8 ranks x 16 threads) must not unwind through this extern "C"
  // function with joinable threads alive (std::terminate): the records from the failed one on are copied by this thread instead.
  int64_t inline_from = n;
  for (int k = 1; k < t; ++k) {
    const int64_t lo = k * per, hi = lo + per < n ? lo + per : n;
    if (lo >= hi) break;
    try {
      pool.emplace_back(run, lo, hi);
    } catch (...) {
      inline_from = lo;
      break;
    }
  }
  run(0, per < n ? per : n);
  if (inline_from < n) run(inline_from, n);
  for (auto& th : pool) th.join();
  return VDQN_OK;
}
