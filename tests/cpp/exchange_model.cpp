// exchange_model.cpp -- CPU model of the device-side pose gather (hector_slam_amd/csrc/pose_exchange.h), test infrastructure.
//
// The SAME layout, tags and flow-control rule as the HIP kernels (the shared header), with the mailboxes in POSIX shared
// memory and the GPU's system-scope 8-byte stores / loads as relaxed 64-bit atomics: one process per rank, started by
// tests/test_exchange_protocol.py under a world-size-2 (or 3) gloo group that carries the shared-memory names the way the
// GPU path carries its IPC handles.  Every rank runs, per epoch e: "matcher" (a random delay, skewed per rank), POST e to
// every rank's mailbox (granule by granule, with random pauses so that posts of different ranks interleave), WAIT e - lag
// and check every value.  What it proves that a 1-GPU box cannot: that with depth >= 2 + 2 lag no rank ever finds rows of
// a LATER epoch in a buffer it has not unpacked yet, however the ranks drift -- and (negative control, --force) that the
// check does see the overwrite when the mailbox is one buffer too shallow.
//
// usage: exchange_model RANK WORLD TOTAL_ROWS COLS DEPTH LAG EPOCHS SEED FORCE NAME_0 .. NAME_{WORLD-1}
// prints one JSON line; exit code 0 = no violation, no timeout, every value as expected.
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "pose_exchange.h"

using hsm::ExchangeLayout;
using Clock = std::chrono::steady_clock;

static void shard_bounds(int total, int rank, int world, int* b, int* e) {  // hsm_shard_bounds
  const int base = total / world, rem = total % world;
  *b = rank * base + (rank < rem ? rank : rem);
  *e = *b + base + (rank < rem ? 1 : 0);
}

// the value rank r posts for (epoch, row, col): any function all ranks can recompute
static uint32_t value_of(uint64_t epoch, int row, int col) {
  uint64_t x = epoch * 0x9E3779B97F4A7C15ull ^ ((uint64_t)row << 20) ^ (uint64_t)col;
  x ^= x >> 29;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 32;
  return (uint32_t)x;
}

int main(int argc, char** argv) {
  if (argc < 10) {
    fprintf(stderr, "usage: %s RANK WORLD TOTAL_ROWS COLS DEPTH LAG EPOCHS SEED FORCE NAME...\n", argv[0]);
    return 2;
  }
  const int rank = atoi(argv[1]), world = atoi(argv[2]);
  ExchangeLayout lay{world, atoi(argv[3]), atoi(argv[4]), atoi(argv[5])};
  const int lag = atoi(argv[6]);
  const uint64_t epochs = (uint64_t)atoll(argv[7]);
  const unsigned seed = (unsigned)atoi(argv[8]);
  const bool force = atoi(argv[9]) != 0;
  if (argc != 10 + world || world > hsm::kExchangeMaxWorld) return 2;
  if (!force && hsm::exchange_min_depth(lag) > lay.depth) {
    fprintf(stderr, "depth %d < 2 + 2 * lag (%d): the runtime refuses this (pass FORCE=1 for the negative control)\n", lay.depth, lag);
    return 2;
  }
  // map every rank's mailbox; the own one is created (zero filled) here, the others are opened once they exist
  std::vector<std::atomic<uint64_t>*> box((size_t)world, nullptr);
  for (int pass = 0; pass < 2; ++pass)
    for (int r = 0; r < world; ++r) {
      if ((pass == 0) != (r == rank)) continue;
      int fd = -1;
      for (int tries = 0; tries < 2000 && fd < 0; ++tries) {
        fd = shm_open(argv[10 + r], r == rank ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
        if (fd < 0) usleep(5000);
      }
      if (fd < 0) {
        perror("shm_open");
        return 2;
      }
      if (r == rank && ftruncate(fd, (off_t)lay.bytes()) != 0) {
        perror("ftruncate");
        return 2;
      }
      // a peer's object may exist before it has been sized: wait until it has its full size
      for (int tries = 0; tries < 2000; ++tries) {
        const off_t sz = lseek(fd, 0, SEEK_END);
        if ((size_t)sz >= lay.bytes()) break;
        usleep(5000);
      }
      void* p = mmap(nullptr, lay.bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (p == MAP_FAILED) {
        perror("mmap");
        return 2;
      }
      box[(size_t)r] = reinterpret_cast<std::atomic<uint64_t>*>(p);
    }
  std::mt19937 rng(seed * 7919u + (unsigned)rank);
  auto pause_us = [&](int max_us) {
    if (max_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % (unsigned)(max_us + 1)));
  };
  int first = 0, end = 0;
  shard_bounds(lay.total_rows, rank, world, &first, &end);
  uint64_t posted = 0, waited = 0;
  long violations = 0, timeouts = 0, wrong = 0, refused = 0;
  const auto t_start = Clock::now();
  auto post = [&](uint64_t e) {
    if (!force && !hsm::exchange_post_is_safe(e, waited, lay.depth)) {
      ++refused;  // (the runtime's own check; never trips when the caller keeps to `lag`)
      return;
    }
    // peers in a per-epoch random order, rows in chunks with pauses in between: posts of different ranks interleave
    std::vector<int> order((size_t)world);
    for (int r = 0; r < world; ++r) order[(size_t)r] = r;
    std::shuffle(order.begin(), order.end(), rng);
    for (int p : order) {
      std::atomic<uint64_t>* dst = box[(size_t)p] + lay.buffer_of(e) + (size_t)first * (size_t)lay.cols;
      const int n = (end - first) * lay.cols;
      for (int i = 0; i < n; ++i) {
        dst[i].store(hsm::exchange_pack(value_of(e, first + i / lay.cols, i % lay.cols), e), std::memory_order_relaxed);
        if ((rng() & 1023u) == 0) pause_us(30);
      }
    }
    posted = e;
  };
  auto wait = [&](uint64_t e) {
    const std::atomic<uint64_t>* src = box[(size_t)rank] + lay.buffer_of(e);
    const size_t n = lay.buffer_granules();
    const auto t0 = Clock::now();
    for (size_t i = 0; i < n; ++i) {
      for (;;) {
        const uint64_t g = src[i].load(std::memory_order_relaxed);
        if (hsm::exchange_carries(g, e)) {
          if (hsm::exchange_value(g) != value_of(e, (int)(i / (size_t)lay.cols), (int)(i % (size_t)lay.cols))) ++wrong;
          break;
        }
        // what the buffer may legitimately still hold: zeros (never used) or epoch e - depth.  Anything else is a LATER
        // epoch written over rows this rank has not unpacked: the flow control failed.
        const uint32_t tag = (uint32_t)(g >> 32);
        const bool stale_ok = g == 0 || (e > (uint64_t)lay.depth && tag == (uint32_t)(e - (uint64_t)lay.depth));
        if (!stale_ok) {
          ++violations;
          break;
        }
        if (Clock::now() - t0 > std::chrono::seconds(20)) {
          ++timeouts;
          break;
        }
        std::this_thread::yield();
      }
      if (timeouts) break;
    }
    waited = e;
  };
  for (uint64_t e = 1; e <= epochs && !timeouts; ++e) {
    // the "matcher": ranks are skewed (rank 0 fast, the last rank slow), with occasional long stalls
    pause_us(20 + 60 * rank);
    if ((rng() % 37u) == 0) pause_us(3000);
    const bool wait_first = lag > 0 && (rng() & 1u) != 0;  // inside one launch the two parts run in either order (lag 0: the wait needs the own post)
    const uint64_t w = e > (uint64_t)lag ? e - (uint64_t)lag : 0;
    if (wait_first && w) wait(w);
    post(e);
    if (!wait_first && w) wait(w);
  }
  while (waited < posted && !timeouts) wait(waited + 1);  // drain
  // keep the mailbox mapped until every peer has drained too: a rank that finishes early must not unlink rows others still read
  const double secs = std::chrono::duration<double>(Clock::now() - t_start).count();
  printf("{\"rank\": %d, \"epochs\": %llu, \"posted\": %llu, \"waited\": %llu, \"violations\": %ld, \"timeouts\": %ld, \"wrong_values\": %ld, "
         "\"refused_posts\": %ld, \"depth\": %d, \"lag\": %d, \"seconds\": %.3f}\n",
         rank, (unsigned long long)epochs, (unsigned long long)posted, (unsigned long long)waited, violations, timeouts, wrong, refused,
         lay.depth, lag, secs);
  return (violations || timeouts || wrong || refused) ? 1 : 0;
}
