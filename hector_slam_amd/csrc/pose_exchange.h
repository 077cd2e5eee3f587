// pose_exchange.h -- the mailbox protocol behind the multi-GPU pose gather (the call being sharded is
// MapRepMultiMap::matchData, HSL/slam_main/MapRepMultiMap.h:116-132, as a batch over pose hypotheses / scans).
//
// The only exchange step of the sharded path is "every rank ends up with every rank's [B/G, 3] poses".  Round 5 did it with
// a collective per batched match (ncclAllGather / torch.distributed): 45 us of host time per enqueue and an RCCL kernel
// that takes CUs from the matcher's single generation of workgroups -- 103 us per step against 58.5 without.  The payload
// is 48 KiB per rank: nothing about it needs a collective library.  This protocol is plain stores:
//
//   * every rank owns a MAILBOX in its own HBM (uncached / fine-grained device memory, exported once with hipIpcGetMemHandle
//     or, inside one process, used through peer access): `depth` buffers of [total_rows][cols] 8-byte GRANULES
//     { value bits : 32 | epoch tag : 32 };
//   * POST (epoch e): a rank writes its rows -- granule by granule, one system-scope 8-byte store each (global_store_dwordx2
//     sc0 sc1: single-copy atomic on every path, xGMI included) -- into buffer e % depth of EVERY rank's mailbox, its own too;
//   * WAIT (epoch e): a rank polls the granules of buffer e % depth of its own mailbox until each carries tag e and unpacks
//     the values into a dense [total_rows][cols] fp32 array.  A granule carries its own tag, so there is no flag, no fence and
//     no ordering requirement between granules: data and "it has arrived" travel in the same store.
//
// Flow control without acknowledgements: a rank posts epoch e only behind (stream order) its wait for epoch e - 1 - lag, so when
// a buffer comes up for reuse every rank has long unpacked what it held -- provided depth >= 2 + 2 lag (min_depth below;
// derivation in DESIGN.md 6).  lag = 0: post and wait of the same epoch in one launch (the synchronous gather hsm_group_* uses);
// lag = 1: the wait for batch k's poses runs behind batch k+1's matcher, so the transfer hides behind a whole launch.
//
// This header is shared by the device kernels (pose_exchange.hip), the host runtime and the CPU model of the protocol that
// tests/cpp/exchange_model.cpp runs between two processes over shared memory: one definition of the layout and the tags.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HSM_XHD __host__ __device__
#else
#define HSM_XHD
#endif

namespace hsm {

constexpr int kExchangeMaxWorld = 16;

struct ExchangeLayout {
  int world;       // ranks
  int total_rows;  // rows of the gathered array (all ranks' shards, hsm_shard_bounds order)
  int cols;        // floats per row (3: pose; 9: H)
  int depth;       // buffers per mailbox

  HSM_XHD size_t buffer_granules() const { return (size_t)total_rows * (size_t)cols; }
  HSM_XHD size_t granules() const { return (size_t)depth * buffer_granules(); }
  HSM_XHD size_t bytes() const { return granules() * sizeof(uint64_t); }
  // first granule of the buffer epoch e travels in
  HSM_XHD size_t buffer_of(uint64_t epoch) const { return (size_t)(epoch % (uint64_t)depth) * buffer_granules(); }
};

// epochs count from 1 (a zero-filled mailbox matches no epoch of its first `depth` uses); the tag is the epoch's low word:
// what a buffer held before is epoch e - depth, a different tag for every depth that is not a multiple of 2^32
HSM_XHD inline uint64_t exchange_pack(uint32_t value_bits, uint64_t epoch) { return ((uint64_t)(uint32_t)epoch << 32) | value_bits; }
HSM_XHD inline bool exchange_carries(uint64_t granule, uint64_t epoch) { return (uint32_t)(granule >> 32) == (uint32_t)epoch; }
HSM_XHD inline uint32_t exchange_value(uint64_t granule) { return (uint32_t)granule; }
HSM_XHD inline int exchange_min_depth(int lag) { return 2 + 2 * lag; }
// may a rank that has waited for every epoch <= waited post `epoch` into a mailbox of `depth` buffers?  (the host runtime
// refuses a post that could overwrite rows a peer has not unpacked yet)
HSM_XHD inline bool exchange_post_is_safe(uint64_t epoch, uint64_t waited, int depth) {
  // posting e needs the own wait for e - 1 - lag complete, with depth >= 2 + 2 lag: lag <= (depth - 2) / 2
  const uint64_t lag_max = (uint64_t)((depth - 2) / 2);
  return epoch <= waited + 1 + lag_max;
}

// What a matcher launch needs to take part in an exchange itself (MatchParams::xp; world == 0: it does not): its epilogue posts every
// scan's pose to every rank's mailbox, and `wait_blocks` extra workgroups at the end of its grid -- dispatched as the matcher's own
// workgroups retire -- wait for an earlier epoch and unpack it.  No launch of its own for the exchange at all.
struct ExchangeFused {
  uint64_t* peer[kExchangeMaxWorld];  // every rank's mailbox as mapped in this process
  float* out;                         // [total_rows][cols] of the waited epoch, or nullptr
  unsigned* status;                   // pinned host words (timeouts)
  unsigned long long post_off;        // first granule of the posted epoch's buffer + first_row * cols
  unsigned long long wait_off;        // first granule of the waited epoch's buffer
  unsigned long long timeout_ticks;
  unsigned post_tag, wait_tag;        // low words of the epochs (wait_blocks == 0: nothing to wait for)
  int world, rank, cols, total_granules, match_blocks, wait_blocks;
};

}  // namespace hsm
