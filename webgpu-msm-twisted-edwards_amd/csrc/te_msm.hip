// te_msm.hip -- context, stage orchestration and the C-ABI of libtemsm.so (include/te_msm.h).
//
// Host-side counterpart of compute_msm's orchestration (submission/submission.ts:73-413) and of the
// WebGPU wrapper it drives (implementation/cuzk/gpu.ts:14-229).  Differences that matter:
//   * the context is persistent -- device buffers, streams and kernels survive across calls (the
//     reference re-creates device, buffers and pipelines on every call, submission.ts:96-97,360);
//   * every stage of one MSM is enqueued on one HIP stream with no host synchronisation in between, like the
//     reference's single command-encoder submit (gpu.ts:118); several MSMs are kept in flight on separate
//     work sets / streams so that they overlap on the device;
//   * windows can be sharded over contexts / devices; the reference is single-device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <deque>
#include <algorithm>
#include <functional>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <memory>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/te_msm.h"
#include "host_tail.hpp"
#include "synth.hpp"
#include "kernels.hip.hpp"
#include "probe.hip.hpp"
#include "host_tail377.hpp"
#include "host_sched.hpp"

namespace {

thread_local std::string g_init_error;

// in execution order: everything that needs only the scalars first, so that a host-buffer call can upload the points
// (2/3 of the bytes) while those stages already run
enum { ST_DIGITS = 0, ST_SCATTER, ST_BSORT, ST_ORDER, ST_PREP, ST_ACCUM, ST_TREE, ST_WEIGHTED, ST_COUNT };
// the last two entries are not event intervals: k_accumulate's own first-wave-in .. last-wave-out device clock in ms, and the
// mean shader clock over that span in GHz (not a time: te_msm_stage_ms reports it under its own name)
const char* const kStageNames[ST_COUNT + 2] = {"digits", "part_scatter", "bucket_sort", "order", "prep_points",
                                               "accumulate", "marginal_sums", "weighted_sum", "accumulate_on_device", "accumulate_core_clock_ghz"};

// per-curve sizes: wire format, device accumulator, record slot, partial row (5 points), result
struct curve_sizes { size_t point_in, scalar_in, acc, rec, row, result; };
inline curve_sizes sizes_of(int curve) {
  return curve == TE_MSM_CURVE_BLS12_377_G1 ? curve_sizes{96, 48, sizeof(te::ete_t<14>), sizeof(te::rec_slot<14>), TE377_TAIL_ROW_BYTES, 96}
                                            : curve_sizes{TE_MSM_POINT_BYTES, TE_MSM_SCALAR_BYTES, sizeof(te::ete), sizeof(te::rec_slot<9>), TE_MSM_PARTIAL_BYTES, 64};
}
// bytes of one record: the curve's own (converted per call), or -- rec_kind 1 -- the affine BLS12-377 record of a bound point set
inline size_t rec_bytes_of(int curve, int rec_kind) { return (curve == TE_MSM_CURVE_BLS12_377_G1 && rec_kind == 1) ? sizeof(te::rec_aff377) : sizes_of(curve).rec; }
static_assert(TE377_TAIL_ROW_BYTES == TE_MSM_PARTIAL_BYTES_BLS12_377 && sizeof(te::ete_t<14>) * 5 == TE377_TAIL_ROW_BYTES, "row layout");
constexpr size_t TE_MAX_ROW_BYTES = TE377_TAIL_ROW_BYTES;

struct plan_t {
  int curve = 0;                      // TE_MSM_CURVE_*
  int c = 0, W = 0, nw = 0;           // window bits, total windows, windows of this launch sequence (shard windows x batch)
  int nw1 = 0, batch = 1;             // windows of this shard per MSM; MSMs that share the launch sequence (te_msm_partial_device_batch)
  int w_first = 0, w_step = 1;        // the windows this launch sequence computes: w_first + k * w_step (the device's shard, or 0 / 1 = all: whole MSMs)
  uint32_t B = 0, logB = 0;           // buckets per window = 2^(c-1) (signed digits) or 2^c (unsigned)
  int signed_digits = 1;
  uint32_t dw[4] = {0, 0, 0, 0};      // bits of the four digits of a bucket index (dw[0] lowest)
  uint32_t CH = 0, chunk_len = 0, nst = 0;   // level-1 chunks per window (chunk_len multiple of 4096); padded row stride
  uint32_t seg_len = 64;
  uint32_t S = 0, logS = 0, P = 0;   // level-1 partition: S buckets each, P = B/S partitions per window
  uint32_t packed = 0;               // level-1 entries as one 32-bit word (index | key << 23 | sign << 31): n <= 2^23
  int rec_kind = 0;                  // records k_accumulate gathers: 0 = the curve's own, 1 = affine BLS12-377 (bound point sets)
  // fixed-base windows (bound point sets with a per-window table, kernels.hip.hpp k_fb_digits): fb_rb > 0 -- the "windows" of this
  // plan are fb_rb + 1 pseudo-windows (rows) of 2^15 buckets over ONE bucket set; fb_c / fb_W: window bits and windows of the real decomposition
  int fb_rb = 0, fb_c = 0, fb_W = 0;
};

struct graph_key { const void *pts, *sc, *out; uint64_t n, generation; int c, w_first, w_step, seg_len, sort; };

// Everything one MSM in flight needs on the device.  A GPU owns TE_MSM_WORKSETS work sets with their own streams so that
// several MSMs overlap ON THE DEVICE: the launch gaps and the latency-bound reduction tail of one are filled by the wide kernels of the
// other (te_msm_submit_device alternates them; "workset" option for te_msm_partial_device callers).
struct workset_t {
  hipStream_t stream = nullptr, copy_stream = nullptr;   // copy_stream: host-buffer uploads beside the compute stream (created on first use)
  int hw_queue_class = -1;            // which of the measured hardware-queue classes `stream` is on (-1: not probed)
  hipEvent_t ev_copy = nullptr, ev_start = nullptr;
  size_t cap[40] = {};                                  // per-buffer capacity in bytes (ensure())
  uint8_t* d_recs = nullptr;          // record slots of the plan's curve (te::rec_slot<N>)
  uint16_t *d_digits = nullptr, *d_part_keys = nullptr;
  uint32_t *d_counts1 = nullptr, *d_part_start = nullptr, *d_part_count = nullptr, *d_part_idx = nullptr, *d_seg_part_base = nullptr;
  uint32_t *d_bucket_count = nullptr, *d_bucket_start = nullptr, *d_bucket_cursor = nullptr, *d_sorted = nullptr;
  uint32_t *d_seg_base = nullptr, *d_seg_bucket = nullptr, *d_seg_lenv = nullptr, *d_order = nullptr;
  uint32_t *d_split_list = nullptr, *d_chunk_list = nullptr;
  uint8_t *d_seg_out = nullptr, *d_buckets = nullptr, *d_red[4] = {};   // accumulators of the plan's curve (te::ete_t<N>); d_red: ping/pong of the two fold chains
  // ONE zeroed block per MSM (a single memset), words: [0] final-carry flag, [2..3] non-zero window digits as ONE 64-bit count (= entries
  // accumulated; W * n passes 2^32 inside the allowed range n < 2^31: round-5 advisor; [0..3] survive the pieces of a host-buffer MSM),
  // [4] number of segments, [5..7] split / giant
  // bucket counters, [Z_ROWS..) the partial rows of the MSM (so that flag and rows come back in ONE device-to-host copy),
  // [Z_HIST..) segment-length histogram (TE_HIST_COPIES copies), [Z_CURSOR..) reservation cursors of the schedule, [Z_END..) the level-1 histogram
  // counts1[window][chunk][partition], then bucket_count[window][bucket].  d_err .. d_bucket_count point into d_zero.
  uint32_t *d_zero = nullptr; size_t zero_words = 0;
  size_t zero_clean_words = 0;        // words of d_zero known to be zero on the set's stream: the block is cleared AFTER an MSM's read-back
                                      // (finish_sequence), so that the next MSM on the set starts with its first kernel, not a fill
  uint32_t *d_err = nullptr, *d_num_seg = nullptr, *d_size_hist = nullptr, *d_size_cursor = nullptr;
  uint32_t* d_part_ticket = nullptr;  // [window][partition]: pieces of a multi-piece partition counted so far (k_l2_local), behind bucket_count in the zeroed block
  uint8_t* d_partials = nullptr;      // = d_zero + Z_ROWS: TE_MAX_WINDOWS rows
  uint32_t* h_err = nullptr;          // pinned: mirror of d_zero[0 .. Z_ROWS + rows)
  uint8_t* h_partials = nullptr;      // = h_err + Z_ROWS
  uint32_t* h_err_dev = nullptr;      // the device-visible address of h_err: k_reduce_tail writes flag words and rows there itself
  bool rows_on_host = false;          // the last launch sequence did so: fetch_rows has nothing to copy
  hipEvent_t ev_done = nullptr;       // the set is free again (everything of its last MSM, the clearing of the zeroed block included)
  hipEvent_t ev_result = nullptr;     // flag + rows of its last MSM are in host memory (recorded before the clearing: what a caller waits for)
  hipEvent_t ev[ST_COUNT + 1] = {};
  plan_t plan; uint64_t n = 0; bool used = false;
  te_sched::slot_t slot;              // ticket of an MSM submitted on this set and not collected yet (0 = none) + the host-thread job of an asynchronous submit
  std::string job_err;                // why that job failed (written by the device's host thread before the job is marked done)
  int prof_level = 0;                 // profile level the events of the last enqueue were recorded at
  hipStream_t last_stream = nullptr;  // stream of the previous MSM on this set: a different one must wait for it (scratch reuse)
  uint64_t generation = 0;            // bumped whenever ensure() reallocates a buffer of this set
  hipGraphExec_t g_front = nullptr, g_back = nullptr; graph_key g_key = {};
  // host-buffer MSMs (te_msm_run, te_msm_submit): device copies of the caller's buffers, sized in bytes for the curve of the
  // call, and the "piece i has arrived" events of an upload in pieces
  void *d_in_points = nullptr, *d_in_scalars = nullptr; size_t cap_in_points = 0, cap_in_scalars = 0;
  std::vector<hipEvent_t> piece_events;
  // option "host_staging": the set's own pinned ring for host-buffer uploads (allocated on first use; te_msm_trim / destroy free it)
  uint8_t* h_ring = nullptr; std::vector<hipEvent_t> ring_ev; size_t ring_next = 0;
  uint64_t idle_calls = 0;            // te_msm_trim: context-level calls since the set was last used
  uint32_t* d_fb_remap = nullptr;     // fixed-base windows: [rows][cap] table index | sign << 31 of every entry (beside d_digits' codes)
  uint32_t* d_fb_fill = nullptr;      // ... [rows] entries reserved per row, in the zeroed block
  const void* fb_scalars = nullptr; uint64_t fb_n = 0;   // ... the scalars (device memory) of the MSM in flight: a row overflow falls back to the ordinary windows
  int slab = -1;                      // shared record slab the MSM in flight on this set uses (-1: its own d_recs, or a bound point set)
  const uint8_t* recs_last = nullptr; // where the records of the set's last MSM are (te_msm_debug_read "records"): d_recs or a shared slab
  te_bases* bound = nullptr;          // the bound point set the set's ticket in flight gathers from (te_msm_submit_scalars*): released only after the collect
};
constexpr int TE_MAX_WINDOWS = 64;    // window_bits >= 4
// words [Z_CLOCK, Z_ROWS): k_accumulate's profiling words, 4 x TE_CLK_SLOTS 64-bit values (first wave in / last wave out on the
// wall clock, core and wall ticks summed over the waves), see the kernel
constexpr size_t Z_KEEP = 4;            // words [0, Z_KEEP) are kept from piece to piece of one host-buffer MSM (flag, pad, 64-bit entry count)
constexpr size_t Z_ENTRIES = 2;         // the 64-bit entry count (8-byte aligned)
constexpr size_t Z_CLOCK = 8;
static_assert(Z_KEEP + 4 <= Z_CLOCK && Z_ENTRIES % 2 == 0 && Z_ENTRIES + 2 <= Z_KEEP, "flag words");
constexpr size_t Z_ROWS = Z_CLOCK + 4 * 2 * TE_CLK_SLOTS, Z_HIST = Z_ROWS + (size_t)TE_MAX_WINDOWS * TE_MAX_ROW_BYTES / 4, Z_CURSOR = Z_HIST + 1024 * TE_HIST_COPIES, Z_END = Z_CURSOR + 1024;

struct gpu_t {
  int device = 0;
  int w_first = 0, w_step = 1;
  workset_t ws[TE_MSM_WORKSETS];
  int last_ws = 0;
  int wall_clock_khz = 0;                // rate of wall_clock64() on this device
  int in_flight = 0;                     // submitted and not collected
  bool queues_probed = false;            // the hardware-queue measurement has run (spread_streams_over_queues)
  // SHARED RECORD SLABS (round 6): whole-MSM calls from device-resident inputs that name the SAME point buffer while they are in
  // flight convert into, and gather from, ONE record slab instead of one per work set (acquire_shared_recs).
  // users: work sets whose latest MSM names the slab.  pending / ev[]: users that let go of it while their MSM could still be running (the
  // building blocks te_msm_partial_device[_batch], whose completion the engine does not see): an event recorded behind that MSM; whoever
  // takes the slab for ANOTHER point buffer waits for those events first (joining the same buffer needs no wait: the same bytes).
  struct rec_slab_t { const void* src = nullptr; uint64_t n = 0; int curve = 0; uint8_t* d = nullptr; size_t cap = 0; int users = 0;
                      uint32_t pending = 0; hipEvent_t ev[TE_MSM_WORKSETS] = {}; };
  std::vector<rec_slab_t> slabs;
  // asynchronous scalars-only tickets (bound bases) cross the link ONE AT A TIME per device: lanes that upload side by side share the
  // link, every ticket reaches the device late and the tickets move in a convoy (two in flight: 1.05-1.23 ms per MSM with 2-4 lanes,
  // 0.90-0.92 with one; tools/exp_bound_lanes_depth.py).  Host-buffer tickets (points + scalars) do not take it: their pageable copies
  // fill each other's pin / unpin gaps (1.79 vs 1.89 ms with one lane).  TE_MSM_SCALAR_UPLOADS_SERIAL=0: off (experiments).
  std::unique_ptr<std::mutex> scalar_link{new std::mutex};
  // EXPERIMENT TE_MSM_SERIAL_ACCUMULATE=1 (tools/exp_serial_accumulate.py): the k_accumulate launches of a device form a chain -- each waits
  // for the one enqueued before it on another stream --, so that ONE accumulation runs at full width beside the small kernels of the other
  // MSMs in flight instead of two at half speed
  std::unique_ptr<std::mutex> acc_mu{new std::mutex};
  hipEvent_t acc_ev[16] = {}; uint32_t acc_next = 0; hipEvent_t acc_prev = nullptr; hipStream_t acc_prev_stream = nullptr;
  bool streams_exported = false;         // te_msm_workset_stream handed a handle out: te_msm_destroy parks the streams instead of destroying them
  bool streams_final = false;            // ... and the work sets' streams will not be re-dealt any more
};

}  // namespace

struct te_ctx {
  std::vector<gpu_t> devs;
  std::string err;
  std::mutex err_mu;           // the per-device host threads of a multi-device te_msm_run report into the one string
  std::vector<std::unique_ptr<te_sched::worker_t>> workers;   // devs[i]'s host thread (host_sched.hpp); created by the first call that needs them
  std::vector<std::unique_ptr<te_sched::worker_t>> lanes;     // the upload lanes of asynchronous tickets: opt_upload_threads per device (host_sched.hpp)
  uint64_t next_lane = 0;
  int opt_upload_threads = 4;       // option "upload_threads" (env TE_MSM_UPLOAD_THREADS)
  uint64_t next_ticket = 1;         // tickets are handed out in order, over all devices; a ticket lives on the work set whose slot holds it
  int last_dev = -1;                // the device the previous ticket went to (te_sched::pick_device deals idle devices round-robin)
  int opt_host_staging = 0;         // host-buffer uploads through the work sets' own pinned rings (staged_copy) instead of straight from the caller's memory
  std::vector<std::unique_ptr<te_sched::worker_t>> stagers;   // the threads that fill those rings (created on first use)
  int opt_stage_device_inputs = 0;  // tickets for device-resident inputs: copy them to the chosen device even when it is the one that holds them (tests on a one-GPU box)
  int64_t stat_peer_bytes = 0;      // bytes those copies moved (get_option "peer_bytes")
  int64_t stat_fb_fallbacks = 0;    // fixed-base MSMs whose rows overflowed (skewed scalars) and were run again with the ordinary windows
  int64_t stat_entries = 0;         // non-zero window digits (= accumulated entries) of the MSM whose result was fetched last (get_option "entries_accumulated")
  std::vector<double> host_split;   // TE_MSM_HOST_SPLIT (relative piece weights of a host-buffer upload; experiments), read once
  int opt_host_shard_min = 4096;    // multi-device te_msm_run: points per device below which fewer devices are used
  int opt_queue_probe = 1;          // 1 = the first te_msm_submit* measures the hardware queues (lazily); 0 = never; te_msm_probe_queues does it now
  int opt_window_bits = 0;
  int opt_sort = 1;
  int opt_curve = TE_MSM_CURVE_TE_BLS12;   // which group the point buffers are in (option "curve")
  int opt_signed = 1;          // signed window digits (the reference's shipped behaviour); 0 = plain unsigned windows, 2^c buckets
  int opt_profile = 0;
  int opt_seg_len = 0;         // work segment: at most this many entries of one bucket per thread; 0 = from n (make_plan)
  int opt_host_chunks = 0;     // te_msm_run: pieces a large host buffer is uploaded and processed in (0 = choose from n)
  int opt_graph = 0;           // replay the launch sequence around k_accumulate as HIP graphs
  int opt_workset = 0;         // work set used by te_msm_run* / te_msm_partial_device
  int opt_fuse_prep = 1;       // device-resident inputs: convert the points in the launch of the sort's first level (k_part_scatter_prep)
  int opt_fold_pairs = 1;      // first fold level of small MSMs with two lanes per output (k_sum_groups<N, true>)
  int opt_packed = 1;          // level-1 sort entries as one 32-bit word where n <= 2^23 (make_plan)
  int opt_prezero = 1;         // clear a work set's zeroed block behind an MSM's read-back instead of in front of the next MSM's first kernel
  float stage_ms[ST_COUNT + 2] = {};
  bool have_stage_ms = false;
  int64_t stat_peer_copies = 0;  // hipMemcpyPeerAsync calls issued so far (multi-device contexts fed from device 0's memory; get_option "peer_copies")
  std::vector<te_bases*> bases;  // the bound point sets of the context (te_msm_bind_points), freed by te_msm_release_points / te_msm_destroy
  int opt_bind_affine = 1;       // te_msm_bind_points, BLS12-377: affine records (one inversion per point, once) instead of the projective ones of the per-call conversion
  int opt_scalar_chunks = 0;     // te_msm_run_scalars / te_msm_submit_scalars: pieces the scalars of a bound set are uploaded and processed in (0 = from n)
  int opt_bind_fixed_base = 0;   // te_msm_bind_points (Twisted-Edwards curve): window bits c of a per-window table 2^(c w) P_i (16..21; 0 = none): MSMs over
                                 // the set then run fixed-base windows -- one bucket set for all windows (W x the record memory)
  int opt_share_records = 1;     // shared record slabs for calls in flight that name the same device-resident point buffer (A/B: option "share_records", env TE_MSM_SHARE_RECORDS)
  int opt_lane_host_waits = 1;   // asynchronous tickets (a lane thread enqueues them): the thread WAITS for each upload before it enqueues the kernels that read
                                 // it, instead of putting a stream wait in front of them (see lane_wait; A/B: option "lane_host_waits", env TE_MSM_LANE_HOST_WAITS)
  int opt_exp_table_replicas = 1; // EXPERIMENT (profiles/r06_fixed_base_windows.txt): te_msm_bind_points keeps this many copies of the records and
                                 // the windows of a device-scalar MSM gather from different copies -- the gather footprint of a per-window table
};

// A bound point set (include/te_msm.h, "resident bases"): the records of n points on EVERY device of its context, converted
// once.  The reference hands the same point buffer to compute_msm for every run of a size (full_benchmarks.ts:63-68,100-105).
struct te_bases {
  te_ctx* ctx = nullptr;
  uint64_t n = 0;
  int curve = TE_MSM_CURVE_TE_BLS12, rec_kind = 0;
  size_t rec_bytes = 0;
  std::vector<uint8_t*> recs;    // recs[i]: n records in the memory of ctx->devs[i]
  int replicas = 1;              // experiment "exp_table_replicas": copies of the records behind each other in recs[i]
  int fb_c = 0, fb_W = 0;        // fixed-base windows: recs[i] holds fb_W tables of n records, table w = records of 2^(fb_c w) P_i (table 0 = the ordinary records)
  int in_flight = 0;             // tickets not collected that gather from it (the set cannot be released under them)
};

namespace {

#define HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      char buf_[512];                                                                            \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      { std::lock_guard<std::mutex> lk_((ctx)->err_mu); (ctx)->err = buf_; }                     \
      return TE_MSM_EDEVICE;                                                                     \
    }                                                                                            \
  } while (0)

int set_err(te_ctx* ctx, int code, const char* msg) { std::lock_guard<std::mutex> lk(ctx->err_mu); ctx->err = msg; return code; }

// The engine selects devices with hipSetDevice, which is state of the CALLING thread: every entry point that may do so puts the
// caller's device back when it returns.  (A host that shares the thread with another HIP user -- PyTorch in bench.py's rank 0,
// which opens all N devices in its own process -- would otherwise find its "current device" moved to wherever the last ticket
// went: tensors on the wrong GPU, collectives on the wrong communicator.)
struct device_guard {
  int prev = -1;
  device_guard() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
  ~device_guard() { if (prev >= 0) (void)hipSetDevice(prev); }
  device_guard(const device_guard&) = delete;
  device_guard& operator=(const device_guard&) = delete;
};

uint32_t ilog2(uint32_t v) { uint32_t l = 0; while ((1u << l) < v) l++; return l; }

int auto_window_bits(uint64_t n) {
  // measured on MI355X (profiles/r01_window_sweep.txt, r04_n_sweep.txt, r04_n_sweep_small.txt): 16 bits from 3 * 2^17 points up,
  // 15 bits from 2^16 (fewer buckets to reduce, and at that size the fixed stages weigh more than the additions; at n = 2^18
  // 15 bits: 0.295 ms per MSM pipelined / 0.479 ms latency, 16 bits: 0.328 / 0.503; at 2^19 16 bits win the latency by 2 %, 15 bits
  // the throughput by 3 %).  Below the harness sizes the bucket reduction is the whole cost: 13 bits for 2^14 .. 2^16 (2^15: 0.099
  // ms per MSM / 0.235 ms latency against 0.112 / 0.247 with 15 bits), 11 bits for 2^11 .. 2^14 (2^13: 0.080 / 0.200 against
  // 0.095 / 0.235 with 14), below that about log2(n) + 1 bits so that the W * 2^(c-1) buckets do not dwarf the n points.
  if (n >= (3ull << 17)) return 16;
  if (n >= (1ull << 16)) return 15;
  if (n >= (1ull << 14)) return 13;
  if (n >= (1ull << 11)) return 11;
  int lg = 0; while ((1ull << (lg + 1)) <= n) lg++;
  int c = lg + 1;
  if (c < 8) c = 8;
  if (c > 11) c = 11;
  return c;
}

// Windows of the decomposition.  Unsigned digits cover all 256 bits of a scalar record.  Signed digits: the scalars of this
// boundary are below p < 2^253 (harness generator reference/webgpu/utils.ts:118-124), so the windows only have to reach bit 254
// -- 255 bits: 17 windows of 15 bits where 256 bits take 18 (and 51 of 5 instead of 52; every other size of the allowed range
// c in [4, 16] gives the same count).  Which scalars above p are still ACCEPTED therefore depends on c -- and through the
// automatic window size on n: everything below 2^(cW) - sum_w 2^(cw + c - 1), i.e. about 2^255.9 for 16 x 16 bits (the reference's
// plan, miscellaneous/utils.ts:52-95) and 2^254 - 2^240 for 17 x 15 bits (n = 2^16 .. 2^18); canonical scalars (< p < 2^253) always
// fit.  INTEGRATION.md section 5 states the range; option "window_bits" = 16 gives a caller the reference's own acceptance at any n.
// The 18th window of a 15-bit plan held bits 255..269 and never received a digit or a carry: 16 384 empty buckets folded and a
// tail block run for nothing at n = 2^16 .. 2^18, where the reduction is ~40 % of the device span.  The error rule stays exact:
// k_digits flags any bit of s + sum_w 2^(cw + c - 1) at or above c * W, the reference's "final carry is 1"
// (miscellaneous/utils.ts:80-83) -- with c * W = 255 that is every scalar from 2^254 - 2^240 up (all of them >= 2 p).
int num_windows_for(int c, int signed_digits) { return ((signed_digits ? 255 : 256) + c - 1) / c; }

// whole: every window (a whole MSM: host buffers, tickets of a multi-device context) instead of the device's window shard
void make_plan(const te_ctx* ctx, const gpu_t& d, uint64_t n, plan_t& p, int force_c = 0, int batch = 1, uint32_t force_seg = 0, bool whole = false) {
  p.curve = ctx->opt_curve;
  p.c = force_c ? force_c : ctx->opt_window_bits ? ctx->opt_window_bits : auto_window_bits(n);
  p.signed_digits = ctx->opt_signed;
  p.W = num_windows_for(p.c, p.signed_digits);
  p.w_first = whole ? 0 : d.w_first; p.w_step = whole ? 1 : d.w_step;
  p.nw = 0;
  for (int w = p.w_first; w < p.W; w += p.w_step) p.nw++;
  p.nw1 = p.nw; p.batch = batch; p.nw *= batch;
  p.logB = (uint32_t)(p.signed_digits ? p.c - 1 : p.c); p.B = 1u << p.logB;
  for (int k = 0; k < 4; k++) p.dw[k] = (p.logB + 3u - (uint32_t)k) / 4u;       // 15 -> 4,4,4,3
#ifndef TE_SCATTER_BLOCKS
#define TE_SCATTER_BLOCKS 1024u       // measured at n = 2^20: 512 blocks 81.8 us, 1024 80.3, 2048 98.4, 4096 264 (every block sums P x CH counts)
#endif
  uint32_t ch = p.nw > 0 ? TE_SCATTER_BLOCKS / (uint32_t)p.nw : 1u;     // ~4 blocks of 512 threads per CU in k_part_scatter
  if (ch < 1) ch = 1;
  if (ch > 256) ch = 256;
#ifndef TE_SCATTER_MINCHUNK
#define TE_SCATTER_MINCHUNK 4096u      // one tile (8192: k_part_scatter_prep 12.1 -> 10.7 us at n = 2^16, equal from 2^18 on)
#endif
  const uint32_t by_n = (uint32_t)((n + TE_SCATTER_MINCHUNK - 1) / TE_SCATTER_MINCHUNK);      // at least one tile of digits per chunk
  if (ch > by_n) ch = by_n ? by_n : 1;
  p.nst = (uint32_t)((n + 7) & ~(uint64_t)7);
  p.chunk_len = (uint32_t)((p.nst + ch - 1) / ch);
  p.chunk_len = (p.chunk_len + 4095u) & ~4095u;               // whole 4096-entry tiles
  ch = (p.nst + p.chunk_len - 1) / p.chunk_len;
  p.CH = ch;
  // level-1 partition = S consecutive buckets of one window
  p.S = p.B < 256u ? p.B : 256u;
  if (force_seg) {
    p.seg_len = force_seg;               // the pieces of one host-buffer MSM share the geometry (and the buffers) of the largest
  } else if (ctx->opt_seg_len) {
    p.seg_len = (uint32_t)ctx->opt_seg_len;
  } else {
    // Twice the mean bucket size, as a power of two in [16, 64].  A thread works one segment off serially (~8-11 us per
    // addition with four waves per SIMD): at n = 2^20 (32 entries per bucket) 64 balances the longest chains against the
    // VALU-bound whole, at smaller n the whole shrinks and long segments set the kernel's duration -- measured latency of
    // one MSM (profiles/r03_segment_len_sweep.txt): 2^16 0.366 -> 0.346 ms, 2^17 0.466 -> 0.410, 2^18 0.582 -> 0.53-0.54
    // with 16 / 16 / 16-32 instead of 64; 2^19 equal at 32; 2^20 1.23 (64) against 1.32 (32).
    uint64_t a = (2 * n + p.B - 1) / p.B, s2 = 16;
    while (s2 < a && s2 < 64) s2 <<= 1;
    p.seg_len = (uint32_t)s2;
  }
  p.P = p.B / p.S; p.logS = ilog2(p.S);
  // 4 bytes per level-1 entry instead of 6 (-32 MB written and -32 MB read per 2^20-point MSM) where the index fits 23 bits;
  // option "packed_sort" = 0 (env TE_MSM_PACKED=0): the general form (A/B measurements, tests; what larger n uses)
  p.packed = (ctx->opt_packed && n <= (1ull << 23)) ? 1u : 0u;
}

template <typename T> int ensure(te_ctx* ctx, workset_t& ws, T*& ptr, size_t& cap_bytes, size_t need_elems) {
  const size_t need = need_elems * sizeof(T);
  if (ptr && need <= cap_bytes) return 0;
  ws.generation++;                     // captured graphs hold the old pointer
  if (ptr) HIP_TRY(ctx, hipFree(ptr));
  ptr = nullptr; cap_bytes = 0;
  HIP_TRY(ctx, hipMalloc((void**)&ptr, need ? need : 16));
  cap_bytes = need;
  return 0;
}

// need_recs = false: the launch sequence gathers from a bound point set, the work set needs no record slab of its own
int ensure_buffers(te_ctx* ctx, gpu_t& d, workset_t& ws, uint64_t n, const plan_t& p, bool need_recs = true) {
  HIP_TRY(ctx, hipSetDevice(d.device));
  // + 64: k_accumulate fetches its sorted indices TE_IDX_STRIP at a time and may read that far past the end of a list
  const size_t nd = (size_t)p.nw * p.nst + 64, wb = (size_t)p.nw * p.B, ab = sizes_of(p.curve).acc;
  int rc = 0;
  if (need_recs && (rc = ensure(ctx, ws, ws.d_recs, ws.cap[0], (size_t)n * sizes_of(p.curve).rec * (size_t)p.batch))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_digits, ws.cap[1], nd))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_sorted, ws.cap[2], nd))) return rc;
  {
    const size_t c1 = (size_t)p.nw * p.CH * p.P;
    ws.zero_words = Z_END + c1 + wb + (size_t)p.nw * p.P + 64;            // (+ 64: the row fill counters of fixed-base windows)
    { const uint32_t* before = ws.d_zero; if ((rc = ensure(ctx, ws, ws.d_zero, ws.cap[3], ws.zero_words))) return rc; if (ws.d_zero != before) ws.zero_clean_words = 0; }
    ws.d_err = ws.d_zero; ws.d_num_seg = ws.d_zero + Z_KEEP; ws.d_size_hist = ws.d_zero + Z_HIST; ws.d_size_cursor = ws.d_zero + Z_CURSOR;
    ws.d_partials = reinterpret_cast<uint8_t*>(ws.d_zero + Z_ROWS);
    ws.d_counts1 = ws.d_zero + Z_END; ws.d_bucket_count = ws.d_counts1 + c1; ws.d_part_ticket = ws.d_bucket_count + wb;
    ws.d_fb_fill = ws.d_part_ticket + (size_t)p.nw * p.P;
  }
  if (p.fb_rb && (rc = ensure(ctx, ws, ws.d_fb_remap, ws.cap[27], nd))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_bucket_start, ws.cap[5], wb))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_bucket_cursor, ws.cap[16], wb))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_seg_part_base, ws.cap[17], (size_t)p.nw * p.P))) return rc;
  const size_t smax = wb + (size_t)p.nw * (n / (uint64_t)p.seg_len) + 16;      // segments <= buckets + entries / seg_len
  if ((rc = ensure(ctx, ws, ws.d_order, ws.cap[7], smax))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_seg_bucket, ws.cap[20], smax))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_seg_lenv, ws.cap[21], smax))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_seg_out, ws.cap[22], smax * ab))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_seg_base, ws.cap[23], wb + 1))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_split_list, ws.cap[24], wb + 1))) return rc;
  // a giant bucket contributes one chunk per TE_GIANT_RUN parts: at most one per bucket plus one per TE_GIANT_RUN segments
  if ((rc = ensure(ctx, ws, ws.d_chunk_list, ws.cap[26], 2 * (wb + smax / TE_GIANT_RUN + 2)))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_part_start, ws.cap[6], (size_t)p.nw * p.P))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_buckets, ws.cap[8], wb * ab))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_part_count, ws.cap[9], (size_t)p.nw * p.P + (size_t)p.nw))) return rc;      // + the overflow pieces of each window
  if ((rc = ensure(ctx, ws, ws.d_part_keys, ws.cap[14], p.packed ? 8 : nd))) return rc;       // packed level-1 entries carry their key
  if ((rc = ensure(ctx, ws, ws.d_part_idx, ws.cap[15], nd))) return rc;
  // fold levels (by 8, 4 or 2): the first output is at most B/2 per window, the second at most B/4
  if ((rc = ensure(ctx, ws, ws.d_red[0], ws.cap[10], (wb / 2 + 1) * ab))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_red[1], ws.cap[11], (wb / 4 + 1) * ab))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_red[2], ws.cap[12], (wb / 2 + 1) * ab))) return rc;
  if ((rc = ensure(ctx, ws, ws.d_red[3], ws.cap[13], (wb / 4 + 1) * ab))) return rc;
  return 0;
}

template <int C> void launch_digits(const te::batch_ptrs& sc, int batch, uint16_t* dg, const te::digits_params& prm, uint32_t* err, uint32_t* counts1, hipStream_t s) {
  hipLaunchKernelGGL(te::k_digits<C>, dim3((prm.nst + TE_DIG_BLOCK - 1u) / TE_DIG_BLOCK, batch), dim3(TE_DIG_THREADS), 0, s, sc, dg, prm, err, counts1);
}
static_assert(TE_BATCH_MAX == TE_MSM_MAX_BATCH, "batch tables");

// One MSM's device work in three parts, so that the parts before and after the dominant kernel can be replayed as HIP
// graphs (one launch each instead of ~30: the host-side enqueue cost, ~0.35 ms, is what bounds small MSMs and the
// per-rank step of a window-sharded one) while k_accumulate stays an ordinary launch bracketed by timing events.
struct msm_launch {
  te_ctx* ctx; gpu_t& d; workset_t& ws; plan_t p;
  const void* d_points; const void* d_scalars; uint64_t n; void* d_partials_out;   // batch > 1: d_points / d_scalars are arrays of p.batch device pointers (on the host)
  int prof;                       // event marks inside front()/back() only at profile level 2 (never inside a capture)
  hipStream_t stream;
  bool own_rows = false;          // rows go to ws.d_partials: the caller fetches flag + rows with one copy
  bool onto = false;              // a later piece of a host-buffer MSM: keep the final-carry flag, add onto the buckets
  bool host_rows = false;         // own rows go straight to the work set's pinned host block, written by k_reduce_tail (no copy at all)
  uint8_t* recs_rw = nullptr;     // a shared record slab: the conversion writes it and k_accumulate gathers from it (instead of ws.d_recs)
  uint8_t* recs_out() const { return recs_rw ? recs_rw : ws.d_recs; }
  const uint32_t* fb_remap = nullptr;   // fixed-base windows: the digit rows were filled by k_fb_digits (no k_digits launch); the level-1 scatter maps positions through it
  int table_replicas = 1;         // experiment "exp_table_replicas": window k gathers from copy k / ceil(windows / copies) of the bound records
  const uint8_t* bound = nullptr; // records of a bound point set (te_msm_bind_points), already offset to this launch's first point: no conversion,
                                  // k_accumulate gathers from here instead of ws.d_recs (p.rec_kind tells which record form)
  // Own rows of a context that computes ALL windows can be written to host memory by the tail kernel (every row slot is
  // rewritten by every MSM).  Not with window shards (rows of foreign windows must read as zero: they come from the cleared
  // device block), not with captured graphs (fixed pointers), not with "prezero" = 0 (stage verifiers read the device rows).
  static bool rows_to_host(const te_ctx* ctx, const gpu_t& d, const plan_t& p, bool own_rows) {
    return own_rows && p.nw > 0 && p.batch == 1 && p.w_first == 0 && p.w_step == 1 && ctx->opt_prezero && !ctx->opt_graph;
  }
  uint32_t n32() const { return (uint32_t)n; }
  uint32_t total() const { return (uint32_t)p.nw * p.B; }
  uint32_t smax() const { return total() + (uint32_t)((uint64_t)p.nw * (n / p.seg_len)); }
  uint32_t chunk_cap() const { return total() + smax() / TE_GIANT_RUN + 2u; }     // entries of d_chunk_list (pairs), see ensure_buffers
  const void* points_of(int m) const { return p.batch > 1 ? static_cast<const void* const*>(d_points)[m] : d_points; }
  const void* scalars_of(int m) const { return p.batch > 1 ? static_cast<const void* const*>(d_scalars)[m] : d_scalars; }
  void mark(int i) const { if (prof >= 2 || (prof == 1 && (i == ST_ACCUM || i == ST_ACCUM + 1))) (void)hipEventRecord(ws.ev[i], stream); }

  // device-resident inputs, no per-stage timing: the record conversion rides in the launch of the sort's first level
  bool can_fuse_prep() const { return ctx->opt_fuse_prep && prof < 2 && p.nw > 0; }
  int front() {
    if (can_fuse_prep()) return front_scalars(true);
    if (int rc = front_scalars()) return rc;
    return front_points();
  }

  // the distinct point buffers of the sequence (grid rows of the conversion) and the record slab of each
  int prep_rows(te::batch_ptrs& tab, te::batch_slabs& row_slab) const {
    const te::batch_slabs sl = slabs();
    memset(&tab, 0, sizeof tab); memset(&row_slab, 0, sizeof row_slab);
    int rows = 0;
    for (int m = 0; m < p.batch; m++) if ((int)sl.s[m] == m) { tab.p[rows] = (const uint4*)points_of(m); row_slab.s[rows] = (uint32_t)m; rows++; }
    return rows;
  }

  // record slab of MSM m: MSMs of one call that name the same point buffer share one conversion (same pointer in one call =
  // same data; nothing is remembered across calls).  slab = index of the first MSM with that pointer.
  te::batch_slabs replica_slabs() const { te::batch_slabs r; for (int j = 0; j < TE_BATCH_MAX; j++) r.s[j] = (uint32_t)j; return r; }
  te::batch_slabs slabs() const {
    te::batch_slabs r; memset(&r, 0, sizeof r);
    for (int m = 0; m < p.batch; m++) {
      int first = m;
      for (int j = 0; j < m; j++) if (points_of(j) == points_of(m)) { first = j; break; }
      r.s[m] = (uint32_t)first;
    }
    return r;
  }

  // points -> records (needs only the points; runs last of the front part): one launch, one grid row per DISTINCT point buffer
  int front_points() {
    const uint32_t n32 = this->n32();
    mark(ST_PREP);
    te::batch_ptrs tab; te::batch_slabs row_slab;
    const int rows = prep_rows(tab, row_slab);
    if (p.curve == TE_MSM_CURVE_BLS12_377_G1)
      hipLaunchKernelGGL(te::k_prep_points377, dim3((n32 + 255) / 256, rows), dim3(256), 0, stream, tab, row_slab, reinterpret_cast<te::rec_slot<14>*>(recs_out()), n32);
    else
      hipLaunchKernelGGL(te::k_prep_points, dim3((n32 + 255) / 256, rows), dim3(256), 0, stream, tab, row_slab, reinterpret_cast<te::pnt_slot*>(recs_out()), n32);
    return 0;
  }

  // scalars -> digits, two-level counting sort, segment schedule (needs only the scalars); with_prep: the points -> records
  // conversion shares the launch of the sort's first level (k_part_scatter_prep)
  int front_scalars(bool with_prep = false) {
    const uint32_t n32 = this->n32();
    // flags, counters, histograms, bucket counts (a later piece of the same MSM keeps the final-carry flag and the entry count)
    // -- unless the block is still clean from the clearing that followed the set's previous MSM (finish_sequence)
    if (onto || ws.zero_clean_words < ws.zero_words)
      HIP_TRY(ctx, hipMemsetAsync(ws.d_zero + (onto ? Z_KEEP : 0), 0, (ws.zero_words - (onto ? Z_KEEP : 0)) * sizeof(uint32_t), stream));
    ws.zero_clean_words = 0;
    if (!p.fb_rb) mark(ST_DIGITS);                         // (fixed-base windows: enqueue_fixed_base marked it in front of k_fb_digits)
    te::sort_geom sg;
    sg.n = n32; sg.nst = p.nst; sg.B = p.B; sg.logS = p.logS; sg.S = p.S; sg.P = p.P; sg.CH = p.CH; sg.chunk_len = p.chunk_len; sg.half = p.signed_digits ? p.B : 0u; sg.packed = p.packed;
    if (p.nw > 0 && !p.fb_rb) {
      te::digits_params prm; memset(&prm, 0, sizeof prm);
      if (p.signed_digits) for (int w = 0; w < p.W; w++) { const int bit = w * p.c + p.c - 1; if (bit < 320) prm.half[bit >> 5] |= 1u << (bit & 31); }
      prm.zero_digit = p.signed_digits ? 1u << (p.c - 1) : 0u;
      prm.sc_stride = (uint32_t)(sizes_of(p.curve).scalar_in / 16);
      prm.n = n32; prm.nst = p.nst; prm.num_windows = p.W; prm.w_first = p.w_first; prm.w_step = p.w_step; prm.nw_local = p.nw1;
      prm.half_code = sg.half; prm.logS = p.logS; prm.P = p.P; prm.CH = p.CH; prm.chunk_len = p.chunk_len;
      // one launch over the MSMs of the sequence: MSM m (blockIdx.y) fills digit rows and level-1 counts [m * nw1, (m + 1) * nw1)
      te::batch_ptrs sc; memset(&sc, 0, sizeof sc);
      for (int m = 0; m < p.batch; m++) sc.p[m] = (const uint4*)scalars_of(m);
      uint16_t* dg = ws.d_digits;
      uint32_t* c1 = ws.d_counts1;
      switch (p.c) {
        case 4: launch_digits<4>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 5: launch_digits<5>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 6: launch_digits<6>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 7: launch_digits<7>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 8: launch_digits<8>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 9: launch_digits<9>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 10: launch_digits<10>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 11: launch_digits<11>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 12: launch_digits<12>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 13: launch_digits<13>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 14: launch_digits<14>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        case 15: launch_digits<15>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
        default: launch_digits<16>(sc, p.batch, dg, prm, ws.d_err, c1, stream); break;
      }
    }
    const uint32_t cap_w = p.B + (uint32_t)(n / p.seg_len);        // segment ids of one window (see k_part_scatter)
    mark(ST_SCATTER);
    if (p.nw > 0) {
      te::scatter_args sa;
      sa.digits = ws.d_digits; sa.counts1 = ws.d_counts1; sa.part_keys = ws.d_part_keys; sa.part_idx = ws.d_part_idx; sa.part_start = ws.d_part_start;
      sa.part_count = ws.d_part_count; sa.seg_part_base = ws.d_seg_part_base; sa.entries = reinterpret_cast<unsigned long long*>(ws.d_zero + Z_ENTRIES); sa.seg_len = p.seg_len; sa.cap_w = cap_w; sa.nw = (uint32_t)p.nw; sa.g = sg; sa.remap = fb_remap; sa.row_fill = ws.d_fb_fill;
      if (with_prep) {
        te::batch_ptrs tab; te::batch_slabs row_slab;
        const uint32_t rows = (uint32_t)prep_rows(tab, row_slab), sblocks = p.CH * (uint32_t)p.nw;
        if (bls()) {                                      // 512 points per conversion block, one per thread
          const uint32_t per_row = (n32 + 511u) / 512u;
          hipLaunchKernelGGL(te::k_part_scatter_prep377, dim3(sblocks + rows * per_row), dim3(512), 0, stream, sa, sblocks, tab, row_slab,
                             reinterpret_cast<te::rec_slot<14>*>(recs_out()), n32, per_row, rows * per_row);
        } else {
          const uint32_t per_row = (n32 + 255u) / 256u;
          hipLaunchKernelGGL(te::k_part_scatter_prep, dim3(sblocks + rows * per_row), dim3(512), 0, stream, sa, sblocks, tab, row_slab,
                             reinterpret_cast<te::pnt_slot*>(recs_out()), n32, per_row, rows * per_row);
        }
      } else {
        hipLaunchKernelGGL(te::k_part_scatter, dim3(p.CH, p.nw), dim3(512), 0, stream, sa);
      }
    }
    mark(ST_BSORT);
    if (p.nw > 0) {
      // level 2: one block per partition (+ extra blocks for the pieces of over-long partitions) counts, plans and places
      // it; d_num_seg[1..3] = split / giant bucket counters, zeroed with the rest
      const uint32_t l2_blocks = p.P + n32 / TE_L2_CAP + 1u;
      te::l2_args la;
      la.part_keys = ws.d_part_keys; la.part_idx = ws.d_part_idx; la.part_start = ws.d_part_start; la.part_count = ws.d_part_count;
      la.bucket_count = ws.d_bucket_count; la.sorted = ws.d_sorted; la.part_ticket = ws.d_part_ticket; la.g = sg;
      la.pa.part_start = ws.d_part_start; la.pa.part_count = ws.d_part_count; la.pa.seg_part_base = ws.d_seg_part_base;
      la.pa.bucket_start = ws.d_bucket_start; la.pa.bucket_cursor = ws.d_bucket_cursor; la.pa.seg_base = ws.d_seg_base; la.pa.seg_bucket = ws.d_seg_bucket;
      la.pa.seg_lenv = ws.d_seg_lenv; la.pa.size_hist = ws.d_size_hist; la.pa.split_list = ws.d_split_list; la.pa.split_count = ws.d_num_seg + 1;
      la.pa.chunk_list = ws.d_chunk_list; la.pa.seg_len = p.seg_len; la.pa.cap_w = cap_w; la.pa.chunk_cap = chunk_cap();
      if (p.packed) hipLaunchKernelGGL(te::k_l2_local<true>, dim3(l2_blocks, p.nw), dim3(256), 0, stream, la);
      else hipLaunchKernelGGL(te::k_l2_local<false>, dim3(l2_blocks, p.nw), dim3(256), 0, stream, la);
      // the segment schedule (counts the valid segments, d_num_seg[0]; with "sort_buckets" = 0 the schedule is simply not used)
      // + the placement of the pieces of over-long partitions in one launch
      te::order_args oa;
      oa.lenv = ws.d_seg_lenv; oa.ids = smax(); oa.size_hist = ws.d_size_hist; oa.rel_cursor = ws.d_size_cursor; oa.order = ws.d_order; oa.num_segments = ws.d_num_seg;
      // ~128 schedule blocks, more where a block's slice of the id space would not fit its registers (TE_ORDER_REGS * 256 ids)
      {
        const uint32_t by_regs = (smax() + TE_ORDER_REGS * 256u * (uint32_t)p.nw - 1u) / (TE_ORDER_REGS * 256u * (uint32_t)p.nw);
        oa.order_cols = std::min(64u, std::max(by_regs, (uint32_t)std::max(1, 128 / p.nw)));
      }
      oa.max_len = std::min(p.seg_len, 1023u);                     // no segment is longer: only the histogram chunks up to there are scanned
      if (p.packed)
        hipLaunchKernelGGL(te::k_l2_place_order<true>, dim3(oa.order_cols + l2_blocks, p.nw), dim3(256), 0, stream, ws.d_part_keys, ws.d_part_idx, ws.d_part_start,
                           ws.d_part_count, ws.d_bucket_cursor, ws.d_sorted, sg, oa);
      else
        hipLaunchKernelGGL(te::k_l2_place_order<false>, dim3(oa.order_cols + l2_blocks, p.nw), dim3(256), 0, stream, ws.d_part_keys, ws.d_part_idx, ws.d_part_start,
                           ws.d_part_count, ws.d_bucket_cursor, ws.d_sorted, sg, oa);
    }
    mark(ST_ORDER);
    return 0;
  }

  bool bls() const { return p.curve == TE_MSM_CURVE_BLS12_377_G1; }

  // K3: one thread per segment (at most seg_len entries of one bucket): 7-product mixed additions (8 for BLS12-377)
  template <int N, int RK> int accumulate_t() {
    mark(ST_ACCUM);
    if (p.nw > 0) {
      const uint32_t n32 = this->n32(), smax = this->smax();
      const uint32_t* order = ctx->opt_sort ? ws.d_order : nullptr;
      using slot_t = typename te::rec_kind<N, RK>::slot;
      static const bool serial = [] { const char* e = getenv("TE_MSM_SERIAL_ACCUMULATE"); return e && e[0] == '1'; }();
      std::unique_lock<std::mutex> chain(*d.acc_mu, std::defer_lock);
      if (serial && !ctx->opt_graph) {
        chain.lock();
        if (d.acc_prev && d.acc_prev_stream != stream) HIP_TRY(ctx, hipStreamWaitEvent(stream, d.acc_prev, 0));
      }
      hipLaunchKernelGGL((te::k_accumulate<N, RK>), dim3((smax + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const slot_t*>(bound ? bound : recs_out()), ws.d_sorted,
                         ws.d_bucket_start, ws.d_bucket_count, ws.d_seg_base, ws.d_seg_bucket, ws.d_seg_lenv, order, ws.d_num_seg,
                         reinterpret_cast<te::ete_t<N>*>(ws.d_buckets), reinterpret_cast<te::ete_t<N>*>(ws.d_seg_out), n32, p.logB, p.seg_len, smax, onto ? 1u : 0u,
                         table_replicas > 1 ? (uint32_t)((p.nw1 + table_replicas - 1) / table_replicas) : (uint32_t)p.nw1, table_replicas > 1 ? replica_slabs() : slabs(),
                         prof ? reinterpret_cast<unsigned long long*>(ws.d_zero + Z_CLOCK) : nullptr);
      if (chain.owns_lock()) {
        hipEvent_t& ev = d.acc_ev[d.acc_next++ % 16u];
        if (!ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(ev, stream));
        d.acc_prev = ev; d.acc_prev_stream = stream;
      }
    }
    return 0;
  }
  int accumulate() { return bls() ? (p.rec_kind == 1 ? accumulate_t<14, 1>() : accumulate_t<14, 0>()) : accumulate_t<9, 0>(); }

  // sums of the buckets that were accumulated in several parts: buckets cut into 2..16 parts (quads) and the TE_GIANT_RUN-part runs
  // of giant buckets (blocks; the block that finishes a bucket's last run adds the runs up) in one launch
  template <int N> int combine_t() {
    if (p.nw <= 0) return 0;
    using E = te::ete_t<N>;
    hipLaunchKernelGGL(te::k_seg_combine_all<N>, dim3(256 + 1024), dim3(256), 0, stream, ws.d_split_list, ws.d_num_seg + 1, ws.d_chunk_list,
                       ws.d_bucket_count, ws.d_seg_base, reinterpret_cast<E*>(ws.d_seg_out), reinterpret_cast<E*>(ws.d_buckets), p.seg_len, chunk_cap(), 256u,
                       ws.d_bucket_start, ws.d_bucket_cursor);
    return 0;
  }
  int combine() { return bls() ? combine_t<14>() : combine_t<9>(); }

  // K4/K5: buckets -> one row per window (see kernels.hip.hpp, K4a and k_reduce_tail).  Bucket index j = hi * L + lo:
  //   rows chain  xin[hi * r + g]  -- the low L = 2^(w0+w1) part folded from the top, 8 (4, 2) sub-blocks at a time
  //   cols chain  yin[h * L + lo]  -- the high H = 2^(w2+w3) part folded from the top
  // Wide levels (both chains per launch, one thread or one quad per output) run until at most 4 partial sums per output
  // are left; k_reduce_tail (one block per window and digit) does the rest and writes the row.  n = 2^20, c = 16: two fold
  // launches (32768 -> 4096 -> 512 points per window and chain) + the tail, against nine launches of the first version.
  template <int N> int reduce_t() {
    if (p.nw > 0) {
      using E = te::ete_t<N>;
      const uint32_t w0 = p.dw[0], w1 = p.dw[1], w2 = p.dw[2], w3 = p.dw[3];
      const uint32_t L = 1u << (w0 + w1), H = 1u << (w2 + w3);
      struct chain_t { const E* cur; uint32_t n, r; E* buf[2]; int pp; };
      E* const bk = reinterpret_cast<E*>(ws.d_buckets);
      chain_t ch[2] = {{bk, p.B, L, {reinterpret_cast<E*>(ws.d_red[0]), reinterpret_cast<E*>(ws.d_red[1])}, 0},
                       {bk, p.B, H, {reinterpret_cast<E*>(ws.d_red[2]), reinterpret_cast<E*>(ws.d_red[3])}, 0}};
      for (;;) {
        te::sum_jobs_t<N> js; memset(&js, 0, sizeof js);
        uint32_t most = 0; int nj = 0;
        for (int i = 0; i < 2; i++) {
          chain_t& c = ch[i];
          if (c.r <= 4u) continue;
          const uint32_t K = (c.r % 8u == 0) ? 8u : (c.r % 4u == 0) ? 4u : 2u;
          // (folding by 4 where that keeps a level above the 65536 outputs a thread-per-output launch needs -- n = 2^16, 2^17 --
          // was measured: 42 + 26 + 16 us against 62 + 16 us: one more dependent level costs more than the cheaper first one saves)
          te::sum_job_t<N>& j = js.j[nj++];
          j.in = c.cur; j.out = c.buf[c.pp]; j.K = K; j.n_out = c.n / K;
          j.inner = i == 0 ? c.r / K : (c.r / K) * L;          // rows: sub-blocks inside one hi; cols: whole slabs of L
          j.in_per_window = c.n; j.out_per_window = c.n / K;
          c.cur = j.out; c.pp ^= 1; c.r /= K; c.n /= K;
          most = std::max(most, j.n_out * (uint32_t)p.nw);
        }
        if (!nj) break;
        // From 32768 outputs on a level runs one thread per output (9 products per addition, k_sum_groups); below that four
        // lanes per output (16 lane-products per addition, but three dependent products instead of nine: k_sum_groups_team).
        // The line was at 65536 ("one wave per SIMD") until round 3: at n = 2^16..2^18 (18 windows x 16384 buckets, first level
        // 36864 outputs) the thread form is as fast for one MSM (0.344 against 0.346 ms) and leaves more of the VALU to the
        // other MSMs in flight: 0.1645 against 0.1785 ms per MSM at 2^16, 0.224 against 0.239 at 2^17.  16384 would also move
        // the second level of unsigned 16-bit windows (64 blocks of serial additions: slower for one MSM).
        if (most >= 65536u || (most >= 32768u && !ctx->opt_fold_pairs)) {
          uint32_t blocks = (most + 255) / 256; if (blocks > 4096) blocks = 4096;
          hipLaunchKernelGGL((te::k_sum_groups<N, false>), dim3(blocks, nj), dim3(256), 0, stream, js, (uint32_t)p.nw);
        } else if (most >= 32768u) {
          // too few outputs for one thread each to fill the machine (n <= 2^18: 36 864 per chain): two lanes per output, half
          // the dependent additions (option "fold_pairs"; every level folds by 8, 4 or 2 -- K is even)
          uint32_t blocks = (2 * most + 255) / 256; if (blocks > 4096) blocks = 4096;
          hipLaunchKernelGGL((te::k_sum_groups<N, true>), dim3(blocks, nj), dim3(256), 0, stream, js, (uint32_t)p.nw);
        } else {                            // latency-bound level: four lanes per output.  (Sixteen lanes per output as a tree --
                                            // 3 dependent team additions instead of 7 -- was measured: 22.5 us against 18.5 at
                                            // n = 2^20: four times the lanes put four waves on every SIMD and each addition slows down.)
          uint32_t blocks = (most * 4 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
          hipLaunchKernelGGL(te::k_sum_groups_team<N>, dim3(blocks, nj), dim3(256), 0, stream, js, (uint32_t)p.nw);
        }
      }
      mark(ST_WEIGHTED);
      te::tail_params_t<N> tp;
      tp.xin = ch[0].cur; tp.yin = ch[1].cur; tp.rx = ch[0].r; tp.ry = ch[1].r;
      tp.x_per_window = ch[0].n; tp.y_per_window = ch[1].n;
      tp.w[0] = w0; tp.w[1] = w1; tp.w[2] = w2; tp.w[3] = w3;
      // host_rows: the rows (and the flag words in front of them) are written to the pinned host block by the kernel itself
      uint8_t* const rows_base = host_rows ? reinterpret_cast<uint8_t*>(ws.h_err_dev + Z_ROWS) : static_cast<uint8_t*>(d_partials_out);
      tp.flag_src = host_rows ? ws.d_zero : nullptr; tp.flag_dst = host_rows ? ws.h_err_dev : nullptr; tp.flag_words = (uint32_t)Z_ROWS;
      tp.rows = reinterpret_cast<E*>(rows_base) + (size_t)p.w_first * 5; tp.row_stride = (uint32_t)p.w_step * 5u;
      tp.win_per_msm = (uint32_t)p.nw1; tp.msm_stride = (uint32_t)p.W * 5u;          // batch: MSM m's W rows follow MSM m-1's
      const size_t lds_bytes = (size_t)(std::max(H, L) + 16u) * sizeof(E);
      hipLaunchKernelGGL(te::k_reduce_tail<N>, dim3(4, p.nw), dim3(1024), lds_bytes, stream, tp);
    } else {
      mark(ST_WEIGHTED);
    }
    mark(ST_COUNT);
    if (!own_rows) HIP_TRY(ctx, hipMemcpyAsync(ws.h_err, ws.d_zero, Z_ROWS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));   // flag + clock words
    return 0;
  }
  int reduce() { return bls() ? reduce_t<14>() : reduce_t<9>(); }

  // recombination of split buckets, digit marginals, weighted sums, error flag read-back
  int back() {
    if (int rc = combine()) return rc;
    return reduce();
  }
};

// Captures fn (a sequence of launches on ws.stream) into an executable graph.
template <typename F> int capture_graph(te_ctx* ctx, workset_t& ws, hipGraphExec_t& exec, F&& fn) {
  if (exec) { (void)hipGraphExecDestroy(exec); exec = nullptr; }
  hipGraph_t graph = nullptr;
  HIP_TRY(ctx, hipStreamBeginCapture(ws.stream, hipStreamCaptureModeThreadLocal));
  const int rc = fn();
  const hipError_t e = hipStreamEndCapture(ws.stream, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  HIP_TRY(ctx, e);
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  HIP_TRY(ctx, ei);
  return 0;
}

// d_partials_out == nullptr: the rows go to the work set's own buffer (ws.d_partials, inside the block that is zeroed per
// MSM) and the caller fetches flag + rows with fetch_rows().
// upload_points (optional): enqueues the host-to-device copy of the points on the given (side) stream -- te_msm_run; the
// upload then overlaps digits, sort and schedule.
// With an upload, the points -> records conversion follows it on the work set's side stream, beside the scalar-only
// stages, and joins before the accumulation.  Running it there for device-resident inputs as well (side_stream) was
// measured and is not used: for one MSM the conversion and the sort stages are both bandwidth-bound and merely slow each
// other down (latency 1.36 -> 1.38 ms), and with several MSMs in flight every extra stream competes for the runtime's few
// hardware queues (four by default; GPU_MAX_HW_QUEUES=8 did not help) and serialises the others: 941 -> 862 MSM/s.
// the copy stream of a work set exists from its first host-buffer MSM on (te_msm_run): a context that only ever sees
// device-resident inputs owns one stream per work set
int need_copy_stream(te_ctx* ctx, workset_t& ws) {
  if (ws.copy_stream) return 0;
  // TE_MSM_COPY_PRIORITY=1 (experiment, tools/exp_bound_copy_queue.py): the runtime keeps streams of another priority on hardware
  // queues of their own, so an upload would never stand behind a kernel of a work set that shares its queue
  static const int prio = [] { const char* e = getenv("TE_MSM_COPY_PRIORITY"); return e ? atoi(e) : 0; }();
  if (prio) {
    int least = 0, greatest = 0;
    HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(ctx, hipStreamCreateWithPriority(&ws.copy_stream, hipStreamNonBlocking, prio > 0 ? greatest : least));
    return 0;
  }
  HIP_TRY(ctx, hipStreamCreateWithFlags(&ws.copy_stream, hipStreamNonBlocking));
  return 0;
}

// The runtime multiplexes all streams of a process onto a few hardware queues (four unless GPU_MAX_HW_QUEUES says otherwise)
// and kernels of one queue run in order: MSMs in flight on two streams of the same queue do not overlap.  Which stream
// gets which queue follows from how many streams the process created before (tools/queue_probe.hip: 0 1 2 3 3 2 1 0 3 2
// 1 0 ...), so a context created after other streams -- PyTorch's, RCCL's, another context's -- found its first four work
// sets on two queues: n = 2^16 / 2^17 / 2^18 ran at 0.23 / 0.32 / 0.42 instead of 0.19 / 0.24 / 0.35 ms per MSM.
// Only callers that keep several MSMs in flight care, so the measurement is LAZY: te_msm_init hands the streams out in
// creation order, and the first te_msm_submit_device (every work set idle, device synchronised) measures which of them
// share a queue (pairs of k_spin kernels: one duration when they overlap, two when they are serialised; ~16 ms once)
// and re-deals them so that sets 0..3 and sets 4..7 each sit on as many different queues as there are.  One-shot callers
// (te_msm_run, compute_msm with force_recompile) never pay it.  The host-timed pairs can be disturbed by other work on
// the GPU, so a classification is accepted only when a second, independent measurement gives the same classes;
// otherwise the creation order stays.  Option "queue_probe" = 0 (env TE_MSM_QUEUE_PROBE=0) turns the lazy measurement off;
// te_msm_probe_queues runs it at a moment the caller chooses.
// classes[i] = index of the hardware queue class of streams[i] (classes numbered by first appearance); returns the number of
// classes, or -1 when the measurement could not run (then classes[] is all -1)
int classify_streams_by_queue(gpu_t& d, const hipStream_t* streams, int n, int* cls) {
  for (int i = 0; i < n; i++) cls[i] = -1;
  if (d.wall_clock_khz <= 0 || n < 2) return -1;
  uint32_t* const flag = nullptr;       // k_spin takes an optional output word; none here (hipMalloc / hipFree would synchronise the whole device)
  const unsigned long long ticks = (unsigned long long)d.wall_clock_khz * 3 / 10;      // 0.3 ms
  auto pair_ms = [&](int a, int b) {
    (void)hipStreamSynchronize(streams[a]); (void)hipStreamSynchronize(streams[b]);
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(te::k_spin, dim3(1), dim3(64), 0, streams[a], ticks, flag);
    hipLaunchKernelGGL(te::k_spin, dim3(1), dim3(64), 0, streams[b], ticks, flag);
    (void)hipStreamSynchronize(streams[a]); (void)hipStreamSynchronize(streams[b]);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  (void)pair_ms(0, 1);                                                                    // first launch: code upload
  // one kernel alone (the same stream twice runs them back to back: half of that), best of two
  const double alone = std::min(pair_ms(0, 0), pair_ms(0, 0)) * 0.5;
  auto shared = [&](int a, int b) { return std::min(pair_ms(a, b), pair_ms(a, b)) > 1.6 * alone; };
  int ncls = 0;
  for (int i = 0; i < n; i++) {
    if (cls[i] >= 0) continue;
    cls[i] = ncls;
    for (int j = i + 1; j < n; j++) if (cls[j] < 0 && shared(i, j)) cls[j] = ncls;
    ncls++;
  }
  if (hipGetLastError() != hipSuccess) { for (int i = 0; i < n; i++) cls[i] = -1; return -1; }
  return ncls;
}

// order[k] = which stream comes k-th when the streams are dealt round-robin over their classes: first one stream of every
// class, then the next of every class, ... (identity when nothing was measured)
void deal_over_classes(const int* cls, int n, int ncls, int* order) {
  int k = 0;
  if (ncls > 1 && ncls < n) {
    std::vector<char> used(n, 0);
    while (k < n)
      for (int c = 0; c < ncls && k < n; c++)
        for (int i = 0; i < n; i++) if (!used[i] && cls[i] == c) { used[i] = 1; order[k++] = i; break; }
  } else {
    for (int i = 0; i < n; i++) order[i] = i;
  }
}

// EXPORTED STREAMS ARE NEVER DESTROYED.  te_msm_workset_stream hands a work set's stream out as a raw hipStream_t so that a
// caller can order its own work behind an MSM (PyTorch: torch.cuda.ExternalStream).  Such callers REMEMBER the handle in places
// the engine cannot see: PyTorch's pinned-memory allocator keeps, for every pinned block, the streams it was used on and records
// an event on each of them when the block is released -- for a tensor that outlives the context that is after te_msm_destroy,
// possibly at interpreter exit; hipEventRecord on a destroyed stream fails, the error is raised inside a deleter, and the
// process aborts (round 5: tools/exp_batch_small.py "dumped core" after its last result line; profiles/r06_batch_small_abort.txt).
// So the streams of a device whose handles were exported are PARKED by te_msm_destroy -- synchronised, kept alive, and taken
// over by the next context on that device -- and stay valid hipStream_t values until the process exits.  A context that never
// exported a handle destroys its streams as before.  (env TE_MSM_PARK_STREAMS=0: the old behaviour, for the diagnosis only.)
struct parked_streams_t { std::mutex mu; std::vector<std::pair<int, hipStream_t>> v; };
parked_streams_t& parked_streams() { static parked_streams_t* p = new parked_streams_t(); return *p; }     // (leaked on purpose: no destructor at exit)
bool park_exported_streams() { static const bool on = [] { const char* e = getenv("TE_MSM_PARK_STREAMS"); return !(e && e[0] == '0'); }(); return on; }
hipStream_t take_parked_stream(int device) {
  parked_streams_t& ps = parked_streams();
  std::lock_guard<std::mutex> lk(ps.mu);
  for (size_t i = 0; i < ps.v.size(); i++)
    if (ps.v[i].first == device) { hipStream_t s = ps.v[i].second; ps.v.erase(ps.v.begin() + (long)i); return s; }
  return nullptr;
}
void park_stream(int device, hipStream_t s) {
  parked_streams_t& ps = parked_streams();
  std::lock_guard<std::mutex> lk(ps.mu);
  ps.v.emplace_back(device, s);
}

// te_msm_init: one compute stream per work set, in creation order (no measurement); parked streams of the device first
int create_workset_streams(gpu_t& d) {
  for (int i = 0; i < TE_MSM_WORKSETS; i++) {
    d.ws[i].stream = take_parked_stream(d.device);
    if (d.ws[i].stream) d.streams_exported = true;      // somebody may still hold this handle: it goes back to the pool, never to hipStreamDestroy
    if (!d.ws[i].stream && hipStreamCreateWithFlags(&d.ws[i].stream, hipStreamNonBlocking) != hipSuccess) {
      for (int j = 0; j < i; j++) { if (d.streams_exported) park_stream(d.device, d.ws[j].stream); else (void)hipStreamDestroy(d.ws[j].stream); d.ws[j].stream = nullptr; }
      d.ws[i].stream = nullptr;
      return -1;
    }
    d.ws[i].hw_queue_class = -1;
  }
  return 0;
}

// The measurement itself (first te_msm_submit* of a context unless option "queue_probe" = 0, or te_msm_probe_queues at a moment
// the caller chooses).  Never fatal: any failure leaves the creation order in place.  It waits for THIS context's streams only
// -- no device-wide synchronisation: other streams of the process (PyTorch's, RCCL's) keep running; should their kernels
// disturb the host-timed pairs, the two passes disagree and the creation order stays.
void spread_streams_over_queues(gpu_t& d) {
  d.queues_probed = true;
  for (const workset_t& ws : d.ws) if (te_sched::slot_ticket(ws.slot)) return;            // cannot happen on the first submit; be safe
  for (workset_t& ws : d.ws) {
    if (ws.used && ws.ev_done && hipEventSynchronize(ws.ev_done) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipStreamSynchronize(ws.stream) != hipSuccess) { (void)hipGetLastError(); return; }
  }
  hipStream_t cand[TE_MSM_WORKSETS];
  for (int i = 0; i < TE_MSM_WORKSETS; i++) cand[i] = d.ws[i].stream;
  int cls[TE_MSM_WORKSETS], again[TE_MSM_WORKSETS], order[TE_MSM_WORKSETS];
  const int ncls = classify_streams_by_queue(d, cand, TE_MSM_WORKSETS, cls);
  const int ncls2 = ncls > 0 ? classify_streams_by_queue(d, cand, TE_MSM_WORKSETS, again) : -1;
  const bool agree = ncls > 0 && ncls2 == ncls && memcmp(cls, again, sizeof cls) == 0;
  if (getenv("TE_MSM_QUEUE_DUMP")) {
    fprintf(stderr, "[te_msm] hardware-queue classes of %d streams (%s):", TE_MSM_WORKSETS, agree ? "two measurements agree" : "measurements disagree: creation order kept");
    for (int i = 0; i < TE_MSM_WORKSETS; i++) fprintf(stderr, " %d/%d", cls[i], ncls2 > 0 ? again[i] : -1);
    fprintf(stderr, "\n");
  }
  if (!agree) return;
  deal_over_classes(cls, TE_MSM_WORKSETS, ncls, order);
  for (int i = 0; i < TE_MSM_WORKSETS; i++) {
    workset_t& ws = d.ws[i];
    // the set's buffers were last used on its old stream: the context's work is over (synchronised above), so the new stream
    // needs no wait; last_stream follows so that enqueue_partial does not add one
    if (ws.last_stream == ws.stream) ws.last_stream = cand[order[i]];
    ws.stream = cand[order[i]]; ws.hw_queue_class = cls[order[i]];
  }
}

int finish_sequence(te_ctx* ctx, workset_t& ws, hipStream_t stream);

// SHARED RECORD SLABS.  Every work set used to own the record slab its MSM converts into: four MSMs in flight gather from
// 4 x 128 MB of records (n = 2^20) -- twice the 256 MB Infinity Cache -- although a caller with MSMs in flight nearly always names ONE
// point buffer (the harness: full_benchmarks.ts:63-68,100-105; a prover: its SRS).  A gather footprint beyond the cache costs clock under
// the power ceiling (profiles/r06_fixed_base_windows.txt, step 1: 512 MB instead of 128 MB: -7 % MSMs in flight).  So whole-MSM calls
// from device-resident inputs that name the same point buffer (pointer, n, curve) while they are in flight share one slab: EVERY call
// still converts its points -- nothing is remembered across calls: the buffer's contents may change between calls, and a call's own
// accumulation is ordered behind its own conversion on its stream -- but all of them write the same bytes to the same place and gather
// from there (inputs of a call in flight must not change: include/te_msm.h).  The reference converts per call as well
// (convert_point_coords...wgsl:37-77).  A slab serves another point buffer only when no call in flight uses it any more.
// (the slab's earlier users may still be running -- see rec_slab_t::pending: `stream`, on which the new user's conversion is about to be
// enqueued, waits for them; nullptr: the host waits -- trim, destroy)
int settle_slab(te_ctx* ctx, gpu_t::rec_slab_t& sl, hipStream_t stream, bool host) {
  for (int wi = 0; wi < TE_MSM_WORKSETS; wi++) if ((sl.pending >> wi) & 1u) {
    if (host) HIP_TRY(ctx, hipEventSynchronize(sl.ev[wi])); else HIP_TRY(ctx, hipStreamWaitEvent(stream, sl.ev[wi], 0));
  }
  sl.pending = 0;
  return 0;
}
int acquire_shared_recs(te_ctx* ctx, gpu_t& d, const void* src, uint64_t n, int curve, hipStream_t stream, uint8_t** out) {
  const size_t need = (size_t)n * sizes_of(curve).rec;
  int pick = -1;
  // the same buffer, in use or just let go of (its records are the same bytes: no wait, and `pending` stays for whoever re-purposes the slab)
  for (size_t i = 0; i < d.slabs.size(); i++) if ((d.slabs[i].users > 0 || d.slabs[i].pending) && d.slabs[i].d && d.slabs[i].src == src && d.slabs[i].n == n && d.slabs[i].curve == curve) {
    d.slabs[i].users++; *out = d.slabs[i].d; return (int)i;
  }
  // a free slab: one nobody is waiting on first, the smallest that is large enough
  if (pick < 0) for (size_t i = 0; i < d.slabs.size(); i++) if (d.slabs[i].users == 0 && !d.slabs[i].pending && d.slabs[i].d && d.slabs[i].cap >= need && (pick < 0 || d.slabs[i].cap < d.slabs[(size_t)pick].cap)) pick = (int)i;
  if (pick < 0) for (size_t i = 0; i < d.slabs.size(); i++) if (d.slabs[i].users == 0 && d.slabs[i].d && d.slabs[i].cap >= need && (pick < 0 || d.slabs[i].cap < d.slabs[(size_t)pick].cap)) pick = (int)i;
  if (pick < 0) for (size_t i = 0; i < d.slabs.size(); i++) if (d.slabs[i].users == 0) { pick = (int)i; break; }
  if (pick < 0) { d.slabs.emplace_back(); pick = (int)d.slabs.size() - 1; }
  gpu_t::rec_slab_t& sl = d.slabs[(size_t)pick];
  if (!sl.d || sl.cap < need) {
    if (int rc = settle_slab(ctx, sl, nullptr, true)) return rc;
    if (sl.d) HIP_TRY(ctx, hipFree(sl.d));
    sl.d = nullptr; sl.cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)&sl.d, need ? need : 16));
    sl.cap = need;
  }
  if (int rc = settle_slab(ctx, sl, stream, false)) return rc;          // another buffer's records are about to be overwritten
  sl.src = src; sl.n = n; sl.curve = curve; sl.users++;
  *out = sl.d;
  return pick;
}
// completed: the set's MSM is known to be over (a collected ticket, a synchronous call).  Otherwise (the building blocks: the set is simply
// used again) an event behind that MSM -- on the stream it ran on, BEFORE anything new is enqueued there -- guards the slab.
void release_shared_recs(gpu_t& d, workset_t& ws, bool completed = true) {
  if (ws.slab >= 0 && (size_t)ws.slab < d.slabs.size() && d.slabs[(size_t)ws.slab].users > 0) {
    gpu_t::rec_slab_t& sl = d.slabs[(size_t)ws.slab];
    sl.users--;
    if (!completed && ws.last_stream) {
      const int wi = (int)(&ws - d.ws);
      if (!sl.ev[wi]) (void)hipEventCreateWithFlags(&sl.ev[wi], hipEventDisableTiming);
      if (sl.ev[wi] && hipEventRecord(sl.ev[wi], ws.last_stream) == hipSuccess) sl.pending |= 1u << wi;
      else (void)hipStreamSynchronize(ws.last_stream);                  // (no event: the host waits instead)
    }
  }
  ws.slab = -1;
}

// bases: the launch sequence gathers from a bound point set (d_points is not read: no conversion); batch must be 1
// share_recs: a whole-MSM call whose completion the engine sees (a ticket, a synchronous call): its records may live in a shared slab
int enqueue_partial(te_ctx* ctx, gpu_t& d, workset_t& ws, const void* d_points, const void* d_scalars, uint64_t n,
                    void* d_partials_out, hipStream_t stream, const std::function<int(hipStream_t)>* upload_points = nullptr, int force_c = 0,
                    bool side_stream = false, int batch = 1, bool whole = false, const te_bases* bases = nullptr, bool share_recs = false) {
  plan_t p; make_plan(ctx, d, n, p, force_c, batch, 0, whole);
  if (bases) p.rec_kind = bases->rec_kind;
  // (a batch shares when all its MSMs name ONE point buffer: the batch then holds one conversion anyway -- slab 0 of msm_launch::slabs())
  const void* share_src = d_points;
  if (batch > 1 && d_points) { share_src = static_cast<const void* const*>(d_points)[0]; for (int m = 1; m < batch; m++) if (static_cast<const void* const*>(d_points)[m] != share_src) share_src = nullptr; }
  share_recs = share_recs && ctx->opt_share_records && !bases && !upload_points && !ctx->opt_graph && share_src != nullptr;
  if (batch > 1 && (uint64_t)p.nw * p.nst >= (1ull << 31)) return set_err(ctx, TE_MSM_EINVAL, "batch too large for this n: windows x points must stay below 2^31");
  HIP_TRY(ctx, hipSetDevice(d.device));
  if ((uint64_t)p.nw * p.B + (uint64_t)p.nw * (n / p.seg_len) + 1024u >= (1ull << 32))
    return set_err(ctx, TE_MSM_EINVAL, "segment_len is too small for this n: more than 2^32 segments");
  if (int rc = ensure_buffers(ctx, d, ws, n, p, bases == nullptr && !share_recs)) return rc;
  release_shared_recs(d, ws, false);                        // (a slab the set's previous MSM still named: guarded by an event unless that MSM was seen to end)
  uint8_t* shared = nullptr;
  if (share_recs) { const int si = acquire_shared_recs(ctx, d, share_src, n, p.curve, stream, &shared); if (si < 0) return si; ws.slab = si; }
  if (ws.used && ws.last_stream != stream) HIP_TRY(ctx, hipStreamWaitEvent(stream, ws.ev_done, 0));   // the set's buffers are still the previous MSM's
  ws.plan = p; ws.n = n; ws.used = true; ws.last_stream = stream; __atomic_store_n(&d.last_ws, (int)(&ws - d.ws), __ATOMIC_RELAXED);
  ws.prof_level = ctx->opt_profile;
  const bool own_rows = d_partials_out == nullptr;
  if (own_rows) d_partials_out = ws.d_partials;
  // profile 1: two events around the dominant kernel only (what bench.py times live); 2: every stage boundary
  // (an event between two kernels costs ~4 us of idle stream time, 11 of them ~2 % of a 2^20 MSM)
  msm_launch L{ctx, d, ws, p, d_points, d_scalars, n, d_partials_out, ctx->opt_profile, stream, own_rows};
  L.host_rows = msm_launch::rows_to_host(ctx, d, p, own_rows);
  ws.rows_on_host = L.host_rows;
  L.recs_rw = shared;
  ws.recs_last = bases ? nullptr : (shared ? shared : ws.d_recs);
  if (bases) {
    // resident bases: the scalar-only stages, then the accumulation straight from the bound records (never captured: option "graph"
    // holds the pointers of device-resident point buffers)
    L.bound = bases->recs[(size_t)(&d - ctx->devs.data())];
    L.table_replicas = bases->replicas;
    if (int rc = L.front_scalars()) return rc;
    L.mark(ST_PREP);
    if (int rc = L.accumulate()) return rc;
    L.mark(ST_TREE);
    if (int rc = L.back()) return rc;
  } else if (ctx->opt_graph && ctx->opt_profile < 2 && !upload_points && batch == 1) {
    // the graphs hold pointers and geometry: re-captured when any of them changes (including a buffer reallocation);
    // the captured front always clears the zeroed block itself
    ws.zero_clean_words = 0;
    graph_key key; memset(&key, 0, sizeof key);          // padding bytes take part in the memcmp below
    key.pts = d_points; key.sc = d_scalars; key.out = d_partials_out; key.n = n; key.generation = ws.generation;
    key.c = p.c; key.w_first = p.w_first; key.w_step = p.w_step; key.seg_len = (int)p.seg_len;
    key.sort = ctx->opt_sort | (ctx->opt_signed << 1) | (ctx->opt_curve << 2) | (ctx->opt_packed << 4) | (ctx->opt_fold_pairs << 5);
    if (!ws.g_front || !ws.g_back || memcmp(&key, &ws.g_key, sizeof key) != 0) {
      if (ws.g_front || ws.g_back) HIP_TRY(ctx, hipEventSynchronize(ws.ev_done));   // a previous replay may still be running
      msm_launch C = L; C.stream = ws.stream; C.prof = 0;
      if (int rc = capture_graph(ctx, ws, ws.g_front, [&] { return C.front(); })) return rc;
      if (int rc = capture_graph(ctx, ws, ws.g_back, [&] { return C.back(); })) return rc;
      ws.g_key = key;
    }
    HIP_TRY(ctx, hipGraphLaunch(ws.g_front, stream));
    if (int rc = L.accumulate()) return rc;
    L.mark(ST_TREE);
    HIP_TRY(ctx, hipGraphLaunch(ws.g_back, stream));
  } else if (ctx->opt_profile >= 2 || !(side_stream || upload_points)) {
    if (!upload_points && L.can_fuse_prep()) {
      if (int rc = L.front()) return rc;
    } else {
      if (int rc = L.front_scalars()) return rc;
      if (upload_points) { if (int rc = (*upload_points)(stream)) return rc; }
      if (int rc = L.front_points()) return rc;
    }
    if (int rc = L.accumulate()) return rc;
    L.mark(ST_TREE);
    if (int rc = L.back()) return rc;
  } else {
    // side stream: [upload of the points] -> records; it starts behind everything already enqueued on `stream`
    HIP_TRY(ctx, hipEventRecord(ws.ev_start, stream));
    if (int rc = need_copy_stream(ctx, ws)) return rc;
    HIP_TRY(ctx, hipStreamWaitEvent(ws.copy_stream, ws.ev_start, 0));
    if (upload_points) { if (int rc = (*upload_points)(ws.copy_stream)) return rc; }
    msm_launch S = L; S.stream = ws.copy_stream;
    if (int rc = S.front_points()) return rc;
    HIP_TRY(ctx, hipEventRecord(ws.ev_copy, ws.copy_stream));
    if (int rc = L.front_scalars()) return rc;
    HIP_TRY(ctx, hipStreamWaitEvent(stream, ws.ev_copy, 0));
    if (int rc = L.accumulate()) return rc;
    L.mark(ST_TREE);
    if (int rc = L.back()) return rc;
  }
  // rows in the caller's buffer: the sequence ends here (the flag's read-back is part of reduce()); rows in the set's own
  // block: the caller's fetch_rows() ends it
  if (!own_rows) { if (int rc = finish_sequence(ctx, ws, stream)) return rc; }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// one device-to-host copy: final-carry flag + the W rows of the work set's own row buffer (enqueue_partial with nullptr);
// ends the launch sequence
int fetch_rows(te_ctx* ctx, workset_t& ws, hipStream_t stream) {
  // (nothing to copy when k_reduce_tail wrote flag words and rows to the pinned block itself)
  if (!ws.rows_on_host)
    HIP_TRY(ctx, hipMemcpyAsync(ws.h_err, ws.d_zero, Z_ROWS * 4 + (size_t)ws.plan.W * sizes_of(ws.plan.curve).row, hipMemcpyDeviceToHost, stream));
  return finish_sequence(ctx, ws, stream);
}

// End of an MSM's launch sequence on `stream`: flag and rows are on their way to the host (ev_result); behind them the
// zeroed block is cleared for the set's NEXT MSM -- 4 us of fill and a launch gap that would otherwise sit in front of that
// MSM's first kernel, on its critical path -- and ev_done marks the set free.  Not with captured graphs (the fill is part of
// the captured front) and not with "prezero" = 0 (stage verifiers read counters and rows from the block afterwards).
int finish_sequence(te_ctx* ctx, workset_t& ws, hipStream_t stream) {
  HIP_TRY(ctx, hipEventRecord(ws.ev_result, stream));
  if (ctx->opt_prezero && !ctx->opt_graph) {
    HIP_TRY(ctx, hipMemsetAsync(ws.d_zero, 0, ws.zero_words * sizeof(uint32_t), stream));
    ws.zero_clean_words = ws.zero_words;
  }
  HIP_TRY(ctx, hipEventRecord(ws.ev_done, stream));
  return 0;
}

// Stage times of the MSM that last ran on `ws`, read from the events recorded at ITS profile level (the option may have
// changed since).  Never fatal: a failed read leaves "no stage times" instead of losing the MSM's result.
int collect_stage_ms(te_ctx* ctx, const gpu_t& d, workset_t& ws) {
  if (!ws.prof_level) return 0;
  bool ok = true;
  for (int i = 0; i < ST_COUNT; i++) {
    float ms = -1.0f;
    if (ws.prof_level >= 2 || i == ST_ACCUM) ok = ok && hipEventElapsedTime(&ms, ws.ev[i], ws.ev[i + 1]) == hipSuccess;
    ctx->stage_ms[i] = ms;
  }
  {
    // k_accumulate stamped ~clock of its first wave and the clock of its last one (atomic max on zeroed words); they came
    // back with the flag.  Unlike the event interval this excludes the time the launch waited behind other streams' kernels.
    uint64_t raw[4 * TE_CLK_SLOTS]; memcpy(raw, ws.h_err + Z_CLOCK, sizeof raw);
    uint64_t c[4] = {0, 0, 0, 0};                      // max ~start, max end, sum of core ticks, sum of wall ticks over the copies
    for (uint32_t i = 0; i < TE_CLK_SLOTS; i++) {
      c[0] = std::max(c[0], raw[i]); c[1] = std::max(c[1], raw[TE_CLK_SLOTS + i]);
      c[2] += raw[2 * TE_CLK_SLOTS + i]; c[3] += raw[3 * TE_CLK_SLOTS + i];
    }
    const int khz = d.wall_clock_khz;
    const bool have = c[0] && c[1] && khz > 0 && c[1] > ~c[0];
    const double ms = have ? (double)(c[1] - ~c[0]) / khz : -1.0;
    ctx->stage_ms[ST_COUNT] = (float)ms;
    // c[2] / c[3]: core ticks / wall ticks summed over the kernel's waves; wall ticks run at khz
    ctx->stage_ms[ST_COUNT + 1] = (have && c[2] && c[3]) ? (float)((double)c[2] / (double)c[3] * khz * 1e-6) : -1.0f;
  }
  if (!ok) (void)hipGetLastError();
  ctx->have_stage_ms = ok;
  return 0;
}

void free_workset_buffers(workset_t& ws) {      // the big device buffers of a work set (te_msm_trim, free_dev); streams, events and the pinned block stay
  void** ptrs[] = {(void**)&ws.d_recs, (void**)&ws.d_digits, (void**)&ws.d_part_keys, (void**)&ws.d_zero, (void**)&ws.d_part_start, (void**)&ws.d_part_count,
                   (void**)&ws.d_part_idx, (void**)&ws.d_seg_part_base, (void**)&ws.d_bucket_start, (void**)&ws.d_bucket_cursor, (void**)&ws.d_sorted,
                   (void**)&ws.d_seg_base, (void**)&ws.d_seg_bucket, (void**)&ws.d_seg_lenv, (void**)&ws.d_order, (void**)&ws.d_split_list,
                   (void**)&ws.d_chunk_list, (void**)&ws.d_seg_out, (void**)&ws.d_buckets, (void**)&ws.d_red[0], (void**)&ws.d_red[1],
                   (void**)&ws.d_red[2], (void**)&ws.d_red[3], &ws.d_in_points, &ws.d_in_scalars, (void**)&ws.d_fb_remap};
  for (void** q : ptrs) if (*q) { (void)hipFree(*q); *q = nullptr; }
  if (ws.h_ring) { (void)hipHostFree(ws.h_ring); ws.h_ring = nullptr; }       // the pinned ring of option "host_staging" (its events stay)
  memset(ws.cap, 0, sizeof ws.cap); ws.cap_in_points = ws.cap_in_scalars = 0;
  ws.zero_words = ws.zero_clean_words = 0;
  ws.d_err = ws.d_num_seg = ws.d_size_hist = ws.d_size_cursor = ws.d_counts1 = ws.d_bucket_count = ws.d_part_ticket = nullptr; ws.d_partials = nullptr;
  if (ws.g_front) { (void)hipGraphExecDestroy(ws.g_front); ws.g_front = nullptr; }
  if (ws.g_back) { (void)hipGraphExecDestroy(ws.g_back); ws.g_back = nullptr; }
  ws.generation++; ws.used = false;
}

void free_dev(gpu_t& d) {
  (void)hipSetDevice(d.device);
  for (auto& sl : d.slabs) { if (sl.d) { (void)hipFree(sl.d); sl.d = nullptr; } for (hipEvent_t& e : sl.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; } }
  d.slabs.clear();
  for (hipEvent_t& e : d.acc_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
  d.acc_prev = nullptr; d.acc_prev_stream = nullptr;
  for (workset_t& ws : d.ws) {
    free_workset_buffers(ws);
    if (ws.h_err) (void)hipHostFree(ws.h_err);
    for (hipEvent_t e : ws.ring_ev) if (e) (void)hipEventDestroy(e);
    if (ws.ev_done) (void)hipEventDestroy(ws.ev_done);
    if (ws.ev_result) (void)hipEventDestroy(ws.ev_result);
    for (auto& ev : ws.ev) if (ev) (void)hipEventDestroy(ev);
    if (ws.ev_copy) (void)hipEventDestroy(ws.ev_copy);
    if (ws.ev_start) (void)hipEventDestroy(ws.ev_start);
    for (hipEvent_t e : ws.piece_events) (void)hipEventDestroy(e);
    if (ws.copy_stream) (void)hipStreamDestroy(ws.copy_stream);
    // a stream whose handle left the library stays alive (see "EXPORTED STREAMS ARE NEVER DESTROYED"); te_msm_destroy has synchronised it
    if (ws.stream) { if (d.streams_exported && park_exported_streams()) park_stream(d.device, ws.stream); else (void)hipStreamDestroy(ws.stream); }
    ws.stream = nullptr; ws.copy_stream = nullptr;
  }
}

// device copies of a host-buffer call's inputs, in bytes of the call's curve (64 + 32 per point for the Twisted-Edwards
// wire format, 96 + 48 for BLS12-377)
int ensure_staging(te_ctx* ctx, workset_t& ws, size_t bytes_points, size_t bytes_scalars) {
  if (bytes_points > ws.cap_in_points) {
    if (ws.d_in_points) HIP_TRY(ctx, hipFree(ws.d_in_points));
    ws.d_in_points = nullptr; ws.cap_in_points = 0;
    HIP_TRY(ctx, hipMalloc(&ws.d_in_points, bytes_points));
    ws.cap_in_points = bytes_points;
  }
  if (bytes_scalars > ws.cap_in_scalars) {
    if (ws.d_in_scalars) HIP_TRY(ctx, hipFree(ws.d_in_scalars));
    ws.d_in_scalars = nullptr; ws.cap_in_scalars = 0;
    HIP_TRY(ctx, hipMalloc(&ws.d_in_scalars, bytes_scalars));
    ws.cap_in_scalars = bytes_scalars;
  }
  return 0;
}

// Is this host address pinned (hipHostMalloc) or registered (hipHostRegister) memory?  A copy from such memory does not
// stage: hipMemcpyAsync returns before the data has left it.  Ordinary (pageable) memory is unknown to the runtime: the query
// fails with hipErrorInvalidValue or reports hipMemoryTypeUnregistered, depending on the ROCm version.
bool host_memory_is_pinned(const void* p) {
  hipPointerAttribute_t at; memset(&at, 0, sizeof at);
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

// ---- option "host_staging" ------------------------------------------------------------------------------------------------
// A copy from pageable memory is fast only while the runtime still holds the caller's pages registered from an earlier copy
// of the SAME range: te_msm_run of 2^20 points takes 2.35 ms from buffers it has seen, 4.7-5.0 ms at the first touch of a
// buffer, and 4.4-27 ms when every call brings freshly allocated buffers (registration and the release of the previous
// buffers' registration sit on the call's critical path; eight threads touching new ranges at once serialise in it: round
// 4's "9.5 ms median" of the eight-device call -- profiles/r05_host_buffers_first_touch.txt, r05_point_shard_stamps_D8.txt).
// A prover that allocates its buffers per call never sees the fast case.  With "host_staging" = 1 the engine does not depend
// on the history of the caller's memory: the buffers are copied, 2 MB at a time, by a small crew of host threads into a pinned
// ring of the work set and travel from there; the call returns when the last chunk has left the caller's buffer.
constexpr size_t TE_RING_SLOT = 2u << 20;      // bytes per slot: 48 DMA calls per 96 MB, ~0.2 ms of enqueue cost
constexpr int TE_RING_SLOTS = 16;              // 32 MB pinned per work set that stages
constexpr int TE_STAGERS_MAX = 32;
// crew size: each thread copies ~5 GB/s into the ring while the DMA engine reads behind it (env TE_MSM_STAGERS, measurements:
// profiles/r05_host_staging_copy_ab.txt)
int stager_count() {
  static const int n = [] { const char* e = getenv("TE_MSM_STAGERS"); int v = e ? atoi(e) : 8; return v < 1 ? 1 : v > TE_STAGERS_MAX ? TE_STAGERS_MAX : v; }();
  return n;
}
// chunk copy of the crew: the destination (a ring slot) is read next by the DMA engine, never by a core -- streaming stores keep it
// out of the caches and skip the read-for-ownership of every destination line (env TE_MSM_STAGING_COPY=memcpy: plain memcpy, A/B)
#if defined(__x86_64__)
__attribute__((target("avx2"))) void copy_streaming_avx2(uint8_t* to, const uint8_t* from, size_t len) {
  size_t i = 0;
  if ((reinterpret_cast<uintptr_t>(to) & 31u) == 0) {
    for (; i + 128 <= len; i += 128) {
      const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(from + i)), b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(from + i + 32));
      const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(from + i + 64)), d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(from + i + 96));
      _mm256_stream_si256(reinterpret_cast<__m256i*>(to + i), a); _mm256_stream_si256(reinterpret_cast<__m256i*>(to + i + 32), b);
      _mm256_stream_si256(reinterpret_cast<__m256i*>(to + i + 64), c); _mm256_stream_si256(reinterpret_cast<__m256i*>(to + i + 96), d);
    }
    _mm_sfence();
  }
  if (i < len) memcpy(to + i, from + i, len - i);
}
#endif
void staging_chunk_copy(uint8_t* to, const uint8_t* from, size_t len) {
#if defined(__x86_64__)
  static const bool streaming = [] { const char* e = getenv("TE_MSM_STAGING_COPY"); return !(e && e[0] == 'm') && __builtin_cpu_supports("avx2"); }();
  if (streaming) { copy_streaming_avx2(to, from, len); return; }
#endif
  memcpy(to, from, len);
}
int ensure_ring(te_ctx* ctx, workset_t& ws) {
  if (ws.h_ring) return 0;
  HIP_TRY(ctx, hipHostMalloc((void**)&ws.h_ring, TE_RING_SLOT * TE_RING_SLOTS, hipHostMallocDefault));
  if (ws.ring_ev.empty()) {
    ws.ring_ev.resize(TE_RING_SLOTS, nullptr);
    for (auto& e : ws.ring_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  ws.ring_next = 0;
  return 0;
}
// the crew is shared by everything of the context that stages (its threads only ever run memcpy jobs); created under the
// context's error lock: the devices' host threads may get here together
int ensure_stagers(te_ctx* ctx) {
  std::lock_guard<std::mutex> lk(ctx->err_mu);
  while ((int)ctx->stagers.size() < stager_count()) ctx->stagers.emplace_back(new te_sched::worker_t());
  return 0;
}
// dst (device) <- src (host), `bytes`, on `stream`, through the set's ring; returns when src has been read completely
int staged_copy(te_ctx* ctx, workset_t& ws, void* dst, const uint8_t* src, size_t bytes, hipStream_t stream) {
  if (int rc = ensure_ring(ctx, ws)) return rc;
  if (int rc = ensure_stagers(ctx)) return rc;
  struct pending_t { te_sched::job_ref job; te_sched::worker_t* who; int slot; size_t off, len; };
  std::deque<pending_t> fifo;
  auto flush_one = [&]() -> int {
    pending_t p = fifo.front(); fifo.pop_front();
    (void)p.who->wait(p.job);
    HIP_TRY(ctx, hipMemcpyAsync(static_cast<uint8_t*>(dst) + p.off, ws.h_ring + (size_t)p.slot * TE_RING_SLOT, p.len, hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipEventRecord(ws.ring_ev[(size_t)p.slot], stream));
    return 0;
  };
  // whatever happens, no crew thread may still be reading the caller's buffer when this function returns
  struct drain_t { std::deque<pending_t>& f; ~drain_t() { for (auto& p : f) (void)p.who->wait(p.job); } } drain_on_exit{fifo};
  for (size_t off = 0; off < bytes; off += TE_RING_SLOT) {
    const size_t len = std::min(TE_RING_SLOT, bytes - off);
    const size_t turn = ws.ring_next++;
    const int slot = (int)(turn % TE_RING_SLOTS);
    if ((int)fifo.size() >= std::min(stager_count(), TE_RING_SLOTS - 4)) { if (int rc = flush_one()) return rc; }   // one chunk per crew thread filling, the rest of the ring draining
    if (turn >= (size_t)TE_RING_SLOTS) HIP_TRY(ctx, hipEventSynchronize(ws.ring_ev[(size_t)slot]));   // the slot's previous chunk has left it
    uint8_t* to = ws.h_ring + (size_t)slot * TE_RING_SLOT; const uint8_t* from = src + off;
    te_sched::worker_t* who = ctx->stagers[turn % (size_t)stager_count()].get();
    fifo.push_back({who->post([to, from, len] { staging_chunk_copy(to, from, len); return 0; }), who, slot, off, len});
  }
  while (!fifo.empty()) { if (int rc = flush_one()) return rc; }
  return 0;
}
// every host-to-device copy of a caller's buffer goes through here
int upload(te_ctx* ctx, workset_t& ws, void* dst, const uint8_t* src, size_t bytes, hipStream_t stream) {
  if (ctx->opt_host_staging && bytes >= (256u << 10)) return staged_copy(ctx, ws, dst, src, bytes, stream);
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
  return 0;
}

// The uploads of a host-buffer MSM on work set `ws` (side stream) must not overtake the set's previous MSM, which may still read the
// staging area.  A marker on the set's stream and a stream wait say so -- but the marker is a packet in the hardware queue the set's
// stream shares with other work sets, and stands there behind THEIR kernels (a k_accumulate of 1.2-1.4 ms with eight tickets in flight):
// a third of the 32 MB scalar uploads of bound-bases tickets took 1.6-1.9 ms instead of 0.61 (TE_MSM_TRACE_HOST stamps,
// profiles/r06_bound_host_tickets_gap.txt).  A set whose previous MSM has delivered its result -- every ticket that was collected --
// needs no marker: everything that read the staging area precedes ev_result.  TE_MSM_COPY_MARKER=1: always (the old behaviour).
int copy_stream_behind_previous(te_ctx* ctx, workset_t& ws) {
  if (int rc = need_copy_stream(ctx, ws)) return rc;
  static const bool always = [] { const char* e = getenv("TE_MSM_COPY_MARKER"); return e && e[0] == '1'; }();
  if (!always) {
    if (!ws.used) return 0;
    const hipError_t q = hipEventQuery(ws.ev_result);
    if (q == hipSuccess) return 0;
    (void)hipGetLastError();                                                      // hipErrorNotReady is an answer, not an error
  }
  HIP_TRY(ctx, hipEventRecord(ws.ev_start, ws.stream));
  HIP_TRY(ctx, hipStreamWaitEvent(ws.copy_stream, ws.ev_start, 0));
  return 0;
}

const char* const kFinalCarry = "final carry is 1: a scalar does not fit the signed window decomposition";

// "The kernels behind this point read what the upload recorded in `ev` (on the copy stream) brought."  Two forms:
//   the calling thread owns the call (te_msm_run*, te_msm_submit) and its buffers are PINNED: a stream wait -- nothing of the host is in
//     the MSM's critical path (pageable buffers: the host form as well, see caller_may_wait_on_host below);
//   a LANE thread enqueues an asynchronous ticket (wait_for_pinned == false): the thread itself waits for the upload and enqueues the
//     kernels afterwards.  A stream wait sits in the hardware queue of the work set's stream until the upload is over (2-3 ms with four
//     lanes sharing the link), and the runtime multiplexes the eight work-set streams onto four hardware queues, whose packets run in
//     order: the kernels of ANOTHER ticket that shares the queue stood behind that wait -- tickets from host scalars over bound bases
//     ran at 1.02-1.09 ms per MSM where the same tickets from device scalars take 0.89-0.92, with neither the link nor the device busy
//     (profiles/r06_lane_host_waits.txt).  The lane thread has nothing else to do.
//     It waits for the copy STREAM, not for an event recorded behind the upload: an event record is one more packet in a hardware
//     queue the side stream shares with other work sets' streams, and stood behind their kernels for 0.5-0.7 ms in every fourth
//     ticket (stamps: "upload awaited"); the stream's last command -- the copy -- is known to the runtime without a packet.
//     TE_MSM_LANE_EVENT_WAITS=1: the event form (experiments).
// lane: the host form is wanted (a lane thread, or a calling thread whose copies have blocked anyway)
int lane_wait(te_ctx* ctx, workset_t& ws, hipEvent_t ev, bool lane) {
  static const bool by_event = [] { const char* e = getenv("TE_MSM_LANE_EVENT_WAITS"); return e && e[0] == '1'; }();
  if (lane && ctx->opt_lane_host_waits && !by_event) { HIP_TRY(ctx, hipStreamSynchronize(ws.copy_stream)); return 0; }
  HIP_TRY(ctx, hipEventRecord(ev, ws.copy_stream));
  if (lane && ctx->opt_lane_host_waits) HIP_TRY(ctx, hipEventSynchronize(ev));
  else HIP_TRY(ctx, hipStreamWaitEvent(ws.stream, ev, 0));
  return 0;
}

// te_msm_run* / te_msm_submit (the calling thread uploads): from PAGEABLE memory hipMemcpyAsync returns when the copy is over, so
// waiting for the copy stream on the host costs nothing and keeps the event record and the stream wait out of the hardware queues
// as well: te_msm_run 2.30 vs 2.35 ms, te_msm_run_scalars 1.355 vs 1.38, te_msm_submit x8 in flight 1.90 vs 1.93-2.01 at 2^20;
// 0.78 vs 0.81, 0.573 vs 0.595, 0.555 vs 0.56-0.63 at 2^18 (tools/exp_caller_host_waits.py, profiles/r06_caller_host_waits.txt).
// Pinned sources keep the stream waits (their copies are asynchronous).  TE_MSM_CALLER_HOST_WAITS=0: stream waits for all.
bool caller_may_wait_on_host(const te_ctx* ctx, const void* a, const void* b) {
  static const bool on = [] { const char* e = getenv("TE_MSM_CALLER_HOST_WAITS"); return !(e && e[0] == '0'); }();
  return on && ctx->opt_lane_host_waits && !(a && host_memory_is_pinned(a)) && !(b && host_memory_is_pinned(b));
}

// pieces a host buffer of n points is uploaded and processed in on ONE device (option "host_chunks", else from n).
// measured (tools/sweep_host_chunks.py, ms for 1 / 2 / 3 pieces): 2^17 0.672 / 0.659 / 0.773, 2^18 1.005 / 0.927 / 1.001,
// 2^19 1.642 / 1.407 / 1.424, 2^20 2.991 / 2.464 / 2.380
int host_pieces(const te_ctx* ctx, uint64_t n) {
  int K = ctx->opt_host_chunks ? ctx->opt_host_chunks : (n >= (3ull << 18) ? 3 : n >= (1ull << 17) ? 2 : 1);
  if ((uint64_t)K > n) K = (int)n;
  return K < 1 ? 1 : K;
}

// A host-buffer MSM (or one device's slice of it) on work set `ws` of device `d`: the buffers are uploaded and processed in K
// pieces -- while piece i is on the GPU, piece i+1 crosses PCIe -- whose additions land on the SAME buckets, one reduction at
// the end, flag + rows on their way to the set's pinned block when the call returns.  Window bits are forced to c (the slices
// of a point-sharded MSM must agree on the geometry of their rows).  Does not wait: the caller synchronises ws.ev_result.
// All windows, whatever the device's window shard says (host buffers always are whole MSMs or point slices of one).
// wait_for_pinned: the caller promises its buffers only until this call returns (te_msm_run, te_msm_submit).  Copies from
// PAGEABLE memory have left the caller's buffer when hipMemcpyAsync returns (the runtime stages them); from pinned or
// registered memory they are truly asynchronous -- then the call waits for the last piece's upload before it returns.
int enqueue_host_slice(te_ctx* ctx, gpu_t& d, workset_t& ws, const uint8_t* src_points, const uint8_t* src_scalars, uint64_t n, int c, int K,
                       bool wait_for_pinned = true) {
  HIP_TRY(ctx, hipSetDevice(d.device));
  plan_t pf; make_plan(ctx, d, n, pf, c, 1, 0, true);
  const curve_sizes sz = sizes_of(pf.curve);
  if (int rc = ensure_staging(ctx, ws, n * sz.point_in, n * sz.scalar_in)) return rc;
  const bool host_waits = !wait_for_pinned || caller_may_wait_on_host(ctx, src_points, src_scalars);
  uint8_t* dpts = static_cast<uint8_t*>(ws.d_in_points);
  uint8_t* dscs = static_cast<uint8_t*>(ws.d_in_scalars);
  // Pieces of n / K points: piece i crosses PCIe on the side stream while piece i-1 is converted and ACCUMULATED ONTO THE SAME
  // BUCKETS on the main stream; one bucket reduction at the end.  (The first version ran every piece as a complete MSM with
  // its own reduction of all W x 2^(c-1) buckets: K reductions, which made more than four pieces a loss.)  Pageable copies
  // return once the data has left the caller's buffer, so the host alternates between staging a piece and enqueueing the
  // previous piece's kernels.
  // Measured (n = 2^20, tools/host_path.py, TE_MSM_TRACE_HOST=1, profiles/r03_host_buffer_path.txt): three equal pieces 2.48 ms,
  // two 2.56, four 2.59 (the device falls behind the uploads: every piece pays a sort and ~10 launches on a fraction of the
  // points), falling piece sizes 2.51-2.56; the link alone needs 1.85 ms.
  // piece boundaries: equal pieces, or TE_MSM_HOST_SPLIT="w0,w1,..." (relative weights, experiments; read once in te_msm_init)
  std::vector<uint64_t> bounds((size_t)K + 1, n);
  {
    std::vector<double> wgt((size_t)K, 1.0);
    for (size_t i = 0; i < ctx->host_split.size() && i < (size_t)K; i++) wgt[i] = ctx->host_split[i];
    double tot = 0, run = 0; for (double v : wgt) tot += v;
    for (int i = 0; i < K; i++) { bounds[(size_t)i] = (uint64_t)((double)n * (run / tot)); run += wgt[(size_t)i]; }
    bounds[0] = 0; bounds[(size_t)K] = n;
    for (int i = 1; i <= K; i++) if (bounds[(size_t)i] < bounds[(size_t)i - 1]) bounds[(size_t)i] = bounds[(size_t)i - 1];
  }
  auto piece_lo = [&](int i) -> uint64_t { return i >= K ? n : bounds[(size_t)i]; };     // first point of piece i
  uint64_t m_max = 0;
  for (int i = 0; i < K; i++) m_max = std::max(m_max, piece_lo(i + 1) - piece_lo(i));
  uint32_t seg_all = 0;
  {
    plan_t pm; make_plan(ctx, d, m_max, pm, pf.c, 1, 0, true);
    seg_all = pm.seg_len;
    if (int rc = ensure_buffers(ctx, d, ws, m_max, pm)) return rc;        // every buffer at its final size before the first piece
  }
  release_shared_recs(d, ws, false); ws.recs_last = ws.d_recs;      // a host-buffer MSM converts into the set's own slab
  if (ws.used && ws.last_stream != ws.stream) HIP_TRY(ctx, hipStreamWaitEvent(ws.stream, ws.ev_done, 0));
  if (int rc = copy_stream_behind_previous(ctx, ws)) return rc;
  const bool tr = getenv("TE_MSM_TRACE_HOST") != nullptr;
  const auto t00 = std::chrono::steady_clock::now();
  // (index of the device in the context's list / its HIP id; the absolute time tells the threads of one call apart from the next call's)
  auto stamp = [&](const char* what, int i) {
    if (!tr) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[te_msm_run dev %d/%d] %8.1f us  %s %d   (t = %.1f us)\n", (int)(&d - ctx->devs.data()), d.device,
            std::chrono::duration<double, std::micro>(now - t00).count(), what, i, std::chrono::duration<double, std::micro>(now.time_since_epoch()).count());
  };
  std::vector<hipEvent_t>& evs = ws.piece_events;
  while ((int)evs.size() < K + 1) { hipEvent_t e; HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); evs.push_back(e); }
  plan_t p;
  bool first = true;
  int last_piece = K - 1;
  while (last_piece > 0 && piece_lo(last_piece) == piece_lo(last_piece + 1)) last_piece--;       // empty tail pieces
  // ALL scalars first, in one copy (a third of the bytes): every extra pageable copy call costs ~43 us of pipeline drain
  // (six calls for three pieces ended at 2.02 ms where one pair of calls takes 1.85), and with the scalars there the sort
  // of piece i + 1 is enqueued behind piece i's accumulation and runs while piece i + 1's points are still crossing PCIe.
  // (Twisted-Edwards only: there the link is the bottleneck.  BLS12-377 -- 144 MB to upload, but 2.0 ms of accumulation -- is
  // bound by the device, which must not sit idle while 48 MB of scalar records arrive: piece by piece, 4.1 against 4.7 ms.)
  const bool scalars_first = pf.curve == TE_MSM_CURVE_TE_BLS12;
  if (scalars_first) {
    if (int rc = upload(ctx, ws, dscs, src_scalars, n * sz.scalar_in, ws.copy_stream)) return rc;
    if (int rc = lane_wait(ctx, ws, evs[K], host_waits)) return rc;
    stamp("scalars staged", -1);
  }
  for (int i = 0; i < K; i++) {
    const uint64_t lo = piece_lo(i), hi = piece_lo(i + 1), m = hi - lo;
    if (m == 0) continue;
    make_plan(ctx, d, m, p, pf.c, 1, seg_all, true);
    if (int rc = ensure_buffers(ctx, d, ws, m, p)) return rc;              // no reallocation: only the pointers into the zeroed block move
    msm_launch L{ctx, d, ws, p, dpts + lo * sz.point_in, dscs + lo * sz.scalar_in, m, ws.d_partials, 0, ws.stream, true, !first};
    L.host_rows = msm_launch::rows_to_host(ctx, d, p, true);
    ws.rows_on_host = L.host_rows;
    first = false;
    if (!scalars_first) {
      if (int rc = upload(ctx, ws, dscs + lo * sz.scalar_in, src_scalars + lo * sz.scalar_in, m * sz.scalar_in, ws.copy_stream)) return rc;
      if (int rc = lane_wait(ctx, ws, evs[K], host_waits)) return rc;
      stamp("scalars staged", i);
    }
    if (int rc = L.front_scalars()) return rc;                             // digits, sort and schedule run while the piece's points cross PCIe
    stamp("scalar stages enqueued", i);
    if (int rc = upload(ctx, ws, dpts + lo * sz.point_in, src_points + lo * sz.point_in, m * sz.point_in, ws.copy_stream)) return rc;
    stamp("points staged", i);
    if (int rc = lane_wait(ctx, ws, evs[i], host_waits)) return rc;
    if (int rc = L.front_points()) return rc;
    if (int rc = L.accumulate()) return rc;
    if (int rc = L.combine()) return rc;
    if (i == last_piece) { if (int rc = L.reduce()) return rc; }
    stamp("piece enqueued", i);
  }
  ws.plan = p; ws.n = piece_lo(last_piece + 1) - piece_lo(last_piece); ws.used = true; ws.last_stream = ws.stream; ws.prof_level = 0;
  __atomic_store_n(&d.last_ws, (int)(&ws - d.ws), __ATOMIC_RELAXED);      // (the device's host thread may be the writer: asynchronous submits)
  if (int rc = fetch_rows(ctx, ws, ws.stream)) return rc;
  HIP_TRY(ctx, hipGetLastError());
  if (wait_for_pinned && !ctx->opt_host_staging && (host_memory_is_pinned(src_points) || host_memory_is_pinned(src_scalars))) {
    // everything on the copy stream is ordered: the last recorded upload event covers the scalars and every piece
    HIP_TRY(ctx, hipEventSynchronize(evs[(size_t)last_piece]));
    stamp("pinned source: uploads awaited", -1);
  }
  return 0;
}

// pieces the scalars of a bound point set are uploaded and processed in on ONE device (option "scalar_chunks", else from n)
int scalar_pieces(const te_ctx* ctx, uint64_t n) {
  int K = ctx->opt_scalar_chunks ? ctx->opt_scalar_chunks : (n >= (3ull << 18) ? 3 : n >= (1ull << 18) ? 2 : 1);
  if ((uint64_t)K > n) K = (int)n;
  return K < 1 ? 1 : K;
}

// An MSM over a BOUND point set from host scalars (or one device's slice of it) on work set `ws` of device `d`: enqueue_host_slice
// without the points.  `recs` are the records of the slice's first point on this device (rec_kind: their form).  The scalars
// travel in K pieces on the set's copy stream -- piece i + 1 crosses PCIe while piece i is decomposed, sorted and accumulated
// ONTO THE SAME BUCKETS on the main stream (its records are at recs + lo * record bytes) --, one bucket reduction at the end,
// flag + rows on their way to the set's pinned block when the call returns.  Does not wait: the caller synchronises ws.ev_result.
// All windows, window bits forced to c.  wait_for_pinned: as enqueue_host_slice.
int enqueue_scalar_slice(te_ctx* ctx, gpu_t& d, workset_t& ws, const uint8_t* recs, int rec_kind, const uint8_t* src_scalars, uint64_t n, int c, int K,
                         bool wait_for_pinned = true) {
  HIP_TRY(ctx, hipSetDevice(d.device));
  plan_t pf; make_plan(ctx, d, n, pf, c, 1, 0, true);
  pf.rec_kind = rec_kind;
  const curve_sizes sz = sizes_of(pf.curve);
  const size_t rec_bytes = rec_bytes_of(pf.curve, rec_kind);
  if (int rc = ensure_staging(ctx, ws, 0, n * sz.scalar_in)) return rc;
  uint8_t* dscs = static_cast<uint8_t*>(ws.d_in_scalars);
  if (K < 1) K = 1;
  if ((uint64_t)K > n) K = (int)n;
  auto piece_lo = [&](int i) -> uint64_t { return i >= K ? n : (uint64_t)(((unsigned __int128)n * (unsigned)i) / (unsigned)K); };
  uint64_t m_max = 0;
  for (int i = 0; i < K; i++) m_max = std::max(m_max, piece_lo(i + 1) - piece_lo(i));
  uint32_t seg_all = 0;
  {
    plan_t pm; make_plan(ctx, d, m_max, pm, pf.c, 1, 0, true);
    pm.rec_kind = rec_kind;
    seg_all = pm.seg_len;
    if (int rc = ensure_buffers(ctx, d, ws, m_max, pm, false)) return rc;      // every buffer at its final size before the first piece
  }
  release_shared_recs(d, ws, false); ws.recs_last = nullptr;
  if (ws.used && ws.last_stream != ws.stream) HIP_TRY(ctx, hipStreamWaitEvent(ws.stream, ws.ev_done, 0));
  if (int rc = copy_stream_behind_previous(ctx, ws)) return rc;                 // the staging area may still be read by the set's previous MSM
  std::vector<hipEvent_t>& evs = ws.piece_events;
  while ((int)evs.size() < K + 1) { hipEvent_t e; HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); evs.push_back(e); }
  plan_t p;
  bool first = true;
  static const bool tr = getenv("TE_MSM_TRACE_HOST") != nullptr;          // host stamps of every piece (stderr)
  const auto t00 = std::chrono::steady_clock::now();
  auto stamp = [&](const char* what, int i) {
    if (tr) fprintf(stderr, "[scalar slice ws %d] %8.1f us  %s %d\n", (int)(&ws - d.ws), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t00).count(), what, i);
  };
  for (int i = 0; i < K; i++) {
    const uint64_t lo = piece_lo(i), m = piece_lo(i + 1) - lo;
    if (m == 0) continue;
    {
      static const bool serial = [] { const char* e = getenv("TE_MSM_SCALAR_UPLOADS_SERIAL"); return !(e && e[0] == '0'); }();
      std::unique_lock<std::mutex> link(*d.scalar_link, std::defer_lock);
      if (!wait_for_pinned && serial && ctx->opt_lane_host_waits) link.lock();       // (a lane thread: see gpu_t::scalar_link)
      stamp("upload begins", i);
      if (int rc = upload(ctx, ws, dscs + lo * sz.scalar_in, src_scalars + lo * sz.scalar_in, m * sz.scalar_in, ws.copy_stream)) return rc;
      stamp("upload call returned", i);
      if (int rc = lane_wait(ctx, ws, evs[(size_t)i], !wait_for_pinned || caller_may_wait_on_host(ctx, src_scalars, nullptr))) return rc;
      stamp("upload awaited", i);
    }
    make_plan(ctx, d, m, p, pf.c, 1, seg_all, true);
    p.rec_kind = rec_kind;
    if (int rc = ensure_buffers(ctx, d, ws, m, p, false)) return rc;           // no reallocation: only the pointers into the zeroed block move
    msm_launch L{ctx, d, ws, p, nullptr, dscs + lo * sz.scalar_in, m, ws.d_partials, 0, ws.stream, true, !first};
    L.host_rows = msm_launch::rows_to_host(ctx, d, p, true);
    L.bound = recs + lo * rec_bytes;
    ws.rows_on_host = L.host_rows;
    first = false;
    if (int rc = L.front_scalars()) return rc;
    if (int rc = L.accumulate()) return rc;
    if (int rc = L.combine()) return rc;
    if (i == K - 1) { if (int rc = L.reduce()) return rc; }
    stamp("piece enqueued", i);
  }
  ws.plan = p; ws.n = piece_lo(K) - piece_lo(K - 1); ws.used = true; ws.last_stream = ws.stream; ws.prof_level = 0;
  __atomic_store_n(&d.last_ws, (int)(&ws - d.ws), __ATOMIC_RELAXED);
  if (int rc = fetch_rows(ctx, ws, ws.stream)) return rc;
  HIP_TRY(ctx, hipGetLastError());
  if (wait_for_pinned && !ctx->opt_host_staging && host_memory_is_pinned(src_scalars)) HIP_TRY(ctx, hipEventSynchronize(evs[(size_t)K - 1]));
  return 0;
}

// ---- fixed-base windows (kernels.hip.hpp, "FIXED-BASE WINDOWS"): the plan of an MSM over a bound set that carries a per-window table.
// The sort / accumulation / reduction see fb_rb + 1 pseudo-windows ("rows") of 2^15 buckets and `cap` entries each: an unsigned
// geometry (codes lo + 1, half = 0), general sort entries (the table index needs 24 bits).
constexpr uint32_t FB_LOBITS = TE_FB_LOBITS;
int fb_windows_for(int c) { return (255 + c - 1) / c; }
void make_plan_fb(const te_ctx* ctx, const gpu_t& d, uint64_t n, int fb_c, plan_t& p) {
  const int W = fb_windows_for(fb_c);
  const uint32_t rb = 1u << ((uint32_t)fb_c - 1u - FB_LOBITS);
  // a regular row holds (W - 1) n / rb entries for well-spread digits, the extra row the top window's n: capacity for the larger of
  // the two plus a sixteenth (an overflow -- badly skewed scalars -- is detected on the device and falls back to the ordinary windows)
  const uint64_t reg = ((uint64_t)(W - 1) * n + rb - 1) / rb, big = std::max<uint64_t>(reg, n);
  const uint64_t cap = (big + big / 16 + 4096 + 7) & ~(uint64_t)7;
  make_plan(ctx, d, cap, p, 16, 1, 0, true);           // chunk geometry, segment length and partitions for rows of `cap` entries
  p.fb_rb = (int)rb; p.fb_c = fb_c; p.fb_W = W;
  p.signed_digits = 0; p.c = (int)FB_LOBITS; p.W = (int)rb + 1; p.nw = p.nw1 = (int)rb + 1; p.batch = 1; p.w_first = 0; p.w_step = 1;
  p.logB = FB_LOBITS; p.B = 1u << FB_LOBITS;
  for (int k = 0; k < 4; k++) p.dw[k] = (p.logB + 3u - (uint32_t)k) / 4u;
  p.S = 256u; p.P = p.B / p.S; p.logS = 8u;
  p.packed = 0; p.rec_kind = 0;
  // chunks per row: as make_plan, for the row count of this plan
  uint32_t ch = TE_SCATTER_BLOCKS / (uint32_t)p.nw; if (ch < 1) ch = 1; if (ch > 256) ch = 256;
  const uint32_t by_n = (uint32_t)((cap + TE_SCATTER_MINCHUNK - 1) / TE_SCATTER_MINCHUNK); if (ch > by_n) ch = by_n ? by_n : 1;
  p.nst = (uint32_t)cap;
  p.chunk_len = ((p.nst + ch - 1) / ch + 4095u) & ~4095u;
  p.CH = (p.nst + p.chunk_len - 1) / p.chunk_len;
  if (!ctx->opt_seg_len) {                              // twice the mean bucket size of the ONE bucket set, as a power of two in [16, 64]
    const uint64_t a = (2ull * (uint64_t)W * n + ((uint64_t)rb << FB_LOBITS) - 1) / ((uint64_t)rb << FB_LOBITS); uint64_t s2 = 16;
    while (s2 < a && s2 < 64) s2 <<= 1;
    p.seg_len = (uint32_t)s2;
  }
}

template <int C> void launch_fb_digits(const void* d_scalars, const te::fb_digit_args& a, uint32_t n, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL(te::k_fb_digits<C>, dim3((n + TE_FB_THREADS - 1u) / TE_FB_THREADS), dim3(TE_FB_THREADS), lds, st, static_cast<const uint4*>(d_scalars), a);
}

// An MSM of n scalars (device memory) over the fixed-base table of `bases` on device `d`: the scalars' digits address the table
// entries (w, idx_base + i) -- idx_base: the first point of this launch inside the bound set (a device's slice of a lone
// multi-device call).  Rows in the set's own block; the caller ends the sequence with fetch_rows().
int enqueue_fixed_base(te_ctx* ctx, gpu_t& d, workset_t& ws, const te_bases* bases, const void* d_scalars, uint64_t n, uint64_t idx_base, hipStream_t stream) {
  HIP_TRY(ctx, hipSetDevice(d.device));
  plan_t p; make_plan_fb(ctx, d, n, bases->fb_c, p);
  const uint64_t cap = p.nst;
  if ((uint64_t)p.nw * cap >= (1ull << 31) || (uint64_t)bases->fb_W * bases->n >= (1ull << 31))
    return set_err(ctx, TE_MSM_EINVAL, "fixed-base windows: windows x points must stay below 2^31");
  if (int rc = ensure_buffers(ctx, d, ws, cap, p, false)) return rc;
  release_shared_recs(d, ws, false); ws.recs_last = nullptr;
  if (ws.used && ws.last_stream != stream) HIP_TRY(ctx, hipStreamWaitEvent(stream, ws.ev_done, 0));
  ws.plan = p; ws.n = cap; ws.used = true; ws.last_stream = stream; __atomic_store_n(&d.last_ws, (int)(&ws - d.ws), __ATOMIC_RELAXED);
  ws.prof_level = ctx->opt_profile;
  ws.fb_scalars = d_scalars; ws.fb_n = n;
  msm_launch L{ctx, d, ws, p, nullptr, d_scalars, cap, ws.d_partials, ctx->opt_profile, stream, true};
  L.host_rows = msm_launch::rows_to_host(ctx, d, p, true);
  ws.rows_on_host = L.host_rows;
  L.bound = bases->recs[(size_t)(&d - ctx->devs.data())];
  L.fb_remap = ws.d_fb_remap;
  // the zeroed block (flags, counters, level-1 histogram, row fill) -- then the digits.  The digit rows are NOT cleared: the sort's
  // first level reads every row only up to its fill (scatter_args.row_fill)
  if (ws.zero_clean_words < ws.zero_words) HIP_TRY(ctx, hipMemsetAsync(ws.d_zero, 0, ws.zero_words * sizeof(uint32_t), stream));
  ws.zero_clean_words = ws.zero_words;                                  // front_scalars below must not clear it again (it would wipe the histogram)
  L.mark(ST_DIGITS);
  {
    te::fb_digit_args a; memset(&a, 0, sizeof a);
    for (int w = 0; w < p.fb_W; w++) { const int bit = w * p.fb_c + p.fb_c - 1; if (bit < 320) a.half[bit >> 5] |= 1u << (bit & 31); }
    a.n = (uint32_t)n; a.idx_base = (uint32_t)idx_base; a.n_table = (uint32_t)bases->n; a.W = (uint32_t)p.fb_W;
    a.rows = (uint32_t)p.nw; a.rb = (uint32_t)p.fb_rb; a.cap = (uint32_t)cap;
    a.chunk_len = p.chunk_len; a.CH = p.CH; a.P = p.P; a.logS = p.logS;
    a.digits = ws.d_digits; a.remap = ws.d_fb_remap; a.row_fill = ws.d_fb_fill; a.counts1 = ws.d_counts1; a.err = ws.d_err;
    const size_t lds = (size_t)(2u * a.rows + a.rows * 2u * a.P) * sizeof(uint32_t);
    switch (p.fb_c) {
      case 16: launch_fb_digits<16>(d_scalars, a, a.n, lds, stream); break;
      case 17: launch_fb_digits<17>(d_scalars, a, a.n, lds, stream); break;
      case 18: launch_fb_digits<18>(d_scalars, a, a.n, lds, stream); break;
      case 19: launch_fb_digits<19>(d_scalars, a, a.n, lds, stream); break;
      case 20: launch_fb_digits<20>(d_scalars, a, a.n, lds, stream); break;
      default: launch_fb_digits<21>(d_scalars, a, a.n, lds, stream); break;
    }
  }
  // the engine's own stages from here on (front_scalars skips k_digits for a fixed-base plan)
  if (int rc = L.front_scalars()) return rc;
  L.mark(ST_PREP);
  if (int rc = L.accumulate()) return rc;
  L.mark(ST_TREE);
  if (int rc = L.back()) return rc;
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// the host tail of the rows a plan produced (ordinary windows: Horner; fixed-base windows: one bucket set, no window doublings)
void fold_rows(const plan_t& p, const uint8_t* rows, uint8_t* out) {
  if (p.fb_rb) te_host::fixed_base_to_affine(rows, p.fb_rb + 1, p.fb_rb, (int)p.logB, out);
  else if (p.curve == TE_MSM_CURVE_BLS12_377_G1) te377_host::horner_to_affine(rows, p.c, (int)p.logB, p.W, out);
  else te_host::horner_to_affine(rows, p.c, (int)p.logB, p.W, out);
}

// the host thread of device i of the context (host_sched.hpp), created on first use
te_sched::worker_t& worker_of(te_ctx* ctx, size_t i) { return te_sched::worker_of(*ctx, i); }
// every job posted to the context's host threads has run (asynchronous submits touch work sets, options and the error
// string: calls that change or read those beside them wait first)
void drain_workers(te_ctx* ctx) { te_sched::drain_workers(*ctx); }

// the lowest-numbered work set of a device that no ticket owns (-1: none)
int free_workset_index(const gpu_t& d) { return te_sched::free_set_index(d, TE_MSM_WORKSETS); }
const char* const kAllSetsOwned = "every work set holds a submitted MSM that has not been collected: te_msm_collect one first";

// the non-zero window digits the device counted for the MSM whose flag words are in the set's pinned block (word 1)
int64_t entries_of(const workset_t& ws) { uint64_t v; memcpy(&v, ws.h_err + Z_ENTRIES, sizeof v); return (int64_t)v; }
void note_entries(te_ctx* ctx, const workset_t& ws) { ctx->stat_entries = entries_of(ws); }

// te_msm_run for a large MSM on one device: enqueue_host_slice on a free work set, wait, host tail
int run_host_chunked(te_ctx* ctx, const uint8_t* src_points, const uint8_t* src_scalars, uint64_t n, int K, uint8_t out[64]) {
  gpu_t& d = ctx->devs[0];
  workset_t& ws = d.ws[0];
  plan_t pf; make_plan(ctx, d, n, pf, 0, 1, 0, true);
  if (int rc = enqueue_host_slice(ctx, d, ws, src_points, src_scalars, n, pf.c, K)) return rc;
  HIP_TRY(ctx, hipEventSynchronize(ws.ev_result));
  note_entries(ctx, ws);
  if (*ws.h_err) return set_err(ctx, TE_MSM_ESCALAR, kFinalCarry);
  if (pf.curve == TE_MSM_CURVE_BLS12_377_G1) te377_host::horner_to_affine(ws.h_partials, pf.c, (int)pf.logB, pf.W, out);
  else te_host::horner_to_affine(ws.h_partials, pf.c, (int)pf.logB, pf.W, out);
  return 0;
}

// te_msm_run on a context of D > 1 devices: POINT sharding.  Points and scalars are cut into contiguous slices, one per
// device; one host thread per device stages its slice over that device's own PCIe link (pageable copies block the thread
// that issues them) and runs ALL windows on its n/D points (enqueue_host_slice, itself in pieces for large slices); the D x W
// rows are linear in the bucket contents, so the host tail folds their sum (horner_to_affine_multi).  Every device pays the
// reduction of all W x B buckets -- which is why device-resident inputs (te_msm_run_device, one process per GPU) shard
// the WINDOWS instead -- but on this boundary the cost is the link: 1.85 of 2.47 ms at n = 2^20 on one device.  The window size
// follows the slice (all slices share it).  Reference: compute_msm uploads inside the call (cuzk/gpu.ts:33-46,
// submission.ts:73-78); multi-device is its README's future work (README.md:551).
// This is the form for the LONE call.  A caller with several MSMs to do keeps whole MSMs in flight instead, one per device
// (te_msm_submit / te_msm_submit_async below): no replicated bucket reduction, no row merge.
// bases: the MSM runs over a bound point set -- every device holds all records, so device i takes slice i of the SCALARS over its
// link and gathers from its own copy of the records at the slice's offset (src_points is not read).
int run_host_sharded(te_ctx* ctx, const uint8_t* src_points, const uint8_t* src_scalars, uint64_t n, uint8_t* out, const te_bases* bases = nullptr) {
  const size_t nd = ctx->devs.size();
  uint64_t per_min = ctx->opt_host_shard_min > 0 ? (uint64_t)ctx->opt_host_shard_min : 1;
  size_t D = (size_t)std::min<uint64_t>(nd, std::max<uint64_t>(1, n / per_min));
  const uint64_t per = (n + D - 1) / D;
  plan_t p0; make_plan(ctx, ctx->devs[0], per, p0, 0, 1, 0, true);     // geometry of every slice's rows (window bits from the slice size)
  const curve_sizes sz = sizes_of(p0.curve);
  const int K = bases ? scalar_pieces(ctx, per) : host_pieces(ctx, per);
  // a work set per device that no ticket owns (tickets and lone calls may be mixed)
  std::vector<int> wsel(D, 0);
  for (size_t i = 0; i < D; i++) { wsel[i] = free_workset_index(ctx->devs[i]); if (wsel[i] < 0) return set_err(ctx, TE_MSM_ESTATE, kAllSetsOwned); }
  auto slice = [&](size_t i) -> int {
    const uint64_t lo = std::min<uint64_t>(n, per * i), hi = std::min<uint64_t>(n, lo + per);
    gpu_t& d = ctx->devs[i];
    if (hi == lo) return 0;
    workset_t& ws = d.ws[wsel[i]];
    if (bases) {
      if (int rc = enqueue_scalar_slice(ctx, d, ws, bases->recs[i] + lo * bases->rec_bytes, bases->rec_kind, src_scalars + lo * sz.scalar_in, hi - lo, p0.c, K)) return rc;
    } else {
      if (int rc = enqueue_host_slice(ctx, d, ws, src_points + lo * sz.point_in, src_scalars + lo * sz.scalar_in, hi - lo, p0.c, K)) return rc;
    }
    HIP_TRY(ctx, hipEventSynchronize(ws.ev_result));
    return 0;
  };
  std::vector<te_sched::job_ref> jobs(D);
  for (size_t i = 1; i < D; i++) jobs[i] = worker_of(ctx, i).post([&slice, i] { return slice(i); });
  int rc = slice(0);
  for (size_t i = 1; i < D; i++) { const int r = worker_of(ctx, i).wait(jobs[i]); if (!rc) rc = r; }
  if (rc) return rc;
  std::vector<const uint8_t*> sets;
  int64_t entries = 0;
  for (size_t i = 0; i < D; i++) {
    if (per * i >= n) break;
    workset_t& ws = ctx->devs[i].ws[wsel[i]];
    if (*ws.h_err) return set_err(ctx, TE_MSM_ESCALAR, kFinalCarry);
    entries += entries_of(ws);
    sets.push_back(ws.h_partials);
  }
  ctx->stat_entries = entries;
  const int ns = (int)sets.size();
  if (ns <= 2) {                                            // one or two sets: summed on the fly inside the fold
    if (p0.curve == TE_MSM_CURVE_BLS12_377_G1) te377_host::horner_to_affine_multi(sets.data(), ns, p0.c, (int)p0.logB, p0.W, out);
    else te_host::horner_to_affine_multi(sets.data(), ns, p0.c, (int)p0.logB, p0.W, out);
    return 0;
  }
  // more: the sets' rows are summed window by window by the devices' host threads (5 x (sets - 1) additions per window,
  // ~0.2 us each: summed on the fly by the one thread that folds, eight sets cost ~100 us of a 0.7 ms call), then one fold
  // over the merged points.  Thread t merges windows t, t + D, ...: distinct elements of merged[] and present[].
  const bool bls = p0.curve == TE_MSM_CURVE_BLS12_377_G1;
  std::vector<te_host::Pt> m9(bls ? 0 : (size_t)p0.W * 5);
  std::vector<te377_host::Pt> m14(bls ? (size_t)p0.W * 5 : 0);
  std::vector<uint8_t> present((size_t)p0.W, 0);
  auto merge = [&](size_t t) -> int {
    for (int w = (int)t; w < p0.W; w += (int)D) {
      if (bls) te377_host::merge_window_rows(sets.data(), ns, w, m14.data(), present.data());
      else te_host::merge_window_rows(sets.data(), ns, w, m9.data(), present.data());
    }
    return 0;
  };
  for (size_t i = 1; i < D; i++) jobs[i] = worker_of(ctx, i).post([&merge, i] { return merge(i); });
  (void)merge(0);
  for (size_t i = 1; i < D; i++) (void)worker_of(ctx, i).wait(jobs[i]);
  if (bls) te377_host::horner_to_affine_points(m14.data(), present.data(), p0.c, (int)p0.logB, p0.W, out);
  else te_host::horner_to_affine_points(m9.data(), present.data(), p0.c, (int)p0.logB, p0.W, out);
  return 0;
}

// te_msm_run_device on a context of D > 1 devices: WINDOW shards (device i computes windows i, i + D, ...), inputs resident
// on the first device.  Every device needs all n points and scalars; they travel as a scatter + all-gather over the
// point-to-point xGMI links instead of D - 1 full copies out of the first device's memory (SURVEY.md 8e "Inputs"):
//   phase 1  device i >= 1 pulls slice i (n / D points and their scalars) from the source           -- issued here, in turn: cheap,
//            and the "slice i has arrived" events must exist before any other device waits for them
//   phase 2  device i pulls slice 0 from the source and slice j from device j's staging area (j != i, j >= 1), behind the
//            event of phase 1 -- D - 1 different links into every device, 2 (D - 1) / D of the input out of the first
//            device instead of D - 1 times all of it
// Phase 2, the ~30 launches of the device's share and its read-back are enqueued by the device's own host thread: D enqueue
// sequences side by side instead of one after the other on the calling thread (0.35 ms of host time each).
// (One physical GPU named several times -- the only form a one-GPU box can run -- makes every copy a device-to-device copy.)
int run_device_window_shards(te_ctx* ctx, const void* src_points, const void* src_scalars, uint64_t n, uint8_t* out) {
  const size_t nd = ctx->devs.size();
  plan_t p0; make_plan(ctx, ctx->devs[0], n, p0);
  const curve_sizes sz = sizes_of(p0.curve);
  std::vector<int> wsel(nd, 0);
  for (size_t i = 0; i < nd; i++) { wsel[i] = free_workset_index(ctx->devs[i]); if (wsel[i] < 0) return set_err(ctx, TE_MSM_ESTATE, kAllSetsOwned); }
  const int src_dev = ctx->devs[0].device;
  const uint64_t per = (n + nd - 1) / nd;
  auto lo_of = [&](size_t j) { return std::min<uint64_t>(n, per * j); };
  const uint8_t* sp = static_cast<const uint8_t*>(src_points); const uint8_t* ss = static_cast<const uint8_t*>(src_scalars);
  // phase 1 (calling thread)
  for (size_t i = 1; i < nd; i++) {
    gpu_t& d = ctx->devs[i]; workset_t& ws = d.ws[wsel[i]];
    HIP_TRY(ctx, hipSetDevice(d.device));
    if (int rc = ensure_staging(ctx, ws, n * sz.point_in, n * sz.scalar_in)) return rc;
    if (ws.used && ws.ev_done) HIP_TRY(ctx, hipStreamWaitEvent(ws.stream, ws.ev_done, 0));      // the staging area may still be read by the set's previous MSM
    const uint64_t lo = lo_of(i), m = lo_of(i + 1) - lo;
    if (m) {
      HIP_TRY(ctx, hipMemcpyPeerAsync(static_cast<uint8_t*>(ws.d_in_points) + lo * sz.point_in, d.device, sp + lo * sz.point_in, src_dev, m * sz.point_in, ws.stream));
      HIP_TRY(ctx, hipMemcpyPeerAsync(static_cast<uint8_t*>(ws.d_in_scalars) + lo * sz.scalar_in, d.device, ss + lo * sz.scalar_in, src_dev, m * sz.scalar_in, ws.stream));
      ctx->stat_peer_copies += 2; ctx->stat_peer_bytes += (int64_t)(m * (sz.point_in + sz.scalar_in));
    }
    HIP_TRY(ctx, hipEventRecord(ws.ev_copy, ws.stream));
  }
  // phase 2 + the device's share, one host thread per device
  std::vector<int64_t> copies(nd, 0), bytes(nd, 0);
  auto share = [&](size_t i) -> int {
    gpu_t& d = ctx->devs[i]; workset_t& ws = d.ws[wsel[i]];
    HIP_TRY(ctx, hipSetDevice(d.device));
    const void *dp = src_points, *ds = src_scalars;
    if (i > 0) {
      for (size_t j = 0; j < nd; j++) {
        if (j == i) continue;
        const uint64_t lo = lo_of(j), m = lo_of(j + 1) - lo;
        if (!m) continue;
        const gpu_t& o = ctx->devs[j]; const workset_t& ows = o.ws[wsel[j]];
        const uint8_t* fp = j == 0 ? sp : static_cast<const uint8_t*>(ows.d_in_points);
        const uint8_t* fs = j == 0 ? ss : static_cast<const uint8_t*>(ows.d_in_scalars);
        if (j > 0) HIP_TRY(ctx, hipStreamWaitEvent(ws.stream, ows.ev_copy, 0));
        HIP_TRY(ctx, hipMemcpyPeerAsync(static_cast<uint8_t*>(ws.d_in_points) + lo * sz.point_in, d.device, fp + lo * sz.point_in, o.device, m * sz.point_in, ws.stream));
        HIP_TRY(ctx, hipMemcpyPeerAsync(static_cast<uint8_t*>(ws.d_in_scalars) + lo * sz.scalar_in, d.device, fs + lo * sz.scalar_in, o.device, m * sz.scalar_in, ws.stream));
        copies[i] += 2; bytes[i] += (int64_t)(m * (sz.point_in + sz.scalar_in));
      }
      dp = ws.d_in_points; ds = ws.d_in_scalars;
    }
    if (int rc = enqueue_partial(ctx, d, ws, dp, ds, n, nullptr, ws.stream)) return rc;
    return fetch_rows(ctx, ws, ws.stream);
  };
  std::vector<te_sched::job_ref> jobs(nd);
  for (size_t i = 1; i < nd; i++) jobs[i] = worker_of(ctx, i).post([&share, i] { return share(i); });
  int rc = share(0);
  for (size_t i = 1; i < nd; i++) { const int r = worker_of(ctx, i).wait(jobs[i]); if (!rc) rc = r; }
  // a device whose staging area other devices read must not reuse it before they are done: every set's next MSM waits for
  // its own ev_done only, so the call ends with all of them complete (below) -- the copies are over by then
  for (size_t i = 0; i < nd; i++) { ctx->stat_peer_copies += copies[i]; ctx->stat_peer_bytes += bytes[i]; }
  std::vector<uint8_t> merged((size_t)p0.W * sz.row, 0);
  int64_t entries = 0; bool carry = false;
  for (size_t i = 0; i < nd; i++) {
    gpu_t& d = ctx->devs[i];
    workset_t& ws = d.ws[wsel[i]];
    if (rc) { (void)hipSetDevice(d.device); (void)hipStreamSynchronize(ws.stream); continue; }      // leave nothing of a failed call in flight
    HIP_TRY(ctx, hipSetDevice(d.device));
    HIP_TRY(ctx, hipEventSynchronize(ws.ev_done));
    carry = carry || *ws.h_err != 0;
    entries += entries_of(ws);
    for (int w = d.w_first; w < p0.W; w += d.w_step)
      memcpy(&merged[(size_t)w * sz.row], ws.h_partials + (size_t)w * sz.row, sz.row);
  }
  if (rc) return rc;
  ctx->stat_entries = entries;
  if (carry) return set_err(ctx, TE_MSM_ESCALAR, kFinalCarry);
  (void)collect_stage_ms(ctx, ctx->devs[0], ctx->devs[0].ws[wsel[0]]);
  if (p0.curve == TE_MSM_CURVE_BLS12_377_G1) te377_host::horner_to_affine(merged.data(), p0.c, (int)p0.logB, p0.W, out);
  else te_host::horner_to_affine(merged.data(), p0.c, (int)p0.logB, p0.W, out);
  return 0;
}

int run_common(te_ctx* ctx, const void* src_points, const void* src_scalars, bool src_is_host, uint64_t n, uint8_t out[64]) {
  if (!ctx || !out) return TE_MSM_EINVAL;
  if (n >= (1ull << 31)) return set_err(ctx, TE_MSM_EINVAL, "n must be < 2^31");
  if (n == 0) {                                                     // empty sum = identity: (0, 1), or infinity (all zero)
    memset(out, 0, sizes_of(ctx->opt_curve).result);
    if (ctx->opt_curve == TE_MSM_CURVE_TE_BLS12) out[32] = 1;
    return 0;
  }
  if (!src_points || !src_scalars) return set_err(ctx, TE_MSM_EINVAL, "null input buffer");
  const size_t nd = ctx->devs.size();
  // several devices: host buffers -> slices of the points, one upload thread per device; device-resident inputs -> window shards
  if (nd > 1) {
    if (src_is_host) return run_host_sharded(ctx, static_cast<const uint8_t*>(src_points), static_cast<const uint8_t*>(src_scalars), n, out);
    return run_device_window_shards(ctx, src_points, src_scalars, n, out);
  }
  gpu_t& d = ctx->devs[0];
  if (src_is_host && !ctx->opt_profile && ctx->opt_workset == 0 && d.w_step == 1 && d.in_flight == 0) {
    // no tickets in flight, no stage timing requested: every work set is free for the pieces
    const int K = host_pieces(ctx, n);
    if (K > 1) return run_host_chunked(ctx, static_cast<const uint8_t*>(src_points), static_cast<const uint8_t*>(src_scalars), n, K, out);
  }
  // the work set this call runs on: the selected one, unless a submitted MSM still owns it -- then any free one; with every
  // set owned by a ticket the call is refused
  int wsel = ctx->opt_workset;
  if (te_sched::slot_ticket(d.ws[wsel].slot)) {
    wsel = free_workset_index(d);
    if (wsel < 0) return set_err(ctx, TE_MSM_ESTATE, kAllSetsOwned);
  }
  plan_t p0; make_plan(ctx, d, n, p0);
  const curve_sizes sz = sizes_of(p0.curve);
  workset_t& ws = d.ws[wsel];
  HIP_TRY(ctx, hipSetDevice(d.device));
  const void *dp = src_points, *ds = src_scalars;
  if (src_is_host) {
    if (int rc = ensure_staging(ctx, ws, n * sz.point_in, n * sz.scalar_in)) return rc;
    // scalars first; the points follow from inside enqueue_partial (pageable copies return when the data has left
    // the caller's buffer, so the scalar-only stages enqueued in between run while the points are still in flight)
    if (int rc = upload(ctx, ws, ws.d_in_scalars, static_cast<const uint8_t*>(src_scalars), n * sz.scalar_in, ws.stream)) return rc;
    dp = ws.d_in_points; ds = ws.d_in_scalars;
  }
  const std::function<int(hipStream_t)> upload_points = [&](hipStream_t side) -> int {
    // on the side stream, beside the scalar-only kernels on ws.stream; the conversion to records follows it there
    return upload(ctx, ws, ws.d_in_points, static_cast<const uint8_t*>(src_points), n * sz.point_in, side);
  };
  if (int rc = enqueue_partial(ctx, d, ws, dp, ds, n, nullptr, ws.stream, src_is_host ? &upload_points : nullptr, 0, false, 1, false, nullptr, !src_is_host)) return rc;
  if (int rc = fetch_rows(ctx, ws, ws.stream)) return rc;
  HIP_TRY(ctx, hipEventSynchronize(ws.ev_result));          // (pinned sources included: the call ends after its uploads)
  release_shared_recs(d, ws);                               // the call is over: its record slab (if shared) may serve another point buffer
  note_entries(ctx, ws);
  if (*ws.h_err) return set_err(ctx, TE_MSM_ESCALAR, kFinalCarry);
  (void)collect_stage_ms(ctx, d, ws);
  // a window-sharded single-device context (te_msm_set_window_shard) folds its own rows only: the others read as zero
  if (p0.curve == TE_MSM_CURVE_BLS12_377_G1) te377_host::horner_to_affine(ws.h_partials, p0.c, (int)p0.logB, p0.W, out);
  else te_host::horner_to_affine(ws.h_partials, p0.c, (int)p0.logB, p0.W, out);
  return 0;
}

}  // namespace

// ================================================================================================ C-ABI
extern "C" {

int te_msm_init(const int* device_ids, int n_dev, te_ctx** out) {
  device_guard restore_callers_device;
  if (!out || n_dev < 1 || n_dev > 64) { g_init_error = "te_msm_init: bad arguments"; return TE_MSM_EINVAL; }
  *out = nullptr;
  if (!te_host::tail_selftest() || !te377_host::tail_selftest()) { g_init_error = "te_msm_init: host tail constants self-test failed"; return TE_MSM_ESTATE; }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count < 1) {
    g_init_error = std::string("te_msm_init: no usable HIP device (") + hipGetErrorString(e) + "); this library has no CPU fallback";
    return TE_MSM_EDEVICE;
  }
  te_ctx* ctx = new te_ctx();
  if (const char* e = getenv("TE_MSM_FUSE_PREP")) ctx->opt_fuse_prep = e[0] != '0';      // A/B measurements; option "fuse_prep"
  if (const char* e = getenv("TE_MSM_QUEUE_PROBE")) ctx->opt_queue_probe = e[0] != '0';  // option "queue_probe"
  if (const char* e = getenv("TE_MSM_PACKED")) ctx->opt_packed = e[0] != '0';            // option "packed_sort"
  if (const char* e = getenv("TE_MSM_HOST_STAGING")) ctx->opt_host_staging = e[0] != '0'; // option "host_staging"
  if (const char* e = getenv("TE_MSM_UPLOAD_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 16) ctx->opt_upload_threads = v; }   // option "upload_threads"
  if (const char* e = getenv("TE_MSM_FOLD_PAIRS")) ctx->opt_fold_pairs = e[0] != '0';    // option "fold_pairs"
  if (const char* e = getenv("TE_MSM_SHARE_RECORDS")) ctx->opt_share_records = e[0] != '0';       // option "share_records"
  if (const char* e = getenv("TE_MSM_LANE_HOST_WAITS")) ctx->opt_lane_host_waits = e[0] != '0';   // option "lane_host_waits"
  if (const char* e = getenv("TE_MSM_HOST_SPLIT")) {                                     // relative piece weights "w0,w1,..." (experiments)
    const char* q = e;
    while (*q) { char* end = nullptr; const double v = strtod(q, &end); if (end == q) break; ctx->host_split.push_back(v > 0 ? v : 1.0); q = *end == ',' ? end + 1 : end; }
  }
  ctx->devs.resize(n_dev);
  for (int i = 0; i < n_dev; i++) {
    gpu_t& d = ctx->devs[i];
    d.device = device_ids ? device_ids[i] : i;
    d.w_first = i; d.w_step = n_dev;
    if (d.device < 0 || d.device >= count) { g_init_error = "te_msm_init: device id out of range"; delete ctx; return TE_MSM_EINVAL; }
    hipError_t er = hipSetDevice(d.device);
    if (er == hipSuccess) { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, d.device) == hipSuccess) d.wall_clock_khz = khz; }
    if (er == hipSuccess)                // k_reduce_tail keeps up to 256 + 16 points in LDS (39 KB / 61 KB)
      er = hipFuncSetAttribute(reinterpret_cast<const void*>(te::k_reduce_tail<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (er == hipSuccess)
      er = hipFuncSetAttribute(reinterpret_cast<const void*>(te::k_reduce_tail<14>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (er == hipSuccess && create_workset_streams(d) != 0) er = hipErrorOutOfMemory;
    for (workset_t& ws : d.ws) {       // the small fixed allocations of every work set; the big buffers come with the first MSM
      if (er == hipSuccess) er = hipEventCreateWithFlags(&ws.ev_copy, hipEventDisableTiming);
      if (er == hipSuccess) er = hipEventCreateWithFlags(&ws.ev_start, hipEventDisableTiming);
      if (er == hipSuccess) er = hipHostMalloc((void**)&ws.h_err, Z_ROWS * 4 + (size_t)TE_MAX_WINDOWS * TE_MAX_ROW_BYTES, hipHostMallocDefault);
      if (er == hipSuccess) ws.h_partials = reinterpret_cast<uint8_t*>(ws.h_err + Z_ROWS);
      if (er == hipSuccess) er = hipHostGetDevicePointer((void**)&ws.h_err_dev, ws.h_err, 0);
      if (er == hipSuccess) er = hipEventCreateWithFlags(&ws.ev_done, hipEventDisableTiming);
      if (er == hipSuccess) er = hipEventCreateWithFlags(&ws.ev_result, hipEventDisableTiming);
      for (auto& evn : ws.ev) if (er == hipSuccess) er = hipEventCreate(&evn);
      if (er == hipSuccess) *ws.h_err = 0;
    }
    if (er != hipSuccess) {
      g_init_error = std::string("te_msm_init: ") + hipGetErrorString(er);
      for (auto& dd : ctx->devs) free_dev(dd);
      delete ctx; return TE_MSM_EDEVICE;
    }
  }
  *out = ctx;
  return 0;
}

void te_msm_destroy(te_ctx* ctx) {
  device_guard restore_callers_device;
  if (!ctx) return;
  ctx->lanes.clear();                   // finishes the uploads of tickets that were never collected, joins the threads
  ctx->workers.clear();                 // joins the per-device host threads (idle between calls)
  ctx->stagers.clear();
  for (auto& d : ctx->devs) {
    (void)hipSetDevice(d.device);
    for (workset_t& ws : d.ws) {        // this context's streams only: other work of the process is none of its business
      if (ws.stream) (void)hipStreamSynchronize(ws.stream);
      if (ws.copy_stream) (void)hipStreamSynchronize(ws.copy_stream);
      if (ws.last_stream && ws.last_stream != ws.stream && ws.ev_done) (void)hipEventSynchronize(ws.ev_done);
    }
    free_dev(d);
  }
  for (te_bases* b : ctx->bases) {      // bound point sets the caller did not release
    for (size_t i = 0; i < b->recs.size() && i < ctx->devs.size(); i++)
      if (b->recs[i]) { (void)hipSetDevice(ctx->devs[i].device); (void)hipFree(b->recs[i]); }
    delete b;
  }
  delete ctx;
}

// (a device's host thread may be writing ctx->err this very moment -- an asynchronous submit that fails: a copy taken under
// the lock, per calling thread; valid until that thread asks again)
const char* te_msm_last_error(const te_ctx* ctx) {
  if (!ctx) return g_init_error.c_str();
  thread_local std::string copy;
  { std::lock_guard<std::mutex> lk(const_cast<te_ctx*>(ctx)->err_mu); copy = ctx->err; }
  return copy.c_str();
}

int te_msm_run(te_ctx* ctx, const uint8_t* points_xy_le, const uint8_t* scalars_le, uint64_t n, uint8_t out_xy_le[64]) {
  device_guard restore_callers_device;
  return run_common(ctx, points_xy_le, scalars_le, true, n, out_xy_le);
}

int te_msm_run_device(te_ctx* ctx, const void* d_points_xy_le, const void* d_scalars_le, uint64_t n, uint8_t out_xy_le[64]) {
  device_guard restore_callers_device;
  return run_common(ctx, d_points_xy_le, d_scalars_le, false, n, out_xy_le);
}

namespace {
// Tickets.  A ticket is a whole MSM in flight on ONE work set of ONE device of the context; its number is context-wide.
//   single-device context   the MSMs overlap on the device (own stream and buffers per work set)
//   D devices               a ticket goes to the device with the fewest in flight (ties: the device that holds the inputs, then
//                           round-robin): one whole MSM per device -- D PCIe links for host buffers, no replicated bucket
//                           reduction, no row merge -- the throughput form for a prover with several MSMs to do
//                           (ui/Benchmark.tsx:32 awaits an async call; nothing stops a caller from having several in flight;
//                           multi-device is the reference README's future work, README.md:551).  te_msm_run* stay the forms
//                           for the lone call (point slices / window shards over all devices).

// the lowest-numbered free work set of the device (each has its own stream: the MSMs overlap on the device).  Not ticket % sets:
// with fewer MSMs in flight than sets only as many sets as needed are ever touched -- no buffer allocation in the middle
// of a run, and a smaller footprint in the Infinity Cache.
// probe: the lazy hardware-queue measurement (16 ms, once per device) -- device-resident tickets only: a host-buffer ticket is bound
// by its upload, not by how the work sets' streams share the hardware queues
int take_free_workset(te_ctx* ctx, gpu_t& d, bool probe) {
  if (d.in_flight >= TE_MSM_WORKSETS) return set_err(ctx, TE_MSM_ESTATE, "every work set has an MSM in flight: collect one first");
  if (probe && !d.queues_probed && ctx->opt_queue_probe && d.in_flight == 0) spread_streams_over_queues(d);      // once per device, with nothing of it in flight
  // the work sets' streams can still be re-dealt by a later te_msm_submit_device as long as the measurement is on and has not run
  d.streams_final = d.queues_probed || !ctx->opt_queue_probe;
  const int wi = free_workset_index(d);
  if (wi < 0) return set_err(ctx, TE_MSM_ESTATE, "every work set has an MSM in flight: collect one first");
  return wi;
}
// the device the next ticket goes to (index into ctx->devs); prefer: the device that holds the inputs, or -1
// (the bookkeeping itself is host_sched.hpp's: the same code runs under ThreadSanitizer in tests/csrc/sched_harness.cpp)
int pick_device(te_ctx* ctx, int prefer) { return te_sched::pick_device_of(*ctx, TE_MSM_WORKSETS, prefer); }
void hand_out_ticket(te_ctx* ctx, int di, workset_t& ws, uint64_t* ticket, te_sched::job_ref job = nullptr) {
  te_sched::hand_out(*ctx, di, ws, ticket, std::move(job));          // te_msm_ticket_wait looks the ticket up from other threads
}
workset_t* workset_of_ticket(te_ctx* ctx, uint64_t ticket, gpu_t** dev = nullptr) {
  int di = -1;
  workset_t* ws = te_sched::find_ticket(*ctx, ticket, &di);
  if (ws && dev) *dev = &ctx->devs[(size_t)di];
  return ws;
}
// index into ctx->devs of the device whose memory holds p (device-resident inputs of a ticket), -1 if none of the context's
int device_index_of_pointer(te_ctx* ctx, const void* p) {
  hipPointerAttribute_t at; memset(&at, 0, sizeof at);
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
  if (at.type != hipMemoryTypeDevice) return -1;
  for (size_t i = 0; i < ctx->devs.size(); i++) if (ctx->devs[i].device == at.device) return (int)i;
  return -1;
}
const char* const kNoTicket = "no such ticket in flight (already collected, or never handed out)";
int fixed_base_settle(te_ctx* ctx, gpu_t& d, workset_t& ws, const te_bases* bases);     // (resident bases, further down)
}  // namespace

int te_msm_submit_device(te_ctx* ctx, const void* d_points_xy_le, const void* d_scalars_le, uint64_t n, uint64_t* ticket) {
  device_guard restore_callers_device;
  if (!ctx || !ticket) return TE_MSM_EINVAL;
  if (!d_points_xy_le || !d_scalars_le || n == 0 || n >= (1ull << 31)) return set_err(ctx, TE_MSM_EINVAL, "bad arguments");
  const bool multi = ctx->devs.size() > 1;
  // several devices: the inputs may live on any of them; the ticket goes to the least loaded one and pulls them over xGMI
  const int owner = multi ? device_index_of_pointer(ctx, d_points_xy_le) : 0;
  if (multi && (owner < 0 || device_index_of_pointer(ctx, d_scalars_le) != owner))
    return set_err(ctx, TE_MSM_EINVAL, "te_msm_submit_device: points and scalars must be resident on one device of the context");
  const int di = pick_device(ctx, owner);
  if (di < 0) return set_err(ctx, TE_MSM_ESTATE, "every work set has an MSM in flight: collect one first");
  gpu_t& d = ctx->devs[(size_t)di];
  HIP_TRY(ctx, hipSetDevice(d.device));
  const int wi = take_free_workset(ctx, d, true);
  if (wi < 0) return wi;
  workset_t& ws = d.ws[wi];
  const bool stage = multi && (d.device != ctx->devs[(size_t)owner].device || ctx->opt_stage_device_inputs);
  const int src_dev = ctx->devs[(size_t)owner].device;
  if (stage) { const curve_sizes sz = sizes_of(ctx->opt_curve); ctx->stat_peer_copies += 2; ctx->stat_peer_bytes += (int64_t)(n * (sz.point_in + sz.scalar_in)); }
  workset_t* wsp = &ws; gpu_t* dvp = &d;
  // the peer copies (if any), the ~10 launches of the MSM and its read-back
  auto work = [ctx, dvp, wsp, d_points_xy_le, d_scalars_le, n, stage, src_dev, multi]() -> int {
    gpu_t& d = *dvp; workset_t& ws = *wsp;
    HIP_TRY(ctx, hipSetDevice(d.device));
    const void *dp = d_points_xy_le, *ds = d_scalars_le;
    if (stage) {
      const curve_sizes sz = sizes_of(ctx->opt_curve);
      if (int rc = ensure_staging(ctx, ws, n * sz.point_in, n * sz.scalar_in)) return rc;
      if (ws.used) HIP_TRY(ctx, hipStreamWaitEvent(ws.stream, ws.ev_done, 0));
      HIP_TRY(ctx, hipMemcpyPeerAsync(ws.d_in_scalars, d.device, d_scalars_le, src_dev, n * sz.scalar_in, ws.stream));
      HIP_TRY(ctx, hipMemcpyPeerAsync(ws.d_in_points, d.device, d_points_xy_le, src_dev, n * sz.point_in, ws.stream));
      dp = ws.d_in_points; ds = ws.d_in_scalars;
    }
    // a single-device context keeps its window shard (te_msm_set_window_shard); on several devices a ticket is a whole MSM
    if (int rc = enqueue_partial(ctx, d, ws, dp, ds, n, nullptr, ws.stream, nullptr, 0, false, 1, multi, nullptr, !stage)) return rc;
    return fetch_rows(ctx, ws, ws.stream);
  };
  // (Handing this to a host thread, as te_msm_submit_async does with uploads, was measured: no gain at n = 2^16 .. 2^18, 3-7 %
  // slower at 2^19 / 2^20 -- the submitting thread is not what bounds small MSMs in flight; profiles/r05_enqueue_async_experiment.txt.)
  if (int rc = work()) return rc;
  hand_out_ticket(ctx, di, ws, ticket);
  return 0;
}

namespace {
// The first times SEVERAL host threads copy to one device at the same moment, the ROCm runtime brings further copy (SDMA)
// engines online -- each time a pause of ~7 ms for every thread that is copying to that device (they are released together;
// profiles/r05_point_shard_rehearsal.txt, r05_point_shard_sdma_experiment.txt: ~5 such pauses for eight threads, one for four,
// once per process).  With upload lanes that would fall into the first few asynchronous tickets of a process; instead the lanes
// of every device of the context copy 2 MB each in lockstep, twice, before the first asynchronous ticket is taken (15-30 ms the
// first time in a process, ~1 ms for a later context).  env TE_MSM_WARM_UPLOADS=0 turns it off.
void warm_upload_lanes(te_ctx* ctx) {
  static std::mutex mu; static uint64_t warmed = 0;           // per process and HIP device (ids < 64)
  const int L = ctx->opt_upload_threads;
  std::vector<size_t> todo;                                   // the context's devices whose lanes have not copied in lockstep yet: all of them at once
  {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t di = 0; di < ctx->devs.size(); di++) {
      const int dev = ctx->devs[di].device;
      if (dev < 64 && ((warmed >> dev) & 1u)) continue;
      if (dev < 64) warmed |= 1ull << dev;
      todo.push_back(di);
    }
  }
  if (const char* e = getenv("TE_MSM_WARM_UPLOADS")) if (e[0] == '0') return;
  if (L < 2 || todo.empty()) return;
  struct gate_t { std::mutex m; std::condition_variable cv; int waiting = 0, round = 0, parties = 0; } gate;
  gate.parties = L * (int)todo.size();
  auto arrive = [&gate]() {
    std::unique_lock<std::mutex> lk(gate.m);
    const int r = gate.round;
    if (++gate.waiting == gate.parties) { gate.waiting = 0; gate.round++; gate.cv.notify_all(); }
    else gate.cv.wait(lk, [&] { return gate.round != r; });
  };
  constexpr size_t SZ = 2u << 20;
  std::vector<te_sched::job_ref> jobs;
  std::vector<te_sched::worker_t*> who;
  for (size_t di : todo) {
    const int dev = ctx->devs[di].device;
    for (int l = 0; l < L; l++) {
      te_sched::worker_t& w = te_sched::next_lane_of(*ctx, di, L);
      who.push_back(&w);
      jobs.push_back(w.post([dev, &arrive]() -> int {
        hipStream_t st = nullptr; void* dbuf = nullptr;
        std::vector<uint8_t> h(SZ, 1);
        const bool ok = hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipMalloc(&dbuf, SZ) == hipSuccess;
        for (int r = 0; r < 2; r++) {
          arrive();                                           // every lane reaches the gate, whatever happened to its allocations
          if (ok) { (void)hipMemcpyAsync(dbuf, h.data(), SZ, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
        }
        if (dbuf) (void)hipFree(dbuf);
        if (st) (void)hipStreamDestroy(st);
        (void)hipGetLastError();
        return 0;
      }));
    }
  }
  for (size_t i = 0; i < jobs.size(); i++) (void)who[i]->wait(jobs[i]);
}

int submit_host(te_ctx* ctx, const uint8_t* points_xy_le, const uint8_t* scalars_le, uint64_t n, uint64_t* ticket, bool async) {
  if (!ctx || !ticket) return TE_MSM_EINVAL;
  if (!points_xy_le || !scalars_le || n == 0 || n >= (1ull << 31)) return set_err(ctx, TE_MSM_EINVAL, "bad arguments");
  if (ctx->devs.size() == 1 && ctx->devs[0].w_step != 1) return set_err(ctx, TE_MSM_ESTATE, "te_msm_submit computes whole MSMs: reset the window shard first");
  const int di = pick_device(ctx, -1);
  if (di < 0) return set_err(ctx, TE_MSM_ESTATE, "every work set has an MSM in flight: collect one first");
  gpu_t& d = ctx->devs[(size_t)di];
  HIP_TRY(ctx, hipSetDevice(d.device));
  const int wi = take_free_workset(ctx, d, false);
  if (wi < 0) return wi;
  workset_t& ws = d.ws[wi];
  plan_t pf; make_plan(ctx, d, n, pf, 0, 1, 0, true);
  const int c = pf.c, K = host_pieces(ctx, n);
  if (!async) {
    if (int rc = enqueue_host_slice(ctx, d, ws, points_xy_le, scalars_le, n, c, K, true)) return rc;
    hand_out_ticket(ctx, di, ws, ticket);
    return 0;
  }
  // the upload and the enqueue run on the device's host thread: this call returns at once, D of them keep D links busy.
  // The job reads the context's options as they are NOW only by accident of timing -- so te_msm_set_option waits for the host
  // threads first (drain_workers); the plan's window bits and pieces are fixed here.
  ws.job_err.clear();
  workset_t* wsp = &ws; gpu_t* dp = &d;
  warm_upload_lanes(ctx);                            // (once per process and device: all of the context's devices together)
  te_sched::job_ref job = te_sched::next_lane_of(*ctx, (size_t)di, ctx->opt_upload_threads).post([ctx, dp, wsp, points_xy_le, scalars_le, n, c, K]() -> int {
    const int rc = enqueue_host_slice(ctx, *dp, *wsp, points_xy_le, scalars_le, n, c, K, false);
    if (rc) { std::lock_guard<std::mutex> lk(ctx->err_mu); wsp->job_err = ctx->err; }
    return rc;
  });
  hand_out_ticket(ctx, di, ws, ticket, std::move(job));
  return 0;
}
// the enqueue of an asynchronous ticket has run (any thread); its status
int await_job(te_ctx*, gpu_t&, workset_t& ws) { return te_sched::await_job(ws); }
void retire_ticket(te_ctx* ctx, gpu_t& d, workset_t& ws) {
  if (ws.bound) { ws.bound->in_flight--; ws.bound = nullptr; }       // the ticket gathered from a bound point set: it may be released now
  release_shared_recs(d, ws);                                        // the ticket's MSM is over: its record slab may serve another point buffer
  te_sched::retire(*ctx, (int)(&d - ctx->devs.data()), ws);
}
}  // namespace

int te_msm_submit(te_ctx* ctx, const uint8_t* points_xy_le, const uint8_t* scalars_le, uint64_t n, uint64_t* ticket) {
  device_guard restore_callers_device;
  return submit_host(ctx, points_xy_le, scalars_le, n, ticket, false);
}

int te_msm_submit_async(te_ctx* ctx, const uint8_t* points_xy_le, const uint8_t* scalars_le, uint64_t n, uint64_t* ticket) {
  device_guard restore_callers_device;
  return submit_host(ctx, points_xy_le, scalars_le, n, ticket, true);
}

int te_msm_ticket_wait(te_ctx* ctx, uint64_t ticket) {
  device_guard restore_callers_device;
  if (!ctx) return TE_MSM_EINVAL;
  gpu_t* d = nullptr;
  workset_t* ws = workset_of_ticket(ctx, ticket, &d);
  if (!ws) return set_err(ctx, TE_MSM_ESTATE, kNoTicket);
  if (const int rc = await_job(ctx, *d, *ws)) return rc;      // te_msm_collect reports it (and frees the ticket)
  HIP_TRY(ctx, hipEventSynchronize(ws->ev_result));
  return 0;
}

int te_msm_ticket_device(te_ctx* ctx, uint64_t ticket, int* device_index, int* device_id) {
  if (!ctx) return TE_MSM_EINVAL;
  gpu_t* d = nullptr;
  if (!workset_of_ticket(ctx, ticket, &d)) return set_err(ctx, TE_MSM_ESTATE, kNoTicket);
  if (device_index) *device_index = (int)(d - ctx->devs.data());
  if (device_id) *device_id = d->device;
  return 0;
}

int te_msm_collect(te_ctx* ctx, uint64_t ticket, uint8_t out_xy_le[64]) {
  device_guard restore_callers_device;
  if (!ctx || !out_xy_le) return TE_MSM_EINVAL;
  gpu_t* dp = nullptr;
  workset_t* wsp = workset_of_ticket(ctx, ticket, &dp);
  if (!wsp) return set_err(ctx, TE_MSM_ESTATE, kNoTicket);
  workset_t& ws = *wsp; gpu_t& d = *dp;
  if (const int jrc = await_job(ctx, d, ws)) {
    // the upload / enqueue failed on the device's host thread: the ticket is over, the set must not keep half an MSM
    (void)hipSetDevice(d.device);
    (void)hipStreamSynchronize(ws.stream); if (ws.copy_stream) (void)hipStreamSynchronize(ws.copy_stream);
    ws.zero_clean_words = 0;
    retire_ticket(ctx, d, ws);
    return set_err(ctx, jrc, ws.job_err.empty() ? "the asynchronous submit failed" : ws.job_err.c_str());
  }
  HIP_TRY(ctx, hipEventSynchronize(ws.ev_result));   // on failure the ticket stays collectable
  if (ws.bound) { if (int rc = fixed_base_settle(ctx, d, ws, ws.bound)) return rc; }      // (a fixed-base ticket whose rows overflowed: run again with the ordinary windows)
  (void)collect_stage_ms(ctx, d, ws);
  note_entries(ctx, ws);
  retire_ticket(ctx, d, ws);                         // the MSM is over, with a result or with a scalar-range error
  if (*ws.h_err) return set_err(ctx, TE_MSM_ESCALAR, kFinalCarry);
  fold_rows(ws.plan, ws.h_partials, out_xy_le);
  return 0;
}

// ---- resident bases (include/te_msm.h): points bound once, scalars per call ---------------------------------------------------
namespace {
bool valid_bases(const te_ctx* ctx, const te_bases* b) { return b && std::find(ctx->bases.begin(), ctx->bases.end(), b) != ctx->bases.end(); }
const char* const kBadBases = "not a bound point set of this context (released, or bound to another context)";
const char* const kBasesCurve = "the point set was bound under another curve than the one selected now (option \"curve\")";

void free_bases(te_ctx* ctx, te_bases* b) {
  for (size_t i = 0; i < b->recs.size() && i < ctx->devs.size(); i++)
    if (b->recs[i]) { (void)hipSetDevice(ctx->devs[i].device); (void)hipFree(b->recs[i]); b->recs[i] = nullptr; }
}

// The records of the set on device i: raw points -> (temporary) -> records, on a stream of its own; returns when they are there.
// src on the host: the device's own upload (D devices: D links side by side, one host thread each); src on a device of the
// context: read in place on its holder, pulled over the peer link elsewhere.
int bind_on_device(te_ctx* ctx, te_bases* b, size_t i, const void* src, bool src_is_host, int src_dev) {
  gpu_t& d = ctx->devs[i];
  HIP_TRY(ctx, hipSetDevice(d.device));
  const curve_sizes sz = sizes_of(b->curve);
  const uint64_t n = b->n;
  uint8_t* recs = nullptr; void* raw = nullptr; void* proj = nullptr; hipStream_t st = nullptr;
  struct cleanup_t { void*& raw; void*& proj; hipStream_t& st; ~cleanup_t() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } if (raw) (void)hipFree(raw); if (proj) (void)hipFree(proj); } } cleanup{raw, proj, st};
  HIP_TRY(ctx, hipMalloc((void**)&recs, (size_t)n * b->rec_bytes * (size_t)b->replicas * (size_t)(b->fb_c ? b->fb_W : 1)));
  b->recs[i] = recs;                                                       // (freed by the caller on failure: free_bases)
  HIP_TRY(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const void* pts = src;
  if (src_is_host) {
    HIP_TRY(ctx, hipMalloc(&raw, (size_t)n * sz.point_in));
    HIP_TRY(ctx, hipMemcpyAsync(raw, src, (size_t)n * sz.point_in, hipMemcpyHostToDevice, st));
    pts = raw;
  } else if (src_dev != d.device || ctx->opt_stage_device_inputs) {
    HIP_TRY(ctx, hipMalloc(&raw, (size_t)n * sz.point_in));
    HIP_TRY(ctx, hipMemcpyPeerAsync(raw, d.device, src, src_dev, (size_t)n * sz.point_in, st));
    { std::lock_guard<std::mutex> lk(ctx->err_mu); ctx->stat_peer_copies += 1; ctx->stat_peer_bytes += (int64_t)(n * sz.point_in); }
    pts = raw;
  }
  const uint32_t n32 = (uint32_t)n;
  te::batch_ptrs tab; te::batch_slabs row_slab; memset(&tab, 0, sizeof tab); memset(&row_slab, 0, sizeof row_slab);
  tab.p[0] = (const uint4*)pts;
  if (b->curve == TE_MSM_CURVE_BLS12_377_G1) {
    te::rec_slot<14>* pr = reinterpret_cast<te::rec_slot<14>*>(recs);
    if (b->rec_kind == 1) { HIP_TRY(ctx, hipMalloc(&proj, (size_t)n * sizeof(te::rec_slot<14>))); pr = static_cast<te::rec_slot<14>*>(proj); }
    hipLaunchKernelGGL(te::k_prep_points377, dim3((n32 + 255) / 256, 1), dim3(256), 0, st, tab, row_slab, pr, n32);
    if (b->rec_kind == 1) {
      const uint32_t groups = (n32 + TE_AFF_GROUP - 1u) / TE_AFF_GROUP;
      hipLaunchKernelGGL(te::k_affine377, dim3((groups + 255) / 256), dim3(256), 0, st, pr, reinterpret_cast<te::rec_aff377*>(recs), n32);
    }
  } else {
    hipLaunchKernelGGL(te::k_prep_points, dim3((n32 + 255) / 256, 1), dim3(256), 0, st, tab, row_slab, reinterpret_cast<te::pnt_slot*>(recs), n32);
  }
  for (int r = 1; r < b->replicas; r++)
    HIP_TRY(ctx, hipMemcpyAsync(recs + (size_t)r * n * b->rec_bytes, recs, (size_t)n * b->rec_bytes, hipMemcpyDeviceToDevice, st));
  if (b->fb_c) {
    // fixed-base windows: table w = the records of 2^(c w) P_i; the extended points travel from window to window by c doublings
    HIP_TRY(ctx, hipMalloc(&proj, (size_t)n * sizeof(te::ete)));       // (the second temporary: unused on this curve otherwise)
    te::ete* ext = static_cast<te::ete*>(proj);
    const dim3 grid((n32 + 255) / 256), grid_g(((n32 + TE_AFF_GROUP - 1u) / TE_AFF_GROUP + 255) / 256);
    hipLaunchKernelGGL(te::k_fb_ext_from_recs, grid, dim3(256), 0, st, reinterpret_cast<const te::pnt_slot*>(recs), ext, n32);
    for (int w = 1; w < b->fb_W; w++) {
      hipLaunchKernelGGL(te::k_fb_double, grid, dim3(256), 0, st, ext, n32, (uint32_t)b->fb_c);
      hipLaunchKernelGGL(te::k_fb_records, grid_g, dim3(256), 0, st, ext, reinterpret_cast<te::pnt_slot*>(recs) + (size_t)w * n, n32);
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return 0;
}

int bind_common(te_ctx* ctx, const void* src, bool src_is_host, uint64_t n, te_bases** out) {
  if (!ctx || !out) return TE_MSM_EINVAL;
  *out = nullptr;
  if (n >= (1ull << 31)) return set_err(ctx, TE_MSM_EINVAL, "n must be < 2^31");
  if (n > 0 && !src) return set_err(ctx, TE_MSM_EINVAL, "null point buffer");
  drain_workers(ctx);
  const size_t nd = ctx->devs.size();
  int src_dev = -1;
  if (!src_is_host && n > 0) {
    const int owner = nd > 1 ? device_index_of_pointer(ctx, src) : 0;
    if (owner < 0) return set_err(ctx, TE_MSM_EINVAL, "te_msm_bind_points_device: the points must be resident on a device of the context");
    src_dev = ctx->devs[(size_t)owner].device;
  }
  te_bases* b = new te_bases();
  b->ctx = ctx; b->n = n; b->curve = ctx->opt_curve;
  b->rec_kind = (b->curve == TE_MSM_CURVE_BLS12_377_G1 && ctx->opt_bind_affine) ? 1 : 0;
  b->rec_bytes = rec_bytes_of(b->curve, b->rec_kind);
  b->replicas = ctx->opt_exp_table_replicas;
  if (ctx->opt_bind_fixed_base && b->curve == TE_MSM_CURVE_TE_BLS12 && n > 0) {
    b->fb_c = ctx->opt_bind_fixed_base; b->fb_W = fb_windows_for(b->fb_c); b->replicas = 1;
    if ((uint64_t)b->fb_W * n >= (1ull << 31)) { delete b; return set_err(ctx, TE_MSM_EINVAL, "bind_fixed_base: windows x points must stay below 2^31"); }
  }
  b->recs.assign(nd, nullptr);
  int rc = 0;
  if (n > 0) {
    std::vector<te_sched::job_ref> jobs(nd);
    for (size_t i = 1; i < nd; i++) jobs[i] = worker_of(ctx, i).post([=] { return bind_on_device(ctx, b, i, src, src_is_host, src_dev); });
    rc = bind_on_device(ctx, b, 0, src, src_is_host, src_dev);
    for (size_t i = 1; i < nd; i++) { const int r = worker_of(ctx, i).wait(jobs[i]); if (!rc) rc = r; }
  }
  if (rc) { free_bases(ctx, b); delete b; return rc; }
  // a set bound from HOST memory announces tickets from host scalars: the lanes' first concurrent copies (15-30 ms once per process,
  // warm_upload_lanes) happen here, in the call that blocks anyway, not in the first te_msm_submit_scalars -- which the N-API addon
  // issues on the JavaScript thread (js/promise_protocol.hpp: enter)
  if (src_is_host && n > 0) warm_upload_lanes(ctx);
  ctx->bases.push_back(b);
  *out = b;
  return 0;
}

// A whole MSM over a fixed-base set from HOST scalars on work set `ws`: all scalars cross PCIe in one copy (the digits of every
// window address one bucket set: there are no pieces to accumulate onto), then the fixed-base launch sequence and its read-back.
int enqueue_fixed_base_host(te_ctx* ctx, gpu_t& d, workset_t& ws, const te_bases* bases, const uint8_t* src_scalars, uint64_t n, bool wait_for_pinned) {
  HIP_TRY(ctx, hipSetDevice(d.device));
  const curve_sizes sz = sizes_of(bases->curve);
  if (int rc = ensure_staging(ctx, ws, 0, n * sz.scalar_in)) return rc;
  if (ws.used && ws.ev_done) HIP_TRY(ctx, hipStreamWaitEvent(ws.stream, ws.ev_done, 0));      // the staging area may still be read by the set's previous MSM
  if (!wait_for_pinned && ctx->opt_lane_host_waits) {
    // an asynchronous ticket: the lane thread waits for the upload itself (lane_wait), no wait enters the work set's stream
    if (ws.used && ws.ev_done) HIP_TRY(ctx, hipEventSynchronize(ws.ev_done));
    if (int rc = need_copy_stream(ctx, ws)) return rc;
    if (int rc = upload(ctx, ws, ws.d_in_scalars, src_scalars, n * sz.scalar_in, ws.copy_stream)) return rc;
    if (int rc = lane_wait(ctx, ws, ws.ev_copy, true)) return rc;
  } else {
    if (int rc = upload(ctx, ws, ws.d_in_scalars, src_scalars, n * sz.scalar_in, ws.stream)) return rc;
    if (wait_for_pinned && !ctx->opt_host_staging && host_memory_is_pinned(src_scalars)) {
      HIP_TRY(ctx, hipEventRecord(ws.ev_copy, ws.stream));
      HIP_TRY(ctx, hipEventSynchronize(ws.ev_copy));
    }
  }
  if (int rc = enqueue_fixed_base(ctx, d, ws, bases, ws.d_in_scalars, n, 0, ws.stream)) return rc;
  return fetch_rows(ctx, ws, ws.stream);
}

// The rows of a fixed-base MSM are on the host (ev_result passed).  A row of the pseudo-window buffers overflowed -- badly skewed
// scalars: all equal, a few distinct values -- : the MSM runs again with the ordinary windows over table 0 of the same set (the
// ordinary records), on the same work set; the caller then folds ws.plan's rows as usual.  0 = the rows in the pinned block are final.
int fixed_base_settle(te_ctx* ctx, gpu_t& d, workset_t& ws, const te_bases* bases) {
  if (!ws.plan.fb_rb || *ws.h_err || !ws.h_err[1]) return 0;
  ctx->stat_fb_fallbacks++;
  HIP_TRY(ctx, hipSetDevice(d.device));
  if (int rc = enqueue_partial(ctx, d, ws, nullptr, ws.fb_scalars, ws.fb_n, nullptr, ws.stream, nullptr, 0, false, 1, true, bases)) return rc;
  if (int rc = fetch_rows(ctx, ws, ws.stream)) return rc;
  HIP_TRY(ctx, hipEventSynchronize(ws.ev_result));
  return 0;
}

// te_msm_run_scalars_device on a context of D > 1 devices: WINDOW shards (device i computes windows i, i + D, ...).  Every
// device needs all n scalars: the holder reads them in place, the others pull them over their peer link (32 bytes per point:
// a third of what run_device_window_shards moves, so one copy each instead of its scatter + all-gather); every device gathers
// from its own copy of the bound records.  One host thread per device enqueues its copy, its share and its read-back.
int run_bound_window_shards(te_ctx* ctx, const te_bases* bases, const void* d_scalars, uint64_t n, uint8_t* out) {
  const size_t nd = ctx->devs.size();
  plan_t p0; make_plan(ctx, ctx->devs[0], n, p0);
  const curve_sizes sz = sizes_of(p0.curve);
  const int owner = device_index_of_pointer(ctx, d_scalars);
  if (owner < 0) return set_err(ctx, TE_MSM_EINVAL, "te_msm_run_scalars_device: the scalars must be resident on a device of the context");
  const int src_dev = ctx->devs[(size_t)owner].device;
  std::vector<int> wsel(nd, 0);
  for (size_t i = 0; i < nd; i++) { wsel[i] = free_workset_index(ctx->devs[i]); if (wsel[i] < 0) return set_err(ctx, TE_MSM_ESTATE, kAllSetsOwned); }
  std::vector<int64_t> copies(nd, 0);
  auto share = [&](size_t i) -> int {
    gpu_t& d = ctx->devs[i]; workset_t& ws = d.ws[wsel[i]];
    HIP_TRY(ctx, hipSetDevice(d.device));
    const void* ds = d_scalars;
    if (d.device != src_dev || ctx->opt_stage_device_inputs) {
      if (int rc = ensure_staging(ctx, ws, 0, n * sz.scalar_in)) return rc;
      if (ws.used && ws.ev_done) HIP_TRY(ctx, hipStreamWaitEvent(ws.stream, ws.ev_done, 0));
      HIP_TRY(ctx, hipMemcpyPeerAsync(ws.d_in_scalars, d.device, d_scalars, src_dev, n * sz.scalar_in, ws.stream));
      copies[i] = 1; ds = ws.d_in_scalars;
    }
    if (int rc = enqueue_partial(ctx, d, ws, nullptr, ds, n, nullptr, ws.stream, nullptr, 0, false, 1, false, bases)) return rc;
    return fetch_rows(ctx, ws, ws.stream);
  };
  std::vector<te_sched::job_ref> jobs(nd);
  for (size_t i = 1; i < nd; i++) jobs[i] = worker_of(ctx, i).post([&share, i] { return share(i); });
  int rc = share(0);
  for (size_t i = 1; i < nd; i++) { const int r = worker_of(ctx, i).wait(jobs[i]); if (!rc) rc = r; }
  for (size_t i = 0; i < nd; i++) { ctx->stat_peer_copies += copies[i]; ctx->stat_peer_bytes += copies[i] * (int64_t)(n * sz.scalar_in); }
  std::vector<uint8_t> merged((size_t)p0.W * sz.row, 0);
  int64_t entries = 0; bool carry = false;
  for (size_t i = 0; i < nd; i++) {
    gpu_t& d = ctx->devs[i]; workset_t& ws = d.ws[wsel[i]];
    if (rc) { (void)hipSetDevice(d.device); (void)hipStreamSynchronize(ws.stream); continue; }
    HIP_TRY(ctx, hipSetDevice(d.device));
    HIP_TRY(ctx, hipEventSynchronize(ws.ev_done));
    carry = carry || *ws.h_err != 0;
    entries += entries_of(ws);
    for (int w = d.w_first; w < p0.W; w += d.w_step) memcpy(&merged[(size_t)w * sz.row], ws.h_partials + (size_t)w * sz.row, sz.row);
  }
  if (rc) return rc;
  ctx->stat_entries = entries;
  if (carry) return set_err(ctx, TE_MSM_ESCALAR, kFinalCarry);
  if (p0.curve == TE_MSM_CURVE_BLS12_377_G1) te377_host::horner_to_affine(merged.data(), p0.c, (int)p0.logB, p0.W, out);
  else te_host::horner_to_affine(merged.data(), p0.c, (int)p0.logB, p0.W, out);
  return 0;
}

int run_scalars_common(te_ctx* ctx, te_bases* bases, const void* src, bool src_is_host, uint8_t* out) {
  if (!ctx || !out) return TE_MSM_EINVAL;
  if (!valid_bases(ctx, bases)) return set_err(ctx, TE_MSM_EINVAL, kBadBases);
  if (bases->curve != ctx->opt_curve) return set_err(ctx, TE_MSM_EINVAL, kBasesCurve);
  const uint64_t n = bases->n;
  if (n == 0) {
    memset(out, 0, sizes_of(ctx->opt_curve).result);
    if (ctx->opt_curve == TE_MSM_CURVE_TE_BLS12) out[32] = 1;
    return 0;
  }
  if (!src) return set_err(ctx, TE_MSM_EINVAL, "null scalar buffer");
  size_t di = 0;
  if (ctx->devs.size() > 1) {
    if (!bases->fb_c) {
      if (src_is_host) return run_host_sharded(ctx, nullptr, static_cast<const uint8_t*>(src), n, out, bases);
      return run_bound_window_shards(ctx, bases, src, n, out);
    }
    // a fixed-base set has no windows or point slices to shard (one bucket set): the lone call runs on ONE device -- the holder of
    // device-resident scalars, else the first; MSMs in flight (tickets) are how several devices work on such a set
    if (!src_is_host) { const int owner = device_index_of_pointer(ctx, src); if (owner < 0) return set_err(ctx, TE_MSM_EINVAL, "te_msm_run_scalars_device: the scalars must be resident on a device of the context"); di = (size_t)owner; }
  }
  gpu_t& d = ctx->devs[di];
  int wsel = ctx->devs.size() > 1 ? 0 : ctx->opt_workset;
  if (te_sched::slot_ticket(d.ws[wsel].slot)) {
    wsel = free_workset_index(d);
    if (wsel < 0) return set_err(ctx, TE_MSM_ESTATE, kAllSetsOwned);
  }
  workset_t& ws = d.ws[wsel];
  HIP_TRY(ctx, hipSetDevice(d.device));
  plan_t p0; make_plan(ctx, d, n, p0);
  const curve_sizes sz = sizes_of(p0.curve);
  const bool fb = bases->fb_c && (ctx->devs.size() > 1 || d.w_step == 1);      // (a window-sharded single-device context keeps the ordinary windows)
  if (fb) {
    if (src_is_host) { if (int rc = enqueue_fixed_base_host(ctx, d, ws, bases, static_cast<const uint8_t*>(src), n, true)) return rc; }
    else { if (int rc = enqueue_fixed_base(ctx, d, ws, bases, src, n, 0, ws.stream)) return rc; if (int rc = fetch_rows(ctx, ws, ws.stream)) return rc; }
  } else if (src_is_host && !ctx->opt_profile && d.w_step == 1) {
    // whole MSM, no stage timing: the scalars in pieces
    if (int rc = enqueue_scalar_slice(ctx, d, ws, bases->recs[di], bases->rec_kind, static_cast<const uint8_t*>(src), n, p0.c, scalar_pieces(ctx, n))) return rc;
  } else {
    const void* ds = src;
    if (src_is_host) {
      if (int rc = ensure_staging(ctx, ws, 0, n * sz.scalar_in)) return rc;
      if (ws.used && ws.ev_done) HIP_TRY(ctx, hipStreamWaitEvent(ws.stream, ws.ev_done, 0));
      if (int rc = upload(ctx, ws, ws.d_in_scalars, static_cast<const uint8_t*>(src), n * sz.scalar_in, ws.stream)) return rc;
      ds = ws.d_in_scalars;
    }
    if (int rc = enqueue_partial(ctx, d, ws, nullptr, ds, n, nullptr, ws.stream, nullptr, 0, false, 1, false, bases)) return rc;
    if (int rc = fetch_rows(ctx, ws, ws.stream)) return rc;
  }
  HIP_TRY(ctx, hipEventSynchronize(ws.ev_result));
  if (int rc = fixed_base_settle(ctx, d, ws, bases)) return rc;
  note_entries(ctx, ws);
  if (*ws.h_err) return set_err(ctx, TE_MSM_ESCALAR, kFinalCarry);
  (void)collect_stage_ms(ctx, d, ws);
  fold_rows(ws.plan, ws.h_partials, out);
  return 0;
}
}  // namespace

int te_msm_bind_points(te_ctx* ctx, const uint8_t* points_xy_le, uint64_t n, te_bases** out) {
  device_guard restore_callers_device;
  return bind_common(ctx, points_xy_le, true, n, out);
}

int te_msm_bind_points_device(te_ctx* ctx, const void* d_points_xy_le, uint64_t n, te_bases** out) {
  device_guard restore_callers_device;
  return bind_common(ctx, d_points_xy_le, false, n, out);
}

uint64_t te_msm_bases_count(const te_bases* bases) { return bases ? bases->n : 0; }

int te_msm_release_points(te_ctx* ctx, te_bases* bases) {
  device_guard restore_callers_device;
  if (!ctx) return TE_MSM_EINVAL;
  if (!valid_bases(ctx, bases)) return set_err(ctx, TE_MSM_EINVAL, kBadBases);
  if (bases->in_flight > 0) return set_err(ctx, TE_MSM_ESTATE, "te_msm_release_points: a ticket over this point set is in flight: collect it first");
  drain_workers(ctx);
  for (gpu_t& d : ctx->devs) {            // lone calls are over when they return; the streams may still clear their zeroed blocks -- nothing reads the records
    (void)hipSetDevice(d.device);
    for (workset_t& ws : d.ws) if (ws.used && ws.ev_done) (void)hipEventSynchronize(ws.ev_done);
  }
  free_bases(ctx, bases);
  ctx->bases.erase(std::find(ctx->bases.begin(), ctx->bases.end(), bases));
  delete bases;
  return 0;
}

int te_msm_run_scalars(te_ctx* ctx, te_bases* bases, const uint8_t* scalars_le, uint8_t out_xy_le[64]) {
  device_guard restore_callers_device;
  return run_scalars_common(ctx, bases, scalars_le, true, out_xy_le);
}

int te_msm_run_scalars_device(te_ctx* ctx, te_bases* bases, const void* d_scalars_le, uint8_t out_xy_le[64]) {
  device_guard restore_callers_device;
  return run_scalars_common(ctx, bases, d_scalars_le, false, out_xy_le);
}

int te_msm_submit_scalars(te_ctx* ctx, te_bases* bases, const uint8_t* scalars_le, uint64_t* ticket) {
  device_guard restore_callers_device;
  if (!ctx || !ticket) return TE_MSM_EINVAL;
  if (!valid_bases(ctx, bases)) return set_err(ctx, TE_MSM_EINVAL, kBadBases);
  if (bases->curve != ctx->opt_curve) return set_err(ctx, TE_MSM_EINVAL, kBasesCurve);
  const uint64_t n = bases->n;
  if (!scalars_le || n == 0) return set_err(ctx, TE_MSM_EINVAL, "bad arguments (an empty point set has no tickets: te_msm_run_scalars returns the identity)");
  if (ctx->devs.size() == 1 && ctx->devs[0].w_step != 1) return set_err(ctx, TE_MSM_ESTATE, "te_msm_submit_scalars computes whole MSMs: reset the window shard first");
  const int di = pick_device(ctx, -1);
  if (di < 0) return set_err(ctx, TE_MSM_ESTATE, "every work set has an MSM in flight: collect one first");
  gpu_t& d = ctx->devs[(size_t)di];
  HIP_TRY(ctx, hipSetDevice(d.device));
  const int wi = take_free_workset(ctx, d, false);
  if (wi < 0) return wi;
  workset_t& ws = d.ws[wi];
  plan_t pf; make_plan(ctx, d, n, pf, 0, 1, 0, true);
  // pieces shorten ONE MSM's way through the device (its upload hides under its own first pieces); with other tickets in flight on
  // the device the upload hides under THEIR device work, and one piece keeps the sort and the accumulation at full width
  const int c = pf.c, K = (ctx->opt_scalar_chunks || d.in_flight == 0) ? scalar_pieces(ctx, n) : 1;
  ws.job_err.clear();
  workset_t* wsp = &ws; gpu_t* dp = &d;
  const uint8_t* recs = bases->recs[(size_t)di]; const int kind = bases->rec_kind;
  warm_upload_lanes(ctx);
  te_sched::job_ref job = te_sched::next_lane_of(*ctx, (size_t)di, ctx->opt_upload_threads).post([ctx, dp, wsp, bases, recs, kind, scalars_le, n, c, K]() -> int {
    const int rc = bases->fb_c ? enqueue_fixed_base_host(ctx, *dp, *wsp, bases, scalars_le, n, false)
                               : enqueue_scalar_slice(ctx, *dp, *wsp, recs, kind, scalars_le, n, c, K, false);
    if (rc) { std::lock_guard<std::mutex> lk(ctx->err_mu); wsp->job_err = ctx->err; }
    return rc;
  });
  ws.bound = bases; bases->in_flight++;
  hand_out_ticket(ctx, di, ws, ticket, std::move(job));
  return 0;
}

int te_msm_submit_scalars_device(te_ctx* ctx, te_bases* bases, const void* d_scalars_le, uint64_t* ticket) {
  device_guard restore_callers_device;
  if (!ctx || !ticket) return TE_MSM_EINVAL;
  if (!valid_bases(ctx, bases)) return set_err(ctx, TE_MSM_EINVAL, kBadBases);
  if (bases->curve != ctx->opt_curve) return set_err(ctx, TE_MSM_EINVAL, kBasesCurve);
  const uint64_t n = bases->n;
  if (!d_scalars_le || n == 0) return set_err(ctx, TE_MSM_EINVAL, "bad arguments (an empty point set has no tickets: te_msm_run_scalars_device returns the identity)");
  const bool multi = ctx->devs.size() > 1;
  const int owner = multi ? device_index_of_pointer(ctx, d_scalars_le) : 0;
  if (multi && owner < 0) return set_err(ctx, TE_MSM_EINVAL, "te_msm_submit_scalars_device: the scalars must be resident on a device of the context");
  const int di = pick_device(ctx, owner);
  if (di < 0) return set_err(ctx, TE_MSM_ESTATE, "every work set has an MSM in flight: collect one first");
  gpu_t& d = ctx->devs[(size_t)di];
  HIP_TRY(ctx, hipSetDevice(d.device));
  const int wi = take_free_workset(ctx, d, true);
  if (wi < 0) return wi;
  workset_t& ws = d.ws[wi];
  const int src_dev = ctx->devs[(size_t)owner].device;
  const void* ds = d_scalars_le;
  if (multi && (d.device != src_dev || ctx->opt_stage_device_inputs)) {
    const curve_sizes sz = sizes_of(ctx->opt_curve);
    if (int rc = ensure_staging(ctx, ws, 0, n * sz.scalar_in)) return rc;
    if (ws.used) HIP_TRY(ctx, hipStreamWaitEvent(ws.stream, ws.ev_done, 0));
    HIP_TRY(ctx, hipMemcpyPeerAsync(ws.d_in_scalars, d.device, d_scalars_le, src_dev, n * sz.scalar_in, ws.stream));
    ctx->stat_peer_copies += 1; ctx->stat_peer_bytes += (int64_t)(n * sz.scalar_in);
    ds = ws.d_in_scalars;
  }
  if (bases->fb_c && (multi || d.w_step == 1)) { if (int rc = enqueue_fixed_base(ctx, d, ws, bases, ds, n, 0, ws.stream)) return rc; }
  else if (int rc = enqueue_partial(ctx, d, ws, nullptr, ds, n, nullptr, ws.stream, nullptr, 0, false, 1, multi, bases)) return rc;
  if (int rc = fetch_rows(ctx, ws, ws.stream)) return rc;
  ws.bound = bases; bases->in_flight++;
  hand_out_ticket(ctx, di, ws, ticket);
  return 0;
}

int te_msm_probe_queues(te_ctx* ctx) {
  device_guard restore_callers_device;
  if (!ctx) return TE_MSM_EINVAL;
  drain_workers(ctx);
  int classes = 0;
  for (gpu_t& d : ctx->devs) {
    if (d.in_flight) return set_err(ctx, TE_MSM_ESTATE, "te_msm_probe_queues: collect the MSMs in flight first");
    HIP_TRY(ctx, hipSetDevice(d.device));
    spread_streams_over_queues(d);
    d.streams_final = true;
    int c = -1; for (const workset_t& ws : d.ws) c = std::max(c, ws.hw_queue_class);
    classes = std::max(classes, c + 1);
  }
  return classes;          // 0: the measurement was not accepted (its two passes disagreed): creation order kept
}

int te_msm_trim(te_ctx* ctx, int keep_worksets) {
  device_guard restore_callers_device;
  if (!ctx || keep_worksets < 0) return TE_MSM_EINVAL;
  drain_workers(ctx);
  int freed = 0;
  for (gpu_t& d : ctx->devs) {
    HIP_TRY(ctx, hipSetDevice(d.device));
    for (int i = keep_worksets; i < TE_MSM_WORKSETS; i++) {
      workset_t& ws = d.ws[i];
      if (te_sched::slot_ticket(ws.slot) || !(ws.d_recs || ws.d_in_points || ws.d_zero)) continue;       // owned by a ticket, or nothing to free
      if (ws.used) HIP_TRY(ctx, hipEventSynchronize(ws.ev_done));
      if (ws.copy_stream) HIP_TRY(ctx, hipStreamSynchronize(ws.copy_stream));
      free_workset_buffers(ws);
      freed++;
    }
    for (auto& sl : d.slabs) if (sl.users == 0 && sl.d) {                    // idle shared record slabs
      if (int rc = settle_slab(ctx, sl, nullptr, true)) return rc;
      for (workset_t& ws : d.ws) if (ws.recs_last == sl.d) ws.recs_last = nullptr;
      HIP_TRY(ctx, hipFree(sl.d)); sl.d = nullptr; sl.cap = 0; sl.src = nullptr;
    }
  }
  return freed;
}

int te_msm_set_option(te_ctx* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return TE_MSM_EINVAL;
  drain_workers(ctx);               // asynchronous submits read the options on the devices' host threads
  if (!strcmp(key, "window_bits")) { if (value != 0 && (value < 4 || value > 16)) return set_err(ctx, TE_MSM_EINVAL, "window_bits must be 0 or in [4,16]"); ctx->opt_window_bits = (int)value; return 0; }
  if (!strcmp(key, "sort_buckets")) { ctx->opt_sort = value ? 1 : 0; return 0; }
  if (!strcmp(key, "signed_digits")) { ctx->opt_signed = value ? 1 : 0; return 0; }
  if (!strcmp(key, "curve")) {
    if (value != TE_MSM_CURVE_TE_BLS12 && value != TE_MSM_CURVE_BLS12_377_G1) return set_err(ctx, TE_MSM_EINVAL, "unknown curve");
    ctx->opt_curve = (int)value; return 0;
  }
  if (!strcmp(key, "profile")) { ctx->opt_profile = value < 0 ? 0 : (value > 2 ? 2 : (int)value); ctx->have_stage_ms = false; return 0; }
  if (!strcmp(key, "graph")) { ctx->opt_graph = value ? 1 : 0; return 0; }
  if (!strcmp(key, "prezero")) { ctx->opt_prezero = value ? 1 : 0; return 0; }
  if (!strcmp(key, "fuse_prep")) { ctx->opt_fuse_prep = value ? 1 : 0; return 0; }
  if (!strcmp(key, "host_chunks")) { if (value < 0 || value > 64) return set_err(ctx, TE_MSM_EINVAL, "host_chunks out of range"); ctx->opt_host_chunks = (int)value; return 0; }
  if (!strcmp(key, "workset")) { if (value < 0 || value >= TE_MSM_WORKSETS) return set_err(ctx, TE_MSM_EINVAL, "workset out of range"); ctx->opt_workset = (int)value; return 0; }
  if (!strcmp(key, "segment_len")) { if (value < 0 || value > 1000000) return set_err(ctx, TE_MSM_EINVAL, "segment_len must be 0 (from n) or in [1, 1e6]"); ctx->opt_seg_len = (int)value; return 0; }
  if (!strcmp(key, "host_shard_min")) { if (value < 1 || value > (1ll << 30)) return set_err(ctx, TE_MSM_EINVAL, "host_shard_min out of range"); ctx->opt_host_shard_min = (int)value; return 0; }
  if (!strcmp(key, "queue_probe")) { ctx->opt_queue_probe = value ? 1 : 0; return 0; }
  if (!strcmp(key, "packed_sort")) { ctx->opt_packed = value ? 1 : 0; return 0; }
  if (!strcmp(key, "fold_pairs")) { ctx->opt_fold_pairs = value ? 1 : 0; return 0; }
  if (!strcmp(key, "stage_device_inputs")) { ctx->opt_stage_device_inputs = value ? 1 : 0; return 0; }
  if (!strcmp(key, "host_staging")) { ctx->opt_host_staging = value ? 1 : 0; return 0; }
  if (!strcmp(key, "upload_threads")) { if (value < 1 || value > 16) return set_err(ctx, TE_MSM_EINVAL, "upload_threads must be in [1, 16]"); ctx->opt_upload_threads = (int)value; return 0; }
  if (!strcmp(key, "bind_affine")) { ctx->opt_bind_affine = value ? 1 : 0; return 0; }
  if (!strcmp(key, "lane_host_waits")) { ctx->opt_lane_host_waits = value ? 1 : 0; return 0; }
  if (!strcmp(key, "share_records")) { ctx->opt_share_records = value ? 1 : 0; return 0; }
  if (!strcmp(key, "bind_fixed_base")) { if (value != 0 && (value < 16 || value > 21)) return set_err(ctx, TE_MSM_EINVAL, "bind_fixed_base must be 0 or in [16, 21]"); ctx->opt_bind_fixed_base = (int)value; return 0; }
  if (!strcmp(key, "exp_table_replicas")) { if (value < 1 || value > TE_BATCH_MAX) return set_err(ctx, TE_MSM_EINVAL, "exp_table_replicas must be in [1, 8]"); ctx->opt_exp_table_replicas = (int)value; return 0; }
  if (!strcmp(key, "scalar_chunks")) { if (value < 0 || value > 64) return set_err(ctx, TE_MSM_EINVAL, "scalar_chunks out of range"); ctx->opt_scalar_chunks = (int)value; return 0; }
  return set_err(ctx, TE_MSM_EINVAL, "unknown option");
}

int te_msm_get_option(te_ctx* ctx, const char* key, int64_t* value) {
  if (!ctx || !key || !value) return TE_MSM_EINVAL;
  if (!strcmp(key, "window_bits")) { *value = ctx->opt_window_bits; return 0; }
  if (!strcmp(key, "sort_buckets")) { *value = ctx->opt_sort; return 0; }
  if (!strcmp(key, "signed_digits")) { *value = ctx->opt_signed; return 0; }
  if (!strcmp(key, "curve")) { *value = ctx->opt_curve; return 0; }
  if (!strcmp(key, "profile")) { *value = ctx->opt_profile; return 0; }
  if (!strcmp(key, "num_devices")) { *value = (int64_t)ctx->devs.size(); return 0; }
  if (!strcmp(key, "peer_copies")) { *value = ctx->stat_peer_copies; return 0; }
  if (!strcmp(key, "peer_bytes")) { *value = ctx->stat_peer_bytes; return 0; }
  if (!strcmp(key, "entries_accumulated")) { *value = ctx->stat_entries; return 0; }
  if (!strcmp(key, "stage_device_inputs")) { *value = ctx->opt_stage_device_inputs; return 0; }
  if (!strcmp(key, "host_staging")) { *value = ctx->opt_host_staging; return 0; }
  if (!strcmp(key, "upload_threads")) { *value = ctx->opt_upload_threads; return 0; }
  if (!strcmp(key, "segment_len")) { *value = ctx->opt_seg_len; return 0; }
  if (!strcmp(key, "segment_len_used")) { drain_workers(ctx); const gpu_t& d0 = ctx->devs[0]; *value = d0.ws[d0.last_ws].used ? (int64_t)d0.ws[d0.last_ws].plan.seg_len : 0; return 0; }
  if (!strcmp(key, "workset")) { *value = ctx->opt_workset; return 0; }
  if (!strcmp(key, "graph")) { *value = ctx->opt_graph; return 0; }
  if (!strcmp(key, "prezero")) { *value = ctx->opt_prezero; return 0; }
  if (!strcmp(key, "fuse_prep")) { *value = ctx->opt_fuse_prep; return 0; }
  if (!strcmp(key, "host_chunks")) { *value = ctx->opt_host_chunks; return 0; }
  if (!strcmp(key, "host_shard_min")) { *value = ctx->opt_host_shard_min; return 0; }
  if (!strcmp(key, "queue_probe")) { *value = ctx->opt_queue_probe; return 0; }
  if (!strcmp(key, "packed_sort")) { *value = ctx->opt_packed; return 0; }
  if (!strcmp(key, "fold_pairs")) { *value = ctx->opt_fold_pairs; return 0; }
  if (!strcmp(key, "streams_final")) { *value = ctx->devs[0].streams_final ? 1 : 0; return 0; }
  if (!strcmp(key, "bind_affine")) { *value = ctx->opt_bind_affine; return 0; }
  if (!strcmp(key, "lane_host_waits")) { *value = ctx->opt_lane_host_waits; return 0; }
  if (!strcmp(key, "share_records")) { *value = ctx->opt_share_records; return 0; }
  if (!strcmp(key, "record_slabs")) { int64_t t = 0; for (const gpu_t& d : ctx->devs) for (const auto& sl : d.slabs) if (sl.d) t++; *value = t; return 0; }
  if (!strcmp(key, "scalar_chunks")) { *value = ctx->opt_scalar_chunks; return 0; }
  if (!strcmp(key, "bind_fixed_base")) { *value = ctx->opt_bind_fixed_base; return 0; }
  if (!strcmp(key, "fixed_base_fallbacks")) { *value = ctx->stat_fb_fallbacks; return 0; }
  if (!strcmp(key, "bases_bound")) { *value = (int64_t)ctx->bases.size(); return 0; }
  if (!strcmp(key, "bases_bytes")) { int64_t t = 0; for (const te_bases* b : ctx->bases) for (const uint8_t* r : b->recs) if (r) t += (int64_t)(b->n * b->rec_bytes) * b->replicas * (b->fb_c ? b->fb_W : 1); *value = t; return 0; }
  if (!strcmp(key, "in_flight")) { int64_t t = 0; for (const gpu_t& d : ctx->devs) t += d.in_flight; *value = t; return 0; }
  if (!strcmp(key, "device_bytes")) {      drain_workers(ctx);      // device memory this context holds in work-set buffers (te_msm_trim gives it back)
    int64_t tot = 0;
    for (const gpu_t& d : ctx->devs) {
      for (const workset_t& ws : d.ws) { for (size_t cb : ws.cap) tot += (int64_t)cb; tot += (int64_t)(ws.cap_in_points + ws.cap_in_scalars); }
      for (const auto& sl : d.slabs) if (sl.d) tot += (int64_t)sl.cap;
    }
    *value = tot; return 0;
  }
  return set_err(ctx, TE_MSM_EINVAL, "unknown option");
}

int te_msm_set_window_shard(te_ctx* ctx, int first, int step) {
  if (!ctx) return TE_MSM_EINVAL;
  if (ctx->devs.size() != 1) return set_err(ctx, TE_MSM_ESTATE, "window shards are set automatically for multi-device contexts");
  if (first < 0 || step < 1) return set_err(ctx, TE_MSM_EINVAL, "need first >= 0 and step >= 1");
  drain_workers(ctx);               // asynchronous enqueues read the shard on the device's host threads
  ctx->devs[0].w_first = first; ctx->devs[0].w_step = step;
  return 0;
}

int te_msm_plan(te_ctx* ctx, uint64_t n, int* window_bits, int* num_windows) {
  if (!ctx) return TE_MSM_EINVAL;
  plan_t p; make_plan(ctx, ctx->devs[0], n ? n : 1, p);
  if (window_bits) *window_bits = p.c;
  if (num_windows) *num_windows = p.W;
  return 0;
}

int te_msm_partial_device(te_ctx* ctx, const void* d_points_xy_le, const void* d_scalars_le, uint64_t n, void* d_partials, void* stream) {
  device_guard restore_callers_device;
  if (!ctx) return TE_MSM_EINVAL;
  if (ctx->devs.size() != 1) return set_err(ctx, TE_MSM_ESTATE, "te_msm_partial_device needs a single-device context");
  if (!d_points_xy_le || !d_scalars_le || !d_partials || n == 0 || n >= (1ull << 31)) return set_err(ctx, TE_MSM_EINVAL, "bad arguments");
  gpu_t& d = ctx->devs[0];
  workset_t& ws = d.ws[ctx->opt_workset];
  if (te_sched::slot_ticket(ws.slot)) return set_err(ctx, TE_MSM_ESTATE, "the selected work set holds a submitted MSM that has not been collected");
  HIP_TRY(ctx, hipSetDevice(d.device));
  return enqueue_partial(ctx, d, ws, d_points_xy_le, d_scalars_le, n, d_partials, stream == TE_MSM_OWN_STREAM ? ws.stream : (hipStream_t)stream,
                         nullptr, 0, false, 1, false, nullptr, true);
}

int te_msm_partial_device_batch(te_ctx* ctx, const void* const* d_points_xy_le, const void* const* d_scalars_le, uint64_t n, int count,
                                void* d_partials, void* stream) {
  device_guard restore_callers_device;
  if (!ctx) return TE_MSM_EINVAL;
  if (ctx->devs.size() != 1) return set_err(ctx, TE_MSM_ESTATE, "te_msm_partial_device_batch needs a single-device context");
  if (!d_points_xy_le || !d_scalars_le || !d_partials || n == 0 || n >= (1ull << 31) || count < 1 || count > TE_MSM_MAX_BATCH)
    return set_err(ctx, TE_MSM_EINVAL, "bad arguments");
  for (int m = 0; m < count; m++) if (!d_points_xy_le[m] || !d_scalars_le[m]) return set_err(ctx, TE_MSM_EINVAL, "bad arguments");
  gpu_t& d = ctx->devs[0];
  workset_t& ws = d.ws[ctx->opt_workset];
  if (te_sched::slot_ticket(ws.slot)) return set_err(ctx, TE_MSM_ESTATE, "the selected work set holds a submitted MSM that has not been collected");
  HIP_TRY(ctx, hipSetDevice(d.device));
  hipStream_t st = stream == TE_MSM_OWN_STREAM ? ws.stream : (hipStream_t)stream;
  if (count == 1) return enqueue_partial(ctx, d, ws, d_points_xy_le[0], d_scalars_le[0], n, d_partials, st, nullptr, 0, false, 1, false, nullptr, true);
  // the pointer arrays are only read while the launches are enqueued
  return enqueue_partial(ctx, d, ws, d_points_xy_le, d_scalars_le, n, d_partials, st, nullptr, 0, false, count, false, nullptr, true);
}

int te_msm_workset_stream(te_ctx* ctx, int workset, void** stream, int* hw_queue_class) {
  if (!ctx || workset < 0 || workset >= TE_MSM_WORKSETS || !stream) return TE_MSM_EINVAL;
  if (ctx->devs.size() != 1) return set_err(ctx, TE_MSM_ESTATE, "te_msm_workset_stream needs a single-device context");
  const workset_t& ws = ctx->devs[0].ws[workset];
  ctx->devs[0].streams_exported = true;       // the handle leaves the library: the device's streams are parked, not destroyed, from now on
  *stream = (void*)ws.stream;
  if (hw_queue_class) *hw_queue_class = ws.hw_queue_class;
  return 0;
}

int te_msm_partial_wait(te_ctx* ctx, int workset) {
  device_guard restore_callers_device;
  if (!ctx || workset < 0 || workset >= TE_MSM_WORKSETS) return TE_MSM_EINVAL;
  if (ctx->devs.size() != 1) return set_err(ctx, TE_MSM_ESTATE, "te_msm_partial_wait needs a single-device context");
  drain_workers(ctx);
  workset_t& ws = ctx->devs[0].ws[workset];
  if (!ws.used) return 0;
  HIP_TRY(ctx, hipEventSynchronize(ws.ev_result));
  (void)collect_stage_ms(ctx, ctx->devs[0], ws);     // stage times of that launch sequence, when it was profiled (te_msm_stage_ms)
  note_entries(ctx, ws);
  if (*ws.h_err) return set_err(ctx, TE_MSM_ESCALAR, "final carry is 1: a scalar does not fit the signed window decomposition");
  return 0;
}

int te_msm_finalize(te_ctx* ctx, const uint8_t* partials, int window_bits, int num_windows, uint8_t out_xy_le[64]) {
  device_guard restore_callers_device;
  if (!ctx || !partials || !out_xy_le || window_bits < 2 || window_bits > 16 || num_windows < 1 || num_windows > 128) return TE_MSM_EINVAL;
  gpu_t& d = ctx->devs[0];
  workset_t& ws = d.ws[d.last_ws];
  int bucket_bits = ctx->opt_signed ? window_bits - 1 : window_bits;
  if (ws.used) {
    HIP_TRY(ctx, hipEventSynchronize(ws.ev_result));
    if (*ws.h_err) return set_err(ctx, TE_MSM_ESCALAR, "final carry is 1: a scalar does not fit the signed window decomposition");
    (void)collect_stage_ms(ctx, d, ws);
    note_entries(ctx, ws);
    // the rows were produced under ws.plan: its digit form decides, not an option changed since
    if (window_bits != ws.plan.c || num_windows != ws.plan.W)
      return set_err(ctx, TE_MSM_ESTATE, "te_msm_finalize: window_bits / num_windows differ from the plan of the last te_msm_partial_device call (use te_msm_finalize_host_ex for rows produced elsewhere)");
    bucket_bits = (int)ws.plan.logB;
    if (ws.plan.curve == TE_MSM_CURVE_BLS12_377_G1) { te377_host::horner_to_affine(partials, window_bits, bucket_bits, num_windows, out_xy_le); return 0; }
  } else if (ctx->opt_curve == TE_MSM_CURVE_BLS12_377_G1) {
    te377_host::horner_to_affine(partials, window_bits, bucket_bits, num_windows, out_xy_le); return 0;
  }
  te_host::horner_to_affine(partials, window_bits, bucket_bits, num_windows, out_xy_le);
  return 0;
}

int te_msm_host_tail_features(void) {
  int f = te_host::have_adx() ? 1 : 0;
#if defined(__x86_64__)
  if (te_host::have_ifma()) f |= 2;
#endif
  return f;
}

int te_msm_finalize_host(const uint8_t* partials, int window_bits, int num_windows, uint8_t out_xy_le[64]) {
  if (!partials || !out_xy_le || window_bits < 2 || window_bits > 16 || num_windows < 1 || num_windows > 128) return TE_MSM_EINVAL;
  return te_msm_finalize_host_ex(partials, window_bits, window_bits - 1, num_windows, out_xy_le);
}

int te_msm_finalize_host_ex(const uint8_t* partials, int window_bits, int bucket_bits, int num_windows, uint8_t out_xy_le[64]) {
  if (!partials || !out_xy_le || window_bits < 2 || window_bits > 16 || num_windows < 1 || num_windows > 128) return TE_MSM_EINVAL;
  if (bucket_bits != window_bits && bucket_bits != window_bits - 1) return TE_MSM_EINVAL;
  if (!te_host::tail_selftest()) return TE_MSM_ESTATE;
  te_host::horner_to_affine(partials, window_bits, bucket_bits, num_windows, out_xy_le);
  return 0;
}

int te_msm_finalize_gathered(const uint8_t* gathered, int world, int window_bits, int bucket_bits, int num_windows,
                             uint8_t out_xy_le[64]) {
  if (!gathered || world < 1 || num_windows < 1 || num_windows > 128) return TE_MSM_EINVAL;
  std::vector<uint8_t> merged((size_t)num_windows * TE_MSM_PARTIAL_BYTES);
  for (int w = 0; w < num_windows; w++)
    memcpy(&merged[(size_t)w * TE_MSM_PARTIAL_BYTES],
           gathered + ((size_t)(w % world) * num_windows + w) * TE_MSM_PARTIAL_BYTES, TE_MSM_PARTIAL_BYTES);
  return te_msm_finalize_host_ex(merged.data(), window_bits, bucket_bits, num_windows, out_xy_le);
}

int te_msm_finalize_host_curve(int curve, const uint8_t* partials, int window_bits, int bucket_bits, int num_windows, uint8_t* out_xy_le) {
  if (curve == TE_MSM_CURVE_TE_BLS12) return te_msm_finalize_host_ex(partials, window_bits, bucket_bits, num_windows, out_xy_le);
  if (curve != TE_MSM_CURVE_BLS12_377_G1) return TE_MSM_EINVAL;
  if (!partials || !out_xy_le || window_bits < 2 || window_bits > 16 || num_windows < 1 || num_windows > 128) return TE_MSM_EINVAL;
  if (bucket_bits != window_bits && bucket_bits != window_bits - 1) return TE_MSM_EINVAL;
  if (!te377_host::tail_selftest()) return TE_MSM_ESTATE;
  te377_host::horner_to_affine(partials, window_bits, bucket_bits, num_windows, out_xy_le);
  return 0;
}

int te_msm_finalize_gathered_curve(int curve, const uint8_t* gathered, int world, int window_bits, int bucket_bits, int num_windows,
                                   uint8_t* out_xy_le) {
  if (!gathered || world < 1 || num_windows < 1 || num_windows > 128) return TE_MSM_EINVAL;
  if (curve != TE_MSM_CURVE_TE_BLS12 && curve != TE_MSM_CURVE_BLS12_377_G1) return TE_MSM_EINVAL;
  const size_t row = sizes_of(curve).row;
  std::vector<uint8_t> merged((size_t)num_windows * row);
  for (int w = 0; w < num_windows; w++)
    memcpy(&merged[(size_t)w * row], gathered + ((size_t)(w % world) * num_windows + w) * row, row);
  return te_msm_finalize_host_curve(curve, merged.data(), window_bits, bucket_bits, num_windows, out_xy_le);
}

int te_msm_finalize_sum_curve(int curve, const uint8_t* const* row_sets, int sets, int window_bits, int bucket_bits, int num_windows, uint8_t* out_xy_le) {
  if (curve != TE_MSM_CURVE_TE_BLS12 && curve != TE_MSM_CURVE_BLS12_377_G1) return TE_MSM_EINVAL;
  if (!row_sets || sets < 1 || sets > 4096 || !out_xy_le || window_bits < 2 || window_bits > 16 || num_windows < 1 || num_windows > 128) return TE_MSM_EINVAL;
  if (bucket_bits != window_bits && bucket_bits != window_bits - 1) return TE_MSM_EINVAL;
  for (int i = 0; i < sets; i++) if (!row_sets[i]) return TE_MSM_EINVAL;
  // three or more sets: merged window by window first, then folded (the two-step form the multi-device te_msm_run spreads over
  // its host threads); one or two: summed on the fly
  std::vector<uint8_t> present((size_t)num_windows, 0);
  if (curve == TE_MSM_CURVE_BLS12_377_G1) {
    if (!te377_host::tail_selftest()) return TE_MSM_ESTATE;
    if (sets <= 2) { te377_host::horner_to_affine_multi(row_sets, sets, window_bits, bucket_bits, num_windows, out_xy_le); return 0; }
    std::vector<te377_host::Pt> m((size_t)num_windows * 5);
    for (int w = 0; w < num_windows; w++) te377_host::merge_window_rows(row_sets, sets, w, m.data(), present.data());
    te377_host::horner_to_affine_points(m.data(), present.data(), window_bits, bucket_bits, num_windows, out_xy_le);
  } else {
    if (!te_host::tail_selftest()) return TE_MSM_ESTATE;
    if (sets <= 2) { te_host::horner_to_affine_multi(row_sets, sets, window_bits, bucket_bits, num_windows, out_xy_le); return 0; }
    std::vector<te_host::Pt> m((size_t)num_windows * 5);
    for (int w = 0; w < num_windows; w++) te_host::merge_window_rows(row_sets, sets, w, m.data(), present.data());
    te_host::horner_to_affine_points(m.data(), present.data(), window_bits, bucket_bits, num_windows, out_xy_le);
  }
  return 0;
}

int te_msm_stage_ms(te_ctx* ctx, float* ms, const char** names, int max_stages) {
  if (!ctx || !ms) return TE_MSM_EINVAL;
  if (!ctx->have_stage_ms) return set_err(ctx, TE_MSM_ESTATE, "no profiled run yet (set option profile=1)");
  int k = 0;
  for (int i = 0; i < ST_COUNT + 2 && k < max_stages; i++) {
    if (ctx->stage_ms[i] < 0) continue;            // not measured at this profile level
    ms[k] = ctx->stage_ms[i]; if (names) names[k] = kStageNames[i]; k++;
  }
  return k;
}

int te_msm_synth_inputs(uint64_t seed, uint64_t n, int fixed_point, uint8_t* points_xy_le, uint8_t* scalars_le) {
  if (n >= (1ull << 31) || !te_host::tail_selftest()) return TE_MSM_EINVAL;
  if (scalars_le) te_host::synth_scalars(seed, n, scalars_le);
  if (points_xy_le) {
    if (fixed_point == 1) te_host::synth_points_fixed(n, points_xy_le);
    else if (fixed_point == 2) te_host::synth_points_random(seed, n, points_xy_le);
    else if (fixed_point == 0) te_host::synth_points(seed, n, points_xy_le);
    else return TE_MSM_EINVAL;
  }
  return 0;
}

int te_msm_synth_inputs_bls12_377(uint64_t seed, uint64_t n, uint8_t* points_xy_le, uint8_t* scalars_le) {
  if (n >= (1ull << 31) || !te377_host::tail_selftest()) return TE_MSM_EINVAL;
  if (scalars_le) te377_host::synth_scalars(seed, n, scalars_le);
  if (points_xy_le) te377_host::synth_points(seed, n, points_xy_le);
  return 0;
}

int64_t te_msm_bases_read(te_ctx* ctx, const te_bases* bases, int device_index, uint64_t first, uint64_t count, void* dst, uint64_t cap, int* record_bytes) {
  device_guard restore_callers_device;
  if (!ctx || !dst) return TE_MSM_EINVAL;
  if (!valid_bases(ctx, bases)) return set_err(ctx, TE_MSM_EINVAL, kBadBases);
  const uint64_t total = bases->n * (uint64_t)bases->replicas * (uint64_t)(bases->fb_c ? bases->fb_W : 1);      // (a fixed-base set: table w at records [w n, (w + 1) n))
  if (device_index < 0 || (size_t)device_index >= ctx->devs.size() || first > total || count > total - first) return set_err(ctx, TE_MSM_EINVAL, "bad arguments");
  if (record_bytes) *record_bytes = (int)bases->rec_bytes;
  uint64_t bytes = count * bases->rec_bytes;
  if (bytes > cap) bytes = cap;
  if (!bytes) return 0;
  HIP_TRY(ctx, hipSetDevice(ctx->devs[(size_t)device_index].device));
  HIP_TRY(ctx, hipMemcpy(dst, bases->recs[(size_t)device_index] + first * bases->rec_bytes, bytes, hipMemcpyDeviceToHost));
  return (int64_t)bytes;
}

int64_t te_msm_debug_read(te_ctx* ctx, const char* stage, void* dst, uint64_t cap) {
  device_guard restore_callers_device;
  if (!ctx || !stage || !dst) return TE_MSM_EINVAL;
  drain_workers(ctx);
  gpu_t& d = ctx->devs[0];
  workset_t& ws = d.ws[d.last_ws];
  if (!ws.used) return set_err(ctx, TE_MSM_ESTATE, "no run yet");
  const plan_t& p = ws.plan; const uint64_t n = ws.n;
  const void* src = nullptr; uint64_t bytes = 0;
  if (ws.zero_clean_words && (!strcmp(stage, "bucket_count") || !strcmp(stage, "num_segments") || !strcmp(stage, "partials")))
    return set_err(ctx, TE_MSM_ESTATE, "this stage lives in the block that is cleared behind an MSM's read-back: set option prezero = 0 before the run to keep it");
  if (!strcmp(stage, "records")) {
    if (!ws.recs_last) return set_err(ctx, TE_MSM_ESTATE, "the last run gathered from a bound point set: the work set holds no records of its own");
    src = ws.recs_last; bytes = n * sizes_of(p.curve).rec;
  }
  else if (!strcmp(stage, "digits")) { src = ws.d_digits; bytes = (uint64_t)p.nw * p.nst * 2; }   // row stride nst = n rounded up to 8
  else if (!strcmp(stage, "bucket_count")) { src = ws.d_bucket_count; bytes = (uint64_t)p.nw * p.B * 4; }
  else if (!strcmp(stage, "bucket_start")) { src = ws.d_bucket_start; bytes = (uint64_t)p.nw * p.B * 4; }
  else if (!strcmp(stage, "sorted")) { src = ws.d_sorted; bytes = (uint64_t)p.nw * n * 4; }
  else if (!strcmp(stage, "num_segments")) { src = ws.d_num_seg; bytes = 4; }
  else if (!strcmp(stage, "seg_bucket")) { src = ws.d_seg_bucket; bytes = ((uint64_t)p.nw * p.B + (uint64_t)p.nw * (n / p.seg_len)) * 4; }
  else if (!strcmp(stage, "seg_len")) { src = ws.d_seg_lenv; bytes = ((uint64_t)p.nw * p.B + (uint64_t)p.nw * (n / p.seg_len)) * 4; }
  else if (!strcmp(stage, "order")) { src = ws.d_order; bytes = ((uint64_t)p.nw * p.B + (uint64_t)p.nw * (n / p.seg_len)) * 4; }
  else if (!strcmp(stage, "part_start")) { src = ws.d_part_start; bytes = (uint64_t)p.nw * p.P * 4; }
  else if (!strcmp(stage, "part_count")) { src = ws.d_part_count; bytes = (uint64_t)p.nw * p.P * 4; }
  else if (!strcmp(stage, "part_keys") && !p.packed) { src = ws.d_part_keys; bytes = (uint64_t)p.nw * p.nst * 2; }
  else if (!strcmp(stage, "part_idx") && !p.packed) { src = ws.d_part_idx; bytes = (uint64_t)p.nw * p.nst * 4; }
  else if (!strcmp(stage, "part_keys") || !strcmp(stage, "part_idx")) {
    // packed level-1 entries (index | bucket low bits << 23 | sign << 31): taken apart here into the two views of the general form
    const bool want_keys = stage[5] == 'k';
    const uint64_t words = (uint64_t)p.nw * p.nst;
    std::vector<uint32_t> pk(words);
    HIP_TRY(ctx, hipSetDevice(d.device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(pk.data(), ws.d_part_idx, words * 4, hipMemcpyDeviceToHost));
    uint64_t out_bytes = words * (want_keys ? 2 : 4);
    if (out_bytes > cap) out_bytes = cap;
    if (want_keys) { uint16_t* o = static_cast<uint16_t*>(dst); for (uint64_t i = 0; i < out_bytes / 2; i++) o[i] = (uint16_t)(((pk[i] >> 23) & 0xffu) | ((pk[i] >> 31) << 15)); }
    else { uint32_t* o = static_cast<uint32_t*>(dst); for (uint64_t i = 0; i < out_bytes / 4; i++) o[i] = pk[i] & 0x7fffffu; }
    return (int64_t)out_bytes;
  }
  else if (!strcmp(stage, "buckets")) { src = ws.d_buckets; bytes = (uint64_t)p.nw * p.B * sizes_of(p.curve).acc; }
  else if (!strcmp(stage, "partials")) { src = ws.d_partials; bytes = (uint64_t)p.W * sizes_of(p.curve).row; }
  else return set_err(ctx, TE_MSM_EINVAL, "unknown stage");
  if (bytes > cap) bytes = cap;
  HIP_TRY(ctx, hipSetDevice(d.device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return (int64_t)bytes;
}

}  // extern "C"
