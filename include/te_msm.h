/*
 * te_msm.h -- C-ABI of the MI355X-native Twisted-Edwards-BLS12 MSM engine (libtemsm.so).
 *
 * This is the drop-in boundary for the reference's hot path.  Nothing like it exists in the
 * reference (a browser program); each entry point names the reference interface it stands behind
 * (paths relative to /root/reference/src).  INTEGRATION.md shows the N-API binding that keeps
 * `compute_msm(bufferPoints, bufferScalars)` (submission/submission.ts:73-78) intact.
 *
 * Wire format (README.md:297-299; encoder reference/webgpu/utils.ts:90-99):
 *   points : n x (x[32 B little-endian] || y[32 B little-endian]), canonical affine, NOT Montgomery
 *   scalars: n x 32 B little-endian integers (< p < 2^253; signed digits accept anything below 2^254 - 2^240 at every
 *            window size -- TE_MSM_ESCALAR above --, unsigned digits any 256-bit value)
 *   result : x[32 B LE] || y[32 B LE], canonical affine  ( == result.toAffine(), submission.ts:412 )
 *
 * All functions return 0 on success or a negative TE_MSM_E* code; te_msm_last_error() gives text.
 * A context is not thread-safe: serialise the calls on one context (the one exception is te_msm_ticket_wait).
 * Every entry point leaves the calling thread's current HIP device as it found it (hipSetDevice is per-thread state that a
 * multi-device context has to change while it works).
 * There is NO CPU fallback: without a usable HIP device te_msm_init fails.
 */
#ifndef TE_MSM_H
#define TE_MSM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct te_ctx te_ctx;

#define TE_MSM_OK            0
#define TE_MSM_EINVAL      (-1)   /* bad argument */
#define TE_MSM_EDEVICE     (-2)   /* HIP error (no device, out of memory, launch failure) */
#define TE_MSM_ESCALAR     (-3)   /* a scalar left a final carry: the reference's "final carry is 1"
                                     (submission/miscellaneous/utils.ts:80-83) */
#define TE_MSM_ESTATE      (-4)   /* call order / capacity */

/* Groups (option "curve").  0: the Twisted-Edwards BLS12 curve of the competition (default; everything above).
 * 1: BLS12-377 G1, y^2 = x^3 + 1 over the 377-bit base field (README.md:57-73, BASELINE config 5): points are
 * n x (x || y) with 48-byte little-endian coordinates (96 bytes), scalars n x 48-byte little-endian records holding
 * values below 2^256 (README.md:325-331), the result is x || y in 96 bytes (the point at infinity as 96 zero bytes) --
 * `out_xy_le` of te_msm_run / run_device / collect must then hold TE_MSM_RESULT_BYTES_MAX bytes.  Inputs must lie in G1 (the
 * subgroup of prime order r, as the harness generates them): the engine works in the curve's twisted-Edwards form, whose
 * map from y^2 = x^3 + 1 is undefined at the points of order 2 and 4.  Every entry point serves both curves; partial
 * rows of this curve are TE_MSM_PARTIAL_BYTES_BLS12_377 bytes per window (te_msm_partial_device, te_msm_finalize*). */
#define TE_MSM_CURVE_TE_BLS12      0
#define TE_MSM_CURVE_BLS12_377_G1  1
#define TE_MSM_RESULT_BYTES_MAX    96

#define TE_MSM_POINT_BYTES   64
#define TE_MSM_SCALAR_BYTES  32
#define TE_MSM_WORKSETS      8    /* MSMs one context can have in flight (submit/collect, partial_device) */
#define TE_MSM_PARTIAL_BYTES 720  /* per window: 5 extended points x 144 B (see te_msm_partial_device) */
#define TE_MSM_PARTIAL_BYTES_BLS12_377 1120   /* the same row for BLS12-377 G1: 5 points x 224 B (14 limbs per coordinate) */

/* Replaces get_device() + per-call buffer/pipeline creation (implementation/cuzk/gpu.ts:14-25,
 * submission.ts:96-97).  The context is persistent: buffers and streams live across calls.
 * n_dev == 1: one GPU.  n_dev > 1: the listed devices (ids may repeat) work for this one context -- this is how a
 * single-process host (the reference's JavaScript, README.md:551 "multi-device" future work) uses a node's GPUs.
 * Two shapes:
 *   MSMs in flight (te_msm_submit / te_msm_submit_async / te_msm_submit_device + te_msm_collect): one WHOLE MSM per
 *     device, a ticket goes to the device with the fewest in flight -- D PCIe links, no replicated bucket reduction, no
 *     row merge.  The throughput form for a caller with several MSMs to do (concurrent compute_msm promises).
 *   the lone call (te_msm_run / te_msm_run_device): every device works on the one MSM:
 *   te_msm_run (host buffers): POINT shards.  Points and scalars are cut into n_dev contiguous slices; one
 *     host thread per device uploads its slice over that device's own PCIe link (the upload is most of a
 *     host-buffer call: 1.85 of 2.47 ms at n = 2^20 on one device) and runs all windows on it; the host tail
 *     folds the sum of the devices' rows.  Window bits follow the slice size.  Option "host_shard_min": fewer
 *     devices are used when a slice would hold fewer points than that (default 4096).
 *   te_msm_run_device (inputs resident on the first device): WINDOW shards; the inputs travel as a scatter + all-gather
 *     over the peer links (slice i to device i, then every device pulls the other slices from their holders:
 *     2 (D - 1) / D of the input leaves the first device instead of D - 1 copies of it), enqueued by one host thread per device.
 * For one-process-per-GPU deployments use n_dev == 1 plus te_msm_set_window_shard / te_msm_partial_device /
 * te_msm_finalize and exchange the partial sums yourself (bench.py does it with an RCCL all-gather). */
int te_msm_init(const int* device_ids, int n_dev, te_ctx** out);
void te_msm_destroy(te_ctx* ctx);
const char* te_msm_last_error(const te_ctx* ctx);   /* ctx may be NULL: last init error */

/* compute_msm(bufferPoints, bufferScalars) -- submission.ts:73-413.  Host buffers; the callee copies
 * them to the device and does not retain them.  out_xy_le receives 64 bytes. */
int te_msm_run(te_ctx* ctx, const uint8_t* points_xy_le, const uint8_t* scalars_le, uint64_t n,
               uint8_t out_xy_le[64]);

/* Same with inputs already resident in device memory of the context's (first) device. */
int te_msm_run_device(te_ctx* ctx, const void* d_points_xy_le, const void* d_scalars_le, uint64_t n,
                      uint8_t out_xy_le[64]);

/* Pipelined form of te_msm_run_device: te_msm_submit_device enqueues every device stage plus
 * the 11 KB read-back and returns at once with a ticket; te_msm_collect waits for that MSM, runs the host tail and
 * writes the result.  Up to TE_MSM_WORKSETS MSMs may be in flight, each on its own stream and device work set: the host
 * tail of MSM k overlaps the device work of MSM k+1, and on the GPU the launch gaps and the latency-bound reduction tail
 * of one MSM are filled by the wide kernels of the others (the reference's full_benchmarks.ts loop awaits each call; a
 * prover calling MSMs back to back does not have to).
 * Inputs must stay valid AND UNCHANGED until the ticket is collected.  Tickets may be collected in any order (until round 4: in
 * submission order only).
 * Calls in flight that name the SAME point buffer (pointer, n) share one record slab (round 6, option "share_records" = 1): every call
 * still converts its points -- nothing is remembered across calls, the buffer may hold other points for the next call once this one is
 * collected --, but all of them convert into, and gather from, the same 128 bytes per point instead of one slab per work set: four MSMs in
 * flight keep their gathers inside the 256 MB Infinity Cache (+4-5 % MSM/s at n = 2^20: profiles/r06_share_records_ab.txt).  This is why
 * "unchanged" matters: overwriting a point buffer that a ticket in flight still names, and submitting it again, would rewrite the records
 * under that ticket.
 * Contexts of several devices: the inputs may be resident on ANY device of the context (both buffers on the same one); the
 * ticket goes to the device with the fewest MSMs in flight -- ties to the device that holds the inputs -- and a device that
 * does not hold them pulls them over its peer link first (hipMemcpyPeerAsync on the work set's stream; option
 * "stage_device_inputs" = 1 forces that copy even onto the holder: tests on a one-GPU box).  A ticket is a whole MSM there
 * (the window shards of a multi-device context belong to te_msm_run_device).
 * A work set owned by an uncollected ticket is never reused underneath it: te_msm_run / te_msm_run_device move to a
 * free work set (TE_MSM_ESTATE when all TE_MSM_WORKSETS are owned), te_msm_partial_device on such a set returns
 * TE_MSM_ESTATE.  A ticket is consumed by te_msm_collect whether it ends in a result or in TE_MSM_ESCALAR. */
int te_msm_submit_device(te_ctx* ctx, const void* d_points_xy_le, const void* d_scalars_le, uint64_t n, uint64_t* ticket);
int te_msm_collect(te_ctx* ctx, uint64_t ticket, uint8_t out_xy_le[64]);
/* The same pipeline for HOST buffers -- what N concurrent compute_msm() promises of a JavaScript prover map onto
 * (ui/Benchmark.tsx:32 awaits an async call; nothing stops a caller from having several in flight): uploads the buffers
 * (in pieces, like te_msm_run) into the staging area of a free work set, enqueues every device stage and returns a ticket
 * for te_msm_collect.  The call returns when the data has LEFT the caller's buffers (pageable copies are staged by the
 * calling thread; for pinned / hipHostRegister'ed buffers, whose copies are truly asynchronous, the call waits for the last
 * upload), so points_xy_le / scalars_le may be released as soon as it returns; the upload of MSM k+1 overlaps the
 * device work of MSM k.  Contexts of several devices: the ticket goes to the device with the fewest in flight (idle
 * devices in turn); the uploads of consecutive calls still follow each other on the calling thread -- see te_msm_submit_async. */
int te_msm_submit(te_ctx* ctx, const uint8_t* points_xy_le, const uint8_t* scalars_le, uint64_t n, uint64_t* ticket);
/* The same with the upload itself in the background: the call picks device and work set, hands upload + enqueue to that
 * device's host thread and returns AT ONCE.  D calls in a row put D uploads on D PCIe links at the same time -- what a
 * single-threaded caller (the JavaScript event loop behind the N-API addon, a Python prover) needs to keep D devices busy
 * from host buffers: an EXTRAPOLATION from one device -- 8 x (1 / 2.2 ms) MSM/s, eight times the measured one-device rate, against
 * 1 / 0.69 ms for point slices of one call (one device's rehearsed share); two physical devices have never run either.
 * The price: points_xy_le / scalars_le must stay valid and unchanged until te_msm_ticket_wait or te_msm_collect has
 * returned for the ticket.  A failure of the upload or the enqueue (TE_MSM_EDEVICE) is reported by te_msm_collect, which
 * frees the ticket.  Every device has several upload threads (option "upload_threads", default 4): while one ticket's copy is
 * inside the runtime, the next ticket's is being prepared -- on ONE device that alone takes tickets in flight from 2.10 to
 * 1.8-1.9 ms per 2^20-point MSM; the uploads of one device's tickets may therefore complete out of submission order (each owns
 * its work set: nothing depends on the order).  Works on single-device contexts too. */
int te_msm_submit_async(te_ctx* ctx, const uint8_t* points_xy_le, const uint8_t* scalars_le, uint64_t n, uint64_t* ticket);
/* Blocks until the MSM of `ticket` has left the device (its rows are in host memory); te_msm_collect then returns without
 * waiting.  This is the ONE entry point that may be called from another thread while the context is in use elsewhere -- it
 * only waits on the ticket's event -- so a multi-threaded host (libuv's pool under the N-API addon) serialises submit and
 * collect with a lock of its own and waits outside it. */
int te_msm_ticket_wait(te_ctx* ctx, uint64_t ticket);
/* Where a ticket in flight runs: index into the context's device list and the HIP device id there (either may be NULL). */
int te_msm_ticket_device(te_ctx* ctx, uint64_t ticket, int* device_index, int* device_id);

/* ---- resident bases: points bound once, scalars per call ------------------------------------------------------------------
 * The reference's harness hands the SAME bufferPoints to compute_msm for every run of a size (ui/AllBenchmarks.tsx:213-222,
 * submission/miscellaneous/full_benchmarks.ts:63-68,100-105: one buffer, six calls), and a prover runs every MSM over one
 * SRS; te_msm_run re-uploads 64 of its 96 bytes per point and converts the points again on every call.  A bound point set
 * pays that once: te_msm_bind_points uploads the points, converts them to records ON EVERY DEVICE of the context and keeps
 * only the records (128 bytes per point; BLS12-377: 168 bytes, see below); the MSMs over it take scalars alone -- 32 of 96
 * bytes over PCIe, no conversion.  compute_msm's signature (submission.ts:73-78) is untouched: the N-API addon offers
 * setBases(buffer) as an opt-in, after which compute_msm(thatBuffer, scalars) takes this path.
 *   The curve is the one selected at bind time (option "curve"); the MSMs must run under the same curve (TE_MSM_EINVAL
 *   otherwise).  Window bits, digit form, segment length follow the options at the time of each MSM, as for te_msm_run.
 *   BLS12-377 G1: because the conversion is paid once, the bound records are AFFINE -- one inversion per point at bind time
 *   (Montgomery's trick, a fixed Fermat chain per eight points) -- 168-byte records and 7 field products per accumulated
 *   point instead of 224 bytes and 8 (option "bind_affine" = 0 keeps the projective records: A/B measurements).  Results are
 *   identical either way.
 *   A bound set holds device memory until te_msm_release_points or te_msm_destroy (option "bases_bytes" reports it); it may be
 *   released only when no ticket that uses it is in flight (TE_MSM_ESTATE).  The caller's point buffer is not retained.
 * te_msm_bind_points_device: the same from points already resident on a device of the context. */
typedef struct te_bases te_bases;
int te_msm_bind_points(te_ctx* ctx, const uint8_t* points_xy_le, uint64_t n, te_bases** out);
int te_msm_bind_points_device(te_ctx* ctx, const void* d_points_xy_le, uint64_t n, te_bases** out);
int te_msm_release_points(te_ctx* ctx, te_bases* bases);
/* number of points of a bound set (0 for NULL) */
uint64_t te_msm_bases_count(const te_bases* bases);
/* compute_msm over a bound set: scalars_le holds te_msm_bases_count(bases) scalars (host memory, the wire format of te_msm_run).
 * One device: the scalars are uploaded and processed in pieces (option "host_chunks"; piece i + 1 crosses PCIe while piece i
 * is sorted and accumulated onto the same buckets).  Several devices: every device takes a slice of the scalars over its own
 * link and the matching slice of its copy of the records (the point shards of te_msm_run without the points). */
int te_msm_run_scalars(te_ctx* ctx, te_bases* bases, const uint8_t* scalars_le, uint8_t out_xy_le[64]);
/* the same with the scalars resident on a device of the context (te_msm_run_device without the points) */
int te_msm_run_scalars_device(te_ctx* ctx, te_bases* bases, const void* d_scalars_le, uint8_t out_xy_le[64]);
/* Tickets over a bound set (te_msm_ticket_wait / te_msm_collect as for every ticket; the ticket goes to the device with the
 * fewest in flight, every device holds the records).  te_msm_submit_scalars is ASYNCHRONOUS like te_msm_submit_async: it
 * returns at once, the upload runs on one of the device's upload lanes, and scalars_le must stay valid and unchanged until
 * te_msm_ticket_wait or te_msm_collect has returned for the ticket.  With MSMs in flight the 32 bytes per point of one MSM cross
 * PCIe (0.6 ms at n = 2^20) under the device work of the others (1.0 ms): the boundary stops being the link. */
int te_msm_submit_scalars(te_ctx* ctx, te_bases* bases, const uint8_t* scalars_le, uint64_t* ticket);
int te_msm_submit_scalars_device(te_ctx* ctx, te_bases* bases, const void* d_scalars_le, uint64_t* ticket);

/* Options (the reference hard-codes these: chunk_size submission.ts:80, dispatch table :109-142).
 *   "window_bits"   c in [4,16]; 0 = choose from n (default)
 *   "signed_digits" 1 = signed window digits, 2^(c-1) buckets per window (default; the reference's shipped behaviour,
 *                   miscellaneous/utils.ts:52-95); 0 = plain unsigned windows, 2^c buckets (utils.ts:34-50): same
 *                   result, accepts any 256-bit scalar (no final-carry error)
 *   "curve"         TE_MSM_CURVE_TE_BLS12 (default) or TE_MSM_CURVE_BLS12_377_G1, see above
 *   "sort_buckets"  1 = schedule buckets by descending size (default), 0 = natural order
 *   "segment_len"   a bucket longer than this is accumulated by several threads and the parts summed afterwards; 0 = from n
 *                   (default: twice the mean bucket size as a power of two in [16, 64]; read-only "segment_len_used" reports
 *                   what the last MSM ran with)
 *   "profile"       1 = HIP events around the dominant kernel (accumulate) only, 2 = around every stage
 *                   (te_msm_stage_ms); 0 = none (default)
 *   "graph"         1 = replay the launches before and after the accumulate kernel (five and five) as two HIP graphs, captured on
 *                   first use and re-captured when pointers, n or options change; 0 = launch every kernel (default: on
 *                   ROCm 7.2 / MI355X the replay measured ~5 % slower than plain launches).
 *                   Ignored at profile level 2.
 *   "host_chunks"   te_msm_run / te_msm_submit: pieces the point buffer (of one device's slice) is uploaded and processed
 *                   in, so that PCIe transfer and device work overlap; 0 = from n (3 from 3 * 2^18 points, 2 from 2^17,
 *                   else 1), 1 = whole.  Twisted-Edwards: all scalars go first, in one copy (the link is the bottleneck);
 *                   BLS12-377: scalars piece by piece with their points (the device is).  The result does not depend on it.
 *   "host_shard_min" multi-device te_msm_run: smallest slice worth a device of its own (default 4096 points)
 *   "upload_threads" host threads per device that take te_msm_submit_async / te_msm_submit_scalars tickets (1..16, default 4; env
 *                   TE_MSM_UPLOAD_THREADS).  Scalars-only tickets (bound bases) cross the link one at a time per device whatever the
 *                   number of threads (side by side they only delay each other: profiles/r06_bound_host_tickets_gap.txt)
 *   "host_staging"  0 (default) = host buffers are copied straight from the caller's memory: the fastest form while the caller
 *                   REUSES its buffers (the runtime keeps pages it has copied from registered: 2.35 ms per 2^20-point call), but
 *                   the first call on a buffer costs 4.7-5.0 ms and a caller that allocates fresh buffers for every call pays
 *                   4.4-27 ms per call (profiles/r05_host_buffers_first_touch.txt).  1 = the engine copies the buffers, 2 MB at a
 *                   time, with a crew of eight host threads into a pinned ring of the work set (32 MB, allocated on first use)
 *                   and uploads from there: the same cost whatever the history of the caller's memory (env TE_MSM_HOST_STAGING).
 *                   A caller that can register its arena once (hipHostRegister) should do that instead and keep 0.
 *   "queue_probe"   1 (default) = the first te_msm_submit_device on a device measures the hardware queues of its work sets' streams
 *                   (see te_msm_workset_stream); 0 = never (env TE_MSM_QUEUE_PROBE=0).  te_msm_probe_queues does it on request.
 *                   te_msm_submit / te_msm_submit_async never trigger it: a host-buffer ticket is bound by its upload
 *                   (until round 4 they did: 16 ms at the first submit).
 *   "workset"       which of the TE_MSM_WORKSETS device work sets te_msm_run* / te_msm_partial_device use (default 0)
 *   "stage_device_inputs"  see te_msm_submit_device (default 0)
 *   "bind_affine"   1 (default) = te_msm_bind_points converts BLS12-377 points to AFFINE records (one inversion per point, once);
 *                   0 = keeps the projective records of the per-call conversion (A/B measurements).  Read at bind time.
 *   "bind_fixed_base" c in [16, 21] (0 = off, default): te_msm_bind_points (Twisted-Edwards curve) also tabulates 2^(c w) P_i for the
 *                   ceil(255 / c) windows w (W x 128 bytes per point: 1.7 GB for c = 20 at n = 2^20, built once).  MSMs over such a set run
 *                   FIXED-BASE WINDOWS: every window's digit addresses the SAME set of 2^(c-1) buckets, so c can grow: 13 n + 2^20 additions
 *                   at c = 20 where 16-bit windows need 16 n + 2^20, and no window doublings in the host tail (csrc/kernels.hip.hpp,
 *                   "FIXED-BASE WINDOWS"; measurements: profiles/r06_fixed_base_windows.txt).  Same results.  The rows
 *                   of such an MSM are sized for well-spread digits; badly skewed scalars (all equal, ...) overflow one, which the engine
 *                   detects and answers by running that MSM again with the ordinary windows (read-only option "fixed_base_fallbacks"
 *                   counts them).  A lone call on several devices runs on ONE of them (one bucket set: nothing to shard); tickets use all.
 *                   "window_bits", "signed_digits", "segment_len" = 0 do not apply to such MSMs; a window-sharded single-device context
 *                   keeps the ordinary windows.  Read at bind time.
 *   "exp_table_replicas" 1..8 (default 1; EXPERIMENT, read at bind time): keeps that many copies of the bound records and lets the windows
 *                   of a device-scalar MSM gather from different copies -- the gather footprint of a per-window table with the arithmetic
 *                   unchanged (profiles/r06_fixed_base_windows.txt, step 1).  Same results.
 *   "share_records" 1 (default) = whole-MSM calls from device-resident inputs (te_msm_submit_device, te_msm_run_device) that name the same
 *                   point buffer while in flight share one record slab, see te_msm_submit_device; 0 = one slab per work set (A/B; env
 *                   TE_MSM_SHARE_RECORDS).  Read-only "record_slabs": slabs allocated.
 *   "lane_host_waits" 1 (default) = the lane thread of an asynchronous ticket (te_msm_submit_async, te_msm_submit_scalars) waits for each of
 *                   its uploads on the host before it enqueues the kernels that read it; 0 = a stream wait in front of those kernels, which
 *                   holds up other tickets' kernels in a shared hardware queue (A/B; env TE_MSM_LANE_HOST_WAITS; profiles/r06_lane_host_waits.txt).
 *                   The calling thread of te_msm_run* / te_msm_submit waits the same way when its buffers are pageable (the copy call
 *                   blocks for the copy anyway); pinned buffers keep the stream waits.  No packet of the upload path enters a hardware
 *                   queue shared with other tickets' kernels (profiles/r06_bound_host_tickets_gap.txt, r06_caller_host_waits.txt).
 *                   The earlier forms stay behind environment switches for A/B runs, read once per process and covered by
 *                   tests/test_gpu_tickets.py::test_upload_path_switches_agree: TE_MSM_COPY_MARKER=1, TE_MSM_LANE_EVENT_WAITS=1,
 *                   TE_MSM_SCALAR_UPLOADS_SERIAL=0, TE_MSM_CALLER_HOST_WAITS=0, TE_MSM_COPY_PRIORITY=1|-1; TE_MSM_SERIAL_ACCUMULATE=1
 *                   chains the accumulations of the MSMs in flight (slower: profiles/r06_serial_accumulate_experiment.txt).
 *   "scalar_chunks" te_msm_run_scalars / te_msm_submit_scalars: pieces the scalars of a bound point set are uploaded and processed
 *                   in; 0 = from n (default), 1 = whole.  The result does not depend on it.
 *   read-only:      "num_devices", "segment_len_used", "peer_copies" / "peer_bytes" (hipMemcpyPeerAsync calls a multi-device
 *                   context issued, and the bytes they moved), "entries_accumulated" (non-zero window digits of the MSM whose
 *                   result was fetched last, counted on the device: the points k_accumulate gathered -- all windows of this
 *                   context's shard, all MSMs of a batch; bench.py prices its roofline with it),
 *                   "bases_bytes" (device memory held by bound point sets, all devices), "bases_bound" (how many sets),
 *                   "fixed_base_fallbacks" (fixed-base MSMs answered by the ordinary windows after a row overflow),
 *                   "in_flight" (tickets not collected, all devices), "streams_final" (te_msm_workset_stream's handles will not change any
 *                   more), "device_bytes" (device memory held in work-set buffers, see te_msm_trim)
 *   "prezero"       1 (default) = a work set's block of counters is cleared BEHIND an MSM's read-back, for its next MSM
 *                   (the next MSM starts with its first kernel instead of a fill); 0 = cleared in front of every MSM --
 *                   te_msm_debug_read of "bucket_count" / "num_segments" / "partials" needs 0 (it refuses otherwise)
 *   "packed_sort"   1 (default) = the sort's level-1 entries are one 32-bit word (index | key << 23 | sign << 31) where n <= 2^23:
 *                   4 bytes per entry instead of 6; 0 = the general form (u16 key + u32 index; what larger n always uses).
 *                   Same result (A/B measurements, tests; env TE_MSM_PACKED)
 *   "fold_pairs"    1 (default) = a fold level of the bucket reduction with 32 768 .. 65 535 outputs (n <= 2^18) runs two lanes per
 *                   output -- half the dependent additions; 0 = one thread per output (A/B measurements; env TE_MSM_FOLD_PAIRS)
 *   "fuse_prep"     1 (default) = device-resident Twisted-Edwards inputs: the points -> records conversion shares the launch
 *                   of the sort's first level; 0 = a launch of its own (A/B measurements; env TE_MSM_FUSE_PREP) */
int te_msm_set_option(te_ctx* ctx, const char* key, int64_t value);
int te_msm_get_option(te_ctx* ctx, const char* key, int64_t* value);

/* ---- window-sharded building blocks (multi-GPU: one process per GPU) --------------------------
 * This context computes windows w = first + k*step (k >= 0) of every MSM.  Default: first 0, step 1.
 * Number of windows W of a plan: ceil(255 / c) for signed digits (scalars below 2^254 - 2^240; 17 windows of 15 bits, 16 of
 * 16), ceil(256 / c) for unsigned ones; te_msm_plan reports it. */
int te_msm_set_window_shard(te_ctx* ctx, int first, int step);
/* Geometry for n points under the current options: window bits c and total number of windows W. */
int te_msm_plan(te_ctx* ctx, uint64_t n, int* window_bits, int* num_windows);
/* Runs every device stage for this context's windows and leaves the partial sums in DEVICE memory:
 * d_partials is W x TE_MSM_PARTIAL_BYTES; only the rows of this context's windows are written
 * (others untouched -- zero the buffer first; an all-zero row means "window not present").
 * Row w = [ T | W0 | W1 | W2 | W3 ]: T = sum of the window's buckets, Wk = sum_v v * (sum of the buckets whose index has
 * digit k equal to v), digits of (c+2-k)/4 bits; as extended points (x|y|z|t, each 9 limbs of 29 bits in
 * u32 words, Montgomery form R = 2^261, lazily reduced).  Asynchronous on `stream`: any hipStream_t (NULL is HIP's
 * default stream, which is also what PyTorch calls its default stream), or TE_MSM_OWN_STREAM for the context's
 * private stream; returns after enqueueing. */
#define TE_MSM_OWN_STREAM ((void*)(intptr_t)-1)
int te_msm_partial_device(te_ctx* ctx, const void* d_points_xy_le, const void* d_scalars_le, uint64_t n,
                          void* d_partials, void* stream);
/* `count` (1..TE_MSM_MAX_BATCH) MSMs of the same n in ONE sequence of launches: MSM m reads d_points_xy_le[m] /
 * d_scalars_le[m] (arrays of device pointers, in host memory, read before the call returns) and its W rows go to
 * d_partials + m * W * row bytes.  The windows of the batch are sorted, accumulated and reduced together, as if they were
 * count x (windows of this shard) windows of one MSM: what a rank of a D-GPU window-sharded job needs, because its W/D windows per
 * MSM are too little work for a launch sequence of their own (rehearsed per-rank step at D = 8: 0.25 ms per MSM one by one,
 * 0.17 ms in batches of eight).  MSMs of one call that name the SAME point buffer share one conversion
 * of it (same pointer in one call = same data; nothing is remembered across calls) -- a prover's batch over one SRS converts
 * it once per call; the scalars of all MSMs are decomposed by one launch.  There is no reference
 * counterpart: the reference awaits one compute_msm at a time (full_benchmarks.ts:97-110).  A scalar out of range anywhere
 * in the batch fails the whole batch (te_msm_partial_wait).  Same work set and stream rules as te_msm_partial_device. */
#define TE_MSM_MAX_BATCH 8
int te_msm_partial_device_batch(te_ctx* ctx, const void* const* d_points_xy_le, const void* const* d_scalars_le, uint64_t n,
                                int count, void* d_partials, void* stream);
/* A context owns TE_MSM_WORKSETS device work sets (option "workset" selects the one te_msm_partial_device uses), so a
 * caller can keep that many MSMs in flight on as many streams: their kernels overlap on the GPU.  te_msm_partial_wait blocks until the
 * last te_msm_partial_device call on that work set has finished and returns its status (TE_MSM_ESCALAR if a scalar was
 * out of range). */
int te_msm_partial_wait(te_ctx* ctx, int workset);
/* The private stream of a work set (what TE_MSM_OWN_STREAM selects) as a hipStream_t, for callers that order their own work
 * -- a collective, a copy -- behind te_msm_partial_device without a host round trip (PyTorch: torch.cuda.ExternalStream).
 * LIFETIME: a handle returned here stays a VALID hipStream_t until the process exits.  te_msm_destroy synchronises the streams
 * of a context whose handles were handed out and parks them (the next context on the same device takes them over) instead of
 * destroying them, because callers remember such handles where the engine cannot see them: PyTorch's pinned-memory allocator
 * records an event on every stream a pinned block was used on when the block is released -- for a tensor that outlives the
 * context, after te_msm_destroy or at interpreter exit -- and hipEventRecord on a destroyed stream aborts the process (round 5:
 * tools/exp_batch_small.py; profiles/r06_batch_small_abort.txt, tests/test_gpu_stream_export.py).  What the caller still owes:
 * work it enqueues on the handle AFTER te_msm_destroy is ordered with whatever the next owner of the stream runs there -- finish
 * (synchronise) your own work on the handle before the context goes away.
 * The FIRST te_msm_submit* of a context (not te_msm_init: one-shot callers never pay the ~16 ms) measures which of
 * its streams the runtime put on the same hardware queue (kernels of one queue run in order; see csrc/te_msm.hip,
 * spread_streams_over_queues), twice, and -- when both measurements agree -- re-deals them so that work sets 0..3, and 4..7,
 * sit on different queues: MSMs in flight on the context's own streams overlap whatever other streams the process has
 * created.  The measurement waits for the context's own streams only (no device-wide synchronisation: other streams of the
 * process keep running).  The handles may change at that call: query them after it, or call te_msm_probe_queues first
 * (option "streams_final" tells).  *hw_queue_class (optional) receives the measured class of the work set's stream, -1
 * before the measurement, when it is off (option "queue_probe" = 0) or when its two passes disagreed (creation order kept). */
int te_msm_workset_stream(te_ctx* ctx, int workset, void** stream, int* hw_queue_class);
/* Runs that measurement NOW (about 16 ms; nothing of this context may be in flight), at a quiet moment the caller chooses,
 * and fixes the work sets' streams for the life of the context.  Returns the number of hardware-queue classes found
 * (0: the two passes disagreed, creation order kept) or a negative error. */
int te_msm_probe_queues(te_ctx* ctx);
/* Gives device memory back: frees the buffers (about 0.7 GB each at n = 2^20) of the idle work sets numbered >=
 * keep_worksets, staging areas included (they are allocated on first use and otherwise held until te_msm_destroy -- fine on
 * 288 GB, unfriendly next to a prover).  Work sets owned by an uncollected ticket are skipped.  Returns the number of
 * work sets freed.  A later call simply allocates again. */
int te_msm_trim(te_ctx* ctx, int keep_worksets);
/* Host tail (replaces submission.ts:362-412: de-Montgomery, sum, Horner, toAffine): folds the W rows
 * (host memory; rows of absent windows all-zero are skipped as identity) into the affine result.
 * Waits for, and reports a pending TE_MSM_ESCALAR of, the last te_msm_partial_device call of this context.  The digit
 * form (signed / unsigned) is the one that call ran with, whatever the option says now; window_bits and num_windows
 * must be that call's (te_msm_plan), otherwise TE_MSM_ESTATE -- rows from elsewhere go to te_msm_finalize_host_ex. */
int te_msm_finalize(te_ctx* ctx, const uint8_t* partials, int window_bits, int num_windows,
                    uint8_t out_xy_le[64]);

/* The same tail without a context (pure host code, no device needed): used when the rows were produced
 * elsewhere, e.g. gathered from other ranks.  Does not know about scalar-range errors. */
int te_msm_finalize_host(const uint8_t* partials, int window_bits, int num_windows, uint8_t out_xy_le[64]);
/* Which form of the host tail's arithmetic this process uses: bit 0 = the mulx / adcx / adox field product (BMI2 + ADX; Twisted-Edwards
 * curve), bit 1 = the AVX-512 IFMA accumulator of both curves (the four coordinates of the running point in vector lanes: two vector
 * products per doubling; about half the time of the scalar form).  Both forms are checked against the portable one at te_msm_init; env
 * TE_MSM_HOST_TAIL=scalar turns bit 1 off, TE_MSM_HOST_MUL=c both (A/B measurements, tests). */
int te_msm_host_tail_features(void);
/* Same for either digit form: bucket_bits = window_bits - 1 (signed digits, what te_msm_finalize_host assumes) or
 * window_bits (option "signed_digits" = 0). */
int te_msm_finalize_host_ex(const uint8_t* partials, int window_bits, int bucket_bits, int num_windows, uint8_t out_xy_le[64]);
/* The tail for an all-gathered buffer: `gathered` holds `world` consecutive W x 720 B buffers, the r-th written by the
 * rank that owns windows {w : w mod world == r} (te_msm_set_window_shard(r, world)); row w is taken from buffer
 * w mod world.  Saves the caller the merge. */
int te_msm_finalize_gathered(const uint8_t* gathered, int world, int window_bits, int bucket_bits, int num_windows,
                             uint8_t out_xy_le[64]);

/* The two context-free tails for either curve (curve = TE_MSM_CURVE_*; rows of TE_MSM_PARTIAL_BYTES or
 * TE_MSM_PARTIAL_BYTES_BLS12_377 bytes; out_xy_le 64 or 96 bytes). */
int te_msm_finalize_host_curve(int curve, const uint8_t* partials, int window_bits, int bucket_bits, int num_windows, uint8_t* out_xy_le);
int te_msm_finalize_gathered_curve(int curve, const uint8_t* gathered, int world, int window_bits, int bucket_bits, int num_windows,
                                   uint8_t* out_xy_le);

/* The tail over the SUM of several row buffers (each W rows): the rows are linear in the bucket contents, so an MSM whose
 * POINTS were cut into slices -- every slice run through all windows with the same window_bits, e.g. one slice per GPU
 * (te_msm_run on a multi-device context does exactly this internally) or per process -- is the fold of the slices' rows
 * added up.  All-zero rows are skipped.  Pure host code. */
int te_msm_finalize_sum_curve(int curve, const uint8_t* const* row_sets, int sets, int window_bits, int bucket_bits, int num_windows,
                              uint8_t* out_xy_le);

/* ---- harness inputs (host code, no device needed).  The reference's harness generates its own random inputs when the
 * ZPrize files are not used (ui/AllBenchmarks.tsx:99-131, reference/webgpu/utils.ts:81-88,118-124): seeded scalars =
 * 256 random bits reduced mod p; points, by `fixed_point`: 0 = n distinct subgroup points (a + i*b)*G (an arithmetic
 * progression: one addition per point), 1 = the harness's one fixed point replicated n times (ui/AllBenchmarks.tsx:105-112),
 * 2 = n independent points a_i * G with seeded-random a_i ("random points", SURVEY.md 8d set (R); fixed-base table, all host
 * threads: about a second per 2^20).  Either output pointer may be NULL. */
#define TE_MSM_SYNTH_CHAIN  0
#define TE_MSM_SYNTH_FIXED  1
#define TE_MSM_SYNTH_RANDOM 2
int te_msm_synth_inputs(uint64_t seed, uint64_t n, int fixed_point, uint8_t* points_xy_le, uint8_t* scalars_le);
/* The same scheme for BLS12-377 G1: 96-byte points (a + i*b)*G, 48-byte scalar records (values below r). */
int te_msm_synth_inputs_bls12_377(uint64_t seed, uint64_t n, uint8_t* points_xy_le, uint8_t* scalars_le);

/* ---- measurement / stage verification (the reference's `debug` flags, submission.ts:892-1363) --- */
/* Per-stage device time of the last run in ms (needs option "profile" >= 1; level 1 reports "accumulate" only), from HIP events
 * on the engine's stream, plus "accumulate_on_device": the dominant kernel's own device clock from its first wave in to its last
 * wave out -- what a kernel trace reports; with several MSMs in flight the event interval also contains the time the launch
 * waited behind other streams' kernels -- and "accumulate_core_clock_ghz" (a frequency, not a time): the mean shader clock the
 * kernel's waves ran at, from per-wave shader-clock and wall-clock ticks.  Returns the number of entries written; names[i]
 * points to static strings. */
int te_msm_stage_ms(te_ctx* ctx, float* ms, const char** names, int max_stages);
/* Copies an intermediate buffer of the last run to host memory.  stage is one of
 * "records" (n x 128 B), "digits" / "part_keys" (nw rows of u16, row stride n rounded up to 8), "part_idx" (same rows, u32),
 * "part_start" / "part_count" (nw x P u32), "bucket_count" / "bucket_start" (nw x B u32), "sorted" (nw x n u32), "num_segments" (u32),
 * "seg_bucket" / "seg_len" / "order" (num_segments u32),
 * "buckets" (nw x B x 144 B), "partials" (W x 720 B).  Returns bytes copied
 * (<= cap) or a negative error. */
int64_t te_msm_debug_read(te_ctx* ctx, const char* stage, void* dst, uint64_t cap);
/* Copies records [first, first + count) of a bound point set from the memory of device `device_index` of the context to dst
 * (stage verification of te_msm_bind_points).  Record layouts, little-endian u32 limb words of 29 bits, Montgomery form:
 * Twisted-Edwards BLS12: 128-byte slots, hm | hp | dt of 9 words each; BLS12-377: 168 bytes, hm | hp | dt of 14 words (affine,
 * option "bind_affine" = 1) or 224 bytes, hm | hp | dt | z (projective).  *record_bytes (optional) receives the slot size.
 * Returns bytes copied (<= cap) or a negative error. */
int64_t te_msm_bases_read(te_ctx* ctx, const te_bases* bases, int device_index, uint64_t first, uint64_t count, void* dst, uint64_t cap,
                          int* record_bytes);

#ifdef __cplusplus
}
#endif
#endif /* TE_MSM_H */
