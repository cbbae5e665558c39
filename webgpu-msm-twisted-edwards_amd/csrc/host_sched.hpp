// host_sched.hpp -- the parts of the engine's host side that involve more than one thread: the per-device host thread (a
// FIFO of jobs) and the bookkeeping of MSMs in flight (tickets on work sets, which device the next one goes to).
//
// No HIP in here.  te_msm.hip builds the C-ABI's submit / ticket_wait / collect on these pieces, and
// tests/csrc/sched_harness.cpp drives the SAME code under ThreadSanitizer with a stand-in for the device (SURVEY.md
// section 5, "race detection / sanitizers"; the reference is single-threaded JavaScript and has no counterpart --
// its async call convention is ui/Benchmark.tsx:32, multi-device its README's future work, README.md:551).
#pragma once
#include <stdint.h>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>

namespace te_sched {

// One unit of work for a device's host thread.  `done` / `rc` are guarded by the mutex of the worker that runs it (`owner`).
class worker_t;
struct job_t {
  std::function<int()> fn;
  worker_t* owner = nullptr;     // set by post(): whoever holds the job can wait for it without knowing which thread took it
  int rc = 0;
  bool done = false;
};
using job_ref = std::shared_ptr<job_t>;

// One host thread per device of a context.  Pageable host-to-device copies block the thread that issues them while the data
// is staged, so D uploads issued from one thread run one after another; D threads drive D PCIe links at once.  The thread is
// persistent (a wake-up costs microseconds; creating a thread and its HIP thread state per call would cost more than a
// small MSM) and works its jobs off in the order they were posted -- tickets that went to one device keep their order.
class worker_t {
 public:
  worker_t() : th_([this] { loop(); }) {}
  worker_t(const worker_t&) = delete;
  worker_t& operator=(const worker_t&) = delete;
  // finishes what was posted, then joins (the jobs refer to state of the context: the context drops its workers first)
  ~worker_t() {
    { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
  }
  job_ref post(std::function<int()> fn) {
    job_ref j = std::make_shared<job_t>();
    j->fn = std::move(fn);
    j->owner = this;
    { std::lock_guard<std::mutex> lk(mu_); q_.push_back(j); }
    cv_.notify_all();
    return j;
  }
  // blocks until the job has run; any thread, any number of times
  int wait(const job_ref& j) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return j->done; });
    return j->rc;
  }
  // blocks until nothing is queued or running
  void drain() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return q_.empty() && !running_; });
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [&] { return !q_.empty() || quit_; });
      if (q_.empty()) return;                    // quit, and nothing left to do
      job_ref j = q_.front();
      q_.pop_front();
      running_ = true;
      lk.unlock();
      const int r = j->fn();
      j->fn = nullptr;                           // drop what the closure holds before anybody is told
      lk.lock();
      j->rc = r; j->done = true; running_ = false;
      cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<job_ref> q_;
  bool quit_ = false, running_ = false;
  std::thread th_;                               // last: the thread starts in the constructor and uses everything above
};

// Ticket state of one work set.  A ticket is published with release semantics AFTER everything a waiter needs (the job, the
// plan) has been written, and looked up with acquire loads: te_msm_ticket_wait may run on any thread while another thread
// submits or collects OTHER tickets.  One ticket is waited for / collected by one thread at a time (the caller's protocol).
struct slot_t {
  uint64_t ticket = 0;       // 0 = free
  job_ref job;               // asynchronous submit: the enqueue running on the device's host thread (null: enqueued by the submitting thread)
};
inline void slot_publish(slot_t& s, uint64_t ticket) { __atomic_store_n(&s.ticket, ticket, __ATOMIC_RELEASE); }
inline uint64_t slot_ticket(const slot_t& s) { return __atomic_load_n(&s.ticket, __ATOMIC_ACQUIRE); }
inline void slot_release(slot_t& s) { s.job.reset(); __atomic_store_n(&s.ticket, (uint64_t)0, __ATOMIC_RELEASE); }

// Which device does the next whole MSM go to?  The one with the fewest MSMs in flight among those that still have a free
// work set; ties go to `prefer` (the device that already holds the inputs, or -1), then to the device after `last` in
// round-robin order (so that D tickets on D idle devices spread over all of them).  -1: every work set of every device is taken.
inline int pick_device(const int* in_flight, int n_dev, int sets_per_dev, int prefer, int last) {
  int best = -1;
  for (int k = 1; k <= n_dev; k++) {
    const int i = (last + k) % n_dev;
    if (in_flight[i] >= sets_per_dev) continue;
    if (best < 0 || in_flight[i] < in_flight[best] || (in_flight[i] == in_flight[best] && i == prefer)) best = i;
  }
  return best;
}

// ---- the bookkeeping of tickets, written once for anything shaped like the engine's context:
//   ctx.devs[i]            devices (random access), each with  int in_flight  and work sets  ws[0 .. sets)  holding a  slot_t slot
//   ctx.next_ticket        uint64_t, tickets are handed out in order over all devices
//   ctx.last_dev           int, the device the previous ticket went to
//   ctx.workers[i]         std::unique_ptr<worker_t>, device i's host thread (the lone call's slices and shares)
//   ctx.lanes, ctx.next_lane   the upload lanes of asynchronous tickets, see next_lane_of
// te_msm.hip instantiates these with te_ctx / gpu_t / workset_t, tests/csrc/sched_harness.cpp with a stand-in device.
// Except for find_ticket and await_job -- which only read, with acquire loads, and may run on any thread -- everything here
// runs under the caller's serialisation of the context.

// the lowest-numbered work set of a device that no ticket owns (-1: none)
template <class Dev> int free_set_index(const Dev& d, int sets) {
  for (int i = 0; i < sets; i++) if (!slot_ticket(d.ws[i].slot)) return i;
  return -1;
}
// the device the next ticket goes to (index into ctx.devs; -1: everything is taken); prefer: the device that holds the inputs, or -1
template <class Ctx> int pick_device_of(const Ctx& ctx, int sets, int prefer) {
  const int nd = (int)ctx.devs.size();
  if (nd == 1) return ctx.devs[0].in_flight < sets ? 0 : -1;
  int fl[64];
  for (int i = 0; i < nd && i < 64; i++) fl[i] = ctx.devs[(size_t)i].in_flight;
  return pick_device(fl, nd < 64 ? nd : 64, sets, prefer, ctx.last_dev);
}
// the ticket exists from here on: job first, then the number (release), then the counters
template <class Ctx, class Set> void hand_out(Ctx& ctx, int di, Set& ws, uint64_t* ticket, job_ref job = nullptr) {
  *ticket = ctx.next_ticket++;
  ws.slot.job = std::move(job);
  slot_publish(ws.slot, *ticket);
  ctx.devs[(size_t)di].in_flight++; ctx.last_dev = di;
}
// the work set that holds a ticket, or nullptr (any thread)
template <class Ctx> auto find_ticket(Ctx& ctx, uint64_t ticket, int* dev_index = nullptr) -> decltype(&ctx.devs[0].ws[0]) {
  if (!ticket) return nullptr;
  for (size_t i = 0; i < ctx.devs.size(); i++)
    for (auto& ws : ctx.devs[i].ws) if (slot_ticket(ws.slot) == ticket) { if (dev_index) *dev_index = (int)i; return &ws; }
  return nullptr;
}
// the enqueue of an asynchronous ticket has run on the host thread that took it (any thread may ask); its status
template <class Set> int await_job(Set& ws) {
  const job_ref job = ws.slot.job;
  if (!job) return 0;
  return job->owner->wait(job);
}
// the ticket is over (collected, with a result or with an error)
template <class Ctx, class Set> void retire(Ctx& ctx, int dev_index, Set& ws) {
  ctx.devs[(size_t)dev_index].in_flight--;
  slot_release(ws.slot);
}
// device i's host thread, created on first use
template <class Ctx> worker_t& worker_of(Ctx& ctx, size_t i) {
  if (ctx.workers.size() < ctx.devs.size()) ctx.workers.resize(ctx.devs.size());
  if (!ctx.workers[i]) ctx.workers[i].reset(new worker_t());
  return *ctx.workers[i];
}
// Upload lanes: the host threads that take ASYNCHRONOUS tickets (upload + enqueue of one whole MSM each), `lanes` per device.
// One pageable upload keeps its thread inside the runtime while the data is staged; a second thread preparing the next
// ticket's copy meanwhile keeps the link busy -- on ONE device 2.10 ms per 2^20-point MSM with one upload thread, 1.9-1.8 with
// four to eight (profiles/r05_upload_lanes.txt).  Tickets of one device may therefore finish their uploads out of submission
// order; each owns its work set, so nothing depends on the order.   ctx.lanes: std::vector<std::unique_ptr<worker_t>>,
// ctx.next_lane: a counter.
template <class Ctx> worker_t& next_lane_of(Ctx& ctx, size_t dev_index, int lanes) {
  const size_t need = ctx.devs.size() * (size_t)lanes;
  if (ctx.lanes.size() < need) ctx.lanes.resize(need);
  const size_t i = dev_index * (size_t)lanes + (size_t)(ctx.next_lane++ % (uint64_t)lanes);
  if (!ctx.lanes[i]) ctx.lanes[i].reset(new worker_t());
  return *ctx.lanes[i];
}
template <class Ctx> void drain_workers(Ctx& ctx) {
  for (auto& w : ctx.workers) if (w) w->drain();
  for (auto& w : ctx.lanes) if (w) w->drain();
}

}  // namespace te_sched
