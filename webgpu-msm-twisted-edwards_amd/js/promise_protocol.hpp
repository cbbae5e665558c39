// promise_protocol.hpp -- how concurrent compute_msm() promises map onto the engine's tickets: the lock protocol of the
// N-API addon, free of N-API, so that addon.cc (the real engine through the C-ABI) and tests/csrc/sched_harness.cpp (a
// stand-in engine, under ThreadSanitizer) run the SAME code.
//
// Threads: ONE thread calls enter() for every new promise (the JavaScript thread); any number of pool threads call
// execute() for the jobs that were entered, started in the order they were entered (libuv's pool is a FIFO).  The engine's
// context is not thread-safe: every call into it happens under `mu_`, except Api::ticket_wait -- the one entry point the
// C-ABI allows beside others (include/te_msm.h).
//
// Jobs become tickets in the order they were entered (`waiting_` is a FIFO), by whoever holds the engine's lock at a moment
// when a work set is free:
//   enter    queues the job; when the context exists and the engine's lock is FREE (and the job is not a lone one on several
//            devices: that one is every device's), turns the waiting jobs into tickets right there, oldest first (submit_async returns at once: device and work set are picked, the upload is
//            handed to that device's host thread) -- so the number of MSMs in flight is bounded by the engine's work sets,
//            not by the size of the pool.  The lock is only TRIED: a pool thread may be inside te_msm_init, a lone
//            multi-device run or a collect's host tail for milliseconds, and the JavaScript thread must not stall on it
//            (round-5 advisor); that thread submits the waiting jobs itself when it leaves the engine.
//   execute  submits what can be submitted (oldest first), then: the job is a ticket -> wait for it outside the lock, collect
//            under it, and hand the freed work set to the oldest waiting job; the job is the oldest waiting one and ALONE
//            (nothing else pending) on several devices, or empty -> the lone call, Api::run (every device works on the one
//            MSM); otherwise every work set is taken by OLDER jobs' tickets -- those jobs were started before this one, so
//            each has a pool thread that will collect -- and the job waits for one of their collects.
// Round 5 submitted from enter() whatever was waiting: the first promise of a burst stayed unsubmitted while promises 2 .. 1 +
// 8 D took every work set, and its pool thread then parked on capacity -- with a pool of ONE thread (UV_THREADPOOL_SIZE=1) and
// nine promises on one device nobody was left to collect (round-5 advisor).  With FIFO submission a thread can only wait for
// tickets of jobs older than its own, all of which are running: no thread parks on capacity while a submitted ticket has none.
//
// Resident bases (include/te_msm.h): set_bases(points, n) binds a point buffer once; a job whose points are THAT buffer
// (same address and length: the addon keeps the Buffer alive, so the address cannot be another object's) takes the
// scalars-only path -- Api::submit_scalars / Api::run_scalars -- every other job the ordinary one.  The binding survives a
// reset (a new context binds the buffer again).
// Reference side: the async call convention of the entry point (ui/Benchmark.tsx:32, submission.ts:73-78); the harness hands
// one point buffer to six calls per size (submission/miscellaneous/full_benchmarks.ts:63-68,100-105).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

namespace te_promise {

struct job_t {
  const uint8_t* points = nullptr; const uint8_t* scalars = nullptr;   // alive and unchanged until the promise settles
  uint64_t n = 0;
  uint64_t ticket = 0; bool submitted = false;
  bool failed = false;           // could not be submitted, for good (rc / err say why)
  bool bound = false;            // went down the scalars-only path (statistics)
  int rc = 0; std::string err;
  uint8_t out[64];
};

struct stats_t { uint64_t submitted_in_enter = 0, submitted_in_execute = 0, lone_runs = 0, bound_jobs = 0; int64_t max_in_flight = 0; };

// Api: static functions over an engine context type --
//   using ctx_t = ...;  using bases_t = ...;  static constexpr int ESTATE = <the engine's "capacity / call order" code>;
//   int init(const int* ids, int n, ctx_t** out);  void destroy(ctx_t*);  const char* last_error(ctx_t* or nullptr);
//   int run(ctx_t*, points, scalars, n, out);  int submit_async(ctx_t*, points, scalars, n, uint64_t* ticket);
//   int ticket_wait(ctx_t*, ticket);  int collect(ctx_t*, ticket, out);  int64_t in_flight(ctx_t*);  int64_t num_devices(ctx_t*);
//   int bind(ctx_t*, points, n, bases_t** out);  int release(ctx_t*, bases_t*);
//   int run_scalars(ctx_t*, bases_t*, scalars, out);  int submit_scalars(ctx_t*, bases_t*, scalars, uint64_t* ticket);
template <class Api> class protocol {
 public:
  using ctx_t = typename Api::ctx_t;
  using bases_t = typename Api::bases_t;

  // the thread that creates promises, before the job is handed to the pool
  void enter(job_t* j) {
    {
      std::lock_guard<std::mutex> q(qmu_);
      j->rc = 0; j->err.clear(); j->submitted = j->failed = false;
      pending_++;
      waiting_.push_back(j);
    }
    std::unique_lock<std::mutex> lk(mu_, std::try_to_lock);            // never wait for a pool thread that is inside the engine
    if (lk.owns_lock()) drain(true);
  }

  // a pool thread; the promise is settled from j->rc / j->out / j->err afterwards
  void execute(job_t* j) {
    std::unique_lock<std::mutex> lk(mu_);
    bool lone_done = false;
    if (!settled_submit(j)) {
      if ((j->rc = ensure_context(j->err)) != 0) {
        std::lock_guard<std::mutex> q(qmu_);
        j->failed = true;
        waiting_.erase(std::remove(waiting_.begin(), waiting_.end(), j), waiting_.end());
      } else {
        for (;;) {
          bool front, alone;
          { std::lock_guard<std::mutex> q(qmu_); front = !waiting_.empty() && waiting_.front() == j; alone = pending_ == 1; }
          if (front && (j->n == 0 || (alone && Api::num_devices(ctx_) > 1))) {
            // the lone call: every device works on this one MSM (or: nothing to do)
            { std::lock_guard<std::mutex> q(qmu_); waiting_.pop_front(); }
            if (j->n > 0 && is_bound(j)) { j->rc = Api::run_scalars(ctx_, bases_, j->scalars, j->out); j->bound = true; stats_.bound_jobs++; }
            else j->rc = Api::run(ctx_, j->points, j->scalars, j->n, j->out);
            if (j->rc) j->err = Api::last_error(ctx_);
            stats_.lone_runs++;
            lone_done = true;
            break;
          }
          drain(false);                                                  // oldest first; may well submit j
          if (settled_submit(j)) break;
          cv_.wait(lk);                                                  // older jobs own every work set (or an older job is ahead): one of THEIR threads collects
        }
      }
    }
    if (j->submitted) {
      ctx_t* const ctx = ctx_;                                           // (cannot change: drop_context waits for pending_ == 0)
      lk.unlock();
      const int wrc = Api::ticket_wait(ctx, j->ticket);                  // the one call that may run beside others
      lk.lock();
      j->rc = Api::collect(ctx, j->ticket, j->out);
      if (j->rc) j->err = Api::last_error(ctx);
      else if (wrc) { j->rc = wrc; j->err = "ticket_wait failed"; }
    }
    (void)lone_done;
    { std::lock_guard<std::mutex> q(qmu_); pending_--; }
    drain(false);                                                        // the freed work set goes to the oldest waiting job
    cv_.notify_all();
  }

  // waits until no promise is pending, then drops the context (compute_msm's force_recompile; a new device list)
  void reset() { std::unique_lock<std::mutex> lk(mu_); drop_context(lk); }
  void set_devices(const std::vector<int>& ids) { std::unique_lock<std::mutex> lk(mu_); drop_context(lk); devices_ = ids; }
  std::vector<int> devices() { std::lock_guard<std::mutex> lk(mu_); return devices_.empty() ? Api::default_devices() : devices_; }
  int pending() { std::lock_guard<std::mutex> q(qmu_); return pending_; }
  stats_t stats() { std::lock_guard<std::mutex> lk(mu_); return stats_; }

  // Binds `points` (n points, alive and unchanged until clear_bases / another set_bases: the caller holds the buffer) once no
  // promise is pending; points == nullptr unbinds.  Returns the engine's code; err receives its text.
  int set_bases(const uint8_t* points, uint64_t n, std::string& err) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return pending() == 0; });
    if (bases_ && ctx_) (void)Api::release(ctx_, bases_);
    bases_ = nullptr; bases_points_ = nullptr; bases_n_ = 0;
    if (!points) return 0;
    int rc = ensure_context(err);
    if (rc) return rc;
    bases_points_ = points; bases_n_ = n;
    rc = bind_now(err);
    if (rc) { bases_points_ = nullptr; bases_n_ = 0; }
    return rc;
  }
  bool has_bases() { std::lock_guard<std::mutex> lk(mu_); return bases_points_ != nullptr; }

 private:
  static bool lone_in_enter() { static const bool on = [] { const char* e = getenv("TE_MSM_LONE_IN_ENTER"); return !(e && e[0] == '0'); }(); return on; }
  bool settled_submit(const job_t* j) const { return j->submitted || j->failed; }
  bool is_bound(const job_t* j) const { return bases_ && j->points == bases_points_ && j->n == bases_n_; }
  // with mu_ held
  int bind_now(std::string& err) {
    const int rc = Api::bind(ctx_, bases_points_, bases_n_, &bases_);
    if (rc) { err = Api::last_error(ctx_); bases_ = nullptr; }
    return rc;
  }
  int ensure_context(std::string& err) {
    if (ctx_) return 0;
    const std::vector<int> ids = devices_.empty() ? Api::default_devices() : devices_;
    const int rc = Api::init(ids.data(), (int)ids.size(), &ctx_);
    if (rc) { err = Api::last_error(nullptr); ctx_ = nullptr; return rc; }
    if (bases_points_ && !bases_) { std::string e; (void)bind_now(e); }     // a new context binds the caller's buffer again (on failure: the ordinary path)
    return 0;
  }
  // with mu_ held: one attempt to turn the job into a ticket.  true = settled (submitted, or failed for good: j->rc);
  // false = every work set is taken (the caller waits for a collect and tries again)
  bool try_submit(job_t* j) {
    const bool bound = is_bound(j);
    j->rc = bound ? Api::submit_scalars(ctx_, bases_, j->scalars, &j->ticket) : Api::submit_async(ctx_, j->points, j->scalars, j->n, &j->ticket);
    if (j->rc == 0) {
      j->submitted = true; j->bound = bound;
      if (bound) stats_.bound_jobs++;
      const int64_t fl = Api::in_flight(ctx_);
      if (fl > stats_.max_in_flight) stats_.max_in_flight = fl;
      return true;
    }
    if (j->rc == Api::ESTATE && Api::in_flight(ctx_) > 0) return false;   // capacity
    j->err = Api::last_error(ctx_);
    j->failed = true;
    return true;
  }
  // with mu_ held: waiting jobs become tickets, oldest first, until one does not fit.  in_enter: a job that is alone on SEVERAL
  // devices is left to its pool thread (by then it knows whether it still is: a lone call on several devices uses all of them), and
  // an empty one always (Api::run answers it).
  void drain(bool in_enter) {
    if (!ctx_) return;                                                   // the very first call creates the context in its pool thread
    for (;;) {
      job_t* f; int pend;
      { std::lock_guard<std::mutex> q(qmu_); if (waiting_.empty()) return; f = waiting_.front(); pend = pending_; }
      if (f->n == 0) return;
      // a lone job on ONE device becomes a ticket right in enter() as well: its upload starts while libuv still hands the job to a pool
      // thread (the harness awaits every call, ui/Benchmark.tsx:32: every call is a lone one).  TE_MSM_LONE_IN_ENTER=0: left to its pool thread.
      if (in_enter && pend == 1 && (Api::num_devices(ctx_) > 1 || !lone_in_enter())) return;
      if (!in_enter && pend == 1 && Api::num_devices(ctx_) > 1) return;  // (its own thread runs it as the lone call)
      const uint64_t before = stats_.submitted_in_enter + stats_.submitted_in_execute;
      (void)before;
      if (!try_submit(f)) return;
      if (f->submitted) { if (in_enter) stats_.submitted_in_enter++; else stats_.submitted_in_execute++; }
      { std::lock_guard<std::mutex> q(qmu_); waiting_.pop_front(); }
      cv_.notify_all();                                                  // f's thread may be waiting to learn that it is a ticket now
    }
  }
  void drop_context(std::unique_lock<std::mutex>& lk) {
    cv_.wait(lk, [this] { return pending() == 0; });
    if (ctx_) { Api::destroy(ctx_); ctx_ = nullptr; bases_ = nullptr; }   // (the records went with the context; bases_points_ stays: bound again on the next context)
  }

  std::mutex mu_;                 // guards the context (it is not thread-safe) and everything below except the queue
  std::mutex qmu_;                // guards waiting_ and pending_ only; never held across an engine call (lock order: mu_, then qmu_)
  std::condition_variable cv_;    // a work set became free / a waiting job became a ticket / the context went idle (with mu_)
  ctx_t* ctx_ = nullptr;
  std::vector<int> devices_;      // empty: Api::default_devices()
  std::deque<job_t*> waiting_;    // entered, not submitted yet, oldest first
  int pending_ = 0;               // promises that are not settled yet
  bases_t* bases_ = nullptr;      // the bound point set of ctx_ (nullptr: none, or not bound on this context yet)
  const uint8_t* bases_points_ = nullptr; uint64_t bases_n_ = 0;          // the caller's buffer that set_bases named
  stats_t stats_;
};

}  // namespace te_promise
