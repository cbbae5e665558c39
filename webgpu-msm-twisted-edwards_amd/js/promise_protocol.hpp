// promise_protocol.hpp -- how concurrent compute_msm() promises map onto the engine's tickets: the lock protocol of the
// N-API addon, free of N-API, so that addon.cc (the real engine through the C-ABI) and tests/csrc/sched_harness.cpp (a
// stand-in engine, under ThreadSanitizer) run the SAME code.
//
// Threads: ONE thread calls enter() for every new promise (the JavaScript thread); any number of pool threads call
// execute() for the jobs that were entered (libuv's pool).  The engine's context is not thread-safe: every call into it
// happens under `mu`, except Api::ticket_wait -- the one entry point the C-ABI allows beside others (include/te_msm.h).
//
//   enter    other promises are pending and the context exists: the job becomes a ticket right here (submit_async returns
//            at once: device and work set are picked, the upload is handed to that device's host thread).  So the number
//            of MSMs in flight is bounded by the engine's work sets, not by the size of the pool.
//   execute  not submitted yet: the first promise of a burst (by now it can see whether it is alone: a LONE call on
//            several devices uses all of them for its one MSM, Api::run = point slices), or one that found every work
//            set taken (it waits for a collect), or the very first call (it creates the context).
//            Then: wait for the ticket outside the lock, collect under it.
// Reference side: the async call convention of the entry point (ui/Benchmark.tsx:32, submission.ts:73-78).
#pragma once
#include <stdint.h>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

namespace te_promise {

struct job_t {
  const uint8_t* points = nullptr; const uint8_t* scalars = nullptr;   // alive and unchanged until the promise settles
  uint64_t n = 0;
  uint64_t ticket = 0; bool submitted = false;
  int rc = 0; std::string err;
  uint8_t out[64];
};

// Api: static functions over an engine context type --
//   using ctx_t = ...;  static constexpr int ESTATE = <the engine's "capacity / call order" code>;
//   int init(const int* ids, int n, ctx_t** out);  void destroy(ctx_t*);  const char* last_error(ctx_t* or nullptr);
//   int run(ctx_t*, points, scalars, n, out);  int submit_async(ctx_t*, points, scalars, n, uint64_t* ticket);
//   int ticket_wait(ctx_t*, ticket);  int collect(ctx_t*, ticket, out);  int64_t in_flight(ctx_t*);  int64_t num_devices(ctx_t*);
template <class Api> class protocol {
 public:
  using ctx_t = typename Api::ctx_t;

  // the thread that creates promises, before the job is handed to the pool
  void enter(job_t* j) {
    std::lock_guard<std::mutex> lk(mu_);
    pending_++;
    if (ctx_ && j->n > 0 && pending_ > 1) (void)try_submit(j);        // a failure or a full house is dealt with in execute()
    j->rc = 0; j->err.clear();
  }

  // a pool thread; the promise is settled from j->rc / j->out / j->err afterwards
  void execute(job_t* j) {
    std::unique_lock<std::mutex> lk(mu_);
    if (!j->submitted) {
      if ((j->rc = ensure_context(j->err)) == 0) {
        if (j->n == 0 || (Api::num_devices(ctx_) > 1 && pending_ == 1)) {
          j->rc = Api::run(ctx_, j->points, j->scalars, j->n, j->out);   // the lone call: every device works on this one MSM
          if (j->rc) j->err = Api::last_error(ctx_);
        } else {
          while (!try_submit(j)) cv_.wait(lk);                           // every work set is taken: wait for a collect
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
    pending_--;
    cv_.notify_all();
  }

  // waits until no promise is pending, then drops the context (compute_msm's force_recompile; a new device list)
  void reset() { std::unique_lock<std::mutex> lk(mu_); drop_context(lk); }
  void set_devices(const std::vector<int>& ids) { std::unique_lock<std::mutex> lk(mu_); drop_context(lk); devices_ = ids; }
  std::vector<int> devices() { std::lock_guard<std::mutex> lk(mu_); return devices_.empty() ? Api::default_devices() : devices_; }
  int pending() { std::lock_guard<std::mutex> lk(mu_); return pending_; }

 private:
  // with mu_ held
  int ensure_context(std::string& err) {
    if (ctx_) return 0;
    const std::vector<int> ids = devices_.empty() ? Api::default_devices() : devices_;
    const int rc = Api::init(ids.data(), (int)ids.size(), &ctx_);
    if (rc) { err = Api::last_error(nullptr); ctx_ = nullptr; }
    return rc;
  }
  // with mu_ held: one attempt to turn the job into a ticket.  true = settled (submitted, or failed for good: j->rc);
  // false = every work set is taken (the caller waits for a collect and tries again)
  bool try_submit(job_t* j) {
    j->rc = Api::submit_async(ctx_, j->points, j->scalars, j->n, &j->ticket);
    if (j->rc == 0) { j->submitted = true; return true; }
    if (j->rc == Api::ESTATE && Api::in_flight(ctx_) > 0) return false;   // capacity
    j->err = Api::last_error(ctx_);
    return true;
  }
  void drop_context(std::unique_lock<std::mutex>& lk) {
    cv_.wait(lk, [this] { return pending_ == 0; });
    if (ctx_) { Api::destroy(ctx_); ctx_ = nullptr; }
  }

  std::mutex mu_;                 // guards the context (it is not thread-safe) and the fields below
  std::condition_variable cv_;    // a work set became free / the context went idle
  ctx_t* ctx_ = nullptr;
  std::vector<int> devices_;      // empty: Api::default_devices()
  int pending_ = 0;               // promises that are not settled yet
};

}  // namespace te_promise
