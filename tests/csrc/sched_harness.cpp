// sched_harness.cpp -- ThreadSanitizer harness of the engine's multi-threaded host code, with a stand-in for the device.
//
// What runs here is the product's own code: te_sched::worker_t and the ticket bookkeeping (csrc/host_sched.hpp, the code
// behind te_msm_submit / te_msm_submit_async / te_msm_ticket_wait / te_msm_collect), the lock protocol of the N-API addon
// (js/promise_protocol.hpp, the code behind concurrent compute_msm promises) and the host tail with its multi-thread row
// merge (csrc/host_tail.hpp: te_host::merge_window_rows, as the multi-device te_msm_run uses it).  What is faked is the GPU:
// a "device" is a thread that computes an MSM's partial rows with the device arithmetic compiled for the host
// (fpc_partial_rows, tests/csrc/fpcheck.cpp) and then signals an event, the way a stream signals ev_result.
// The fake engine below mirrors csrc/te_msm.hip's submit_host / te_msm_ticket_wait / te_msm_collect line by line around
// those shared pieces.  Built with -fsanitize=thread by tests/sanitize/Makefile; any report fails the run.
// SURVEY.md section 5 "race detection / sanitizers" (the reference is single-threaded JavaScript).
#include "fpcheck.cpp"
#include "../../webgpu-msm-twisted-edwards_amd/csrc/host_tail.hpp"
#include "../../webgpu-msm-twisted-edwards_amd/csrc/synth.hpp"
#include "../../webgpu-msm-twisted-edwards_amd/csrc/host_sched.hpp"
#include "../../webgpu-msm-twisted-edwards_amd/js/promise_protocol.hpp"
#include <stdio.h>
#include <atomic>
#include <chrono>
#include <string>

namespace {

constexpr int LANES = 2;         // upload lanes per fake device (the engine: option "upload_threads", 4)
constexpr int SETS = 3;          // work sets per fake device (the engine has TE_MSM_WORKSETS = 8; fewer here so that capacity waits happen)
constexpr int WBITS = 4;         // window bits of every fake MSM (64 windows of 8 buckets: the arithmetic runs under the sanitizer; 4 x 64 = 256 bits, so an all-ones scalar leaves a final carry)
constexpr int NWIN = (256 + WBITS - 1) / WBITS;
enum { FK_OK = 0, FK_EINVAL = -1, FK_EDEVICE = -2, FK_ESCALAR = -3, FK_ESTATE = -4 };      // include/te_msm.h: TE_MSM_E*

// what a recorded HIP event is to the engine: "the rows are in host memory"
struct fake_event {
  std::mutex mu; std::condition_variable cv; bool done = true;
  void reset() { std::lock_guard<std::mutex> lk(mu); done = false; }
  void signal() { { std::lock_guard<std::mutex> lk(mu); done = true; } cv.notify_all(); }
  void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
};
struct fake_set {
  te_sched::slot_t slot; std::string job_err;
  std::vector<uint8_t> staged_points, staged_scalars, rows;     // the set's staging area and its pinned row block
  int flag = 0;                                                  // final-carry flag of its last MSM
  std::unique_ptr<fake_event> ev_result{new fake_event()};
};
struct fake_dev {
  int device = 0, in_flight = 0;
  fake_set ws[SETS];
  std::unique_ptr<te_sched::worker_t> stream{new te_sched::worker_t()};      // the device itself: works the enqueued MSMs off in order
};
struct fake_ctx {
  std::vector<fake_dev> devs;
  uint64_t next_ticket = 1; int last_dev = -1;
  std::vector<std::unique_ptr<te_sched::worker_t>> workers;
  std::vector<std::unique_ptr<te_sched::worker_t>> lanes; uint64_t next_lane = 0;     // upload lanes of asynchronous tickets
  std::string err; std::mutex err_mu;
  std::atomic<int> fail_next_upload{0};                          // test hook: the next asynchronous upload fails with FK_EDEVICE
};
int set_err(fake_ctx* c, int code, const char* msg) { std::lock_guard<std::mutex> lk(c->err_mu); c->err = msg; return code; }

// upload + enqueue of one whole MSM on work set ws of device d (csrc/te_msm.hip: enqueue_host_slice): reads the CALLER's
// buffers on the calling thread, then the device works on the staged copy
int enqueue_host(fake_ctx* c, fake_dev& d, fake_set& ws, const uint8_t* points, const uint8_t* scalars, uint64_t n) {
  if (c->fail_next_upload.exchange(0)) return set_err(c, FK_EDEVICE, "hipMemcpyAsync failed (injected)");
  ws.staged_points.assign(points, points + 64 * n);
  ws.staged_scalars.assign(scalars, scalars + 32 * n);
  ws.ev_result->reset();
  fake_set* w = &ws;
  d.stream->post([w, n]() -> int {
    w->rows.assign((size_t)NWIN * 720, 0);
    w->flag = fpc_partial_rows(w->staged_points.data(), w->staged_scalars.data(), n, WBITS, 0, 1, w->rows.data()) ? 1 : 0;
    w->ev_result->signal();
    return 0;
  });
  return 0;
}

// ---- the C-ABI of the stand-in engine (same shapes as include/te_msm.h)
int fk_init(const int* ids, int n, fake_ctx** out) {
  if (!ids || n < 1 || n > 64) return FK_EINVAL;
  fake_ctx* c = new fake_ctx();
  c->devs = std::vector<fake_dev>((size_t)n);
  for (int i = 0; i < n; i++) c->devs[(size_t)i].device = ids[i];
  *out = c;
  return 0;
}
void fk_destroy(fake_ctx* c) {
  c->lanes.clear();                                              // finishes the uploads that were never collected
  c->workers.clear();
  for (auto& d : c->devs) d.stream.reset();                      // ... and the device work behind them
  delete c;
}
int fk_submit(fake_ctx* c, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint64_t* ticket, bool async) {
  if (!points || !scalars || !n) return set_err(c, FK_EINVAL, "bad arguments");
  const int di = te_sched::pick_device_of(*c, SETS, -1);
  if (di < 0) return set_err(c, FK_ESTATE, "every work set has an MSM in flight: collect one first");
  fake_dev& d = c->devs[(size_t)di];
  const int wi = te_sched::free_set_index(d, SETS);
  if (wi < 0) return set_err(c, FK_ESTATE, "every work set has an MSM in flight: collect one first");
  fake_set& ws = d.ws[wi];
  if (!async) {
    if (int rc = enqueue_host(c, d, ws, points, scalars, n)) return rc;
    te_sched::hand_out(*c, di, ws, ticket);
    return 0;
  }
  ws.job_err.clear();
  fake_set* wsp = &ws; fake_dev* dp = &d;
  te_sched::job_ref job = te_sched::next_lane_of(*c, (size_t)di, LANES).post([c, dp, wsp, points, scalars, n]() -> int {
    const int rc = enqueue_host(c, *dp, *wsp, points, scalars, n);
    if (rc) { std::lock_guard<std::mutex> lk(c->err_mu); wsp->job_err = c->err; }
    return rc;
  });
  te_sched::hand_out(*c, di, ws, ticket, std::move(job));
  return 0;
}
int fk_ticket_wait(fake_ctx* c, uint64_t ticket) {
  int di = -1;
  fake_set* ws = te_sched::find_ticket(*c, ticket, &di);
  if (!ws) return set_err(c, FK_ESTATE, "no such ticket in flight");
  if (const int rc = te_sched::await_job(*ws)) return rc;
  ws->ev_result->wait();
  return 0;
}
int fk_collect(fake_ctx* c, uint64_t ticket, uint8_t out[64]) {
  int di = -1;
  fake_set* wsp = te_sched::find_ticket(*c, ticket, &di);
  if (!wsp) return set_err(c, FK_ESTATE, "no such ticket in flight");
  fake_set& ws = *wsp;
  if (const int jrc = te_sched::await_job(ws)) {
    te_sched::retire(*c, di, ws);
    return set_err(c, jrc, ws.job_err.empty() ? "the asynchronous submit failed" : ws.job_err.c_str());
  }
  ws.ev_result->wait();
  te_sched::retire(*c, di, ws);
  if (ws.flag) return set_err(c, FK_ESCALAR, "final carry is 1");
  te_host::horner_to_affine(ws.rows.data(), WBITS, WBITS - 1, NWIN, out);
  return 0;
}
// the lone call on several devices (csrc/te_msm.hip: run_host_sharded): point slices, one host thread per device, the sets' rows
// merged window by window BY THOSE THREADS (distinct elements of merged[] / present[]), one fold
int fk_run(fake_ctx* c, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out[64]) {
  if (n == 0) { memset(out, 0, 64); out[32] = 1; return 0; }
  const size_t D = std::min<size_t>(c->devs.size(), (size_t)n);
  const uint64_t per = (n + D - 1) / D;
  std::vector<int> wsel(D);
  for (size_t i = 0; i < D; i++) { wsel[i] = te_sched::free_set_index(c->devs[i], SETS); if (wsel[i] < 0) return set_err(c, FK_ESTATE, "every work set holds a submitted MSM"); }
  auto slice = [&](size_t i) -> int {
    const uint64_t lo = std::min<uint64_t>(n, per * i), hi = std::min<uint64_t>(n, lo + per);
    fake_set& ws = c->devs[i].ws[wsel[i]];
    if (hi == lo) { ws.rows.assign((size_t)NWIN * 720, 0); ws.flag = 0; return 0; }
    if (int rc = enqueue_host(c, c->devs[i], ws, points + 64 * lo, scalars + 32 * lo, hi - lo)) return rc;
    ws.ev_result->wait();
    return 0;
  };
  std::vector<te_sched::job_ref> jobs(D);
  for (size_t i = 1; i < D; i++) jobs[i] = te_sched::worker_of(*c, i).post([&slice, i] { return slice(i); });
  int rc = slice(0);
  for (size_t i = 1; i < D; i++) { const int r = te_sched::worker_of(*c, i).wait(jobs[i]); if (!rc) rc = r; }
  if (rc) return rc;
  std::vector<const uint8_t*> sets;
  for (size_t i = 0; i < D; i++) { if (c->devs[i].ws[wsel[i]].flag) return set_err(c, FK_ESCALAR, "final carry is 1"); sets.push_back(c->devs[i].ws[wsel[i]].rows.data()); }
  const int ns = (int)sets.size();
  std::vector<te_host::Pt> merged((size_t)NWIN * 5); std::vector<uint8_t> present((size_t)NWIN, 0);
  auto merge = [&](size_t t) -> int { for (int w = (int)t; w < NWIN; w += (int)D) te_host::merge_window_rows(sets.data(), ns, w, merged.data(), present.data()); return 0; };
  for (size_t i = 1; i < D; i++) jobs[i] = te_sched::worker_of(*c, i).post([&merge, i] { return merge(i); });
  (void)merge(0);
  for (size_t i = 1; i < D; i++) (void)te_sched::worker_of(*c, i).wait(jobs[i]);
  te_host::horner_to_affine_points(merged.data(), present.data(), WBITS, WBITS - 1, NWIN, out);
  return 0;
}
int64_t fk_in_flight(fake_ctx* c) { int64_t t = 0; for (auto& d : c->devs) t += d.in_flight; return t; }

// a bound point set of the stand-in engine: it only remembers the caller's buffer (the real one keeps records on the devices)
struct fake_bases { const uint8_t* points; uint64_t n; };
struct FakeApi {
  using ctx_t = fake_ctx;
  using bases_t = fake_bases;
  static constexpr int ESTATE = FK_ESTATE;
  static int bind(fake_ctx*, const uint8_t* p, uint64_t n, fake_bases** out) { *out = new fake_bases{p, n}; return 0; }
  static int release(fake_ctx*, fake_bases* b) { delete b; return 0; }
  static int run_scalars(fake_ctx* c, fake_bases* b, const uint8_t* s, uint8_t* out) { return fk_run(c, b->points, s, b->n, out); }
  static int submit_scalars(fake_ctx* c, fake_bases* b, const uint8_t* s, uint64_t* t) { return fk_submit(c, b->points, s, b->n, t, true); }
  static std::vector<int> default_devices() { return {0}; }
  static int init(const int* ids, int n, fake_ctx** out) { return fk_init(ids, n, out); }
  static void destroy(fake_ctx* c) { fk_destroy(c); }
  static const char* last_error(fake_ctx* c) {                   // as te_msm_last_error: a copy taken under the lock
    if (!c) return "init failed";
    thread_local std::string copy;
    { std::lock_guard<std::mutex> lk(c->err_mu); copy = c->err; }
    return copy.c_str();
  }
  static int run(fake_ctx* c, const uint8_t* p, const uint8_t* s, uint64_t n, uint8_t* out) { return fk_run(c, p, s, n, out); }
  static int submit_async(fake_ctx* c, const uint8_t* p, const uint8_t* s, uint64_t n, uint64_t* t) { return fk_submit(c, p, s, n, t, true); }
  static int ticket_wait(fake_ctx* c, uint64_t t) { return fk_ticket_wait(c, t); }
  static int collect(fake_ctx* c, uint64_t t, uint8_t* out) { return fk_collect(c, t, out); }
  static int64_t in_flight(fake_ctx* c) { return fk_in_flight(c); }
  static int64_t num_devices(fake_ctx* c) { return (int64_t)c->devs.size(); }
};

int fails = 0;
std::mutex fail_mu;
#define CHECK(cond, ...) do { if (!(cond)) { std::lock_guard<std::mutex> lk_(fail_mu); fails++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

struct msm_case { std::vector<uint8_t> pts, sc; uint64_t n; uint8_t want[64]; };
std::vector<msm_case> make_cases() {
  std::vector<msm_case> cs;
  for (uint64_t n : {1ull, 7ull, 24ull, 33ull, 48ull}) {
    msm_case m; m.n = n; m.pts.resize(64 * n); m.sc.resize(32 * n);
    te_host::synth_points(500 + n, n, m.pts.data()); te_host::synth_scalars(600 + n, n, m.sc.data());
    std::vector<uint8_t> rows((size_t)NWIN * 720, 0);
    fpc_partial_rows(m.pts.data(), m.sc.data(), n, WBITS, 0, 1, rows.data());
    te_host::horner_to_affine(rows.data(), WBITS, WBITS - 1, NWIN, m.want);
    cs.push_back(std::move(m));
  }
  return cs;
}

// 1. worker_t alone: posts and waits from several threads, drain, destruction with work queued
void test_worker() {
  std::atomic<int> ran{0};
  {
    te_sched::worker_t w;
    std::vector<std::thread> th;
    for (int t = 0; t < 4; t++) th.emplace_back([&] {
      for (int i = 0; i < 200; i++) { auto j = w.post([&ran, i] { ran++; return i; }); CHECK(w.wait(j) == i, "job status"); CHECK(w.wait(j) == i, "second wait"); }
    });
    for (auto& x : th) x.join();
    w.drain();
    CHECK(ran == 800, "every job ran once");
    for (int i = 0; i < 50; i++) w.post([&ran] { std::this_thread::sleep_for(std::chrono::microseconds(50)); ran++; return 0; });
  }                                                              // ~worker_t finishes what was posted
  CHECK(ran == 850, "the destructor finished the queue (%d)", ran.load());
  printf("ok worker_t\n");
}

// 2. the C-ABI protocol a multi-threaded host follows (include/te_msm.h): submit and collect under the caller's lock,
//    te_msm_ticket_wait outside it; blocking and asynchronous submits mixed; an upload that fails; capacity
void test_tickets(const std::vector<msm_case>& cs, int D) {
  std::vector<int> ids((size_t)D, 0);
  fake_ctx* c = nullptr;
  CHECK(fk_init(ids.data(), D, &c) == 0, "init");
  std::mutex mu; std::condition_variable cv;
  std::atomic<int> done{0}, refused{0};
  std::vector<std::thread> th;
  for (int t = 0; t < 6; t++) th.emplace_back([&, t] {
    for (int i = 0; i < 12; i++) {
      const msm_case& m = cs[(size_t)(t + i) % cs.size()];
      uint64_t ticket = 0; int rc;
      {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
          rc = fk_submit(c, m.pts.data(), m.sc.data(), m.n, &ticket, (t + i) % 3 != 0);
          if (rc != FK_ESTATE) break;
          refused++;
          cv.wait(lk);                                           // every set is taken: wait for somebody's collect
        }
      }
      CHECK(rc == 0, "submit rc %d", rc);
      if (i % 2) CHECK(fk_ticket_wait(c, ticket) == 0, "ticket_wait");      // beside the others, no lock
      uint8_t out[64];
      { std::lock_guard<std::mutex> lk(mu); rc = fk_collect(c, ticket, out); }
      cv.notify_all();
      CHECK(rc == 0 && memcmp(out, m.want, 64) == 0, "result of ticket %llu (rc %d)", (unsigned long long)ticket, rc);
      done++;
    }
  });
  for (auto& x : th) x.join();
  CHECK(done == 72 && fk_in_flight(c) == 0, "all collected");
  // an asynchronous upload that fails: reported by collect, the ticket is freed, the set is usable again
  c->fail_next_upload = 1;
  uint64_t t1 = 0, t2 = 0; uint8_t out[64];
  CHECK(fk_submit(c, cs[2].pts.data(), cs[2].sc.data(), cs[2].n, &t1, true) == 0, "submit");
  CHECK(fk_submit(c, cs[3].pts.data(), cs[3].sc.data(), cs[3].n, &t2, true) == 0, "submit");
  const int r1 = fk_collect(c, t1, out), r2 = fk_collect(c, t2, out);
  CHECK((r1 == FK_EDEVICE) != (r2 == FK_EDEVICE) && (r1 == 0 || r2 == 0), "one of the two uploads failed (%d, %d)", r1, r2);
  CHECK(fk_in_flight(c) == 0 && fk_collect(c, t1, out) == FK_ESTATE, "a ticket is consumed once");
  // a scalar out of range belongs to its ticket
  std::vector<uint8_t> bad(cs[2].sc); memset(&bad[32 * 3], 0xff, 32);
  CHECK(fk_submit(c, cs[2].pts.data(), bad.data(), cs[2].n, &t1, true) == 0 && fk_submit(c, cs[2].pts.data(), cs[2].sc.data(), cs[2].n, &t2, false) == 0, "submit");
  CHECK(fk_collect(c, t2, out) == 0 && memcmp(out, cs[2].want, 64) == 0 && fk_collect(c, t1, out) == FK_ESCALAR, "error in one ticket only");
  // a lone call (point slices on the devices' host threads, work sets no ticket owns) beside asynchronous tickets whose uploads
  // are still running on the lanes
  {
    uint64_t tk[4]; uint8_t o2[64];
    for (int i = 0; i < 4; i++) CHECK(fk_submit(c, cs[(size_t)i].pts.data(), cs[(size_t)i].sc.data(), cs[(size_t)i].n, &tk[i], true) == 0 || D * SETS < 4, "submit beside a lone call");
    if (D * SETS > 4) CHECK(fk_run(c, cs[4].pts.data(), cs[4].sc.data(), cs[4].n, o2) == 0 && memcmp(o2, cs[4].want, 64) == 0, "lone call beside tickets");
    for (int i = 0; i < 4; i++) if (te_sched::find_ticket(*c, tk[i])) CHECK(fk_collect(c, tk[i], o2) == 0 && memcmp(o2, cs[(size_t)i].want, 64) == 0, "ticket beside a lone call");
  }
  // tickets never collected: destroy finishes their uploads first
  CHECK(fk_submit(c, cs[4].pts.data(), cs[4].sc.data(), cs[4].n, &t1, true) == 0, "submit");
  fk_destroy(c);
  printf("ok tickets on %d device(s): 72 MSMs from 6 threads, %d capacity waits\n", D, refused.load());
}

// 3. the N-API addon's lock protocol (js/promise_protocol.hpp) over the stand-in engine: one thread creates promises in
//    bursts (the JavaScript thread), a small pool settles them; lone calls on several devices; a reset in between
void test_promises(const std::vector<msm_case>& cs, int D, int pool_threads) {
  te_promise::protocol<FakeApi> proto;
  proto.set_devices(std::vector<int>((size_t)D, 0));
  std::mutex qmu; std::condition_variable qcv; std::deque<te_promise::job_t*> q; bool quit = false;
  std::atomic<int> settled{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < pool_threads; t++) pool.emplace_back([&] {
    for (;;) {
      te_promise::job_t* j = nullptr;
      { std::unique_lock<std::mutex> lk(qmu); qcv.wait(lk, [&] { return quit || !q.empty(); }); if (q.empty()) return; j = q.front(); q.pop_front(); }
      proto.execute(j);
      settled++;
    }
  });
  auto burst = [&](int k, int first) {
    std::vector<te_promise::job_t> jobs((size_t)k);
    for (int i = 0; i < k; i++) {
      const msm_case& m = cs[(size_t)(first + i) % cs.size()];
      jobs[(size_t)i].points = m.pts.data(); jobs[(size_t)i].scalars = m.sc.data(); jobs[(size_t)i].n = m.n;
      proto.enter(&jobs[(size_t)i]);                                        // the JavaScript thread
      { std::lock_guard<std::mutex> lk(qmu); q.push_back(&jobs[(size_t)i]); }
      qcv.notify_one();
    }
    const int target = settled + 0;
    (void)target;
    while (proto.pending() > 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
    (void)proto.stats();        // (takes the engine's lock: the last pool thread has LEFT execute(), so the next enter() finds the lock free --
                                //  the addon gets the same from libuv: a job's completion callback runs after its execute() has returned)
    // (pending() == 0: every execute() has left the protocol; the results below were written before that under its lock)
    for (int i = 0; i < k; i++) {
      const msm_case& m = cs[(size_t)(first + i) % cs.size()];
      CHECK(jobs[(size_t)i].rc == 0 && memcmp(jobs[(size_t)i].out, m.want, 64) == 0, "promise %d of a burst of %d (rc %d: %s)", i, k, jobs[(size_t)i].rc, jobs[(size_t)i].err.c_str());
    }
  };
  burst(1, 0);                                                              // the very first call creates the context
  burst(1, 1);                                                              // a lone call: point slices on several devices
  burst(2 * D, 2);
  burst(D * SETS + 5, 0);                                                   // more than there are work sets: capacity waits inside execute()
  burst(3 * D * SETS + 1, 1);                                               // far more: with a pool of ONE thread round 5's protocol hung here (every
                                                                            // set taken by younger tickets, the oldest job parked on capacity)
  {
    const te_promise::stats_t st = proto.stats();
    CHECK(st.max_in_flight <= (int64_t)D * SETS && st.max_in_flight >= 1, "tickets in flight (%lld)", (long long)st.max_in_flight);
    CHECK(st.submitted_in_enter + st.submitted_in_execute + st.lone_runs >= (uint64_t)(2 + 2 * D + D * SETS + 5), "every job went one way or the other");
  }
  // the harness's call pattern (every call awaited: bursts of one): on ONE device a lone job becomes a ticket in enter() -- the engine's
  // lock is free between the calls --, on several it stays the lone call of its pool thread (round 6)
  {
    const te_promise::stats_t a = proto.stats();
    for (int i = 0; i < 6; i++) burst(1, i);
    const te_promise::stats_t b = proto.stats();
    if (D == 1) CHECK(b.submitted_in_enter == a.submitted_in_enter + 6 && b.lone_runs == a.lone_runs, "awaited calls on one device are tickets from enter() (%llu -> %llu)",
                      (unsigned long long)a.submitted_in_enter, (unsigned long long)b.submitted_in_enter);
    else CHECK(b.lone_runs == a.lone_runs + 6 && b.submitted_in_enter == a.submitted_in_enter, "awaited calls on %d devices are lone calls", D);
  }
  // resident bases: jobs over the bound buffer take the scalars-only path, the others the ordinary one; the binding survives a reset
  {
    std::string err;
    const msm_case& mb = cs[3];
    CHECK(proto.set_bases(mb.pts.data(), mb.n, err) == 0 && proto.has_bases(), "set_bases (%s)", err.c_str());
    const uint64_t before = proto.stats().bound_jobs;
    burst((int)cs.size() + 3, 0);                                           // case 3 (twice) is the bound buffer, the others are not
    CHECK(proto.stats().bound_jobs >= before + 1, "jobs over the bound buffer took the scalars-only path");
    proto.reset();
    burst(1, 3);                                                            // a new context binds the buffer again
    CHECK(proto.stats().bound_jobs >= before + 2, "bound again after a reset");
    CHECK(proto.set_bases(nullptr, 0, err) == 0 && !proto.has_bases(), "unbind");
  }
  proto.reset();                                                            // compute_msm's force_recompile
  burst(3, 3);
  proto.set_devices({0});
  burst(7, 1);
  { std::lock_guard<std::mutex> lk(qmu); quit = true; }
  qcv.notify_all();
  for (auto& x : pool) x.join();
  proto.reset();
  printf("ok promises on %d device(s), pool of %d: %d settled\n", D, pool_threads, settled.load());
}

}  // namespace

int main() {
  if (!te_host::tail_selftest()) { printf("host tail self-test failed\n"); return 1; }
  const std::vector<msm_case> cs = make_cases();
  test_worker();
  for (int D : {1, 2, 4}) test_tickets(cs, D);
  test_promises(cs, 1, 1);                       // UV_THREADPOOL_SIZE=1: the deterministic hang of round 5's protocol (advisor)
  test_promises(cs, 2, 1);
  test_promises(cs, 1, 4);
  test_promises(cs, 4, 2);
  test_promises(cs, 3, 6);
  printf(fails ? "sched_harness: %d FAILURES\n" : "sched_harness: all checks passed\n", fails);
  return fails ? 1 : 0;
}
