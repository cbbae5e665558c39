// addon.cc -- N-API binding of libtemsm.so for the reference's host language (JavaScript / TypeScript).
//
// Exposes  msmNative(points: Buffer, scalars: Buffer): Promise<Buffer /*64 bytes: x || y little-endian*/>
// which compute_msm.js wraps into the reference's entry point
//   compute_msm(bufferPoints, bufferScalars, log_result, force_recompile): Promise<{x: bigint, y: bigint}>
// (submission/submission.ts:73-78).  The MSM runs on a libuv worker thread (napi_create_async_work), so the
// JS event loop is not blocked -- the reference's call is async for the same reason (ui/Benchmark.tsx:32).
// Errors reject the promise, as the reference's `throw`s do (implementation/cuzk/gpu.ts:19-22).
//
// Devices: TE_MSM_DEVICES="0,1,2,3" (or setDevices([0,1,2,3])) puts several GPUs of the node behind the one entry point
// (a context of n_dev > 1 devices, include/te_msm.h).  Default: device 0.
// Concurrency: promises in flight at the same time (a prover that does not await each call) become TICKETS of the engine --
// te_msm_submit_async under the lock (it returns at once: the upload runs on the chosen device's host thread),
// te_msm_ticket_wait outside it on a libuv pool thread, te_msm_collect under it again.  On one device the upload of one
// MSM overlaps the device work of the previous ones; on D devices every ticket is a whole MSM on the device with the fewest
// in flight -- D uploads on D PCIe links at once.  Calls made while others are pending are submitted right from the
// JavaScript thread, so the number in flight is not bounded by libuv's pool (4 threads unless UV_THREADPOOL_SIZE says
// otherwise) but by the engine's work sets (TE_MSM_WORKSETS per device); beyond that, calls queue inside their pool thread.
// The LONE call on several devices -- nothing else pending when it starts -- uses all of them for its one MSM
// (te_msm_run: point slices, one upload thread per device): the latency form.
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/te_msm.h"
#include "promise_protocol.hpp"

namespace {

// the engine behind te_promise::protocol (js/promise_protocol.hpp holds the lock protocol itself)
struct EngineApi {
  using ctx_t = te_ctx;
  using bases_t = te_bases;
  static constexpr int ESTATE = TE_MSM_ESTATE;
  static std::vector<int> default_devices() {        // TE_MSM_DEVICES="0,1,..", else device 0
    std::vector<int> ids;
    const char* e = getenv("TE_MSM_DEVICES");
    if (e && *e) {
      const char* q = e;
      while (*q) { char* end = nullptr; const long v = strtol(q, &end, 10); if (end == q) break; ids.push_back((int)v); q = *end == ',' ? end + 1 : end; }
    }
    if (ids.empty()) ids.push_back(0);
    return ids;
  }
  static int init(const int* ids, int n, te_ctx** out) { return te_msm_init(ids, n, out); }
  static void destroy(te_ctx* c) { te_msm_destroy(c); }
  static const char* last_error(te_ctx* c) { return te_msm_last_error(c); }
  static int run(te_ctx* c, const uint8_t* p, const uint8_t* s, uint64_t n, uint8_t* out) { return te_msm_run(c, p, s, n, out); }
  static int submit_async(te_ctx* c, const uint8_t* p, const uint8_t* s, uint64_t n, uint64_t* t) { return te_msm_submit_async(c, p, s, n, t); }
  static int ticket_wait(te_ctx* c, uint64_t t) { return te_msm_ticket_wait(c, t); }
  static int collect(te_ctx* c, uint64_t t, uint8_t* out) { return te_msm_collect(c, t, out); }
  static int64_t in_flight(te_ctx* c) { int64_t v = 0; te_msm_get_option(c, "in_flight", &v); return v; }
  static int64_t num_devices(te_ctx* c) { int64_t v = 1; te_msm_get_option(c, "num_devices", &v); return v; }
  // resident bases (include/te_msm.h): the opt-in behind setBases(buffer)
  static int bind(te_ctx* c, const uint8_t* p, uint64_t n, te_bases** out) { return te_msm_bind_points(c, p, n, out); }
  static int release(te_ctx* c, te_bases* b) { return te_msm_release_points(c, b); }
  static int run_scalars(te_ctx* c, te_bases* b, const uint8_t* s, uint8_t* out) { return te_msm_run_scalars(c, b, s, out); }
  static int submit_scalars(te_ctx* c, te_bases* b, const uint8_t* s, uint64_t* t) { return te_msm_submit_scalars(c, b, s, t); }
};
te_promise::protocol<EngineApi> g_proto;
napi_ref g_bases_ref = nullptr;       // the Buffer that setBases bound: kept alive, so that its address cannot become another object's

struct Job {
  napi_async_work work = nullptr;
  napi_deferred deferred = nullptr;
  napi_ref points_ref = nullptr, scalars_ref = nullptr;   // keep the JS Buffers alive until completion (asynchronous uploads read them)
  te_promise::job_t j;
};

void Execute(napi_env, void* data) { g_proto.execute(&static_cast<Job*>(data)->j); }

void Complete(napi_env env, napi_status, void* data) {
  Job* j = static_cast<Job*>(data);
  if (j->j.rc == 0) {
    napi_value buf; void* dst = nullptr;
    napi_create_buffer_copy(env, 64, j->j.out, &dst, &buf);
    napi_resolve_deferred(env, j->deferred, buf);
  } else {
    napi_value msg, errv;
    std::string m = "te_msm error " + std::to_string(j->j.rc) + ": " + j->j.err;
    napi_create_string_utf8(env, m.c_str(), m.size(), &msg);
    napi_create_error(env, nullptr, msg, &errv);
    napi_reject_deferred(env, j->deferred, errv);
  }
  napi_delete_reference(env, j->points_ref);
  napi_delete_reference(env, j->scalars_ref);
  napi_delete_async_work(env, j->work);
  delete j;
}

napi_value MsmNative(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  bool is0 = false, is1 = false;
  if (argc >= 2) { napi_is_buffer(env, argv[0], &is0); napi_is_buffer(env, argv[1], &is1); }
  if (!is0 || !is1) { napi_throw_type_error(env, nullptr, "msmNative(points: Buffer, scalars: Buffer)"); return nullptr; }
  void *p = nullptr, *s = nullptr; size_t pl = 0, sl = 0;
  napi_get_buffer_info(env, argv[0], &p, &pl);
  napi_get_buffer_info(env, argv[1], &s, &sl);
  if (sl % 32 != 0 || pl != 2 * sl) {
    napi_throw_range_error(env, nullptr, "points must be 64*n bytes and scalars 32*n bytes");
    return nullptr;
  }
  Job* j = new Job();
  j->j.points = static_cast<const uint8_t*>(p); j->j.scalars = static_cast<const uint8_t*>(s); j->j.n = sl / 32;
  napi_create_reference(env, argv[0], 1, &j->points_ref);
  napi_create_reference(env, argv[1], 1, &j->scalars_ref);
  napi_value promise, name;
  napi_create_promise(env, &j->deferred, &promise);
  napi_create_string_utf8(env, "te_msm_run", NAPI_AUTO_LENGTH, &name);
  napi_create_async_work(env, nullptr, name, Execute, Complete, j, &j->work);
  // other calls pending, the context exists and no pool thread is inside the engine: the waiting calls become tickets right here,
  // oldest first (microseconds), and their pool threads will only wait and collect (te_promise::protocol::enter)
  g_proto.enter(&j->j);
  napi_queue_async_work(env, j->work);
  return promise;
}

// resetContext(): drops the cached engine context (compute_msm's force_recompile) once no promise is pending
napi_value ResetContext(napi_env env, napi_callback_info) {
  g_proto.reset();
  napi_value u; napi_get_undefined(env, &u); return u;
}

// setDevices([0, 1, ...]): the GPUs later calls are sharded over (overrides TE_MSM_DEVICES; [] = back to the environment)
napi_value SetDevices(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  bool is_arr = false;
  if (argc >= 1) napi_is_array(env, argv[0], &is_arr);
  if (!is_arr) { napi_throw_type_error(env, nullptr, "setDevices(ids: number[])"); return nullptr; }
  uint32_t len = 0; napi_get_array_length(env, argv[0], &len);
  if (len > 64) { napi_throw_range_error(env, nullptr, "setDevices: at most 64 devices"); return nullptr; }
  std::vector<int> ids;
  for (uint32_t i = 0; i < len; i++) {
    napi_value v; int32_t id = 0;
    napi_get_element(env, argv[0], i, &v);
    if (napi_get_value_int32(env, v, &id) != napi_ok || id < 0) { napi_throw_type_error(env, nullptr, "setDevices: device ids are non-negative integers"); return nullptr; }
    ids.push_back(id);
  }
  g_proto.set_devices(ids);
  napi_value u; napi_get_undefined(env, &u); return u;
}

// setBases(points: Buffer | null): binds this point buffer (te_msm_bind_points: uploaded and converted once, on every device);
// compute_msm(thatBuffer, scalars) -- the same Buffer object: address and length are compared, its contents must not change --
// then uploads and decomposes the scalars only.  Every other buffer takes the ordinary path; setBases(null) unbinds.
// compute_msm's signature (submission.ts:73-78) is untouched; the reference's harness passes one point buffer to six calls per
// size (submission/miscellaneous/full_benchmarks.ts:63-68,100-105).  Blocks until no promise is pending (like resetContext).
napi_value SetBases(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  napi_valuetype vt = napi_undefined;
  if (argc >= 1) napi_typeof(env, argv[0], &vt);
  bool is_buf = false;
  if (argc >= 1 && vt == napi_object) napi_is_buffer(env, argv[0], &is_buf);
  if (!(is_buf || argc == 0 || vt == napi_null || vt == napi_undefined)) { napi_throw_type_error(env, nullptr, "setBases(points: Buffer | null)"); return nullptr; }
  std::string err;
  if (!is_buf) {
    (void)g_proto.set_bases(nullptr, 0, err);
    if (g_bases_ref) { napi_delete_reference(env, g_bases_ref); g_bases_ref = nullptr; }
  } else {
    void* p = nullptr; size_t pl = 0;
    napi_get_buffer_info(env, argv[0], &p, &pl);
    if (pl % 64 != 0) { napi_throw_range_error(env, nullptr, "setBases: points must be 64*n bytes"); return nullptr; }
    napi_ref ref = nullptr;
    napi_create_reference(env, argv[0], 1, &ref);
    const int rc = g_proto.set_bases(static_cast<const uint8_t*>(p), pl / 64, err);
    if (g_bases_ref) napi_delete_reference(env, g_bases_ref);         // the previous buffer is unbound either way
    g_bases_ref = nullptr;
    if (rc) {
      napi_delete_reference(env, ref);
      const std::string m = "te_msm error " + std::to_string(rc) + ": " + err;
      napi_throw_error(env, nullptr, m.c_str());
      return nullptr;
    }
    g_bases_ref = ref;
  }
  napi_value u; napi_get_undefined(env, &u); return u;
}

// getStats(): how the promises so far were mapped onto the engine -- tickets submitted from the JavaScript thread / from pool
// threads, lone calls, jobs over bound bases, and the largest number of tickets seen in flight at a submit
napi_value GetStats(napi_env env, napi_callback_info) {
  const te_promise::stats_t st = g_proto.stats();
  napi_value o; napi_create_object(env, &o);
  const struct { const char* k; double v; } f[] = {{"submittedInEnter", (double)st.submitted_in_enter}, {"submittedInExecute", (double)st.submitted_in_execute},
                                                    {"loneRuns", (double)st.lone_runs}, {"boundJobs", (double)st.bound_jobs}, {"maxInFlight", (double)st.max_in_flight}};
  for (const auto& e : f) { napi_value v; napi_create_double(env, e.v, &v); napi_set_named_property(env, o, e.k, v); }
  return o;
}

// getDevices(): the device list the next context is (or the current one was) created with
napi_value GetDevices(napi_env env, napi_callback_info) {
  const std::vector<int> ids = g_proto.devices();
  napi_value arr; napi_create_array_with_length(env, ids.size(), &arr);
  for (size_t i = 0; i < ids.size(); i++) { napi_value v; napi_create_int32(env, ids[i], &v); napi_set_element(env, arr, (uint32_t)i, v); }
  return arr;
}

napi_value Init(napi_env env, napi_value exports) {
  const struct { const char* name; napi_callback fn; } fns[] = {
      {"msmNative", MsmNative}, {"resetContext", ResetContext}, {"setDevices", SetDevices}, {"getDevices", GetDevices},
      {"setBases", SetBases}, {"getStats", GetStats}};
  for (const auto& f : fns) {
    napi_value v;
    napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &v);
    napi_set_named_property(env, exports, f.name, v);
  }
  return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
