// addon.cc -- N-API binding of libtemsm.so for the reference's host language (JavaScript / TypeScript).
//
// Exposes  msmNative(points: Buffer, scalars: Buffer): Promise<Buffer /*64 bytes: x || y little-endian*/>
// which compute_msm.js wraps into the reference's entry point
//   compute_msm(bufferPoints, bufferScalars, log_result, force_recompile): Promise<{x: bigint, y: bigint}>
// (submission/submission.ts:73-78).  The MSM runs on a libuv worker thread (napi_create_async_work), so the
// JS event loop is not blocked -- the reference's call is async for the same reason (ui/Benchmark.tsx:32).
// Errors reject the promise, as the reference's `throw`s do (implementation/cuzk/gpu.ts:19-22).
#include <node_api.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include <string>

#include "../../include/te_msm.h"

namespace {

std::mutex g_mu;            // one context, one MSM at a time (the engine context is not thread-safe)
te_ctx* g_ctx = nullptr;

struct Job {
  napi_async_work work = nullptr;
  napi_deferred deferred = nullptr;
  napi_ref points_ref = nullptr, scalars_ref = nullptr;   // keep the JS Buffers alive until completion
  const uint8_t* points = nullptr; const uint8_t* scalars = nullptr;
  uint64_t n = 0;
  int rc = 0; std::string err;
  uint8_t out[64];
};

void Execute(napi_env, void* data) {
  Job* j = static_cast<Job*>(data);
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx) {
    int dev = 0;
    j->rc = te_msm_init(&dev, 1, &g_ctx);
    if (j->rc) { j->err = te_msm_last_error(nullptr); g_ctx = nullptr; return; }
  }
  j->rc = te_msm_run(g_ctx, j->points, j->scalars, j->n, j->out);
  if (j->rc) j->err = te_msm_last_error(g_ctx);
}

void Complete(napi_env env, napi_status, void* data) {
  Job* j = static_cast<Job*>(data);
  if (j->rc == 0) {
    napi_value buf; void* dst = nullptr;
    napi_create_buffer_copy(env, 64, j->out, &dst, &buf);
    napi_resolve_deferred(env, j->deferred, buf);
  } else {
    napi_value msg, errv;
    std::string m = "te_msm error " + std::to_string(j->rc) + ": " + j->err;
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
  j->points = static_cast<const uint8_t*>(p); j->scalars = static_cast<const uint8_t*>(s); j->n = sl / 32;
  napi_create_reference(env, argv[0], 1, &j->points_ref);
  napi_create_reference(env, argv[1], 1, &j->scalars_ref);
  napi_value promise, name;
  napi_create_promise(env, &j->deferred, &promise);
  napi_create_string_utf8(env, "te_msm_run", NAPI_AUTO_LENGTH, &name);
  napi_create_async_work(env, nullptr, name, Execute, Complete, j, &j->work);
  napi_queue_async_work(env, j->work);
  return promise;
}

// resetContext(): drops the cached engine context (compute_msm's force_recompile)
napi_value ResetContext(napi_env env, napi_callback_info) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_ctx) { te_msm_destroy(g_ctx); g_ctx = nullptr; }
  napi_value u; napi_get_undefined(env, &u); return u;
}

napi_value Init(napi_env env, napi_value exports) {
  napi_value f1, f2;
  napi_create_function(env, "msmNative", NAPI_AUTO_LENGTH, MsmNative, nullptr, &f1);
  napi_set_named_property(env, exports, "msmNative", f1);
  napi_create_function(env, "resetContext", NAPI_AUTO_LENGTH, ResetContext, nullptr, &f2);
  napi_set_named_property(env, exports, "resetContext", f2);
  return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
