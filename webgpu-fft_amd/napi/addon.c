/*
 * addon.c — thin N-API binding of include/mi355fft.h for the JavaScript host (webgpu-fft_amd/js).
 *
 * One JS function per C-ABI entry point, no logic of its own: handles are napi externals, C status codes
 * become `throw new Error(mi355fft_last_error())` (the reference throws synchronously from createPlan/exec,
 * SURVEY.md 8b), and the two blocking calls (queue wait, buffer readback) also exist as Promise-returning
 * variants that run on a libuv worker thread so `await device.queue.onSubmittedWorkDone()` and
 * `await downloadComplex(...)` keep the reference's async shape (utils/webgpu.js:29-55).
 *
 * Built as plain C against the system Node headers (N-API v8, /usr/include/node), linked to
 * ../lib/libmi355fft.so with an $ORIGIN rpath: see napi/Makefile.
 */
#define NAPI_VERSION 6
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mi355fft.h"

#define NAPI_CALL(env, call)                                                         \
  do {                                                                               \
    napi_status s_ = (call);                                                         \
    if (s_ != napi_ok) {                                                             \
      napi_throw_error((env), NULL, "mi355fft addon: N-API call failed: " #call);   \
      return NULL;                                                                   \
    }                                                                                \
  } while (0)

static napi_value throw_last(napi_env env) {
  const char* m = mi355fft_last_error();
  napi_throw_error(env, NULL, (m && *m) ? m : "mi355fft: unknown error");
  return NULL;
}
#define MI_CALL(env, call) do { if ((call) != MI355FFT_OK) return throw_last(env); } while (0)

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv) {
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok) return 0;
  for (size_t i = argc; i < want; ++i) napi_get_undefined(env, &argv[i]);
  return 1;
}
static void* get_ext(napi_env env, napi_value v) {
  napi_valuetype t;
  if (napi_typeof(env, v, &t) != napi_ok || t != napi_external) return NULL;
  void* p = NULL;
  napi_get_value_external(env, v, &p);
  return p;
}
static int get_i64(napi_env env, napi_value v, int64_t* out) {
  napi_valuetype t;
  if (napi_typeof(env, v, &t) != napi_ok || t != napi_number) return 0;
  double d;
  if (napi_get_value_double(env, v, &d) != napi_ok) return 0;
  *out = (int64_t)d;
  return 1;
}
static int64_t prop_i64(napi_env env, napi_value obj, const char* name, int64_t dflt) {
  napi_value v;
  bool has = false;
  if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return dflt;
  if (napi_get_named_property(env, obj, name, &v) != napi_ok) return dflt;
  int64_t r;
  return get_i64(env, v, &r) ? r : dflt;
}
static void* prop_ext(napi_env env, napi_value obj, const char* name) {
  napi_value v;
  bool has = false;
  if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return NULL;
  if (napi_get_named_property(env, obj, name, &v) != napi_ok) return NULL;
  return get_ext(env, v);
}
static void prop_i64_array(napi_env env, napi_value obj, const char* name, int64_t* dst, int max) {
  napi_value v;
  bool has = false, is_arr = false;
  if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return;
  if (napi_get_named_property(env, obj, name, &v) != napi_ok) return;
  if (napi_is_array(env, v, &is_arr) != napi_ok || !is_arr) return;
  uint32_t n = 0;
  napi_get_array_length(env, v, &n);
  for (uint32_t i = 0; i < n && (int)i < max; ++i) {
    napi_value e;
    int64_t x;
    if (napi_get_element(env, v, i, &e) == napi_ok && get_i64(env, e, &x)) dst[i] = x;
  }
}
static napi_value make_ext(napi_env env, void* p) {
  napi_value v;
  if (napi_create_external(env, p, NULL, NULL, &v) != napi_ok) return NULL;
  return v;
}
static napi_value make_num(napi_env env, double d) {
  napi_value v;
  napi_create_double(env, d, &v);
  return v;
}
static napi_value undefined(napi_env env) {
  napi_value v;
  napi_get_undefined(env, &v);
  return v;
}
/* pointer + byte length of a TypedArray / ArrayBuffer / DataView argument */
static int get_bytes(napi_env env, napi_value v, void** data, size_t* len) {
  bool is = false;
  if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
    napi_typedarray_type t;
    size_t n, off;
    napi_value ab;
    if (napi_get_typedarray_info(env, v, &t, &n, data, &ab, &off) != napi_ok) return 0;
    static const size_t esz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
    *len = n * esz[t];
    return 1;
  }
  if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, data, len) == napi_ok;
  if (napi_is_dataview(env, v, &is) == napi_ok && is) {
    napi_value ab;
    size_t off;
    return napi_get_dataview_info(env, v, len, data, &ab, &off) == napi_ok;
  }
  return 0;
}

/* ---- library / device -------------------------------------------------------------------------- */
static napi_value js_abi_version(napi_env env, napi_callback_info info) { (void)info; return make_num(env, mi355fft_abi_version()); }

static napi_value js_device_count(napi_env env, napi_callback_info info) {
  (void)info;
  int n = 0;
  MI_CALL(env, mi355fft_device_count(&n));
  return make_num(env, n);
}
static napi_value js_device_open(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  int64_t ord = 0;
  get_i64(env, a[0], &ord);
  mi355fft_device* d = NULL;
  MI_CALL(env, mi355fft_device_open((int)ord, &d));
  return make_ext(env, d);
}
static napi_value js_device_close(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  MI_CALL(env, mi355fft_device_close((mi355fft_device*)get_ext(env, a[0])));
  return undefined(env);
}
static napi_value js_device_info(napi_env env, napi_callback_info info) {
  napi_value a[1], o, s;
  if (!get_args(env, info, 1, a)) return NULL;
  uint64_t tot = 0, fr = 0;
  int cus = 0;
  char arch[64] = {0};
  MI_CALL(env, mi355fft_device_info((mi355fft_device*)get_ext(env, a[0]), &tot, &fr, &cus, arch));
  NAPI_CALL(env, napi_create_object(env, &o));
  napi_set_named_property(env, o, "hbmTotal", make_num(env, (double)tot));
  napi_set_named_property(env, o, "hbmFree", make_num(env, (double)fr));
  napi_set_named_property(env, o, "computeUnits", make_num(env, cus));
  napi_create_string_utf8(env, arch, NAPI_AUTO_LENGTH, &s);
  napi_set_named_property(env, o, "arch", s);
  return o;
}

/* ---- buffers ------------------------------------------------------------------------------------ */
static napi_value js_buffer_alloc(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  int64_t bytes = 0;
  if (!get_i64(env, a[1], &bytes) || bytes < 0) { napi_throw_error(env, NULL, "createBuffer: size must be a non-negative number"); return NULL; }
  mi355fft_buffer* b = NULL;
  MI_CALL(env, mi355fft_buffer_alloc((mi355fft_device*)get_ext(env, a[0]), (uint64_t)bytes, &b));
  return make_ext(env, b);
}
static napi_value js_buffer_free(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  MI_CALL(env, mi355fft_buffer_free((mi355fft_buffer*)get_ext(env, a[0])));
  return undefined(env);
}
static napi_value js_buffer_write(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  int64_t off = 0;
  get_i64(env, a[1], &off);
  void* data = NULL;
  size_t len = 0;
  if (!get_bytes(env, a[2], &data, &len)) { napi_throw_error(env, NULL, "writeBuffer: data must be a TypedArray, DataView or ArrayBuffer"); return NULL; }
  MI_CALL(env, mi355fft_buffer_write((mi355fft_buffer*)get_ext(env, a[0]), (uint64_t)off, data, len));
  return undefined(env);
}
static napi_value js_buffer_read(napi_env env, napi_callback_info info) {
  napi_value a[3], ab;
  if (!get_args(env, info, 3, a)) return NULL;
  int64_t off = 0, len = 0;
  get_i64(env, a[1], &off);
  get_i64(env, a[2], &len);
  void* data = NULL;
  NAPI_CALL(env, napi_create_arraybuffer(env, (size_t)len, &data, &ab));
  MI_CALL(env, mi355fft_buffer_read((mi355fft_buffer*)get_ext(env, a[0]), (uint64_t)off, data, (uint64_t)len));
  return ab;
}

/* ---- async: queue wait / buffer readback on a libuv worker, resolved as a Promise ----------------- */
typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  mi355fft_device* dev;
  mi355fft_buffer* buf;
  uint64_t off, len;
  void* host;
  int rc;
  char err[512];
} async_job;

static void async_exec(napi_env env, void* p) {
  (void)env;
  async_job* j = (async_job*)p;
  if (j->buf) j->rc = mi355fft_buffer_read(j->buf, j->off, j->host, j->len);
  else j->rc = mi355fft_queue_wait(j->dev);
  if (j->rc) { strncpy(j->err, mi355fft_last_error(), sizeof j->err - 1); j->err[sizeof j->err - 1] = 0; }
}
static void async_done(napi_env env, napi_status st, void* p) {
  async_job* j = (async_job*)p;
  napi_value v;
  if (st == napi_ok && j->rc == 0) {
    if (j->buf) {
      void* data = NULL;
      napi_create_arraybuffer(env, (size_t)j->len, &data, &v);
      if (data && j->len) memcpy(data, j->host, (size_t)j->len);
    } else napi_get_undefined(env, &v);
    napi_resolve_deferred(env, j->deferred, v);
  } else {
    napi_value msg;
    napi_create_string_utf8(env, j->rc ? j->err : "mi355fft: async work cancelled", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &v);
    napi_reject_deferred(env, j->deferred, v);
  }
  napi_delete_async_work(env, j->work);
  free(j->host);
  free(j);
}
static napi_value start_async(napi_env env, async_job* j, const char* name) {
  napi_value promise, rn;
  NAPI_CALL(env, napi_create_promise(env, &j->deferred, &promise));
  napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &rn);
  NAPI_CALL(env, napi_create_async_work(env, NULL, rn, async_exec, async_done, j, &j->work));
  NAPI_CALL(env, napi_queue_async_work(env, j->work));
  return promise;
}
static napi_value js_queue_wait_async(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  async_job* j = (async_job*)calloc(1, sizeof *j);
  j->dev = (mi355fft_device*)get_ext(env, a[0]);
  return start_async(env, j, "mi355fft.queueWait");
}
static napi_value js_buffer_read_async(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  int64_t off = 0, len = 0;
  get_i64(env, a[1], &off);
  get_i64(env, a[2], &len);
  async_job* j = (async_job*)calloc(1, sizeof *j);
  j->buf = (mi355fft_buffer*)get_ext(env, a[0]);
  j->off = (uint64_t)off;
  j->len = (uint64_t)len;
  j->host = malloc((size_t)(len > 0 ? len : 1));
  return start_async(env, j, "mi355fft.bufferRead");
}

/* ---- plans -------------------------------------------------------------------------------------- */
static void fill_side(napi_env env, napi_value parent, const char* name, mi355fft_side_layout* s) {
  napi_value v;
  bool has = false;
  napi_valuetype t;
  if (napi_has_named_property(env, parent, name, &has) != napi_ok || !has) return;
  if (napi_get_named_property(env, parent, name, &v) != napi_ok) return;
  if (napi_typeof(env, v, &t) != napi_ok || t != napi_object) return;
  s->strided = 1;
  prop_i64_array(env, v, "strides", s->strides, MI355FFT_MAX_RANK);
  s->offset_elements = prop_i64(env, v, "offset", 0);
  s->batch_stride_elements = prop_i64(env, v, "batchStride", 0);
}
static napi_value get_obj_prop(napi_env env, napi_value parent, const char* name) {
  napi_value v;
  bool has = false;
  napi_valuetype t;
  if (napi_has_named_property(env, parent, name, &has) != napi_ok || !has) return NULL;
  if (napi_get_named_property(env, parent, name, &v) != napi_ok) return NULL;
  if (napi_typeof(env, v, &t) != napi_ok || t != napi_object) return NULL;
  return v;
}
static void fill_view(napi_env env, napi_value parent, const char* name, mi355fft_io_view* dst) {
  napi_value v = get_obj_prop(env, parent, name);
  if (!v) return;
  dst->enabled = 1;
  dst->clear_outside = (int32_t)prop_i64(env, v, "clearOutside", 0);
  prop_i64_array(env, v, "shape", dst->shape, MI355FFT_MAX_RANK);
  prop_i64_array(env, v, "offset", dst->offset, MI355FFT_MAX_RANK);
}
static void fill_range(napi_env env, napi_value parent, const char* name, mi355fft_zero_range* dst) {
  napi_value v = get_obj_prop(env, parent, name);
  if (!v) return;
  dst->enabled = 1;
  prop_i64_array(env, v, "start", dst->start, MI355FFT_MAX_RANK);
  prop_i64_array(env, v, "end", dst->end, MI355FFT_MAX_RANK);
}
static napi_value js_plan_create(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  napi_value o = a[1];
  mi355fft_plan_desc d;
  memset(&d, 0, sizeof d);
  d.struct_size = sizeof d;
  d.type = (int32_t)prop_i64(env, o, "type", -1);
  d.direction = (int32_t)prop_i64(env, o, "direction", 0);
  d.normalize = (int32_t)prop_i64(env, o, "normalize", 0);
  d.in_place = (int32_t)prop_i64(env, o, "inPlace", 0);
  d.batch = prop_i64(env, o, "batch", 1);
  napi_value shape;
  bool is_arr = false;
  if (napi_get_named_property(env, o, "shape", &shape) == napi_ok && napi_is_array(env, shape, &is_arr) == napi_ok && is_arr) {
    uint32_t n = 0;
    napi_get_array_length(env, shape, &n);
    d.rank = (int32_t)n;
    prop_i64_array(env, o, "shape", d.shape, MI355FFT_MAX_RANK);
  }
  fill_side(env, o, "input", &d.input);
  fill_side(env, o, "output", &d.output);
  d.conv_mode = (int32_t)prop_i64(env, o, "convMode", 0);
  d.conv_boundary = (int32_t)prop_i64(env, o, "convBoundary", 0);
  d.conv_kernel_count = (int32_t)prop_i64(env, o, "convKernelCount", 1);
  d.conv_output_layout = (int32_t)prop_i64(env, o, "convOutputLayout", 0);
  prop_i64_array(env, o, "convKernelShape", d.conv_kernel_shape, MI355FFT_MAX_RANK);
  d.conv_output_kernel_stride_elements = prop_i64(env, o, "convOutputKernelStrideElements", 0);
  fill_view(env, o, "ioInput", &d.io_input);
  fill_view(env, o, "ioOutput", &d.io_output);
  fill_range(env, o, "zeroRead", &d.zero_read);
  fill_range(env, o, "zeroWrite", &d.zero_write);
  d.axes_mask = (uint32_t)prop_i64(env, o, "axesMask", 0);
  mi355fft_plan* p = NULL;
  MI_CALL(env, mi355fft_plan_create((mi355fft_device*)get_ext(env, a[0]), &d, &p));
  return make_ext(env, p);
}
static napi_value js_plan_workspace_bytes(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  uint64_t b = 0;
  MI_CALL(env, mi355fft_plan_workspace_bytes((mi355fft_plan*)get_ext(env, a[0]), &b));
  return make_num(env, (double)b);
}
static napi_value js_plan_describe(napi_env env, napi_callback_info info) {
  napi_value a[1], o, s;
  if (!get_args(env, info, 1, a)) return NULL;
  char text[1024] = {0};
  int launches = 0;
  MI_CALL(env, mi355fft_plan_describe((mi355fft_plan*)get_ext(env, a[0]), text, sizeof text, &launches));
  NAPI_CALL(env, napi_create_object(env, &o));
  napi_create_string_utf8(env, text, NAPI_AUTO_LENGTH, &s);
  napi_set_named_property(env, o, "route", s);
  napi_set_named_property(env, o, "launches", make_num(env, launches));
  return o;
}
static napi_value js_plan_exec(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  mi355fft_exec_args x;
  memset(&x, 0, sizeof x);
  x.struct_size = sizeof x;
  napi_valuetype t;
  if (napi_typeof(env, a[2], &t) == napi_ok && t == napi_object) {
    x.input = (mi355fft_buffer*)prop_ext(env, a[2], "input");
    x.output = (mi355fft_buffer*)prop_ext(env, a[2], "output");
    x.temp = (mi355fft_buffer*)prop_ext(env, a[2], "temp");
    x.kernel = (mi355fft_buffer*)prop_ext(env, a[2], "kernel");
    x.input_offset_bytes = (uint64_t)prop_i64(env, a[2], "inputOffsetBytes", 0);
    x.output_offset_bytes = (uint64_t)prop_i64(env, a[2], "outputOffsetBytes", 0);
    x.kernel_offset_bytes = (uint64_t)prop_i64(env, a[2], "kernelOffsetBytes", 0);
  }
  MI_CALL(env, mi355fft_plan_exec((mi355fft_plan*)get_ext(env, a[0]), (mi355fft_encoder*)get_ext(env, a[1]), &x));
  return undefined(env);
}
static napi_value js_plan_destroy(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  MI_CALL(env, mi355fft_plan_destroy((mi355fft_plan*)get_ext(env, a[0])));
  return undefined(env);
}
static napi_value js_plan_release(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  MI_CALL(env, mi355fft_plan_release((mi355fft_plan*)get_ext(env, a[0])));
  return undefined(env);
}

/* ---- encoder / queue ------------------------------------------------------------------------------ */
static napi_value js_encoder_begin(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  mi355fft_encoder* e = NULL;
  MI_CALL(env, mi355fft_encoder_begin((mi355fft_device*)get_ext(env, a[0]), &e));
  return make_ext(env, e);
}
static napi_value js_encoder_copy_buffer(napi_env env, napi_callback_info info) {
  napi_value a[6];
  if (!get_args(env, info, 6, a)) return NULL;
  int64_t so = 0, dof = 0, n = 0;
  get_i64(env, a[2], &so);
  get_i64(env, a[4], &dof);
  get_i64(env, a[5], &n);
  MI_CALL(env, mi355fft_encoder_copy_buffer((mi355fft_encoder*)get_ext(env, a[0]), (mi355fft_buffer*)get_ext(env, a[1]), (uint64_t)so,
                                            (mi355fft_buffer*)get_ext(env, a[3]), (uint64_t)dof, (uint64_t)n));
  return undefined(env);
}
static napi_value js_encoder_finish(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  int64_t g = 0;
  get_i64(env, a[1], &g);
  mi355fft_commands* c = NULL;
  MI_CALL(env, mi355fft_encoder_finish((mi355fft_encoder*)get_ext(env, a[0]), (int)g, &c));
  return make_ext(env, c);
}
static napi_value js_encoder_discard(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  MI_CALL(env, mi355fft_encoder_discard((mi355fft_encoder*)get_ext(env, a[0])));
  return undefined(env);
}
static napi_value js_queue_submit(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  MI_CALL(env, mi355fft_queue_submit((mi355fft_device*)get_ext(env, a[0]), (mi355fft_commands*)get_ext(env, a[1])));
  return undefined(env);
}
static napi_value js_commands_release(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  MI_CALL(env, mi355fft_commands_release((mi355fft_commands*)get_ext(env, a[0])));
  return undefined(env);
}
static napi_value js_queue_wait(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  MI_CALL(env, mi355fft_queue_wait((mi355fft_device*)get_ext(env, a[0])));
  return undefined(env);
}

/* ---- test / bench support -------------------------------------------------------------------------- */
static napi_value js_fill_random(napi_env env, napi_callback_info info) {
  napi_value a[7];
  if (!get_args(env, info, 7, a)) return NULL;
  int64_t off = 0, rf = 0, rows = 0, seed = 0, first = 0;
  get_i64(env, a[2], &off); get_i64(env, a[3], &rf); get_i64(env, a[4], &rows); get_i64(env, a[5], &seed); get_i64(env, a[6], &first);
  MI_CALL(env, mi355fft_fill_random((mi355fft_device*)get_ext(env, a[0]), (mi355fft_buffer*)get_ext(env, a[1]), (uint64_t)off, (uint64_t)rf,
                                    (uint64_t)rows, (uint32_t)seed, (uint64_t)first));
  return undefined(env);
}
static napi_value js_sumsq(napi_env env, napi_callback_info info) {
  napi_value a[4];
  if (!get_args(env, info, 4, a)) return NULL;
  int64_t off = 0, n = 0;
  get_i64(env, a[2], &off); get_i64(env, a[3], &n);
  double out = 0;
  MI_CALL(env, mi355fft_sumsq((mi355fft_device*)get_ext(env, a[0]), (mi355fft_buffer*)get_ext(env, a[1]), (uint64_t)off, (uint64_t)n, &out));
  return make_num(env, out);
}

static napi_value init(napi_env env, napi_value exports) {
  static const struct { const char* name; napi_callback fn; } fns[] = {
    {"abiVersion", js_abi_version}, {"deviceCount", js_device_count}, {"deviceOpen", js_device_open}, {"deviceClose", js_device_close},
    {"deviceInfo", js_device_info}, {"bufferAlloc", js_buffer_alloc}, {"bufferFree", js_buffer_free}, {"bufferWrite", js_buffer_write},
    {"bufferRead", js_buffer_read}, {"bufferReadAsync", js_buffer_read_async}, {"planCreate", js_plan_create},
    {"planWorkspaceBytes", js_plan_workspace_bytes}, {"planDescribe", js_plan_describe}, {"planExec", js_plan_exec},
    {"planDestroy", js_plan_destroy}, {"planRelease", js_plan_release}, {"encoderBegin", js_encoder_begin},
    {"encoderCopyBuffer", js_encoder_copy_buffer}, {"encoderFinish", js_encoder_finish}, {"encoderDiscard", js_encoder_discard},
    {"queueSubmit", js_queue_submit}, {"commandsRelease", js_commands_release}, {"queueWait", js_queue_wait},
    {"queueWaitAsync", js_queue_wait_async}, {"fillRandom", js_fill_random}, {"sumsq", js_sumsq},
  };
  for (size_t i = 0; i < sizeof fns / sizeof fns[0]; ++i) {
    napi_value f;
    if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
    if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
  }
  return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
