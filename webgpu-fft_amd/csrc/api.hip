// api.hip — the C ABI of libmi355fft.so (include/mi355fft.h): devices, buffers, plans, the recording
// encoder and the queue.  Each entry point cites the reference interface it replaces in the header.
//
// Record vs execute: plan.exec only appends resolved launches to the encoder; finish()
// freezes them into a command list — optionally a hipGraph captured from replaying the list on a capture
// stream — and queue.submit enqueues it on the device stream.  There is no CPU fallback anywhere: with no
// HIP device every call that needs one fails with MI355FFT_ERR_HIP and says so.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mi355fft.h"
#include "hip_launcher.hpp"
#include "plan.hpp"

using namespace mi355;

#define MI_API extern "C" __attribute__((visibility("default")))

namespace {
thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(expr)                                                                                       \
  do {                                                                                                      \
    const hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) return fail(MI355FFT_ERR_HIP, "HIP error %d (%s) in %s", (int)e_, hipGetErrorString(e_), #expr); \
  } while (0)
}  // namespace

struct mi355fft_device {
  int ordinal = 0;
  hipStream_t stream = nullptr;
  hipStream_t capture_stream = nullptr;
  int compute_units = 256;
  std::string arch;
  unsigned* sticky = nullptr;     // device word raised by kernels whose bounded waits timed out (kern_xcd.hpp)
  bool sticky_armed = false;      // a kernel that can raise it has been submitted since the last check
  bool xcd_disabled = false;      // a bounded wait HAS timed out on this device (its workgroups were not co-resident: another
                                  // stream / process / CU mask shares the GPU): new plans, and existing ones at their next exec,
                                  // take the routes without cross-workgroup synchronisation
};
// Recorded command lists hold raw device pointers into plans (tables, workspace arena) and caller buffers.  Each of those
// owners carries an "alive" token; a command list keeps a copy of the token of everything it references and submit refuses a
// list whose plan or buffer has been destroyed (the reference raises a validation error there; writing into freed or
// re-allocated HBM is not an option).
typedef std::shared_ptr<bool> AliveToken;
struct mi355fft_buffer {
  mi355fft_device* dev = nullptr;
  void* ptr = nullptr;
  uint64_t bytes = 0;
  bool owned = false;
  AliveToken alive = std::make_shared<bool>(true);
};
struct RecordedOp { Step step; void* ptr[5]; };
struct mi355fft_encoder {
  mi355fft_device* dev = nullptr;
  std::vector<RecordedOp> ops;
  std::vector<AliveToken> deps;
};
struct mi355fft_commands {
  mi355fft_device* dev = nullptr;
  std::vector<RecordedOp> ops;
  std::vector<AliveToken> deps;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};
struct mi355fft_plan {
  mi355fft_device* dev = nullptr;
  AliveToken alive = std::make_shared<bool>(true);
  PlanIR ir;
  void* table = nullptr;
  void* arena = nullptr;
  uint64_t arena_bytes = 0;
  bool destroyed = false;
  bool uses_xcd_sync = false;     // has a step whose workgroups wait for each other (needs co-residency)
};

namespace {

bool lines_dispatch(int family, int id, const LineArgs& a, unsigned grid, HipLauncher& l) {
  switch (family) {
    case FAM_ROW_SMALL: return launch_lines_family<FAM_ROW_SMALL>(id, a, grid, l);
    case FAM_ROW_1K: return launch_lines_family<FAM_ROW_1K>(id, a, grid, l);
    case FAM_ROW_BIG: return launch_lines_family<FAM_ROW_BIG>(id, a, grid, l);
    case FAM_PASS_A: return launch_lines_family<FAM_PASS_A>(id, a, grid, l);
    case FAM_PASS_B: return launch_lines_family<FAM_PASS_B>(id, a, grid, l);
  }
  return false;
}

int replay(const std::vector<RecordedOp>& ops, HipLauncher& l) {
  for (const RecordedOp& op : ops) {
    const bool ok = dispatch_step(op.step, op.ptr, l, [&](int fam, int id, const LineArgs& a, unsigned grid) { return lines_dispatch(fam, id, a, grid, l); },
                                  [&](int id, const XcdFusedArgs& a, unsigned grid) { return launch_xcd_fused(id, a, grid, l); });
    if (!ok) return fail(MI355FFT_ERR_UNSUPPORTED, "no kernel instance for step kind %d variant %d", (int)op.step.kind, op.step.variant);
    if (l.status != hipSuccess) return fail(MI355FFT_ERR_HIP, "HIP error %d (%s) launching step kind %d", (int)l.status, hipGetErrorString(l.status), (int)op.step.kind);
  }
  return MI355FFT_OK;
}

// Synchronises the device stream and reports a bounded wait that gave up inside a kernel (kern_xcd.hpp / kern_xcd_res.hpp raise
// the sticky word).  Every host-visible hand-over of results goes through here — queue_wait, buffer reads, reductions — so a
// timed-out submit can never be read back as if it had succeeded.
int sync_and_check(mi355fft_device* dev) {
  HIP_TRY(hipSetDevice(dev->ordinal));
  HIP_TRY(hipStreamSynchronize(dev->stream));
  if (dev->sticky_armed) {
    dev->sticky_armed = false;
    unsigned word = 0;
    HIP_TRY(hipMemcpy(&word, dev->sticky, sizeof word, hipMemcpyDeviceToHost));
    if (word) {
      (void)hipMemset(dev->sticky, 0, sizeof word);
      dev->xcd_disabled = true;
      return fail(MI355FFT_ERR_HIP, "XCD-fused FFT kernel gave up waiting (%s%s%s): its workgroups were not all co-resident; results of that submit are "
                  "invalid.  Plans on this device now use the routes without cross-workgroup synchronisation: record and submit again.", (word & 1u) ? "registration " : "", (word & 2u) ? "barrier " : "",
                  (word & 16u) ? "registration-count-overflow" : (word & 4u) ? "group-size" : "");
    }
  }
  return MI355FFT_OK;
}

void destroy_commands(mi355fft_commands* c) {
  if (c->exec) (void)hipGraphExecDestroy(c->exec);
  if (c->graph) (void)hipGraphDestroy(c->graph);
  delete c;
}

}  // namespace

MI_API int mi355fft_abi_version(void) { return MI355FFT_ABI_VERSION; }
MI_API const char* mi355fft_last_error(void) { return g_err.c_str(); }

// ---- device ----------------------------------------------------------------------------------------
MI_API int mi355fft_device_count(int* count) {
  if (!count) return fail(MI355FFT_ERR_INVALID, "count is NULL");
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(MI355FFT_ERR_HIP, "hipGetDeviceCount failed: %s (is this an MI355X box with /dev/kfd?)", hipGetErrorString(e)); }
  *count = n;
  return MI355FFT_OK;
}

MI_API int mi355fft_device_open(int ordinal, mi355fft_device** out) {
  if (!out) return fail(MI355FFT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int n = 0;
  int rc = mi355fft_device_count(&n);
  if (rc) return rc;
  if (n <= 0) return fail(MI355FFT_ERR_HIP, "Expected a HIP device: none visible (libmi355fft has no CPU path)");
  if (ordinal < 0 || ordinal >= n) return fail(MI355FFT_ERR_INVALID, "device ordinal %d out of range [0,%d)", ordinal, n);
  HIP_TRY(hipSetDevice(ordinal));
  std::unique_ptr<mi355fft_device> d(new mi355fft_device());
  d->ordinal = ordinal;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, ordinal));
  d->compute_units = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  d->arch = prop.gcnArchName;
  HIP_TRY(hipStreamCreate(&d->stream));
  HIP_TRY(hipStreamCreateWithFlags(&d->capture_stream, hipStreamNonBlocking));
  HIP_TRY(hipMalloc((void**)&d->sticky, 64));
  HIP_TRY(hipMemset(d->sticky, 0, 64));
  *out = d.release();
  return MI355FFT_OK;
}

MI_API int mi355fft_device_close(mi355fft_device* dev) {
  if (!dev) return MI355FFT_OK;
  (void)hipSetDevice(dev->ordinal);
  (void)hipStreamSynchronize(dev->stream);
  (void)hipStreamDestroy(dev->stream);
  (void)hipStreamDestroy(dev->capture_stream);
  if (dev->sticky) (void)hipFree(dev->sticky);
  delete dev;
  return MI355FFT_OK;
}

MI_API int mi355fft_device_info(mi355fft_device* dev, uint64_t* hbm_total, uint64_t* hbm_free, int* compute_units, char arch[64]) {
  if (!dev) return fail(MI355FFT_ERR_INVALID, "Expected a device");
  HIP_TRY(hipSetDevice(dev->ordinal));
  size_t fr = 0, tot = 0;
  HIP_TRY(hipMemGetInfo(&fr, &tot));
  if (hbm_total) *hbm_total = tot;
  if (hbm_free) *hbm_free = fr;
  if (compute_units) *compute_units = dev->compute_units;
  if (arch) { std::snprintf(arch, 64, "%s", dev->arch.c_str()); }
  return MI355FFT_OK;
}

MI_API void* mi355fft_device_stream(mi355fft_device* dev) { return dev ? (void*)dev->stream : nullptr; }

// ---- buffers ---------------------------------------------------------------------------------------
MI_API int mi355fft_buffer_alloc(mi355fft_device* dev, uint64_t bytes, mi355fft_buffer** out) {
  if (!dev || !out) return fail(MI355FFT_ERR_INVALID, "Expected a device and an out pointer");
  *out = nullptr;
  if (bytes == 0) return fail(MI355FFT_ERR_INVALID, "createBuffer: size must be > 0");
  HIP_TRY(hipSetDevice(dev->ordinal));
  void* p = nullptr;
  const hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail(MI355FFT_ERR_NOMEM, "hipMalloc(%llu bytes) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
  mi355fft_buffer* b = new mi355fft_buffer();
  b->dev = dev; b->ptr = p; b->bytes = bytes; b->owned = true;
  *out = b;
  return MI355FFT_OK;
}

MI_API int mi355fft_buffer_wrap(mi355fft_device* dev, void* device_ptr, uint64_t bytes, mi355fft_buffer** out) {
  if (!dev || !out || !device_ptr || bytes == 0) return fail(MI355FFT_ERR_INVALID, "buffer_wrap: device, pointer, size and out are required");
  mi355fft_buffer* b = new mi355fft_buffer();
  b->dev = dev; b->ptr = device_ptr; b->bytes = bytes; b->owned = false;
  *out = b;
  return MI355FFT_OK;
}

MI_API int mi355fft_buffer_free(mi355fft_buffer* buf) {
  if (!buf) return MI355FFT_OK;
  *buf->alive = false;                              // command lists that reference it can no longer be submitted
  if (buf->owned && buf->ptr) {
    (void)hipSetDevice(buf->dev->ordinal);
    (void)hipStreamSynchronize(buf->dev->stream);   // submitted work may still use it (a sticky error stays armed for the next check)
    (void)hipFree(buf->ptr);
  }
  delete buf;
  return MI355FFT_OK;
}

MI_API uint64_t mi355fft_buffer_size(const mi355fft_buffer* buf) { return buf ? buf->bytes : 0; }
MI_API void* mi355fft_buffer_device_ptr(const mi355fft_buffer* buf) { return buf ? buf->ptr : nullptr; }

MI_API int mi355fft_buffer_write(mi355fft_buffer* buf, uint64_t offset_bytes, const void* src, uint64_t bytes) {
  if (!buf || (!src && bytes)) return fail(MI355FFT_ERR_INVALID, "writeBuffer: buffer and data are required");
  if (offset_bytes + bytes > buf->bytes) return fail(MI355FFT_ERR_INVALID, "writeBuffer: range [%llu, %llu) exceeds buffer size %llu",
                                                     (unsigned long long)offset_bytes, (unsigned long long)(offset_bytes + bytes), (unsigned long long)buf->bytes);
  if (!bytes) return MI355FFT_OK;
  HIP_TRY(hipSetDevice(buf->dev->ordinal));
  HIP_TRY(hipMemcpyAsync((char*)buf->ptr + offset_bytes, src, bytes, hipMemcpyHostToDevice, buf->dev->stream));
  HIP_TRY(hipStreamSynchronize(buf->dev->stream));
  return MI355FFT_OK;
}

MI_API int mi355fft_buffer_read(mi355fft_buffer* buf, uint64_t offset_bytes, void* dst, uint64_t bytes) {
  if (!buf || (!dst && bytes)) return fail(MI355FFT_ERR_INVALID, "readback: buffer and destination are required");
  if (offset_bytes + bytes > buf->bytes) return fail(MI355FFT_ERR_INVALID, "readback: range [%llu, %llu) exceeds buffer size %llu",
                                                     (unsigned long long)offset_bytes, (unsigned long long)(offset_bytes + bytes), (unsigned long long)buf->bytes);
  if (!bytes) return MI355FFT_OK;
  { const int rc = sync_and_check(buf->dev); if (rc) return rc; }
  HIP_TRY(hipMemcpy(dst, (const char*)buf->ptr + offset_bytes, bytes, hipMemcpyDeviceToHost));
  return MI355FFT_OK;
}

// ---- plans -----------------------------------------------------------------------------------------
namespace {
bool step_needs_coresidency(const Step& s) { return s.kind == ST_XCD_RES || (s.kind == ST_XCD_FUSED && s.i[12] == 0); }

// Plans `desc` for `dev` into p->ir and uploads the tables.  Steps whose workgroups synchronise with each other are only kept
// when the runtime's occupancy answer covers the grid the planner sized for them (all of it resident at once); otherwise, and
// on a device where such a kernel has already timed out, the plan is built again without those routes.
int build_and_upload(mi355fft_device* dev, const mi355fft_plan_desc& desc, mi355fft_plan* p) {
  PlannerOptions opt = planner_options_from_env();
  opt.compute_units = dev->compute_units;
  HIP_TRY(hipSetDevice(dev->ordinal));
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (dev->xcd_disabled || attempt == 1) opt.xcd_shared = 0;
    p->ir = PlanIR();
    std::string err;
    const int rc = build_plan(desc, opt, p->ir, err);
    if (rc) return fail(rc, "%s", err.c_str());
    // raise dynamic-LDS limits now (hipFuncSetAttribute is not legal inside a later stream capture) and check co-residency
    HipLauncher l;
    l.sticky = dev->sticky;
    l.prepare_only = true;
    std::vector<RecordedOp> probe;
    for (const Step& s : p->ir.steps) if (s.kind == ST_LINES || s.kind == ST_XCD_FUSED || s.kind == ST_XCD_RES) { RecordedOp op; op.step = s; std::memset(op.ptr, 0, sizeof op.ptr); probe.push_back(op); }
    const int prc = replay(probe, l);
    if (prc) return prc;
    bool fits = true;
    p->uses_xcd_sync = false;
    for (const Step& s : p->ir.steps) {
      if (!step_needs_coresidency(s)) continue;
      p->uses_xcd_sync = true;
      HipLauncher q;
      q.sticky = dev->sticky;
      q.occupancy_query = true;
      std::vector<RecordedOp> one(1);
      one[0].step = s;
      std::memset(one[0].ptr, 0, sizeof one[0].ptr);
      const int qrc = replay(one, q);
      if (qrc) return qrc;
      const long long need = ((long long)s.grid + dev->compute_units - 1) / dev->compute_units;
      if (q.min_blocks_per_cu < need) fits = false;
    }
    if (fits || attempt == 1) break;
  }
  const size_t tbytes = p->ir.table.size() * sizeof(float2h);
  hipError_t e = hipMalloc(&p->table, tbytes);
  if (e != hipSuccess) return fail(MI355FFT_ERR_NOMEM, "hipMalloc(twiddle tables, %zu bytes) failed: %s", tbytes, hipGetErrorString(e));
  e = hipMemcpy(p->table, p->ir.table.data(), tbytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(p->table); p->table = nullptr; return fail(MI355FFT_ERR_HIP, "uploading twiddle tables failed: %s", hipGetErrorString(e)); }
  return MI355FFT_OK;
}
}  // namespace

MI_API int mi355fft_plan_create(mi355fft_device* dev, const mi355fft_plan_desc* desc, mi355fft_plan** out) {
  if (!dev) return fail(MI355FFT_ERR_INVALID, "Expected a device");
  if (!desc || !out) return fail(MI355FFT_ERR_INVALID, "createPlan: options and out pointer are required");
  *out = nullptr;
  std::unique_ptr<mi355fft_plan> p(new mi355fft_plan());
  p->dev = dev;
  const int rc = build_and_upload(dev, *desc, p.get());
  if (rc) return rc;
  *out = p.release();
  return MI355FFT_OK;
}

MI_API int mi355fft_plan_workspace_bytes(const mi355fft_plan* plan, uint64_t* bytes) {
  if (!plan || !bytes) return fail(MI355FFT_ERR_INVALID, "plan and out pointer are required");
  *bytes = plan->ir.work_bytes;
  return MI355FFT_OK;
}

MI_API int mi355fft_plan_describe(const mi355fft_plan* plan, char* text, size_t text_bytes, int* launches_per_exec) {
  if (!plan) return fail(MI355FFT_ERR_INVALID, "plan is required");
  if (text && text_bytes) std::snprintf(text, text_bytes, "%s", plan->ir.route.c_str());
  if (launches_per_exec) *launches_per_exec = (int)plan->ir.steps.size();
  return MI355FFT_OK;
}

MI_API int mi355fft_plan_exec(mi355fft_plan* plan, mi355fft_encoder* enc, const mi355fft_exec_args* args) {
  if (!plan) return fail(MI355FFT_ERR_INVALID, "plan is required");
  if (plan->destroyed) return fail(MI355FFT_ERR_DESTROYED, "plan destroyed");
  if (!enc) return fail(MI355FFT_ERR_INVALID, "exec requires a command encoder");
  if (!args || args->struct_size != sizeof(mi355fft_exec_args)) return fail(MI355FFT_ERR_INVALID, "exec options missing or ABI size mismatch");
  if (enc->dev != plan->dev) return fail(MI355FFT_ERR_INVALID, "encoder and plan belong to different devices");
  if (plan->dev->xcd_disabled && plan->uses_xcd_sync) {
    // a kernel with cross-workgroup waits has timed out on this device: rebuild this plan on the routes without them.  Command
    // lists recorded from the old steps reference the old tables / arena and can no longer be submitted.
    (void)hipSetDevice(plan->dev->ordinal);
    (void)hipStreamSynchronize(plan->dev->stream);
    *plan->alive = false;
    plan->alive = std::make_shared<bool>(true);
    if (plan->table) (void)hipFree(plan->table);
    if (plan->arena) (void)hipFree(plan->arena);
    plan->table = plan->arena = nullptr;
    plan->arena_bytes = 0;
    const mi355fft_plan_desc desc_copy = plan->ir.desc;
    const int rrc = build_and_upload(plan->dev, desc_copy, plan);
    if (rrc) return rrc;
  }
  const mi355fft_plan_desc& d = plan->ir.desc;
  if (!args->input) return fail(MI355FFT_ERR_INVALID, "exec requires input");
  if (!d.in_place && !args->output) return fail(MI355FFT_ERR_INVALID, "exec requires output when inPlace=false");
  if (d.in_place && args->output && (args->output != args->input || args->output_offset_bytes != args->input_offset_bytes))
    return fail(MI355FFT_ERR_INVALID, "inPlace=true requires output omitted or equal to input");
  if (args->input_offset_bytes % 8 || args->output_offset_bytes % 8 || args->kernel_offset_bytes % 8)
    return fail(MI355FFT_ERR_INVALID, "inputOffsetBytes/outputOffsetBytes must be multiples of 8");
  if (d.type == MI355FFT_FFTCONV && !args->kernel) return fail(MI355FFT_ERR_INVALID, "fftconv exec requires kernel");
  mi355fft_buffer* out = d.in_place ? args->input : args->output;
  const uint64_t out_off = d.in_place ? args->input_offset_bytes : args->output_offset_bytes;
  if (!d.in_place && d.type != MI355FFT_C2C && args->output->ptr == args->input->ptr)
    return fail(MI355FFT_ERR_INVALID, "input and output must be different buffers for this plan type");
  if (args->input_offset_bytes + plan->ir.in_bytes > args->input->bytes)
    return fail(MI355FFT_ERR_INVALID, "input buffer/view too small: need %llu bytes at offset %llu, have %llu", (unsigned long long)plan->ir.in_bytes,
                (unsigned long long)args->input_offset_bytes, (unsigned long long)args->input->bytes);
  if (out_off + plan->ir.out_bytes > out->bytes)
    return fail(MI355FFT_ERR_INVALID, "output buffer/view too small: need %llu bytes at offset %llu, have %llu", (unsigned long long)plan->ir.out_bytes,
                (unsigned long long)out_off, (unsigned long long)out->bytes);
  if (d.type == MI355FFT_FFTCONV && args->kernel_offset_bytes + plan->ir.kernel_bytes > args->kernel->bytes)
    return fail(MI355FFT_ERR_INVALID, "kernel buffer too small: need %llu bytes, have %llu", (unsigned long long)plan->ir.kernel_bytes,
                (unsigned long long)args->kernel->bytes);
  // workspace: caller's temp when it is big enough and does not alias input/output, else the plan's arena
  void* work = nullptr;
  if (plan->ir.work_bytes) {
    if (args->temp && args->temp->bytes >= plan->ir.work_bytes && args->temp->ptr != args->input->ptr && args->temp->ptr != out->ptr) {
      work = args->temp->ptr;
    } else {
      if (!plan->arena) {
        HIP_TRY(hipSetDevice(plan->dev->ordinal));
        const hipError_t e = hipMalloc(&plan->arena, plan->ir.work_bytes);
        if (e != hipSuccess) return fail(MI355FFT_ERR_NOMEM, "hipMalloc(workspace, %llu bytes) failed: %s", (unsigned long long)plan->ir.work_bytes, hipGetErrorString(e));
        plan->arena_bytes = plan->ir.work_bytes;
      }
      work = plan->arena;
    }
  }
  char* base[5];
  base[BUF_INPUT] = (char*)args->input->ptr + args->input_offset_bytes;
  base[BUF_OUTPUT] = (char*)out->ptr + out_off;
  base[BUF_WORK] = (char*)work;
  base[BUF_KERNEL] = args->kernel ? (char*)args->kernel->ptr + args->kernel_offset_bytes : nullptr;
  base[BUF_TABLE] = (char*)plan->table;
  for (const Step& s : plan->ir.steps) {
    RecordedOp op;
    op.step = s;
    for (int i = 0; i < 5; ++i) op.ptr[i] = s.p[i].buf == BUF_NONE ? nullptr : base[s.p[i].buf] + s.p[i].off;
    enc->ops.push_back(op);
  }
  enc->deps.push_back(plan->alive);
  enc->deps.push_back(args->input->alive);
  enc->deps.push_back(out->alive);
  if (args->kernel) enc->deps.push_back(args->kernel->alive);
  if (args->temp && work == args->temp->ptr) enc->deps.push_back(args->temp->alive);
  return MI355FFT_OK;
}

MI_API int mi355fft_plan_destroy(mi355fft_plan* plan) {
  if (!plan || plan->destroyed) return MI355FFT_OK;
  *plan->alive = false;
  (void)hipSetDevice(plan->dev->ordinal);
  (void)hipStreamSynchronize(plan->dev->stream);
  if (plan->table) (void)hipFree(plan->table);
  if (plan->arena) (void)hipFree(plan->arena);
  plan->table = plan->arena = nullptr;
  plan->destroyed = true;
  return MI355FFT_OK;
}

MI_API int mi355fft_plan_release(mi355fft_plan* plan) {
  if (!plan) return MI355FFT_OK;
  mi355fft_plan_destroy(plan);
  delete plan;
  return MI355FFT_OK;
}

// ---- encoder / queue -------------------------------------------------------------------------------
MI_API int mi355fft_encoder_begin(mi355fft_device* dev, mi355fft_encoder** out) {
  if (!dev || !out) return fail(MI355FFT_ERR_INVALID, "Expected a device and an out pointer");
  mi355fft_encoder* e = new mi355fft_encoder();
  e->dev = dev;
  *out = e;
  return MI355FFT_OK;
}

MI_API int mi355fft_encoder_copy_buffer(mi355fft_encoder* enc, mi355fft_buffer* src, uint64_t src_offset, mi355fft_buffer* dst,
                                        uint64_t dst_offset, uint64_t bytes) {
  if (!enc || !src || !dst) return fail(MI355FFT_ERR_INVALID, "copyBufferToBuffer: encoder, source and destination are required");
  if (src_offset + bytes > src->bytes || dst_offset + bytes > dst->bytes) return fail(MI355FFT_ERR_INVALID, "copyBufferToBuffer: range exceeds buffer size");
  if (bytes % 4 || src_offset % 4 || dst_offset % 4) return fail(MI355FFT_ERR_INVALID, "copyBufferToBuffer: offsets and size must be multiples of 4");
  RecordedOp op;
  op.step.kind = ST_COPY;
  op.step.i[0] = (int64_t)bytes;
  std::memset(op.ptr, 0, sizeof op.ptr);
  op.ptr[0] = (char*)src->ptr + src_offset;
  op.ptr[1] = (char*)dst->ptr + dst_offset;
  enc->ops.push_back(op);
  enc->deps.push_back(src->alive);
  enc->deps.push_back(dst->alive);
  return MI355FFT_OK;
}

MI_API int mi355fft_encoder_discard(mi355fft_encoder* enc) { delete enc; return MI355FFT_OK; }

MI_API int mi355fft_encoder_finish(mi355fft_encoder* enc, int use_graph, mi355fft_commands** out) {
  if (!enc || !out) return fail(MI355FFT_ERR_INVALID, "finish: encoder and out pointer are required");
  *out = nullptr;
  std::unique_ptr<mi355fft_encoder> owner(enc);
  mi355fft_commands* c = new mi355fft_commands();
  c->dev = enc->dev;
  c->ops.swap(enc->ops);
  c->deps.swap(enc->deps);
  // use_graph: 0 = op list, 1 = hipGraph, 2 = auto: a graph pays off once a list has many launches (measured:
  // a 1-launch list replays in 5 us as an op list and 11 us as a graph; 64-launch lists are on par)
  if (use_graph == 2) use_graph = c->ops.size() >= 8 ? 1 : 0;
  if (use_graph && !c->ops.empty()) {
    mi355fft_device* dev = c->dev;
    hipError_t e = hipSetDevice(dev->ordinal);
    if (e == hipSuccess) e = hipStreamBeginCapture(dev->capture_stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) { destroy_commands(c); return fail(MI355FFT_ERR_HIP, "hipStreamBeginCapture failed: %s", hipGetErrorString(e)); }
    HipLauncher l;
    l.stream = dev->capture_stream;
    l.sticky = dev->sticky;
    const int rc = replay(c->ops, l);
    e = hipStreamEndCapture(dev->capture_stream, &c->graph);
    if (rc) { destroy_commands(c); return rc; }
    if (e != hipSuccess) { destroy_commands(c); return fail(MI355FFT_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e)); }
    e = hipGraphInstantiate(&c->exec, c->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { destroy_commands(c); return fail(MI355FFT_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
  }
  *out = c;
  return MI355FFT_OK;
}

MI_API int mi355fft_queue_submit(mi355fft_device* dev, mi355fft_commands* cmds) {
  if (!dev || !cmds) return fail(MI355FFT_ERR_INVALID, "submit: device and command list are required");
  if (cmds->dev != dev) return fail(MI355FFT_ERR_INVALID, "command list belongs to a different device");
  for (const AliveToken& t : cmds->deps) if (!*t) return fail(MI355FFT_ERR_DESTROYED, "command buffer references a destroyed plan or buffer");
  HIP_TRY(hipSetDevice(dev->ordinal));
  for (const RecordedOp& op : cmds->ops) if (op.step.kind == ST_XCD_FUSED || op.step.kind == ST_XCD_RES) dev->sticky_armed = true;
  if (cmds->exec) { HIP_TRY(hipGraphLaunch(cmds->exec, dev->stream)); return MI355FFT_OK; }
  HipLauncher l;
  l.stream = dev->stream;
  l.sticky = dev->sticky;
  return replay(cmds->ops, l);
}

MI_API int mi355fft_commands_release(mi355fft_commands* cmds) {
  if (!cmds) return MI355FFT_OK;
  (void)hipSetDevice(cmds->dev->ordinal);
  (void)hipStreamSynchronize(cmds->dev->stream);
  destroy_commands(cmds);
  return MI355FFT_OK;
}

MI_API int mi355fft_queue_wait(mi355fft_device* dev) {
  if (!dev) return fail(MI355FFT_ERR_INVALID, "Expected a device");
  return sync_and_check(dev);
}

// ---- synthetic inputs / reductions -----------------------------------------------------------------
MI_API int mi355fft_fill_random(mi355fft_device* dev, mi355fft_buffer* buf, uint64_t offset_bytes, uint64_t row_floats, uint64_t rows,
                                uint32_t seed0, uint64_t first_transform) {
  if (!dev || !buf) return fail(MI355FFT_ERR_INVALID, "fill_random: device and buffer are required");
  if (offset_bytes % 4 || offset_bytes + row_floats * rows * 4 > buf->bytes) return fail(MI355FFT_ERR_INVALID, "fill_random: range exceeds buffer");
  if (!row_floats || !rows) return MI355FFT_OK;
  HIP_TRY(hipSetDevice(dev->ordinal));
  const uint64_t total = row_floats * rows;
  const unsigned grid = (unsigned)std::min<uint64_t>((total + 255) / 256, (uint64_t)dev->compute_units * 16);
  hipLaunchKernelGGL(fill_random_kernel, dim3(grid), dim3(256), 0, dev->stream, (float*)((char*)buf->ptr + offset_bytes),
                     (unsigned long long)row_floats, (unsigned long long)rows, (unsigned)seed0, (unsigned long long)first_transform);
  HIP_TRY(hipGetLastError());
  return MI355FFT_OK;
}

MI_API int mi355fft_diff_sumsq(mi355fft_device* dev, mi355fft_buffer* a, uint64_t a_offset_bytes, mi355fft_buffer* b, uint64_t b_offset_bytes,
                               double alpha, uint64_t count, double* out) {
  if (!dev || !a || !out) return fail(MI355FFT_ERR_INVALID, "sumsq: device, buffer and out are required");
  if (a_offset_bytes % 4 || a_offset_bytes + count * 4 > a->bytes) return fail(MI355FFT_ERR_INVALID, "sumsq: range exceeds buffer a");
  if (b && (b_offset_bytes % 4 || b_offset_bytes + count * 4 > b->bytes)) return fail(MI355FFT_ERR_INVALID, "sumsq: range exceeds buffer b");
  *out = 0.0;
  if (!count) return MI355FFT_OK;
  HIP_TRY(hipSetDevice(dev->ordinal));
  const unsigned grid = (unsigned)std::min<uint64_t>((count + 255) / 256, (uint64_t)dev->compute_units * 8);
  double* partial = nullptr;
  HIP_TRY(hipMalloc((void**)&partial, grid * sizeof(double)));
  hipLaunchKernelGGL(diff_sumsq_kernel, dim3(grid), dim3(256), 0, dev->stream, (const float*)((const char*)a->ptr + a_offset_bytes),
                     b ? (const float*)((const char*)b->ptr + b_offset_bytes) : (const float*)nullptr, alpha, (unsigned long long)count, partial);
  hipError_t e = hipGetLastError();
  std::vector<double> host(grid);
  int sticky_rc = MI355FFT_OK;
  if (e == hipSuccess) sticky_rc = sync_and_check(dev);     // the reduced buffers may be results of a submit that timed out
  if (e == hipSuccess && !sticky_rc) e = hipMemcpy(host.data(), partial, grid * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(partial);
  if (sticky_rc) return sticky_rc;
  if (e != hipSuccess) return fail(MI355FFT_ERR_HIP, "sumsq failed: %s", hipGetErrorString(e));
  double s = 0.0;
  for (double v : host) s += v;
  *out = s;
  return MI355FFT_OK;
}

MI_API int mi355fft_sumsq(mi355fft_device* dev, mi355fft_buffer* buf, uint64_t offset_bytes, uint64_t count, double* out) {
  return mi355fft_diff_sumsq(dev, buf, offset_bytes, nullptr, 0, 0.0, count, out);
}
