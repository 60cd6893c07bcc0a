/*
 * mi355fft.h — C ABI of libmi355fft.so, the MI355X (gfx950) FFT engine behind the webgpufft plan API.
 *
 * This is the drop-in boundary for the reference's one data-parallel hot path (batched c2c / r2c /
 * c2r / fftconv on interleaved-complex f32).  Every entry point is `extern "C"`, takes plain pointers
 * and sizes, returns an int status (0 = ok) and leaves a human-readable message retrievable with
 * mi355fft_last_error() (thread-local), which the N-API addon turns into `throw new Error(msg)` — the
 * reference's error behaviour (synchronous throws, SURVEY.md 8b "Errors").
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference repo).
 * The host-side bindings (N-API for Node, ctypes for the Python harness) are shown in INTEGRATION.md.
 */
#ifndef MI355FFT_H
#define MI355FFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355FFT_ABI_VERSION 1
#define MI355FFT_MAX_RANK 8

/* status codes */
#define MI355FFT_OK 0
#define MI355FFT_ERR_INVALID 1     /* bad argument / option validation failed (reference: throw new Error) */
#define MI355FFT_ERR_UNSUPPORTED 2 /* valid in the reference, not built yet (see DESIGN.md "out of scope") */
#define MI355FFT_ERR_HIP 3         /* a HIP runtime call failed */
#define MI355FFT_ERR_DESTROYED 4   /* exec after destroy ("plan destroyed", c2c.js:3608) */
#define MI355FFT_ERR_NOMEM 5

typedef struct mi355fft_device mi355fft_device;   /* replaces the GPUDevice argument (base_plan.js:32-38) */
typedef struct mi355fft_buffer mi355fft_buffer;   /* replaces GPUBuffer ({size, destroy()}, common.js:55-57) */
typedef struct mi355fft_plan mi355fft_plan;       /* replaces C2CPlan/R2CPlan/C2RPlan/FftConvPlan/FftPlan */
typedef struct mi355fft_encoder mi355fft_encoder; /* replaces GPUCommandEncoder: plan.exec RECORDS into it */
typedef struct mi355fft_commands mi355fft_commands; /* replaces GPUCommandBuffer = encoder.finish() */

/* createPlan opts.type (runtime/create_plan.js:12-23); only the hot-path members */
enum { MI355FFT_C2C = 0, MI355FFT_R2C = 1, MI355FFT_C2R = 2, MI355FFT_FFTCONV = 3,
       /* real-to-real transforms over real f32 buffers (runtime/plans/dct_fft.js; create_plan.js:15) */
       MI355FFT_DCT1 = 4, MI355FFT_DCT2 = 5, MI355FFT_DCT3 = 6, MI355FFT_DCT4 = 7,
       MI355FFT_DST1 = 8, MI355FFT_DST2 = 9, MI355FFT_DST3 = 10, MI355FFT_DST4 = 11 };
/* opts.direction: forward = exp(-i...), inverse = exp(+i...) (kernels/stockham_stage.js:35) */
enum { MI355FFT_FORWARD = 0, MI355FFT_INVERSE = 1 };
/* opts.normalize (runtime/common.js:35-40): none -> 1, backward -> 1/N on inverse only, unitary -> 1/sqrt(N) */
enum { MI355FFT_NORM_NONE = 0, MI355FFT_NORM_BACKWARD = 1, MI355FFT_NORM_UNITARY = 2 };
/* opts.fftConv.mode / boundary / outputLayout (runtime/plans/fftconv.js:329-338) */
enum { MI355FFT_CONVOLUTION = 0, MI355FFT_CORRELATION = 1 };
enum { MI355FFT_CIRCULAR = 0, MI355FFT_LINEAR_FULL = 1, MI355FFT_LINEAR_SAME = 2, MI355FFT_LINEAR_VALID = 3 };
enum { MI355FFT_KERNEL_MAJOR = 0, MI355FFT_BATCH_MAJOR = 1 };

/* One side (input or output) of a plan after the host has resolved layout.{strides,offsetElements,
 * batchStrideElements}, layout.whdcn and fftConv.channelPolicy (runtime/layout_semantics.js:178-232,
 * runtime/plans/fftconv.js:213-281).  Units are ELEMENTS of the side's element type (complex for c2c,
 * real for the real side of r2c/c2r).  strided == 0 means dense: axis 0 fastest, batch outermost
 * (kernels/nd_line_base.js:9-12). */
typedef struct mi355fft_side_layout {
  int32_t strided;
  int32_t reserved;
  int64_t strides[MI355FFT_MAX_RANK];
  int64_t offset_elements;
  int64_t batch_stride_elements;
} mi355fft_side_layout;

/* opts.ioView.{input,output} after normalisation (runtime/ioview.js:7-37): the physical view has `shape` and sits at
 * logical coordinate `offset` (may be negative).  Input: logical[c] = view[c - offset] inside the view, 0 outside
 * (kernels/ioview.js generateEmbedComplexWGSL).  Output: view[v] = logical[v + offset] where that is inside the logical
 * domain; other view elements are zeroed when clear_outside, else left untouched (generateExtractComplexWGSL). */
typedef struct mi355fft_io_view {
  int32_t enabled;
  int32_t clear_outside;
  int64_t shape[MI355FFT_MAX_RANK];
  int64_t offset[MI355FFT_MAX_RANK];
} mi355fft_io_view;

/* opts.zeroPad.{read,write} (runtime/zero_pad.js:11-45): logical elements outside [start, end) are zeroed — read: after
 * input load/embedding and before the transform; write: after transform + normalisation, before output extraction. */
typedef struct mi355fft_zero_range {
  int32_t enabled;
  int32_t reserved;
  int64_t start[MI355FFT_MAX_RANK];
  int64_t end[MI355FFT_MAX_RANK];
} mi355fft_zero_range;

/* createPlan(device, opts) option object (docs/API.md:9-109), hot-path subset. */
typedef struct mi355fft_plan_desc {
  uint32_t struct_size;            /* sizeof(mi355fft_plan_desc): ABI guard */
  int32_t type;                    /* MI355FFT_C2C ... */
  int32_t rank;                    /* shape.length, 1..MI355FFT_MAX_RANK */
  int32_t direction;
  int32_t normalize;
  int32_t in_place;                /* c2c only (docs/API.md:135) */
  int64_t shape[MI355FFT_MAX_RANK];/* logical transform domain, axis 0 fastest */
  int64_t batch;                   /* >= 1 */
  mi355fft_side_layout input;
  mi355fft_side_layout output;
  /* type == MI355FFT_FFTCONV only */
  int32_t conv_mode;
  int32_t conv_boundary;
  int32_t conv_kernel_count;       /* fftConv.kernelCount, >= 1 */
  int32_t conv_output_layout;      /* dense output: [kernel][batch][logical] or [batch][kernel][logical] */
  int64_t conv_kernel_shape[MI355FFT_MAX_RANK]; /* all 0 => same as shape */
  int64_t conv_output_kernel_stride_elements;   /* strided output: lane step per kernel (fftconv.js:868-871) */
  /* c2c, r2c, c2r: padding / embedding / range zeroing (docs/API.md "ioView", "zeroPad").  When an ioView side is enabled,
   * that side's layout (dense or strided) describes the VIEW's physical shape; for r2c the output side (for c2r the input
   * side) views and ranges live on the packed domain (shape[0]/2+1 bins along axis 0). */
  mi355fft_io_view io_input;
  mi355fft_io_view io_output;
  mi355fft_zero_range zero_read;
  mi355fft_zero_range zero_write;
  /* c2c only: createFftPlan({axes}) (plan.js:1307,1335-1339) — bit a set => axis a is transformed; 0 => all axes.  The
   * normalisation factor still uses prod(shape) over ALL axes, as the reference's does (plan.js:1334,1382). */
  uint32_t axes_mask;
  uint32_t reserved2;
} mi355fft_plan_desc;

/* plan.exec(commandEncoder, {input, output?, temp?, inputOffsetBytes, outputOffsetBytes, kernel?})
 * (runtime/plans/c2c.js:3607-3612, runtime/plans/fftconv.js:1415-1429) */
typedef struct mi355fft_exec_args {
  uint32_t struct_size;
  uint32_t reserved;
  mi355fft_buffer* input;
  mi355fft_buffer* output;         /* NULL => in place (c2c, in_place plans only) */
  mi355fft_buffer* temp;           /* optional caller workspace; NULL => plan-owned arena */
  mi355fft_buffer* kernel;         /* fftconv: kernel_count packed kernels of prod(kernel_shape) complex */
  uint64_t input_offset_bytes;     /* multiples of 8 (plan.js:861-862) */
  uint64_t output_offset_bytes;
  uint64_t kernel_offset_bytes;
} mi355fft_exec_args;

/* ---- library ------------------------------------------------------------------------------------- */
int mi355fft_abi_version(void);
/* thread-local message of the last failing call on this thread ("" if none) */
const char* mi355fft_last_error(void);

/* ---- device: replaces `navigator.gpu.requestAdapter().requestDevice()` + device.queue -------------- */
int mi355fft_device_count(int* count);
int mi355fft_device_open(int ordinal, mi355fft_device** out);
int mi355fft_device_close(mi355fft_device* dev);
/* device.limits analogue: total/free HBM bytes, compute units, gcn arch name (<= 63 chars) */
int mi355fft_device_info(mi355fft_device* dev, uint64_t* hbm_total, uint64_t* hbm_free, int* compute_units, char arch[64]);
/* the HIP stream (hipStream_t) all submitted work runs on — for callers that time with hipEvents */
void* mi355fft_device_stream(mi355fft_device* dev);

/* ---- buffers: device.createBuffer({size}) / buffer.destroy() / queue.writeBuffer / mapAsync readback
 *      (utils/webgpu.js:9-23, 29-55) ------------------------------------------------------------------- */
int mi355fft_buffer_alloc(mi355fft_device* dev, uint64_t bytes, mi355fft_buffer** out);
/* non-owning view of device memory the caller allocated (hipMalloc / torch tensor data_ptr) */
int mi355fft_buffer_wrap(mi355fft_device* dev, void* device_ptr, uint64_t bytes, mi355fft_buffer** out);
int mi355fft_buffer_free(mi355fft_buffer* buf); /* idempotent for NULL */
uint64_t mi355fft_buffer_size(const mi355fft_buffer* buf);
void* mi355fft_buffer_device_ptr(const mi355fft_buffer* buf);
/* queue.writeBuffer(buffer, offset, data): ordered before later submits, returns after the host data
 * has been consumed */
int mi355fft_buffer_write(mi355fft_buffer* buf, uint64_t offset_bytes, const void* src, uint64_t bytes);
/* readback: waits for all submitted work, then copies device -> host */
int mi355fft_buffer_read(mi355fft_buffer* buf, uint64_t offset_bytes, void* dst, uint64_t bytes);

/* ---- plans: createPlan / createFftPlan (runtime/create_plan.js:12, plan.js:1298) ------------------- */
int mi355fft_plan_create(mi355fft_device* dev, const mi355fft_plan_desc* desc, mi355fft_plan** out);
/* plan.getWorkspaceSizeBytes() (runtime/plans/c2c.js:1154-1199): bytes of `temp` that make exec
 * allocation-free; the plan owns an arena of this size when temp is not supplied */
int mi355fft_plan_workspace_bytes(const mi355fft_plan* plan, uint64_t* bytes);
/* plan.exec(...): validates and RECORDS the transform into `enc`; nothing runs until submit
 * (README.md:37-39).  Several execs may share one encoder and run in order (complete.suite.js:648-651). */
int mi355fft_plan_exec(mi355fft_plan* plan, mi355fft_encoder* enc, const mi355fft_exec_args* args);
/* plan.destroy(): idempotent (base_plan.js:49-53); exec afterwards fails with "plan destroyed" */
int mi355fft_plan_destroy(mi355fft_plan* plan);
/* releases the handle itself (after destroy); separate so a destroyed JS plan object can still throw */
int mi355fft_plan_release(mi355fft_plan* plan);
/* number of kernel launches one exec records, and a short description of the chosen route
 * (the reference exposes route metadata the same way: plan._largeRouteMode, c2c.js:661-666) */
int mi355fft_plan_describe(const mi355fft_plan* plan, char* text, size_t text_bytes, int* launches_per_exec);

/* ---- encoder / queue: device.createCommandEncoder(), encoder.copyBufferToBuffer, encoder.finish(),
 *      queue.submit([cmds]), queue.onSubmittedWorkDone() ------------------------------------------------ */
int mi355fft_encoder_begin(mi355fft_device* dev, mi355fft_encoder** out);
int mi355fft_encoder_copy_buffer(mi355fft_encoder* enc, mi355fft_buffer* src, uint64_t src_offset, mi355fft_buffer* dst,
                                 uint64_t dst_offset, uint64_t bytes);
/* finish() consumes the encoder.  use_graph: 1 instantiates the recorded launches as a hipGraph (the "hipGraph
 * stage executor"); 0 keeps an op list that submit replays onto the stream; 2 = auto (graph for lists of >= 8
 * launches, op list below: a single launch replays faster as a plain launch). */
int mi355fft_encoder_finish(mi355fft_encoder* enc, int use_graph, mi355fft_commands** out);
int mi355fft_encoder_discard(mi355fft_encoder* enc);
/* queue.submit: enqueues; returns without waiting.  WebGPU command buffers are single-use; here a
 * command list may be submitted any number of times (bench.py replays one list per step). */
int mi355fft_queue_submit(mi355fft_device* dev, mi355fft_commands* cmds);
int mi355fft_commands_release(mi355fft_commands* cmds);
/* queue.onSubmittedWorkDone(): blocks the calling thread until the stream drains (the addon calls it
 * from a libuv worker so the JS thread gets a Promise) */
int mi355fft_queue_wait(mi355fft_device* dev);

/* ---- synthetic inputs (bench.py, tests): device twin of the oracle's seeded PRNG -------------------
 * Fills `count` float32 at buf+offset with ((u*2-1)*0.5) where u is draw (first_draw + i) of stream
 * stream_seed(seed0, transform) — rows of `row_floats` floats each belong to consecutive transforms
 * starting at `first_transform` (oracle.c: oracle_stream_seed / mulberry32_at).  Runs immediately on
 * the device stream (not recorded). */
int mi355fft_fill_random(mi355fft_device* dev, mi355fft_buffer* buf, uint64_t offset_bytes, uint64_t row_floats,
                         uint64_t rows, uint32_t seed0, uint64_t first_transform);
/* sum of squares (f64) of `count` float32 at buf+offset — Parseval / linearity checks at full size */
int mi355fft_sumsq(mi355fft_device* dev, mi355fft_buffer* buf, uint64_t offset_bytes, uint64_t count, double* out);
/* sum of squares of (a - alpha*b) over `count` float32 */
int mi355fft_diff_sumsq(mi355fft_device* dev, mi355fft_buffer* a, uint64_t a_offset_bytes, mi355fft_buffer* b,
                        uint64_t b_offset_bytes, double alpha, uint64_t count, double* out);

#ifdef __cplusplus
}
#endif
#endif /* MI355FFT_H */
