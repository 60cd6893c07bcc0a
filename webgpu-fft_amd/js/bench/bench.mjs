// bench.mjs — the reference's Node bench shape (bench/bench_1d_1024.js:26-65, bench/bench.js:30-69: warm-up, N timed
// exec+submit, one onSubmittedWorkDone, average ms) driven through the JavaScript host, for BASELINE configs 1-3.
//   node webgpu-fft_amd/js/bench/bench.mjs [cfg1|cfg2|cfg3|all]
import { requestDevice, createPlan } from "../index.js";
import native from "../native.js";

const CONFIGS = {
  cfg1: { N: 1024, batch: 1, warm: 10, iters: 200 },
  cfg2: { N: 1024, batch: 65536, warm: 5, iters: 50 },
  cfg3: { N: 1 << 20, batch: 4096, warm: 1, iters: 5 },
};

async function run(name) {
  const c = CONFIGS[name];
  const device = await requestDevice();
  const bytes = c.N * c.batch * 8;
  if (device.info.hbmFree < 2 * bytes + 2 * 2 ** 30) { console.log(JSON.stringify({ config: name, skipped: "not enough HBM" })); device.destroy(); return; }
  const input = device.createBuffer({ size: bytes, usage: GPUBufferUsage.STORAGE | GPUBufferUsage.COPY_DST });
  const output = device.createBuffer({ size: bytes, usage: GPUBufferUsage.STORAGE | GPUBufferUsage.COPY_SRC });
  native.fillRandom(device._h, input._h, 0, 2 * c.N, c.batch, 0x5eed0003, 0);   // synthetic input generated on the device
  const plan = createPlan(device, { type: "c2c", shape: [c.N], batch: c.batch, direction: "forward", normalize: "none" });
  // the reference records a fresh encoder per iteration; recording here is a host-side list append
  const step = () => { const enc = device.createCommandEncoder(); plan.exec(enc, { input, output }); const cb = enc.finish(); device.queue.submit([cb]); return cb; };
  let cbs = [];
  for (let i = 0; i < c.warm; i++) cbs.push(step());
  await device.queue.onSubmittedWorkDone();
  cbs.forEach((cb) => cb.release());
  cbs = [];
  const t0 = process.hrtime.bigint();
  for (let i = 0; i < c.iters; i++) cbs.push(step());
  await device.queue.onSubmittedWorkDone();
  const ms = Number(process.hrtime.bigint() - t0) / 1e6 / c.iters;
  cbs.forEach((cb) => cb.release());
  console.log(JSON.stringify({ config: name, N: c.N, batch: c.batch, avg_ms: ms, gpoints_per_s: (c.N * c.batch) / (ms * 1e6), route: plan._route.trim(),
    launches_per_exec: plan._launchesPerExec, node: process.version, host: "JavaScript (Node ESM) -> N-API -> C ABI -> HIP" }));
  plan.destroy(); input.destroy(); output.destroy(); device.destroy();
}

const which = process.argv[2] || "all";
(async () => { for (const n of which === "all" ? Object.keys(CONFIGS) : [which]) await run(n); })().catch((e) => { console.error(e); process.exit(1); });
