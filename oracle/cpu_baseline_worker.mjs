// cpu_baseline_worker.mjs — one Node process of bench.py's `cpu_baseline` leg (SURVEY.md 8d "CPU baseline beside it").
//
// TEST / MEASUREMENT INFRASTRUCTURE ONLY (oracle/): times the reference's Node CPU correctness path — fft1dRef, the
// restatement of src/utils/math.js:25-88 in oracle.mjs, pinned bit-exact to the reference-generated fixtures — on seeded
// inputs of the bench workload (math.js:150-158 generator).  bench.py starts one of these per host core and adds up the
// points; nothing in the product imports this file.
//   node cpu_baseline_worker.mjs <N> <transforms> <seed> <min_seconds>   -> one JSON line {"points":…, "seconds":…, "rounds":…}
import { fft1dRef, mulberry32, randomComplexInterleaved } from "./oracle.mjs";

const N = parseInt(process.argv[2], 10);
const transforms = parseInt(process.argv[3], 10);
const seed = parseInt(process.argv[4], 10) >>> 0;
const minSeconds = parseFloat(process.argv[5]);
const rng = mulberry32(seed);
const inputs = [];
for (let b = 0; b < transforms; b++) inputs.push(randomComplexInterleaved(N, rng));
let checksum = 0, rounds = 0;
const t0 = process.hrtime.bigint();
let seconds = 0;
do {
  for (let b = 0; b < transforms; b++) {
    const out = fft1dRef(inputs[b], N, "forward");
    checksum += out[0] + out[2 * N - 1];
  }
  rounds++;
  seconds = Number(process.hrtime.bigint() - t0) / 1e9;
} while (seconds < minSeconds);
process.stdout.write(JSON.stringify({ points: N * transforms * rounds, seconds: seconds, rounds: rounds, checksum: checksum, node: process.version }) + "\n");
