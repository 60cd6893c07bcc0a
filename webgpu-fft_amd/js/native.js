// native.js — loads the N-API addon (webgpu-fft_amd/lib/mi355fft.node, built by napi/Makefile).
// There is no JavaScript or CPU fallback: if the addon is missing the import throws.
import { createRequire } from "module";
import { fileURLToPath } from "url";
import path from "path";

const require = createRequire(import.meta.url);
const here = path.dirname(fileURLToPath(import.meta.url));
const addonPath = path.join(here, "..", "lib", "mi355fft.node");

let native;
try {
  native = require(addonPath);
} catch (e) {
  throw new Error(
    "webgpufft-mi355: cannot load the HIP addon at " + addonPath + " (" + e.message +
      "). Build it with `python -c \"import __graft_entry__ as g; g.build()\"` (make -C webgpu-fft_amd/csrc && make -C webgpu-fft_amd/napi)."
  );
}
if (native.abiVersion() !== 1) throw new Error("webgpufft-mi355: addon ABI version mismatch");

export default native;
