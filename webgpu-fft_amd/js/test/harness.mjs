// minimal test runner (Node 12 has no node:test): sequential async tests, non-zero exit on failure
const tests = [];
export function test(name, fn) { tests.push({ name, fn }); }
export function assert(cond, msg) { if (!cond) throw new Error(msg || "assertion failed"); }
export function assertThrows(fn, re, what) {
  try { fn(); } catch (e) { if (re && !re.test(e.message)) throw new Error((what || "throws") + ": message " + JSON.stringify(e.message) + " does not match " + re); return; }
  throw new Error((what || "throws") + ": did not throw");
}
export function deepEqual(a, b) {
  if (a === b) return true;
  if (typeof a !== typeof b || a === null || b === null || typeof a !== "object") return false;
  if (Array.isArray(a) !== Array.isArray(b)) return false;
  const ka = Object.keys(a).sort(), kb = Object.keys(b).sort();
  if (ka.length !== kb.length || !ka.every((k, i) => k === kb[i])) return false;
  return ka.every((k) => deepEqual(a[k], b[k]));
}
export async function run() {
  let failed = 0;
  for (const t of tests) {
    try { await t.fn(); console.log("ok   - " + t.name); } catch (e) { failed++; console.log("FAIL - " + t.name + "\n       " + (e && e.stack ? e.stack.split("\n").slice(0, 4).join("\n       ") : e)); }
  }
  console.log("# " + (tests.length - failed) + "/" + tests.length + " passed");
  process.exit(failed ? 1 : 0);
}
