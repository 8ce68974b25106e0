#!/usr/bin/env python3
"""Hunt for the crash of a reference client program seen once in round 3 (Weak 2).

Runs the reference's unmodified test programs (oracle/_ref/reftest_*) against libpll_amd.so in a loop -- every
program, every attribute mode, several processes side by side on one GPU, optionally with a parent that holds GPU
state of its own the way the pytest process does -- with oracle/segv_backtrace.so preloaded and the C library's
heap checks switched on, and compares every output with the expected text.  Anything but "rc 0, expected text"
is logged with the child's stderr.

  python3 tools/crash_soak.py --runs 20000 --workers 12 --log gpurun_out/crash_soak.log

Knobs per run are drawn at random (seeded) so that a hit names its configuration:
  MALLOC_PERTURB_ (freed/allocated memory filled with a byte), MALLOC_CHECK_=3 (a third of the runs),
  PLLHIP_SPIN=0/1, PLLHIP_AA_EXACT=0/1, mirror mode on/off is the program's own business (they use pll_show_*).
"""
import argparse
import os
import random
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden", "reference_out")
EXAMPLES = ["example_rooted", "example_rooted-tacg", "example_heterotachy", "example_newton"]
TESTS = ["example_unrooted"] + EXAMPLES + ["00010_NMDU_lkcalc", "00011_NMAU_lkcalc", "00012_NMOU_lkcalc",
         "00020_NMDR_lkcalc", "00021_NMAR_lkcalc", "00022_NMOR_lkcalc", "00030_NMDU_gamma", "00032_NMOU_gamma",
         "alpha-cats", "derivatives", "derivatives-oddstates", "hky", "pmatrix", "protein-models"]
MODES = [[], ["tv"], ["avx2"], ["avx2", "tv"], ["avx"], ["sse", "tv"]]


def expected_text(name):
    where = os.path.join(BIN, "expected") if (name == "protein-models" or name in EXAMPLES) else OUT
    path = os.path.join(where, name + ".out")
    return open(path).read() if os.path.exists(path) else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=2000)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=0, help="stop after this long (0: run all)")
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "crash_soak.log"))
    ap.add_argument("--only", default="", help="comma-separated program names")
    ap.add_argument("--skip", default="protein-models", help="comma-separated program names left out (slow ones)")
    ap.add_argument("--parent-gpu", type=int, default=1, help="the parent keeps a partition alive on the GPU")
    ap.add_argument("--exact", default="both", choices=["both", "0", "1"])
    ap.add_argument("--malloc-check", type=float, default=0.33, help="share of the runs with MALLOC_CHECK_=3")
    ap.add_argument("--spin", default="mix", choices=["mix", "0", "1"], help="PLLHIP_SPIN of the runs (mix: 0 in a quarter)")
    ap.add_argument("--env", action="append", default=[], help="KEY=VALUE for every run (may repeat)")
    args = ap.parse_args()

    names = [t for t in TESTS if t not in args.skip.split(",")]
    if args.only:
        names = [t for t in names if t in args.only.split(",")]
    combos = []
    for n in names:
        if not os.path.exists(os.path.join(BIN, "reftest_" + n)):
            continue
        for m in (MODES if not n.startswith("example_") else [[]]):
            combos.append((n, m))
    exp = {n: expected_text(n) for n in names}
    skip_text = open(os.path.join(OUT, "skip.out")).read().strip()
    helper = os.path.join(ROOT, "oracle", "segv_backtrace.so")

    keep = None
    if args.parent_gpu:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import libpll_amd
        from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
        from helpers import make_case, build_partition
        lib = libpll_amd.load()
        case = make_case(4, "random", 12, 500, seed=3)
        keep = build_partition(lib, case, ATTRIB_PATTERN_TIP)
        keep.update_partials(case["plan"].ops)

    os.makedirs(os.path.dirname(args.log), exist_ok=True)
    log = open(args.log, "a")
    lock = threading.Lock()
    state = {"next": 0, "bad": 0, "done": 0, "by": {}}
    t0 = time.time()

    def say(s):
        with lock:
            log.write(s + "\n")
            log.flush()

    say("== crash_soak: %d runs, %d workers, %d (program, mode) combinations, seed %d, parent_gpu %d, exact %s, spin %s, env %s"
        % (args.runs, args.workers, len(combos), args.seed, args.parent_gpu, args.exact, args.spin, args.env))

    def worker(w):
        rng = random.Random(args.seed * 1000 + w)
        while True:
            with lock:
                i = state["next"]
                if i >= args.runs or (args.seconds and time.time() - t0 > args.seconds):
                    return
                state["next"] = i + 1
            name, mode = combos[rng.randrange(len(combos))]
            env = dict(os.environ)
            knobs = {}
            knobs["MALLOC_PERTURB_"] = str(rng.randrange(1, 256))
            if rng.random() < args.malloc_check:
                knobs["MALLOC_CHECK_"] = "3"
            knobs["PLLHIP_SPIN"] = args.spin if args.spin != "mix" else ("0" if rng.random() < 0.25 else "1")
            for kv in args.env:
                knobs[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
            knobs["PLLHIP_AA_EXACT"] = {"both": str(rng.randrange(2)), "0": "0", "1": "1"}[args.exact]
            knobs["SEGV_BACKTRACE_MAPS"] = "1"
            env.update(knobs)
            if os.path.exists(helper):
                env["LD_PRELOAD"] = helper
            try:
                run = subprocess.run([os.path.join(BIN, "reftest_" + name)] + mode, capture_output=True, text=True,
                                     timeout=600, env=env, errors="replace")
                rc, out, err = run.returncode, run.stdout, run.stderr
            except subprocess.TimeoutExpired as e:
                rc, out, err = -999, "", "TIMEOUT " + str(e)
            ok = rc == 0 and (out == exp[name] or out.strip() == skip_text or exp[name] is None)
            # 20-state programs on the approximate path print the same digits or they do not: report separately
            with lock:
                state["done"] += 1
                state["by"][name] = state["by"].get(name, 0) + 1
                if not ok:
                    state["bad"] += 1
            if not ok:
                kind = "rc %d" % rc if rc != 0 else "OUTPUT DIFFERS"
                cut = err.find("--- maps")   # (the backtrace first; of the map only what is executable)
                head = err if cut < 0 else err[:cut] + "--- maps (r-x)\n" + "\n".join(
                    l for l in err[cut:].splitlines() if " r-x" in l or "r-xp" in l)
                say("== run %d: %s %s: %s; knobs %s\n%s\n-- last output lines:\n%s"
                    % (i, name, "+".join(mode) or "cpu", kind, knobs, head[:12000], "\n".join(out.splitlines()[-4:])))
                if rc == 0:
                    gl, el = out.splitlines(), (exp[name] or "").splitlines()
                    diff = [(k, g, e) for k, (g, e) in enumerate(zip(gl, el)) if g != e][:3]
                    say("   first differences: %s" % (diff,))

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(args.workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.time() - t0
    say("== done: %d runs in %.0f s, %d not clean; per program %s" % (state["done"], dt, state["bad"], state["by"]))
    print("crash_soak: %d runs in %.0f s, %d not clean" % (state["done"], dt, state["bad"]))
    if keep is not None:
        keep.destroy()
    return 1 if state["bad"] else 0


if __name__ == "__main__":
    sys.exit(main())
