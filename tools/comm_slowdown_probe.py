#!/usr/bin/env python3
"""developer tool (round 6): bench.py --force-comm (the RCCL path on one rank) runs the SAME whole-list kernel 5 % slower
than the plain line and its bare stores 22 % slower (profiles/r6_bench_force_comm.json: 1,480 vs 1,400 us, box ceiling
5.74 vs 7.33 TB/s).  Which step of bringing RCCL up costs that?  One process, one partition (BASELINE config 2), the
list kernel (HIP events around pll_update_partials) and the bare stores (pll_amd_write_ceiling) measured after each step:
  0  nothing but libpll_amd (and torch imported)
  1  torch.distributed.init_process_group("nccl", world_size=1) + one all-reduce
  2  a SECOND partition created now (its arenas allocated after RCCL came up), measured instead of the first
  3  pll_amd_comm_init on the first partition (the library's own communicator), lnL through the all-reduce
  python3 tools/comm_slowdown_probe.py [order]      order: a permutation / subset of 123, default 123"""
import os, sys, ctypes
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault("PLLHIP_DEVELOPER", "1")
os.environ.setdefault("PLL_AMD_AUTO_MIRROR_MB", "0")
import torch
import torch.distributed as dist
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP

order = sys.argv[1] if len(sys.argv) > 1 else "123"
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
torch.cuda.set_device(0)
lib = libpll_amd.load()
lib.lib.pll_amd_set_device(0)
S, R, taxa = 4, 4, 64
plan = W.balanced_tree(taxa, seed=42)
cats = lib.compute_gamma_cats(W.GAMMA_ALPHA, R)
seqs = W.simulated_alignment(plan, sites, W.GTR_RATES, W.GTR_FREQS, cats, seed=42)
fi = [0] * R


def measure(p, label):
    for _ in range(30):
        p.update_partials(plan.ops)
    p.wait()
    us = []
    for _ in range(9):
        p.wait()
        p.timer_start()
        for _ in range(10):
            p.update_partials(plan.ops)
        us.append(p.timer_stop_ms() * 1e3 / 10)
    us.sort()
    # a step as bench.py times it: list + edge lnL, wall clock, synchronised at the end only
    import time
    for _ in range(5):
        lnl = p.compute_edge_loglikelihood(*plan.root_edge, fi)
    p.wait()
    t0 = time.perf_counter()
    for _ in range(50):
        p.update_partials(plan.ops)
        lnl = p.compute_edge_loglikelihood(*plan.root_edge, fi)
    p.wait()
    step = (time.perf_counter() - t0) / 50 * 1e6
    ms, nbytes = p.write_ceiling(plan.ops, 20)
    for _ in range(3):
        p.update_partials(plan.ops)
    p.wait()
    print("%-64s list %8.1f us (min %8.1f)   step %8.1f us   bare stores %8.1f us = %6.1f GB/s   lnL %.6f"
          % (label, us[len(us) // 2], us[0], step, ms * 1e3, nbytes / ms / 1e6, lnl), flush=True)


p = W.setup_partition(lib, plan, seqs, S, R, ATTRIB_PATTERN_TIP)
measure(p, "0 libpll_amd alone")
measure(p, "0 again")
for ch in order:
    if ch == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        measure(p, "1 torch.distributed up (nccl, one rank), one all-reduce done")
    elif ch == "2":
        p2 = W.setup_partition(lib, plan, seqs, S, R, ATTRIB_PATTERN_TIP)
        measure(p2, "2 a partition created now")
        measure(p, "2 the first partition again")
        del p2
    elif ch == "3":
        buf = ctypes.create_string_buffer(128)
        if not lib.lib.pll_amd_comm_unique_id(buf):
            raise SystemExit("pll_amd_comm_unique_id failed: " + lib.errmsg())
        p.comm_init(0, 1, bytes(buf.raw))
        measure(p, "3 the library's communicator on the first partition")
measure(p, "end: the first partition once more")
if dist.is_initialized():
    dist.destroy_process_group()
