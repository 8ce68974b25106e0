"""world_size-2 and world_size-8 runs of the site-sharding scheme on CPU (gloo).

The multi-GPU path shards alignment columns across ranks with no data-path
collective; the only exchange is a sum of per-shard lnL (and of d/dd).  Here
two -- and, as the driver's scaling bench does, eight -- processes each evaluate
their shard -- through the oracle, since there is no GPU -- using the same
shard_bounds / site_range code bench.py and the product use, all-reduce the
scalar over gloo and must reproduce the unsharded value.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from libpll_amd import workload as W
from libpll_amd.pllapi import PllLibrary, ATTRIB_PATTERN_TIP
from oracle_api import Oracle, OracleRun
from helpers import make_case

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
orc = Oracle(os.path.join(sys.argv[1], "oracle", "liboracle.so"))
amd = PllLibrary(os.path.join(sys.argv[1], "libpll_amd", "libpll_amd.so"))
TOTAL = int(sys.argv[2])
case = make_case(4, "random", 10, TOTAL, seed=11)
S, R, plan = 4, 4, case["plan"]
dp = lambda a: a.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double))
vals, vecs, inv = np.zeros(S), np.zeros((S, S)), np.zeros((S, S))
rates = np.ascontiguousarray(case["rates"]); freqs = np.ascontiguousarray(case["freqs"])
assert amd.lib.pll_amd_eigen_decompose(S, dp(rates), dp(freqs), dp(vals), dp(vecs), dp(inv))
model = dict(states=S, rate_cats=R, rates=amd.compute_gamma_cats(0.7, R),
             rate_weights=np.full(R, 0.25), eigenvals=vals, eigenvecs=vecs, inv_eigenvecs=inv,
             freqs=freqs, pinv=0.0)
cmap = amd.map("nt")
codes = np.stack([cmap[np.frombuffer(s, dtype=np.uint8)].astype(np.uint8) for s in case["seqs"]])

def evaluate(lo, hi):
    o = OracleRun(orc, model, plan, ATTRIB_PATTERN_TIP, tipcodes=codes[:, lo:hi],
                  tipmap=np.zeros(256, dtype=np.uint32), pattern_weights=case["pw"][lo:hi])
    o.update_partials()
    e = plan.root_edge
    lnl = o.edge_loglikelihood(*e)
    d, dd = o.derivatives(o.sumtable(e[0], e[2], e[1], e[3]), 0.1)
    return np.array([lnl, d, dd])

b = W.shard_bounds(TOTAL, world, granule=256)
assert len(b) == world + 1 and all(x < y for x, y in zip(b, b[1:]))
mine = torch.from_numpy(evaluate(b[rank], b[rank + 1]))
dist.all_reduce(mine, op=dist.ReduceOp.SUM)
full = evaluate(0, TOTAL)
err = float(np.max(np.abs(mine.numpy() - full) / np.abs(full)))
assert err < 1e-13, (mine, full)
if rank == 0:
    print("SHARDED_OK bounds=%s err=%.2e" % (b, err))
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(script, nproc, *args):
    """`nproc` ranks of `script` the way the driver launches bench.py (torch.distributed.run, 127.0.0.1)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1", "--master-port",
                          str(free_port()), str(script), ROOT] + [str(a) for a in args],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout


# (world_size 8: what SCALE_rNN runs -- VERDICT r5 item 6a; eight ranges on multiples of 256 need a few thousand sites)
@pytest.mark.parametrize("nproc,total", [(2, 1000), (8, 4000)], ids=["two-ranks", "eight-ranks"])
def test_site_sharding_gloo(tmp_path, orc, amd, nproc, total):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    assert "SHARDED_OK" in launch(script, nproc, total)


# ---------------------------------------------------------------- the product's HOST logic per rank

HOST_WORKER = r'''
import ctypes as C
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from libpll_amd import workload as W
from libpll_amd.pllapi import PllLibrary

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
amd = PllLibrary(os.path.join(sys.argv[1], "libpll_amd", "libpll_amd.so"))

# (1) the site ranges every rank derives for itself: contiguous, on multiples of 256, none empty,
# the same on every rank -- and the last one is where the ascertainment-bias sites go
total = 10_000
b = W.shard_bounds(total, world)
assert b[0] == 0 and b[-1] == total and all(x < y for x, y in zip(b, b[1:]))
assert all(x % 256 == 0 for x in b[1:-1])
mine = torch.tensor(b, dtype=torch.int64)
every = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(every, mine)
assert all(bool((e == mine).all()) for e in every)
carries_extra_sites = rank == world - 1
# too few sites for the ranks: refused identically everywhere, before any collective could hang
try:
    W.shard_bounds(100, 4)
    refused = False
except ValueError:
    refused = True
assert refused
# the site counts of the BASELINE configs, split `world` ways (what each rank of the driver's scaling runs derives:
# C2 weak = world x 1,000,000; C4 = 8,000,000; C5 = 500,000; C3 = 200,000): whole, contiguous, nobody empty, every
# inner boundary on a multiple of 256, the same everywhere -- and a split that would leave a rank empty is refused
# on EVERY rank (a rank that went on alone would hang the others in the first collective)
for t in (world * 1_000_000, 8_000_000, 500_000, 200_000, 256 * world):
    bb = W.shard_bounds(t, world)
    assert bb[0] == 0 and bb[-1] == t and len(bb) == world + 1 and all(x < y for x, y in zip(bb, bb[1:])), (t, bb)
    assert all(x % 256 == 0 for x in bb[1:-1])
    assert max(y - x for x, y in zip(bb, bb[1:])) - min(y - x for x, y in zip(bb, bb[1:])) <= 256 * world, (t, bb)
    tb = torch.tensor(bb, dtype=torch.int64)
    lo_, hi_ = tb.clone(), tb.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN); dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    assert torch.equal(lo_, hi_)
verdicts = []
for t in (256 * (world - 1), 256 * world + 1, world - 1):
    try:
        W.shard_bounds(t, world)
        verdicts.append(0)
    except ValueError:
        verdicts.append(1)
v = torch.tensor(verdicts, dtype=torch.int64)
vmin = v.clone()
dist.all_reduce(vmin, op=dist.ReduceOp.MIN)
assert torch.equal(v, vmin) and (world == 1 or verdicts[-1] == 1), verdicts
# "everybody has a value" (bench.py's reference check: a rank without the reference build must make ALL ranks skip
# the comparison, not sum a partial value): the agreement is an all-reduce of a flag, MIN
for someone_lacks_it in (False, True):
    have = 0 if (someone_lacks_it and rank == world - 1) else 1
    flag = torch.tensor([have], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    assert int(flag.item()) == (0 if someone_lacks_it else 1)

# (2) the PRODUCT's planner (pllhip_fused_plan_dry: pure host logic) on the op list every rank is
# handed: the same order, slots and reloads on every rank -- the ranks then launch the same kernels
plan = W.random_tree(40, seed=5)
ops = np.ascontiguousarray(plan.ops)
n = len(ops)
order = (C.c_uint * n)(); slots = (C.c_int * (6 * n))(); hbm = C.c_uint()
rc = amd.lib.pllhip_fused_plan_dry(C.c_uint(40), C.c_uint(38), C.c_uint(38), C.c_int(1), ops.ctypes.data_as(C.c_void_p),
                                   C.c_uint(n), C.c_uint(6), order, C.byref(hbm), slots)
assert rc == 0, amd.errmsg()
sig = torch.tensor(list(order) + list(slots) + [hbm.value], dtype=torch.int64)
sigs = [torch.zeros_like(sig) for _ in range(world)]
dist.all_gather(sigs, sig)
assert all(bool((x == sig).all()) for x in sigs)
assert sorted(order) == list(range(n))

# (3) the rest of the host pipeline in front of the path, per rank on its own columns: pattern
# compression of the rank's slice (weights sum to the slice's length), the tree builders' op list
seqs = W.random_alignment(40, total, 4, seed=9)
lo, hi = b[rank], b[rank + 1]
rows = [s[lo:hi] for s in seqs]
arr = (C.c_char_p * 40)(*rows)
length = C.c_int(hi - lo)
amd.lib.pll_compress_site_patterns.restype = C.POINTER(C.c_uint)
w = amd.lib.pll_compress_site_patterns(arr, (C.c_uint * 256).in_dll(amd.lib, "pll_map_nt"), 40, C.byref(length))
assert bool(w), amd.errmsg()
weights = np.ctypeslib.as_array(w, shape=(length.value,))
assert int(weights.sum()) == hi - lo and 0 < length.value <= hi - lo
counts = torch.tensor([hi - lo, int(weights.sum())], dtype=torch.int64)
dist.all_reduce(counts, op=dist.ReduceOp.SUM)
assert counts.tolist() == [total, total]
if rank == 0:
    print("HOST_OK bounds=%s reloads=%d patterns(rank 0)=%d" % (b, hbm.value, length.value))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("nproc", [2, 8], ids=["two-ranks", "eight-ranks"])
def test_product_host_logic_gloo(tmp_path, amd, nproc):
    """What each rank of the one-process-per-GPU mode does on the HOST, with the product library (no
    device needed): its site range -- also at the BASELINE configs' site counts split 2 and 8 ways --, the
    op-list planner's dry run (identical plans on all ranks), pattern compression of its own columns."""
    script = tmp_path / "host_worker.py"
    script.write_text(HOST_WORKER)
    assert "HOST_OK" in launch(script, nproc)


# ---------------------------------------------------------------- bench.py's alignment: one, indexed by global site

ALIGN_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from libpll_amd import workload as W
from libpll_amd.pllapi import PllLibrary, ATTRIB_PATTERN_TIP, ATTRIB_ARCH_AVX2

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ref = PllLibrary(os.path.join(sys.argv[1], "oracle", "_ref", "libpll_ref.so"))
plan = W.balanced_tree(16, seed=42)
cat = ref.compute_gamma_cats(W.GAMMA_ALPHA, 4)
total, block = int(sys.argv[2]), 700      # (ranges that begin and end inside blocks)
b = W.shard_bounds(total, world)
lo, hi = b[rank], b[rank + 1]
mine = W.global_alignment(plan, lo, hi, W.GTR_RATES, W.GTR_FREQS, cat, seed=42, block=block)
whole = W.global_alignment(plan, 0, total, W.GTR_RATES, W.GTR_FREQS, cat, seed=42, block=block)
# a rank's range IS a slice of the alignment a one-GPU run evaluates whole
assert all(w[lo:hi] == m for w, m in zip(whole, mine))
assert len(set(whole[0][i * block:(i + 1) * block] for i in range(total // block))) > 1, "blocks differ"
# ... with a cycle of distinct blocks (BASELINE config 4's generator) as well
rep = W.global_alignment(plan, lo, hi, W.GTR_RATES, W.GTR_FREQS, cat, seed=7, block=block, distinct=3)
rep_whole = W.global_alignment(plan, 0, total, W.GTR_RATES, W.GTR_FREQS, cat, seed=7, block=block, distinct=3)
assert all(w[lo:hi] == m for w, m in zip(rep_whole, rep))
assert rep_whole[0][0:block] == rep_whole[0][3 * block:4 * block]
# the sum over the ranks of the reference's lnL of each range == the reference's lnL of the whole: what
# bench.py reports as lnl_rel_err_vs_reference at N > 1 compares the product's all-reduced lnL with exactly this sum
attrs = ATTRIB_PATTERN_TIP | ATTRIB_ARCH_AVX2
part, n = W.reference_lnl(ref, plan, mine, 4, 4, attrs, chunk=900)
assert n == hi - lo
t = torch.tensor([part], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
full, _ = W.reference_lnl(ref, plan, whole, 4, 4, attrs, chunk=total)
assert abs(t.item() - full) <= 1e-13 * abs(full), (t.item(), full)
# a time-boxed prefix is whole chunks, at least one
pre, n_pre = W.reference_lnl(ref, plan, whole, 4, 4, attrs, chunk=900, budget_s=0.0)
assert n_pre == 900
if rank == 0:
    print("ALIGN_OK bounds=%s lnl=%.6f" % (b, full))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("nproc", [2, 8], ids=["two-ranks", "eight-ranks"])
def test_bench_alignment_is_one_alignment_gloo(tmp_path, ref, nproc):
    """bench.py at N > 1: every rank makes its own columns of ONE alignment
    (W.global_alignment), so the N-GPU lnL has something to be compared with; the per-rank reference
    values sum to the reference's value of the whole.  With 2 ranks and with the 8 the driver's scaling bench starts."""
    script = tmp_path / "align_worker.py"
    script.write_text(ALIGN_WORKER)
    # (5,000 sites split two ways; eight ranges on multiples of 256 sites need more than 8 x 512)
    assert "ALIGN_OK" in launch(script, nproc, 5000 if nproc == 2 else 12000)


# ---------------------------------------------------------------- the product, one rank per GPU

PRODUCT_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
import ctypes
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_AB_LEWIS
from oracle_api import Oracle
from helpers import make_case, build_partition, oracle_run, rel_err
amd = libpll_amd.load()
amd.lib.pll_amd_set_device(local)
orc = Oracle(os.path.join(sys.argv[1], "oracle", "liboracle.so"))

def sliced(case, lo, hi):
    c = dict(case)
    c["seqs"] = [s[lo:hi] for s in case["seqs"]]
    c["pw"] = case["pw"][lo:hi]
    c["sites"] = hi - lo
    return c

def leg(p, plan, R):
    p.update_partials(plan.ops)
    e = plan.root_edge
    lnl = p.compute_edge_loglikelihood(*e, [0] * R)
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
    d, dd = p.compute_likelihood_derivatives(e[1], e[3], 0.1, [0] * R, st)
    return np.array([lnl, d, dd])

for states, sites in ((4, 5000), (20, 1500)):
    case = make_case(states, "random", 14, sites, seed=31 + states)
    plan, R = case["plan"], case["rate_cats"]
    # the unsharded product and the oracle, on every rank (no communicator yet)
    whole = build_partition(amd, case, ATTRIB_PATTERN_TIP)
    o = oracle_run(orc, amd, whole, case, ATTRIB_PATTERN_TIP)
    full = leg(whole, plan, R)
    o.update_partials()
    e = plan.root_edge
    want = np.array([o.edge_loglikelihood(*e), *o.derivatives(o.sumtable(e[0], e[2], e[1], e[3]), 0.1)])
    whole.destroy()
    assert rel_err(full, want) < 1e-10, (full, want)
    # this rank's range as a partition of its own, summed over the ranks by RCCL inside the library
    b = W.shard_bounds(sites, world, granule=256)
    mine = build_partition(amd, sliced(case, b[rank], b[rank + 1]), ATTRIB_PATTERN_TIP)
    uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
    if rank == 0:
        buf = ctypes.create_string_buffer(128)
        assert amd.lib.pll_amd_comm_unique_id(buf)
        uid.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
    dist.broadcast(uid, src=0)
    mine.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))
    got = leg(mine, plan, R)
    assert rel_err(got, full) < 1e-12, (rank, got, full)
    # every rank holds the SAME sum (all-reduce, not reduce)
    t = torch.tensor(got, dtype=torch.float64, device="cuda")
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi)
    mine.destroy()
if rank == 0:
    print("PRODUCT_SHARDED_OK world=%d" % world)
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_product_site_sharding_over_rccl(tmp_path):
    """The product's one-process-per-GPU path with more than one rank: every rank builds a
    partition for its site range, pll_amd_comm_init joins them, and lnL / d / dd returned by
    the ordinary API calls are the RCCL sum -- equal to the unsharded product (1e-12) and the
    oracle (1e-10) on every rank.  Needs two devices; the ranks are fresh child processes
    (started before this process has touched a GPU through torch)."""
    import torch
    n = torch.cuda.device_count()      # (does not initialise the GPU)
    if n < 2:
        pytest.skip("one device visible: the multi-rank RCCL path needs two")
    script = tmp_path / "product_worker.py"
    script.write_text(PRODUCT_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PLLHIP_AA_EXACT", None)   # (the default path: 20-state kernels on the matrix cores)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node=%d" % min(n, 4), "--master-addr", "127.0.0.1", "--master-port",
                          str(free_port()), str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "PRODUCT_SHARDED_OK" in out.stdout
