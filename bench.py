#!/usr/bin/env python3
"""Headline benchmark: check-ins/sec of one MobGT train step (+ attention-kernel HBM GB/s vs roofline).

    python bench.py --gpus N --steps K --warmup W [--workload fsq|gow|big]

With N > 1 and no torch.distributed environment the script launches itself as N ranks (one process per GPU,
`python -m torch.distributed.run --nproc-per-node N`); it can equally be started under torch.distributed.run by the caller.

Workloads (mobgt_amd/workloads.py): `fsq` (default; BASELINE.json configs[1], the configuration the metric
is quoted on), `gow` (configs[2]), `big` (configs[4], per-GPU slice).  A "step" = forward + GradientTailLoss + backward
(+ gradient all-reduce) + AdamW on one pre-collated batch resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["fsq", "gow", "big"], default="fsq")
    ap.add_argument("--variant", choices=["fq", "stock"], default="fq",
                    help="fq: model_fqandtoyo.Graphormer (what entry.py trains; the headline); stock: graphormer/model.py's "
                         "pre-LN Graphormer (C = 128, d = 16) on the same trajectories")
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--n-batches", type=int, default=None, help="distinct pre-collated batches cycled through (default 8; big: 2)")
    ap.add_argument("--pois", type=int, default=None, help="override the workload's POI count")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--gemm-dtype", choices=["bf16", "f32"], default="bf16",
                    help="dtype of the encoder layers' GEMM-facing activations; attention MFMA operands, the bias "
                         "and the GCN adjacency product follow --dtype")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="op-by-op encoder layers (torch ops + HIP attention)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one all-reduce after the whole backward instead of two overlapped buckets")
    ap.add_argument("--ddp-form", default="auto", choices=["auto", "default"],
                    help="N > 1: 'auto' times the one-graph step and the overlapped forms for 20 steps on this job's ranks and takes the "
                         "fastest (train.choose_ddp_form); 'default': the one-graph step (or what MOBGT_DDP_OVERLAP / MOBGT_DDP_PARTS say)")
    ap.add_argument("--grad-comm", choices=["fp32", "bf16"], default="fp32",
                    help="N > 1: dtype the gradient buckets are all-reduced in (bf16 halves the bytes; DESIGN 5)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N = 1 only: a process group of ONE rank over RCCL, the step in its data-parallel form (layer-wise buckets, "
                         "all-reduces on RCCL's stream beside the replays): what the multi-rank structure costs without a wire")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the eval-step comparison with the oracle")
    ap.add_argument("--no-stress", action="store_true", help="skip the c5-shape attention roofline measurements")
    ap.add_argument("--no-loop", action="store_true", help="skip the fresh-batch-every-step measurement (value_with_collate)")
    ap.add_argument("--no-tail", action="store_true", help="--workload gow: skip the 814-node tail batch (roofline_gow_tail)")
    ap.add_argument("--loop-steps", type=int, default=300, help="timed steps of the fresh-batch loop")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not start rocprofv3 child passes for roofline.traffic / mfma_busy_pct (fall back to profiles/attn_pmc.json)")
    ap.add_argument("--no-gemm-autotune", action="store_true", help="leave hipBLASLt's default algorithm choice (no TunableOp)")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--no-sub", action="store_true",
                    help="headline run only: do not append the secondary workloads (gow, big, stock) as child runs")
    ap.add_argument("--sub-steps", type=int, default=100, help="timed steps of the gow / stock child runs (big: a tenth)")
    ap.add_argument("--seed", type=int, default=1)
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a torch.distributed environment: become the launcher of N ranks.  Nothing in
    this process has touched the GPU yet (torch is not even imported), so starting children is safe on this pool."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


ARGS = parse() if __name__ == "__main__" else argparse.Namespace(gpus=1, no_gemm_autotune=True, no_live_pmc=True)
if ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ and __name__ == "__main__":
    raise SystemExit(self_launch(ARGS))

# The train step's remaining library GEMMs (GCN / embedding / head: tall-skinny, a few hundred rows) are sensitive to
# hipBLASLt's algorithm choice: let PyTorch's TunableOp pick per shape during the eager warm-up passes that precede
# graph capture (<= 30 ms of trials per new shape).  Must be set before torch is imported.
if not ARGS.no_gemm_autotune:
    import tempfile
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "30")
    os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME",
                          os.path.join(tempfile.gettempdir(), "mobgt_tunableop_%d_pid" + str(os.getpid()) + ".csv"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


# ------------------------------------------------------------------------------------ attention roofline
def attn_fwd_bytes(G, T, C, H, s_x, s_b):
    """read Q,K,V + write O, read bias, write log-sum-exp."""
    return G * (4 * T * C * s_x + H * T * T * s_b + H * T * 4)


def attn_bwd_bytes(G, T, C, H, s_x, s_b, s_g, one_pass=False):
    """DESIGN §3.1.  Two passes: dQ pass reads Q,K,V,O,dO + writes dQ (6 T C), reads the bias, writes dBias; dK/dV pass reads
    Q,K,V,O,dO + writes dK,dV (7 T C) and reads the transposed bias; both read LSE.  ONE pass (round 4, T > 64, bf16): Q,K,V,O,dO
    read and dQ,dK,dV written once (8 T C), the transposed bias read once, dBias written once, LSE + rowsum(dO O) per row."""
    if one_pass:
        return G * (8 * T * C * s_x + H * T * T * (s_b + s_g) + H * T * 8)
    dq = G * (6 * T * C * s_x + H * T * T * (s_b + s_g) + H * T * 8)
    dkv = G * (7 * T * C * s_x + H * T * T * s_b + H * T * 8)
    return dq + dkv


LAST_FIRST_REPLAY_US = [None]        # per-launch time of the last _graph_time call's FIRST replay (the device coming out of idle)


def _graph_time(fn, reps):
    """Average duration (s) of fn() at SUSTAINED clocks: `reps` launches captured in a hipGraph, HIP events around one replay on
    the stream the graph runs on (the events bracket back-to-back kernels, not the Python/ctypes launch path).  The graph is
    replayed until 60 ms of continuous execution (at most 40 replays) lie behind the device, then five more replays are timed
    and the MEDIAN is returned: out of idle the c5 kernels read 156 -> 146 -> 142 -> 138 -> 136 -> 135.5 -> 135.8 ... us per
    launch over successive 7 ms replays (backward; forward 59 -> 54) -- the clock governor's ramp, which a training job's
    back-to-back steps never see (the same kernels inside the S-BIG step under rocprofv3: 55.0 / 129.3 + 7.4 us).  The first
    replay's figure is kept in LAST_FIRST_REPLAY_US and reported beside the sustained one."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()

    def once():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    first = once()
    LAST_FIRST_REPLAY_US[0] = first * 1e3 / reps
    spent, n = first, 1
    while spent < 60.0 and n < 40:
        spent += once()
        n += 1
    times = sorted(once() for _ in range(5))
    return times[2] / 1e3 / reps


ROTATE_MIN_SET_BYTES = 32 << 20       # input sets at least this large are rotated (smaller ones are cache-resident in the step too)
ROTATE_TOTAL_BYTES = 768 << 20        # ... over enough distinct sets that a set's lines have left the 256 MiB Infinity Cache


LAST_ATTN_FIRST_US = {"fwd": None, "bwd": None}     # time_attention's first-replay figures (see _graph_time)


def time_attention(G, H, T, d, io_dtype, bias_dtype, reps=50, p_drop=0.1, backward=False):
    """(forward s, backward s or None, number of input sets) per launch of mobgt_attn_bias_fwd / mobgt_attn_bias_bwd (dQ +
    dK/dV passes) in the TRAINING instantiation (attention dropout p_drop, bf16 dBias slices when the bias is bf16).

    Large shapes (c5: 177 MB read per forward launch) ROTATE over distinct Q/K/V/bias sets so that no launch finds its
    inputs in the 256 MiB Infinity Cache -- inside the S-BIG step a layer's bias was last read ~800 us and > 500 MB of
    traffic earlier.  Round 2 replayed ONE set back to back: a warm-cache figure (55 us) that the step's own rocprof
    numbers (65 us) did not reproduce.  Small shapes (S-FSQ: ~1 MB) keep one set: in the step they
    are cache-resident as well (the bias is re-read by six layers within 0.3 ms)."""
    from mobgt_amd import ops
    C = H * d
    s_x = 2 if io_dtype == torch.bfloat16 else 4
    s_b = 2 if bias_dtype == torch.bfloat16 else 4
    set_bytes = attn_fwd_bytes(G, T, C, H, s_x, s_b)
    n_sets = 1 if set_bytes < ROTATE_MIN_SET_BYTES else max(3, -(-ROTATE_TOTAL_BYTES // set_bytes))
    g = torch.Generator(device="cpu").manual_seed(0)
    sets = []
    for _ in range(n_sets):
        qkv = torch.randn(G, T, 3 * C, generator=g).to("cuda").to(io_dtype)
        bias = torch.randn(G, H, T, T, generator=g).to("cuda")
        pack = ops.pack_bias(bias, G, H, T, dtype=bias_dtype)
        del bias
        sets.append(dict(qkv=qkv, pack=pack, q=qkv[..., :C], k=qkv[..., C:2 * C], v=qkv[..., 2 * C:]))
    reps = -(-reps // n_sets) * n_sets
    it = [0]

    def fwd():
        s = sets[it[0] % n_sets]
        it[0] += 1
        ops._attn_fwd(s["q"], s["k"], s["v"], s["pack"], d ** -0.5, p_drop, 1, None)
    t_f = _graph_time(fwd, reps)
    LAST_ATTN_FIRST_US["fwd"] = LAST_FIRST_REPLAY_US[0]
    t_b = None
    if backward:
        for s in sets:
            s["out"], s["lse"] = ops._attn_fwd(s["q"], s["k"], s["v"], s["pack"], d ** -0.5, p_drop, 1, None)
            s["dout"] = torch.randn(G, T, C, generator=g).to("cuda").to(io_dtype)
            s["dqkv"] = torch.empty_like(s["qkv"])
            s["pack"].needs_grad, s["pack"].n_use = True, 1
            s["pack"].grad_buffer()
        it[0] = 0

        def bwd():
            s = sets[it[0] % n_sets]
            it[0] += 1
            s["pack"].n_bwd = 0
            dqkv = s["dqkv"]
            ops._attn_bwd(s["q"], s["k"], s["v"], s["out"], s["lse"], s["dout"], dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:],
                          s["pack"], d ** -0.5, p_drop, 1, None)
        t_b = _graph_time(bwd, reps)
        LAST_ATTN_FIRST_US["bwd"] = LAST_FIRST_REPLAY_US[0]
    return t_f, t_b, n_sets


# ---------------------------------------------------------------------------------------- live PMC counters
def live_pmc(shapes, p_drop, keep_dir=None, timeout=300):
    """HBM traffic and MFMA-busy % of the attention kernels at `shapes` = [(G, T, d), ...] (distinct d; bf16), measured NOW:
    three child runs of `rocprofv3 --pmc <set> --kernel-trace -- python3 tools/attn_bwd_bench.py` (counters in their own
    passes, never with --stats / tracing domains: MI355X_MICROARCH.md "HBM" and "rocprofv3 PMC slots").  FETCH_SIZE /
    WRITE_SIZE are KB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests of 16-B/lane streaming reads at 64 B),
    WRITE_SIZE is exact.  MFMA-busy % = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles =
    GRBM_GUI_ACTIVE / 8 (rocprofv3 sums GRBM over the 8 XCDs).  -> {d: {"fwd" | "dq" | "dkv" | "both": {...}}}, or None
    when rocprofv3 is missing or a pass fails (the bench line then falls back to profiles/attn_pmc.json and says so)."""
    import csv
    import glob
    import re
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None or os.environ.get("MOBGT_NO_LIVE_PMC") == "1":
        return None
    tool = os.path.join(ROOT, "tools", "attn_bwd_bench.py")
    passes = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"],
              "sq": ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU",
                     "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]}
    acc = {}
    tmp = tempfile.mkdtemp(prefix="mobgt_pmc_")
    env = dict(os.environ, REPS="4", P=str(p_drop), SHAPES=",".join("%d:%d:%d" % s for s in shapes), TMPDIR=tmp)
    for k in list(env):
        if k.startswith("PYTORCH_TUNABLEOP"):
            del env[k]
    pat = re.compile(r"attn_(fwd|bwd_dq|bwd_dkv|bwd_both|bwd_one)_kernel(?:<|ILi)(\d+)")
    try:
        for name, counters in passes.items():
            out = os.path.join(tmp, name)
            cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "r", "--",
                                                       sys.executable, tool]
            r = subprocess.run(cmd, env=env, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
            files = glob.glob(os.path.join(out, "**", "r_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            if keep_dir:
                os.makedirs(keep_dir, exist_ok=True)
                shutil.copy(files[0], os.path.join(keep_dir, f"attn_pmc_{name}.csv"))
            for row in csv.DictReader(open(files[0])):
                m = pat.search(row["Kernel_Name"])
                if m:
                    kern = {"fwd": "fwd", "bwd_dq": "dq", "bwd_dkv": "dkv", "bwd_both": "both", "bwd_one": "one"}[m.group(1)]
                    acc.setdefault(int(m.group(2)), {}).setdefault(kern, {}).setdefault(row["Counter_Name"], []).append(
                        float(row["Counter_Value"]))
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for dd, kerns in acc.items():
        for kern, cs in kerns.items():
            m = {c: sum(v[1:]) / max(len(v) - 1, 1) for c, v in cs.items()}           # first (cold) launch skipped
            if not all(c in m for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")):
                return None
            cyc = max(m["GRBM_GUI_ACTIVE"] / 8.0, 1.0)
            res.setdefault(dd, {})[kern] = dict(
                traffic_bytes=int(m["FETCH_SIZE"] * 2048 + m["WRITE_SIZE"] * 1024),
                fetch_bytes_corrected=int(m["FETCH_SIZE"] * 2048), write_bytes=int(m["WRITE_SIZE"] * 1024),
                mfma_busy_pct=round(100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 2),
                valu_busy_pct=round(100.0 * m.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (1024.0 * cyc), 2),
                wait_any_frac=round(m.get("SQ_WAIT_ANY", 0.0) / max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3))
    return res or None


def live_chain_pmc(rows, keep_dir=None, timeout=180):
    """HBM traffic per launch of the chain forward (cluster form where the row count allows it) at `rows` rows, measured NOW
    like the attention counters: two child runs of `rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3
    tools/chain_pmc.py rows 20` (stand-alone launches, each behind a 64 MB filler: cold L2, as in the step where a layer's
    weights were last read ~0.5 ms earlier).  -> dict or None."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None or os.environ.get("MOBGT_NO_LIVE_PMC") == "1":
        return None
    tool = os.path.join(ROOT, "tools", "chain_pmc.py")
    tmp = tempfile.mkdtemp(prefix="mobgt_cpmc_")
    env = dict(os.environ, TMPDIR=tmp)
    for k in list(env):
        if k.startswith("PYTORCH_TUNABLEOP"):
            del env[k]
    vals = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, c)
            cmd = ["rocprofv3", "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "r", "--",
                   sys.executable, tool, str(rows), "20"]
            r = subprocess.run(cmd, env=env, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
            files = glob.glob(os.path.join(out, "**", "r_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            v = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                 if row["Counter_Name"] == c and "layer_chain_fwd" in row["Kernel_Name"]]
            if len(v) < 2:
                return None
            vals[c] = sum(v[1:]) / (len(v) - 1)
            if keep_dir:
                os.makedirs(keep_dir, exist_ok=True)
                with open(os.path.join(keep_dir, f"chain_pmc_{c}.csv"), "w") as f:
                    for line in open(files[0]):
                        if "Counter_Name" in line or "layer_chain" in line:
                            f.write(line)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = int(vals["FETCH_SIZE"] * 2048), int(vals["WRITE_SIZE"] * 1024)      # KB; FETCH_SIZE doubled on gfx950
    return dict(rows=rows, traffic_bytes=fetch + write, fetch_bytes_corrected=fetch, write_bytes=write)


# ------------------------------------------------------------------------- secondary workloads as child runs
def run_sub_workloads(args):
    """The other workloads this repo quotes numbers for, under the SAME clock as the headline:
    `python bench.py --workload gow`, `--variant stock`, `--workload big` as child processes of this run (each builds its
    model, captures its graphs, times its steps and checks itself against the oracle where the oracle can run), plus one
    `rocprofv3 --kernel-trace` child pass of the S-BIG step for the IN-STEP durations of the attention kernels (the one
    bias all 12 layers share is partly Infinity-Cache-resident there, unlike the all-cold roofline_stress figures)."""
    import csv
    import glob
    import shutil
    import tempfile
    me = os.path.abspath(__file__)
    common = ["--gpus", "1", "--no-sub", "--no-cpu-baseline", "--no-stress", "--no-loop", "--no-live-pmc", "--seed", str(args.seed)]
    specs = {"gow": ["--workload", "gow", "--steps", str(args.sub_steps), "--warmup", "10"],
             "stock": ["--variant", "stock", "--steps", str(args.sub_steps), "--warmup", "10"],
             "big": ["--workload", "big", "--steps", str(max(args.sub_steps // 10, 5)), "--warmup", "3"]}
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("PYTORCH_TUNABLEOP"):          # (each child picks its own file name)
            del env[k]
    keep = ("value", "ms_per_step", "steps", "warmup", "final_loss", "long_run", "parity", "ms_per_step_chunks", "host_stalls")
    res = {}
    for name, extra in specs.items():
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, me] + common + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               timeout=600, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                res[name] = dict(error=f"rc {r.returncode}: {r.stderr[-300:]}")
                continue
            j = json.loads(line[-1])
            e = {k: j.get(k) for k in keep}
            e["workload"] = j["config"]["workload"]
            e["padded_nodes_per_batch"] = j["config"]["padded_nodes_per_batch"]
            if j.get("roofline_gow_tail"):
                e["tail"] = j["roofline_gow_tail"]
            if j.get("roofline"):
                e["attn_fwd_at_timed_shapes"] = {k: j["roofline"].get(k) for k in ("achieved", "frac", "avg_launch_us", "bytes_per_launch")}
            if name == "big":
                e["parity"] = ("no oracle at P = 100 000 (it would need the dense 100 000^2 adjacency); the sparse path is pinned against "
                               "the oracle on the same universe densified at P = 1 500 (tests/test_gpu_sparse.py) and the 12-layer "
                               "stack / bias / attention at this size in tests/test_gpu_c5.py")
            e["child_wall_s"] = round(time.perf_counter() - t0, 1)
            res[name] = e
        except Exception as ex:
            res[name] = dict(error=repr(ex))
    # in-step kernel durations of the S-BIG step: one kernel-trace pass (no counters, no --stats)
    if shutil.which("rocprofv3") is not None and "error" not in res.get("big", {"error": 1}):
        tmp = tempfile.mkdtemp(prefix="mobgt_bigtrace_")
        try:
            cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, "o"), "-o", "r", "--",
                   sys.executable, me] + common + ["--workload", "big", "--steps", "6", "--warmup", "2", "--no-parity"]
            r = subprocess.run(cmd, env=dict(env, TMPDIR=tmp), cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            files = glob.glob(os.path.join(tmp, "o", "**", "r_kernel_trace.csv"), recursive=True)
            if r.returncode == 0 and files:
                rows = list(csv.DictReader(open(files[0])))
                rows.sort(key=lambda x: int(x["Start_Timestamp"]))
                idx = [i for i, x in enumerate(rows) if "step_prologue_kernel" in x["Kernel_Name"]]
                seg = rows[idx[-3]:idx[-2]]                              # one replayed step (third from the end)
                wall = (int(rows[idx[-2]]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
                short = {"attn_fwd_kernel": "attn_fwd_us", "attn_bwd_dq_kernel": "attn_bwd_dq_us", "attn_bwd_dkv_kernel": "attn_bwd_dkv_us",
                         "attn_bwd_one_kernel": "attn_bwd_one_us", "build_bias_kernel": "build_bias_us", "build_bias_bwd": "build_bias_bwd_us"}
                durs = {}
                for x in seg:
                    for pat, key in short.items():
                        if pat in x["Kernel_Name"]:
                            durs.setdefault(key, []).append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
                            break
                b5f = attn_fwd_bytes(16, 785, 256, 8, 2, 2)
                b5b = attn_bwd_bytes(16, 785, 256, 8, 2, 2, 2, one_pass="attn_bwd_one_us" in durs)
                for pat, key in (("attn_dq_finish_kernel", "attn_dq_finish_us"),):
                    v = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in seg if pat in x["Kernel_Name"]]
                    if v:
                        durs[key] = v
                ins = dict(step_wall_us_under_profiler=round(wall, 1), kernels_per_step=len(seg),
                           **{k: round(sum(v) / len(v), 2) for k, v in durs.items()},
                           launches={k: len(v) for k, v in durs.items()},
                           source="rocprofv3 --kernel-trace child pass of this run (6 replayed steps, the third from the end)")
                if "attn_fwd_us" in ins:
                    ins["attn_fwd_frac_of_8TBs"] = round(b5f / (ins["attn_fwd_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 3)
                tb = sum(ins.get(k, 0.0) for k in ("attn_bwd_dq_us", "attn_bwd_dkv_us", "attn_bwd_one_us", "attn_dq_finish_us"))
                if tb > 0:
                    ins["attn_bwd_frac_of_8TBs"] = round(b5b / (tb * 1e-6) / 1e9 / HBM_PEAK_GBS, 3)
                res["big"]["in_step"] = ins
                keep_dir = os.path.join(ROOT, "gpurun_out", "sub_big")
                os.makedirs(keep_dir, exist_ok=True)
                with open(os.path.join(keep_dir, "big_step_summary.txt"), "w") as f:
                    f.write(json.dumps(ins, indent=1) + "\n")
            else:
                res["big"]["in_step"] = dict(error=f"rc {r.returncode}")
        except Exception as ex:
            res["big"]["in_step"] = dict(error=repr(ex))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return res


# --------------------------------------------------------------------------------- the Gowalla tail
def time_gow_tail(model, coll, uni, ts, args, w):
    """BASELINE configs[2] names "longer SPD paths" and BASELINE.md section 4 row 2 prices its long-N bucket (T 815, C 192, d 24:
    attention forward <= 60 us): the timed S-GOW pool reaches 141 padded nodes, the data's ONE 814-node trajectory does not occur
    in eight batches drawn from the histogram.  Here, under this run's clock: a batch of 16 trajectories that holds an 814-node
    one (the other 15 drawn from the histogram) as a 9th batch of the SAME trainer -- its step time --, the bias assembly and its
    backward on that batch, and the attention kernels at that shape with every input rotated through > 768 MB (all-cold)."""
    from mobgt_amd import synth, workloads
    ns = [814] + [int(v) for v in workloads.gowalla_node_counts(15, 4242)]
    trajs = synth.make_batch_of_trajectories(seed=4242, G=16, P=w["P"], n_user=w["n_user"], cat_of_poi=uni.cat_of_poi, n_nodes=ns)
    b = coll(trajs)
    G, N = b.x.shape[:2]
    T, H, L = N + 1, w["model"]["num_heads"], w["model"]["n_layers"]
    out = dict(workload=f"S-GOW tail batch: node counts {sorted(ns, reverse=True)[:4]} ..., padded T {T}, C 192, d 24, {L} layers")
    i = ts.add_batch(b)
    for _ in range(3):
        ts.step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 20
    for _ in range(k):
        ts.step(i)
    torch.cuda.synchronize()
    out["step_ms"] = (time.perf_counter() - t0) / k * 1e3
    out["check_ins_per_s_at_this_batch"] = G / (out["step_ms"] * 1e-3)
    # bias assembly forward / backward (six bf16 dBias slices summed into the tables' gradients) on this batch
    model.eval()
    with torch.no_grad():
        t_bias = _graph_time(lambda: model.assemble_bias(b), 10)
    model.train()
    pack = model.assemble_bias(b)
    pack.needs_grad, pack.n_use = True, L
    buf = pack.grad_buffer()
    buf.copy_(torch.randn(buf.shape[1:], device=buf.device).bfloat16().expand_as(buf))
    pack.n_bwd = L
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    pack.token.backward()
    ev[1].record()
    torch.cuda.synchronize()
    ld = pack.ld
    idx_b = b.rel_pos.numel() * b.rel_pos.element_size() + b.poi_pos.numel() * b.poi_pos.element_size() + b.edge_input.numel() * b.edge_input.element_size()
    by_f = 2 * G * H * T * ld * 2 + idx_b                       # both bias copies written + every index read once
    by_b = L * G * H * T * ld * 2 + idx_b                       # L dBias slices read + the indices
    out["build_bias"] = dict(us=t_bias * 1e6, bytes=by_f, achieved_GBs=by_f / t_bias / 1e9, frac=by_f / t_bias / 1e9 / HBM_PEAK_GBS)
    tb_ = ev[0].elapsed_time(ev[1]) * 1e-3
    out["build_bias_bwd"] = dict(us=tb_ * 1e6, bytes=by_b, achieved_GBs=by_b / tb_ / 1e9, frac=by_b / tb_ / 1e9 / HBM_PEAK_GBS,
                                 note="one eager call (hop-table backward and host launch gaps included): an upper bound")
    # the attention kernels at this shape, all-cold
    tf, tbw, nset = time_attention(16, H, 815, 24, torch.bfloat16, torch.bfloat16, reps=24, p_drop=0.1, backward=True)
    bf_ = attn_fwd_bytes(16, 815, 192, H, 2, 2)
    bb_ = attn_bwd_bytes(16, 815, 192, H, 2, 2, 2, one_pass=True)
    out["attn_fwd"] = dict(us=tf * 1e6, bytes=bf_, achieved_GBs=bf_ / tf / 1e9, frac=bf_ / tf / 1e9 / HBM_PEAK_GBS, input_sets_rotated=nset,
                           baseline_md_row2_target_us=60.0)
    out["attn_bwd"] = dict(us=tbw * 1e6, bytes=bb_, achieved_GBs=bb_ / tbw / 1e9, frac=bb_ / tbw / 1e9 / HBM_PEAK_GBS,
                           kernel="attn_bwd_one_kernel + attn_dq_finish_kernel")
    return out


# --------------------------------------------------------------------------------- fresh batch every step
def time_epoch_loop(model, coll, name, uni, args):
    """check-ins/s of train.EpochLoop over a pool of raw trajectories: epoch 0 warms up (captures one step graph per shape
    bucket that occurs), then whole epochs are timed until --loop-steps steps have run.  Every timed step packs 16 raw
    trajectories on the host, copies them to the device and replays [collate + forward + loss + backward + AdamW]."""
    from mobgt_amd import workloads
    from mobgt_amd.train import EpochLoop
    n_pool = 8 if name == "big" else 64
    pool = workloads.make_pool(name, n_pool, args.batch_size, uni, seed0=5000)
    dataset = [t for trajs in pool for t in trajs]
    loop = EpochLoop(model, coll, dataset, batch_size=args.batch_size, seed=args.seed, use_graph=True)
    loop.run_epoch(0)
    loop.run_epoch(1)                                  # (a second lap: buckets the first shuffle did not produce)
    torch.cuda.synchronize()
    graphs_before = len(loop.slots)
    steps, ep = 0, 2
    # The host has ~300 us of slack per step here (a round-4 probe: docs/NOTEBOOK.md): a generation-2 pass of Python's collector over a
    # large heap (tens of ms) would show up as +0.05-0.1 ms per step over a 0.2 s timed region.  What exists is moved out of the
    # collector's sight, as a long-running trainer would do once after start-up; the loop's own garbage is still collected.
    import gc
    gc.collect()
    gc.freeze()
    per_epoch = []
    t0 = time.perf_counter()
    while steps < args.loop_steps:
        te = time.perf_counter()
        n_ep = loop.run_epoch(ep)["steps"]
        steps += n_ep
        ep += 1
        per_epoch.append((time.perf_counter() - te) / max(n_ep, 1) * 1e3)          # (host-side: launches run ahead of the GPU)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    gc.unfreeze()
    loss = float(loop.ts.loss_out.item())
    if loss != loss:
        raise RuntimeError("loop diverged")
    # the same sequence of step graphs with NO fresh data (each bucket replays on the batch it saw last): what the GPU needs
    # for this mix of shapes -- the loop's batches are not the 8 pre-collated ones `value` is quoted on
    from mobgt_amd.data import bucket_nodes
    seq = []
    for e in range(2, ep):
        for ids in loop.batches_of_epoch(e):
            tr = [dataset[i] for i in ids]
            seq.append(loop.slots[(len(tr), bucket_nodes(max(len(t["node_name"]) for t in tr), loop.buckets))]["index"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in seq:
        loop.ts.step(i)
    torch.cuda.synchronize()
    el_replay = time.perf_counter() - t0
    return dict(value=args.batch_size * steps / el, unit="check-ins/s", ms_per_step=el / steps * 1e3, steps=steps,
                ms_per_step_same_graphs_no_input=el_replay / len(seq) * 1e3, ms_per_step_by_epoch_host_side=[round(v, 4) for v in per_epoch],
                dataset_trajectories=len(dataset), shape_buckets=sorted(k[1] for k in loop.slots),
                graphs_captured_inside_timed_region=len(loop.slots) - graphs_before, final_loss=loss,
                what="fresh batch every step: host pack of raw trajectories + H2D + device collate (SPD / edge paths / "
                     "degrees / distance bins; on the copy stream beside the previous step) + [forward + loss + backward + "
                     "AdamW] as one hipGraph per shape bucket; ms_per_step_same_graphs_no_input = the same graph sequence "
                     "replayed without new input")


# ------------------------------------------------------------------------------------------ CPU baseline
def chain_fwd_bytes(R, C, F):
    """Algorithmic HBM bytes of one mobgt_layer_chain_fwd launch: the layer's bf16 weights once (Wo, W1, W2, next Wqkv),
    a (bf16) and x (f32) in; x1, x2, out (f32), z, out_a, qkv (bf16), u, h (bf16) out."""
    weights = 2 * (C * C + 2 * F * C + 3 * C * C)
    return weights + R * (2 * C + 4 * C) + R * (3 * 4 * C + 2 * 2 * C + 2 * 3 * C + 2 * 2 * F)


def time_chain(R, C, F, reps=50, p_drop=0.1):
    """Seconds per launch of mobgt_layer_chain_fwd (csrc/chain.hip) on R rows, weights packed as the model packs them."""
    import ctypes
    from mobgt_amd import _lib
    from mobgt_amd.ops import _p, _stream
    lib = _lib.lib()
    bf = lambda *s: (torch.randn(*s, device="cuda") * 0.05).bfloat16()

    def pack(w):
        out = torch.empty_like(w)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        _lib.check(lib.mobgt_pack_mfma_b(1, (vp * 1)(w.data_ptr()), (vp * 1)(out.data_ptr()), (ci * 1)(w.shape[0]),
                                         (ci * 1)(w.shape[1]), None, _stream()), "mobgt_pack_mfma_b")
        return out
    a, x = bf(R, C), torch.randn(R, C, device="cuda")
    wo, w1, w2, wq = pack(bf(C, C)), pack(bf(F, C)), pack(bf(C, F)), pack(bf(3 * C, C))
    bo, b1, b2, bq = bf(C), bf(F), bf(C), bf(3 * C)
    ln = [torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")]
    x1, x2, out = (torch.empty(R, C, device="cuda") for _ in range(3))
    z, out_a, u, h, qkv = bf(R, C), bf(R, C), bf(R, F), bf(R, F), bf(R, 3 * C)
    st = torch.empty(4, R, device="cuda")
    from mobgt_amd.fused_layer import chain_workspace
    ws = chain_workspace(a.device, C, R)

    def fn():
        _lib.check(lib.mobgt_layer_chain_fwd(_p(a), _p(x), _p(wo), _p(bo), _p(ln[0]), _p(ln[1]), _p(w1), _p(b1), _p(w2), _p(b2),
                                             _p(ln[2]), _p(ln[3]), _p(wq), _p(bq), _p(x1), _p(z), _p(u), _p(h), _p(x2), _p(out),
                                             _p(out_a), _p(qkv), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), R, C, F, p_drop, 1, None,
                                             9, 10, _p(ws), _stream()), "mobgt_layer_chain_fwd")
    return _graph_time(fn, reps)


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                            # cgroup v2 CPU quota of the container, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def _cpu_batches(batches):
    from types import SimpleNamespace
    cb = []
    for b in batches:
        c = SimpleNamespace()
        for f in ("attn_bias", "rel_pos", "poi_pos", "edge_input", "x", "in_degree", "out_degree", "user", "y", "time_normal"):
            t = getattr(b, f).cpu()
            setattr(c, f, t.float() if t.dtype.is_floating_point else t.long())
        cb.append(c)
    return cb


def _oracle_consts(model, uni):
    """The oracle's own constants (fp32 (D+I)^-1 (A+I) etc., model_fqandtoyo.py:650-700 / 787-838) from the universe."""
    from oracle import model_oracle as mo
    return mo.fq_constants(uni, model.dataset_name, diag_inverse=True, num_bins=model.poi_pos_encoder.num_embeddings)


def cpu_baseline(model, batches, pools, uni, args, n_layers):
    """The oracle (CPU restatement of the reference, oracle/model_oracle.py) timed on the host cores for the same step
    definition on the same batches: forward + GradientTailLoss + backward + AdamW, train mode (kind "port").  Reported
    at 8 torch threads (the README's --num_workers 8 scale) and at the usable core count (capped at 32: torch CPU eager
    does not scale further on these small ops); `value` is the better of the two.  `with_collate`: the same plus the
    oracle's preprocess_item + collator on the raw trajectories (what the reference does per batch)."""
    from oracle import model_oracle as mo
    consts = _oracle_consts(model, uni)
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=2e-4, weight_decay=0.01)
    cb = _cpu_batches(batches)
    kw = dict(n_layers=n_layers, H=8, D=20, p=0.1, p_in=0.1, p_att=0.1, training=True, hidden=model.hidden_dim)
    G = len(cb[0].y)

    def one_step(i):
        b = cb[i % len(cb)]
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = mo.fq_training_loss(sd, b, consts, **kw)
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    cores = usable_cores()
    cands = sorted({min(8, cores), min(32, cores)})
    budget = args.cpu_seconds / len(cands)
    by_threads = {}
    for c in cands:
        torch.set_num_threads(c)
        one_step(0)                                  # warm-up at this thread count
        n, t_used = 0, 0.0
        while (n < 3 or t_used < budget) and n < 60:
            t_used += one_step(n)
            n += 1
        by_threads[c] = dict(value=G * n / t_used, steps=n, seconds=round(t_used, 2))
    best = max(by_threads, key=lambda c: by_threads[c]["value"])
    torch.set_num_threads(best)
    # including the reference-shaped data path: preprocess_item (Floyd-Warshall + edge paths, C restatement) + collator
    with_collate = None
    try:
        from oracle import collator_oracle as co
        from mobgt_amd import synth
        if uni.distance is not None:
            t0 = time.perf_counter()
            nb = min(2, len(pools))
            for trajs in pools[:nb]:
                items = [co.preprocess_item(synth.trajectory_to_item(t, idx=i)) for i, t in enumerate(trajs)]
                co.collator_poi(items, uni.distance, max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
            t_coll = (time.perf_counter() - t0) / nb
            t_step = G / by_threads[best]["value"]
            with_collate = dict(value=G / (t_step + t_coll), collate_s_per_batch=round(t_coll, 3), batches=nb)
    except Exception as e:                           # never lose the bench line over the auxiliary figure
        with_collate = dict(error=repr(e))
    total = sum(v["seconds"] for v in by_threads.values())
    return dict(value=by_threads[best]["value"], unit="check-ins/s", cores=best, kind="port",
                by_threads={str(k): v for k, v in by_threads.items()}, with_collate=with_collate,
                sample=f"{sum(v['steps'] for v in by_threads.values())} train steps (fwd+loss+bwd+AdamW, fp32, train mode) of the "
                       f"oracle cycling over the same pre-collated batches, at {' and '.join(str(c) for c in cands)} torch threads "
                       f"({cores} usable cores), after warm-up; {total:.1f} s of CPU work")


def cpu_baseline_stock(model, batches, args, n_layers):
    """--variant stock: the oracle's model.py restatement (oracle.graphormer_stock_forward) + cross_entropy + AdamW on the host."""
    from oracle import model_oracle as mo
    import torch.nn.functional as F
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=2e-4, weight_decay=0.01)
    cb = _cpu_batches(batches)
    G = len(cb[0].y)
    cores = usable_cores()
    torch.set_num_threads(min(8, cores))

    def one_step(i):
        b = cb[i % len(cb)]
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        logits = mo.graphormer_stock_forward(sd, b, n_layers, 8, 20, p=0.1, p_in=0.1, p_att=0.1, training=True)
        F.cross_entropy(logits, b.y.view(-1), ignore_index=0).backward()
        opt.step()
        return time.perf_counter() - t0
    one_step(0)
    n, used = 0, 0.0
    while (n < 3 or used < args.cpu_seconds) and n < 60:
        used += one_step(n)
        n += 1
    return dict(value=G * n / used, unit="check-ins/s", cores=min(8, cores), kind="port",
                sample=f"{n} train steps (fwd+cross_entropy+bwd+AdamW, fp32, train mode) of the oracle's model.py restatement "
                       f"on the same pre-collated batches at {min(8, cores)} torch threads; {used:.1f} s of CPU work")


def _parity_summary(per, what):
    """Batch 0's figures at the top level (the keys earlier rounds reported) + the worst case over ALL timed batches."""
    worst = max(per, key=lambda e: e["max_abs_logit_err"])
    out = dict(per[0])
    out.update(batches=len(per), worst_batch=per.index(worst), worst_max_abs_logit_err=worst["max_abs_logit_err"],
               worst_rel_loss_err=max(abs(e["loss_hip"] - e["loss_oracle"]) / max(abs(e["loss_oracle"]), 1e-30) for e in per),
               per_batch=[dict(loss_hip=e["loss_hip"], loss_oracle=e["loss_oracle"], max_abs_logit_err=e["max_abs_logit_err"]) for e in per],
               mode=what)
    return out


def oracle_parity_stock(model, batches, n_layers):
    """--variant stock: eval-mode logits and loss of the timed model vs oracle.graphormer_stock_forward on every timed batch."""
    from oracle import model_oracle as mo
    import torch.nn.functional as F
    was = model.training
    model.eval()
    try:
        per = []
        sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        for i, bt in enumerate(batches):
            with torch.no_grad():
                logits = model(bt).float().cpu()
                loss = float(model.training_step(bt, i))
                b = _cpu_batches([bt])[0]
                ref = mo.graphormer_stock_forward(sd, b, n_layers, 8, 20)
                ref_loss = float(F.cross_entropy(ref, b.y.view(-1), ignore_index=0))
            per.append(dict(loss_hip=loss, loss_oracle=ref_loss, max_abs_logit_err=float((logits - ref).abs().max()),
                            max_abs_logit=float(ref.abs().max())))
        return _parity_summary(per, "eval (dropout off), every timed batch, weights after the timed steps; top-level keys = batch 0")
    finally:
        model.train(was)


def oracle_parity(model, batches, uni, n_layers):
    """Eval-mode forward + training_step loss of the timed model vs the oracle (fp32, CPU) on EVERY timed batch."""
    from oracle import model_oracle as mo
    was = model.training
    model.eval()
    try:
        sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        consts = _oracle_consts(model, uni)
        per = []
        for i, bt in enumerate(batches):
            with torch.no_grad():
                logits = model(bt)[0].float().cpu()
                loss = float(model.training_step(bt, i))
                b = _cpu_batches([bt])[0]
                ref, _ = mo.graphormer_fq_forward(sd, b, consts, n_layers=n_layers, H=8, D=20, hidden=model.hidden_dim)
                ref_loss = float(mo.gradient_tail_loss(ref, b.y - 1, 0.2))
            per.append(dict(loss_hip=loss, loss_oracle=ref_loss, max_abs_logit_err=float((logits - ref).abs().max()),
                            max_abs_logit=float(ref.abs().max())))
        return _parity_summary(per, "eval (dropout off), every timed batch, weights after the timed steps; top-level keys = batch 0")
    finally:
        model.train(was)


def main():
    args = ARGS
    # ONE line on stdout, whatever else the process loads: RCCL prints a version banner (five lines) on stdout when its
    # communicator comes up (seen on the GPU box, round 4), TunableOp and the profiler children have their own chatter.  The real
    # stdout is kept aside for the JSON line; file descriptor 1 points at stderr from here on (C-level writers included).
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # MOBGT_TEST_SHARED_GPU=1: developer/test switch to exercise the multi-rank code path on a ONE-GPU box (all ranks on
    # cuda:0, gloo instead of RCCL).  Never set by the driver; numbers from such a run are meaningless.
    shared = os.environ.get("MOBGT_TEST_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_comm = bool(args.force_comm) and world == 1
    ddp = world > 1 or force_comm
    if force_comm:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MOBGT_FORCE_COMM="1")
        os.environ.setdefault("MASTER_PORT", str(port))
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from mobgt_amd.train import recommended_env
        for k_, v_ in recommended_env().items():
            os.environ.setdefault(k_, v_)
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from mobgt_amd import workloads
    from mobgt_amd.train import TrainStep, broadcast_parameters

    name = args.workload
    w = workloads.WORKLOADS[name]
    n_layers = w["model"]["n_layers"]
    bf16 = args.dtype == "bf16"
    n_batches = args.n_batches or (2 if name == "big" else 8)
    uni, model, coll = workloads.build(name, dev, seed=args.seed, dtype=args.dtype, gemm_dtype=args.gemm_dtype,
                                       fused=not args.unfused, P=args.pois, variant=args.variant)
    stock = args.variant == "stock"
    broadcast_parameters(model)
    # Length-balanced sharding (SURVEY §8e hazard): every rank draws the SAME pool of world x n_batches batches.  One rank: the
    # pool's batches as they are.  More ranks: the pool's trajectories are dealt by `data.balanced_batches` -- the dealing
    # `train.EpochLoop` uses -- so that at each synchronous step all ranks work on batches of the same or the neighbouring shape
    # bucket (per-rank work is still n_batches x 16 trajectories: weak scaling).
    raw = workloads.make_pool(name, n_batches * world, args.batch_size, uni)
    if world > 1:
        from mobgt_amd.data import balanced_batches
        flat = [t for trajs in raw for t in trajs]
        steps = balanced_batches([len(t["node_name"]) for t in flat], world, args.batch_size, epoch=0, seed=args.seed)
        mine = [[flat[i] for i in s[rank]] for s in steps[:n_batches]]
    else:
        mine = raw
    batches, shapes = [], []
    for trajs in mine:
        b = coll(trajs)
        batches.append(b)
        shapes.append((len(b), b.x.shape[1] + 1))
    torch.cuda.synchronize()

    ts_kw = dict(autocast_dtype=torch.bfloat16 if (bf16 and args.gemm_dtype == "bf16" and args.unfused) else None,
                 use_graph=not args.no_graph, overlap=not args.no_overlap, seed=args.seed,
                 grad_comm_dtype=torch.bfloat16 if args.grad_comm == "bf16" else None)
    ddp_form, ddp_form_ms = None, None
    if (ddp and world > 1 and args.ddp_form == "auto" and not args.no_graph and not args.no_overlap
            and os.environ.get("MOBGT_DDP_OVERLAP") is None and os.environ.get("MOBGT_DDP_HOST_EXCHANGE") is None):
        # (round 6) which form of the data-parallel step wins is a property of the node, not of this script: measured here, on
        # the ranks of this very job, agreed on across them
        from mobgt_amd.train import choose_ddp_form
        ddp_form, form_kw, ddp_form_ms = choose_ddp_form(model, batches, steps=20, warmup=5, **ts_kw)
        for k_, v_ in form_kw.pop("_env", {}).items():
            os.environ[k_] = v_
        ts_kw.update(form_kw)
    ts = TrainStep(model, batches, **ts_kw)
    ts.prepare()
    for i in range(args.warmup):
        ts.step(i)
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    # the timed region: EXACTLY args.steps steps; events after every fifth of them give the spread without a host sync
    n_chunks = 5 if args.steps >= 5 else 1
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_chunks + 1)]
    bounds = [round(k * args.steps / n_chunks) for k in range(n_chunks + 1)]
    # Host hygiene (round 4): right behind the synchronisation the host is not ahead of the GPU, so a pause of the Python
    # process lands in the measurement in full -- one final run of this round read 1.33 ms/step in its first fifth and 0.577 in
    # the other four (+30 ms somewhere in 40 steps).  Python's collector is taken out of the timed region (collected before,
    # switched off inside, as a long-running trainer does after start-up); the host's per-step gaps are recorded (one clock
    # read per step) and reported as `host_stalls`, so that a pause that still happens (a neighbour on the box's CPU) can be
    # told from a slow GPU step.
    import gc
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    host_t = [0.0] * (args.steps + 1)
    t0 = time.perf_counter()
    marks[0].record()
    host_t[0] = t0
    for i in range(args.steps):
        ts.step(args.warmup + i)
        if i + 1 in bounds[1:]:
            marks[bounds.index(i + 1)].record()
        host_t[i + 1] = time.perf_counter()
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    gaps = [(host_t[i + 1] - host_t[i]) * 1e3 for i in range(args.steps)]
    gap_thr = max(2.0, 5.0 * elapsed * 1e3 / max(args.steps, 1))
    host_stalls = {"gc": "collected before, disabled inside the timed region", "max_step_gap_ms": round(max(gaps), 3) if gaps else 0.0,
                   "threshold_ms": round(gap_thr, 2), "gaps_over_threshold": [[i, round(g, 2)] for i, g in enumerate(gaps) if g > gap_thr][:8]}
    chunk_ms = [marks[k].elapsed_time(marks[k + 1]) / max(bounds[k + 1] - bounds[k], 1) for k in range(n_chunks)]
    loss = float(ts.loss_out.item())
    if loss != loss or abs(loss) == float("inf"):
        raise SystemExit(f"bench.py: training diverged (final loss {loss}) -- the timing would be meaningless")
    # A short timed region (the driver's default call asks for 20 steps = 15 ms of work) is honoured exactly -- `value` is
    # those K steps -- and a 200-step measurement of the same loop is reported beside it so that the line can be judged
    # against run-to-run noise
    long_run = None
    if args.steps < 100:
        k_long = 200
        torch.cuda.synchronize()
        if ddp:
            dist.barrier()
        t1 = time.perf_counter()
        for i in range(k_long):
            ts.step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        if ddp:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        if ddp:
            tl = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tl, op=dist.ReduceOp.MAX)
            el = float(tl.item())
        long_run = dict(steps=k_long, ms_per_step=el / k_long * 1e3, value=args.batch_size * world * k_long / el)
    # ---- what a replay boundary costs (VERDICT r5 weak #12 asked for graph wall and inter-replay gap separately): the same loop
    # with EIGHT steps per graph replay (train.TrainStep.step_group: nothing of a step lives on the host, so k steps can be one
    # graph) against the one-step replays above.  Reported beside the headline, which stays on one replay per step.
    grouped_run = None
    if (not args.no_graph and name != "big" and getattr(ts, "fused_opt", False) and ts.sched_dev is not None
            and os.environ.get("MOBGT_BENCH_NO_GROUPED") != "1"):
        try:
            kg, n_g = 8, 200
            base = args.warmup + args.steps + (200 if args.steps < 100 else 0)
            for s in range(0, 2 * kg * max(1, len(batches) // kg + 1), kg):     # capture every group key outside the timing
                ts.step_group(base + s, kg)
            torch.cuda.synchronize()
            if ddp:
                dist.barrier()
            tg = time.perf_counter()
            for s in range(0, n_g, kg):
                ts.step_group(base + s, kg)
            torch.cuda.synchronize()
            if ddp:
                dist.barrier()
            torch.cuda.synchronize()
            el_g = time.perf_counter() - tg
            torch.cuda.synchronize()
            ts1 = time.perf_counter()
            for s in range(n_g):
                ts.step(base + s)
            torch.cuda.synchronize()
            el_1 = time.perf_counter() - ts1
            grouped_run = dict(steps=n_g, steps_per_replay=kg, ms_per_step=el_g / n_g * 1e3, value=args.batch_size * world * n_g / el_g,
                               ms_per_step_one_replay_per_step=el_1 / n_g * 1e3,
                               replay_boundary_us=round((el_1 - el_g) / n_g * kg / (kg - 1) * 1e6, 2),
                               note="k steps as ONE graph replay (no host round trip between them) vs one replay per step, same batches, back to back")
        except Exception as e:                       # never lose the headline line over a secondary figure
            grouped_run = dict(error=repr(e))
    # peer waits that gave up (csrc/chain.hip WS_FAULT & co.): a step whose cluster lost co-residency carries on with garbage
    # sums -- a number measured over such steps is not a measurement.  Polled once, behind the timed region.
    peer_faults = ts.check_faults(on_fault="return")
    if peer_faults:
        raise SystemExit(f"bench.py: workgroups gave up waiting for their peers during the timed steps {peer_faults} -- the gradients "
                         "of those steps are invalid and so is the timing")
    rccl_ranks, comm_ranks, comm_backend, exposed_us = None, None, None, None
    if ddp:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        comm_backend = str(dist.get_backend())
        comm_ranks = int(one.item())                 # ranks that took part in an all-reduce on the data-path backend
        rccl_ranks = comm_ranks if comm_backend == "nccl" else None     # ("nccl" IS RCCL on ROCm; a gloo run is not one)
        # exposed all-reduce time: the same steps once more with the gradient exchange switched off
        k2 = min(args.steps, 100)
        times = []
        for comm in (True, False):
            ts.comm = comm
            for i in range(len(batches)):             # (the step graphs without the exchange are captured on first use: not in the timing)
                ts.step(args.warmup + args.steps + i)
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for i in range(k2):
                ts.step(args.warmup + args.steps + i)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t1) / k2)
        ts.comm = True
        tt = torch.tensor(times, device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        exposed_us = float((tt[0] - tt[1]).item()) * 1e6

    if rank == 0:
        m = w["model"]
        H, C = m["num_heads"], m["hidden_dim"] + (0 if stock else 64)
        # (round 4: this leg runs FIRST, right behind the timed region.  Behind the roofline legs below -- graph-timed attention /
        # chain launches over > 1 GB of rotating buffers -- the same loop measured 0.70 instead of 0.63 ms per step while the
        # replay of the same graphs without new input stayed at 0.61; the cause was not found: two round-4 probes (docs/NOTEBOOK.md),
        # which run the c5 stress measurement in front of the loop, do not reproduce it.)
        # ---- the same step fed like the reference feeds it: a NEW batch every step (data.py:282-295), collated inside the
        # replayed step (train.EpochLoop: raw trajectories -> pinned staging -> one H2D copy -> [DeviceCollator.finish +
        # forward + loss + backward + AdamW] as one graph per shape bucket).  Secondary metric of SURVEY 8(d): check-ins/s
        # INCLUDING collate / preprocess; its CPU counterpart is cpu_baseline.with_collate.
        with_collate = None
        # (not for `big`: its raw format -- the reference's pickles hold DENSE N x N int64 count matrices, gen_pickles.py:820-832 --
        # is 16 x 784^2 x 8 B = 79 MB of host arrays per batch, and packing them takes the host 140 ms per step: a statement
        # about numpy, not about this path)
        if world == 1 and not force_comm and not args.no_loop and not args.no_graph and not stock and name != "big":
            try:
                with_collate = time_epoch_loop(model, coll, name, uni, args)
            except Exception as e:                       # never lose the headline line over the secondary figure
                with_collate = dict(error=repr(e))
        d = C // H
        io_dt = torch.bfloat16 if (bf16 and args.gemm_dtype == "bf16") else torch.float32
        b_dt = torch.bfloat16 if bf16 else torch.float32
        s_x, s_b = (2 if io_dt == torch.bfloat16 else 4), (2 if bf16 else 4)
        s_g = 2 if bf16 else 8
        p_att = m["attention_dropout_rate"]
        # dominant hand kernel of the named path: the bias-fused attention forward (training instantiation: dropout on),
        # at the shapes the timed region ran
        used = [shapes[(args.warmup + i) % len(shapes)] for i in range(args.steps)]
        uniq = sorted(set(used))
        timed = {s: time_attention(s[0], H, s[1], d, io_dt, b_dt, p_drop=p_att, reps=50 if s[1] < 400 else 24) for s in uniq}
        dur = {s: v[0] for s, v in timed.items()}
        tot_b = sum(attn_fwd_bytes(g, t, C, H, s_x, s_b) for g, t in used)
        tot_t = sum(dur[s] for s in used)
        achieved = tot_b / tot_t / 1e9
        # PMC counters of the attention kernels, measured in this run by rocprofv3 child processes (separate --pmc passes,
        # corrected per the guide): at the most frequent timed shape and at the c5 stress shape.  Fallback (no rocprofv3,
        # --no-live-pmc): the figures of profiles/attn_pmc.json, labelled as such.
        mode_shape = max(uniq, key=lambda sh: (used.count(sh), sh[1]))
        pmc_live, pmc_file = None, {}
        if bf16 and io_dt == torch.bfloat16 and not args.no_live_pmc and world == 1 and not force_comm:
            want = [(mode_shape[0], mode_shape[1], d)]
            if not args.no_stress and d != 32:
                want.append((16, 785, 32))
            pmc_live = live_pmc(want, p_att, keep_dir=os.path.join(ROOT, "gpurun_out", "pmc_live"))
        if pmc_live is None:
            try:
                pmc_file = json.load(open(os.path.join(ROOT, "profiles", "attn_pmc.json")))
            except Exception:
                pass

        def pmc_of(dd, kern, file_key):
            """(traffic bytes, MFMA-busy %, source) of one kernel"""
            if pmc_live is not None and dd in pmc_live and kern in pmc_live[dd]:
                e = pmc_live[dd][kern]
                return e["traffic_bytes"], e["mfma_busy_pct"], "live: rocprofv3 --pmc child passes of this run", e
            e = pmc_file.get(file_key, {}) if bf16 else {}
            return e.get("traffic_bytes"), e.get("mfma_busy_pct"), ("profiles/attn_pmc.json (earlier run)" if e else None), e
        big_shape = name == "big"
        tr, mb, src, _ = pmc_of(d, "fwd", "c5_fwd_drop_bf16" if big_shape else ("fsq_attn_fwd_drop_bf16" if name == "fsq" else "-"))
        roof = dict(kernel="attn_fwd_kernel<DROP=true>", bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=tr, mfma_busy_pct=mb, counters_source=src,
                    counters_shape="G%d T%d d%d" % (mode_shape[0], mode_shape[1], d),
                    bytes_per_launch=tot_b / len(used), avg_launch_us=tot_t / len(used) * 1e6,
                    input_sets_rotated=max(v[2] for v in timed.values()))
        # ... and the kernel that now takes the largest share of the timed step: the row-local chain of an encoder layer
        roofc = None
        F = m["ffn_dim"]
        if io_dt == torch.bfloat16 and (C, F) in ((192, 1024), (256, 1024)) and not args.unfused and not stock:
            rows = sorted(set(g * t for g, t in used))
            durc = {r: time_chain(r, C, F, reps=50 if r < 4096 else 10, p_drop=m["dropout_rate"]) for r in rows}
            tb = sum(chain_fwd_bytes(g * t, C, F) for g, t in used)
            tt_ = sum(durc[g * t] for g, t in used)
            if not pmc_file:
                try:
                    pmc_file = json.load(open(os.path.join(ROOT, "profiles", "attn_pmc.json")))
                except Exception:
                    pass
            cl = all(g * t <= 16 * 128 for g, t in used)             # (rows up to which the chain kernels run their cluster form)
            pm, pm_src = None, None
            if C == 192 and not args.no_live_pmc and world == 1 and not force_comm:
                mode_rows = max(rows, key=lambda r: sum(1 for g, t in used if g * t == r))
                pm = live_chain_pmc(mode_rows, keep_dir=os.path.join(ROOT, "gpurun_out", "pmc_live"))
                pm_src = "live: rocprofv3 --pmc child passes of this run (tools/chain_pmc.py, stand-alone launches behind a 64 MB filler)"
            if pm is None:
                pm = pmc_file.get("fsq_chain_fwd", {}) if (name == "fsq" and C == 192 and pmc_file) else {}
                pm_src = "profiles/attn_pmc.json (tools/chain_pmc.sh, stand-alone launches at R = %s)" % pm.get("rows") if pm else None
            roofc = dict(kernel="layer_chain_fwd_cl_kernel (cluster form: 4 / 2 workgroups per 16-row block)" if cl else "layer_chain_fwd_kernel",
                         bound="hbm", achieved=tb / tt_ / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                         frac=tb / tt_ / 1e9 / HBM_PEAK_GBS,
                         traffic=pm.get("traffic_bytes"), traffic_rows=pm.get("rows"),
                         counters_source=pm_src,
                         bytes_per_launch=tb / len(used),
                         avg_launch_us=tt_ / len(used) * 1e6,
                         note="not HBM-bound at this size: per workgroup a chain of latency-bound phases (first touch, weight "
                              "stream through one L1, two LayerNorms, one hand-over); see DESIGN.md 3.5")
        gow_tail = None
        if name == "gow" and not args.no_tail and world == 1 and not force_comm and not stock and not args.no_graph and bf16:
            try:
                gow_tail = time_gow_tail(model, coll, uni, ts, args, w)
            except Exception as e:                       # never lose the line over a secondary figure
                gow_tail = dict(error=repr(e))
        roof5 = roof5b = None
        if not args.no_stress:
            # the same kernels at the HBM-roofline stress shape (BASELINE configs[4]: G16 x 784 nodes, C 256, d 32), training
            # instantiation (attention dropout 0.1, bf16 dBias slices): forward, and both backward passes together; the
            # launches rotate over distinct input sets (see time_attention) -- the in-step, cold-cache figure
            t5f, t5b, nset5 = time_attention(16, 8, 785, 32, b_dt, b_dt, reps=24, p_drop=0.1, backward=True)
            b5f = attn_fwd_bytes(16, 785, 256, 8, s_b, s_b)
            b5b = attn_bwd_bytes(16, 785, 256, 8, s_b, s_b, s_g)
            tr, mb, src, e = pmc_of(32, "fwd", "c5_fwd_drop_bf16")
            roof5 = dict(kernel="attn_fwd_kernel<DROP=true>", workload="c5 G16 T785 C256 d32, dropout 0.1", bound="hbm",
                         achieved=b5f / t5f / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=b5f / t5f / 1e9 / HBM_PEAK_GBS,
                         traffic=tr, mfma_busy_pct=mb, valu_busy_pct=e.get("valu_busy_pct"), wait_any_frac=e.get("wait_any_frac"),
                         counters_source=src, avg_launch_us=t5f * 1e6, bytes_per_launch=b5f, input_sets_rotated=nset5,
                         first_replay_after_idle_us=LAST_ATTN_FIRST_US["fwd"],
                         clock_state="sustained: the graph of 24 launches is replayed until 60 ms of continuous execution lie behind "
                                     "the device, then the median of five replays (bench._graph_time); first_replay_after_idle_us is "
                                     "what one replay out of idle reads",
                         cache_state="all inputs rotated through > 768 MB: every launch reads cold HBM (a lower bound of the S-BIG "
                                     "step, where the one bias all 12 layers share is partly still in the Infinity Cache: "
                                     "profiles/r3_bench_big_step_summary.txt)")
            from mobgt_amd import ops as _ops
            one = bool(_ops._ATTN_ONE_PASS[0]) and b_dt == torch.bfloat16
            if one:
                b5b = attn_bwd_bytes(16, 785, 256, 8, s_b, s_b, s_g, one_pass=True)
                tro, mbo, srco, eo = pmc_of(32, "one", "-")
                roof5b = dict(kernel="attn_bwd_one_kernel + attn_dq_finish_kernel (one pass over the bias; rowsum(dO O) inside the pass, the dQ "
                                     "accumulator re-zeroed by the finishing launch)",
                              workload="c5 G16 T785 C256 d32, dropout 0.1", bound="hbm", achieved=b5b / t5b / 1e9, peak=HBM_PEAK_GBS,
                              unit="GB/s", frac=b5b / t5b / 1e9 / HBM_PEAK_GBS, traffic=tro, mfma_busy_pct={"one": mbo},
                              valu_busy_pct=eo.get("valu_busy_pct"), wait_any_frac=eo.get("wait_any_frac"),
                              counters_source=srco, avg_launch_us=t5b * 1e6, bytes_per_launch=b5b, input_sets_rotated=nset5,
                              first_replay_after_idle_us=LAST_ATTN_FIRST_US["bwd"],
                              clock_state="sustained (see roofline_stress)",
                              note="ONE backward pass (S / P / dS once per pair, transposed bias read once, dBias written once; dQ "
                                   "summed over key blocks by f32 atomics).  `frac` uses the bytes THIS algorithm has to move.  The "
                                   "pass is bound by vector issue and latency, not by HBM: DESIGN 3.1")
            else:
                trq, mbq, srcq, _ = pmc_of(32, "dq", "c5_bwd_dq_drop_bf16")
                trk, mbk, _, _ = pmc_of(32, "dkv", "c5_bwd_dkv_drop_bf16")
                roof5b = dict(kernel="attn_bwd_dq_kernel + attn_bwd_dkv_kernel", workload="c5 G16 T785 C256 d32, dropout 0.1",
                              bound="hbm", achieved=b5b / t5b / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                              frac=b5b / t5b / 1e9 / HBM_PEAK_GBS,
                              traffic=(trq + trk) if (trq is not None and trk is not None) else None,
                              mfma_busy_pct={"dq": mbq, "dkv": mbk}, counters_source=srcq,
                              avg_launch_us=t5b * 1e6, bytes_per_launch=b5b, input_sets_rotated=nset5)
        parity = None
        if not args.no_parity and uni.distance is not None:
            parity = oracle_parity_stock(model, batches, n_layers) if stock else oracle_parity(model, batches, uni, n_layers)
        cpu = None
        if not args.no_cpu_baseline and uni.distance is not None:
            cpu = (cpu_baseline_stock(model, batches, args, n_layers) if stock
                   else cpu_baseline(model, batches, list(mine), uni, args, n_layers))
        subs = None
        if world == 1 and not force_comm and name == "fsq" and not stock and not args.no_sub and not args.no_graph and not args.unfused and bf16:
            subs = run_sub_workloads(args)
        G_total = args.batch_size * world
        out = {
            "metric": "check-ins/sec (train step)", "value": G_total * args.steps / elapsed, "unit": "check-ins/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workloads.describe(name, args.pois, args.variant), "variant": args.variant,
                       "global_batch": G_total, "per_gpu_batch": args.batch_size,
                       "padded_nodes_per_batch": [s[1] - 1 for s in shapes], "parallelism": f"dp{world}",
                       "gemm_autotune": "torch TunableOp (hipBLASLt algorithm per shape)" if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1" else "off",
                       "hip_graphs": not args.no_graph, "fused_encoder_layers": not args.unfused,
                       "precision": {"attention_mfma_operands": args.dtype, "attn_bias": args.dtype,
                                     "attention_io": "bf16" if io_dt == torch.bfloat16 else "f32",
                                     "gcn_adjacency_product": args.dtype, "library_gemms": args.gemm_dtype if bf16 else "f32",
                                     "accumulate_softmax_layernorm_adamw": "f32"}},
            "final_loss": loss, "ms_per_step_chunks": [round(c, 4) for c in chunk_ms], "host_stalls": host_stalls, "long_run": long_run, "grouped_run": grouped_run,
            "value_with_collate": with_collate,
            "comm_backend": comm_backend, "comm_ranks": comm_ranks, "grad_comm_dtype": args.grad_comm if ddp else None, "forced_comm": force_comm or None, "ddp_one_graph": bool(getattr(ts, "one_graph", False)) if ddp else None, "ddp_form": ddp_form, "ddp_form_ms_per_step": ddp_form_ms, "rccl_ranks": rccl_ranks, "allreduce_exposed_us": exposed_us,
            "parity": parity, "roofline": roof, "roofline_chain": roofc, "roofline_stress": roof5, "roofline_stress_bwd": roof5b,
            "roofline_gow_tail": gow_tail,
            "cpu_baseline": cpu, "workloads": subs,
        }
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
