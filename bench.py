#!/usr/bin/env python3
"""Headline benchmark: check-ins/sec of one MobGT train step (+ attention-kernel HBM GB/s vs roofline).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY §8d "S-FSQ"): the fq Graphormer (`model_fqandtoyo`, the class
`entry.py` runs) on a synthetic Foursquare-TKY-sized universe (P = 7856 POIs, 300 categories, 1080 users),
hidden_dim 128, 6 layers, 8 heads, ffn 1024, multi_hop_max_dist 20, 16 trajectories per GPU per step,
README hyper-parameters (dropout 0.1 everywhere, AdamW, PolynomialDecayLR).  A "step" = forward +
GradientTailLoss + backward (+ gradient all-reduce) + AdamW on one pre-collated batch resident in HBM.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The train step's library GEMMs (tall-skinny, a few hundred rows) are sensitive to hipBLASLt's algorithm choice:
# let PyTorch's TunableOp pick per shape during the eager warm-up passes that precede graph capture (<= 30 ms of
# trials per new shape, ~20 s for the ~120 shapes of the default run; measured +8.6 % check-ins/s).  Must be set
# before torch is imported; `--no-gemm-autotune` (or the variables themselves) turns it off.
if "--no-gemm-autotune" not in sys.argv:
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

MODEL_ARGS = dict(n_layers=6, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                  ffn_dim=1024, dataset_name="foursquaregraph", warmup_updates=40000, tot_updates=400000, peak_lr=2e-4,
                  end_lr=1e-9, edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--n-batches", type=int, default=8, help="distinct pre-collated batches cycled through")
    ap.add_argument("--pois", type=int, default=7856)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--gemm-dtype", choices=["bf16", "f32"], default="bf16",
                    help="dtype of the library GEMMs (projections / FFN / head); attention MFMA operands, the bias "
                         "and the GCN adjacency product follow --dtype")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="op-by-op encoder layers (torch ops + HIP attention)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one all-reduce after the whole backward instead of two overlapped buckets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-autotune", action="store_true", help="leave hipBLASLt's default algorithm choice (no TunableOp)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--seed", type=int, default=1)
    return ap.parse_args()


def attn_algorithmic_bytes(G, T, C, H, s_x, s_b):
    """read Q,K,V + write O, read bias, write log-sum-exp."""
    return G * (4 * T * C * s_x + H * T * T * s_b + H * T * 4)


def time_attention_kernel(G, H, T, d, io_dtype, bias_dtype, reps=50, p_drop=0.0):
    """Average duration (s) of one mobgt_attn_bias_fwd launch, HIP events on the launching stream."""
    from mobgt_amd import ops
    C = H * d
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = torch.randn(G, T, 3 * C, generator=g).to(dev).to(io_dtype)
    bias = torch.randn(G, H, T, T, generator=g).to(dev)
    pack = ops.pack_bias(bias, G, H, T, dtype=bias_dtype)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    # the launches are captured in a hipGraph so that the events bracket back-to-back kernels rather than the
    # Python/ctypes launch path (a ~3 us kernel would otherwise read as ~12 us of host time)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ops._attn_fwd(q, k, v, pack, d ** -0.5, p_drop, 1, None)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            ops._attn_fwd(q, k, v, pack, d ** -0.5, p_drop, 1, None)
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / reps


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                            # cgroup v2 CPU quota of the container, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(model, batches, uni, seconds, max_steps=40):
    """The oracle (CPU restatement of the reference, oracle/model_oracle.py) timed on the host cores for the
    same step definition on the same batches: forward + GradientTailLoss + backward + AdamW, train mode.
    torch CPU eager does not scale to hundreds of threads on these small ops (256 threads measured 1000x
    slower than 8), so the thread count is probed over {8, 16, 32} <= usable cores and the fastest is used."""
    from oracle import model_oracle as mo
    from types import SimpleNamespace
    # constants as model_fqandtoyo.__init__ derives them; taken from the already-built module so that the
    # baseline does not spend a minute re-inverting the 7856^2 degree matrix (not part of a step)
    p2c = model.poi2cat.cpu().numpy()
    consts = SimpleNamespace(X=model.X.float().cpu(), D_A=model.D_A.float().cpu(), C_X=model.C_X.float().cpu(),
                             C_A=model.C_A.float().cpu(), poi2cat={i: int(c) for i, c in enumerate(p2c)})
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=2e-4, weight_decay=0.01)
    cb = []
    for b in batches:
        c = SimpleNamespace()
        for f in ("attn_bias", "rel_pos", "poi_pos", "edge_input", "x", "in_degree", "out_degree", "user", "y", "time_normal"):
            t = getattr(b, f).cpu()
            setattr(c, f, t.float() if t.dtype.is_floating_point else t.long())
        cb.append(c)
    kw = dict(n_layers=6, H=8, D=20, p=0.1, p_in=0.1, p_att=0.1, training=True)
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
    cands = [c for c in (8, 16, 32) if c <= cores] or [cores]
    best, best_t = cands[0], None
    for c in cands:                                 # probe: 1 warm-up + 1 timed step on batch 0
        torch.set_num_threads(c)
        one_step(0)
        t = one_step(0)
        if best_t is None or t < best_t:
            best, best_t = c, t
        if t > seconds:
            break
    torch.set_num_threads(best)
    one_step(0)
    n, t_used = 0, 0.0
    while n < max_steps and t_used < seconds:
        t_used += one_step(n)
        n += 1
    return dict(value=G * n / t_used, unit="check-ins/s", cores=best, kind="port",
                sample=f"{n} train steps (fwd+loss+bwd+AdamW, fp32, train mode, {best} torch threads of {cores} usable "
                       f"cores) of the oracle cycling over the same pre-collated S-FSQ batches, after warm-up; "
                       f"{t_used:.1f} s of CPU work")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # MOBGT_TEST_SHARED_GPU=1: developer switch to exercise the multi-rank code path on a ONE-GPU box (all ranks on
    # cuda:0, gloo instead of RCCL).  Never set by the driver; numbers from such a run are meaningless.
    shared = os.environ.get("MOBGT_TEST_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from mobgt_amd import synth
    from mobgt_amd.data import DeviceCollator, make_bin_table
    from mobgt_amd.model_fqandtoyo import Graphormer
    from mobgt_amd.train import TrainStep, broadcast_parameters

    bf16 = args.dtype == "bf16"
    torch.manual_seed(args.seed)
    uni = synth.make_universe(P=args.pois, n_cat=300, n_user=1080, seed=args.seed)
    num_bins, _, table = make_bin_table(uni.distance)
    model = Graphormer(universe=uni, num_bins=num_bins + 2, bias_dtype=torch.bfloat16 if bf16 else torch.float32,
                       gcn_dtype=torch.bfloat16 if bf16 else torch.float32,
                       act_dtype=torch.bfloat16 if (bf16 and args.gemm_dtype == "bf16") else torch.float32,
                       fused_layers=not args.unfused, **MODEL_ARGS).to(dev)
    broadcast_parameters(model)
    coll = DeviceCollator(dev, bin_table=table, multi_hop_max_dist=20, rel_pos_max=1024)
    # Length-bucketed sharding (SURVEY §8e hazard): every rank draws the SAME pool of world x n_batches batches,
    # the pool is ordered by padded size and dealt round-robin, so that at each synchronous step all ranks work
    # on batches of neighbouring size (per-rank work is still n_batches x 16 trajectories: weak scaling).
    pool = []
    for i in range(args.n_batches * world):
        trajs = synth.make_batch_of_trajectories(seed=1000 + i, G=args.batch_size, P=args.pois, n_user=1080,
                                                 cat_of_poi=uni.cat_of_poi, hi=256)
        pool.append((max(len(t["node_name"]) for t in trajs), i, trajs))
    pool.sort(key=lambda e: (e[0], e[1]))
    order = sorted(range(args.n_batches), key=lambda j: pool[j * world][1])   # size-mixed order over time
    mine = [pool[j * world + rank] for j in order]                            # same slot order on all ranks
    batches, shapes = [], []
    for _, _, trajs in mine:
        b = coll(trajs)
        batches.append(b)
        shapes.append((len(b), b.x.shape[1] + 1))
    torch.cuda.synchronize()

    ts = TrainStep(model, batches, autocast_dtype=torch.bfloat16 if (bf16 and args.gemm_dtype == "bf16" and args.unfused) else None, use_graph=not args.no_graph, overlap=not args.no_overlap,
                   seed=args.seed)
    ts.prepare()
    for i in range(args.warmup):
        ts.step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts.step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    loss = float(ts.loss_out.item())
    if loss != loss or abs(loss) == float("inf"):
        raise SystemExit(f"bench.py: training diverged (final loss {loss}) -- the timing would be meaningless")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        H, d, C = 8, 24, 192
        io_dt = torch.bfloat16 if (bf16 and args.gemm_dtype == "bf16") else torch.float32
        b_dt = torch.bfloat16 if bf16 else torch.float32
        s_x, s_b = (2 if io_dt == torch.bfloat16 else 4), (2 if bf16 else 4)
        # dominant hand kernel of the named path: the bias-fused attention forward, at the shapes the timed region ran
        used = [shapes[(args.warmup + i) % len(shapes)] for i in range(args.steps)]
        uniq = sorted(set(used))
        dur = {s: time_attention_kernel(s[0], H, s[1], d, io_dt, b_dt, p_drop=0.1) for s in uniq}
        tot_b = sum(attn_algorithmic_bytes(g, t, C, H, s_x, s_b) for g, t in used)
        tot_t = sum(dur[s] for s in used)
        achieved = tot_b / tot_t / 1e9
        roof = dict(kernel="attn_fwd_kernel", bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=None,
                    bytes_per_launch=tot_b / len(used), avg_launch_us=tot_t / len(used) * 1e6)
        # the same kernel at the HBM-roofline stress shape (BASELINE configs[4]: G16 x 784 nodes, C 256, d 32)
        t5 = time_attention_kernel(16, 8, 785, 32, b_dt, b_dt, reps=30)
        b5 = attn_algorithmic_bytes(16, 785, 256, 8, s_b, s_b)
        traffic5 = None
        try:        # PMC-measured HBM bytes per launch (profiles/attn_pmc.json: separate rocprofv3 --pmc passes, corrected)
            if bf16:
                traffic5 = json.load(open(os.path.join(ROOT, "profiles", "attn_pmc.json")))["c5_G16_H8_T785_d32_bf16"]["traffic_bytes"]
        except Exception:
            pass
        roof5 = dict(kernel="attn_fwd_kernel", workload="c5 G16 T785 C256 d32", bound="hbm", achieved=b5 / t5 / 1e9,
                     peak=HBM_PEAK_GBS, unit="GB/s", frac=b5 / t5 / 1e9 / HBM_PEAK_GBS, traffic=traffic5,
                     avg_launch_us=t5 * 1e6, bytes_per_launch=b5)
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(model, batches, uni, args.cpu_seconds)
        G_total = args.batch_size * world
        out = {
            "metric": "check-ins/sec (train step)", "value": G_total * args.steps / elapsed, "unit": "check-ins/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "S-FSQ (BASELINE configs[1]): model_fqandtoyo Graphormer, foursquaregraph, P=%d, "
                                   "hidden 128 (C=192, d=24), 6 layers, 8 heads, ffn 1024, multi_hop_max_dist 20, "
                                   "dropout 0.1, fwd+GradientTailLoss+bwd+allreduce+AdamW" % args.pois,
                       "global_batch": G_total, "per_gpu_batch": args.batch_size,
                       "padded_nodes_per_batch": [s[1] - 1 for s in shapes], "parallelism": f"dp{world}",
                       "gemm_autotune": "torch TunableOp (hipBLASLt algorithm per shape)" if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1" else "off", "hip_graphs": not args.no_graph, "fused_encoder_layers": not args.unfused,
                       "precision": {"attention_mfma_operands": args.dtype, "attn_bias": args.dtype,
                                     "attention_io": "bf16" if io_dt == torch.bfloat16 else "f32",
                                     "gcn_adjacency_product": args.dtype, "library_gemms": args.gemm_dtype if bf16 else "f32",
                                     "accumulate_softmax_layernorm_adamw": "f32"}},
            "final_loss": loss,
            "roofline": roof, "roofline_stress": roof5, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
