"""Worker of tests/test_gpu_train.py::test_two_rank_train_step_*: one data-parallel rank of `train.TrainStep`.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment (torch.distributed.run or the test's Popen).
Backend: RCCL ("nccl") with one GPU per rank when the box has >= WORLD_SIZE GPUs, otherwise gloo with all ranks sharing
cuda:0 (the exchange then goes through host memory -- same TrainStep code path, two-phase overlap included).
Writes rank<r>.pt with the loss history, the flat parameter / gradient buffers and both Adam moments."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_dir, steps = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    multi = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if multi else 0)
    torch.cuda.set_device(dev)
    from mobgt_amd.train import recommended_env
    for k_, v_ in recommended_env().items():
        os.environ.setdefault(k_, v_)
    if multi:
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    from mobgt_amd import synth
    from mobgt_amd.data import DeviceCollator, make_bin_table
    from mobgt_amd.model_fqandtoyo import Graphormer
    from mobgt_amd.train import TrainStep, broadcast_parameters
    args = dict(n_layers=2, num_heads=8, hidden_dim=64, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                ffn_dim=128, warmup_updates=4, tot_updates=100, peak_lr=1e-3, end_lr=1e-9, edge_type="multi_hop",
                multi_hop_max_dist=20, attention_dropout_rate=0.1, dataset_name="foursquaregraph")
    uni = synth.make_universe(P=400, n_cat=12, n_user=1080, seed=3)
    nb, _, table = make_bin_table(uni.distance)
    torch.manual_seed(100 + rank)                         # different init per rank, equalised by the broadcast
    model = Graphormer(universe=uni, num_bins=nb + 2, bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16,
                       act_dtype=torch.bfloat16, **args).to(dev)
    broadcast_parameters(model)
    coll = DeviceCollator(dev, bin_table=table)
    # (MOBGT_TEST_DATA_RANK: a ONE-rank run on the data another rank of a two-rank run sees -- the mean-of-gradients test)
    drank = int(os.environ.get("MOBGT_TEST_DATA_RANK", rank))
    batches = [coll(synth.make_batch_of_trajectories(seed=10 + 7 * drank + i, G=4, P=400, n_user=1080, cat_of_poi=uni.cat_of_poi))
               for i in range(2)]                         # different data per rank
    comm = os.environ.get("MOBGT_TEST_GRAD_COMM")
    ts = TrainStep(model, batches, use_graph=True, seed=5, grad_comm_dtype=torch.bfloat16 if comm == "bf16" else None)
    ts.prepare()
    losses = [float(ts.step(i)) for i in range(steps)]
    torch.cuda.synchronize()
    ts.check_faults(on_fault="raise")                      # (two ranks may share one GPU here: a lost co-residency must be loud)
    torch.save(dict(losses=losses, params=ts.flat_params.tensor.detach().cpu(), grads=ts.flat.flat.cpu(),
                    exp_avg=ts.exp_avg.cpu(), exp_avg_sq=ts.exp_avg_sq.cpu(), overlap=ts.overlap, backend=dist.get_backend(),
                    forced=ts.force_comm, one_graph=ts.one_graph, parts=len(ts.parts), comm_dtype=str(ts.comm_buf.dtype) if ts.comm_buf is not None else None,
                    shadow=None if ts.shadow_flat is None else ts.shadow_flat.float().cpu()),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
