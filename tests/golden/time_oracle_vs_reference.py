"""Build-container check behind bench.py's `cpu_baseline`: the oracle (`oracle/model_oracle.py`, the CPU
restatement that is timed on the GPU box as the baseline) must neither sandbag nor flatter the reference.  This script
imports the reference itself (as the fixture generators do), builds the fq Graphormer of BOTH on one synthetic universe
with identical weights, and times the same train step -- forward + GradientTailLoss + backward + AdamW, train mode -- on
identical pre-collated batches, at 8 torch threads.

    python tests/golden/time_oracle_vs_reference.py [P] [steps]

Prints the two step times and their ratio, and writes tests/golden/oracle_vs_reference_timing.json.  Measured in the build
container (8 cores): see the committed json.  /root/reference is needed: this never runs on the GPU box.
"""
import copy
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import  # noqa: E402

_ref_import.install()
from make_golden import write_universe, in_ws  # noqa: E402
from make_golden_model import build_items, cpu_cuda_alias  # noqa: E402
from inputs import fill_params  # noqa: E402
from mobgt_amd import synth  # noqa: E402
from oracle import model_oracle as mo  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    import collator as rcoll
    import model_fqandtoyo as rfq
    torch.set_num_threads(8)
    uni = synth.make_universe(P=P, n_cat=50, n_user=1080, seed=1)
    write_universe(uni)
    args = dict(n_layers=6, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                ffn_dim=1024, dataset_name="foursquaregraph", warmup_updates=40000, tot_updates=400000, peak_lr=2e-4,
                end_lr=1e-9, edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.1)
    batches = []
    for i in range(3):
        trajs = synth.make_batch_of_trajectories(seed=1000 + i, G=16, P=P, n_user=1080, cat_of_poi=uni.cat_of_poi, hi=256)
        items = build_items(trajs)
        with in_ws():
            b = rcoll.collator_foursquare(copy.deepcopy(items), max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
        batches.append(b)
    with in_ws():
        ref = rfq.Graphormer(**args).train()
    fill_params(ref, 78)
    nb = ref.poi_pos_encoder.weight.shape[0]
    for b in batches:
        b.poi_pos = b.poi_pos.clamp(max=nb - 1)                  # the reference's latent OOB
    consts = mo.fq_constants(uni, "foursquaregraph")
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in ref.state_dict().items()}
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=2e-4, weight_decay=0.01)
    opt_or = torch.optim.AdamW(list(sd.values()), lr=2e-4, weight_decay=0.01)
    kw = dict(n_layers=6, H=8, D=20, p=0.1, p_in=0.1, p_att=0.1, training=True)

    def ref_step(b):
        opt_ref.zero_grad(set_to_none=True)
        with cpu_cuda_alias():
            loss = ref.training_step(b, 0)
        loss.backward()
        opt_ref.step()

    def oracle_step(b):
        opt_or.zero_grad(set_to_none=True)
        loss = mo.fq_training_loss(sd, b, consts, **kw)
        loss.backward()
        opt_or.step()

    def timeit(fn):
        fn(batches[0])                                          # warm-up
        t0 = time.perf_counter()
        for i in range(steps):
            fn(batches[i % len(batches)])
        return (time.perf_counter() - t0) / steps
    # interleave so that neither side gets the quieter half of the run
    t_ref, t_or = [], []
    for _ in range(2):
        t_ref.append(timeit(ref_step))
        t_or.append(timeit(oracle_step))
    tr, to = min(t_ref), min(t_or)
    out = dict(P=P, steps_per_run=steps, threads=8, padded_nodes=[int(b.x.shape[1]) for b in batches],
               reference_s_per_step=tr, oracle_s_per_step=to, oracle_over_reference=to / tr,
               reference_checkins_per_s=16 / tr, oracle_checkins_per_s=16 / to)
    print(json.dumps(out, indent=1))
    json.dump(out, open(os.path.join(HERE, "oracle_vs_reference_timing.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
