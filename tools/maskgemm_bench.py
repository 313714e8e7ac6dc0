"""Developer tool: time mobgt_mask_gemm at the S-FSQ size against the dense bf16 product it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobgt_amd import synth
from mobgt_amd.modelGNN import MaskAdj, mask_gemm
from mobgt_amd.model_fqandtoyo import calculate_laplacian_matrix
P = int(os.environ.get("P", 7856))
uni = synth.make_universe(P=P, n_cat=300, n_user=8, seed=1)
adj = MaskAdj(*[t.cuda() for t in MaskAdj.from_dense01(uni.graph_dist)])
dense = torch.from_numpy(calculate_laplacian_matrix(uni.graph_dist)).float().cuda().bfloat16()
for N in (16, 64):
    x = torch.randn(P, N).cuda()
    xb = x.bfloat16()
    t1 = bench._graph_time(lambda: mask_gemm(adj, x), 20)
    t2 = bench._graph_time(lambda: mask_gemm(adj, x, transposed=True), 20)
    t3 = bench._graph_time(lambda: dense @ xb, 20)
    print(f"P={P} N={N}: mask_gemm {t1*1e6:.1f} us, transposed {t2*1e6:.1f} us, dense bf16 {t3*1e6:.1f} us")
