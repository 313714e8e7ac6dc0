"""Developer tool (GPU box): run the dropout-on fq layer parity test body with the allocator's free blocks pre-filled with NaN /
huge values: an uninitialised read shows up as NaN or as a changed result."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
fill = sys.argv[1] if len(sys.argv) > 1 else "nan"
blocks = [torch.empty(n, device="cuda") for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12) for _ in range(4)]
for b in blocks:
    b.fill_(float("nan") if fill == "nan" else 3.0e4)
del blocks
import pytest
sys.exit(pytest.main(["-x", "-q", os.path.join(ROOT, "tests/test_gpu_train_parity.py"), "-k", "f32 or bf16_launches"]))
